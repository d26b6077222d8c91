import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
from slotformer_amd import ops
from gemm_bench import timeit
dev = torch.device('cuda:0')
for name, M, N, K, ln, cfgs in [('qkv', 1344, 768, 256, True, (4, 22, 16)), ('ffn2', 1344, 256, 1024, False, (3, 24)), ('outp', 1344, 256, 256, False, (3,)), ('tiny', 224, 128, 128, True, (3,))]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K**-0.5; b = torch.randn(N, device=dev)
    g, be = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    for cfg in cfgs:
        os.environ['SF_GEMM_CFG'] = str(cfg)
        row = []
        for dbg, lab in [(0, 'full'), (1, 'noMFMA'), (2, 'noLoads'), (3, 'noMFMA+noLoads'), (4, 'noLNstats'), (8, 'noStores'), (15, 'nothing')]:
            os.environ['SF_GEMM_DBG'] = str(dbg)
            row.append(f'{lab}:{timeit(lambda: ops.linear(x, w, b, ln=(g, be) if ln else None)):.1f}')
        print(name, 'cfg', cfg, '  '.join(row))
os.environ['SF_GEMM_DBG'] = '0'
# empty-ish kernels for the floor
y = torch.randn(224, 128, device=dev); gg = torch.ones(128, device=dev)
print('layernorm 224x128:', round(timeit(lambda: ops.layernorm(y, gg, gg)), 2))
z = torch.zeros(64, device=dev)
print('torch add_ 64 elems:', round(timeit(lambda: z.add_(1.0)), 2))
