import sys, os, torch
sys.path.insert(0, '/root/repo')
from slotformer_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
b = torch.zeros(64, device=dev)
w = torch.zeros(64, 64, 5, 5, device=dev)
wf = ops.pack_conv_frag(ops.pack_conv_weight(w))
print('lib', os.environ.get('SF_LIB_PATH'))
for name, x in (('ones', torch.ones(1, 8, 64, 64, device=dev)), ('zeros', torch.zeros(1, 8, 64, 64, device=dev))):
    for nwg in (1, 8):
        out = ops.conv5x5_ws(x, wf, b, relu=False, n_workgroups=nwg)
        bad = ~(out == 0)
        print(name, 'nwg', nwg, 'bad', int(bad.sum()), 'of', out.numel(), 'per row', bad.sum(dim=(0, 2, 3)).tolist(), 'per pxblock', bad.view(1, 8, 2, 32, 64).sum(dim=(0, 1, 3, 4)).tolist(), 'per chblock', bad.view(1, 8, 64, 2, 32).sum(dim=(0, 1, 2, 4)).tolist())
w = torch.randn(64, 64, 5, 5, device=dev) * 0.03
wf = ops.pack_conv_frag(ops.pack_conv_weight(w))
x = torch.randn(2, 8, 64, 64, device=dev)
ref = ops.conv5x5_frag(x, wf, b, relu=False)
out = ops.conv5x5_ws(x, wf, b, relu=False, n_workgroups=3)
d = (out - ref).abs()
print('random: max', d.max().item(), 'bad frac', (d > 0).float().mean().item(), 'per row max', [round(v, 3) for v in d.amax(dim=(2, 3)).flatten().tolist()])
