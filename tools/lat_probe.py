"""Separate in-kernel time from launch gaps: run under rocprofv3 --kernel-trace and compare the
trace's kernel durations with the wall time per launch of a hipGraph replay."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from slotformer_amd import ops
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from gemm_bench import timeit
dev = torch.device('cuda:0')
x = torch.randn(1344, 256, device=dev); w = torch.randn(768, 256, device=dev) / 16; b = torch.randn(768, device=dev)
g, be = torch.ones(256, device=dev), torch.zeros(256, device=dev)
xs = torch.randn(224, 128, device=dev); ws = torch.randn(128, 128, device=dev) / 11
gs, bs = torch.ones(128, device=dev), torch.zeros(128, device=dev)
for cfg in (4, 16, 2):
    os.environ['SF_GEMM_CFG'] = str(cfg)
    print('qkv ln cfg', cfg, 'replay us/launch', round(timeit(lambda: ops.linear(x, w, b, ln=(g, be))), 2))
    print('qkv noln cfg', cfg, 'replay us/launch', round(timeit(lambda: ops.linear(x, w, b)), 2))
os.environ['SF_GEMM_CFG'] = '3'
print('q_sa ln replay', round(timeit(lambda: ops.linear(xs, ws, None, ln=(gs, bs))), 2))
print('q_sa noln replay', round(timeit(lambda: ops.linear(xs, ws, None)), 2))
print('layernorm tiny replay', round(timeit(lambda: ops.layernorm(xs, gs, bs)), 2))
