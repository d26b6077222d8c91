"""conv_first_kernel (3 -> 64, 5 x 5, stride 2, NCHW frames -> NHWC maps): us per launch of 192 / 32 frames at 128 x 128, GB/s, and a checksum of the output
(run with SF_LIB_PATH=<another build> to compare builds bit for bit).    python tools/conv_first_probe.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from slotformer_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
w = (0.1 * torch.randn(64, 3, 5, 5, generator=g)).to(dev)
b = (0.1 * torch.randn(64, generator=g)).to(dev)
for F_ in (192, 32, 5):
    img = torch.randn(F_, 3, 128, 128, generator=g).to(dev)
    out = ops.conv2d_first(img, w, b, 2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        ops.conv2d_first(img, w, b, 2)
    e0.record()
    for _ in range(20):
        ops.conv2d_first(img, w, b, 2)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    nbytes = img.numel() * 4 + out.numel() * 4
    print(f'frames {F_:4d}: {us:7.1f} us per launch (with the wrapper)  {nbytes / us / 1e3:7.1f} GB/s   checksum {out.double().sum().item():.10e}  {out.view(-1)[12345 % out.numel()].item():.8e}  max {out.abs().max().item():.8e}')
