"""pixel_feat_tok_kernel (form 3 of sf_pixel_feat_f32) at the C2 shape (32 frames x 4096 pixels): us per launch whole chip, checksum.
    [SF_LIB_PATH=...] python tools/pixel_tok_probe.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from slotformer_amd import _lib  # noqa: E402

dev = torch.device('cuda:0')
lib = _lib.lib()
torch.manual_seed(0)
for frames in (32, 192):
    M = frames * 4096
    x = torch.randn(M, 64, device=dev)
    g0, b0 = 1 + 0.1 * torch.randn(64, device=dev), 0.1 * torch.randn(64, device=dev)
    w1, bb1 = 0.15 * torch.randn(128, 64, device=dev), 0.1 * torch.randn(128, device=dev)
    w2, bb2 = 0.1 * torch.randn(128, 128, device=dev), 0.1 * torch.randn(128, device=dev)
    g1, b1 = 1 + 0.1 * torch.randn(128, device=dev), 0.1 * torch.randn(128, device=dev)
    ptrs = [v.data_ptr() for v in (x, g0, b0, w1, bb1, w2, bb2, g1, b1)]
    out = torch.empty(M, 128, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: _lib.check(lib.sf_pixel_feat_f32(*ptrs, out.data_ptr(), M, 1e-5, 3, st))  # noqa: E731
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f'frames {frames}: {e0.elapsed_time(e1) * 1e3 / 30:7.1f} us per launch   checksum {out.double().sum().item():.10e} {out.abs().max().item():.8e}')
