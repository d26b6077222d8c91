"""What in the encode slows the rollout pair?  Two rollout graphs loop on the 160-CU mask while ONE kind of encode kernel
loops on the 96-CU mask:  python tools/interference_probe.py"""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from slotformer_amd import engine, ops, _lib  # noqa: E402
from slotformer_amd import pipeline as pl  # noqa: E402

dev = torch.device('cuda:0')
savi, roll = bench.build_models(dev)
lib = _lib.lib()
lib.sf_set_seam_fused(0)
lib.sf_set_ffn_rows64(1)
img = bench.synthetic_img(32).to(dev)
noise = torch.randn(32, 6, 7, 128, device=dev)


def masked(words):
    arr = (C.c_uint * 8)(*words)
    h = C.c_void_p()
    _lib.check(lib.sf_stream_create_cu_mask(C.byref(h), arr, 8))
    return torch.cuda.ExternalStream(h.value, device=dev)


with torch.no_grad():
    s_enc = masked(pl.ENC_WORDS_P)
    rolls = [masked(pl.ROLL_WORDS_P), masked(pl.ROLL_WORDS_P)]
    bufs = [torch.randn(32, 56, 7, 128, device=dev) for _ in rolls]
    graphs = []
    for ri in range(2):
        engine.rollout(roll, bufs[ri], 6, 50, ws_slot=('ip', ri))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            engine.rollout(roll, bufs[ri], 6, 50, ws_slot=('ip', ri))
        graphs.append(g)
    kf = torch.randn(32, 4096, 128, device=dev)
    q = torch.randn(32, 7, 128, device=dev)
    x64 = torch.randn(32, 64, 64, 64, device=dev)
    plan = engine.encoder_plan(savi)
    convs = [m for m in savi.encoder.modules() if isinstance(m, torch.nn.Conv2d)]
    wp = ops.pack_conv_weight(convs[1].weight.detach().float())
    cb = convs[1].bias.detach().float()

    def k_none():
        pass

    def k_conv():
        for _ in range(18):
            ops.conv2d_nhwc(x64, wp, cb)

    def k_sa():
        for _ in range(36):
            ops.slot_attn_iter(kf, kf, q)

    def k_cnn():
        engine.savi_cnn(savi, img, 0, 6, ws_slot='ip')

    def k_enc():
        engine.savi_encode(savi, img, noise=noise, ws_slot='ip')

    for name, fn in (('nothing', k_none), ('18 x conv 64->64', k_conv), ('36 x Slot-Attention iteration', k_sa), ('CNN of 6 frames', k_cnn),
                     ('full encode', k_enc)):
        torch.cuda.synchronize()
        n = 4
        for rep in range(2):
            ev = []
            for _ in range(n):
                for _k in range(2):
                    with torch.cuda.stream(s_enc):
                        fn()
                for ri, st in enumerate(rolls):
                    with torch.cuda.stream(st):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(st); graphs[ri].replay(); e1.record(st)
                        ev.append((e0, e1))
            t0 = time.perf_counter()
            s_enc.synchronize()
            torch.cuda.synchronize()
        d = sorted(a.elapsed_time(b) for a, b in ev)
        print(f'encode mask runs {name:32s}: rollout (one of two chains) median {d[len(d) // 2]:6.3f} ms  min {d[0]:6.3f}', flush=True)
