"""Encode on a CU-masked stream, K rollout chains on PLAIN (unmasked) streams: ms per batch, free-running."""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from slotformer_amd import engine, _lib  # noqa: E402

dev = torch.device('cuda:0')
savi, roll = bench.build_models(dev)
lib = _lib.lib()
lib.sf_set_ffn_rows64(int(os.environ.get('FFN64', '1')))
lib.sf_set_seam_fused(0)
img = bench.synthetic_img(32).to(dev)
noise = torch.randn(32, 6, 7, 128, device=dev)
with torch.no_grad():
    MIX = os.environ.get('MIX')   # "m,u": m masked rollout chains on the complement + u unmasked ones
    cfgs = ((3, 2, 0), (3, 3, 0), (3, 4, 0), (3, 3, -1), (4, 3, 0), (4, 4, 0))
    if MIX:
        m_, u_ = [int(x) for x in MIX.split(',')]
        cfgs = ((3, m_ + u_, 0), )
    for enc_rows, nroll, prio in cfgs:
        words = [0xffffffff if w >= 8 - enc_rows else 0 for w in range(8)]
        arr = (C.c_uint * 8)(*words)
        h = C.c_void_p()
        _lib.check(lib.sf_stream_create_cu_mask(C.byref(h), arr, 8))
        s_enc = torch.cuda.ExternalStream(h.value, device=dev)
        rolls = [torch.cuda.Stream(device=dev, priority=prio) for _ in range(nroll)]
        hm = []
        if MIX:
            rw = [~w & 0xffffffff for w in words]
            for i in range(m_):
                arr2 = (C.c_uint * 8)(*rw)
                h2 = C.c_void_p()
                _lib.check(lib.sf_stream_create_cu_mask(C.byref(h2), arr2, 8))
                hm.append(h2)
                rolls[i] = torch.cuda.ExternalStream(h2.value, device=dev)
        bufs = [torch.randn(32, 56, 7, 128, device=dev) for _ in rolls]
        graphs = []
        for ri in range(nroll):
            engine.rollout(roll, bufs[ri], 6, 50, ws_slot=('up', ri))
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                engine.rollout(roll, bufs[ri], 6, 50, ws_slot=('up', ri))
            graphs.append(g)
        torch.cuda.synchronize()
        n = 4
        for rep in range(2):
            t0 = time.perf_counter()
            for _ in range(n):
                for _k in range(nroll):
                    with torch.cuda.stream(s_enc):
                        engine.savi_encode(savi, img, noise=noise, ws_slot=('up', 0))
                for ri, st in enumerate(rolls):
                    with torch.cuda.stream(st):
                        graphs[ri].replay()
            s_enc.synchronize()
            t_enc = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        nb = n * nroll
        print(f'encode on {32 * enc_rows} masked CUs, {nroll} rollout chains ({MIX or "all unmasked"}; priority {prio}): all done {1e3 * dt / nb:6.3f} ms per batch; '
              f'encode done after {1e3 * t_enc / nb:6.3f} ms per batch', flush=True)
        del graphs, rolls
        torch.cuda.synchronize()
        lib.sf_stream_destroy(h)
        for h2 in hm:
            lib.sf_stream_destroy(h2)
