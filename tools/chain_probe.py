"""Rollout chains side by side on one CU mask, no encode: ms per rollout.  python tools/chain_probe.py [rows]"""
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
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 5
lib.sf_set_ffn_rows64(int(os.environ.get('FFN64', '0')))
words = [0xffffffff if w < rows else 0 for w in range(8)]
with torch.no_grad():
    for nroll in (1, 2, 3):
        hs, sts = [], []
        for _ in range(nroll):
            arr = (C.c_uint * 8)(*words)
            h = C.c_void_p()
            _lib.check(lib.sf_stream_create_cu_mask(C.byref(h), arr, 8))
            hs.append(h)
            sts.append(torch.cuda.ExternalStream(h.value, device=dev))
        bufs = [torch.randn(32, 56, 7, 128, device=dev) for _ in sts]
        graphs = []
        for ri, st in enumerate(sts):
            engine.rollout(roll, bufs[ri], 6, 50, ws_slot=('cp', ri))
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                engine.rollout(roll, bufs[ri], 6, 50, ws_slot=('cp', ri))
            graphs.append(g)
        torch.cuda.synchronize()
        n = 6
        for rep in range(2):
            t0 = time.perf_counter()
            for _ in range(n):
                for ri, st in enumerate(sts):
                    with torch.cuda.stream(st):
                        graphs[ri].replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f'{32 * rows} CUs, seam {os.environ.get("SF_SEAM_FUSED", "1")}, parts {os.environ.get("SF_FFN_PARTS", "1")}, ffn64 {os.environ.get("FFN64", "0")}: {nroll} chain(s): '
              f'{1e3 * dt / n:7.3f} ms per round = {1e3 * dt / n / nroll:6.3f} ms per rollout', flush=True)
        del graphs, sts
        torch.cuda.synchronize()
        for h in hs:
            lib.sf_stream_destroy(h)
