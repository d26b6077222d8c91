"""How many CU-masked queues can run side by side?  K streams on the encode masks, each encoding a share of the videos,
beside the rollout graph (free-running).  GPU_MAX_HW_QUEUES=<n> python tools/queue_probe.py"""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from slotformer_amd import engine, _lib  # noqa: E402
from slotformer_amd import pipeline as pl  # noqa: E402

dev = torch.device('cuda:0')
savi, roll = bench.build_models(dev)
lib = _lib.lib()
img = bench.synthetic_img(32).to(dev)
noise = torch.randn(32, 6, 7, 128, device=dev)
print('GPU_MAX_HW_QUEUES =', os.environ.get('GPU_MAX_HW_QUEUES'))
with torch.no_grad():
    handles = []

    def masked(words):
        arr = (C.c_uint * 8)(*words)
        h = C.c_void_p()
        _lib.check(lib.sf_stream_create_cu_mask(C.byref(h), arr, 8))
        handles.append(h)
        return torch.cuda.ExternalStream(h.value, device=dev)

    def release():
        torch.cuda.synchronize()
        while handles:
            lib.sf_stream_destroy(handles.pop())

    SE = lambda ses, rows: [sum(0xff << (8 * se) for se in ses) if w in rows else 0 for w in range(8)]
    ALL = range(8)
    parts = {
        'a roll SE1-3 r0-6 (168) | enc SE0 (64) + SE1-3 r7 (24)': (SE((1, 2, 3), range(7)), [(SE((0,), ALL), 64), (SE((1, 2, 3), (7,)), 24)]),
        'b roll SE1-3 (192) | enc SE0 (64)': (SE((1, 2, 3), ALL), [(SE((0,), ALL), 64)]),
        'c roll r0-4 (160) | enc r5-7 (96)': (SE((0, 1, 2, 3), range(5)), [(SE((0, 1, 2, 3), (5, 6, 7)), 96)]),
        'd roll SE2-3 (128) | enc SE0-1 (128)': (SE((2, 3), ALL), [(SE((0, 1), ALL), 128)]),
        'e roll SE1-3 r0-5 (144) | enc SE0 (64) + SE1-3 r6-7 (48)': (SE((1, 2, 3), range(6)), [(SE((0,), ALL), 64), (SE((1, 2, 3), (6, 7)), 48)]),
        'f roll r0-3 (128) | enc r4-7 (128)': (SE((0, 1, 2, 3), range(4)), [(SE((0, 1, 2, 3), (4, 5, 6, 7)), 128)]),
        'g roll SE1-3 r0-4 (120) | enc SE0 (64) + SE1-3 r5-7 (72)': (SE((1, 2, 3), range(5)), [(SE((0,), ALL), 64), (SE((1, 2, 3), (5, 6, 7)), 72)]),
    }
    lib.sf_set_ffn_rows64(int(os.environ.get('FFN64', '0')))
    NROLLS = [int(x) for x in os.environ.get('NROLLS', '1,2').split(',')]
    sel = sys.argv[1:] or list('abcdefg')
    for name, (rw, encs) in parts.items():
        if name[0] not in sel:
            continue
        tot = sum(c for _, c in encs)
        cuts, acc = [0], 0
        for _, c in encs:
            acc += c
            cuts.append(round(32 * acc / tot))
        lanes = [(masked(w), cuts[i], cuts[i + 1]) for i, (w, _) in enumerate(encs)]
        for nroll in NROLLS:
            rolls = [masked(rw) for _ in range(nroll)]
            bufs = [torch.randn(32, 56, 7, 128, device=dev) for _ in rolls]
            for ri, st in enumerate(rolls):
                with torch.cuda.stream(st):
                    engine.rollout(roll, bufs[ri], 6, 50, ws_slot=('qp', ri))
            torch.cuda.synchronize()
            n = 5
            for rep in range(2):
                evl, evr = [], []
                t0 = time.perf_counter()
                for _ in range(n):
                    for _k in range(nroll):
                        for li, (st, lo, hi) in enumerate(lanes):
                            with torch.cuda.stream(st):
                                post, _, _ = engine.savi_encode(savi, img[lo:hi], noise=noise[lo:hi], ws_slot=('qp', li))
                    for ri, st in enumerate(rolls):
                        with torch.cuda.stream(st):
                            engine.rollout(roll, bufs[ri], 6, 50, ws_slot=('qp', ri))
                for st, _, _ in lanes:
                    st.synchronize()
                t_enc = time.perf_counter() - t0
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            nb = n * nroll
            print(f'{name:58s} chains {nroll}: all done {1e3 * dt / nb:6.3f} ms per batch; encode lanes done after {1e3 * t_enc / nb:6.3f} ms per batch '
                  f'(videos {[hi - lo for _, lo, hi in lanes]})', flush=True)
            del rolls
        del lanes
        release()
