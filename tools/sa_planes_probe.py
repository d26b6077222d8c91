"""One Slot-Attention iteration of 32 frames: exact-f32 tile kernel on f32 rows vs split-bf16 planes kernel on bf16 hi | lo rows; whole chip and a 128-CU mask.
   python tools/sa_planes_probe.py"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from slotformer_amd import engine, _lib  # noqa: E402
from slotformer_amd import pipeline as pl  # noqa: E402

dev = torch.device('cuda:0')
savi, roll = bench.build_models(dev, bench.bench_configs()['C2'])
lib = _lib.lib()
img = bench.synthetic_img(32, 6, 128).to(dev)
noise = torch.randn(32, 6, 7, 128, device=dev)
h = C.c_void_p()
_lib.check(lib.sf_stream_create_cu_mask(C.byref(h), (C.c_uint * 8)(*pl.ENC_WORDS_P), 8))
masked = torch.cuda.ExternalStream(h.value, device=dev)
with torch.no_grad():
    for name, st in (('whole chip', torch.cuda.current_stream()), ('128-CU mask', masked)):
        for mode in (0, 1, 2):
            lib.sf_set_slot_attn_planes(1 if mode else 0)
            lib.sf_set_pixel_tok(1 if mode == 2 else 0)
            with torch.cuda.stream(st):
                for _ in range(3):
                    engine.savi_encode(savi, img, noise=noise, side_stream=None, ws_slot=('sap', mode, name))
                st.synchronize()
                lib.sf_profile_enable(0x7f)
                bench.read_profile(lib)
                for _ in range(3):
                    engine.savi_encode(savi, img, noise=noise, side_stream=None, ws_slot=('sap', mode, name))
                st.synchronize()
                lib.sf_profile_enable(0)
            pr = bench.read_profile(lib)
            print(f'{name:12s} planes={min(mode, 1)} pixel_tok={int(mode == 2)}: ' + '  '.join(f'{k} {v["avg_us"]:.1f} us x {v["launches"] // 3}' for k, v in pr.items() if k in ('slot_attn_iter', 'slot_update', 'linear_gemm')))
