"""Which 64 CUs?  Encode alone (C2, B=32, 6 frames) on streams whose CU mask has the SAME number of CUs but a different
bit pattern per 32-bit word, with the per-class event breakdown:  python tools/mask_probe.py [pattern ...]"""
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
img = bench.synthetic_img(32).to(dev)
noise = torch.randn(32, 6, 7, 128, device=dev)


def masked(words):
    arr = (C.c_uint * 8)(*words)
    h = C.c_void_p()
    _lib.check(lib.sf_stream_create_cu_mask(C.byref(h), arr, 8))
    return h, torch.cuda.ExternalStream(h.value, device=dev)


def run(name, words):
    h, st = masked(words) if words else (None, torch.cuda.current_stream())
    with torch.no_grad(), torch.cuda.stream(st):
        for _ in range(2):
            engine.savi_encode(savi, img, noise=noise)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            engine.savi_encode(savi, img, noise=noise)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 5
        lib.sf_profile_enable(0x7f)
        bench.read_profile(lib)
        engine.savi_encode(savi, img, noise=noise)
        torch.cuda.synchronize()
        lib.sf_profile_enable(0)
        prof = bench.read_profile(lib)
    cus = sum(bin(w).count('1') for w in words) if words else 256
    print(f'{name:28s} {cus:3d} CUs: encode {ms:6.3f} ms  ' +
          '  '.join(f'{k.split("_")[0]}:{v["avg_us"]:.0f}' for k, v in prof.items()), flush=True)
    if h is not None:
        lib.sf_stream_destroy(h)


def run_roll(name, words):
    from slotformer_amd.pipeline import EncodeRolloutPipeline  # noqa: F401
    h, st = masked(words)
    buf = torch.randn(32, 56, 7, 128, device=dev)
    with torch.no_grad(), torch.cuda.stream(st):
        for _ in range(2):
            engine.rollout(roll, buf, 6, 50)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            engine.rollout(roll, buf, 6, 50)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 5
        lib.sf_profile_enable(0xff)
        bench.read_profile(lib)
        engine.rollout(roll, buf, 6, 50)
        torch.cuda.synchronize()
        lib.sf_profile_enable(0)
        prof = bench.read_profile(lib)
    print(f'{name:28s} {sum(bin(w).count("1") for w in words):3d} CUs: rollout {ms:6.3f} ms (eager)  ' +
          '  '.join(f'{k}:{v["avg_us"]:.1f}x{v["launches"]}' for k, v in prof.items()), flush=True)
    lib.sf_stream_destroy(h)


if len(sys.argv) > 1 and sys.argv[1] == 'ffn':      # SF_DBG=lf=32: where and when the FFN workgroups of the last launch ran
    from slotformer_amd.pipeline import encode_mask_words
    buf = torch.randn(32, 56, 7, 128, device=dev)
    for spec in ('rows2', 'rows3', 0xff):
        h, st = masked([~x & 0xffffffff for x in encode_mask_words(spec)])
        with torch.no_grad(), torch.cuda.stream(st):
            for _ in range(2):
                engine.rollout(roll, buf, 6, 3)
            torch.cuda.synchronize()
        out = (C.c_ulonglong * 256)()
        lib.sf_debug_read_wg.argtypes = [C.POINTER(C.c_ulonglong)]
        lib.sf_debug_read_wg(out)
        t0 = min(out[4 * i + 2] for i in range(32) if out[4 * i + 3])
        print(f'complement of {spec}:')
        for i in range(32):
            hw, xcc, a, b = out[4 * i:4 * i + 4]
            print(f'  blk {i:2d} xcc {xcc & 15} se {(hw >> 13) & 7} cu {(hw >> 8) & 15:2d} start {(a - t0) * 0.01:6.2f} end {(b - t0) * 0.01:6.2f}',
                  end='\n' if i % 4 == 3 else ' |')
        lib.sf_stream_destroy(h)
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == 'three':    # three-way split: rollout on SE1-3 rows 0-6, encode on SE0 and on row 7 of SE1-3
    roll_w = [0xffffff00] * 7 + [0]
    run_roll('SE1-3 rows 0-6', roll_w)
    run_roll('SE1-3 (all rows)', [0xffffff00] * 8)
    full_img, full_noise = img, noise
    for nb, words, name in ((32, [0xff] * 8, 'SE0'), (23, [0xff] * 8, 'SE0'), (24, [0xff] * 8, 'SE0'), (9, [0] * 7 + [0xffffff00], 'row 7 of SE1-3'),
                            (8, [0] * 7 + [0xffffff00], 'row 7 of SE1-3')):
        img, noise = full_img[:nb].contiguous(), full_noise[:nb].contiguous()
        run(f'{name}, {nb} videos', words)
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == 'gap':      # SF_DBG=lf=16: block 0 of the last attention and the last FFN launch, absolute ticks
    from slotformer_amd.pipeline import encode_mask_words
    buf = torch.randn(32, 56, 7, 128, device=dev)
    lib.sf_debug_read_ts.argtypes = [C.POINTER(C.c_longlong)]
    for spec in ('rows2', 'rows3', 'rows4', 0xff, 0xffff):
        h, st = masked([~x & 0xffffffff for x in encode_mask_words(spec)])
        with torch.no_grad(), torch.cuda.stream(st):
            for _ in range(2):
                engine.rollout(roll, buf, 6, 3)
            torch.cuda.synchronize()
        out = (C.c_longlong * 64)()
        lib.sf_debug_read_ts(out)
        ts = list(out)
        a0, a1, f0, f1 = ts[16], ts[25], ts[0], ts[8]
        print(f'complement of {spec}: attention block 0 runs {(a1 - a0) * 0.01:.2f} us; its end -> FFN block 0 start {(f0 - a1) * 0.01:.2f} us; '
              f'FFN block 0 runs {(f1 - f0) * 0.01:.2f} us', flush=True)
        lib.sf_stream_destroy(h)
    sys.exit(0)

if len(sys.argv) > 2 and sys.argv[1] == 'roll':     # only the rollout, on the complement of one mask (for rocprofv3)
    from slotformer_amd.pipeline import encode_mask_words
    spec = sys.argv[2] if sys.argv[2].startswith('rows') else int(sys.argv[2], 16)
    run_roll(f'complement of {spec}', [~x & 0xffffffff for x in encode_mask_words(spec)])
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == 'xcd':
    from slotformer_amd.pipeline import encode_mask_words
    for spec in ('rows2', 'rows3', 'rows4', 0xff, 0xffff):
        w = encode_mask_words(spec)
        run(f'{spec}', w)
        run_roll(f'complement of {spec}', [~x & 0xffffffff for x in w])
    sys.exit(0)

pats = [int(a, 0) for a in sys.argv[1:]] or [0xff, 0x11111111, 0x0f0f, 0x3333, 0x00ff0000, 0x01010101 | 0x02020202,
                                               0x55550000, 0xff000000, 0x000f000f | 0]
run('no mask', None)
for p in pats:
    run(f'word {p:#010x} x 8', [p] * 8)
# the same 64 CUs concentrated: two whole words, four half words
run('2 whole words', [0xffffffff, 0xffffffff, 0, 0, 0, 0, 0, 0])
run('4 half words', [0xffff, 0xffff, 0xffff, 0xffff, 0, 0, 0, 0])
run('alternating words', [0xffff, 0, 0xffff, 0, 0xffff, 0, 0xffff, 0])
