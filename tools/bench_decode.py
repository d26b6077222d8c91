#!/usr/bin/env python
"""StoSAVi.decode on the device: milliseconds per call and fraction of the split-bf16 MFMA roof (row N2).
    python tools/bench_decode.py [--frames 32] [--res 128] [--reps 20] [--no-masks]
Algorithmic FLOPs per slot image (savi.py:252-293): every transposed-convolution layer 2 * out_pixels * Cout * Cin * taps_per_output
(6.25 for stride 2, 25 for stride 1) + the 1x1 head; 1.135 GFLOP at 128 x 128 (7.95 GFLOP per frame of 7 slots)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402


def decode_flops_per_slot_image(m):
    ch, ks, size = list(m.dec_channels), m.dec_ks, m.dec_resolution[0]
    f = 0.0
    for i in range(len(ch) - 1):
        st = m.decoder[i][0].stride[0]
        size *= st
        f += 2.0 * size * size * ch[i + 1] * ch[i] * ks * ks / (st * st)
    return f + 2.0 * size * size * ch[-1] * 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=32)
    ap.add_argument('--res', type=int, default=128)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--slots', type=int, default=7)
    args = ap.parse_args()
    import golden_util as gu
    from slotformer_amd import engine, _lib
    from slotformer_amd.base_slots import build_model
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    cfg = gu.savi_cfg(args.res, args.slots, kernel_mlp=False, pred='mlp', rnn=False)
    m = build_model(gu.ParamsView(cfg)).eval().to(dev)
    slots = torch.randn(args.frames, args.slots, 128, device=dev)
    out = {}
    with torch.no_grad():
        for _ in range(3):
            engine.savi_decode(m, slots)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(args.reps):
            e0.record()
            engine.savi_decode(m, slots)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        ms = ts[len(ts) // 2]
        fl = decode_flops_per_slot_image(m) * args.frames * args.slots
        out = {'frames': args.frames, 'res': args.res, 'slots': args.slots, 'ms_median': ms, 'ms_min': ts[0], 'gflop': fl / 1e9,
               'tflops': fl / ms / 1e9, 'frac_of_bf16x3_roof': fl / ms / 1e9 / (2500.0 / 3), 'frames_per_s': args.frames / ms * 1e3}
        # per-class HIP-event timings of one call (library brackets)
        lib = _lib.lib()
        import ctypes as C
        lib.sf_profile_enable(0xff)
        engine.savi_decode(m, slots)
        torch.cuda.synchronize()
        lib.sf_profile_enable(0)
        names = ['conv_nhwc_implicit_gemm', 'conv_first', 'linear_gemm', 'slot_attn_iter', 'slot_update', 'attention', 'ffn_fused', 'seam']
        for c, nme in enumerate(names):
            msv, n, w = C.c_double(), C.c_longlong(), C.c_double()
            lib.sf_profile_read(c, C.byref(msv), C.byref(n), C.byref(w))
            if n.value:
                out['class_' + nme] = {'launches': n.value, 'total_ms': msv.value}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
