"""The opt-in two-product fp16 arithmetic of the 4-row-tile convolution (sf_set_conv_fp16x2): error of the encode against the reference
fixtures and against the default split-bf16 path, and the speed of the kernel / the encode.   python tools/conv_fp16x2_probe.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import bench  # noqa: E402
import golden_util as gu  # noqa: E402
from slotformer_amd import engine, ops, _lib  # noqa: E402
from test_engine_gpu import build, rel_err  # noqa: E402

dev = torch.device('cuda:0')
lib = _lib.lib()
with torch.no_grad():
    for name, cfg, B, T, seed, nseed in (('savi_c1', gu.C1_SAVI, 2, 3, 101, None), ('savi_c2', gu.C2_SAVI, 2, 3, 103, 7), ('savi_c5', gu.C5_SAVI, 2, 1, 105, None)):
        g = gu.load_golden(name)
        m, _ = build(cfg, g, seed, dev)
        m.testing = True
        data = {'img': gu.seeded_img(B, T, cfg['resolution'][0]).to(dev)}
        if nseed is not None:
            data['noise'] = gu.seeded_normal((B, T, cfg['slot_dict']['num_slots'], cfg['slot_dict']['slot_size']), nseed).to(dev)
        outs = {}
        for mode in (0, 1):
            lib.sf_set_conv_fp16x2(mode)
            outs[mode] = m(data)['post_slots'].clone()
        lib.sf_set_conv_fp16x2(0)
        print(f'{name}: post_slots rel err vs the reference fixture  bf16x3 {rel_err(outs[0], g["post_slots"]):.2e}   fp16x2 conv {rel_err(outs[1], g["post_slots"]):.2e}'
              f'   fp16x2 vs bf16x3 {rel_err(outs[1], outs[0].cpu().numpy()):.2e}')
    savi, roll = bench.build_models(dev, bench.bench_configs()['C2'])
    img = bench.synthetic_img(32, 6, 128).to(dev)
    noise = torch.randn(32, 6, 7, 128, device=dev)
    x = torch.randn(32, 64, 64, 64, device=dev)
    w = torch.randn(64, 64, 5, 5, device=dev) * 0.03
    b = torch.randn(64, device=dev) * 0.1
    wf = ops.pack_conv_frag(ops.pack_conv_weight(w))
    ref = torch.nn.functional.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=2)).permute(0, 2, 3, 1)
    for mode in (0, 1):
        lib.sf_set_conv_fp16x2(mode)
        y = ops.conv5x5_frag(x, wf, b)
        e = ((y.double() - ref).abs().max() / ref.abs().max()).item()
        for _ in range(3):
            ops.conv5x5_frag(x, wf, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ops.conv5x5_frag(x, wf, b)
        torch.cuda.synchronize()
        tc = (time.perf_counter() - t0) / 20
        for _ in range(3):
            engine.savi_encode(savi, img, noise=noise)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            engine.savi_encode(savi, img, noise=noise)
        torch.cuda.synchronize()
        te = (time.perf_counter() - t0) / 10
        print(f'mode {"fp16x2" if mode else "bf16x3"}: conv 32 frames {1e6 * tc:6.1f} us (max err / max |ref| {e:.2e});  encode B=32 x 6 frames whole chip {1e3 * te:.3f} ms', flush=True)
    lib.sf_set_conv_fp16x2(0)
