"""Throughput of the other BASELINE.json configs at full size (not the bench metric; sanity / DESIGN table).

    python tools/bench_configs.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import golden_util as gu  # noqa: E402
from slotformer_amd.base_slots import build_model  # noqa: E402
from slotformer_amd.video_prediction.models import SlotRollouter, SingleStepSlotRollouter  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


@torch.no_grad()
def run(name, scfg, rcfg, B, T, H, single=False, steve=False):
    torch.manual_seed(0)
    full = dict(scfg)
    if steve:
        full.update(dvae_dict=dict(down_factor=4, vocab_size=64, dvae_ckp_path=''),
                    dec_dict=dict(dec_type='slate', dec_num_layers=1, dec_num_heads=4, dec_d_model=64),
                    loss_dict=dict(use_img_recon_loss=False))
    savi = build_model(gu.ParamsView(full)).eval().to(dev)
    savi.testing = True
    rd = rcfg['rollout_dict']
    roll = (SingleStepSlotRollouter if single else SlotRollouter)(**rd).eval().to(dev)
    res = scfg['resolution'][0]
    img = torch.rand(B, T, 3, res, res, device=dev) * 2 - 1
    key = 'slots' if steve else 'post_slots'
    slots = savi({'img': img})[key]
    t_enc = timeit(lambda: savi({'img': img}))
    hist = rd['history_len']
    t_roll = timeit(lambda: roll(slots[:, -hist:].contiguous(), H))
    fps = B * (T + H) / (t_enc + t_roll)
    x_in = slots[:, -hist:].contiguous()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y_g = roll(x_in, H)
    t_graph = timeit(g.replay)
    same = bool(torch.equal(y_g, roll(x_in, H)))
    print(f'{name:34s} B={B:3d} {res}x{res} T={T} H={H}: encode {1e3 * t_enc:7.2f} ms  rollout {1e3 * t_roll:7.2f} ms (eager)  '
          f'-> {fps:9.0f} frames/s   [{1e6 * t_roll / H:.0f} us/step]   hipGraph replay of the same rollout {1e3 * t_graph:7.2f} ms '
          f'[{1e6 * t_graph / H:.0f} us/step, same bits {same}]')


if __name__ == '__main__':
    only = sys.argv[1] if len(sys.argv) > 1 else ''
    _run = run

    def run(name, *a, **k):   # noqa: F811  (optional filter: python tools/bench_configs.py "C4 ref")
        if only in name:
            _run(name, *a, **k)
    run('C1 OBJ3D SAVi 64^2, 6 slots', gu.C1_SAVI, gu.C1_ROLL, 4, 6, 10)
    run('C1 (batch 32)', gu.C1_SAVI, gu.C1_ROLL, 32, 6, 10)
    run('C2 CLEVRER StoSAVi 128^2', gu.C2_SAVI, gu.C2_ROLL, 32, 6, 50)
    run('C4 Physion STEVE 128^2, 6+40', gu.C4_STEVE, gu.C4_ROLL, 16, 6, 40, steve=True)
    run('C4 reference window (15 burn-in)', gu.C4_STEVE, gu.C4_ROLL_REF, 16, 15, 10, steve=True)
    run('C5 PHYRE SAVi 128^2, 1+80', gu.C5_SAVI, gu.C5_ROLL, 64, 1, 80, single=True)
