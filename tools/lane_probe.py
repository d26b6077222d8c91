"""Who slows whom in the three-way CU partition?  Times (per batch, steady state) of subsets of {lane 0, lane 1, rollout}
running side by side:  python tools/lane_probe.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from slotformer_amd.pipeline import EncodeRolloutPipeline  # noqa: E402

dev = torch.device('cuda:0')
savi, roll = bench.build_models(dev)
img = bench.synthetic_img(32).to(dev)
noise = torch.randn(32, 6, 7, 128, device=dev)
with torch.no_grad():
    pipe = EncodeRolloutPipeline(savi, roll, 32, 6, 50, partition=os.environ.get('PART', 'three'))

    def go(name, use_lanes, use_roll, n=12):
        torch.cuda.synchronize()
        for rep in range(2):
            t0 = time.perf_counter()
            for _ in range(n):
                for li in use_lanes:
                    st, lo, hi = pipe.lanes[li]
                    with torch.cuda.stream(st):
                        pipe._encode(img, noise, pipe.bufs[0], None, lo, hi, li)
                if use_roll:
                    with torch.cuda.stream(pipe.s_roll):
                        pipe.graphs[1].replay()
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f'{name:34s} {1e3 * dt / n:7.3f} ms per batch   (host issue {1e3 * t_issue / n:.3f} ms)', flush=True)

    go('lane 0 alone', [0], False)
    if len(pipe.lanes) > 1:
        go('lane 1 alone', [1], False)
        go('lanes 0 + 1', [0, 1], False)
    go('rollout alone', [], True)
    go('rollout + lane 0', [0], True)
    if len(pipe.lanes) > 1:
        go('rollout + lane 1', [1], True)
        go('rollout + lanes 0 + 1', [0, 1], True)

    if os.environ.get('PIPE'):
        ring = [bench.synthetic_img(32, seed=100 + k).to(dev) for k in range(3)]
        out = torch.empty(12, 32, 56, 7, 128, device=dev)
        for fill in (True,):
            pipe.fill_whole_chip = fill
            recs = []
            enc0, roll0 = pipe._encode, pipe._rollout

            def enc(*a, **k):
                st = torch.cuda.current_stream()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st); enc0(*a, **k); e1.record(st)
                recs.append(('E%d' % (a[6] if len(a) > 6 else 9), e0, e1))

            def rol(gi):
                st = torch.cuda.current_stream()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st); roll0(gi); e1.record(st)
                recs.append(('R', e0, e1))

            pipe._encode, pipe._rollout = enc, rol
            pipe.run([ring[j % 3] for j in range(12)], None, out=out)
            torch.cuda.synchronize()
            pipe._encode, pipe._rollout = enc0, roll0
            print('fill on the whole chip:', fill)
            for k in ('E9', 'E0', 'E1', 'R'):
                print('   ', k, ' '.join(f'{a.elapsed_time(b):.2f}' for n_, a, b in recs if n_ == k))
        if not os.environ.get('FREE'):
            sys.exit(0)

    if os.environ.get('COPY'):
        ring = [bench.synthetic_img(32, seed=100 + k).to(dev) for k in range(3)]
        n = 12
        out = torch.empty(n, 32, 56, 7, 128, device=dev)
        imgs = [ring[j % 3] for j in range(n)]

        def run_copy(prelude, epilogue, timed_wrap, host_lane_wait=False):
            NB = 2
            cur = torch.cuda.current_stream(dev)
            s_roll, lanes = pipe.s_roll, pipe.lanes
            nl = len(lanes)
            if prelude:
                for st, _, _ in lanes:
                    st.wait_stream(cur)
                s_roll.wait_stream(cur)
            ev_enc = [[torch.cuda.Event() for _ in range(nl)] for _ in range(n)]
            ev_roll = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
            recs = []
            for j in range(n):
                if host_lane_wait and j >= NB:
                    ev_roll[j - NB].synchronize()
                for li, (st, lo, hi) in enumerate(lanes):
                    with torch.cuda.stream(st):
                        if j >= NB and not host_lane_wait:
                            st.wait_event(ev_roll[j - NB])
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        if timed_wrap:
                            e0.record(st)
                        pipe._encode(imgs[j], None, pipe.bufs[j % NB], None, lo, hi, li)
                        if timed_wrap:
                            e1.record(st)
                            recs.append((li, e0, e1))
                        ev_enc[j][li].record(st)
                with torch.cuda.stream(s_roll):
                    for e in ev_enc[j]:
                        s_roll.wait_event(e)
                    pipe._rollout(j % NB)
                    out[j].copy_(pipe.bufs[j % NB])
                    ev_roll[j].record(s_roll)
            if epilogue == 'host':
                ev_roll[-1].synchronize()
                for st, _, _ in lanes:
                    cur.wait_stream(st)
                cur.wait_stream(s_roll)
            elif epilogue:
                for st, _, _ in lanes:
                    cur.wait_stream(st)
                cur.wait_stream(s_roll)
            torch.cuda.synchronize()
            iv = [ev_roll[i].elapsed_time(ev_roll[i + 1]) for i in range(n - 1)]
            print(f'prelude {prelude} epilogue {epilogue} timed_wrap {timed_wrap} host_lane_wait {host_lane_wait}: intervals', ' '.join(f'{x:.2f}' for x in iv))
            for k in range(nl):
                if recs:
                    print('    lane', k, ' '.join(f'{a.elapsed_time(b):.2f}' for n_, a, b in recs if n_ == k))

        run_copy(True, 'host', False)
        run_copy(True, 'host', False, host_lane_wait=True)
        run_copy(True, 'host', True, host_lane_wait=True)
        sys.exit(0)

    if os.environ.get('FREE'):
        ring = [bench.synthetic_img(32, seed=100 + k).to(dev) for k in range(3)]
        outb = torch.empty(12, 32, 56, 7, 128, device=dev)
        for variant in ('lanes wait for rollout j-2 + out copy',):
            evs = {0: [], 1: [], 'R': []}
            ev_l = {}
            ev_r = {}
            for it in range(12):
                alt = variant != 'same input' and variant != 'ring inputs + fresh noise'
                gi = it % 2 if alt else 1
                im = img if variant == 'same input' else ring[it % 3]
                nz = noise if variant == 'same input' else None
                for li in (0, 1):
                    st, lo, hi = pipe.lanes[li]
                    with torch.cuda.stream(st):
                        if it >= 2:
                            st.wait_event(ev_r[it - 2])
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(st); pipe._encode(im, nz, pipe.bufs[gi], None, lo, hi, li); e1.record(st)
                        evs[li].append((e0, e1))
                        ev_l[(it, li)] = e1
                with torch.cuda.stream(pipe.s_roll):
                    if True:
                        pipe.s_roll.wait_event(ev_l[(it, 0)]); pipe.s_roll.wait_event(ev_l[(it, 1)])
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(pipe.s_roll); pipe.graphs[gi].replay(); e1.record(pipe.s_roll)
                    evs['R'].append((e0, e1))
                    ev_r[it] = e1
                    if 'out copy' in variant:
                        outb[it].copy_(pipe.bufs[gi])
                        ev_r[it] = torch.cuda.Event(enable_timing='untimed' not in variant)
                        ev_r[it].record(pipe.s_roll)
            torch.cuda.synchronize()
            print(variant)
            for k, v in evs.items():
                print('   ', k, ' '.join(f'{a.elapsed_time(b):.2f}' for a, b in v))
        sys.exit(0)

    # the real schedule (events between the streams)
    from slotformer_amd import _lib
    lib = _lib.lib()
    ring = [bench.synthetic_img(32, seed=100 + k).to(dev) for k in range(3)]
    out = torch.empty(20, 32, 56, 7, 128, device=dev)
    for name, nzs, prof in (('pipe.run, fixed noise', [noise] * 20, 0), ('pipe.run, fresh noise', None, 0), ('pipe.run, fresh noise, conv+SA brackets', None, 9)):
        pipe.run([ring[j % 3] for j in range(20)], nzs, out=out)
        torch.cuda.synchronize()
        lib.sf_profile_enable(prof)
        t0 = time.perf_counter()
        pipe.run([ring[j % 3] for j in range(20)], nzs, out=out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        lib.sf_profile_enable(0)
        bench.read_profile(lib)
        ev = pipe.completion_events
        iv = [ev[i].elapsed_time(ev[i + 1]) for i in range(len(ev) - 1)]
        print(f'{name:44s} {1e3 * dt / 20:7.3f} ms per batch; completion intervals {" ".join(f"{x:.2f}" for x in iv)}', flush=True)

    # per-stage durations inside the real schedule
    recs = []
    enc0, roll0 = pipe._encode, pipe._rollout

    def enc(*a, **k):
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); enc0(*a, **k); e1.record(st)
        recs.append(('E%d' % (a[6] if len(a) > 6 else k.get('lane', 0)), e0, e1))

    def rol(gi):
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); roll0(gi); e1.record(st)
        recs.append(('R', e0, e1))

    pipe._encode, pipe._rollout = enc, rol
    pipe.run([ring[j % 3] for j in range(12)], None, out=out[:12])
    torch.cuda.synchronize()
    base = recs[0][1]
    for name, e0, e1 in recs:
        print(f'  {name}: start {base.elapsed_time(e0):8.2f}  end {base.elapsed_time(e1):8.2f}  ({e0.elapsed_time(e1):.2f} ms)')
