"""Where a rollout unit's wall time goes inside the pipeline: kernel durations vs the gaps between dependent launches, per hardware queue.
   usage (GPU box):  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gap_qt -o trace -- python $R/bench.py --steps 60 ...
                     python tools/unit_gap_probe.py gpurun_out/gap_qt  [ms_before_end_lo ms_before_end_hi]
   Prints, for every queue, over the window [t_lo, t_hi] of the trace's time span: busy time, span, per-kernel (count, mean duration, mean gap in front of it)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0   # window = [end - lo ms, end - hi ms] of the trace (the run under study is the last thing traced)
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
f = glob.glob(f'{d}/**/*kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((r['Queue_Id'], int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]))
t0 = min(r[1] for r in rows)
t1 = max(r[2] for r in rows)
# the last long busy region = the timed window: take [lo, hi] of the LAST 'pipe run' -- found as the region after the largest idle gap in the second half
ev = sorted(rows, key=lambda r: r[1])
a = t1 - lo * 1e6
b = t1 - hi * 1e6
print(f'trace span {1e-6 * (t1 - t0):.1f} ms, window {1e-6 * (a - t0):.1f} .. {1e-6 * (b - t0):.1f} ms')
byq = collections.defaultdict(list)
for r in ev:
    if r[1] >= a and r[2] <= b:
        byq[r[0]].append(r)
for q, rs in sorted(byq.items()):
    rs.sort(key=lambda r: r[1])
    busy = sum(r[2] - r[1] for r in rs)
    span = rs[-1][2] - rs[0][1]
    print(f'== queue {q}: {len(rs)} launches, busy {1e-6 * busy:.2f} ms of span {1e-6 * span:.2f} ms ({100.0 * busy / max(span, 1):.1f} %)')
    st = collections.defaultdict(lambda: [0, 0, 0, []])
    prev_end = None
    for r in rs:
        s = st[r[3]]
        s[0] += 1
        s[1] += r[2] - r[1]
        if prev_end is not None:
            g = r[1] - prev_end
            s[2] += g
            s[3].append(g)
        prev_end = max(prev_end or 0, r[2])
    for k, s in sorted(st.items(), key=lambda kv: -kv[1][1]):
        gs = sorted(s[3])
        med = gs[len(gs) // 2] if gs else 0
        p90 = gs[int(len(gs) * 0.9)] if gs else 0
        print(f'   {k:44s} n {s[0]:5d}  dur {1e-3 * s[1] / s[0]:8.1f} us  gap-before mean {1e-3 * s[2] / max(len(gs), 1):7.1f} median {1e-3 * med:7.1f} p90 {1e-3 * p90:7.1f} us   total dur {1e-6 * s[1]:7.2f} ms  total gap {1e-6 * s[2]:7.2f} ms')
