"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into small text summaries."""
import csv
import glob
import os
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(R, 'gpurun_out')


def short(name):
    return name.split('(')[0].replace('void ', '')[:70]


def find(d, pat):
    fs = glob.glob(os.path.join(OUT, d, '**', pat), recursive=True)
    return fs[0] if fs else None


lines = []
f = find(f'{tag}_trace', '*kernel_stats.csv')
if f:
    lines.append(f'# rocprofv3 --kernel-trace --stats  (bench.py --steps 20 --warmup 5 --no-cpu-baseline --windows 3 --min-timed-s 0: the command the driver runs -- the default hipGraph + CU-partitioned pipeline, no decode stage)  [{os.path.basename(f)}]')
    lines.append(f'{"kernel":72s} {"calls":>7s} {"total_us":>12s} {"avg_us":>10s} {"pct":>6s}')
    for r in csv.DictReader(open(f)):
        lines.append(f'{short(r["Name"]):72s} {r["Calls"]:>7s} {float(r["TotalDurationNs"]) / 1e3:12.1f} '
                     f'{float(r["AverageNs"]) / 1e3:10.2f} {float(r["Percentage"]):6.2f}')
# the roofline kernels per HIP queue: the encode stream of the pipelined (timed) region is its own CU-masked queue, the
# untimed split / isolated passes of bench.py run on the default stream with all 256 CUs
f = find(f'{tag}_trace', '*kernel_trace.csv')
if f:
    per = defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = short(r['Kernel_Name'])
        if n.startswith(('deconv5x5s2_kernel', 'decode_combine_kernel', 'decode_seg_kernel', 'ffn_qkv_tile_kernel', 'pixel_feat_stream_kernel', 'conv5x5_rows4_kernel', 'conv5x5_ws_kernel', 'layer_tok_kernel', 'qkv_rows_kernel', 'attn_core_kernel', 'ffn_tile_kernel', 'conv5x5_halo_kernel', 'sa_attn_mfma_kernel', 'sa_attn_tile_kernel', 'sa_attn_planes_kernel', 'pixel_feat_tok_kernel', 'pixel_mlp_kv_kernel', 'ffn_partial_kernel', 'ffn64_parts_kernel', 'ffn_wide_parts_kernel', 'attn_oproj_kernel', 'sa_attn_fold_kernel', 'sa_slot_update_kernel', 'conv5x5_halo256_kernel')):
            per[(n, r['Queue_Id'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    lines.append('# per-queue durations of the encode kernels  [trace_kernel_trace.csv]  (queue with the most launches of a '
                 'kernel = the CU-masked encode stream of the timed region = bench.py roofline.avg_launch_us; the others = '
                 'full-chip passes = *_isolated)')
    for (n, q), v in sorted(per.items()):
        v.sort()
        lines.append(f'{n:40s} queue {q:>3s}  calls {len(v):5d}  avg_us {sum(v) / len(v) / 1e3:9.2f}  median_us {v[len(v) // 2] / 1e3:9.2f}')
for cname, d in (('FETCH_SIZE', f'{tag}_pmc_fetch'), ('WRITE_SIZE', f'{tag}_pmc_write'), ('MFMA', f'{tag}_pmc_mfma')):
    f = find(d, '*counter_collection.csv')
    if not f:
        lines.append(f'# {cname}: no counter csv found')
        continue
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        a = agg[short(r['Kernel_Name'])][r['Counter_Name']]
        a[0] += float(r['Counter_Value'])
        a[1] += 1
    lines.append(f'# PMC pass {cname}: per-dispatch average  [{os.path.basename(f)}]')
    for k, cs in sorted(agg.items(), key=lambda kv: -sum(v[0] for v in kv[1].values()))[:14]:
        lines.append(f'{k:72s} ' + '  '.join(f'{c}={v[0] / v[1]:.4g} (n={v[1]})' for c, v in cs.items()))
# ---- PMC traffic json for bench.py's roofline.traffic ------------------------------------------------
import json
import re


def counter_avg(d, counter, pred):
    f = find(d, '*counter_collection.csv')
    if not f:
        return None
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter and pred(r['Kernel_Name']):
            tot += float(r['Counter_Value'])
            n += 1
    return tot / n if n else None


def stats_avg_us(pred):
    """Average duration (us) of the kernels matching `pred` in the kernel trace of this round, on the HIP queue where they run
    FASTEST with at least 8 launches: for the encode kernels that is the whole-chip (isolated) pass of bench.py, not the
    64-CU partition of the timed region; the rollout kernels run on every queue alike."""
    f = find(f'{tag}_trace', '*kernel_trace.csv')
    if not f:
        return None
    per = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pred(r['Kernel_Name']):
            per[r['Queue_Id']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    avgs = [sum(v) / len(v) / 1e3 for v in per.values() if len(v) >= 8]
    return min(avgs) if avgs else None


def stats_encode_queue_us(pred):
    """Mean duration (us) on the HIP queue where the matching kernels run SLOWEST (>= 8 launches): the CU-masked encode stream of the
    timed region (the kernel on its 128-CU partition beside the rollouts; the whole-chip fill streams and bench.py's isolated passes are faster)."""
    f = find(f'{tag}_trace', '*kernel_trace.csv')
    if not f:
        return None
    per = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pred(r['Kernel_Name']):
            per[r['Queue_Id']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    avgs = [sum(v) / len(v) / 1e3 for v in per.values() if len(v) >= 8]
    return max(avgs) if avgs else None


_HAS_WS = None


def is_conv(name):  # conv5x5_halo_kernel, or sf_gemm_kernel<..., ALOAD=1 (NHWC im2col), LN, BF3>
    global _HAS_WS
    if _HAS_WS is None:
        f0 = find(f'{tag}_trace', '*kernel_stats.csv')
        _HAS_WS = bool(f0) and any('conv5x5_ws_kernel' in r['Name'] for r in csv.DictReader(open(f0)))
    if 'conv5x5_rows4_kernel<false, true>' in name:   # the decoder's head-fused stride-1 layer (64 x 64 decode leg): its own class
        return False
    if _HAS_WS:
        # the pipeline's convolution launches (all time steps of a batch per launch) run the weights-stationary kernel; the 4-row-tile launches of the
        # same trace are the step-by-step passes of bench.py's untimed two-branch encode (a sixth of the frames per launch): not averaged with them
        return 'conv5x5_ws_kernel' in name
    if 'conv5x5_halo' in name or 'conv5x5_rows4' in name:
        return True
    m = re.search(r'sf_gemm_kernel<([^>]*)>', name)
    return bool(m) and m.group(1).replace(' ', '').split(',')[8] == '1'


sys.path.insert(0, R)
from slotformer_amd.build import source_tree_hash  # noqa: E402
traffic = {'source': f'profiles/{tag}_profile_summary.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)',
           # sha256 over slotformer_amd/csrc + include/ at the time of the trace: bench.py uses these numbers only while it matches
           'source_tree': source_tree_hash(),
           'correction': 'bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reports half of wide coalesced reads)'}
# MFMA-busy fraction: SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs that ran the kernel; GRBM_GUI_ACTIVE is
# the launch duration in cycles -> busy / (active * 1024 SIMDs) = share of the chip's matrix-pipe time that was used
def stats_class_avg_us(pred):
    """Call-weighted mean duration (us) of ALL launches of the kernels matching `pred` in the kernel trace (sum of total time /
    sum of calls over the matching rows of *_kernel_stats.csv): the graph-replay launches of the timed schedule dominate it."""
    f = find(f'{tag}_trace', '*kernel_stats.csv')
    if not f:
        return None, None
    tot, calls, names = 0.0, 0, []
    for r in csv.DictReader(open(f)):
        if pred(r['Name']):
            tot += float(r['TotalDurationNs'])
            calls += int(r['Calls'])
            names.append(short(r['Name']))
    return (tot / calls / 1e3 if calls else None), names


# rollout kernel classes = the kernels of the THROUGHPUT-form rollout units the timed region replays (all-heads attention
# workgroups, wide FFN workgroups on finished rows + the last layer's 32-row FFN with the step boundary); the latency-form kernels
# of the drain / tail units (attn_oproj_kernel, ffn_wide_parts_kernel<1, 4>, ffn_partial_kernel<4>) are not part of the class
# Row-tile forms (attn_rows.hip / ffn_tile.hip): an attention block is TWO launches (qkv_rows_kernel + attn_core_kernel); the class figures are
# per BLOCK: summed over both kernels and divided by the launches of attn_core_kernel (one per block)
def _has(pat):
    f = find(f'{tag}_trace', '*kernel_stats.csv')
    return bool(f) and any(pat in r['Name'] for r in csv.DictReader(open(f)))


ROWS_FORMS = _has('attn_core_kernel')
is_attn = (lambda n: 'qkv_rows_kernel' in n or 'attn_core_kernel' in n) if ROWS_FORMS else (lambda n: 'attn_all_kernel' in n)  # noqa: E731
is_ffn = (lambda n: 'ffn_tile_kernel' in n or 'ffn_qkv_tile_kernel' in n or 'ffn_partial_kernel<1>' in n) if (_has('ffn_tile_kernel') or _has('ffn_qkv_tile_kernel')) else (lambda n: 'ffn_wide_parts_kernel<2, 1>' in n or 'ffn_partial_kernel<1>' in n)  # noqa: E731


def per_block(key, total_per_launch_avg, pred):
    """attention in the row-tile forms: per-launch averages over two kernels -> per attention block (x launches / blocks)"""
    if key != 'attention' or not ROWS_FORMS or total_per_launch_avg is None:
        return total_per_launch_avg
    f = find(f'{tag}_trace', '*kernel_stats.csv')
    calls = sum(int(r['Calls']) for r in csv.DictReader(open(f)) if pred(r['Name']))
    blocks = sum(int(r['Calls']) for r in csv.DictReader(open(f)) if 'attn_core_kernel' in r['Name'])
    return total_per_launch_avg * calls / blocks if blocks else total_per_launch_avg
for key, pred, whole_chip in (('conv_nhwc_implicit_gemm', is_conv, True), ('slot_attn_iter', lambda n: 'sa_attn_mfma' in n or 'sa_attn_fold' in n or 'sa_attn_tile' in n or 'sa_attn_planes' in n, True),
                              ('layer_tok', lambda n: 'layer_tok_kernel' in n, False), ('ffn_fused', is_ffn, False), ('attention', is_attn, False), ('pixel_mlp', lambda n: 'pixel_mlp_kv_kernel' in n or 'pixel_feat_stream_kernel' in n or 'pixel_feat_tok_kernel' in n, True),
                              # the decoder's last layer with the 1x1 head in its epilogue (bench.py --decode): every launch is a whole-chip launch of the decode stream
                              ('deconv_head', lambda n: 'deconv5x5s2_kernel<64, true>' in n, False), ('deconv_head_64x64', lambda n: 'conv5x5_rows4_kernel<false, true>' in n, False)):
    fe, wr = counter_avg(f'{tag}_pmc_fetch', 'FETCH_SIZE', pred), counter_avg(f'{tag}_pmc_write', 'WRITE_SIZE', pred)
    fe, wr = per_block(key, fe, pred), per_block(key, wr, pred)
    ent = {}
    if fe is not None and wr is not None:
        ent.update({'FETCH_SIZE_KB': fe, 'WRITE_SIZE_KB': wr, 'traffic_bytes_per_launch': (2 * fe + wr) * 1024})
    busy = per_block(key, counter_avg(f'{tag}_pmc_mfma', 'SQ_VALU_MFMA_BUSY_CYCLES', pred), pred)
    if whole_chip:
        dur, names = stats_avg_us(pred), None
        rule = 'mean duration on the HIP queue where the kernel runs fastest (>= 8 launches): the whole-chip passes of bench.py'
    else:
        dur, names = stats_class_avg_us(pred)
        dur = per_block(key, dur, pred)
        rule = 'sum(TotalDurationNs) / sum(Calls) over the matching rows of the kernel_stats csv (all queues)' + (
            '; x launches / attention blocks (two launches per block: qkv_rows_kernel + attn_core_kernel)' if (key == 'attention' and ROWS_FORMS) else '')
    if busy is not None and dur:
        # SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of the 1024 SIMDs: / 1024 = matrix-pipe time per SIMD, at the 2.4 GHz
        # peak clock (a lower bound of the time: the chip clocks lower under load) over the launch duration of the trace pass
        ent.update({'SQ_VALU_MFMA_BUSY_CYCLES': busy, 'avg_launch_us_trace': dur, 'avg_launch_us_rule': rule,
                    'mfma_busy_us_per_simd': busy / 1024.0 / 2400.0, 'mfma_busy_frac': busy / 1024.0 / 2400.0 / dur})
        if names:
            ent['kernels'] = names
    if whole_chip and ent:
        eq = stats_encode_queue_us(pred)
        if eq:
            ent['avg_launch_us_trace_encode_queue'] = eq
    if ent:
        traffic[key] = ent
json.dump(traffic, open(os.path.join(OUT, f'{tag}_pmc_traffic.json'), 'w'), indent=1)
txt = '\n'.join(lines)
open(os.path.join(OUT, f'{tag}_profile_summary.txt'), 'w').write(txt + '\n')
print(txt)
