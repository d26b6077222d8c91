"""Host-to-host timing of harness.extract_and_rollout (pinned in, pinned out), call by call.   python tools/pcie_probe.py [batches]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from slotformer_amd import harness

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda:0')
cfg = bench.bench_configs()['C2']
savi, roll = bench.build_models(dev, cfg)[:2]
B, T, H = cfg[3], cfg[4], cfg[5]
vids = (torch.rand(n * B, T, 3, 128, 128) * 2 - 1).pin_memory()
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = harness.extract_and_rollout(savi, roll, vids, H, batch_size=B, to_host=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'call {it}: {1e3 * dt:8.1f} ms  {n * B * (T + H) / dt / 1e3:7.1f} k frames/s  pinned={out.is_pinned()}', flush=True)
    if it == 1:
        del out
vd = vids.to(dev)
for it in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = harness.extract_and_rollout(savi, roll, vd, H, batch_size=B)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'device-resident call {it}: {1e3 * dt:8.1f} ms  {n * B * (T + H) / dt / 1e3:7.1f} k frames/s', flush=True)
if os.environ.get('SF_PIPE_TRACE') == '1':
    pipe = list(harness._PIPES.values())[0][2]
    out = harness.extract_and_rollout(savi, roll, vids, H, batch_size=B, to_host=True)
    tl = pipe.timeline
    print('host in/out  encode_end', [round(v, 1) for v in tl['encode_end_ms']])
    print('             rollout_start', [round(v, 1) for v in tl['rollout_start_ms']], 'end', [round(v, 1) for v in tl['rollout_end_ms']])
    out = harness.extract_and_rollout(savi, roll, vd, H, batch_size=B)
    tl = pipe.timeline
    print('device       encode_end', [round(v, 1) for v in tl['encode_end_ms']])
    print('             rollout_start', [round(v, 1) for v in tl['rollout_start_ms']], 'end', [round(v, 1) for v in tl['rollout_end_ms']])
for name, vin, th in (('host in, device out', vids, False), ('device in, host out', vd, True)):
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = harness.extract_and_rollout(savi, roll, vin, H, batch_size=B, to_host=th)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f'{name}: {1e3 * dt:8.1f} ms  {n * B * (T + H) / dt / 1e3:7.1f} k frames/s', flush=True)
