"""What slows a token-stationary rollout unit inside the pipeline?  Unit graphs (192 videos) loop on the rollout streams while ONE kind of work loops on
the encode lane:  python tools/unit_interference_probe.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from slotformer_amd import engine, ops, _lib  # noqa: E402
from slotformer_amd.pipeline import EncodeRolloutPipeline  # noqa: E402

dev = torch.device('cuda:0')
savi, roll = bench.build_models(dev, bench.bench_configs()['C2'])
lib = _lib.lib()
B, T, H = 32, 6, 50
with torch.no_grad():
    pipe = EncodeRolloutPipeline(savi, roll, B, T, H)
    print('split', pipe.split, 'tok', pipe.tok, 'G', pipe.G)
    img = bench.synthetic_img(B, T, 128).to(dev)
    rolls, lane = pipe.roll_streams, pipe.lanes[0][0]
    planes = torch.zeros(engine.savi_planes_bytes(savi, B, T), dtype=torch.uint8, device=dev)
    post = torch.zeros(B, T, 7, 128, device=dev)
    x64 = torch.randn(192, 64, 64, 64, device=dev)
    convs = [m for m in savi.encoder.modules() if isinstance(m, torch.nn.Conv2d)]
    wf = ops.pack_conv_frag(ops.pack_conv_weight(convs[1].weight.detach().float()))
    cb = convs[1].bias.detach().float()
    big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    big2 = torch.empty(1 << 30, dtype=torch.uint8, device=dev)

    def k_none():
        pass

    def k_feat():
        engine.savi_features(savi, img, planes, ws_slot='uip')

    def k_conv():
        for _ in range(3):
            ops.conv5x5_ws(x64, wf, cb)

    def k_copy():
        big2.copy_(big)

    def k_chain():
        engine.savi_slots_chain(savi, planes, 1, B, T, post, ws_slot='uip')

    engine.savi_features(savi, img, planes, ws_slot='uip')
    torch.cuda.synchronize()
    for name, fn, reps in (('nothing', k_none, 1), ('features of a batch (lane)', k_feat, 14), ('3 x conv_ws 192 frames (lane)', k_conv, 16),
                           ('1 GiB device copy (lane)', k_copy, 60), ('slot chain of a batch (lane)', k_chain, 30)):
        for nunits in (1, 2):
            torch.cuda.synchronize()
            res = []
            for rep in range(2):
                evs = []
                with torch.cuda.stream(lane):
                    for _ in range(reps):
                        fn()
                for ri in range(nunits):
                    st = rolls[ri]
                    with torch.cuda.stream(st):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(st)
                        pipe.units[ri].graph.replay()
                        e1.record(st)
                        evs.append((e0, e1))
                le0 = torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                res = [a.elapsed_time(b) for a, b in evs]
            msg = ''
            if 'lt' in os.environ.get('SF_DBG', ''):
                # (-DLT_STAMPS build through SF_LIB_PATH: shader cycles of workgroup 0's wave 0 over the first layer of the LAST launch)
                import ctypes as C
                ts = (C.c_longlong * 16)()
                lib.sf_debug_read_ts_layer_tok(ts)
                msg = (f'   layer 0 of the last launch: attention DMA waits {ts[8]} barrier waits {ts[9]} work {ts[10]} | FFN DMA waits {ts[11] - ts[8]} '
                       f'barrier waits {ts[12] - ts[9]} work {ts[13] - ts[10]} | wall {(ts[6] - ts[0]) / 100:.1f} us')
            print(f'{name:34s} units in flight {nunits}: ' + ' '.join(f'{t:6.2f} ms' for t in res) + msg)
