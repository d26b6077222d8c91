"""Token-stationary FFN block (csrc/ffn_tok.hip) against the row-tile form (csrc/ffn_tile.hip: ffn_tile_kernel) and a plain PyTorch fp32 reference:
max relative error of both, and us per launch of each at M rows.     python tools/ffn_tok_probe.py [M ...]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import golden_util as gu  # noqa: E402
from slotformer_amd import _lib, engine  # noqa: E402
from slotformer_amd.video_prediction.models import SlotRollouter  # noqa: E402

dev = torch.device('cuda:0')
lib = _lib.lib()
torch.manual_seed(3)
r = SlotRollouter(**dict(gu.C2_ROLL['rollout_dict'])).eval().to(dev)
plan = engine.rollouter_plan(r)
w = plan.struct.layers[1]
layer = r.transformer_encoder.layers[1]
st = torch.cuda.current_stream().cuda_stream
packed = torch.empty(lib.sf_ffn_tok_packed_bytes(), dtype=torch.uint8, device=dev)
_lib.check(lib.sf_pack_ffn_tok_weights(layer.linear1.weight.data_ptr(), layer.linear2.weight.data_ptr(), packed.data_ptr(), 256, 1024, st))


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for M in [int(a) for a in sys.argv[1:]] or [5376, 2304, 10752, 130]:
    g = torch.Generator().manual_seed(M)
    x2 = torch.randn(M, 256, generator=g).to(dev)
    with torch.no_grad():
        ref = x2 + F.linear(F.relu(F.linear(F.layer_norm(x2, (256, ), layer.norm2.weight, layer.norm2.bias), layer.linear1.weight, layer.linear1.bias)),
                            layer.linear2.weight, layer.linear2.bias)
    yt = torch.full((M, 256), float('nan'), device=dev)
    yk = torch.full((M, 256), float('nan'), device=dev)
    tile = lambda: _lib.check(lib.sf_ffn_block_rows_f32(C.byref(w), x2.data_ptr(), yt.data_ptr(), M, 1024, st))  # noqa: E731
    tok = lambda: _lib.check(lib.sf_ffn_block_tok_f32(C.byref(w), packed.data_ptr(), x2.data_ptr(), yk.data_ptr(), M, st))  # noqa: E731
    tile()
    tok()
    torch.cuda.synchronize()
    rel = lambda a: ((a - ref).abs().max() / ref.abs().max()).item()  # noqa: E731
    print(f'M {M:6d}: rel err row tiles {rel(yt):.2e}  token-stationary {rel(yk):.2e}  max |diff| between them {(yt - yk).abs().max().item():.2e}   '
          f'us per launch: row tiles ({(M + 63) // 64} workgroups) {timeit(tile):7.2f}   token-stationary ({(M + 127) // 128} workgroups) {timeit(tok):7.2f}')
ts = (C.c_longlong * 8)()
lib.sf_debug_read_ts_ffn_tok(ts)
if any(ts):
    print('cycle counters of workgroup 0, wave 0 (s_memtime; -DTK_STAMPS build): barrier waits', ts[0], ' first product', ts[1], ' second product + conversion', ts[2], ' loop total', ts[3],
          f' in {ts[4]} ticks of the 100 MHz wall clock -> shader clock {ts[3] / (ts[4] * 10.0):.3f} GHz; prologue {ts[5] / 100:.2f} us, entry to exit {ts[6] / 100:.2f} us')
