"""Token-stationary whole-layer launch (csrc/layer_tok.hip) against a float64 PyTorch restatement of nn.TransformerEncoderLayer (norm_first, relu) and
against the row-tile forms it replaces (attention rows + core, FFN tile):  max relative error, us per launch.

    [SF_DBG=lt] python tools/layer_tok_probe.py [B L ...]        (pairs; default: 128 42, 192 42, 64 36, 256 8, 256 48, 3 42, 1 42)"""
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
with torch.no_grad():   # biases / LayerNorm parameters away from their (zero / one) initial values: a wrong index must show
    for p in r.parameters():
        if p.dim() == 1:
            p.add_(0.3 * torch.randn_like(p))
plan = engine.rollouter_plan(r)
w = plan.struct.layers[1]
layer = r.transformer_encoder.layers[1]
st = torch.cuda.current_stream().cuda_stream
NL = int(os.environ.get('LT_NL', '1'))   # consecutive layers per launch (layers 0 .. NL - 1 of the rollouter; 1: layer 1 alone)


def reference(x, layer=layer):
    """x [B, L, 256] -> the layer in float64"""
    d = lambda t: t.detach().double()  # noqa: E731
    xx = x.double()
    B, L, D = xx.shape
    h = F.layer_norm(xx, (D, ), d(layer.norm1.weight), d(layer.norm1.bias))
    qkv = F.linear(h, d(layer.self_attn.in_proj_weight), d(layer.self_attn.in_proj_bias))
    q, k, v = qkv.split(D, dim=-1)
    hd = lambda t: t.reshape(B, L, 8, 32).transpose(1, 2)  # noqa: E731
    a = torch.softmax(hd(q) @ hd(k).transpose(-1, -2) / 32 ** 0.5, dim=-1) @ hd(v)
    x2 = xx + F.linear(a.transpose(1, 2).reshape(B, L, D), d(layer.self_attn.out_proj.weight), d(layer.self_attn.out_proj.bias))
    h2 = F.layer_norm(x2, (D, ), d(layer.norm2.weight), d(layer.norm2.bias))
    return x2 + F.linear(F.relu(F.linear(h2, d(layer.linear1.weight), d(layer.linear1.bias))), d(layer.linear2.weight), d(layer.linear2.bias))


def timeit(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


args = [int(a) for a in sys.argv[1:]]
shapes = list(zip(args[0::2], args[1::2])) or [(128, 42), (192, 42), (64, 36), (256, 8), (256, 48), (3, 42), (1, 42), (128, 16)]
for B, L in shapes:
    g = torch.Generator().manual_seed(B * 100 + L)
    x = torch.randn(B, L, 256, generator=g).to(dev)
    with torch.no_grad():
        if NL == 1:
            ref = reference(x)
        else:
            ref = x
            for k in range(NL):
                ref = reference(ref, r.transformer_encoder.layers[k])
    y = torch.full((B, L, 256), float('nan'), device=dev)
    wp = C.byref(w) if NL == 1 else plan.struct.layers
    tok = lambda: _lib.check(lib.sf_layer_tok_block_f32(wp, NL, x.data_ptr(), y.data_ptr(), B, L, st))  # noqa: E731
    tok()
    torch.cuda.synchronize()
    err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
    # position independence: the same video at another place of the batch
    msg = ''
    if B >= 8:
        x2 = x.clone()
        x2[5] = x[0]
        y2 = torch.empty_like(y)
        _lib.check(lib.sf_layer_tok_block_f32(wp, NL, x2.data_ptr(), y2.data_ptr(), B, L, st))
        torch.cuda.synchronize()
        msg = f'  video 0 at position 5: max |diff| {(y2[5] - y[0]).abs().max().item():.2e}, others unchanged: {bool((y2[:5] == y[:5]).all() and (y2[6:] == y[6:]).all())}'
    # the forms it replaces on the same rows
    planes = torch.empty(lib.sf_attn_rows_planes_bytes(B), dtype=torch.uint8, device=dev)
    x2b = torch.empty(B * L, 256, device=dev)
    yb = torch.empty(B * L, 256, device=dev)
    old = lambda: (_lib.check(lib.sf_attn_block_rows_f32(C.byref(w), x.data_ptr(), x2b.data_ptr(), planes.data_ptr(), B, L, L, st)),  # noqa: E731
                   _lib.check(lib.sf_ffn_block_rows_f32(C.byref(w), x2b.data_ptr(), yb.data_ptr(), B * L, 1024, st)))
    old()
    torch.cuda.synchronize()
    err_old = ((yb.view(B, L, 256).double() - ref).abs().max() / ref.abs().max()).item() if NL == 1 else float('nan')
    vpw = 128 // L
    print(f'B {B:4d} L {L:3d}: rel err token-stationary {err:.2e}  (row-tile forms {err_old:.2e}; between them {(yb.view(B, L, 256) - y).abs().max().item():.2e}){msg}\n'
          f'              us per launch: token-stationary ({(B + vpw - 1) // vpw} workgroups) {timeit(tok):7.2f}   attention rows + core + FFN tile {timeit(old):7.2f}', flush=True)
ts = (C.c_longlong * 16)()
lib.sf_debug_read_ts_layer_tok(ts)
if any(ts):
    t0 = ts[0]
    print('wall-clock stamps of workgroup 0 (us from entry): rows requested, first stages out', (ts[1] - t0) / 100, ' vectors in LDS', (ts[2] - t0) / 100, ' LN1 + b_o', (ts[3] - t0) / 100,
          ' attention block done', (ts[4] - t0) / 100, ' LN2 + b2', (ts[5] - t0) / 100, ' FFN block done (first layer)', (ts[6] - t0) / 100, ' rows stored', (ts[7] - t0) / 100)
    if any(ts[8:14]):
        print(f'shader cycles of wave 0 (-DLT_STAMPS build), attention block: DMA waits {ts[8]}  barrier waits {ts[9]}  work {ts[10]}   '
              f'FFN block: DMA waits {ts[11] - ts[8]}  barrier waits {ts[12] - ts[9]}  work {ts[13] - ts[10]}')
