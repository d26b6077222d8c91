"""sf_slate_attention_bwd_f32 at the STEVE Physion decoder's shapes (72 sequences of 1025 tokens, 4 heads of 48; cross-attention to 6 slots): us per call,
fraction of the split-bf16 MFMA roof on ALGORITHMIC flops (causal: half the score matrix), error against torch autograd at a small shape.
    python tools/slate_bwd_probe.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from slotformer_amd import ops  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (B, Lq, Lk, H, hd, causal) in [(72, 1025, 1025, 4, 48, True), (72, 1025, 6, 4, 48, False), (96, 1025, 1025, 4, 48, True)]:
    d = H * hd
    g = torch.Generator(device='cpu').manual_seed(1)
    q, k, v, go = (torch.randn(B, L, d, generator=g).to(dev) for L in (Lq, Lk, Lk, Lq))
    out = ops.slate_attention(q, k, v, H, causal)
    t = timeit(lambda: ops.slate_attention_bwd(q, k, v, out, go, H, causal))
    flops = 5 * 2.0 * B * H * Lq * Lk * hd * (0.5 if causal else 1.0)
    print(f'B {B} Lq {Lq} Lk {Lk} H {H} hd {hd} causal {causal}: {t:8.1f} us per call   {flops / t / 1e6:7.1f} TFLOP/s algorithmic = {flops / t / 1e6 / 833.3:.3f} of the split-bf16 roof', flush=True)
# error at a small shape against autograd
B, L, H, hd = 2, 200, 4, 48
d = H * hd
g = torch.Generator(device='cpu').manual_seed(2)
q, k, v, go = (torch.randn(B, L, d, generator=g) for _ in range(4))
qo, ko, vo = (t.clone().double().requires_grad_(True) for t in (q, k, v))
qh, kh, vh = (t.view(B, -1, H, hd).transpose(1, 2) for t in (qo, ko, vo))
s = (qh @ kh.transpose(-1, -2)) * hd ** -0.5
s = s.masked_fill(torch.triu(torch.ones(L, L, dtype=torch.bool), 1), float('-inf'))
ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, L, d)
ref.backward(go.double())
qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
out = ops.slate_attention(qd, kd, vd, H, True)
dq, dk, dv = ops.slate_attention_bwd(qd, kd, vd, out, go.to(dev), H, True)
err = lambda a, b: ((a.cpu().double() - b).abs().max() / b.abs().max()).item()  # noqa: E731
print(f'rel err vs float64 autograd: dq {err(dq, qo.grad):.2e} dk {err(dk, ko.grad):.2e} dv {err(dv, vo.grad):.2e}')
if 'sabts' in os.environ.get('SF_DBG', ''):
    # -DSAB_STAMPS build (SF_LIB_PATH): shader cycles of wave 0 of workgroup (0, 0, 0) per phase, summed over its query blocks
    import ctypes as C
    from slotformer_amd import _lib
    B, Lq, H, hd = 72, 1025, 4, 48
    d = H * hd
    q, k, v, go = (torch.randn(B, Lq, d, device=dev) for _ in range(4))
    out = ops.slate_attention(q, k, v, H, True)
    ops.slate_attention_bwd(q, k, v, out, go, H, True)
    torch.cuda.synchronize()
    ts = (C.c_longlong * 16)()
    lib = _lib.lib()
    lib.sf_debug_read_ts_sab.argtypes = [C.POINTER(C.c_longlong)]
    lib.sf_debug_read_ts_sab(ts)
    n = max(ts[8], 1)
    names = ['prologue', 'wait at loop top (barrier)', 'put Q / dO + barrier', 'S, dP products', 'exp / dS / split / writes', 'barrier', 'dV, dK, dQ products', 'atomics']
    print(f'key block 0, {n} query blocks, cycles per query block: ' + ', '.join(f'{nm} {ts[i] / (n if i else 1):.0f}' for i, nm in enumerate(names)))
# forward (sf_slate_attention_f32: the flash kernel) at the decoder's shape
for (B, L, H, hd) in [(72, 1025, 4, 48), (12, 1025, 4, 64)]:
    d = H * hd
    g = torch.Generator(device='cpu').manual_seed(3)
    q, k, v = (torch.randn(B, L, d, generator=g).to(dev) for _ in range(3))
    t = timeit(lambda: ops.slate_attention(q, k, v, H, True))
    flops = 2 * 2.0 * B * H * L * L * hd * 0.5
    o = ops.slate_attention(q[:2], k[:2], v[:2], H, True)
    qh, kh, vh = (x[:2].double().cpu().view(2, L, H, hd).transpose(1, 2) for x in (q, k, v))
    sc = (qh @ kh.transpose(-1, -2)) * hd ** -0.5
    sc = sc.masked_fill(torch.triu(torch.ones(L, L, dtype=torch.bool), 1), float('-inf'))
    ref = (torch.softmax(sc, -1) @ vh).transpose(1, 2).reshape(2, L, d)
    print(f'forward B {B} L {L} H {H} hd {hd}: {t:8.1f} us per call  {flops / t / 1e6:7.1f} TFLOP/s algorithmic = {flops / t / 1e6 / 833.3:.3f} of the split-bf16 roof;  rel err vs float64 {err(o, ref):.2e}')
