"""Device time of the training-side kernels outside the rollout (row N1): Slot-Attention iteration backward at the CLEVRER
encode shape (32 videos x 6 frames, 4096 pixels, 7 slots, D = 128) against its HBM roofline (read K, V; write dK, dV), and
the whole SlotAttention module forward + backward next to the same step in torch eager (the reference's formulation,
savi.py:56-102, on the ROCm PyTorch build of the same GPU)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from slotformer_amd import ops  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def eager_slot_attention(sa, inputs, slots):
    """savi.py:56-102 with torch ops on the module's own parameters."""
    import torch.nn.functional as F
    x = sa.norm_inputs(inputs)
    k, v = sa.project_k(x), sa.project_v(x)
    B, N, D = slots.shape
    for _ in range(sa.num_iterations):
        prev = slots
        q = sa.project_q(slots)
        attn = F.softmax(sa.attn_scale * torch.einsum('bnc,bmc->bnm', k, q), dim=-1) + sa.eps
        attn = attn / attn.sum(dim=1, keepdim=True)
        upd = torch.einsum('bnm,bnc->bmc', attn, v)
        slots = sa.gru(upd.reshape(B * N, D), prev.reshape(B * N, D)).view(B, N, D)
        slots = slots + sa.mlp(slots)
    return slots


def module_bench(F_, HW=4096, N=7, D=128, H=256, iters=2):
    from slotformer_amd.base_slots.models.savi import SlotAttention
    torch.manual_seed(0)
    sa = SlotAttention(D, iters, N, D, H).to(dev)
    x = torch.randn(F_, HW, D, device=dev, requires_grad=True)
    s0 = torch.randn(F_, N, D, device=dev, requires_grad=True)
    dout = torch.randn(F_, N, D, device=dev)

    def ours():
        sa(x, s0).backward(dout)

    def eager():
        eager_slot_attention(sa, x, s0).backward(dout)

    t_ours, t_eager = timeit(ours, 10), timeit(eager, 10)
    return {'frames': F_, 'iters': iters, 'hip_ms': round(t_ours, 3), 'torch_eager_ms': round(t_eager, 3),
            'speedup': round(t_eager / t_ours, 2)}


def decode_bench(res, frames):
    """SAVi spatial-broadcast decoder (row N2) inference time: 7 slots, D = 128, 8x8 grid -> res x res."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import golden_util as gu
    from slotformer_amd.base_slots import build_model
    torch.manual_seed(0)
    m = build_model(gu.ParamsView(gu.savi_cfg(res, 7, kernel_mlp=False, pred='mlp', rnn=False))).to(dev).eval()
    slots = torch.randn(frames, 7, 128, device=dev)
    with torch.no_grad():
        ms = timeit(lambda: m.decode(slots), 5)
    # dense-equivalent flops: every transposed conv as its useful taps only (Hout*Wout*Cout*Cin*ks*ks / stride^2)
    ch, size, fl = list(m.dec_channels), m.dec_resolution[0], 0
    for i in range(len(ch) - 1):
        st = m.decoder[i][0].stride[0]
        size *= st
        fl += 2 * size * size * ch[i] * ch[i + 1] * 25 / (st * st)
    fl = (fl + 2 * size * size * ch[-1] * 4) * 7 * frames
    return {'res': res, 'frames': frames, 'ms': round(ms, 3), 'useful_TFLOPs': round(fl / ms / 1e9, 1),
            'deconv_by_parity_class': os.environ.get('SF_DECONV_CLASSES', '1') != '0'}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'decode':
        print(json.dumps({'savi_decode': [decode_bench(128, 32), decode_bench(64, 320)]}))
        return
    B, HW, N, D = 192, 4096, 7, 128
    g = torch.Generator(device='cpu').manual_seed(0)
    k, v = (torch.randn(B, HW, D, generator=g).to(dev) for _ in range(2))
    q, du = (torch.randn(B, N, D, generator=g).to(dev) for _ in range(2))
    pn, pd, _ = ops.slot_attn_iter(k, v, q)
    dk, dv = torch.empty_like(k), torch.empty_like(v)
    fwd = timeit(lambda: ops.slot_attn_iter(k, v, q))
    bwd = timeit(lambda: ops.slot_attn_iter_bwd(k, v, q, pn, pd, du))
    acc = timeit(lambda: ops.slot_attn_iter_bwd(k, v, q, pn, pd, du, dk=dk, dv=dv))
    kv = 2 * B * HW * D * 4
    print(json.dumps({'shape': dict(frames=B, HW=HW, N=N, D=D),
                      'sa_iter_fwd': {'ms': round(fwd, 4), 'GBps': round(kv / fwd / 1e6, 1), 'bytes': kv},
                      'sa_iter_bwd': {'ms': round(bwd, 4), 'GBps': round(2 * kv / bwd / 1e6, 1), 'bytes': 2 * kv},
                      'sa_iter_bwd_accumulate': {'ms': round(acc, 4), 'GBps': round(3 * kv / acc / 1e6, 1), 'bytes': 3 * kv},
                      'hbm_peak_GBps': 8000,
                      'slot_attention_module_fwd_bwd': [module_bench(32), module_bench(192)]}))


if __name__ == '__main__':
    main()
