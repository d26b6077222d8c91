"""Microbenchmark sf_linear_f32 / sf_conv2d_nhwc_f32 tile configurations on the GPU.

    python tools/gemm_bench.py            # rollout + encoder shapes x config ids (SF_GEMM_CFG)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from slotformer_amd import ops  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, iters=20, reps=5):
    """Device time per call: capture `iters` back-to-back calls into a hipGraph, time replays."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (iters * reps) * 1e3


SHAPES = [  # name, M, N, K, ln
    ('qkv', 1344, 768, 256, True), ('outp', 1344, 256, 256, False), ('ffn1', 1344, 1024, 256, True),
    ('ffn2', 1344, 256, 1024, False), ('ffn1_last', 224, 1024, 256, True), ('ffn2_last', 224, 256, 1024, False),
    ('inproj', 1344, 256, 128, False), ('q_sa', 224, 128, 128, True),
    ('pix_fc1', 131072, 128, 64, True), ('pix_fc2', 131072, 128, 128, False), ('pix_kv', 131072, 256, 128, True),
]
TRAIN_SHAPES = [  # STEVE decoder at the Physion training shape (72 frames x 1024 tokens, width 192, vocab 4096) and the dVAE
    ('head', 73728, 4096, 192, False), ('head_dx', 73728, 192, 4096, False), ('qkv', 73728, 576, 192, False),
    ('proj', 73728, 192, 192, False), ('ffn1', 73728, 768, 192, False), ('ffn2', 73728, 192, 768, False),
    ('dvae_1x1', 32768 * 4, 64, 64, False), ('dvae_logits', 32768, 4096, 64, False), ('dvae_z', 32768, 64, 4096, False),
]
CFGS = {'small': [3, 4, 7, 105, 106, 107, 108, 109, 115, 121, 122, 123], 'big': [100, 102, 103, 130, 131, 132, 133]}


def main():
    train = len(sys.argv) > 1 and sys.argv[1] == 'train'
    for name, M, N, K, ln in (TRAIN_SHAPES if train else SHAPES):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * K**-0.5
        b = torch.randn(N, device=dev)
        g, be = torch.ones(K, device=dev), torch.zeros(K, device=dev)
        res = []
        ref = None
        for cfg in ([100, 102, 103, 107, 130, 131, 132, 133] if train else CFGS['big' if M > 10000 else 'small']):
            os.environ['SF_GEMM_CFG'] = str(cfg)
            try:
                out = ops.linear(x, w, b, ln=(g, be) if ln else None)
                if ref is None:
                    ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(x, (K, )) if ln else x, w, b)
                err = (out - ref).abs().max().item()
                t = timeit(lambda: ops.linear(x, w, b, ln=(g, be) if ln else None))
                res.append((t, cfg, err))
            except RuntimeError as e:
                res.append((float('inf'), cfg, str(e)[:40]))
        res.sort()
        gf = 2.0 * M * N * K
        print(f'{name:10s} M={M} N={N} K={K} ln={ln}: ' + '  '.join(f'cfg{c}:{t:.1f}us' for t, c, _ in res) +
              f'   best {gf / res[0][0] / 1e6:.1f} TF  maxerr {max(e for _, _, e in res if isinstance(e, float)):.1e}')
    if train:
        os.environ.pop('SF_GEMM_CFG', None)
        return
    # conv
    x = torch.randn(32, 64, 64, 64, device=dev)
    w = ops.pack_conv_weight(torch.randn(64, 64, 5, 5, device=dev) * 0.03)
    b = torch.randn(64, device=dev)
    res = []
    for cfg in [28, 1, 100]:
        os.environ['SF_GEMM_CFG'] = str(cfg)
        t = timeit(lambda: ops.conv2d_nhwc(x, w, b), iters=5)
        res.append((t, cfg))
    os.environ.pop('SF_GEMM_CFG', None)
    res.append((timeit(lambda: ops.conv2d_nhwc(x, w, b), iters=5), 'halo'))
    res.sort()
    gf = 2.0 * 32 * 4096 * 64 * 1600
    print('conv 32 frames: ' + '  '.join(f'cfg{c}:{t:.0f}us' for t, c in res) + f'   best {gf / res[0][0] / 1e6:.1f} TF')
    os.environ.pop('SF_GEMM_CFG', None)


if __name__ == '__main__':
    main()
