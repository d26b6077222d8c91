"""CPU experiment (numpy, no GPU): error of a Winograd F(2x2, 5x5) form of the 64 -> 64 convolution under split-bf16 products against float64, beside the direct form.
    python tools/probes/winograd_error_probe.py      (needs sympy for the Toom-Cook matrices)"""
import numpy as np
np.random.seed(0)
# Winograd F(2,5) in 1-D via Cook-Toom with points 0, 1, -1, 2, -2, inf ; nested for 2-D.  n = m + r - 1 = 6
from fractions import Fraction as Fr
pts = [0, 1, -1, 2, -2]
m, r = 2, 5
n = m + r - 1
# Build matrices by the Toom-Cook construction: Y = A^T [(G g) * (B^T d)]
# Using polynomial evaluation: G (n x r): row i = [p_i^0 ... p_i^(r-1)] / N_i ; last row = [0..0 1]
# B^T from Lagrange ... use the standard construction via Vandermonde inverse
import sympy as sp
x = sp.symbols('x')
def toom(m, r, pts):
    n = m + r - 1
    # evaluation matrices
    def V(k):  # n x k Vandermonde incl. infinity
        M = sp.zeros(n, k)
        for i, p in enumerate(pts):
            for j in range(k):
                M[i, j] = sp.Integer(p) ** j
        M[n - 1, k - 1] = 1
        return M
    Vg, Vd = V(r), V(m)   # filter eval (n x r), output-size eval (n x m)
    # product polynomial degree n-1 interpolation: C = inverse of V(n)
    Vn = V(n)
    C = Vn.inv()          # n x n
    # linear convolution y_full = C (Vg g * Vd' d') ; correlation form via transposition principle:
    # y = A^T [ (G g) * (B^T d) ] with A^T = Vd^T, G = Vg, B^T = C^T
    return Vd.T, Vg, C.T
AT, G, BT = toom(m, r, pts)
AT = np.array(AT.tolist(), dtype=np.float64); G = np.array(G.tolist(), dtype=np.float64); BT = np.array(BT.tolist(), dtype=np.float64)
# check 1-D correctness in float64
g = np.random.randn(r); d = np.random.randn(n)
y = AT @ ((G @ g) * (BT @ d))
ref = np.array([np.dot(d[i:i + r], g) for i in range(m)])
print('1-D check', np.abs(y - ref).max())
def bf16(x):
    x = np.asarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.view(np.float32)
def split(x):
    h = bf16(x); l = bf16(x.astype(np.float32) - h)
    return h, l
def mm3(a, b):  # split-bf16 product, f32 accumulate: a [M,K], b [K,N]
    ah, al = split(a); bh, bl = split(b)
    return (ah.astype(np.float32) @ bh + ah @ bl + al @ bh).astype(np.float32)
C = 64; H = W = 32
xin = np.maximum(np.random.randn(C, H + 4, W + 4), 0).astype(np.float32)   # ReLU'd activations with halo
w = (0.03 * np.random.randn(C, C, 5, 5)).astype(np.float32)
# float64 direct
ref = np.zeros((C, H, W))
for ky in range(5):
    for kx in range(5):
        ref += np.einsum('oc,chw->ohw', w[:, :, ky, kx].astype(np.float64), xin[:, ky:ky + H, kx:kx + W].astype(np.float64))
# direct in split-bf16
dirv = np.zeros((C, H, W), dtype=np.float32)
for ky in range(5):
    for kx in range(5):
        dirv += mm3(w[:, :, ky, kx], xin[:, ky:ky + H, kx:kx + W].reshape(C, -1)).reshape(C, H, W)
# Winograd 2-D: U = G w G^T (per oc, c) [6x6]; V = B^T d B (per c, tile) ; M = sum_c U * V ; Y = A^T M A
U = np.einsum('ik,ockl,jl->ocij', G, w.astype(np.float64), G).astype(np.float32)           # transforms in f32 (from f64 here: weights are transformed once)
th, tw = H // 2, W // 2
tiles = np.zeros((C, th, tw, 6, 6), dtype=np.float32)
for ty in range(th):
    for tx in range(tw):
        tiles[:, ty, tx] = xin[:, 2 * ty:2 * ty + 6, 2 * tx:2 * tx + 6]
BT32 = BT.astype(np.float32)
V = np.einsum('ik,ctxkl,jl->ctxij', BT32, tiles, BT32).astype(np.float32)                  # input transform in f32
Mm = np.zeros((C, th, tw, 6, 6), dtype=np.float32)
for i in range(6):
    for j in range(6):
        Mm[:, :, :, i, j] = mm3(U[:, :, i, j], V[:, :, :, i, j].reshape(C, -1)).reshape(C, th, tw)
AT32 = AT.astype(np.float32)
Y = np.einsum('ik,otxkl,jl->otxij', AT32, Mm, AT32).astype(np.float32)
out = np.zeros((C, H, W), dtype=np.float32)
for ty in range(th):
    for tx in range(tw):
        out[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = Y[:, ty, tx]
sc = np.abs(ref).max()
print('direct split-bf16 vs f64: max rel', np.abs(dirv - ref).max() / sc)
print('winograd F(2x2,5x5) split-bf16 vs f64: max rel', np.abs(out - ref).max() / sc)
# exact-f32 winograd for reference
Mf = np.einsum('ocij,ctxij->otxij', U.astype(np.float32), V).astype(np.float32)
Yf = np.einsum('ik,otxkl,jl->otxij', AT32, Mf, AT32)
outf = np.zeros((C, H, W), dtype=np.float32)
for ty in range(th):
    for tx in range(tw):
        outf[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = Yf[:, ty, tx]
print('winograd f32 vs f64: max rel', np.abs(outf - ref).max() / sc)
print('|V| max / |d| max', np.abs(V).max() / np.abs(xin).max(), ' |U| max / |w| max', np.abs(U).max() / np.abs(w).max())
# balanced scaling: G row i / s_i, B^T row i * s_i with s_i = sum |G row i|
for name, sfun in (('s = sum|G row|', lambda G_: np.abs(G_).sum(1)), ('s = sqrt(sum|G row| / sum|BT row|)', lambda G_: np.sqrt(np.abs(G_).sum(1) / np.abs(BT).sum(1)))):
    sv = sfun(G)
    G2 = G / sv[:, None]; BT2 = BT * sv[:, None]
    U2 = np.einsum('ik,ockl,jl->ocij', G2, w.astype(np.float64), G2).astype(np.float32)
    V2 = np.einsum('ik,ctxkl,jl->ctxij', BT2.astype(np.float32), tiles, BT2.astype(np.float32)).astype(np.float32)
    M2 = np.zeros((C, th, tw, 6, 6), dtype=np.float32)
    for i in range(6):
        for j in range(6):
            M2[:, :, :, i, j] = mm3(U2[:, :, i, j], V2[:, :, :, i, j].reshape(C, -1)).reshape(C, th, tw)
    Y2 = np.einsum('ik,otxkl,jl->otxij', AT32, M2, AT32).astype(np.float32)
    o2 = np.zeros((C, H, W), dtype=np.float32)
    for ty in range(th):
        for tx in range(tw):
            o2[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = Y2[:, ty, tx]
    print(name, ': winograd split-bf16 vs f64 max rel', np.abs(o2 - ref).max() / sc, ' |U|/|w|', np.abs(U2).max() / np.abs(w).max(), ' |V|/|d|', np.abs(V2).max() / np.abs(xin).max())
