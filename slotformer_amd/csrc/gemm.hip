// Fused f32 GEMM core for gfx950:  C = epi( LN?(A) . W^T + bias ) (+ residual)
//
//   * exact-f32 MFMA (v_mfma_f32_32x32x2_f32): bitwise an fmaf chain, 157 TF peak.
//   * A [M,K] and W [N,K] are both K-contiguous (torch nn.Linear layout), so both
//     operands are staged as rows x k into LDS and read back with ds_read_b128;
//     the contraction index is permuted identically for A and B (k = 8*kb+4*h+s).
//   * A-loaders: plain rows (with a batched row map), im2col over an NHWC feature
//     map (5x5 conv as implicit GEMM) and im2col over the NCHW input image.
//   * prologue: per-row LayerNorm over K; epilogue: bias, ReLU, residual (with an
//     optional row modulo: position-embedding add), batched row map on C.
//
// Replaces the ATen/cuDNN call sites of SURVEY.md 2.3: conv2d (savi.py:230-240),
// per-pixel MLP (savi.py:245-250, 372-375), K/V projection (savi.py:66-70) and
// every nn.Linear of the rollout Transformer (slotformer.py:115-121).
#include <stdlib.h>

#include "sf_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { ALOAD_PLAIN = 0, ALOAD_CONV_NHWC = 1, ALOAD_CONV_NCHW = 2, ALOAD_DECONV_NHWC = 3 };

struct SfGemmArgs {
  const float* A;
  SfRowMap amap;
  const float* W;
  int ldw;
  const float* bias;
  const float* ln_g;
  const float* ln_b;
  float ln_eps;
  int ln_relu;
  const float* res;
  SfRowMap rmap;
  int res_mod;
  float* C;
  SfRowMap cmap;
  int M, N, K, relu;
  // conv (im2col loaders): output cH x cW, input cInH x cInW, cCin channels, cKs taps, stride
  int cH, cW, cInH, cInW, cCin, cKs, cStride;
  long long cFrameStride;
  // transposed conv, one output-parity class per launch (cSub = stride, 0 = off): rows enumerate the outputs
  // (cSub*cy + cPy, cSub*cx + cPx) of a cH x cW class grid; only the taps ky = cKy0 + cSub*ty, kx = cKx0 + cSub*tx that
  // reach such an output are contracted (k = (ty*cNtx + tx)*cCin + cin), input pixel (cy + cQy - ty, cx + cQx - tx)
  int cSub, cPy, cPx, cKy0, cKx0, cNtx, cQy, cQx;
  // dropout on act(acc + bias) before the residual (training, rollout_train.hip): element (row, col) is kept iff
  // sf_mix32((row * N + col) ^ drop_seed) >> 8 >= drop_thresh and scaled by drop_scale; drop_thresh == 0: off
  uint32_t drop_seed, drop_thresh;
  float drop_scale;
  // mask_mode (training backward passes): `res` is not added but gates the result, C = res > 0 ? act(acc + bias) * mask_scale : 0
  // (the ReLU / dropout adjoint folded into the GEMM that produces the gradient)
  int mask_mode;
  float mask_scale;
  int bf1;  // split-bf16 kernels: 1 = contract the hi parts only (precision mode 2); 2 = operands rounded to fp16, one fp16 MFMA (mode 3)
  int dbg;  // ablation bits (SF_DBG=gemm=<bits>, tools only): 1 no MFMA, 2 no main-loop loads, 4 no LN stats, 8 no stores
};

// BKT: k-chunk staged per barrier; KW waves split each chunk; PD: prefetch distance in chunks
// (PD = 2 keeps two chunks of global loads in flight behind the one being computed).
// BF3: split-bf16 arithmetic -- every f32 operand is split into hi + lo bf16 when staged into LDS and each
// k16 step issues hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with f32 accumulation (~2^-17 relative per
// operand; the dropped lo*lo term is ~2^-16 smaller still); ~5x fewer matrix-pipe cycles than exact-f32 MFMA.
template <int BM, int BN, int WM, int WN, int KW, int BKT, int PD, int NBUF, int ALOAD, bool LN, bool BF3>
__global__ __launch_bounds__(WM* WN* KW * 64) void sf_gemm_kernel(SfGemmArgs p) {
  constexpr int NT = WM * WN * KW * 64;
  static_assert(NT == 256 || NT == 512 || NT == 1024, "4, 8 or 16 waves");
  static_assert(BKT % (8 * KW) == 0 && (PD == 0 || PD == 1 || PD == 2) && (NBUF == 2 || (NBUF == 1 && PD == 1)), "bad chunking");
  static_assert(PD != 0 || ALOAD == ALOAD_PLAIN, "preload-all is for plain A rows");
  constexpr int RM = BM / (32 * WM), RN = BN / (32 * WN);
  constexpr int NKB = BKT / (8 * KW);  // 8-wide k blocks per wave per chunk
  constexpr int LSTR = BKT + 4;   // f32 row stride (floats)
  constexpr int LB = BKT + 8;     // bf16 row stride (elements) per hi/lo plane: (2*BKT+16) B = odd # of 16-B slots
  static_assert(!BF3 || (BKT % (16 * KW) == 0 && ALOAD != ALOAD_CONV_NCHW), "BF3 chunking");
  constexpr int NK16 = BKT / (16 * KW);
  constexpr int C4N = BKT / 4;
  constexpr int RS = NT / C4N;
  constexpr int A_IT = BM * C4N / NT;
  constexpr int B_IT = BN * C4N / NT;
  constexpr int RS_SC = NT / BKT;
  constexpr int A_SC = BM * BKT / NT;
  constexpr int B_SC = BN * BKT / NT;
  constexpr bool SCALAR = (ALOAD == ALOAD_CONV_NCHW);
  static_assert(A_IT >= 1 && B_IT >= 1, "tile too small");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + NBUF * BM * LSTR;
  // BF3 planes (bf16): A_hi | A_lo | B_hi | B_lo, each [NBUF][rows][LB]
  __bf16* Ah = (__bf16*)smem;
  __bf16* Al = Ah + NBUF * BM * LB;
  __bf16* Bh = Al + NBUF * BM * LB;
  __bf16* Bl = Bh + NBUF * BN * LB;
  float* stats = BF3 ? (float*)(Bl + NBUF * BN * LB) : (Bs + NBUF * BN * LSTR);

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wk = wave % KW, wn = (wave / KW) % WN, wm = wave / (KW * WN);
  // XCD-aware tile order for the conv: workgroup b lands on XCD b % 8, so give each XCD a contiguous
  // run of output tiles (neighbouring tiles share 4 of their 6 halo rows -> hits in that XCD's L2).
  int bid = blockIdx.x;
  if constexpr (ALOAD == ALOAD_CONV_NHWC || ALOAD == ALOAD_DECONV_NHWC) {
    const int nb = gridDim.x;
    if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  }
  const int m0 = bid * BM, n0 = blockIdx.y * BN;
  const int M = p.M, N = p.N, K = p.K;

  // ---- per-thread loader state ---------------------------------------------------
  // Every global load below is UNCONDITIONAL from a clamped (always valid) address and the value
  // is zeroed by a select where it matters: a branch around a load makes hipcc drain vmcnt(0)
  // at the join, which serialises the whole prefetch pipeline.  Rows >= M / >= N only feed
  // outputs that are never stored, so they need no zeroing at all.
  const int c4 = t % C4N, r0 = t / C4N;     // vector loader
  const int kk = t % BKT, r0s = t / BKT;    // scalar loader
  const float* arow[SCALAR ? 1 : A_IT];
  int cf[SCALAR ? A_SC : A_IT], cy[SCALAR ? A_SC : A_IT], cx[SCALAR ? A_SC : A_IT];
  long long rowbase[SCALAR ? 1 : A_IT];  // NHWC im2col: element offset of the row's own pixel
  if constexpr (ALOAD == ALOAD_PLAIN) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int m = min(m0 + r0 + i * RS, M - 1);
      arow[i] = p.A + sf_row_off(p.amap, m);
    }
  } else {
    constexpr int NI = SCALAR ? A_SC : A_IT;
    const int hw = p.cH * p.cW;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int m = min(m0 + (SCALAR ? (r0s + i * RS_SC) : (r0 + i * RS)), M - 1);
      const int f = m / hw, rem = m - f * hw;
      cf[i] = f;
      cy[i] = rem / p.cW;
      cx[i] = rem - cy[i] * p.cW;
      if constexpr (ALOAD == ALOAD_CONV_NHWC)   // the row's centre pixel in the input: (cy, cx) * stride
        rowbase[i] = (long long)f * p.cFrameStride + ((long long)(cy[i] * p.cStride) * p.cInW + cx[i] * p.cStride) * p.cCin;
    }
  }
  const float* wrow[SCALAR ? 1 : B_IT];
  if constexpr (!SCALAR) {
#pragma unroll
    for (int i = 0; i < B_IT; ++i) wrow[i] = p.W + (long long)min(n0 + r0 + i * RS, N - 1) * p.ldw;
  }

  struct Regs {
    f32x4 ra[SCALAR ? 1 : A_IT], rb[SCALAR ? 1 : B_IT];
    float sa[SCALAR ? A_SC : 1], sb[SCALAR ? B_SC : 1];
  };
  Regs R0, R1;

  auto load_tiles = [&](int kc, Regs& R) {
    f32x4(&ra)[SCALAR ? 1 : A_IT] = R.ra;
    f32x4(&rb)[SCALAR ? 1 : B_IT] = R.rb;
    float(&sa)[SCALAR ? A_SC : 1] = R.sa;
    float(&sb)[SCALAR ? B_SC : 1] = R.sb;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if ((p.dbg & 2) && kc > 0) return;
    if constexpr (!SCALAR) {
      const int k = kc * BKT + 4 * c4;
      const bool kok = k < K;
      const int kc4 = kok ? k : 0;
      int kw4 = kc4;   // offset of this k in a weight row (differs from k only for the parity-class transposed conv)
      if constexpr (ALOAD == ALOAD_PLAIN) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
          const f32x4 v = *(const f32x4*)(arow[i] + kc4);
          ra[i] = kok ? v : zero4;
        }
      } else if constexpr (ALOAD == ALOAD_DECONV_NHWC) {
        if (p.cSub) {
          const int tap = kc4 / p.cCin, cin = kc4 - tap * p.cCin;
          const int ty = tap / p.cNtx, tx = tap - ty * p.cNtx;
          kw4 = ((p.cKy0 + p.cSub * ty) * p.cKs + p.cKx0 + p.cSub * tx) * p.cCin + cin;
#pragma unroll
          for (int i = 0; i < A_IT; ++i) {
            const int iy = cy[i] + p.cQy - ty, ix = cx[i] + p.cQx - tx;
            const bool ok = kok && (unsigned)iy < (unsigned)p.cInH && (unsigned)ix < (unsigned)p.cInW;
            const int yc = min(max(iy, 0), p.cInH - 1), xc = min(max(ix, 0), p.cInW - 1);
            const f32x4 v = *(const f32x4*)(p.A + (long long)cf[i] * p.cFrameStride +
                                            ((long long)(yc * p.cInW + xc) * p.cCin + cin));
            ra[i] = ok ? v : zero4;
          }
        } else {
        // ConvTranspose2d gather: out(oy,ox) += in(iy,ix) * W[ky][kx] with oy = iy*s - pad + ky, i.e. the tap
        // contributes iff (oy + pad - ky) is a non-negative multiple of s inside the input.  k = tap*Cin + cin.
        const int tap = kc4 / p.cCin, cin = kc4 - tap * p.cCin;
        const int ky = tap / p.cKs, kx = tap - ky * p.cKs, pad = p.cKs >> 1, st = p.cStride;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
          const int ny = cy[i] + pad - ky, nx = cx[i] + pad - kx;
          const int iy = ny / st, ix = nx / st;
          const bool ok = kok && ny >= 0 && nx >= 0 && iy * st == ny && ix * st == nx && iy < p.cInH && ix < p.cInW;
          const int yc = min(max(iy, 0), p.cInH - 1), xc = min(max(ix, 0), p.cInW - 1);
          const f32x4 v = *(const f32x4*)(p.A + (long long)cf[i] * p.cFrameStride +
                                          ((long long)(yc * p.cInW + xc) * p.cCin + cin));
          ra[i] = ok ? v : zero4;
        }
        }
      } else {  // NHWC im2col: k = tap * Cin + cin
        int tap, cin, ky, kx;
        if (p.cCin == 64 && p.cKs == 5) {  // the reference's encoder shape: constant-folded index math
          tap = kc4 >> 6;
          cin = kc4 & 63;
          ky = tap / 5;
          kx = tap - ky * 5;
        } else {
          tap = kc4 / p.cCin;
          cin = kc4 - tap * p.cCin;
          ky = tap / p.cKs;
          kx = tap - ky * p.cKs;
        }
        const int pad = p.cKs >> 1;
        const int dy = ky - pad, dx = kx - pad;
        const long long tapoff = (long long)(dy * p.cInW + dx) * p.cCin + cin;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
          const int yy = cy[i] * p.cStride + dy, xx = cx[i] * p.cStride + dx;
          const bool ok = kok && (unsigned)yy < (unsigned)p.cInH && (unsigned)xx < (unsigned)p.cInW;
          // out-of-image taps read the row's own pixel (always valid) and are zeroed by the select
          const f32x4 v = *(const f32x4*)(p.A + rowbase[i] + (ok ? tapoff : (long long)cin));
          ra[i] = ok ? v : zero4;
        }
      }
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const f32x4 v = *(const f32x4*)(wrow[i] + kw4);
        rb[i] = kok ? v : zero4;
      }
    } else {  // NCHW image im2col, scalar: k = (c * ks + ky) * ks + kx
      const int k = kc * BKT + kk;
      const bool kok = k < K;
      const int kcl = kok ? k : 0;
      const int ks2 = p.cKs * p.cKs;
      const int c = kcl / ks2, r = kcl - c * ks2;
      const int ky = r / p.cKs, kx = r - ky * p.cKs, pad = p.cKs >> 1;
#pragma unroll
      for (int i = 0; i < A_SC; ++i) {
        const int yy = cy[i] * p.cStride + ky - pad, xx = cx[i] * p.cStride + kx - pad;
        const bool ok = kok && (unsigned)yy < (unsigned)p.cInH && (unsigned)xx < (unsigned)p.cInW;
        const int yc = min(max(yy, 0), p.cInH - 1), xc = min(max(xx, 0), p.cInW - 1);
        const float v = p.A[(long long)cf[i] * p.cFrameStride + ((long long)(c * p.cInH + yc) * p.cInW + xc)];
        sa[i] = ok ? v : 0.f;
      }
#pragma unroll
      for (int i = 0; i < B_SC; ++i) {
        const int n = min(n0 + r0s + i * RS_SC, N - 1);
        const float v = p.W[(long long)n * p.ldw + kcl];
        sb[i] = kok ? v : 0.f;
      }
    }
  };

  auto store_tiles = [&](int buf, int kc, Regs& R) {
    f32x4(&ra)[SCALAR ? 1 : A_IT] = R.ra;
    f32x4(&rb)[SCALAR ? 1 : B_IT] = R.rb;
    float(&sa)[SCALAR ? A_SC : 1] = R.sa;
    float(&sb)[SCALAR ? B_SC : 1] = R.sb;
    float* as = As + buf * BM * LSTR;
    float* bs = Bs + buf * BN * LSTR;
    if constexpr (!SCALAR) {
      if constexpr (LN) {
        const int k = kc * BKT + 4 * c4;
        const bool kok = k < K;
        const f32x4 g = *(const f32x4*)(p.ln_g + (kok ? k : 0));
        const f32x4 b = *(const f32x4*)(p.ln_b + (kok ? k : 0));
        const float lo = p.ln_relu ? 0.f : -INFINITY;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
          const int r = r0 + i * RS;
          const float mean = stats[r], rstd = stats[BM + r];
          f32x4 v = (ra[i] - mean) * rstd * g + b;
          v[0] = fmaxf(v[0], lo);
          v[1] = fmaxf(v[1], lo);
          v[2] = fmaxf(v[2], lo);
          v[3] = fmaxf(v[3], lo);
          ra[i] = kok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      if constexpr (BF3) {
        auto put = [&](__bf16* hi_plane, __bf16* lo_plane, int row, f32x4 v) {
          if (p.bf1 == 2) {   // precision mode 3 (probe): the hi plane holds the operand rounded to fp16, the lo plane is unused
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            *(f16x4*)(hi_plane + row * LB + 4 * c4) = __builtin_convertvector(v, f16x4);
            return;
          }
          const bf16x4 h = __builtin_convertvector(v, bf16x4);
          const bf16x4 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), bf16x4);
          *(bf16x4*)(hi_plane + row * LB + 4 * c4) = h;
          *(bf16x4*)(lo_plane + row * LB + 4 * c4) = l;
        };
#pragma unroll
        for (int i = 0; i < A_IT; ++i) put(Ah + buf * BM * LB, Al + buf * BM * LB, r0 + i * RS, ra[i]);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) put(Bh + buf * BN * LB, Bl + buf * BN * LB, r0 + i * RS, rb[i]);
      } else {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *(f32x4*)(as + (r0 + i * RS) * LSTR + 4 * c4) = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *(f32x4*)(bs + (r0 + i * RS) * LSTR + 4 * c4) = rb[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_SC; ++i) as[(r0s + i * RS_SC) * LSTR + kk] = sa[i];
#pragma unroll
      for (int i = 0; i < B_SC; ++i) bs[(r0s + i * RS_SC) * LSTR + kk] = sb[i];
    }
  };

  // ---- main loop -------------------------------------------------------------------
  f32x16 acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (K + BKT - 1) / BKT;
  const int a_off = (wm * RM * 32 + (lane & 31)) * LSTR + wk * (NKB * 8) + 4 * (lane >> 5);
  const int b_off = (wn * RN * 32 + (lane & 31)) * LSTR + wk * (NKB * 8) + 4 * (lane >> 5);
  auto compute = [&](int buf) {
    if constexpr (BF3) {
      const int ao = (wm * RM * 32 + (lane & 31)) * LB + wk * (NK16 * 16) + 8 * (lane >> 5);
      const int bo = (wn * RN * 32 + (lane & 31)) * LB + wk * (NK16 * 16) + 8 * (lane >> 5);
      const __bf16* ah = Ah + buf * BM * LB + ao;
      const __bf16* al = Al + buf * BM * LB + ao;
      const __bf16* bh = Bh + buf * BN * LB + bo;
      const __bf16* bl = Bl + buf * BN * LB + bo;
#pragma unroll
      for (int ks = 0; ks < NK16; ++ks) {
        bf16x8 xh[RM], xl[RM], yh[RN], yl[RN];
#pragma unroll
        for (int i = 0; i < RM; ++i) {
          xh[i] = *(const bf16x8*)(ah + i * 32 * LB + ks * 16);
          xl[i] = *(const bf16x8*)(al + i * 32 * LB + ks * 16);
        }
#pragma unroll
        for (int j = 0; j < RN; ++j) {
          yh[j] = *(const bf16x8*)(bh + j * 32 * LB + ks * 16);
          yl[j] = *(const bf16x8*)(bl + j * 32 * LB + ks * 16);
        }
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
          for (int j = 0; j < RN; ++j) {
            if (p.bf1 == 2) {   // single-pass fp16 (probe mode 3)
              typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xh[i]), __builtin_bit_cast(f16x8, yh[j]), acc[i][j], 0, 0, 0);
              continue;
            }
            if (!p.bf1) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[i], yh[j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], yl[j], acc[i][j], 0, 0, 0);
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], yh[j], acc[i][j], 0, 0, 0);
          }
      }
      return;
    }
    const float* as = As + buf * BM * LSTR + a_off;
    const float* bs = Bs + buf * BN * LSTR + b_off;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      f32x4 a[RM], b[RN];
#pragma unroll
      for (int i = 0; i < RM; ++i) a[i] = *(const f32x4*)(as + i * 32 * LSTR + kb * 8);
#pragma unroll
      for (int j = 0; j < RN; ++j) b[j] = *(const f32x4*)(bs + j * 32 * LSTR + kb * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
          for (int j = 0; j < RN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    }
  };
  auto compute_dbg = [&](int buf) {
    if (!(p.dbg & 1)) compute(buf);
  };
  // first chunk(s) are requested BEFORE the LayerNorm statistics pass so that pass does not add a
  // serial memory round trip in front of the main loop
  if constexpr (PD != 0) load_tiles(0, R0);
  if constexpr (PD == 2) {
    if (nk > 1) load_tiles(1, R1);
  }
  // PD == 0 ("preload all", K <= 4 chunks): every chunk is requested up front into its own register set and the
  // LayerNorm statistics come from those registers (a row's chunk is held by C4N consecutive lanes), so the
  // kernel pays a single memory latency and no separate statistics pass.
  Regs RR[PD == 0 ? 4 : 1];
  if constexpr (PD == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < nk) load_tiles(q, RR[q]);
    if constexpr (LN) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < nk) s += (RR[q].ra[i][0] + RR[q].ra[i][1]) + (RR[q].ra[i][2] + RR[q].ra[i][3]);  // k >= K loads are 0
        s = sf_group_sum<C4N>(s);
        const float mean = s / (float)K;
        float vs = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < nk && q * BKT + 4 * c4 < K) {
            const f32x4 dv = RR[q].ra[i] - mean;
            vs += (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]);
          }
        vs = sf_group_sum<C4N>(vs);
        if (c4 == 0) {
          stats[r0 + i * RS] = mean;
          stats[BM + r0 + i * RS] = 1.0f / sqrtf(vs / (float)K + p.ln_eps);
        }
      }
      __syncthreads();
    }
  }
  // ---- LayerNorm statistics for this tile's rows --------------------------------
  // NT/BM threads per row, every load of a pass independent (in flight together); the serial
  // per-row version this replaces cost ~30 us per launch in dependent L2 round trips.
  if (PD == 0) {
    // statistics already in LDS (above)
  } else if (LN && (p.dbg & 4)) {
    if (t < BM) {
      stats[t] = 0.f;
      stats[BM + t] = 1.f;
    }
    __syncthreads();
  } else if constexpr (LN) {
    constexpr int TPR = NT / BM;
    const int r = t / TPR, sub = t % TPR;
    const int m = min(m0 + r, M - 1);
    const float* rowp = p.A + sf_row_off(p.amap, m);
    float s = 0.f;
    {
#pragma unroll 4
      for (int k = sub * 4; k < K; k += TPR * 4) {
        const f32x4 v = *(const f32x4*)(rowp + k);
        s += (v[0] + v[1]) + (v[2] + v[3]);
      }
    }
    s = sf_group_sum<TPR>(s);
    const float mean = s / (float)K;
    float vs = 0.f;
    {
#pragma unroll 4
      for (int k = sub * 4; k < K; k += TPR * 4) {
        const f32x4 v = *(const f32x4*)(rowp + k) - mean;
        vs += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      }
    }
    vs = sf_group_sum<TPR>(vs);
    if (sub == 0) {
      stats[r] = mean;
      stats[BM + r] = 1.0f / sqrtf(vs / (float)K + p.ln_eps);
    }
    __syncthreads();
  }

  if constexpr (PD == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q < nk) {
        store_tiles(q & 1, q, RR[q]);  // buffer q&1 was last read two chunks ago; a barrier lies in between
        __syncthreads();
        compute_dbg(q & 1);
      }
    }
    __syncthreads();
  } else if constexpr (PD == 1 && NBUF == 2) {
    store_tiles(0, 0, R0);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
      const bool has_next = kc + 1 < nk;
      if (has_next) load_tiles(kc + 1, R0);
      compute_dbg(kc & 1);
      if (has_next) store_tiles((kc + 1) & 1, kc + 1, R0);
      __syncthreads();
    }
  } else if constexpr (PD == 1) {  // one LDS buffer holding a wide chunk (whole K when nk == 1)
    store_tiles(0, 0, R0);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
      const bool has_next = kc + 1 < nk;
      if (has_next) load_tiles(kc + 1, R0);
      compute_dbg(0);
      if (has_next) {
        __syncthreads();
        store_tiles(0, kc + 1, R0);
      }
      __syncthreads();
    }
  } else {
    store_tiles(0, 0, R0);
    __syncthreads();
    for (int kc = 0; kc < nk; kc += 2) {
      if (kc + 2 < nk) load_tiles(kc + 2, R0);
      compute_dbg(0);
      if (kc + 1 < nk) store_tiles(1, kc + 1, R1);
      __syncthreads();
      if (kc + 1 >= nk) break;
      if (kc + 3 < nk) load_tiles(kc + 3, R1);
      compute_dbg(1);
      if (kc + 2 < nk) store_tiles(0, kc + 2, R0);
      __syncthreads();
    }
  }

  // ---- split-K (across waves) reduction through LDS ------------------------------------
  if constexpr (KW > 1) {
    float* red = smem;  // [(KW-1)][WM*WN][RM*RN*16][64]
    if (wk > 0) {
      float* dst = red + ((long long)((wk - 1) * WM * WN + wm * WN + wn) * RM * RN * 16) * 64 + lane;
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((i * RN + j) * 16 + r) * 64] = acc[i][j][r];
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int q = 0; q < KW - 1; ++q) {
      const float* src = red + ((long long)(q * WM * WN + wm * WN + wn) * RM * RN * 16) * 64 + lane;
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += src[((i * RN + j) * 16 + r) * 64];
    }
  }

  // ---- epilogue ----------------------------------------------------------------------
  // All residual loads of a 32x32 block are issued together from clamped addresses (wave-uniform
  // branch only); per-element branches here cost one serialized memory round trip each.
#pragma unroll
  for (int j = 0; j < RN; ++j) {
    const int col = n0 + (wn * RN + j) * 32 + (lane & 31);
    const bool colok = col < N;
    const int colc = colok ? col : N - 1;
    const float bias = p.bias ? p.bias[colc] : 0.f;
#pragma unroll
    for (int i = 0; i < RM; ++i) {
      const int rbase = m0 + (wm * RM + i) * 32 + 4 * (lane >> 5);
      float rv[16];
      if (p.res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rowc = min(rbase + (r & 3) + 8 * (r >> 2), M - 1);
          const int rr = p.res_mod > 0 ? rowc % p.res_mod : rowc;
          rv[r] = p.res[sf_row_off(p.rmap, rr) + colc];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = 0.f;
      }
      const float lo = p.relu ? 0.f : -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        float t = fmaxf(acc[i][j][r] + bias, lo);
        if (p.drop_thresh) t = (sf_mix32((uint32_t)(row * N + col) ^ p.drop_seed) >> 8) >= p.drop_thresh ? t * p.drop_scale : 0.f;
        const float v = p.mask_mode ? (rv[r] > 0.f ? t * p.mask_scale : 0.f) : t + rv[r];
        int orow = row;
        if constexpr (ALOAD == ALOAD_DECONV_NHWC) {
          if (p.cSub) {   // class-grid row -> output pixel
            const int hw = p.cH * p.cW, f = row / hw, rem = row - f * hw, yy = rem / p.cW, xx = rem - yy * p.cW;
            orow = (f * p.cH * p.cSub + yy * p.cSub + p.cPy) * (p.cW * p.cSub) + xx * p.cSub + p.cPx;
          }
        }
        if (row < M && colok && !((p.dbg & 8) && v != 12345.f)) p.C[sf_row_off(p.cmap, orow) + col] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int KW, int BKT, int PD, int NBUF, int ALOAD, bool LN, bool BF3>
static int launch_cfg(const SfGemmArgs& a, hipStream_t stream) {
  constexpr int LSTR = BF3 ? BKT + 8 : BKT + 4;  // BF3: two bf16 planes of (BKT+8) elements = (BKT+8) floats per row
  constexpr size_t lds_main = (size_t)(NBUF * (BM + BN) * LSTR + 2 * BM) * sizeof(float);
  constexpr size_t lds_red = (size_t)(KW - 1) * BM * BN * sizeof(float);
  constexpr size_t lds = lds_main > lds_red ? lds_main : lds_red;
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = sf_gemm_kernel<BM, BN, WM, WN, KW, BKT, PD, NBUF, ALOAD, LN, BF3>;
  SF_TRY(sf_ensure_dyn_lds((const void*)kern, (size_t)(lds)));
  dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN);
  const int cls = ALOAD == ALOAD_PLAIN ? SF_K_LINEAR : (ALOAD == ALOAD_CONV_NCHW ? SF_K_CONV_FIRST : SF_K_CONV_NHWC);
  sf_prof_begin(cls, stream, 2.0 * (double)a.M * (double)a.N * (double)a.K);
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * KW * 64), lds, stream, a);
  sf_prof_end(cls, stream);
  SF_CHECK_LAUNCH();
  return 0;
}

// tuning override (tools/gemm_bench.py): SF_DBG=gemmcfg=<id> forces a tile configuration
static int forced_cfg() {
  static const int v = sf_dbg("gemmcfg");   // (SF_DBG=gemmcfg=<id>)
  return v > 0 ? v : -1;
}

// Tile configurations (ids are what tools/gemm_bench.py sweeps via SF_DBG=gemmcfg).
//   <BM, BN, WM, WN, KW, BKT, PD, NBUF>;  ids < 100: exact-f32 MFMA, ids >= 100: split-bf16 (BF3).
template <int ALOAD, bool LN>
static int launch_by_id(int id, const SfGemmArgs& a, hipStream_t st) {
  switch (id) {
    case 1: return launch_cfg<128, 64, 4, 1, 1, 32, 1, 2, ALOAD, LN, false>(a, st);
    case 3: return launch_cfg<32, 32, 1, 1, 4, 128, 1, 2, ALOAD, LN, false>(a, st);
    case 4: return launch_cfg<32, 64, 1, 2, 2, 64, 1, 2, ALOAD, LN, false>(a, st);
    case 7: return launch_cfg<32, 32, 1, 1, 4, 128, 2, 2, ALOAD, LN, false>(a, st);
    case 28: return launch_cfg<128, 64, 4, 2, 1, 32, 2, 2, ALOAD, LN, false>(a, st);
    case 31: return launch_cfg<128, 128, 2, 4, 1, 32, 1, 2, ALOAD, LN, false>(a, st);
    default: break;
  }
  if constexpr (ALOAD != ALOAD_CONV_NCHW) {
    switch (id) {
      case 100: return launch_cfg<128, 64, 4, 2, 1, 32, 2, 2, ALOAD, LN, true>(a, st);
      case 102: return launch_cfg<128, 128, 2, 4, 1, 32, 1, 2, ALOAD, LN, true>(a, st);
      case 103: return launch_cfg<128, 128, 2, 4, 1, 64, 2, 2, ALOAD, LN, true>(a, st);
      case 105: return launch_cfg<32, 32, 1, 1, 4, 128, 2, 2, ALOAD, LN, true>(a, st);
      case 106: return launch_cfg<32, 32, 1, 1, 4, 128, 1, 2, ALOAD, LN, true>(a, st);
      case 107: return launch_cfg<64, 64, 2, 2, 1, 64, 2, 2, ALOAD, LN, true>(a, st);
      case 108: return launch_cfg<64, 64, 2, 2, 2, 128, 2, 2, ALOAD, LN, true>(a, st);
      case 109: return launch_cfg<32, 64, 1, 2, 4, 128, 2, 2, ALOAD, LN, true>(a, st);
      case 115: return launch_cfg<32, 32, 1, 1, 8, 256, 2, 2, ALOAD, LN, true>(a, st);
      // single-buffered 128x64 tiles: 31 / 55 KB of LDS, several workgroups per CU cover each other's prologue / epilogue
      case 132: return launch_cfg<128, 64, 4, 2, 1, 64, 1, 1, ALOAD, LN, true>(a, st);
      case 133: return launch_cfg<128, 64, 4, 2, 1, 32, 1, 1, ALOAD, LN, true>(a, st);
      default: break;
    }
  }
  if constexpr (ALOAD == ALOAD_PLAIN) {  // preload-all variants (K <= 4 chunks)
    if ((id == 121 || id == 122 || id == 123) && a.K > 4 * (id == 123 ? 128 : 64))
      return sf_set_err(-1, "preload-all GEMM configuration needs K <= 4 chunks", __FILE__, __LINE__);
    switch (id) {
      case 121: return launch_cfg<64, 64, 2, 2, 1, 64, 0, 2, ALOAD, LN, true>(a, st);
      case 122: return launch_cfg<64, 64, 2, 2, 2, 64, 0, 2, ALOAD, LN, true>(a, st);
      case 123: return launch_cfg<32, 64, 1, 2, 4, 128, 0, 2, ALOAD, LN, true>(a, st);
      // single-buffered 128x128 tiles: 41 / 74 KB of LDS, so two or three workgroups share a CU and cover each other's
      // prologue / epilogue (the double-buffered 128x128 tiles above own a CU alone)
      case 130: return launch_cfg<128, 128, 2, 4, 1, 32, 1, 1, ALOAD, LN, true>(a, st);
      case 131: return launch_cfg<128, 128, 2, 4, 1, 64, 1, 1, ALOAD, LN, true>(a, st);
      default: break;
    }
  }
  return sf_set_err(-1, "unknown GEMM configuration id", __FILE__, __LINE__);
}

// 0: exact f32 MFMA everywhere; 1: split-bf16 MFMA (default); 2: single-pass bf16 MFMA with f32 accumulation in the GEMM / conv
// / weight-gradient cores (the "AMP-bf16" policy of the training path: fp32 storage and master weights, one bf16 rounding of
// each operand; kernels without a single-pass variant keep mode 1).  SF_PRECISION=f32 | bf16 selects 0 | 2 at load time.
static int g_precision = -1;
extern "C" int sf_get_precision(void) {
  const int t = sf_thread_opts().precision;   // a per-call option of this thread (sf_rollout_opts) wins over the process default
  if (t >= 0) return t;
  if (g_precision < 0) {
    const char* e = getenv("SF_PRECISION");
    g_precision = (e && (e[0] == 'f' || e[0] == '0')) ? 0 : ((e && strcmp(e, "bf16") == 0) ? 2 : 1);
  }
  return g_precision;
}
extern "C" int sf_set_precision(int mode) {
  if (mode < 0 || mode > 2) return sf_set_err(-1, "invalid argument: precision mode must be 0 (f32), 1 (bf16x3) or 2 (bf16)", __FILE__, __LINE__);   // (3 = single-pass fp16 exists per call only: sf_rollout_opts.precision, a measurement probe)
  g_precision = mode;
  return 0;
}

template <int ALOAD, bool LN>
static int dispatch_tiles(const SfGemmArgs& a, hipStream_t stream) {
  auto tiles = [&](int bm, int bn) {
    return (long long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn);
  };
  if constexpr (ALOAD == ALOAD_CONV_NCHW) {
    return launch_cfg<128, 64, 4, 1, 1, 32, 1, 2, ALOAD, false, false>(a, stream);
  } else {
    const int f = forced_cfg();
    if (f >= 0) return launch_by_id<ALOAD, LN>(f, a, stream);
    const bool bf3 = sf_get_precision() >= 1;
    // choices below come from tools/gemm_bench.py on MI355X (profiles/r01_gemm_configs.txt)
    if constexpr (ALOAD == ALOAD_CONV_NHWC || ALOAD == ALOAD_DECONV_NHWC) {
      static const int conv_cfg = sf_dbg("convcfg") > 0 ? sf_dbg("convcfg") : 132;
      return launch_by_id<ALOAD, LN>(bf3 ? conv_cfg : 28, a, stream);
    } else {
      // big problems (tools/gemm_bench.py [train], profiles/r01_gemm_configs.txt): the single-buffered tiles win everywhere --
      // with 31 .. 74 KB of LDS two or more workgroups share a CU, which hides the fill of the first chunk and the C store
      // that a double-buffered 128x128 workgroup (147 KB, alone on its CU) leaves exposed
      if (a.N > 64 && tiles(128, 128) >= 384) {
        if (!bf3) return launch_by_id<ALOAD, LN>(31, a, stream);
        return launch_by_id<ALOAD, LN>((a.N % 128 == 0 || a.N >= 512) ? 131 : 133, a, stream);
      }
      if (tiles(128, 64) >= 384) return launch_by_id<ALOAD, LN>(bf3 ? (a.K >= 2048 ? 132 : 133) : 1, a, stream);
      // small-M regime (rollout / slot-level GEMMs): latency-bound, favour many small workgroups
      if (bf3) {
        if (a.K >= 512) return launch_by_id<ALOAD, LN>(a.M >= 512 ? 109 : 115, a, stream);
        const long long t64 = tiles(64, 64);
        if (t64 > 300) return launch_by_id<ALOAD, LN>(a.K <= 256 ? 122 : 107, a, stream);
        if (t64 >= 192) return launch_by_id<ALOAD, LN>(108, a, stream);
        return launch_by_id<ALOAD, LN>(a.M >= 512 ? 109 : 106, a, stream);
      }
      if (a.K >= 512) return launch_by_id<ALOAD, LN>(7, a, stream);
      if (tiles(32, 64) >= 256) return launch_by_id<ALOAD, LN>(4, a, stream);
      return launch_by_id<ALOAD, LN>(3, a, stream);
    }
  }
}

int sf_gemm_dispatch(const SfGemmArgs& a_in, int aload, hipStream_t stream) {
  if (a_in.M <= 0 || a_in.N <= 0) return 0;
  SfGemmArgs a = a_in;
  {
    a.dbg = sf_dbg("gemm");
  }
  a.bf1 = sf_get_precision() == 2 ? 1 : (sf_get_precision() == 3 ? 2 : 0);
  const bool ln = a.ln_g != nullptr;
  if (aload == ALOAD_PLAIN) return ln ? dispatch_tiles<ALOAD_PLAIN, true>(a, stream)
                                      : dispatch_tiles<ALOAD_PLAIN, false>(a, stream);
  if (aload == ALOAD_CONV_NHWC) return dispatch_tiles<ALOAD_CONV_NHWC, false>(a, stream);
  if (aload == ALOAD_DECONV_NHWC) return dispatch_tiles<ALOAD_DECONV_NHWC, false>(a, stream);
  return dispatch_tiles<ALOAD_CONV_NCHW, false>(a, stream);
}

// internal C++ helper used by the engine
int sf_linear_ex(const float* A, SfRowMap amap, const float* W, const float* bias, const float* ln_g,
                 const float* ln_b, float ln_eps, const float* res, SfRowMap rmap, int res_mod,
                 float* C, SfRowMap cmap, int M, int N, int K, int relu, hipStream_t stream, int ln_relu) {
  SfGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.amap = amap; a.W = W; a.ldw = K; a.bias = bias;
  a.ln_g = ln_g; a.ln_b = ln_b; a.ln_eps = ln_eps; a.ln_relu = ln_relu;
  a.res = res; a.rmap = rmap; a.res_mod = res_mod;
  a.C = C; a.cmap = cmap; a.M = M; a.N = N; a.K = K; a.relu = relu;
  return sf_gemm_dispatch(a, ALOAD_PLAIN, stream);
}

// C = res + dropout(act(A . W^T + bias)): the GEMM of a training forward pass
int sf_linear_dropout_ex(const float* A, const float* W, const float* bias, const float* res, float* C, int M, int N, int K,
                         int relu, uint32_t drop_seed, uint32_t drop_thresh, float drop_scale, hipStream_t stream) {
  SfGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.amap = sf_rows(K); a.W = W; a.ldw = K; a.bias = bias;
  a.res = res; a.rmap = sf_rows(N); a.C = C; a.cmap = sf_rows(N); a.M = M; a.N = N; a.K = K; a.relu = relu;
  a.drop_seed = drop_seed; a.drop_thresh = drop_thresh; a.drop_scale = drop_scale;
  return sf_gemm_dispatch(a, ALOAD_PLAIN, stream);
}

// C = mask > 0 ? (A . W^T) * scale : 0   (the data-gradient GEMM of a ReLU(+dropout) layer; mask = its saved output)
int sf_linear_masked_ex(const float* A, const float* W, const float* mask, float scale, float* C, long long M, int N, int K,
                        hipStream_t stream) {
  SfGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.amap = sf_rows(K); a.W = W; a.ldw = K;
  a.res = mask; a.rmap = sf_rows(N); a.mask_mode = 1; a.mask_scale = scale;
  a.C = C; a.cmap = sf_rows(N); a.M = (int)M; a.N = N; a.K = K;
  return sf_gemm_dispatch(a, ALOAD_PLAIN, stream);
}

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" {

int sf_linear_f32(const float* A, int lda, const float* W, const float* bias, const float* ln_gamma,
                  const float* ln_beta, float ln_eps, const float* residual, int ldr, float* C, int ldc,
                  int M, int N, int K, int relu, void* stream) {
  SF_REQUIRE(A && W && C, "null pointer");
  SF_REQUIRE(M >= 0 && N > 0 && K > 0 && (K % 4) == 0, "K must be a positive multiple of 4");
  SF_REQUIRE(lda >= K && ldc >= N && (lda % 4) == 0, "bad leading dimension");
  SF_REQUIRE((ln_gamma == nullptr) == (ln_beta == nullptr), "ln_gamma/ln_beta must come together");
  SF_REQUIRE(residual == nullptr || ldr >= N, "bad residual leading dimension");
  return sf_linear_ex(A, sf_rows(lda), W, bias, ln_gamma, ln_beta, ln_eps, residual, sf_rows(ldr), 0, C,
                      sf_rows(ldc), M, N, K, relu, (hipStream_t)stream, 0);
}

// 5x5 (ks x ks) conv, stride 1, "same" padding, NHWC in -> NHWC out, Cin multiple of 4.
// w_packed: [Cout][ks][ks][Cin]  (= torch weight.permute(0,2,3,1)); add: optional [H*W][Cout] table
// added after the activation (soft position embedding, utils.py:60-63).
int sf_conv2d_nhwc_f32(const float* in, const float* w_packed, const float* bias, const float* add,
                       float* out, int F, int H, int W, int Cin, int Cout, int ks, int relu,
                       void* stream) {
  SF_REQUIRE(in && w_packed && out, "null pointer");
  SF_REQUIRE(F >= 0 && H > 0 && W > 0 && Cin > 0 && (Cin % 4) == 0 && Cout > 0 && (ks & 1), "bad conv shape");
  if (sf_get_precision() >= 1 && forced_cfg() < 0) {
    // encoder shape (5x5, 64->64, 64-wide rows): halo-resident kernel (conv_halo.hip)
    constexpr bool halo_on = true;
    if (halo_on) {
      const int rc = sf_conv5x5_halo_ex(in, w_packed, bias, add, out, F, H, W, Cin, Cout, ks, relu, (hipStream_t)stream);
      if (rc != 1) return rc;
    }
  }
  SfGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = in; a.W = w_packed; a.ldw = ks * ks * Cin; a.bias = bias;
  a.res = add; a.rmap = sf_rows(Cout); a.res_mod = H * W;
  a.C = out; a.cmap = sf_rows(Cout);
  a.M = F * H * W; a.N = Cout; a.K = ks * ks * Cin; a.relu = relu;
  if (relu == 2) {   // `add` is a per-frame ReLU mask (see conv_halo.hip)
    a.relu = 0; a.res_mod = 0; a.mask_mode = 1; a.mask_scale = 1.f;
  }
  a.cH = H; a.cW = W; a.cInH = H; a.cInW = W; a.cCin = Cin; a.cKs = ks; a.cStride = 1;
  a.cFrameStride = (long long)H * W * Cin;
  return sf_gemm_dispatch(a, ALOAD_CONV_NHWC, (hipStream_t)stream);
}

}  // extern "C"

// Conv2d(k, stride, padding=k/2) on NHWC, [F,Hin,Win,Cin] -> [F,Hin/stride,Win/stride,Cout]; w_packed [Cout][ks][ks][Cin].
// With w_packed = pack_conv(torch ConvTranspose2d weight [Cin_t, Cout_t, k, k]) this is the backward-data pass of
// sf_conv_transpose2d_nhwc_f32 (training of the image loss, savi_decode_train.hip).
int sf_conv2d_nhwc_strided_ex(const float* in, const float* w_packed, const float* bias, float* out, int F, int Hin, int Win,
                              int Cin, int Cout, int ks, int stride, int relu, hipStream_t stream, const float* relu_mask) {
  SF_REQUIRE(in && w_packed && out, "null pointer");
  SF_REQUIRE(F >= 0 && Hin > 0 && Win > 0 && Cin > 0 && (Cin % 4) == 0 && Cout > 0 && (ks & 1) && stride >= 1 &&
                 Hin % stride == 0 && Win % stride == 0, "bad strided conv shape");
  if (stride == 1)   // the plain "same" convolution, with its halo-resident fast path
    return sf_conv2d_nhwc_f32(in, w_packed, bias, relu_mask, out, F, Hin, Win, Cin, Cout, ks, relu_mask ? 2 : relu, (void*)stream);
  SfGemmArgs a;
  memset(&a, 0, sizeof(a));
  const int Ho = Hin / stride, Wo = Win / stride;
  a.A = in; a.W = w_packed; a.ldw = ks * ks * Cin; a.bias = bias;
  a.C = out; a.cmap = sf_rows(Cout); a.rmap = sf_rows(Cout);
  a.M = F * Ho * Wo; a.N = Cout; a.K = ks * ks * Cin; a.relu = relu;
  if (relu_mask) {   // out = relu_mask > 0 ? conv : 0, relu_mask laid out like out
    a.res = relu_mask; a.mask_mode = 1; a.mask_scale = 1.f;
  }
  a.cH = Ho; a.cW = Wo; a.cInH = Hin; a.cInW = Win; a.cCin = Cin; a.cKs = ks; a.cStride = stride;
  a.cFrameStride = (long long)Hin * Win * Cin;
  return sf_gemm_dispatch(a, ALOAD_CONV_NHWC, stream);
}

extern "C" {
// ConvTranspose2d(k, stride, padding=k/2, output_padding=stride-1): NHWC in [F,Hin,Win,Cin] -> NHWC out
// [F,Hin*stride,Win*stride,Cout]; w_packed [Cout][ks][ks][Cin] = torch weight[Cin,Cout,k,k].permute(1,2,3,0).
int sf_conv_transpose2d_nhwc_f32(const float* in, const float* w_packed, const float* bias, float* out, int F,
                                 int Hin, int Win, int Cin, int Cout, int ks, int stride, int relu, void* stream) {
  SF_REQUIRE(in && w_packed && out, "null pointer");
  SF_REQUIRE(F >= 0 && Hin > 0 && Win > 0 && Cin > 0 && (Cin % 4) == 0 && Cout > 0 && (ks & 1) && stride >= 1,
             "bad deconv shape");
  SfGemmArgs a;
  memset(&a, 0, sizeof(a));
  const int Ho = Hin * stride, Wo = Win * stride, pad = ks / 2;
  a.A = in; a.W = w_packed; a.ldw = ks * ks * Cin; a.bias = bias;
  a.C = out; a.cmap = sf_rows(Cout); a.rmap = sf_rows(Cout);
  a.N = Cout; a.relu = relu;
  a.cInH = Hin; a.cInW = Win; a.cCin = Cin; a.cKs = ks; a.cStride = stride;
  a.cFrameStride = (long long)Hin * Win * Cin;
  constexpr int by_class = 1;   // (0: the single-launch gather over all ks*ks taps)
  if (stride > 1 && by_class) {
    // one launch per output-parity class: an output (oy, ox) only receives the taps with ky = (oy + pad) mod stride
    // (mod stride), so the ks*ks-tap gather would multiply (stride^2 - 1) / stride^2 structural zeros
    for (int py = 0; py < stride; ++py)
      for (int px = 0; px < stride; ++px) {
        SfGemmArgs c = a;
        c.cSub = stride; c.cPy = py; c.cPx = px;
        c.cKy0 = (py + pad) % stride; c.cKx0 = (px + pad) % stride;
        const int nty = (ks - c.cKy0 + stride - 1) / stride, ntx = (ks - c.cKx0 + stride - 1) / stride;
        c.cNtx = ntx;
        c.cQy = (py + pad - c.cKy0) / stride; c.cQx = (px + pad - c.cKx0) / stride;
        c.cH = Hin; c.cW = Win;   // class grid: Ho / stride x Wo / stride
        c.M = F * Hin * Win; c.K = nty * ntx * Cin;
        SF_TRY(sf_gemm_dispatch(c, ALOAD_DECONV_NHWC, (hipStream_t)stream));
      }
    return 0;
  }
  a.M = F * Ho * Wo; a.K = ks * ks * Cin;
  a.cH = Ho; a.cW = Wo;
  return sf_gemm_dispatch(a, ALOAD_DECONV_NHWC, (hipStream_t)stream);
}

// first conv: NCHW image (frame f at img + f*frame_stride floats) -> NHWC, "same" padding k//2.
// weight: torch layout [Cout][Cin][ks][ks].
int sf_conv2d_nchw_in_f32(const float* img, long long frame_stride, const float* weight,
                          const float* bias, const float* add, float* out, int F, int Cin, int Hin,
                          int Win, int Cout, int ks, int stride, int relu, void* stream) {
  SF_REQUIRE(img && weight && out, "null pointer");
  SF_REQUIRE(F >= 0 && Cin > 0 && Hin > 0 && Win > 0 && Cout > 0 && (ks & 1) && stride >= 1, "bad conv shape");
  if (sf_get_precision() >= 1 && forced_cfg() < 0) {
    // 3 -> 64 channels, 5x5, 64-wide output: input halo + patch matrix in LDS (conv_first.hip)
    constexpr bool on = true;
    if (on) {
      const int rc = sf_conv_first_ex(img, frame_stride, weight, bias, add, out, F, Cin, Hin, Win, Cout, ks, stride, relu,
                                      (hipStream_t)stream);
      if (rc != 1) return rc;
    }
  }
  const int pad = ks / 2;
  const int Ho = (Hin + 2 * pad - ks) / stride + 1, Wo = (Win + 2 * pad - ks) / stride + 1;
  SfGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = img; a.W = weight; a.ldw = Cin * ks * ks; a.bias = bias;
  a.res = add; a.rmap = sf_rows(Cout); a.res_mod = Ho * Wo;
  a.C = out; a.cmap = sf_rows(Cout);
  a.M = F * Ho * Wo; a.N = Cout; a.K = Cin * ks * ks; a.relu = relu;
  a.cH = Ho; a.cW = Wo; a.cInH = Hin; a.cInW = Win; a.cCin = Cin; a.cKs = ks; a.cStride = stride;
  a.cFrameStride = frame_stride;
  return sf_gemm_dispatch(a, ALOAD_CONV_NCHW, (hipStream_t)stream);
}

}  // extern "C"
