// The SAVi image encoder under autograd: conv stack + soft position embedding + per-pixel MLP
// (savi.py:220-250, 367-377; utils.py:52-63) with every weight gradient.
//
//   forward : conv0 (NCHW image, 3 -> 64, stride 2 at 128x128) + ReLU, convs 1.. (64 -> 64 on the halo-resident 5x5 kernel)
//             + ReLU except after the last, + position table, LN -> Linear + ReLU -> Linear.  All layer outputs are kept.
//   backward: MLP (weight gradients on the split-bf16 TN contraction of rollout_train.hip), LayerNorm, position-embedding
//             Linear(4 -> C) from the frame-summed gradient, then per conv: bias = column sum, weight gradient =
//             conv_wgrad_kernel (the same TN contraction with the activation rows shifted by the tap: one 64x64 output
//             tile per tap, no im2col buffer), data gradient = the convolution of the gradient with the flipped /
//             transposed kernel on the same halo-resident kernel, ReLU mask.  conv0's weight gradient contracts against
//             an im2col matrix of the image (K = 75 padded to 128; the only materialised patch matrix).
#include "../../include/slotformer_hip.h"
#include "sf_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CW_P 68
__device__ __forceinline__ void cw_split8(const f32x8 v, bf16x8& hi, bf16x8& lo) {
  hi = __builtin_convertvector(v, bf16x8);
  lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x8), bf16x8);
}

// dW[n][tap][k] = sum over pixels (f, y, x) of A[f, y, x, n] * X[f, y*s + dy, x*s + dx, k]:  A [rows, CA] lives on an H x W
// grid, X [*, 64] on an Hx x Wx grid, (dy, dx) = tap - ks/2, zero outside the image.
//   conv weight gradient       : A = dY (n = c_out), X = the layer input (k = c_in), s = 1
//   transposed-conv weight grad: A = the layer input (n = c_in), X = dY on the s-times finer output grid (k = c_out)
// One workgroup owns ONE KERNEL ROW ky of one 64-wide n tile: the chunk of A rows is staged once and contracted against the
// five kx-shifted copies of the X rows (loaded back to back, so the shifted re-reads hit in L2 -- with one tap per
// workgroup the operands were streamed from HBM 25 times and the kernel sat at 59 TFLOP/s, HBM-bound).
// grid (5 * CA/64, splits); partial [split][CA][25*64].  Structure of grad_gemm_tn_kernel otherwise.
template <int MODE>   // 0: all four split products, 1: split-bf16 (no lo*lo), 2: single-pass bf16
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                         float* __restrict__ partial, long long rows, int rps, int H, int W,
                                                         int CA, int s, int Hx, int Wx) {
  constexpr int KS = 5;
  __shared__ float Ys[32 * CW_P];
  __shared__ float Xs[KS][32 * CW_P];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nt = CA / 64, ky = blockIdx.x / nt, n0 = (blockIdx.x % nt) * 64, dy = ky - KS / 2;
  const long long r0 = (long long)blockIdx.y * rps;
  const long long r1 = r0 + rps < rows ? r0 + rps : rows;
  const int wn = (wave >> 1) * 32, wk = (wave & 1) * 32;
  const int lr = tid >> 4, lc = (tid & 15) * 4;
  f32x16 acc[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const int nrows = (int)(r1 - r0);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int hw = H * W;
  f32x4 py[2], px[2][KS];   // ext-vector types: arrays of HIP's float4 struct were kept in scratch memory
  // Every load is UNCONDITIONAL from a clamped (always valid) address and zeroed by a select afterwards: a branch around a
  // load makes the compiler wait for all outstanding loads at the join, which serialises the prefetch with the MFMAs.
  const int rlast = (int)(rows - 1);
  // rows rel + lr, rel + lr + 16 of this workgroup's range (a macro, not a lambda: captured arrays ended up in scratch)
#define CW_FETCH(rel)                                                                                                    \
  _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                                        \
    const int rr0 = (rel) + lr + 16 * h;                                                                                 \
    const bool rok = rr0 < nrows;                                                                                        \
    const int r = min((int)r0 + rr0, rlast);                                                                             \
    const f32x4 a_ = *reinterpret_cast<const f32x4*>(dY + (long long)r * CA + n0 + lc);                                \
    py[h] = rok ? a_ : zero4;                                                                                            \
    const int f = r / hw, rr = r - f * hw, y = rr / W, x = rr - y * W;                                                   \
    const int yy = y * s + dy;                                                                                           \
    const bool yok = rok && (unsigned)yy < (unsigned)Hx;                                                                 \
    const float* xrow = X + ((long long)(f * Hx + min(max(yy, 0), Hx - 1)) * Wx) * 64 + lc;                              \
    _Pragma("unroll") for (int t = 0; t < KS; ++t) {                                                                     \
      const int xx = x * s + t - KS / 2;                                                                                 \
      const f32x4 v_ = *reinterpret_cast<const f32x4*>(xrow + (long long)min(max(xx, 0), Wx - 1) * 64);                \
      px[h][t] = (yok && (unsigned)xx < (unsigned)Wx) ? v_ : zero4;                                                      \
    }                                                                                                                    \
  }
  CW_FETCH(0)
  for (int rb = 0; rb < nrows; rb += 32) {
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<f32x4*>(&Ys[(lr + 16 * h) * CW_P + lc]) = py[h];
#pragma unroll
      for (int t = 0; t < KS; ++t) *reinterpret_cast<f32x4*>(&Xs[t][(lr + 16 * h) * CW_P + lc]) = px[h][t];
    }
    __syncthreads();
    CW_FETCH(rb + 32)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ro = (16 * kk + 8 * (lane >> 5)) * CW_P;
      f32x8 a;
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = Ys[ro + j * CW_P + wn + (lane & 31)];
      bf16x8 ah, al;
      cw_split8(a, ah, al);
#pragma unroll
      for (int t = 0; t < KS; ++t) {
        f32x8 b;
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = Xs[t][ro + j * CW_P + wk + (lane & 31)];
        bf16x8 bh, bl;
        cw_split8(b, bh, bl);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
        if (MODE != 2) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
        }
        if (MODE == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, acc[t], 0, 0, 0);
      }
    }
  }
  const int K = KS * KS * 64;
  float* out = partial + (long long)blockIdx.y * CA * K;
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wn + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      out[(long long)n * K + (ky * KS + t) * 64 + wk + (lane & 31)] = acc[t][r];
    }
}

// The same contraction for grids whose rows are a multiple of 32 pixels wide (the 64x64 and 32x32 maps that carry nearly all
// the work): a chunk of 32 A pixels lies in one image row, so the X rows its five kx taps need are ONE window of 31*S + 5
// consecutive pixels of image row y*S + dy.  The window is loaded once into LDS and every tap reads its rows out of it
// (row j*S + kx) -- 2.3x (S = 1) / 1.5x (S = 2) fewer global loads than fetching five shifted copies.
template <int MODE, int S>
__global__ __launch_bounds__(256) void conv_wgrad_win_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                             float* __restrict__ partial, long long rows, int rps, int H, int W,
                                                             int CA, int Hx, int Wx) {
  constexpr int KS = 5, WR = 31 * S + 5, NW = (WR * 16 + 255) / 256;
  __shared__ float Ys[32 * CW_P];
  __shared__ float Xw[(WR + 1) * CW_P];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nt = CA / 64, ky = blockIdx.x / nt, n0 = (blockIdx.x % nt) * 64, dy = ky - KS / 2;
  const long long r0 = (long long)blockIdx.y * rps;
  const long long r1 = r0 + rps < rows ? r0 + rps : rows;
  const int wn = (wave >> 1) * 32, wk = (wave & 1) * 32;
  const int lr = tid >> 4, lc = (tid & 15) * 4;
  f32x16 acc[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const int nrows = (int)(r1 - r0);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int hw = H * W, rlast = (int)(rows - 1);
  f32x4 pa[2], pw[NW];
  // unconditional loads from clamped addresses + selects (see conv_wgrad_kernel)
#define CWW_FETCH(rel)                                                                                                  \
  {                                                                                                                     \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                                     \
      const int rr0 = (rel) + lr + 16 * h;                                                                              \
      const f32x4 a_ = *reinterpret_cast<const f32x4*>(dY + (long long)min((int)r0 + rr0, rlast) * CA + n0 + lc);       \
      pa[h] = rr0 < nrows ? a_ : zero4;                                                                                 \
    }                                                                                                                   \
    const int rc = min((int)r0 + (rel), rlast);                                                                         \
    const int f = rc / hw, rr = rc - f * hw, y = rr / W, x0 = rr - y * W;                                               \
    const int yy = y * S + dy;                                                                                          \
    const bool yok = (rel) < nrows && (unsigned)yy < (unsigned)Hx;                                                      \
    const float* xrow = X + ((long long)(f * Hx + min(max(yy, 0), Hx - 1)) * Wx) * 64 + lc;                             \
    _Pragma("unroll") for (int i = 0; i < NW; ++i) {                                                                    \
      const int wr = lr + 16 * i;                                                                                       \
      const int xx = x0 * S - KS / 2 + wr;                                                                              \
      const f32x4 v_ = *reinterpret_cast<const f32x4*>(xrow + (long long)min(max(xx, 0), Wx - 1) * 64);                 \
      pw[i] = (yok && wr < WR && (unsigned)xx < (unsigned)Wx) ? v_ : zero4;                                             \
    }                                                                                                                   \
  }
  CWW_FETCH(0)
  for (int rb = 0; rb < nrows; rb += 32) {
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) *reinterpret_cast<f32x4*>(&Ys[(lr + 16 * h) * CW_P + lc]) = pa[h];
#pragma unroll
    for (int i = 0; i < NW; ++i)
      if (lr + 16 * i <= WR) *reinterpret_cast<f32x4*>(&Xw[(lr + 16 * i) * CW_P + lc]) = pw[i];
    __syncthreads();
    CWW_FETCH(rb + 32)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int m0 = 16 * kk + 8 * (lane >> 5);
      f32x8 a;
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = Ys[(m0 + j) * CW_P + wn + (lane & 31)];
      bf16x8 ah, al;
      cw_split8(a, ah, al);
#pragma unroll
      for (int t = 0; t < KS; ++t) {
        f32x8 b;
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = Xw[((m0 + j) * S + t) * CW_P + wk + (lane & 31)];
        bf16x8 bh, bl;
        cw_split8(b, bh, bl);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
        if (MODE != 2) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
        }
        if (MODE == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, acc[t], 0, 0, 0);
      }
    }
  }
#undef CWW_FETCH
  const int K = KS * KS * 64;
  float* out = partial + (long long)blockIdx.y * CA * K;
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wn + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      out[(long long)n * K + (ky * KS + t) * 64 + wk + (lane & 31)] = acc[t][r];
    }
}

// out[i] = sum_g partial[g][i] written through an index map:  OHWI [Cout][taps][Cin] -> torch OIHW [Cout][Cin][taps]
__global__ __launch_bounds__(256) void reduce_to_oihw_kernel(const float* __restrict__ partial, float* __restrict__ out, int G, int Cout,
                                                             int Cin, int taps) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int total = Cout * Cin * taps;
  if (i >= total) return;
  float a = 0.f;
  for (int g = 0; g < G; ++g) a += partial[(long long)g * total + i];
  const int ci = i % Cin, rem = i / Cin, tap = rem % taps, co = rem / taps;
  out[((long long)co * Cin + ci) * taps + tap] = a;
}

// backward-data kernel of a "same" convolution: wb[ci][ky][kx][co] = w_oihw[co][ci][ks-1-ky][ks-1-kx]
__global__ __launch_bounds__(256) void pack_conv_bwd_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin,
                                                            int ks) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Cout * Cin * ks * ks) return;
  const int co = i % Cout, rem = i / Cout, kx = rem % ks, rem2 = rem / ks, ky = rem2 % ks, ci = rem2 / ks;
  out[i] = w[(((long long)co * Cin + ci) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)];
}

// im2col of the NCHW image for conv0's weight gradient: P[m][k], k = (ci*ks + ky)*ks + kx (torch weight order), zero-padded
// to KP columns; m = (f, y, x) over the Ho x Wo outputs
__global__ __launch_bounds__(256) void im2col_first_kernel(const float* __restrict__ img, long long frame_stride, float* __restrict__ P,
                                                           long long M, int Cin, int Hin, int Win, int Ho, int Wo, int ks, int stride,
                                                           int KP) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * KP) return;
  const int k = (int)(i % KP);
  const long long m = i / KP;
  float v = 0.f;
  if (k < Cin * ks * ks) {
    const int kx = k % ks, ky = (k / ks) % ks, ci = k / (ks * ks);
    const int hw = Ho * Wo;
    const long long f = m / hw;
    const int rr = (int)(m - f * hw), y = rr / Wo, x = rr - y * Wo;
    const int yy = y * stride + ky - ks / 2, xx = x * stride + kx - ks / 2;
    if ((unsigned)yy < (unsigned)Hin && (unsigned)xx < (unsigned)Win) v = img[f * frame_stride + ((long long)ci * Hin + yy) * Win + xx];
  }
  P[i] = v;
}
__global__ __launch_bounds__(256) void crop_cols_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int KP, int K) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R * K) return;
  out[i] = in[(long long)(i / K) * KP + i % K];
}

// position embedding (utils.py:52-63): table = grid W^T + b is added to every frame, so
//   d_table[p][c] = sum_f d[f][p][c];   dW[c][j] = sum_p d_table[p][c] grid[p][j];   db[c] = sum_p d_table[p][c]
__global__ __launch_bounds__(256) void frame_sum_kernel(const float* __restrict__ d, float* __restrict__ dt, int F, long long per) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= per) return;
  float a = 0.f;
  for (int f = 0; f < F; ++f) a += d[f * per + i];
  dt[i] = a;
}
__global__ __launch_bounds__(256) void pos_dense_grad_kernel(const float* __restrict__ dt, const float* __restrict__ grid, float* __restrict__ dw,
                                                             float* __restrict__ db, int HW, int C) {
  __shared__ float red[5][256];
  const int c = blockIdx.x, t = threadIdx.x;
  float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int p = t; p < HW; p += 256) {
    const float g = dt[(long long)p * C + c];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] += g * grid[p * 4 + j];
    a[4] += g;
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) red[j][t] = a[j];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s)
#pragma unroll
      for (int j = 0; j < 5; ++j) red[j][t] += red[j][t + s];
    __syncthreads();
  }
  if (t < 4) dw[c * 4 + t] = red[t][0];
  if (t == 4) db[c] = red[4][0];
}

// weight gradient of a 64-input-channel "same" convolution (s = 1, A = dY) or of a transposed convolution (A = its input, X = dY
// on the finer grid), written in torch layout [CA][64][ks][ks]
int sf_conv_wgrad_ex(const float* A, int CA, int H, int W, const float* X, int Hx, int Wx, int s, int ks, long long rows,
                     float* out_oihw, float* partial, hipStream_t st) {
  SF_REQUIRE(CA % 64 == 0 && ks == 5 && rows > 0 && rows < (1LL << 31), "conv weight gradient: channel multiple of 64, 5x5 kernels");
  const int taps = ks * ks, nt = CA / 64;
  int splits = 768 / (ks * nt);
  if (splits < 1) splits = 1;
  if ((long long)splits * 64 > rows) splits = (int)((rows + 63) / 64);
  int rps = (int)((rows + splits - 1) / splits);
  rps = (rps + 31) & ~31;
  const dim3 grid(ks * nt, splits);
  const int mode = sf_get_precision();
  constexpr bool win_on = true;
#define CW_LAUNCH(KERN) hipLaunchKernelGGL(KERN, grid, dim3(256), 0, st, A, X, partial, rows, rps, H, W, CA, Hx, Wx)
  if (win_on && W % 32 == 0 && (s == 1 || s == 2)) {   // one LDS window per chunk instead of five shifted fetches
    if (s == 1) {
      if (mode == 0) CW_LAUNCH((conv_wgrad_win_kernel<0, 1>));
      else if (mode == 1) CW_LAUNCH((conv_wgrad_win_kernel<1, 1>));
      else CW_LAUNCH((conv_wgrad_win_kernel<2, 1>));
    } else {
      if (mode == 0) CW_LAUNCH((conv_wgrad_win_kernel<0, 2>));
      else if (mode == 1) CW_LAUNCH((conv_wgrad_win_kernel<1, 2>));
      else CW_LAUNCH((conv_wgrad_win_kernel<2, 2>));
    }
  } else if (mode == 0)
    hipLaunchKernelGGL(conv_wgrad_kernel<0>, grid, dim3(256), 0, st, A, X, partial, rows, rps, H, W, CA, s, Hx, Wx);
  else if (mode == 1)
    hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, dim3(256), 0, st, A, X, partial, rows, rps, H, W, CA, s, Hx, Wx);
  else
    hipLaunchKernelGGL(conv_wgrad_kernel<2>, grid, dim3(256), 0, st, A, X, partial, rows, rps, H, W, CA, s, Hx, Wx);
#undef CW_LAUNCH
  SF_CHECK_LAUNCH();
  const int total = CA * 64 * taps;
  hipLaunchKernelGGL(reduce_to_oihw_kernel, dim3((total + 255) / 256), dim3(256), 0, st, partial, out_oihw, splits, CA, 64, taps);
  SF_CHECK_LAUNCH();
  return 0;
}
size_t sf_conv_wgrad_partial_floats(int CA, int ks) { return (size_t)160 * CA * ks * ks * 64; }

// SoftPositionEmbed gradients from d [F][HW][C] (the table is added to every frame); dtab: scratch [HW*C]
int sf_pos_dense_grad_ex(const float* d, int F, int HW, int C, const float* grid, float* dw, float* db, float* dtab, hipStream_t st) {
  const long long per = (long long)HW * C;
  hipLaunchKernelGGL(frame_sum_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, d, dtab, F, per);
  SF_CHECK_LAUNCH();
  hipLaunchKernelGGL(pos_dense_grad_kernel, dim3(C), dim3(256), 0, st, dtab, grid, dw, db, HW, C);
  SF_CHECK_LAUNCH();
  return 0;
}

namespace {
struct FDims {
  int F, res, L, C, Hd, Co, stride;
  long long M;
};
struct FWs {
  float* wp[8];    // forward weights OHWI (layers >= 1)
  float* wb[8];    // backward-data weights (layers >= 1)
  float* wf[8];    // fragment-ordered copy of the weights in use (forward, then backward-data) for the 4-row-tile kernel (conv_rows4.hip)
  float *table, *a[8], *xn, *h1, *g1, *ga, *gb, *w1t, *w2t, *dtab, *dw0, *partial;
  size_t total;
};
FWs carve(const FDims& d, float* base) {
  FWs w;
  memset(&w, 0, sizeof(w));
  size_t off = 0;
  auto take = [&](size_t n) {
    float* p = base ? base + off : nullptr;
    off += (n + 63) & ~(size_t)63;
    return p;
  };
  const size_t M = d.M, C = d.C, wsz = C * 25 * C;
  for (int i = 1; i < d.L; ++i) {
    w.wp[i] = take(wsz);
    w.wb[i] = take(wsz);
    w.wf[i] = take((sf_conv_frag_bytes((int)C, (int)C, 5) + 3) / 4);
  }
  w.table = take((size_t)4096 * C);
  for (int i = 0; i < d.L; ++i) w.a[i] = take(M * C);
  w.xn = take(M * C);
  w.h1 = take(M * d.Hd);
  w.g1 = take(M * (d.Hd > 128 ? d.Hd : 128));
  w.ga = take(M * (C > d.Co ? C : d.Co));
  w.gb = take(M * C);
  w.w1t = take((size_t)d.Hd * C);
  w.w2t = take((size_t)d.Hd * d.Co);
  w.dtab = take((size_t)4096 * C);
  w.dw0 = take((size_t)C * 128);
  size_t pf = sf_grad_partial_floats(d.M, d.Hd, d.C);
  const size_t alt[] = {sf_grad_partial_floats(d.M, d.Co, d.Hd), sf_grad_partial_floats(d.M, C, 128), sf_conv_wgrad_partial_floats(C, 5)};
  for (size_t x : alt) pf = x > pf ? x : pf;
  w.partial = take(pf);
  w.total = off;
  return w;
}
int check(const sf_savi_features* m, FDims& d, int F) {
  SF_REQUIRE(m && F >= 1, "bad arguments");
  SF_REQUIRE(m->resolution == 64 || m->resolution == 128, "resolution must be 64 or 128 (savi.py:226,236)");
  SF_REQUIRE(m->layers >= 2 && m->layers <= 8 && m->channels == 64 && m->ks == 5, "training needs the reference encoder: 64 channels, 5x5");
  SF_REQUIRE(m->hidden % 64 == 0 && m->out_channels % 64 == 0 && m->hidden <= 1024, "MLP widths must be multiples of 64");
  d.F = F; d.res = m->resolution; d.L = m->layers; d.C = m->channels; d.Hd = m->hidden; d.Co = m->out_channels;
  d.stride = m->resolution == 128 ? 2 : 1;
  d.M = (long long)F * 4096;
  for (int i = 0; i < d.L; ++i) SF_REQUIRE(m->conv_w[i] && m->conv_b[i], "null conv parameter");
  SF_REQUIRE(m->pos_grid && m->pos_w && m->pos_b && m->ln_g && m->ln_b && m->fc1_w && m->fc1_b && m->fc2_w && m->fc2_b, "null parameter");
  return 0;
}
int gemm(const float* A, const float* W, const float* bias, float* C, long long M, int N, int K, int relu, hipStream_t st) {
  return sf_linear_ex(A, sf_rows(K), W, bias, nullptr, nullptr, 0.f, nullptr, sf_rows(N), 0, C, sf_rows(N), (int)M, N, K, relu, st);
}
// 5x5 64 -> 64 convolution on the 64 x 64 grid: the 4-row-tile kernel on a fragment-ordered copy of the OHWI weights (conv_rows4.hip; the same
// products in the same order as the halo kernel behind sf_conv2d_nhwc_f32, which stays the fallback for the arithmetic modes rows4 does not cover)
int conv5(const float* in, const float* w_ohwi, float* w_frag, const float* bias, const float* add, float* out, int F, int C, int relu, hipStream_t st) {
  if (w_frag && sf_get_precision() == 1 && sf_conv_frag_bytes(C, C, 5)) {
    SF_TRY(sf_pack_conv_frag_weights(w_ohwi, w_frag, C, C, 5, st));
    const int rc = sf_conv5x5_rows4_ex(in, w_frag, bias, add, out, F, 64, 64, C, C, 5, relu, st);
    if (rc != 1) return rc;
  }
  return sf_conv2d_nhwc_f32(in, w_ohwi, bias, add, out, F, 64, 64, C, C, 5, relu, st);
}
}  // namespace

extern "C" {

size_t sf_savi_features_train_workspace_bytes(const sf_savi_features* m, int F) {
  FDims d;
  if (check(m, d, F) != 0) return 0;
  return carve(d, nullptr).total * sizeof(float) + 256;
}

int sf_savi_features_train_fwd_f32(const sf_savi_features* m, const float* img, long long frame_stride, int F, float* out, void* ws,
                                   size_t ws_bytes, void* stream) {
  FDims d;
  SF_TRY(check(m, d, F));
  SF_REQUIRE(img && out && ws, "null pointer");
  const FWs w = carve(d, (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255));
  SF_REQUIRE(w.total * sizeof(float) + 256 <= ws_bytes, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int C = d.C, L = d.L;
  SF_TRY(sf_pos_embed_table_f32(m->pos_grid, m->pos_w, m->pos_b, w.table, 4096, C, st));
  for (int i = 1; i < L; ++i) SF_TRY(sf_pack_conv_weight_f32(m->conv_w[i], w.wp[i], C, C, 5, st));
  SF_TRY(sf_conv2d_nchw_in_f32(img, frame_stride, m->conv_w[0], m->conv_b[0], nullptr, w.a[0], F, 3, d.res, d.res, C, 5, d.stride, 1, st));
  for (int i = 1; i < L; ++i) {
    const bool last = i == L - 1;
    SF_TRY(conv5(w.a[i - 1], w.wp[i], w.wf[i], m->conv_b[i], last ? w.table : nullptr, w.a[i], F, C, last ? 0 : 1, st));
  }
  SF_TRY(sf_layernorm_ex(w.a[L - 1], sf_rows(C), m->ln_g, m->ln_b, w.xn, sf_rows(C), (int)d.M, C, 1e-5f, st));
  SF_TRY(gemm(w.xn, m->fc1_w, m->fc1_b, w.h1, d.M, d.Hd, C, 1, st));
  return gemm(w.h1, m->fc2_w, m->fc2_b, out, d.M, d.Co, d.Hd, 0, st);
}

int sf_savi_features_train_bwd_f32(const sf_savi_features* m, const float* img, long long frame_stride, const float* d_out,
                                   const sf_savi_features_grads* g, int F, void* ws, size_t ws_bytes, void* stream) {
  FDims d;
  SF_TRY(check(m, d, F));
  SF_REQUIRE(img && d_out && g && ws, "null pointer");
  const FWs w = carve(d, (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255));
  SF_REQUIRE(w.total * sizeof(float) + 256 <= ws_bytes, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int C = d.C, L = d.L, Hd = d.Hd, Co = d.Co;
  const long long M = d.M;
  // per-pixel MLP
  SF_TRY(sf_grad_weight_ex(d_out, w.h1, g->fc2_w, M, Co, Hd, w.partial, st));
  SF_TRY(sf_grad_bias_ex(d_out, g->fc2_b, M, Co, w.partial, st));
  SF_TRY(sf_transpose_ex(m->fc2_w, w.w2t, Co, Hd, st));   // [Co,Hd] -> [Hd,Co]
  SF_TRY(sf_linear_masked_ex(d_out, w.w2t, w.h1, 1.f, w.g1, M, Hd, Co, st));   // through fc2 and the ReLU (mask = its output)
  SF_TRY(sf_grad_weight_ex(w.g1, w.xn, g->fc1_w, M, Hd, C, w.partial, st));
  SF_TRY(sf_grad_bias_ex(w.g1, g->fc1_b, M, Hd, w.partial, st));
  SF_TRY(sf_transpose_ex(m->fc1_w, w.w1t, Hd, C, st));    // [Hd,C] -> [C,Hd]
  SF_TRY(gemm(w.g1, w.w1t, nullptr, w.ga, M, C, Hd, 0, st));   // d(xn)
  SF_TRY(sf_grad_ln_ex(w.a[L - 1], w.ga, g->ln_g, g->ln_b, M, C, 1e-5f, w.partial, st));
  SF_TRY(sf_ln_bwd_ex(w.a[L - 1], w.ga, m->ln_g, nullptr, w.gb, M, C, 1e-5f, st));   // gb = d(conv_last + table)
  // soft position embedding
  SF_TRY(sf_pos_dense_grad_ex(w.gb, F, 4096, C, m->pos_grid, g->pos_w, g->pos_b, w.dtab, st));
  // conv stack, last to first; cur = gradient w.r.t. the (pre-activation) output of conv i
  float* cur = w.gb;
  float* nxt = w.ga;
  for (int i = L - 1; i >= 1; --i) {
    SF_TRY(sf_grad_bias_ex(cur, g->conv_b[i], M, C, w.partial, st));
    SF_TRY(sf_conv_wgrad_ex(cur, C, 64, 64, w.a[i - 1], 64, 64, 1, 5, M, g->conv_w[i], w.partial, st));
    // data gradient: convolution with the flipped / transposed kernel, then the ReLU of the layer below
    hipLaunchKernelGGL(pack_conv_bwd_kernel, dim3((C * C * 25 + 255) / 256), dim3(256), 0, st, m->conv_w[i], w.wb[i], C, C, 5);
    SF_CHECK_LAUNCH();
    SF_TRY(conv5(cur, w.wb[i], w.wf[i], nullptr, w.a[i - 1], nxt, F, C, 2, st));   // relu = 2: gated by a[i-1] > 0
    float* t = cur;
    cur = nxt;
    nxt = t;
  }
  // conv0: bias + weight gradient against the image patches
  SF_TRY(sf_grad_bias_ex(cur, g->conv_b[0], M, C, w.partial, st));
  {
    const long long n = M * 128;
    hipLaunchKernelGGL(im2col_first_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, img, frame_stride, w.g1, M, 3, d.res,
                       d.res, 64, 64, 5, d.stride, 128);
    SF_CHECK_LAUNCH();
    SF_TRY(sf_grad_weight_ex(cur, w.g1, w.dw0, M, C, 128, w.partial, st));
    hipLaunchKernelGGL(crop_cols_kernel, dim3((C * 75 + 255) / 256), dim3(256), 0, st, w.dw0, g->conv_w[0], C, 128, 75);
    SF_CHECK_LAUNCH();
  }
  return 0;
}

}  // extern "C"
