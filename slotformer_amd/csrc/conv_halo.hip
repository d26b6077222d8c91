// 5x5 / 64->64 / 64-pixel-wide NHWC convolution with the input halo resident in LDS (split-bf16 MFMA).
//
// The generic implicit-GEMM path (gemm.hip, im2col loader) re-fetches every input pixel once per tap
// (25x) through L2 and is fill-bound at ~140 us per layer.  Here a workgroup owns 2 output rows
// (128 pixels x 64 channels): the 6 x 68 input halo is loaded ONCE, split into bf16 hi/lo planes in
// LDS, and the A operand of tap (ky,kx) is read straight from the halo at a shifted address -- no
// per-tap staging of activations at all.  Only the weights (16 KB per tap) stream through a double
// buffer.  Fill bytes per tile: 104 KB halo + 400 KB weights instead of 1.2 MB.
//
//   8 waves: (row 0/1) x (pixel block 0/1) x (cin half 0/1), two 32x32 accumulators (both cout blocks) each;
//   per tap and wave: 2 k16-steps x 2 cout blocks x (hi*hi + hi*lo + lo*hi) v_mfma_f32_32x32x16_bf16.
//   LDS: halo planes 2 x 6*68*72 bf16 (pixel stride 144 B = 9 16-B slots: conflict-free b128)
//        + weight planes 2 buffers x 2 x 64*72 bf16  = 154 KB.
// Reference call site: the encoder convs i > 0, savi.py:231-239 (+ SoftPositionEmbed add, utils.py:60-63).
#include "sf_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int CH = 64, KS = 5, PAD = 2, TW = 64, TR = 2;     // channels, taps, tile width / rows
constexpr int HR = TR + KS - 1, HWD = TW + KS - 1;           // halo 6 x 68 pixels
constexpr int PS = CH + 8;                                   // bf16 elements per pixel / weight row (144 B)
constexpr int HALO = HR * HWD * PS;                          // elements per halo plane
constexpr int WBUF = CH * PS;                                // elements per weight plane buffer
constexpr int NT = 512;
constexpr size_t LDS_BYTES = (size_t)(2 * HALO + 4 * WBUF) * sizeof(__bf16);
}  // namespace

template <bool BF1>   // BF1: single-pass bf16 (precision mode 2): only the hi*hi product, no lo-plane reads
__global__ __launch_bounds__(NT) void conv5x5_halo_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                          const float* __restrict__ bias, const float* __restrict__ add,
                                                          float* __restrict__ out, int H, int relu) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  __bf16* Hh = lds;
  __bf16* Hl = Hh + HALO;
  __bf16* Wh = Hl + HALO;        // [2][64][PS]
  __bf16* Wl = Wh + 2 * WBUF;    // [2][64][PS]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // XCD-aware tile order (same reasoning as gemm.hip): consecutive row pairs of a frame share halo rows in L2
  int bid = blockIdx.x;
  {
    const int nb = gridDim.x;
    if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  }
  const int tiles_per_frame = H / TR;
  const int f = bid / tiles_per_frame, y0 = (bid - f * tiles_per_frame) * TR;
  const float* inf = in + (long long)f * H * TW * CH;

  auto split_store = [&](__bf16* hp, __bf16* lp, int off, f32x4 v) {
    const bf16x4 hi = __builtin_convertvector(v, bf16x4);
    const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
    *(bf16x4*)(hp + off) = hi;
    *(bf16x4*)(lp + off) = lo;
  };

  // ---- weights of tap 0 (prefetch) -------------------------------------------------------------
  // per tap: 64 cout rows x 64 cin f32 = 1024 float4 -> 2 per thread; w layout [cout][tap][cin]
  const int wc4 = t & 15, wr0 = t >> 4;  // float4 column, cout row (and +32)
  // 5-slot register ring: the weights of tap t+5 are requested while tap t is computed, so every load has
  // four taps (~1.5 us) to land before it is split into LDS -- slot index = kx is static (ky loop outside).
  f32x4 wreg[KS][2];
  auto load_w = [&](int tap, f32x4(&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) r[i] = *(const f32x4*)(w + ((long long)(wr0 + 32 * i) * (KS * KS) + tap) * CH + 4 * wc4);
  };
  auto store_w = [&](int buf, const f32x4(&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) split_store(Wh + buf * WBUF, Wl + buf * WBUF, (wr0 + 32 * i) * PS + 4 * wc4, r[i]);
  };
#pragma unroll
  for (int q = 0; q < KS; ++q) load_w(q, wreg[q]);

  // ---- halo fill: 6 x 68 pixels x 16 float4, zero outside the image ------------------------------
  {
    constexpr int TOTAL = HR * HWD * (CH / 4);  // 6528 float4
    constexpr int IT = (TOTAL + NT - 1) / NT;   // 13
    f32x4 hv[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int idx = min(t + NT * i, TOTAL - 1);
      const int pix = idx >> 4, c4 = idx & 15;
      const int hy = pix / HWD, hx = pix - hy * HWD;
      const int gy = y0 - PAD + hy, gx = hx - PAD;
      const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)TW;
      const int yc = min(max(gy, 0), H - 1), xc = min(max(gx, 0), TW - 1);
      const f32x4 v = *(const f32x4*)(inf + ((long long)yc * TW + xc) * CH + 4 * c4);
      hv[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int idx = t + NT * i;
      if (idx < TOTAL) split_store(Hh, Hl, (idx >> 4) * PS + 4 * (idx & 15), hv[i]);
    }
  }
  store_w(0, wreg[0]);
  __syncthreads();

  // ---- 25 taps ---------------------------------------------------------------------------------------
  // wave = (output row, 32-pixel block, half of the input channels): it accumulates BOTH 32-cout blocks over
  // its 32 input channels, so one A fragment feeds two MFMA groups -- 12 ds_read_b128 per 12 MFMAs per tap
  // instead of 16 (the loop is LDS-read-bound).  The two k-halves are added through LDS after the last tap.
  // acc[0] = the cout block this wave finishes (block kh), acc[1] = the one it hands to its partner: the runtime
  // choice sits in the weight ADDRESS, never in a register index (that turns every MFMA into select + hazard nops).
  const int row = wave >> 2, pxb = ((wave >> 1) & 1) * 32, kh = wave & 1;
  f32x16 acc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const int a_lane = (lane & 31) * PS + 8 * (lane >> 5) + 32 * kh;   // pixel (or cout) row + k half, + this wave's cin half
  for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
      const int tap = ky * KS + kx;
      const int buf = tap & 1;
      // slot kx held tap's weights; they went to LDS during the previous tap -> refill it with tap + 5
      if (tap + KS < KS * KS) load_w(tap + KS, wreg[kx]);
      const __bf16* ah = Hh + ((row + ky) * HWD + pxb + kx) * PS + a_lane;
      const __bf16* al = Hl + ((row + ky) * HWD + pxb + kx) * PS + a_lane;
      const __bf16* bh = Wh + buf * WBUF + a_lane;
      const __bf16* bl = Wl + buf * WBUF + a_lane;
      const int cq[2] = {kh * 32 * PS, (1 - kh) * 32 * PS};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 xh = *(const bf16x8*)(ah + ks * 16);
        if constexpr (BF1) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const bf16x8 yh = *(const bf16x8*)(bh + cq[q] + ks * 16);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh, acc[q], 0, 0, 0);
          }
        } else {
          const bf16x8 xl = *(const bf16x8*)(al + ks * 16);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const bf16x8 yh = *(const bf16x8*)(bh + cq[q] + ks * 16), yl = *(const bf16x8*)(bl + cq[q] + ks * 16);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh, acc[q], 0, 0, 0);
          }
        }
      }
      // weights of tap+1 (slot (kx+1)%5, requested four taps ago) -> the buffer last read one barrier ago
      if (tap + 1 < KS * KS) store_w(buf ^ 1, wreg[(kx + 1) % KS]);
      __syncthreads();
    }
  }

  // ---- combine the two k-halves: wave kh keeps cout block kh and receives that block from its partner ----------
  float* X = (float*)lds;   // [8 waves][16][64] f32 = 32 KB over the (now dead) halo
#pragma unroll
  for (int r = 0; r < 16; ++r) X[(wave * 16 + r) * 64 + lane] = acc[1][r];
  __syncthreads();
  f32x16 fin = acc[0];
#pragma unroll
  for (int r = 0; r < 16; ++r) fin[r] += X[((wave ^ 1) * 16 + r) * 64 + lane];

  // ---- epilogue: bias, ReLU, optional per-position table, NHWC store ------------------------------------
  const int co = kh * 32 + (lane & 31);
  const float bv = bias ? bias[co] : 0.f;
  // relu == 2 (training, backward-data pass): `add` is the forward activation of the layer below, laid out like `out`, and
  // acts as its ReLU mask -- out = add > 0 ? conv : 0
  const bool mask = relu == 2;
  const float lo = relu == 1 ? 0.f : -INFINITY;
  const int y = y0 + row;
  float av[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int px = pxb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    av[r] = add ? add[((long long)(mask ? f * H + y : y) * TW + px) * CH + co] : 0.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int px = pxb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const float t = fmaxf(fin[r] + bv, lo);
    out[(((long long)f * H + y) * TW + px) * CH + co] = mask ? (av[r] > 0.f ? t : 0.f) : t + av[r];
  }
}

// Returns 1 when the specialised kernel does not apply (caller uses the implicit-GEMM path).
int sf_conv5x5_halo_ex(const float* in, const float* w_packed, const float* bias, const float* add, float* out, int F,
                       int H, int W, int Cin, int Cout, int ks, int relu, hipStream_t st) {
  if (W != TW || Cin != CH || Cout != CH || ks != KS || (H % TR) != 0 || F <= 0) return 1;
  SF_TRY(sf_ensure_dyn_lds((const void*)conv5x5_halo_kernel<false>, LDS_BYTES));
  SF_TRY(sf_ensure_dyn_lds((const void*)conv5x5_halo_kernel<true>, LDS_BYTES));
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  sf_prof_begin(SF_K_CONV_NHWC, st, 2.0 * (double)F * H * W * Cout * ks * ks * Cin);
  if (sf_get_precision() == 2)
    hipLaunchKernelGGL(conv5x5_halo_kernel<true>, dim3(F * (H / TR)), dim3(NT), LDS_BYTES, st, in, w_packed, bias, add, out, H,
                       relu);
  else
    hipLaunchKernelGGL(conv5x5_halo_kernel<false>, dim3(F * (H / TR)), dim3(NT), LDS_BYTES, st, in, w_packed, bias, add, out, H,
                       relu);
  sf_prof_end(SF_K_CONV_NHWC, st);
  SF_CHECK_LAUNCH();
  return 0;
}
