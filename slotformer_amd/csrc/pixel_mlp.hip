// Fused per-pixel chain of the SAVi encoder for one 128-pixel tile per workgroup (split-bf16 MFMA):
//
//   x[128,64] -> LN(64) -> fc1 (64->128) + ReLU -> fc2 (128->128) -> LN(128) -> [Wk;Wv] (128->256) -> kv[128,256]
//
// (encoder_out_layer, savi.py:245-250,372-375, then SlotAttention.norm_inputs / project_k / project_v,
// savi.py:66-70).  The three GEMMs of the unfused path move 436 MB per time step (B=32) through memory; here
// the intermediates never leave LDS (they are kept as bf16 hi/lo planes, the LN(128) input as an f32 tile) and
// only x (33.5 MB) is read and k|v (134 MB) written.  Weights (224 KB f32) are re-streamed per tile from L2 with
// the next stage's weights prefetched into registers during the current stage's MFMAs.
//
// 8 waves: wave w owns rows 32*(w>>1) .. +31 and columns 64*(w&1) .. +63 of every 128-wide output (2 accumulators).
#include "sf_internal.h"
#include "slot_chain.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// the per-pixel chain in its pixel-stationary form (pixel_feat_tok_kernel below): process default; sf_set_pixel_tok(0): the tile kernels
static int g_pixel_tok = 1;
extern "C" int sf_get_pixel_tok(void) { return g_pixel_tok; }
extern "C" int sf_set_pixel_tok(int on) {
  g_pixel_tok = on ? 1 : 0;
  return 0;
}

namespace {
constexpr int PM_NT = 512, PM_ROWS = 128, PM_C0 = 64, PM_C1 = 128, PM_ND = 256;
constexpr int PM_LB0 = PM_C0 + 8, PM_LB1 = PM_C1 + 8;                 // bf16 row strides (odd # of 16-B slots)
constexpr int PM_PLANE = PM_ROWS * PM_LB1;                             // elements of one (largest) plane
constexpr int PM_H2S = PM_C1 + 4;                                      // f32 row stride of the LN(128) input tile
constexpr size_t PM_LDS = (size_t)4 * PM_PLANE * sizeof(__bf16) + (2 * PM_ROWS + 4 * PM_C1) * sizeof(float);
static_assert((size_t)PM_ROWS * PM_H2S * 4 <= (size_t)2 * PM_PLANE * 2, "f32 tile must fit in the B planes");
}  // namespace

// FEAT: stop after the LayerNorm of the Slot-Attention inputs and write those 128 features per pixel (f32) to `kv` instead of
// k|v = [Wk;Wv] LN(h2) -- the folded Slot Attention (engine.hip) works on the normalised features directly.
template <bool FEAT>
__global__ __launch_bounds__(PM_NT) void pixel_mlp_kv_kernel(
    const float* __restrict__ x, const float* __restrict__ ln0_g, const float* __restrict__ ln0_b,
    const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
    const float* __restrict__ b2, const float* __restrict__ ln1_g, const float* __restrict__ ln1_b,
    const float* __restrict__ wkv, float* __restrict__ kv, int M, float eps) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  __bf16* Ah = lds;               // A planes (activations)
  __bf16* Al = Ah + PM_PLANE;
  __bf16* Bh = Al + PM_PLANE;     // B planes (weights); also the f32 h2 tile
  __bf16* Bl = Bh + PM_PLANE;
  float* stats = (float*)(Bl + PM_PLANE);  // [2][128]
  float* PV = stats + 2 * PM_ROWS;         // b1 | b2 | ln1 gamma | ln1 beta, 128 floats each
  float* H2 = (float*)Bh;                  // [128][PM_H2S] f32 (aliases both B planes)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int m0 = blockIdx.x * PM_ROWS;

  auto split4 = [&](f32x4 v, bf16x4& hi, bf16x4& lo) {
    hi = __builtin_convertvector(v, bf16x4);
    lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
  };

  // small parameter vectors first: vmcnt retires in order, so a request issued after the weight tiles would wait for
  // all of them (measured on the rollout kernels: 2-4 us per late load)
  const int c4 = t & 15, r0 = t >> 4;  // rows r0 + 32*i
  const f32x4 g0 = *(const f32x4*)(ln0_g + 4 * c4), be0 = *(const f32x4*)(ln0_b + 4 * c4);
  float pvv = 0.f;
  {
    const float* src = (t < 128) ? b1 : (t < 256) ? b2 : (t < 384) ? ln1_g : ln1_b;
    pvv = src[t & 127];
  }
  // ---- P0: x tile and W1 (16 float4 per 64-wide row); LN(64) statistics straight from the registers --------
  f32x4 xr[4], wr1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = min(m0 + r0 + 32 * i, M - 1);
    xr[i] = *(const f32x4*)(x + (long long)row * PM_C0 + 4 * c4);
    wr1[i] = *(const f32x4*)(w1 + (long long)(r0 + 32 * i) * PM_C0 + 4 * c4);
  }
  // W2: 128 rows x 32 float4 -> 8 per thread (rows q0 + 16*i); requested now, consumed after GEMM1
  const int d4 = t & 31, q0 = t >> 5;
  f32x4 wr2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) wr2[i] = *(const f32x4*)(w2 + (long long)(q0 + 16 * i) * PM_C1 + 4 * d4);
  PV[t] = pvv;
  {
    const f32x4 g = g0, be = be0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float s = sf_sum16((xr[i][0] + xr[i][1]) + (xr[i][2] + xr[i][3]));  // the 16 lanes holding this row
      const float mean = s * (1.0f / PM_C0);
      const f32x4 dv = xr[i] - mean;
      const float vs = sf_sum16((dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]));
      const float rstd = 1.0f / sqrtf(vs * (1.0f / PM_C0) + eps);
      bf16x4 hi, lo;
      split4(dv * rstd * g + be, hi, lo);
      const int off = (r0 + 32 * i) * PM_LB0 + 4 * c4;
      *(bf16x4*)(Ah + off) = hi;
      *(bf16x4*)(Al + off) = lo;
      split4(wr1[i], hi, lo);
      *(bf16x4*)(Bh + off) = hi;
      *(bf16x4*)(Bl + off) = lo;
    }
  }
  __syncthreads();

  const int rb = wave >> 1, cb0 = (wave & 1) * 2;  // row block, first of two column blocks
  auto gemm = [&](int LB, int nk16, f32x16 (&acc)[2]) {
    const int ao = (rb * 32 + (lane & 31)) * LB + 8 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int ks = 0; ks < nk16; ++ks) {
      const bf16x8 xh = *(const bf16x8*)(Ah + ao + ks * 16), xl = *(const bf16x8*)(Al + ao + ks * 16);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int bo = ((cb0 + j) * 32 + (lane & 31)) * LB + 8 * (lane >> 5) + ks * 16;
        const bf16x8 yh = *(const bf16x8*)(Bh + bo), yl = *(const bf16x8*)(Bl + bo);
        // transposed (weights as the A operand): acc[j][4 g + q] = out[pixel = lane & 31][column 8 g + 4 (lane >> 5) + q]
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh, xl, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yl, xh, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh, xh, acc[j], 0, 0, 0);
      }
    }
  };
  auto store_w128 = [&](const f32x4 (&wr)[8]) {  // 128 x 128 f32 weights in registers -> B planes
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      bf16x4 hi, lo;
      split4(wr[i], hi, lo);
      const int off = (q0 + 16 * i) * PM_LB1 + 4 * d4;
      *(bf16x4*)(Bh + off) = hi;
      *(bf16x4*)(Bl + off) = lo;
    }
  };

  // ---- P1: h1 = relu(LN(x) W1^T + b1) -> A planes (K = 128 layout) ----------------------------------------
  f32x16 acc[2];
  gemm(PM_LB0, PM_C0 / 16, acc);
  __syncthreads();  // every wave is done with the K=64 planes
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = rb * 32 + (lane & 31);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = (cb0 + j) * 32 + 8 * g + 4 * (lane >> 5);
      const f32x4 bv = *(const f32x4*)(PV + n);
      const f32x4 v = {fmaxf(acc[j][4 * g] + bv[0], 0.f), fmaxf(acc[j][4 * g + 1] + bv[1], 0.f), fmaxf(acc[j][4 * g + 2] + bv[2], 0.f),
                       fmaxf(acc[j][4 * g + 3] + bv[3], 0.f)};
      bf16x4 hi, lo;
      split4(v, hi, lo);
      *(bf16x4*)(Ah + row * PM_LB1 + n) = hi;
      *(bf16x4*)(Al + row * PM_LB1 + n) = lo;
    }
  }
  store_w128(wr2);
  f32x4 wr3[8];  // first half of [Wk;Wv] (rows 0..127), consumed after the LN(128)
#pragma unroll
  for (int i = 0; i < 8; ++i) wr3[i] = FEAT ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(wkv + (long long)(q0 + 16 * i) * PM_C1 + 4 * d4);
  __syncthreads();

  // ---- P2: h2 = h1 W2^T + b2 -> f32 tile (over the B planes) ------------------------------------------------
  gemm(PM_LB1, PM_C1 / 16, acc);
  __syncthreads();  // W2 planes are dead
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = rb * 32 + (lane & 31);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = (cb0 + j) * 32 + 8 * g + 4 * (lane >> 5);
      const f32x4 bv = *(const f32x4*)(PV + PM_C1 + n);
      *(f32x4*)(H2 + row * PM_H2S + n) = f32x4{acc[j][4 * g] + bv[0], acc[j][4 * g + 1] + bv[1], acc[j][4 * g + 2] + bv[2],
                                             acc[j][4 * g + 3] + bv[3]};
    }
  }
  __syncthreads();

  // ---- P3: LN(128) of every row (4 threads per row, 32 channels each) -> A planes ---------------------------
  {
    const int row = t >> 2, part = t & 3;
    f32x4 hv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) hv[i] = *(const f32x4*)(H2 + row * PM_H2S + part * 32 + 4 * i);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (hv[i][0] + hv[i][1]) + (hv[i][2] + hv[i][3]);
    s += sf_dpp<0xB1>(s);   // the 4 lanes holding this row
    s += sf_dpp<0x4E>(s);
    const float mean = s * (1.0f / PM_C1);
    float vs = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 dv = hv[i] - mean;
      vs += (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]);
    }
    vs += sf_dpp<0xB1>(vs);
    vs += sf_dpp<0x4E>(vs);
    const float rstd = 1.0f / sqrtf(vs * (1.0f / PM_C1) + eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = part * 32 + 4 * i;
      const f32x4 g = *(const f32x4*)(PV + 2 * PM_C1 + c), be = *(const f32x4*)(PV + 3 * PM_C1 + c);
      if constexpr (FEAT) {
        if (m0 + row < M) *(f32x4*)(kv + (long long)(m0 + row) * PM_C1 + c) = (hv[i] - mean) * rstd * g + be;
      } else {
        bf16x4 hi, lo;
        split4((hv[i] - mean) * rstd * g + be, hi, lo);
        *(bf16x4*)(Ah + row * PM_LB1 + c) = hi;
        *(bf16x4*)(Al + row * PM_LB1 + c) = lo;
      }
    }
  }
  if constexpr (FEAT) return;
  __syncthreads();  // everyone has read its part of the f32 tile: the B planes may be overwritten

  // ---- P4: k|v = LN(h2) [Wk;Wv]^T, two 128-column halves ---------------------------------------------------
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    store_w128(wr3);
    if (half == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) wr3[i] = *(const f32x4*)(wkv + (long long)(PM_C1 + q0 + 16 * i) * PM_C1 + 4 * d4);
    }
    __syncthreads();
    gemm(PM_LB1, PM_C1 / 16, acc);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = m0 + rb * 32 + (lane & 31);
      if (row < M) {
        float* dst = kv + (long long)row * PM_ND + half * PM_C1 + (cb0 + j) * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(f32x4*)(dst + 8 * g) = f32x4{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
      }
    }
    __syncthreads();  // before the second half overwrites the weight planes
  }
}

// Returns 1 when the fused kernel does not apply (caller runs the three GEMMs).
int sf_pixel_mlp_kv_ex(const float* x, const float* ln0_g, const float* ln0_b, const float* w1, const float* b1,
                       const float* w2, const float* b2, const float* ln1_g, const float* ln1_b, const float* wkv,
                       float* kv, int M, int C0, int C1, int ND, float eps, hipStream_t st) {
  if (C0 != PM_C0 || C1 != PM_C1 || ND != PM_ND || M <= 0 || !b1 || !b2) return 1;
  SF_TRY(sf_ensure_dyn_lds((const void*)pixel_mlp_kv_kernel<false>, (size_t)(PM_LDS)));
  static_assert(PM_LDS <= 160 * 1024, "LDS budget");
  sf_prof_begin(SF_K_LINEAR, st, 2.0 * M * (double)(PM_C0 * PM_C1 + PM_C1 * PM_C1 + PM_C1 * PM_ND));
  hipLaunchKernelGGL(pixel_mlp_kv_kernel<false>, dim3((M + PM_ROWS - 1) / PM_ROWS), dim3(PM_NT), PM_LDS, st, x, ln0_g, ln0_b,
                     w1, b1, w2, b2, ln1_g, ln1_b, wkv, kv, M, eps);
  sf_prof_end(SF_K_LINEAR, st);
  SF_CHECK_LAUNCH();
  return 0;
}

bool sf_pixel_mlp_feat_ok(int C0, int C1) { return C0 == PM_C0 && C1 == PM_C1; }
template <bool PLANES>
static int sf_pixel_feat_tok_launch(const float* x, const float* ln0_g, const float* ln0_b, const float* w1, const float* b1, const float* w2, const float* b2,
                                    const float* ln1_g, const float* ln1_b, void* out, int M, float eps, hipStream_t st);

static int sf_pixel_feat_stream_launch(const float* x, const float* ln0_g, const float* ln0_b, const float* w1, const float* b1, const float* w2,
                                       const float* b2, const float* ln1_g, const float* ln1_b, float* feat, int M, float eps, hipStream_t st,
                                       int tile = 0);

// encoder_out_layer + norm_inputs only: feat [M][128] = LN(fc2(relu(fc1(LN(x)))))
int sf_pixel_mlp_feat_ex(const float* x, const float* ln0_g, const float* ln0_b, const float* w1, const float* b1, const float* w2,
                         const float* b2, const float* ln1_g, const float* ln1_b, float* feat, int M, float eps, hipStream_t st) {
  if (M <= 0) return 0;
  // weights resident in registers, PF_TPW tiles per workgroup (pixel_feat_stream_kernel below)
  constexpr int stream = 1;
  if (sf_get_pixel_tok()) return sf_pixel_feat_tok_launch<false>(x, ln0_g, ln0_b, w1, b1, w2, b2, ln1_g, ln1_b, feat, M, eps, st);
  if (stream) return sf_pixel_feat_stream_launch(x, ln0_g, ln0_b, w1, b1, w2, b2, ln1_g, ln1_b, feat, M, eps, st);
  SF_TRY(sf_ensure_dyn_lds((const void*)pixel_mlp_kv_kernel<true>, (size_t)(PM_LDS)));
  sf_prof_begin(SF_K_LINEAR, st, 2.0 * M * (double)(PM_C0 * PM_C1 + PM_C1 * PM_C1));
  hipLaunchKernelGGL(pixel_mlp_kv_kernel<true>, dim3((M + PM_ROWS - 1) / PM_ROWS), dim3(PM_NT), PM_LDS, st, x, ln0_g, ln0_b, w1, b1, w2,
                     b2, ln1_g, ln1_b, nullptr, feat, M, eps);
  sf_prof_end(SF_K_LINEAR, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// ================================================================================================
// The per-pixel chain at STEVE's width (steve_physion_params.py: encoder_out_layer 64 -> 192 -> 192, Slot-Attention inputs 192), for the
// FOLDED Slot Attention (engine.hip): x [M][64] -> LN(64) -> fc1 + ReLU -> fc2 -> LN(192) -> features [M][192] (f32).  The three generic
// GEMM launches this replaces move 317 MB per time step of 16 frames (244 us on the encode partition); here a workgroup holds 128
// pixels as split-bf16 planes in LDS and streams the two weight matrices (packed once in fragment order, sf_pack_linear_weights)
// from memory as the MFMA A operand (stream_mfma.h): 16.8 MB in, 50 MB out.
//   fc1 / fc2: wave w < 6 owns column block w for all four token blocks (one set of fragment reads, four accumulators)
//   LN(192): a token's 192 values sit in six waves' accumulators -- two rounds of partial sums through LDS (mean, then squared deviations)
#include "stream_mfma.h"

namespace {
constexpr int PW_NT = 512, PW_ROWS = 128, PW_C0 = 64, PW_C1 = 192, PW_KP = PW_C1 + 8, PW_NB = PW_C1 / 32;
constexpr size_t PW_LDS = (size_t)2 * PW_ROWS * PW_KP * 2 + (size_t)(2 * PW_NB * PW_ROWS + 4 * PW_C1) * 4;
}  // namespace

__global__ __launch_bounds__(PW_NT) void pixel_mlp_feat192_kernel(const float* __restrict__ x, const float* __restrict__ ln0_g,
                                                                  const float* __restrict__ ln0_b, const uint4* __restrict__ w1p,
                                                                  const float* __restrict__ b1, const uint4* __restrict__ w2p,
                                                                  const float* __restrict__ b2, const float* __restrict__ ln1_g,
                                                                  const float* __restrict__ ln1_b, float* __restrict__ feat, int M, float eps) {
  extern __shared__ __attribute__((aligned(16))) __bf16 pw_lds[];
  __bf16* Ph = pw_lds;                       // [128][PW_KP]
  __bf16* Pl = Ph + PW_ROWS * PW_KP;
  float* S1 = (float*)(Pl + PW_ROWS * PW_KP);   // [6][128] partial sums
  float* S2 = S1 + PW_NB * PW_ROWS;             // [6][128] partial sums of squared deviations
  float* PV = S2 + PW_NB * PW_ROWS;             // b1 | b2 | ln1 gamma | ln1 beta
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int tok = lane & 31, kg = lane >> 5;
  const int m0 = blockIdx.x * PW_ROWS;
  const int c4 = t & 15, r0 = t >> 4;
  // small vectors and the x tile first (the loads retire in order), then the first weight fragments
  const f32x4 g0 = *(const f32x4*)(ln0_g + 4 * c4), be0 = *(const f32x4*)(ln0_b + 4 * c4);
  float pvv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = t + PW_NT * i;   // 4 * 192 = 768 values
    const float* src = j < PW_C1 ? b1 : j < 2 * PW_C1 ? b2 : j < 3 * PW_C1 ? ln1_g : ln1_b;
    pvv[i] = j < 4 * PW_C1 ? src[j % PW_C1] : 0.f;
  }
  f32x4 xr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) xr[i] = *(const f32x4*)(x + (long long)min(m0 + r0 + 32 * i, M - 1) * PW_C0 + 4 * c4);
  PsBuf s;
  constexpr int KS1 = PW_C0 / 16, KS2 = PW_C1 / 16;
  if (wave < PW_NB) ps_prime<PsChunk<KS1>::CH>(s, w1p, PW_NB, wave, 0, lane);
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (t + PW_NT * i < 4 * PW_C1) PV[t + PW_NT * i] = pvv[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float sm = sf_sum16((xr[i][0] + xr[i][1]) + (xr[i][2] + xr[i][3]));   // the 16 lanes holding this row
    const float mean = sm * (1.0f / PW_C0);
    const f32x4 dv = xr[i] - mean;
    const float vs = sf_sum16((dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]));
    const float rstd = 1.0f / sqrtf(vs * (1.0f / PW_C0) + eps);
    ps_split4(Ph, Pl, (r0 + 32 * i) * PW_KP + 4 * c4, dv * rstd * g0 + be0);
  }
  __syncthreads();
  f32x16 acc[4];
  // ---- h1 = relu(W1 LN(x) + b1) -> planes ----
  if (wave < PW_NB) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
    ps_blockN<KS1, PsChunk<KS2>::CH, 4>(acc, s, w1p, PW_NB, wave, 0, w2p, PW_NB, wave, 0, Ph, Pl, PW_KP, lane);
  }
  __syncthreads();   // every wave is done with the LN(x) planes
  if (wave < PW_NB) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = 32 * wave + 8 * g + 4 * kg;
        const f32x4 bv = *(const f32x4*)(PV + n);
        ps_split4(Ph, Pl, (32 * rb + tok) * PW_KP + n,
                  f32x4{fmaxf(acc[rb][4 * g] + bv[0], 0.f), fmaxf(acc[rb][4 * g + 1] + bv[1], 0.f), fmaxf(acc[rb][4 * g + 2] + bv[2], 0.f),
                        fmaxf(acc[rb][4 * g + 3] + bv[3], 0.f)});
      }
  }
  __syncthreads();
  // ---- h2 = W2 h1 + b2 (in the accumulators), LayerNorm over the 192 channels across the six waves, features out ----
  if (wave < PW_NB) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
    ps_blockN<KS2, 1, 4>(acc, s, w2p, PW_NB, wave, 0, nullptr, 1, 0, 0, Ph, Pl, PW_KP, lane);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      float sm = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *(const f32x4*)(PV + PW_C1 + 32 * wave + 8 * g + 4 * kg);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[rb][4 * g + q] += bv[q];
          sm += acc[rb][4 * g + q];
        }
      }
      sm += __shfl_xor(sm, 32, 64);
      if (kg == 0) S1[wave * PW_ROWS + 32 * rb + tok] = sm;
    }
  }
  __syncthreads();
  float mean[4];
  if (wave < PW_NB) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      float sm = 0.f;
#pragma unroll
      for (int j = 0; j < PW_NB; ++j) sm += S1[j * PW_ROWS + 32 * rb + tok];
      mean[rb] = sm * (1.0f / PW_C1);
      float vs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc[rb][r] - mean[rb];
        vs += d * d;
      }
      vs += __shfl_xor(vs, 32, 64);
      if (kg == 0) S2[wave * PW_ROWS + 32 * rb + tok] = vs;
    }
  }
  __syncthreads();
  if (wave < PW_NB) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      float vs = 0.f;
#pragma unroll
      for (int j = 0; j < PW_NB; ++j) vs += S2[j * PW_ROWS + 32 * rb + tok];
      const float rstd = 1.0f / sqrtf(vs * (1.0f / PW_C1) + eps);
      const int row = m0 + 32 * rb + tok;
      if (row < M) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = 32 * wave + 8 * g + 4 * kg;
          const f32x4 gm = *(const f32x4*)(PV + 2 * PW_C1 + n), bt = *(const f32x4*)(PV + 3 * PW_C1 + n);
          *(f32x4*)(feat + (long long)row * PW_C1 + n) =
              f32x4{(acc[rb][4 * g] - mean[rb]) * rstd * gm[0] + bt[0], (acc[rb][4 * g + 1] - mean[rb]) * rstd * gm[1] + bt[1],
                    (acc[rb][4 * g + 2] - mean[rb]) * rstd * gm[2] + bt[2], (acc[rb][4 * g + 3] - mean[rb]) * rstd * gm[3] + bt[3]};
        }
      }
    }
  }
}

// features of the folded Slot Attention at width 192 from the packed fc1 [192][64] / fc2 [192][192] copies
int sf_pixel_mlp_feat192_ex(const float* x, const float* ln0_g, const float* ln0_b, const void* w1p, const float* b1, const void* w2p,
                            const float* b2, const float* ln1_g, const float* ln1_b, float* feat, int M, float eps, hipStream_t st) {
  static_assert(PW_LDS <= 160 * 1024, "LDS budget");
  SF_TRY(sf_ensure_dyn_lds((const void*)pixel_mlp_feat192_kernel, PW_LDS));
  sf_prof_begin(SF_K_LINEAR, st, 2.0 * M * (double)(PW_C0 * PW_C1 + PW_C1 * PW_C1));
  hipLaunchKernelGGL(pixel_mlp_feat192_kernel, dim3((M + PW_ROWS - 1) / PW_ROWS), dim3(PW_NT), PW_LDS, st, x, ln0_g, ln0_b, (const uint4*)w1p, b1,
                     (const uint4*)w2p, b2, ln1_g, ln1_b, feat, M, eps);
  sf_prof_end(SF_K_LINEAR, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// ================================================================================================
// The same per-pixel chain up to the normalised Slot-Attention inputs (FEAT form) with the weights RESIDENT IN REGISTERS:
//   feat[M,128] = LN(128)( fc2( relu( fc1( LN(64)(x) ) ) ) )
// pixel_mlp_kv_kernel<true> re-streams and re-splits fc1 / fc2 (96 KB of f32) for every 128-pixel tile, passes them through LDS
// behind workgroup barriers and runs its phases one after the other (10 us per tile on its CU, 0.14 of the MFMA roof).  Here a
// workgroup walks PF_TPW consecutive tiles: wave = (32-column block, half of the tile's rows) keeps ITS fragments of fc1 (4
// k-steps) and fc2 (8 k-steps) as split-bf16 MFMA A operands in 96 registers for all of them -- read once from the f32 matrices, no
// packed copy needed -- the rows of the next tile are requested while the current one is in fc1, and its LayerNorm(64) planes are
// written while the current one is in fc2.  Same products in the same order, same LayerNorm reductions: the bits of
// pixel_mlp_kv_kernel<true>.
// TR = 64 (form 2): the same kernel on 64-pixel tiles with four waves (wave = column block, both row blocks), 55 KB of LDS and 256 threads: TWO
// workgroups per CU, each on its own chain of barrier-separated phases -- one computes while the other waits.  Same bits (every row's arithmetic
// is its own).
namespace {
constexpr int PF_PIX = 512;                                // pixels per workgroup (tiles per workgroup = PF_PIX / TR)
template <int TR> struct PfCfg {
  static constexpr int NT = 4 * TR;                                      // 512 threads for 128-pixel tiles, 256 for 64
  static constexpr int TPW = PF_PIX / TR;
  static constexpr size_t A0 = (size_t)2 * TR * PM_LB0 * 2;              // LN(64)(x) planes hi | lo: 36,864 B at TR = 128
  static constexpr size_t H1 = (size_t)2 * TR * PM_LB1 * 2;              // relu(fc1) planes hi | lo: 69,632 B; later the f32 fc2 tile [TR][PM_H2S]
  static constexpr size_t LDS = A0 + H1 + 4 * PM_C1 * sizeof(float);
  static_assert((size_t)TR * PM_H2S * 4 <= H1, "the f32 tile fits over the hidden planes");
};
}  // namespace

template <int TR, bool PLANES = false>
__global__ __launch_bounds__(PfCfg<TR>::NT) void pixel_feat_stream_kernel(
    const float* __restrict__ x, const float* __restrict__ ln0_g, const float* __restrict__ ln0_b, const float* __restrict__ w1,
    const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ ln1_g,
    const float* __restrict__ ln1_b, float* __restrict__ feat, int M, float eps, int PF_TPW) {
  constexpr int NT = PfCfg<TR>::NT, RPP = NT / 16;   // RPP: rows per LayerNorm(64) pass
  constexpr size_t PF_A0 = PfCfg<TR>::A0, PF_H1 = PfCfg<TR>::H1;
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  __bf16* Ah = lds;                                   // [TR][PM_LB0]
  __bf16* Al = Ah + TR * PM_LB0;
  __bf16* Hh = (__bf16*)((char*)lds + PF_A0);         // [TR][PM_LB1]
  __bf16* Hl = Hh + TR * PM_LB1;
  float* H2 = (float*)Hh;                             // [TR][PM_H2S] f32 (over the hidden planes)
  float* PV = (float*)((char*)lds + PF_A0 + PF_H1);   // b1 | b2 | ln1 gamma | ln1 beta
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int cb = wave & 3, rh = wave >> 2;            // column block of the 128-wide outputs, half of the tile's rows
  const int ntiles = (M + TR - 1) / TR;
  const int tile0 = blockIdx.x * PF_TPW;

  auto split4 = [&](f32x4 v, bf16x4& hi, bf16x4& lo) {
    hi = __builtin_convertvector(v, bf16x4);
    lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
  };
  const int c4 = t & 15, r0 = t >> 4;
  const f32x4 g0 = *(const f32x4*)(ln0_g + 4 * c4), be0 = *(const f32x4*)(ln0_b + 4 * c4);
#pragma unroll
  for (int u = t; u < 512; u += NT) {
    const float* src = (u < 128) ? b1 : (u < 256) ? b2 : (u < 384) ? ln1_g : ln1_b;
    PV[u] = src[u & 127];
  }
  f32x4 xr[4];
  auto request = [&](int tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = min(tile * TR + r0 + RPP * i, M - 1);
      xr[i] = *(const f32x4*)(x + (long long)row * PM_C0 + 4 * c4);
    }
  };
  request(tile0);
  // ---- this wave's weight fragments: element j of (matrix, k-step) = W[32 cb + (lane & 31)][16 ks + 8 (lane >> 5) + j] ----
  bf16x8 w1f[4][2], w2f[8][2];
  {
    const float* p1 = w1 + (long long)(cb * 32 + (lane & 31)) * PM_C0 + 8 * (lane >> 5);
    const float* p2 = w2 + (long long)(cb * 32 + (lane & 31)) * PM_C1 + 8 * (lane >> 5);
    auto frag = [&](const float* p, bf16x8& hi, bf16x8& lo) {
      const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
      bf16x4 ah, al, bh, bl;
      split4(a, ah, al);
      split4(b, bh, bl);
      hi = __builtin_shufflevector(ah, bh, 0, 1, 2, 3, 4, 5, 6, 7);
      lo = __builtin_shufflevector(al, bl, 0, 1, 2, 3, 4, 5, 6, 7);
    };
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) frag(p1 + 16 * ks, w1f[ks][0], w1f[ks][1]);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) frag(p2 + 16 * ks, w2f[ks][0], w2f[ks][1]);
  }
  // LayerNorm(64) of the rows in xr -> A planes (pixel_mlp_kv_kernel's arithmetic)
  auto ln64 = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float s = sf_sum16((xr[i][0] + xr[i][1]) + (xr[i][2] + xr[i][3]));
      const float mean = s * (1.0f / PM_C0);
      const f32x4 dv = xr[i] - mean;
      const float vs = sf_sum16((dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]));
      const float rstd = 1.0f / sqrtf(vs * (1.0f / PM_C0) + eps);
      bf16x4 hi, lo;
      split4(dv * rstd * g0 + be0, hi, lo);
      const int off = (r0 + RPP * i) * PM_LB0 + 4 * c4;
      *(bf16x4*)(Ah + off) = hi;
      *(bf16x4*)(Al + off) = lo;
    }
  };
  ln64();
  __syncthreads();   // A planes of the first tile, PV
  const int tok = lane & 31, kg = lane >> 5;
#pragma unroll 1
  for (int ti = 0; ti < PF_TPW; ++ti) {
    const int tile = tile0 + ti;
    if (tile >= ntiles) break;
    const bool more = ti + 1 < PF_TPW && tile + 1 < ntiles;
    if (more) request(tile + 1);   // lands under fc1
    // ---- fc1 + ReLU: rows 64 rh .. + 63 x columns 32 cb .. + 31 -> hidden planes ----
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ao = ((2 * rh + j) * 32 + tok) * PM_LB0 + 8 * kg + ks * 16;
        const bf16x8 xh = *(const bf16x8*)(Ah + ao), xl = *(const bf16x8*)(Al + ao);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[ks][0], xl, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[ks][1], xh, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[ks][0], xh, acc[j], 0, 0, 0);
      }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (2 * rh + j) * 32 + tok;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = cb * 32 + 8 * g + 4 * kg;
        const f32x4 bv = *(const f32x4*)(PV + n);
        const f32x4 v = {fmaxf(acc[j][4 * g] + bv[0], 0.f), fmaxf(acc[j][4 * g + 1] + bv[1], 0.f), fmaxf(acc[j][4 * g + 2] + bv[2], 0.f),
                         fmaxf(acc[j][4 * g + 3] + bv[3], 0.f)};
        bf16x4 hi, lo;
        split4(v, hi, lo);
        *(bf16x4*)(Hh + row * PM_LB1 + n) = hi;
        *(bf16x4*)(Hl + row * PM_LB1 + n) = lo;
      }
    }
    __syncthreads();   // hidden planes complete; every wave is done with the A planes
    if (more) ln64();  // the next tile's A planes, under fc2
    // ---- fc2 ----
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ao = ((2 * rh + j) * 32 + tok) * PM_LB1 + 8 * kg + ks * 16;
        const bf16x8 xh = *(const bf16x8*)(Hh + ao), xl = *(const bf16x8*)(Hl + ao);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[ks][0], xl, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[ks][1], xh, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[ks][0], xh, acc[j], 0, 0, 0);
      }
    __syncthreads();   // every wave is done with the hidden planes: the f32 tile takes their place
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (2 * rh + j) * 32 + tok;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = cb * 32 + 8 * g + 4 * kg;
        const f32x4 bv = *(const f32x4*)(PV + PM_C1 + n);
        *(f32x4*)(H2 + row * PM_H2S + n) = f32x4{acc[j][4 * g] + bv[0], acc[j][4 * g + 1] + bv[1], acc[j][4 * g + 2] + bv[2],
                                               acc[j][4 * g + 3] + bv[3]};
      }
    }
    __syncthreads();
    // ---- LayerNorm(128) of every row (4 threads per row, 32 channels each) -> feat ----
    {
      const int row = t >> 2, part = t & 3;
      f32x4 hv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) hv[i] = *(const f32x4*)(H2 + row * PM_H2S + part * 32 + 4 * i);
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s += (hv[i][0] + hv[i][1]) + (hv[i][2] + hv[i][3]);
      s += sf_dpp<0xB1>(s);
      s += sf_dpp<0x4E>(s);
      const float mean = s * (1.0f / PM_C1);
      float vs = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4 dv = hv[i] - mean;
        vs += (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]);
      }
      vs += sf_dpp<0xB1>(vs);
      vs += sf_dpp<0x4E>(vs);
      const float rstd = 1.0f / sqrtf(vs * (1.0f / PM_C1) + eps);
      const int grow = tile * TR + row;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = part * 32 + 4 * i;
        const f32x4 g = *(const f32x4*)(PV + 2 * PM_C1 + c), be = *(const f32x4*)(PV + 3 * PM_C1 + c);
        const f32x4 y = (hv[i] - mean) * rstd * g + be;
        if constexpr (PLANES) {
          // rows of 512 B: bf16 hi of the 128 channels | bf16 lo (what slot_chain.hip streams: no split inside its pixel loop)
          bf16x4 hi, lo;
          split4(y, hi, lo);
          if (grow < M) {
            __bf16* pr = (__bf16*)feat + (long long)grow * (2 * PM_C1);
            *(bf16x4*)(pr + c) = hi;
            *(bf16x4*)(pr + PM_C1 + c) = lo;
          }
        } else {
          if (grow < M) *(f32x4*)(feat + (long long)grow * PM_C1 + c) = y;
        }
      }
    }
    __syncthreads();   // the f32 tile is read: the next tile's hidden planes may be written (its A planes are complete)
  }
}

template <int TR, bool PLANES = false>
static int sf_pixel_feat_stream_launch_t(const float* x, const float* ln0_g, const float* ln0_b, const float* w1, const float* b1, const float* w2,
                                         const float* b2, const float* ln1_g, const float* ln1_b, float* feat, int M, float eps, hipStream_t st) {
  using Cfg = PfCfg<TR>;
  static_assert(Cfg::LDS <= 160 * 1024, "LDS budget");
  SF_TRY(sf_ensure_dyn_lds((const void*)pixel_feat_stream_kernel<TR, PLANES>, Cfg::LDS));
  const int ntiles = (M + TR - 1) / TR;
  sf_prof_begin(SF_K_LINEAR, st, 2.0 * M * (double)(PM_C0 * PM_C1 + PM_C1 * PM_C1));
  constexpr int env_pix = 0;
  const int tpw = env_pix >= TR ? env_pix / TR : Cfg::TPW;
  hipLaunchKernelGGL((pixel_feat_stream_kernel<TR, PLANES>), dim3((ntiles + tpw - 1) / tpw), dim3(Cfg::NT), Cfg::LDS, st, x, ln0_g, ln0_b, w1, b1,
                     w2, b2, ln1_g, ln1_b, feat, M, eps, tpw);
  sf_prof_end(SF_K_LINEAR, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// tile: 128 (one workgroup per CU) or 64 (two); 0 = the default
static int sf_pixel_feat_stream_launch(const float* x, const float* ln0_g, const float* ln0_b, const float* w1, const float* b1, const float* w2,
                                       const float* b2, const float* ln1_g, const float* ln1_b, float* feat, int M, float eps, hipStream_t st,
                                       int tile) {
  if (tile == 0) tile = 64;
  if (tile == 64) return sf_pixel_feat_stream_launch_t<64>(x, ln0_g, ln0_b, w1, b1, w2, b2, ln1_g, ln1_b, feat, M, eps, st);
  return sf_pixel_feat_stream_launch_t<128>(x, ln0_g, ln0_b, w1, b1, w2, b2, ln1_g, ln1_b, feat, M, eps, st);
}

// ================================================================================================
// PIXEL-STATIONARY form (round 6): the tile kernels above pay four workgroup barriers and three LDS round trips of the activations per 64 pixels -- 7.4 us
// per tile for 72 MFMAs per wave (59 us per launch of 32 frames, 0.085 of the roof, whatever the tile size).  Here a wave owns 32 pixels for the whole chain:
// every product runs transposed, D^T[feature][pixel] = W . A^T, with the pixels as the MFMA B operand in registers, and the accumulator layout of a 32 x 32
// block (lane = (pixel, half h), register r = row 8 (r >> 2) + 4 h + (r & 3)) IS the B operand of the next product once that product's weight fragments are
// packed in the matching k order (csrc/layer_tok.hip established this for the rollout layers):
//   x[pixel][64] (a lane loads the 32 contiguous channels 32 h .. + 31 of its pixel: one cache line) -> LayerNorm(64) (in registers + one exchange with the
//   lane 32 away) -> fc1 (4 blocks x 4 k-steps) + b1 + ReLU -> hi | lo fragments -> fc2 (4 blocks x 8 k-steps) + b2 -> LayerNorm(128) -> rows out.
// The MFMA row order of fc2's output blocks is chosen so that a lane ends up with the 64 CONTIGUOUS features 64 h .. + 63 of its pixel (row m = 8 g + 4 h + q
// of block ob = feature 64 h + 16 ob + 4 g + q): 128-byte segments out, f32 or bf16 hi | lo.
// No activation ever touches LDS; the 96 KB of weight fragments (split from the f32 matrices by the workgroup itself) sit in LDS for the whole launch;
// no barrier after the prologue.  Eight waves per workgroup (two per SIMD), one workgroup per CU of the stream, the rows of a wave's next tile in flight
// while the current one is multiplied, two accumulators alternating in every product.
namespace {
constexpr int PT_NT = 512;
constexpr int PT_W1 = 0, PT_W2 = 32 * 1024, PT_VEC = 96 * 1024;   // fc1 fragments [blk][ks][plane], fc2 fragments [ob][blk][s][plane] (1 KB each), vectors
constexpr int PV_G0 = 0, PV_B0 = 64, PV_B1 = 128, PV_B2 = 256, PV_G1 = 384, PV_BE1 = 512, PV_N = 640;
constexpr size_t PT_LDS = (size_t)PT_VEC + PV_N * 4;
}  // namespace

template <bool PLANES>
__global__ __launch_bounds__(PT_NT) void pixel_feat_tok_kernel(const float* __restrict__ x, const float* __restrict__ ln0_g, const float* __restrict__ ln0_b,
                                                               const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                               const float* __restrict__ b2, const float* __restrict__ ln1_g, const float* __restrict__ ln1_b,
                                                               void* __restrict__ out, int M, float eps, int tpw) {
  extern __shared__ __attribute__((aligned(16))) char pt_lds[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n = lane & 31, h = lane >> 5;
  // ---- weights -> split-bf16 fragments in LDS.  Fragment = 64 lanes x 16 B; lane (i, hh), element j:
  //        fc1 (blk, ks):     W1[32 blk + i][32 hh + 8 ks + j]                         (k order of the input fragments: a lane holds channels 32 hh .. + 31)
  //        fc2 (ob, blk, s):  W2[64 (i >> 2 & 1) ... see out_row][32 blk + 8 (2 s + (j >> 2)) + 4 hh + (j & 3)]
  //      fc2's MFMA row m = 8 g + 4 h' + q of block ob is feature 64 h' + 16 ob + 4 g + q ----
  {
    // (all twelve row requests of a thread in flight before the first conversion: six dependent round trips otherwise)
    f32x4 wa[6], wb[6];
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int f = t + it * PT_NT, fl = f & 63, fi = fl & 31, fh = fl >> 5, fr = f >> 6;
      const float* p;
      int second;
      if (fr < 16) {
        const int blk = fr >> 2, ks = fr & 3;
        p = w1 + (long long)(32 * blk + fi) * PM_C0 + 32 * fh + 8 * ks;
        second = 4;
      } else {
        const int g2 = fr - 16, ob = g2 >> 3, blk = (g2 >> 1) & 3, sx = g2 & 1;
        const int orow = 64 * ((fi >> 2) & 1) + 16 * ob + 4 * (fi >> 3) + (fi & 3);
        p = w2 + (long long)orow * PM_C1 + 32 * blk + 16 * sx + 4 * fh;
        second = 8;
      }
      wa[it] = *(const f32x4*)p;              // j = 0..3
      wb[it] = *(const f32x4*)(p + second);   // j = 4..7 (fc2: columns + 8 .. + 11)
    }
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int f = t + it * PT_NT, fl = f & 63, fr = f >> 6;
      const bf16x4 ah = __builtin_convertvector(wa[it], bf16x4), bh = __builtin_convertvector(wb[it], bf16x4);
      const bf16x4 al = __builtin_convertvector(wa[it] - __builtin_convertvector(ah, f32x4), bf16x4);
      const bf16x4 bl = __builtin_convertvector(wb[it] - __builtin_convertvector(bh, f32x4), bf16x4);
      char* dst = pt_lds + (size_t)fr * 2048 + fl * 16;
      *(bf16x8*)dst = __builtin_shufflevector(ah, bh, 0, 1, 2, 3, 4, 5, 6, 7);
      *(bf16x8*)(dst + 1024) = __builtin_shufflevector(al, bl, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  }
  float* PV = (float*)(pt_lds + PT_VEC);
  for (int i = t; i < PV_N; i += PT_NT)
    PV[i] = i < PV_B0 ? ln0_g[i] : i < PV_B1 ? ln0_b[i - PV_B0] : i < PV_B2 ? b1[i - PV_B1] : i < PV_G1 ? b2[i - PV_B2] : i < PV_BE1 ? ln1_g[i - PV_G1] : ln1_b[i - PV_BE1];
  __syncthreads();
  auto other = [&](float v) {   // the value of the lane 32 away (layer_tok.hip lt_xother)
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, h ? r2[0] : r2[1]);
  };
  auto pairsum = [&](float v) {
    const float o = other(v);
    return h ? o + v : v + o;   // (lower half first in both lanes: the two halves of a pixel get the same bits)
  };
  auto split8 = [](const f32x4 a, const f32x4 b, bf16x8& hi, bf16x8& lo) {
    const bf16x4 h0 = __builtin_convertvector(a, bf16x4), h1 = __builtin_convertvector(b, bf16x4);
    const bf16x4 l0 = __builtin_convertvector(a - __builtin_convertvector(h0, f32x4), bf16x4);
    const bf16x4 l1 = __builtin_convertvector(b - __builtin_convertvector(h1, f32x4), bf16x4);
    hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  const char* wl = pt_lds + lane * 16;
  // tile = 32 pixels; consecutive tiles go to consecutive WAVES (a workgroup covers 256 consecutive pixels per round: its stores fill whole rows together)
  const long long ntile = ((long long)M + 31) / 32, tstep = (long long)gridDim.x * (PT_NT / 64);
  long long tile = (long long)blockIdx.x * (PT_NT / 64) + wave;
  auto row_of = [&](long long tl) {
    const long long px = tl * 32 + n;
    return px < M ? px : (long long)M - 1;
  };
  f32x4 xn[8];   // the NEXT tile's rows, requested while the current tile is multiplied
  if (tile < ntile) {
    const float* xr = x + row_of(tile) * PM_C0 + 32 * h;
#pragma unroll
    for (int i = 0; i < 8; ++i) xn[i] = *(const f32x4*)(xr + 4 * i);
  }
#pragma unroll 1
  for (int ti = 0; ti < tpw; ++ti, tile += tstep) {
    if (tile >= ntile) break;
    const long long p0 = tile * 32;
    const long long pix = row_of(tile);
    // ---- the pixel's channels 32 h .. + 31, LayerNorm(64) ----
    f32x4 xv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) xv[i] = xn[i];
    if (tile + tstep < ntile) {
      const float* xr = x + row_of(tile + tstep) * PM_C0 + 32 * h;
#pragma unroll
      for (int i = 0; i < 8; ++i) xn[i] = *(const f32x4*)(xr + 4 * i);
    }
    float s0 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s0 += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
    const float mean = pairsum(s0) * (1.0f / PM_C0);
    float v0 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      xv[i] -= mean;
      v0 += (xv[i][0] * xv[i][0] + xv[i][1] * xv[i][1]) + (xv[i][2] * xv[i][2] + xv[i][3] * xv[i][3]);
    }
    const float rstd = 1.0f / sqrtf(pairsum(v0) * (1.0f / PM_C0) + eps);
    bf16x8 xh[4], xl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = 32 * h + 8 * ks;
      split8(xv[2 * ks] * rstd * *(const f32x4*)(PV + PV_G0 + c) + *(const f32x4*)(PV + PV_B0 + c),
             xv[2 * ks + 1] * rstd * *(const f32x4*)(PV + PV_G0 + c + 4) + *(const f32x4*)(PV + PV_B0 + c + 4), xh[ks], xl[ks]);
    }
    // ---- fc1 + b1 + ReLU: hidden block blk -> the two virtual k-steps (blk, s) of fc2 ----
    bf16x8 hh[8], hl[8];
#pragma unroll
    for (int bp = 0; bp < 2; ++bp) {   // two hidden blocks at a time: their accumulators alternate (no MFMA waits for the one in front of it)
      f32x16 acc[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 wh[2], wlo[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const char* wp = wl + PT_W1 + ((2 * bp + e) * 4 + ks) * 2048;
          wh[e] = *(const bf16x8*)wp;
          wlo[e] = *(const bf16x8*)(wp + 1024);
        }
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[0], xl[ks], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[1], xl[ks], acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[0], xh[ks], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[1], xh[ks], acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[0], xh[ks], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[1], xh[ks], acc[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);   // (left alone the scheduler requests every fragment of the product first: 564 spilled registers)
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int blk = 2 * bp + e;
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
          f32x4 u[2];
#pragma unroll
          for (int gg = 0; gg < 2; ++gg) {
            const int g = 2 * sx + gg;
            const f32x4 bb = *(const f32x4*)(PV + PV_B1 + 32 * blk + 8 * g + 4 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) u[gg][q] = fmaxf(acc[e][4 * g + q] + bb[q], 0.f);
          }
          split8(u[0], u[1], hh[2 * blk + sx], hl[2 * blk + sx]);
        }
      }
    }
    // ---- fc2 + b2: output block ob = features 64 h + 16 ob + 4 g + q of this lane's pixel ----
    f32x16 Y[4];
#pragma unroll
    for (int op = 0; op < 2; ++op) {
#pragma unroll
      for (int r = 0; r < 16; ++r) Y[2 * op][r] = Y[2 * op + 1][r] = 0.f;
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        bf16x8 wh[2], wlo[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const char* wp = wl + PT_W2 + (((2 * op + e) * 4 + (k2 >> 1)) * 2 + (k2 & 1)) * 2048;
          wh[e] = *(const bf16x8*)wp;
          wlo[e] = *(const bf16x8*)(wp + 1024);
        }
        Y[2 * op] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[0], hl[k2], Y[2 * op], 0, 0, 0);
        Y[2 * op + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[1], hl[k2], Y[2 * op + 1], 0, 0, 0);
        Y[2 * op] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[0], hh[k2], Y[2 * op], 0, 0, 0);
        Y[2 * op + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[1], hh[k2], Y[2 * op + 1], 0, 0, 0);
        Y[2 * op] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[0], hh[k2], Y[2 * op], 0, 0, 0);
        Y[2 * op + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[1], hh[k2], Y[2 * op + 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float s1 = 0.f;
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bb = *(const f32x4*)(PV + PV_B2 + 64 * h + 16 * ob + 4 * g);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          Y[ob][4 * g + q] += bb[q];
          s1 += Y[ob][4 * g + q];
        }
      }
    const float mean1 = pairsum(s1) * (1.0f / PM_C1);
    float v1 = 0.f;
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        Y[ob][r] -= mean1;
        v1 += Y[ob][r] * Y[ob][r];
      }
    const float rstd1 = 1.0f / sqrtf(pairsum(v1) * (1.0f / PM_C1) + eps);
    const bool live = p0 + n < M;
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        f32x4 y[2];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int g = 2 * gp + gg, c = 64 * h + 16 * ob + 4 * g;
          const f32x4 ga = *(const f32x4*)(PV + PV_G1 + c), be = *(const f32x4*)(PV + PV_BE1 + c);
#pragma unroll
          for (int q = 0; q < 4; ++q) y[gg][q] = Y[ob][4 * g + q] * rstd1 * ga[q] + be[q];
        }
        const int c0 = 64 * h + 16 * ob + 8 * gp;   // eight consecutive features
        if (live) {
          if constexpr (PLANES) {
            bf16x8 yh, yl;
            split8(y[0], y[1], yh, yl);
            __bf16* pr = (__bf16*)out + pix * (2 * PM_C1);
            *(bf16x8*)(pr + c0) = yh;
            *(bf16x8*)(pr + PM_C1 + c0) = yl;
          } else {
            float* pr = (float*)out + pix * PM_C1;
            *(f32x4*)(pr + c0) = y[0];
            *(f32x4*)(pr + c0 + 4) = y[1];
          }
        }
      }
  }
}

template <bool PLANES>
static int sf_pixel_feat_tok_launch(const float* x, const float* ln0_g, const float* ln0_b, const float* w1, const float* b1, const float* w2, const float* b2,
                                    const float* ln1_g, const float* ln1_b, void* out, int M, float eps, hipStream_t st) {
  static_assert(PT_LDS <= 160 * 1024, "LDS budget");
  SF_TRY(sf_ensure_dyn_lds((const void*)pixel_feat_tok_kernel<PLANES>, PT_LDS));
  // one workgroup per CU of the stream where the problem is that large (each pays 96 KB of weights once); consecutive tiles to consecutive waves
  const long long ntile = ((long long)M + 31) / 32;
  long long grid = (ntile + (PT_NT / 64) - 1) / (PT_NT / 64);
  const int cus = sf_stream_cus((void*)st);
  if (grid > cus) grid = cus;
  const int tpw = (int)((ntile + grid * (PT_NT / 64) - 1) / (grid * (PT_NT / 64)));
  sf_prof_begin(SF_K_LINEAR, st, 2.0 * M * (double)(PM_C0 * PM_C1 + PM_C1 * PM_C1));
  hipLaunchKernelGGL(pixel_feat_tok_kernel<PLANES>, dim3((unsigned)grid), dim3(PT_NT), PT_LDS, st, x, ln0_g, ln0_b, w1, b1, w2, b2, ln1_g, ln1_b, out, M, eps, tpw);
  sf_prof_end(SF_K_LINEAR, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// the same with the result as bf16 hi | lo rows of 512 B (slot_chain.h): the Slot-Attention inputs of the video-stationary slot branch
int sf_pixel_mlp_feat_planes_ex(const float* x, const float* ln0_g, const float* ln0_b, const float* w1, const float* b1, const float* w2,
                                const float* b2, const float* ln1_g, const float* ln1_b, void* planes, int M, float eps, hipStream_t st) {
  if (sf_get_pixel_tok()) return sf_pixel_feat_tok_launch<true>(x, ln0_g, ln0_b, w1, b1, w2, b2, ln1_g, ln1_b, planes, M, eps, st);
  return sf_pixel_feat_stream_launch_t<64, true>(x, ln0_g, ln0_b, w1, b1, w2, b2, ln1_g, ln1_b, (float*)planes, M, eps, st);
}

// Kernel-level entry point (include/slotformer_hip.h): feat [M][128] = LN(128)(fc2(relu(fc1(LN(64)(x))))) -- encoder_out_layer followed by
// SlotAttention.norm_inputs (savi.py:245-250, 66-70).  form 0: one 128-pixel tile per workgroup, weights through LDS; 1: weights resident in
// registers, four tiles per workgroup; 2: the same on 64-pixel tiles, 256 threads, two workgroups per CU (0-2: the same bits); 3: pixel-stationary
// (pixel_feat_tok_kernel: another summation order, split-bf16 rounding apart).
extern "C" int sf_pixel_feat_f32(const float* x, const float* ln0_g, const float* ln0_b, const float* w1, const float* b1, const float* w2,
                                 const float* b2, const float* ln1_g, const float* ln1_b, float* feat, int M, float eps, int form, void* stream) {
  SF_REQUIRE(x && ln0_g && ln0_b && w1 && b1 && w2 && b2 && ln1_g && ln1_b && feat && M > 0, "sf_pixel_feat_f32: null pointer / empty problem");
  SF_REQUIRE(form >= 0 && form <= 3, "sf_pixel_feat_f32: form must be 0 .. 3");
  SF_REQUIRE(sf_get_precision() == 1, "sf_pixel_feat_f32: split-bf16 mode only");
  hipStream_t st = (hipStream_t)stream;
  if (form == 3) return sf_pixel_feat_tok_launch<false>(x, ln0_g, ln0_b, w1, b1, w2, b2, ln1_g, ln1_b, feat, M, eps, st);   // pixel-stationary
  if (form >= 1) return sf_pixel_feat_stream_launch(x, ln0_g, ln0_b, w1, b1, w2, b2, ln1_g, ln1_b, feat, M, eps, st, form == 1 ? 128 : 64);
  SF_TRY(sf_ensure_dyn_lds((const void*)pixel_mlp_kv_kernel<true>, (size_t)(PM_LDS)));
  hipLaunchKernelGGL(pixel_mlp_kv_kernel<true>, dim3((M + PM_ROWS - 1) / PM_ROWS), dim3(PM_NT), PM_LDS, st, x, ln0_g, ln0_b, w1, b1, w2, b2,
                     ln1_g, ln1_b, nullptr, feat, M, eps);
  SF_CHECK_LAUNCH();
  return 0;
}
