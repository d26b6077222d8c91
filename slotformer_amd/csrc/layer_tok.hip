// One nn.TransformerEncoderLayer (norm_first, relu; d_model 256, 8 heads, ffn 1024) of the rollout Transformer in a TOKEN-STATIONARY form:
//     x2 = x + out_proj(MHA(LN1(x))) + b_o ;   y = x2 + lin2(relu(lin1(LN2(x2))))          (slotformer.py:72-80, torch nn.TransformerEncoderLayer)
// A workgroup owns WHOLE videos -- vpw = 128 / L of them, up to 128 token rows -- and each of its four waves (one per SIMD, up to 512 registers) owns 32
// tokens for the whole layer.  Every linear product runs transposed, D^T[feature][token] = W . A^T, with the wave's activations as the MFMA B operand in
// registers: the accumulator layout of a 32 x 32 block (lane = (token, half h), registers = features 8 g + 4 h + q) IS a B operand of the next product once
// the weight fragments are packed with the matching permuted k order (ffn_tok.hip established this for the FFN pair).  So LN1 -> q|k|v -> scores ->
// softmax -> PV -> out-projection -> LN2 -> FFN1 -> ReLU -> FFN2 is ONE chain of register-resident products per wave:
//   * the residual stream lives in the accumulators the out-projection / FFN2 add into (x + b_o, then x2 + b2, are their initial values);
//   * waves share WEIGHTS, not activations: the layer's 3 MB of split-bf16 fragments, packed in consumption order (sf_pack_layer_tok_weights: 96 stages of
//     32 fragments = one 32-row block of a matrix each), stream global -> LDS through a three-stage ring (global_load_lds, one fragment per
//     wave-instruction, a counted vmcnt wait + ONE barrier per stage);
//   * the only activations that cross waves are a head's keys and values: the owner writes its k accumulators as they lie (they are the A fragments of
//     S^T = K Q^T) and computes v with the MFMA operands swapped, which leaves V^T fragments (lane = dim, registers = keys) in its accumulators --
//     32 KB of LDS per head, handed over by the stage barriers; the scores of a wave's 32 queries against the (at most three) 32-key blocks that hold
//     their videos' tokens stay in registers through the softmax and feed the PV product as its B operand.
// No activation plane, no q|k|v round trip through memory, no per-video attention launch: a layer is one launch of ceil(B / vpw) workgroups
// (43 for a rollout unit of 128 videos at L = 42) instead of 84 FFN + 128 attention workgroups over two launches.
#include "../../include/slotformer_hip.h"
#include "sf_internal.h"
#include "layer_fused.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int LT_NT = 256, LT_D = 256, LT_F = 1024, LT_NH = 8, LT_TOK = 128, LT_MAXL = 8;
constexpr int LT_STAGE = 32 * 1024, LT_RING = 3;
constexpr int LT_NST_ATT = 4 * LT_NH, LT_NST = LT_NST_ATT + 2 * (LT_F / 32);   // 32 attention + 64 FFN stages per layer
constexpr int LT_KV = LT_RING * LT_STAGE;     // K fragments [s][plane][token block][64 lanes] x 16 B, then V^T fragments [token block][s][plane][64 lanes] x 16 B
constexpr int LT_VT = LT_KV + 16 * 1024;
constexpr int LT_PAR = LT_KV + 32 * 1024;     // the layer's vectors (f32)
constexpr int P_LN1G = 0, P_LN1B = 256, P_BQKV = 512, P_BO = 1280, P_LN2G = 1536, P_LN2B = 1792, P_B1 = 2048, P_B2 = 3072, P_N = 3328;
constexpr size_t LT_LDS = (size_t)LT_PAR + (size_t)P_N * 4;
constexpr size_t LT_BLOB = (size_t)LT_NST * LT_STAGE + (size_t)P_N * 4;   // a layer's fragments + its vectors
static_assert(LT_LDS <= 160 * 1024, "LDS budget");

struct LtArgs {
  const float* x;      // MODE 0: [B * L][256] rows
  const float* ring;   // MODE 1: projection ring [B][RF][N][256] ...
  const float* pe;     //         ... + position table [L][256]
  float* y;            // [B * L][256]
  const char* blob[LT_MAXL];   // sf_pack_layer_tok_weights copies of the layers this launch runs
  float eps;
  int nl, B, L, vpw, RF, N, f0, dbg_ts;
};

__device__ __forceinline__ bf16x8 cat8(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
// eight f32 -> hi | lo bf16 fragments
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, bf16x8& hi, bf16x8& lo) {
  const bf16x4 h0 = __builtin_convertvector(a, bf16x4), h1 = __builtin_convertvector(b, bf16x4);
  const bf16x4 l0 = __builtin_convertvector(a - __builtin_convertvector(h0, f32x4), bf16x4);
  const bf16x4 l1 = __builtin_convertvector(b - __builtin_convertvector(h1, f32x4), bf16x4);
  hi = cat8(h0, h1);
  lo = cat8(l0, l1);
}
__device__ __forceinline__ f32x4 quad(const f32x16& a, int g) { return f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]}; }
// the value of the lane that holds the other half of this token's features / keys (lane ^ 32): one v_permlane32_swap, no LDS.  (The instruction swaps the
// upper half of its first operand with the lower half of its second; whether the compiler gives the two copies of `v` one register or two, the partner's
// value is result 0 in the upper half and result 1 in the lower.)
__device__ __forceinline__ float lt_xother(float v, int h) {
#ifdef LT_SHFL
  return __shfl_xor(v, 32, 64);
#endif
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, h ? r[0] : r[1]);
}
__device__ __forceinline__ float lt_xmax(float v, int h) { return fmaxf(v, lt_xother(v, h)); }
__device__ __forceinline__ float lt_xsum(float v, int h) {
  const float o = lt_xother(v, h);
  return h ? o + v : v + o;   // (lower half's value first in both lanes: the two halves of a token get the same bits)
}

// the ring as one wave sees it during a stage: `rd` = this lane's read address of fragment 0 of the current stage, (`src`, `dst`) = where piece 0 of the
// stage two ahead comes from (per lane) / goes to (wave-uniform); a product issues one piece per fragment group
struct LtRing {
  const char* rd;
  const char* src;
  char* dst;
};
__device__ __forceinline__ void lt_dma_now(const LtRing& R, int f) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(R.src + f * 1024),
                                   (void __attribute__((address_space(3)))*)(R.dst + f * 1024), 16, 0, 0);
}
__device__ __forceinline__ void lt_dma(const LtRing& R, int f) {
#if defined(LT_NODMA) || defined(LT_BURST)
  return;
#endif
#ifndef LT_NOIMMOFF
  // the instruction's immediate offset applies to BOTH addresses: pieces 4 k .. 4 k + 3 share one per-lane source address and one M0 value
  // (354 instead of 365 us per 3-layer launch of 64 workgroups: three instructions less around every piece)
  const void __attribute__((address_space(1)))* s = (const void __attribute__((address_space(1)))*)(R.src + (f >> 2) * 4096);
  void __attribute__((address_space(3)))* d = (void __attribute__((address_space(3)))*)(R.dst + (f >> 2) * 4096);
  switch (f & 3) {
    case 0: __builtin_amdgcn_global_load_lds(s, d, 16, 0, 0); break;
    case 1: __builtin_amdgcn_global_load_lds(s, d, 16, 1024, 0); break;
    case 2: __builtin_amdgcn_global_load_lds(s, d, 16, 2048, 0); break;
    default: __builtin_amdgcn_global_load_lds(s, d, 16, 3072, 0); break;
  }
#else
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(R.src + f * 1024),
                                   (void __attribute__((address_space(3)))*)(R.dst + f * 1024), 16, 0, 0);
#endif
}
#define LT_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
// keeps the MFMAs on either side in source order (every other class may cross): the scheduler otherwise groups the MFMAs of one accumulator, and a chain
// of dependent MFMAs issues every ~44 cycles instead of 32
#define LT_PIN() __builtin_amdgcn_sched_barrier(0x7F6)
#define LT_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
struct LtNoSide {
  __device__ __forceinline__ void operator()(int) const {}
};
// the issue pattern of a fragment group: its first MFMA, the four fragment reads of the next group, then the other five MFMAs with up to NV VALU
// instructions of the side work behind each (a wave alone on its SIMD hides about five issue slots under an MFMA)
template <int NV, bool READS>
__device__ __forceinline__ void lt_group_pattern() {
  LT_SGB(0x008, 1);
  if constexpr (READS) LT_SGB(0x100, 4);
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    if constexpr (NV > 0) LT_SGB(0x002, NV);
    LT_SGB(0x008, 1);
  }
  if constexpr (NV > 0) LT_SGB(0x002, NV);
  __builtin_amdgcn_sched_barrier(0);
}

// D^T[32 features][32 tokens] (SW: D[32 tokens][32 features]) = W block . A^T over the 256 input channels: fragment group g = (hi, lo) of the virtual
// k-steps 2 g and 2 g + 1.  Consecutive MFMAs ALTERNATE between two accumulators (even / odd k-steps, summed at the end): anything issued between two
// MFMAs on the SAME accumulator costs ~43 cycles (MI355X_MICROARCH.md), between MFMAs on different ones ~6.  The reads of group g + 1 go behind the first
// MFMA of group g (ffn_tok.hip); side(g) = VALU work of ANOTHER chain (the previous product's conversion, a softmax slice) that the scheduler places
// between this group's MFMAs.
template <bool SW, int NV, class Side>
__device__ __forceinline__ void lt_row_product(const LtRing& R, const bf16x8 (&xh)[16], const bf16x8 (&xl)[16], f32x16& a, Side&& side) {
  bf16x8 wb[2][4];
  f32x16 a1;
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = a1[r] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) wb[0][i] = *(const bf16x8*)(R.rd + i * 1024);
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int b = g & 1, ks = 2 * g;
    if constexpr (!SW) a = LT_MFMA(wb[b][0], xl[ks], a); else a = LT_MFMA(xl[ks], wb[b][0], a);
#ifndef LT_NOREAD
    if (g + 1 < 8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) wb[b ^ 1][i] = *(const bf16x8*)(R.rd + (4 * (g + 1) + i) * 1024);
    }
#else
    if (g == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) wb[1][i] = *(const bf16x8*)(R.rd + (4 + i) * 1024);
    }
#endif
    lt_dma(R, g);
    LT_PIN();
    if constexpr (!SW) {
      a1 = LT_MFMA(wb[b][2], xl[ks + 1], a1);
      LT_PIN();
      a = LT_MFMA(wb[b][1], xh[ks], a);
      LT_PIN();
      a1 = LT_MFMA(wb[b][3], xh[ks + 1], a1);
      LT_PIN();
      a = LT_MFMA(wb[b][0], xh[ks], a);
      LT_PIN();
      a1 = LT_MFMA(wb[b][2], xh[ks + 1], a1);
    } else {
      a1 = LT_MFMA(xl[ks + 1], wb[b][2], a1);
      LT_PIN();
      a = LT_MFMA(xh[ks], wb[b][1], a);
      LT_PIN();
      a1 = LT_MFMA(xh[ks + 1], wb[b][3], a1);
      LT_PIN();
      a = LT_MFMA(xh[ks], wb[b][0], a);
      LT_PIN();
      a1 = LT_MFMA(xh[ks + 1], wb[b][2], a1);
    }
    LT_PIN();
#ifndef LT_NOSIDE
    side(g);
    if (g + 1 < 8) lt_group_pattern<NV, true>(); else lt_group_pattern<NV, false>();
#else
    if (g + 1 < 8) lt_group_pattern<0, true>(); else lt_group_pattern<0, false>();
#endif
  }
#ifdef LT_NOSIDE
#pragma unroll
  for (int g = 0; g < 8; ++g) side(g);
#endif
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] += a1[r];
}
// Y^T[256 features][32 tokens] += W[:, 32-wide slice] . I^T: fragment group p = the (hi, lo) fragments of the slice's two virtual k-steps for the output
// blocks 2 p and 2 p + 1, whose accumulators alternate (see above)
template <int NV, class Side>
__device__ __forceinline__ void lt_kslice_product(const LtRing& R, const bf16x8 (&ih)[2], const bf16x8 (&il)[2], f32x16 (&Y)[8], Side&& side) {
  bf16x8 wb[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) wb[0][i] = *(const bf16x8*)(R.rd + i * 1024);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int b = p & 1, oa = 2 * p, ob = 2 * p + 1;
    Y[oa] = LT_MFMA(wb[b][0], il[0], Y[oa]);
#ifndef LT_NOREAD
    if (p + 1 < 4) {
#pragma unroll
      for (int i = 0; i < 8; ++i) wb[b ^ 1][i] = *(const bf16x8*)(R.rd + (8 * (p + 1) + i) * 1024);
    }
#else
    if (p == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) wb[1][i] = *(const bf16x8*)(R.rd + (8 + i) * 1024);
    }
#endif
    lt_dma(R, 2 * p);
    LT_PIN();
    Y[ob] = LT_MFMA(wb[b][4], il[0], Y[ob]);
    LT_PIN();
    Y[oa] = LT_MFMA(wb[b][1], ih[0], Y[oa]);
    LT_PIN();
    Y[ob] = LT_MFMA(wb[b][5], ih[0], Y[ob]);
    LT_PIN();
    Y[oa] = LT_MFMA(wb[b][0], ih[0], Y[oa]);
    LT_PIN();
    Y[ob] = LT_MFMA(wb[b][4], ih[0], Y[ob]);
    lt_dma(R, 2 * p + 1);
    LT_PIN();
    Y[oa] = LT_MFMA(wb[b][2], il[1], Y[oa]);
    LT_PIN();
    Y[ob] = LT_MFMA(wb[b][6], il[1], Y[ob]);
    LT_PIN();
    Y[oa] = LT_MFMA(wb[b][3], ih[1], Y[oa]);
    LT_PIN();
    Y[ob] = LT_MFMA(wb[b][7], ih[1], Y[ob]);
    LT_PIN();
    Y[oa] = LT_MFMA(wb[b][2], ih[1], Y[oa]);
    LT_PIN();
    Y[ob] = LT_MFMA(wb[b][6], ih[1], Y[ob]);
    LT_PIN();
#ifndef LT_NOSIDE
    side(2 * p);
    side(2 * p + 1);
#endif
    // first MFMA, the eight fragment reads of the next pair, then the other eleven MFMAs with the side work between them
    LT_SGB(0x008, 1);
    if (p + 1 < 4) LT_SGB(0x100, 8);
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      if constexpr (NV > 0) LT_SGB(0x002, NV);
      LT_SGB(0x008, 1);
    }
    if constexpr (NV > 0) LT_SGB(0x002, NV);
    __builtin_amdgcn_sched_barrier(0);
  }
#ifdef LT_NOSIDE
#pragma unroll
  for (int g = 0; g < 8; ++g) side(g);
#endif
}

// LayerNorm over the 256 features of a token held as X[ob][4 g + q] = feature 32 ob + 8 g + 4 h + q by lanes (token, h = 0 / 1); gamma / beta in LDS at
// float offsets GOFF / BOFF behind pb (= the vector block + 4 h floats, an opaque per-lane base: every read is base + immediate):
// the normalised row as the 16 hi | lo fragments of the virtual k-steps (block ob, s): registers 8 s .. 8 s + 7
template <int GOFF, int BOFF>
__device__ __forceinline__ void lt_layernorm(const f32x16 (&X)[8], const char* pb, int h, float eps, bf16x8 (&xh)[16], bf16x8 (&xl)[16]) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int r = 0; r < 16; r += 4) {
      s0 += X[ob][r];
      s1 += X[ob][r + 1];
      s2 += X[ob][r + 2];
      s3 += X[ob][r + 3];
    }
  const float mu = lt_xsum((s0 + s1) + (s2 + s3), h) * (1.0f / LT_D);
  float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int r = 0; r < 16; r += 4) {
      const float d0 = X[ob][r] - mu, d1 = X[ob][r + 1] - mu, d2 = X[ob][r + 2] - mu, d3 = X[ob][r + 3] - mu;
      q0 += d0 * d0;
      q1 += d1 * d1;
      q2 += d2 * d2;
      q3 += d3 * d3;
    }
  const float rs = 1.0f / sqrtf(lt_xsum((q0 + q1) + (q2 + q3), h) * (1.0f / LT_D) + eps);
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int s2i = 0; s2i < 2; ++s2i) {
      f32x4 v[2];
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const int g = 2 * s2i + gg, c = 32 * ob + 8 * g;
        v[gg] = (quad(X[ob], g) - mu) * rs * *(const f32x4*)(pb + (GOFF + c) * 4) + *(const f32x4*)(pb + (BOFF + c) * 4);
      }
      split8(v[0], v[1], xh[2 * ob + s2i], xl[2 * ob + s2i]);
      __builtin_amdgcn_sched_barrier(0);   // (left alone the scheduler requests all 64 gamma / beta vectors first: 200 spilled registers)
    }
}
// X[ob][4 g + q] += vec[32 ob + 8 g + 4 h + q]  (vec at float offset OFF behind pb)
template <int OFF>
__device__ __forceinline__ void lt_add_vec(f32x16 (&X)[8], const char* pb) {
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bv = *(const f32x4*)(pb + (OFF + 32 * ob + 8 * g) * 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) X[ob][4 * g + q] += bv[q];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
}  // namespace

// ---- weights of one layer -> 96 stages x 32 fragments x 64 lanes x 16 B in CONSUMPTION order (the kernel below), then the layer's vectors ---------------
//   attention, stage a:  0 / 1 / 2: q / k / v rows of head 0;  3 + 4 (i - 1) + {0, 1, 2, 3}: q / k / v rows of head i and the out_proj columns of head i - 1
//                        (i = 1..7);  31: the out_proj columns of head 7     (q rows 32 i, k rows 256 + 32 i, v rows 512 + 32 i of in_proj_w [768][256])
//   FFN, stage 32 + f:   f = 0: lin1 rows of hidden block 0;  f = 2 j - 1: lin1 rows of block j,  f = 2 j: lin2 columns of block j - 1  (j = 1..31);
//                        f = 63: lin2 columns of block 31
//   row stages:    fragment f = 2 vk + plane, virtual k-step vk = (input block ib, s);  column stages: fragment f = 4 ob + 2 s + plane (output block ob)
//   element j of lane (i, h) of a fragment over columns c0 .. c0 + 31 at k-step s:  W[r0 + i][c0 + 8 (2 s + (j >> 2)) + 4 h + (j & 3)]
//   (the k order of an accumulator: register 8 s + j of lane (token, h) holds feature 8 (2 s + (j >> 2)) + 4 h + (j & 3) of its 32-block)
__global__ void pack_layer_tok_kernel(const float* __restrict__ win, const float* __restrict__ wout, const float* __restrict__ w1,
                                      const float* __restrict__ w2, uint4* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= LT_NST * 32 * 64) return;
  const int lane = idx & 63, f = (idx >> 6) & 31, st = idx >> 11;
  const int pl = f & 1, h = lane >> 5, i = lane & 31;
  const float* W;
  int ld, r0 = 0, c0 = 0, s;
  bool slice;
  if (st < LT_NST_ATT) {
    int hd, kind;   // 0 q, 1 k, 2 v, 3 out_proj columns
    if (st < 3) {
      hd = 0; kind = st;
    } else if (st == LT_NST_ATT - 1) {
      hd = 7; kind = 3;
    } else {
      const int tq = st - 3;
      kind = tq & 3;
      hd = 1 + (tq >> 2) - (kind == 3 ? 1 : 0);
    }
    slice = kind == 3;
    ld = LT_D;
    if (!slice) {
      W = win; r0 = (kind == 0 ? 0 : kind == 1 ? 256 : 512) + 32 * hd;
    } else {
      W = wout; c0 = 32 * hd;
    }
  } else {
    const int ff = st - LT_NST_ATT;
    int hb;
    if (ff == 0) {
      slice = false; hb = 0;
    } else if (ff == 63) {
      slice = true; hb = 31;
    } else if (ff & 1) {
      slice = false; hb = (ff + 1) >> 1;
    } else {
      slice = true; hb = (ff >> 1) - 1;
    }
    if (!slice) {
      W = w1; ld = LT_D; r0 = 32 * hb;
    } else {
      W = w2; ld = LT_F; c0 = 32 * hb;
    }
  }
  if (!slice) {
    const int vk = f >> 1;
    c0 = 32 * (vk >> 1);
    s = vk & 1;
  } else {
    r0 = 32 * (f >> 2);
    s = (f >> 1) & 1;
  }
  const float* src = W + (long long)(r0 + i) * ld + c0 + 4 * h;
  union {
    __bf16 b[8];
    uint4 u;
  } o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = src[8 * (2 * s + (j >> 2)) + (j & 3)];
    const __bf16 hi = (__bf16)v;
    o.b[j] = pl ? (__bf16)(v - (float)hi) : hi;
  }
  out[idx] = o.u;
}

__device__ long long lt_ts[16];   // wall-clock stamps of workgroup 0, wave 0 (SF_DBG=lt; sf_debug_read_ts_layer_tok)
#define LTS(i) do { if (A.dbg_ts && blockIdx.x == 0 && threadIdx.x == 0) lt_ts[i] = wall_clock64(); } while (0)

template <int MODE>
__global__ __launch_bounds__(LT_NT) void layer_tok_kernel(LtArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* P = (float*)(smem + LT_PAR);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int L = A.L;
  const int v0 = blockIdx.x * A.vpw;
  const int nvalid = min(A.vpw, A.B - v0) * L;   // token rows of this workgroup's videos
  const int tl = wave * 32 + n;
  const int te = min(tl, nvalid - 1);             // (rows past the end repeat the last one: finite values, never stored)
  const int vl = te / L, tok = te - vl * L;
  const long long row = (long long)(v0 + vl) * L + tok;
  LTS(0);
  // ---- the token's row in accumulator layout: X[ob][4 g + q] = x[32 ob + 8 g + 4 h + q] ----
  f32x16 X[8];
  if constexpr (MODE == 0) {
    const float* xr = A.x + row * LT_D + 4 * h;
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = *(const f32x4*)(xr + 32 * ob + 8 * g);
#pragma unroll
        for (int q = 0; q < 4; ++q) X[ob][4 * g + q] = v[q];
      }
  } else {
    // layer 0 of a rollout step: x = ring[b][(f0 + tok / N) % RF][tok % N] + pe[tok]   (slotformer.py:115-117; the in-projections are cached per frame)
    const int fr = tok / A.N, sl = tok - fr * A.N;
    const float* rr = A.ring + (((long long)(v0 + vl) * A.RF + (A.f0 + fr) % A.RF) * A.N + sl) * LT_D + 4 * h;
    const float* pr = A.pe + (long long)tok * LT_D + 4 * h;
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = *(const f32x4*)(rr + 32 * ob + 8 * g);
#pragma unroll
        for (int q = 0; q < 4; ++q) X[ob][4 * g + q] = v[q];
      }
    __builtin_amdgcn_sched_barrier(0);   // (the position rows behind the ring rows, a block at a time: 64 vectors in flight at once do not fit)
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = *(const f32x4*)(pr + 32 * ob + 8 * g);
#pragma unroll
        for (int q = 0; q < 4; ++q) X[ob][4 * g + q] += v[q];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- weight ring: the launch's stages (96 per layer, layer after layer) are ONE stream; stage gs lives in slot gs % 3; this wave copies fragments
  //      8 wave .. 8 wave + 7 of every stage, one per fragment group of the product that runs two stages earlier ----
  const unsigned wlane = (unsigned)(wave * 8 * 1024 + lane * 16);
  char* const wdst = smem + (wave * 8) * 1024;
  // prefetch cursor: the stage two ahead of consumer stage cs of the current layer -- stage cs + 2 of this layer's blob, the first two stages of the next
  // layer's behind the end, and past the end of the launch the last stage again (re-requested into a free slot: every stage issues its eight pieces, one
  // wait count fits all).  Scalar selects only: a branch here would split the stage's basic block (and the side work sinks into the block of its use).
  int cur = 0;   // slot of the next stage to be consumed
  int cs = -2;   // consumer stage inside the current layer (the two calls of the prologue request stages 0 and 1)
  const char* base_cur = A.blob[0];
  const char* base_nxt = A.blob[0];
  bool last_layer = false;
  auto stage_src = [&]() -> const char* {
    const int t2 = cs + 2;
    const bool wrap = t2 >= LT_NST;
    const char* b = wrap ? base_nxt : base_cur;
    const int so = wrap ? (last_layer ? LT_NST - 1 : t2 - LT_NST) : t2;
    ++cs;
    return b + (size_t)so * LT_STAGE + wlane;
  };
  {
    const char* s0 = stage_src();
    const char* s1 = stage_src();
    LtRing R0{nullptr, s0, wdst}, R1{nullptr, s1, wdst + LT_STAGE};
#pragma unroll
    for (int f = 0; f < 8; ++f) lt_dma(R0, f);
#pragma unroll
    for (int f = 0; f < 8; ++f) lt_dma(R1, f);
  }
  // the stage about to be consumed has landed (every wave waits for its own pieces, then the barrier), every wave is done with the previous one
  // (whose slot the pieces of stage gs + 2 go to), LDS writes of the previous stage (keys / values) are visible
#ifdef LT_STAMPS
  long long c_wait = 0, c_bar = 0, c_last = 0, c_work = 0;   // shader cycles (s_memtime) of workgroup 0's wave 0: DMA waits, barrier waits, everything else
#endif
  auto stage_begin = [&]() -> LtRing {
#ifdef LT_STAMPS
    const long long c0 = __builtin_readcyclecounter();
    if (c_last) c_work += c0 - c_last;
#endif
#ifdef LT_NODMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef LT_STAMPS
    const long long c1 = __builtin_readcyclecounter();
#endif
#ifndef LT_NOBAR
    __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");
#ifdef LT_STAMPS
    c_last = __builtin_readcyclecounter();
    c_wait += c1 - c0;
    c_bar += c_last - c1;
#endif
    const int nb = cur == 0 ? 2 : cur - 1;
    LtRing R{smem + cur * LT_STAGE + lane * 16, stage_src(), wdst + nb * LT_STAGE};
#ifdef LT_BURST
#pragma unroll
    for (int f = 0; f < 8; ++f) lt_dma_now(R, f);
#endif
    cur = cur == 2 ? 0 : cur + 1;
    return R;
  };
  // ---- key blocks this wave's queries need: the tokens of the videos its 32 rows belong to (at most three 32-key blocks, sf_layer_tok_ok); a block
  //      index past the last one is clamped for the READS (finite values), its keys fail the range test of the mask ----
  const int wf = min(wave * 32, nvalid - 1);
  const int kb0 = ((wf / L) * L) >> 5;
  const int d0 = 32 * kb0 + 4 * h - vl * L;   // key index of register (kbi, 4 g + q) minus the video's first key: d0 + 32 kbi + 8 g + q
  const char* pb;   // the vectors as this lane reads them: + 4 h floats; opaque, so that every read is this base + an immediate offset
  {
    unsigned pbo = (unsigned)(LT_PAR + 16 * h);
    asm volatile("" : "+v"(pbo));
    pb = smem + pbo;
  }
  char* const kwr = smem + LT_KV + wave * 1024 + lane * 16;          // + (s * 2 + plane) * 4096
  char* const vwr = smem + LT_VT + wave * 4096 + lane * 16;          // + (s * 2 + plane) * 1024
  const char* krd[3];
  const char* vrd[3];
#pragma unroll
  for (int kbi = 0; kbi < 3; ++kbi) {
    const int kb = min(kb0 + kbi, 3);
    krd[kbi] = smem + LT_KV + kb * 1024 + lane * 16;                 // + (s * 2 + plane) * 4096
    vrd[kbi] = smem + LT_VT + kb * 4096 + lane * 16;                 // + (s * 2 + plane) * 1024
  }
  const float qscale = 0.2550348616841918f;   // log2(e) / sqrt(32): the scores in the exponent's base
  constexpr float NEG = -3.0e38f;
  LTS(1);

#pragma unroll 1
  for (int l = 0; l < A.nl; ++l) {
    cs = 0;
    base_cur = A.blob[l];
    last_layer = l + 1 >= A.nl;
    base_nxt = A.blob[last_layer ? l : l + 1];
    // ---- the layer's vectors -> LDS (behind the first barrier every wave is done with the previous layer's) ----
    __syncthreads();
    {
      const float* vsrc = (const float*)(A.blob[l] + (size_t)LT_NST * LT_STAGE);
      for (int i = t; i < P_N / 4; i += LT_NT) *(f32x4*)(P + 4 * i) = *(const f32x4*)(vsrc + 4 * i);
    }
    __syncthreads();
    if (l == 0) LTS(2);
    // ================================================= attention block =================================================
    bf16x8 xh[16], xl[16];
    lt_layernorm<P_LN1G, P_LN1B>(X, pb, h, A.eps, xh, xl);
    lt_add_vec<P_BO>(X, pb);   // X = x + b_o: the out-projection slices add into it
    if (l == 0) LTS(3);
    // One head behind: while head i's q / k / v products run, the scores / softmax / PV of head i - 1 are their side work.
    //   D_i [Wq_i]:      S^T(i - 1) = K Q^T;  q(i) product   | mask, max, first exponentials of head i - 1
    //   A_i [Wk_i]:      k(i) product                         | remaining exponentials, sum, P -> hi | lo fragments;   then O(i - 1) = V^T P
    //   B_i [Wv_i]:      v(i) product (operands swapped)      | k(i) -> K fragments in LDS, q(i) -> fragments, O(i - 1) / sum -> fragments
    //   C_i [Wo_{i-1}]:  X += Wo[:, head i - 1] O(i - 1)      | v(i) -> V^T fragments in LDS
    // K / V^T of ONE head live in LDS: K(i) is written behind barrier B_i (the last reader of K(i - 1) is S^T(i - 1) in D_i), V^T(i) behind barrier C_i
    // (the last reader of V^T(i - 1) is the PV product at the end of A_i).
    f32x16 qa, ka, va, S[3], O;
    bf16x8 qh[2], ql[2], ph[3][2], pl[3][2], oh[2], ol[2];
    float mx0 = NEG, mx1 = NEG, mx = 0.f, sm = 0.f, rinv = 0.f;
    auto zero16 = [](f32x16& a) {
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = 0.f;
    };
    int hq = 0;   // head whose q / k / v products run (a runtime value inside the loop)
    // -- pieces of side work --
    auto mask_chunk = [&](int c) {   // registers 8 (c & 1) .. + 7 of score block c >> 1: keys of other videos -> NEG; running maxima
      const int kbi = c >> 1;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int r = 8 * (c & 1) + rr;
        const bool ok = (unsigned)(d0 + 32 * kbi + 8 * (r >> 2) + (r & 3)) < (unsigned)L;
        const float v = ok ? S[kbi][r] : NEG;
        S[kbi][r] = v;
        if (rr & 1) mx1 = fmaxf(mx1, v); else mx0 = fmaxf(mx0, v);
      }
    };
    auto exp_chunk = [&](int c) {
      const int kbi = c >> 1;
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int r = 8 * (c & 1) + rr;
#ifdef LT_EXPF
        const float e = __expf((S[kbi][r] - mx) * 0.6931471805599453f);
#else
        const float e = __builtin_amdgcn_exp2f(S[kbi][r] - mx);
#endif
        S[kbi][r] = e;
        if (rr & 1) s1 += e; else s0 += e;
      }
      sm += s0 + s1;
    };
    auto psplit_chunk = [&](int c) { split8(quad(S[c >> 1], 2 * (c & 1)), quad(S[c >> 1], 2 * (c & 1) + 1), ph[c >> 1][c & 1], pl[c >> 1][c & 1]); };
    auto kconv = [&](int s) {
      const float* bk = (const float*)pb + P_BQKV + 256 + 32 * hq;
      bf16x8 fh, fl;
      split8(quad(ka, 2 * s) + *(const f32x4*)(bk + 16 * s), quad(ka, 2 * s + 1) + *(const f32x4*)(bk + 16 * s + 8), fh, fl);
      *(bf16x8*)(kwr + (s * 2) * 4096) = fh;
      *(bf16x8*)(kwr + (s * 2 + 1) * 4096) = fl;
    };
    auto qconv = [&](int s) {
      const float* bq = (const float*)pb + P_BQKV + 32 * hq;
      split8((quad(qa, 2 * s) + *(const f32x4*)(bq + 16 * s)) * qscale, (quad(qa, 2 * s + 1) + *(const f32x4*)(bq + 16 * s + 8)) * qscale, qh[s], ql[s]);
    };
    auto vconv = [&](int s) {
      const float bv = P[P_BQKV + 512 + 32 * hq + n];
      bf16x8 fh, fl;
      split8(quad(va, 2 * s) + bv, quad(va, 2 * s + 1) + bv, fh, fl);
      *(bf16x8*)(vwr + (s * 2) * 1024) = fh;
      *(bf16x8*)(vwr + (s * 2 + 1) * 1024) = fl;
    };
    auto oconv = [&](int s) { split8(quad(O, 2 * s) * rinv, quad(O, 2 * s + 1) * rinv, oh[s], ol[s]); };
    // scores of the previous head: S^T[key][query] = K Q^T for the wave's three key blocks (q fragments of the previous B stage)
    auto scores = [&]() {
      bf16x8 kh[3][2], kl[3][2];
#pragma unroll
      for (int kbi = 0; kbi < 3; ++kbi)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          kh[kbi][s] = *(const bf16x8*)(krd[kbi] + (s * 2) * 4096);
          kl[kbi][s] = *(const bf16x8*)(krd[kbi] + (s * 2 + 1) * 4096);
        }
#pragma unroll
      for (int kbi = 0; kbi < 3; ++kbi) zero16(S[kbi]);
      // (the three blocks' accumulators in turn: no two consecutive MFMAs on one accumulator)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int kbi = 0; kbi < 3; ++kbi) { S[kbi] = LT_MFMA(kh[kbi][s], ql[s], S[kbi]); LT_PIN(); }
#pragma unroll
        for (int kbi = 0; kbi < 3; ++kbi) { S[kbi] = LT_MFMA(kl[kbi][s], qh[s], S[kbi]); LT_PIN(); }
#pragma unroll
        for (int kbi = 0; kbi < 3; ++kbi) { S[kbi] = LT_MFMA(kh[kbi][s], qh[s], S[kbi]); LT_PIN(); }
      }
      mx0 = mx1 = NEG;
      sm = 0.f;
    };
    auto pv = [&]() {
      f32x16 Op[3];
      bf16x8 vh[3][2], vlo[3][2];
#pragma unroll
      for (int kbi = 0; kbi < 3; ++kbi)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          vh[kbi][s] = *(const bf16x8*)(vrd[kbi] + (s * 2) * 1024);
          vlo[kbi][s] = *(const bf16x8*)(vrd[kbi] + (s * 2 + 1) * 1024);
        }
#pragma unroll
      for (int kbi = 0; kbi < 3; ++kbi) zero16(Op[kbi]);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int kbi = 0; kbi < 3; ++kbi) { Op[kbi] = LT_MFMA(vh[kbi][s], pl[kbi][s], Op[kbi]); LT_PIN(); }
#pragma unroll
        for (int kbi = 0; kbi < 3; ++kbi) { Op[kbi] = LT_MFMA(vlo[kbi][s], ph[kbi][s], Op[kbi]); LT_PIN(); }
#pragma unroll
        for (int kbi = 0; kbi < 3; ++kbi) { Op[kbi] = LT_MFMA(vh[kbi][s], ph[kbi][s], Op[kbi]); LT_PIN(); }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) O[r] = (Op[0][r] + Op[1][r]) + Op[2][r];
    };
    auto sideD = [&](int g) {
      if (g < 6) mask_chunk(g);
      if (g == 6) mx = lt_xmax(fmaxf(mx0, mx1), h);
      if (g == 7) exp_chunk(0);
    };
    auto sideA = [&](int g) {
      if (g == 0) { exp_chunk(1); exp_chunk(2); }
      if (g == 1) { exp_chunk(3); exp_chunk(4); }
      if (g == 2) { exp_chunk(5); rinv = 1.0f / lt_xsum(sm, h); psplit_chunk(0); }
      if (g == 3) { psplit_chunk(1); psplit_chunk(2); }
      if (g == 4) { psplit_chunk(3); psplit_chunk(4); }
      if (g == 5) psplit_chunk(5);
    };
    // ---- head 0: nothing behind it yet ----
    {
      const LtRing R = stage_begin();   // D_0
      lt_row_product<false, 0>(R, xh, xl, qa, LtNoSide{});
    }
    {
      const LtRing R = stage_begin();   // A_0
      lt_row_product<false, 0>(R, xh, xl, ka, LtNoSide{});
    }
    {
      const LtRing R = stage_begin();   // B_0
      lt_row_product<true, 5>(R, xh, xl, va, [&](int g) {
        if (g < 2) kconv(g);
        else if (g < 4) qconv(g - 2);
      });
      vconv(0);
      vconv(1);
    }
#pragma unroll 1
    for (hq = 1; hq < LT_NH; ++hq) {
      {
        const LtRing R = stage_begin();   // D_i
        scores();
          lt_row_product<false, 6>(R, xh, xl, qa, sideD);
      }
      {
        const LtRing R = stage_begin();   // A_i
          lt_row_product<false, 6>(R, xh, xl, ka, sideA);
        pv();
      }
      {
        const LtRing R = stage_begin();   // B_i
          lt_row_product<true, 5>(R, xh, xl, va, [&](int g) {
          if (g < 2) kconv(g);
          else if (g < 4) qconv(g - 2);
          else if (g < 6) oconv(g - 4);
        });
      }
      {
        const LtRing R = stage_begin();   // C_i
        lt_kslice_product<5>(R, oh, ol, X, [&](int g) {
          if (g < 2) vconv(g);
        });
      }
    }
    // ---- head 7's scores, softmax and PV have no product left to hide under ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // V^T(7) of every wave is in LDS
    asm volatile("" ::: "memory");
    scores();
#pragma unroll
    for (int g = 0; g < 8; ++g) sideD(g);
#pragma unroll
    for (int g = 0; g < 6; ++g) sideA(g);
    pv();
    oconv(0);
    oconv(1);
    {
      const LtRing R = stage_begin();   // C_8
      lt_kslice_product<0>(R, oh, ol, X, LtNoSide{});
    }
    if (l == 0) LTS(4);
#ifdef LT_STAMPS
    if (l == 0 && blockIdx.x == 0 && threadIdx.x == 0) { lt_ts[8] = c_wait; lt_ts[9] = c_bar; lt_ts[10] = c_work; }
#endif
    // ==================================================== FFN block ====================================================
    // X = x2; LN2 -> fragments; X += b2 becomes the accumulator of the second product.  Stage order W1_0, (W1_j, W2_{j-1}) for j = 1..31, W2_31: block
    // j - 1's bias / ReLU / hi | lo split is the side work of block j's first product.
    lt_layernorm<P_LN2G, P_LN2B>(X, pb, h, A.eps, xh, xl);
    lt_add_vec<P_B2>(X, pb);
    if (l == 0) LTS(5);
    f32x16 Ha, Hb;
    bf16x8 hh[2], hl[2];
    f32x4 b1q[4];   // lin1 bias of the block being converted, requested a stage ahead (a read inside the side work would hold its VALU back)
    auto b1_load = [&](int blk) {
      const float* b1 = (const float*)pb + P_B1 + 32 * blk;
#pragma unroll
      for (int g = 0; g < 4; ++g) b1q[g] = *(const f32x4*)(b1 + 8 * g);
    };
    auto hconv = [&](const f32x16& Hx, int s) {
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      const f32x4 u0 = __builtin_elementwise_max(quad(Hx, 2 * s) + b1q[2 * s], z4);
      const f32x4 u1 = __builtin_elementwise_max(quad(Hx, 2 * s + 1) + b1q[2 * s + 1], z4);
      split8(u0, u1, hh[s], hl[s]);
    };
    // one step = [W1_j: first product of block j into Hn | bias / ReLU / split of block j - 1 (Hp)] + [W2_{j-1}: second product of block j - 1]
    auto ffn_step = [&](f32x16& Hn, const f32x16& Hp, int j) {
      {
        const LtRing R = stage_begin();
        lt_row_product<false, 4>(R, xh, xl, Hn, [&](int g) {
          if (g >= 1 && g < 3) hconv(Hp, g - 1);
        });
        asm volatile("" : "+v"(hh[0]), "+v"(hh[1]), "+v"(hl[0]), "+v"(hl[1]));   // (the conversion belongs to THIS stage's MFMA stream)
      }
      {
        const LtRing R = stage_begin();
        b1_load(j);
        lt_kslice_product<0>(R, hh, hl, X, LtNoSide{});
      }
    };
    {
      const LtRing R = stage_begin();
      b1_load(0);
      lt_row_product<false, 0>(R, xh, xl, Ha, LtNoSide{});
    }
    ffn_step(Hb, Ha, 1);
#pragma unroll 1
    for (int j = 2; j < LT_F / 32; j += 2) {   // (two steps per trip: the accumulators trade places without a copy)
      ffn_step(Ha, Hb, j);
      ffn_step(Hb, Ha, j + 1);
    }
    hconv(Hb, 0);
    hconv(Hb, 1);
    {
      const LtRing R = stage_begin();
      lt_kslice_product<0>(R, hh, hl, X, LtNoSide{});
    }
    if (l == 0) LTS(6);
#ifdef LT_STAMPS
    if (l == 0 && blockIdx.x == 0 && threadIdx.x == 0) { lt_ts[11] = c_wait; lt_ts[12] = c_bar; lt_ts[13] = c_work; }
#endif
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-requested last stage: no LDS write may be pending when the workgroup leaves
  // ---- finished rows ----
  if (tl < nvalid) {
    float* yr = A.y + row * LT_D + 4 * h;
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
      for (int g = 0; g < 4; ++g) *(f32x4*)(yr + 32 * ob + 8 * g) = quad(X[ob], g);
  }
  LTS(7);
}

extern "C" int sf_debug_read_ts_layer_tok(long long* out16) {
  hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(lt_ts), sizeof(long long) * 16);
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" size_t sf_layer_tok_packed_bytes(void) { return LT_BLOB; }

// the layer's four matrices as fragments in consumption order + its eight vectors (all pointers of `w` device, torch layouts)
extern "C" int sf_pack_layer_tok_weights(const sf_tfm_layer* w, void* packed, int d_model, int num_heads, int ffn, void* stream) {
  SF_REQUIRE(w && packed, "sf_pack_layer_tok_weights: null pointer");
  SF_REQUIRE(w->in_proj_w && w->out_proj_w && w->lin1_w && w->lin2_w && w->norm1_g && w->norm1_b && w->in_proj_b && w->out_proj_b && w->norm2_g &&
                 w->norm2_b && w->lin1_b && w->lin2_b, "sf_pack_layer_tok_weights: null weight");
  SF_REQUIRE(d_model == LT_D && num_heads == LT_NH && ffn == LT_F, "sf_pack_layer_tok_weights: d_model 256, 8 heads, ffn 1024 only");
  hipStream_t st = (hipStream_t)stream;
  const int total = LT_NST * 32 * 64;
  hipLaunchKernelGGL(pack_layer_tok_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w->in_proj_w, w->out_proj_w, w->lin1_w, w->lin2_w, (uint4*)packed);
  SF_CHECK_LAUNCH();
  float* vec = (float*)((char*)packed + (size_t)LT_NST * LT_STAGE);
  const struct {
    const float* src;
    int off, n;
  } parts[8] = {{w->norm1_g, P_LN1G, 256}, {w->norm1_b, P_LN1B, 256}, {w->in_proj_b, P_BQKV, 768}, {w->out_proj_b, P_BO, 256},
                {w->norm2_g, P_LN2G, 256}, {w->norm2_b, P_LN2B, 256}, {w->lin1_b, P_B1, 1024},   {w->lin2_b, P_B2, 256}};
  for (const auto& p : parts) {
    const hipError_t e = hipMemcpyAsync(vec + p.off, p.src, (size_t)p.n * 4, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return sf_set_err((int)e, hipGetErrorString(e), __FILE__, __LINE__);
  }
  return 0;
}

// Videos per 128-token workgroup for sequences of L tokens: the largest count of WHOLE videos for which the keys of the videos a wave's 32 rows belong to
// lie inside three 32-key blocks (42 tokens: 3; 50 tokens: 1 -- two would put the keys of wave 1's rows in four blocks; 65..96 tokens, the reference's
// Physion window of 15 frames x 6 slots: 1, the last wave(s) idle); 0: the sequence does not fit a workgroup
int sf_layer_tok_vpw(int L) {
  if (L < 1 || L > 96) return 0;
  for (int vpw = LT_TOK / L; vpw >= 1; --vpw) {
    const int nvalid = vpw * L;
    bool ok = true;
    for (int w = 0; w < 4 && ok; ++w) {
      const int wf = w * 32 < nvalid ? w * 32 : nvalid - 1, wl = w * 32 + 31 < nvalid ? w * 32 + 31 : nvalid - 1;
      const int kb0 = ((wf / L) * L) >> 5, kb1 = (((wl / L) * L) + L - 1) >> 5;
      ok = kb1 - kb0 + 1 <= 3;
    }
    if (ok) return vpw;
  }
  return 0;
}
bool sf_layer_tok_ok(int L) { return sf_layer_tok_vpw(L) > 0; }

// `nl` consecutive layers in ONE launch.  mode 0: xin [B * L][256] rows;  mode 1: the first of them is layer 0 of a rollout step --
// x = ring[b][(f0 + r / nslots) % ring_frames][r % nslots] + pe[r].  y [B * L][256] finished rows of the last of them.
int sf_layer_tok_ex(int mode, const float* xin, const float* ring, int ring_frames, int nslots, int f0, const float* pe, const sf_tfm_layer* layers, int nl,
                    float eps, float* y, int B, int L, hipStream_t st) {
  bool ok = layers && nl >= 1 && nl <= LT_MAXL && sf_layer_tok_ok(L) && B >= 1 && y && (mode == 0 ? xin != nullptr : (ring && pe && nslots >= 1 && ring_frames >= 1));
  for (int l = 0; ok && l < nl; ++l) ok = layers[l].tok_packed != nullptr;
  if (!ok)
    return sf_set_err(-1, "invalid argument: the token-stationary layers need sf_pack_layer_tok_weights fragments, 1..8 layers and 1 <= L <= 96 rows per video", __FILE__, __LINE__);
  static const int dbg = sf_dbg("lt");
  LtArgs A;
  A.x = xin; A.ring = ring; A.pe = pe; A.y = y;
  for (int l = 0; l < LT_MAXL; ++l) A.blob[l] = (const char*)layers[l < nl ? l : nl - 1].tok_packed;
  A.eps = eps; A.nl = nl; A.B = B; A.L = L; A.vpw = sf_layer_tok_vpw(L); A.RF = ring_frames; A.N = nslots; A.f0 = f0; A.dbg_ts = dbg;
  const int nwg = (B + A.vpw - 1) / A.vpw;
  const double flops = nl * ((double)B * L * (2.0 * LT_D * (3 * LT_D + LT_D + 2 * LT_F)) + (double)B * LT_NH * 4.0 * L * L * 32);
  if (mode == 0) {
    SF_TRY(sf_ensure_dyn_lds((const void*)layer_tok_kernel<0>, LT_LDS));
    sf_prof_begin(SF_K_LAYER_TOK, st, flops);
    hipLaunchKernelGGL(layer_tok_kernel<0>, dim3(nwg), dim3(LT_NT), LT_LDS, st, A);
  } else {
    SF_TRY(sf_ensure_dyn_lds((const void*)layer_tok_kernel<1>, LT_LDS));
    sf_prof_begin(SF_K_LAYER_TOK, st, flops);
    hipLaunchKernelGGL(layer_tok_kernel<1>, dim3(nwg), dim3(LT_NT), LT_LDS, st, A);
  }
  sf_prof_end(SF_K_LAYER_TOK, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// Kernel-level entry point (include/slotformer_hip.h): `nl` whole layers (w[0..nl)) on B sequences of L tokens, for tests against a plain reference
extern "C" int sf_layer_tok_block_f32(const sf_tfm_layer* w, int nl, const float* x, float* y, int B, int L, void* stream) {
  SF_REQUIRE(w && x && y && B > 0 && nl >= 1 && nl <= LT_MAXL, "sf_layer_tok_block_f32: null pointer / empty problem / more than 8 layers");
  SF_REQUIRE(sf_get_precision() == 1, "sf_layer_tok_block_f32: split-bf16 mode only");
  return sf_layer_tok_ex(0, x, nullptr, 1, 1, 0, nullptr, w, nl, 1e-5f, y, B, L, (hipStream_t)stream);
}
