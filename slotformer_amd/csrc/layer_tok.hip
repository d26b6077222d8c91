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
constexpr int LT_NT = 256, LT_D = 256, LT_F = 1024, LT_NH = 8, LT_TOK = 128;
constexpr int LT_STAGE = 32 * 1024, LT_RING = 3;
constexpr int LT_NST_ATT = 4 * LT_NH, LT_NST = LT_NST_ATT + 2 * (LT_F / 32);   // 32 attention + 64 FFN stages
constexpr int LT_KV = LT_RING * LT_STAGE;     // K fragments [s][plane][token block][64 lanes] x 16 B, then V^T fragments [token block][s][plane][64 lanes] x 16 B
constexpr int LT_VT = LT_KV + 16 * 1024;
constexpr int LT_PAR = LT_KV + 32 * 1024;     // the layer's vectors (f32)
constexpr int P_LN1G = 0, P_LN1B = 256, P_BQKV = 512, P_BO = 1280, P_LN2G = 1536, P_LN2B = 1792, P_B1 = 2048, P_B2 = 3072, P_N = 3328;
constexpr size_t LT_LDS = (size_t)LT_PAR + (size_t)P_N * 4;
static_assert(LT_LDS <= 160 * 1024, "LDS budget");

struct LtArgs {
  const float* x;      // MODE 0: [B * L][256] rows
  const float* ring;   // MODE 1: projection ring [B][RF][N][256] ...
  const float* pe;     //         ... + position table [L][256]
  float* y;            // [B * L][256]
  const char* wp;      // sf_pack_layer_tok_weights
  const float *ln1g, *ln1b, *bqkv, *bo, *ln2g, *ln2b, *b1, *b2;
  float eps;
  int B, L, vpw, RF, N, f0;
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

// the ring as one wave sees it during a stage: `rd` = this lane's read address of fragment 0 of the current stage, (`src`, `dst`) = where piece 0 of the
// stage two ahead comes from (per lane) / goes to (wave-uniform); a product issues one piece per fragment group
struct LtRing {
  const char* rd;
  const char* src;
  char* dst;
};
__device__ __forceinline__ void lt_dma(const LtRing& R, int f) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(R.src + f * 1024),
                                   (void __attribute__((address_space(3)))*)(R.dst + f * 1024), 16, 0, 0);
}
#define LT_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// D^T[32 features][32 tokens] (SW: D[32 tokens][32 features]) = W block . A^T over the 256 input channels: fragment group g = (hi, lo) of the virtual
// k-steps 2 g and 2 g + 1; even k-steps into a0, odd into a1.  The reads of group g + 1 go behind the first MFMA of group g (ffn_tok.hip).
template <bool SW>
__device__ __forceinline__ void lt_row_product(const LtRing& R, const bf16x8 (&xh)[16], const bf16x8 (&xl)[16], f32x16& a0, f32x16& a1) {
  bf16x8 wb[2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wb[0][i] = *(const bf16x8*)(R.rd + i * 1024);
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int b = g & 1, ks = 2 * g;
    if constexpr (!SW) a0 = LT_MFMA(wb[b][0], xl[ks], a0); else a0 = LT_MFMA(xl[ks], wb[b][0], a0);
    if (g + 1 < 8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) wb[b ^ 1][i] = *(const bf16x8*)(R.rd + (4 * (g + 1) + i) * 1024);
    }
    lt_dma(R, g);
    if constexpr (!SW) {
      a1 = LT_MFMA(wb[b][2], xl[ks + 1], a1);
      a0 = LT_MFMA(wb[b][1], xh[ks], a0);
      a1 = LT_MFMA(wb[b][3], xh[ks + 1], a1);
      a0 = LT_MFMA(wb[b][0], xh[ks], a0);
      a1 = LT_MFMA(wb[b][2], xh[ks + 1], a1);
    } else {
      a1 = LT_MFMA(xl[ks + 1], wb[b][2], a1);
      a0 = LT_MFMA(xh[ks], wb[b][1], a0);
      a1 = LT_MFMA(xh[ks + 1], wb[b][3], a1);
      a0 = LT_MFMA(xh[ks], wb[b][0], a0);
      a1 = LT_MFMA(xh[ks + 1], wb[b][2], a1);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (g + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// Y^T[256 features][32 tokens] += W[:, 32-wide slice] . I^T: fragment group ob = (hi, lo) of the slice's two virtual k-steps for output block ob
__device__ __forceinline__ void lt_kslice_product(const LtRing& R, const bf16x8 (&ih)[2], const bf16x8 (&il)[2], f32x16 (&Y)[8]) {
  bf16x8 wb[2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wb[0][i] = *(const bf16x8*)(R.rd + i * 1024);
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int b = g & 1;
    Y[g] = LT_MFMA(wb[b][0], il[0], Y[g]);
    if (g + 1 < 8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) wb[b ^ 1][i] = *(const bf16x8*)(R.rd + (4 * (g + 1) + i) * 1024);
    }
    lt_dma(R, g);
    Y[g] = LT_MFMA(wb[b][1], ih[0], Y[g]);
    Y[g] = LT_MFMA(wb[b][0], ih[0], Y[g]);
    Y[g] = LT_MFMA(wb[b][2], il[1], Y[g]);
    Y[g] = LT_MFMA(wb[b][3], ih[1], Y[g]);
    Y[g] = LT_MFMA(wb[b][2], ih[1], Y[g]);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (g + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// LayerNorm over the 256 features of a token held as X[ob][4 g + q] = feature 32 ob + 8 g + 4 h + q by lanes (token, h = 0 / 1); gamma / beta in LDS at
// float offsets GOFF / BOFF behind pb (= the vector block + 4 h floats, an opaque per-lane base: every read is base + immediate):
// the normalised row as the 16 hi | lo fragments of the virtual k-steps (block ob, s): registers 8 s .. 8 s + 7
template <int GOFF, int BOFF>
__device__ __forceinline__ void lt_layernorm(const f32x16 (&X)[8], const char* pb, float eps, bf16x8 (&xh)[16], bf16x8 (&xl)[16]) {
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
  float s = (s0 + s1) + (s2 + s3);
  s += __shfl_xor(s, 32, 64);
  const float mu = s * (1.0f / LT_D);
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
  float q = (q0 + q1) + (q2 + q3);
  q += __shfl_xor(q, 32, 64);
  const float rs = 1.0f / sqrtf(q * (1.0f / LT_D) + eps);
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

// ---- weights of one layer -> 96 stages x 32 fragments x 64 lanes x 16 B, in consumption order -------------------------------------------------------
//   stage 4 hd + 0 / 1 / 2: rows 256 + 32 hd (k), 512 + 32 hd (v), 32 hd (q) of in_proj_w [768][256]: fragment f = 2 vk + plane, vk = (input block ib, s)
//   stage 4 hd + 3:         columns 32 hd .. + 31 of out_proj_w [256][256]: fragment f = 4 ob + 2 s + plane (output block ob)
//   stage 32 + 2 hb:        rows 32 hb .. + 31 of lin1_w [1024][256] (as the in_proj stages)
//   stage 33 + 2 hb:        columns 32 hb .. + 31 of lin2_w [256][1024] (as the out_proj stages)
//   element j of lane (i, h) of a fragment over columns c0 .. c0 + 31 at k-step s:  W[r0 + i][c0 + 8 (2 s + (j >> 2)) + 4 h + (j & 3)]
//   (the k order of an accumulator: register 8 s + j of lane (token, h) holds feature 8 (2 s + (j >> 2)) + 4 h + (j & 3) of its 32-block)
__global__ void pack_layer_tok_kernel(const float* __restrict__ win, const float* __restrict__ wout, const float* __restrict__ w1,
                                      const float* __restrict__ w2, uint4* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= LT_NST * 32 * 64) return;
  const int lane = idx & 63, f = (idx >> 6) & 31, st = idx >> 11;
  const int pl = f & 1, h = lane >> 5, i = lane & 31;
  const float* W;
  int ld, r0, c0, s;
  bool slice;
  if (st < LT_NST_ATT) {
    const int hd = st >> 2, kind = st & 3;
    slice = kind == 3;
    if (!slice) {
      W = win; ld = LT_D; r0 = (kind == 0 ? 256 : kind == 1 ? 512 : 0) + 32 * hd;
    } else {
      W = wout; ld = LT_D; c0 = 32 * hd;
    }
  } else {
    const int hb = (st - LT_NST_ATT) >> 1;
    slice = (st - LT_NST_ATT) & 1;
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

__device__ long long lt_ts[16];   // wall-clock stamps of workgroup 0, wave 0 (SF_LT_DBG=1; sf_debug_read_ts_layer_tok)
#define LTS(i) do { if (A.dbg_ts && blockIdx.x == 0 && threadIdx.x == 0) lt_ts[i] = wall_clock64(); } while (0)

namespace {
struct LtArgsK : LtArgs {
  int dbg_ts;
};
}  // namespace

template <int MODE>
__global__ __launch_bounds__(LT_NT) void layer_tok_kernel(LtArgsK A) {
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
  // ---- the layer's vectors -> LDS ----
  for (int i = t; i < P_N / 4; i += LT_NT) {
    const float* src = i < 64 ? A.ln1g + 4 * i : i < 128 ? A.ln1b + 4 * (i - 64) : i < 320 ? A.bqkv + 4 * (i - 128) : i < 384 ? A.bo + 4 * (i - 320)
                     : i < 448 ? A.ln2g + 4 * (i - 384) : i < 512 ? A.ln2b + 4 * (i - 448) : i < 768 ? A.b1 + 4 * (i - 512) : A.b2 + 4 * (i - 768);
    *(f32x4*)(P + 4 * i) = *(const f32x4*)src;
  }
  // ---- weight ring: stage st lives in buffer st % 3; this wave copies fragments 8 wave .. 8 wave + 7 of every stage ----
  const char* wsrc = A.wp + (size_t)(wave * 8) * 1024 + lane * 16;
  char* const wdst = smem + (wave * 8) * 1024;
  int st = 0, cur = 0;   // current stage and its buffer
  {
    LtRing R0{nullptr, wsrc, wdst}, R1{nullptr, wsrc + LT_STAGE, wdst + LT_STAGE};
#pragma unroll
    for (int f = 0; f < 8; ++f) lt_dma(R0, f);
#pragma unroll
    for (int f = 0; f < 8; ++f) lt_dma(R1, f);
  }
  // the stage about to be consumed has landed (every wave waits for its own pieces, then the barrier), every wave is done with the previous one
  // (whose buffer the pieces of stage st + 2 go to), LDS writes of the previous stage (keys / values) are visible
  auto stage_begin = [&]() -> LtRing {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int nst = min(st + 2, LT_NST - 1), nb = cur == 0 ? 2 : cur - 1;   // (the last two iterations re-request the last stage into a free buffer)
    LtRing R{smem + cur * LT_STAGE + lane * 16, wsrc + (size_t)nst * LT_STAGE, wdst + nb * LT_STAGE};
    ++st;
    cur = cur == 2 ? 0 : cur + 1;
    return R;
  };
  __syncthreads();   // the vectors (this also drains the first two stages: only here)
  LTS(1);
  // ---- key blocks this wave's queries need: the tokens of the videos its 32 rows belong to (at most three 32-key blocks, sf_layer_tok_ok) ----
  const int wf = min(wave * 32, nvalid - 1), wl = min(wave * 32 + 31, nvalid - 1);
  const int kb0 = ((wf / L) * L) >> 5, nkb = ((((wl / L) * L) + L - 1) >> 5) - kb0 + 1;
  const int d0 = 32 * kb0 + 4 * h - vl * L;   // key index of register (kbi, 4 g + q) minus the video's first key: d0 + 32 kbi + 8 g + q
  // ---- attention block ----
  bf16x8 xh[16], xl[16];
  const char* pb;   // the vectors as this lane reads them: + 4 h floats; opaque, so that every read is this base + an immediate offset
  {
    unsigned pbo = (unsigned)(LT_PAR + 16 * h);
    asm volatile("" : "+v"(pbo));
    pb = smem + pbo;
  }
  lt_layernorm<P_LN1G, P_LN1B>(X, pb, A.eps, xh, xl);
  lt_add_vec<P_BO>(X, pb);
  LTS(2);
  const float scale = 0.17677669529663687f;   // 1 / sqrt(32)
  char* const kwr = smem + LT_KV + wave * 1024 + lane * 16;          // + (s * 2 + plane) * 4096
  char* const vwr = smem + LT_VT + wave * 4096 + lane * 16;          // + (s * 2 + plane) * 1024
  const char* const krd = smem + LT_KV + lane * 16;                  // + (s * 2 + plane) * 4096 + kb * 1024
  const char* const vrd = smem + LT_VT + lane * 16;                  // + kb * 4096 + (s * 2 + plane) * 1024
#pragma unroll 1
  for (int hd = 0; hd < LT_NH; ++hd) {
    f32x16 a0, a1;
    // -- k of head hd: the accumulators ARE the A fragments of S^T = K Q^T (lane = key, registers = dims) --
    {
      const LtRing R = stage_begin();
#pragma unroll
      for (int r = 0; r < 16; ++r) a0[r] = a1[r] = 0.f;
      lt_row_product<false>(R, xh, xl, a0, a1);
      const float* bk = (const float*)pb + P_BQKV + 256 + 32 * hd;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 fh, fl;
        split8(quad(a0, 2 * s) + quad(a1, 2 * s) + *(const f32x4*)(bk + 16 * s), quad(a0, 2 * s + 1) + quad(a1, 2 * s + 1) + *(const f32x4*)(bk + 16 * s + 8), fh, fl);
        *(bf16x8*)(kwr + (s * 2) * 4096) = fh;
        *(bf16x8*)(kwr + (s * 2 + 1) * 4096) = fl;
      }
    }
    // -- v of head hd with the operands swapped: lane = dim, registers = this wave's tokens -> V^T fragments --
    {
      const LtRing R = stage_begin();
#pragma unroll
      for (int r = 0; r < 16; ++r) a0[r] = a1[r] = 0.f;
      lt_row_product<true>(R, xh, xl, a0, a1);
      const float bv = P[P_BQKV + 512 + 32 * hd + n];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 fh, fl;
        split8(quad(a0, 2 * s) + quad(a1, 2 * s) + bv, quad(a0, 2 * s + 1) + quad(a1, 2 * s + 1) + bv, fh, fl);
        *(bf16x8*)(vwr + (s * 2) * 1024) = fh;
        *(bf16x8*)(vwr + (s * 2 + 1) * 1024) = fl;
      }
    }
    // -- q of head hd (scaled): B fragments of S^T --
    bf16x8 qh[2], ql[2];
    {
      const LtRing R = stage_begin();
#pragma unroll
      for (int r = 0; r < 16; ++r) a0[r] = a1[r] = 0.f;
      lt_row_product<false>(R, xh, xl, a0, a1);
      const float* bq = (const float*)pb + P_BQKV + 32 * hd;
#pragma unroll
      for (int s = 0; s < 2; ++s)
        split8((quad(a0, 2 * s) + quad(a1, 2 * s) + *(const f32x4*)(bq + 16 * s)) * scale,
               (quad(a0, 2 * s + 1) + quad(a1, 2 * s + 1) + *(const f32x4*)(bq + 16 * s + 8)) * scale, qh[s], ql[s]);
    }
    // -- scores, softmax, PV, out-projection slice (behind this stage's barrier every wave's keys and values of head hd are in LDS) --
    {
      const LtRing R = stage_begin();
      f32x16 S[3];
#pragma unroll
      for (int kbi = 0; kbi < 3; ++kbi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) S[kbi][r] = 0.f;
        if (kbi < nkb) {
          const char* kp = krd + (kb0 + kbi) * 1024;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const bf16x8 kh = *(const bf16x8*)(kp + (s * 2) * 4096), kl = *(const bf16x8*)(kp + (s * 2 + 1) * 4096);
            S[kbi] = LT_MFMA(kh, ql[s], S[kbi]);
            S[kbi] = LT_MFMA(kl, qh[s], S[kbi]);
            S[kbi] = LT_MFMA(kh, qh[s], S[kbi]);
          }
        }
      }
      // keys of other videos (and key blocks this wave skipped) get -3e38: exp -> 0
      float mx = -3.0e38f;
#pragma unroll
      for (int kbi = 0; kbi < 3; ++kbi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool ok = kbi < nkb && (unsigned)(d0 + 32 * kbi + 8 * (r >> 2) + (r & 3)) < (unsigned)L;
          S[kbi][r] = ok ? S[kbi][r] : -3.0e38f;
          mx = fmaxf(mx, S[kbi][r]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int kbi = 0; kbi < 3; ++kbi) {
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          S[kbi][r] = __expf(S[kbi][r] - mx);
          s4[r & 3] += S[kbi][r];
        }
        sum += (s4[0] + s4[1]) + (s4[2] + s4[3]);
      }
      sum += __shfl_xor(sum, 32, 64);
      const float rinv = 1.0f / sum;
      f32x16 O;
#pragma unroll
      for (int r = 0; r < 16; ++r) O[r] = 0.f;
#pragma unroll
      for (int kbi = 0; kbi < 3; ++kbi) {
        if (kbi < nkb) {
          const char* vp = vrd + (kb0 + kbi) * 4096;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            bf16x8 ph, pl;
            split8(quad(S[kbi], 2 * s), quad(S[kbi], 2 * s + 1), ph, pl);
            const bf16x8 vh = *(const bf16x8*)(vp + (s * 2) * 1024), vlo = *(const bf16x8*)(vp + (s * 2 + 1) * 1024);
            O = LT_MFMA(vh, pl, O);
            O = LT_MFMA(vlo, ph, O);
            O = LT_MFMA(vh, ph, O);
          }
        }
      }
      bf16x8 oh[2], ol[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) split8(quad(O, 2 * s) * rinv, quad(O, 2 * s + 1) * rinv, oh[s], ol[s]);
      lt_kslice_product(R, oh, ol, X);
    }
  }
  LTS(3);
  // ---- FFN block: X = x2 (+ b_o added above); LN2 -> fragments; X += b2 becomes the accumulator of the second product ----
  lt_layernorm<P_LN2G, P_LN2B>(X, pb, A.eps, xh, xl);
  lt_add_vec<P_B2>(X, pb);
  LTS(4);
#pragma unroll 1
  for (int hb = 0; hb < LT_F / 32; ++hb) {
    f32x16 a0, a1;
    bf16x8 hh[2], hl[2];
    {
      const LtRing R = stage_begin();
#pragma unroll
      for (int r = 0; r < 16; ++r) a0[r] = a1[r] = 0.f;
      lt_row_product<false>(R, xh, xl, a0, a1);
      const float* b1 = (const float*)pb + P_B1 + 32 * hb;
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f32x4 u0 = quad(a0, 2 * s) + quad(a1, 2 * s) + *(const f32x4*)(b1 + 16 * s);
        f32x4 u1 = quad(a0, 2 * s + 1) + quad(a1, 2 * s + 1) + *(const f32x4*)(b1 + 16 * s + 8);
        u0 = __builtin_elementwise_max(u0, z4);
        u1 = __builtin_elementwise_max(u1, z4);
        split8(u0, u1, hh[s], hl[s]);
      }
    }
    {
      const LtRing R = stage_begin();
      lt_kslice_product(R, hh, hl, X);
    }
  }
  LTS(5);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-requested last stage: no LDS write may be pending when the workgroup leaves
  // ---- finished rows ----
  if (tl < nvalid) {
    float* yr = A.y + row * LT_D + 4 * h;
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
      for (int g = 0; g < 4; ++g) *(f32x4*)(yr + 32 * ob + 8 * g) = quad(X[ob], g);
  }
  LTS(6);
}

extern "C" int sf_debug_read_ts_layer_tok(long long* out16) {
  hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(lt_ts), sizeof(long long) * 16);
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" size_t sf_layer_tok_packed_bytes(void) { return (size_t)LT_NST * LT_STAGE; }

extern "C" int sf_pack_layer_tok_weights(const float* in_proj_w, const float* out_proj_w, const float* lin1_w, const float* lin2_w, void* packed,
                                         int d_model, int num_heads, int ffn, void* stream) {
  SF_REQUIRE(in_proj_w && out_proj_w && lin1_w && lin2_w && packed, "sf_pack_layer_tok_weights: null pointer");
  SF_REQUIRE(d_model == LT_D && num_heads == LT_NH && ffn == LT_F, "sf_pack_layer_tok_weights: d_model 256, 8 heads, ffn 1024 only");
  const int total = LT_NST * 32 * 64;
  hipLaunchKernelGGL(pack_layer_tok_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, in_proj_w, out_proj_w, lin1_w, lin2_w,
                     (uint4*)packed);
  SF_CHECK_LAUNCH();
  return 0;
}

// Rows per video the token-stationary layer takes: whole videos per 128-token workgroup, and the keys of the videos a wave's 32 rows belong to inside
// three 32-key blocks
bool sf_layer_tok_ok(int L) {
  if (L < 1 || L > 64) return false;
  const int vpw = LT_TOK / L, nvalid = vpw * L;
  for (int w = 0; w < 4; ++w) {
    const int wf = w * 32 < nvalid ? w * 32 : nvalid - 1, wl = w * 32 + 31 < nvalid ? w * 32 + 31 : nvalid - 1;
    const int kb0 = ((wf / L) * L) >> 5, kb1 = (((wl / L) * L) + L - 1) >> 5;
    if (kb1 - kb0 + 1 > 3) return false;
  }
  return true;
}

// mode 0: xin [B * L][256] rows;  mode 1: layer 0 of a rollout step -- x = ring[b][(f0 + r / nslots) % ring_frames][r % nslots] + pe[r].
// y [B * L][256] finished rows of the whole layer.
int sf_layer_tok_ex(int mode, const float* xin, const float* ring, int ring_frames, int nslots, int f0, const float* pe, const sf_tfm_layer& w,
                    float eps, float* y, int B, int L, hipStream_t st) {
  if (!w.tok_packed || !sf_layer_tok_ok(L) || B < 1 || !y || (mode == 0 ? !xin : (!ring || !pe || nslots < 1 || ring_frames < 1)))
    return sf_set_err(-1, "invalid argument: the token-stationary layer needs sf_pack_layer_tok_weights fragments and 1 <= L <= 64 rows per video", __FILE__, __LINE__);
  static const int dbg = getenv("SF_LT_DBG") ? atoi(getenv("SF_LT_DBG")) : 0;
  LtArgsK A;
  A.x = xin; A.ring = ring; A.pe = pe; A.y = y; A.wp = (const char*)w.tok_packed;
  A.ln1g = w.norm1_g; A.ln1b = w.norm1_b; A.bqkv = w.in_proj_b; A.bo = w.out_proj_b; A.ln2g = w.norm2_g; A.ln2b = w.norm2_b; A.b1 = w.lin1_b; A.b2 = w.lin2_b;
  A.eps = eps; A.B = B; A.L = L; A.vpw = LT_TOK / L; A.RF = ring_frames; A.N = nslots; A.f0 = f0; A.dbg_ts = dbg;
  const int nwg = (B + A.vpw - 1) / A.vpw;
  const double flops = (double)B * L * (2.0 * LT_D * (3 * LT_D + LT_D + 2 * LT_F)) + (double)B * LT_NH * 4.0 * L * L * 32;
  if (mode == 0) {
    SF_TRY(sf_ensure_dyn_lds((const void*)layer_tok_kernel<0>, LT_LDS));
    sf_prof_begin(SF_K_FFN, st, flops);
    hipLaunchKernelGGL(layer_tok_kernel<0>, dim3(nwg), dim3(LT_NT), LT_LDS, st, A);
  } else {
    SF_TRY(sf_ensure_dyn_lds((const void*)layer_tok_kernel<1>, LT_LDS));
    sf_prof_begin(SF_K_FFN, st, flops);
    hipLaunchKernelGGL(layer_tok_kernel<1>, dim3(nwg), dim3(LT_NT), LT_LDS, st, A);
  }
  sf_prof_end(SF_K_FFN, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// Kernel-level entry point (include/slotformer_hip.h): one whole layer on B sequences of L tokens, for tests against a plain reference
extern "C" int sf_layer_tok_block_f32(const sf_tfm_layer* w, const float* x, float* y, int B, int L, void* stream) {
  SF_REQUIRE(w && x && y && B > 0, "sf_layer_tok_block_f32: null pointer / empty problem");
  SF_REQUIRE(w->norm1_g && w->norm1_b && w->in_proj_b && w->out_proj_b && w->norm2_g && w->norm2_b && w->lin1_b && w->lin2_b && w->tok_packed,
             "sf_layer_tok_block_f32: null weight (sf_pack_layer_tok_weights fragments needed)");
  SF_REQUIRE(sf_get_precision() == 1, "sf_layer_tok_block_f32: split-bf16 mode only");
  return sf_layer_tok_ex(0, x, nullptr, 1, 1, 0, nullptr, *w, 1e-5f, y, B, L, (hipStream_t)stream);
}
