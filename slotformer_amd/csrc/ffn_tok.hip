// FFN block of a rollout layer in a TOKEN-STATIONARY form (experimental; tools/ffn_tok_probe.py):
//     y = x2 + lin2(relu(lin1(LN2(x2))))            (nn.TransformerEncoderLayer, norm_first; slotformer.py:72-80)
// ffn_tile.hip gives the eight waves of a 64-row tile 32 hidden columns each, so every 256-wide hidden chunk is an all-to-all through LDS planes
// behind two workgroup barriers, and both waves of a SIMD sit in the same phase at the same time (matrix pipe 56 % busy, profiles/r04_sq_counters.txt).
// Here a wave owns 32 TOKENS for the whole block:
//   * H^T[block] = W1[block] . LN2(x)^T  -- weights as the A operand, the wave's LN2(x) fragments (B operand) resident in registers;
//   * the accumulator layout of H^T (lane = (token, row group), rows 8 g + 4 (lane >> 5) + q) IS a B operand of the second product once W2's fragments
//     are packed with the matching permuted k order: bias, ReLU and the hi / lo split are register arithmetic -- no plane, no barrier;
//   * Y^T += W2[:, block] . H^T[block] into eight resident accumulators.
// Waves share WEIGHTS instead of activations: the fragments of hidden block b (32 KB of W1 rows + 32 KB of W2 columns, packed contiguously by
// sf_pack_ffn_tok_weights) go global -> LDS once per workgroup (global_load_lds, 16 B per lane: a fragment is lane-linear) into a two-stage ring,
// one barrier per block.  One wave per SIMD (four waves = 128 tokens per workgroup), up to 512 registers per lane.
// The sum over the hidden dimension runs through ONE accumulator per output block (ffn_tile.hip: four chunk partials, summed): the results differ
// in the last bits.
#include "../../include/slotformer_hip.h"
#include "sf_internal.h"
#include "layer_fused.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int TK_NT = 256, TK_D = 256, TK_F = 1024, TK_NHB = TK_F / 32, TK_WGT = 128;   // threads, widths, hidden blocks, tokens per workgroup
constexpr int TK_STAGE = 64 * 1024;                                                      // bytes of a stage: 32 + 32 fragments of 1 KB
constexpr int TK_NST = TK_NHB + 1;                                                       // stage j = W1 block j (j < 32) + W2 block j - 1 (j > 0)
constexpr int TK_RP = 1024 + 16;                                                         // byte pitch of a staged f32 row (bank-spread for the 16-byte column pieces)
constexpr size_t TK_ROWS_LDS = (size_t)TK_WGT * TK_RP;                                   // the workgroup's rows staged whole (prologue: x2, epilogue: y) over the ring
constexpr size_t TK_LDS = TK_ROWS_LDS + (size_t)(TK_F + 32) * 4;                         // ring (128 KB) / staged rows (130 KB) + lin1 bias (+ 32 zeros for the idle block)
static_assert(TK_ROWS_LDS >= (size_t)2 * TK_STAGE && TK_LDS <= 160 * 1024, "LDS budget");

struct TokArgs {
  const float* x2;
  const float *ln_g, *ln_b;
  float ln_eps;
  const uint4* wp;   // sf_pack_ffn_tok_weights
  const float *b1, *b2;
  float* y;
  int M, dbg;
};

__device__ __forceinline__ bf16x8 cat8(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
}  // namespace

// lin1_w [1024][256], lin2_w [256][1024] (torch layout) -> per hidden block hb 64 fragments of 64 lanes x 16 B:
//   (stage st = 0..32 holds W1 block hb = st in its first half and W2 block hb = st - 1 in its second: the first product of block b + 1 and the second
//    product of block b run in the same loop iteration; the two idle halves are zeros)
//   f < 32:  W1, k-step ks = f >> 1, plane f & 1:        element j = lin1_w[32 hb + (lane & 31)][16 ks + 8 (lane >> 5) + j]
//   f >= 32: W2, output block ob = (f - 32) >> 2, k-step s = ((f - 32) >> 1) & 1, plane f & 1:
//                                                         element j = lin2_w[32 ob + (lane & 31)][32 hb + 8 (2 s + (j >> 2)) + 4 (lane >> 5) + (j & 3)]
//   (the k order of an accumulator: register 8 s + j of lane (token, h) holds hidden row 8 (2 s + (j >> 2)) + 4 h + (j & 3))
__global__ void pack_ffn_tok_kernel(const float* __restrict__ w1, const float* __restrict__ w2, uint4* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= TK_NST * 64 * 64) return;
  const int lane = idx & 63, f = (idx >> 6) & 63, st = idx >> 12;
  const int pl = f & 1, h = lane >> 5;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
  if (f < 32) {
    if (st < TK_NHB) {   // W1 block st (the last stage's first half is idle: zeros)
      const int ks = f >> 1;
      const float* src = w1 + (long long)(32 * st + (lane & 31)) * TK_D + 16 * ks + 8 * h;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[j];
    }
  } else if (st > 0) {   // W2 block st - 1 (the first stage's second half is idle: zeros)
    const int ob = (f - 32) >> 2, s = ((f - 32) >> 1) & 1;
    const float* src = w2 + (long long)(32 * ob + (lane & 31)) * TK_F + 32 * (st - 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[8 * (2 * s + (j >> 2)) + 4 * h + (j & 3)];
  }
  union {
    __bf16 b[8];
    uint4 u;
  } o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 hi = (__bf16)v[j];
    o.b[j] = pl ? (__bf16)(v[j] - (float)hi) : hi;
  }
  out[idx] = o.u;
}

__device__ long long tk_ts[8];   // phase cycles of workgroup 0, wave 0 (SF_TOK_DBG=1; sf_debug_read_ts_ffn_tok)
#ifdef TK_STAMPS
#define TKS(...) __VA_ARGS__
#else
#define TKS(...)
#endif

__global__ __launch_bounds__(TK_NT) void ffn_tok_kernel(TokArgs A) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) char tk_lds[];
  float* B1 = (float*)(tk_lds + TK_ROWS_LDS);
  TKS(const long long w_entry = wall_clock64();)
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int M = A.M;
  const int tok = blockIdx.x * TK_WGT + wave * 32 + n;
  const int tokc = min(tok, M - 1);
  // ---- stage 0 on its way: wave w copies fragments 16 w .. 16 w + 15 of a stage ----
  // piece f (0..15) of stage hb: one fragment per wave (global_load_lds: 64 lanes x 16 B = the fragment, lane-linear on both sides)
  auto stage_piece = [&](int hb, int f, int buf) {
    const char* src = (const char*)A.wp + (size_t)hb * TK_STAGE + (size_t)(wave * 16 + f) * 1024 + lane * 16;
    char* dst = tk_lds + buf * TK_STAGE + (wave * 16 + f) * 1024;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
  };
  auto stage_load = [&](int hb) {
#pragma unroll
    for (int f = 0; f < 16; ++f) stage_piece(hb, f, hb & 1);
  };
  // ---- the workgroup's rows, whole (one wave-instruction = one 1 KB row, global -> LDS), into the ring's space; lin1's bias behind it ----
  char* rows = tk_lds + (size_t)(wave * 32) * TK_RP;   // this wave's 32 rows
  {
    const int r0 = blockIdx.x * TK_WGT + wave * 32;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) {
      const char* src = (const char*)(A.x2 + (long long)min(r0 + r, M - 1) * TK_D) + lane * 16;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)(rows + r * TK_RP), 16, 0, 0);
    }
  }
  *(f32x4*)(B1 + 4 * t) = *(const f32x4*)(A.b1 + 4 * t);
  if (t < 8) *(f32x4*)(B1 + TK_F + 4 * t) = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();   // (drains the wave's own row requests; the rows are read by the wave that requested them)
  // ---- LN2 of the wave's 32 tokens: lane (token n, half h) holds channels 16 ks + 8 h .. + 7 of every k-step -> B-operand fragments ----
  bf16x8 xh[16], xl[16];
  {
    const char* xr = rows + n * TK_RP + 32 * h;
    f32x4 v[16][2];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      v[ks][0] = *(const f32x4*)(xr + 64 * ks);
      v[ks][1] = *(const f32x4*)(xr + 64 * ks + 16);
    }
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) s += ((v[ks][0][0] + v[ks][0][1]) + (v[ks][0][2] + v[ks][0][3])) + ((v[ks][1][0] + v[ks][1][1]) + (v[ks][1][2] + v[ks][1][3]));
    s += __shfl_xor(s, 32, 64);
    const float mu = s * (1.0f / TK_D);
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const f32x4 d0 = v[ks][0] - mu, d1 = v[ks][1] - mu;
      q += ((d0[0] * d0[0] + d0[1] * d0[1]) + (d0[2] * d0[2] + d0[3] * d0[3])) + ((d1[0] * d1[0] + d1[1] * d1[1]) + (d1[2] * d1[2] + d1[3] * d1[3]));
    }
    q += __shfl_xor(q, 32, 64);
    const float rs = 1.0f / sqrtf(q * (1.0f / TK_D) + A.ln_eps);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int c = 16 * ks + 8 * h;
      const f32x4 y0 = (v[ks][0] - mu) * rs * *(const f32x4*)(A.ln_g + c) + *(const f32x4*)(A.ln_b + c);
      const f32x4 y1 = (v[ks][1] - mu) * rs * *(const f32x4*)(A.ln_g + c + 4) + *(const f32x4*)(A.ln_b + c + 4);
      const bf16x4 h0 = __builtin_convertvector(y0, bf16x4), h1 = __builtin_convertvector(y1, bf16x4);
      const bf16x4 l0 = __builtin_convertvector(y0 - __builtin_convertvector(h0, f32x4), bf16x4);
      const bf16x4 l1 = __builtin_convertvector(y1 - __builtin_convertvector(h1, f32x4), bf16x4);
      xh[ks] = cat8(h0, h1);
      xl[ks] = cat8(l0, l1);
    }
  }
  __syncthreads();   // every wave has its fragments: the staged rows make way for the weight ring
  stage_load(0);
  f32x16 Y[8];
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) Y[ob][r] = 0.f;

  // Software pipeline over the hidden blocks: iteration j runs the FIRST product of block j (stage j's first half) and the SECOND product of block
  // j - 1 (its second half, on the hi / lo fragments `hh`, `hl` the previous iteration left), and turns block j's accumulators into the next
  // iteration's fragments between the MFMAs of the second product -- bias / ReLU / split have independent MFMAs to hide under.  Iteration 0's second
  // product and iteration 32's first run on zero fragments.
  bf16x8 hh[2], hl[2];
  {
    const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    hh[0] = hh[1] = hl[0] = hl[1] = z8;
  }
  TKS(long long c_bar = 0, c_p1 = 0, c_p2 = 0; const long long c_start = __builtin_readcyclecounter(); const long long w_start = wall_clock64();)
  for (int j = 0; j < TK_NST; ++j) {
    TKS(const long long c0 = __builtin_readcyclecounter();)
    __syncthreads();   // stage j has landed (the compiler drains vmcnt before the barrier); every wave is done with stage j - 1
    const int jn = min(j + 1, TK_NST - 1), nb = (j + 1) & 1;   // the next stage: one fragment per wave and group, between the MFMAs
    const char* st = tk_lds + (j & 1) * TK_STAGE + lane * 16;
    bf16x8 wb[2][4], wc[2][4];
    auto rd = [&](int g, int buf) {
#pragma unroll
      for (int i = 0; i < 4; ++i) wb[buf][i] = *(const bf16x8*)(st + g * 4096 + i * 1024);
    };
    auto rdc = [&](int g, int buf) {
#pragma unroll
      for (int i = 0; i < 4; ++i) wc[buf][i] = *(const bf16x8*)(st + g * 4096 + i * 1024);
    };
    TKS(const long long c1 = __builtin_readcyclecounter();)
    rd(0, 0);
    // ---- H^T[block j] = W1[block j] . LN2(x)^T: two accumulator chains (even / odd k-steps) ----
    f32x16 Ha, Hb;
#pragma unroll
    for (int r = 0; r < 16; ++r) Ha[r] = Hb[r] = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int ks = 2 * g, b = g & 1;
      // (the reads of group g + 1 go BEHIND the first MFMA of group g: the wait in front of it then covers exactly the reads it needs)
      Ha = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[b][0], xl[ks], Ha, 0, 0, 0);
      rd(g + 1, (g + 1) & 1);
      if (g == 7) rdc(9, 0);
      stage_piece(jn, g, nb);
      Hb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[b][2], xl[ks + 1], Hb, 0, 0, 0);
      Ha = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[b][1], xh[ks], Ha, 0, 0, 0);
      Hb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[b][3], xh[ks + 1], Hb, 0, 0, 0);
      Ha = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[b][0], xh[ks], Ha, 0, 0, 0);
      Hb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[b][2], xh[ks + 1], Hb, 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (g == 7)
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      else
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- Y^T += W2[:, block j - 1] . H^T[block j - 1], two output blocks at a time (groups 8 + 2 op in wb[op & 1], 9 + 2 op in wc[op & 1]); between
    //      them, a quarter of block j's bias / ReLU / split per pair ----
    TKS(const long long c2 = __builtin_readcyclecounter();)
    bf16x8 nh[2], nl[2];
    bf16x4 ph, plo;
#pragma unroll
    for (int op = 0; op < 4; ++op) {
      const int oa = 2 * op, ob = 2 * op + 1, b = op & 1;
      if (op + 1 < 4) {
        rd(10 + 2 * op, b ^ 1);
        rdc(11 + 2 * op, b ^ 1);
      }
      Y[oa] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[b][0], hl[0], Y[oa], 0, 0, 0);
      Y[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[b][0], hl[0], Y[ob], 0, 0, 0);
      Y[oa] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[b][1], hh[0], Y[oa], 0, 0, 0);
      Y[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[b][1], hh[0], Y[ob], 0, 0, 0);
      Y[oa] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[b][0], hh[0], Y[oa], 0, 0, 0);
      Y[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[b][0], hh[0], Y[ob], 0, 0, 0);
      stage_piece(jn, 8 + 2 * op, nb);
      stage_piece(jn, 9 + 2 * op, nb);
      {
        // accumulator registers 4 op .. 4 op + 3 of block j: hidden rows 8 op + 4 h + q  (op = 2 s + half: k-step s of the next second product)
        const f32x4 bv = *(const f32x4*)(B1 + 32 * j + 8 * op + 4 * h);
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaxf((Ha[4 * op + q] + Hb[4 * op + q]) + bv[q], 0.f);
        const bf16x4 hi = __builtin_convertvector(v, bf16x4);
        const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
        if (op & 1) {
          nh[op >> 1] = cat8(ph, hi);
          nl[op >> 1] = cat8(plo, lo);
        } else {
          ph = hi;
          plo = lo;
        }
      }
      Y[oa] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[b][2], hl[1], Y[oa], 0, 0, 0);
      Y[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[b][2], hl[1], Y[ob], 0, 0, 0);
      Y[oa] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[b][3], hh[1], Y[oa], 0, 0, 0);
      Y[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[b][3], hh[1], Y[ob], 0, 0, 0);
      Y[oa] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[b][2], hh[1], Y[oa], 0, 0, 0);
      Y[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[b][2], hh[1], Y[ob], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    hh[0] = nh[0]; hh[1] = nh[1]; hl[0] = nl[0]; hl[1] = nl[1];
    TKS(const long long c3 = __builtin_readcyclecounter(); c_bar += c1 - c0; c_p1 += c2 - c1; c_p2 += c3 - c2;)
  }
  TKS(if (blockIdx.x == 0 && t == 0) { tk_ts[0] = c_bar; tk_ts[1] = c_p1; tk_ts[2] = c_p2; tk_ts[3] = __builtin_readcyclecounter() - c_start; tk_ts[4] = wall_clock64() - w_start; tk_ts[5] = w_start - w_entry; })
  // ---- y = x2 + (Y + b2): lane (token n, half h) holds output columns 32 ob + 8 g + 4 h + q.  The accumulators go to LDS as rows (over the dead ring; the
  //      16-byte pieces of a lane are bank-spread by the row pitch), then every wave-instruction adds the residual row and the bias and stores ONE whole row ----
  __syncthreads();   // every wave is done with the last stage
  {
    char* yr = rows + n * TK_RP + 16 * h;
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 p = {Y[ob][4 * g], Y[ob][4 * g + 1], Y[ob][4 * g + 2], Y[ob][4 * g + 3]};
        *(f32x4*)(yr + (32 * ob + 8 * g) * 4) = p;
      }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes (rows are read back by the wave that wrote them)
  __builtin_amdgcn_wave_barrier();
  {
    const int r0 = blockIdx.x * TK_WGT + wave * 32;
    const f32x4 b2v = *(const f32x4*)(A.b2 + 4 * lane);
    // (the residual rows in two batches of 16 requests: one row per wave-instruction, all in flight before the first is used)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 res[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) res[i] = *(const f32x4*)(A.x2 + (long long)min(r0 + 16 * half + i, M - 1) * TK_D + 4 * lane);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = 16 * half + i;
        const f32x4 acc = *(const f32x4*)(rows + r * TK_RP + lane * 16);
        if (r0 + r < M) *(f32x4*)(A.y + (long long)(r0 + r) * TK_D + 4 * lane) = acc + (res[i] + b2v);
      }
    }
  }
  TKS(if (blockIdx.x == 0 && t == 0) tk_ts[6] = wall_clock64() - w_entry;)
}

extern "C" int sf_debug_read_ts_ffn_tok(long long* out8) {
  hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(tk_ts), sizeof(long long) * 8);
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" size_t sf_ffn_tok_packed_bytes(void) { return (size_t)TK_NST * TK_STAGE; }

extern "C" int sf_pack_ffn_tok_weights(const float* lin1_w, const float* lin2_w, void* packed, int d_model, int ffn, void* stream) {
  SF_REQUIRE(lin1_w && lin2_w && packed, "sf_pack_ffn_tok_weights: null pointer");
  SF_REQUIRE(d_model == TK_D && ffn == TK_F, "sf_pack_ffn_tok_weights: d_model 256 and ffn 1024 only");
  const int total = TK_NST * 64 * 64;
  hipLaunchKernelGGL(pack_ffn_tok_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, lin1_w, lin2_w, (uint4*)packed);
  SF_CHECK_LAUNCH();
  return 0;
}

// y [M][256] = x2 + lin2(relu(lin1(LN2(x2)))) on finished rows x2 [M][256]; tok_packed: sf_pack_ffn_tok_weights copy of the layer's two matrices
extern "C" int sf_ffn_block_tok_f32(const sf_tfm_layer* w, const void* tok_packed, const float* x2, float* y, int M, void* stream) {
  SF_REQUIRE(w && tok_packed && x2 && y && M > 0, "sf_ffn_block_tok_f32: null pointer / empty problem");
  SF_REQUIRE(w->norm2_g && w->norm2_b && w->lin1_b && w->lin2_b, "sf_ffn_block_tok_f32: null weight");
  SF_REQUIRE(sf_get_precision() == 1, "sf_ffn_block_tok_f32: split-bf16 mode only");
  hipStream_t st = (hipStream_t)stream;
  TokArgs A;
  A.x2 = x2; A.ln_g = w->norm2_g; A.ln_b = w->norm2_b; A.ln_eps = 1e-5f; A.wp = (const uint4*)tok_packed; A.b1 = w->lin1_b; A.b2 = w->lin2_b;
  A.y = y; A.M = M;
  static const int dbg = getenv("SF_TOK_DBG") ? atoi(getenv("SF_TOK_DBG")) : 0;
  A.dbg = dbg;
  SF_TRY(sf_ensure_dyn_lds((const void*)ffn_tok_kernel, TK_LDS));
  sf_prof_begin(SF_K_FFN, st, 4.0 * M * (double)TK_D * TK_F);
  hipLaunchKernelGGL(ffn_tok_kernel, dim3((M + TK_WGT - 1) / TK_WGT), dim3(TK_NT), TK_LDS, st, A);
  sf_prof_end(SF_K_FFN, st);
  SF_CHECK_LAUNCH();
  return 0;
}
