// Two-launch Transformer encoder layer for the rollout (d_model 256, 8 heads of 32, ffn 1024, pre-LN, L <= 64).
//
// The rollout is a chain of short dependent kernels (50 steps x 4 layers); at M = B*L = 1344 rows every GEMM is bound
// by launch + first-load latency (~10-15 us each), not by MFMA or bandwidth.  This file cuts the chain from 18 to 8
// launches per step:
//
//   attn_oproj_kernel  (one WG per (head pair, video)):
//       LN1 -> q|k|v of two heads (split-bf16 MFMA, packed register-resident weights) -> softmax(qk^T)v (f32 MFMA,
//       scores transposed so the softmax stays in registers) -> partial_hp = [o_h0 | o_h1] . Wo[:, 64hp:64hp+64]^T
//       (+ x and the out-proj bias on the two column blocks of the pair)
//   ffn_partial_kernel (one WG per (32-row tile, 256-wide hidden chunk)):
//       x2 = sum of the 4 head-pair partials;  LN2 -> relu(. W1_c^T + b1_c) kept in LDS -> y_c = h_c . W2[:, c]^T;
//       the last workgroup of a tile to arrive sums the 4 chunk partials in fixed order and, on the last layer, also
//       runs the step boundary (out-proj -> slots, in-proj of the new frame -> projection ring)
//   step_boundary_kernel: the same boundary stand-alone (ring initialisation from the burn-in frames, fallback).
//
// The reduction over heads is DEFERRED to the consumer's prologue, which sums the partial buffers while loading them;
// the reduction over hidden chunks is finished inside the FFN launch by the last arriver -- both deterministic (fixed
// order, no atomics on data).  Small parameter vectors are requested before the weight fragments in every kernel
// (vmcnt retires in order), and wave reductions use DPP, not ds_bpermute.
// Reference call site: nn.TransformerEncoderLayer(norm_first=True) as configured at slotformer.py:72-80.
#include "../../include/slotformer_hip.h"
#include "sf_internal.h"
#include "layer_fused.h"
#include <stdlib.h>
#include <type_traits>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int LF_NT = 512;
constexpr int LF_D = 256, LF_HD = 32, LF_NH = 8;   // model width, head width, heads
constexpr int LF_NP = 4;                            // head-PAIR partials written by the attention kernel
constexpr int LF_HC = 256;                          // hidden chunk of the FFN kernel
constexpr int LF_NCH = 4;                           // ffn / LF_HC (ffn = 1024)

// ---- attention kernel geometry ----
constexpr int FA_ROWS = 64, FA_KC = 64;             // padded sequence length, width of an activation load chunk
constexpr int NK = LF_D / FA_KC;                    // 4
constexpr int A_IT = FA_ROWS * (FA_KC / 4) / LF_NT; // 2
constexpr int QSTR = LF_HD + 4;                     // f32 row pitch of the q / k / v tiles (36)

// ---- FFN kernel geometry ----
constexpr int FB_ROWS = 32;
constexpr int FB_AP = LF_D + 8;     // bf16 pitch of the A / H planes (528 B = 33 slots)
constexpr int FB_XP = LF_D + 4;     // f32 pitch of the x2 stash
constexpr size_t FB_LDS = (size_t)4 * FB_ROWS * FB_AP * 2 + (size_t)FB_ROWS * FB_XP * 4;   // A + H planes, x2 stash

__device__ __forceinline__ void split4(__bf16* hp, __bf16* lp, int off, f32x4 v) {
  const bf16x4 hi = __builtin_convertvector(v, bf16x4);
  const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
  *(bf16x4*)(hp + off) = hi;
  *(bf16x4*)(lp + off) = lo;
}
}  // namespace

__device__ unsigned lf_seam_timeouts;   // seam hand-offs that gave up waiting (sf_seam_timeouts): must stay 0
__device__ long long lf_ts[64];   // phase timestamps of one workgroup (SF_DBG=lf=16), read by sf_debug_read_ts
#define LF_TS(i) do { if ((dbg & 16) && blockIdx.x == 0 && threadIdx.x == 0) lf_ts[i] = wall_clock64(); } while (0)
#define LF_TL(i) do { if ((dbg & 16) && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 256) lf_ts[i + 10] = wall_clock64(); } while (0)
#define LF_TQ(i) do { if constexpr (SEAM) { if ((dbg & 16) && hp == 0 && b == 0 && threadIdx.x == 0) lf_ts[i] = wall_clock64(); } } while (0)
#define LF_TA(i) do { if ((dbg & 16) && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) lf_ts[i] = wall_clock64(); } while (0)

// ================================================================================================
// Attention + out-projection of TWO heads per workgroup (grid: 4 head pairs x B videos = 128 workgroups at B = 32, so
// every workgroup has a CU to itself even on the 168-CU rollout partition; with one head per workgroup 64 CUs ran two
// and the kernel took 22 us there vs 16.7 us on the whole chip).
//
// x: layer input rows.  RING = false: xin [B][L][256] contiguous.  RING = true (layer 0 of a rollout step): the rows are
// the cached in-projections of the window's frames, held in a ring of `ring_frames` frames per video
// (xin [B][ring_frames][nslots][256], window starts at frame f0), plus the per-token position table pe [L][256]
// (slotformer.py:115-117: x = in_proj(window) + pe).  ap [4][B*Lq][256]: head-pair partials of the out-projection.
//
// Packed weights (sf_pack_attn_weights), split-bf16 in MFMA-fragment order, loaded straight into registers:
//   q|k|v:    uint4 index = ((((hp*6 + cb)*16 + ks)*2 + plane)*64 + lane)
//             element j = in_proj_w[(cb % 3)*256 + (2 hp + cb / 3)*32 + (lane & 31)][ks*16 + 8 (lane >> 5) + j]
//   out-proj: uint4 index = ((((hp*4 + ks)*8 + nb)*2 + plane)*64 + lane)
//             element j = out_proj_w[nb*32 + (lane & 31)][hp*64 + ks*16 + 8 (lane >> 5) + j]
constexpr int A2_AP = LF_D + 8;                 // bf16 pitch of the LN(x) planes (528 B = 33 slots)
constexpr int A2_OP = 2 * LF_HD + 8;            // bf16 pitch of the O planes (144 B = 9 slots)
constexpr int A2_XS = 2 * LF_HD + 4;            // f32 pitch of the residual stash
constexpr size_t A2_QKV_OFF = 0;                                             // [2 heads][q,k,v][64][QSTR] f32 (over the planes)
constexpr size_t A2_PLANES = (size_t)2 * FA_ROWS * A2_AP * 2;               // 67,584
constexpr size_t A2_ST_OFF = A2_PLANES;                                      // softmax stats [2][8][32] f32
constexpr size_t A2_OT_OFF = A2_ST_OFF + 2 * 8 * 32 * 4;                     // PV partials [8][32][QSTR] f32
constexpr size_t A2_O_OFF = A2_OT_OFF + (size_t)8 * 32 * QSTR * 4;           // O planes [2][64][A2_OP] bf16
constexpr size_t A2_XS_OFF = A2_O_OFF + (size_t)2 * FA_ROWS * A2_OP * 2;     // residual stash [64][A2_XS] f32
constexpr size_t A2_GB_OFF = A2_XS_OFF + (size_t)FA_ROWS * A2_XS * 4;        // LN gamma | beta [2][256], q|k|v bias [192]
constexpr size_t A2_LDS = A2_GB_OFF + (2 * LF_D + 6 * 32) * 4;
// q / k / v^T of the attention core as split-bf16 planes: pitches of 80 / 144 bytes keep the 16-byte fragment reads of a
// 16-lane group on distinct banks (5 r and 9 r are permutations mod 16)
constexpr int AT_QP = LF_HD + 8, AT_QPL = FA_ROWS * AT_QP;     // q / k plane [64][40]
constexpr int AT_VP = FA_ROWS + 8, AT_VPL = LF_HD * AT_VP;     // v^T plane [32][72]
static_assert((size_t)(8 * AT_QPL + 4 * AT_VPL) * 2 <= A2_PLANES, "q/k/v tiles must fit over the dead LN(x) planes");

__global__ void pack_attn_kernel(const float* __restrict__ win, const float* __restrict__ wo, uint4* __restrict__ pq,
                                 uint4* __restrict__ po) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = idx & 63, plane = (idx >> 6) & 1;
  union {
    __bf16 h[8];
    uint4 u;
  } o;
  if (idx < 4 * 6 * 16 * 2 * 64) {
    const int ks = (idx >> 7) & 15, cb = (idx >> 11) % 6, hp = (idx >> 11) / 6;
    const float* src = win + (long long)((cb % 3) * LF_D + (2 * hp + cb / 3) * LF_HD + (lane & 31)) * LF_D + ks * 16 + 8 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = src[j];
      const __bf16 ah = (__bf16)a;
      o.h[j] = plane ? (__bf16)(a - (float)ah) : ah;
    }
    pq[idx] = o.u;
  }
  if (idx < 4 * 4 * 8 * 2 * 64) {
    const int nb = (idx >> 7) & 7, ks = (idx >> 10) & 3, hp = idx >> 12;
    const float* src = wo + (long long)(nb * 32 + (lane & 31)) * LF_D + hp * 64 + ks * 16 + 8 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = src[j];
      const __bf16 ah = (__bf16)a;
      o.h[j] = plane ? (__bf16)(a - (float)ah) : ah;
    }
    po[idx] = o.u;
  }
}

// Seam between the last FFN launch of rollout step s and the layer-0 attention of step s+1 when both run in ONE launch
// (seam_kernel below): the step boundary of a 32-row tile publishes `epoch` in seam.flags[tile] after its write-through ring
// stores have been acknowledged; an attention workgroup polls the flags of the (at most two) tiles that hold its video's new
// frame, then reads those rows with sc1 loads.  R1 form of cdna_hip_programming.md Guideline 16; every spin is bounded.
struct SeamArgs {
  unsigned* flags;      // [tiles] + error word at [SEAM_ERR]
  unsigned epoch;       // 0: no seam in this launch
  int rows_per_tile;    // 32
};
constexpr int SEAM_ERR = 1023;

struct AttnArgs {
  const float* xin;
  long long x_batch_stride;
  const float* pe;
  int f0, ring_frames, nslots;
  const float *ln_g, *ln_b;
  float ln_eps;
  const uint4* wqkv_p;
  const float* bias;
  const uint4* wo_p;
  const float* bo;
  float* ap;
  long long ap_stride;
  int L, Lq, dbg;
  int nvideos;
  long long xparts_stride;   // PART: xin is the first of the previous layer's LF_NCH FFN chunk partials, this many floats apart
};

// PART: the layer input is the sum of the previous layer's four FFN chunk partials, ((p0 + p1) + p2) + p3 -- the order the
// FFN's own last-arriver reduction uses, so both ways of finishing a layer give the same bits.  It moves the reduction out
// of the FFN launch (arrival counter + barrier + write-through reads of three partials on its critical path, ~4 us) into
// the next attention's prologue (+129 KB of loads, ~1.3 us).
template <bool RING, bool SEAM, bool PART = false>
__device__ __forceinline__ void attn_body(const AttnArgs& A, const int hp, const int b, const SeamArgs seam) {
  // no floating-point contraction in this function: whether the compiler fuses a*b + c into an fma may differ between two
  // instantiations / variants of the same source, and the variants must produce the same bits (LayerNorm statistics)
#pragma clang fp contract(off)
  const float* __restrict__ xin = A.xin;
  const long long x_batch_stride = A.x_batch_stride;
  const float* __restrict__ pe = A.pe;
  const int f0 = A.f0, ring_frames = A.ring_frames, nslots = A.nslots;
  const float* __restrict__ ln_g = A.ln_g;
  const float* __restrict__ ln_b = A.ln_b;
  const float ln_eps = A.ln_eps;
  const uint4* __restrict__ wqkv_p = A.wqkv_p;
  const float* __restrict__ bias = A.bias;
  const uint4* __restrict__ wo_p = A.wo_p;
  const float* __restrict__ bo = A.bo;
  float* __restrict__ ap = A.ap;
  const long long ap_stride = A.ap_stride;
  const int L = A.L, Lq = A.Lq, dbg = A.dbg;
  constexpr int d = LF_D, HD = LF_HD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int s_seam_bad;
  bool seam_bad = false;   // SEAM: the hand-off wait gave up
  __bf16* Ah = (__bf16*)smem;                                  // [64][A2_AP]  LN1(x), all of K
  __bf16* Al = Ah + FA_ROWS * A2_AP;
  __bf16* QKp = (__bf16*)((char*)smem + A2_QKV_OFF);           // q, k: [2 heads][q,k][hi,lo][64][AT_QP] (over the dead planes)
  __bf16* Vtp = QKp + 8 * AT_QPL;                              // v^T:  [2 heads][hi,lo][32][AT_VP]
  float* SM = (float*)((char*)smem + A2_ST_OFF);               // [8][32] running max
  float* SL = SM + 8 * 32;                                     // [8][32] sum of exp
  float* OT = (float*)((char*)smem + A2_OT_OFF);               // [8][32][QSTR]
  __bf16* Oh = (__bf16*)((char*)smem + A2_O_OFF);              // [64][A2_OP]  (k = head * 32 + channel)
  __bf16* Ol = Oh + FA_ROWS * A2_OP;
  float* Xs = (float*)((char*)smem + A2_XS_OFF);               // [64][A2_XS]: x[:, 64 hp : 64 hp + 64]
  float* GB = (float*)((char*)smem + A2_GB_OFF);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* xb = xin + (long long)b * x_batch_stride;
  const int c4 = t & 15, r0 = t >> 4;
  LF_TA(16);
  LF_TQ(9);
  // ---- prologue.  A workgroup ingests ~100 GB/s through one in-order request path: the 196 KB of q|k|v weight fragments
  //      take ~2 us to issue, a wave cannot run the LayerNorm arithmetic before its own last request has been accepted,
  //      and the arithmetic itself needs all 8 waves (two per SIMD) to run at full VALU rate.  So every wave requests
  //      its share of the layer input first, then only the fragments of k-steps 0..7, normalises, and requests
  //      k-steps 8..15 afterwards -- they land while the projection consumes the first half.
  //      Column blocks (32 of the 192 q|k|v columns of the head pair): waves 0..3 own cb = wave over all of K; blocks
  //      4 / 5 are split over K between waves 4 / 5 (k-steps 0..7) and 6 / 7 (k-steps 8..15): waves w and w+4 share a
  //      SIMD, so every SIMD runs 96 + 48 MFMAs. ----
  const int nrb = L > 32 ? 2 : 1;
  // small parameter vectors first (vmcnt retires in order)
  f32x4 gbv = {0.f, 0.f, 0.f, 0.f};
  if (t < 128) gbv = *(const f32x4*)((t < 64 ? ln_g : ln_b) + 4 * (t & 63));
  float qkvb = 0.f;   // q|k|v bias of the two heads: column block cb = (head, which) -> 6 x 32 values
  if (t >= 128 && t < 128 + 192) {
    const int cb = (t - 128) >> 5, j = (t - 128) & 31;
    qkvb = bias[(cb % 3) * d + (2 * hp + cb / 3) * HD + j];
  }
  // ---- activations (+ position table in ring mode) ----
  bool aok[A_IT];
  const float* arow[A_IT];
  const float* prow[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int r = r0 + 32 * i;
    aok[i] = r < L;
    const int rc = min(r, L - 1);
    if constexpr (RING) {
      const int fr = rc / nslots, sl = rc - fr * nslots;
      arow[i] = xb + ((long long)((f0 + fr) % ring_frames) * nslots + sl) * d + 4 * c4;
      prow[i] = pe + (long long)rc * d + 4 * c4;
    } else {
      arow[i] = xb + (long long)rc * d + 4 * c4;
      prow[i] = nullptr;
    }
  }
  // SEAM: the rows of the window's NEWEST frame (the last nslots rows) are being produced by the step boundary in this very
  // launch; their loads wait for the flag (below) -- until then they read row 0 of the video so that every load stays
  // unconditional
  bool late[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    late[i] = false;
    if constexpr (SEAM) {
      const int r = r0 + 32 * i;
      late[i] = r >= L - nslots && r < L;
    }
  }
  f32x4 ra[NK][A_IT];
  if constexpr (PART) {
    const long long ps = A.xparts_stride;
#pragma unroll
    for (int kc = 0; kc < NK; ++kc)
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const float* p = arow[i] + kc * FA_KC;
        const f32x4 p0 = *(const f32x4*)p, p1 = *(const f32x4*)(p + ps), p2 = *(const f32x4*)(p + 2 * ps), p3 = *(const f32x4*)(p + 3 * ps);
        ra[kc][i] = ((p0 + p1) + p2) + p3;
      }
  } else {
#pragma unroll
    for (int kc = 0; kc < NK; ++kc)
#pragma unroll
      for (int i = 0; i < A_IT; ++i) ra[kc][i] = *(const f32x4*)((late[i] ? arow[0] : arow[i]) + kc * FA_KC);
  }
  if constexpr (RING) {
    f32x4 tp[NK][A_IT];
#pragma unroll
    for (int kc = 0; kc < NK; ++kc)
#pragma unroll
      for (int i = 0; i < A_IT; ++i) tp[kc][i] = *(const f32x4*)(prow[i] + kc * FA_KC);
#pragma unroll
    for (int kc = 0; kc < NK; ++kc)
#pragma unroll
      for (int i = 0; i < A_IT; ++i) ra[kc][i] += tp[kc][i];   // (late rows: unused values)
  }
  // ---- weight fragments: wq[ks][plane]; waves >= 4 hold 8 k-steps (in wq[0..7]) ----
  bf16x8 wq[16][2];
  const int wcb = wave < 4 ? wave : 4 + (wave & 1), wk0 = wave >= 6 ? 8 : 0;
  const uint4* wqp = wqkv_p + (((long long)(hp * 6 + wcb) * 16 + wk0) * 2) * 64 + lane;
  if (wave < 6) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      wq[ks][0] = __builtin_bit_cast(bf16x8, wqp[(ks * 2) * 64]);
      wq[ks][1] = __builtin_bit_cast(bf16x8, wqp[(ks * 2 + 1) * 64]);
    }
  }
  LF_TA(17);
  LF_TQ(10);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // LayerNorm of ONE row held by 16 consecutive lanes (statistics from the registers), residual stash (the 64 columns of this
  // head pair are chunk kc == hp) and LN(x) planes.  One code path for the stand-alone and the seam kernel: a rollout restarted
  // from its own output must reproduce the original bit for bit, whichever kernel ran the layer.
  auto ln_row = [&](const f32x4& v0, const f32x4& v1, const f32x4& v2, const f32x4& v3, int r, bool ok) {
#pragma clang fp contract(off)
    const f32x4 vv[NK] = {v0, v1, v2, v3};
#pragma unroll
    for (int kc = 0; kc < NK; ++kc)
      if (kc == hp) *(f32x4*)(Xs + r * A2_XS + 4 * c4) = vv[kc];
    float sm = 0.f;
#pragma unroll
    for (int kc = 0; kc < NK; ++kc) sm += (vv[kc][0] + vv[kc][1]) + (vv[kc][2] + vv[kc][3]);
    const float mu = sf_sum16(sm) / (float)d;
    float vs = 0.f;
#pragma unroll
    for (int kc = 0; kc < NK; ++kc) {
      const f32x4 dv = vv[kc] - mu;
      vs += (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]);
    }
    const float rs = 1.0f / sqrtf(sf_sum16(vs) / (float)d + ln_eps);
#pragma unroll
    for (int kc = 0; kc < NK; ++kc) {
      const int k = kc * FA_KC + 4 * c4;
      const f32x4 g = *(const f32x4*)(GB + k), be = *(const f32x4*)(GB + LF_D + k);
      split4(Ah, Al, r * A2_AP + k, ok ? (vv[kc] - mu) * rs * g + be : zero4);
    }
  };
  // q|k|v^T = W . LN(x)^T (weights as the MFMA A operand) for token block 0 (SEL 0), token block 1 (SEL 1) or both with the
  // same fragment reads (SEL 2); no weight planes, no barriers.  v column blocks (cb % 3 == 2) run with the operands SWAPPED
  // (LN(x) as A, weights as B): their accumulators then hold 16 tokens of ONE channel per lane, which is what the transposed
  // V planes of the attention core are written from.
  f32x16 acc[2];
#pragma unroll
  for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q2][r] = 0.f;
  const bool vblock = (wcb % 3) == 2;
  auto proj = [&](auto selc) {
    constexpr int SEL = decltype(selc)::value;
    const int ao = (lane & 31) * A2_AP + 8 * (lane >> 5) + (wave >= 6 ? 8 * 16 : 0);
    if (!vblock) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (ks >= 8 && wave >= 4) continue;
        if constexpr (SEL != 1) {
          const bf16x8 xh0 = *(const bf16x8*)(Ah + ao + ks * 16), xl0 = *(const bf16x8*)(Al + ao + ks * 16);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[ks][0], xl0, acc[0], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[ks][1], xh0, acc[0], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[ks][0], xh0, acc[0], 0, 0, 0);
        }
        if (SEL != 0 && nrb == 2) {
          const bf16x8 xh1 = *(const bf16x8*)(Ah + ao + 32 * A2_AP + ks * 16), xl1 = *(const bf16x8*)(Al + ao + 32 * A2_AP + ks * 16);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[ks][0], xl1, acc[1], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[ks][1], xh1, acc[1], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[ks][0], xh1, acc[1], 0, 0, 0);
        }
        // SEAM variants: many values live across the hand-off (late rows, statistics): keep the scheduler from hoisting the
        // fragment reads of the whole loop (it spilled 80 registers without the fence)
        if (SEL != 2 && (ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (ks >= 8 && wave >= 4) continue;
        if constexpr (SEL != 1) {
          const bf16x8 xh0 = *(const bf16x8*)(Ah + ao + ks * 16), xl0 = *(const bf16x8*)(Al + ao + ks * 16);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl0, wq[ks][0], acc[0], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh0, wq[ks][1], acc[0], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh0, wq[ks][0], acc[0], 0, 0, 0);
        }
        if (SEL != 0 && nrb == 2) {
          const bf16x8 xh1 = *(const bf16x8*)(Ah + ao + 32 * A2_AP + ks * 16), xl1 = *(const bf16x8*)(Al + ao + 32 * A2_AP + ks * 16);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl1, wq[ks][0], acc[1], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh1, wq[ks][1], acc[1], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh1, wq[ks][0], acc[1], 0, 0, 0);
        }
        if (SEL != 2 && (ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // second half of the fragments: k-steps 8..15 of waves 0..3, the (upper-K) halves of waves 6 / 7
  auto second_half = [&]() {
    __builtin_amdgcn_sched_barrier(0);   // keep these requests BEHIND the arithmetic above in the instruction stream
    if (wave < 4) {
#pragma unroll
      for (int ks = 8; ks < 16; ++ks) {
        wq[ks][0] = __builtin_bit_cast(bf16x8, wqp[(ks * 2) * 64]);
        wq[ks][1] = __builtin_bit_cast(bf16x8, wqp[(ks * 2 + 1) * 64]);
      }
    } else if (wave >= 6) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        wq[ks][0] = __builtin_bit_cast(bf16x8, wqp[(ks * 2) * 64]);
        wq[ks][1] = __builtin_bit_cast(bf16x8, wqp[(ks * 2 + 1) * 64]);
      }
    }
  };
  if (t < 128) *(f32x4*)(GB + 4 * t) = gbv;
  if (t >= 128 && t < 128 + 192) GB[2 * LF_D + (t - 128)] = qkvb;
  if constexpr (SEAM) {
    // the newest frame lies entirely inside token block 1 (the host launches the seam only then; C2: rows 35..41): everything
    // that does not depend on it -- LayerNorm of the old rows, the projection of token block 0 -- runs BEFORE the wait.  A thread
    // holds at most one late row (i = 1); nothing of it is kept in registers across the wait.
    __syncthreads();   // gamma / beta are in LDS
    ln_row(ra[0][0], ra[1][0], ra[2][0], ra[3][0], r0, aok[0]);
    if (!late[1]) ln_row(ra[0][1], ra[1][1], ra[2][1], ra[3][1], r0 + 32, aok[1]);
    second_half();
    __syncthreads();
    proj(std::integral_constant<int, 0>{});
    // ---- the hand-off: tile flags of this video's new frame, then its rows with sc1 loads (+ their position rows) ----
    if (t == 0) {
      s_seam_bad = 0;
      const int t0 = (b * nslots) / seam.rows_per_tile, t1 = (b * nslots + nslots - 1) / seam.rows_per_tile;
      const long long c0 = wall_clock64();
      while (__hip_atomic_load(seam.flags + t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seam.epoch ||
             __hip_atomic_load(seam.flags + t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seam.epoch) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - c0 > 20000000LL) {   // 0.2 s at 100 MHz: a producer is not resident -- give up, flag the error
          __hip_atomic_store(seam.flags + SEAM_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          atomicAdd(&lf_seam_timeouts, 1u);
          s_seam_bad = 1;   // the rows were never handed over: this workgroup's output becomes NaN (loud), see the stores
          break;
        }
      }
    }
    __syncthreads();
    seam_bad = s_seam_bad != 0;
    LF_TQ(11);
    if (late[1]) {
      const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0, 0x7fffffff, 0x00020000);
      const unsigned off = (unsigned)((arow[1] - xin) * 4);
      f32x4 rl[NK];
#pragma unroll
      for (int kc = 0; kc < NK; ++kc)
        rl[kc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, off + kc * FA_KC * 4, 0, 16)) +
                 *(const f32x4*)(prow[1] + kc * FA_KC);
      ln_row(rl[0], rl[1], rl[2], rl[3], r0 + 32, true);
    }
    __syncthreads();
    LF_TQ(12);
    proj(std::integral_constant<int, 1>{});
  } else {
    __syncthreads();   // gamma / beta are in LDS
    ln_row(ra[0][0], ra[1][0], ra[2][0], ra[3][0], r0, aok[0]);
    ln_row(ra[0][1], ra[1][1], ra[2][1], ra[3][1], r0 + 32, aok[1]);
    second_half();
    __syncthreads();
    LF_TA(18);
    proj(std::integral_constant<int, 2>{});
  }
  // the upper k-half of column blocks 4 / 5 meets the lower half in LDS (scratch = the PV-partial tile, free until then)
  float* PS = (float*)((char*)smem + A2_OT_OFF);   // [2 waves][2 token blocks][16][64]
  if (wave >= 6) {
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
      for (int r = 0; r < 16; ++r) PS[(((wave - 6) * 2 + q2) * 16 + r) * 64 + lane] = acc[q2][r];
  }
  // out-proj fragments of column block `wave` (K = 64: 4 k-steps): requested now, consumed at the end
  bf16x8 wof[4][2];
  {
    const uint4* wp = wo_p + (((long long)(hp * 4) * 8 + wave) * 2) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      wof[ks][0] = __builtin_bit_cast(bf16x8, wp[((ks * 8) * 2) * 64]);
      wof[ks][1] = __builtin_bit_cast(bf16x8, wp[((ks * 8) * 2 + 1) * 64]);
    }
  }
  __syncthreads();   // every wave is done with the LN(x) planes: q, k, v take their place
  LF_TA(19);
  LF_TQ(13);
  const float scale = 1.0f / sqrtf((float)HD);
  if (wave == 4 || wave == 5) {
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q2][r] += PS[(((wave - 4) * 2 + q2) * 16 + r) * 64 + lane];
  }
  if (wave < 6) {
    const int which = wave % 3, hh = wave / 3, kg = lane >> 5;
    const float mul = which == 0 ? scale : 1.f;
#pragma unroll
    for (int rbk = 0; rbk < 2; ++rbk) {
      if (rbk < nrb) {
        const int tokn = rbk * 32 + (lane & 31);
        if (which < 2) {
          // q (pre-scaled) and k: split-bf16 planes [token][channel], 8-byte stores
          __bf16* ph = QKp + ((hh * 2 + which) * 2) * AT_QPL;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 bv = *(const f32x4*)(GB + 2 * LF_D + wave * 32 + 8 * g + 4 * kg);
            split4(ph, ph + AT_QPL, tokn * AT_QP + 8 * g + 4 * kg,
                   f32x4{(acc[rbk][4 * g] + bv[0]) * mul, (acc[rbk][4 * g + 1] + bv[1]) * mul, (acc[rbk][4 * g + 2] + bv[2]) * mul,
                         (acc[rbk][4 * g + 3] + bv[3]) * mul});
          }
        } else {
          // v: TRANSPOSED split-bf16 planes [channel][key position].  The swapped-operand projection left token
          // (r & 3) + 8 (r >> 2) + 4 kg of channel lane & 31 in register r, so registers 8 s .. 8 s + 7 go to key positions
          // 16 s + 8 kg .. + 7 as ONE 16-byte store per plane -- i.e. inside every group of 16 keys bits 2 and 3 of the key
          // index are swapped, which is exactly the order in which a lane of the PV MFMA holds its probabilities
          __bf16* vh = Vtp + (hh * 2) * AT_VPL;
          const float bch = GB[2 * LF_D + wave * 32 + (lane & 31)];
#pragma unroll
          for (int sg = 0; sg < 2; ++sg) {
            const f32x4 a0 = {acc[rbk][8 * sg] + bch, acc[rbk][8 * sg + 1] + bch, acc[rbk][8 * sg + 2] + bch, acc[rbk][8 * sg + 3] + bch};
            const f32x4 a1 = {acc[rbk][8 * sg + 4] + bch, acc[rbk][8 * sg + 5] + bch, acc[rbk][8 * sg + 6] + bch, acc[rbk][8 * sg + 7] + bch};
            const bf16x4 h0 = __builtin_convertvector(a0, bf16x4), h1 = __builtin_convertvector(a1, bf16x4);
            const bf16x4 l0 = __builtin_convertvector(a0 - __builtin_convertvector(h0, f32x4), bf16x4);
            const bf16x4 l1 = __builtin_convertvector(a1 - __builtin_convertvector(h1, f32x4), bf16x4);
            const int off = (lane & 31) * AT_VP + rbk * 32 + 16 * sg + 8 * kg;
            *(bf16x8*)(vh + off) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            *(bf16x8*)(vh + AT_VPL + off) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
          }
        }
      }
    }
  }
  __syncthreads();
  LF_TA(20);

  // ---- attention of both heads at once: wave = (head hh = wave >> 2, query block qb, key block kb).  Scores are computed
  //      TRANSPOSED, S^T[key][query] = k q^T on the f32 MFMA: a lane then holds 16 keys of ONE query column, so the softmax
  //      over keys is a per-lane reduction plus one exchange with lane ^ 32, and p feeds the PV MFMA straight from
  //      registers (B operand: key pair (k, k+4)). ----
  const int hh = wave >> 2, sub = wave & 3;
  const int qb = (nrb == 2) ? (sub >> 1) : 0, kb = (nrb == 2) ? (sub & 1) : 0;
  const bool score_wave = sub < nrb * nrb;
  const __bf16* Qh = QKp + ((hh * 2 + 0) * 2) * AT_QPL;
  const __bf16* Kh = QKp + ((hh * 2 + 1) * 2) * AT_QPL;
  const __bf16* Vh = Vtp + (hh * 2) * AT_VPL;
  f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
  float m_w = -INFINITY, l_w = 0.f;
  if (score_wave) {
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
    const int ko = (kb * 32 + (lane & 31)) * AT_QP + 8 * (lane >> 5), qo = (qb * 32 + (lane & 31)) * AT_QP + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
      const bf16x8 kh = *(const bf16x8*)(Kh + ko + 16 * ks), kl = *(const bf16x8*)(Kh + AT_QPL + ko + 16 * ks);
      const bf16x8 qh = *(const bf16x8*)(Qh + qo + 16 * ks), ql = *(const bf16x8*)(Qh + AT_QPL + qo + 16 * ks);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql, sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh, sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh, sacc, 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      sacc[r] = key < L ? sacc[r] : -INFINITY;
      mx = fmaxf(mx, sacc[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sacc[r] = (mx == -INFINITY) ? 0.f : expf(sacc[r] - mx);
      sum += sacc[r];
    }
    sum += __shfl_xor(sum, 32, 64);
    m_w = mx;
    l_w = sum;
    if (lane < 32) {
      SM[wave * 32 + lane] = mx;
      SL[wave * 32 + lane] = sum;
    }
    // PV on the split-bf16 MFMA: register r of the score tile is key (r & 3) + 8 (r >> 2) + 4 (lane >> 5), so registers
    // 8 s .. 8 s + 7 are exactly the 8 keys of k-step s that this lane contracts over (B operand), in the order the V^T
    // planes were permuted to (A operand)
    const int vo = (lane & 31) * AT_VP + kb * 32 + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const f32x4 p0 = {sacc[8 * ks], sacc[8 * ks + 1], sacc[8 * ks + 2], sacc[8 * ks + 3]};
      const f32x4 p1 = {sacc[8 * ks + 4], sacc[8 * ks + 5], sacc[8 * ks + 6], sacc[8 * ks + 7]};
      const bf16x4 h0 = __builtin_convertvector(p0, bf16x4), h1 = __builtin_convertvector(p1, bf16x4);
      const bf16x4 l0 = __builtin_convertvector(p0 - __builtin_convertvector(h0, f32x4), bf16x4);
      const bf16x4 l1 = __builtin_convertvector(p1 - __builtin_convertvector(h1, f32x4), bf16x4);
      const bf16x8 ph = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
      const bf16x8 pl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
      const bf16x8 vh = *(const bf16x8*)(Vh + vo + 16 * ks), vl = *(const bf16x8*)(Vh + AT_VPL + vo + 16 * ks);
      oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, oacc, 0, 0, 0);
      oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, oacc, 0, 0, 0);
      oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, oacc, 0, 0, 0);
    }
  }
  __syncthreads();
  LF_TA(21);
  if (score_wave) {
    float fsc;
    if (nrb == 2) {
      const float m_o = SM[(wave ^ 1) * 32 + (lane & 31)], l_o = SL[(wave ^ 1) * 32 + (lane & 31)];
      const float mg = fmaxf(m_w, m_o);
      const float fw = (m_w == -INFINITY) ? 0.f : expf(m_w - mg), fo = (m_o == -INFINITY) ? 0.f : expf(m_o - mg);
      fsc = fw / (l_w * fw + l_o * fo);
    } else {
      fsc = 1.0f / l_w;
    }
    float* od = OT + ((wave * 32) + (lane & 31)) * QSTR + 4 * (lane >> 5);
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *(f32x4*)(od + 8 * g) = f32x4{oacc[4 * g] * fsc, oacc[4 * g + 1] * fsc, oacc[4 * g + 2] * fsc, oacc[4 * g + 3] * fsc};
  }
  __syncthreads();
  LF_TA(23);
  {
    // O planes [token][head*32 + ch]: thread = (row, head, float4 of 8): 64 x 2 x 8 = 1024 items, 2 per thread
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = t + LF_NT * it;
      const int row = idx >> 4, h2 = (idx >> 3) & 1, q4 = idx & 7;
      const int qbr = row >> 5, qq = row & 31;
      f32x4 v = zero4;
      if (row < nrb * 32 && row >= L - Lq && row < L) {
        const int w0 = h2 * 4 + (nrb == 2 ? 2 * qbr : 0);
        v = *(const f32x4*)(OT + (w0 * 32 + qq) * QSTR + 4 * q4);
        if (nrb == 2) v += *(const f32x4*)(OT + ((w0 + 1) * 32 + qq) * QSTR + 4 * q4);
      }
      split4(Oh, Ol, row * A2_OP + h2 * 32 + 4 * q4, v);
    }
  }
  __syncthreads();
  LF_TA(24);

  // ---- head-pair partial^T = Wo[32 w .. 32 w + 31, 64 hp : 64 hp + 64] . [o_h0 | o_h1]^T: wave w owns output columns 32 w ..
  //      32 w + 31.  Transposed (weights as the MFMA A operand) so that a lane owns 4 CONSECUTIVE columns of one token: the
  //      residual, the bias and the stores are 16-byte vectors, written through (sc1) -- nothing of the 5.5 MB of partials is
  //      left dirty in the L2s for the kernel boundary to flush ----
  const int nq0 = L - Lq;
  const __amdgpu_buffer_rsrc_t apr = __builtin_amdgcn_make_buffer_rsrc(ap, 0, 0x7fffffff, 0x00020000);
  const bool mine = (wave >> 1) == hp;   // residual + bias live on the two column blocks of this head pair
  f32x4 bo4[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bo4[g] = *(const f32x4*)(bo + wave * 32 + 8 * g + 4 * (lane >> 5));
  for (int rbk = 0; rbk < nrb; ++rbk) {
    if (rbk * 32 + 32 <= nq0) continue;   // no query rows in this block
    f32x16 pacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
    const int ao = (rbk * 32 + (lane & 31)) * A2_OP + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 xh = *(const bf16x8*)(Oh + ao + ks * 16), xl = *(const bf16x8*)(Ol + ao + ks * 16);
      pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wof[ks][0], xl, pacc, 0, 0, 0);
      pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wof[ks][1], xh, pacc, 0, 0, 0);
      pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wof[ks][0], xh, pacc, 0, 0, 0);
    }
    const int row = rbk * 32 + (lane & 31);
    if (row >= nq0 && row < L && !((dbg & 8) && !mine)) {
      const unsigned off = (unsigned)((((long long)hp * ap_stride + ((long long)b * Lq + (row - nq0)) * d + wave * 32 + 4 * (lane >> 5))) * 4);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {pacc[4 * g], pacc[4 * g + 1], pacc[4 * g + 2], pacc[4 * g + 3]};
        if (mine) v += *(const f32x4*)(Xs + row * A2_XS + (wave & 1) * 32 + 8 * g + 4 * (lane >> 5)) + bo4[g];
        if constexpr (SEAM) {
          // a hand-off that timed out must not produce plausible numbers: NaN travels through the FFN into the slots
          if (seam_bad) v = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), apr, off + 32 * g, 0, 16);
      }
    }
  }
  LF_TA(25);
  LF_TQ(14);
}

template <bool RING, bool PART = false>
__global__ __launch_bounds__(LF_NT) void attn_oproj_kernel(AttnArgs A) {
  if constexpr (PART) {
    // 1-D grid: block i runs on XCD i % 8; the four head pairs of a video share an XCD, so the 172 KB of chunk partials a
    // video's workgroups all read cross the fabric once (video b on XCD b % 8)
    const int i = blockIdx.x, k = i >> 3, b = (i & 7) + 8 * (k >> 2);
    if (b >= A.nvideos) return;
    attn_body<RING, false, PART>(A, k & 3, b, SeamArgs{nullptr, 0u, 32});
    return;
  }
  attn_body<RING, false, PART>(A, blockIdx.x, blockIdx.y, SeamArgs{nullptr, 0u, 32});
}

// ================================================================================================
// Attention block of a layer with ONE workgroup per video and ALL 8 heads (per-call option attn_heads_per_wg = 8): the
// throughput form.  The head-pair kernel above gives a rollout chain short launches (128 workgroups of ~12 us at B = 32), at
// the price that the four workgroups of a video each ingest the layer input (172 KB as chunk partials) and normalise it, and
// that the FFN behind them ingests FOUR head-pair partials per row.  A CU ingests only ~70-100 GB/s, and with several rollout
// units sharing the CUs (pipeline 'pair') CU time is what bounds the rollout.  Here the input is loaded and normalised once,
// the LN(x) planes stay resident, the four head pairs run one after the other (their weight fragments arrive behind the
// previous pair's attention core) and accumulate the out-projection in registers: the output is the FINISHED row
//     x2 = x + out_proj(MHA(LN1(x))) + b_o          [B*Lq][256]
// -- one tensor instead of four partials, so the FFN launch behind it ingests 1 KB per row instead of 4.
// Every product, and the order in which the head pairs' out-projection partials are summed -- ((p0 + p1) + p2) + p3, with the
// residual and the bias on the partial of the pair that owns the columns -- are the head-pair kernel's and its consumer's: the
// two forms give the same bits.
constexpr size_t A8_PLANES_OFF = 0;                                            // LN(x) planes [2][64][A2_AP] bf16, resident
constexpr size_t A8_QK_OFF = A2_PLANES;                                        // q, k planes of one head pair; later OT
constexpr size_t A8_VT_OFF = A8_QK_OFF + (size_t)8 * AT_QPL * 2;               // v^T planes of one head pair; later the O planes
constexpr size_t A8_ST_OFF = A8_VT_OFF + (size_t)4 * AT_VPL * 2;               // softmax stats [2][8][32] f32
constexpr size_t A8_PS_OFF = A8_ST_OFF + 2 * 8 * 32 * 4;                       // k-half exchange [2 waves][2][16][64] f32
constexpr size_t A8_GB_OFF = A8_PS_OFF + (size_t)2 * 2 * 16 * 64 * 4;          // gamma | beta [2][256], q|k|v bias [768]
constexpr size_t A8_LDS = A8_GB_OFF + (2 * LF_D + 3 * LF_D) * 4;
static_assert((size_t)8 * 32 * QSTR * 4 <= (size_t)8 * AT_QPL * 2, "the PV partials must fit over the q / k planes");
static_assert((size_t)2 * FA_ROWS * A2_OP * 2 <= (size_t)4 * AT_VPL * 2, "the O planes must fit over the v^T planes");
static_assert(A8_LDS <= 160 * 1024, "all-heads attention kernel: LDS budget");

template <bool RING, bool PART>
__global__ __launch_bounds__(LF_NT) void attn_all_kernel(AttnArgs A) {
#pragma clang fp contract(off)
  const int b = blockIdx.x;   // video (block i runs on XCD i % 8)
#define ATS(i) do { if ((A.dbg & 16) && blockIdx.x == 0 && threadIdx.x == 0) lf_ts[48 + (i)] = wall_clock64(); } while (0)
  ATS(0);
  const float* __restrict__ xin = A.xin;
  const float* __restrict__ pe = A.pe;
  const int f0 = A.f0, ring_frames = A.ring_frames, nslots = A.nslots;
  const uint4* __restrict__ wqkv_p = A.wqkv_p;
  const uint4* __restrict__ wo_p = A.wo_p;
  const int L = A.L, Lq = A.Lq;
  constexpr int d = LF_D, HD = LF_HD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* Ah = (__bf16*)smem;                                  // [64][A2_AP]  LN1(x), all of K: resident
  __bf16* Al = Ah + FA_ROWS * A2_AP;
  __bf16* QKp = (__bf16*)((char*)smem + A8_QK_OFF);            // q, k: [2 heads][q,k][hi,lo][64][AT_QP]
  __bf16* Vtp = (__bf16*)((char*)smem + A8_VT_OFF);            // v^T:  [2 heads][hi,lo][32][AT_VP]
  float* OT = (float*)((char*)smem + A8_QK_OFF);               // [8][32][QSTR] PV partials (over the dead q / k planes)
  __bf16* Oh = (__bf16*)((char*)smem + A8_VT_OFF);             // [64][A2_OP] O planes (over the dead v^T planes)
  __bf16* Ol = Oh + FA_ROWS * A2_OP;
  float* SM = (float*)((char*)smem + A8_ST_OFF);
  float* SL = SM + 8 * 32;
  float* PS = (float*)((char*)smem + A8_PS_OFF);
  float* GB = (float*)((char*)smem + A8_GB_OFF);
  const int t_ = threadIdx.x, lane_ = t_ & 63, wave_ = t_ >> 6;
  const int t = t_, lane = lane_, wave = wave_;
  const float* xb = xin + (long long)b * A.x_batch_stride;
  const int c4 = t & 15, r0 = t >> 4;
  const int nrb = L > 32 ? 2 : 1;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // ---- parameters first (vmcnt retires in order): gamma | beta, the q|k|v bias of all heads ----
  f32x4 gbv = zero4, bqv = zero4;
  if (t < 128) gbv = *(const f32x4*)((t < 64 ? A.ln_g : A.ln_b) + 4 * (t & 63));
  if (t >= 128 && t < 128 + 192) bqv = *(const f32x4*)(A.bias + 4 * (t - 128));
  // ---- the layer input, once ----
  bool aok[A_IT];
  const float* arow[A_IT];
  const float* prow[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int r = r0 + 32 * i;
    aok[i] = r < L;
    const int rc = min(r, L - 1);
    if constexpr (RING) {
      const int fr = rc / nslots, sl = rc - fr * nslots;
      arow[i] = xb + ((long long)((f0 + fr) % ring_frames) * nslots + sl) * d + 4 * c4;
      prow[i] = pe + (long long)rc * d + 4 * c4;
    } else {
      arow[i] = xb + (long long)rc * d + 4 * c4;
      prow[i] = nullptr;
    }
  }
  f32x4 ra[NK][A_IT];
  if constexpr (PART) {
    const long long ps = A.xparts_stride;
#pragma unroll
    for (int kc = 0; kc < NK; ++kc)
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const float* p = arow[i] + kc * FA_KC;
        const f32x4 p0 = *(const f32x4*)p, p1 = *(const f32x4*)(p + ps), p2 = *(const f32x4*)(p + 2 * ps), p3 = *(const f32x4*)(p + 3 * ps);
        ra[kc][i] = ((p0 + p1) + p2) + p3;
      }
  } else {
#pragma unroll
    for (int kc = 0; kc < NK; ++kc)
#pragma unroll
      for (int i = 0; i < A_IT; ++i) ra[kc][i] = *(const f32x4*)(arow[i] + kc * FA_KC);
  }
  if constexpr (RING) {
#pragma unroll
    for (int kc = 0; kc < NK; ++kc)
#pragma unroll
      for (int i = 0; i < A_IT; ++i) ra[kc][i] += *(const f32x4*)(prow[i] + kc * FA_KC);
  }
  // ---- weight fragments of a head pair: wq[ks][plane]; waves >= 4 hold 8 k-steps (in wq[0..7]); column blocks as in
  //      attn_body (waves 0..3 own one of the 6 q|k|v blocks over all of K, blocks 4 / 5 are split over K) ----
  bf16x8 wq[16][2];
  const int wcb = wave < 4 ? wave : 4 + (wave & 1), wk0 = wave >= 6 ? 8 : 0;
  auto wq_ptr = [&](int hp) { return wqkv_p + (((long long)(hp * 6 + wcb) * 16 + wk0) * 2) * 64 + lane; };
  auto load_first = [&](int hp) {   // k-steps 0..7 of waves 0..5
    const uint4* wqp = wq_ptr(hp);
    if (wave < 6) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        wq[ks][0] = __builtin_bit_cast(bf16x8, wqp[(ks * 2) * 64]);
        wq[ks][1] = __builtin_bit_cast(bf16x8, wqp[(ks * 2 + 1) * 64]);
      }
    }
  };
  auto load_second = [&](int hp) {  // k-steps 8..15 of waves 0..3, the (upper-K) halves of waves 6 / 7
    const uint4* wqp = wq_ptr(hp);
    if (wave < 4) {
#pragma unroll
      for (int ks = 8; ks < 16; ++ks) {
        wq[ks][0] = __builtin_bit_cast(bf16x8, wqp[(ks * 2) * 64]);
        wq[ks][1] = __builtin_bit_cast(bf16x8, wqp[(ks * 2 + 1) * 64]);
      }
    } else if (wave >= 6) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        wq[ks][0] = __builtin_bit_cast(bf16x8, wqp[(ks * 2) * 64]);
        wq[ks][1] = __builtin_bit_cast(bf16x8, wqp[(ks * 2 + 1) * 64]);
      }
    }
  };
  load_first(0);
  ATS(1);
  if (t < 128) *(f32x4*)(GB + 4 * t) = gbv;
  if (t >= 128 && t < 128 + 192) *(f32x4*)(GB + 2 * LF_D + 4 * (t - 128)) = bqv;
  __syncthreads();   // gamma / beta are in LDS
  // LayerNorm of ONE row held by 16 consecutive lanes -> LN(x) planes (attn_body's arithmetic)
  auto ln_row = [&](const f32x4& v0, const f32x4& v1, const f32x4& v2, const f32x4& v3, int r, bool ok) {
#pragma clang fp contract(off)
    const f32x4 vv[NK] = {v0, v1, v2, v3};
    float sm = 0.f;
#pragma unroll
    for (int kc = 0; kc < NK; ++kc) sm += (vv[kc][0] + vv[kc][1]) + (vv[kc][2] + vv[kc][3]);
    const float mu = sf_sum16(sm) / (float)d;
    float vs = 0.f;
#pragma unroll
    for (int kc = 0; kc < NK; ++kc) {
      const f32x4 dv = vv[kc] - mu;
      vs += (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]);
    }
    const float rs = 1.0f / sqrtf(sf_sum16(vs) / (float)d + A.ln_eps);
#pragma unroll
    for (int kc = 0; kc < NK; ++kc) {
      const int k = kc * FA_KC + 4 * c4;
      const f32x4 g = *(const f32x4*)(GB + k), be = *(const f32x4*)(GB + LF_D + k);
      split4(Ah, Al, r * A2_AP + k, ok ? (vv[kc] - mu) * rs * g + be : zero4);
    }
  };
  ln_row(ra[0][0], ra[1][0], ra[2][0], ra[3][0], r0, aok[0]);
  ln_row(ra[0][1], ra[1][1], ra[2][1], ra[3][1], r0 + 32, aok[1]);
  const int nq0 = L - Lq;
  // the residual x of the query rows is PARKED in the output rows (row-major ownership here, read back in the accumulator
  // layout at the end: same workgroup, barriers in between, lines no CU has cached) -- it would cost 32 registers for the
  // whole kernel, or a second pass over the input
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int r = r0 + 32 * i;
    if (r >= nq0 && r < L) {
#pragma unroll
      for (int kc = 0; kc < NK; ++kc) *(f32x4*)(A.ap + ((long long)b * Lq + (r - nq0)) * d + kc * FA_KC + 4 * c4) = ra[kc][i];
    }
  }
  __syncthreads();
  ATS(2);
  const float scale = 1.0f / sqrtf((float)HD);
  // out-projection: wave w owns output columns 32 w .. 32 w + 31 of both token blocks.  Every head pair's partial is computed
  // exactly as the head-pair kernel computes it (fresh accumulator; the residual and the bias ride on the partial of pair
  // w >> 1) and the partials are summed in ITS consumer's order, ((p0 + p1) + p2) + p3 -- so x2 has the bits the FFN launch
  // would have formed from the four partial buffers, and the two forms of the attention block are interchangeable.  The
  // running sum lives in global scratch (A.ap + A.ap_stride; written and read back by the same thread, L2-resident) instead
  // of 32 registers carried through the head-pair loop.
#pragma unroll 1
  for (int hp = 0; hp < LF_NH / 2; ++hp) {
    // ---- q|k|v^T = W . LN(x)^T of this head pair for both token blocks (attn_body's proj, SEL 2).  The first half of its
    //      weight fragments was requested behind the previous pair's projection; the second half is requested here and lands
    //      while the first eight k-steps run (all sixteen in flight across the attention core would not fit the registers) ----
    // LDS addresses are recomputed per head pair from these opaque copies: hoisted out of the loop, the dozens of
    // loop-invariant addresses (regions more than 64 KB apart: no common base + immediate) were spilled to scratch
    int lane = lane_, wave = wave_, t = t_;
    asm volatile("" : "+v"(lane), "+v"(wave), "+v"(t));
    if (hp == 1) ATS(3);
    load_second(hp);
    f32x16 acc[2];
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q2][r] = 0.f;
    const bool vblock = (wcb % 3) == 2;
    {
      const int ao = (lane & 31) * A2_AP + 8 * (lane >> 5) + (wave >= 6 ? 8 * 16 : 0);
      if (!vblock) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          if (ks >= 8 && wave >= 4) continue;
          const bf16x8 xh0 = *(const bf16x8*)(Ah + ao + ks * 16), xl0 = *(const bf16x8*)(Al + ao + ks * 16);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[ks][0], xl0, acc[0], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[ks][1], xh0, acc[0], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[ks][0], xh0, acc[0], 0, 0, 0);
          if (nrb == 2) {
            const bf16x8 xh1 = *(const bf16x8*)(Ah + ao + 32 * A2_AP + ks * 16), xl1 = *(const bf16x8*)(Al + ao + 32 * A2_AP + ks * 16);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[ks][0], xl1, acc[1], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[ks][1], xh1, acc[1], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[ks][0], xh1, acc[1], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          if (ks >= 8 && wave >= 4) continue;
          const bf16x8 xh0 = *(const bf16x8*)(Ah + ao + ks * 16), xl0 = *(const bf16x8*)(Al + ao + ks * 16);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl0, wq[ks][0], acc[0], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh0, wq[ks][1], acc[0], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh0, wq[ks][0], acc[0], 0, 0, 0);
          if (nrb == 2) {
            const bf16x8 xh1 = *(const bf16x8*)(Ah + ao + 32 * A2_AP + ks * 16), xl1 = *(const bf16x8*)(Al + ao + 32 * A2_AP + ks * 16);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl1, wq[ks][0], acc[1], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh1, wq[ks][1], acc[1], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh1, wq[ks][0], acc[1], 0, 0, 0);
          }
        }
      }
    }
    if (hp == 1) ATS(4);
    // the upper k-half of column blocks 4 / 5 meets the lower half in LDS
    if (wave >= 6) {
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
        for (int r = 0; r < 16; ++r) PS[(((wave - 6) * 2 + q2) * 16 + r) * 64 + lane] = acc[q2][r];
    }
    // the weight registers are free: the NEXT head pair's first half is requested now and lands behind the attention core;
    // then the out-proj fragments of THIS pair (column block `wave`, K = 64: 4 k-steps)
    __builtin_amdgcn_sched_barrier(0);
    if (hp + 1 < LF_NH / 2) load_first(hp + 1);
    bf16x8 wof[4][2];
    {
      const uint4* wp = wo_p + (((long long)(hp * 4) * 8 + wave) * 2) * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        wof[ks][0] = __builtin_bit_cast(bf16x8, wp[((ks * 8) * 2) * 64]);
        wof[ks][1] = __builtin_bit_cast(bf16x8, wp[((ks * 8) * 2 + 1) * 64]);
      }
    }
    __syncthreads();   // the exchange is visible; the previous pair's O planes / PV partials are dead (its out-proj is done)
    if (wave == 4 || wave == 5) {
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q2][r] += PS[(((wave - 4) * 2 + q2) * 16 + r) * 64 + lane];
    }
    if (wave < 6) {
      const int which = wave % 3, hh = wave / 3, kg = lane >> 5;
      const float mul = which == 0 ? scale : 1.f;
      const float* bq = GB + 2 * LF_D + which * LF_D + (2 * hp + hh) * HD;   // bias of this (q|k|v, head): 32 values
#pragma unroll
      for (int rbk = 0; rbk < 2; ++rbk) {
        if (rbk < nrb) {
          const int tokn = rbk * 32 + (lane & 31);
          if (which < 2) {
            __bf16* ph = QKp + ((hh * 2 + which) * 2) * AT_QPL;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 bv = *(const f32x4*)(bq + 8 * g + 4 * kg);
              split4(ph, ph + AT_QPL, tokn * AT_QP + 8 * g + 4 * kg,
                     f32x4{(acc[rbk][4 * g] + bv[0]) * mul, (acc[rbk][4 * g + 1] + bv[1]) * mul, (acc[rbk][4 * g + 2] + bv[2]) * mul,
                           (acc[rbk][4 * g + 3] + bv[3]) * mul});
            }
          } else {
            __bf16* vh = Vtp + (hh * 2) * AT_VPL;
            const float bch = bq[lane & 31];
#pragma unroll
            for (int sg = 0; sg < 2; ++sg) {
              const f32x4 a0 = {acc[rbk][8 * sg] + bch, acc[rbk][8 * sg + 1] + bch, acc[rbk][8 * sg + 2] + bch, acc[rbk][8 * sg + 3] + bch};
              const f32x4 a1 = {acc[rbk][8 * sg + 4] + bch, acc[rbk][8 * sg + 5] + bch, acc[rbk][8 * sg + 6] + bch, acc[rbk][8 * sg + 7] + bch};
              const bf16x4 h0 = __builtin_convertvector(a0, bf16x4), h1 = __builtin_convertvector(a1, bf16x4);
              const bf16x4 l0 = __builtin_convertvector(a0 - __builtin_convertvector(h0, f32x4), bf16x4);
              const bf16x4 l1 = __builtin_convertvector(a1 - __builtin_convertvector(h1, f32x4), bf16x4);
              const int off = (lane & 31) * AT_VP + rbk * 32 + 16 * sg + 8 * kg;
              *(bf16x8*)(vh + off) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
              *(bf16x8*)(vh + AT_VPL + off) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
          }
        }
      }
    }
    __syncthreads();
    if (hp == 1) ATS(5);
    // ---- attention of both heads of the pair (attn_body's core): wave = (head, query block, key block) ----
    const int hh = wave >> 2, sub = wave & 3;
    const int qb = (nrb == 2) ? (sub >> 1) : 0, kb = (nrb == 2) ? (sub & 1) : 0;
    const bool score_wave = sub < nrb * nrb;
    const __bf16* Qh = QKp + ((hh * 2 + 0) * 2) * AT_QPL;
    const __bf16* Kh = QKp + ((hh * 2 + 1) * 2) * AT_QPL;
    const __bf16* Vh = Vtp + (hh * 2) * AT_VPL;
    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    float m_w = -INFINITY, l_w = 0.f;
    if (score_wave) {
      f32x16 sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
      const int ko = (kb * 32 + (lane & 31)) * AT_QP + 8 * (lane >> 5), qo = (qb * 32 + (lane & 31)) * AT_QP + 8 * (lane >> 5);
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) {
        const bf16x8 kh = *(const bf16x8*)(Kh + ko + 16 * ks), kl = *(const bf16x8*)(Kh + AT_QPL + ko + 16 * ks);
        const bf16x8 qh = *(const bf16x8*)(Qh + qo + 16 * ks), ql = *(const bf16x8*)(Qh + AT_QPL + qo + 16 * ks);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql, sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh, sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh, sacc, 0, 0, 0);
      }
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        sacc[r] = key < L ? sacc[r] : -INFINITY;
        mx = fmaxf(mx, sacc[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sacc[r] = (mx == -INFINITY) ? 0.f : expf(sacc[r] - mx);
        sum += sacc[r];
      }
      sum += __shfl_xor(sum, 32, 64);
      m_w = mx;
      l_w = sum;
      if (lane < 32) {
        SM[wave * 32 + lane] = mx;
        SL[wave * 32 + lane] = sum;
      }
      const int vo = (lane & 31) * AT_VP + kb * 32 + 8 * (lane >> 5);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const f32x4 p0 = {sacc[8 * ks], sacc[8 * ks + 1], sacc[8 * ks + 2], sacc[8 * ks + 3]};
        const f32x4 p1 = {sacc[8 * ks + 4], sacc[8 * ks + 5], sacc[8 * ks + 6], sacc[8 * ks + 7]};
        const bf16x4 h0 = __builtin_convertvector(p0, bf16x4), h1 = __builtin_convertvector(p1, bf16x4);
        const bf16x4 l0 = __builtin_convertvector(p0 - __builtin_convertvector(h0, f32x4), bf16x4);
        const bf16x4 l1 = __builtin_convertvector(p1 - __builtin_convertvector(h1, f32x4), bf16x4);
        const bf16x8 ph = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 pl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 vh = *(const bf16x8*)(Vh + vo + 16 * ks), vl = *(const bf16x8*)(Vh + AT_VPL + vo + 16 * ks);
        oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, oacc, 0, 0, 0);
        oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, oacc, 0, 0, 0);
        oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, oacc, 0, 0, 0);
      }
    }
    if (hp == 1) ATS(6);
    __syncthreads();   // every wave is done with q, k, v^T: the PV partials take the q / k planes, the O planes the v^T planes
    if (score_wave) {
      float fsc;
      if (nrb == 2) {
        const float m_o = SM[(wave ^ 1) * 32 + (lane & 31)], l_o = SL[(wave ^ 1) * 32 + (lane & 31)];
        const float mg = fmaxf(m_w, m_o);
        const float fw = (m_w == -INFINITY) ? 0.f : expf(m_w - mg), fo = (m_o == -INFINITY) ? 0.f : expf(m_o - mg);
        fsc = fw / (l_w * fw + l_o * fo);
      } else {
        fsc = 1.0f / l_w;
      }
      float* od = OT + ((wave * 32) + (lane & 31)) * QSTR + 4 * (lane >> 5);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(f32x4*)(od + 8 * g) = f32x4{oacc[4 * g] * fsc, oacc[4 * g + 1] * fsc, oacc[4 * g + 2] * fsc, oacc[4 * g + 3] * fsc};
    }
    __syncthreads();
    {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int idx = t + LF_NT * it;
        const int row = idx >> 4, h2 = (idx >> 3) & 1, q4 = idx & 7;
        const int qbr = row >> 5, qq = row & 31;
        f32x4 v = zero4;
        if (row < nrb * 32 && row >= L - Lq && row < L) {
          const int w0 = h2 * 4 + (nrb == 2 ? 2 * qbr : 0);
          v = *(const f32x4*)(OT + (w0 * 32 + qq) * QSTR + 4 * q4);
          if (nrb == 2) v += *(const f32x4*)(OT + ((w0 + 1) * 32 + qq) * QSTR + 4 * q4);
        }
        split4(Oh, Ol, row * A2_OP + h2 * 32 + 4 * q4, v);
      }
    }
    __syncthreads();
    // ---- out-projection partial of this head pair: p^T = Wo[32 w .. 32 w + 31, 64 hp : 64 hp + 64] . [o_h0 | o_h1]^T (+ the parked
    //      residual and the bias on the pair that owns these columns); running sum s = hp == 0 ? p : s + p; the last pair
    //      writes the finished rows over the parked residual (a wave reads and writes only ITS 32 columns) ----
    if (hp == 1) ATS(7);
    const bool mine = (wave >> 1) == hp;
#pragma unroll
    for (int rbk = 0; rbk < 2; ++rbk) {
      if (rbk >= nrb || rbk * 32 + 32 <= nq0) continue;   // no query rows in this block (uniform)
      const int row = rbk * 32 + (lane & 31);
      const bool ok = row >= nq0 && row < L;
      float* o = A.ap + ((long long)b * Lq + (min(max(row, nq0), L - 1) - nq0)) * d + wave * 32 + 4 * (lane >> 5);
      float* sb = o + A.ap_stride;
      // unconditional loads (a branch around a load drains the memory pipeline): the parked residual, the bias, the running sum
      f32x4 xr[4], sv[4], bv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        xr[g] = *(const f32x4*)(o + 8 * g);
        sv[g] = *(const f32x4*)(sb + 8 * g);
        bv[g] = *(const f32x4*)(A.bo + wave * 32 + 8 * g + 4 * (lane >> 5));
      }
      f32x16 pacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
      const int ao = row * A2_OP + 8 * (lane >> 5);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 xh = *(const bf16x8*)(Oh + ao + ks * 16), xl = *(const bf16x8*)(Ol + ao + ks * 16);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wof[ks][0], xl, pacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wof[ks][1], xh, pacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wof[ks][0], xh, pacc, 0, 0, 0);
      }
      float* dstp = hp == LF_NH / 2 - 1 ? o : sb;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 p = {pacc[4 * g], pacc[4 * g + 1], pacc[4 * g + 2], pacc[4 * g + 3]};
        const f32x4 pm = p + (xr[g] + bv[g]);
        f32x4 v, r;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = mine ? pm[q] : p[q];
        const f32x4 sp = sv[g] + v;
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = hp == 0 ? v[q] : sp[q];
        if (ok) *(f32x4*)(dstp + 8 * g) = r;
      }
    }
    if (hp == 1) ATS(8);
    // (the barrier behind the next pair's projection orders these LDS reads before its plane writes)
  }
  ATS(9);
}

// ================================================================================================
// Packed FFN weights (sf_pack_ffn_weights): both matrices pre-split into bf16 hi/lo and stored in the order the
// MFMA B-operand fragments are consumed, so that a wave loads its fragments straight from memory into registers
// (1 KB contiguous per wave-wide load) and the weights never pass through LDS:
//     uint4 index = ((((c * 16 + ks) * 8 + wave) * 2 + plane) * 64 + lane),   8 bf16 per uint4
//     lin1_packed: element j = W1[c*256 + wave*32 + (lane & 31)][ks*16 + 8*(lane >> 5) + j]
//     lin2_packed: element j = W2[wave*32 + (lane & 31)][c*256 + ks*16 + 8*(lane >> 5) + j]
// c = hidden chunk (4), ks = 16-wide k step (16), plane 0 = hi, 1 = lo.
__global__ void pack_ffn_kernel(const float* __restrict__ w1, const float* __restrict__ w2, uint4* __restrict__ p1,
                                uint4* __restrict__ p2, int ffn) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // one uint4 of each matrix
  const int total = (ffn / LF_HC) * 16 * 8 * 2 * 64;
  if (idx >= total) return;
  const int lane = idx & 63, plane = (idx >> 6) & 1, wave = (idx >> 7) & 7, ks = (idx >> 10) & 15, c = idx >> 14;
  const int nl = wave * 32 + (lane & 31), k0 = ks * 16 + 8 * (lane >> 5);
  const float* s1 = w1 + (long long)(c * LF_HC + nl) * LF_D + k0;
  const float* s2 = w2 + (long long)nl * ffn + c * LF_HC + k0;
  union {
    __bf16 h[8];
    uint4 u;
  } o1, o2;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float a = s1[j], bq = s2[j];
    const __bf16 ah = (__bf16)a, bh = (__bf16)bq;
    o1.h[j] = plane ? (__bf16)(a - (float)ah) : ah;
    o2.h[j] = plane ? (__bf16)(bq - (float)bh) : bh;
  }
  p1[idx] = o1.u;
  p2[idx] = o2.u;
}

// ---- step boundary (slotformer.py:121-124, then :115 of the next step) on 32 finished rows held as split-bf16 planes ----
//   pred = y . Wout^T + b_out -> slots[b][frame][n][0:128];  proj = pred . Win^T + b_in -> ring[b][frame % R][n][0:256]
// Used by step_boundary_kernel and, fused, by the last-arriving workgroup of ffn_partial_kernel on the last layer.
struct SbArgs {
  const uint4* wout_p;
  const float* b_out;
  const uint4* win_p;
  const float* b_in;
  float* slots;
  long long slots_bs, slots_off;
  float* ring;
  long long ring_bs, ring_off;
  int nslots, enabled;
  unsigned* seam_flags;   // non-NULL: ring rows are written through and seam_flags[tile] = seam_epoch is published afterwards
  unsigned seam_epoch;
};
constexpr int SB_C = 128;                    // slot size
constexpr int SB_YP = LF_D + 8, SB_PP = SB_C + 8;

struct SbFrags {
  bf16x8 w1[8][2], w2[8][2];
  f32x4 bo4[4], bi4[4];
};

__device__ __forceinline__ void sb_load(const SbArgs& a, SbFrags& f, int lane, int wave) {
  const int nb1 = wave & 3, kh = wave >> 2;
  const int c1 = nb1 * 32 + 4 * (lane >> 5), c2 = wave * 32 + 4 * (lane >> 5);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f.bo4[g] = *(const f32x4*)(a.b_out + c1 + 8 * g);
    f.bi4[g] = *(const f32x4*)(a.b_in + c2 + 8 * g);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      f.w1[k][pl] = __builtin_bit_cast(bf16x8, a.wout_p[((((long long)(kh * 8 + k) * 4 + nb1) * 2 + pl) * 64) + lane]);
      f.w2[k][pl] = __builtin_bit_cast(bf16x8, a.win_p[((((long long)k * 8 + wave) * 2 + pl) * 64) + lane]);
    }
}

// Y planes [32][SB_YP] hold the rows (all threads must have passed a barrier after filling them); R [4][16][64] f32 and
// P planes [32][SB_PP] are scratch.  Contains barriers: call from all 512 threads.
__device__ __forceinline__ void sb_compute(const SbArgs& a, const SbFrags& f, const __bf16* Yh, const __bf16* Yl, float* R,
                                           __bf16* Ph, __bf16* Pl, int row0, int M, int lane, int wave) {
  const int tok = lane & 31, nb1 = wave & 3, kh = wave >> 2;
  const int c1 = nb1 * 32 + 4 * (lane >> 5), c2 = wave * 32 + 4 * (lane >> 5);
  const int m = row0 + tok;
  const int mb = min(m, M - 1) / a.nslots, mn = min(m, M - 1) - mb * a.nslots;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  {
    const int ao = tok * SB_YP + 8 * (lane >> 5) + kh * 128;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bf16x8 xh = *(const bf16x8*)(Yh + ao + k * 16), xl = *(const bf16x8*)(Yl + ao + k * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w1[k][0], xl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w1[k][1], xh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w1[k][0], xh, acc, 0, 0, 0);
    }
  }
  if (kh == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) R[(nb1 * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (kh == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = acc[4 * g + q] + R[(nb1 * 16 + 4 * g + q) * 64 + lane] + f.bo4[g][q];
      split4(Ph, Pl, tok * SB_PP + c1 + 8 * g, v);
      if (m < M) *(f32x4*)(a.slots + mb * a.slots_bs + a.slots_off + (long long)mn * SB_C + c1 + 8 * g) = v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  {
    const int ao = tok * SB_PP + 8 * (lane >> 5);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bf16x8 xh = *(const bf16x8*)(Ph + ao + k * 16), xl = *(const bf16x8*)(Pl + ao + k * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w2[k][0], xl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w2[k][1], xh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w2[k][0], xh, acc, 0, 0, 0);
    }
  }
  if (m < M) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = {acc[4 * g] + f.bi4[g][0], acc[4 * g + 1] + f.bi4[g][1], acc[4 * g + 2] + f.bi4[g][2],
                       acc[4 * g + 3] + f.bi4[g][3]};
      const long long eo = mb * a.ring_bs + a.ring_off + (long long)mn * LF_D + c2 + 8 * g;
      if (a.seam_flags)   // read by another workgroup of this launch: write through (sc1), no L2-resident dirty line
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), __builtin_amdgcn_make_buffer_rsrc(a.ring, 0, 0x7fffffff, 0x00020000),
                                               (unsigned)(eo * 4), 0, 16);
      else
        *(f32x4*)(a.ring + eo) = v;
    }
  }
}

// ap [8][M][256] head partials -> xout [M][256] finished layer output (xp [4][M][256]: chunk-partial scratch);
// grid = ffn_blocks(tiles) workgroups, tile = 32 rows, 4 hidden chunks per tile.
//
// Both GEMMs run "transposed" (weights as the MFMA A operand, activations as B): a lane then owns 4 CONSECUTIVE
// output columns of one token, so the hidden activations go to LDS as packed 8-byte stores and the results leave
// as 16-byte vectors.
struct FfnArgs {
  const float* ap;
  long long ap_stride;
  const float *ln_g, *ln_b;
  float ln_eps;
  const uint4* w1p;
  const float* b1;
  const uint4* w2p;
  const float* b2;
  float* xp;
  long long xp_stride;
  float* xout;
  int* counters;
  int ntiles, M, dbg;
  int parts_only;   // the chunk partials are the output (the next attention sums them, attn_body<.., PART>): no reduction here
  int np;           // input partials per row: LF_NP (head-pair partials of attn_oproj_kernel) or 1 (finished rows of attn_all_kernel)
  SbArgs sb;
};

template <int NP>   // input partials per row (LF_NP, or 1: finished rows)
__device__ __forceinline__ void ffn_body(const FfnArgs& F, const int blk) {
  // no floating-point contraction in this function: whether the compiler fuses a*b + c into an fma may differ between two
  // instantiations / variants of the same source, and the variants must produce the same bits (LayerNorm statistics)
#pragma clang fp contract(off)
  const float* __restrict__ ap = F.ap;
  const long long ap_stride = F.ap_stride;
  const float* __restrict__ ln_g = F.ln_g;
  const float* __restrict__ ln_b = F.ln_b;
  const float ln_eps = F.ln_eps;
  const uint4* __restrict__ w1p = F.w1p;
  const float* __restrict__ b1 = F.b1;
  const uint4* __restrict__ w2p = F.w2p;
  const float* __restrict__ b2 = F.b2;
  float* __restrict__ xp = F.xp;
  const long long xp_stride = F.xp_stride;
  float* __restrict__ xout = F.xout;
  int* __restrict__ counters = F.counters;
  const int ntiles = F.ntiles, M = F.M, dbg = F.dbg;
  const SbArgs& sb = F.sb;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int s_last;
  __bf16* Ah = (__bf16*)smem;                  // [32][FB_AP]  LN2(x2)
  __bf16* Al = Ah + FB_ROWS * FB_AP;
  __bf16* Hh = Al + FB_ROWS * FB_AP;           // [32][FB_AP]  relu(h_c)
  __bf16* Hl = Hh + FB_ROWS * FB_AP;
  float* X2 = (float*)(Hl + FB_ROWS * FB_AP);  // [32][FB_XP]  x2 (residual), later the output tile
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // block b runs on XCD b % 8: the four hidden chunks of a row tile share an XCD (one L2 fetch of the head partials);
  // tiles are dealt round-robin over the XCDs
  // -- whole groups of 8 tiles.  The ntiles % 8 tiles left over would put 4 more workgroups on the first XCDs only
  // (24 on XCDs 0 and 1 against 20 elsewhere for the 42 tiles of a B = 32 window): behind whole groups their chunks are
  // dealt one per XCD instead (21 everywhere -- the launch then fits 21 CUs per XCD; the partial exchange is
  // write-through, so a tile whose chunks sit in different XCDs is only a little slower, never wrong).
  const int full = (ntiles >> 3) << 5;
  int c = (blk >> 3) & (LF_NCH - 1), tile = (blk >> 5) * 8 + (blk & 7);
  if (full > 0 && blk >= full) {
    c = (blk - full) & (LF_NCH - 1);
    tile = (ntiles & ~7) + ((blk - full) >> 2);
  }
  if (tile >= ntiles) return;
  const int row0 = tile * FB_ROWS;
  // the residual x2 and the bias b2 ride on chunk 0 -- a rule that does not depend on where a row sits in the batch, so a
  // video's bits are the same however videos are grouped into batches / rollout units
  const bool carry = c == 0;
  LF_TS(0);

  // small parameter vectors first: vmcnt retires in order, so a late request would wait for every weight fragment
  const int tok = lane & 31, nb = wave * 32 + 4 * (lane >> 5);   // FFN outputs: this lane's token and first column (+ 8 g)
  f32x4 b1v[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) b1v[g] = *(const f32x4*)(b1 + c * LF_HC + nb + 8 * g);
  // ---- weight fragments: wf[ks][plane], 16 k-steps; FFN1's are requested in the prologue, FFN2's as FFN1 consumes them ----
  const uint4* w1c = w1p + ((long long)(c * 16) * 8 + wave) * 128 + lane;
  const uint4* w2c = w2p + ((long long)(c * 16) * 8 + wave) * 128 + lane;
  bf16x8 wf[16][2];
  auto ldw = [&](const uint4* base, int ks, int plane) {
    return __builtin_bit_cast(bf16x8, base[(ks * 8) * 128 + plane * 64]);
  };
  // ---- prologue.  The workgroup ingests ~100 GB/s through one in-order request path and a wave cannot start the
  //      LayerNorm arithmetic before its own last request has been accepted: the head-pair partials go first, then only
  //      k-steps 0..7 of W1; x2 = sum of the partials, LayerNorm, LN2 planes; k-steps 8..15 are requested after the
  //      arithmetic and land while FFN1 consumes the first half.  Wave `wave` owns rows wave + 8 i, lane = float4 column. ----
  const f32x4 lng = *(const f32x4*)(ln_g + 4 * lane), lnb = *(const f32x4*)(ln_b + 4 * lane);
  f32x4 x2[4];
  {
    f32x4 pr[NP][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gr = min(row0 + wave + 8 * i, M - 1);
      const float* p = ap + (long long)gr * LF_D + 4 * lane;
#pragma unroll
      for (int q = 0; q < NP; ++q) pr[q][i] = *(const f32x4*)(p + ((dbg & 1) ? 0 : q) * ap_stride);
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      wf[ks][0] = ldw(w1c, ks, 0);
      wf[ks][1] = ldw(w1c, ks, 1);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 s = pr[0][i];
#pragma unroll
      for (int q = 1; q < NP; ++q) s += pr[q][i];
      x2[i] = s;
    }
  }
  LF_TS(1);
  {
    const f32x4 g = lng, be = lnb;
    float mean[4], rstd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mean[i] = sf_sum64((x2[i][0] + x2[i][1]) + (x2[i][2] + x2[i][3])) * (1.0f / LF_D);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 dv = x2[i] - mean[i];
      rstd[i] = 1.0f / sqrtf(sf_sum64((dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3])) * (1.0f / LF_D) + ln_eps);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = wave + 8 * i;
      split4(Ah, Al, r * FB_AP + 4 * lane, (x2[i] - mean[i]) * rstd[i] * g + be);
      if (carry) *(f32x4*)(X2 + r * FB_XP + 4 * lane) = x2[i];
    }
  }
  __builtin_amdgcn_sched_barrier(0);   // keep these requests BEHIND the arithmetic above in the instruction stream
#pragma unroll
  for (int ks = 8; ks < 16; ++ks) {
    wf[ks][0] = ldw(w1c, ks, 0);
    wf[ks][1] = ldw(w1c, ks, 1);
  }
  __syncthreads();
  LF_TS(2);

  // ---- FFN1 (transposed): acc[4g+q] = h[token = lane&31][hidden = 32 wave + 8g + 4(lane>>5) + q] ----
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int ao = (lane & 31) * FB_AP + 8 * (lane >> 5);
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const bf16x8 xh = *(const bf16x8*)(Ah + ao + ks * 16), xl = *(const bf16x8*)(Al + ao + ks * 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][0], xl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][1], xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][0], xh, acc, 0, 0, 0);
    if (!(dbg & 2)) {   // this step's registers are free again: request the matching FFN2 fragment
      wf[ks][0] = ldw(w2c, ks, 0);
      wf[ks][1] = ldw(w2c, ks, 1);
    }
  }
  LF_TS(3);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 bv = b1v[g];
    f32x4 hv;
#pragma unroll
    for (int q = 0; q < 4; ++q) hv[q] = fmaxf(acc[4 * g + q] + bv[q], 0.f);
    split4(Hh, Hl, tok * FB_AP + nb + 8 * g, hv);
  }
  __syncthreads();
  LF_TS(4);

  // ---- FFN2 partial (transposed): acc[4g+q] = y_c[token][out column 32 wave + 8g + 4(lane>>5) + q] ----
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const bf16x8 xh = *(const bf16x8*)(Hh + ao + ks * 16), xl = *(const bf16x8*)(Hl + ao + ks * 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][0], xl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][1], xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][0], xh, acc, 0, 0, 0);
  }
  LF_TS(5);
  SbFrags sbf;

  // ---- output tile -> LDS (in place over the x2 stash; the carrying chunk adds the residual and the bias), re-read row-major ----
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float* o = X2 + tok * FB_XP + nb + 8 * g;
    f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    if (carry) v += *(const f32x4*)o + *(const f32x4*)(b2 + nb + 8 * g);
    *(f32x4*)o = v;
  }
  __syncthreads();
  // thread t owns float4 (row = wave + 8 i, column 4 lane): 1 KB contiguous per wave-wide access
  const __amdgpu_buffer_rsrc_t xpr = __builtin_amdgcn_make_buffer_rsrc(xp, 0, 0x7fffffff, 0x00020000);
  f32x4 mine[4];
  unsigned off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave + 8 * i;
    mine[i] = *(const f32x4*)(X2 + r * FB_XP + 4 * lane);
    off[i] = (unsigned)(((long long)min(row0 + r, M - 1) * LF_D + 4 * lane) * 4);
  }
  // ---- chunk partial -> memory; the LAST of the tile's four workgroups to arrive sums them in fixed order
  //      (c = 0..3, so the result does not depend on which one is last) and writes the finished rows.
  //      Write-through (sc1) stores put the values at the device coherence point once acknowledged, without the
  //      whole-L2 write-back a __threadfence() would cost (~30 us per launch here). ----
  const unsigned cstride = (unsigned)(xp_stride * 4);
  if (F.parts_only) {   // plain stores: the kernel boundary publishes them
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (row0 + wave + 8 * i < M)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mine[i]), xpr, off[i] + c * cstride, 0, 0);
    LF_TS(8);
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (row0 + wave + 8 * i < M)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mine[i]), xpr, off[i] + c * cstride, 0, 16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every partial store of this thread is acknowledged
  __syncthreads();                                    // ... and of the whole workgroup
  LF_TS(6);
  // last layer of a rollout step: the fragments of the fused step boundary are requested by every workgroup now -- after
  // the store acknowledgement (vmcnt retires in order: requested earlier they would sit in front of the stores), early
  // enough to land while the arrival counter and the other chunks' partials are fetched (requesting them only in the
  // last arriver was measured slower: 18.4 vs 17.1 us)
  if (sb.enabled) sb_load(sb, sbf, lane, wave);
  if (t == 0)
    s_last = (__hip_atomic_fetch_add(counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == LF_NCH - 1);
  __syncthreads();
  LF_TS(7);
  if (s_last) {
    f32x4 oth[LF_NCH][4];
#pragma unroll
    for (int cc = 0; cc < LF_NCH; ++cc)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        oth[cc][i] = (cc == c) ? mine[i]
                               : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xpr, off[i] + cc * cstride, 0, 16));
    if (t == 0) __hip_atomic_store(counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    if (!sb.enabled) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 sum = ((oth[0][i] + oth[1][i]) + oth[2][i]) + oth[3][i];
        if (row0 + wave + 8 * i < M) *(f32x4*)(xout + (long long)(row0 + wave + 8 * i) * LF_D + 4 * lane) = sum;
      }
    } else {
      // fused step boundary: the finished rows go straight into split-bf16 planes (over the dead LN2 planes), then
      // out-proj -> slots and in-proj -> ring, by this workgroup alone (s_last is uniform: barriers are safe)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        split4(Ah, Al, (wave + 8 * i) * FB_AP + 4 * lane, ((oth[0][i] + oth[1][i]) + oth[2][i]) + oth[3][i]);
      __syncthreads();
      sb_compute(sb, sbf, Ah, Al, (float*)Hh, Hh + 4 * 16 * 64 * 2, Hh + 4 * 16 * 64 * 2 + 32 * SB_PP, row0, M, lane, wave);
      if (sb.seam_flags) {
        // publish the tile: every storing wave drains its write-through ring stores, then ONE lane raises the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) __hip_atomic_store(sb.seam_flags + tile, sb.seam_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  LF_TS(8);
}

__device__ unsigned long long lf_wg[4 * 64];   // SF_DBG=lf=32: {HW_ID, XCC_ID, start, end} of the first 64 FFN workgroups
template <int NP>
__global__ __launch_bounds__(LF_NT) void ffn_partial_kernel(FfnArgs F) {
  const unsigned long long t0 = (F.dbg & 32) ? wall_clock64() : 0ull;
  ffn_body<NP>(F, blockIdx.x);
  if ((F.dbg & 32) && threadIdx.x == 0 && blockIdx.x < 64) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    lf_wg[4 * blockIdx.x] = hw;
    lf_wg[4 * blockIdx.x + 1] = xcc;
    lf_wg[4 * blockIdx.x + 2] = t0;
    lf_wg[4 * blockIdx.x + 3] = wall_clock64();
  }
}

// One launch for the seam between two rollout steps: blocks [0, nffn) run the last layer's FFN + step boundary of step s,
// the others the layer-0 attention of step s+1 (one per (head pair, video)), which requests its weights and the five old
// frames of its window at once and waits only for the 7 new rows of ITS video.  The producers have the lower block indices
// (dispatched first); all nffn + 4 B workgroups are co-resident at one per CU on the 168-CU rollout partition for B <= 32.
__global__ __launch_bounds__(LF_NT) void seam_kernel(FfnArgs F, AttnArgs A, SeamArgs seam, int nffn) {
  if ((int)blockIdx.x < nffn) {
    ffn_body<LF_NP>(F, blockIdx.x);
  } else {
    const int u = blockIdx.x - nffn;
    attn_body<true, true>(A, u & 3, u >> 2, seam);
  }
}

// ================================================================================================
// Generic fragment-order packing of an nn.Linear weight W [N][K] (N % 32 == 0, K % 16 == 0):
//     uint4 index = (((ks * (N/32) + nb) * 2 + plane) * 64 + lane),  element j = W[nb*32 + (lane & 31)][ks*16 + 8*(lane >> 5) + j]
__global__ void pack_linear_kernel(const float* __restrict__ w, uint4* __restrict__ out, int N, int K) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int NB = N / 32, total = (K / 16) * NB * 2 * 64;
  if (idx >= total) return;
  const int lane = idx & 63, plane = (idx >> 6) & 1, nb = (idx >> 7) % NB, ks = (idx >> 7) / NB;
  const float* src = w + (long long)(nb * 32 + (lane & 31)) * K + ks * 16 + 8 * (lane >> 5);
  union {
    __bf16 h[8];
    uint4 u;
  } o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float a = src[j];
    const __bf16 ah = (__bf16)a;
    o.h[j] = plane ? (__bf16)(a - (float)ah) : ah;
  }
  out[idx] = o.u;
}

// Stand-alone step boundary (one workgroup per 32 rows); y [M = B*rows_per_video][256] are the finished rows of the last
// layer.  With proj_only != 0 the out-projection is skipped and rows that are already in `slots` are in-projected (the
// burn-in frames, rows per video = n_in * N) -- the same arithmetic as inside the rollout, so a rollout restarted from
// its own output reproduces the original bit for bit.
constexpr size_t SB_LDS = (size_t)2 * 32 * SB_YP * 2 + (size_t)4 * 16 * 64 * 4 + (size_t)2 * 32 * SB_PP * 2;
__global__ __launch_bounds__(LF_NT) void step_boundary_kernel(const float* __restrict__ y, SbArgs sb, int M, int proj_only) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* Yh = (__bf16*)smem;                 // [32][SB_YP]
  __bf16* Yl = Yh + 32 * SB_YP;
  float* R = (float*)(Yl + 32 * SB_YP);       // [4][16][64] k-half exchange
  __bf16* Ph = (__bf16*)(R + 4 * 16 * 64);    // [32][SB_PP]
  __bf16* Pl = Ph + 32 * SB_PP;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int row0 = blockIdx.x * 32;
  SbFrags f;
  sb_load(sb, f, lane, wave);
  if (proj_only) {
    // slot rows -> split-bf16 planes directly (2 float4 per thread: 32 rows x 32 float4), then only the in-projection
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = t + LF_NT * i, r = idx >> 5, q4 = idx & 31;
      const int gm = min(row0 + r, M - 1), gb = gm / sb.nslots, gn = gm - gb * sb.nslots;
      const f32x4 v = *(const f32x4*)(sb.slots + gb * sb.slots_bs + sb.slots_off + (long long)gn * SB_C + 4 * q4);
      split4(Ph, Pl, r * SB_PP + 4 * q4, v);
    }
    __syncthreads();
    const int tok = lane & 31, c2 = wave * 32 + 4 * (lane >> 5);
    const int m = row0 + tok;
    const int mb = min(m, M - 1) / sb.nslots, mn = min(m, M - 1) - mb * sb.nslots;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int ao = tok * SB_PP + 8 * (lane >> 5);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bf16x8 xh = *(const bf16x8*)(Ph + ao + k * 16), xl = *(const bf16x8*)(Pl + ao + k * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w2[k][0], xl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w2[k][1], xh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w2[k][0], xh, acc, 0, 0, 0);
    }
    if (m < M) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = {acc[4 * g] + f.bi4[g][0], acc[4 * g + 1] + f.bi4[g][1], acc[4 * g + 2] + f.bi4[g][2],
                         acc[4 * g + 3] + f.bi4[g][3]};
        *(f32x4*)(sb.ring + mb * sb.ring_bs + sb.ring_off + (long long)mn * LF_D + c2 + 8 * g) = v;
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave + 8 * i;
    const f32x4 v = *(const f32x4*)(y + (long long)min(row0 + r, M - 1) * LF_D + 4 * lane);
    split4(Yh, Yl, r * SB_YP + 4 * lane, v);
  }
  __syncthreads();
  sb_compute(sb, f, Yh, Yl, R, Ph, Pl, row0, M, lane, wave);
}

// ------------------------------------------------------------------------------------------------
// Wide variants of the chunk-partial FFN (per-call option ffn_rows / sf_set_ffn_rows64): one workgroup runs NH 64-row
// halves -- 64 or 128 rows -- against ONE load of its 256-wide chunk of W1 / W2.  Every W1 fragment stays in registers
// until the LAST half has consumed it and is then replaced by the matching W2 fragment (the W2 requests ride behind the
// MFMAs of that half).  Per 128 rows a workgroup ingests 524 KB of weights + 512 KB of head-pair partials instead of
// 4 x 652 KB with 32-row tiles, and the launch has a quarter of the workgroups: with several rollouts sharing the CUs
// (pipeline partitions 'pair' / 'pair2') the CUs are the bound, and the FFN's CU time per row drops by 40 % (64 rows) /
// ~60 % (128 rows).  Same arithmetic per row as ffn_body (k order, operands, LayerNorm), so the partials are
// bit-identical.
// LDS: per half one 67.6 KB region in which the LN2 planes, the hidden planes and the f32 output tile take turns.  The
// residual needs neither LDS nor registers: chunk 0 (which carries x2 + b2, as in ffn_body) stores x2 + b2 into its OUTPUT
// rows while it normalises them, and the thread that stored a float4 reads it back behind the last FFN2 MFMA (into the
// registers the weight fragments have left) and adds the FFN2 partial: y_0 + (x2 + b2), the same sum as ffn_body's.
// Prologue: the head-pair partials arrive in 32-row quarters through two register buffers (2 x 64 VGPRs), the first
// half of W1 is requested behind the first two quarters, the second half once the last quarter is in flight.
constexpr int F6_ROWS = 64;
constexpr size_t F6_R1 = (size_t)2 * F6_ROWS * FB_AP * 2;                      // planes (hi, lo) of one half >= its f32 output tile
static_assert((size_t)F6_ROWS * FB_XP * 4 <= F6_R1, "output tile must fit the plane region");

template <int NH, int NP>
__global__ __launch_bounds__(LF_NT) void ffn_wide_parts_kernel(FfnArgs F) {
  // no floating-point contraction in this function: whether the compiler fuses a*b + c into an fma may differ between two
  // instantiations / variants of the same source, and the variants must produce the same bits (LayerNorm statistics)
#pragma clang fp contract(off)
  constexpr int ROWS = F6_ROWS * NH, NQ = 2 * NH;
  const float* __restrict__ ap = F.ap;
  const long long ap_stride = F.ap_stride;
  const uint4* __restrict__ w1p = F.w1p;
  const uint4* __restrict__ w2p = F.w2p;
  const int M = F.M;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  auto PH = [&](int h) { return (__bf16*)((char*)smem + (size_t)h * F6_R1); };   // [64][FB_AP] hi plane of half h (lo follows)
  auto OTH = [&](int h) { return (float*)((char*)smem + (size_t)h * F6_R1); };   // [64][FB_XP] output tile of half h
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // row tiles dealt like ffn_body's: the four chunks of a tile share an XCD; left-over tiles one chunk per XCD
  const int ntw = (M + ROWS - 1) / ROWS, blk = blockIdx.x;
  const int full = (ntw >> 3) << 5;
  int c = (blk >> 3) & (LF_NCH - 1), tp = (blk >> 5) * 8 + (blk & 7);
  if (full > 0 && blk >= full) {
    c = (blk - full) & (LF_NCH - 1);
    tp = (ntw & ~7) + ((blk - full) >> 2);
  }
  if (tp >= ntw) return;
  const int row0 = tp * ROWS;
#define WTS(i) do { if ((F.dbg & 16) && blockIdx.x == 0 && threadIdx.x == 0) lf_ts[32 + i] = wall_clock64(); } while (0)
  WTS(0);
  const int tok = lane & 31, nb = wave * 32 + 4 * (lane >> 5);
  // every global access below is a buffer access: ONE 32-bit lane offset per stream (weights, partial rows, output rows) plus
  // scalar offsets, instead of dozens of 64-bit lane addresses -- the registers are needed for data in flight
  const __amdgpu_buffer_rsrc_t w1c = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(w1p), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2c = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(w2p), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t apr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ap), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t xpr = __builtin_amdgcn_make_buffer_rsrc(F.xp, 0, 0x7fffffff, 0x00020000);
  const unsigned wlane = (unsigned)((wave * 128 + lane) * 16);
  bf16x8 wf[16][2];
  auto ldw = [&](const __amdgpu_buffer_rsrc_t& base, int ks, int plane) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(base, wlane, (unsigned)((((c * 16 + ks) * 8) * 128 + plane * 64) * 16), 0));
  };
  const f32x4 lng = *(const f32x4*)(F.ln_g + 4 * lane), lnb = *(const f32x4*)(F.ln_b + 4 * lane);
  // ---- prologue (wave owns rows wave + 8 i of each 32-row quarter, lane = float4 column) ----
  f32x4 pr[2][NP][4];
  const f32x4 b2v = *(const f32x4*)(F.b2 + 4 * lane);
  // thread t owns float4 (row = wave + 8 i, column 4 lane) of the input partials and of the output: byte offset of (tile row r)
  const unsigned aps = (unsigned)(ap_stride * 4), xps = (unsigned)((long long)c * F.xp_stride * 4);
  auto rowoff = [&](int r) { return (unsigned)(((row0 + r) * LF_D + 4 * lane) * 4); };
  auto request = [&](int q, f32x4 (&dst)[NP][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned off = (unsigned)((min(row0 + 32 * q + wave + 8 * i, M - 1) * LF_D + 4 * lane) * 4);
#pragma unroll
      for (int pq = 0; pq < NP; ++pq) dst[pq][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(apr, off, pq * aps, 0));
    }
  };
  auto consume = [&](int q, const f32x4 (&src)[NP][4]) {
#pragma clang fp contract(off)
    f32x4 x2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 sacc = src[0][i];
#pragma unroll
      for (int pq = 1; pq < NP; ++pq) sacc += src[pq][i];
      x2[i] = sacc;
    }
    float mean[4], rstd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mean[i] = sf_sum64((x2[i][0] + x2[i][1]) + (x2[i][2] + x2[i][3])) * (1.0f / LF_D);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 dv = x2[i] - mean[i];
      rstd[i] = 1.0f / sqrtf(sf_sum64((dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3])) * (1.0f / LF_D) + F.ln_eps);
    }
    __bf16* Ph = PH(q >> 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 32 * (q & 1) + wave + 8 * i;
      split4(Ph, Ph + F6_ROWS * FB_AP, r * FB_AP + 4 * lane, (x2[i] - mean[i]) * rstd[i] * lng + lnb);
      // chunk 0 carries the residual and the bias (ffn_body's rule): parked in this thread's own output float4
      if (c == 0 && row0 + 32 * q + wave + 8 * i < M)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x2[i] + b2v), xpr, rowoff(32 * q + wave + 8 * i), xps, 0);
    }
  };
  const int ao = (lane & 31) * FB_AP + 8 * (lane >> 5);
  f32x4 b1v[4];
  auto load_b1 = [&]() {   // the FFN1 bias of this lane's hidden columns
#pragma unroll
    for (int g = 0; g < 4; ++g) b1v[g] = *(const f32x4*)(F.b1 + c * LF_HC + nb + 8 * g);
  };
  auto load_w1 = [&](int k0, int k1) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      if (ks >= k0 && ks < k1) {
        wf[ks][0] = ldw(w1c, ks, 0);
        wf[ks][1] = ldw(w1c, ks, 1);
      }
  };
  // FFN1 of half h over k-steps [k0, k1); REPL: every fragment's registers then take the matching W2 fragment
  auto ffn1 = [&](int h, int k0, int k1, f32x16& a0, f32x16& a1, bool repl) {
    const __bf16* Ph = PH(h);
    const __bf16* Pl = Ph + F6_ROWS * FB_AP;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks < k0 || ks >= k1) continue;
      const bf16x8 xh0 = *(const bf16x8*)(Ph + ao + ks * 16), xl0 = *(const bf16x8*)(Pl + ao + ks * 16);
      const bf16x8 xh1 = *(const bf16x8*)(Ph + ao + 32 * FB_AP + ks * 16), xl1 = *(const bf16x8*)(Pl + ao + 32 * FB_AP + ks * 16);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][0], xl0, a0, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][1], xh0, a0, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][0], xh0, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][0], xl1, a1, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][1], xh1, a1, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][0], xh1, a1, 0, 0, 0);
      if (repl) {
        wf[ks][0] = ldw(w2c, ks, 0);
        wf[ks][1] = ldw(w2c, ks, 1);
      }
    }
  };
  // relu(h + b1) of half h -> its hidden planes (over its LN2 planes: every wave must have read them)
  auto hidden = [&](int h, const f32x16& a0, const f32x16& a1) {
    __bf16* Hh = PH(h);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bv = b1v[g];
      f32x4 h0, h1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        h0[q] = fmaxf(a0[4 * g + q] + bv[q], 0.f);
        h1[q] = fmaxf(a1[4 * g + q] + bv[q], 0.f);
      }
      split4(Hh, Hh + F6_ROWS * FB_AP, tok * FB_AP + nb + 8 * g, h0);
      split4(Hh, Hh + F6_ROWS * FB_AP, (32 + tok) * FB_AP + nb + 8 * g, h1);
    }
  };
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  if constexpr (NH == 1) {
    // ---- 64 rows: both quarters and k-steps 0..7 of W1 requested up front, k-steps 8..15 behind the LayerNorm arithmetic ----
    request(0, pr[0]);
    WTS(1);
    load_w1(0, 8);
    request(1, pr[1]);
    consume(0, pr[0]);
    consume(1, pr[1]);
    __builtin_amdgcn_sched_barrier(0);   // requests stay BEHIND the arithmetic above in the instruction stream
    load_w1(8, 16);
    load_b1();
    WTS(2);
    __syncthreads();
    ffn1(0, 0, 16, acc0, acc1, true);
    WTS(3);
    __syncthreads();   // every wave has read the LN2 planes: the hidden planes take their place
    hidden(0, acc0, acc1);
  } else {
    // ---- 128 rows.  A workgroup ingests ~1 MB (512 KB of partials, 2 x 256 KB of weights) at ~100 GB/s: as long as the matrix
    //      pipe waits for all of it the launch is twice as long as its MFMAs.  So the SECOND half's partials arrive behind the
    //      first half's FFN1: half A's quarters + k-steps 0..7 of W1 go out first; FFN1 of half A over k-steps 0..7 runs while
    //      half B's quarters land; k-steps 8..15 of W1 are requested behind half B's LayerNorm and consumed after FFN1 of half
    //      B over k-steps 0..7 (whose fragment registers then take W2's); k-steps 8..15 of both halves follow. ----
    f32x16 accB0, accB1;
#pragma unroll
    for (int r = 0; r < 16; ++r) accB0[r] = accB1[r] = 0.f;
    request(0, pr[0]);
    WTS(1);
    request(1, pr[1]);
    load_w1(0, 8);
    consume(0, pr[0]);
    __builtin_amdgcn_sched_barrier(0);
    request(2, pr[0]);
    consume(1, pr[1]);
    __builtin_amdgcn_sched_barrier(0);
    request(3, pr[1]);
    WTS(2);
    __syncthreads();   // LN2 planes of half A complete
    ffn1(0, 0, 8, acc0, acc1, false);
    __builtin_amdgcn_sched_barrier(0);
    consume(2, pr[0]);
    __builtin_amdgcn_sched_barrier(0);
    load_w1(8, 16);
    consume(3, pr[1]);
    load_b1();
    WTS(3);
    __syncthreads();   // LN2 planes of half B complete
    ffn1(1, 0, 8, accB0, accB1, true);
    ffn1(0, 8, 16, acc0, acc1, false);
    ffn1(1, 8, 16, accB0, accB1, true);
    WTS(4);
    __syncthreads();   // every wave has read the LN2 planes: the hidden planes take their place
    hidden(0, acc0, acc1);
    hidden(1, accB0, accB1);
  }
  __syncthreads();
  WTS(5);

  // ---- FFN2 partial, half by half; the f32 output tile of a half replaces its hidden planes ----
  f32x4 res[8 * NH];
#pragma unroll
  for (int i = 0; i < 8 * NH; ++i) res[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const __bf16* Ph = PH(h);
    const __bf16* Pl = Ph + F6_ROWS * FB_AP;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const bf16x8 xh0 = *(const bf16x8*)(Ph + ao + ks * 16), xl0 = *(const bf16x8*)(Pl + ao + ks * 16);
      const bf16x8 xh1 = *(const bf16x8*)(Ph + ao + 32 * FB_AP + ks * 16), xl1 = *(const bf16x8*)(Pl + ao + 32 * FB_AP + ks * 16);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][0], xl0, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][1], xh0, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][0], xh0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][0], xl1, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][1], xh1, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][0], xh1, acc1, 0, 0, 0);
    }
    WTS(6 + h);
    if (h == NH - 1 && c == 0) {
      // the weight fragments are dead: their registers take this thread's parked x2 + b2 back (same thread, same addresses)
#pragma unroll
      for (int i = 0; i < 8 * NH; ++i)
        res[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xpr, rowoff(min(wave + 8 * i, M - 1 - row0)), xps, 0));
    }
    __syncthreads();   // every wave has read the hidden planes of this half
    float* OT = OTH(h);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *(f32x4*)(OT + tok * FB_XP + nb + 8 * g) = f32x4{acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]};
      *(f32x4*)(OT + (32 + tok) * FB_XP + nb + 8 * g) = f32x4{acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]};
    }
  }
  __syncthreads();
  WTS(8);
  // 1 KB contiguous per wave-wide store; the kernel boundary publishes.  Chunk 0: y_0 + (x2 + b2), ffn_body's sum.
#pragma unroll
  for (int i = 0; i < 8 * NH; ++i) {
    const int r = wave + 8 * i;   // row of the tile; half i >> 3
    f32x4 v = *(const f32x4*)(OTH(i >> 3) + (r & 63) * FB_XP + 4 * lane);
    if (c == 0) v += res[i];
    if (row0 + r < M) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), xpr, rowoff(r), xps, 0);
  }
  WTS(9);
}

static int g_ffn_rows64 = 0;
extern "C" int sf_set_ffn_rows64(int on) {
  g_ffn_rows64 = on ? 1 : 0;
  return 0;
}
extern "C" int sf_get_ffn_rows64(void) { return g_ffn_rows64; }

// ------------------------------------------------------------------------------------------------
// timing ablations (wrong results): 1 FFN reads one head partial, 2 FFN skips the W2 loads, 4 attention reads one
// input partial, 8 attention stores one column block
static int lf_dbg() {
  static const int v = sf_dbg("lf");
  return v;
}

bool sf_layer_fused_ok(int d, int heads, int ffn, int L) {
  return d == LF_D && heads == LF_NH && ffn == LF_NCH * LF_HC && L >= 1 && L <= FA_ROWS;
}

static AttnArgs make_attn_args(const float* xin, long long x_batch_stride, const float* pe, int f0, int ring_frames, int nslots,
                               const sf_tfm_layer& w, float eps, float* ap, long long ap_stride, int L, int Lq) {
  AttnArgs A;
  A.xin = xin; A.x_batch_stride = x_batch_stride; A.pe = pe; A.f0 = f0; A.ring_frames = ring_frames; A.nslots = nslots;
  A.ln_g = w.norm1_g; A.ln_b = w.norm1_b; A.ln_eps = eps; A.wqkv_p = (const uint4*)w.attn_in_packed; A.bias = w.in_proj_b;
  A.wo_p = (const uint4*)w.attn_out_packed; A.bo = w.out_proj_b; A.ap = ap; A.ap_stride = ap_stride; A.L = L; A.Lq = Lq;
  A.dbg = lf_dbg();
  A.xparts_stride = 0;
  A.nvideos = 0;
  return A;
}

template <bool RING, bool PART = false>
static int launch_attn(const float* xin, long long x_batch_stride, const float* pe, int f0, int ring_frames, int nslots,
                       const sf_tfm_layer& w, float eps, float* ap, long long ap_stride, int B, int L, int Lq,
                       hipStream_t st, long long xparts_stride = 0) {
  if (!w.attn_in_packed || !w.attn_out_packed)
    return sf_set_err(-1, "invalid argument: fused attention needs packed weights (sf_pack_attn_weights)", __FILE__, __LINE__);
  if ((long long)LF_NP * ap_stride * 4 >= 0x7fffffffLL)
    return sf_set_err(-1, "invalid argument: head-pair partial buffer beyond the 2 GB a buffer descriptor addresses", __FILE__, __LINE__);
  auto kern = attn_oproj_kernel<RING, PART>;
  SF_TRY(sf_ensure_dyn_lds((const void*)kern, (size_t)(A2_LDS)));
  sf_prof_begin(SF_K_MHA, st, 6.0 * B * L * (double)LF_D * LF_D + 4.0 * (double)B * LF_NH * Lq * L * LF_HD +
                                  2.0 * B * Lq * (double)LF_D * LF_D);
  AttnArgs A = make_attn_args(xin, x_batch_stride, pe, f0, ring_frames, nslots, w, eps, ap, ap_stride, L, Lq);
  A.xparts_stride = xparts_stride;
  A.nvideos = B;
  hipLaunchKernelGGL(kern, PART ? dim3(((B + 7) / 8) * 8 * (LF_NH / 2)) : dim3(LF_NH / 2, B), dim3(LF_NT), A2_LDS, st, A);
  sf_prof_end(SF_K_MHA, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// x [B][L][256] -> ap: 8 head-partial buffers [B*Lq, 256]
int sf_attn_oproj_ex(const float* xin, const sf_tfm_layer& w, float eps, float* ap, long long ap_stride, int B, int L,
                     int Lq, hipStream_t st) {
  static_assert(A2_LDS <= 160 * 1024, "attention+out-proj kernel: LDS budget");
  return launch_attn<false>(xin, (long long)L * LF_D, nullptr, 0, 1, 1, w, eps, ap, ap_stride, B, L, Lq, st);
}

// the layer input is the sum of the previous layer's four FFN chunk partials xparts[c] ([B*L, 256] each, xparts_stride
// floats apart; written by sf_ffn_parts_ex)
int sf_attn_oproj_parts_ex(const float* xparts, long long xparts_stride, const sf_tfm_layer& w, float eps, float* ap,
                           long long ap_stride, int B, int L, int Lq, hipStream_t st) {
  return launch_attn<false, true>(xparts, (long long)L * LF_D, nullptr, 0, 1, 1, w, eps, ap, ap_stride, B, L, Lq, st, xparts_stride);
}

// layer 0 of a rollout step: x = ring[b][(f0 + r / nslots) % ring_frames][r % nslots] + pe[r]
int sf_attn_oproj_ring_ex(const float* ring, int ring_frames, int nslots, int f0, const float* pe, const sf_tfm_layer& w,
                          float eps, float* ap, long long ap_stride, int B, int L, int Lq, hipStream_t st) {
  return launch_attn<true>(ring, (long long)ring_frames * nslots * LF_D, pe, f0, ring_frames, nslots, w, eps, ap, ap_stride,
                           B, L, Lq, st);
}

// ---- all 8 heads in one workgroup per video: x2 [B*Lq][256] = x + out_proj(MHA(LN1(x))) + b_o (finished rows) ----
template <bool RING, bool PART>
static int launch_attn_all(const float* xin, long long x_batch_stride, const float* pe, int f0, int ring_frames, int nslots,
                           const sf_tfm_layer& w, float eps, float* x2, int B, int L, int Lq, hipStream_t st, long long xparts_stride = 0) {
  if (!w.attn_in_packed || !w.attn_out_packed)
    return sf_set_err(-1, "invalid argument: fused attention needs packed weights (sf_pack_attn_weights)", __FILE__, __LINE__);
  auto kern = attn_all_kernel<RING, PART>;
  SF_TRY(sf_ensure_dyn_lds((const void*)kern, A8_LDS));
  sf_prof_begin(SF_K_MHA, st, 6.0 * B * L * (double)LF_D * LF_D + 4.0 * (double)B * LF_NH * Lq * L * LF_HD +
                                  2.0 * B * Lq * (double)LF_D * LF_D);
  // x2 [2][B*Lq][256]: the finished rows, and behind them the running sum of the head pairs' partials (scratch)
  AttnArgs A = make_attn_args(xin, x_batch_stride, pe, f0, ring_frames, nslots, w, eps, x2, (long long)B * Lq * LF_D, L, Lq);
  A.xparts_stride = xparts_stride;
  A.nvideos = B;
  hipLaunchKernelGGL(kern, dim3(B), dim3(LF_NT), A8_LDS, st, A);
  sf_prof_end(SF_K_MHA, st);
  SF_CHECK_LAUNCH();
  return 0;
}
int sf_attn_all_ex(const float* xin, const sf_tfm_layer& w, float eps, float* x2, int B, int L, int Lq, hipStream_t st) {
  return launch_attn_all<false, false>(xin, (long long)L * LF_D, nullptr, 0, 1, 1, w, eps, x2, B, L, Lq, st);
}
int sf_attn_all_parts_ex(const float* xparts, long long xparts_stride, const sf_tfm_layer& w, float eps, float* x2, int B, int L,
                         int Lq, hipStream_t st) {
  return launch_attn_all<false, true>(xparts, (long long)L * LF_D, nullptr, 0, 1, 1, w, eps, x2, B, L, Lq, st, xparts_stride);
}
int sf_attn_all_ring_ex(const float* ring, int ring_frames, int nslots, int f0, const float* pe, const sf_tfm_layer& w, float eps,
                        float* x2, int B, int L, int Lq, hipStream_t st) {
  return launch_attn_all<true, false>(ring, (long long)ring_frames * nslots * LF_D, pe, f0, ring_frames, nslots, w, eps, x2, B, L, Lq, st);
}

static SbArgs make_sb(const void* wout_packed, const float* b_out, const void* win_packed, const float* b_in, float* slots,
                      long long slots_bs, long long slots_off, float* ring, long long ring_bs, long long ring_off,
                      int rows_per_video);

static FfnArgs make_ffn_args(const float* ap, long long ap_stride, const sf_tfm_layer& w, float eps, float* xp, long long xp_stride,
                             float* xout, int* counters, int M, const SbArgs& sb) {
  FfnArgs F;
  F.ap = ap; F.ap_stride = ap_stride; F.ln_g = w.norm2_g; F.ln_b = w.norm2_b; F.ln_eps = eps;
  F.w1p = (const uint4*)w.lin1_packed; F.b1 = w.lin1_b; F.w2p = (const uint4*)w.lin2_packed; F.b2 = w.lin2_b;
  F.xp = xp; F.xp_stride = xp_stride; F.xout = xout; F.counters = counters;
  F.ntiles = (M + FB_ROWS - 1) / FB_ROWS; F.M = M; F.dbg = lf_dbg(); F.sb = sb;
  F.parts_only = 0;
  F.np = LF_NP;
  return F;
}

// 8 tiles x 4 chunks per group of 32 consecutive blocks; left-over tiles behind whole groups: 4 blocks each (ffn_body)
static int ffn_blocks(int tiles) { return tiles < 8 ? 32 : (tiles / 8) * 32 + (tiles % 8) * 4; }

static int launch_ffn(const float* ap, long long ap_stride, const sf_tfm_layer& w, float eps, float* xp, long long xp_stride,
                      float* xout, int* counters, int M, int ffn, const SbArgs& sb, hipStream_t st, int parts_only = 0, int np = LF_NP) {
  static_assert(FB_LDS <= 160 * 1024, "FFN kernel: LDS budget");
  static_assert((size_t)4 * 16 * 64 * 4 + (size_t)2 * 32 * SB_PP * 2 <= (size_t)2 * FB_ROWS * FB_AP * 2, "boundary scratch fits the H planes");
  if (!w.lin1_packed || !w.lin2_packed || ffn != LF_NCH * LF_HC)
    return sf_set_err(-1, "invalid argument: fused FFN needs packed weights (sf_pack_ffn_weights) and ffn == 1024", __FILE__, __LINE__);
  SF_TRY(sf_ensure_dyn_lds((const void*)ffn_partial_kernel<LF_NP>, (size_t)(FB_LDS)));
  SF_TRY(sf_ensure_dyn_lds((const void*)ffn_partial_kernel<1>, (size_t)(FB_LDS)));
  const int tiles = (M + FB_ROWS - 1) / FB_ROWS;
  sf_prof_begin(SF_K_FFN, st, 4.0 * M * (double)LF_D * ffn);
  FfnArgs F = make_ffn_args(ap, ap_stride, w, eps, xp, xp_stride, xout, counters, M, sb);
  F.parts_only = parts_only;
  F.np = np;
  if (np == 1)
    hipLaunchKernelGGL(ffn_partial_kernel<1>, dim3(ffn_blocks(tiles)), dim3(LF_NT), FB_LDS, st, F);
  else
    hipLaunchKernelGGL(ffn_partial_kernel<LF_NP>, dim3(ffn_blocks(tiles)), dim3(LF_NT), FB_LDS, st, F);
  sf_prof_end(SF_K_FFN, st);
  SF_CHECK_LAUNCH();
  return 0;
}

int sf_ffn_partial_ex(const float* ap, long long ap_stride, const sf_tfm_layer& w, float eps, float* xp,
                      long long xp_stride, float* xout, int* counters, int M, int ffn, hipStream_t st, int np) {
  SbArgs sb;
  memset(&sb, 0, sizeof(sb));
  return launch_ffn(ap, ap_stride, w, eps, xp, xp_stride, xout, counters, M, ffn, sb, st, 0, np);
}

// the four chunk partials xp[c] ARE the output (summed by the next layer's sf_attn_oproj_parts_ex).  Rows per workgroup:
// the calling thread's option (sf_rollout_opts.ffn_rows), else 64 when sf_set_ffn_rows64(1), else 32; halved while the
// launch would have fewer rows than one workgroup covers.
template <int NH, int NP>
static int launch_ffn_wide(const FfnArgs& F, int ffn, hipStream_t st) {
  constexpr size_t LDS = (size_t)NH * F6_R1;
  static_assert(LDS <= 160 * 1024, "wide FFN kernel: LDS budget");
  SF_TRY(sf_ensure_dyn_lds((const void*)ffn_wide_parts_kernel<NH, NP>, LDS));
  if ((long long)LF_NP * F.ap_stride * 4 >= 0x7fffffffLL || (long long)LF_NCH * F.xp_stride * 4 >= 0x7fffffffLL)
    return sf_set_err(-1, "invalid argument: partial buffers beyond the 2 GB a buffer descriptor addresses", __FILE__, __LINE__);
  const int ntw = (F.M + 64 * NH - 1) / (64 * NH);
  sf_prof_begin(SF_K_FFN, st, 4.0 * F.M * (double)LF_D * ffn);
  hipLaunchKernelGGL((ffn_wide_parts_kernel<NH, NP>), dim3(ffn_blocks(ntw)), dim3(LF_NT), LDS, st, F);
  sf_prof_end(SF_K_FFN, st);
  SF_CHECK_LAUNCH();
  return 0;
}

int sf_ffn_parts_ex(const float* ap, long long ap_stride, const sf_tfm_layer& w, float eps, float* xp, long long xp_stride, int M,
                    int ffn, hipStream_t st, int np) {
  SbArgs sb;
  memset(&sb, 0, sizeof(sb));
  int rows = sf_thread_opts().ffn_rows > 0 ? sf_thread_opts().ffn_rows : (g_ffn_rows64 ? 64 : 32);
  while (rows > 32 && M < rows) rows >>= 1;
  if (rows <= 32) return launch_ffn(ap, ap_stride, w, eps, xp, xp_stride, nullptr, nullptr, M, ffn, sb, st, 1, np);
  if (!w.lin1_packed || !w.lin2_packed || ffn != LF_NCH * LF_HC)
    return sf_set_err(-1, "invalid argument: fused FFN needs packed weights (sf_pack_ffn_weights) and ffn == 1024", __FILE__, __LINE__);
  FfnArgs F = make_ffn_args(ap, ap_stride, w, eps, xp, xp_stride, nullptr, nullptr, M, sb);
  F.parts_only = 1;
  F.np = np;
  if (np == 1) return rows >= 128 ? launch_ffn_wide<2, 1>(F, ffn, st) : launch_ffn_wide<1, 1>(F, ffn, st);
  return rows >= 128 ? launch_ffn_wide<2, LF_NP>(F, ffn, st) : launch_ffn_wide<1, LF_NP>(F, ffn, st);
}

// Building block behind the layers 0 .. n-2 of a rollout step, exported for kernel-level tests: head-pair partials
// ap [4][M][256] (ap_stride floats apart) -> the four hidden-chunk partials xp [4][M][256] of
//   x2 = sum(ap);  y = x2 + lin2(relu(lin1(LN2(x2))))      (nn.TransformerEncoderLayer, norm_first, slotformer.py:72-80)
// whose sum ((p0 + p1) + p2) + p3 is y.  rows_per_wg: 32 / 64 / 128 (0 = the calling thread's default).
extern "C" int sf_ffn_chunk_partials_f32(const sf_tfm_layer* w, const float* ap, long long ap_stride, float* xp, long long xp_stride,
                                         int M, int ffn, int rows_per_wg, void* stream) {
  SF_REQUIRE(w && ap && xp && M > 0, "sf_ffn_chunk_partials_f32: null pointer / empty problem");
  SF_REQUIRE(rows_per_wg == 0 || rows_per_wg == 32 || rows_per_wg == 64 || rows_per_wg == 128, "sf_ffn_chunk_partials_f32: rows_per_wg must be 0, 32, 64 or 128");
  SF_REQUIRE(w->norm2_g && w->norm2_b && w->lin1_b && w->lin2_b && w->lin1_packed && w->lin2_packed, "sf_ffn_chunk_partials_f32: null weight (packed FFN weights needed)");
  SfThreadOpts saved = sf_thread_opts();
  if (rows_per_wg) sf_thread_opts().ffn_rows = rows_per_wg;
  const int rc = sf_ffn_parts_ex(ap, ap_stride, *w, 1e-5f, xp, xp_stride, M, ffn, (hipStream_t)stream, LF_NP);
  sf_thread_opts() = saved;
  return rc;
}

// Attention block of a pre-LN layer, exported for kernel-level tests: x [B][L][256] -> the last Lq rows of every video of
//   x2 = x + out_proj(MHA(LN1(x))) + b_o      (nn.TransformerEncoderLayer._sa_block + residual, norm_first; slotformer.py:72-80)
// heads_per_wg = 8: `out` [2][B*Lq][256]: out[0] receives x2 (one workgroup per video, all heads), out[1] is scratch;
// heads_per_wg = 2: `out` [4][B*Lq][256] receives the four head-pair partials whose sum ((p0 + p1) + p2) + p3 is x2.
extern "C" int sf_attn_block_f32(const sf_tfm_layer* w, const float* x, float* out, int B, int L, int Lq, int heads_per_wg,
                                 void* stream) {
  SF_REQUIRE(w && x && out && B > 0, "sf_attn_block_f32: null pointer / empty problem");
  SF_REQUIRE(L >= 1 && L <= FA_ROWS && Lq >= 1 && Lq <= L, "sf_attn_block_f32: needs 1 <= Lq <= L <= 64");
  SF_REQUIRE(heads_per_wg == 2 || heads_per_wg == 8, "sf_attn_block_f32: heads_per_wg must be 2 or 8");
  SF_REQUIRE(w->norm1_g && w->norm1_b && w->in_proj_b && w->out_proj_b && w->attn_in_packed && w->attn_out_packed,
             "sf_attn_block_f32: null weight (packed attention weights needed)");
  if (heads_per_wg == 8) return sf_attn_all_ex(x, *w, 1e-5f, out, B, L, Lq, (hipStream_t)stream);
  return sf_attn_oproj_ex(x, *w, 1e-5f, out, (long long)B * Lq * LF_D, B, L, Lq, (hipStream_t)stream);
}

// last layer of a rollout step: the FFN's last-arriving workgroups also run the step boundary (out-proj -> slots frame
// `frame`, in-proj -> ring); M must be B * nslots
int sf_ffn_boundary_ex(const float* ap, long long ap_stride, const sf_tfm_layer& w, float eps, float* xp, long long xp_stride,
                       int* counters, int ffn, const void* wout_packed, const float* b_out, const void* win_packed,
                       const float* b_in, float* slots, long long slots_bs, int frame, float* ring, int ring_frames, int nslots,
                       int B, hipStream_t st, int np) {
  const SbArgs sb = make_sb(wout_packed, b_out, win_packed, b_in, slots, slots_bs, (long long)frame * nslots * SB_C, ring,
                            (long long)ring_frames * nslots * LF_D, (long long)(frame % ring_frames) * nslots * LF_D, nslots);
  return launch_ffn(ap, ap_stride, w, eps, xp, xp_stride, nullptr, counters, B * nslots, ffn, sb, st, 0, np);
}

// ONE launch: last-layer FFN + step boundary of step s (writes slots frame `frame`, ring frame `frame`) and the layer-0
// attention of step s+1 (window starting at ring frame f0_next).  seam_flags: >= 1024 words zeroed at the start of the
// rollout; epoch = s + 1.  ap_ffn: the last layer's head-pair partials (input); ap_attn: where the attention writes its own.
int sf_seam_ex(const float* ap_ffn, long long pst_ffn, const sf_tfm_layer& wl, float eps, float* xp, long long xp_stride,
               int* counters, int ffn, const void* wout_packed, const float* b_out, const void* win_packed, const float* b_in,
               float* slots, long long slots_bs, int frame, float* ring, int ring_frames, int nslots, int B,
               const sf_tfm_layer& w0, int f0_next, const float* pe, float* ap_attn, long long pst_attn, int L, int Lq,
               unsigned* seam_flags, unsigned epoch, hipStream_t st) {
  if (!wl.lin1_packed || !wl.lin2_packed || ffn != LF_NCH * LF_HC || !w0.attn_in_packed || !w0.attn_out_packed)
    return sf_set_err(-1, "invalid argument: the seam launch needs packed weights", __FILE__, __LINE__);
  constexpr size_t LDS = A2_LDS > FB_LDS ? A2_LDS : FB_LDS;
  SF_TRY(sf_ensure_dyn_lds((const void*)seam_kernel, LDS));
  SbArgs sb = make_sb(wout_packed, b_out, win_packed, b_in, slots, slots_bs, (long long)frame * nslots * SB_C, ring,
                      (long long)ring_frames * nslots * LF_D, (long long)(frame % ring_frames) * nslots * LF_D, nslots);
  sb.seam_flags = seam_flags;
  sb.seam_epoch = epoch;
  const int M = B * nslots, tiles = (M + FB_ROWS - 1) / FB_ROWS, nffn = ffn_blocks(tiles);
  const FfnArgs F = make_ffn_args(ap_ffn, pst_ffn, wl, eps, xp, xp_stride, nullptr, counters, M, sb);
  const AttnArgs A = make_attn_args(ring, (long long)ring_frames * nslots * LF_D, pe, f0_next, ring_frames, nslots, w0, eps, ap_attn,
                                    pst_attn, L, Lq);
  const SeamArgs seam{seam_flags, epoch, FB_ROWS};
  // algorithmic work: the last layer's FFN on M rows + the next step's layer-0 attention
  sf_prof_begin(SF_K_SEAM, st, 4.0 * M * (double)LF_D * ffn + 6.0 * B * L * (double)LF_D * LF_D + 4.0 * (double)B * LF_NH * Lq * L * LF_HD +
                                   2.0 * B * Lq * (double)LF_D * LF_D);
  hipLaunchKernelGGL(seam_kernel, dim3(nffn + (LF_NH / 2) * B), dim3(LF_NT), LDS, st, F, A, seam, nffn);
  sf_prof_end(SF_K_SEAM, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// the seam launch needs the next window's newest frame entirely inside token block 1 (rows >= 32)
bool sf_seam_window_ok(int L, int nslots) { return L > 32 && L <= FA_ROWS && L - nslots >= 32; }

// workgroups of a seam launch (producers + consumers): they must all be co-resident, one per CU
int sf_seam_blocks(int B, int nslots) {
  const int tiles = (B * nslots + FB_ROWS - 1) / FB_ROWS;
  return ffn_blocks(tiles) + (LF_NH / 2) * B;
}

bool sf_step_boundary_ok(int d, int slot_size) { return d == LF_D && slot_size == SB_C; }

static SbArgs make_sb(const void* wout_packed, const float* b_out, const void* win_packed, const float* b_in, float* slots,
                      long long slots_bs, long long slots_off, float* ring, long long ring_bs, long long ring_off,
                      int rows_per_video) {
  SbArgs a;
  a.wout_p = (const uint4*)wout_packed;
  a.b_out = b_out;
  a.win_p = (const uint4*)win_packed;
  a.b_in = b_in;
  a.slots = slots;
  a.slots_bs = slots_bs;
  a.slots_off = slots_off;
  a.ring = ring;
  a.ring_bs = ring_bs;
  a.ring_off = ring_off;
  a.nslots = rows_per_video;
  a.enabled = 1;
  a.seam_flags = nullptr;
  a.seam_epoch = 0;
  return a;
}

static int launch_boundary(const float* y, const SbArgs& sb, int M, int proj_only, hipStream_t st) {
  SF_TRY(sf_ensure_dyn_lds((const void*)step_boundary_kernel, (size_t)(SB_LDS)));
  sf_prof_begin(SF_K_LINEAR, st, (proj_only ? 2.0 : 4.0) * M * (double)LF_D * SB_C);
  hipLaunchKernelGGL(step_boundary_kernel, dim3((M + 31) / 32), dim3(LF_NT), SB_LDS, st, y, sb, M, proj_only);
  sf_prof_end(SF_K_LINEAR, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// frame index `frame` of the [B][T][N][128] slots buffer (slots_bs floats per video) and of the projection ring
int sf_step_boundary_ex(const float* y, const void* wout_packed, const float* b_out, const void* win_packed,
                        const float* b_in, float* slots, long long slots_bs, int frame, float* ring, int ring_frames,
                        int nslots, int B, hipStream_t st) {
  const SbArgs sb = make_sb(wout_packed, b_out, win_packed, b_in, slots, slots_bs, (long long)frame * nslots * SB_C, ring,
                            (long long)ring_frames * nslots * LF_D, (long long)(frame % ring_frames) * nslots * LF_D, nslots);
  return launch_boundary(y, sb, B * nslots, 0, st);
}

// in-projection of the first n_frames (<= ring_frames) frames of every video -> ring slots 0 .. n_frames-1
int sf_ring_init_ex(const void* wout_packed, const float* b_out, const void* win_packed, const float* b_in, float* slots,
                    long long slots_bs, int n_frames, float* ring, int ring_frames, int nslots, int B, hipStream_t st) {
  const SbArgs sb = make_sb(wout_packed, b_out, win_packed, b_in, slots, slots_bs, 0, ring,
                            (long long)ring_frames * nslots * LF_D, 0, n_frames * nslots);
  return launch_boundary(nullptr, sb, B * n_frames * nslots, 1, st);
}

extern "C" size_t sf_packed_linear_bytes(int N, int K) { return (size_t)N * K * 4; }

// W [N][K] (torch nn.Linear layout) -> fragment-ordered split-bf16 copy (layout above pack_linear_kernel).
extern "C" int sf_pack_linear_weights(const float* w, void* packed, int N, int K, void* stream) {
  SF_REQUIRE(w && packed, "sf_pack_linear_weights: null pointer");
  SF_REQUIRE(N > 0 && K > 0 && (N % 32) == 0 && (K % 16) == 0, "sf_pack_linear_weights: needs N % 32 == 0 and K % 16 == 0");
  const int total = (K / 16) * (N / 32) * 2 * 64;
  hipLaunchKernelGGL(pack_linear_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (uint4*)packed, N, K);
  SF_CHECK_LAUNCH();
  return 0;
}

// Number of seam hand-offs (rollout steps fused across the step boundary) that timed out waiting for their producer since
// the library was loaded.  Anything but 0 means a rollout result is invalid (a producer workgroup was not resident).
extern "C" int sf_seam_timeouts(void) {
  unsigned v = 0;
  hipError_t e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(lf_seam_timeouts), sizeof(v));
  return e == hipSuccess ? (int)v : -(int)e;
}

extern "C" int sf_debug_read_wg(unsigned long long* out256) {
  hipError_t e = hipMemcpyFromSymbol(out256, HIP_SYMBOL(lf_wg), sizeof(unsigned long long) * 256);
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int sf_debug_read_ts(long long* out64) {
  hipError_t e = hipMemcpyFromSymbol(out64, HIP_SYMBOL(lf_ts), sizeof(long long) * 64);
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" size_t sf_attn_packed_bytes(int d_model, int which) { return (size_t)d_model * d_model * (which == 0 ? 3 : 1) * 4; }

// in_proj_w [3d, d] and out_proj_w [d, d] of nn.MultiheadAttention (d = 256, 8 heads) -> fragment-ordered split-bf16 copies
// for attn_oproj_kernel (layouts above it); outputs need sf_attn_packed_bytes(d, 0) and (d, 1) bytes.
extern "C" int sf_pack_attn_weights(const float* in_proj_w, const float* out_proj_w, void* in_packed, void* out_packed,
                                    int d_model, int num_heads, void* stream) {
  SF_REQUIRE(in_proj_w && out_proj_w && in_packed && out_packed, "sf_pack_attn_weights: null pointer");
  SF_REQUIRE(d_model == LF_D && num_heads == LF_NH, "sf_pack_attn_weights: needs d_model == 256 and 8 heads");
  const int total = 4 * 6 * 16 * 2 * 64;
  hipLaunchKernelGGL(pack_attn_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, in_proj_w, out_proj_w,
                     (uint4*)in_packed, (uint4*)out_packed);
  SF_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t sf_ffn_packed_bytes(int d_model, int ffn) { return (size_t)d_model * ffn * 4; }

// lin1_w [ffn, d], lin2_w [d, ffn] (torch layouts) -> fragment-ordered split-bf16 copies (layout above).
extern "C" int sf_pack_ffn_weights(const float* lin1_w, const float* lin2_w, void* lin1_packed, void* lin2_packed,
                                   int d_model, int ffn, void* stream) {
  SF_REQUIRE(lin1_w && lin2_w && lin1_packed && lin2_packed, "sf_pack_ffn_weights: null pointer");
  SF_REQUIRE(d_model == LF_D && ffn > 0 && (ffn % LF_HC) == 0, "sf_pack_ffn_weights: needs d_model == 256 and ffn % 256 == 0");
  const int total = (ffn / LF_HC) * 16 * 8 * 2 * 64;
  hipLaunchKernelGGL(pack_ffn_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, lin1_w, lin2_w,
                     (uint4*)lin1_packed, (uint4*)lin2_packed, ffn);
  SF_CHECK_LAUNCH();
  return 0;
}


int sf_ffn_tiles(int M) { return (M + FB_ROWS - 1) / FB_ROWS; }
