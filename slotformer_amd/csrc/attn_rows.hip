// Attention block of a rollout layer in its ROW-TILE form (per-call option attn_qkv_rows = 128): two launches.
//
//   qkv_rows_kernel   one workgroup per (128-row tile, q | k | v): LN1 of 128 consecutive token rows of the batch -- rows of
//                     three videos at L = 42, no padding to 64 per video -- times the 256 q (or k, or v) columns of all eight
//                     heads; wave w owns head w's 32 columns with its weight fragments register-resident for all of K.  The
//                     results leave as split-bf16 planes in the order the attention core's MFMA fragments are read:
//                         planes [video][head][Qh, Ql, Kh, Kl, Vh, Vl][2048 bf16]
//                         q, k: [64 tokens][32];  v^T: [32][64 keys], keys in the order the score accumulators hold them
//   attn_core_kernel  one workgroup per video, wave = head: q, k, v^T fragments straight from memory into registers (no LDS, no
//                     barrier between the heads), scores^T -> in-register softmax -> PV per (query block, key block), the two
//                     key blocks merged in registers; O planes of all heads in LDS; out-projection (wave = 32 output columns)
//                     with the residual and the bias: finished rows x2 [B*Lq][256].
//
// Why: the all-heads workgroup of layer_fused.hip (attn_all_kernel) pads every video to 64 token rows (42 used: a third of its
// projection MFMAs multiply zeros), ingests all 1040 KB of attention weights per video through one CU and walks the four head
// pairs one after the other behind workgroup barriers (40-46 us on its CU).  Here the projection runs on packed rows with one
// weight load per 128 rows, and the eight heads of a video run side by side.
//
// Every product, the order of the k-steps, the split-K of the k / v blocks of odd heads, the softmax merge and the order in
// which the head pairs' out-projection partials are summed are attn_body's / attn_all_kernel's: the three forms of the
// attention block give the same bits (tests/test_rollout_opts_gpu.py).
// Reference: nn.TransformerEncoderLayer(norm_first=True)._sa_block + residual as configured at slotformer.py:72-80.
#include "../../include/slotformer_hip.h"
#include "sf_internal.h"
#include "layer_fused.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int NT = 512;
constexpr int D = 256, HD = 32, NH = 8;
constexpr int AP = D + 8;                       // bf16 pitch of the LN(x) / O planes (528 B = 33 slots)
constexpr int TRMAX = 128;                      // rows per projection tile: 128 or 64 (template parameter TR)
constexpr int KC = 64, NK = D / KC;             // width of an activation load chunk
constexpr int PL = 2048;                        // bf16 elements of one plane
constexpr size_t K1_PLANES = (size_t)2 * TRMAX * AP * 2;          // hi | lo planes of a 128-row tile: 135,168 B
constexpr size_t K1_LDS = K1_PLANES + 2 * D * 4;                  // + gamma | beta
constexpr size_t K2_LDS = (size_t)2 * 64 * AP * 2;                // O planes [hi, lo][64][AP]
static_assert(K1_LDS <= 160 * 1024 && K2_LDS <= 160 * 1024, "LDS budget");

__device__ __forceinline__ void split4(__bf16* hp, __bf16* lp, int off, f32x4 v) {
  const bf16x4 hi = __builtin_convertvector(v, bf16x4);
  const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
  *(bf16x4*)(hp + off) = hi;
  *(bf16x4*)(lp + off) = lo;
}

// position of key t inside its video's v^T rows: the score accumulators of a 32x32 MFMA hold keys (r & 3) + 8 (r >> 2) + 4 (lane >> 5),
// and eight consecutive registers are one B fragment of the PV product -- bits 2 and 3 of the key index trade places
__device__ __forceinline__ int vpos(int t) { return (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1); }

struct RowsArgs {
  const float* xin;
  long long x_batch_stride, xparts_stride;
  const float* pe;
  int f0, ring_frames, nslots;
  const float *ln_g, *ln_b;
  float ln_eps;
  const uint4* wqkv_p;
  const float* bias;
  __bf16* planes;
  float* x2;
  int B, L, Lq, nt, ng, main_blocks;
  int dbg;
};

// m / L for 0 <= m < 2^22 (one float multiply and two corrections instead of a 32-bit integer division)
__device__ __forceinline__ int div_rows(int m, int L, float invL) {
  int q = (int)((float)m * invL);
  q -= (q * L > m);
  q += ((q + 1) * L <= m);
  return q;
}
}  // namespace

__device__ long long ar_ts[32];   // phase timestamps of workgroup 0 (SF_DBG=lf=16)
#define RTS(i) do { if ((A.dbg & 16) && blockIdx.x == 0 && threadIdx.x == 0) ar_ts[i] = wall_clock64(); } while (0)

// RING: the rows are the cached in-projections of the window's frames (ring of `ring_frames` frames per video) + the position table;
// PART: the rows are the sum of the previous layer's four FFN chunk partials, ((p0 + p1) + p2) + p3.
// One workgroup per 128-row tile runs the groups G0 .. G0 + NG - 1 (0 q, 1 k, 2 v) one after the other on ONE ingest and ONE
// LayerNorm of its rows: wave w owns head w's 32 columns of every group and streams their weight fragments (2 KB per k-step)
// through a ring of four two-k-step slots, three chunks in flight across group boundaries; each fragment feeds the four row
// blocks of the tile (12 MFMAs).  QROWS: the tile's rows are the last Lq rows of every video (the q tiles of the last layer).
template <bool RING, bool PART, int G0, int NG, bool QROWS, int TR>
__device__ __forceinline__ void qkv_rows_body(const RowsArgs& A, const int tile) {
#pragma clang fp contract(off)
  constexpr int NRB = TR / 32;   // 32-row blocks of the tile
  constexpr int NP = PART ? 4 : 1;
  RTS(0);
  const int L = A.L, Lq = A.Lq;
  const int Lg = QROWS ? Lq : L, tg = QROWS ? L - Lq : 0, Mg = A.B * Lg, row0 = tile * TR;
  const float invL = 1.0f / (float)Lg;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* Ah = (__bf16*)smem;             // [128][AP]
  __bf16* Al = Ah + TR * AP;
  float* GB = (float*)((char*)smem + (size_t)2 * TR * AP * 2);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int c4 = t & 15, r0 = t >> 4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 gbv = zero4;
  if (t < 128) gbv = *(const f32x4*)((t < 64 ? A.ln_g : A.ln_b) + 4 * (t & 63));
  // ---- rows: thread (r0, c4) owns float4 column c4 of every 64-wide chunk of rows r0 + 32 p; two register buffers, two passes
  //      in flight ----
  f32x4 pr[2][NP][NK], pev[2][NK];
  auto request = [&](int p, int s) {
    const int mc = min(row0 + 32 * p + r0, Mg - 1);
    const int b = div_rows(mc, Lg, invL), tk = tg + mc - b * Lg;
    const float* row;
    if constexpr (RING) {
      const int fr = tk / A.nslots, sl = tk - fr * A.nslots;
      row = A.xin + (long long)b * A.x_batch_stride + ((long long)((A.f0 + fr) % A.ring_frames) * A.nslots + sl) * D + 4 * c4;
      const float* pp = A.pe + (long long)tk * D + 4 * c4;
#pragma unroll
      for (int kc = 0; kc < NK; ++kc) pev[s][kc] = *(const f32x4*)(pp + kc * KC);
    } else {
      row = A.xin + (long long)b * A.x_batch_stride + (long long)tk * D + 4 * c4;
    }
#pragma unroll
    for (int kc = 0; kc < NK; ++kc)
#pragma unroll
      for (int q = 0; q < NP; ++q) pr[s][q][kc] = *(const f32x4*)(row + (long long)q * A.xparts_stride + kc * KC);
  };
  auto sum = [&](int s, f32x4 (&vv)[NK]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int kc = 0; kc < NK; ++kc) {
      if constexpr (PART)
        vv[kc] = ((pr[s][0][kc] + pr[s][1][kc]) + pr[s][2][kc]) + pr[s][3][kc];
      else
        vv[kc] = pr[s][0][kc];
      if constexpr (RING) vv[kc] += pev[s][kc];
    }
  };
  auto ln = [&](int p, const f32x4 (&vv)[NK]) {
#pragma clang fp contract(off)
    const int m = row0 + 32 * p + r0;
    const bool ok = m < Mg;
    // the residual of the query rows is parked in the output rows (picked up by attn_core_kernel)
    if (G0 == 0 && ok) {   // (tiles that run q hold query rows only: row m of the tile's row space is row m of x2)
#pragma unroll
      for (int kc = 0; kc < NK; ++kc) *(f32x4*)(A.x2 + (long long)m * D + kc * KC + 4 * c4) = vv[kc];
    }
    float sm = 0.f;
#pragma unroll
    for (int kc = 0; kc < NK; ++kc) sm += (vv[kc][0] + vv[kc][1]) + (vv[kc][2] + vv[kc][3]);
    const float mu = sf_sum16(sm) / (float)D;
    float vs = 0.f;
#pragma unroll
    for (int kc = 0; kc < NK; ++kc) {
      const f32x4 dv = vv[kc] - mu;
      vs += (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]);
    }
    const float rs = 1.0f / sqrtf(sf_sum16(vs) / (float)D + A.ln_eps);
    const int r = 32 * p + r0;
#pragma unroll
    for (int kc = 0; kc < NK; ++kc) {
      const int k = kc * KC + 4 * c4;
      const f32x4 gm = *(const f32x4*)(GB + k), be = *(const f32x4*)(GB + D + k);
      split4(Ah, Al, r * AP + k, ok ? (vv[kc] - mu) * rs * gm + be : zero4);
    }
  };
  // ---- weight ring: slot = chunk & 3, chunk gc = group index * 8 + (k-step / 2); buffer loads with scalar fragment offsets ----
  bf16x8 ring[4][2][2];
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(A.wqkv_p), 0, 0x7fffffff, 0x00020000);
  const unsigned hpb = (unsigned)(((wave >> 1) * 6 + 3 * (wave & 1)) * 16 * 2048);   // + g * 16 * 2048
  auto load_chunk = [&](int gc) {
    const int g = G0 + gc / 8, ks0 = (gc % 8) * 2;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        ring[gc & 3][k][pl] = __builtin_bit_cast(
            bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, (unsigned)(lane * 16), hpb + (unsigned)((g * 16 + ks0 + k) * 2048 + pl * 1024), 0));
  };
  const float scale = 1.0f / sqrtf((float)HD);
  const int kg = lane >> 5;
  const int ao = (lane & 31) * AP + 8 * (lane >> 5);

  // ---- ingest + LayerNorm ----
  request(0, 0);
  request(1, 1);
  load_chunk(0);
  load_chunk(1);
  load_chunk(2);
  if (t < 128) *(f32x4*)(GB + 4 * t) = gbv;
  __syncthreads();   // gamma | beta
  RTS(1);
  {
    f32x4 vv[NK];
    sum(0, vv);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NRB > 2) request(2, 0);
    ln(0, vv);
    sum(1, vv);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NRB > 2) request(3, 1);
    ln(1, vv);
    if constexpr (NRB > 2) {
      sum(0, vv);
      ln(2, vv);
      sum(1, vv);
      ln(3, vv);
    }
  }
  __syncthreads();   // LN planes of the tile
  RTS(2);

  // ---- the groups ----
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int g = G0 + gi;
    const bool vblock = g == 2;
    // the k / v blocks of odd heads are split over K in attn_body (two waves, lower + upper half): same sum here
    const bool splitk = (wave & 1) && g >= 1;
    const float* bq = A.bias + g * D + wave * HD;
    f32x4 bv4[4];   // bias of this lane's output columns 8 gq + 4 kg .. + 3 of the head
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) bv4[gq] = *(const f32x4*)(bq + 8 * gq + 4 * kg);
    f32x16 acc[NRB], sav[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int gc = gi * 8 + c;
      if (gc + 3 < NG * 8) load_chunk(gc + 3);
      if (c == 4 && splitk) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
          sav[rb] = acc[rb];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
        }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int ks = 2 * c + k;
        const bf16x8 w0 = ring[gc & 3][k][0], w1 = ring[gc & 3][k][1];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
          const bf16x8 xh = *(const bf16x8*)(Ah + ao + rb * 32 * AP + ks * 16), xl = *(const bf16x8*)(Al + ao + rb * 32 * AP + ks * 16);
          acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xl, acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, xh, acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xh, acc[rb], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // the requests stay one ring slot per chunk
    }
    if (splitk) {
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) acc[rb] = sav[rb] + acc[rb];
    }
    if (gi == 0) RTS(3);
    // ---- epilogue of the group: bias (+ scale on q), split, fragment planes ----
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      if (!vblock) {
        const int m = row0 + rb * 32 + (lane & 31);
        if (m < Mg) {
          const int b = div_rows(m, Lg, invL), tk = tg + m - b * Lg;
          __bf16* ph = A.planes + ((long long)(b * NH + wave) * 6 + 2 * g) * PL + tk * HD + 4 * kg;
          const float mul = g == 0 ? scale : 1.f;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const f32x4 bv = bv4[gq];
            split4(ph, ph + PL, 8 * gq,
                   f32x4{(acc[rb][4 * gq] + bv[0]) * mul, (acc[rb][4 * gq + 1] + bv[1]) * mul, (acc[rb][4 * gq + 2] + bv[2]) * mul,
                         (acc[rb][4 * gq + 3] + bv[3]) * mul});
          }
        }
      } else {
        // v^T planes [dim][key position]: this lane holds sixteen dims of ONE token; the 32 lanes of a half wave write 32
        // consecutive tokens -- neighbouring 2-byte words of the same dim row (a wave-wide store touches a few cache lines;
        // with the accumulators transposed -- lane = dim, attn_body's way into ITS LDS planes -- it touched 64)
        const int m = row0 + rb * 32 + (lane & 31);
        if (m < Mg) {
          const int b = div_rows(m, L, invL), tk = m - b * L;
          __bf16* pv = A.planes + ((long long)(b * NH + wave) * 6 + 4) * PL + vpos(tk);
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const f32x4 bv = bv4[gq];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float v = acc[rb][4 * gq + q] + bv[q];
              const __bf16 h = (__bf16)v;
              const int dim = 8 * gq + 4 * kg + q;
              pv[dim * 64] = h;
              pv[PL + dim * 64] = (__bf16)(v - (float)h);
            }
          }
        }
      }
    }
  }
  RTS(5);
}

template <bool RING, bool PART, int TR>
__global__ __launch_bounds__(NT) void qkv_rows_kernel(RowsArgs A) {
  const int blk = blockIdx.x;
  if (blk >= A.nt) {   // last layer: q over the newest frame's rows only
    qkv_rows_body<RING, PART, 0, 1, true, TR>(A, blk - A.nt);
  } else if (A.ng == 3) {
    qkv_rows_body<RING, PART, 0, 3, false, TR>(A, blk);
  } else {
    qkv_rows_body<RING, PART, 1, 2, false, TR>(A, blk);
  }
}

// ================================================================================================
namespace {
struct CoreArgs {
  const __bf16* planes;
  const uint4* wo_p;
  const float* bo;
  float* x2;
  int L, Lq, dbg;
};
}  // namespace

__global__ __launch_bounds__(NT) void attn_core_kernel(CoreArgs A) {
#pragma clang fp contract(off)
  const int b = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int L = A.L, Lq = A.Lq;
  const int nrb = L > 32 ? 2 : 1, nq0 = L - Lq;
  const bool skip_q0 = nrb == 2 && nq0 >= 32;   // no query rows in token block 0 (last layer)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* Oh = (__bf16*)smem;   // [64][AP]
  __bf16* Ol = Oh + 64 * AP;
  if ((A.dbg & 16) && b == 0 && t == 0) ar_ts[8] = wall_clock64();
  // ---- fragments of head `wave`, straight from the planes ----
  const __bf16* P = A.planes + (long long)(b * NH + wave) * 6 * PL;
  const int oqk = (lane & 31) * HD + 8 * (lane >> 5), ov = (lane & 31) * 64 + 8 * (lane >> 5);
  bf16x8 qf[2][2][2], kf[2][2][2], vf[2][2][2];   // [token block][k-step][hi, lo]
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    if (blk >= nrb) continue;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        kf[blk][ks][pl] = *(const bf16x8*)(P + (2 + pl) * PL + blk * 32 * HD + oqk + 16 * ks);
        vf[blk][ks][pl] = *(const bf16x8*)(P + (4 + pl) * PL + ov + blk * 32 + 16 * ks);
      }
  }
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    if (blk >= nrb || (blk == 0 && skip_q0)) continue;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) qf[blk][ks][pl] = *(const bf16x8*)(P + pl * PL + blk * 32 * HD + oqk + 16 * ks);
  }
  // out-projection fragments of output columns 32 wave .. 32 wave + 31: head pairs 0, 1 now, 2 and 3 behind the attention core
  bf16x8 wof[4][4][2];
  auto load_wo = [&](int hp) {
    const uint4* wp = A.wo_p + (((long long)(hp * 4) * 8 + wave) * 2) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      wof[hp][ks][0] = __builtin_bit_cast(bf16x8, wp[((ks * 8) * 2) * 64]);
      wof[hp][ks][1] = __builtin_bit_cast(bf16x8, wp[((ks * 8) * 2 + 1) * 64]);
    }
  };
  load_wo(0);
  load_wo(1);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // ---- attention core of this head: per query block the two key blocks, merged in registers (attn_body's arithmetic) ----
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int row = qb * 32 + (lane & 31);
    if (qb >= nrb || (qb == 0 && skip_q0)) {
      // rows of this block are never read by the out-projection (block skipped there) -- nothing to write
      continue;
    }
    f32x16 oacc[2];
    float mxs[2], sms[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[kb][r] = 0.f;
      mxs[kb] = -INFINITY;
      sms[kb] = 0.f;
      if (kb >= nrb) continue;
      f32x16 sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][ks][0], qf[qb][ks][1], sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][ks][1], qf[qb][ks][0], sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][ks][0], qf[qb][ks][0], sacc, 0, 0, 0);
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
      mxs[kb] = mx;
      sms[kb] = sum;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const f32x4 p0 = {sacc[8 * ks], sacc[8 * ks + 1], sacc[8 * ks + 2], sacc[8 * ks + 3]};
        const f32x4 p1 = {sacc[8 * ks + 4], sacc[8 * ks + 5], sacc[8 * ks + 6], sacc[8 * ks + 7]};
        const bf16x4 h0 = __builtin_convertvector(p0, bf16x4), h1 = __builtin_convertvector(p1, bf16x4);
        const bf16x4 l0 = __builtin_convertvector(p0 - __builtin_convertvector(h0, f32x4), bf16x4);
        const bf16x4 l1 = __builtin_convertvector(p1 - __builtin_convertvector(h1, f32x4), bf16x4);
        const bf16x8 ph = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 pl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        oacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kb][ks][0], pl, oacc[kb], 0, 0, 0);
        oacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kb][ks][1], ph, oacc[kb], 0, 0, 0);
        oacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kb][ks][0], ph, oacc[kb], 0, 0, 0);
      }
    }
    float f0, f1 = 0.f;
    if (nrb == 2) {
      const float mg = fmaxf(mxs[0], mxs[1]);
      const float e0 = (mxs[0] == -INFINITY) ? 0.f : expf(mxs[0] - mg), e1 = (mxs[1] == -INFINITY) ? 0.f : expf(mxs[1] - mg);
      f0 = e0 / (sms[0] * e0 + sms[1] * e1);
      f1 = e1 / (sms[1] * e1 + sms[0] * e0);
    } else {
      f0 = 1.0f / sms[0];
    }
    const bool inq = row >= nq0 && row < L;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v = f32x4{oacc[0][4 * g] * f0, oacc[0][4 * g + 1] * f0, oacc[0][4 * g + 2] * f0, oacc[0][4 * g + 3] * f0};
      if (nrb == 2) v += f32x4{oacc[1][4 * g] * f1, oacc[1][4 * g + 1] * f1, oacc[1][4 * g + 2] * f1, oacc[1][4 * g + 3] * f1};
      split4(Oh, Ol, row * AP + wave * HD + 8 * g + 4 * (lane >> 5), inq ? v : zero4);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  load_wo(2);
  load_wo(3);
  if ((A.dbg & 16) && b == 0 && t == 0) ar_ts[9] = wall_clock64();
  __syncthreads();   // the O planes of all heads
  // ---- out-projection: the four head pairs' partials, each from a fresh accumulator, summed ((p0 + p1) + p2) + p3 with the
  //      residual and the bias on the partial of the pair that owns these columns (attn_body / attn_all_kernel) ----
  // Head pairs 0 and 1 of both row blocks first -- their fragments are resident -- while the fragments of pairs 2 and 3,
  // requested behind the core, are still on their way.
  f32x4 s[2][4], xr[2][4], bv[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bv[g] = *(const f32x4*)(A.bo + wave * 32 + 8 * g + 4 * (lane >> 5));
#pragma unroll
  for (int rbk = 0; rbk < 2; ++rbk) {
    if (rbk >= nrb || rbk * 32 + 32 <= nq0) continue;   // no query rows in this block (uniform)
    const int row = rbk * 32 + (lane & 31);
    const float* o = A.x2 + ((long long)b * Lq + (min(max(row, nq0), L - 1) - nq0)) * D + wave * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int g = 0; g < 4; ++g) xr[rbk][g] = *(const f32x4*)(o + 8 * g);
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int rbk = 0; rbk < 2; ++rbk) {
      if (rbk >= nrb || rbk * 32 + 32 <= nq0) continue;
      const int row = rbk * 32 + (lane & 31);
      const int ao = row * AP + 8 * (lane >> 5);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int hp = 2 * half + hh;
        f32x16 pacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 xh = *(const bf16x8*)(Oh + ao + hp * 64 + ks * 16), xl = *(const bf16x8*)(Ol + ao + hp * 64 + ks * 16);
          pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wof[hp][ks][0], xl, pacc, 0, 0, 0);
          pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wof[hp][ks][1], xh, pacc, 0, 0, 0);
          pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wof[hp][ks][0], xh, pacc, 0, 0, 0);
        }
        const bool mine = (wave >> 1) == hp;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 p = {pacc[4 * g], pacc[4 * g + 1], pacc[4 * g + 2], pacc[4 * g + 3]};
          const f32x4 pm = p + (xr[rbk][g] + bv[g]);
          f32x4 v;
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = mine ? pm[q] : p[q];
          s[rbk][g] = hp == 0 ? v : s[rbk][g] + v;
        }
      }
    }
  }
#pragma unroll
  for (int rbk = 0; rbk < 2; ++rbk) {
    if (rbk >= nrb || rbk * 32 + 32 <= nq0) continue;
    const int row = rbk * 32 + (lane & 31);
    if (row >= nq0 && row < L) {
      float* o = A.x2 + ((long long)b * Lq + (row - nq0)) * D + wave * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int g = 0; g < 4; ++g) *(f32x4*)(o + 8 * g) = s[rbk][g];
    }
  }
  if ((A.dbg & 16) && b == 0 && t == 0) ar_ts[10] = wall_clock64();
}

// bytes of the fragment planes of B videos (workspace of the row-tile form)
size_t sf_attn_rows_plane_bytes(int B) { return (size_t)B * NH * 6 * PL * 2; }

static int ar_dbg() {
  static const int v = sf_dbg("lf");
  return v;
}

static int launch_core(const sf_tfm_layer& w, float* x2, void* planes, int B, int L, int Lq, hipStream_t st) {
  CoreArgs C;
  C.planes = (const __bf16*)planes; C.wo_p = (const uint4*)w.attn_out_packed; C.bo = w.out_proj_b; C.x2 = x2; C.L = L; C.Lq = Lq;
  C.dbg = ar_dbg();
  SF_TRY(sf_ensure_dyn_lds((const void*)attn_core_kernel, K2_LDS));
  hipLaunchKernelGGL(attn_core_kernel, dim3(B), dim3(NT), K2_LDS, st, C);
  SF_CHECK_LAUNCH();
  return 0;
}

// x2 [B*Lq][256] = x + out_proj(MHA(LN1(x))) + b_o for the last Lq rows of every video, as two launches.
//   mode 0: x [B][L][256] (x_batch_stride floats per video);  mode 1: the sum of four chunk partials xparts_stride floats apart;
//   mode 2: ring rows + position table (layer 0 of a rollout step).
// planes: sf_attn_rows_plane_bytes(B) bytes whose v^T planes hold finite values in the key positions >= L (zeroed once per rollout).
int sf_attn_rows_ex(int mode, const float* xin, long long x_batch_stride, long long xparts_stride, const float* pe, int f0,
                    int ring_frames, int nslots, const sf_tfm_layer& w, float eps, float* x2, void* planes, int B, int L, int Lq,
                    hipStream_t st) {
  if (!w.attn_in_packed || !w.attn_out_packed)
    return sf_set_err(-1, "invalid argument: fused attention needs packed weights (sf_pack_attn_weights)", __FILE__, __LINE__);
  if (!planes || L < 1 || L > 64 || Lq < 1 || Lq > L || (long long)B * L >= (1 << 22))
    return sf_set_err(-1, "invalid argument: row-tile attention needs planes, 1 <= Lq <= L <= 64 and B * L < 2^22", __FILE__, __LINE__);
  RowsArgs A;
  A.xin = xin; A.x_batch_stride = x_batch_stride; A.xparts_stride = xparts_stride; A.pe = pe; A.f0 = f0;
  A.ring_frames = ring_frames; A.nslots = nslots; A.ln_g = w.norm1_g; A.ln_b = w.norm1_b; A.ln_eps = eps;
  A.wqkv_p = (const uint4*)w.attn_in_packed; A.bias = w.in_proj_b; A.planes = (__bf16*)planes; A.x2 = x2;
  A.B = B; A.L = L; A.Lq = Lq; A.dbg = ar_dbg();
  // rows per tile: 64 -- twice the workgroups of 128-row tiles at half the time each (the launch is a link of a dependent chain; a
  // unit of 128 videos alone: 14.6 vs 17.6 ms, C2 417 vs 412 k frames/s, C5 385 vs 381 k) for twice the weight stream out of the
  // L2s
  const int TRr = 64;
  A.nt = (B * L + TRr - 1) / TRr;
  const int ntq = (B * Lq + TRr - 1) / TRr;
  int extra = 0;
  if (Lq == L) {
    A.ng = 3;
  } else {
    A.ng = 2;
    extra = ntq;
  }
  A.main_blocks = A.nt;
  const int blocks = A.nt + extra;
  const size_t lds = (size_t)2 * TRr * AP * 2 + 2 * D * 4;
  sf_prof_begin(SF_K_MHA, st, 6.0 * B * L * (double)D * D + 4.0 * (double)B * NH * Lq * L * HD + 2.0 * B * Lq * (double)D * D);
  // (a failed LDS-attribute call closes the class timer's pending event before it returns)
#define SF_QKV_TRY(x)                 \
  do {                                \
    const int rc_ = (x);              \
    if (rc_) {                        \
      sf_prof_end(SF_K_MHA, st);      \
      return rc_;                     \
    }                                 \
  } while (0)
#define SF_QKV_LAUNCH(RING_, PART_)                                                                                         \
  do {                                                                                                                      \
    if (TRr == 128) {                                                                                                       \
      SF_QKV_TRY(sf_ensure_dyn_lds((const void*)qkv_rows_kernel<RING_, PART_, 128>, lds));                                  \
      hipLaunchKernelGGL((qkv_rows_kernel<RING_, PART_, 128>), dim3(blocks), dim3(NT), lds, st, A);                         \
    } else {                                                                                                                \
      SF_QKV_TRY(sf_ensure_dyn_lds((const void*)qkv_rows_kernel<RING_, PART_, 64>, lds));                                   \
      hipLaunchKernelGGL((qkv_rows_kernel<RING_, PART_, 64>), dim3(blocks), dim3(NT), lds, st, A);                          \
    }                                                                                                                       \
  } while (0)
  if (mode == 2)
    SF_QKV_LAUNCH(true, false);
  else if (mode == 1)
    SF_QKV_LAUNCH(false, true);
  else
    SF_QKV_LAUNCH(false, false);
#undef SF_QKV_LAUNCH
#undef SF_QKV_TRY
  SF_CHECK_LAUNCH();
  const int rc = launch_core(w, x2, planes, B, L, Lq, st);
  sf_prof_end(SF_K_MHA, st);
  return rc;
}

// The second launch of the row-tile attention block alone: q, k, v^T planes (written by qkv_rows_kernel or by the fused FFN + q|k|v tile
// launch of ffn_tile.hip) and the parked residual rows in x2 -> finished rows x2 [B * Lq][256]
int sf_attn_core_ex(const sf_tfm_layer& w, float* x2, void* planes, int B, int L, int Lq, hipStream_t st) {
  if (!w.attn_out_packed || !planes || !x2 || L < 1 || L > 64 || Lq < 1 || Lq > L)
    return sf_set_err(-1, "invalid argument: attention core needs packed out-projection weights, planes and 1 <= Lq <= L <= 64", __FILE__, __LINE__);
  sf_prof_begin(SF_K_MHA, st, 4.0 * (double)B * NH * Lq * L * HD + 2.0 * B * Lq * (double)D * D);
  const int rc = launch_core(w, x2, planes, B, L, Lq, st);
  sf_prof_end(SF_K_MHA, st);
  return rc;
}

extern "C" size_t sf_attn_rows_planes_bytes(int B) { return B > 0 ? sf_attn_rows_plane_bytes(B) : 0; }

// Kernel-level entry point of the row-tile attention block (include/slotformer_hip.h), for tests against a plain reference.
extern "C" int sf_attn_block_rows_f32(const sf_tfm_layer* w, const float* x, float* out, void* planes, int B, int L, int Lq, void* stream) {
  SF_REQUIRE(w && x && out && planes && B > 0, "sf_attn_block_rows_f32: null pointer / empty problem");
  SF_REQUIRE(L >= 1 && L <= 64 && Lq >= 1 && Lq <= L, "sf_attn_block_rows_f32: needs 1 <= Lq <= L <= 64");
  SF_REQUIRE(w->norm1_g && w->norm1_b && w->in_proj_b && w->out_proj_b && w->attn_in_packed && w->attn_out_packed,
             "sf_attn_block_rows_f32: null weight (packed attention weights needed)");
  hipError_t e = hipMemsetAsync(planes, 0, sf_attn_rows_plane_bytes(B), (hipStream_t)stream);
  if (e != hipSuccess) return sf_set_err((int)e, hipGetErrorString(e), __FILE__, __LINE__);
  return sf_attn_rows_ex(0, x, (long long)L * D, 0, nullptr, 0, 1, 1, *w, 1e-5f, out, planes, B, L, Lq, (hipStream_t)stream);
}

extern "C" int sf_debug_read_ts_rows(long long* out32) {
  hipError_t e = hipMemcpyFromSymbol(out32, HIP_SYMBOL(ar_ts), sizeof(long long) * 32);
  return e == hipSuccess ? 0 : (int)e;
}
