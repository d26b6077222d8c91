// FFN block of a rollout layer in its ROW-TILE form (per-call option ffn_tile = 1): ONE workgroup per 64-row tile runs all four
// 256-wide hidden chunks one after the other and writes FINISHED rows
//     y = x2 + lin2(relu(lin1(LN2(x2))))            (nn.TransformerEncoderLayer, norm_first; slotformer.py:72-80)
// The chunk-partial kernels of layer_fused.hip give every (row tile, hidden chunk) its own workgroup: each ingests and normalises
// the tile's rows again, holds its CU for 24-28 us around 10 us of MFMAs, and leaves four partial tensors for the next launch to sum.
// Here the rows are ingested and normalised once, the weight fragments of W1 / W2 (2 MB per tile) stream through a register ring
// without touching LDS, the only workgroup barriers are the two hand-overs of the hidden planes per chunk, and the next launch reads
// 1 KB per row instead of 4.
// Same products in the same order as ffn_body / ffn_wide_parts_kernel, the four chunks' FFN2 partials each from a fresh
// accumulator and summed ((p0 + p1) + p2) + p3 with the residual and the bias on p0 -- the bits of the chunk-partial forms.
#include "../../include/slotformer_hip.h"
#include "sf_internal.h"
#include "layer_fused.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int NT = 512, D = 256, HC = 256, NCH = 4;
constexpr int AP = D + 8;                          // bf16 pitch of the LN2 / hidden planes
constexpr int ROWS = 64;
constexpr size_t PLANES = (size_t)2 * ROWS * AP * 2;   // hi | lo planes of 64 rows: 67,584 B
constexpr size_t FT_LDS = 2 * PLANES;                  // LN2 planes + hidden planes
static_assert(FT_LDS <= 160 * 1024, "LDS budget");
constexpr int RD = 4;                              // weight ring depth in two-k-step chunks: RD - 1 in flight

__device__ __forceinline__ void split4(__bf16* hp, __bf16* lp, int off, f32x4 v) {
  const bf16x4 hi = __builtin_convertvector(v, bf16x4);
  const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
  *(bf16x4*)(hp + off) = hi;
  *(bf16x4*)(lp + off) = lo;
}

struct TileArgs {
  const float* x2;          // [M][256] finished rows of the attention block
  const float *ln_g, *ln_b;
  float ln_eps;
  const uint4* w1p;
  const float* b1;
  const uint4* w2p;
  const float* b2;
  float* y;                 // [M][256]
  int M, dbg;
};
}  // namespace

__device__ long long ft_ts[16];
#define FTS(i) do { if (A.dbg && blockIdx.x == 0 && threadIdx.x == 0) ft_ts[i] = wall_clock64(); } while (0)

// The FFN of one 64-row tile: on return ysum[rb][g] holds y[row 32 rb + (lane & 31)][32 wave + 8 g + 4 (lane >> 5) .. + 3]; every wave has passed the
// last FFN2 (no barrier behind it).  LDS: LN2 planes at 0, hidden planes at PLANES.
__device__ __forceinline__ void ffn_tile_body(const TileArgs& A, f32x4 (&ysum)[2][4]) {
#pragma clang fp contract(off)
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int row0 = blockIdx.x * ROWS, M = A.M;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* Xh = (__bf16*)smem;                          // LN2(x2) planes [64][AP], resident for the four chunks
  __bf16* Xl = Xh + ROWS * AP;
  __bf16* Hh = (__bf16*)((char*)smem + PLANES);        // hidden planes of the current chunk
  __bf16* Hl = Hh + ROWS * AP;
  FTS(0);
  // ---- weight ring: global chunk index gc = (c * 2 + phase) * 8 + k-step pair; phase 0 = W1 of chunk c, 1 = W2 of chunk c ----
  bf16x8 ring[RD][2][2];
  const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(A.w1p), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(A.w2p), 0, 0x7fffffff, 0x00020000);
  const unsigned wlane = (unsigned)(lane * 16), wwave = (unsigned)(wave * 2048);
  auto load_chunk = [&](int gc) {
    const int c = gc >> 4, ph = (gc >> 3) & 1, ks0 = (gc & 7) * 2;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const unsigned so = wwave + (unsigned)(((c * 16 + ks0 + k) * 8) * 2048 + pl * 1024);
        ring[gc % RD][k][pl] = __builtin_bit_cast(bf16x8, ph ? __builtin_amdgcn_raw_buffer_load_b128(w2r, wlane, so, 0)
                                                              : __builtin_amdgcn_raw_buffer_load_b128(w1r, wlane, so, 0));
      }
  };
  constexpr int NGC = NCH * 2 * 8;   // 64 ring chunks
  // ---- rows: ffn_wide_parts_kernel's LayerNorm (wave owns rows wave + 8 i, lane = float4 column) ----
  const f32x4 lng = *(const f32x4*)(A.ln_g + 4 * lane), lnb = *(const f32x4*)(A.ln_b + 4 * lane);
  f32x4 xr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) xr[i] = *(const f32x4*)(A.x2 + (long long)min(row0 + wave + 8 * i, M - 1) * D + 4 * lane);
#pragma unroll
  for (int q = 0; q < RD - 1; ++q) load_chunk(q);
  {
    float mean[8], rstd[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) mean[i] = sf_sum64((xr[i][0] + xr[i][1]) + (xr[i][2] + xr[i][3])) * (1.0f / D);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 dv = xr[i] - mean[i];
      rstd[i] = 1.0f / sqrtf(sf_sum64((dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3])) * (1.0f / D) + A.ln_eps);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) split4(Xh, Xl, (wave + 8 * i) * AP + 4 * lane, (xr[i] - mean[i]) * rstd[i] * lng + lnb);
  }
  const int tok = lane & 31, kg = lane >> 5, nb = wave * 32 + 4 * kg;
  __syncthreads();   // LN2 planes
  FTS(1);
  const int ao = tok * AP + 8 * kg;
  // a0 / a1 += W . P^T over 16 k-steps: fragments of planes (Ph, Pl) for both row blocks, weights from the ring (chunks gc0 .. gc0 + 7).
  // The plane fragments of k-step ks + 1 are requested BEFORE the six MFMAs of k-step ks (two register buffers), so the LDS latency
  // sits under the matrix pipe; stop_at_end: no weight requests across the end of this product (see chunk 0 below).
  auto product = [&](const __bf16* Ph, const __bf16* Pl, const int gc0, const bool stop_at_end, f32x16& a0, f32x16& a1) {
    bf16x8 xf[2][4];
    auto read_x = [&](int ks, int buf) {
      xf[buf][0] = *(const bf16x8*)(Ph + ao + ks * 16);
      xf[buf][1] = *(const bf16x8*)(Pl + ao + ks * 16);
      xf[buf][2] = *(const bf16x8*)(Ph + ao + 32 * AP + ks * 16);
      xf[buf][3] = *(const bf16x8*)(Pl + ao + 32 * AP + ks * 16);
    };
    read_x(0, 0);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int gc = gc0 + (ks >> 1), k = ks & 1;
      const bool req = k == 0 && gc + RD - 1 < NGC && !(stop_at_end && (ks >> 1) + RD - 1 >= 8);
      if (req) load_chunk(gc + RD - 1);
      if (ks + 1 < 16) read_x(ks + 1, (ks + 1) & 1);
      const bf16x8 w0 = ring[gc % RD][k][0], w1 = ring[gc % RD][k][1];
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xf[ks & 1][1], a0, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, xf[ks & 1][0], a0, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xf[ks & 1][0], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xf[ks & 1][3], a1, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, xf[ks & 1][2], a1, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xf[ks & 1][2], a1, 0, 0, 0);
      if (req) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
      if (ks + 1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    // ---- FFN1 of chunk c: hidden columns c * 256 + 32 wave .. + 31 of both row blocks ----
    f32x4 b1v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) b1v[g] = *(const f32x4*)(A.b1 + c * HC + nb + 8 * g);
    f32x16 a0, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) a0[r] = a1[r] = 0.f;
    product(Xh, Xl, (c * 2) * 8, false, a0, a1);
    if (c > 0) __syncthreads();   // every wave has read the hidden planes of chunk c - 1 (its FFN2)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bv = b1v[g];
      f32x4 h0, h1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        h0[q] = fmaxf(a0[4 * g + q] + bv[q], 0.f);
        h1[q] = fmaxf(a1[4 * g + q] + bv[q], 0.f);
      }
      split4(Hh, Hl, tok * AP + nb + 8 * g, h0);
      split4(Hh, Hl, (32 + tok) * AP + nb + 8 * g, h1);
    }
    __syncthreads();   // hidden planes of chunk c
    if (c == 0) FTS(2);
    // ---- FFN2 partial of chunk c: output columns 32 wave .. + 31, fresh accumulators ----
#pragma unroll
    for (int r = 0; r < 16; ++r) a0[r] = a1[r] = 0.f;
    product(Hh, Hl, (c * 2 + 1) * 8, c == 0, a0, a1);
    // running sum of the chunk partials: p0 = y0 + (x2 + b2), then (s + p1) + p2 ... -- the consumer's order
    if (c == 0) {
      // the residual + bias ride on chunk 0: the rows just read for the LayerNorm (L2 hits), requested together with the first
      // fragments of chunk 1 into the registers the ring has left (requested earlier they were spilled behind vmcnt(0) waits)
      // (an opaque zero that depends on the last MFMA: the row requests cannot be hoisted into the loop above)
      int late = 0;
      asm volatile("" : "+v"(late) : "v"(a0[0]), "v"(a1[0]));
      f32x4 b2v[4], r0[4], r1[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        b2v[g] = *(const f32x4*)(A.b2 + nb + 8 * g + late);
        r0[g] = *(const f32x4*)(A.x2 + (long long)min(row0 + tok, M - 1) * D + nb + 8 * g + late);
        r1[g] = *(const f32x4*)(A.x2 + (long long)min(row0 + 32 + tok, M - 1) * D + nb + 8 * g + late);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 p0 = {a0[4 * g], a0[4 * g + 1], a0[4 * g + 2], a0[4 * g + 3]};
        const f32x4 p1 = {a1[4 * g], a1[4 * g + 1], a1[4 * g + 2], a1[4 * g + 3]};
        ysum[0][g] = p0 + (r0[g] + b2v[g]);
        ysum[1][g] = p1 + (r1[g] + b2v[g]);
        // (pinned here: left to the scheduler the sums sank to the end of the kernel and the rows were spilled meanwhile)
        asm volatile("" : "+v"(ysum[0][g]), "+v"(ysum[1][g]));
      }
      // the first fragments of chunk 1 (behind the sums: their registers held the rows)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < RD - 1; ++q) load_chunk(16 + q);
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 p0 = {a0[4 * g], a0[4 * g + 1], a0[4 * g + 2], a0[4 * g + 3]};
        const f32x4 p1 = {a1[4 * g], a1[4 * g + 1], a1[4 * g + 2], a1[4 * g + 3]};
        ysum[0][g] = ysum[0][g] + p0;
        ysum[1][g] = ysum[1][g] + p1;
      }
    }
    if (c == 0) FTS(3);
  }
  FTS(4);
}

__global__ __launch_bounds__(NT) void ffn_tile_kernel(TileArgs A) {
  f32x4 ysum[2][4];
  ffn_tile_body(A, ysum);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tok = lane & 31, nb = wave * 32 + 4 * (lane >> 5), row0 = blockIdx.x * ROWS, M = A.M;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int row = row0 + 32 * rb + tok;
    if (row < M) {
#pragma unroll
      for (int g = 0; g < 4; ++g) *(f32x4*)(A.y + (long long)row * D + nb + 8 * g) = ysum[rb][g];
    }
  }
  FTS(5);
}

// ------------------------------------------------------------------------------------------------
// FFN of layer l AND LN1 + q|k|v of layer l + 1 on the same 64-row tile in one launch (per-call option ffn_tile = 2, behind the
// row-tile attention form): the finished rows y never leave the workgroup -- they go to an f32 tile in LDS (over the dead hidden
// planes), are normalised with layer l + 1's LN1 into the (dead) LN2 plane region, and the three projection groups stream their
// weight fragments exactly as qkv_rows_kernel does (attn_rows.hip: same products, same order, same plane layout).  The rows the next
// attention core needs as its residual are parked in ITS output buffer (xpark, another buffer than x2: a parked row of the last
// layer's shorter row space would land on rows another tile has not read yet).  One launch, one first-byte latency and one ingest
// less per layer of a rollout step.
namespace {
constexpr int HD = 32, NH = 8, PL = 2048;      // head width, heads, bf16 elements of a fragment plane (attn_rows.hip)
constexpr int YP = D + 4;                      // f32 pitch of the y tile
static_assert((size_t)ROWS * YP * 4 <= PLANES, "the y tile fits over the hidden planes");
constexpr size_t FQ_LDS = 2 * PLANES + 2 * D * 4;   // + LN1 gamma | beta
static_assert(FQ_LDS <= 160 * 1024, "LDS budget");
struct NextArgs {
  const float *ln_g, *ln_b;     // LN1 of the next layer
  const uint4* wqkv_p;          // its packed q|k|v weights (sf_pack_attn_weights)
  const float* bias;            // its in_proj bias [768]
  __bf16* planes;               // fragment planes [B][8][6][2048]
  float* xpark;                 // [B * Lq][256]: residual rows of the next attention block
  int B, L, Lq;                 // videos, rows per video, query rows per video of the NEXT attention block
};
__device__ __forceinline__ int vpos(int t) { return (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1); }
__device__ __forceinline__ int div_rows(int m, int L, float invL) {
  int q = (int)((float)m * invL);
  q -= (q * L > m);
  q += ((q + 1) * L <= m);
  return q;
}
}  // namespace

__global__ __launch_bounds__(NT) void ffn_qkv_tile_kernel(TileArgs A, NextArgs N) {
#pragma clang fp contract(off)
  f32x4 ysum[2][4];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* GB = (float*)((char*)smem + 2 * PLANES);
  if (t < 128) *(f32x4*)(GB + 4 * t) = *(const f32x4*)((t < 64 ? N.ln_g : N.ln_b) + 4 * (t & 63));   // (visible behind the FFN's barriers)
  ffn_tile_body(A, ysum);
  const int tok = lane & 31, kg = lane >> 5, nb = wave * 32 + 4 * kg, row0 = blockIdx.x * ROWS, M = A.M;
  const int L = N.L, Lq = N.Lq, nq0 = L - Lq;
  const float invL = 1.0f / (float)L;
  // ---- weight ring of the projections (qkv_rows_kernel's): chunk gc = group * 8 + k-step pair, slot gc & 3 ----
  bf16x8 ring[4][2][2];
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(N.wqkv_p), 0, 0x7fffffff, 0x00020000);
  const unsigned hpb = (unsigned)(((wave >> 1) * 6 + 3 * (wave & 1)) * 16 * 2048);
  auto load_chunk = [&](int gc) {
    const int g = gc / 8, ks0 = (gc % 8) * 2;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        ring[gc & 3][k][pl] = __builtin_bit_cast(
            bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, (unsigned)(lane * 16), hpb + (unsigned)((g * 16 + ks0 + k) * 2048 + pl * 1024), 0));
  };
  load_chunk(0);
  load_chunk(1);
  load_chunk(2);
  __syncthreads();   // every wave is done with the hidden planes (its last FFN2): the y tile takes their place
  float* Y = (float*)((char*)smem + PLANES);       // [64][YP] f32
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int r = 32 * rb + tok, m = row0 + r;
    int b = 0, tk = 0;
    if (m < M) {
      b = div_rows(m, L, invL);
      tk = m - b * L;
    }
    const bool park = m < M && tk >= nq0;
    float* dst = N.xpark + ((long long)b * Lq + (tk - nq0)) * D + nb;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *(f32x4*)(Y + r * YP + nb + 8 * g) = ysum[rb][g];
      if (park) *(f32x4*)(dst + 8 * g) = ysum[rb][g];   // the residual of the next attention block's query rows
    }
  }
  __syncthreads();   // y tile
  // ---- LN1 of the next layer: qkv_rows_kernel's arithmetic (16 lanes per row, float4 column c4 of every 64-wide chunk) ----
  __bf16* Ah = (__bf16*)smem;   // [64][AP] (the LN2 planes are dead)
  __bf16* Al = Ah + ROWS * AP;
  {
    const int c4 = t & 15, r0 = t >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = 32 * p + r0;
      const bool ok = row0 + r < M;
      f32x4 vv[4];
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) vv[kc] = *(const f32x4*)(Y + min(r, M - 1 - row0) * YP + kc * 64 + 4 * c4);
      float sm = 0.f;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) sm += (vv[kc][0] + vv[kc][1]) + (vv[kc][2] + vv[kc][3]);
      const float mu = sf_sum16(sm) / (float)D;
      float vs = 0.f;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        const f32x4 dv = vv[kc] - mu;
        vs += (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]);
      }
      const float rs = 1.0f / sqrtf(sf_sum16(vs) / (float)D + A.ln_eps);
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        const int k = kc * 64 + 4 * c4;
        const f32x4 gm = *(const f32x4*)(GB + k), be = *(const f32x4*)(GB + D + k);
        split4(Ah, Al, r * AP + k, ok ? (vv[kc] - mu) * rs * gm + be : zero4);
      }
    }
  }
  __syncthreads();   // LN1 planes
  FTS(6);
  // ---- q, k, v of head `wave` for the tile's rows (qkv_rows_body's group loop, two row blocks) ----
  const float scale = 1.0f / sqrtf((float)HD);
  const int ao = tok * AP + 8 * kg;
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const bool splitk = (wave & 1) && g >= 1;   // the k / v blocks of odd heads: lower + upper K half (attn_body)
    const float* bq = N.bias + g * D + wave * HD;
    f32x4 bv4[4];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) bv4[gq] = *(const f32x4*)(bq + 8 * gq + 4 * kg);
    f32x16 acc[2], sav[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int gc = g * 8 + c;
      if (gc + 3 < 24) load_chunk(gc + 3);
      if (c == 4 && splitk) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
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
        for (int rb = 0; rb < 2; ++rb) {
          const bf16x8 xh = *(const bf16x8*)(Ah + ao + rb * 32 * AP + ks * 16), xl = *(const bf16x8*)(Al + ao + rb * 32 * AP + ks * 16);
          acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xl, acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, xh, acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xh, acc[rb], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (splitk) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) acc[rb] = sav[rb] + acc[rb];
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int m = row0 + rb * 32 + tok;
      if (m >= M) continue;
      const int b = div_rows(m, L, invL), tk = m - b * L;
      if (g < 2) {
        if (g == 0 && tk < nq0) continue;   // q of rows that are no query rows (last layer) is never read
        __bf16* ph = N.planes + ((long long)(b * NH + wave) * 6 + 2 * g) * PL + tk * HD + 4 * kg;
        const float mul = g == 0 ? scale : 1.f;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const f32x4 bv = bv4[gq];
          split4(ph, ph + PL, 8 * gq,
                 f32x4{(acc[rb][4 * gq] + bv[0]) * mul, (acc[rb][4 * gq + 1] + bv[1]) * mul, (acc[rb][4 * gq + 2] + bv[2]) * mul,
                       (acc[rb][4 * gq + 3] + bv[3]) * mul});
        }
      } else {
        __bf16* pv = N.planes + ((long long)(b * NH + wave) * 6 + 4) * PL + vpos(tk);
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
    FTS(7 + g);
  }
  FTS(5);
}

// x2 [M][256] (finished rows of the attention block) -> y [M][256] = x2 + FFN(LN2(x2)), one workgroup per 64 rows
int sf_ffn_tile_ex(const float* x2, const sf_tfm_layer& w, float eps, float* y, int M, int ffn, hipStream_t st) {
  if (!w.lin1_packed || !w.lin2_packed || ffn != NCH * HC)
    return sf_set_err(-1, "invalid argument: the row-tile FFN needs packed weights (sf_pack_ffn_weights) and ffn == 1024", __FILE__, __LINE__);
  static const int dbg = sf_dbg("lf") & 16;
  TileArgs A;
  A.x2 = x2; A.ln_g = w.norm2_g; A.ln_b = w.norm2_b; A.ln_eps = eps; A.w1p = (const uint4*)w.lin1_packed; A.b1 = w.lin1_b;
  A.w2p = (const uint4*)w.lin2_packed; A.b2 = w.lin2_b; A.y = y; A.M = M; A.dbg = dbg;
  SF_TRY(sf_ensure_dyn_lds((const void*)ffn_tile_kernel, FT_LDS));
  sf_prof_begin(SF_K_FFN, st, 4.0 * M * (double)D * ffn);
  hipLaunchKernelGGL(ffn_tile_kernel, dim3((M + ROWS - 1) / ROWS), dim3(NT), FT_LDS, st, A);
  sf_prof_end(SF_K_FFN, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// FFN of layer `w` on finished rows x2 [M][256] fused with LN1 + q|k|v of layer `wn` (M = B * L rows of B videos): fragment planes + parked
// residual rows (xpark [B * Lq][256], a buffer other than x2) for sf_attn_core_ex
int sf_ffn_qkv_tile_ex(const float* x2, const sf_tfm_layer& w, const sf_tfm_layer& wn, float eps, float* xpark, void* planes, int B, int L, int Lq,
                       int ffn, hipStream_t st) {
  if (!w.lin1_packed || !w.lin2_packed || ffn != NCH * HC || !wn.attn_in_packed)
    return sf_set_err(-1, "invalid argument: the fused FFN + q|k|v tile launch needs packed FFN and attention weights and ffn == 1024", __FILE__, __LINE__);
  if (!planes || !xpark || xpark == x2 || L < 1 || L > 64 || Lq < 1 || Lq > L || (long long)B * L >= (1 << 22))
    return sf_set_err(-1, "invalid argument: fused FFN + q|k|v tiles need planes, a parking buffer other than the input, 1 <= Lq <= L <= 64", __FILE__, __LINE__);
  static const int dbg = sf_dbg("lf") & 16;
  const int M = B * L;
  TileArgs A;
  A.x2 = x2; A.ln_g = w.norm2_g; A.ln_b = w.norm2_b; A.ln_eps = eps; A.w1p = (const uint4*)w.lin1_packed; A.b1 = w.lin1_b;
  A.w2p = (const uint4*)w.lin2_packed; A.b2 = w.lin2_b; A.y = nullptr; A.M = M; A.dbg = dbg;
  NextArgs N;
  N.ln_g = wn.norm1_g; N.ln_b = wn.norm1_b; N.wqkv_p = (const uint4*)wn.attn_in_packed; N.bias = wn.in_proj_b; N.planes = (__bf16*)planes;
  N.xpark = xpark; N.B = B; N.L = L; N.Lq = Lq;
  SF_TRY(sf_ensure_dyn_lds((const void*)ffn_qkv_tile_kernel, FQ_LDS));
  sf_prof_begin(SF_K_FFN, st, 4.0 * M * (double)D * ffn + 6.0 * M * (double)D * D);
  hipLaunchKernelGGL(ffn_qkv_tile_kernel, dim3((M + ROWS - 1) / ROWS), dim3(NT), FQ_LDS, st, A, N);
  sf_prof_end(SF_K_FFN, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// Kernel-level entry point (include/slotformer_hip.h): the FFN block on finished rows, for tests against a plain reference.
extern "C" int sf_ffn_block_rows_f32(const sf_tfm_layer* w, const float* x2, float* y, int M, int ffn, void* stream) {
  SF_REQUIRE(w && x2 && y && M > 0, "sf_ffn_block_rows_f32: null pointer / empty problem");
  SF_REQUIRE(w->norm2_g && w->norm2_b && w->lin1_b && w->lin2_b && w->lin1_packed && w->lin2_packed,
             "sf_ffn_block_rows_f32: null weight (packed FFN weights needed)");
  return sf_ffn_tile_ex(x2, *w, 1e-5f, y, M, ffn, (hipStream_t)stream);
}

extern "C" int sf_debug_read_ts_ffn_tile(long long* out16) {
  hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(ft_ts), sizeof(long long) * 16);
  return e == hipSuccess ? 0 : (int)e;
}
