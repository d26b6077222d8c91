// The slot branch of the SAVi encode as ONE video-stationary launch (round 6; savi.py:76-100 iterations, :393-402 per-step chain).
//
// Per time step the reference runs, per video, a strictly sequential chain: Slot-Attention iteration -> GRU / MLP slot update -> next iteration ->
// ... -> predictor + kernel distribution + sampling of the NEXT step.  Videos never meet.  The per-launch form (sa_attn_tile_kernel over the whole
// batch, then a seven-workgroup slot update, 26 launches per six frames) spends 1.2 ms of a 2.84 ms encode lane with most of its 128 CUs idle:
// every launch is a boundary, the update occupies 7 CUs, and the attention launch is bound by exact-f32 MFMAs on 8 slots padded to 16.
// Here ONE workgroup owns ONE video for the whole clip and walks T steps x `iters` iterations without leaving its CU:
//   * attention over the frame's HW pixels: the normalised features arrive as bf16 hi | lo rows (pixel_feat_stream_kernel PLANES form: 512 B per
//     pixel), a wave streams its 1/8 of the pixels through a 16 KB LDS tile of 32 rows (registers one tile ahead), logits X . Q^T and weighted sums
//     A^T . X both on v_mfma_f32_16x16x32_bf16 as split-bf16 products (hi.lo + lo.hi + hi.hi): the A operand of the logits is a 16-byte row read,
//     the B operand of the sums (contraction over PIXELS) comes out of the same row-major tile through ds_read_b64_tr_b16; a 16-byte-chunk XOR
//     swizzle keyed on (row & 3, row bit 3) makes the writes and both kinds of reads bank-conflict free without padding;
//   * softmax over the slots inside the accumulator's 16-lane rows (two DPP all-reduces), + eps, the denominators in a register;
//   * the eight waves' sums meet in LDS, one record [N][128] + [N] goes to memory, and the SAME workgroup runs the matrix-core slot update on it
//     (um_body of slot_update_body.h with R = N rows: GRU, residual MLP, q of the next iteration; on a step's last iteration its NEXT form: the
//     predictor, kernel distribution, sampling and first q of the following step).
// A batch of 32 videos is 32 workgroups on 32 CUs for ~0.4 ms instead of 26 whole-lane launches: 16 CU-ms instead of 154.
#include "slot_update_body.h"
#include "slot_chain.h"

typedef unsigned int u32x4c __attribute__((ext_vector_type(4)));

namespace {

constexpr int SC_D = 128, SC_NT = 512, SC_NW = 8, SC_TP = 32;          // slot size, threads, waves, pixels per tile
constexpr int SC_TILE = 2 * SC_TP * 256;                               // hi | lo planes of [32][128] bf16: 16 KB per wave
constexpr int SC_ATP = 20;                                             // f32 pitch of a pixel row of the attention tile (16 slots + 4: 80 B)
constexpr int SC_AT = SC_TP * SC_ATP * 4;                              // attention tile a[32 pixels][16 slots] f32: 2560 B per wave
constexpr int SC_REDP = SC_D + 4;                                      // f32 pitch of a reduction row
constexpr size_t SC_LDS_SA = (size_t)SC_NW * (SC_TILE + SC_AT);        // 151,552 B
static_assert((size_t)SC_NW * 8 * SC_REDP * 4 + SC_NW * 8 * 4 <= (size_t)SC_NW * SC_TILE, "the reduction rows fit over the dead tiles");
// behind both phases' regions: the record (num [8][128], den [8 -> 32]) and the queries [8][128] -- they pass between the attention pass and the update
// through LDS (generic pointers in UmArgs), not through memory
constexpr size_t SC_REC = SC_LDS_SA > UM_LDS_NEXT ? SC_LDS_SA : UM_LDS_NEXT;
constexpr size_t SC_ARGS = SC_REC + (size_t)(8 * SC_D + 32 + 8 * SC_D) * 4;   // the update's arguments: UmArgs, UmVar
constexpr size_t SC_LDS = SC_ARGS + ((sizeof(UmArgs) + 15) / 16) * 16 + ((sizeof(UmVar) + 15) / 16) * 16;
static_assert(SC_LDS <= 160 * 1024, "slot chain: LDS budget");

// 16-byte chunk position of (row, chunk) inside a 256-byte plane row
__device__ __forceinline__ int sc_swz(int row) { return 2 * ((row & 3) | (((row >> 3) & 1) << 2)); }

__device__ __forceinline__ void sc_split8(const f32x4 a, const f32x4 b, bf16x8& hi, bf16x8& lo) {
  const bf16x4 h0 = __builtin_convertvector(a, bf16x4), h1 = __builtin_convertvector(b, bf16x4);
  const bf16x4 l0 = __builtin_convertvector(a - __builtin_convertvector(h0, f32x4), bf16x4);
  const bf16x4 l1 = __builtin_convertvector(b - __builtin_convertvector(h1, f32x4), bf16x4);
  hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
  lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
}
#define SC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

#ifdef SC_STAMPS
__device__ long long sc_ts[32];
__device__ int sc_tsn;
#define SCTS() do { if (blockIdx.x == 0 && threadIdx.x == 0 && sc_tsn < 32) sc_ts[sc_tsn++] = wall_clock64(); } while (0)
#else
#define SCTS() do { } while (0)
#endif

// One Slot-Attention pass over `NW * ppw` pixels of ONE frame by a workgroup of NW waves: the sums num [N][128], den [N] of those pixels.
//   frame: [HW][256] bf16 (hi 128 | lo 128 per pixel);  pix_base: the workgroup's first pixel, ppw: pixels per wave (a multiple of 32);
//   qg: [N][128] f32 queries (unscaled);  pnum / pden: where the sums go;  attn: NULL or this frame's [N][HW] rows.
//   GIO false (the slot chain): queries and record in LDS;  true (the batch-wide launch): both in memory, and a second, zeroed record behind the first
//   (the slot update reads every second record: slot_update_body.h, pstep 2).
template <bool ATTN, int NW, bool GIO>
__device__ __forceinline__ void sc_attend_core(const __bf16* __restrict__ frame, const float* qg, float scale, float eps, int HW, int N, int pix_base, int ppw,
                                               float* pnum, float* pden, long long zero_off_num, long long zero_off_den, float* __restrict__ attn, char* smem,
                                               int rev) {
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int i16 = lane & 15, kb = lane >> 4;
  SCTS();
  char* tile = smem + wave * SC_TILE;
  const int ntiles = ppw / SC_TP;
  const int pix0 = pix_base + wave * ppw;
  // ---- rows: instruction u brings pixel rows 2 u and 2 u + 1 of a tile whole (512 B each: hi | lo); TWO tiles ahead in registers (one tile in flight per
  //      wave left the pass latency-bound: 2.2 us per tile, 60 GB/s per CU).  rev: the tiles in descending order -- the second iteration over a frame
  //      starts with the rows the first one read last (the ones still in this XCD's L2) ----
  // (explicit address spaces: inside a non-inlined function the pointers are generic, and FLAT loads count on lgkmcnt as well -- every LDS wait of the
  //  tile loop then waited for the rows in flight: 2.2 us per tile whatever the prefetch depth)
  typedef const u32x4c __attribute__((address_space(1))) * gvec;
  typedef float __attribute__((address_space(3))) * lflt;
  typedef f32x4 __attribute__((address_space(3))) * lvec;
  // (a non-inlined function receives even uniform arguments in vector registers: the frame base back into scalar registers, so that a load is
  //  `global_load v, v_off, s[base] offset:imm` -- four offset registers for the sixteen loads of a tile, not sixteen 64-bit addresses)
  const unsigned long long fb0 = (unsigned long long)frame + (unsigned long long)pix0 * 512;
  const unsigned long long fb = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(fb0 >> 32)) << 32) |
                                (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)fb0);   // (the builtin returns a SIGNED int)
  const unsigned voff = (lane >> 5) * 512 + (lane & 31) * 16;
  const lflt qL = (lflt)qg, pnL = (lflt)pnum, pdL = (lflt)pden;
  auto tile_at = [&](int k) { return rev ? ntiles - 1 - k : k; };
  const char __attribute__((address_space(1)))* src = (const char __attribute__((address_space(1)))*)fb;
  u32x4c stage[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) stage[u] = *(gvec)(src + (long long)tile_at(0) * (SC_TP * 512) + (voff + u * 1024));
  typedef char __attribute__((address_space(3))) * lchr;
  typedef const bf16x8 __attribute__((address_space(3))) * lfrag;
  const lchr tileL = (lchr)tile;
  const lflt atL = (lflt)(smem + NW * SC_TILE + wave * SC_AT);   // attention tile a[pixel][slot] f32, pitch SC_ATP floats
  // ---- queries as the A operand of the logits: lane (slot i = i16, k group kb) holds q[i][32 ks + 8 kb .. + 7] * scale * log2(e) (the softmax runs in
  //      base 2: one v_exp_f32 per value); zero beyond N slots ----
  // Slots 0-3 sit in rows 0-3 of the 16-row operand, slots 4-7 in rows 8-11: an accumulator then holds slots 0-3 in the lanes of 16-lane row 0 and
  // slots 4-7 in those of row 2 -- 32 lanes apart, one v_permlane32_swap away (rows 1 and 3 are padding)
  const int qslot = i16 < 4 ? i16 : (i16 >= 8 && i16 < 12 ? i16 - 4 : 99);
  const int sl0 = kb == 0 ? 0 : (kb == 2 ? 4 : 99);   // first of the four slots this lane's accumulator registers hold
  bf16x8 qh[4], ql[4];
  {
    const float s2 = scale * 1.4426950408889634f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
      if (qslot < N) {
        const int qo = qslot * SC_D + 32 * ks + 8 * kb;
        if constexpr (GIO) {
          typedef const f32x4 __attribute__((address_space(1))) * gq;
          a = *(gq)((const float __attribute__((address_space(1)))*)qg + qo) * s2;
          b = *(gq)((const float __attribute__((address_space(1)))*)qg + qo + 4) * s2;
        } else {
          a = *(lvec)(qL + qo) * s2;
          b = *(lvec)(qL + qo + 4) * s2;
        }
      }
      sc_split8(a, b, qh[ks], ql[ks]);
    }
  }
  // per-lane LDS offsets.  Tile write: lane = (row within the pair l >> 5, 16-byte chunk c = l & 31 of the 512-byte row: plane c >> 4, chunk c & 15)
  const int wplane = (lane & 31) >> 4, wchunk = lane & 15, wrow = lane >> 5;
  // logits B operand: pixel row i16 (+ 16 pb), chunk 4 ks + kb
  const int arow_off = i16 * 256, aswz = sc_swz(i16);
  // sums B operand (ds_read_b64_tr_b16): lane (j = i16, kb) points at row 8 kb + (j >> 2) (+ 4 h2), channels 16 cb + 4 (j & 3) .. + 3
  const int brow = 8 * kb + (i16 >> 2);
  const int bswz = sc_swz(brow) << 4;   // (the same for row + 4: bits 0, 1 and 3 of the row)
  const int boff = brow * 256 + ((i16 & 3) >> 1) * 16 + (i16 & 1) * 8;
  f32x4 nacc[8];
#pragma unroll
  for (int cb = 0; cb < 8; ++cb) nacc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // The logits come out TRANSPOSED, D[slot][pixel] = Q . X^T: lane (pixel i16, kb) holds slots 4 kb + r in its four registers, so the softmax over the
  // slots is three in-register steps and ONE exchange with the lane 32 away (v_permlane32_swap) -- with the slots across the lanes it was two 8-lane DPP
  // all-reduces and a division per value: 280 of a tile's 520 instructions, and the pass ran issue-bound at 1.75 us per tile.  Slots beyond N start
  // their accumulator at -3e38 (their exponential is 0) and get no eps.
  f32x4 sbias, epsv, den4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    sbias[r] = (sl0 + r < N) ? 0.f : -3.0e38f;
    epsv[r] = (sl0 + r < N) ? eps : 0.f;
  }
  // the value of `v` in the lane 32 away (layer_tok.hip lt_xother: whether the compiler gives the two copies of `v` one register or two, the partner's
  // value is result 0 in the upper half of the wave and result 1 in the lower -- the lane's OWN value is in neither when it is one register)
  const int hi32 = lane >> 5;
  auto other32 = [&](float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, hi32 ? r2[0] : r2[1]);
  };
#pragma unroll 1
  for (int k = 0; k < ntiles; ++k) {
    const int tix = tile_at(k);
    // ---- staged rows -> the tile (swizzled chunks); then the rows of the next tile, in flight while this one is multiplied ----
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int row = 2 * u + wrow;
      *(u32x4c __attribute__((address_space(3)))*)(tileL + wplane * 8192 + row * 256 + ((wchunk ^ sc_swz(row)) << 4)) = stage[u];
    }
#ifndef SC_NOLOAD
    if (k + 1 < ntiles) {
      const long long to = (long long)tile_at(k + 1) * (SC_TP * 512);
#pragma unroll
      for (int u = 0; u < 16; ++u) stage[u] = *(gvec)(src + to + (voff + u * 1024));
    }
#endif
    __builtin_amdgcn_wave_barrier();
    // ---- logits[slot][pixel] = Q . X^T: two 16-pixel blocks ----
    f32x4 acc[2] = {sbias, sbias};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const lchr ap = tileL + pb * 4096 + arow_off + (((4 * ks + kb) ^ aswz) << 4);
        const bf16x8 xh = *(lfrag)ap, xl = *(lfrag)(ap + 8192);
        acc[pb] = SC_MFMA(qh[ks], xl, acc[pb]);
        acc[pb] = SC_MFMA(ql[ks], xh, acc[pb]);
        acc[pb] = SC_MFMA(qh[ks], xh, acc[pb]);
      }
    // ---- acc[pb][r]: slot 4 kb + r of pixel 16 pb + i16 (base-2 logits).  a = softmax over the slots + eps -> the attention tile [pixel][slot] ----
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      const float m4 = fmaxf(fmaxf(acc[pb][0], acc[pb][1]), fmaxf(acc[pb][2], acc[pb][3]));
      const float mx = fmaxf(m4, other32(m4));
      f32x4 e;
#pragma unroll
      for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(acc[pb][r] - mx);
      const float s4 = (e[0] + e[1]) + (e[2] + e[3]);
      const float so = other32(s4);
      const float rinv = __builtin_amdgcn_rcpf(hi32 ? so + s4 : s4 + so);   // (slots 0-3 first in both lanes: the same bits)
      f32x4 av;
#pragma unroll
      for (int r = 0; r < 4; ++r) av[r] = __builtin_fmaf(e[r], rinv, epsv[r]);
      if (ATTN && sl0 < N) {
        float __attribute__((address_space(1)))* ag = (float __attribute__((address_space(1)))*)attn + (long long)sl0 * HW + pix0 + SC_TP * tix + 16 * pb + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (sl0 + r < N) ag[(long long)r * HW] = e[r] * rinv;
      }
      den4 += av;
      *(f32x4 __attribute__((address_space(3)))*)(atL + (16 * pb + i16) * SC_ATP + 4 * kb) = av;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- num[slot][channel] += A^T . X over the tile's 32 pixels: A (slot i16, pixels 8 kb .. + 7) gathered from the attention tile and split here,
    //      B (pixels 8 kb .. + 7, channel 16 cb + i16) transposed out of the row-major tile ----
    bf16x8 ath, atl;
    {
      f32x4 a0, a1;
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        a0[e2] = atL[(8 * kb + e2) * SC_ATP + i16];
        a1[e2] = atL[(8 * kb + 4 + e2) * SC_ATP + i16];
      }
      sc_split8(a0, a1, ath, atl);
    }
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
      const lchr bp = tileL + boff + ((cb << 5) ^ bswz);
      typedef bf16x4 __attribute__((address_space(3))) * lp;
      const bf16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(bp)), h1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(bp + 1024));
      const bf16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(bp + 8192)), l1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(bp + 8192 + 1024));
      const bf16x8 bh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7), bl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
      nacc[cb] = SC_MFMA(ath, bl, nacc[cb]);
      nacc[cb] = SC_MFMA(atl, bh, nacc[cb]);
      nacc[cb] = SC_MFMA(ath, bh, nacc[cb]);
    }
    __builtin_amdgcn_wave_barrier();   // the tile and the attention tile are rewritten by the next tile
  }
  SCTS();
  // ---- den4[r]: slot sl0 + r over the pixels of this lane; over the wave = over the 16 lanes of the row ----
#pragma unroll
  for (int r = 0; r < 4; ++r) den4[r] = sf_sum16(den4[r]);
  // ---- the eight waves' sums meet over the dead tiles: nacc[cb][r] = num[slot sl0 + r][channel 16 cb + i16] ----
  float* red = (float*)smem;                                   // [NW waves][8 slots][REDP]
  float* redd = red + NW * 8 * SC_REDP;                        // [NW waves][8 slots]
  __syncthreads();
  if (sl0 < 8) {   // (16-lane rows 0 and 2: slots 0-3 and 4-7)
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 8 + sl0 + r) * SC_REDP + 16 * cb + i16] = nacc[cb][r];
  }
  if (sl0 < 8 && i16 == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) redd[wave * 8 + sl0 + r] = den4[r];
  }
  __syncthreads();
  for (int idx = t; idx < N * SC_D; idx += 64 * NW) {
    const int n = idx >> 7, d = idx & 127;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[(w * 8 + n) * SC_REDP + d];
    if constexpr (GIO) {
      ((float __attribute__((address_space(1)))*)pnum)[idx] = s;
      ((float __attribute__((address_space(1)))*)pnum)[zero_off_num + idx] = 0.f;
    } else {
      pnL[idx] = s;
    }
  }
  if (t < N) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += redd[w * 8 + t];
    if constexpr (GIO) {
      ((float __attribute__((address_space(1)))*)pden)[t] = s;
      ((float __attribute__((address_space(1)))*)pden)[zero_off_den + t] = 0.f;
    } else {
      pdL[t] = s;
    }
  }
}

// the chain's pass: the whole frame by the workgroup's eight waves, queries and record in LDS.  (A non-inlined function: see below; it receives even uniform
// arguments in vector registers -- back into scalar registers.)
template <bool ATTN>
__device__ __attribute__((noinline)) void sc_attend(const __bf16* __restrict__ frame, const float* qg, float scale, float eps, int HW, int N, float* pnum,
                                                    float* pden, float* __restrict__ attn, char* smem, int rev) {
  HW = __builtin_amdgcn_readfirstlane(HW);
  N = __builtin_amdgcn_readfirstlane(N);
  rev = __builtin_amdgcn_readfirstlane(rev);
  sc_attend_core<ATTN, SC_NW, false>(frame, qg, scale, eps, HW, N, 0, HW / SC_NW, pnum, pden, 0, 0, attn, smem, rev);
}

// The two phases are separate (non-inlined) functions: inlined into one kernel body the attention pass (two staged tiles: ~250 registers) and the update
// (250) spilled 400 registers.  Inside them nothing is a generic pointer: the attention pass casts its operands to their address spaces, the update reads
// its arguments from LDS copies (UmArgs once per launch, UmVar per call), requests its weights and rows with global loads, and recovers the LDS
// address space of its planes by a cast (um_rows).
typedef __attribute__((address_space(3))) UmArgs LUmArgs;
typedef __attribute__((address_space(3))) UmVar LUmVar;
template <bool NEXT>
__device__ __attribute__((noinline)) void sc_update(const LUmArgs* ap, const LUmVar* vp, float* lds) {
  um_rows<NEXT, LUmArgs, LUmVar>(*ap, *vp, lds, 0);
}

// this call's rows -> the LDS copy (one word per thread)
__device__ __forceinline__ void sc_put_var(LUmVar* lv, const UmVar& um) {
  static_assert(sizeof(UmVar) % 4 == 0 && sizeof(UmVar) / 4 <= 64, "UmVar: one word per lane of a wave");
  const unsigned* srcw = (const unsigned*)&um;
  unsigned __attribute__((address_space(3)))* dstw = (unsigned __attribute__((address_space(3)))*)lv;
  unsigned w = 0;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(UmVar) / 4); ++i)
    if ((int)threadIdx.x == i) w = srcw[i];
  if (threadIdx.x < sizeof(UmVar) / 4) dstw[threadIdx.x] = w;
}

struct ScArgs {
  const __bf16* feat;       // [NB][T][B][HW][256] bf16: hi 128 | lo 128 per pixel (frame (t, b) of batch h at (((h * T + t) * B + b) * HW) rows)
  int B, T, HW, N, iters;   // B: videos per BATCH (the launch has one workgroup per video of NB batches)
  long long post_bs;        // floats between two videos' rows of `post`
  float scale, eps;
  float *slotsA, *slotsB, *lat;   // [B * N][128] each; slotsA holds the sampled slots of step 0 on entry
  float* q;                       // [B * N][128]: project_q of them on entry
  float* post;                    // video v, step t at post + v * post_bs + t * N * 128
  float* attn;                    // NULL or [B][T][N][HW]: the last iteration's attention of every step
  const float* noise;             // NULL or [B][T][N][128]
  float* kdist;                   // NULL or [B][T][N][256]
  UmArgs um;                      // weights / vectors of the update and of the NEXT-step prologue (row pointers filled per video)
};

}  // namespace

__global__ __launch_bounds__(SC_NT) void slot_chain_kernel(ScArgs A) {
  extern __shared__ __attribute__((aligned(16))) float sc_lds[];
  char* smem = (char*)sc_lds;
  const int v = blockIdx.x;
  const int N = A.N, T = A.T;
  const long long ro = (long long)v * N * SC_D;
  float* sA = A.slotsA + ro;
  float* sB = A.slotsB + ro;
  float* sL = A.lat + ro;
  float* pn = (float*)(smem + SC_REC);          // record: num [N][128] ...
  float* pd = pn + 8 * SC_D;                    // ... den [N]
  float* qv = pd + 32;                          // queries [N][128]
  for (int i = threadIdx.x; i < N * SC_D; i += SC_NT) qv[i] = A.q[ro + i];   // project_q of step 0's sampled slots (sf_slot_prologue_ex)
  LUmArgs* la = (LUmArgs*)(smem + SC_ARGS);
  LUmVar* lv = (LUmVar*)(smem + SC_ARGS + ((sizeof(UmArgs) + 15) / 16) * 16);
  {
    const unsigned* srcw = (const unsigned*)&A.um;
    unsigned __attribute__((address_space(3)))* dstw = (unsigned __attribute__((address_space(3)))*)la;
    for (int i = threadIdx.x; i < (int)(sizeof(UmArgs) / 4); i += SC_NT) dstw[i] = srcw[i];
  }
  UmVar um;
  um.part_num = pn; um.part_den = pd; um.P = 1; um.pstep = 1; um.R = N;
  um.noise = nullptr; um.kdist_out = nullptr; um.nx_slots = nullptr;
  __syncthreads();
#pragma unroll 1
  for (int t = 0; t < T; ++t) {
    const __bf16* frame = A.feat + (((long long)(v / A.B) * T + t) * A.B + (v % A.B)) * A.HW * 256;
    float* s_in = sA;
    float* s_out = sB;
#pragma unroll 1
    for (int it = 0; it < A.iters; ++it) {
      const bool last_it = it == A.iters - 1;
      float* arow = (A.attn && last_it) ? A.attn + (((long long)v * T + t) * N) * A.HW : nullptr;
      if (arow)
        sc_attend<true>(frame, qv, A.scale, A.eps, A.HW, N, pn, pd, arow, smem, it & 1);
      else
        sc_attend<false>(frame, qv, A.scale, A.eps, A.HW, N, pn, pd, arow, smem, it & 1);
      __syncthreads();   // the record is complete (LDS); every wave is done with the reduction rows
      SCTS();
      um.slots_prev = s_in;
      float* post_t = A.post + (long long)v * A.post_bs + (long long)t * N * SC_D;
      if (last_it && t + 1 < T) {
        // the step's last update + the prologue of step t + 1 (predictor, kernel distribution, sampling, first q): um_body<true>
        um.slots_out = (s_out == sA) ? sL : s_out;
        um.out2 = post_t;
        um.q_out = qv;
        um.nx_slots = sA;
        um.noise = A.noise ? A.noise + (((long long)v * T + t + 1) * N) * SC_D : nullptr;
        um.kdist_out = A.kdist ? A.kdist + (((long long)v * T + t + 1) * N) * 2 * SC_D : nullptr;
        sc_put_var(lv, um);
        __syncthreads();
        sc_update<true>(la, lv, sc_lds);
      } else {
        um.slots_out = s_out;
        um.out2 = last_it ? post_t : nullptr;
        um.q_out = last_it ? nullptr : qv;
        um.noise = nullptr;
        um.kdist_out = nullptr;
        um.nx_slots = nullptr;
        sc_put_var(lv, um);
        __syncthreads();
        sc_update<false>(la, lv, sc_lds);
      }
      __syncthreads();   // the update's rows / q are in memory, its LDS is free
      SCTS();
      float* tmp = s_in;
      s_in = s_out;
      s_out = tmp;
    }
  }
}

// ---- the same pass as a batch-wide launch (round 6): one Slot-Attention iteration of B frames on feature rows kept as bf16 hi | lo ----
// sa_attn_tile_kernel (slot_attn.hip) runs logits and weighted sums as exact-f32 `16x16x4` MFMAs (256 flop per clock and CU, half of them on the zero rows that
// pad 8 slots to 16): on a 128-CU partition it is bound by them (31 us for 32 frames against 21.5 on the whole chip).  Here the products are split-bf16
// `16x16x32` MFMAs (a third of the pipe time) and the rows need no split inside the loop.  A workgroup of four waves owns 512 pixels of a frame (74 KB of LDS:
// two workgroups per CU) and writes ONE record; the record behind it is zeroed (the slot update reads every second record: the launch replaces the tile
// kernel without touching the update).
constexpr int SP_NW = 4, SP_PIX = 512;
constexpr size_t SP_LDS = (size_t)SP_NW * (SC_TILE + SC_AT);
template <bool ATTN>
__global__ __launch_bounds__(64 * SP_NW, 2) void sa_attn_planes_kernel(const __bf16* __restrict__ planes, long long batch_stride_rows, const float* __restrict__ q,
                                                                       float scale, float eps, float* __restrict__ part_num, float* __restrict__ part_den,
                                                                       float* __restrict__ attn, long long attn_bs, int HW, int N, int P) {
  extern __shared__ __attribute__((aligned(16))) float sp_lds[];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const __bf16* frame = planes + (long long)b * batch_stride_rows * 256;
  float* pn = part_num + (((long long)b * P + 2 * chunk) * N) * SC_D;
  float* pd = part_den + ((long long)b * P + 2 * chunk) * N;
  sc_attend_core<ATTN, SP_NW, true>(frame, q + (long long)b * N * SC_D, scale, eps, HW, N, chunk * SP_PIX, SP_PIX / SP_NW, pn, pd, (long long)N * SC_D, N,
                                    ATTN ? attn + (long long)b * attn_bs : nullptr, (char*)sp_lds, 0);
}

bool sf_slot_attn_planes_ok(int HW, int D, int N) { return D == SC_D && HW % SP_PIX == 0 && N >= 1 && N <= 8; }

// One Slot-Attention iteration of B frames from rows of 512 B (sf_pixel_mlp_feat_planes_ex): frame b at planes + b * batch_stride_rows rows; records as
// sf_slot_attn_iter_ex writes them for keys == values at width 128 (P = HW / 256 per frame: sums in the even records, zeros in the odd ones).
int sf_slot_attn_planes_ex(const void* planes, long long batch_stride_rows, const float* q, float* part_num, float* part_den, float* attn_out,
                           long long attn_batch_stride, int B, int HW, int N, float scale, float eps, hipStream_t st) {
  SF_REQUIRE(planes && q && part_num && part_den, "sf_slot_attn_planes_ex: null pointer");
  SF_REQUIRE(sf_slot_attn_planes_ok(HW, SC_D, N) && B >= 0, "sf_slot_attn_planes_ex: slot size 128, HW a multiple of 512, at most 8 slots");
  if (B == 0) return 0;
  const int P = HW / 256;
  sf_prof_begin(SF_K_SA_ITER, st, (double)B * HW * 512.0);
  if (attn_out) {
    SF_TRY(sf_ensure_dyn_lds((const void*)sa_attn_planes_kernel<true>, SP_LDS));
    hipLaunchKernelGGL(sa_attn_planes_kernel<true>, dim3(HW / SP_PIX, B), dim3(64 * SP_NW), SP_LDS, st, (const __bf16*)planes, batch_stride_rows, q, scale, eps,
                       part_num, part_den, attn_out, attn_batch_stride, HW, N, P);
  } else {
    SF_TRY(sf_ensure_dyn_lds((const void*)sa_attn_planes_kernel<false>, SP_LDS));
    hipLaunchKernelGGL(sa_attn_planes_kernel<false>, dim3(HW / SP_PIX, B), dim3(64 * SP_NW), SP_LDS, st, (const __bf16*)planes, batch_stride_rows, q, scale, eps,
                       part_num, part_den, nullptr, 0, HW, N, P);
  }
  sf_prof_end(SF_K_SA_ITER, st);
  SF_CHECK_LAUNCH();
  return 0;
}

bool sf_slot_chain_ok(int D, int H, int HW, int N) { return D == SC_D && H == UM_H && HW >= 256 && HW % (SC_NW * SC_TP) == 0 && N >= 1 && N <= 8; }

int sf_slot_chain_ex(const void* feat_planes, int NB, int B, int T, int HW, int N, int iters, float scale, float eps, float ln_eps, float* slotsA, float* slotsB,
                     float* lat, float* q, float* post, long long post_bs, float* attn, const float* noise, float* kdist, const SfChainWeights* w,
                     hipStream_t st) {
  SF_REQUIRE(feat_planes && slotsA && slotsB && lat && q && post && w, "sf_slot_chain_ex: null pointer");
  SF_REQUIRE(sf_slot_chain_ok(SC_D, UM_H, HW, N) && NB >= 1 && B >= 1 && T >= 1 && iters >= 1 && post_bs >= (long long)T * N * SC_D, "sf_slot_chain_ex: bad shape");
  SF_REQUIRE(w->gru_ih_p && w->gru_hh_p && w->gru_b_ih && w->gru_b_hh && w->ln_g && w->ln_b && w->w1_p && w->b1 && w->w2_p && w->b2 && w->q_ln_g && w->q_ln_b &&
                 w->q_w_p && w->pm_ln_g && w->pm_ln_b && w->pm_w0_p && w->pm_b0 && w->pm_w2_p && w->pm_b2 && w->kd_w_p && w->kd_b,
             "sf_slot_chain_ex: null weight");
  ScArgs A;
  memset(&A, 0, sizeof(A));
  A.feat = (const __bf16*)feat_planes; A.B = B; A.T = T; A.HW = HW; A.N = N; A.iters = iters; A.scale = scale; A.eps = eps;
  A.slotsA = slotsA; A.slotsB = slotsB; A.lat = lat; A.q = q; A.post = post; A.post_bs = post_bs; A.attn = attn; A.noise = noise; A.kdist = kdist;
  UmArgs& a = A.um;
  a.w_ih_p = (const uint4*)w->gru_ih_p; a.w_hh_p = (const uint4*)w->gru_hh_p; a.b_ih = w->gru_b_ih; a.b_hh = w->gru_b_hh; a.ln_g = w->ln_g; a.ln_b = w->ln_b;
  a.w1_p = (const uint4*)w->w1_p; a.b1 = w->b1; a.w2_p = (const uint4*)w->w2_p; a.b2 = w->b2; a.q_ln_g = w->q_ln_g; a.q_ln_b = w->q_ln_b;
  a.q_w_p = (const uint4*)w->q_w_p; a.ln_eps = ln_eps; a.N = N; a.out2_bs = 0; a.noise_bs = 0; a.kdist_bs = 0;
  a.pm_ln_g = w->pm_ln_g; a.pm_ln_b = w->pm_ln_b; a.pm_w0_p = (const uint4*)w->pm_w0_p; a.pm_b0 = w->pm_b0; a.pm_w2_p = (const uint4*)w->pm_w2_p; a.pm_b2 = w->pm_b2;
  a.pm_norm_first = w->pm_norm_first; a.kd_w_p = (const uint4*)w->kd_w_p; a.kd_b = w->kd_b;
  SF_TRY(sf_ensure_dyn_lds((const void*)slot_chain_kernel, SC_LDS));
  // algorithmic bytes: every iteration reads its frame's feature rows once (one read of the Slot-Attention inputs per iteration)
  sf_prof_begin(SF_K_SA_ITER, st, (double)NB * B * T * iters * HW * 512.0);
  hipLaunchKernelGGL(slot_chain_kernel, dim3(NB * B), dim3(SC_NT), SC_LDS, st, A);
  sf_prof_end(SF_K_SA_ITER, st);
  SF_CHECK_LAUNCH();
  return 0;
}

#ifdef SC_STAMPS
extern "C" int sf_debug_read_ts_chain(long long* out32) {
  int zero = 0;
  hipError_t e = hipMemcpyFromSymbol(out32, HIP_SYMBOL(sc_ts), sizeof(long long) * 32);
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(sc_tsn), &zero, sizeof(int));
  return e == hipSuccess ? 0 : (int)e;
}
#endif
