// 5x5 / 64->64 / 64-pixel-wide NHWC convolution with the WEIGHTS STATIONARY IN REGISTERS (split-bf16, the products and their order of conv_rows4.hip).
//
// conv_rows4.hip streams the 800 KB of weight fragments of a 4-row tile from L2 through every wave's registers and pays halo fill, the exchange of
// the cin halves and the epilogue in front of / behind the 25 taps of every tile: its waves issue an MFMA every 50-56 cycles against the pipe's 32
// (profiles/r04_sq_counters.txt).  The whole weight set of a layer -- 64 x 1600 values in two bf16 planes, 410 KB -- fits the register file of ONE CU
// (4 SIMDs x 128 KB).  So here
//   * a workgroup is four waves, one per SIMD, 512 registers each; wave = (cout block cb, cin half kh) keeps ITS 100 weight fragments (400 registers)
//     for the whole launch: no weight traffic inside the loop at all;
//   * a workgroup owns a contiguous range of output rows of the launch (frames x rows, any count) and walks it row by row: LDS holds a ring of six
//     halo rows (two bf16 planes, 117 KB); while the 300 MFMAs of a row run, the row after the next window is loaded, split and written to the ring and
//     the previous row's results leave -- fill, exchange and epilogue ride under the matrix pipe instead of in front of it;
//   * the two rows of zeros between two frames of the stream are shared by the frame above and the frame below (a frame change costs two exposed fills);
//   * per row and wave 2 x 50 x (2 ds_read_b128 + 3 MFMA); the two cin halves of a pixel block meet through a double-buffered 16 KB exchange: a wave
//     computes the block its partner finishes first, hands it over, keeps the accumulators of its own block and stores them one row later.
// Same arithmetic as conv_rows4.hip bit for bit: per cin half taps ascending, k-steps ascending, x_lo w_hi + x_hi w_lo + x_hi w_hi, then half 0 + half 1,
// bias, ReLU, per-position table.  Reads the fragment copy of sf_pack_conv_frag_weights unchanged.
// Reference call site: the encoder convs i > 0, savi.py:231-239 (+ SoftPositionEmbed add, utils.py:60-63).
#include "sf_internal.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

extern "C" int sf_get_conv_fp16x2(void);
namespace {
constexpr int CH = 64, KS = 5, TW = 64, NTAP = KS * KS;
constexpr int HWD = TW + KS - 1;                  // 68 pixels per halo row
constexpr int PS = CH + 8;                        // bf16 elements per pixel (144 B: conflict-free ds_read_b128)
constexpr int ROWE = HWD * PS;                    // elements per ring row and plane (9792 B)
constexpr int NR = 6;                             // ring rows: five under the taps + one being filled
constexpr int NT = 256;
constexpr int PLANE_B = NR * ROWE * 2;            // 58,752 B
constexpr int X_B = 4 * 16 * 64 * 4;              // one exchange buffer: [wave][16][64] f32 = 16 KB
constexpr int LDS_B = 2 * PLANE_B + 2 * X_B + 256;      // 150,272 B (ring + the partner exchange + the wave's own parked block)
static_assert(LDS_B <= 160 * 1024, "LDS budget");
static_assert(PLANE_B + (32 + 4) * PS * 2 + 64 < 65536, "lo-plane reads stay inside the ds immediate offset");
}  // namespace

__device__ long long cw_ts[16];   // phase time stamps of workgroup 0 (SF_DBG=conv; sf_debug_read_ts_conv_ws)
#define WTS(i) do { if (dbg && blockIdx.x == 0 && threadIdx.x == 0) cw_ts[i] = wall_clock64(); } while (0)

#define WS_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

struct CwArgs {
  const float* in;
  const uint4* wf;
  const float* bias;
  const float* add;
  float* out;
  int F, H, relu, dbg;
  long long rows;   // F * H
};

template <bool BIAS, bool ADD>
__global__ __launch_bounds__(NT) void conv5x5_ws_kernel(CwArgs A) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  __bf16* Hh = (__bf16*)lds_raw;
  float* X = (float*)(lds_raw + 2 * PLANE_B);
  float* Bs = X + 2 * (X_B / 4);   // the bias, 64 floats
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int cb = wave & 1, kh = wave >> 1;
  const int H = A.H, HP = H + 2, dbg = A.dbg;
  // this workgroup's output rows [g0, g1) of the F * H rows of the launch
  const long long g0 = A.rows * blockIdx.x / gridDim.x, g1 = A.rows * (blockIdx.x + 1) / gridDim.x;
  if (g0 >= g1) return;
  WTS(0);

  // stream of halo rows: index s = f * (H + 2) + r;  r < 2: a row of zeros (below frame f - 1 and above frame f), else row r - 2 of frame f.
  // Output row (f, y) reads stream rows s0 .. s0 + 4 with s0 = f * (H + 2) + y; ring slot of stream row s = s % NR.  All of it wave-uniform 32-bit
  // state carried from row to row (no division inside the loop).
  // stream row (f, r), r may run into the next frame: its source row (clamped into the input: always loadable) and whether it is a row of zeros --
  // selects, no branch: the row loop below is ONE basic block
  auto row_src = [&](int f, int r, bool& zero) __attribute__((always_inline)) -> const float* {
    const bool nxt = r >= HP;
    r = nxt ? r - HP : r;
    f = nxt ? f + 1 : f;
    zero = (r < 2) | (f >= A.F);
    const int fr = min(f, A.F - 1) * H + max(r - 2, 0);
    return A.in + ((long long)fr * TW) * CH;
  };
  // fill of one stream row by the whole workgroup: thread t takes float4 t + 256 i of the row's 1024 (pixel (t + 256 i) >> 4, channels 4 ((t + 256 i) & 15)),
  // in two halves (i = 0, 1 and i = 2, 3) so that a half in flight holds eight registers
  const int fill_off = ((t >> 4) + 2) * PS + 4 * (t & 15);   // + 16 i * PS elements
  auto fill_load = [&](const float* src, int half, f32x4 (&hv)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) hv[i] = *(const f32x4*)(src + 4 * (t + NT * (2 * half + i)));
  };
  auto fill_write = [&](int slot, bool zero, int half, const f32x4 (&hv)[2]) __attribute__((always_inline)) {
    __bf16* rh = Hh + slot * ROWE + fill_off + 32 * half * PS;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const f32x4 x = zero ? f32x4{0.f, 0.f, 0.f, 0.f} : hv[i];
      const bf16x4 hi = __builtin_convertvector(x, bf16x4);
      const bf16x4 lo = __builtin_convertvector(x - __builtin_convertvector(hi, f32x4), bf16x4);
      *(bf16x4*)(rh + 16 * i * PS) = hi;
      *(bf16x4*)(rh + 16 * i * PS + NR * ROWE) = lo;
    }
  };
  auto fill_sync = [&](int f, int r, int slot) __attribute__((always_inline)) {
    bool zero;
    const float* src = row_src(f, r, zero);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 hv[2];
      fill_load(src, half, hv);
      fill_write(slot, zero, half, hv);
    }
  };
  auto wrap = [&](int slot) __attribute__((always_inline)) -> int { return slot >= NR ? slot - NR : slot; };

  // B-operand address of a lane inside a ring row (bytes): pixel (lane & 31) [+ 32 pb + kx], cin 32 kh + 8 (lane >> 5) [+ 16 k]
  const unsigned lane_b = (unsigned)(((lane & 31) * PS + 32 * kh + 8 * (lane >> 5)) * 2);
  const int ob = kh;                      // the pixel block this wave finishes (its partner, wave ^ 2, finishes the other one)
  // exchange [which][wave][r4][lane][4] f32, 16-byte accesses, lane-contiguous.  which 0: the accumulators of the block the partner finishes (read by
  // the partner), which 1: those of the wave's own block (read back by itself one row later: no accumulators live across rows)
  float* Xw = X + wave * 1024 + lane * 4;
  const float* Xp = X + (wave ^ 2) * 1024 + lane * 4;
  float* Xo = X + 4096 + wave * 1024 + lane * 4;
  const int c0 = cb * 32 + 4 * (lane >> 5);
  const float lo_clip = A.relu == 1 ? 0.f : -INFINITY;

  // the first window
  int f = (int)(g0 / H), y = (int)(g0 - (long long)f * H);
  int slot0 = 0;   // ring slot of stream row s0 (the stream is renumbered from this workgroup's first row)
  {
    // (all five rows requested before the first is converted: one memory latency, not five)
    f32x4 hw[5][2][2];
    bool zw[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const float* src = row_src(f, y + j, zw[j]);
      fill_load(src, 0, hw[j][0]);
      fill_load(src, 1, hw[j][1]);
    }
    // ring to zero first (the two pad columns at both ends of a row stay zero for the whole launch)
    for (int i = t; i < 2 * PLANE_B / 16; i += NT) ((uint4*)lds_raw)[i] = uint4{0u, 0u, 0u, 0u};
    if (t < CH) Bs[t] = BIAS ? A.bias[t] : 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      fill_write(j, zw[j], 0, hw[j][0]);
      fill_write(j, zw[j], 1, hw[j][1]);
    }
  }
  __syncthreads();
  WTS(1);
  // ---- the wave's weights: 25 taps x 2 k-steps x (hi, lo) of (cout block cb, cin half kh); requested behind the first window so that the fill's
  //      registers are free again (the compiler spilled half of the fragments around it otherwise) ----
  bf16x8 wt[NTAP][2][2];
  {
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(A.wf), 0, 0x7fffffff, 0x00020000);
    const unsigned wbase = (unsigned)((((2 * kh) * 2 + cb) * 2) * 1024);   // + tap * 16 KB + k * 4 KB + plane * 1 KB
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap)
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          wt[tap][k][pl] = __builtin_bit_cast(
              bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, (unsigned)(lane * 16), wbase + (unsigned)(tap * 16384 + k * 4096 + pl * 1024), 0));
  }

  // (the "previous row" of the first row is that row itself: its epilogue stores garbage where the row's own epilogue, one row later, stores the
  //  results -- same lanes, same addresses, program order; no branch in the loop)
  int fp = f, yp = y;

  // one pixel block of one output row: 50 steps of (2 fragment reads, 3 MFMAs); side(s) is issued in front of step s's MFMAs
#ifndef WS_PF
#define WS_PF 2
#endif
  auto block = [&](const unsigned (&ra)[5], int pb, f32x16& acc, auto&& side) __attribute__((always_inline)) {
    const unsigned pbo = (unsigned)(pb * 32 * PS * 2);
    constexpr int PF = WS_PF, NB = PF + 1;   // fragment reads issued PF steps ahead
    bf16x8 xh[NB], xl[NB];
    auto rd = [&](int s, int buf) __attribute__((always_inline)) {
      const int tap = s >> 1, k = s & 1, ky = tap / KS, kx = tap - ky * KS;
      const char* p = lds_raw + (ra[ky] + pbo);
      xl[buf] = *(const bf16x8*)(p + (kx * PS + 16 * k) * 2 + PLANE_B);
      xh[buf] = *(const bf16x8*)(p + (kx * PS + 16 * k) * 2);
    };
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < PF; ++s) rd(s, s % NB);
#pragma unroll
    for (int s = 0; s < 2 * NTAP; ++s) {
      const int tap = s >> 1, k = s & 1;
      if (s + PF < 2 * NTAP) rd(s + PF, (s + PF) % NB);
#ifndef WS_NOSB
      __builtin_amdgcn_sched_barrier(0);
#endif
      side(s);
      acc = WS_MFMA(wt[tap][k][0], xl[s % NB], acc);
      acc = WS_MFMA(wt[tap][k][1], xh[s % NB], acc);
      acc = WS_MFMA(wt[tap][k][0], xh[s % NB], acc);
#ifndef WS_NOSB
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  };

  // the epilogue of the previous row's own block: own half + the partner's, bias, ReLU, per-position table, NHWC store -- four pieces of four channels,
  // each in two stages eight steps apart: the reads (exchange, bias, table) are requested, and consumed once they have landed behind the fragment reads
  // of the steps in between (a wave that waits for them at once drains its LDS queue in front of every piece: 1400-3300 cycles per row)
  f32x4 e_oth, e_own, e_bv, e_av = {0.f, 0.f, 0.f, 0.f};
  auto epilogue_request = [&](int g, int yq) __attribute__((always_inline)) {
    const int px = ob * 32 + (lane & 31);
    e_oth = *(const f32x4*)(Xp + g * 256);
    e_own = *(const f32x4*)(Xo + g * 256);
    e_bv = *(const f32x4*)(Bs + c0 + 8 * g);
    if constexpr (ADD) e_av = *(const f32x4*)(A.add + ((long long)yq * TW + px) * CH + c0 + 8 * g);
  };
  auto epilogue_store = [&](int g, int fq, int yq) __attribute__((always_inline)) {
    const int px = ob * 32 + (lane & 31);
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q)   // (half 0 + half 1; float addition commutes, so both waves of a pair produce the bits of conv_rows4.hip)
      v[q] = fmaxf((e_own[q] + e_oth[q]) + e_bv[q], lo_clip) + e_av[q];
    *(f32x4*)(A.out + (((long long)(fq * H + yq) * TW) + px) * CH + c0 + 8 * g) = v;
  };
  auto put_acc = [&](float* dst, const f32x16& acc) __attribute__((always_inline)) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) *(f32x4*)(dst + r4 * 256) = f32x4{acc[4 * r4], acc[4 * r4 + 1], acc[4 * r4 + 2], acc[4 * r4 + 3]};
  };

  const int nrows = (int)(g1 - g0);
#ifdef WS_STAMPS
  long long c_a = 0, c_b = 0, c_c = 0, c_d = 0;
#define WS_CLK() __builtin_readcyclecounter()
#endif
#pragma unroll 1
  for (int it = 0; it < nrows; ++it) {
#ifdef WS_STAMPS
    const long long k0 = WS_CLK();
#endif
    const bool frame_change = (y == H - 1);
#ifndef WS_PIN
#define WS_PIN 60
#endif
#if WS_PIN > 0
    // the register class of every weight fragment, stated once per row: fragments 0 .. WS_PIN - 1 live in AGPRs, the rest in VGPRs (the allocator
    // otherwise moves them between the classes and spills three to six of them inside the loop)
#pragma unroll
    for (int i = 0; i < NTAP * 4; ++i) {
      if (i < WS_PIN) asm volatile("" : "+a"(wt[i >> 2][(i >> 1) & 1][i & 1]));
      else asm volatile("" : "+v"(wt[i >> 2][(i >> 1) & 1][i & 1]));
    }
#endif
    unsigned ra[5];
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) ra[ky] = (unsigned)(wrap(slot0 + ky) * ROWE * 2) + lane_b;
    // the row behind the window (stream row s0 + 5) goes into the slot row it - 1 released (behind the workgroup's last row too: a loadable row nobody reads)
    bool fzero;
    const float* fsrc = row_src(f, y + 5, fzero);
    const int fslot = wrap(slot0 + 5);
    const int fq = fp, yq = yp;
    f32x4 hv[2];
    f32x16 acc;
    // ---- the block the partner finishes; beside it the previous row leaves and the first half of the fill arrives ----
    block(ra, 1 - ob, acc, [&](int s) __attribute__((always_inline)) {
#ifdef WS_NOSIDE
      return;
#endif
      if (s == 0) fill_load(fsrc, 0, hv);
      if (s >= 2 && s < 42 && (s - 2) % 10 == 0) epilogue_request((s - 2) / 10, yq);
      if (s >= 10 && s < 50 && s % 10 == 0) epilogue_store(s / 10 - 1, fq, yq);
      if (s == 44) fill_write(fslot, fzero, 0, hv);
    });
#ifdef WS_STAMPS
    const long long k1 = WS_CLK();
#endif
#ifndef WS_NOBAR
    __syncthreads();   // every wave has read the exchange of the previous row
#endif
#ifdef WS_STAMPS
    const long long k2 = WS_CLK();
#endif
    put_acc(Xw, acc);
    // ---- its own block; the second half of the fill ----
    block(ra, ob, acc, [&](int s) __attribute__((always_inline)) {
#ifdef WS_NOSIDE
      return;
#endif
      if (s == 0) fill_load(fsrc, 1, hv);
      if (s == 40) fill_write(fslot, fzero, 1, hv);
    });
    put_acc(Xo, acc);
    fp = f;
    yp = y;
#ifdef WS_STAMPS
    const long long k3 = WS_CLK();
#endif
#ifndef WS_NOBAR
    __syncthreads();
#endif
#ifdef WS_STAMPS
    const long long k4 = WS_CLK();
    c_a += k1 - k0; c_b += k2 - k1; c_c += k3 - k2; c_d += k4 - k3;
#endif
    if (frame_change && it + 1 < nrows) {   // the next window starts three stream rows further on, two of them not yet in the ring
      fill_sync(f, y + 6, slot0);           // (slot0 + 6) % NR: the slots of stream rows s0, s0 + 1 are free behind the barrier
      fill_sync(f, y + 7, wrap(slot0 + 1));
      __syncthreads();
    }
    if (frame_change) { y = 0; ++f; slot0 = wrap(slot0 + 3); } else { ++y; slot0 = wrap(slot0 + 1); }
  }
  WTS(2);
#ifdef WS_STAMPS
  if (blockIdx.x == 0 && lane == 0) { cw_ts[4 + wave] = c_a; cw_ts[8 + wave] = c_c; if (wave == 0) { cw_ts[12] = c_b; cw_ts[13] = c_d; cw_ts[14] = nrows; } }
#endif
  // the last row's own block
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    epilogue_request(g, yp);
    epilogue_store(g, fp, yp);
  }
  WTS(3);
}

// Returns 1 when the kernel does not apply (the caller falls back to the 4-row-tile kernel).
int sf_conv5x5_ws_ex(const float* in, const void* w_frag, const float* bias, const float* add, float* out, int F, int H, int W, int Cin, int Cout, int ks,
                     int relu, int n_workgroups, hipStream_t st) {
  if (!w_frag || W != TW || Cin != CH || Cout != CH || ks != KS || H < 1 || F <= 0 || relu < 0 || relu > 1 || sf_get_precision() != 1) return 1;
  // (the opt-in two-product fp16 arithmetic lives in the 4-row-tile kernel only: with it on, every stream takes that kernel -- one arithmetic per process)
  if (sf_get_conv_fp16x2()) return 1;
  static const int dbg = sf_dbg("conv");
  const long long rows = (long long)F * H;
  int nwg = n_workgroups;
  if (nwg <= 0) {
    // one workgroup per CU of the stream; a launch too small to give each of them four rows does not pay for 410 KB of weights per workgroup
    static const bool on = []() { const char* e = getenv("SF_CONV_WS"); return !(e && e[0] == '0'); }();
    // ... of a stream whose CUs are its own: one made by sf_stream_create_cu_mask, or one that captures for such a stream (sf_stream_set_cus).  On any
    // other stream the launch shares the chip with whatever else runs, and a workgroup that holds its CU for the whole launch (0.5 ms at 192 frames)
    // keeps the latency-bound rollout launches and the masked encode lane waiting: those keep the 4-row tiles (27 us per workgroup) -- measured in
    // the pipeline: hybrid-lane encodes with persistent workgroups cost 6 % at 60 batches
    nwg = sf_stream_cus_known((void*)st);
    if (!on || nwg <= 0 || rows < 4LL * nwg) return 1;
  }
  if (nwg > rows) nwg = (int)rows;
  const void* kfn = bias ? (add ? (const void*)conv5x5_ws_kernel<true, true> : (const void*)conv5x5_ws_kernel<true, false>)
                         : (add ? (const void*)conv5x5_ws_kernel<false, true> : (const void*)conv5x5_ws_kernel<false, false>);
  SF_TRY(sf_ensure_dyn_lds(kfn, LDS_B));
  CwArgs a{in, (const uint4*)w_frag, bias, add, out, F, H, relu, dbg, rows};
  sf_prof_begin(SF_K_CONV_NHWC, st, 2.0 * (double)F * H * W * Cout * ks * ks * Cin);
  if (bias && add) hipLaunchKernelGGL((conv5x5_ws_kernel<true, true>), dim3(nwg), dim3(NT), LDS_B, st, a);
  else if (bias) hipLaunchKernelGGL((conv5x5_ws_kernel<true, false>), dim3(nwg), dim3(NT), LDS_B, st, a);
  else if (add) hipLaunchKernelGGL((conv5x5_ws_kernel<false, true>), dim3(nwg), dim3(NT), LDS_B, st, a);
  else hipLaunchKernelGGL((conv5x5_ws_kernel<false, false>), dim3(nwg), dim3(NT), LDS_B, st, a);
  sf_prof_end(SF_K_CONV_NHWC, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// The weights-stationary kernel by itself (tests, tools): n_workgroups 0 = one per CU of the stream, else that many (any split of the F * H rows).
extern "C" int sf_conv5x5_ws_f32(const float* in, const void* w_frag, const float* bias, const float* add, float* out, int F, int H, int W, int relu,
                                 int n_workgroups, void* stream) {
  SF_REQUIRE(in && w_frag && out, "sf_conv5x5_ws_f32: null pointer");
  SF_REQUIRE(F > 0 && W == TW && H > 0 && (relu == 0 || relu == 1) && n_workgroups >= 0, "sf_conv5x5_ws_f32: needs a 64-pixel-wide grid, relu 0 / 1");
  SF_REQUIRE(sf_get_precision() == 1, "sf_conv5x5_ws_f32: split-bf16 mode only (the fragments are split-bf16)");
  const int rc = sf_conv5x5_ws_ex(in, w_frag, bias, add, out, F, H, W, CH, CH, KS, relu, n_workgroups > 0 ? n_workgroups : sf_stream_cus(stream),
                                  (hipStream_t)stream);
  return rc == 1 ? sf_set_err(-1, "sf_conv5x5_ws_f32: the kernel does not apply", __FILE__, __LINE__) : rc;
}

extern "C" int sf_debug_read_ts_conv_ws(long long* out16) {
  hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(cw_ts), sizeof(long long) * 16);
  return e == hipSuccess ? 0 : (int)e;
}
