// 5x5 / 64->64 / 64-pixel-wide NHWC convolution, 4 output rows per workgroup, weights streamed as MFMA fragments (split-bf16).
//
// conv_halo.hip holds a 2-row tile and passes every tap's weights through LDS behind a workgroup barrier: 12 LDS fragment reads
// per 12 MFMAs and 25 barriers -- its waves are parked in s_waitcnt / barriers half of the time and the matrix pipe is 40 % busy
// (profiles/r03_probes.txt section 9).  Here
//   * the weights never touch LDS: they are packed once per weight version in fragment order (sf_pack_conv_frag_weights,
//     split into bf16 hi / lo) and each wave pulls ITS fragments -- the A operand -- straight from memory into a ring of four
//     taps' registers, three taps in flight.  No barrier inside the 25 taps: the eight waves run free;
//   * a workgroup owns 4 output rows (256 pixels x 64 channels); the 8 x 68 halo is split into bf16 planes in LDS once (153 KB);
//   * wave = (cout block, row pair, half of the input channels): one weight fragment feeds FOUR 32-pixel blocks -- 16 LDS
//     fragment reads per 24 MFMAs;
//   * the two cin halves meet in LDS after the last tap (each wave finishes two of its four pixel blocks).
// Per workgroup 800 KB of weight fragments + 139 KB of halo for 600 MFMAs per SIMD (19.2 k cycles).
// Same products, same order per cin half as conv_halo.hip (taps ascending, k-steps ascending, x_lo w_hi + x_hi w_lo + x_hi w_hi).
// Reference call site: the encoder convs i > 0, savi.py:231-239 (+ SoftPositionEmbed add, utils.py:60-63).
#include "sf_internal.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int CH = 64, KS = 5, PAD = 2, TW = 64, TR = 4;     // channels, taps, tile width / rows
constexpr int HR = TR + KS - 1, HWD = TW + KS - 1;           // halo 8 x 68 pixels
constexpr int PS = CH + 8;                                   // bf16 elements per pixel (144 B = 9 16-B slots: conflict-free b128)
constexpr int HALO = HR * HWD * PS;                          // elements per halo plane
constexpr int NT = 512;
constexpr int NTAP = KS * KS;
constexpr int FRAG_BYTES = NTAP * 4 * 2 * 2 * 64 * 16;      // 409,600: one uint4 per (tap, k-step, cout block, plane, lane)
constexpr int FRAG16_BYTES = FRAG_BYTES / 2;                // + the single-fp16 copy (one plane)
constexpr size_t LDS_BYTES = (size_t)2 * HALO * sizeof(__bf16);   // 156,672
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert((size_t)8 * 32 * 64 * 4 <= LDS_BYTES, "the exchange of the cin halves fits over the dead halo");
}  // namespace

// w_ohwi [64][25][64] f32 -> fragment order, split-bf16:
//   uint4 index = ((((tap * 4 + ks) * 2 + cb) * 2 + plane) * 64 + lane),  element j = w[cb*32 + (lane & 31)][tap][ks*16 + 8 (lane >> 5) + j]
__global__ void pack_conv_frag_kernel(const float* __restrict__ w, uint4* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NTAP * 4 * 2 * 2 * 64) return;
  const int lane = idx & 63, plane = (idx >> 6) & 1, cb = (idx >> 7) & 1, ks = (idx >> 8) & 3, tap = idx >> 10;
  const float* src = w + ((long long)(cb * 32 + (lane & 31)) * NTAP + tap) * CH + ks * 16 + 8 * (lane >> 5);
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
  // the single-fp16 copy of the same fragments (F16X2 form of the kernel) behind the split-bf16 planes: uint4 index FRAG_BYTES / 16 + (((tap * 4 + ks) * 2 + cb) * 64 + lane)
  if (plane == 0) {
    union {
      _Float16 h[8];
      uint4 u;
    } q;
#pragma unroll
    for (int j = 0; j < 8; ++j) q.h[j] = (_Float16)src[j];
    out[FRAG_BYTES / 16 + (((tap * 4 + ks) * 2 + cb) * 64 + lane)] = q.u;
  }
}

__device__ long long cr_ts[16];   // phase timestamps of workgroup 0 (SF_DBG=conv; sf_debug_read_ts_conv)
#define CTS(i) do { if (dbg && blockIdx.x == 0 && threadIdx.x == 0) cr_ts[i] = wall_clock64(); } while (0)

// F16X2 (opt-in, sf_set_conv_fp16x2 / SF_CONV_FP16X2=1; NOT the default arithmetic): the activations split into two fp16 terms (22 mantissa
// bits), the weights rounded to ONE fp16 (11 bits: 2^-12 relative per weight) -- two MFMAs per product instead of three and half the weight
// stream.  Measured error and speed: profiles/r03_probes.txt section 14.
// HEAD (the SAVi decoder's stride-1 last layer at 64 x 64, savi.py:262-289: a convolution with the flipped kernel + ReLU, then the 1x1 output
// convolution 64 -> 4): `add` = head_w [4][64], `head_b` [4], out = dec [F][H * 64][4] -- the 64-channel activation never reaches memory.
template <bool F16X2, bool HEAD = false>
__device__ __forceinline__ void conv5x5_rows4_body(const float* __restrict__ in, const uint4* __restrict__ wf,
                                                   const float* __restrict__ bias, const float* __restrict__ add,
                                                   float* __restrict__ out, int H, int relu, int dbg,
                                                   const float* __restrict__ head_b, __bf16* lds, const int bid0, const int nb) {
  __bf16* Hh = lds;
  __bf16* Hl = Hh + HALO;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform: the fragment offsets below stay in scalar registers
  // XCD-aware tile order: consecutive tiles of a frame share halo rows in one L2
  int bid = bid0;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  const int tiles_per_frame = H / TR;
  const int f = bid / tiles_per_frame, y0 = (bid - f * tiles_per_frame) * TR;
  const float* inf = in + (long long)f * H * TW * CH;
  const int cb = wave & 1, rp = (wave >> 1) & 1, kh = wave >> 2;   // cout block, row pair, cin half

  // ---- weight ring: slot = tap & 3 holds the tap's 2 k-steps x (hi, lo) of this wave's (cout block, cin half) ----
  constexpr int RD = 6;   // ring depth in taps: RD - 1 in flight
  bf16x8 ring[RD][2][2];   // (F16X2: [.][k][0] only, holding 8 fp16)
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wf), 0, 0x7fffffff, 0x00020000);
  // (every workgroup walks the taps in step -- all CUs of an XCD want the same 16 KB at the same time; four copies of the fragments read by
  //  alternate waves / workgroups, i.e. other L2 channels, changed nothing: profiles/r03_probes.txt section 12)
  const unsigned wbase = F16X2 ? (unsigned)(FRAG_BYTES + ((2 * kh) * 2 + cb) * 1024)            // + tap * 8 KB + ks2 * 2 KB
                               : (unsigned)((((2 * kh) * 2 + cb) * 2) * 1024);                  // + tap * 16 KB + ks2 * 4 KB + plane * 1 KB
  auto load_tap = [&](int tap) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if constexpr (F16X2) {
        ring[tap % RD][k][0] = __builtin_bit_cast(
            bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, (unsigned)(lane * 16), wbase + (unsigned)(tap * 8192 + k * 2048), 0));
      } else {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          ring[tap % RD][k][pl] = __builtin_bit_cast(
              bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, (unsigned)(lane * 16), wbase + (unsigned)(tap * 16384 + k * 4096 + pl * 1024), 0));
      }
    }
  };
  CTS(0);
#pragma unroll
  for (int q = 0; q < RD - 1; ++q) load_tap(q);

  // ---- halo fill: 8 x 68 pixels x 16 float4, zero outside the image ----
  {
    constexpr int TOTAL = HR * HWD * (CH / 4);  // 8704 float4
    constexpr int IT = (TOTAL + NT - 1) / NT;   // 17
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
      if (idx < TOTAL) {
        const int off = (idx >> 4) * PS + 4 * (idx & 15);
        if constexpr (F16X2) {
          const f16x4 hi = __builtin_convertvector(hv[i], f16x4);
          const f16x4 lo = __builtin_convertvector(hv[i] - __builtin_convertvector(hi, f32x4), f16x4);
          *(f16x4*)(Hh + off) = hi;
          *(f16x4*)(Hl + off) = lo;
        } else {
          const bf16x4 hi = __builtin_convertvector(hv[i], bf16x4);
          const bf16x4 lo = __builtin_convertvector(hv[i] - __builtin_convertvector(hi, f32x4), bf16x4);
          *(bf16x4*)(Hh + off) = hi;
          *(bf16x4*)(Hl + off) = lo;
        }
      }
    }
  }
  CTS(1);
  __syncthreads();
  CTS(2);

  // ---- 25 taps, no barrier: acc[i] = (row 2 rp + (i >> 1), pixel block i & 1) x cout block cb over this wave's cin half ----
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // the activation fragments of half-tap s + 1 are requested BEFORE the 12 MFMAs of half-tap s (two register buffers): the LDS
  // latency sits under the matrix pipe instead of in front of every MFMA pair
  const int x_lane = (lane & 31) * PS + 8 * (lane >> 5) + 32 * kh + 2 * rp * HWD * PS;
  bf16x8 xf[2][4][2];
  auto read_x = [&](int s, int buf) {
    const int tap = s >> 1, k = s & 1;
    const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = (((i >> 1) + ky) * HWD + (i & 1) * 32 + kx) * PS + k * 16;
      xf[buf][i][0] = *(const bf16x8*)(Hh + x_lane + off);
      xf[buf][i][1] = *(const bf16x8*)(Hl + x_lane + off);
    }
  };
  read_x(0, 0);
#pragma unroll
  for (int s = 0; s < 2 * NTAP; ++s) {
    const int tap = s >> 1, k = s & 1;
    if (k == 0 && tap + RD - 1 < NTAP) load_tap(tap + RD - 1);
    if (s + 1 < 2 * NTAP) read_x(s + 1, (s + 1) & 1);
    if constexpr (F16X2) {
      const f16x8 w = __builtin_bit_cast(f16x8, ring[tap % RD][k][0]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, __builtin_bit_cast(f16x8, xf[s & 1][i][1]), acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, __builtin_bit_cast(f16x8, xf[s & 1][i][0]), acc[i], 0, 0, 0);
      }
      if (k == 0 && tap + RD - 1 < NTAP) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
      if (s + 1 < 2 * NTAP) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    } else {
      const bf16x8 wh = ring[tap % RD][k][0], wl = ring[tap % RD][k][1];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xf[s & 1][i][1], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xf[s & 1][i][0], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xf[s & 1][i][0], acc[i], 0, 0, 0);
      }
      // issue order inside the half-tap: the weight requests, the 8 fragment reads of the NEXT half-tap, then the 12 MFMAs
      if (k == 0 && tap + RD - 1 < NTAP) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
      if (s + 1 < 2 * NTAP) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
    }
    __builtin_amdgcn_sched_barrier(0);   // requests stay in their half-tap
    if (s == 9) CTS(3);
    if (s == 29) CTS(4);
  }
  CTS(5);
  if (dbg && blockIdx.x == 0 && lane == 0) cr_ts[8 + wave] = wall_clock64();

  // ---- the two cin halves meet: wave kh finishes row 2 rp + kh (pixel blocks 2 kh, 2 kh + 1) and hands the other row over ----
  __syncthreads();   // every wave is done with the halo
  CTS(6);
  float* X = (float*)lds;   // [8 waves][2 blocks][16][64] f32 = 64 KB
  // (selects, not a runtime register index)
  f32x16 fin[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      X[((wave * 2 + i) * 16 + r) * 64 + lane] = kh ? acc[i][r] : acc[2 + i][r];
      fin[i][r] = kh ? acc[2 + i][r] : acc[i][r];
    }
  __syncthreads();
  const int pw = wave ^ 4;   // the partner: same cout block and row pair, the other cin half
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) fin[i][r] += X[((pw * 2 + i) * 16 + r) * 64 + lane];

  if constexpr (HEAD) {
    // ---- epilogue with the 1x1 head: s[j] = sum_c head_w[j][c] relu(y[c] + b[c]) over the lane's 16 channels, the other half wave's
    //      (lane ^ 32), then the other cout block's (wave ^ 1) through LDS; cout block 0 adds head_b and stores 16 bytes per pixel ----
    float* XH = (float*)lds + 16 * 1024;   // behind the 64 KB exchange of the cin halves: [rp * 2 + kh][i][32 pixels][4]
    const int c0h = cb * 32 + 4 * (lane >> 5);
    const int yh = y0 + 2 * rp + kh;
    f32x4 sums[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x4 sj = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = bias ? *(const f32x4*)(bias + c0h + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float yv = fmaxf(fin[i][4 * g + q] + bv[q], 0.f);
#pragma unroll
          for (int j = 0; j < 4; ++j) sj[j] = fmaf(yv, add[j * CH + c0h + 8 * g + q], sj[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) sj[j] += __shfl_xor(sj[j], 32, 64);
      sums[i] = sj;
      if (cb == 1 && lane < 32) *(f32x4*)(XH + (((rp * 2 + kh) * 2 + i) * 32 + lane) * 4) = sj;
    }
    __syncthreads();
    if (cb == 0 && lane < 32) {
      const f32x4 hb = *(const f32x4*)head_b;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f32x4 oth = *(const f32x4*)(XH + (((rp * 2 + kh) * 2 + i) * 32 + lane) * 4);
        *(f32x4*)(out + (((long long)f * H + yh) * TW + i * 32 + lane) * 4) = (sums[i] + oth) + hb;
      }
    }
    CTS(7);
    return;
  }
  // ---- epilogue: bias, ReLU, optional per-position table, NHWC store (a lane holds 4 x 4 consecutive channels of one pixel) ----
  // relu == 2 (training, backward-data pass): `add` is the forward activation of the layer below, laid out like `out`, and
  // acts as its ReLU mask -- out = add > 0 ? conv : 0
  const bool mask = relu == 2;
  const float lo = relu == 1 ? 0.f : -INFINITY;
  const int y = y0 + 2 * rp + kh;
  const int c0 = cb * 32 + 4 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int px = i * 32 + (lane & 31);
    const long long o = (((long long)f * H + y) * TW + px) * CH + c0;
    const long long ao = ((long long)(mask ? f * H + y : y) * TW + px) * CH + c0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bv = bias ? *(const f32x4*)(bias + c0 + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 av = add ? *(const f32x4*)(add + ao + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float tv = fmaxf(fin[i][4 * g + q] + bv[q], lo);
        v[q] = mask ? (av[q] > 0.f ? tv : 0.f) : tv + av[q];
      }
      *(f32x4*)(out + o + 8 * g) = v;
    }
  }
  CTS(7);
}

template <bool F16X2, bool HEAD = false>
__global__ __launch_bounds__(NT) void conv5x5_rows4_kernel(const float* __restrict__ in, const uint4* __restrict__ wf,
                                                           const float* __restrict__ bias, const float* __restrict__ add,
                                                           float* __restrict__ out, int H, int relu, int dbg,
                                                           const float* __restrict__ head_b = nullptr) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  conv5x5_rows4_body<F16X2, HEAD>(in, wf, bias, add, out, H, relu, dbg, head_b, lds, blockIdx.x, gridDim.x);
}

extern "C" size_t sf_conv_frag_bytes(int Cout, int Cin, int ks) { return (Cout == CH && Cin == CH && ks == KS) ? (size_t)FRAG_BYTES + FRAG16_BYTES : 0; }

// opt-in arithmetic of the 4-row-tile convolution: 0 (default) split-bf16, three products; 1 activations as two fp16 terms x weights as ONE fp16, two products
static int g_conv_fp16x2 = -1;
extern "C" int sf_set_conv_fp16x2(int on) {
  g_conv_fp16x2 = on ? 1 : 0;
  return 0;
}
extern "C" int sf_get_conv_fp16x2(void) {
  if (g_conv_fp16x2 < 0) {
    const char* e = getenv("SF_CONV_FP16X2");   // (a product switch: the opt-in arithmetic)
    g_conv_fp16x2 = (e && e[0] == '1') ? 1 : 0;
  }
  return g_conv_fp16x2;
}

// w_ohwi [Cout][ks][ks][Cin] (sf_pack_conv_weight_f32) -> fragment-ordered split-bf16 copy for conv5x5_rows4_kernel (64 -> 64, 5 x 5)
extern "C" int sf_pack_conv_frag_weights(const float* w_ohwi, void* frag, int Cout, int Cin, int ks, void* stream) {
  SF_REQUIRE(w_ohwi && frag, "sf_pack_conv_frag_weights: null pointer");
  SF_REQUIRE(Cout == CH && Cin == CH && ks == KS, "sf_pack_conv_frag_weights: needs a 64 -> 64 channel 5 x 5 convolution");
  const int total = NTAP * 4 * 2 * 2 * 64;
  hipLaunchKernelGGL(pack_conv_frag_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_ohwi, (uint4*)frag);
  SF_CHECK_LAUNCH();
  return 0;
}

// Returns 1 when the kernel does not apply (caller falls back to sf_conv2d_nhwc_f32's kernels).
int sf_conv5x5_rows4_ex(const float* in, const void* w_frag, const float* bias, const float* add, float* out, int F, int H, int W,
                        int Cin, int Cout, int ks, int relu, hipStream_t st) {
  if (!w_frag || W != TW || Cin != CH || Cout != CH || ks != KS || (H % TR) != 0 || F <= 0 || sf_get_precision() != 1) return 1;
  static const int dbg = sf_dbg("conv");
  // (the LDS attribute first: a failure here must not leave the class timer's pending event open)
  const bool f16 = sf_get_conv_fp16x2();
  SF_TRY(sf_ensure_dyn_lds(f16 ? (const void*)conv5x5_rows4_kernel<true> : (const void*)conv5x5_rows4_kernel<false>, LDS_BYTES));
  sf_prof_begin(SF_K_CONV_NHWC, st, 2.0 * (double)F * H * W * Cout * ks * ks * Cin);
  if (f16)
    hipLaunchKernelGGL(conv5x5_rows4_kernel<true>, dim3(F * (H / TR)), dim3(NT), LDS_BYTES, st, in, (const uint4*)w_frag, bias, add, out, H, relu, dbg);
  else
    hipLaunchKernelGGL(conv5x5_rows4_kernel<false>, dim3(F * (H / TR)), dim3(NT), LDS_BYTES, st, in, (const uint4*)w_frag, bias, add, out, H, relu, dbg);
  sf_prof_end(SF_K_CONV_NHWC, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// The decoder's stride-1 last layer + 1x1 head (HEAD form above).  Returns 1 when the kernel does not apply.
int sf_conv5x5_rows4_head_ex(const float* in, const void* w_frag, const float* bias, const float* head_w, const float* head_b, float* dec,
                             int F, int H, int W, int Cin, int Cout, int ks, hipStream_t st) {
  if (!w_frag || !head_w || !head_b || W != TW || Cin != CH || Cout != CH || ks != KS || (H % TR) != 0 || F <= 0 || sf_get_precision() != 1) return 1;
  static const int dbg = sf_dbg("conv");
  SF_TRY(sf_ensure_dyn_lds((const void*)conv5x5_rows4_kernel<false, true>, LDS_BYTES));
  sf_prof_begin(SF_K_DECONV, st, 2.0 * (double)F * H * W * Cout * ks * ks * Cin + 2.0 * (double)F * H * W * Cout * 4);
  hipLaunchKernelGGL((conv5x5_rows4_kernel<false, true>), dim3(F * (H / TR)), dim3(NT), LDS_BYTES, st, in, (const uint4*)w_frag, bias, head_w, dec, H,
                     1, dbg, head_b);
  sf_prof_end(SF_K_DECONV, st);
  SF_CHECK_LAUNCH();
  return 0;
}

extern "C" int sf_conv5x5_frag_f32(const float* in, const void* w_frag, const float* bias, const float* add, float* out, int F, int H,
                                   int W, int relu, void* stream) {
  SF_REQUIRE(in && w_frag && out, "sf_conv5x5_frag_f32: null pointer");
  SF_REQUIRE(F > 0 && W == TW && H > 0 && (H % TR) == 0, "sf_conv5x5_frag_f32: needs a 64-pixel-wide grid with H % 4 == 0");
  SF_REQUIRE(sf_get_precision() == 1, "sf_conv5x5_frag_f32: split-bf16 mode only (the fragments are split-bf16)");
  return sf_conv5x5_rows4_ex(in, w_frag, bias, add, out, F, H, W, CH, CH, KS, relu, (hipStream_t)stream);
}

extern "C" int sf_debug_read_ts_conv(long long* out16) {
  hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(cr_ts), sizeof(long long) * 16);
  return e == hipSuccess ? 0 : (int)e;
}
