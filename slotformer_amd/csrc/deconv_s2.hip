// 5x5 / stride 2 / 64->64 transposed convolution (ConvTranspose2d(k=5, stride=2, padding=2, output_padding=1): H x W -> 2H x 2W) on NHWC
// maps, split-bf16 MFMA, weights streamed as fragments -- the decoder layers of StoSAVi.decode (savi.py:252-293,504-525; the layer
// shapes come from nerv's deconv_out_shape(out_size, stride, ks // 2, ks, stride - 1), savi.py:276-277).
//
// An output pixel (2 y + py, 2 x + px) only receives the taps with ky = py (mod 2), kx = px (mod 2), read at input (y + dy, x + dx) with
// dy = (py + 2 - ky) / 2 in {-1, 0, 1}: the four OUTPUT-PARITY classes are four ordinary convolutions on the INPUT grid with 3x3, 3x2, 2x3
// and 2x2 taps (25 taps for 4 output pixels: 6.25 per pixel).  The generic path ran them as four gather GEMMs (one launch per class, every
// input pixel re-fetched through L2 per tap: 0.075 of the MFMA roof).  Here
//   * a workgroup owns 256 INPUT pixels (TRI rows x WIN columns) = 1024 output pixels x 64 channels; their (TRI + 2) x (WIN + 2) halo is
//     split into bf16 hi / lo planes in LDS once;
//   * wave = (cout block of 32, 64 of the 256 pixels) and keeps the accumulators of ALL FOUR classes (4 x 2 x 16 registers): an
//     activation fragment read at input offset (dy, dx) feeds every class that has a tap there -- 36 fragment reads of 2 x 2 x 16 bytes
//     per wave for 600 MFMAs (the 25 taps read one by one: 100);
//   * the weights never touch LDS: packed once per weight version in CONSUMPTION order (sf_pack_deconv_frag_weights: unit = (dy, dx,
//     k-step, class), split into bf16 hi / lo, MFMA A-operand layout) and pulled by each wave straight from memory into a register ring,
//     RD - 1 units in flight.  No barrier between the halo fill and the epilogue;
//   * HEAD form (the last decoder layer): ReLU + the 1x1 output convolution 64 -> 4 (rgb + mask logit, savi.py:286-289) in the epilogue --
//     the [R, 2H, 2W, 64] activation (940 MB for 32 frames of 7 slots at 128 x 128) never reaches HBM, 16 bytes per output pixel do.
// Per workgroup: 410 KB of weight fragments per wave pair + <= 101 KB of halo for 600 MFMAs per wave (4800 per workgroup).
// Products per output element: taps in (dy, dx) order, k-steps ascending, x_lo w_hi + x_hi w_lo + x_hi w_hi (the order of conv_rows4.hip).
#include "sf_internal.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int CH = 64, KS = 5, NT = 512;
constexpr int PS = CH + 8;                                   // bf16 elements per halo pixel (144 B = 9 16-B slots: conflict-free b128)
constexpr int NUNIT = 100;                                   // 25 taps x 4 k-steps
constexpr int UNIT_U4 = 2 * 2 * 64;                          // uint4 per unit: [cout block][plane][lane]
constexpr int FRAG_BYTES = NUNIT * UNIT_U4 * 16;             // 409,600
constexpr int HEAD_X_BYTES = 4 * 8 * 32 * 16;                // HEAD: exchange of the two cout blocks' partial head sums (16 KB)

template <int WIN>
struct Geo {
  static constexpr int TRI = 256 / WIN;                      // input rows per tile
  static constexpr int HR = TRI + 2, HWD = WIN + 2;          // halo
  static constexpr int HALO = HR * HWD * PS;                 // elements per plane
  static constexpr size_t LDS = (size_t)2 * HALO * sizeof(__bf16);
};

// taps of class parity p at halo offset d (0..2 = input offset -1..+1): k = p + 4 - 2 d, valid while <= 4
__host__ __device__ constexpr int tap_of(int p, int d) { return p + 4 - 2 * d; }
// step s = (dy * 3 + dx) * 4 + k-step: the units (taps at this offset) it consumes, and the units before it
__host__ __device__ constexpr int units_of(int s) { return (((s >> 2) / 3) == 0 ? 1 : 2) * (((s >> 2) % 3) == 0 ? 1 : 2); }
__host__ __device__ constexpr int units_before(int s) {
  int n = 0;
  for (int i = 0; i < s; ++i) n += units_of(i);
  return n;
}
static_assert(units_before(36) == NUNIT, "25 taps x 4 k-steps");

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
}  // namespace

// w [Cout][5][5][Cin] f32 (sf_pack_deconv_weight_f32 of the torch ConvTranspose2d weight [Cin, Cout, 5, 5]) -> consumption-ordered
// split-bf16 fragments:  uint4 index = ((unit * 2 + cb) * 2 + plane) * 64 + lane,
//   unit = running index over (dy, dx, k-step ks, py, px) with ky = py + 4 - 2 dy <= 4, kx = px + 4 - 2 dx <= 4,
//   element j = w[cb * 32 + (lane & 31)][ky][kx][ks * 16 + 8 (lane >> 5) + j]
__global__ void pack_deconv_frag_kernel(const float* __restrict__ w, uint4* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NUNIT * UNIT_U4) return;
  const int lane = idx & 63, plane = (idx >> 6) & 1, cb = (idx >> 7) & 1, unit = idx >> 8;
  int u = 0, ky = -1, kx = -1, ks = -1;
  for (int dy = 0; dy < 3; ++dy)
    for (int dx = 0; dx < 3; ++dx)
      for (int k = 0; k < 4; ++k)
        for (int py = 0; py < 2; ++py)
          for (int px = 0; px < 2; ++px) {
            if (tap_of(py, dy) > 4 || tap_of(px, dx) > 4) continue;
            if (u == unit) {
              ky = tap_of(py, dy);
              kx = tap_of(px, dx);
              ks = k;
            }
            ++u;
          }
  const float* src = w + ((long long)(cb * 32 + (lane & 31)) * (KS * KS) + ky * KS + kx) * CH + ks * 16 + 8 * (lane >> 5);
  union {
    __bf16 h[8];
    uint4 u4;
  } o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float a = src[j];
    const __bf16 ah = (__bf16)a;
    o.h[j] = plane ? (__bf16)(a - (float)ah) : ah;
  }
  out[idx] = o.u4;
}

__device__ long long dc_ts[16];   // phase timestamps of workgroup 0 (SF_DBG=deconv; sf_debug_read_ts_deconv)
#define DTS(i) do { if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dc_ts[i] = wall_clock64(); } while (0)

// in [R][H][WIN][64] f32 NHWC.  HEAD = false: out [R][2H][2WIN][64] = relu?(deconv + bias).
// HEAD = true: out = dec [R][2H * 2WIN][4] = head_w [4][64] . relu(deconv + bias) + head_b.
template <int WIN, bool HEAD>
__global__ __launch_bounds__(NT) void deconv5x5s2_kernel(const float* __restrict__ in, const uint4* __restrict__ wf,
                                                         const float* __restrict__ bias, const float* __restrict__ head_w,
                                                         const float* __restrict__ head_b, float* __restrict__ out, int H, int relu,
                                                         int dbg) {
  using G = Geo<WIN>;
  constexpr int TRI = G::TRI, HR = G::HR, HWD = G::HWD, HALO = G::HALO;
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  __bf16* Hh = lds;
  __bf16* Hl = Hh + HALO;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform: the fragment offsets below stay in scalar registers
  // XCD-aware tile order: consecutive tiles of an image share halo rows in one L2
  int bid = blockIdx.x;
  {
    const int nb = gridDim.x;
    if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  }
  const int tiles_per_img = H / TRI;
  const int r = bid / tiles_per_img, y0 = (bid - r * tiles_per_img) * TRI;
  const float* inr = in + (long long)r * H * WIN * CH;
  const int cb = wave & 1, q = wave >> 1;                    // cout block, quarter of the tile's 256 pixels

  // ---- weight ring: RD units (one unit = this wave's (hi, lo) fragment of one (tap, k-step)), RD - 1 in flight ----
  constexpr int RD = 5;   // (7 -- the deepest without spills -- measured the same: 509.7 vs 514.7 us per 256 images)
  bf16x8 ring[RD][2];
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wf), 0, 0x7fffffff, 0x00020000);
  const unsigned wbase = (unsigned)(cb * 2048);              // + unit * 4 KB + plane * 1 KB
  auto load_unit = [&](int u) {
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
      ring[u % RD][pl] = __builtin_bit_cast(
          bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, (unsigned)(lane * 16), wbase + (unsigned)(u * 4096 + pl * 1024), 0));
  };
  DTS(0);
#pragma unroll
  for (int u = 0; u < RD - 1; ++u) load_unit(u);

  // ---- halo fill: HR x HWD pixels x 16 float4, zero outside the image ----
  {
    constexpr int TOTAL = HR * HWD * (CH / 4);
    constexpr int IT = (TOTAL + NT - 1) / NT;
    f32x4 hv[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int idx = min(t + NT * i, TOTAL - 1);
      const int pix = idx >> 4, c4 = idx & 15;
      const int hy = pix / HWD, hx = pix - hy * HWD;
      const int gy = y0 - 1 + hy, gx = hx - 1;
      const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)WIN;
      const int yc = min(max(gy, 0), H - 1), xc = min(max(gx, 0), WIN - 1);
      const f32x4 v = *(const f32x4*)(inr + ((long long)yc * WIN + xc) * CH + 4 * c4);
      hv[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int idx = t + NT * i;
      if (idx < TOTAL) {
        const int off = (idx >> 4) * PS + 4 * (idx & 15);
        const bf16x4 hi = __builtin_convertvector(hv[i], bf16x4);
        const bf16x4 lo = __builtin_convertvector(hv[i] - __builtin_convertvector(hi, f32x4), bf16x4);
        *(bf16x4*)(Hh + off) = hi;
        *(bf16x4*)(Hl + off) = lo;
      }
    }
  }
  DTS(1);
  __syncthreads();
  DTS(2);

  // ---- 36 (dy, dx, k-step) steps, no barrier: acc[class][pixel block] over all 64 input channels ----
  f32x16 acc[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[c][i][e] = 0.f;
  // pixel of this lane in pixel block pb: tile pixel q * 64 + pb * 32 + (lane & 31) -> (row, col) of the tile
  int xb[2];
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    const int pt = q * 64 + pb * 32 + (lane & 31);
    xb[pb] = ((pt / WIN) * HWD + (pt % WIN)) * PS + 8 * (lane >> 5);
  }
  bf16x8 xf[2][2][2];   // [buffer][pixel block][plane]
  auto read_x = [&](int s, int buf) {
    const int ks = s & 3, dd = s >> 2, dy = dd / 3, dx = dd - 3 * dy;
    const int off = (dy * HWD + dx) * PS + ks * 16;
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      xf[buf][pb][0] = *(const bf16x8*)(Hh + xb[pb] + off);
      xf[buf][pb][1] = *(const bf16x8*)(Hl + xb[pb] + off);
    }
  };
  read_x(0, 0);
  static_for<0, 36>([&](auto S) {
    constexpr int s = decltype(S)::value;
    constexpr int dd = s >> 2, dy = dd / 3, dx = dd - 3 * dy;
    constexpr int u0 = units_before(s), nun = units_of(s);
    if constexpr (s + 1 < 36) read_x(s + 1, (s + 1) & 1);
    int k = 0;   // (a compile-time constant in every unrolled iteration)
#pragma unroll
    for (int py = 0; py < 2; ++py) {
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        if (tap_of(py, dy) > 4 || tap_of(px, dx) > 4) continue;
        const int u = u0 + k;
        if (u + RD - 1 < NUNIT) load_unit(u + RD - 1);
        const bf16x8 wh = ring[u % RD][0], wl = ring[u % RD][1];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          acc[py * 2 + px][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xf[s & 1][pb][1], acc[py * 2 + px][pb], 0, 0, 0);
          acc[py * 2 + px][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xf[s & 1][pb][0], acc[py * 2 + px][pb], 0, 0, 0);
          acc[py * 2 + px][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xf[s & 1][pb][0], acc[py * 2 + px][pb], 0, 0, 0);
        }
        ++k;
      }
    }
    // issue order inside the step: the 4 fragment reads of the NEXT step, then per unit its weight request (into the ring slot the
    // previous unit has just released) and its 6 MFMAs
    if constexpr (s + 1 < 36) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    static_for<0, nun>([&](auto K) {
      constexpr int kk = decltype(K)::value;
      if constexpr (u0 + kk + RD - 1 < NUNIT) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
    });
    __builtin_amdgcn_sched_barrier(0);   // requests stay in their step
    if (s == 11) DTS(3);
    if (s == 23) DTS(4);
  });
  DTS(5);
  if (dbg && blockIdx.x == 0 && lane == 0) dc_ts[8 + wave] = wall_clock64();

  const int Ho = 2 * H, Wo = 2 * WIN;
  const int c0 = cb * 32 + 4 * (lane >> 5);
  const float lo = relu ? 0.f : -INFINITY;
  f32x4 bv[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bv[g] = bias ? *(const f32x4*)(bias + c0 + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (!HEAD) {
    // ---- epilogue: bias, ReLU, NHWC store (a lane holds 4 x 4 consecutive channels of one output pixel per class) ----
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      const int pt = q * 64 + pb * 32 + (lane & 31);
      const int row = pt / WIN, col = pt % WIN;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int oy = 2 * (y0 + row) + (c >> 1), ox = 2 * col + (c & 1);
        const long long o = (((long long)r * Ho + oy) * Wo + ox) * CH + c0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[c][pb][4 * g + e] + bv[g][e], lo);
          *(f32x4*)(out + o + 8 * g) = v;
        }
      }
    }
  } else {
    // ---- epilogue with the 1x1 head: s[j] = sum_c head_w[j][c] relu(y[c] + b[c]); the lane's 16 channels, then the other half wave's
    //      (lane ^ 32), then the other cout block's through LDS; cout block 0 adds head_b and stores 16 bytes per output pixel ----
    float* X = (float*)(lds + 2 * HALO);   // [q][class * 2 + pb][32 pixels][4] behind the halo planes
    f32x4 hw[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) hw[j][g] = *(const f32x4*)(head_w + j * CH + c0 + 8 * g);
    f32x4 sums[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        f32x4 sj = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float y = fmaxf(acc[c][pb][4 * g + e] + bv[g][e], lo);
#pragma unroll
            for (int j = 0; j < 4; ++j) sj[j] = fmaf(y, hw[j][g][e], sj[j]);
          }
#pragma unroll
        for (int j = 0; j < 4; ++j) sj[j] += __shfl_xor(sj[j], 32, 64);
        sums[c][pb] = sj;
        if (cb == 1 && lane < 32) *(f32x4*)(X + ((q * 8 + c * 2 + pb) * 32 + lane) * 4) = sj;
      }
    __syncthreads();
    if (cb == 0) {
      const f32x4 hb = *(const f32x4*)head_b;
      const int half = lane >> 5;   // the half wave stores the px = half class: 32 columns x 2 = 1 KB contiguous per store
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const int pt = q * 64 + pb * 32 + (lane & 31);
        const int row = pt / WIN, col = pt % WIN;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          const f32x4 mine = half ? sums[py * 2 + 1][pb] : sums[py * 2][pb];
          const f32x4 oth = *(const f32x4*)(X + ((q * 8 + (py * 2 + half) * 2 + pb) * 32 + (lane & 31)) * 4);
          const int oy = 2 * (y0 + row) + py, ox = 2 * col + half;
          *(f32x4*)(out + (((long long)r * Ho + oy) * Wo + ox) * 4) = (mine + oth) + hb;
        }
      }
    }
  }
  DTS(7);
}

extern "C" size_t sf_deconv_frag_bytes(int Cout, int Cin, int ks, int stride) {
  return (Cout == CH && Cin == CH && ks == KS && stride == 2) ? (size_t)FRAG_BYTES : 0;
}

// w_ohwi [Cout][ks][ks][Cin] (sf_pack_deconv_weight_f32) -> consumption-ordered split-bf16 fragments for deconv5x5s2_kernel
extern "C" int sf_pack_deconv_frag_weights(const float* w_ohwi, void* frag, int Cout, int Cin, int ks, int stride, void* stream) {
  SF_REQUIRE(w_ohwi && frag, "sf_pack_deconv_frag_weights: null pointer");
  SF_REQUIRE(Cout == CH && Cin == CH && ks == KS && stride == 2, "sf_pack_deconv_frag_weights: needs a 64 -> 64 channel 5 x 5 stride-2 transposed convolution");
  const int total = NUNIT * UNIT_U4;
  hipLaunchKernelGGL(pack_deconv_frag_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_ohwi, (uint4*)frag);
  SF_CHECK_LAUNCH();
  return 0;
}

template <int WIN, bool HEAD>
static int launch_deconv(const float* in, const void* wf, const float* bias, const float* head_w, const float* head_b, float* out, int R,
                         int H, int relu, hipStream_t st, int dbg) {
  using G = Geo<WIN>;
  const size_t lds = G::LDS + (HEAD ? HEAD_X_BYTES : 0);
  static_assert(Geo<WIN>::LDS + HEAD_X_BYTES <= 160 * 1024, "LDS budget");
  SF_TRY(sf_ensure_dyn_lds((const void*)deconv5x5s2_kernel<WIN, HEAD>, lds));
  // (class 8 = the head-fused last layer alone: the kernel the decode roofline is quoted on; the plain layers count as convolutions)
  constexpr int cls = HEAD ? SF_K_DECONV : SF_K_CONV_NHWC;
  sf_prof_begin(cls, st, 2.0 * (double)R * H * WIN * CH * CH * KS * KS + (HEAD ? 2.0 * (double)R * 4 * H * WIN * CH * 4 : 0.0));
  hipLaunchKernelGGL((deconv5x5s2_kernel<WIN, HEAD>), dim3(R * (H / G::TRI)), dim3(NT), lds, st, in, (const uint4*)wf, bias, head_w, head_b,
                     out, H, relu, dbg);
  sf_prof_end(cls, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// Returns 1 when the kernel does not apply (the caller falls back to sf_conv_transpose2d_nhwc_f32's kernels).
// head_w != NULL: the HEAD form (W == 64 only): out = dec [R][2H * 2W][4].
int sf_deconv5x5s2_ex(const float* in, const void* w_frag, const float* bias, const float* head_w, const float* head_b, float* out, int R,
                      int H, int W, int Cin, int Cout, int ks, int stride, int relu, hipStream_t st) {
  if (!w_frag || Cin != CH || Cout != CH || ks != KS || stride != 2 || R <= 0 || sf_get_precision() != 1) return 1;
  if (W != 64 && W != 32 && W != 16) return 1;
  if (H % (256 / W) != 0) return 1;
  if (head_w && (W != 64 || !head_b)) return 1;
  static const int dbg = sf_dbg("deconv");
  if (head_w) return launch_deconv<64, true>(in, w_frag, bias, head_w, head_b, out, R, H, relu, st, dbg);
  if (W == 64) return launch_deconv<64, false>(in, w_frag, bias, nullptr, nullptr, out, R, H, relu, st, dbg);
  if (W == 32) return launch_deconv<32, false>(in, w_frag, bias, nullptr, nullptr, out, R, H, relu, st, dbg);
  return launch_deconv<16, false>(in, w_frag, bias, nullptr, nullptr, out, R, H, relu, st, dbg);
}

// C ABI: in [R][H][W][64] -> out [R][2H][2W][64] (relu: 0 / 1), W in {16, 32, 64}, H % (256 / W) == 0; split-bf16 mode only
extern "C" int sf_deconv5x5s2_frag_f32(const float* in, const void* w_frag, const float* bias, float* out, int R, int H, int W, int relu,
                                       void* stream) {
  SF_REQUIRE(in && w_frag && out, "sf_deconv5x5s2_frag_f32: null pointer");
  SF_REQUIRE(R > 0 && (W == 16 || W == 32 || W == 64) && H > 0 && H % (256 / W) == 0, "sf_deconv5x5s2_frag_f32: needs W in {16, 32, 64} and H % (256 / W) == 0");
  SF_REQUIRE(sf_get_precision() == 1, "sf_deconv5x5s2_frag_f32: split-bf16 mode only (the fragments are split-bf16)");
  return sf_deconv5x5s2_ex(in, w_frag, bias, nullptr, nullptr, out, R, H, W, CH, CH, KS, 2, relu, (hipStream_t)stream);
}

// C ABI, HEAD form: in [R][H][64][64] -> dec [R][2H * 128][4] = head_w [4][64] . relu(deconv(in) + bias) + head_b [4]
extern "C" int sf_deconv5x5s2_head_frag_f32(const float* in, const void* w_frag, const float* bias, const float* head_w,
                                            const float* head_b, float* dec, int R, int H, int W, void* stream) {
  SF_REQUIRE(in && w_frag && head_w && head_b && dec, "sf_deconv5x5s2_head_frag_f32: null pointer");
  SF_REQUIRE(R > 0 && W == 64 && H > 0 && H % 4 == 0, "sf_deconv5x5s2_head_frag_f32: needs a 64-pixel-wide input with H % 4 == 0");
  SF_REQUIRE(sf_get_precision() == 1, "sf_deconv5x5s2_head_frag_f32: split-bf16 mode only (the fragments are split-bf16)");
  return sf_deconv5x5s2_ex(in, w_frag, bias, head_w, head_b, dec, R, H, W, CH, CH, KS, 2, 1, (hipStream_t)stream);
}

extern "C" int sf_debug_read_ts_deconv(long long* out16) {
  hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(dc_ts), sizeof(long long) * 16);
  return e == hipSuccess ? 0 : (int)e;
}
