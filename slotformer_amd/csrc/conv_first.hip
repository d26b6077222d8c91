// First encoder convolution: 5x5, Cin = 3 (NCHW image) -> 64 channels (NHWC), stride 1 or 2, output 64 pixels wide
// (savi.py:231-239, i == 0: 64x64 inputs at stride 1, 128x128 at stride 2).
//
// The generic path (gemm.hip, NCHW im2col loader) gathers every A element of the [pixels x 75] patch matrix with its
// own global load and is loader-bound (39 us on the whole chip, 132 us on the encode partition for 32 frames).  Here
// a workgroup owns 2 output rows: the 3-channel input halo (7 x 131 pixels at stride 2) is loaded once into LDS, the
// patch matrix [128 pixels x 80] is built from it in LDS as split-bf16 planes, and one 32x32 MFMA block per wave
// produces the 128 x 64 outputs (K = 75 padded to 80: 5 k16-steps x 3 MFMAs).
#include "sf_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int CF_NT = 512, CF_CO = 64, CF_CI = 3, CF_KS = 5, CF_K = CF_CI * CF_KS * CF_KS, CF_KP = 80;   // 75 -> 80
constexpr int CF_TW = 64, CF_TR = 2;
constexpr int CF_PP = CF_KP + 8;   // bf16 row pitch of the patch / weight planes: 176 B = 11 slots (odd)

template <int STRIDE>
struct CfGeo {
  static constexpr int HR = (CF_TR - 1) * STRIDE + CF_KS;       // halo rows
  static constexpr int HC = (CF_TW - 1) * STRIDE + CF_KS;       // halo columns
  static constexpr int HCP = (HC + 3) / 4 * 4 + 4;              // padded row pitch (floats)
  static constexpr size_t lds = (size_t)(CF_CI * HR * HCP + 4) * 4 + (size_t)2 * (CF_TR * CF_TW + CF_CO) * CF_PP * 2;   // (+ four zero floats behind the halo: the padding entries of the patch matrix)
};

__device__ __forceinline__ void put4(__bf16* hp, __bf16* lp, int off, f32x4 v) {
  const bf16x4 hi = __builtin_convertvector(v, bf16x4);
  const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
  *(bf16x4*)(hp + off) = hi;
  *(bf16x4*)(lp + off) = lo;
}
}  // namespace

// PERSISTENT form (round 6): a workgroup walks tiles (2 output rows of a frame) blockIdx.x, + gridDim.x, ...: the weights are split into their planes ONCE,
// the halo elements a thread fetches and the 20 patch entries it builds are located once (no index division inside the loop), and the halo of the NEXT
// tile is in flight (registers) while the current one is multiplied.  Same products in the same order as the one-tile-per-workgroup form it replaces
// (104 -> 76 us per 192 frames at 128 x 128 with the wrapper: 3.2 TB/s; the same bits).
#ifndef CF_WPS
#define CF_WPS 4   // waves per SIMD the register budget allows: 4 = two 8-wave workgroups per CU (128 registers, 34 spilled: 75.7 us per 192 frames), 2 = one (184 registers, no spill: 95.1 us)
#endif
template <int STRIDE>
__global__ __launch_bounds__(CF_NT, CF_WPS) void conv_first_kernel(const float* __restrict__ img, long long frame_stride,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           const float* __restrict__ add, float* __restrict__ out, int Ho,
                                                           int Hin, int Win, int relu, int fgroup, long long group_stride, int ntiles) {
  using G = CfGeo<STRIDE>;
  constexpr int HR = G::HR, HC = G::HC, HCP = G::HCP;
  constexpr int NH = (CF_CI * HR * HC + CF_NT - 1) / CF_NT;   // halo elements per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* In = smem;                                       // [3][HR][HCP]
  constexpr int ZERO = CF_CI * HR * HCP;                  // In[ZERO] = 0: the k >= 75 entries of the patch matrix
  __bf16* Ah = (__bf16*)(In + ZERO + 4);                  // [128][CF_PP] patch matrix
  __bf16* Al = Ah + CF_TR * CF_TW * CF_PP;
  __bf16* Wh = Al + CF_TR * CF_TW * CF_PP;                // [64][CF_PP]
  __bf16* Wl = Wh + CF_CO * CF_PP;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int tiles = Ho / CF_TR;
  if (t < 4) In[ZERO + t] = 0.f;

  // ---- weights [64][75] -> split planes, k 75..79 zero (once per workgroup) ----
  for (int idx = t; idx < CF_CO * (CF_KP / 4); idx += CF_NT) {
    const int co = idx / (CF_KP / 4), k4 = idx - co * (CF_KP / 4);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 4 * k4 + e;
      v[e] = k < CF_K ? w[co * CF_K + k] : 0.f;
    }
    put4(Wh, Wl, co * CF_PP + 4 * k4, v);
  }
  // ---- the 20 patch entries of this thread: entry (p, k) = In[c][STRIDE*row + ky][STRIDE*x + kx], k = c*25 + ky*5 + kx ----
  const int p = t & 127, kg = t >> 7;          // pixel, group of 20 k values
  unsigned poff[10];   // two 16-bit LDS offsets (floats) per word; ZERO: k >= 75
  {
    const int prow = p >> 6, x = p & 63;
#pragma unroll
    for (int q = 0; q < 10; ++q) {
      unsigned pr[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int k = kg * 20 + 2 * q + e;
        const int c = k / 25, r2 = k - c * 25, ky = r2 / 5, kx = r2 - ky * 5;
        pr[e] = k < CF_K ? (unsigned)((c * HR + STRIDE * prow + ky) * HCP + STRIDE * x + kx) : (unsigned)ZERO;
      }
      poff[q] = pr[0] | (pr[1] << 16);
    }
  }
  static_assert(CF_CI * HR * HCP < 0xffff, "16-bit patch offsets");
  auto frame_of = [&](int tile) -> const float* {
    const int f = tile / tiles;
    // frame f of the launch = frame f % fgroup of group f / fgroup (the batched encode: groups = time steps, frame_stride = T frames apart)
    return img + (long long)(f % fgroup) * frame_stride + (long long)(f / fgroup) * group_stride;
  };
  float hv[NH];
  auto fetch = [&](int tile) {
    const float* inf = frame_of(tile);
    const int y0 = (tile % tiles) * CF_TR;
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int idx = min(t + i * CF_NT, CF_CI * HR * HC - 1);   // (threads past the last element repeat it: same value to the same word)
      const int c = idx / (HR * HC), rem = idx - c * (HR * HC);
      const int r = rem / HC, j = rem - r * HC;
      const int gy = y0 * STRIDE - 2 + r, gx = j - 2;
      const bool ok = (unsigned)gy < (unsigned)Hin && (unsigned)gx < (unsigned)Win;
      const float v = inf[((long long)c * Hin + min(max(gy, 0), Hin - 1)) * Win + min(max(gx, 0), Win - 1)];
      hv[i] = ok ? v : 0.f;
    }
  };
  auto put_halo = [&]() {
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int idx = min(t + i * CF_NT, CF_CI * HR * HC - 1);
      const int c = idx / (HR * HC), rem = idx - c * (HR * HC);
      const int r = rem / HC, j = rem - r * HC;
      In[(c * HR + r) * HCP + j] = hv[i];
    }
  };
  const int row = wave >> 2, pxb = ((wave >> 1) & 1) * 32, cb = (wave & 1) * 32;
  const int ao = (row * 64 + pxb + (lane & 31)) * CF_PP + 8 * (lane >> 5);
  const int bo = (cb + (lane & 31)) * CF_PP + 8 * (lane >> 5);
  const int co = cb + (lane & 31);
  const float bv = bias ? bias[co] : 0.f;
  const float lo = relu ? 0.f : -INFINITY;
  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
#pragma unroll 1
  for (; tile < ntiles; tile += gridDim.x) {
    put_halo();        // (every wave is past the previous tile's second barrier: its patch matrix is complete, the halo tile is free)
    __syncthreads();   // the halo (and, the first time, the weight planes and the zero words) is in LDS; every wave is done with the previous patch planes
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);   // in flight while this tile is multiplied
    // ---- patch matrix ----
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = In[(poff[2 * q + (e >> 1)] >> (16 * (e & 1))) & 0xffffu];
      put4(Ah, Al, p * CF_PP + kg * 20 + 4 * q, v);
    }
    __syncthreads();   // patch planes complete
    // ---- [128 x 80] . [80 x 64]: wave = (row, 32-pixel block, 32-cout block) ----
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < CF_KP / 16; ++ks) {
      const bf16x8 xh = *(const bf16x8*)(Ah + ao + ks * 16), xl = *(const bf16x8*)(Al + ao + ks * 16);
      const bf16x8 yh = *(const bf16x8*)(Wh + bo + ks * 16), yl = *(const bf16x8*)(Wl + bo + ks * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh, acc, 0, 0, 0);
    }
    const int f = tile / tiles, y = (tile % tiles) * CF_TR + row;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int px = pxb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      float v = fmaxf(acc[r] + bv, lo);
      if (add) v += add[((long long)y * CF_TW + px) * CF_CO + co];
      out[(((long long)f * Ho + y) * CF_TW + px) * CF_CO + co] = v;
    }
  }
}

template <int STRIDE>
static int launch_cf(const float* img, long long frame_stride, const float* w, const float* bias, const float* add,
                     float* out, int F, int Ho, int Hin, int Win, int relu, int fgroup, long long group_stride, hipStream_t st) {
  auto kern = conv_first_kernel<STRIDE>;
  constexpr size_t lds = CfGeo<STRIDE>::lds;
  static_assert(lds <= 80 * 1024, "first conv: two workgroups per CU");
  SF_TRY(sf_ensure_dyn_lds((const void*)kern, (size_t)(lds)));
  sf_prof_begin(SF_K_CONV_FIRST, st, 2.0 * (double)F * Ho * CF_TW * CF_CO * CF_K);
  // one workgroup (CF_WPS 4: two) per CU of the stream, each walking its share of the tiles (19 KB of weights are split once per workgroup)
  const int ntiles = F * (Ho / CF_TR);
  int grid = (CF_WPS / 2) * sf_stream_cus((void*)st);
  if (grid > ntiles) grid = ntiles;   // (small launches: one tile per workgroup, as many workgroups as tiles)
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(CF_NT), lds, st, img, frame_stride, w, bias, add, out, Ho, Hin, Win, relu, fgroup, group_stride, ntiles);
  sf_prof_end(SF_K_CONV_FIRST, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// Returns 1 when the specialised kernel does not apply (caller uses the implicit-GEMM path).
int sf_conv_first_ex(const float* img, long long frame_stride, const float* w, const float* bias, const float* add,
                     float* out, int F, int Cin, int Hin, int Win, int Cout, int ks, int stride, int relu, hipStream_t st) {
  return sf_conv_first_grouped_ex(img, frame_stride, F > 0 ? F : 1, 0, w, bias, add, out, F, Cin, Hin, Win, Cout, ks, stride, relu, st);
}
// frames in groups: output frame i = input frame (i % fgroup) * frame_stride + (i / fgroup) * group_stride (one launch for all time steps of a batch)
int sf_conv_first_grouped_ex(const float* img, long long frame_stride, int fgroup, long long group_stride, const float* w, const float* bias,
                             const float* add, float* out, int F, int Cin, int Hin, int Win, int Cout, int ks, int stride, int relu, hipStream_t st) {
  if (Cin != CF_CI || Cout != CF_CO || ks != CF_KS || F <= 0 || fgroup <= 0) return 1;
  if (stride == 2 && Hin == 128 && Win == 128) return launch_cf<2>(img, frame_stride, w, bias, add, out, F, 64, Hin, Win, relu, fgroup, group_stride, st);
  if (stride == 1 && Hin == 64 && Win == 64) return launch_cf<1>(img, frame_stride, w, bias, add, out, F, 64, Hin, Win, relu, fgroup, group_stride, st);
  return 1;
}
