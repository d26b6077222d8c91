// Shared helpers for the slotformer_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define SF_WAVE 64

extern thread_local char sf_err_buf[512];

static inline int sf_set_err(int code, const char* msg, const char* file, int line) {
  snprintf(sf_err_buf, sizeof(sf_err_buf), "%s (%s:%d, code %d)", msg, file, line, code);
  return code;
}

// argument errors are negative, hipError_t values positive.
#define SF_REQUIRE(cond, msg)                                             \
  do {                                                                    \
    if (!(cond)) return sf_set_err(-1, "invalid argument: " msg, __FILE__, __LINE__); \
  } while (0)

#define SF_CHECK_LAUNCH()                                                 \
  do {                                                                    \
    hipError_t e_ = hipGetLastError();                                    \
    if (e_ != hipSuccess) return sf_set_err((int)e_, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

#define SF_TRY(expr)                \
  do {                              \
    int rc_ = (expr);               \
    if (rc_ != 0) return rc_;       \
  } while (0)

// row m of a logical [M, *] matrix -> element offset.  rows_per_batch <= 0: m * ld.
struct SfRowMap {
  long long batch_stride;
  int rows_per_batch;
  int ld;
  long long base;  // element offset added to every row
};

static inline SfRowMap sf_rows(int ld) { return SfRowMap{0, 0, ld, 0}; }
static inline SfRowMap sf_rows_batched(int ld, int rows_per_batch, long long batch_stride,
                                       long long base = 0) {
  return SfRowMap{batch_stride, rows_per_batch, ld, base};
}

__device__ __forceinline__ long long sf_row_off(const SfRowMap& m, int row) {
  if (m.rows_per_batch <= 0) return m.base + (long long)row * m.ld;
  int b = row / m.rows_per_batch;
  int r = row - b * m.rows_per_batch;
  return m.base + (long long)b * m.batch_stride + (long long)r * m.ld;
}

// 32-bit finalizer used by the dropout masks of the training path (rollout_train.hip)
__host__ __device__ static inline uint32_t sf_mix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x7feb352du;
  h ^= h >> 15;
  h *= 0x846ca68bu;
  h ^= h >> 16;
  return h;
}

// ---- DPP reductions (one v_add_f32_dpp per step instead of a ds_bpermute round trip through LDS) ----
// quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140
template <int CTRL>
__device__ __forceinline__ float sf_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// all-reduce over aligned groups of 8 / 16 lanes (every lane gets the group's result)
__device__ __forceinline__ float sf_sum8(float v) {
  v += sf_dpp<0xB1>(v);
  v += sf_dpp<0x4E>(v);
  v += sf_dpp<0x141>(v);
  return v;
}
__device__ __forceinline__ float sf_max8(float v) {
  v = fmaxf(v, sf_dpp<0xB1>(v));
  v = fmaxf(v, sf_dpp<0x4E>(v));
  v = fmaxf(v, sf_dpp<0x141>(v));
  return v;
}
__device__ __forceinline__ float sf_sum16(float v) {
  v = sf_sum8(v);
  v += sf_dpp<0x140>(v);
  return v;
}
// all-reduce over the 64 lanes: 16-lane rows by DPP, then the four row sums through scalar registers
__device__ __forceinline__ float sf_sum64(float v) {
  v = sf_sum16(v);
  const int b = __builtin_bit_cast(int, v);
  return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
         (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}

__device__ __forceinline__ float sf_max16(float v) {
  v = sf_max8(v);
  return fmaxf(v, sf_dpp<0x140>(v));
}
// all-reduce over aligned groups of W lanes (W a power of two); all lanes of the group must be active
template <int W>
__device__ __forceinline__ float sf_group_sum(float v) {
  static_assert(W >= 1 && W <= 64 && (W & (W - 1)) == 0, "group width");
  if constexpr (W >= 2) v += sf_dpp<0xB1>(v);
  if constexpr (W >= 4) v += sf_dpp<0x4E>(v);
  if constexpr (W >= 8) v += sf_dpp<0x141>(v);
  if constexpr (W >= 16) v += sf_dpp<0x140>(v);
  if constexpr (W >= 32) v += __shfl_xor(v, 16, 64);
  if constexpr (W >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}
// whole-wave all-reduces (every lane of a fully active wave gets the result)
__device__ __forceinline__ float sf_wave_sum(float v) { return sf_sum64(v); }
__device__ __forceinline__ float sf_wave_max(float v) {
  v = sf_max16(v);
  const int b = __builtin_bit_cast(int, v);
  return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))),
               fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48))));
}

__device__ __forceinline__ float sf_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
