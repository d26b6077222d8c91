// LDS tiles as two bf16 PLANES (hi, lo) for split-bf16 32x32x16 MFMA products (slate_attn_bwd.hip, steve_decoder.hip): an f32 element lives as
// bf16(x) in the hi plane and bf16(x - hi) in the lo plane of a row-major tile whose row pitch is a multiple of 16 bytes.  An operand fragment
// (lane (i, kk): eight consecutive contraction indices of row / column i) is
//   * one ds_read_b128 per plane when the contraction index runs along the tile's rows (pl_rd), or
//   * two ds_read_b64_tr_b16 per plane when it runs DOWN the tile's rows (pl_rd_tr: lane l = (j = l & 15, g = l >> 4) of a 16-lane group points at
//     four contiguous elements of row k0 + 8 kk + (j >> 2), columns col0 + 16 (g & 1) + 4 (j & 3) .. + 3, and receives column col0 + (l & 31) of four
//     consecutive rows; the second read starts `second` rows further down: 4 for the natural k order 8 kk .. 8 kk + 7).
#pragma once
#include <hip/hip_runtime.h>

typedef __bf16 pl_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pl_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pl_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pl_f32x2 __attribute__((ext_vector_type(2)));
typedef float pl_f32x16 __attribute__((ext_vector_type(16)));
#define PL_LDS(T, p) ((T __attribute__((address_space(3)))*)(p))

struct PlFrag {
  pl_bf16x8 h, l;
};
// two f32 -> packed bf16 hi pair and lo pair
__device__ __forceinline__ void pl_split2(float a, float b, unsigned& hi, unsigned& lo) {
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector((pl_f32x2{a, b}), pl_bf16x2));
  const float fa = __builtin_bit_cast(float, hi << 16), fb = __builtin_bit_cast(float, hi & 0xffff0000u);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector((pl_f32x2{a - fa, b - fb}), pl_bf16x2));
}
__device__ __forceinline__ PlFrag pl_rd(const char* lds, int hoff, int loff, int pitch, int row0, int k0, int lane) {
  const int o = (row0 + (lane & 31)) * pitch + (k0 + 8 * (lane >> 5)) * 2;
  PlFrag f;
  f.h = *(const pl_bf16x8*)(lds + hoff + o);
  f.l = *(const pl_bf16x8*)(lds + loff + o);
  return f;
}
// kstep: rows per kk half (8 for the natural order: lane half kk holds rows k0 + 8 kk ..; 4 for the accumulator order of a 32x32 block, where half kk
// holds rows k0 + 4 kk .. + 3 and k0 + 8 + 4 kk .. + 3);  second: distance of the second group of four rows (4 / 8)
__device__ __forceinline__ PlFrag pl_rd_tr(const char* lds, int hoff, int loff, int pitch, int k0, int col0, int lane, int kstep = 8, int second = 4) {
  const int j = lane & 15, g = lane >> 4, kk = lane >> 5;
  const int o = (k0 + kstep * kk + (j >> 2)) * pitch + (col0 + 16 * (g & 1) + 4 * (j & 3)) * 2;
  PlFrag f;
  const pl_bf16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(PL_LDS(pl_bf16x4, lds + hoff + o));
  const pl_bf16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(PL_LDS(pl_bf16x4, lds + hoff + o + second * pitch));
  const pl_bf16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(PL_LDS(pl_bf16x4, lds + loff + o));
  const pl_bf16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(PL_LDS(pl_bf16x4, lds + loff + o + second * pitch));
  f.h = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
  f.l = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
  return f;
}
// acc += a . b in split-bf16: hi.lo + lo.hi + hi.hi
__device__ __forceinline__ void pl_mma(pl_f32x16& acc, const PlFrag& a, const PlFrag& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
}
