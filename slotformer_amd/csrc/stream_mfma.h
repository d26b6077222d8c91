// Weight-streaming split-bf16 products for kernels that hold a few rows in LDS and pull whole weight matrices through one CU
// (pred_step.hip, rollout_small.hip): fragment-ordered packed weights (sf_pack_linear_weights) as the MFMA A operand straight from
// memory, a ring of four fragment slots per wave with three chunks in flight.
#pragma once
#include "layer_fused.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ void ps_split4(__bf16* hp, __bf16* lp, int off, f32x4 v) {
  const bf16x4 hi = __builtin_convertvector(v, bf16x4);
  const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
  *(bf16x4*)(hp + off) = hi;
  *(bf16x4*)(lp + off) = lo;
}

__device__ __forceinline__ float ps_sigmoid(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float ps_tanh(float x) { return 1.0f - 2.0f * __frcp_rn(__expf(2.0f * x) + 1.0f); }

// fragment (ks, nb, plane) of a packed [N][K] matrix (pack_linear_kernel, layer_fused.hip): 64 lanes x 16 B
// (buffer load: the descriptor and the fragment's offset are wave-uniform -- scalar registers, scalar arithmetic -- and the only
//  vector address is lane * 16; with flat 64-bit addresses per fragment the unrolled request loops spilled)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 ps_frag(const uint4* p, int nblocks, int nb, int ks, int pl, int lane) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(p), 0, 0x7fffffff, 0x00020000);
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, ((ks * nblocks + nb) * 2 + pl) * 1024, 0));
}

// the fragment ring of a wave: four slots of up to 3 k-steps; three chunks are in flight while one is consumed (with two slots of
// 4-6 k-steps -- one chunk in flight, 64 KB per CU -- the kernel ran at 40 GB/s: the stream is bound by latency x bytes in flight)
struct PsBuf {
  bf16x8 w[4][3][2];
};

template <int CH>
__device__ __forceinline__ void ps_load(PsBuf& s, int slot, const uint4* p, int nblocks, int nb, int ks0, int lane) {
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    s.w[slot][k][0] = ps_frag(p, nblocks, nb, ks0 + k, 0, lane);
    s.w[slot][k][1] = ps_frag(p, nblocks, nb, ks0 + k, 1, lane);
  }
}

// chunk size for a product over KS k-steps: a multiple of four chunks of <= 3 k-steps
template <int KS>
struct PsChunk {
  static constexpr int CH = (KS % 3 == 0) ? 3 : (KS % 8 == 0) ? 2 : 1;
  static_assert(KS % CH == 0 && ((KS / CH) % 4) == 0, "a multiple of four chunks");
};

// chunks 0..2 of a product (column block nb of the packed matrix p, from k-step ks0) into slots 0..2: what ps_block expects on entry
template <int CH>
__device__ __forceinline__ void ps_prime(PsBuf& s, const uint4* p, int nblocks, int nb, int ks0, int lane) {
#pragma unroll
  for (int c = 0; c < 3; ++c) ps_load<CH>(s, c, p, nblocks, nb, ks0 + c * CH, lane);
}

// acc[4 g + q] += out[token = lane & 31][column 32 nb + 8 g + 4 (lane >> 5) + q] over k-steps ks0 .. ks0 + KS - 1 of column block nb
// of the packed matrix p.  Chunks 0..2 must already be in slots 0..2 (ps_prime); every iteration requests the chunk three ahead --
// behind the end of this product, chunks 0..2 of the NEXT one (pn, nblocksn, nbn, ksn: chunks of CHN k-steps; pn NULL: none).
template <int KS, int CHN>
__device__ __forceinline__ void ps_block(f32x16& acc, PsBuf& s, const uint4* p, int nblocks, int nb, int ks0, const uint4* pn, int nblocksn,
                                         int nbn, int ksn, const __bf16* Xh, const __bf16* Xl, int stride, int lane) {
  constexpr int CH = PsChunk<KS>::CH, NC = KS / CH;
  const int ao = (lane & 31) * stride + 8 * (lane >> 5);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (c + 3 < NC)
      ps_load<CH>(s, (c + 3) & 3, p, nblocks, nb, ks0 + (c + 3) * CH, lane);
    else if (pn)
      ps_load<CHN>(s, (c + 3) & 3, pn, nblocksn, nbn, ksn + (c + 3 - NC) * CHN, lane);
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int ks = c * CH + k;
      const bf16x8 xh = *(const bf16x8*)(Xh + ao + ks * 16), xl = *(const bf16x8*)(Xl + ao + ks * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s.w[c & 3][k][0], xl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s.w[c & 3][k][1], xh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s.w[c & 3][k][0], xh, acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);   // the requests stay one ring slot per chunk: hoisted further they spill
  }
}



// the same for NRB token blocks (rows 32 rb .. 32 rb + 31 -> acc[rb]) on one set of fragment reads
template <int KS, int CHN, int NRB>
__device__ __forceinline__ void ps_blockN(f32x16 (&acc)[NRB], PsBuf& s, const uint4* p, int nblocks, int nb, int ks0, const uint4* pn,
                                          int nblocksn, int nbn, int ksn, const __bf16* Xh, const __bf16* Xl, int stride, int lane) {
  constexpr int CH = PsChunk<KS>::CH, NC = KS / CH;
  const int ao = (lane & 31) * stride + 8 * (lane >> 5);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (c + 3 < NC)
      ps_load<CH>(s, (c + 3) & 3, p, nblocks, nb, ks0 + (c + 3) * CH, lane);
    else if (pn)
      ps_load<CHN>(s, (c + 3) & 3, pn, nblocksn, nbn, ksn + (c + 3 - NC) * CHN, lane);
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int ks = c * CH + k;
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        const bf16x8 xh = *(const bf16x8*)(Xh + ao + rb * 32 * stride + ks * 16), xl = *(const bf16x8*)(Xl + ao + rb * 32 * stride + ks * 16);
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s.w[c & 3][k][0], xl, acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s.w[c & 3][k][1], xh, acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s.w[c & 3][k][0], xh, acc[rb], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
}  // namespace
