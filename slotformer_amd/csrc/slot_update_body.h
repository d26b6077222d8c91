// Body of the matrix-core slot update (slot_update_mfma.hip holds the description and the launch wrapper): shared between the stand-alone kernel and
// the heterogeneous convolution + slot-update launch of conv_rows4.hip.
#pragma once
#include "sf_internal.h"


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifdef UM_STAMPS
__device__ long long um_ts[16];
#define UMTS(i) do { if (block == 0 && threadIdx.x == 0) um_ts[i] = wall_clock64(); } while (0)
#else
#define UMTS(i) do { } while (0)
#endif

namespace {

constexpr int UM_D = 128, UM_H = 256, UM_ROWS = 32, UM_NT = 512;
constexpr int UM_DP = UM_D + 8, UM_HP = UM_H + 8;   // bf16 plane strides (rows 272 / 528 B apart: 16-B aligned, bank-spread)
constexpr int UM_FP = UM_D + 4;                     // f32 row stride
// small parameter vectors, staged in LDS by the first requests of the kernel (a global load issued later would queue behind
// the weight fragments: vmcnt retires in order)
constexpr int UV_BIH = 0, UV_BHH = 3 * UM_D, UV_LNG = 6 * UM_D, UV_LNB = 7 * UM_D, UV_B1 = 8 * UM_D, UV_B2 = 8 * UM_D + UM_H,
              UV_QG = 9 * UM_D + UM_H, UV_QB = 10 * UM_D + UM_H, UM_NV = 11 * UM_D + UM_H;
// the NEXT form (the slot prologue of the following time step at the tail of the update): LayerNorm and biases of the residual-MLP predictor,
// bias of the kernel-distribution layer
constexpr int UV_PG = UM_NV, UV_PBT = UV_PG + UM_D, UV_PB0 = UV_PBT + UM_D, UV_PB2 = UV_PB0 + UM_H, UV_KB = UV_PB2 + UM_D, UM_NV_NEXT = UV_KB + 2 * UM_D;
constexpr int UM_KP = 2 * UM_D + 8;   // f32 pitch of the kernel-distribution rows (they take the place of the hidden planes)
static_assert(UM_H == 2 * UM_D && UM_NV == 1664 && UM_NV_NEXT == 5 * 512, "parameter-vector staging");

__device__ __forceinline__ void um_split4(__bf16* hp, __bf16* lp, int off, f32x4 v) {
  const bf16x4 hi = __builtin_convertvector(v, bf16x4);
  const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
  *(bf16x4*)(hp + off) = hi;
  *(bf16x4*)(lp + off) = lo;
}

// sigmoid / tanh on the hardware exponential (v_exp_f32, ~1 ulp): the gate arithmetic of 32 x 128 elements runs on four waves, and
// the libm forms took 5.6 us of a 20 us kernel; their error is far below the split-bf16 products feeding them
__device__ __forceinline__ float um_sigmoid(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float um_tanh(float x) { return 1.0f - 2.0f * __frcp_rn(__expf(2.0f * x) + 1.0f); }

// fragment (ks, nb, plane) of a packed [N][K] matrix (pack_linear_kernel, layer_fused.hip): 64 lanes x 16 B
// (an explicitly GLOBAL load: where the body runs inside a non-inlined function -- slot_chain.hip -- the pointer is generic, and a flat load counts on
//  lgkmcnt as well: every LDS wait of the update would wait for the weight stream)
__device__ __forceinline__ bf16x8 um_frag(const uint4* p, int nblocks, int nb, int ks, int pl, int lane) {
  typedef const uint4 __attribute__((address_space(1))) * gptr;
  return __builtin_bit_cast(bf16x8, ((gptr)p)[((long long)(ks * nblocks + nb) * 2 + pl) * 64 + lane]);
}

template <int NKS>
struct UmFrags {
  bf16x8 w[NKS][2];
};

template <int NKS>
__device__ __forceinline__ void um_load(UmFrags<NKS>& f, const uint4* p, int nblocks, int nb, int ks0, int lane) {
#pragma unroll
  for (int k = 0; k < NKS; ++k) {
    f.w[k][0] = um_frag(p, nblocks, nb, ks0 + k, 0, lane);
    f.w[k][1] = um_frag(p, nblocks, nb, ks0 + k, 1, lane);
  }
}

// acc[4 g + q] += out[token = lane & 31][column 32 nb + 8 g + 4 (lane >> 5) + q]; X planes [32][stride], k-steps ks0 .. ks0+NKS-1
template <int NKS>
__device__ __forceinline__ void um_gemm(f32x16& acc, const UmFrags<NKS>& f, const __bf16* Xh, const __bf16* Xl, int stride, int ks0,
                                        int lane) {
  const int ao = (lane & 31) * stride + 8 * (lane >> 5);
#pragma unroll
  for (int k = 0; k < NKS; ++k) {
    const bf16x8 xh = *(const bf16x8*)(Xh + ao + (ks0 + k) * 16), xl = *(const bf16x8*)(Xl + ao + (ks0 + k) * 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w[k][0], xl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w[k][1], xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w[k][0], xh, acc, 0, 0, 0);
  }
}

struct UmArgs {
  const float *part_num, *part_den;
  int P;        // records READ per row: records 0, pstep, 2 pstep, .. of the P * pstep the producer wrote
  int pstep;    // 2: the one-pass Slot-Attention kernel writes its sums into the even records and zeros into the odd ones (x + 0 = x: the same sums)
  const float* slots_prev;
  const uint4 *w_ih_p, *w_hh_p;   // [3D][D] packed
  const float *b_ih, *b_hh, *ln_g, *ln_b;
  const uint4* w1_p;              // [H][D]
  const float* b1;
  const uint4* w2_p;              // [D][H]
  const float* b2;
  float* slots_out;
  float* out2;                    // optional second destination: row (b, n) at out2 + b * out2_bs + n * D
  long long out2_bs;
  const float *q_ln_g, *q_ln_b;   // optional q projection (q_out NULL: off)
  const uint4* q_w_p;             // [D][D]
  float* q_out;
  int R, N;
  float ln_eps;
  // NEXT form only (um_body<true>): ResidualMLPPredictor (predictor.py:65-73), kernel_dist_layer + sampling (savi.py:190-200,355-365,401-402) of the
  // following time step; q_out then receives project_q of the SAMPLED slots
  const float *pm_ln_g, *pm_ln_b;
  const uint4* pm_w0_p;           // [2D][D]
  const float* pm_b0;
  const uint4* pm_w2_p;           // [D][2D]
  const float* pm_b2;
  int pm_norm_first;
  const uint4* kd_w_p;            // [2D][D]
  const float* kd_b;
  const float* noise;             // row (b, n) at noise + b * noise_bs + n * D, or NULL: the mean
  long long noise_bs;
  float* kdist_out;               // row (b, n) at kdist_out + b * kdist_bs + n * 2D, or NULL
  long long kdist_bs;
  float* nx_slots;                // [R][D] the sampled slots
};

}  // namespace

constexpr size_t UM_LDS = (size_t)(4 * UM_ROWS * UM_DP + 2 * UM_ROWS * UM_HP) * 2   // U, H/LN planes (hi, lo each) + hidden planes
                          + (size_t)2 * UM_ROWS * UM_FP * 4                          // previous / new rows as f32
                          + (size_t)4 * 2 * 16 * 64 * 4                              // wave-pair exchange: [4 blocks][2][16][64]
                          + (size_t)UM_ROWS * 64 * 4                                 // denominators [32][64]
                          + (size_t)UM_NV * 4;                                       // bias / LayerNorm vectors
constexpr size_t UM_LDS_NEXT = UM_LDS + (size_t)(UM_NV_NEXT - UM_NV) * 4;
static_assert((size_t)UM_ROWS * UM_KP * 4 <= (size_t)2 * UM_ROWS * UM_HP * 2, "kernel-distribution rows fit the hidden planes");

// The whole slot update of rows 32 * block .. + 31 by one 512-thread workgroup; um_lds: UM_LDS bytes of dynamic LDS.  A device function so that
// the workgroups can also ride as extra blocks of another launch (conv_rows4.hip: conv5x5_rows4_update_kernel).
// what differs between the rows a workgroup may be given: the stand-alone kernels take them from UmArgs (um_body below); the video-stationary slot chain
// (slot_chain.hip) passes its video's rows -- the weights and vectors then stay where the launch put them (kernel arguments: scalar loads at the point of
// use, not forty live registers)
// explicitly GLOBAL accesses (see um_frag): G4 = f32x4 load / store target, GF = float load
#define UM_G4(p) (*(f32x4 __attribute__((address_space(1)))*)(float __attribute__((address_space(1)))*)(p))
#define UM_CG4(p) (*(const f32x4 __attribute__((address_space(1)))*)(const float __attribute__((address_space(1)))*)(p))
#define UM_CGF(p) (*(const float __attribute__((address_space(1)))*)(p))
struct UmVar {
  const float *part_num, *part_den;
  int P, pstep;
  const float* slots_prev;
  float *slots_out, *out2, *q_out;
  const float* noise;
  float *kdist_out, *nx_slots;
  int R;
};

// ArgsT / VarT: UmArgs / UmVar, or their LDS-resident forms (address space 3) when the body runs inside a non-inlined function
template <bool NEXT = false, class ArgsT = UmArgs, class VarT = UmVar>
__device__ __forceinline__ void um_rows(const ArgsT& a, const VarT& rv, float* um_lds_generic, const int block) {
  // (generic -> LDS -> generic: inside a non-inlined function the address-space inference then still makes every access below a DS instruction)
  float* um_lds = (float*)(float __attribute__((address_space(3)))*)um_lds_generic;
  __bf16* Uh = (__bf16*)um_lds;                    // [32][DP]  updates
  __bf16* Ul = Uh + UM_ROWS * UM_DP;
  __bf16* Xh = Ul + UM_ROWS * UM_DP;               // [32][DP]  previous slots, later LN(h'), later LN_q(x)
  __bf16* Xl = Xh + UM_ROWS * UM_DP;
  __bf16* Hh = Xl + UM_ROWS * UM_DP;               // [32][HP]  relu hidden
  __bf16* Hl = Hh + UM_ROWS * UM_HP;
  float* Fp = (float*)(Hl + UM_ROWS * UM_HP);      // [32][FP]  previous slots (f32)
  float* Fn = Fp + UM_ROWS * UM_FP;                // [32][FP]  h', later the finished rows
  float* EX = Fn + UM_ROWS * UM_FP;                // [4][2][16][64] exchange between the two waves of a block
  float* DN = EX + 4 * 2 * 16 * 64;                // [32][64] denominators of the partial records
  float* PV = DN + UM_ROWS * 64;                   // parameter vectors (UV_* offsets)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int row0 = block * UM_ROWS;
  const int cb = wave & 3, half = wave >> 2;
  const int tok = lane & 31, kg = lane >> 5;

  UMTS(0);
  // ---- requests in the order they are needed: partial records, previous slots and the parameter vectors, THEN the weight
  //      fragments (the loads retire in order) ----
  float dnv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + UM_NT * i, r = idx >> 6, p = idx & 63, row = row0 + r;
    dnv[i] = 0.f;
    if (row < rv.R && p < rv.P) {
      const int b = row / a.N, n = row - b * a.N;
      dnv[i] = rv.part_den[((long long)b * rv.P * rv.pstep + (long long)p * rv.pstep) * a.N + n];
    }
  }
  const int ur = t >> 4, uc = (t & 15) * 8;   // thread = (row, 8 features)
  const bool rok = row0 + ur < rv.R;
  const int urow = min(row0 + ur, rv.R - 1), ub = urow / a.N, un = urow - ub * a.N;
  const float* pn = rv.part_num + ((long long)ub * rv.P * rv.pstep * a.N + un) * UM_D + uc;
  const long long pst = (long long)rv.pstep * a.N * UM_D;   // floats between two records read
  f32x4 pa[8][2];   // 8 partial records in flight at a time
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int pc = min(p, rv.P - 1);
    pa[p][0] = *(const f32x4*)(pn + pc * pst);
    pa[p][1] = *(const f32x4*)(pn + pc * pst + 4);
  }
  const f32x4 h0 = UM_CG4(rv.slots_prev + (long long)urow * UM_D + uc), h1 = UM_CG4(rv.slots_prev + (long long)urow * UM_D + uc + 4);
  constexpr int NPV = NEXT ? 5 : 4;
  float pvv[NPV];
#pragma unroll
  for (int i = 0; i < NPV; ++i) {
    const int j = t + UM_NT * i;   // UM_NV = 1664 <= 4 * 512; NEXT: 2560 = 5 * 512
    float v = 0.f;
    if (j < UV_BHH) v = UM_CGF(a.b_ih + (j));
    else if (j < UV_LNG) v = UM_CGF(a.b_hh + (j - UV_BHH));
    else if (j < UV_LNB) v = UM_CGF(a.ln_g + (j - UV_LNG));
    else if (j < UV_B1) v = UM_CGF(a.ln_b + (j - UV_LNB));
    else if (j < UV_B2) v = UM_CGF(a.b1 + (j - UV_B1));
    else if (j < UV_QG) v = UM_CGF(a.b2 + (j - UV_B2));
    else if (j < UV_QB) v = rv.q_out ? UM_CGF(a.q_ln_g + (j - UV_QG)) : 0.f;
    else if (j < UM_NV) v = rv.q_out ? UM_CGF(a.q_ln_b + (j - UV_QB)) : 0.f;
    else if constexpr (NEXT) {
      if (j < UV_PBT) v = UM_CGF(a.pm_ln_g + (j - UV_PG));
      else if (j < UV_PB0) v = UM_CGF(a.pm_ln_b + (j - UV_PBT));
      else if (j < UV_PB2) v = UM_CGF(a.pm_b0 + (j - UV_PB0));
      else if (j < UV_KB) v = UM_CGF(a.pm_b2 + (j - UV_PB2));
      else v = UM_CGF(a.kd_b + (j - UV_KB));
    }
    pvv[i] = v;
  }
  // the GRU fragments of this wave: half 0 = gate r (W_ir u + W_hr h) then W_in u; half 1 = gate z then W_hn h
  UmFrags<8> fa, fb;
  um_load(fa, a.w_ih_p, 12, half * 4 + cb, 0, lane);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    DN[t + UM_NT * i] = dnv[i];
    if (t + UM_NT * i < (NEXT ? UM_NV_NEXT : UM_NV)) PV[t + UM_NT * i] = pvv[i];
  }
  if constexpr (NEXT) PV[t + UM_NT * 4] = pvv[4];
  // ---- updates = sum_p num / sum_p den (records summed in order p = 0 .. P-1) ----
  f32x4 n0 = {0.f, 0.f, 0.f, 0.f}, n1 = n0;
#pragma unroll
  for (int p = 0; p < 8; ++p)
    if (p < rv.P) {
      n0 += pa[p][0];
      n1 += pa[p][1];
    }
  if (rv.P > 8) {   // (a second round of requests only when there are more than eight records to read)
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int pc = min(8 + p, rv.P - 1);
      pa[p][0] = *(const f32x4*)(pn + pc * pst);
      pa[p][1] = *(const f32x4*)(pn + pc * pst + 4);
    }
  }
  um_load(fb, a.w_hh_p, 12, half * 4 + cb, 0, lane);
  if (rv.P > 8) {
#pragma unroll
    for (int p = 0; p < 8; ++p)
      if (8 + p < rv.P) {
        n0 += pa[p][0];
        n1 += pa[p][1];
      }
    for (int p = 16; p < rv.P; ++p) {
      n0 += *(const f32x4*)(pn + p * pst);
      n1 += *(const f32x4*)(pn + p * pst + 4);
    }
  }
  UMTS(1);
  __syncthreads();
  UMTS(2);
  {
    float den = 0.f;
    for (int p = 0; p < rv.P; ++p) den += DN[ur * 64 + p];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    um_split4(Uh, Ul, ur * UM_DP + uc, rok ? n0 / den : zero4);
    um_split4(Uh, Ul, ur * UM_DP + uc + 4, rok ? n1 / den : zero4);
    um_split4(Xh, Xl, ur * UM_DP + uc, rok ? h0 : zero4);
    um_split4(Xh, Xl, ur * UM_DP + uc + 4, rok ? h1 : zero4);
    *(f32x4*)(Fp + ur * UM_FP + uc) = rok ? h0 : zero4;
    *(f32x4*)(Fp + ur * UM_FP + uc + 4) = rok ? h1 : zero4;
  }
  __syncthreads();

  UMTS(3);
  // ---- GRU: gate accumulator of this half (r or z), then its share of the n gate ----
  f32x16 g1, g2;
#pragma unroll
  for (int r = 0; r < 16; ++r) g1[r] = g2[r] = 0.f;
  um_gemm(g1, fa, Uh, Ul, UM_DP, 0, lane);
  um_load(fa, half == 0 ? a.w_ih_p : a.w_hh_p, 12, 8 + cb, 0, lane);   // n gate: W_in (half 0) / W_hn (half 1)
  um_gemm(g1, fb, Xh, Xl, UM_DP, 0, lane);
  UmFrags<8> fm;
  um_load(fm, a.w1_p, 8, wave, 0, lane);                                 // MLP layer 1 of this wave (lands under the gate math)
  um_gemm(g2, fa, half == 0 ? Uh : Xh, half == 0 ? Ul : Xl, UM_DP, 0, lane);
  UMTS(4);
  // biases: column c = 32 cb + 8 g + 4 kg + q
  if (half == 1) {
    // z = sigmoid(W_iz u + W_hz h + b_iz + b_hz), ghn = W_hn h + b_hn  -> exchange
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * cb + 8 * g + 4 * kg;
      const f32x4 bz = *(const f32x4*)(PV + UV_BIH + UM_D + c) + *(const f32x4*)(PV + UV_BHH + UM_D + c), bn = *(const f32x4*)(PV + UV_BHH + 2 * UM_D + c);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        EX[((cb * 2 + 0) * 16 + 4 * g + q) * 64 + lane] = um_sigmoid(g1[4 * g + q] + bz[q]);
        EX[((cb * 2 + 1) * 16 + 4 * g + q) * 64 + lane] = g2[4 * g + q] + bn[q];
      }
    }
  }
  else {
    // r = sigmoid(W_ir u + W_hr h + b_ir + b_hr) while the other half works on z (in place: g1 holds r behind the barrier)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * cb + 8 * g + 4 * kg;
      const f32x4 br = *(const f32x4*)(PV + UV_BIH + c) + *(const f32x4*)(PV + UV_BHH + c);
#pragma unroll
      for (int q = 0; q < 4; ++q) g1[4 * g + q] = um_sigmoid(g1[4 * g + q] + br[q]);
    }
  }
  __syncthreads();
  if (half == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * cb + 8 * g + 4 * kg;
      const f32x4 bn = *(const f32x4*)(PV + UV_BIH + 2 * UM_D + c);
      const f32x4 hp = *(const f32x4*)(Fp + tok * UM_FP + c);
      f32x4 hn;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float rr = g1[4 * g + q];
        const float z = EX[((cb * 2 + 0) * 16 + 4 * g + q) * 64 + lane], ghn = EX[((cb * 2 + 1) * 16 + 4 * g + q) * 64 + lane];
        const float nn = um_tanh(g2[4 * g + q] + bn[q] + rr * ghn);
        hn[q] = (1.f - z) * nn + z * hp[q];
      }
      *(f32x4*)(Fn + tok * UM_FP + c) = hn;
    }
  }
  __syncthreads();

  UMTS(5);
  // ---- LN(h') -> X planes (thread = (row, 8 features); a row is 16 consecutive lanes) ----
  UmFrags<8> f2;
  um_load(f2, a.w2_p, 4, cb, half * 8, lane);   // MLP layer 2: K half of this wave
  UmFrags<8> fp0, fp2, fk;                        // NEXT: predictor layer 1 / layer 2 (K half) / kernel distribution
  {
    const f32x4 v0 = *(const f32x4*)(Fn + ur * UM_FP + uc), v1 = *(const f32x4*)(Fn + ur * UM_FP + uc + 4);
    const float mu = sf_sum16(((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]))) / (float)UM_D;
    const f32x4 d0 = v0 - mu, d1 = v1 - mu;
    const float var = sf_sum16(((d0[0] * d0[0] + d0[1] * d0[1]) + (d0[2] * d0[2] + d0[3] * d0[3])) +
                               ((d1[0] * d1[0] + d1[1] * d1[1]) + (d1[2] * d1[2] + d1[3] * d1[3]))) / (float)UM_D;
    const float rs = 1.0f / sqrtf(var + a.ln_eps);
    um_split4(Xh, Xl, ur * UM_DP + uc, d0 * rs * *(const f32x4*)(PV + UV_LNG + uc) + *(const f32x4*)(PV + UV_LNB + uc));
    um_split4(Xh, Xl, ur * UM_DP + uc + 4, d1 * rs * *(const f32x4*)(PV + UV_LNG + uc + 4) + *(const f32x4*)(PV + UV_LNB + uc + 4));
  }
  __syncthreads();

  UMTS(6);
  // ---- hidden = relu(W1 LN + b1): wave = hidden block ----
#pragma unroll
  for (int r = 0; r < 16; ++r) g1[r] = 0.f;
  um_gemm(g1, fm, Xh, Xl, UM_DP, 0, lane);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c = 32 * wave + 8 * g + 4 * kg;
    const f32x4 bb = *(const f32x4*)(PV + UV_B1 + c);
    const f32x4 hv = {fmaxf(g1[4 * g] + bb[0], 0.f), fmaxf(g1[4 * g + 1] + bb[1], 0.f), fmaxf(g1[4 * g + 2] + bb[2], 0.f),
                      fmaxf(g1[4 * g + 3] + bb[3], 0.f)};
    um_split4(Hh, Hl, tok * UM_HP + c, hv);
  }
  __syncthreads();

  UMTS(7);
  // ---- x = h' + W2 hidden + b2: wave = (block, K half), halves meet in LDS ----
  UmFrags<4> fq;
  if (rv.q_out) um_load(fq, a.q_w_p, 4, cb, half * 4, lane);
  if constexpr (NEXT) um_load(fp0, a.pm_w0_p, 8, wave, 0, lane);   // (fm is dead: three fragment sets live at a time)
#pragma unroll
  for (int r = 0; r < 16; ++r) g2[r] = 0.f;
  um_gemm(g2, f2, Hh, Hl, UM_HP, half * 8, lane);
  if constexpr (NEXT) um_load(fp2, a.pm_w2_p, 4, cb, half * 8, lane);
  if (half == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) EX[(cb * 2 * 16 + r) * 64 + lane] = g2[r];
  }
  __syncthreads();
  if (half == 0) {
    const int row = row0 + tok;
    const int b = min(row, rv.R - 1) / a.N, n = min(row, rv.R - 1) - b * a.N;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * cb + 8 * g + 4 * kg;
      const f32x4 bb = *(const f32x4*)(PV + UV_B2 + c), hn = *(const f32x4*)(Fn + tok * UM_FP + c);
      f32x4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = hn[q] + (g2[4 * g + q] + EX[(cb * 2 * 16 + 4 * g + q) * 64 + lane]) + bb[q];
      if (row < rv.R) {
        UM_G4(rv.slots_out + (long long)row * UM_D + c) = v;
        if (rv.out2) UM_G4(rv.out2 + (long long)b * a.out2_bs + (long long)n * UM_D + c) = v;
      }
      *(f32x4*)(Fp + tok * UM_FP + c) = v;   // (Fp is dead: the finished rows for the q projection)
    }
  }
  UMTS(8);
  if (rv.q_out == nullptr) return;
  __syncthreads();

  if constexpr (NEXT) {
    // ===== the slot prologue of the NEXT time step on the finished rows (rows are independent: no other workgroup is involved) =====
    // noise of this thread's 8 features, requested now
    f32x4 nz0 = {0.f, 0.f, 0.f, 0.f}, nz1 = nz0;
    if (rv.noise && rok) {
      const float* np = rv.noise + (long long)ub * a.noise_bs + (long long)un * UM_D + uc;
      nz0 = UM_CG4(np);
      nz1 = UM_CG4(np + 4);
    }
    // ---- LN_p(x) -> X planes; the residual rows (LN_p(x) if norm_first, else x) -> Fn ----
    {
      const f32x4 v0 = *(const f32x4*)(Fp + ur * UM_FP + uc), v1 = *(const f32x4*)(Fp + ur * UM_FP + uc + 4);
      const float mu = sf_sum16(((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]))) / (float)UM_D;
      const f32x4 d0 = v0 - mu, d1 = v1 - mu;
      const float var = sf_sum16(((d0[0] * d0[0] + d0[1] * d0[1]) + (d0[2] * d0[2] + d0[3] * d0[3])) +
                                 ((d1[0] * d1[0] + d1[1] * d1[1]) + (d1[2] * d1[2] + d1[3] * d1[3]))) / (float)UM_D;
      const float rs = 1.0f / sqrtf(var + a.ln_eps);
      const f32x4 y0 = d0 * rs * *(const f32x4*)(PV + UV_PG + uc) + *(const f32x4*)(PV + UV_PBT + uc);
      const f32x4 y1 = d1 * rs * *(const f32x4*)(PV + UV_PG + uc + 4) + *(const f32x4*)(PV + UV_PBT + uc + 4);
      um_split4(Xh, Xl, ur * UM_DP + uc, y0);
      um_split4(Xh, Xl, ur * UM_DP + uc + 4, y1);
      *(f32x4*)(Fn + ur * UM_FP + uc) = a.pm_norm_first ? y0 : v0;
      *(f32x4*)(Fn + ur * UM_FP + uc + 4) = a.pm_norm_first ? y1 : v1;
    }
    __syncthreads();
    // ---- hidden = relu(W0 LN_p + b0): wave = hidden block ----
#pragma unroll
    for (int r = 0; r < 16; ++r) g1[r] = 0.f;
    um_gemm(g1, fp0, Xh, Xl, UM_DP, 0, lane);
    um_load(fk, a.kd_w_p, 8, wave, 0, lane);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * wave + 8 * g + 4 * kg;
      const f32x4 bb = *(const f32x4*)(PV + UV_PB0 + c);
      const f32x4 hv = {fmaxf(g1[4 * g] + bb[0], 0.f), fmaxf(g1[4 * g + 1] + bb[1], 0.f), fmaxf(g1[4 * g + 2] + bb[2], 0.f),
                        fmaxf(g1[4 * g + 3] + bb[3], 0.f)};
      um_split4(Hh, Hl, tok * UM_HP + c, hv);
    }
    __syncthreads();
    // ---- lat = residual + W2 hidden + b2: wave = (block, K half), halves meet in LDS; lat -> X planes ----
#pragma unroll
    for (int r = 0; r < 16; ++r) g2[r] = 0.f;
    um_gemm(g2, fp2, Hh, Hl, UM_HP, half * 8, lane);
    if (half == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) EX[(cb * 2 * 16 + r) * 64 + lane] = g2[r];
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 32 * cb + 8 * g + 4 * kg;
        const f32x4 bb = *(const f32x4*)(PV + UV_PB2 + c), res = *(const f32x4*)(Fn + tok * UM_FP + c);
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (g2[4 * g + q] + EX[(cb * 2 * 16 + 4 * g + q) * 64 + lane]) + bb[q] + res[q];
        um_split4(Xh, Xl, tok * UM_DP + c, v);
      }
    }
    __syncthreads();
    // ---- kernel distribution = Wkd lat + bkd: wave = 32 of its 2D columns; rows -> LDS (in place of the hidden planes) and kdist_out ----
    float* KD = (float*)Hh;   // [32][UM_KP]
#pragma unroll
    for (int r = 0; r < 16; ++r) g1[r] = 0.f;
    um_gemm(g1, fk, Xh, Xl, UM_DP, 0, lane);
    {
      const int row = row0 + tok;
      const int b = min(row, rv.R - 1) / a.N, n = min(row, rv.R - 1) - b * a.N;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 32 * wave + 8 * g + 4 * kg;
        const f32x4 bb = *(const f32x4*)(PV + UV_KB + c);
        const f32x4 v = {g1[4 * g] + bb[0], g1[4 * g + 1] + bb[1], g1[4 * g + 2] + bb[2], g1[4 * g + 3] + bb[3]};
        *(f32x4*)(KD + tok * UM_KP + c) = v;
        if (rv.kdist_out && row < rv.R) UM_G4(rv.kdist_out + (long long)b * a.kdist_bs + (long long)n * 2 * UM_D + c) = v;
      }
    }
    __syncthreads();
    // ---- slots = mu + noise * exp(0.5 logvar)  (thread = (row, 8 features)); the rows replace x for the q projection below ----
    {
      const f32x4 m0 = *(const f32x4*)(KD + ur * UM_KP + uc), m1 = *(const f32x4*)(KD + ur * UM_KP + uc + 4);
      const f32x4 l0 = *(const f32x4*)(KD + ur * UM_KP + UM_D + uc), l1 = *(const f32x4*)(KD + ur * UM_KP + UM_D + uc + 4);
      f32x4 v0 = m0, v1 = m1;
      if (rv.noise) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v0[q] += nz0[q] * expf(l0[q] * 0.5f);
          v1[q] += nz1[q] * expf(l1[q] * 0.5f);
        }
      }
      *(f32x4*)(Fp + ur * UM_FP + uc) = v0;
      *(f32x4*)(Fp + ur * UM_FP + uc + 4) = v1;
      if (rok) {
        UM_G4(rv.nx_slots + (long long)urow * UM_D + uc) = v0;
        UM_G4(rv.nx_slots + (long long)urow * UM_D + uc + 4) = v1;
      }
    }
  }

  UMTS(9);
  // ---- q = LN_q(x) Wq^T ----
  {
    const f32x4 v0 = *(const f32x4*)(Fp + ur * UM_FP + uc), v1 = *(const f32x4*)(Fp + ur * UM_FP + uc + 4);
    const float mu = sf_sum16(((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]))) / (float)UM_D;
    const f32x4 d0 = v0 - mu, d1 = v1 - mu;
    const float var = sf_sum16(((d0[0] * d0[0] + d0[1] * d0[1]) + (d0[2] * d0[2] + d0[3] * d0[3])) +
                               ((d1[0] * d1[0] + d1[1] * d1[1]) + (d1[2] * d1[2] + d1[3] * d1[3]))) / (float)UM_D;
    const float rs = 1.0f / sqrtf(var + a.ln_eps);
    um_split4(Xh, Xl, ur * UM_DP + uc, d0 * rs * *(const f32x4*)(PV + UV_QG + uc) + *(const f32x4*)(PV + UV_QB + uc));
    um_split4(Xh, Xl, ur * UM_DP + uc + 4, d1 * rs * *(const f32x4*)(PV + UV_QG + uc + 4) + *(const f32x4*)(PV + UV_QB + uc + 4));
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) g1[r] = 0.f;
  um_gemm(g1, fq, Xh, Xl, UM_DP, half * 4, lane);
  if (half == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) EX[((cb * 2 + 1) * 16 + r) * 64 + lane] = g1[r];
  }
  __syncthreads();
  if (half == 0 && row0 + tok < rv.R) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * cb + 8 * g + 4 * kg;
      f32x4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = g1[4 * g + q] + EX[((cb * 2 + 1) * 16 + 4 * g + q) * 64 + lane];
      *(f32x4*)(rv.q_out + (long long)(row0 + tok) * UM_D + c) = v;
    }
  }
  UMTS(10);
}

// the rows 32 * block .. + 31 of the launch's arguments
template <bool NEXT = false>
__device__ __forceinline__ void um_body(const UmArgs& a, float* um_lds, const int block) {
  const UmVar rv{a.part_num, a.part_den, a.P, a.pstep, a.slots_prev, a.slots_out, a.out2, a.q_out, a.noise, a.kdist_out, a.nx_slots, a.R};
  um_rows<NEXT>(a, rv, um_lds, block);
}
