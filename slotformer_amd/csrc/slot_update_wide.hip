// Slot update on the matrix cores at slot size 192 / slot MLP 384 (the STEVE Physion configuration, steve_physion_params.py:41-52): the same
// function as sa_slot_update_kernel (slot_attn.hip; savi.py:95-100 + the next iteration's project_q, savi.py:45-48,79) for 32 rows per workgroup.
//
// slot_update_mfma.hip (slot size 128) keeps the WHOLE fragment set of a product in registers and requests the next set before the current one
// runs: three sets of 64 registers.  At 192 a set is 96 registers and the natural wave count is 12 (six 32-column blocks x two halves: 170 registers
// per lane), so the products here pull their fragments through a four-slot fragment ring instead (three chunks of two k-steps in flight per wave,
// 144 KB per CU), each product handing the ring over to the next one: the 1.6 MB of weights arrive as ONE stream across the barriers.
//
//   updates = sum_p num / sum_p den                                 thread = (row, 8 features), records summed in order p = 0..P-1
//   GRUCell: r, z, n gates                                          wave = (column block w % 6, half w / 6): half 0 runs r and W_in u,
//                                                                   half 1 runs z and W_hn h; they meet in LDS
//   x = h' + W2 relu(W1 LN(h') + b1) + b2                           W1: wave = hidden block (12); W2: wave = (block, K half)
//   q = LN_q(x) Wq^T                                                the six waves of half 0, K = 192 each
// The VALU kernel takes 58 us per launch at this size (profiles/r04_kernel_stats_C4.csv), whatever the row count.
#include "stream_mfma.h"

namespace {

constexpr int UW_D = 192, UW_H = 384, UW_ROWS = 32;
constexpr int UW_NB = UW_D / 32;          // 6 column blocks
constexpr int UW_NW = 2 * UW_NB;          // 12 waves
constexpr int UW_NT = 64 * UW_NW;         // 768 threads
constexpr int UW_KS = UW_D / 16;          // 12 k-steps over a slot row
constexpr int UW_DP = UW_D + 8, UW_HP = UW_H + 8;   // bf16 plane pitches
constexpr int UW_FP = UW_D + 4;                     // f32 row pitch
constexpr int UW_TPR = UW_D / 8;                    // 24 threads per row in the (row, 8 features) phases
static_assert(UW_ROWS * UW_TPR == UW_NT && UW_H == 2 * UW_D && UW_H / 32 == UW_NW, "wave / thread mappings");
// parameter vectors staged in LDS by the first requests of the kernel
constexpr int WV_BIH = 0, WV_BHH = 3 * UW_D, WV_LNG = 6 * UW_D, WV_LNB = 7 * UW_D, WV_B1 = 8 * UW_D, WV_B2 = 10 * UW_D, WV_QG = 11 * UW_D,
              WV_QB = 12 * UW_D, UW_NV = 13 * UW_D;
static_assert(UW_NV <= 4 * UW_NT, "parameter-vector staging");

struct UwArgs {
  const float *part_num, *part_den;
  int P;
  const float* slots_prev;
  const uint4 *w_ih_p, *w_hh_p;   // [3D][D] packed (sf_pack_linear_weights)
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
};

// region A: U planes + X planes; region B: hidden planes (before them: the denominators, then the GRU exchange); C: h' rows, later the finished
// rows; D: previous rows (f32), later the K-half exchange of the second MLP layer; then the parameter vectors
constexpr size_t UW_A = (size_t)4 * UW_ROWS * UW_DP * 2, UW_B = (size_t)2 * UW_ROWS * UW_HP * 2, UW_C = (size_t)UW_ROWS * UW_FP * 4;
constexpr size_t UW_LDS = UW_A + UW_B + 2 * UW_C + (size_t)UW_NV * 4;
static_assert((size_t)UW_NB * 2 * 16 * 64 * 4 <= UW_B && (size_t)UW_ROWS * 64 * 4 <= UW_B, "GRU exchange / denominators fit the hidden planes");
static_assert((size_t)UW_NB * 16 * 64 * 4 <= UW_C, "K-half exchange fits the previous rows");
static_assert(UW_LDS <= 160 * 1024, "slot update (192): LDS budget");

// LayerNorm of the f32 rows F -> split-bf16 planes (threads 0..511: thread = (row, 12 features); a row is 16 consecutive lanes)
__device__ __forceinline__ void uw_layernorm(const float* F, __bf16* Xh, __bf16* Xl, const float* g, const float* b, float eps, int t) {
  if (t >= 16 * UW_ROWS) return;
  const int r = t >> 4, c = (t & 15) * 12;
  f32x4 v[3], d[3];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    v[i] = *(const f32x4*)(F + r * UW_FP + c + 4 * i);
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mu = sf_sum16(s) / (float)UW_D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    d[i] = v[i] - mu;
    q += (d[i][0] * d[i][0] + d[i][1] * d[i][1]) + (d[i][2] * d[i][2] + d[i][3] * d[i][3]);
  }
  const float rs = 1.0f / sqrtf(sf_sum16(q) / (float)UW_D + eps);
#pragma unroll
  for (int i = 0; i < 3; ++i)
    ps_split4(Xh, Xl, r * UW_DP + c + 4 * i, d[i] * rs * *(const f32x4*)(g + c + 4 * i) + *(const f32x4*)(b + c + 4 * i));
}

// The fragment ring of a wave: four slots of TWO k-steps, three chunks (12 KB) in flight while one is consumed (the four-slot ring of three
// k-steps of stream_mfma.h is 96 registers: with two accumulators it spills at the 168 registers a lane has with twelve waves per CU).
// A product over 12 k-steps is six chunks, so consecutive products start at slot 0, 2, 0, ... (S0).
constexpr int UW_CH = 2, UW_NC = UW_KS / UW_CH;
struct UwRing {
  bf16x8 w[4][UW_CH][2];
};
__device__ __forceinline__ void uw_load(UwRing& s, int slot, const uint4* p, int nblocks, int nb, int ks0, int lane) {
#pragma unroll
  for (int k = 0; k < UW_CH; ++k) {
    s.w[slot][k][0] = ps_frag(p, nblocks, nb, ks0 + k, 0, lane);
    s.w[slot][k][1] = ps_frag(p, nblocks, nb, ks0 + k, 1, lane);
  }
}
// acc[4 g + q] += out[token = lane & 31][column 32 nb + 8 g + 4 (lane >> 5) + q] over k-steps ks0 .. ks0 + 11 of column block nb of the packed
// matrix p (X planes: k-step 0 at the pointer).  Chunks 0..2 are already in slots S0 .. S0 + 2; every iteration requests the chunk three ahead --
// behind the end of this product chunks 0..2 of the NEXT one (pn NULL: none), which starts at slot (S0 + 6) & 3.
template <int S0>
__device__ __forceinline__ void uw_block(f32x16& acc, UwRing& s, const uint4* p, int nblocks, int nb, int ks0, const uint4* pn, int nblocksn, int nbn,
                                         int ksn, const __bf16* Xh, const __bf16* Xl, int stride, int lane) {
  const int ao = (lane & 31) * stride + 8 * (lane >> 5);
#pragma unroll
  for (int c = 0; c < UW_NC; ++c) {
    if (c + 3 < UW_NC)
      uw_load(s, (S0 + c + 3) & 3, p, nblocks, nb, ks0 + (c + 3) * UW_CH, lane);
    else if (pn)
      uw_load(s, (S0 + c + 3) & 3, pn, nblocksn, nbn, ksn + (c + 3 - UW_NC) * UW_CH, lane);
#pragma unroll
    for (int k = 0; k < UW_CH; ++k) {
      const int ks = c * UW_CH + k;
      const bf16x8 xh = *(const bf16x8*)(Xh + ao + ks * 16), xl = *(const bf16x8*)(Xl + ao + ks * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s.w[(S0 + c) & 3][k][0], xl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s.w[(S0 + c) & 3][k][1], xh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s.w[(S0 + c) & 3][k][0], xh, acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);   // the requests stay one ring slot per chunk
  }
}

__global__ __launch_bounds__(UW_NT) void sa_slot_update_wide_kernel(UwArgs a) {
  extern __shared__ __attribute__((aligned(16))) float uw_lds[];
  __bf16* Uh = (__bf16*)uw_lds;                    // [32][DP]  updates
  __bf16* Ul = Uh + UW_ROWS * UW_DP;
  __bf16* Xh = Ul + UW_ROWS * UW_DP;               // [32][DP]  previous slots, later LN(h'), later LN_q(x)
  __bf16* Xl = Xh + UW_ROWS * UW_DP;
  __bf16* Hh = Xl + UW_ROWS * UW_DP;               // [32][HP]  relu hidden
  __bf16* Hl = Hh + UW_ROWS * UW_HP;
  float* EX = (float*)Hh;                          // [6 blocks][2][16][64]  GRU exchange (before the hidden planes exist)
  float* DN = (float*)Hh;                          // [32][64] denominators (before the exchange)
  float* Fn = (float*)(Hl + UW_ROWS * UW_HP);      // [32][FP]  h', later the finished rows
  float* Fp = Fn + UW_ROWS * UW_FP;                // [32][FP]  previous slots (f32)
  float* EX2 = Fp;                                 // [6 blocks][16][64]  K-half exchange of the second MLP layer (Fp is dead by then)
  float* PV = Fp + UW_ROWS * UW_FP;                // parameter vectors (WV_* offsets)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int row0 = blockIdx.x * UW_ROWS;
  const int cb = wave % UW_NB, half = wave / UW_NB;
  const int tok = lane & 31, kg = lane >> 5;

  // ---- requests in the order they are needed: denominators, partial records, previous slots, parameter vectors, THEN the weight fragments ----
  float dnv[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = t + UW_NT * i, r = idx >> 6, p = idx & 63, row = row0 + r;
    dnv[i] = 0.f;
    if (r < UW_ROWS && row < a.R && p < a.P) {
      const int b = row / a.N, n = row - b * a.N;
      dnv[i] = a.part_den[((long long)b * a.P + p) * a.N + n];
    }
  }
  const int ur = t / UW_TPR, uc = (t - ur * UW_TPR) * 8;   // thread = (row, 8 features)
  const bool rok = row0 + ur < a.R;
  const int urow = min(row0 + ur, a.R - 1), ub = urow / a.N, un = urow - ub * a.N;
  const float* pn = a.part_num + ((long long)ub * a.P * a.N + un) * UW_D + uc;
  constexpr int NPA = 4;   // partial records in flight at a time (eight, with the fragment ring behind them, spill at 168 registers per lane)
  f32x4 pa[NPA][2];
#pragma unroll
  for (int p = 0; p < NPA; ++p) {
    const int pc = min(p, a.P - 1);
    pa[p][0] = *(const f32x4*)(pn + (long long)pc * a.N * UW_D);
    pa[p][1] = *(const f32x4*)(pn + (long long)pc * a.N * UW_D + 4);
  }
  const f32x4 h0 = *(const f32x4*)(a.slots_prev + (long long)urow * UW_D + uc), h1 = *(const f32x4*)(a.slots_prev + (long long)urow * UW_D + uc + 4);
  float pvv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = t + UW_NT * i;
    float v = 0.f;
    if (j < WV_BHH) v = a.b_ih[j];
    else if (j < WV_LNG) v = a.b_hh[j - WV_BHH];
    else if (j < WV_LNB) v = a.ln_g[j - WV_LNG];
    else if (j < WV_B1) v = a.ln_b[j - WV_LNB];
    else if (j < WV_B2) v = a.b1[j - WV_B1];
    else if (j < WV_QG) v = a.b2[j - WV_B2];
    else if (j < WV_QB) v = a.q_out ? a.q_ln_g[j - WV_QG] : 0.f;
    else if (j < UW_NV) v = a.q_out ? a.q_ln_b[j - WV_QB] : 0.f;
    pvv[i] = v;
  }
  // the GRU products of this wave: half 0 = gate r (W_ir u + W_hr h) then W_in u; half 1 = gate z then W_hn h  (row blocks r 0-5, z 6-11, n 12-17)
  const int nb1 = half * UW_NB + cb, nb3 = 2 * UW_NB + cb;
  const uint4* p3 = half == 0 ? a.w_ih_p : a.w_hh_p;
  UwRing ring;
#pragma unroll
  for (int c = 0; c < 3; ++c) uw_load(ring, c, a.w_ih_p, 3 * UW_NB, nb1, c * UW_CH, lane);
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (t + UW_NT * i < UW_ROWS * 64) DN[t + UW_NT * i] = dnv[i];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (t + UW_NT * i < UW_NV) PV[t + UW_NT * i] = pvv[i];
  // ---- updates = sum_p num / sum_p den (records summed in order p = 0 .. P-1) ----
  f32x4 n0 = {0.f, 0.f, 0.f, 0.f}, n1 = n0;
  for (int p0 = 0; p0 < a.P; p0 += NPA) {
#pragma unroll
    for (int p = 0; p < NPA; ++p)
      if (p0 + p < a.P) {
        n0 += pa[p][0];
        n1 += pa[p][1];
      }
    if (p0 + NPA < a.P) {
#pragma unroll
      for (int p = 0; p < NPA; ++p) {
        const int pc = min(p0 + NPA + p, a.P - 1);
        pa[p][0] = *(const f32x4*)(pn + (long long)pc * a.N * UW_D);
        pa[p][1] = *(const f32x4*)(pn + (long long)pc * a.N * UW_D + 4);
      }
    }
  }
  __syncthreads();
  {
    float den = 0.f;
    for (int p = 0; p < a.P; ++p) den += DN[ur * 64 + p];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    ps_split4(Uh, Ul, ur * UW_DP + uc, rok ? n0 / den : zero4);
    ps_split4(Uh, Ul, ur * UW_DP + uc + 4, rok ? n1 / den : zero4);
    ps_split4(Xh, Xl, ur * UW_DP + uc, rok ? h0 : zero4);
    ps_split4(Xh, Xl, ur * UW_DP + uc + 4, rok ? h1 : zero4);
    *(f32x4*)(Fp + ur * UW_FP + uc) = rok ? h0 : zero4;
    *(f32x4*)(Fp + ur * UW_FP + uc + 4) = rok ? h1 : zero4;
  }
  __syncthreads();

  // ---- GRU: gate accumulator of this half (r or z), then its share of the n gate ----
  f32x16 g1, g2;
#pragma unroll
  for (int r = 0; r < 16; ++r) g1[r] = g2[r] = 0.f;
  uw_block<0>(g1, ring, a.w_ih_p, 3 * UW_NB, nb1, 0, a.w_hh_p, 3 * UW_NB, nb1, 0, Uh, Ul, UW_DP, lane);
  uw_block<2>(g1, ring, a.w_hh_p, 3 * UW_NB, nb1, 0, p3, 3 * UW_NB, nb3, 0, Xh, Xl, UW_DP, lane);
  uw_block<0>(g2, ring, p3, 3 * UW_NB, nb3, 0, a.w1_p, UW_NW, wave, 0, half == 0 ? Uh : Xh, half == 0 ? Ul : Xl, UW_DP, lane);
  // (the first chunks of the MLP's first layer are in flight under the gate math)
  if (half == 1) {
    // z = sigmoid(W_iz u + W_hz h + b_iz + b_hz), ghn = W_hn h + b_hn  -> exchange
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * cb + 8 * g + 4 * kg;
      const f32x4 bz = *(const f32x4*)(PV + WV_BIH + UW_D + c) + *(const f32x4*)(PV + WV_BHH + UW_D + c), bn = *(const f32x4*)(PV + WV_BHH + 2 * UW_D + c);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        EX[((cb * 2 + 0) * 16 + 4 * g + q) * 64 + lane] = ps_sigmoid(g1[4 * g + q] + bz[q]);
        EX[((cb * 2 + 1) * 16 + 4 * g + q) * 64 + lane] = g2[4 * g + q] + bn[q];
      }
    }
  }
  __syncthreads();
  if (half == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * cb + 8 * g + 4 * kg;
      const f32x4 br = *(const f32x4*)(PV + WV_BIH + c) + *(const f32x4*)(PV + WV_BHH + c), bn = *(const f32x4*)(PV + WV_BIH + 2 * UW_D + c);
      const f32x4 hp = *(const f32x4*)(Fp + tok * UW_FP + c);
      f32x4 hn;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float rr = ps_sigmoid(g1[4 * g + q] + br[q]);
        const float z = EX[((cb * 2 + 0) * 16 + 4 * g + q) * 64 + lane], ghn = EX[((cb * 2 + 1) * 16 + 4 * g + q) * 64 + lane];
        const float nn = ps_tanh(g2[4 * g + q] + bn[q] + rr * ghn);
        hn[q] = (1.f - z) * nn + z * hp[q];
      }
      *(f32x4*)(Fn + tok * UW_FP + c) = hn;
    }
  }
  __syncthreads();

  // ---- LN(h') -> X planes ----
  uw_layernorm(Fn, Xh, Xl, PV + WV_LNG, PV + WV_LNB, a.ln_eps, t);
  __syncthreads();

  // ---- hidden = relu(W1 LN + b1): wave = hidden block; the exchange words in the hidden planes were read before the barrier above ----
#pragma unroll
  for (int r = 0; r < 16; ++r) g1[r] = g2[r] = 0.f;
  uw_block<2>(g1, ring, a.w1_p, UW_NW, wave, 0, a.w2_p, UW_NB, cb, half * UW_KS, Xh, Xl, UW_DP, lane);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c = 32 * wave + 8 * g + 4 * kg;
    const f32x4 bb = *(const f32x4*)(PV + WV_B1 + c);
    const f32x4 hv = {fmaxf(g1[4 * g] + bb[0], 0.f), fmaxf(g1[4 * g + 1] + bb[1], 0.f), fmaxf(g1[4 * g + 2] + bb[2], 0.f),
                      fmaxf(g1[4 * g + 3] + bb[3], 0.f)};
    ps_split4(Hh, Hl, tok * UW_HP + c, hv);
  }
  __syncthreads();

  // ---- x = h' + W2 hidden + b2: wave = (block, K half), halves meet in LDS ----
  const bool want_q = a.q_out != nullptr;
  uw_block<0>(g2, ring, a.w2_p, UW_NB, cb, half * UW_KS, (want_q && half == 0) ? a.q_w_p : nullptr, UW_NB, cb, 0, Hh + half * UW_KS * 16,
                     Hl + half * UW_KS * 16, UW_HP, lane);
  if (half == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) EX2[(cb * 16 + r) * 64 + lane] = g2[r];
  }
  __syncthreads();
  if (half == 0) {
    const int row = row0 + tok;
    const int b = min(row, a.R - 1) / a.N, n = min(row, a.R - 1) - b * a.N;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * cb + 8 * g + 4 * kg;
      const f32x4 bb = *(const f32x4*)(PV + WV_B2 + c), hn = *(const f32x4*)(Fn + tok * UW_FP + c);
      f32x4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = hn[q] + (g2[4 * g + q] + EX2[(cb * 16 + 4 * g + q) * 64 + lane]) + bb[q];
      if (row < a.R) {
        *(f32x4*)(a.slots_out + (long long)row * UW_D + c) = v;
        if (a.out2) *(f32x4*)(a.out2 + (long long)b * a.out2_bs + (long long)n * UW_D + c) = v;
      }
      *(f32x4*)(Fn + tok * UW_FP + c) = v;   // in place: the finished rows for the q projection
    }
  }
  if (!want_q) return;
  __syncthreads();

  // ---- q = LN_q(x) Wq^T: the six waves of half 0 ----
  uw_layernorm(Fn, Xh, Xl, PV + WV_QG, PV + WV_QB, a.ln_eps, t);
  __syncthreads();
  if (half == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) g1[r] = 0.f;
    uw_block<2>(g1, ring, a.q_w_p, UW_NB, cb, 0, nullptr, 0, 0, 0, Xh, Xl, UW_DP, lane);
    if (row0 + tok < a.R) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 32 * cb + 8 * g + 4 * kg;
        const f32x4 v = {g1[4 * g], g1[4 * g + 1], g1[4 * g + 2], g1[4 * g + 3]};
        *(f32x4*)(a.q_out + (long long)(row0 + tok) * UW_D + c) = v;
      }
    }
  }
}

}  // namespace

bool sf_slot_update_wide_ok(int D, int H, int P) { return D == UW_D && H == UW_H && P >= 1 && P <= 64; }

// Packed operands as for sf_slot_update_mfma_ex (sf_pack_linear_weights of the torch-layout matrices).
int sf_slot_update_wide_ex(const float* part_num, const float* part_den, int P, const float* slots_prev, const void* gru_ih_p, const void* gru_hh_p,
                           const float* gru_b_ih, const float* gru_b_hh, const float* ln_g, const float* ln_b, const void* w1_p, const float* b1,
                           const void* w2_p, const float* b2, float* slots_out, float* out2, long long out2_bs, const float* q_ln_g,
                           const float* q_ln_b, const void* q_w_p, float* q_out, int B, int N, float ln_eps, hipStream_t st) {
  SF_REQUIRE(part_num && part_den && slots_prev && slots_out && gru_ih_p && gru_hh_p && gru_b_ih && gru_b_hh && ln_g && ln_b && w1_p && b1 &&
                 w2_p && b2, "sf_slot_update_wide_ex: null pointer");
  SF_REQUIRE(q_out == nullptr || (q_ln_g && q_ln_b && q_w_p), "q projection requested without its weights");
  SF_REQUIRE(N >= 1 && P >= 1 && P <= 64, "bad slot shape");
  if (B == 0) return 0;
  SF_TRY(sf_ensure_dyn_lds((const void*)sa_slot_update_wide_kernel, UW_LDS));
  UwArgs a;
  memset(&a, 0, sizeof(a));
  a.part_num = part_num; a.part_den = part_den; a.P = P; a.slots_prev = slots_prev;
  a.w_ih_p = (const uint4*)gru_ih_p; a.w_hh_p = (const uint4*)gru_hh_p; a.b_ih = gru_b_ih; a.b_hh = gru_b_hh; a.ln_g = ln_g; a.ln_b = ln_b;
  a.w1_p = (const uint4*)w1_p; a.b1 = b1; a.w2_p = (const uint4*)w2_p; a.b2 = b2; a.slots_out = slots_out; a.out2 = out2; a.out2_bs = out2_bs;
  a.q_ln_g = q_ln_g; a.q_ln_b = q_ln_b; a.q_w_p = (const uint4*)q_w_p; a.q_out = q_out; a.R = B * N; a.N = N; a.ln_eps = ln_eps;
  const int R = B * N;
  sf_prof_begin(SF_K_SA_UPDATE, st, 2.0 * R * ((double)6 * UW_D * UW_D + 2.0 * UW_D * UW_H + (q_out ? (double)UW_D * UW_D : 0.0)));
  hipLaunchKernelGGL(sa_slot_update_wide_kernel, dim3((R + UW_ROWS - 1) / UW_ROWS), dim3(UW_NT), UW_LDS, st, a);
  sf_prof_end(SF_K_SA_UPDATE, st);
  SF_CHECK_LAUNCH();
  return 0;
}
