// Slot update on the matrix cores (slot size 128, MLP 256): the same function as sa_slot_update_kernel (slot_attn.hip; savi.py:95-100
// + the next iteration's project_q, savi.py:45-48,79), for 32 rows (= slots) per workgroup.
//
// The VALU kernel is a chain of five thread-per-output products, each a round trip of weight requests (32 us for 224 rows,
// whatever the row count).  Here every product is a transposed split-bf16 MFMA GEMM -- the weights, packed once per weight
// version in fragment order (sf_pack_linear_weights), are the A operand straight from memory, the 32 rows are the B operand
// from LDS planes -- and the fragments of the NEXT product are requested before the current one runs, so the 721 KB of
// weights arrive as one stream (~7 us at a workgroup's ~100 GB/s) instead of five latency-bound bursts.
//
//   updates = sum_p num / sum_p den                                 thread = (row, 8 features), partials summed in order p = 0..P-1
//   GRUCell: r, z, n gates                                          wave = (hidden block w & 3, half w >> 2): half 0 runs r and
//                                                                   W_in u, half 1 runs z and W_hn h; they meet in LDS
//   x = h' + W2 relu(W1 LN(h') + b1) + b2                           W1: wave = hidden block; W2 / Wq: wave = (block, K half)
//   q = LN_q(x) Wq^T
// In-kernel timestamps (224 rows, whole chip): requests issued 2.2 us, updates ready 7.6, GRU products 10.1, gates 13.9, LN 15.2,
// MLP 16.3 / 18.1, q 19.9 us; 23 us per launch between events against 33 us for the VALU kernel.
#include "slot_update_body.h"

//
// NEXT form (round 4): on a time step's LAST iteration the same workgroups go on with the slot prologue of the following step -- residual-MLP
// predictor, kernel distribution, sampling, and the q of its first iteration from the sampled slots (savi.py:393-402; the stand-alone
// sa_slot_prologue_kernel of slot_attn.hip: 26-30 us of thread-per-output products) -- as three more streamed-fragment products: rows are
// independent, so nothing but the launch and its round trips goes away.
template <bool NEXT>
__global__ __launch_bounds__(UM_NT) void sa_slot_update_mfma_kernel(UmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float um_lds[];
  um_body<NEXT>(a, um_lds, blockIdx.x);
}

bool sf_slot_update_mfma_ok(int D, int H, int P) { return D == UM_D && H == UM_H && P >= 1 && P <= 64; }

// Packed operands: sf_pack_linear_weights of the torch-layout matrices (GRUCell weight_ih / weight_hh [3D][D], mlp[1].weight
// [H][D], mlp[3].weight [D][H], project_q[1].weight [D][D]).
int sf_slot_update_mfma_ex(const float* part_num, const float* part_den, int P, const float* slots_prev, const void* gru_ih_p,
                           const void* gru_hh_p, const float* gru_b_ih, const float* gru_b_hh, const float* ln_g, const float* ln_b,
                           const void* w1_p, const float* b1, const void* w2_p, const float* b2, float* slots_out, float* out2,
                           long long out2_bs, const float* q_ln_g, const float* q_ln_b, const void* q_w_p, float* q_out, int B, int N,
                           float ln_eps, hipStream_t st, const SfNextStep* next, int p_step) {
  SF_REQUIRE(part_num && part_den && slots_prev && slots_out && gru_ih_p && gru_hh_p && gru_b_ih && gru_b_hh && ln_g && ln_b && w1_p &&
                 b1 && w2_p && b2, "sf_slot_update_mfma_ex: null pointer");
  SF_REQUIRE(!next || (q_out && next->pm_ln_g && next->pm_ln_b && next->pm_w0_p && next->pm_b0 && next->pm_w2_p && next->pm_b2 && next->kd_w_p &&
                       next->kd_b && next->slots && next->slots != slots_out),
             "sf_slot_update_mfma_ex: the next-step form needs q_out, the predictor / kernel-distribution operands and its own slot buffer");
  SF_REQUIRE(q_out == nullptr || (q_ln_g && q_ln_b && q_w_p), "q projection requested without its weights");
  SF_REQUIRE(N >= 1 && P >= 1 && P <= 64 && (p_step == 1 || (p_step == 2 && P % 2 == 0)), "bad slot shape");
  if (B == 0) return 0;
  static_assert(UM_LDS_NEXT <= 160 * 1024, "slot update: LDS budget");
  SF_TRY(next ? sf_ensure_dyn_lds((const void*)sa_slot_update_mfma_kernel<true>, (size_t)(UM_LDS_NEXT))
              : sf_ensure_dyn_lds((const void*)sa_slot_update_mfma_kernel<false>, (size_t)(UM_LDS)));
  UmArgs a;
  memset(&a, 0, sizeof(a));
  a.part_num = part_num; a.part_den = part_den; a.P = P / p_step; a.pstep = p_step; a.slots_prev = slots_prev;
  a.w_ih_p = (const uint4*)gru_ih_p; a.w_hh_p = (const uint4*)gru_hh_p; a.b_ih = gru_b_ih; a.b_hh = gru_b_hh; a.ln_g = ln_g; a.ln_b = ln_b;
  a.w1_p = (const uint4*)w1_p; a.b1 = b1; a.w2_p = (const uint4*)w2_p; a.b2 = b2; a.slots_out = slots_out; a.out2 = out2; a.out2_bs = out2_bs;
  a.q_ln_g = q_ln_g; a.q_ln_b = q_ln_b; a.q_w_p = (const uint4*)q_w_p; a.q_out = q_out; a.R = B * N; a.N = N; a.ln_eps = ln_eps;
  const int R = B * N;
  if (next) {
    a.pm_ln_g = next->pm_ln_g; a.pm_ln_b = next->pm_ln_b; a.pm_w0_p = (const uint4*)next->pm_w0_p; a.pm_b0 = next->pm_b0;
    a.pm_w2_p = (const uint4*)next->pm_w2_p; a.pm_b2 = next->pm_b2; a.pm_norm_first = next->norm_first; a.kd_w_p = (const uint4*)next->kd_w_p;
    a.kd_b = next->kd_b; a.noise = next->noise; a.noise_bs = next->noise_bs; a.kdist_out = next->kdist_out; a.kdist_bs = next->kdist_bs;
    a.nx_slots = next->slots;
  }
  sf_prof_begin(SF_K_SA_UPDATE, st,
                2.0 * R * ((double)6 * UM_D * UM_D + 2.0 * UM_D * UM_H + (q_out ? (double)UM_D * UM_D : 0.0) + (next ? 6.0 * UM_D * UM_D : 0.0)));
  if (next)
    hipLaunchKernelGGL(sa_slot_update_mfma_kernel<true>, dim3((R + UM_ROWS - 1) / UM_ROWS), dim3(UM_NT), UM_LDS_NEXT, st, a);
  else
    hipLaunchKernelGGL(sa_slot_update_mfma_kernel<false>, dim3((R + UM_ROWS - 1) / UM_ROWS), dim3(UM_NT), UM_LDS, st, a);
  sf_prof_end(SF_K_SA_UPDATE, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// C ABI (include/slotformer_hip.h): device pointers; the five matrices as sf_pack_linear_weights copies
extern "C" int sf_slot_update_packed_f32(const float* part_num, const float* part_den, int P, const float* slots_prev, const void* gru_ih_packed,
                                         const void* gru_hh_packed, const float* gru_b_ih, const float* gru_b_hh, const float* ln_g,
                                         const float* ln_b, const void* mlp_w1_packed, const float* mlp_b1, const void* mlp_w2_packed,
                                         const float* mlp_b2, float* slots_out, const float* q_ln_g, const float* q_ln_b,
                                         const void* q_w_packed, float* q_out, int B, int N, int D, int H, float ln_eps, void* stream) {
  if (sf_slot_update_wide_ok(D, H, P))
    return sf_slot_update_wide_ex(part_num, part_den, P, slots_prev, gru_ih_packed, gru_hh_packed, gru_b_ih, gru_b_hh, ln_g, ln_b, mlp_w1_packed,
                                  mlp_b1, mlp_w2_packed, mlp_b2, slots_out, nullptr, 0, q_ln_g, q_ln_b, q_w_packed, q_out, B, N, ln_eps,
                                  (hipStream_t)stream);
  SF_REQUIRE(sf_slot_update_mfma_ok(D, H, P),
             "sf_slot_update_packed_f32: slot size 128 with slot MLP size 256, or 192 with 384, and at most 64 partial records");
  return sf_slot_update_mfma_ex(part_num, part_den, P, slots_prev, gru_ih_packed, gru_hh_packed, gru_b_ih, gru_b_hh, ln_g, ln_b, mlp_w1_packed,
                                mlp_b1, mlp_w2_packed, mlp_b2, slots_out, nullptr, 0, q_ln_g, q_ln_b, q_w_packed, q_out, B, N, ln_eps,
                                (hipStream_t)stream);
}

#ifdef UM_STAMPS
extern "C" int sf_debug_read_ts_um(long long* out16) {
  hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(um_ts), sizeof(long long) * 16);
  return e == hipSuccess ? 0 : (int)e;
}
#endif
