// Training of the Slot-Attention module: SlotAttention.forward (savi.py:56-102) with everything
// the backward pass needs kept in the caller's workspace, and its backward pass.
//
//   forward :  xn = LN(inputs);  k|v = xn [Wk;Wv]^T;  per iteration:  q = LN_q(slots) Wq^T;  attention half (slot_attn.hip,
//              one HBM pass over k|v);  updates = sum(num)/sum(den);  GRUCell;  slots = h + MLP(LN_m(h)).
//   backward:  the iterations in reverse -- MLP, GRU gates, attention half (slot_attn_bwd.hip, dk|dv accumulated across
//              the iterations in one buffer), q projection -- then the k|v projection and the input LayerNorm.
// As in rollout_train.hip nothing is reduced early: the per-iteration activations and gradients are stacked, every
// weight gradient is one split-bf16 MFMA contraction over the stacked rows, bias / LayerNorm gradients are column sums.
#include <math.h>

#include "../../include/slotformer_hip.h"
#include "sf_internal.h"

// updates[r][:] = sum_p num / sum_p den, r = b * N + n
__global__ __launch_bounds__(128) void sa_updates_kernel(const float* __restrict__ pn, const float* __restrict__ pd, int P,
                                                         float* __restrict__ upd, int N, int D) {
  const int n = blockIdx.x, b = blockIdx.y;
  float den = 0.f;
  for (int p = 0; p < P; ++p) den += pd[((long long)b * P + p) * N + n];
  const float inv = 1.f / den;
  for (int c = threadIdx.x; c < D; c += 128) {
    float num = 0.f;
    for (int p = 0; p < P; ++p) num += pn[(((long long)b * P + p) * N + n) * D + c];
    upd[((long long)b * N + n) * D + c] = num * inv;
  }
}

// nn.GRUCell pointwise part, gate order (r, z, n): gi, gh [R, 3D] (biases included), hp [R, D]
__global__ __launch_bounds__(256) void gru_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                      const float* __restrict__ hp, float* __restrict__ h, long long total,
                                                      int D) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long r = i / D;
  const int c = (int)(i - r * D);
  const float* a = gi + r * 3 * D + c;
  const float* b = gh + r * 3 * D + c;
  const float rr = sf_sigmoid(a[0] + b[0]);
  const float z = sf_sigmoid(a[D] + b[D]);
  const float n = tanhf(a[2 * D] + rr * b[2 * D]);
  h[i] = (1.f - z) * n + z * hp[i];
}
// its adjoint: dh -> dgi, dgh [R, 3D] and the direct path dhp = z * dh
__global__ __launch_bounds__(256) void gru_bwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                      const float* __restrict__ hp, const float* __restrict__ dh,
                                                      float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ dhp,
                                                      long long total, int D) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long r = i / D;
  const int c = (int)(i - r * D);
  const float* a = gi + r * 3 * D + c;
  const float* b = gh + r * 3 * D + c;
  const float rr = sf_sigmoid(a[0] + b[0]);
  const float z = sf_sigmoid(a[D] + b[D]);
  const float n = tanhf(a[2 * D] + rr * b[2 * D]);
  const float g = dh[i];
  const float dn = g * (1.f - z) * (1.f - n * n);   // gradient at the tanh argument
  const float dz = g * (hp[i] - n) * z * (1.f - z);
  const float dr = dn * b[2 * D] * rr * (1.f - rr);
  float* oa = dgi + r * 3 * D + c;
  float* ob = dgh + r * 3 * D + c;
  oa[0] = dr; oa[D] = dz; oa[2 * D] = dn;
  ob[0] = dr; ob[D] = dz; ob[2 * D] = dn * rr;
  dhp[i] = g * z;
}

namespace {

struct SaDims {
  int B, HW, N, D, Cin, H, I, P;
  long long M, R;
};

struct SaWs {
  float *wkv, *xn, *kv;
  float *sp, *sn, *q, *pn, *pd, *upd, *gi, *gh, *h, *hn, *hid;   // stacked over iterations
  float *dkv, *dxn, *ds, *dpre, *dhn, *dgi, *dgh, *dq, *dsn;      // ds .. dsn stacked over iterations
  float *dh, *dupd, *dsp, *dhp;
  float *w2t, *w1t, *wiht, *whht, *wqt, *wkvt, *dwkv;
  float *iter_ws, *partial;
  size_t iter_ws_bytes, total;
};

SaWs carve(const SaDims& d, float* base) {
  SaWs w;
  memset(&w, 0, sizeof(w));
  size_t off = 0;
  auto take = [&](size_t n) {
    float* p = base ? base + off : nullptr;
    off += (n + 63) & ~(size_t)63;
    return p;
  };
  const size_t M = d.M, R = d.R, D = d.D, I = d.I, H = d.H, C = d.Cin;
  w.wkv = take(2 * D * C);
  w.xn = take(M * C);
  w.kv = take(M * 2 * D);
  w.sp = take(I * R * D); w.sn = take(I * R * D); w.q = take(I * R * D);
  w.pn = take(I * (size_t)d.B * d.P * d.N * D); w.pd = take(I * (size_t)d.B * d.P * d.N);
  w.upd = take(I * R * D); w.gi = take(I * R * 3 * D); w.gh = take(I * R * 3 * D);
  w.h = take(I * R * D); w.hn = take(I * R * D); w.hid = take(I * R * H);
  w.dkv = take(M * 2 * D); w.dxn = take(M * C);
  w.ds = take(I * R * D); w.dpre = take(I * R * H); w.dhn = take(I * R * D);
  w.dgi = take(I * R * 3 * D); w.dgh = take(I * R * 3 * D); w.dq = take(I * R * D); w.dsn = take(I * R * D);
  w.dh = take(R * D); w.dupd = take(R * D); w.dsp = take(R * D); w.dhp = take(R * D);
  w.w2t = take(H * D); w.w1t = take(H * D); w.wiht = take(3 * D * D); w.whht = take(3 * D * D); w.wqt = take(D * D);
  w.wkvt = take(2 * D * C); w.dwkv = take(2 * D * C);
  w.iter_ws_bytes = sf_slot_attn_iter_bwd_workspace_bytes(d.B, d.HW, d.N, d.D);
  w.iter_ws = take(w.iter_ws_bytes / 4 + 64);
  size_t pf = sf_grad_partial_floats(d.M, 2 * d.D, d.Cin);
  const size_t alt[] = {sf_grad_partial_floats(I * R, 3 * d.D, d.D), sf_grad_partial_floats(I * R, d.H, d.D),
                        sf_grad_partial_floats(I * R, d.D, d.H), sf_grad_partial_floats(I * R, d.D, d.D)};
  for (size_t a : alt) pf = a > pf ? a : pf;
  w.partial = take(pf);
  w.total = off;
  return w;
}

int check(const sf_slot_attention* m, SaDims& d, int B, int HW, int iters) {
  SF_REQUIRE(m, "null model");
  SF_REQUIRE(B > 0 && HW > 0 && iters >= 1 && iters <= 8, "bad sizes");
  d.B = B; d.HW = HW; d.N = m->num_slots; d.D = m->slot_size; d.Cin = m->in_features; d.H = m->mlp_hidden; d.I = iters;
  SF_REQUIRE(d.D == 64 || d.D == 128 || d.D == 192 || d.D == 256, "slot_size must be 64 / 128 / 192 / 256");
  SF_REQUIRE(d.Cin % 64 == 0 && d.H % 64 == 0 && d.Cin <= 1024, "in_features and mlp_hidden must be multiples of 64");
  SF_REQUIRE(d.N >= 1 && d.N <= 8, "1..8 slots");
  d.P = sf_slot_attn_num_partials(HW);
  SF_REQUIRE((HW % d.P) == 0 && ((HW / d.P) % 16) == 0, "HW must be a multiple of 16");
  d.M = (long long)B * HW;
  d.R = (long long)B * d.N;
  return 0;
}

int gemm(const float* A, const float* W, const float* bias, const float* res, float* C, long long M, int N, int K, int relu,
         hipStream_t st) {
  return sf_linear_ex(A, sf_rows(K), W, bias, nullptr, nullptr, 0.f, res, sf_rows(N), 0, C, sf_rows(N), (int)M, N, K, relu, st);
}

int copy(float* dst, const float* src, size_t n, hipStream_t st) {
  hipError_t e = hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return sf_set_err((int)e, hipGetErrorString(e), __FILE__, __LINE__);
  return 0;
}

}  // namespace

extern "C" {

size_t sf_slot_attention_train_workspace_bytes(const sf_slot_attention* m, int B, int HW, int iters) {
  SaDims d;
  if (check(m, d, B, HW, iters) != 0) return 0;
  return carve(d, nullptr).total * sizeof(float) + 256;
}

int sf_slot_attention_train_fwd_f32(const sf_slot_attention* m, const float* inputs, const float* slots_in, int B, int HW,
                                    int iters, float* slots_out, void* ws, size_t ws_bytes, void* stream) {
  SaDims d;
  SF_TRY(check(m, d, B, HW, iters));
  SF_REQUIRE(inputs && slots_in && slots_out && ws, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  const SaWs w = carve(d, (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255));
  SF_REQUIRE(w.total * sizeof(float) + 256 <= ws_bytes, "workspace too small");
  const int D = d.D, C = d.Cin, H = d.H, N = d.N;
  const long long M = d.M, R = d.R;
  const float scale = 1.f / sqrtf((float)D);
  SF_TRY(copy(w.wkv, m->wk, (size_t)D * C, st));
  SF_TRY(copy(w.wkv + (size_t)D * C, m->wv, (size_t)D * C, st));
  SF_TRY(sf_layernorm_ex(inputs, sf_rows(C), m->norm_in_g, m->norm_in_b, w.xn, sf_rows(C), (int)M, C, 1e-5f, st));
  SF_TRY(gemm(w.xn, w.wkv, nullptr, nullptr, w.kv, M, 2 * D, C, 0, st));
  for (int it = 0; it < d.I; ++it) {
    const size_t o = (size_t)it * R;
    float* sp = w.sp + o * D;
    const float* prev = it == 0 ? slots_in : slots_out;
    SF_TRY(copy(sp, prev, (size_t)R * D, st));
    SF_TRY(sf_layernorm_ex(sp, sf_rows(D), m->q_ln_g, m->q_ln_b, w.sn + o * D, sf_rows(D), (int)R, D, 1e-5f, st));
    SF_TRY(gemm(w.sn + o * D, m->wq, nullptr, nullptr, w.q + o * D, R, D, D, 0, st));
    float* pn = w.pn + (size_t)it * B * d.P * N * D;
    float* pd = w.pd + (size_t)it * B * d.P * N;
    SF_TRY(sf_slot_attn_iter_ex(w.kv, w.kv + D, 2 * D, (long long)HW * 2 * D, w.q + o * D, pn, pd, nullptr, 0, B, HW, N, D, scale,
                                m->eps, st));
    hipLaunchKernelGGL(sa_updates_kernel, dim3(N, B), dim3(128), 0, st, pn, pd, d.P, w.upd + o * D, N, D);
    SF_CHECK_LAUNCH();
    SF_TRY(gemm(w.upd + o * D, m->gru_w_ih, m->gru_b_ih, nullptr, w.gi + o * 3 * D, R, 3 * D, D, 0, st));
    SF_TRY(gemm(sp, m->gru_w_hh, m->gru_b_hh, nullptr, w.gh + o * 3 * D, R, 3 * D, D, 0, st));
    hipLaunchKernelGGL(gru_fwd_kernel, dim3((unsigned)((R * D + 255) / 256)), dim3(256), 0, st, w.gi + o * 3 * D, w.gh + o * 3 * D,
                       sp, w.h + o * D, R * D, D);
    SF_CHECK_LAUNCH();
    SF_TRY(sf_layernorm_ex(w.h + o * D, sf_rows(D), m->mlp_ln_g, m->mlp_ln_b, w.hn + o * D, sf_rows(D), (int)R, D, 1e-5f, st));
    SF_TRY(gemm(w.hn + o * D, m->mlp_w1, m->mlp_b1, nullptr, w.hid + o * H, R, H, D, 1, st));
    SF_TRY(gemm(w.hid + o * H, m->mlp_w2, m->mlp_b2, w.h + o * D, slots_out, R, D, H, 0, st));
  }
  return 0;
}

int sf_slot_attention_train_bwd_f32(const sf_slot_attention* m, const float* inputs, const float* d_slots_out, float* d_inputs,
                                    float* d_slots_in, const sf_slot_attention_grads* g, int B, int HW, int iters, void* ws,
                                    size_t ws_bytes, void* stream) {
  SaDims d;
  SF_TRY(check(m, d, B, HW, iters));
  SF_REQUIRE(inputs && d_slots_out && d_slots_in && g && ws, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  const SaWs w = carve(d, (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255));
  SF_REQUIRE(w.total * sizeof(float) + 256 <= ws_bytes, "workspace too small");
  const int D = d.D, C = d.Cin, H = d.H, N = d.N, I = d.I;
  const long long M = d.M, R = d.R;
  const float scale = 1.f / sqrtf((float)D);
  SF_TRY(sf_transpose_ex(m->mlp_w2, w.w2t, D, H, st));       // [D,H] -> [H,D]
  SF_TRY(sf_transpose_ex(m->mlp_w1, w.w1t, H, D, st));       // [H,D] -> [D,H]
  SF_TRY(sf_transpose_ex(m->gru_w_ih, w.wiht, 3 * D, D, st));
  SF_TRY(sf_transpose_ex(m->gru_w_hh, w.whht, 3 * D, D, st));
  SF_TRY(sf_transpose_ex(m->wq, w.wqt, D, D, st));
  SF_TRY(sf_transpose_ex(w.wkv, w.wkvt, 2 * D, C, st));      // [2D,C] -> [C,2D]
  const float* ds_in = d_slots_out;
  for (int it = I - 1; it >= 0; --it) {
    const size_t o = (size_t)it * R;
    float* ds = w.ds + o * D;
    SF_TRY(copy(ds, ds_in, (size_t)R * D, st));
    // slots = h + W2 relu(W1 LN_m(h) + b1) + b2
    SF_TRY(sf_linear_masked_ex(ds, w.w2t, w.hid + o * H, 1.f, w.dpre + o * H, R, H, D, st));   // through W2 and the ReLU
    SF_TRY(gemm(w.dpre + o * H, w.w1t, nullptr, nullptr, w.dhn + o * D, R, D, H, 0, st));
    SF_TRY(sf_ln_bwd_ex(w.h + o * D, w.dhn + o * D, m->mlp_ln_g, ds, w.dh, R, D, 1e-5f, st));
    // GRUCell
    hipLaunchKernelGGL(gru_bwd_kernel, dim3((unsigned)((R * D + 255) / 256)), dim3(256), 0, st, w.gi + o * 3 * D, w.gh + o * 3 * D,
                       w.sp + o * D, w.dh, w.dgi + o * 3 * D, w.dgh + o * 3 * D, w.dhp, R * D, D);
    SF_CHECK_LAUNCH();
    SF_TRY(gemm(w.dgi + o * 3 * D, w.wiht, nullptr, nullptr, w.dupd, R, D, 3 * D, 0, st));
    SF_TRY(gemm(w.dgh + o * 3 * D, w.whht, nullptr, w.dhp, w.dsp, R, D, 3 * D, 0, st));
    // attention half: dk|dv accumulate over the iterations
    SF_TRY(sf_slot_attn_iter_bwd_f32(w.kv, w.kv + D, 2 * D, (long long)HW * 2 * D, w.q + o * D, w.pn + (size_t)it * B * d.P * N * D,
                                     w.pd + (size_t)it * B * d.P * N, d.P, w.dupd, w.dkv, w.dkv + D, it != I - 1, w.dq + o * D, B, HW,
                                     N, D, scale, m->eps, w.iter_ws, w.iter_ws_bytes, st));
    // q = LN_q(slots) Wq^T
    SF_TRY(gemm(w.dq + o * D, w.wqt, nullptr, nullptr, w.dsn + o * D, R, D, D, 0, st));
    SF_TRY(sf_ln_bwd_ex(w.sp + o * D, w.dsn + o * D, m->q_ln_g, w.dsp, it == 0 ? d_slots_in : w.dh, R, D, 1e-5f, st));
    ds_in = w.dh;   // the gradient w.r.t. this iteration's input slots feeds the previous iteration
  }
  // parameter gradients over the stacked iterations
  const long long rows = (long long)I * R;
  SF_TRY(sf_grad_weight_ex(w.ds, w.hid, g->mlp_w2, rows, D, H, w.partial, st));
  SF_TRY(sf_grad_bias_ex(w.ds, g->mlp_b2, rows, D, w.partial, st));
  SF_TRY(sf_grad_weight_ex(w.dpre, w.hn, g->mlp_w1, rows, H, D, w.partial, st));
  SF_TRY(sf_grad_bias_ex(w.dpre, g->mlp_b1, rows, H, w.partial, st));
  SF_TRY(sf_grad_ln_ex(w.h, w.dhn, g->mlp_ln_g, g->mlp_ln_b, rows, D, 1e-5f, w.partial, st));
  SF_TRY(sf_grad_weight_ex(w.dgi, w.upd, g->gru_w_ih, rows, 3 * D, D, w.partial, st));
  SF_TRY(sf_grad_bias_ex(w.dgi, g->gru_b_ih, rows, 3 * D, w.partial, st));
  SF_TRY(sf_grad_weight_ex(w.dgh, w.sp, g->gru_w_hh, rows, 3 * D, D, w.partial, st));
  SF_TRY(sf_grad_bias_ex(w.dgh, g->gru_b_hh, rows, 3 * D, w.partial, st));
  SF_TRY(sf_grad_weight_ex(w.dq, w.sn, g->wq, rows, D, D, w.partial, st));
  SF_TRY(sf_grad_ln_ex(w.sp, w.dsn, g->q_ln_g, g->q_ln_b, rows, D, 1e-5f, w.partial, st));
  // k|v projection and the input LayerNorm
  SF_TRY(sf_grad_weight_ex(w.dkv, w.xn, w.dwkv, M, 2 * D, C, w.partial, st));
  SF_TRY(copy(g->wk, w.dwkv, (size_t)D * C, st));
  SF_TRY(copy(g->wv, w.dwkv + (size_t)D * C, (size_t)D * C, st));
  SF_TRY(gemm(w.dkv, w.wkvt, nullptr, nullptr, w.dxn, M, C, 2 * D, 0, st));
  SF_TRY(sf_grad_ln_ex(inputs, w.dxn, g->norm_in_g, g->norm_in_b, M, C, 1e-5f, w.partial, st));
  if (d_inputs) SF_TRY(sf_ln_bwd_ex(inputs, w.dxn, m->norm_in_g, nullptr, d_inputs, M, C, 1e-5f, st));
  return 0;
}

}  // extern "C"
