// Host-buffer twins of the whole-path entry points + the K/V producer export.
//
// The device entry points never allocate; a caller that has no HIP binding of its own (plain C, ctypes + numpy)
// uses the sf_device_* helpers to place the WEIGHTS on the GPU once, and the *_host twins for the per-call
// data: each twin has the signature of its device counterpart, stages the data buffers through temporary
// device memory (hipMalloc / H2D / run / D2H / hipFree, synchronous) and takes a NULL workspace to mean
// "allocate it for this call".  They exist for integration and for the PCIe-inclusive measurement in DESIGN.md;
// the timed bench path is the device-resident one.
#include "sf_internal.h"
#include "../../include/slotformer_hip.h"

namespace {
#define SF_HIP(expr)                                                                           \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return sf_set_err((int)e_, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// RAII staging buffer: optional host source copied in, optional host destination copied back by flush().
struct Staged {
  void* d = nullptr;
  void* host = nullptr;
  size_t bytes = 0;
  bool back = false;
  ~Staged() {
    if (d) (void)hipFree(d);
  }
  int init(const void* src, void* dst, size_t nbytes, hipStream_t st) {
    bytes = nbytes;
    host = dst;
    back = dst != nullptr;
    if (nbytes == 0 || (src == nullptr && dst == nullptr)) return 0;
    SF_HIP(hipMalloc(&d, nbytes));
    if (src) SF_HIP(hipMemcpyAsync(d, src, nbytes, hipMemcpyHostToDevice, st));
    return 0;
  }
  int flush(hipStream_t st) {
    if (d && back) SF_HIP(hipMemcpyAsync(host, d, bytes, hipMemcpyDeviceToHost, st));
    return 0;
  }
};
}  // namespace

extern "C" {

int sf_device_alloc(void** dev_out, size_t bytes) {
  SF_REQUIRE(dev_out != nullptr && bytes > 0, "sf_device_alloc: null output or zero size");
  SF_HIP(hipMalloc(dev_out, bytes));
  return 0;
}

int sf_device_free(void* dev) {
  if (dev) SF_HIP(hipFree(dev));
  return 0;
}

int sf_device_upload(void* dev, const void* host, size_t bytes) {
  SF_REQUIRE(dev != nullptr && host != nullptr, "sf_device_upload: null pointer");
  SF_HIP(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
  return 0;
}

int sf_device_download(void* host, const void* dev, size_t bytes) {
  SF_REQUIRE(dev != nullptr && host != nullptr, "sf_device_download: null pointer");
  SF_HIP(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
  return 0;
}

int sf_device_synchronize(void) {
  SF_HIP(hipDeviceSynchronize());
  return 0;
}

// K/V producer (savi.py:245-250 encoder_out_layer, then savi.py:66-70 norm_inputs + project_k / project_v):
// feat [M, C0] channels-last CNN features (position embedding already added) -> kv [M, 2D] = (k | v).
// ws: M * 2 * C1 floats, used only when the fused kernel does not apply (exact-f32 mode or other sizes).
int sf_kv_producer_f32(const float* feat, const float* ln0_g, const float* ln0_b, const float* fc1_w,
                       const float* fc1_b, const float* fc2_w, const float* fc2_b, const float* ln1_g,
                       const float* ln1_b, const float* kv_w, float* kv, int M, int C0, int C1, int D, float ln_eps,
                       void* ws, size_t ws_bytes, void* stream) {
  SF_REQUIRE(feat && fc1_w && fc2_w && kv_w && kv && M > 0 && C0 > 0 && C1 > 0 && D > 0, "sf_kv_producer_f32: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int fused = 1;
  if (sf_get_precision() >= 1)
    fused = sf_pixel_mlp_kv_ex(feat, ln0_g, ln0_b, fc1_w, fc1_b, fc2_w, fc2_b, ln1_g, ln1_b, kv_w, kv, M, C0, C1, 2 * D,
                               ln_eps, st);
  if (fused != 1) return fused;
  SF_REQUIRE(ws != nullptr && ws_bytes >= (size_t)M * 2 * C1 * sizeof(float), "sf_kv_producer_f32: workspace too small");
  float* h1 = (float*)ws;
  float* h2 = h1 + (size_t)M * C1;
  SF_TRY(sf_linear_ex(feat, sf_rows(C0), fc1_w, fc1_b, ln0_g, ln0_b, ln_eps, nullptr, sf_rows(C1), 0, h1, sf_rows(C1), M,
                      C1, C0, 1, st));
  SF_TRY(sf_linear_ex(h1, sf_rows(C1), fc2_w, fc2_b, nullptr, nullptr, ln_eps, nullptr, sf_rows(C1), 0, h2, sf_rows(C1), M,
                      C1, C1, 0, st));
  SF_TRY(sf_linear_ex(h2, sf_rows(C1), kv_w, nullptr, ln1_g, ln1_b, ln_eps, nullptr, sf_rows(2 * D), 0, kv, sf_rows(2 * D),
                      M, 2 * D, C1, 0, st));
  return 0;
}

size_t sf_kv_producer_workspace_bytes(int M, int C1) { return (size_t)M * 2 * C1 * sizeof(float); }

int sf_rollout_f32_host(const sf_rollouter* m, float* slots_host, int B, int T_total, int pred_len, void* ws,
                        size_t ws_bytes, void* stream) {
  SF_REQUIRE(m && slots_host && B > 0 && T_total > 0, "sf_rollout_f32_host: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const size_t nb = (size_t)B * T_total * m->num_slots * m->slot_size * sizeof(float);
  Staged s, w;
  SF_TRY(s.init(slots_host, slots_host, nb, st));
  if (ws == nullptr) {
    ws_bytes = sf_rollout_workspace_bytes(m, B);
    SF_HIP(hipMalloc(&w.d, ws_bytes));
    ws = w.d;
  }
  SF_TRY(sf_rollout_f32(m, (float*)s.d, B, T_total, pred_len, ws, ws_bytes, stream));
  SF_TRY(s.flush(st));
  SF_HIP(hipStreamSynchronize(st));
  return 0;
}

int sf_savi_encode_f32_host(const sf_savi_encoder* m, const float* img_host, const float* noise_host,
                            const float* prev_slots_host, float* lstm_h_host, float* lstm_c_host, int state_valid,
                            float* post_slots_host, float* kernel_dist_host, float* attn_host, int B, int T, void* ws,
                            size_t ws_bytes, void* stream) {
  SF_REQUIRE(m && img_host && post_slots_host && B > 0 && T > 0, "sf_savi_encode_f32_host: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const size_t f4 = sizeof(float);
  const size_t N = m->num_slots, D = m->slot_size, res = m->resolution;
  Staged img, noise, prev, h, c, post, kd, attn, w;
  SF_TRY(img.init(img_host, nullptr, (size_t)B * T * 3 * res * res * f4, st));
  SF_TRY(noise.init(noise_host, nullptr, (size_t)B * T * N * D * f4, st));
  SF_TRY(prev.init(prev_slots_host, nullptr, (size_t)B * N * D * f4, st));
  const size_t hb = m->pred_rnn ? (size_t)B * N * m->pred_hidden * f4 : 0;
  SF_TRY(h.init(lstm_h_host, lstm_h_host, hb, st));
  SF_TRY(c.init(lstm_c_host, lstm_c_host, hb, st));
  SF_TRY(post.init(nullptr, post_slots_host, (size_t)B * T * N * D * f4, st));
  SF_TRY(kd.init(nullptr, kernel_dist_host, (size_t)B * T * N * 2 * D * f4, st));
  SF_TRY(attn.init(nullptr, attn_host, (size_t)B * T * N * 64 * 64 * f4, st));
  if (ws == nullptr) {
    ws_bytes = sf_savi_encode_workspace_bytes(m, B);
    SF_HIP(hipMalloc(&w.d, ws_bytes));
    ws = w.d;
  }
  SF_TRY(sf_savi_encode_f32(m, (const float*)img.d, (const float*)noise.d, (const float*)prev.d, (float*)h.d,
                            (float*)c.d, state_valid, (float*)post.d, (float*)kd.d, (float*)attn.d, B, T, ws, ws_bytes,
                            stream));
  SF_TRY(h.flush(st));
  SF_TRY(c.flush(st));
  SF_TRY(post.flush(st));
  SF_TRY(kd.flush(st));
  SF_TRY(attn.flush(st));
  SF_HIP(hipStreamSynchronize(st));
  return 0;
}

int sf_savi_decode_f32_host(const sf_savi_decoder* m, const float* slots_host, float* recon_combined_host,
                            float* recons_host, float* masks_host, int F, void* ws, size_t ws_bytes, void* stream) {
  SF_REQUIRE(m && slots_host && recon_combined_host && F > 0, "sf_savi_decode_f32_host: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const size_t f4 = sizeof(float), N = m->num_slots, px = (size_t)m->resolution * m->resolution;
  Staged s, rc, rs, mk, w;
  SF_TRY(s.init(slots_host, nullptr, (size_t)F * N * m->slot_size * f4, st));
  SF_TRY(rc.init(nullptr, recon_combined_host, (size_t)F * 3 * px * f4, st));
  SF_TRY(rs.init(nullptr, recons_host, (size_t)F * N * 3 * px * f4, st));
  SF_TRY(mk.init(nullptr, masks_host, (size_t)F * N * px * f4, st));
  if (ws == nullptr) {
    ws_bytes = sf_savi_decode_workspace_bytes(m, F);
    SF_HIP(hipMalloc(&w.d, ws_bytes));
    ws = w.d;
  }
  SF_TRY(sf_savi_decode_f32(m, (const float*)s.d, (float*)rc.d, (float*)rs.d, (float*)mk.d, F, ws, ws_bytes, stream));
  SF_TRY(rc.flush(st));
  SF_TRY(rs.flush(st));
  SF_TRY(mk.flush(st));
  SF_HIP(hipStreamSynchronize(st));
  return 0;
}

int sf_slot_attn_iter_f32_host(const float* k_host, const float* v_host, int ld, long long batch_stride,
                               const float* q_host, float* part_num_host, float* part_den_host, float* attn_out_host,
                               int B, int HW, int N, int D, float scale, float eps, void* stream) {
  SF_REQUIRE(k_host && v_host && q_host && part_num_host && part_den_host && B > 0, "sf_slot_attn_iter_f32_host: bad arguments");
  SF_REQUIRE(ld >= D && batch_stride >= (long long)HW * ld, "sf_slot_attn_iter_f32_host: bad strides");
  hipStream_t st = (hipStream_t)stream;
  const size_t f4 = sizeof(float);
  const int P = sf_slot_attn_num_partials(HW);
  // k and v may be two views into one (k | v) buffer (ld = 2D): stage each over its own extent
  const size_t ext = ((size_t)(B - 1) * batch_stride + (size_t)(HW - 1) * ld + D) * f4;
  Staged k, v, q, pn, pd, at;
  SF_TRY(k.init(k_host, nullptr, ext, st));
  SF_TRY(v.init(v_host, nullptr, ext, st));
  SF_TRY(q.init(q_host, nullptr, (size_t)B * N * D * f4, st));
  SF_TRY(pn.init(nullptr, part_num_host, (size_t)B * P * N * D * f4, st));
  SF_TRY(pd.init(nullptr, part_den_host, (size_t)B * P * N * f4, st));
  SF_TRY(at.init(nullptr, attn_out_host, (size_t)B * N * HW * f4, st));
  SF_TRY(sf_slot_attn_iter_f32((const float*)k.d, (const float*)v.d, ld, batch_stride, (const float*)q.d, (float*)pn.d,
                               (float*)pd.d, (float*)at.d, B, HW, N, D, scale, eps, stream));
  SF_TRY(pn.flush(st));
  SF_TRY(pd.flush(st));
  SF_TRY(at.flush(st));
  SF_HIP(hipStreamSynchronize(st));
  return 0;
}

// host twin of sf_slot_update_f32: the partial sums, the previous slots and the result are host buffers; the weights are
// device pointers (sf_device_upload) like every model parameter
int sf_slot_update_f32_host(const float* part_num_host, const float* part_den_host, int P, const float* slots_prev_host,
                            const float* gru_w_ih, const float* gru_w_hh, const float* gru_b_ih, const float* gru_b_hh,
                            const float* ln_g, const float* ln_b, const float* mlp_w1, const float* mlp_b1,
                            const float* mlp_w2, const float* mlp_b2, float* slots_out_host, int B, int N, int D, int H,
                            float ln_eps, void* stream) {
  SF_REQUIRE(part_num_host && part_den_host && slots_prev_host && slots_out_host && B > 0 && P > 0, "sf_slot_update_f32_host: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const size_t f4 = sizeof(float);
  Staged pn, pd, sp, so;
  SF_TRY(pn.init(part_num_host, nullptr, (size_t)B * P * N * D * f4, st));
  SF_TRY(pd.init(part_den_host, nullptr, (size_t)B * P * N * f4, st));
  SF_TRY(sp.init(slots_prev_host, nullptr, (size_t)B * N * D * f4, st));
  SF_TRY(so.init(nullptr, slots_out_host, (size_t)B * N * D * f4, st));
  SF_TRY(sf_slot_update_f32((const float*)pn.d, (const float*)pd.d, P, (const float*)sp.d, gru_w_ih, gru_w_hh, gru_b_ih, gru_b_hh,
                            ln_g, ln_b, mlp_w1, mlp_b1, mlp_w2, mlp_b2, (float*)so.d, B, N, D, H, ln_eps, stream));
  SF_TRY(so.flush(st));
  SF_HIP(hipStreamSynchronize(st));
  return 0;
}

// host twin of sf_kv_producer_f32: feat / kv are host buffers, the weights device pointers; ws == NULL -> allocated here
int sf_kv_producer_f32_host(const float* feat_host, const float* ln0_g, const float* ln0_b, const float* fc1_w,
                            const float* fc1_b, const float* fc2_w, const float* fc2_b, const float* ln1_g,
                            const float* ln1_b, const float* kv_w, float* kv_host, int M, int C0, int C1, int D, float ln_eps,
                            void* ws, size_t ws_bytes, void* stream) {
  SF_REQUIRE(feat_host && kv_host && M > 0 && C0 > 0 && C1 > 0 && D > 0, "sf_kv_producer_f32_host: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const size_t f4 = sizeof(float);
  Staged f, kv, w;
  SF_TRY(f.init(feat_host, nullptr, (size_t)M * C0 * f4, st));
  SF_TRY(kv.init(nullptr, kv_host, (size_t)M * 2 * D * f4, st));
  if (ws == nullptr) {
    ws_bytes = sf_kv_producer_workspace_bytes(M, C1);
    if (ws_bytes) SF_HIP(hipMalloc(&w.d, ws_bytes));
    ws = w.d;
  }
  SF_TRY(sf_kv_producer_f32((const float*)f.d, ln0_g, ln0_b, fc1_w, fc1_b, fc2_w, fc2_b, ln1_g, ln1_b, kv_w, (float*)kv.d, M, C0,
                            C1, D, ln_eps, ws, ws_bytes, stream));
  SF_TRY(kv.flush(st));
  SF_HIP(hipStreamSynchronize(st));
  return 0;
}

int sf_rollout_bf16_host(const sf_rollouter* m, float* slots_host, int B, int T_total, int pred_len, void* ws,
                         size_t ws_bytes, void* stream) {
  SF_REQUIRE(m && slots_host && B > 0 && T_total > 0, "sf_rollout_bf16_host: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const size_t nb = (size_t)B * T_total * m->num_slots * m->slot_size * sizeof(float);
  Staged s, w;
  SF_TRY(s.init(slots_host, slots_host, nb, st));
  if (ws == nullptr) {
    ws_bytes = sf_rollout_workspace_bytes(m, B);
    SF_HIP(hipMalloc(&w.d, ws_bytes));
    ws = w.d;
  }
  SF_TRY(sf_rollout_bf16(m, (float*)s.d, B, T_total, pred_len, ws, ws_bytes, stream));
  SF_TRY(s.flush(st));
  SF_HIP(hipStreamSynchronize(st));
  return 0;
}

}  // extern "C"
