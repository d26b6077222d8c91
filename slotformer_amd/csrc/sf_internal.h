// Internal C++ entry points shared between the translation units of libslotformer_hip.so.
#pragma once
#include "sf_common.h"

// Per-call options of the calling THREAD (the reference drives `forward` from one host thread per GPU,
// base_slots/extract_slots.py:128): set by the *_opts entry points for the duration of a call, never process-wide.
// precision / seam < 0 and ffn_rows / attn_heads == 0 mean "the process default" (sf_set_precision, sf_set_seam_fused,
// sf_set_ffn_rows64).
struct SfThreadOpts {
  int precision = -1;    // 0 exact f32, 1 split-bf16, 2 single-pass bf16
  int seam = -1;         // seam launches of the rollout off / on
  int ffn_rows = 0;      // rows per workgroup of the chunk-partial FFN launches: 32 / 64 / 128
  int attn_heads = 0;    // heads per workgroup of the layer attention launches: 2 (head pairs, four partials) / 8 (finished rows)
  int attn_rows = 0;     // 128: the attention block as q|k|v row tiles + one core workgroup per video (attn_rows.hip; finished rows)
  int ffn_tile = 0;      // 1: the FFN block as one workgroup per 64-row tile over all hidden chunks (ffn_tile.hip; finished rows)
  int layer_tok = 0;     // 1 / -1: the layers before the last as one token-stationary launch each (layer_tok.hip) on / off; 0: the process default
  int cus = 0;           // CUs the call's stream may use (its CU mask); 0: the whole chip.  Seam launches need all their workgroups co-resident
};
SfThreadOpts& sf_thread_opts();
// hipFuncAttributeMaxDynamicSharedMemorySize, once per (kernel, device) -- a process may drive several GPUs
int sf_ensure_dyn_lds(const void* kernel, size_t bytes);

int sf_linear_ex(const float* A, SfRowMap amap, const float* W, const float* bias, const float* ln_g,
                 const float* ln_b, float ln_eps, const float* res, SfRowMap rmap, int res_mod,
                 float* C, SfRowMap cmap, int M, int N, int K, int relu, hipStream_t stream, int ln_relu = 0);
int sf_linear_dropout_ex(const float* A, const float* W, const float* bias, const float* res, float* C, int M, int N, int K,
                         int relu, uint32_t drop_seed, uint32_t drop_thresh, float drop_scale, hipStream_t stream);
int sf_layernorm_ex(const float* x, SfRowMap xmap, const float* g, const float* b, float* y, SfRowMap ymap,
                    int rows, int D, float eps, hipStream_t st);
int sf_mha_ex(const float* qkv, float* out, int B, int L, int Lq, int d, int nheads, hipStream_t st);
int sf_lstm_pointwise_ex(const float* gates, const float* c_prev, float* h_out, float* c_out, int R, int H,
                         hipStream_t st);
int sf_sample_dist_ex(const float* dist, const float* noise, SfRowMap nmap, float* out, int R, int D,
                      hipStream_t st);
int sf_copy_rows_ex(const float* src, SfRowMap smap, float* dst, SfRowMap dmap, int rows, int cols,
                    hipStream_t st);
int sf_sa_pick_partials(int HW);
bool sf_pixel_mlp_feat_ok(int C0, int C1);
int sf_pixel_mlp_feat192_ex(const float* x, const float* ln0_g, const float* ln0_b, const void* w1p, const float* b1, const void* w2p,
                            const float* b2, const float* ln1_g, const float* ln1_b, float* feat, int M, float eps, hipStream_t st);
int sf_pixel_mlp_feat_ex(const float* x, const float* ln0_g, const float* ln0_b, const float* w1, const float* b1, const float* w2,
                         const float* b2, const float* ln1_g, const float* ln1_b, float* feat, int M, float eps, hipStream_t st);
bool sf_slot_update_mfma_ok(int D, int H, int P);
// The slot prologue of the NEXT time step (ResidualMLPPredictor -> kernel_dist Linear -> sampling; slot size 128) as the tail of a matrix-core slot
// update: the three matrices as sf_pack_linear_weights copies; the update's q_out then holds project_q of the sampled slots (slot_update_body.h)
struct SfNextStep {
  const float *pm_ln_g, *pm_ln_b;
  const void* pm_w0_p;
  const float* pm_b0;
  const void* pm_w2_p;
  const float* pm_b2;
  int norm_first;
  const void* kd_w_p;
  const float* kd_b;
  const float* noise;   // row (b, n) at noise + b * noise_bs + n * D, or NULL
  long long noise_bs;
  float* kdist_out;     // row (b, n) at kdist_out + b * kdist_bs + n * 2D, or NULL
  long long kdist_bs;
  float* slots;         // [B * N][D] the sampled slots (must not alias the update's slots_out)
};
int sf_slot_update_mfma_ex(const float* part_num, const float* part_den, int P, const float* slots_prev, const void* gru_ih_p,
                           const void* gru_hh_p, const float* gru_b_ih, const float* gru_b_hh, const float* ln_g, const float* ln_b,
                           const void* w1_p, const float* b1, const void* w2_p, const float* b2, float* slots_out, float* out2,
                           long long out2_bs, const float* q_ln_g, const float* q_ln_b, const void* q_w_p, float* q_out, int B, int N,
                           float ln_eps, hipStream_t st, const SfNextStep* next = nullptr, int p_step = 1);
// 1 when sf_slot_attn_iter_ex takes its one-pass tile kernel for this shape (sums in the even partial records, zeros in the odd ones)
bool sf_slot_attn_sparse_records(const float* k, const float* v, int HW, int D);
// the same update at slot size 192 / slot MLP 384 (slot_update_wide.hip); operands as for sf_slot_update_mfma_ex
bool sf_slot_update_wide_ok(int D, int H, int P);
int sf_slot_update_wide_ex(const float* part_num, const float* part_den, int P, const float* slots_prev, const void* gru_ih_p, const void* gru_hh_p,
                           const float* gru_b_ih, const float* gru_b_hh, const float* ln_g, const float* ln_b, const void* w1_p, const float* b1,
                           const void* w2_p, const float* b2, float* slots_out, float* out2, long long out2_bs, const float* q_ln_g,
                           const float* q_ln_b, const void* q_w_p, float* q_out, int B, int N, float ln_eps, hipStream_t st);
int sf_slot_update_ex(const float* part_num, const float* part_den, int P, const float* slots_prev,
                      const float* gru_w_ih, const float* gru_w_hh, const float* gru_b_ih, const float* gru_b_hh,
                      const float* ln_g, const float* ln_b, const float* mlp_w1, const float* mlp_b1, const float* mlp_w2,
                      const float* mlp_b2, float* slots_out, float* out2, long long out2_bs, const float* q_ln_g,
                      const float* q_ln_b, const float* q_w, float* q_out, int B, int N, int D, int H, float ln_eps,
                      hipStream_t st);
int sf_slot_prologue_ex(const float* prev, const float* init, const float* pm_ln_g, const float* pm_ln_b, const float* pm_w0_t,
                        const float* pm_b0, const float* pm_w2_t, const float* pm_b2, int norm_first, const float* kd_w_t,
                        const float* kd_b, const float* noise, long long noise_bs, float* kdist_out, long long kdist_bs,
                        const float* q_ln_g, const float* q_ln_b, const float* q_w_t, float* slots_out, float* q_out, int B,
                        int N, int D, float ln_eps, hipStream_t st);
int sf_slot_attn_iter_ex(const float* k, const float* v, int ld, long long batch_stride, const float* q,
                         float* part_num, float* part_den, float* attn_out, long long attn_batch_stride, int B,
                         int HW, int N, int D, float scale, float eps, hipStream_t st);

// kernel classes for the optional HIP-event timer (sf_runtime.cpp)
enum { SF_K_CONV_NHWC = 0, SF_K_CONV_FIRST = 1, SF_K_LINEAR = 2, SF_K_SA_ITER = 3, SF_K_SA_UPDATE = 4,
       SF_K_MHA = 5, SF_K_FFN = 6, SF_K_SEAM = 7, SF_K_DECONV = 8, SF_K_LAYER_TOK = 9, SF_K_NUM = 10 };
void sf_prof_begin(int cls, hipStream_t st, double work);
void sf_prof_end(int cls, hipStream_t st);
void sf_prof_suppress(int on);
int sf_qkv_attn_ex(const float* x, const float* ln_g, const float* ln_b, float ln_eps, const float* in_proj_w,
                   const float* in_proj_b, float* out, int B, int L, int Lq, int d, int nheads, hipStream_t st);
extern "C" int sf_get_precision(void);
// weights-stationary form of the same convolution (conv_ws.hip; the same bits); returns 1 when it does not apply
int sf_conv5x5_ws_ex(const float* in, const void* w_frag, const float* bias, const float* add, float* out, int F, int H, int W, int Cin, int Cout, int ks,
                     int relu, int n_workgroups, hipStream_t st);
extern "C" int sf_stream_cus(void* stream);
// one environment variable for every in-kernel time stamp / tuning override of the tools: SF_DBG="conv,lt,lf=16,sa,deconv,gemm,gemmcfg=132" -> value of
// `key` (1 when named without a value, 0 when absent)
int sf_dbg(const char* key);
extern "C" int sf_stream_cus_known(void* stream);   // 0: not a stream of sf_stream_create_cu_mask / sf_stream_set_cus
int sf_conv5x5_rows4_ex(const float* in, const void* w_frag, const float* bias, const float* add, float* out, int F, int H, int W,
                        int Cin, int Cout, int ks, int relu, hipStream_t st);
// decoder layers on fragment weights (deconv_s2.hip, conv_rows4.hip); return 1 when the kernel does not apply
int sf_deconv5x5s2_ex(const float* in, const void* w_frag, const float* bias, const float* head_w, const float* head_b, float* out, int R,
                      int H, int W, int Cin, int Cout, int ks, int stride, int relu, hipStream_t st);
int sf_conv5x5_rows4_head_ex(const float* in, const void* w_frag, const float* bias, const float* head_w, const float* head_b, float* dec,
                             int F, int H, int W, int Cin, int Cout, int ks, hipStream_t st);
int sf_conv5x5_halo_ex(const float* in, const float* w_packed, const float* bias, const float* add, float* out, int F,
                       int H, int W, int Cin, int Cout, int ks, int relu, hipStream_t st);
int sf_pixel_mlp_kv_ex(const float* x, const float* ln0_g, const float* ln0_b, const float* w1, const float* b1,
                       const float* w2, const float* b2, const float* ln1_g, const float* ln1_b, const float* wkv,
                       float* kv, int M, int C0, int C1, int ND, float eps, hipStream_t st);

// two-launch Transformer layer of the rollout (layer_fused.hip); sf_tfm_layer comes from the public header
bool sf_layer_fused_ok(int d, int heads, int ffn, int L);
int sf_conv_first_grouped_ex(const float* img, long long frame_stride, int fgroup, long long group_stride, const float* w, const float* bias,
                             const float* add, float* out, int F, int Cin, int Hin, int Win, int Cout, int ks, int stride, int relu, hipStream_t st);
int sf_conv_first_ex(const float* img, long long frame_stride, const float* w, const float* bias, const float* add,
                     float* out, int F, int Cin, int Hin, int Win, int Cout, int ks, int stride, int relu, hipStream_t st);

// training building blocks (rollout_train.hip), shared with slot_attn_train.hip
size_t sf_grad_partial_floats(long long rows, int N, int K);
int sf_grad_weight_ex(const float* Y, const float* X, float* dW, long long rows, int N, int K, float* partial, hipStream_t st);
int sf_grad_bias_ex(const float* Y, float* db, long long rows, int n, float* partial, hipStream_t st);
int sf_grad_ln_ex(const float* x, const float* dy, float* dgamma, float* dbeta, long long rows, int D, float eps, float* partial,
                  hipStream_t st);
int sf_ln_bwd_ex(const float* x, const float* dy, const float* gamma, const float* dres, float* out, long long rows, int D,
                 float eps, hipStream_t st);
int sf_transpose_ex(const float* in, float* out, int R, int Cn, hipStream_t st);
int sf_relu_bwd_ex(float* dh, const float* h, long long n, hipStream_t st);
int sf_conv2d_nhwc_strided_ex(const float* in, const float* w_packed, const float* bias, float* out, int F, int Hin, int Win,
                              int Cin, int Cout, int ks, int stride, int relu, hipStream_t stream, const float* relu_mask = nullptr);
int sf_linear_masked_ex(const float* A, const float* W, const float* mask, float scale, float* C, long long M, int N, int K,
                        hipStream_t stream);
int sf_conv_wgrad_ex(const float* A, int CA, int H, int W, const float* X, int Hx, int Wx, int s, int ks, long long rows,
                     float* out_oihw, float* partial, hipStream_t st);
size_t sf_conv_wgrad_partial_floats(int CA, int ks);
int sf_pos_dense_grad_ex(const float* d, int F, int HW, int C, const float* grid, float* dw, float* db, float* dtab, hipStream_t st);
int sf_slate_flash_train_ex(const float* q, const float* k, const float* v, float* out, float* lse, int ldq, int ldk, int ldv, int ldo,
                            long long q_bs, long long k_bs, long long v_bs, long long o_bs, int B, int L, int num_heads, int head_dim,
                            unsigned drop_seed, unsigned drop_thresh, float drop_scale, hipStream_t st);
