/* slotformer_hip.h -- C ABI of libslotformer_hip.so (gfx950 / MI355X only).
 *
 * The reference (pairlab/SlotFormer) has NO native/FFI boundary: its hot path is Python
 * nn.Modules calling ATen/cuDNN.  This header is therefore the boundary a maintainer would
 * bind (ctypes); every entry point cites the reference call site whose
 * arithmetic it replaces (file:line under the reference tree).
 *
 * Conventions: extern "C"; int return (0 = ok, <0 = argument error, >0 = hipError_t);
 * sf_last_error_string() describes the last failure on the calling thread; every pointer is a
 * DEVICE pointer to row-major contiguous fp32 unless stated; the caller owns every buffer
 * (inputs, outputs, workspace -- query *_workspace_bytes()); `stream` is a hipStream_t passed
 * as void*; calls are asynchronous on that stream, re-entrant, never synchronise, and never
 * allocate device memory.
 */
#ifndef SLOTFORMER_HIP_H
#define SLOTFORMER_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

int sf_version(void);
const char* sf_last_error_string(void);

/* Matrix-arithmetic mode of every GEMM/conv kernel: 1 (default) = split-bf16 ("bf16x3": each f32 operand
 * = hi + lo bf16, hi*hi + hi*lo + lo*hi on bf16 MFMA with f32 accumulation, ~2^-17 relative per operand);
 * 0 = exact f32 MFMA (bitwise an fmaf chain); 2 = single-pass bf16 (hi*hi only, f32 accumulation, fp32 storage and master
 * weights) in the GEMM / convolution / weight-gradient cores -- the AMP-bf16 policy of the training path (row N1); kernels
 * without a single-pass variant keep mode 1.  Environment SF_PRECISION=f32 | bf16 selects 0 | 2 at load time. */
int sf_get_precision(void);
int sf_set_precision(int mode);

/* Optional per-kernel-class HIP-event timer (bench.py roofline): events bracket every launch of a
 * class on the launch stream while enabled (never during hipGraph capture).  Classes: 0 conv
 * NHWC implicit GEMM, 1 first conv, 2 linear, 3 slot-attention iteration, 4 slot update, 5 attention (incl. the fused
 * attention + out-proj kernel), 6 fused FFN, 7 seam launch, 8 the decoder's fragment-weight transposed convolutions / head-fused last layer.
 * sf_profile_read sums elapsed ms, launches and algorithmic work (FLOP, or bytes for class 3)
 * since the last read. */
int sf_profile_enable(int class_mask); /* bit c enables class c; 0 disables */
/* Seam launches of the rollout (last FFN + step boundary of step s and the first attention of step s+1 in one grid): on by
 * default (SF_SEAM_FUSED=0 in the environment: off).  They shorten the critical path of ONE rollout; when several rollouts
 * share the same CUs (pipeline partition 'pair') the spinning consumers waste CU time and the graphs are captured with it off. */
int sf_set_seam_fused(int on);
int sf_get_seam_fused(void);
/* 64 rows per workgroup in the chunk-partial FFN launches of the rollout (two 32-row tiles against one load of the weight
 * chunk: 40 % less CU time per launch, a slightly longer launch).  Off by default; the 'pair' pipeline, whose rollout CUs are
 * shared by two rollouts and therefore throughput-bound, captures its graphs with it on.  Bit-identical results either way. */
int sf_set_ffn_rows64(int on);
int sf_get_ffn_rows64(void);
int sf_profile_sample(int every);      /* bracket every `every`-th launch of an enabled class only (default 1: all) */
int sf_profile_read(int kernel_class, double* total_ms, long long* launches, double* work);

/* ---- building blocks ------------------------------------------------------------------ */

/* C[M,N] = act( LN?(A)[M,K] @ W[N,K]^T + bias ) + residual.   nn.Linear / F.layer_norm call
 * sites: savi.py:66-70,80,245-250; predictor.py:58-73; slotformer.py:115,121; torch
 * TransformerEncoderLayer linears.  W is the torch weight as stored ([out,in]).  ln_gamma/
 * ln_beta (both or neither) fuse a LayerNorm over K into the A load; residual may be NULL. */
int sf_linear_f32(const float* A, int lda, const float* W, const float* bias, const float* ln_gamma,
                  const float* ln_beta, float ln_eps, const float* residual, int ldr, float* C, int ldc,
                  int M, int N, int K, int relu, void* stream);

/* nn.LayerNorm over the last dim (savi.py:41,50,66; predictor.py:57). */
int sf_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, int rows, int D,
                     float eps, void* stream);

/* First encoder conv (savi.py:231-239, i == 0): NCHW image, frame f at img + f*frame_stride
 * floats; weight [Cout,Cin,ks,ks]; padding ks/2; output NHWC [F,Ho,Wo,Cout]; optional `add`
 * table [Ho*Wo,Cout] added after the activation. */
int sf_conv2d_nchw_in_f32(const float* img, long long frame_stride, const float* weight, const float* bias,
                          const float* add, float* out, int F, int Cin, int Hin, int Win, int Cout, int ks,
                          int stride, int relu, void* stream);

/* Encoder convs i > 0 (savi.py:231-239): stride 1, padding ks/2, NHWC in -> NHWC out,
 * w_packed [Cout,ks,ks,Cin] (sf_pack_conv_weight_f32); `add` as above (used for the soft
 * position embedding, utils.py:60-63 / savi.py:370). */
int sf_conv2d_nhwc_f32(const float* in, const float* w_packed, const float* bias, const float* add, float* out,
                       int F, int H, int W, int Cin, int Cout, int ks, int relu, void* stream);
int sf_pack_conv_weight_f32(const float* w_oihw, float* w_ohwi, int Cout, int Cin, int ks, void* stream);
/* 64 -> 64 channel 5 x 5 convolutions on a 64-pixel-wide grid: w_ohwi (above) -> split-bf16 copy in MFMA-fragment order,
 * sf_conv_frag_bytes(64, 64, 5) bytes; sf_conv5x5_frag_f32 is sf_conv2d_nhwc_f32 for that shape on the fragment copy (H % 4 == 0,
 * split-bf16 mode; csrc/conv_rows4.hip).  relu / add as sf_conv2d_nhwc_f32. */
/* Per-pixel chain of the SAVi encoder up to the normalised Slot-Attention inputs (64 -> 128 -> 128 channels; savi.py:245-250, 66-70):
 * feat [M][128] = LN(fc2(relu(fc1(LN(x))))), x [M][64]; torch-layout weights w1 [128][64], w2 [128][128].  form 0: one 128-pixel tile per
 * workgroup, weights through LDS; form 1: weights resident in registers, four tiles per workgroup; form 2 (the one the encode uses): the same on
 * 64-pixel tiles with 256 threads, two workgroups per CU (csrc/pixel_mlp.hip).  Same bits. */
int sf_pixel_feat_f32(const float* x, const float* ln0_g, const float* ln0_b, const float* w1, const float* b1, const float* w2,
                      const float* b2, const float* ln1_g, const float* ln1_b, float* feat, int M, float eps, int form, void* stream);
size_t sf_conv_frag_bytes(int Cout, int Cin, int ks);
/* Opt-in arithmetic of the fragment-weight convolution (process-wide; default 0 = split-bf16, three MFMAs per product, like every other bf16x3
 * kernel): 1 = activations as two fp16 terms x weights rounded to ONE fp16 (2^-12 per weight), two MFMAs per product and half the weight stream.
 * A measured trade (profiles/r03_probes.txt section 14), not the path the parity numbers are quoted on.  Also SF_CONV_FP16X2=1. */
int sf_set_conv_fp16x2(int on);
int sf_get_conv_fp16x2(void);
int sf_pack_conv_frag_weights(const float* w_ohwi, void* frag, int Cout, int Cin, int ks, void* stream);
int sf_conv5x5_frag_f32(const float* in, const void* w_frag, const float* bias, const float* add, float* out, int F, int H, int W,
                        int relu, void* stream);
/* The same convolution on the same fragment copy with the WEIGHTS STATIONARY IN REGISTERS (csrc/conv_ws.hip): one four-wave workgroup per CU keeps the
 * layer's 410 KB of split-bf16 fragments in its register file and walks its share of the F * H output rows through a ring of six halo rows in LDS --
 * what the encode launches for layers i > 0 when a launch gives every CU of its stream at least four rows (savi.py:231-239).  Bit-identical to
 * sf_conv5x5_frag_f32.  Any H >= 1; relu 0 / 1; n_workgroups 0 = one per CU of the stream (sf_stream_create_cu_mask streams: their CUs), else
 * that many (any split of the rows gives the same bits). */
int sf_conv5x5_ws_f32(const float* in, const void* w_frag, const float* bias, const float* add, float* out, int F, int H, int W,
                      int relu, int n_workgroups, void* stream);

/* 64 -> 64 channel 5 x 5 stride-2 transposed convolutions (padding 2, output_padding 1: H x W -> 2H x 2W; the decoder layers of savi.py:252-293):
 * w_ohwi (sf_pack_deconv_weight_f32) -> split-bf16 copy in the kernel's consumption order, sf_deconv_frag_bytes(64, 64, 5, 2) bytes;
 * sf_deconv5x5s2_frag_f32: in [R][H][W][64] -> out [R][2H][2W][64] (+ bias, relu 0 / 1), W in {16, 32, 64}, H % (256 / W) == 0;
 * sf_deconv5x5s2_head_frag_f32 (W == 64): dec [R][2H * 2W][4] = head_w [4][64] . relu(deconv(in) + bias) + head_b [4] -- the last decoder layer with
 * the 1x1 output convolution (savi.py:286-289) in its epilogue.  Split-bf16 mode only (csrc/deconv_s2.hip). */
size_t sf_deconv_frag_bytes(int Cout, int Cin, int ks, int stride);
int sf_pack_deconv_frag_weights(const float* w_ohwi, void* frag, int Cout, int Cin, int ks, int stride, void* stream);
int sf_deconv5x5s2_frag_f32(const float* in, const void* w_frag, const float* bias, float* out, int R, int H, int W, int relu, void* stream);
int sf_deconv5x5s2_head_frag_f32(const float* in, const void* w_frag, const float* bias, const float* head_w, const float* head_b, float* dec,
                                 int R, int H, int W, void* stream);
/* First decoder layer on its broadcast input (sf_savi_decoder.l0_weff / l0_posterm): out [R][2 res][2 res][C1] = relu(table[r][class(p)] + posterm[p]),
 * table [R][25 * C1]. */
int sf_decode_l0_expand_f32(const float* table, const float* posterm, float* out, int R, int res, int C1, void* stream);

/* Decoder building blocks (savi.py:252-293,504-525).  ConvTranspose2d(k, stride, padding=k/2,
 * output_padding=stride-1) as a gather implicit GEMM: NHWC in [F,Hin,Win,Cin] -> [F,Hin*s,Win*s,Cout];
 * w_packed [Cout,ks,ks,Cin] from the torch weight [Cin,Cout,ks,ks] (sf_pack_deconv_weight_f32). */
int sf_conv_transpose2d_nhwc_f32(const float* in, const float* w_packed, const float* bias, float* out, int F,
                                 int Hin, int Win, int Cin, int Cout, int ks, int stride, int relu, void* stream);
int sf_pack_deconv_weight_f32(const float* w_iohw, float* w_ohwi, int Cin, int Cout, int ks, void* stream);
/* out[R,P,D] = slots[R,D] (broadcast over the P = r*r positions) + table[P,D]  (savi.py:512-517). */
int sf_slot_broadcast_f32(const float* slots, const float* table, float* out, int R, int P, int D, void* stream);
/* dec [F*N,HW,4] (rgb + mask logit per slot) -> masks = softmax over slots [F,N,1,H,W], recons [F,N,3,H,W],
 * recon_combined = sum_n recons*masks [F,3,H,W]  (savi.py:519-525); recons / masks may be NULL. */
int sf_decode_combine_f32(const float* dec, float* recon_combined, float* recons, float* masks, int F, int N, int HW,
                          void* stream);
/* ... followed by postproc_mask (vp_utils.py:20-41) on those masks: seg [F,H,W] = the slot with the smallest peak mask value over the frame
 * (the background) where the best mask value is below fg_thre, else the argmax over slots; as int64 (the reference's dtype) and / or
 * uint8, either may be NULL.  slot_max: [F * N] words of scratch.  recon_combined is always written; recons / masks may be NULL. */
int sf_decode_combine_seg_f32(const float* dec, float* recon_combined, float* recons, float* masks, long long* seg_i64,
                              unsigned char* seg_u8, float fg_thre, unsigned* slot_max, int F, int N, int HW, void* stream);
/* postproc_mask (vp_utils.py:20-41) on given masks [F,N,HW] (any float values) -> seg [F,HW] int64 and / or uint8; slot_max: [F * N] words. */
int sf_postproc_mask_f32(const float* masks, long long* seg_i64, unsigned char* seg_u8, float fg_thre, unsigned* slot_max, int F, int N,
                         int HW, void* stream);

/* table[HW,C] = dense(grid)  (SoftPositionEmbed, utils.py:52-63; grid [HW,4]). */
int sf_pos_embed_table_f32(const float* grid, const float* dense_w, const float* dense_b, float* table, int HW,
                           int C, void* stream);

/* One Slot-Attention iteration, attention half (savi.py:82-89 / steve.py:50-60): logits,
 * softmax over slots, +eps, and the per-slot sums over pixels as P partial records per batch.
 * k,v rows have leading dimension ld and batch_stride floats between batches. */
int sf_slot_attn_num_partials(int HW);
int sf_slot_attn_iter_f32(const float* k, const float* v, int ld, long long batch_stride, const float* q,
                          float* part_num, float* part_den, float* attn_out, int B, int HW, int N, int D,
                          float scale, float eps, void* stream);

/* bf16-STORAGE variant (SURVEY.md 8(b2) `sf_slot_attn_iter_bf16`): k, v are bf16 rows (ld / batch_stride in elements), half the
 * bytes of this HBM-bound kernel; logits, softmax, the weighted sums and every output stay f32.  Rounding K/V to bf16 costs
 * ~2e-3 relative on the updates (tests/test_kernels_gpu.py) -- outside the encode path's 5e-5, hence an option only. */
int sf_slot_attn_iter_bf16(const void* k, const void* v, int ld, long long batch_stride, const float* q, float* part_num,
                           float* part_den, float* attn_out, int B, int HW, int N, int D, float scale, float eps, void* stream);

/* Backward of sf_slot_attn_iter_f32 (row N1: savi.py:82-94 under autograd).  Inputs of the forward call (k, v, q, the
 * partial records it produced) plus d_updates [B,N,D], the gradient w.r.t. updates = sum(num) / sum(den).  Writes
 * dq [B,N,D] and dk / dv (same row layout as k / v); accumulate != 0 adds into dk / dv instead (the iterations of one
 * frame share k and v).  slot_size 64 / 128 / 192 / 256, at most 8 slots. */
size_t sf_slot_attn_iter_bwd_workspace_bytes(int B, int HW, int N, int D);
int sf_slot_attn_iter_bwd_f32(const float* k, const float* v, int ld, long long batch_stride, const float* q,
                              const float* part_num, const float* part_den, int P, const float* d_updates, float* dk,
                              float* dv, int accumulate, float* dq, int B, int HW, int N, int D, float scale, float eps,
                              void* ws, size_t ws_bytes, void* stream);

/* Slot update (savi.py:95-100): updates = sum(num)/sum(den); GRUCell (r,z,n); slots + MLP(LN(slots)).
 * The four weight MATRICES are passed transposed ([in,out] = torch weight.t().contiguous()):
 * gru_w_ih [D,3D], gru_w_hh [D,3D], mlp_w1 [D,H], mlp_w2 [H,D]; biases / LN vectors as in torch. */
int sf_slot_update_f32(const float* part_num, const float* part_den, int P, const float* slots_prev,
                       const float* gru_w_ih, const float* gru_w_hh, const float* gru_b_ih,
                       const float* gru_b_hh, const float* ln_g, const float* ln_b, const float* mlp_w1,
                       const float* mlp_b1, const float* mlp_w2, const float* mlp_b2, float* slots_out, int B,
                       int N, int D, int H, float ln_eps, void* stream);

/* The same slot update on the matrix cores (split-bf16 products; slot size D = 128 with slot MLP size H = 256, or D = 192 with H = 384), plus the
 * next iteration's q = project_q(slots_out) (savi.py:45-48,79) when q_out is not NULL.  The five matrices are
 * sf_pack_linear_weights copies of the TORCH-layout weights: GRUCell weight_ih / weight_hh [3D, D], mlp[1].weight [H, D],
 * mlp[3].weight [D, H], project_q[1].weight [D, D]. */
int sf_slot_update_packed_f32(const float* part_num, const float* part_den, int P, const float* slots_prev,
                              const void* gru_ih_packed, const void* gru_hh_packed, const float* gru_b_ih,
                              const float* gru_b_hh, const float* ln_g, const float* ln_b, const void* mlp_w1_packed,
                              const float* mlp_b1, const void* mlp_w2_packed, const float* mlp_b2, float* slots_out,
                              const float* q_ln_g, const float* q_ln_b, const void* q_w_packed, float* q_out, int B, int N,
                              int D, int H, float ln_eps, void* stream);

/* nn.MultiheadAttention core for short sequences: qkv [B*L,3d] (q|k|v), out [B*Lq,d]; queries
 * are the last Lq rows of each sequence (Lq == L: all). */
int sf_mha_f32(const float* qkv, float* out, int B, int L, int Lq, int d_model, int num_heads, void* stream);

/* Fused self-attention block of one TransformerEncoderLayer (split-bf16 MFMA projection, f32 attention):
 * att[B*Lq,d] = MHA(LN?(x)) before out_proj; x [B*L,d]; in_proj_w [3d,d], in_proj_b [3d] (torch packed q|k|v);
 * ln_gamma/ln_beta may both be NULL.  Needs L <= 64, d % 64 == 0, head_dim in {16,32,48,64}. */
int sf_qkv_attention_f32(const float* x, const float* ln_gamma, const float* ln_beta, float ln_eps,
                         const float* in_proj_w, const float* in_proj_b, float* out, int B, int L, int Lq, int d_model,
                         int num_heads, void* stream);

/* nn.LSTM single step pointwise part (predictor.py:116-117), gates [R,4H] (i,f,g,o) pre-summed. */
int sf_lstm_pointwise_f32(const float* gates, const float* c_prev, float* h_out, float* c_out, int R, int H,
                          void* stream);

/* StoSAVi._sample_dist (savi.py:355-365): out = mu (+ noise*exp(logvar/2)); noise may be NULL. */
int sf_sample_dist_f32(const float* dist, const float* noise, float* out, int R, int D, void* stream);

/* F.interpolate(bilinear, align_corners=False) on R planes (steve.py:230-238). */
int sf_bilinear_resize_f32(const float* in, float* out, long long R, int Hi, int Wi, int Ho, int Wo,
                           void* stream);

/* ---- STEVE image side (row N2, second half): dVAE blocks -------------------------------------- */
/* Conv2dBlock's normalisation (steve_utils.py:124-126): F.group_norm(x, 1, gamma, beta, eps) over (C,H,W) per
 * sample, then ReLU when `relu`; NHWC x [F,H,W,C].  pixel_shuffle == 2 also applies the nn.PixelShuffle(2) that
 * follows the block in dVAE.py:44,49 (y [F,2H,2W,C/4]); 1 = none. */
size_t sf_groupnorm1_workspace_bytes(int F);
int sf_groupnorm1_nhwc_f32(const float* x, const float* gamma, const float* beta, float* y, int F, int H, int W, int C,
                           float eps, int relu, int pixel_shuffle, void* ws, size_t ws_bytes, void* stream);
/* Adjoint of the call above (row N1: the dVAE under autograd, dVAE.py:113-139).  x is the INPUT of the forward call (the
 * statistics and the ReLU gate are recomputed from it), dy the gradient of its output ([F,H,W,C], or [F,2H,2W,C/4] with
 * pixel_shuffle 2); writes dx [F,H,W,C], dgamma [C], dbeta [C]. */
size_t sf_groupnorm1_bwd_workspace_bytes(int F, int H, int W, int C);
int sf_groupnorm1_nhwc_bwd_f32(const float* x, const float* gamma, const float* beta, const float* dy, float* dx, float* dgamma,
                               float* dbeta, int F, int H, int W, int C, float eps, int relu, int pixel_shuffle, void* ws,
                               size_t ws_bytes, void* stream);

/* STEVE's slot-conditioned Transformer decoder (steve_transformer.py): attention core (bias-free projections are
 * plain sf_linear_f32 calls), token + position embedding, greedy token pick, token cross-entropy. */
int sf_slate_attention_f32(const float* q, const float* k, const float* v, float* out, int ldq, int ldk, int ldv, int ldo,
                           int B, int Lq, int Lk, int num_heads, int head_dim, int causal, void* stream);
/* Adjoint of sf_slate_attention_strided_f32 (row N1: the STEVE decoder's causal self-attention and slot cross-attention
 * under autograd, steve_transformer.py:61-116), flash style: `out` is the forward result, d_out its gradient; dq / dk / dv
 * have the layouts of q / k / v.  dq is cleared here and accumulated with float atomics.  head_dim even, <= 64; causal
 * needs Lq == Lk. */
size_t sf_slate_attention_bwd_workspace_bytes(int B, int Lq, int num_heads);
/* The same pair with dropout on the attention WEIGHTS (the nn.Dropout inside steve_transformer.py's MultiHeadAttention, active
 * in train() mode): masks are a pure function of (seed, sequence, head, query, key).  The forward also returns the row
 * log-sum-exp lse [B][H][Lq], which the backward takes instead of recomputing it (lse == NULL: recompute). */
int sf_slate_attention_train_fwd_f32(const float* q, const float* k, const float* v, float* out, float* lse, int ldq, int ldk,
                                     int ldv, int ldo, long long q_bs, long long k_bs, long long v_bs, long long o_bs, int B,
                                     int Lq, int Lk, int num_heads, int head_dim, int causal, float dropout_p,
                                     unsigned long long seed, void* stream);
int sf_slate_attention_train_bwd_f32(const float* q, const float* k, const float* v, const float* out, const float* d_out,
                                     const float* lse, float* dq, float* dk, float* dv, int ldq, int ldk, int ldv, int ldo,
                                     long long q_bs, long long k_bs, long long v_bs, long long o_bs, int B, int Lq, int Lk,
                                     int num_heads, int head_dim, int causal, float dropout_p, unsigned long long seed, void* ws,
                                     size_t ws_bytes, void* stream);
int sf_slate_attention_bwd_f32(const float* q, const float* k, const float* v, const float* out, const float* d_out, float* dq,
                               float* dk, float* dv, int ldq, int ldk, int ldv, int ldo, long long q_bs, long long k_bs,
                               long long v_bs, long long o_bs, int B, int Lq, int Lk, int num_heads, int head_dim, int causal,
                               void* ws, size_t ws_bytes, void* stream);
/* the same with explicit batch strides (floats), so that k/v may be a partially filled K/V cache */
int sf_slate_attention_strided_f32(const float* q, const float* k, const float* v, float* out, int ldq, int ldk, int ldv,
                                   int ldo, long long q_bs, long long k_bs, long long v_bs, long long o_bs, int B, int Lq,
                                   int Lk, int num_heads, int head_dim, int causal, void* stream);
int sf_embed_tokens_f32(const long long* idx, const float* tok_emb, const float* pos, float* out, int B, int L, int d,
                        void* stream);
int sf_argmax_rows_f32(const float* x, long long ld, long long* out, long long R, int V, void* stream);
int sf_cross_entropy_f32(const float* x, const long long* target, float* loss_rows, float* mean_out, long long R, int V,
                         void* stream);
/* y = softmax((x + add) * scale) per row (add may be NULL): the Gumbel-softmax relaxation of steve_utils.py:26-41 with
 * add = Gumbel noise, scale = 1/tau (steve_slotformer.py:97-98). */
int sf_softmax_rows_f32(const float* x, const float* add, float scale, float* y, long long R, int V, void* stream);
/* Backward of sf_cross_entropy_f32's mean loss w.r.t. the logits: dx = (softmax(x) - onehot(target)) * g[0] / R, with g the
 * upstream gradient as a DEVICE scalar (no host read); V <= 16384 (steve.py:341-344 under autograd). */
int sf_cross_entropy_bwd_f32(const float* x, const long long* target, const float* g, float* dx, long long R, int V, void* stream);
/* y = log_softmax(x) per row (the z_logits of dVAE.py:127). */
int sf_log_softmax_rows_f32(const float* x, float* y, long long R, int V, void* stream);
/* y[r, :] = softmax((x[r, :] + g[r, :]) * scale), g ~ Gumbel(0, 1) generated inside the kernel as a pure function of
 * (seed, r * V + j) -- g = -log(-log(u)), u = ((mix32(idx ^ s) >> 9) + 0.5) * 2^-23 -- so the noise of
 * steve_utils.py:30-35 never travels through memory (training only; R * V < 2^32). */
int sf_gumbel_softmax_rows_f32(const float* x, unsigned long long seed, float scale, float* y, long long R, int V, void* stream);
/* Adjoint of sf_softmax_rows_f32 w.r.t. x (and add) given its output y: dx = scale * y * (dy - sum(y * dy)) per row -- the
 * Gumbel-softmax relaxation of steve_utils.py:26-41 under autograd. */
int sf_softmax_rows_bwd_f32(const float* y, const float* dy, float scale, float* dx, long long R, int V, void* stream);

/* K/V-cached greedy generation of the dVAE token grid (STEVETransformerDecoder.generate, sample=False,
 * steve_transformer.py:305-333; used by STEVESlotFormer.decode, steve_slotformer.py:92-93).  One token per step; the
 * arithmetic of a step equals the reference's forward on the prefix.  All pointers device, torch layouts; wqkv =
 * cat(proj_q, proj_k, proj_v).weight [3d,d], wkv_c = cat(proj_k, proj_v).weight of the cross-attention [2d,d]. */
typedef struct {
  const float *ln1_g, *ln1_b, *wqkv, *wo;          /* self-attention */
  const float *ln2_g, *ln2_b, *wq_c, *wkv_c, *wo_c; /* encoder_decoder_attn (keys / values = the slots) */
  const float *ln3_g, *ln3_b, *w1, *b1, *w2, *b2;   /* FFN */
  int is_first;                                     /* first block: input normalised in place (steve_transformer.py:186) */
} sf_slate_block;

typedef struct {
  int d_model, num_heads, num_layers, vocab_size, num_slots, max_len;
  const float *in_proj_w, *in_proj_b, *tok_emb /*[V+1,d]*/, *pos_emb /*[max_len+1,d]*/, *lnf_g, *lnf_b, *head_w /*[V,d]*/;
  const sf_slate_block* blocks; /* HOST array [num_layers] */
} sf_slate_decoder;

size_t sf_slate_generate_workspace_bytes(const sf_slate_decoder* m, int B, int steps);
/* slots [B,N,d] -> tokens_out int64 [B,steps]; logits_out [B,steps,V] or NULL. */
int sf_slate_generate_f32(const sf_slate_decoder* m, const float* slots, int B, int steps, long long* tokens_out,
                          float* logits_out, void* ws, size_t ws_bytes, void* stream);

/* ---- whole-path engines ------------------------------------------------------------------ */

/* One nn.TransformerEncoderLayer (relu, batch_first); all pointers device, torch layouts. */
typedef struct {
  const float *norm1_g, *norm1_b, *in_proj_w, *in_proj_b, *out_proj_w, *out_proj_b;
  const float *norm2_g, *norm2_b, *lin1_w, *lin1_b, *lin2_w, *lin2_b;
  /* optional (NULL = absent): sf_pack_ffn_weights() copies of lin1_w / lin2_w.  When every layer of a rollouter
   * has them (d_model 256, 8 heads, ffn 1024, norm_first, split-bf16 mode) sf_rollout_f32 runs each layer as two
   * launches (attention + out-proj partials; FFN1 + FFN2) instead of four. */
  const void *lin1_packed, *lin2_packed;
  /* optional, same rule: sf_pack_attn_weights() copies of in_proj_w / out_proj_w */
  const void *attn_in_packed, *attn_out_packed;
  /* optional: sf_pack_layer_tok_weights() copy of all four matrices (the layer's fragments in consumption order) -- with it the rollout runs the layers
   * before the last as ONE token-stationary launch each (csrc/layer_tok.hip; sf_rollout_opts.layer_tok) */
  const void* tok_packed;
} sf_tfm_layer;

/* Pre-split attention weights (d_model 256, 8 heads) in MFMA-fragment order: in_packed needs
 * sf_attn_packed_bytes(d, 0) bytes, out_packed sf_attn_packed_bytes(d, 1). */
size_t sf_attn_packed_bytes(int d_model, int which);
int sf_pack_attn_weights(const float* in_proj_w, const float* out_proj_w, void* in_packed, void* out_packed, int d_model,
                         int num_heads, void* stream);

/* Pre-split (bf16 hi/lo) FFN weights in MFMA-fragment order; each output needs sf_ffn_packed_bytes() bytes. */
size_t sf_ffn_packed_bytes(int d_model, int ffn);
int sf_pack_ffn_weights(const float* lin1_w, const float* lin2_w, void* lin1_packed, void* lin2_packed, int d_model,
                        int ffn, void* stream);

/* FFN half of a pre-LN nn.TransformerEncoderLayer (d_model 256, ffn 1024; slotformer.py:72-80) on M rows whose input arrives
 * as 4 partial sums ap [4][M][256] (ap_stride floats apart; the head-pair partials of the layer's attention launch):
 *   x2 = ((ap0 + ap1) + ap2) + ap3;   y = x2 + lin2(relu(lin1(LN2(x2))))
 * written as the four hidden-chunk partials xp [4][M][256] with y = ((xp0 + xp1) + xp2) + xp3 (the next layer's attention
 * launch sums them while loading).  rows_per_wg: 32 / 64 / 128 rows per workgroup (0 = default); every choice gives the
 * same bits.  Needs w->lin1_packed / lin2_packed. */
int sf_ffn_chunk_partials_f32(const sf_tfm_layer* w, const float* ap, long long ap_stride, float* xp, long long xp_stride, int M,
                              int ffn, int rows_per_wg, void* stream);

/* Attention half of the same layer on B sequences of L <= 64 tokens, x [B][L][256]: the last Lq rows of every sequence of
 *   x2 = x + out_proj(MHA(LN1(x))) + b_o        (8 heads of 32)
 * heads_per_wg = 8: one workgroup per sequence runs all heads, out [2][B*Lq][256]: out[0] = x2, out[1] scratch;  heads_per_wg = 2: one workgroup per
 * (head pair, sequence), out [4][B*Lq][256] = the four head-pair partials with x2 = ((p0 + p1) + p2) + p3 -- the input format
 * of sf_ffn_chunk_partials_f32.  Both forms give the same bits.  Needs w->attn_in_packed / attn_out_packed. */
int sf_attn_block_f32(const sf_tfm_layer* w, const float* x, float* out, int B, int L, int Lq, int heads_per_wg, void* stream);
/* The same block in its row-tile form (sf_rollout_opts.attn_qkv_rows = 128; csrc/attn_rows.hip): LN1 + q|k|v on 128-row tiles of the
 * whole batch, then one workgroup per sequence (wave = head) + out-projection.  out [B*Lq][256] = x2;  planes: scratch of
 * sf_attn_rows_planes_bytes(B) bytes (q, k, v^T of every (sequence, head) as split-bf16 MFMA-fragment planes; cleared by the call). */
/* The FFN block y = x2 + lin2(relu(lin1(LN2(x2)))) on finished rows x2 [M][256] in its row-tile form (sf_rollout_opts.ffn_tile). */
int sf_ffn_block_rows_f32(const sf_tfm_layer* w, const float* x2, float* y, int M, int ffn, void* stream);
/* Whole pre-LN layers  y = x2 + lin2(relu(lin1(LN2(x2)))),  x2 = x + out_proj(MHA(LN1(x))) + b_o  on B sequences of L <= 96 tokens, x, y [B][L][256],
 * in the token-stationary form (csrc/layer_tok.hip): a 128-token workgroup owns whole sequences (as many as keep the keys of a wave's rows inside three
 * 32-key blocks: three of 42 tokens, one of 50 or of 65..96), a wave 32 tokens; every product of a layer keeps its
 * activations in registers (the accumulator layout of one product is the B operand of the next), the weight fragments stream global -> LDS once per
 * workgroup, only a head's keys / values cross waves; `nl` (1..8) consecutive layers w[0..nl) run in ONE launch (the rows never leave the registers
 * between them).  tok_packed: sf_pack_layer_tok_weights copy of a layer (sf_layer_tok_packed_bytes() bytes: its four matrices as fragments in consumption
 * order + its eight vectors).  A sequence's result does not depend on the other sequences of the call; it may differ in the last bits with its position
 * modulo the sequences per workgroup. */
size_t sf_layer_tok_packed_bytes(void);
int sf_pack_layer_tok_weights(const sf_tfm_layer* w, void* packed, int d_model, int num_heads, int ffn, void* stream);
int sf_layer_tok_block_f32(const sf_tfm_layer* w, int nl, const float* x, float* y, int B, int L, void* stream);
int sf_debug_read_ts_layer_tok(long long* out16);   /* wall-clock stamps (10 ns) of workgroup 0 with SF_DBG=lt */
size_t sf_attn_rows_planes_bytes(int B);
int sf_attn_block_rows_f32(const sf_tfm_layer* w, const float* x, float* out, void* planes, int B, int L, int Lq, void* stream);

/* SlotRollouter / SingleStepSlotRollouter (slotformer.py:48-134, single_step_slotformer.py:6-90). */
typedef struct {
  int num_slots, slot_size, d_model, num_layers, num_heads, ffn_dim, norm_first;
  int window_len;   /* history_len (sliding window) or cond_len (single-step growing window) */
  int single_step;  /* 0: SlotRollouter, 1: SingleStepSlotRollouter */
  const float *in_proj_w, *in_proj_b, *out_proj_w, *out_proj_b;
  const float* pe_tok;          /* [window_len*num_slots, d_model]: enc_t_pe per slot (+ enc_slots_pe) */
  const sf_tfm_layer* layers;   /* HOST array [num_layers] */
  /* optional (NULL = absent): sf_pack_linear_weights() copies of in_proj_w [d_model, slot_size] and out_proj_w
   * [slot_size, d_model].  With them (slot_size 128, d_model 256, packed FFN weights in every layer) a rollout step
   * projects only the newly predicted frame (the in-projections of older frames are cached in the workspace) and
   * out-proj + in-proj run as one launch. */
  const void *in_proj_packed, *out_proj_packed;
} sf_rollouter;

/* nn.Linear weight W [N, K] -> split-bf16 copy in MFMA-fragment order (N % 32 == 0, K % 16 == 0);
 * sf_packed_linear_bytes(N, K) bytes. */
size_t sf_packed_linear_bytes(int N, int K);
int sf_pack_linear_weights(const float* w, void* packed, int N, int K, void* stream);

size_t sf_rollout_workspace_bytes(const sf_rollouter* m, int B);
/* slots: [B, T_total, N, C]; the first n_in frames hold the burn-in slots (n_in = window_len, or 1
 * for single_step); frames n_in .. n_in+pred_len-1 are written.  T_total >= n_in + pred_len. */
int sf_rollout_f32(const sf_rollouter* m, float* slots, int B, int T_total, int pred_len, void* ws,
                   size_t ws_bytes, void* stream);

/* Per-call options of a rollout: they apply to THIS call on the calling thread only (thread-local inside the library; the
 * reference drives forward() from one host thread per GPU, base_slots/extract_slots.py:128), whereas sf_set_precision /
 * sf_set_seam_fused / sf_set_ffn_rows64 set process-wide DEFAULTS.  Every choice is bit-identical except `precision` and `layer_tok` (see there). */
typedef struct {
  int precision;    /* -1: default; 0 exact f32, 1 split-bf16, 2 single-pass bf16 (= sf_rollout_bf16), 3 single-pass fp16 (a
                     * measurement probe: the linear layers on one fp16 MFMA per product, profiles/r03_probes.txt) */
  int seam_fused;   /* -1: default; 0 / 1: seam launches off / on */
  int ffn_rows;     /* 0: default; 32 / 64 / 128 rows per workgroup of the chunk-partial FFN launches */
  int attn_heads_per_wg; /* 0: default (2: one workgroup per head pair and video, four partial outputs summed by the FFN launch);
                          * 8: one workgroup per video runs all heads and writes finished rows -- fewer, longer workgroups with
                          * a third of the bytes through a CU: the throughput form (the same bits as the default) */
  int attn_qkv_rows; /* 0: default (off); 128: the attention block as TWO launches -- LN1 + q|k|v on 128-row tiles of the whole
                      * batch (no padding of a video to 64 rows, one weight load per 128 rows), then one workgroup per video
                      * whose eight waves run the eight heads side by side + the out-projection: finished rows like
                      * attn_heads_per_wg = 8, the same bits, less CU time per row (csrc/attn_rows.hip) */
  int ffn_tile;      /* 0: default (off); 1: behind an attention block that writes finished rows (attn_heads_per_wg = 8 or attn_qkv_rows),
                      * the FFN block of the layers before the last as ONE workgroup per 64-row tile that runs all four hidden chunks
                      * on one ingest / LayerNorm with streamed weight fragments and writes finished rows (csrc/ffn_tile.hip) instead of
                      * one workgroup per (tile, hidden chunk) and four partial tensors: the same bits, half the CU time;
                      * 2 (with attn_qkv_rows): that launch also runs LN1 + q|k|v of the NEXT layer on its tile -- the rows never leave
                      * the workgroup, the next attention block is its core launch alone (one launch less per layer, the same bits) */
  int cus_available; /* 0: the whole chip; else the number of CUs the call's stream may use (its CU mask).  Seam launches hand rows over inside
                      * a grid and need every workgroup of it resident at once: they are used only when the grid fits min(160, cus_available) */
  int layer_tok;     /* 0: the process default (sf_set_layer_tok / SF_LAYER_TOK=1; OFF unless set); 1 / -1: on / off.  On (and every layer before the last has tok_packed, windows
                      * of <= 96 tokens; 65..96 -- the reference's Physion window of 15 frames x 6 slots -- outside the pipeline's units): those layers run as ONE token-stationary launch each (csrc/layer_tok.hip) -- a 128-token workgroup owns whole
                      * videos, every product of the layer keeps its activations in registers -- instead of an attention-core launch per video plus an
                      * FFN + q|k|v launch per 64-row tile; the row-pruned last layer keeps the row-tile forms.  Not bit-identical to the other forms
                      * (one accumulator per output block instead of per-chunk partial sums): 1e-6-level differences per layer */
} sf_rollout_opts;
int sf_set_layer_tok(int on);   /* process default of sf_rollout_opts.layer_tok == 0 */
int sf_get_layer_tok(void);
int sf_rollout_opts_f32(const sf_rollouter* m, float* slots, int B, int T_total, int pred_len, void* ws, size_t ws_bytes,
                        void* stream, const sf_rollout_opts* opts); /* opts == NULL: sf_rollout_f32 */
/* 1 when sf_rollout_f32 runs this model's Transformer layers as the fused per-video / per-row launches (d_model 256, 8 heads,
 * ffn 1024, window <= 64 tokens, packed weights, split-bf16 mode): a video's result then does not depend on the batch it is in -- bit for bit in the
 * row-tile / latency forms; with layer_tok on, to ~1e-6 per layer (a video's last bits depend on its position modulo the videos of a workgroup) */
int sf_rollout_is_fused(const sf_rollouter* m);
/* 1 when the layers before the last can run as token-stationary launches (sf_rollout_opts.layer_tok): fused-layer path, tok_packed on those layers,
 * every window of the rollout within the kernel's limits */
int sf_rollout_tok_ok(const sf_rollouter* m);
/* 1 when sf_rollout_f32 would run seam launches for this model / batch with the calling thread's defaults */
int sf_rollout_uses_seam(const sf_rollouter* m, int B);
/* ... with the options of the call in question (NULL: the thread's defaults): a caller on a CU-masked stream passes its cus_available */
int sf_rollout_uses_seam_opts(const sf_rollouter* m, int B, const sf_rollout_opts* opts);

/* ---- differentiable building blocks for the slot-level layers --------------------------------
 * (predictor, kernel distribution: savi.py:190-200, predictor.py:47-73).  Backward of y = act(x W^T + b): dW [N,K],
 * db [N] (or NULL), dx [M,K] (or NULL); with relu != 0, dy is first masked in place by y > 0.  N, K multiples of 64. */
size_t sf_linear_bwd_workspace_bytes(long long M, int N, int K);
int sf_linear_bwd_f32(const float* x, const float* W, const float* y, float* dy, float* dx, float* dW, float* db, long long M,
                      int N, int K, int relu, void* ws, size_t ws_bytes, void* stream);
/* Attention core of nn.MultiheadAttention under autograd (predictor.py:33-38 in train mode): qkv [B*L, 3d] (q|k|v) -> ctx
 * [B*L, d]; dropout on the softmax weights with masks that are a pure function of (seed, element); the backward call
 * recomputes the probabilities.  L <= 128, head_dim <= 64. */
int sf_mha_train_fwd_f32(const float* qkv, float* ctx, int B, int L, int d_model, int num_heads, float dropout_p,
                         unsigned long long seed, void* stream);
int sf_mha_train_bwd_f32(const float* qkv, const float* d_ctx, float* d_qkv, int B, int L, int d_model, int num_heads,
                         float dropout_p, unsigned long long seed, void* stream);
/* y = res + dropout(x) (nn.Dropout in train mode; res may be NULL; n % 4 == 0).  Its backward is the same call on dy. */
int sf_dropout_f32(const float* x, const float* res, float* y, long long n, float dropout_p, unsigned long long seed,
                   void* stream);
/* torch.optim.Adam (the reference's optimiser: slotformer_clevrer_params.py:16-19, stosavi_clevrer_params.py:14-17; no
 * weight decay, no amsgrad) over one flat fp32 bucket of n elements; step is the 1-based step count. */
int sf_adam_flat_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, int step, float lr,
                     float beta1, float beta2, float eps, void* stream);
/* Backward of nn.LayerNorm over the last dimension (D <= 1024, D % 4 == 0). */
size_t sf_layernorm_bwd_workspace_bytes(int D);
int sf_layernorm_bwd_f32(const float* x, const float* dy, const float* gamma, float* dx, float* dgamma, float* dbeta,
                         long long rows, int D, float eps, void* ws, size_t ws_bytes, void* stream);

/* ---- the SAVi image encoder under autograd ------------------------------------------------
 * conv stack + soft position embedding + per-pixel MLP (savi.py:220-250, 367-377; utils.py:52-63) with all weight
 * gradients.  Parameters in TORCH layouts (conv weights [Cout,Cin,5,5] for every layer; the node packs what its kernels
 * need into the workspace on every call, because an optimizer step changes them).  The reference encoder only: 64
 * channels, 5x5 kernels, a 64x64 feature map (first conv stride 2 at 128x128 input). */
typedef struct {
  int resolution, layers, channels, ks, hidden, out_channels;
  const float* conv_w[8];
  const float* conv_b[8];
  const float *pos_grid, *pos_w, *pos_b;   /* SoftPositionEmbed: grid [64*64, 4] (buffer), dense.weight [C,4], dense.bias [C] */
  const float *ln_g, *ln_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;   /* encoder_out_layer */
} sf_savi_features;

typedef struct {
  float* conv_w[8];
  float* conv_b[8];
  float *pos_w, *pos_b, *ln_g, *ln_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} sf_savi_features_grads;

size_t sf_savi_features_train_workspace_bytes(const sf_savi_features* m, int F);
/* frame f (NCHW [3,H,W]) at img + f*frame_stride floats -> out [F, 64*64, out_channels] */
int sf_savi_features_train_fwd_f32(const sf_savi_features* m, const float* img, long long frame_stride, int F, float* out,
                                   void* ws, size_t ws_bytes, void* stream);
int sf_savi_features_train_bwd_f32(const sf_savi_features* m, const float* img, long long frame_stride, const float* d_out,
                                   const sf_savi_features_grads* g, int F, void* ws, size_t ws_bytes, void* stream);

/* ---- training of the Slot-Attention module ------------------------------------------------
 * SlotAttention (savi.py:36-102) under autograd: parameters in torch layouts, gradients with the same shapes (written,
 * not accumulated).  The forward keeps its activations in the caller's workspace for the backward call.
 * slot_size 64 / 128 / 192 / 256, in_features and mlp_hidden multiples of 64, at most 8 slots and 8 iterations. */
typedef struct {
  int in_features, slot_size, mlp_hidden, num_slots;
  const float *norm_in_g, *norm_in_b, *wk, *wv;                  /* norm_inputs, project_k / project_v [D, in] */
  const float *q_ln_g, *q_ln_b, *wq;                             /* project_q = LayerNorm + Linear [D, D] */
  const float *gru_w_ih, *gru_w_hh, *gru_b_ih, *gru_b_hh;        /* nn.GRUCell [3D, D] x2, [3D] x2 */
  const float *mlp_ln_g, *mlp_ln_b, *mlp_w1, *mlp_b1, *mlp_w2, *mlp_b2;
  float eps;
} sf_slot_attention;

typedef struct {
  float *norm_in_g, *norm_in_b, *wk, *wv, *q_ln_g, *q_ln_b, *wq, *gru_w_ih, *gru_w_hh, *gru_b_ih, *gru_b_hh;
  float *mlp_ln_g, *mlp_ln_b, *mlp_w1, *mlp_b1, *mlp_w2, *mlp_b2;
} sf_slot_attention_grads;

size_t sf_slot_attention_train_workspace_bytes(const sf_slot_attention* m, int B, int HW, int iters);
/* inputs [B, HW, in_features], slots_in [B, N, D] -> slots_out [B, N, D] after `iters` iterations */
int sf_slot_attention_train_fwd_f32(const sf_slot_attention* m, const float* inputs, const float* slots_in, int B, int HW,
                                    int iters, float* slots_out, void* ws, size_t ws_bytes, void* stream);
/* d_slots_out [B, N, D] -> d_slots_in, parameter gradients and (if d_inputs != NULL) d_inputs [B, HW, in_features] */
int sf_slot_attention_train_bwd_f32(const sf_slot_attention* m, const float* inputs, const float* d_slots_out, float* d_inputs,
                                    float* d_slots_in, const sf_slot_attention_grads* g, int B, int HW, int iters, void* ws,
                                    size_t ws_bytes, void* stream);

/* ---- training of the rollout Transformer ------------------------------------
 * SlotFormer.forward in train mode (slotformer.py:263-282) -> SlotRollouter.forward (:85-126) under autograd, i.e. what
 * `loss.backward()` of calc_train_loss (:284-318) differentiates.  The forward keeps every activation of every rollout
 * step in the caller's workspace; the backward pass reads them, so the workspace must stay untouched in between.
 * Gradient buffers mirror the parameter leaves of sf_tfm_layer / sf_rollouter (same shapes) and are WRITTEN, not
 * accumulated.  Dropout (nn.TransformerEncoderLayer default p = 0.1 in train mode): masks are a pure function of
 * (seed, step, layer, site, element); pass the same dropout_p / seed to both calls.  Both rollouters (x holds
 * history_len burn-in frames, or ONE frame for single_step), norm_first layers, slot_size / d_model / ffn_dim multiples of 64, window of at most 128 tokens. */
typedef struct {
  float *norm1_g, *norm1_b, *in_proj_w, *in_proj_b, *out_proj_w, *out_proj_b;
  float *norm2_g, *norm2_b, *lin1_w, *lin1_b, *lin2_w, *lin2_b;
} sf_tfm_layer_grads;

typedef struct {
  float *in_proj_w, *in_proj_b, *out_proj_w, *out_proj_b;
  const sf_tfm_layer_grads* layers; /* HOST array [num_layers] */
  float* pe_tok; /* [window_len * num_slots, d_model] gradient of the folded position table sf_rollouter.pe_tok (the caller sums it
                    over slots / frames into enc_t_pe / enc_slots_pe, slotformer.py:103-109), or NULL: tables are fixed ('sin') */
} sf_rollouter_grads;

size_t sf_rollout_train_workspace_bytes(const sf_rollouter* m, int B, int pred_len);
/* x [B, history_len, N, C] burn-in slots -> pred [B, pred_len, N, C] */
int sf_rollout_train_fwd_f32(const sf_rollouter* m, const float* x, float* pred, int B, int pred_len, float dropout_p,
                             unsigned long long seed, void* ws, size_t ws_bytes, void* stream);
/* d_pred [B, pred_len, N, C] -> parameter gradients in *g and (if d_x != NULL) d_x [B, history_len, N, C] */
int sf_rollout_train_bwd_f32(const sf_rollouter* m, const float* d_pred, float* d_x, const sf_rollouter_grads* g, int B,
                             int pred_len, float dropout_p, unsigned long long seed, void* ws, size_t ws_bytes,
                             void* stream);

/* StoSAVi / STEVE encoder side (savi.py:177-250,295-322,367-416; steve.py:198-240). */
typedef struct {
  int resolution;      /* input H == W (64 or 128) */
  int enc_layers;      /* number of convs (<= 8) */
  int enc_channels[9]; /* enc_channels[0] = 3 */
  int enc_ks;
  int enc_out_channels, num_slots, slot_size, slot_mlp_size, num_iterations;
  const float* conv_w[8]; /* [0]: torch [C1,3,ks,ks]; [i>0]: packed [Cout,ks,ks,Cin] */
  const float* conv_b[8]; /* may be NULL */
  const float* pos_table; /* [64*64, C_last] */
  const float *enc_ln_g, *enc_ln_b, *enc_fc1_w, *enc_fc1_b, *enc_fc2_w, *enc_fc2_b;
  const float *sa_norm_in_g, *sa_norm_in_b, *sa_q_ln_g, *sa_q_ln_b, *sa_q_w;
  const float* sa_kv_w; /* [2*slot_size, enc_out_channels] = cat(project_k.weight, project_v.weight) */
  const float *gru_w_ih, *gru_w_hh, *gru_b_ih, *gru_b_hh; /* matrices transposed: [D,3D] */
  const float *mlp_ln_g, *mlp_ln_b, *mlp_w1, *mlp_b1, *mlp_w2, *mlp_b2; /* mlp_w1 [D,H], mlp_w2 [H,D] (transposed) */
  const float* init_latents; /* [N, D] */
  int kd_mode;               /* 0: none (STEVE), 1: Linear, 2: Linear-LN-ReLU-Linear (kernel_mlp) */
  const float *kd_w0, *kd_b0, *kd_ln_g, *kd_ln_b, *kd_w3, *kd_b3;
  int pred_type;             /* 0: ResidualMLPPredictor, 1: TransformerPredictor */
  int pred_rnn, pred_norm_first, pred_num_layers, pred_num_heads, pred_ffn_dim, pred_hidden;
  const float *pm_ln_g, *pm_ln_b, *pm_w0, *pm_b0, *pm_w2, *pm_b2;
  const sf_tfm_layer* pred_layers; /* HOST array */
  const float *lstm_w_ih, *lstm_w_hh, *lstm_b_ih, *lstm_b_hh, *proj_w, *proj_b;
  float sa_eps;
  const float* sa_q_w_t; /* optional: project_q weight transposed [in, out] (coalesced reads for the slot-update kernel, which
                          * then also emits the next iteration's q); NULL: q comes from a separate LN-fused GEMM launch */
  /* optional transposed ([in, out]) copies of the ResidualMLPPredictor weights and of the single-Linear kernel_dist layer:
   * with them (and sa_q_w_t) the per-frame slot prologue -- predictor, kernel distribution, sampling, q projection -- is ONE
   * launch (pred_type 0, no LSTM, kd_mode 1); NULL: separate LayerNorm / GEMM / sampling launches */
  const float *pm_w0_t, *pm_w2_t, *kd_w0_t;
  /* optional (slot size 128, slot MLP 256): sf_pack_linear_weights copies of the torch-layout GRUCell weight_ih / weight_hh
   * [3D, D], Slot-Attention mlp[1].weight [H, D], mlp[3].weight [D, H] and project_q[1].weight [D, D]; with all five the slot
   * update after every Slot-Attention iteration runs on the matrix cores (split-bf16, like the other bf16x3 kernels) */
  const void *sa_gru_ih_p, *sa_gru_hh_p, *sa_mlp_w1_p, *sa_mlp_w2_p, *sa_q_w_p;
  /* optional: Slot Attention with the key / value projections FOLDED away (project_k / project_v are bias-free Linears,
   * savi.py:44-45, so  k.q = x.(Wk^T q)  and  sum_p a v = Wv (sum_p a x)):  the attention runs on the normalised pixel features x
   * themselves -- half the bytes of k|v, and the [Wk;Wv] GEMM disappears -- with
   *   sa_fold_q_w   = Wk^T Wq   [C, D]  in place of project_q[1].weight  (its transposed [D, C] and packed copies beside it)
   *   sa_fold_gru_ih = W_ih Wv  [3D, C] in place of GRUCell.weight_ih    (transposed [C, 3D] and packed copies)
   * Used when all are given, enc_out_channels == slot_size == 128 (or 192 with enc_fc1_p / enc_fc2_p) and the split-bf16 mode is on. */
  const float *sa_fold_q_w, *sa_fold_q_w_t, *sa_fold_gru_ih_t;
  const void *sa_fold_q_w_p, *sa_fold_gru_ih_p;
  /* optional (pred_type 1, pre-LN, 4 heads, N <= 8; slot size / ffn / LSTM hidden 128 / 512 / 256 or 64 / 128 / 128):
   * HOST array of sf_pack_linear_weights copies -- [4 l + 0..3] = predictor layer l's in_proj_weight [3D, D], out_proj.weight
   * [D, D], linear1.weight [F, D], linear2.weight [D, F]; then, with the LSTM wrapper, weight_ih_l0 [4H, D], weight_hh_l0 [4H, H],
   * out_projector.weight [D, H].  With it the predictor step of a frame (predictor.py:20-44,76-135) is ONE launch instead of
   * 8 + 4 (split-bf16, like the other bf16x3 kernels); NULL: the unfused chain. */
  const void** pred_packed;
  /* optional (encoder_out_layer 64 -> 192 -> 192, i.e. STEVE on Physion): sf_pack_linear_weights copies of enc_fc1_w [192, 64] and
   * enc_fc2_w [192, 192]; with them and the sa_fold_* copies the per-pixel chain up to the normalised Slot-Attention inputs is one
   * launch (pixel_mlp.hip) and Slot Attention runs folded at width 192 as well */
  const void *enc_fc1_p, *enc_fc2_p;
  /* optional (NULL = absent), per conv i > 0 with 64 -> 64 channels, 5 x 5, on the 64 x 64 grid: sf_pack_conv_frag_weights copies of
   * conv_w[i].  With one, the conv runs on 4-row tiles with its weights streamed as MFMA fragments (csrc/conv_rows4.hip; split-bf16
   * mode) instead of the 2-row tile kernel that passes every tap's weights through LDS -- the same products in the same order. */
  const void* conv_w_frag[8];
  /* optional (slot size 128; pred_type 0 without LSTM, kd_mode 1): sf_pack_linear_weights copies of the ResidualMLPPredictor's mlp[0].weight
   * [2D, D] and mlp[2].weight [D, 2D] and of the kernel_dist Linear [2D, D].  With them and the matrix-core slot update (sa_*_p above), the slot
   * prologue of time step t + 1 -- predictor, kernel distribution, sampling, first q (savi.py:393-402) -- runs at the tail of step t's last slot
   * update instead of as its own launch (split-bf16 products, like the update's). */
  const void *pm_w0_p, *pm_w2_p, *kd_w0_p;
} sf_savi_encoder;

size_t sf_savi_encode_workspace_bytes(const sf_savi_encoder* m, int B);
/* img [B,T,3,H,W]; noise [B,T,N,D] or NULL (deterministic / STEVE); prev_slots [B,N,D] or NULL
 * (NULL: frame 0 starts from init_latents and the LSTM state is reset, savi.py:474-475);
 * lstm_h/lstm_c [B*N, pred_hidden] in/out (used iff pred_rnn; state_valid == 0: zeros);
 * outputs: post_slots [B,T,N,D], kernel_dist [B,T,N,2D] or NULL, attn [B,T,N,64*64] or NULL. */
int sf_savi_encode_f32(const sf_savi_encoder* m, const float* img, const float* noise, const float* prev_slots,
                       float* lstm_h, float* lstm_c, int state_valid, float* post_slots, float* kernel_dist,
                       float* attn, int B, int T, void* ws, size_t ws_bytes, void* stream);

/* The CNN stack alone (savi.py:231-244,367-371: convs + soft position embedding) for time steps [t0, t1) of every
 * video: feat [t1-t0][B][64*64][C_last] channels-last.  It does not depend on the slots, so it may run ahead of the
 * encode on another stream; sf_savi_encode_pre_f32 is sf_savi_encode_f32 with the features of the first n_pre time
 * steps taken from feat_pre (bench.py computes part of the NEXT batch's convolutions on the rollout stream's CUs while
 * that stream would otherwise idle). */
size_t sf_savi_cnn_workspace_bytes(const sf_savi_encoder* m, int B);
int sf_savi_cnn_f32(const sf_savi_encoder* m, const float* img, int B, int T, int t0, int t1, float* feat, void* ws,
                    size_t ws_bytes, void* stream);
int sf_savi_encode_pre_f32(const sf_savi_encoder* m, const float* img, const float* feat_pre, int n_pre, const float* noise,
                           const float* prev_slots, float* lstm_h, float* lstm_c, int state_valid, float* post_slots,
                           float* kernel_dist, float* attn, int B, int T, void* ws, size_t ws_bytes, void* stream);
/* The same as TWO branches: `stream` runs the image features (CNN + per-pixel chain) of all T steps back to back, `side_stream` follows one
 * step behind with the slot branches (prologue, Slot-Attention iterations, slot updates), ordered by events; `stream` continues behind the
 * last slot update.  Captured into a hipGraph the two are parallel branches.  side_stream NULL = sf_savi_encode_pre_f32.  Same bits.
 * Workspace: sf_savi_encode_fork_workspace_bytes(m, B, T) (the Slot-Attention inputs of all T steps stay resident). */
size_t sf_savi_encode_fork_workspace_bytes(const sf_savi_encoder* m, int B, int T);
/* Workspace of the one-stream encode (side_stream NULL) that runs the 64 -> 64 convolutions of ALL T time steps as one launch per layer (three
 * buffers of B * T frames on top of sf_savi_encode_workspace_bytes): sf_savi_encode_fork_f32 takes that form when it is handed this much and the
 * encoder has fragment weights on every layer behind the first (B <= 32, T >= 2, split-bf16); the same bits as the step-by-step order. */
size_t sf_savi_encode_batched_workspace_bytes(const sf_savi_encoder* m, int B, int T);
/* Where the slot prologue of time step t + 1 runs (process-wide; default 1).  1: with the packed predictor / kernel-distribution
 * copies of sf_savi_encoder (pm_w0_p, pm_w2_p, kd_w0_p) and the matrix-core slot update, at the tail of step t's last slot update -- one launch fewer per
 * time step, its products on the split-bf16 matrix path like the update's; 0: as its own launch on every step (fp32 thread-per-output products).  The
 * two agree to split-bf16 rounding (~1e-6 relative), not bit for bit; step 0 of a call always takes the stand-alone launch. */
int sf_set_encode_fuse_next(int on);
int sf_get_encode_fuse_next(void);
/* The slot branch of a batched encode (all T steps' convolutions per launch, sf_savi_encode_batched_workspace_bytes) as ONE video-stationary launch
 * (csrc/slot_chain.hip; savi.py:76-100, 393-402): one workgroup per video walks the T steps x num_iterations Slot-Attention iterations, their slot
 * updates and the per-step prologues on feature rows kept as bf16 hi | lo -- 7 launches per encode instead of ~60.  OPT-IN: process default 0
 * (sf_set_slot_chain(1)).  Measured (profiles/r06_probes.txt): 0.80 ms per batch of 32 videos x 6 frames on 32 CUs against 0.58 ms for
 * the per-iteration launches on the whole chip and 1.2 ms on a 128-CU partition -- fewer CU-ms, more latency; the default keeps the per-iteration launches.  Applies to the CLEVRER shape of the slot branch (slot size 128, slot MLP 256,
 * residual-MLP predictor, single-Linear kernel distribution, folded Slot Attention, up to 8 slots, HW a multiple of 256); everything else keeps the
 * per-iteration launches.  The two forms agree to split-bf16 rounding (the attention products are split-bf16 here, exact f32 there). */
int sf_set_slot_chain(int on);
int sf_get_slot_chain(void);
/* The Slot-Attention iterations of sf_savi_encode_* (folded form, slot size 128, 4096 pixels) on feature rows kept as bf16 hi | lo: logits and weighted sums as
 * split-bf16 v_mfma_f32_16x16x32_bf16 products (sa_attn_planes_kernel, csrc/slot_chain.hip) instead of exact-f32 16x16x4 MFMAs on f32 rows
 * (sa_attn_tile_kernel) -- a third of the matrix-pipe time, the same records, split-bf16 rounding apart (~5e-6).  Process default 1; 0: the f32 rows. */
int sf_set_slot_attn_planes(int on);
int sf_get_slot_attn_planes(void);
/* The per-pixel chain of the encoder (encoder_out_layer + SlotAttention.norm_inputs, savi.py:245-250, 66; 64 -> 128 -> 128 channels) in its pixel-stationary
 * form (pixel_feat_tok_kernel, csrc/pixel_mlp.hip: a wave owns 32 pixels for the whole chain, activations in registers, weights as fragments in LDS, no
 * barrier behind the prologue).  Process default 1; 0: the tile kernels.  The two agree to split-bf16 rounding (different summation order). */
int sf_set_pixel_tok(int on);
int sf_get_pixel_tok(void);
/* The encode in two halves, for callers that overlap them (the batch pipeline: features on the encode lane, the slot branch of a whole rollout unit in
 * front of its rollout).  sf_savi_chain_ok: 1 when both apply to this model at B videos x T frames (the conditions of sf_set_slot_chain above).
 *   sf_savi_features_planes_f32: CNN + encoder_out_layer + SlotAttention.norm_inputs (savi.py:231-250, 66) of B x T frames -> planes [T][B][64 * 64] rows
 *     of 512 B (bf16 hi | lo of the 128 channels; sf_savi_planes_bytes); workspace sf_savi_features_workspace_bytes.
 *   sf_savi_slots_chain_f32: the per-step chain of StoSAVi.encode (savi.py:393-416) with its Slot-Attention iterations (:76-100) for NB batches of B
 *     videos from planes [NB][T][B][64 * 64][512 B]: noise NULL or [NB * B][T][N][D], prev_slots NULL or [NB * B][N][D]; video v's slots of step t ->
 *     post + v * post_bs + t * N * D (post_bs in floats: a [NB * B][T + H][N][D] rollout buffer takes them in place); kernel_dist NULL or
 *     [NB * B][T][N][2 D]; attn NULL or [NB * B][T][N][64 * 64]; workspace sf_savi_slots_chain_workspace_bytes(m, NB * B).
 * The two in sequence are what sf_savi_encode_f32 runs when it is given sf_savi_encode_batched_workspace_bytes: the same bits. */
int sf_savi_chain_ok(const sf_savi_encoder* m, int B, int T);
size_t sf_savi_planes_bytes(const sf_savi_encoder* m, int B, int T);
size_t sf_savi_features_workspace_bytes(const sf_savi_encoder* m, int B, int T);
int sf_savi_features_planes_f32(const sf_savi_encoder* m, const float* img, int B, int T, void* planes, void* ws, size_t ws_bytes, void* stream);
size_t sf_savi_slots_chain_workspace_bytes(const sf_savi_encoder* m, int videos);
int sf_savi_slots_chain_f32(const sf_savi_encoder* m, const void* planes, const float* noise, const float* prev_slots, float* post, long long post_bs,
                            float* kernel_dist, float* attn, int NB, int B, int T, void* ws, size_t ws_bytes, void* stream);
int sf_savi_encode_fork_f32(const sf_savi_encoder* m, const float* img, const float* feat_pre, int n_pre, const float* noise,
                            const float* prev_slots, float* lstm_h, float* lstm_c, int state_valid, float* post_slots,
                            float* kernel_dist, float* attn, int B, int T, void* ws, size_t ws_bytes, void* stream, void* side_stream);

/* StoSAVi.decode (savi.py:504-525): spatial broadcast + position embedding -> transposed-conv stack -> 1x1 conv
 * -> softmax-over-slots recombination. */
typedef struct {
  int resolution;      /* output H == W */
  int dec_layers;      /* number of transposed convs (<= 8) */
  int dec_channels[9]; /* dec_channels[0] == slot_size */
  int dec_strides[8];
  int dec_ks, dec_res; /* kernel size; broadcast resolution (square) */
  int num_slots, slot_size;
  const float* deconv_w[8]; /* packed [Cout,ks,ks,Cin] */
  const float* deconv_b[8]; /* may be NULL */
  const float *out_w, *out_b; /* final 1x1 conv: [4, C_last], [4] */
  const float* pos_table;     /* [dec_res*dec_res, slot_size] */
  /* optional (NULL = absent), stride-1 layers only: deconv_w spatially flipped ([Cout][ks-1-ky][ks-1-kx][Cin]).  A
   * stride-1 transposed convolution is an ordinary convolution with the flipped kernel, which lets those layers run on
   * the encoder's halo-resident 5x5 kernel (64 -> 64 channels, 64 pixels wide). */
  const float* deconv_w_flipped[8];
  /* optional (NULL = absent), per 64 -> 64 channel 5 x 5 stride-2 layer: sf_pack_deconv_frag_weights copies of deconv_w[i] (split-bf16,
   * consumption order) for the parity-class kernel with streamed weights (csrc/deconv_s2.hip; split-bf16 mode, input 16 / 32 / 64 wide).
   * On the last layer that kernel also applies the 1x1 head, and the [F*N, H, W, 64] activation never reaches memory. */
  const void* deconv_w_frag[8];
  /* optional (NULL = absent): the FIRST layer (5 x 5, stride 2) on its broadcast input.  Its input is slot + pos_table[p] (savi.py:512-517),
   * so  out[r, p] = relu( (sum of the taps valid at p) . slot[r] + const[p] ):  l0_weff [25 * C1, slot_size] holds the tap sums of the 5 x 5
   * border / parity classes of an output pixel (class of a coordinate o = 2 o2 + par: par 0 -> {0: o2 == 0, 1: interior, 2: o2 == res - 1},
   * par 1 -> {3: o2 < res - 1, 4: o2 == res - 1}; row (vy * 5 + vx) * C1 + co), l0_posterm [(2 dec_res)^2, C1] the transposed convolution of
   * the position table + bias.  One [F*N, D] x [D, 25 C1] product and an expansion replace 98 % of the layer's multiplications. */
  const float* l0_weff;
  const float* l0_posterm;
} sf_savi_decoder;

size_t sf_savi_decode_workspace_bytes(const sf_savi_decoder* m, int F);
/* slots [F,N,D] -> recon_combined [F,3,H,W], recons [F,N,3,H,W] (or NULL), masks [F,N,1,H,W] (or NULL). */
int sf_savi_decode_f32(const sf_savi_decoder* m, const float* slots, float* recon_combined, float* recons,
                       float* masks, int F, void* ws, size_t ws_bytes, void* stream);
/* The same + the segmentation test_vp.py scores (postproc_mask of the decoded masks, video_prediction/test_vp.py:55-63 -> vp_utils.py:20-41):
 * seg_i64 / seg_u8 [F,H,W] (either or both NULL); recons / masks may be NULL -- a caller that only scores frames and segmentations never
 * materialises the per-slot tensors.  Workspace as sf_savi_decode_f32. */
int sf_savi_decode_seg_f32(const sf_savi_decoder* m, const float* slots, float* recon_combined, float* recons, float* masks,
                           long long* seg_i64, unsigned char* seg_u8, float fg_thre, int F, void* ws, size_t ws_bytes, void* stream);

/* Row N1: the same decoder under autograd, data gradient only (the image term of SlotFormer's training loss,
 * slotformer.py:313-326; the decoder is frozen there).  The forward keeps every layer output in the workspace; the
 * backward maps d_recon_combined [F,3,H,W] to d_slots [F,N,D].  deconv_w_bwd: HOST array [dec_layers] of device pointers
 * to sf_pack_conv_weight_f32(torch ConvTranspose2d weight [Cin,Cout,k,k]) = [Cin][k][k][Cout]: the adjoint of a
 * transposed convolution is a strided convolution with the same weights. */
typedef struct { /* gradients in torch layouts: ConvTranspose2d weight [Cin,Cout,k,k], head Conv2d [4,C,1,1], dense [D,4] */
  float* deconv_w[8];
  float* deconv_b[8];
  float *out_w, *out_b, *pos_w, *pos_b;
} sf_savi_decoder_grads;

size_t sf_savi_decode_train_workspace_bytes(const sf_savi_decoder* m, int F);
int sf_savi_decode_train_fwd_f32(const sf_savi_decoder* m, const float* slots, float* recon_combined, float* recons,
                                 float* masks, int F, void* ws, size_t ws_bytes, void* stream);
/* g != NULL (SAVi's own training, savi.py:527-538): also the decoder's parameter gradients; pos_grid [dec_res^2, 4] is
 * decoder_pos_embedding.grid.  Needs the reference decoder widths (64 channels after the first layer). */
int sf_savi_decode_train_bwd_f32(const sf_savi_decoder* m, const float* const* deconv_w_bwd, const float* d_recon,
                                 float* d_slots, const float* pos_grid, const sf_savi_decoder_grads* g, int F, void* ws,
                                 size_t ws_bytes, void* stream);


/* ---- K/V producer (SURVEY.md 8(b2) sf_kv_producer_*) ---------------------------------------- */
/* encoder_out_layer (savi.py:245-250: LN -> Linear -> ReLU -> Linear) followed by Slot Attention's
 * norm_inputs + project_k / project_v (savi.py:66-70): feat [M,C0] channels-last CNN features (position
 * embedding already added, savi.py:371-375) -> kv [M,2D] = (k | v).  kv_w = cat(project_k.weight,
 * project_v.weight) [2D,C1].  One fused kernel in split-bf16 mode at (C0,C1,2D) = (64,128,256); otherwise
 * three GEMMs through ws (sf_kv_producer_workspace_bytes). */
size_t sf_kv_producer_workspace_bytes(int M, int C1);
int sf_kv_producer_f32(const float* feat, const float* ln0_g, const float* ln0_b, const float* fc1_w,
                       const float* fc1_b, const float* fc2_w, const float* fc2_b, const float* ln1_g,
                       const float* ln1_b, const float* kv_w, float* kv, int M, int C0, int C1, int D, float ln_eps,
                       void* ws, size_t ws_bytes, void* stream);

/* ---- device-memory helpers and host-buffer twins (SURVEY.md 8(b2) "_host twin") -------------- */
/* For callers without a HIP binding of their own (plain C, ctypes + numpy): place the weights on the GPU
 * once with sf_device_alloc / sf_device_upload, fill the model structs with those device pointers, and call
 * the *_host twins, whose DATA buffers (everything that is not inside a model struct) are host pointers.
 * A twin has its device counterpart's signature; ws == NULL means "allocate the workspace for this call".
 * Twins are synchronous (the results are in host memory on return) and allocate/free staging memory. */
int sf_device_alloc(void** dev_out, size_t bytes);
int sf_device_free(void* dev);
int sf_device_upload(void* dev, const void* host, size_t bytes);
int sf_device_download(void* host, const void* dev, size_t bytes);
int sf_device_synchronize(void);
/* A HIP stream restricted to a subset of the CUs (hipExtStreamCreateWithCUMask; bit i of cu_mask = CU i,
 * n_words 32-bit words): partitions the GPU between the encode of batch i+1 and the rollout of batch i. */
int sf_stream_create_cu_mask(void** stream_out, const unsigned int* cu_mask, int n_words);
/* CUs a launch on `stream` may occupy: the popcount of the mask of a stream made by sf_stream_create_cu_mask, else the device's CU count.  The
 * persistent kernels (csrc/conv_ws.hip) launch one workgroup per CU of their stream. */
int sf_stream_cus(void* stream);
/* The CU count the library is to assume for launches issued on `stream` (cus <= 0: forget): for a stream that captures a graph which will be replayed
 * on a CU-masked stream. */
int sf_stream_set_cus(void* stream, int cus);
int sf_stream_destroy(void* stream);
/* One wave busy for `us` microseconds on `stream` (1..100000): two of them on two streams tell whether the streams share a hardware queue. */
int sf_debug_spin(int us, void* stream);
/* One wave that counts shader cycles (out2[0]) over `us` microseconds of the constant 100 MHz counter (out2[1] ticks): the clock the chip sustains
 * under whatever else is running (tools/clock_probe.py); out2: device buffer of two int64. */
int sf_debug_clock_probe(int us, long long* out2, void* stream);
int sf_slot_attn_iter_f32_host(const float* k_host, const float* v_host, int ld, long long batch_stride,
                               const float* q_host, float* part_num_host, float* part_den_host, float* attn_out_host,
                               int B, int HW, int N, int D, float scale, float eps, void* stream);
int sf_rollout_f32_host(const sf_rollouter* m, float* slots_host, int B, int T_total, int pred_len, void* ws,
                        size_t ws_bytes, void* stream);
/* Health check of the rollout's seam launches (the last FFN + step boundary of step s and the first attention of step s+1 run
 * in one grid and hand 7 rows per video over through memory, slotformer.py:121-124 -> :115): number of hand-offs that gave up
 * waiting since the library was loaded (synchronises the device).  Must be 0; SF_SEAM_FUSED=0 disables the seam launches. */
int sf_seam_timeouts(void);

/* Single-pass bf16 variant of sf_rollout_f32 (SURVEY.md 8(b2) `sf_rollout_bf16`; the reference's `--fp16` AMP,
 * scripts/train.py:84,105, and BASELINE.json's literal "bf16"): same arguments and storage (f32), every matrix product with
 * operands rounded to bf16 and f32 accumulation.  ~8e-3 relative on the 6+50 path of config C2 -- outside the 1e-3 parity
 * bar, so it is an option, never the default. */
int sf_rollout_bf16(const sf_rollouter* m, float* slots, int B, int T_total, int pred_len, void* ws, size_t ws_bytes,
                    void* stream);
int sf_rollout_bf16_host(const sf_rollouter* m, float* slots_host, int B, int T_total, int pred_len, void* ws,
                         size_t ws_bytes, void* stream);
/* host twins of the slot update (savi.py:95-100) and of the K/V producer (savi.py:245-250,66-70): data buffers on the host,
 * weights on the device */
int sf_slot_update_f32_host(const float* part_num_host, const float* part_den_host, int P, const float* slots_prev_host,
                            const float* gru_w_ih, const float* gru_w_hh, const float* gru_b_ih, const float* gru_b_hh,
                            const float* ln_g, const float* ln_b, const float* mlp_w1, const float* mlp_b1,
                            const float* mlp_w2, const float* mlp_b2, float* slots_out_host, int B, int N, int D, int H,
                            float ln_eps, void* stream);
int sf_kv_producer_f32_host(const float* feat_host, const float* ln0_g, const float* ln0_b, const float* fc1_w,
                            const float* fc1_b, const float* fc2_w, const float* fc2_b, const float* ln1_g,
                            const float* ln1_b, const float* kv_w, float* kv_host, int M, int C0, int C1, int D, float ln_eps,
                            void* ws, size_t ws_bytes, void* stream);
int sf_savi_encode_f32_host(const sf_savi_encoder* m, const float* img_host, const float* noise_host,
                            const float* prev_slots_host, float* lstm_h_host, float* lstm_c_host, int state_valid,
                            float* post_slots_host, float* kernel_dist_host, float* attn_host, int B, int T, void* ws,
                            size_t ws_bytes, void* stream);
int sf_savi_decode_f32_host(const sf_savi_decoder* m, const float* slots_host, float* recon_combined_host,
                            float* recons_host, float* masks_host, int F, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
