// Prototypes of the two-launch Transformer layer (layer_fused.hip); needs sf_tfm_layer from the public header.
#pragma once
#include "../../include/slotformer_hip.h"
#include "sf_internal.h"

// pred_step.hip: Transformer predictor (+ LSTM wrapper) of one frame in one launch; 1 = shape not covered
int sf_pred_step_ex(const float* prev, const sf_tfm_layer* layers, int nlayers, int heads, int ffn, int norm_first, const void* const* packed,
                    const float* lstm_b_ih, const float* lstm_b_hh, const float* proj_b, int hidden, float* lstm_h, float* lstm_c,
                    float* out, int B, int N, int D, float eps, hipStream_t st);

// x [B][L][256] -> ap: 8 head-partial buffers [B*Lq, 256]
int sf_attn_oproj_ex(const float* xin, const sf_tfm_layer& w, float eps, float* ap, long long ap_stride, int B, int L,
                     int Lq, hipStream_t st);
// layer 0 of a rollout step: x = ring[b][(f0 + r / nslots) % ring_frames][r % nslots] + pe[r]
int sf_attn_oproj_parts_ex(const float* xparts, long long xparts_stride, const sf_tfm_layer& w, float eps, float* ap,
                           long long ap_stride, int B, int L, int Lq, hipStream_t st);
// np: input partials per row -- 4 (head-pair partials of sf_attn_oproj_*_ex) or 1 (finished rows of sf_attn_all_*_ex)
int sf_ffn_parts_ex(const float* ap, long long ap_stride, const sf_tfm_layer& w, float eps, float* xp, long long xp_stride, int M,
                    int ffn, hipStream_t st, int np = 4);
// all 8 heads in ONE workgroup per video: x2 [B*Lq][256] = x + out_proj(MHA(LN1(x))) + b_o (finished rows, not partials)
int sf_attn_all_ex(const float* xin, const sf_tfm_layer& w, float eps, float* x2, int B, int L, int Lq, hipStream_t st);
int sf_attn_all_parts_ex(const float* xparts, long long xparts_stride, const sf_tfm_layer& w, float eps, float* x2, int B, int L,
                         int Lq, hipStream_t st);
int sf_attn_all_ring_ex(const float* ring, int ring_frames, int nslots, int f0, const float* pe, const sf_tfm_layer& w, float eps,
                        float* x2, int B, int L, int Lq, hipStream_t st);
int sf_attn_core_ex(const sf_tfm_layer& w, float* x2, void* planes, int B, int L, int Lq, hipStream_t st);
// FFN of layer w fused with LN1 + q|k|v of layer wn on 64-row tiles (ffn_tile.hip): planes + parked residual rows for sf_attn_core_ex
int sf_ffn_qkv_tile_ex(const float* x2, const sf_tfm_layer& w, const sf_tfm_layer& wn, float eps, float* xpark, void* planes, int B, int L, int Lq,
                       int ffn, hipStream_t st);
// row-tile form of the FFN block (ffn_tile.hip): x2 [M][256] finished rows -> y [M][256] finished rows, one workgroup per 64 rows
int sf_ffn_tile_ex(const float* x2, const sf_tfm_layer& w, float eps, float* y, int M, int ffn, hipStream_t st);
// token-stationary whole layers (layer_tok.hip): `nl` consecutive layers in one launch; mode 0: xin [B * L][256] rows; mode 1: the first is layer 0 of a
// rollout step, x = ring rows + position table
bool sf_layer_tok_ok(int L);
int sf_layer_tok_vpw(int L);   // videos per 128-token workgroup for sequences of L tokens (0: refused)
int sf_layer_tok_ex(int mode, const float* xin, const float* ring, int ring_frames, int nslots, int f0, const float* pe, const sf_tfm_layer* layers, int nl,
                    float eps, float* y, int B, int L, hipStream_t st);
// row-tile form (attn_rows.hip): q|k|v projection on 128-row tiles of the batch + one core / out-projection workgroup per video;
// mode 0: x [B][L][256], 1: four chunk partials, 2: ring rows + position table.  planes: sf_attn_rows_plane_bytes(B) bytes
size_t sf_attn_rows_plane_bytes(int B);
int sf_attn_rows_ex(int mode, const float* xin, long long x_batch_stride, long long xparts_stride, const float* pe, int f0,
                    int ring_frames, int nslots, const sf_tfm_layer& w, float eps, float* x2, void* planes, int B, int L, int Lq,
                    hipStream_t st);
int sf_attn_oproj_ring_ex(const float* ring, int ring_frames, int nslots, int f0, const float* pe, const sf_tfm_layer& w,
                          float eps, float* ap, long long ap_stride, int B, int L, int Lq, hipStream_t st);
// out-proj of the finished rows y [B*nslots, 256] -> slots frame `frame`; in-proj of those rows -> projection ring
bool sf_step_boundary_ok(int d, int slot_size);
int sf_step_boundary_ex(const float* y, const void* wout_packed, const float* b_out, const void* win_packed,
                        const float* b_in, float* slots, long long slots_bs, int frame, float* ring, int ring_frames,
                        int nslots, int B, hipStream_t st);
// ap: 8 head partials [M, 256] -> xout [M, 256] (finished layer output); xp: scratch for the 4 hidden-chunk partials
// [4][M, 256]; counters: sf_ffn_tiles(M) ints, zero before the first launch (the kernel leaves them zero)
int sf_ffn_partial_ex(const float* ap, long long ap_stride, const sf_tfm_layer& w, float eps, float* xp,
                      long long xp_stride, float* xout, int* counters, int M, int ffn, hipStream_t st, int np = 4);
int sf_ffn_tiles(int M);
// in-projection of the first n_frames frames of every video -> ring slots 0 .. n_frames-1 (same arithmetic as the step kernel)
int sf_ring_init_ex(const void* wout_packed, const float* b_out, const void* win_packed, const float* b_in, float* slots,
                    long long slots_bs, int n_frames, float* ring, int ring_frames, int nslots, int B, hipStream_t st);
// last layer of a rollout step (M = B * nslots rows): FFN + fused step boundary in one launch
int sf_ffn_boundary_ex(const float* ap, long long ap_stride, const sf_tfm_layer& w, float eps, float* xp, long long xp_stride,
                       int* counters, int ffn, const void* wout_packed, const float* b_out, const void* win_packed,
                       const float* b_in, float* slots, long long slots_bs, int frame, float* ring, int ring_frames, int nslots,
                       int B, hipStream_t st, int np = 4);
// ONE launch for the seam between two rollout steps: last-layer FFN + step boundary of step s (blocks [0, nffn)) and the
// layer-0 attention of step s+1 (one block per (head pair, video)), handed over per 32-row tile through seam_flags
int sf_seam_ex(const float* ap_ffn, long long pst_ffn, const sf_tfm_layer& wl, float eps, float* xp, long long xp_stride,
               int* counters, int ffn, const void* wout_packed, const float* b_out, const void* win_packed, const float* b_in,
               float* slots, long long slots_bs, int frame, float* ring, int ring_frames, int nslots, int B,
               const sf_tfm_layer& w0, int f0_next, const float* pe, float* ap_attn, long long pst_attn, int L, int Lq,
               unsigned* seam_flags, unsigned epoch, hipStream_t st);
int sf_seam_blocks(int B, int nslots);
bool sf_seam_window_ok(int L, int nslots);
