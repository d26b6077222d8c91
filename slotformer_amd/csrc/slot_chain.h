// The video-stationary slot branch of the SAVi encode (slot_chain.hip) and the feature form it reads (pixel_mlp.hip).
#pragma once
#include "sf_internal.h"

// operands of the chain: the slot update's packed matrices (sf_pack_linear_weights) and vectors, and those of the NEXT-step prologue
struct SfChainWeights {
  const void *gru_ih_p, *gru_hh_p;
  const float *gru_b_ih, *gru_b_hh, *ln_g, *ln_b;
  const void* w1_p;
  const float* b1;
  const void* w2_p;
  const float* b2;
  const float *q_ln_g, *q_ln_b;
  const void* q_w_p;
  const float *pm_ln_g, *pm_ln_b;
  const void* pm_w0_p;
  const float* pm_b0;
  const void* pm_w2_p;
  const float* pm_b2;
  int pm_norm_first;
  const void* kd_w_p;
  const float* kd_b;
};

// slot size 128, slot MLP 256, HW a multiple of 256, at most 8 slots
bool sf_slot_chain_ok(int D, int H, int HW, int N);
// One launch, one workgroup per video of NB batches of B videos: T steps x iters Slot-Attention iterations with their slot updates and the per-step
// prologues (savi.py:76-100, 393-402).  feat_planes: [NB][T][B][HW] rows of 512 B (bf16 hi 128 | lo 128: sf_pixel_mlp_feat_planes_ex per batch).  On
// entry slotsA / q hold the sampled slots of step 0 and their project_q for the NB * B videos (sf_slot_prologue_ex); video v's slots of step t go to
// post + v * post_bs + t * N * 128; attn NULL or [NB * B][T][N][HW]; noise NULL or [NB * B][T][N][128]; kdist NULL or [NB * B][T][N][256] (steps >= 1).
int sf_slot_chain_ex(const void* feat_planes, int NB, int B, int T, int HW, int N, int iters, float scale, float eps, float ln_eps, float* slotsA, float* slotsB,
                     float* lat, float* q, float* post, long long post_bs, float* attn, const float* noise, float* kdist, const SfChainWeights* w,
                     hipStream_t st);
// One Slot-Attention iteration of B frames as a batch-wide launch on rows of 512 B (bf16 hi | lo): split-bf16 16x16x32 MFMAs instead of the exact-f32 ones of
// sa_attn_tile_kernel; the same records (slot size 128, HW a multiple of 512)
bool sf_slot_attn_planes_ok(int HW, int D, int N);
int sf_slot_attn_planes_ex(const void* planes, long long batch_stride_rows, const float* q, float* part_num, float* part_den, float* attn_out,
                           long long attn_batch_stride, int B, int HW, int N, float scale, float eps, hipStream_t st);
// encoder_out_layer + SlotAttention.norm_inputs (sf_pixel_mlp_feat_ex) with the result as bf16 hi | lo rows of 512 B: planes [M][256] bf16
int sf_pixel_mlp_feat_planes_ex(const float* x, const float* ln0_g, const float* ln0_b, const float* w1, const float* b1, const float* w2,
                                const float* b2, const float* ln1_g, const float* ln1_b, void* planes, int M, float eps, hipStream_t st);
