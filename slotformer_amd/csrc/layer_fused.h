// Prototypes of the two-launch Transformer layer (layer_fused.hip); needs sf_tfm_layer from the public header.
#pragma once
#include "../../include/slotformer_hip.h"
#include "sf_internal.h"

// xin: np_in (1 or 4) partial buffers [B*L, 256] xin_stride floats apart; ap: 8 head-partial buffers [B*Lq, 256]
int sf_attn_oproj_ex(const float* xin, long long xin_stride, int np_in, const sf_tfm_layer& w, float eps, float* ap,
                     long long ap_stride, int B, int L, int Lq, hipStream_t st);
// ap: 8 head partials [M, 256] -> xout [M, 256] (finished layer output); xp: scratch for the 4 hidden-chunk partials
// [4][M, 256]; counters: sf_ffn_tiles(M) ints, zero before the first launch (the kernel leaves them zero)
int sf_ffn_partial_ex(const float* ap, long long ap_stride, const sf_tfm_layer& w, float eps, float* xp,
                      long long xp_stride, float* xout, int* counters, int M, int ffn, hipStream_t st);
int sf_ffn_tiles(int M);
