// Small fused kernels around the GEMM core: LayerNorm, multi-head self-attention for short
// sequences (L <= ~128 tokens: rollout window / slot predictor), LSTM cell pointwise part,
// stochastic-kernel sampling, row copies, weight packing, position-embedding table, bilinear
// mask resize.  gfx950 only.
#include "sf_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------
// LayerNorm over the last dim, one wave per row (nn.LayerNorm, biased variance).
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, SfRowMap xmap,
                                                        const float* __restrict__ g,
                                                        const float* __restrict__ b, float* __restrict__ y,
                                                        SfRowMap ymap, int rows, int D, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + sf_row_off(xmap, row);
  float* yr = y + sf_row_off(ymap, row);
  float s = 0.f;
  for (int k = lane; k < D; k += 64) s += xr[k];
  const float mean = sf_wave_sum(s) / (float)D;
  float v = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float d = xr[k] - mean;
    v += d * d;
  }
  const float rstd = 1.0f / sqrtf(sf_wave_sum(v) / (float)D + eps);
  for (int k = lane; k < D; k += 64) yr[k] = (xr[k] - mean) * rstd * g[k] + b[k];
}

int sf_layernorm_ex(const float* x, SfRowMap xmap, const float* g, const float* b, float* y, SfRowMap ymap,
                    int rows, int D, float eps, hipStream_t st) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, xmap, g, b, y, ymap, rows, D,
                     eps);
  SF_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------
// Multi-head self-attention for short sequences (nn.MultiheadAttention, no mask, eval).
// qkv: [B*L, 3d] packed (q|k|v), head h owns columns [h*HD, (h+1)*HD).  One workgroup per
// (head, batch); K and V of the head live in LDS; one thread per query row; two-pass softmax
// (max, then exp/sum) as torch computes it.  Queries may be restricted to the last Lq rows.
template <int HD>
__global__ __launch_bounds__(128) void mha_small_kernel(const float* __restrict__ qkv, int ld,
                                                        float* __restrict__ out, int ldo, int L, int Lq,
                                                        int d, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;            // [L][HD]
  float* Vs = smem + L * HD;   // [L][HD]
  const int h = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const float* base = qkv + (long long)b * L * ld + h * HD;
  for (int idx = t; idx < L * (HD / 4); idx += blockDim.x) {
    const int j = idx / (HD / 4), c = idx - j * (HD / 4);
    *(f32x4*)(Ks + j * HD + 4 * c) = *(const f32x4*)(base + (long long)j * ld + d + 4 * c);
    *(f32x4*)(Vs + j * HD + 4 * c) = *(const f32x4*)(base + (long long)j * ld + 2 * d + 4 * c);
  }
  __syncthreads();
  for (int iq = t; iq < Lq; iq += blockDim.x) {
    const int i = L - Lq + iq;
    float q[HD];
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      const f32x4 v = *(const f32x4*)(base + (long long)i * ld + 4 * c);
      q[4 * c] = v[0] * scale;
      q[4 * c + 1] = v[1] * scale;
      q[4 * c + 2] = v[2] * scale;
      q[4 * c + 3] = v[3] * scale;
    }
    float mx = -INFINITY;
    for (int j = 0; j < L; ++j) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) s = fmaf(q[c], Ks[j * HD + c], s);
      mx = fmaxf(mx, s);
    }
    float acc[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] = 0.f;
    float sum = 0.f;
    for (int j = 0; j < L; ++j) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) s = fmaf(q[c], Ks[j * HD + c], s);
      const float p = expf(s - mx);
      sum += p;
#pragma unroll
      for (int c = 0; c < HD; ++c) acc[c] = fmaf(p, Vs[j * HD + c], acc[c]);
    }
    const float inv = 1.0f / sum;
    float* o = out + ((long long)b * Lq + iq) * ldo + h * HD;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      f32x4 v = {acc[4 * c] * inv, acc[4 * c + 1] * inv, acc[4 * c + 2] * inv, acc[4 * c + 3] * inv};
      *(f32x4*)(o + 4 * c) = v;
    }
  }
}

// Workgroup-parallel variant used whenever Q/K/V + the score matrix of one (head, batch) fit in
// LDS: 256 threads share the L x L scores (QK^T), the row softmax and PV.  K rows are padded by
// one float so the column-parallel score pass is bank-conflict free.
template <int HD>
__global__ __launch_bounds__(256) void mha_tile_kernel(const float* __restrict__ qkv, int ld,
                                                       float* __restrict__ out, int ldo, int L, int Lq, int d,
                                                       float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KS = HD + 1;
  float* Qs = smem;                 // [Lq][HD]   (pre-scaled)
  float* Ks = Qs + Lq * HD;         // [L][HD+1]
  float* Vs = Ks + L * KS;          // [L][HD]
  float* Ss = Vs + L * HD;          // [Lq][L+1]
  float* inv = Ss + Lq * (L + 1);   // [Lq]
  const int h = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const int SS = L + 1;
  const float* base = qkv + (long long)b * L * ld + h * HD;
  for (int idx = t; idx < L * (HD / 4); idx += 256) {
    const int j = idx / (HD / 4), c = idx - j * (HD / 4);
    const f32x4 kk = *(const f32x4*)(base + (long long)j * ld + d + 4 * c);
    const f32x4 vv = *(const f32x4*)(base + (long long)j * ld + 2 * d + 4 * c);
    Ks[j * KS + 4 * c] = kk[0];
    Ks[j * KS + 4 * c + 1] = kk[1];
    Ks[j * KS + 4 * c + 2] = kk[2];
    Ks[j * KS + 4 * c + 3] = kk[3];
    *(f32x4*)(Vs + j * HD + 4 * c) = vv;
    if (j >= L - Lq) {
      const f32x4 qq = *(const f32x4*)(base + (long long)j * ld + 4 * c) * scale;
      *(f32x4*)(Qs + (j - (L - Lq)) * HD + 4 * c) = qq;
    }
  }
  __syncthreads();
  // scores: thread -> (query i, key j); lanes run over j (distinct padded K rows), Q row broadcast
  for (int e = t; e < Lq * L; e += 256) {
    const int i = e / L, j = e - i * L;
    const float* qr = Qs + i * HD;
    const float* kr = Ks + j * KS;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) s = fmaf(qr[c], kr[c], s);
    Ss[i * SS + j] = s;
  }
  __syncthreads();
  // row softmax (max, exp, sum) -- 4 lanes per row
  for (int i = t >> 2; i < Lq; i += 64) {
    const int sub = t & 3;
    float* row = Ss + i * SS;
    float mx = -INFINITY;
    for (int j = sub; j < L; j += 4) mx = fmaxf(mx, row[j]);
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    float sum = 0.f;
    for (int j = sub; j < L; j += 4) {
      const float p = expf(row[j] - mx);
      row[j] = p;
      sum += p;
    }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    if (sub == 0) inv[i] = 1.0f / sum;
  }
  __syncthreads();
  // O = P V: thread -> (query i, channel c); lanes run over c (contiguous V), P broadcast
  for (int e = t; e < Lq * HD; e += 256) {
    const int i = e / HD, c = e - i * HD;
    const float* pr = Ss + i * SS;
    float acc = 0.f;
    for (int j = 0; j < L; ++j) acc = fmaf(pr[j], Vs[j * HD + c], acc);
    out[((long long)b * Lq + i) * ldo + h * HD + c] = acc * inv[i];
  }
}

int sf_mha_ex(const float* qkv, float* out, int B, int L, int Lq, int d, int nheads, hipStream_t st) {
  if (B <= 0 || L <= 0) return 0;
  SF_REQUIRE(nheads > 0 && d % nheads == 0, "d_model must be divisible by num_heads");
  const int hd = d / nheads;
  SF_REQUIRE(hd == 16 || hd == 32 || hd == 48 || hd == 64, "head_dim must be 16/32/48/64");
  SF_REQUIRE(Lq >= 1 && Lq <= L && L <= 512, "bad sequence length");
  const float scale = 1.0f / sqrtf((float)hd);
  dim3 grid(nheads, B);
  const size_t lds_tile = ((size_t)Lq * hd + (size_t)L * (hd + 1) + (size_t)L * hd + (size_t)Lq * (L + 1) + Lq) *
                          sizeof(float);
  sf_prof_begin(SF_K_MHA, st, 4.0 * (double)B * nheads * Lq * L * hd);
  if (lds_tile <= 160 * 1024) {
#define MHA_LAUNCH(HD)                                                                                              \
  {                                                                                                                 \
    SF_TRY(sf_ensure_dyn_lds((const void*)mha_tile_kernel<HD>, (size_t)160 * 1024));                                   \
    hipLaunchKernelGGL(mha_tile_kernel<HD>, grid, dim3(256), lds_tile, st, qkv, 3 * d, out, d, L, Lq, d, scale);     \
  }
    switch (hd) {
      case 16: MHA_LAUNCH(16); break;
      case 32: MHA_LAUNCH(32); break;
      case 48: MHA_LAUNCH(48); break;
      default: MHA_LAUNCH(64); break;
    }
#undef MHA_LAUNCH
  } else {
    const size_t lds = (size_t)2 * L * hd * sizeof(float);
    SF_REQUIRE(lds <= 64 * 1024, "sequence too long for the short-sequence attention kernels");
    dim3 block(Lq <= 64 ? 64 : 128);
#define MHA_LAUNCH(HD) \
  hipLaunchKernelGGL(mha_small_kernel<HD>, grid, block, lds, st, qkv, 3 * d, out, d, L, Lq, d, scale)
    switch (hd) {
      case 16: MHA_LAUNCH(16); break;
      case 32: MHA_LAUNCH(32); break;
      case 48: MHA_LAUNCH(48); break;
      default: MHA_LAUNCH(64); break;
    }
#undef MHA_LAUNCH
  }
  sf_prof_end(SF_K_MHA, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------
// LSTM cell pointwise part (gate order i,f,g,o):  gates [R,4H] already = x W_ih^T + b_ih + h W_hh^T + b_hh
__global__ void lstm_pointwise_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev,
                                      float* __restrict__ h_out, float* __restrict__ c_out, int R, int H) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * H) return;
  const int r = idx / H, j = idx - r * H;
  const float* g = gates + (long long)r * 4 * H;
  const float i = sf_sigmoid(g[j]), f = sf_sigmoid(g[H + j]), gg = tanhf(g[2 * H + j]), o = sf_sigmoid(g[3 * H + j]);
  const float c = f * (c_prev ? c_prev[idx] : 0.f) + i * gg;
  c_out[idx] = c;
  h_out[idx] = o * tanhf(c);
}

int sf_lstm_pointwise_ex(const float* gates, const float* c_prev, float* h_out, float* c_out, int R, int H,
                         hipStream_t st) {
  if (R <= 0) return 0;
  hipLaunchKernelGGL(lstm_pointwise_kernel, dim3((R * H + 255) / 256), dim3(256), 0, st, gates, c_prev, h_out,
                     c_out, R, H);
  SF_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------
// kernels = mu (+ noise * exp(0.5 * logvar))  (savi.py:355-365); dist rows [R, 2D]; noise row map
__global__ void sample_dist_kernel(const float* __restrict__ dist, const float* __restrict__ noise,
                                   SfRowMap nmap, float* __restrict__ out, int R, int D) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * D) return;
  const int r = idx / D, j = idx - r * D;
  const float mu = dist[(long long)r * 2 * D + j];
  float v = mu;
  if (noise) v = mu + noise[sf_row_off(nmap, r) + j] * expf(dist[(long long)r * 2 * D + D + j] * 0.5f);
  out[idx] = v;
}

int sf_sample_dist_ex(const float* dist, const float* noise, SfRowMap nmap, float* out, int R, int D,
                      hipStream_t st) {
  if (R <= 0) return 0;
  hipLaunchKernelGGL(sample_dist_kernel, dim3((R * D + 255) / 256), dim3(256), 0, st, dist, noise, nmap, out, R,
                     D);
  SF_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------
__global__ void copy_rows_kernel(const float* __restrict__ src, SfRowMap smap, float* __restrict__ dst,
                                 SfRowMap dmap, int rows, int cols) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  const int r = idx / cols, j = idx - r * cols;
  dst[sf_row_off(dmap, r) + j] = src[sf_row_off(smap, r) + j];
}

int sf_copy_rows_ex(const float* src, SfRowMap smap, float* dst, SfRowMap dmap, int rows, int cols,
                    hipStream_t st) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(copy_rows_kernel, dim3((rows * cols + 255) / 256), dim3(256), 0, st, src, smap, dst, dmap,
                     rows, cols);
  SF_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------
// [Cout][Cin][ks][ks] -> [Cout][ks][ks][Cin]
__global__ void pack_conv_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int ks) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = Cout * Cin * ks * ks;
  if (idx >= total) return;
  const int ci = idx % Cin, rem = idx / Cin, tap = rem % (ks * ks), co = rem / (ks * ks);
  out[idx] = w[((long long)co * Cin + ci) * ks * ks + tap];
}

// [Cin][Cout][ks][ks] (ConvTranspose2d) -> [Cout][ks][ks][Cin]
__global__ void pack_deconv_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int Cout, int ks) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = Cout * Cin * ks * ks;
  if (idx >= total) return;
  const int ci = idx % Cin, rem = idx / Cin, tap = rem % (ks * ks), co = rem / (ks * ks);
  out[idx] = w[((long long)ci * Cout + co) * ks * ks + tap];
}

// spatial broadcast + soft position embedding (savi.py:512-517): out[r, p, c] = slots[r, c] + table[p, c]
__global__ void slot_broadcast_kernel(const float* __restrict__ slots, const float* __restrict__ table,
                                      float* __restrict__ out, int R, int P, int D) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)R * P * D) return;
  const int c = idx % D;
  const long long rp = idx / D;
  const int pp = rp % P;
  const long long r = rp / P;
  out[idx] = slots[r * D + c] + table[(long long)pp * D + c];
}

__global__ void zero_u32_kernel(unsigned* p, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}

// First decoder layer on its broadcast input (sf_savi_decoder.l0_weff / l0_posterm): a 5 x 5 stride-2 transposed convolution of
// slot + pos_table[p] is  (sum of the taps that reach output pixel p from INSIDE the map) . slot + const[p]; the tap set depends on p only
// through the parity and border class of its two coordinates (5 x 5 classes).  table [R][25 * C1] = slots . l0_weff^T (one GEMM),
// out [R][2 res][2 res][C1] = relu(table[r][class(oy) * 5 + class(ox)] + posterm[oy][ox])   (savi.py:512-518 with nerv's deconv + ReLU)
__device__ __forceinline__ int l0_class(int o, int res) {
  const int o2 = o >> 1;
  if (o & 1) return o2 == res - 1 ? 4 : 3;
  return o2 == 0 ? 0 : (o2 == res - 1 ? 2 : 1);
}
__global__ void decode_l0_expand_kernel(const float* __restrict__ table, const float* __restrict__ posterm, float* __restrict__ out,
                                        int R, int res, int C1) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one float4 of channels
  const int c4n = C1 / 4, W = 2 * res;
  if (idx >= (long long)R * W * W * c4n) return;
  const int c4 = idx % c4n;
  const long long rp = idx / c4n;
  const int pix = rp % (W * W);
  const long long r = rp / (W * W);
  const int oy = pix / W, ox = pix - oy * W;
  const int cls = l0_class(oy, res) * 5 + l0_class(ox, res);
  const f32x4 a = *(const f32x4*)(table + (r * 25 + cls) * C1 + 4 * c4);
  const f32x4 b = *(const f32x4*)(posterm + (long long)pix * C1 + 4 * c4);
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = fmaxf(a[e] + b[e], 0.f);
  *(f32x4*)(out + idx * 4) = v;
}

// decoder head (savi.py:519-525): dec [F*N, HW, 4] (r,g,b,mask-logit per slot) ->
// masks = softmax over slots, recon_combined = sum_n recons * masks; all outputs NCHW-per-frame.
// slot_max != NULL (the segmentation of vp_utils.py:20-41 follows): also the per-(frame, slot) maximum of the mask over the pixels, as
// float bits in unsigned words cleared before the launch (masks are positive: the bit patterns order like the values; max is
// order-independent, so the atomics stay deterministic).
// the N (<= DC_NMAX) head outputs of a pixel in registers: ONE 16-byte load per slot (the first version walked the slots three times --
// max, sum, outputs -- with dependent loads: 32 us per 36 frames where the bytes need 15); masks[n] = exp(logit_n - max) / sum
constexpr int DC_NMAX = 16;
template <int NN>
__device__ __forceinline__ void slot_softmax_regs(const float* __restrict__ dec, long long f, int N, int HW, int pix, f32x4 (&v)[NN], float (&m)[NN]) {
  float mx = -INFINITY;
#pragma unroll
  for (int n = 0; n < NN; ++n) {
    if (n < N) {
      v[n] = *(const f32x4*)(dec + ((f * N + n) * HW + pix) * 4);
      mx = fmaxf(mx, v[n][3]);
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int n = 0; n < NN; ++n) {
    m[n] = n < N ? expf(v[n][3] - mx) : 0.f;
    sum += m[n];
  }
  const float inv = 1.0f / sum;
#pragma unroll
  for (int n = 0; n < NN; ++n) m[n] *= inv;
}

template <int NN>
__global__ void decode_combine_kernel(const float* __restrict__ dec, float* __restrict__ recon,
                                      float* __restrict__ recons, float* __restrict__ masks, unsigned* __restrict__ slot_max, int F, int N,
                                      int HW) {
  const long long idx0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = idx0 < (long long)F * HW;
  const long long idx = valid ? idx0 : (long long)F * HW - 1;
  const int pix = idx % HW;
  const long long f = idx / HW;
  f32x4 v[NN];
  float m[NN];
  slot_softmax_regs<NN>(dec, f, N, HW, pix, v, m);
  float acc[3] = {0.f, 0.f, 0.f};
  // a workgroup's 256 pixels lie in one frame when HW % 256 == 0: the waves meet in LDS, one global atomic per workgroup and slot
  // (per wave they queued 256 deep on every (frame, slot) word: +0.18 ms per 32 frames)
  const bool wg_one_frame = (HW % 256) == 0;
  __shared__ unsigned wg_max[256];
  if (slot_max && wg_one_frame) {
    wg_max[threadIdx.x] = 0u;
    __syncthreads();
  }
#pragma unroll
  for (int n = 0; n < NN; ++n) {
    if (n >= N) break;
    if (masks && valid) masks[(f * N + n) * HW + pix] = m[n];
    if (slot_max) {
      if (wg_one_frame) {
        const float wm = sf_wave_max(valid ? m[n] : 0.f);
        if ((threadIdx.x & 63) == 0) atomicMax(&wg_max[n], __float_as_uint(wm));
      } else if (valid) {
        atomicMax(slot_max + f * N + n, __float_as_uint(m[n]));
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (recons && valid) recons[((f * N + n) * 3 + c) * HW + pix] = v[n][c];
      acc[c] = fmaf(v[n][c], m[n], acc[c]);
    }
  }
  if (valid) {
#pragma unroll
    for (int c = 0; c < 3; ++c) recon[(f * 3 + c) * HW + pix] = acc[c];
  }
  if (slot_max && wg_one_frame) {
    __syncthreads();
    const long long f0 = ((long long)blockIdx.x * blockDim.x) / HW;   // the workgroup's frame
    if ((int)threadIdx.x < N && f0 < F) atomicMax(slot_max + f0 * N + threadIdx.x, wg_max[threadIdx.x]);
  }
}

// postproc_mask (vp_utils.py:20-41) on the decoder's own masks, recomputed from dec with the expressions of decode_combine_kernel (the
// same bits): the slot whose peak mask value over the frame is smallest is the background (first index on ties, as torch.argmin); a
// pixel whose best mask value is below fg_thre goes to it, every other pixel to its argmax over slots (first index on ties).
template <int NN>
__global__ void decode_seg_kernel(const float* __restrict__ dec, const unsigned* __restrict__ slot_max, long long* __restrict__ seg64,
                                  unsigned char* __restrict__ seg8, int F, int N, int HW, float thre) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)F * HW) return;
  const int pix = idx % HW;
  const long long f = idx / HW;
  f32x4 v[NN];
  float m[NN];
  slot_softmax_regs<NN>(dec, f, N, HW, pix, v, m);
  int bg = 0, best = 0;
  float bgv = __uint_as_float(slot_max[f * N]), bestv = -1.f;
#pragma unroll
  for (int n = 0; n < NN; ++n) {
    if (n >= N) break;
    const float sm = __uint_as_float(slot_max[f * N + n]);
    if (sm < bgv) {
      bgv = sm;
      bg = n;
    }
    if (m[n] > bestv) {
      bestv = m[n];
      best = n;
    }
  }
  const int out = bestv < thre ? bg : best;
  if (seg64) seg64[idx] = out;
  if (seg8) seg8[idx] = (unsigned char)out;
}

// postproc_mask on given masks [F][N][HW]: per-(frame, slot) maxima (one workgroup per row), then the rule above per pixel
__global__ __launch_bounds__(256) void mask_rowmax_kernel(const float* __restrict__ masks, unsigned* __restrict__ slot_max, int HW) {
  const float* row = masks + (long long)blockIdx.x * HW;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < HW; i += 256) m = fmaxf(m, row[i]);
  m = sf_wave_max(m);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  // stored as an order-preserving key of the float (masks may be any float here): flip the sign bit / all bits
  if (threadIdx.x == 0) {
    const float r = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    const unsigned b = __float_as_uint(r);
    slot_max[blockIdx.x] = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  }
}
__global__ void mask_seg_kernel(const float* __restrict__ masks, const unsigned* __restrict__ slot_key, long long* __restrict__ seg64,
                                unsigned char* __restrict__ seg8, int F, int N, int HW, float thre) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)F * HW) return;
  const int pix = idx % HW;
  const long long f = idx / HW;
  int bg = 0, best = 0;
  unsigned bgk = slot_key[f * N];
  float bestv = masks[(f * N) * HW + pix];
  for (int n = 1; n < N; ++n) {
    const unsigned k = slot_key[f * N + n];
    if (k < bgk) {
      bgk = k;
      bg = n;
    }
    const float m = masks[(f * N + n) * HW + pix];
    if (m > bestv) {
      bestv = m;
      best = n;
    }
  }
  const int out = bestv < thre ? bg : best;
  if (seg64) seg64[idx] = out;
  if (seg8) seg8[idx] = (unsigned char)out;
}

// table[p, c] = sum_j grid[p, j] * w[c, j] + b[c]      (SoftPositionEmbed, utils.py:52-63)
__global__ void pos_table_kernel(const float* __restrict__ grid, const float* __restrict__ w,
                                 const float* __restrict__ b, float* __restrict__ out, int HW, int C) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= HW * C) return;
  const int p = idx / C, c = idx - p * C;
  float acc = 0.f;
  for (int j = 0; j < 4; ++j) acc = fmaf(grid[p * 4 + j], w[c * 4 + j], acc);
  out[idx] = acc + b[c];
}

// F.interpolate(mode='bilinear', align_corners=False) of [R, Hi, Wi] planes to [R, Ho, Wo] (steve.py:230-238)
__global__ void bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Hi, int Wi,
                                int Ho, int Wo) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)R * Ho * Wo) return;
  const int x = idx % Wo, y = (idx / Wo) % Ho;
  const long long r = idx / ((long long)Wo * Ho);
  const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
  float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
  fy = fy < 0.f ? 0.f : fy;
  fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0), x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
  const float* p = in + r * Hi * Wi;
  out[idx] = hy * (hx * p[y0 * Wi + x0] + lx * p[y0 * Wi + x1]) + ly * (hx * p[y1 * Wi + x0] + lx * p[y1 * Wi + x1]);
}

extern "C" {

int sf_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, int rows, int D,
                     float eps, void* stream) {
  SF_REQUIRE(x && gamma && beta && y && rows >= 0 && D > 0, "bad layernorm arguments");
  return sf_layernorm_ex(x, sf_rows(D), gamma, beta, y, sf_rows(D), rows, D, eps, (hipStream_t)stream);
}

int sf_mha_f32(const float* qkv, float* out, int B, int L, int Lq, int d_model, int num_heads, void* stream) {
  SF_REQUIRE(qkv && out, "null pointer");
  return sf_mha_ex(qkv, out, B, L, Lq, d_model, num_heads, (hipStream_t)stream);
}

int sf_lstm_pointwise_f32(const float* gates, const float* c_prev, float* h_out, float* c_out, int R, int H,
                          void* stream) {
  SF_REQUIRE(gates && h_out && c_out && R >= 0 && H > 0, "bad lstm arguments");
  return sf_lstm_pointwise_ex(gates, c_prev, h_out, c_out, R, H, (hipStream_t)stream);
}

int sf_sample_dist_f32(const float* dist, const float* noise, float* out, int R, int D, void* stream) {
  SF_REQUIRE(dist && out && R >= 0 && D > 0, "bad sample_dist arguments");
  return sf_sample_dist_ex(dist, noise, sf_rows(D), out, R, D, (hipStream_t)stream);
}

int sf_pack_conv_weight_f32(const float* w_oihw, float* w_ohwi, int Cout, int Cin, int ks, void* stream) {
  SF_REQUIRE(w_oihw && w_ohwi && Cout > 0 && Cin > 0 && ks > 0, "bad pack arguments");
  const int total = Cout * Cin * ks * ks;
  hipLaunchKernelGGL(pack_conv_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw,
                     w_ohwi, Cout, Cin, ks);
  SF_CHECK_LAUNCH();
  return 0;
}

int sf_pack_deconv_weight_f32(const float* w_iohw, float* w_ohwi, int Cin, int Cout, int ks, void* stream) {
  SF_REQUIRE(w_iohw && w_ohwi && Cout > 0 && Cin > 0 && ks > 0, "bad pack arguments");
  const int total = Cout * Cin * ks * ks;
  hipLaunchKernelGGL(pack_deconv_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_iohw,
                     w_ohwi, Cin, Cout, ks);
  SF_CHECK_LAUNCH();
  return 0;
}

int sf_slot_broadcast_f32(const float* slots, const float* table, float* out, int R, int P, int D, void* stream) {
  SF_REQUIRE(slots && table && out && R >= 0 && P > 0 && D > 0, "bad broadcast arguments");
  const long long total = (long long)R * P * D;
  if (total == 0) return 0;
  hipLaunchKernelGGL(slot_broadcast_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     slots, table, out, R, P, D);
  SF_CHECK_LAUNCH();
  return 0;
}

int sf_decode_l0_expand_f32(const float* table, const float* posterm, float* out, int R, int res, int C1, void* stream) {
  SF_REQUIRE(table && posterm && out && R >= 0 && res >= 2 && C1 > 0 && (C1 % 4) == 0, "bad l0-expand arguments");
  const long long total = (long long)R * 4 * res * res * (C1 / 4);
  if (total == 0) return 0;
  hipLaunchKernelGGL(decode_l0_expand_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, table, posterm,
                     out, R, res, C1);
  SF_CHECK_LAUNCH();
  return 0;
}

int sf_decode_combine_seg_f32(const float* dec, float* recon_combined, float* recons, float* masks, long long* seg_i64,
                              unsigned char* seg_u8, float fg_thre, unsigned* slot_max, int F, int N, int HW, void* stream) {
  SF_REQUIRE(dec && recon_combined && F >= 0 && N >= 1 && HW > 0, "bad combine arguments");
  const bool seg = seg_i64 || seg_u8;
  SF_REQUIRE(!seg || slot_max, "the segmentation needs the slot_max scratch ([F * N] words)");
  SF_REQUIRE(N <= DC_NMAX, "at most 16 slots");
  const long long total = (long long)F * HW;
  if (total == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (seg) {
    hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)(((long long)F * N + 255) / 256)), dim3(256), 0, st, slot_max, (long long)F * N);
    SF_CHECK_LAUNCH();
  }
  if (N <= 8)
    hipLaunchKernelGGL(decode_combine_kernel<8>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dec, recon_combined, recons, masks,
                       seg ? slot_max : nullptr, F, N, HW);
  else
    hipLaunchKernelGGL(decode_combine_kernel<DC_NMAX>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dec, recon_combined, recons, masks,
                       seg ? slot_max : nullptr, F, N, HW);
  SF_CHECK_LAUNCH();
  if (seg) {
    if (N <= 8)
      hipLaunchKernelGGL(decode_seg_kernel<8>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dec, slot_max, seg_i64, seg_u8, F, N, HW, fg_thre);
    else
      hipLaunchKernelGGL(decode_seg_kernel<DC_NMAX>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dec, slot_max, seg_i64, seg_u8, F, N, HW, fg_thre);
    SF_CHECK_LAUNCH();
  }
  return 0;
}

int sf_decode_combine_f32(const float* dec, float* recon_combined, float* recons, float* masks, int F, int N, int HW,
                          void* stream) {
  return sf_decode_combine_seg_f32(dec, recon_combined, recons, masks, nullptr, nullptr, 0.5f, nullptr, F, N, HW, stream);
}

int sf_postproc_mask_f32(const float* masks, long long* seg_i64, unsigned char* seg_u8, float fg_thre, unsigned* slot_max, int F, int N,
                         int HW, void* stream) {
  SF_REQUIRE(masks && (seg_i64 || seg_u8) && slot_max && F >= 0 && N >= 1 && N <= 255 && HW > 0, "bad postproc arguments");
  const long long total = (long long)F * HW;
  if (total == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mask_rowmax_kernel, dim3((unsigned)(F * N)), dim3(256), 0, st, masks, slot_max, HW);
  SF_CHECK_LAUNCH();
  hipLaunchKernelGGL(mask_seg_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, masks, slot_max, seg_i64, seg_u8, F, N, HW,
                     fg_thre);
  SF_CHECK_LAUNCH();
  return 0;
}

int sf_pos_embed_table_f32(const float* grid, const float* dense_w, const float* dense_b, float* table, int HW,
                           int C, void* stream) {
  SF_REQUIRE(grid && dense_w && dense_b && table && HW > 0 && C > 0, "bad pos-embed arguments");
  hipLaunchKernelGGL(pos_table_kernel, dim3((HW * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, grid,
                     dense_w, dense_b, table, HW, C);
  SF_CHECK_LAUNCH();
  return 0;
}

int sf_bilinear_resize_f32(const float* in, float* out, long long R, int Hi, int Wi, int Ho, int Wo,
                           void* stream) {
  SF_REQUIRE(in && out && R >= 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "bad resize arguments");
  const long long total = R * Ho * Wo;
  if (total == 0) return 0;
  hipLaunchKernelGGL(bilinear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     in, out, (int)R, Hi, Wi, Ho, Wo);
  SF_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
