// Training of the rollout Transformer: forward with saved activations and the backward pass of
// SlotRollouter.forward (slotformer.py:85-126) with torch's pre-LN nn.TransformerEncoderLayer (slotformer.py:72-80),
// dropout included (the reference trains with the layer's default p = 0.1).
//
// Design (MI355X, 288 GB of HBM): nothing is recomputed and nothing is reduced early.
//   * every activation of every rollout step is kept, stacked over the steps ([S][M][width] per layer and kind), and so
//     is every gradient that feeds a weight gradient;
//   * the backward-through-time loop only runs the data-gradient chain (dX = dY . W on the forward GEMM core against
//     transposed weight copies, LayerNorm / attention / dropout backward kernels);
//   * after the loop each weight gradient is ONE contraction over all S*M rows: dW = dY^T . X on a split-bf16 MFMA kernel
//     that reads both operands in their natural row-major layout (no transposes), split over row ranges into partial
//     tiles that a second kernel sums in a fixed order (deterministic; no atomics);
//   * bias and LayerNorm-parameter gradients are column sums over the same stacked buffers.
// Gradients are written (not accumulated) into caller-owned buffers (sf_rollouter_grads), which the host lays out as one
// flat bucket for the data-parallel all-reduce.
//
// Dropout masks are a pure function of (seed, rollout step, layer, site, element index): keep = mix32(idx ^ site_seed)
// >> 8 >= p * 2^24, so the backward pass regenerates them and tests can rebuild them on the host.
#include <math.h>
#include <stdlib.h>

#include "../../include/slotformer_hip.h"
#include "sf_internal.h"
#include "bf16_planes.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------------------------------------
// dropout mask
// ---------------------------------------------------------------------------------------------------------------------
enum { SITE_ATTN_P = 0, SITE_ATTN_O = 1, SITE_FFN_H = 2, SITE_FFN_O = 3 };
static inline uint32_t site_seed(unsigned long long seed, int step, int layer, int site) {
  const uint32_t tag = (uint32_t)((step * 64 + layer) * 4 + site);
  return sf_mix32((uint32_t)seed ^ sf_mix32((uint32_t)(seed >> 32) + 0x9e3779b9u * (tag + 1u)));
}
__device__ __forceinline__ bool sf_keep(uint32_t sseed, uint32_t idx, uint32_t thresh) {
  return (sf_mix32(idx ^ sseed) >> 8) >= thresh;
}
static inline uint32_t drop_thresh(float p) { return (uint32_t)((double)p * 16777216.0); }

// y = res + keep(x) / (1 - p)   (res may be NULL); n4 float4 elements
__global__ __launch_bounds__(256) void residual_dropout_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                               float* __restrict__ y, long long n4, uint32_t sseed,
                                                               uint32_t thresh, float inv_keep) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 v = reinterpret_cast<const float4*>(x)[i];
  float4 r = res ? reinterpret_cast<const float4*>(res)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  const uint32_t e = (uint32_t)(4 * i);
  r.x += sf_keep(sseed, e, thresh) ? v.x * inv_keep : 0.f;
  r.y += sf_keep(sseed, e + 1, thresh) ? v.y * inv_keep : 0.f;
  r.z += sf_keep(sseed, e + 2, thresh) ? v.z * inv_keep : 0.f;
  r.w += sf_keep(sseed, e + 3, thresh) ? v.w * inv_keep : 0.f;
  reinterpret_cast<float4*>(y)[i] = r;
}

// dpre = hdn > 0 ? dh * inv_keep : 0   (hdn is the saved post-ReLU, post-dropout hidden activation)
__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(float* __restrict__ dh, const float* __restrict__ hdn,
                                                               long long n4, float inv_keep) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 g = reinterpret_cast<float4*>(dh)[i];
  const float4 h = reinterpret_cast<const float4*>(hdn)[i];
  g.x = h.x > 0.f ? g.x * inv_keep : 0.f;
  g.y = h.y > 0.f ? g.y * inv_keep : 0.f;
  g.z = h.z > 0.f ? g.z * inv_keep : 0.f;
  g.w = h.w > 0.f ? g.w * inv_keep : 0.f;
  reinterpret_cast<float4*>(dh)[i] = g;
}

// ---------------------------------------------------------------------------------------------------------------------
// window assembly:  x0[b, l, :] = tok[s + l / N][b * N + l % N][:] + pe[l]   and its adjoint (dtok += dx)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void window_assemble_kernel(const float* __restrict__ tok, const float* __restrict__ pe,
                                                              float* __restrict__ x0, int B, int L, int N, int d4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)B * L * d4;
  if (i >= total) return;
  const int c = (int)(i % d4);
  const int row = (int)(i / d4);
  const int b = row / L, l = row - b * L;
  const int f = l / N, n = l - f * N;
  const float4 t = reinterpret_cast<const float4*>(tok)[((long long)f * B * N + (long long)b * N + n) * d4 + c];
  const float4 p = reinterpret_cast<const float4*>(pe)[(long long)l * d4 + c];
  reinterpret_cast<float4*>(x0)[i] = make_float4(t.x + p.x, t.y + p.y, t.z + p.z, t.w + p.w);
}
__global__ __launch_bounds__(256) void window_scatter_kernel(const float* __restrict__ dx, float* __restrict__ dtok, int B,
                                                             int L, int N, int d4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)B * L * d4;
  if (i >= total) return;
  const int c = (int)(i % d4);
  const int row = (int)(i / d4);
  const int b = row / L, l = row - b * L;
  const int f = l / N, n = l - f * N;
  float4* dst = reinterpret_cast<float4*>(dtok) + ((long long)f * B * N + (long long)b * N + n) * d4 + c;
  const float4 g = reinterpret_cast<const float4*>(dx)[i];
  float4 t = *dst;
  t.x += g.x; t.y += g.y; t.z += g.z; t.w += g.w;
  *dst = t;
}

// adjoint of the "+ pe[l]" of the window assembly: dpe[l, :] += sum_b dx[b, l, :]   (one thread per (l, float4 column), the batch
// in a loop: L * d / 4 <= 8192 threads, each reading B rows d floats apart in the same column -> coalesced across the workgroup)
__global__ __launch_bounds__(256) void pe_grad_kernel(const float* __restrict__ dx, float* __restrict__ dpe, int B, int L, int d4) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= L * d4) return;
  float4 a = reinterpret_cast<float4*>(dpe)[i];
  for (int b = 0; b < B; ++b) {
    const float4 g = reinterpret_cast<const float4*>(dx)[(long long)b * L * d4 + i];
    a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
  }
  reinterpret_cast<float4*>(dpe)[i] = a;
}

// y += x  (float4)
__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ y, const float* __restrict__ x, long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 a = reinterpret_cast<float4*>(y)[i];
  const float4 b = reinterpret_cast<const float4*>(x)[i];
  a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  reinterpret_cast<float4*>(y)[i] = a;
}

// out[c][r] = in[r][c]
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cn) {
  __shared__ float t[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < R && c0 + tx < Cn) t[i][tx] = in[(long long)(r0 + i) * Cn + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < Cn && r0 + tx < R) out[(long long)(c0 + i) * R + r0 + tx] = t[tx][i];
}

// ---------------------------------------------------------------------------------------------------------------------
// attention with saved-nothing backward: one workgroup per (head, video); L <= 128, head_dim <= 64.
// qkv [B*L, 3d] (q|k|v); ctx / dctx [B*L, d]; dqkv [B*L, 3d].  Dropout acts on the softmax weights (nn.MultiheadAttention).
// ---------------------------------------------------------------------------------------------------------------------
struct AttnLds {
  float *q, *k, *v, *g, *P, *dS;
};
__device__ __forceinline__ void attn_probs(const float* qkv, int L, int d, int hd, int b, int h, float scale, float* q,
                                           float* k, float* v, float* P, int hp, int lp) {
  const int tid = threadIdx.x;
  for (int i = tid; i < L * hd; i += 256) {
    const int r = i / hd, c = i - r * hd;
    const float* row = qkv + ((long long)b * L + r) * 3 * d + h * hd + c;
    q[r * hp + c] = row[0] * scale;
    k[r * hp + c] = row[d];
    v[r * hp + c] = row[2 * d];
  }
  __syncthreads();
  for (int i = tid; i < L * L; i += 256) {
    const int r = i / L, c = i - r * L;
    float s = 0.f;
    for (int e = 0; e < hd; ++e) s = fmaf(q[r * hp + e], k[c * hp + e], s);
    P[r * lp + c] = s;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  for (int r = wave; r < L; r += 4) {
    const float a0 = lane < L ? P[r * lp + lane] : -INFINITY;
    const float a1 = lane + 64 < L ? P[r * lp + lane + 64] : -INFINITY;
    const float mx = sf_wave_max(fmaxf(a0, a1));
    const float e0 = lane < L ? __expf(a0 - mx) : 0.f;
    const float e1 = lane + 64 < L ? __expf(a1 - mx) : 0.f;
    const float inv = 1.f / sf_wave_sum(e0 + e1);
    if (lane < L) P[r * lp + lane] = e0 * inv;
    if (lane + 64 < L) P[r * lp + lane + 64] = e1 * inv;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void attn_train_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ ctx, int L,
                                                             int d, int hd, float scale, uint32_t sseed, uint32_t thresh,
                                                             float inv_keep) {
  extern __shared__ float lds[];
  const int hp = hd + 1, lp = L + 1;
  float* q = lds;
  float* k = q + L * hp;
  float* v = k + L * hp;
  float* P = v + L * hp;
  const int h = blockIdx.x, b = blockIdx.y, H = gridDim.x;
  attn_probs(qkv, L, d, hd, b, h, scale, q, k, v, P, hp, lp);
  const int tid = threadIdx.x;
  const uint32_t base = (uint32_t)(((long long)b * H + h) * L * L);
  for (int i = tid; i < L * hd; i += 256) {
    const int r = i / hd, c = i - r * hd;
    float s = 0.f;
    for (int j = 0; j < L; ++j) {
      float p = P[r * lp + j];
      if (thresh) p = sf_keep(sseed, base + r * L + j, thresh) ? p * inv_keep : 0.f;
      s = fmaf(p, v[j * hp + c], s);
    }
    ctx[((long long)b * L + r) * d + h * hd + c] = s;
  }
}

__global__ __launch_bounds__(256) void attn_train_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dctx,
                                                             float* __restrict__ dqkv, int L, int d, int hd, float scale,
                                                             uint32_t sseed, uint32_t thresh, float inv_keep) {
  extern __shared__ float lds[];
  const int hp = hd + 1, lp = L + 1;
  float* q = lds;
  float* k = q + L * hp;
  float* v = k + L * hp;
  float* g = v + L * hp;
  float* P = g + L * hp;
  float* dS = P + L * lp;
  const int h = blockIdx.x, b = blockIdx.y, H = gridDim.x;
  const int tid = threadIdx.x;
  for (int i = tid; i < L * hd; i += 256) {
    const int r = i / hd, c = i - r * hd;
    g[r * hp + c] = dctx[((long long)b * L + r) * d + h * hd + c];
  }
  attn_probs(qkv, L, d, hd, b, h, scale, q, k, v, P, hp, lp);   // q holds q * scale
  const uint32_t base = (uint32_t)(((long long)b * H + h) * L * L);
  // dPd[i][j] = g[i] . v[j];  dP = mask * dPd / keep;  stash dP in dS
  for (int i = tid; i < L * L; i += 256) {
    const int r = i / L, c = i - r * L;
    float s = 0.f;
    for (int e = 0; e < hd; ++e) s = fmaf(g[r * hp + e], v[c * hp + e], s);
    if (thresh) s = sf_keep(sseed, base + r * L + c, thresh) ? s * inv_keep : 0.f;
    dS[r * lp + c] = s;
  }
  __syncthreads();
  // dV[j][c] = sum_i Pd[i][j] g[i][c]   (Pd = dropped probabilities)
  for (int i = tid; i < L * hd; i += 256) {
    const int j = i / hd, c = i - j * hd;
    float s = 0.f;
    for (int r = 0; r < L; ++r) {
      float p = P[r * lp + j];
      if (thresh) p = sf_keep(sseed, base + r * L + j, thresh) ? p * inv_keep : 0.f;
      s = fmaf(p, g[r * hp + c], s);
    }
    dqkv[((long long)b * L + j) * 3 * d + 2 * d + h * hd + c] = s;
  }
  __syncthreads();
  // dS = P * (dP - rowsum(dP * P))
  const int wave = tid >> 6, lane = tid & 63;
  for (int r = wave; r < L; r += 4) {
    const float p0 = lane < L ? P[r * lp + lane] : 0.f, p1 = lane + 64 < L ? P[r * lp + lane + 64] : 0.f;
    const float d0 = lane < L ? dS[r * lp + lane] : 0.f, d1 = lane + 64 < L ? dS[r * lp + lane + 64] : 0.f;
    const float dot = sf_wave_sum(p0 * d0 + p1 * d1);
    if (lane < L) dS[r * lp + lane] = p0 * (d0 - dot);
    if (lane + 64 < L) dS[r * lp + lane + 64] = p1 * (d1 - dot);
  }
  __syncthreads();
  // dq[i][c] = scale * sum_j dS[i][j] k[j][c];  dk[j][c] = sum_i dS[i][j] (q[i][c] * scale)
  for (int i = tid; i < L * hd; i += 256) {
    const int r = i / hd, c = i - r * hd;
    float sq = 0.f, sk = 0.f;
    for (int j = 0; j < L; ++j) {
      sq = fmaf(dS[r * lp + j], k[j * hp + c], sq);
      sk = fmaf(dS[j * lp + r], q[j * hp + c], sk);
    }
    float* o = dqkv + ((long long)b * L + r) * 3 * d + h * hd + c;
    o[0] = sq * scale;
    o[d] = sk;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same two kernels on the exact-f32 matrix pipe (v_mfma_f32_32x32x2_f32) for head_dim 32 / 64 and windows of at most
// 96 tokens: every product (q k^T, P v, g v^T, Pd^T g, dS k, dS^T q) is a set of 32x32 output tiles whose operands are
// read from zero-padded LDS tiles through (row, k) -> address functors, one tile per wave at a time.
// ---------------------------------------------------------------------------------------------------------------------
template <class FA, class FB>
__device__ __forceinline__ f32x16 mm32(FA a, FB b, int K, int lane) {
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int i0 = lane & 31, kk = lane >> 5;
  for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a(i0, k + kk), b(k + kk, i0), acc, 0, 0, 0);
  return acc;
}
#define MM_ROW(r, lane) (((r) & 3) + 8 * ((r) >> 2) + 4 * ((lane) >> 5))

// load q (scaled), k, v [L, HD] of (video b, head h) into zero-padded [Lp][HD + 1] tiles
template <int HD>
__device__ __forceinline__ void attn_load_qkv(const float* qkv, int L, int Lp, int d, int b, int h, float scale, float* q,
                                              float* k, float* v) {
  constexpr int HP = HD + 1, C4 = HD / 4;
  for (int i = threadIdx.x; i < Lp * C4; i += 256) {
    const int r = i / C4, c = (i - r * C4) * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), bb = a, cc = a;
    if (r < L) {
      const float* row = qkv + ((long long)b * L + r) * 3 * d + h * HD + c;
      a = *reinterpret_cast<const float4*>(row);
      bb = *reinterpret_cast<const float4*>(row + d);
      cc = *reinterpret_cast<const float4*>(row + 2 * d);
    }
    float* qo = q + r * HP + c;
    float* ko = k + r * HP + c;
    float* vo = v + r * HP + c;
    qo[0] = a.x * scale; qo[1] = a.y * scale; qo[2] = a.z * scale; qo[3] = a.w * scale;
    ko[0] = bb.x; ko[1] = bb.y; ko[2] = bb.z; ko[3] = bb.w;
    vo[0] = cc.x; vo[1] = cc.y; vo[2] = cc.z; vo[3] = cc.w;
  }
}

// S = q k^T into P (all Lp x Lp tiles), then row softmax.  P <- probabilities (zero outside L x L); if Pd != nullptr it
// receives the dropped probabilities, else (drop_in_place) P itself is dropped.
template <int HD>
__device__ __forceinline__ void attn_probs_mfma(const float* q, const float* k, float* P, float* Pd, bool drop_in_place, int L,
                                                int Lp, uint32_t sseed, uint32_t thresh, float inv_keep, uint32_t base) {
  constexpr int HP = HD + 1;
  const int lp = Lp + 1, nt = Lp / 32;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int t = wave; t < nt * nt; t += 4) {
    const int ti = (t / nt) * 32, tj = (t % nt) * 32;
    const f32x16 acc = mm32([&](int i, int kk) { return q[(ti + i) * HP + kk]; },
                            [&](int kk, int j) { return k[(tj + j) * HP + kk]; }, HD, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) P[(ti + MM_ROW(r, lane)) * lp + tj + (lane & 31)] = acc[r];
  }
  __syncthreads();
  for (int r = wave; r < Lp; r += 4) {
    float p0 = 0.f, p1 = 0.f;
    if (r < L) {
      const float a0 = lane < L ? P[r * lp + lane] : -INFINITY;
      const float a1 = lane + 64 < L ? P[r * lp + lane + 64] : -INFINITY;
      const float mx = sf_wave_max(fmaxf(a0, a1));
      const float e0 = lane < L ? __expf(a0 - mx) : 0.f;
      const float e1 = lane + 64 < L ? __expf(a1 - mx) : 0.f;
      const float inv = 1.f / sf_wave_sum(e0 + e1);
      p0 = e0 * inv;
      p1 = e1 * inv;
    }
    float d0 = p0, d1 = p1;
    if (thresh && r < L) {
      d0 = sf_keep(sseed, base + r * L + lane, thresh) ? p0 * inv_keep : 0.f;
      d1 = sf_keep(sseed, base + r * L + lane + 64, thresh) ? p1 * inv_keep : 0.f;
    }
    if (lane < Lp) {
      P[r * lp + lane] = drop_in_place ? d0 : p0;
      if (Pd) Pd[r * lp + lane] = d0;
    }
    if (lane + 64 < Lp) {
      P[r * lp + lane + 64] = drop_in_place ? d1 : p1;
      if (Pd) Pd[r * lp + lane + 64] = d1;
    }
  }
  __syncthreads();
}

template <int HD>
__global__ __launch_bounds__(256) void attn_train_fwd_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ ctx, int L,
                                                                  int Lp, int d, float scale, uint32_t sseed, uint32_t thresh,
                                                                  float inv_keep) {
  extern __shared__ float lds[];
  constexpr int HP = HD + 1;
  const int lp = Lp + 1, nt = Lp / 32;
  float* q = lds;
  float* k = q + Lp * HP;
  float* v = k + Lp * HP;
  float* P = v + Lp * HP;
  const int h = blockIdx.x, b = blockIdx.y, H = gridDim.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  attn_load_qkv<HD>(qkv, L, Lp, d, b, h, scale, q, k, v);
  __syncthreads();
  attn_probs_mfma<HD>(q, k, P, nullptr, true, L, Lp, sseed, thresh, inv_keep, (uint32_t)(((long long)b * H + h) * L * L));
  const int Le = (L + 1) & ~1;
  for (int t = wave; t < nt * (HD / 32); t += 4) {
    const int ti = (t / (HD / 32)) * 32, tj = (t % (HD / 32)) * 32;
    const f32x16 acc = mm32([&](int i, int kk) { return P[(ti + i) * lp + kk]; },
                            [&](int kk, int j) { return v[kk * HP + tj + j]; }, Le, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = ti + MM_ROW(r, lane);
      if (row < L) ctx[((long long)b * L + row) * d + h * HD + tj + (lane & 31)] = acc[r];
    }
  }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_train_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ dctx,
                                                                  float* __restrict__ dqkv, int L, int Lp, int d, float scale,
                                                                  uint32_t sseed, uint32_t thresh, float inv_keep) {
  extern __shared__ float lds[];
  constexpr int HP = HD + 1, C4 = HD / 4, HT = HD / 32;
  const int lp = Lp + 1, nt = Lp / 32;
  float* q = lds;
  float* k = q + Lp * HP;
  float* v = k + Lp * HP;
  float* g = v + Lp * HP;
  float* P = g + Lp * HP;
  float* dS = P + Lp * lp;
  const int h = blockIdx.x, b = blockIdx.y, H = gridDim.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  attn_load_qkv<HD>(qkv, L, Lp, d, b, h, scale, q, k, v);
  for (int i = threadIdx.x; i < Lp * C4; i += 256) {
    const int r = i / C4, c = (i - r * C4) * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < L) a = *reinterpret_cast<const float4*>(dctx + ((long long)b * L + r) * d + h * HD + c);
    float* go = g + r * HP + c;
    go[0] = a.x; go[1] = a.y; go[2] = a.z; go[3] = a.w;
  }
  __syncthreads();
  // P = softmax, dS buffer = dropped probabilities Pd
  attn_probs_mfma<HD>(q, k, P, dS, false, L, Lp, sseed, thresh, inv_keep, (uint32_t)(((long long)b * H + h) * L * L));
  const int Le = (L + 1) & ~1;
  // dPd = g v^T (kept in registers: at most 3 tiles per wave for Lp = 96)
  f32x16 dp[3];
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    const int t = wave + 4 * n;
    if (t < nt * nt) {
      const int ti = (t / nt) * 32, tj = (t % nt) * 32;
      dp[n] = mm32([&](int i, int kk) { return g[(ti + i) * HP + kk]; }, [&](int kk, int j) { return v[(tj + j) * HP + kk]; }, HD,
                   lane);
    }
  }
  // dV = Pd^T g
  for (int t = wave; t < nt * HT; t += 4) {
    const int ti = (t / HT) * 32, tj = (t % HT) * 32;
    const f32x16 acc = mm32([&](int i, int kk) { return dS[kk * lp + ti + i]; },
                            [&](int kk, int j) { return g[kk * HP + tj + j]; }, Le, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = ti + MM_ROW(r, lane);
      if (row < L) dqkv[((long long)b * L + row) * 3 * d + 2 * d + h * HD + tj + (lane & 31)] = acc[r];
    }
  }
  __syncthreads();
  // dP = mask * dPd / keep  (the mask is where Pd is non-zero: softmax weights are strictly positive), over Pd in place
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    const int t = wave + 4 * n;
    if (t < nt * nt) {
      const int ti = (t / nt) * 32, tj = (t % nt) * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float* e = &dS[(ti + MM_ROW(r, lane)) * lp + tj + (lane & 31)];
        *e = (thresh == 0 || *e != 0.f) ? dp[n][r] * (thresh ? inv_keep : 1.f) : 0.f;
      }
    }
  }
  __syncthreads();
  // dS = P * (dP - rowsum(dP * P))
  for (int r = wave; r < L; r += 4) {
    const float p0 = lane < L ? P[r * lp + lane] : 0.f, p1 = lane + 64 < L ? P[r * lp + lane + 64] : 0.f;
    const float d0 = lane < L ? dS[r * lp + lane] : 0.f, d1 = lane + 64 < L ? dS[r * lp + lane + 64] : 0.f;
    const float dot = sf_wave_sum(p0 * d0 + p1 * d1);
    if (lane < Lp) dS[r * lp + lane] = p0 * (d0 - dot);
    if (lane + 64 < Lp) dS[r * lp + lane + 64] = p1 * (d1 - dot);
  }
  __syncthreads();
  // dq = scale * dS k (tiles 0 .. nt*HT-1), dk = dS^T (q * scale) (the next nt*HT tiles)
  for (int t = wave; t < 2 * nt * HT; t += 4) {
    const bool isk = t >= nt * HT;
    const int tt = isk ? t - nt * HT : t;
    const int ti = (tt / HT) * 32, tj = (tt % HT) * 32;
    f32x16 acc;
    if (!isk)
      acc = mm32([&](int i, int kk) { return dS[(ti + i) * lp + kk]; }, [&](int kk, int j) { return k[kk * HP + tj + j]; }, Le,
                 lane);
    else
      acc = mm32([&](int i, int kk) { return dS[kk * lp + ti + i]; }, [&](int kk, int j) { return q[kk * HP + tj + j]; }, Le,
                 lane);
    const float sc = isk ? 1.f : scale;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = ti + MM_ROW(r, lane);
      if (row < L) dqkv[((long long)b * L + row) * 3 * d + (isk ? d : 0) + h * HD + tj + (lane & 31)] = acc[r] * sc;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm backward (one wave per row, D <= 1024, D % 4 == 0):  out = dres + dLN(dy; x, gamma)
// ---------------------------------------------------------------------------------------------------------------------
// out2 (optional): the same gradient pushed through the dropout of the block below it (mask * out / keep), i.e. the
// gradient w.r.t. that block's pre-dropout output -- saves a separate pass over the tensor.
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                     const float* __restrict__ gamma, const float* __restrict__ dres,
                                                     float* __restrict__ out, float* __restrict__ out2, uint32_t sseed2,
                                                     uint32_t thresh, float inv_keep, int rows, int D, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int n4 = D >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (long long)row * D);
  const float4* gr = reinterpret_cast<const float4*>(dy + (long long)row * D);
  const float4* gm = reinterpret_cast<const float4*>(gamma);
  float4 xv[4], gv[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    xv[i] = c < n4 ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
  }
  const float mean = sf_wave_sum(s) / D;
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < n4) {
      xv[i].x -= mean; xv[i].y -= mean; xv[i].z -= mean; xv[i].w -= mean;
      var += (xv[i].x * xv[i].x + xv[i].y * xv[i].y) + (xv[i].z * xv[i].z + xv[i].w * xv[i].w);
    }
  }
  const float rstd = rsqrtf(sf_wave_sum(var) / D + eps);
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < n4) {
      const float4 w = gm[c];
      float4 g = gr[c];
      g.x *= w.x; g.y *= w.y; g.z *= w.z; g.w *= w.w;
      xv[i].x *= rstd; xv[i].y *= rstd; xv[i].z *= rstd; xv[i].w *= rstd;   // x-hat
      gv[i] = g;
      m1 += (g.x + g.y) + (g.z + g.w);
      m2 += (g.x * xv[i].x + g.y * xv[i].y) + (g.z * xv[i].z + g.w * xv[i].w);
    }
  }
  m1 = sf_wave_sum(m1) / D;
  m2 = sf_wave_sum(m2) / D;
  const float4* rr = dres ? reinterpret_cast<const float4*>(dres + (long long)row * D) : nullptr;
  float4* orow = reinterpret_cast<float4*>(out + (long long)row * D);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < n4) {
      float4 r = rr ? rr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      r.x += rstd * (gv[i].x - m1 - xv[i].x * m2);
      r.y += rstd * (gv[i].y - m1 - xv[i].y * m2);
      r.z += rstd * (gv[i].z - m1 - xv[i].z * m2);
      r.w += rstd * (gv[i].w - m1 - xv[i].w * m2);
      orow[c] = r;
      if (out2) {
        if (thresh) {
          const uint32_t e = (uint32_t)(row * D + 4 * c);
          r.x = sf_keep(sseed2, e, thresh) ? r.x * inv_keep : 0.f;
          r.y = sf_keep(sseed2, e + 1, thresh) ? r.y * inv_keep : 0.f;
          r.z = sf_keep(sseed2, e + 2, thresh) ? r.z * inv_keep : 0.f;
          r.w = sf_keep(sseed2, e + 3, thresh) ? r.w * inv_keep : 0.f;
        }
        reinterpret_cast<float4*>(out2 + (long long)row * D)[c] = r;
      }
    }
  }
}

// gradient w.r.t. the final layer output: zero except the last N tokens of every video (dxlast [B*N, d]); dfo = the same
// pushed through the top layer's FFN-output dropout
__global__ __launch_bounds__(256) void scatter_last_kernel(const float* __restrict__ dxlast, float* __restrict__ dx,
                                                           float* __restrict__ dfo, int B, int L, int N, int d4,
                                                           uint32_t sseed, uint32_t thresh, float inv_keep) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * L * d4) return;
  const int c = (int)(i % d4);
  const int row = (int)(i / d4);
  const int b = row / L, l = row - b * L;
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (l >= L - N) r = reinterpret_cast<const float4*>(dxlast)[((long long)b * N + (l - (L - N))) * d4 + c];
  reinterpret_cast<float4*>(dx)[i] = r;
  if (thresh) {
    const uint32_t e = (uint32_t)(4 * i);
    r.x = sf_keep(sseed, e, thresh) ? r.x * inv_keep : 0.f;
    r.y = sf_keep(sseed, e + 1, thresh) ? r.y * inv_keep : 0.f;
    r.z = sf_keep(sseed, e + 2, thresh) ? r.z * inv_keep : 0.f;
    r.w = sf_keep(sseed, e + 3, thresh) ? r.w * inv_keep : 0.f;
  }
  reinterpret_cast<float4*>(dfo)[i] = r;
}

// LayerNorm parameter gradients, stage 1: workgroup g handles rows [g*rpg, (g+1)*rpg); partial[g][0][D] = sum dy*xhat,
// partial[g][1][D] = sum dy.  D <= 1024, D % 4 == 0.
__global__ __launch_bounds__(256) void ln_param_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ partial, long long rows, int rpg, int D,
                                                               float eps) {
  __shared__ float red[3][2][1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n4 = D >> 2;
  float4 ag[4], ab[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) ag[i] = ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long r0 = (long long)blockIdx.x * rpg;
  const long long r1 = r0 + rpg < rows ? r0 + rpg : rows;
  for (long long row = r0 + wave; row < r1; row += 4) {
    const float4* xr = reinterpret_cast<const float4*>(x + row * D);
    const float4* gr = reinterpret_cast<const float4*>(dy + row * D);
    float4 xv[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      xv[i] = c < n4 ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      s += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
    }
    const float mean = sf_wave_sum(s) / D;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      if (c < n4) {
        xv[i].x -= mean; xv[i].y -= mean; xv[i].z -= mean; xv[i].w -= mean;
        var += (xv[i].x * xv[i].x + xv[i].y * xv[i].y) + (xv[i].z * xv[i].z + xv[i].w * xv[i].w);
      }
    }
    const float rstd = rsqrtf(sf_wave_sum(var) / D + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      if (c < n4) {
        const float4 g = gr[c];
        ag[i].x += g.x * xv[i].x * rstd; ag[i].y += g.y * xv[i].y * rstd;
        ag[i].z += g.z * xv[i].z * rstd; ag[i].w += g.w * xv[i].w * rstd;
        ab[i].x += g.x; ab[i].y += g.y; ab[i].z += g.z; ab[i].w += g.w;
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      if (c < n4) {
        reinterpret_cast<float4*>(red[wave - 1][0])[c] = ag[i];
        reinterpret_cast<float4*>(red[wave - 1][1])[c] = ab[i];
      }
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      if (c < n4) {
        for (int w = 0; w < 3; ++w) {
          const float4 a = reinterpret_cast<float4*>(red[w][0])[c], b = reinterpret_cast<float4*>(red[w][1])[c];
          ag[i].x += a.x; ag[i].y += a.y; ag[i].z += a.z; ag[i].w += a.w;
          ab[i].x += b.x; ab[i].y += b.y; ab[i].z += b.z; ab[i].w += b.w;
        }
        reinterpret_cast<float4*>(partial + ((long long)blockIdx.x * 2) * D)[c] = ag[i];
        reinterpret_cast<float4*>(partial + ((long long)blockIdx.x * 2 + 1) * D)[c] = ab[i];
      }
    }
  }
}

// column sums, stage 1: partial[g][n] = sum over rows [g*rpg, (g+1)*rpg) of y[row][n].  The 256 threads of a workgroup
// cover min(n, 256) columns x 256 / that many row lanes (narrow matrices -- conv / head gradients have n = 4 .. 64 and
// millions of rows -- keep every lane busy), four loads in flight per lane, row lanes combined through LDS.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ y, float* __restrict__ partial,
                                                             long long rows, int rpg, int n) {
  __shared__ float red[256];
  int cols = 1;
  while (cols < n && cols < 256) cols <<= 1;   // columns per workgroup (power of two)
  const int lanes = 256 / cols;
  const int cl = threadIdx.x % cols, rl = threadIdx.x / cols;
  const int c = blockIdx.x * cols + cl;
  const long long r0 = (long long)blockIdx.y * rpg;
  const long long r1 = r0 + rpg < rows ? r0 + rpg : rows;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < n) {
    long long r = r0 + rl;
    for (; r + 3LL * lanes < r1; r += 4LL * lanes) {
      s0 += y[r * n + c];
      s1 += y[(r + lanes) * n + c];
      s2 += y[(r + 2LL * lanes) * n + c];
      s3 += y[(r + 3LL * lanes) * n + c];
    }
    for (; r < r1; r += lanes) s0 += y[r * n + c];
  }
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rl == 0 && c < n) {
    float a = red[cl];
    for (int i = 1; i < lanes; ++i) a += red[i * cols + cl];
    partial[(long long)blockIdx.y * n + c] = a;
  }
}

// stage 2 of every split reduction: out[i] = sum_g partial[g][i]  (fixed order: four interleaved chains, then their sum)
// (columns >= n4a go to out2 when it is given: one launch writes d_gamma and d_beta of a LayerNorm)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                              int G, long long n4, float* __restrict__ out2, long long n4a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4* p = reinterpret_cast<const float4*>(partial) + i;
  float4 a[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  int g = 0;
  for (; g + 4 <= G; g += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 b = p[(long long)(g + u) * n4];
      a[u].x += b.x; a[u].y += b.y; a[u].z += b.z; a[u].w += b.w;
    }
  }
  for (; g < G; ++g) {
    const float4 b = p[(long long)g * n4];
    a[0].x += b.x; a[0].y += b.y; a[0].z += b.z; a[0].w += b.w;
  }
  float4 r;
  r.x = (a[0].x + a[1].x) + (a[2].x + a[3].x);
  r.y = (a[0].y + a[1].y) + (a[2].y + a[3].y);
  r.z = (a[0].z + a[1].z) + (a[2].z + a[3].z);
  r.w = (a[0].w + a[1].w) + (a[2].w + a[3].w);
  if (out2 && i >= n4a) reinterpret_cast<float4*>(out2)[i - n4a] = r;
  else reinterpret_cast<float4*>(out)[i] = r;
}

// the same reduction for few columns and many partials (bias / LayerNorm / 64x64 weight gradients: up to 512 partials of 16
// to 1024 float4 columns, where one thread per column leaves the chip idle behind a handful of long serial chains): a
// workgroup takes 16 columns, its 16 thread rows each sum every 16th partial, LDS combines the rows in a fixed tree.
__global__ __launch_bounds__(256) void reduce_partials_wide_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                                   int G, long long n4, float* __restrict__ out2, long long n4a) {
  const int c = threadIdx.x & 15, q = threadIdx.x >> 4;
  const long long i = (long long)blockIdx.x * 16 + c;
  __shared__ f32x4 sh[16][17];
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (i < n4) {
    const f32x4* p = reinterpret_cast<const f32x4*>(partial) + i;
    for (int g = q; g < G; g += 16) a += p[(long long)g * n4];
  }
  sh[q][c] = a;
  __syncthreads();
#pragma unroll
  for (int w = 8; w > 0; w >>= 1) {
    if (q < w) sh[q][c] += sh[q + w][c];
    __syncthreads();
  }
  if (q == 0 && i < n4) {
    if (out2 && i >= n4a) reinterpret_cast<f32x4*>(out2)[i - n4a] = sh[0][c];
    else reinterpret_cast<f32x4*>(out)[i] = sh[0][c];
  }
}
inline void launch_reduce_partials(const float* partial, float* out, int G, long long n4, hipStream_t st, float* out2 = nullptr,
                                   long long n4a = 0) {
  if (G >= 32 && n4 <= 16384)
    hipLaunchKernelGGL(reduce_partials_wide_kernel, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, st, partial, out, G, n4, out2, n4a);
  else
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, partial, out, G, n4, out2, n4a);
}

// ---------------------------------------------------------------------------------------------------------------------
// weight-gradient contraction  dW[n][k] = sum_m Y[m][n] * X[m][k]   (Y [rows, N], X [rows, K], both row-major)
// split-bf16 on v_mfma_f32_32x32x16_bf16; 64x64 output tile per workgroup (4 waves, one 32x32 quadrant each), the
// contraction index is the row index: a chunk of 32 rows of both operands is staged in LDS as f32 and every lane reads
// its 8 values per k16 step with a stride of one LDS row (pitch 68: the two half-waves hit disjoint banks).
// grid (N/64 * K/64, splits): split z handles rows [z*rps, (z+1)*rps) and writes partial[z][N][K].
// ---------------------------------------------------------------------------------------------------------------------
#define TN_P 68
typedef float f32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void tn_split8(const f32x8 v, bf16x8& hi, bf16x8& lo) {
  hi = __builtin_convertvector(v, bf16x8);
  lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x8), bf16x8);
}
// The 32-row chunk of both operands lives in LDS as bf16 hi / lo PLANES (bf16_planes.h; every element split once, where it is staged, instead of once per
// wave that reads it): the contraction index is the tile's ROW index, so the operand fragments come out through ds_read_b64_tr_b16 -- two transposing
// reads per plane and k16 step instead of eight scalar reads + a conversion chain.  Row pitch 192 bytes: the four rows x two 16-column groups a
// half-wave touches per read land on disjoint banks.
#define TN_PB 192
template <int MODE>   // 0: hi*hi + hi*lo + lo*hi + lo*lo, 1: without lo*lo (split-bf16), 2: hi*hi only (single-pass bf16)
__global__ __launch_bounds__(256) void grad_gemm_tn_kernel(const float* __restrict__ Y, const float* __restrict__ X,
                                                           float* __restrict__ partial, long long rows, int rps, int N,
                                                           int K) {
  constexpr int PT = 32 * TN_PB, YH = 0, YL = PT, XH = 2 * PT, XL = 3 * PT;
  __shared__ __attribute__((aligned(16))) char tsm[4 * PT + 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tk = K / 64;
  const int n0 = (blockIdx.x / tk) * 64, k0 = (blockIdx.x % tk) * 64;
  const long long r0 = (long long)blockIdx.y * rps;
  const long long r1 = r0 + rps < rows ? r0 + rps : rows;
  const int wn = (wave >> 1) * 32, wk = (wave & 1) * 32;
  // loader: thread -> (row = tid >> 4 (+16), float4 column = tid & 15)
  const int lr = tid >> 4, lc = (tid & 15) * 4;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int nrows = (int)(r1 - r0);   // <= rps
  const float* Yb = Y + r0 * N + n0 + lc;
  const float* Xb = X + r0 * K + k0 + lc;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 py0 = zero4, py1 = zero4, px0 = zero4, px1 = zero4;
  if (lr < nrows) {
    py0 = *reinterpret_cast<const float4*>(Yb + (long long)lr * N);
    px0 = *reinterpret_cast<const float4*>(Xb + (long long)lr * K);
  }
  if (lr + 16 < nrows) {
    py1 = *reinterpret_cast<const float4*>(Yb + (long long)(lr + 16) * N);
    px1 = *reinterpret_cast<const float4*>(Xb + (long long)(lr + 16) * K);
  }
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  auto stage = [&](const float4 v, int hoff, int loff, int row) {
    unsigned h0, l0, h1, l1;
    pl_split2(v.x, v.y, h0, l0);
    pl_split2(v.z, v.w, h1, l1);
    *(u32x2_*)(tsm + hoff + row * TN_PB + lc * 2) = u32x2_{h0, h1};
    if (MODE != 2) *(u32x2_*)(tsm + loff + row * TN_PB + lc * 2) = u32x2_{l0, l1};
  };
  for (int rb = 0; rb < nrows; rb += 32) {
    __syncthreads();
    stage(py0, YH, YL, lr);
    stage(py1, YH, YL, lr + 16);
    stage(px0, XH, XL, lr);
    stage(px1, XH, XL, lr + 16);
    __syncthreads();
    py0 = py1 = px0 = px1 = zero4;
    const int ra = rb + 32 + lr;
    if (ra < nrows) {
      py0 = *reinterpret_cast<const float4*>(Yb + (long long)ra * N);
      px0 = *reinterpret_cast<const float4*>(Xb + (long long)ra * K);
    }
    if (ra + 16 < nrows) {
      py1 = *reinterpret_cast<const float4*>(Yb + (long long)(ra + 16) * N);
      px1 = *reinterpret_cast<const float4*>(Xb + (long long)(ra + 16) * K);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const PlFrag a = pl_rd_tr(tsm, YH, MODE != 2 ? YL : YH, TN_PB, 16 * t, wn, lane);   // lane (n, kk): rows 16 t + 8 kk .. + 7 of column wn + n
      const PlFrag b = pl_rd_tr(tsm, XH, MODE != 2 ? XL : XH, TN_PB, 16 * t, wk, lane);
      if (MODE != 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, acc, 0, 0, 0);
      }
      if (MODE == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.l, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
    }
  }
  float* out = partial + (long long)blockIdx.y * N * K;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n0 + wn + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    out[(long long)n * K + k0 + wk + (lane & 31)] = acc[r];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
namespace {

inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

struct Dims {
  int B, S, hist, N, C, d, ffn, nl, H, L, M, R, T;   // L, M: the largest window (tokens per video, rows per step)
  int W, single;                                      // window length in frames; growing-window (single-step) rollouter
  // step s sees frames f0(s) .. f0(s) + Lw(s)/N - 1 (single_step_slotformer.py:79-88: the window grows to W frames first)
  int Lw(int s) const { return (single ? (s + 1 < W ? s + 1 : W) : W) * N; }
  int f0(int s) const { return single ? (s + 1 > W ? s + 1 - W : 0) : s; }
  size_t off(int s) const {   // rows of all steps before s in the stacked buffers
    size_t r = 0;
    for (int i = 0; i < s; ++i) r += (size_t)B * Lw(i);
    return r;
  }
};

// workspace carve-up (in floats); identical for forward and backward
struct Ws {
  float *slots_all, *tok_all, *xf, *xlast, *tmp;
  float *xin[16], *a1[16], *qkv[16], *ctx[16], *xmid[16], *a2[16], *hdn[16];
  // backward only
  float *dtok, *dpred, *dx, *dx2, *dctx, *dxlast;
  float *dqkv[16], *dao[16], *dpre[16], *dfo[16], *da1[16], *da2[16];
  float *wt_in[16], *wt_o[16], *wt_1[16], *wt_2[16], *wt_inproj, *wt_outproj;
  float* partial;
  size_t partial_floats;
  size_t total_floats;
};

inline size_t tn_partial_floats(const Dims& D);

Ws carve(const Dims& D, float* base) {
  Ws w;
  memset(&w, 0, sizeof(w));
  size_t off = 0;
  auto take = [&](size_t n) {
    float* p = base ? base + off : nullptr;
    off += (n + 63) & ~(size_t)63;
    return p;
  };
  const size_t M = D.M, R = D.R, S = D.S, d = D.d, f = D.ffn, Mt = D.off(D.S);
  w.slots_all = take((size_t)D.T * R * D.C);
  w.tok_all = take((size_t)D.T * R * d);
  w.xf = take(Mt * d);
  w.xlast = take(S * R * d);
  w.tmp = take(M * (f > 3 * d ? f : 3 * d));
  for (int l = 0; l < D.nl; ++l) {
    w.xin[l] = take(Mt * d);
    w.a1[l] = take(Mt * d);
    w.qkv[l] = take(Mt * 3 * d);
    w.ctx[l] = take(Mt * d);
    w.xmid[l] = take(Mt * d);
    w.a2[l] = take(Mt * d);
    w.hdn[l] = take(Mt * f);
  }
  w.dtok = take((size_t)D.T * R * d);
  w.dpred = take(S * R * D.C);
  w.dx = take(M * d);
  w.dx2 = take(M * d);
  w.dctx = take(M * d);
  w.dxlast = take(R * d);
  for (int l = 0; l < D.nl; ++l) {
    w.dqkv[l] = take(Mt * 3 * d);
    w.dao[l] = take(Mt * d);
    w.dpre[l] = take(Mt * f);
    w.dfo[l] = take(Mt * d);
    w.da1[l] = take(Mt * d);
    w.da2[l] = take(Mt * d);
    w.wt_in[l] = take(3 * d * d);
    w.wt_o[l] = take(d * d);
    w.wt_1[l] = take(f * d);
    w.wt_2[l] = take(f * d);
  }
  w.wt_inproj = take((size_t)D.C * d);
  w.wt_outproj = take((size_t)D.C * d);
  w.partial_floats = tn_partial_floats(D);
  w.partial = take(w.partial_floats);
  w.total_floats = off;
  return w;
}

// number of row splits of a weight-gradient contraction: enough workgroups to fill the chip a few times over
inline int tn_splits(long long rows, int N, int K) {
  const long long tiles = (long long)(N / 64) * (K / 64);
  long long s = (512 + tiles - 1) / tiles;
  const long long max_s = (rows + 63) / 64;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return (int)s;
}
inline size_t tn_partial_floats(const Dims& D) {
  const long long rows = (long long)D.off(D.S);
  size_t best = 0;
  auto upd = [&](long long r, int N, int K) {
    const size_t n = (size_t)tn_splits(r, N, K) * N * K;
    if (n > best) best = n;
  };
  upd(rows, 3 * D.d, D.d);
  upd(rows, D.d, D.d);
  upd(rows, D.ffn, D.d);
  upd(rows, D.d, D.ffn);
  upd((long long)D.T * D.R, D.d, D.C);
  upd((long long)D.S * D.R, D.C, D.d);
  // column-sum / LayerNorm partials share the buffer: at most 512 groups x 2 x max width
  const size_t cs = (size_t)512 * 2 * (size_t)(D.ffn > 3 * D.d ? D.ffn : 3 * D.d);
  return best > cs ? best : cs;
}

int check_model(const sf_rollouter* m, Dims& D, int B, int pred_len) {
  SF_REQUIRE(m && m->layers, "null model");
  SF_REQUIRE(m->norm_first, "training needs norm_first layers (all reference configurations)");
  SF_REQUIRE(B > 0 && pred_len > 0, "bad sizes");
  D.B = B; D.S = pred_len; D.W = m->window_len; D.single = m->single_step ? 1 : 0; D.hist = D.single ? 1 : D.W;
  D.N = m->num_slots; D.C = m->slot_size; D.d = m->d_model;
  D.ffn = m->ffn_dim; D.nl = m->num_layers; D.H = m->num_heads;
  D.L = D.W * D.N; D.M = B * D.L; D.R = B * D.N; D.T = D.hist + pred_len;
  SF_REQUIRE(D.nl >= 1 && D.nl <= 16, "1..16 layers");
  SF_REQUIRE(D.d % 64 == 0 && D.ffn % 64 == 0 && D.C % 64 == 0, "slot_size, d_model and ffn_dim must be multiples of 64");
  SF_REQUIRE(D.d <= 1024, "d_model <= 1024");
  SF_REQUIRE(D.d % D.H == 0 && D.d / D.H <= 64, "head_dim <= 64");
  SF_REQUIRE(D.L <= 128, "window of at most 128 tokens");
  return 0;
}

int launch_transpose(const float* in, float* out, int R, int Cn, hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(Cn, 32), cdiv(R, 32)), dim3(256), 0, st, in, out, R, Cn);
  SF_CHECK_LAUNCH();
  return 0;
}

int gemm(const float* A, const float* W, const float* bias, const float* res, float* Cc, int M, int N, int K, int relu,
         hipStream_t st) {
  return sf_linear_ex(A, sf_rows(K), W, bias, nullptr, nullptr, 0.f, res, sf_rows(N), 0, Cc, sf_rows(N), M, N, K, relu, st);
}

int attn_lds_bytes(int L, int hd, bool bwd) {
  return (int)(((bwd ? 4 : 3) * L * (hd + 1) + (bwd ? 2 : 1) * L * (L + 1)) * sizeof(float));
}

inline bool attn_use_mfma(int L, int hd) {
  return (hd == 32 || hd == 64) && L <= 96;
}
template <class Kern>
int set_lds(Kern kern, size_t bytes) {
  if (bytes <= 64 * 1024) return 0;
  return sf_ensure_dyn_lds((const void*)kern, (size_t)160 * 1024);
}
int launch_attn_fwd(const float* qkv, float* ctx, int B, int H, int L, int d, uint32_t sseed, uint32_t thr, float inv_keep,
                    hipStream_t st) {
  const int hd = d / H;
  const float scale = 1.f / sqrtf((float)hd);
  if (attn_use_mfma(L, hd)) {
    const int Lp = (L + 31) & ~31;
    const size_t bytes = ((size_t)3 * Lp * (hd + 1) + (size_t)Lp * (Lp + 1)) * sizeof(float);
    if (hd == 32) {
      SF_TRY(set_lds(attn_train_fwd_mfma_kernel<32>, bytes));
      hipLaunchKernelGGL(attn_train_fwd_mfma_kernel<32>, dim3(H, B), dim3(256), bytes, st, qkv, ctx, L, Lp, d, scale, sseed, thr,
                         inv_keep);
    } else {
      SF_TRY(set_lds(attn_train_fwd_mfma_kernel<64>, bytes));
      hipLaunchKernelGGL(attn_train_fwd_mfma_kernel<64>, dim3(H, B), dim3(256), bytes, st, qkv, ctx, L, Lp, d, scale, sseed, thr,
                         inv_keep);
    }
  } else {
    hipLaunchKernelGGL(attn_train_fwd_kernel, dim3(H, B), dim3(256), attn_lds_bytes(L, hd, false), st, qkv, ctx, L, d, hd, scale,
                       sseed, thr, inv_keep);
  }
  SF_CHECK_LAUNCH();
  return 0;
}
int launch_attn_bwd(const float* qkv, const float* dctx, float* dqkv, int B, int H, int L, int d, uint32_t sseed, uint32_t thr,
                    float inv_keep, hipStream_t st) {
  const int hd = d / H;
  const float scale = 1.f / sqrtf((float)hd);
  if (attn_use_mfma(L, hd)) {
    const int Lp = (L + 31) & ~31;
    const size_t bytes = ((size_t)4 * Lp * (hd + 1) + (size_t)2 * Lp * (Lp + 1)) * sizeof(float);
    if (hd == 32) {
      SF_TRY(set_lds(attn_train_bwd_mfma_kernel<32>, bytes));
      hipLaunchKernelGGL(attn_train_bwd_mfma_kernel<32>, dim3(H, B), dim3(256), bytes, st, qkv, dctx, dqkv, L, Lp, d, scale, sseed,
                         thr, inv_keep);
    } else {
      SF_TRY(set_lds(attn_train_bwd_mfma_kernel<64>, bytes));
      hipLaunchKernelGGL(attn_train_bwd_mfma_kernel<64>, dim3(H, B), dim3(256), bytes, st, qkv, dctx, dqkv, L, Lp, d, scale, sseed,
                         thr, inv_keep);
    }
  } else {
    hipLaunchKernelGGL(attn_train_bwd_kernel, dim3(H, B), dim3(256), attn_lds_bytes(L, hd, true), st, qkv, dctx, dqkv, L, d, hd,
                       scale, sseed, thr, inv_keep);
  }
  SF_CHECK_LAUNCH();
  return 0;
}

// dW = Y^T X over `rows` rows, written to dW [N, K]
int grad_weight(const float* Y, const float* X, float* dW, long long rows, int N, int K, const Ws& w, hipStream_t st) {
  const int splits = tn_splits(rows, N, K);
  int rps = (int)((rows + splits - 1) / splits);
  rps = (rps + 31) & ~31;
  float* dst = splits == 1 ? dW : w.partial;
  const dim3 grid((N / 64) * (K / 64), splits);
  if (sf_get_precision() == 0)
    hipLaunchKernelGGL(grad_gemm_tn_kernel<0>, grid, dim3(256), 0, st, Y, X, dst, rows, rps, N, K);
  else if (sf_get_precision() == 1)
    hipLaunchKernelGGL(grad_gemm_tn_kernel<1>, grid, dim3(256), 0, st, Y, X, dst, rows, rps, N, K);
  else
    hipLaunchKernelGGL(grad_gemm_tn_kernel<2>, grid, dim3(256), 0, st, Y, X, dst, rows, rps, N, K);
  SF_CHECK_LAUNCH();
  if (splits > 1) {
    const long long n4 = (long long)N * K / 4;
    launch_reduce_partials(w.partial, dW, splits, n4, st);
    SF_CHECK_LAUNCH();
  }
  return 0;
}

int grad_bias(const float* Y, float* db, long long rows, int n, const Ws& w, hipStream_t st) {
  int G = (int)((rows + 127) / 128);
  if (G > 512) G = 512;
  const int rpg = (int)((rows + G - 1) / G);
  G = (int)((rows + rpg - 1) / rpg);
  int cols = 1;
  while (cols < n && cols < 256) cols <<= 1;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(n, cols), G), dim3(256), 0, st, Y, w.partial, rows, rpg, n);
  SF_CHECK_LAUNCH();
  launch_reduce_partials(w.partial, db, G, (long long)n / 4, st);
  SF_CHECK_LAUNCH();
  return 0;
}

int grad_ln(const float* x, const float* dy, float* dgamma, float* dbeta, long long rows, int D, float eps, const Ws& w,
            hipStream_t st) {
  int G = (int)((rows + 63) / 64);
  if (G > 512) G = 512;
  const int rpg = (int)((rows + G - 1) / G);
  G = (int)((rows + rpg - 1) / rpg);
  hipLaunchKernelGGL(ln_param_partial_kernel, dim3(G), dim3(256), 0, st, x, dy, w.partial, rows, rpg, D, eps);
  SF_CHECK_LAUNCH();
  // partial is [G][2][D]: one reduction over both rows, the first D columns to d_gamma and the rest to d_beta
  if ((((uintptr_t)dgamma | (uintptr_t)dbeta) & 15) == 0) {
    launch_reduce_partials(w.partial, dgamma, G, (long long)2 * D / 4, st, dbeta, (long long)D / 4);
    SF_CHECK_LAUNCH();
    return 0;
  }
  // destinations that are not 16-byte aligned (views at odd offsets of a flat bucket): through a [2][D] scratch behind the partials
  float* both = w.partial + (size_t)G * 2 * D;
  launch_reduce_partials(w.partial, both, G, (long long)2 * D / 4, st);
  SF_CHECK_LAUNCH();
  hipError_t e = hipMemcpyAsync(dgamma, both, D * sizeof(float), hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(dbeta, both + D, D * sizeof(float), hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return sf_set_err((int)e, hipGetErrorString(e), __FILE__, __LINE__);
  return 0;
}

}  // namespace

// ---- the same building blocks for the other training nodes (slot_attn_train.hip); declared in sf_internal.h ----
size_t sf_grad_partial_floats(long long rows, int N, int K) {
  const size_t a = (size_t)tn_splits(rows, N, K) * N * K;
  const size_t b = (size_t)513 * 2 * (size_t)(N > K ? N : K);   // column-sum / LayerNorm partials
  return a > b ? a : b;
}
int sf_grad_weight_ex(const float* Y, const float* X, float* dW, long long rows, int N, int K, float* partial, hipStream_t st) {
  SF_REQUIRE(N % 64 == 0 && K % 64 == 0, "weight-gradient contraction needs multiples of 64");
  Ws w;
  w.partial = partial;
  return grad_weight(Y, X, dW, rows, N, K, w, st);
}
int sf_grad_bias_ex(const float* Y, float* db, long long rows, int n, float* partial, hipStream_t st) {
  Ws w;
  w.partial = partial;
  return grad_bias(Y, db, rows, n, w, st);
}
int sf_grad_ln_ex(const float* x, const float* dy, float* dgamma, float* dbeta, long long rows, int D, float eps, float* partial,
                  hipStream_t st) {
  Ws w;
  w.partial = partial;
  return grad_ln(x, dy, dgamma, dbeta, rows, D, eps, w, st);
}
int sf_ln_bwd_ex(const float* x, const float* dy, const float* gamma, const float* dres, float* out, long long rows, int D,
                 float eps, hipStream_t st) {
  SF_REQUIRE(D % 4 == 0 && D <= 1024, "LayerNorm width");
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, dy, gamma, dres, out, (float*)nullptr, 0u, 0u, 1.f,
                     (int)rows, D, eps);
  SF_CHECK_LAUNCH();
  return 0;
}
int sf_transpose_ex(const float* in, float* out, int R, int Cn, hipStream_t st) { return launch_transpose(in, out, R, Cn, st); }
int sf_relu_bwd_ex(float* dh, const float* h, long long n, hipStream_t st) {
  hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, st, dh, h, n / 4, 1.f);
  SF_CHECK_LAUNCH();
  return 0;
}

extern "C" {

// ---- differentiable building blocks for the slot-level layers (predictor, kernel distribution; savi.py:190-200,
// predictor.py:47-73): backward of y = act(x W^T + b) and of LayerNorm, on the same kernels as the big nodes ----
size_t sf_linear_bwd_workspace_bytes(long long M, int N, int K) {
  return (sf_grad_partial_floats(M, N, K) + (size_t)N * K + 128) * sizeof(float) + 256;
}
// dy [M,N] is overwritten when relu != 0 (masked by y > 0).  dx [M,K] (may be NULL), dW [N,K], db [N] (may be NULL).
int sf_linear_bwd_f32(const float* x, const float* W, const float* y, float* dy, float* dx, float* dW, float* db, long long M,
                      int N, int K, int relu, void* ws, size_t ws_bytes, void* stream) {
  SF_REQUIRE(x && W && dy && dW && ws && M > 0, "null pointer");
  SF_REQUIRE(N % 64 == 0 && K % 64 == 0, "linear backward needs multiples of 64");
  SF_REQUIRE(ws_bytes >= sf_linear_bwd_workspace_bytes(M, N, K), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  float* wt = partial + ((sf_grad_partial_floats(M, N, K) + 63) & ~(size_t)63);
  if (relu) {
    SF_REQUIRE(y != nullptr, "relu backward needs the forward output");
    SF_TRY(sf_relu_bwd_ex(dy, y, M * N, st));
  }
  SF_TRY(sf_grad_weight_ex(dy, x, dW, M, N, K, partial, st));
  if (db) SF_TRY(sf_grad_bias_ex(dy, db, M, N, partial, st));
  if (dx) {
    SF_TRY(launch_transpose(W, wt, N, K, st));   // [N,K] -> [K,N]
    SF_TRY(sf_linear_ex(dy, sf_rows(N), wt, nullptr, nullptr, nullptr, 0.f, nullptr, sf_rows(K), 0, dx, sf_rows(K), (int)M, K, N, 0, st));
  }
  return 0;
}
// attention core of nn.MultiheadAttention for training (qkv [B*L, 3d] -> ctx [B*L, d]; dropout on the softmax weights, masks
// a pure function of (seed, element)) and its backward (recomputes the probabilities; nothing saved but qkv)
int sf_mha_train_fwd_f32(const float* qkv, float* ctx, int B, int L, int d_model, int num_heads, float dropout_p,
                         unsigned long long seed, void* stream) {
  SF_REQUIRE(qkv && ctx && B > 0 && L > 0 && L <= 128 && num_heads > 0 && d_model % num_heads == 0 && d_model / num_heads <= 64,
             "bad attention shape");
  SF_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p in [0, 1)");
  return launch_attn_fwd(qkv, ctx, B, num_heads, L, d_model, site_seed(seed, 0, 0, SITE_ATTN_P), drop_thresh(dropout_p),
                         1.f / (1.f - dropout_p), (hipStream_t)stream);
}
int sf_mha_train_bwd_f32(const float* qkv, const float* d_ctx, float* d_qkv, int B, int L, int d_model, int num_heads,
                         float dropout_p, unsigned long long seed, void* stream) {
  SF_REQUIRE(qkv && d_ctx && d_qkv && B > 0 && L > 0 && L <= 128 && num_heads > 0 && d_model % num_heads == 0 &&
                 d_model / num_heads <= 64, "bad attention shape");
  return launch_attn_bwd(qkv, d_ctx, d_qkv, B, num_heads, L, d_model, site_seed(seed, 0, 0, SITE_ATTN_P), drop_thresh(dropout_p),
                         1.f / (1.f - dropout_p), (hipStream_t)stream);
}
// y = res + dropout(x) (res may be NULL); the backward of the dropout is the same call on the gradient
int sf_dropout_f32(const float* x, const float* res, float* y, long long n, float dropout_p, unsigned long long seed, void* stream) {
  SF_REQUIRE(x && y && n >= 0 && n % 4 == 0 && n < (1LL << 32), "bad dropout arguments");
  SF_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p in [0, 1)");
  if (n == 0) return 0;
  hipLaunchKernelGGL(residual_dropout_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, res, y, n / 4,
                     site_seed(seed, 0, 0, SITE_ATTN_O), drop_thresh(dropout_p), 1.f / (1.f - dropout_p));
  SF_CHECK_LAUNCH();
  return 0;
}

// torch.optim.Adam (no amsgrad, no weight decay: the reference's optimiser, slotformer_clevrer_params.py:16-19) over ONE flat
// fp32 bucket: p, g, m, v [n]; step = the 1-based step count.  One launch per optimiser step.
__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                        float bc1, float bc2_sqrt) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  // torch: denom = sqrt(v) / sqrt(bias_correction2) + eps;  p -= (lr / bias_correction1) * m / denom
  p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
}
int sf_adam_flat_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, int step, float lr, float beta1,
                     float beta2, float eps, void* stream) {
  SF_REQUIRE(param && grad && exp_avg && exp_avg_sq && n >= 0 && step >= 1, "bad Adam arguments");
  if (n == 0) return 0;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adam_flat_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr,
                     beta1, beta2, eps, bc1, sqrtf(bc2));
  SF_CHECK_LAUNCH();
  return 0;
}

size_t sf_layernorm_bwd_workspace_bytes(int D) { return ((size_t)513 * 2 * D + 128) * sizeof(float) + 256; }
int sf_layernorm_bwd_f32(const float* x, const float* dy, const float* gamma, float* dx, float* dgamma, float* dbeta, long long rows,
                         int D, float eps, void* ws, size_t ws_bytes, void* stream) {
  SF_REQUIRE(x && dy && gamma && dx && dgamma && dbeta && ws && rows > 0, "null pointer");
  SF_REQUIRE(ws_bytes >= sf_layernorm_bwd_workspace_bytes(D), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  SF_TRY(sf_grad_ln_ex(x, dy, dgamma, dbeta, rows, D, eps, partial, st));
  return sf_ln_bwd_ex(x, dy, gamma, nullptr, dx, rows, D, eps, st);
}

size_t sf_rollout_train_workspace_bytes(const sf_rollouter* m, int B, int pred_len) {
  Dims D;
  if (check_model(m, D, B, pred_len) != 0) return 0;
  return carve(D, nullptr).total_floats * sizeof(float) + 256;
}

int sf_rollout_train_fwd_f32(const sf_rollouter* m, const float* x, float* pred, int B, int pred_len, float dropout_p,
                             unsigned long long seed, void* ws, size_t ws_bytes, void* stream) {
  Dims D;
  SF_TRY(check_model(m, D, B, pred_len));
  SF_REQUIRE(x && pred && ws, "null pointer");
  SF_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p in [0, 1)");
  hipStream_t st = (hipStream_t)stream;
  float* base = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  const Ws w = carve(D, base);
  SF_REQUIRE(w.total_floats * sizeof(float) + 256 <= ws_bytes, "workspace too small");
  const int d = D.d, f = D.ffn, R = D.R, N = D.N, C = D.C, hd = d / D.H;
  const uint32_t thr = drop_thresh(dropout_p);
  const float inv_keep = 1.f / (1.f - dropout_p);
  const float scale = 1.f / sqrtf((float)hd);
  const int albytes = attn_lds_bytes(D.L, hd, false);
  SF_REQUIRE(albytes <= 64 * 1024, "attention tile does not fit LDS");

  // burn-in frames -> frame-major slots_all, then their in-projections
  for (int t = 0; t < D.hist; ++t)
    SF_TRY(sf_copy_rows_ex(x, sf_rows_batched(C, N, (long long)D.hist * N * C, (long long)t * N * C),
                           w.slots_all + (size_t)t * R * C, sf_rows(C), R, C, st));
  SF_TRY(gemm(w.slots_all, m->in_proj_w, m->in_proj_b, nullptr, w.tok_all, D.hist * R, d, C, 0, st));

  for (int s = 0; s < D.S; ++s) {
    const size_t so = D.off(s);
    const int L = D.Lw(s), M = B * L;   // this step's window
    if (s > 0) {
      const int fr = D.hist + s - 1;
      SF_TRY(gemm(w.slots_all + (size_t)fr * R * C, m->in_proj_w, m->in_proj_b, nullptr, w.tok_all + (size_t)fr * R * d, R,
                  d, C, 0, st));
    }
    hipLaunchKernelGGL(window_assemble_kernel, dim3(cdiv((long long)M * d / 4, 256)), dim3(256), 0, st,
                       w.tok_all + (size_t)D.f0(s) * R * d, m->pe_tok + (size_t)(D.L - L) * d, w.xin[0] + so * d, B, L, N, d / 4);
    SF_CHECK_LAUNCH();
    for (int l = 0; l < D.nl; ++l) {
      const sf_tfm_layer& ly = m->layers[l];
      float* xin = w.xin[l] + so * d;
      float* a1 = w.a1[l] + so * d;
      float* qkv = w.qkv[l] + so * 3 * d;
      float* ctx = w.ctx[l] + so * d;
      float* xmid = w.xmid[l] + so * d;
      float* a2 = w.a2[l] + so * d;
      float* hdn = w.hdn[l] + so * f;
      float* xnext = (l + 1 < D.nl ? w.xin[l + 1] : w.xf) + so * d;
      SF_TRY(sf_layernorm_ex(xin, sf_rows(d), ly.norm1_g, ly.norm1_b, a1, sf_rows(d), M, d, 1e-5f, st));
      SF_TRY(gemm(a1, ly.in_proj_w, ly.in_proj_b, nullptr, qkv, M, 3 * d, d, 0, st));
      SF_TRY(launch_attn_fwd(qkv, ctx, B, D.H, L, d, site_seed(seed, s, l, SITE_ATTN_P), thr, inv_keep, st));
      SF_TRY(sf_linear_dropout_ex(ctx, ly.out_proj_w, ly.out_proj_b, xin, xmid, M, d, d, 0, site_seed(seed, s, l, SITE_ATTN_O), thr,
                                  inv_keep, st));
      SF_TRY(sf_layernorm_ex(xmid, sf_rows(d), ly.norm2_g, ly.norm2_b, a2, sf_rows(d), M, d, 1e-5f, st));
      SF_TRY(sf_linear_dropout_ex(a2, ly.lin1_w, ly.lin1_b, nullptr, hdn, M, f, d, 1, site_seed(seed, s, l, SITE_FFN_H), thr, inv_keep,
                                  st));
      SF_TRY(sf_linear_dropout_ex(hdn, ly.lin2_w, ly.lin2_b, xmid, xnext, M, d, f, 0, site_seed(seed, s, l, SITE_FFN_O), thr,
                                  inv_keep, st));
    }
    // last N tokens of every video -> xlast[s]; prediction = out_proj(xlast)  (slotformer.py:121)
    SF_TRY(sf_copy_rows_ex(w.xf + so * d, sf_rows_batched(d, N, (long long)L * d, (long long)(L - N) * d),
                           w.xlast + (size_t)s * R * d, sf_rows(d), R, d, st));
    float* fr = w.slots_all + (size_t)(D.hist + s) * R * C;
    SF_TRY(gemm(w.xlast + (size_t)s * R * d, m->out_proj_w, m->out_proj_b, nullptr, fr, R, C, d, 0, st));
    SF_TRY(sf_copy_rows_ex(fr, sf_rows(C), pred, sf_rows_batched(C, N, (long long)D.S * N * C, (long long)s * N * C), R, C,
                           st));
  }
  return 0;
}

int sf_rollout_train_bwd_f32(const sf_rollouter* m, const float* d_pred, float* d_x, const sf_rollouter_grads* g, int B,
                             int pred_len, float dropout_p, unsigned long long seed, void* ws, size_t ws_bytes,
                             void* stream) {
  Dims D;
  SF_TRY(check_model(m, D, B, pred_len));
  SF_REQUIRE(d_pred && g && g->layers && ws, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  float* base = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  const Ws w = carve(D, base);
  SF_REQUIRE(w.total_floats * sizeof(float) + 256 <= ws_bytes, "workspace too small");
  const int d = D.d, f = D.ffn, R = D.R, N = D.N, C = D.C, hd = d / D.H;
  const uint32_t thr = drop_thresh(dropout_p);
  const float inv_keep = 1.f / (1.f - dropout_p);
  const float scale = 1.f / sqrtf((float)hd);
  const int albytes = attn_lds_bytes(D.L, hd, true);
  SF_REQUIRE(albytes <= 64 * 1024, "attention tile does not fit LDS");

  // transposed weight copies: the data-gradient GEMMs run on the forward core (C = A . W^T)
  for (int l = 0; l < D.nl; ++l) {
    const sf_tfm_layer& ly = m->layers[l];
    SF_TRY(launch_transpose(ly.in_proj_w, w.wt_in[l], 3 * d, d, st));   // [3d,d] -> [d,3d]
    SF_TRY(launch_transpose(ly.out_proj_w, w.wt_o[l], d, d, st));
    SF_TRY(launch_transpose(ly.lin1_w, w.wt_1[l], f, d, st));           // [f,d] -> [d,f]
    SF_TRY(launch_transpose(ly.lin2_w, w.wt_2[l], d, f, st));           // [d,f] -> [f,d]
  }
  SF_TRY(launch_transpose(m->in_proj_w, w.wt_inproj, d, C, st));        // [d,C] -> [C,d]
  SF_TRY(launch_transpose(m->out_proj_w, w.wt_outproj, C, d, st));      // [C,d] -> [d,C]

  hipError_t e = hipMemsetAsync(w.dtok, 0, (size_t)D.T * R * d * sizeof(float), st);
  if (e != hipSuccess) return sf_set_err((int)e, hipGetErrorString(e), __FILE__, __LINE__);
  if (g->pe_tok) {
    e = hipMemsetAsync(g->pe_tok, 0, (size_t)D.L * d * sizeof(float), st);
    if (e != hipSuccess) return sf_set_err((int)e, hipGetErrorString(e), __FILE__, __LINE__);
  }
  // d_pred [B,S,N,C] -> step-major dpred [S][R][C]
  for (int s = 0; s < D.S; ++s)
    SF_TRY(sf_copy_rows_ex(d_pred, sf_rows_batched(C, N, (long long)D.S * N * C, (long long)s * N * C),
                           w.dpred + (size_t)s * R * C, sf_rows(C), R, C, st));

  for (int s = D.S - 1; s >= 0; --s) {
    const size_t so = D.off(s);
    const int L = D.Lw(s), M = B * L;   // this step's window
    const long long Md4 = (long long)M * d / 4;
    float* dp = w.dpred + (size_t)s * R * C;
    // feedback through the in-projection of the predicted frame (zero for the last step)
    if (s + 1 < D.S) {
      SF_TRY(gemm(w.dtok + (size_t)(D.hist + s) * R * d, w.wt_inproj, nullptr, dp, dp, R, C, d, 0, st));
    }
    // through out_proj into the last N tokens of the final layer output
    SF_TRY(gemm(dp, w.wt_outproj, nullptr, nullptr, w.dxlast, R, d, C, 0, st));
    hipLaunchKernelGGL(scatter_last_kernel, dim3(cdiv(Md4, 256)), dim3(256), 0, st, w.dxlast, w.dx, w.dfo[D.nl - 1] + so * d, B, L, N,
                       d / 4, site_seed(seed, s, D.nl - 1, SITE_FFN_O), thr, inv_keep);
    SF_CHECK_LAUNCH();
    float* dx = w.dx;     // gradient w.r.t. the current layer's output
    float* dxo = w.dx2;   // gradient w.r.t. the layer's mid-point (after the attention block)
    for (int l = D.nl - 1; l >= 0; --l) {
      const sf_tfm_layer& ly = m->layers[l];
      float* dfo = w.dfo[l] + so * d;     // x2 = xmid + drop(ffn_o): filled by the stage above
      float* dpre = w.dpre[l] + so * f;
      float* da2 = w.da2[l] + so * d;
      float* dao = w.dao[l] + so * d;
      float* dqkv = w.dqkv[l] + so * 3 * d;
      float* da1 = w.da1[l] + so * d;
      // through linear2, the hidden dropout and the ReLU in one GEMM: the saved hidden activation is the mask
      SF_TRY(sf_linear_masked_ex(dfo, w.wt_2[l], w.hdn[l] + so * f, thr ? inv_keep : 1.f, dpre, M, f, d, st));
      SF_TRY(gemm(dpre, w.wt_1[l], nullptr, nullptr, da2, M, d, f, 0, st));
      // xmid = xin + drop(attn_o): dxo = gradient w.r.t. xmid, dao = the same through the attention-output dropout
      hipLaunchKernelGGL(ln_bwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, w.xmid[l] + so * d, da2, ly.norm2_g, dx, dxo, dao,
                         site_seed(seed, s, l, SITE_ATTN_O), thr, inv_keep, M, d, 1e-5f);
      SF_CHECK_LAUNCH();
      SF_TRY(gemm(dao, w.wt_o[l], nullptr, nullptr, w.dctx, M, d, d, 0, st));
      SF_TRY(launch_attn_bwd(w.qkv[l] + so * 3 * d, w.dctx, dqkv, B, D.H, L, d, site_seed(seed, s, l, SITE_ATTN_P), thr, inv_keep,
                             st));
      SF_TRY(gemm(dqkv, w.wt_in[l], nullptr, nullptr, da1, M, d, 3 * d, 0, st));
      // dx = gradient w.r.t. this layer's input = the output of layer l-1, whose FFN-output dropout is folded in
      hipLaunchKernelGGL(ln_bwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, w.xin[l] + so * d, da1, ly.norm1_g, dxo, dx,
                         l > 0 ? w.dfo[l - 1] + so * d : (float*)nullptr, site_seed(seed, s, l > 0 ? l - 1 : 0, SITE_FFN_O), thr,
                         inv_keep, M, d, 1e-5f);
      SF_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(window_scatter_kernel, dim3(cdiv(Md4, 256)), dim3(256), 0, st, dx, w.dtok + (size_t)D.f0(s) * R * d, B, L,
                       N, d / 4);
    SF_CHECK_LAUNCH();
    if (g->pe_tok) {   // learnable position tables: this step's window used rows D.L - L .. D.L - 1 of the folded table
      hipLaunchKernelGGL(pe_grad_kernel, dim3(cdiv((long long)L * d / 4, 256)), dim3(256), 0, st, dx, g->pe_tok + (size_t)(D.L - L) * d, B,
                         L, d / 4);
      SF_CHECK_LAUNCH();
    }
  }

  // gradient w.r.t. the burn-in slots
  if (d_x) {
    SF_TRY(gemm(w.dtok, w.wt_inproj, nullptr, nullptr, w.tmp, D.hist * R, C, d, 0, st));
    for (int t = 0; t < D.hist; ++t)
      SF_TRY(sf_copy_rows_ex(w.tmp + (size_t)t * R * C, sf_rows(C), d_x,
                             sf_rows_batched(C, N, (long long)D.hist * N * C, (long long)t * N * C), R, C, st));
  }

  // parameter gradients: one contraction per weight over all stacked rows.  The in-projection saw frames 0 .. T-2 (the
  // last predicted frame is never fed back).
  const long long rows = (long long)D.off(D.S);
  const long long in_rows = (long long)(D.T - 1) * R;
  SF_TRY(grad_weight(w.dtok, w.slots_all, g->in_proj_w, in_rows, d, C, w, st));
  SF_TRY(grad_bias(w.dtok, g->in_proj_b, in_rows, d, w, st));
  SF_TRY(grad_weight(w.dpred, w.xlast, g->out_proj_w, (long long)D.S * R, C, d, w, st));
  SF_TRY(grad_bias(w.dpred, g->out_proj_b, (long long)D.S * R, C, w, st));
  for (int l = 0; l < D.nl; ++l) {
    const sf_tfm_layer_grads& gl = g->layers[l];
    SF_TRY(grad_weight(w.dqkv[l], w.a1[l], gl.in_proj_w, rows, 3 * d, d, w, st));
    SF_TRY(grad_bias(w.dqkv[l], gl.in_proj_b, rows, 3 * d, w, st));
    SF_TRY(grad_weight(w.dao[l], w.ctx[l], gl.out_proj_w, rows, d, d, w, st));
    SF_TRY(grad_bias(w.dao[l], gl.out_proj_b, rows, d, w, st));
    SF_TRY(grad_weight(w.dpre[l], w.a2[l], gl.lin1_w, rows, f, d, w, st));
    SF_TRY(grad_bias(w.dpre[l], gl.lin1_b, rows, f, w, st));
    SF_TRY(grad_weight(w.dfo[l], w.hdn[l], gl.lin2_w, rows, d, f, w, st));
    SF_TRY(grad_bias(w.dfo[l], gl.lin2_b, rows, d, w, st));
    SF_TRY(grad_ln(w.xin[l], w.da1[l], gl.norm1_g, gl.norm1_b, rows, d, 1e-5f, w, st));
    SF_TRY(grad_ln(w.xmid[l], w.da2[l], gl.norm2_g, gl.norm2_b, rows, d, 1e-5f, w, st));
  }
  return 0;
}

}  // extern "C"
