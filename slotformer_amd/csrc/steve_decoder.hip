// Kernels of the STEVE image side (second half): the dVAE tokenizer / detokenizer blocks
// (reference: slotformer/base_slots/models/dVAE.py, steve_utils.py:100-126) and the slot-conditioned Transformer
// decoder (steve_transformer.py).  The convolutions and linears run on the GEMM core (gemm.hip); this file holds
// what is specific to these models.
#include "../../include/slotformer_hip.h"
#include "sf_internal.h"
#include "bf16_planes.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int GN_P = 64;   // partial records per sample

// ---- GroupNorm(num_groups = 1) statistics: per sample, over all H*W*C elements (steve_utils.py:124-126) ----
// grid (GN_P, F): block (p, f) reduces its slice to (sum, sum of squares) in double.
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, double* __restrict__ part, long long n) {
  const int p = blockIdx.x, f = blockIdx.y, t = threadIdx.x;
  const long long n4 = n / 4, per = (n4 + GN_P - 1) / GN_P;
  const long long lo = p * per, hi = lo + per < n4 ? lo + per : n4;
  const f32x4* xs = (const f32x4*)(x + (long long)f * n);
  float s = 0.f, q = 0.f;
  for (long long i = lo + t; i < hi; i += 256) {
    const f32x4 v = xs[i];
    s += (v[0] + v[1]) + (v[2] + v[3]);
    q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  __shared__ double sh[2][4];
  double ds = (double)sf_sum64(s), dq = (double)sf_sum64(q);
  if ((t & 63) == 0) {
    sh[0][t >> 6] = ds;
    sh[1][t >> 6] = dq;
  }
  __syncthreads();
  if (t == 0) {
    part[((long long)f * GN_P + p) * 2 + 0] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    part[((long long)f * GN_P + p) * 2 + 1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
  }
}

// y = act((x - mean) * rstd * gamma[c] + beta[c]); NHWC in.  shuffle == 2 also applies nn.PixelShuffle(2)
// (dVAE.py:44,49): in channel c*4 + i*2 + j of pixel (yy, xx) -> out pixel (2yy + i, 2xx + j), channel c.
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const double* __restrict__ part,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ y, int H, int W, int C, float eps, int relu,
                                                       int shuffle) {
  const int f = blockIdx.y, t = threadIdx.x;
  const long long n = (long long)H * W * C;
  __shared__ float stat[2];
  if (t < 64) {
    double s = part[((long long)f * GN_P + t) * 2], q = part[((long long)f * GN_P + t) * 2 + 1];
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o, 64);
      q += __shfl_xor(q, o, 64);
    }
    if (t == 0) {
      const double mean = s / (double)n;
      const double var = q / (double)n - mean * mean;
      stat[0] = (float)mean;
      stat[1] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
    }
  }
  __syncthreads();
  const float mean = stat[0], rstd = stat[1];
  const long long i4 = (long long)blockIdx.x * 256 + t;
  if (i4 * 4 >= n) return;
  const long long e = i4 * 4;
  const int c = (int)(e % C);
  const f32x4 v = *(const f32x4*)(x + (long long)f * n + e);
  const f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
  f32x4 o = (v - mean) * rstd * g + b;
  if (relu) {
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = fmaxf(o[k], 0.f);
  }
  if (shuffle == 1) {
    *(f32x4*)(y + (long long)f * n + e) = o;
  } else {
    const long long pix = e / C;
    const int yy = (int)(pix / W), xx = (int)(pix - (long long)yy * W);
    const int Co = C / 4, co = c / 4;
    float* yo = y + (long long)f * n;
#pragma unroll
    for (int k = 0; k < 4; ++k)   // k = i*2 + j
      yo[((long long)(2 * yy + (k >> 1)) * (2 * W) + 2 * xx + (k & 1)) * Co + co] = o[k];
  }
}

// ---- attention of the slate Transformer decoder (steve_transformer.py:12-55): softmax(q k^T * hd^-0.5 [+ causal mask]) v.
// One thread per query row, keys/values streamed through LDS in tiles of 64 with an online softmax, so the sequence
// length (1 + 1023 image tokens at 128x128) is unbounded; the same kernel serves the cross-attention to the N slots.
// q rows [B][Lq] (leading dim ldq), k/v rows [B][Lk]; head h uses columns h*HD .. h*HD+HD-1 of each.
template <int HD>
__global__ __launch_bounds__(256) void slate_attn_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, float* __restrict__ out, int ldq,
                                                         int ldk, int ldv, int ldo, long long q_bs, long long k_bs,
                                                         long long v_bs, long long o_bs, int Lq, int Lk, int causal,
                                                         float scale) {
  __shared__ __attribute__((aligned(16))) float Ks[64][HD];
  __shared__ __attribute__((aligned(16))) float Vs[64][HD];
  const int t = threadIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int i = blockIdx.x * 256 + t;
  const bool live = i < Lq;
  const float* qr = q + (long long)b * q_bs + (long long)(live ? i : Lq - 1) * ldq + h * HD;
  float qv[HD], o[HD];
#pragma unroll
  for (int c = 0; c < HD; c += 4) {
    const f32x4 x = *(const f32x4*)(qr + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qv[c + e] = x[e] * scale;
      o[c + e] = 0.f;
    }
  }
  float m = -INFINITY, l = 0.f;
  const int kend = causal ? min(Lk, blockIdx.x * 256 + 256) : Lk;   // keys this block can see at all
  const float* kb = k + (long long)b * k_bs + h * HD;
  const float* vb = v + (long long)b * v_bs + h * HD;
  for (int k0 = 0; k0 < kend; k0 += 64) {
    constexpr int F4 = 64 * HD / 4;   // float4 per tile
    for (int idx = t; idx < F4; idx += 256) {
      const int r = idx / (HD / 4), c4 = idx - r * (HD / 4);
      const int j = min(k0 + r, Lk - 1);
      *(f32x4*)&Ks[r][4 * c4] = *(const f32x4*)(kb + (long long)j * ldk + 4 * c4);
      *(f32x4*)&Vs[r][4 * c4] = *(const f32x4*)(vb + (long long)j * ldv + 4 * c4);
    }
    __syncthreads();
    const int nj = min(64, kend - k0);
    for (int jj = 0; jj < nj; ++jj) {
      float sc = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) sc += qv[c] * Ks[jj][c];
      const int j = k0 + jj;
      if (live && (!causal || j <= i)) {
        const float mn = fmaxf(m, sc);
        const float corr = (m == -INFINITY) ? 0.f : expf(m - mn);
        const float pj = expf(sc - mn);
        l = l * corr + pj;
#pragma unroll
        for (int c = 0; c < HD; ++c) o[c] = o[c] * corr + pj * Vs[jj][c];
        m = mn;
      }
    }
    __syncthreads();
  }
  if (live) {
    const float inv = 1.0f / l;
    float* orow = out + (long long)b * o_bs + (long long)i * ldo + h * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 4) *(f32x4*)(orow + c) = f32x4{o[c] * inv, o[c + 1] * inv, o[c + 2] * inv, o[c + 3] * inv};
  }
}

// Causal self-attention for long sequences on the exact-f32 MFMA (flash style: online softmax over 64-key tiles, no
// L x L score matrix).  One workgroup per (64-query tile, head, batch); wave = (query block qb of 32, key half kh of the
// tile).  Scores are computed TRANSPOSED, S^T[key][query] = k q^T, so a lane holds 16 keys of ONE query column: the
// softmax statistics are per-lane reductions plus one exchange with lane ^ 32, and the probabilities feed the PV MFMA
// straight from registers (register r = the key pair (k_r, k_r + 4) of the B operand).  The two key halves of a query
// block keep separate running (max, sum, O) and are merged through LDS at the end.
// TRAIN (row N1, sf_slate_attention_train_fwd_f32): dropout on the attention weights -- the register-resident probabilities are
// scaled by a hashed keep mask before they feed the PV MFMA, the row sums stay undropped -- and the row log-sum-exp is stored.
struct SlateTrainArgs {
  float* lse;                       // [B][H][L]
  unsigned drop_seed, drop_thresh;  // element ((b*H + h)*L + q)*L + k, as in slate_attn_bwd.hip
  float drop_scale;
};
template <int HD, bool TRAIN>
__global__ __launch_bounds__(256) void slate_flash_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v, float* __restrict__ out, int ldq,
                                                          int ldk, int ldv, int ldo, long long q_bs, long long k_bs,
                                                          long long v_bs, long long o_bs, int L, float scale, SlateTrainArgs ta) {
  constexpr int P = HD + 4;                 // f32 row pitch of the tiles ((HD+4)/4 odd for HD = 16, 32, 48, 64)
  constexpr int CB = (HD + 31) / 32;        // 32-channel blocks of the output
  extern __shared__ __attribute__((aligned(16))) float fsm[];
  float* Qs = fsm;                          // [64][P] (scaled)
  float* Ks = Qs + 64 * P;                  // [64][P]
  float* Vs = Ks + 64 * P;                  // [64][CB*32 + 4]  (channels padded to the MFMA block)
  constexpr int PV = CB * 32 + 4;
  float* SM = Vs + 64 * PV;                 // [4][32] running max, [4][32] running sum
  float* OT = SM + 2 * 4 * 32;              // [4 waves][32 queries][PV]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int q0 = qt * 64;
  const float* qb_ = q + (long long)b * q_bs + h * HD;
  const float* kb_ = k + (long long)b * k_bs + h * HD;
  const float* vb_ = v + (long long)b * v_bs + h * HD;
  // Q tile (scaled) and zeroed channel padding of V
  for (int idx = t; idx < 64 * (HD / 4); idx += 256) {
    const int r = idx / (HD / 4), c4 = idx - r * (HD / 4);
    const f32x4 x = *(const f32x4*)(qb_ + (long long)min(q0 + r, L - 1) * ldq + 4 * c4);
    *(f32x4*)(Qs + r * P + 4 * c4) = x * scale;
  }
  if (CB * 32 > HD) {
    for (int idx = t; idx < 64 * (CB * 32 - HD); idx += 256) {
      const int r = idx / (CB * 32 - HD), c = idx - r * (CB * 32 - HD);
      Vs[r * PV + HD + c] = 0.f;
    }
  }
  const int qblk = wave >> 1, kh = wave & 1;
  const int qcol = q0 + qblk * 32 + (lane & 31);            // this lane's query
  f32x16 oacc[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[cb][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const int ntiles = qt + 1;                                // causal: key tiles 0 .. qt
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();                                        // previous tile fully consumed (and Q written)
    const int k0 = kt * 64;
    for (int idx = t; idx < 64 * (HD / 4); idx += 256) {
      const int r = idx / (HD / 4), c4 = idx - r * (HD / 4);
      const int j = min(k0 + r, L - 1);
      *(f32x4*)(Ks + r * P + 4 * c4) = *(const f32x4*)(kb_ + (long long)j * ldk + 4 * c4);
      *(f32x4*)(Vs + r * PV + 4 * c4) = *(const f32x4*)(vb_ + (long long)j * ldv + 4 * c4);
    }
    __syncthreads();
    // S^T block: keys k0 + 32 kh + .., queries of block qblk
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
    const float* kp = Ks + (kh * 32 + (lane & 31)) * P + 4 * (lane >> 5);
    const float* qp = Qs + (qblk * 32 + (lane & 31)) * P + 4 * (lane >> 5);
#pragma unroll
    for (int c = 0; c < HD / 8; ++c) {
      const f32x4 a = *(const f32x4*)(kp + c * 8), bq = *(const f32x4*)(qp + c * 8);
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2], bq[s2], sacc, 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      sacc[r] = (key <= qcol && key < L) ? sacc[r] : -INFINITY;
      mx = fmaxf(mx, sacc[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float corr = (m == -INFINITY) ? 0.f : expf(m - mn);
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sacc[r] = (mn == -INFINITY) ? 0.f : expf(sacc[r] - mn);
      sum += sacc[r];
    }
    sum += __shfl_xor(sum, 32, 64);
    l = l * corr + sum;
    m = mn;
    if constexpr (TRAIN) {
      if (ta.drop_thresh) {
        const unsigned base = (unsigned)((((long long)b * gridDim.y + h) * L + min(qcol, L - 1)) * L);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          sacc[r] = (sf_mix32((base + (unsigned)key) ^ ta.drop_seed) >> 8) >= ta.drop_thresh ? sacc[r] * ta.drop_scale : 0.f;
        }
      }
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[cb][r] *= corr;
      const float* vp = Vs + (kh * 32 + 4 * (lane >> 5)) * PV + cb * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r)
        oacc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[((r & 3) + 8 * (r >> 2)) * PV], sacc[r], oacc[cb], 0, 0, 0);
    }
  }
  // merge the two key halves of each query block
  if (lane < 32) {
    SM[wave * 32 + lane] = m;
    SM[4 * 32 + wave * 32 + lane] = l;
  }
  __syncthreads();
  {
    const float m_o = SM[(wave ^ 1) * 32 + (lane & 31)], l_o = SM[4 * 32 + (wave ^ 1) * 32 + (lane & 31)];
    const float mg = fmaxf(m, m_o);
    const float fw = (m == -INFINITY) ? 0.f : expf(m - mg), fo = (m_o == -INFINITY) ? 0.f : expf(m_o - mg);
    const float fsc = fw / (l * fw + l_o * fo);
    if constexpr (TRAIN) {
      if (kh == 0 && lane < 32 && qcol < L) ta.lse[((long long)b * gridDim.y + h) * L + qcol] = mg + logf(l * fw + l_o * fo);
    }
    float* od = OT + (wave * 32 + (lane & 31)) * PV + 4 * (lane >> 5);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(f32x4*)(od + cb * 32 + 8 * g) = f32x4{oacc[cb][4 * g] * fsc, oacc[cb][4 * g + 1] * fsc, oacc[cb][4 * g + 2] * fsc,
                                               oacc[cb][4 * g + 3] * fsc};
  }
  __syncthreads();
  for (int idx = t; idx < 64 * (HD / 4); idx += 256) {
    const int r = idx / (HD / 4), c4 = idx - r * (HD / 4);
    const int qi = q0 + r;
    if (qi < L) {
      const int w0 = (r >> 5) * 2, qq = r & 31;
      const f32x4 o = *(const f32x4*)(OT + (w0 * 32 + qq) * PV + 4 * c4) + *(const f32x4*)(OT + ((w0 + 1) * 32 + qq) * PV + 4 * c4);
      *(f32x4*)(out + (long long)b * o_bs + (long long)qi * ldo + h * HD + 4 * c4) = o;
    }
  }
}

// The same flash kernel on split-bf16 MFMAs (library precision modes 1 / 2; head_dim a multiple of 16): 21 MFMAs of 32 cycles per wave and key tile
// instead of 56 exact-f32 MFMAs of 64.  K_j and V_j live in LDS as bf16 hi / lo planes [64 keys][HD + 8] (bf16_planes.h): the A operand of
// S^T = K Q^T is one 16-byte read per plane, the A operand of O^T = V^T P^T comes out of the V planes through transposing reads in the ORDER THE
// ACCUMULATORS HOLD THE KEYS (register 8 s + e of lane half h = key 16 s + 4 h + e, 8 s + 4 + e = key 16 s + 8 + 4 h + e), so the probabilities go from the
// S^T accumulators into the PV product's B operand without leaving the registers; Q's fragments are loaded once.  exp in base 2 (log2(e) folded into Q).
// The next key tile's rows are on their way (registers) while the current one is used.
#ifndef SLATE_BF3_WPS
#define SLATE_BF3_WPS 3
#endif
template <int HD, bool TRAIN>
__global__ __launch_bounds__(256, SLATE_BF3_WPS) void slate_flash_bf3_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                              float* __restrict__ out, int ldq, int ldk, int ldv, int ldo, long long q_bs,
                                                              long long k_bs, long long v_bs, long long o_bs, int L, float scale, SlateTrainArgs ta) {
  constexpr int CB = (HD + 31) / 32, NS = HD / 16, PB = (HD + 8) * 2;   // channel blocks, k16 steps over the channels, row pitch of the planes (bytes)
  constexpr int PT = 64 * PB;                                           // one plane
  constexpr int PV = CB * 32 + 4;                                       // f32 pitch of the merge buffer
  constexpr int KH = 0, KL = PT, VH = 2 * PT, VL = 3 * PT;              // (+ 64 bytes of slack behind VL: the transposing reads of the last channel block)
  constexpr int NV = HD / 16;                                           // float4 per thread of a [64][HD] tile
  extern __shared__ __attribute__((aligned(16))) char bsm[];
  float* SM = (float*)bsm;                                              // [4][32] running max, [4][32] running sum: in the planes' place, after the loop
  float* OT = SM + 2 * 4 * 32;                                          // [4 waves][32 queries][PV]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int q0 = qt * 64;
  const float* qb_ = q + (long long)b * q_bs + h * HD;
  const float* kb_ = k + (long long)b * k_bs + h * HD;
  const float* vb_ = v + (long long)b * v_bs + h * HD;
  const int qblk = wave >> 1, kh = wave & 1, half = lane >> 5;
  const int qcol = q0 + qblk * 32 + (lane & 31);            // this lane's query
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  // Q fragments (B operand of S^T: lane (query, kk) holds channels 16 s + 8 kk .. + 7), scaled into the exponent's base
  PlFrag qf[NS];
  {
    const float* qr = qb_ + (long long)min(qcol, L - 1) * ldq + 8 * half;
    const float sc = scale * LOG2E;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const f32x4 a = *(const f32x4*)(qr + 16 * s) * sc, c = *(const f32x4*)(qr + 16 * s + 4) * sc;
      unsigned h0, l0, h1, l1, h2, l2, h3, l3;
      pl_split2(a[0], a[1], h0, l0);
      pl_split2(a[2], a[3], h1, l1);
      pl_split2(c[0], c[1], h2, l2);
      pl_split2(c[2], c[3], h3, l3);
      typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
      qf[s].h = __builtin_bit_cast(pl_bf16x8, (u32x4_{h0, h1, h2, h3}));
      qf[s].l = __builtin_bit_cast(pl_bf16x8, (u32x4_{l0, l1, l2, l3}));
    }
  }
  // columns HD .. HD + 7 of the planes (+ the slack) feed only output channels >= HD: finite values
  for (int i = t; i < 4 * 64; i += 256) *(uint4*)(bsm + (i >> 6) * PT + (i & 63) * PB + HD * 2) = uint4{0u, 0u, 0u, 0u};
  if (t < 16) *(uint4*)(bsm + 4 * PT + 16 * t) = uint4{0u, 0u, 0u, 0u};
  f32x16 oacc[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[cb][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const int ntiles = qt + 1;                                // causal: key tiles 0 .. qt
  f32x4 tk[NV], tv[NV];
  auto fetch = [&](int kt) {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int i = t + 256 * n, r = i / (HD / 4), c = (i - r * (HD / 4)) * 4;
      const int j = min(kt * 64 + r, L - 1);
      tk[n] = *(const f32x4*)(kb_ + (long long)j * ldk + c);
      tv[n] = *(const f32x4*)(vb_ + (long long)j * ldv + c);
    }
  };
  auto put = [&](const f32x4 (&tt)[NV], int hoff, int loff) {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int i = t + 256 * n, r = i / (HD / 4), c = (i - r * (HD / 4)) * 4;
      unsigned h0, l0, h1, l1;
      pl_split2(tt[n][0], tt[n][1], h0, l0);
      pl_split2(tt[n][2], tt[n][3], h1, l1);
      typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
      *(u32x2_*)(bsm + hoff + r * PB + c * 2) = u32x2_{h0, h1};
      *(u32x2_*)(bsm + loff + r * PB + c * 2) = u32x2_{l0, l1};
    }
  };
  fetch(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();                                        // previous tile fully consumed
    const int k0 = kt * 64;
    put(tk, KH, KL);
    put(tv, VH, VL);
    __syncthreads();
    if (kt + 1 < ntiles) fetch(kt + 1);
    // S^T block: keys k0 + 32 kh + .., queries of block qblk
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const PlFrag ka = pl_rd(bsm, KH, KL, PB, kh * 32, 16 * s, lane);
      pl_mma(sacc, ka, qf[s]);
    }
    float mx = -INFINITY;
    const bool full = kt < qt && k0 + 64 <= L;              // no key of this tile is masked for any query of the workgroup
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (!full) {
        const int key = k0 + kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        sacc[r] = (key <= qcol && key < L) ? sacc[r] : -INFINITY;
      }
      mx = fmaxf(mx, sacc[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float corr = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - mn);
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sacc[r] = (mn == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(sacc[r] - mn);
      sum += sacc[r];
    }
    sum += __shfl_xor(sum, 32, 64);
    l = l * corr + sum;
    m = mn;
    if constexpr (TRAIN) {
      if (ta.drop_thresh) {
        const unsigned base = (unsigned)((((long long)b * gridDim.y + h) * L + min(qcol, L - 1)) * L);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          sacc[r] = (sf_mix32((base + (unsigned)key) ^ ta.drop_seed) >> 8) >= ta.drop_thresh ? sacc[r] * ta.drop_scale : 0.f;
        }
      }
    }
    // probabilities -> the PV product's B operand (registers 8 s .. 8 s + 7 = the keys of virtual k-step s)
    PlFrag pf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      unsigned h0, l0, h1, l1, h2, l2, h3, l3;
      pl_split2(sacc[8 * s], sacc[8 * s + 1], h0, l0);
      pl_split2(sacc[8 * s + 2], sacc[8 * s + 3], h1, l1);
      pl_split2(sacc[8 * s + 4], sacc[8 * s + 5], h2, l2);
      pl_split2(sacc[8 * s + 6], sacc[8 * s + 7], h3, l3);
      typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
      pf[s].h = __builtin_bit_cast(pl_bf16x8, (u32x4_{h0, h1, h2, h3}));
      pf[s].l = __builtin_bit_cast(pl_bf16x8, (u32x4_{l0, l1, l2, l3}));
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[cb][r] *= corr;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const PlFrag va = pl_rd_tr(bsm, VH, VL, PB, kh * 32 + 16 * s, cb * 32, lane, 4, 8);   // lane (channel, kk): keys 16 s + 4 kk + e, 16 s + 8 + 4 kk + e
        pl_mma(oacc[cb], va, pf[s]);
      }
    }
  }
  // merge the two key halves of each query block
  __syncthreads();                                          // (the merge buffers take the planes' place)
  if (lane < 32) {
    SM[wave * 32 + lane] = m;
    SM[4 * 32 + wave * 32 + lane] = l;
  }
  __syncthreads();
  {
    const float m_o = SM[(wave ^ 1) * 32 + (lane & 31)], l_o = SM[4 * 32 + (wave ^ 1) * 32 + (lane & 31)];
    const float mg = fmaxf(m, m_o);
    const float fw = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - mg), fo = (m_o == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_o - mg);
    const float fsc = fw / (l * fw + l_o * fo);
    if constexpr (TRAIN) {
      if (kh == 0 && lane < 32 && qcol < L) ta.lse[((long long)b * gridDim.y + h) * L + qcol] = (mg + log2f(l * fw + l_o * fo)) * LN2;
    }
    float* od = OT + (wave * 32 + (lane & 31)) * PV + 4 * half;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(f32x4*)(od + cb * 32 + 8 * g) = f32x4{oacc[cb][4 * g] * fsc, oacc[cb][4 * g + 1] * fsc, oacc[cb][4 * g + 2] * fsc,
                                               oacc[cb][4 * g + 3] * fsc};
  }
  __syncthreads();
  for (int idx = t; idx < 64 * (HD / 4); idx += 256) {
    const int r = idx / (HD / 4), c4 = idx - r * (HD / 4);
    const int qi = q0 + r;
    if (qi < L) {
      const int w0 = (r >> 5) * 2, qq = r & 31;
      const f32x4 o = *(const f32x4*)(OT + (w0 * 32 + qq) * PV + 4 * c4) + *(const f32x4*)(OT + ((w0 + 1) * 32 + qq) * PV + 4 * c4);
      *(f32x4*)(out + (long long)b * o_bs + (long long)qi * ldo + h * HD + 4 * c4) = o;
    }
  }
}
// (probes: SF_DBG=flash1 keeps the exact-f32 flash kernel in the split-bf16 modes)
static bool slate_flash_f32_forced() {
  static const bool f = getenv("SF_DBG") && strstr(getenv("SF_DBG"), "flash1");
  return f;
}
template <int HD>
constexpr size_t slate_flash_bf3_lds() {
  const size_t planes = (size_t)4 * 64 * (HD + 8) * 2 + 256, merge = ((size_t)2 * 4 * 32 + (size_t)4 * 32 * (((HD + 31) / 32) * 32 + 4)) * sizeof(float);
  return planes > merge ? planes : merge;
}

// Single-query attention (K/V-cached decoding, Lq == 1): one workgroup per (head, batch), the KEYS are spread over the
// 256 threads (online softmax per thread, then one block-wide merge), instead of one thread walking all keys.
template <int HD>
__global__ __launch_bounds__(256) void slate_decode_attn_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                const float* __restrict__ v, float* __restrict__ out, int ldk,
                                                                int ldv, long long q_bs, long long k_bs, long long v_bs,
                                                                long long o_bs, int Lk, float scale) {
  const int t = threadIdx.x, h = blockIdx.x, b = blockIdx.y, lane = t & 63, wave = t >> 6;
  const float* qr = q + (long long)b * q_bs + h * HD;
  float qv[HD], o[HD];
#pragma unroll
  for (int c = 0; c < HD; c += 4) {
    const f32x4 x = *(const f32x4*)(qr + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qv[c + e] = x[e] * scale;
      o[c + e] = 0.f;
    }
  }
  float m = -INFINITY, l = 0.f;
  const float* kb = k + (long long)b * k_bs + h * HD;
  const float* vb = v + (long long)b * v_bs + h * HD;
  for (int j = t; j < Lk; j += 256) {
    const float* kr = kb + (long long)j * ldk;
    const float* vr = vb + (long long)j * ldv;
    float sc = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const f32x4 x = *(const f32x4*)(kr + c);
      sc += (qv[c] * x[0] + qv[c + 1] * x[1]) + (qv[c + 2] * x[2] + qv[c + 3] * x[3]);
    }
    const float mn = fmaxf(m, sc);
    const float corr = (m == -INFINITY) ? 0.f : expf(m - mn), pj = expf(sc - mn);
    l = l * corr + pj;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const f32x4 x = *(const f32x4*)(vr + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[c + e] = o[c + e] * corr + pj * x[e];
    }
    m = mn;
  }
  // block-wide merge: global max, rescale, sum
  __shared__ float s_m[4], s_l[4], s_o[4][HD];
  const float wm = sf_wave_max(m);
  if (lane == 0) s_m[wave] = wm;
  __syncthreads();
  const float gm = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
  const float f = (m == -INFINITY) ? 0.f : expf(m - gm);
  const float wl = sf_wave_sum(l * f);
  if (lane == 0) s_l[wave] = wl;
#pragma unroll
  for (int c = 0; c < HD; ++c) {
    const float wo = sf_wave_sum(o[c] * f);
    if (lane == 0) s_o[wave][c] = wo;
  }
  __syncthreads();
  if (t < HD) {
    const float tl = (s_l[0] + s_l[1]) + (s_l[2] + s_l[3]);
    out[(long long)b * o_bs + h * HD + t] = ((s_o[0][t] + s_o[1][t]) + (s_o[2][t] + s_o[3][t])) / tl;
  }
}

// x[b, t] = tok_emb[idx[b, t]] + pos[t]   (steve_transformer.py:291-296: BOS already prepended by the caller)
// idx element of (batch b, position t) = idx[b * idx_bs + t]; const_tok >= 0 replaces the lookup (the BOS token)
__global__ void embed_kernel(const long long* __restrict__ idx, long long idx_bs, long long const_tok,
                             const float* __restrict__ emb, const float* __restrict__ pos, float* __restrict__ out, int L,
                             int d, long long total4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int d4 = d / 4;
  const long long row = i / d4;
  const int c4 = (int)(i - row * d4), tpos = (int)(row % L);
  const long long tok = const_tok >= 0 ? const_tok : idx[(row / L) * idx_bs + tpos];
  const f32x4 e = *(const f32x4*)(emb + tok * d + 4 * c4), p = *(const f32x4*)(pos + (long long)tpos * d + 4 * c4);
  *(f32x4*)(out + row * d + 4 * c4) = e + p;
}

// first index of the row maximum (torch.argmax / topk(k=1) tie rule); one workgroup of 256 threads per row
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, long long ld, long long* __restrict__ out,
                                                          long long out_stride, int V) {
  const long long r = blockIdx.x;
  const float* xr = x + r * ld;
  const int t = threadIdx.x;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int j = t; j < V; j += 256) {
    const float v = xr[j];
    if (v > best) {
      best = v;
      bi = j;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) {
      best = ob;
      bi = oi;
    }
  }
  __shared__ float sb[4];
  __shared__ int si[4];
  if ((t & 63) == 0) {
    sb[t >> 6] = best;
    si[t >> 6] = bi;
  }
  __syncthreads();
  if (t == 0) {
    for (int w = 1; w < 4; ++w)
      if (sb[w] > best || (sb[w] == best && si[w] < bi)) {
        best = sb[w];
        bi = si[w];
      }
    out[r * out_stride] = bi;
  }
}

// per-row cross-entropy  -log softmax(x)[target]  (F.cross_entropy, steve.py:341-344); one workgroup per row
__global__ __launch_bounds__(256) void xent_rows_kernel(const float* __restrict__ x, const long long* __restrict__ tgt,
                                                        float* __restrict__ loss, int V) {
  const long long r = blockIdx.x;
  const float* xr = x + r * (long long)V;
  const int t = threadIdx.x;
  __shared__ float sh[4];
  float mx = -INFINITY;
  for (int j = t; j < V; j += 256) mx = fmaxf(mx, xr[j]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((t & 63) == 0) sh[t >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  float se = 0.f;
  for (int j = t; j < V; j += 256) se += expf(xr[j] - mx);
  se = sf_sum64(se);
  if ((t & 63) == 0) sh[t >> 6] = se;
  __syncthreads();
  if (t == 0) loss[r] = logf((sh[0] + sh[1]) + (sh[2] + sh[3])) + mx - xr[tgt[r]];
}

// y[r] = softmax((x[r] + add[r]) * scale) over V columns; add may be null.  One workgroup per row.
// Gumbel(0, 1) sample of element idx as a pure function of (seed, idx): -log(E) with E = -log(u) ~ Exp(1), u uniform on the 2^23
// midpoints of (0, 1) -- the noise of steve_utils.py:30-35 without a trip through memory (tests rebuild it on the host).
__device__ __forceinline__ float sf_gumbel(uint32_t idx, uint32_t sseed) {
  const float u = ((float)(sf_mix32(idx ^ sseed) >> 9) + 0.5f) * 1.1920929e-7f;
  return -logf(-logf(u));
}
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                           float scale, float* __restrict__ y, int V, int noise, uint32_t sseed,
                                                           int logout) {
  const long long r = blockIdx.x;
  const float* xr = x + r * (long long)V;
  const float* ar = add ? add + r * (long long)V : nullptr;
  float* yr = y + r * (long long)V;
  const int t = threadIdx.x;
  __shared__ float sh[4];
  float mx = -INFINITY;
  if (noise) {   // y holds x + g between the passes
    for (int j = t; j < V; j += 256) {
      const float v = (xr[j] + sf_gumbel((uint32_t)(r * V + j), sseed)) * scale;
      yr[j] = v;
      mx = fmaxf(mx, v);
    }
    xr = yr;
    scale = 1.f;
  } else {
    for (int j = t; j < V; j += 256) mx = fmaxf(mx, (xr[j] + (ar ? ar[j] : 0.f)) * scale);
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((t & 63) == 0) sh[t >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  float se = 0.f;
  for (int j = t; j < V; j += 256) {
    const float c = (xr[j] + (ar ? ar[j] : 0.f)) * scale - mx;
    const float e = expf(c);
    yr[j] = logout ? c : e;
    se += e;
  }
  se = sf_sum64(se);
  if ((t & 63) == 0) sh[t >> 6] = se;
  __syncthreads();
  const float tot = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  const float inv = 1.0f / tot, lg = logf(tot);
  for (int j = t; j < V; j += 256) yr[j] = logout ? yr[j] - lg : yr[j] * inv;
}

// the same for rows of at most 1024 * NV floats (V % 4 == 0, 16-byte aligned rows): the row lives in registers between the
// passes, so x (and add) are read once and y is written once -- 3 row passes of HBM traffic instead of 7, 2 with hashed noise
// d/dx of mean_r(-log softmax(x[r])[target[r]]) times an upstream scalar: dx = (softmax(x) - onehot(target)) * g[0] * inv_rows
// (F.cross_entropy's backward, steve.py:341-344), one read of x and one write of dx
template <int NV>
__global__ __launch_bounds__(256) void xent_bwd_rows_kernel(const float* __restrict__ x, const long long* __restrict__ target,
                                                            const float* __restrict__ g, float inv_rows, float* __restrict__ dx, int V) {
  const long long r = blockIdx.x;
  const int t = threadIdx.x;
  __shared__ float sh[2][4];
  const int tg = (int)target[r];
  f32x4 v[NV];
  float mx = -INFINITY;
  for (int n = 0; n < NV; ++n) {
    for (int k = 0; k < 4; ++k) {
      const int j = (t + 256 * n) * 4 + k;
      v[n][k] = j < V ? x[r * V + j] : -INFINITY;
      mx = fmaxf(mx, v[n][k]);
    }
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((t & 63) == 0) sh[0][t >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sh[0][0], sh[0][1]), fmaxf(sh[0][2], sh[0][3]));
  float se = 0.f;
#pragma unroll
  for (int n = 0; n < NV; ++n) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[n][k] = expf(v[n][k] - mx);
      se += v[n][k];
    }
  }
  se = sf_sum64(se);
  if ((t & 63) == 0) sh[1][t >> 6] = se;
  __syncthreads();
  const float inv = 1.0f / ((sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]));
  const float gs = g[0] * inv_rows;
  for (int n = 0; n < NV; ++n) {
    for (int k = 0; k < 4; ++k) {
      const int j = (t + 256 * n) * 4 + k;
      if (j < V) dx[r * V + j] = (v[n][k] * inv - (j == tg ? 1.f : 0.f)) * gs;
    }
  }
}

template <int NV>
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                               float scale, float* __restrict__ y, int V, int noise, uint32_t sseed,
                                                               int logout) {
  const long long r = blockIdx.x;
  const int t = threadIdx.x;
  __shared__ float sh[2][4];
  f32x4 v[NV];
  float mx = -INFINITY;
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    const int j = (t + 256 * n) * 4;
    if (j < V) {
      f32x4 a = *(const f32x4*)(x + r * V + j);
      if (add) a += *(const f32x4*)(add + r * V + j);
      if (noise) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += sf_gumbel((uint32_t)(r * V + j + k), sseed);
      }
      v[n] = a * scale;
      mx = fmaxf(fmaxf(mx, fmaxf(v[n][0], v[n][1])), fmaxf(v[n][2], v[n][3]));
    } else {
      v[n] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((t & 63) == 0) sh[0][t >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sh[0][0], sh[0][1]), fmaxf(sh[0][2], sh[0][3]));
  float se = 0.f;
#pragma unroll
  for (int n = 0; n < NV; ++n) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float c = v[n][k] - mx, e = expf(c);
      v[n][k] = logout ? c : e;
      se += e;
    }
  }
  se = sf_sum64(se);
  if ((t & 63) == 0) sh[1][t >> 6] = se;
  __syncthreads();
  const float tot = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
  const float inv = 1.0f / tot, lg = logf(tot);
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    const int j = (t + 256 * n) * 4;
    if (j < V) *(f32x4*)(y + r * V + j) = logout ? v[n] - lg : v[n] * inv;
  }
}
template <int NV>
__global__ __launch_bounds__(256) void softmax_bwd_rows_reg_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                                   float scale, float* __restrict__ dx, int V) {
  const long long r = blockIdx.x;
  const int t = threadIdx.x;
  __shared__ float sh[4];
  f32x4 yv[NV], gv[NV];
  float d = 0.f;
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    const int j = (t + 256 * n) * 4;
    if (j < V) {
      yv[n] = *(const f32x4*)(y + r * V + j);
      gv[n] = *(const f32x4*)(dy + r * V + j);
      d += (yv[n][0] * gv[n][0] + yv[n][1] * gv[n][1]) + (yv[n][2] * gv[n][2] + yv[n][3] * gv[n][3]);
    }
  }
  d = sf_sum64(d);
  if ((t & 63) == 0) sh[t >> 6] = d;
  __syncthreads();
  const float dot = (sh[0] + sh[1]) + (sh[2] + sh[3]);
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    const int j = (t + 256 * n) * 4;
    if (j < V) *(f32x4*)(dx + r * V + j) = yv[n] * (gv[n] - dot) * scale;
  }
}

// mean of n floats in a fixed order (single workgroup, double accumulation)
__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ x, float* __restrict__ out, long long n) {
  __shared__ double sh[256];
  double a = 0.0;
  for (long long i = threadIdx.x; i < n; i += 256) a += (double)x[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(sh[0] / (double)n);
}
// ---- backward of GroupNorm(1 group) + ReLU (+ PixelShuffle) (row N1: the dVAE under autograd) ----
// prep: g = dy (gathered back through the pixel shuffle) gated by the ReLU of the recomputed output; writes g and g * xhat
// (their column sums are d_beta / d_gamma) and per-block partial sums of gamma*g and gamma*g*xhat (the two per-sample means
// of the GroupNorm adjoint).  apply: dx = rstd * (gamma g - S1/n - xhat S2/n).
__device__ __forceinline__ void gn_frame_stats(const double* part, int f, long long n, float eps, float* stat) {
  const int t = threadIdx.x;
  if (t < 64) {
    double s = part[((long long)f * GN_P + t) * 2], q = part[((long long)f * GN_P + t) * 2 + 1];
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o, 64);
      q += __shfl_xor(q, o, 64);
    }
    if (t == 0) {
      const double mean = s / (double)n;
      const double var = q / (double)n - mean * mean;
      stat[0] = (float)mean;
      stat[1] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
    }
  }
  __syncthreads();
}
__global__ __launch_bounds__(256) void gn_bwd_prep_kernel(const float* __restrict__ x, const double* __restrict__ part,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ dy, float* __restrict__ gb,
                                                          float* __restrict__ gxb, float* __restrict__ sums, int H, int W, int C,
                                                          float eps, int relu, int shuffle) {
  const int f = blockIdx.y, t = threadIdx.x;
  const long long n = (long long)H * W * C;
  __shared__ float stat[2];
  __shared__ float red[2][4];
  gn_frame_stats(part, f, n, eps, stat);
  const float mean = stat[0], rstd = stat[1];
  const long long i4 = (long long)blockIdx.x * 256 + t;
  float s1 = 0.f, s2 = 0.f;
  if (i4 * 4 < n) {
    const long long e = i4 * 4;
    const int c = (int)(e % C);
    const f32x4 v = *(const f32x4*)(x + (long long)f * n + e);
    const f32x4 g4 = *(const f32x4*)(gamma + c), b4 = *(const f32x4*)(beta + c);
    const f32x4 xh = (v - mean) * rstd;
    const f32x4 o = xh * g4 + b4;
    f32x4 d;
    if (shuffle == 1) {
      d = *(const f32x4*)(dy + (long long)f * n + e);
    } else {
      const long long pix = e / C;
      const int yy = (int)(pix / W), xx = (int)(pix - (long long)yy * W);
      const int Co = C / 4, co = c / 4;
      const float* yo = dy + (long long)f * n;
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] = yo[((long long)(2 * yy + (k >> 1)) * (2 * W) + 2 * xx + (k & 1)) * Co + co];
    }
    f32x4 g;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      g[k] = (relu && o[k] <= 0.f) ? 0.f : d[k];
      s1 += g4[k] * g[k];
      s2 += g4[k] * g[k] * xh[k];
    }
    *(f32x4*)(gb + (long long)f * n + e) = g;
    *(f32x4*)(gxb + (long long)f * n + e) = g * xh;
  }
  s1 = sf_sum64(s1);
  s2 = sf_sum64(s2);
  if ((t & 63) == 0) {
    red[0][t >> 6] = s1;
    red[1][t >> 6] = s2;
  }
  __syncthreads();
  if (t == 0) {
    sums[((long long)f * gridDim.x + blockIdx.x) * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    sums[((long long)f * gridDim.x + blockIdx.x) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}
// one block per frame: the frame's two sums from the per-block partials, in a fixed order, as means over the n elements
__global__ __launch_bounds__(256) void gn_bwd_sums_kernel(const float* __restrict__ sums, float* __restrict__ means, int nb,
                                                          long long n) {
  const int f = blockIdx.x, t = threadIdx.x;
  __shared__ double sh[2][256];
  double a = 0.0, b = 0.0;
  for (int i = t; i < nb; i += 256) {
    a += (double)sums[((long long)f * nb + i) * 2];
    b += (double)sums[((long long)f * nb + i) * 2 + 1];
  }
  sh[0][t] = a;
  sh[1][t] = b;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (t < w) {
      sh[0][t] += sh[0][t + w];
      sh[1][t] += sh[1][t + w];
    }
    __syncthreads();
  }
  if (t == 0) {
    means[f * 2] = (float)(sh[0][0] / (double)n);
    means[f * 2 + 1] = (float)(sh[1][0] / (double)n);
  }
}
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, const double* __restrict__ part,
                                                           const float* __restrict__ gamma, const float* __restrict__ gb,
                                                           const float* __restrict__ means, float* __restrict__ dx, int H, int W,
                                                           int C, float eps) {
  const int f = blockIdx.y, t = threadIdx.x;
  const long long n = (long long)H * W * C;
  __shared__ float stat[2];
  gn_frame_stats(part, f, n, eps, stat);
  const float mean = stat[0], rstd = stat[1], m1 = means[f * 2], m2 = means[f * 2 + 1];
  const long long i4 = (long long)blockIdx.x * 256 + t;
  if (i4 * 4 >= n) return;
  const long long e = i4 * 4;
  const int c = (int)(e % C);
  const f32x4 v = *(const f32x4*)(x + (long long)f * n + e);
  const f32x4 g4 = *(const f32x4*)(gamma + c);
  const f32x4 g = *(const f32x4*)(gb + (long long)f * n + e);
  const f32x4 xh = (v - mean) * rstd;
  *(f32x4*)(dx + (long long)f * n + e) = (g4 * g - m1 - xh * m2) * rstd;
}

// softmax backward over rows: dx = scale * y * (dy - sum(y * dy))   (y = softmax((x + add) * scale))
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ y, const float* __restrict__ dy, float scale,
                                                               float* __restrict__ dx, int V) {
  const long long r = blockIdx.x;
  const float* yr = y + r * (long long)V;
  const float* gr = dy + r * (long long)V;
  const int t = threadIdx.x;
  __shared__ float sh[4];
  float d = 0.f;
  for (int j = t; j < V; j += 256) d += yr[j] * gr[j];
  d = sf_sum64(d);
  if ((t & 63) == 0) sh[t >> 6] = d;
  __syncthreads();
  const float dot = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  for (int j = t; j < V; j += 256) dx[r * (long long)V + j] = scale * yr[j] * (gr[j] - dot);
}
}  // namespace

extern "C" {

// out[b, i, h*hd + c] = sum_j softmax_j(q[b,i,h] . k[b,j,h] * hd^-0.5 (j <= i if causal)) v[b,j,h,c]
// q rows [Lq] of batch b start at q + b*q_bs (leading dim ldq), k/v rows [Lk] at k + b*k_bs / v + b*v_bs, out at
// out + b*o_bs; head_dim in {16, 32, 48, 64}.  Explicit batch strides let k/v be a partially filled K/V cache.
int sf_slate_attention_strided_f32(const float* q, const float* k, const float* v, float* out, int ldq, int ldk, int ldv,
                                   int ldo, long long q_bs, long long k_bs, long long v_bs, long long o_bs, int B, int Lq,
                                   int Lk, int num_heads, int head_dim, int causal, void* stream) {
  SF_REQUIRE(q && k && v && out, "sf_slate_attention_f32: null pointer");
  SF_REQUIRE(B >= 0 && Lq > 0 && Lk > 0 && num_heads > 0, "sf_slate_attention_f32: bad shape");
  SF_REQUIRE((ldq % 4) == 0 && (ldk % 4) == 0 && (ldv % 4) == 0 && (ldo % 4) == 0 && (q_bs % 4) == 0 && (k_bs % 4) == 0 &&
                 (v_bs % 4) == 0 && (o_bs % 4) == 0, "sf_slate_attention_f32: leading dims / strides must be multiples of 4");
  SF_REQUIRE(!causal || Lq == Lk, "sf_slate_attention_f32: causal attention needs Lq == Lk");
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((Lq + 255) / 256, num_heads, B);
  const float scale = 1.0f / sqrtf((float)head_dim);
  if (causal && Lq >= 128) {   // long causal self-attention: MFMA flash kernel
#define SLATE_FLASH(HD_)                                                                                                     \
  if (head_dim == HD_ && sf_get_precision() >= 1 && !slate_flash_f32_forced()) {   /* split-bf16 modes: the bf16-plane kernel */                           \
    SF_TRY(sf_ensure_dyn_lds((const void*)slate_flash_bf3_kernel<HD_, false>, slate_flash_bf3_lds<HD_>()));                    \
    hipLaunchKernelGGL((slate_flash_bf3_kernel<HD_, false>), dim3((Lq + 63) / 64, num_heads, B), dim3(256), slate_flash_bf3_lds<HD_>(), st, q, k, v, \
                       out, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, Lq, scale, SlateTrainArgs{nullptr, 0u, 0u, 1.f});     \
    SF_CHECK_LAUNCH();                                                                                                       \
    return 0;                                                                                                                \
  }                                                                                                                          \
  if (head_dim == HD_) {                                                                                                     \
    constexpr int CB_ = (HD_ + 31) / 32;                                                                                     \
    constexpr size_t lds_ = ((size_t)2 * 64 * (HD_ + 4) + (size_t)64 * (CB_ * 32 + 4) + 2 * 4 * 32 +                          \
                             (size_t)4 * 32 * (CB_ * 32 + 4)) * sizeof(float);                                               \
    SF_TRY(sf_ensure_dyn_lds((const void*)slate_flash_kernel<HD_, false>, lds_));                                                       \
    hipLaunchKernelGGL((slate_flash_kernel<HD_, false>), dim3((Lq + 63) / 64, num_heads, B), dim3(256), lds_, st, q, k, v, out, ldq,   \
                       ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, Lq, scale, SlateTrainArgs{nullptr, 0u, 0u, 1.f});                   \
    SF_CHECK_LAUNCH();                                                                                                       \
    return 0;                                                                                                                \
  }
    SLATE_FLASH(16)
    SLATE_FLASH(32)
    SLATE_FLASH(48)
    SLATE_FLASH(64)
#undef SLATE_FLASH
  }
  if (Lq == 1 && !causal) {   // K/V-cached decoding: spread the keys over a workgroup
#define SLATE_DEC(HD_)                                                                                                       \
  if (head_dim == HD_) {                                                                                                     \
    hipLaunchKernelGGL(slate_decode_attn_kernel<HD_>, dim3(num_heads, B), dim3(256), 0, st, q, k, v, out, ldk, ldv, q_bs, k_bs, \
                       v_bs, o_bs, Lk, scale);                                                                               \
    SF_CHECK_LAUNCH();                                                                                                       \
    return 0;                                                                                                                \
  }
    SLATE_DEC(16)
    SLATE_DEC(32)
    SLATE_DEC(48)
    SLATE_DEC(64)
#undef SLATE_DEC
  }
#define SLATE_CASE(HD_)                                                                                                  \
  if (head_dim == HD_) {                                                                                                 \
    hipLaunchKernelGGL(slate_attn_kernel<HD_>, grid, dim3(256), 0, st, q, k, v, out, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, \
                       o_bs, Lq, Lk, causal, scale);                                                                     \
    SF_CHECK_LAUNCH();                                                                                                   \
    return 0;                                                                                                            \
  }
  SLATE_CASE(16)
  SLATE_CASE(32)
  SLATE_CASE(48)
  SLATE_CASE(64)
#undef SLATE_CASE
  return sf_set_err(-1, "invalid argument: sf_slate_attention_f32 head_dim must be 16, 32, 48 or 64", __FILE__, __LINE__);
}

}  // extern "C"

// causal self-attention training forward on the flash kernel; returns 1 when it does not apply (caller uses the generic kernel)
int sf_slate_flash_train_ex(const float* q, const float* k, const float* v, float* out, float* lse, int ldq, int ldk, int ldv, int ldo,
                            long long q_bs, long long k_bs, long long v_bs, long long o_bs, int B, int L, int num_heads, int head_dim,
                            unsigned drop_seed, unsigned drop_thresh, float drop_scale, hipStream_t st) {
  if (L < 128 || !(head_dim == 16 || head_dim == 32 || head_dim == 48 || head_dim == 64)) return 1;
  const float scale = 1.0f / sqrtf((float)head_dim);
  const SlateTrainArgs ta{lse, drop_seed, drop_thresh, drop_scale};
#define SLATE_FLASH_T(HD_)                                                                                                     \
  if (head_dim == HD_ && sf_get_precision() >= 1 && !slate_flash_f32_forced()) {                                               \
    SF_TRY(sf_ensure_dyn_lds((const void*)slate_flash_bf3_kernel<HD_, true>, slate_flash_bf3_lds<HD_>()));                       \
    hipLaunchKernelGGL((slate_flash_bf3_kernel<HD_, true>), dim3((L + 63) / 64, num_heads, B), dim3(256), slate_flash_bf3_lds<HD_>(), st, q, k, v, \
                       out, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, L, scale, ta);                                          \
    SF_CHECK_LAUNCH();                                                                                                         \
    return 0;                                                                                                                  \
  }                                                                                                                            \
  if (head_dim == HD_) {                                                                                                       \
    constexpr int CB_ = (HD_ + 31) / 32;                                                                                       \
    constexpr size_t lds_ = ((size_t)2 * 64 * (HD_ + 4) + (size_t)64 * (CB_ * 32 + 4) + 2 * 4 * 32 +                            \
                             (size_t)4 * 32 * (CB_ * 32 + 4)) * sizeof(float);                                                 \
    SF_TRY(sf_ensure_dyn_lds((const void*)slate_flash_kernel<HD_, true>, lds_));                                                       \
    hipLaunchKernelGGL((slate_flash_kernel<HD_, true>), dim3((L + 63) / 64, num_heads, B), dim3(256), lds_, st, q, k, v, out, ldq, \
                       ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, L, scale, ta);                                                   \
    SF_CHECK_LAUNCH();                                                                                                         \
    return 0;                                                                                                                  \
  }
  SLATE_FLASH_T(16)
  SLATE_FLASH_T(32)
  SLATE_FLASH_T(48)
  SLATE_FLASH_T(64)
#undef SLATE_FLASH_T
  return 1;
}

extern "C" {
// contiguous batches: q [B*Lq, ldq], k [B*Lk, ldk], v [B*Lk, ldv], out [B*Lq, ldo]
int sf_slate_attention_f32(const float* q, const float* k, const float* v, float* out, int ldq, int ldk, int ldv, int ldo,
                           int B, int Lq, int Lk, int num_heads, int head_dim, int causal, void* stream) {
  return sf_slate_attention_strided_f32(q, k, v, out, ldq, ldk, ldv, ldo, (long long)Lq * ldq, (long long)Lk * ldk,
                                        (long long)Lk * ldv, (long long)Lq * ldo, B, Lq, Lk, num_heads, head_dim, causal, stream);
}

// out [R = B*L, d] = tok_emb[idx] + pos[t];  idx int64 [B, L] (values < rows of tok_emb), pos [>= L, d]
int sf_embed_tokens_f32(const long long* idx, const float* tok_emb, const float* pos, float* out, int B, int L, int d,
                        void* stream) {
  SF_REQUIRE(idx && tok_emb && pos && out && B >= 0 && L > 0 && d > 0 && (d % 4) == 0, "sf_embed_tokens_f32: bad arguments");
  const long long total4 = (long long)B * L * (d / 4);
  if (total4 == 0) return 0;
  hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx,
                     (long long)L, -1LL, tok_emb, pos, out, L, d, total4);
  SF_CHECK_LAUNCH();
  return 0;
}

// out[r] = first index of max(x[r, 0:V]);  x rows ld floats apart
int sf_argmax_rows_f32(const float* x, long long ld, long long* out, long long R, int V, void* stream) {
  SF_REQUIRE(x && out && R >= 0 && V > 0 && ld >= V, "sf_argmax_rows_f32: bad arguments");
  if (R == 0) return 0;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, x, ld, out, 1LL, V);
  SF_CHECK_LAUNCH();
  return 0;
}

// y[r, :] = softmax((x[r, :] + add[r, :]) * scale); add may be NULL.  With add = Gumbel noise and scale = 1/tau this is
// steve_utils.py:26-41 gumbel_softmax(log_softmax(x), tau) (the log-partition shift cancels in the softmax).
static int softmax_rows_launch(const float* x, const float* add, float scale, float* y, long long R, int V, int noise,
                               uint32_t sseed, hipStream_t st, int logout = 0) {
  const bool vec = (V % 4) == 0 && V <= 4096 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)add) & 15) == 0;
  if (vec && V <= 1024)
    hipLaunchKernelGGL(softmax_rows_reg_kernel<1>, dim3((unsigned)R), dim3(256), 0, st, x, add, scale, y, V, noise, sseed, logout);
  else if (vec)
    hipLaunchKernelGGL(softmax_rows_reg_kernel<4>, dim3((unsigned)R), dim3(256), 0, st, x, add, scale, y, V, noise, sseed, logout);
  else
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)R), dim3(256), 0, st, x, add, scale, y, V, noise, sseed, logout);
  SF_CHECK_LAUNCH();
  return 0;
}
int sf_softmax_rows_f32(const float* x, const float* add, float scale, float* y, long long R, int V, void* stream) {
  SF_REQUIRE(x && y && R >= 0 && V > 0, "sf_softmax_rows_f32: bad arguments");
  if (R == 0) return 0;
  return softmax_rows_launch(x, add, scale, y, R, V, 0, 0u, (hipStream_t)stream);
}
// backward of sf_cross_entropy_f32's mean: dx[r, :] = (softmax(x[r, :]) - onehot(target[r])) * g[0] / R, g a DEVICE scalar (the
// upstream gradient of the loss).  V <= 16384.
int sf_cross_entropy_bwd_f32(const float* x, const long long* target, const float* g, float* dx, long long R, int V, void* stream) {
  SF_REQUIRE(x && target && g && dx && R > 0 && V > 0 && V <= 16384, "sf_cross_entropy_bwd_f32: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const float inv_rows = 1.f / (float)R;
  if (V <= 1024) hipLaunchKernelGGL(xent_bwd_rows_kernel<1>, dim3((unsigned)R), dim3(256), 0, st, x, target, g, inv_rows, dx, V);
  else if (V <= 4096) hipLaunchKernelGGL(xent_bwd_rows_kernel<4>, dim3((unsigned)R), dim3(256), 0, st, x, target, g, inv_rows, dx, V);
  else hipLaunchKernelGGL(xent_bwd_rows_kernel<16>, dim3((unsigned)R), dim3(256), 0, st, x, target, g, inv_rows, dx, V);
  SF_CHECK_LAUNCH();
  return 0;
}
// y[r, :] = log_softmax(x[r, :]) (F.log_softmax over the last dim: the z_logits of dVAE.py:127)
int sf_log_softmax_rows_f32(const float* x, float* y, long long R, int V, void* stream) {
  SF_REQUIRE(x && y && R >= 0 && V > 0, "sf_log_softmax_rows_f32: bad arguments");
  if (R == 0) return 0;
  return softmax_rows_launch(x, nullptr, 1.f, y, R, V, 0, 0u, (hipStream_t)stream, 1);
}
// y[r, :] = softmax((x[r, :] + g[r, :]) * scale) with g ~ Gumbel(0, 1) generated in the kernel as a pure function of
// (seed, r * V + j): steve_utils.py:26-41 with tau = 1 / scale, the noise never written to memory.  R * V < 2^32.
int sf_gumbel_softmax_rows_f32(const float* x, unsigned long long seed, float scale, float* y, long long R, int V, void* stream) {
  SF_REQUIRE(x && y && R >= 0 && V > 0, "sf_gumbel_softmax_rows_f32: bad arguments");
  SF_REQUIRE(R * (long long)V < (1LL << 32), "sf_gumbel_softmax_rows_f32: more than 2^32 elements");
  if (R == 0) return 0;
  const uint32_t sseed = sf_mix32((uint32_t)seed ^ sf_mix32((uint32_t)(seed >> 32) + 0x9e3779b9u));
  return softmax_rows_launch(x, nullptr, scale, y, R, V, 1, sseed, (hipStream_t)stream);
}

// loss_rows[r] = -log softmax(x[r])[target[r]];  mean_out[0] = mean over rows (F.cross_entropy default reduction)
int sf_cross_entropy_f32(const float* x, const long long* target, float* loss_rows, float* mean_out, long long R, int V,
                         void* stream) {
  SF_REQUIRE(x && target && loss_rows && mean_out && R > 0 && V > 0, "sf_cross_entropy_f32: bad arguments");
  hipLaunchKernelGGL(xent_rows_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, x, target, loss_rows, V);
  SF_CHECK_LAUNCH();
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_rows, mean_out, R);
  SF_CHECK_LAUNCH();
  return 0;
}


size_t sf_groupnorm1_workspace_bytes(int F) { return (size_t)(F > 0 ? F : 0) * GN_P * 2 * sizeof(double); }

// F.group_norm(x, 1, gamma, beta, eps) (+ ReLU) on NHWC x [F,H,W,C]; pixel_shuffle 1 (none) or 2
// (y [F,2H,2W,C/4], nn.PixelShuffle(2) applied to the normalised map).
int sf_groupnorm1_nhwc_f32(const float* x, const float* gamma, const float* beta, float* y, int F, int H, int W, int C,
                           float eps, int relu, int pixel_shuffle, void* ws, size_t ws_bytes, void* stream) {
  SF_REQUIRE(x && gamma && beta && y && ws, "sf_groupnorm1_nhwc_f32: null pointer");
  SF_REQUIRE(F >= 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0, "sf_groupnorm1_nhwc_f32: bad shape");
  SF_REQUIRE(pixel_shuffle == 1 || (pixel_shuffle == 2 && (C % 16) == 0), "sf_groupnorm1_nhwc_f32: pixel_shuffle must be 1 or 2");
  SF_REQUIRE(ws_bytes >= sf_groupnorm1_workspace_bytes(F), "sf_groupnorm1_nhwc_f32: workspace too small");
  if (F == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)H * W * C;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(GN_P, F), dim3(256), 0, st, x, (double*)ws, n);
  SF_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)((n / 4 + 255) / 256), F), dim3(256), 0, st, x, (const double*)ws, gamma,
                     beta, y, H, W, C, eps, relu, pixel_shuffle);
  SF_CHECK_LAUNCH();
  return 0;
}


// ---- backward of the two calls above (row N1: the dVAE under autograd, dVAE.py:113-139) ----
static size_t gn_bwd_blocks(int H, int W, int C) { return (((size_t)H * W * C) / 4 + 255) / 256; }
static size_t gn_pad(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
size_t sf_groupnorm1_bwd_workspace_bytes(int F, int H, int W, int C) {
  if (F <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
  const size_t n = (size_t)H * W * C, nb = gn_bwd_blocks(H, W, C);
  return gn_pad(sf_groupnorm1_workspace_bytes(F)) + 2 * gn_pad((size_t)F * n * sizeof(float)) + gn_pad((size_t)F * nb * 2 * sizeof(float)) +
         gn_pad((size_t)F * 2 * sizeof(float)) + gn_pad(sf_grad_partial_floats((long long)F * H * W, C, C) * sizeof(float)) + 256;
}
// x: the INPUT of the forward call; dy: gradient of its output ([F,H,W,C], or [F,2H,2W,C/4] with pixel_shuffle 2).
// dx [F,H,W,C]; dgamma / dbeta [C] (overwritten).
int sf_groupnorm1_nhwc_bwd_f32(const float* x, const float* gamma, const float* beta, const float* dy, float* dx, float* dgamma,
                               float* dbeta, int F, int H, int W, int C, float eps, int relu, int pixel_shuffle, void* ws,
                               size_t ws_bytes, void* stream) {
  SF_REQUIRE(x && gamma && beta && dy && dx && dgamma && dbeta && ws, "sf_groupnorm1_nhwc_bwd_f32: null pointer");
  SF_REQUIRE(F > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0, "sf_groupnorm1_nhwc_bwd_f32: bad shape");
  SF_REQUIRE(pixel_shuffle == 1 || (pixel_shuffle == 2 && (C % 16) == 0), "sf_groupnorm1_nhwc_bwd_f32: pixel_shuffle must be 1 or 2");
  SF_REQUIRE(ws_bytes >= sf_groupnorm1_bwd_workspace_bytes(F, H, W, C), "sf_groupnorm1_nhwc_bwd_f32: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)H * W * C;
  const unsigned nb = (unsigned)gn_bwd_blocks(H, W, C);
  char* p = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  double* part = (double*)p;
  p += gn_pad(sf_groupnorm1_workspace_bytes(F));
  float* gb = (float*)p;
  p += gn_pad((size_t)F * n * sizeof(float));
  float* gxb = (float*)p;
  p += gn_pad((size_t)F * n * sizeof(float));
  float* sums = (float*)p;
  p += gn_pad((size_t)F * nb * 2 * sizeof(float));
  float* means = (float*)p;
  p += gn_pad((size_t)F * 2 * sizeof(float));
  float* partial = (float*)p;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(GN_P, F), dim3(256), 0, st, x, part, n);
  SF_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_bwd_prep_kernel, dim3(nb, F), dim3(256), 0, st, x, (const double*)part, gamma, beta, dy, gb, gxb, sums, H, W, C,
                     eps, relu, pixel_shuffle);
  SF_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_bwd_sums_kernel, dim3(F), dim3(256), 0, st, (const float*)sums, means, (int)nb, n);
  SF_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(nb, F), dim3(256), 0, st, x, (const double*)part, gamma, (const float*)gb,
                     (const float*)means, dx, H, W, C, eps);
  SF_CHECK_LAUNCH();
  SF_TRY(sf_grad_bias_ex(gb, dbeta, (long long)F * H * W, C, partial, st));
  SF_TRY(sf_grad_bias_ex(gxb, dgamma, (long long)F * H * W, C, partial, st));
  return 0;
}

// dx[r, :] = scale * y[r, :] * (dy[r, :] - sum(y[r, :] * dy[r, :])): backward of sf_softmax_rows_f32 w.r.t. x given its output y
int sf_softmax_rows_bwd_f32(const float* y, const float* dy, float scale, float* dx, long long R, int V, void* stream) {
  SF_REQUIRE(y && dy && dx && R >= 0 && V > 0, "sf_softmax_rows_bwd_f32: bad arguments");
  if (R == 0) return 0;
  const bool vec = (V % 4) == 0 && V <= 4096 && (((uintptr_t)y | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0;
  if (vec && V <= 1024)
    hipLaunchKernelGGL(softmax_bwd_rows_reg_kernel<1>, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, y, dy, scale, dx, V);
  else if (vec)
    hipLaunchKernelGGL(softmax_bwd_rows_reg_kernel<4>, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, y, dy, scale, dx, V);
  else
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, y, dy, scale, dx, V);
  SF_CHECK_LAUNCH();
  return 0;
}


// ---- K/V-cached greedy generation (STEVETransformerDecoder.generate with sample=False, steve_transformer.py:305-333,
//      computed one token per step; the arithmetic of a step is that of the reference's forward on the prefix) ----
size_t sf_slate_generate_workspace_bytes(const sf_slate_decoder* m, int B, int steps) {
  if (!m || B <= 0 || steps <= 0) return 0;
  const size_t d = m->d_model, N = m->num_slots, L = m->num_layers;
  auto pad = [](size_t nfloat) { return ((nfloat * sizeof(float)) + 255) & ~(size_t)255; };
  return pad(B * N * d) + L * pad(B * N * 2 * d) + L * pad((size_t)B * steps * 3 * d) + 6 * pad(B * d) + pad(B * 4 * d) +
         pad((size_t)B * m->vocab_size) + 4096;
}

int sf_slate_generate_f32(const sf_slate_decoder* m, const float* slots, int B, int steps, long long* tokens_out,
                          float* logits_out, void* ws, size_t ws_bytes, void* stream) {
  SF_REQUIRE(m && slots && tokens_out && ws, "sf_slate_generate_f32: null pointer");
  SF_REQUIRE(B >= 1 && steps >= 1 && steps - 1 <= m->max_len, "sf_slate_generate_f32: bad batch / step count");
  SF_REQUIRE(m->d_model > 0 && (m->d_model % 4) == 0 && m->num_heads > 0 && (m->d_model % m->num_heads) == 0 &&
                 m->num_layers >= 0 && m->vocab_size > 0 && m->num_slots > 0, "sf_slate_generate_f32: bad decoder shape");
  SF_REQUIRE(m->in_proj_w && m->in_proj_b && m->tok_emb && m->pos_emb && m->lnf_g && m->lnf_b && m->head_w &&
                 (m->num_layers == 0 || m->blocks), "sf_slate_generate_f32: null weight");
  SF_REQUIRE(ws_bytes >= sf_slate_generate_workspace_bytes(m, B, steps), "sf_slate_generate_f32: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int d = m->d_model, H = m->num_heads, N = m->num_slots, V = m->vocab_size, NL = m->num_layers;
  const float eps = 1e-5f;
  char* p = (char*)ws;
  auto take = [&](size_t nfloat) {
    float* r = (float*)p;
    p += ((nfloat * sizeof(float)) + 255) & ~(size_t)255;
    return r;
  };
  float* mem = take((size_t)B * N * d);
  float* memkv[16];
  float* cache[16];
  SF_REQUIRE(NL <= 16, "sf_slate_generate_f32: at most 16 decoder blocks");
  for (int i = 0; i < NL; ++i) memkv[i] = take((size_t)B * N * 2 * d);
  for (int i = 0; i < NL; ++i) cache[i] = take((size_t)B * steps * 3 * d);
  float* x = take((size_t)B * d);
  float* xn = take((size_t)B * d);
  float* att = take((size_t)B * d);
  float* qc = take((size_t)B * d);
  float* x2 = take((size_t)B * d);
  float* x3 = take((size_t)B * d);
  float* hid = take((size_t)B * 4 * d);
  float* lg = take((size_t)B * V);
  const SfRowMap rd = sf_rows(d);
  // slots -> memory, and every block's cross-attention keys / values (once)
  SF_TRY(sf_linear_ex(slots, rd, m->in_proj_w, m->in_proj_b, nullptr, nullptr, eps, nullptr, rd, 0, mem, rd, B * N, d, d, 0, st));
  for (int i = 0; i < NL; ++i) {
    const sf_slate_block& k = m->blocks[i];
    SF_REQUIRE(k.ln1_g && k.ln1_b && k.wqkv && k.wo && k.ln2_g && k.ln2_b && k.wq_c && k.wkv_c && k.wo_c && k.ln3_g && k.ln3_b &&
                   k.w1 && k.b1 && k.w2 && k.b2, "sf_slate_generate_f32: null block weight");
    SF_TRY(sf_linear_ex(mem, rd, k.wkv_c, nullptr, nullptr, nullptr, eps, nullptr, rd, 0, memkv[i], sf_rows(2 * d), B * N, 2 * d, d,
                        0, st));
  }
  const long long cbs = (long long)steps * 3 * d;
  for (int t = 0; t < steps; ++t) {
    // token t-1 (BOS at t = 0) + position t
    {
      const long long total4 = (long long)B * (d / 4);
      hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, tokens_out + (t > 0 ? t - 1 : 0),
                         (long long)steps, t == 0 ? (long long)V : -1LL, m->tok_emb, m->pos_emb + (long long)t * d, x, 1, d, total4);
      SF_CHECK_LAUNCH();
    }
    float* cur = x;
    float* nxt = x2;
    for (int i = 0; i < NL; ++i) {
      const sf_slate_block& k = m->blocks[i];
      const SfRowMap crow = sf_rows_batched(3 * d, 1, cbs, (long long)t * 3 * d);   // row (b) of the cache at token t
      if (k.is_first) {   // the first block normalises its input in place (steve_transformer.py:186-190)
        SF_TRY(sf_layernorm_ex(cur, rd, k.ln1_g, k.ln1_b, xn, rd, B, d, eps, st));
        SF_TRY(sf_linear_ex(xn, rd, k.wqkv, nullptr, nullptr, nullptr, eps, nullptr, rd, 0, cache[i], crow, B, 3 * d, d, 0, st));
        cur = xn;
      } else {
        SF_TRY(sf_linear_ex(cur, rd, k.wqkv, nullptr, k.ln1_g, k.ln1_b, eps, nullptr, rd, 0, cache[i], crow, B, 3 * d, d, 0, st));
      }
      SF_TRY(sf_slate_attention_strided_f32(cache[i] + (long long)t * 3 * d, cache[i] + d, cache[i] + 2 * d, att, 3 * d, 3 * d,
                                            3 * d, d, cbs, cbs, cbs, d, B, 1, t + 1, H, d / H, 0, stream));
      SF_TRY(sf_linear_ex(att, rd, k.wo, nullptr, nullptr, nullptr, eps, cur, rd, 0, nxt, rd, B, d, d, 0, st));
      // cross-attention to the slots
      SF_TRY(sf_linear_ex(nxt, rd, k.wq_c, nullptr, k.ln2_g, k.ln2_b, eps, nullptr, rd, 0, qc, rd, B, d, d, 0, st));
      SF_TRY(sf_slate_attention_strided_f32(qc, memkv[i], memkv[i] + d, att, d, 2 * d, 2 * d, d, d, (long long)N * 2 * d,
                                            (long long)N * 2 * d, d, B, 1, N, H, d / H, 0, stream));
      SF_TRY(sf_linear_ex(att, rd, k.wo_c, nullptr, nullptr, nullptr, eps, nxt, rd, 0, x3, rd, B, d, d, 0, st));
      // FFN
      SF_TRY(sf_linear_ex(x3, rd, k.w1, k.b1, k.ln3_g, k.ln3_b, eps, nullptr, rd, 0, hid, sf_rows(4 * d), B, 4 * d, d, 1, st));
      float* dst = (cur == x || cur == xn) ? x2 : x;
      if (dst == x3) dst = x;
      SF_TRY(sf_linear_ex(hid, sf_rows(4 * d), k.w2, k.b2, nullptr, nullptr, eps, x3, rd, 0, dst, rd, B, d, 4 * d, 0, st));
      cur = dst;
      nxt = (cur == x) ? x2 : x;
    }
    // final LayerNorm + vocabulary head; logits of step t go to logits_out[b][t] (if wanted) and to the argmax
    float* ldst = logits_out ? logits_out : lg;
    const SfRowMap lmap = logits_out ? sf_rows_batched(V, 1, (long long)steps * V, (long long)t * V) : sf_rows(V);
    SF_TRY(sf_linear_ex(cur, rd, m->head_w, nullptr, m->lnf_g, m->lnf_b, eps, nullptr, rd, 0, ldst, lmap, B, V, d, 0, st));
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)B), dim3(256), 0, st, logits_out ? logits_out + (long long)t * V : lg,
                       logits_out ? (long long)steps * V : (long long)V, tokens_out + t, (long long)steps, V);
    SF_CHECK_LAUNCH();
  }
  return 0;
}

}  // extern "C"
