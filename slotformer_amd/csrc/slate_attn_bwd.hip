// Backward of the slot-conditioned decoder's attention (STEVE image side under autograd): the adjoint of
// sf_slate_attention_strided_f32 (steve_transformer.py:61-116: causal self-attention over the patch tokens, cross-attention
// from the tokens to the slots), flash style -- no L x L matrix ever reaches memory.
//
//   stats kernel : per 64-query block, one pass over its key blocks -> row log-sum-exp LSE_i and D_i = dO_i . O_i
//   main kernel  : one workgroup per (64-key block j, head, sequence) keeps K_j, V_j and the dK_j, dV_j accumulators on chip and
//                  walks the query blocks i (i >= j when causal): S = Q_i K_j^T, P = exp(S - LSE_i) (masked), dV_j += P^T dO_i,
//                  dP = dO_i V_j^T, dS = P (dP - D_i), dK_j += dS^T Q_i, and dQ_i += dS K_j by float atomics.
// Every product is a set of 32x32 tiles of exact-f32 MFMA (v_mfma_f32_32x32x2_f32) read from zero-padded LDS tiles.
#include <math.h>

#include "../../include/slotformer_hip.h"
#include "sf_internal.h"
#include "bf16_planes.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// one 32x32 output tile, C[i][j] = sum_k a(i, k) b(k, j), operands read from LDS through the functors.
// BF3 = false: exact-f32 MFMA (32x32x2, 64 cycles per 2 k).  BF3 = true (library precision modes 1 / 2): the operands are
// split into bf16 hi + lo on the fly and contracted as hi*hi + hi*lo + lo*hi on the 32x32x16 bf16 MFMA -- 5x less
// matrix-pipe time per k, which is what bounds these kernels (K must then be a multiple of 16).
template <bool BF3, class FA, class FB>
__device__ __forceinline__ f32x16 sab_mm32(FA a, FB b, int K, int lane) {
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int i0 = lane & 31, kk = lane >> 5;
  if constexpr (BF3) {
    for (int k = 0; k < K; k += 16) {
      f32x8 av, bv;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        av[j] = a(i0, k + 8 * kk + j);
        bv[j] = b(k + 8 * kk + j, i0);
      }
      const bf16x8 ah = __builtin_convertvector(av, bf16x8), bh = __builtin_convertvector(bv, bf16x8);
      const bf16x8 al = __builtin_convertvector(av - __builtin_convertvector(ah, f32x8), bf16x8);
      const bf16x8 bl = __builtin_convertvector(bv - __builtin_convertvector(bh, f32x8), bf16x8);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    }
  } else {
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a(i0, k + kk), b(k + kk, i0), acc, 0, 0, 0);
  }
  return acc;
}
// both operands k-contiguous in LDS (rows 16-byte aligned): A[i][k] at arow + i * pa, B[j][k] at brow + j * pb -- two ds_read_b128
// per operand per k16 step instead of eight scalar reads
template <bool BF3>
__device__ __forceinline__ f32x16 sab_mm32_kk(const float* A, int pa, const float* Bm, int pb, int K, int lane) {
  if constexpr (!BF3) {
    return sab_mm32<false>([&](int i, int k) { return A[i * pa + k]; }, [&](int k, int j) { return Bm[j * pb + k]; }, K, lane);
  } else {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float* ar = A + (lane & 31) * pa + 8 * (lane >> 5);
    const float* br = Bm + (lane & 31) * pb + 8 * (lane >> 5);
    for (int k = 0; k < K; k += 16) {
      const f32x4 a0 = *(const f32x4*)(ar + k), a1 = *(const f32x4*)(ar + k + 4);
      const f32x4 b0 = *(const f32x4*)(br + k), b1 = *(const f32x4*)(br + k + 4);
      const f32x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      const f32x8 bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      const bf16x8 ah = __builtin_convertvector(av, bf16x8), bh = __builtin_convertvector(bv, bf16x8);
      const bf16x8 al = __builtin_convertvector(av - __builtin_convertvector(ah, f32x8), bf16x8);
      const bf16x8 bl = __builtin_convertvector(bv - __builtin_convertvector(bh, f32x8), bf16x8);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    }
    return acc;
  }
}
#define SAB_ROW(r, lane) (((r) & 3) + 8 * ((r) >> 2) + 4 * ((lane) >> 5))

struct SabArgs {
  const float *q, *k, *v, *o, *dout;
  float *dq, *dk, *dv, *lse, *dsum;   // lse, dsum: [B][H][Lq]
  int ldq, ldk, ldv, ldo;
  long long q_bs, k_bs, v_bs, o_bs;
  int Lq, Lk, H, causal;
  float scale;
  uint32_t drop_seed, drop_thresh;   // dropout on the attention weights: element ((b*H + h)*Lq + q)*Lk + k, see rollout_train.hip
  float drop_scale;
  int have_lse;                      // lse was written by the training forward: the stats kernel only computes dsum
};
__device__ __forceinline__ float sab_drop(const SabArgs& p, int b, int h, int q, int k) {
  if (!p.drop_thresh) return 1.f;
  const uint32_t idx = (uint32_t)((((long long)b * p.H + h) * p.Lq + q) * p.Lk + k);
  return (sf_mix32(idx ^ p.drop_seed) >> 8) >= p.drop_thresh ? p.drop_scale : 0.f;
}

// rows [r0, r0 + 64) of a [L, ld] matrix (head slice at +hoff) -> zero-padded LDS tile [64][P]
template <int HDP>
__device__ __forceinline__ void sab_load(const float* base, int ld, int r0, int L, int hd, float mul, float* dst) {
  constexpr int P = HDP + 4;
  for (int i = threadIdx.x; i < 64 * HDP; i += 256) {
    const int r = i / HDP, c = i - r * HDP;
    dst[r * P + c] = (r0 + r < L && c < hd) ? base[(long long)(r0 + r) * ld + c] * mul : 0.f;
  }
}

// the same tile in two steps, so that the global loads of the NEXT tile fly while the current one is being used:
// fetch -> registers (16-byte loads from clamped addresses, zeroed by a select), put -> LDS
template <int HDP>
struct SabTile {
  f32x4 v[HDP / 16];
};
template <int HDP>
__device__ __forceinline__ void sab_fetch(const float* base, int ld, int r0, int L, int hd, SabTile<HDP>& t) {
  constexpr int C4 = HDP / 4;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int n = 0; n < HDP / 16; ++n) {
    const int i = threadIdx.x + 256 * n, r = i / C4, c = (i - r * C4) * 4;
    const f32x4 x = *(const f32x4*)(base + (long long)min(r0 + r, L - 1) * ld + min(c, hd - 4));
    t.v[n] = (r0 + r < L && c < hd) ? x : z;
  }
}
template <int HDP>
__device__ __forceinline__ void sab_put(const SabTile<HDP>& t, float mul, float* dst) {
  constexpr int C4 = HDP / 4, P = HDP + 4;
#pragma unroll
  for (int n = 0; n < HDP / 16; ++n) {
    const int i = threadIdx.x + 256 * n, r = i / C4, c = (i - r * C4) * 4;
    *(f32x4*)(dst + r * P + c) = t.v[n] * mul;
  }
}

template <int HDP, bool BF3>
__global__ __launch_bounds__(256) void slate_attn_stats_kernel(SabArgs p, int hd) {
  constexpr int P = HDP + 4;
  extern __shared__ float lds[];
  float* Qs = lds;
  float* Ks = Qs + 64 * P;
  float* Ss = Ks + 64 * P;     // [64][68]
  float* rm = Ss + 64 * 68;    // running max [64]
  float* rl = rm + 64;         // running sum [64]
  const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q0 = qb * 64;
  if (!p.have_lse) sab_load<HDP>(p.q + (long long)b * p.q_bs + h * hd, p.ldq, q0, p.Lq, hd, p.scale, Qs);
  if (tid < 64) {
    rm[tid] = -INFINITY;
    rl[tid] = 0.f;
  }
  const int nkb = p.have_lse ? 0 : (p.causal ? qb + 1 : (p.Lk + 63) / 64);
  for (int kb = 0; kb < nkb; ++kb) {
    __syncthreads();
    sab_load<HDP>(p.k + (long long)b * p.k_bs + h * hd, p.ldk, kb * 64, p.Lk, hd, 1.f, Ks);
    __syncthreads();
    {
      const int ti = (wave >> 1) * 32, tj = (wave & 1) * 32;
      const f32x16 acc = sab_mm32_kk<BF3>(Qs + ti * P, P, Ks + tj * P, P, HDP, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) Ss[(ti + SAB_ROW(r, lane)) * 68 + tj + (lane & 31)] = acc[r];
    }
    __syncthreads();
    for (int r = wave; r < 64; r += 4) {   // online max / sum of row r over this key block
      const int qi = q0 + r, kj = kb * 64 + lane;
      const bool ok = qi < p.Lq && kj < p.Lk && (!p.causal || kj <= qi);
      const float s = ok ? Ss[r * 68 + lane] : -INFINITY;
      const float mx = sf_wave_max(s);
      const float mo = rm[r], mn = fmaxf(mo, mx);
      const float e = (ok && mn > -INFINITY) ? expf(s - mn) : 0.f;
      const float sum = sf_wave_sum(e);
      if (lane == 0) {
        rl[r] = rl[r] * ((mo == -INFINITY) ? 0.f : expf(mo - mn)) + sum;
        rm[r] = mn;
      }
    }
  }
  __syncthreads();
  // LSE and D = dO . O  (one thread group of 4 per row)
  for (int r = wave * 16 + (lane >> 2); r < 64; r += 64) {
    const int qi = q0 + r;
    float d = 0.f;
    if (qi < p.Lq) {
      const float* o = p.o + (long long)b * p.o_bs + (long long)qi * p.ldo + h * hd;
      const float* g = p.dout + (long long)b * p.o_bs + (long long)qi * p.ldo + h * hd;
      for (int c = lane & 3; c < hd; c += 4) d += o[c] * g[c];
    }
    d += sf_dpp<0xB1>(d);
    d += sf_dpp<0x4E>(d);
    if ((lane & 3) == 0 && qi < p.Lq) {
      const long long idx = ((long long)b * p.H + h) * p.Lq + qi;
      if (!p.have_lse) p.lse[idx] = rm[r] + logf(rl[r]);
      p.dsum[idx] = d;
    }
  }
}

template <int HDP, bool BF3>
__global__ __launch_bounds__(256) void slate_attn_bwd_kernel(SabArgs p, int hd) {
  constexpr int P = HDP + 4, CT = HDP / 32;   // channel tiles
  extern __shared__ float lds[];
  float* Ks = lds;
  float* Vs = Ks + 64 * P;
  float* Qs = Vs + 64 * P;
  float* Gs = Qs + 64 * P;     // dO
  float* Ps = Gs + 64 * P;     // [64 queries][65]
  float* Ds = Ps + 64 * 68;    // dS
  float* ls = Ds + 64 * 68;    // lse [64], dsum [64]
  const int kb = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int k0 = kb * 64;
  sab_load<HDP>(p.k + (long long)b * p.k_bs + h * hd, p.ldk, k0, p.Lk, hd, 1.f, Ks);
  sab_load<HDP>(p.v + (long long)b * p.v_bs + h * hd, p.ldv, k0, p.Lk, hd, 1.f, Vs);
  // accumulators: dV_j and dK_j are [64 keys][HDP]: 2 * CT tiles each, tile t -> wave t % 4 (CT <= 2: at most one each per wave)
  f32x16 dvacc, dkacc;
#pragma unroll
  for (int i = 0; i < 16; ++i) dvacc[i] = dkacc[i] = 0.f;
  const int at = wave;                           // accumulator tile of this wave (valid if at < 2 * CT)
  const int ati = (at / CT) * 32, atj = (at % CT) * 32;
  const int nqb = (p.Lq + 63) / 64;
  const float* qbase = p.q + (long long)b * p.q_bs + h * hd;
  const float* gbase = p.dout + (long long)b * p.o_bs + h * hd;
  const int qb0 = p.causal ? kb : 0;
  SabTile<HDP> tq, tg;
  float pl = 0.f, pd = 0.f;   // lse / dsum of the prefetched block (threads < 64)
  auto prefetch = [&](int qbn) {
    sab_fetch<HDP>(qbase, p.ldq, qbn * 64, p.Lq, hd, tq);
    sab_fetch<HDP>(gbase, p.ldo, qbn * 64, p.Lq, hd, tg);
    if (tid < 64) {
      const long long idx = ((long long)b * p.H + h) * p.Lq + min(qbn * 64 + tid, p.Lq - 1);
      pl = p.lse[idx];
      pd = p.dsum[idx];
    }
  };
  if (qb0 < nqb) prefetch(qb0);
  for (int qb = qb0; qb < nqb; ++qb) {
    const int q0 = qb * 64;
    __syncthreads();
    sab_put<HDP>(tq, p.scale, Qs);
    sab_put<HDP>(tg, 1.f, Gs);
    if (tid < 64) {
      ls[tid] = pl;
      ls[64 + tid] = pd;
    }
    __syncthreads();
    if (qb + 1 < nqb) prefetch(qb + 1);   // in flight during the five products below
    const int ti = (wave >> 1) * 32, tj = (wave & 1) * 32;
    // S tile and dP tile of this wave (queries ti.., keys tj..)
    const f32x16 s = sab_mm32_kk<BF3>(Qs + ti * P, P, Ks + tj * P, P, HDP, lane);
    const f32x16 dp = sab_mm32_kk<BF3>(Gs + ti * P, P, Vs + tj * P, P, HDP, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = ti + SAB_ROW(r, lane), kc = tj + (lane & 31);
      const int qi = q0 + qr, kj = k0 + kc;
      const bool ok = qi < p.Lq && kj < p.Lk && (!p.causal || kj <= qi);
      const float pv = ok ? expf(s[r] - ls[qr]) : 0.f;
      const float mk = ok ? sab_drop(p, b, h, qi, kj) : 0.f;   // dropout factor on this attention weight (1 when off)
      Ps[qr * 68 + kc] = pv * mk;                               // what multiplied V in the forward pass
      Ds[qr * 68 + kc] = pv * (dp[r] * mk - ls[64 + qr]);
    }
    __syncthreads();
    // dV_j += P^T dO_i ; dK_j += dS^T (Q_i * scale)   (contraction over the 64 queries)
    if (at < 2 * CT) {
      // tiles 0 .. 2*CT-1 cover [64 keys][HDP]; ati = key offset, atj = channel offset
      const f32x16 a1 = sab_mm32<BF3>([&](int i, int kk) { return Ps[kk * 68 + ati + i]; }, [&](int kk, int j) { return Gs[kk * P + atj + j]; }, 64,
                                 lane);
      const f32x16 a2 = sab_mm32<BF3>([&](int i, int kk) { return Ds[kk * 68 + ati + i]; }, [&](int kk, int j) { return Qs[kk * P + atj + j]; }, 64,
                                 lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dvacc[r] += a1[r];
        dkacc[r] += a2[r];
      }
    }
    // dQ_i += scale * dS K_j  (tiles over [64 queries][HDP]; atomics: several key blocks add into the same rows)
    for (int t = wave; t < 2 * CT; t += 4) {
      const int qi0 = (t / CT) * 32, c0 = (t % CT) * 32;
      const f32x16 a3 = sab_mm32<BF3>([&](int i, int kk) { return Ds[(qi0 + i) * 68 + kk]; }, [&](int kk, int j) { return Ks[kk * P + c0 + j]; }, 64,
                                 lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = q0 + qi0 + SAB_ROW(r, lane), c = c0 + (lane & 31);
        if (qi < p.Lq && c < hd) atomicAdd(p.dq + (long long)b * p.q_bs + (long long)qi * p.ldq + h * hd + c, a3[r] * p.scale);
      }
    }
  }
  if (at < 2 * CT) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kj = k0 + ati + SAB_ROW(r, lane), c = atj + (lane & 31);
      if (kj < p.Lk && c < hd) {
        p.dv[(long long)b * p.v_bs + (long long)kj * p.ldv + h * hd + c] = dvacc[r];
        p.dk[(long long)b * p.k_bs + (long long)kj * p.ldk + h * hd + c] = dkacc[r];   // Q was scaled when loaded
      }
    }
  }
}

// The same kernel for 32 < head_dim <= 48 in the split-bf16 modes (the Physion decoder: width 192, 4 heads), shaped so that TWO
// workgroups share a CU (two waves per SIMD hide each other's LDS / MFMA latencies; the 64-wide version above leaves one):
//  * LDS tiles are 48 + 4 floats wide and the S / dP contractions run over 48 channels, not the zero-padded 64;
//  * the K_j and V_j operand fragments of S and dP (fixed for the whole query loop) live in registers, already split into
//    bf16 hi / lo, so V_j needs no LDS tile at all and those two products read one operand from LDS instead of two.
// 3 tiles of [64][52] + P / dS [64][68] each = 75 KB per workgroup.  Output-channel tiles still span 64 columns: the reads
// past column 51 of a row land in the next row and only feed output columns >= head_dim, which are never stored.
struct SabFrag {
  bf16x8 h[3], l[3];
};
__device__ __forceinline__ void sab_frag_load(const float* base, int ld, int row, int L, int hd, int lane, SabFrag& f) {
  const float* src = base + (long long)min(row, L - 1) * ld;
  const bool rok = row < L;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int c = 16 * s + 8 * (lane >> 5);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 a = *(const f32x4*)(src + min(c, hd - 4)), b = *(const f32x4*)(src + min(c + 4, hd - 4));
    a = (rok && c < hd) ? a : z;
    b = (rok && c + 4 < hd) ? b : z;
    const f32x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    f.h[s] = __builtin_convertvector(v, bf16x8);
    f.l[s] = __builtin_convertvector(v - __builtin_convertvector(f.h[s], f32x8), bf16x8);
  }
}
// Every LDS tile of this kernel holds its elements ALREADY split: one 32-bit word = bf16 hi (upper half) | bf16 lo (lower half),
// produced once where the element is written.  An operand fragment is then 8 words -> two v_perm_b32 per word pair instead of
// the ~24 conversion instructions per fragment that the f32 tiles cost every time a wave read them (30 fragments per block
// pair and wave: the conversions were the largest share of this kernel's issue slots).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned sab_pack1(float v) {
  const __bf16 h = (__bf16)v;
  const __bf16 l = (__bf16)(v - (float)h);
  return ((unsigned)__builtin_bit_cast(unsigned short, h) << 16) | (unsigned)__builtin_bit_cast(unsigned short, l);
}
__device__ __forceinline__ void sab_unpack8(const unsigned (&w)[8], bf16x8& h, bf16x8& l) {
  u32x4 hv, lv;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    hv[q] = __builtin_amdgcn_perm(w[2 * q + 1], w[2 * q], 0x07060302u);
    lv[q] = __builtin_amdgcn_perm(w[2 * q + 1], w[2 * q], 0x05040100u);
  }
  h = __builtin_bit_cast(bf16x8, hv);
  l = __builtin_bit_cast(bf16x8, lv);
}
// 32x32 tile, A rows k-contiguous in a packed LDS tile (pitch pa words, 48 channels), B from pre-split register fragments
__device__ __forceinline__ f32x16 sab_mm32_frag(const unsigned* A, int pa, const SabFrag& f, int lane) {
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const unsigned* ar = A + (lane & 31) * pa + 8 * (lane >> 5);
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const u32x4 a0 = *(const u32x4*)(ar + 16 * s), a1 = *(const u32x4*)(ar + 16 * s + 4);
    const unsigned w[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    bf16x8 ah, al;
    sab_unpack8(w, ah, al);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, f.h[s], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, f.l[s], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, f.h[s], acc, 0, 0, 0);
  }
  return acc;
}
// packed variants of sab_load / sab_put for the 48-wide tiles
__device__ __forceinline__ void sab_load_packed48(const float* base, int ld, int r0, int L, int hd, unsigned* dst) {
  for (int i = threadIdx.x; i < 64 * 48; i += 256) {
    const int r = i / 48, c = i - r * 48;
    dst[r * 52 + c] = sab_pack1((r0 + r < L && c < hd) ? base[(long long)(r0 + r) * ld + c] : 0.f);
  }
}
__device__ __forceinline__ void sab_put_packed48(const SabTile<48>& t, float mul, unsigned* dst) {
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    const int i = threadIdx.x + 256 * n, r = i / 12, c = (i - r * 12) * 4;
    const f32x4 v = t.v[n] * mul;
    const u32x4 w = {sab_pack1(v[0]), sab_pack1(v[1]), sab_pack1(v[2]), sab_pack1(v[3])};
    *(u32x4*)(dst + r * 52 + c) = w;
  }
}
// acc += A B for operands read word-wise (packed hi|lo) through the functors; the k loop is kept rolled: with two workgroups
// per CU each wave has 256 registers, and the other wave of the SIMD covers the latency an unrolled loop would hide
template <class FA, class FB>
__device__ __forceinline__ void sab_mm32_acc(f32x16& acc, FA a, FB b, int K, int lane) {
  const int i0 = lane & 31, kk = lane >> 5;
#pragma unroll 1
  for (int k = 0; k < K; k += 16) {
    unsigned aw[8], bw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      aw[j] = a(i0, k + 8 * kk + j);
      bw[j] = b(k + 8 * kk + j, i0);
    }
    bf16x8 ah, al, bh, bl;
    sab_unpack8(aw, ah, al);
    sab_unpack8(bw, bh, bl);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
  }
}
__global__ __launch_bounds__(256, 2) void slate_attn_bwd48_kernel(SabArgs p, int hd) {
  constexpr int HDL = 48, P = HDL + 4;
  extern __shared__ float lds[];
  unsigned* Ks = (unsigned*)lds;   // every tile: packed bf16 hi | lo words
  unsigned* Qs = Ks + 64 * P;
  unsigned* Gs = Qs + 64 * P;      // dO
  unsigned* Ps = Gs + 64 * P;      // [64 queries][68]
  unsigned* Ds = Ps + 64 * 68;     // dS
  float* ls = (float*)(Ds + 64 * 68);   // lse [64], dsum [64]
  const int kb = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int k0 = kb * 64;
  const int ti = (wave >> 1) * 32, tj = (wave & 1) * 32;   // S / dP tile of this wave: queries ti.., keys tj..
  sab_load_packed48(p.k + (long long)b * p.k_bs + h * hd, p.ldk, k0, p.Lk, hd, Ks);
  SabFrag kf, vf;
  sab_frag_load(p.k + (long long)b * p.k_bs + h * hd, p.ldk, k0 + tj + (lane & 31), p.Lk, hd, lane, kf);
  sab_frag_load(p.v + (long long)b * p.v_bs + h * hd, p.ldv, k0 + tj + (lane & 31), p.Lk, hd, lane, vf);
  f32x16 dvacc, dkacc;
#pragma unroll
  for (int i = 0; i < 16; ++i) dvacc[i] = dkacc[i] = 0.f;
  const int ati = (wave >> 1) * 32, atj = (wave & 1) * 32;   // dV / dK tile of this wave: keys ati.., channels atj..
  const int nqb = (p.Lq + 63) / 64;
  const float* qbase = p.q + (long long)b * p.q_bs + h * hd;
  const float* gbase = p.dout + (long long)b * p.o_bs + h * hd;
  const int qb0 = p.causal ? kb : 0;
  SabTile<HDL> tq, tg;
  float pl = 0.f, pd = 0.f;
  auto prefetch = [&](int qbn) {
    sab_fetch<HDL>(qbase, p.ldq, qbn * 64, p.Lq, hd, tq);
    sab_fetch<HDL>(gbase, p.ldo, qbn * 64, p.Lq, hd, tg);
    if (tid < 64) {
      const long long idx = ((long long)b * p.H + h) * p.Lq + min(qbn * 64 + tid, p.Lq - 1);
      pl = p.lse[idx];
      pd = p.dsum[idx];
    }
  };
  if (qb0 < nqb) prefetch(qb0);
  for (int qb = qb0; qb < nqb; ++qb) {
    const int q0 = qb * 64;
    __syncthreads();
    sab_put_packed48(tq, p.scale, Qs);
    sab_put_packed48(tg, 1.f, Gs);
    if (tid < 64) {
      ls[tid] = pl;
      ls[64 + tid] = pd;
    }
    __syncthreads();
    if (qb + 1 < nqb) prefetch(qb + 1);
    const f32x16 s = sab_mm32_frag(Qs + ti * P, P, kf, lane);
    const f32x16 dp = sab_mm32_frag(Gs + ti * P, P, vf, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = ti + SAB_ROW(r, lane), kc = tj + (lane & 31);
      const int qi = q0 + qr, kj = k0 + kc;
      const bool ok = qi < p.Lq && kj < p.Lk && (!p.causal || kj <= qi);
      const float pv = ok ? expf(s[r] - ls[qr]) : 0.f;
      const float mk = ok ? sab_drop(p, b, h, qi, kj) : 0.f;
      Ps[qr * 68 + kc] = sab_pack1(pv * mk);
      Ds[qr * 68 + kc] = sab_pack1(pv * (dp[r] * mk - ls[64 + qr]));
    }
    __syncthreads();
    sab_mm32_acc(dvacc, [&](int i, int kk) { return Ps[kk * 68 + ati + i]; }, [&](int kk, int j) { return Gs[kk * P + atj + j]; }, 64, lane);
    sab_mm32_acc(dkacc, [&](int i, int kk) { return Ds[kk * 68 + ati + i]; }, [&](int kk, int j) { return Qs[kk * P + atj + j]; }, 64, lane);
    {
      const int qi0 = (wave >> 1) * 32, c0 = (wave & 1) * 32;
      f32x16 a3;
#pragma unroll
      for (int i = 0; i < 16; ++i) a3[i] = 0.f;
      sab_mm32_acc(a3, [&](int i, int kk) { return Ds[(qi0 + i) * 68 + kk]; }, [&](int kk, int j) { return Ks[kk * P + c0 + j]; }, 64, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = q0 + qi0 + SAB_ROW(r, lane), c = c0 + (lane & 31);
        if (qi < p.Lq && c < hd) atomicAdd(p.dq + (long long)b * p.q_bs + (long long)qi * p.ldq + h * hd + c, a3[r] * p.scale);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int kj = k0 + ati + SAB_ROW(r, lane), c = atj + (lane & 31);
    if (kj < p.Lk && c < hd) {
      p.dv[(long long)b * p.v_bs + (long long)kj * p.ldv + h * hd + c] = dvacc[r];
      p.dk[(long long)b * p.k_bs + (long long)kj * p.ldk + h * hd + c] = dkacc[r];
    }
  }
}

// The 48-wide kernel again with every LDS tile as two bf16 PLANES (hi, lo) instead of packed words, in the layouts the 32x32x16 MFMA operands want:
//  * Q_i, dO_i (and K_j, V_j once, before the loop) query-major [64][56]: the A operands of S = Q K^T and dP = dO V^T are one ds_read_b128 per plane
//    and k16-step; the B operands of the products that contract over QUERIES (dV += P^T dO, dK += dS^T Q) come out of the same tiles through
//    ds_read_b64_tr_b16 (two transposing reads per plane and step: lane (i, kk) receives rows 8 kk .. 8 kk + 7 of column i);
//  * P^T, dS^T key-major [64 keys][72]: written from the S / dP accumulators as they lie (a lane holds one key and runs of four consecutive queries:
//    one ds_write_b64 per run and plane), read as A operands of dV / dK by ds_read_b128, and -- transposing -- as the A operand of dQ += dS K;
//  * K_j's fragments for S (B operand, contraction over channels) and for dQ (B operand, contraction over keys) and V_j's for dP live in registers.
// No conversion, no v_perm and no scalar LDS read in the loop: ~650 instructions per query block and wave instead of ~1650 (54 MFMAs either way).
// exp runs in base 2 (Q' = Q scale log2(e), LSE' = LSE log2(e); dK is multiplied by ln 2 at the end).  65.5 KB of LDS: two workgroups per CU.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
namespace {
constexpr int V2_QP = 112, V2_TP = 144;          // row pitches in BYTES: query-major tiles [64][56 bf16], key-major tiles [64][72 bf16]
constexpr int V2_QT = 64 * V2_QP, V2_TT = 64 * V2_TP;
constexpr int V2_QH = 0, V2_QL = V2_QT, V2_GH = 2 * V2_QT, V2_GL = 3 * V2_QT;
constexpr int V2_PH = 4 * V2_QT, V2_PL = V2_PH + V2_TT, V2_DH = V2_PH + 2 * V2_TT, V2_DL = V2_PH + 3 * V2_TT;
constexpr int V2_LS = V2_PH + 4 * V2_TT;         // lse' [64], dsum [64]
constexpr int V2_LDS = V2_LS + 512 + 256;        // (+ slack: the transposing reads of the second channel tile run past column 55 of the last row)
// prefetched rows (SabTile<48>: three float4 per thread) -> hi / lo planes of a query-major tile
__device__ __forceinline__ void v2_put(const SabTile<48>& t, float mul, char* lds, int hoff, int loff) {
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    const int i = threadIdx.x + 256 * n, r = i / 12, c = (i - r * 12) * 4;
    const f32x4 v = t.v[n] * mul;
    unsigned h0, l0, h1, l1;
    pl_split2(v[0], v[1], h0, l0);
    pl_split2(v[2], v[3], h1, l1);
    *(u32x2*)(lds + hoff + r * V2_QP + c * 2) = u32x2{h0, h1};
    *(u32x2*)(lds + loff + r * V2_QP + c * 2) = u32x2{l0, l1};
  }
}
}  // namespace
#ifdef SAB_STAMPS
__device__ long long sab_ts[16];
#define SABT(i) do { if (stamp) { const long long c_ = __builtin_readcyclecounter(); acc_ts[i] += c_ - c_last; c_last = c_; } } while (0)
#else
#define SABT(i)
#endif
__global__ __launch_bounds__(256, 2) void slate_attn_bwd48v2_kernel(SabArgs p, int hd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ls = (float*)(smem + V2_LS);
  const int kb = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k0 = kb * 64;
  const int ti = (wave >> 1) * 32, tj = (wave & 1) * 32;   // S / dP tile: queries ti.., keys tj..;  dV / dK tile: keys ti.., channels tj..;  dQ tile: queries ti.., channels tj..
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
#ifdef SAB_STAMPS
  const bool stamp = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0;
  long long acc_ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c_last = __builtin_readcyclecounter();
#endif
  // ---- K_j, V_j -> planes (in the Q / dO tiles' place), then their fragments -> registers ----
  SabTile<48> tq, tg;
  const int nqb = (p.Lq + 63) / 64;
  const float* qbase = p.q + (long long)b * p.q_bs + h * hd;
  const float* gbase = p.dout + (long long)b * p.o_bs + h * hd;
  const int qb0 = p.causal ? kb : 0;
  float pl = 0.f, pd = 0.f;
  auto prefetch = [&](int qbn) {
    sab_fetch<48>(qbase, p.ldq, qbn * 64, p.Lq, hd, tq);
    sab_fetch<48>(gbase, p.ldo, qbn * 64, p.Lq, hd, tg);
    if (tid < 64) {
      const long long idx = ((long long)b * p.H + h) * p.Lq + min(qbn * 64 + tid, p.Lq - 1);
      pl = p.lse[idx];
      pd = p.dsum[idx];
    }
  };
  {
    SabTile<48> tk, tv;
    sab_fetch<48>(p.k + (long long)b * p.k_bs + h * hd, p.ldk, k0, p.Lk, hd, tk);
    sab_fetch<48>(p.v + (long long)b * p.v_bs + h * hd, p.ldv, k0, p.Lk, hd, tv);
    if (qb0 < nqb) prefetch(qb0);   // (the first query block's rows are on their way while K_j / V_j are laid out)
    // (columns 48 .. 55 of the query-major planes feed only output columns >= 48, which are never stored -- but they must hold finite values)
    for (int i = tid; i < 4 * 64; i += 256) *(uint4*)(smem + (i >> 6) * V2_QT + (i & 63) * V2_QP + 96) = uint4{0u, 0u, 0u, 0u};
    v2_put(tk, 1.f, smem, V2_QH, V2_QL);
    v2_put(tv, 1.f, smem, V2_GH, V2_GL);
  }
  __syncthreads();
  PlFrag kf[3], vf[3], kq[4];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    kf[s] = pl_rd(smem, V2_QH, V2_QL, V2_QP, tj, 16 * s, lane);   // B operand of S: lane (key, kk) holds eight channels
    vf[s] = pl_rd(smem, V2_GH, V2_GL, V2_QP, tj, 16 * s, lane);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) kq[s] = pl_rd_tr(smem, V2_QH, V2_QL, V2_QP, 16 * s, tj, lane);   // B operand of dQ: lane (channel, kk) holds eight keys
  f32x16 dvacc, dkacc;
#pragma unroll
  for (int i = 0; i < 16; ++i) dvacc[i] = dkacc[i] = 0.f;
  const int kc = tj + (lane & 31), half = lane >> 5;   // this lane's key column of the S / dP tile
  SABT(0);
  for (int qb = qb0; qb < nqb; ++qb) {
    const int q0 = qb * 64;
    __syncthreads();
    SABT(1);
    v2_put(tq, p.scale * LOG2E, smem, V2_QH, V2_QL);
    v2_put(tg, 1.f, smem, V2_GH, V2_GL);
    if (tid < 64) {
      ls[tid] = pl * LOG2E;
      ls[64 + tid] = pd;
    }
    __syncthreads();
    SABT(2);
    if (qb + 1 < nqb) prefetch(qb + 1);
    f32x16 s, dp;
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = dp[i] = 0.f;
#pragma unroll
    for (int st = 0; st < 3; ++st) {
      const PlFrag qa = pl_rd(smem, V2_QH, V2_QL, V2_QP, ti, 16 * st, lane);
      const PlFrag ga = pl_rd(smem, V2_GH, V2_GL, V2_QP, ti, 16 * st, lane);
      pl_mma(s, qa, kf[st]);
      pl_mma(dp, ga, vf[st]);
    }
    SABT(3);
    // ---- P = exp2(S' - LSE') (masked, dropped), dS = P (dP mask - D): this lane's key, four runs of four consecutive queries -> P^T / dS^T planes ----
    const bool full = (!p.causal || qb > kb) && q0 + 64 <= p.Lq && k0 + 64 <= p.Lk;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int qr = ti + 8 * g4 + 4 * half;
      const f32x4 l4 = *(const f32x4*)(ls + qr), d4 = *(const f32x4*)(ls + 64 + qr);
      float pv[4], dsv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g4 + e;
        float pe = __builtin_amdgcn_exp2f(s[r] - l4[e]);
        if (!full) {
          const int qi = q0 + qr + e, kj = k0 + kc;
          const bool ok = qi < p.Lq && kj < p.Lk && (!p.causal || kj <= qi);
          pe = ok ? pe : 0.f;
        }
        float mk = 1.f;
        if (p.drop_thresh) mk = sab_drop(p, b, h, min(q0 + qr + e, p.Lq - 1), min(k0 + kc, p.Lk - 1));
        pv[e] = pe * mk;
        dsv[e] = pe * (dp[r] * mk - d4[e]);
      }
      unsigned ph0, pl0, ph1, pl1, dh0, dl0, dh1, dl1;
      pl_split2(pv[0], pv[1], ph0, pl0);
      pl_split2(pv[2], pv[3], ph1, pl1);
      pl_split2(dsv[0], dsv[1], dh0, dl0);
      pl_split2(dsv[2], dsv[3], dh1, dl1);
      const int o = kc * V2_TP + qr * 2;
      *(u32x2*)(smem + V2_PH + o) = u32x2{ph0, ph1};
      *(u32x2*)(smem + V2_PL + o) = u32x2{pl0, pl1};
      *(u32x2*)(smem + V2_DH + o) = u32x2{dh0, dh1};
      *(u32x2*)(smem + V2_DL + o) = u32x2{dl0, dl1};
    }
    SABT(4);
    __syncthreads();
    SABT(5);
    // ---- dV_j += P^T dO_i,  dK_j += dS^T Q'_i  (contraction over the 64 queries);  dQ_i += scale dS K_j  (contraction over the 64 keys) ----
    f32x16 a3;
#pragma unroll
    for (int i = 0; i < 16; ++i) a3[i] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const PlFrag pa = pl_rd(smem, V2_PH, V2_PL, V2_TP, ti, 16 * st, lane);
      const PlFrag gb = pl_rd_tr(smem, V2_GH, V2_GL, V2_QP, 16 * st, tj, lane);
      pl_mma(dvacc, pa, gb);
      const PlFrag da = pl_rd(smem, V2_DH, V2_DL, V2_TP, ti, 16 * st, lane);
      const PlFrag qbf = pl_rd_tr(smem, V2_QH, V2_QL, V2_QP, 16 * st, tj, lane);
      pl_mma(dkacc, da, qbf);
      const PlFrag dsa = pl_rd_tr(smem, V2_DH, V2_DL, V2_TP, 16 * st, ti, lane);
      pl_mma(a3, dsa, kq[st]);
    }
    SABT(6);
    {
      float* dqp = p.dq + (long long)b * p.q_bs + h * hd + tj + (lane & 31);
      const bool cok = tj + (lane & 31) < hd;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = q0 + ti + SAB_ROW(r, lane);
        if (qi < p.Lq && cok) atomicAdd(dqp + (long long)qi * p.ldq, a3[r] * p.scale);
      }
    }
    SABT(7);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int kj = k0 + ti + SAB_ROW(r, lane), c = tj + (lane & 31);
    if (kj < p.Lk && c < hd) {
      p.dv[(long long)b * p.v_bs + (long long)kj * p.ldv + h * hd + c] = dvacc[r];
      p.dk[(long long)b * p.k_bs + (long long)kj * p.ldk + h * hd + c] = dkacc[r] * LN2;   // (Q' carried log2(e))
    }
  }
#ifdef SAB_STAMPS
  if (stamp) {
    for (int i = 0; i < 8; ++i) sab_ts[i] = acc_ts[i];
    sab_ts[8] = nqb - qb0;
  }
#endif
}
#ifdef SAB_STAMPS
extern "C" int sf_debug_read_ts_sab(long long* out16) {
  hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(sab_ts), sizeof(long long) * 16);
  return e == hipSuccess ? 0 : (int)e;
}
#endif

// Training forward: the same attention with dropout on the weights and the row log-sum-exp kept for the backward pass.
// One workgroup per (64-query block, head, sequence): S tiles -> LDS, online row max / sum, the (dropped) weights back to LDS,
// O accumulated as 32x32 tiles in registers and rescaled per row when the running max moves.
template <int HDP, bool BF3>
__global__ __launch_bounds__(256) void slate_attn_fwd_train_kernel(SabArgs p, float* __restrict__ out, int hd) {
  constexpr int P = HDP + 4, CT = HDP / 32;
  extern __shared__ float lds[];
  float* Qs = lds;
  float* Ks = Qs + 64 * P;
  float* Vs = Ks + 64 * P;
  float* Ss = Vs + 64 * P;     // [64][68]
  float* rm = Ss + 64 * 68;
  float* rl = rm + 64;
  float* rs = rl + 64;         // per-row rescale of this step
  const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q0 = qb * 64;
  sab_load<HDP>(p.q + (long long)b * p.q_bs + h * hd, p.ldq, q0, p.Lq, hd, p.scale, Qs);
  if (tid < 64) {
    rm[tid] = -INFINITY;
    rl[tid] = 0.f;
  }
  f32x16 oacc;
#pragma unroll
  for (int i = 0; i < 16; ++i) oacc[i] = 0.f;
  const int at = wave, ati = (at / CT) * 32, atj = (at % CT) * 32;   // O tile of this wave (valid if at < 2 * CT)
  const int nkb = p.causal ? qb + 1 : (p.Lk + 63) / 64;
  for (int kb = 0; kb < nkb; ++kb) {
    __syncthreads();
    sab_load<HDP>(p.k + (long long)b * p.k_bs + h * hd, p.ldk, kb * 64, p.Lk, hd, 1.f, Ks);
    sab_load<HDP>(p.v + (long long)b * p.v_bs + h * hd, p.ldv, kb * 64, p.Lk, hd, 1.f, Vs);
    __syncthreads();
    {
      const int ti = (wave >> 1) * 32, tj = (wave & 1) * 32;
      const f32x16 acc = sab_mm32_kk<BF3>(Qs + ti * P, P, Ks + tj * P, P, HDP, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) Ss[(ti + SAB_ROW(r, lane)) * 68 + tj + (lane & 31)] = acc[r];
    }
    __syncthreads();
    for (int r = wave; r < 64; r += 4) {
      const int qi = q0 + r, kj = kb * 64 + lane;
      const bool ok = qi < p.Lq && kj < p.Lk && (!p.causal || kj <= qi);
      const float s = ok ? Ss[r * 68 + lane] : -INFINITY;
      const float mx = sf_wave_max(s);
      const float mo = rm[r], mn = fmaxf(mo, mx);
      const float e = (ok && mn > -INFINITY) ? expf(s - mn) : 0.f;
      const float sum = sf_wave_sum(e);
      Ss[r * 68 + lane] = ok ? e * sab_drop(p, b, h, qi, kj) : 0.f;
      if (lane == 0) {
        const float sc = (mo == -INFINITY) ? 0.f : expf(mo - mn);
        rl[r] = rl[r] * sc + sum;
        rm[r] = mn;
        rs[r] = sc;
      }
    }
    __syncthreads();
    if (at < 2 * CT) {
      const f32x16 a = sab_mm32<BF3>([&](int i, int kk) { return Ss[(ati + i) * 68 + kk]; }, [&](int kk, int j) { return Vs[kk * P + atj + j]; }, 64,
                                lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[r] = oacc[r] * rs[ati + SAB_ROW(r, lane)] + a[r];
    }
  }
  __syncthreads();
  if (at < 2 * CT) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = ati + SAB_ROW(r, lane), qi = q0 + qr, c = atj + (lane & 31);
      if (qi < p.Lq && c < hd) out[(long long)b * p.o_bs + (long long)qi * p.ldo + h * hd + c] = oacc[r] / rl[qr];
    }
  }
  if (tid < 64 && q0 + tid < p.Lq) p.lse[((long long)b * p.H + h) * p.Lq + q0 + tid] = rm[tid] + logf(rl[tid]);
}

static uint32_t sab_site_seed(unsigned long long seed) {
  return sf_mix32((uint32_t)seed ^ sf_mix32((uint32_t)(seed >> 32) + 0x9e3779b9u));
}

extern "C" {

// Training forward with dropout on the attention weights (nn.Dropout inside steve_transformer.py's MultiHeadAttention, :46-48):
// out as sf_slate_attention_strided_f32, lse [B][H][Lq] (kept for sf_slate_attention_train_bwd_f32).
int sf_slate_attention_train_fwd_f32(const float* q, const float* k, const float* v, float* out, float* lse, int ldq, int ldk, int ldv,
                                     int ldo, long long q_bs, long long k_bs, long long v_bs, long long o_bs, int B, int Lq, int Lk,
                                     int num_heads, int head_dim, int causal, float dropout_p, unsigned long long seed, void* stream) {
  SF_REQUIRE(q && k && v && out && lse, "null pointer");
  SF_REQUIRE(B > 0 && Lq > 0 && Lk > 0 && num_heads > 0, "bad sizes");
  SF_REQUIRE(head_dim >= 2 && head_dim <= 64 && head_dim % 2 == 0, "head_dim must be even and <= 64");
  SF_REQUIRE(!causal || Lq == Lk, "causal attention needs Lq == Lk");
  SF_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p in [0, 1)");
  SabArgs a;
  memset(&a, 0, sizeof(a));
  a.q = q; a.k = k; a.v = v; a.lse = lse;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs;
  a.Lq = Lq; a.Lk = Lk; a.H = num_heads; a.causal = causal; a.scale = 1.f / sqrtf((float)head_dim);
  a.drop_seed = sab_site_seed(seed); a.drop_thresh = (uint32_t)((double)dropout_p * 16777216.0); a.drop_scale = 1.f / (1.f - dropout_p);
  if (causal) {   // the long causal self-attention runs on the inference flash kernel (scores in registers) with the mask added
    const int rc = sf_slate_flash_train_ex(q, k, v, out, lse, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, B, Lq, num_heads, head_dim,
                                           a.drop_seed, a.drop_thresh, a.drop_scale, (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  const int hdp = head_dim <= 32 ? 32 : 64;
  const size_t lds = ((size_t)3 * 64 * (hdp + 4) + 64 * 68 + 192) * sizeof(float);
  const dim3 g((Lq + 63) / 64, num_heads, B);
  const bool bf3 = sf_get_precision() >= 1;
  if (hdp == 32) {
    if (bf3) hipLaunchKernelGGL((slate_attn_fwd_train_kernel<32, true>), g, dim3(256), lds, (hipStream_t)stream, a, out, head_dim);
    else hipLaunchKernelGGL((slate_attn_fwd_train_kernel<32, false>), g, dim3(256), lds, (hipStream_t)stream, a, out, head_dim);
  } else {
    SF_TRY(sf_ensure_dyn_lds((const void*)slate_attn_fwd_train_kernel<64, true>, (size_t)160 * 1024));
    SF_TRY(sf_ensure_dyn_lds((const void*)slate_attn_fwd_train_kernel<64, false>, (size_t)160 * 1024));
    if (bf3) hipLaunchKernelGGL((slate_attn_fwd_train_kernel<64, true>), g, dim3(256), lds, (hipStream_t)stream, a, out, head_dim);
    else hipLaunchKernelGGL((slate_attn_fwd_train_kernel<64, false>), g, dim3(256), lds, (hipStream_t)stream, a, out, head_dim);
  }
  SF_CHECK_LAUNCH();
  return 0;
}

size_t sf_slate_attention_bwd_workspace_bytes(int B, int Lq, int num_heads) {
  return (size_t)2 * B * num_heads * Lq * sizeof(float) + 256;
}

// Adjoint of sf_slate_attention_strided_f32: dq / dk / dv have the layouts of q / k / v, d_out and out the layout of out.
// dq must not alias q (it is zeroed here and accumulated with float atomics: its low bits depend on the arrival order).
int sf_slate_attention_train_bwd_f32(const float* q, const float* k, const float* v, const float* out, const float* d_out,
                                     const float* lse, float* dq, float* dk, float* dv, int ldq, int ldk, int ldv, int ldo,
                                     long long q_bs, long long k_bs, long long v_bs, long long o_bs, int B, int Lq, int Lk,
                                     int num_heads, int head_dim, int causal, float dropout_p, unsigned long long seed, void* ws,
                                     size_t ws_bytes, void* stream) {
  SF_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p in [0, 1)");
  SF_REQUIRE(q && k && v && out && d_out && dq && dk && dv && ws, "null pointer");
  SF_REQUIRE(B > 0 && Lq > 0 && Lk > 0 && num_heads > 0, "bad sizes");
  SF_REQUIRE(head_dim >= 4 && head_dim <= 64 && head_dim % 4 == 0, "head_dim must be a multiple of 4, at most 64");
  SF_REQUIRE(ldq % 4 == 0 && ldo % 4 == 0, "rows must be 16-byte aligned");
  SF_REQUIRE(!causal || Lq == Lk, "causal attention needs Lq == Lk");
  SF_REQUIRE(ws_bytes >= sf_slate_attention_bwd_workspace_bytes(B, Lq, num_heads), "workspace too small");
  SF_REQUIRE(q_bs >= (long long)(Lq - 1) * ldq + num_heads * head_dim, "dq is cleared over whole batches: q_bs too small");
  hipStream_t st = (hipStream_t)stream;
  SabArgs a;
  memset(&a, 0, sizeof(a));
  a.q = q; a.k = k; a.v = v; a.o = out; a.dout = d_out; a.dq = dq; a.dk = dk; a.dv = dv;
  float* wsf = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  a.have_lse = lse != nullptr;
  a.lse = lse ? const_cast<float*>(lse) : wsf;
  a.dsum = wsf + (size_t)B * num_heads * Lq;
  a.drop_seed = sab_site_seed(seed); a.drop_thresh = (uint32_t)((double)dropout_p * 16777216.0); a.drop_scale = 1.f / (1.f - dropout_p);
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs;
  a.Lq = Lq; a.Lk = Lk; a.H = num_heads; a.causal = causal; a.scale = 1.f / sqrtf((float)head_dim);
  // clear the head columns of dq: one memset when dq is a dense [B, Lq, H*hd] tensor, else row by row per sequence (dq may be a
  // column slice of a wider packed tensor)
  if (ldq == num_heads * head_dim && q_bs == (long long)Lq * ldq) {
    hipError_t e = hipMemsetAsync(dq, 0, (size_t)B * Lq * ldq * sizeof(float), st);
    if (e != hipSuccess) return sf_set_err((int)e, hipGetErrorString(e), __FILE__, __LINE__);
  } else if (q_bs == (long long)Lq * ldq) {   // column slice of a packed [B, Lq, ldq] tensor: all B*Lq rows in one call
    hipError_t e = hipMemset2DAsync(dq, (size_t)ldq * sizeof(float), 0, (size_t)num_heads * head_dim * sizeof(float), (size_t)B * Lq, st);
    if (e != hipSuccess) return sf_set_err((int)e, hipGetErrorString(e), __FILE__, __LINE__);
  } else {
    for (int b = 0; b < B; ++b) {
      hipError_t e = hipMemset2DAsync(dq + (long long)b * q_bs, (size_t)ldq * sizeof(float), 0, (size_t)num_heads * head_dim * sizeof(float),
                                      (size_t)Lq, st);
      if (e != hipSuccess) return sf_set_err((int)e, hipGetErrorString(e), __FILE__, __LINE__);
    }
  }
  const int hdp = head_dim <= 32 ? 32 : 64;
  const size_t lds1 = ((size_t)2 * 64 * (hdp + 4) + 64 * 68 + 128) * sizeof(float);
  const size_t lds2 = ((size_t)4 * 64 * (hdp + 4) + 2 * 64 * 68 + 128) * sizeof(float);
  const dim3 g1((Lq + 63) / 64, num_heads, B), g2((Lk + 63) / 64, num_heads, B);
#define SAB_GO(HDP, BF)                                                                                                          \
  {                                                                                                                              \
    SF_TRY(sf_ensure_dyn_lds((const void*)slate_attn_bwd_kernel<HDP, BF>, (size_t)160 * 1024));                                    \
    hipLaunchKernelGGL((slate_attn_stats_kernel<HDP, BF>), g1, dim3(256), lds1, st, a, head_dim);                                \
    hipLaunchKernelGGL((slate_attn_bwd_kernel<HDP, BF>), g2, dim3(256), lds2, st, a, head_dim);                                  \
  }
  const bool bf3 = sf_get_precision() >= 1;
  constexpr bool bwd48 = true;
  if (bf3 && head_dim > 32 && head_dim <= 48 && bwd48 && ldk % 4 == 0 && ldv % 4 == 0) {
    constexpr size_t lds48 = ((size_t)3 * 64 * 52 + 2 * 64 * 68 + 128) * sizeof(float);
    static const bool v1 = getenv("SF_DBG") && strstr(getenv("SF_DBG"), "sab1");   // (probes: the packed-word kernel this one replaced)
    hipLaunchKernelGGL((slate_attn_stats_kernel<64, true>), g1, dim3(256), lds1, st, a, head_dim);
    if (v1) {
      SF_TRY(sf_ensure_dyn_lds((const void*)slate_attn_bwd48_kernel, (size_t)(lds48)));
      hipLaunchKernelGGL(slate_attn_bwd48_kernel, g2, dim3(256), lds48, st, a, head_dim);
    } else {
      SF_TRY(sf_ensure_dyn_lds((const void*)slate_attn_bwd48v2_kernel, (size_t)V2_LDS));
      hipLaunchKernelGGL(slate_attn_bwd48v2_kernel, g2, dim3(256), (size_t)V2_LDS, st, a, head_dim);
    }
  } else if (hdp == 32) {
    if (bf3) SAB_GO(32, true) else SAB_GO(32, false)
  } else {
    if (bf3) SAB_GO(64, true) else SAB_GO(64, false)
  }
#undef SAB_GO
  SF_CHECK_LAUNCH();
  return 0;
}

int sf_slate_attention_bwd_f32(const float* q, const float* k, const float* v, const float* out, const float* d_out, float* dq,
                               float* dk, float* dv, int ldq, int ldk, int ldv, int ldo, long long q_bs, long long k_bs,
                               long long v_bs, long long o_bs, int B, int Lq, int Lk, int num_heads, int head_dim, int causal,
                               void* ws, size_t ws_bytes, void* stream) {
  return sf_slate_attention_train_bwd_f32(q, k, v, out, d_out, nullptr, dq, dk, dv, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, B, Lq,
                                          Lk, num_heads, head_dim, causal, 0.f, 0ULL, ws, ws_bytes, stream);
}

}  // extern "C"
