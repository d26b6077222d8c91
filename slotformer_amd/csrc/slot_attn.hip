// Slot Attention iteration for gfx950 (savi.py:76-100 / steve.py:44-71).
//
// Kernel 1 (HBM-bound): one pass over K and V.  For every pixel: logits over the N slots,
// softmax over slots, +eps, and accumulation of  num[n,:] += a_n * v  and  den[n] += a_n.
// The reference's renormalisation over pixels (attn / attn.sum(dim=1)) followed by the
// weighted mean is algebraically  updates[n,:] = num[n,:] / den[n],  so a single pass with
// per-workgroup partials suffices.  One wave owns one pixel per step: the 64 lanes hold the
// D channels (D/64 consecutive floats per lane, fully coalesced 512/768 B rows), the slot
// queries live in registers, the per-pixel softmax is wave-uniform.
//
// Kernel 2 (fused slot update): reduce the partials, GRUCell (gate order r,z,n), residual
// LayerNorm-MLP.  One workgroup per (batch, slot) row, one thread per output feature, weights
// pre-transposed to [in, out] so the k-loop is a stream of independent coalesced loads.
#include "sf_internal.h"
#include <stdlib.h>

#define SA_NMAX 8
#define SA_DMAX 256
#define SA_HMAX 512


template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// full 64-lane sum, result in every lane
__device__ __forceinline__ float wave_allsum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

template <int VPT>
__global__ __launch_bounds__(256) void sa_attn_partial_kernel(
    const float* __restrict__ k, const float* __restrict__ v, int ld, long long batch_stride,
    const float* __restrict__ q, float scale, float eps, float* __restrict__ part_num,
    float* __restrict__ part_den, float* __restrict__ attn_out, long long attn_bs, int HW, int N,
    int P) {
  constexpr int D = 64 * VPT;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pix_per_wg = HW / P, pix_per_wave = pix_per_wg / 4;
  const int pix0 = chunk * pix_per_wg + wave * pix_per_wave;

  float qr[SA_NMAX][VPT], num[SA_NMAX][VPT], den[SA_NMAX];
#pragma unroll
  for (int n = 0; n < SA_NMAX; ++n) {
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      qr[n][j] = (n < N) ? q[((long long)b * N + n) * D + lane * VPT + j] * scale : 0.f;
      num[n][j] = 0.f;
    }
    den[n] = 0.f;
  }
  const float* kb = k + (long long)b * batch_stride + lane * VPT;
  const float* vb = v + (long long)b * batch_stride + lane * VPT;

  __shared__ float s_attn[4][SA_NMAX][64];  // attn staging for coalesced mask rows
  __shared__ float s_red[4][SA_NMAX][D + 1];

  constexpr int U = 4;
  for (int p0 = 0; p0 < pix_per_wave; p0 += U) {
    float kx[U][VPT], vx[U][VPT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long off = (long long)(pix0 + p0 + u) * ld;
#pragma unroll
      for (int j = 0; j < VPT; ++j) {
        kx[u][j] = kb[off + j];
        vx[u][j] = vb[off + j];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float s[SA_NMAX];
#pragma unroll
      for (int n = 0; n < SA_NMAX; ++n) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < VPT; ++j) acc = fmaf(kx[u][j], qr[n][j], acc);
        s[n] = acc;
      }
#pragma unroll
      for (int n = 0; n < SA_NMAX; ++n)
        if (n < N) s[n] = wave_allsum(s[n]);
      float mx = s[0];
#pragma unroll
      for (int n = 1; n < SA_NMAX; ++n)
        if (n < N) mx = fmaxf(mx, s[n]);
      float sum = 0.f;
#pragma unroll
      for (int n = 0; n < SA_NMAX; ++n) {
        s[n] = (n < N) ? expf(s[n] - mx) : 0.f;
        sum += s[n];
      }
      const float inv = 1.0f / sum;
#pragma unroll
      for (int n = 0; n < SA_NMAX; ++n) {
        if (n < N) {
          const float a0 = s[n] * inv;
          if (attn_out) {
            if (lane == ((p0 + u) & 63)) s_attn[wave][n][(p0 + u) & 63] = a0;
          }
          const float a = a0 + eps;
          den[n] += a;
#pragma unroll
          for (int j = 0; j < VPT; ++j) num[n][j] = fmaf(a, vx[u][j], num[n][j]);
        }
      }
    }
    if (attn_out && (((p0 + U) & 63) == 0 || p0 + U >= pix_per_wave)) {
      // flush up to 64 pixels of attention: rows of [B, N, HW]
      __builtin_amdgcn_wave_barrier();
      const int base = (p0 + U - 1) & ~63;
      const int cnt = p0 + U - base;
      for (int n = 0; n < N; ++n)
        if (lane < cnt) attn_out[(long long)b * attn_bs + (long long)n * HW + pix0 + base + lane] = s_attn[wave][n][lane];
    }
  }

  // cross-wave reduction, then one partial per workgroup
#pragma unroll
  for (int n = 0; n < SA_NMAX; ++n) {
#pragma unroll
    for (int j = 0; j < VPT; ++j) s_red[wave][n][lane * VPT + j] = num[n][j];
    if (lane == 0) s_red[wave][n][D] = den[n];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < N * (D + 1); idx += 256) {
    const int n = idx / (D + 1), d = idx - n * (D + 1);
    const float t = s_red[0][n][d] + s_red[1][n][d] + s_red[2][n][d] + s_red[3][n][d];
    if (d < D)
      part_num[(((long long)b * P + chunk) * N + n) * D + d] = t;
    else
      part_den[((long long)b * P + chunk) * N + n] = t;
  }
}

// -----------------------------------------------------------------------------------------
// v2 of the iteration kernel (used when HW % 256 == 0): no cross-lane traffic in the hot loop.
//   phase 1: ONE LANE PER PIXEL -- the lane walks its own K row (float4 loads), the N scaled
//            queries are broadcast from LDS; logits, softmax over slots and +eps are lane-local.
//   phase 2: the sums over pixels  num[n,d] = sum_p a[p,n] v[p,d]  are a [D x 64] . [64 x N]
//            product per wave: v_mfma_f32_16x16x4_f32 with V loaded straight into A-operand
//            layout (lane i owns the DB = D/16 consecutive channels DB*i .. DB*i+DB-1, so a
//            wave reads whole 512/768-B rows) and the attention tile as B operand from LDS.
typedef float f32x4v __attribute__((ext_vector_type(4)));

// KT = float: f32 K/V rows (the product path).  KT = __bf16 (sf_slot_attn_iter_bf16): K/V STORED as bf16 -- half the HBM
// bytes of this HBM-bound kernel -- widened to f32 when loaded; all arithmetic stays f32.
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
template <typename KT>
__device__ __forceinline__ f32x4v sa_load4(const KT* p) {
  if constexpr (sizeof(KT) == 4) {
    return *(const f32x4v*)p;
  } else {
    return __builtin_convertvector(*(const bf16x4v*)p, f32x4v);
  }
}

template <int D, typename KT = float>
__global__ __launch_bounds__(256) void sa_attn_mfma_kernel(
    const KT* __restrict__ k, const KT* __restrict__ v, int ld, long long batch_stride,
    const float* __restrict__ q, float scale, float eps, float* __restrict__ part_num,
    float* __restrict__ part_den, float* __restrict__ attn_out, long long attn_bs, int HW, int N, int P) {
  constexpr int DB = D / 16;  // channel blocks of the MFMA (each 16 channels wide, strided by DB)
  constexpr int NS = SA_NMAX;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pix0 = chunk * 256 + wave * 64;  // 64 pixels per wave
  __shared__ __attribute__((aligned(16))) float s_q[NS][D];
  __shared__ float s_a[4][64][NS + 1];
  __shared__ float s_red[4][NS][D + 1];
  for (int idx = threadIdx.x; idx < NS * D; idx += 256) {
    const int n = idx / D, d = idx - n * D;
    s_q[n][d] = n < N ? q[((long long)b * N + n) * D + d] * scale : 0.f;
  }
  __syncthreads();

  // ---- phase 1: lane = pixel.  The K rows reach their lanes through an LDS slab: a load instruction with one pixel row
  //      per lane touches 64 separate cache lines (rows are 1 KB apart), so the wave instead loads a 32-channel slab of
  //      its 64 rows with 8 lanes per row (8 x 128 B contiguous per instruction), parks it in LDS and every lane reads
  //      ITS row back; the next slab is requested before the current one is consumed. ----
  __shared__ __attribute__((aligned(16))) float s_k[4][64][36];
  const KT* kslab = k + (long long)b * batch_stride + (long long)(pix0 + (lane >> 3)) * ld + 4 * (lane & 7);
  float s[NS];
#pragma unroll
  for (int n = 0; n < NS; ++n) s[n] = 0.f;
  // (round 4) ALL slabs of the wave's 64 rows are requested up front -- up to four (D <= 128: the whole tile, 128 registers; two waves
  // per SIMD leave 256) -- instead of one slab ahead: the four dependent round trips of phase 1 (~2 us each from HBM) were most of a
  // workgroup's ~19 us, and a workgroup's latency IS the launch time (one round of workgroups on the whole chip, two on a 128-CU mask)
  constexpr int NSL = D / 32, PF = D <= 128 ? NSL : 1;   // (wider slots: one slab ahead as before -- the tile would not fit the registers)
  f32x4v nx[PF][8];
#pragma unroll
  for (int c = 0; c < PF; ++c)
#pragma unroll
    for (int u = 0; u < 8; ++u) nx[c][u] = sa_load4<KT>(kslab + (long long)(8 * u) * ld + 32 * c);
#pragma unroll(D <= 128 ? NSL : 1)
  for (int sl = 0; sl < NSL; ++sl) {
    const int d0 = 32 * sl;
#pragma unroll
    for (int u = 0; u < 8; ++u) *(f32x4v*)&s_k[wave][(lane >> 3) + 8 * u][4 * (lane & 7)] = nx[sl % PF][u];
    if (sl + PF < NSL) {
#pragma unroll
      for (int u = 0; u < 8; ++u) nx[sl % PF][u] = sa_load4<KT>(kslab + (long long)(8 * u) * ld + d0 + 32 * PF);
    }
    __builtin_amdgcn_wave_barrier();
    f32x4v kx[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) kx[u] = *(const f32x4v*)&s_k[wave][lane][4 * u];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int n = 0; n < NS; ++n) {
        const f32x4v qq = *(const f32x4v*)(&s_q[n][d0 + 4 * u]);
        s[n] = fmaf(kx[u][0], qq[0], s[n]);
        s[n] = fmaf(kx[u][1], qq[1], s[n]);
        s[n] = fmaf(kx[u][2], qq[2], s[n]);
        s[n] = fmaf(kx[u][3], qq[3], s[n]);
      }
    }
  }
  float mx = s[0];
#pragma unroll
  for (int n = 1; n < NS; ++n)
    if (n < N) mx = fmaxf(mx, s[n]);
  float sum = 0.f;
#pragma unroll
  for (int n = 0; n < NS; ++n) {
    s[n] = n < N ? expf(s[n] - mx) : 0.f;
    sum += s[n];
  }
  const float inv = 1.0f / sum;
  float den[NS];
#pragma unroll
  for (int n = 0; n < NS; ++n) {
    const float a0 = s[n] * inv;
    if (attn_out && n < N) attn_out[(long long)b * attn_bs + (long long)n * HW + pix0 + lane] = a0;
    const float a = n < N ? a0 + eps : 0.f;
    den[n] = a;
    s_a[wave][lane][n] = a;
  }
  __builtin_amdgcn_wave_barrier();

  // ---- phase 2: num^T[d, n] += V^T . A on the f32 MFMA ------------------------------------------
  typedef float f32x4a __attribute__((ext_vector_type(4)));
  f32x4a acc[DB];
#pragma unroll
  for (int j = 0; j < DB; ++j) acc[j] = f32x4a{0.f, 0.f, 0.f, 0.f};
  const int li = lane & 15, lk = lane >> 4;  // A: row i = li (channels DB*li..), k = lk (pixel in group of 4)
  const KT* vbase = v + (long long)b * batch_stride + (long long)(pix0 + lk) * ld + DB * li;
  // (round 4) the 16 k-steps fully unrolled: every V row of the wave is requested before the first MFMA needs one (DB / 4 16-byte loads
  // per k-step: 32 .. 64 registers in flight), instead of four k-steps at a time
  // (wider slots: four k-steps at a time as before -- the whole tile would not fit the registers)
  constexpr int PFV = D <= 128 ? 16 : 4;
#pragma unroll
  for (int k0 = 0; k0 < 16; k0 += PFV) {
    f32x4v vq[PFV][DB / 4];
#pragma unroll
    for (int ks = 0; ks < PFV; ++ks)
#pragma unroll
      for (int j4 = 0; j4 < DB / 4; ++j4) vq[ks][j4] = sa_load4<KT>(vbase + (long long)(4 * (k0 + ks)) * ld + 4 * j4);
#pragma unroll
    for (int ks = 0; ks < PFV; ++ks) {
      const float bop = li < NS ? s_a[wave][4 * (k0 + ks) + lk][li] : 0.f;  // B[k = pixel][j = slot]
#pragma unroll
      for (int j = 0; j < DB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vq[ks][j >> 2][j & 3], bop, acc[j], 0, 0, 0);
    }
  }
  // acc[j][r]: channel d = DB * (4*lk + r) + j, slot = li
#pragma unroll
  for (int n = 0; n < NS; ++n) den[n] = sf_wave_sum(den[n]);
  if (li < NS) {
#pragma unroll
    for (int j = 0; j < DB; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_red[wave][li][DB * (4 * lk + r) + j] = acc[j][r];
  }
  if (lane == 0) {
#pragma unroll
    for (int n = 0; n < NS; ++n) s_red[wave][n][D] = den[n];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < N * (D + 1); idx += 256) {
    const int n = idx / (D + 1), d = idx - n * (D + 1);
    const float t = s_red[0][n][d] + s_red[1][n][d] + s_red[2][n][d] + s_red[3][n][d];
    if (d < D)
      part_num[(((long long)b * P + chunk) * N + n) * D + d] = t;
    else
      part_den[((long long)b * P + chunk) * N + n] = t;
  }
}

// -----------------------------------------------------------------------------------------
// Fused slot update.  One workgroup per (batch, slot) row, ONE THREAD PER OUTPUT FEATURE: the
// weights are stored transposed ([in, out]) so consecutive threads read consecutive addresses
// and every load of the k-loop is independent (deep unroll keeps them in flight); the input
// vector is broadcast from LDS.  No cross-lane reductions at all.
template <int UNR>
__device__ __forceinline__ float col_dot(const float* __restrict__ wt, int ldw, int col, const float* x, int K) {
  float acc = 0.f;
  int k = 0;
  for (; k + UNR <= K; k += UNR) {
    float w[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) w[u] = wt[(long long)(k + u) * ldw + col];
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc = fmaf(w[u], x[k + u], acc);
  }
  for (; k < K; ++k) acc = fmaf(wt[(long long)k * ldw + col], x[k], acc);
  return acc;
}

// two dot products sharing the loop: 2*UNR independent coalesced loads in flight per thread
template <int UNR>
__device__ __forceinline__ void col_dot2(const float* __restrict__ wa, const float* __restrict__ wb, int ldw, int col,
                                         const float* xa, const float* xb, int K, float& ra, float& rb) {
  float a = 0.f, b = 0.f;
  int k = 0;
  for (; k + UNR <= K; k += UNR) {
    float va[UNR], vb[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      va[u] = wa[(long long)(k + u) * ldw + col];
      vb[u] = wb[(long long)(k + u) * ldw + col];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      a = fmaf(va[u], xa[k + u], a);
      b = fmaf(vb[u], xb[k + u], b);
    }
  }
  for (; k < K; ++k) {
    a = fmaf(wa[(long long)k * ldw + col], xa[k], a);
    b = fmaf(wb[(long long)k * ldw + col], xb[k], b);
  }
  ra = a;
  rb = b;
}

// Slot update (savi.py:95-100) for SU_R rows (= slots) per workgroup: updates = sum(num) / sum(den); GRUCell; slots + MLP(LN).
// One workgroup per slot (the first version) made every workgroup stream all 655 KB of GRU / MLP weights for a single
// row -- 147 MB through the CUs for 224 rows, 27 us.  Here a thread still owns one output feature, but carries it for
// 8 rows at once: every weight is loaded once per workgroup and used 8 times, the row vectors sit in LDS k-major
// ([feature][row], two 16-byte broadcasts per k).  The accumulation order over k is the same as before, per row.
// Optionally the kernel also emits q = LN_q(slots_out) Wq^T for the NEXT Slot-Attention iteration (savi.py:79, the
// LN-fused GEMM launch it replaces) and a second copy of slots_out into a strided [B,T,N,D] tensor (the encode loop's
// post_slots, replacing a copy launch).
constexpr int SU_R = 8;

template <int NG>   // NG = number of accumulator groups of the caller (1: one matrix, 2: two matrices sharing the loop)
struct SuAcc {
  float a[NG][SU_R];
};

// acc[r] += w * x[k][r] over k; x k-major in LDS ([K][SU_R]); w column `col` of a [K][ldw] matrix (coalesced over threads)
template <int UNR>
__device__ __forceinline__ void su_col_dot(const float* __restrict__ w, int ldw, int col, const float* x, int K, float (&acc)[SU_R]) {
#pragma unroll
  for (int r = 0; r < SU_R; ++r) acc[r] = 0.f;
  for (int k = 0; k < K; k += UNR) {
    float wv[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) wv[u] = w[(long long)(k + u) * ldw + col];
    // every weight request of this batch is in flight before the first multiply: without the fence the scheduler (short of
    // registers for the hoisted LDS reads) issued ONE load per group of multiplies and waited for it -- 128 round trips
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const f32x4v x0 = *(const f32x4v*)(x + (k + u) * SU_R), x1 = *(const f32x4v*)(x + (k + u) * SU_R + 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[r] = fmaf(wv[u], x0[r], acc[r]);
        acc[r + 4] = fmaf(wv[u], x1[r], acc[r + 4]);
      }
    }
  }
}

// LayerNorm statistics of the SU_R rows held k-major in `x` ([D][SU_R]) -> stat[r] = {mean, rstd}; wave w handles rows
// w, w + nwaves, ...  (same summation order per row as the single-row kernel had: lane-strided partial sums, wave reduce)
__device__ __forceinline__ void su_ln_stats(const float* x, float* stat, int D, float eps, int wave, int lane, int nwaves) {
  for (int r = wave; r < SU_R; r += nwaves) {
    float sm = 0.f;
    for (int d = lane; d < D; d += 64) sm += x[d * SU_R + r];
    const float mean = sf_wave_sum(sm) / (float)D;
    float vv = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float c = x[d * SU_R + r] - mean;
      vv += c * c;
    }
    const float rstd = 1.0f / sqrtf(sf_wave_sum(vv) / (float)D + eps);
    if (lane == 0) {
      stat[2 * r] = mean;
      stat[2 * r + 1] = rstd;
    }
  }
}

struct SuExtra {
  float* out2;              // optional second destination of slots_out: row (b, n) at out2 + b * out2_bs + n * D
  long long out2_bs;
  const float* q_ln_g;      // optional q projection of the result (all NULL: off)
  const float* q_ln_b;
  const float* q_w;         // [D][D] TRANSPOSED project_q weight ([in][out])
  float* q_out;             // [R][D]
};

// number of K slices a [K x Nout] product is split into across the workgroup's threads: as many as there are spare threads,
// while a slice keeps at least 64 k-steps (one batch of weight requests per thread = ONE memory round trip per phase)
__host__ __device__ inline int su_kslices(int nthreads, int Nout, int K) {
  int ks = 1;
  while (2 * ks * Nout <= nthreads && K / (2 * ks) >= 64 && (K % (2 * ks * 64)) == 0) ks *= 2;
  return ks;
}

// part[(slice * Nout + j) * SU_R + r] = sum over the slice's k of w[k][j] * x[k][r]
__device__ __forceinline__ void su_gemm(const float* __restrict__ w, int Nout, int K, const float* x, float* part, int t, int nt) {
  const int ks = su_kslices(nt, Nout, K);
  if (t < Nout * ks) {
    const int j = t % Nout, kh = t / Nout, kr = K / ks;
    float acc[SU_R];
    su_col_dot<64>(w + (long long)kh * kr * Nout, Nout, j, x + kh * kr * SU_R, kr, acc);
#pragma unroll
    for (int r = 0; r < SU_R; ++r) part[(kh * Nout + j) * SU_R + r] = acc[r];
  }
}
// fixed-order sum of the K slices of output feature j, row r
__device__ __forceinline__ float su_sum(const float* part, int ks, int Nout, int j, int r) {
  float v = part[j * SU_R + r];
  for (int kh = 1; kh < ks; ++kh) v += part[(kh * Nout + j) * SU_R + r];
  return v;
}

template <int NT>
__global__ __launch_bounds__(NT) void sa_slot_update_kernel(
    const float* __restrict__ part_num, const float* __restrict__ part_den, int P,
    const float* __restrict__ slots_prev, const float* __restrict__ w_ih_t,
    const float* __restrict__ w_hh_t, const float* __restrict__ b_ih, const float* __restrict__ b_hh,
    const float* __restrict__ ln_g, const float* __restrict__ ln_b, const float* __restrict__ w1_t,
    const float* __restrict__ b1, const float* __restrict__ w2_t, const float* __restrict__ b2,
    float* __restrict__ slots_out, SuExtra ex, int R, int N, int D, int H, int pmax, float ln_eps) {
  extern __shared__ __attribute__((aligned(16))) float su_lds[];
  float* s_u = su_lds;                    // [D][SU_R]   updates; later the finished rows (input of the q projection)
  float* s_h = s_u + D * SU_R;            // [D][SU_R]   previous slots
  float* s_hn = s_h + D * SU_R;           // [D][SU_R]   GRU output
  float* s_ln = s_hn + D * SU_R;          // [D][SU_R]
  float* s_hid = s_ln + D * SU_R;         // [H][SU_R]
  float* s_stat = s_hid + H * SU_R;       // [SU_R][2]
  float* s_pa = s_stat + 2 * SU_R;        // K-slice partials of the running product ([slices][Nout][SU_R], pmax floats)
  float* s_pb = s_pa + pmax;              // second partial buffer (the GRU has two products)
  const int row0 = blockIdx.x * SU_R;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nt = blockDim.x, nwaves = nt >> 6;

  // updates = sum_p num / sum_p den; previous slots (rows beyond R: zeros).  The per-row denominators first (one lane per
  // partial), then every (row, feature) pair with its P numerator loads in flight at once.
  for (int r = wave; r < SU_R; r += nwaves) {
    const int row = row0 + r;
    float dv = 0.f;
    if (row < R && lane < P) {
      const int b = row / N, n = row - b * N;
      dv = part_den[((long long)b * P + lane) * N + n];
    }
    s_pa[r * 64 + lane] = dv;
  }
  __syncthreads();
  for (int idx = t; idx < SU_R * D; idx += nt) {
    const int r = idx / D, d = idx - r * D, row = row0 + r;
    float u = 0.f, h = 0.f;
    if (row < R) {
      const int b = row / N, n = row - b * N;
      float den = 0.f, a = 0.f;
      const float* pn = part_num + ((long long)b * P * N + n) * D + d;
#pragma unroll 16
      for (int p = 0; p < P; ++p) a += pn[(long long)p * N * D];
      for (int p = 0; p < P; ++p) den += s_pa[r * 64 + p];
      u = a / den;
      h = slots_prev[(long long)row * D + d];
    }
    s_u[d * SU_R + r] = u;
    s_h[d * SU_R + r] = h;
  }
  __syncthreads();
  // GRU gate pre-activations: gi = W_ih u, gh = W_hh h (K slices across the threads)
  su_gemm(w_ih_t, 3 * D, D, s_u, s_pa, t, nt);
  su_gemm(w_hh_t, 3 * D, D, s_h, s_pb, t, nt);
  __syncthreads();
  {
    const int ks = su_kslices(nt, 3 * D, D);
    for (int idx = t; idx < SU_R * D; idx += nt) {
      const int d = idx / SU_R, r = idx - d * SU_R;
      const float gir = su_sum(s_pa, ks, 3 * D, d, r) + b_ih[d], ghr = su_sum(s_pb, ks, 3 * D, d, r) + b_hh[d];
      const float giz = su_sum(s_pa, ks, 3 * D, D + d, r) + b_ih[D + d], ghz = su_sum(s_pb, ks, 3 * D, D + d, r) + b_hh[D + d];
      const float gin = su_sum(s_pa, ks, 3 * D, 2 * D + d, r) + b_ih[2 * D + d], ghn = su_sum(s_pb, ks, 3 * D, 2 * D + d, r) + b_hh[2 * D + d];
      const float rr = sf_sigmoid(gir + ghr);
      const float z = sf_sigmoid(giz + ghz);
      const float nn = tanhf(gin + rr * ghn);
      s_hn[idx] = (1.f - z) * nn + z * s_h[idx];
    }
  }
  __syncthreads();
  su_ln_stats(s_hn, s_stat, D, ln_eps, wave, lane, nwaves);
  __syncthreads();
  for (int idx = t; idx < SU_R * D; idx += nt) {
    const int d = idx / SU_R, r = idx - d * SU_R;
    s_ln[idx] = (s_hn[idx] - s_stat[2 * r]) * s_stat[2 * r + 1] * ln_g[d] + ln_b[d];
  }
  __syncthreads();
  su_gemm(w1_t, H, D, s_ln, s_pa, t, nt);
  __syncthreads();
  {
    const int ks = su_kslices(nt, H, D);
    for (int idx = t; idx < SU_R * H; idx += nt) {
      const int j = idx / SU_R, r = idx - j * SU_R;
      s_hid[idx] = fmaxf(su_sum(s_pa, ks, H, j, r) + b1[j], 0.f);
    }
  }
  __syncthreads();
  su_gemm(w2_t, D, H, s_hid, s_pb, t, nt);
  __syncthreads();
  {
    const int ks = su_kslices(nt, D, H);
    for (int idx = t; idx < SU_R * D; idx += nt) {
      const int r = idx / D, d = idx - r * D, row = row0 + r;    // consecutive threads = consecutive features: coalesced stores
      const float v = s_hn[d * SU_R + r] + su_sum(s_pb, ks, D, d, r) + b2[d];
      s_u[d * SU_R + r] = v;
      if (row < R) {
        slots_out[(long long)row * D + d] = v;
        if (ex.out2) {
          const int b = row / N, n = row - b * N;
          ex.out2[(long long)b * ex.out2_bs + (long long)n * D + d] = v;
        }
      }
    }
  }
  if (ex.q_out == nullptr) return;
  // ---- q = LN_q(slots_out) Wq^T for the next iteration (project_q, savi.py:45-48,79) ----
  __syncthreads();
  su_ln_stats(s_u, s_stat, D, ln_eps, wave, lane, nwaves);
  __syncthreads();
  for (int idx = t; idx < SU_R * D; idx += nt) {
    const int d = idx / SU_R, r = idx - d * SU_R;
    s_ln[idx] = (s_u[idx] - s_stat[2 * r]) * s_stat[2 * r + 1] * ex.q_ln_g[d] + ex.q_ln_b[d];
  }
  __syncthreads();
  su_gemm(ex.q_w, D, D, s_ln, s_pa, t, nt);   // q_w: project_q weight transposed ([in][out])
  __syncthreads();
  {
    const int ks = su_kslices(nt, D, D);
    for (int idx = t; idx < SU_R * D; idx += nt) {
      const int r = idx / D, d = idx - r * D;
      if (row0 + r < R) ex.q_out[(long long)(row0 + r) * D + d] = su_sum(s_pa, ks, D, d, r);
    }
  }
}

// Slot prologue of one encoder time step in ONE launch (the per-frame chain of StoSAVi.encode before the Slot-Attention
// iterations, savi.py:393-402, for the CLEVRER configuration):
//   lat    = t == 0 ? init_latents : ResidualMLPPredictor(prev_slots)       (predictor.py:65-73)
//   kdist  = kernel_dist_layer(lat)   (single Linear, savi.py:190-200)      -> also written to kernel_dist[:, t]
//   slots  = mu + noise * exp(0.5 logvar)   (noise NULL: mu)                (savi.py:355-365)
//   q      = project_q(slots) = LN_q(slots) Wq^T                            (savi.py:45-48,79)
// It replaces LayerNorm + two predictor GEMMs + the kernel-distribution GEMM + the sampling kernel + a copy + the q GEMM (7
// launches of 5-7 us on 224 rows).  Same machinery as the slot-update kernel: SU_R rows per workgroup, row vectors k-major in
// LDS, a thread owns one output feature of a K slice, weights TRANSPOSED ([in][out]).
struct SpArgs {
  const float* prev;        // [R][D] previous slots, or NULL: lat = init_latents[n]
  const float* init;        // [N][D]
  const float *pm_ln_g, *pm_ln_b, *pm_w0_t, *pm_b0, *pm_w2_t, *pm_b2;   // w0_t [D][2D], w2_t [2D][D]
  int norm_first;
  const float *kd_w_t, *kd_b;   // [D][2D]
  const float* noise;       // row (b, n) at noise + b * noise_bs + n * D, or NULL
  long long noise_bs;
  float* kdist_out;         // row (b, n) at kdist_out + b * kdist_bs + n * 2D, or NULL
  long long kdist_bs;
  const float *q_ln_g, *q_ln_b, *q_w_t;
  float* slots_out;         // [R][D]
  float* q_out;             // [R][D]
};

__global__ __launch_bounds__(768) void sa_slot_prologue_kernel(SpArgs a, int R, int N, int D, int pmax, float ln_eps) {
  extern __shared__ __attribute__((aligned(16))) float su_lds[];
  const int D2 = 2 * D;
  float* s_x = su_lds;                    // [D][SU_R]   predictor input, later the sampled slots
  float* s_ln = s_x + D * SU_R;           // [D][SU_R]
  float* s_lat = s_ln + D * SU_R;         // [D][SU_R]
  float* s_hid = s_lat + D * SU_R;        // [2D][SU_R]  predictor hidden, later the kernel distribution
  float* s_stat = s_hid + D2 * SU_R;      // [SU_R][2]
  float* s_pa = s_stat + 2 * SU_R;        // K-slice partials (pmax floats)
  const int row0 = blockIdx.x * SU_R;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nt = blockDim.x, nwaves = nt >> 6;
  for (int idx = t; idx < SU_R * D; idx += nt) {
    const int r = idx / D, d = idx - r * D, row = row0 + r;
    float v = 0.f;
    if (row < R) v = a.prev ? a.prev[(long long)row * D + d] : a.init[(long long)(row % N) * D + d];
    s_x[d * SU_R + r] = v;
  }
  __syncthreads();
  if (a.prev) {
    // ResidualMLPPredictor: x = LN(x); out = W2 relu(W0 x + b0) + b2 + (norm_first ? LN(x) : x)
    su_ln_stats(s_x, s_stat, D, ln_eps, wave, lane, nwaves);
    __syncthreads();
    for (int idx = t; idx < SU_R * D; idx += nt) {
      const int d = idx / SU_R, r = idx - d * SU_R;
      s_ln[idx] = (s_x[idx] - s_stat[2 * r]) * s_stat[2 * r + 1] * a.pm_ln_g[d] + a.pm_ln_b[d];
    }
    __syncthreads();
    su_gemm(a.pm_w0_t, D2, D, s_ln, s_pa, t, nt);
    __syncthreads();
    {
      const int ks = su_kslices(nt, D2, D);
      for (int idx = t; idx < SU_R * D2; idx += nt) {
        const int j = idx / SU_R, r = idx - j * SU_R;
        s_hid[idx] = fmaxf(su_sum(s_pa, ks, D2, j, r) + a.pm_b0[j], 0.f);
      }
    }
    __syncthreads();
    su_gemm(a.pm_w2_t, D, D2, s_hid, s_pa, t, nt);
    __syncthreads();
    {
      const int ks = su_kslices(nt, D, D2);
      for (int idx = t; idx < SU_R * D; idx += nt) {
        const int d = idx / SU_R, r = idx - d * SU_R;
        s_lat[idx] = su_sum(s_pa, ks, D, d, r) + a.pm_b2[d] + (a.norm_first ? s_ln[idx] : s_x[idx]);
      }
    }
  } else {
    for (int idx = t; idx < SU_R * D; idx += nt) s_lat[idx] = s_x[idx];
  }
  __syncthreads();
  // kernel distribution + sampling
  su_gemm(a.kd_w_t, D2, D, s_lat, s_pa, t, nt);
  __syncthreads();
  {
    const int ks = su_kslices(nt, D2, D);
    for (int idx = t; idx < SU_R * D2; idx += nt) {
      const int r = idx / D2, j = idx - r * D2, row = row0 + r;      // consecutive threads = consecutive features
      const float v = su_sum(s_pa, ks, D2, j, r) + a.kd_b[j];
      s_hid[j * SU_R + r] = v;
      if (a.kdist_out && row < R) a.kdist_out[(long long)(row / N) * a.kdist_bs + (long long)(row % N) * D2 + j] = v;
    }
  }
  __syncthreads();
  for (int idx = t; idx < SU_R * D; idx += nt) {
    const int r = idx / D, d = idx - r * D, row = row0 + r;
    float v = s_hid[d * SU_R + r];
    if (a.noise && row < R)
      v += a.noise[(long long)(row / N) * a.noise_bs + (long long)(row % N) * D + d] * expf(s_hid[(D + d) * SU_R + r] * 0.5f);
    s_x[d * SU_R + r] = v;
    if (row < R) a.slots_out[(long long)row * D + d] = v;
  }
  __syncthreads();
  // q = LN_q(slots) Wq^T
  su_ln_stats(s_x, s_stat, D, ln_eps, wave, lane, nwaves);
  __syncthreads();
  for (int idx = t; idx < SU_R * D; idx += nt) {
    const int d = idx / SU_R, r = idx - d * SU_R;
    s_ln[idx] = (s_x[idx] - s_stat[2 * r]) * s_stat[2 * r + 1] * a.q_ln_g[d] + a.q_ln_b[d];
  }
  __syncthreads();
  su_gemm(a.q_w_t, D, D, s_ln, s_pa, t, nt);
  __syncthreads();
  {
    const int ks = su_kslices(nt, D, D);
    for (int idx = t; idx < SU_R * D; idx += nt) {
      const int r = idx / D, d = idx - r * D;
      if (row0 + r < R) a.q_out[(long long)(row0 + r) * D + d] = su_sum(s_pa, ks, D, d, r);
    }
  }
}

// returns 1 when the fused prologue does not apply (D not a multiple of 64, more than 768 / 2 D ... see the requirements)
int sf_slot_prologue_ex(const float* prev, const float* init, const float* pm_ln_g, const float* pm_ln_b, const float* pm_w0_t,
                        const float* pm_b0, const float* pm_w2_t, const float* pm_b2, int norm_first, const float* kd_w_t,
                        const float* kd_b, const float* noise, long long noise_bs, float* kdist_out, long long kdist_bs,
                        const float* q_ln_g, const float* q_ln_b, const float* q_w_t, float* slots_out, float* q_out, int B,
                        int N, int D, float ln_eps, hipStream_t st) {
  if ((D % 64) != 0 || 2 * D > 768 || !kd_w_t || !kd_b || !q_w_t || !q_ln_g || !q_ln_b || !slots_out || !q_out || !init) return 1;
  if (prev && (!pm_ln_g || !pm_ln_b || !pm_w0_t || !pm_b0 || !pm_w2_t || !pm_b2)) return 1;
  if (B == 0) return 0;
  const int threads = 768, R = B * N, D2 = 2 * D;
  auto pmaxf = [&](int Nout, int K) { return su_kslices(threads, Nout, K) * Nout * SU_R; };
  int pmax = pmaxf(D2, D);
  if (pmaxf(D, D2) > pmax) pmax = pmaxf(D, D2);
  if (pmaxf(D, D) > pmax) pmax = pmaxf(D, D);
  const size_t lds = ((size_t)3 * D * SU_R + (size_t)D2 * SU_R + 2 * SU_R + (size_t)pmax) * sizeof(float);
  SF_TRY(sf_ensure_dyn_lds((const void*)sa_slot_prologue_kernel, lds));
  SpArgs a{prev, init, pm_ln_g, pm_ln_b, pm_w0_t, pm_b0, pm_w2_t, pm_b2, norm_first, kd_w_t, kd_b, noise, noise_bs, kdist_out,
           kdist_bs, q_ln_g, q_ln_b, q_w_t, slots_out, q_out};
  hipLaunchKernelGGL(sa_slot_prologue_kernel, dim3((R + SU_R - 1) / SU_R), dim3(threads), lds, st, a, R, N, D, pmax, ln_eps);
  SF_CHECK_LAUNCH();
  return 0;
}

// -----------------------------------------------------------------------------------------
int sf_sa_pick_partials(int HW) {
  if (HW % 256 == 0) return HW / 256;  // MFMA iteration kernel: 4 waves x 64 pixels per workgroup
  // pixels per workgroup: 4 waves x (multiple of 4) pixels; keep >= 128 pixels per workgroup
  int P = HW / 128;
  while (P > 1 && (HW % P != 0 || (HW / P) % 16 != 0)) --P;
  if (P < 1) P = 1;
  return P;
}

extern "C" {

// number of per-(batch) partial records the iteration kernel emits for HW keys
int sf_slot_attn_num_partials(int HW) { return sf_sa_pick_partials(HW); }

// One Slot-Attention iteration, attention half (savi.py:82-89).
//   k, v     : [B, HW, D] rows with leading dimension ld floats and batch_stride floats per batch
//   q        : [B, N, D] unscaled queries (project_q output); scale = D^-0.5 applied inside
//   part_num : [B, P, N, D], part_den: [B, P, N]  (P = sf_slot_attn_num_partials(HW))
//   attn_out : optional [B, N, HW] softmax-over-slots attention (STEVE seg mask, steve.py:54-55)
int sf_slot_attn_iter_f32(const float* k, const float* v, int ld, long long batch_stride, const float* q,
                          float* part_num, float* part_den, float* attn_out, int B, int HW, int N, int D,
                          float scale, float eps, void* stream) {
  return sf_slot_attn_iter_ex(k, v, ld, batch_stride, q, part_num, part_den, attn_out, (long long)N * HW, B, HW,
                              N, D, scale, eps, (hipStream_t)stream);
}
}  // extern "C"

// bf16-storage variant of the iteration (SURVEY.md 8(b2) `sf_slot_attn_iter_bf16`): k, v point to bf16 rows (ld, batch_stride in
// ELEMENTS); q, the partial records and attn_out are f32 as in sf_slot_attn_iter_f32.  Needs HW % 256 == 0.
extern "C" int sf_slot_attn_iter_bf16(const void* k, const void* v, int ld, long long batch_stride, const float* q,
                                      float* part_num, float* part_den, float* attn_out, int B, int HW, int N, int D, float scale,
                                      float eps, void* stream) {
  SF_REQUIRE(k && v && q && part_num && part_den, "null pointer");
  SF_REQUIRE(B >= 0 && HW > 0 && (HW % 256) == 0 && N >= 1 && N <= SA_NMAX, "need 1 <= num_slots <= 8 and HW % 256 == 0");
  SF_REQUIRE(D == 64 || D == 128 || D == 192 || D == 256, "slot_size must be 64/128/192/256");
  SF_REQUIRE(ld >= D && (ld % 4) == 0 && (batch_stride % 4) == 0, "k/v rows must be 8-byte aligned");
  if (B == 0) return 0;
  const int P = sf_sa_pick_partials(HW);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(P, B), block(256);
  const __bf16* kb = (const __bf16*)k;
  const __bf16* vb = (const __bf16*)v;
  const long long abs_ = (long long)N * HW;
  sf_prof_begin(SF_K_SA_ITER, st, 2.0 * (double)B * HW * D * sizeof(__bf16));
#define SA_LAUNCHB(DD)                                                                                                   \
  hipLaunchKernelGGL((sa_attn_mfma_kernel<DD, __bf16>), grid, block, 0, st, kb, vb, ld, batch_stride, q, scale, eps, part_num, \
                     part_den, attn_out, abs_, HW, N, P)
  switch (D / 64) {
    case 1: SA_LAUNCHB(64); break;
    case 2: SA_LAUNCHB(128); break;
    case 3: SA_LAUNCHB(192); break;
    default: SA_LAUNCHB(256); break;
  }
#undef SA_LAUNCHB
  sf_prof_end(SF_K_SA_ITER, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// -----------------------------------------------------------------------------------------
// One-pass form on matrix cores for keys == values (round 4; the folded Slot Attention of engine.hip).  The two-pass kernel fetches 1.8x
// the unique bytes (its second walk over the rows misses L2), and the first one-pass form (above) lost because a 33 KB tile per wave left
// four waves per CU.  Here a wave owns 32 pixels (16.5 KB tile, eight waves per workgroup of 256 pixels: two waves per SIMD as before):
//   * the wave's 32 rows arrive with sixteen 16-byte loads per lane, two WHOLE rows per instruction (perfect coalescing), and are parked in
//     an LDS tile of pitch D + 4 floats (conflict-free 16-byte row reads);
//   * logits: X . Q^T on v_mfma_f32_16x16x4_f32 -- the tile rows are the A operand, the scaled queries (zero beyond N slots) sit in
//     registers as the B operand.  One 16-byte LDS read feeds four MFMAs: MFMA e of a read contracts channels {16 s + 4 g + e}, g = lane >> 4
//     (the contraction runs over all channels, so which instruction takes which channel is free as long as both operands agree);
//   * softmax over slots: the 8 slots of a pixel are lanes 0..7 of a 16-lane row of the accumulator: two DPP all-reduces (max, sum);
//   * weighted sums: A^T . X with the attention tile (through 2 KB of LDS, transposed) as A operand and the SAME LDS tile as B operand,
//     again four MFMAs per 16-byte read (output column j of MFMA (h, e) is channel 64 h + 4 j + e).
// 128 MFMAs of 32 cycles per wave and 32 pixels: 9.7 TB/s of rows at two waves per SIMD on the whole chip -- above the HBM roof.
// Arithmetic: exact f32 products, f32 accumulation; the summation order differs from the VALU logits of sa_attn_mfma_kernel (rounding-level
// differences, every fixture keeps its tolerance).
__device__ long long sa_ts[16];   // phase stamps of workgroup (0, 0), wave 0 (sf_debug_sa_stamps(1); sf_debug_read_ts_sa)
__device__ int sa_dbg_on;
#define SATS(i) do { if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) sa_ts[i] = wall_clock64(); } while (0)
template <int D>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void sa_attn_tile_kernel(
    const float* __restrict__ x, int ld, long long batch_stride, const float* __restrict__ q, float scale, float eps,
    float* __restrict__ part_num, float* __restrict__ part_den, float* __restrict__ attn_out, long long attn_bs, int HW, int N, int P) {
  // A workgroup owns 512 pixels of one frame (TWO partial records: it writes its sums into the first and zeros into the second -- the slot
  // update adds the records), a wave 64 of them as FOUR tiles of 16 rows through one 8.4 KB LDS tile, with the rows of the next two tiles
  // in flight while a tile is multiplied (a register stage): the loads never stop while the matrix cores work.  80 KB of LDS: two workgroups per CU.
  // (single-shot tiles -- every workgroup loads, then computes -- ran the chip in lockstep phases at 2.5 TB/s, profiles/r04_probes.txt section 3)
  const int dbg = sa_dbg_on;
  SATS(0);
  constexpr int NS = SA_NMAX, XP = D + 4, NW = 8, TP = 16, NT4 = 4;   // tile pitch, waves, pixels per tile, tiles per wave
  constexpr int NSLAB = D / 16, NH = D / 64;
  extern __shared__ __attribute__((aligned(16))) float sa_lds[];
  float* s_x = sa_lds;                          // [NW][TP][XP]
  float* s_a = s_x + NW * TP * XP;              // [NW][TP][17]: attention tile, pixel-major (pitch 17: conflict-free column reads)
  float* s_q = s_a + NW * TP * 17;              // [NS][D]: the scaled queries
  float* s_red = sa_lds;                        // [NW][NS][XP] over the dead tiles
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pix0 = chunk * (NW * NT4 * TP) + wave * NT4 * TP;
  const int li = lane & 15, lg = lane >> 4;
  typedef float f32x4a __attribute__((ext_vector_type(4)));
  static_assert(D == 128, "sa_attn_tile_kernel: D = 128 (two whole rows per load instruction)");
  // ---- rows: instruction u of a tile brings rows 2 u and 2 u + 1 whole (lane & 31 = the 16-byte column); two tiles requested at once ----
  const float* xs = x + (long long)b * batch_stride + (long long)(pix0 + (lane >> 5)) * ld + 4 * (lane & 31);
  float* tile = s_x + wave * TP * XP;
  f32x4v stage[8];   // (one tile ahead: with two stages in flight the kernel spilled at 128 registers -- four waves per SIMD are what two workgroups per CU need)
#pragma unroll
  for (int u = 0; u < 8; ++u) stage[u] = *(const f32x4v*)(xs + (long long)(2 * u) * ld);
  for (int idx = threadIdx.x; idx < NS * D; idx += 512) {
    const int n = idx / D, d = idx - n * D;
    s_q[idx] = n < N ? q[((long long)b * N + n) * D + d] * scale : 0.f;
  }
  __syncthreads();
  SATS(1);
  f32x4a nacc[NH][4];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) nacc[h][e] = f32x4a{0.f, 0.f, 0.f, 0.f};
  float den = 0.f;
  float* at = s_a + wave * TP * 17;
#pragma unroll 1
  for (int ti = 0; ti < NT4; ++ti) {
#pragma unroll
    for (int u = 0; u < 8; ++u) *(f32x4v*)(tile + (2 * u + (lane >> 5)) * XP + 4 * (lane & 31)) = stage[u];
    if (ti + 1 < NT4) {   // the stage is free: the rows of the next tile, in flight while this one is multiplied
#pragma unroll
      for (int u = 0; u < 8; ++u) stage[u] = *(const f32x4v*)(xs + (long long)(TP * (ti + 1) + 2 * u) * ld);
    }
    __builtin_amdgcn_wave_barrier();
    // ---- logits[pixel][slot]: A = tile rows (i = li: pixel, k = lg), B (k = lg, j = li: slot) = the scaled queries from LDS (zero beyond 8) ----
    f32x4a acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sl = 0; sl < NSLAB; ++sl) {
      const f32x4v xa = *(const f32x4v*)(tile + li * XP + 16 * sl + 4 * lg);
      f32x4v qb = *(const f32x4v*)(s_q + (li & 7) * D + 16 * sl + 4 * lg);
      if (li >= NS) qb = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[e], qb[e], acc, 0, 0, 0);
    }
    // acc[r]: pixel 4 lg + r of the tile, slot li.  Softmax over the slots (lanes li = 0..7 of every 16-lane row), + eps
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool live = li < N;
      const float v = live ? acc[r] : -INFINITY;
      const float mx = sf_max8(li < 8 ? v : -INFINITY);
      const float ex = live ? expf(v - mx) : 0.f;
      const float sum = sf_sum8(li < 8 ? ex : 0.f);
      const float a0 = ex / sum;
      const int pix = 4 * lg + r;
      if (attn_out && live) attn_out[(long long)b * attn_bs + (long long)li * HW + pix0 + TP * ti + pix] = a0;
      const float a = live ? a0 + eps : 0.f;
      den += a;
      at[pix * 17 + li] = a;   // (lanes li >= N: zero columns -- the padded slot rows of the A operand below)
    }
    __builtin_amdgcn_wave_barrier();
    // ---- num[slot][channel] += A^T . X: A (i = li: slot, k = lg: pixel 4 ks + lg) from the attention tile, B (k = lg: pixel, j = li) four
    //      channels per 16-byte read: MFMA (h, e) column li is channel 64 h + 4 li + e ----
#pragma unroll
    for (int ks = 0; ks < TP / 4; ++ks) {
      const float aop = at[(4 * ks + lg) * 17 + li];
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const f32x4v xb = *(const f32x4v*)(tile + (4 * ks + lg) * XP + 64 * h + 4 * li);
#pragma unroll
        for (int e = 0; e < 4; ++e) nacc[h][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop, xb[e], nacc[h][e], 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();   // the tile and the attention tile are rewritten by the next tile
    if (ti < 4) SATS(2 + ti);
  }
  // den: the slot's sum over this wave's pixels = over the lanes with the same li
  den += __shfl_xor(den, 16, 64);
  den += __shfl_xor(den, 32, 64);
  // ---- cross-wave reduction over the dead tiles: nacc[h][e][r] = num[slot 4 lg + r][channel 64 h + 4 li + e] (slots < 8: lg < 2) ----
  __syncthreads();
  if (lg < 2) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        f32x4v o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = nacc[h][e][r];
        *(f32x4v*)(s_red + (wave * NS + 4 * lg + r) * XP + 64 * h + 4 * li) = o;
      }
  }
  if (lg == 0 && li < NS) s_red[(wave * NS + li) * XP + D] = den;
  __syncthreads();
  SATS(6);
  // the workgroup's sums go to record 2 chunk, zeros to record 2 chunk + 1 (P = HW / 256 records per frame, summed by the slot update)
  for (int idx = threadIdx.x; idx < N * (D + 1); idx += 512) {
    const int n = idx / (D + 1), d = idx - n * (D + 1);
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += s_red[(w * NS + n) * XP + d];
    if (d < D) {
      part_num[(((long long)b * P + 2 * chunk) * N + n) * D + d] = t;
      part_num[(((long long)b * P + 2 * chunk + 1) * N + n) * D + d] = 0.f;
    } else {
      part_den[((long long)b * P + 2 * chunk) * N + n] = t;
      part_den[((long long)b * P + 2 * chunk + 1) * N + n] = 0.f;
    }
  }
  SATS(7);
}

bool sf_slot_attn_sparse_records(const float* k, const float* v, int HW, int D) { return HW % 512 == 0 && k == v && D == 128; }

int sf_slot_attn_iter_ex(const float* k, const float* v, int ld, long long batch_stride, const float* q,
                         float* part_num, float* part_den, float* attn_out, long long attn_batch_stride, int B,
                         int HW, int N, int D, float scale, float eps, hipStream_t st) {
  SF_REQUIRE(k && v && q && part_num && part_den, "null pointer");
  SF_REQUIRE(B >= 0 && HW > 0 && N >= 1 && N <= SA_NMAX, "need 1 <= num_slots <= 8");
  SF_REQUIRE(D == 64 || D == 128 || D == 192 || D == 256, "slot_size must be 64/128/192/256");
  SF_REQUIRE(ld >= D, "bad leading dimension");
  const int P = sf_sa_pick_partials(HW);
  SF_REQUIRE((HW % P) == 0 && ((HW / P) % 16) == 0, "HW must be a multiple of 16");
  SF_REQUIRE((ld % 4) == 0 && (batch_stride % 4) == 0, "k/v rows must be 16-byte aligned");
  if (B == 0) return 0;
  dim3 grid(P, B), block(256);
#define SA_LAUNCH(VPT)                                                                            \
  hipLaunchKernelGGL(sa_attn_partial_kernel<VPT>, grid, block, 0, st, k, v, ld, batch_stride, q, \
                     scale, eps, part_num, part_den, attn_out, attn_batch_stride, HW, N, P)
  // algorithmic bytes: one read of K and V -- of the ONE feature array when keys and values are the same rows
  // (the folded form, engine.hip: k == v == the normalised pixel features)
  sf_prof_begin(SF_K_SA_ITER, st, (k == v ? 1.0 : 2.0) * (double)B * HW * D * sizeof(float));
  // (a single-shot one-pass kernel -- one 33 KB tile per wave, one workgroup per CU -- measured slower than the two-pass kernel and is gone:
  //  profiles/r03_probes.txt)
  // (round 4, for keys == values at width 128.  32 frames: 21.3 vs 26.4 us on the whole chip, 31.2 vs 35.5
  //  on a 128-CU mask, 67 instead of 120-132 MB fetched; its single-shot forms -- one tile per wave -- were no faster: profiles/r04_probes.txt section 3)
  if (HW % 512 == 0 && k == v && D == 128 && P == HW / 256) {
    // keys == values: every row read once, both products on the matrix cores (sa_attn_tile_kernel: 512 pixels per workgroup)
    constexpr size_t LDS = (size_t)(8 * 16 * (128 + 4) + 8 * 16 * 17 + SA_NMAX * 128) * sizeof(float);   // 80,384 B: two workgroups per CU
    static_assert(2 * LDS <= 160 * 1024, "one-pass Slot Attention: two workgroups per CU");
    SF_TRY(sf_ensure_dyn_lds((const void*)sa_attn_tile_kernel<128>, LDS));
    hipLaunchKernelGGL(sa_attn_tile_kernel<128>, dim3(HW / 512, B), dim3(512), LDS, st, k, ld, batch_stride, q, scale, eps, part_num, part_den, attn_out,
                       attn_batch_stride, HW, N, P);
  } else if (HW % 256 == 0) {
#define SA_LAUNCH2(DD)                                                                                 \
  hipLaunchKernelGGL(sa_attn_mfma_kernel<DD>, grid, block, 0, st, k, v, ld, batch_stride, q, scale, eps, \
                     part_num, part_den, attn_out, attn_batch_stride, HW, N, P)
    switch (D / 64) {
      case 1: SA_LAUNCH2(64); break;
      case 2: SA_LAUNCH2(128); break;
      case 3: SA_LAUNCH2(192); break;
      default: SA_LAUNCH2(256); break;
    }
#undef SA_LAUNCH2
  } else {
    switch (D / 64) {
      case 1: SA_LAUNCH(1); break;
      case 2: SA_LAUNCH(2); break;
      case 3: SA_LAUNCH(3); break;
      default: SA_LAUNCH(4); break;
    }
  }
#undef SA_LAUNCH
  sf_prof_end(SF_K_SA_ITER, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// Slot update (savi.py:95-100): reduce partials -> GRUCell -> slots + MLP(LN(slots)); optional extras (second strided
// copy of the result, q projection for the next iteration).  Weight matrices are TRANSPOSED torch weights
// ([in, out] = weight.t().contiguous()), q_w included.
int sf_slot_update_ex(const float* part_num, const float* part_den, int P, const float* slots_prev,
                      const float* gru_w_ih, const float* gru_w_hh, const float* gru_b_ih, const float* gru_b_hh,
                      const float* ln_g, const float* ln_b, const float* mlp_w1, const float* mlp_b1, const float* mlp_w2,
                      const float* mlp_b2, float* slots_out, float* out2, long long out2_bs, const float* q_ln_g,
                      const float* q_ln_b, const float* q_w, float* q_out, int B, int N, int D, int H, float ln_eps,
                      hipStream_t st) {
  SF_REQUIRE(part_num && part_den && slots_prev && slots_out, "null pointer");
  SF_REQUIRE(gru_w_ih && gru_w_hh && gru_b_ih && gru_b_hh && ln_g && ln_b && mlp_w1 && mlp_b1 && mlp_w2 &&
                 mlp_b2, "null weight pointer");
  SF_REQUIRE(D > 0 && D <= SA_DMAX && (D % 64) == 0 && H > 0 && H <= SA_HMAX && (H % 64) == 0 && N >= 1 && P >= 1 && P <= 64,
             "bad slot shape (slot_size and slot_mlp_size must be multiples of 64, at most 64 partial records)");
  SF_REQUIRE(q_out == nullptr || (q_ln_g && q_ln_b && q_w), "q projection requested without its weights");
  if (B == 0) return 0;
  // 768 threads whenever 3 D fits: the spare threads take K slices of every product (one round trip of weight requests per
  // phase instead of two to four)
  int threads = 3 * D > H ? 3 * D : H;
  threads = (threads + 63) & ~63;
  if (threads < 768) threads = 768;
  const int R = B * N;
  auto pmaxf = [&](int Nout, int K) { return su_kslices(threads, Nout, K) * Nout * SU_R; };
  int pmax = pmaxf(3 * D, D);
  if (pmaxf(H, D) > pmax) pmax = pmaxf(H, D);
  if (pmaxf(D, H) > pmax) pmax = pmaxf(D, H);
  if (pmaxf(D, D) > pmax) pmax = pmaxf(D, D);
  if (pmax < SU_R * 64) pmax = SU_R * 64;
  const size_t lds = ((size_t)4 * D * SU_R + (size_t)H * SU_R + 2 * SU_R + 2 * (size_t)pmax) * sizeof(float);
  auto kern = sa_slot_update_kernel<768>;
  SF_TRY(sf_ensure_dyn_lds((const void*)kern, lds));
  SuExtra ex{out2, out2_bs, q_ln_g, q_ln_b, q_w, q_out};
  sf_prof_begin(SF_K_SA_UPDATE, st, 0.0);
  hipLaunchKernelGGL(kern, dim3((R + SU_R - 1) / SU_R), dim3(threads), lds, st, part_num, part_den, P, slots_prev, gru_w_ih,
                     gru_w_hh, gru_b_ih, gru_b_hh, ln_g, ln_b, mlp_w1, mlp_b1, mlp_w2, mlp_b2, slots_out, ex, R, N, D, H,
                     pmax, ln_eps);
  sf_prof_end(SF_K_SA_UPDATE, st);
  SF_CHECK_LAUNCH();
  return 0;
}

extern "C" {
int sf_slot_update_f32(const float* part_num, const float* part_den, int P, const float* slots_prev,
                       const float* gru_w_ih, const float* gru_w_hh, const float* gru_b_ih,
                       const float* gru_b_hh, const float* ln_g, const float* ln_b, const float* mlp_w1,
                       const float* mlp_b1, const float* mlp_w2, const float* mlp_b2, float* slots_out,
                       int B, int N, int D, int H, float ln_eps, void* stream) {
  return sf_slot_update_ex(part_num, part_den, P, slots_prev, gru_w_ih, gru_w_hh, gru_b_ih, gru_b_hh, ln_g, ln_b, mlp_w1,
                           mlp_b1, mlp_w2, mlp_b2, slots_out, nullptr, 0, nullptr, nullptr, nullptr, nullptr, B, N, D, H,
                           ln_eps, (hipStream_t)stream);
}

}  // extern "C"

extern "C" int sf_debug_read_ts_sa(long long* out16) {
  hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(sa_ts), sizeof(long long) * 16);
  return e == hipSuccess ? 0 : (int)e;
}
extern "C" int sf_debug_sa_stamps(int on) {
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(sa_dbg_on), &on, sizeof(int));
  return e == hipSuccess ? 0 : (int)e;
}
