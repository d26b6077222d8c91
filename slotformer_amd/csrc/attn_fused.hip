// Fused per-head attention block:  att[:, h] = MHA_h( LN?(x) )  for one (head, sequence) per workgroup.
//
//   LN1 -> q|k|v projection of head h (split-bf16 MFMA, f32 accumulate) -> scores -> softmax -> PV
//
// replaces three launches of the unfused chain (fused-LN QKV GEMM over all heads, then the
// attention kernel) and the [B*L, 3d] qkv round trip through memory.  Each sequence has L <= 64
// tokens (rollout window: 36-48; slot predictor: 6-8), so a (head, sequence) problem is a
// [64 x d] . [d x 3*hd] GEMM plus a 64x64 attention -- it fits one CU's LDS.
// Reference call sites: nn.TransformerEncoderLayer self-attention half (slotformer.py:72-80,119;
// predictor.py:33-44).
#include "sf_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int FA_NT = 512;   // 8 waves: the projection is fill-rate bound, more loads in flight per CU
// padded sequence length: template parameter ROWS = 64 (two 32-row MFMA blocks) or 128 (four: the 15-frame Physion window, 90 tokens)
constexpr int FA_KC = 64;    // k-chunk of d staged per barrier
constexpr int FA_LB = FA_KC + 8;

template <int HD, int ROWS>
struct FaCfg {
  static constexpr int FA_ROWS = ROWS;
  static constexpr int FA_SS = ROWS + 4;             // row stride of the score / V^T tiles (17 / 33 16-B slots: conflict-free b128)
  static constexpr int NC = 3 * HD;                  // q|k|v columns of one head
  static constexpr int NCP = (NC + 31) / 32 * 32;    // padded to MFMA blocks
  static constexpr int CBLK = NCP / 32;
  static constexpr int MAXB = ((ROWS / 32) * CBLK + 7) / 8;    // 32x32 projection blocks per wave
  static constexpr int B_IT = NCP * (FA_KC / 4) / FA_NT;
  static constexpr int QSTR = HD + 4;                // f32 row stride of Q and K tiles ((HD+4)/4 is odd)
  static constexpr int HDP = (HD + 31) / 32 * 32;    // V^T rows padded to MFMA blocks
  static constexpr size_t plane_floats = (size_t)(FA_ROWS + NCP) * FA_LB + 2 * FA_ROWS;  // 2 bf16 planes + stats
  static constexpr size_t attn_floats = (size_t)2 * FA_ROWS * QSTR + (size_t)HDP * FA_SS + (size_t)FA_ROWS * FA_SS + FA_ROWS;
  static constexpr size_t lds_bytes = (plane_floats > attn_floats ? plane_floats : attn_floats) * sizeof(float);
};
}  // namespace

// NK = d / 64 chunks; every chunk's global loads are issued before the first wait, so the projection
// pays one memory latency instead of NK.  Attention (scores, PV) runs on the exact-f32 MFMA.
template <int HD, int NK, int ROWS>
__global__ __launch_bounds__(FA_NT) void qkv_attn_kernel(const float* __restrict__ x, const float* __restrict__ ln_g,
                                                         const float* __restrict__ ln_b, float ln_eps,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, int L, int Lq, int d) {
  using C = FaCfg<HD, ROWS>;
  constexpr int NCP = C::NCP, CBLK = C::CBLK, MAXB = C::MAXB, B_IT = C::B_IT, QSTR = C::QSTR, HDP = C::HDP;
  constexpr int FA_ROWS = ROWS, FA_SS = C::FA_SS;
  constexpr int A_IT = FA_ROWS * (FA_KC / 4) / FA_NT;  // 2
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* Ah = (__bf16*)smem;
  __bf16* Al = Ah + FA_ROWS * FA_LB;
  __bf16* Bh = Al + FA_ROWS * FA_LB;
  __bf16* Bl = Bh + NCP * FA_LB;
  float* stats = (float*)(Bl + NCP * FA_LB);  // [2][64]
  const int h = blockIdx.x, b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* xb = x + (long long)b * L * d;
  const int c4 = t & 15, r0 = t >> 4;  // 16 float4 per 64-wide chunk row; 32 rows per pass

  // ---- per-thread source rows (clamped; rows >= L are zeroed by a select) ----------------
  const float* arow[A_IT];
  bool aok[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int r = r0 + 32 * i;
    aok[i] = r < L;
    arow[i] = xb + (long long)min(r, L - 1) * d;
  }
  const float* wrow[B_IT];
  bool wok[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int n = r0 + 32 * i;              // local column: which * HD + j
    const int which = n / HD, j = n - which * HD;
    wok[i] = n < C::NC;
    wrow[i] = w + (long long)(wok[i] ? which * d + h * HD + j : 0) * d;
  }
  f32x4 ra[NK][A_IT], rb[NK][B_IT];
#pragma unroll
  for (int kc = 0; kc < NK; ++kc) {
    const int k = kc * FA_KC + 4 * c4;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) ra[kc][i] = *(const f32x4*)(arow[i] + k);
#pragma unroll
    for (int i = 0; i < B_IT; ++i) rb[kc][i] = *(const f32x4*)(wrow[i] + k);
  }

  // ---- LayerNorm statistics straight from the registers: row r0+32*i is held by the 16 lanes c4 = 0..15 -------
  if (ln_g) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      float s = 0.f;
#pragma unroll
      for (int kc = 0; kc < NK; ++kc) s += (ra[kc][i][0] + ra[kc][i][1]) + (ra[kc][i][2] + ra[kc][i][3]);
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
      const float mean = s / (float)d;
      float vs = 0.f;
#pragma unroll
      for (int kc = 0; kc < NK; ++kc) {
        const f32x4 dv = ra[kc][i] - mean;
        vs += (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]);
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) vs += __shfl_xor(vs, o, 64);
      if (c4 == 0) {
        stats[r0 + 32 * i] = mean;
        stats[FA_ROWS + r0 + 32 * i] = 1.0f / sqrtf(vs / (float)d + ln_eps);
      }
    }
  } else if (t < FA_ROWS) {
    stats[t] = 0.f;
    stats[FA_ROWS + t] = 1.f;
  }
  __syncthreads();

  auto put = [&](__bf16* hp, __bf16* lp, int row, f32x4 v) {
    const bf16x4 hi = __builtin_convertvector(v, bf16x4);
    const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
    *(bf16x4*)(hp + row * FA_LB + 4 * c4) = hi;
    *(bf16x4*)(lp + row * FA_LB + 4 * c4) = lo;
  };
  auto store_chunk = [&](int kc, const f32x4(&rA)[A_IT], const f32x4(&rB)[B_IT]) {
    const int k = kc * FA_KC + 4 * c4;
    f32x4 g = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
    if (ln_g) {
      g = *(const f32x4*)(ln_g + k);
      be = *(const f32x4*)(ln_b + k);
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int r = r0 + 32 * i;
      const f32x4 v = (rA[i] - stats[r]) * stats[FA_ROWS + r] * g + be;
      put(Ah, Al, r, aok[i] ? v : zero4);
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) put(Bh, Bl, r0 + 32 * i, wok[i] ? rB[i] : zero4);
  };

  // ---- q|k|v = LN(x) . W_h^T on split-bf16 MFMA -----------------------------------------------
  f32x16 acc[MAXB];
#pragma unroll
  for (int q = 0; q < MAXB; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const int nrb = (L + 31) / 32;   // row blocks beyond the sequence are skipped entirely
  const int nblk = nrb * CBLK;
#pragma unroll
  for (int kc = 0; kc < NK; ++kc) {
    store_chunk(kc, ra[kc], rb[kc]);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < MAXB; ++q) {
      const int blk = wave + 8 * q;
      if (blk < nblk) {
        const int rbk = blk / CBLK, cbk = blk - rbk * CBLK;
        const int ao = (rbk * 32 + (lane & 31)) * FA_LB + 8 * (lane >> 5);
        const int bo = (cbk * 32 + (lane & 31)) * FA_LB + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < FA_KC / 16; ++ks) {
          const bf16x8 xh = *(const bf16x8*)(Ah + ao + ks * 16), xl = *(const bf16x8*)(Al + ao + ks * 16);
          const bf16x8 yh = *(const bf16x8*)(Bh + bo + ks * 16), yl = *(const bf16x8*)(Bl + bo + ks * 16);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh, acc[q], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // ---- spill q (scaled), k as [token][channel] and v transposed as [channel][token]; planes are dead ----
  float* Qs = smem;                       // [64][QSTR]
  float* Ks = Qs + FA_ROWS * QSTR;        // [64][QSTR]
  float* VT = Ks + FA_ROWS * QSTR;        // [HDP][FA_SS]
  float* Ss = VT + HDP * FA_SS;           // [64][FA_SS]
  float* inv = Ss + FA_ROWS * FA_SS;      // [64]
  const float scale = 1.0f / sqrtf((float)HD);
  if constexpr (HDP > HD) {               // zero the padded V^T rows (they feed the PV MFMA)
    for (int idx = t; idx < (HDP - HD) * FA_SS; idx += FA_NT) VT[HD * FA_SS + idx] = 0.f;
  }
#pragma unroll
  for (int q = 0; q < MAXB; ++q) {
    const int blk = wave + 8 * q;
    if (blk < nblk) {
      const int rbk = blk / CBLK, cbk = blk - rbk * CBLK;
      const int n = cbk * 32 + (lane & 31);
      const int which = n / HD, j = n - which * HD;
      if (n < C::NC) {
        const float bv = bias[which * d + h * HD + j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbk * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const float val = acc[q][r] + bv;
          if (which == 0)
            Qs[row * QSTR + j] = val * scale;
          else if (which == 1)
            Ks[row * QSTR + j] = val;
          else
            VT[j * FA_SS + row] = val;
        }
      }
    }
  }
  if (nrb == 1) {  // tokens 32..63 do not exist: their V^T columns must not inject garbage into PV
    for (int idx = t; idx < HDP * 32; idx += FA_NT) VT[(idx >> 5) * FA_SS + 32 + (idx & 31)] = 0.f;
  }
  __syncthreads();

  // ---- scores S = q k^T on the f32 MFMA: one 32x32 block per wave ----------------------------------
  const int nsb = nrb * nrb;
  for (int sblk = wave; sblk < nsb; sblk += 8) {
    const int rbk = sblk / nrb, cbk = sblk - rbk * nrb;
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
    const float* qp = Qs + (rbk * 32 + (lane & 31)) * QSTR + 4 * (lane >> 5);
    const float* kp = Ks + (cbk * 32 + (lane & 31)) * QSTR + 4 * (lane >> 5);
#pragma unroll
    for (int kb = 0; kb < HD / 8; ++kb) {
      const f32x4 a = *(const f32x4*)(qp + kb * 8), bq = *(const f32x4*)(kp + kb * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bq[s], sacc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rbk * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      Ss[row * FA_SS + cbk * 32 + (lane & 31)] = sacc[r];
    }
  }
  __syncthreads();

  // ---- row softmax over the L real keys, 8 lanes per query row; padded key columns become 0 ------------
#pragma unroll
  for (int i0 = 0; i0 < FA_ROWS; i0 += 64) {
    const int i = i0 + (t >> 3), sub = t & 7;
    const int kmax = nrb * 32;
    if (i >= L - Lq && i < L) {
      float* row = Ss + i * FA_SS;
      float mx = -INFINITY;
      for (int j = sub; j < L; j += 8) mx = fmaxf(mx, row[j]);
      mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
      float sum = 0.f;
      for (int j = sub; j < kmax; j += 8) {
        const float p = j < L ? expf(row[j] - mx) : 0.f;
        row[j] = p;
        sum += p;
      }
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      sum += __shfl_xor(sum, 4, 64);
      if (sub == 0) inv[i] = 1.0f / sum;
    }
  }
  __syncthreads();

  // ---- O = P V on the f32 MFMA: block (row block, 32-channel block) per wave -----------------------------
  constexpr int NVB = HDP / 32;
  if (wave < nrb * NVB) {
    const int rbk = wave / NVB, cbk = wave - rbk * NVB;
    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    const float* pp = Ss + (rbk * 32 + (lane & 31)) * FA_SS + 4 * (lane >> 5);
    const float* vp = VT + (cbk * 32 + (lane & 31)) * FA_SS + 4 * (lane >> 5);
    for (int kb = 0; kb < nrb * 4; ++kb) {
      const f32x4 a = *(const f32x4*)(pp + kb * 8), bq = *(const f32x4*)(vp + kb * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s) oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bq[s], oacc, 0, 0, 0);
    }
    const int c = cbk * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rbk * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row >= L - Lq && row < L && c < HD)
        out[((long long)b * Lq + (row - (L - Lq))) * d + h * HD + c] = oacc[r] * inv[row];
    }
  }
}

template <int HD, int NK, int ROWS>
static int launch_fa(const float* x, const float* ln_g, const float* ln_b, float ln_eps, const float* w,
                     const float* bias, float* out, int B, int L, int Lq, int d, int nheads, hipStream_t st) {
  constexpr size_t lds = FaCfg<HD, ROWS>::lds_bytes;
  static_assert(lds <= 160 * 1024, "fused attention: LDS budget");
  auto kern = qkv_attn_kernel<HD, NK, ROWS>;
  SF_TRY(sf_ensure_dyn_lds((const void*)kern, (size_t)(lds)));
  sf_prof_begin(SF_K_MHA, st, 6.0 * B * L * (double)d * d + 4.0 * (double)B * nheads * Lq * L * HD);
  hipLaunchKernelGGL(kern, dim3(nheads, B), dim3(FA_NT), lds, st, x, ln_g, ln_b, ln_eps, w, bias, out, L, Lq, d);
  sf_prof_end(SF_K_MHA, st);
  SF_CHECK_LAUNCH();
  return 0;
}

// Returns 1 when the fused path does not apply (caller falls back to GEMM + attention kernels).
int sf_qkv_attn_ex(const float* x, const float* ln_g, const float* ln_b, float ln_eps, const float* in_proj_w,
                   const float* in_proj_b, float* out, int B, int L, int Lq, int d, int nheads, hipStream_t st) {
  if (B <= 0) return 0;
  if (nheads <= 0 || d % nheads) return 1;
  const int hd = d / nheads;
  if (L > 128 || L < 1 || Lq < 1 || Lq > L || (d % FA_KC) != 0 || d > 256) return 1;
  // windows of 65..128 tokens (the reference's 15-frame Physion window: 90): four token blocks, head width 32 only
  if (L > 64) {
    if (hd == 32 && d == 4 * FA_KC) return launch_fa<32, 4, 128>(x, ln_g, ln_b, ln_eps, in_proj_w, in_proj_b, out, B, L, Lq, d, nheads, st);
    return 1;
  }
#define FA_CASE(HD_, NK_) \
  if (hd == HD_ && d == NK_ * FA_KC) \
    return launch_fa<HD_, NK_, 64>(x, ln_g, ln_b, ln_eps, in_proj_w, in_proj_b, out, B, L, Lq, d, nheads, st)
  // the shapes the reference configures: rollout d=128/256 (8 heads), predictor d=128/192 (4 heads)
  FA_CASE(16, 2);
  FA_CASE(32, 2);
  FA_CASE(32, 4);
  FA_CASE(48, 3);
  FA_CASE(64, 4);
#undef FA_CASE
  return 1;
}

extern "C" {
// att[B*Lq, d] = concat_h softmax(q_h k_h^T / sqrt(hd)) v_h with [q|k|v] = LN?(x) in_proj_w^T + in_proj_b;
// x [B*L, d]; queries are the last Lq tokens of each sequence; ln_gamma/ln_beta NULL = no LayerNorm.
// Supported: L <= 64 and (d, head_dim) in {(128,16), (128,32), (256,32), (192,48), (256,64)}; negative code otherwise.
int sf_qkv_attention_f32(const float* x, const float* ln_gamma, const float* ln_beta, float ln_eps,
                         const float* in_proj_w, const float* in_proj_b, float* out, int B, int L, int Lq, int d_model,
                         int num_heads, void* stream) {
  SF_REQUIRE(x && in_proj_w && in_proj_b && out, "null pointer");
  SF_REQUIRE((ln_gamma == nullptr) == (ln_beta == nullptr), "ln_gamma/ln_beta must come together");
  const int rc = sf_qkv_attn_ex(x, ln_gamma, ln_beta, ln_eps, in_proj_w, in_proj_b, out, B, L, Lq, d_model, num_heads,
                                (hipStream_t)stream);
  if (rc == 1) return sf_set_err(-1, "invalid argument: fused attention needs L <= 64, d % 64 == 0, head_dim 16/32/48/64",
                                 __FILE__, __LINE__);
  return rc;
}
}
