// Backward of the attention half of one Slot-Attention iteration (forward: slot_attn.hip,
// reference savi.py:82-94 / steve.py:50-66 under autograd).
//
//   forward, per frame:  l[p,n] = scale k[p].q[n];  a = softmax_n(l);  a' = a + eps;
//                        num[n] = sum_p a'[p,n] v[p];  den[n] = sum_p a'[p,n];  U[n] = num[n] / den[n]
//   backward from dU:    g[n] = dU[n] / den[n];   c[n] = dU[n].U[n] / den[n]
//                        da'[p,n] = g[n].v[p] - c[n];        dv[p] = sum_n a'[p,n] g[n]
//                        dl[p,n]  = a[p,n] (da'[p,n] - sum_m a[p,m] da'[p,m])
//                        dk[p]    = scale sum_n dl[p,n] q[n];   dq[n] = scale sum_p dl[p,n] k[p]
//
// HBM-bound like the forward: one pass that reads K and V and writes (or accumulates into) dK and dV -- 4 D floats per
// pixel against ~70 D flops.  16 lanes share a pixel (lane j owns the float4 channel groups j, j+16, ...: a 16-lane row
// reads whole 256-byte segments), so the 2N dot products per pixel are 4-step DPP row reductions with no LDS or
// cross-row traffic.  The running dq sums live in registers; the scaled queries and g sit in LDS and are read as broadcasts
// (the 16 lanes of a row read 16 consecutive float4, every row the same ones) -- with them in registers too the kernel
// needed 348 VGPRs, one wave per SIMD, and reached 3.3 TB/s; this way two workgroups share a CU.  Every workgroup emits one
// dq partial record, summed in a fixed order by a second kernel (deterministic).
#include "../../include/slotformer_hip.h"
#include "sf_internal.h"

#define SAB_NMAX 8
#define SAB_PIX 256   // pixels per workgroup

// g = dU / den, c = dU.U / den per (frame, slot); one wave per (b, n)
__global__ __launch_bounds__(64) void sa_bwd_prep_kernel(const float* __restrict__ part_num, const float* __restrict__ part_den,
                                                        int P, const float* __restrict__ d_updates, float* __restrict__ g,
                                                        float* __restrict__ c, int N, int D) {
  const int n = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  float den = 0.f;
  for (int p = 0; p < P; ++p) den += part_den[((long long)b * P + p) * N + n];
  const float inv = 1.f / den;
  float dot = 0.f;
  for (int ch = lane; ch < D; ch += 64) {
    float num = 0.f;
    for (int p = 0; p < P; ++p) num += part_num[(((long long)b * P + p) * N + n) * D + ch];
    const float du = d_updates[((long long)b * N + n) * D + ch];
    g[((long long)b * N + n) * D + ch] = du * inv;
    dot += du * (num * inv);
  }
  dot = sf_wave_sum(dot);
  if (lane == 0) c[(long long)b * N + n] = dot * inv;
}

template <int D>
__global__ __launch_bounds__(256, (D <= 128 ? 2 : 1)) void sa_iter_bwd_kernel(const float* __restrict__ k, const float* __restrict__ v, int ld,
                                                          long long batch_stride, const float* __restrict__ q,
                                                          const float* __restrict__ g, const float* __restrict__ c,
                                                          float* __restrict__ dk, float* __restrict__ dv, int accumulate,
                                                          float* __restrict__ dq_part, int HW, int N, float scale, float eps) {
  constexpr int C4 = D / 64;   // float4 groups per lane
  extern __shared__ float red[];
  const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
  const int tid = threadIdx.x, j = tid & 15, grp = tid >> 4;   // 16 pixel groups per workgroup
  // red: [16][D] floats for the final dq reduction, then qs [NMAX][D] (scaled queries) and gg [NMAX][D]
  float* qsl = red + 16 * D;
  float* ggl = qsl + SAB_NMAX * D;
  for (int i = tid; i < SAB_NMAX * D; i += 256) {
    const int n = i / D;
    qsl[i] = n < N ? q[(long long)b * N * D + i] * scale : 0.f;
    ggl[i] = n < N ? g[(long long)b * N * D + i] : 0.f;
  }
  float4 dqa[SAB_NMAX][C4];
  float cc[SAB_NMAX];
#pragma unroll
  for (int n = 0; n < SAB_NMAX; ++n) {
    cc[n] = n < N ? c[(long long)b * N + n] : 0.f;
#pragma unroll
    for (int i = 0; i < C4; ++i) dqa[n][i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
#define QS(n, i) (*reinterpret_cast<const float4*>(&qsl[(n) * D + 4 * (j + 16 * (i))]))
#define GG(n, i) (*reinterpret_cast<const float4*>(&ggl[(n) * D + 4 * (j + 16 * (i))]))
  const long long fb = (long long)b * batch_stride;
  const int p0 = chunk * SAB_PIX;
  float4 kk[C4], vv[C4];
  auto row = [&](int p) { return fb + (long long)p * ld + 4 * j; };
  {
    const int p = p0 + grp;
#pragma unroll
    for (int i = 0; i < C4; ++i) {
      kk[i] = p < HW ? *reinterpret_cast<const float4*>(k + row(p) + 64 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
      vv[i] = p < HW ? *reinterpret_cast<const float4*>(v + row(p) + 64 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  for (int step = 0; step < SAB_PIX / 16; ++step) {
    const int p = p0 + step * 16 + grp;
    float4 kc[C4], vc[C4];
#pragma unroll
    for (int i = 0; i < C4; ++i) {
      kc[i] = kk[i];
      vc[i] = vv[i];
    }
    const int pn = p + 16;
    if (step + 1 < SAB_PIX / 16) {
#pragma unroll
      for (int i = 0; i < C4; ++i) {
        kk[i] = pn < HW ? *reinterpret_cast<const float4*>(k + row(pn) + 64 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
        vv[i] = pn < HW ? *reinterpret_cast<const float4*>(v + row(pn) + 64 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float4 ok[C4], ov[C4];
    if (accumulate && p < HW) {
#pragma unroll
      for (int i = 0; i < C4; ++i) {
        ok[i] = *reinterpret_cast<const float4*>(dk + row(p) + 64 * i);
        ov[i] = *reinterpret_cast<const float4*>(dv + row(p) + 64 * i);
      }
    } else {
#pragma unroll
      for (int i = 0; i < C4; ++i) ok[i] = ov[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s[SAB_NMAX], t[SAB_NMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int n = 0; n < SAB_NMAX; ++n) {
      float a = 0.f, e = 0.f;
#pragma unroll
      for (int i = 0; i < C4; ++i) {
        const float4 qv = QS(n, i), gv = GG(n, i);
        a += (kc[i].x * qv.x + kc[i].y * qv.y) + (kc[i].z * qv.z + kc[i].w * qv.w);
        e += (vc[i].x * gv.x + vc[i].y * gv.y) + (vc[i].z * gv.z + vc[i].w * gv.w);
      }
      s[n] = n < N ? sf_sum16(a) : -INFINITY;
      t[n] = sf_sum16(e);
      mx = fmaxf(mx, s[n]);
    }
    float sum = 0.f;
#pragma unroll
    for (int n = 0; n < SAB_NMAX; ++n) {
      s[n] = n < N ? __expf(s[n] - mx) : 0.f;
      sum += s[n];
    }
    const float inv = 1.f / sum;
    float dot = 0.f;
#pragma unroll
    for (int n = 0; n < SAB_NMAX; ++n) {
      s[n] *= inv;            // a[p,n]
      t[n] -= cc[n];          // da'[p,n]
      dot += s[n] * t[n];
    }
#pragma unroll
    for (int n = 0; n < SAB_NMAX; ++n) {
      const float dl = s[n] * (t[n] - dot);
      const float ap = n < N ? s[n] + eps : 0.f;
#pragma unroll
      for (int i = 0; i < C4; ++i) {
        const float4 qv = QS(n, i), gv = GG(n, i);
        ok[i].x += dl * qv.x; ok[i].y += dl * qv.y; ok[i].z += dl * qv.z; ok[i].w += dl * qv.w;
        ov[i].x += ap * gv.x; ov[i].y += ap * gv.y; ov[i].z += ap * gv.z; ov[i].w += ap * gv.w;
        dqa[n][i].x += dl * kc[i].x; dqa[n][i].y += dl * kc[i].y; dqa[n][i].z += dl * kc[i].z; dqa[n][i].w += dl * kc[i].w;
      }
    }
    if (p < HW) {
#pragma unroll
      for (int i = 0; i < C4; ++i) {
        *reinterpret_cast<float4*>(dk + row(p) + 64 * i) = ok[i];
        *reinterpret_cast<float4*>(dv + row(p) + 64 * i) = ov[i];
      }
    }
  }
  // dq partial of this workgroup: sum the 16 pixel groups through LDS, slot by slot ([16][D] floats at a time)
  for (int n = 0; n < N; ++n) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < C4; ++i) {
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int m = 0; m < SAB_NMAX; ++m)
        if (m == n) val = dqa[m][i];
      *reinterpret_cast<float4*>(&red[grp * D + 4 * (j + 16 * i)]) = val;
    }
    __syncthreads();
    for (int ch = tid; ch < D; ch += 256) {
      float a = 0.f;
#pragma unroll
      for (int gi = 0; gi < 16; ++gi) a += red[gi * D + ch];
      dq_part[(((long long)b * nchunks + chunk) * N + n) * D + ch] = a;
    }
  }
}

// dq[b][n][:] = scale * sum_chunks partial
__global__ __launch_bounds__(256) void sa_bwd_dq_reduce_kernel(const float* __restrict__ part, float* __restrict__ dq, int nchunks,
                                                               int ND, float scale) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ND) return;
  float a = 0.f;
  for (int ch = 0; ch < nchunks; ++ch) a += part[((long long)b * nchunks + ch) * ND + i];
  dq[(long long)b * ND + i] = a * scale;
}

extern "C" {

size_t sf_slot_attn_iter_bwd_workspace_bytes(int B, int HW, int N, int D) {
  const size_t nchunks = (size_t)(HW + SAB_PIX - 1) / SAB_PIX;
  return ((size_t)B * N * D + (size_t)B * N + (size_t)B * nchunks * N * D) * sizeof(float) + 256;
}

int sf_slot_attn_iter_bwd_f32(const float* k, const float* v, int ld, long long batch_stride, const float* q,
                              const float* part_num, const float* part_den, int P, const float* d_updates, float* dk, float* dv,
                              int accumulate, float* dq, int B, int HW, int N, int D, float scale, float eps, void* ws,
                              size_t ws_bytes, void* stream) {
  SF_REQUIRE(k && v && q && part_num && part_den && d_updates && dk && dv && dq && ws, "null pointer");
  SF_REQUIRE(B >= 0 && HW > 0 && N >= 1 && N <= SAB_NMAX && P >= 1, "need 1 <= num_slots <= 8");
  SF_REQUIRE(D == 64 || D == 128 || D == 192 || D == 256, "slot_size must be 64 / 128 / 192 / 256");
  SF_REQUIRE(ld >= D && (ld % 4) == 0 && (batch_stride % 4) == 0, "k/v rows must be 16-byte aligned");
  SF_REQUIRE(ws_bytes >= sf_slot_attn_iter_bwd_workspace_bytes(B, HW, N, D), "workspace too small");
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  float* g = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  float* c = g + (size_t)B * N * D;
  float* part = c + (((size_t)B * N + 3) & ~(size_t)3);
  const int nchunks = (HW + SAB_PIX - 1) / SAB_PIX;
  hipLaunchKernelGGL(sa_bwd_prep_kernel, dim3(N, B), dim3(64), 0, st, part_num, part_den, P, d_updates, g, c, N, D);
  SF_CHECK_LAUNCH();
  const size_t lds = (size_t)(16 + 2 * SAB_NMAX) * D * sizeof(float);
#define SAB_LAUNCH(DD)                                                                                                       \
  hipLaunchKernelGGL(sa_iter_bwd_kernel<DD>, dim3(nchunks, B), dim3(256), lds, st, k, v, ld, batch_stride, q, g, c, dk, dv, \
                     accumulate, part, HW, N, scale, eps)
  switch (D) {
    case 64: SAB_LAUNCH(64); break;
    case 128: SAB_LAUNCH(128); break;
    case 192: SAB_LAUNCH(192); break;
    default: SAB_LAUNCH(256); break;
  }
#undef SAB_LAUNCH
  SF_CHECK_LAUNCH();
  hipLaunchKernelGGL(sa_bwd_dq_reduce_kernel, dim3((N * D + 255) / 256, B), dim3(256), 0, st, part, dq, nchunks, N * D, scale);
  SF_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
