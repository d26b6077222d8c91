// Probe: per-CU global->LDS fill rate vs loads in flight (register-staged float4 loads + ds_write).
// Each workgroup streams `bytes_per_wg` from a buffer region into LDS, U float4 loads per thread
// in flight.  mode 0: each WG reads its own region (cold per kernel: MALL/HBM); mode 1: all WGs
// read the same 256 KB (L2 hits after first touch).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, int NT>
__global__ __launch_bounds__(NT) void fill(const float* __restrict__ src, float* __restrict__ sink, long long per_wg_floats,
                                           long long wg_stride_floats, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const float* base = src + (long long)blockIdx.x * wg_stride_floats;
  const int t = threadIdx.x;
  float acc = 0.f;
  const long long step = (long long)NT * 4 * U;  // floats per batch
  for (int it = 0; it < iters; ++it) {
    for (long long off = 0; off + step <= per_wg_floats; off += step) {
      f4 r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) r[u] = *(const f4*)(base + off + ((long long)u * NT + t) * 4);
#pragma unroll
      for (int u = 0; u < U; ++u) *(f4*)(lds + ((u * NT + t) * 4) % (8192)) = r[u];
      __syncthreads();
      acc += lds[(t * 7) % 8192];
      __syncthreads();
    }
  }
  if (acc == 12345.678f) sink[0] = acc;
}

template <int U, int NT>
void run(const float* d, float* sink, int nwg, long long per_wg_bytes, int mode) {
  long long per = per_wg_bytes / 4, stride = mode == 0 ? per : 0;
  int iters = 4;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  fill<U, NT><<<nwg, NT, 8192 * 4>>>(d, sink, per, stride, 1);
  hipDeviceSynchronize();
  hipEventRecord(a);
  fill<U, NT><<<nwg, NT, 8192 * 4>>>(d, sink, per, stride, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double bytes = (double)nwg * per_wg_bytes * iters;
  printf("mode %d NT %4d U %2d nwg %4d per_wg %4lld KB: %7.1f us  %6.2f TB/s  %6.1f GB/s per WG\n", mode, NT, U, nwg,
         per_wg_bytes / 1024, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / nwg);
}

int main() {
  float *d, *sink;
  size_t total = (size_t)512 * 1024 * 1024;
  hipMalloc(&d, total); hipMalloc(&sink, 16);
  hipMemset(d, 0, total);
  for (int mode = 0; mode < 2; ++mode) {
    for (long long kb : {128, 512}) {
      run<4, 256>(d, sink, 256, kb * 1024, mode);
      run<8, 256>(d, sink, 256, kb * 1024, mode);
      run<16, 256>(d, sink, 256, kb * 1024, mode);
      run<8, 512>(d, sink, 256, kb * 1024, mode);
      run<8, 1024>(d, sink, 256, kb * 1024, mode);
      run<8, 256>(d, sink, 512, kb * 1024, mode);
      run<8, 256>(d, sink, 1024, kb * 1024, mode);
    }
  }
  return 0;
}
