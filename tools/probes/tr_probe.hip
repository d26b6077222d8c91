// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds u16 value = its own index; lane l reads at element address addr[l] (4-element aligned);
// prints, per lane, the four values returned.    hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe && ./tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[4 * threadIdx.x + j] = v[j];
}
int main() {
  int h_addr[64];
  // lane (j = l & 15, kb = l >> 4): row = 8 kb + (j >> 2), 4 contiguous elements at column 4 (j & 3); row stride 128 elements
  for (int l = 0; l < 64; ++l) { int j = l & 15, kb = l >> 4; h_addr[l] = (8 * kb + (j >> 2)) * 128 + 4 * (j & 3); }
  int* d_addr; short* d_out; short h_out[256];
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    int j = l & 15, kb = l >> 4;
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) {
      int v = h_out[4 * l + e], row = v / 128, col = v % 128;
      printf(" (r%d,c%d)", row, col);
      if (row != 8 * kb + e || col != j) bad++;
    }
    printf("\n");
  }
  printf("expected lane (j, kb) elem e = (row 8 kb + e, col j): %s (%d mismatches)\n", bad ? "NO" : "YES", bad);
  return 0;
}
