// Probe: where do the workgroups of a SMALL launch land under a CU mask?  Each workgroup (512 threads, 100 KB of LDS: one
// per CU) records XCC_ID / HW_ID and its start and end time, and spins ~10 us in between.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/dispatch_probe.hip -o /tmp/dispatch_probe && /tmp/dispatch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

struct Rec {
  unsigned hw, xcc;
  unsigned long long t0, t1;
};

__global__ __launch_bounds__(512) void probe(Rec* out, int spin_ticks) {
  extern __shared__ float lds[];
  const unsigned long long t0 = wall_clock64();
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
  __syncthreads();
  if (threadIdx.x == 0) {
    out[blockIdx.x].hw = hw;
    out[blockIdx.x].xcc = xcc;
    out[blockIdx.x].t0 = t0;
    out[blockIdx.x].t1 = wall_clock64();
  }
}

static void run(const char* name, const unsigned* words, int nblk, Rec* d) {
  hipStream_t st;
  if (words) {
    if (hipExtStreamCreateWithCUMask(&st, 8, words) != hipSuccess) { printf("%s: mask rejected\n", name); return; }
  } else {
    hipStreamCreate(&st);
  }
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe, dim3(nblk), dim3(512), 100 * 1024, st, d, 1000);
    hipStreamSynchronize(st);
  }
  std::vector<Rec> h(nblk);
  hipMemcpy(h.data(), d, nblk * sizeof(Rec), hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull, tmax = 0;
  for (auto& r : h) { tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1); }
  printf("%-26s %3d blocks: span %.1f us\n", name, nblk, (tmax - tmin) * 0.01);
  for (int i = 0; i < nblk; ++i) {
    const Rec& r = h[i];
    // gfx9 HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
    printf("  blk %2d xcc %u se %u cu %2u start %6.1f end %6.1f%s", i, r.xcc & 15, (r.hw >> 13) & 7, (r.hw >> 8) & 15, (r.t0 - tmin) * 0.01,
           (r.t1 - tmin) * 0.01, (i % 4 == 3) ? "\n" : " |");
  }
  hipStreamDestroy(st);
}

int main(int argc, char** argv) {
  const int nblk = argc > 1 ? atoi(argv[1]) : 32;
  Rec* d;
  hipMalloc(&d, 1024 * sizeof(Rec));
  run("no mask", nullptr, nblk, d);
  unsigned w[8];
  for (int i = 0; i < 8; ++i) w[i] = ~0xffu;
  run("complement of 0xff x 8", w, nblk, d);
  for (int i = 0; i < 8; ++i) w[i] = i < 3 ? 0u : 0xffffffffu;
  run("rows 3..7", w, nblk, d);
  for (int i = 0; i < 8; ++i) w[i] = i < 2 ? 0u : 0xffffffffu;
  run("rows 2..7", w, nblk, d);
  for (int i = 0; i < 8; ++i) w[i] = i < 3 ? 0xffffffffu : 0u;
  run("rows 0..2", w, nblk, d);
  return 0;
}
