// Probe: cost of an all-gather among the 8 workgroups of a cluster inside one persistent launch.
// 256 WGs (one per CU), clusters of 8 placed on one XCD (block b runs on XCD b % 8 -- speed only;
// the protocol is placement-independent: agent-scope release/acquire, bounded spins).
// Each round: every WG writes its 5.4 KB slice (42 x 32 f32), all 8 publish, every WG reads the 43 KB.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define SLICE 1344  // floats per member per round (42 x 32)
#define NT 512

__device__ __forceinline__ bool wait_all(unsigned* flags, unsigned target, unsigned* err) {
  // wave 0, lanes 0..7 poll one flag each (relaxed, agent scope); bounded
  const int lane = threadIdx.x & 63;
  bool ok = true;
  if (threadIdx.x < 64) {
    unsigned spins = 0;
    for (;;) {
      unsigned v = lane < 8 ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
      if (__all((int)(v >= target))) break;
      if (++spins > 2000000u || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok && lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return ok;
}

template <int MODE>
__global__ __launch_bounds__(NT) void exch(float* buf, unsigned* flags, unsigned* err, float* sink, int rounds, int same_xcd) {
  const int bid = blockIdx.x, t = threadIdx.x;
  int cluster, member;
  if (same_xcd) { cluster = (bid & 7) * 4 + ((bid >> 3) & 3); member = bid >> 5; }
  else { cluster = bid >> 3; member = bid & 7; }
  float* cbuf = buf + (size_t)cluster * 2 * 8 * SLICE;
  unsigned* cflags = flags + cluster * 8;
  __shared__ float lds[8 * SLICE];
  float acc = 0.f;
  for (int r = 1; r <= rounds; ++r) {
    float* slot = cbuf + (size_t)(r & 1) * 8 * SLICE;
    // produce my slice
    for (int i = t; i < SLICE / 4; i += NT) {
      f4 v = {(float)r, (float)member, (float)i, acc};
      if (MODE == 0) *(f4*)(slot + member * SLICE + 4 * i) = v;
      else __builtin_nontemporal_store(v, (f4*)(slot + member * SLICE + 4 * i));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(cflags + member, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    wait_all(cflags, (unsigned)r, err);
    if (t == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    // consume all 8 slices
    for (int i = t; i < 8 * SLICE / 4; i += NT) *(f4*)(lds + 4 * i) = *(const f4*)(slot + 4 * i);
    __syncthreads();
    acc += lds[(t * 13) % (8 * SLICE)];
    // verify: element 0 of every slice must carry this round's tag
    if (t < 8 && lds[t * SLICE] != (float)r) __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (acc == 1234.5f) sink[0] = acc;
}

int main() {
  float *buf, *sink; unsigned *flags, *err;
  hipMalloc(&buf, (size_t)32 * 2 * 8 * SLICE * 4); hipMalloc(&sink, 16);
  hipMalloc(&flags, 32 * 8 * 4); hipMalloc(&err, 4);
  for (int same = 1; same >= 0; --same) for (int mode = 0; mode < 2; ++mode) {
    for (int rounds : {50, 400}) {
      hipMemset(flags, 0, 32 * 8 * 4); hipMemset(err, 0, 4);
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      if (mode == 0) exch<0><<<256, NT>>>(buf, flags, err, sink, rounds, same);
      else exch<1><<<256, NT>>>(buf, flags, err, sink, rounds, same);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      unsigned e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
      printf("same_xcd %d store_mode %d rounds %3d: %8.1f us total  %6.2f us/round  err %u\n", same, mode, rounds, ms * 1e3, ms * 1e3 / rounds, e);
    }
  }
  return 0;
}
