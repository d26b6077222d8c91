// Probe: communication floor of a weight-stationary "systolic" rollout -- ONE persistent launch in which every
// workgroup owns a fixed (stage, weight slice) of the 4-layer Transformer and the videos of a batch flow through the
// 8 stages (A0 F0 A1 F1 A2 F2 A3 F3) as micro-batches of P videos, handed from stage to stage through memory:
//
//   producer: 16-byte write-through (sc1) stores of its partial rows -> every wave drains vmcnt(0) -> __syncthreads
//             -> ONE lane: relaxed agent-scope atomic add on the video's arrival counter of that stage
//   consumer: ONE lane polls the counter (relaxed agent-scope load + s_sleep) until all producer slices of the video
//             have arrived -> __syncthreads -> 16-byte sc1 loads of every producer's partial (they bypass the L1; no
//             buffer_inv needed because the producer stored write-through) -> fixed-order sum
//
// This is the R1 form of cdna_hip_programming.md Guideline 16 (MI355X_MICROARCH.md rows handoff-flag / publish-large);
// round 1's cluster_exchange_probe used plain stores + a release fence (7.5-11.7 us per exchange).  Every loaded word
// is verified, the ring runs under uneven load (the A stages have 4 slices, the F stages 8), every spin is bounded.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/systolic_ring_probe.hip -o /tmp/ring && /tmp/ring
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define NT 512
#define NSTAGE 8
#define MAXV 64

struct Args {
  float* buf;            // [NSTAGE][MAXV][8 slices][64 rows][256]
  unsigned* cnt;         // [NSTAGE][MAXV] arrival counters
  unsigned* err;         // [4]: 0 error code, 1 mismatching words, 2 timeouts
  long long* tstamp;     // [steps] completion time of stage 7, video 0 (ticks)
  int P, B, steps, rows, delay_a, delay_f, red_mode;
};

__device__ __forceinline__ int nslices(int stage) { return (stage & 1) ? 8 : 4; }

__device__ __forceinline__ float tagv(int step, int stage, int slice) { return (float)(step * 64 + stage * 8 + slice); }

// one lane polls; returns false on timeout / abort (uniform across the workgroup through LDS)
__device__ __forceinline__ bool wait_count(unsigned* c, unsigned target, unsigned* err, int* s_ok) {
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    int ok = 1;
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > 20000000LL || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {  // 0.2 s
        ok = 0;
        __hip_atomic_fetch_add(err + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(err, 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    *s_ok = ok;
  }
  __syncthreads();
  return *s_ok != 0;
}

__global__ __launch_bounds__(NT) void ring_kernel(Args a) {
  __shared__ int s_ok;
  __shared__ float lds[64 * 260];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // role: 48 (stage, slice) pairs x P parallel videos
  const int role = blockIdx.x / a.P, p = blockIdx.x % a.P;
  int stage, slice;
  if (role < 16) { stage = 2 * (role >> 2); slice = role & 3; }          // A_l: 4 slices
  else { stage = 2 * ((role - 16) >> 3) + 1; slice = (role - 16) & 7; }   // F_l: 8 slices
  const int prev = (stage + NSTAGE - 1) % NSTAGE, nprev = nslices(prev);
  const int nmb = (a.B + a.P - 1) / a.P;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.buf, 0, 0x7fffffff, 0x00020000);
  const size_t vstride = (size_t)8 * 64 * 256;   // floats per (stage, video)
  float keep = 0.f;
  for (int s = 0; s < a.steps; ++s) {
    for (int mb = 0; mb < nmb; ++mb) {
      const int v = mb * a.P + p;
      if (v >= a.B) continue;
      // ---- wait for the producers of this video (stage 0 of step 0 starts immediately) ----
      const int pstep = (stage == 0) ? s - 1 : s;
      if (pstep >= 0) {
        if (!wait_count(a.cnt + prev * MAXV + v, (unsigned)(nprev * (pstep + 1)), a.err, &s_ok)) return;
        // ---- load + sum the producers' partials: thread = (row = wave + 8 i, float4 column lane) ----
        const unsigned base = (unsigned)(((size_t)prev * MAXV + v) * vstride * 4);
        unsigned bad = 0;
        if (a.red_mode == 0) {
          for (int i = 0; i * 8 + wave < a.rows; ++i) {
            const int r = wave + 8 * i;
            f4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int q = 0; q < nprev; ++q) {
              const f4 x = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, base + (unsigned)(((q * 64 + r) * 256 + 4 * lane) * 4), 0, 16));
              bad += (x[0] != tagv(pstep, prev, q)) | (x[1] != (float)v) | (x[2] != (float)(r * 64 + lane));
              sum += x;
            }
            *(f4*)(lds + r * 260 + 4 * lane) = sum;
          }
        } else {
          // distributed-reduce form: only ONE partial buffer (already reduced by the producer stage) is read
          for (int i = 0; i * 8 + wave < a.rows; ++i) {
            const int r = wave + 8 * i;
            const f4 x = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, base + (unsigned)(((0 * 64 + r) * 256 + 4 * lane) * 4), 0, 16));
            bad += (x[0] != tagv(pstep, prev, 0)) | (x[1] != (float)v) | (x[2] != (float)(r * 64 + lane));
            *(f4*)(lds + r * 260 + 4 * lane) = x;
          }
        }
        if (bad) __hip_atomic_fetch_add(a.err + 1, bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        keep += lds[(t * 7) % (a.rows * 260)];
      }
      // ---- stand-in for the unit's arithmetic ----
      const int delay = (stage & 1) ? a.delay_f : a.delay_a;
      if (delay > 0) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < delay) __builtin_amdgcn_s_sleep(1);
      }
      // ---- publish this slice's partial rows (write-through), then arrive ----
      const unsigned obase = (unsigned)((((size_t)stage * MAXV + v) * vstride + (size_t)slice * 64 * 256) * 4);
      const int wrows = (a.red_mode == 0 || slice == 0) ? a.rows : 0;   // reduced form: one buffer per (stage, video)
      for (int i = 0; i * 8 + wave < wrows; ++i) {
        const int r = wave + 8 * i;
        const f4 o = {tagv(s, stage, slice), (float)v, (float)(r * 64 + lane), keep * 0.f};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, o), rs, obase + (unsigned)((r * 256 + 4 * lane) * 4), 0, 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) {
        __hip_atomic_fetch_add(a.cnt + stage * MAXV + v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (stage == NSTAGE - 1 && slice == 0 && v == 0) a.tstamp[s] = wall_clock64();
      }
    }
  }
  if (keep == 1234.5f) a.buf[0] = keep;
}

int main(int argc, char** argv) {
  Args a;
  const size_t nbuf = (size_t)NSTAGE * MAXV * 8 * 64 * 256;
  hipMalloc(&a.buf, nbuf * 4);
  hipMalloc(&a.cnt, NSTAGE * MAXV * 4);
  hipMalloc(&a.err, 16);
  hipMalloc(&a.tstamp, 4096 * 8);
  struct Cfg { int P, B, rows, da, df, red; const char* name; };
  const Cfg cfgs[] = {
      {4, 32, 42, 0, 0, 0, "P=4 B=32 rows 42, no compute, consumer sums all partials"},
      {4, 32, 42, 0, 0, 1, "P=4 B=32 rows 42, no compute, one pre-reduced buffer"},
      {4, 4, 42, 0, 0, 0, "P=4 B=4 (one micro-batch: pure latency of the 8-stage ring)"},
      {4, 4, 42, 0, 0, 1, "P=4 B=4, one pre-reduced buffer"},
      {4, 32, 42, 800, 600, 0, "P=4 B=32, 8 us / 6 us stand-in compute per A / F unit"},
      {4, 32, 42, 800, 600, 1, "P=4 B=32, 8 / 6 us compute, pre-reduced"},
      {4, 32, 42, 1200, 900, 0, "P=4 B=32, 12 / 9 us compute"},
      {5, 35, 42, 800, 600, 0, "P=5 B=35 (240 workgroups), 8 / 6 us compute"},
  };
  for (const Cfg& c : cfgs) {
    a.P = c.P; a.B = c.B; a.rows = c.rows; a.delay_a = c.da; a.delay_f = c.df; a.red_mode = c.red;
    for (int steps : {10, 50}) {
      a.steps = steps;
      hipMemset(a.cnt, 0, NSTAGE * MAXV * 4);
      hipMemset(a.err, 0, 16);
      hipMemset(a.buf, 0xff, nbuf * 4);
      hipDeviceSynchronize();
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      ring_kernel<<<48 * c.P, NT>>>(a);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      unsigned e[4];
      hipMemcpy(e, a.err, 16, hipMemcpyDeviceToHost);
      std::vector<long long> ts(steps);
      hipMemcpy(ts.data(), a.tstamp, steps * 8, hipMemcpyDeviceToHost);
      const double steady = steps > 2 ? (double)(ts[steps - 1] - ts[1]) / (steps - 2) / 100.0 : 0.0;
      printf("%-64s steps %2d: %8.1f us total %7.2f us/step (steady %6.2f)  err %u bad-words %u timeouts %u\n", c.name, steps,
             ms * 1e3, ms * 1e3 / steps, steady, e[0], e[1], e[2]);
      fflush(stdout);
    }
  }
  return 0;
}
