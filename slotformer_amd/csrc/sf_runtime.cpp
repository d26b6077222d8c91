// Runtime bits of libslotformer_hip.so: error string, version, and the optional per-kernel-class
// HIP-event timer used by bench.py for the roofline object (events are recorded on the stream the
// kernel is launched on; nothing is recorded while the stream is being captured into a hipGraph).
#include <stdlib.h>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "sf_internal.h"

thread_local char sf_err_buf[512] = "";

namespace {
struct Rec {
  hipEvent_t a, b;
  double work;
};
std::mutex g_mu;
unsigned g_mask = 0;
int g_every = 1;                       // sampling: every g_every-th launch of a class is bracketed
unsigned g_seen[SF_K_NUM] = {0};
std::vector<Rec> g_recs[SF_K_NUM];
thread_local hipEvent_t t_pending = nullptr;
thread_local int t_suppress = 0;
std::mutex g_stream_mu;
std::map<void*, int> g_stream_cus;   // CU count of the streams made by sf_stream_create_cu_mask
}  // namespace

int sf_dbg(const char* key) {
  static const std::map<std::string, int> vals = [] {
    std::map<std::string, int> m;
    const char* e = getenv("SF_DBG");
    std::string s = e ? e : "";
    size_t i = 0;
    while (i < s.size()) {
      size_t j = s.find(',', i);
      if (j == std::string::npos) j = s.size();
      std::string tok = s.substr(i, j - i);
      size_t q = tok.find('=');
      if (!tok.empty()) m[q == std::string::npos ? tok : tok.substr(0, q)] = q == std::string::npos ? 1 : atoi(tok.c_str() + q + 1);
      i = j + 1;
    }
    return m;
  }();
  auto it = vals.find(key);
  return it == vals.end() ? 0 : it->second;
}

SfThreadOpts& sf_thread_opts() {
  static thread_local SfThreadOpts o;
  return o;
}

int sf_ensure_dyn_lds(const void* kernel, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return sf_set_err((int)e, hipGetErrorString(e), __FILE__, __LINE__);
  std::lock_guard<std::mutex> lk(mu);
  auto it = done.find({kernel, dev});
  if (it != done.end() && it->second >= bytes) return 0;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return sf_set_err((int)e, hipGetErrorString(e), __FILE__, __LINE__);
  done[{kernel, dev}] = bytes;
  return 0;
}

// nested suppression of the class timer on the calling thread (used for launches that must not be mixed into a
// per-class average, e.g. convolutions of another batch computed on a different CU partition)
void sf_prof_suppress(int on) { t_suppress += on ? 1 : -1; }

void sf_prof_begin(int cls, hipStream_t st, double work) {
  if (cls < 0 || cls >= SF_K_NUM || !(g_mask & (1u << cls)) || t_suppress > 0) return;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if ((g_seen[cls]++ % (unsigned)g_every) != 0) return;
  }
  Rec r;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  r.work = work;
  hipEventRecord(r.a, st);
  t_pending = r.b;
  std::lock_guard<std::mutex> lk(g_mu);
  g_recs[cls].push_back(r);
}

void sf_prof_end(int cls, hipStream_t st) {
  if (!t_pending) return;
  hipEventRecord(t_pending, st);
  t_pending = nullptr;
}

extern "C" {

int sf_version(void) { return 102; }

// A stream restricted to a subset of the compute units (bit i of cu_mask = CU i of the device; n_words 32-bit
// words).  Used to partition the 256 CUs between the throughput-bound encode and the latency-bound rollout chain
// so that short rollout kernels never queue behind long convolution workgroups (batch pipelining).
int sf_stream_create_cu_mask(void** stream_out, const unsigned int* cu_mask, int n_words) {
  if (!stream_out || !cu_mask || n_words <= 0) return sf_set_err(-1, "invalid argument: sf_stream_create_cu_mask", __FILE__, __LINE__);
  hipStream_t st = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, (const uint32_t*)cu_mask);
  if (e != hipSuccess) return sf_set_err((int)e, hipGetErrorString(e), __FILE__, __LINE__);
  *stream_out = (void*)st;
  {
    // the CU count of the stream: persistent kernels size their grid by it (sf_stream_cus; conv_ws.hip)
    int c = 0;
    for (int i = 0; i < n_words; ++i) c += __builtin_popcount(cu_mask[i]);
    std::lock_guard<std::mutex> lk(g_stream_mu);
    g_stream_cus[(void*)st] = c;
  }
  return 0;
}

// CUs a launch on `stream` may occupy: the popcount of the mask of a stream made by sf_stream_create_cu_mask, else the whole device
int sf_stream_cus(void* stream) {
  static std::map<int, int> dev_cus;   // CU count per device (a host thread may drive several GPUs), under g_stream_mu
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::lock_guard<std::mutex> lk(g_stream_mu);
  auto it = g_stream_cus.find(stream);
  if (it != g_stream_cus.end() && it->second > 0) return it->second;
  auto dc = dev_cus.find(dev);
  if (dc != dev_cus.end()) return dc->second;
  int n = 0;
  if (!(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)) n = 256;
  dev_cus[dev] = n;
  return n;
}

// 0 for a stream the library knows nothing about (sf_stream_cus answers the whole device for those)
int sf_stream_cus_known(void* stream) {
  std::lock_guard<std::mutex> lk(g_stream_mu);
  auto it = g_stream_cus.find(stream);
  return it != g_stream_cus.end() ? it->second : 0;
}

// Tell the library how many CUs the launches issued on `stream` will run on when that is not the stream's own mask: a stream that CAPTURES a graph
// replayed on a CU-masked stream (the pipeline's encode graphs).  cus <= 0 forgets the entry.
int sf_stream_set_cus(void* stream, int cus) {
  std::lock_guard<std::mutex> lk(g_stream_mu);
  if (cus > 0) g_stream_cus[stream] = cus;
  else g_stream_cus.erase(stream);
  return 0;
}

int sf_stream_destroy(void* stream) {
  if (!stream) return 0;
  {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    g_stream_cus.erase(stream);
  }
  hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) return sf_set_err((int)e, hipGetErrorString(e), __FILE__, __LINE__);
  return 0;
}
const char* sf_last_error_string(void) { return sf_err_buf; }

int sf_profile_enable(int class_mask) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_mask = (unsigned)class_mask;
  return 0;
}

// bracket only every `every`-th launch of an enabled class (1: all).  Two event records around a launch cost the stream a few
// microseconds; sampling keeps a live measurement from slowing the stream it measures.
int sf_profile_sample(int every) {
  SF_REQUIRE(every >= 1, "sf_profile_sample: every >= 1");
  std::lock_guard<std::mutex> lk(g_mu);
  g_every = every;
  for (int c = 0; c < SF_K_NUM; ++c) g_seen[c] = 0;
  return 0;
}

// Sum of elapsed ms / launches / caller-declared algorithmic work of one kernel class since the
// last read; synchronises on the recorded events and clears them.
int sf_profile_read(int cls, double* total_ms, long long* launches, double* work) {
  SF_REQUIRE(cls >= 0 && cls < SF_K_NUM && total_ms && launches && work, "bad profile class");
  std::vector<Rec> recs;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    recs.swap(g_recs[cls]);
  }
  double ms = 0, w = 0;
  for (auto& r : recs) {
    float e = 0.f;
    hipEventSynchronize(r.b);
    if (hipEventElapsedTime(&e, r.a, r.b) == hipSuccess) ms += e;
    w += r.work;
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  *total_ms = ms;
  *launches = (long long)recs.size();
  *work = w;
  return 0;
}
}
