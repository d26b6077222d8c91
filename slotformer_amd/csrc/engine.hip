// Whole-path engines: SAVi/STEVE slot extraction and SlotFormer autoregressive rollout.
// Host-side C++ that sequences the HIP kernels on one stream (hipGraph-capturable: no
// allocation, no synchronisation, no host<->device copies).
//
//   sf_savi_encode_f32 : StoSAVi.encode / STEVE.encode  (savi.py:379-416, steve.py:198-240)
//   sf_rollout_f32     : SlotRollouter.forward / SingleStepSlotRollouter.forward
//                        (slotformer.py:85-126, single_step_slotformer.py:49-90)
//
// The encoder works one time step at a time (all B frames of step t): CNN -> per-pixel MLP ->
// K/V -> Slot-Attention iterations, so every intermediate of a step stays resident in the
// 256 MB Infinity Cache and memory is O(1) in T (the reference's OOM-probing temporal chunking,
// savi.py:431-463, is unnecessary; results are identical for any chunking).
// The rollout keeps all slots in one [B, T_total, N, C] buffer: the Transformer window of step
// s is a strided view of it (no torch.cat, slotformer.py:124), and predictions are written in
// place by the out_proj GEMM epilogue.
#include "../../include/slotformer_hip.h"
#include "sf_internal.h"
#include <map>
#include <vector>
#include "layer_fused.h"
#include "slot_chain.h"
#include <stdlib.h>

namespace {

// set by sf_rollout_bf16 for the duration of its call: every contraction on the generic GEMM core (which honours precision
// mode 2 = single-pass bf16), none on the split-bf16-only fused kernels
thread_local bool t_plain_gemms = false;

struct Bump {
  char* p;
  size_t left;
  bool ok = true;
  float* take(size_t nfloat) {
    size_t bytes = ((nfloat * sizeof(float)) + 255) & ~(size_t)255;
    if (bytes > left) {
      ok = false;
      return nullptr;
    }
    float* r = (float*)p;
    p += bytes;
    left -= bytes;
    return r;
  }
};

inline size_t pad256(size_t nfloat) { return ((nfloat * sizeof(float)) + 255) & ~(size_t)255; }

struct TfmWs {
  float *x2, *qkv, *att, *hid, *y;
};

size_t tfm_ws_bytes(int M, int d, int ffn) {
  return pad256((size_t)M * d) * 3 + pad256((size_t)M * 3 * d) + pad256((size_t)M * ffn);
}

bool tfm_ws_take(Bump& bp, TfmWs& w, int M, int d, int ffn) {
  w.x2 = bp.take((size_t)M * d);
  w.qkv = bp.take((size_t)M * 3 * d);
  w.att = bp.take((size_t)M * d);
  w.hid = bp.take((size_t)M * ffn);
  w.y = bp.take((size_t)M * d);
  return bp.ok;
}

// One nn.TransformerEncoderLayer on x [B*L, d] (in place unless Lq < L).
// Lq < L (norm_first only): only the last Lq rows of every sequence are produced, into ws.y
// ([B*Lq, d]) -- used for the final layer of a rollout step, whose other rows are never read
// (slotformer.py:121).  Returns the output pointer through *out.
// xp / counters non-NULL (pre-LN, d = 256, ffn = 1024, packed FFN weights): the FFN half runs as ONE launch of the fused kernel
// of layer_fused.hip on the finished attention rows (xp [4][B*Lq][d] chunk-partial scratch, counters zeroed by the caller).
int tfm_layer(const sf_tfm_layer& w, float* x, TfmWs& ws, int B, int L, int Lq, int d, int heads, int ffn,
              int norm_first, hipStream_t st, float** out, float* xp = nullptr, int* counters = nullptr) {
  const int M = B * L, Mq = B * Lq;
  const float eps = 1e-5f;
  const SfRowMap rd = sf_rows(d);
  if (norm_first) {
    // split-bf16 mode: LN1 + per-head q|k|v projection + attention in one launch (attn_fused.hip)
    int fused = 1;
    if (sf_get_precision() >= 1 && !t_plain_gemms)
      fused = sf_qkv_attn_ex(x, w.norm1_g, w.norm1_b, eps, w.in_proj_w, w.in_proj_b, ws.att, B, L, Lq, d, heads, st);
    if (fused < 0 || fused > 1) return fused;
    if (fused == 1) {
      SF_TRY(sf_linear_ex(x, rd, w.in_proj_w, w.in_proj_b, w.norm1_g, w.norm1_b, eps, nullptr, rd, 0, ws.qkv,
                          sf_rows(3 * d), M, 3 * d, d, 0, st));
      SF_TRY(sf_mha_ex(ws.qkv, ws.att, B, L, Lq, d, heads, st));
    }
    // residual rows: last Lq rows of each sequence of x
    const SfRowMap xr = (Lq == L) ? rd : sf_rows_batched(d, Lq, (long long)L * d, (long long)(L - Lq) * d);
    SF_TRY(sf_linear_ex(ws.att, rd, w.out_proj_w, w.out_proj_b, nullptr, nullptr, eps, x, xr, 0, ws.x2, rd, Mq,
                        d, d, 0, st));
    float* dst = (Lq == L) ? x : ws.y;
    if (xp && counters) {
      SF_TRY(sf_ffn_partial_ex(ws.x2, (long long)Mq * d, w, eps, xp, (long long)Mq * d, dst, counters, Mq, ffn, st, 1));
    } else {
      SF_TRY(sf_linear_ex(ws.x2, rd, w.lin1_w, w.lin1_b, w.norm2_g, w.norm2_b, eps, nullptr, rd, 0, ws.hid,
                          sf_rows(ffn), Mq, ffn, d, 1, st));
      SF_TRY(sf_linear_ex(ws.hid, sf_rows(ffn), w.lin2_w, w.lin2_b, nullptr, nullptr, eps, ws.x2, rd, 0, dst, rd,
                          Mq, d, ffn, 0, st));
    }
    *out = dst;
  } else {
    if (Lq != L) return sf_set_err(-1, "row pruning requires norm_first", __FILE__, __LINE__);
    int fused = 1;
    if (sf_get_precision() >= 1 && !t_plain_gemms)
      fused = sf_qkv_attn_ex(x, nullptr, nullptr, eps, w.in_proj_w, w.in_proj_b, ws.att, B, L, L, d, heads, st);
    if (fused < 0 || fused > 1) return fused;
    if (fused == 1) {
      SF_TRY(sf_linear_ex(x, rd, w.in_proj_w, w.in_proj_b, nullptr, nullptr, eps, nullptr, rd, 0, ws.qkv,
                          sf_rows(3 * d), M, 3 * d, d, 0, st));
      SF_TRY(sf_mha_ex(ws.qkv, ws.att, B, L, L, d, heads, st));
    }
    SF_TRY(sf_linear_ex(ws.att, rd, w.out_proj_w, w.out_proj_b, nullptr, nullptr, eps, x, rd, 0, ws.y, rd, M, d,
                        d, 0, st));
    SF_TRY(sf_layernorm_ex(ws.y, rd, w.norm1_g, w.norm1_b, ws.x2, rd, M, d, eps, st));
    SF_TRY(sf_linear_ex(ws.x2, rd, w.lin1_w, w.lin1_b, nullptr, nullptr, eps, nullptr, rd, 0, ws.hid,
                        sf_rows(ffn), M, ffn, d, 1, st));
    SF_TRY(sf_linear_ex(ws.hid, sf_rows(ffn), w.lin2_w, w.lin2_b, nullptr, nullptr, eps, ws.x2, rd, 0, ws.y, rd,
                        M, d, ffn, 0, st));
    SF_TRY(sf_layernorm_ex(ws.y, rd, w.norm2_g, w.norm2_b, x, rd, M, d, eps, st));
    *out = x;
  }
  return 0;
}

int check_layers(const sf_tfm_layer* l, int n) {
  if (n > 0 && !l) return -1;
  for (int i = 0; i < n; ++i) {
    const sf_tfm_layer& w = l[i];
    if (!w.norm1_g || !w.norm1_b || !w.in_proj_w || !w.in_proj_b || !w.out_proj_w || !w.out_proj_b ||
        !w.norm2_g || !w.norm2_b || !w.lin1_w || !w.lin1_b || !w.lin2_w || !w.lin2_b)
      return -1;
  }
  return 0;
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------
size_t sf_rollout_workspace_bytes(const sf_rollouter* m, int B) {
  if (!m || B <= 0) return 0;
  const int Lmax = m->window_len * m->num_slots;
  const size_t M = (size_t)B * Lmax;
  // + head partials [8][M][d], hidden-chunk partials [4][M][d], two layer-output buffers, the tile counters of the
  // two-launch layer and the ring of cached in-projections [B][window_len + 1][N][d]
  return pad256(M * m->d_model) + tfm_ws_bytes((int)M, m->d_model, m->ffn_dim) + pad256(8 * M * m->d_model) +
         pad256(4 * M * m->d_model) + 2 * pad256(M * m->d_model) +
         pad256((size_t)B * (m->window_len + 1) * m->num_slots * m->d_model) + 4096 + 4096 + 4096 +
         pad256(sf_attn_rows_plane_bytes(B) / 4);   // q / k / v^T fragment planes of the row-tile attention form
}

extern "C" int sf_get_seam_fused(void);
extern "C" int sf_get_layer_tok(void);

// a[0..n) = b[0..n) = 0 (n a multiple of 4, both 16-byte aligned)
// One wave kept busy for a given time (wall_clock64: the constant 100 MHz counter).  The pipeline launches two of them on two streams to
// find out whether the streams share a hardware queue (then they run one after the other): pipeline.py _pick_free_streams.
__global__ void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
extern "C" int sf_debug_spin(int us, void* stream) {
  SF_REQUIRE(us > 0 && us <= 100000, "sf_debug_spin: 1..100000 microseconds");
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)us * 100);
  SF_CHECK_LAUNCH();
  return 0;
}

// One wave that measures the shader clock: s_memtime (shader cycles) against the constant 100 MHz counter over `ticks` of the latter; out[0] = cycles,
// out[1] = 10 ns ticks.  tools/clock_probe.py launches it on an idle stream while the pipeline runs: the clock the chip sustains under that load.
__global__ void clock_probe_kernel(long long ticks, long long* out) {
  const long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
  long long w1 = w0;
  while (w1 - w0 < ticks) {
    __builtin_amdgcn_s_sleep(8);
    w1 = wall_clock64();
  }
  const long long c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = w1 - w0;
  }
}
extern "C" int sf_debug_clock_probe(int us, long long* out2, void* stream) {
  SF_REQUIRE(us > 0 && us <= 100000 && out2, "sf_debug_clock_probe: 1..100000 microseconds, a device buffer of two int64");
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)us * 100, out2);
  SF_CHECK_LAUNCH();
  return 0;
}

// clears a[0..n) and b[0..n): float4 stores where both pointers are 16-byte aligned and four elements remain, scalars otherwise
// (launch with ceil(n / 4) threads)
__global__ void zero_f32_kernel(float* a, float* b, long long n) {
  const long long i = 4 * ((long long)blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  const bool vec = i + 4 <= n && ((((unsigned long long)(a + i)) | ((unsigned long long)(b + i))) & 15ull) == 0;
  if (vec) {
    *(float4*)(a + i) = float4{0.f, 0.f, 0.f, 0.f};
    *(float4*)(b + i) = float4{0.f, 0.f, 0.f, 0.f};
  } else {
    for (long long j = i; j < n && j < i + 4; ++j) {
      a[j] = 0.f;
      b[j] = 0.f;
    }
  }
}

// block 0 clears a[0..1023], block 1 clears b[0..1023] (b may be NULL)
__global__ void zero_words_kernel(unsigned* a, unsigned* b) {
  unsigned* p = blockIdx.x == 0 ? a : b;
  if (p) p[threadIdx.x] = 0u;
}

// A seam launch hands rows over INSIDE a grid (consumers poll producers with a bounded wait): every workgroup of it must be resident at
// once, one per CU.  160 fit the whole chip with room for neighbours; a caller whose stream carries a CU mask says how many CUs that is
// (sf_rollout_opts.cus_available) and the seam is used only when the grid fits them -- otherwise the two launches it replaces.
static int seam_capacity() {
  const int cus = sf_thread_opts().cus;
  return cus > 0 && cus < 160 ? cus : 160;
}

int sf_rollout_f32(const sf_rollouter* m, float* slots, int B, int T_total, int pred_len, void* ws,
                   size_t ws_bytes, void* stream) {
  SF_REQUIRE(m && slots && ws, "null pointer");
  SF_REQUIRE(B >= 1 && pred_len >= 0, "bad batch / pred_len");
  SF_REQUIRE(m->num_slots >= 1 && m->slot_size > 0 && (m->slot_size % 4) == 0 && m->d_model > 0 &&
                 (m->d_model % 4) == 0 && m->ffn_dim > 0 && (m->ffn_dim % 4) == 0 && m->num_layers >= 1 &&
                 m->window_len >= 1, "bad rollouter shape");
  SF_REQUIRE(m->in_proj_w && m->in_proj_b && m->out_proj_w && m->out_proj_b && m->pe_tok, "null weight");
  SF_REQUIRE(check_layers(m->layers, m->num_layers) == 0, "null transformer-layer weight");
  const int n_in = m->single_step ? 1 : m->window_len;
  SF_REQUIRE(T_total >= n_in + pred_len, "slots buffer shorter than burn-in + pred_len");
  SF_REQUIRE(ws_bytes >= sf_rollout_workspace_bytes(m, B), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int N = m->num_slots, C = m->slot_size, d = m->d_model, W = m->window_len;
  const int Lmax = W * N;
  Bump bp{(char*)ws, ws_bytes};
  float* x = bp.take((size_t)B * Lmax * d);
  TfmWs tw;
  if (!x || !tfm_ws_take(bp, tw, B * Lmax, d, m->ffn_dim))
    return sf_set_err(-1, "workspace too small", __FILE__, __LINE__);
  float* apb = bp.take((size_t)8 * B * Lmax * d);   // head-pair partials use the first 4
  float* xpb = bp.take((size_t)4 * B * Lmax * d);
  float* xa = bp.take((size_t)B * Lmax * d);
  float* xb2 = bp.take((size_t)B * Lmax * d);
  int* counters = (int*)bp.take(1024);
  unsigned* seam_flags = (unsigned*)bp.take(1024);   // per-tile epochs of the seam launches + an error word
  const int RF = W + 1;   // frames in the projection ring: the window being read + the frame being written
  float* ring = bp.take((size_t)B * RF * N * d);
  float* planes = bp.take(sf_attn_rows_plane_bytes(B) / 4);   // attn_rows.hip: q, k, v^T of every (video, head) as fragment planes
  if (!apb || !xpb || !xa || !xb2 || !counters || !seam_flags || !ring || !planes)
    return sf_set_err(-1, "workspace too small", __FILE__, __LINE__);
  // two-launch layers (layer_fused.hip): split-bf16 mode, pre-LN, d=256 / 8 heads / ffn 1024, window <= 64 tokens
  bool packed = true;
  for (int l = 0; l < m->num_layers; ++l)
    packed = packed && m->layers[l].lin1_packed && m->layers[l].lin2_packed && m->layers[l].attn_in_packed && m->layers[l].attn_out_packed;
  const bool fused_layers = packed && !t_plain_gemms && sf_get_precision() >= 1 && m->norm_first &&
                            sf_layer_fused_ok(d, m->num_heads, m->ffn_dim, Lmax);
  // step boundary in one launch (out-proj of step s + in-proj of the new frame for step s+1) with cached in-projections
  const bool ring_mode = fused_layers && m->in_proj_packed && m->out_proj_packed && sf_step_boundary_ok(d, C);
  const bool boundary_fused = ring_mode;
  // seam launches (layer_fused.hip): the last-layer FFN + boundary of step s and the layer-0 attention of step s+1 in one
  // grid; needs every workgroup of it co-resident at one per CU -- 160 fit the 168-CU rollout partition
  const int seam_opt = sf_thread_opts().seam;
  // row-tile form of the attention block (per-call option attn_qkv_rows = 128, attn_rows.hip): q|k|v projection on 128-row tiles
  // of the batch + one core / out-projection workgroup per video; finished rows like the all-heads form
  bool tok_packed = fused_layers && m->num_layers >= 2;
  for (int l = 0; l + 1 < m->num_layers; ++l) tok_packed = tok_packed && m->layers[l].tok_packed;
  {
    const int lt = sf_thread_opts().layer_tok;
    tok_packed = tok_packed && (lt > 0 || (lt == 0 && sf_get_layer_tok() != 0));
    for (int nf = m->single_step ? 1 : W; nf <= W && tok_packed; ++nf) tok_packed = sf_layer_tok_ok(nf * N);
  }
  const bool tok_layers = tok_packed;
  const bool attn_rows = fused_layers && (sf_thread_opts().attn_rows == 128 || tok_layers);
  const bool seam = boundary_fused && (seam_opt >= 0 ? seam_opt != 0 : sf_get_seam_fused() != 0) && sf_seam_blocks(B, N) <= seam_capacity() &&
                    sf_thread_opts().attn_heads != 8 && !attn_rows;
  // layers 0 .. n-2 leave their output as four FFN chunk partials that the next attention sums while loading (the last layer's
  // FFN sums them itself: its last-arriving workgroup)
  const bool parts_mode = ring_mode;
  // throughput form of the attention block (per-call option attn_heads_per_wg = 8): one workgroup per video runs all 8 heads
  // and writes finished rows; the FFN behind it reads one row instead of four head-pair partials (layer_fused.hip)
  const bool all_heads = fused_layers && (sf_thread_opts().attn_heads == 8 || attn_rows);
  // row-tile form of the FFN block behind finished attention rows (per-call option ffn_tile, ffn_tile.hip): finished rows out
  const bool ffn_tile = all_heads && sf_thread_opts().ffn_tile >= 1;
  // ... fused with LN1 + q|k|v of the NEXT layer on the same tiles (ffn_tile = 2, behind the row-tile attention form): the next attention
  // block is then its core launch alone
  const bool ffn_qkv = ffn_tile && attn_rows && sf_thread_opts().ffn_tile == 2;
  const int np = all_heads ? 1 : 4;
  // token-stationary whole-layer launches for the layers before the last (per-call option layer_tok, layer_tok.hip): finished rows in, finished rows
  // out; the (row-pruned) last layer runs in the row-tile forms behind them
  const bool layer_tok = tok_layers;
  if (ring_mode) {
    // in-projection (without PE) of the burn-in frames -> ring slots 0 .. n_in-1
    SF_TRY(sf_ring_init_ex(m->out_proj_packed, m->out_proj_b, m->in_proj_packed, m->in_proj_b, slots,
                           (long long)T_total * N * C, n_in, ring, RF, N, B, st));
  }
  // windows of 65..128 tokens (the reference's Physion window: 15 frames x 6 slots): the attention kernels of layer_fused.hip
  // hold two token blocks, so the layers run as LN1 + q|k|v + attention in one launch (attn_fused.hip, four token blocks), the
  // out-projection GEMM, and the fused FFN kernel on its finished rows
  const bool long_ffn = !fused_layers && packed && !t_plain_gemms && sf_get_precision() >= 1 && m->norm_first &&
                        sf_layer_fused_ok(d, m->num_heads, m->ffn_dim, 1) && Lmax > 64 && sf_ffn_tiles(B * Lmax) <= 1024;
  // ... and the layers before the last of such a window as token-stationary launches (per-call option layer_tok; layer_tok.hip: one video per workgroup)
  bool long_tok_packed = long_ffn && m->num_layers >= 2 && Lmax <= 96 && !m->single_step && sf_layer_tok_ok(Lmax);
  for (int l = 0; l + 1 < m->num_layers; ++l) long_tok_packed = long_tok_packed && m->layers[l].tok_packed;
  const bool long_tok = long_tok_packed && (sf_thread_opts().layer_tok > 0 || (sf_thread_opts().layer_tok == 0 && sf_get_layer_tok() != 0));
  if (fused_layers || long_ffn) {
    SF_REQUIRE(sf_ffn_tiles(B * Lmax) <= 1024, "batch too large for the fused-layer tile counters");
    // zeroed by a KERNEL, not hipMemsetAsync: the rollout is captured into hipGraphs, and the memset nodes of a graph were seen
    // to stop clearing these words after an OLDER graph exec had been destroyed (stale seam epochs -> consumers read ring rows
    // before they were written; found by tests/test_pipeline_gpu.py when a second pipeline followed a first in one process)
    hipLaunchKernelGGL(zero_words_kernel, dim3(2), dim3(1024), 0, st, (unsigned*)counters, seam ? seam_flags : nullptr);
    SF_CHECK_LAUNCH();
    if (attn_rows) {
      // key positions >= L of the v^T planes meet probability 0 in the PV product: they must hold finite values
      const long long nfl = (long long)(sf_attn_rows_plane_bytes(B) / 4), half = nfl / 2;
      hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)(((half + 3) / 4 + 255) / 256)), dim3(256), 0, st, planes, planes + half, half);
      SF_CHECK_LAUNCH();
    }
  }
  const long long bs = (long long)T_total * N * C;
  auto window = [&](int s, int& nf, int& f0) {   // frames of the Transformer window of step s
    if (!m->single_step) {
      nf = W;
      f0 = s;
    } else {
      const int have = s + 1;
      nf = have < W ? have : W;
      f0 = have - nf;
    }
  };
  float* apb2 = apb + (size_t)4 * B * Lmax * d;   // second set of head-pair partials (the seam's attention writes there)
  float* ap_l0 = apb;                              // where the layer-0 attention of the CURRENT step put its partials
  bool attn0_done = false;                         // ... and whether it already ran (inside the previous seam launch)
  for (int s = 0; s < pred_len; ++s) {
    int nf, f0;
    window(s, nf, f0);
    const int L = nf * N, M = B * L, pe_off = (W - nf) * N;
    if (ring_mode) {
      const float* cin = nullptr;
      bool parts_in = false;   // the current layer's input is still the previous layer's four FFN chunk partials in xpb
      float* parked = nullptr; // ffn_qkv: the buffer the previous layer's fused launch parked this layer's residual rows in (planes written)
      for (int l = 0; l < m->num_layers; ++l) {
        const bool lastl = (l == m->num_layers - 1);
        const int Lq = lastl ? N : L;   // last layer: only the newest frame's rows are read (slotformer.py:121)
        const long long pst = (long long)B * Lq * d;
        float* xo = (cin == xa) ? xb2 : xa;
        float* apl = (l == 0) ? ap_l0 : apb;
        if (layer_tok && !lastl) {
          // all layers before the last in ONE launch (up to eight; the rows stay in registers between them)
          const int nlt = (m->num_layers - 1 - l) < 8 ? (m->num_layers - 1 - l) : 8;
          SF_TRY(sf_layer_tok_ex(l == 0 ? 1 : 0, cin, ring, RF, N, f0, m->pe_tok + (long long)pe_off * d, m->layers + l, nlt, 1e-5f, xo, B, L, st));
          cin = xo;
          parts_in = false;
          if (l == 0) ap_l0 = apb;
          attn0_done = false;
          l += nlt - 1;
          continue;
        }
        if (parked) {
          apl = parked;
          SF_TRY(sf_attn_core_ex(m->layers[l], apl, planes, B, L, Lq, st));
        } else if (attn_rows) {
          if (l == 0)
            SF_TRY(sf_attn_rows_ex(2, ring, (long long)RF * N * d, 0, m->pe_tok + (long long)pe_off * d, f0, RF, N, m->layers[l], 1e-5f, apl,
                                   planes, B, L, Lq, st));
          else if (parts_in)
            SF_TRY(sf_attn_rows_ex(1, xpb, (long long)L * d, (long long)B * L * d, nullptr, 0, 1, 1, m->layers[l], 1e-5f, apl, planes, B, L,
                                   Lq, st));
          else
            SF_TRY(sf_attn_rows_ex(0, cin, (long long)L * d, 0, nullptr, 0, 1, 1, m->layers[l], 1e-5f, apl, planes, B, L, Lq, st));
        } else if (all_heads) {
          if (l == 0)
            SF_TRY(sf_attn_all_ring_ex(ring, RF, N, f0, m->pe_tok + (long long)pe_off * d, m->layers[l], 1e-5f, apl, B, L, Lq, st));
          else if (parts_in)
            SF_TRY(sf_attn_all_parts_ex(xpb, (long long)B * L * d, m->layers[l], 1e-5f, apl, B, L, Lq, st));
          else
            SF_TRY(sf_attn_all_ex(cin, m->layers[l], 1e-5f, apl, B, L, Lq, st));
        } else if (l == 0) {
          if (!attn0_done)
            SF_TRY(sf_attn_oproj_ring_ex(ring, RF, N, f0, m->pe_tok + (long long)pe_off * d, m->layers[l], 1e-5f, apl, pst, B, L,
                                         Lq, st));
        } else if (parts_in) {
          SF_TRY(sf_attn_oproj_parts_ex(xpb, (long long)B * L * d, m->layers[l], 1e-5f, apl, pst, B, L, Lq, st));
        } else {
          SF_TRY(sf_attn_oproj_ex(cin, m->layers[l], 1e-5f, apl, pst, B, L, Lq, st));
        }
        attn0_done = false;
        if (lastl && boundary_fused) {
          // last layer: FFN + step boundary in one launch.  pred = out_proj(last rows) -> frame n_in + s; its
          // in-projection -> the ring   (slotformer.py:121-124, :115)
          int nf1 = 0, f01 = 0;
          if (s + 1 < pred_len) window(s + 1, nf1, f01);
          if (seam && s + 1 < pred_len && sf_seam_window_ok(nf1 * N, N)) {
            const int L1 = nf1 * N, Lq1 = (m->num_layers == 1) ? N : L1;
            float* ap_next = (apl == apb) ? apb2 : apb;
            SF_TRY(sf_seam_ex(apl, pst, m->layers[l], 1e-5f, xpb, pst, counters, m->ffn_dim, m->out_proj_packed, m->out_proj_b,
                              m->in_proj_packed, m->in_proj_b, slots, bs, n_in + s, ring, RF, N, B, m->layers[0], f01,
                              m->pe_tok + (long long)((W - nf1) * N) * d, ap_next, (long long)B * Lq1 * d, L1, Lq1, seam_flags,
                              (unsigned)(s + 1), st));
            ap_l0 = ap_next;
            attn0_done = true;
          } else {
            SF_TRY(sf_ffn_boundary_ex(apl, pst, m->layers[l], 1e-5f, xpb, pst, counters, m->ffn_dim, m->out_proj_packed,
                                      m->out_proj_b, m->in_proj_packed, m->in_proj_b, slots, bs, n_in + s, ring, RF, N, B, st, np));
            ap_l0 = apb;
          }
          cin = nullptr;
        } else if (ffn_qkv && !lastl) {
          float* park = (apl == apb) ? apb2 : apb;
          SF_TRY(sf_ffn_qkv_tile_ex(apl, m->layers[l], m->layers[l + 1], 1e-5f, park, planes, B, L, (l + 1 == m->num_layers - 1) ? N : L,
                                    m->ffn_dim, st));
          parked = park;
          cin = nullptr;
          parts_in = false;
          if (l == 0) ap_l0 = apb;
        } else if (ffn_tile && !lastl) {
          SF_TRY(sf_ffn_tile_ex(apl, m->layers[l], 1e-5f, xo, B * Lq, m->ffn_dim, st));
          cin = xo;
          parts_in = false;
          if (l == 0) ap_l0 = apb;
        } else if (parts_mode && !lastl) {
          // the chunk partials are the layer output: the next attention sums them
          SF_TRY(sf_ffn_parts_ex(apl, pst, m->layers[l], 1e-5f, xpb, pst, B * Lq, m->ffn_dim, st, np));
          parts_in = true;
          cin = nullptr;
          if (l == 0) ap_l0 = apb;
        } else {
          SF_TRY(sf_ffn_partial_ex(apl, pst, m->layers[l], 1e-5f, xpb, pst, xo, counters, B * Lq, m->ffn_dim, st, np));
          cin = xo;
          parts_in = false;
          if (l == 0) ap_l0 = apb;
        }
      }
      if (cin != nullptr)
        SF_TRY(sf_step_boundary_ex(cin, m->out_proj_packed, m->out_proj_b, m->in_proj_packed, m->in_proj_b, slots, bs, n_in + s,
                                   ring, RF, N, B, st));
      continue;
    }
    // x = in_proj(window) + pe   (slotformer.py:115-117; single_step_slotformer.py:79-81)
    SfRowMap pmap = sf_rows(d);
    pmap.base = (long long)pe_off * d;
    SF_TRY(sf_linear_ex(slots, sf_rows_batched(C, L, bs, (long long)f0 * N * C), m->in_proj_w, m->in_proj_b,
                        nullptr, nullptr, 0.f, m->pe_tok, pmap, L, x, sf_rows(d), M, d, C, 0, st));
    if (fused_layers) {
      // every layer is two launches: attention + out-proj head partials, then the FFN (which also finishes the sums)
      const float* cin = x;
      bool parts_in = false;
      float* parked = nullptr;   // (as in the ring path)
      float* apn = apb;          // attention output / FFN input of the current layer
      for (int l = 0; l < m->num_layers; ++l) {
        const bool lastl = (l == m->num_layers - 1);
        const int Lq = lastl ? N : L;
        const long long pst = (long long)B * Lq * d;
        float* xo = (cin == xa) ? xb2 : xa;
        if (layer_tok && !lastl) {
          const int nlt = (m->num_layers - 1 - l) < 8 ? (m->num_layers - 1 - l) : 8;
          SF_TRY(sf_layer_tok_ex(0, cin, nullptr, 1, 1, 0, nullptr, m->layers + l, nlt, 1e-5f, xo, B, L, st));
          cin = xo;
          parts_in = false;
          l += nlt - 1;
          continue;
        }
        if (parked) {
          apn = parked;
          SF_TRY(sf_attn_core_ex(m->layers[l], apn, planes, B, L, Lq, st));
        } else if (attn_rows) {
          apn = apb;
          if (parts_in)
            SF_TRY(sf_attn_rows_ex(1, xpb, (long long)L * d, (long long)B * L * d, nullptr, 0, 1, 1, m->layers[l], 1e-5f, apb, planes, B, L,
                                   Lq, st));
          else
            SF_TRY(sf_attn_rows_ex(0, cin, (long long)L * d, 0, nullptr, 0, 1, 1, m->layers[l], 1e-5f, apb, planes, B, L, Lq, st));
        } else if (all_heads) {
          if (parts_in)
            SF_TRY(sf_attn_all_parts_ex(xpb, (long long)B * L * d, m->layers[l], 1e-5f, apb, B, L, Lq, st));
          else
            SF_TRY(sf_attn_all_ex(cin, m->layers[l], 1e-5f, apb, B, L, Lq, st));
        } else if (parts_in) {
          SF_TRY(sf_attn_oproj_parts_ex(xpb, (long long)B * L * d, m->layers[l], 1e-5f, apb, pst, B, L, Lq, st));
        } else {
          SF_TRY(sf_attn_oproj_ex(cin, m->layers[l], 1e-5f, apb, pst, B, L, Lq, st));
        }
        if (!lastl && ffn_qkv) {
          float* park = (apn == apb) ? apb2 : apb;
          SF_TRY(sf_ffn_qkv_tile_ex(apn, m->layers[l], m->layers[l + 1], 1e-5f, park, planes, B, L, (l + 1 == m->num_layers - 1) ? N : L,
                                    m->ffn_dim, st));
          parked = park;
          cin = nullptr;
          parts_in = false;
        } else if (!lastl && ffn_tile) {
          SF_TRY(sf_ffn_tile_ex(apn, m->layers[l], 1e-5f, xo, B * Lq, m->ffn_dim, st));
          cin = xo;
          parts_in = false;
        } else if (!lastl) {
          SF_TRY(sf_ffn_parts_ex(apn, pst, m->layers[l], 1e-5f, xpb, pst, B * Lq, m->ffn_dim, st, np));
          parts_in = true;
        } else {
          SF_TRY(sf_ffn_partial_ex(apn, pst, m->layers[l], 1e-5f, xpb, pst, xo, counters, B * Lq, m->ffn_dim, st, np));
          cin = xo;
          parts_in = false;
        }
      }
      SF_TRY(sf_linear_ex(cin, sf_rows(d), m->out_proj_w, m->out_proj_b, nullptr, nullptr, 0.f, nullptr, sf_rows(C), 0,
                          slots, sf_rows_batched(C, N, bs, (long long)(n_in + s) * N * C), B * N, C, d, 0, st));
      continue;
    }
    float* cur = x;
    int Lc = L;
    int l0 = 0;
    if (long_tok) {
      // windows of 65..96 tokens (the reference's Physion window, slotformer_physion_params.py: 15 frames x 6 slots): the layers before the last as
      // token-stationary launches of one video per workgroup, up to eight layers per launch; the row-pruned last layer in the long-window forms below
      while (l0 + 1 < m->num_layers) {
        const int nlt = (m->num_layers - 1 - l0) < 8 ? (m->num_layers - 1 - l0) : 8;
        float* xo = (cur == xa) ? xb2 : xa;
        SF_TRY(sf_layer_tok_ex(0, cur, nullptr, 1, 1, 0, nullptr, m->layers + l0, nlt, 1e-5f, xo, B, L, st));
        cur = xo;
        l0 += nlt;
      }
    }
    for (int l = l0; l < m->num_layers; ++l) {
      const bool last = (l == m->num_layers - 1);
      const int Lq = (last && m->norm_first) ? N : Lc;
      float* outp = nullptr;
      SF_TRY(tfm_layer(m->layers[l], cur, tw, B, Lc, Lq, d, m->num_heads, m->ffn_dim, m->norm_first, st, &outp,
                       long_ffn ? xpb : nullptr, long_ffn ? counters : nullptr));
      cur = outp;
      Lc = Lq;
    }
    // pred = out_proj(x[:, -N:]) written straight into frame n_in + s   (slotformer.py:121-124)
    const SfRowMap lastmap =
        (Lc == N) ? sf_rows(d) : sf_rows_batched(d, N, (long long)Lc * d, (long long)(Lc - N) * d);
    SF_TRY(sf_linear_ex(cur, lastmap, m->out_proj_w, m->out_proj_b, nullptr, nullptr, 0.f, nullptr, sf_rows(C), 0,
                        slots, sf_rows_batched(C, N, bs, (long long)(n_in + s) * N * C), B * N, C, d, 0, st));
  }
  return 0;
}

// SURVEY.md 8(b2) `sf_rollout_bf16`: the same rollout with every matrix product in SINGLE-PASS bf16 (operands rounded to
// 8 mantissa bits, f32 accumulation; f32 storage, LayerNorm, softmax) -- the arithmetic of the reference's `--fp16` AMP
// (scripts/train.py:84,105) and of BASELINE.json's literal "bf16".  Measured on the 6+50 path of config C2 against the
// reference fixture: ~8e-3 relative (tools/precision_probe.py, tests/test_engine_gpu.py), i.e. OUTSIDE the 1e-3 parity bar --
// which is why sf_rollout_f32 (split-bf16, 1e-5) is the default and the product path.  Same arguments as sf_rollout_f32.
// Seam launches on / off (default: on unless SF_SEAM_FUSED=0).  A seam launch saves a kernel boundary on the critical path of
// ONE rollout chain; its 128 consumer workgroups spin until the 28 producers are done, which is CU time a second chain running
// on the same CUs could use -- the 'pair' pipeline captures its graphs with the seam off.
// Process default of the token-stationary layer launches (sf_rollout_opts.layer_tok == 0): OFF unless SF_LAYER_TOK=1 / sf_set_layer_tok(1)
static int g_layer_tok = -1;
extern "C" int sf_get_layer_tok(void) {
  if (g_layer_tok < 0) {
    const char* e = getenv("SF_LAYER_TOK");
    g_layer_tok = (e && e[0] == '1') ? 1 : 0;   // (off until the round's validation is complete)
  }
  return g_layer_tok;
}
extern "C" int sf_set_layer_tok(int on) {
  g_layer_tok = on ? 1 : 0;
  return 0;
}
static int g_seam_fused = -1;
extern "C" int sf_get_seam_fused(void) {
  if (g_seam_fused < 0) {
    const char* e = getenv("SF_SEAM_FUSED");
    g_seam_fused = !(e && e[0] == '0');
  }
  return g_seam_fused;
}
extern "C" int sf_set_seam_fused(int on) {
  g_seam_fused = on ? 1 : 0;
  return 0;
}

namespace {
// the calling thread's options for the duration of one engine call
struct OptsScope {
  SfThreadOpts saved;
  explicit OptsScope(const SfThreadOpts& o) : saved(sf_thread_opts()) { sf_thread_opts() = o; }
  ~OptsScope() { sf_thread_opts() = saved; }
};
}  // namespace

// sf_rollout_f32 with per-call options (include/slotformer_hip.h, sf_rollout_opts): arithmetic mode, seam launches, rows
// per FFN workgroup, videos per attention workgroup.  The options live in thread-local state for the duration of the call,
// so concurrent calls from other host threads (one per GPU in the reference's drivers, extract_slots.py:128) keep theirs.
int sf_rollout_opts_f32(const sf_rollouter* m, float* slots, int B, int T_total, int pred_len, void* ws, size_t ws_bytes,
                        void* stream, const sf_rollout_opts* opts) {
  if (!opts) return sf_rollout_f32(m, slots, B, T_total, pred_len, ws, ws_bytes, stream);
  SF_REQUIRE(opts->precision >= -1 && opts->precision <= 3, "sf_rollout_opts: precision must be -1 (default), 0, 1, 2 or 3");
  SF_REQUIRE(opts->ffn_rows == 0 || opts->ffn_rows == 32 || opts->ffn_rows == 64 || opts->ffn_rows == 128,
             "sf_rollout_opts: ffn_rows must be 0 (default), 32, 64 or 128");
  SF_REQUIRE(opts->attn_heads_per_wg == 0 || opts->attn_heads_per_wg == 2 || opts->attn_heads_per_wg == 8,
             "sf_rollout_opts: attn_heads_per_wg must be 0 (default), 2 or 8");
  SF_REQUIRE(opts->attn_qkv_rows == 0 || opts->attn_qkv_rows == 128, "sf_rollout_opts: attn_qkv_rows must be 0 (off) or 128");
  SF_REQUIRE(opts->ffn_tile >= 0 && opts->ffn_tile <= 2, "sf_rollout_opts: ffn_tile must be 0, 1 or 2");
  SF_REQUIRE(opts->cus_available >= 0 && opts->cus_available <= 256, "sf_rollout_opts: cus_available must be 0 (whole chip) .. 256");
  SfThreadOpts o = sf_thread_opts();
  if (opts->precision >= 0) o.precision = opts->precision;
  if (opts->seam_fused >= 0) o.seam = opts->seam_fused ? 1 : 0;
  if (opts->ffn_rows > 0) o.ffn_rows = opts->ffn_rows;
  if (opts->attn_heads_per_wg > 0) o.attn_heads = opts->attn_heads_per_wg;
  if (opts->attn_qkv_rows > 0) o.attn_rows = opts->attn_qkv_rows;
  if (opts->ffn_tile > 0) o.ffn_tile = opts->ffn_tile;
  if (opts->cus_available > 0) o.cus = opts->cus_available;
  if (opts->layer_tok != 0) o.layer_tok = opts->layer_tok > 0 ? 1 : -1;
  OptsScope scope(o);
  const bool plain = (o.precision == 2 || o.precision == 3);
  const bool old_plain = t_plain_gemms;
  if (plain) t_plain_gemms = true;
  const int rc = sf_rollout_f32(m, slots, B, T_total, pred_len, ws, ws_bytes, stream);
  t_plain_gemms = old_plain;
  return rc;
}

// 1 when sf_rollout_f32 runs this model's layers as the two fused launches of layer_fused.hip (per-video / per-row kernels whose
// results do not depend on how videos are grouped into batches); 0: the generic GEMM path, whose tile / split-K choice -- and
// with it the summation order -- follows the batch size
int sf_rollout_is_fused(const sf_rollouter* m) {
  if (!m || !m->layers) return 0;
  bool packed = true;
  for (int l = 0; l < m->num_layers; ++l)
    packed = packed && m->layers[l].lin1_packed && m->layers[l].lin2_packed && m->layers[l].attn_in_packed && m->layers[l].attn_out_packed;
  return packed && sf_get_precision() >= 1 && m->norm_first && sf_layer_fused_ok(m->d_model, m->num_heads, m->ffn_dim, m->window_len * m->num_slots);
}

// 1 when sf_rollout_f32 can run this model's layers before the last as token-stationary launches (sf_rollout_opts.layer_tok): the fused-layer path,
// sf_pack_layer_tok_weights fragments on every layer but the last, every window of the rollout inside the kernel's key-block limit
int sf_rollout_tok_ok(const sf_rollouter* m) {
  if (!sf_rollout_is_fused(m) || m->num_layers < 2) return 0;
  for (int l = 0; l + 1 < m->num_layers; ++l)
    if (!m->layers[l].tok_packed) return 0;
  for (int nf = m->single_step ? 1 : m->window_len; nf <= m->window_len; ++nf)
    if (!sf_layer_tok_ok(nf * m->num_slots)) return 0;
  return 1;
}

// 1 when sf_rollout_f32 would use seam launches for this model / batch with the calling thread's defaults (a caller that
// wants to verify sf_seam_timeouts() after the call only needs to when this is non-zero)
int sf_rollout_uses_seam_opts(const sf_rollouter* m, int B, const sf_rollout_opts* opts) {
  if (!m || B <= 0 || !m->layers) return 0;
  bool packed = m->in_proj_packed && m->out_proj_packed;
  for (int l = 0; l < m->num_layers; ++l)
    packed = packed && m->layers[l].lin1_packed && m->layers[l].lin2_packed && m->layers[l].attn_in_packed && m->layers[l].attn_out_packed;
  // the call's own options where given (a CU-masked caller: cus_available bounds the grid a seam launch may have), else the thread's defaults
  int seam_opt = sf_thread_opts().seam, cus = sf_thread_opts().cus, attn_heads = sf_thread_opts().attn_heads, attn_rows = sf_thread_opts().attn_rows;
  int tok = sf_thread_opts().layer_tok;
  if (opts) {
    if (opts->seam_fused >= 0) seam_opt = opts->seam_fused ? 1 : 0;
    if (opts->cus_available > 0) cus = opts->cus_available;
    if (opts->attn_heads_per_wg > 0) attn_heads = opts->attn_heads_per_wg;
    if (opts->attn_qkv_rows > 0) attn_rows = opts->attn_qkv_rows;
    if (opts->layer_tok != 0) tok = opts->layer_tok > 0 ? 1 : -1;
  }
  const int cap = cus > 0 && cus < 160 ? cus : 160;
  const bool tok_on = (tok > 0 || (tok == 0 && sf_get_layer_tok() != 0)) && sf_rollout_tok_ok(m);
  return packed && sf_get_precision() >= 1 && m->norm_first &&
         sf_layer_fused_ok(m->d_model, m->num_heads, m->ffn_dim, m->window_len * m->num_slots) &&
         sf_step_boundary_ok(m->d_model, m->slot_size) && (seam_opt >= 0 ? seam_opt != 0 : sf_get_seam_fused() != 0) &&
         sf_seam_blocks(B, m->num_slots) <= cap && attn_heads != 8 && attn_rows != 128 && !tok_on;
}
int sf_rollout_uses_seam(const sf_rollouter* m, int B) { return sf_rollout_uses_seam_opts(m, B, nullptr); }

int sf_rollout_bf16(const sf_rollouter* m, float* slots, int B, int T_total, int pred_len, void* ws, size_t ws_bytes,
                    void* stream) {
  sf_rollout_opts o = {2, -1, 0, 0};
  return sf_rollout_opts_f32(m, slots, B, T_total, pred_len, ws, ws_bytes, stream, &o);
}

// ---------------------------------------------------------------------------------------------
static int enc_chunk(int B) { return B < 32 ? B : 32; }

// the slot branch of a batched encode as one video-stationary launch (slot_chain.hip): OPT-IN (sf_set_slot_chain(1)); the default keeps
// the per-iteration launches (Slot-Attention iteration over the batch + slot update), which are faster for a batch alone on its CUs
static int g_slot_chain = 0;
int sf_get_slot_chain(void) { return g_slot_chain; }
// the Slot-Attention iterations of the per-step encode on feature rows kept as bf16 hi | lo (sa_attn_planes_kernel, slot_chain.hip: split-bf16 16x16x32
// MFMAs) instead of f32 rows and the exact-f32 tile kernel: process default below; sf_set_slot_attn_planes(0 / 1)
static int g_sa_planes = 1;
int sf_get_slot_attn_planes(void) { return g_sa_planes; }
int sf_set_slot_attn_planes(int on) {
  g_sa_planes = on ? 1 : 0;
  return 0;
}
int sf_set_slot_chain(int on) {
  g_slot_chain = on ? 1 : 0;
  return 0;
}

// the slot prologue of step t + 1 at the tail of step t's last slot update (slot_update_mfma.hip, NEXT form): on by default where it applies;
// sf_set_encode_fuse_next(0): the prologue as its own launch (sa_slot_prologue_kernel) on every step
static int g_enc_fuse_next = -1;
int sf_set_encode_fuse_next(int on) {
  g_enc_fuse_next = on ? 1 : 0;
  return 0;
}
int sf_get_encode_fuse_next(void) {
  if (g_enc_fuse_next < 0) {
    g_enc_fuse_next = 1;
  }
  return g_enc_fuse_next;
}

static size_t enc_ws_bytes(const sf_savi_encoder* m, int B, int kv_steps);
size_t sf_savi_encode_workspace_bytes(const sf_savi_encoder* m, int B) { return enc_ws_bytes(m, B, 2); }   // (two: the interleaved order below)
// the forked form keeps the Slot-Attention inputs of up to ENC_FORK_AHEAD time steps (the feature branch runs that far ahead of the slot branch)
static constexpr int ENC_FORK_AHEAD = 4;
static int enc_fork_steps(int T) { return T < 2 ? 2 : (T < ENC_FORK_AHEAD ? T : ENC_FORK_AHEAD); }
size_t sf_savi_encode_fork_workspace_bytes(const sf_savi_encoder* m, int B, int T) { return T >= 1 ? enc_ws_bytes(m, B, enc_fork_steps(T)) : 0; }
// One-stream encode with the convolutions of ALL T time steps as one launch per layer (the weights-stationary kernel of conv_ws.hip pays its 410 KB of
// weights per workgroup once per launch: 81 instead of 99 us per time step and layer on a 128-CU partition): the activations of B * T frames in three
// buffers on top of sf_savi_encode_workspace_bytes.  sf_savi_encode_fork_f32 (side_stream NULL) takes this form when it is handed that much workspace.
static size_t enc_batched_extra(const sf_savi_encoder* m, int B, int T) {
  int cmax = 0;
  for (int i = 1; i <= m->enc_layers && i < 9; ++i) cmax = m->enc_channels[i] > cmax ? m->enc_channels[i] : cmax;
  // + the Slot-Attention inputs of all B * T frames as bf16 hi | lo rows of 512 B (the video-stationary slot branch, slot_chain.hip)
  return 3 * pad256((size_t)B * T * 64 * 64 * cmax) + pad256((size_t)B * T * 64 * 64 * 128);
}
static bool enc_batched_ok(const sf_savi_encoder* m, int B, int T) {
  // (at most 384 frames per call: 1.2 GB of activations + 0.8 GB of feature rows; longer clips keep the per-step order)
  if (!m || T < 2 || B > enc_chunk(B) || (long long)B * T > 384 || m->enc_layers < 2 || sf_get_precision() != 1) return false;
  for (int i = 1; i < m->enc_layers; ++i)
    if (!m->conv_w_frag[i] || m->enc_channels[i] != 64 || m->enc_channels[i + 1] != 64) return false;
  return true;
}
size_t sf_savi_encode_batched_workspace_bytes(const sf_savi_encoder* m, int B, int T) {
  if (!m || B <= 0 || T <= 0) return 0;
  return enc_ws_bytes(m, B, 2) + (enc_batched_ok(m, B, T) ? enc_batched_extra(m, B, T) : 0);
}

static size_t enc_ws_bytes(const sf_savi_encoder* m, int B, int kv_steps) {
  if (!m || B <= 0) return 0;
  const size_t HW = 64 * 64;
  int cmax = 0;
  for (int i = 1; i <= m->enc_layers && i < 9; ++i) cmax = m->enc_channels[i] > cmax ? m->enc_channels[i] : cmax;
  const int Bc = enc_chunk(B), N = m->num_slots, D = m->slot_size, Ce = m->enc_out_channels;
  const int P = sf_sa_pick_partials((int)HW);
  const int R = B * N;
  const int hidp = m->pred_ffn_dim > 2 * D ? m->pred_ffn_dim : 2 * D;
  size_t t = 0;
  t += 2 * pad256((size_t)Bc * HW * cmax);
  t += 2 * pad256((size_t)Bc * HW * Ce);
  t += pad256((size_t)kv_steps * B * HW * 2 * D);
  t += 6 * pad256((size_t)R * D) + 2 * pad256((size_t)R * 2 * D);
  t += pad256((size_t)B * P * N * D) + pad256((size_t)B * P * N);
  t += tfm_ws_bytes(R, D, hidp) + pad256((size_t)R * 4 * (m->pred_hidden > 0 ? m->pred_hidden : 1));
  return t + 8192;
}

// CNN stack (savi.py:231-244: convs + soft position embedding) for `nb` frames: frame i at src + i*frame_stride;
// the last conv writes into `dst` (NHWC [nb,64,64,C_last]); featA/featB are ping-pong scratch.
// layers [i0, i1) of the stack
static int run_cnn_layers(const sf_savi_encoder* m, const float* src, long long frame_stride, int nb, float* dst, float* featA, float* featB,
                          int i0, int i1, hipStream_t st) {
  const int res = m->resolution;
  for (int i = i0; i < i1; ++i) {
    const bool lastc = (i == m->enc_layers - 1);
    const int cin = m->enc_channels[i], cout = m->enc_channels[i + 1];
    const float* add = lastc ? m->pos_table : nullptr;
    float* out = lastc ? dst : ((i & 1) ? featB : featA);
    const float* cur = i == 0 ? nullptr : (((i - 1) & 1) ? featB : featA);   // the previous layer's output
    if (i == 0) {
      SF_TRY(sf_conv2d_nchw_in_f32(src, frame_stride, m->conv_w[0], m->conv_b[0], add, out, nb, cin, res, res, cout,
                                   m->enc_ks, res == 128 ? 2 : 1, lastc ? 0 : 1, st));
    } else {
      // 64 -> 64 channels with a fragment-ordered weight copy: 4-row tiles, weights streamed as MFMA fragments (conv_rows4.hip)
      int rc = 1;
      // (weights stationary in registers where the launch gives every CU of the stream its rows: conv_ws.hip -- the same bits)
      if (rc == 1 && m->conv_w_frag[i])
        rc = sf_conv5x5_ws_ex(cur, m->conv_w_frag[i], m->conv_b[i], add, out, nb, 64, 64, cin, cout, m->enc_ks, lastc ? 0 : 1, 0, st);
      if (rc == 1 && m->conv_w_frag[i])
        rc = sf_conv5x5_rows4_ex(cur, m->conv_w_frag[i], m->conv_b[i], add, out, nb, 64, 64, cin, cout, m->enc_ks, lastc ? 0 : 1, st);
      if (rc < 0 || rc > 1) return rc;
      if (rc == 1)
        SF_TRY(sf_conv2d_nhwc_f32(cur, m->conv_w[i], m->conv_b[i], add, out, nb, 64, 64, cin, cout, m->enc_ks, lastc ? 0 : 1, st));
    }
  }
  return 0;
}
static int run_cnn(const sf_savi_encoder* m, const float* src, long long frame_stride, int nb, float* dst, float* featA,
                   float* featB, hipStream_t st) {
  return run_cnn_layers(m, src, frame_stride, nb, dst, featA, featB, 0, m->enc_layers, st);
}

size_t sf_savi_cnn_workspace_bytes(const sf_savi_encoder* m, int B) {
  if (!m || B <= 0) return 0;
  int cmax = 0;
  for (int i = 1; i <= m->enc_layers && i < 9; ++i) cmax = m->enc_channels[i] > cmax ? m->enc_channels[i] : cmax;
  return 2 * pad256((size_t)enc_chunk(B) * 64 * 64 * cmax) + 4096;
}

// CNN features of time steps [t0, t1) of every video: feat [t1-t0][B][64*64][C_last].  Independent of the slots, so a
// caller may compute them ahead of sf_savi_encode_pre_f32 on another stream (bench.py runs part of the NEXT batch's
// convolutions on the rollout stream's CUs while that stream would otherwise idle).
int sf_savi_cnn_f32(const sf_savi_encoder* m, const float* img, int B, int T, int t0, int t1, float* feat, void* ws,
                    size_t ws_bytes, void* stream) {
  SF_REQUIRE(m && img && feat && ws, "null pointer");
  SF_REQUIRE(B >= 1 && T >= 1 && t0 >= 0 && t1 >= t0 && t1 <= T, "bad batch / time range");
  SF_REQUIRE(m->resolution == 64 || m->resolution == 128, "resolution must be 64 or 128 (savi.py:226,236)");
  SF_REQUIRE(m->enc_layers >= 1 && m->enc_layers <= 8 && m->pos_table, "bad CNN config");
  SF_REQUIRE(ws_bytes >= sf_savi_cnn_workspace_bytes(m, B), "workspace too small");
  for (int i = 0; i < m->enc_layers; ++i) SF_REQUIRE(m->conv_w[i] != nullptr, "null conv weight");
  int cmax = 0;
  for (int i = 1; i <= m->enc_layers; ++i) cmax = m->enc_channels[i] > cmax ? m->enc_channels[i] : cmax;
  const int Bc = enc_chunk(B), HW = 64 * 64, Cl = m->enc_channels[m->enc_layers];
  Bump bp{(char*)ws, ws_bytes};
  float* featA = bp.take((size_t)Bc * HW * cmax);
  float* featB = bp.take((size_t)Bc * HW * cmax);
  if (!featA || !featB) return sf_set_err(-1, "workspace too small", __FILE__, __LINE__);
  const long long frame_elems = (long long)3 * m->resolution * m->resolution;
  // these launches usually run on another stream / CU partition than the encode: keep them out of the per-class
  // HIP-event timer so that bench.py's live conv figure stays "launches of the encode stream"
  sf_prof_suppress(1);
  int rc = 0;
  for (int t = t0; t < t1 && rc == 0; ++t)
    for (int b0 = 0; b0 < B && rc == 0; b0 += Bc) {
      const int nb = (B - b0 < Bc) ? (B - b0) : Bc;
      rc = run_cnn(m, img + ((long long)b0 * T + t) * frame_elems, (long long)T * frame_elems, nb,
                   feat + ((long long)(t - t0) * B + b0) * HW * Cl, featA, featB, (hipStream_t)stream);
    }
  sf_prof_suppress(0);
  return rc;
}

// ---- the encode in two halves (round 6): image features of a batch as bf16 hi | lo rows, and the slot branch of a GROUP of batches as one
//      video-stationary launch (slot_chain.hip).  The batch pipeline runs the first on its encode lane and the second in front of the group's rollout, on
//      the rollout stream: the slot branch is 32 workgroups of ~0.5 ms per batch that would leave the other 96 CUs of the lane idle. ----
// model-level conditions of the video-stationary slot branch (the CLEVRER shape of StoSAVi: savi.py:76-100, 393-402)
static bool enc_chain_model_ok(const sf_savi_encoder* m) {
  if (!m || sf_get_precision() != 1) return false;
  const int HW = 64 * 64, D = m->slot_size, Ce = m->enc_out_channels, Hm = m->slot_mlp_size, N = m->num_slots;
  const int Cl = m->enc_channels[m->enc_layers];
  const bool fold = m->sa_fold_q_w && m->sa_fold_q_w_t && m->sa_fold_gru_ih_t && Ce == D && Cl == 64 && sf_pixel_mlp_feat_ok(Cl, Ce);
  const bool su = m->sa_fold_gru_ih_p && m->sa_gru_hh_p && m->sa_mlp_w1_p && m->sa_mlp_w2_p && m->sa_fold_q_w_p && sf_slot_update_mfma_ok(D, Hm, sf_sa_pick_partials(HW));
  const bool prologue = m->pred_type == 0 && !m->pred_rnn && m->kd_mode == 1 && m->pm_w0_t && m->pm_w2_t && m->kd_w0_t && m->pm_w0_p && m->pm_w2_p && m->kd_w0_p &&
                        m->pm_ln_g && m->pm_ln_b && m->pm_b0 && m->pm_b2 && m->kd_b0 && m->init_latents;
  return fold && su && prologue && sf_get_encode_fuse_next() && sf_slot_chain_ok(D, Hm, HW, N) && m->enc_fc1_w && m->enc_fc2_w;
}
// prologue of step 0 + the chain, for NB batches of B videos; the four row buffers hold NB * B * N rows of D floats each
static int enc_slots_chain(const sf_savi_encoder* m, const void* planes, const float* noise, const float* prev, float* post, long long post_bs, float* kernel_dist,
                           float* attn, int NB, int B, int T, float* slotsA, float* slotsB, float* latents, float* q, hipStream_t st) {
  const int N = m->num_slots, D = m->slot_size, HW = 64 * 64;
  const float ln_eps = 1e-5f;
  const int pr = sf_slot_prologue_ex(prev, m->init_latents, m->pm_ln_g, m->pm_ln_b, m->pm_w0_t, m->pm_b0, m->pm_w2_t, m->pm_b2, m->pred_norm_first, m->kd_w0_t,
                                     m->kd_b0, noise, (long long)T * N * D, kernel_dist, (long long)T * N * 2 * D, m->sa_q_ln_g, m->sa_q_ln_b, m->sa_fold_q_w_t,
                                     slotsA, q, NB * B, N, D, ln_eps, st);
  if (pr != 0) return pr < 0 ? pr : sf_set_err(-1, "the one-launch slot prologue does not apply to this model", __FILE__, __LINE__);
  SfChainWeights cw;
  cw.gru_ih_p = m->sa_fold_gru_ih_p; cw.gru_hh_p = m->sa_gru_hh_p; cw.gru_b_ih = m->gru_b_ih; cw.gru_b_hh = m->gru_b_hh; cw.ln_g = m->mlp_ln_g; cw.ln_b = m->mlp_ln_b;
  cw.w1_p = m->sa_mlp_w1_p; cw.b1 = m->mlp_b1; cw.w2_p = m->sa_mlp_w2_p; cw.b2 = m->mlp_b2; cw.q_ln_g = m->sa_q_ln_g; cw.q_ln_b = m->sa_q_ln_b; cw.q_w_p = m->sa_fold_q_w_p;
  cw.pm_ln_g = m->pm_ln_g; cw.pm_ln_b = m->pm_ln_b; cw.pm_w0_p = m->pm_w0_p; cw.pm_b0 = m->pm_b0; cw.pm_w2_p = m->pm_w2_p; cw.pm_b2 = m->pm_b2;
  cw.pm_norm_first = m->pred_norm_first; cw.kd_w_p = m->kd_w0_p; cw.kd_b = m->kd_b0;
  return sf_slot_chain_ex(planes, NB, B, T, HW, N, m->num_iterations, 1.0f / sqrtf((float)D), m->sa_eps, ln_eps, slotsA, slotsB, latents, q, post, post_bs, attn,
                          noise, kernel_dist, &cw, st);
}

int sf_savi_chain_ok(const sf_savi_encoder* m, int B, int T) { return (enc_chain_model_ok(m) && enc_batched_ok(m, B, T)) ? 1 : 0; }
size_t sf_savi_planes_bytes(const sf_savi_encoder* m, int B, int T) { return (m && B > 0 && T > 0) ? (size_t)B * T * 64 * 64 * 512 : 0; }
size_t sf_savi_features_workspace_bytes(const sf_savi_encoder* m, int B, int T) {
  if (!m || B <= 0 || T <= 0) return 0;
  int cmax = 0;
  for (int i = 1; i <= m->enc_layers && i < 9; ++i) cmax = m->enc_channels[i] > cmax ? m->enc_channels[i] : cmax;
  return 3 * pad256((size_t)B * T * 64 * 64 * cmax) + 4096;
}
// Image features of B videos x T frames as the Slot-Attention inputs of the video-stationary slot branch: CNN (savi.py:231-244), encoder_out_layer
// (:245-250) and SlotAttention.norm_inputs (:66) -> planes [T][B][64 * 64] rows of 512 B (bf16 hi 128 | lo 128).  Independent of any slots.
int sf_savi_features_planes_f32(const sf_savi_encoder* m, const float* img, int B, int T, void* planes, void* ws, size_t ws_bytes, void* stream) {
  SF_REQUIRE(m && img && planes && ws, "sf_savi_features_planes_f32: null pointer");
  SF_REQUIRE(B >= 1 && T >= 1 && sf_savi_chain_ok(m, B, T), "sf_savi_features_planes_f32: the video-stationary slot branch does not apply (sf_savi_chain_ok)");
  SF_REQUIRE(ws_bytes >= sf_savi_features_workspace_bytes(m, B, T), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int HW = 64 * 64, res = m->resolution;
  int cmax = 0;
  for (int i = 1; i <= m->enc_layers; ++i) cmax = m->enc_channels[i] > cmax ? m->enc_channels[i] : cmax;
  Bump bp{(char*)ws, ws_bytes};
  float* big[3];
  for (int i = 0; i < 3; ++i) big[i] = bp.take((size_t)B * T * HW * cmax);
  if (!bp.ok) return sf_set_err(-1, "workspace too small", __FILE__, __LINE__);
  const long long frame_elems0 = (long long)3 * res * res;
  const int c1 = m->enc_channels[1];
  int rc0 = sf_conv_first_grouped_ex(img, (long long)T * frame_elems0, B, frame_elems0, m->conv_w[0], m->conv_b[0], nullptr, big[0], B * T, m->enc_channels[0], res, res, c1,
                                     m->enc_ks, res == 128 ? 2 : 1, 1, st);
  if (rc0 < 0 || rc0 > 1) return rc0;
  for (int t = 0; t < T && rc0 == 1; ++t)
    SF_TRY(sf_conv2d_nchw_in_f32(img + (long long)t * frame_elems0, (long long)T * frame_elems0, m->conv_w[0], m->conv_b[0], nullptr,
                                 big[0] + (long long)t * B * HW * c1, B, m->enc_channels[0], res, res, c1, m->enc_ks, res == 128 ? 2 : 1, 1, st));
  SF_TRY(run_cnn_layers(m, nullptr, 0, B * T, big[2], big[0], big[1], 1, m->enc_layers, st));
  return sf_pixel_mlp_feat_planes_ex(big[2], m->enc_ln_g, m->enc_ln_b, m->enc_fc1_w, m->enc_fc1_b, m->enc_fc2_w, m->enc_fc2_b, m->sa_norm_in_g, m->sa_norm_in_b, planes,
                                     B * T * HW, 1e-5f, st);
}
size_t sf_savi_slots_chain_workspace_bytes(const sf_savi_encoder* m, int videos) {
  return (m && videos > 0) ? 4 * pad256((size_t)videos * m->num_slots * m->slot_size) + 4096 : 0;
}
// The slot branch of NB batches of B videos over their T frames (StoSAVi.encode's per-step chain, savi.py:393-416, and the Slot-Attention iterations,
// :76-100) from sf_savi_features_planes_f32 rows: planes [NB][T][B][64 * 64][256 bf16]; noise NULL or [NB * B][T][N][D]; prev_slots NULL or [NB * B][N][D];
// video v's slots of step t -> post + v * post_bs + t * N * D; kernel_dist NULL or [NB * B][T][N][2 D]; attn NULL or [NB * B][T][N][64 * 64].
int sf_savi_slots_chain_f32(const sf_savi_encoder* m, const void* planes, const float* noise, const float* prev_slots, float* post, long long post_bs,
                            float* kernel_dist, float* attn, int NB, int B, int T, void* ws, size_t ws_bytes, void* stream) {
  SF_REQUIRE(m && planes && post && ws, "sf_savi_slots_chain_f32: null pointer");
  SF_REQUIRE(NB >= 1 && B >= 1 && T >= 1 && enc_chain_model_ok(m), "sf_savi_slots_chain_f32: the video-stationary slot branch does not apply (sf_savi_chain_ok)");
  SF_REQUIRE(noise == nullptr || m->kd_mode != 0, "noise given but the model has no kernel_dist layer");
  SF_REQUIRE(ws_bytes >= sf_savi_slots_chain_workspace_bytes(m, NB * B), "workspace too small");
  const size_t R = (size_t)NB * B * m->num_slots;
  Bump bp{(char*)ws, ws_bytes};
  float* sA = bp.take(R * m->slot_size);
  float* sB = bp.take(R * m->slot_size);
  float* lat = bp.take(R * m->slot_size);
  float* q = bp.take(R * m->slot_size);
  if (!bp.ok) return sf_set_err(-1, "workspace too small", __FILE__, __LINE__);
  return enc_slots_chain(m, planes, noise, prev_slots, post, post_bs, kernel_dist, attn, NB, B, T, sA, sB, lat, q, (hipStream_t)stream);
}

int sf_savi_encode_f32(const sf_savi_encoder* m, const float* img, const float* noise, const float* prev_slots,
                       float* lstm_h, float* lstm_c, int state_valid, float* post_slots, float* kernel_dist,
                       float* attn, int B, int T, void* ws, size_t ws_bytes, void* stream) {
  return sf_savi_encode_pre_f32(m, img, nullptr, 0, noise, prev_slots, lstm_h, lstm_c, state_valid, post_slots, kernel_dist,
                                attn, B, T, ws, ws_bytes, stream);
}

// As sf_savi_encode_f32, with the CNN features of the first n_pre time steps already computed by sf_savi_cnn_f32
// (feat_pre [n_pre][B][64*64][C_last]; NULL / 0: compute everything here).
int sf_savi_encode_pre_f32(const sf_savi_encoder* m, const float* img, const float* feat_pre, int n_pre, const float* noise,
                           const float* prev_slots, float* lstm_h, float* lstm_c, int state_valid, float* post_slots,
                           float* kernel_dist, float* attn, int B, int T, void* ws, size_t ws_bytes, void* stream) {
  return sf_savi_encode_fork_f32(m, img, feat_pre, n_pre, noise, prev_slots, lstm_h, lstm_c, state_valid, post_slots, kernel_dist, attn,
                                 B, T, ws, ws_bytes, stream, nullptr);
}

// fork / join events of the forked encode, per host thread (created on first use, kept for the life of the thread: the library's
// only host-side objects; no device memory)
static hipEvent_t enc_fork_event(int i) {
  // keyed by the CURRENT device: an event belongs to the device it was created on, and one host thread may drive several GPUs (engine.py keeps its
  // side streams per device); recording a device-0 event on a device-1 stream is hipErrorInvalidHandle
  thread_local std::map<int, std::vector<hipEvent_t>> per_dev;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::vector<hipEvent_t>& ev = per_dev[dev];
  while ((int)ev.size() <= i) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    ev.push_back(e);
  }
  return ev[i];
}

// The encode as TWO branches (side_stream != NULL).  The image features of a time step (CNN + per-pixel chain: ~420 us of dense
// launches per step at C2) do not depend on the slots; the slot branch of a step (prologue, Slot-Attention iterations, slot updates:
// ~140 us, half of it in seven-workgroup launches that leave the chip idle) needs only that step's features.  `stream` runs the
// features of all T steps back to back; `side_stream` follows one step behind with the slot branches, ordered by events (fork after
// the features of step t, join at the end: `stream` waits for the last slot update).  Captured into a hipGraph the two become parallel
// branches of the graph.  Same kernels, same arguments, same bits.  Workspace: sf_savi_encode_fork_workspace_bytes(m, B, T).
int sf_savi_encode_fork_f32(const sf_savi_encoder* m, const float* img, const float* feat_pre, int n_pre, const float* noise,
                            const float* prev_slots, float* lstm_h, float* lstm_c, int state_valid, float* post_slots,
                            float* kernel_dist, float* attn, int B, int T, void* ws, size_t ws_bytes, void* stream, void* side_stream) {
  SF_REQUIRE(m && img && post_slots && ws, "null pointer");
  const bool fork = side_stream != nullptr && side_stream != stream;
  SF_REQUIRE(n_pre >= 0 && n_pre <= T && (n_pre == 0 || feat_pre != nullptr), "bad precomputed-feature arguments");
  SF_REQUIRE(B >= 1 && T >= 1, "bad batch / clip length");
  SF_REQUIRE(m->resolution == 64 || m->resolution == 128, "resolution must be 64 or 128 (savi.py:226,236)");
  SF_REQUIRE(m->enc_layers >= 1 && m->enc_layers <= 8 && m->enc_channels[0] > 0 && (m->enc_ks & 1), "bad CNN config");
  SF_REQUIRE(m->num_slots >= 1 && m->num_slots <= 8 && m->num_iterations >= 1, "bad slot config");
  SF_REQUIRE(m->pos_table && m->enc_ln_g && m->enc_ln_b && m->enc_fc1_w && m->enc_fc1_b && m->enc_fc2_w &&
                 m->enc_fc2_b && m->sa_norm_in_g && m->sa_norm_in_b && m->sa_q_ln_g && m->sa_q_ln_b &&
                 m->sa_q_w && m->sa_kv_w && m->init_latents, "null encoder weight");
  SF_REQUIRE(m->kd_mode >= 0 && m->kd_mode <= 2, "bad kd_mode");
  SF_REQUIRE(m->kd_mode == 0 || (m->kd_w0 && m->kd_b0), "null kernel_dist weight");
  SF_REQUIRE(m->kd_mode != 2 || (m->kd_ln_g && m->kd_ln_b && m->kd_w3 && m->kd_b3), "null kernel_dist weight");
  SF_REQUIRE(noise == nullptr || m->kd_mode != 0, "noise given but the model has no kernel_dist layer");
  if (m->pred_type == 0)
    SF_REQUIRE(m->pm_ln_g && m->pm_ln_b && m->pm_w0 && m->pm_b0 && m->pm_w2 && m->pm_b2, "null predictor weight");
  else
    SF_REQUIRE(check_layers(m->pred_layers, m->pred_num_layers) == 0, "null predictor weight");
  if (m->pred_rnn)
    SF_REQUIRE(lstm_h && lstm_c && m->pred_hidden > 0 && m->lstm_w_ih && m->lstm_w_hh && m->lstm_b_ih &&
                   m->lstm_b_hh && m->proj_w && m->proj_b, "null LSTM state / weight");
  const int KV = fork ? enc_fork_steps(T) : 2;   // resident Slot-Attention inputs: a ring of KV time steps
  SF_REQUIRE(ws_bytes >= enc_ws_bytes(m, B, KV), "workspace too small");
  for (int i = 0; i < m->enc_layers; ++i) SF_REQUIRE(m->conv_w[i] != nullptr, "null conv weight");

  hipStream_t st_main = (hipStream_t)stream, st_side = fork ? (hipStream_t)side_stream : (hipStream_t)stream;
  hipStream_t st = st_main;
  const int HW = 64 * 64, res = m->resolution;
  const int N = m->num_slots, D = m->slot_size, Ce = m->enc_out_channels, Hm = m->slot_mlp_size;
  const int R = B * N, Bc = enc_chunk(B), P = sf_sa_pick_partials(HW);
  int cmax = 0;
  for (int i = 1; i <= m->enc_layers; ++i) cmax = m->enc_channels[i] > cmax ? m->enc_channels[i] : cmax;
  const int hidp = m->pred_ffn_dim > 2 * D ? m->pred_ffn_dim : 2 * D;

  Bump bp{(char*)ws, ws_bytes};
  float* featA = bp.take((size_t)Bc * HW * cmax);
  float* featB = bp.take((size_t)Bc * HW * cmax);
  float* h1 = bp.take((size_t)Bc * HW * Ce);
  float* h2 = bp.take((size_t)Bc * HW * Ce);
  float* kv_base = bp.take((size_t)KV * B * HW * 2 * D);
  const size_t kv_step = (size_t)B * HW * 2 * D;
  float* slotsA = bp.take((size_t)R * D);
  float* slotsB = bp.take((size_t)R * D);
  float* latents = bp.take((size_t)R * D);
  float* q = bp.take((size_t)R * D);
  float* lnbuf = bp.take((size_t)R * D);
  float* px = bp.take((size_t)R * D);
  float* kdist = bp.take((size_t)R * 2 * D);
  float* kdtmp = bp.take((size_t)R * 2 * D);
  float* pnum = bp.take((size_t)B * P * N * D);
  float* pden = bp.take((size_t)B * P * N);
  TfmWs tw;
  bool ok = tfm_ws_take(bp, tw, R, D, hidp);
  float* gates = bp.take((size_t)R * 4 * (m->pred_hidden > 0 ? m->pred_hidden : 1));
  if (!ok || !bp.ok) return sf_set_err(-1, "workspace too small", __FILE__, __LINE__);
  // all T steps' convolutions as one launch per layer (sf_savi_encode_batched_workspace_bytes): when the caller brought the room for it
  float* big[3] = {nullptr, nullptr, nullptr};
  const bool batched = !fork && n_pre == 0 && enc_batched_ok(m, B, T) && ws_bytes >= enc_ws_bytes(m, B, KV) + enc_batched_extra(m, B, T);
  float* planes = nullptr;
  if (batched) {
    for (int i = 0; i < 3; ++i) big[i] = bp.take((size_t)B * T * HW * cmax);
    planes = bp.take((size_t)B * T * HW * 128);
    if (!bp.ok) return sf_set_err(-1, "workspace too small", __FILE__, __LINE__);
  }

  const float* prev = prev_slots;
  if (m->pred_rnn && (prev == nullptr || !state_valid)) {
    // RNNPredictorWrapper.reset(): hidden_state = None -> zeros on first use (predictor.py:132-135)
    // (a KERNEL, not hipMemsetAsync: the encode may be captured into a hipGraph, and memset nodes were seen to stop clearing
    //  after an older graph exec had been destroyed -- see zero_words_kernel above; pipeline encode graphs hit it at once)
    const long long nz = (long long)R * m->pred_hidden;
    hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)(((nz + 3) / 4 + 255) / 256)), dim3(256), 0, st, lstm_h, lstm_c, nz);
    SF_CHECK_LAUNCH();
  }
  const long long frame_elems = (long long)3 * res * res;
  const float ln_eps = 1e-5f;
  // Slot Attention on the normalised pixel features with the key / value projections folded into project_q and the GRU input
  // matrix (include/slotformer_hip.h, sa_fold_*; widths 128 and 192); without the folded copies, or at other widths: k|v as the reference computes them
  const bool feat192 = m->enc_channels[m->enc_layers] == 64 && Ce == 192 && m->enc_fc1_p && m->enc_fc2_p;
  const bool fold = sf_get_precision() >= 1 && m->sa_fold_q_w && m->sa_fold_q_w_t && m->sa_fold_gru_ih_t && Ce == D &&
                    (sf_pixel_mlp_feat_ok(m->enc_channels[m->enc_layers], Ce) || feat192);
  const bool sa_planes = fold && !feat192 && sf_get_slot_attn_planes() && sf_slot_attn_planes_ok(HW, D, N) && m->enc_channels[m->enc_layers] == 64 &&
                         P == HW / 256;
  const float* q_w = fold ? m->sa_fold_q_w : m->sa_q_w;
  const float* q_w_t = fold ? m->sa_fold_q_w_t : m->sa_q_w_t;
  const float* gru_ih_t = fold ? m->sa_fold_gru_ih_t : m->gru_w_ih;
  const void* q_w_p = fold ? m->sa_fold_q_w_p : m->sa_q_w_p;
  const void* gru_ih_p = fold ? m->sa_fold_gru_ih_p : m->sa_gru_ih_p;

  // slot update on the matrix cores (slot_update_mfma.hip) when the packed copies are there (slot size 128); otherwise the VALU kernel
  const bool su_packed = sf_get_precision() >= 1 && gru_ih_p && m->sa_gru_hh_p && m->sa_mlp_w1_p && m->sa_mlp_w2_p && q_w_p;
  const bool su_mfma = su_packed && sf_slot_update_mfma_ok(D, Hm, P);
  const bool su_wide = su_packed && sf_slot_update_wide_ok(D, Hm, P);   // slot size 192 (slot_update_wide.hip)
  // time-step order; forked: the features of step t are enqueued on `stream`, its slot branch on `side_stream` behind them (event),
  // and the features of step t + KV wait for the slot branch of step t to release its ring slot
  // the one-launch slot prologue applies (CLEVRER configuration); with packed copies of its three matrices and the matrix-core slot update, the
  // prologue of step t + 1 runs at the tail of step t's last update (slot_update_mfma.hip, NEXT form)
  const bool can_prologue = m->pred_type == 0 && !m->pred_rnn && m->kd_mode == 1 && m->pm_w0_t && m->pm_w2_t && m->kd_w0_t && q_w_t;
  const bool can_fuse_next = sf_get_encode_fuse_next() && can_prologue && su_mfma && m->pm_w0_p && m->pm_w2_p && m->kd_w0_p && m->pm_ln_g && m->pm_ln_b && m->pm_b0 &&
                             m->pm_b2 && m->kd_b0;
  bool next_done = false;
  const bool planes_all = false;
  if (batched) {
    // layer 0 per time step (the frames of one step lie T frames apart), every later layer over the B * T frames in [t][b] order; the last one
    // leaves the features of step t at big[2] + t * B * HW * Cl
    const long long frame_elems0 = (long long)3 * res * res;
    const int c1 = m->enc_channels[1];
    int rc0 = 1;
    if (sf_get_precision() >= 1)   // one launch for the B * T frames where the first-layer kernel applies (3 -> 64 channels at 128 x 128 / 64 x 64)
      rc0 = sf_conv_first_grouped_ex(img, (long long)T * frame_elems0, B, frame_elems0, m->conv_w[0], m->conv_b[0], nullptr, big[0], B * T, m->enc_channels[0], res,
                                     res, c1, m->enc_ks, res == 128 ? 2 : 1, 1, st_main);
    if (rc0 < 0 || rc0 > 1) return rc0;
    for (int t = 0; t < T && rc0 == 1; ++t)
      SF_TRY(sf_conv2d_nchw_in_f32(img + (long long)t * frame_elems0, (long long)T * frame_elems0, m->conv_w[0], m->conv_b[0], nullptr,
                                   big[0] + (long long)t * B * HW * c1, B, m->enc_channels[0], res, res, c1, m->enc_ks, res == 128 ? 2 : 1, 1, st_main));
    SF_TRY(run_cnn_layers(m, nullptr, 0, B * T, big[2], big[0], big[1], 1, m->enc_layers, st_main));
    // (the per-pixel chain stays per time step: as ONE launch for the B * T frames it takes 333 instead of 6 x 65 us on the lane, but the Slot-Attention
    //  iterations then read their rows from HBM instead of the cache the launch in front of them left warm: 18.9 -> 20.8 us each -- no gain, probes r06)
    // ---- the slot branch of all T steps as ONE video-stationary launch (slot_chain.hip): the per-pixel chain of the B * T frames in one launch
    //      (feature rows as bf16 hi | lo), the prologue of step 0, then one workgroup per video ----
    if (sf_get_slot_chain() && enc_chain_model_ok(m)) {
      SF_TRY(sf_pixel_mlp_feat_planes_ex(big[2], m->enc_ln_g, m->enc_ln_b, m->enc_fc1_w, m->enc_fc1_b, m->enc_fc2_w, m->enc_fc2_b, m->sa_norm_in_g,
                                         m->sa_norm_in_b, planes, B * T * HW, ln_eps, st_main));
      // (the chain's row buffers: the slot buffers of this function's workspace)
      return enc_slots_chain(m, planes, noise, prev, post_slots, (long long)T * N * D, kernel_dist, attn, 1, B, T, slotsA, slotsB, latents, q, st_main);
    }
  }
  for (int t = 0; t < T; ++t) {
    float* kv = planes_all ? (float*)((char*)planes + (size_t)t * B * HW * 512) : kv_base + (size_t)(t % KV) * kv_step;
    // ---- CNN encoder + per-pixel MLP + K/V for the B frames of step t ---------------------
    st = st_main;
    if (fork && t >= KV) {   // the ring slot is free once the slot branch of step t - KV has read it
      if (hipStreamWaitEvent(st_main, enc_fork_event(T + 1 + (t - KV)), 0) != hipSuccess) return sf_set_err((int)hipGetLastError(), "hipStreamWaitEvent", __FILE__, __LINE__);
    }
    for (int b0 = 0; b0 < B && !planes_all; b0 += Bc) {
      const int nb = (B - b0 < Bc) ? (B - b0) : Bc;
      const int Cl0 = m->enc_channels[m->enc_layers];
      const float* cur;
      if (t < n_pre) {
        cur = feat_pre + ((long long)t * B + b0) * HW * Cl0;   // computed ahead of time by sf_savi_cnn_f32
      } else if (batched) {
        cur = big[2] + ((long long)t * B + b0) * HW * Cl0;
      } else {
        float* dstf = (m->enc_layers & 1) ? featA : featB;   // the buffer the last conv does not read
        SF_TRY(run_cnn(m, img + ((long long)b0 * T + t) * frame_elems, (long long)T * frame_elems, nb, dstf, featA, featB, st));
        cur = dstf;
      }
      const int Cl = m->enc_channels[m->enc_layers];
      const int Mp = nb * HW;
      // encoder_out_layer (LN -> Linear -> ReLU -> Linear, savi.py:245-250) and k|v = [Wk;Wv] LN(inputs)
      // (savi.py:66-70): one fused kernel per 128-pixel tile in split-bf16 mode (pixel_mlp.hip), else three GEMMs
      if (fold && feat192) {
        SF_TRY(sf_pixel_mlp_feat192_ex(cur, m->enc_ln_g, m->enc_ln_b, m->enc_fc1_p, m->enc_fc1_b, m->enc_fc2_p, m->enc_fc2_b,
                                       m->sa_norm_in_g, m->sa_norm_in_b, kv + (long long)b0 * HW * Ce, Mp, ln_eps, st));
        continue;
      }
      if (fold && sa_planes) {   // rows of 512 B (bf16 hi | lo): what sa_attn_planes_kernel streams
        SF_TRY(sf_pixel_mlp_feat_planes_ex(cur, m->enc_ln_g, m->enc_ln_b, m->enc_fc1_w, m->enc_fc1_b, m->enc_fc2_w, m->enc_fc2_b, m->sa_norm_in_g, m->sa_norm_in_b,
                                           (char*)kv + (size_t)b0 * HW * 512, Mp, ln_eps, st));
        continue;
      }
      if (fold) {
        SF_TRY(sf_pixel_mlp_feat_ex(cur, m->enc_ln_g, m->enc_ln_b, m->enc_fc1_w, m->enc_fc1_b, m->enc_fc2_w, m->enc_fc2_b,
                                    m->sa_norm_in_g, m->sa_norm_in_b, kv + (long long)b0 * HW * Ce, Mp, ln_eps, st));
        continue;
      }
      float* kv_dst = kv + (long long)b0 * HW * 2 * D;
      int fused = 1;
      if (sf_get_precision() >= 1)
        fused = sf_pixel_mlp_kv_ex(cur, m->enc_ln_g, m->enc_ln_b, m->enc_fc1_w, m->enc_fc1_b, m->enc_fc2_w,
                                   m->enc_fc2_b, m->sa_norm_in_g, m->sa_norm_in_b, m->sa_kv_w, kv_dst, Mp, Cl, Ce,
                                   2 * D, ln_eps, st);
      if (fused < 0 || fused > 1) return fused;
      if (fused == 1) {
        SF_TRY(sf_linear_ex(cur, sf_rows(Cl), m->enc_fc1_w, m->enc_fc1_b, m->enc_ln_g, m->enc_ln_b, ln_eps, nullptr,
                            sf_rows(Ce), 0, h1, sf_rows(Ce), Mp, Ce, Cl, 1, st));
        SF_TRY(sf_linear_ex(h1, sf_rows(Ce), m->enc_fc2_w, m->enc_fc2_b, nullptr, nullptr, ln_eps, nullptr,
                            sf_rows(Ce), 0, h2, sf_rows(Ce), Mp, Ce, Ce, 0, st));
        SF_TRY(sf_linear_ex(h2, sf_rows(Ce), m->sa_kv_w, nullptr, m->sa_norm_in_g, m->sa_norm_in_b, ln_eps, nullptr,
                            sf_rows(2 * D), 0, kv_dst, sf_rows(2 * D), Mp, 2 * D, Ce, 0, st));
      }
    }
    if (fork) {
      hipEvent_t e = enc_fork_event(t);
      SF_REQUIRE(e != nullptr, "hipEventCreate failed");
      if (hipEventRecord(e, st_main) != hipSuccess || hipStreamWaitEvent(st_side, e, 0) != hipSuccess)
        return sf_set_err((int)hipGetLastError(), "fork event", __FILE__, __LINE__);
      st = st_side;
    }
    // ---- one-launch slot prologue (CLEVRER configuration: residual-MLP predictor without LSTM, single-Linear kernel
    //      distribution): init / predictor -> kernel_dist -> sampling -> q of the first iteration (slot_attn.hip) ----
    float* s_in = slotsA;
    float* s_out = slotsB;
    int prologue = 1;
    if (next_done) {   // computed at the tail of the previous step's last slot update (NEXT form below)
      prologue = 0;
      next_done = false;
    } else if (can_prologue)
      prologue = sf_slot_prologue_ex(prev, m->init_latents, m->pm_ln_g, m->pm_ln_b, m->pm_w0_t, m->pm_b0, m->pm_w2_t, m->pm_b2,
                                     m->pred_norm_first, m->kd_w0_t, m->kd_b0, noise ? noise + (long long)t * N * D : nullptr,
                                     (long long)T * N * D, kernel_dist ? kernel_dist + (long long)t * N * 2 * D : nullptr,
                                     (long long)T * N * 2 * D, m->sa_q_ln_g, m->sa_q_ln_b, q_w_t, s_in, q, B, N, D, ln_eps,
                                     st);
    if (prologue < 0 || prologue > 1) return prologue;
    if (prologue == 1) {
      // ---- slot initialisation: init_latents or predictor(prev_slots)  (savi.py:393-398) ------
      const float* lat;
      if (prev == nullptr) {
        SF_TRY(sf_copy_rows_ex(m->init_latents, sf_rows_batched(D, N, 0, 0), latents, sf_rows(D), R, D, st));
        lat = latents;
      } else {
        // Transformer predictor (+ LSTM wrapper) in one launch (pred_step.hip) when its packed weights are there
        int pstep = 1;
        if (m->pred_type == 1 && m->pred_packed && sf_get_precision() >= 1 && !t_plain_gemms)
          pstep = sf_pred_step_ex(prev, m->pred_layers, m->pred_num_layers, m->pred_num_heads, m->pred_ffn_dim, m->pred_norm_first,
                                  m->pred_packed, m->lstm_b_ih, m->lstm_b_hh, m->proj_b, m->pred_hidden, m->pred_rnn ? lstm_h : nullptr,
                                  m->pred_rnn ? lstm_c : nullptr, latents, B, N, D, 1e-5f, st);
        if (pstep < 0 || pstep > 1) return pstep;
        if (pstep == 0) {
          lat = latents;
        } else {
          const float* pout;
          if (m->pred_type == 0) {
            // ResidualMLPPredictor (predictor.py:65-73)
            SF_TRY(sf_layernorm_ex(prev, sf_rows(D), m->pm_ln_g, m->pm_ln_b, lnbuf, sf_rows(D), R, D, ln_eps, st));
            SF_TRY(sf_linear_ex(lnbuf, sf_rows(D), m->pm_w0, m->pm_b0, nullptr, nullptr, ln_eps, nullptr, sf_rows(D), 0,
                                tw.hid, sf_rows(2 * D), R, 2 * D, D, 1, st));
            SF_TRY(sf_linear_ex(tw.hid, sf_rows(2 * D), m->pm_w2, m->pm_b2, nullptr, nullptr, ln_eps,
                                m->pred_norm_first ? lnbuf : prev, sf_rows(D), 0, px, sf_rows(D), R, D, 2 * D, 0, st));
            pout = px;
          } else {
            // TransformerPredictor over the N slots (predictor.py:20-44)
            SF_TRY(sf_copy_rows_ex(prev, sf_rows(D), px, sf_rows(D), R, D, st));
            float* cur = px;
            for (int l = 0; l < m->pred_num_layers; ++l) {
              float* outp = nullptr;
              SF_TRY(tfm_layer(m->pred_layers[l], cur, tw, B, N, N, D, m->pred_num_heads, m->pred_ffn_dim,
                               m->pred_norm_first, st, &outp));
              cur = outp;
            }
            pout = cur;
          }
          if (m->pred_rnn) {
            // nn.LSTM, seq len 1, batch B*N (predictor.py:113-120)
            const int Hh = m->pred_hidden;
            SF_TRY(sf_linear_ex(pout, sf_rows(D), m->lstm_w_ih, m->lstm_b_ih, nullptr, nullptr, ln_eps, nullptr,
                                sf_rows(4 * Hh), 0, gates, sf_rows(4 * Hh), R, 4 * Hh, D, 0, st));
            SF_TRY(sf_linear_ex(lstm_h, sf_rows(Hh), m->lstm_w_hh, m->lstm_b_hh, nullptr, nullptr, ln_eps, gates,
                                sf_rows(4 * Hh), 0, gates, sf_rows(4 * Hh), R, 4 * Hh, Hh, 0, st));
            SF_TRY(sf_lstm_pointwise_ex(gates, lstm_c, lstm_h, lstm_c, R, Hh, st));
            SF_TRY(sf_linear_ex(lstm_h, sf_rows(Hh), m->proj_w, m->proj_b, nullptr, nullptr, ln_eps, nullptr,
                                sf_rows(D), 0, latents, sf_rows(D), R, D, Hh, 0, st));
            lat = latents;
          } else {
            lat = pout;
          }
        }
      }
      // ---- kernel distribution + sampling (savi.py:401-402) --------------------------------------
      if (m->kd_mode == 0) {
        SF_TRY(sf_copy_rows_ex(lat, sf_rows(D), s_in, sf_rows(D), R, D, st));
      } else {
        if (m->kd_mode == 1) {
          SF_TRY(sf_linear_ex(lat, sf_rows(D), m->kd_w0, m->kd_b0, nullptr, nullptr, ln_eps, nullptr, sf_rows(2 * D), 0,
                              kdist, sf_rows(2 * D), R, 2 * D, D, 0, st));
        } else {
          SF_TRY(sf_linear_ex(lat, sf_rows(D), m->kd_w0, m->kd_b0, nullptr, nullptr, ln_eps, nullptr, sf_rows(2 * D), 0,
                              kdtmp, sf_rows(2 * D), R, 2 * D, D, 0, st));
          SF_TRY(sf_linear_ex(kdtmp, sf_rows(2 * D), m->kd_w3, m->kd_b3, m->kd_ln_g, m->kd_ln_b, ln_eps, nullptr,
                              sf_rows(2 * D), 0, kdist, sf_rows(2 * D), R, 2 * D, 2 * D, 0, st, /*ln_relu=*/1));
        }
        const SfRowMap nmap = sf_rows_batched(D, N, (long long)T * N * D, (long long)t * N * D);
        SF_TRY(sf_sample_dist_ex(kdist, noise, nmap, s_in, R, D, st));
        if (kernel_dist)
          SF_TRY(sf_copy_rows_ex(kdist, sf_rows(2 * D), kernel_dist,
                                 sf_rows_batched(2 * D, N, (long long)T * N * 2 * D, (long long)t * N * 2 * D), R,
                                 2 * D, st));
      }
      // q of the first iteration: LN-fused GEMM; every later q comes out of the slot-update kernel, which also writes the
      // last iteration's result straight into post_slots[:, t]
      SF_TRY(sf_linear_ex(s_in, sf_rows(D), q_w, nullptr, m->sa_q_ln_g, m->sa_q_ln_b, ln_eps, nullptr,
                          sf_rows(D), 0, q, sf_rows(D), R, D, D, 0, st));
    }
    // ---- Slot Attention iterations (savi.py:76-100) -------------------------------------------
    const float scale = 1.0f / sqrtf((float)D);
    for (int it = 0; it < m->num_iterations; ++it) {
      const bool last_it = (it == m->num_iterations - 1);
      float* aout = (attn && last_it) ? attn + (long long)t * N * HW : nullptr;
      if (fold && sa_planes)   // the same iteration on the bf16 hi | lo rows (split-bf16 MFMAs; the same records)
        SF_TRY(sf_slot_attn_planes_ex(kv, HW, q, pnum, pden, aout, (long long)T * N * HW, B, HW, N, scale, m->sa_eps, st));
      else if (fold)   // keys = values = the normalised features (q is Wk^T q here, the GRU input matrix is W_ih Wv)
        SF_TRY(sf_slot_attn_iter_ex(kv, kv, Ce, (long long)HW * Ce, q, pnum, pden, aout, (long long)T * N * HW, B, HW, N, D, scale,
                                    m->sa_eps, st));
      else
        SF_TRY(sf_slot_attn_iter_ex(kv, kv + D, 2 * D, (long long)HW * 2 * D, q, pnum, pden, aout,
                                    (long long)T * N * HW, B, HW, N, D, scale, m->sa_eps, st));
      if (su_mfma) {
        bool rode = false;
        // (the one-pass Slot-Attention kernel leaves every second partial record zero: the update reads the others -- eight, one round of requests)
        const int p_step = (fold && P == HW / 256 && (sa_planes || sf_slot_attn_sparse_records(kv, kv, HW, D))) ? 2 : 1;
        const bool fuse_next = can_fuse_next && last_it && t + 1 < T && prologue == 0;
        if (fuse_next) {
          SfNextStep nx;
          nx.pm_ln_g = m->pm_ln_g; nx.pm_ln_b = m->pm_ln_b; nx.pm_w0_p = m->pm_w0_p; nx.pm_b0 = m->pm_b0; nx.pm_w2_p = m->pm_w2_p; nx.pm_b2 = m->pm_b2;
          nx.norm_first = m->pred_norm_first; nx.kd_w_p = m->kd_w0_p; nx.kd_b = m->kd_b0;
          nx.noise = noise ? noise + (long long)(t + 1) * N * D : nullptr; nx.noise_bs = (long long)T * N * D;
          nx.kdist_out = kernel_dist ? kernel_dist + (long long)(t + 1) * N * 2 * D : nullptr; nx.kdist_bs = (long long)T * N * 2 * D;
          nx.slots = slotsA;   // where the next step's iterations start
          // (the finished rows of step t go to post_slots[:, t]; their ping-pong copy is not read again, and must not alias the sampled slots)
          SF_TRY(sf_slot_update_mfma_ex(pnum, pden, P, s_in, gru_ih_p, m->sa_gru_hh_p, m->gru_b_ih, m->gru_b_hh, m->mlp_ln_g, m->mlp_ln_b,
                                        m->sa_mlp_w1_p, m->mlp_b1, m->sa_mlp_w2_p, m->mlp_b2, s_out == slotsA ? latents : s_out,
                                        post_slots + (long long)t * N * D, (long long)T * N * D, m->sa_q_ln_g, m->sa_q_ln_b, q_w_p, q, B, N, ln_eps, st,
                                        &nx, p_step));
          next_done = true;
          rode = true;
        }
        if (!rode)
          SF_TRY(sf_slot_update_mfma_ex(pnum, pden, P, s_in, gru_ih_p, m->sa_gru_hh_p, m->gru_b_ih, m->gru_b_hh, m->mlp_ln_g,
                                        m->mlp_ln_b, m->sa_mlp_w1_p, m->mlp_b1, m->sa_mlp_w2_p, m->mlp_b2, s_out,
                                        last_it ? post_slots + (long long)t * N * D : nullptr, (long long)T * N * D, m->sa_q_ln_g,
                                        m->sa_q_ln_b, q_w_p, last_it ? nullptr : q, B, N, ln_eps, st, nullptr, p_step));
        float* tmp = s_in;
        s_in = s_out;
        s_out = tmp;
        continue;
      }
      if (su_wide) {
        SF_TRY(sf_slot_update_wide_ex(pnum, pden, P, s_in, gru_ih_p, m->sa_gru_hh_p, m->gru_b_ih, m->gru_b_hh, m->mlp_ln_g, m->mlp_ln_b,
                                      m->sa_mlp_w1_p, m->mlp_b1, m->sa_mlp_w2_p, m->mlp_b2, s_out,
                                      last_it ? post_slots + (long long)t * N * D : nullptr, (long long)T * N * D, m->sa_q_ln_g, m->sa_q_ln_b,
                                      q_w_p, last_it ? nullptr : q, B, N, ln_eps, st));
        float* tmp = s_in;
        s_in = s_out;
        s_out = tmp;
        continue;
      }
      SF_TRY(sf_slot_update_ex(pnum, pden, P, s_in, gru_ih_t, m->gru_w_hh, m->gru_b_ih, m->gru_b_hh, m->mlp_ln_g,
                               m->mlp_ln_b, m->mlp_w1, m->mlp_b1, m->mlp_w2, m->mlp_b2, s_out,
                               last_it ? post_slots + (long long)t * N * D : nullptr, (long long)T * N * D,
                               m->sa_q_ln_g, m->sa_q_ln_b, q_w_t, (last_it || !q_w_t) ? nullptr : q, B, N, D, Hm, ln_eps,
                               st));
      if (!last_it && !q_w_t)   // no transposed copy of project_q given: the LN-fused GEMM produces q
        SF_TRY(sf_linear_ex(s_out, sf_rows(D), q_w, nullptr, m->sa_q_ln_g, m->sa_q_ln_b, ln_eps, nullptr, sf_rows(D), 0, q,
                            sf_rows(D), R, D, D, 0, st));
      float* tmp = s_in;
      s_in = s_out;
      s_out = tmp;
    }
    // s_in now holds post_slots of step t (also written to post_slots[:, t] by the last slot update)
    // prev_slots for the next step must not alias the ping-pong buffers that step overwrites:
    // keep it in `lnbuf`-independent storage (q is rewritten first, so use latents' twin `px`?)
    // -> simplest: the next step reads `prev` only before it writes slotsA/slotsB.
    prev = s_in;
    if (fork && t + KV < T) {   // the features of step t + KV may overwrite this step's ring slot now
      hipEvent_t e = enc_fork_event(T + 1 + t);
      SF_REQUIRE(e != nullptr, "hipEventCreate failed");
      if (hipEventRecord(e, st_side) != hipSuccess) return sf_set_err((int)hipGetLastError(), "hipEventRecord", __FILE__, __LINE__);
    }
  }
  if (fork) {   // join: the calling stream continues behind the last slot update
    hipEvent_t e = enc_fork_event(T);
    SF_REQUIRE(e != nullptr, "hipEventCreate failed");
    if (hipEventRecord(e, st_side) != hipSuccess || hipStreamWaitEvent(st_main, e, 0) != hipSuccess)
      return sf_set_err((int)hipGetLastError(), "fork join", __FILE__, __LINE__);
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// StoSAVi.decode (savi.py:504-525)
// floats of the largest activation map of ONE slot image: max over the layers of size_i^2 x channels_i (the broadcast input, every layer output)
static size_t dec_maxact(const sf_savi_decoder* m) {
  size_t best = (size_t)m->dec_res * m->dec_res * m->dec_channels[0];
  int size = m->dec_res;
  for (int i = 0; i < m->dec_layers && i < 8; ++i) {
    size *= m->dec_strides[i] > 0 ? m->dec_strides[i] : 1;
    const size_t a = (size_t)size * size * m->dec_channels[i + 1];
    best = a > best ? a : best;
  }
  const size_t tab = (size_t)25 * m->dec_channels[1];   // the first layer's class table
  return best > tab ? best : tab;
}
// frames per chunk: one activation buffer <= 1 GiB (7 slots at 128 x 128 x 64: 36 frames = 252 slot images per launch -- one round of
// 16-pixel-wide tiles, four of 32-wide, sixteen of 64-wide; chunks of 4 frames, the round-1 size, left most of the chip idle in the small layers)
static int dec_chunk(const sf_savi_decoder* m, int F) {
  const double per_frame = (double)m->num_slots * (double)dec_maxact(m);
  int fc = (int)(256.0 * 1024 * 1024 / per_frame);
  if (fc < 1) fc = 1;
  return fc < F ? fc : F;
}

size_t sf_savi_decode_workspace_bytes(const sf_savi_decoder* m, int F) {
  if (!m || F <= 0) return 0;
  const size_t R = (size_t)dec_chunk(m, F) * m->num_slots, HW = (size_t)m->resolution * m->resolution;
  return 2 * pad256(R * dec_maxact(m)) + pad256(R * HW * 4) + pad256(R) + 4096;
}

int sf_savi_decode_f32(const sf_savi_decoder* m, const float* slots, float* recon_combined, float* recons,
                       float* masks, int F, void* ws, size_t ws_bytes, void* stream) {
  return sf_savi_decode_seg_f32(m, slots, recon_combined, recons, masks, nullptr, nullptr, 0.5f, F, ws, ws_bytes, stream);
}

int sf_savi_decode_seg_f32(const sf_savi_decoder* m, const float* slots, float* recon_combined, float* recons, float* masks,
                           long long* seg_i64, unsigned char* seg_u8, float fg_thre, int F, void* ws, size_t ws_bytes, void* stream) {
  SF_REQUIRE(m && slots && recon_combined && ws, "null pointer");
  SF_REQUIRE(F >= 1 && m->dec_layers >= 1 && m->dec_layers <= 8 && m->num_slots >= 1 && m->dec_res >= 1 && (m->dec_ks & 1),
             "bad decoder config");
  SF_REQUIRE(m->dec_channels[0] == m->slot_size && (m->slot_size % 4) == 0, "dec_channels[0] must equal slot_size");
  SF_REQUIRE(m->pos_table && m->out_w && m->out_b, "null decoder weight");
  int size = m->dec_res;
  for (int i = 0; i < m->dec_layers; ++i) {
    SF_REQUIRE(m->deconv_w[i] != nullptr && m->dec_strides[i] >= 1 && (m->dec_channels[i + 1] % 4) == 0, "bad deconv layer");
    size *= m->dec_strides[i];
  }
  SF_REQUIRE(size == m->resolution, "decoder output size does not match the resolution (savi.py:279-284)");
  SF_REQUIRE(ws_bytes >= sf_savi_decode_workspace_bytes(m, F), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int N = m->num_slots, D = m->slot_size, res = m->resolution, HW = res * res;
  const int Fc = dec_chunk(m, F);
  const size_t maxact = dec_maxact(m);
  Bump bp{(char*)ws, ws_bytes};
  float* bufA = bp.take((size_t)Fc * N * maxact);
  float* bufB = bp.take((size_t)Fc * N * maxact);
  float* dec = bp.take((size_t)Fc * N * HW * 4);
  unsigned* slot_max = (unsigned*)bp.take((size_t)Fc * N);
  if (!bp.ok) return sf_set_err(-1, "workspace too small", __FILE__, __LINE__);
  const bool bf3 = sf_get_precision() == 1;
  const int nl = m->dec_layers, Cl = m->dec_channels[nl];
  for (int f0 = 0; f0 < F; f0 += Fc) {
    const int nf = (F - f0 < Fc) ? (F - f0) : Fc, R = nf * N;
    float* cur = bufA;
    float* nxt = bufB;
    int hin = m->dec_res;
    bool head_done = false;
    for (int i = 0; i < nl; ++i) {
      const int Ci = m->dec_channels[i], Co = m->dec_channels[i + 1], sd = m->dec_strides[i];
      const bool last = i == nl - 1;
      int rc = 1;
      if (i == 0) {
        if (m->l0_weff && m->l0_posterm && sd == 2 && m->dec_ks == 5 && hin >= 2) {
          // the first layer on its broadcast input: table [R][25 Co] = slots . l0_weff^T, then the expansion (include/slotformer_hip.h)
          SF_TRY(sf_linear_ex(slots + (long long)f0 * N * D, sf_rows(D), m->l0_weff, nullptr, nullptr, nullptr, 0.f, nullptr, sf_rows(25 * Co), 0,
                              nxt, sf_rows(25 * Co), R, 25 * Co, D, 0, st));
          SF_TRY(sf_decode_l0_expand_f32(nxt, m->l0_posterm, cur, R, hin, Co, st));
          hin *= sd;
          continue;
        }
        SF_TRY(sf_slot_broadcast_f32(slots + (long long)f0 * N * D, m->pos_table, cur, R, m->dec_res * m->dec_res, D, st));
      }
      if (bf3 && sd == 2 && m->deconv_w_frag[i]) {
        // parity-class kernel with streamed weight fragments; on the last layer with the 1x1 head in its epilogue (deconv_s2.hip)
        const bool head = last && hin == 64 && Cl == 64;
        rc = sf_deconv5x5s2_ex(cur, m->deconv_w_frag[i], m->deconv_b[i], head ? m->out_w : nullptr, head ? m->out_b : nullptr,
                               head ? dec : nxt, R, hin, hin, Ci, Co, m->dec_ks, sd, 1, st);
        if (rc < 0 || rc > 1) return rc;
        if (rc == 0 && head) head_done = true;
      }
      if (rc == 1 && bf3 && sd == 1 && last && m->deconv_w_frag[i] && hin == 64) {
        // stride-1 last layer (the 64 x 64 configurations): the encoder's 4-row-tile convolution on the flipped kernel, 1x1 head in the epilogue
        rc = sf_conv5x5_rows4_head_ex(cur, m->deconv_w_frag[i], m->deconv_b[i], m->out_w, m->out_b, dec, R, hin, hin, Ci, Co, m->dec_ks, st);
        if (rc < 0 || rc > 1) return rc;
        if (rc == 0) head_done = true;
      }
      if (rc == 1) {
        if (sd == 1 && m->deconv_w_flipped[i])   // = convolution with the flipped kernel (halo-resident 5x5 path)
          SF_TRY(sf_conv2d_nhwc_f32(cur, m->deconv_w_flipped[i], m->deconv_b[i], nullptr, nxt, R, hin, hin, Ci, Co, m->dec_ks, 1, st));
        else
          SF_TRY(sf_conv_transpose2d_nhwc_f32(cur, m->deconv_w[i], m->deconv_b[i], nxt, R, hin, hin, Ci, Co, m->dec_ks, sd, 1, st));
      }
      hin *= sd;
      float* tmp = cur;
      cur = nxt;
      nxt = tmp;
    }
    if (!head_done)
      SF_TRY(sf_linear_ex(cur, sf_rows(Cl), m->out_w, m->out_b, nullptr, nullptr, 0.f, nullptr, sf_rows(4), 0, dec,
                          sf_rows(4), R * HW, 4, Cl, 0, st));
    SF_TRY(sf_decode_combine_seg_f32(dec, recon_combined + (long long)f0 * 3 * HW,
                                     recons ? recons + (long long)f0 * N * 3 * HW : nullptr,
                                     masks ? masks + (long long)f0 * N * HW : nullptr, seg_i64 ? seg_i64 + (long long)f0 * HW : nullptr,
                                     seg_u8 ? seg_u8 + (long long)f0 * HW : nullptr, fg_thre, slot_max, nf, N, HW, st));
  }
  return 0;
}

}  // extern "C"
