// The SAVi spatial-broadcast decoder under autograd, data gradient only: what the image term of
// SlotFormer's training loss (slotformer.py:313-326, `use_img_recon_loss`) back-propagates through -- the decoder is
// frozen there (slotformer.py:203-210), so no weight gradients are needed.
//
//   forward  (savi.py:504-525): broadcast slots over the dec_res x dec_res grid + position table -> ConvTranspose2d+ReLU
//            stack -> 1x1 head -> masks = softmax over slots, recon = sum_n rgb_n * mask_n.  Every layer output is kept.
//   backward: head / recombination adjoint (one kernel), then per layer ReLU mask + the adjoint of the transposed
//            convolution, which is a plain strided convolution of the gradient with the same weights
//            (sf_conv2d_nhwc_strided_ex on the GEMM core's im2col loader), then the sum over the broadcast grid.
#include "../../include/slotformer_hip.h"
#include "sf_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// d_dec [F*N, HW, 4] from d_recon [F, 3, HW]:  d_rgb_n = m_n d_recon;  d_alpha_n = m_n (rgb_n.d_recon - sum_k m_k rgb_k.d_recon)
__global__ __launch_bounds__(256) void decode_combine_bwd_kernel(const float* __restrict__ dec, const float* __restrict__ d_recon,
                                                                 float* __restrict__ d_dec, int F, int N, int HW) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)F * HW) return;
  const int pix = idx % HW;
  const long long f = idx / HW;
  float g[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) g[c] = d_recon[(f * 3 + c) * HW + pix];
  float mx = -INFINITY;
  for (int n = 0; n < N; ++n) mx = fmaxf(mx, dec[((f * N + n) * HW + pix) * 4 + 3]);
  float sum = 0.f, dot = 0.f;
  for (int n = 0; n < N; ++n) {
    const f32x4 v = *(const f32x4*)(dec + ((f * N + n) * HW + pix) * 4);
    const float e = expf(v[3] - mx);
    sum += e;
    dot += e * (v[0] * g[0] + v[1] * g[1] + v[2] * g[2]);
  }
  const float inv = 1.f / sum;
  dot *= inv;
  for (int n = 0; n < N; ++n) {
    const f32x4 v = *(const f32x4*)(dec + ((f * N + n) * HW + pix) * 4);
    const float m = expf(v[3] - mx) * inv;
    f32x4 o;
    o[0] = m * g[0]; o[1] = m * g[1]; o[2] = m * g[2];
    o[3] = m * ((v[0] * g[0] + v[1] * g[1] + v[2] * g[2]) - dot);
    *(f32x4*)(d_dec + ((f * N + n) * HW + pix) * 4) = o;
  }
}

// d_slots[r][c] = sum_p d_x[r][p][c]   (adjoint of the spatial broadcast; the position table is a constant)
__global__ __launch_bounds__(256) void broadcast_bwd_kernel(const float* __restrict__ dx, float* __restrict__ d_slots, int P, int D) {
  const int r = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += 256) {
    float a = 0.f;
    for (int p = 0; p < P; ++p) a += dx[((long long)r * P + p) * D + c];
    d_slots[(long long)r * D + c] = a;
  }
}

// head weight gradient: part[g][j][c] = sum over the rows of chunk g of d_dec[m][j] * act[m][c]   (j < 4, c < Cl <= 64)
__global__ __launch_bounds__(256) void head_grad_partial_kernel(const float* __restrict__ d_dec, const float* __restrict__ act,
                                                                float* __restrict__ part, long long rows, int rpg, int Cl) {
  __shared__ float red[4][4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.x * rpg, r1 = r0 + rpg < rows ? r0 + rpg : rows;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < Cl)
    for (long long r = r0 + rl; r < r1; r += 4) {
      const f32x4 d = *(const f32x4*)(d_dec + r * 4);
      const float x = act[r * Cl + c];
      a[0] += d[0] * x; a[1] += d[1] * x; a[2] += d[2] * x; a[3] += d[3] * x;
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[rl][j][c] = a[j];
  __syncthreads();
  if (rl == 0 && c < Cl)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      part[((long long)blockIdx.x * 4 + j) * Cl + c] = (red[0][j][c] + red[1][j][c]) + (red[2][j][c] + red[3][j][c]);
}
__global__ __launch_bounds__(256) void head_grad_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int G, int n) {
  const int i = threadIdx.x;
  if (i >= n) return;
  float a = 0.f;
  for (int g = 0; g < G; ++g) a += part[(long long)g * n + i];
  out[i] = a;
}

namespace {
struct DecWs {
  float* act[9];   // act[0]: broadcast input, act[l+1]: output of transposed conv l (post-ReLU)
  float *dec, *ga, *gb, *wout_t, *partial, *dtab;
  float* wf;       // fragment-ordered copy of a stride-1 64 -> 64 layer's weights (forward, then backward-data) for conv_rows4.hip
  size_t total;
};
size_t pad64(size_t n) { return (n + 63) & ~(size_t)63; }

DecWs carve(const sf_savi_decoder* m, int F, float* base) {
  DecWs w;
  memset(&w, 0, sizeof(w));
  size_t off = 0;
  auto take = [&](size_t n) {
    float* p = base ? base + off : nullptr;
    off += pad64(n);
    return p;
  };
  const size_t R = (size_t)F * m->num_slots;
  int h = m->dec_res;
  size_t gmax = R * h * h * m->dec_channels[0];
  w.act[0] = take(gmax);
  for (int l = 0; l < m->dec_layers; ++l) {
    h *= m->dec_strides[l];
    const size_t n = R * h * h * m->dec_channels[l + 1];
    w.act[l + 1] = take(n);
    gmax = n > gmax ? n : gmax;
  }
  w.dec = take(R * h * h * 4);
  w.ga = take(gmax);
  w.gb = take(gmax);
  w.wout_t = take((size_t)4 * m->dec_channels[m->dec_layers]);
  {
    size_t pf = (size_t)513 * 2 * 256;   // bias column sums, head partials
    for (int l = 0; l < m->dec_layers; ++l) {
      const size_t a = sf_conv_wgrad_partial_floats(m->dec_channels[l], m->dec_ks);
      pf = a > pf ? a : pf;
    }
    w.partial = take(pf);
  }
  w.dtab = take((size_t)m->dec_res * m->dec_res * m->slot_size);
  w.wf = take((sf_conv_frag_bytes(64, 64, 5) + 3) / 4);
  w.total = off;
  return w;
}

// stride-1 5x5 64 -> 64 layer on a 64-pixel-wide map: the 4-row-tile kernel on a fragment-ordered copy of the OHWI weights (conv_rows4.hip: the same
// products in the same order as the halo kernel behind sf_conv2d_nhwc_f32); 1 = not that shape / arithmetic mode
int conv5_rows4(const float* in, const float* w_ohwi, float* w_frag, const float* bias, const float* add, float* out, int R, int h, int cin, int cout,
                int ks, int relu, hipStream_t st) {
  if (!w_frag || h != 64 || sf_get_precision() != 1 || !sf_conv_frag_bytes(cout, cin, ks)) return 1;
  SF_TRY(sf_pack_conv_frag_weights(w_ohwi, w_frag, cout, cin, ks, st));
  return sf_conv5x5_rows4_ex(in, w_frag, bias, add, out, R, h, h, cin, cout, ks, relu, st);
}

int check(const sf_savi_decoder* m, int F) {
  SF_REQUIRE(m, "null model");
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
  return 0;
}
}  // namespace

extern "C" {

size_t sf_savi_decode_train_workspace_bytes(const sf_savi_decoder* m, int F) {
  if (check(m, F) != 0) return 0;
  return carve(m, F, nullptr).total * sizeof(float) + 256;
}

int sf_savi_decode_train_fwd_f32(const sf_savi_decoder* m, const float* slots, float* recon_combined, float* recons, float* masks,
                                 int F, void* ws, size_t ws_bytes, void* stream) {
  SF_TRY(check(m, F));
  SF_REQUIRE(slots && recon_combined && ws, "null pointer");
  const DecWs w = carve(m, F, (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255));
  SF_REQUIRE(w.total * sizeof(float) + 256 <= ws_bytes, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int N = m->num_slots, D = m->slot_size, R = F * N, HW = m->resolution * m->resolution;
  SF_TRY(sf_slot_broadcast_f32(slots, m->pos_table, w.act[0], R, m->dec_res * m->dec_res, D, st));
  int hin = m->dec_res;
  for (int l = 0; l < m->dec_layers; ++l) {
    if (m->dec_strides[l] == 1 && m->deconv_w_flipped[l]) {   // = convolution with the flipped kernel (4-row-tile / halo-resident 5x5 paths)
      int rc = conv5_rows4(w.act[l], m->deconv_w_flipped[l], w.wf, m->deconv_b[l], nullptr, w.act[l + 1], R, hin, m->dec_channels[l],
                           m->dec_channels[l + 1], m->dec_ks, 1, st);
      if (rc == 1)
        rc = sf_conv2d_nhwc_f32(w.act[l], m->deconv_w_flipped[l], m->deconv_b[l], nullptr, w.act[l + 1], R, hin, hin, m->dec_channels[l],
                                m->dec_channels[l + 1], m->dec_ks, 1, st);
      SF_TRY(rc);
    } else
      SF_TRY(sf_conv_transpose2d_nhwc_f32(w.act[l], m->deconv_w[l], m->deconv_b[l], w.act[l + 1], R, hin, hin, m->dec_channels[l],
                                          m->dec_channels[l + 1], m->dec_ks, m->dec_strides[l], 1, st));
    hin *= m->dec_strides[l];
  }
  const int Cl = m->dec_channels[m->dec_layers];
  SF_TRY(sf_linear_ex(w.act[m->dec_layers], sf_rows(Cl), m->out_w, m->out_b, nullptr, nullptr, 0.f, nullptr, sf_rows(4), 0, w.dec,
                      sf_rows(4), R * HW, 4, Cl, 0, st));
  return sf_decode_combine_f32(w.dec, recon_combined, recons, masks, F, N, HW, st);
}

int sf_savi_decode_train_bwd_f32(const sf_savi_decoder* m, const float* const* deconv_w_bwd, const float* d_recon, float* d_slots,
                                 const float* pos_grid, const sf_savi_decoder_grads* g_out, int F, void* ws, size_t ws_bytes,
                                 void* stream) {
  SF_TRY(check(m, F));
  SF_REQUIRE(deconv_w_bwd && d_recon && d_slots && ws, "null pointer");
  SF_REQUIRE(g_out == nullptr || pos_grid != nullptr, "weight gradients need the position grid");
  if (g_out)
    for (int l = 0; l < m->dec_layers; ++l)
      SF_REQUIRE(m->dec_channels[l + 1] == 64 && m->dec_channels[l] % 64 == 0 && m->dec_channels[m->dec_layers] <= 64,
                 "decoder weight gradients need 64-channel layers (the reference decoder)");
  const DecWs w = carve(m, F, (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255));
  SF_REQUIRE(w.total * sizeof(float) + 256 <= ws_bytes, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int N = m->num_slots, D = m->slot_size, R = F * N, HW = m->resolution * m->resolution, L = m->dec_layers;
  const int Cl = m->dec_channels[L];
  {
    const long long total = (long long)F * HW;
    hipLaunchKernelGGL(decode_combine_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w.dec, d_recon, w.gb, F, N,
                       HW);
    SF_CHECK_LAUNCH();
  }
  if (g_out) {   // 1x1 head: weight [4, Cl] and bias [4]
    const long long rows = (long long)R * HW;
    int G = 512;
    const int rpg = (int)((rows + G - 1) / G);
    G = (int)((rows + rpg - 1) / rpg);
    hipLaunchKernelGGL(head_grad_partial_kernel, dim3(G), dim3(256), 0, st, w.gb, w.act[L], w.partial, rows, rpg, Cl);
    SF_CHECK_LAUNCH();
    hipLaunchKernelGGL(head_grad_reduce_kernel, dim3(1), dim3(256), 0, st, w.partial, g_out->out_w, G, 4 * Cl);
    SF_CHECK_LAUNCH();
    SF_TRY(sf_grad_bias_ex(w.gb, g_out->out_b, rows, 4, w.partial, st));
  }
  // head: d_act[L] = d_dec . W_out   (W_out [4, Cl]; the GEMM core wants its transpose as the weight operand), gated in the
  // epilogue by the ReLU of the last transposed conv (mask = its saved output)
  SF_TRY(sf_transpose_ex(m->out_w, w.wout_t, 4, Cl, st));
  SF_TRY(sf_linear_masked_ex(w.gb, w.wout_t, w.act[L], 1.f, w.ga, (long long)R * HW, Cl, 4, st));
  float* g = w.ga;   // gradient w.r.t. the pre-activation output of layer l
  float* o = w.gb;
  int h = m->resolution;
  for (int l = L - 1; l >= 0; --l) {
    SF_REQUIRE(deconv_w_bwd[l] != nullptr, "null backward weight");
    if (g_out) {   // transposed-conv weight [C_l, C_{l+1}, k, k] and bias: A = the layer input, X = this gradient on the finer grid
      const int hin = h / m->dec_strides[l];
      SF_TRY(sf_grad_bias_ex(g, g_out->deconv_b[l], (long long)R * h * h, m->dec_channels[l + 1], w.partial, st));
      SF_TRY(sf_conv_wgrad_ex(w.act[l], m->dec_channels[l], hin, hin, g, h, h, m->dec_strides[l], m->dec_ks, (long long)R * hin * hin,
                              g_out->deconv_w[l], w.partial, st));
    }
    // adjoint of the transposed conv = strided conv of the gradient; its output is gated by the ReLU of the layer below
    // (act[0] is the broadcast input: no ReLU there)
    int rc = 1;
    if (m->dec_strides[l] == 1 && l >= 1)
      rc = conv5_rows4(g, deconv_w_bwd[l], w.wf, nullptr, w.act[l], o, R, h, m->dec_channels[l + 1], m->dec_channels[l], m->dec_ks, 2, st);
    if (rc == 1)
      rc = sf_conv2d_nhwc_strided_ex(g, deconv_w_bwd[l], nullptr, o, R, h, h, m->dec_channels[l + 1], m->dec_channels[l], m->dec_ks,
                                     m->dec_strides[l], 0, st, l >= 1 ? w.act[l] : nullptr);
    SF_TRY(rc);
    h /= m->dec_strides[l];
    float* t = g;
    g = o;
    o = t;
  }
  if (g_out) SF_TRY(sf_pos_dense_grad_ex(g, R, m->dec_res * m->dec_res, D, pos_grid, g_out->pos_w, g_out->pos_b, w.dtab, st));
  hipLaunchKernelGGL(broadcast_bwd_kernel, dim3(R), dim3(256), 0, st, g, d_slots, m->dec_res * m->dec_res, D);
  SF_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
