// Transformer slot predictor + LSTM wrapper in ONE launch per frame (predictor.py:20-44 TransformerPredictor, :76-135
// RNNPredictorWrapper; called from savi.py:393-398 once per time step):
//
//   x = prev_slots [B][N][D]                                   N <= 8 slots per video: a workgroup holds 32 / N videos (<= 32 rows)
//   per layer (pre-LN nn.TransformerEncoderLayer over the N slots of a video):
//     x += out_proj(MHA(LN1(x)));  x += lin2(relu(lin1(LN2(x))))
//   gates = W_ih x + b_ih + W_hh h + b_hh;  c' = sigmoid(f) c + sigmoid(i) tanh(g);  h' = sigmoid(o) tanh(c')      (nn.LSTM, one step)
//   out = proj(h')
//
// The unfused chain is 8 launches for the two layers + 4 for the LSTM wrapper per frame, each a round trip of a few KB of rows;
// here the rows stay in LDS / registers and the only traffic is the weight stream: every product runs "transposed" on the
// split-bf16 MFMA (weights, packed once per weight version in fragment order by sf_pack_linear_weights, as the A operand
// straight from memory; the <= 32 rows as the B operand from LDS planes), with the fragments of the next chunk requested
// before the current one is consumed.  A workgroup streams all the weights (3.3 MB at slot size 128) through one CU, so the launch
// is bound by that CU's ingest rate, not by arithmetic: 62 us per launch (53 GB/s; 83 us with one chunk in flight per wave instead
// of three) against ~110 us for the 12 launches it replaces (C1, B = 32: encode 2.88 vs 3.02 ms; B = 4: 1.34 vs 1.50 ms).
#include "stream_mfma.h"

namespace {

constexpr int PS_NT = 512, PS_ROWS = 32, PS_MAXL = 4;

struct PsLayer {
  const uint4 *wqkv, *wo, *w1, *w2;   // packed [3D][D], [D][D], [F][D], [D][F]
  const float *bqkv, *bo, *b1, *b2, *n1g, *n1b, *n2g, *n2b;
};

struct PsArgs {
  const float* prev;   // [R][D]
  PsLayer layer[PS_MAXL];
  int nlayers;
  const uint4 *wih, *whh, *wproj;   // packed [4H][D], [4H][H], [D][H]; wih NULL: no LSTM wrapper
  const float *bih, *bhh, *bproj;
  float *h, *c;        // [R][H] in / out
  float* out;          // [R][D]
  int B, N, VP;        // videos, slots per video, videos per workgroup (VP * N <= 32)
  float eps;
};

template <int D, int F, int H>
struct PsCfg {
  static constexpr int FC = F / 2;                       // hidden chunk held in LDS at a time
  static constexpr int KP = (D > H ? D : H) + 8;         // bf16 pitch of the input planes
  static constexpr int XP = D + 4;                       // f32 pitch of the residual rows
  static constexpr int QP = 3 * D + 4;                   // f32 pitch of the q|k|v rows
  static constexpr int HP = FC + 8, SP = H + 8;          // bf16 pitches of the hidden-chunk / previous-state planes
  static constexpr size_t big_qkv = (size_t)PS_ROWS * QP * 4, big_hid = (size_t)2 * PS_ROWS * HP * 2, big_st = (size_t)2 * PS_ROWS * SP * 2;
  static constexpr size_t big_lstm = big_st + (size_t)PS_ROWS * SP * 2;   // previous-state planes + the low plane of the new state
  static constexpr size_t big0 = big_qkv > big_hid ? big_qkv : big_hid;
  static constexpr size_t big = big0 > big_lstm ? big0 : big_lstm;
  static constexpr size_t lds = (size_t)PS_ROWS * XP * 4 + (size_t)2 * PS_ROWS * KP * 2 + big;
};

template <int D, int F, int H, int NH>
__global__ __launch_bounds__(PS_NT) void pred_step_kernel(PsArgs a) {
  using C = PsCfg<D, F, H>;
  constexpr int FC = C::FC, KP = C::KP, XP = C::XP, QP = C::QP, HP = C::HP, SP = C::SP;
  constexpr int HD = D / NH, SUB = 16 / NH, CPT = HD / SUB;   // attention: 16 threads per row = NH heads x SUB channel groups
  static_assert(D % 64 == 0 && F % 64 == 0 && H % 32 == 0 && 16 % NH == 0 && CPT % 4 == 0 && CPT <= 16 && F >= 2 * D && H == 2 * D, "shape");
  constexpr int NQ = D / 64;                                   // f32x4 per thread of a row (16 threads per row)
  extern __shared__ __attribute__((aligned(16))) float ps_lds[];
  float* XF = ps_lds;                                   // [32][XP]  residual stream
  __bf16* Ph = (__bf16*)(XF + PS_ROWS * XP);            // [32][KP]  input planes of the current product
  __bf16* Pl = Ph + PS_ROWS * KP;
  char* BIG = (char*)(Pl + PS_ROWS * KP);
  float* QKV = (float*)BIG;                             // [32][QP]
  __bf16* Hh = (__bf16*)BIG;                            // [32][HP] relu hidden chunk
  __bf16* Hl = Hh + PS_ROWS * HP;
  __bf16* Sh = (__bf16*)BIG;                            // [32][SP] previous LSTM state
  __bf16* Sl = Sh + PS_ROWS * SP;
  __bf16* Nh = (__bf16*)XF;                             // [32][SP] new LSTM state: high plane over the (then dead) residual rows
  __bf16* Nl = Sl + PS_ROWS * SP;                       //          (32 (D + 4) floats = 32 (2 D + 8) bf16), low plane behind the S planes
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int tok = lane & 31, kg = lane >> 5;
  const int RW = a.VP * a.N, R = a.B * a.N;
  const int row0 = blockIdx.x * RW;
  const int nrows = min(RW, R - row0);                  // real rows of this workgroup
  const int ur = t >> 4, c4 = t & 15;                   // thread = (row, 16 threads per row)
  const int urc = min(ur, nrows - 1);
  PsBuf s;
  const bool lstm = a.wih != nullptr;

  // first fragments of the first product, then the rows
  {
    const float* src = a.prev + (long long)(row0 + urc) * D;
#pragma unroll
    for (int i = 0; i < NQ; ++i) *(f32x4*)(XF + ur * XP + 4 * (c4 + 16 * i)) = *(const f32x4*)(src + 4 * (c4 + 16 * i));
  }
  // LayerNorm of row ur (held by 16 consecutive lanes) of XF -> the input planes
  // (gamma / beta are requested one phase ahead, ln_fetch: a global load issued behind the weight requests of the ring would wait
  //  for all of them -- the loads retire in order)
  f32x4 gg[NQ], bb[NQ];
  auto ln_fetch = [&](const float* g, const float* b) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      gg[i] = *(const f32x4*)(g + 4 * (c4 + 16 * i));
      bb[i] = *(const f32x4*)(b + 4 * (c4 + 16 * i));
    }
  };
  auto layer_norm = [&]() {
    f32x4 v[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) v[i] = *(const f32x4*)(XF + ur * XP + 4 * (c4 + 16 * i));
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i) sm += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mu = sf_sum16(sm) / (float)D;
    float vs = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      v[i] = v[i] - mu;
      vs += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
    }
    const float rs = 1.0f / sqrtf(sf_sum16(vs) / (float)D + a.eps);
#pragma unroll
    for (int i = 0; i < NQ; ++i) ps_split4(Ph, Pl, ur * KP + 4 * (c4 + 16 * i), v[i] * rs * gg[i] + bb[i]);
  };

  ln_fetch(a.layer[0].n1g, a.layer[0].n1b);
  if (wave < 3 * D / 32) ps_prime<PsChunk<D / 16>::CH>(s, a.layer[0].wqkv, 3 * D / 32, wave, 0, lane);
  __syncthreads();
  for (int l = 0; l < a.nlayers; ++l) {
    const PsLayer& w = a.layer[l];
    // ---- LN1 -> planes; q|k|v = W LN1(x) + b -> QKV (f32) ----
    layer_norm();
    ln_fetch(w.n2g, w.n2b);
    __syncthreads();
    {
      constexpr int KS = D / 16, NB = 3 * D / 32;
      for (int nb = wave; nb < NB; nb += 8) {
        f32x4 bv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[g] = *(const f32x4*)(w.bqkv + 32 * nb + 8 * g + 4 * kg);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const bool more = nb + 8 < NB;
        // behind the last block of this wave: the first out-projection fragments (waves < D / 32 run that product)
        const bool nxt = more || wave < D / 32;
        ps_block<KS, PsChunk<KS>::CH>(acc, s, w.wqkv, NB, nb, 0, nxt ? (more ? w.wqkv : w.wo) : nullptr, more ? NB : D / 32,
                                      more ? nb + 8 : wave, 0, Ph, Pl, KP, lane);
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(f32x4*)(QKV + tok * QP + 32 * nb + 8 * g + 4 * kg) =
              f32x4{acc[4 * g] + bv[g][0], acc[4 * g + 1] + bv[g][1], acc[4 * g + 2] + bv[g][2], acc[4 * g + 3] + bv[g][3]};
      }
    }
    __syncthreads();
    // ---- attention over the N slots of a video: thread = (row, head, channel group); scores of the row against its video's keys ----
    {
      const int head = c4 / SUB, sub = c4 % SUB;
      const int v0 = (urc / a.N) * a.N;                 // first row of this row's video
      const int co = head * HD + sub * CPT;
      f32x4 q[CPT / 4];
#pragma unroll
      for (int i = 0; i < CPT / 4; ++i) q[i] = *(const f32x4*)(QKV + urc * QP + co + 4 * i);
      float sc[8];
      const float scale = 1.0f / sqrtf((float)HD);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int jr = v0 + min(j, a.N - 1);
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < CPT / 4; ++i) {
          const f32x4 kk = *(const f32x4*)(QKV + jr * QP + D + co + 4 * i);
          d += (q[i][0] * kk[0] + q[i][1] * kk[1]) + (q[i][2] * kk[2] + q[i][3] * kk[3]);
        }
#pragma unroll
        for (int o = 1; o < SUB; o <<= 1) d += __shfl_xor(d, o, 64);
        sc[j] = j < a.N ? d * scale : -INFINITY;
      }
      float mx = sc[0];
#pragma unroll
      for (int j = 1; j < 8; ++j) mx = fmaxf(mx, sc[j]);
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sc[j] = expf(sc[j] - mx);
        sum += sc[j];
      }
      const float inv = 1.0f / sum;
      f32x4 o[CPT / 4];
#pragma unroll
      for (int i = 0; i < CPT / 4; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int jr = v0 + min(j, a.N - 1);
#pragma unroll
        for (int i = 0; i < CPT / 4; ++i) o[i] += sc[j] * *(const f32x4*)(QKV + jr * QP + 2 * D + co + 4 * i);
      }
#pragma unroll
      for (int i = 0; i < CPT / 4; ++i) ps_split4(Ph, Pl, ur * KP + co + 4 * i, o[i] * inv);
    }
    __syncthreads();
    // ---- x += Wo O + bo (waves < D / 32: one column block each); the others request their first lin1 fragments ----
    constexpr int NB1 = FC / 32;
    if (wave < D / 32) {
      constexpr int KS = D / 16;
      f32x4 bv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) bv[g] = *(const f32x4*)(w.bo + 32 * wave + 8 * g + 4 * kg);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      ps_block<KS, PsChunk<KS>::CH>(acc, s, w.wo, D / 32, wave, 0, wave < NB1 ? w.w1 : nullptr, F / 32, wave, 0, Ph, Pl, KP, lane);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float* xp = XF + tok * XP + 32 * wave + 8 * g + 4 * kg;
        const f32x4 x = *(const f32x4*)xp;
        *(f32x4*)xp = f32x4{x[0] + (acc[4 * g] + bv[g][0]), x[1] + (acc[4 * g + 1] + bv[g][1]), x[2] + (acc[4 * g + 2] + bv[g][2]),
                            x[3] + (acc[4 * g + 3] + bv[g][3])};
      }
    } else if (wave < NB1) {
      ps_prime<PsChunk<D / 16>::CH>(s, w.w1, F / 32, wave, 0, lane);
    }
    __syncthreads();
    // ---- LN2 -> planes; FFN in two hidden chunks: hidden = relu(W1 LN2 + b1) -> H planes, y += W2 hidden ----
    layer_norm();
    if (l + 1 < a.nlayers) ln_fetch(a.layer[l + 1].n1g, a.layer[l + 1].n1b);
    __syncthreads();
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll 1
    for (int fc = 0; fc < 2; ++fc) {
      {
        constexpr int KS = D / 16;
        for (int j = wave; j < NB1; j += 8) {
          const int nb = fc * NB1 + j;
          f32x4 bv[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) bv[g] = *(const f32x4*)(w.b1 + 32 * nb + 8 * g + 4 * kg);
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          const bool more = j + 8 < NB1;
          if (more)
            ps_block<KS, PsChunk<KS>::CH>(acc, s, w.w1, F / 32, nb, 0, w.w1, F / 32, nb + 8, 0, Ph, Pl, KP, lane);
          else   // behind the last block: the first lin2 fragments of this chunk
            ps_block<KS, PsChunk<FC / 16>::CH>(acc, s, w.w1, F / 32, nb, 0, wave < D / 32 ? w.w2 : nullptr, D / 32, wave, fc * (FC / 16), Ph, Pl,
                                               KP, lane);
#pragma unroll
          for (int g = 0; g < 4; ++g)
            ps_split4(Hh, Hl, tok * HP + 32 * j + 8 * g + 4 * kg,
                      f32x4{fmaxf(acc[4 * g] + bv[g][0], 0.f), fmaxf(acc[4 * g + 1] + bv[g][1], 0.f), fmaxf(acc[4 * g + 2] + bv[g][2], 0.f),
                            fmaxf(acc[4 * g + 3] + bv[g][3], 0.f)});
        }
      }
      __syncthreads();
      if (wave < D / 32) {
        constexpr int KS = FC / 16;
        // behind this chunk's lin2: the first lin1 fragments of the next chunk, or of what follows the layer
        const uint4* pn = nullptr;
        int nbn = 0, nblocksn = 1;
        if (fc == 0 && wave < NB1) {
          pn = w.w1; nblocksn = F / 32; nbn = NB1 + wave;
        } else if (fc == 1 && l + 1 < a.nlayers) {
          pn = a.layer[l + 1].wqkv; nblocksn = 3 * D / 32; nbn = wave;
        } else if (fc == 1 && lstm) {
          pn = a.wih; nblocksn = 4 * H / 32; nbn = wave;
        }
        ps_block<KS, PsChunk<D / 16>::CH>(acc2, s, w.w2, D / 32, wave, fc * KS, pn, nblocksn, nbn, 0, Hh, Hl, HP, lane);
      } else if (fc == 0 && wave < NB1) {
        ps_prime<PsChunk<D / 16>::CH>(s, w.w1, F / 32, NB1 + wave, 0, lane);
      } else if (fc == 1) {
        if (l + 1 < a.nlayers) {
          if (wave < 3 * D / 32) ps_prime<PsChunk<D / 16>::CH>(s, a.layer[l + 1].wqkv, 3 * D / 32, wave, 0, lane);
        } else if (lstm && wave < H / 32)
          ps_prime<PsChunk<D / 16>::CH>(s, a.wih, 4 * H / 32, wave, 0, lane);
      }
      __syncthreads();
    }
    if (wave < D / 32) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *(const f32x4*)(w.b2 + 32 * wave + 8 * g + 4 * kg);
        float* xp = XF + tok * XP + 32 * wave + 8 * g + 4 * kg;
        const f32x4 x = *(const f32x4*)xp;
        *(f32x4*)xp = f32x4{x[0] + (acc2[4 * g] + bv[0]), x[1] + (acc2[4 * g + 1] + bv[1]), x[2] + (acc2[4 * g + 2] + bv[2]),
                            x[3] + (acc2[4 * g + 3] + bv[3])};
      }
    }
    __syncthreads();
  }

  if (!lstm) {
    if (ur < nrows) {
      float* dst = a.out + (long long)(row0 + ur) * D;
#pragma unroll
      for (int i = 0; i < NQ; ++i) *(f32x4*)(dst + 4 * (c4 + 16 * i)) = *(const f32x4*)(XF + ur * XP + 4 * (c4 + 16 * i));
    }
    return;
  }

  // ---- LSTM step: the predictor output and the previous state as planes ----
#pragma unroll
  for (int i = 0; i < NQ; ++i) ps_split4(Ph, Pl, ur * KP + 4 * (c4 + 16 * i), *(const f32x4*)(XF + ur * XP + 4 * (c4 + 16 * i)));
  {
    const float* hs = a.h + (long long)(row0 + urc) * H;
    for (int i = c4; i < H / 4; i += 16) ps_split4(Sh, Sl, ur * SP + 4 * i, *(const f32x4*)(hs + 4 * i));
  }
  __syncthreads();
  // wave = hidden block jb: the four gate blocks (i, f, g, o: rows jb, H/32 + jb, ...) of W_ih (K = D) and W_hh (K = H)
  constexpr int NJ = H / 32, KSD = D / 16, KSH = H / 16;
  constexpr int CHD = PsChunk<KSD>::CH, CHH = PsChunk<KSH>::CH;
  const int grow = row0 + min(tok, nrows - 1);
  // combined bias b_ih + b_hh of gate q for this lane's columns of hidden block jb -- requested one gate AHEAD of its use (a global
  // load issued behind the ring's weight requests waits for all of them: the loads retire in order)
  // (buffer loads: scalar descriptor + scalar block offset, the vector part of the address is 16 kg)
  const __amdgpu_buffer_rsrc_t r_bih = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bih), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_bhh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bhh), 0, 0x7fffffff, 0x00020000);
  auto gate_bias = [&](f32x4 (&b)[4], int q, int jb) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int so = (q * H + 32 * jb + 8 * g) * 4;
      b[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_bih, kg * 16, so, 0)) +
             __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_bhh, kg * 16, so, 0));
    }
  };
  const __amdgpu_buffer_rsrc_t r_c = __builtin_amdgcn_make_buffer_rsrc(a.c, 0, 0x7fffffff, 0x00020000);
#pragma unroll 1
  for (int jb = wave; jb < NJ; jb += 8) {
    // gate order i, g, f, o: every gate's accumulator is folded into the running values right behind its two products (all four
    // accumulators live at once spilled at slot size 192)
    f32x16 ig, cn;
#pragma unroll 1
    for (int qi = 0; qi < 4; ++qi) {   // (a run-time loop: unrolled, the addresses of all eight products of a block stayed live)
      const int q = qi == 1 ? 2 : qi == 2 ? 1 : qi, qn = qi == 0 ? 2 : qi == 1 ? 1 : qi == 2 ? 3 : 0;
      const bool last = (qi == 3);
      const bool more = jb + 8 < NJ;
      // bias (and, for gate f, the previous cell state) requested BEFORE the two products, used behind them
      f32x4 bq[4], cp[4];
      gate_bias(bq, q, jb);
      if (qi == 2) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          cp[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_c, (grow * H + 4 * kg) * 4, (32 * jb + 8 * g) * 4, 0));
      }
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      ps_block<KSD, CHH>(acc, s, a.wih, 4 * NJ, q * NJ + jb, 0, a.whh, 4 * NJ, q * NJ + jb, 0, Ph, Pl, KP, lane);
      const uint4* pn = !last ? a.wih : (more ? a.wih : (wave < D / 32 ? a.wproj : nullptr));
      const int nblocksn = (!last || more) ? 4 * NJ : D / 32;
      const int nbn = !last ? qn * NJ + jb : (more ? jb + 8 : wave);   // (qn = 0 behind the last gate: gate i of the next block)
      // (the chunk size of the next product: CHD for W_ih, CHH for the projection, K = H)
      if (!last || more)
        ps_block<KSH, CHD>(acc, s, a.whh, 4 * NJ, q * NJ + jb, 0, pn, nblocksn, nbn, 0, Sh, Sl, SP, lane);
      else
        ps_block<KSH, CHH>(acc, s, a.whh, 4 * NJ, q * NJ + jb, 0, pn, nblocksn, nbn, 0, Sh, Sl, SP, lane);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cc = 32 * jb + 8 * g + 4 * kg;
        if (q == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ig[4 * g + e] = ps_sigmoid(acc[4 * g + e] + bq[g][e]);
        } else if (q == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ig[4 * g + e] *= ps_tanh(acc[4 * g + e] + bq[g][e]);
        } else if (q == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) cn[4 * g + e] = ps_sigmoid(acc[4 * g + e] + bq[g][e]) * cp[g][e] + ig[4 * g + e];
        } else {
          f32x4 hn, cv;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            cv[e] = cn[4 * g + e];
            hn[e] = ps_sigmoid(acc[4 * g + e] + bq[g][e]) * ps_tanh(cv[e]);
          }
          if (tok < nrows) {
            *(f32x4*)(a.c + (long long)grow * H + cc) = cv;
            *(f32x4*)(a.h + (long long)grow * H + cc) = hn;
          }
          ps_split4(Nh, Nl, tok * SP + cc, hn);
        }
      }
    }
  }
  __syncthreads();
  // ---- out = Wp h' + bp ----
  if (wave < D / 32) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    ps_block<KSH, CHH>(acc, s, a.wproj, D / 32, wave, 0, nullptr, 1, 0, 0, Nh, Nl, SP, lane);
    if (tok < nrows) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cc = 32 * wave + 8 * g + 4 * kg;
        const f32x4 bv = *(const f32x4*)(a.bproj + cc);
        *(f32x4*)(a.out + (long long)grow * D + cc) =
            f32x4{acc[4 * g] + bv[0], acc[4 * g + 1] + bv[1], acc[4 * g + 2] + bv[2], acc[4 * g + 3] + bv[3]};
      }
    }
  }
}

template <int D, int F, int H, int NH>
int launch_pred_step(const PsArgs& a, hipStream_t st) {
  constexpr size_t lds = PsCfg<D, F, H>::lds;
  static_assert(lds <= 160 * 1024, "predictor step: LDS budget");
  auto kern = pred_step_kernel<D, F, H, NH>;
  SF_TRY(sf_ensure_dyn_lds((const void*)kern, lds));
  const int grid = (a.B + a.VP - 1) / a.VP;
  const double R = (double)a.B * a.N;
  sf_prof_begin(SF_K_LINEAR, st, 2.0 * R * (a.nlayers * (4.0 * D * D + 2.0 * D * F) + (a.wih ? 4.0 * H * (D + H) + (double)D * H : 0.0)));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(PS_NT), lds, st, a);
  sf_prof_end(SF_K_LINEAR, st);
  SF_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// One launch for the Transformer predictor (pre-LN layers, 4 heads) and its LSTM wrapper.  packed: HOST array of
// sf_pack_linear_weights copies, [4 l + 0..3] = layer l's in_proj_weight [3D][D], out_proj.weight [D][D], linear1.weight [F][D],
// linear2.weight [D][F]; then (with the wrapper) weight_ih_l0 [4H][D], weight_hh_l0 [4H][H], out_projector.weight [D][H].
// Returns 1 when the shape is not covered (the caller runs the unfused chain).
int sf_pred_step_ex(const float* prev, const sf_tfm_layer* layers, int nlayers, int heads, int ffn, int norm_first, const void* const* packed,
                    const float* lstm_b_ih, const float* lstm_b_hh, const float* proj_b, int hidden, float* lstm_h, float* lstm_c,
                    float* out, int B, int N, int D, float eps, hipStream_t st) {
  if (!packed || !norm_first || nlayers < 1 || nlayers > PS_MAXL || N < 1 || N > 8 || heads != 4 || B <= 0) return 1;
  const bool lstm = lstm_h != nullptr;
  for (int i = 0; i < 4 * nlayers + (lstm ? 3 : 0); ++i)
    if (!packed[i]) return 1;
  PsArgs a;
  a.prev = prev;
  a.nlayers = nlayers;
  for (int l = 0; l < nlayers; ++l) {
    PsLayer& w = a.layer[l];
    w.wqkv = (const uint4*)packed[4 * l];
    w.wo = (const uint4*)packed[4 * l + 1];
    w.w1 = (const uint4*)packed[4 * l + 2];
    w.w2 = (const uint4*)packed[4 * l + 3];
    w.bqkv = layers[l].in_proj_b; w.bo = layers[l].out_proj_b; w.b1 = layers[l].lin1_b; w.b2 = layers[l].lin2_b;
    w.n1g = layers[l].norm1_g; w.n1b = layers[l].norm1_b; w.n2g = layers[l].norm2_g; w.n2b = layers[l].norm2_b;
  }
  a.wih = lstm ? (const uint4*)packed[4 * nlayers] : nullptr;
  a.whh = lstm ? (const uint4*)packed[4 * nlayers + 1] : nullptr;
  a.wproj = lstm ? (const uint4*)packed[4 * nlayers + 2] : nullptr;
  a.bih = lstm_b_ih; a.bhh = lstm_b_hh; a.bproj = proj_b;
  a.h = lstm_h; a.c = lstm_c; a.out = out;
  a.B = B; a.N = N; a.VP = PS_ROWS / N; a.eps = eps;
  const int H = lstm ? hidden : 0;
  if (D == 128 && ffn == 512 && (H == 256 || !lstm)) return launch_pred_step<128, 512, 256, 4>(a, st);
  // slot size 192 (ffn 768, LSTM 384: 7.4 MB of weights per workgroup) measured 174 us per launch against ~180 us for the unfused
  // chain, whose GEMMs stream the weights through many CUs at once, and the C4 pipeline lost 4 % with it: not dispatched
  if (D == 64 && ffn == 128 && (H == 128 || !lstm)) return launch_pred_step<64, 128, 128, 4>(a, st);
  return 1;
}
