/*
 * taco_oracle.c - CPU restatement of the reference's Tacotron 2 mel-inference arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker the HIP kernels are compared against; it is
 * compiled into oracle/libtaco_oracle.so and may be loaded only by tests/, by
 * __graft_entry__.smoke() and by the cpu_baseline leg of bench.py.  The product never calls it.
 *
 * Parity pinning: the functions below are validated against the reference itself
 * (stefantaubert/tacotron, imported in the build container by tools/make_golden.py) through the
 * committed fixtures in tests/golden/: outputs of Tacotron2.inference on synthetic
 * reference-format checkpoints.  The reference's own known-answer test
 * (src/tacotron_tests/synthesizer_py/test_Synthesizer.py:11-46) needs a checkpoint download that
 * is unavailable offline, so it is not one of the pins.
 *
 * All arithmetic is fp32 like the reference's (pure fp32 eager PyTorch).  Dot products use 16
 * interleaved partial sums, a fixed order of this file's own choosing: the reference's order is
 * whatever its BLAS picks, so agreement with it is to rounding (about 1e-6), not bitwise.
 *
 * Each function cites the reference lines it restates (paths relative to the reference root).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/taco.h"

#define BN_EPS 1e-5f

static float dotf(const float* a, const float* b, int n) {
  float acc[16];
  for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
  int i = 0;
  for (; i + 16 <= n; i += 16)
    for (int j = 0; j < 16; ++j) acc[j] += a[i + j] * b[i + j];
  for (int j = 0; i < n; ++i, ++j) acc[j] += a[i] * b[i];
  for (int s = 8; s >= 1; s >>= 1)
    for (int j = 0; j < s; ++j) acc[j] += acc[j + s];
  return acc[0];
}

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

int taco_oracle_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void taco_oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* Conv1d (zero padding (k-1)/2, stride 1) + eval-mode BatchNorm1d + activation on one utterance.
 * x [c_in][T], y [c_out][T]; only columns < len are computed, the rest are written as zero, and
 * input columns >= len are never read (they count as padding).
 * reference: ConvNorm src/tacotron/layers.py:17-35; BatchNorm1d in eval mode uses running
 * statistics, y = (x - mean) / sqrt(var + eps) * weight + bias with eps = 1e-5.
 * act: 0 none, 1 relu, 2 tanh. */
static void conv_bn_act(const float* x, float* y, int c_in, int c_out, int k, int T, int len,
                        const float* w, const float* b, const float* bn_w, const float* bn_b,
                        const float* bn_mean, const float* bn_var, int act) {
  const int pad = (k - 1) / 2;
#pragma omp parallel for schedule(static)
  for (int co = 0; co < c_out; ++co) {
    const float invstd = 1.0f / sqrtf(bn_var[co] + BN_EPS);
    const float alpha = invstd * bn_w[co];
    const float beta = bn_b[co] - bn_mean[co] * alpha;
    float* yr = y + (size_t)co * T;
    for (int t = 0; t < T; ++t) yr[t] = 0.0f;
    /* taps accumulate in (ci, kk) order into each output column */
    for (int ci = 0; ci < c_in; ++ci) {
      const float* wr = w + ((size_t)co * c_in + ci) * k;
      const float* xr = x + (size_t)ci * T;
      for (int kk = 0; kk < k; ++kk) {
        const int sh = kk - pad;
        const int t0 = sh < 0 ? -sh : 0;
        const int t1 = sh > 0 ? len - sh : len;
        const float wv = wr[kk];
        for (int t = t0; t < t1; ++t) yr[t] += wv * xr[t + sh];
      }
    }
    for (int t = 0; t < len; ++t) {
      float v = (yr[t] + b[co]) * alpha + beta;
      if (act == 1) v = v > 0.0f ? v : 0.0f;
      else if (act == 2) v = tanhf(v);
      yr[t] = v;
    }
  }
}

/* One LSTM cell update; gate rows ordered i, f, g, o (torch.nn.LSTM / LSTMCell).
 * gates = (W_ih x + b_ih) + (W_hh h + b_hh); c' = sig(f) c + sig(i) tanh(g); h' = sig(o) tanh(c'). */
static void lstm_cell(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                      const float* x, int n_x, float* h, float* c, int H, float* scratch) {
  float* gates = scratch; /* [4H] */
#pragma omp parallel for schedule(static)
  for (int r = 0; r < 4 * H; ++r)
    gates[r] = (dotf(w_ih + (size_t)r * n_x, x, n_x) + b_ih[r]) + (dotf(w_hh + (size_t)r * H, h, H) + b_hh[r]);
  for (int u = 0; u < H; ++u) {
    const float ig = sigmoidf_(gates[u]);
    const float fg = sigmoidf_(gates[H + u]);
    const float gg = tanhf(gates[2 * H + u]);
    const float og = sigmoidf_(gates[3 * H + u]);
    const float cn = fg * c[u] + ig * gg;
    c[u] = cn;
    h[u] = og * tanhf(cn);
  }
}

/* reference: get_vector x(1..4) + cat + transpose (src/tacotron/model.py:752-773, 585-609),
 * Encoder.inference (model.py:280-289; batched length semantics of Encoder.forward 262-278),
 * speaker concat (model.py:779-782), memory_layer (model.py:403). */
int taco_oracle_encode(const taco_dims* d, const taco_weights* w, const int64_t* const* ids,
                       const int64_t* speakers, const int32_t* lens, int32_t B, int32_t T_in,
                       float* memory, float* processed_memory) {
  const int E = d->E, H = d->H_e, M = d->M, T = T_in;
  float* x = (float*)malloc(sizeof(float) * (size_t)E * T);
  float* y = (float*)malloc(sizeof(float) * (size_t)E * T);
  float* xt = (float*)malloc(sizeof(float) * (size_t)E);
  float* hs = (float*)malloc(sizeof(float) * (size_t)(2 * H + 4 * H));
  if (!x || !y || !xt || !hs) return TACO_ENOMEM;
  for (int b = 0; b < B; ++b) {
    const int len = lens ? lens[b] : T;
    if (len < 1 || len > T) return TACO_EINVAL;
    /* input vectors, channel-major [E][T] */
    memset(x, 0, sizeof(float) * (size_t)E * T);
    int off = 0;
    for (int c = 0; c < d->n_channels; ++c) {
      const int wd = d->ch_width[c];
      for (int t = 0; t < len; ++t) {
        const int64_t id = ids[c][(size_t)b * T + t];
        if (id < 0 || id >= d->ch_rows[c]) return TACO_EINVAL;
        if (w->ch_table[c]) {
          for (int j = 0; j < wd; ++j) x[(size_t)(off + j) * T + t] = w->ch_table[c][(size_t)id * wd + j];
        } else {
          x[(size_t)(off + id) * T + t] = 1.0f;
        }
      }
      off += wd;
    }
    if (off != E) return TACO_EINVAL;
    for (int l = 0; l < d->n_enc; ++l) {
      conv_bn_act(x, y, E, E, d->k_enc, T, len, w->enc_conv_w[l], w->enc_conv_b[l], w->enc_bn_w[l],
                  w->enc_bn_b[l], w->enc_bn_mean[l], w->enc_bn_var[l], 1);
      float* tmp = x; x = y; y = tmp;
    }
    float* mem = memory + (size_t)b * T * M;
    memset(mem, 0, sizeof(float) * (size_t)T * M);
    for (int dir = 0; dir < 2; ++dir) {
      float* h = hs; float* c = hs + H; float* scratch = hs + 2 * H;
      memset(h, 0, sizeof(float) * 2 * H);
      for (int s = 0; s < len; ++s) {
        const int t = dir == 0 ? s : len - 1 - s;
        for (int e = 0; e < E; ++e) xt[e] = x[(size_t)e * T + t];
        lstm_cell(w->enc_lstm_w_ih[dir], w->enc_lstm_w_hh[dir], w->enc_lstm_b_ih[dir],
                  w->enc_lstm_b_hh[dir], xt, E, h, c, H, scratch);
        memcpy(mem + (size_t)t * M + dir * H, h, sizeof(float) * H);
      }
    }
    if (d->S > 0) {
      for (int t = 0; t < len; ++t) {
        const int64_t id = speakers[(size_t)b * T + t];
        if (id < 0 || id >= d->spk_rows) return TACO_EINVAL;
        float* dst = mem + (size_t)t * M + 2 * H;
        if (w->spk_table) memcpy(dst, w->spk_table + (size_t)id * d->S, sizeof(float) * d->S);
        else dst[id] = 1.0f;
      }
    }
    float* pm = processed_memory + (size_t)b * T * d->A;
    memset(pm, 0, sizeof(float) * (size_t)T * d->A);
    for (int t = 0; t < len; ++t)
      for (int a = 0; a < d->A; ++a)
        pm[(size_t)t * d->A + a] = dotf(w->memory_w + (size_t)a * M, mem + (size_t)t * M, M);
  }
  free(x); free(y); free(xt); free(hs);
  return TACO_OK;
}

/* Tap record per captured step: prenet out [P], attention h [R_a], attention c [R_a],
 * context [M], decoder h [R_d], decoder c [R_d]. */
int32_t taco_oracle_tap_stride(const taco_dims* d) { return d->P + 2 * d->R_a + d->M + 2 * d->R_d; }

/* reference: Decoder.inference (src/tacotron/model.py:529-568): go frame and zero states
 * (358-404), Prenet.forward with always-on dropout (136-141), Decoder.decode (454-491),
 * Attention.forward / get_alignment_energies / LocationLayer (96-117, 73-94, 20-44), stop test
 * (555-562), parse_decoder_outputs (425-452). */
int taco_oracle_decode(const taco_dims* d, const taco_weights* w, const float* memory,
                       const float* processed_memory, const int32_t* lens, int32_t B, int32_t T_in,
                       const uint8_t* masks, int32_t T_cap, int32_t max_steps, float gate_threshold,
                       const float* forced_mel, float* mel, float* gate, float* align,
                       int32_t* n_frames, uint8_t* reached_max, float* taps, int32_t n_tap_steps) {
  const int P = d->P, M = d->M, Ra = d->R_a, Rd = d->R_d, A = d->A, F = d->F, K = d->K, NM = d->n_mel;
  if (max_steps < 1 || max_steps > T_cap) return TACO_EINVAL;
  const int pad = (K - 1) / 2;
  const int n_xa = P + M, n_xd = Ra + M, n_hc = Rd + M;
  float* buf = (float*)calloc((size_t)(NM + 2 * P + n_xa + 2 * Ra + n_xd + 2 * Rd + n_hc + 4 * (Ra > Rd ? Ra : Rd)
                                       + 3 * T_in + (size_t)F * T_in + A + T_in), sizeof(float));
  if (!buf) return TACO_ENOMEM;
  float* p = buf;
  float* dec_in = p; p += NM;
  float* pre1 = p; p += P;
  float* pre2 = p; p += P;
  float* xa = p; p += n_xa;
  float* ah = p; p += Ra;
  float* ac = p; p += Ra;
  float* xd = p; p += n_xd;
  float* dh = p; p += Rd;
  float* dc = p; p += Rd;
  float* hc = p; p += n_hc;
  float* scratch = p; p += 4 * (Ra > Rd ? Ra : Rd);
  float* aw = p; p += T_in;
  float* cum = p; p += T_in;
  float* energy = p; p += T_in;
  float* conv = p; p += (size_t)F * T_in;
  float* q = p; p += A;
  float* ctx_tmp = p; p += T_in;
  (void)ctx_tmp;
  const int tap_stride = taco_oracle_tap_stride(d);

  for (int b = 0; b < B; ++b) {
    const int len = lens ? lens[b] : T_in;
    const float* mem = memory + (size_t)b * T_in * M;
    const float* pm = processed_memory + (size_t)b * T_in * A;
    float* mel_b = mel + (size_t)b * NM * T_cap;
    float* gate_b = gate + (size_t)b * T_cap;
    memset(mel_b, 0, sizeof(float) * (size_t)NM * T_cap);
    memset(gate_b, 0, sizeof(float) * (size_t)T_cap);
    if (align) memset(align + (size_t)b * T_cap * T_in, 0, sizeof(float) * (size_t)T_cap * T_in);
    memset(buf, 0, sizeof(float) * (size_t)(p - buf));
    float* ctx = xa + P; /* previous context lives inside the attention-LSTM input */
    int n = 0, reached = 0;
    for (int t = 0; t < max_steps; ++t) {
      /* decoder input: zeros, own previous frame, or the forced previous frame */
      if (t > 0) {
        const float* src = forced_mel ? forced_mel + (size_t)b * NM * T_cap : mel_b;
        for (int j = 0; j < NM; ++j) dec_in[j] = src[(size_t)j * T_cap + (t - 1)];
      }
      /* prenet: relu(W x) * keep * 2, twice */
      const uint8_t* m1 = masks + ((size_t)t * 2 + 0) * P;
      const uint8_t* m2 = masks + ((size_t)t * 2 + 1) * P;
      for (int r = 0; r < P; ++r) {
        float v = dotf(w->prenet_w1 + (size_t)r * NM, dec_in, NM);
        v = v > 0.0f ? v : 0.0f;
        pre1[r] = m1[r] ? v * 2.0f : 0.0f;
      }
#pragma omp parallel for schedule(static)
      for (int r = 0; r < P; ++r) {
        float v = dotf(w->prenet_w2 + (size_t)r * P, pre1, P);
        v = v > 0.0f ? v : 0.0f;
        pre2[r] = m2[r] ? v * 2.0f : 0.0f;
      }
      /* attention LSTM on [prenet out (+) previous context] */
      memcpy(xa, pre2, sizeof(float) * P);
      lstm_cell(w->att_w_ih, w->att_w_hh, w->att_b_ih, w->att_b_hh, xa, n_xa, ah, ac, Ra, scratch);
      /* location-sensitive attention; conv input = (previous weights, cumulative weights) */
      for (int a = 0; a < A; ++a) q[a] = dotf(w->query_w + (size_t)a * Ra, ah, Ra);
#pragma omp parallel for schedule(static)
      for (int f = 0; f < F; ++f) {
        const float* w0 = w->loc_conv_w + ((size_t)f * 2 + 0) * K;
        const float* w1 = w->loc_conv_w + ((size_t)f * 2 + 1) * K;
        for (int s = 0; s < len; ++s) {
          float acc = 0.0f;
          for (int kk = 0; kk < K; ++kk) {
            const int ss = s + kk - pad;
            if (ss >= 0 && ss < len) acc += w0[kk] * aw[ss];
          }
          for (int kk = 0; kk < K; ++kk) {
            const int ss = s + kk - pad;
            if (ss >= 0 && ss < len) acc += w1[kk] * cum[ss];
          }
          conv[(size_t)f * T_in + s] = acc;
        }
      }
      float emax = -INFINITY;
#pragma omp parallel for schedule(static) reduction(max : emax)
      for (int s = 0; s < len; ++s) {
        float e = 0.0f;
        for (int a = 0; a < A; ++a) {
          float loc = 0.0f;
          for (int f = 0; f < F; ++f) loc += w->loc_dense_w[(size_t)a * F + f] * conv[(size_t)f * T_in + s];
          e += w->v_w[a] * tanhf(q[a] + loc + pm[(size_t)s * A + a]);
        }
        energy[s] = e;
        if (e > emax) emax = e;
      }
      float esum = 0.0f;
      for (int s = 0; s < len; ++s) { energy[s] = expf(energy[s] - emax); esum += energy[s]; }
      for (int s = 0; s < len; ++s) { aw[s] = energy[s] / esum; cum[s] += aw[s]; }
#pragma omp parallel for schedule(static)
      for (int m = 0; m < M; ++m) {
        float acc = 0.0f;
        for (int s = 0; s < len; ++s) acc += aw[s] * mem[(size_t)s * M + m];
        ctx[m] = acc;
      }
      /* decoder LSTM on [attention hidden (+) context] */
      memcpy(xd, ah, sizeof(float) * Ra);
      memcpy(xd + Ra, ctx, sizeof(float) * M);
      lstm_cell(w->dec_w_ih, w->dec_w_hh, w->dec_b_ih, w->dec_b_hh, xd, n_xd, dh, dc, Rd, scratch);
      /* projections on [decoder hidden (+) context] */
      memcpy(hc, dh, sizeof(float) * Rd);
      memcpy(hc + Rd, ctx, sizeof(float) * M);
      for (int j = 0; j < NM; ++j)
        mel_b[(size_t)j * T_cap + t] = dotf(w->proj_w + (size_t)j * n_hc, hc, n_hc) + w->proj_b[j];
      const float g = dotf(w->gate_w, hc, n_hc) + w->gate_b[0];
      gate_b[t] = g;
      if (align) memcpy(align + ((size_t)b * T_cap + t) * T_in, aw, sizeof(float) * len);
      if (taps && t < n_tap_steps) {
        float* tp = taps + ((size_t)b * n_tap_steps + t) * tap_stride;
        memcpy(tp, pre2, sizeof(float) * P); tp += P;
        memcpy(tp, ah, sizeof(float) * Ra); tp += Ra;
        memcpy(tp, ac, sizeof(float) * Ra); tp += Ra;
        memcpy(tp, ctx, sizeof(float) * M); tp += M;
        memcpy(tp, dh, sizeof(float) * Rd); tp += Rd;
        memcpy(tp, dc, sizeof(float) * Rd);
      }
      n = t + 1;
      if (!forced_mel && sigmoidf_(g) > gate_threshold) break;
      if (n == max_steps) { reached = forced_mel ? 0 : 1; break; }
    }
    n_frames[b] = n;
    reached_max[b] = (uint8_t)reached;
  }
  free(buf);
  return TACO_OK;
}

/* reference: Postnet.forward (src/tacotron/model.py:205-211) and the residual add (788-789). */
int taco_oracle_postnet(const taco_dims* d, const taco_weights* w, const float* mel,
                        const int32_t* n_frames, int32_t B, int32_t T_cap, float* out) {
  const int NM = d->n_mel, C = d->C_post;
  const int cmax = C > NM ? C : NM;
  float* x = (float*)malloc(sizeof(float) * (size_t)cmax * T_cap);
  float* y = (float*)malloc(sizeof(float) * (size_t)cmax * T_cap);
  if (!x || !y) return TACO_ENOMEM;
  for (int b = 0; b < B; ++b) {
    const int len = n_frames ? n_frames[b] : T_cap;
    const float* mb = mel + (size_t)b * NM * T_cap;
    float* ob = out + (size_t)b * NM * T_cap;
    memcpy(x, mb, sizeof(float) * (size_t)NM * T_cap);
    float* cur = x; float* nxt = y;
    for (int l = 0; l < d->n_post; ++l) {
      const int c_in = l == 0 ? NM : C;
      const int c_out = l == d->n_post - 1 ? NM : C;
      conv_bn_act(cur, nxt, c_in, c_out, d->k_post, T_cap, len, w->post_conv_w[l], w->post_conv_b[l],
                  w->post_bn_w[l], w->post_bn_b[l], w->post_bn_mean[l], w->post_bn_var[l],
                  l == d->n_post - 1 ? 0 : 2);
      float* tmp = cur; cur = nxt; nxt = tmp;
    }
    for (int j = 0; j < NM; ++j)
      for (int t = 0; t < T_cap; ++t)
        ob[(size_t)j * T_cap + t] = t < len ? mb[(size_t)j * T_cap + t] + cur[(size_t)j * T_cap + t] : 0.0f;
  }
  free(x); free(y);
  return TACO_OK;
}
