// Persistent autoregressive decoder for gfx950 (MI355X): every frame of one utterance in ONE launch.
//
// Replaces the reference's per-frame Python loop Decoder.inference -> Prenet.forward ->
// Decoder.decode -> Attention.forward (reference: src/tacotron/model.py:529-568, 136-141, 454-491,
// 96-117, 73-94, 20-44).
//
// Shape of the computation at batch 1: ~18.4 M weights are touched per frame by GEMVs, so the frame
// is bounded by how fast weights reach the ALUs and by the latency of the dependent steps, not by
// FLOPs.  Design:
//   * one workgroup of 512 threads per CU (256 workgroups), resident for the whole utterance;
//   * both LSTM weight matrices stay in VGPRs for the whole launch: workgroup j owns hidden units
//     4j..4j+3 of each LSTM, wave w holds two gate rows (2*(w&1), +1) of unit (w>>1) - 2*(32+44)
//     registers per lane, laid out so a lane's 4 weights of a chunk meet one ds_read_b128 of the
//     activation vector (each 16-byte read feeds both rows);
//   * the small matrices (prenet, query, location, projection) are spread one row per workgroup and
//     kept in LDS; per-utterance attention memory columns / processed-memory rows likewise;
//   * workgroups exchange activation vectors through 8-byte {epoch, fp32} granules written with
//     agent-scope (write-through) stores and polled with agent-scope loads: the value is its own
//     ready flag, so no grid barrier and no cache fence sits between the dependent steps;
//   * everything that does not depend on the value being waited for (the recurrent halves of the
//     LSTM dot products, location features) is computed before the wait.
// Granule slots are double-buffered by frame parity and every workgroup publishes a per-frame
// heartbeat that all others collect before starting the next frame, so no workgroup can run far
// enough ahead to overwrite a word another one still has to read.  Every wait is bounded by a
// wall-clock deadline and a global abort word, so the grid always drains.
#include "taco_device.h"

namespace {

constexpr int NT = TACO_DEC_THREADS;
constexpr int CH_A = TACO_NP + TACO_NM + TACO_NA;   // 8 chunks per attention-LSTM row
constexpr int CH_D = TACO_NA + TACO_NM + TACO_ND;   // 11 chunks per decoder-LSTM row
constexpr int HALO = 16;                            // >= (K-1)/2 zero columns around the weights

// LDS carve (float offsets, all multiples of 4)
constexpr int L_XP = 0;                                        // prenet out            [256]
constexpr int L_XCTX = L_XP + TACO_NP * TACO_CHUNK;            // attention context     [768]
constexpr int L_XAH = L_XCTX + TACO_NM * TACO_CHUNK;           // attention hidden      [1024]
constexpr int L_XDH = L_XAH + TACO_NA * TACO_CHUNK;            // decoder hidden        [1024]
constexpr int L_MELIN = L_XDH + TACO_ND * TACO_CHUNK;          // decoder input + gate  [128]
constexpr int L_P1 = L_MELIN + 128;                            // prenet layer-1 out    [256]
constexpr int L_Q = L_P1 + TACO_NP * TACO_CHUNK;               // processed query       [128]
constexpr int L_EW = L_Q + TACO_MAX_A;                         // energies              [1024]
constexpr int L_AW = L_EW + TACO_MAX_TIN;                      // attention weights, halo both sides
constexpr int L_CUM = L_AW + TACO_MAX_TIN + 2 * HALO;          // cumulative weights, halo both sides
constexpr int L_U = L_CUM + TACO_MAX_TIN + 2 * HALO;           // location term + processed memory [slots][128]
constexpr int L_PMROW = L_U + TACO_E_SLOTS * TACO_MAX_A;       // processed-memory rows [slots][128]
constexpr int L_MEMCOL = L_PMROW + TACO_E_SLOTS * TACO_MAX_A;  // memory columns [slots][1024]
constexpr int L_W1ROW = L_MEMCOL + TACO_C_SLOTS * TACO_MAX_TIN;  // prenet layer-1 row [128]
constexpr int L_W2ROW = L_W1ROW + 128;                         // prenet layer-2 row [256]
constexpr int L_QROW = L_W2ROW + TACO_NP * TACO_CHUNK;         // query row [1024]
constexpr int L_PROJROW = L_QROW + TACO_NA * TACO_CHUNK;       // projection row [1024 + 768]
constexpr int L_LOCCONV = L_PROJROW + (TACO_ND + TACO_NM) * TACO_CHUNK;   // [F][2][K] <= 1984
constexpr int L_LOCDT = L_LOCCONV + 1984;                      // location dense, transposed [F][128]
constexpr int L_V = L_LOCDT + TACO_MAX_F * TACO_MAX_A;         // v [128]
constexpr int L_CONVOUT = L_V + TACO_MAX_A;                    // location conv out [slots][32]
constexpr int L_GATES = L_CONVOUT + TACO_E_SLOTS * TACO_MAX_F; // LSTM gate pre-activations [16]
constexpr int L_RED = L_GATES + 16;                            // block-reduction scratch [32]
constexpr int L_CELLS = L_RED + 32;                            // LSTM cell states [4 + 4]
constexpr int L_TOTAL = L_CELLS + 8;

struct DecArgs {
  int P, M, Ra, Rd, A, F, K, n_mel, T_in, n_wg, proj_off;
  const int32_t* lens;   // device, or NULL = T_in symbols
  int b;                 // utterance index into lens
  const float *prenet_w1, *prenet_w2, *att_w_ih, *att_w_hh, *att_b_ih, *att_b_hh, *query_w, *v_w, *loc_conv_w,
      *loc_dense_w, *dec_w_ih, *dec_w_hh, *dec_b_ih, *dec_b_hh, *proj_w, *proj_b, *gate_w, *gate_b;
  const float *memory, *pm;
  const uint8_t* masks;
  int T_cap, max_steps;
  float thr;
  const float* forced_mel;
  float *mel, *gate, *align;
  taco_granule* gran;
  taco_dec_layout lay;
  taco_dec_ctl* ctl;
  unsigned long long timeout_ticks;
};

// Dot product of NC register-resident chunks with the activation vector in LDS.  Lane l holds
// columns 256*c + 4*l .. +3 of its row, which is exactly one 16-byte LDS read per chunk.
template <int NC>
__device__ __forceinline__ void dot_chunks(const float (*w0)[4], const float (*w1)[4], const float* x_lds, int lane,
                                           float (&acc)[2][4]) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float4 x = *reinterpret_cast<const float4*>(x_lds + c * TACO_CHUNK + lane * 4);
    acc[0][0] = fmaf(w0[c][0], x.x, acc[0][0]);
    acc[0][1] = fmaf(w0[c][1], x.y, acc[0][1]);
    acc[0][2] = fmaf(w0[c][2], x.z, acc[0][2]);
    acc[0][3] = fmaf(w0[c][3], x.w, acc[0][3]);
    acc[1][0] = fmaf(w1[c][0], x.x, acc[1][0]);
    acc[1][1] = fmaf(w1[c][1], x.y, acc[1][1]);
    acc[1][2] = fmaf(w1[c][2], x.z, acc[1][2]);
    acc[1][3] = fmaf(w1[c][3], x.w, acc[1][3]);
  }
}

// Strided dot of an LDS row with an LDS vector by one wave (lane-strided, then wave sum).
__device__ __forceinline__ float wave_dot(const float* a, const float* b, int n, int lane) {
  float acc = 0.0f;
  for (int k = lane; k < n; k += 64) acc = fmaf(a[k], b[k], acc);
  return taco_wave_sum(acc);
}

__global__ __launch_bounds__(TACO_DEC_THREADS) void taco_decoder_kernel(const DecArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = blockIdx.x;
  const int n_wg = a.n_wg;
  const int len = a.lens ? min(a.lens[a.b], a.T_in) : a.T_in;

  for (int i = tid; i < L_TOTAL; i += NT) lds[i] = 0.0f;
  __syncthreads();

  // ---------------- one-time loads: this workgroup's slices of the model and of the utterance ----
  const int Ua = (a.Ra + n_wg - 1) / n_wg;          // hidden units per workgroup (<= 4)
  const int Ud = (a.Rd + n_wg - 1) / n_wg;
  const int unit = wave >> 1, gate0 = (wave & 1) * 2;   // this wave's rows: gates gate0, gate0 + 1 of `unit`
  const int gu_a = j * Ua + unit, gu_d = j * Ud + unit;
  const bool act_a = unit < Ua && gu_a < a.Ra;       // wave-uniform
  const bool act_d = unit < Ud && gu_d < a.Rd;

  float wa[2][CH_A][4];
  float wd[2][CH_D][4];
  float bias_a[2] = {0.0f, 0.0f}, bias_d[2] = {0.0f, 0.0f};
  {
    const int n_ih_a = a.P + a.M, n_ih_d = a.Ra + a.M;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int ra = (gate0 + r) * a.Ra + gu_a;
      const int rd = (gate0 + r) * a.Rd + gu_d;
#pragma unroll
      for (int c = 0; c < CH_A; ++c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = 0.0f;
          if (act_a) {
            if (c < TACO_NP) {
              const int col = c * TACO_CHUNK + lane * 4 + e;
              if (col < a.P) v = a.att_w_ih[(size_t)ra * n_ih_a + col];
            } else if (c < TACO_NP + TACO_NM) {
              const int col = (c - TACO_NP) * TACO_CHUNK + lane * 4 + e;
              if (col < a.M) v = a.att_w_ih[(size_t)ra * n_ih_a + a.P + col];
            } else {
              const int col = (c - TACO_NP - TACO_NM) * TACO_CHUNK + lane * 4 + e;
              if (col < a.Ra) v = a.att_w_hh[(size_t)ra * a.Ra + col];
            }
          }
          wa[r][c][e] = v;
        }
      }
#pragma unroll
      for (int c = 0; c < CH_D; ++c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = 0.0f;
          if (act_d) {
            if (c < TACO_NA) {
              const int col = c * TACO_CHUNK + lane * 4 + e;
              if (col < a.Ra) v = a.dec_w_ih[(size_t)rd * n_ih_d + col];
            } else if (c < TACO_NA + TACO_NM) {
              const int col = (c - TACO_NA) * TACO_CHUNK + lane * 4 + e;
              if (col < a.M) v = a.dec_w_ih[(size_t)rd * n_ih_d + a.Ra + col];
            } else {
              const int col = (c - TACO_NA - TACO_NM) * TACO_CHUNK + lane * 4 + e;
              if (col < a.Rd) v = a.dec_w_hh[(size_t)rd * a.Rd + col];
            }
          }
          wd[r][c][e] = v;
        }
      }
      if (act_a) bias_a[r] = a.att_b_ih[ra] + a.att_b_hh[ra];
      if (act_d) bias_d[r] = a.dec_b_ih[rd] + a.dec_b_hh[rd];
    }
  }

  const bool own_prenet = j < a.P;
  const bool own_q = j < a.A;
  const int proj_row = j - a.proj_off;
  const bool own_proj = proj_row >= 0 && proj_row <= a.n_mel;   // row n_mel is the stop-token row
  const int n_hc = a.Rd + a.M;
  float proj_bias = 0.0f;
  if (own_prenet) {
    for (int k = tid; k < a.n_mel; k += NT) lds[L_W1ROW + k] = a.prenet_w1[(size_t)j * a.n_mel + k];
    for (int k = tid; k < a.P; k += NT) lds[L_W2ROW + k] = a.prenet_w2[(size_t)j * a.P + k];
  }
  if (own_q)
    for (int k = tid; k < a.Ra; k += NT) lds[L_QROW + k] = a.query_w[(size_t)j * a.Ra + k];
  if (own_proj) {
    const float* src = proj_row < a.n_mel ? a.proj_w + (size_t)proj_row * n_hc : a.gate_w;
    for (int k = tid; k < n_hc; k += NT) lds[L_PROJROW + k] = src[k];
    proj_bias = proj_row < a.n_mel ? a.proj_b[proj_row] : a.gate_b[0];
  }
  for (int k = tid; k < a.F * 2 * a.K; k += NT) lds[L_LOCCONV + k] = a.loc_conv_w[k];
  for (int k = tid; k < a.A * a.F; k += NT) {
    const int aa = k / a.F, f = k % a.F;
    lds[L_LOCDT + f * TACO_MAX_A + aa] = a.loc_dense_w[k];
  }
  for (int k = tid; k < a.A; k += NT) lds[L_V + k] = a.v_w[k];
  // per-utterance: processed-memory rows of my positions, memory columns of my context outputs
  for (int slot = 0; slot < TACO_E_SLOTS; ++slot) {
    const int s = j + slot * n_wg;
    if (s < len)
      for (int k = tid; k < a.A; k += NT) lds[L_PMROW + slot * TACO_MAX_A + k] = a.pm[(size_t)s * a.A + k];
  }
  for (int slot = 0; slot < TACO_C_SLOTS; ++slot) {
    const int m = j + slot * n_wg;
    if (m < a.M)
      for (int s = tid; s < len; s += NT) lds[L_MEMCOL + slot * TACO_MAX_TIN + s] = a.memory[(size_t)s * a.M + m];
  }
  __syncthreads();

  taco_wait_ctx wctx;
  wctx.abort_word = (taco_gu32*)&a.ctl->abort_code;
  wctx.deadline = wall_clock64() + a.timeout_ticks;
  wctx.failed = 0;

  const bool forced = a.forced_mel != nullptr;
  const int pad = (a.K - 1) / 2;
  int stopped = 0;
  int t = 0;
  for (;; ++t) {
    // ============ frame boundary: previous frame's output + every workgroup's heartbeat ============
    if (t > 0) {
      const taco_granule* gp = a.gran + (size_t)((t - 1) & 1) * a.lay.per_parity;
      const unsigned ep_prev = (unsigned)t;
      if (tid <= a.n_mel) {
        const float v = taco_wait_one(gp + a.lay.mel, tid, ep_prev, wctx, 0x100u);
        if (!forced || tid == a.n_mel) lds[L_MELIN + tid] = v;
      } else if (tid >= 128 && tid < 128 + n_wg) {
        (void)taco_wait_one(gp + a.lay.beat, tid - 128, ep_prev, wctx, 0x200u);
      }
      if (forced && tid < a.n_mel) lds[L_MELIN + tid] = a.forced_mel[(size_t)tid * a.T_cap + (t - 1)];
      if (__syncthreads_or(wctx.failed)) break;
      if (!forced && taco_sigmoid(lds[L_MELIN + a.n_mel]) > a.thr) { stopped = 1; break; }
    }
    if (t == a.max_steps) break;
    wctx.deadline = wall_clock64() + a.timeout_ticks;

    const unsigned ep = (unsigned)t + 1u;
    taco_granule* g = a.gran + (size_t)(t & 1) * a.lay.per_parity;
    unsigned keep1 = 0, keep2 = 0;
    if (own_prenet) {
      keep1 = a.masks[((size_t)t * 2 + 0) * a.P + j];
      keep2 = a.masks[((size_t)t * 2 + 1) * a.P + j];
    }

    // ============ S1: prenet layer 1, row j ============
    if (own_prenet && wave == 0) {
      float v = wave_dot(lds + L_W1ROW, lds + L_MELIN, a.n_mel, lane);
      v = v > 0.0f ? v : 0.0f;
      v = keep1 ? v * 2.0f : 0.0f;
      if (lane == 0) taco_publish(g + a.lay.p1, j, ep, v);
    }
    // meanwhile: attention-LSTM partial sums over [previous context | previous hidden]
    float acc_a[2][4] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
    if (act_a) {
      dot_chunks<TACO_NM>(wa[0] + TACO_NP, wa[1] + TACO_NP, lds + L_XCTX, lane, acc_a);
      dot_chunks<TACO_NA>(wa[0] + TACO_NP + TACO_NM, wa[1] + TACO_NP + TACO_NM, lds + L_XAH, lane, acc_a);
    }

    // ============ S2: prenet layer 2, row j ============
    taco_gather<NT>(g + a.lay.p1, a.P, ep, lds + L_P1, wctx, 0x300u);
    if (own_prenet && wave == 0) {
      float v = wave_dot(lds + L_W2ROW, lds + L_P1, a.P, lane);
      v = v > 0.0f ? v : 0.0f;
      v = keep2 ? v * 2.0f : 0.0f;
      if (lane == 0) taco_publish(g + a.lay.p, j, ep, v);
    }

    // ============ S3: attention LSTM ============
    taco_gather<NT>(g + a.lay.p, a.P, ep, lds + L_XP, wctx, 0x400u);
    if (act_a) {
      dot_chunks<TACO_NP>(wa[0], wa[1], lds + L_XP, lane, acc_a);
      const float s0 = taco_wave_sum((acc_a[0][0] + acc_a[0][1]) + (acc_a[0][2] + acc_a[0][3]));
      const float s1 = taco_wave_sum((acc_a[1][0] + acc_a[1][1]) + (acc_a[1][2] + acc_a[1][3]));
      if (lane == 0) {
        lds[L_GATES + wave * 2 + 0] = s0 + bias_a[0];
        lds[L_GATES + wave * 2 + 1] = s1 + bias_a[1];
      }
    }
    __syncthreads();
    if (tid < Ua && j * Ua + tid < a.Ra) {
      const float ig = taco_sigmoid(lds[L_GATES + tid * 4 + 0]);
      const float fg = taco_sigmoid(lds[L_GATES + tid * 4 + 1]);
      const float gg = tanhf(lds[L_GATES + tid * 4 + 2]);
      const float og = taco_sigmoid(lds[L_GATES + tid * 4 + 3]);
      const float c = fg * lds[L_CELLS + tid] + ig * gg;
      lds[L_CELLS + tid] = c;
      taco_publish(g + a.lay.ah, j * Ua + tid, ep, og * tanhf(c));
    }
    // meanwhile: decoder-LSTM partial over the previous decoder hidden state ...
    float acc_d[2][4] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
    if (act_d) dot_chunks<TACO_ND>(wd[0] + TACO_NA + TACO_NM, wd[1] + TACO_NA + TACO_NM, lds + L_XDH, lane, acc_d);
    // ... and the location features of my positions (they use the previous frame's weights)
    if (tid < 256) {
      const int f = tid >> 3, part = tid & 7;
      for (int slot = 0; slot < TACO_E_SLOTS; ++slot) {
        const int s = j + slot * n_wg;
        if (s >= len) break;
        float acc = 0.0f;
        if (f < a.F) {
          for (int kk = part; kk < 2 * a.K; kk += 8) {
            const int c = kk >= a.K ? 1 : 0;
            const int k = kk - c * a.K;
            const float x = lds[(c ? L_CUM : L_AW) + HALO + s + k - pad];
            acc = fmaf(lds[L_LOCCONV + (f * 2 + c) * a.K + k], x, acc);
          }
        }
        acc += __shfl_xor(acc, 4, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 1, 64);
        if (part == 0 && f < a.F) lds[L_CONVOUT + slot * TACO_MAX_F + f] = acc;
      }
    }
    __syncthreads();
    if (tid < TACO_E_SLOTS * TACO_MAX_A) {
      const int slot = tid >> 7, aa = tid & 127;
      const int s = j + slot * n_wg;
      if (s < len && aa < a.A) {
        float loc = 0.0f;
        for (int f = 0; f < a.F; ++f)
          loc = fmaf(lds[L_LOCDT + f * TACO_MAX_A + aa], lds[L_CONVOUT + slot * TACO_MAX_F + f], loc);
        lds[L_U + slot * TACO_MAX_A + aa] = loc + lds[L_PMROW + slot * TACO_MAX_A + aa];
      }
    }

    // ============ S4: query projection, row j ============
    taco_gather<NT>(g + a.lay.ah, a.Ra, ep, lds + L_XAH, wctx, 0x500u);
    if (own_q && wave == 0) {
      const float v = wave_dot(lds + L_QROW, lds + L_XAH, a.Ra, lane);
      if (lane == 0) taco_publish(g + a.lay.q, j, ep, v);
    }
    if (act_d) dot_chunks<TACO_NA>(wd[0], wd[1], lds + L_XAH, lane, acc_d);

    // ============ S5: energies of my positions ============
    taco_gather<NT>(g + a.lay.q, a.A, ep, lds + L_Q, wctx, 0x600u);
    if (wave < TACO_E_SLOTS) {
      const int s = j + wave * n_wg;
      if (s < len) {
        float acc = 0.0f;
        for (int aa = lane; aa < a.A; aa += 64)
          acc = fmaf(lds[L_V + aa], tanhf(lds[L_Q + aa] + lds[L_U + wave * TACO_MAX_A + aa]), acc);
        acc = taco_wave_sum(acc);
        if (lane == 0) taco_publish(g + a.lay.e, s, ep, acc);
      }
    }

    // ============ S6: softmax over all positions (every workgroup), context columns of mine ============
    taco_gather<NT>(g + a.lay.e, len, ep, lds + L_EW, wctx, 0x700u);
    {
      constexpr int PER = TACO_MAX_TIN / NT;   // positions per thread
      float ev[PER], ex[PER];
      float wm = -INFINITY;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int s = tid + i * NT;
        ev[i] = s < len ? lds[L_EW + s] : -INFINITY;
        wm = fmaxf(wm, ev[i]);
      }
      wm = taco_wave_max(wm);
      if (lane == 0) lds[L_RED + wave] = wm;
      __syncthreads();
      float mx = lds[L_RED];
#pragma unroll
      for (int i = 1; i < NT / 64; ++i) mx = fmaxf(mx, lds[L_RED + i]);
      float ws = 0.0f;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int s = tid + i * NT;
        ex[i] = s < len ? expf(ev[i] - mx) : 0.0f;
        ws += ex[i];
      }
      ws = taco_wave_sum(ws);
      if (lane == 0) lds[L_RED + 16 + wave] = ws;
      __syncthreads();
      float sum = 0.0f;
#pragma unroll
      for (int i = 0; i < NT / 64; ++i) sum += lds[L_RED + 16 + i];
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int s = tid + i * NT;
        if (s < len) {
          const float wgt = ex[i] / sum;
          lds[L_AW + HALO + s] = wgt;
          lds[L_CUM + HALO + s] += wgt;
          if (j == 0 && a.align) a.align[(size_t)t * a.T_in + s] = wgt;
        }
      }
      __syncthreads();
    }
    if (wave < TACO_C_SLOTS) {
      const int m = j + wave * n_wg;
      if (m < a.M) {
        float acc = 0.0f;
        for (int s = lane; s < len; s += 64)
          acc = fmaf(lds[L_AW + HALO + s], lds[L_MEMCOL + wave * TACO_MAX_TIN + s], acc);
        acc = taco_wave_sum(acc);
        if (lane == 0) taco_publish(g + a.lay.ctx, m, ep, acc);
      }
    }

    // ============ S7: decoder LSTM ============
    taco_gather<NT>(g + a.lay.ctx, a.M, ep, lds + L_XCTX, wctx, 0x800u);
    if (tid == 0) taco_publish(g + a.lay.beat, j, ep, 0.0f);   // heartbeat: frame t's inputs consumed
    if (act_d) {
      dot_chunks<TACO_NM>(wd[0] + TACO_NA, wd[1] + TACO_NA, lds + L_XCTX, lane, acc_d);
      const float s0 = taco_wave_sum((acc_d[0][0] + acc_d[0][1]) + (acc_d[0][2] + acc_d[0][3]));
      const float s1 = taco_wave_sum((acc_d[1][0] + acc_d[1][1]) + (acc_d[1][2] + acc_d[1][3]));
      if (lane == 0) {
        lds[L_GATES + wave * 2 + 0] = s0 + bias_d[0];
        lds[L_GATES + wave * 2 + 1] = s1 + bias_d[1];
      }
    }
    __syncthreads();
    if (tid < Ud && j * Ud + tid < a.Rd) {
      const float ig = taco_sigmoid(lds[L_GATES + tid * 4 + 0]);
      const float fg = taco_sigmoid(lds[L_GATES + tid * 4 + 1]);
      const float gg = tanhf(lds[L_GATES + tid * 4 + 2]);
      const float og = taco_sigmoid(lds[L_GATES + tid * 4 + 3]);
      const float c = fg * lds[L_CELLS + 4 + tid] + ig * gg;
      lds[L_CELLS + 4 + tid] = c;
      taco_publish(g + a.lay.dh, j * Ud + tid, ep, og * tanhf(c));
    }
    // meanwhile: context half of my projection row
    float acc_p = 0.0f;
    if (own_proj && wave == 1)
      for (int k = lane; k < a.M; k += 64) acc_p = fmaf(lds[L_PROJROW + a.Rd + k], lds[L_XCTX + k], acc_p);

    // ============ S8: mel / stop-token projection, row proj_row ============
    taco_gather<NT>(g + a.lay.dh, a.Rd, ep, lds + L_XDH, wctx, 0x900u);
    if (own_proj && wave == 1) {
      for (int k = lane; k < a.Rd; k += 64) acc_p = fmaf(lds[L_PROJROW + k], lds[L_XDH + k], acc_p);
      const float v = taco_wave_sum(acc_p) + proj_bias;
      if (lane == 0) {
        taco_publish(g + a.lay.mel, proj_row, ep, v);
        if (proj_row < a.n_mel) a.mel[(size_t)proj_row * a.T_cap + t] = v;
        else a.gate[t] = v;
      }
    }
  }

  if (j == 0 && tid == 0) {
    a.ctl->n_frames = (uint32_t)t;
    a.ctl->reached_max = (!stopped && !forced && t == a.max_steps) ? 1u : 0u;
  }
}

}  // namespace

size_t taco_decoder_lds_bytes() { return (size_t)L_TOTAL * sizeof(float); }

static void fill_layout(taco_dec_layout& lay, int n_wg) {
  int off = 0;
  auto take = [&](int n) {
    const int at = off;
    off += (n + 15) / 16 * 16;   // 128-byte aligned vectors
    return at;
  };
  lay.p1 = take(TACO_NP * TACO_CHUNK);
  lay.p = take(TACO_NP * TACO_CHUNK);
  lay.ah = take(TACO_NA * TACO_CHUNK);
  lay.q = take(TACO_MAX_A);
  lay.e = take(TACO_MAX_TIN);
  lay.ctx = take(TACO_NM * TACO_CHUNK);
  lay.dh = take(TACO_ND * TACO_CHUNK);
  lay.mel = take(128);
  lay.beat = take(n_wg);
  lay.per_parity = off;
}

size_t taco_decoder_granules(int n_wg) {
  taco_dec_layout lay;
  fill_layout(lay, n_wg);
  return (size_t)lay.per_parity * 2;
}

int taco_decoder_check_dims(const taco_dims& d, int n_wg) {
  if (d.P > TACO_NP * TACO_CHUNK || d.P > n_wg) TACO_FAIL(TACO_EINVAL, "prenet_dim %d exceeds the decoder kernel's capacity %d", d.P, TACO_NP * TACO_CHUNK < n_wg ? TACO_NP * TACO_CHUNK : n_wg);
  if (d.M > TACO_NM * TACO_CHUNK || d.M > TACO_C_SLOTS * n_wg) TACO_FAIL(TACO_EINVAL, "memory width %d exceeds the decoder kernel's capacity %d", d.M, TACO_NM * TACO_CHUNK);
  if (d.R_a > TACO_NA * TACO_CHUNK || d.R_a > TACO_UNITS_PER_WG * n_wg) TACO_FAIL(TACO_EINVAL, "attention_rnn_dim %d exceeds the decoder kernel's capacity", d.R_a);
  if (d.R_d > TACO_ND * TACO_CHUNK || d.R_d > TACO_UNITS_PER_WG * n_wg) TACO_FAIL(TACO_EINVAL, "decoder_rnn_dim %d exceeds the decoder kernel's capacity", d.R_d);
  if (d.A > TACO_MAX_A || d.A > n_wg) TACO_FAIL(TACO_EINVAL, "attention_dim %d exceeds the decoder kernel's capacity %d", d.A, TACO_MAX_A);
  if (d.F > TACO_MAX_F) TACO_FAIL(TACO_EINVAL, "attention_location_n_filters %d exceeds %d", d.F, TACO_MAX_F);
  if (d.K > TACO_MAX_K || (d.K & 1) == 0) TACO_FAIL(TACO_EINVAL, "attention_location_kernel_size %d must be odd and <= %d", d.K, TACO_MAX_K);
  if (d.n_mel > TACO_MAX_MEL || d.n_mel + 1 > n_wg) TACO_FAIL(TACO_EINVAL, "n_mel_channels %d exceeds %d", d.n_mel, TACO_MAX_MEL);
  return TACO_OK;
}

int taco_launch_decoder(taco_handle h, const float* memory, const float* pm, int T_in, const int32_t* lens,
                        int b, const uint8_t* masks, int T_cap, int max_steps, float thr, const float* forced_mel,
                        float* mel, float* gate, float* align, int32_t* n_frames, uint8_t* reached,
                        hipStream_t s) {
  const taco_dims& d = h->d;
  const int n_wg = h->n_cu;
  if (T_in < 1 || T_in > TACO_MAX_TIN || T_in > TACO_E_SLOTS * n_wg) TACO_FAIL(TACO_EINVAL, "T_in = %d symbols is outside the decoder kernel's capacity 1..%d", T_in, TACO_MAX_TIN);
  if (max_steps < 1 || max_steps > T_cap) TACO_FAIL(TACO_EINVAL, "max_steps %d outside 1..T_cap=%d", max_steps, T_cap);

  DecArgs a;
  a.P = d.P; a.M = d.M; a.Ra = d.R_a; a.Rd = d.R_d; a.A = d.A; a.F = d.F; a.K = d.K; a.n_mel = d.n_mel;
  a.T_in = T_in; a.lens = lens; a.b = b; a.n_wg = n_wg;
  a.proj_off = (d.A + d.n_mel + 1 <= n_wg) ? d.A : 0;
  const taco_weights& w = h->w;
  a.prenet_w1 = w.prenet_w1; a.prenet_w2 = w.prenet_w2;
  a.att_w_ih = w.att_w_ih; a.att_w_hh = w.att_w_hh; a.att_b_ih = w.att_b_ih; a.att_b_hh = w.att_b_hh;
  a.query_w = w.query_w; a.v_w = w.v_w; a.loc_conv_w = w.loc_conv_w; a.loc_dense_w = w.loc_dense_w;
  a.dec_w_ih = w.dec_w_ih; a.dec_w_hh = w.dec_w_hh; a.dec_b_ih = w.dec_b_ih; a.dec_b_hh = w.dec_b_hh;
  a.proj_w = w.proj_w; a.proj_b = w.proj_b; a.gate_w = w.gate_w; a.gate_b = w.gate_b;
  a.memory = memory; a.pm = pm; a.masks = masks; a.T_cap = T_cap; a.max_steps = max_steps; a.thr = thr;
  a.forced_mel = forced_mel; a.mel = mel; a.gate = gate; a.align = align;
  a.gran = h->dec_gran; a.lay = h->dec_lay; a.ctl = h->dec_ctl;
  a.timeout_ticks = 20000000ull;   // 0.2 s of the 100 MHz wall clock per frame

  const size_t gran_bytes = (size_t)h->dec_lay.per_parity * 2 * sizeof(taco_granule);
  TACO_HIP(hipMemsetAsync(h->dec_gran, 0, gran_bytes, s));
  TACO_HIP(hipMemsetAsync(h->dec_ctl, 0, sizeof(taco_dec_ctl), s));
  TACO_HIP(hipMemsetAsync(mel, 0, sizeof(float) * (size_t)d.n_mel * T_cap, s));
  TACO_HIP(hipMemsetAsync(gate, 0, sizeof(float) * (size_t)T_cap, s));
  if (align) TACO_HIP(hipMemsetAsync(align, 0, sizeof(float) * (size_t)T_cap * T_in, s));

  TACO_HIP(hipEventRecord(h->ev_start, s));
  hipLaunchKernelGGL(taco_decoder_kernel, dim3(n_wg), dim3(NT), taco_decoder_lds_bytes(), s, a);
  TACO_HIP(hipGetLastError());
  TACO_HIP(hipEventRecord(h->ev_stop, s));
  h->timing_valid = 1;

  // results: frame count and flags come back through the control block
  TACO_HIP(hipMemcpyAsync(h->dec_ctl_host, h->dec_ctl, sizeof(taco_dec_ctl), hipMemcpyDeviceToHost, s));
  TACO_HIP(hipStreamSynchronize(s));
  if (h->dec_ctl_host->abort_code != 0)
    TACO_FAIL(TACO_ETIMEDOUT, "decoder hand-off wait gave up (code %u, info 0x%x) after %u frames",
              h->dec_ctl_host->abort_code, h->dec_ctl_host->abort_info, h->dec_ctl_host->n_frames);
  const int32_t nf = (int32_t)h->dec_ctl_host->n_frames;
  const uint8_t rm = (uint8_t)h->dec_ctl_host->reached_max;
  h->last_frames = nf;
  TACO_HIP(hipMemcpyAsync(n_frames, &nf, sizeof(nf), hipMemcpyHostToDevice, s));
  TACO_HIP(hipMemcpyAsync(reached, &rm, sizeof(rm), hipMemcpyHostToDevice, s));
  TACO_HIP(hipStreamSynchronize(s));
  return TACO_OK;
}

void taco_decoder_fill_layout(taco_dec_layout* lay, int n_wg) { fill_layout(*lay, n_wg); }

int taco_decoder_prepare(taco_handle h) {
  // the kernel needs more than the default 64 KiB of dynamic LDS
  TACO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(taco_decoder_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)taco_decoder_lds_bytes()));
  int per_cu = 0;
  TACO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, taco_decoder_kernel, NT, taco_decoder_lds_bytes()));
  if (per_cu < 1) TACO_FAIL(TACO_EINVAL, "decoder kernel does not fit on a CU (occupancy query returned %d)", per_cu);
  (void)h;
  return TACO_OK;
}
