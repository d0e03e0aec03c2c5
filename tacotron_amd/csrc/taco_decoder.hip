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
//   * the attention context never travels: ctx = sum_s w_s * memory_s enters the frame only through
//     W * ctx (both LSTMs' input rows and the projection rows), and W * ctx = (W * memory^T) * w.
//     G = W_ctx * memory^T is computed once per utterance (a GEMM before this launch); the kernel keeps
//     its G rows where the context columns of the weight rows would sit (registers / LDS) and
//     multiplies them with the attention weights every workgroup already has after its softmax;
//   * everything that does not depend on the value being waited for (the recurrent halves of the
//     LSTM dot products, location features) is computed before the wait.
// Granule slots are double-buffered by frame parity and every workgroup publishes a per-frame
// heartbeat that all others collect before starting the next frame, so no workgroup can run far
// enough ahead to overwrite a word another one still has to read.  Every wait is bounded by a
// wall-clock deadline and a global abort word, so the grid always drains.
#include "taco_device.h"
#ifdef TACO_STAMPS
#include <stdio.h>
#include <vector>
#endif

namespace {

constexpr int NT = TACO_DEC_THREADS;
// The context part of an LSTM row holds the row's folded context weights G[row][s], one column per
// input position: NMF chunks cover TACO_MAX_TIN positions.
constexpr int NMF = 2;
static_assert(NMF * TACO_CHUNK == TACO_MAX_TIN, "folded context chunks must cover the position capacity");
constexpr int CH_A = TACO_NP + NMF + TACO_NA;       // 7 chunks per attention-LSTM row
constexpr int CH_D = TACO_NA + NMF + TACO_ND;       // 10 chunks per decoder-LSTM row
constexpr int HALO = 16;                            // >= (K-1)/2 zero columns around the weights

// LDS carve (float offsets, all multiples of 4)
constexpr int L_XP = 0;                                        // prenet out            [256]
constexpr int L_XAH = L_XP + TACO_NP * TACO_CHUNK;             // attention hidden      [1024]
constexpr int L_XDH = L_XAH + TACO_NA * TACO_CHUNK;            // decoder hidden        [1024]
constexpr int L_MELIN = L_XDH + TACO_ND * TACO_CHUNK;          // decoder input + gate  [128]
constexpr int L_P1 = L_MELIN + 128;                            // prenet layer-1 out    [256]
constexpr int L_Q = L_P1 + TACO_NP * TACO_CHUNK;               // processed query       [128]
constexpr int L_EW = L_Q + TACO_MAX_A;                         // energies              [1024]
constexpr int L_AW = L_EW + TACO_MAX_TIN;                      // attention weights, halo both sides
constexpr int L_CUM = L_AW + TACO_MAX_TIN + 2 * HALO;          // cumulative weights, halo both sides
constexpr int L_U = L_CUM + TACO_MAX_TIN + 2 * HALO;           // location term + processed memory [slots][128]
constexpr int L_PMROW = L_U + TACO_E_SLOTS * TACO_MAX_A;       // processed-memory rows [slots][128]
constexpr int TACO_W1_ROWS = TACO_MAX_MEL;                     // n_mel capacity of the replicated prenet layer 1

constexpr int L_W1T = L_PMROW + TACO_E_SLOTS * TACO_MAX_A;     // prenet layer 1, transposed [n_mel][256], in EVERY workgroup
constexpr int P1_KPARTS = TACO_DEC_THREADS / 64;               // layer 1 is split 8 ways over its input columns
constexpr int P1_KPER = TACO_W1_ROWS / P1_KPARTS;              // 10 input columns per part
static_assert(P1_KPER * P1_KPARTS == TACO_W1_ROWS && TACO_NP * TACO_CHUNK == 256, "layer-1 tiling");
constexpr int L_P1HALF = L_W1T + TACO_W1_ROWS * TACO_NP * TACO_CHUNK;   // layer-1 partial sums [8][256]
constexpr int L_W2ROW = L_P1HALF + P1_KPARTS * TACO_NP * TACO_CHUNK;   // prenet layer-2 row [256]
constexpr int L_QROW = L_W2ROW + TACO_NP * TACO_CHUNK;         // query row [1024]
constexpr int L_PROJROW = L_QROW + TACO_NA * TACO_CHUNK;       // projection row [1024 hidden + 512 folded context]
constexpr int L_LOCCONV = L_PROJROW + (TACO_ND + NMF) * TACO_CHUNK;       // [F][2][K] <= 1984
constexpr int L_LOCDT = L_LOCCONV + 1984;                      // location dense, transposed [F][128]
constexpr int L_V = L_LOCDT + TACO_MAX_F * TACO_MAX_A;         // v [128]
constexpr int L_CONVOUT = L_V + TACO_MAX_A;                    // location conv out [slots][32]
constexpr int L_GATES = L_CONVOUT + TACO_E_SLOTS * TACO_MAX_F; // LSTM gate pre-activations [16]
constexpr int L_RED = L_GATES + 16;                            // block-reduction scratch [32]
constexpr int L_CELLS = L_RED + 32;                            // LSTM cell states [4 + 4]
constexpr int L_BIAS = L_CELLS + 8;                            // b_ih + b_hh of my rows [16 + 16]
constexpr int L_PROJB = L_BIAS + 32;                           // bias of my projection row [4]
constexpr int L_KEEP = L_PROJB + 4;                            // prenet keep bytes [2 frame parities][2 layers][256 B]
constexpr int L_TOTAL = L_KEEP + 2 * 2 * 64;
static_assert(L_TOTAL * 4 + 512 <= 160 * 1024, "decoder LDS carve exceeds a CU's 160 KiB");

struct DecArgs {
  int P, M, Ra, Rd, A, F, K, n_mel, T_in, n_wg, proj_off;
  const int32_t* lens;   // device, or NULL = T_in symbols
  int b;                 // utterance index into lens
  const float *prenet_w1, *prenet_w2, *att_w_ih, *att_w_hh, *att_b_ih, *att_b_hh, *query_w, *v_w, *loc_conv_w,
      *loc_dense_w, *dec_w_ih, *dec_w_hh, *dec_b_ih, *dec_b_hh, *proj_w, *proj_b, *gate_w, *gate_b;
  const float* pm;
  const float* G;        // folded context weights [4*Ra + 4*Rd + n_mel + 1][T_in], zero at positions >= len
  const uint8_t* masks;
  int T_cap, max_steps;
  float thr;
  const float* forced_mel;
  float *mel, *gate, *align;
  taco_granule* gran;
  taco_dec_layout lay;
  taco_dec_ctl* ctl;
#ifdef TACO_STAMPS
  unsigned long long* stamps;   // diagnostic build only: [2 workgroups][TACO_STAMP_FRAMES][TACO_STAMP_POINTS]
#endif
};

#ifdef TACO_STAMPS
constexpr int TACO_STAMP_FRAMES = 256, TACO_STAMP_POINTS = 12;
#define TACO_STAMP(k)                                                                                   \
  do {                                                                                                  \
    if (tid == 0 && (j == 0 || j == 200) && t < TACO_STAMP_FRAMES)                                      \
      a.stamps[((size_t)(j == 0 ? 0 : 1) * TACO_STAMP_FRAMES + t) * TACO_STAMP_POINTS + (k)] = wall_clock64(); \
  } while (0)
#else
#define TACO_STAMP(k) do {} while (0)
#endif

// Dot product of NC register-resident chunks with the activation vector in LDS.  Lane l holds
// columns 256*c + 4*l .. +3 of its row, which is exactly one 16-byte LDS read per chunk.
template <int NC>
__device__ __forceinline__ void dot_chunks(const float (*w0)[4], const float (*w1)[4], const float* x_lds, int lane,
                                           float (&acc)[2]) {
  float p[2][4] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float4 x = *reinterpret_cast<const float4*>(x_lds + c * TACO_CHUNK + lane * 4);
    p[0][0] = fmaf(w0[c][0], x.x, p[0][0]);
    p[0][1] = fmaf(w0[c][1], x.y, p[0][1]);
    p[0][2] = fmaf(w0[c][2], x.z, p[0][2]);
    p[0][3] = fmaf(w0[c][3], x.w, p[0][3]);
    p[1][0] = fmaf(w1[c][0], x.x, p[1][0]);
    p[1][1] = fmaf(w1[c][1], x.y, p[1][1]);
    p[1][2] = fmaf(w1[c][2], x.z, p[1][2]);
    p[1][3] = fmaf(w1[c][3], x.w, p[1][3]);
  }
  acc[0] += (p[0][0] + p[0][1]) + (p[0][2] + p[0][3]);   // one live partial per row between hand-offs
  acc[1] += (p[1][0] + p[1][1]) + (p[1][2] + p[1][3]);
}

// Per-lane partial of the dot product of an LDS row with an LDS vector, NC chunks of 256 columns:
// lane l takes columns 256*c + 4*l .. +3 (16-byte LDS reads, all issued before the first FMA).
// Both arrays are zero beyond the row's true length, so no bounds are checked.
template <int NC>
__device__ __forceinline__ float row_partial(const float* w_lds, const float* x_lds, int lane) {
  float4 wv[NC], xv[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    wv[c] = *reinterpret_cast<const float4*>(w_lds + c * TACO_CHUNK + lane * 4);
    xv[c] = *reinterpret_cast<const float4*>(x_lds + c * TACO_CHUNK + lane * 4);
  }
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    a0 = fmaf(wv[c].x, xv[c].x, a0);
    a1 = fmaf(wv[c].y, xv[c].y, a1);
    a2 = fmaf(wv[c].z, xv[c].z, a2);
    a3 = fmaf(wv[c].w, xv[c].w, a3);
  }
  return (a0 + a1) + (a2 + a3);
}

// Applies the four LSTM gate non-linearities in four lanes at once and finishes the cell update:
// lanes 4u..4u+3 of wave 0 hold gates i, f, g, o of unit u.  Returns h' in lane 4u (valid there).
__device__ __forceinline__ float lstm_cell_lanes(const float* gates_lds, const float* bias_lds, float* cell_lds,
                                                 int lane) {
  const int u = lane >> 2, gi = lane & 3;
  const float pre = gates_lds[lane & 15] + bias_lds[lane & 15];
  const float act = gi == 2 ? taco_tanh(pre) : taco_sigmoid(pre);
  const float ig = __shfl(act, (lane & ~3) + 0, 64);
  const float fg = __shfl(act, (lane & ~3) + 1, 64);
  const float gg = __shfl(act, (lane & ~3) + 2, 64);
  const float og = __shfl(act, (lane & ~3) + 3, 64);
  const float c = fg * cell_lds[u & 3] + ig * gg;
  const float hv = og * taco_tanh(c);
  if (gi == 0 && u < 4) cell_lds[u] = c;
  return hv;
}

// STD = true fixes the extents of the usual checkpoint geometry (prenet 256, both LSTMs 1024, attention
// 128 / 32 filters / kernel 31, 80 mel channels, 256 workgroups) at compile time: fewer live scalars and
// no bounds arithmetic in the frame loop.  STD = false takes every extent from the arguments.
template <bool STD>
__global__ __launch_bounds__(TACO_DEC_THREADS) void taco_decoder_kernel(const DecArgs a) {
  const int P = STD ? 256 : a.P;
  const int Ra = STD ? 1024 : a.Ra;
  const int Rd = STD ? 1024 : a.Rd;
  const int A = STD ? 128 : a.A;
  const int F = STD ? 32 : a.F;
  const int K = STD ? 31 : a.K;
  const int NMEL = STD ? 80 : a.n_mel;
  const int M = a.M;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = blockIdx.x;
  const int n_wg = STD ? 256 : a.n_wg;
  const int len = a.lens ? min(a.lens[a.b], a.T_in) : a.T_in;

  for (int i = tid; i < L_TOTAL; i += NT) lds[i] = 0.0f;
  __syncthreads();

  // ---------------- one-time loads: this workgroup's slices of the model and of the utterance ----
  const int Ua = (Ra + n_wg - 1) / n_wg;          // hidden units per workgroup (<= 4)
  const int Ud = (Rd + n_wg - 1) / n_wg;
  const int unit = wave >> 1, gate0 = (wave & 1) * 2;   // this wave's rows: gates gate0, gate0 + 1 of `unit`
  const int gu_a = j * Ua + unit, gu_d = j * Ud + unit;
  const bool act_a = unit < Ua && gu_a < Ra;       // wave-uniform
  const bool act_d = unit < Ud && gu_d < Rd;

  float wa[2][CH_A][4];
  float wd[2][CH_D][4];
  const float* G_att = a.G;                                  // rows in the order of att_w_ih's rows
  const float* G_dec = a.G + (size_t)4 * Ra * a.T_in;        // then dec_w_ih's rows
  const float* G_proj = G_dec + (size_t)4 * Rd * a.T_in;     // then the projection rows, stop-token row last
  {
    const int n_ih_a = P + M, n_ih_d = Ra + M;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int ra = (gate0 + r) * Ra + gu_a;
      const int rd = (gate0 + r) * Rd + gu_d;
#pragma unroll
      for (int c = 0; c < CH_A; ++c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = 0.0f;
          if (act_a) {
            if (c < TACO_NP) {
              const int col = c * TACO_CHUNK + lane * 4 + e;
              if (col < P) v = a.att_w_ih[(size_t)ra * n_ih_a + col];
            } else if (c < TACO_NP + NMF) {
              const int col = (c - TACO_NP) * TACO_CHUNK + lane * 4 + e;
              if (col < a.T_in) v = G_att[(size_t)ra * a.T_in + col];
            } else {
              const int col = (c - TACO_NP - NMF) * TACO_CHUNK + lane * 4 + e;
              if (col < Ra) v = a.att_w_hh[(size_t)ra * Ra + col];
            }
          }
          wa[r][c][e] = v;
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the one-time load phase from inflating register pressure
      }
#pragma unroll
      for (int c = 0; c < CH_D; ++c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = 0.0f;
          if (act_d) {
            if (c < TACO_NA) {
              const int col = c * TACO_CHUNK + lane * 4 + e;
              if (col < Ra) v = a.dec_w_ih[(size_t)rd * n_ih_d + col];
            } else if (c < TACO_NA + NMF) {
              const int col = (c - TACO_NA) * TACO_CHUNK + lane * 4 + e;
              if (col < a.T_in) v = G_dec[(size_t)rd * a.T_in + col];
            } else {
              const int col = (c - TACO_NA - NMF) * TACO_CHUNK + lane * 4 + e;
              if (col < Rd) v = a.dec_w_hh[(size_t)rd * Rd + col];
            }
          }
          wd[r][c][e] = v;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (act_a && lane == 0) lds[L_BIAS + wave * 2 + r] = a.att_b_ih[ra] + a.att_b_hh[ra];
      if (act_d && lane == 0) lds[L_BIAS + 16 + wave * 2 + r] = a.dec_b_ih[rd] + a.dec_b_hh[rd];
    }
  }

  const bool own_prenet = j < P;
  const bool own_q = j < A;
  const int proj_row = j - a.proj_off;
  const bool own_proj = proj_row >= 0 && proj_row <= NMEL;   // row n_mel is the stop-token row
  const int n_hc = Rd + M;
  _Pragma("unroll 1") for (int k = tid; k < P * NMEL; k += NT) {        // every workgroup keeps all of layer 1
    const int r = k / NMEL, c = k % NMEL;
    lds[L_W1T + c * (TACO_NP * TACO_CHUNK) + r] = a.prenet_w1[k];
  }
  if (own_prenet)
    _Pragma("unroll 1") for (int k = tid; k < P; k += NT) lds[L_W2ROW + k] = a.prenet_w2[(size_t)j * P + k];
  if (own_q)
    _Pragma("unroll 1") for (int k = tid; k < Ra; k += NT) lds[L_QROW + k] = a.query_w[(size_t)j * Ra + k];
  if (own_proj) {
    const float* src = proj_row < NMEL ? a.proj_w + (size_t)proj_row * n_hc : a.gate_w;
    _Pragma("unroll 1") for (int k = tid; k < Rd; k += NT) lds[L_PROJROW + k] = src[k];
    _Pragma("unroll 1") for (int k = tid; k < a.T_in; k += NT)
      lds[L_PROJROW + TACO_ND * TACO_CHUNK + k] = G_proj[(size_t)proj_row * a.T_in + k];
    if (tid == 0) lds[L_PROJB] = proj_row < NMEL ? a.proj_b[proj_row] : a.gate_b[0];
  }
  _Pragma("unroll 1") for (int k = tid; k < F * 2 * K; k += NT) lds[L_LOCCONV + k] = a.loc_conv_w[k];
  _Pragma("unroll 1") for (int k = tid; k < A * F; k += NT) {
    const int aa = k / F, f = k % F;
    lds[L_LOCDT + f * TACO_MAX_A + aa] = a.loc_dense_w[k];
  }
  _Pragma("unroll 1") for (int k = tid; k < A; k += NT) lds[L_V + k] = a.v_w[k];
  // per-utterance: processed-memory rows of my positions
  for (int slot = 0; slot < TACO_E_SLOTS; ++slot) {
    const int s = (n_wg - 1 - j) + slot * n_wg;
    if (s < len)
      _Pragma("unroll 1") for (int k = tid; k < A; k += NT) {
        const float v = a.pm[(size_t)s * A + k];
        lds[L_PMROW + slot * TACO_MAX_A + k] = v;
        lds[L_U + slot * TACO_MAX_A + k] = v;     // location term is zero before the first frame
      }
  }
  __syncthreads();

  taco_wait_ctx wctx;
  wctx.abort_word = (taco_gu32*)&a.ctl->abort_code;
  wctx.failed = 0;

  const bool forced = a.forced_mel != nullptr;
  const int pad = (K - 1) / 2;
  int stopped = 0;
  int t = 0;
  // attention-LSTM partial sums over [previous context | previous hidden]: computed at the end of
  // each frame for the next one, in the shadow of the frame-output hand-off (zero before frame 0)
  float acc_a[2] = {0.0f, 0.0f};
  // Prenet keep bytes travel global -> LDS by LDS-DMA one frame ahead (no registers held across the
  // frame); frame 0's are loaded here.
  unsigned char* keep_lds = reinterpret_cast<unsigned char*>(lds + L_KEEP);
  for (int k = tid; k < 2 * P; k += NT) keep_lds[(k / P) * 256 + (k % P)] = a.masks[k];
  __syncthreads();
  for (;; ++t) {
    // ============ frame boundary: previous frame's output + every workgroup's heartbeat ============
    if (t > 0) {
      const taco_granule* gp = a.gran + (size_t)((t - 1) & 1) * a.lay.per_parity;
      const unsigned ep_prev = (unsigned)t;
      if (tid <= NMEL) {
        const float v = taco_wait_one(gp + a.lay.mel + (j % TACO_REP_SMALL) * a.lay.mel_cap, tid, ep_prev, wctx, 0x100u);
        if (!forced || tid == NMEL) lds[L_MELIN + tid] = v;
      } else if (tid >= 128 && tid < 128 + n_wg) {
        (void)taco_wait_one(gp + a.lay.beat + (j % TACO_REP_SMALL) * a.lay.beat_cap, tid - 128, ep_prev, wctx, 0x200u);
      }
      if (forced && tid < NMEL) lds[L_MELIN + tid] = a.forced_mel[(size_t)tid * a.T_cap + (t - 1)];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the keep-byte DMA issued last frame has landed
      if (__syncthreads_or(wctx.failed)) break;
      if (!forced && taco_sigmoid(lds[L_MELIN + NMEL]) > a.thr) { stopped = 1; break; }
    }
    if (t == a.max_steps) break;
      TACO_STAMP(0);

    const unsigned ep = (unsigned)t + 1u;
    taco_granule* g = a.gran + (size_t)(t & 1) * a.lay.per_parity;
    // keep-bits of the two prenet dropouts: layer 1 for the row this thread finishes, layer 2 for row j
    constexpr int PW = TACO_NP * TACO_CHUNK;
    const unsigned char* keep_now = keep_lds + (t & 1) * 512;
    if (wave == NT / 64 - 1 && t + 1 < a.max_steps && lane * 4 < P) {
      // next frame's keep bytes, fetched a frame ahead (a first-touch line comes from HBM)
      const unsigned char* src = a.masks + (size_t)(t + 1) * 2 * P + lane * 4;
      float* dst = lds + L_KEEP + ((t + 1) & 1) * 128;
      typedef const __attribute__((address_space(1))) void* gptr_t;
      typedef __attribute__((address_space(3))) void* lptr_t;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 4, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(src + P), (lptr_t)(dst + 64), 4, 0, 0);
    }

    // ============ S1: prenet layer 1, ALL rows, replicated in every workgroup (20 k MACs buy one hop) ====
    // thread = (4 adjacent rows, one of 8 slices of the input columns); rows beyond P, columns beyond
    // n_mel are zero in LDS
    {
      const int rq = lane, kp = wave;
      float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll 2
      for (int i = 0; i < P1_KPER; ++i) {
        const int k = kp * P1_KPER + i;
        const float4 wv = *reinterpret_cast<const float4*>(lds + L_W1T + k * PW + rq * 4);
        const float x = lds[L_MELIN + k];
        acc.x = fmaf(wv.x, x, acc.x);
        acc.y = fmaf(wv.y, x, acc.y);
        acc.z = fmaf(wv.z, x, acc.z);
        acc.w = fmaf(wv.w, x, acc.w);
      }
      *reinterpret_cast<float4*>(lds + L_P1HALF + kp * PW + rq * 4) = acc;
    }
    taco_lds_barrier();
    if (tid < PW) {
      float v = 0.0f;
#pragma unroll
      for (int i = 0; i < P1_KPARTS; ++i) v += lds[L_P1HALF + i * PW + tid];
      v = v > 0.0f ? v : 0.0f;
      lds[L_P1 + tid] = keep_now[tid] ? v * 2.0f : 0.0f;
    }
    taco_lds_barrier();
    TACO_STAMP(1);

    // ============ S2: prenet layer 2, row j ============
    TACO_STAMP(2);
    if (own_prenet && wave == 0) {
      float v = taco_wave_sum(row_partial<TACO_NP>(lds + L_W2ROW, lds + L_P1, lane));
      v = v > 0.0f ? v : 0.0f;
      v = keep_now[256 + j] ? v * 2.0f : 0.0f;
      if (lane == 0) taco_publish_rep<TACO_REP_SMALL>(g + a.lay.p, a.lay.p_cap, j, ep, v);
    }

    // ============ S3: attention LSTM ============
    taco_gather<NT>(g + a.lay.p + (j % TACO_REP_SMALL) * a.lay.p_cap, P, ep, lds + L_XP, wctx, 0x400u);
    TACO_STAMP(3);
    if (act_a) {
      dot_chunks<TACO_NP>(wa[0], wa[1], lds + L_XP, lane, acc_a);
      const float s0 = taco_wave_sum(acc_a[0]);
      const float s1 = taco_wave_sum(acc_a[1]);
      if (lane == 0) {
        lds[L_GATES + wave * 2 + 0] = s0;
        lds[L_GATES + wave * 2 + 1] = s1;
      }
    }
    taco_lds_barrier();
    if (wave == 0) {
      const float hv = lstm_cell_lanes(lds + L_GATES, lds + L_BIAS, lds + L_CELLS, lane);
      const int u = lane >> 2;
      if ((lane & 3) == 0 && u < Ua && j * Ua + u < Ra) taco_publish_rep<TACO_REP_BIG>(g + a.lay.ah, a.lay.ah_cap, j * Ua + u, ep, hv);
    }
    // meanwhile: decoder-LSTM partial over the previous decoder hidden state
    float acc_d[2] = {0.0f, 0.0f};
    if (act_d) dot_chunks<TACO_ND>(wd[0] + TACO_NA + NMF, wd[1] + TACO_NA + NMF, lds + L_XDH, lane, acc_d);

    // ============ S4: query projection, row j ============
    TACO_STAMP(4);
    taco_gather<NT>(g + a.lay.ah + (j % TACO_REP_BIG) * a.lay.ah_cap, Ra, ep, lds + L_XAH, wctx, 0x500u);
    TACO_STAMP(5);
    if (own_q && wave == 0) {
      const float v = taco_wave_sum(row_partial<TACO_NA>(lds + L_QROW, lds + L_XAH, lane));
      if (lane == 0) taco_publish_rep<TACO_REP_SMALL>(g + a.lay.q, a.lay.q_cap, j, ep, v);
    }
    if (act_d) dot_chunks<TACO_NA>(wd[0], wd[1], lds + L_XAH, lane, acc_d);

    // ============ S5: energies of my positions ============
    taco_gather<NT>(g + a.lay.q + (j % TACO_REP_SMALL) * a.lay.q_cap, A, ep, lds + L_Q, wctx, 0x600u);
    TACO_STAMP(6);
    if (wave < TACO_E_SLOTS) {
      const int s = (n_wg - 1 - j) + wave * n_wg;
      if (s < len) {
        float acc = 0.0f;
        for (int aa = lane; aa < A; aa += 64)
          acc = fmaf(lds[L_V + aa], taco_tanh(lds[L_Q + aa] + lds[L_U + wave * TACO_MAX_A + aa]), acc);
        acc = taco_wave_sum(acc);
        if (lane == 0) taco_publish_rep<TACO_REP_SMALL>(g + a.lay.e, a.lay.e_cap, s, ep, acc);
      }
    }

    // ============ S6: softmax over all positions (every workgroup), context columns of mine ============
    taco_gather<NT>(g + a.lay.e + (j % TACO_REP_SMALL) * a.lay.e_cap, len, ep, lds + L_EW, wctx, 0x700u);
    TACO_STAMP(7);
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
      taco_lds_barrier();
      float mx = lds[L_RED];
#pragma unroll
      for (int i = 1; i < NT / 64; ++i) mx = fmaxf(mx, lds[L_RED + i]);
      float ws = 0.0f;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int s = tid + i * NT;
        ex[i] = s < len ? taco_exp(ev[i] - mx) : 0.0f;
        ws += ex[i];
      }
      ws = taco_wave_sum(ws);
      if (lane == 0) lds[L_RED + 16 + wave] = ws;
      taco_lds_barrier();
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
      taco_lds_barrier();
    }
    // heartbeat: every hand-off vector of frame t except the decoder hidden state and the frame
    // output has been consumed by this workgroup
    if (tid == NT - 64) taco_publish_rep<TACO_REP_SMALL>(g + a.lay.beat, a.lay.beat_cap, j, ep, 0.0f);

    // ============ S7: decoder LSTM; the context enters as G * w, no hand-off ============
    TACO_STAMP(8);
    const float* aw = lds + L_AW + HALO;          // this frame's attention weights, zero beyond len
    if (act_d) {
      dot_chunks<1>(wd[0] + TACO_NA, wd[1] + TACO_NA, aw, lane, acc_d);
      if (len > TACO_CHUNK) dot_chunks<1>(wd[0] + TACO_NA + 1, wd[1] + TACO_NA + 1, aw + TACO_CHUNK, lane, acc_d);
      const float s0 = taco_wave_sum(acc_d[0]);
      const float s1 = taco_wave_sum(acc_d[1]);
      if (lane == 0) {
        lds[L_GATES + wave * 2 + 0] = s0;
        lds[L_GATES + wave * 2 + 1] = s1;
      }
    }
    taco_lds_barrier();
    if (wave == 0) {
      const float hv = lstm_cell_lanes(lds + L_GATES, lds + L_BIAS + 16, lds + L_CELLS + 4, lane);
      const int u = lane >> 2;
      if ((lane & 3) == 0 && u < Ud && j * Ud + u < Rd) taco_publish_rep<TACO_REP_BIG>(g + a.lay.dh, a.lay.dh_cap, j * Ud + u, ep, hv);
    }
    TACO_STAMP(9);
    // meanwhile: context half of my projection row
    float acc_p = 0.0f;
    if (own_proj && wave == 1) acc_p = row_partial<NMF>(lds + L_PROJROW + TACO_ND * TACO_CHUNK, aw, lane);

    // while the decoder hidden state travels: location features of my positions for the NEXT frame
    // (they need only this frame's weights and cumulative weights, which are final now)
    {
      // location conv: 16 lanes per filter (32 filters x 16 = all 512 threads), taps split 16 ways
      const int f = tid >> 4, part = tid & 15;
      for (int slot = 0; slot < TACO_E_SLOTS; ++slot) {
        const int s = (n_wg - 1 - j) + slot * n_wg;
        if (s >= len) break;
        float acc = 0.0f;
        if (f < F) {
#pragma unroll 4
          for (int kk = part; kk < 2 * K; kk += 16) {
            const int c = kk >= K ? 1 : 0;
            const int k = kk - c * K;
            const float x = lds[(c ? L_CUM : L_AW) + HALO + s + k - pad];
            acc = fmaf(lds[L_LOCCONV + (f * 2 + c) * K + k], x, acc);
          }
        }
        acc = taco_sum16(acc);
        if (part == 0 && f < F) lds[L_CONVOUT + slot * TACO_MAX_F + f] = acc;
      }
    }
    taco_lds_barrier();
    {
      // location dense + processed memory: 4 lanes per attention dimension (128 x 4 = 512 threads)
      const int aa = tid >> 2, fq = tid & 3;
      for (int slot = 0; slot < TACO_E_SLOTS; ++slot) {
        const int s = (n_wg - 1 - j) + slot * n_wg;
        if (s >= len) break;
        float loc = 0.0f;
        if (aa < A) {
#pragma unroll 4
          for (int f = fq; f < F; f += 4)
            loc = fmaf(lds[L_LOCDT + f * TACO_MAX_A + aa], lds[L_CONVOUT + slot * TACO_MAX_F + f], loc);
        }
        loc = taco_sum4(loc);
        if (fq == 0 && aa < A) lds[L_U + slot * TACO_MAX_A + aa] = loc + lds[L_PMROW + slot * TACO_MAX_A + aa];
      }
    }

    // ============ S8: mel / stop-token projection, row proj_row ============
    TACO_STAMP(10);
    taco_gather<NT>(g + a.lay.dh + (j % TACO_REP_BIG) * a.lay.dh_cap, Rd, ep, lds + L_XDH, wctx, 0x900u);
    TACO_STAMP(11);
    if (own_proj && wave == 1) {
      acc_p += row_partial<TACO_ND>(lds + L_PROJROW, lds + L_XDH, lane);
      const float v = taco_wave_sum(acc_p) + lds[L_PROJB];
      if (lane == 0) {
        taco_publish_rep<TACO_REP_SMALL>(g + a.lay.mel, a.lay.mel_cap, proj_row, ep, v);
        if (proj_row < NMEL) a.mel[(size_t)proj_row * a.T_cap + t] = v;
        else a.gate[t] = v;
      }
    }
    // while the frame output travels: next frame's attention-LSTM partial sums over
    // [folded context | attention hidden state]
    acc_a[0] = 0.0f;
    acc_a[1] = 0.0f;
    if (act_a) {
      dot_chunks<1>(wa[0] + TACO_NP, wa[1] + TACO_NP, aw, lane, acc_a);
      if (len > TACO_CHUNK) dot_chunks<1>(wa[0] + TACO_NP + 1, wa[1] + TACO_NP + 1, aw + TACO_CHUNK, lane, acc_a);
      dot_chunks<TACO_NA>(wa[0] + TACO_NP + NMF, wa[1] + TACO_NP + NMF, lds + L_XAH, lane, acc_a);
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
  auto take = [&](int n, int reps, int32_t& cap) {
    const int at = off;
    cap = (n + 15) / 16 * 16;    // 128-byte aligned replicas
    off += cap * reps;
    return at;
  };
  lay.p = take(TACO_NP * TACO_CHUNK, TACO_REP_SMALL, lay.p_cap);
  lay.ah = take(TACO_NA * TACO_CHUNK, TACO_REP_BIG, lay.ah_cap);
  lay.q = take(TACO_MAX_A, TACO_REP_SMALL, lay.q_cap);
  lay.e = take(TACO_MAX_TIN, TACO_REP_SMALL, lay.e_cap);
  lay.ctx = 0; lay.ctx_cap = 0;   // the context is folded into G: no hand-off vector
  lay.dh = take(TACO_ND * TACO_CHUNK, TACO_REP_BIG, lay.dh_cap);
  lay.mel = take(128, TACO_REP_SMALL, lay.mel_cap);
  lay.beat = take(n_wg, TACO_REP_SMALL, lay.beat_cap);
  lay.per_parity = off;
}

size_t taco_decoder_granules(int n_wg) {
  taco_dec_layout lay;
  fill_layout(lay, n_wg);
  return (size_t)lay.per_parity * 2;
}

int taco_decoder_check_dims(const taco_dims& d, int n_wg) {
  if (d.P % 4 != 0) TACO_FAIL(TACO_EINVAL, "prenet_dim %d must be a multiple of 4", d.P);
  if (d.P > TACO_NP * TACO_CHUNK || d.P > n_wg) TACO_FAIL(TACO_EINVAL, "prenet_dim %d exceeds the decoder kernel's capacity %d", d.P, TACO_NP * TACO_CHUNK < n_wg ? TACO_NP * TACO_CHUNK : n_wg);
  if (d.M > TACO_NM * TACO_CHUNK) TACO_FAIL(TACO_EINVAL, "memory width %d exceeds the decoder's capacity %d", d.M, TACO_NM * TACO_CHUNK);
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
  a.proj_off = 0;   // projection rows on workgroups 0..n_mel; attention positions count down from the last workgroup
  const taco_weights& w = h->w;
  a.prenet_w1 = w.prenet_w1; a.prenet_w2 = w.prenet_w2;
  a.att_w_ih = w.att_w_ih; a.att_w_hh = w.att_w_hh; a.att_b_ih = w.att_b_ih; a.att_b_hh = w.att_b_hh;
  a.query_w = w.query_w; a.v_w = w.v_w; a.loc_conv_w = w.loc_conv_w; a.loc_dense_w = w.loc_dense_w;
  a.dec_w_ih = w.dec_w_ih; a.dec_w_hh = w.dec_w_hh; a.dec_b_ih = w.dec_b_ih; a.dec_b_hh = w.dec_b_hh;
  a.proj_w = w.proj_w; a.proj_b = w.proj_b; a.gate_w = w.gate_w; a.gate_b = w.gate_b;
  a.pm = pm; a.G = h->ws_G; a.masks = masks; a.T_cap = T_cap; a.max_steps = max_steps; a.thr = thr;
  a.forced_mel = forced_mel; a.mel = mel; a.gate = gate; a.align = align;
  a.gran = h->dec_gran; a.lay = h->dec_lay; a.ctl = h->dec_ctl;

  // folded context weights of this utterance: G = ctx_w_cat [rows][M] * memory^T [M][T_in], run as a
  // kernel-size-1 convolution over the position-major memory (columns >= len come out as zero)
  const int rc = taco_launch_conv(memory, h->ws_G, h->ctx_w_cat, nullptr, nullptr, nullptr, nullptr,
                                  lens ? lens + b : nullptr, 1, d.M, h->ctx_rows, 1, T_in, 0, /*in_time_major*/ 1, s);
  if (rc != TACO_OK) return rc;

  const size_t gran_bytes = (size_t)h->dec_lay.per_parity * 2 * sizeof(taco_granule);
  TACO_HIP(hipMemsetAsync(h->dec_gran, 0, gran_bytes, s));
  TACO_HIP(hipMemsetAsync(h->dec_ctl, 0, sizeof(taco_dec_ctl), s));
  TACO_HIP(hipMemsetAsync(mel, 0, sizeof(float) * (size_t)d.n_mel * T_cap, s));
  TACO_HIP(hipMemsetAsync(gate, 0, sizeof(float) * (size_t)T_cap, s));
  if (align) TACO_HIP(hipMemsetAsync(align, 0, sizeof(float) * (size_t)T_cap * T_in, s));

#ifdef TACO_STAMPS
  static unsigned long long* stamps_dev = nullptr;
  const size_t stamp_count = (size_t)2 * TACO_STAMP_FRAMES * TACO_STAMP_POINTS;
  if (!stamps_dev) TACO_HIP(hipMalloc((void**)&stamps_dev, stamp_count * 8));
  TACO_HIP(hipMemsetAsync(stamps_dev, 0, stamp_count * 8, s));
  a.stamps = stamps_dev;
#endif
  TACO_HIP(hipEventRecord(h->ev_start, s));
  const bool std_geom = d.P == 256 && d.R_a == 1024 && d.R_d == 1024 && d.A == 128 && d.F == 32 && d.K == 31 &&
                        d.n_mel == 80 && n_wg == 256;
  if (std_geom) hipLaunchKernelGGL(taco_decoder_kernel<true>, dim3(n_wg), dim3(NT), taco_decoder_lds_bytes(), s, a);
  else hipLaunchKernelGGL(taco_decoder_kernel<false>, dim3(n_wg), dim3(NT), taco_decoder_lds_bytes(), s, a);
  TACO_HIP(hipGetLastError());
  TACO_HIP(hipEventRecord(h->ev_stop, s));
  h->timing_valid = 1;

  // results: frame count and flags come back through the control block
  TACO_HIP(hipMemcpyAsync(h->dec_ctl_host, h->dec_ctl, sizeof(taco_dec_ctl), hipMemcpyDeviceToHost, s));
  TACO_HIP(hipStreamSynchronize(s));
  if (h->dec_ctl_host->abort_code != 0)
    TACO_FAIL(TACO_ETIMEDOUT, "decoder hand-off wait gave up (code %u, info 0x%x) after %u frames",
              h->dec_ctl_host->abort_code, h->dec_ctl_host->abort_info, h->dec_ctl_host->n_frames);
#ifdef TACO_STAMPS
  {
    std::vector<unsigned long long> st(stamp_count);
    TACO_HIP(hipMemcpy(st.data(), stamps_dev, stamp_count * 8, hipMemcpyDeviceToHost));
    const int nfr = (int)h->dec_ctl_host->n_frames < TACO_STAMP_FRAMES ? (int)h->dec_ctl_host->n_frames : TACO_STAMP_FRAMES;
    static const char* names[TACO_STAMP_POINTS] = {"frame top done", "S1 publish p1 + attn partial", "gather p1", "S2 + gather p",
        "S3 attn LSTM", "gather ah", "S4 q + gather q", "S5 energies + gather e", "S6 softmax",
        "S7 dec LSTM", "proj half + loc features", "gather dh"};
    for (int wg = 0; wg < 2 && nfr > 8; ++wg) {
      double acc[TACO_STAMP_POINTS + 1] = {0};
      int cnt = 0;
      for (int t = 4; t + 1 < nfr; ++t, ++cnt) {
        const unsigned long long* r = &st[((size_t)wg * TACO_STAMP_FRAMES + t) * TACO_STAMP_POINTS];
        const unsigned long long* nx = &st[((size_t)wg * TACO_STAMP_FRAMES + t + 1) * TACO_STAMP_POINTS];
        for (int k = 1; k < TACO_STAMP_POINTS; ++k) acc[k] += (double)(r[k] - r[k - 1]) * 0.01;
        acc[TACO_STAMP_POINTS] += (double)(nx[0] - r[TACO_STAMP_POINTS - 1]) * 0.01;
      }
      fprintf(stderr, "[taco stamps] workgroup %s, mean over %d frames (us):\n", wg == 0 ? "0" : "200", cnt);
      double tot = 0;
      for (int k = 1; k < TACO_STAMP_POINTS; ++k) { fprintf(stderr, "  %-32s %7.2f\n", names[k], acc[k] / cnt); tot += acc[k] / cnt; }
      fprintf(stderr, "  %-32s %7.2f\n  total %7.2f\n", "S8 proj + frame top (mel, beats)", acc[TACO_STAMP_POINTS] / cnt, tot + acc[TACO_STAMP_POINTS] / cnt);
    }
  }
#endif
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
  TACO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(taco_decoder_kernel<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)taco_decoder_lds_bytes()));
  TACO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(taco_decoder_kernel<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)taco_decoder_lds_bytes()));
  int per_cu = 0;
  TACO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, taco_decoder_kernel<true>, NT, taco_decoder_lds_bytes()));
  if (per_cu < 1) TACO_FAIL(TACO_EINVAL, "decoder kernel does not fit on a CU (occupancy query returned %d)", per_cu);
  TACO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, taco_decoder_kernel<false>, NT, taco_decoder_lds_bytes()));
  if (per_cu < 1) TACO_FAIL(TACO_EINVAL, "decoder kernel does not fit on a CU (occupancy query returned %d)", per_cu);
  (void)h;
  return TACO_OK;
}
