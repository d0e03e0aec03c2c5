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
//     4j..4j+3 of each LSTM, wave w holds two gate rows (2*(w&1), +1) of unit (w>>1), laid out so a
//     lane's 4 weights of a chunk meet one ds_read_b128 of the activation vector;
//   * workgroups exchange activation vectors through 8-byte {epoch, fp32} granules written with
//     agent-scope atomic exchanges whose result is dropped (carried out at the memory side, see
//     taco_send) and polled with agent-scope loads: the value is its own ready flag, so no grid
//     barrier and no cache fence sits between the dependent steps;
//   * the attention context never travels: ctx = sum_s w_s * memory_s enters the frame only through
//     W * ctx (both LSTMs' input rows and the projection rows), and W * ctx = (W * memory^T) * w.
//     G = W_ctx * memory^T is computed once per utterance (a GEMM before this launch); the kernel keeps
//     its G rows where the context columns of the weight rows would sit (registers / LDS) and
//     multiplies them with the attention weights every workgroup already has after its softmax;
//   * an all-gather costs the more the more workgroups produce and consume it, so the small matrices
//     live in three small groups of producer workgroups, and a frame's chain of hand-offs is
//       decoder hidden (all 256) -> P group (32 workgroups, 8 prenet rows each): layer-1 rows
//       layer-1 output (P)       -> P group: layer-2 rows
//       prenet output (P)        -> all: attention LSTM
//       attention hidden (all)   -> Q group (8: 16 query rows each, the location term and the energy
//                                   partial sums of their 16 attention dimensions for EVERY position)
//       energy partials (Q)      -> all: sum of the 8 partials, softmax, decoder LSTM with G * w
//     five dependent hand-offs per frame, all but one with a small group on one side; the M group
//     (27: projection / stop-token rows, 3 each) writes the frame's output beside that chain: only its
//     stop token is read back (by everyone, at the frame boundary);
//   * prenet layer 1 never sees a materialised frame: W1 (Wp [dh ; ctx] + b) = (W1 Wp_h) dh + (W1 Gp) w + W1 b,
//     folded once per model / utterance, so a P wave gets its layer-1 row straight from the decoder
//     hidden state (teacher forcing applies the plain W1 row to the given frame instead);
//   * the location layers are folded into one filter per attention dimension, E = dense * conv
//     (computed once per model), so a Q workgroup needs no intermediate and reads E through scalar loads;
//   * everything that does not depend on the value being waited for (the recurrent halves of the
//     LSTM dot products, next frame's location term) is computed before the wait.
//   * a consumer does not poll while it waits: it naps (s_sleep) for as long as the same hand-off took
//     in the previous frames and then fetches the vector once; polls that reach memory before the value
//     cost a further round trip each and delay the very stores they wait for (see taco_nap in
//     taco_device.h).  Replicas of every vector spread what polling remains over more cache lines;
//   * the P group runs ahead of the stop test (prenet layer 2 and its attention-LSTM rows of frame t+1
//     are published before frame t's stop token is looked at; results of a frame that never runs are
//     never read), because its values head the chain;
// Granule slots are double-buffered by frame parity.  Every workgroup consumes the prenet output, the
// attention hidden state, the energy partials and the stop token of frame t before it publishes its
// decoder hidden state of frame t, and nothing of frame t+2 can be published before all decoder hidden
// states of frame t+1 are out, so no word is overwritten while someone still has to read it.  Every
// wait is bounded by a poll budget and a global abort word, so the grid always drains.
#include "taco_device.h"
#include <math.h>
#include <stdlib.h>
#include <vector>
#ifdef TACO_STAMPS
#include <stdio.h>
#endif

namespace {

#ifndef TACO_Q_POS
#define TACO_Q_POS 64          // positions per Q workgroup
#endif
constexpr int NT = TACO_DEC_THREADS;
constexpr int NW = NT / 64;                          // waves per workgroup
// The context part of an LSTM row holds the row's folded context weights G[row][s], one column per
// input position: NMF chunks cover TACO_MAX_TIN positions.
constexpr int NMF = 2;
static_assert(NMF * TACO_CHUNK == TACO_FOLD_TIN, "folded context chunks must cover the positions kept on chip");
// A kernel instance holds NCTX <= NMF of them in registers: utterances of up to 256 positions run the
// instance with one chunk, which leaves 16 registers per lane to the rest of the frame.  Utterances of more than
// NMF * 256 positions run the LONG instance: one chunk in registers, the columns of the other positions read from
// global memory (L2) every frame - slower per frame, but no input is refused for its length up to TACO_MAX_TIN.
constexpr int NPOS_LONG = TACO_MAX_TIN / TACO_DEC_THREADS;     // positions per thread in the softmax of the LONG instance
constexpr int HALO = 16;                            // >= (K-1)/2 zero columns around the weights
constexpr int PW = TACO_NP * TACO_CHUNK;            // prenet width capacity (256)
constexpr int QD = TACO_MAX_A / TACO_NQ_WG;         // attention dimensions per Q workgroup (16)
static_assert(QD == 2 * NW && QD % 4 == 0, "a Q workgroup makes two query rows per wave, dimensions in quads");

// LDS carve (float offsets, all multiples of 4): a part every workgroup has, then one region whose
// content depends on the workgroup's group
constexpr int L_XP = 0;                                        // prenet out            [256]
constexpr int L_XAH = L_XP + PW;                               // attention hidden      [1024]
constexpr int L_XDH = L_XAH + TACO_NA * TACO_CHUNK;            // decoder hidden        [1024]
constexpr int L_PE = L_XDH + TACO_ND * TACO_CHUNK;             // energy partial sums   [groups][len]
constexpr int L_AW = L_PE + TACO_NQ_WG * TACO_MAX_TIN;         // attention weights, halo both sides
constexpr int L_CUM = L_AW + TACO_MAX_TIN + 2 * HALO;          // cumulative weights, halo both sides
constexpr int L_GATES = L_CUM + TACO_MAX_TIN + 2 * HALO;       // LSTM gate pre-activations [16]
constexpr int L_RED = L_GATES + 16;                            // block-reduction scratch [32]
constexpr int L_CELLS = L_RED + 32;                            // LSTM cell states [4 + 4]
constexpr int L_BIAS = L_CELLS + 8;                            // b_ih + b_hh of my rows [16 + 16]
constexpr int L_GATE_IN = L_BIAS + 32;                         // stop token of the previous frame, failure flag [4]
constexpr int L_X = L_GATE_IN + 4;                             // unnormalised attention weights of this frame
constexpr int L_ROLE = L_X + TACO_MAX_TIN;
static_assert(TACO_FOLD_TIN == NT && TACO_MAX_TIN % NT == 0, "the softmax takes one position per thread (LONG: TACO_MAX_TIN / NT)");
// P group: per wave one row of the prenet: layer 1 folded into the projection [1024 hidden + 512 folded
//          context] and its bias, the plain layer-1 row [128] (teacher forcing), the layer-2 row [256];
//          the gathered layer-1 output [256]; keep bytes [2 parities][2 layers][256 B]
constexpr int PROJ_ROW_LEN = (TACO_ND + NMF) * TACO_CHUNK;
constexpr int LP_WF = L_ROLE;
constexpr int LP_W1R = LP_WF + TACO_P_ROWS_WG * PROJ_ROW_LEN;
constexpr int LP_W2 = LP_W1R + TACO_P_ROWS_WG * 128;
constexpr int LP_BIAS = LP_W2 + TACO_P_ROWS_WG * PW;
constexpr int LP_H1 = LP_BIAS + TACO_P_ROWS_WG;
constexpr int LP_KEEP = LP_H1 + PW;
constexpr int LP_END = LP_KEEP + 2 * 2 * 64;
static_assert(TACO_MAX_MEL <= 128 && PW == 256, "prenet row capacities");
// Q group: 16 query rows [1024], location term + processed memory U[16][positions], v and query
//          of my dimensions [16], processed memory of my dimensions [16][256],
//          the folded location filter of my dimensions
constexpr int LQ_ROWS = L_ROLE;
constexpr int LQ_U = LQ_ROWS + QD * TACO_NA * TACO_CHUNK;
constexpr int LQ_V = LQ_U + QD * TACO_Q_POS;                    // (U is indexed by the position inside my range of TACO_Q_POS)
constexpr int LQ_PE8 = L_PE;                                   // per-wave energy partials live where the gathered partials
                                                                // will land (dead until then; a barrier separates the uses)
static_assert(NW <= TACO_NQ_WG, "per-wave partials fit the gathered-partials area");
constexpr int LQ_PM_POS = 256;                                 // processed memory of my dimensions is kept in LDS for
constexpr int LQ_PM = LQ_V + QD;                               // utterances up to this many positions [16][256]
constexpr int LQ_E = LQ_PM + QD * LQ_PM_POS;        // folded location filter as MFMA A operands [16 k-steps][64 lanes]
constexpr int LOC_KSTEPS = (2 * TACO_MAX_K + 3) / 4;           // 16 steps of 4 taps cover 2 x 31
static_assert(QD == 16 && LOC_KSTEPS == 16, "the location term is a 16 x 16 x (16 x 4) matrix-core product per 16 positions");
constexpr int LQ_END = LQ_E + LOC_KSTEPS * 64;
// M group: 3 projection rows [1024 hidden + 512 folded context], their biases [4]
constexpr int LM_ROWS = L_ROLE;
constexpr int LM_BIAS = LM_ROWS + TACO_MEL_ROWS_WG * PROJ_ROW_LEN;
constexpr int LM_END = LM_BIAS + 4;
constexpr int L_TOTAL = LP_END > LQ_END ? (LP_END > LM_END ? LP_END : LM_END) : (LQ_END > LM_END ? LQ_END : LM_END);
static_assert(L_TOTAL * 4 + 512 <= 160 * 1024, "decoder LDS carve exceeds a CU's 160 KiB");
static_assert(TACO_P_ROWS_WG == NW && TACO_MEL_ROWS_WG < NW, "one prenet row per wave; projection rows on waves 1..");

struct DecArgs {
  int P, M, Ra, Rd, A, F, K, n_mel, T_in, n_wg;
  const int32_t* lens;   // device, or NULL = T_in symbols
  int b;                 // utterance index into lens
  const float *prenet_w1, *prenet_w2, *wf_h, *wf_b, *att_w_ih, *att_w_hh, *att_b_ih, *att_b_hh, *query_w, *v_w, *loc_fold,
      *dec_w_ih, *dec_w_hh, *dec_b_ih, *dec_b_hh, *proj_w, *proj_b, *gate_w, *gate_b;
  const float* pm;
  const float* G;        // folded context weights [4*Ra + 4*Rd + n_mel + 1][T_in], zero at positions >= len
  const uint8_t* masks;
  int T_cap, max_steps;
  float thr;
  int shift_energies;    // 0: the energies are small enough for exp() without subtracting their maximum
  const float* forced_mel;
  float *mel, *gate, *align;
  int32_t* n_frames_out;   // this utterance's entries of the caller's result arrays
  uint8_t* reached_out;
  taco_granule* gran;
  taco_dec_layout lay;
  taco_dec_ctl* ctl;
  unsigned long long* frame_clock;   // optional: clock reading per frame (measurement), or NULL
  int frame_clock_cap;
#ifdef TACO_STAMPS
  unsigned long long* stamps;   // diagnostic build only: [3 workgroups][TACO_STAMP_FRAMES][TACO_STAMP_POINTS]
#endif
};

#ifdef TACO_STAMPS
constexpr int TACO_STAMP_FRAMES = 256, TACO_STAMP_POINTS = 12;
#define TACO_STAMP(k)                                                                                   \
  do {                                                                                                  \
    if (tid == 0 && (j == 0 || j == 32 || j == 200) && t < TACO_STAMP_FRAMES)                           \
      a.stamps[((size_t)(j == 0 ? 0 : j == 32 ? 1 : 2) * TACO_STAMP_FRAMES + t) * TACO_STAMP_POINTS + (k)] = wall_clock64(); \
  } while (0)
#define TACO_QSTAMP(k)                                                                                  \
  do {                                                                                                  \
    if (tid == 0 && j == 32 && t < TACO_STAMP_FRAMES)                                                   \
      a.stamps[((size_t)3 * TACO_STAMP_FRAMES + t) * TACO_STAMP_POINTS + (k)] = wall_clock64();         \
  } while (0)
#define TACO_QALL(k)                                                                                    \
  do {                                                                                                  \
    if (tid == 0 && is_q && q_half == 0 && t < TACO_STAMP_FRAMES)                                       \
      a.stamps[((size_t)(4 + q_grp) * TACO_STAMP_FRAMES + t) * TACO_STAMP_POINTS + (k)] = wall_clock64(); \
  } while (0)
#define TACO_PSTAMP(k)                                                                                  \
  do {                                                                                                  \
    if (tid == 0 && j == 0 && t < TACO_STAMP_FRAMES)                                                    \
      a.stamps[((size_t)3 * TACO_STAMP_FRAMES + t) * TACO_STAMP_POINTS + 6 + (k)] = wall_clock64();     \
  } while (0)
#else
#define TACO_PSTAMP(k) do {} while (0)
#define TACO_STAMP(k) do {} while (0)
#define TACO_QSTAMP(k) do {} while (0)
#define TACO_QALL(k) do {} while (0)
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

// One register-resident chunk of two rows against four activation values held in registers.
__device__ __forceinline__ void dot_reg(const float (&w0)[4], const float (&w1)[4], const float4 x, float (&acc)[2]) {
  acc[0] += (fmaf(w0[0], x.x, w0[1] * x.y)) + (fmaf(w0[2], x.z, w0[3] * x.w));
  acc[1] += (fmaf(w1[0], x.x, w1[1] * x.y)) + (fmaf(w1[2], x.z, w1[3] * x.w));
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

// The same for two LDS rows against one LDS vector: the vector is read once.
template <int NC>
__device__ __forceinline__ void row_partial2(const float* w0_lds, const float* w1_lds, const float* x_lds, int lane,
                                             float& r0, float& r1) {
  float4 a[NC], b[NC], xv[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    a[c] = *reinterpret_cast<const float4*>(w0_lds + c * TACO_CHUNK + lane * 4);
    b[c] = *reinterpret_cast<const float4*>(w1_lds + c * TACO_CHUNK + lane * 4);
    xv[c] = *reinterpret_cast<const float4*>(x_lds + c * TACO_CHUNK + lane * 4);
  }
  float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f, q0 = 0.0f, q1 = 0.0f, q2 = 0.0f, q3 = 0.0f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    p0 = fmaf(a[c].x, xv[c].x, p0); p1 = fmaf(a[c].y, xv[c].y, p1); p2 = fmaf(a[c].z, xv[c].z, p2); p3 = fmaf(a[c].w, xv[c].w, p3);
    q0 = fmaf(b[c].x, xv[c].x, q0); q1 = fmaf(b[c].y, xv[c].y, q1); q2 = fmaf(b[c].z, xv[c].z, q2); q3 = fmaf(b[c].w, xv[c].w, q3);
  }
  r0 = (p0 + p1) + (p2 + p3);
  r1 = (q0 + q1) + (q2 + q3);
}

// Applies the four LSTM gate non-linearities in four lanes at once and finishes the cell update:
// lanes 4u..4u+3 of wave 0 hold gates i, f, g, o of unit u.  Returns h' in all four lanes of the unit.
__device__ __forceinline__ float lstm_cell_lanes(const float* gates_lds, const float* bias_lds, float* cell_lds,
                                                 int lane) {
  const int u = lane >> 2, gi = lane & 3;
  const float pre = gates_lds[lane & 15] + bias_lds[lane & 15];
  const float act = gi == 2 ? taco_tanh(pre) : taco_sigmoid(pre);
  const float ig = taco_quad_bcast<0>(act);
  const float fg = taco_quad_bcast<1>(act);
  const float gg = taco_quad_bcast<2>(act);
  const float og = taco_quad_bcast<3>(act);
  const float c = fg * cell_lds[u & 3] + ig * gg;
  const float hv = og * taco_tanh(c);
  if (gi == 0 && u < 4) cell_lds[u] = c;
  return hv;
}

// STD = true fixes the extents of the usual checkpoint geometry (prenet 256, both LSTMs 1024, attention
// 128 / kernel 31, 80 mel channels, 256 workgroups) at compile time: fewer live scalars and no bounds
// arithmetic in the frame loop.  STD = false takes every extent from the arguments.
template <bool STD, int NCTX, bool LONG = false>
__global__ __launch_bounds__(TACO_DEC_THREADS) void taco_decoder_kernel(const DecArgs a) {
  static_assert(!LONG || NCTX == 1, "the LONG instance keeps one chunk of folded context weights in registers");
  constexpr int NPOS = LONG ? NPOS_LONG : 1;          // positions per thread in the softmax
  constexpr int ONCHIP = NCTX * TACO_CHUNK;           // positions whose folded LSTM columns are in registers
  constexpr int ROW_ONCHIP = LONG ? NMF * TACO_CHUNK : ONCHIP;   // ... and in the LDS rows of the P / M groups
  constexpr int CH_A = TACO_NP + NCTX + TACO_NA;    // chunks per attention-LSTM row (6 or 7)
  constexpr int CH_D = TACO_NA + NCTX + TACO_ND;    // chunks per decoder-LSTM row (9 or 10)
  const int P = STD ? 256 : a.P;
  const int Ra = STD ? 1024 : a.Ra;
  const int Rd = STD ? 1024 : a.Rd;
  const int A = STD ? 128 : a.A;
  const int K = STD ? 31 : a.K;
  const int NMEL = STD ? 80 : a.n_mel;
  const int M = a.M;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keep it scalar
  const int j = blockIdx.x;
  const int n_wg = STD ? 256 : a.n_wg;
  const int len = a.lens ? min(a.lens[a.b], a.T_in) : a.T_in;
  const int len4 = (len + 3) & ~3;                 // stride of an energy-partial group: keeps 16-byte reads aligned

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
            } else if (c < TACO_NP + NCTX) {
              const int col = (c - TACO_NP) * TACO_CHUNK + lane * 4 + e;
              if (col < a.T_in) v = G_att[(size_t)ra * a.T_in + col];
            } else {
              const int col = (c - TACO_NP - NCTX) * TACO_CHUNK + lane * 4 + e;
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
            } else if (c < TACO_NA + NCTX) {
              const int col = (c - TACO_NA) * TACO_CHUNK + lane * 4 + e;
              if (col < a.T_in) v = G_dec[(size_t)rd * a.T_in + col];
            } else {
              const int col = (c - TACO_NA - NCTX) * TACO_CHUNK + lane * 4 + e;
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

  // Groups (disjoint): P = workgroups [0, n_p_wg), Q = the next n_q_wg, M = the next n_m_wg.
  const int n_p_wg = (P + TACO_P_ROWS_WG - 1) / TACO_P_ROWS_WG;
  const int qd = (A + TACO_NQ_WG - 1) / TACO_NQ_WG;            // attention dimensions per Q workgroup (<= QD)
  const int n_q_wg = (A + qd - 1) / qd;
  const int n_m_wg = (NMEL + 1 + TACO_MEL_ROWS_WG - 1) / TACO_MEL_ROWS_WG;
  // The Q group is multiplied by the input length: one workgroup per group of dimensions AND per 64 positions
  // (location term, energy terms, partial sums of those positions).  That stage is the frame's longest piece of work
  // and its length goes with the positions a workgroup has: with 64 a wave's energy loop is one pass and its location
  // tiles one round.  (Measured against one workgroup per group of dimensions: 7.62 -> 7.44 us per frame at 102
  // positions, 8.36 -> 7.68 at 200, 10.6 -> 8.5 at 300, 12.9 -> 8.55 at 512; below 65 positions nothing changes.)
  const int Q_SPLIT = (len + TACO_Q_POS - 1) / TACO_Q_POS;
  const int q_off = n_p_wg, m_off = n_p_wg + n_q_wg * Q_SPLIT;
  const bool is_p = j < n_p_wg;
  const bool is_q = j >= q_off && j < m_off;
  const int q_grp = is_q ? (j - q_off) % n_q_wg : 0;           // my group of dimensions
  const int q_half = is_q ? (j - q_off) / n_q_wg : 0;          // my part of the positions
  const int q_lo = min(q_half * TACO_Q_POS, len4);                    // my positions [q_lo, q_hi); [q_lo, q_hi4) with the partials' padding
  const int q_hi4 = min(q_lo + TACO_Q_POS, len4);
  const int q_hi = min(q_hi4, len);
  const bool is_m = j >= m_off && j < m_off + n_m_wg;
  const int p_row = is_p ? j * TACO_P_ROWS_WG + wave : P;                          // >= P: none
  const bool own_p = p_row < P;
  const int q_base = is_q ? q_grp * qd : 0;                                        // my first attention dimension
  const int q_cnt = is_q ? min(qd, A - q_base) : 0;                                // my dimensions
  const int m_row = (is_m && wave >= 1 && wave <= TACO_MEL_ROWS_WG) ? (j - m_off) * TACO_MEL_ROWS_WG + (wave - 1)
                                                                    : NMEL + 1;    // row n_mel is the stop-token row
  const bool own_m = m_row <= NMEL;
  float* my_proj = lds + LM_ROWS + (wave - 1) * PROJ_ROW_LEN;                        // valid for own_m waves
  const int n_hc = Rd + M;
  float* my_wf = lds + LP_WF + wave * PROJ_ROW_LEN;                                  // valid for own_p waves
  if (own_p) {
    const float* G_pre = G_proj + (size_t)(NMEL + 1) * a.T_in;                       // layer 1 x projection's folded context rows
    _Pragma("unroll 1") for (int k = lane; k < Rd; k += 64) my_wf[k] = a.wf_h[(size_t)p_row * Rd + k];
    _Pragma("unroll 1") for (int k = lane; k < min(a.T_in, NMF * TACO_CHUNK); k += 64) my_wf[TACO_ND * TACO_CHUNK + k] = G_pre[(size_t)p_row * a.T_in + k];
    _Pragma("unroll 1") for (int k = lane; k < NMEL; k += 64) lds[LP_W1R + wave * 128 + k] = a.prenet_w1[(size_t)p_row * NMEL + k];
    _Pragma("unroll 1") for (int k = lane; k < P; k += 64) lds[LP_W2 + wave * PW + k] = a.prenet_w2[(size_t)p_row * P + k];
    if (lane == 0) lds[LP_BIAS + wave] = a.wf_b[p_row];
  }
  if (is_q) {
    _Pragma("unroll 1") for (int k = tid; k < q_cnt * Ra; k += NT) {
      const int r = k / Ra, c = k % Ra;
      lds[LQ_ROWS + r * (TACO_NA * TACO_CHUNK) + c] = a.query_w[(size_t)(q_base + r) * Ra + c];
    }
    if (tid < q_cnt) lds[LQ_V + tid] = a.v_w[q_base + tid];
    if (NCTX == 1 && !LONG)
      _Pragma("unroll 1") for (int k = tid; k < q_cnt * min(len, LQ_PM_POS); k += NT) {
        const int d = k % q_cnt, sp = k / q_cnt;
        lds[LQ_PM + d * LQ_PM_POS + sp] = a.pm[(size_t)sp * A + q_base + d];
      }
  }
  if (own_m) {
    const float* src = m_row < NMEL ? a.proj_w + (size_t)m_row * n_hc : a.gate_w;
    _Pragma("unroll 1") for (int k = lane; k < Rd; k += 64) my_proj[k] = src[k];
    _Pragma("unroll 1") for (int k = lane; k < min(a.T_in, NMF * TACO_CHUNK); k += 64) my_proj[TACO_ND * TACO_CHUNK + k] = G_proj[(size_t)m_row * a.T_in + k];
    if (lane == 0) lds[LM_BIAS + wave] = m_row < NMEL ? a.proj_b[m_row] : a.gate_b[0];
  }
  __syncthreads();

  taco_wait_ctx wctx;
  wctx.abort_word = (taco_gu32*)&a.ctl->abort_code;
  wctx.failed = 0;
  wctx.spun = 0;
  // this wave's naps before it polls for each of the five vectors of a frame (the stop token's single poller
  // needs none).  Measured: each nap shortens the frame, 2.2 us together.
  int nap_p = 0, nap_pe = 0, nap_dh = 0, nap_h1 = 0, nap_ah = 0;

  const bool forced = a.forced_mel != nullptr;
  const int pad = (K - 1) / 2;
  int stopped = 0;
  int t = 0;
  // attention-LSTM partial sums over [folded context | previous hidden]: computed at the end of
  // each frame for the next one, in the shadow of the frame-output hand-off (zero before frame 0)
  float acc_a[2] = {0.0f, 0.0f};
  // Prenet keep bytes are fetched one frame ahead; frame 0's are loaded here.  Only the P group uses them.
  unsigned char* keep_lds = reinterpret_cast<unsigned char*>(lds + LP_KEEP);
  if (is_p) {
    for (int k = tid; k < 2 * P; k += NT) keep_lds[(k / P) * 256 + (k % P)] = a.masks[k];
  }
  // Location term + processed memory of my 16 attention dimensions for every position, on the matrix
  // cores: U[dim][s] = pm[s][dim] + sum_{c,k} E[dim][c][k] * x_c[s + k - pad] is, per tile of 16 positions,
  // a 16 x 16 product over the 2K taps (v_mfma_f32_16x16x4_f32: lane l supplies A[row l&15][k = l>>4] and
  // B[k = l>>4][col l&15]; the result holds column l&15, rows 4*(l>>4) + r in register r).  A wave takes
  // every 8th tile.  With all-zero attention weights (before frame 0) U = processed memory.
  typedef float taco_f32x4 __attribute__((ext_vector_type(4)));
  if (is_q) {
    _Pragma("unroll 1") for (int i = tid; i < LOC_KSTEPS * 64; i += NT) {
      const int m = i >> 6, l = i & 63;
      const int ck = 4 * m + (l >> 4), dim = q_base + (l & 15);
      lds[LQ_E + i] = (ck < 2 * K && (l & 15) < q_cnt) ? a.loc_fold[(size_t)dim * 2 * K + ck] : 0.0f;
    }
  }
  constexpr int LOC_UNROLL = (STD && NCTX == 1 && !LONG) ? LOC_KSTEPS : 2;   // as far as the instance's registers allow without spilling
                                                                    // (4 and 8 measured slower in the two-chunk instance)
  // The two-chunk instance does the work in two passes (the wave's even and odd tiles), which a frame places in two
  // different shadows; the one-chunk instance does everything in pass 0 (splitting it measured 0.1 us slower there).
  auto update_location_term = [&](int pass) {
    const int col = lane & 15, kq = lane >> 4;
    // the products of one tile of 16 positions
    auto tile_products = [&](int s0) -> taco_f32x4 {
      taco_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll LOC_UNROLL
      for (int m = 0; m < LOC_KSTEPS; ++m) {
        int ck = 4 * m + kq;
        ck = ck < 2 * K ? ck : 0;                  // taps beyond 2K meet a zero in A; any valid address will do
        const int c = ck >= K ? 1 : 0;
        const float xv = lds[(c ? L_CUM : L_AW) + HALO + s0 + col + (ck - c * K) - pad];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[LQ_E + m * 64 + lane], xv, acc, 0, 0, 0);
      }
      return acc;
    };
    if (NCTX == 1 && !LONG) {
      // every position's processed memory is in LDS (the one-chunk instance runs inputs of up to LQ_PM_POS positions)
      if (pass != 0) return;
      _Pragma("unroll 1") for (int s0 = q_lo + wave * 16; s0 < q_hi; s0 += NW * 16) {
        const taco_f32x4 acc = tile_products(s0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = 4 * kq + r;
          lds[LQ_U + d * TACO_Q_POS + (s0 - q_lo) + col] = acc[r] + lds[LQ_PM + d * LQ_PM_POS + s0 + col];
        }
      }
    } else {
      // processed memory from global memory (L2), asked for before the products so that the round trip runs
      // beside them.  (Keeping the first LQ_PM_POS positions in LDS and fetching only the rest, ahead of the
      // other tiles, cost registers this instance does not have: it spilled and came out slower.)
      _Pragma("unroll 1") for (int s0 = q_lo + (wave + NW * pass) * 16; s0 < q_hi; s0 += 2 * NW * 16) {
        const int s = s0 + col;
        float pmv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (s < len && 4 * kq + r < q_cnt) pmv[r] = a.pm[(size_t)s * A + q_base + 4 * kq + r];
        const taco_f32x4 acc = tile_products(s0);
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[LQ_U + (4 * kq + r) * TACO_Q_POS + (s - q_lo)] = acc[r] + pmv[r];
      }
    }
  };
  __syncthreads();
  if (is_q) { update_location_term(0); update_location_term(1); }
  __syncthreads();
  for (;; ++t) {
    const unsigned ep = (unsigned)t + 1u;
    taco_granule* g = a.gran + (size_t)(t & 1) * a.lay.per_parity;
    // P group first: prenet layer 2 of this frame heads the frame's chain of hand-offs, so it does not
    // wait for the stop test.  Should the utterance turn out to have ended, the published values are
    // never read (their slots are cleared before the next launch).
    int stop_known = 0;
    taco_sent p_sent = {nullptr, 0};      // what this wave publishes as prenet output, and where (P group)
    TACO_PSTAMP(0);
    if (is_p && t < a.max_steps) {
      // the poll for the previous frame's stop token travels while the group works: issued here (by every thread,
      // one address), looked at right after the gather below, with whose polls it has returned
      taco_granule stop_early = 0;
      if (t > 0)
        stop_early = __hip_atomic_load((const taco_ggran*)(a.gran + (size_t)((t - 1) & 1) * a.lay.per_parity + a.lay.mel +
                                                            (j % TACO_REP_SMALL) * a.lay.mel_cap) + NMEL, TACO_RLX_AGENT);
      const unsigned char* keep_now = keep_lds + (t & 1) * 512;
      // next frame's keep bytes, fetched a frame ahead (a first-touch line comes from HBM): loaded here, stored to
      // LDS after the gather below, by when they have arrived.  No branch around either (every wave fetches and
      // stores the same words, clamped to valid ones), so that the compiler sees them waited for on every path
      // and has no reason to wait for ALL outstanding memory operations later, with the publish on its way.
      const int keep_lane = min(lane * 4, P - 4);
      const unsigned char* keep_src = a.masks + (size_t)min(t + 1, a.max_steps - 1) * 2 * P + keep_lane;
      const unsigned keep_next0 = *reinterpret_cast<const unsigned*>(keep_src);
      const unsigned keep_next1 = *reinterpret_cast<const unsigned*>(keep_src + P);
      // ---- prenet layer 2, one row per wave, on the layer-1 output the P group exchanged (all zero for
      //      frame 0: the go frame is zero and the layers have no bias) ----
      if (t > 0) taco_gather<NT>(g + a.lay.h1 + (j % TACO_REP_SMALL) * a.lay.h1_cap, P, ep, lds + LP_H1, wctx, 0x300u, nap_h1);
      TACO_PSTAMP(1);
      {
        unsigned* dst = reinterpret_cast<unsigned*>(lds + LP_KEEP + ((t + 1) & 1) * 128);
        dst[keep_lane >> 2] = keep_next0;
        dst[64 + (keep_lane >> 2)] = keep_next1;
      }
      // (looked at before the publish below: a wait for a load that follows a store also waits for the store
      // to be acknowledged)
      stop_known = t > 0 && (unsigned)(stop_early >> 32) == ep - 1u;
      if (tid == NT - 1 && stop_known) lds[L_GATE_IN] = __uint_as_float((unsigned)stop_early);
      if (own_p) {
        float v = taco_wave_sum(row_partial<TACO_NP>(lds + LP_W2 + wave * PW, lds + LP_H1, lane));
        v = v > 0.0f ? v : 0.0f;
        v = keep_now[256 + p_row] ? v * 2.0f : 0.0f;
        p_sent = taco_publish_lanes_kept<TACO_REP_SMALL>(g + a.lay.p, a.lay.p_cap, p_row, ep, v, lane);
      }
      TACO_PSTAMP(2);
    }
    // ============ frame boundary: the previous frame's stop token ============
    // 0 = go on, 1 = the previous frame was the last, 2 = a wait has given up
    auto stop_test = [&]() -> int {
      if (t == 0) return 0;
      const taco_granule* gp = a.gran + (size_t)((t - 1) & 1) * a.lay.per_parity;
      const taco_granule* melv = gp + a.lay.mel + (j % TACO_REP_SMALL) * a.lay.mel_cap;
      // One thread fetches the stop token.  (The P group does not poll for its layer-1 vector together
      // with it, although both are published at the same time: 32 x 256 early pollers on those lines delay
      // the stores they are waiting for by more than the second round trip costs.)
      if (tid == NT - 1 && !stop_known) lds[L_GATE_IN] = taco_wait_one(melv, NMEL, (unsigned)t, wctx, 0x100u);
      if (wctx.failed) lds[L_GATE_IN + 1] = 1.0f;
      taco_lds_barrier_only();
      if (lds[L_GATE_IN + 1] != 0.0f) return 2;
      return (!forced && taco_sigmoid(lds[L_GATE_IN]) > a.thr) ? 1 : 0;
    };
    // The P group, having published this frame's prenet output, goes on to its attention-LSTM rows before it
    // looks at the stop token: its hidden-state values are the last the Q group waits for.  (What it computes
    // for a frame that turns out not to exist is never read.)
    const bool p_ahead = is_p && t < a.max_steps;
    if (!p_ahead) {
      const int st = stop_test();
      if (st == 1) stopped = 1;
      if (st != 0) break;
    }
    if (t == a.max_steps) break;
    TACO_STAMP(0);
    // Q group: this frame's location term from the previous frame's weights; the first half of the tiles while the
    // P group works, the second half after this workgroup's attention-LSTM values have left (a Q workgroup that
    // arrives late at its attention-LSTM rows holds up every workgroup's gather of the hidden state)
    if (is_q && t > 0) update_location_term(0);
    TACO_STAMP(1);

    // ============ S3: attention LSTM ============
    taco_gather<NT>(g + a.lay.p + (j % TACO_REP_SMALL) * a.lay.p_cap, P, ep, lds + L_XP, wctx, 0x400u, nap_p);
    taco_keep_alive(p_sent);
    TACO_STAMP(2);
    if (act_a) {
      dot_chunks<TACO_NP>(wa[0], wa[1], lds + L_XP, lane, acc_a);
      float s0, s1;
      taco_wave_sum2(acc_a[0], acc_a[1], s0, s1);
      if (lane == 0) {
        lds[L_GATES + wave * 2 + 0] = s0;
        lds[L_GATES + wave * 2 + 1] = s1;
      }
    }
    taco_lds_barrier();
    if (wave == 0) {
      const float hv = lstm_cell_lanes(lds + L_GATES, lds + L_BIAS, lds + L_CELLS, lane);
      const int u = lane >> 2;
      if (u < Ua && j * Ua + u < Ra)      // the four lanes of a unit hold the same value: they share the replicas
        for (int r = lane & 3; r < TACO_REP_BIG; r += 4) taco_publish(g + a.lay.ah + (size_t)r * a.lay.ah_cap, j * Ua + u, ep, hv);
    }
    TACO_STAMP(3);
    if (p_ahead) {
      const int st = stop_test();
      if (st == 1) stopped = 1;
      if (st != 0) break;
    }
    if (is_q && t > 0) update_location_term(1);
    // meanwhile: decoder-LSTM partial over the previous decoder hidden state
    float acc_d[2] = {0.0f, 0.0f};
    if (act_d) dot_chunks<TACO_ND>(wd[0] + TACO_NA + NCTX, wd[1] + TACO_NA + NCTX, lds + L_XDH, lane, acc_d);

    // ============ S4: query rows and energy partial sums (Q group) ============
    taco_gather<NT>(g + a.lay.ah + (j % TACO_REP_BIG) * a.lay.ah_cap, Ra, ep, lds + L_XAH, wctx, 0x500u, nap_ah);
    TACO_STAMP(4);
    if (is_q) {
      TACO_QSTAMP(0);
      TACO_QALL(0);
      // wave w: query values of dimensions 2w, 2w+1, then their energy terms for every position (lane =
      // position); the 8 waves' partial sums meet in LDS
      float qv[2], vv[2];
      row_partial2<TACO_NA>(lds + LQ_ROWS + (wave * 2) * (TACO_NA * TACO_CHUNK), lds + LQ_ROWS + (wave * 2 + 1) * (TACO_NA * TACO_CHUNK),
                            lds + L_XAH, lane, qv[0], qv[1]);
      taco_wave_sum2(qv[0], qv[1], qv[0], qv[1]);
      vv[0] = lds[LQ_V + wave * 2];           // rows beyond my dimensions: zero row, v = 0, finite U
      vv[1] = lds[LQ_V + wave * 2 + 1];
      TACO_QSTAMP(1);
      _Pragma("unroll 2") for (int s = q_lo + lane; s < q_hi; s += 64) {
        const float e0 = vv[0] * taco_tanh(qv[0] + lds[LQ_U + (wave * 2 + 0) * TACO_Q_POS + (s - q_lo)]);
        const float e1 = vv[1] * taco_tanh(qv[1] + lds[LQ_U + (wave * 2 + 1) * TACO_Q_POS + (s - q_lo)]);
        lds[LQ_PE8 + wave * TACO_MAX_TIN + s] = e0 + e1;
      }
      TACO_QSTAMP(2);
      taco_lds_barrier();
      TACO_QSTAMP(3);
      {
        const int grp = q_grp;
        _Pragma("unroll 1") for (int s = q_lo + tid; s < q_hi4; s += NT) {
          float e = 0.0f;
          if (s < len) {
#pragma unroll
            for (int w8 = 0; w8 < NW; ++w8) e += lds[LQ_PE8 + w8 * TACO_MAX_TIN + s];
          }
          taco_publish_rep<TACO_REP_PE>(g + a.lay.pe, a.lay.pe_cap, grp * len4 + s, ep, e);
        }
      }
      TACO_QSTAMP(4);
      TACO_QALL(1);
      taco_lds_barrier();                          // the per-wave partials' area is about to receive the gathered ones
      TACO_QSTAMP(5);
    }
    if (act_d) dot_chunks<TACO_NA>(wd[0], wd[1], lds + L_XAH, lane, acc_d);
    TACO_STAMP(5);

    // ============ S6: energies = sum of the groups' partials; softmax over all positions (every workgroup) ============
    taco_gather<NT>(g + a.lay.pe + (j % TACO_REP_PE) * a.lay.pe_cap, n_q_wg * len4, ep, lds + L_PE, wctx, 0x700u, nap_pe);
    TACO_STAMP(6);
    // One position per thread: the energy is the sum of the groups' partials; exp() without a shift where
    // the energies are known to be small (softmax does not depend on the shift: it only guards the range),
    // else shifted by their maximum, which costs one more exchange through LDS.  The waves' partial sums and
    // the unnormalised weights meet in LDS; then every wave picks the weights up in the layout of its
    // folded-context columns (lane l: positions 4l..4l+3 of each chunk) and normalises them in registers.
    float4 wgt[NCTX];
    float w_own[NPOS];
    float inv_sum;
    {
      float e[NPOS], x[NPOS];
#pragma unroll
      for (int q = 0; q < NPOS; ++q) {
        const int sq = tid + q * NT;
        e[q] = 0.0f;
        if (sq < len)
          for (int gq = 0; gq < n_q_wg; ++gq) e[q] += lds[L_PE + gq * len4 + sq];
      }
      float mx = 0.0f;
      if (a.shift_energies) {
        float m = tid < len ? e[0] : -INFINITY;
#pragma unroll
        for (int q = 1; q < NPOS; ++q) m = fmaxf(m, tid + q * NT < len ? e[q] : -INFINITY);
        m = taco_wave_max(m);
        if (lane == 0) lds[L_RED + 8 + wave] = m;
        taco_lds_barrier();
        const float4 m0 = *reinterpret_cast<const float4*>(lds + L_RED + 8);
        const float4 m1 = *reinterpret_cast<const float4*>(lds + L_RED + 12);
        mx = fmaxf(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)), fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w)));
      }
      float xs = 0.0f;
#pragma unroll
      for (int q = 0; q < NPOS; ++q) {
        x[q] = tid + q * NT < len ? taco_exp(e[q] - mx) : 0.0f;
        lds[L_X + tid + q * NT] = x[q];
        xs += x[q];
      }
      const float part = taco_wave_sum(xs);
      if (lane == 0) lds[L_RED + wave] = part;
      taco_lds_barrier();
      const float4 s0 = *reinterpret_cast<const float4*>(lds + L_RED);
      const float4 s1 = *reinterpret_cast<const float4*>(lds + L_RED + 4);
      const float inv = 1.0f / (((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w)));
      inv_sum = inv;
#pragma unroll
      for (int c = 0; c < NCTX; ++c) {
        const float4 xv = *reinterpret_cast<const float4*>(lds + L_X + c * TACO_CHUNK + lane * 4);
        wgt[c] = make_float4(xv.x * inv, xv.y * inv, xv.z * inv, xv.w * inv);
      }
#pragma unroll
      for (int q = 0; q < NPOS; ++q) w_own[q] = x[q] * inv;
    }

    // ============ S7: decoder LSTM; the context enters as G * w, no hand-off ============
    TACO_STAMP(7);
    const float* aw = lds + L_AW + HALO;          // this frame's attention weights, zero beyond len
    if (act_d) {
#pragma unroll
      for (int c = 0; c < NCTX; ++c) dot_reg(wd[0][TACO_NA + c], wd[1][TACO_NA + c], wgt[c], acc_d);
      if (LONG) {
        // the folded context columns of the positions beyond the registers': from global memory, against the
        // unnormalised weights every thread has just left in LDS
        float g0 = 0.0f, g1 = 0.0f;
        const float* r0 = G_dec + (size_t)(gate0 * Rd + gu_d) * a.T_in;
        const float* r1 = G_dec + (size_t)((gate0 + 1) * Rd + gu_d) * a.T_in;
        for (int sp = ONCHIP + lane; sp < len; sp += 64) {
          const float xv = lds[L_X + sp];
          g0 = fmaf(r0[sp], xv, g0);
          g1 = fmaf(r1[sp], xv, g1);
        }
        acc_d[0] = fmaf(g0, inv_sum, acc_d[0]);
        acc_d[1] = fmaf(g1, inv_sum, acc_d[1]);
      }
      float s0, s1;
      taco_wave_sum2(acc_d[0], acc_d[1], s0, s1);
      if (lane == 0) {
        lds[L_GATES + wave * 2 + 0] = s0;
        lds[L_GATES + wave * 2 + 1] = s1;
      }
    }
    taco_lds_barrier();
    if (wave == 0) {
      const float hv = lstm_cell_lanes(lds + L_GATES, lds + L_BIAS + 16, lds + L_CELLS + 4, lane);
      const int u = lane >> 2;
      if (u < Ud && j * Ud + u < Rd)
        for (int r = lane & 3; r < TACO_REP_BIG; r += 4) taco_publish(g + a.lay.dh + (size_t)r * a.lay.dh_cap, j * Ud + u, ep, hv);
    }
    // Behind the publish: each thread leaves its position's weight in LDS for the projection rows, the location
    // term and the next frame's attention-LSTM partial sums, accumulates it, and the last workgroup writes the
    // alignment row.
#pragma unroll
    for (int q = 0; q < NPOS; ++q) {
      const int sq = tid + q * NT;
      lds[L_AW + HALO + sq] = w_own[q];
      if (sq < len) {
        lds[L_CUM + HALO + sq] += w_own[q];
        if (j == n_wg - 1 && a.align) a.align[(size_t)t * a.T_in + sq] = w_own[q];
      }
    }
    if (a.frame_clock && j == n_wg - 1 && tid == 0 && t < a.frame_clock_cap) a.frame_clock[t] = wall_clock64();
    taco_lds_barrier();
    TACO_STAMP(8);
    // meanwhile: folded-context half of my projection rows (M group)
    float acc_p = 0.0f;
    constexpr int NROW = ROW_ONCHIP / TACO_CHUNK;
    if (own_m) acc_p = row_partial<NROW>(my_proj + TACO_ND * TACO_CHUNK, aw, lane);
    if (own_p && !forced) acc_p = row_partial<NROW>(my_wf + TACO_ND * TACO_CHUNK, aw, lane) + (lane == 0 ? lds[LP_BIAS + wave] : 0.0f);   // ... of my folded layer-1 row, with its bias
    if (LONG && (own_m || (own_p && !forced))) {
      // ... and the positions beyond the LDS rows', from global memory
      const float* grow = own_m ? G_proj + (size_t)m_row * a.T_in : G_proj + (size_t)(NMEL + 1 + p_row) * a.T_in;
      for (int sp = ROW_ONCHIP + lane; sp < len; sp += 64) acc_p = fmaf(grow[sp], aw[sp], acc_p);
    }
    TACO_STAMP(9);

    // ============ S8: mel / stop-token projection rows (M group), next frame's prenet layer 1 (P group) ============
    taco_gather<NT>(g + a.lay.dh + (j % TACO_REP_BIG) * a.lay.dh_cap, Rd, ep, lds + L_XDH, wctx, 0x900u, nap_dh);
    TACO_STAMP(10);
    if (own_m) {
      acc_p += row_partial<TACO_ND>(my_proj, lds + L_XDH, lane);
      const float v = taco_wave_sum(acc_p) + lds[LM_BIAS + wave];
      if (m_row == NMEL) taco_publish_lanes<TACO_REP_SMALL>(g + a.lay.mel, a.lay.mel_cap, m_row, ep, v, lane);
      if (lane == 0) {
        if (m_row < NMEL) a.mel[(size_t)m_row * a.T_cap + t] = v;
        else a.gate[t] = v;
      }
    }
    // P group: the next frame's prenet layer-1 rows.  Free running, layer 1 is applied to this frame's
    // output without materialising it: W1 (Wp [dh ; ctx] + b) = (W1 Wp_h) dh + (W1 Gp) w + W1 b, one
    // folded row per wave.  Teacher forcing applies the plain layer-1 row to the given frame.
    if (own_p && t + 1 < a.max_steps) {
      float x1;
      if (forced) {
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int k = lane + 64 * i;
          if (k < NMEL) acc = fmaf(lds[LP_W1R + wave * 128 + k], a.forced_mel[(size_t)k * a.T_cap + t], acc);
        }
        x1 = taco_wave_sum(acc);
      } else {
        x1 = taco_wave_sum(row_partial<TACO_ND>(my_wf, lds + L_XDH, lane) + acc_p);
      }
      x1 = x1 > 0.0f ? x1 : 0.0f;
      x1 = keep_lds[((t + 1) & 1) * 512 + p_row] ? x1 * 2.0f : 0.0f;
      taco_granule* gn = a.gran + (size_t)((t + 1) & 1) * a.lay.per_parity;
      taco_publish_lanes<TACO_REP_SMALL>(gn + a.lay.h1, a.lay.h1_cap, p_row, ep + 1u, x1, lane);
    }
    TACO_STAMP(11);
    // while the frame output travels: next frame's attention-LSTM partial sums over
    // [folded context | attention hidden state]
    acc_a[0] = 0.0f;
    acc_a[1] = 0.0f;
    if (act_a) {
      dot_chunks<1>(wa[0] + TACO_NP, wa[1] + TACO_NP, aw, lane, acc_a);
      if (NCTX > 1) dot_chunks<1>(wa[0] + TACO_NP + NCTX - 1, wa[1] + TACO_NP + NCTX - 1, aw + TACO_CHUNK, lane, acc_a);
      if (LONG) {
        const float* r0 = G_att + (size_t)(gate0 * Ra + gu_a) * a.T_in;
        const float* r1 = G_att + (size_t)((gate0 + 1) * Ra + gu_a) * a.T_in;
        float g0 = 0.0f, g1 = 0.0f;
        for (int sp = ONCHIP + lane; sp < len; sp += 64) {
          const float xv = aw[sp];
          g0 = fmaf(r0[sp], xv, g0);
          g1 = fmaf(r1[sp], xv, g1);
        }
        acc_a[0] += g0;
        acc_a[1] += g1;
      }
      dot_chunks<TACO_NA>(wa[0] + TACO_NP + NCTX, wa[1] + TACO_NP + NCTX, lds + L_XAH, lane, acc_a);
    }
  }

  if (j == 0 && tid == 0) {
    const unsigned reached = (!stopped && !forced && t == a.max_steps) ? 1u : 0u;
    a.ctl->n_frames = (uint32_t)t;
    a.ctl->reached_max = reached;
    *a.n_frames_out = t;
    *a.reached_out = (uint8_t)reached;
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
  lay.h1 = take(TACO_NP * TACO_CHUNK, TACO_REP_SMALL, lay.h1_cap);
  lay.p = take(TACO_NP * TACO_CHUNK, TACO_REP_SMALL, lay.p_cap);
  lay.ah = take(TACO_NA * TACO_CHUNK, TACO_REP_BIG, lay.ah_cap);
  lay.pe = take(TACO_NQ_WG * TACO_MAX_TIN, TACO_REP_PE, lay.pe_cap);
  lay.dh = take(TACO_ND * TACO_CHUNK, TACO_REP_BIG, lay.dh_cap);
  lay.mel = take(128, TACO_REP_SMALL, lay.mel_cap);
  lay.per_parity = off;
}

size_t taco_decoder_granules(int n_wg) {
  taco_dec_layout lay;
  fill_layout(lay, n_wg);
  return (size_t)lay.per_parity * 2;
}

int taco_decoder_check_dims(const taco_dims& d, int n_wg) {
  if (d.P % 4 != 0) TACO_FAIL(TACO_EINVAL, "prenet_dim %d must be a multiple of 4", d.P);
  if (d.P > TACO_NP * TACO_CHUNK) TACO_FAIL(TACO_EINVAL, "prenet_dim %d exceeds the decoder kernel's capacity %d", d.P, TACO_NP * TACO_CHUNK);
  if (d.M > TACO_NM * TACO_CHUNK) TACO_FAIL(TACO_EINVAL, "memory width %d exceeds the decoder's capacity %d", d.M, TACO_NM * TACO_CHUNK);
  if (d.R_a > TACO_NA * TACO_CHUNK || d.R_a > TACO_UNITS_PER_WG * n_wg) TACO_FAIL(TACO_EINVAL, "attention_rnn_dim %d exceeds the decoder kernel's capacity", d.R_a);
  if (d.R_d > TACO_ND * TACO_CHUNK || d.R_d > TACO_UNITS_PER_WG * n_wg) TACO_FAIL(TACO_EINVAL, "decoder_rnn_dim %d exceeds the decoder kernel's capacity", d.R_d);
  if (d.A > TACO_MAX_A) TACO_FAIL(TACO_EINVAL, "attention_dim %d exceeds the decoder kernel's capacity %d", d.A, TACO_MAX_A);
  if (d.F < 1) TACO_FAIL(TACO_EINVAL, "attention_location_n_filters %d", d.F);
  if (d.K > TACO_MAX_K || (d.K & 1) == 0) TACO_FAIL(TACO_EINVAL, "attention_location_kernel_size %d must be odd and <= %d", d.K, TACO_MAX_K);
  if (d.n_mel > TACO_MAX_MEL) TACO_FAIL(TACO_EINVAL, "n_mel_channels %d exceeds %d", d.n_mel, TACO_MAX_MEL);
  const int qd = (d.A + TACO_NQ_WG - 1) / TACO_NQ_WG;
  const int groups = (d.P + TACO_P_ROWS_WG - 1) / TACO_P_ROWS_WG + ((TACO_MAX_TIN + TACO_Q_POS - 1) / TACO_Q_POS) * ((d.A + qd - 1) / qd) +      // (the Q group at the longest input)
                     (d.n_mel + 1 + TACO_MEL_ROWS_WG - 1) / TACO_MEL_ROWS_WG;
  if (groups > n_wg) TACO_FAIL(TACO_EINVAL, "the decoder's producer groups need %d workgroups, the device has %d CUs", groups, n_wg);
  return TACO_OK;
}

#ifdef TACO_STAMPS
static unsigned long long* g_stamps_dev = nullptr;
#endif

int taco_launch_decoder(taco_handle h, const float* memory, const float* pm, int T_in, const int32_t* lens,
                        int b, int launch_idx, int own_len, const uint8_t* masks, int T_cap, int max_steps, float thr, const float* forced_mel,
                        float* mel, float* gate, float* align, int32_t* n_frames, uint8_t* reached,
                        hipStream_t s) {
  const taco_dims& d = h->d;
  const int n_wg = h->n_cu;
  if (T_in < 1 || T_in > TACO_MAX_TIN) TACO_FAIL(TACO_EINVAL, "T_in = %d symbols is outside the decoder kernel's capacity 1..%d", T_in, TACO_MAX_TIN);
  if (max_steps < 1 || max_steps > T_cap) TACO_FAIL(TACO_EINVAL, "max_steps %d outside 1..T_cap=%d", max_steps, T_cap);

  DecArgs a;
  a.P = d.P; a.M = d.M; a.Ra = d.R_a; a.Rd = d.R_d; a.A = d.A; a.F = d.F; a.K = d.K; a.n_mel = d.n_mel;
  a.T_in = T_in; a.lens = lens; a.b = b; a.n_wg = n_wg;
  const taco_weights& w = h->w;
  a.prenet_w1 = w.prenet_w1; a.prenet_w2 = w.prenet_w2; a.wf_h = h->wf_h; a.wf_b = h->wf_b;
  a.att_w_ih = w.att_w_ih; a.att_w_hh = w.att_w_hh; a.att_b_ih = w.att_b_ih; a.att_b_hh = w.att_b_hh;
  a.query_w = w.query_w; a.v_w = w.v_w; a.loc_fold = h->loc_fold;
  a.dec_w_ih = w.dec_w_ih; a.dec_w_hh = w.dec_w_hh; a.dec_b_ih = w.dec_b_ih; a.dec_b_hh = w.dec_b_hh;
  a.proj_w = w.proj_w; a.proj_b = w.proj_b; a.gate_w = w.gate_w; a.gate_b = w.gate_b;
  a.pm = pm; a.G = h->ws_G; a.masks = masks; a.T_cap = T_cap; a.max_steps = max_steps; a.thr = thr;
  a.shift_energies = h->energy_bound > 64.0f ? 1 : 0;
  a.forced_mel = forced_mel; a.mel = mel; a.gate = gate; a.align = align; a.n_frames_out = n_frames; a.reached_out = reached;
  a.gran = h->dec_gran; a.lay = h->dec_lay; a.ctl = h->dec_ctl + launch_idx;
  a.frame_clock = launch_idx == 0 ? h->frame_clock : nullptr;
  a.frame_clock_cap = h->frame_clock_cap;

  // folded context weights of this utterance: G = ctx_w_cat [rows][M] * memory^T [M][T_in], run as a
  // kernel-size-1 convolution over the position-major memory (columns >= len come out as zero)
  const int rc = taco_launch_conv(memory, h->ws_G, h->ctx_w_cat, nullptr, nullptr, nullptr, nullptr,
                                  lens ? lens + b : nullptr, 1, d.M, h->ctx_rows, 1, T_in, 0, /*in_time_major*/ 1, h->conv_split, TACO_CONV_SPLIT_CAP, h->n_cu, s);
  if (rc != TACO_OK) return rc;
  // ... and prenet layer 1 applied to the projection's folded rows: [P][T_in] = W1 [P][n_mel] * G_proj [n_mel][T_in]
  {
    float* g_proj = h->ws_G + (size_t)(4 * d.R_a + 4 * d.R_d) * T_in;
    const int rc2 = taco_launch_conv(g_proj, h->ws_G + (size_t)h->ctx_rows * T_in, h->w.prenet_w1, nullptr, nullptr, nullptr,
                                     nullptr, lens ? lens + b : nullptr, 1, d.n_mel, d.P, 1, T_in, 0, 0, h->conv_split, TACO_CONV_SPLIT_CAP, h->n_cu, s);
    if (rc2 != TACO_OK) return rc2;
  }

  const size_t gran_bytes = (size_t)h->dec_lay.per_parity * 2 * sizeof(taco_granule);
  TACO_HIP(hipMemsetAsync(h->dec_gran, 0, gran_bytes, s));

#ifdef TACO_STAMPS
  const size_t stamp_count = (size_t)(4 + TACO_NQ_WG) * TACO_STAMP_FRAMES * TACO_STAMP_POINTS;
  if (!g_stamps_dev) TACO_HIP(hipMalloc((void**)&g_stamps_dev, stamp_count * 8));
  TACO_HIP(hipMemsetAsync(g_stamps_dev, 0, stamp_count * 8, s));
  a.stamps = g_stamps_dev;
#endif
  if (launch_idx == 0) TACO_HIP(hipEventRecord(h->ev_start, s));
  const bool std_geom = d.P == 256 && d.R_a == 1024 && d.R_d == 1024 && d.A == 128 && d.F == 32 && d.K == 31 &&
                        d.n_mel == 80 && n_wg == 256;
  // own_len = this utterance's number of symbols as the host knows it (T_in when it does not): an utterance that
  // fits the first folded-context chunk runs the one-chunk instance whatever the batch's padded width
  const bool one_chunk = own_len <= TACO_CHUNK;
  auto kern = std_geom ? (one_chunk ? taco_decoder_kernel<true, 1> : taco_decoder_kernel<true, 2>)
                       : (one_chunk ? taco_decoder_kernel<false, 1> : taco_decoder_kernel<false, 2>);
  // beyond the positions whose folded context columns fit on chip: the instance that reads the rest from global memory
  if (own_len > TACO_FOLD_TIN) kern = std_geom ? taco_decoder_kernel<true, 1, true> : taco_decoder_kernel<false, 1, true>;
  hipLaunchKernelGGL(kern, dim3(n_wg), dim3(NT), taco_decoder_lds_bytes(), s, a);
  TACO_HIP(hipGetLastError());
  return TACO_OK;
}

// After the last utterance of a taco_decode call has been enqueued: one wait for the whole batch, then
// the control blocks (and what a preceding taco_encode on this stream found) are checked.
#ifdef TACO_BT_STAMPS
void taco_decoder_batch_dump_stamps(int n_frames);
#endif
int taco_decoder_collect(taco_handle h, int n_launches, hipStream_t s) {
  const int B = n_launches;
  TACO_HIP(hipEventRecord(h->ev_stop, s));
  h->timing_valid = 1;
  TACO_HIP(hipMemcpyAsync(h->dec_ctl_host, h->dec_ctl, sizeof(taco_dec_ctl) * (size_t)B, hipMemcpyDeviceToHost, s));
  TACO_HIP(hipMemcpyAsync(h->enc_ctl_host, h->enc_ctl, 64, hipMemcpyDeviceToHost, s));
  TACO_HIP(hipStreamSynchronize(s));
  {
    const int erc = taco_check_encoder_flags(h, s);
    if (erc != TACO_OK) return erc;
  }
  int longest = 0;
  for (int b = 0; b < B; ++b) {
    const taco_dec_ctl& c = h->dec_ctl_host[b];
    if (c.abort_code != 0)
      TACO_FAIL(TACO_ETIMEDOUT, "decoder hand-off wait gave up (launch %d of the call, code %u, info 0x%x) after %u frames", b,
                c.abort_code, c.abort_info, c.n_frames);
    if ((int)c.n_frames > longest) longest = (int)c.n_frames;
  }
#ifdef TACO_STAMPS
  if (g_stamps_dev) {
    const size_t stamp_count = (size_t)(4 + TACO_NQ_WG) * TACO_STAMP_FRAMES * TACO_STAMP_POINTS;
    std::vector<unsigned long long> st(stamp_count);
    TACO_HIP(hipMemcpy(st.data(), g_stamps_dev, stamp_count * 8, hipMemcpyDeviceToHost));
    const int n_last = (int)h->dec_ctl_host[B - 1].n_frames;
    const int nfr = n_last < TACO_STAMP_FRAMES ? n_last : TACO_STAMP_FRAMES;
    static const char* names[TACO_STAMP_POINTS] = {"frame top done", "S1 + S2 (P group)", "gather p", "S3 attn LSTM",
        "gather ah", "S4 q + energy partials (Q group)", "gather energy partials", "S6 softmax",
        "S7 dec LSTM", "proj half / location term", "gather dh", "S8 proj (M group)"};
    for (int wg = 0; wg < 3 && nfr > 8; ++wg) {
      double acc[TACO_STAMP_POINTS + 1] = {0};
      int cnt = 0;
      for (int t = 4; t + 1 < nfr; ++t, ++cnt) {
        const unsigned long long* r = &st[((size_t)wg * TACO_STAMP_FRAMES + t) * TACO_STAMP_POINTS];
        const unsigned long long* nx = &st[((size_t)wg * TACO_STAMP_FRAMES + t + 1) * TACO_STAMP_POINTS];
        for (int k = 1; k < TACO_STAMP_POINTS; ++k) acc[k] += (double)(r[k] - r[k - 1]) * 0.01;
        acc[TACO_STAMP_POINTS] += (double)(nx[0] - r[TACO_STAMP_POINTS - 1]) * 0.01;
      }
      fprintf(stderr, "[taco stamps] workgroup %s, mean over %d frames (us):\n", wg == 0 ? "0 (P group)" : wg == 1 ? "32 (Q group)" : "200", cnt);
      double tot = 0;
      for (int k = 1; k < TACO_STAMP_POINTS; ++k) { fprintf(stderr, "  %-32s %7.2f\n", names[k], acc[k] / cnt); tot += acc[k] / cnt; }
      fprintf(stderr, "  %-32s %7.2f\n  total %7.2f\n", "attn partial + frame top (mel / stop)", acc[TACO_STAMP_POINTS] / cnt, tot + acc[TACO_STAMP_POINTS] / cnt);
    }
    if (nfr > 8) {
      // the same stamps on one time axis: mean offset of every point from workgroup 0's frame top of that frame
      fprintf(stderr, "[taco stamps] offsets from workgroup 0's frame top (us): point, workgroups 0 / 32 / 200\n");
      for (int k = 0; k < TACO_STAMP_POINTS; ++k) {
        double off[3] = {0, 0, 0};
        int cnt = 0;
        for (int t = 4; t + 1 < nfr; ++t, ++cnt) {
          const unsigned long long base = st[((size_t)0 * TACO_STAMP_FRAMES + t) * TACO_STAMP_POINTS];
          for (int wg = 0; wg < 3; ++wg)
            off[wg] += ((double)st[((size_t)wg * TACO_STAMP_FRAMES + t) * TACO_STAMP_POINTS + k] - (double)base) * 0.01;
        }
        fprintf(stderr, "  %-34s %7.2f %7.2f %7.2f\n", names[k], off[0] / cnt, off[1] / cnt, off[2] / cnt);
      }
    }
    if (nfr > 8) {
      static const char* qn[6] = {"", "query rows", "energy terms", "barrier", "sum + publish", "barrier"};
      double acc[6] = {0};
      int cnt = 0;
      for (int t = 4; t + 1 < nfr; ++t, ++cnt) {
        const unsigned long long* r = &st[((size_t)3 * TACO_STAMP_FRAMES + t) * TACO_STAMP_POINTS];
        for (int k = 1; k < 6; ++k) acc[k] += (double)(r[k] - r[k - 1]) * 0.01;
      }
      {
        double pa[3] = {0, 0, 0};
        int pc = 0;
        for (int t = 4; t + 1 < nfr; ++t, ++pc) {
          const unsigned long long* r = &st[((size_t)3 * TACO_STAMP_FRAMES + t) * TACO_STAMP_POINTS + 6];
          const unsigned long long top = st[((size_t)0 * TACO_STAMP_FRAMES + t) * TACO_STAMP_POINTS];
          pa[0] += (double)(r[1] - r[0]) * 0.01;
          pa[1] += (double)(r[2] - r[1]) * 0.01;
          pa[2] += (double)(top - r[2]) * 0.01;
        }
        fprintf(stderr, "[taco stamps] workgroup 0 before its frame top: layer-1 gather %.2f, layer 2 + publish %.2f, stop test %.2f\n",
                pa[0] / pc, pa[1] / pc, pa[2] / pc);
      }
      fprintf(stderr, "[taco stamps] every Q workgroup, offsets from workgroup 0's frame top: hidden state arrived / partials published\n ");
      for (int q = 0; q < TACO_NQ_WG; ++q) {
        double o0 = 0, o1 = 0;
        int c2 = 0;
        for (int t = 4; t + 1 < nfr; ++t, ++c2) {
          const unsigned long long base = st[((size_t)0 * TACO_STAMP_FRAMES + t) * TACO_STAMP_POINTS];
          const unsigned long long* r = &st[((size_t)(4 + q) * TACO_STAMP_FRAMES + t) * TACO_STAMP_POINTS];
          o0 += ((double)r[0] - (double)base) * 0.01;
          o1 += ((double)r[1] - (double)base) * 0.01;
        }
        fprintf(stderr, " %.2f/%.2f", o0 / c2, o1 / c2);
      }
      fprintf(stderr, "\n");
      fprintf(stderr, "[taco stamps] inside the Q group's stage (workgroup 32):\n");
      for (int k = 1; k < 6; ++k) fprintf(stderr, "  %-32s %7.2f\n", qn[k], acc[k] / cnt);
    }
  }
#endif
#ifdef TACO_BT_STAMPS
  taco_decoder_batch_dump_stamps(longest);
#endif
  h->last_frames = longest;
  return TACO_OK;
}

void taco_decoder_fill_layout(taco_dec_layout* lay, int n_wg) { fill_layout(*lay, n_wg); }

int taco_decoder_prepare(taco_handle h) {
  // the kernel needs more than the default 64 KiB of dynamic LDS
  const void* kernels[6] = {reinterpret_cast<const void*>(taco_decoder_kernel<true, 1>), reinterpret_cast<const void*>(taco_decoder_kernel<true, 2>),
                            reinterpret_cast<const void*>(taco_decoder_kernel<false, 1>), reinterpret_cast<const void*>(taco_decoder_kernel<false, 2>),
                            reinterpret_cast<const void*>(taco_decoder_kernel<true, 1, true>), reinterpret_cast<const void*>(taco_decoder_kernel<false, 1, true>)};
  for (const void* k : kernels) {
    // the kernel needs more than the default 64 KiB of dynamic LDS; one workgroup must fit on a CU
    TACO_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)taco_decoder_lds_bytes()));
    int per_cu = 0;
    TACO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, NT, taco_decoder_lds_bytes()));
    if (per_cu < 1) TACO_FAIL(TACO_EINVAL, "decoder kernel does not fit on a CU (occupancy query returned %d)", per_cu);
  }
  // |energy| <= sum_a |v_a| (each term is v_a * tanh(.)): up to 64 the softmax needs no shift by the maximum
  // (exp stays within 1.6e-28 .. 6.3e27, normal numbers, and a sum over 512 positions is far from overflow).
  // TACO_SOFTMAX_SHIFT=1 in the environment forces the shifted form (for timing it).
  std::vector<float> v((size_t)h->d.A);
  TACO_HIP(hipMemcpy(v.data(), h->w.v_w, sizeof(float) * v.size(), hipMemcpyDeviceToHost));
  double bound = 0.0;
  for (float x : v) bound += fabs((double)x);
  h->energy_bound = bound == bound ? (float)bound : INFINITY;     // NaN weights: take the guarded path
  const char* force = getenv("TACO_SOFTMAX_SHIFT");
  if (force && force[0] == '1') h->energy_bound = INFINITY;
  return TACO_OK;
}
