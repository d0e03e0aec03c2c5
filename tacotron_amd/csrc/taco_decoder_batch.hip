// Lockstep decoder for gfx950 (MI355X): up to 16 utterances advance one frame at a time inside ONE
// persistent launch, with the LSTM gate products on the matrix cores.
//
// Semantics per utterance are those of the reference's Decoder.inference loop at batch size 1
// (reference: src/tacotron/model.py:529-568, 136-141, 454-491, 96-117); the reference itself cannot
// batch (its stop test raises for more than one utterance, model.py:555), so a batch is defined as
// independent utterances that happen to share a launch: per-utterance lengths, attention over each
// utterance's own positions only, one stop flag each, and the same prenet keep-masks for all (the
// caller seeds every line alike, reference: src/tacotron_cli/inference.py:137-141, 241).
//
// Why lockstep: at batch 1 a frame is a chain of cross-CU hand-offs of about a microsecond each with
// little work in between, and the matrix cores idle.  The hand-offs cost the same whether they carry
// one utterance or sixteen, and sixteen utterances turn every weight row's dot product into a column
// of a 16 x 16 tile: the LSTM gate rows become v_mfma_f32_16x16x4_f32 products (exact f32).
//
// Layout of the work (one workgroup of 512 threads per CU, resident for the whole launch):
//   * every workgroup j owns hidden units 4j..4j+3 of both LSTM cells = 16 gate rows each.  The rows
//     stay in registers as MFMA A operands for the whole launch, split over the 8 waves by columns
//     (wave w holds a 1/8 slice of every row); a wave's partial 16 x 16 tile meets the other waves'
//     in LDS, wave 0 adds them, applies the cell update for its 4 units x 16 utterances (one per lane)
//     and publishes the hidden values;
//   * activations travel as 8-byte {epoch, fp32} granules, utterance-major ([utterance][k]), and are
//     loaded straight into MFMA B operands: lane (utterance u = l & 15, q = l >> 4) fetches 16 bytes =
//     the two granules k = 8 p + 2 q, +1 of pair p, which are its operands of two consecutive 4-deep
//     steps (the weights are held in the same column order);
//   * the attention context is computed explicitly (the batch-1 kernel folds it into per-utterance
//     weight rows, which a batch cannot share): 8 "Q" workgroups per utterance each hold 16 attention
//     dimensions (query rows, location filter, processed memory) and an eighth of the memory columns;
//     they exchange partial energies among themselves, each runs the softmax, and each publishes its
//     columns of the context;
//   * prenet layer 1 is folded into the mel projection, W1 (Wp [dh ; ctx] + b) = (W1 Wp) [dh ; ctx] + W1 b,
//     so the "P" workgroups (16 prenet rows each) get the next frame's layer-1 rows from the decoder
//     hidden state and the context without waiting for a materialised frame; the "M" workgroups
//     (16 projection / stop-token rows each) write the frame beside the dependency chain.  Both read
//     their rows from LDS as MFMA A operands and share the B operands of the LSTM loads.
// A frame's chain: dh, ctx -> h1 (P) -> p (P) -> attention LSTM (all) -> ah -> energies (Q) -> partial
// energies (Q to Q) -> softmax, context (Q) -> ctx -> decoder LSTM (all) -> dh.
// An utterance that has stopped keeps its slot (its values stay finite and are no longer written out);
// the launch ends when every utterance has stopped or max_steps frames have been produced.
#include "taco_device.h"

#include <math.h>
#include <vector>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = TACO_DEC_THREADS;
constexpr int NW = NT / 64;
constexpr int NB = TACO_BATCH_NB;
static_assert(NB == 16, "one utterance per column of a 16 x 16 matrix-core tile");
// k capacities of the exchanged vectors: multiples of 64 = 8 waves x (pairs of 4-deep steps)
constexpr int KP = TACO_BT_KP, KH = TACO_BT_KH, KC = TACO_BT_KC;
constexpr int NPW_P = KP / 8 / NW;     // pairs of k-steps per wave: 4
constexpr int NPW_H = KH / 8 / NW;     // 16
constexpr int NPW_C = KC / 8 / NW;     // 9
constexpr int TIN = TACO_BT_TIN;       // positions per utterance (256)
constexpr int HALO = 16;
constexpr int QD = 16;                 // attention dimensions per Q workgroup
constexpr int NQ = TACO_NQ_WG;         // Q workgroups per utterance (8)
constexpr int MEMW = TACO_BT_MEMW;     // memory columns per Q workgroup (68)
constexpr int CPG = 7;                 // position groups of the context sum (7 x 68 threads)
constexpr int LOC_KSTEPS = 16;
constexpr int RP = TACO_BT_ROWS;       // rows per P / M workgroup (one MFMA tile)
constexpr int W1_KSTEPS = (TACO_MAX_MEL + 3) / 4;    // 20
static_assert(2 * TACO_MAX_K <= 4 * LOC_KSTEPS, "location taps fit 16 steps of 4");
static_assert(CPG * MEMW <= NT && NQ * MEMW >= 544, "context sum thread map");

// ---- LDS carve (float offsets) ----
constexpr int L_RED = 0;                         // [8 waves][64 lanes][4]: partial tiles of an LSTM stage
constexpr int L_BIAS = L_RED + NW * 256;         // [2 cells][4 units][4 gates]
constexpr int L_STOP = L_BIAS + 32;              // stop tokens of the previous frame [16], failure flag at +16
constexpr int L_UTT = L_STOP + 32;               // utterance index of every slot [16] (int)
constexpr int L_ROLE = L_UTT + 16;
// P: folded layer-1 rows [400 k-steps][64], layer-2 rows [64][64], plain layer-1 rows [20][64], partial tiles, bias
constexpr int KS_HC = (KH + KC) / 4;             // 400 k-steps of [dh ; ctx]
constexpr int LP_WF = L_ROLE;
constexpr int LP_W2 = LP_WF + KS_HC * 64;
constexpr int LP_W1 = LP_W2 + (KP / 4) * 64;
constexpr int LP_RED = LP_W1 + W1_KSTEPS * 64;
constexpr int LP_BIAS = LP_RED + NW * 256;
constexpr int LP_END = LP_BIAS + RP;
// M: projection rows [400][64], partial tiles, bias
constexpr int LM_W = L_ROLE;
constexpr int LM_RED = LM_W + KS_HC * 64;
constexpr int LM_BIAS = LM_RED + NW * 256;
constexpr int LM_END = LM_BIAS + RP;
// Q: query rows [16][1024], memory slice [256][68], location filter [16][64], hidden state of my utterance /
//    gathered partial energies / context partial sums (one area, used one after the other), weights and
//    cumulative weights with halos, query values, unnormalised weights, softmax scratch
constexpr int LQ_ROWS = L_ROLE;
constexpr int LQ_MEM = LQ_ROWS + QD * KH;
constexpr int LQ_E = LQ_MEM + TIN * MEMW;
constexpr int LQ_XPE = LQ_E + LOC_KSTEPS * 64;
constexpr int LQ_AW = LQ_XPE + NQ * TIN;
constexpr int LQ_CUM = LQ_AW + TIN + 2 * HALO;
constexpr int LQ_Q = LQ_CUM + TIN + 2 * HALO;
constexpr int LQ_X = LQ_Q + QD;
constexpr int LQ_SRED = LQ_X + TIN;
constexpr int LQ_END = LQ_SRED + 32;
static_assert(NQ * TIN >= KH && NQ * TIN >= CPG * MEMW, "shared Q scratch area");
constexpr int L_TOTAL = LP_END > LQ_END ? (LP_END > LM_END ? LP_END : LM_END) : (LQ_END > LM_END ? LQ_END : LM_END);
static_assert(L_TOTAL * 4 <= 160 * 1024, "batched decoder LDS carve exceeds a CU's 160 KiB");

struct BatArgs {
  int P, M, Ra, Rd, A, K, n_mel, T_in, n_wg;
  int n_utts;
  int utt[NB];             // utterance index (into lens, memory, outputs) of every slot
  const int32_t* lens;     // device, or NULL = T_in symbols each
  const float *prenet_w1, *prenet_w2, *wf_h, *wf_c, *wf_b, *att_w_ih, *att_w_hh, *att_b_ih, *att_b_hh, *query_w, *v_w,
      *loc_fold, *dec_w_ih, *dec_w_hh, *dec_b_ih, *dec_b_hh, *proj_w, *proj_b, *gate_w, *gate_b;
  const float* memory;     // [B][T_in][M]
  const float* pm;         // [B][T_in][A]
  const uint8_t* masks;
  int T_cap, max_steps;
  float thr;
  int shift_energies;
  const float* forced_mel; // [B][n_mel][T_cap] or NULL
  float *mel, *gate, *align;
  int32_t* n_frames;
  uint8_t* reached;
  taco_granule* gran;
  unsigned gran_bytes;
  taco_bat_layout lay;
  taco_dec_ctl* ctl;
  unsigned long long* frame_clock;
  int frame_clock_cap;
};

// The two granules {payload, epoch} x 2 of `NPAIR` pairs into B operands.  k0 = the k of this lane's first granule;
// granules at k >= klim (beyond the vector's true length) and lanes whose utterance slot is empty take zeros
// without looking at the tags.  A tag that is not there yet is waited for on its own.
template <int NPAIR>
__device__ __forceinline__ void bt_load_pairs(__amdgpu_buffer_rsrc_t rs, const taco_granule* gran, unsigned gidx,
                                              unsigned ep, bool live, int k0, int klim, float (&b)[2 * NPAIR],
                                              taco_wait_ctx& ctx, unsigned info) {
  u32x4 v[NPAIR];
#pragma unroll
  for (int i = 0; i < NPAIR; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (gidx + 8u * i) * 8u, 0, 16 /* sc1 */);
#pragma unroll
  for (int i = 0; i < NPAIR; ++i) {
    float x0 = __uint_as_float(v[i].x), x1 = __uint_as_float(v[i].z);
    const bool need0 = live && k0 + 8 * i < klim, need1 = live && k0 + 8 * i + 1 < klim;
    if (need0 && v[i].y != ep) { ctx.spun = 1; x0 = taco_wait_one(gran, (int)(gidx + 8u * i), ep, ctx, info); }
    if (need1 && v[i].w != ep) { ctx.spun = 1; x1 = taco_wait_one(gran, (int)(gidx + 8u * i + 1u), ep, ctx, info); }
    b[2 * i] = need0 ? x0 : 0.0f;
    b[2 * i + 1] = need1 ? x1 : 0.0f;
  }
}

template <int N>
__device__ __forceinline__ f32x4 bt_mfma_regs(const float (&w)[N], const float (&b)[N], f32x4 acc) {
#pragma unroll
  for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i], b[i], acc, 0, 0, 0);
  return acc;
}
// A operands from LDS ([k-step][lane] floats)
template <int N>
__device__ __forceinline__ f32x4 bt_mfma_lds(const float* w_lds, int lane, const float (&b)[N], f32x4 acc) {
#pragma unroll
  for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w_lds[i * 64 + lane], b[i], acc, 0, 0, 0);
  return acc;
}

// The 8 waves' partial tiles -> LDS; after the barrier wave 0 holds the sum (other waves get their own partial back).
__device__ __forceinline__ f32x4 bt_reduce_tiles(float* red, f32x4 acc, int wave, int lane) {
  *reinterpret_cast<f32x4*>(red + (wave * 64 + lane) * 4) = acc;
  taco_lds_barrier();
  if (wave == 0) {
#pragma unroll
    for (int w = 1; w < NW; ++w) acc += *reinterpret_cast<const f32x4*>(red + (w * 64 + lane) * 4);
  }
  return acc;
}

__device__ __forceinline__ float bt_lstm_cell(f32x4 pre, f32x4 bias, float& c) {
  const float ig = taco_sigmoid(pre[0] + bias[0]);
  const float fg = taco_sigmoid(pre[1] + bias[1]);
  const float gg = taco_tanh(pre[2] + bias[2]);
  const float og = taco_sigmoid(pre[3] + bias[3]);
  c = fg * c + ig * gg;
  return og * taco_tanh(c);
}

template <int R>
__device__ __forceinline__ void bt_publish(taco_granule* base, int cap, int idx, unsigned epoch, float value) {
  const taco_granule word = ((taco_granule)epoch << 32) | (taco_granule)__float_as_uint(value);
#pragma unroll
  for (int r = 0; r < R; ++r) taco_send((taco_ggran*)base + (size_t)r * cap + idx, word);
}

__device__ __forceinline__ void bt_nap_begin(int nap, taco_wait_ctx& ctx) {
  taco_nap(nap);
  ctx.spun = 0;
  asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(TACO_DEC_THREADS) void taco_decoder_batch_kernel(const BatArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = blockIdx.x;
  const int P = a.P, M = a.M, Ra = a.Ra, Rd = a.Rd, A = a.A, K = a.K, NMEL = a.n_mel;
  const int utt = lane & 15, kq = lane >> 4;      // B operand / result column = utterance slot; k offset / unit
  const bool utt_live = utt < a.n_utts;
  constexpr int REP = TACO_BT_REP;
  const int rep = j % REP;

  for (int i = tid; i < L_TOTAL; i += NT) lds[i] = 0.0f;
  __syncthreads();

  // ---------------- roles ----------------
  const int n_p_wg = (P + RP - 1) / RP;
  const int n_m_wg = (NMEL + 1 + RP - 1) / RP;
  const int qd = (A + NQ - 1) / NQ;                       // attention dimensions per Q workgroup (<= 16)
  const int n_dg = (A + qd - 1) / qd;                     // dimension groups in use (<= 8)
  const int q_off = n_p_wg + n_m_wg;
  const bool is_p = j < n_p_wg;
  const bool is_m = j >= n_p_wg && j < q_off;
  const int q_id = j - q_off;
  const bool is_q = q_id >= 0 && q_id < a.n_utts * NQ && (q_id % NQ) < n_dg;
  const int q_slot = is_q ? q_id / NQ : 0;
  const int dg = is_q ? q_id % NQ : 0;
  const int q_b = a.utt[q_slot];                          // my utterance (Q)
  const int len = is_q ? (a.lens ? min(a.lens[q_b], a.T_in) : a.T_in) : 0;
  const int len4 = (len + 3) & ~3;
  const int q_base = dg * qd;
  const int q_cnt = is_q ? min(qd, A - q_base) : 0;
  const int cw = (M + NQ - 1) / NQ;                       // memory columns per Q workgroup (<= MEMW)
  const int c0 = dg * cw;
  const int c_cnt = is_q ? max(0, min(cw, M - c0)) : 0;
  const bool forced = a.forced_mel != nullptr;
  const int pad = (K - 1) / 2;

  // ---------------- one-time loads: LSTM rows as A operands in registers ----------------
  // A operand of lane l: row (l & 15) = 4 * unit + gate of the workgroup's 16 rows, column k of step i:
  // k = 8 * (pair) + 2 * (l >> 4) + (i & 1), pair = wave * pairs_per_wave + (i >> 1)
  const int row16 = lane & 15;
  const int hid_a = j * 4 + (row16 >> 2), gate_a = row16 & 3;
  const bool own_a = hid_a < Ra, own_d = hid_a < Rd;
  const size_t ra = (size_t)gate_a * Ra + hid_a, rd = (size_t)gate_a * Rd + hid_a;
  float wa_p[2 * NPW_P], wa_c[2 * NPW_C], wa_h[2 * NPW_H];
  float wd_a[2 * NPW_H], wd_c[2 * NPW_C], wd_h[2 * NPW_H];
#define BT_K(NPW, i) (8 * (wave * (NPW) + ((i) >> 1)) + 2 * kq + ((i) & 1))
#pragma unroll
  for (int i = 0; i < 2 * NPW_P; ++i) {
    const int k = BT_K(NPW_P, i);
    wa_p[i] = (own_a && k < P) ? a.att_w_ih[ra * (P + M) + k] : 0.0f;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 2 * NPW_C; ++i) {
    const int k = BT_K(NPW_C, i);
    wa_c[i] = (own_a && k < M) ? a.att_w_ih[ra * (P + M) + P + k] : 0.0f;
    wd_c[i] = (own_d && k < M) ? a.dec_w_ih[rd * (Ra + M) + Ra + k] : 0.0f;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 2 * NPW_H; ++i) {
    const int k = BT_K(NPW_H, i);
    wa_h[i] = (own_a && k < Ra) ? a.att_w_hh[ra * Ra + k] : 0.0f;
    wd_a[i] = (own_d && k < Ra) ? a.dec_w_ih[rd * (Ra + M) + k] : 0.0f;
    wd_h[i] = (own_d && k < Rd) ? a.dec_w_hh[rd * Rd + k] : 0.0f;
    if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
  }
  if (tid < NB) reinterpret_cast<int*>(lds + L_UTT)[tid] = a.utt[tid];
  if (tid < 16) {     // biases as [unit][gate]
    const int u = tid >> 2, gt = tid & 3, hu = j * 4 + u;
    lds[L_BIAS + tid] = hu < Ra ? a.att_b_ih[gt * Ra + hu] + a.att_b_hh[gt * Ra + hu] : 0.0f;
    lds[L_BIAS + 16 + tid] = hu < Rd ? a.dec_b_ih[gt * Rd + hu] + a.dec_b_hh[gt * Rd + hu] : 0.0f;
  }

  // ---------------- one-time loads of the roles ----------------
  // A operands kept in LDS as [global k-step][lane]: k-step ks of a vector's wave slice, lane l -> row l & 15,
  // column k as above.  Global k-steps of [dh ; ctx]: dh 0..255 (wave w: 32 w ..), ctx 256..399 (wave w: 256 + 18 w ..).
  const int n_hc = Rd + M;
  if (is_p || is_m) {
    float* dst = lds + (is_p ? LP_WF : LM_W);
    const int n_rows = is_p ? P : NMEL + 1;
    _Pragma("unroll 1") for (int e = tid; e < KS_HC * 64; e += NT) {
      const int ks = e >> 6, l = e & 63;
      const int r = (is_p ? j : j - n_p_wg) * RP + (l & 15), q4 = l >> 4;
      float v = 0.0f;
      if (r < n_rows) {
        if (ks < KH / 4) {
          const int k = 8 * (ks >> 1) + 2 * q4 + (ks & 1);
          if (k < Rd) v = is_p ? a.wf_h[(size_t)r * Rd + k] : (r < NMEL ? a.proj_w[(size_t)r * n_hc + k] : a.gate_w[k]);
        } else {
          const int kc = ks - KH / 4;
          const int k = 8 * (kc >> 1) + 2 * q4 + (kc & 1);
          if (k < M) v = is_p ? a.wf_c[(size_t)r * M + k] : (r < NMEL ? a.proj_w[(size_t)r * n_hc + Rd + k] : a.gate_w[Rd + k]);
        }
      }
      dst[e] = v;
    }
    if (tid < RP) {
      const int r = (is_p ? j : j - n_p_wg) * RP + tid;
      if (is_p) lds[LP_BIAS + tid] = r < P ? a.wf_b[r] : 0.0f;
      else lds[LM_BIAS + tid] = r < NMEL ? a.proj_b[r] : (r == NMEL ? a.gate_b[0] : 0.0f);
    }
  }
  if (is_p) {
    _Pragma("unroll 1") for (int e = tid; e < (KP / 4) * 64; e += NT) {
      const int ks = e >> 6, l = e & 63;
      const int r = j * RP + (l & 15), k = 8 * (ks >> 1) + 2 * (l >> 4) + (ks & 1);
      lds[LP_W2 + e] = (r < P && k < P) ? a.prenet_w2[(size_t)r * P + k] : 0.0f;
    }
    _Pragma("unroll 1") for (int e = tid; e < W1_KSTEPS * 64; e += NT) {
      const int ks = e >> 6, l = e & 63;
      const int r = j * RP + (l & 15), k = 8 * (ks >> 1) + 2 * (l >> 4) + (ks & 1);
      lds[LP_W1 + e] = (r < P && k < NMEL) ? a.prenet_w1[(size_t)r * NMEL + k] : 0.0f;
    }
  }
  // Q: processed memory of my dimensions for my tiles (wave w: tiles w and w + 8 of 16 positions), in registers;
  //    lane (col = l & 15, kq): position 16 * tile + col, dimensions q_base + 4 kq .. + 3
  float pmv[2][4] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
  float vq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (is_q) {
    const float* mem_b = a.memory + (size_t)q_b * a.T_in * M;
    const float* pm_b = a.pm + (size_t)q_b * a.T_in * A;
    _Pragma("unroll 1") for (int e = tid; e < q_cnt * Ra; e += NT) {
      const int r = e / Ra, c = e % Ra;
      lds[LQ_ROWS + r * KH + c] = a.query_w[(size_t)(q_base + r) * Ra + c];
    }
    _Pragma("unroll 1") for (int e = tid; e < len * MEMW; e += NT) {
      const int s = e / MEMW, cc = e % MEMW;
      lds[LQ_MEM + e] = cc < c_cnt ? mem_b[(size_t)s * M + c0 + cc] : 0.0f;
    }
    _Pragma("unroll 1") for (int i = tid; i < LOC_KSTEPS * 64; i += NT) {
      const int m = i >> 6, l = i & 63;
      const int ck = 4 * m + (l >> 4), dim = q_base + (l & 15);
      lds[LQ_E + i] = (ck < 2 * K && (l & 15) < q_cnt) ? a.loc_fold[(size_t)dim * 2 * K + ck] : 0.0f;
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      const int s = 16 * (wave + NW * ti) + utt;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (s < len && 4 * kq + r < q_cnt) pmv[ti][r] = pm_b[(size_t)s * A + q_base + 4 * kq + r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * kq + r < q_cnt) vq[r] = a.v_w[q_base + 4 * kq + r];
  }
  __syncthreads();

  taco_wait_ctx wctx;
  wctx.abort_word = (taco_gu32*)&a.ctl->abort_code;
  wctx.failed = 0;
  wctx.spun = 0;
  int nap_h1 = 0, nap_p = 0, nap_ahq = 0, nap_ah = 0, nap_pe = 0, nap_ctx = 0, nap_dh = 0;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.gran, 0, (int)a.gran_bytes, 0x00020000);

  // location term of my two tiles: U = processed memory + sum over taps of E * {previous, cumulative} weights
  float U[2][4];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int r = 0; r < 4; ++r) U[ti][r] = pmv[ti][r];
  auto update_location_term = [&]() {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      const int s0 = 16 * (wave + NW * ti);
      if (s0 < len) {
        f32x4 acc = {pmv[ti][0], pmv[ti][1], pmv[ti][2], pmv[ti][3]};
#pragma unroll 4
        for (int m = 0; m < LOC_KSTEPS; ++m) {
          int ck = 4 * m + kq;
          ck = ck < 2 * K ? ck : 0;
          const int c = ck >= K ? 1 : 0;
          const float xv = lds[(c ? LQ_CUM : LQ_AW) + HALO + s0 + utt + (ck - c * K) - pad];
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[LQ_E + m * 64 + lane], xv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) U[ti][r] = acc[r];
      }
    }
  };

  f32x4 acc_a = {0.0f, 0.0f, 0.0f, 0.0f};     // attention-LSTM partial over [context ; hidden] of the previous frame
  f32x4 acc_dn = {0.0f, 0.0f, 0.0f, 0.0f};    // decoder-LSTM partial over the previous decoder hidden state
  float c_a = 0.0f, c_d = 0.0f;               // cell states (wave 0: lane = (unit kq, utterance utt))
  unsigned alive = a.n_utts >= 32 ? 0xffffffffu : ((1u << a.n_utts) - 1u);
  int my_frames = 0;                           // thread tid < 16 of every workgroup: frames of utterance slot tid
  int t = 0;
  // byte offsets (in granules) of this lane's first pair in each vector, replica `rep`
  const int k0_p = 8 * (wave * NPW_P) + 2 * kq, k0_h = 8 * (wave * NPW_H) + 2 * kq, k0_c = 8 * (wave * NPW_C) + 2 * kq;
  const unsigned g_lane_p = (unsigned)(utt * KP + k0_p);
  const unsigned g_lane_h = (unsigned)(utt * KH + k0_h);
  const unsigned g_lane_c = (unsigned)(utt * KC + k0_c);
  unsigned keep_next = 0;                      // P, wave 0: keep bytes of my 4 rows, layer 0 (low) / layer 1 of the coming frame
  unsigned keep_l1 = 0, keep_l2 = 0;
  if (is_p && wave == 0) {
    const int r0 = j * RP + 4 * kq;
    if (r0 < P) {
      keep_l2 = *reinterpret_cast<const unsigned*>(a.masks + P + r0);              // frame 0, layer 2
    }
  }

  for (;; ++t) {
    const unsigned ep = (unsigned)t + 1u;
    const unsigned par = (unsigned)(t & 1) * (unsigned)a.lay.per_parity;
    taco_granule* g = a.gran + par;

    // ============ P: prenet layer 2 of this frame (heads the chain; does not wait for the stop test) ============
    if (is_p && t < a.max_steps) {
      f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
      // keep bytes of the next frame, asked for a frame ahead (a first-touch line comes from HBM)
      unsigned kn1 = 0, kn2 = 0;
      if (wave == 0) {
        const int r0 = min(j * RP + 4 * kq, P - 4);
        const uint8_t* src = a.masks + (size_t)min(t + 1, a.max_steps - 1) * 2 * P + r0;
        kn1 = *reinterpret_cast<const unsigned*>(src);
        kn2 = *reinterpret_cast<const unsigned*>(src + P);
      }
      if (t > 0) {
        float b[2 * NPW_P];
        bt_nap_begin(nap_h1, wctx);
        bt_load_pairs<NPW_P>(rs, a.gran, par + a.lay.h1 + rep * a.lay.h1_cap + g_lane_p, ep, utt_live, k0_p, P, b, wctx, 0x300u);
        nap_h1 = taco_nap_adjust(nap_h1, wctx.spun);
        acc = bt_mfma_lds<2 * NPW_P>(lds + LP_W2 + wave * (2 * NPW_P) * 64, lane, b, acc);
      }
      acc = bt_reduce_tiles(lds + LP_RED, acc, wave, lane);
      if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = j * RP + 4 * kq + r;
          float v = acc[r] > 0.0f ? acc[r] : 0.0f;
          v = ((keep_l2 >> (8 * r)) & 0xffu) ? v * 2.0f : 0.0f;
          if (row < P) bt_publish<REP>(g + a.lay.p, a.lay.p_cap, utt * KP + row, ep, v);
        }
        keep_l1 = kn1;
        keep_next = kn2;
      }
      taco_lds_barrier();      // LP_RED is reused by the layer-1 rows below
    }

    // ============ frame boundary: the previous frame's stop tokens ============
    if (t > 0) {
      if (wave == 0 && lane < NB) lds[L_STOP + lane] = taco_wait_one(a.gran + (((t - 1) & 1) ? a.lay.per_parity : 0) + a.lay.stop + rep * a.lay.stop_cap, lane, (unsigned)t, wctx, 0x100u);
      if (wctx.failed) lds[L_STOP + 16] = 1.0f;
      taco_lds_barrier();
      if (lds[L_STOP + 16] != 0.0f) break;
      const bool fired = !forced && taco_sigmoid(lds[L_STOP + utt]) > a.thr;
      const unsigned fmask = (unsigned)__ballot(fired) & 0xffffu;
      const unsigned ended = alive & fmask;
      if (tid < NB && ((ended >> tid) & 1u)) my_frames = t;
      alive &= ~fmask;
    }
    if (alive == 0u || t == a.max_steps) break;

    // ============ attention LSTM ============
    {
      float b[2 * NPW_P];
      bt_nap_begin(nap_p, wctx);
      bt_load_pairs<NPW_P>(rs, a.gran, par + a.lay.p + rep * a.lay.p_cap + g_lane_p, ep, utt_live, k0_p, P, b, wctx, 0x400u);
      nap_p = taco_nap_adjust(nap_p, wctx.spun);
      acc_a = bt_mfma_regs<2 * NPW_P>(wa_p, b, acc_a);
      const f32x4 pre = bt_reduce_tiles(lds + L_RED, acc_a, wave, lane);
      if (wave == 0) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(lds + L_BIAS + 4 * kq);
        const float hv = bt_lstm_cell(pre, bias, c_a);
        const int hu = j * 4 + kq;
        if (hu < Ra) bt_publish<REP>(g + a.lay.ah, a.lay.ah_cap, utt * KH + hu, ep, hv);
      }
    }

    // ============ Q: query values and partial energies of my 16 dimensions, every position of my utterance ============
    if (is_q) {
      bt_nap_begin(nap_ahq, wctx);
      taco_collect<NT>(g + a.lay.ah + rep * a.lay.ah_cap + q_slot * KH, Ra, ep, lds + LQ_XPE, wctx, 0x500u);
      nap_ahq = taco_nap_adjust(nap_ahq, wctx.spun);
      taco_lds_barrier();
      {
        // wave w: query values of dimensions 2w, 2w + 1
        const float* r0 = lds + LQ_ROWS + (2 * wave) * KH;
        const float* r1 = r0 + KH;
        float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
        for (int c = 0; c < KH / 256; ++c) {
          const float4 x = *reinterpret_cast<const float4*>(lds + LQ_XPE + c * 256 + lane * 4);
          const float4 u0 = *reinterpret_cast<const float4*>(r0 + c * 256 + lane * 4);
          const float4 u1 = *reinterpret_cast<const float4*>(r1 + c * 256 + lane * 4);
          p0 += (fmaf(u0.x, x.x, u0.y * x.y)) + (fmaf(u0.z, x.z, u0.w * x.w));
          p1 += (fmaf(u1.x, x.x, u1.y * x.y)) + (fmaf(u1.z, x.z, u1.w * x.w));
        }
        float s0, s1;
        taco_wave_sum2(p0, p1, s0, s1);
        if (lane == 0) {
          lds[LQ_Q + 2 * wave] = s0;
          lds[LQ_Q + 2 * wave + 1] = s1;
        }
      }
      taco_lds_barrier();
      {
        const f32x4 q4 = *reinterpret_cast<const f32x4*>(lds + LQ_Q + 4 * kq);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
          const int s = 16 * (wave + NW * ti) + utt;
          if (16 * (wave + NW * ti) < len4) {
            float e = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) e = fmaf(vq[r], taco_tanh(q4[r] + U[ti][r]), e);
            e += __shfl_xor(e, 16, 64);
            e += __shfl_xor(e, 32, 64);
            if (kq == 0 && s < len4)
              taco_publish(g + a.lay.pe + (q_slot * NQ + dg) * TIN, s, ep, s < len ? e : 0.0f);
          }
        }
      }
    }

    // ============ all: the attention hidden state feeds the decoder LSTM (this frame) and the attention LSTM (next) ============
    f32x4 acc_d = acc_dn;
    acc_a = {0.0f, 0.0f, 0.0f, 0.0f};
    {
      bt_nap_begin(nap_ah, wctx);
#pragma unroll
      for (int hlf = 0; hlf < 2; ++hlf) {
        float b[NPW_H];
        bt_load_pairs<NPW_H / 2>(rs, a.gran, par + a.lay.ah + rep * a.lay.ah_cap + g_lane_h + 8u * (NPW_H / 2) * hlf, ep, utt_live, k0_h + 8 * (NPW_H / 2) * hlf, Ra, b, wctx, 0x600u);
        if (hlf == 0) nap_ah = taco_nap_adjust(nap_ah, wctx.spun);
#pragma unroll
        for (int i = 0; i < NPW_H; ++i) {
          acc_d = __builtin_amdgcn_mfma_f32_16x16x4f32(wd_a[NPW_H * hlf + i], b[i], acc_d, 0, 0, 0);
          acc_a = __builtin_amdgcn_mfma_f32_16x16x4f32(wa_h[NPW_H * hlf + i], b[i], acc_a, 0, 0, 0);
        }
      }
    }

    // ============ Q: energies of my utterance, softmax, my columns of the context ============
    if (is_q) {
      taco_lds_barrier();                  // the hidden-state copy in LQ_XPE has been read by every wave
      bt_nap_begin(nap_pe, wctx);
      _Pragma("unroll 1") for (int dgi = 0; dgi < n_dg; ++dgi)
        taco_collect<NT>(g + a.lay.pe + (q_slot * NQ + dgi) * TIN, len4, ep, lds + LQ_XPE + dgi * TIN, wctx, 0x700u);
      nap_pe = taco_nap_adjust(nap_pe, wctx.spun);
      taco_lds_barrier();
      float w_own = 0.0f;
      {
        float e = 0.0f;
        if (tid < len)
          for (int dgi = 0; dgi < n_dg; ++dgi) e += lds[LQ_XPE + dgi * TIN + tid];
        float mx = 0.0f;
        if (a.shift_energies) {
          const float m = taco_wave_max(tid < len ? e : -INFINITY);
          if (lane == 0) lds[LQ_SRED + 8 + wave] = m;
          taco_lds_barrier();
          const float4 m0 = *reinterpret_cast<const float4*>(lds + LQ_SRED + 8);
          const float4 m1 = *reinterpret_cast<const float4*>(lds + LQ_SRED + 12);
          mx = fmaxf(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)), fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w)));
        }
        const float x = tid < len ? taco_exp(e - mx) : 0.0f;
        const float part = taco_wave_sum(x);
        if (lane == 0) lds[LQ_SRED + wave] = part;
        taco_lds_barrier();
        const float4 s0 = *reinterpret_cast<const float4*>(lds + LQ_SRED);
        const float4 s1 = *reinterpret_cast<const float4*>(lds + LQ_SRED + 4);
        const float inv = 1.0f / (((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w)));
        w_own = x * inv;
      }
      if (tid < TIN) lds[LQ_AW + HALO + tid] = w_own;
      if (tid < len) {
        lds[LQ_CUM + HALO + tid] += w_own;
        if (dg == 0 && a.align && ((alive >> q_slot) & 1u)) a.align[((size_t)q_b * a.T_cap + t) * a.T_in + tid] = w_own;
      }
      taco_lds_barrier();
      // context columns: thread (column cc, position group pg) sums its positions; the groups meet in LDS
      {
        const int cc = tid % MEMW, pg = tid / MEMW;
        if (pg < CPG) {
          float s = 0.0f;
          for (int sp = pg; sp < len; sp += CPG) s = fmaf(lds[LQ_AW + HALO + sp], lds[LQ_MEM + sp * MEMW + cc], s);
          lds[LQ_XPE + pg * MEMW + cc] = s;
        }
      }
      taco_lds_barrier();
      if (tid < c_cnt) {
        float s = 0.0f;
#pragma unroll
        for (int pg = 0; pg < CPG; ++pg) s += lds[LQ_XPE + pg * MEMW + tid];
        bt_publish<REP>(g + a.lay.ctx, a.lay.ctx_cap, q_slot * KC + c0 + tid, ep, s);
      }
      // next frame's location term, while the context travels
      update_location_term();
    }
    if (a.frame_clock && j == a.n_wg - 1 && tid == 0 && t < a.frame_clock_cap) a.frame_clock[t] = wall_clock64();

    // ============ all: the context finishes the decoder LSTM; P / M start their rows ============
    f32x4 acc_r = {0.0f, 0.0f, 0.0f, 0.0f};     // P: folded layer-1 rows; M: projection rows
    {
      float b[2 * NPW_C];
      bt_nap_begin(nap_ctx, wctx);
      bt_load_pairs<NPW_C>(rs, a.gran, par + a.lay.ctx + rep * a.lay.ctx_cap + g_lane_c, ep, utt_live, k0_c, M, b, wctx, 0x800u);
      nap_ctx = taco_nap_adjust(nap_ctx, wctx.spun);
      acc_d = bt_mfma_regs<2 * NPW_C>(wd_c, b, acc_d);
      const f32x4 pre = bt_reduce_tiles(lds + L_RED, acc_d, wave, lane);
      if (wave == 0) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(lds + L_BIAS + 16 + 4 * kq);
        const float hv = bt_lstm_cell(pre, bias, c_d);
        const int hu = j * 4 + kq;
        if (hu < Rd) bt_publish<REP>(g + a.lay.dh, a.lay.dh_cap, utt * KH + hu, ep, hv);
      }
      if (is_p && !forced) acc_r = bt_mfma_lds<2 * NPW_C>(lds + LP_WF + (KH / 4 + wave * 2 * NPW_C) * 64, lane, b, acc_r);
      if (is_m) acc_r = bt_mfma_lds<2 * NPW_C>(lds + LM_W + (KH / 4 + wave * 2 * NPW_C) * 64, lane, b, acc_r);
      acc_a = bt_mfma_regs<2 * NPW_C>(wa_c, b, acc_a);
    }

    // ============ all: the decoder hidden state; P: next frame's layer-1 rows; M: the frame's output ============
    acc_dn = {0.0f, 0.0f, 0.0f, 0.0f};
    {
      bt_nap_begin(nap_dh, wctx);
#pragma unroll
      for (int hlf = 0; hlf < 2; ++hlf) {
        float b[NPW_H];
        bt_load_pairs<NPW_H / 2>(rs, a.gran, par + a.lay.dh + rep * a.lay.dh_cap + g_lane_h + 8u * (NPW_H / 2) * hlf, ep, utt_live, k0_h + 8 * (NPW_H / 2) * hlf, Rd, b, wctx, 0x900u);
        if (hlf == 0) nap_dh = taco_nap_adjust(nap_dh, wctx.spun);
        if (is_p && !forced) acc_r = bt_mfma_lds<NPW_H>(lds + LP_WF + (wave * 2 * NPW_H + NPW_H * hlf) * 64, lane, b, acc_r);
        if (is_m) acc_r = bt_mfma_lds<NPW_H>(lds + LM_W + (wave * 2 * NPW_H + NPW_H * hlf) * 64, lane, b, acc_r);
        if (hlf == 1 && (is_p || is_m)) {
          // the rows are what other workgroups wait for: finish them before the recurrent products below
          if (is_p && forced && t + 1 < a.max_steps) {
            // teacher forcing: the plain layer-1 rows on the given frame (waves 0..4 take 4 k-steps each)
            if (wave < (W1_KSTEPS + 3) / 4) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const int ks = 4 * wave + i;
                const int k = 8 * (ks >> 1) + 2 * kq + (ks & 1);
                float xv = 0.0f;
                if (ks < W1_KSTEPS && utt_live && k < NMEL) xv = a.forced_mel[((size_t)reinterpret_cast<const int*>(lds + L_UTT)[utt] * NMEL + k) * a.T_cap + t];
                if (ks < W1_KSTEPS) acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[LP_W1 + ks * 64 + lane], xv, acc_r, 0, 0, 0);
              }
            }
          }
          const f32x4 sum = bt_reduce_tiles(lds + (is_p ? LP_RED : LM_RED), acc_r, wave, lane);
          if (wave == 0) {
            if (is_p) {
              if (t + 1 < a.max_steps) {
                taco_granule* gn = a.gran + (((t + 1) & 1) ? a.lay.per_parity : 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int row = j * RP + 4 * kq + r;
                  float v = sum[r] + (forced ? 0.0f : lds[LP_BIAS + 4 * kq + r]);
                  v = v > 0.0f ? v : 0.0f;
                  v = ((keep_l1 >> (8 * r)) & 0xffu) ? v * 2.0f : 0.0f;
                  if (row < P) bt_publish<REP>(gn + a.lay.h1, a.lay.h1_cap, utt * KP + row, ep + 1u, v);
                }
                keep_l2 = keep_next;
              }
            } else {
              const int b_out = reinterpret_cast<const int*>(lds + L_UTT)[utt];
              const bool wr = utt_live && ((alive >> utt) & 1u);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int row = (j - n_p_wg) * RP + 4 * kq + r;
                const float v = sum[r] + lds[LM_BIAS + 4 * kq + r];
                if (row < NMEL) {
                  if (wr) a.mel[((size_t)b_out * NMEL + row) * a.T_cap + t] = v;
                } else if (row == NMEL) {
                  bt_publish<REP>(g + a.lay.stop, a.lay.stop_cap, utt, ep, v);
                  if (wr) a.gate[(size_t)b_out * a.T_cap + t] = v;
                }
              }
            }
          }
        }
#pragma unroll
        for (int i = 0; i < NPW_H; ++i) acc_dn = __builtin_amdgcn_mfma_f32_16x16x4f32(wd_h[NPW_H * hlf + i], b[i], acc_dn, 0, 0, 0);
      }
    }
  }

  if (j == 0 && tid < a.n_utts) {
    const bool was_alive = (alive >> tid) & 1u;
    const int frames = was_alive ? t : my_frames;
    const unsigned reached = (was_alive && !forced && t == a.max_steps) ? 1u : 0u;
    const int b_out = reinterpret_cast<const int*>(lds + L_UTT)[tid];
    a.n_frames[b_out] = frames;
    a.reached[b_out] = (uint8_t)reached;
    if (tid == 0) {
      a.ctl->n_frames = (uint32_t)t;
      a.ctl->reached_max = reached;
    }
  }
}

}  // namespace

size_t taco_decoder_batch_lds_bytes() { return (size_t)L_TOTAL * sizeof(float); }

void taco_decoder_batch_fill_layout(taco_bat_layout* lay) {
  int off = 0;
  auto take = [&](int n, int reps, int32_t& cap) {
    const int at = off;
    cap = (n + 15) / 16 * 16;
    off += cap * reps;
    return at;
  };
  lay->h1 = take(NB * KP, TACO_BT_REP, lay->h1_cap);
  lay->p = take(NB * KP, TACO_BT_REP, lay->p_cap);
  lay->ah = take(NB * KH, TACO_BT_REP, lay->ah_cap);
  int32_t pe_cap;
  lay->pe = take(NB * NQ * TIN, 1, pe_cap);
  lay->ctx = take(NB * KC, TACO_BT_REP, lay->ctx_cap);
  lay->dh = take(NB * KH, TACO_BT_REP, lay->dh_cap);
  lay->stop = take(NB, TACO_BT_REP, lay->stop_cap);
  lay->per_parity = off;
}

size_t taco_decoder_batch_granules() {
  taco_bat_layout lay;
  taco_decoder_batch_fill_layout(&lay);
  return (size_t)lay.per_parity * 2;
}

// Whether the lockstep kernel can run this model at all (extents within its capacities).
int taco_decoder_batch_supported(const taco_dims& d, int n_wg) {
  if (d.P > KP || d.P % 4 != 0) return 0;
  if (d.R_a > KH || d.R_d > KH || d.R_a > 4 * n_wg || d.R_d > 4 * n_wg) return 0;
  if (d.M > NQ * MEMW || d.M > KC) return 0;
  if (d.A > TACO_MAX_A || d.K > TACO_MAX_K || (d.K & 1) == 0 || d.n_mel > TACO_MAX_MEL) return 0;
  const int groups = (d.P + RP - 1) / RP + (d.n_mel + 1 + RP - 1) / RP + NB * NQ;
  if (groups > n_wg) return 0;
  return 1;
}

int taco_decoder_batch_prepare(taco_handle h) {
  const void* k = reinterpret_cast<const void*>(taco_decoder_batch_kernel);
  TACO_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)taco_decoder_batch_lds_bytes()));
  int per_cu = 0;
  TACO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, NT, taco_decoder_batch_lds_bytes()));
  if (per_cu < 1) TACO_FAIL(TACO_EINVAL, "lockstep decoder kernel does not fit on a CU (occupancy query returned %d)", per_cu);
  return TACO_OK;
}

// One lockstep launch for the utterances utt[0..n) (n <= 16, every one of at most TACO_BT_TIN symbols).
int taco_launch_decoder_batch(taco_handle h, const float* memory, const float* pm, int T_in, const int32_t* lens,
                              const int* utt, int n, int group, const uint8_t* masks, int T_cap, int max_steps, float thr,
                              const float* forced_mel, float* mel, float* gate, float* align, int32_t* n_frames,
                              uint8_t* reached, hipStream_t s) {
  const taco_dims& d = h->d;
  if (n < 1 || n > NB) TACO_FAIL(TACO_EINVAL, "lockstep group of %d utterances (1..%d)", n, NB);
  if (T_in < 1 || T_in > TACO_MAX_TIN) TACO_FAIL(TACO_EINVAL, "T_in = %d", T_in);
  if (max_steps < 1 || max_steps > T_cap) TACO_FAIL(TACO_EINVAL, "max_steps %d outside 1..T_cap=%d", max_steps, T_cap);
  BatArgs a;
  a.P = d.P; a.M = d.M; a.Ra = d.R_a; a.Rd = d.R_d; a.A = d.A; a.K = d.K; a.n_mel = d.n_mel; a.T_in = T_in; a.n_wg = h->n_cu;
  a.n_utts = n;
  for (int i = 0; i < NB; ++i) a.utt[i] = utt[i < n ? i : 0];
  a.lens = lens;
  const taco_weights& w = h->w;
  a.prenet_w1 = w.prenet_w1; a.prenet_w2 = w.prenet_w2; a.wf_h = h->wf_h; a.wf_c = h->wf_c; a.wf_b = h->wf_b;
  a.att_w_ih = w.att_w_ih; a.att_w_hh = w.att_w_hh; a.att_b_ih = w.att_b_ih; a.att_b_hh = w.att_b_hh;
  a.query_w = w.query_w; a.v_w = w.v_w; a.loc_fold = h->loc_fold;
  a.dec_w_ih = w.dec_w_ih; a.dec_w_hh = w.dec_w_hh; a.dec_b_ih = w.dec_b_ih; a.dec_b_hh = w.dec_b_hh;
  a.proj_w = w.proj_w; a.proj_b = w.proj_b; a.gate_w = w.gate_w; a.gate_b = w.gate_b;
  a.memory = memory; a.pm = pm; a.masks = masks; a.T_cap = T_cap; a.max_steps = max_steps; a.thr = thr;
  a.shift_energies = h->energy_bound > 64.0f ? 1 : 0;
  a.forced_mel = forced_mel; a.mel = mel; a.gate = gate; a.align = align; a.n_frames = n_frames; a.reached = reached;
  a.gran = h->bat_gran; a.lay = h->bat_lay;
  a.gran_bytes = (unsigned)((size_t)h->bat_lay.per_parity * 2 * sizeof(taco_granule));
  a.ctl = h->dec_ctl + group;
  a.frame_clock = group == 0 ? h->frame_clock : nullptr;
  a.frame_clock_cap = h->frame_clock_cap;
  TACO_HIP(hipMemsetAsync(h->bat_gran, 0, a.gran_bytes, s));
  if (group == 0) TACO_HIP(hipEventRecord(h->ev_start, s));
  hipLaunchKernelGGL(taco_decoder_batch_kernel, dim3(h->n_cu), dim3(NT), taco_decoder_batch_lds_bytes(), s, a);
  TACO_HIP(hipGetLastError());
  return TACO_OK;
}
