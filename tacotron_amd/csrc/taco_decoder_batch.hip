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

constexpr int NT = TACO_BT_THREADS;     // 4 waves, one per SIMD: each may use the SIMD's whole register file (512 per lane)
constexpr int NW = NT / 64;
constexpr int NB = TACO_BATCH_NB;
static_assert(NW == 4, "softmax and query reductions are written for 4 waves");
static_assert(NB == 16, "one utterance per column of a 16 x 16 matrix-core tile");
// k capacities of the exchanged vectors: multiples of 64 = 8 waves x (pairs of 4-deep steps)
constexpr int KP = TACO_BT_KP, KH = TACO_BT_KH, KC = TACO_BT_KC;
constexpr int NPW_P = KP / 8 / NW;     // pairs of k-steps per wave: 8
constexpr int NPW_H = KH / 8 / NW;     // 32
constexpr int NPW_C = KC / 8 / NW;     // 18
constexpr int HCH = 8;                 // pairs per load burst of the long vectors
constexpr int TIN = TACO_BT_TIN;       // positions per utterance (256)
constexpr int HALO = 16;
constexpr int QD = 16;                 // attention dimensions per Q workgroup
constexpr int NQ = TACO_NQ_WG;         // Q workgroups per utterance (8)
constexpr int MEMW = TACO_BT_MEMW;     // memory columns per Q workgroup (68)
constexpr int CQ = TACO_BT_MEMW / 4;   // column quads of a memory slice (17)
constexpr int CPG = 15;                // position groups of the context sum (15 x 17 threads)
constexpr int QT = (TACO_BT_TIN / 16 + NW - 1) / NW;   // tiles of 16 positions per wave (4: wave w takes tiles w, w + 4, ...)
constexpr int QR = 16 / NW;            // query rows per wave (4)
constexpr int LOC_KSTEPS = 16;
constexpr int RP = TACO_BT_ROWS;       // rows per P / M workgroup (one MFMA tile)
constexpr int W1_KSTEPS = (TACO_MAX_MEL + 3) / 4;    // 20
static_assert(2 * TACO_MAX_K <= 4 * LOC_KSTEPS, "location taps fit 16 steps of 4");
static_assert(QR == 4 && CPG * CQ <= NT && MEMW % 4 == 0 && NQ * MEMW >= 544, "context sum thread map");

// ---- LDS carve (float offsets) ----
constexpr int L_RED = 0;                         // [4 waves][64 lanes][4]: partial tiles of an LSTM stage
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
constexpr int LQ_SRED = LQ_Q + QD;
constexpr int LQ_U = LQ_SRED + 32;               // location term + processed memory [position][16 dimensions]
constexpr int LQ_END = LQ_U + TIN * QD;
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
#ifdef TACO_BT_STAMPS
  unsigned long long* stamps;     // diagnostic build only: [4 workgroups][BT_STAMP_FRAMES][BT_STAMP_POINTS]
#endif
};

#ifdef TACO_BT_STAMPS
constexpr int BT_STAMP_FRAMES = 256, BT_STAMP_POINTS = 16;
#define BT_STAMP(k)                                                                                         \
  do {                                                                                                      \
    if (tid == 0 && stamp_wg >= 0 && t < BT_STAMP_FRAMES)                                                   \
      a.stamps[((size_t)stamp_wg * BT_STAMP_FRAMES + t) * BT_STAMP_POINTS + (k)] = wall_clock64();          \
  } while (0)
#else
#define BT_STAMP(k) do {} while (0)
#endif

// B operands come straight from the hand-off words: lane (utterance u = l & 15, q = l >> 4) fetches 16 bytes = the two
// granules {payload, epoch} x 2 of a pair.  Every granule of every vector is published every frame (padding beyond
// a vector's true length as zeros, empty utterance slots as copies of utterance 0), so the check is uniform: all
// tags must carry the epoch.  bt_issue starts a burst of NPAIR such loads, bt_finish checks it and, when a tag is
// not there yet, fetches the burst again after a short sleep (one compact loop per burst: after the nap the values
// are there at the first attempt as a rule).  Issuing the next burst before finishing the current one keeps the
// memory round trip beside the matrix products.
template <int NPAIR>
__device__ __forceinline__ void bt_issue(__amdgpu_buffer_rsrc_t rs, unsigned gidx, u32x4 (&v)[NPAIR]) {
#pragma unroll
  for (int i = 0; i < NPAIR; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (gidx + 8u * i) * 8u, 0, 16 /* sc1 */);
}
template <int NPAIR>
__device__ __forceinline__ void bt_finish(__amdgpu_buffer_rsrc_t rs, unsigned gidx, unsigned ep, u32x4 (&v)[NPAIR],
                                          float (&b)[2 * NPAIR], taco_wait_ctx& ctx, unsigned info) {
  unsigned spins = 0;
  for (;;) {
    unsigned bad = 0;
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) bad |= (v[i].y ^ ep) | (v[i].w ^ ep);
    if (__all(bad == 0u) || ctx.failed) break;
    ctx.spun = 1;
    __builtin_amdgcn_s_sleep(TACO_POLL_SLEEP);
    if ((++spins & 255u) == 0u) {
      if (__hip_atomic_load(ctx.abort_word, TACO_RLX_AGENT) != 0u) { ctx.failed = 1; break; }
      if (spins >= TACO_SPIN_LIMIT) {
        __hip_atomic_store(ctx.abort_word, TACO_ABORT_TIMEOUT, TACO_RLX_AGENT);
        __hip_atomic_store(ctx.abort_word + 1, info, TACO_RLX_AGENT);
        ctx.failed = 1;
        break;
      }
    }
    asm volatile("" ::: "memory");
    bt_issue<NPAIR>(rs, gidx, v);
  }
#pragma unroll
  for (int i = 0; i < NPAIR; ++i) {
    b[2 * i] = __uint_as_float(v[i].x);
    b[2 * i + 1] = __uint_as_float(v[i].z);
  }
}
template <int NPAIR>
__device__ __forceinline__ void bt_load_pairs(__amdgpu_buffer_rsrc_t rs, unsigned gidx, unsigned ep, float (&b)[2 * NPAIR],
                                              taco_wait_ctx& ctx, unsigned info) {
  u32x4 v[NPAIR];
  bt_issue<NPAIR>(rs, gidx, v);
  bt_finish<NPAIR>(rs, gidx, ep, v, b, ctx, info);
}

// Gathers up to NPT pairs of granules per thread into LDS as plain floats (Q workgroups: the hidden state and the
// partial energies of their own utterance).  src[i] = granule index of the pair, dst[i] = LDS float index, n = pairs in use.
template <int NPT>
__device__ __forceinline__ void bt_gather_pairs(__amdgpu_buffer_rsrc_t rs, const unsigned (&src)[NPT], const int (&dst)[NPT], int n,
                                                unsigned ep, float* lds, taco_wait_ctx& ctx, unsigned info) {
  u32x4 v[NPT];
  unsigned spins = 0;
  for (;;) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, src[i] * 8u, 0, 16 /* sc1 */);
    unsigned bad = 0;
#pragma unroll
    for (int i = 0; i < NPT; ++i)
      if (i < n) bad |= (v[i].y ^ ep) | (v[i].w ^ ep);
    if (__all(bad == 0u) || ctx.failed) break;
    ctx.spun = 1;
    __builtin_amdgcn_s_sleep(TACO_POLL_SLEEP);
    if ((++spins & 255u) == 0u) {
      if (__hip_atomic_load(ctx.abort_word, TACO_RLX_AGENT) != 0u) { ctx.failed = 1; break; }
      if (spins >= TACO_SPIN_LIMIT) {
        __hip_atomic_store(ctx.abort_word, TACO_ABORT_TIMEOUT, TACO_RLX_AGENT);
        __hip_atomic_store(ctx.abort_word + 1, info, TACO_RLX_AGENT);
        ctx.failed = 1;
        break;
      }
    }
    asm volatile("" ::: "memory");
  }
#pragma unroll
  for (int i = 0; i < NPT; ++i)
    if (i < n) *reinterpret_cast<float2*>(lds + dst[i]) = make_float2(__uint_as_float(v[i].x), __uint_as_float(v[i].z));
}

// An LDS index behind a compiler barrier.  The carve of a Q workgroup reaches far past the 64 KB an LDS instruction's
// immediate offset covers, so the compiler forms one address register per access, finds them all loop-invariant and
// hoists them out of the frame loop - dozens of registers this kernel does not have (they ended up in scratch memory).
// An index made opaque inside the frame is recomputed there (one or two vector operations) and the accesses near it
// become immediate offsets from it.
#define BT_OPQ(expr) ({ int v_ = (expr); asm volatile("" : "+v"(v_)); v_; })

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

// The waves' partial tiles -> LDS at float index `red`; after the barrier wave 0 holds the sum (other waves get their own
// partial back).
__device__ __forceinline__ f32x4 bt_reduce_tiles(float* lds, int red, f32x4 acc, int wave, int lane) {
  const int at = BT_OPQ(red + lane * 4);
  *reinterpret_cast<f32x4*>(lds + at + wave * 256) = acc;
  taco_lds_barrier();
  if (wave == 0) {
#pragma unroll
    for (int w = 1; w < NW; ++w) acc += *reinterpret_cast<const f32x4*>(lds + at + w * 256);
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

__global__ __launch_bounds__(TACO_BT_THREADS) void taco_decoder_batch_kernel(const BatArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = blockIdx.x;
  const int P = a.P, M = a.M, Ra = a.Ra, Rd = a.Rd, A = a.A, K = a.K, NMEL = a.n_mel;
  const int utt = lane & 15, kq = lane >> 4;      // B operand / result column = utterance slot; k offset / unit
  constexpr int REP = TACO_BT_REP;
  const int rep = j % REP;

  for (int i = tid; i < L_TOTAL; i += NT) lds[i] = 0.0f;
  __syncthreads();

  // ---------------- roles ----------------
  const int n_p_wg = KP / RP;                             // rows beyond P have zero weights and publish zeros
  const int n_m_wg = (NMEL + 1 + RP - 1) / RP;
  const int qd = (A + NQ - 1) / NQ;                       // attention dimensions per Q workgroup (<= 16)
  const int n_dg = NQ;                                    // every group runs (one without dimensions contributes zeros)
  const int q_off = n_p_wg + n_m_wg;
  const bool is_p = j < n_p_wg;
  const bool is_m = j >= n_p_wg && j < q_off;
  const int q_id = j - q_off;
  const bool is_q = q_id >= 0 && q_id < NB * NQ;          // empty slots run as copies of utterance 0, unwritten
  const int q_slot = is_q ? q_id / NQ : 0;
  const int dg = is_q ? q_id % NQ : 0;
  const int q_b = a.utt[q_slot];                          // my utterance (Q)
  const int len = is_q ? (a.lens ? min(a.lens[q_b], a.T_in) : a.T_in) : 0;
  const int len4 = (len + 3) & ~3;
  const int q_base = dg * qd;
  const int q_cnt = is_q ? max(0, min(qd, A - q_base)) : 0;
  const int cw = (M + NQ - 1) / NQ;                       // memory columns per Q workgroup (<= MEMW)
  const int c0 = dg * cw;
  const int c_cnt = is_q ? max(0, min(cw, M - c0)) : 0;    // memory columns I hold (I publish cw: the rest are zeros)
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
  float vq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (is_q) {
    const float* mem_b = a.memory + (size_t)q_b * a.T_in * M;
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

  // Q: the location term of my tiles, U = processed memory + sum over taps of E * {previous, cumulative} weights.
  // Lane (position col = l & 15, kq): dimensions 4 kq .. + 3 of the tile's 16 x 16 result; the B operand of 4-tap step m
  // is weight (or cumulative weight) number 4 m + kq of the 2K taps, read from the halo-padded LDS copies.  The
  // processed memory is read again every frame (L2-resident, asked for before the products, added after them):
  // keeping it in registers would cost 16 of them for the whole launch.
  const float* pm_lane = a.pm + (size_t)q_b * a.T_in * A + q_base + 4 * kq;      // + position * A
  const bool pm_vec = (A & 3) == 0 && (q_base & 3) == 0 && q_cnt == QD;         // my 4 dimensions are one aligned 16-byte load
  auto load_pm = [&](int s, float (&out)[4]) {
    if (pm_vec) {
      const float4 v = s < len ? *reinterpret_cast<const float4*>(pm_lane + (size_t)s * A) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) out[r] = (s < len && 4 * kq + r < q_cnt) ? pm_lane[(size_t)s * A + r] : 0.0f;
    }
  };
  if (is_q) {
#pragma unroll
    for (int ti = 0; ti < QT; ++ti) {
      const int s = 16 * (wave + NW * ti) + utt;
      if (s < TIN) {
        float pm0[4];
        load_pm(s, pm0);
        *reinterpret_cast<float4*>(lds + LQ_U + s * QD + 4 * kq) = make_float4(pm0[0], pm0[1], pm0[2], pm0[3]);
      }
    }
  }
  // (lane_o: the lane number behind a compiler barrier, taken anew every frame.  Every address below derives from it, so
  // none of them is loop-invariant: hoisted out of the frame loop, the 16 + 16 per-step addresses would sit in
  // registers - or, with the LSTM rows filling the register file, in scratch memory - for the whole launch.)
  auto update_location_term = [&](int lane_o) {
    const int col = lane_o & 15, kqo = lane_o >> 4;
    const int x_base = BT_OPQ(LQ_AW + HALO + col + kqo - pad);     // LDS index of tap kq at position 0 (previous weights)
    const float* e_lane = lds + BT_OPQ(LQ_E + lane_o);
    float* u_lane = lds + BT_OPQ(LQ_U + col * QD + 4 * kqo);
#pragma unroll
    for (int tp = 0; tp < QT; tp += 2) {       // two tiles at a time: two independent accumulation chains
      const int s0 = 16 * (wave + NW * tp), s1 = 16 * (wave + NW * (tp + 1));
      if (s0 < len) {
        const int s1c = s1 < len ? s1 : s0;      // a tile past the end repeats the first (its result is not used)
        float pm0[4], pm1[4];
        load_pm(s0 + col, pm0);
        load_pm(s1c + col, pm1);
        f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
        // tap ck = 4 m + kq: previous weights for ck < K, cumulative weights (tap ck - K) from K on; taps beyond 2K
        // meet a zero in A, any valid address will do.  Operands are read four steps ahead of their products.
        float ev[2][4], x0[2][4], x1[2][4];
        auto read_group = [&](int gq, float (&e)[4], float (&xa)[4], float (&xb)[4]) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int m = 4 * gq + i;
            const int ck = 4 * m + kqo;
            const int off = x_base + 4 * m + (ck >= K ? (ck < 2 * K ? LQ_CUM - LQ_AW - K : -4 * m - kqo) : 0);
            e[i] = e_lane[m * 64];
            xa[i] = lds[off + s0];
            xb[i] = lds[off + s1c];
          }
        };
        read_group(0, ev[0], x0[0], x1[0]);
#pragma unroll
        for (int gq = 0; gq < LOC_KSTEPS / 4; ++gq) {
          if (gq + 1 < LOC_KSTEPS / 4) read_group(gq + 1, ev[(gq + 1) & 1], x0[(gq + 1) & 1], x1[(gq + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ev[gq & 1][i], x0[gq & 1][i], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ev[gq & 1][i], x1[gq & 1][i], acc1, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        *reinterpret_cast<float4*>(u_lane + s0 * QD) = make_float4(acc0[0] + pm0[0], acc0[1] + pm0[1], acc0[2] + pm0[2], acc0[3] + pm0[3]);
        if (s1 < len)
          *reinterpret_cast<float4*>(u_lane + s1 * QD) = make_float4(acc1[0] + pm1[0], acc1[1] + pm1[1], acc1[2] + pm1[2], acc1[3] + pm1[3]);
      }
    }
  };
  // Q: what each thread gathers.  Hidden state of my utterance: pairs tid, tid + 256 of its 512.  Partial energies:
  // thread (group tid >> 5, l = tid & 31) takes pairs l, l + 32, l + 64, l + 96 of that group's positions.
  const int len16 = (len + 15) & ~15;
  const int pe_n = max(0, min(4, (len16 / 2 - (tid & 31) + 31) / 32));
  const unsigned pe_src0 = (unsigned)(a.lay.pe + (q_slot * NQ + (tid >> 5)) * TIN + 2 * (tid & 31));

  f32x4 acc_a = {0.0f, 0.0f, 0.0f, 0.0f};     // attention-LSTM partial over [context ; hidden] of the previous frame
  f32x4 acc_dn = {0.0f, 0.0f, 0.0f, 0.0f};    // decoder-LSTM partial over the previous decoder hidden state
  float c_a = 0.0f, c_d = 0.0f;               // cell states (wave 0: lane = (unit kq, utterance utt))
  unsigned alive = a.n_utts >= 32 ? 0xffffffffu : ((1u << a.n_utts) - 1u);
  int my_frames = 0;                           // thread tid < 16 of every workgroup: frames of utterance slot tid
  int t = 0;
  // granule index of this lane's first pair in each vector, replica `rep`
  const unsigned g_lane_p = (unsigned)(rep * a.lay.p_cap + utt * KP + 8 * (wave * NPW_P) + 2 * kq);
  const unsigned g_lane_h = (unsigned)(rep * a.lay.ah_cap + utt * KH + 8 * (wave * NPW_H) + 2 * kq);
  const unsigned g_lane_c = (unsigned)(rep * a.lay.ctx_cap + utt * KC + 8 * (wave * NPW_C) + 2 * kq);
  unsigned keep_next = 0;                      // P, wave 0: keep bytes of my 4 rows, layer 0 (low) / layer 1 of the coming frame
  unsigned keep_l1 = 0, keep_l2 = 0;
  if (is_p && wave == 0) {
    const int r0 = j * RP + 4 * kq;
    if (r0 < P) {
      keep_l2 = *reinterpret_cast<const unsigned*>(a.masks + P + r0);              // frame 0, layer 2
    }
  }

  // A stream of the bursts of HCH pairs of this lane's slice of one hidden-state vector, double-buffered: BODY(ch, b) sees
  // burst ch's 2 * HCH operands.  PRE(ch) runs after the next burst has been issued and before this one is waited for (LDS
  // reads of A operands go there).  The scheduling barriers pin the order issue-next / wait-current / products: left to
  // itself the compiler, short of registers, moves the next burst's wait in among this burst's products.
#define BT_SB() __builtin_amdgcn_sched_barrier(0)
#define BT_STREAM_H(base_, info_, ADJ, PRE, BODY)                                                  \
  {                                                                                                \
    u32x4 va[HCH], vb[HCH];                                                                        \
    const unsigned base = (base_);                                                                 \
    bt_issue<HCH>(rs, base, va);                                                                   \
    BT_SB();                                                                                       \
    _Pragma("unroll") for (int ch = 0; ch < NPW_H / HCH; ch += 2) {                                \
      bt_issue<HCH>(rs, base + 8u * HCH * (ch + 1), vb);                                           \
      BT_SB();                                                                                     \
      {                                                                                            \
        float b[2 * HCH];                                                                          \
        PRE(ch);                                                                                   \
        bt_finish<HCH>(rs, base + 8u * HCH * ch, ep, va, b, wctx, info_);                          \
        if (ch == 0) { ADJ; }                                                                      \
        BT_SB();                                                                                   \
        BODY(ch, b);                                                                               \
        BT_SB();                                                                                   \
      }                                                                                            \
      if (ch + 2 < NPW_H / HCH) {                                                                  \
        bt_issue<HCH>(rs, base + 8u * HCH * (ch + 2), va);                                         \
        BT_SB();                                                                                   \
      }                                                                                            \
      {                                                                                            \
        float b[2 * HCH];                                                                          \
        PRE(ch + 1);                                                                               \
        bt_finish<HCH>(rs, base + 8u * HCH * (ch + 1), ep, vb, b, wctx, info_);                    \
        BT_SB();                                                                                   \
        BODY(ch + 1, b);                                                                           \
        BT_SB();                                                                                   \
      }                                                                                            \
    }                                                                                              \
  }
#define BT_NO_PRE(ch)
#define BT_BODY_AH(ch, b)                                                                                          \
  _Pragma("unroll") for (int i = 0; i < HCH; ++i) {                                                                \
    acc_a = __builtin_amdgcn_mfma_f32_16x16x4f32(wa_h[2 * HCH * (ch) + i], b[i], acc_a, 0, 0, 0);                  \
    acc_a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa_h[2 * HCH * (ch) + HCH + i], b[HCH + i], acc_a3, 0, 0, 0);    \
  }

#ifdef TACO_BT_STAMPS
  const int stamp_wg = j == 0 ? 0 : j == n_p_wg ? 1 : j == q_off ? 2 : j == a.n_wg - 1 ? 3 : -1;
#endif
  for (;; ++t) {
    const unsigned ep = (unsigned)t + 1u;
    const unsigned par = (unsigned)(t & 1) * (unsigned)a.lay.per_parity;
    taco_granule* g = a.gran + par;
    const int lane_f = BT_OPQ(lane);            // see BT_OPQ: addresses derived from these are recomputed every frame
    const int tid_f = lane_f + 64 * wave;
    const int utt_f = lane_f & 15, kq_f = lane_f >> 4;

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
        bt_load_pairs<NPW_P>(rs, par + a.lay.h1 + g_lane_p, ep, b, wctx, 0x300u);
        nap_h1 = taco_nap_adjust(nap_h1, wctx.spun);
        f32x4 acc2 = {0.0f, 0.0f, 0.0f, 0.0f};
        const float* w2 = lds + BT_OPQ(LP_W2 + wave * (2 * NPW_P) * 64 + lane_f);
#pragma unroll
        for (int i = 0; i < NPW_P; ++i) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[i * 64], b[i], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[(NPW_P + i) * 64], b[NPW_P + i], acc2, 0, 0, 0);
        }
        acc += acc2;
      }
      acc = bt_reduce_tiles(lds, LP_RED, acc, wave, lane_f);
      if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = j * RP + 4 * kq + r;
          float v = acc[r] > 0.0f ? acc[r] : 0.0f;
          v = ((keep_l2 >> (8 * r)) & 0xffu) ? v * 2.0f : 0.0f;
          bt_publish<REP>(g + a.lay.p, a.lay.p_cap, utt * KP + row, ep, row < P ? v : 0.0f);
        }
        keep_l1 = kn1;
        keep_next = kn2;
      }
      taco_lds_barrier();      // LP_RED is reused by the layer-1 rows below
      if (t > 0) {
        // while the prenet output travels: the hidden columns of this frame's attention-LSTM rows, on the previous frame's
        // attention hidden state (the other workgroups did this at the end of that frame; here it would have sat between
        // the layer-1 rows and layer 2, which everybody waits for)
        const unsigned ep_prev = ep - 1u;
        const unsigned par_prev = (unsigned)((t - 1) & 1) * (unsigned)a.lay.per_parity;
        f32x4 acc_a3 = {0.0f, 0.0f, 0.0f, 0.0f};
        {
          const unsigned ep = ep_prev;
          BT_STREAM_H(par_prev + a.lay.ah + g_lane_h, 0x680u, (void)0, BT_NO_PRE, BT_BODY_AH)
        }
        acc_a += acc_a3;
      }
    }

    BT_STAMP(0);
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
    BT_STAMP(1);

    // ============ attention LSTM ============
    {
      float b[2 * NPW_P];
      bt_nap_begin(nap_p, wctx);
      bt_load_pairs<NPW_P>(rs, par + a.lay.p + g_lane_p, ep, b, wctx, 0x400u);
      nap_p = taco_nap_adjust(nap_p, wctx.spun);
      BT_STAMP(2);
      f32x4 acc2 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int i = 0; i < NPW_P; ++i) {
        acc_a = __builtin_amdgcn_mfma_f32_16x16x4f32(wa_p[i], b[i], acc_a, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa_p[NPW_P + i], b[NPW_P + i], acc2, 0, 0, 0);
      }
      acc_a += acc2;
      const f32x4 pre = bt_reduce_tiles(lds, L_RED, acc_a, wave, lane_f);
      if (wave == 0) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(lds + L_BIAS + 4 * kq_f);
        const float hv = bt_lstm_cell(pre, bias, c_a);
        const int hu = j * 4 + kq;
        bt_publish<REP>(g + a.lay.ah, a.lay.ah_cap, utt * KH + hu, ep, hu < Ra ? hv : 0.0f);
      }
    }
    BT_STAMP(3);

    // ============ Q: query values and partial energies of my 16 dimensions, every position of my utterance ============
    if (is_q) {
      {
        const unsigned src[2] = {par + a.lay.ah + rep * a.lay.ah_cap + q_slot * KH + 2 * tid, par + a.lay.ah + rep * a.lay.ah_cap + q_slot * KH + 2 * (tid + NT)};
        const int d0 = BT_OPQ(LQ_XPE + 2 * tid_f);
        const int dst[2] = {d0, d0 + 2 * NT};
        bt_nap_begin(nap_ahq, wctx);
        bt_gather_pairs<2>(rs, src, dst, 2, ep, lds, wctx, 0x500u);
        nap_ahq = taco_nap_adjust(nap_ahq, wctx.spun);
      }
      taco_lds_barrier();
      BT_STAMP(4);
      {
        // wave w: query values of dimensions 4w .. 4w + 3
        const float* r0 = lds + BT_OPQ(LQ_ROWS + (QR * wave) * KH + lane_f * 4);
        const float* xq = lds + BT_OPQ(LQ_XPE + lane_f * 4);
        float pq[QR];
#pragma unroll
        for (int r = 0; r < QR; ++r) pq[r] = 0.0f;
#pragma unroll
        for (int c = 0; c < KH / 256; ++c) {
          const float4 x = *reinterpret_cast<const float4*>(xq + c * 256);
#pragma unroll
          for (int r = 0; r < QR; ++r) {
            const float4 u = *reinterpret_cast<const float4*>(r0 + r * KH + c * 256);
            pq[r] += (fmaf(u.x, x.x, u.y * x.y)) + (fmaf(u.z, x.z, u.w * x.w));
          }
        }
        float s0, s1, s2, s3;
        taco_wave_sum2(pq[0], pq[1], s0, s1);
        taco_wave_sum2(pq[2], pq[3], s2, s3);
        if (lane == 0) *reinterpret_cast<float4*>(lds + LQ_Q + QR * wave) = make_float4(s0, s1, s2, s3);
      }
      taco_lds_barrier();
      {
        const float* uq = lds + BT_OPQ(LQ_Q + 4 * kq_f);       // query values; the U rows follow within immediate reach
        const f32x4 q4 = *reinterpret_cast<const f32x4*>(uq);
        const float* ul = uq + (LQ_U - LQ_Q) + ((16 * wave + utt_f) * QD);
#pragma unroll
        for (int ti = 0; ti < QT; ++ti) {
          const int s = 16 * (wave + NW * ti) + utt_f;
          if (16 * (wave + NW * ti) < len16) {
            const f32x4 u4 = *reinterpret_cast<const f32x4*>(ul + ti * (16 * NW * QD));
            float e = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) e = fmaf(vq[r], taco_tanh(q4[r] + u4[r]), e);
            e += __shfl_xor(e, 16, 64);
            e += __shfl_xor(e, 32, 64);
            if (kq_f == 0) taco_publish(g + a.lay.pe + (q_slot * NQ + dg) * TIN, s, ep, s < len ? e : 0.0f);
          }
        }
      }
    }
    BT_STAMP(5);

    // ============ all: the attention hidden state into this frame's decoder-LSTM rows ============
    f32x4 acc_d = acc_dn;
    {
      f32x4 acc_d2 = {0.0f, 0.0f, 0.0f, 0.0f};
      bt_nap_begin(nap_ah, wctx);
#define BT_BODY_DA(ch, b)                                                                                          \
  _Pragma("unroll") for (int i = 0; i < HCH; ++i) {                                                                \
    acc_d = __builtin_amdgcn_mfma_f32_16x16x4f32(wd_a[2 * HCH * (ch) + i], b[i], acc_d, 0, 0, 0);                  \
    acc_d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wd_a[2 * HCH * (ch) + HCH + i], b[HCH + i], acc_d2, 0, 0, 0);    \
  }
      BT_STREAM_H(par + a.lay.ah + g_lane_h, 0x600u, nap_ah = taco_nap_adjust(nap_ah, wctx.spun), BT_NO_PRE, BT_BODY_DA)
#undef BT_BODY_DA
      acc_d += acc_d2;
    }
    BT_STAMP(6);

    // ============ Q: energies of my utterance, softmax, my columns of the context ============
    if (is_q) {
      bt_nap_begin(nap_pe, wctx);
      {
        const unsigned src[4] = {par + pe_src0, par + pe_src0 + 64u, par + pe_src0 + 128u, par + pe_src0 + 192u};
        const int pd = BT_OPQ(LQ_XPE + (tid_f >> 5) * TIN + 2 * (tid_f & 31));
        const int dst[4] = {pd, pd + 64, pd + 128, pd + 192};
        bt_gather_pairs<4>(rs, src, dst, pe_n, ep, lds, wctx, 0x700u);
      }
      nap_pe = taco_nap_adjust(nap_pe, wctx.spun);
      taco_lds_barrier();
      BT_STAMP(7);
      float w_own = 0.0f;
      {
        float e = 0.0f;
        const float* pe_t = lds + BT_OPQ(LQ_XPE + tid_f);
        if (tid < len) {
#pragma unroll
          for (int dgi = 0; dgi < NQ; ++dgi) e += pe_t[dgi * TIN];
        }
        float* sred = lds + BT_OPQ(LQ_SRED + (lane_f & 0));
        float mx = 0.0f;
        if (a.shift_energies) {
          const float m = taco_wave_max(tid < len ? e : -INFINITY);
          if (lane == 0) sred[8 + wave] = m;
          taco_lds_barrier();
          const float4 m0 = *reinterpret_cast<const float4*>(sred + 8);
          mx = fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w));
        }
        const float x = tid < len ? taco_exp(e - mx) : 0.0f;
        const float part = taco_wave_sum(x);
        if (lane == 0) sred[wave] = part;
        taco_lds_barrier();
        const float4 s0 = *reinterpret_cast<const float4*>(sred);
        const float inv = 1.0f / ((s0.x + s0.y) + (s0.z + s0.w));
        w_own = x * inv;
      }
      float* aw_t = lds + BT_OPQ(LQ_AW + HALO + tid_f);
      if (tid < TIN) aw_t[0] = w_own;
      if (tid < len) {
        aw_t[LQ_CUM - LQ_AW] += w_own;
        if (dg == 0 && a.align && ((alive >> q_slot) & 1u)) a.align[((size_t)q_b * a.T_cap + t) * a.T_in + tid] = w_own;
      }
      taco_lds_barrier();
      // context columns: thread (column quad cg, position group pg) sums its positions for 4 columns; the 15 groups
      // meet in LDS (the partial-energy area is free again)
      {
        const int cg = tid_f % CQ, pg = tid_f / CQ;
        if (pg < CPG) {
          float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          const float* mrow = lds + BT_OPQ(LQ_MEM + 4 * cg);
          const float* wv = lds + BT_OPQ(LQ_AW + HALO + (lane_f & 0));
          int sp = pg;
          for (; sp + 3 * CPG < len; sp += 4 * CPG) {
            const float w0 = wv[sp], w1 = wv[sp + CPG], w2 = wv[sp + 2 * CPG], w3 = wv[sp + 3 * CPG];
            const float4 m0 = *reinterpret_cast<const float4*>(mrow + sp * MEMW);
            const float4 m1 = *reinterpret_cast<const float4*>(mrow + (sp + CPG) * MEMW);
            const float4 m2 = *reinterpret_cast<const float4*>(mrow + (sp + 2 * CPG) * MEMW);
            const float4 m3 = *reinterpret_cast<const float4*>(mrow + (sp + 3 * CPG) * MEMW);
            s.x = fmaf(w0, m0.x, s.x); s.y = fmaf(w0, m0.y, s.y); s.z = fmaf(w0, m0.z, s.z); s.w = fmaf(w0, m0.w, s.w);
            s.x = fmaf(w1, m1.x, s.x); s.y = fmaf(w1, m1.y, s.y); s.z = fmaf(w1, m1.z, s.z); s.w = fmaf(w1, m1.w, s.w);
            s.x = fmaf(w2, m2.x, s.x); s.y = fmaf(w2, m2.y, s.y); s.z = fmaf(w2, m2.z, s.z); s.w = fmaf(w2, m2.w, s.w);
            s.x = fmaf(w3, m3.x, s.x); s.y = fmaf(w3, m3.y, s.y); s.z = fmaf(w3, m3.z, s.z); s.w = fmaf(w3, m3.w, s.w);
          }
          for (; sp < len; sp += CPG) {
            const float w0 = wv[sp];
            const float4 m0 = *reinterpret_cast<const float4*>(mrow + sp * MEMW);
            s.x = fmaf(w0, m0.x, s.x); s.y = fmaf(w0, m0.y, s.y); s.z = fmaf(w0, m0.z, s.z); s.w = fmaf(w0, m0.w, s.w);
          }
          *reinterpret_cast<float4*>(lds + BT_OPQ(LQ_XPE + pg * MEMW + 4 * cg)) = s;
        }
      }
      taco_lds_barrier();
      if (tid < cw) {
        float s = 0.0f;
        const float* cp = lds + BT_OPQ(LQ_XPE + tid_f);
#pragma unroll
        for (int pg = 0; pg < CPG; ++pg) s += cp[pg * MEMW];
        bt_publish<REP>(g + a.lay.ctx, a.lay.ctx_cap, q_slot * KC + c0 + tid, ep, tid < c_cnt ? s : 0.0f);
      } else if (dg == 0) {
        // the padding columns of the exchanged vector
        for (int c = NQ * cw + (tid - cw); c < KC; c += NT - cw) bt_publish<REP>(g + a.lay.ctx, a.lay.ctx_cap, q_slot * KC + c, ep, 0.0f);
      }
      BT_STAMP(8);
      // next frame's location term, while the context travels
      update_location_term(lane_f);
      BT_STAMP(9);
    }
    if (a.frame_clock && j == a.n_wg - 1 && tid == 0 && t < a.frame_clock_cap) a.frame_clock[t] = wall_clock64();

    // ============ all: the context finishes the decoder LSTM ============
    f32x4 acc_r = {0.0f, 0.0f, 0.0f, 0.0f};     // P: folded layer-1 rows; M: projection rows
    {
      constexpr int CCH = NPW_C / 2;          // two bursts of 9 pairs
      bt_nap_begin(nap_ctx, wctx);
      f32x4 acc_d2 = {0.0f, 0.0f, 0.0f, 0.0f};
      u32x4 va[CCH], vb[CCH];
      float b0[2 * CCH], b1[2 * CCH];
      bt_issue<CCH>(rs, par + a.lay.ctx + g_lane_c, va);
      bt_issue<CCH>(rs, par + a.lay.ctx + g_lane_c + 8u * CCH, vb);
      bt_finish<CCH>(rs, par + a.lay.ctx + g_lane_c, ep, va, b0, wctx, 0x800u);
      nap_ctx = taco_nap_adjust(nap_ctx, wctx.spun);
      bt_finish<CCH>(rs, par + a.lay.ctx + g_lane_c + 8u * CCH, ep, vb, b1, wctx, 0x800u);
      BT_STAMP(10);
      // the decoder LSTM's rows first (two chains), its cell update and publish, then everything else on these operands
#pragma unroll
      for (int i = 0; i < 2 * CCH; ++i) {
        acc_d = __builtin_amdgcn_mfma_f32_16x16x4f32(wd_c[i], b0[i], acc_d, 0, 0, 0);
        acc_d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wd_c[2 * CCH + i], b1[i], acc_d2, 0, 0, 0);
      }
      acc_d += acc_d2;
      const f32x4 pre = bt_reduce_tiles(lds, L_RED, acc_d, wave, lane_f);
      if (wave == 0) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(lds + L_BIAS + 16 + 4 * kq_f);
        const float hv = bt_lstm_cell(pre, bias, c_d);
        const int hu = j * 4 + kq;
        bt_publish<REP>(g + a.lay.dh, a.lay.dh_cap, utt * KH + hu, ep, hu < Rd ? hv : 0.0f);
      }
      BT_STAMP(11);
      // while the decoder hidden state travels: the context columns of the P / M rows and of next frame's attention LSTM
      if ((is_p && !forced) || is_m) {
        const float* crow = lds + BT_OPQ((is_p ? LP_WF : LM_W) + (KH / 4 + wave * 2 * NPW_C) * 64 + lane_f);
        acc_r = bt_mfma_lds<2 * CCH>(crow, 0, b0, acc_r);
        acc_r = bt_mfma_lds<2 * CCH>(crow + 2 * CCH * 64, 0, b1, acc_r);
      }
      acc_a = {0.0f, 0.0f, 0.0f, 0.0f};
      f32x4 acc_a2 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int i = 0; i < 2 * CCH; ++i) {
        acc_a = __builtin_amdgcn_mfma_f32_16x16x4f32(wa_c[i], b0[i], acc_a, 0, 0, 0);
        acc_a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa_c[2 * CCH + i], b1[i], acc_a2, 0, 0, 0);
      }
      acc_a += acc_a2;
    }
    // ... and its hidden columns, on a second fetch of this frame's attention hidden state (P / M do this after their rows)
    if (!(is_p || is_m)) {
      f32x4 acc_a3 = {0.0f, 0.0f, 0.0f, 0.0f};
      BT_STREAM_H(par + a.lay.ah + g_lane_h, 0x680u, (void)0, BT_NO_PRE, BT_BODY_AH)
      acc_a += acc_a3;
    }

    // ============ the decoder hidden state; P: next frame's layer-1 rows; M: the frame's output ============
    bt_nap_begin(nap_dh, wctx);
    if (is_p || is_m) {
      // these rows are what other workgroups wait for: they come first; the operands are fetched a second time
      // for the recurrent products below
      const float* rows = lds + BT_OPQ((is_p ? LP_WF : LM_W) + (wave * 2 * NPW_H) * 64 + lane_f);
      const bool use_rows = is_m || !forced;
      f32x4 acc_r2 = {0.0f, 0.0f, 0.0f, 0.0f};
      float aop[2 * HCH];
#define BT_PRE_ROWS(ch)                                                                                            \
  if (use_rows) {                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 2 * HCH; ++i) aop[i] = rows[(2 * HCH * (ch) + i) * 64];                  \
  }
#define BT_BODY_ROWS(ch, b)                                                                                        \
  if (use_rows) {                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < HCH; ++i) {                                                              \
      acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[i], b[i], acc_r, 0, 0, 0);                                  \
      acc_r2 = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[HCH + i], b[HCH + i], acc_r2, 0, 0, 0);                    \
    }                                                                                                              \
  }
      BT_STREAM_H(par + a.lay.dh + g_lane_h, 0x900u, nap_dh = taco_nap_adjust(nap_dh, wctx.spun); BT_STAMP(12), BT_PRE_ROWS, BT_BODY_ROWS)
#undef BT_PRE_ROWS
#undef BT_BODY_ROWS
      acc_r += acc_r2;
      if (is_p && forced && t + 1 < a.max_steps) {
        // teacher forcing: the plain layer-1 rows on the given frame (wave w takes k-steps 5w .. 5w + 4)
        constexpr int W1_PER_WAVE = (W1_KSTEPS + NW - 1) / NW;
        const float* w1row = lds + BT_OPQ(LP_W1 + W1_PER_WAVE * wave * 64 + lane_f);
#pragma unroll
        for (int i = 0; i < W1_PER_WAVE; ++i) {
          const int ks = W1_PER_WAVE * wave + i;
          const int k = 8 * (ks >> 1) + 2 * kq_f + (ks & 1);
          float xv = 0.0f;
          if (ks < W1_KSTEPS && k < NMEL) xv = a.forced_mel[((size_t)reinterpret_cast<const int*>(lds + L_UTT)[utt_f] * NMEL + k) * a.T_cap + t];
          if (ks < W1_KSTEPS) acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(w1row[i * 64], xv, acc_r, 0, 0, 0);
        }
      }
      const f32x4 sum = bt_reduce_tiles(lds, is_p ? LP_RED : LM_RED, acc_r, wave, lane_f);
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
              bt_publish<REP>(gn + a.lay.h1, a.lay.h1_cap, utt * KP + row, ep + 1u, row < P ? v : 0.0f);
            }
            keep_l2 = keep_next;
          }
        } else {
          const int b_out = reinterpret_cast<const int*>(lds + L_UTT)[utt];
          const bool wr = (alive >> utt) & 1u;
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
      BT_STAMP(13);
      if (is_m) {     // (P: at the top of the next frame, after it has published the prenet output)
        f32x4 acc_a3 = {0.0f, 0.0f, 0.0f, 0.0f};
        BT_STREAM_H(par + a.lay.ah + g_lane_h, 0x680u, (void)0, BT_NO_PRE, BT_BODY_AH)
        acc_a += acc_a3;
      }
    }
    {
      const bool first_fetch = !(is_p || is_m);
      acc_dn = {0.0f, 0.0f, 0.0f, 0.0f};
      f32x4 acc_dn2 = {0.0f, 0.0f, 0.0f, 0.0f};
#define BT_BODY_DH(ch, b)                                                                                          \
  _Pragma("unroll") for (int i = 0; i < HCH; ++i) {                                                                \
    acc_dn = __builtin_amdgcn_mfma_f32_16x16x4f32(wd_h[2 * HCH * (ch) + i], b[i], acc_dn, 0, 0, 0);                \
    acc_dn2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wd_h[2 * HCH * (ch) + HCH + i], b[HCH + i], acc_dn2, 0, 0, 0);  \
  }
      BT_STREAM_H(par + a.lay.dh + g_lane_h, 0x980u, if (first_fetch) { nap_dh = taco_nap_adjust(nap_dh, wctx.spun); BT_STAMP(12); }, BT_NO_PRE, BT_BODY_DH)
#undef BT_BODY_DH
      acc_dn += acc_dn2;
    }
    BT_STAMP(14);
  }
#undef BT_STREAM_H
#undef BT_BODY_AH

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
  if (d.R_a > KH || d.R_d > KH) return 0;
  if (d.M > NQ * MEMW || d.M > KC) return 0;
  if (d.A > TACO_MAX_A || d.K > TACO_MAX_K || (d.K & 1) == 0 || d.n_mel > TACO_MAX_MEL) return 0;
  const int groups = KP / RP + (d.n_mel + 1 + RP - 1) / RP + NB * NQ;
  if (groups > TACO_BT_WGS || n_wg < TACO_BT_WGS) return 0;    // the kernel runs 256 workgroups, one per CU
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

#ifdef TACO_BT_STAMPS
#include <stdio.h>
static unsigned long long* g_bt_stamps = nullptr;
// Diagnostic build: mean time between consecutive stamp points of the last launch, per stamped workgroup.
void taco_decoder_batch_dump_stamps(int n_frames) {
  if (!g_bt_stamps) return;
  const size_t n = (size_t)4 * BT_STAMP_FRAMES * BT_STAMP_POINTS;
  std::vector<unsigned long long> st(n);
  if (hipMemcpy(st.data(), g_bt_stamps, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
  const int nfr = n_frames < BT_STAMP_FRAMES ? n_frames : BT_STAMP_FRAMES;
  static const char* wgn[4] = {"P (0)", "M (first)", "Q (first)", "plain (last)"};
  static const char* names[BT_STAMP_POINTS] = {"top: P layer 2 + publish p", "stop test", "p loaded", "att MFMA+reduce+cell+publish ah",
      "Q: ah gathered", "Q: query+energies+publish pe", "bulk ah loads + MFMAs", "Q: pe gathered", "Q: softmax+ctx+publish",
      "Q: location term", "ctx loaded", "dec MFMA+reduce+cell+publish dh", "dh first half loaded", "P/M rows+epilogue (incl 2nd half load)",
      "recurrent MFMAs", ""};
  for (int wg = 0; wg < 4; ++wg) {
    double acc[BT_STAMP_POINTS] = {0};
    int cnt = 0;
    for (int t = 40; t + 1 < nfr; ++t, ++cnt) {
      const unsigned long long* r = &st[((size_t)wg * BT_STAMP_FRAMES + t) * BT_STAMP_POINTS];
      const unsigned long long* nx = &st[((size_t)wg * BT_STAMP_FRAMES + t + 1) * BT_STAMP_POINTS];
      unsigned long long prev = r[0];
      for (int k = 1; k < 15; ++k) {
        if (r[k] == 0) continue;
        acc[k] += (double)(r[k] - prev) * 0.01;
        prev = r[k];
      }
      acc[0] += (double)(nx[0] - prev) * 0.01;
    }
    if (cnt < 1) continue;
    fprintf(stderr, "[bt stamps] workgroup %s, mean over %d frames (us):\n", wgn[wg], cnt);
    double tot = 0;
    for (int k = 1; k < 15; ++k) if (acc[k] > 0) { fprintf(stderr, "  %-44s %7.2f\n", names[k], acc[k] / cnt); tot += acc[k] / cnt; }
    fprintf(stderr, "  %-44s %7.2f\n  total %7.2f\n", names[0], acc[0] / cnt, tot + acc[0] / cnt);
  }
}
#endif

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
  a.P = d.P; a.M = d.M; a.Ra = d.R_a; a.Rd = d.R_d; a.A = d.A; a.K = d.K; a.n_mel = d.n_mel; a.T_in = T_in; a.n_wg = TACO_BT_WGS;
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
#ifdef TACO_BT_STAMPS
  {
    const size_t n = (size_t)4 * BT_STAMP_FRAMES * BT_STAMP_POINTS;
    if (!g_bt_stamps) TACO_HIP(hipMalloc((void**)&g_bt_stamps, n * 8));
    TACO_HIP(hipMemsetAsync(g_bt_stamps, 0, n * 8, s));
    a.stamps = g_bt_stamps;
  }
#endif
  if (group == 0) TACO_HIP(hipEventRecord(h->ev_start, s));
  hipLaunchKernelGGL(taco_decoder_batch_kernel, dim3(TACO_BT_WGS), dim3(NT), taco_decoder_batch_lds_bytes(), s, a);
  TACO_HIP(hipGetLastError());
  return TACO_OK;
}
