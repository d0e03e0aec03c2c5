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
// little work in between, and the matrix cores idle.  The hand-offs' latency is the same whether they
// carry one utterance or sixteen, and sixteen utterances turn every weight row's dot product into a
// column of a 16 x 16 tile: the LSTM gate rows become v_mfma_f32_16x16x4_f32 products (exact f32).
//
// Layout of the work (one workgroup of 256 threads per CU, resident for the whole launch; four waves,
// one per SIMD, so that each may use its SIMD's whole register file):
//   * every workgroup j owns hidden units 4j..4j+3 of both LSTM cells = 16 gate rows each.  The rows
//     stay in registers as MFMA A operands for the whole launch (280 per lane, most of them in the
//     accumulation half of the file), split over the 4 waves by columns; a wave's partial 16 x 16 tile
//     meets the other waves' in LDS, wave 0 adds them, applies the cell update for its 4 units x 16
//     utterances (one per lane) and publishes the hidden values;
//   * what bounds a frame is how fast every CU can take in everybody's activations (each of the 256 CUs
//     needs every hidden value of every utterance), so an exchanged value is 4 bytes, not a {tag, value}
//     pair: an fp32 whose two lowest mantissa bits carry the generation of its slot (1 or 2, alternating
//     between the uses of a slot; 0 = never written).  The consumer uses the word as it is - tag bits
//     included, a relative perturbation of at most 2^-22, the same for everyone - and checks the tag
//     bits of what it loaded.  Vectors are laid out [k / 16][utterance][k % 16] and are loaded straight into
//     MFMA B operands: lane (utterance u = l & 15, q = l >> 4) fetches 16 bytes = the words
//     k = 16 i + 4 q .. + 3 of load i, its operands of four consecutive 4-deep steps (the weights are held
//     in the same column order); the 64 lanes of load i together read 1 KB of consecutive memory.  Every vector is
//     fetched once per workgroup and frame;
//   * the attention context is computed explicitly (the batch-1 kernel folds it into per-utterance
//     weight rows, which a batch cannot share): 8 "Q" workgroups per utterance each hold 16 attention
//     dimensions (query rows, location filter, location term) and an eighth of the memory columns;
//     they exchange partial energies among themselves, each runs the softmax, and each publishes its
//     columns of the context;
//   * prenet layer 1 is folded into the mel projection, W1 (Wp [dh ; ctx] + b) = (W1 Wp) [dh ; ctx] + W1 b,
//     so the "P" workgroups (16 prenet rows each) get the next frame's layer-1 rows from the decoder
//     hidden state and the context without waiting for a materialised frame; the "M" workgroups
//     (16 projection / stop-token rows each) write the frame beside the dependency chain.  Both read
//     their rows from LDS as MFMA A operands and share the B operands of the LSTM loads.
// A frame's chain: dh, ctx -> h1 (P) -> p (P) -> attention LSTM (all) -> ah -> energies (Q) -> partial
// energies (Q to Q) -> softmax, context (Q) -> ctx -> decoder LSTM (all) -> dh.  Products that nobody
// waits for (next frame's attention-LSTM columns, the recurrent columns) run in the shadows of the hops.
// Slots refill: when an utterance stops (stop token, or max_steps frames), its slot takes the next utterance of the
// call that fits the kernel - every workgroup sees the same stop tokens and keeps the same books, so all agree on
// who sits where without talking - and that utterance starts from the initial state at the next lockstep frame that
// is a multiple of four while the other fifteen carry on.  Columns of a tile never mix, and with starts on multiples
// of four an utterance's words carry the same tag bits whenever it runs (the tags have period four), so its result is
// bit for bit the same whatever its slot, its neighbours or its start.  With nobody left to take a slot the slot decodes on with its outputs
// switched off (its values stay finite); slots never filled run as copies of the first utterance; the launch ends
// when no slot holds a live utterance.
#include "taco_device.h"

#include <math.h>
#include <type_traits>
#include <vector>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) unsigned int bt_gword;

constexpr int NT = TACO_BT_THREADS;
constexpr int NW = NT / 64;
constexpr int NB = TACO_BATCH_NB;
static_assert(NW == 4, "softmax and query reductions are written for 4 waves");
static_assert(NB == 16, "one utterance per column of a 16 x 16 matrix-core tile");
// k capacities of the exchanged vectors: multiples of 64 = 4 waves x (16-byte loads of 4 words x 4 lane quarters)
constexpr int KP = TACO_BT_KP, KH = TACO_BT_KH, KC = TACO_BT_KC;
constexpr int NL_P = KP / 16 / NW;     // 16-byte loads per wave and lane: 4   (4 k-steps each)
constexpr int NL_H = KH / 16 / NW;     // 16
constexpr int NL_C = KC / 16 / NW;     // 9
constexpr int BURST = 4;               // loads checked together
constexpr int TIN = TACO_BT_TIN;       // positions per utterance
constexpr int HALO = 16;
constexpr int QD = 16;                 // attention dimensions per Q workgroup
constexpr int NQ = TACO_NQ_WG;         // Q workgroups per utterance (8)
constexpr int MEMW = TACO_BT_MEMW;     // memory columns per Q workgroup (68)
constexpr int CQ = MEMW / 4;           // column quads of a memory slice (17)
constexpr int CPG = 15;                // position groups of the context sum (15 x 17 threads)
constexpr int QT = (TIN / 16 + NW - 1) / NW;   // tiles of 16 positions per wave (wave w takes tiles w, w + 4, ...)
constexpr int QR = 16 / NW;            // query rows per wave (4)
constexpr int LOC_KSTEPS = 16;
constexpr int RP = TACO_BT_ROWS;       // rows per P / M workgroup (one MFMA tile)
constexpr int W1_KSTEPS = (TACO_MAX_MEL + 3) / 4;    // 20
static_assert(TACO_MAX_K <= 32 && LOC_KSTEPS == 16, "location taps: 8 steps of 4 for each of the two weight arrays");
static_assert(QR == 4 && QT % 2 == 0 && CPG * CQ <= NT && MEMW % 4 == 0 && NQ * MEMW >= 544, "Q thread maps");
static_assert(NL_H % BURST == 0 && NL_P == BURST, "burst structure");

// ---- LDS carve (float offsets) ----
constexpr int L_RED = 0;                         // [4 waves][64 lanes][4]: partial tiles of an LSTM stage
constexpr int L_BIAS = L_RED + NW * 256;         // [2 cells][4 units][4 gates]
constexpr int L_STOP = L_BIAS + 32;              // stop tokens of the previous frame [16], failure flag at +16
constexpr int L_ROLE = L_STOP + 32;
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
// Q: query rows [16][1024], memory slice [positions][68], location filter [16][64], hidden state of my utterance /
//    gathered partial energies / context partial sums (one area, used one after the other), weights and
//    cumulative weights with halos, query values, softmax scratch, location term [positions][16]
constexpr int LQ_ROWS = L_ROLE;
constexpr int LQ_MEM = LQ_ROWS + QD * KH;
constexpr int LQ_E = LQ_MEM + TIN * MEMW;
constexpr int LQ_XPE = LQ_E + LOC_KSTEPS * 64;
constexpr int LQ_AW = LQ_XPE + NQ * TIN;
constexpr int LQ_CUM = LQ_AW + TIN + 2 * HALO;
constexpr int LQ_Q = LQ_CUM + TIN + 2 * HALO;
constexpr int LQ_SRED = LQ_Q + QD;
constexpr int LQ_U = LQ_SRED + 32;
constexpr int LQ_END = LQ_U + TIN * QD;
static_assert(NQ * TIN >= KH && NQ * TIN >= CPG * MEMW, "shared Q scratch area");
constexpr int BT_CARRY_R = 10;                         // acc_a (4), acc_dn (4), c_a, c_d
constexpr int BT_CARRY_Q = 2 * (TIN + 2 * HALO);       // LQ_AW .. the end of LQ_CUM
static_assert(LQ_CUM == LQ_AW + TIN + 2 * HALO && LQ_Q == LQ_CUM + TIN + 2 * HALO, "the weights and cumulative weights are one block of LDS");
constexpr int L_TOTAL = LP_END > LQ_END ? (LP_END > LM_END ? LP_END : LM_END) : (LQ_END > LM_END ? LQ_END : LM_END);
static_assert(L_TOTAL * 4 <= 160 * 1024, "batched decoder LDS carve exceeds a CU's 160 KiB");

struct BatArgs {
  int P, M, Ra, Rd, A, K, n_mel, T_in, n_wg;
  int B;                   // utterances of the call; those of at most TACO_BT_TIN symbols are decoded here, in index order
  int t_limit;             // lockstep frames after which the launch ends whatever its books say (never reached)
  const int32_t* lens;     // device, or NULL = T_in symbols each
  const float *prenet_w1, *prenet_w2, *wf_h, *wf_c, *wf_b, *att_w_ih, *att_w_hh, *att_b_ih, *att_b_hh, *query_w, *v_w,
      *loc_fold, *dec_w_ih, *dec_w_hh, *dec_b_ih, *dec_b_hh, *proj_w, *proj_b, *gate_w, *gate_b;
  const float4* wstream;   // attention-LSTM input columns re-packed per (workgroup, wave, load, lane): see taco_bt_pack_kernel
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
  uint32_t* words;         // hand-off memory, two parity blocks
  unsigned words_bytes;
  taco_bat_layout lay;
  taco_dec_ctl* ctl;
  // Leaving utterances unfinished and taking them up again in a later launch (see "leaving and resuming" in the kernel)
  const int32_t* resume;   // device [B] or NULL: the slot whose saved state utterance i continues, -1 = it starts anew
  int leave;               // 1: end the launch once a slot has no successor, the running utterances' state goes to `carry_*`
  float* carry_r;          // [workgroups][waves][64 lanes][BT_CARRY_R]: carried partial sums and cell states of every column
  float* carry_q;          // [NB * NQ][BT_CARRY_Q]: a Q workgroup's attention weights and cumulative weights (with halos)
  const uint32_t* carry_p; // [NB * KP] words, exchange layout: the prenet output the left utterances were about to consume
  int32_t* carry_ctl;      // [NB]: frames a slot's utterance had produced when it was left
  int32_t* slot_out;       // device [B]: the slot an utterance was left in (entries of finished utterances are not touched)
  unsigned long long* frame_clock;
  int frame_clock_cap;
#ifdef TACO_BT_STAMPS
  unsigned long long* stamps;     // diagnostic build only: [workgroups][BT_STAMP_FRAMES][BT_STAMP_POINTS]
#endif
};

#ifdef TACO_BT_STAMPS
constexpr int BT_STAMP_FRAMES = 256, BT_STAMP_POINTS = 32;
#define BT_STAMP(k)                                                                                         \
  do {                                                                                                      \
    if (tid == 0 && stamp_wg >= 0 && t < BT_STAMP_FRAMES)                                                   \
      a.stamps[((size_t)stamp_wg * BT_STAMP_FRAMES + t) * BT_STAMP_POINTS + (k)] = wall_clock64();          \
  } while (0)
#define BT_STAMP_W(k, w)                                                                                   \
  do {                                                                                                      \
    if (tid == 64 * (w) && stamp_wg >= 0 && t < BT_STAMP_FRAMES)                                            \
      a.stamps[((size_t)stamp_wg * BT_STAMP_FRAMES + t) * BT_STAMP_POINTS + (k)] = wall_clock64();          \
  } while (0)
#else
#define BT_STAMP(k) do {} while (0)
#define BT_STAMP_W(k, w) do {} while (0)
#endif

// An LDS index behind a compiler barrier.  The carve of a Q workgroup reaches far past the 64 KB an LDS instruction's
// immediate offset covers, so the compiler forms one address register per access, finds them all loop-invariant and
// hoists them out of the frame loop - dozens of registers this kernel does not have (they ended up in scratch memory).
// An index made opaque inside the frame is recomputed there (one or two vector operations) and the accesses near it
// become immediate offsets from it.
#define BT_OPQ(expr) ({ int v_ = (expr); asm volatile("" : "+v"(v_)); v_; })
#define BT_SB() __builtin_amdgcn_sched_barrier(0)

// ---- tagged words ----
__device__ __forceinline__ unsigned bt_tagged(float value, unsigned tag) { return (__float_as_uint(value) & ~3u) | tag; }
// One value to all R replicas of a vector (replica stride `cap` words): atomic exchanges whose result is dropped,
// carried out at the memory side.
template <int R>
__device__ __forceinline__ void bt_publish(uint32_t* base, int cap, int idx, unsigned tag, float value) {
  const unsigned word = bt_tagged(value, tag);
#pragma unroll
  for (int r = 0; r < R; ++r) (void)__hip_atomic_exchange((bt_gword*)base + (size_t)r * cap + idx, word, TACO_RLX_AGENT);
}

// Four consecutive words (16-byte aligned word index widx) to all R replicas with write-through 16-byte stores: every
// word carries its own tag, so nothing depends on the 16 bytes landing together.  (One 16-byte store per four values
// instead of four memory-side atomics: a frame publishes some 80 000 words chip-wide.)
template <int R>
__device__ __forceinline__ void bt_store4(__amdgpu_buffer_rsrc_t rs, unsigned widx, int cap, unsigned tag, float v0, float v1, float v2,
                                          float v3) {
  const u32x4 w = {bt_tagged(v0, tag), bt_tagged(v1, tag), bt_tagged(v2, tag), bt_tagged(v3, tag)};
#pragma unroll
  for (int r = 0; r < R; ++r) __builtin_amdgcn_raw_buffer_store_b128(w, rs, (widx + (unsigned)(r * cap)) * 4u, 0, 16 /* sc1 */);
}
// A hidden value per lane (lane = utterance + 16 * unit) -> the lanes of unit 0 collect their utterance's four units
// and store them as one 16-byte piece at word index `widx` (= vector offset + utterance * K + 4 * workgroup).
template <int R>
__device__ __forceinline__ void bt_publish_units(__amdgpu_buffer_rsrc_t rs, unsigned widx, int cap, unsigned tag, float hv, int lane) {
  // rows of 16 lanes [r0 r1 r2 r3] = the four units: v_permlane16_swap(x, x) leaves [r1 r1 r3 r3] in its second result,
  // v_permlane32_swap(x, x) [r2 r3 r2 r3], and the 16-swap of that [r3 r3 r3 r3] - three vector instructions where
  // three __shfl would be three round trips through the LDS crossbar, on the path every other workgroup waits for
  const unsigned x = __float_as_uint(hv);
  const auto s16 = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  const auto s32 = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  const auto s48 = __builtin_amdgcn_permlane16_swap(s32[1], s32[1], false, false);
  if (lane < 16) bt_store4<R>(rs, widx, cap, tag, hv, __uint_as_float(s16[1]), __uint_as_float(s32[1]), __uint_as_float(s48[1]));
}

// B operands come straight from the hand-off words.  bt_issue starts N 16-byte loads (word index widx + 16 i),
// bt_check looks at the tag bits of loads [i0, i0 + CNT) and, while one is not there yet, fetches those loads again
// after a short sleep (after the nap the values are there at the first attempt as a rule).  All loads of a vector are
// issued at once and checked burst by burst, so the memory round trips run beside the matrix products.
template <int N>
__device__ __forceinline__ void bt_issue(__amdgpu_buffer_rsrc_t rs, unsigned widx, u32x4 (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (widx + 256u * i) * 4u, 0, 16 /* sc1 */);
}
template <int N, int I0, int CNT>
__device__ __forceinline__ void bt_check(__amdgpu_buffer_rsrc_t rs, unsigned widx, unsigned tag, u32x4 (&v)[N], taco_wait_ctx& ctx,
                                         unsigned info) {
  unsigned spins = 0;
  for (;;) {
    unsigned bad = 0;
#pragma unroll
    for (int i = I0; i < I0 + CNT; ++i) bad |= (v[i].x ^ tag) | (v[i].y ^ tag) | (v[i].z ^ tag) | (v[i].w ^ tag);
    if (__all((bad & 3u) == 0u) || ctx.failed) break;
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
#pragma unroll
    for (int i = I0; i < I0 + CNT; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (widx + 256u * i) * 4u, 0, 16);
  }
}
#define BT_F(x) __uint_as_float(x)
// Where value k of utterance slot u sits inside an exchanged vector: [k / 16][slot][k % 16].  One load instruction of a
// wave (16 slots x 4 quarter-lanes x 16 bytes) then covers 1 KB of consecutive memory - eight whole cache lines, each
// asked for once - instead of sixteen half lines that the next instruction asks for again.
__device__ __forceinline__ constexpr unsigned bt_word(int k, int u) { return (unsigned)(((k >> 4) * NB + u) * 16 + (k & 15)); }

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

// The naps before the first fetch of a vector are steered as in the batch-1 kernel (value not there at the first look:
// longer next frame; all there: a little shorter), with a step ratio of 8 : 1 - here a look that comes too early costs a
// second fetch of a burst, less than a nap that is too long costs every frame (measured on the batch of 64: 4 : 1 -0.4 %,
// 8 : 1 +1.2 %, 16 : 1 = the batch-1 kernel's ratio 0, 32 : 1 -3.3 %, 64 : 1 -7.7 %; no naps at all -15 %).
#ifndef BT_NAP_UP_STEP
#define BT_NAP_UP_STEP 8
#endif
constexpr int BT_NAP_UP = BT_NAP_UP_STEP, BT_NAP_DOWN = 1;
// Extra time (units of s_sleep 1, about 64 clocks) that workgroups not on the frame's chain keep away from a vector
#ifndef BT_DELAY_AH
#define BT_DELAY_AH 50
#endif
// Whether a vector's nap is steered by all its bursts (1: a look is early if ANY piece of the vector was not there) or by
// the first one only
#ifndef BT_NAPALL_AH
#define BT_NAPALL_AH 1
#endif
#ifndef BT_NAPALL_CTX
#define BT_NAPALL_CTX 1
#endif
#ifndef BT_NAPALL_DH
#define BT_NAPALL_DH 1
#endif
#ifndef BT_DELAY_DH
#define BT_DELAY_DH 0
#endif
// Diagnostic (timing only, results unchanged): workgroups of role BT_PROBE_ROLE (0 plain, 1 P, 2 M, 3 Q) lose BT_PROBE_N
// naps of 64 clocks at point BT_PROBE_AT of every frame.  How much of that the frame gets longer by says whether the
// point lies on the frame's critical path for that role (scripts/probe_delays.sh).
#ifndef BT_PROBE_AT
#define BT_PROBE_AT 0
#endif
#ifndef BT_PROBE_ROLE
#define BT_PROBE_ROLE 0
#endif
#ifndef BT_PROBE_N
#define BT_PROBE_N 16
#endif
#define BT_PROBE(point)                                                                              \
  if constexpr (BT_PROBE_AT == (point) && ROLE == BT_PROBE_ROLE) {                                   \
    for (int i_ = 0; i_ < BT_PROBE_N; ++i_) __builtin_amdgcn_s_sleep(1);                             \
  }
__device__ __forceinline__ int bt_nap_adjust(int n, int spun) {
  const bool any = __builtin_amdgcn_readfirstlane(__any(spun) ? 1 : 0) != 0;
  return any ? min(n + BT_NAP_UP, TACO_NAP_MAX) : max(n - BT_NAP_DOWN, 0);
}

__device__ __forceinline__ void bt_nap_begin(int nap, taco_wait_ctx& ctx) {
  taco_nap(nap);
  ctx.spun = 0;
  asm volatile("" ::: "memory");
}

// Four fp32 values -> four bf16 (round to nearest even), the operand of one v_mfma_f32_16x16x16_bf16: lane (l & 15, l >> 4)
// supplies k = 4 (l >> 4) .. + 3 of a 16-deep step - the four consecutive k a lane's 16-byte load holds anyway.
__device__ __forceinline__ unsigned bt_pk2(float a, float b) {
  const f32x2 f = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
__device__ __forceinline__ s16x4 bt_pack(float a, float b, float c, float d) {
  const u32x2 r = {bt_pk2(a, b), bt_pk2(c, d)};
  return __builtin_bit_cast(s16x4, r);
}
__device__ __forceinline__ s16x4 bt_pack(u32x4 v) { return bt_pack(BT_F(v.x), BT_F(v.y), BT_F(v.z), BT_F(v.w)); }

// The LSTM gate rows a lane keeps for the whole launch, NL 16-byte loads' worth of columns: 4 NL fp32 A operands of
// 4-deep steps, or (BF) NL bf16 A operands of 16-deep steps - the same columns in half the registers.
template <bool BF, int NL>
struct BtRows {
  float f[BF ? 1 : 4 * NL];
  s16x4 h[BF ? NL : 1];
};

// 4 k-steps of one load: four exact-f32 products on two independent accumulation chains, or (BF) one bf16 product
// (16 deep, a quarter of the issue slots and an eighth of the matrix-core time).  W: BtRows; W4: the four weights as a float4.
#define BT_MFMA4F(W, i, V, accA, accB)                                                             \
  accA = __builtin_amdgcn_mfma_f32_16x16x4f32(W[4 * (i) + 0], BT_F(V.x), accA, 0, 0, 0);           \
  accB = __builtin_amdgcn_mfma_f32_16x16x4f32(W[4 * (i) + 1], BT_F(V.y), accB, 0, 0, 0);           \
  accA = __builtin_amdgcn_mfma_f32_16x16x4f32(W[4 * (i) + 2], BT_F(V.z), accA, 0, 0, 0);           \
  accB = __builtin_amdgcn_mfma_f32_16x16x4f32(W[4 * (i) + 3], BT_F(V.w), accB, 0, 0, 0);
#define BT_MFMA4(W, i, V, accA, accB)                                                              \
  if constexpr (BF) {                                                                              \
    if ((i) & 1) accB = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((W).h[i], bt_pack(V), accB, 0, 0, 0);      \
    else accA = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((W).h[i], bt_pack(V), accA, 0, 0, 0);    \
  } else {                                                                                         \
    BT_MFMA4F((W).f, i, V, accA, accB)                                                             \
  }
#define BT_MFMA4S(W4, V, accA, accB)                                                               \
  if constexpr (BF) {                                                                              \
    accA = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bt_pack((W4).x, (W4).y, (W4).z, (W4).w), bt_pack(V), accA, 0, 0, 0); \
  } else {                                                                                         \
    const float w4_[4] = {(W4).x, (W4).y, (W4).z, (W4).w};                                         \
    BT_MFMA4F(w4_, 0, V, accA, accB)                                                               \
  }

template <bool BF>
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
  const int q_off = n_p_wg + n_m_wg;
  const bool is_p = j < n_p_wg;
  const bool is_m = j >= n_p_wg && j < q_off;
  const int q_id = j - q_off;
  const bool is_q = q_id >= 0 && q_id < NB * NQ;          // empty slots run as copies of utterance 0, unwritten
  const int q_slot = is_q ? q_id / NQ : 0;
  const int dg = is_q ? q_id % NQ : 0;
  const int q_base = dg * qd;
  const int q_cnt = is_q ? max(0, min(qd, A - q_base)) : 0;
  const int cw = MEMW;                                    // memory columns per Q workgroup (the last ones may lie beyond M: zeros)
  const int c0 = dg * cw;
  const int c_cnt = is_q ? max(0, min(cw, M - c0)) : 0;    // memory columns I hold (I publish cw: the rest are zeros)
  const bool forced = a.forced_mel != nullptr;
  const int pad = (K - 1) / 2;

  // ---------------- one-time loads: LSTM rows as A operands in registers ----------------
  // (228 of the 280 per lane.  The attention LSTM's prenet and context columns - 52 - are read from a re-packed copy in
  // global memory every frame instead, L2-resident and asked for well before they are used: with them the rows would not
  // leave the vector half of the register file to everything else, and the compiler spilled rows to scratch memory.)
  // A operand of lane l: row (l & 15) = 4 * unit + gate of the workgroup's 16 rows, column k of step s of a vector's
  // wave slice: k = 16 * (wave * loads_per_wave + (s >> 2)) + 4 * (l >> 4) + (s & 3)
  const int row16 = lane & 15;
  const int hid_a = j * 4 + (row16 >> 2), gate_a = row16 & 3;
  const bool own_a = hid_a < Ra, own_d = hid_a < Rd;
  const size_t ra = (size_t)gate_a * Ra + hid_a, rd = (size_t)gate_a * Rd + hid_a;
  BtRows<BF, NL_H> wa_h, wd_a, wd_h;
  BtRows<BF, NL_C> wd_c;
#define BT_K(NL, s) (16 * (wave * (NL) + ((s) >> 2)) + 4 * kq + ((s) & 3))
  if constexpr (BF) {
#pragma unroll
    for (int li = 0; li < NL_C; ++li) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = BT_K(NL_C, 4 * li + e);
        v[e] = (own_d && k < M) ? a.dec_w_ih[rd * (Ra + M) + Ra + k] : 0.0f;
      }
      wd_c.h[li] = bt_pack(v[0], v[1], v[2], v[3]);
      if ((li & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int li = 0; li < NL_H; ++li) {
      float va[4], vd[4], vh[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = BT_K(NL_H, 4 * li + e);
        va[e] = (own_a && k < Ra) ? a.att_w_hh[ra * Ra + k] : 0.0f;
        vd[e] = (own_d && k < Ra) ? a.dec_w_ih[rd * (Ra + M) + k] : 0.0f;
        vh[e] = (own_d && k < Rd) ? a.dec_w_hh[rd * Rd + k] : 0.0f;
      }
      wa_h.h[li] = bt_pack(va[0], va[1], va[2], va[3]);
      wd_a.h[li] = bt_pack(vd[0], vd[1], vd[2], vd[3]);
      wd_h.h[li] = bt_pack(vh[0], vh[1], vh[2], vh[3]);
      if ((li & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4 * NL_C; ++i) {
      const int k = BT_K(NL_C, i);
      wd_c.f[i] = (own_d && k < M) ? a.dec_w_ih[rd * (Ra + M) + Ra + k] : 0.0f;
      if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4 * NL_H; ++i) {
      const int k = BT_K(NL_H, i);
      wa_h.f[i] = (own_a && k < Ra) ? a.att_w_hh[ra * Ra + k] : 0.0f;
      wd_a.f[i] = (own_d && k < Ra) ? a.dec_w_ih[rd * (Ra + M) + k] : 0.0f;
      wd_h.f[i] = (own_d && k < Rd) ? a.dec_w_hh[rd * Rd + k] : 0.0f;
      if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef BT_K

  // ---------------- who sits where ----------------
  // Utterances are taken in index order, those too long for this kernel skipped (the host decodes them with the
  // batch-1 kernel).  `cursor` and the books below move only on what the stop tokens say, which every workgroup reads
  // alike: no word is exchanged about the assignment.
  int cursor = 0;
  auto next_utterance = [&]() -> int {
    while (cursor < a.B) {
      const int i = cursor++;
      if ((a.lens ? a.lens[i] : a.T_in) <= TIN && !(a.resume && a.resume[i] >= 0)) return i;
    }
    return -1;
  };
  unsigned alive = 0;                          // slots holding an utterance that still produces frames
  unsigned waiting = 0;                        // slots whose next utterance is booked and starts at the coming multiple of four
  int utt_u = -1, t0_u = 0;                    // my column's slot: its utterance, and the lockstep frame that one started in
  // Leaving and resuming.  A launch told to `leave` ends as soon as a slot has no successor - at a lockstep frame that is
  // a multiple of four, so that every running utterance has produced a multiple of four frames plus its start's - and
  // saves what its running utterances carry from one frame to the next: per column the partial sums over the previous
  // frame's vectors and the cell states, per Q workgroup the weights and cumulative weights; the prenet output they
  // were about to consume stays in the exchange buffer (the host copies it aside).  A later launch that is given
  // `resume` puts those utterances back into their slots at its frame 0 with a start frame of minus the frames they
  // have: own frame indices, tag bits and arithmetic continue as if the launch had never ended.
  unsigned carried = 0;
  if (a.resume) {
    for (int i = 0; i < a.B; ++i) {
      const int sl = a.resume[i];
      if (sl >= 0 && sl < NB) {
        carried |= 1u << sl;
        if (utt == sl) { utt_u = i; t0_u = -a.carry_ctl[sl]; }
      }
    }
  }
  alive = carried;
  for (int sl = 0; sl < NB; ++sl) {
    if ((carried >> sl) & 1u) continue;
    const int nu = next_utterance();
    if (nu >= 0) alive |= 1u << sl;
    if (utt == sl) utt_u = nu;
  }
  {
    // slots never filled run as copies of the first running utterance, unwritten
    const int first = alive ? __builtin_amdgcn_readlane(utt_u, __builtin_ctz(alive)) : 0;
    if (utt_u < 0) utt_u = first;
  }
  const bool col_carried = (carried >> utt) & 1u;
  int q_b = __builtin_amdgcn_readlane(utt_u, q_slot);     // Q: my slot's utterance, its start frame, its length
  int q_t0 = __builtin_amdgcn_readlane(t0_u, q_slot);
  int len = 0;
  if (tid < 16) {     // biases as [unit][gate]
    const int u = tid >> 2, gt = tid & 3, hu = j * 4 + u;
    lds[L_BIAS + tid] = hu < Ra ? a.att_b_ih[gt * Ra + hu] + a.att_b_hh[gt * Ra + hu] : 0.0f;
    lds[L_BIAS + 16 + tid] = hu < Rd ? a.dec_b_ih[gt * Rd + hu] + a.dec_b_hh[gt * Rd + hu] : 0.0f;
  }

  // ---------------- one-time loads of the roles ----------------
  // A operands kept in LDS as [global k-step][lane]: k-step ks = 4 * load + e, lane l -> row l & 15,
  // column k = 16 * load + 4 * (l >> 4) + e.  Global k-steps of [dh ; ctx]: dh 0..255, ctx 256..399.
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
          const int k = 16 * (ks >> 2) + 4 * q4 + (ks & 3);
          if (k < Rd) v = is_p ? a.wf_h[(size_t)r * Rd + k] : (r < NMEL ? a.proj_w[(size_t)r * n_hc + k] : a.gate_w[k]);
        } else {
          const int kc = ks - KH / 4;
          const int k = 16 * (kc >> 2) + 4 * q4 + (kc & 3);
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
      const int r = j * RP + (l & 15), k = 16 * (ks >> 2) + 4 * (l >> 4) + (ks & 3);
      lds[LP_W2 + e] = (r < P && k < P) ? a.prenet_w2[(size_t)r * P + k] : 0.0f;
    }
    _Pragma("unroll 1") for (int e = tid; e < W1_KSTEPS * 64; e += NT) {
      const int ks = e >> 6, l = e & 63;
      const int r = j * RP + (l & 15), k = 16 * (ks >> 2) + 4 * (l >> 4) + (ks & 3);
      lds[LP_W1 + e] = (r < P && k < NMEL) ? a.prenet_w1[(size_t)r * NMEL + k] : 0.0f;
    }
  }
  if constexpr (BF) {
    // P: the folded layer-1 rows and the layer-2 rows become bf16 A operands of 16-deep steps, [load][lane] x 8 bytes,
    // in place over their fp32 copies (a load = 4 k-steps; the bf16 copy of loads 0 .. L ends before the fp32 words of
    // load L + 1 begin).  The projection / stop-token rows of the M workgroups stay fp32: they make the frame itself.
    if (is_p) {
      auto to_bf16 = [&](int base, int n_loads) {
        u32x2* out = reinterpret_cast<u32x2*>(lds + base);
        _Pragma("unroll 1") for (int e0 = 0; e0 < n_loads * 64; e0 += NT) {
          const int e = e0 + tid;
          float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          if (e < n_loads * 64) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = lds[base + ((e >> 6) * 4 + q) * 64 + (e & 63)];
          }
          __syncthreads();       // all four words of every element of this pass are read before one is overwritten
          if (e < n_loads * 64) out[e] = __builtin_bit_cast(u32x2, bt_pack(v[0], v[1], v[2], v[3]));
          __syncthreads();
        }
      };
      __syncthreads();
      to_bf16(LP_WF, KS_HC / 4);
      to_bf16(LP_W2, KP / 16);
    }
  }
  float vq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (is_q) {
    _Pragma("unroll 1") for (int e = tid; e < q_cnt * Ra; e += NT) {
      const int r = e / Ra, c = e % Ra;
      lds[LQ_ROWS + r * KH + c] = a.query_w[(size_t)(q_base + r) * Ra + c];
    }
    _Pragma("unroll 1") for (int i = tid; i < LOC_KSTEPS * 64; i += NT) {
      const int m = i >> 6, l = i & 63;
      const int ck = 4 * m + (l >> 4), dim = q_base + (l & 15);       // tap 32 c + k, see update_location_term
      const int c = ck >> 5, k = ck & 31;
      lds[LQ_E + i] = (k < K && (l & 15) < q_cnt) ? a.loc_fold[((size_t)dim * 2 + c) * K + k] : 0.0f;
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
  int nap_h1 = 0, nap_p = 0, nap_ahq = 0, nap_ah = 0, nap_ctx = 0, nap_dh = 0;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.words, 0, (int)a.words_bytes, 0x00020000);

  // Q: the location term of my tiles, U = processed memory + sum over taps of E * {previous, cumulative} weights.
  // Lane (position col = l & 15, kq): dimensions 4 kq .. + 3 of the tile's 16 x 16 result; the B operand of 4-tap step m
  // is weight (or cumulative weight) number 4 m + kq of the 2K taps, read from the halo-padded LDS copies.  The
  // processed memory is read again every frame (L2-resident, asked for before the products, added after them):
  // keeping it in registers would cost 16 of them for the whole launch.
  const float* pm_lane = a.pm + q_base + 4 * kq;      // + (utterance * T_in + position) * A: set when the slot is filled
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
  // Q: what belongs to the utterance in my slot - its memory columns, no attention weights yet, the location term of
  // zero weights (= the processed memory).  Runs at the start and whenever the slot takes a new utterance; the caller
  // has a barrier behind it, and one in front (nobody still reads the old utterance's copies).
  int len16 = 0, pe_n = 0;
  auto fill_slot = [&]() {
    len = a.lens ? min(a.lens[q_b], a.T_in) : a.T_in;
    len16 = (len + 15) & ~15;
    pe_n = max(0, min(2, (len16 / 4 - (tid & 31) + 31) / 32));      // 16-byte loads (4 positions each) of my group: l, l + 32
    pm_lane = a.pm + (size_t)q_b * a.T_in * A + q_base + 4 * kq;
    const float* mem_b = a.memory + (size_t)q_b * a.T_in * M;
    _Pragma("unroll 1") for (int e = tid; e < len * MEMW; e += NT) {
      const int s = e / MEMW, cc = e % MEMW;
      lds[LQ_MEM + e] = cc < c_cnt ? mem_b[(size_t)s * M + c0 + cc] : 0.0f;
    }
    _Pragma("unroll 1") for (int i = tid; i < 2 * (TIN + 2 * HALO); i += NT) lds[LQ_AW + i] = 0.0f;     // weights and cumulative weights
#pragma unroll
    for (int ti = 0; ti < QT; ++ti) {
      const int s = 16 * (wave + NW * ti) + utt;
      if (s < TIN) {
        float pm0[4];
        load_pm(s, pm0);
        *reinterpret_cast<float4*>(lds + LQ_U + s * QD + 4 * kq) = make_float4(pm0[0], pm0[1], pm0[2], pm0[3]);
      }
    }
  };
  if (is_q) fill_slot();
  __syncthreads();
  const bool q_carried = is_q && ((carried >> q_slot) & 1u);
  if (q_carried) {
    // a resumed utterance: the weights it had when it was left (the location term follows below)
    const float* src = a.carry_q + (size_t)q_id * BT_CARRY_Q;
    for (int i = tid; i < BT_CARRY_Q; i += NT) lds[LQ_AW + i] = src[i];
  }
  __syncthreads();
  // (lane_o: the lane number behind a compiler barrier, taken anew every frame, see BT_OPQ.)
  // Taps are numbered 32 c + k (c = 0: previous weights, 1: cumulative weights; k >= K carries a zero filter value), so
  // that step m = 8 c + k / 4 reads weight number k of array c at a compile-time distance from one per-lane address:
  // the loop has no address arithmetic, only LDS reads at immediate offsets beside the products.
  auto update_location_term = [&](int lane_o) {
    const int col = lane_o & 15, kqo = lane_o >> 4;
    const float* x_lane = lds + BT_OPQ(LQ_AW + HALO + col + kqo - pad);     // tap kq of the previous weights at position `col`
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
        const float* xa = x_lane + s0;
        const float* xb = x_lane + s1c;
        f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
        float ev[2][4], x0[2][4], x1[2][4];      // operands are read four steps ahead of their products
        auto read_group = [&](int gq, float (&e)[4], float (&x_a)[4], float (&x_b)[4]) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int m = 4 * gq + i;
            const int off = (m >> 3) * (LQ_CUM - LQ_AW) + ((4 * m) & 31);
            e[i] = e_lane[m * 64];
            x_a[i] = xa[off];
            x_b[i] = xb[off];
          }
        };
        read_group(0, ev[0], x0[0], x1[0]);
#pragma unroll
        for (int gq = 0; gq < LOC_KSTEPS / 4; ++gq) {
          if (gq + 1 < LOC_KSTEPS / 4) read_group(gq + 1, ev[(gq + 1) & 1], x0[(gq + 1) & 1], x1[(gq + 1) & 1]);
          BT_SB();
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ev[gq & 1][i], x0[gq & 1][i], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ev[gq & 1][i], x1[gq & 1][i], acc1, 0, 0, 0);
          }
          BT_SB();
        }
        *reinterpret_cast<float4*>(u_lane + s0 * QD) = make_float4(acc0[0] + pm0[0], acc0[1] + pm0[1], acc0[2] + pm0[2], acc0[3] + pm0[3]);
        if (s1 < len)
          *reinterpret_cast<float4*>(u_lane + s1 * QD) = make_float4(acc1[0] + pm1[0], acc1[1] + pm1[1], acc1[2] + pm1[2], acc1[3] + pm1[3]);
      }
    }
  };
  // Q: partial energies are gathered by thread (group tid >> 5, l = tid & 31): pairs of granules l, l + 32, l + 64, l + 96
  // of that group's positions
  const unsigned pe_src0 = (unsigned)(a.lay.pe + (q_slot * NQ + (tid >> 5)) * TIN + 4 * (tid & 31));   // word index

  f32x4 acc_a = {0.0f, 0.0f, 0.0f, 0.0f};     // attention-LSTM partial over [context ; hidden] of the previous frame
  f32x4 acc_dn = {0.0f, 0.0f, 0.0f, 0.0f};    // decoder-LSTM partial over the previous decoder hidden state
  float c_a = 0.0f, c_d = 0.0f;               // cell states (wave 0: lane = (unit kq, utterance utt))
  int t = 0;
  int left = 0;                               // the launch ended with utterances left unfinished
  float* carry_mine = a.carry_r + ((size_t)(j * NW + wave) * 64 + lane) * BT_CARRY_R;
  if (q_carried) update_location_term(lane);  // (the barrier in front of the frame loop's first energies covers it)
  if (col_carried) {
    acc_a = f32x4{carry_mine[0], carry_mine[1], carry_mine[2], carry_mine[3]};
    acc_dn = f32x4{carry_mine[4], carry_mine[5], carry_mine[6], carry_mine[7]};
    c_a = carry_mine[8];
    c_d = carry_mine[9];
  }
  // word index of this lane's first load in each vector, replica `rep`
  const unsigned w_lane_h1 = (unsigned)(a.lay.h1 + rep * a.lay.h1_cap) + bt_word(16 * (wave * NL_P) + 4 * kq, utt);
  const unsigned w_lane_p = (unsigned)(a.lay.p + rep * a.lay.p_cap) + bt_word(16 * (wave * NL_P) + 4 * kq, utt);
  const unsigned w_lane_ah = (unsigned)(a.lay.ah + rep * a.lay.ah_cap) + bt_word(16 * (wave * NL_H) + 4 * kq, utt);
  const unsigned w_lane_dh = (unsigned)(a.lay.dh + rep * a.lay.dh_cap) + bt_word(16 * (wave * NL_H) + 4 * kq, utt);
  const unsigned w_lane_c = (unsigned)(a.lay.ctx + rep * a.lay.ctx_cap) + bt_word(16 * (wave * NL_C) + 4 * kq, utt);
  const float4* ws_lane = a.wstream + ((size_t)(j * NW + wave) * (NL_P + NL_C)) * 64 + lane;   // load i: ws_lane[i * 64]
  // (BF: the attention LSTM's prenet and context columns fit the registers too - 26 per lane - and are not streamed)
  s16x4 wa_p[BF ? NL_P : 1], wa_c[BF ? NL_C : 1];
  if constexpr (BF) {
#pragma unroll
    for (int i = 0; i < NL_P; ++i) { const float4 w = ws_lane[i * 64]; wa_p[i] = bt_pack(w.x, w.y, w.z, w.w); }
#pragma unroll
    for (int i = 0; i < NL_C; ++i) { const float4 w = ws_lane[(NL_P + i) * 64]; wa_c[i] = bt_pack(w.x, w.y, w.z, w.w); }
  }
  unsigned keep_next = 0;                      // P, wave 0: keep bytes of my 4 rows, layer 0 (low) / layer 1 of the coming frame
  unsigned keep_l1 = 0, keep_l2 = 0;
  if (is_p && wave == 0) {
    const int r0 = j * RP + 4 * kq;
    if (r0 < P) keep_l2 = *reinterpret_cast<const unsigned*>(a.masks + P + r0);     // frame 0, layer 2
  }

#ifdef TACO_BT_STAMPS
  const int stamp_wg = j;
#endif
  // The frame loop exists once per role (plain LSTM rows only, P, M, Q): a workgroup's role never changes, and a loop
  // that knows its role at compile time keeps only that role's values in registers - the one loop for all roles needed
  // the union of them and spilled.
  auto frame_loop = [&](auto role_tag) __attribute__((always_inline)) {
  constexpr int ROLE = decltype(role_tag)::value;
  constexpr bool is_p = ROLE == 1, is_m = ROLE == 2, is_q = ROLE == 3;
  for (;; ++t) {
    const unsigned ep = (unsigned)t + 1u;                                  // epoch of the 8-byte granules
    const unsigned tag = (((unsigned)t >> 1) & 1u) + 1u;                   // generation of this frame's words
    const unsigned par = (unsigned)(t & 1) * (unsigned)a.lay.per_parity;   // word offset of this frame's parity block
    uint32_t* gw = a.words + par;
    const int lane_f = BT_OPQ(lane);            // see BT_OPQ: addresses derived from these are recomputed every frame
    const int tid_f = lane_f + 64 * wave;
    const int utt_f = lane_f & 15, kq_f = lane_f >> 4;

    // ============ P: prenet layer 2 of this frame (heads the chain; does not wait for the stop test) ============
    if (is_p) {
      f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
      // keep bytes of my column's next frame (each utterance counts its own frames), asked for a frame ahead (a
      // first-touch line comes from HBM)
      unsigned kn1 = 0, kn2 = 0;
      if (wave == 0) {
        const int r0 = min(j * RP + 4 * kq, P - 4);
        const uint8_t* src = a.masks + (size_t)max(0, min(t - t0_u + 1, a.max_steps - 1)) * 2 * P + r0;
        kn1 = *reinterpret_cast<const unsigned*>(src);
        kn2 = *reinterpret_cast<const unsigned*>(src + P);
      }
      if (t > 0) {
        u32x4 v[NL_P];
        const float* w2 = lds + BT_OPQ(LP_W2 + wave * (4 * NL_P) * 64 + lane_f);
        bt_nap_begin(nap_h1, wctx);
        bt_issue<NL_P>(rs, par + w_lane_h1, v);
        float aop[BF ? 1 : 4 * NL_P];
        u32x2 aoph[BF ? NL_P : 1];
        if constexpr (BF) {
          const u32x2* w2h = reinterpret_cast<const u32x2*>(lds + LP_W2) + BT_OPQ(wave * NL_P * 64 + lane_f);
#pragma unroll
          for (int i = 0; i < NL_P; ++i) aoph[i] = w2h[i * 64];
        } else {
#pragma unroll
          for (int i = 0; i < 4 * NL_P; ++i) aop[i] = w2[i * 64];
        }
        bt_check<NL_P, 0, NL_P>(rs, par + w_lane_h1, tag, v, wctx, 0x300u);
        nap_h1 = bt_nap_adjust(nap_h1, wctx.spun);
        BT_STAMP(23);
        BT_SB();
        f32x4 acc2 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < NL_P; ++i) {
          if constexpr (BF) { acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, aoph[i]), bt_pack(v[i]), acc, 0, 0, 0); }
          else { BT_MFMA4F(aop, i, v[i], acc, acc2) }
        }
        acc += acc2;
      }
      acc = bt_reduce_tiles(lds, LP_RED, acc, wave, lane_f);
      if (wave == 0) {
        float pv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = j * RP + 4 * kq + r;
          float v = acc[r] > 0.0f ? acc[r] : 0.0f;
          v = ((keep_l2 >> (8 * r)) & 0xffu) ? v * 2.0f : 0.0f;
          pv[r] = row < P ? v : 0.0f;
        }
        bt_store4<REP>(rs, par + (unsigned)a.lay.p + bt_word(j * RP + 4 * kq, utt), a.lay.p_cap, tag, pv[0], pv[1], pv[2], pv[3]);
        keep_l1 = kn1;
        keep_next = kn2;
      }
      taco_lds_barrier();      // LP_RED is reused by the layer-1 rows below
    }

    BT_STAMP(0);
    // ============ frame boundary: the previous frame's stop tokens ============
    unsigned fresh = 0;                         // slots whose utterance starts with this frame
    if (t > 0) {
      const taco_granule* g_stop = reinterpret_cast<const taco_granule*>(a.words + (((t - 1) & 1) ? a.lay.per_parity : 0) + a.lay.stop + rep * a.lay.stop_cap);
      if (wave == 0 && lane < NB) lds[L_STOP + lane] = taco_wait_one(g_stop, lane, (unsigned)t, wctx, 0x100u);
      if (wctx.failed) lds[L_STOP + 16] = 1.0f;
      taco_lds_barrier();
      if (lds[L_STOP + 16] != 0.0f) break;
      const bool fired = !forced && taco_sigmoid(lds[L_STOP + utt]) > a.thr;
      const unsigned fmask = (unsigned)__ballot(fired) & 0xffffu;
      const unsigned lmask = (unsigned)__ballot(t - t0_u >= a.max_steps) & 0xffffu;
      unsigned ended = alive & (fmask | lmask);
      alive &= ~ended;
      while (ended) {
        // the slot's utterance is complete (the stopping frame is kept); the next one that fits is booked for the slot
        // and starts at the next frame that is a multiple of four (see the head of the file)
        const int sl = __builtin_ctz(ended);
        ended &= ended - 1u;
        const int done = __builtin_amdgcn_readlane(utt_u, sl), done_t0 = __builtin_amdgcn_readlane(t0_u, sl);
        if (j == 0 && tid == 0) {
          a.n_frames[done] = t - done_t0;
          a.reached[done] = (uint8_t)((((fmask >> sl) & 1u) == 0u && !forced) ? 1u : 0u);
        }
        const int nu = next_utterance();
        if (nu >= 0) {
          waiting |= 1u << sl;
          if (utt == sl) { utt_u = nu; t0_u = (t + 3) & ~3; }
        }
      }
      if ((t & 3) == 0 && waiting) {
        fresh = waiting;
        alive |= waiting;
        waiting = 0u;
      }
      if (fresh) {
        // a slot's new utterance starts from the initial state: nothing carried over from the frame before in its
        // column (partial sums over the previous hidden states and context, cell states; the prenet output of its
        // first frame is zeroed where it is consumed)
        if ((fresh >> utt) & 1u) {
          acc_a = {0.0f, 0.0f, 0.0f, 0.0f};
          acc_dn = {0.0f, 0.0f, 0.0f, 0.0f};
          c_a = 0.0f;
          c_d = 0.0f;
          if (is_p && wave == 0) {
            const int r0 = min(j * RP + 4 * kq, P - 4);
            const uint8_t* src = a.masks + (size_t)min(1, a.max_steps - 1) * 2 * P + r0;
            keep_l1 = *reinterpret_cast<const unsigned*>(src);
            keep_next = *reinterpret_cast<const unsigned*>(src + P);
          }
        }
        if (is_q && ((fresh >> q_slot) & 1u)) {
          q_b = __builtin_amdgcn_readlane(utt_u, q_slot);
          q_t0 = t;
          fill_slot();
          taco_lds_barrier();
        }
      }
    }
    if ((alive | waiting) == 0u || t >= a.t_limit) break;
    if (a.leave && t > 0 && (t & 3) == 0 && cursor >= a.B && ((alive | waiting) & 0xffffu) != 0xffffu) { left = 1; break; }
    BT_STAMP(1);

    // ============ attention LSTM ============
    {
      u32x4 v[NL_P];
      float4 wp4[BF ? 1 : NL_P];
      if constexpr (!BF) {
#pragma unroll
        for (int i = 0; i < NL_P; ++i) wp4[i] = ws_lane[i * 64];       // my rows' prenet columns (L2), beside the wait below
      }
      bt_nap_begin(nap_p, wctx);
      bt_issue<NL_P>(rs, par + w_lane_p, v);
      bt_check<NL_P, 0, NL_P>(rs, par + w_lane_p, tag, v, wctx, 0x400u);
      nap_p = bt_nap_adjust(nap_p, wctx.spun);
      if (fresh || t == 0) { // the first frame of an utterance is fed the all-zero frame: its prenet output is zero (exact
                             // zeros also for the utterances that start with the launch, whose published words are
                             // zeros with tag bits: both kinds of start then take the same arithmetic)
        if (t == 0 || ((fresh >> utt) & 1u)) {
#pragma unroll
          for (int i = 0; i < NL_P; ++i) v[i] = u32x4{0u, 0u, 0u, 0u};
        }
        if (t == 0 && col_carried && t0_u != 0) {      // a resumed utterance: the prenet output it was about to consume when it
                                                       // was left (one that was left before its first frame starts from zeros)
          const u32x4* src = reinterpret_cast<const u32x4*>(a.carry_p + bt_word(16 * (wave * NL_P) + 4 * kq, utt));
#pragma unroll
          for (int i = 0; i < NL_P; ++i) v[i] = src[64 * i];
        }
      }
      BT_STAMP(2);
      BT_PROBE(1)
      f32x4 acc2 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int i = 0; i < NL_P; ++i) {
        if constexpr (BF) { acc_a = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wa_p[i], bt_pack(v[i]), acc_a, 0, 0, 0); }
        else { BT_MFMA4S(wp4[i], v[i], acc_a, acc2) }
      }
      acc_a += acc2;
      const f32x4 pre = bt_reduce_tiles(lds, L_RED, acc_a, wave, lane_f);
      if (wave == 0) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(lds + L_BIAS + 4 * kq_f);
        const float hv = bt_lstm_cell(pre, bias, c_a);
        const int hu = j * 4 + kq;
        bt_publish_units<REP>(rs, par + (unsigned)a.lay.ah + bt_word(j * 4, utt), a.lay.ah_cap, tag, hu < Ra ? hv : 0.0f, lane);
      }
    }
    BT_STAMP(3);

    u32x4 vah[NL_H];        // the attention hidden state of all utterances: my wave's share of its columns
    // ============ Q: query values and partial energies of my 16 dimensions, every position of my utterance ============
    if (is_q) {
      {
        // the hidden state of my utterance: 1024 words, 16 bytes per thread
        u32x4 v[1];
        const unsigned src = par + (unsigned)(a.lay.ah + rep * a.lay.ah_cap) + bt_word(4 * tid_f, q_slot);
        bt_nap_begin(nap_ahq, wctx);
        v[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, src * 4u, 0, 16);
        // (the loop of bt_check fetches load i from widx + 256 i: with one load that is the same address)
        bt_check<1, 0, 1>(rs, src, tag, v, wctx, 0x500u);
        nap_ahq = bt_nap_adjust(nap_ahq, wctx.spun);
        *reinterpret_cast<float4*>(lds + BT_OPQ(LQ_XPE + 4 * tid_f)) = make_float4(BT_F(v[0].x), BT_F(v[0].y), BT_F(v[0].z), BT_F(v[0].w));
#ifndef BT_NO_AH_AHEAD
        // the first half of everybody's hidden state (its products run while the partial energies travel) is asked for
        // now, a query and an energy stage ahead: my utterance's values have come from all the workgroups, so the
        // other utterances' are there too as a rule, and their round trip is over when the products begin
        BT_SB();
#pragma unroll
        for (int i = 0; i < NL_H / 2; ++i) vah[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (par + w_lane_ah + 256u * i) * 4u, 0, 16);
#endif
      }
      taco_lds_barrier();
      BT_STAMP(4);
      BT_PROBE(2)
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
      BT_STAMP(16);
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
            // (the four quarters of the wave hold four dimensions each of the same 16 positions: v_permlane16_swap /
            // v_permlane32_swap put a lane's partner quarter beside it in one vector instruction each, where a
            // __shfl_xor is a round trip through the LDS crossbar; same additions, same bits)
            {
              const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(e), __float_as_uint(e), false, false);
              e = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
              const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(e), __float_as_uint(e), false, false);
              e = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
            }
            // four neighbouring positions meet in the first lane of their quad: one 16-byte store of tagged words
            e = s < len ? e : 0.0f;
            const float e1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(e), 0x55, 0xf, 0xf, true));
            const float e2 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(e), 0xaa, 0xf, 0xf, true));
            const float e3 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(e), 0xff, 0xf, 0xf, true));
            if (kq_f == 0 && (utt_f & 3) == 0)
              bt_store4<1>(rs, par + (unsigned)(a.lay.pe + (q_slot * NQ + dg) * TIN + s), 0, tag, e, e1, e2, e3);
          }
        }
      }
    }
    BT_STAMP(5);

    // ============ all: the attention hidden state: its decoder-LSTM columns (this frame) and its attention-LSTM columns
    //              (next frame) on the same operands ============
    f32x4 acc_d = acc_dn;
    f32x4 acc_an = {0.0f, 0.0f, 0.0f, 0.0f};
    u32x4 v_pe[2];          // Q: my share of the slot's partial energies
    {
      f32x4 acc_d2 = {0.0f, 0.0f, 0.0f, 0.0f}, acc_an2 = {0.0f, 0.0f, 0.0f, 0.0f};
      if (is_q) {
        // a Q workgroup has the softmax ahead of it and everybody waiting for its context: it takes half of the vector
        // here - about what the partial energies take to travel - and the other half while its context travels
        wctx.spun = 0;
#ifdef BT_NO_AH_AHEAD
#pragma unroll
        for (int i = 0; i < NL_H / 2; ++i) vah[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (par + w_lane_ah + 256u * i) * 4u, 0, 16);
#endif
      } else {
        // nobody but the Q workgroups is in a hurry for this vector (its products are needed when the context arrives):
        // the others stay away from memory a little longer, so that the Q workgroups' fetches are served first
        for (int i = 0; i < BT_DELAY_AH; ++i) __builtin_amdgcn_s_sleep(1);
        bt_nap_begin(nap_ah, wctx);
        bt_issue<NL_H>(rs, par + w_lane_ah, vah);
      }
      BT_SB();
#define BT_AH_BURST(c)                                                                             \
  bt_check<NL_H, BURST * (c), BURST>(rs, par + w_lane_ah, tag, vah, wctx, 0x600u);                  \
  if ((c) == (BT_NAPALL_AH ? (is_q ? 1 : 3) : 0)) nap_ah = bt_nap_adjust(nap_ah, wctx.spun);       \
  BT_SB();                                                                                         \
  _Pragma("unroll") for (int i = BURST * (c); i < BURST * (c) + BURST; ++i) {                      \
    BT_MFMA4(wd_a, i, vah[i], acc_d, acc_d2)                                                       \
    BT_MFMA4(wa_h, i, vah[i], acc_an, acc_an2)                                                     \
  }                                                                                                \
  BT_SB();
      BT_AH_BURST(0)
      if constexpr (is_q) {
        // the partial energies of my utterance are asked for here, a burst of products ahead of their use: their round
        // trip runs beside the second burst instead of after it
#pragma unroll
        for (int i = 0; i < 2; ++i) v_pe[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (par + pe_src0 + 128u * i) * 4u, 0, 16);
        BT_SB();
      }
      BT_AH_BURST(1)
      if (!is_q) { BT_AH_BURST(2) BT_AH_BURST(3) }
#undef BT_AH_BURST
      acc_d += acc_d2;
      acc_an += acc_an2;
    }
    BT_STAMP(6);

    // ============ Q: energies of my utterance, softmax, my columns of the context ============
    if (is_q) {
      {
        // partial energies of my utterance's 8 workgroups: thread (group tid >> 5, l = tid & 31) takes positions
        // 4 l .. 4 l + 3 and 4 (l + 32) .. of that group
        u32x4 (&v)[2] = v_pe;
        const unsigned src = par + pe_src0;
        const int pd = BT_OPQ(LQ_XPE + (tid_f >> 5) * TIN + 4 * (tid_f & 31));
        unsigned spins = 0;
        bool again = false;
        wctx.spun = 0;
        for (;;) {
          if (again) {
#pragma unroll
            for (int i = 0; i < 2; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (src + 128u * i) * 4u, 0, 16);
          }
          again = true;
          unsigned bad = 0;
#pragma unroll
          for (int i = 0; i < 2; ++i)
            if (i < pe_n) bad |= (v[i].x ^ tag) | (v[i].y ^ tag) | (v[i].z ^ tag) | (v[i].w ^ tag);
          if (__all((bad & 3u) == 0u) || wctx.failed) break;
          wctx.spun = 1;
          __builtin_amdgcn_s_sleep(TACO_POLL_SLEEP);
          if ((++spins & 255u) == 0u) {
            if (__hip_atomic_load(wctx.abort_word, TACO_RLX_AGENT) != 0u) { wctx.failed = 1; break; }
            if (spins >= TACO_SPIN_LIMIT) {
              __hip_atomic_store(wctx.abort_word, TACO_ABORT_TIMEOUT, TACO_RLX_AGENT);
              __hip_atomic_store(wctx.abort_word + 1, 0x700u, TACO_RLX_AGENT);
              wctx.failed = 1;
              break;
            }
          }
          asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
          if (i < pe_n) *reinterpret_cast<float4*>(lds + pd + 128 * i) = make_float4(BT_F(v[i].x), BT_F(v[i].y), BT_F(v[i].z), BT_F(v[i].w));
      }
      taco_lds_barrier();
      BT_STAMP(7);
      BT_PROBE(3)
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
        BT_STAMP(17);
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
        if (dg == 0 && a.align && ((alive >> q_slot) & 1u)) a.align[((size_t)q_b * a.T_cap + (t - q_t0)) * a.T_in + tid] = w_own;
      }
      taco_lds_barrier();
      BT_STAMP(18);
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
      BT_STAMP(19);
      if (tid < 4 * CQ) {
        // four threads per column quad: each adds up to four of the 15 groups' partial sums, the four meet inside their
        // quad of lanes, the first stores 16 bytes (columns at or beyond M are zeros)
        const int cgq = tid_f >> 2, part = tid_f & 3;
        float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const float* cp = lds + BT_OPQ(LQ_XPE + 4 * cgq + part * MEMW);
#pragma unroll
        for (int i = 0; i < (CPG + 3) / 4; ++i) {
          if (part + 4 * i < CPG) {
            const float4 v = *reinterpret_cast<const float4*>(cp + 4 * i * MEMW);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
          }
        }
        auto quad_sum = [](float v) {
          v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xb1, 0xf, 0xf, true));      // lanes 1 0 3 2
          v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4e, 0xf, 0xf, true));      // lanes 2 3 0 1
          return v;
        };
        s.x = quad_sum(s.x); s.y = quad_sum(s.y); s.z = quad_sum(s.z); s.w = quad_sum(s.w);
        if (part == 0) bt_store4<REP>(rs, par + (unsigned)a.lay.ctx + bt_word(c0 + 4 * cgq, q_slot), a.lay.ctx_cap, tag, s.x, s.y, s.z, s.w);
      } else if (dg == 0 && tid < 4 * CQ + (KC - NQ * MEMW) / 4) {
        // the padding columns of the exchanged vector
        bt_store4<REP>(rs, par + (unsigned)a.lay.ctx + bt_word(NQ * MEMW + 4 * (tid - 4 * CQ), q_slot), a.lay.ctx_cap, tag, 0.0f, 0.0f, 0.0f, 0.0f);
      }
      BT_STAMP(8);
    }
    if (a.frame_clock && j == a.n_wg - 1 && tid == 0 && t < a.frame_clock_cap) a.frame_clock[t] = wall_clock64();

    // ============ all: the context finishes the decoder LSTM ============
    f32x4 acc_r = {0.0f, 0.0f, 0.0f, 0.0f};     // P: folded layer-1 rows; M: projection rows
    f32x4 acc_r2 = {0.0f, 0.0f, 0.0f, 0.0f};
    {
      u32x4 vc[NL_C];
      f32x4 acc_d2 = {0.0f, 0.0f, 0.0f, 0.0f};
      if (is_q) {
        // Q: the other half of the attention hidden state (see above); the context loads go out before its last products
        u32x4 vah[NL_H / 2];
#pragma unroll
        for (int i = 0; i < NL_H / 2; ++i) vah[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (par + w_lane_ah + 256u * (NL_H / 2 + i)) * 4u, 0, 16);
        BT_SB();
        bt_check<NL_H / 2, 0, BURST>(rs, par + w_lane_ah + 256u * (NL_H / 2), tag, vah, wctx, 0x680u);
        BT_SB();
#pragma unroll
        for (int i = 0; i < BURST; ++i) {
          BT_MFMA4(wd_a, NL_H / 2 + i, vah[i], acc_d, acc_d2)
          BT_MFMA4(wa_h, NL_H / 2 + i, vah[i], acc_an, acc_an)
        }
        BT_SB();
        bt_issue<NL_C>(rs, par + w_lane_c, vc);
        BT_SB();
        bt_check<NL_H / 2, BURST, BURST>(rs, par + w_lane_ah + 256u * (NL_H / 2), tag, vah, wctx, 0x680u);
        BT_SB();
#pragma unroll
        for (int i = BURST; i < 2 * BURST; ++i) {
          BT_MFMA4(wd_a, NL_H / 2 + i, vah[i], acc_d, acc_d2)
          BT_MFMA4(wa_h, NL_H / 2 + i, vah[i], acc_an, acc_an)
        }
        BT_SB();
        wctx.spun = 0;
      } else {
        bt_nap_begin(nap_ctx, wctx);
        bt_issue<NL_C>(rs, par + w_lane_c, vc);
      }
      BT_SB();
      bt_check<NL_C, 0, 3>(rs, par + w_lane_c, tag, vc, wctx, 0x800u);
      if (!BT_NAPALL_CTX) nap_ctx = bt_nap_adjust(nap_ctx, wctx.spun);
      BT_STAMP(10);
      BT_SB();
#pragma unroll
      for (int i = 0; i < 3; ++i) { BT_MFMA4(wd_c, i, vc[i], acc_d, acc_d2) }
      BT_SB();
      bt_check<NL_C, 3, 3>(rs, par + w_lane_c, tag, vc, wctx, 0x800u);
      BT_SB();
#pragma unroll
      for (int i = 3; i < 6; ++i) { BT_MFMA4(wd_c, i, vc[i], acc_d, acc_d2) }
      BT_SB();
      bt_check<NL_C, 6, 3>(rs, par + w_lane_c, tag, vc, wctx, 0x800u);
      if (BT_NAPALL_CTX) nap_ctx = bt_nap_adjust(nap_ctx, wctx.spun);
      BT_SB();
#pragma unroll
      for (int i = 6; i < 9; ++i) { BT_MFMA4(wd_c, i, vc[i], acc_d, acc_d2) }
      acc_d += acc_d2;
      BT_PROBE(4)
      BT_STAMP(20);
      BT_STAMP_W(24, 1); BT_STAMP_W(25, 2); BT_STAMP_W(26, 3);
      const f32x4 pre = bt_reduce_tiles(lds, L_RED, acc_d, wave, lane_f);
      BT_STAMP(21);
      if (wave == 0) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(lds + L_BIAS + 16 + 4 * kq_f);
        const float hv = bt_lstm_cell(pre, bias, c_d);
        const int hu = j * 4 + kq;
        bt_publish_units<REP>(rs, par + (unsigned)a.lay.dh + bt_word(j * 4, utt), a.lay.dh_cap, tag, hu < Rd ? hv : 0.0f, lane);
      }
      BT_STAMP(11);
      BT_SB();
      float4 wc4[BF ? 1 : NL_C];
      if constexpr (!BF) {
#pragma unroll
        for (int i = 0; i < NL_C; ++i) wc4[i] = ws_lane[(NL_P + i) * 64];   // my attention-LSTM rows' context columns (L2)
      }
      // while the decoder hidden state travels: the context columns of the P / M rows ...
      if (BF && is_p) {
        if (!forced) {
          const u32x2* crow = reinterpret_cast<const u32x2*>(lds + LP_WF) + BT_OPQ((KH / 16 + wave * NL_C) * 64 + lane_f);
#pragma unroll
          for (int i = 0; i < NL_C; ++i) {
            const s16x4 aoph = __builtin_bit_cast(s16x4, crow[i * 64]);
            if (i & 1) acc_r2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(aoph, bt_pack(vc[i]), acc_r2, 0, 0, 0);
            else acc_r = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(aoph, bt_pack(vc[i]), acc_r, 0, 0, 0);
          }
        }
      } else if ((is_p && !forced) || is_m) {
        const float* crow = lds + BT_OPQ((is_p ? LP_WF : LM_W) + (KH / 4 + wave * 4 * NL_C) * 64 + lane_f);
#pragma unroll
        for (int i = 0; i < NL_C; ++i) {
          float aop[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) aop[e] = crow[(4 * i + e) * 64];
          BT_MFMA4F(aop, 0, vc[i], acc_r, acc_r2)
        }
      }
      // ... and the context columns of next frame's attention LSTM
      acc_a = acc_an;
      f32x4 acc_a2 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int i = 0; i < NL_C; ++i) {
        if constexpr (BF) {
          if (i & 1) acc_a2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wa_c[i], bt_pack(vc[i]), acc_a2, 0, 0, 0);
          else acc_a = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wa_c[i], bt_pack(vc[i]), acc_a, 0, 0, 0);
        } else { BT_MFMA4S(wc4[i], vc[i], acc_a, acc_a2) }
      }
      acc_a += acc_a2;
    }

    // ============ the decoder hidden state; P: next frame's layer-1 rows; M: the frame's output ============
    {
      u32x4 vdh[NL_H];
      // (the P workgroups head the next frame with this vector and the M workgroups make the frame of it; the others need
      // it for products that have until the next context to finish, and let those two groups fetch first)
      if (!(is_p || is_m)) { for (int i = 0; i < BT_DELAY_DH; ++i) __builtin_amdgcn_s_sleep(1); }
      bt_nap_begin(nap_dh, wctx);
      BT_PROBE(5)
      bt_issue<NL_H>(rs, par + w_lane_dh, vdh);
      BT_SB();
      if (is_p || is_m) {
        // these rows are what other workgroups wait for: they come first, on operands that stay in registers for the
        // recurrent products afterwards
        const float* rows = lds + BT_OPQ((is_p ? LP_WF : LM_W) + (wave * 4 * NL_H) * 64 + lane_f);
        const u32x2* rows_h = reinterpret_cast<const u32x2*>(lds + LP_WF) + BT_OPQ(wave * NL_H * 64 + lane_f);   // BF, P only
        const bool use_rows = is_m || !forced;
        const bool rows_bf = BF && is_p;
#define BT_ROWS_BURST(c)                                                                           \
  {                                                                                                \
    float aop[4 * BURST];                                                                          \
    u32x2 aoph[BURST];                                                                             \
    if (use_rows) {                                                                                \
      if (rows_bf) {                                                                               \
        _Pragma("unroll") for (int i = 0; i < BURST; ++i) aoph[i] = rows_h[(BURST * (c) + i) * 64]; \
      } else {                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 4 * BURST; ++i) aop[i] = rows[(4 * BURST * (c) + i) * 64]; \
      }                                                                                            \
    }                                                                                              \
    bt_check<NL_H, BURST * (c), BURST>(rs, par + w_lane_dh, tag, vdh, wctx, 0x900u);                \
    if ((c) == 0) { BT_STAMP(12); }                                                              \
    if ((c) == (BT_NAPALL_DH ? 3 : 0)) nap_dh = bt_nap_adjust(nap_dh, wctx.spun);                  \
    BT_SB();                                                                                       \
    if (use_rows) {                                                                                \
      if (rows_bf) {                                                                               \
        _Pragma("unroll") for (int i = 0; i < BURST; ++i) {                                        \
          if (i & 1) acc_r2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, aoph[i]), bt_pack(vdh[BURST * (c) + i]), acc_r2, 0, 0, 0); \
          else acc_r = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, aoph[i]), bt_pack(vdh[BURST * (c) + i]), acc_r, 0, 0, 0); \
        }                                                                                          \
      } else {                                                                                     \
        _Pragma("unroll") for (int i = 0; i < BURST; ++i) { BT_MFMA4F(aop, i, vdh[BURST * (c) + i], acc_r, acc_r2) } \
      }                                                                                            \
    }                                                                                              \
    BT_SB();                                                                                       \
  }
        BT_ROWS_BURST(0) BT_ROWS_BURST(1) BT_ROWS_BURST(2) BT_ROWS_BURST(3)
#undef BT_ROWS_BURST
        acc_r += acc_r2;
        BT_STAMP(9);
        BT_STAMP_W(27, 1); BT_STAMP_W(28, 3);
        if (is_p && forced) {
          // teacher forcing: the plain layer-1 rows on the given frame (wave w takes k-steps 5w .. 5w + 4)
          constexpr int W1_PER_WAVE = (W1_KSTEPS + NW - 1) / NW;
          const float* w1row = lds + BT_OPQ(LP_W1 + W1_PER_WAVE * wave * 64 + lane_f);
#pragma unroll
          for (int i = 0; i < W1_PER_WAVE; ++i) {
            const int ks = W1_PER_WAVE * wave + i;
            const int k = 16 * (ks >> 2) + 4 * kq_f + (ks & 3);
            float xv = 0.0f;
            if (ks < W1_KSTEPS && k < NMEL) xv = a.forced_mel[((size_t)utt_u * NMEL + k) * a.T_cap + max(0, min(t - t0_u, a.T_cap - 1))];
            if (ks < W1_KSTEPS) acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(w1row[i * 64], xv, acc_r, 0, 0, 0);
          }
        }
        const f32x4 sum = bt_reduce_tiles(lds, is_p ? LP_RED : LM_RED, acc_r, wave, lane_f);
        BT_STAMP(15);
        if (wave == 0) {
          if (is_p) {
            {
              const unsigned par_n = ((t + 1) & 1) ? (unsigned)a.lay.per_parity : 0u;
              const unsigned tag_n = ((((unsigned)t + 1u) >> 1) & 1u) + 1u;
              float hv1[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int row = j * RP + 4 * kq + r;
                float v = sum[r] + (forced ? 0.0f : lds[LP_BIAS + 4 * kq + r]);
                v = v > 0.0f ? v : 0.0f;
                v = ((keep_l1 >> (8 * r)) & 0xffu) ? v * 2.0f : 0.0f;
                hv1[r] = row < P ? v : 0.0f;
              }
              bt_store4<REP>(rs, par_n + (unsigned)a.lay.h1 + bt_word(j * RP + 4 * kq, utt), a.lay.h1_cap, tag_n, hv1[0], hv1[1], hv1[2], hv1[3]);
              keep_l2 = keep_next;
            }
          } else {
            const int b_out = utt_u, tb = t - t0_u;      // my column's utterance and its own frame index
            const bool wr = (alive >> utt) & 1u;
            taco_granule* g_stop = reinterpret_cast<taco_granule*>(gw + a.lay.stop);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = (j - n_p_wg) * RP + 4 * kq + r;
              const float v = sum[r] + lds[LM_BIAS + 4 * kq + r];
              if (row < NMEL) {
                if (wr) a.mel[((size_t)b_out * NMEL + row) * a.T_cap + tb] = v;
              } else if (row == NMEL) {
#pragma unroll
                for (int rr = 0; rr < REP; ++rr) taco_publish(g_stop + (size_t)rr * (a.lay.stop_cap / 2), utt, ep, v);
                if (wr) a.gate[(size_t)b_out * a.T_cap + tb] = v;
              }
            }
          }
        }
        BT_STAMP(13);
        acc_dn = {0.0f, 0.0f, 0.0f, 0.0f};
        f32x4 acc_dn2 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < NL_H; ++i) { BT_MFMA4(wd_h, i, vdh[i], acc_dn, acc_dn2) }
        acc_dn += acc_dn2;
      } else {
        acc_dn = {0.0f, 0.0f, 0.0f, 0.0f};
        f32x4 acc_dn2 = {0.0f, 0.0f, 0.0f, 0.0f};
#define BT_DH_BURST(c)                                                                             \
  bt_check<NL_H, BURST * (c), BURST>(rs, par + w_lane_dh, tag, vdh, wctx, 0x980u);                  \
  if ((c) == 0) { BT_STAMP(12); }                                                                \
  if ((c) == (BT_NAPALL_DH ? 3 : 0)) nap_dh = bt_nap_adjust(nap_dh, wctx.spun);                    \
  BT_SB();                                                                                         \
  _Pragma("unroll") for (int i = BURST * (c); i < BURST * (c) + BURST; ++i) { BT_MFMA4(wd_h, i, vdh[i], acc_dn, acc_dn2) } \
  BT_SB();
        BT_DH_BURST(0) BT_DH_BURST(1) BT_DH_BURST(2) BT_DH_BURST(3)
#undef BT_DH_BURST
        acc_dn += acc_dn2;
      }
    }
    // Q: next frame's location term, while the P group works on the next frame's prenet output
    BT_PROBE(6)
    BT_STAMP(22);
    if (is_q) update_location_term(lane_f);
    BT_STAMP(14);
  }
  };
  if (is_p) frame_loop(std::integral_constant<int, 1>{});
  else if (is_m) frame_loop(std::integral_constant<int, 2>{});
  else if (is_q) frame_loop(std::integral_constant<int, 3>{});
  else frame_loop(std::integral_constant<int, 0>{});

  if (left) {
    // what the running utterances carry into their next frame (every column's: the columns of idle slots are not used)
    carry_mine[0] = acc_a[0]; carry_mine[1] = acc_a[1]; carry_mine[2] = acc_a[2]; carry_mine[3] = acc_a[3];
    carry_mine[4] = acc_dn[0]; carry_mine[5] = acc_dn[1]; carry_mine[6] = acc_dn[2]; carry_mine[7] = acc_dn[3];
    carry_mine[8] = c_a;
    carry_mine[9] = c_d;
    if (is_q) {
      float* dst = a.carry_q + (size_t)q_id * BT_CARRY_Q;
      for (int i = tid; i < BT_CARRY_Q; i += NT) dst[i] = lds[LQ_AW + i];
    }
    if (j == 0 && wave == 0 && lane < NB && ((alive >> lane) & 1u)) {
      a.carry_ctl[lane] = t - t0_u;
      a.n_frames[utt_u] = t - t0_u;        // frames so far
      a.reached[utt_u] = (uint8_t)2u;      // 2 = left unfinished
      a.slot_out[utt_u] = lane;
    }
  }
  if (j == 0 && tid == 0) {
    a.ctl->n_frames = (uint32_t)t;       // lockstep frames of the launch
    a.ctl->reached_max = left ? 2u : 0u;
    a.ctl->left_slots = left ? (alive & 0xffffu) : 0u;      // the slots whose utterances were left (their state is in carry_*)
  }
}

// One-time re-packing of the attention LSTM's input columns (prenet output | context) into the order the frame loop
// streams them in: out[((wg * 4 + wave) * 13 + load) * 64 + lane] = the four A operands (one float4) of lane `lane` for
// the four k-steps of `load` (loads 0..3: prenet columns, 4..12: context columns); same row / column map as the
// register-resident rows.
__global__ void taco_bt_pack_kernel(const float* att_w_ih, int P, int M, int Ra, float4* out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int NLD = NL_P + NL_C;
  if (idx >= TACO_BT_WGS * NW * NLD * 64) return;
  const int lane = idx & 63, li = (idx >> 6) % NLD, wave = ((idx >> 6) / NLD) % NW, wg = (idx >> 6) / (NLD * NW);
  const int row16 = lane & 15, kq = lane >> 4;
  const int hid = wg * 4 + (row16 >> 2), gate = row16 & 3;
  float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (hid < Ra) {
    const float* row = att_w_ih + ((size_t)gate * Ra + hid) * (P + M);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (li < NL_P) {
        const int k = 16 * (wave * NL_P + li) + 4 * kq + e;
        if (k < P) v[e] = row[k];
      } else {
        const int k = 16 * (wave * NL_C + (li - NL_P)) + 4 * kq + e;
        if (k < M) v[e] = row[P + k];
      }
    }
  }
  out[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

}  // namespace

size_t taco_decoder_batch_wstream_floats() { return (size_t)TACO_BT_WGS * NW * (NL_P + NL_C) * 64 * 4; }

int taco_decoder_batch_pack(taco_handle h) {
  const int n = TACO_BT_WGS * NW * (NL_P + NL_C) * 64;
  hipLaunchKernelGGL(taco_bt_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, h->w.att_w_ih, h->d.P, h->d.M, h->d.R_a,
                     reinterpret_cast<float4*>(h->bat_wstream));
  TACO_HIP(hipGetLastError());
  return TACO_OK;
}

size_t taco_decoder_batch_lds_bytes() { return (size_t)L_TOTAL * sizeof(float); }

// After a launch that left utterances unfinished at lockstep frame `t_left`: the prenet output they were about to consume
// (published at that frame's top, replica 0 of that frame's parity block) goes from the exchange buffer to the carry area.
int taco_decoder_batch_keep_prenet_output(taco_handle h, int t_left, hipStream_t s) {
  const size_t words_r = (size_t)TACO_BT_WGS * NW * 64 * BT_CARRY_R + (size_t)NB * NQ * BT_CARRY_Q;
  const uint32_t* src = h->bat_gran + (size_t)(t_left & 1) * h->bat_lay.per_parity + h->bat_lay.p;
  TACO_HIP(hipMemcpyAsync(h->bat_carry + words_r, src, sizeof(uint32_t) * (size_t)NB * KP, hipMemcpyDeviceToDevice, s));
  return TACO_OK;
}

void taco_decoder_batch_fill_layout(taco_bat_layout* lay) {
  int off = 0;
  auto take = [&](int n, int reps, int32_t* cap) {
    const int at = off;
    const int c = (n + 31) / 32 * 32;      // 128-byte aligned replicas
    if (cap) *cap = c;
    off += c * reps;
    return at;
  };
  lay->h1 = take(NB * KP, TACO_BT_REP, &lay->h1_cap);
  lay->p = take(NB * KP, TACO_BT_REP, &lay->p_cap);
  lay->ah = take(NB * KH, TACO_BT_REP, &lay->ah_cap);
  lay->ctx = take(NB * KC, TACO_BT_REP, &lay->ctx_cap);
  lay->dh = take(NB * KH, TACO_BT_REP, &lay->dh_cap);
  lay->pe = take(NB * NQ * TIN, 1, nullptr);
  lay->stop = take(2 * NB, TACO_BT_REP, &lay->stop_cap);
  lay->per_parity = off;
}

size_t taco_decoder_batch_handoff_bytes() {
  taco_bat_layout lay;
  taco_decoder_batch_fill_layout(&lay);
  return (size_t)lay.per_parity * 2 * sizeof(uint32_t);
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
  for (const void* k : {reinterpret_cast<const void*>(taco_decoder_batch_kernel<false>), reinterpret_cast<const void*>(taco_decoder_batch_kernel<true>)}) {
    TACO_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)taco_decoder_batch_lds_bytes()));
    int per_cu = 0;
    TACO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, NT, taco_decoder_batch_lds_bytes()));
    if (per_cu < 1) TACO_FAIL(TACO_EINVAL, "lockstep decoder kernel does not fit on a CU (occupancy query returned %d)", per_cu);
  }
  return TACO_OK;
}

#ifdef TACO_BT_STAMPS
#include <stdio.h>
static unsigned long long* g_bt_stamps = nullptr;
// Diagnostic build: when each stamp point of the last launch is reached, on the common clock, in microseconds from the
// first P workgroup's frame top of the same frame (mean over the frames): a timeline of four single workgroups, then
// per stamp point the earliest / mean / latest workgroup of each role.
void taco_decoder_batch_dump_stamps(int n_frames) {
  if (!g_bt_stamps) return;
  constexpr int NWG = TACO_BT_WGS;
  const size_t n = (size_t)NWG * BT_STAMP_FRAMES * BT_STAMP_POINTS;
  std::vector<unsigned long long> st(n);
  if (hipMemcpy(st.data(), g_bt_stamps, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
  const int nfr = n_frames < BT_STAMP_FRAMES ? n_frames : BT_STAMP_FRAMES;
  static const char* names[BT_STAMP_POINTS] = {"frame top: P layer 2 done, p published", "stop test done", "p loaded", "att MFMA+reduce+cell, ah published",
      "Q: ah of my utterance gathered", "Q: query+energies, pe published", "ah loads + products done", "Q: pe gathered", "Q: softmax+ctx, ctx published",
      "P/M: all row products issued", "ctx first burst loaded", "dec MFMA+reduce+cell, dh published", "slack products done, dh first burst loaded",
      "P/M: rows reduced, epilogue, h1 / frame published", "recurrent MFMAs (+ Q: location term) done", "P/M: tiles reduced",
      "Q: query values in LDS", "Q: energies summed, exp done", "Q: weights normalised, in LDS", "Q: context partial sums in LDS",
      "ctx: all bursts loaded, products issued", "dec tiles reduced", "recurrent products done (before Q's location term)", "P: h1 loaded (frame top)",
      "ctx products issued, wave 1", "ctx products issued, wave 2", "ctx products issued, wave 3", "P/M: row products issued, wave 1", "P/M: row products issued, wave 3",
      "", "", ""};
  const int n_p = KP / RP, n_m = (TACO_MAX_MEL + 1 + RP - 1) / RP, q_off = n_p + n_m;
  std::vector<double> at((size_t)NWG * BT_STAMP_POINTS, 0.0);
  std::vector<int> cnt((size_t)NWG * BT_STAMP_POINTS, 0);
  for (int wg = 0; wg < NWG; ++wg)
    for (int t = 40; t + 1 < nfr; ++t) {
      const unsigned long long* r = &st[((size_t)wg * BT_STAMP_FRAMES + t) * BT_STAMP_POINTS];
      const unsigned long long base = st[((size_t)0 * BT_STAMP_FRAMES + t) * BT_STAMP_POINTS];
      for (int k = 0; k < BT_STAMP_POINTS; ++k)
        if (r[k] != 0) { at[(size_t)wg * BT_STAMP_POINTS + k] += ((double)r[k] - (double)base) * 0.01; ++cnt[(size_t)wg * BT_STAMP_POINTS + k]; }
    }
  for (size_t i = 0; i < at.size(); ++i) if (cnt[i] > 0) at[i] /= cnt[i];
  const int shown[4] = {0, n_p, q_off, NWG - 1};
  static const char* wgn[4] = {"P (0)", "M (first)", "Q (first)", "plain (last)"};
  for (int w = 0; w < 4; ++w) {
    const int wg = shown[w];
    int order[BT_STAMP_POINTS], m = 0;
    for (int k = 0; k < BT_STAMP_POINTS; ++k) if (cnt[(size_t)wg * BT_STAMP_POINTS + k] > 0) order[m++] = k;
    if (m == 0) continue;
    const double* a0 = &at[(size_t)wg * BT_STAMP_POINTS];
    for (int x = 1; x < m; ++x)
      for (int y = x; y > 0 && a0[order[y]] < a0[order[y - 1]]; --y) { const int z = order[y]; order[y] = order[y - 1]; order[y - 1] = z; }
    fprintf(stderr, "[bt stamps] workgroup %s (us from the P group's frame top; step since the point before):\n", wgn[w]);
    for (int x = 0; x < m; ++x)
      fprintf(stderr, "  %7.2f  %+6.2f  %s\n", a0[order[x]], x ? a0[order[x]] - a0[order[x - 1]] : 0.0, names[order[x]]);
  }
  static const char* rolen[4] = {"P", "M", "Q", "plain"};
  fprintf(stderr, "[bt stamps] per stamp point and role: earliest / mean / latest workgroup (its number)\n");
  for (int k = 0; k < BT_STAMP_POINTS; ++k) {
    for (int role = 0; role < 4; ++role) {
      double lo = 1e30, hi = -1e30, sum = 0.0; int c = 0, wlo = -1, whi = -1;
      for (int wg = 0; wg < NWG; ++wg) {
        const int r = wg < n_p ? 0 : wg < q_off ? 1 : wg < q_off + NB * NQ ? 2 : 3;
        if (r != role || cnt[(size_t)wg * BT_STAMP_POINTS + k] == 0) continue;
        const double v = at[(size_t)wg * BT_STAMP_POINTS + k];
        if (v < lo) { lo = v; wlo = wg; }
        if (v > hi) { hi = v; whi = wg; }
        sum += v; ++c;
      }
      if (c > 0) fprintf(stderr, "  [%2d] %-5s %7.2f (%3d) / %7.2f / %7.2f (%3d)  %s\n", k, rolen[role], lo, wlo, sum / c, hi, whi, names[k]);
    }
  }
}
#endif

// One lockstep launch for all utterances of a call that fit the kernel (own length <= TACO_BT_TIN; `n_fit` of the B,
// counted by the caller): sixteen at a time, a slot taking the next one as soon as its utterance has stopped.
size_t taco_decoder_batch_carry_floats() {
  // per column partial sums and cell states | per Q workgroup weights and cumulative weights | the prenet output vector (words)
  return (size_t)TACO_BT_WGS * NW * 64 * BT_CARRY_R + (size_t)NB * NQ * BT_CARRY_Q + (size_t)NB * KP + 64;
}

int taco_launch_decoder_batch(taco_handle h, const float* memory, const float* pm, int T_in, const int32_t* lens,
                              int B, int n_fit, int group, const uint8_t* masks, int T_cap, int max_steps, float thr,
                              const float* forced_mel, float* mel, float* gate, float* align, int32_t* n_frames,
                              uint8_t* reached, const int32_t* resume, int leave, int32_t* slot_out, hipStream_t s) {
  const taco_dims& d = h->d;
  if (n_fit < 1 || n_fit > B) TACO_FAIL(TACO_EINVAL, "lockstep launch for %d of %d utterances", n_fit, B);
  if (!lens && T_in > TIN) TACO_FAIL(TACO_EINVAL, "lockstep launch: T_in = %d without lengths (capacity %d)", T_in, TIN);
  if (T_in < 1 || T_in > TACO_MAX_TIN) TACO_FAIL(TACO_EINVAL, "T_in = %d", T_in);
  if (max_steps < 1 || max_steps > T_cap) TACO_FAIL(TACO_EINVAL, "max_steps %d outside 1..T_cap=%d", max_steps, T_cap);
  BatArgs a;
  a.P = d.P; a.M = d.M; a.Ra = d.R_a; a.Rd = d.R_d; a.A = d.A; a.K = d.K; a.n_mel = d.n_mel; a.T_in = T_in; a.n_wg = TACO_BT_WGS;
  a.B = B;
  const long long most = (long long)n_fit * (max_steps + 4) + 4;     // every lockstep frame is a frame of at least one utterance, or one of up to 3 before a start
  a.t_limit = most > 0x7fffffffLL ? 0x7fffffff : (int)most;
  a.lens = lens;
  const taco_weights& w = h->w;
  a.prenet_w1 = w.prenet_w1; a.prenet_w2 = w.prenet_w2; a.wf_h = h->wf_h; a.wf_c = h->wf_c; a.wf_b = h->wf_b;
  a.att_w_ih = w.att_w_ih; a.att_w_hh = w.att_w_hh; a.att_b_ih = w.att_b_ih; a.att_b_hh = w.att_b_hh;
  a.query_w = w.query_w; a.v_w = w.v_w; a.loc_fold = h->loc_fold;
  a.dec_w_ih = w.dec_w_ih; a.dec_w_hh = w.dec_w_hh; a.dec_b_ih = w.dec_b_ih; a.dec_b_hh = w.dec_b_hh;
  a.proj_w = w.proj_w; a.proj_b = w.proj_b; a.gate_w = w.gate_w; a.gate_b = w.gate_b;
  a.wstream = reinterpret_cast<const float4*>(h->bat_wstream);
  a.memory = memory; a.pm = pm; a.masks = masks; a.T_cap = T_cap; a.max_steps = max_steps; a.thr = thr;
  a.shift_energies = h->energy_bound > 64.0f ? 1 : 0;
  a.forced_mel = forced_mel; a.mel = mel; a.gate = gate; a.align = align; a.n_frames = n_frames; a.reached = reached;
  a.words = h->bat_gran; a.lay = h->bat_lay;
  a.words_bytes = (unsigned)taco_decoder_batch_handoff_bytes();
  a.ctl = h->dec_ctl + group;
  a.resume = resume;
  a.leave = leave;
  a.carry_r = h->bat_carry;
  a.carry_q = a.carry_r + (size_t)TACO_BT_WGS * NW * 64 * BT_CARRY_R;
  a.carry_p = reinterpret_cast<const uint32_t*>(a.carry_q + (size_t)NB * NQ * BT_CARRY_Q);
  a.carry_ctl = reinterpret_cast<int32_t*>(const_cast<uint32_t*>(a.carry_p) + (size_t)NB * KP);
  a.slot_out = slot_out;
  if ((resume || leave) && !slot_out) TACO_FAIL(TACO_EINVAL, "lockstep launch that leaves or resumes utterances needs slot_out");
  a.frame_clock = group == 0 ? h->frame_clock : nullptr;
  a.frame_clock_cap = h->frame_clock_cap;
  TACO_HIP(hipMemsetAsync(h->bat_gran, 0, a.words_bytes, s));
#ifdef TACO_BT_STAMPS
  {
    const size_t n = (size_t)TACO_BT_WGS * BT_STAMP_FRAMES * BT_STAMP_POINTS;
    if (!g_bt_stamps) TACO_HIP(hipMalloc((void**)&g_bt_stamps, n * 8));
    TACO_HIP(hipMemsetAsync(g_bt_stamps, 0, n * 8, s));
    a.stamps = g_bt_stamps;
  }
#endif
  if (group == 0) TACO_HIP(hipEventRecord(h->ev_start, s));
  if (h->bat_bf16) hipLaunchKernelGGL(taco_decoder_batch_kernel<true>, dim3(TACO_BT_WGS), dim3(NT), taco_decoder_batch_lds_bytes(), s, a);
  else hipLaunchKernelGGL(taco_decoder_batch_kernel<false>, dim3(TACO_BT_WGS), dim3(NT), taco_decoder_batch_lds_bytes(), s, a);
  TACO_HIP(hipGetLastError());
  return TACO_OK;
}
