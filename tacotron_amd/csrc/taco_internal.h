// Internal declarations shared by the translation units of libtaco_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/taco.h"

// ---- error plumbing -------------------------------------------------------------------------
void taco_set_error(const char* fmt, ...);
#define TACO_FAIL(code, ...)   \
  do {                         \
    taco_set_error(__VA_ARGS__); \
    return (code);             \
  } while (0)
#define TACO_HIP(call)                                                                      \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) TACO_FAIL(TACO_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

// ---- compile-time capacities of the persistent decoder ----------------------------------------
// LSTM rows live in registers in chunks of 256 columns (one ds_read_b128 per lane per chunk).
constexpr int TACO_CHUNK = 256;
constexpr int TACO_NP = 1;    // prenet-width chunks            -> P   <= 256
constexpr int TACO_NM = 3;    // memory-width chunks            -> M   <= 768
constexpr int TACO_NA = 4;    // attention-LSTM hidden chunks   -> R_a <= 1024
constexpr int TACO_ND = 4;    // decoder-LSTM hidden chunks     -> R_d <= 1024
constexpr int TACO_MAX_A = 128;
constexpr int TACO_MAX_F = 32;
constexpr int TACO_MAX_K = 31;
constexpr int TACO_MAX_MEL = 80;            // prenet layer 1 is replicated in every workgroup's LDS
constexpr int TACO_MAX_TIN = 1024;          // symbols per utterance (the batch-1 decoder's capacity)
constexpr int TACO_FOLD_TIN = 512;          // positions whose folded context weights a workgroup keeps on chip (registers / LDS);
                                            // longer utterances read the rest from global memory every frame
constexpr int TACO_DEC_THREADS = 512;       // 8 waves: two LSTM gate rows per wave
constexpr int TACO_UNITS_PER_WG = 4;        // 4 hidden units x 4 gates = 16 rows per LSTM per workgroup
// Producer groups of the decoder's short vectors (see taco_decoder.hip)
constexpr int TACO_P_ROWS_WG = 8;           // prenet layer-2 rows per P workgroup (one per wave)
constexpr int TACO_NQ_WG = 8;               // Q workgroups: attention dimensions are split 8 ways
constexpr int TACO_MEL_ROWS_WG = 3;         // projection / stop-token rows per M workgroup

// Tagged 8-byte hand-off word: high half = epoch (frame + 1), low half = fp32 payload.
typedef unsigned long long taco_granule;

// Granule vectors of the decoder, in units of granules from the base; two parities each.  A vector
// is stored in several identical replicas (producers write all, consumer workgroup j polls replica
// j % R): 256 pollers on the few cache lines of a short vector is what limits a hand-off's latency.
#ifndef TACO_RS
#define TACO_RS 4
#define TACO_RP 2
#define TACO_RB 2
#endif
constexpr int TACO_REP_SMALL = TACO_RS;   // prenet layer outputs, stop token (<= 256 granules)
constexpr int TACO_REP_PE = TACO_RP;      // energy partial sums (8 groups x positions)
constexpr int TACO_REP_BIG = TACO_RB;     // LSTM hidden states (1024 granules)
struct taco_dec_layout {
  int32_t h1, p, ah, pe, dh, mel;                // offsets of replica 0
  int32_t h1_cap, p_cap, ah_cap, pe_cap, dh_cap, mel_cap;   // replica strides
  int32_t per_parity;                            // granules per parity block
};

// ---- the lockstep (batched) decoder: capacities and hand-off layout ----
constexpr int TACO_BT_THREADS = 256;    // 4 waves: the LSTM rows fill most of the CU's register file
constexpr int TACO_BT_WGS = 256;        // workgroups of a launch: 4 hidden units of each cell per workgroup
constexpr int TACO_BATCH_NB = 16;       // utterances per lockstep launch = columns of one matrix-core tile
constexpr int TACO_BT_KP = 256;         // k capacity of the prenet vectors
constexpr int TACO_BT_KH = 1024;        // ... of the LSTM hidden states
constexpr int TACO_BT_KC = 576;         // ... of the attention context (M <= 8 * TACO_BT_MEMW = 544)
constexpr int TACO_BT_TIN = 224;        // symbols per utterance (what the Q workgroups' LDS has room for)
constexpr int TACO_BT_MEMW = 68;        // memory columns per Q workgroup
constexpr int TACO_BT_ROWS = 16;        // rows per P / M workgroup
#ifndef TACO_BT_REP
#define TACO_BT_REP 2                   // replicas of every all-to-all vector
#endif
struct taco_bat_layout {                // offsets in 4-byte words inside one parity block
  // all-to-all vectors, [utterance][k] tagged fp32 words, TACO_BT_REP replicas each
  int32_t h1, p, ah, ctx, dh;
  int32_t h1_cap, p_cap, ah_cap, ctx_cap, dh_cap;             // replica strides (words)
  // 8-byte {epoch, fp32} granules (even word offsets): partial energies [utterance][8 groups][positions], stop tokens [16]
  int32_t pe, stop;
  int32_t stop_cap;                                           // replica stride of the stop tokens (words)
  int32_t per_parity;                                         // words per parity block
};

struct taco_dec_ctl {      // device-visible control block (zeroed before every launch)
  uint32_t abort_code;     // first failure code raised by any workgroup (0 = none)
  uint32_t abort_info;     // vector id / frame of the failed wait
  uint32_t n_frames;       // frames produced
  uint32_t reached_max;
  uint32_t left_slots;     // lockstep launch that left utterances unfinished: bit s = slot s holds one
  uint32_t pad_[11];
};

struct taco_handle_s {
  int device;
  int n_cu;
  taco_dims d;
  taco_weights w;            // device pointers into `pool` (original checkpoint layout)
  float* pool;               // owned copy of every weight tensor
  size_t pool_floats;
  // folded BatchNorm: y = conv * alpha + beta
  float* enc_alpha[TACO_MAX_CONVS];
  float* enc_beta[TACO_MAX_CONVS];
  float* post_alpha[TACO_MAX_CONVS];
  float* post_beta[TACO_MAX_CONVS];
  float* lstm_bias_sum;      // [2][4*H_e]  b_ih + b_hh of the encoder LSTM
  float* lstm_w_ih_cat;      // [2*4*H_e][E] both directions stacked
  // Context columns of every matrix the attention context feeds, stacked row-wise:
  // attention_rnn.weight_ih[:, P:], decoder_rnn.weight_ih[:, R_a:], linear_projection[:, R_d:],
  // gate_layer[:, R_d:]  ->  [4*R_a + 4*R_d + n_mel + 1][M]
  float* ctx_w_cat;
  int ctx_rows;
  // Location layers folded into one filter per attention dimension:
  // loc_fold[a][c][k] = sum_f location_dense[a][f] * location_conv[f][c][k], rows a >= A zero, [TACO_MAX_A + 4][2][K]
  float* loc_fold;
  // Prenet layer 1 folded into the mel projection: wf_h [P][R_d] = W1 * linear_projection[:, :R_d], wf_b [P] = W1 * bias
  float* wf_h;
  float* wf_b;
  float* wf_c;               // [P][M] = W1 * linear_projection[:, R_d:] (the lockstep decoder's explicit context columns)
  float* ws_G;               // [ctx_rows][T_in] folded context weights G = ctx_w_cat * memory^T
  // workspaces (grow-only)
  float* ws_act[2];          // activation ping-pong [B][C_max][T]
  size_t ws_act_floats;
  unsigned bat_left_slots;   // slots whose utterances the handle's last lockstep launch left unfinished (what a resuming call may name)
  float* conv_split;         // partial tiles and tile counters of the convolutions whose tiles are shared by several workgroups
  float* ws_gx;              // [B][2][4*H_e][T] LSTM input projections
  size_t ws_gx_floats;
  taco_granule* enc_gran;    // encoder LSTM hand-off words
  size_t enc_gran_count;
  int enc_group;             // utterances per launch of the encoder recurrence (0 = not worked out yet)
  taco_granule* dec_gran;    // decoder hand-off words, 2 parities
  taco_dec_layout dec_lay;
  uint32_t* bat_gran;        // lockstep decoder hand-off words, 2 parities (NULL when the model exceeds its capacities)
  taco_bat_layout bat_lay;
  float* bat_wstream;        // attention-LSTM input columns in the lockstep kernel's streaming order
  float* bat_carry;          // state of the utterances a lockstep launch left unfinished (taco_decode_resumable)
  int bat_ok;                // the lockstep kernel can run this model
  int bat_bf16;              // its LSTM gate rows are held and multiplied as bf16 (taco_set_precision)
  taco_dec_ctl* dec_ctl;     // device, one block per launch of a taco_decode call (grow-only)
  taco_dec_ctl* dec_ctl_host;  // pinned host mirror
  int dec_ctl_cap;
  uint32_t* enc_ctl;         // device: [0] abort code, [1] abort info of the encoder LSTM kernel, [2] 1 + channel of an
                             // out-of-range input id (speakers: 1 + TACO_MAX_CHANNELS)
  uint32_t* enc_ctl_host;    // pinned host mirror [16]
  hipEvent_t ev_start, ev_stop;
  int last_frames;
  int timing_valid;
  float energy_bound;        // sum |v_a|: no attention energy can exceed it in magnitude
  unsigned long long* frame_clock;   // caller's device buffer for per-frame clock readings, or NULL
  int frame_clock_cap;
};

// ---- launchers implemented in the kernel translation units ----------------------------------
int taco_launch_embed(taco_handle h, const int64_t* const* ids_dev, const int32_t* lens, int B, int T,
                      float* x, hipStream_t s);
// Conv1d(k, pad (k-1)/2) + per-channel affine (folded BatchNorm) + activation (+ residual).
// in [B][c_in][T], out [B][c_out][T]; layout bit 0: in is [B][T][c_in], bit 1: out is [B][T][c_out];
// columns >= lens[b] are written as zero.
int taco_launch_conv(const float* in, float* out, const float* w, const float* bias, const float* alpha,
                     const float* beta, const float* residual, const int32_t* lens, int B, int c_in,
                     int c_out, int k, int T, int act, int layout, float* split_ws, int split_cap, int n_cu, hipStream_t s);
// A tile's sum over the input channels is made of UNITS of TACO_CONV_UNIT_SLABS slabs: the products of a unit form one
// accumulation chain, the units' sums are added in order - whether one workgroup runs all units of a tile or, in launches
// with few tiles, every unit has a workgroup of its own.  So a result does not depend on how large its launch was
// (an utterance's encoder and postnet outputs are bit-identical whatever batch it is in).
// split_ws: workspace of taco_conv_split_floats(split_cap) floats (split_cap workgroups' partial tiles, then one counter
// per tile; the counters must be zero when the first launch uses them and every launch leaves them zero), or NULL =
// every tile by one workgroup.  Launches that share a workspace must not overlap.
constexpr int TACO_CONV_UNIT_SLABS = 4;
constexpr int TACO_CONV_PARTIAL = 4 * 16 * 64;               // floats a workgroup leaves: four waves' 32 x 32 accumulators
constexpr int TACO_CONV_SPLIT_CAP = 512;                     // workgroup partials in a handle's workspace (8 MiB)
constexpr int TACO_CONV_SPLIT_CAP_BUILTIN = 256;             // ... in the built-in scratch of taco_postnet_noalloc (4 MiB)
constexpr size_t taco_conv_split_floats(int cap) { return (size_t)cap * TACO_CONV_PARTIAL + (size_t)cap; }
int taco_conv_prepare();
int taco_launch_enc_lstm(taco_handle h, const float* gx, const int32_t* lens, int B, int T, float* memory,
                         hipStream_t s);
size_t taco_enc_lstm_granules(int n_cu);
// Reads back the encoder's flags (the caller has synchronised the stream after copying them to
// enc_ctl_host); clears them on the device when one is set.
int taco_check_encoder_flags(taco_handle h, hipStream_t s);
int taco_launch_memory_tail(taco_handle h, const int64_t* speakers, const int32_t* lens, int B, int T,
                            float* memory, float* pm, hipStream_t s);
int taco_launch_loc_fold(const float* dense_w, const float* conv_w, int A, int F, int K, int rows, float* out, hipStream_t s);
int taco_launch_slice_cols(const float* src, int ld, int col0, int rows, int cols, float* dst, hipStream_t s);
int taco_launch_decoder(taco_handle h, const float* memory, const float* pm, int T_in, const int32_t* lens,
                        int b, int launch_idx, int own_len, const uint8_t* masks, int T_cap, int max_steps, float thr, const float* forced_mel,
                        float* mel, float* gate, float* align, int32_t* n_frames, uint8_t* reached,
                        hipStream_t s);
int taco_decoder_collect(taco_handle h, int n_launches, hipStream_t s);
// lockstep decoder (taco_decoder_batch.hip)
int taco_launch_decoder_batch(taco_handle h, const float* memory, const float* pm, int T_in, const int32_t* lens,
                              int B, int n_fit, int launch_idx, const uint8_t* masks, int T_cap, int max_steps, float thr,
                              const float* forced_mel, float* mel, float* gate, float* align, int32_t* n_frames,
                              uint8_t* reached, const int32_t* resume, int leave, int32_t* slot_out, hipStream_t s);
size_t taco_decoder_batch_carry_floats();
int taco_decoder_batch_keep_prenet_output(taco_handle h, int t_left, hipStream_t s);
size_t taco_decoder_batch_lds_bytes();
size_t taco_decoder_batch_handoff_bytes();
void taco_decoder_batch_fill_layout(taco_bat_layout* lay);
int taco_decoder_batch_supported(const taco_dims& d, int n_wg);
int taco_decoder_batch_prepare(taco_handle h);
size_t taco_decoder_batch_wstream_floats();
int taco_decoder_batch_pack(taco_handle h);   // fills bat_wstream (after the weight pool is in place)
size_t taco_decoder_lds_bytes();
size_t taco_decoder_granules(int n_wg);
void taco_decoder_fill_layout(taco_dec_layout* lay, int n_wg);
int taco_decoder_check_dims(const taco_dims& d, int n_wg);
int taco_decoder_prepare(taco_handle h);
