/*
 * taco.h - C ABI of the MI355X Tacotron 2 mel-inference library (libtaco_hip.so).
 *
 * The library replaces the arithmetic of one reference call,
 *     Tacotron2.inference(symbols, stresses, tones, durations, speakers, max_decoder_steps)
 *     (reference: src/tacotron/model.py:751-791, called from src/tacotron/synthesizer.py:205-213),
 * with hand-written gfx950 kernels.  Entry points take plain pointers and sizes; no torch types
 * cross this boundary.  Every pointer in a data argument is a DEVICE pointer owned by the caller
 * and valid on the handle's device, except where a parameter is documented as a host pointer.
 * All floating-point tensors are fp32, ids are int64, as in the reference.
 *
 * Conventions: every function returns 0 on success or a negative TACO_E* code and stores a
 * message retrievable with taco_last_error() (per thread).  No exception crosses the boundary.
 * A handle is bound to one device, owns its packed weight copy and its workspaces, and is not
 * thread-safe: the reference keeps per-call decoder state on the module too
 * (src/tacotron/model.py:385-404), so one handle serves one call at a time.
 * Work is enqueued on the hipStream_t the caller passes (as void*); calls return without
 * synchronising unless stated otherwise.
 */
#ifndef TACO_H_
#define TACO_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TACO_MAX_CHANNELS 4   /* symbol, stress, tone, duration - concatenated in this order */
#define TACO_MAX_CONVS 8

#define TACO_OK 0
#define TACO_EINVAL (-22)     /* bad argument or unsupported extent */
#define TACO_ENOMEM (-12)     /* device allocation failed */
#define TACO_EHIP (-5)        /* a HIP runtime call failed */
#define TACO_ERANGE (-34)     /* an input id lies outside its table (reported at the next synchronising call) */
#define TACO_ETIMEDOUT (-110) /* the persistent decoder gave up waiting on a hand-off */

/* Tensor extents of one checkpoint (reference: src/tacotron/model.py:220-260, 293-356). */
typedef struct taco_dims {
  int32_t E;        /* encoder channels = sum of input-channel widths */
  int32_t H_e;      /* BiLSTM hidden size per direction = ceil(E/2) */
  int32_t S;        /* speaker vector width, 0 if the checkpoint models no speakers */
  int32_t M;        /* memory width = 2*H_e + S */
  int32_t P;        /* prenet width */
  int32_t R_a;      /* attention LSTM width */
  int32_t R_d;      /* decoder LSTM width */
  int32_t A;        /* attention space width */
  int32_t F;        /* location filters */
  int32_t K;        /* location kernel size (odd) */
  int32_t n_mel;
  int32_t C_post;   /* postnet hidden channels */
  int32_t k_post;   /* postnet kernel size (odd) */
  int32_t n_post;   /* postnet conv layers (>= 2) */
  int32_t k_enc;    /* encoder kernel size (odd) */
  int32_t n_enc;    /* encoder conv layers */
  int32_t n_channels;                        /* input channels in use, 1..4 */
  int32_t ch_width[TACO_MAX_CHANNELS];       /* columns the channel adds to E */
  int32_t ch_rows[TACO_MAX_CHANNELS];        /* table rows / one-hot classes (ids are < rows) */
  int32_t spk_rows;                          /* speaker table rows / one-hot classes */
} taco_dims;

/* Weights in the checkpoint's own layout (state-dict tensors, row-major, contiguous).
 * The library copies and re-packs what it needs during taco_create; the caller may free
 * these afterwards.  A NULL table pointer means the channel is one-hot encoded
 * (reference: get_vector, src/tacotron/model.py:585-609). */
typedef struct taco_weights {
  const float* ch_table[TACO_MAX_CHANNELS];  /* [ch_rows, ch_width] or NULL = one-hot */
  const float* spk_table;                    /* [spk_rows, S] or NULL = one-hot (or S == 0) */
  /* encoder: n_enc x (Conv1d + BatchNorm1d, eval statistics) then a 1-layer BiLSTM */
  const float* enc_conv_w[TACO_MAX_CONVS];   /* [E, E, k_enc] */
  const float* enc_conv_b[TACO_MAX_CONVS];   /* [E] */
  const float* enc_bn_w[TACO_MAX_CONVS];     /* [E] */
  const float* enc_bn_b[TACO_MAX_CONVS];
  const float* enc_bn_mean[TACO_MAX_CONVS];
  const float* enc_bn_var[TACO_MAX_CONVS];
  const float* enc_lstm_w_ih[2];             /* [4*H_e, E]; index 0 forward, 1 reverse; gate order i,f,g,o */
  const float* enc_lstm_w_hh[2];             /* [4*H_e, H_e] */
  const float* enc_lstm_b_ih[2];             /* [4*H_e] */
  const float* enc_lstm_b_hh[2];
  /* decoder */
  const float* prenet_w1;                    /* [P, n_mel] */
  const float* prenet_w2;                    /* [P, P] */
  const float* att_w_ih;                     /* [4*R_a, P+M]  input = prenet out (+) previous context */
  const float* att_w_hh;                     /* [4*R_a, R_a] */
  const float* att_b_ih;
  const float* att_b_hh;
  const float* query_w;                      /* [A, R_a] */
  const float* memory_w;                     /* [A, M] */
  const float* v_w;                          /* [A] */
  const float* loc_conv_w;                   /* [F, 2, K]  channel 0 = previous weights, 1 = cumulative */
  const float* loc_dense_w;                  /* [A, F] */
  const float* dec_w_ih;                     /* [4*R_d, R_a+M]  input = attention hidden (+) context */
  const float* dec_w_hh;                     /* [4*R_d, R_d] */
  const float* dec_b_ih;
  const float* dec_b_hh;
  const float* proj_w;                       /* [n_mel, R_d+M]  input = decoder hidden (+) context */
  const float* proj_b;                       /* [n_mel] */
  const float* gate_w;                       /* [R_d+M] */
  const float* gate_b;                       /* [1] */
  /* postnet: n_post x (Conv1d + BatchNorm1d); tanh after all but the last */
  const float* post_conv_w[TACO_MAX_CONVS];  /* [c_out, c_in, k_post] */
  const float* post_conv_b[TACO_MAX_CONVS];
  const float* post_bn_w[TACO_MAX_CONVS];
  const float* post_bn_b[TACO_MAX_CONVS];
  const float* post_bn_mean[TACO_MAX_CONVS];
  const float* post_bn_var[TACO_MAX_CONVS];
} taco_weights;

typedef struct taco_handle_s* taco_handle;

/* Message of the calling thread's most recent failure ("" if none). */
const char* taco_last_error(void);

/* Library version string, e.g. "taco-hip 0.1 gfx950". */
const char* taco_version(void);

/* Builds the device-side model: validates extents, allocates and fills the packed weight copy
 * and the workspaces.  Replaces Tacotron2.__init__ + load_state_dict + .to(device)
 * (reference: src/tacotron/training.py:762-768, src/tacotron/synthesizer.py:78-90).
 * `dims` and the taco_weights struct itself are host memory; the pointers inside it are
 * device pointers.  Synchronises the device before returning. */
int taco_create(const taco_dims* dims, const taco_weights* weights, int device, taco_handle* out);

int taco_destroy(taco_handle h);

/* Bytes of weights one decoder frame touches (the roofline denominator of the step). */
int64_t taco_step_weight_bytes(taco_handle h);

/* Input vectors + encoder + speaker concatenation + attention memory projection.
 * Replaces get_vector x(1..4), Encoder.inference, the speaker concat and memory_layer
 * (reference: src/tacotron/model.py:752-782, 280-289, 403).
 *   ids[c]            [B, T_in] int64 per input channel c < n_channels (others ignored)
 *   speakers          [B, T_in] int64, or NULL when S == 0
 *   lens              [B] int32 valid symbols per utterance (1..T_in); rows are right-padded.
 *                     NULL means every utterance has T_in symbols.
 *   memory            out [B, T_in, M]; rows at or past lens[b] are zero
 *   processed_memory  out [B, T_in, A]
 * For a padded utterance the result equals the unpadded B=1 call: activations past lens[b]
 * are zero before every convolution and the reverse LSTM starts at position lens[b]-1. */
int taco_encode(taco_handle h, const int64_t* const* ids, const int64_t* speakers, const int32_t* lens,
                int32_t B, int32_t T_in, float* memory, float* processed_memory, void* stream);

/* Autoregressive decoder: all frames of an utterance inside one persistent launch, no host round trip per frame.
 * Utterances of up to 224 symbols go through the lockstep kernel: ONE launch advances sixteen of them frame by frame
 * (their LSTM gate rows as matrix-core tiles), and when one stops the next utterance of the call takes its slot, so
 * a large B keeps all sixteen slots busy; an utterance's result there is bit-identical whatever else is in the call.
 * Longer utterances (up to 1024 symbols; T_in beyond that returns TACO_EINVAL), and an utterance left alone, each get a
 * launch of the batch-1 kernel (same results to 1e-4 under teacher forcing).  The launches of one call run back to back
 * on `stream`.
 * Environment: TACO_DECODER=single or =lockstep forces one kernel.
 * Replaces Decoder.inference (reference: src/tacotron/model.py:529-568, 454-491, 96-117, 136-141).
 *   memory, processed_memory, lens   as produced by / passed to taco_encode
 *                (the lengths are copied to the host once per call: a length outside 1..T_in is refused
 *                with TACO_EINVAL, and in batches wider than 256 positions each utterance short enough
 *                runs the faster kernel instance)
 *   masks        [T_cap, 2, P] uint8 (0 or 1): keep-masks of the two always-on prenet dropouts
 *                for frame t, layer 0/1, shared by all utterances; kept activations are scaled by 2
 *   max_steps    frames to produce at most, 1..T_cap (the reference's max_decoder_steps)
 *   gate_threshold   stop when sigmoid(gate) > gate_threshold (strict); the stopping frame is kept
 *   forced_mel   NULL, or [B, n_mel, T_cap]: teacher forcing - frame t>0 is fed forced_mel[..., t-1]
 *                instead of its own previous output, and exactly max_steps frames are produced
 *   mel          out [B, n_mel, T_cap]  (columns at or past n_frames[b] are zero)
 *   gate         out [B, T_cap]         pre-sigmoid stop energies
 *   align        out [B, T_cap, T_in] attention weights, or NULL
 *   n_frames     out [B] int32 frames produced
 *   reached_max  out [B] uint8, 1 when max_steps frames were produced without the stop firing */
int taco_decode(taco_handle h, const float* memory, const float* processed_memory, const int32_t* lens,
                int32_t B, int32_t T_in, const uint8_t* masks, int32_t T_cap, int32_t max_steps,
                float gate_threshold, const float* forced_mel, float* mel, float* gate, float* align,
                int32_t* n_frames, uint8_t* reached_max, void* stream);

/* taco_decode that may stop before every utterance has finished, and take the unfinished ones up again in a later call
 * (no counterpart in the reference, whose loop decodes one utterance at a time: src/tacotron/model.py:529-568).  A call
 * of many utterances ends with a tail in which most of the sixteen slots are empty; with `leave_unfinished` the lockstep
 * launch ends as soon as a slot has no successor instead, and the utterances still running keep their state inside the
 * handle.  For each of them n_frames[b] = the frames it has so far, reached_max[b] = 2 and slot_out[b] = its slot; the
 * frames are in mel / gate / align as far as they go.  The caller puts those utterances into its next call - the same
 * memory / processed_memory rows again, at any index - with resume_slot[b] = that slot: they go on from their next
 * frame, writing columns n_frames[b].. of the NEW call's outputs (the caller copies the earlier columns over), bit for
 * bit as if they had never been interrupted.  The handle keeps ONE such state: every left utterance must be resumed by
 * the next lockstep call on the handle (or abandoned by not naming its slot), with the same masks and limits.
 *   resume_slot       device [B] int32 or NULL: slot to continue from, -1 = the utterance starts anew (a slot in which the
 *                     handle's last lockstep launch left no utterance, or one named twice, is refused with TACO_EINVAL)
 *   leave_unfinished  0 = run every utterance to its end (what taco_decode does)
 *   slot_out          device [B] int32 (required unless both of the above are NULL / 0): -1, or the slot b was left in
 * Utterances too long for the lockstep decoder (over 224 symbols) are decoded to their end by the batch-1 kernel as in
 * taco_decode and cannot be resumed. */
int taco_decode_resumable(taco_handle h, const float* memory, const float* processed_memory, const int32_t* lens,
                          int32_t B, int32_t T_in, const uint8_t* masks, int32_t T_cap, int32_t max_steps,
                          float gate_threshold, const float* forced_mel, float* mel, float* gate, float* align,
                          int32_t* n_frames, uint8_t* reached_max, const int32_t* resume_slot,
                          int32_t leave_unfinished, int32_t* slot_out, void* stream);

/* Waits for the work enqueued on `stream` and reports what the encoder stage found: TACO_ERANGE when
 * an id passed to taco_encode was outside its embedding table / one-hot width (the reference's
 * nn.Embedding / F.one_hot raise for those: src/tacotron/model.py:585-609), TACO_ETIMEDOUT when the
 * encoder recurrence gave up on a hand-off, else 0.  The flags are cleared by the call.
 * taco_encode itself never blocks; taco_decode performs the same check at its own synchronisation,
 * so calling this is optional. */
int taco_encode_status(taco_handle h, void* stream);

/* mel + postnet(mel).  Replaces Postnet.forward and the residual add
 * (reference: src/tacotron/model.py:205-211, 788-789).
 *   mel       [B, n_mel, T_cap]
 *   n_frames  [B] int32 valid frames per utterance (convolutions zero-pad at each utterance's
 *             own end), or NULL meaning T_cap frames each
 *   out       [B, n_mel, T_cap]  (columns at or past n_frames[b] are zero) */
int taco_postnet(taco_handle h, const float* mel, const int32_t* n_frames, int32_t B, int32_t T_cap,
                 float* out, void* stream);

/* taco_postnet for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same arguments and result as
 * taco_postnet, but the layer activations and the convolutions' partial tiles live in a scratch area built into the
 * library (12 MiB per device: 8 MiB of activations, 4 MiB of partial tiles; held until the process ends) instead of
 * the handle's workspaces, and the call makes no hipMalloc / hipFree:
 *   limit   B * max(C_post, n_mel) * T_cap <= 512 * TACO_NOALLOC_COLS floats, else TACO_EINVAL;
 *   sharing every call on a device uses that device's one scratch area: calls that overlap on one GPU (two streams,
 *           two handles) race; calls on one stream are ordered like any other work on it. */
#define TACO_NOALLOC_COLS 2048
int taco_postnet_noalloc(taco_handle h, const float* mel, const int32_t* n_frames, int32_t B, int32_t T_cap,
                         float* out, void* stream);

/* Device-side duration of the most recent taco_decode on this handle, measured with HIP events
 * recorded on the caller's stream around the call's persistent launches: writes microseconds in total and
 * the largest frame count among those launches - for a batch-1 launch the frames its utterance produced, for a
 * lockstep launch its number of lockstep frames (every one advances up to sixteen utterances; refills and the up
 * to three frames a slot waits before a start included), NOT the frames of the longest utterance.  Synchronises on
 * the stop event. */
int taco_last_decode_timing(taco_handle h, double* total_us, int32_t* frames);

/* Per-frame step latency (measurement helper, no counterpart in the reference): from now on the decoder
 * kernel stores, for utterance 0 of every taco_decode call on this handle, the reading of the device's
 * 100 MHz constant clock at the moment frame t's attention weights are final into ticks[t], t < capacity
 * (device memory owned by the caller, which must stay valid until it is replaced or switched off with
 * ticks = NULL).  Differences of consecutive entries x 10 ns are the latencies of the individual frames. */
int taco_set_frame_clock(taco_handle h, uint64_t* ticks, int32_t capacity);

/* Arithmetic of the lockstep decoder's LSTM gate products (the decoder of batches; the batch-1 decoder, the encoder and
 * the postnet always compute in fp32).  The reference computes in fp32 only (src/tacotron/model.py:454-491 runs on the
 * checkpoint's fp32 tensors); the bf16 setting has no counterpart there and is never chosen by the library itself.
 *   TACO_PRECISION_F32   exact-f32 matrix-core products, results within the fp32 parity bounds (default);
 *   TACO_PRECISION_BF16  the gate rows of both LSTM cells are kept as bf16 and multiplied with the hidden states,
 *                        prenet outputs and contexts rounded to bf16 (fp32 accumulation, fp32 cell states and
 *                        gates): faster, mel frames within 2e-3 of the reference's under teacher forcing
 *                        (measured 4.4e-4 over 1000 frames), and the stop frame of a free-running utterance
 *                        may differ from the fp32 one.
 * Takes effect from the next taco_decode on the handle; any other value returns TACO_EINVAL. */
#define TACO_PRECISION_F32 0
#define TACO_PRECISION_BF16 1
int taco_set_precision(taco_handle h, int32_t precision);

#ifdef __cplusplus
}
#endif
#endif /* TACO_H_ */
