// C ABI of libtaco_hip.so (declared in include/taco.h): handle life cycle and the three stages
// of Tacotron2.inference - encode, decode, postnet - as enqueue-on-stream calls.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "taco_internal.h"

static thread_local char g_err[512] = "";

void taco_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* taco_last_error(void) { return g_err; }
extern "C" const char* taco_version(void) { return "taco-hip 0.1 gfx950"; }

// ---- scratch of the entry point that allocates nothing (taco_postnet_noalloc) ----
// Two activation buffers of up to TACO_NOALLOC_COLS columns x 512 channels each, inside the code object: every device
// the library runs on holds its own copy (8 MiB, 12 with the workspace that follows them) until the process ends.  The name below is a host-side stand-in, not
// a device address: taco_create asks for the address once per device, with that device current, and every use goes
// through the pointer kept here.
#define TACO_NOALLOC_HALF (512u * (unsigned)TACO_NOALLOC_COLS)
// ... followed by the convolutions' partial-tile workspace (4 MiB + one counter per tile; the counters start as the
// zeros a code object's arrays are loaded with, and every launch leaves them zero)
static __device__ __align__(256) float taco_noalloc_scratch[2u * TACO_NOALLOC_HALF + taco_conv_split_floats(TACO_CONV_SPLIT_CAP_BUILTIN)];
static float* g_noalloc_ptr[64] = {nullptr};

namespace {

__global__ void taco_bn_fold_kernel(const float* w, const float* b, const float* mean, const float* var, float* alpha,
                                    float* beta, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    // eval BatchNorm1d: (x - mean) / sqrt(var + 1e-5) * weight + bias  ==  x * alpha + beta
    const float invstd = 1.0f / sqrtf(var[i] + 1e-5f);
    const float al = invstd * w[i];
    alpha[i] = al;
    beta[i] = b[i] - mean[i] * al;
  }
}

__global__ void taco_add_kernel(const float* a, const float* b, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}

struct PoolPlan {
  std::vector<const float**> slots;   // address of the pointer field inside the handle's weight struct
  std::vector<const float*> src;
  std::vector<size_t> count;
  size_t total = 0;
  void add(const float** slot, const float* src_ptr, size_t n) {
    if (!src_ptr) { *slot = nullptr; return; }
    slots.push_back(slot);
    src.push_back(src_ptr);
    count.push_back(n);
    total += (n + 63) / 64 * 64;      // 256-byte aligned tensors
  }
};

int ensure_floats(float** p, size_t* have, size_t want) {
  if (*have >= want) return TACO_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *have = 0;
  if (hipMalloc((void**)p, want * sizeof(float)) != hipSuccess) TACO_FAIL(TACO_ENOMEM, "hipMalloc of %zu bytes failed", want * sizeof(float));
  *have = want;
  return TACO_OK;
}

// Makes the handle's device current for the duration of an entry point and gives the caller its own back.
struct DeviceScope {
  int prev = -1;
  hipError_t status;
  explicit DeviceScope(int device) {
    (void)hipGetDevice(&prev);
    status = hipSetDevice(device);
  }
  ~DeviceScope() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

}  // namespace

extern "C" int taco_create(const taco_dims* dims, const taco_weights* weights, int device, taco_handle* out) {
  if (!dims || !weights || !out) TACO_FAIL(TACO_EINVAL, "taco_create: NULL argument");
  *out = nullptr;
  const taco_dims& d = *dims;
  if (d.n_channels < 1 || d.n_channels > TACO_MAX_CHANNELS) TACO_FAIL(TACO_EINVAL, "n_channels = %d", d.n_channels);
  if (d.n_enc < 1 || d.n_enc > TACO_MAX_CONVS || d.n_post < 2 || d.n_post > TACO_MAX_CONVS) TACO_FAIL(TACO_EINVAL, "unsupported number of convolution layers (encoder %d, postnet %d)", d.n_enc, d.n_post);
  int esum = 0;
  for (int c = 0; c < d.n_channels; ++c) esum += d.ch_width[c];
  if (esum != d.E) TACO_FAIL(TACO_EINVAL, "channel widths sum to %d but E = %d", esum, d.E);
  if (d.H_e != (d.E + 1) / 2 || d.M != 2 * d.H_e + d.S) TACO_FAIL(TACO_EINVAL, "inconsistent extents: E=%d H_e=%d S=%d M=%d", d.E, d.H_e, d.S, d.M);

  DeviceScope on_device(device);
  TACO_HIP(on_device.status);
  {
    const int prc = taco_conv_prepare();
    if (prc != TACO_OK) return prc;
  }
  hipDeviceProp_t prop;
  TACO_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) TACO_FAIL(TACO_EINVAL, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);

  if (device >= 0 && device < 64 && !g_noalloc_ptr[device]) {
    float* ptr = nullptr;
    TACO_HIP(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(taco_noalloc_scratch)));
    g_noalloc_ptr[device] = ptr;
  }
  taco_handle h = new (std::nothrow) taco_handle_s();
  if (!h) TACO_FAIL(TACO_ENOMEM, "out of host memory");
  memset(h, 0, sizeof(*h));
  h->device = device;
  h->n_cu = prop.multiProcessorCount;
  h->d = d;
  int rc = taco_decoder_check_dims(d, h->n_cu);
  if (rc != TACO_OK) { delete h; return rc; }

  // ---- one owned copy of every tensor, in the checkpoint's layout ----
  const taco_weights& w = *weights;
  taco_weights& hw = h->w;
  PoolPlan plan;
  for (int c = 0; c < d.n_channels; ++c) plan.add(&hw.ch_table[c], w.ch_table[c], (size_t)d.ch_rows[c] * d.ch_width[c]);
  plan.add(&hw.spk_table, d.S > 0 ? w.spk_table : nullptr, (size_t)d.spk_rows * d.S);
  for (int l = 0; l < d.n_enc; ++l) {
    plan.add(&hw.enc_conv_w[l], w.enc_conv_w[l], (size_t)d.E * d.E * d.k_enc);
    plan.add(&hw.enc_conv_b[l], w.enc_conv_b[l], d.E);
    plan.add(&hw.enc_bn_w[l], w.enc_bn_w[l], d.E);
    plan.add(&hw.enc_bn_b[l], w.enc_bn_b[l], d.E);
    plan.add(&hw.enc_bn_mean[l], w.enc_bn_mean[l], d.E);
    plan.add(&hw.enc_bn_var[l], w.enc_bn_var[l], d.E);
  }
  for (int dir = 0; dir < 2; ++dir) {
    plan.add(&hw.enc_lstm_w_ih[dir], w.enc_lstm_w_ih[dir], (size_t)4 * d.H_e * d.E);
    plan.add(&hw.enc_lstm_w_hh[dir], w.enc_lstm_w_hh[dir], (size_t)4 * d.H_e * d.H_e);
    plan.add(&hw.enc_lstm_b_ih[dir], w.enc_lstm_b_ih[dir], (size_t)4 * d.H_e);
    plan.add(&hw.enc_lstm_b_hh[dir], w.enc_lstm_b_hh[dir], (size_t)4 * d.H_e);
  }
  plan.add(&hw.prenet_w1, w.prenet_w1, (size_t)d.P * d.n_mel);
  plan.add(&hw.prenet_w2, w.prenet_w2, (size_t)d.P * d.P);
  plan.add(&hw.att_w_ih, w.att_w_ih, (size_t)4 * d.R_a * (d.P + d.M));
  plan.add(&hw.att_w_hh, w.att_w_hh, (size_t)4 * d.R_a * d.R_a);
  plan.add(&hw.att_b_ih, w.att_b_ih, (size_t)4 * d.R_a);
  plan.add(&hw.att_b_hh, w.att_b_hh, (size_t)4 * d.R_a);
  plan.add(&hw.query_w, w.query_w, (size_t)d.A * d.R_a);
  plan.add(&hw.memory_w, w.memory_w, (size_t)d.A * d.M);
  plan.add(&hw.v_w, w.v_w, d.A);
  plan.add(&hw.loc_conv_w, w.loc_conv_w, (size_t)d.F * 2 * d.K);
  plan.add(&hw.loc_dense_w, w.loc_dense_w, (size_t)d.A * d.F);
  plan.add(&hw.dec_w_ih, w.dec_w_ih, (size_t)4 * d.R_d * (d.R_a + d.M));
  plan.add(&hw.dec_w_hh, w.dec_w_hh, (size_t)4 * d.R_d * d.R_d);
  plan.add(&hw.dec_b_ih, w.dec_b_ih, (size_t)4 * d.R_d);
  plan.add(&hw.dec_b_hh, w.dec_b_hh, (size_t)4 * d.R_d);
  plan.add(&hw.proj_w, w.proj_w, (size_t)d.n_mel * (d.R_d + d.M));
  plan.add(&hw.proj_b, w.proj_b, d.n_mel);
  plan.add(&hw.gate_w, w.gate_w, (size_t)(d.R_d + d.M));
  plan.add(&hw.gate_b, w.gate_b, 1);
  for (int l = 0; l < d.n_post; ++l) {
    const int c_in = l == 0 ? d.n_mel : d.C_post, c_out = l == d.n_post - 1 ? d.n_mel : d.C_post;
    plan.add(&hw.post_conv_w[l], w.post_conv_w[l], (size_t)c_out * c_in * d.k_post);
    plan.add(&hw.post_conv_b[l], w.post_conv_b[l], c_out);
    plan.add(&hw.post_bn_w[l], w.post_bn_w[l], c_out);
    plan.add(&hw.post_bn_b[l], w.post_bn_b[l], c_out);
    plan.add(&hw.post_bn_mean[l], w.post_bn_mean[l], c_out);
    plan.add(&hw.post_bn_var[l], w.post_bn_var[l], c_out);
  }
  // required tensors must be present (tables may be NULL = one-hot)
  const float* const required[] = {w.enc_conv_w[0], w.enc_lstm_w_ih[0], w.enc_lstm_w_hh[1], w.prenet_w1, w.prenet_w2,
                                   w.att_w_ih, w.att_w_hh, w.att_b_ih, w.att_b_hh, w.query_w, w.memory_w, w.v_w,
                                   w.loc_conv_w, w.loc_dense_w, w.dec_w_ih, w.dec_w_hh, w.dec_b_ih, w.dec_b_hh,
                                   w.proj_w, w.proj_b, w.gate_w, w.gate_b, w.post_conv_w[0]};
  for (const float* p : required)
    if (!p) { delete h; TACO_FAIL(TACO_EINVAL, "taco_create: a required weight pointer is NULL"); }
  // ... and so must every per-layer tensor the layer counts call for
  for (int l = 0; l < d.n_enc; ++l)
    for (const float* p : {w.enc_conv_w[l], w.enc_conv_b[l], w.enc_bn_w[l], w.enc_bn_b[l], w.enc_bn_mean[l], w.enc_bn_var[l]})
      if (!p) { delete h; TACO_FAIL(TACO_EINVAL, "taco_create: a tensor of encoder convolution %d is NULL", l); }
  for (int dir = 0; dir < 2; ++dir)
    for (const float* p : {w.enc_lstm_w_ih[dir], w.enc_lstm_w_hh[dir], w.enc_lstm_b_ih[dir], w.enc_lstm_b_hh[dir]})
      if (!p) { delete h; TACO_FAIL(TACO_EINVAL, "taco_create: a tensor of the encoder LSTM (direction %d) is NULL", dir); }
  for (int l = 0; l < d.n_post; ++l)
    for (const float* p : {w.post_conv_w[l], w.post_conv_b[l], w.post_bn_w[l], w.post_bn_b[l], w.post_bn_mean[l], w.post_bn_var[l]})
      if (!p) { delete h; TACO_FAIL(TACO_EINVAL, "taco_create: a tensor of postnet convolution %d is NULL", l); }

  // extra owned tensors: folded BatchNorm, stacked LSTM input weights and bias sums
  const size_t n_bn_enc = (size_t)d.n_enc * 2 * ((d.E + 63) / 64 * 64);
  size_t n_bn_post = 0;
  for (int l = 0; l < d.n_post; ++l) n_bn_post += 2 * (size_t)(((l == d.n_post - 1 ? d.n_mel : d.C_post) + 63) / 64 * 64);
  const size_t n_wcat = (size_t)8 * d.H_e * d.E, n_bsum = (size_t)8 * d.H_e;
  h->ctx_rows = 4 * d.R_a + 4 * d.R_d + d.n_mel + 1;
  const size_t n_ctxw = (size_t)h->ctx_rows * d.M;
  const int fold_rows = TACO_MAX_A + 4;                      // a quad of dimensions may reach past A: zero rows
  const size_t n_fold = (size_t)fold_rows * 2 * d.K;
  h->pool_floats = plan.total + n_bn_enc + n_bn_post + (n_wcat + 63) / 64 * 64 + (n_bsum + 63) / 64 * 64 +
                   (n_ctxw + 63) / 64 * 64 + (n_fold + 63) / 64 * 64 + ((size_t)d.P * d.R_d + 63) / 64 * 64 +
                   ((size_t)d.P + 63) / 64 * 64 + ((size_t)d.n_mel * d.R_d + 63) / 64 * 64 +
                   ((size_t)d.P * d.M + 63) / 64 * 64 + ((size_t)d.n_mel * d.M + 63) / 64 * 64;
  if (hipMalloc((void**)&h->pool, h->pool_floats * sizeof(float)) != hipSuccess) {
    delete h;
    TACO_FAIL(TACO_ENOMEM, "hipMalloc of %zu bytes for the weight pool failed", h->pool_floats * sizeof(float));
  }
  // Tensors are padded to 64 floats and some kernels read a ragged last slab past a row's end, relying on finite
  // values there: the padding must not hold whatever bit patterns the allocation had before.
  if (hipMemset(h->pool, 0, h->pool_floats * sizeof(float)) != hipSuccess) {
    taco_destroy(h);
    TACO_FAIL(TACO_EHIP, "hipMemset of the weight pool failed");
  }
#define TACO_CREATE_HIP(call)                                                                          \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      taco_set_error("%s failed: %s", #call, hipGetErrorString(e_));                                   \
      taco_destroy(h);                                                                                 \
      return TACO_EHIP;                                                                                \
    }                                                                                                  \
  } while (0)
  float* cur = h->pool;
  for (size_t i = 0; i < plan.slots.size(); ++i) {
    TACO_CREATE_HIP(hipMemcpy(cur, plan.src[i], plan.count[i] * sizeof(float), hipMemcpyDeviceToDevice));
    *plan.slots[i] = cur;
    cur += (plan.count[i] + 63) / 64 * 64;
  }
  auto fold = [&](const float* bw, const float* bb, const float* bm, const float* bv, int n, float** alpha, float** beta) {
    *alpha = cur; cur += (n + 63) / 64 * 64;
    *beta = cur; cur += (n + 63) / 64 * 64;
    hipLaunchKernelGGL(taco_bn_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, bw, bb, bm, bv, *alpha, *beta, n);
  };
  for (int l = 0; l < d.n_enc; ++l)
    fold(hw.enc_bn_w[l], hw.enc_bn_b[l], hw.enc_bn_mean[l], hw.enc_bn_var[l], d.E, &h->enc_alpha[l], &h->enc_beta[l]);
  for (int l = 0; l < d.n_post; ++l)
    fold(hw.post_bn_w[l], hw.post_bn_b[l], hw.post_bn_mean[l], hw.post_bn_var[l],
         l == d.n_post - 1 ? d.n_mel : d.C_post, &h->post_alpha[l], &h->post_beta[l]);
  h->lstm_w_ih_cat = cur; cur += (n_wcat + 63) / 64 * 64;
  h->lstm_bias_sum = cur; cur += (n_bsum + 63) / 64 * 64;
  for (int dir = 0; dir < 2; ++dir) {
    TACO_CREATE_HIP(hipMemcpy(h->lstm_w_ih_cat + (size_t)dir * 4 * d.H_e * d.E, hw.enc_lstm_w_ih[dir],
                              (size_t)4 * d.H_e * d.E * sizeof(float), hipMemcpyDeviceToDevice));
    const int n = 4 * d.H_e;
    hipLaunchKernelGGL(taco_add_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, hw.enc_lstm_b_ih[dir],
                       hw.enc_lstm_b_hh[dir], h->lstm_bias_sum + (size_t)dir * n, n);
  }
  h->ctx_w_cat = cur; cur += (n_ctxw + 63) / 64 * 64;
  {
    float* dst = h->ctx_w_cat;
    rc = taco_launch_slice_cols(hw.att_w_ih, d.P + d.M, d.P, 4 * d.R_a, d.M, dst, 0);
    dst += (size_t)4 * d.R_a * d.M;
    if (rc == TACO_OK) rc = taco_launch_slice_cols(hw.dec_w_ih, d.R_a + d.M, d.R_a, 4 * d.R_d, d.M, dst, 0);
    dst += (size_t)4 * d.R_d * d.M;
    if (rc == TACO_OK) rc = taco_launch_slice_cols(hw.proj_w, d.R_d + d.M, d.R_d, d.n_mel, d.M, dst, 0);
    dst += (size_t)d.n_mel * d.M;
    if (rc == TACO_OK) rc = taco_launch_slice_cols(hw.gate_w, d.R_d + d.M, d.R_d, 1, d.M, dst, 0);
    if (rc != TACO_OK) { taco_destroy(h); return rc; }
  }
  h->loc_fold = cur; cur += (n_fold + 63) / 64 * 64;
  rc = taco_launch_loc_fold(hw.loc_dense_w, hw.loc_conv_w, d.A, d.F, d.K, fold_rows, h->loc_fold, 0);
  if (rc != TACO_OK) { taco_destroy(h); return rc; }
  {
    // wf_h = W1 [P][n_mel] * Wp_h [n_mel][R_d] and wf_b = W1 * proj_b, as kernel-size-1 convolutions
    h->wf_h = cur; cur += ((size_t)d.P * d.R_d + 63) / 64 * 64;
    h->wf_b = cur; cur += ((size_t)d.P + 63) / 64 * 64;
    float* wp_h = cur; cur += ((size_t)d.n_mel * d.R_d + 63) / 64 * 64;
    rc = taco_launch_slice_cols(hw.proj_w, d.R_d + d.M, 0, d.n_mel, d.R_d, wp_h, 0);
    if (rc == TACO_OK) rc = taco_launch_conv(wp_h, h->wf_h, hw.prenet_w1, nullptr, nullptr, nullptr, nullptr, nullptr, 1, d.n_mel, d.P, 1, d.R_d, 0, 0, nullptr, 0, h->n_cu, 0);
    if (rc == TACO_OK) rc = taco_launch_conv(hw.proj_b, h->wf_b, hw.prenet_w1, nullptr, nullptr, nullptr, nullptr, nullptr, 1, d.n_mel, d.P, 1, 1, 0, 0, nullptr, 0, h->n_cu, 0);
    // wf_c = W1 [P][n_mel] * Wp_c [n_mel][M]: the same fold for the projection's context columns (lockstep decoder)
    h->wf_c = cur; cur += ((size_t)d.P * d.M + 63) / 64 * 64;
    float* wp_c = cur; cur += ((size_t)d.n_mel * d.M + 63) / 64 * 64;
    if (rc == TACO_OK) rc = taco_launch_slice_cols(hw.proj_w, d.R_d + d.M, d.R_d, d.n_mel, d.M, wp_c, 0);
    if (rc == TACO_OK) rc = taco_launch_conv(wp_c, h->wf_c, hw.prenet_w1, nullptr, nullptr, nullptr, nullptr, nullptr, 1, d.n_mel, d.P, 1, d.M, 0, 0, nullptr, 0, h->n_cu, 0);
    if (rc != TACO_OK) { taco_destroy(h); return rc; }
  }
  TACO_CREATE_HIP(hipGetLastError());
  TACO_CREATE_HIP(hipMalloc((void**)&h->ws_G, sizeof(float) * (size_t)(h->ctx_rows + d.P) * TACO_MAX_TIN));

  // ---- hand-off words, control blocks, events ----
  taco_decoder_fill_layout(&h->dec_lay, h->n_cu);
  TACO_CREATE_HIP(hipMalloc((void**)&h->dec_gran, taco_decoder_granules(h->n_cu) * sizeof(taco_granule)));
  h->bat_ok = taco_decoder_batch_supported(d, h->n_cu);
  if (h->bat_ok) {
    taco_decoder_batch_fill_layout(&h->bat_lay);
    TACO_CREATE_HIP(hipMalloc((void**)&h->bat_gran, taco_decoder_batch_handoff_bytes()));
    TACO_CREATE_HIP(hipMalloc((void**)&h->bat_wstream, taco_decoder_batch_wstream_floats() * sizeof(float)));
    TACO_CREATE_HIP(hipMalloc((void**)&h->bat_carry, taco_decoder_batch_carry_floats() * sizeof(float)));
    TACO_CREATE_HIP(hipMemset(h->bat_carry, 0, taco_decoder_batch_carry_floats() * sizeof(float)));
    rc = taco_decoder_batch_prepare(h);
    if (rc == TACO_OK) rc = taco_decoder_batch_pack(h);
    if (rc != TACO_OK) { taco_destroy(h); return rc; }
  }
  TACO_CREATE_HIP(hipMalloc((void**)&h->conv_split, taco_conv_split_floats(TACO_CONV_SPLIT_CAP) * sizeof(float)));
  TACO_CREATE_HIP(hipMemset(h->conv_split, 0, taco_conv_split_floats(TACO_CONV_SPLIT_CAP) * sizeof(float)));
  h->enc_gran_count = taco_enc_lstm_granules(h->n_cu);
  TACO_CREATE_HIP(hipMalloc((void**)&h->enc_gran, h->enc_gran_count * sizeof(taco_granule)));
  h->dec_ctl_cap = 64;
  TACO_CREATE_HIP(hipMalloc((void**)&h->dec_ctl, sizeof(taco_dec_ctl) * h->dec_ctl_cap));
  TACO_CREATE_HIP(hipMalloc((void**)&h->enc_ctl, 64));
  TACO_CREATE_HIP(hipMemset(h->enc_ctl, 0, 64));
  TACO_CREATE_HIP(hipHostMalloc((void**)&h->enc_ctl_host, 64, hipHostMallocDefault));
  TACO_CREATE_HIP(hipHostMalloc((void**)&h->dec_ctl_host, sizeof(taco_dec_ctl) * h->dec_ctl_cap, hipHostMallocDefault));
  TACO_CREATE_HIP(hipEventCreate(&h->ev_start));
  TACO_CREATE_HIP(hipEventCreate(&h->ev_stop));
  rc = taco_decoder_prepare(h);
  if (rc != TACO_OK) { taco_destroy(h); return rc; }
  TACO_CREATE_HIP(hipDeviceSynchronize());
#undef TACO_CREATE_HIP
  *out = h;
  return TACO_OK;
}

extern "C" int taco_destroy(taco_handle h) {
  if (!h) return TACO_OK;
  DeviceScope on_device(h->device);
  (void)hipDeviceSynchronize();
  if (h->pool) (void)hipFree(h->pool);
  for (int i = 0; i < 2; ++i)
    if (h->ws_act[i]) (void)hipFree(h->ws_act[i]);
  if (h->ws_gx) (void)hipFree(h->ws_gx);
  if (h->ws_G) (void)hipFree(h->ws_G);
  if (h->enc_gran) (void)hipFree(h->enc_gran);
  if (h->conv_split) (void)hipFree(h->conv_split);
  if (h->dec_gran) (void)hipFree(h->dec_gran);
  if (h->bat_gran) (void)hipFree(h->bat_gran);
  if (h->bat_wstream) (void)hipFree(h->bat_wstream);
  if (h->bat_carry) (void)hipFree(h->bat_carry);
  if (h->dec_ctl) (void)hipFree(h->dec_ctl);
  if (h->enc_ctl) (void)hipFree(h->enc_ctl);
  if (h->dec_ctl_host) (void)hipHostFree(h->dec_ctl_host);
  if (h->enc_ctl_host) (void)hipHostFree(h->enc_ctl_host);
  if (h->ev_start) (void)hipEventDestroy(h->ev_start);
  if (h->ev_stop) (void)hipEventDestroy(h->ev_stop);
  delete h;
  return TACO_OK;
}

extern "C" int64_t taco_step_weight_bytes(taco_handle h) {
  if (!h) return 0;
  const taco_dims& d = h->d;
  const int64_t n = (int64_t)d.P * d.n_mel + (int64_t)d.P * d.P + 4ll * d.R_a * (d.P + d.M + d.R_a + 2) +
                    (int64_t)d.A * d.R_a + 2ll * d.F * d.K + (int64_t)d.A * d.F + d.A +
                    4ll * d.R_d * (d.R_a + d.M + d.R_d + 2) + (int64_t)(d.n_mel + 1) * (d.R_d + d.M + 1);
  return n * 4;
}

static int ensure_act(taco_handle h, size_t floats) {
  if (h->ws_act_floats >= floats) return TACO_OK;
  size_t have0 = h->ws_act_floats, have1 = h->ws_act_floats;
  int rc = ensure_floats(&h->ws_act[0], &have0, floats);
  if (rc == TACO_OK) rc = ensure_floats(&h->ws_act[1], &have1, floats);
  h->ws_act_floats = rc == TACO_OK ? floats : 0;
  return rc;
}

extern "C" int taco_encode(taco_handle h, const int64_t* const* ids, const int64_t* speakers, const int32_t* lens,
                           int32_t B, int32_t T_in, float* memory, float* processed_memory, void* stream) {
  if (!h || !ids || !memory || !processed_memory) TACO_FAIL(TACO_EINVAL, "taco_encode: NULL argument");
  if (B < 1 || T_in < 1) TACO_FAIL(TACO_EINVAL, "taco_encode: B = %d, T_in = %d", B, T_in);
  if (T_in > TACO_MAX_TIN) TACO_FAIL(TACO_EINVAL, "taco_encode: T_in = %d symbols exceeds the decoder's capacity %d", T_in, TACO_MAX_TIN);
  hipStream_t s = (hipStream_t)stream;
  const taco_dims& d = h->d;
  DeviceScope on_device(h->device);
  TACO_HIP(on_device.status);
  int rc = ensure_act(h, (size_t)B * d.E * T_in);
  if (rc != TACO_OK) return rc;
  rc = ensure_floats(&h->ws_gx, &h->ws_gx_floats, (size_t)B * 8 * d.H_e * T_in);
  if (rc != TACO_OK) return rc;

  rc = taco_launch_embed(h, ids, lens, B, T_in, h->ws_act[0], s);
  if (rc != TACO_OK) return rc;
  int cur = 0;
  for (int l = 0; l < d.n_enc; ++l) {
    rc = taco_launch_conv(h->ws_act[cur], h->ws_act[cur ^ 1], h->w.enc_conv_w[l], h->w.enc_conv_b[l], h->enc_alpha[l],
                          h->enc_beta[l], nullptr, lens, B, d.E, d.E, d.k_enc, T_in, /*relu*/ 1, 0, h->conv_split, TACO_CONV_SPLIT_CAP, h->n_cu, s);
    if (rc != TACO_OK) return rc;
    cur ^= 1;
  }
  // LSTM input projections of both directions for every position: a kernel-size-1 convolution
  rc = taco_launch_conv(h->ws_act[cur], h->ws_gx, h->lstm_w_ih_cat, h->lstm_bias_sum, nullptr, nullptr, nullptr, lens,
                        B, d.E, 8 * d.H_e, 1, T_in, 0, 0, h->conv_split, TACO_CONV_SPLIT_CAP, h->n_cu, s);
  if (rc != TACO_OK) return rc;
  TACO_HIP(hipMemsetAsync(memory, 0, sizeof(float) * (size_t)B * T_in * d.M, s));
  rc = taco_launch_enc_lstm(h, h->ws_gx, lens, B, T_in, memory, s);
  if (rc != TACO_OK) return rc;
  return taco_launch_memory_tail(h, speakers, lens, B, T_in, memory, processed_memory, s);
}

int taco_check_encoder_flags(taco_handle h, hipStream_t s) {
  const uint32_t abort_code = h->enc_ctl_host[0], info = h->enc_ctl_host[1], bad = h->enc_ctl_host[2];
  if (abort_code == 0 && bad == 0) return TACO_OK;
  TACO_HIP(hipMemsetAsync(h->enc_ctl, 0, 64, s));
  if (bad != 0) {
    if (bad > (uint32_t)TACO_MAX_CHANNELS) TACO_FAIL(TACO_ERANGE, "a speaker id passed to taco_encode is outside the speaker table");
    TACO_FAIL(TACO_ERANGE, "an id of input channel %u passed to taco_encode is outside its table", bad - 1u);
  }
  TACO_FAIL(TACO_ETIMEDOUT, "encoder recurrence hand-off wait gave up (code %u, info 0x%x)", abort_code, info);
}

extern "C" int taco_encode_status(taco_handle h, void* stream) {
  if (!h) TACO_FAIL(TACO_EINVAL, "taco_encode_status: NULL handle");
  hipStream_t s = (hipStream_t)stream;
  DeviceScope on_device(h->device);
  TACO_HIP(on_device.status);
  TACO_HIP(hipMemcpyAsync(h->enc_ctl_host, h->enc_ctl, 64, hipMemcpyDeviceToHost, s));
  TACO_HIP(hipStreamSynchronize(s));
  return taco_check_encoder_flags(h, s);
}

extern "C" int taco_decode(taco_handle h, const float* memory, const float* processed_memory, const int32_t* lens,
                           int32_t B, int32_t T_in, const uint8_t* masks, int32_t T_cap, int32_t max_steps,
                           float gate_threshold, const float* forced_mel, float* mel, float* gate, float* align,
                           int32_t* n_frames, uint8_t* reached_max, void* stream) {
  return taco_decode_resumable(h, memory, processed_memory, lens, B, T_in, masks, T_cap, max_steps, gate_threshold, forced_mel,
                               mel, gate, align, n_frames, reached_max, nullptr, 0, nullptr, stream);
}

extern "C" int taco_decode_resumable(taco_handle h, const float* memory, const float* processed_memory, const int32_t* lens,
                                     int32_t B, int32_t T_in, const uint8_t* masks, int32_t T_cap, int32_t max_steps,
                                     float gate_threshold, const float* forced_mel, float* mel, float* gate, float* align,
                                     int32_t* n_frames, uint8_t* reached_max, const int32_t* resume_slot,
                                     int32_t leave_unfinished, int32_t* slot_out, void* stream) {
  if (!h || !memory || !processed_memory || !masks || !mel || !gate || !n_frames || !reached_max)
    TACO_FAIL(TACO_EINVAL, "taco_decode: NULL argument");
  if ((resume_slot || leave_unfinished) && !slot_out) TACO_FAIL(TACO_EINVAL, "taco_decode_resumable: slot_out is required when utterances are resumed or may be left unfinished");
  if ((resume_slot || leave_unfinished) && !h->bat_ok) TACO_FAIL(TACO_EINVAL, "taco_decode_resumable: this model does not run on the lockstep decoder, which is the one that can leave utterances unfinished");
  if (B < 1) TACO_FAIL(TACO_EINVAL, "taco_decode: B = %d", B);
  // everything that sizes a memset or a launch is checked before anything is enqueued
  if (T_in < 1 || T_in > TACO_MAX_TIN) TACO_FAIL(TACO_EINVAL, "taco_decode: T_in = %d symbols is outside the decoder's capacity 1..%d", T_in, TACO_MAX_TIN);
  if (T_cap < 1) TACO_FAIL(TACO_EINVAL, "taco_decode: T_cap = %d", T_cap);
  if (max_steps < 1 || max_steps > T_cap) TACO_FAIL(TACO_EINVAL, "taco_decode: max_steps %d outside 1..T_cap=%d", max_steps, T_cap);
  hipStream_t s = (hipStream_t)stream;
  const taco_dims& d = h->d;
  DeviceScope on_device(h->device);
  TACO_HIP(on_device.status);
  if (B > h->dec_ctl_cap) {
    taco_dec_ctl *dev = nullptr, *host = nullptr;
    const int cap = B + 64;
    if (hipMalloc((void**)&dev, sizeof(taco_dec_ctl) * cap) != hipSuccess) TACO_FAIL(TACO_ENOMEM, "hipMalloc of %d control blocks failed", cap);
    if (hipHostMalloc((void**)&host, sizeof(taco_dec_ctl) * cap, hipHostMallocDefault) != hipSuccess) {
      (void)hipFree(dev);
      TACO_FAIL(TACO_ENOMEM, "hipHostMalloc of %d control blocks failed", cap);
    }
    TACO_HIP(hipStreamSynchronize(s));
    (void)hipFree(h->dec_ctl);
    (void)hipHostFree(h->dec_ctl_host);
    h->dec_ctl = dev; h->dec_ctl_host = host; h->dec_ctl_cap = cap;
  }
  // outputs start as zeros (columns at or past n_frames stay zero), control blocks cleared: once per call
  TACO_HIP(hipMemsetAsync(h->dec_ctl, 0, sizeof(taco_dec_ctl) * (size_t)B, s));
  TACO_HIP(hipMemsetAsync(mel, 0, sizeof(float) * (size_t)B * d.n_mel * T_cap, s));
  TACO_HIP(hipMemsetAsync(gate, 0, sizeof(float) * (size_t)B * T_cap, s));
  if (align) TACO_HIP(hipMemsetAsync(align, 0, sizeof(float) * (size_t)B * T_cap * T_in, s));
  // Utterances are independent; each one is a persistent launch of its own, enqueued back to back with no
  // host round trip in between.
  // The lengths of a padded batch come to the host (one small copy and wait per call): they are checked, and in a
  // batch wider than one folded-context chunk every utterance short enough runs the faster one-chunk instance.
  std::vector<int32_t> host_resume;
  if (slot_out) TACO_HIP(hipMemsetAsync(slot_out, 0xff, sizeof(int32_t) * (size_t)B, s));      // -1: not left unfinished
  if (resume_slot) {
    host_resume.resize((size_t)B);
    TACO_HIP(hipMemcpyAsync(host_resume.data(), resume_slot, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, s));
  }
  std::vector<int32_t> host_lens;
  if (lens) {
    host_lens.resize((size_t)B);
    TACO_HIP(hipMemcpyAsync(host_lens.data(), lens, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, s));
    TACO_HIP(hipStreamSynchronize(s));
    for (int b = 0; b < B; ++b)
      if (host_lens[b] < 1 || host_lens[b] > T_in) TACO_FAIL(TACO_EINVAL, "taco_decode: lens[%d] = %d is outside 1..T_in=%d", b, host_lens[b], T_in);
  }
  // Which decoder runs what.  Utterances of up to TACO_BT_TIN symbols go through the lockstep kernel - one persistent
  // launch for all of them, sixteen in flight, in the caller's order; a lone utterance and the long ones run the
  // batch-1 kernel, one launch each.  TACO_DECODER=single / lockstep in the environment forces one of them
  // (lockstep still only for utterances within its capacity).
  int min_group = 2;
  if (const char* mode = getenv("TACO_DECODER")) {
    if (mode[0] == 's') min_group = 1 << 30;
    else if (mode[0] == 'l') min_group = 1;
  }
  std::vector<int> fits;
  std::vector<int> rest;
  bool resuming = false;
  if (resume_slot) {
    if (host_lens.empty()) TACO_HIP(hipStreamSynchronize(s));
    unsigned taken = 0;
    for (int b = 0; b < B; ++b) {
      const int sl = host_resume[b];
      if (sl < 0) continue;
      const int own = host_lens.empty() ? T_in : host_lens[b];
      if (sl >= TACO_BATCH_NB || ((taken >> sl) & 1u) || own > TACO_BT_TIN)
        TACO_FAIL(TACO_EINVAL, "taco_decode_resumable: resume_slot[%d] = %d (a slot is 0..%d, taken once, and its utterance fits the lockstep decoder)", b, sl, TACO_BATCH_NB - 1);
      if (!((h->bat_left_slots >> sl) & 1u))
        TACO_FAIL(TACO_EINVAL, "taco_decode_resumable: resume_slot[%d] = %d, but the handle's last lockstep launch left no utterance in that slot", b, sl);
      taken |= 1u << sl;
      resuming = true;
    }
  }
  for (int b = 0; b < B; ++b) {
    const int own = host_lens.empty() ? T_in : host_lens[b];
    if (h->bat_ok && own <= TACO_BT_TIN) fits.push_back(b);
    else rest.push_back(b);
  }
  if (resuming || leave_unfinished) min_group = 1;      // saved state lives in the lockstep kernel only
  if ((int)fits.size() < min_group) {
    rest.insert(rest.end(), fits.begin(), fits.end());
    fits.clear();
  }
  int n_launches = 0;
  if (!fits.empty()) {
    h->bat_left_slots = 0;      // whatever an earlier launch left is taken up now or gone (the launch overwrites the saved state)
    // one launch for all of them: the kernel itself walks the utterances in index order, skipping those beyond its capacity
    const int rc = taco_launch_decoder_batch(h, memory, processed_memory, T_in, lens, B, (int)fits.size(), n_launches, masks, T_cap,
                                             max_steps, gate_threshold, forced_mel, mel, gate, align, n_frames, reached_max,
                                             resuming ? resume_slot : nullptr, leave_unfinished ? 1 : 0, slot_out, s);
    if (rc != TACO_OK) return rc;
    ++n_launches;
  }
  for (int b : rest) {
    const int rc = taco_launch_decoder(
        h, memory + (size_t)b * T_in * d.M, processed_memory + (size_t)b * T_in * d.A, T_in, lens, b, n_launches,
        host_lens.empty() ? T_in : host_lens[b], masks, T_cap,
        max_steps, gate_threshold, forced_mel ? forced_mel + (size_t)b * d.n_mel * T_cap : nullptr,
        mel + (size_t)b * d.n_mel * T_cap, gate + (size_t)b * T_cap, align ? align + (size_t)b * T_cap * T_in : nullptr,
        n_frames + b, reached_max + b, s);
    if (rc != TACO_OK) return rc;
    ++n_launches;
  }
  const int crc = taco_decoder_collect(h, n_launches, s);
  if (crc != TACO_OK) return crc;
  // a lockstep launch (always the call's first) that left utterances unfinished: keep the prenet output they were about to consume
  if (leave_unfinished && !fits.empty() && h->dec_ctl_host[0].reached_max == 2u) {
    const int rc = taco_decoder_batch_keep_prenet_output(h, (int)h->dec_ctl_host[0].n_frames, s);
    if (rc == TACO_OK) h->bat_left_slots = h->dec_ctl_host[0].left_slots;
    return rc;
  }
  return TACO_OK;
}

extern "C" int taco_postnet(taco_handle h, const float* mel, const int32_t* n_frames, int32_t B, int32_t T_cap,
                            float* out, void* stream) {
  if (!h || !mel || !out) TACO_FAIL(TACO_EINVAL, "taco_postnet: NULL argument");
  if (B < 1 || T_cap < 1) TACO_FAIL(TACO_EINVAL, "taco_postnet: B = %d, T_cap = %d", B, T_cap);
  hipStream_t s = (hipStream_t)stream;
  const taco_dims& d = h->d;
  DeviceScope on_device(h->device);
  TACO_HIP(on_device.status);
  const int cmax = d.C_post > d.n_mel ? d.C_post : d.n_mel;
  int rc = ensure_act(h, (size_t)B * cmax * T_cap);
  if (rc != TACO_OK) return rc;
  const float* cur_in = mel;
  int cur = 0;
  for (int l = 0; l < d.n_post; ++l) {
    const bool last = l == d.n_post - 1;
    const int c_in = l == 0 ? d.n_mel : d.C_post, c_out = last ? d.n_mel : d.C_post;
    float* dst = last ? out : h->ws_act[cur];
    rc = taco_launch_conv(cur_in, dst, h->w.post_conv_w[l], h->w.post_conv_b[l], h->post_alpha[l], h->post_beta[l],
                          last ? mel : nullptr, n_frames, B, c_in, c_out, d.k_post, T_cap, last ? 0 : /*tanh*/ 2, 0, h->conv_split,
                          TACO_CONV_SPLIT_CAP, h->n_cu, s);
    if (rc != TACO_OK) return rc;
    cur_in = dst;
    cur ^= 1;
  }
  return TACO_OK;
}

extern "C" int taco_postnet_noalloc(taco_handle h, const float* mel, const int32_t* n_frames, int32_t B, int32_t T_cap,
                                    float* out, void* stream) {
  if (!h || !mel || !out) TACO_FAIL(TACO_EINVAL, "taco_postnet_noalloc: NULL argument");
  if (B < 1 || T_cap < 1) TACO_FAIL(TACO_EINVAL, "taco_postnet_noalloc: B = %d, T_cap = %d", B, T_cap);
  hipStream_t s = (hipStream_t)stream;
  const taco_dims& d = h->d;
  const int cmax = d.C_post > d.n_mel ? d.C_post : d.n_mel;
  if ((uint64_t)B * (uint64_t)cmax * (uint64_t)T_cap > (uint64_t)TACO_NOALLOC_HALF)
    TACO_FAIL(TACO_EINVAL, "taco_postnet_noalloc: B * channels * T_cap = %d * %d * %d exceeds the built-in scratch of %u floats per "
                           "buffer (%d columns of 512 channels); use taco_postnet", B, cmax, T_cap, TACO_NOALLOC_HALF, TACO_NOALLOC_COLS);
  DeviceScope on_device(h->device);
  TACO_HIP(on_device.status);
  float* scratch = (h->device >= 0 && h->device < 64) ? g_noalloc_ptr[h->device] : nullptr;
  if (!scratch) TACO_FAIL(TACO_EINVAL, "taco_postnet_noalloc: no scratch address for device %d", h->device);
  float* buf[2] = {scratch, scratch + TACO_NOALLOC_HALF};
  // partial tiles and their counters in the built-in array too: the call stays off every workspace of the handle, so it
  // may overlap taco_encode / taco_postnet of the same handle on another stream
  float* split_ws = scratch + 2u * TACO_NOALLOC_HALF;
  const float* cur_in = mel;
  int cur = 0;
  for (int l = 0; l < d.n_post; ++l) {
    const bool last = l == d.n_post - 1;
    const int c_in = l == 0 ? d.n_mel : d.C_post, c_out = last ? d.n_mel : d.C_post;
    float* dst = last ? out : buf[cur];
    // every layer writes all B x c_out x T_cap elements of its output (zeros at or past n_frames[b]), so nothing of the
    // scratch's previous contents is ever read
    const int rc = taco_launch_conv(cur_in, dst, h->w.post_conv_w[l], h->w.post_conv_b[l], h->post_alpha[l], h->post_beta[l],
                                    last ? mel : nullptr, n_frames, B, c_in, c_out, d.k_post, T_cap, last ? 0 : /*tanh*/ 2, 0, split_ws,
                                    TACO_CONV_SPLIT_CAP_BUILTIN, h->n_cu, s);
    if (rc != TACO_OK) return rc;
    cur_in = dst;
    cur ^= 1;
  }
  return TACO_OK;
}

extern "C" int taco_set_frame_clock(taco_handle h, uint64_t* ticks, int32_t capacity) {
  if (!h) TACO_FAIL(TACO_EINVAL, "null handle");
  if (ticks && capacity < 1) TACO_FAIL(TACO_EINVAL, "frame clock capacity %d", capacity);
  h->frame_clock = reinterpret_cast<unsigned long long*>(ticks);
  h->frame_clock_cap = ticks ? capacity : 0;
  return TACO_OK;
}

extern "C" int taco_set_precision(taco_handle h, int32_t precision) {
  if (!h) TACO_FAIL(TACO_EINVAL, "null handle");
  if (precision != TACO_PRECISION_F32 && precision != TACO_PRECISION_BF16) TACO_FAIL(TACO_EINVAL, "unknown precision %d", precision);
  h->bat_bf16 = precision == TACO_PRECISION_BF16 ? 1 : 0;
  return TACO_OK;
}

extern "C" int taco_last_decode_timing(taco_handle h, double* total_us, int32_t* frames) {
  if (!h || !h->timing_valid) TACO_FAIL(TACO_EINVAL, "no decode has been timed on this handle");
  TACO_HIP(hipEventSynchronize(h->ev_stop));
  float ms = 0.0f;
  TACO_HIP(hipEventElapsedTime(&ms, h->ev_start, h->ev_stop));
  if (total_us) *total_us = (double)ms * 1000.0;
  if (frames) *frames = h->last_frames;
  return TACO_OK;
}
