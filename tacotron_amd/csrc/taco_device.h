// Device-side building blocks: wave reduction and the tagged-granule hand-off between workgroups.
#pragma once

#include <hip/hip_runtime.h>

#include "taco_internal.h"

typedef __attribute__((address_space(1))) taco_granule taco_ggran;   // global-memory granule
typedef __attribute__((address_space(1))) unsigned int taco_gu32;

#define TACO_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// Abort codes written to the control block.
#define TACO_ABORT_TIMEOUT 1u

__device__ __forceinline__ float taco_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// Sum over the 64 lanes of a wave; every lane receives the total.
#if defined(TACO_DPP_REDUCE)
template <int CTRL>
__device__ __forceinline__ float taco_dpp_rot(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float taco_wave_sum(float v) {
  // rotate-and-add inside each row of 16 lanes (row_ror:8,4,2,1), then add the four row totals
  v += taco_dpp_rot<0x128>(v);
  v += taco_dpp_rot<0x124>(v);
  v += taco_dpp_rot<0x122>(v);
  v += taco_dpp_rot<0x121>(v);
  const int vi = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
  return (r0 + r1) + (r2 + r3);
}
#else
__device__ __forceinline__ float taco_wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
#endif

__device__ __forceinline__ float taco_wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}

// ---- hand-off: one 8-byte word carries {epoch, fp32}; the data is its own flag -------------------
__device__ __forceinline__ void taco_publish(taco_granule* base, int idx, unsigned epoch, float value) {
  const taco_granule word = ((taco_granule)epoch << 32) | (taco_granule)__float_as_uint(value);
  __hip_atomic_store((taco_ggran*)base + idx, word, TACO_RLX_AGENT);
}

struct taco_wait_ctx {
  taco_gu32* abort_word;     // global abort flag
  unsigned long long deadline;  // wall_clock64() value after which waits give up
  int failed;                // sticky, per thread
};

// Spins until granule idx carries `epoch`; returns the payload.  Gives up when the global abort
// flag is raised or the deadline passes (then raises the flag itself), so the grid always drains.
__device__ __forceinline__ float taco_wait_one(const taco_granule* base, int idx, unsigned epoch,
                                               taco_wait_ctx& ctx, unsigned info) {
  const taco_ggran* g = (const taco_ggran*)base + idx;
  taco_granule word = __hip_atomic_load(g, TACO_RLX_AGENT);
  unsigned spins = 0;
  while ((unsigned)(word >> 32) != epoch) {
    if (ctx.failed) break;
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 255u) == 0u) {
      if (__hip_atomic_load(ctx.abort_word, TACO_RLX_AGENT) != 0u) { ctx.failed = 1; break; }
      if (wall_clock64() > ctx.deadline) {
        __hip_atomic_store(ctx.abort_word, TACO_ABORT_TIMEOUT, TACO_RLX_AGENT);
        __hip_atomic_store(ctx.abort_word + 1, info, TACO_RLX_AGENT);
        ctx.failed = 1;
        break;
      }
    }
    word = __hip_atomic_load(g, TACO_RLX_AGENT);
  }
  return __uint_as_float((unsigned)word);
}

// All threads of the workgroup collect granules [0, n) into LDS and synchronise.
template <int NTHREADS>
__device__ __forceinline__ void taco_gather(const taco_granule* base, int n, unsigned epoch, float* lds_dst,
                                            taco_wait_ctx& ctx, unsigned info) {
  for (int idx = threadIdx.x; idx < n; idx += NTHREADS) lds_dst[idx] = taco_wait_one(base, idx, epoch, ctx, info);
  __syncthreads();
}
