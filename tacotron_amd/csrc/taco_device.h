// Device-side building blocks: wave reduction and the tagged-granule hand-off between workgroups.
#pragma once

#include <hip/hip_runtime.h>

#include "taco_internal.h"

typedef __attribute__((address_space(1))) taco_granule taco_ggran;   // global-memory granule
typedef __attribute__((address_space(1))) unsigned int taco_gu32;

#define TACO_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// Abort codes written to the control block.
#define TACO_ABORT_TIMEOUT 1u

// exp / sigmoid / tanh on the hardware exp2 and reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each).  The
// product x*log2(e) is carried in two floats so the exponent argument is exact to ~2^-48; results
// agree with libm's to about 2 ulp (tanh: 2e-7 absolute), far inside the parity tolerance, at a
// fraction of the instruction count of the library calls.
__device__ __forceinline__ float taco_exp(float x) {
  const float hi = x * 1.44269502e+0f;
  float lo = fmaf(x, 1.44269502e+0f, -hi);
  lo = fmaf(x, 1.92596299e-8f, lo);
  const float r = __builtin_amdgcn_exp2f(hi);
  return fmaf(r, lo * 6.93147182e-1f, r);
}
__device__ __forceinline__ float taco_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + taco_exp(-x)); }
__device__ __forceinline__ float taco_tanh(float x) {
  const float ax = fminf(fabsf(x), 15.0f);
  const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(taco_exp(2.0f * ax) + 1.0f);
  return copysignf(t, x);
}

// Wave-wide (64-lane) reductions; every lane receives the result.  The default form rotates and
// combines inside each row of 16 lanes with DPP (row_ror:8,4,2,1 - no LDS crossbar traffic) and then
// combines the four row results read with v_readlane; -DTACO_SHFL_REDUCE selects the plain
// __shfl_xor butterfly instead (same result up to the association order of the additions).
#if !defined(TACO_SHFL_REDUCE)
template <int CTRL>
__device__ __forceinline__ float taco_dpp_rot(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float taco_lane_value(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float taco_wave_sum(float v) {
  v += taco_dpp_rot<0x128>(v);
  v += taco_dpp_rot<0x124>(v);
  v += taco_dpp_rot<0x122>(v);
  v += taco_dpp_rot<0x121>(v);
  return (taco_lane_value(v, 0) + taco_lane_value(v, 16)) + (taco_lane_value(v, 32) + taco_lane_value(v, 48));
}
// Two wave-wide sums for the price of one: v_permlane32_swap puts the upper half of `a` beside the lower half of
// `b`, so that one addition leaves a's lanes folded onto lanes 0..31 and b's onto lanes 32..63; the rotations inside
// the rows of 16 and two v_readlane per value finish both.
__device__ __forceinline__ void taco_wave_sum2(float a, float b, float& sum_a, float& sum_b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  float v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  v += taco_dpp_rot<0x128>(v);
  v += taco_dpp_rot<0x124>(v);
  v += taco_dpp_rot<0x122>(v);
  v += taco_dpp_rot<0x121>(v);
  sum_a = taco_lane_value(v, 0) + taco_lane_value(v, 16);
  sum_b = taco_lane_value(v, 32) + taco_lane_value(v, 48);
}
// Sum over each aligned group of 8 lanes (quad swaps, then the half-row mirror); every lane of a
// group receives the group's total.
__device__ __forceinline__ float taco_sum8(float v) {
  v += taco_dpp_rot<0xB1>(v);    // quad_perm [1,0,3,2]
  v += taco_dpp_rot<0x4E>(v);    // quad_perm [2,3,0,1]
  v += taco_dpp_rot<0x141>(v);   // row_half_mirror
  return v;
}
// Sum over each row of 16 lanes / each quad; every lane of the group receives the total.
__device__ __forceinline__ float taco_sum16(float v) {
  v += taco_dpp_rot<0x128>(v);
  v += taco_dpp_rot<0x124>(v);
  v += taco_dpp_rot<0x122>(v);
  v += taco_dpp_rot<0x121>(v);
  return v;
}
__device__ __forceinline__ float taco_sum4(float v) {
  v += taco_dpp_rot<0xB1>(v);
  v += taco_dpp_rot<0x4E>(v);
  return v;
}
// Lane K of each quad, to all four lanes of the quad (quad_perm [K,K,K,K]).
template <int K>
__device__ __forceinline__ float taco_quad_bcast(float v) { return taco_dpp_rot<K * 0x55>(v); }
__device__ __forceinline__ float taco_wave_max(float v) {
  v = fmaxf(v, taco_dpp_rot<0x128>(v));
  v = fmaxf(v, taco_dpp_rot<0x124>(v));
  v = fmaxf(v, taco_dpp_rot<0x122>(v));
  v = fmaxf(v, taco_dpp_rot<0x121>(v));
  return fmaxf(fmaxf(taco_lane_value(v, 0), taco_lane_value(v, 16)),
               fmaxf(taco_lane_value(v, 32), taco_lane_value(v, 48)));
}
#else
__device__ __forceinline__ float taco_wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float taco_wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ void taco_wave_sum2(float a, float b, float& sum_a, float& sum_b) {
  sum_a = taco_wave_sum(a);
  sum_b = taco_wave_sum(b);
}
__device__ __forceinline__ float taco_sum8(float v) {
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  return v;
}
__device__ __forceinline__ float taco_sum16(float v) {
  v = taco_sum8(v);
  return v + __shfl_xor(v, 8, 64);
}
__device__ __forceinline__ float taco_sum4(float v) {
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  return v;
}
template <int K>
__device__ __forceinline__ float taco_quad_bcast(float v) { return __shfl(v, (threadIdx.x & 60) + K, 64); }
#endif

// ---- hand-off: one 8-byte word carries {epoch, fp32}; the data is its own flag -------------------
// A word leaves as an atomic exchange whose result nobody looks at (global_atomic_swap_x2 without return),
// not as a store: the exchange is carried out at the memory side right away, and the frame came out
// 0.07 us shorter than with write-through stores.
__device__ __forceinline__ void taco_send(taco_ggran* where, taco_granule word) {
  (void)__hip_atomic_exchange(where, word, TACO_RLX_AGENT);
}
__device__ __forceinline__ void taco_publish(taco_granule* base, int idx, unsigned epoch, float value) {
  const taco_granule word = ((taco_granule)epoch << 32) | (taco_granule)__float_as_uint(value);
  taco_send((taco_ggran*)base + idx, word);
}

// A store's address and data registers are read by the memory pipeline some time after the store has been
// issued, and the compiler makes a wave wait for ALL its outstanding vector-memory operations (the
// acknowledgement of the write-through store included, which takes longer than a poll) before it lets the
// wave overwrite them.  A publish on the critical chain therefore keeps both alive - taco_keep_alive()
// further down the frame - so that the register allocator cannot hand them out again right after the store.
struct taco_sent {
  taco_ggran* where;
  taco_granule word;
};
__device__ __forceinline__ void taco_keep_alive(const taco_sent& s) { asm volatile("" ::"v"(s.where), "v"(s.word)); }
// One value to the first R replicas' slot idx, spread over the first R lanes of a wave that all hold `value`.
template <int R>
__device__ __forceinline__ taco_sent taco_publish_lanes_kept(taco_granule* base, int cap, int idx, unsigned epoch, float value,
                                                             int lane) {
  taco_sent s;
  s.where = (taco_ggran*)base + (size_t)lane * cap + idx;
  s.word = ((taco_granule)epoch << 32) | (taco_granule)__float_as_uint(value);
  if (lane < R) taco_send(s.where, s.word);
  return s;
}

// Publishes one value to all R replicas of a vector (replica stride `cap` granules).
template <int R>
__device__ __forceinline__ void taco_publish_rep(taco_granule* base, int cap, int idx, unsigned epoch, float value) {
  const taco_granule word = ((taco_granule)epoch << 32) | (taco_granule)__float_as_uint(value);
#pragma unroll
  for (int r = 0; r < R; ++r) taco_send((taco_ggran*)base + (size_t)r * cap + idx, word);
}

// The same, spread over the first R lanes of a wave that all hold `value`: one store instruction.
template <int R>
__device__ __forceinline__ void taco_publish_lanes(taco_granule* base, int cap, int idx, unsigned epoch, float value,
                                                   int lane) {
  const taco_granule word = ((taco_granule)epoch << 32) | (taco_granule)__float_as_uint(value);
  if (lane < R) taco_send((taco_ggran*)base + (size_t)lane * cap + idx, word);
}

#ifndef TACO_POLL_SLEEP
#define TACO_POLL_SLEEP 1        // x 64 clocks between two polls of a granule that has not arrived
#endif

struct taco_wait_ctx {
  taco_gu32* abort_word;     // global abort flag
  int failed;                // sticky, per thread
  int spun;                  // set when a wait found its granule not there yet
};

// A wait gives up after this many polls (each poll is a round trip to memory of roughly a
// microsecond, so about a quarter of a second) or as soon as another wave has raised the abort flag.
constexpr unsigned TACO_SPIN_LIMIT = 1u << 18;

// Spins until granule idx carries `epoch`; returns the payload.  Gives up when the global abort
// flag is raised or the poll budget is spent (then raises the flag itself), so the grid always drains.
__device__ __forceinline__ float taco_wait_one(const taco_granule* base, int idx, unsigned epoch,
                                               taco_wait_ctx& ctx, unsigned info) {
  const taco_ggran* g = (const taco_ggran*)base + idx;
  taco_granule word = __hip_atomic_load(g, TACO_RLX_AGENT);
  unsigned spins = 0;
  while ((unsigned)(word >> 32) != epoch) {
    ctx.spun = 1;
    if (ctx.failed) break;
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
    word = __hip_atomic_load(g, TACO_RLX_AGENT);
  }
  return __uint_as_float((unsigned)word);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding
// global store of the wave (vmcnt(0)); after a publish that is a write-through store on its way to
// memory, and nothing here needs to wait for it: granule stores are consumed by other workgroups.
__device__ __forceinline__ void taco_lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// The same barrier spelled out, for places where the compiler's own version also waits for the wave's
// outstanding vector-memory operations (it does where a poll loop joins just before the barrier): the wave's
// publish store may still be on its way then, and its acknowledgement takes longer than a poll.
__device__ __forceinline__ void taco_lds_barrier_only() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Naps.  A poll that reaches memory just before the value does costs a whole further round trip, and
// polls on a line that is still being written delay the write itself.  The frames of an utterance repeat
// with nearly the same timing, so each wave keeps, per hand-off, the time it stays away from memory before
// its first poll and steers it by what the polls find: a value not there yet - nap longer next frame, by
// TACO_NAP_UP; all there at once - shorter, by TACO_NAP_DOWN.  The nap is kept in quarters of its unit,
// one s_sleep 1 (about 64 clocks); up / down = 16 : 1 keeps it near the late end of the arrival times
// (measured best among 4:1 .. 32:1).
#ifndef TACO_NAP_UP
#define TACO_NAP_UP 16
#endif
#ifndef TACO_NAP_DOWN
#define TACO_NAP_DOWN 1
#endif
constexpr int TACO_NAP_MAX = 4 * 120;      // about 3.5 us
__device__ __forceinline__ void taco_nap(int n) {
  for (int i = n >> 2; i > 0; --i) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ int taco_nap_adjust(int n, int spun) {
  const bool any = __builtin_amdgcn_readfirstlane(__any(spun) ? 1 : 0) != 0;
  return any ? min(n + TACO_NAP_UP, TACO_NAP_MAX) : max(n - TACO_NAP_DOWN, 0);
}

// Granules [0, n) into LDS, W (2 or 4) per thread at a time with all their first polls in flight together: a
// thread's later granules do not wait for the round trips of its earlier ones.  (After a nap the values are
// there as a rule; one that is not is waited for on its own.)
template <int NTHREADS, int W>
__device__ __forceinline__ void taco_collect_w(const taco_granule* base, int n, unsigned epoch, float* lds_dst,
                                               taco_wait_ctx& ctx, unsigned info) {
  const taco_ggran* g = (const taco_ggran*)base;
  for (int i0 = threadIdx.x; i0 < n; i0 += W * NTHREADS) {
    taco_granule w[W];
#pragma unroll
    for (int k = 0; k < W; ++k) w[k] = i0 + k * NTHREADS < n ? __hip_atomic_load(g + i0 + k * NTHREADS, TACO_RLX_AGENT) : 0;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const int idx = i0 + k * NTHREADS;
      if (idx < n) {
        const bool ok = (unsigned)(w[k] >> 32) == epoch;
        if (!ok) ctx.spun = 1;
        lds_dst[idx] = ok ? __uint_as_float((unsigned)w[k]) : taco_wait_one(base, idx, epoch, ctx, info);
      }
    }
  }
}
// Two at a time for vectors of up to two granules per thread (measured: the four-wide form costs 0.09 us there),
// four at a time for longer ones (saves 0.1 - 0.4 us on the energy partials of inputs over 128 positions).
template <int NTHREADS>
__device__ __forceinline__ void taco_collect(const taco_granule* base, int n, unsigned epoch, float* lds_dst,
                                             taco_wait_ctx& ctx, unsigned info) {
  if (n <= 2 * NTHREADS) taco_collect_w<NTHREADS, 2>(base, n, epoch, lds_dst, ctx, info);
  else taco_collect_w<NTHREADS, 4>(base, n, epoch, lds_dst, ctx, info);
}

// All threads of the workgroup collect granules [0, n) into LDS and synchronise.
template <int NTHREADS>
__device__ __forceinline__ void taco_gather(const taco_granule* base, int n, unsigned epoch, float* lds_dst,
                                            taco_wait_ctx& ctx, unsigned info) {
  for (int idx = threadIdx.x; idx < n; idx += NTHREADS) lds_dst[idx] = taco_wait_one(base, idx, epoch, ctx, info);
  taco_lds_barrier();
}
// The same after this wave's nap for the hand-off, which is adjusted by what the polls found.
template <int NTHREADS>
__device__ __forceinline__ void taco_gather(const taco_granule* base, int n, unsigned epoch, float* lds_dst,
                                            taco_wait_ctx& ctx, unsigned info, int& nap) {
  taco_nap(nap);
  ctx.spun = 0;
  taco_collect<NTHREADS>(base, n, epoch, lds_dst, ctx, info);
  nap = taco_nap_adjust(nap, ctx.spun);
  taco_lds_barrier();
}
