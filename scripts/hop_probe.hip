// Latency probes behind the decoder's inter-workgroup hand-off on MI355X:
//   1. round trip of one dependent agent-scope load (what a single poll costs), for a word that was
//      last written by a workgroup on another XCD / on the same XCD / by nobody;
//   2. ping-pong between two workgroups (store -> remote poll sees it -> store back), workgroups on the
//      same XCD (blocks 0 and 8) and on different XCDs (blocks 0 and 1), with 1, 2 or 4 polls in flight;
//   3. the 256-workgroup all-gather round of hop_bench with staggered polls in flight.
//   hipcc -O3 --offload-arch=gfx950 -o hop_probe hop_probe.hip && ./hop_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define SPIN_CAP 2000000u

__device__ __forceinline__ u64 ld(const u64* p) { return __hip_atomic_load((const gu64*)p, RLX_AGENT); }
__device__ __forceinline__ void st(u64* p, u64 v) { __hip_atomic_store((gu64*)p, v, RLX_AGENT); }

// ---- 1. dependent loads -----------------------------------------------------------------------
__global__ void chase_kernel(u64* words, int writer_block, int reader_block, int n, u64* out_clk) {
  if ((int)blockIdx.x == writer_block && threadIdx.x == 0) {
    for (int i = 0; i < 64; ++i) st(words + i * 16, 0);     // 64 words on 64 lines, all zero
  }
  if ((int)blockIdx.x == reader_block && threadIdx.x == 0) {
    for (volatile int spin = 0; spin < 20000; ++spin) {}     // let the writer's stores land
    u64 idx = 0;
    const u64 t0 = wall_clock64();
    for (int i = 0; i < n; ++i) idx = ld(words + ((idx + i) & 63) * 16);   // value 0: address depends on it
    const u64 t1 = wall_clock64();
    out_clk[0] = t1 - t0;
    out_clk[1] = idx;
  }
}

// ---- 2. ping-pong -----------------------------------------------------------------------------
template <int INFLIGHT>
__device__ __forceinline__ bool wait_tag(const u64* p, unsigned want, int stagger) {
  u64 w[INFLIGHT];
#pragma unroll
  for (int i = 0; i < INFLIGHT; ++i) {
    w[i] = ld(p);
    if (i + 1 < INFLIGHT) {
      if (stagger == 2) __builtin_amdgcn_s_sleep(2);
      else if (stagger == 4) __builtin_amdgcn_s_sleep(4);
      else if (stagger == 8) __builtin_amdgcn_s_sleep(8);
    }
  }
  unsigned spins = 0;
  for (;;) {
    if ((unsigned)(w[0] >> 32) == want) return true;
#pragma unroll
    for (int i = 0; i + 1 < INFLIGHT; ++i) w[i] = w[i + 1];
    w[INFLIGHT - 1] = ld(p);
    if (++spins > SPIN_CAP) return false;
  }
}

template <int INFLIGHT>
__global__ void pingpong_kernel(u64* a, u64* b, int peer_block, int rounds, int stagger, u64* out_clk) {
  const int me = blockIdx.x;
  if (threadIdx.x != 0 || (me != 0 && me != peer_block)) return;
  const u64 t0 = wall_clock64();
  bool ok = true;
  for (int r = 1; r <= rounds && ok; ++r) {
    if (me == 0) {
      st(a, (u64)r << 32);
      ok = wait_tag<INFLIGHT>(b, (unsigned)r, stagger);
    } else {
      ok = wait_tag<INFLIGHT>(a, (unsigned)r, stagger);
      st(b, (u64)r << 32);
    }
  }
  if (me == 0) { out_clk[0] = wall_clock64() - t0; out_clk[1] = ok ? 1 : 0; }
}

__device__ __forceinline__ u64 ld_sys(const u64* p) { return __hip_atomic_load((const gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(u64* p, u64 v) { __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// ---- 3. all-gather rounds: dense layout vs one 128-byte line per producer ---------------------------
struct GArgs {
  u64* gran; int n_wg, per_wg, rounds, replicas, line_per_producer, pollers, n_prod, sys_scope; float* out; unsigned* fail;
};

__device__ __forceinline__ float wait_val(const u64* p, unsigned ep, unsigned* fail, int sys_scope = 0) {
  u64 w = sys_scope ? ld_sys(p) : ld(p);
  unsigned spins = 0;
  while ((unsigned)(w >> 32) != ep) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 1023u) == 0u) {
      if (__hip_atomic_load((__attribute__((address_space(1))) unsigned*)fail, RLX_AGENT) != 0u) return 0.0f;
      if (spins > SPIN_CAP) { __hip_atomic_store((__attribute__((address_space(1))) unsigned*)fail, 1u, RLX_AGENT); return 0.0f; }
    }
    w = sys_scope ? ld_sys(p) : ld(p);
  }
  return __uint_as_float((unsigned)w);
}

// slot of granule k of producer j inside one replica
__device__ __forceinline__ int slot_of(const GArgs& a, int j, int k) {
  return a.line_per_producer ? j * 16 + k : j * a.per_wg + k;
}

__global__ __launch_bounds__(512) void gather_kernel(const GArgs a) {
  __shared__ float vec[4096];
  const int tid = threadIdx.x, j = blockIdx.x;
  const int n = a.n_prod * a.per_wg;
  const int rep_stride = a.line_per_producer ? a.n_wg * 16 : n;
  float carry = (float)j;
  for (int r = 0; r < a.rounds; ++r) {
    const unsigned ep = (unsigned)r + 1u;
    u64* g = a.gran + (size_t)(r & 1) * a.replicas * rep_stride;
    if (j < a.n_prod && tid < a.per_wg * a.replicas) {       // one lane per (granule, replica): a single store instruction
      const int k = tid % a.per_wg, c = tid / a.per_wg;
      const float v = carry * 0.5f + (float)k;
      const u64 word = ((u64)ep << 32) | (u64)__float_as_uint(v);
      if (a.sys_scope & 2) st_sys(g + (size_t)c * rep_stride + slot_of(a, j, k), word);
      else st(g + (size_t)c * rep_stride + slot_of(a, j, k), word);
    }
    const u64* src = g + (size_t)(j % a.replicas) * rep_stride;
    if (tid < a.pollers)
      for (int idx = tid; idx < n; idx += a.pollers)
        vec[idx] = wait_val(src + slot_of(a, idx / a.per_wg, idx % a.per_wg), ep, a.fail, a.sys_scope & 1);
    __syncthreads();
    carry = vec[(j * 7 + r) % n] * 0.25f + vec[(j + 1) % n] * 0.125f;
    __syncthreads();
  }
  if (tid == 0) a.out[j] = carry;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("device %s, %d CUs, wall clock 100 MHz\n", prop.gcnArchName, prop.multiProcessorCount);
  u64 *words, *clk; float* out; unsigned* fail;
  hipMalloc((void**)&words, 1 << 20);
  hipMalloc((void**)&clk, 64);
  hipMalloc((void**)&out, 1024);
  hipMalloc((void**)&fail, 4);
  u64 h[2];

  const int n_chase = 2000;
  struct { int w, r; const char* name; } chase[] = {{1, 0, "written on another XCD"}, {8, 0, "written on the same XCD"}, {0, 0, "written by the reader"}};
  for (auto& c : chase) {
    hipMemset(words, 0, 1 << 20);
    hipLaunchKernelGGL(chase_kernel, dim3(16), dim3(64), 0, 0, words, c.w, c.r, n_chase, clk);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("dependent agent-scope load, word %-26s %7.1f ns\n", c.name, (double)h[0] * 10.0 / n_chase);
  }

  const int rounds = 2000;
  struct { int peer, inflight, stagger; const char* name; } pp[] = {
      {1, 1, 0, "other XCD, 1 poll in flight"}, {8, 1, 0, "same XCD,  1 poll in flight"},
      {1, 2, 4, "other XCD, 2 polls, stagger 4"}, {1, 4, 2, "other XCD, 4 polls, stagger 2"},
      {1, 4, 4, "other XCD, 4 polls, stagger 4"}, {8, 4, 2, "same XCD,  4 polls, stagger 2"}};
  for (auto& c : pp) {
    hipMemset(words, 0, 4096);
    u64* a = words; u64* b = words + 64;
    if (c.inflight == 1) hipLaunchKernelGGL(pingpong_kernel<1>, dim3(16), dim3(64), 0, 0, a, b, c.peer, rounds, c.stagger, clk);
    else if (c.inflight == 2) hipLaunchKernelGGL(pingpong_kernel<2>, dim3(16), dim3(64), 0, 0, a, b, c.peer, rounds, c.stagger, clk);
    else hipLaunchKernelGGL(pingpong_kernel<4>, dim3(16), dim3(64), 0, 0, a, b, c.peer, rounds, c.stagger, clk);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("ping-pong %-32s one way %7.1f ns%s\n", c.name, (double)h[0] * 10.0 / rounds / 2.0, h[1] ? "" : "  (TIMEOUT)");
  }

  struct { int per_wg, replicas, n_prod, scope, mem; const char* name; } gc[] = {
      {4, 2, 256, 0, 0, "256x4 R=2 agent/agent, hipMalloc"}, {4, 2, 256, 3, 0, "256x4 R=2 system/system, hipMalloc"},
      {4, 2, 256, 1, 0, "256x4 R=2 agent store, system load"}, {4, 2, 256, 0, 1, "256x4 R=2 agent, fine-grained"},
      {4, 2, 256, 3, 1, "256x4 R=2 system, fine-grained"}, {4, 2, 256, 0, 2, "256x4 R=2 agent, uncached"},
      {4, 2, 256, 3, 2, "256x4 R=2 system, uncached"},
      {1, 8, 256, 0, 0, "256x1 R=8 agent, hipMalloc"}, {1, 8, 256, 0, 1, "256x1 R=8 agent, fine-grained"},
      {1, 8, 256, 0, 2, "256x1 R=8 agent, uncached"}, {1, 1, 256, 0, 2, "256x1 R=1 agent, uncached"},
      {4, 1, 256, 0, 2, "256x4 R=1 agent, uncached"}, {8, 8, 32, 0, 2, "32x8 R=8 agent, uncached"}, {8, 1, 32, 0, 2, "32x8 R=1 agent, uncached"}};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t max_gran = (size_t)2 * 8 * 4096;
  u64* gmem[3] = {nullptr, nullptr, nullptr};
  hipMalloc((void**)&gmem[0], max_gran * 8);
  if (hipExtMallocWithFlags((void**)&gmem[1], max_gran * 8, hipDeviceMallocFinegrained) != hipSuccess) gmem[1] = nullptr;
  if (hipExtMallocWithFlags((void**)&gmem[2], max_gran * 8, hipDeviceMallocUncached) != hipSuccess) gmem[2] = nullptr;
  for (auto& c : gc) {
    GArgs a;
    if (!gmem[c.mem]) { printf("all-gather %-40s (allocation kind unavailable)\n", c.name); continue; }
    a.gran = gmem[c.mem]; a.n_wg = 256; a.per_wg = c.per_wg; a.rounds = 2000; a.replicas = c.replicas; a.line_per_producer = 0;
    a.pollers = 512; a.n_prod = c.n_prod; a.sys_scope = c.scope; a.out = out; a.fail = fail;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemsetAsync(a.gran, 0, max_gran * 8, 0);
      hipMemsetAsync(fail, 0, 4, 0);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(gather_kernel, dim3(256), dim3(512), 0, 0, a);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    unsigned f = 0;
    hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
    printf("all-gather %-40s %7.3f us/round%s\n", c.name, best * 1000.0f / a.rounds, f ? "  (TIMEOUT)" : "");
    fflush(stdout);
  }
  return 0;
}
