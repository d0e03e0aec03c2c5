// Microbenchmark of the decoder's inter-workgroup hand-off: 256 resident workgroups, each publishes
// PER_WG tagged granules per round and then collects all N = 256*PER_WG of them; rounds are chained
// (round r+1's values depend on round r's), like the dependent steps of a decoder frame.
// Prints microseconds per round for several polling variants.
//   hipcc -O3 --offload-arch=gfx950 -o hop_bench hop_bench.hip && ./hop_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Args {
  u64* gran;        // [2 parities][replicas][N]
  int n;            // granules per vector
  int per_wg;       // granules each workgroup publishes
  int rounds;
  int mode;         // 0: every thread polls its own granules; 1: one wave polls everything;
                    // 2: poll a 1/16 sample first, then one full pass (re-polling stragglers)
  int replicas;     // copies each producer writes; consumer b reads copy b % replicas
  int sleep;        // s_sleep argument between polls
  int nthreads;
  float* out;
  unsigned* fail;
};

__device__ __forceinline__ float wait_one(const u64* base, int idx, unsigned epoch, int sl, unsigned* fail) {
  const gu64* g = (const gu64*)base + idx;
  u64 w = __hip_atomic_load(g, RLX_AGENT);
  unsigned spins = 0;
  while ((unsigned)(w >> 32) != epoch) {
    if (sl == 1) __builtin_amdgcn_s_sleep(1);
    else if (sl == 2) __builtin_amdgcn_s_sleep(2);
    else if (sl == 4) __builtin_amdgcn_s_sleep(4);
    else if (sl == 8) __builtin_amdgcn_s_sleep(8);
    else if (sl == 16) __builtin_amdgcn_s_sleep(16);
    if ((++spins & 1023u) == 0u) {
      if (__hip_atomic_load((__attribute__((address_space(1))) unsigned*)fail, RLX_AGENT) != 0u) break;
      if (spins > 1000000u) { __hip_atomic_store((__attribute__((address_space(1))) unsigned*)fail, 1u, RLX_AGENT); break; }
    }
    w = __hip_atomic_load(g, RLX_AGENT);
  }
  return __uint_as_float((unsigned)w);
}

__global__ __launch_bounds__(512) void hop_kernel(const Args a) {
  __shared__ float vec[4096];
  const int tid = threadIdx.x, j = blockIdx.x;
  float carry = (float)j;
  for (int r = 0; r < a.rounds; ++r) {
    const unsigned ep = (unsigned)r + 1u;
    u64* g = a.gran + (size_t)(r & 1) * a.replicas * a.n;
    if (tid < a.per_wg) {
      const float v = carry * 0.5f + (float)tid;
      const u64 word = ((u64)ep << 32) | (u64)__float_as_uint(v);
      for (int c = 0; c < a.replicas; ++c)
        __hip_atomic_store((gu64*)g + (size_t)c * a.n + j * a.per_wg + tid, word, RLX_AGENT);
    }
    const u64* src = g + (size_t)(j % a.replicas) * a.n;
    if (a.mode == 0) {
      for (int idx = tid; idx < a.n; idx += a.nthreads) vec[idx] = wait_one(src, idx, ep, a.sleep, a.fail);
    } else if (a.mode == 1) {
      if (tid < 64)
        for (int idx = tid; idx < a.n; idx += 64) vec[idx] = wait_one(src, idx, ep, a.sleep, a.fail);
    } else {
      if (tid < a.n / 16) (void)wait_one(src, tid * 16 + (j & 15), ep, a.sleep, a.fail);
      __syncthreads();
      for (int idx = tid; idx < a.n; idx += a.nthreads) vec[idx] = wait_one(src, idx, ep, a.sleep, a.fail);
    }
    __syncthreads();
    carry = vec[(j * 7 + r) % a.n] * 0.25f + vec[(j + 1) % a.n] * 0.125f;
    __syncthreads();
  }
  if (tid == 0) a.out[j] = carry;
}

int main(int argc, char** argv) {
  const int n_wg = 256;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("device %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
  u64* gran; float* out; unsigned* fail;
  const size_t max_gran = (size_t)2 * 8 * 4096;
  hipMalloc((void**)&gran, max_gran * 8);
  hipMalloc((void**)&out, n_wg * 4);
  hipMalloc((void**)&fail, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  struct Cfg { int per_wg, mode, replicas, sleep, nthreads; const char* name; };
  const Cfg cfgs[] = {
      {4, 0, 1, 1, 512, "N=1024 all-threads poll, sleep 1"},
      {4, 0, 1, 0, 512, "N=1024 all-threads poll, no sleep"},
      {4, 0, 1, 4, 512, "N=1024 all-threads poll, sleep 4"},
      {4, 0, 1, 16, 512, "N=1024 all-threads poll, sleep 16"},
      {4, 1, 1, 1, 512, "N=1024 one wave polls, sleep 1"},
      {4, 2, 1, 1, 512, "N=1024 sample then full pass, sleep 1"},
      {4, 0, 2, 1, 512, "N=1024 all-threads, 2 replicas"},
      {4, 0, 8, 1, 512, "N=1024 all-threads, 8 replicas"},
      {4, 2, 8, 1, 512, "N=1024 sample+full, 8 replicas"},
      {1, 0, 1, 1, 512, "N=256 all-threads poll, sleep 1"},
      {1, 0, 8, 1, 512, "N=256 all-threads, 8 replicas"},
      {2, 0, 1, 1, 512, "N=512 all-threads poll, sleep 1"},
      {4, 0, 1, 1, 256, "N=1024, 256 threads/wg"},
      {4, 0, 1, 1, 1024, "N=1024, 1024 threads/wg"},
  };
  for (const Cfg& c : cfgs) {
    Args a;
    a.gran = gran; a.n = n_wg * c.per_wg; a.per_wg = c.per_wg; a.rounds = 2000; a.mode = c.mode;
    a.replicas = c.replicas; a.sleep = c.sleep; a.nthreads = c.nthreads; a.out = out; a.fail = fail;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemsetAsync(gran, 0, max_gran * 8, 0);
      hipMemsetAsync(fail, 0, 4, 0);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(hop_kernel, dim3(n_wg), dim3(c.nthreads), 0, 0, a);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    unsigned f = 0;
    hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
    printf("%-44s %7.3f us/round%s\n", c.name, best * 1000.0f / a.rounds, f ? "  (TIMEOUT)" : "");
    fflush(stdout);
  }
  return 0;
}
