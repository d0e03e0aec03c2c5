// Cost of a frame-like chain of dependent all-gathers between resident workgroups when the
// producer and consumer sets of each hop differ (MI355X, 256 workgroups x 512 threads).
// A hop h has producers [0, np_h) x per_h granules and consumers [0, nc_h); a workgroup waits only
// where it consumes and publishes only after everything it consumed before.  Every cycle ends with a
// hop that all workgroups consume, so nobody runs more than a cycle ahead (slots are double-buffered
// by cycle parity).
//   hipcc -O3 --offload-arch=gfx950 -o chain_probe chain_probe.hip && ./chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define MAX_HOPS 8
#define REPL 8

struct Hop { int n_prod, per, n_cons; };
struct Args { u64* gran; Hop hop[MAX_HOPS]; int n_hops, cycles; float* out; unsigned* fail; };

__device__ __forceinline__ float wait_val(const u64* p, unsigned ep, unsigned* fail) {
  u64 w = __hip_atomic_load((const gu64*)p, RLX_AGENT);
  unsigned spins = 0;
  while ((unsigned)(w >> 32) != ep) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 1023u) == 0u) {
      if (__hip_atomic_load((__attribute__((address_space(1))) unsigned*)fail, RLX_AGENT) != 0u) return 0.0f;
      if (spins > 2000000u) { __hip_atomic_store((__attribute__((address_space(1))) unsigned*)fail, 1u, RLX_AGENT); return 0.0f; }
    }
    w = __hip_atomic_load((const gu64*)p, RLX_AGENT);
  }
  return __uint_as_float((unsigned)w);
}

__global__ __launch_bounds__(512) void chain_kernel(const Args a) {
  __shared__ float vec[4096];
  const int tid = threadIdx.x, j = blockIdx.x;
  float carry = (float)j;
  for (int c = 0; c < a.cycles; ++c) {
    const unsigned ep = (unsigned)c + 1u;
    for (int h = 0; h < a.n_hops; ++h) {
      const Hop hp = a.hop[h];
      const int n = hp.n_prod * hp.per;
      u64* g = a.gran + ((size_t)(c & 1) * MAX_HOPS + h) * REPL * 4096;
      if (j < hp.n_prod)
        for (int it = tid; it < hp.per * REPL; it += 512) {
          const int k = it % hp.per, r = it / hp.per;
          const float v = carry * 0.5f + (float)k;
          __hip_atomic_store((gu64*)(g + (size_t)r * 4096 + j * hp.per + k), ((u64)ep << 32) | (u64)__float_as_uint(v), RLX_AGENT);
        }
      if (j < hp.n_cons) {
        const u64* src = g + (size_t)(j % REPL) * 4096;
        for (int idx = tid; idx < n; idx += 512) vec[idx] = wait_val(src + idx, ep, a.fail);
        __syncthreads();
        carry = vec[(j * 7 + c) % n] * 0.25f + vec[(j + 1) % n] * 0.125f;
        __syncthreads();
      }
    }
  }
  if (tid == 0) a.out[j] = carry;
}

int main() {
  u64* gran; float* out; unsigned* fail;
  const size_t n_gran = (size_t)2 * MAX_HOPS * REPL * 4096;
  hipMalloc((void**)&gran, n_gran * 8);
  hipMalloc((void**)&out, 1024);
  hipMalloc((void**)&fail, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  struct Cfg { const char* name; int n; Hop h[MAX_HOPS]; } cfgs[] = {
      {"6 hops, all produce / all consume (256x1,256x4,128x1,102x1,256x4,81x1)", 6,
       {{256, 1, 256}, {256, 4, 256}, {128, 1, 256}, {102, 1, 256}, {256, 4, 256}, {81, 1, 256}}},
      {"6 hops, concentrated producers, all consume (32x8,256x4,32x4,13x8,256x4,27x3)", 6,
       {{32, 8, 256}, {256, 4, 256}, {32, 4, 256}, {13, 8, 256}, {256, 4, 256}, {27, 3, 256}}},
      {"6 hops, concentrated producers, only the needed consumers", 6,
       {{32, 8, 256}, {256, 4, 256}, {32, 4, 13}, {13, 8, 256}, {256, 4, 256}, {27, 3, 256}}},
      {"5 hops: dh 256x4->32, mel 27x3->32, p 32x8->256, ah 256x4->256, pe 8x102->256", 5,
       {{256, 4, 32}, {27, 3, 32}, {32, 8, 256}, {256, 4, 256}, {8, 102, 256}}},
      {"5 hops: same with ah consumed by 8 only (+ final all)", 5,
       {{256, 4, 32}, {27, 3, 32}, {32, 8, 256}, {256, 4, 8}, {8, 102, 256}}},
      {"1 hop 256x4 -> 256", 1, {{256, 4, 256}}},
      {"1 hop 256x1 -> 256", 1, {{256, 1, 256}}},
      {"2 hops 256x4 -> 8, 8x102 -> 256", 2, {{256, 4, 8}, {8, 102, 256}}},
      {"2 hops 256x4 -> 256, 8x102 -> 256", 2, {{256, 4, 256}, {8, 102, 256}}},
      {"2 hops 256x4 -> 32, 27x3 -> 256", 2, {{256, 4, 32}, {27, 3, 256}}},
      {"3 hops 256x4 -> 32, 27x3 -> 32, 32x8 -> 256", 3, {{256, 4, 32}, {27, 3, 32}, {32, 8, 256}}},
      {"3 hops 256x4 -> 256, 81x1 -> 256, 256x1 -> 256", 3, {{256, 4, 256}, {81, 1, 256}, {256, 1, 256}}},
  };
  for (const Cfg& c : cfgs) {
    Args a;
    a.gran = gran; a.n_hops = c.n; a.cycles = 1500; a.out = out; a.fail = fail;
    for (int h = 0; h < c.n; ++h) a.hop[h] = c.h[h];
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemsetAsync(gran, 0, n_gran * 8, 0);
      hipMemsetAsync(fail, 0, 4, 0);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(chain_kernel, dim3(256), dim3(512), 0, 0, a);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    unsigned f = 0;
    hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
    printf("%-86s %7.3f us/cycle%s\n", c.name, best * 1000.0f / a.cycles, f ? "  (TIMEOUT)" : "");
    fflush(stdout);
  }
  return 0;
}
