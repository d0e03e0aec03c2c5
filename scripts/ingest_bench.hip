// Microbenchmark of the lockstep decoder's all-to-all exchange: 256 resident workgroups of 256 threads; per round each
// workgroup publishes its 16 x 16 bytes of a K x 16-utterance vector (tagged 4-byte words, write-through 16-byte stores,
// REP replicas) and then every workgroup takes in the whole vector as the decoder does (each wave a quarter of it,
// LOADS 16-byte loads per lane, all issued before the first is looked at; pieces not there yet are fetched again).
// Rounds are chained (a round's values depend on the previous round's sum).  Prints microseconds per round.
//   hipcc -O3 --offload-arch=gfx950 -o ingest_bench ingest_bench.hip && ./ingest_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int NWG = 256, NT = 256, NB = 16, KH = 1024, LOADS = KH / 16 / 4;   // 16 loads per wave and lane

struct Args {
  unsigned* words;      // [2 parities][replicas][KH * NB]
  unsigned bytes;
  int rounds, replicas, mode, rot, nap;
  float* out;
  unsigned* fail;
};

__device__ __forceinline__ unsigned word_at(int k, int u) { return (unsigned)(((k >> 4) * NB + u) * 16 + (k & 15)); }

template <int AUX>
__device__ __forceinline__ void run(const Args& a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = blockIdx.x;
  const int utt = lane & 15, kq = lane >> 4;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.words, 0, (int)a.bytes, 0x00020000);
  const unsigned cap = KH * NB;
  float carry = (float)(j & 7) * 0.125f;
  const int rot = a.rot ? (j * a.rot) & (LOADS - 1) : 0;
  for (int r = 0; r < a.rounds; ++r) {
    const unsigned tag = (((unsigned)r >> 1) & 1u) + 1u;
    const unsigned par = (unsigned)(r & 1) * (unsigned)a.replicas * cap;
    if (a.mode != 2 || r == 0) {
      if (wave == 0 && lane < 16) {
        const float v = carry + (float)lane;
        const unsigned w = (__float_as_uint(v) & ~3u) | (a.mode == 2 ? 0u : tag);
        const u32x4 q = {w, w, w, w};
        for (int c = 0; c < a.replicas; ++c)
          __builtin_amdgcn_raw_buffer_store_b128(q, rs, (par + c * cap + word_at(4 * j, lane)) * 4u, 0, 16);
      }
    }
    for (int n = 0; n < a.nap; n += 8) __builtin_amdgcn_s_sleep(8);
    const unsigned base = par + (unsigned)(j % a.replicas) * cap + word_at(16 * (wave * LOADS) + 4 * kq, utt);
    u32x4 v[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (base + 256u * ((i + rot) & (LOADS - 1))) * 4u, 0, AUX);
    unsigned spins = 0;
    for (;;) {
      unsigned bad = 0;
      if (a.mode != 2) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) bad |= (v[i].x ^ tag) | (v[i].y ^ tag) | (v[i].z ^ tag) | (v[i].w ^ tag);
      }
      if (__all((bad & 3u) == 0u)) break;
      if (++spins > 200000u) { atomicExch(a.fail, 1u); break; }
      __builtin_amdgcn_s_sleep(1);
#pragma unroll
      for (int i = 0; i < LOADS; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (base + 256u * ((i + rot) & (LOADS - 1))) * 4u, 0, AUX);
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) s += __uint_as_float(v[i].x) + __uint_as_float(v[i].w);
    carry = s * 1e-9f + carry * 0.5f;
    if (a.mode == 1) __syncthreads();
  }
  if (tid == 0) a.out[j] = carry;
}

__global__ __launch_bounds__(NT) void ingest_kernel(const Args a, int aux) {
  extern __shared__ float one_per_cu[];      // 96 KB of LDS asked for at launch: one workgroup per CU, as in the decoder
  if (a.rounds < 0) one_per_cu[threadIdx.x] = 0.0f;
  if (aux == 16) run<16>(a);
  else if (aux == 17) run<17>(a);
  else if (aux == 0) run<0>(a);
  else run<1>(a);
}

int main() {
  unsigned* words; float* out; unsigned* fail;
  const int max_rep = 8;
  const size_t bytes = (size_t)2 * max_rep * KH * NB * 4;
  hipMalloc(&words, bytes); hipMalloc(&out, NWG * 4); hipMalloc(&fail, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct V { const char* name; int replicas, mode, rot, nap, aux; };
  const V vs[] = {
    {"publish + gather, 2 replicas, sc1 loads (the decoder today)", 2, 0, 0, 0, 16},
    {"same, 1 replica", 1, 0, 0, 0, 16},
    {"same, 4 replicas", 4, 0, 0, 0, 16},
    {"same, 8 replicas", 8, 0, 0, 0, 16},
    {"2 replicas, load order rotated by workgroup", 2, 0, 1, 0, 16},
    {"2 replicas, rotated by 5 x workgroup", 2, 0, 5, 0, 16},
    {"2 replicas, nap 8 before the loads", 2, 0, 0, 8, 16},
    {"2 replicas, nap 16 before the loads", 2, 0, 0, 16, 16},
    {"2 replicas, nap 32 before the loads", 2, 0, 0, 32, 16},
    {"2 replicas, sc0 sc1 loads", 2, 0, 0, 0, 17},
    {"2 replicas, sc0 loads", 2, 0, 0, 0, 1},
    {"static data, no publish, sc1 loads (pure intake)", 2, 2, 0, 0, 16},
    {"static data, sc1 loads, rotated", 2, 2, 1, 0, 16},
    {"static data, plain loads (L1 may serve)", 2, 2, 0, 0, 0},
    {"static data, 8 replicas, sc1", 8, 2, 0, 0, 16},
  };
  const size_t lds = 96 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(ingest_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (const V& v : vs) {
    Args a; a.words = words; a.bytes = (unsigned)bytes; a.rounds = 2000; a.replicas = v.replicas; a.mode = v.mode; a.rot = v.rot; a.nap = v.nap;
    a.out = out; a.fail = fail;
    hipMemset(words, 0, bytes); hipMemset(fail, 0, 4);
    hipLaunchKernelGGL(ingest_kernel, dim3(NWG), dim3(NT), lds, 0, a, v.aux);
    hipDeviceSynchronize();
    hipMemset(words, 0, bytes);
    hipEventRecord(e0);
    hipLaunchKernelGGL(ingest_kernel, dim3(NWG), dim3(NT), lds, 0, a, v.aux);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned f = 0; hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
    printf("%-62s %7.3f us per round%s\n", v.name, ms * 1000.0f / a.rounds, f ? "  (GAVE UP)" : "");
    fflush(stdout);
  }
  return 0;
}
