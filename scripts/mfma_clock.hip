// How fast does one wave issue dependent v_mfma_f32_32x32x2_f32, and at what shader clock?
//   hipcc -O3 --offload-arch=gfx950 -o mfma_clock mfma_clock.hip && ./mfma_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int n, unsigned long long* clk) {
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-3f;
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  const unsigned long long w1 = wall_clock64(), c1 = clock64();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = w1 - w0; clk[1] = c1 - c0; }
}
int main() {
  float* out; unsigned long long* clk;
  hipMalloc((void**)&out, 4096 * 256 * 4); hipMalloc((void**)&clk, 16);
  for (int blocks : {1, 16, 256, 1024}) {
    for (int n : {64, 1024}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, n, clk);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, n, clk);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      const double ns = h[0] * 10.0, mf = 16.0 * n;
      printf("blocks %4d, %5d MFMAs/wave: kernel %8.1f us; in-kernel %8.1f us = %6.1f ns per MFMA, shader clock %.2f GHz, %.1f cycles per MFMA\n",
             blocks, 16 * n, ms * 1e3, ns * 1e-3, ns / mf, h[1] / ns, (double)h[1] / mf);
    }
  }
  return 0;
}
