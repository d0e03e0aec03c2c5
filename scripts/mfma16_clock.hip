// How fast does one wave per SIMD issue v_mfma_f32_16x16x4_f32 with 1, 2 or 4 independent accumulation chains?
//   hipcc -O3 --offload-arch=gfx950 -o mfma16_clock mfma16_clock.hip && ./mfma16_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(256) void k(float* out, int n, unsigned long long* clk) {
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-3f;
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u % CH] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % CH], 0, 0, 0);
  }
  const unsigned long long w1 = wall_clock64(), c1 = clock64();
  float s = 0;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = w1 - w0; clk[1] = c1 - c0; }
}
template <int CH>
void go(float* out, unsigned long long* clk) {
  for (int blocks : {1, 256}) {
    const int n = 256;
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(256), 0, 0, out, n, clk);
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(256), 0, 0, out, n, clk);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ns = h[0] * 10.0, mf = 16.0 * n;
    printf("%d chain(s), blocks %4d: %6.1f ns per MFMA, shader clock %.2f GHz, %.1f cycles per MFMA\n", CH, blocks, ns / mf, h[1] / ns, (double)h[1] / mf);
  }
}
int main() {
  float* out; unsigned long long* clk;
  hipMalloc((void**)&out, 1024 * 256 * 4); hipMalloc((void**)&clk, 16);
  go<1>(out, clk); go<2>(out, clk); go<4>(out, clk);
  return 0;
}
