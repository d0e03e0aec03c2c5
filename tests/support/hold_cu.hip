// Test support (not part of the product): one small workgroup that keeps 64 KiB of a compute unit's LDS for a bounded
// time, so that a kernel which needs a whole CU per workgroup cannot place one there meanwhile.
#include <hip/hip_runtime.h>

namespace {
__global__ void hold_cu_kernel(unsigned long long ticks, unsigned* sink) {
  extern __shared__ unsigned held[];
  held[threadIdx.x] = threadIdx.x;
  const unsigned long long t0 = wall_clock64();          // the device's constant 100 MHz clock
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
  if (sink) sink[threadIdx.x] = held[threadIdx.x];       // (never taken: keeps the array alive)
}
}  // namespace

// Enqueues the holder on `stream` for `seconds` (at most 10); returns 0 or a HIP error code.
extern "C" int hold_cu(void* stream, double seconds) {
  if (!(seconds > 0.0) || seconds > 10.0) return -22;
  const unsigned long long ticks = (unsigned long long)(seconds * 100e6);
  hipLaunchKernelGGL(hold_cu_kernel, dim3(1), dim3(64), 64 * 1024, (hipStream_t)stream, ticks, (unsigned*)nullptr);
  return (int)hipGetLastError();
}
