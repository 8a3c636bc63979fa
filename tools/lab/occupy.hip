// dev experiment: a long-running kernel that pins NB workgroups (160 KiB of LDS each, so nothing of ours co-resides on their
// CUs) for `us` microseconds -- a stand-in for a collective's kernel running next to the step.  Built as a tiny .so.
#include <hip/hip_runtime.h>
extern "C" __global__ void occupy_kernel(long long cycles, int* sink) {
  extern __shared__ char smem[];
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) { __builtin_amdgcn_s_sleep(64); }
  if (threadIdx.x == 0 && smem[0] == 123) sink[0] = 1;
}
extern "C" int occupy_launch(int nb, double us, void* stream) {
  static int* sink = nullptr;
  if (!sink) (void)hipMalloc(&sink, 4);
  (void)hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipLaunchKernelGGL(occupy_kernel, dim3(nb), dim3(256), 163840, (hipStream_t)stream, (long long)(us * 100.0), sink);   // wall_clock64: 100 MHz
  return (int)hipGetLastError();
}
