// dev microbenchmark: sustained MFMA issue rate of ONE wave per SIMD vs TWO, for 16x16x32 and 32x32x16 bf16,
// accumulators in AGPRs (asm) or wherever the compiler puts them (builtin).  No memory traffic at all.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC, bool ASM>
__global__ __launch_bounds__(512) void k16(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(threadIdx.x * 3 + e); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (ASM) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(512) void k32(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(threadIdx.x * 3 + e); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> static void run(const char* name, F launch, int threads, double flop_per_mfma, int nacc, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(iters / 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch(iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = 256.0 * threads / 64;
  const double mfmas = waves * (double)iters * nacc;
  printf("%-34s %8.3f ms  %8.1f TFLOP/s   %6.2f ns per MFMA per SIMD\n", name, ms, mfmas * flop_per_mfma / ms / 1e9,
         ms * 1e6 / ((double)iters * nacc * (threads / 256.0)));
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int it = 20000;
  run("16x16x32 builtin 1 wave/SIMD x16acc", [&](int n) { hipLaunchKernelGGL((k16<16, false>), dim3(256), dim3(256), 0, 0, out, n); }, 256, 16384, 16, it);
  run("16x16x32 builtin 2 wave/SIMD x16acc", [&](int n) { hipLaunchKernelGGL((k16<16, false>), dim3(256), dim3(512), 0, 0, out, n); }, 512, 16384, 16, it);
  run("16x16x32 asm-agpr 1 wave/SIMD x16acc", [&](int n) { hipLaunchKernelGGL((k16<16, true>), dim3(256), dim3(256), 0, 0, out, n); }, 256, 16384, 16, it);
  run("16x16x32 asm-agpr 2 wave/SIMD x16acc", [&](int n) { hipLaunchKernelGGL((k16<16, true>), dim3(256), dim3(512), 0, 0, out, n); }, 512, 16384, 16, it);
  run("16x16x32 asm-agpr 1 wave/SIMD x4acc", [&](int n) { hipLaunchKernelGGL((k16<4, true>), dim3(256), dim3(256), 0, 0, out, n); }, 256, 16384, 4, it * 4);
  run("32x32x16 builtin 1 wave/SIMD x4acc", [&](int n) { hipLaunchKernelGGL((k32<4>), dim3(256), dim3(256), 0, 0, out, n); }, 256, 32768, 4, it * 2);
  run("32x32x16 builtin 2 wave/SIMD x4acc", [&](int n) { hipLaunchKernelGGL((k32<4>), dim3(256), dim3(512), 0, 0, out, n); }, 512, 32768, 4, it * 2);
  return 0;
}
