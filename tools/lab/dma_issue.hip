// dev microbenchmark: what does one global_load_lds (16 B per lane -> LDS) cost a wave that is otherwise issuing
// back-to-back MFMAs?  256 workgroups x {256, 512} threads (1 or 2 waves per SIMD), loop of NM MFMAs + ND DMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int NM, int ND>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, float* out, int iters, int stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(threadIdx.x * 3 + e); }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* p = src + (size_t)blockIdx.x * 1048576 + (threadIdx.x & 63) * 16 + wave * 1024;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16 / NM; ++g) {
#pragma unroll
      for (int i = 0; i < NM; ++i)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[g * NM + i]) : "v"(a), "v"(b));
#pragma unroll
      for (int d = 0; d < ND; ++d)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p + ((it * 16 + g * ND + d) & 63) * stride), (lds_ptr_t)(smem + wave * 16384 + ((g * ND + d) & 15) * 1024), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)");
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + smem[threadIdx.x];
}
template <int NM, int ND> static void run(const char* name, int threads, const char* src, float* out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int it = 20000;
  (void)hipFuncSetAttribute((const void*)k<NM, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipLaunchKernelGGL((k<NM, ND>), dim3(256), dim3(threads), 131072, 0, src, out, 100, 4096);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NM, ND>), dim3(256), dim3(threads), 131072, 0, src, out, it, 4096);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s %d waves/SIMD: %7.1f ns per 16 MFMAs per wave (MFMA alone ~130)\n", name, threads / 256, ms * 1e6 / it);
}
int main() {
  char* src; float* out;
  (void)hipMalloc(&src, 256u * 1048576u + 1048576u); (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMemset(src, 0, 256u * 1048576u);
  for (int threads : {256, 512}) {
    run<16, 0>("16 MFMA, no DMA", threads, src, out);
    run<16, 1>("16 MFMA + 1 DMA", threads, src, out);
    run<8, 1>("(8 MFMA + 1 DMA) x2", threads, src, out);
    run<4, 1>("(4 MFMA + 1 DMA) x4", threads, src, out);
    run<16, 4>("16 MFMA + 4 DMA bunched", threads, src, out);
  }
  return 0;
}
