// dev harness: ablations of the streaming NT kernel (DBG bit 0: no DMA in the loop, bit 2: no epilogue), fixed tile walk
#include "../../meant_amd/csrc/gemm_bf16.hip"
#include <cstdio>
#ifndef KERNEL
#define KERNEL gemm_bf16_nt256s_kernel
#define THREADS 512
#define LDSB (RING * T2_BYTES)
#define PERSISTENT 1
#endif
// pseudo-random operands: with all-zero data the chip is not power-limited and everything looks 17 % faster
__global__ void fill_kernel(bf16* p, int64_t n, unsigned seed) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f));
  }
}
template <int DBG> static void run(const char* name, GemmBf16Args a, int reps) {
  const int ntm = (int)(a.M / 256), ntn = (int)(a.N / 256);
  (void)hipFuncSetAttribute((const void*)KERNEL<DBG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((KERNEL<DBG, false>), dim3(PERSISTENT ? 256 : ntm * ntn), dim3(THREADS), LDSB, 0, a, ntm, ntn, (TileSched*)nullptr, 0);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((KERNEL<DBG, false>), dim3(PERSISTENT ? 256 : ntm * ntn), dim3(THREADS), LDSB, 0, a, ntm, ntn, (TileSched*)nullptr, 0);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double ksteps = (double)ntm * ntn / 256.0 * (a.K / 64);
  printf("%-28s M=%lld N=%lld K=%lld: %7.3f ms %7.1f TF  %6.3f us per K-step per CU\n", name, (long long)a.M, (long long)a.N, (long long)a.K, ms,
         2.0 * a.M * a.N * a.K / ms / 1e9, ms * 1e3 / ksteps);
}
int main() {
  const int64_t shapes[3][3] = {{786432, 768, 768}, {786432, 2304, 768}, {786432, 768, 3072}};
  for (auto& sh : shapes) {
    const int64_t M = sh[0], N = sh[1], K = sh[2];
    bf16 *A, *B, *C;
    (void)hipMalloc(&A, M * K * 2); (void)hipMalloc(&B, N * K * 2); (void)hipMalloc(&C, M * N * 2);
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, A, M * K, 1u); hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, B, N * K, 2u);
    GemmBf16Args a{};
    a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.ldc = N;
    run<0>("full", a, 10);
    run<64>("nontemporal C stores", a, 10);
    run<0>("full", a, 10);
    run<64>("nontemporal C stores", a, 10);
    run<32>("direct epilogue (no LDS)", a, 10);
    run<1>("no DMA in loop", a, 10);
    run<16>("epilogue without global stores", a, 10);
    run<8>("waves 4-7 never wait for stores", a, 10);
    (void)hipFree(A); (void)hipFree(B); (void)hipFree(C);
  }
  return 0;
}
