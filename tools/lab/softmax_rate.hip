// dev microbenchmark: cost of one tile's online-softmax VALU work (32 queries x 64 keys per wave, 32 scores per lane)
// with 1, 2 and 3 waves per SIMD, no memory, no MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
__global__ __launch_bounds__(768) void k(float* out, int iters, float c1) {
  f32x16 s[2];
  for (int b = 0; b < 2; ++b) for (int e = 0; e < 16; ++e) s[b][e] = (float)((threadIdx.x * 7 + e * 3 + b) % 13) * 0.1f;
  float m_run = -1e30f, l_run = 0.f, acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    float tmx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) tmx[e & 3] = fmaxf(tmx[e & 3], s[b][e]);
#pragma unroll
    for (int e = 0; e < 4; ++e) tmx[e] *= c1;
    float tmax = fmaxf(fmaxf(tmx[0], tmx[1]), fmaxf(tmx[2], tmx[3]));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    if (!__all(tmax <= m_run)) { const float mn = fmaxf(m_run, tmax); l_run *= __builtin_amdgcn_exp2f(m_run - mn); m_run = mn; }
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
    f32x16 p[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) { p[b][e] = __builtin_amdgcn_exp2f(fmaf(s[b][e], c1, -m_run)); ps[e & 3] += p[b][e]; }
    float psum = (ps[0] + ps[1]) + (ps[2] + ps[3]);
    psum += __shfl_xor(psum, 32, 64);
    l_run += psum;
    bf16x8 f[4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int j = 0; j < 8; ++j) { f[2 * b][j] = (__bf16)p[b][j]; f[2 * b + 1][j] = (__bf16)p[b][8 + j]; }
    // feed the result back so nothing is dead and scores change
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) s[b][e] = s[b][e] * 0.999f + (float)f[(2 * b + (e >> 3))][e & 7] * 1e-3f;
  }
  for (int b = 0; b < 2; ++b) for (int e = 0; e < 16; ++e) acc += s[b][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + l_run + m_run;
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 768 * 4);
  for (int threads : {256, 512, 768}) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int it = 20000;
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, 1000, 0.05f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, it, 0.05f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%d waves/SIMD: %.3f ms  -> %.1f ns per wave-tile per SIMD (x clock = cycles)\n", threads / 256, ms, ms * 1e6 / (it * (threads / 256)));
  }
  return 0;
}
