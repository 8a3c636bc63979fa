// dev experiment: a THIN streaming kernel -- persistent, one 4-wave workgroup per CU, no LDS, <= 64 VGPRs -- as a stand-in for an
// HBM-bound pass (norm, cast) written so that it can be co-resident with the streaming GEMM's workgroup (which leaves 64 VGPRs per
// SIMD lane and no LDS on its CU).  thin_copy_launch copies n16 16-byte chunks with 8 loads in flight per lane.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
extern "C" __global__ __launch_bounds__(256) void thin_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long long n16) {
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 7 * stride < n16; i += 8 * stride) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) __builtin_nontemporal_store(v[u], dst + i + u * stride);
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}
extern "C" int thin_copy_launch(const void* src, void* dst, long long n16, int nblocks, void* stream) {
  hipLaunchKernelGGL(thin_copy_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, (u32x4*)dst, n16);
  return (int)hipGetLastError();
}
