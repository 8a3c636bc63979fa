// lab: does the immediate offset of global_load_lds_dwordx4 (SGPR-base form) move the LDS destination as well as the global source?
// one wave: m0 = 4096; global_load_lds_dwordx4 v(lane*16), s[g] offset:1024.  Reports where in LDS the bytes landed and which global bytes they are.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned* g, unsigned* out) {
  __shared__ unsigned lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  unsigned base = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)lds + 4096;
  unsigned voff = threadIdx.x * 16;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\ts_waitcnt vmcnt(0)" ::"v"(voff), "s"(g), "s"(base) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 8192; i += 64) out[i] = lds[i];
}
int main() {
  unsigned *g, *o, h[8192], src[4096];
  for (int i = 0; i < 4096; ++i) src[i] = i;       // word i holds i: byte offset = 4 i
  (void)hipMalloc(&g, sizeof(src)); (void)hipMalloc(&o, sizeof(h));
  (void)hipMemcpy(g, src, sizeof(src), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o);
  (void)hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  int first = -1, last = -1;
  for (int i = 0; i < 8192; ++i) if (h[i] != 0xdeadbeefu) { if (first < 0) first = i; last = i; }
  printf("m0 = LDS byte 4096, inst offset 1024: data landed in LDS bytes [%d, %d], first word holds global word %u (byte %u)\n", first * 4, last * 4 + 3,
         first >= 0 ? h[first] : 0, first >= 0 ? h[first] * 4 : 0);
  return 0;
}
