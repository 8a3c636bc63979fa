// dev microbenchmark: the per-tile compute of the attention forward (S^T = K Q^T, online softmax, O^T += V^T P^T) on
// a K/V tile that already sits in LDS -- no DMA, no barriers, no masks -- to see what the instruction stream alone
// costs per 32-query x 64-key wave-tile at 1, 2 and 3 waves per SIMD, and what reordering buys.
//   MODE 0: the kernel's order  (S for both 32-key halves, softmax over 64 keys, PV for both halves)
//   MODE 2: kernel order + one __syncthreads per tile;  MODE 3: + the next tile's K/V staged by DMA (double buffer) as in the kernel
//   MODE 1: half-tile pipeline  (S0, S1 issued; softmax(half 0) while S1 runs; PV0 issued; softmax(half 1) while PV0 runs; PV1)
#include "../../meant_amd/csrc/common.h"
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __forceinline__ int swz(int r) { return ((r & 2) << 1) | ((r >> 2) & 3); }
__device__ __forceinline__ bf16x8 frag_row(const char* tile, int row0, int ks, int lane) {
  const int r = row0 + (lane & 31), c = 2 * ks + (lane >> 5);
  return *reinterpret_cast<const bf16x8*>(tile + r * 128 + ((c ^ swz(r)) << 4));
}
struct TrOff { unsigned o[2][2]; };
__device__ __forceinline__ TrOff make_troff(int lane) {
  TrOff t; const int h = lane >> 5, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
  for (int u = 0; u < 2; ++u) for (int b = 0; b < 2; ++b) {
    const int r = 8 * u + 4 * h + q, col = 32 * b + ((lane >> 4) & 1) * 16 + 4 * p;
    t.o[u][b] = (unsigned)(r * 128 + ((((col >> 3)) ^ swz(r)) << 4) + (col & 7) * 2);
  }
  return t;
}
template <int KROW0> __device__ __forceinline__ void tr_issue(unsigned a, const TrOff& t, int b, u32x2& lo, u32x2& hi) {
  lo = lds_read_tr16<KROW0 * 128>(a + t.o[0][b]); hi = lds_read_tr16<KROW0 * 128>(a + t.o[1][b]);
}
__device__ __forceinline__ void acc_to_frags(const f32x16& a, bf16x8& f0, bf16x8& f1) {
  for (int j = 0; j < 8; ++j) { f0[j] = (bf16)a[j]; f1[j] = (bf16)a[8 + j]; }
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const bf16* __restrict__ q, float* out, int iters, float c1, const char* __restrict__ kv) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384 / 2; i += blockDim.x) reinterpret_cast<bf16*>(smem)[i] = (bf16)(float)((i * 7 % 13) - 6) * (bf16)0.1f;
  __syncthreads();
  const char* Kt = smem; const char* Vt = smem + 8192;
  const int wave = threadIdx.x >> 6;
  const char* src = kv + (size_t)blockIdx.x * 65536 + wave * 4096 + lane * 16;
  bf16x8 qf[4];
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(q + (threadIdx.x & 63) * 64 + ks * 16 + 8 * (lane >> 5));
  f32x16 oacc[2];
  for (int b = 0; b < 2; ++b) for (int e = 0; e < 16; ++e) oacc[b][e] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const TrOff troff = make_troff(lane);
  const unsigned vaddr = lds_addr(Vt);
  auto softmax_half = [&](f32x16& s) {
    float tm[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int e = 0; e < 16; ++e) tm[e & 3] = fmaxf(tm[e & 3], s[e]);
    float tmax = fmaxf(fmaxf(tm[0], tm[1]), fmaxf(tm[2], tm[3])) * c1;
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    if (!__all(tmax <= m_run)) {
      const float mn = fmaxf(m_run, tmax), al = __builtin_amdgcn_exp2f(m_run - mn);
      l_run *= al; m_run = mn;
      for (int b = 0; b < 2; ++b) for (int e = 0; e < 16; ++e) oacc[b][e] *= al;
    }
    float ps[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 16; ++e) { const float p = __builtin_amdgcn_exp2f(fmaf(s[e], c1, -m_run)); s[e] = p; ps[e & 3] += p; }
    float psum = (ps[0] + ps[1]) + (ps[2] + ps[3]);
    psum += __shfl_xor(psum, 32, 64);
    l_run += psum;
  };
  auto pv_half = [&](auto SB, f32x16& s) {
    constexpr int sb = decltype(SB)::value;
    u32x2 lo[2][2], hi[2][2];
    tr_issue<32 * sb>(vaddr, troff, 0, lo[0][0], hi[0][0]); tr_issue<32 * sb>(vaddr, troff, 1, lo[0][1], hi[0][1]);
    tr_issue<32 * sb + 16>(vaddr, troff, 0, lo[1][0], hi[1][0]); tr_issue<32 * sb + 16>(vaddr, troff, 1, lo[1][1], hi[1][1]);
    bf16x8 pf[2]; acc_to_frags(s, pf[0], pf[1]);
    lds_wait_all();
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int b = 0; b < 2; ++b) oacc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack_tr(lo[s2][b], hi[s2][b]), pf[s2], oacc[b], 0, 0, 0);
  };
  for (int it = 0; it < iters; ++it) {
    if (MODE == 3) {                                          // stage 16 KiB (K + V tile) into the other buffer: 4 pieces per wave
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + ((it & 3) * 16384) + i * 1024), (lds_ptr_t)(smem + 16384 + wave * 4096 + i * 1024), 16, 0, 0);
    }
    f32x16 sacc[2];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      for (int e = 0; e < 16; ++e) sacc[sb][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) sacc[sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(Kt, 32 * sb, ks, lane), qf[ks], sacc[sb], 0, 0, 0);
    }
    if (MODE == 0 || MODE >= 2) {
      // one softmax over all 64 keys (as in the kernel)
      float tm[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int e = 0; e < 16; ++e) tm[e & 3] = fmaxf(tm[e & 3], sacc[sb][e]);
      float tmax = fmaxf(fmaxf(tm[0], tm[1]), fmaxf(tm[2], tm[3])) * c1;
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      if (!__all(tmax <= m_run)) {
        const float mn = fmaxf(m_run, tmax), al = __builtin_amdgcn_exp2f(m_run - mn);
        l_run *= al; m_run = mn;
        for (int b = 0; b < 2; ++b) for (int e = 0; e < 16; ++e) oacc[b][e] *= al;
      }
      float ps[4] = {0, 0, 0, 0};
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float p = __builtin_amdgcn_exp2f(fmaf(sacc[sb][e], c1, -m_run)); sacc[sb][e] = p; ps[e & 3] += p; }
      float psum = (ps[0] + ps[1]) + (ps[2] + ps[3]);
      psum += __shfl_xor(psum, 32, 64);
      l_run += psum;
      pv_half(std::integral_constant<int, 0>{}, sacc[0]);
      pv_half(std::integral_constant<int, 1>{}, sacc[1]);
    } else {
      softmax_half(sacc[0]);                                  // S of half 1 is still in the matrix pipe
      pv_half(std::integral_constant<int, 0>{}, sacc[0]);
      softmax_half(sacc[1]);                                  // PV of half 0 runs underneath
      pv_half(std::integral_constant<int, 1>{}, sacc[1]);
    }
    if (MODE >= 2) __syncthreads();
  }
  float s = l_run + m_run;
  for (int b = 0; b < 2; ++b) for (int e = 0; e < 16; ++e) s += oacc[b][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> static void run(const char* name, int blocks_per_cu, const bf16* q, float* out, const char* kv) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int it = 4000;
  (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 51200);
  hipLaunchKernelGGL((k<MODE>), dim3(256 * blocks_per_cu), dim3(256), 51200, 0, q, out, 100, 0.05f, kv);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(256 * blocks_per_cu), dim3(256), 51200, 0, q, out, it, 0.05f, kv);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %d waves/SIMD: %7.1f ns per wave-tile per SIMD\n", name, blocks_per_cu, ms * 1e6 / (it * blocks_per_cu));
}
int main() {
  bf16* q; float* out;
  (void)hipMalloc(&q, 64 * 64 * 2 + 4096); (void)hipMalloc(&out, 256 * 3 * 256 * 4);
  (void)hipMemset(q, 0, 64 * 64 * 2);
  char* kv; (void)hipMalloc(&kv, 768 * 65536 + 65536); (void)hipMemset(kv, 0, 768 * 65536);
  for (int w : {2, 3}) { run<0>("kernel order", w, q, out, kv); run<2>("+ barrier per tile", w, q, out, kv); run<3>("+ DMA staging + barrier", w, q, out, kv); }
  return 0;
}
