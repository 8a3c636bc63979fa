// Shared device/host helpers for libmeant_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include "../../include/meant_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define WAVE 64

// ---- error plumbing -------------------------------------------------------------------------
void meant_set_error(const char* fmt, ...);

#define MEANT_REQUIRE(cond, code, ...)      \
  do {                                      \
    if (!(cond)) {                          \
      meant_set_error(__VA_ARGS__);         \
      return (code);                        \
    }                                       \
  } while (0)

#define MEANT_LAUNCH_CHECK(name)                                                     \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      meant_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));        \
      return MEANT_ERR_LAUNCH;                                                       \
    }                                                                                \
  } while (0)

// ---- per-device state, runtime options, route counters (core.hip) -------------------------------------------
#define MEANT_MAX_DEVICES 64
// index of the calling thread's current HIP device, or -1
int meant_current_device(void);
// raise hipFuncAttributeMaxDynamicSharedMemorySize of `kernel` once per DEVICE (the attribute is per device)
int meant_raise_dyn_lds(const void* kernel, int bytes);
#define MEANT_RAISE_LDS(kernel, bytes)                                        \
  do {                                                                        \
    int rc__ = meant_raise_dyn_lds((const void*)(kernel), (int)(bytes));      \
    if (rc__) return rc__;                                                    \
  } while (0)

enum meant_option_id {
  MEANT_OPT_NT_STREAM = 0, MEANT_OPT_NT_DYNAMIC, MEANT_OPT_DETERMINISTIC, MEANT_OPT_NT_GRID_CAP, MEANT_OPT_ATTN_SHORT, MEANT_OPT_NT_RAGGED, MEANT_OPT_NT_SPLIT, MEANT_OPT_ATTN_BWD1, MEANT_OPT_NT_PP, MEANT_OPT_TN_PP, MEANT_OPT_COUNT
};
int meant_opt(int id);

enum meant_route_id {
  ROUTE_NT128 = 0, ROUTE_NT256, ROUTE_NT256S, ROUTE_NT256S_ROT, ROUTE_NT_SPLIT, ROUTE_TN128, ROUTE_TN256, ROUTE_TN256_DET, ROUTE_TN_TAIL,
  ROUTE_GEMM_F32, ROUTE_ATTN_FWD, ROUTE_ATTN_FWD_D128, ROUTE_ATTN_FWD_D96, ROUTE_ATTN_BWD,
  ROUTE_ATTN_BWD_D128, ROUTE_ATTN_BWD_D96, ROUTE_ATTN_GENERIC, ROUTE_ATTN_CLS, ROUTE_ATTN_SHORT, ROUTE_NT_OVERLAP, ROUTE_ATTN_BWD1, MEANT_ROUTE_COUNT
};
void meant_route_hit(int route);

static inline bool meant_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- storage-type helpers (math is always float) ------------------------------------------------
template <typename T> struct Vec8;  // 8 contiguous elements
template <> struct Vec8<float> {
  f32x4 lo, hi;
  __device__ __forceinline__ float get(int i) const { return i < 4 ? lo[i] : hi[i - 4]; }
  __device__ __forceinline__ void set(int i, float v) { if (i < 4) lo[i] = v; else hi[i - 4] = v; }
};
template <> struct Vec8<bf16> {
  bf16x8 v;
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
  __device__ __forceinline__ unsigned raw(int i) const {                 // the stored bits of element i
    typedef __attribute__((ext_vector_type(4))) unsigned uvec4;
    return (__builtin_bit_cast(uvec4, v)[i >> 1] >> ((i & 1) * 16)) & 0xffffu;
  }
};

// one rotary pair (2j, 2j+1) with the products contracted the same way wherever it is evaluated (a free choice of fma by the
// compiler made the fused regroup + rotary pass differ from the in-place kernel in the last bit)
__device__ __forceinline__ void rotary_pair(float t0, float t1, float a0, float a1, float b0, float b1, float& o0, float& o1) {
  o0 = __builtin_fmaf(t0, a0, -(t1 * b0));
  o1 = __builtin_fmaf(t1, a1, t0 * b1);
}

template <typename T> __device__ __forceinline__ Vec8<T> load8(const T* p);
template <> __device__ __forceinline__ Vec8<float> load8<float>(const float* p) {
  Vec8<float> r; r.lo = *reinterpret_cast<const f32x4*>(p); r.hi = *reinterpret_cast<const f32x4*>(p + 4); return r;
}
template <> __device__ __forceinline__ Vec8<bf16> load8<bf16>(const bf16* p) {
  Vec8<bf16> r; r.v = *reinterpret_cast<const bf16x8*>(p); return r;
}
// streaming forms for tensors a kernel touches once (nontemporal: no reuse expected in L2)
template <typename T> __device__ __forceinline__ Vec8<T> load8s(const T* p);
template <> __device__ __forceinline__ Vec8<float> load8s<float>(const float* p) {
  Vec8<float> r;
  r.lo = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  r.hi = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 4));
  return r;
}
template <> __device__ __forceinline__ Vec8<bf16> load8s<bf16>(const bf16* p) {
  Vec8<bf16> r; r.v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p)); return r;
}
template <typename T> __device__ __forceinline__ void store8s(T* p, const Vec8<T>& v);
template <> __device__ __forceinline__ void store8s<float>(float* p, const Vec8<float>& v) {
  __builtin_nontemporal_store(v.lo, reinterpret_cast<f32x4*>(p)); __builtin_nontemporal_store(v.hi, reinterpret_cast<f32x4*>(p + 4));
}
template <> __device__ __forceinline__ void store8s<bf16>(bf16* p, const Vec8<bf16>& v) {
  __builtin_nontemporal_store(v.v, reinterpret_cast<bf16x8*>(p));
}
template <typename T> __device__ __forceinline__ void store8(T* p, const Vec8<T>& v);
template <> __device__ __forceinline__ void store8<float>(float* p, const Vec8<float>& v) {
  *reinterpret_cast<f32x4*>(p) = v.lo; *reinterpret_cast<f32x4*>(p + 4) = v.hi;
}
template <> __device__ __forceinline__ void store8<bf16>(bf16* p, const Vec8<bf16>& v) {
  *reinterpret_cast<bf16x8*>(p) = v.v;
}

__device__ __forceinline__ float to_f(float x) { return x; }
__device__ __forceinline__ float to_f(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float x) { return (bf16)x; }

// 64-lane butterfly reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// The same sums on the VALU: four DPP adds leave every lane with the sum of its row of 16, four v_readlane collect the rows.
// __shfl_xor compiles to six DEPENDENT ds_bpermute_b32 + s_waitcnt lgkmcnt(0) pairs (~64 clocks of LDS latency each), which a
// streaming kernel at 2-3 waves per SIMD cannot hide; EXEC must be all ones (wave-uniform control flow around the call).
#define MEANT_DPP_ADD(v, CTRL) ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (CTRL), 0xf, 0xf, true)))
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = MEANT_DPP_ADD(v, 0xB1);                          // quad_perm [1,0,3,2]
  v = MEANT_DPP_ADD(v, 0x4E);                          // quad_perm [2,3,0,1]
  v = MEANT_DPP_ADD(v, 0x141);                         // row_half_mirror
  v = MEANT_DPP_ADD(v, 0x140);                         // row_mirror
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}
// 1 / (||x|| / sqrt(d) + eps) from the row's sum of squares (utils/rms_norm.py:50-57), IEEE square root and division.  The
// hardware v_sqrt_f32 / v_rcp_f32 pair (1 ulp each, ~30 instructions fewer per row) was measured on top of the DPP sums:
// no difference in the pooled kernels (0.481 / 0.799 ms against 0.498 / 0.766 ms forward / backward, HEAD 0.647 / 0.867 on
// both boxes), so the exact form stays
__device__ __forceinline__ float rms_rinv(float sumsq, float inv_sqrt_d, float eps) {
  return 1.0f / (sqrtf(sumsq) * inv_sqrt_d + eps);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
// exact-erf GELU for the bf16 tier's GEMM epilogues, where libm's erff (~40 VALU instructions per element) is paid on
// 65536 outputs per tile with the matrix pipe idle: Phi(x) through the Abramowitz-Stegun 7.1.26 rational form of erfc
// (|error| <= 7.5e-8 in Phi, three orders below bf16 rounding), one v_rcp and one v_exp per element, written so that
// the negative tail has no cancellation.
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float half_erfc = 0.5f * p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);   // 0.5 * erfc(|x|/sqrt2)
  const float phi = x < 0.f ? half_erfc : 1.0f - half_erfc;
  return x * phi;
}
// Phi(x) alone (same rational erfc): gelu(x) = x * Phi(x), gelu'(x) = Phi(x) + x * pdf(x)
__device__ __forceinline__ float gelu_phi_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float half_erfc = 0.5f * p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
  return x < 0.f ? half_erfc : 1.0f - half_erfc;
}
// Phi(x) of a STORED bf16 pre-activation by table (phi_table.inc, tools/gen_phi_table.py): the gelu-on-load norm kernels are
// VALU-bound on the rational form above (a v_rcp, a v_exp and ~12 more operations per element); a bf16 x has 1664 magnitudes
// per sign between 2^-10 and 8 and Phi is constant to bf16 rounding outside them, so Phi(x) is one clamp of the raw bits and
// one LDS read of a 13 KiB table.  phi_slot: index of raw bf16 bits (low 16 bits of `bits`) into the table, x > 0 half first.
constexpr unsigned PHI_LO = 0x3A80u, PHI_HI = 0x40FFu, PHI_N = PHI_HI - PHI_LO + 1;
__device__ __forceinline__ unsigned phi_slot(unsigned bits) {
  unsigned m = bits & 0x7FFFu;
  m = m < PHI_LO ? PHI_LO : m;
  m = m > PHI_HI ? PHI_HI : m;
  return (m - PHI_LO) + ((bits >> 15) & 1u) * PHI_N;
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// d/dx of the above with the same rational erfc; Phi and the density share one exponential
__device__ __forceinline__ float gelu_erf_grad_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);        // exp(-x^2 / 2)
  const float half_erfc = 0.5f * p * t * e;
  const float phi = x < 0.f ? half_erfc : 1.0f - half_erfc;
  return fmaf(x * 0.39894228040143268f, e, phi);
}

// counter-based uniform in [0,1): splitmix-style hash of (seed, index); same value in fwd and bwd
__device__ __forceinline__ float hash_uniform(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}
// Dropout keep-multipliers for 8 consecutive elements starting at element index `base` (a multiple of 8): a 16-bit draw per
// element (keep iff draw >= p * 65536); the multiplier is 1 / (1 - p_q) for the quantised p_q the draws really implement.
// Forward and backward call this with the same (seed, base) and get the same mask.
// When p is a multiple of 1/256 (nn.Dropout()'s default 0.5 is) 8 bits per element decide exactly the same keep probability:
// two 32-bit draws serve the 8 elements (four for the 16-bit form).
#ifndef DROP_HASH32
#define DROP_HASH32 1
#endif
// 32 well-mixed bits of (seed, block, stream j): murmur3's 32-bit finaliser over the block counter xor a seed word -- two
// quarter-rate multiplies against the ~30 of a 64-bit splitmix round (a 64 x 64 multiply is four of them, three per round)
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ uint32_t draw32(uint64_t seed, uint64_t blk, uint32_t j) {
  const uint32_t hi = (uint32_t)(blk >> 32);                              // nonzero only past 2^35 elements
  const uint32_t k = (uint32_t)blk ^ ((hi << 17) | (hi >> 15));
  const uint32_t s = ((uint32_t)seed + 0x9E3779B9u * (j + 1)) ^ ((uint32_t)(seed >> 32) * 0x7FEB352Du);   // wave-uniform
  return mix32(k ^ s);
}
__device__ __forceinline__ void keep_scale8(float p, uint64_t seed, uint64_t base, float (&m)[8]) {
  const unsigned thr = (unsigned)(p * 65536.0f);
  const float sc = 65536.0f / (float)(65536u - thr);
  if ((thr & 0xffu) == 0) {                            // wave-uniform: p comes from the kernel arguments
    const unsigned thr8 = thr >> 8;
#if DROP_HASH32
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t z = draw32(seed, base >> 3, (uint32_t)h);
#pragma unroll
      for (int q = 0; q < 4; ++q) m[4 * h + q] = ((z >> (8 * q)) & 0xffu) >= thr8 ? sc : 0.0f;
    }
#else
    uint64_t z = seed + ((base >> 3) | (1ull << 62)) * 0x9E3779B97F4A7C15ull;     // a counter space of its own
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
#pragma unroll
    for (int q = 0; q < 8; ++q) m[q] = ((unsigned)(z >> (8 * q)) & 0xffu) >= thr8 ? sc : 0.0f;
#endif
    return;
  }
#if DROP_HASH32
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const uint32_t z = draw32(seed, base >> 3, 2u + (uint32_t)h);
#pragma unroll
    for (int q = 0; q < 2; ++q) m[2 * h + q] = ((z >> (16 * q)) & 0xffffu) >= thr ? sc : 0.0f;
  }
#else
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint64_t z = seed + ((base >> 2) + h) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
#pragma unroll
    for (int q = 0; q < 4; ++q) m[4 * h + q] = ((unsigned)(z >> (16 * q)) & 0xffffu) >= thr ? sc : 0.0f;
  }
#endif
}

// ---- LDS helpers -------------------------------------------------------------------------------
// 32-bit LDS byte address of a pointer into __shared__ memory
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)p;
}
// ds_read_b64_tr_b16 issued as inline asm.  Why not the builtin: with global_load_lds DMA in flight hipcc
// (ROCm 7.2) puts an s_waitcnt vmcnt(0) in front of every builtin transposed LDS read, which drains the
// prefetch of the next tile before the current one is consumed.  The asm form is invisible to that pass;
// the caller owns the ordering: lds_wait_all() (s_waitcnt lgkmcnt(0) + a scheduling fence) must sit between
// these reads and the first use of their results, and the DMA that filled the tile must have been retired
// by a vmcnt wait + barrier before (the kernels do that once per tile).
template <int OFF>
__device__ __forceinline__ u32x2 lds_read_tr16(unsigned addr) {
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
__device__ __forceinline__ void lds_wait_all() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// counted form: returns when at most N of this wave's LDS operations are still outstanding (they retire in order)
template <int N>
__device__ __forceinline__ void lds_wait_upto() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ bf16x8 pack_tr(u32x2 lo, u32x2 hi) {
  u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8, v);
}

#define DISPATCH_DTYPE(dtype, T, ...)                                           \
  do {                                                                          \
    if ((dtype) == MEANT_F32) { using T = float; __VA_ARGS__; }                 \
    else if ((dtype) == MEANT_BF16) { using T = bf16; __VA_ARGS__; }            \
    else { meant_set_error("unknown dtype %d", (int)(dtype)); return MEANT_ERR_ARG; } \
  } while (0)
