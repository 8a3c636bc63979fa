// K1: RMSNorm / LayerNorm forward + backward.  HBM-bound streaming kernels.
//   one 64-lane wavefront per row, 16-byte (8-element bf16 / 2x16-byte f32) loads, the row kept in
//   registers between the reduction and the normalisation (one read + one write per element),
//   butterfly reduction over the wave, gain-gradient partials reduced per block then summed by a strip
//   kernel.  The backward optionally folds in the two elementwise passes that surround it in the encoder:
//   the add of the gradient that arrives through the residual branch sharing x, and the GELU derivative
//   when x = gelu(pre) (meant/meant.py:64,107).
// Reference semantics: utils/rms_norm.py:40-57 (eps added to the RMS, outside the sqrt).
#include "internal.h"

namespace {

constexpr int MAXC = 4;               // chunks of 8 per lane kept in registers -> d <= 2048
constexpr int NORM_THREADS = 256;     // 4 waves = 4 rows in flight per block
constexpr int NORM_MAX_BLOCKS = 1024;
constexpr int NORM_PACKED_BLOCKS = 2048;

// exact libm erf in the fp32 tier, the rational erfc (|error| 7.5e-8 in Phi) in the bf16 tier
template <typename T> __device__ __forceinline__ float gelu_grad_t(float x);
template <> __device__ __forceinline__ float gelu_grad_t<float>(float x) { return gelu_erf_grad(x); }
template <> __device__ __forceinline__ float gelu_grad_t<bf16>(float x) { return gelu_erf_grad_fast(x); }

// Phi(x) for every bf16 x with 2^-10 <= |x| < 8 (common.h: phi_slot); each workgroup of a gelu-on-load kernel copies it to LDS
__device__ const float g_phi_tab[2 * PHI_N] = {
#include "phi_table.inc"
};
template <typename T> constexpr bool PHI_TABLE = sizeof(T) == 2;          // bf16 tier only: an fp32 pre-activation has no small domain
// Forward only.  In the backward (two waves per SIMD, nothing to cover the LDS latency of a 64-lane gather, ~10 clocks of bank
// conflicts each) the table measured 10 % SLOWER than the rational form, and a mix of the two (some elements of every 8 by
// table, the rest by arithmetic, to use both pipes) was slower than the table alone in the forward (tools/probe_norm_pooled3.py).
__device__ __forceinline__ void phi_tab_to_lds(float* dst) {
  for (int j = threadIdx.x; j < (int)(2 * PHI_N); j += NORM_THREADS) dst[j] = g_phi_tab[j];
  __syncthreads();
}


// d_part / offset: the partial and the bias form of utils/rms_norm.py:44-57 (RMSNorm(d, p, bias=True)): the statistics are taken over
// the first d_part = int(d p) elements of a row only (d_part == d: the full-width form the MEANT path uses), and `offset` (float [d],
// may be null) is added to the scaled result.
template <typename T>
__global__ __launch_bounds__(NORM_THREADS) void rmsnorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                                    T* __restrict__ y, float* __restrict__ rinv_out,
                                                                    int64_t rows, int d, float eps, float drop_p,
                                                                    uint64_t seed, int d_part = 0, const float* __restrict__ offset = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = d >> 3;
  if (d_part <= 0) d_part = d;
  const float inv_sqrt_d = rsqrtf((float)d_part);
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const T* xr = x + row * d;
    Vec8<T> v[MAXC];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        v[c] = load8<T>(xr + ch * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float f = v[c].get(i); ss += ch * 8 + i < d_part ? f * f : 0.f; }
      }
    }
    ss = wave_sum(ss);
    const float r = 1.0f / (sqrtf(ss) * inv_sqrt_d + eps);
    if (lane == 0) rinv_out[row] = r;
    T* yr = y + row * d;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(scale + ch * 8);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(scale + ch * 8 + 4);
        Vec8<T> o;
        float km[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
        if (drop_p > 0.f) keep_scale8(drop_p, seed, (uint64_t)row * d + ch * 8, km);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          o.set(i, (i < 4 ? g0[i] : g1[i - 4]) * (v[c].get(i) * r) * km[i] + (offset ? offset[ch * 8 + i] : 0.f));
        store8<T>(yr + ch * 8, o);
      }
    }
  }
}

// dx_j = r * g_j dy_j  -  x_j * c * r^2 / (n sqrt(d)),  c = sum_i g_i dy_i x_i,  n sqrt(d) = (1/r - eps) d
// partial form (d_part < d): the statistics see the first d_part elements only, so the second term exists for j < d_part alone and
// n sqrt(d_part) = (1/r - eps) d_part; partial_off (may be null): per-block column sums of dy, the gradient of the offset
template <typename T>
__global__ __launch_bounds__(NORM_THREADS) void rmsnorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                    const float* __restrict__ scale,
                                                                    const float* __restrict__ rinv, T* __restrict__ dx,
                                                                    float* __restrict__ partial, int64_t rows, int d,
                                                                    float eps, float drop_p, uint64_t seed,
                                                                    const T* __restrict__ dres, const T* __restrict__ gelu_pre,
                                                                    int d_part = 0, float* __restrict__ partial_off = nullptr) {
  __shared__ float red[4][512];   // reused per chunk: [wave][64 lanes * 8]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = d >> 3;
  if (d_part <= 0) d_part = d;
  float gacc[MAXC][8], oacc[MAXC][8];
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) { gacc[c][i] = 0.f; oacc[c][i] = 0.f; }

  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const T* xr = x + row * d;
    const T* dyr = dy + row * d;
    const float r = rinv[row];
    Vec8<T> xv[MAXC];
    float gd[MAXC][8];
    float cdot = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        xv[c] = load8<T>(xr + ch * 8);
        const Vec8<T> dv = load8<T>(dyr + ch * 8);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(scale + ch * 8);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(scale + ch * 8 + 4);
        float km[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
        if (drop_p > 0.f) keep_scale8(drop_p, seed, (uint64_t)row * d + ch * 8, km);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float dyi = dv.get(i) * km[i];
          const float xi = xv[c].get(i);
          if (partial_off) oacc[c][i] += dyi;
          gacc[c][i] += dyi * xi * r;
          const float t = (i < 4 ? g0[i] : g1[i - 4]) * dyi;
          gd[c][i] = t;
          cdot += t * xi;
        }
      }
    }
    cdot = wave_sum(cdot);
    const float nsd = (1.0f / r - eps) * (float)d_part;     // ||x_part|| * sqrt(d_part)
    const float k = nsd > 0.f ? cdot * r * r / nsd : 0.f;
    T* dxr = dx + row * d;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        Vec8<T> o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o.set(i, r * gd[c][i] - (ch * 8 + i < d_part ? k * xv[c].get(i) : 0.f));
        if (dres) {                                  // gradient arriving through the residual branch that shares x
          const Vec8<T> rv = load8<T>(dres + row * d + ch * 8);
#pragma unroll
          for (int i = 0; i < 8; ++i) o.set(i, o.get(i) + rv.get(i));
        }
        if (gelu_pre) {                              // x = gelu(pre): chain through the activation in the same pass
          const Vec8<T> pv = load8<T>(gelu_pre + row * d + ch * 8);
#pragma unroll
          for (int i = 0; i < 8; ++i) o.set(i, o.get(i) * gelu_grad_t<T>(pv.get(i)));
        }
        store8<T>(dxr + ch * 8, o);
      }
    }
  }
  // block reduction of the gain-gradient partials: 4 waves -> 1 row of `partial`
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) red[wave][lane * 8 + i] = gacc[c][i];
    __syncthreads();
    if (wave == 0) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float s = red[0][lane * 8 + i] + red[1][lane * 8 + i] + red[2][lane * 8 + i] + red[3][lane * 8 + i];
          partial[(int64_t)blockIdx.x * d + ch * 8 + i] = s;
        }
      }
    }
  }
  if (partial_off) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) red[wave][lane * 8 + i] = oacc[c][i];
      __syncthreads();
      if (wave == 0) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            partial_off[(int64_t)blockIdx.x * d + ch * 8 + i] = red[0][lane * 8 + i] + red[1][lane * 8 + i] + red[2][lane * 8 + i] + red[3][lane * 8 + i];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Packed-row variants for the widths the encoder uses.  One wave takes R consecutive rows as ONE flat run of
// R * d / 8 chunks dealt lane-cyclically, C = R * d / 512 chunks per lane, every lane equally loaded
// (d = 768: R = 2, C = 3 -- the generic kernel gives half the lanes two chunks and the other half one), all loads of
// a row group issued before the first use, and twice the waves per CU in flight.
template <typename T, int C>
__global__ __launch_bounds__(NORM_THREADS) void rmsnorm_fwd_packed_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                                           T* __restrict__ y, float* __restrict__ rinv_out,
                                                                           int64_t rows, int d, int R, float eps, float drop_p,
                                                                           uint64_t seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = d >> 3;
  const float inv_sqrt_d = rsqrtf((float)d);
  int rsel[C], col[C];                                 // which of the R rows / which column a lane's chunk c belongs to
  f32x4 g0[C], g1[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int k = lane + 64 * c;
    rsel[c] = k / nchunk;
    col[c] = (k - rsel[c] * nchunk) * 8;
    g0[c] = *reinterpret_cast<const f32x4*>(scale + col[c]);
    g1[c] = *reinterpret_cast<const f32x4*>(scale + col[c] + 4);
  }
  const int64_t ngroups = rows / R;
  for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < ngroups; grp += (int64_t)gridDim.x * 4) {
    const int64_t row0 = grp * R;
    const T* xr = x + row0 * d;
    Vec8<T> v[C];
    float ss[C];
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = load8s<T>(xr + (lane + 64 * c) * 8);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      ss[c] = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float f = v[c].get(i); ss[c] += f * f; }
    }
    float rr[C];
#pragma unroll
    for (int c = 0; c < C; ++c) rr[c] = 0.f;
    for (int r = 0; r < R; ++r) {                      // R <= 4 segmented sums
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) s += rsel[c] == r ? ss[c] : 0.f;
      s = wave_sum_dpp(s);
      const float rv = rms_rinv(s, inv_sqrt_d, eps);
      if (lane == 0) rinv_out[row0 + r] = rv;
#pragma unroll
      for (int c = 0; c < C; ++c) rr[c] = rsel[c] == r ? rv : rr[c];
    }
    T* yr = y + row0 * d;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float km[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
      if (drop_p > 0.f) keep_scale8(drop_p, seed, (uint64_t)(row0 + rsel[c]) * d + col[c], km);
      Vec8<T> o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o.set(i, (i < 4 ? g0[c][i] : g1[c][i - 4]) * (v[c].get(i) * rr[c]) * km[i]);
      store8s<T>(yr + (lane + 64 * c) * 8, o);
    }
  }
}

// statistics only: rinv[row] = 1 / (||x_row|| / sqrt(d) + eps), nothing else written (the normalisation itself rides the
// consumer GEMM's epilogue as a per-row factor: meant_linear_fwd_rowscale).  One read of x.
template <typename T, int C>
__global__ __launch_bounds__(NORM_THREADS) void rmsnorm_stats_packed_kernel(const T* __restrict__ x, float* __restrict__ rinv_out,
                                                                             int64_t rows, int d, int R, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = d >> 3;
  const float inv_sqrt_d = rsqrtf((float)d);
  int rsel[C];
#pragma unroll
  for (int c = 0; c < C; ++c) rsel[c] = (lane + 64 * c) / nchunk;
  const int64_t ngroups = rows / R;
  for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < ngroups; grp += (int64_t)gridDim.x * 4) {
    const int64_t row0 = grp * R;
    const T* xr = x + row0 * d;
    Vec8<T> v[C];
    float ss[C];
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = load8<T>(xr + (lane + 64 * c) * 8);       // x stays cacheable: the consumer GEMM reads it next
#pragma unroll
    for (int c = 0; c < C; ++c) {
      ss[c] = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float f = v[c].get(i); ss[c] += f * f; }
    }
    for (int r = 0; r < R; ++r) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) s += rsel[c] == r ? ss[c] : 0.f;
      s = wave_sum_dpp(s);
      if (lane == 0) rinv_out[row0 + r] = rms_rinv(s, inv_sqrt_d, eps);
    }
  }
}

// Pooled form: the consumer of this norm (POOL = 1) or of its input (POOL = 2) is the sequence mean-pool (meant/meant.py:231).
// pooled[g, :] (float) = mean over the `group_rows` rows of group g of y (POOL = 1: y itself is not written at all -- it has
// no other reader once the stack's last Linear is evaluated on the pooled features) or of x (POOL = 2: the residual
// operand).  A workgroup owns whole groups: its four waves split a group's rows into four contiguous runs, keep column
// sums in registers, and combine them through LDS in a fixed order -- no atomics, so the forward stays bit-reproducible.
// group_rows is a multiple of R (a wave's step never straddles a group).
// GELU_IN: x holds the PRE-activation h of the Linear + GELU that feeds this norm (meant/meant.py:63-64, :106-107); the
// activation gelu(h), rounded to the storage type as the GEMM epilogue would have stored it, is formed on load, so that
// tensor is never written or read.
template <typename T, int C, int POOL, bool GELU_IN>
__global__ __launch_bounds__(NORM_THREADS) void rmsnorm_fwd_pooled_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                                           T* __restrict__ y, float* __restrict__ rinv_out,
                                                                           int64_t ngroups, int d, int R, float eps, float drop_p,
                                                                           uint64_t seed, float* __restrict__ pooled, int group_rows) {
  __shared__ float red[4][C * 64 * 8];                 // per wave: the column sums of its C * 64 chunks
  __shared__ float phi_s[(GELU_IN && PHI_TABLE<T>) ? 2 * PHI_N : 1];
  if constexpr (GELU_IN && PHI_TABLE<T>) phi_tab_to_lds(phi_s);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nchunk = d >> 3;
  const float inv_sqrt_d = rsqrtf((float)d);
  int rsel[C], col[C];
  f32x4 g0[C], g1[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int k = lane + 64 * c;
    rsel[c] = k / nchunk;
    col[c] = (k - rsel[c] * nchunk) * 8;
    g0[c] = *reinterpret_cast<const f32x4*>(scale + col[c]);
    g1[c] = *reinterpret_cast<const f32x4*>(scale + col[c] + 4);
  }
  const int steps = group_rows / R;                    // steps of R rows per group
  const int per = (steps + 3) / 4;
  const int s_lo = wave * per, s_hi = s_lo + per < steps ? s_lo + per : steps;
  const float inv_group = 1.0f / (float)group_rows;
  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    float psum[C][8];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) psum[c][i] = 0.f;
    for (int st = s_lo; st < s_hi; ++st) {
      const int64_t row0 = g * group_rows + (int64_t)st * R;
      const T* xr = x + row0 * d;
      Vec8<T> v[C];
      float ss[C];
#pragma unroll
      for (int c = 0; c < C; ++c) v[c] = load8s<T>(xr + (lane + 64 * c) * 8);
      if (GELU_IN) {
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if constexpr (PHI_TABLE<T>) v[c].set(i, v[c].get(i) * phi_s[phi_slot(v[c].raw(i))]);
            else v[c].set(i, gelu_erf_fast(v[c].get(i)));
          }
      }
#pragma unroll
      for (int c = 0; c < C; ++c) {
        ss[c] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float f = v[c].get(i); ss[c] += f * f; }
      }
      float rr[C];
#pragma unroll
      for (int c = 0; c < C; ++c) rr[c] = 0.f;
      for (int r = 0; r < R; ++r) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) s += rsel[c] == r ? ss[c] : 0.f;
        s = wave_sum_dpp(s);
        const float rv = rms_rinv(s, inv_sqrt_d, eps);
        if (lane == 0) rinv_out[row0 + r] = rv;
#pragma unroll
        for (int c = 0; c < C; ++c) rr[c] = rsel[c] == r ? rv : rr[c];
      }
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float km[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
        if (drop_p > 0.f) keep_scale8(drop_p, seed, (uint64_t)(row0 + rsel[c]) * d + col[c], km);
        Vec8<T> o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o.set(i, (i < 4 ? g0[c][i] : g1[c][i - 4]) * (v[c].get(i) * rr[c]) * km[i]);
        if (POOL == 1) {
          // the mean is taken over the values the literal path would have stored, i.e. after rounding to the activation type
#pragma unroll
          for (int i = 0; i < 8; ++i) psum[c][i] += o.get(i);
        } else {
          if (y) store8s<T>(y + row0 * d + (lane + 64 * c) * 8, o);     // y == NULL: statistics + means of x only
#pragma unroll
          for (int i = 0; i < 8; ++i) psum[c][i] += v[c].get(i);
        }
      }
    }
    // fixed-order combine: chunk k = lane + 64 c of wave w holds column (k mod nchunk) * 8 of row k / nchunk
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) red[wave][(lane + 64 * c) * 8 + i] = psum[c][i];
    __syncthreads();
    for (int j = threadIdx.x; j < d; j += NORM_THREADS) {
      float s = 0.f;
      for (int w = 0; w < 4; ++w)
        for (int r = 0; r < R; ++r) s += red[w][r * d + j];
      pooled[g * d + j] = s * inv_group;
    }
    __syncthreads();
  }
}

// BC bit 0: dy is the gradient of the POOLED features, float [groups, d]: row r receives dy_g[r / group_rows] / group_rows
// (the backward of POOL = 1 above -- no [tokens, d] broadcast is ever materialised); bit 1: the same for dres (POOL = 2).
// BC bit 2: x is not stored at all -- it is gelu(gelu_pre), formed on load (the forward's GELU_IN)
// CHAIN: the result dpre (gradient of the pre-activation gelu_pre = r_up (x_up W'^T) + b_up of a Linear whose input norm rides
// its GEMM as a per-row factor, meant_linear_fwd_rowscale) is what three more things are made of, all in this pass:
//   dx            <- r_up[row] * dpre : the A operand of BOTH backward GEMMs of that Linear (dW' = (r dpre)^T x_up, and
//                    (r dpre) W' = the first term of d x_up);
//   kcoef[row]    <- rowdot(dpre, pre - b_up) r_up^2 / ((1 - eps_up r_up) d_up): the factor of x_up in the norm's own term,
//                    d x_up = (r dpre) W' - kcoef x_up, which the input-gradient GEMM subtracts in its epilogue;
//   partial2[blk] <- column sums of the UNSCALED dpre (the Linear's bias gradient; the scaled tensor cannot give it).
struct NormChain { const float* up_rinv; const float* up_bias; float* kcoef; float* partial2; float up_eps; int up_d; };
template <typename T, int C, int BC, bool CHAIN = false>
__global__ __launch_bounds__(NORM_THREADS, 2) void rmsnorm_bwd_packed_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                           const float* __restrict__ scale,
                                                                           const float* __restrict__ rinv, T* __restrict__ dx,
                                                                           float* __restrict__ partial, int64_t rows, int d, int R,
                                                                           float eps, float drop_p, uint64_t seed,
                                                                           const T* __restrict__ dres, const T* __restrict__ gelu_pre,
                                                                           int group_rows, NormChain ch = NormChain{}) {
  __shared__ float red[4][C * 64 * 8];                 // per wave: the gain-gradient sums of its C * 64 chunks
  __shared__ float gain_s[CHAIN ? C * 64 * 8 : 1];     // CHAIN: the gains live in LDS (24 registers the extra accumulators need)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float inv_group = BC ? 1.0f / (float)group_rows : 0.f;
  const int nchunk = d >> 3;
  int rsel[C], col[C];
  f32x4 g0[CHAIN ? 1 : C], g1[CHAIN ? 1 : C];
  float gacc[C][8];
  if constexpr (CHAIN) {
    for (int j = threadIdx.x; j < d; j += NORM_THREADS) gain_s[j] = scale[j];
    __syncthreads();
  }
  float bacc[CHAIN ? C : 1][8];                       // CHAIN: column sums of the unscaled result
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int k = lane + 64 * c;
    rsel[c] = k / nchunk;
    col[c] = (k - rsel[c] * nchunk) * 8;
    if constexpr (!CHAIN) {
      g0[c] = *reinterpret_cast<const f32x4*>(scale + col[c]);
      g1[c] = *reinterpret_cast<const f32x4*>(scale + col[c] + 4);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) gacc[c][i] = 0.f;
    if constexpr (CHAIN) {
#pragma unroll
      for (int i = 0; i < 8; ++i) bacc[c][i] = 0.f;
    }
  }
  const int64_t ngroups = rows / R;
  for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < ngroups; grp += (int64_t)gridDim.x * 4) {
    const int64_t row0 = grp * R;
    const int64_t off = row0 * d;
    Vec8<T> xv[C], dv[C], rv[C], pv[C];
    Vec8<float> dvg[C], rvg[C];                        // broadcast (pooled) gradients, fp32
    const int64_t pg = BC ? (int64_t)((unsigned)row0 / (unsigned)group_rows) : 0;   // group of this step's rows (32-bit divide: rows < 2^31 is checked)
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int64_t o = off + (lane + 64 * c) * 8;
      if (gelu_pre) pv[c] = load8s<T>(gelu_pre + o);
      if (BC & 4) {                                   // x = pre * Phi(pre) is not stored: keep Phi (x and gelu' are both one multiply-add away)
        if constexpr (!CHAIN) {
#pragma unroll
          for (int i = 0; i < 8; ++i) xv[c].set(i, gelu_phi_fast(pv[c].get(i)));
        }
      } else xv[c] = load8s<T>(x + o);
      if (BC & 1) dvg[c] = load8<float>(reinterpret_cast<const float*>(dy) + pg * d + col[c]);
      else dv[c] = load8s<T>(dy + o);
      if (BC & 2) rvg[c] = load8<float>(reinterpret_cast<const float*>(dres) + pg * d + col[c]);
      else if (dres) rv[c] = load8s<T>(dres + o);
    }
    // CHAIN: three more accumulator sets than the plain kernel have to fit the 256 registers of two waves per SIMD, so the
    // chunks are worked on ONE AFTER THE OTHER (scheduling fences): the fp32 temporaries of one chunk at a time
    if constexpr (CHAIN && (BC & 4) != 0) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int i = 0; i < 8; ++i) xv[c].set(i, gelu_phi_fast(pv[c].get(i)));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float rr[C], cd[C], gd[C][8];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      rr[c] = rinv[row0 + rsel[c]];
      float km[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
      if (drop_p > 0.f) keep_scale8(drop_p, seed, (uint64_t)(row0 + rsel[c]) * d + col[c], km);
      cd[c] = 0.f;
      f32x4 ga0, ga1;
      if constexpr (CHAIN) {
        ga0 = *reinterpret_cast<const f32x4*>(gain_s + col[c]);
        ga1 = *reinterpret_cast<const f32x4*>(gain_s + col[c] + 4);
      } else { ga0 = g0[c]; ga1 = g1[c]; }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float dyi = ((BC & 1) ? dvg[c].get(i) * inv_group : dv[c].get(i)) * km[i];
        const float xi = (BC & 4) ? pv[c].get(i) * xv[c].get(i) : xv[c].get(i);
        gacc[c][i] += dyi * xi * rr[c];
        const float t = (i < 4 ? ga0[i] : ga1[i - 4]) * dyi;
        gd[c][i] = t;
        cd[c] += t * xi;
      }
      if constexpr (CHAIN) __builtin_amdgcn_sched_barrier(0);
    }
    float kk[C];
#pragma unroll
    for (int c = 0; c < C; ++c) kk[c] = 0.f;
    for (int r = 0; r < R; ++r) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) s += rsel[c] == r ? cd[c] : 0.f;
      s = wave_sum_dpp(s);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        if (rsel[c] == r) {
          const float nsd = (__builtin_amdgcn_rcpf(rr[c]) - eps) * (float)d;          // ||x|| * sqrt(d)
          kk[c] = nsd > 0.f ? s * rr[c] * rr[c] * __builtin_amdgcn_rcpf(nsd) : 0.f;
        }
      }
    }
    float rd[CHAIN ? C : 1], ru[CHAIN ? C : 1];
    if constexpr (CHAIN) {
#pragma unroll
      for (int c = 0; c < C; ++c) ru[c] = ch.up_rinv[row0 + rsel[c]];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      Vec8<T> o;
      f32x4 ub0 = {0.f, 0.f, 0.f, 0.f}, ub1 = ub0;
      if constexpr (CHAIN) {
        rd[c] = 0.f;
        ub0 = *reinterpret_cast<const f32x4*>(ch.up_bias + col[c]);
        ub1 = *reinterpret_cast<const f32x4*>(ch.up_bias + col[c] + 4);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xi2 = (BC & 4) ? pv[c].get(i) * xv[c].get(i) : xv[c].get(i);
        float val = rr[c] * gd[c][i] - kk[c] * xi2;
        if (BC & 2) val += rvg[c].get(i) * inv_group;
        else if (dres) val += rv[c].get(i);
        if (BC & 4) {                                   // gelu'(pre) = Phi + pre * pdf(pre), Phi kept from the first pass
          const float pr = pv[c].get(i);
          val *= fmaf(pr * 0.39894228040143268f, __builtin_amdgcn_exp2f(-0.72134752044448170f * pr * pr), xv[c].get(i));
        } else if (gelu_pre) val *= gelu_grad_t<T>(pv[c].get(i));
        if constexpr (CHAIN) {
          bacc[c][i] += val;
          rd[c] += val * (pv[c].get(i) - (i < 4 ? ub0[i] : ub1[i - 4]));
          o.set(i, val * ru[c]);
        } else o.set(i, val);
      }
      if constexpr (CHAIN) {
        store8<T>(dx + off + (lane + 64 * c) * 8, o);       // read twice right away (dW and dX GEMMs): keep it cacheable
        __builtin_amdgcn_sched_barrier(0);
      } else store8s<T>(dx + off + (lane + 64 * c) * 8, o);
    }
    if constexpr (CHAIN) {
      for (int r = 0; r < R; ++r) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) s += rsel[c] == r ? rd[c] : 0.f;
        s = wave_sum_dpp(s);
        if (lane == 0) {
          const float rv1 = ch.up_rinv[row0 + r];
          const float den = (1.0f - ch.up_eps * rv1) * (float)ch.up_d;      // = ||x_up|| sqrt(d) r
          ch.kcoef[row0 + r] = den > 0.f ? s * rv1 * rv1 / den : 0.f;
        }
      }
    }
  }
  __syncthreads();
  // fixed-order combine (no atomics: the gain gradient is bit-reproducible): chunk k = lane + 64 c of wave w holds column
  // (k mod nchunk) * 8 of row k / nchunk of the wave's R-row steps
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) red[wave][(lane + 64 * c) * 8 + i] = gacc[c][i];
  __syncthreads();
  for (int j = threadIdx.x; j < d; j += NORM_THREADS) {
    float s = 0.f;
    for (int w = 0; w < 4; ++w)
      for (int r = 0; r < R; ++r) s += red[w][r * d + j];
    partial[(int64_t)blockIdx.x * d + j] = s;
  }
  if constexpr (CHAIN) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) red[wave][(lane + 64 * c) * 8 + i] = bacc[c][i];
    __syncthreads();
    for (int j = threadIdx.x; j < d; j += NORM_THREADS) {
      float s = 0.f;
      for (int w = 0; w < 4; ++w)
        for (int r = 0; r < R; ++r) s += red[w][r * d + j];
      ch.partial2[(int64_t)blockIdx.x * d + j] = s;
    }
  }
}

// rows per wave R and chunks per lane C of the packed kernels, or R = 0 if the shape does not pack
inline void norm_packing(int64_t rows, int64_t d, int& R, int& C) {
  const int nchunk = (int)(d >> 3);
  for (R = 1; R <= 4; R *= 2) {
    if ((R * nchunk) % 64 == 0 && R * nchunk / 64 <= 3 && rows % R == 0) { C = R * nchunk / 64; return; }
  }
  R = 0; C = 0;
}

// out[j] = sum_p partial[p, j]   (and optionally a second array at partial + np*d)
__global__ void colreduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int np, int d) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= d) return;
  float s = 0.f;
  for (int p = 0; p < np; ++p) s += partial[(int64_t)p * d + j];
  out[j] = s;
}

template <typename T>
__global__ __launch_bounds__(NORM_THREADS) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                                      const float* __restrict__ beta, T* __restrict__ y,
                                                                      float* __restrict__ stats, int64_t rows, int d, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = d >> 3;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const T* xr = x + row * d;
    Vec8<T> v[MAXC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        v[c] = load8<T>(xr + ch * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[c].get(i);
      }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float t = v[c].get(i) - mean; q += t * t; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
    if (lane == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
    T* yr = y + row * d;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        Vec8<T> o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o.set(i, (v[c].get(i) - mean) * rstd * gamma[ch * 8 + i] + beta[ch * 8 + i]);
        store8<T>(yr + ch * 8, o);
      }
    }
  }
}

// one wave per row; dgamma/dbeta partials per block written to partial[2][gridDim][d]
template <typename T>
__global__ __launch_bounds__(NORM_THREADS) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                      const float* __restrict__ gamma,
                                                                      const float* __restrict__ stats, T* __restrict__ dx,
                                                                      float* __restrict__ partial, int64_t rows, int d) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = d >> 3;
  float ga[MAXC][8], ba[MAXC][8];
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) { ga[c][i] = 0.f; ba[c][i] = 0.f; }
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    float xh[MAXC][8], gd[MAXC][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        const Vec8<T> xv = load8<T>(x + row * d + ch * 8);
        const Vec8<T> dv = load8<T>(dy + row * d + ch * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float h = (xv.get(i) - mean) * rstd;
          const float dyi = dv.get(i);
          xh[c][i] = h;
          ga[c][i] += dyi * h;
          ba[c][i] += dyi;
          const float t = dyi * gamma[ch * 8 + i];
          gd[c][i] = t;
          s1 += t;
          s2 += t * h;
        }
      }
    }
    s1 = wave_sum(s1) / (float)d;
    s2 = wave_sum(s2) / (float)d;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        Vec8<T> o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o.set(i, rstd * (gd[c][i] - s1 - xh[c][i] * s2));
        store8<T>(dx + row * d + ch * 8, o);
      }
    }
  }
  for (int which = 0; which < 2; ++which) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) red[wave][lane * 8 + i] = which ? ba[c][i] : ga[c][i];
      __syncthreads();
      if (wave == 0) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            partial[((int64_t)which * gridDim.x + blockIdx.x) * d + ch * 8 + i] =
                red[0][lane * 8 + i] + red[1][lane * 8 + i] + red[2][lane * 8 + i] + red[3][lane * 8 + i];
        }
      }
    }
  }
}

inline int packed_blocks(int64_t groups) {
  int64_t b = ceil_div(groups, 4);
  return (int)(b < 1 ? 1 : (b > NORM_PACKED_BLOCKS ? NORM_PACKED_BLOCKS : b));
}
// the packed backward kernels hold two workgroups per CU (205 VGPRs): more than 512 workgroups run in rounds, each paying the
// prologue (gain loads) and the epilogue (LDS reduction of the gain gradient, partial row) again -- at the MLM step's 32 k rows
// the 2048-block launch spent two thirds of its 98 us there
inline int packed_blocks_bwd(int64_t groups) {
  int64_t b = ceil_div(groups, 4);
  const int64_t cap = 2 * (int64_t)meant_num_cus();
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}
inline int norm_blocks(int64_t rows) {
  int64_t b = ceil_div(rows, 4);
  return (int)(b < 1 ? 1 : (b > NORM_MAX_BLOCKS ? NORM_MAX_BLOCKS : b));
}

}  // namespace

extern "C" int meant_rmsnorm_fwd(const void* x, const float* scale, void* y, float* rinv, int64_t rows, int64_t d,
                                 float eps, float drop_p, uint64_t seed, int dtype, void* stream) {
  MEANT_REQUIRE(x && scale && y && rinv, MEANT_ERR_ARG, "rmsnorm_fwd: null pointer");
  MEANT_REQUIRE(rows >= 0 && d > 0 && d % 8 == 0 && d <= MAXC * 512, MEANT_ERR_UNSUPPORTED,
                "rmsnorm_fwd: d=%lld must be a multiple of 8 and <= %d", (long long)d, MAXC * 512);
  MEANT_REQUIRE(meant_aligned16(x) && meant_aligned16(y) && meant_aligned16(scale), MEANT_ERR_ARG, "rmsnorm_fwd: 16-byte alignment");
  MEANT_REQUIRE(drop_p >= 0.f && drop_p < 1.f, MEANT_ERR_ARG, "rmsnorm_fwd: drop_p out of range");
  if (rows == 0) return MEANT_OK;
  int R, C;
  norm_packing(rows, d, R, C);
  if (R) {
    const int nb = packed_blocks(rows / R);
#define LAUNCH_FWD(CC)                                                                                                        \
    DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((rmsnorm_fwd_packed_kernel<T, CC>), dim3(nb), dim3(NORM_THREADS), 0, (hipStream_t)stream, \
                                                (const T*)x, scale, (T*)y, rinv, rows, (int)d, R, eps, drop_p, seed))
    if (C == 1) LAUNCH_FWD(1); else if (C == 2) LAUNCH_FWD(2); else LAUNCH_FWD(3);
#undef LAUNCH_FWD
    MEANT_LAUNCH_CHECK("rmsnorm_fwd");
    return MEANT_OK;
  }
  DISPATCH_DTYPE(dtype, T,
                 hipLaunchKernelGGL(rmsnorm_fwd_kernel<T>, dim3(norm_blocks(rows)), dim3(NORM_THREADS), 0, (hipStream_t)stream,
                                    (const T*)x, scale, (T*)y, rinv, rows, (int)d, eps, drop_p, seed));
  MEANT_LAUNCH_CHECK("rmsnorm_fwd");
  return MEANT_OK;
}

extern "C" size_t meant_rmsnorm_bwd_ws(int64_t rows, int64_t d) {
  const int64_t nb = norm_blocks(rows) > packed_blocks(rows) ? norm_blocks(rows) : packed_blocks(rows);
  return (size_t)nb * d * sizeof(float) * 2;
}

extern "C" int meant_rmsnorm_bwd(const void* dy, const void* x, const float* scale, const float* rinv, void* dx,
                                 float* dscale, int64_t rows, int64_t d, float eps, float drop_p, uint64_t seed,
                                 const void* dres, const void* gelu_pre, int dtype, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  MEANT_REQUIRE(dy && x && scale && rinv && dx && dscale && workspace, MEANT_ERR_ARG, "rmsnorm_bwd: null pointer");
  MEANT_REQUIRE(rows > 0 && d > 0 && d % 8 == 0 && d <= MAXC * 512, MEANT_ERR_UNSUPPORTED, "rmsnorm_bwd: unsupported d=%lld", (long long)d);
  MEANT_REQUIRE(workspace_bytes >= meant_rmsnorm_bwd_ws(rows, d), MEANT_ERR_WORKSPACE, "rmsnorm_bwd: workspace too small");
  int R, C;
  norm_packing(rows, d, R, C);
  if (R) {
    const int nbp = packed_blocks_bwd(rows / R);
#define LAUNCH_BWD(CC)                                                                                                        \
    DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((rmsnorm_bwd_packed_kernel<T, CC, 0>), dim3(nbp), dim3(NORM_THREADS), 0, (hipStream_t)stream, \
                                                (const T*)dy, (const T*)x, scale, rinv, (T*)dx, (float*)workspace, rows, (int)d, R, eps,    \
                                                drop_p, seed, (const T*)dres, (const T*)gelu_pre, 1))
    if (C == 1) LAUNCH_BWD(1); else if (C == 2) LAUNCH_BWD(2); else LAUNCH_BWD(3);
#undef LAUNCH_BWD
    MEANT_LAUNCH_CHECK("rmsnorm_bwd");
    return colsum_launch(workspace, d, dscale, nbp, d, MEANT_F32, 0, (hipStream_t)stream);
  }
  const int nb = norm_blocks(rows);
  DISPATCH_DTYPE(dtype, T,
                 hipLaunchKernelGGL(rmsnorm_bwd_kernel<T>, dim3(nb), dim3(NORM_THREADS), 0, (hipStream_t)stream, (const T*)dy,
                                    (const T*)x, scale, rinv, (T*)dx, (float*)workspace, rows, (int)d, eps, drop_p, seed,
                                    (const T*)dres, (const T*)gelu_pre));
  MEANT_LAUNCH_CHECK("rmsnorm_bwd");
  return colsum_launch(workspace, d, dscale, nb, d, MEANT_F32, 0, (hipStream_t)stream);
}

// ---- partial / bias forms of the reference class (utils/rms_norm.py:44-57); not used by any MEANT model, generic kernels ----
extern "C" int meant_rmsnorm_partial_fwd(const void* x, const float* scale, const float* offset, void* y, float* rinv, int64_t rows,
                                         int64_t d, int64_t d_part, float eps, int dtype, void* stream) {
  MEANT_REQUIRE(x && scale && y && rinv, MEANT_ERR_ARG, "rmsnorm_partial_fwd: null pointer");
  MEANT_REQUIRE(rows >= 0 && d > 0 && d % 8 == 0 && d <= MAXC * 512 && d_part >= 1 && d_part <= d, MEANT_ERR_UNSUPPORTED,
                "rmsnorm_partial_fwd: d=%lld must be a multiple of 8 and <= %d, 1 <= d_part=%lld <= d", (long long)d, MAXC * 512, (long long)d_part);
  MEANT_REQUIRE(meant_aligned16(x) && meant_aligned16(y) && meant_aligned16(scale), MEANT_ERR_ARG, "rmsnorm_partial_fwd: 16-byte alignment");
  if (rows == 0) return MEANT_OK;
  DISPATCH_DTYPE(dtype, T,
                 hipLaunchKernelGGL(rmsnorm_fwd_kernel<T>, dim3(norm_blocks(rows)), dim3(NORM_THREADS), 0, (hipStream_t)stream,
                                    (const T*)x, scale, (T*)y, rinv, rows, (int)d, eps, 0.f, (uint64_t)0, (int)d_part, offset));
  MEANT_LAUNCH_CHECK("rmsnorm_partial_fwd");
  return MEANT_OK;
}

extern "C" int meant_rmsnorm_partial_bwd(const void* dy, const void* x, const float* scale, const float* rinv, void* dx, float* dscale,
                                         float* doffset, int64_t rows, int64_t d, int64_t d_part, float eps, int dtype, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  MEANT_REQUIRE(dy && x && scale && rinv && dx && dscale && workspace, MEANT_ERR_ARG, "rmsnorm_partial_bwd: null pointer");
  MEANT_REQUIRE(rows > 0 && d > 0 && d % 8 == 0 && d <= MAXC * 512 && d_part >= 1 && d_part <= d, MEANT_ERR_UNSUPPORTED, "rmsnorm_partial_bwd: unsupported shape");
  MEANT_REQUIRE(workspace_bytes >= meant_rmsnorm_bwd_ws(rows, d), MEANT_ERR_WORKSPACE, "rmsnorm_partial_bwd: workspace too small");
  const int nb = norm_blocks(rows);
  float* part1 = (float*)workspace;
  float* part2 = doffset ? part1 + (size_t)nb * d : nullptr;
  DISPATCH_DTYPE(dtype, T,
                 hipLaunchKernelGGL(rmsnorm_bwd_kernel<T>, dim3(nb), dim3(NORM_THREADS), 0, (hipStream_t)stream, (const T*)dy,
                                    (const T*)x, scale, rinv, (T*)dx, part1, rows, (int)d, eps, 0.f, (uint64_t)0,
                                    (const T*)nullptr, (const T*)nullptr, (int)d_part, part2));
  MEANT_LAUNCH_CHECK("rmsnorm_partial_bwd");
  int rc = colsum_launch(part1, d, dscale, nb, d, MEANT_F32, 0, (hipStream_t)stream);
  if (rc || !doffset) return rc;
  return colsum_launch(part2, d, doffset, nb, d, MEANT_F32, 0, (hipStream_t)stream);
}

// ---- pooled forms (the norms whose output or input feeds the sequence mean-pool only) ----
extern "C" int meant_rmsnorm_pooled_ok(int64_t rows, int64_t d, int64_t group_rows) {
  if (rows <= 0 || d <= 0 || d % 8 || d > MAXC * 512 || group_rows <= 0 || rows % group_rows) return 0;
  int R, C;
  norm_packing(rows, d, R, C);
  return R != 0 && group_rows % R == 0;
}

extern "C" int meant_rmsnorm_fwd_pooled(const void* x, const float* scale, void* y, float* rinv, float* pooled, int64_t rows, int64_t d,
                                        int64_t group_rows, int pool_input, int gelu_input, float eps, float drop_p, uint64_t seed,
                                        int dtype, void* stream) {
  MEANT_REQUIRE(x && scale && rinv && pooled, MEANT_ERR_ARG, "rmsnorm_fwd_pooled: null pointer");   // y == NULL with pool_input: statistics + means of x only
  MEANT_REQUIRE(!(pool_input && gelu_input), MEANT_ERR_UNSUPPORTED, "rmsnorm_fwd_pooled: gelu_input goes with pool_input == 0");
  MEANT_REQUIRE(meant_rmsnorm_pooled_ok(rows, d, group_rows), MEANT_ERR_UNSUPPORTED,
                "rmsnorm_fwd_pooled: rows=%lld d=%lld group_rows=%lld is not a packed shape (see meant_rmsnorm_pooled_ok)", (long long)rows,
                (long long)d, (long long)group_rows);
  MEANT_REQUIRE(meant_aligned16(x) && meant_aligned16(scale) && (!y || meant_aligned16(y)), MEANT_ERR_ARG, "rmsnorm_fwd_pooled: 16-byte alignment");
  MEANT_REQUIRE(drop_p >= 0.f && drop_p < 1.f, MEANT_ERR_ARG, "rmsnorm_fwd_pooled: drop_p out of range");
  int R, C;
  norm_packing(rows, d, R, C);
  const int64_t ngroups = rows / group_rows;
  const int nb = (int)(ngroups < NORM_PACKED_BLOCKS ? ngroups : NORM_PACKED_BLOCKS);
#define LAUNCH_FWDP(CC, PP, GG)                                                                                               \
    DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((rmsnorm_fwd_pooled_kernel<T, CC, PP, GG>), dim3(nb), dim3(NORM_THREADS), 0, (hipStream_t)stream, \
                                                (const T*)x, scale, (T*)y, rinv, ngroups, (int)d, R, eps, drop_p, seed, pooled, (int)group_rows))
#define LAUNCH_FWDP_C(PP, GG) { if (C == 1) LAUNCH_FWDP(1, PP, GG); else if (C == 2) LAUNCH_FWDP(2, PP, GG); else LAUNCH_FWDP(3, PP, GG); }
  if (pool_input) LAUNCH_FWDP_C(2, false)
  else if (gelu_input) LAUNCH_FWDP_C(1, true)
  else LAUNCH_FWDP_C(1, false)
#undef LAUNCH_FWDP_C
#undef LAUNCH_FWDP
  MEANT_LAUNCH_CHECK("rmsnorm_fwd_pooled");
  return MEANT_OK;
}

extern "C" int meant_rmsnorm_bwd_pooled(const void* dy, int dy_pooled, const void* x, const float* scale, const float* rinv, void* dx,
                                        float* dscale, int64_t rows, int64_t d, int64_t group_rows, float eps, float drop_p, uint64_t seed,
                                        const void* dres, int dres_pooled, const void* gelu_pre, int dtype, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  MEANT_REQUIRE(dy && (x || gelu_pre) && scale && rinv && dx && dscale && workspace, MEANT_ERR_ARG, "rmsnorm_bwd_pooled: null pointer");
  MEANT_REQUIRE((dy_pooled || dres_pooled) && (!dres_pooled || dres), MEANT_ERR_ARG, "rmsnorm_bwd_pooled: nothing pooled (use meant_rmsnorm_bwd)");
  MEANT_REQUIRE(meant_rmsnorm_pooled_ok(rows, d, group_rows) && rows < 2147483647LL, MEANT_ERR_UNSUPPORTED, "rmsnorm_bwd_pooled: not a packed shape");
  MEANT_REQUIRE(workspace_bytes >= meant_rmsnorm_bwd_ws(rows, d), MEANT_ERR_WORKSPACE, "rmsnorm_bwd_pooled: workspace too small");
  int R, C;
  norm_packing(rows, d, R, C);
  const int nbp = packed_blocks_bwd(rows / R);
  const int bc = (dy_pooled ? 1 : 0) | (dres_pooled ? 2 : 0) | (x ? 0 : 4);
  MEANT_REQUIRE(bc == 1 || bc == 2 || bc == 3 || bc == 5, MEANT_ERR_UNSUPPORTED, "rmsnorm_bwd_pooled: unsupported combination");
#define LAUNCH_BWDP(CC, BB)                                                                                                   \
    DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((rmsnorm_bwd_packed_kernel<T, CC, BB>), dim3(nbp), dim3(NORM_THREADS), 0, (hipStream_t)stream, \
                                                (const T*)dy, (const T*)x, scale, rinv, (T*)dx, (float*)workspace, rows, (int)d, R, eps,    \
                                                drop_p, seed, (const T*)dres, (const T*)gelu_pre, (int)group_rows))
#define LAUNCH_BWDP_C(BB) { if (C == 1) LAUNCH_BWDP(1, BB); else if (C == 2) LAUNCH_BWDP(2, BB); else LAUNCH_BWDP(3, BB); }
  if (bc == 1) LAUNCH_BWDP_C(1) else if (bc == 2) LAUNCH_BWDP_C(2) else if (bc == 3) LAUNCH_BWDP_C(3) else LAUNCH_BWDP_C(5)
#undef LAUNCH_BWDP_C
#undef LAUNCH_BWDP
  MEANT_LAUNCH_CHECK("rmsnorm_bwd_pooled");
  return colsum_launch(workspace, d, dscale, nbp, d, MEANT_F32, 0, (hipStream_t)stream);
}

// ---- RMSNorm folded into the consumer Linear (meant_linear_fwd_rowscale): statistics pass and chained backward ----
extern "C" int meant_rmsnorm_stats(const void* x, float* rinv, int64_t rows, int64_t d, float eps, int dtype, void* stream) {
  MEANT_REQUIRE(x && rinv, MEANT_ERR_ARG, "rmsnorm_stats: null pointer");
  MEANT_REQUIRE(meant_aligned16(x), MEANT_ERR_ARG, "rmsnorm_stats: 16-byte alignment");
  int R, C;
  norm_packing(rows, d, R, C);
  MEANT_REQUIRE(rows > 0 && d > 0 && d % 8 == 0 && d <= MAXC * 512 && R != 0, MEANT_ERR_UNSUPPORTED,
                "rmsnorm_stats: rows=%lld d=%lld is not a packed shape (see meant_rmsnorm_pooled_ok)", (long long)rows, (long long)d);
  const int nb = packed_blocks(rows / R);
#define LAUNCH_STATS(CC)                                                                                                      \
    DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((rmsnorm_stats_packed_kernel<T, CC>), dim3(nb), dim3(NORM_THREADS), 0, (hipStream_t)stream, \
                                                (const T*)x, rinv, rows, (int)d, R, eps))
  if (C == 1) LAUNCH_STATS(1); else if (C == 2) LAUNCH_STATS(2); else LAUNCH_STATS(3);
#undef LAUNCH_STATS
  MEANT_LAUNCH_CHECK("rmsnorm_stats");
  return MEANT_OK;
}

extern "C" int meant_rmsnorm_bwd_chain(const void* dy, int dy_pooled, const void* x, const float* scale, const float* rinv, void* dx_scaled,
                                       float* dscale, int64_t rows, int64_t d, int64_t group_rows, float eps, float drop_p, uint64_t seed,
                                       const void* gelu_pre, const float* up_rinv, const float* up_bias, float up_eps, int64_t up_d,
                                       float* kcoef, float* dbias_up, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  MEANT_REQUIRE(dy && gelu_pre && scale && rinv && dx_scaled && dscale && up_rinv && up_bias && kcoef && dbias_up && workspace, MEANT_ERR_ARG,
                "rmsnorm_bwd_chain: null pointer");
  MEANT_REQUIRE((x != nullptr) != (dy_pooled != 0), MEANT_ERR_UNSUPPORTED,
                "rmsnorm_bwd_chain: either token-level dy with the stored activation x, or pooled dy with x formed from gelu_pre");
  int R, C;
  norm_packing(rows, d, R, C);
  MEANT_REQUIRE(rows > 0 && rows < 2147483647LL && R != 0 && (!dy_pooled || (group_rows > 0 && rows % group_rows == 0 && group_rows % R == 0)),
                MEANT_ERR_UNSUPPORTED, "rmsnorm_bwd_chain: not a packed shape");
  MEANT_REQUIRE(workspace_bytes >= meant_rmsnorm_bwd_ws(rows, d), MEANT_ERR_WORKSPACE, "rmsnorm_bwd_chain: workspace too small");
  const int nbp = packed_blocks_bwd(rows / R);
  float* part1 = (float*)workspace;
  float* part2 = part1 + (size_t)nbp * d;
  MEANT_REQUIRE(up_d > 0 && up_d < (1LL << 30), MEANT_ERR_ARG, "rmsnorm_bwd_chain: bad up_d");
  const NormChain ch{up_rinv, up_bias, kcoef, part2, up_eps, (int)up_d};
#define LAUNCH_CH(CC, BB)                                                                                                     \
    DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((rmsnorm_bwd_packed_kernel<T, CC, BB, true>), dim3(nbp), dim3(NORM_THREADS), 0, (hipStream_t)stream, \
                                                (const T*)dy, (const T*)x, scale, rinv, (T*)dx_scaled, part1, rows, (int)d, R, eps, drop_p, seed,  \
                                                (const T*)nullptr, (const T*)gelu_pre, (int)(dy_pooled ? group_rows : 1), ch))
#define LAUNCH_CH_C(BB) { if (C == 1) LAUNCH_CH(1, BB); else if (C == 2) LAUNCH_CH(2, BB); else LAUNCH_CH(3, BB); }
  if (dy_pooled) LAUNCH_CH_C(5) else LAUNCH_CH_C(0)
#undef LAUNCH_CH_C
#undef LAUNCH_CH
  MEANT_LAUNCH_CHECK("rmsnorm_bwd_chain");
  int rc = colsum_launch(part1, d, dscale, nbp, d, MEANT_F32, 0, (hipStream_t)stream);
  if (rc) return rc;
  return colsum_launch(part2, d, dbias_up, nbp, d, MEANT_F32, 1, (hipStream_t)stream);      // += : a gradient sink may be handed in
}

extern "C" int meant_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats,
                                   int64_t rows, int64_t d, float eps, int dtype, void* stream) {
  MEANT_REQUIRE(x && gamma && beta && y && stats, MEANT_ERR_ARG, "layernorm_fwd: null pointer");
  MEANT_REQUIRE(rows > 0 && d % 8 == 0 && d <= MAXC * 512, MEANT_ERR_UNSUPPORTED, "layernorm_fwd: unsupported d=%lld", (long long)d);
  DISPATCH_DTYPE(dtype, T,
                 hipLaunchKernelGGL(layernorm_fwd_kernel<T>, dim3(norm_blocks(rows)), dim3(NORM_THREADS), 0, (hipStream_t)stream,
                                    (const T*)x, gamma, beta, (T*)y, stats, rows, (int)d, eps));
  MEANT_LAUNCH_CHECK("layernorm_fwd");
  return MEANT_OK;
}

extern "C" int meant_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* stats, void* dx,
                                   float* dgamma, float* dbeta, int64_t rows, int64_t d, int dtype, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  MEANT_REQUIRE(dy && x && gamma && stats && dx && dgamma && dbeta && workspace, MEANT_ERR_ARG, "layernorm_bwd: null pointer");
  MEANT_REQUIRE(rows > 0 && d % 8 == 0 && d <= MAXC * 512, MEANT_ERR_UNSUPPORTED, "layernorm_bwd: unsupported d=%lld", (long long)d);
  MEANT_REQUIRE(workspace_bytes >= meant_rmsnorm_bwd_ws(rows, d), MEANT_ERR_WORKSPACE, "layernorm_bwd: workspace too small");
  const int nb = norm_blocks(rows);
  DISPATCH_DTYPE(dtype, T,
                 hipLaunchKernelGGL(layernorm_bwd_kernel<T>, dim3(nb), dim3(NORM_THREADS), 0, (hipStream_t)stream, (const T*)dy,
                                    (const T*)x, gamma, stats, (T*)dx, (float*)workspace, rows, (int)d));
  MEANT_LAUNCH_CHECK("layernorm_bwd");
  int rc = colsum_launch(workspace, d, dgamma, nb, d, MEANT_F32, 0, (hipStream_t)stream);
  if (rc) return rc;
  return colsum_launch((const float*)workspace + (size_t)nb * d, d, dbeta, nb, d, MEANT_F32, 0, (hipStream_t)stream);
}
