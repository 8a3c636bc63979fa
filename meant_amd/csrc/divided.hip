// Divided space-time attention glue (src/meant/timesformer_pytorch.py:108-145): what surrounds the flash
// attention core when tokens are regrouped between the projection and the core.
//   * meant_gather_rows     -- the '(b n) f' / '(b f) n' regrouping as a row gather by an index table (and its inverse for
//                              the outputs, and the cls-token concatenation at the model's entry): 16 bytes per lane, a wave
//                              per row; index -1 writes zeros, -2 a caller-supplied row.
//   * meant_group_scatter   -- backward of the regrouping: every non-cls row of a group goes home by a copy (the groups
//                              partition them), the cls row, which every group carries at position 0, is the sum over groups.
//   * meant_attn_cls_fwd/bwd -- the cls query's attention over ALL tokens (:116-119): one query per (video, head), so no MFMA
//                              tile fits; scores by a thread per key, probabilities in LDS, the weighted sum by a thread per
//                              head column.  The backward ADDS its dK / dV (every row) and dQ (row 0) to the gradient buffer
//                              the regrouped attention's backward has already filled -- no separate tensor, no add pass.
// All HBM / latency bound and small next to the GEMMs (a few hundred MB per layer).
#include "common.h"

namespace {

constexpr int DIV_THREADS = 256;

template <typename T>
__global__ __launch_bounds__(DIV_THREADS) void gather_rows_kernel(const T* __restrict__ src, const int* __restrict__ idx, const T* __restrict__ fill,
                                                                   T* __restrict__ dst, int64_t n, int W) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = W >> 3;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
    const int id = idx[r];
    const T* s = id >= 0 ? src + (int64_t)id * W : fill;
    for (int ch = lane; ch < nch; ch += 64) {
      Vec8<T> v;
      if (id >= 0 || (id == -2 && fill)) v = load8<T>(s + ch * 8);
      else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v.set(k, 0.f);
      }
      store8s<T>(dst + r * W + ch * 8, v);
    }
  }
}

// the regrouping gather of a packed q|k|v buffer with the rotary map applied on the way (row r sits at position r % S of its
// group): one pass instead of gather + in-place rotary.  Same arithmetic as rotary_kernel (elementwise.hip): fp32 on the stored
// values, one rounding, pairs (2j, 2j+1), columns >= R of a head and the v third copied.
template <typename T>
__global__ __launch_bounds__(DIV_THREADS) void gather_rows_rot_kernel(const T* __restrict__ src, const int* __restrict__ idx, T* __restrict__ dst, int64_t n,
                                                                       int S, int HD, int Dh, int R, const float* __restrict__ qa,
                                                                       const float* __restrict__ qb, const float* __restrict__ ka,
                                                                       const float* __restrict__ kb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int W = 3 * HD, nch = W >> 3;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
    const int id = idx[r];
    const int pos = (int)(r % S);
    for (int ch = lane; ch < nch; ch += 64) {
      Vec8<T> v;
      if (id >= 0) v = load8<T>(src + (int64_t)id * W + ch * 8);
      else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v.set(k, 0.f);
      }
      const int col = ch * 8, which = col / HD, dh = (col - which * HD) % Dh;
      if (which < 2 && dh < R && id >= 0) {            // R % 8 == 0: a chunk is rotated as a whole or not at all
        const float* A = (which ? ka : qa) + (int64_t)pos * R + dh;
        const float* B = (which ? kb : qb) + (int64_t)pos * R + dh;
        Vec8<T> o;
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
          float o0, o1;
          rotary_pair(v.get(k), v.get(k + 1), A[k], A[k + 1], B[k], B[k + 1], o0, o1);
          o.set(k, o0);
          o.set(k + 1, o1);
        }
        v = o;
      }
      store8s<T>(dst + r * W + ch * 8, v);
    }
  }
}

// rows s >= 1 of every group: dsrc[b, index[g, s], :] = ddst[b, g, s, :]
template <typename T>
__global__ __launch_bounds__(DIV_THREADS) void group_scatter_rows_kernel(const T* __restrict__ ddst, const int* __restrict__ index, T* __restrict__ dsrc,
                                                                          int64_t B, int64_t L, int G, int S, int W) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = W >> 3;
  const int64_t n = B * G * (int64_t)S;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
    const int64_t b = r / ((int64_t)G * S);
    const int gs = (int)(r - b * (int64_t)G * S);
    if (gs % S == 0) continue;                        // position 0 of a group is the shared cls row
    T* d = dsrc + (b * L + index[gs]) * W;
    for (int ch = lane; ch < nch; ch += 64) store8s<T>(d + ch * 8, load8s<T>(ddst + r * W + ch * 8));
  }
}
// the cls row: dsrc[b, index[0], :] = sum_g ddst[b, g, 0, :]   (fp32 sum, fixed order)
template <typename T>
__global__ __launch_bounds__(DIV_THREADS) void group_scatter_cls_kernel(const T* __restrict__ ddst, const int* __restrict__ index, T* __restrict__ dsrc,
                                                                         int64_t L, int G, int S, int W) {
  const int64_t b = blockIdx.y;
  const int ch = blockIdx.x * DIV_THREADS + threadIdx.x;
  if (ch * 8 >= W) return;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int g = 0; g < G; ++g) {
    const Vec8<T> v = load8<T>(ddst + ((b * G + g) * (int64_t)S) * W + ch * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += v.get(k);
  }
  Vec8<T> o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o.set(k, acc[k]);
  store8<T>(dsrc + (b * L + index[0]) * W + ch * 8, o);
}

// ---- cls query attention -----------------------------------------------------------------------
// block = (head h, video b), 256 threads.  LDS: p[L] floats (+ ds[L] in the backward), then small reduction areas.
constexpr float CLS_PAD = -1e9f;                     // (1 - mask) * -1e9 on the scaled score, as the text path's padding term

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}

template <typename T>
__global__ __launch_bounds__(DIV_THREADS) void attn_cls_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, int64_t ld_out,
                                                                    float* __restrict__ stats, const float* __restrict__ key_mask, int L, int H,
                                                                    int Dh, float scale) {
  extern __shared__ float sm[];
  float* p = sm;                                     // [L]
  float* red = sm + L;                               // [4], then [rpi][Dh] partial outputs
  const int h = blockIdx.x, tid = threadIdx.x;
  const int64_t b = blockIdx.y;
  const int64_t D = (int64_t)H * Dh, ld = 3 * D;
  const T* base = qkv + b * L * ld + h * Dh;
  // scores: a thread per key, the query broadcast from the cls row
  float mx = -INFINITY;
  for (int j = tid; j < L; j += DIV_THREADS) {
    const T* kr = base + (int64_t)j * ld + D;
    float s = 0.f;
    for (int c = 0; c < Dh; c += 8) {
      const Vec8<T> kv = load8<T>(kr + c), qv = load8<T>(base + c);
#pragma unroll
      for (int k = 0; k < 8; ++k) s += kv.get(k) * qv.get(k);
    }
    s *= scale;
    if (key_mask) s += (1.0f - key_mask[b * L + j]) * CLS_PAD;
    p[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = block_reduce(mx, red, true);
  float sum = 0.f;
  for (int j = tid; j < L; j += DIV_THREADS) {
    const float e = __expf(p[j] - mx);
    p[j] = e;
    sum += e;
  }
  sum = block_reduce(sum, red, false);                // (its barriers also publish p[])
  if (tid == 0) { stats[(b * H + h) * 2] = mx; stats[(b * H + h) * 2 + 1] = __logf(sum); }
  // out[d] = sum_j p_j v[j][d] / sum: a thread owns 8 columns of the rows j = rg, rg + rpi, ... (16-byte loads), the
  // rpi = 256 / (Dh / 8) row groups are combined through LDS
  const int nc = Dh >> 3, rpi = DIV_THREADS / nc;
  const int c8 = (tid % nc) * 8, rg = tid / nc;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = rg; j < L; j += rpi) {
    const Vec8<T> vv = load8<T>(base + (int64_t)j * ld + 2 * D + c8);
    const float pj = p[j];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += pj * vv.get(k);
  }
  float* part = red + 4;                             // [rpi][Dh]
#pragma unroll
  for (int k = 0; k < 8; ++k) part[rg * Dh + c8 + k] = acc[k];
  __syncthreads();
  if (tid < Dh) {
    float o = 0.f;
    for (int r = 0; r < rpi; ++r) o += part[r * Dh + tid];
    out[b * ld_out + h * Dh + tid] = from_f<T>(o / sum);
  }
}

template <typename T>
__global__ __launch_bounds__(DIV_THREADS) void attn_cls_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ out, int64_t ld_out,
                                                                    const T* __restrict__ dout, int64_t ld_dout, const float* __restrict__ stats,
                                                                    const float* __restrict__ key_mask, T* __restrict__ dqkv, int L, int H, int Dh,
                                                                    float scale) {
  extern __shared__ float sm[];
  float* p = sm;                                     // [L]
  float* ds = sm + L;                                // [L]
  float* red = ds + L;                               // [4], then [rpi][Dh] partial dQ
  const int h = blockIdx.x, tid = threadIdx.x;
  const int64_t b = blockIdx.y;
  const int64_t D = (int64_t)H * Dh, ld = 3 * D;
  const T* base = qkv + b * L * ld + h * Dh;
  T* dbase = dqkv + b * L * ld + h * Dh;
  const T* dor = dout + b * ld_dout + h * Dh;
  const T* orow = out + b * ld_out + h * Dh;
  const float mx = stats[(b * H + h) * 2], lsum = stats[(b * H + h) * 2 + 1];
  float delta = 0.f;                                  // dout . out
  for (int c = tid; c < Dh; c += DIV_THREADS) delta += to_f(dor[c]) * to_f(orow[c]);
  delta = block_reduce(delta, red, false);
  for (int j = tid; j < L; j += DIV_THREADS) {
    const T* kr = base + (int64_t)j * ld + D;
    const T* vr = base + (int64_t)j * ld + 2 * D;
    float s = 0.f, dp = 0.f;
    for (int c = 0; c < Dh; c += 8) {
      const Vec8<T> kv = load8<T>(kr + c), qv = load8<T>(base + c), vv = load8<T>(vr + c), dv = load8<T>(dor + c);
#pragma unroll
      for (int k = 0; k < 8; ++k) { s += kv.get(k) * qv.get(k); dp += vv.get(k) * dv.get(k); }
    }
    s *= scale;
    if (key_mask) s += (1.0f - key_mask[b * L + j]) * CLS_PAD;
    const float pj = __expf((s - mx) - lsum);
    p[j] = pj;
    ds[j] = pj * (dp - delta) * scale;               // d score_j * scale
  }
  __syncthreads();
  // dK[j] += ds_j q, dV[j] += p_j dout, dQ = sum_j ds_j k[j]: a thread owns 8 columns of the rows j = rg, rg + rpi, ...
  // (16-byte read-modify-writes; the (video, head) blocks touch disjoint columns, so nothing races)
  const int nc = Dh >> 3, rpi = DIV_THREADS / nc;
  const int c8 = (tid % nc) * 8, rg = tid / nc;
  const Vec8<T> qv = load8<T>(base + c8), dov = load8<T>(dor + c8);
  float dq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = rg; j < L; j += rpi) {
    const int64_t o = (int64_t)j * ld;
    const Vec8<T> kv = load8<T>(base + o + D + c8);
    Vec8<T> gk = load8<T>(dbase + o + D + c8), gv = load8<T>(dbase + o + 2 * D + c8);
    const float dsj = ds[j], pj = p[j];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      dq[k] += dsj * kv.get(k);
      gk.set(k, gk.get(k) + dsj * qv.get(k));
      gv.set(k, gv.get(k) + pj * dov.get(k));
    }
    store8<T>(dbase + o + D + c8, gk);
    store8<T>(dbase + o + 2 * D + c8, gv);
  }
  float* part = red + 4;                             // [rpi][Dh]
#pragma unroll
  for (int k = 0; k < 8; ++k) part[rg * Dh + c8 + k] = dq[k];
  __syncthreads();
  if (tid < Dh) {
    float o = 0.f;
    for (int r = 0; r < rpi; ++r) o += part[r * Dh + tid];
    dbase[tid] = from_f<T>(to_f(dbase[tid]) + o);    // the q part of the cls row
  }
}

// time token shift (src/meant/timesformer_pytorch.py:28-53): of the patch tokens [b, f, n, d] the first d/3 columns come from
// the NEXT frame, the second third stays, the third third comes from the PREVIOUS frame (zeros beyond the clip's ends); the
// cls row and the columns past 3 * (d / 3) pass through.  transpose = the adjoint (the two shifts swap).
template <typename T>
__global__ __launch_bounds__(DIV_THREADS) void token_shift_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t B, int f, int n, int d,
                                                                   int chunk, int transpose) {
  const int nch = d >> 3;
  const int64_t L = 1 + (int64_t)f * n, total = B * L * nch;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % nch);
    const int64_t row = i / nch, b = row / L, t = row - b * L;
    int64_t src = row;
    bool zero = false;
    if (t > 0) {
      const int which = (ch * 8) / chunk;              // 0, 1, 2: shifted thirds; >= 3: the remainder columns
      int sh = which == 0 ? 1 : (which == 2 ? -1 : 0); // source frame = frame + sh
      if (transpose) sh = -sh;
      const int fi = (int)((t - 1) / n) + sh;
      zero = fi < 0 || fi >= f;
      src = row + (int64_t)sh * n;
    }
    Vec8<T> v;
    if (zero) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v.set(k, 0.f);
    } else v = load8<T>(x + src * d + ch * 8);
    store8<T>(y + row * d + ch * 8, v);
  }
}

// inverted dropout y = x * keep / (1 - p) with the counter-based mask of the fused RMSNorm dropout (common.h keep_scale8): the
// same (seed, element index) gives the same mask, so the backward is the same call on dy
template <typename T>
__global__ __launch_bounds__(DIV_THREADS) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n8, float p, uint64_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float km[8];
    keep_scale8(p, seed, (uint64_t)i * 8, km);
    const Vec8<T> v = load8<T>(x + i * 8);
    Vec8<T> o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.set(k, v.get(k) * km[k]);
    store8<T>(y + i * 8, o);
  }
}

inline int rows_blocks(int64_t n) {
  int64_t b = ceil_div(n, 4);
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

#define DIV_REQ(cond, ...) MEANT_REQUIRE(cond, MEANT_ERR_ARG, __VA_ARGS__)

extern "C" int meant_gather_rows(const void* src, const int32_t* idx, const void* fill, void* dst, int64_t n, int64_t W, int dtype, void* stream) {
  DIV_REQ(src && idx && dst && n > 0 && W > 0 && W % 8 == 0 && W < (1LL << 30), "gather_rows: bad argument (W must be a multiple of 8)");
  DIV_REQ(meant_aligned16(src) && meant_aligned16(dst) && (!fill || meant_aligned16(fill)), "gather_rows: 16-byte alignment");
  DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(gather_rows_kernel<T>, dim3(rows_blocks(n)), dim3(DIV_THREADS), 0, (hipStream_t)stream, (const T*)src, idx,
                                              (const T*)fill, (T*)dst, n, (int)W));
  MEANT_LAUNCH_CHECK("gather_rows");
  return MEANT_OK;
}

extern "C" int meant_gather_rows_rot(const void* src, const int32_t* idx, void* dst, int64_t n, int64_t S, int H, int Dh, int R, const float* qa,
                                     const float* qb, const float* ka, const float* kb, int dtype, void* stream) {
  DIV_REQ(src && idx && dst && n > 0 && S > 0 && H > 0 && Dh > 0 && Dh % 8 == 0 && R >= 0 && R <= Dh && R % 8 == 0 && (int64_t)H * Dh * 3 < (1LL << 30),
          "gather_rows_rot: bad argument (Dh and the rotary dim must be multiples of 8, R <= Dh)");
  DIV_REQ(R == 0 || (qa && qb && ka && kb), "gather_rows_rot: null rotary table");
  DIV_REQ(meant_aligned16(src) && meant_aligned16(dst) && (R == 0 || (meant_aligned16(qa) && meant_aligned16(qb) && meant_aligned16(ka) && meant_aligned16(kb))),
          "gather_rows_rot: 16-byte alignment");
  DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(gather_rows_rot_kernel<T>, dim3(rows_blocks(n)), dim3(DIV_THREADS), 0, (hipStream_t)stream, (const T*)src, idx,
                                              (T*)dst, n, (int)S, H * Dh, Dh, R, qa, qb, ka, kb));
  MEANT_LAUNCH_CHECK("gather_rows_rot");
  return MEANT_OK;
}

extern "C" int meant_group_scatter(const void* ddst, const int32_t* index, void* dsrc, int64_t B, int64_t L, int64_t G, int64_t S, int64_t W, int dtype,
                                   void* stream) {
  DIV_REQ(ddst && index && dsrc && B > 0 && L > 0 && G > 0 && S > 1 && W > 0 && W % 8 == 0 && G * S < (1LL << 31) && B <= 65535,
          "group_scatter: bad argument");
  DIV_REQ(meant_aligned16(ddst) && meant_aligned16(dsrc), "group_scatter: 16-byte alignment");
  DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(group_scatter_rows_kernel<T>, dim3(rows_blocks(B * G * S)), dim3(DIV_THREADS), 0, (hipStream_t)stream,
                                              (const T*)ddst, index, (T*)dsrc, B, L, (int)G, (int)S, (int)W));
  MEANT_LAUNCH_CHECK("group_scatter");
  DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(group_scatter_cls_kernel<T>, dim3((unsigned)ceil_div(W / 8, DIV_THREADS), (unsigned)B), dim3(DIV_THREADS), 0,
                                              (hipStream_t)stream, (const T*)ddst, index, (T*)dsrc, L, (int)G, (int)S, (int)W));
  MEANT_LAUNCH_CHECK("group_scatter");
  return MEANT_OK;
}

static int cls_check(const char* name, int64_t B, int64_t L, int H, int Dh, size_t& lds, int arrays) {
  MEANT_REQUIRE(B > 0 && B <= 65535 && L > 0 && H > 0 && H <= 65535 && (Dh == 32 || Dh == 64 || Dh == 128 || Dh == 256), MEANT_ERR_UNSUPPORTED,
                "%s: head dim %d must be 32, 64, 128 or 256 (B=%lld, L=%lld, H=%d)", name, Dh, (long long)B, (long long)L, H);
  lds = ((size_t)arrays * L + 4 + DIV_THREADS * 8) * sizeof(float);     // scores (+ d scores), reduction words, [row groups][Dh] partials
  MEANT_REQUIRE(lds <= 160 * 1024 - 1024, MEANT_ERR_UNSUPPORTED, "%s: L=%lld tokens do not fit the LDS score buffer", name, (long long)L);
  return MEANT_OK;
}

extern "C" int meant_attn_cls_fwd(const void* qkv, void* out, int64_t ld_out, float* stats, const float* key_mask, int64_t B, int64_t L, int H,
                                  int Dh, float scale, int dtype, void* stream) {
  DIV_REQ(qkv && out && stats && ld_out >= (int64_t)H * Dh && meant_aligned16(qkv), "attn_cls_fwd: bad argument");
  size_t lds;
  int rc = cls_check("attn_cls_fwd", B, L, H, Dh, lds, 1);
  if (rc) return rc;
  meant_route_hit(ROUTE_ATTN_CLS);
#define LAUNCH_CLS_F(T)                                                                                                              \
  {                                                                                                                                  \
    MEANT_RAISE_LDS(attn_cls_fwd_kernel<T>, lds);                                                                                    \
    hipLaunchKernelGGL(attn_cls_fwd_kernel<T>, dim3((unsigned)H, (unsigned)B), dim3(DIV_THREADS), lds, (hipStream_t)stream, (const T*)qkv, \
                       (T*)out, ld_out, stats, key_mask, (int)L, H, Dh, scale);                                                      \
  }
  if (dtype == MEANT_F32) LAUNCH_CLS_F(float) else if (dtype == MEANT_BF16) LAUNCH_CLS_F(bf16) else { meant_set_error("attn_cls_fwd: unknown dtype"); return MEANT_ERR_ARG; }
#undef LAUNCH_CLS_F
  MEANT_LAUNCH_CHECK("attn_cls_fwd");
  return MEANT_OK;
}

extern "C" int meant_attn_cls_bwd(const void* qkv, const void* out, int64_t ld_out, const void* dout, int64_t ld_dout, const float* stats,
                                  const float* key_mask, void* dqkv, int64_t B, int64_t L, int H, int Dh, float scale, int dtype, void* stream) {
  DIV_REQ(qkv && out && dout && stats && dqkv && meant_aligned16(qkv) && meant_aligned16(dout), "attn_cls_bwd: bad argument");
  size_t lds;
  int rc = cls_check("attn_cls_bwd", B, L, H, Dh, lds, 2);
  if (rc) return rc;
#define LAUNCH_CLS_B(T)                                                                                                              \
  {                                                                                                                                  \
    MEANT_RAISE_LDS(attn_cls_bwd_kernel<T>, lds);                                                                                    \
    hipLaunchKernelGGL(attn_cls_bwd_kernel<T>, dim3((unsigned)H, (unsigned)B), dim3(DIV_THREADS), lds, (hipStream_t)stream, (const T*)qkv, \
                       (const T*)out, ld_out, (const T*)dout, ld_dout, stats, key_mask, (T*)dqkv, (int)L, H, Dh, scale);             \
  }
  if (dtype == MEANT_F32) LAUNCH_CLS_B(float) else if (dtype == MEANT_BF16) LAUNCH_CLS_B(bf16) else { meant_set_error("attn_cls_bwd: unknown dtype"); return MEANT_ERR_ARG; }
#undef LAUNCH_CLS_B
  MEANT_LAUNCH_CHECK("attn_cls_bwd");
  return MEANT_OK;
}

extern "C" int meant_token_shift(const void* x, void* y, int64_t B, int64_t frames, int64_t n, int64_t d, int transpose, int dtype, void* stream) {
  DIV_REQ(x && y && x != y && B > 0 && frames > 0 && n > 0 && d >= 24 && d % 8 == 0 && (d / 3) % 8 == 0 && frames * n < (1LL << 30),
          "token_shift: bad argument (d / 3 must be a multiple of 8)");
  DIV_REQ(meant_aligned16(x) && meant_aligned16(y), "token_shift: 16-byte alignment");
  const int64_t total = B * (1 + frames * n) * (d / 8);
  int64_t nb = ceil_div(total, DIV_THREADS);
  if (nb > 8192) nb = 8192;
  DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(token_shift_kernel<T>, dim3((unsigned)nb), dim3(DIV_THREADS), 0, (hipStream_t)stream, (const T*)x, (T*)y,
                                              B, (int)frames, (int)n, (int)d, (int)(d / 3), transpose));
  MEANT_LAUNCH_CHECK("token_shift");
  return MEANT_OK;
}

extern "C" int meant_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream) {
  DIV_REQ(x && y && n > 0 && n % 8 == 0 && p >= 0.f && p < 1.f, "dropout: bad argument (n must be a multiple of 8, 0 <= p < 1)");
  DIV_REQ(meant_aligned16(x) && meant_aligned16(y), "dropout: 16-byte alignment");
  int64_t nb = ceil_div(n / 8, DIV_THREADS);
  if (nb > 8192) nb = 8192;
  DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(dropout_kernel<T>, dim3((unsigned)nb), dim3(DIV_THREADS), 0, (hipStream_t)stream, (const T*)x, (T*)y, n / 8, p, seed));
  MEANT_LAUNCH_CHECK("dropout");
  return MEANT_OK;
}
