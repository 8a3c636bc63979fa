// Streaming / elementwise kernels of the MEANT path (all HBM-bound; fp32 math, T storage):
//   K6 rotary / xPos (in place on the packed q|k|v projection), K7a patchify, K8 sequence mean-pool,
//   K9 temporal (lag-axis) attention core, embedding gather / scatter-add, and small glue
//   (broadcast add, GELU / sigmoid backward, casts, weight transposes).
// Each thread moves 16 bytes (8 bf16) or 2x16 bytes (8 f32) per access; grids are capped and
// grid-strided so that launches stay at a few thousand workgroups.
#include "common.h"

namespace {

constexpr int EW_THREADS = 256;
inline int ew_blocks(int64_t items) {
  int64_t b = ceil_div(items, EW_THREADS);
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// ------------------------------------------------------------------------------------------------
// generic unary/binary elementwise over n elements (vector body + scalar tail)
template <typename T, typename F>
__global__ __launch_bounds__(EW_THREADS) void ew2_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y,
                                                          int64_t n, F f) {
  const int64_t nv = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    const Vec8<T> va = load8<T>(a + i * 8), vb = load8<T>(b + i * 8);
    Vec8<T> o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.set(k, f(va.get(k), vb.get(k)));
    store8<T>(y + i * 8, o);
  }
  for (int64_t i = (nv << 3) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = from_f<T>(f(to_f(a[i]), to_f(b[i])));
}

struct GeluBwdOp { __device__ float operator()(float dy, float pre) const { return dy * gelu_erf_grad(pre); } };
struct SigmoidBwdOp { __device__ float operator()(float dy, float y) const { return dy * y * (1.f - y); } };
struct AddOp { __device__ float operator()(float a, float b) const { return a + b; } };

template <typename TS, typename TD>
__global__ __launch_bounds__(EW_THREADS) void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, int64_t n) {
  const int64_t nv = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    const Vec8<TS> v = load8<TS>(s + i * 8);
    Vec8<TD> o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.set(k, v.get(k));
    store8<TD>(d + i * 8, o);
  }
  for (int64_t i = (nv << 3) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    d[i] = from_f<TD>(to_f(s[i]));
}

template <typename TS, typename TD>
__global__ void transpose_kernel(const TS* __restrict__ s, TD* __restrict__ d, int64_t rows, int64_t cols) {
  __shared__ float tile[32][33];
  const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int64_t r = r0 + j, c = c0 + threadIdx.x;
    tile[j][threadIdx.x] = (r < rows && c < cols) ? to_f(s[r * cols + c]) : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int64_t c = c0 + j, r = r0 + threadIdx.x;   // output row = c, output col = r
    if (c < cols && r < rows) d[c * rows + r] = from_f<TD>(tile[threadIdx.x][j]);
  }
}

// ------------------------------------------------------------------------------------------------
// y[r,:] = x[r,:] + v[r % period, :]
template <typename T>
__global__ __launch_bounds__(EW_THREADS) void add_rowvec_kernel(const T* __restrict__ x, const float* __restrict__ v,
                                                                 T* __restrict__ y, int64_t rows, int d, int64_t period) {
  const int nch = d >> 3;
  const int64_t total = rows * nch;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / nch;
    const int ch = (int)(i - r * nch);
    const Vec8<T> xv = load8<T>(x + r * d + ch * 8);
    const float* vp = v + (r % period) * d + ch * 8;
    Vec8<T> o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.set(k, xv.get(k) + vp[k]);
    store8<T>(y + r * d + ch * 8, o);
  }
}
// dv[i][j] = sum over the rows r = i (mod period) of dy[r][j].  One workgroup = 64 consecutive (i, j) entries x 4 row groups (the
// temporal embedding's gradient is 12 x 1536 entries over 1536 rows: one thread per entry left 72 workgroups walking 128 rows each)
template <typename T>
__global__ __launch_bounds__(256) void add_rowvec_bwd_kernel(const T* __restrict__ dy, float* __restrict__ dv, int64_t rows, int d, int64_t period) {
  __shared__ float part[4][64];
  const int c = threadIdx.x & 63, g4 = threadIdx.x >> 6;
  const int64_t idx = (int64_t)blockIdx.x * 64 + c;
  float s = 0.f;
  if (idx < period * d) {
    const int64_t i = idx / d, j = idx - i * d;
    for (int64_t r = i + g4 * period; r < rows; r += 4 * period) s += to_f(dy[r * d + j]);
  }
  part[g4][c] = s;
  __syncthreads();
  if (g4 == 0 && idx < period * d) dv[idx] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
}

// ------------------------------------------------------------------------------------------------
// K6 rotary.  One thread = one 8-lane chunk of the rotated prefix of one head of q or k.
template <typename T, bool TRANSPOSE>
__global__ __launch_bounds__(EW_THREADS) void rotary_kernel(T* __restrict__ qkv, int64_t T_rows, int S, int H, int Dh, int R,
                                                             const float* __restrict__ qa, const float* __restrict__ qb,
                                                             const float* __restrict__ ka, const float* __restrict__ kb) {
  const int cpr = (R + 7) >> 3;                     // chunks per head
  const int64_t per_row = (int64_t)2 * H * cpr;
  const int64_t total = T_rows * per_row;
  const int64_t ld = (int64_t)3 * H * Dh;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / per_row;
    int rem = (int)(i - t * per_row);
    const int which = rem / (H * cpr);              // 0 = q, 1 = k
    rem -= which * H * cpr;
    const int h = rem / cpr, c = rem - h * cpr;
    const int pos = (int)(t % S);
    T* p = qkv + t * ld + (int64_t)which * H * Dh + h * Dh + c * 8;
    const float* A = (which ? ka : qa) + (int64_t)pos * R + c * 8;
    const float* B = (which ? kb : qb) + (int64_t)pos * R + c * 8;
    Vec8<T> v = load8<T>(p);
    Vec8<T> o = v;
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      if (c * 8 + k < R) {                          // R is even: a pair is in or out as a whole
        const float t0 = v.get(k), t1 = v.get(k + 1);
        if (!TRANSPOSE) {
          float o0, o1;
          rotary_pair(t0, t1, A[k], A[k + 1], B[k], B[k + 1], o0, o1);
          o.set(k, o0);
          o.set(k + 1, o1);
        } else {
          o.set(k, t0 * A[k] + t1 * B[k + 1]);
          o.set(k + 1, t1 * A[k + 1] - t0 * B[k]);
        }
      }
    }
    store8<T>(p, o);
  }
}

// ------------------------------------------------------------------------------------------------
// K7a patchify: out[(g,ph,pw), (p1,p2,c)] = img[g,c,ph*p+p1,pw*p+p2]
template <typename TI, typename T>
__global__ __launch_bounds__(EW_THREADS) void patchify_kernel(const TI* __restrict__ img, T* __restrict__ out, int64_t G, int C,
                                                               int Hh, int Ww, int p) {
  const int nph = Hh / p, npw = Ww / p;
  const int P = p * p * C;
  const int64_t total = G * nph * npw * (int64_t)P;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / P;
    int e = (int)(i - r * P);
    const int c = e % C; e /= C;
    const int p2 = e % p, p1 = e / p;
    const int pw = (int)(r % npw); r /= npw;
    const int ph = (int)(r % nph);
    const int64_t g = r / nph;
    out[i] = from_f<T>(to_f(img[((g * C + c) * Hh + ph * p + p1) * (int64_t)Ww + pw * p + p2]));
  }
}

// C == 4 fast path: a thread emits 8 consecutive outputs = 2 pixels x 4 channels (16 bytes for bf16) from four
// 8-byte pixel-pair loads; consecutive threads walk p2, so both sides are coalesced per channel plane.
template <typename TI, typename T>
__global__ __launch_bounds__(EW_THREADS) void patchify_c4_kernel(const TI* __restrict__ img, T* __restrict__ out, int64_t G, int Hh,
                                                                  int Ww, int p) {
  const int nph = Hh / p, npw = Ww / p;
  const int P = p * p * 4;
  const int hp = p >> 1;                            // pixel pairs per patch row
  const int64_t total = G * nph * npw * (int64_t)p * hp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / (p * hp);
    int e = (int)(i - r * (p * hp));
    const int pp = e % hp, p1 = e / hp;
    const int pw = (int)(r % npw);
    int64_t r2 = r / npw;
    const int ph = (int)(r2 % nph);
    const int64_t g = r2 / nph;
    const int y = ph * p + p1, x = pw * p + 2 * pp;
    Vec8<T> o;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const TI* src = img + ((g * 4 + c) * Hh + y) * (int64_t)Ww + x;
      o.set(c, to_f(src[0]));
      o.set(4 + c, to_f(src[1]));
    }
    store8s<T>(out + r * P + (p1 * p + 2 * pp) * 4, o);
  }
}

// raw-input variant (the device side of the input pipeline): any storage type in, affine normalisation in fp32, the
// compute dtype out.  One thread = one pixel position of a patch row: it reads the C channel values (coalesced along
// the row inside each channel plane) and writes C consecutive outputs.
__device__ __forceinline__ float raw_to_f(double v) { return (float)v; }
__device__ __forceinline__ float raw_to_f(float v) { return v; }
__device__ __forceinline__ float raw_to_f(bf16 v) { return (float)v; }
__device__ __forceinline__ float raw_to_f(unsigned char v) { return (float)v; }
template <typename TI, typename T>
__global__ __launch_bounds__(EW_THREADS) void patchify_raw_kernel(const TI* __restrict__ img, T* __restrict__ out, int64_t G, int C,
                                                                   int Hh, int Ww, int p, float mean, float inv_std) {
  const int nph = Hh / p, npw = Ww / p;
  const int64_t total = G * (int64_t)Hh * Ww;        // pixel positions, ordered (g, ph, pw, p1, p2) = output order / C
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / (p * p);
    const int e = (int)(i - r * (p * p));
    const int p2 = e % p, p1 = e / p;
    const int pw = (int)(r % npw); r /= npw;
    const int ph = (int)(r % nph);
    const int64_t g = r / nph;
    const int64_t pix = (int64_t)(ph * p + p1) * Ww + pw * p + p2;
    T* o = out + i * C;
    for (int c = 0; c < C; ++c) o[c] = from_f<T>((raw_to_f(img[(g * C + c) * (int64_t)Hh * Ww + pix]) - mean) * inv_std);
  }
}

// ------------------------------------------------------------------------------------------------
// GEGLU (src/meant/timesformer_pytorch.py:60-63): h = [a | g] of width 2w per row -> y = a * gelu(g) of width w;
// backward: dh = [dy * gelu(g) | dy * a * gelu'(g)].  One pass each, 16-byte accesses (w % 8 == 0).
template <typename T>
__global__ __launch_bounds__(EW_THREADS) void geglu_fwd_kernel(const T* __restrict__ h, T* __restrict__ y, int64_t rows, int w) {
  const int cw = w >> 3;
  const int64_t total = rows * cw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cw;
    const int c = (int)(i - r * cw) * 8;
    const Vec8<T> a = load8s<T>(h + r * 2 * w + c), g = load8s<T>(h + r * 2 * w + w + c);
    Vec8<T> o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.set(k, a.get(k) * gelu_erf(g.get(k)));
    store8s<T>(y + r * w + c, o);
  }
}
template <typename T>
__global__ __launch_bounds__(EW_THREADS) void geglu_bwd_kernel(const T* __restrict__ h, const T* __restrict__ dy, T* __restrict__ dh,
                                                                int64_t rows, int w) {
  const int cw = w >> 3;
  const int64_t total = rows * cw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cw;
    const int c = (int)(i - r * cw) * 8;
    const Vec8<T> a = load8s<T>(h + r * 2 * w + c), g = load8s<T>(h + r * 2 * w + w + c), d = load8s<T>(dy + r * w + c);
    Vec8<T> da, dg;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      da.set(k, d.get(k) * gelu_erf(g.get(k)));
      dg.set(k, d.get(k) * a.get(k) * gelu_erf_grad(g.get(k)));
    }
    store8s<T>(dh + r * 2 * w + c, da);
    store8s<T>(dh + r * 2 * w + w + c, dg);
  }
}

// ------------------------------------------------------------------------------------------------
// K8 mean over the sequence axis.  block = (g, 256-column slab): 32 chunks x 8 row groups.
template <typename T, typename TO>
__global__ __launch_bounds__(256) void meanpool_fwd_kernel(const T* __restrict__ x, TO* __restrict__ out, int64_t ld_out,
                                                            int64_t col_off, int S, int d) {
  __shared__ float red[8][32][9];
  const int64_t g = blockIdx.x;
  const int chunk = blockIdx.y * 32 + (threadIdx.x & 31);
  const int rg = threadIdx.x >> 5;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (chunk * 8 < d) {
    const T* base = x + g * (int64_t)S * d + chunk * 8;
    for (int s = rg; s < S; s += 8) {
      const Vec8<T> v = load8<T>(base + (int64_t)s * d);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += v.get(k);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[rg][threadIdx.x & 31][k] = acc[k];
  __syncthreads();
  if (rg == 0 && chunk * 8 < d) {
    Vec8<TO> o;
    const float inv = 1.0f / (float)S;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) s += red[r][threadIdx.x][k];
      o.set(k, s * inv);
    }
    store8<TO>(out + g * ld_out + col_off + chunk * 8, o);
  }
}
template <typename T, typename TO>
__global__ __launch_bounds__(EW_THREADS) void meanpool_bwd_kernel(const TO* __restrict__ dout, int64_t ld_out, int64_t col_off,
                                                                   T* __restrict__ dx, int64_t G, int S, int d) {
  const int nch = d >> 3;
  const int64_t total = G * S * (int64_t)nch;
  const float inv = 1.0f / (float)S;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % nch);
    const int64_t gs = i / nch;
    const int64_t g = gs / S;
    const Vec8<TO> v = load8<TO>(dout + g * ld_out + col_off + ch * 8);
    Vec8<T> o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.set(k, v.get(k) * inv);
    store8<T>(dx + gs * d + ch * 8, o);
  }
}

// ------------------------------------------------------------------------------------------------
// K9 temporal attention core: one wave per (b, h); L <= 64 keys, any Dh.
template <typename T>
__global__ __launch_bounds__(256) void temporal_fwd_kernel(const T* __restrict__ q, const T* __restrict__ kv, T* __restrict__ o,
                                                            float* __restrict__ p, int64_t B, int L, int H, int Dh, float scale) {
  __shared__ float sc[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t bh = (int64_t)blockIdx.x * 4 + wave;
  if (bh >= B * H) return;
  const int64_t b = bh / H;
  const int h = (int)(bh - b * H);
  const int D = H * Dh;
  const T* qp = q + b * D + h * Dh;
  for (int l = 0; l < L; ++l) {
    const T* kp = kv + (b * L + l) * (int64_t)(2 * D) + h * Dh;
    float s = 0.f;
    for (int e = lane; e < Dh; e += 64) s += to_f(qp[e]) * to_f(kp[e]);
    s = wave_sum(s) * scale;
    if (lane == 0) sc[wave][l] = s;
  }
  __builtin_amdgcn_wave_barrier();
  float m = -INFINITY;
  for (int l = 0; l < L; ++l) m = fmaxf(m, sc[wave][l]);
  float sum = 0.f;
  for (int l = 0; l < L; ++l) sum += __expf(sc[wave][l] - m);
  const float inv = 1.f / sum;
  __builtin_amdgcn_wave_barrier();
  if (lane < L) {
    const float w = __expf(sc[wave][lane] - m) * inv;
    p[bh * L + lane] = w;
    sc[wave][lane] = w;
  }
  __builtin_amdgcn_wave_barrier();
  for (int e = lane; e < Dh; e += 64) {
    float acc = 0.f;
    for (int l = 0; l < L; ++l) acc += sc[wave][l] * to_f(kv[(b * L + l) * (int64_t)(2 * D) + D + h * Dh + e]);
    o[b * D + h * Dh + e] = from_f<T>(acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void temporal_bwd_kernel(const T* __restrict__ q, const T* __restrict__ kv,
                                                            const float* __restrict__ p, const T* __restrict__ dout,
                                                            T* __restrict__ dq, T* __restrict__ dkv, int64_t B, int L, int H,
                                                            int Dh, float scale) {
  __shared__ float ds[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t bh = (int64_t)blockIdx.x * 4 + wave;
  if (bh >= B * H) return;
  const int64_t b = bh / H;
  const int h = (int)(bh - b * H);
  const int D = H * Dh;
  const T* dop = dout + b * D + h * Dh;
  float dot = 0.f;                                   // sum_l p_l dp_l
  for (int l = 0; l < L; ++l) {
    const T* vp = kv + (b * L + l) * (int64_t)(2 * D) + D + h * Dh;
    float s = 0.f;
    for (int e = lane; e < Dh; e += 64) s += to_f(dop[e]) * to_f(vp[e]);
    s = wave_sum(s);                                 // dp_l
    if (lane == 0) ds[wave][l] = s;
    dot += p[bh * L + l] * s;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < L) ds[wave][lane] = p[bh * L + lane] * (ds[wave][lane] - dot) * scale;
  __builtin_amdgcn_wave_barrier();
  for (int e = lane; e < Dh; e += 64) {
    const float qe = to_f(q[b * D + h * Dh + e]);
    const float doe = to_f(dop[e]);
    float dqe = 0.f;
    for (int l = 0; l < L; ++l) {
      const int64_t row = (b * L + l) * (int64_t)(2 * D);
      dqe += ds[wave][l] * to_f(kv[row + h * Dh + e]);
      dkv[row + h * Dh + e] = from_f<T>(ds[wave][l] * qe);
      dkv[row + D + h * Dh + e] = from_f<T>(p[bh * L + l] * doe);
    }
    dq[b * D + h * Dh + e] = from_f<T>(dqe);
  }
}

// ------------------------------------------------------------------------------------------------
// embedding gather (one wave per token row) and scatter-add of its gradient (f32 atomics)
template <typename T>
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const float* __restrict__ table, const int64_t* __restrict__ ids,
                                                             T* __restrict__ out, int64_t n, int d, int64_t V) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = d >> 3;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
    int64_t id = ids[r];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const float* src = table + id * d;
    for (int ch = lane; ch < nch; ch += 64) {
      const Vec8<float> v = load8<float>(src + ch * 8);
      Vec8<T> o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o.set(k, v.get(k));
      store8s<T>(out + r * d + ch * 8, o);
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const T* __restrict__ dout, const int64_t* __restrict__ ids,
                                                             float* __restrict__ dtable, int64_t n, int d, int64_t V) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
    int64_t id = ids[r];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    float* dst = dtable + id * d;
    for (int e = lane; e < d; e += 64) atomicAdd(dst + e, to_f(dout[r * d + e]));   // 256 contiguous bytes per wave-instruction
  }
}

// scatter-add of the embedding gradient over ids SORTED by the caller (order[j] = original row of the j-th smallest id).
// A wave walks SEG consecutive sorted entries; a lane owns 8-column chunks `lane` and `lane + 64` of the row (16-byte loads, four
// rows requested before the first is consumed) and sums the rows of a run of equal ids in registers.  A run that lies entirely
// inside the wave's stretch belongs to nobody else: its sum is added with a plain read-modify-write.  Only the (at most two) runs
// that cross the stretch's ends use float atomics -- at 12 tokens per id that is one row in eleven (the atomic rate, 1.3 TB/s of added
// bytes, was a third of this kernel's time when every run went that way, and 2-byte loads most of the rest).
constexpr int EMB_SEG = 256;
template <typename T>
__global__ __launch_bounds__(256) void embedding_bwd_sorted_kernel(const T* __restrict__ dout, const int64_t* __restrict__ sorted_ids,
                                                                    const int64_t* __restrict__ order, float* __restrict__ dtable,
                                                                    int64_t n, int d, int64_t V, int det, int64_t id_lo, int64_t id_hi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t j0 = ((int64_t)blockIdx.x * 4 + wave) * EMB_SEG;
  if (j0 >= n) return;
  int64_t j1 = j0 + EMB_SEG < n ? j0 + EMB_SEG : n;
  // [id_lo, id_hi): only the rows of this id range are produced (meant_embedding_bwd_sorted_range: the table's gradient in row
  // slices, each handed to its collective as soon as it is final).  The ids are sorted: a stretch outside the range leaves at once.
  if (sorted_ids[j1 - 1] < id_lo || sorted_ids[j0] >= id_hi) return;
  if (det) {
    // option "deterministic": no atomics.  A run belongs to the wave in whose stretch it STARTS: that wave follows it to its end
    // (however far), the others skip the part of their stretch that continues an earlier run.  Slow for a hot id; a debugging mode.
    if (j0 > 0) {
      const int64_t prev = sorted_ids[j0 - 1];
      while (j0 < j1 && sorted_ids[j0] == prev) ++j0;
      if (j0 == j1) return;
    }
    const int64_t last = sorted_ids[j1 - 1];
    while (j1 < n && sorted_ids[j1] == last) ++j1;
  }
  const int nch = d >> 3;                              // d % 8 == 0, d <= 1024
  const bool has1 = lane + 64 < nch, has0 = lane < nch;
  float acc0[8], acc1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc0[e] = acc1[e] = 0.f;
  int64_t cur = sorted_ids[j0];
  bool shared = !det && j0 > 0 && sorted_ids[j0 - 1] == cur;   // the first run started in the previous stretch

  auto flush = [&](int64_t id, bool atomic) {
    if (id < 0 || id >= V || id < id_lo || id >= id_hi) return;
    float* row = dtable + id * d;
    if (atomic) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (has0) atomicAdd(row + lane * 8 + e, acc0[e]);
        if (has1) atomicAdd(row + (lane + 64) * 8 + e, acc1[e]);
      }
    } else {
      if (has0) {
        Vec8<float> v = load8<float>(row + lane * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v.set(e, v.get(e) + acc0[e]);
        store8<float>(row + lane * 8, v);
      }
      if (has1) {
        Vec8<float> v = load8<float>(row + (lane + 64) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v.set(e, v.get(e) + acc1[e]);
        store8<float>(row + (lane + 64) * 8, v);
      }
    }
  };

  for (int64_t jb = j0; jb < j1; jb += 64) {
    const int64_t jl = jb + lane < j1 ? jb + lane : j1 - 1;
    const long long my_id = sorted_ids[jl], my_ord = order[jl];
    const int cnt = (int)(j1 - jb < 64 ? j1 - jb : 64);
    for (int t0 = 0; t0 < cnt; t0 += 4) {
      Vec8<T> r0[4], r1[4];
      long long ids4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u < cnt ? t0 + u : cnt - 1;
        ids4[u] = __shfl(my_id, t, 64);
        const long long ord = __shfl(my_ord, t, 64);
        const T* src = dout + ord * d;
        if (has0) r0[u] = load8s<T>(src + lane * 8);
        if (has1) r1[u] = load8s<T>(src + (lane + 64) * 8);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (t0 + u >= cnt) break;
        const int64_t id = ids4[u];
        if (id != cur) {
          flush(cur, shared);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc0[e] = acc1[e] = 0.f;
          cur = id;
          shared = false;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (has0) acc0[e] += r0[u].get(e);
          if (has1) acc1[e] += r1[u].get(e);
        }
      }
    }
  }
  flush(cur, !det && (shared || (j1 < n && sorted_ids[j1] == cur)));   // ... or continues into the next stretch
}

}  // namespace

#define EW_REQ(c, ...) MEANT_REQUIRE(c, MEANT_ERR_ARG, __VA_ARGS__)

template <typename F>
static int launch_ew2(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream, F f, const char* name) {
  EW_REQ(a && b && y && n >= 0, "%s: bad argument", name);
  EW_REQ(meant_aligned16(a) && meant_aligned16(b) && meant_aligned16(y), "%s: 16-byte alignment", name);
  if (n == 0) return MEANT_OK;
  DISPATCH_DTYPE(dtype, T,
                 hipLaunchKernelGGL((ew2_kernel<T, F>), dim3(ew_blocks(ceil_div(n, 8))), dim3(EW_THREADS), 0, (hipStream_t)stream,
                                    (const T*)a, (const T*)b, (T*)y, n, f));
  MEANT_LAUNCH_CHECK(name);
  return MEANT_OK;
}

// ---- RMSNorm gain folded into the consumer Linear's weight (meant_linear_fwd_rowscale): W'[n, k] = W[n, k] g[k] -----------
namespace {
__global__ __launch_bounds__(256) void colscale_kernel(const float* __restrict__ w, const float* __restrict__ g, float* __restrict__ out,
                                                        int64_t total, int K) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) out[i] = w[i] * g[i % K];
}
// dW[n, k] += dWp[n, k] g[k];  dg[k] += sum_n dWp[n, k] W[n, k].  One block per 64 columns, 4 row phases, ordered combine.
__global__ __launch_bounds__(256) void colscale_bwd_kernel(const float* __restrict__ dwp, const float* __restrict__ w,
                                                            const float* __restrict__ g, float* __restrict__ dw, float* __restrict__ dg,
                                                            int N, int K) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + c;
  float s = 0.f;
  if (k < K) {
    const float gk = g[k];
    for (int n = ph; n < N; n += 4) {
      const float d = dwp[(int64_t)n * K + k];
      s += d * w[(int64_t)n * K + k];
      dw[(int64_t)n * K + k] += d * gk;
    }
  }
  red[ph][c] = s;
  __syncthreads();
  if (ph == 0 && k < K) dg[k] += (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}
}  // namespace

extern "C" int meant_colscale(const float* w, const float* g, float* out, int64_t N, int64_t K, void* stream) {
  EW_REQ(w && g && out && N > 0 && K > 0 && N * K < (1LL << 40), "colscale: bad argument");
  const int64_t total = N * K;
  const dim3 grid((unsigned)(ceil_div(total, 256) < 4096 ? ceil_div(total, 256) : 4096));
  hipLaunchKernelGGL(colscale_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, g, out, total, (int)K);
  MEANT_LAUNCH_CHECK("colscale");
  return MEANT_OK;
}
extern "C" int meant_colscale_bwd(const float* dwp, const float* w, const float* g, float* dw, float* dg, int64_t N, int64_t K, void* stream) {
  EW_REQ(dwp && w && g && dw && dg && N > 0 && K > 0 && N < (1LL << 30) && K < (1LL << 30), "colscale_bwd: bad argument");
  hipLaunchKernelGGL(colscale_bwd_kernel, dim3((unsigned)ceil_div(K, 64)), dim3(256), 0, (hipStream_t)stream, dwp, w, g, dw, dg, (int)N, (int)K);
  MEANT_LAUNCH_CHECK("colscale_bwd");
  return MEANT_OK;
}

extern "C" int meant_gelu_bwd(const void* dy, const void* pre, void* dx, int64_t n, int dtype, void* stream) {
  return launch_ew2(dy, pre, dx, n, dtype, stream, GeluBwdOp(), "gelu_bwd");
}
extern "C" int meant_sigmoid_bwd(const void* dy, const void* y, void* dx, int64_t n, int dtype, void* stream) {
  return launch_ew2(dy, y, dx, n, dtype, stream, SigmoidBwdOp(), "sigmoid_bwd");
}
extern "C" int meant_add(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream) {
  return launch_ew2(a, b, y, n, dtype, stream, AddOp(), "add");
}

extern "C" int meant_geglu_fwd(const void* h, void* y, int64_t rows, int64_t w, int dtype, void* stream) {
  EW_REQ(h && y && rows >= 0 && w > 0 && w % 8 == 0 && w < (1LL << 30), "geglu_fwd: bad argument (w must be a multiple of 8)");
  EW_REQ(meant_aligned16(h) && meant_aligned16(y), "geglu_fwd: 16-byte alignment");
  if (rows == 0) return MEANT_OK;
  const int64_t total = rows * (w >> 3);
  const dim3 grid((unsigned)(ceil_div(total, EW_THREADS) < 65535 * 8 ? ceil_div(total, EW_THREADS) : 65535 * 8)), block(EW_THREADS);
  DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(geglu_fwd_kernel<T>, grid, block, 0, (hipStream_t)stream, (const T*)h, (T*)y, rows, (int)w));
  MEANT_LAUNCH_CHECK("geglu_fwd");
  return MEANT_OK;
}
extern "C" int meant_geglu_bwd(const void* h, const void* dy, void* dh, int64_t rows, int64_t w, int dtype, void* stream) {
  EW_REQ(h && dy && dh && rows >= 0 && w > 0 && w % 8 == 0 && w < (1LL << 30), "geglu_bwd: bad argument (w must be a multiple of 8)");
  EW_REQ(meant_aligned16(h) && meant_aligned16(dy) && meant_aligned16(dh), "geglu_bwd: 16-byte alignment");
  if (rows == 0) return MEANT_OK;
  const int64_t total = rows * (w >> 3);
  const dim3 grid((unsigned)(ceil_div(total, EW_THREADS) < 65535 * 8 ? ceil_div(total, EW_THREADS) : 65535 * 8)), block(EW_THREADS);
  DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(geglu_bwd_kernel<T>, grid, block, 0, (hipStream_t)stream, (const T*)h, (const T*)dy, (T*)dh, rows, (int)w));
  MEANT_LAUNCH_CHECK("geglu_bwd");
  return MEANT_OK;
}

extern "C" int meant_cast(const void* src, int dtype_src, void* dst, int dtype_dst, int64_t n, void* stream) {
  EW_REQ(src && dst && n >= 0, "cast: bad argument");
  EW_REQ(meant_aligned16(src) && meant_aligned16(dst), "cast: 16-byte alignment");
  if (n == 0) return MEANT_OK;
  const dim3 grid(ew_blocks(ceil_div(n, 8))), block(EW_THREADS);
  hipStream_t st = (hipStream_t)stream;
  if (dtype_src == MEANT_F32 && dtype_dst == MEANT_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16>), grid, block, 0, st, (const float*)src, (bf16*)dst, n);
  else if (dtype_src == MEANT_BF16 && dtype_dst == MEANT_F32) hipLaunchKernelGGL((cast_kernel<bf16, float>), grid, block, 0, st, (const bf16*)src, (float*)dst, n);
  else if (dtype_src == MEANT_F32 && dtype_dst == MEANT_F32) hipLaunchKernelGGL((cast_kernel<float, float>), grid, block, 0, st, (const float*)src, (float*)dst, n);
  else if (dtype_src == MEANT_BF16 && dtype_dst == MEANT_BF16) hipLaunchKernelGGL((cast_kernel<bf16, bf16>), grid, block, 0, st, (const bf16*)src, (bf16*)dst, n);
  else { meant_set_error("cast: unknown dtype"); return MEANT_ERR_ARG; }
  MEANT_LAUNCH_CHECK("cast");
  return MEANT_OK;
}

extern "C" int meant_transpose2d(const void* src, int dtype_src, void* dst, int dtype_dst, int64_t rows, int64_t cols, void* stream) {
  EW_REQ(src && dst && rows > 0 && cols > 0, "transpose2d: bad argument");
  const dim3 grid((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows, 32)), block(32, 8);
  hipStream_t st = (hipStream_t)stream;
  if (dtype_src == MEANT_F32 && dtype_dst == MEANT_BF16) hipLaunchKernelGGL((transpose_kernel<float, bf16>), grid, block, 0, st, (const float*)src, (bf16*)dst, rows, cols);
  else if (dtype_src == MEANT_F32 && dtype_dst == MEANT_F32) hipLaunchKernelGGL((transpose_kernel<float, float>), grid, block, 0, st, (const float*)src, (float*)dst, rows, cols);
  else if (dtype_src == MEANT_BF16 && dtype_dst == MEANT_BF16) hipLaunchKernelGGL((transpose_kernel<bf16, bf16>), grid, block, 0, st, (const bf16*)src, (bf16*)dst, rows, cols);
  else if (dtype_src == MEANT_BF16 && dtype_dst == MEANT_F32) hipLaunchKernelGGL((transpose_kernel<bf16, float>), grid, block, 0, st, (const bf16*)src, (float*)dst, rows, cols);
  else { meant_set_error("transpose2d: unknown dtype"); return MEANT_ERR_ARG; }
  MEANT_LAUNCH_CHECK("transpose2d");
  return MEANT_OK;
}

extern "C" int meant_add_rowvec(const void* x, const float* v, void* y, int64_t rows, int64_t d, int64_t period, int dtype, void* stream) {
  EW_REQ(x && v && y && rows > 0 && d > 0 && d % 8 == 0 && period > 0, "add_rowvec: bad argument");
  DISPATCH_DTYPE(dtype, T,
                 hipLaunchKernelGGL(add_rowvec_kernel<T>, dim3(ew_blocks(rows * (d / 8))), dim3(EW_THREADS), 0, (hipStream_t)stream,
                                    (const T*)x, v, (T*)y, rows, (int)d, period));
  MEANT_LAUNCH_CHECK("add_rowvec");
  return MEANT_OK;
}
extern "C" int meant_add_rowvec_bwd(const void* dy, float* dv, int64_t rows, int64_t d, int64_t period, int dtype, void* stream) {
  EW_REQ(dy && dv && rows > 0 && d > 0 && period > 0, "add_rowvec_bwd: bad argument");
  DISPATCH_DTYPE(dtype, T,
                 hipLaunchKernelGGL(add_rowvec_bwd_kernel<T>, dim3((unsigned)ceil_div(period * d, 64)), dim3(256), 0, (hipStream_t)stream,
                                    (const T*)dy, dv, rows, (int)d, period));
  MEANT_LAUNCH_CHECK("add_rowvec_bwd");
  return MEANT_OK;
}

extern "C" int meant_rotary_qk(void* qkv, int64_t T_rows, int64_t S, int H, int Dh, int R, const float* qa, const float* qb,
                               const float* ka, const float* kb, int transpose, int dtype, void* stream) {
  EW_REQ(qkv && qa && qb && ka && kb, "rotary_qk: null pointer");
  EW_REQ(T_rows > 0 && S > 0 && T_rows % S == 0, "rotary_qk: T=%lld must be a multiple of S=%lld", (long long)T_rows, (long long)S);
  MEANT_REQUIRE(H > 0 && Dh % 8 == 0 && R % 2 == 0 && R <= Dh && R % 8 == 0, MEANT_ERR_UNSUPPORTED,
                "rotary_qk: need Dh%%8==0 and rot_dim%%8==0 (Dh=%d R=%d)", Dh, R);
  EW_REQ(meant_aligned16(qkv), "rotary_qk: 16-byte alignment");
  if (R == 0) return MEANT_OK;
  const int64_t items = T_rows * 2 * H * ((R + 7) / 8);
  DISPATCH_DTYPE(dtype, T, {
    if (transpose) hipLaunchKernelGGL((rotary_kernel<T, true>), dim3(ew_blocks(items)), dim3(EW_THREADS), 0, (hipStream_t)stream, (T*)qkv, T_rows, (int)S, H, Dh, R, qa, qb, ka, kb);
    else hipLaunchKernelGGL((rotary_kernel<T, false>), dim3(ew_blocks(items)), dim3(EW_THREADS), 0, (hipStream_t)stream, (T*)qkv, T_rows, (int)S, H, Dh, R, qa, qb, ka, kb);
  });
  MEANT_LAUNCH_CHECK("rotary_qk");
  return MEANT_OK;
}

extern "C" int meant_patchify_raw(const void* images, int raw_dtype, float mean, float inv_std, void* patches, int64_t G, int C,
                                  int Hh, int Ww, int p, int dtype, void* stream) {
  EW_REQ(images && patches && G > 0 && C > 0 && p > 0 && Hh % p == 0 && Ww % p == 0, "patchify_raw: bad argument");
  EW_REQ(dtype == MEANT_F32 || dtype == MEANT_BF16, "patchify_raw: unknown output dtype");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = G * (int64_t)Hh * Ww;
  const dim3 grid((unsigned)(ceil_div(total, EW_THREADS) < 65535 * 16 ? ceil_div(total, EW_THREADS) : 65535 * 16)), block(EW_THREADS);
#define RAW_LAUNCH(TI)                                                                                                          \
  do {                                                                                                                          \
    if (dtype == MEANT_F32) hipLaunchKernelGGL((patchify_raw_kernel<TI, float>), grid, block, 0, st, (const TI*)images, (float*)patches, G, C, Hh, Ww, p, mean, inv_std); \
    else hipLaunchKernelGGL((patchify_raw_kernel<TI, bf16>), grid, block, 0, st, (const TI*)images, (bf16*)patches, G, C, Hh, Ww, p, mean, inv_std); \
  } while (0)
  switch (raw_dtype) {
    case MEANT_RAW_F32: RAW_LAUNCH(float); break;
    case MEANT_RAW_BF16: RAW_LAUNCH(bf16); break;
    case MEANT_RAW_F64: RAW_LAUNCH(double); break;
    case MEANT_RAW_U8: RAW_LAUNCH(unsigned char); break;
    default: meant_set_error("patchify_raw: unknown raw dtype %d", raw_dtype); return MEANT_ERR_ARG;
  }
#undef RAW_LAUNCH
  MEANT_LAUNCH_CHECK("patchify_raw");
  return MEANT_OK;
}

extern "C" int meant_patchify(const void* images, int images_dtype, void* patches, int64_t G, int C, int Hh, int Ww, int p, int dtype, void* stream) {
  EW_REQ(images && patches && G > 0 && C > 0 && p > 0 && Hh % p == 0 && Ww % p == 0, "patchify: bad argument");
  const int64_t total = G * C * (int64_t)Hh * Ww;
  hipStream_t st = (hipStream_t)stream;
  if (C == 4 && p % 2 == 0 && Ww % 2 == 0 && meant_aligned16(patches)) {
    const dim3 grid4(ew_blocks(total / 8)), block4(EW_THREADS);
    bool ok = true;
    if (images_dtype == MEANT_F32 && dtype == MEANT_F32) hipLaunchKernelGGL((patchify_c4_kernel<float, float>), grid4, block4, 0, st, (const float*)images, (float*)patches, G, Hh, Ww, p);
    else if (images_dtype == MEANT_F32 && dtype == MEANT_BF16) hipLaunchKernelGGL((patchify_c4_kernel<float, bf16>), grid4, block4, 0, st, (const float*)images, (bf16*)patches, G, Hh, Ww, p);
    else if (images_dtype == MEANT_BF16 && dtype == MEANT_BF16) hipLaunchKernelGGL((patchify_c4_kernel<bf16, bf16>), grid4, block4, 0, st, (const bf16*)images, (bf16*)patches, G, Hh, Ww, p);
    else ok = false;
    if (ok) { MEANT_LAUNCH_CHECK("patchify_c4"); return MEANT_OK; }
  }
  const dim3 grid(ew_blocks(total)), block(EW_THREADS);
  if (images_dtype == MEANT_F32 && dtype == MEANT_F32) hipLaunchKernelGGL((patchify_kernel<float, float>), grid, block, 0, st, (const float*)images, (float*)patches, G, C, Hh, Ww, p);
  else if (images_dtype == MEANT_F32 && dtype == MEANT_BF16) hipLaunchKernelGGL((patchify_kernel<float, bf16>), grid, block, 0, st, (const float*)images, (bf16*)patches, G, C, Hh, Ww, p);
  else if (images_dtype == MEANT_BF16 && dtype == MEANT_BF16) hipLaunchKernelGGL((patchify_kernel<bf16, bf16>), grid, block, 0, st, (const bf16*)images, (bf16*)patches, G, C, Hh, Ww, p);
  else if (images_dtype == MEANT_BF16 && dtype == MEANT_F32) hipLaunchKernelGGL((patchify_kernel<bf16, float>), grid, block, 0, st, (const bf16*)images, (float*)patches, G, C, Hh, Ww, p);
  else { meant_set_error("patchify: unknown dtype"); return MEANT_ERR_ARG; }
  MEANT_LAUNCH_CHECK("patchify");
  return MEANT_OK;
}

extern "C" int meant_meanpool_fwd(const void* x, void* out, int64_t ld_out, int64_t col_off, int64_t G, int64_t S, int64_t d, int dtype, int out_dtype, void* stream) {
  EW_REQ(x && out && G > 0 && S > 0 && d > 0 && d % 8 == 0 && ld_out % 8 == 0 && col_off % 8 == 0 && col_off + d <= ld_out, "meanpool_fwd: bad argument");
  EW_REQ(G < 2147483647LL, "meanpool_fwd: too many groups");
  const dim3 grid((unsigned)G, (unsigned)ceil_div(d, 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MEANT_F32 && out_dtype == MEANT_F32) hipLaunchKernelGGL((meanpool_fwd_kernel<float, float>), grid, block, 0, st, (const float*)x, (float*)out, ld_out, col_off, (int)S, (int)d);
  else if (dtype == MEANT_BF16 && out_dtype == MEANT_F32) hipLaunchKernelGGL((meanpool_fwd_kernel<bf16, float>), grid, block, 0, st, (const bf16*)x, (float*)out, ld_out, col_off, (int)S, (int)d);
  else if (dtype == MEANT_BF16 && out_dtype == MEANT_BF16) hipLaunchKernelGGL((meanpool_fwd_kernel<bf16, bf16>), grid, block, 0, st, (const bf16*)x, (bf16*)out, ld_out, col_off, (int)S, (int)d);
  else { meant_set_error("meanpool_fwd: unsupported dtype combination"); return MEANT_ERR_UNSUPPORTED; }
  MEANT_LAUNCH_CHECK("meanpool_fwd");
  return MEANT_OK;
}
extern "C" int meant_meanpool_bwd(const void* dout, int64_t ld_out, int64_t col_off, void* dx, int64_t G, int64_t S, int64_t d, int dtype, int out_dtype, void* stream) {
  EW_REQ(dout && dx && G > 0 && S > 0 && d > 0 && d % 8 == 0 && ld_out % 8 == 0 && col_off % 8 == 0, "meanpool_bwd: bad argument");
  const dim3 grid(ew_blocks(G * S * (d / 8))), block(EW_THREADS);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MEANT_F32 && out_dtype == MEANT_F32) hipLaunchKernelGGL((meanpool_bwd_kernel<float, float>), grid, block, 0, st, (const float*)dout, ld_out, col_off, (float*)dx, G, (int)S, (int)d);
  else if (dtype == MEANT_BF16 && out_dtype == MEANT_F32) hipLaunchKernelGGL((meanpool_bwd_kernel<bf16, float>), grid, block, 0, st, (const float*)dout, ld_out, col_off, (bf16*)dx, G, (int)S, (int)d);
  else if (dtype == MEANT_BF16 && out_dtype == MEANT_BF16) hipLaunchKernelGGL((meanpool_bwd_kernel<bf16, bf16>), grid, block, 0, st, (const bf16*)dout, ld_out, col_off, (bf16*)dx, G, (int)S, (int)d);
  else { meant_set_error("meanpool_bwd: unsupported dtype combination"); return MEANT_ERR_UNSUPPORTED; }
  MEANT_LAUNCH_CHECK("meanpool_bwd");
  return MEANT_OK;
}

extern "C" int meant_temporal_attn_fwd(const void* q, const void* kv, void* o, float* p, int64_t B, int L, int H, int Dh, float scale, int dtype, void* stream) {
  EW_REQ(q && kv && o && p && B > 0 && H > 0 && Dh > 0, "temporal_attn_fwd: bad argument");
  MEANT_REQUIRE(L > 0 && L <= 64, MEANT_ERR_UNSUPPORTED, "temporal_attn_fwd: lag %d not in 1..64", L);
  DISPATCH_DTYPE(dtype, T,
                 hipLaunchKernelGGL(temporal_fwd_kernel<T>, dim3((unsigned)ceil_div(B * H, 4)), dim3(256), 0, (hipStream_t)stream,
                                    (const T*)q, (const T*)kv, (T*)o, p, B, L, H, Dh, scale));
  MEANT_LAUNCH_CHECK("temporal_attn_fwd");
  return MEANT_OK;
}
extern "C" int meant_temporal_attn_bwd(const void* q, const void* kv, const float* p, const void* do_, void* dq, void* dkv, int64_t B, int L, int H, int Dh, float scale, int dtype, void* stream) {
  EW_REQ(q && kv && p && do_ && dq && dkv && B > 0 && H > 0 && Dh > 0, "temporal_attn_bwd: bad argument");
  MEANT_REQUIRE(L > 0 && L <= 64, MEANT_ERR_UNSUPPORTED, "temporal_attn_bwd: lag %d not in 1..64", L);
  DISPATCH_DTYPE(dtype, T,
                 hipLaunchKernelGGL(temporal_bwd_kernel<T>, dim3((unsigned)ceil_div(B * H, 4)), dim3(256), 0, (hipStream_t)stream,
                                    (const T*)q, (const T*)kv, p, (const T*)do_, (T*)dq, (T*)dkv, B, L, H, Dh, scale));
  MEANT_LAUNCH_CHECK("temporal_attn_bwd");
  return MEANT_OK;
}

extern "C" int meant_embedding_fwd(const float* table, const int64_t* ids, void* out, int64_t n, int64_t d, int64_t V, int dtype, void* stream) {
  EW_REQ(table && ids && out && n > 0 && d > 0 && d % 8 == 0 && V > 0, "embedding_fwd: bad argument");
  int64_t nb = ceil_div(n, 4); if (nb > 8192) nb = 8192;
  DISPATCH_DTYPE(dtype, T,
                 hipLaunchKernelGGL(embedding_fwd_kernel<T>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, table, ids, (T*)out, n, (int)d, V));
  MEANT_LAUNCH_CHECK("embedding_fwd");
  return MEANT_OK;
}
extern "C" int meant_embedding_bwd_sorted(const void* dout, const int64_t* sorted_ids, const int64_t* order, float* dtable, int64_t n,
                                          int64_t d, int64_t V, int dtype, void* stream) {
  EW_REQ(dout && sorted_ids && order && dtable && n > 0 && d > 0 && V > 0, "embedding_bwd_sorted: bad argument");
  MEANT_REQUIRE(d <= 1024 && d % 8 == 0, MEANT_ERR_UNSUPPORTED, "embedding_bwd_sorted: d=%lld must be a multiple of 8 and <= 1024", (long long)d);
  EW_REQ(meant_aligned16(dout) && meant_aligned16(dtable), "embedding_bwd_sorted: 16-byte alignment");
  const int64_t nb = ceil_div(n, 4 * EMB_SEG);
  DISPATCH_DTYPE(dtype, T,
                 hipLaunchKernelGGL(embedding_bwd_sorted_kernel<T>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const T*)dout,
                                    sorted_ids, order, dtable, n, (int)d, V, meant_opt(MEANT_OPT_DETERMINISTIC) != 0, (int64_t)0, V));
  MEANT_LAUNCH_CHECK("embedding_bwd_sorted");
  return MEANT_OK;
}

extern "C" int meant_embedding_bwd_sorted_range(const void* dout, const int64_t* sorted_ids, const int64_t* order, float* dtable, int64_t n,
                                                int64_t d, int64_t V, int64_t id_lo, int64_t id_hi, int dtype, void* stream) {
  EW_REQ(dout && sorted_ids && order && dtable && n > 0 && d > 0 && V > 0 && 0 <= id_lo && id_lo <= id_hi && id_hi <= V,
         "embedding_bwd_sorted_range: bad argument");
  MEANT_REQUIRE(d <= 1024 && d % 8 == 0, MEANT_ERR_UNSUPPORTED, "embedding_bwd_sorted_range: d=%lld must be a multiple of 8 and <= 1024", (long long)d);
  EW_REQ(meant_aligned16(dout) && meant_aligned16(dtable), "embedding_bwd_sorted_range: 16-byte alignment");
  if (id_lo == id_hi) return MEANT_OK;
  const int64_t nb = ceil_div(n, 4 * EMB_SEG);
  DISPATCH_DTYPE(dtype, T,
                 hipLaunchKernelGGL(embedding_bwd_sorted_kernel<T>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const T*)dout,
                                    sorted_ids, order, dtable, n, (int)d, V, meant_opt(MEANT_OPT_DETERMINISTIC) != 0, id_lo, id_hi));
  MEANT_LAUNCH_CHECK("embedding_bwd_sorted_range");
  return MEANT_OK;
}

extern "C" int meant_embedding_bwd(const void* dout, const int64_t* ids, float* dtable, int64_t n, int64_t d, int64_t V, int dtype, void* stream) {
  EW_REQ(dout && ids && dtable && n > 0 && d > 0 && V > 0, "embedding_bwd: bad argument");
  int64_t nb = ceil_div(n, 4); if (nb > 8192) nb = 8192;
  DISPATCH_DTYPE(dtype, T,
                 hipLaunchKernelGGL(embedding_bwd_kernel<T>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const T*)dout, ids, dtable, n, (int)d, V));
  MEANT_LAUNCH_CHECK("embedding_bwd");
  return MEANT_OK;
}
