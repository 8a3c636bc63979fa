// Train-step tail of the MEANT path (in_loop_train.py:232-238, 547-548): cross-entropy on the model's
// probabilities, global-norm gradient clipping and AdamW, as fused kernels over the FLAT fp32 buckets
// that the data-parallel reducer already keeps (meant_amd/parallel.py): one launch per bucket instead
// of PyTorch's per-tensor foreach chains, the clip coefficient applied on the fly from a device scalar
// (no host synchronisation anywhere in the step).  HBM-bound: 4 reads + 3 writes of 4 bytes per parameter.
#include "common.h"

namespace {

// out[0] += sum(x^2): block-level tree, one float atomic per block
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  const int64_t nv = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  for (int64_t i = (nv << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += x[i] * x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// torch.optim.AdamW semantics (decoupled weight decay, bias-corrected moments):
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// with g pre-multiplied by the clip coefficient min(1, max_norm / (sqrt(sumsq) + 1e-6)) (clip_grad_norm_).
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                     float wd, float bc1, float bc2, const float* __restrict__ sumsq,
                                                     float max_norm, float grad_scale) {
  float clip = grad_scale;
  if (sumsq && max_norm > 0.f) {
    const float norm = sqrtf(*sumsq) * fabsf(grad_scale);
    const float c = max_norm / (norm + 1e-6f);
    clip *= c < 1.f ? c : 1.f;
  }
  const float step = lr / bc1, isb2 = rsqrtf(bc2), decay = 1.f - lr * wd;
  const int64_t nv = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 pv = *reinterpret_cast<f32x4*>(p + i * 4), mv = *reinterpret_cast<f32x4*>(m + i * 4), vv = *reinterpret_cast<f32x4*>(v + i * 4);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = gv[k] * clip;
      mv[k] = b1 * mv[k] + (1.f - b1) * gk;
      vv[k] = b2 * vv[k] + (1.f - b2) * gk * gk;
      pv[k] = pv[k] * decay - step * mv[k] / (sqrtf(vv[k]) * isb2 + eps);
    }
    *reinterpret_cast<f32x4*>(p + i * 4) = pv;
    *reinterpret_cast<f32x4*>(m + i * 4) = mv;
    *reinterpret_cast<f32x4*>(v + i * 4) = vv;
  }
  for (int64_t i = (nv << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gk = g[i] * clip;
    const float mk = b1 * m[i] + (1.f - b1) * gk, vk = b2 * v[i] + (1.f - b2) * gk * gk;
    m[i] = mk; v[i] = vk;
    p[i] = p[i] * decay - step * mk / (sqrtf(vk) * isb2 + eps);
  }
}

// CrossEntropyLoss(reduction='mean') applied to probabilities (the reference feeds the Sigmoid output to it,
// in_loop_train.py:232): loss = mean_b( logsumexp(x_b) - x_b[t_b] ), dx = (softmax(x_b) - onehot) / B.  One thread per row.
__global__ void ce_probs_kernel(const float* __restrict__ x, const int64_t* __restrict__ target, float* __restrict__ loss,
                                float* __restrict__ dx, int64_t B, int C) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float l = 0.f;
  if (b < B) {
    const float* r = x + b * C;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, r[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(r[c] - mx);
    const float lse = mx + logf(s);
    const int64_t t = target[b];
    if (t < 0 || t >= C) {
      // a class index outside [0, C) (torch raises here; a kernel cannot): poison the loss and this row's gradient so that the
      // caller sees NaN instead of an out-of-bounds read and a silently wrong step
      l = NAN;
      if (dx) for (int c = 0; c < C; ++c) dx[b * C + c] = NAN;
    } else {
      l = (lse - r[t]) / (float)B;
      if (dx) for (int c = 0; c < C; ++c) dx[b * C + c] = (expf(r[c] - lse) - (c == t ? 1.f : 0.f)) / (float)B;
    }
  }
  l = wave_sum(l);
  if ((threadIdx.x & 63) == 0 && !(l == 0.f)) atomicAdd(loss, l);
}

// ------------------------------------------------------------------------------------------------
// Softmax cross-entropy over a large vocabulary (the MLM pretrainer's loss, pretrain_mlm.py:160,178:
// nn.CrossEntropyLoss() on logits [T, V = 64001] with ignore_index = -100).  One workgroup per row, the row is read
// ONCE in the forward (online max / sum-exp in the log2 domain) and once more in the backward, which writes
// d logits = (softmax - onehot) * g in the activation dtype (zeros for ignored rows and for the padding columns
// V..ld-1 that the padded vocabulary GEMM carries).  HBM-bound: 2 B per logit forward, 4 B per logit backward.
template <typename T>
__global__ __launch_bounds__(256) void softmax_ce_fwd_kernel(const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ target,
                                                             int V, int64_t ignore_index, float* __restrict__ row_loss,
                                                             float* __restrict__ lse_out) {
  __shared__ float red_m[4], red_s[4];
  const int64_t row = blockIdx.x;
  const T* x = logits + row * ld;
  const int tid = threadIdx.x;
  float m = -INFINITY, s = 0.f;                        // running max (natural units) and sum of exp(x - m)
  const int nchunk = V >> 3;
  for (int c = tid; c < nchunk; c += 256) {
    const Vec8<T> v = load8<T>(x + c * 8);
    float cm = v.get(0);
#pragma unroll
    for (int i = 1; i < 8; ++i) cm = fmaxf(cm, v.get(i));
    const float mn = fmaxf(m, cm);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += __expf(v.get(i) - mn);
    s = s * __expf(m - mn) + acc;
    m = mn;
  }
  for (int j = (nchunk << 3) + tid; j < V; j += 256) {   // ragged tail (V = 64001)
    const float xv = to_f(x[j]);
    const float mn = fmaxf(m, xv);
    s = s * __expf(m - mn) + __expf(xv - mn);
    m = mn;
  }
  // combine (m, s) pairs across the wave, then across the 4 waves
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float mo = __shfl_xor(m, o, 64), so = __shfl_xor(s, o, 64);
    const float mn = fmaxf(m, mo);
    s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + (mo == -INFINITY ? 0.f : so * __expf(mo - mn));
    m = mn;
  }
  if ((tid & 63) == 0) { red_m[tid >> 6] = m; red_s[tid >> 6] = s; }
  __syncthreads();
  if (tid == 0) {
    float M = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3])), Ssum = 0.f;
    for (int w = 0; w < 4; ++w) Ssum += red_m[w] == -INFINITY ? 0.f : red_s[w] * __expf(red_m[w] - M);
    const float lse = M + __logf(Ssum);
    lse_out[row] = lse;
    const int64_t t = target[row];
    row_loss[row] = (t == ignore_index || t < 0 || t >= V) ? 0.f : lse - to_f(x[t]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_ce_bwd_kernel(const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ target,
                                                             const float* __restrict__ lse, int V, int64_t ignore_index,
                                                             const float* __restrict__ gscale, T* __restrict__ dlogits) {
  const int64_t row = blockIdx.x;
  const T* x = logits + row * ld;
  T* dx = dlogits + row * ld;
  const int tid = threadIdx.x;
  const int64_t t = target[row];
  const bool ignored = (t == ignore_index || t < 0 || t >= V);
  const float g = ignored ? 0.f : gscale[0];
  const float l = lse[row];
  const int nchunk_ld = (int)(ld >> 3);                // ld % 8 == 0 (checked by the launcher): covers the padding too
  for (int c = tid; c < nchunk_ld; c += 256) {
    const Vec8<T> v = load8<T>(x + c * 8);
    Vec8<T> o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = c * 8 + i;
      float d = (j < V && !ignored) ? __expf(v.get(i) - l) : 0.f;
      if (j == t) d -= 1.f;
      o.set(i, d * g);
    }
    store8<T>(dx + c * 8, o);
  }
}

}  // namespace

extern "C" int meant_sumsq_f32(const float* x, int64_t n, float* out_accum, void* stream) {
  MEANT_REQUIRE(x && out_accum && n >= 0 && meant_aligned16(x), MEANT_ERR_ARG, "sumsq_f32: bad argument");
  if (n == 0) return MEANT_OK;
  int64_t nb = ceil_div(n, 256 * 16);
  if (nb > 2048) nb = 2048;
  // option `deterministic`: one workgroup, so the sum has ONE order (the per-block float atomics otherwise land in arrival order and
  // the clip coefficient differs in its last bit from run to run -- enough for a resumed run to leave the original's path: a 1-ulp
  // difference in an fp32 weight flips the bf16 rounding of a few of them in the next step).  ~1 ms per 100 MB of gradients.
  if (meant_opt(MEANT_OPT_DETERMINISTIC) != 0) nb = 1;
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, n, out_accum);
  MEANT_LAUNCH_CHECK("sumsq_f32");
  return MEANT_OK;
}

extern "C" int meant_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int64_t step, const float* sumsq, float max_norm, float grad_scale, void* stream) {
  MEANT_REQUIRE(p && g && m && v && n >= 0 && step >= 1, MEANT_ERR_ARG, "adamw_f32: bad argument");
  MEANT_REQUIRE(meant_aligned16(p) && meant_aligned16(g) && meant_aligned16(m) && meant_aligned16(v), MEANT_ERR_ARG, "adamw_f32: 16-byte alignment");
  if (n == 0) return MEANT_OK;
  // bias corrections in double (1 - 0.999^t loses ~6e-5 relative in float at small t)
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step)), bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  int64_t nb = ceil_div(n, 256 * 4);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                     bc1, bc2, sumsq, max_norm, grad_scale);
  MEANT_LAUNCH_CHECK("adamw_f32");
  return MEANT_OK;
}

extern "C" int meant_ce_probs(const float* probs, const int64_t* target, float* loss_accum, float* dprobs, int64_t B, int C, void* stream) {
  MEANT_REQUIRE(probs && target && loss_accum && B > 0 && C > 0, MEANT_ERR_ARG, "ce_probs: bad argument");
  hipLaunchKernelGGL(ce_probs_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, (hipStream_t)stream, probs, target, loss_accum, dprobs, B, C);
  MEANT_LAUNCH_CHECK("ce_probs");
  return MEANT_OK;
}

extern "C" int meant_softmax_ce_fwd(const void* logits, int64_t ld, const int64_t* target, int64_t T, int64_t V, int64_t ignore_index,
                                    float* row_loss, float* lse, int dtype, void* stream) {
  MEANT_REQUIRE(logits && target && row_loss && lse, MEANT_ERR_ARG, "softmax_ce_fwd: null pointer");
  MEANT_REQUIRE(T >= 0 && V > 0 && V < (1LL << 31) && ld >= V && ld % 8 == 0 && meant_aligned16(logits), MEANT_ERR_ARG,
                "softmax_ce_fwd: need 0 < V <= ld, ld %% 8 == 0, 16-byte aligned logits");
  if (T == 0) return MEANT_OK;
  DISPATCH_DTYPE(dtype, Tt, hipLaunchKernelGGL(softmax_ce_fwd_kernel<Tt>, dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream,
                                               (const Tt*)logits, ld, target, (int)V, ignore_index, row_loss, lse));
  MEANT_LAUNCH_CHECK("softmax_ce_fwd");
  return MEANT_OK;
}

extern "C" int meant_softmax_ce_bwd(const void* logits, int64_t ld, const int64_t* target, const float* lse, int64_t T, int64_t V,
                                    int64_t ignore_index, const float* gscale, void* dlogits, int dtype, void* stream) {
  MEANT_REQUIRE(logits && target && lse && gscale && dlogits, MEANT_ERR_ARG, "softmax_ce_bwd: null pointer");
  MEANT_REQUIRE(T >= 0 && V > 0 && V < (1LL << 31) && ld >= V && ld % 8 == 0 && meant_aligned16(logits) && meant_aligned16(dlogits),
                MEANT_ERR_ARG, "softmax_ce_bwd: need 0 < V <= ld, ld %% 8 == 0, 16-byte aligned buffers");
  if (T == 0) return MEANT_OK;
  DISPATCH_DTYPE(dtype, Tt, hipLaunchKernelGGL(softmax_ce_bwd_kernel<Tt>, dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream,
                                               (const Tt*)logits, ld, target, lse, (int)V, ignore_index, gscale, (Tt*)dlogits));
  MEANT_LAUNCH_CHECK("softmax_ce_bwd");
  return MEANT_OK;
}
