// fp32 parity tier of the attention core (meant/attention.py:43-57, meant/xPosAttention.py:41-63).
// Eager semantics restated with the strided f32-MFMA GEMM: scores materialised per chunk of
// (day, head) groups in the caller's workspace, masked softmax in fp32 with the reference's
// exact mask arithmetic (-inf above the diagonal, then + (1-mask) * -1e9), P V by GEMM again.
// The bf16 tier (attn_bf16.hip) is the fused flash path; this one exists for the 1e-3 fp32 gate.
#include "internal.h"

namespace {

// one wave per score row.  mode 0: compute (max, log-sum), write P.  mode 1: stats given, write P.
// drop_p > 0: dropout ON THE SCORES, where meant/xPosAttention.py:59 has it -- after the causal fill and the padding term, before the
// softmax: a dropped score becomes 0 (also one of a masked position, which thereby becomes visible with weight exp(0 - max)), a kept
// one is divided by 1 - p.  The keep decision of score (g, h, i, j) is hash_uniform(seed, ((g H + h) S + i) S + j) >= p, the same
// in the backward.  Every column is then live, so the causal shortcut (columns <= i only) is off.
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ sc, float* __restrict__ lse,
                                                            const float* __restrict__ key_mask, int64_t g0, int64_t nrows,
                                                            int S, int H, int causal, int mode, float drop_p, uint64_t seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;         // row = ((g_local*H + h)*S + i)
  if (row >= nrows) return;
  const int i = (int)(row % S);
  const int64_t gh = row / S;
  const int64_t g = g0 + gh / H;
  float* p = sc + row * S;
  const float* km = key_mask ? key_mask + g * S : nullptr;
  const int jend = causal ? i + 1 : S;
  const bool drop = drop_p > 0.f;
  const int jlive = drop ? S : jend;                           // columns that can carry weight
  const float keep_scale = drop ? 1.0f / (1.0f - drop_p) : 1.0f;
  const uint64_t rbase = (((uint64_t)g * H + (uint64_t)(gh % H)) * (uint64_t)S + (uint64_t)i) * (uint64_t)S;
  auto score = [&](int j) {
    float v = -INFINITY;
    if (j < jend) {
      v = p[j];
      if (km) v += (1.0f - km[j]) * -1e9f;
    }
    if (drop) v = hash_uniform(seed, rbase + (uint64_t)j) < drop_p ? 0.f : v * keep_scale;
    return v;
  };
  float m = -INFINITY;
  for (int j = lane; j < jlive; j += 64) m = fmaxf(m, score(j));
  // the row statistics are kept as the pair (max, log-sum): with the reference's additive -1e9 padding
  // term a fully padded row has max ~ -1e9, where a single float max+log(sum) would lose the log(sum).
  float logl;
  float* lp = lse + ((g * H + gh % H) * (int64_t)S + i) * 2;
  if (mode == 0) {
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < jlive; j += 64) s += __expf(score(j) - m);
    s = wave_sum(s);
    logl = __logf(s);
    if (lane == 0) { lp[0] = m; lp[1] = logl; }
  } else {
    m = lp[0];
    logl = lp[1];
  }
  for (int j = lane; j < S; j += 64) p[j] = j < jlive ? __expf((score(j) - m) - logl) : 0.f;
}

// dS = P * (dP - rowsum(P * dP)) * scale, in place on dP; with dropout on the scores the gradient of a dropped score is zero and that of
// a kept one carries the 1 / (1 - p)
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ P, float* __restrict__ dP, int64_t g0, int64_t nrows,
                                                                int S, int H, float scale, float drop_p, uint64_t seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= nrows) return;
  const float* p = P + row * S;
  float* d = dP + row * S;
  float dot = 0.f;
  for (int j = lane; j < S; j += 64) dot += p[j] * d[j];
  dot = wave_sum(dot);
  const bool drop = drop_p > 0.f;
  const float keep_scale = drop ? 1.0f / (1.0f - drop_p) : 1.0f;
  const int64_t gh = row / S;
  const uint64_t rbase = (((uint64_t)(g0 + gh / H) * H + (uint64_t)(gh % H)) * (uint64_t)S + (uint64_t)(row % S)) * (uint64_t)S;
  for (int j = lane; j < S; j += 64) {
    float v = p[j] * (d[j] - dot) * scale;
    if (drop) v = hash_uniform(seed, rbase + (uint64_t)j) < drop_p ? 0.f : v * keep_scale;
    d[j] = v;
  }
}

inline int64_t chunk_groups(int64_t G, int64_t S, int H, size_t ws_bytes, int nbuf) {
  const size_t per_g = (size_t)H * S * S * sizeof(float) * nbuf;
  int64_t c = (int64_t)(ws_bytes / per_g);
  if (c > G) c = G;
  const int64_t zcap = 65535 / H;                   // gridDim.z limit of the batched GEMM
  if (c > zcap) c = zcap;
  return c;
}

}  // namespace

size_t attn_f32_ws(int64_t G, int64_t S, int H, int Dh) {
  (void)Dh;
  const size_t per_g = (size_t)H * S * S * sizeof(float) * 2;
  size_t want = per_g * (size_t)G;
  const size_t cap = (size_t)2 << 30;               // 2 GiB of score scratch at most; the call chunks over groups
  if (want > cap) want = (cap / per_g ? cap / per_g : 1) * per_g;
  return want;
}

int attn_f32_fwd(const float* qkv, float* o, float* lse, const float* key_mask, int64_t G, int64_t S, int H, int Dh,
                 float scale, int causal, void* ws, size_t ws_bytes, hipStream_t stream, float drop_p, uint64_t seed) {
  MEANT_REQUIRE(ws, MEANT_ERR_WORKSPACE, "attn_f32_fwd: workspace required");
  const int64_t cg = chunk_groups(G, S, H, ws_bytes, 1);
  MEANT_REQUIRE(cg >= 1, MEANT_ERR_WORKSPACE, "attn_f32_fwd: workspace too small for one group");
  const int64_t D = (int64_t)H * Dh, ld = 3 * D;
  float* sc = (float*)ws;
  for (int64_t g0 = 0; g0 < G; g0 += cg) {
    const int64_t ng = (G - g0 < cg) ? G - g0 : cg;
    const float* base = qkv + g0 * S * ld;
    GemmF32Args a{};
    a.A = base; a.B = base + D; a.C = sc;
    a.M = S; a.N = S; a.K = Dh; a.nb1 = ng; a.nb2 = H;
    a.sA[0] = S * ld; a.sA[1] = Dh; a.sA[2] = ld; a.sA[3] = 1;
    a.sB[0] = S * ld; a.sB[1] = Dh; a.sB[2] = 1; a.sB[3] = ld;
    a.sC[0] = (int64_t)H * S * S; a.sC[1] = S * S; a.sC[2] = S; a.sC[3] = 1;
    a.alpha = scale;
    int rc = gemm_f32_launch(a, stream);
    if (rc) return rc;
    const int64_t nrows = ng * H * S;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)ceil_div(nrows, 4)), dim3(256), 0, stream, sc, lse, key_mask, g0, nrows,
                       (int)S, H, causal, 0, drop_p, seed);
    MEANT_LAUNCH_CHECK("attn_f32_fwd/softmax");
    GemmF32Args b{};
    b.A = sc; b.B = base + 2 * D; b.C = o + g0 * S * D;
    b.M = S; b.N = Dh; b.K = S; b.nb1 = ng; b.nb2 = H;
    b.sA[0] = (int64_t)H * S * S; b.sA[1] = S * S; b.sA[2] = S; b.sA[3] = 1;
    b.sB[0] = S * ld; b.sB[1] = Dh; b.sB[2] = ld; b.sB[3] = 1;
    b.sC[0] = S * D; b.sC[1] = Dh; b.sC[2] = D; b.sC[3] = 1;
    b.alpha = 1.f;
    rc = gemm_f32_launch(b, stream);
    if (rc) return rc;
  }
  return MEANT_OK;
}

int attn_f32_bwd(const float* qkv, const float* o, const float* dout, const float* lse, const float* key_mask, float* dqkv,
                 int64_t G, int64_t S, int H, int Dh, float scale, int causal, void* ws, size_t ws_bytes, hipStream_t stream, float drop_p,
                 uint64_t seed) {
  (void)o;
  MEANT_REQUIRE(ws, MEANT_ERR_WORKSPACE, "attn_f32_bwd: workspace required");
  const int64_t cg = chunk_groups(G, S, H, ws_bytes, 2);
  MEANT_REQUIRE(cg >= 1, MEANT_ERR_WORKSPACE, "attn_f32_bwd: workspace too small for one group");
  const int64_t D = (int64_t)H * Dh, ld = 3 * D;
  const int64_t sS[4] = {(int64_t)H * S * S, S * S, S, 1};     // scores, row-major
  const int64_t sSt[4] = {(int64_t)H * S * S, S * S, 1, S};    // scores, transposed view
  float* P = (float*)ws;
  for (int64_t g0 = 0; g0 < G; g0 += cg) {
    const int64_t ng = (G - g0 < cg) ? G - g0 : cg;
    float* dP = P + (size_t)ng * H * S * S;
    const float* q = qkv + g0 * S * ld;
    const float* k = q + D;
    const float* v = q + 2 * D;
    const float* dO = dout + g0 * S * D;
    float* dq = dqkv + g0 * S * ld;
    const int64_t sQ[4] = {S * ld, Dh, ld, 1};                  // [s, e] view of a q/k/v block
    const int64_t sQt[4] = {S * ld, Dh, 1, ld};                 // [e, s] view
    const int64_t sO[4] = {S * D, Dh, D, 1};
    const int64_t nrows = ng * H * S;
    auto gemm = [&](const float* A, const int64_t* sa, const float* B, const int64_t* sb, float* C, const int64_t* sc_,
                    int64_t M, int64_t N, int64_t K, float alpha) {
      GemmF32Args a{};
      a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.nb1 = ng; a.nb2 = H; a.alpha = alpha;
      for (int i = 0; i < 4; ++i) { a.sA[i] = sa[i]; a.sB[i] = sb[i]; a.sC[i] = sc_[i]; }
      return gemm_f32_launch(a, stream);
    };
    int rc;
    if ((rc = gemm(q, sQ, k, sQt, P, sS, S, S, Dh, scale))) return rc;                 // scores
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)ceil_div(nrows, 4)), dim3(256), 0, stream, P, const_cast<float*>(lse),
                       key_mask, g0, nrows, (int)S, H, causal, 1, drop_p, seed);        // P = exp(scores - lse)
    MEANT_LAUNCH_CHECK("attn_f32_bwd/softmax");
    if ((rc = gemm(P, sSt, dO, sO, dq + 2 * D, sQ, S, Dh, S, 1.f))) return rc;          // dV = P^T dO
    if ((rc = gemm(dO, sO, v, sQt, dP, sS, S, S, Dh, 1.f))) return rc;                  // dP = dO V^T
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)ceil_div(nrows, 4)), dim3(256), 0, stream, P, dP, g0, nrows, (int)S, H, scale,
                       drop_p, seed);
    MEANT_LAUNCH_CHECK("attn_f32_bwd/softmax_bwd");
    if ((rc = gemm(dP, sS, k, sQ, dq, sQ, S, Dh, S, 1.f))) return rc;                   // dQ = dS K
    if ((rc = gemm(dP, sSt, q, sQ, dq + D, sQ, S, Dh, S, 1.f))) return rc;              // dK = dS^T Q
  }
  return MEANT_OK;
}
