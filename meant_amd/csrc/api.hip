// extern "C" entry points for Linear and the attention core: argument checks + dtype dispatch.
#include "internal.h"

// preconditions of the MFMA bf16 NT kernel (global_load_lds moves 16-byte pieces; vector epilogue stores)
static bool bf16_nt_ok(const void* a, int64_t lda, const void* b, int64_t ldb, const void* c, int64_t ldc, const void* r,
                       int64_t ldr, int64_t K) {
  (void)c; (void)ldc; (void)r; (void)ldr;
  return (K % 64) == 0 && (lda % 8) == 0 && (ldb % 8) == 0 && meant_aligned16(a) && meant_aligned16(b);
}

extern "C" int meant_linear_fwd(const void* x, int64_t ldx, const void* w, const float* bias, const void* residual, int64_t ldr,
                                void* y, int64_t ldy, void* preact, int64_t M, int64_t N, int64_t K, int epilogue, int dtype,
                                void* stream) {
  MEANT_REQUIRE(x && w && y, MEANT_ERR_ARG, "linear_fwd: null pointer");
  MEANT_REQUIRE(M > 0 && N > 0 && K > 0 && ldx >= K && ldy >= N, MEANT_ERR_ARG, "linear_fwd: bad shape M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  MEANT_REQUIRE(!(epilogue & MEANT_EPI_RESIDUAL) || residual, MEANT_ERR_ARG, "linear_fwd: residual epilogue without residual pointer");
  if (!(epilogue & MEANT_EPI_RESIDUAL)) residual = nullptr;
  if (dtype == MEANT_F32) {
    MEANT_REQUIRE(!residual || ldr == ldy, MEANT_ERR_UNSUPPORTED, "linear_fwd(f32): residual stride must equal output stride");
    GemmF32Args a{};
    a.A = x; a.B = w; a.C = y;
    a.M = M; a.N = N; a.K = K; a.nb1 = 1; a.nb2 = 1;
    a.sA[2] = ldx; a.sA[3] = 1; a.sB[2] = 1; a.sB[3] = K; a.sC[2] = ldy; a.sC[3] = 1;
    a.alpha = 1.f; a.bias = bias; a.residual = residual; a.preact = preact; a.epilogue = epilogue;
    return gemm_f32_launch(a, (hipStream_t)stream);
  }
  if (dtype == MEANT_BF16 && !bf16_nt_ok(x, ldx, w, K, y, ldy, residual, ldr, K)) {
    // odd shapes (K not a multiple of 64, unaligned rows: e.g. the class head): exact-f32 generic path on bf16 storage
    MEANT_REQUIRE(!residual || ldr == ldy, MEANT_ERR_UNSUPPORTED, "linear_fwd(bf16 generic): residual stride must equal output stride");
    GemmF32Args a{};
    a.in_dtype = MEANT_BF16; a.out_dtype = MEANT_BF16;
    a.A = x; a.B = w; a.C = y;
    a.M = M; a.N = N; a.K = K; a.nb1 = 1; a.nb2 = 1;
    a.sA[2] = ldx; a.sA[3] = 1; a.sB[2] = 1; a.sB[3] = K; a.sC[2] = ldy; a.sC[3] = 1;
    a.alpha = 1.f; a.bias = bias; a.residual = residual; a.preact = preact; a.epilogue = epilogue;
    return gemm_f32_launch(a, (hipStream_t)stream);
  }
  if (dtype == MEANT_BF16) {
    GemmBf16Args a{};
    a.A = (const bf16*)x; a.lda = ldx; a.B = (const bf16*)w; a.ldb = K; a.C = (bf16*)y; a.ldc = ldy;
    a.M = M; a.N = N; a.K = K; a.bias = bias; a.residual = (const bf16*)residual; a.ldr = ldr; a.preact = (bf16*)preact;
    a.epilogue = epilogue;
    return gemm_bf16_nt_launch(a, (hipStream_t)stream);
  }
  meant_set_error("linear_fwd: unknown dtype %d", dtype);
  return MEANT_ERR_ARG;
}

// Fused q|k|v projection + rotary (meant/attention.py:36-40, meant/xPosAttention.py:37-39): qkv[M, 3D] = x W^T + b with
// the rotary / xPos rotation of the q and k blocks applied in the GEMM epilogue (bf16 tier) or by the rotary kernel.
extern "C" int meant_qkv_proj_fwd(const void* x, int64_t ldx, const void* w, const float* bias, void* qkv, int64_t M, int64_t K,
                                  int64_t S, int H, int Dh, int R, const float* qa, const float* qb, const float* ka,
                                  const float* kb, int dtype, void* stream) {
  MEANT_REQUIRE(x && w && qkv && H > 0 && Dh > 0 && S > 0 && M % S == 0, MEANT_ERR_ARG, "qkv_proj_fwd: bad argument");
  const bool rot = qa != nullptr;
  MEANT_REQUIRE(!rot || (qb && ka && kb && R > 0 && R % 2 == 0 && R <= Dh), MEANT_ERR_ARG, "qkv_proj_fwd: bad rotary tables");
  const int64_t D = (int64_t)H * Dh, N = 3 * D;
  const bool fused = rot && dtype == MEANT_BF16 && bf16_nt_ok(x, ldx, w, K, qkv, N, nullptr, 0, K) && Dh % 8 == 0 && R % 8 == 0;
  if (fused) {
    GemmBf16Args a{};
    a.A = (const bf16*)x; a.lda = ldx; a.B = (const bf16*)w; a.ldb = K; a.C = (bf16*)qkv; a.ldc = N;
    a.M = M; a.N = N; a.K = K; a.bias = bias;
    a.rot_qa = qa; a.rot_qb = qb; a.rot_ka = ka; a.rot_kb = kb;
    a.rot_S = (int)S; a.rot_D = (int)D; a.rot_Dh = Dh; a.rot_R = R;
    return gemm_bf16_nt_launch(a, (hipStream_t)stream);
  }
  int rc = meant_linear_fwd(x, ldx, w, bias, nullptr, 0, qkv, N, nullptr, M, N, K, MEANT_EPI_NONE, dtype, stream);
  if (rc || !rot) return rc;
  return meant_rotary_qk(qkv, M, S, H, Dh, R, qa, qb, ka, kb, 0, dtype, stream);
}

extern "C" int meant_linear_bwd_dx(const void* dy, int64_t lddy, const void* wT, void* dx, int64_t lddx, int64_t M, int64_t N,
                                   int64_t K, int dtype, void* stream) {
  MEANT_REQUIRE(dy && wT && dx, MEANT_ERR_ARG, "linear_bwd_dx: null pointer");
  MEANT_REQUIRE(M > 0 && N > 0 && K > 0 && lddy >= N && lddx >= K, MEANT_ERR_ARG, "linear_bwd_dx: bad shape");
  if (dtype == MEANT_F32) {
    GemmF32Args a{};
    a.A = dy; a.B = wT; a.C = dx;
    a.M = M; a.N = K; a.K = N; a.nb1 = 1; a.nb2 = 1;
    a.sA[2] = lddy; a.sA[3] = 1; a.sB[2] = 1; a.sB[3] = N; a.sC[2] = lddx; a.sC[3] = 1;
    a.alpha = 1.f;
    return gemm_f32_launch(a, (hipStream_t)stream);
  }
  if (dtype == MEANT_BF16 && !bf16_nt_ok(dy, lddy, wT, N, dx, lddx, nullptr, 0, N)) {
    GemmF32Args a{};
    a.in_dtype = MEANT_BF16; a.out_dtype = MEANT_BF16;
    a.A = dy; a.B = wT; a.C = dx;
    a.M = M; a.N = K; a.K = N; a.nb1 = 1; a.nb2 = 1;
    a.sA[2] = lddy; a.sA[3] = 1; a.sB[2] = 1; a.sB[3] = N; a.sC[2] = lddx; a.sC[3] = 1;
    a.alpha = 1.f;
    return gemm_f32_launch(a, (hipStream_t)stream);
  }
  if (dtype == MEANT_BF16) {
    GemmBf16Args a{};
    a.A = (const bf16*)dy; a.lda = lddy; a.B = (const bf16*)wT; a.ldb = N; a.C = (bf16*)dx; a.ldc = lddx;
    a.M = M; a.N = K; a.K = N;
    return gemm_bf16_nt_launch(a, (hipStream_t)stream);
  }
  meant_set_error("linear_bwd_dx: unknown dtype %d", dtype);
  return MEANT_ERR_ARG;
}

// ---- Linear with its input RMSNorm folded in (bf16 tier, MFMA kernels only) ------------------------------------------------
// forward:  y = act(row_scale[m] (x w^T) + bias) (+ residual), w = W diag(g) (meant_colscale), row_scale = 1 / (rms(x) + eps)
// (meant_rmsnorm_stats): Linear(RMSNorm(x)) without the normalised tensor ever existing (utils/rms_norm.py:40-57 followed by
// an nn.Linear, meant/meant.py:59-64).
extern "C" int meant_linear_fwd_rowscale(const void* x, int64_t ldx, const void* w, const float* bias, const float* row_scale,
                                         const void* residual, int64_t ldr, void* y, int64_t ldy, void* preact, int64_t M, int64_t N,
                                         int64_t K, int epilogue, int dtype, void* stream) {
  MEANT_REQUIRE(x && w && y && row_scale, MEANT_ERR_ARG, "linear_fwd_rowscale: null pointer");
  MEANT_REQUIRE(M > 0 && N > 0 && K > 0 && ldx >= K && ldy >= N, MEANT_ERR_ARG, "linear_fwd_rowscale: bad shape");
  MEANT_REQUIRE(dtype == MEANT_BF16 && bf16_nt_ok(x, ldx, w, K, y, ldy, residual, ldr, K) && (ldy & 7) == 0 && !(epilogue & MEANT_EPI_SIGMOID),
                MEANT_ERR_UNSUPPORTED, "linear_fwd_rowscale: bf16 tier, K %% 64 == 0, 16-byte aligned rows only");
  MEANT_REQUIRE(!(epilogue & MEANT_EPI_RESIDUAL) || residual, MEANT_ERR_ARG, "linear_fwd_rowscale: residual epilogue without residual pointer");
  GemmBf16Args a{};
  a.A = (const bf16*)x; a.lda = ldx; a.B = (const bf16*)w; a.ldb = K; a.C = (bf16*)y; a.ldc = ldy;
  a.M = M; a.N = N; a.K = K; a.bias = bias; a.residual = (epilogue & MEANT_EPI_RESIDUAL) ? (const bf16*)residual : nullptr; a.ldr = ldr;
  a.preact = (bf16*)preact; a.epilogue = epilogue; a.row_scale = row_scale;
  return gemm_bf16_nt_launch(a, (hipStream_t)stream);
}

// input gradient of the same:  dx = dy_scaled wT^T - coef[m] x[m, :] (+ dres), dy_scaled = row_scale * dy and coef as produced by
// meant_rmsnorm_bwd_chain: the RMSNorm backward's own term rides this GEMM's epilogue; no pass of its own.
extern "C" int meant_linear_bwd_dx_norm(const void* dy_scaled, int64_t lddy, const void* wT, const void* x, int64_t ldx, const float* coef,
                                        const void* dres, int64_t lddres, const float* dres_pooled, int64_t group_rows, void* dx, int64_t lddx,
                                        int64_t M, int64_t N, int64_t K, int dtype, void* stream) {
  MEANT_REQUIRE(dy_scaled && wT && x && coef && dx, MEANT_ERR_ARG, "linear_bwd_dx_norm: null pointer");
  MEANT_REQUIRE(M > 0 && N > 0 && K > 0 && lddy >= N && lddx >= K && ldx >= K, MEANT_ERR_ARG, "linear_bwd_dx_norm: bad shape");
  MEANT_REQUIRE(dtype == MEANT_BF16 && bf16_nt_ok(dy_scaled, lddy, wT, N, dx, lddx, nullptr, 0, N) && (lddx & 7) == 0 && (ldx & 7) == 0 &&
                meant_aligned16(x) && (!dres || ((lddres & 7) == 0 && meant_aligned16(dres))),
                MEANT_ERR_UNSUPPORTED, "linear_bwd_dx_norm: bf16 tier, N %% 64 == 0, 16-byte aligned rows only");
  GemmBf16Args a{};
  a.A = (const bf16*)dy_scaled; a.lda = lddy; a.B = (const bf16*)wT; a.ldb = N; a.C = (bf16*)dx; a.ldc = lddx;
  a.M = M; a.N = K; a.K = N;
  a.residual = (const bf16*)dres; a.ldr = lddres;
  a.sub = (const bf16*)x; a.ldsub = ldx; a.sub_coef = coef;
  if (dres_pooled) {                                   // + dres_pooled[m / group_rows, :] / group_rows: the gradient of mean_s(x)
    MEANT_REQUIRE(group_rows > 0 && M % group_rows == 0 && group_rows < (1LL << 30), MEANT_ERR_ARG, "linear_bwd_dx_norm: bad group_rows");
    a.bres = dres_pooled; a.bres_rows = (int)group_rows; a.bres_scale = 1.0f / (float)group_rows;
  }
  return gemm_bf16_nt_launch(a, (hipStream_t)stream);
}

extern "C" size_t meant_linear_bwd_dw_ws(int64_t M, int64_t N, int64_t K, int dtype) {
  return dtype == MEANT_BF16 ? gemm_bf16_tn_ws(M, N, K) : 0;
}

extern "C" int meant_linear_bwd_dw(const void* dy, int64_t lddy, const void* x, int64_t ldx, float* dw, float* dbias, int64_t M,
                                   int64_t N, int64_t K, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  MEANT_REQUIRE(dy && x && dw, MEANT_ERR_ARG, "linear_bwd_dw: null pointer");
  MEANT_REQUIRE(M > 0 && N > 0 && K > 0 && lddy >= N && ldx >= K, MEANT_ERR_ARG, "linear_bwd_dw: bad shape");
  if (dtype == MEANT_F32) {
    GemmF32Args a{};
    a.A = dy; a.B = x; a.C = dw;
    a.M = N; a.N = K; a.K = M; a.nb1 = 1; a.nb2 = 1;
    a.sA[2] = 1; a.sA[3] = lddy; a.sB[2] = ldx; a.sB[3] = 1; a.sC[2] = K; a.sC[3] = 1;
    a.alpha = 1.f; a.accumulate = 1;
    int rc = gemm_f32_launch(a, (hipStream_t)stream);
    if (rc) return rc;
    if (dbias) return colsum_launch(dy, lddy, dbias, M, N, MEANT_F32, 1, (hipStream_t)stream);
    return MEANT_OK;
  }
  if (dtype == MEANT_BF16 && (N < 8 || K < 8 || (lddy & 7) || (ldx & 7) || !meant_aligned16(dy) || !meant_aligned16(x))) {
    GemmF32Args a{};
    a.in_dtype = MEANT_BF16; a.out_dtype = MEANT_F32;
    a.A = dy; a.B = x; a.C = dw;
    a.M = N; a.N = K; a.K = M; a.nb1 = 1; a.nb2 = 1;
    a.sA[2] = 1; a.sA[3] = lddy; a.sB[2] = ldx; a.sB[3] = 1; a.sC[2] = K; a.sC[3] = 1;
    a.alpha = 1.f; a.accumulate = 1;
    int rc = gemm_f32_launch(a, (hipStream_t)stream);
    if (rc) return rc;
    if (dbias) return colsum_launch(dy, lddy, dbias, M, N, MEANT_BF16, 1, (hipStream_t)stream);
    return MEANT_OK;
  }
  if (dtype == MEANT_BF16)
    return gemm_bf16_tn_launch((const bf16*)dy, lddy, (const bf16*)x, ldx, dw, dbias, M, N, K, workspace, workspace_bytes, (hipStream_t)stream);
  meant_set_error("linear_bwd_dw: unknown dtype %d", dtype);
  return MEANT_ERR_ARG;
}

extern "C" size_t meant_attn_ws(int64_t G, int64_t S, int H, int Dh, int dtype) {
  return dtype == MEANT_F32 ? attn_f32_ws(G, S, H, Dh) : attn_bf16_ws(G, S, H, Dh);
}

extern "C" size_t meant_attn_fwd_ws(int64_t G, int64_t S, int H, int Dh, int dtype) {
  return dtype == MEANT_F32 ? attn_f32_ws(G, S, H, Dh) : attn_bf16_fwd_ws(G, S, H, Dh);
}

static int attn_check(const char* name, int64_t G, int64_t S, int H, int Dh) {
  MEANT_REQUIRE(G > 0 && S > 0 && H > 0 && Dh > 0, MEANT_ERR_ARG, "%s: bad shape G=%lld S=%lld H=%d Dh=%d", name, (long long)G, (long long)S, H, Dh);
  return MEANT_OK;
}

extern "C" int meant_attn_fwd(const void* qkv, void* o, float* lse, const float* key_mask, int64_t G, int64_t S, int H, int Dh,
                              float scale, int causal, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  MEANT_REQUIRE(qkv && o && lse, MEANT_ERR_ARG, "attn_fwd: null pointer");
  int rc = attn_check("attn_fwd", G, S, H, Dh);
  if (rc) return rc;
  if (dtype == MEANT_F32)
    return attn_f32_fwd((const float*)qkv, (float*)o, lse, key_mask, G, S, H, Dh, scale, causal, workspace, workspace_bytes, (hipStream_t)stream);
  if (dtype == MEANT_BF16)
    return attn_bf16_fwd((const bf16*)qkv, (bf16*)o, lse, key_mask, G, S, H, Dh, scale, causal, workspace, workspace_bytes, (hipStream_t)stream);
  meant_set_error("attn_fwd: unknown dtype %d", dtype);
  return MEANT_ERR_ARG;
}

// ---- attention with dropout on the score matrix (meant/xPosAttention.py:59) ----
extern "C" size_t meant_attn_drop_ws(int64_t G, int64_t S, int H, int Dh, int dtype) { return attn_drop_ws(G, S, H, Dh, dtype); }

extern "C" int meant_attn_drop_fwd(const void* qkv, void* o, float* lse, const float* key_mask, int64_t G, int64_t S, int H, int Dh,
                                   float scale, int causal, float drop_p, uint64_t seed, int dtype, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  MEANT_REQUIRE(qkv && o && lse, MEANT_ERR_ARG, "attn_drop_fwd: null pointer");
  MEANT_REQUIRE(drop_p >= 0.f && drop_p < 1.f, MEANT_ERR_ARG, "attn_drop_fwd: drop_p = %g is not in [0, 1)", (double)drop_p);
  int rc = attn_check("attn_drop_fwd", G, S, H, Dh);
  if (rc) return rc;
  MEANT_REQUIRE(workspace && workspace_bytes >= attn_drop_ws(G, S, H, Dh, dtype), MEANT_ERR_WORKSPACE, "attn_drop_fwd: workspace too small");
  if (dtype == MEANT_F32)
    return attn_f32_fwd((const float*)qkv, (float*)o, lse, key_mask, G, S, H, Dh, scale, causal, workspace, workspace_bytes, (hipStream_t)stream,
                        drop_p, seed);
  if (dtype == MEANT_BF16)
    return attn_drop_bf16(false, (const bf16*)qkv, nullptr, nullptr, (bf16*)o, lse, key_mask, nullptr, G, S, H, Dh, scale, causal, drop_p, seed,
                          workspace, workspace_bytes, (hipStream_t)stream);
  meant_set_error("attn_drop_fwd: unknown dtype %d", dtype);
  return MEANT_ERR_ARG;
}

extern "C" int meant_attn_drop_bwd(const void* qkv, const void* o, const void* do_, const float* lse, const float* key_mask, void* dqkv,
                                   int64_t G, int64_t S, int H, int Dh, float scale, int causal, float drop_p, uint64_t seed, int dtype,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  MEANT_REQUIRE(qkv && o && do_ && lse && dqkv, MEANT_ERR_ARG, "attn_drop_bwd: null pointer");
  MEANT_REQUIRE(drop_p >= 0.f && drop_p < 1.f, MEANT_ERR_ARG, "attn_drop_bwd: drop_p = %g is not in [0, 1)", (double)drop_p);
  int rc = attn_check("attn_drop_bwd", G, S, H, Dh);
  if (rc) return rc;
  MEANT_REQUIRE(workspace && workspace_bytes >= attn_drop_ws(G, S, H, Dh, dtype), MEANT_ERR_WORKSPACE, "attn_drop_bwd: workspace too small");
  if (dtype == MEANT_F32)
    return attn_f32_bwd((const float*)qkv, (const float*)o, (const float*)do_, lse, key_mask, (float*)dqkv, G, S, H, Dh, scale, causal, workspace,
                        workspace_bytes, (hipStream_t)stream, drop_p, seed);
  if (dtype == MEANT_BF16)
    return attn_drop_bf16(true, (const bf16*)qkv, (const bf16*)o, (const bf16*)do_, nullptr, const_cast<float*>(lse), key_mask, (bf16*)dqkv, G, S, H, Dh, scale, causal,
                          drop_p, seed, workspace, workspace_bytes, (hipStream_t)stream);
  meant_set_error("attn_drop_bwd: unknown dtype %d", dtype);
  return MEANT_ERR_ARG;
}

extern "C" int meant_attn_bwd(const void* qkv, const void* o, const void* do_, const float* lse, const float* key_mask, void* dqkv,
                              int64_t G, int64_t S, int H, int Dh, float scale, int causal, const float* qa, const float* qb,
                              const float* ka, const float* kb, int R, int dtype, void* workspace, size_t workspace_bytes,
                              void* stream) {
  MEANT_REQUIRE((qa != nullptr) == (qb != nullptr) && (qa != nullptr) == (ka != nullptr) && (qa != nullptr) == (kb != nullptr), MEANT_ERR_ARG,
                "attn_bwd: rotary tables must be given all or none");
  const RotTables rot{qa, qb, ka, kb, R};
  MEANT_REQUIRE(qkv && o && do_ && lse && dqkv, MEANT_ERR_ARG, "attn_bwd: null pointer");
  int rc = attn_check("attn_bwd", G, S, H, Dh);
  if (rc) return rc;
  if (dtype == MEANT_F32) {
    rc = attn_f32_bwd((const float*)qkv, (const float*)o, (const float*)do_, lse, key_mask, (float*)dqkv, G, S, H, Dh, scale, causal,
                      workspace, workspace_bytes, (hipStream_t)stream);
    if (rc || !qa) return rc;
    return meant_rotary_qk(dqkv, G * S, S, H, Dh, R, qa, qb, ka, kb, 1, MEANT_F32, stream);
  }
  if (dtype == MEANT_BF16)
    return attn_bf16_bwd((const bf16*)qkv, (const bf16*)o, (const bf16*)do_, lse, key_mask, (bf16*)dqkv, G, S, H, Dh, scale, causal,
                         rot, workspace, workspace_bytes, (hipStream_t)stream);
  meant_set_error("attn_bwd: unknown dtype %d", dtype);
  return MEANT_ERR_ARG;
}
