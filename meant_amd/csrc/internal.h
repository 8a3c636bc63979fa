// Internal (non-exported) interfaces between the translation units of libmeant_hip.so.
#pragma once
#include "common.h"

// Generic strided batched fp32 GEMM on the f32 MFMA (exact fp32 products, v_mfma_f32_32x32x2_f32):
//   C(b1,b2)[m,n] = act(alpha * sum_k A(m,k) B(k,n) + bias[n]) (+ residual(m,n)) (+ C if accumulate)
struct GemmF32Args {
  const void* A; const void* B; void* C;
  int in_dtype = MEANT_F32;   // storage of A and B (bf16 operands are widened to f32 in LDS: slow, exact, any shape)
  int out_dtype = MEANT_F32;  // storage of C, residual, preact
  int64_t M, N, K, nb1, nb2;
  int64_t sA[4];  // b1, b2, m, k
  int64_t sB[4];  // b1, b2, k, n
  int64_t sC[4];  // b1, b2, m, n
  float alpha;
  int accumulate;
  const float* bias;        // [N] or null
  const void* residual;     // same strides as C, or null
  void* preact;             // same strides as C, or null (value before the activation)
  int epilogue;             // meant_epilogue flags
  int ksplit = 0;           // 0: never split K; -1: the launcher may split K over several workgroups per tile (float atomics)
};
int gemm_f32_launch(const GemmF32Args& a, hipStream_t stream);

// bf16 MFMA GEMMs (gemm_bf16.hip).  All operands K-contiguous ("NT"): C[M,N] = A[M,K] B[N,K]^T.
struct GemmBf16Args {
  const bf16* A; int64_t lda;
  const bf16* B; int64_t ldb;
  bf16* C; int64_t ldc;
  int64_t M, N, K;
  const float* bias;        // [N] or null
  const bf16* residual; int64_t ldr;
  bf16* preact;             // ldc stride, or null
  int epilogue;
  // optional rotary epilogue of the fused QKV projection (null rot_qa = off): tables float [S, R]
  const float* rot_qa; const float* rot_qb; const float* rot_ka; const float* rot_kb;
  int rot_S, rot_D, rot_Dh, rot_R;
  // "extended" epilogue (an RMSNorm folded into this Linear, see meant_linear_fwd_rowscale / meant_linear_bwd_dx_norm):
  //   C = act(row_scale[m] * acc + bias) + residual - sub_coef[m] * sub[m, n] + bres_scale * bres[m / bres_rows, n]   (every part optional)
  const float* row_scale = nullptr;          // [M]
  const bf16* sub = nullptr; int64_t ldsub = 0;
  const float* sub_coef = nullptr;           // [M]
  const float* bres = nullptr;               // float [M / bres_rows, N]: one row per group of bres_rows consecutive output rows
  int bres_rows = 1; float bres_scale = 1.f; //   (the gradient of a sequence mean, broadcast over the sequence's tokens)
};
int gemm_bf16_nt_launch(const GemmBf16Args& a, hipStream_t stream);
// dW[N,K] (float, +=) = dY[M,N]^T X[M,K], reduction over the token axis M; dbias[N] += colsum(dY)
// ws: partial-sum workspace of gemm_bf16_tn_ws(M, N, K) bytes, used (and required) only with the `deterministic` option
int gemm_bf16_tn_launch(const bf16* dY, int64_t lddy, const bf16* X, int64_t ldx, float* dW, float* dbias,
                        int64_t M, int64_t N, int64_t K, void* ws, size_t ws_bytes, hipStream_t stream);
size_t gemm_bf16_tn_ws(int64_t M, int64_t N, int64_t K);

// column sums: out[N] (+)= sum_m x[m, n]
int colsum_launch(const void* x, int64_t ldx, float* out, int64_t M, int64_t N, int dtype, int accumulate, hipStream_t stream);

// attention cores
int attn_f32_fwd(const float* qkv, float* o, float* lse, const float* key_mask, int64_t G, int64_t S, int H, int Dh,
                 float scale, int causal, void* ws, size_t ws_bytes, hipStream_t stream, float drop_p = 0.f, uint64_t seed = 0);
int attn_f32_bwd(const float* qkv, const float* o, const float* dout, const float* lse, const float* key_mask, float* dqkv,
                 int64_t G, int64_t S, int H, int Dh, float scale, int causal, void* ws, size_t ws_bytes, hipStream_t stream, float drop_p = 0.f,
                 uint64_t seed = 0);
// dropout on the score matrix (meant/xPosAttention.py:59): the materialised fp32 core for both dtypes (bf16 through fp32 copies in `ws`)
size_t attn_drop_ws(int64_t G, int64_t S, int H, int Dh, int dtype);
int attn_drop_bf16(bool backward, const bf16* qkv, const bf16* o, const bf16* dout, bf16* o_out, float* lse, const float* key_mask, bf16* dqkv,
                   int64_t G, int64_t S, int H, int Dh, float scale, int causal, float drop_p, uint64_t seed, void* ws, size_t ws_bytes,
                   hipStream_t stream);
size_t attn_f32_ws(int64_t G, int64_t S, int H, int Dh);
int attn_bf16_fwd(const bf16* qkv, bf16* o, float* lse, const float* key_mask, int64_t G, int64_t S, int H, int Dh,
                  float scale, int causal, void* ws, size_t ws_bytes, hipStream_t stream);
// rot: (qa, qb, ka, kb) float [S, R] or all-null: when given, the adjoint of the rotary map is applied to dq and dk
struct RotTables { const float* qa; const float* qb; const float* ka; const float* kb; int R; };
int attn_bf16_bwd(const bf16* qkv, const bf16* o, const bf16* dout, const float* lse, const float* key_mask, bf16* dqkv,
                  int64_t G, int64_t S, int H, int Dh, float scale, int causal, RotTables rot, void* ws, size_t ws_bytes,
                  hipStream_t stream);
size_t attn_bf16_ws(int64_t G, int64_t S, int H, int Dh);
size_t attn_bf16_fwd_ws(int64_t G, int64_t S, int H, int Dh);
// S <= 16: one wave per (group, head) (attn_short.hip); same layouts, statistics and semantics, no workspace
bool attn_short_ok(int64_t S, int Dh);
int attn_short_fwd(const bf16* qkv, bf16* o, float* lse, const float* key_mask, int64_t G, int64_t S, int H, int Dh, float scale,
                   int causal, hipStream_t stream);
int attn_short_bwd(const bf16* qkv, const bf16* dout, const float* lse, const float* key_mask, bf16* dqkv, int64_t G, int64_t S, int H,
                   int Dh, float scale, int causal, RotTables rot, hipStream_t stream);
// single-pass backward (attn_bwd1.hip): Dh = 64, S <= 256 or causal S <= 512; bias2 / flags / masks as prepared by attn_bf16_bwd
bool attn_bwd1_ok(int64_t S, int Dh, int causal);
size_t attn_bwd1_ws(int64_t G, int64_t S, int H, int Dh);
int attn_bwd1_launch(const bf16* qkv, const bf16* o, const bf16* dout, const float* lse, const float* bias2, const int* flags,
                     const uint64_t* masks, bf16* dqkv, void* ws1, int64_t G, int64_t S, int H, float scale, int causal, RotTables rot,
                     hipStream_t stream);
