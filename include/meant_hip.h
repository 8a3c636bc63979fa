/*
 * meant_hip.h  --  C ABI of libmeant_hip.so, the MI355X (gfx950 / CDNA4) implementation of the
 * MEANT multimodal-encoder forward/backward hot path.
 *
 * The reference (biirving/meant) is pure Python; the native code on its hot path is what PyTorch
 * dispatches to (ATen/cuBLAS kernels, flash-attn).  Each entry point below replaces the native
 * work behind one reference call site (cited as file:line, relative to the reference root).  The host side
 * (the Python modules under meant_amd/) binds these with ctypes and keeps the reference's nn.Module surface.
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer owned by the caller (PyTorch);
 *     the library never allocates, frees or retains device memory;
 *   - dtype: MEANT_F32 (0) or MEANT_BF16 (1) is the storage type of activations ("act");
 *     statistics, softmax, accumulators, parameter gradients and norm gains are always float;
 *   - tensors are dense row-major in the documented shape; "ld" arguments are row strides in
 *     elements where a tensor is a column slice of a wider buffer;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no call synchronises;
 *   - return value: 0 on success, a negative meant_status otherwise; meant_last_error() returns a
 *     thread-local message.  Nothing aborts or throws across the boundary.
 */
#ifndef MEANT_HIP_H
#define MEANT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum meant_dtype { MEANT_F32 = 0, MEANT_BF16 = 1 };
/* storage-only types of raw inputs (meant_patchify_raw): the reference keeps its price graphs as float64 .npy
 * (in_loop_train.py:48,589); uint8 is what a rendered chart is before anyone converts it */
enum meant_raw_dtype { MEANT_RAW_F32 = 0, MEANT_RAW_BF16 = 1, MEANT_RAW_F64 = 2, MEANT_RAW_U8 = 3 };

enum meant_status {
  MEANT_OK = 0,
  MEANT_ERR_ARG = -1,         /* bad shape / null pointer / misalignment */
  MEANT_ERR_UNSUPPORTED = -2, /* shape or dtype the kernels do not cover  */
  MEANT_ERR_LAUNCH = -3,      /* hipGetLastError() after a launch         */
  MEANT_ERR_WORKSPACE = -4    /* workspace too small                      */
};

/* epilogue flags of meant_linear_fwd */
enum meant_epilogue {
  MEANT_EPI_NONE = 0,
  MEANT_EPI_GELU = 1,     /* y = gelu_erf(x W^T + b); optional pre-activation copy */
  MEANT_EPI_RESIDUAL = 2, /* y = x W^T + b + residual                             */
  MEANT_EPI_SIGMOID = 4   /* y = sigmoid(x W^T + b)                               */
};

int meant_version(void);
const char* meant_last_error(void);
/* number of compute units of the current device (for sizing partial-sum workspaces) */
int meant_num_cus(void);

/* ---- process-wide switches and diagnostics --------------------------------------------------------
 * The library keeps no mutable global state except: a per-device table filled on first use (CU count, raised LDS
 * limits, the tile counters of the streaming GEMM -- one slot per (device, stream)), these options, and the
 * route counters below.  All of it is safe to use from several host threads and with several devices in one
 * process (the reference's nn.DataParallel, pretrain_mlm.py:329): every call acts on the calling thread's current
 * device, whose memory all pointer arguments must belong to.
 * Options (initial value from the environment variable MEANT_<NAME IN CAPITALS>):
 *   "deterministic"    0|1   parameter gradients (dW, dbias) by ordered reductions instead of float atomics: two runs
 *                            on the same inputs are bit-identical; meant_linear_bwd_dw then needs its workspace
 *   "nt_stream"        1|0   streaming 256x256 NT GEMM / one tile per workgroup           (A/B measurements)
 *   "nt_dynamic"       1|0|3|4 streaming GEMM draws tiles from per-XCD counters / fixed walk (A/B measurements) /
 *                            only XCD 0 uses its own counter, all other tiles go through the steal path (tests) /
 *                            fixed walk in runs of one A row panel (lab: measured slower)
 *   "nt_grid_cap"      0|n   cap the streaming GEMM's grid at n workgroups (tests: many tiles per workgroup, steals)
 *   "nt_ragged"        1|0   M not a multiple of 256: the streaming GEMM's last row tile is moved up to end at row M (it
 *                            recomputes rows of its neighbour bit-identically) / streaming head + 128 x 128 tail launch
 *                            (also what operands that alias the output fall back to)
 *   "attn_short"       1|0   sequences of <= 16 tokens run on the one-wave-per-(group, head) kernels / on the tiled ones
 *   "nt_split"         0|1   streaming GEMM: all operand DMA issued by waves 0-3 at the top of a K-step / B tiles by waves 0-3 at
 *                            the top, A tiles by waves 4-7 after their MFMAs
 *   "attn_bwd1"        1|0   attention backward, head dim 64, S <= 256 or causal S <= 512: one pass (scores and dP computed once,
 *                            dS through LDS into the dQ product; persistent workgroups) / the dQ pass followed by the dK, dV pass
 */
int meant_set_option(const char* name, int value);
int meant_get_option(const char* name, int* value);
/* how many launches took a given kernel route since the last reset ("nt128", "nt256", "nt256s", "nt256s_rot",
 * "nt_split", "nt_overlap", "tn128", "tn256", "tn256_det", "tn_tail", "gemm_f32", "attn_fwd", "attn_fwd_d128",
 * "attn_fwd_d96", "attn_bwd" (the two-pass form), "attn_bwd1" (the single-pass form), "attn_bwd_d128", "attn_bwd_d96",
 * "attn_generic", "attn_cls", "attn_short");
 * -1 for an unknown name.  Tests use it to prove that a shape reaches the kernel it is meant to exercise. */
int64_t meant_route_count(const char* route);
void meant_route_reset(void);
/* tiles the streaming GEMM's workgroups took from another XCD's counter, summed over all launches on the current
 * device so far; synchronises the device (test diagnostics) */
int64_t meant_debug_nt_steals(void);

/* ---- RMSNorm ---------------------------------------------------- utils/rms_norm.py:40-57
 * y = scale * x * rinv,  rinv = 1 / (||x||_2 / sqrt(d) + eps)   (eps outside the sqrt)
 * x,y: act [rows, d]; scale: float [d]; rinv: float [rows] (saved for backward).
 * Optional fused inverted dropout on y (meant/meant.py:105,107): keep-prob 1-p, mask regenerated
 * in backward from (seed, element index); p == 0 disables it. */
int meant_rmsnorm_fwd(const void* x, const float* scale, void* y, float* rinv, int64_t rows, int64_t d,
                      float eps, float drop_p, uint64_t seed, int dtype, void* stream);
/* dx: act [rows, d]; dscale: float [d] (overwritten); workspace: meant_rmsnorm_bwd_ws(rows, d) bytes.
 * Optional fusions (NULL to disable): dres act [rows, d] is added to dx (the gradient of the residual
 * branch that shares x, meant/meant.py:71,74); gelu_pre act [rows, d]: x was gelu(gelu_pre), and dx is
 * multiplied by gelu'(gelu_pre) so that it is the gradient w.r.t. gelu_pre (meant/meant.py:64). */
size_t meant_rmsnorm_bwd_ws(int64_t rows, int64_t d);
int meant_rmsnorm_bwd(const void* dy, const void* x, const float* scale, const float* rinv, void* dx,
                      float* dscale, int64_t rows, int64_t d, float eps, float drop_p, uint64_t seed,
                      const void* dres, const void* gelu_pre, int dtype, void* workspace, size_t workspace_bytes,
                      void* stream);

/* ---- RMSNorm, partial / bias forms of the reference class ------- utils/rms_norm.py:44-57 (RMSNorm(d, p, bias))
 * statistics over the first d_part = int(d * p) elements of a row (1 <= d_part <= d), y = scale * x / (rms_part + eps) + offset
 * (offset float [d] or NULL).  No MEANT model constructs these; served by the generic one-row-per-wave kernels. */
int meant_rmsnorm_partial_fwd(const void* x, const float* scale, const float* offset, void* y, float* rinv, int64_t rows,
                              int64_t d, int64_t d_part, float eps, int dtype, void* stream);
int meant_rmsnorm_partial_bwd(const void* dy, const void* x, const float* scale, const float* rinv, void* dx, float* dscale,
                              float* doffset, int64_t rows, int64_t d, int64_t d_part, float eps, int dtype, void* workspace,
                              size_t workspace_bytes, void* stream);

/* ---- RMSNorm beside the sequence mean-pool -------- utils/rms_norm.py:40-57 + meant/meant.py:74,120 -> :231
 * The last RMSNorm of the last encoder layer has one consumer left once that layer's final Linear is evaluated on the
 * pooled features (mean_s(h W^T + b + x) = mean_s(h) W^T + b + mean_s(x)): the mean over the group_rows tokens of a
 * sequence.  meant_rmsnorm_fwd_pooled writes that mean to pooled[g, :] (float [rows / group_rows, d]; ordered sums, bit-reproducible):
 *   pool_input == 0: of y = dropout(RMSNorm(x)) -- y is NOT written (pass NULL);
 *   pool_input != 0: of x, the residual operand -- y is written as usual.
 * gelu_input != 0 (with pool_input == 0): x is the PRE-activation h of the Linear + GELU in front of the norm
 * (meant/meant.py:63-64,106-107); gelu(h) is formed on load, so the activation tensor is never stored.  The backward
 * then takes x == NULL and h as gelu_pre.
 * meant_rmsnorm_bwd_pooled is meant_rmsnorm_bwd with dy (dy_pooled != 0) and / or dres (dres_pooled != 0) given as the
 * float [groups, d] gradient of the pooled features: row r receives g[r / group_rows] / group_rows, no [rows, d]
 * broadcast is materialised.  Supported where meant_rmsnorm_pooled_ok(rows, d, group_rows) != 0 (the packed widths,
 * d = 768 among them, and group_rows a multiple of the rows a wave packs); the caller falls back to the unfused ops
 * otherwise. */
int meant_rmsnorm_pooled_ok(int64_t rows, int64_t d, int64_t group_rows);
int meant_rmsnorm_fwd_pooled(const void* x, const float* scale, void* y, float* rinv, float* pooled, int64_t rows,
                             int64_t d, int64_t group_rows, int pool_input, int gelu_input, float eps, float drop_p,
                             uint64_t seed, int dtype, void* stream);
int meant_rmsnorm_bwd_pooled(const void* dy, int dy_pooled, const void* x, const float* scale, const float* rinv,
                             void* dx, float* dscale, int64_t rows, int64_t d, int64_t group_rows, float eps,
                             float drop_p, uint64_t seed, const void* dres, int dres_pooled, const void* gelu_pre,
                             int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* ---- RMSNorm folded into the Linear that consumes it -------- utils/rms_norm.py:40-57 followed by an nn.Linear:
 *                                                               meant/meant.py:61-63 (encode2[0] -> encode2[1] -> GELU), :103-106
 * Linear(RMSNorm(x)) = r (x) (x W'^T) + b with W' = W diag(g) and r[m] = 1 / (||x_m|| / sqrt(d) + eps): the normalisation is
 * a per-row factor in the GEMM epilogue (meant_linear_fwd_rowscale), the gain a column scale of the weight (meant_colscale);
 * the normalised tensor never exists, forward or backward.  bf16 tier, packed widths (meant_rmsnorm_pooled_ok) only.
 *   meant_rmsnorm_stats      r[rows] from one read of x.
 *   meant_rmsnorm_bwd_chain  the backward of the NEXT norm down the layer (the one behind the GELU: meant_rmsnorm_bwd with
 *                            gelu_pre, token-level dy and the stored activation x; or its pooled form: dy_pooled != 0, x NULL),
 *                            which is where the gradient dpre of this Linear's output comes from.  In the same pass it also
 *                            writes dx_scaled = up_rinv[m] * dpre (the operand of both backward GEMMs), kcoef[m] =
 *                            rowdot(dpre, gelu_pre - up_bias) up_rinv^2 / ((1 - up_eps up_rinv) up_d), up_d the width of the folded norm (see meant_linear_bwd_dx_norm)
 *                            and dbias_up += column sums of the unscaled dpre (the Linear's bias gradient).
 *   meant_colscale / _bwd    W' = W diag(g);  dW += dW' diag(g), dg[k] += sum_n dW'[n,k] W[n,k]   (float, [N, K]). */
int meant_rmsnorm_stats(const void* x, float* rinv, int64_t rows, int64_t d, float eps, int dtype, void* stream);
int meant_rmsnorm_bwd_chain(const void* dy, int dy_pooled, const void* x, const float* scale, const float* rinv,
                            void* dx_scaled, float* dscale, int64_t rows, int64_t d, int64_t group_rows, float eps,
                            float drop_p, uint64_t seed, const void* gelu_pre, const float* up_rinv, const float* up_bias,
                            float up_eps, int64_t up_d, float* kcoef, float* dbias_up, int dtype, void* workspace, size_t workspace_bytes,
                            void* stream);
int meant_colscale(const float* w, const float* g, float* out, int64_t N, int64_t K, void* stream);
int meant_colscale_bwd(const float* dwp, const float* w, const float* g, float* dw, float* dg, int64_t N, int64_t K,
                       void* stream);

/* ---- LayerNorm (heads of meant_vision / meant_tweet) ----- meant/meant_vision.py:147
 * stats: float [rows, 2] (mean, rstd). */
int meant_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats,
                        int64_t rows, int64_t d, float eps, int dtype, void* stream);
int meant_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* stats, void* dx,
                        float* dgamma, float* dbeta, int64_t rows, int64_t d, int dtype, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ---- Linear ------------------- nn.Linear call sites meant/meant.py:59-64,101-107,132-136,195,204
 *                                   and q/v/k/multi_mad in meant/attention.py:31-33,57-60 etc.
 * y[M,N] = x[M,K] w[N,K]^T + bias[N]  (+ epilogue).  x,w,y,residual,preact: act dtype; bias float
 * (may be NULL).  ldx/ldy/ldr: row strides.  preact (optional, EPI_GELU) receives x W^T + b. */
int meant_linear_fwd(const void* x, int64_t ldx, const void* w, const float* bias, const void* residual,
                     int64_t ldr, void* y, int64_t ldy, void* preact, int64_t M, int64_t N, int64_t K,
                     int epilogue, int dtype, void* stream);
/* Fused q|k|v projection + rotary:  qkv[M, 3*H*Dh] = x[M,K] w[3*H*Dh, K]^T + bias, then the rotation of
 * meant_rotary_qk on the q and k blocks (tables may be NULL: plain projection).  In the bf16 tier the rotation
 * rides the GEMM epilogue (no extra pass over qkv).   meant/attention.py:36-40, meant/xPosAttention.py:37-39 */
int meant_qkv_proj_fwd(const void* x, int64_t ldx, const void* w, const float* bias, void* qkv, int64_t M,
                       int64_t K, int64_t S, int H, int Dh, int R, const float* qa, const float* qb,
                       const float* ka, const float* kb, int dtype, void* stream);
/* Linear with its input RMSNorm folded in (see "RMSNorm folded into the Linear" above).
 * forward : y = act(row_scale[m] (x w^T) + bias) (+ residual); w = W diag(g) in the act dtype; preact as in meant_linear_fwd.
 * backward: dx[M,K] = dy_scaled[M,N] w  -  coef[m] x[m,:]  (+ dres) (+ dres_pooled[m / group_rows, :] / group_rows),
 *           dy_scaled / coef from meant_rmsnorm_bwd_chain; dres: the gradient through the residual branch that shares x
 *           (act dtype [M,K]); dres_pooled: float [M / group_rows, K], the gradient of the sequence means of x.
 *           (dW' = dy_scaled^T x through meant_linear_bwd_dw with dbias NULL.) */
int meant_linear_fwd_rowscale(const void* x, int64_t ldx, const void* w, const float* bias, const float* row_scale,
                              const void* residual, int64_t ldr, void* y, int64_t ldy, void* preact, int64_t M, int64_t N,
                              int64_t K, int epilogue, int dtype, void* stream);
int meant_linear_bwd_dx_norm(const void* dy_scaled, int64_t lddy, const void* wT, const void* x, int64_t ldx,
                             const float* coef, const void* dres, int64_t lddres, const float* dres_pooled,
                             int64_t group_rows, void* dx, int64_t lddx, int64_t M, int64_t N, int64_t K, int dtype,
                             void* stream);
/* dx[M,K] = dy[M,N] w[N,K]   (wT is w transposed, [K,N], act dtype: see meant_transpose2d) */
int meant_linear_bwd_dx(const void* dy, int64_t lddy, const void* wT, void* dx, int64_t lddx, int64_t M,
                        int64_t N, int64_t K, int dtype, void* stream);
/* dw[N,K] += dy[M,N]^T x[M,K]  and  dbias[N] += colsum(dy)  -- float accumulators the caller zeroes
 * (or pre-loads: gradient buckets, accumulation across micro-batches).  dbias may be NULL.
 * workspace: meant_linear_bwd_dw_ws(M, N, K, dtype) bytes (0 unless the "deterministic" option is on: then the
 * per-workgroup partial sums go there and are added up in a fixed order); NULL / 0 otherwise. */
size_t meant_linear_bwd_dw_ws(int64_t M, int64_t N, int64_t K, int dtype);
int meant_linear_bwd_dw(const void* dy, int64_t lddy, const void* x, int64_t ldx, float* dw, float* dbias,
                        int64_t M, int64_t N, int64_t K, int dtype, void* workspace, size_t workspace_bytes,
                        void* stream);

/* generic strided batched GEMM, float storage, f32 MFMA (exact fp32 products):
 * C[b1,b2][m,n] = alpha * sum_k A[b1,b2](m,k) * B[b1,b2](k,n)   (+ C if accumulate)
 * element (i,j) of X in batch (b1,b2) is at X + b1*sXb1 + b2*sXb2 + i*sXr + j*sXc. */
int meant_gemm_f32_strided(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K,
                           int64_t nb1, int64_t nb2, const int64_t* sA /*b1,b2,m,k*/,
                           const int64_t* sB /*b1,b2,k,n*/, const int64_t* sC /*b1,b2,m,n*/, float alpha,
                           int accumulate, void* stream);

/* ---- rotary / xPos -------------------- meant/rotary_embedding_torch.py:31-44,96-110
 * In-place on the q and k column blocks of a packed [T, 3*H*Dh] projection buffer:
 *   out[c] = t[c]*A[pos,c] + rot(t)[c]*B[pos,c]  for lanes c < R of every head,
 *   rot(t)[2j] = -t[2j+1], rot(t)[2j+1] = t[2j];   pos = row mod S.
 * qa,qb,ka,kb: float [S, R] (cos*scale, sin*scale for q and for k).  `transpose` != 0 applies
 * the adjoint (backward). */
int meant_rotary_qk(void* qkv, int64_t T, int64_t S, int H, int Dh, int R, const float* qa, const float* qb,
                    const float* ka, const float* kb, int transpose, int dtype, void* stream);

/* ---- attention core -------- meant/attention.py:43-57, meant/xPosAttention.py:41-63
 * qkv: act [G*S, 3*H*Dh] packed (q | k | v column blocks, already rotated); o: act [G*S, H*Dh];
 * lse: float [G, H, S, 2] = (row max m, log sum_j exp(score_j - m)) of the scaled, masked scores --
 *      kept as a pair because the additive -1e9 padding term makes m ~ -1e9 on fully padded rows.  It is
 *      opaque state handed from meant_attn_fwd to meant_attn_bwd of the SAME dtype (natural-log units in
 *      the f32 tier, log2 units in the bf16 tier);
 * key_mask: float [G, S] of {0,1} or NULL (adds (1-mask)*-1e9 to the scores);
 * causal: scores[i,j] = -inf for j > i;  scale is the caller's (the reference: 1/sqrt(dim), NOT 1/sqrt(Dh)).
 * bf16: fused flash kernels for Dh = 64, 96 (the reference's default: 8 heads at d = 768, meant/meant.py:149) and 128
 * (a caller with another head dim < 128 pads every head to 128 columns with zeros -- zero rows in the projection
 * weight -- and keeps its own scale; other Dh take a slow fp32 detour through `workspace`); f32: materialised scores
 * in `workspace`. */
size_t meant_attn_ws(int64_t G, int64_t S, int H, int Dh, int dtype);
/* what meant_attn_fwd alone needs (mask bias, tile flags); meant_attn_ws covers forward and backward (bf16: the backward's row
 * statistics and, for 256 < S <= 512 at Dh = 64, the single-pass backward's partial-dQ scratch of G*H*64 KiB) */
size_t meant_attn_fwd_ws(int64_t G, int64_t S, int H, int Dh, int dtype);
int meant_attn_fwd(const void* qkv, void* o, float* lse, const float* key_mask, int64_t G, int64_t S, int H,
                   int Dh, float scale, int causal, int dtype, void* workspace, size_t workspace_bytes,
                   void* stream);
/* Attention with dropout ON THE SCORE MATRIX, where meant/xPosAttention.py:59 has it (`scores = self.dropout(scores)`: after the
 * causal fill and the key-padding term, before the softmax; p = 0 in every reference model): a dropped score becomes 0 -- also
 * that of a masked position, which thereby becomes visible --, a kept one is divided by 1 - drop_p.  Score (g, h, i, j) is kept iff
 * the counter-based uniform of (seed, ((g H + h) S + i) S + j) is >= drop_p; forward and backward must be given the same
 * (drop_p, seed).  Same buffers and layouts as meant_attn_fwd / meant_attn_bwd; q and k arrive rotated (meant_rotary_qk) and dqkv
 * is the gradient w.r.t. that rotated buffer (apply meant_rotary_qk with transpose = 1 afterwards).  Materialised scores (the
 * fp32 core; bf16 through fp32 copies in the workspace): correct, not fast. */
size_t meant_attn_drop_ws(int64_t G, int64_t S, int H, int Dh, int dtype);
int meant_attn_drop_fwd(const void* qkv, void* o, float* lse, const float* key_mask, int64_t G, int64_t S, int H, int Dh,
                        float scale, int causal, float drop_p, uint64_t seed, int dtype, void* workspace,
                        size_t workspace_bytes, void* stream);
int meant_attn_drop_bwd(const void* qkv, const void* o, const void* do_, const float* lse, const float* key_mask, void* dqkv,
                        int64_t G, int64_t S, int H, int Dh, float scale, int causal, float drop_p, uint64_t seed,
                        int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* dqkv: act [G*S, 3*H*Dh] (every element written); do_: act [G*S, H*Dh].
 * If the rotary tables qa,qb,ka,kb (float [S, R], as in meant_rotary_qk) are given, dq and dk are returned
 * already pulled back through the rotation (the adjoint of meant_rotary_qk), i.e. dqkv is the gradient of
 * the un-rotated projection; pass NULLs (and R = 0) for the gradient of the rotated buffer. */
int meant_attn_bwd(const void* qkv, const void* o, const void* do_, const float* lse, const float* key_mask,
                   void* dqkv, int64_t G, int64_t S, int H, int Dh, float scale, int causal, const float* qa,
                   const float* qb, const float* ka, const float* kb, int R, int dtype, void* workspace,
                   size_t workspace_bytes, void* stream);

/* ---- divided space-time attention glue ----------------------- src/meant/timesformer_pytorch.py:108-145
 * The fork's TimeSformer regroups the patch tokens between the q|k|v projection and the attention core ('(b n) f' for
 * the time attention, '(b f) n' for the space attention, the cls key / value in front of every group) and lets the cls
 * query attend to every token.  The core itself is meant_attn_fwd / _bwd on the regrouped buffer; these move the rows.
 * dst[r, :] = src[idx[r], :] for idx[r] >= 0, zeros for idx[r] == -1, fill[:] (act [W]) for idx[r] == -2; rows of W
 * elements, W % 8 == 0.  (Regrouping, its inverse for the outputs, and the cls-token concatenation of :211-213.) */
int meant_gather_rows(const void* src, const int32_t* idx, const void* fill, void* dst, int64_t n, int64_t W, int dtype,
                      void* stream);
/* The regrouping gather of a packed q|k|v buffer (rows of 3 * H * Dh) with the rotary map applied on the way: row r of dst is
 * row idx[r] of src (zeros for idx[r] < 0), its q and k heads rotated by the tables' row r % S exactly as meant_rotary_qk
 * (transpose = 0) would rotate them in place afterwards (same arithmetic, one rounding); v is copied.
 * (timesformer_pytorch.py:124-131: rearrange, then rot_emb.rotate_queries_and_keys.) */
int meant_gather_rows_rot(const void* src, const int32_t* idx, void* dst, int64_t n, int64_t S, int H, int Dh, int R,
                          const float* qa, const float* qb, const float* ka, const float* kb, int dtype, void* stream);
/* backward of a regrouping gather with index int32 [G, S] into the L rows of each of B sequences, whose column 0 is
 * the same (cls) row in every group and whose other entries are a permutation of the remaining rows: dsrc [B, L, W]
 * is written completely -- dsrc[b, index[g, s]] = ddst[b, g, s] for s >= 1, dsrc[b, index[0, 0]] = sum_g ddst[b, g, 0]. */
int meant_group_scatter(const void* ddst, const int32_t* index, void* dsrc, int64_t B, int64_t L, int64_t G, int64_t S,
                        int64_t W, int dtype, void* stream);
/* cls query: out[b, h*Dh .. +Dh] (row stride ld_out) = softmax_j(scale q[b, 0, h] . k[b, j, h] + (1 - key_mask[b, j]) * -1e9)
 * v[b, j, h] over all L rows of the packed qkv [B, L, 3*H*Dh] (:116-119, cls_mask :252-253).  stats: float [B, H, 2].
 * The backward ADDS to dqkv (the buffer meant_group_scatter has filled): dq of row 0, dk and dv of every row.
 * Dh in {32, 64, 128, 256}; L bounded by the LDS score buffer (~19 k tokens). */
int meant_attn_cls_fwd(const void* qkv, void* out, int64_t ld_out, float* stats, const float* key_mask, int64_t B, int64_t L,
                       int H, int Dh, float scale, int dtype, void* stream);
int meant_attn_cls_bwd(const void* qkv, const void* out, int64_t ld_out, const void* dout, int64_t ld_dout,
                       const float* stats, const float* key_mask, void* dqkv, int64_t B, int64_t L, int H, int Dh,
                       float scale, int dtype, void* stream);

/* time token shift of the patch tokens of x act [B, 1 + frames*n, d] (PreTokenShift, src/meant/timesformer_pytorch.py:28-53):
 * columns [0, d/3) from the next frame, [d/3, 2d/3) unchanged, [2d/3, 3(d/3)) from the previous frame, zeros beyond the
 * clip's ends; the cls row and any remainder columns pass through.  transpose != 0: the adjoint (backward).  x != y. */
int meant_token_shift(const void* x, void* y, int64_t B, int64_t frames, int64_t n, int64_t d, int transpose, int dtype,
                      void* stream);
/* inverted dropout y = x * keep / (1 - p) (nn.Dropout at :70,:101): counter-based mask from (seed, element index), so the
 * backward is the same call on dy.  n % 8 == 0. */
int meant_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream);

/* ---- temporal attention core ----------------------------- meant/temporal.py:44-56
 * q: act [B, H*Dh] (last lag step), kv: act [B*L, 2*H*Dh] packed (k | v); o: act [B, H*Dh];
 * p: float [B, H, L] softmax weights (saved). */
int meant_temporal_attn_fwd(const void* q, const void* kv, void* o, float* p, int64_t B, int L, int H, int Dh,
                            float scale, int dtype, void* stream);
int meant_temporal_attn_bwd(const void* q, const void* kv, const float* p, const void* do_, void* dq, void* dkv,
                            int64_t B, int L, int H, int Dh, float scale, int dtype, void* stream);

/* ---- patchify ------------------------------------------------------ meant/meant.py:194
 * images: float or act [G, C, Hh, Ww] -> patches act [G*(Hh/p)*(Ww/p), p*p*C], channel fastest. */
int meant_patchify(const void* images, int images_dtype, void* patches, int64_t G, int C, int Hh, int Ww, int p,
                   int dtype, void* stream);

/* ---- input pipeline: raw pixels -> normalised patches in one pass ----- in_loop_train.py:48,589-602 (float64
 * graphs, optional global (x - mean) / std), meant/meant.py:194 (patchify).
 * images: raw [G, C, Hh, Ww] of meant_raw_dtype; patches act [G*(Hh/p)*(Ww/p), p*p*C] = ((float)x - mean) * inv_std. */
int meant_patchify_raw(const void* images, int raw_dtype, float mean, float inv_std, void* patches, int64_t G, int C,
                       int Hh, int Ww, int p, int dtype, void* stream);

/* ---- sequence mean-pool ------------------------------------------- meant/meant.py:231
 * x: act [G, S, d] -> out[g, col_off : col_off+d] of a [G, ld_out] buffer (the concat) whose storage type
 * is out_dtype: the pooled features and everything after them (temporal encoder, head: 0.06 % of the
 * FLOPs) run in fp32 also in the bf16 tier. */
int meant_meanpool_fwd(const void* x, void* out, int64_t ld_out, int64_t col_off, int64_t G, int64_t S, int64_t d,
                       int dtype, int out_dtype, void* stream);
int meant_meanpool_bwd(const void* dout, int64_t ld_out, int64_t col_off, void* dx, int64_t G, int64_t S, int64_t d,
                       int dtype, int out_dtype, void* stream);

/* ---- small elementwise helpers ---- */
/* y[r, :] = x[r, :] + v[(r mod period), :]  (temp_embedding add, meant/meant.py:141-142) */
int meant_add_rowvec(const void* x, const float* v, void* y, int64_t rows, int64_t d, int64_t period, int dtype,
                     void* stream);
/* dv[period, d] (float, overwritten) = sum over rows r = i mod period of dy[r, :] */
int meant_add_rowvec_bwd(const void* dy, float* dv, int64_t rows, int64_t d, int64_t period, int dtype, void* stream);
/* dx = dy * gelu'(pre) */
int meant_gelu_bwd(const void* dy, const void* pre, void* dx, int64_t n, int dtype, void* stream);
/* dx = dy * y * (1-y) */
int meant_sigmoid_bwd(const void* dy, const void* y, void* dx, int64_t n, int dtype, void* stream);
/* GEGLU ---------------------------------------------------- src/meant/timesformer_pytorch.py:60-63
 * h: act [rows, 2w] = [a | g]  ->  y: act [rows, w] = a * gelu(g);   dh = [dy * gelu(g) | dy * a * gelu'(g)] */
int meant_geglu_fwd(const void* h, void* y, int64_t rows, int64_t w, int dtype, void* stream);
int meant_geglu_bwd(const void* h, const void* dy, void* dh, int64_t rows, int64_t w, int dtype, void* stream);
/* y = a + b */
int meant_add(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream);
/* dst(act dtype_dst) = src(dtype_src), n elements */
int meant_cast(const void* src, int dtype_src, void* dst, int dtype_dst, int64_t n, void* stream);
/* dst[c, r] (dtype_dst) = src[r, c] (dtype_src); src is [rows, cols] */
int meant_transpose2d(const void* src, int dtype_src, void* dst, int dtype_dst, int64_t rows, int64_t cols,
                      void* stream);
/* embedding gather: out act [n, d] = table float [V, d] rows ids[n] (int64); and its scatter-add backward
 * into dtable float [V, d] (caller zeroes).            nn.Embedding at meant/meant.py:211 */
int meant_embedding_fwd(const float* table, const int64_t* ids, void* out, int64_t n, int64_t d, int64_t V, int dtype,
                        void* stream);
int meant_embedding_bwd(const void* dout, const int64_t* ids, float* dtable, int64_t n, int64_t d, int64_t V, int dtype,
                        void* stream);
/* same, over ids sorted by the caller: sorted_ids[j] ascending, order[j] = row of dout that carries it.  Rows of
 * equal id are summed on chip before one atomic row add, so repeated tokens do not contend (d <= 1024). */
int meant_embedding_bwd_sorted(const void* dout, const int64_t* sorted_ids, const int64_t* order, float* dtable,
                               int64_t n, int64_t d, int64_t V, int dtype, void* stream);
/* the rows of ids in [id_lo, id_hi) only: the table's gradient produced in row slices, so that a data-parallel caller can start
 * the all-reduce of a slice while the next one is still being summed (the 196 MB table is two thirds of the gradient bytes and
 * final only with the last kernel of backward).  The union over a partition of [0, V) equals meant_embedding_bwd_sorted. */
int meant_embedding_bwd_sorted_range(const void* dout, const int64_t* sorted_ids, const int64_t* order, float* dtable,
                                     int64_t n, int64_t d, int64_t V, int64_t id_lo, int64_t id_hi, int dtype, void* stream);

/* ---- train-step tail ------------------------------------------- in_loop_train.py:232-238,547-548
 * CrossEntropyLoss (mean) applied to the model's probabilities [B, C] as the reference does: loss_accum[0] +=
 * loss (caller zeroes it), dprobs (optional) receives d loss / d probs.  A target outside [0, C) turns the loss and
 * that row of dprobs into NaN (no out-of-bounds access). */
int meant_ce_probs(const float* probs, const int64_t* target, float* loss_accum, float* dprobs, int64_t B, int C,
                   void* stream);
/* ---- large-vocabulary softmax cross-entropy (MLM pretrainer) ----------- pretrain_mlm.py:160,178
 * nn.CrossEntropyLoss() on logits act [T, ld] (V <= ld valid columns, the rest padding of the vocabulary GEMM),
 * target int64 [T] with `ignore_index` rows skipped.  fwd: row_loss[t] = logsumexp - logit[target] (0 if ignored),
 * lse[t] saved.  bwd: dlogits[t, j] = (softmax_j - [j == target]) * gscale[0] for j < V, 0 for ignored rows and for
 * the padding columns; gscale is a device scalar (d loss / n_valid), dlogits may alias logits. */
int meant_softmax_ce_fwd(const void* logits, int64_t ld, const int64_t* target, int64_t T, int64_t V, int64_t ignore_index,
                         float* row_loss, float* lse, int dtype, void* stream);
int meant_softmax_ce_bwd(const void* logits, int64_t ld, const int64_t* target, const float* lse, int64_t T, int64_t V,
                         int64_t ignore_index, const float* gscale, void* dlogits, int dtype, void* stream);
/* out_accum[0] += sum(x^2) over a flat float buffer (global gradient norm; caller zeroes the scalar) */
int meant_sumsq_f32(const float* x, int64_t n, float* out_accum, void* stream);
/* One AdamW step (torch.optim.AdamW semantics, `step` >= 1 for the bias corrections) over flat float buffers.
 * Gradients are read as g * grad_scale * clip with clip = min(1, max_norm / (sqrt(*sumsq) * |grad_scale| + 1e-6))
 * (torch.nn.utils.clip_grad_norm_), taken from the device scalar sumsq: no host round trip.  sumsq NULL or
 * max_norm <= 0 disables clipping. */
int meant_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int64_t step, const float* sumsq, float max_norm,
                    float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MEANT_HIP_H */
