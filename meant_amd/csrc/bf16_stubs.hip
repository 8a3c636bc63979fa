// TEMPORARY: bf16 MFMA paths not written yet.
#include "internal.h"
int gemm_bf16_nt_launch(const GemmBf16Args&, hipStream_t) { meant_set_error("bf16 GEMM not built yet"); return MEANT_ERR_UNSUPPORTED; }
int gemm_bf16_tn_launch(const bf16*, int64_t, const bf16*, int64_t, float*, float*, int64_t, int64_t, int64_t, hipStream_t) { meant_set_error("bf16 GEMM not built yet"); return MEANT_ERR_UNSUPPORTED; }
int attn_bf16_fwd(const bf16*, bf16*, float*, const float*, int64_t, int64_t, int, int, float, int, hipStream_t) { meant_set_error("bf16 attention not built yet"); return MEANT_ERR_UNSUPPORTED; }
int attn_bf16_bwd(const bf16*, const bf16*, const bf16*, const float*, const float*, bf16*, int64_t, int64_t, int, int, float, int, void*, size_t, hipStream_t) { meant_set_error("bf16 attention not built yet"); return MEANT_ERR_UNSUPPORTED; }
size_t attn_bf16_ws(int64_t, int64_t, int, int) { return 0; }
