// Generic strided batched fp32 GEMM on the gfx950 f32-input MFMA (v_mfma_f32_32x32x2_f32).
//
// This is the fp32 parity tier's matrix engine: every product is an exact f32 fma chain (same
// numerics class as the reference's fp32 eager GEMMs), operands may be arbitrarily strided, so one
// kernel covers x W^T, dY W, dY^T X, Q K^T, P V and their transposes.  With f32 MFMA each lane
// supplies ONE element per operand, so any transposition is free in the LDS addressing.
//   block = 256 threads = 4 waves (2 x 2), tile 64 x 64 x BK, one 32x32 accumulator per wave,
//   register-prefetched global loads, LDS k-major ([k][m] / [k][n]) so fragment reads are
//   conflict-free ds_read_b32.  BK = 16, or 64 once K >= 256: an iteration is one global round trip (~1 us), which 8 MFMAs
//   (0.25 us) cannot cover -- the [groups, 768] x [768, 768] products of the pooled tail ran 62 us at BK = 16.  The products
//   are summed in the same k order either way: results are bit-identical.
// Roofline: f32 MFMA peak 157 TFLOP/s (1/16 of bf16) -- the bf16 tier (gemm_bf16.hip) is the
// performance path; this one is for the 1e-3 fp32 gate.
#include "internal.h"
#include <type_traits>

namespace {

constexpr int BM = 64, BN = 64;

template <typename TI, typename TO, int BK>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Args a) {
  constexpr int LPT = BM * BK / 256;                 // loads per thread and operand tile
  // row stride of the k-major LDS tiles: a k-fast operand is stored by lanes with consecutive k (one row each), so the stride must
  // spread BK consecutive rows over the banks: 68 for 16 rows (4 row groups of 16 lanes), 65 for 64 rows
  constexpr int LDS_LD = BK == 16 ? 68 : 65;
  __shared__ float As[BK][LDS_LD];
  __shared__ float Bs[BK][LDS_LD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const int ksplit = a.ksplit > 1 ? a.ksplit : 1;      // split-K (launcher): this block sums k-tiles [kt0, kt1) and adds atomically
  const int64_t bz = blockIdx.z / ksplit;
  const int ks_ = (int)(blockIdx.z - bz * ksplit);
  const int64_t b1 = bz / a.nb2, b2 = bz % a.nb2;
  const TI* A = (const TI*)a.A + b1 * a.sA[0] + b2 * a.sA[1];
  const TI* B = (const TI*)a.B + b1 * a.sB[0] + b2 * a.sB[1];
  TO* Cp = (TO*)a.C;
  const TO* Rp = (const TO*)a.residual;
  TO* Pp = (TO*)a.preact;
  const int64_t coff = b1 * a.sC[0] + b2 * a.sC[1];

  // thread -> (row, k) assignment follows the contiguous axis of each operand
  const bool a_kfast = (a.sA[3] == 1);
  const bool b_kfast = (a.sB[2] == 1);
  int am[LPT], ak[LPT], bn[LPT], bk[LPT];
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    if (a_kfast) { ak[i] = tid % BK; am[i] = tid / BK + (256 / BK) * i; } else { am[i] = tid & 63; ak[i] = (tid >> 6) + 4 * i; }
    if (b_kfast) { bk[i] = tid % BK; bn[i] = tid / BK + (256 / BK) * i; } else { bn[i] = tid & 63; bk[i] = (tid >> 6) + 4 * i; }
  }

  float ra[LPT], rb[LPT];
  // Element addresses of a thread's LPT loads per operand, formed ONCE and advanced by one k-tile per load: the per-element form
  // (two 64-bit multiplies, two compares, a select behind every load) was ~500 vector instructions per tile in front of its 32
  // MFMAs, on a SIMD that holds a single wave.  Rows / columns past the edge read the tile's first row instead: what they
  // produce is never stored.  Only a partial last k-tile (K not a multiple of BK) needs zeros and takes the checked form.
  const TI* pa[LPT];
  const TI* pb[LPT];
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    pa[i] = A + (m0 + (m0 + am[i] < a.M ? am[i] : 0)) * a.sA[2] + (int64_t)ak[i] * a.sA[3];
    pb[i] = B + (int64_t)bk[i] * a.sB[2] + (n0 + (n0 + bn[i] < a.N ? bn[i] : 0)) * a.sB[3];
  }
  const int64_t step_a = (int64_t)BK * a.sA[3], step_b = (int64_t)BK * a.sB[2];
  auto load_tile = [&](int64_t k0) {                  // the pointers stand on k0; leaves them on k0 + BK
    if (k0 + BK <= a.K) {                             // block-uniform
#pragma unroll
      for (int i = 0; i < LPT; ++i) { ra[i] = to_f(*pa[i]); rb[i] = to_f(*pb[i]); }
    } else {
#pragma unroll
      for (int i = 0; i < LPT; ++i) {
        ra[i] = k0 + ak[i] < a.K ? to_f(*pa[i]) : 0.f;
        rb[i] = k0 + bk[i] < a.K ? to_f(*pb[i]) : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < LPT; ++i) { pa[i] += step_a; pb[i] += step_b; }
  };

  f32x16 acc = {};
  const int64_t nk_all = (a.K + BK - 1) / BK, per = (nk_all + ksplit - 1) / ksplit;
  const int64_t kt0 = ks_ * per, nk = kt0 + per < nk_all ? kt0 + per : nk_all;
  if (kt0 >= nk) return;
#pragma unroll
  for (int i = 0; i < LPT; ++i) { pa[i] += kt0 * step_a; pb[i] += kt0 * step_b; }
  load_tile(kt0 * BK);
  // (A two-buffer form of this loop -- the stores of tile kt + 1 and the loads of tile kt + 2 dealt out over the 32 MFMA slots of
  // tile kt, one barrier per tile -- measured 62 us against 44 us at (1536, 768, 768): hipcc cannot count vmcnt across the loop's
  // back edge and waits for every load in flight in front of the first store of each tile.)
  // (All 64 fragment reads of a tile in front of its 32 MFMAs -- a scheduling barrier keeps hipcc from sinking them back --: 53 us.)
  for (int64_t kt = kt0; kt < nk; ++kt) {
#pragma unroll
    for (int i = 0; i < LPT; ++i) { As[ak[i]][am[i]] = ra[i]; Bs[bk[i]][bn[i]] = rb[i]; }
    __syncthreads();
    if (kt + 1 < nk) load_tile((kt + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float fa = As[kk + (lane >> 5)][wm * 32 + (lane & 31)];
      const float fb = Bs[kk + (lane >> 5)][wn * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
    }
    __syncthreads();
  }

  const int64_t n = n0 + wn * 32 + (lane & 31);
  if (n >= a.N) return;
  const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (m >= a.M) continue;
    const int64_t off = coff + m * a.sC[2] + n * a.sC[3];
    if constexpr (std::is_same<TO, float>::value) {
      if (ksplit > 1) { atomicAdd(Cp + off, acc[r] * a.alpha); continue; }
    }
    float v = acc[r] * a.alpha + bias;
    if (Pp) Pp[off] = from_f<TO>(v);
    if (a.epilogue & MEANT_EPI_GELU) v = gelu_erf(v);
    if (a.epilogue & MEANT_EPI_SIGMOID) v = 1.f / (1.f + __expf(-v));
    if (Rp) v += to_f(Rp[off]);
    if (a.accumulate) v += to_f(Cp[off]);
    Cp[off] = from_f<TO>(v);
  }
}

// column sums of a [M, N] activation into a float accumulator: block = 64 columns x 4 row groups over
// one strip of rows, one float atomic per column per strip (out is "+=" by contract).
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int64_t ldx, float* __restrict__ out,
                                                      int64_t M, int64_t N) {
  __shared__ float red[4][64];
  const int64_t c = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;
  float s = 0.f;
  if (c < N) {
    const int64_t rows_per = (M + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per;
    const int64_t r1 = r0 + rows_per < M ? r0 + rows_per : M;
    for (int64_t r = r0 + rg; r < r1; r += 4) s += to_f(x[r * ldx + c]);
  }
  red[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0 && c < N) atomicAdd(out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// N == 1: y[m] (+)= alpha * sum_k A(m, k) b(k) (the bias products of the weight composition, ops._ComposeLinear).  On the 64 x 64
// tiles above a matrix-vector product is 12-36 workgroups walking all of K one after the other (39-106 us).  Here a workgroup
// owns ML rows and spreads K over its 256 / ML lanes per row -- along whichever axis of A is contiguous, so the loads stay
// coalesced -- and sums the partial results in a fixed order: no atomics, the forward bias stays bit-reproducible.
template <int ML>
__global__ __launch_bounds__(256) void gemv_f32_kernel(const float* __restrict__ A, int64_t sAm, int64_t sAk, const float* __restrict__ b,
                                                        int64_t sBk, float* __restrict__ y, int64_t sCm, int64_t M, int64_t K, float alpha,
                                                        int accumulate, int kfast) {
  constexpr int KL = 256 / ML;
  __shared__ float red[ML][KL + 1];
  const int tid = threadIdx.x;
  const int ml = kfast ? tid / KL : tid % ML, kl = kfast ? tid % KL : tid / ML;
  const int64_t m = (int64_t)blockIdx.x * ML + ml;
  float s = 0.f;
  if (m < M) {
    const float* Ar = A + m * sAm;
#pragma unroll 8
    for (int64_t k = kl; k < K; k += KL) s = fmaf(Ar[k * sAk], b[k * sBk], s);
  }
  red[ml][kl] = s;
  __syncthreads();
  const int64_t mo = (int64_t)blockIdx.x * ML + tid;
  if (tid < ML && mo < M) {
    float v = 0.f;
    for (int q = 0; q < KL; ++q) v += red[tid][q];
    v *= alpha;
    y[mo * sCm] = accumulate ? y[mo * sCm] + v : v;
  }
}

}  // namespace

static int gemv_f32_launch(const GemmF32Args& a, hipStream_t stream) {
  const int kfast = a.sA[3] == 1;
  if (kfast)
    hipLaunchKernelGGL(gemv_f32_kernel<4>, dim3((unsigned)ceil_div(a.M, 4)), dim3(256), 0, stream, (const float*)a.A, a.sA[2], a.sA[3], (const float*)a.B,
                       a.sB[2], (float*)a.C, a.sC[2], a.M, a.K, a.alpha, a.accumulate, 1);
  else
    hipLaunchKernelGGL(gemv_f32_kernel<8>, dim3((unsigned)ceil_div(a.M, 8)), dim3(256), 0, stream, (const float*)a.A, a.sA[2], a.sA[3], (const float*)a.B,
                       a.sB[2], (float*)a.C, a.sC[2], a.M, a.K, a.alpha, a.accumulate, 0);
  MEANT_LAUNCH_CHECK("gemv_f32");
  return MEANT_OK;
}

int gemm_f32_launch(const GemmF32Args& a, hipStream_t stream) {
  MEANT_REQUIRE(a.A && a.B && a.C, MEANT_ERR_ARG, "gemm_f32: null pointer");
  MEANT_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.nb1 > 0 && a.nb2 > 0, MEANT_ERR_ARG, "gemm_f32: bad shape");
  if (a.N == 1 && a.nb1 == 1 && a.nb2 == 1 && a.K >= 256 && a.M < (1LL << 31) && a.in_dtype == MEANT_F32 && a.out_dtype == MEANT_F32 &&
      !a.bias && !a.residual && !a.preact && !a.epilogue) {
    meant_route_hit(ROUTE_GEMM_F32);
    return gemv_f32_launch(a, stream);
  }
  const int64_t gy = ceil_div(a.M, BM), gz = a.nb1 * a.nb2;
  MEANT_REQUIRE(gy <= 65535 * 1024LL && gz <= 65535, MEANT_ERR_UNSUPPORTED, "gemm_f32: grid too large");
  void (*kern)(GemmF32Args) = nullptr;
  const bool deep = a.K >= 256;
  if (a.in_dtype == MEANT_F32 && a.out_dtype == MEANT_F32) kern = deep ? gemm_f32_kernel<float, float, 64> : gemm_f32_kernel<float, float, 16>;
  else if (a.in_dtype == MEANT_BF16 && a.out_dtype == MEANT_BF16) kern = deep ? gemm_f32_kernel<bf16, bf16, 64> : gemm_f32_kernel<bf16, bf16, 16>;
  else if (a.in_dtype == MEANT_BF16 && a.out_dtype == MEANT_F32) kern = deep ? gemm_f32_kernel<bf16, float, 64> : gemm_f32_kernel<bf16, float, 16>;
  MEANT_REQUIRE(kern, MEANT_ERR_UNSUPPORTED, "gemm_f32: unsupported dtype combination");
  const size_t esz_in = a.in_dtype == MEANT_F32 ? 4 : 2, esz_out = a.out_dtype == MEANT_F32 ? 4 : 2;
  meant_route_hit(ROUTE_GEMM_F32);
  // Few tiles and a long K (dW1 = Wqkv^T dWc of the weight composition: 144 tiles, K = 2304, 90 us on half the chip's CUs, one
  // wave per SIMD): K is split over up to 4 workgroups per tile that add their partial sums with float atomics.  Plain fp32
  // output without an epilogue only, only for callers that allow it (ksplit = -1: the weight-composition products; the parity
  // tier's Linear gradients stay ordered), never under option deterministic.
  const int64_t tiles = ceil_div(a.N, BN) * gy * gz;
  if (a.ksplit < 0 && a.out_dtype == MEANT_F32 && !a.bias && !a.residual && !a.preact && !a.epilogue && a.sC[3] == 1 && a.nb1 == 1 && a.nb2 == 1 &&
      a.K >= 1024 && tiles <= meant_num_cus() && gy <= 65535 && !meant_opt(MEANT_OPT_DETERMINISTIC)) {
    GemmF32Args s = a;
    s.ksplit = (int)(a.K >= 2048 ? 4 : 2);
    if (!a.accumulate) {
      if (hipMemset2DAsync(a.C, (size_t)a.sC[2] * sizeof(float), 0, (size_t)a.N * sizeof(float), (size_t)a.M, stream) != hipSuccess) {
        meant_set_error("gemm_f32: memset failed");
        return MEANT_ERR_LAUNCH;
      }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(a.N, BN), (unsigned)gy, (unsigned)(gz * s.ksplit)), dim3(256), 0, stream, s);
    MEANT_LAUNCH_CHECK("gemm_f32");
    return MEANT_OK;
  }
  if (gy <= 65535) {
    hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(a.N, BN), (unsigned)gy, (unsigned)gz), dim3(256), 0, stream, a);
  } else {
    // split very tall problems into row slabs so gridDim.y stays legal
    const int64_t slab = 65535LL * BM;
    for (int64_t m = 0; m < a.M; m += slab) {
      GemmF32Args s = a;
      s.A = (const char*)a.A + m * a.sA[2] * esz_in;
      s.C = (char*)a.C + m * a.sC[2] * esz_out;
      if (a.residual) s.residual = (const char*)a.residual + m * a.sC[2] * esz_out;
      if (a.preact) s.preact = (char*)a.preact + m * a.sC[2] * esz_out;
      s.M = (a.M - m < slab) ? a.M - m : slab;
      hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(a.N, BN), (unsigned)ceil_div(s.M, BM), (unsigned)gz), dim3(256), 0, stream, s);
    }
  }
  MEANT_LAUNCH_CHECK("gemm_f32");
  return MEANT_OK;
}

int colsum_launch(const void* x, int64_t ldx, float* out, int64_t M, int64_t N, int dtype, int accumulate, hipStream_t stream) {
  MEANT_REQUIRE(x && out && M > 0 && N > 0, MEANT_ERR_ARG, "colsum: bad argument");
  if (!accumulate) {
    if (hipMemsetAsync(out, 0, (size_t)N * sizeof(float), stream) != hipSuccess) { meant_set_error("colsum: memset failed"); return MEANT_ERR_LAUNCH; }
  }
  int64_t strips = ceil_div(M, 64);
  if (strips > 512) strips = 512;
  if (meant_opt(MEANT_OPT_DETERMINISTIC)) strips = 1;   // one add per column: nothing races (slow for tall inputs; a debugging mode)
  DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(colsum_kernel<T>, dim3((unsigned)ceil_div(N, 64), (unsigned)strips), dim3(256), 0, stream, (const T*)x, ldx, out, M, N));
  MEANT_LAUNCH_CHECK("colsum");
  return MEANT_OK;
}

extern "C" int meant_gemm_f32_strided(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K, int64_t nb1,
                                      int64_t nb2, const int64_t* sA, const int64_t* sB, const int64_t* sC, float alpha,
                                      int accumulate, void* stream) {
  MEANT_REQUIRE(sA && sB && sC, MEANT_ERR_ARG, "gemm_f32_strided: null strides");
  GemmF32Args a{};
  a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.nb1 = nb1; a.nb2 = nb2;
  for (int i = 0; i < 4; ++i) { a.sA[i] = sA[i]; a.sB[i] = sB[i]; a.sC[i] = sC[i]; }
  a.alpha = alpha; a.accumulate = accumulate;
  a.ksplit = -1;
  return gemm_f32_launch(a, (hipStream_t)stream);
}
