// Generic strided batched fp32 GEMM on the gfx950 f32-input MFMA (v_mfma_f32_32x32x2_f32).
//
// This is the fp32 parity tier's matrix engine: every product is an exact f32 fma chain (same
// numerics class as the reference's fp32 eager GEMMs), operands may be arbitrarily strided, so one
// kernel covers x W^T, dY W, dY^T X, Q K^T, P V and their transposes.  With f32 MFMA each lane
// supplies ONE element per operand, so any transposition is free in the LDS addressing.
//   block = 256 threads = 4 waves (2 x 2), tile 64 x 64 x 16, one 32x32 accumulator per wave,
//   register-prefetched global loads, LDS k-major ([k][m] / [k][n]) so fragment reads are
//   conflict-free ds_read_b32.
// Roofline: f32 MFMA peak 157 TFLOP/s (1/16 of bf16) -- the bf16 tier (gemm_bf16.hip) is the
// performance path; this one is for the 1e-3 fp32 gate.
#include "internal.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 16, LDS_LD = 68;

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Args a) {
  __shared__ float As[BK][LDS_LD];
  __shared__ float Bs[BK][LDS_LD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const int64_t b1 = blockIdx.z / a.nb2, b2 = blockIdx.z % a.nb2;
  const TI* A = (const TI*)a.A + b1 * a.sA[0] + b2 * a.sA[1];
  const TI* B = (const TI*)a.B + b1 * a.sB[0] + b2 * a.sB[1];
  TO* Cp = (TO*)a.C;
  const TO* Rp = (const TO*)a.residual;
  TO* Pp = (TO*)a.preact;
  const int64_t coff = b1 * a.sC[0] + b2 * a.sC[1];

  // thread -> (row, k) assignment follows the contiguous axis of each operand
  const bool a_kfast = (a.sA[3] == 1);
  const bool b_kfast = (a.sB[2] == 1);
  int am[4], ak[4], bn[4], bk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (a_kfast) { ak[i] = tid & 15; am[i] = (tid >> 4) + 16 * i; } else { am[i] = tid & 63; ak[i] = (tid >> 6) + 4 * i; }
    if (b_kfast) { bk[i] = tid & 15; bn[i] = (tid >> 4) + 16 * i; } else { bn[i] = tid & 63; bk[i] = (tid >> 6) + 4 * i; }
  }

  float ra[4], rb[4];
  auto load_tile = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m0 + am[i], k = k0 + ak[i];
      ra[i] = (m < a.M && k < a.K) ? to_f(A[m * a.sA[2] + k * a.sA[3]]) : 0.f;
      const int64_t n = n0 + bn[i], kb = k0 + bk[i];
      rb[i] = (n < a.N && kb < a.K) ? to_f(B[kb * a.sB[2] + n * a.sB[3]]) : 0.f;
    }
  };

  f32x16 acc = {};
  const int64_t nk = (a.K + BK - 1) / BK;
  load_tile(0);
  for (int64_t kt = 0; kt < nk; ++kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { As[ak[i]][am[i]] = ra[i]; Bs[bk[i]][bn[i]] = rb[i]; }
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

}  // namespace

int gemm_f32_launch(const GemmF32Args& a, hipStream_t stream) {
  MEANT_REQUIRE(a.A && a.B && a.C, MEANT_ERR_ARG, "gemm_f32: null pointer");
  MEANT_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.nb1 > 0 && a.nb2 > 0, MEANT_ERR_ARG, "gemm_f32: bad shape");
  const int64_t gy = ceil_div(a.M, BM), gz = a.nb1 * a.nb2;
  MEANT_REQUIRE(gy <= 65535 * 1024LL && gz <= 65535, MEANT_ERR_UNSUPPORTED, "gemm_f32: grid too large");
  void (*kern)(GemmF32Args) = nullptr;
  if (a.in_dtype == MEANT_F32 && a.out_dtype == MEANT_F32) kern = gemm_f32_kernel<float, float>;
  else if (a.in_dtype == MEANT_BF16 && a.out_dtype == MEANT_BF16) kern = gemm_f32_kernel<bf16, bf16>;
  else if (a.in_dtype == MEANT_BF16 && a.out_dtype == MEANT_F32) kern = gemm_f32_kernel<bf16, float>;
  MEANT_REQUIRE(kern, MEANT_ERR_UNSUPPORTED, "gemm_f32: unsupported dtype combination");
  const size_t esz_in = a.in_dtype == MEANT_F32 ? 4 : 2, esz_out = a.out_dtype == MEANT_F32 ? 4 : 2;
  meant_route_hit(ROUTE_GEMM_F32);
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
  return gemm_f32_launch(a, (hipStream_t)stream);
}
