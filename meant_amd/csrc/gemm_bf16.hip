// K5/K2: bf16 MFMA GEMMs of the MEANT path (the d x d Linears are 84 % of the step's FLOPs).
//
//   gemm_bf16_nt : C[M,N] = A[M,K] B[N,K]^T (+bias, GELU / sigmoid, +residual), both operands K-contiguous.
//                  Serves x W^T (forward) and dY (W^T)^T (input gradient, with a pre-transposed weight).
//   gemm_bf16_tn : dW[N,K] += dY[M,N]^T X[M,K] (reduction over the token axis M), fp32 atomics into dW.
//
// MI355X mapping
//   * 256 threads = 4 waves (2 x 2), block tile 128 x 128, K-step 64; each wave owns 64 x 64 of C in
//     16 (NT: 16x16x32) or 4 (TN: 32x32x16) MFMA accumulator tiles (64 accumulator VGPRs).
//   * operands go HBM -> LDS directly (global_load_lds, 16 B per lane, 1 KiB per wave-instruction),
//     double-buffered: the DMA of K-tile t+1 is in flight while tile t feeds the MFMAs; one barrier
//     per K-tile.
//   * LDS images are lane-linear (a DMA constraint), so the bank-conflict swizzle is applied on the
//     SOURCE address and again on the read:
//        NT: 128-byte rows, 16-byte chunk c of row r lives at slot c ^ ((r>>1)&7)   -> ds_read_b128
//            of an MFMA fragment (16 rows x one chunk per 16-lane group) touches 16 distinct slots.
//        TN: 256-byte rows (the reduction index is the ROW), chunk c of row r at slot c ^ (4*(r&3));
//            fragments come out of ds_read_b64_tr_b16 (hardware transpose), 4 consecutive rows each.
//   * blockIdx -> tile mapping is XCD-aware: the 8 XCDs have private L2s and blocks are dealt
//     round-robin, so ids are remapped to give each XCD a contiguous run of tiles; consecutive tiles
//     share the A row-panel (N/128 tiles per panel) which is then served from that XCD's L2, and the
//     weight matrix (<= 7 MB) stays resident in every L2.
//   * epilogue: accumulators -> LDS (fp32) -> coalesced 16-byte bf16 stores with bias / activation /
//     residual applied in fp32 and a single rounding.
// Roofline: MFMA-bound (dense bf16 peak 2.5 PFLOP/s); algorithmic FLOPs 2*M*N*K per launch.
#include "internal.h"
#include <type_traits>
#include <atomic>
#include <mutex>
#include <unordered_map>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand per stage
constexpr int NT_LDS = 4 * TILE_BYTES;           // A0 B0 A1 B1 = 64 KiB (also holds the 128x128 fp32 epilogue tile)

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}

// bijective XCD-aware remap of a linear block id (guide: cdna_hip_programming.md T1)
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// ------------------------------------------------------------------------------------------------
// NT kernel
// stage one 128 x 64 bf16 tile (rows row0.., columns k0..k0+63 of a K-contiguous matrix) into LDS.
// wave w issues DMA pieces 4w..4w+3; piece i covers tile rows 8i..8i+7 (8 lanes per 128-byte row).
__device__ __forceinline__ void nt_stage(const bf16* __restrict__ g, int64_t ld, int64_t row0, int64_t nrows, int64_t k0,
                                          char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i;
    const int r = piece * 8 + (lane >> 3);
    const int slot = lane & 7;
    const int c = slot ^ ((r >> 1) & 7);
    int64_t gr = row0 + r;
    gr = gr < nrows ? gr : nrows - 1;            // clamp: rows past the edge are computed and discarded
    glds16(g + gr * ld + k0 + c * 8, lds_tile + piece * 1024);
  }
}

// one output row segment of 8 columns: v = acc + bias already; applies (optional) rotary, activation, residual,
// rounds once and stores 16 bytes.  Rotary epilogue (fused QKV projection): columns are [q | k | v] blocks of
// rot_D = H*Dh columns, a head is Dh columns, lanes c < R of every q/k head are rotated with the tables of the
// token's position m mod S:  out[c] = t[c]*A[pos,c] + rot(t)[c]*B[pos,c]  (meant/rotary_embedding_torch.py:31-44).
// the tail of nt_store_row8's vector path without the store: activation, residual, one rounding
__device__ __forceinline__ bf16x8 nt_finish_row8(const GemmBf16Args& a, int64_t m, int64_t n, float (&v)[8]) {
  if (a.epilogue & MEANT_EPI_GELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = gelu_erf_fast(v[e]);
  }
  if (a.epilogue & MEANT_EPI_SIGMOID) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
  }
  if (a.residual) {
    const bf16x8 rr = *reinterpret_cast<const bf16x8*>(a.residual + m * a.ldr + n);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += (float)rr[e];
  }
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
  return o;
}

// rotate 8 consecutive columns (4 pairs) with table rows A[0..7], B[0..7]
__device__ __forceinline__ void rot_apply8(float (&v)[8], const f32x4& a0, const f32x4& a1, const f32x4& b0, const f32x4& b1) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const float t0 = v[e], t1 = v[e + 1];
    const float A0 = e < 4 ? a0[e] : a1[e - 4], A1 = e < 4 ? a0[e + 1] : a1[e - 3];
    const float B0 = e < 4 ? b0[e] : b1[e - 4], B1 = e < 4 ? b0[e + 1] : b1[e - 3];
    v[e] = t0 * A0 - t1 * B0;
    v[e + 1] = t1 * A1 + t0 * B1;
  }
}

// pre_res: the 8 residual values of this row segment when the caller has requested them ahead of time (streaming kernel)
template <bool STREAM_OUT = false>
__device__ __forceinline__ void nt_store_row8(const GemmBf16Args& a, int64_t m, int64_t n, float (&v)[8], bool vec_ok,
                                              const bf16x8* pre_res = nullptr) {
  if (vec_ok) {
    if (a.preact) {
      bf16x8 p;
#pragma unroll
      for (int e = 0; e < 8; ++e) p[e] = (bf16)v[e];
      *reinterpret_cast<bf16x8*>(a.preact + m * a.ldc + n) = p;
    }
    if (a.rot_qa) {
      const int sec = (int)(n / a.rot_D);
      const int c = (int)((n - (int64_t)sec * a.rot_D) % a.rot_Dh);
      if (sec < 2 && c < a.rot_R) {
        const int pos = (int)(m % a.rot_S);
        const float* A = (sec ? a.rot_ka : a.rot_qa) + (int64_t)pos * a.rot_R + c;
        const float* B = (sec ? a.rot_kb : a.rot_qb) + (int64_t)pos * a.rot_R + c;
        rot_apply8(v, *reinterpret_cast<const f32x4*>(A), *reinterpret_cast<const f32x4*>(A + 4),
                   *reinterpret_cast<const f32x4*>(B), *reinterpret_cast<const f32x4*>(B + 4));
      }
    }
    if (a.epilogue & MEANT_EPI_GELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_erf_fast(v[e]);
    }
    if (a.epilogue & MEANT_EPI_SIGMOID) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
    }
    if (a.residual) {
      const bf16x8 rr = pre_res ? *pre_res : *reinterpret_cast<const bf16x8*>(a.residual + m * a.ldr + n);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += (float)rr[e];
    }
    if (a.sub) {
      const bf16x8 sv = *reinterpret_cast<const bf16x8*>(a.sub + m * a.ldsub + n);
      const float kc = a.sub_coef[m];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaf(-kc, (float)sv[e], v[e]);
    }
    if (a.bres) {
      const float* bp = a.bres + (m / a.bres_rows) * a.N + n;
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaf(a.bres_scale, e < 4 ? b0[e] : b1[e - 4], v[e]);
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
    if (STREAM_OUT) __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(a.C + m * a.ldc + n));
    else *reinterpret_cast<bf16x8*>(a.C + m * a.ldc + n) = o;
  } else {
    for (int e = 0; e < 8 && n + e < a.N; ++e) {
      float x = v[e];
      if (a.preact) a.preact[m * a.ldc + n + e] = (bf16)x;
      if (a.epilogue & MEANT_EPI_GELU) x = gelu_erf_fast(x);
      if (a.epilogue & MEANT_EPI_SIGMOID) x = 1.f / (1.f + __expf(-x));
      if (a.residual) x += (float)a.residual[m * a.ldr + n + e];
      if (a.sub) x -= a.sub_coef[m] * (float)a.sub[m * a.ldsub + n + e];
      if (a.bres) x += a.bres_scale * a.bres[(m / a.bres_rows) * a.N + n + e];
      a.C[m * a.ldc + n + e] = (bf16)x;
    }
  }
}

__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_kernel(GemmBf16Args a, int ntm, int ntn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int id = xcd_remap(blockIdx.x, ntm * ntn);
  const int tm = id / ntn, tn = id - tm * ntn;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int nk = (int)(a.K / BK);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  nt_stage(a.A, a.lda, m0, a.M, 0, smem, wave, lane);
  nt_stage(a.B, a.ldb, n0, a.N, 0, smem + TILE_BYTES, wave, lane);
  __syncthreads();                                  // (drains the DMA: vmcnt(0) + barrier)

  const int frow = lane & 15, fkg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * 2 * TILE_BYTES;
    char* nxt = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
    if (kt + 1 < nk) {
      nt_stage(a.A, a.lda, m0, a.M, (int64_t)(kt + 1) * BK, nxt, wave, lane);
      nt_stage(a.B, a.ldb, n0, a.N, (int64_t)(kt + 1) * BK, nxt + TILE_BYTES, wave, lane);
    }
    const char* At = cur + (wm * 64) * 128;
    const char* Bt = cur + TILE_BYTES + (wn * 64) * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
      const int c = ks * 4 + fkg;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = i * 16 + frow;
        const int off = r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
        af[i] = *reinterpret_cast<const bf16x8*>(At + off);
        bfr[i] = *reinterpret_cast<const bf16x8*>(Bt + off);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: acc -> LDS fp32 [128][132] -> coalesced stores --------------------------------
  constexpr int LDC = BN + 4;
  float* Cs = reinterpret_cast<float*>(smem);       // 128*132*4 = 67584 B <= NT_LDS + slack (see launch)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        Cs[(wm * 64 + i * 16 + fkg * 4 + e) * LDC + wn * 64 + j * 16 + frow] = acc[i][j][e];
  __syncthreads();
  // thread -> (row = pass*16 + tid/16, 8 columns at (tid%16)*8)
  const int cc = (tid & 15) * 8;
  const int64_t n = n0 + cc;
  float bias[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias[e] = (a.bias && n + e < a.N) ? a.bias[n + e] : 0.f;
  const bool vec_ok = (n + 8 <= a.N) && ((a.ldc & 7) == 0) && (!a.residual || (a.ldr & 7) == 0);
#pragma unroll 2
  for (int pass = 0; pass < 8; ++pass) {
    const int r = pass * 16 + (tid >> 4);
    const int64_t m = m0 + r;
    if (m >= a.M || n >= a.N) continue;
    float v[8];
    const f32x4 lo = *reinterpret_cast<const f32x4*>(Cs + r * LDC + cc);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(Cs + r * LDC + cc + 4);
    const float rs = a.row_scale ? a.row_scale[m] : 1.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf(e < 4 ? lo[e] : hi[e - 4], rs, bias[e]);
    nt_store_row8(a, m, n, v, vec_ok);
  }
}

// ------------------------------------------------------------------------------------------------
// NT kernel, 256 x 256 block tile, 512 threads = 8 waves (2 along M x 4 along N), 128 x 64 per wave.
// Why: the 128 x 128 kernel is bound by the L2 -> LDS load path (~40 GB/s per CU sustained = 64 FLOP per
// loaded byte x 40 GB/s x 256 CUs = 650 TFLOP/s, exactly what it measures); a 256 x 256 tile needs half the
// operand bytes per FLOP.  One block per CU (128 KiB of LDS: 2 stages x (A 32 KiB + B 32 KiB)), 2 waves/SIMD.
constexpr int B2 = 256;
constexpr int T2_BYTES = B2 * BK * 2;              // 32 KiB per operand per stage

// 32 pieces of 1 KiB (8 rows each) per tile; wave w issues pieces 4w .. 4w+3
__device__ __forceinline__ void nt256_stage(const bf16* __restrict__ g, int64_t ld, int64_t row0, int64_t nrows, int64_t k0,
                                             char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i;
    const int r = piece * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int64_t gr = row0 + r;
    gr = gr < nrows ? gr : nrows - 1;
    glds16(g + gr * ld + k0 + c * 8, lds_tile + piece * 1024);
  }
}

__global__ __launch_bounds__(512, 2) void gemm_bf16_nt256_kernel(GemmBf16Args a, int ntm, int ntn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int id = xcd_remap(blockIdx.x, ntm * ntn);
  const int tm = id / ntn, tn = id - tm * ntn;
  const int64_t m0 = (int64_t)tm * B2, n0 = (int64_t)tn * B2;
  const int nk = (int)(a.K / BK);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  nt256_stage(a.A, a.lda, m0, a.M, 0, smem, wave, lane);
  nt256_stage(a.B, a.ldb, n0, a.N, 0, smem + T2_BYTES, wave, lane);
  __syncthreads();

  const int frow = lane & 15, fkg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * 2 * T2_BYTES;
    char* nxt = smem + ((kt + 1) & 1) * 2 * T2_BYTES;
    // Only waves 0-3 issue DMA (all 64 pieces of the next stage, 16 each): a wave issues no MFMA while it feeds pieces
    // to the address unit, and the two waves of a SIMD (w, w+4) would otherwise both do that right after the barrier
    // with the matrix pipe idle.  This way waves 4-7 start their MFMAs at once and their partners follow (+5 % on the
    // K-loop; issuing late instead -- after the MFMAs -- exposes the load latency at the barrier and measured -10 %).
    const bool late = __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;
    if (!late && kt + 1 < nk) {
      nt256_stage(a.A, a.lda, m0, a.M, (int64_t)(kt + 1) * BK, nxt, wave, lane);
      nt256_stage(a.B, a.ldb, n0, a.N, (int64_t)(kt + 1) * BK, nxt + T2_BYTES, wave, lane);
      nt256_stage(a.A, a.lda, m0, a.M, (int64_t)(kt + 1) * BK, nxt, wave + 4, lane);
      nt256_stage(a.B, a.ldb, n0, a.N, (int64_t)(kt + 1) * BK, nxt + T2_BYTES, wave + 4, lane);
    }
    const char* At = cur + (wm * 128) * 128;
    const char* Bt = cur + T2_BYTES + (wn * 64) * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[8], bfr[4];
      const int c = ks * 4 + fkg;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = i * 16 + frow;
        af[i] = *reinterpret_cast<const bf16x8*>(At + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = j * 16 + frow;
        bfr[j] = *reinterpret_cast<const bf16x8*>(Bt + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // swapped: acc = C^T tile
      __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
  }

  // ---- epilogue: 4 passes of 64 rows through LDS (fp32 [64][260]) ---------------------------------
  constexpr int LDC = B2 + 4;
  float* Cs = reinterpret_cast<float*>(smem);
  const int cc = (tid & 31) * 8;
  const int64_t n = n0 + cc;
  float bias[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias[e] = (a.bias && n + e < a.N) ? a.bias[n + e] : 0.f;
  const bool vec_ok = (n + 8 <= a.N) && ((a.ldc & 7) == 0) && (!a.residual || (a.ldr & 7) == 0);
  // Operands were fed swapped (B as the MFMA "A"), so acc[i][j][e] = C[m = i*16 + frow][n = j*16 + 4*fkg + e]: a lane
  // owns 4 CONSECUTIVE columns of one row and puts them into the staging tile with one ds_write_b128 (4x fewer LDS
  // store instructions than the natural layout).  Each pass takes 32 rows from each of the two wave rows, so all 8
  // waves write in every pass.
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    if (pass) __syncthreads();
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const int i = pass * 2 + ii;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(Cs + (wm * 32 + ii * 16 + frow) * LDC + wn * 64 + j * 16 + fkg * 4) = acc[i][j];
    }
    __syncthreads();
#pragma unroll 2
    for (int q = 0; q < 4; ++q) {
      const int r = q * 16 + (tid >> 5);               // staging row: (wave row, 32 rows of this pass)
      const int64_t m = m0 + (r >> 5) * 128 + pass * 32 + (r & 31);
      if (m >= a.M || n >= a.N) continue;
      float v[8];
      const f32x4 lo = *reinterpret_cast<const f32x4*>(Cs + r * LDC + cc);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(Cs + r * LDC + cc + 4);
      const float rs = a.row_scale ? a.row_scale[m] : 1.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaf(e < 4 ? lo[e] : hi[e - 4], rs, bias[e]);
      nt_store_row8(a, m, n, v, vec_ok);
    }
  }
}

// The 160 KiB of LDS can hold a ring of FIVE 32 KiB operand tiles (A_0 B_0 A_1 B_1 A_2 ...; tile t lives in slot
// t mod 5) instead of two 64 KiB stages.  While step n computes from A_n / B_n, the DMA fills B_{n+1} (needed next
// step) and A_{n+2} (needed in two), so only half of a stage has to make the issue -> land round trip within one
// K-step; the barrier waits with a COUNTED vmcnt for exactly the tiles the next step needs.  (The two-stage kernel
// above is bound by that round trip: 64 KiB per CU through a 64 B/clk address path is 1024 clocks of issue alone
// before the last piece even starts its trip.)
constexpr int RING = 5;

// ------------------------------------------------------------------------------------------------
// Streaming (persistent) NT kernel: the kernel this file is built around for the tall GEMMs of the step.
//
// Same 256 x 256 tile and 8 waves as gemm_bf16_nt256_kernel, staged through the 5-slot ring; one workgroup per CU walks a
// SEQUENCE of tiles and treats all their K-steps as one stream: the ring keeps prefetching across the tile boundary
// (B of the next step, A of the one after -- whichever tile they belong to), and the epilogue needs no workgroup-wide
// staging buffer: each wave transposes its 16 x 64 pieces through a private 4 KiB patch in the two ring slots the
// finished step has just vacated and stores complete 128-byte lines.  What that buys at K = 768, where a tile is only
// 12 K-steps long:
//   * no prologue bubble per tile (the first operands of tile t+1 land while tile t still computes);
//   * no epilogue bubble: the 128 KiB of output per tile are fire-and-forget stores that drain under the next tile's
//     K-loop.  In the one-tile-per-workgroup kernels all 256 CUs finish together and write 32 MiB in one burst while
//     the matrix pipes wait for the stores (measured: 4 us of a 24 us tile for the stores, 5 us for the rest of the
//     prologue/epilogue).
// Requirements (the launcher falls back to the kernels above otherwise): M % 256 == 0, N % 256 == 0, K % 64 == 0,
// K >= 128, ldc % 8 == 0.
// Tile hand-out of the streaming kernel.  A fixed walk (tile += grid) is fragile: when another kernel -- the other HIP stream's,
// or a collective's -- holds a few CUs at launch, the workgroups that start late still own a full share of the tiles and the
// launch takes ~1.5x as long (measured with 8 of 256 CUs pinned: 1.07 -> 1.66 ms).  So only the FIRST tile of a workgroup is
// fixed; every further tile is drawn from a counter.  There is one counter per XCD so that the tiles an XCD works on stay
// neighbours (shared A panels / weights in its L2); an XCD that runs dry steals from the next one.  Tile k of XCD x is
// id(x, k) = (k / CH) * G + x * CH + k % CH with CH = G / 8 workgroups per XCD -- the same order as the fixed walk.
// The draw costs no stall: lane 0 of wave 4 (a wave that issues no DMA) sends the atomic at the top of K-step 3 and picks the
// answer up behind the wait that ends the step anyway; it publishes the tile id to a per-workgroup mailbox in global memory
// (LDS is full) with a fire-and-forget store, acknowledged by the end of step 4; every wave requests the mailbox at the top of
// step 5 and has it at the end of that step.  (Blocking versions of the same protocol cost 5-7 %: memory latency under this
// kernel's own load is ~4 us, two K-steps.)  The last workgroup to leave zeroes the counters for the next launch.
struct alignas(64) TileSched {
  unsigned next[8];
  unsigned done;
  unsigned steals;       // diagnostics: tiles taken from another XCD's counter (never reset by the kernel)
  unsigned pad[6];
  unsigned mailbox[512];
};
constexpr int N_SCHED_SLOTS = 256;
__device__ TileSched g_tile_sched[N_SCHED_SLOTS];

// EXT: the extended epilogue (GemmBf16Args::row_scale / sub / sub_coef) -- an instantiation of its own so that the plain
// kernel's register budget (236 of 256) is not touched
template <int DBG, bool ROT, bool EXT = false>
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt256s_kernel(GemmBf16Args a, int ntm, int ntn, TileSched* __restrict__ sched, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nk = (int)(a.K / BK);
  const int ntiles = ntm * ntn, G = gridDim.x;
  // workgroup b runs on XCD b % 8 (round-robin dispatch): in every round each XCD takes G/8 consecutive tile ids,
  // i.e. a few rows of tiles that share their A panels and all of W through that XCD's L2
  const int CH = G >> 3, xcd = blockIdx.x & 7;
  // mode 4 (lab, fixed walk only): a workgroup walks RUNS of ntn consecutive tiles -- all column tiles of one A row panel -- before it
  // moves G panels on, so that an A panel is requested by one CU nine times in a row rather than by nine CUs at about the same time
  const int run = mode == 4 ? ntn : 1;
  auto fixed_next = [&](int t) { return (t + 1) % run != 0 ? t + 1 : t + 1 + (G - 1) * run; };
  int tile = (xcd * CH + (blockIdx.x >> 3)) * run;
  const bool dynamic = sched != nullptr && nk >= 8;  // the draw travels during K-steps 3..5 and is needed at step nk - 2
  auto leave = [&]() {                                 // last workgroup out resets the counters
    if (sched && threadIdx.x == 0) {
      if (atomicAdd(&sched->done, 1u) == (unsigned)G - 1u) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sched->next[i] = 0;
        sched->done = 0;
      }
    }
  };
  if (tile >= ntiles) { leave(); return; }
  auto tile_of = [&](int x, unsigned k) { return (int)((k / (unsigned)CH) * (unsigned)G + (unsigned)(x * CH) + k % (unsigned)CH); };

  // waves 0-3 issue all DMA; wave w owns pieces 4w..4w+3 and 4(w+4)..4(w+4)+3 (1 KiB = 8 rows of 128 B) of every tile.
  // Piece p of a wave starts 8 * pp rows below its first one (pp = p, or p + 12 for the second group), which is a
  // uniform offset; the swizzled 16-byte column of a lane only alternates between two values (c0, c0 ^ 4) with the
  // parity of p.  So a lane keeps two 32-bit offsets per operand and everything else lives in scalar registers.
  // SPLIT (DBG bit 7): the DMA issue is shared out in TIME as well: waves 0-3 issue the next step's B tile at the top of a K-step and
  // then compute; their SIMD partners, waves 4-7, compute first and issue the A tile of the step after next when their MFMAs are
  // done -- the part of the step they used to spend waiting at the barrier.  A `global_load_lds` costs the issuing wave 100-185
  // cycles in a phase that is also reading fragments (MI355X_MICROARCH.md, LDS-DMA piece issue cost): sixteen of them in front of a
  // wave's 64 MFMAs were longer than the MFMAs themselves.
  constexpr bool SPLIT = (DBG & 128) != 0;
  const bool issuer = wave < 4;
  const int lw = wave & 3;
  const int r0 = lw * 32 + (lane >> 3);
  const int c0 = (lane & 7) ^ ((r0 >> 1) & 7);
  const unsigned oA[2] = {(unsigned)(r0 * a.lda + c0 * 8), (unsigned)(r0 * a.lda + (c0 ^ 4) * 8)};
  const unsigned oB[2] = {(unsigned)(r0 * a.ldb + c0 * 8), (unsigned)(r0 * a.ldb + (c0 ^ 4) * 8)};
  auto stage = [&](const bf16* base, const unsigned (&off)[2], int64_t ld, int slot) {
    char* dst = smem + slot * T2_BYTES;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int pp = p < 4 ? p : p + 12;
      glds16(base + (int64_t)(8 * pp) * ld + off[p & 1], dst + (lw * 4 + pp) * 1024);
    }
  };
  auto origin = [&](int t, const bf16*& pa, const bf16*& pb, int64_t& m0, int64_t& n0) {
    const int tm = t / ntn, tn = t - tm * ntn;
    m0 = (int64_t)tm * B2;
    m0 = m0 + B2 <= a.M ? m0 : a.M - B2;               // ragged M: the last row tile is moved up to END at row M (see the launcher)
    n0 = (int64_t)tn * B2;
    pa = a.A + m0 * a.lda;
    pb = a.B + n0 * a.ldb;
  };

  const bf16 *pA, *pB, *pAn = nullptr, *pBn = nullptr;      // operand origins of this tile and of the next one
  int64_t m0, n0, m0n = 0, n0n = 0;
  origin(tile, pA, pB, m0, n0);
  int next = dynamic ? -1 : fixed_next(tile);          // dynamic: unknown until the draw of this tile has come back
  bool has_next = !dynamic && next < ntiles;
  if (has_next) origin(next, pAn, pBn, m0n, n0n);
  unsigned* mailbox = sched ? &sched->mailbox[blockIdx.x] : nullptr;

  if (SPLIT) {
    if (issuer) {
      stage(pB, oB, a.ldb, 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      stage(pA, oA, a.lda, 0);
      stage(pA + BK, oA, a.lda, 2);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
  } else if (issuer) {
    stage(pA, oA, a.lda, 0);
    stage(pB, oB, a.ldb, 1);
    stage(pA + BK, oA, a.lda, 2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  bool first_tile = true;
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const int frow = lane & 15, fkg = lane >> 4;
  int sA = 0;                                          // ring slot of the current step's A tile; B sits in the next one
  for (;;) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < nk; ++kt) {
      const int sB = sA + 1 == RING ? 0 : sA + 1;
      const int s3 = sA + 3 >= RING ? sA + 3 - RING : sA + 3, s4 = sA + 4 >= RING ? sA + 4 - RING : sA + 4;
      const bool in1 = kt + 1 < nk, in2 = kt + 2 < nk;
      // tile draw, part 1 (see TileSched): requests only, nothing is waited for here -- they are older than this step's DMA,
      // so the wait at the end of the step covers them.  The registers the answers land in are defined and consumed inside
      // one iteration; hipcc does not know they are in flight, and tools/isa_inflight.py (run by tests/test_isa_guards.py)
      // checks in the ISA that nothing touches them in between.
      unsigned mail = 0, ticket = 0;
      const bool reading = dynamic && kt == 5;
      if (reading) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(mail) : "v"(mailbox) : "memory");
      const bool drawer = dynamic && kt == 3 && wave == 4 && lane == 0;
      // mode 3 (tests): only XCD 0 draws from its own counter; everybody else behaves as if theirs had run dry, so that every
      // tile of the other seven counters is handed out by the steal path below
      const bool own = !(mode == 3 && xcd != 0);
      if (drawer && own) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ticket) : "v"(&sched->next[xcd]), "v"(1u) : "memory");
      const bool more2 = in2 || has_next;
      if (issuer && !(DBG & 1)) {
        if (in1) stage(pB + (int64_t)(kt + 1) * BK, oB, a.ldb, s3);
        else if (has_next) stage(pBn, oB, a.ldb, s3);
        if (!SPLIT) {
          if (in2) stage(pA + (int64_t)(kt + 2) * BK, oA, a.lda, s4);
          else if (has_next) stage(pAn + (int64_t)(kt + 2 - nk) * BK, oA, a.lda, s4);
        }
      }
      const char* At = smem + sA * T2_BYTES + (wm * 128) * 128;
      const char* Bt = smem + sB * T2_BYTES + (wn * 64) * 128;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[8], bfr[4];
        const int c = ks * 4 + fkg;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = i * 16 + frow;
          af[i] = *reinterpret_cast<const bf16x8*>(At + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // DIRECT: MFMA block j is fed the W rows 32 (j / 2) + 8 (frow / 4) + 4 (j % 2) + frow % 4, so that a lane's results of
          // blocks 2 jp and 2 jp + 1 are EIGHT CONSECUTIVE output columns (32 jp + 8 fkg ..) and can be stored without a transpose
          const int r = (DBG & 32) ? 32 * (j >> 1) + 8 * (frow >> 2) + 4 * (j & 1) + (frow & 3) : j * 16 + frow;
          bfr[j] = *reinterpret_cast<const bf16x8*>(Bt + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // swapped: acc = C^T tile
        __builtin_amdgcn_s_setprio(0);
      }
      // the next step needs its A tile (issued a step ago) and its B tile (issued above); the A tile after that may
      // stay in flight.  The output stores of the previous tile are older than all of these and retire first.
      asm volatile("" ::: "memory");
      // Waves 4-7 issue no DMA: all they can have in flight are old output stores and the requests above.  In the three
      // steps of the draw they wait for everything (the stores are long gone by then).
      const bool sched_step = dynamic && !issuer && kt >= 3 && kt <= 5;
      if (SPLIT) {
        // waves 0-3: the B tile issued above (and, older, the previous tile's output stores) must have landed.  Waves 4-7: the A tile
        // they issued at the end of the PREVIOUS step must have; in a tile's first step that request is older than the sixteen (or
        // more) output stores of the tile before, which may stay in flight
        if (!issuer && kt == 0 && !first_tile) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else if ((DBG & 8) && !issuer && !sched_step) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // lab: stores of waves 4-7 drain freely
      else if (more2 && !sched_step && !(DBG & 1)) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      asm volatile("" : "+v"(mail), "+v"(ticket) : : "memory");
      // tile draw, part 2: the answers are here
      if (drawer) {                                    // K-step 3: resolve and publish (acknowledged by the end of step 4)
        int id = own ? tile_of(xcd, (unsigned)CH + ticket) : ntiles;
        if (id >= ntiles) {                            // this XCD is dry: look at the others' counters, take from one with work left
          unsigned seen[8];
#pragma unroll
          for (int x = 0; x < 8; ++x) seen[x] = __hip_atomic_load(&sched->next[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int att = 1; id >= ntiles && att < 8; ++att) {
            const int x2 = (xcd + att) & 7;
            if (tile_of(x2, (unsigned)CH + seen[x2]) >= ntiles) continue;
            id = tile_of(x2, (unsigned)CH + atomicAdd(&sched->next[x2], 1u));
            if (id < ntiles) atomicAdd(&sched->steals, 1u);
          }
        }
        const unsigned pub = id < ntiles ? (unsigned)id : 0xffffffffu;
        asm volatile("global_store_dword %0, %1, off sc1" ::"v"(mailbox), "v"(pub) : "memory");
      }
      if (reading) {                                   // K-step 5: every wave has read what was published
        const unsigned got = (unsigned)__builtin_amdgcn_readfirstlane((int)mail);
        next = mode == 2 ? (tile + G < ntiles ? tile + G : -1) : (int)got;
        has_next = next >= 0;
        if (has_next) origin(next, pAn, pBn, m0n, n0n);
      }
      if (SPLIT && !issuer && !(DBG & 1)) {             // waves 4-7: their MFMAs are done, the A tile of the step after next goes out now
        if (in2) stage(pA + (int64_t)(kt + 2) * BK, oA, a.lda, s4);
        else if (has_next) stage(pAn + (int64_t)(kt + 2 - nk) * BK, oA, a.lda, s4);
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      sA = sA + 2 >= RING ? sA + 2 - RING : sA + 2;
    }

    // ---- epilogue: accumulators -> per-wave LDS patch (fp32, XOR-swizzled) -> 128-byte row segments -------------
    // The two ring slots of the step that just finished are free until the next step issues its DMA, and every wave
    // only talks to its own 4 KiB patch in each of them, so there is no workgroup barrier in here.  Lane layout in:
    // acc[i][j] = C[i*16 + frow][j*16 + 4*fkg .. +3]; out: lane l owns 8 consecutive columns (l & 7) * 8 of row
    // (l >> 3) + 8h, i.e. one store instruction writes 8 complete 128-byte lines.
    if ((DBG & 32) && !ROT) {
      // lab: direct epilogue -- no LDS transpose; a store instruction writes 16 rows x 64 contiguous bytes
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int64_t n = n0 + wn * 64 + 32 * jp + 8 * fkg;
        float bias[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bias[e] = a.bias ? a.bias[n + e] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = acc[i][2 * jp][e] + bias[e]; v[4 + e] = acc[i][2 * jp + 1][e] + bias[4 + e]; }
          nt_store_row8(a, m0 + wm * 128 + i * 16 + frow, n, v, true);
        }
      }
    } else if (!(DBG & 4)) {
      const int f3 = sA + 3 >= RING ? sA + 3 - RING : sA + 3, f4 = sA + 4 >= RING ? sA + 4 - RING : sA + 4;
      float* patch[2] = {reinterpret_cast<float*>(smem + f3 * T2_BYTES + wave * 4096),
                         reinterpret_cast<float*>(smem + f4 * T2_BYTES + wave * 4096)};
      const int orow = lane >> 3, oc = lane & 7;
      const int64_t n = n0 + wn * 64 + oc * 8;
      float bias[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bias[e] = a.bias ? a.bias[n + e] : 0.f;
      // patch I/O goes through inline asm: hipcc does not know the DMA ring is quiescent here and would put a
      // vmcnt(0) -- i.e. a full drain of the output stores -- in front of every plain LDS store
      const unsigned pw[2] = {lds_addr(patch[0]) + frow * 256, lds_addr(patch[1]) + frow * 256};
      // rotary epilogue (fused q|k|v projection): the lane's 8 columns sit at a fixed place c of a q or k head for the
      // whole tile, only the position m mod S changes.  vmcnt retires in order, so a table load issued after a
      // round's output stores would wait for those stores to be acknowledged: the tables of round r+1 are therefore
      // requested BEFORE the stores of round r go out.
      bool rot_on = false;
      const float *tabA = nullptr, *tabB = nullptr;
      if (ROT) {
        const int sec = (int)(n / a.rot_D);
        const int c = (int)((n - (int64_t)sec * a.rot_D) % a.rot_Dh);
        rot_on = sec < 2 && c < a.rot_R;
        tabA = (sec ? a.rot_ka : a.rot_qa) + c;
        tabB = (sec ? a.rot_kb : a.rot_qb) + c;
      }
      f32x4 ta[2][2], tb[2][2];                        // [h][half]
      auto load_tabs = [&](int i) {
        if (ROT && rot_on) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int64_t m = m0 + wm * 128 + i * 16 + orow + 8 * h;
            const int64_t o = (int64_t)(m % a.rot_S) * a.rot_R;
            ta[h][0] = *reinterpret_cast<const f32x4*>(tabA + o);
            ta[h][1] = *reinterpret_cast<const f32x4*>(tabA + o + 4);
            tb[h][0] = *reinterpret_cast<const f32x4*>(tabB + o);
            tb[h][1] = *reinterpret_cast<const f32x4*>(tabB + o + 4);
          }
        }
      };
      load_tabs(0);
      // residual epilogue: a round's 2 x 16 bytes per lane are requested one round ahead, BEFORE the previous round's stores go out
      // (a load issued behind stores waits for their acknowledgement: vmcnt retires in order) -- fetched inline they cost a full
      // memory round trip per round, +25 % (text) / +38 % (vision) on the residual GEMMs of the step
      bf16x8 res_cur[2] = {}, res_nxt[2] = {};
      auto load_res = [&](int i, bf16x8 (&r)[2]) {
        if (!ROT && !EXT && a.residual) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
            r[h] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(a.residual + (m0 + wm * 128 + i * 16 + orow + 8 * h) * a.ldr + n));
        }
      };
      load_res(0, res_cur);
      // extended epilogue: per-row factor, residual, and a second tile operand with a per-row coefficient.  One register
      // set: a round's operands are consumed (both rows finished into `outv`), THEN the next round's are requested into
      // the same registers, THEN the round's stores go out -- loads never queue behind stores (vmcnt retires in order)
      bf16x8 xres[2] = {}, xsub[2] = {};
      float xrs[2] = {1.f, 1.f}, xkc[2] = {0.f, 0.f};
      f32x4 xbr[2][2] = {};                            // the broadcast residual's 8 columns: reloaded only when the group changes
      auto load_ext = [&](int i) {
        if (EXT) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int64_t m = m0 + wm * 128 + i * 16 + orow + 8 * h;
            if (a.row_scale) xrs[h] = a.row_scale[m];
            if (a.residual) xres[h] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(a.residual + m * a.ldr + n));
            if (a.sub) {
              xsub[h] = *reinterpret_cast<const bf16x8*>(a.sub + m * a.ldsub + n);
              xkc[h] = a.sub_coef[m];
            }
            if (a.bres) {
              const float* bp = a.bres + (m / a.bres_rows) * a.N + n;
              xbr[h][0] = *reinterpret_cast<const f32x4*>(bp);
              xbr[h][1] = *reinterpret_cast<const f32x4*>(bp + 4);
            }
          }
        }
      };
      load_ext(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int u = i & 1;                           // alternate between the two free slots
#pragma unroll
        for (int j = 0; j < 4; ++j)
          asm volatile("ds_write_b128 %0, %1" ::"v"(pw[u] + (((j * 4 + fkg) ^ frow) << 4)), "v"(acc[i][j]) : "memory");
        if (EXT && !ROT) {
          bf16x8 outv[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int r = orow + 8 * h;
            const int64_t m = m0 + wm * 128 + i * 16 + r;
            const unsigned base = lds_addr(patch[u]) + r * 256;
            f32x4 lo, hi;
            asm volatile("ds_read_b128 %0, %1" : "=v"(lo) : "v"(base + (((2 * oc) ^ r) << 4)) : "memory");
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(hi) : "v"(base + (((2 * oc + 1) ^ r) << 4)) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(e < 4 ? lo[e] : hi[e - 4], xrs[h], bias[e]);
            if (a.preact) {
              bf16x8 pz;
#pragma unroll
              for (int e = 0; e < 8; ++e) pz[e] = (bf16)v[e];
              *reinterpret_cast<bf16x8*>(a.preact + m * a.ldc + n) = pz;
            }
            if (a.epilogue & MEANT_EPI_GELU) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = gelu_erf_fast(v[e]);
            }
            if (a.residual) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += (float)xres[h][e];
            }
            if (a.sub) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = fmaf(-xkc[h], (float)xsub[h][e], v[e]);
            }
            if (a.bres) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = fmaf(a.bres_scale, e < 4 ? xbr[h][0][e] : xbr[h][1][e - 4], v[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) outv[h][e] = (bf16)v[e];
          }
          __builtin_amdgcn_sched_barrier(0);
          if (i + 1 < 8) load_ext(i + 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int h = 0; h < 2; ++h)
            *reinterpret_cast<bf16x8*>(a.C + (m0 + wm * 128 + i * 16 + orow + 8 * h) * a.ldc + n) = outv[h];
        } else if (!ROT) {
          f32x4 lo[2], hi[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int r = orow + 8 * h;
            const unsigned base = lds_addr(patch[u]) + r * 256;
            asm volatile("ds_read_b128 %0, %1" : "=v"(lo[h]) : "v"(base + (((2 * oc) ^ r) << 4)) : "memory");
            asm volatile("ds_read_b128 %0, %1" : "=v"(hi[h]) : "v"(base + (((2 * oc + 1) ^ r) << 4)) : "memory");
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          if (i + 1 < 8) load_res(i + 1, res_nxt);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int r = orow + 8 * h;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (e < 4 ? lo[h][e] : hi[h][e - 4]) + bias[e];
            if (DBG & 16) { asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7])); }   // lab: everything but the global store
            else nt_store_row8<(DBG & 64) != 0>(a, m0 + wm * 128 + i * 16 + r, n, v, true, a.residual ? &res_cur[h] : nullptr);
          }
          res_cur[0] = res_nxt[0];
          res_cur[1] = res_nxt[1];
        } else {
          // finish both rows, THEN ask for the next round's tables (into the same registers), THEN store
          bf16x8 outv[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int r = orow + 8 * h;
            const int64_t m = m0 + wm * 128 + i * 16 + r;
            const unsigned base = lds_addr(patch[u]) + r * 256;
            f32x4 lo, hi;
            asm volatile("ds_read_b128 %0, %1" : "=v"(lo) : "v"(base + (((2 * oc) ^ r) << 4)) : "memory");
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(hi) : "v"(base + (((2 * oc + 1) ^ r) << 4)) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (e < 4 ? lo[e] : hi[e - 4]) + bias[e];
            if (a.preact) {
              bf16x8 pz;
#pragma unroll
              for (int e = 0; e < 8; ++e) pz[e] = (bf16)v[e];
              *reinterpret_cast<bf16x8*>(a.preact + m * a.ldc + n) = pz;
            }
            if (rot_on) rot_apply8(v, ta[h][0], ta[h][1], tb[h][0], tb[h][1]);
            outv[h] = nt_finish_row8(a, m, n, v);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (i + 1 < 8) load_tabs(i + 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int h = 0; h < 2; ++h)
            *reinterpret_cast<bf16x8*>(a.C + (m0 + wm * 128 + i * 16 + orow + 8 * h) * a.ldc + n) = outv[h];
        }
      }
    }
    first_tile = false;
    if (!has_next) break;
    // the patches live in the ring slots the next step's DMA is about to fill: every wave must be out of them first
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    tile = next; pA = pAn; pB = pBn; m0 = m0n; n0 = n0n;
    if (dynamic) { next = -1; has_next = false; }
    else {
      next = fixed_next(next);
      has_next = next < ntiles;
      if (has_next) origin(next, pAn, pBn, m0n, n0n);
    }
  }
  leave();
}

// ------------------------------------------------------------------------------------------------
// Epilogue of one wave's 128 x 64 accumulator block of a 256 x 256 tile (the streaming kernels' per-wave patch epilogue as a
// function): accumulators -> the wave's two private 4 KiB LDS patches (fp32, XOR-swizzled, alternating per 16-row round) ->
// 128-byte row segments.  No workgroup barrier inside; the caller guarantees that nobody else touches the patches.
// Lane layout in: acc[i][j] = C[i*16 + frow][j*16 + 4*fkg .. +3]; out: lane l owns 8 consecutive columns (l & 7) * 8 of row
// (l >> 3) + 8h, i.e. one store instruction writes 8 complete 128-byte lines.  Patch I/O goes through inline asm: hipcc does
// not know the DMA ring is quiescent there and would put a vmcnt(0) -- a full drain of the output stores and of the prefetch in
// flight -- in front of every plain LDS store.
#ifdef PP_LAB_STAMP
// lab build only (tools/lab/stamp_pp.py): s_memtime sums of waves 0 (leader) and 4 (follower) per workgroup:
// 0 steps kt > 0 (without the wait) | 1 steps kt == 0 | 2 end-of-step vmcnt wait | 3 epilogue until the bias is there | 4 epilogue rounds |
// 5 barriers behind the epilogue | 6 steps | 7 tiles | 8 total ticks | 9 total s_memrealtime ticks (100 MHz)
constexpr int PP_LAB_MAX_WG = 256;
__device__ unsigned long long g_pp_stamp[PP_LAB_MAX_WG * 2 * 16];
__device__ __forceinline__ unsigned long long pp_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define PPSTAMP(i) { const unsigned long long n__ = pp_now(); lab_acc[i] += n__ - lab_t; lab_t = n__; }
#define PPSTAMP_ARGS , unsigned long long (&lab_acc)[10], unsigned long long& lab_t
#define PPSTAMP_PASS , lab_acc, lab_t
#else
#define PPSTAMP(i)
#define PPSTAMP_ARGS
#define PPSTAMP_PASS
#endif
// Epilogue modes: what the launcher knows about a launch is a TEMPLATE parameter, so that the eight rounds are straight-line code.
// With the options as run-time flags every `if (a.residual)` / `if (a.preact)` ... is a branch with loads on one side, and at each
// join hipcc waits for vmcnt(0): every round then waited for the previous round's output stores to be ACKNOWLEDGED by memory
// (2.5 k cycles per round, 20 k of a 64 k-cycle tile at K = 768: round 4, tools/lab/stamp_pp.py) -- a store is fire-and-forget
// only while nothing behind it asks for the counter.
enum { NTE_PLAIN = 0, NTE_RES, NTE_GELU_PRE, NTE_ROT, NTE_EXT, NTE_GENERIC };
// Output stores as write-through (sc1) stores: a plain store leaves its line in the XCD's 4 MiB L2 (MI355X_MICROARCH.md, "stores of
// each flavour"), and a tile's 128 KiB of output per CU push the A panels the neighbouring CUs are about to re-read out of it.
// Measured (profiles/r04_nt256p_hbm_traffic*.json): HBM traffic 1.175x -> 1.08x of the algorithmic bytes at N = K = 768, 1.72x ->
// 1.65x at N = 2304; +1-2 % throughput.  PP_STORE_SC1=0 builds the plain stores.
#ifndef PP_STORE_SC1
#define PP_STORE_SC1 1
#endif
#ifndef PP_ROT_WALK
#define PP_ROT_WALK 1
#endif
#define GAS __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ const GAS T* gp(const T* p) { return (const GAS T*)p; }
template <typename T> __device__ __forceinline__ GAS T* gpw(T* p) { return (GAS T*)p; }

template <int MODE>
__device__ __forceinline__ void nt256_wave_epilogue(const GemmBf16Args& a, f32x4 (&acc)[8][4], float* patch0, float* patch1, int64_t m0,
                                                    int64_t n0, int wm, int wn, int lane PPSTAMP_ARGS) {
  constexpr bool ROT = MODE == NTE_ROT, EXT = MODE == NTE_EXT, GEN = MODE == NTE_GENERIC || EXT;
  // compile-time constants in the specialised modes, run-time flags in the two generic ones
  const bool f_res = MODE == NTE_RES || (GEN && a.residual != nullptr);
  const bool f_pre = MODE == NTE_GELU_PRE || (GEN && a.preact != nullptr);
  const bool f_gelu = MODE == NTE_GELU_PRE || (GEN && (a.epilogue & MEANT_EPI_GELU));
  const bool f_sig = MODE == NTE_GENERIC && (a.epilogue & MEANT_EPI_SIGMOID);
  const int frow = lane & 15, fkg = lane >> 4;
  float* patch[2] = {patch0, patch1};
  const int orow = lane >> 3, oc = lane & 7;
  const int64_t n = n0 + wn * 64 + oc * 8;
  const int64_t mrow = m0 + wm * 128 + orow;          // this lane's row of round 0, h = 0
  float bias[8];
  if (a.bias) {
    const f32x4 b0 = *gp(reinterpret_cast<const f32x4*>(a.bias + n)), b1 = *gp(reinterpret_cast<const f32x4*>(a.bias + n + 4));
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = e < 4 ? b0[e] : b1[e - 4];
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = 0.f;
  }
  const unsigned pw[2] = {lds_addr(patch[0]) + frow * 256, lds_addr(patch[1]) + frow * 256};
  // rotary epilogue (fused q|k|v projection): the lane's 8 columns sit at a fixed place c of a q or k head for the whole tile,
  // only the position m mod S changes.  vmcnt retires in order, so a table load issued after a round's output stores would wait
  // for those stores to be acknowledged: the operands of round r+1 are requested BEFORE the stores of round r go out.
  // The tile lies inside ONE of the q | k | v sections (256 divides H * Dh = 768; the launcher checks it), so "does this tile rotate
  // at all" is wave-uniform; inside a rotating tile every lane requests table rows (a lane whose 8 columns lie beyond the rotary
  // dimension reads the row's first columns and drops them): no divergent branch around the requests.  The position m mod S of a
  // lane's rows is formed ONCE per tile and stepped: a 64-bit modulo per row was ~60 of a round's 440 instructions.
  bool tile_rot = false, rot_on = false;
  const float *tabA = nullptr, *tabB = nullptr;
  unsigned posb = 0;
  const unsigned rS = ROT ? (unsigned)a.rot_S : 1u;
  const bool bigS = rS >= 256u;                        // then a lane's 16 rows of a tile wrap around S at most once
  if (ROT) {
    const int sec = __builtin_amdgcn_readfirstlane((int)((n0 + wn * 64) / a.rot_D));
    const int c = (int)((n - (int64_t)sec * a.rot_D) % a.rot_Dh);
    tile_rot = sec < 2;
    rot_on = tile_rot && c < a.rot_R;
#ifdef PP_LAB_ROT_BCAST
    tabA = (sec ? a.rot_ka : a.rot_qa);
    tabB = (sec ? a.rot_kb : a.rot_qb);
#else
    tabA = (sec ? a.rot_ka : a.rot_qa) + (rot_on ? c : 0);
    tabB = (sec ? a.rot_kb : a.rot_qb) + (rot_on ? c : 0);
#endif
    posb = (unsigned)(mrow % (int64_t)rS);
  }
  auto pos_of = [&](int i, int h) {
    const unsigned p = posb + (unsigned)(i * 16 + 8 * h);
    return bigS ? (p >= rS ? p - rS : p) : p % rS;
  };
  // Operand registers of a round.  In the rotary mode there are TWO sets and a round's operands are requested two rounds ahead
  // (a round is ~500 cycles, an L2 hit under this kernel's own load more): set i & 1 serves round i and is refilled for round
  // i + 2 as soon as round i has consumed it.  The residual mode requests all eight rounds' rows (64 registers) before the first
  // round: nothing is ever requested behind a store (see DEFER).  The extended mode has no registers to spare (253) and keeps one set, one
  // round ahead.  Either way a round's requests go out BEFORE its stores -- loads never queue behind stores (vmcnt retires in order).
  constexpr int NS = MODE == NTE_ROT ? 2 : MODE == NTE_RES ? 8 : 1;   // residual mode: all eight rounds' rows requested up front
  f32x4 ta[NS][2][2], tb[NS][2][2];                // [set][h][half]
  bf16x8 xres[NS][2] = {}, xsub[2] = {};
  float xrs[2] = {1.f, 1.f}, xkc[2] = {0.f, 0.f};
  f32x4 xbr[2][2] = {};
  auto load_ops = [&](int i, int st) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t m = mrow + i * 16 + 8 * h;
      if (ROT) {
        if (tile_rot) {
#ifdef PP_LAB_ROT_BCAST
          const unsigned o = (unsigned)__builtin_amdgcn_readfirstlane((int)(pos_of(i, h) * (unsigned)a.rot_R));   // lab: one address per wave (wrong values)
#else
          const unsigned o = pos_of(i, h) * (unsigned)a.rot_R;
#endif
          ta[st][h][0] = *gp(reinterpret_cast<const f32x4*>(tabA + o));
          ta[st][h][1] = *gp(reinterpret_cast<const f32x4*>(tabA + o + 4));
          tb[st][h][0] = *gp(reinterpret_cast<const f32x4*>(tabB + o));
          tb[st][h][1] = *gp(reinterpret_cast<const f32x4*>(tabB + o + 4));
        }
      } else {
        if (f_res) xres[st][h] = __builtin_nontemporal_load(gp(reinterpret_cast<const bf16x8*>(a.residual + m * a.ldr + n)));
        if (EXT) {
          if (a.row_scale) xrs[h] = *gp(a.row_scale + m);
          if (a.sub) {
            xsub[h] = *gp(reinterpret_cast<const bf16x8*>(a.sub + m * a.ldsub + n));
            xkc[h] = *gp(a.sub_coef + m);
          }
          if (a.bres) {
            const float* bp = a.bres + (m / a.bres_rows) * a.N + n;
            xbr[h][0] = *gp(reinterpret_cast<const f32x4*>(bp));
            xbr[h][1] = *gp(reinterpret_cast<const f32x4*>(bp + 4));
          }
        }
      }
    }
  };
#pragma unroll
  for (int i = 0; i < NS; ++i) load_ops(i, i);
#ifdef PP_LAB_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PPSTAMP(3)
#endif
  // output rows by running pointers (rows 8 apart: cstep8 elements): no 64-bit multiply per row
  const int64_t cstep8 = 8 * a.ldc;
  bf16* cp = a.C + mrow * a.ldc + n;
  bf16* pp = f_pre ? a.preact + mrow * a.ldc + n : nullptr;
  // vmcnt retires in order: a round's operand request issued behind an earlier round's output stores completes only when those
  // stores have been ACKNOWLEDGED by memory (microseconds under this kernel's own load) -- the rotary projection ran at 0.90 PFLOP/s
  // beside 1.14 for the plain epilogue, and neither fewer instructions per round (440 -> 200) nor one table address per wave
  // changed that.  In the rotary mode the finished rows are therefore HELD in registers (the accumulators drain twice as fast as
  // they fill: 16 registers freed, 8 taken per round) and stored after the last round's requests are out: 0.95-0.99 PFLOP/s.
  constexpr bool DEFER = MODE == NTE_ROT;
  bf16x8 held[DEFER ? 8 : 1][2];
  auto patch_write = [&](int i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      asm volatile("ds_write_b128 %0, %1" ::"v"(pw[i & 1] + (((j * 4 + fkg) ^ frow) << 4)), "v"(acc[i][j]) : "memory");
  };
  patch_write(0);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int u = i & 1;                           // the two patches alternate
    const int st = NS == 2 ? u : NS == 8 ? i : 0;
    // both rows of the round come back in ONE asm statement that also waits for them: hipcc does not know an asm ds_read is
    // asynchronous and may copy its destination before a wait that sits in a later statement (tools/isa_inflight_check.py)
    f32x4 lo[2], hi[2];
    {
      const unsigned b0 = lds_addr(patch[u]) + orow * 256, b1 = b0 + 8 * 256;
      const int r0 = orow, r1 = orow + 8;
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(lo[0]), "=&v"(hi[0]), "=&v"(lo[1]), "=&v"(hi[1])
                   : "v"(b0 + (((2 * oc) ^ r0) << 4)), "v"(b0 + (((2 * oc + 1) ^ r0) << 4)), "v"(b1 + (((2 * oc) ^ r1) << 4)),
                     "v"(b1 + (((2 * oc + 1) ^ r1) << 4))
                   : "memory");
    }
    // the NEXT round's accumulators go into the other patch now: their LDS round trip runs under this round's arithmetic
    if (i + 1 < 8) patch_write(i + 1);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 outv[2], prev[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaf(e < 4 ? lo[h][e] : hi[h][e - 4], EXT ? xrs[h] : 1.0f, bias[e]);
      if (f_pre) {
#pragma unroll
        for (int e = 0; e < 8; ++e) prev[h][e] = (bf16)v[e];
      }
      if (ROT && tile_rot) {
        float w8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w8[e] = v[e];
        rot_apply8(w8, ta[st][h][0], ta[st][h][1], tb[st][h][0], tb[st][h][1]);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rot_on ? w8[e] : v[e];
      }
      if (f_gelu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_erf_fast(v[e]);
      }
      if (f_sig) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
      }
      if (f_res) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)xres[st][h][e];
      }
      if (EXT) {
        if (a.sub) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaf(-xkc[h], (float)xsub[h][e], v[e]);
        }
        if (a.bres) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaf(a.bres_scale, e < 4 ? xbr[h][0][e] : xbr[h][1][e - 4], v[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) outv[h][e] = (bf16)v[e];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (i + NS < 8) load_ops(i + NS, st);
    __builtin_amdgcn_sched_barrier(0);
    if (DEFER) {                                   // rotary / residual modes: see `held`
      held[i][0] = outv[0];
      held[i][1] = outv[1];
      continue;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf16* const cq = cp + (h ? cstep8 : 0);
      bf16* const pq = f_pre ? pp + (h ? cstep8 : 0) : nullptr;
#ifndef PP_LAB_NOSTORE
#if PP_STORE_SC1
      // write-through stores (see PP_STORE_SC1).  The s_nop: the data registers of a 16-byte store may be rewritten right behind it
      // only after a wait state, which hipcc inserts for its own stores and cannot know about here
      if (f_pre) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(pq), "v"(prev[h]) : "memory");
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(cq), "v"(outv[h]) : "memory");
#else
      if (f_pre) *gpw(reinterpret_cast<bf16x8*>(pq)) = prev[h];
      *gpw(reinterpret_cast<bf16x8*>(cq)) = outv[h];
#endif
#else
      asm volatile("" ::"v"(outv[h]), "v"(prev[h]), "v"(cq), "v"(pq));
#endif
    }
    cp += 2 * cstep8;
    if (f_pre) pp += 2 * cstep8;
  }
  if (DEFER) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bf16* const cq = cp + (h ? cstep8 : 0);
#if PP_STORE_SC1
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(cq), "v"(held[i][h]) : "memory");
#else
        *gpw(reinterpret_cast<bf16x8*>(cq)) = held[i][h];
#endif
      }
      cp += 2 * cstep8;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the wave is out of its patches
  PPSTAMP(4)
}

// ------------------------------------------------------------------------------------------------
// Streaming NT kernel, PING-PONG form (round 4).  Same tile (256 x 256 x 64), same 8 waves (2 along M x 4 along N, 128 x 64 per
// wave), same five-slot ring, tile walk and per-wave patch epilogue as gemm_bf16_nt256s_kernel -- what changes is WHEN the two
// waves of a SIMD do what.  In the kernel above all eight waves run in lock step: both waves of a SIMD read fragments at the same
// time (matrix pipe idle) and then both queue MFMAs (LDS and the address unit idle); a K-step measures 1.65 us against 1.08 us
// of MFMA time.  Here the workgroup is two GROUPS of four waves, one wave per SIMD each: the LEADER (waves 0-3, rows 0..127 of the
// tile) and the FOLLOWER (waves 4-7, rows 128..255), which runs the same program ONE BARRIER INTERVAL behind.  A K-step is TWO
// phases (one 64-row half of the wave's block x all of K = 64: 32 MFMAs; four phases of 16 and one of 64 both measured slower); a phase is a
// LOAD half (the phase's fragment reads and its share of the prefetch stream's LDS-DMA pieces) and an MFMA half, with a workgroup
// barrier after each.  The one-interval lag puts every MFMA half of one group beside a LOAD half of the other: a SIMD's matrix
// pipe always has exactly one wave feeding it, and fragment reads, DMA issue and address arithmetic of its partner run underneath
// (cdna_hip_programming.md "The 256^2 8-phase template"; MI355X_MICROARCH.md "Two waves per SIMD", items 1, 5, 9).
// A wave issues one instruction every ~4-5 cycles, so a LOAD half hides only while it stays well under (MFMAs per phase) x 16 / 4.5
// instructions: the prefetch cursors are branch-free scalar selects, DMA pieces use the SGPR-base form (no per-piece VALU), and
// everything the epilogue needs is re-read from the kernarg segment per tile instead of living in SGPRs across the K-loop.
//
// Barrier intervals of K-step n, P = 2 (I = 4n + ...), L/M = load / MFMA half of phase p:
//     leader:    L0 M0 L1 M1          follower:  -- L0 M0 L1 | M1
// Prefetch stream (wave w owns pieces 4w .. 4w+3 of every tile: rows 32w .. 32w+31): the first half of a step's phases carries
// B of step n+1 (weights: L2 hits, needed at the next step's L0), the second half A of step n+2 (from HBM: more than a step of
// lead).  Each wave waits for its own pieces of step n+1 with a counted vmcnt (its four youngest requests, A of step n+2, stay
// in flight) just before the barrier in front of the leader's next L0: the leader at the end of its last M, the follower at the end
// of its last L.
// Slot reuse: B of step n+1 goes into the slot of A of step n-1, each wave into rows only its OWN group read (the leader's last
// reads of them are two intervals old, the follower's one and complete -- it has issued the MFMAs that needed them); A of step
// n+2 goes into the slot of B of step n-1, last read in the follower's first load half of that step.
// Tile end: the leader runs its epilogue in the interval in which the follower issues its last MFMAs and then runs its own (the
// two epilogues overlap), one barrier, then the leader's L0 of the next tile while the follower idles one interval.  The patches
// live in the two slots of the tile's last step, each wave inside rows only its own group read; the DMA into those slots starts
// behind that barrier.  After the last tile the cursors keep prefetching (valid rows of the last tile, never read): no branch
// in the K-loop distinguishes it.
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds) {   // sbase, lds: wave-uniform
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
}
// the four pieces (1 KiB each, consecutive in LDS) a wave owns of one tile.  PP_DMA_IMM: ONE M0 write for the group; the
// instruction's immediate offset moves source and destination alike, so piece p's lane offset is built as v[p] - 1024 p (the
// launch code adds 3072 to every v[] and subtracts it from the base: the 32-bit lane offset is unsigned)
#ifndef PP_DMA_IMM
#define PP_DMA_IMM 1
#endif
__device__ __forceinline__ void glds16_x4(const void* sbase, const unsigned (&v)[4], unsigned lds) {
#if PP_DMA_IMM
  asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %4\n\tglobal_load_lds_dwordx4 %1, %4 offset:1024\n\t"
               "global_load_lds_dwordx4 %2, %4 offset:2048\n\tglobal_load_lds_dwordx4 %3, %4 offset:3072"
               ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "s"((const char*)sbase - 3072), "s"(lds) : "memory");
#else
#pragma unroll
  for (int p = 0; p < 4; ++p) glds16_s(sbase, v[p], lds + p * 1024);
#endif
}
typedef const __attribute__((address_space(4))) unsigned* kernarg_words_t;
static_assert(sizeof(GemmBf16Args) % 4 == 0, "kernarg copy by words");

#ifdef PP_LAB_STAMP2
// second lab build (tools/lab/stamp_pp.py with PROBE_FINE=1), P = 2 only: s_memtime sums per half phase of waves 0 and 4:
// 0 L0 | 1 barrier behind L0 | 2 M0 | 3 barrier behind M0 | 4 L1 (+ follower's wait) | 5 barrier behind L1 | 6 M1 | 7 rest of the step up
// to the next L0 (leader's wait, cursors, barrier) | 8 steps.  Stamps taken while LDS reads are in flight are not waited for on the
// spot (the wait would serialise behind the reads): they are read behind the next lgkmcnt(0) the code has anyway.
__device__ unsigned long long g_pp_stamp2[256 * 2 * 16];
#define PP2_NOW(v) asm volatile("s_memtime %0" : "=s"(v)::"memory")
#define PP2_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define PP2_BEGIN() { PP2_NOW(f_a); PP2_SYNC(); f_acc[7] += f_a - f_e; }
#define PP2_BEGIN2(i) { PP2_NOW(f_a); PP2_SYNC(); f_acc[i] += f_a - f_e; }
#define PP2_L_END() { PP2_NOW(f_b); }
#define PP2_M_BEGIN(i) { PP2_NOW(f_c); PP2_SYNC(); f_acc[i] += f_b - f_a; f_acc[i + 1] += f_c - f_b; }
#define PP2_M_END(i) { PP2_NOW(f_e); PP2_SYNC(); f_acc[i] += f_e - f_c; }
#else
#define PP2_BEGIN()
#define PP2_BEGIN2(i)
#define PP2_L_END()
#define PP2_M_BEGIN(i)
#define PP2_M_END(i)
#endif
template <int MODE>
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt256p_kernel(GemmBf16Args a, int ntm, int ntn, TileSched* __restrict__ sched, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const bool leader = wm == 0;
  const int nk = (int)(a.K / BK);
  const int ntiles = ntm * ntn, G = gridDim.x;
  const int CH = G >> 3, xcd = blockIdx.x & 7;
  int tile = xcd * CH + (blockIdx.x >> 3);
  const bool dynamic = sched != nullptr && nk >= 8;
  auto leave = [&]() {
    if (sched && threadIdx.x == 0) {
      if (atomicAdd(&sched->done, 1u) == (unsigned)G - 1u) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sched->next[i] = 0;
        sched->done = 0;
      }
    }
  };
  if (tile >= ntiles) { leave(); return; }
  auto tile_of = [&](int x, unsigned k) { return (int)((k / (unsigned)CH) * (unsigned)G + (unsigned)(x * CH) + k % (unsigned)CH); };

  // DMA: piece q (1 KiB = 8 rows of 128 B) of a 32-piece tile; wave w owns pieces 4w + p, p = 0..3.  Row r of the tile keeps its
  // 16-byte chunk c at slot c ^ ((r >> 1) & 7): for r = 32w + 8p + (lane >> 3) that is (lane & 7) ^ (lane >> 4 & 3) ^ 4 (p & 1).
  const int rl = lane >> 3;
  const int c0 = (lane & 7) ^ (rl >> 1);
  unsigned vA[4], vB[4];                               // byte offsets of this lane's 16 bytes of piece p from the tile's origin
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    vA[p] = (unsigned)(((32 * wave + 8 * p + rl) * a.lda + (c0 ^ ((p & 1) << 2)) * 8) * 2) + (PP_DMA_IMM ? 3072 - 1024 * p : 0);
    vB[p] = (unsigned)(((32 * wave + 8 * p + rl) * a.ldb + (c0 ^ ((p & 1) << 2)) * 8) * 2) + (PP_DMA_IMM ? 3072 - 1024 * p : 0);
  }
  const unsigned lds0 = lds_addr(smem) + wave * 4096;  // this wave's first piece inside slot 0
  // Rotary mode, sequences of P = S / 256 > 1 row panels (mode bits 16..: P; the launcher checks that P divides the panel count):
  // the walk takes the panels of one position range after the other (all panels p with p mod P == 0, then == 1, ...) instead of in
  // memory order.  A tile's epilogue reads the 256 table rows of its positions (98 KB at R = 48); an XCD's 32 workgroups turn its
  // 4 MiB L2 over every ~3 us, and in memory order the ~3.5 panels an XCD has in flight cover all position ranges, so a table row is
  // touched again only every ~3 us per (range, section) -- it was evicted every time: 1.86 GB per launch of table rows from beyond L2
  // at (786432, 2304, 768), HBM-side traffic 2.05x the algorithmic bytes (profiles/r04_nt256p_hbm_traffic.json).  With one range
  // at a time the working set halves and every tile touches it.
  const int rotP = MODE == NTE_ROT ? (mode >> 16) : 1;
  const int rotQ = MODE == NTE_ROT && rotP > 1 ? ntm / rotP : 1;
  auto origin = [&](int t, const bf16*& pa, const bf16*& pb, int64_t& m0, int64_t& n0) {
    int tm = t / ntn;
    const int tn = t - tm * ntn;
    if (MODE == NTE_ROT && rotP > 1) { const int r = tm / rotQ; tm = (tm - r * rotQ) * rotP + r; }
    m0 = (int64_t)tm * B2;
    m0 = m0 + B2 <= a.M ? m0 : a.M - B2;               // ragged M: the last row tile is moved up to END at row M (see the launcher)
    n0 = (int64_t)tn * B2;
    pa = a.A + m0 * a.lda;
    pb = a.B + n0 * a.ldb;
  };

  const bf16 *pA, *pB, *pAn, *pBn;                     // origins of this tile and of the next one (the same when there is none)
  int64_t m0, n0, m0n, n0n;
  origin(tile, pA, pB, m0, n0);
  int next = dynamic ? -1 : tile + G;
  bool has_next = !dynamic && next < ntiles;
  pAn = pA; pBn = pB; m0n = m0; n0n = n0;
  if (has_next) origin(next, pAn, pBn, m0n, n0n);
  unsigned* mailbox = sched ? &sched->mailbox[blockIdx.x] : nullptr;

  // Start stagger (mode bit 8).  All workgroups start together and a tile takes every CU the same time, so all 256 CUs reach their
  // epilogues -- 32 MiB of output stores -- in the same few microseconds and then sit in the store queue (measured: 20 k of a tile's
  // 64 k cycles at K = 768) while HBM writes idle for the rest of the tile.  Workgroup j of the G / 8 on its XCD therefore starts
  // j / (G / 8) of a tile time late: the bursts of different CUs fall beside other CUs' K-loops.  The tiles are handed out by counters,
  // so a late starter simply ends up with fewer of them.
  if (mode & 0x100) {
    const int j = blockIdx.x >> 3;                     // position inside the XCD's group of CH workgroups
    // a K-step is ~1.5 us ~ 3000 cycles, an epilogue ~ 4 of them; s_sleep 127 ~ 8128 cycles
    const int naps = (int)(((int64_t)(nk + 4) * 3000 * j / CH) >> 13);
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(127);
  }
  // prologue: A_0, B_0, A_1 (four pieces per wave and tile); the first two must have landed
  glds16_x4(pA, vA, lds0);
  glds16_x4(pB, vB, lds0 + T2_BYTES);
  glds16_x4(pA + BK, vA, lds0 + 2 * T2_BYTES);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (!leader) {                                       // the follower's one-interval lag
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  // prefetch cursors: what the B pieces (one step ahead) and the A pieces (two steps ahead) of the CURRENT step read
  const bf16* qB = pB + BK;
  const bf16* qA = pA + 2 * BK;

#define PP_BAR()                          \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)

#ifdef PP_LAB_STAMP
  unsigned long long lab_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long lab_t = pp_now();
  const unsigned long long lab_t0 = lab_t, lab_r0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef PP_LAB_STAMP2
  unsigned long long f_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long f_a = 0, f_b = 0, f_c = 0, f_e = 0;
  PP2_NOW(f_e); PP2_SYNC();
#endif
  const int frow = lane & 15, fkg = lane >> 4;
  // fragment read offsets inside a slot: row i*16 + frow, chunk (ks*4 + fkg) ^ ((row >> 1) & 7); i adds 2048 bytes, ks flips bit 6
  const unsigned fo0 = (unsigned)(frow * 128 + ((fkg ^ (frow >> 1)) << 4));
  const unsigned foA[2] = {fo0 + wm * 16384, (fo0 ^ 64u) + wm * 16384};
  const unsigned foB[2] = {fo0 + wn * 8192, (fo0 ^ 64u) + wn * 8192};
  // ---- per-step scalars, software-pipelined: the values of step n+1 are computed in step n's LAST load half (beside the partner
  // group's MFMAs) and pinned in front of its barrier, so that nothing but the moves sits between a step's last barrier and the next
  // step's first DMA piece.  ring slot of A_n: sA; B_n sits in the next one, B_{n+1} goes to sA + 3, A_{n+2} to sA + 4 (mod 5).
  int kspecial = dynamic ? 2 : -1;                     // the draw: request in step 2, published behind it, read in step 4
  int sA = 0;
  unsigned aoff = 0, boff = T2_BYTES, dB = lds0 + 3 * T2_BYTES, dA = lds0 + 4 * T2_BYTES;
  for (;;) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < nk; ++kt) {
      // tile draw, part 1 (see TileSched): requests only; they are older than this step's DMA pieces, so the counted wait that
      // ends the step covers them.  hipcc does not know the register is in flight until then, and wherever control flow joins it
      // is free to copy a register (it did: a v_mov of the ticket at a branch join, found by tools/isa_inflight.py) -- so both
      // requests are STRAIGHT-LINE code: one asm statement each, executed in every step, that narrows EXEC to the lanes that
      // really ask (none in most steps: the instruction is then skipped) and restores it; one fresh register per step takes
      // whichever answer the step has.  The load halves stay short: a dozen scalar instructions per step.
      unsigned req;
      asm volatile("" : "=v"(req));                     // "defined" without an instruction
      const bool special = kt == kspecial;              // kspecial: 2, then 4, then never (-1 without the dynamic hand-out)
      const bool own = !((mode & 0xff) == 3 && xcd != 0);
      const bool drawing = special && kt == 2 && wave == 4;
      {
        // EXEC masks as 32-bit halves that are provably scalar (a 64-bit select of constants ends up on the VALU)
        const int m_draw = __builtin_amdgcn_readfirstlane(drawing && own ? 1 : 0);       // lane 0 of wave 4
        const int m_read = __builtin_amdgcn_readfirstlane(special && kt == 4 ? -1 : 0);  // every lane of every wave
        unsigned long long keep;
        asm volatile("s_mov_b64 %1, exec\n\ts_mov_b32 exec_lo, %2\n\ts_mov_b32 exec_hi, 0\n\tglobal_atomic_add %0, %3, %4, off sc0\n\ts_mov_b64 exec, %1"
                     : "+v"(req), "=&s"(keep) : "s"(m_draw), "v"(sched ? &sched->next[xcd] : nullptr), "v"(1u) : "memory");
        asm volatile("s_mov_b64 %1, exec\n\ts_mov_b32 exec_lo, %2\n\ts_mov_b32 exec_hi, %2\n\tglobal_load_dword %0, %3, off sc1\n\ts_mov_b64 exec, %1"
                     : "+v"(req), "=&s"(keep) : "s"(m_read), "v"(mailbox) : "memory");
      }
      const bool drawer = drawing && lane == 0;

      bf16x8 aq[2][4], bq[2][4];
      auto ldA = [&](int ih, int ks, bf16x8 (&d)[4]) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) d[ii] = *reinterpret_cast<const bf16x8*>(smem + aoff + foA[ks] + (ih * 4 + ii) * 2048);
      };
      auto ldB = [&](int ks, bf16x8 (&d)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = *reinterpret_cast<const bf16x8*>(smem + boff + foB[ks] + j * 2048);
      };
      auto mm = [&](int ih, const bf16x8 (&av)[4], const bf16x8 (&bv)[4]) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[ih * 4 + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv[j], av[ii], acc[ih * 4 + ii][j], 0, 0, 0);   // swapped: acc = C^T tile
      };
      auto wait_step = [&]() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); };
      // ---- phase 0: rows 0..63 of the wave's block, both k-halves ------------------------------------------------------
      PP2_BEGIN()
      glds16_x4(qB, vB, dB);
      ldB(0, bq[0]); ldB(1, bq[1]);
      ldA(0, 0, aq[0]); ldA(0, 1, aq[1]);
      PP2_L_END()
      PP_BAR();
      PP2_M_BEGIN(0)
      __builtin_amdgcn_s_setprio(1);
      mm(0, aq[0], bq[0]); mm(0, aq[1], bq[1]);
      __builtin_amdgcn_s_setprio(0);
      PP2_M_END(2)
      PP_BAR();
      // ---- phase 1: rows 64..127 --------------------------------------------------------------------------------------
      PP2_BEGIN2(3)
      glds16_x4(qA, vA, dA);
      ldA(1, 0, aq[0]); ldA(1, 1, aq[1]);
      // the next step's scalars (see above).  Cursors: one K-block on, or into the next tile when the step they feed is one (B: step
      // kt + 2 of this tile does not exist; A: step kt + 3).  pAn / pBn are final by then: the draw is read behind step 4 and
      // the dynamic hand-out needs nk >= 8.
      int n_sA = sA + 2 >= RING ? sA + 2 - RING : sA + 2;
      const int n_sB = n_sA + 1 == RING ? 0 : n_sA + 1;
      const int n_s3 = n_sA + 3 >= RING ? n_sA + 3 - RING : n_sA + 3, n_s4 = n_sA + 4 >= RING ? n_sA + 4 - RING : n_sA + 4;
      unsigned n_aoff = (unsigned)n_sA * T2_BYTES, n_boff = (unsigned)n_sB * T2_BYTES;
      unsigned n_dB = lds0 + (unsigned)n_s3 * T2_BYTES, n_dA = lds0 + (unsigned)n_s4 * T2_BYTES;
      const bf16* n_qB = kt + 2 == nk ? pBn : qB + BK;
      const bf16* n_qA = kt + 3 == nk ? pAn : qA + BK;
      asm volatile("" : "+s"(n_sA), "+s"(n_aoff), "+s"(n_boff), "+s"(n_dB), "+s"(n_dA), "+s"(n_qB), "+s"(n_qA));
      PP2_L_END()
      __builtin_amdgcn_sched_barrier(0);
#ifdef PP_LAB_STAMP
      if (!leader) { if (kt) PPSTAMP(0) else PPSTAMP(1) }
#endif
      if (!leader) wait_step();                        // follower: its pieces of the next step must be in before the leader's L0
#ifdef PP_LAB_STAMP
      if (!leader) PPSTAMP(2)
#endif
      PP_BAR();
      PP2_M_BEGIN(4)
      __builtin_amdgcn_s_setprio(1);
      mm(1, aq[0], bq[0]); mm(1, aq[1], bq[1]);
      __builtin_amdgcn_s_setprio(0);
      PP2_M_END(6)
      __builtin_amdgcn_sched_barrier(0);
#ifdef PP_LAB_STAMP
      if (leader) { if (kt) PPSTAMP(0) else PPSTAMP(1) }
#endif
      if (leader) wait_step();
#ifdef PP_LAB_STAMP
      if (leader) PPSTAMP(2)
      lab_acc[6] += 1;
#endif
#ifdef PP_LAB_STAMP2
      f_acc[8] += 1;
#endif
      // the follower's last MFMAs of a tile and its epilogue share one interval (see the header)
      if (leader || kt + 1 < nk) PP_BAR();
      sA = n_sA; aoff = n_aoff; boff = n_boff; dB = n_dB; dA = n_dA; qB = n_qB; qA = n_qA;
      if (special) {
        // tile draw, part 2: the answer is here (the step's counted wait covered it)
        asm volatile("" : "+v"(req) : : "memory");
        if (drawer) {
          int id = own ? tile_of(xcd, (unsigned)CH + req) : ntiles;
          if (id >= ntiles) {
            unsigned seen[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) seen[x] = __hip_atomic_load(&sched->next[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int att = 1; id >= ntiles && att < 8; ++att) {
              const int x2 = (xcd + att) & 7;
              if (tile_of(x2, (unsigned)CH + seen[x2]) >= ntiles) continue;
              id = tile_of(x2, (unsigned)CH + atomicAdd(&sched->next[x2], 1u));
              if (id < ntiles) atomicAdd(&sched->steals, 1u);
            }
          }
          const unsigned pub = id < ntiles ? (unsigned)id : 0xffffffffu;
          asm volatile("global_store_dword %0, %1, off sc1" ::"v"(mailbox), "v"(pub) : "memory");
        }
        if (kt == 4) {
          const unsigned got = (unsigned)__builtin_amdgcn_readfirstlane((int)req);
          next = (mode & 0xff) == 2 ? (tile + G < ntiles ? tile + G : -1) : (int)got;
          has_next = next >= 0;
          if (has_next) origin(next, pAn, pBn, m0n, n0n);
        }
        kspecial = kt == 2 ? 4 : -1;
      }
    }
    kspecial = dynamic ? 2 : -1;

    // ---- epilogue: the two slots of the tile's last step are free; each wave's patches lie inside rows only its own group read.
    // Its arguments come fresh from the kernarg segment (the pointer is laundered so that hipcc cannot keep them in SGPRs
    // across the K-loop, where they spill into VGPR lanes and cost the load halves v_readlane / s_nop pairs).
    {
      const int f3 = sA + 3 >= RING ? sA + 3 - RING : sA + 3, f4 = sA + 4 >= RING ? sA + 4 - RING : sA + 4;
      kernarg_words_t ka = (kernarg_words_t)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(ka));
      unsigned kw[sizeof(GemmBf16Args) / 4];
#pragma unroll
      for (unsigned i = 0; i < sizeof(GemmBf16Args) / 4; ++i) kw[i] = ka[i];
      GemmBf16Args ea;
      __builtin_memcpy(&ea, kw, sizeof(GemmBf16Args));
      nt256_wave_epilogue<MODE>(ea, acc, reinterpret_cast<float*>(smem + f3 * T2_BYTES + wave * 4096),
                                    reinterpret_cast<float*>(smem + f4 * T2_BYTES + wave * 4096), m0, n0, wm, wn, lane PPSTAMP_PASS);
    }
    PP_BAR();
#ifdef PP_LAB_STAMP
    lab_acc[7] += 1;
#endif
    if (!has_next) break;
    if (!leader) PP_BAR();                             // the follower idles through the leader's first L0
    PPSTAMP(5)
    tile = next; pA = pAn; pB = pBn; m0 = m0n; n0 = n0n;
    if (dynamic) { next = -1; has_next = false; }
    else {
      next = next + G;
      has_next = next < ntiles;
      if (has_next) origin(next, pAn, pBn, m0n, n0n);
    }
  }
#undef PP_BAR
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the cursors' last (unused) pieces
#ifdef PP_LAB_STAMP2
  if ((wave == 0 || wave == 4) && lane == 0 && blockIdx.x < 256)
    for (int i = 0; i < 9; ++i) g_pp_stamp2[(blockIdx.x * 2 + wm) * 16 + i] = f_acc[i];
#endif
#ifdef PP_LAB_STAMP
  lab_acc[8] = pp_now() - lab_t0;
  lab_acc[9] = __builtin_amdgcn_s_memrealtime() - lab_r0;
  if ((wave == 0 || wave == 4) && lane == 0 && blockIdx.x < PP_LAB_MAX_WG)
    for (int i = 0; i < 10; ++i) g_pp_stamp[(blockIdx.x * 2 + wm) * 16 + i] = lab_acc[i];
#endif
  leave();
}

// ------------------------------------------------------------------------------------------------
// TN kernel: dW[n][k] += sum_m dY[m][n] X[m][k].  LDS tiles are [64 m][128 cols] (256-byte rows).
constexpr int TN_BKM = 64;
constexpr int TN_TILE_BYTES = TN_BKM * 128 * 2;    // 16 KiB

// piece i (1 KiB) covers tile rows 4i..4i+3 (16 lanes per 256-byte row); wave w issues pieces 4w..4w+3.
__device__ __forceinline__ void tn_stage(const bf16* __restrict__ g, int64_t ld, int64_t m0, int64_t mend, int64_t col0,
                                          int64_t ncols, char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i;
    const int r = piece * 4 + (lane >> 4);
    const int slot = lane & 15;
    const int c = slot ^ ((r & 3) << 2);
    const int64_t gm = m0 + r;                     // always < mend: the launcher hands over whole 64-row tiles only
    int64_t gc = col0 + c * 8;
    gc = gc + 8 <= ncols ? gc : (ncols >= 8 ? ncols - 8 : 0);     // column clamp (those outputs are discarded)
    glds16(g + gm * ld + gc, lds_tile + piece * 1024);
  }
}

// fragment for mfma_32x32x16: lane (col = lane&31, h = lane>>5) needs rows 16*ks + 8h + j (j = 0..7) of column
// col0 + col: two transposed reads of 4 consecutive rows each.
__device__ __forceinline__ bf16x8 tn_frag(const char* tile, int ks, int col0, int lane) {
  const int h = lane >> 5;
  const int i16 = lane & 15;                       // lane within its 16-lane group
  const int q = i16 >> 2, p = i16 & 3;
  const int colblk = col0 + ((lane >> 4) & 1) * 16; // groups 0/1 (and 2/3) take adjacent 16-column blocks
  const int cbyte = (colblk + 4 * p) * 2;          // byte offset of this lane's 4 columns inside the row
  const int c16 = cbyte >> 4, sub = cbyte & 15;
  bf16x8 out;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int r = 16 * ks + 8 * h + 4 * t + q;
    const int off = r * 256 + ((c16 ^ ((r & 3) << 2)) << 4) + sub;
    const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(tile + off));
    out[4 * t + 0] = v[0]; out[4 * t + 1] = v[1]; out[4 * t + 2] = v[2]; out[4 * t + 3] = v[3];
  }
  return out;
}

__global__ __launch_bounds__(256, 2) void gemm_bf16_tn_kernel(const bf16* __restrict__ dY, int64_t lddy, const bf16* __restrict__ X,
                                                               int64_t ldx, float* __restrict__ dW, int64_t M, int64_t N,
                                                               int64_t K, int ntn, int ntk, int64_t rows_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);   // a split's tiles run on one XCD and share its L2
  const int tile = bid % (ntn * ntk);
  const int split = bid / (ntn * ntk);
  const int tn = tile / ntk, tk = tile - tn * ntk;
  const int64_t n0 = (int64_t)tn * 128, k0 = (int64_t)tk * 128;
  const int64_t mbeg = (int64_t)split * rows_per_split;
  const int64_t mend = (mbeg + rows_per_split < M) ? mbeg + rows_per_split : M;
  if (mbeg >= mend) return;
  const int nt = (int)((mend - mbeg + TN_BKM - 1) / TN_BKM);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  tn_stage(dY, lddy, mbeg, mend, n0, N, smem, wave, lane);
  tn_stage(X, ldx, mbeg, mend, k0, K, smem + TN_TILE_BYTES, wave, lane);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * 2 * TN_TILE_BYTES;
    char* nxt = smem + ((t + 1) & 1) * 2 * TN_TILE_BYTES;
    if (t + 1 < nt) {
      tn_stage(dY, lddy, mbeg + (int64_t)(t + 1) * TN_BKM, mend, n0, N, nxt, wave, lane);
      tn_stage(X, ldx, mbeg + (int64_t)(t + 1) * TN_BKM, mend, k0, K, nxt + TN_TILE_BYTES, wave, lane);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = tn_frag(cur, ks, wn * 64 + i * 32, lane);
        bfr[i] = tn_frag(cur + TN_TILE_BYTES, ks, wk * 64 + i * 32, lane);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // C layout of 32x32: col = lane&31 (k index), row = (e&3) + 8*(e>>2) + 4*(lane>>5) (n index): every
  // wave-instruction adds two 128-byte row segments -> the full-rate shape for global float atomics.
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t kcol = k0 + wk * 64 + j * 32 + (lane & 31);
      if (kcol >= K) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t nrow = n0 + wn * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (nrow < N) atomicAdd(dW + nrow * K + kcol, acc[i][j][e]);
      }
    }
}


// ------------------------------------------------------------------------------------------------
// TN kernel, 256 x 256 output tile, 8 waves (2 along n x 4 along k), 128 x 64 per wave, 64 token rows per
// stage ([64 m][256] tiles, 512-byte rows, 32 KiB each, 2 stages = 128 KiB).  Same load-path argument as
// the 256 x 256 NT kernel.  The bias gradient (column sums of dY) is accumulated from the dY fragments
// the k-tile-0 blocks already hold -- no separate pass over dY.
constexpr int TN2_TILE = TN_BKM * 256 * 2;         // 32 KiB

__device__ __forceinline__ void tn256_stage(const bf16* __restrict__ g, int64_t ld, int64_t m0, int64_t mend, int64_t col0,
                                             char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i;                // 1 KiB = 2 rows of 512 B
    const int r = piece * 2 + (lane >> 5);
    const int slot = lane & 31;
    const int c = slot ^ ((r & 3) << 2);
    const int64_t gm = m0 + r;                     // always < mend (whole 64-row tiles only)
    glds16(g + gm * ld + col0 + c * 8, lds_tile + piece * 1024);
  }
}

// per-lane byte offset (inside a [64][256] tile) of the transposed-read address for the 32-column block at col0:
// row 8h + q of the first k-step, this lane's 4 columns; k-step ks / half t add (16 ks + 4 t) * 512 as an immediate.
__device__ __forceinline__ unsigned tn256_lane_off(int col0, int lane) {
  const int h = lane >> 5, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
  const int cbyte = (col0 + ((lane >> 4) & 1) * 16 + 4 * p) * 2;
  const int r = 8 * h + q;                          // (r & 3) == q for every (ks, t)
  return (unsigned)(r * 512 + (((cbyte >> 4) ^ (q << 2)) << 4) + (cbyte & 15));
}
template <int KS>
__device__ __forceinline__ void tn256_frag_issue(unsigned addr, u32x2& lo, u32x2& hi) {
  lo = lds_read_tr16<KS * 8192>(addr);
  hi = lds_read_tr16<KS * 8192 + 2048>(addr);
}

// DET: instead of float atomics on dW / dbias (whose order differs from run to run) every workgroup stores its 256 x 256 partial
// tile to `part` [split][tile][256][256] and its bias partials to `pbias` [split][tk][wk][N]; tn256_reduce_kernel then adds them
// up in a fixed order.  Same MFMA stream, so the partial sums themselves are bit-identical between runs.
template <bool DET>
__global__ __launch_bounds__(512, 2) void gemm_bf16_tn256_kernel(const bf16* __restrict__ dY, int64_t lddy, const bf16* __restrict__ X,
                                                                  int64_t ldx, float* __restrict__ dW, float* __restrict__ dbias,
                                                                  int64_t M, int64_t N, int64_t K, int ntn, int ntk,
                                                                  int64_t rows_per_split, float* __restrict__ part,
                                                                  float* __restrict__ pbias) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 2, wk = wave & 3;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = bid % (ntn * ntk);
  const int split = bid / (ntn * ntk);
  const int tn = tile / ntk, tk = tile - tn * ntk;
  const int64_t n0 = (int64_t)tn * 256, k0 = (int64_t)tk * 256;
  const int64_t mbeg = (int64_t)split * rows_per_split;
  const int64_t mend = (mbeg + rows_per_split < M) ? mbeg + rows_per_split : M;
  if (mbeg >= mend) return;
  const int nt = (int)((mend - mbeg + TN_BKM - 1) / TN_BKM);
  // fused dbias: column sums of dY from the A fragments.  The conversions and adds are VALU work the MFMA stream has
  // to wait for, so they are dealt out evenly: of the ntk workgroups that read the same dY tile, the one with
  // tk == t mod ntk takes step t, and inside it wave wk takes k-sub-step ks == wk (everybody ends with atomics).
  const bool do_bias = dbias != nullptr;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  unsigned aoff[4], boff[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = tn256_lane_off(wn * 128 + i * 32, lane);
#pragma unroll
  for (int j = 0; j < 2; ++j) boff[j] = tn256_lane_off(wk * 64 + j * 32, lane);

  tn256_stage(dY, lddy, mbeg, mend, n0, smem, wave, lane);
  tn256_stage(X, ldx, mbeg, mend, k0, smem + TN2_TILE, wave, lane);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * 2 * TN2_TILE;
    char* nxt = smem + ((t + 1) & 1) * 2 * TN2_TILE;
    // as in the NT kernel: waves 0-3 issue all DMA pieces, their SIMD partners (waves 4-7) go straight to the MFMAs
    if (t + 1 < nt && __builtin_amdgcn_readfirstlane(threadIdx.x) < 256) {
      tn256_stage(dY, lddy, mbeg + (int64_t)(t + 1) * TN_BKM, mend, n0, nxt, wave, lane);
      tn256_stage(X, ldx, mbeg + (int64_t)(t + 1) * TN_BKM, mend, k0, nxt + TN2_TILE, wave, lane);
      tn256_stage(dY, lddy, mbeg + (int64_t)(t + 1) * TN_BKM, mend, n0, nxt, wave + 4, lane);
      tn256_stage(X, ldx, mbeg + (int64_t)(t + 1) * TN_BKM, mend, k0, nxt + TN2_TILE, wave + 4, lane);
    }
    const unsigned cbase = lds_addr(cur);
    const bool bias_step = (t % ntk) == tk;
    // fragment reads are software-pipelined one k-step ahead: while the MFMAs of k-step ks run, the 12 transposed
    // reads of k-step ks+1 are in flight; the wait before the MFMAs is a counted lgkmcnt(12)
    u32x2 alo[2][4], ahi[2][4], blo[2][2], bhi[2][2];
    auto issue = [&](auto KS) {
      constexpr int ks = decltype(KS)::value;
#pragma unroll
      for (int i = 0; i < 4; ++i) tn256_frag_issue<ks>(cbase + aoff[i], alo[ks & 1][i], ahi[ks & 1][i]);
#pragma unroll
      for (int j = 0; j < 2; ++j) tn256_frag_issue<ks>(cbase + TN2_TILE + boff[j], blo[ks & 1][j], bhi[ks & 1][j]);
    };
    auto compute = [&](auto KS) {
      constexpr int ks = decltype(KS)::value;
      bf16x8 af[4], bfr[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = pack_tr(alo[ks & 1][i], ahi[ks & 1][i]);
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = pack_tr(blo[ks & 1][j], bhi[ks & 1][j]);
      if (do_bias && ks == wk && bias_step) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) csum[i] += (float)af[i][e];
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
    issue(std::integral_constant<int, 0>{});
    issue(std::integral_constant<int, 1>{});
    lds_wait_upto<12>();
    compute(std::integral_constant<int, 0>{});
    issue(std::integral_constant<int, 2>{});
    lds_wait_upto<12>();
    compute(std::integral_constant<int, 1>{});
    issue(std::integral_constant<int, 3>{});
    lds_wait_upto<12>();
    compute(std::integral_constant<int, 2>{});
    lds_wait_upto<0>();
    compute(std::integral_constant<int, 3>{});
    __syncthreads();
  }
  float* ptile = DET ? part + ((int64_t)split * (ntn * ntk) + tile) * 65536 : nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kl = wk * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int nl = wn * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (DET) ptile[nl * 256 + kl] = acc[i][j][e];
        else atomicAdd(dW + (n0 + nl) * K + k0 + kl, acc[i][j][e]);
      }
    }
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float s2 = csum[i] + __shfl_xor(csum[i], 32, 64);
      if (lane < 32) {
        const int64_t n = n0 + wn * 128 + i * 32 + lane;
        if (DET) pbias[(((int64_t)split * ntk + tk) * 4 + wk) * N + n] = s2;
        else atomicAdd(dbias + n, s2);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// dW kernel, PING-PONG form (round 4): the structure of gemm_bf16_nt256p_kernel on the tiles, fragments and epilogue of
// gemm_bf16_tn256_kernel.  Output tile 256 (n) x 256 (k), waves 2 (n) x 4 (k), 128 x 64 per wave in 4 x 2 accumulators of
// v_mfma_f32_32x32x16_bf16; a STEP is 64 token rows ([64 m][256] tiles of dY and X, 512-byte rows, transposed fragment reads), a
// phase is one half of them (k-steps 0-1: token rows 0..31; k-steps 2-3: rows 32..63): 24 ds_read_b64_tr_b16 + 4 LDS-DMA pieces in
// its LOAD half, 16 MFMAs (512 cycles) in its MFMA half.  The leader (waves 0-3: n rows 0..127) and the follower (waves 4-7) run one
// barrier interval apart, so one wave of every SIMD is always in an MFMA half.
// Ring of five 32 KiB slots: dY_t, X_t, dY_t+1, X_t+1 (landing during step t), dY_t+2 (landing).  Every wave reads all 64 token rows of
// both tiles (its own columns), so a slot of step t-1 is refilled by halves: token rows 0..31 were last read in the follower's first
// load half of step t-1 and are refilled by waves 0-3 (pieces 4w + p = rows 8w + 2p ..) from the leader's L0 of step t on; rows
// 32..63 were last read in the follower's second load half and are refilled by waves 4-7, whose L0 of step t comes behind the
// barrier that follows those reads' completion.
// Steps past the end of the split keep prefetching its last rows (never read): no branch in the loop distinguishes them.
template <bool DET>
__global__ __launch_bounds__(512, 2) void gemm_bf16_tn256p_kernel(const bf16* __restrict__ dY, int64_t lddy, const bf16* __restrict__ X,
                                                                   int64_t ldx, float* __restrict__ dW, float* __restrict__ dbias,
                                                                   int64_t M, int64_t N, int64_t K, int ntn, int ntk,
                                                                   int64_t rows_per_split, float* __restrict__ part,
                                                                   float* __restrict__ pbias) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 2, wk = wave & 3;
  const bool leader = wn == 0;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  // (integer division runs on the VALU: say explicitly that the quotients are wave-uniform, they end up as SGPR operands of asm)
  const int tile = __builtin_amdgcn_readfirstlane(bid % (ntn * ntk));
  const int split = __builtin_amdgcn_readfirstlane(bid / (ntn * ntk));
  const int tn = tile / ntk, tk = tile - tn * ntk;
  const int64_t n0 = (int64_t)tn * 256, k0 = (int64_t)tk * 256;
  const int64_t mbeg = (int64_t)split * rows_per_split;
  const int64_t mend = (mbeg + rows_per_split < M) ? mbeg + rows_per_split : M;
  if (mbeg >= mend) return;
  const int nt = __builtin_amdgcn_readfirstlane((int)((mend - mbeg + TN_BKM - 1) / TN_BKM));
  const bool do_bias = dbias != nullptr;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  unsigned aoff[4], boff[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = tn256_lane_off(wn * 128 + i * 32, lane);
#pragma unroll
  for (int j = 0; j < 2; ++j) boff[j] = tn256_lane_off(wk * 64 + j * 32, lane);

  // DMA: piece q (1 KiB = 2 token rows of 512 B) of a 32-piece tile; wave w owns pieces 4w + p, i.e. rows 8w + 2p + (lane >> 5).
  // 16-byte chunk c of row r sits at slot c ^ (4 (r & 3)), and r & 3 = 2 (p & 1) + (lane >> 5).
  unsigned vY[4], vX[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = 8 * wave + 2 * p + (lane >> 5);
    const int c = (lane & 31) ^ ((r & 3) << 2);
    vY[p] = (unsigned)((r * lddy + c * 8) * 2) + (PP_DMA_IMM ? 3072 - 1024 * p : 0);
    vX[p] = (unsigned)((r * ldx + c * 8) * 2) + (PP_DMA_IMM ? 3072 - 1024 * p : 0);
  }
  const unsigned lds0 = lds_addr(smem) + wave * 4096;
  const bf16* pY = dY + mbeg * lddy + n0;
  const bf16* pX = X + mbeg * ldx + k0;
  const int64_t stepY = (int64_t)TN_BKM * lddy, stepX = (int64_t)TN_BKM * ldx;
  // prologue: dY_0, X_0, dY_1 (the last step's rows again when the split has a single step)
  const bf16* qY = nt > 1 ? pY + stepY : pY;           // cursor of the dY pieces: two steps ahead of the step that issues them
  glds16_x4(pY, vY, lds0);
  glds16_x4(pX, vX, lds0 + T2_BYTES);
  glds16_x4(qY, vY, lds0 + 2 * T2_BYTES);
  const bf16* qX = nt > 1 ? pX + stepX : pX;           // cursor of the X pieces: one step ahead
  qY = nt > 2 ? qY + stepY : qY;
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (!leader) {                                       // the follower's one-interval lag
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
#define PP_BAR()                          \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)
  const unsigned sbase = lds_addr(smem);
  int sA = 0;
  unsigned ya = sbase, xa = sbase + T2_BYTES, dXs = lds0 + 3 * T2_BYTES, dYs = lds0 + 4 * T2_BYTES;
  for (int t = 0; t < nt; ++t) {
    const bool bias_step = do_bias && (t % ntk) == tk;
    u32x2 alo[2][4], ahi[2][4], blo[2][2], bhi[2][2];
    // the 24 transposed reads of one phase (k-steps 2 ph and 2 ph + 1), inline asm: nothing may touch the destinations before the
    // lgkmcnt(0) at the top of the MFMA half (tools/isa_inflight_check.py)
    auto frag_reads = [&](auto PH) {
      constexpr int ph = decltype(PH)::value;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        alo[0][i] = lds_read_tr16<(2 * ph) * 8192>(ya + aoff[i]);
        ahi[0][i] = lds_read_tr16<(2 * ph) * 8192 + 2048>(ya + aoff[i]);
        alo[1][i] = lds_read_tr16<(2 * ph + 1) * 8192>(ya + aoff[i]);
        ahi[1][i] = lds_read_tr16<(2 * ph + 1) * 8192 + 2048>(ya + aoff[i]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        blo[0][j] = lds_read_tr16<(2 * ph) * 8192>(xa + boff[j]);
        bhi[0][j] = lds_read_tr16<(2 * ph) * 8192 + 2048>(xa + boff[j]);
        blo[1][j] = lds_read_tr16<(2 * ph + 1) * 8192>(xa + boff[j]);
        bhi[1][j] = lds_read_tr16<(2 * ph + 1) * 8192 + 2048>(xa + boff[j]);
      }
    };
    auto mfma_half = [&](int ph) {
      lds_wait_all();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 af[4], bfr[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = pack_tr(alo[s2][i], ahi[s2][i]);
#pragma unroll
        for (int j = 0; j < 2; ++j) bfr[j] = pack_tr(blo[s2][j], bhi[s2][j]);
        // fused dbias: column sums of dY from the A fragments, dealt out over the k-tiles' workgroups (step t belongs to the one
        // with tk == t mod ntk) and over a workgroup's waves (k-sub-step ks belongs to wave wk == ks).  The adds of fragment i sit
        // BEHIND its two MFMAs (which took their operands at issue) and in front of the next pair: an MFMA holds the matrix pipe for
        // 32 cycles and the issue port for 8 of them, so a dozen vector instructions per pair ride along
        const bool bias_now = bias_step && 2 * ph + s2 == wk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (bias_now) {
#pragma unroll
            for (int e = 0; e < 8; ++e) csum[i] += (float)af[i][e];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
    };
    // ---- phase 0: token rows 0..31 ----------------------------------------------------------------------------------------
    glds16_x4(qX, vX, dXs);
    frag_reads(std::integral_constant<int, 0>{});
    PP_BAR();
    mfma_half(0);
    PP_BAR();
    // ---- phase 1: token rows 32..63 ---------------------------------------------------------------------------------------
    glds16_x4(qY, vY, dYs);
    frag_reads(std::integral_constant<int, 1>{});
    // the next step's scalars, computed here (beside the partner group's MFMAs) and pinned in front of the barrier
    int n_sA = sA + 2 >= RING ? sA + 2 - RING : sA + 2;
    const int n_sB = n_sA + 1 == RING ? 0 : n_sA + 1;
    const int n_s3 = n_sA + 3 >= RING ? n_sA + 3 - RING : n_sA + 3, n_s4 = n_sA + 4 >= RING ? n_sA + 4 - RING : n_sA + 4;
    unsigned n_ya = sbase + (unsigned)n_sA * T2_BYTES, n_xa = sbase + (unsigned)n_sB * T2_BYTES;
    unsigned n_dX = lds0 + (unsigned)n_s3 * T2_BYTES, n_dY = lds0 + (unsigned)n_s4 * T2_BYTES;
    const bf16* n_qX = t + 2 < nt ? qX + stepX : qX;
    const bf16* n_qY = t + 3 < nt ? qY + stepY : qY;
    asm volatile("" : "+s"(n_sA), "+s"(n_ya), "+s"(n_xa), "+s"(n_dX), "+s"(n_dY), "+s"(n_qX), "+s"(n_qY));
    __builtin_amdgcn_sched_barrier(0);
    if (!leader) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // follower: its pieces of the next step must be in before the leader's L0
    PP_BAR();
    mfma_half(1);
    __builtin_amdgcn_sched_barrier(0);
    if (leader) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    PP_BAR();
    sA = n_sA; ya = n_ya; xa = n_xa; dXs = n_dX; dYs = n_dY; qX = n_qX; qY = n_qY;
  }
  if (leader) PP_BAR();                                // pairs with the barrier behind the follower's last MFMA half
#undef PP_BAR
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the cursors' last (unused) pieces
  float* ptile = DET ? part + ((int64_t)split * (ntn * ntk) + tile) * 65536 : nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kl = wk * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int nl = wn * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (DET) ptile[nl * 256 + kl] = acc[i][j][e];
        else atomicAdd(dW + (n0 + nl) * K + k0 + kl, acc[i][j][e]);
      }
    }
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float s2 = csum[i] + __shfl_xor(csum[i], 32, 64);
      if (lane < 32) {
        const int64_t n = n0 + wn * 128 + i * 32 + lane;
        if (DET) pbias[(((int64_t)split * ntk + tk) * 4 + wk) * N + n] = s2;
        else atomicAdd(dbias + n, s2);
      }
    }
  }
}

// (Measured and not kept, round 4: the same kernel on v_mfma_f32_16x16x32_bf16 -- the shape the chip is said to hold a higher clock on --
// with swapped operands and a patch epilogue whose atomics cover 256 contiguous bytes: 2-3 % SLOWER at all six shapes of the step
// (1.08 / 1.11 / 1.14 against 1.10 / 1.14 / 1.18 PFLOP/s at (N, K) = (768, 768) / (2304, 768) / (768, 2304)); twice the MFMA
// instructions for the same fragment traffic.  The source is kept under tools/lab/gemm_bf16_tn256q_16x16x32.hip.txt.)
// ordered reduction of the deterministic dW path: dW[n][k] += sum_s part[s][tile(n,k)][n%256][k%256] (s ascending), and
// dbias[n] += sum over (s, tk, wk) ascending of pbias.  One thread per 4 consecutive k.
__global__ __launch_bounds__(256) void tn256_reduce_kernel(const float* __restrict__ part, const float* __restrict__ pbias,
                                                            float* __restrict__ dW, float* __restrict__ dbias, int64_t N, int64_t K,
                                                            int ntk, int ntiles, int splits) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nq = N * K / 4;
  if (q < nq) {
    const int64_t n = (q * 4) / K, k = (q * 4) % K;
    const int64_t tile = (n >> 8) * ntk + (k >> 8);
    const float* src = part + tile * 65536 + (n & 255) * 256 + (k & 255);
    f32x4 s = *reinterpret_cast<const f32x4*>(src);
    for (int sp = 1; sp < splits; ++sp) s += *reinterpret_cast<const f32x4*>(src + (int64_t)sp * ntiles * 65536);
    f32x4* dst = reinterpret_cast<f32x4*>(dW + n * K + k);
    *dst = *dst + s;
  }
  if (dbias && q < N) {
    float s = 0.f;
    for (int i = 0; i < splits * ntk * 4; ++i) s += pbias[(int64_t)i * N + q];
    dbias[q] += s;
  }
}

}  // namespace

#ifdef PP_LAB_STAMP2
extern "C" int meant_lab_pp_stamps2(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_pp_stamp2), bytes < sizeof(g_pp_stamp2) ? bytes : sizeof(g_pp_stamp2));
}
#endif
#ifdef PP_LAB_STAMP
extern "C" int meant_lab_pp_stamps(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_pp_stamp), bytes < sizeof(g_pp_stamp) ? bytes : sizeof(g_pp_stamp));
}
#endif

// Scheduler slot of (current device, stream); nullptr when the table is full or the symbol cannot be resolved.
// g_tile_sched is a __device__ array: every device has its own copy at its own address.
static TileSched* tile_sched_for(hipStream_t stream) {
  struct PerDevice { TileSched* base = nullptr; bool tried = false; std::unordered_map<hipStream_t, int> slot; };
  static std::mutex mu;
  static PerDevice table[MEANT_MAX_DEVICES];
  const int dev = meant_current_device();
  if (dev < 0) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  PerDevice& pd = table[dev];
  if (!pd.tried) {
    pd.tried = true;
    if (hipGetSymbolAddress((void**)&pd.base, HIP_SYMBOL(g_tile_sched)) != hipSuccess) pd.base = nullptr;
  }
  if (!pd.base) return nullptr;
  auto it = pd.slot.find(stream);
  if (it != pd.slot.end()) return pd.base + it->second;
  if ((int)pd.slot.size() >= N_SCHED_SLOTS) return nullptr;
  const int idx = (int)pd.slot.size();
  pd.slot.emplace(stream, idx);
  return pd.base + idx;
}

// diagnostics for the tests: tiles that changed XCD in all streaming launches of the current device so far (synchronises)
extern "C" int64_t meant_debug_nt_steals(void) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  static TileSched host[N_SCHED_SLOTS];
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tile_sched), sizeof(host)) != hipSuccess) return -1;
  int64_t n = 0;
  for (int i = 0; i < N_SCHED_SLOTS; ++i) n += host[i].steals;
  return n;
}

int gemm_bf16_nt_launch(const GemmBf16Args& a, hipStream_t stream) {
  MEANT_REQUIRE(a.A && a.B && a.C, MEANT_ERR_ARG, "gemm_bf16_nt: null pointer");
  MEANT_REQUIRE(!a.rot_qa || (a.N == 3 * (int64_t)a.rot_D && a.rot_Dh % 8 == 0 && a.rot_R % 8 == 0 && (a.ldc & 7) == 0 && a.rot_D % a.rot_Dh == 0),
                MEANT_ERR_ARG, "gemm_bf16_nt: rotary epilogue needs N = 3*H*Dh, Dh %% 8 == 0, rot_dim %% 8 == 0");
  MEANT_REQUIRE(a.K % BK == 0, MEANT_ERR_UNSUPPORTED, "gemm_bf16_nt: K=%lld must be a multiple of %d (use the fp32 tier otherwise)", (long long)a.K, BK);
  MEANT_REQUIRE((a.lda % 8) == 0 && (a.ldb % 8) == 0 && meant_aligned16(a.A) && meant_aligned16(a.B), MEANT_ERR_ARG,
                "gemm_bf16_nt: operands must be 16-byte aligned with row strides that are multiples of 8");
  MEANT_RAISE_LDS(gemm_bf16_nt256_kernel, 4 * T2_BYTES);
  MEANT_RAISE_LDS((gemm_bf16_nt256s_kernel<0, false>), RING * T2_BYTES);
  MEANT_RAISE_LDS((gemm_bf16_nt256s_kernel<0, true>), RING * T2_BYTES);
  MEANT_RAISE_LDS((gemm_bf16_nt256s_kernel<0, false, true>), RING * T2_BYTES);
  MEANT_RAISE_LDS((gemm_bf16_nt256p_kernel<NTE_PLAIN>), RING * T2_BYTES);
  MEANT_RAISE_LDS((gemm_bf16_nt256p_kernel<NTE_RES>), RING * T2_BYTES);
  MEANT_RAISE_LDS((gemm_bf16_nt256p_kernel<NTE_GELU_PRE>), RING * T2_BYTES);
  MEANT_RAISE_LDS((gemm_bf16_nt256p_kernel<NTE_ROT>), RING * T2_BYTES);
  MEANT_RAISE_LDS((gemm_bf16_nt256p_kernel<NTE_EXT>), RING * T2_BYTES);
  MEANT_RAISE_LDS((gemm_bf16_nt256p_kernel<NTE_GENERIC>), RING * T2_BYTES);
  MEANT_RAISE_LDS((gemm_bf16_nt256s_kernel<128, false, false>), RING * T2_BYTES);
  MEANT_RAISE_LDS((gemm_bf16_nt256s_kernel<128, true, false>), RING * T2_BYTES);
  MEANT_RAISE_LDS((gemm_bf16_nt256s_kernel<128, false, true>), RING * T2_BYTES);
  const bool ext = a.row_scale || a.sub || a.bres;
  MEANT_REQUIRE(!a.bres || (a.bres_rows > 0 && (a.N & 7) == 0 && meant_aligned16(a.bres)), MEANT_ERR_ARG, "gemm_bf16_nt: bad broadcast residual");
  MEANT_REQUIRE(!ext || (!a.rot_qa && !(a.epilogue & MEANT_EPI_SIGMOID)), MEANT_ERR_UNSUPPORTED, "gemm_bf16_nt: extended epilogue with rotary / sigmoid");
  MEANT_REQUIRE(!a.sub || (a.sub_coef && (a.ldsub & 7) == 0 && meant_aligned16(a.sub)), MEANT_ERR_ARG, "gemm_bf16_nt: bad sub operand");
  // big tall problems: 256 x 256 tiles (half the operand bytes per FLOP) -- once there are enough of them to occupy at least
  // half the CUs (the temporal encoder's 1536^2 Linears make 36: four times as many 128 x 128 tiles finish in a third of the time)
  if (a.M >= 1024 && a.N % 256 == 0 && ceil_div(a.M, B2) * (a.N / B2) * 2 >= meant_num_cus()) {
    const int64_t ntm2 = ceil_div(a.M, B2), ntn2 = a.N / B2;
    MEANT_REQUIRE(ntm2 * ntn2 < 2147483647LL, MEANT_ERR_UNSUPPORTED, "gemm_bf16_nt: grid too large");
    // option nt_stream = 0 forces the one-tile-per-workgroup kernel (A/B measurements)
    const bool stream_ok = meant_opt(MEANT_OPT_NT_STREAM) != 0;
    // Ragged M, option nt_ragged = 1 (default): the streaming kernel takes all of it; its last row tile is moved up so that it
    // ENDS at row M and recomputes up to 255 rows of its neighbour.  Every element of C is a function of its own row of A
    // and column of W with a fixed K order, so both tiles store identical bits (no cost in the K-loop, no second launch:
    // the 32 trailing rows of the TimeSformer's 75 296 tokens cost 12 extra launches of 13-97 us per layer as a split).
    // Needs an out-of-place epilogue: a residual (or A) that aliases C would be read after the neighbour's store.
    const auto overlaps = [](const void* p, int64_t ld, const void* q, int64_t ldq, int64_t rows) {
      const char *p0 = (const char*)p, *q0 = (const char*)q;
      return p && q && p0 < q0 + rows * ldq * 2 && q0 < p0 + rows * ld * 2;
    };
    const bool ragged_overlap = stream_ok && meant_opt(MEANT_OPT_NT_RAGGED) != 0 && a.M % B2 != 0 && a.K >= 2 * BK && (a.ldc & 7) == 0 &&
                                (!a.residual || (a.ldr & 7) == 0) && !overlaps(a.C, a.ldc, a.residual, a.ldr, a.M) &&
                                !overlaps(a.C, a.ldc, a.sub, a.ldsub, a.M) &&
                                !overlaps(a.C, a.ldc, a.A, a.lda, a.M) && (!a.preact || !overlaps(a.preact, a.ldc, a.A, a.lda, a.M));
    // nt_ragged = 0 (or aliasing operands): the streaming kernel takes the first floor(M / 256) * 256 rows, the remaining < 256 rows go
    // to the 128 x 128 kernel as a second launch (row-local epilogues only: the rotary epilogue indexes its tables by the
    // absolute row, so it splits only where the boundary is a multiple of the sequence length).
    const int64_t m_full = (a.M / B2) * B2;
    if (!ragged_overlap && stream_ok && m_full >= 1024 && m_full != a.M && a.K >= 2 * BK && (a.ldc & 7) == 0 && (!a.residual || (a.ldr & 7) == 0) &&
        (!a.rot_qa || m_full % a.rot_S == 0)) {
      GemmBf16Args head = a, tail = a;
      head.M = m_full;
      tail.M = a.M - m_full;
      tail.A = a.A + m_full * a.lda;
      tail.C = a.C + m_full * a.ldc;
      if (a.residual) tail.residual = a.residual + m_full * a.ldr;
      if (a.preact) tail.preact = a.preact + m_full * a.ldc;
      if (a.row_scale) tail.row_scale = a.row_scale + m_full;
      if (a.sub) { tail.sub = a.sub + m_full * a.ldsub; tail.sub_coef = a.sub_coef + m_full; }
      MEANT_REQUIRE(!a.bres || m_full % a.bres_rows == 0, MEANT_ERR_UNSUPPORTED, "gemm_bf16_nt: broadcast residual across the head / tail split");
      if (a.bres) tail.bres = a.bres + (m_full / a.bres_rows) * a.N;
      meant_route_hit(ROUTE_NT_SPLIT);
      const int rc = gemm_bf16_nt_launch(head, stream);
      return rc ? rc : gemm_bf16_nt_launch(tail, stream);
    }
    if (stream_ok && (a.M % B2 == 0 || ragged_overlap) && a.K >= 2 * BK && (a.ldc & 7) == 0 && (!a.residual || (a.ldr & 7) == 0)) {
      if (ragged_overlap) meant_route_hit(ROUTE_NT_OVERLAP);
      int ncu = meant_num_cus() & ~7;
      const int cap = meant_opt(MEANT_OPT_NT_GRID_CAP) & ~7;     // tests: fewer workgroups => more tiles each, dry XCDs steal
      if (cap >= 8 && cap < ncu) ncu = cap;
      const int grid = (int)(ntm2 * ntn2 < ncu ? ((ntm2 * ntn2 + 7) / 8) * 8 : ncu);
      // Tile counters: one slot per (device, stream).  Launches on one stream execute in order and a launch leaves its
      // slot zeroed (last workgroup out), so a slot is never shared by two launches in flight -- by construction, not by
      // distance.  A stream beyond the table's capacity gets the fixed walk (sched = nullptr).  Option nt_dynamic = 0 forces it.
      const int dynmode0 = meant_opt(MEANT_OPT_NT_DYNAMIC);
      const int dynmode = dynmode0;
      TileSched* sched = (dynmode != 0 && dynmode != 4 && grid <= 512) ? tile_sched_for(stream) : nullptr;
      meant_route_hit(a.rot_qa ? ROUTE_NT256S_ROT : ROUTE_NT256S);
      const dim3 g3((unsigned)grid), b3(512);
      if (meant_opt(MEANT_OPT_NT_PP) != 0 && a.K >= 4 * BK) {   // ping-pong form: the two waves of a SIMD alternate between MFMA and load halves
        const int ppopt = meant_opt(MEANT_OPT_NT_PP);
        const int dynmode = dynmode0 | ((ppopt & 8) ? 0x100 : 0);   // lab, bit 3: staggered start
        // rotary mode: panels of one position range after the other (see the kernel's `origin`); PP_ROT_WALK=0 builds without it
        int rotP = 0;
        if (a.rot_qa && PP_ROT_WALK && a.rot_S % B2 == 0 && a.rot_S / B2 > 1 && a.rot_S / B2 < 256 && a.M % B2 == 0 && ntm2 % (a.rot_S / B2) == 0) rotP = (int)(a.rot_S / B2);
        // the epilogue's options as a template parameter (see nt256_wave_epilogue): the combinations the models run get straight-line
        // code, anything else the generic instantiation with run-time flags
        const bool act = (a.epilogue & (MEANT_EPI_GELU | MEANT_EPI_SIGMOID)) != 0;
        int emode = NTE_GENERIC;
        if (a.rot_qa) emode = (!a.residual && !a.preact && !act) ? NTE_ROT : -1;
        else if (ext) emode = NTE_EXT;
        else if (!a.residual && !a.preact && !act) emode = NTE_PLAIN;
        else if (a.residual && !a.preact && !act) emode = NTE_RES;
        else if (!a.residual && a.preact && (a.epilogue & MEANT_EPI_GELU) && !(a.epilogue & MEANT_EPI_SIGMOID)) emode = NTE_GELU_PRE;
        MEANT_REQUIRE(emode >= 0, MEANT_ERR_UNSUPPORTED, "gemm_bf16_nt: rotary epilogue with residual / activation / preact");
#define PP_LAUNCH(MODE_) hipLaunchKernelGGL((gemm_bf16_nt256p_kernel<MODE_>), g3, b3, RING * T2_BYTES, stream, a, (int)ntm2, (int)ntn2, sched, dynmode | (rotP << 16))
        switch (emode) {
          case NTE_PLAIN: PP_LAUNCH(NTE_PLAIN); break;
          case NTE_RES: PP_LAUNCH(NTE_RES); break;
          case NTE_GELU_PRE: PP_LAUNCH(NTE_GELU_PRE); break;
          case NTE_ROT: PP_LAUNCH(NTE_ROT); break;
          case NTE_EXT: PP_LAUNCH(NTE_EXT); break;
          default: PP_LAUNCH(NTE_GENERIC); break;
        }
#undef PP_LAUNCH
      } else if (meant_opt(MEANT_OPT_NT_SPLIT) != 0) {   // DMA issue split in time between the two waves of a SIMD (see the kernel)
        if (a.rot_qa) hipLaunchKernelGGL((gemm_bf16_nt256s_kernel<128, true>), g3, b3, RING * T2_BYTES, stream, a, (int)ntm2, (int)ntn2, sched, dynmode);
        else if (ext) hipLaunchKernelGGL((gemm_bf16_nt256s_kernel<128, false, true>), g3, b3, RING * T2_BYTES, stream, a, (int)ntm2, (int)ntn2, sched, dynmode);
        else hipLaunchKernelGGL((gemm_bf16_nt256s_kernel<128, false>), g3, b3, RING * T2_BYTES, stream, a, (int)ntm2, (int)ntn2, sched, dynmode);
      } else if (a.rot_qa) hipLaunchKernelGGL((gemm_bf16_nt256s_kernel<0, true>), g3, b3, RING * T2_BYTES, stream, a, (int)ntm2, (int)ntn2, sched, dynmode);
      else if (ext) hipLaunchKernelGGL((gemm_bf16_nt256s_kernel<0, false, true>), g3, b3, RING * T2_BYTES, stream, a, (int)ntm2, (int)ntn2, sched, dynmode);
      else hipLaunchKernelGGL((gemm_bf16_nt256s_kernel<0, false>), g3, b3, RING * T2_BYTES, stream, a, (int)ntm2, (int)ntn2, sched, dynmode);
    } else {
      meant_route_hit(ROUTE_NT256);
      hipLaunchKernelGGL(gemm_bf16_nt256_kernel, dim3((unsigned)(ntm2 * ntn2)), dim3(512), 4 * T2_BYTES, stream, a, (int)ntm2, (int)ntn2);
    }
    MEANT_LAUNCH_CHECK("gemm_bf16_nt256");
    return MEANT_OK;
  }
  const int64_t ntm = ceil_div(a.M, BM), ntn = ceil_div(a.N, BN);
  MEANT_REQUIRE(ntm * ntn < 2147483647LL, MEANT_ERR_UNSUPPORTED, "gemm_bf16_nt: grid too large");
  const size_t lds = 128 * (128 + 4) * sizeof(float);   // 67584 B: covers the 64 KiB of staging buffers too
  MEANT_RAISE_LDS(gemm_bf16_nt_kernel, lds);
  meant_route_hit(ROUTE_NT128);
  hipLaunchKernelGGL(gemm_bf16_nt_kernel, dim3((unsigned)(ntm * ntn)), dim3(256), lds, stream, a, (int)ntm, (int)ntn);
  MEANT_LAUNCH_CHECK("gemm_bf16_nt");
  return MEANT_OK;
}

static int tn_tail(const bf16* dY, int64_t lddy, const bf16* X, int64_t ldx, float* dW, float* dbias, int64_t M, int64_t N, int64_t K,
                   hipStream_t stream) {
  // fewer than 64 trailing token rows: exact generic kernel, accumulating into the same dW / dbias
  meant_route_hit(ROUTE_TN_TAIL);
  GemmF32Args g{};
  g.in_dtype = MEANT_BF16; g.out_dtype = MEANT_F32;
  g.A = dY; g.B = X; g.C = dW;
  g.M = N; g.N = K; g.K = M; g.nb1 = 1; g.nb2 = 1;
  g.sA[2] = 1; g.sA[3] = lddy; g.sB[2] = ldx; g.sB[3] = 1; g.sC[2] = K; g.sC[3] = 1;
  g.alpha = 1.f; g.accumulate = 1;
  int rc = gemm_f32_launch(g, stream);
  if (rc) return rc;
  if (dbias) return colsum_launch(dY, lddy, dbias, M, N, MEANT_BF16, 1, stream);
  return MEANT_OK;
}

// launch geometry of the 256 x 256 dW kernel (shared by the launcher and the workspace query)
static void tn256_geometry(int64_t M, int64_t N, int64_t K, int64_t& splits2, int64_t& rows2) {
  // one resident block per CU: size the launch to whole rounds of the CU count (a lone block in an extra
  // round would cost a full round of time) -- one round when every block still gets >= 32 tiles of 64 rows
  int ncu = meant_num_cus();
  if (ncu <= 0) ncu = 256;
  const int64_t tiles2 = (N / 256) * (K / 256);
  splits2 = ncu / tiles2;
  if (splits2 < 1) splits2 = 1;
  const int64_t max2 = ceil_div(M, 32 * TN_BKM);
  if (splits2 > max2) splits2 = max2;
  rows2 = ceil_div(ceil_div(M, splits2), TN_BKM) * TN_BKM;
  splits2 = ceil_div(M, rows2);
}
static bool tn256_ok(int64_t M, int64_t N, int64_t K) { return N % 256 == 0 && K % 256 == 0 && M >= 4096; }

size_t gemm_bf16_tn_ws(int64_t M, int64_t N, int64_t K) {
  if (!meant_opt(MEANT_OPT_DETERMINISTIC)) return 0;
  M -= M % TN_BKM;
  if (!tn256_ok(M, N, K)) return 0;
  int64_t splits2, rows2;
  tn256_geometry(M, N, K, splits2, rows2);
  return (size_t)(splits2 * N * K + splits2 * (K / 256) * 4 * N) * sizeof(float);
}

int gemm_bf16_tn_launch(const bf16* dY, int64_t lddy, const bf16* X, int64_t ldx, float* dW, float* dbias, int64_t M, int64_t N,
                        int64_t K, void* ws, size_t ws_bytes, hipStream_t stream) {
  const bool det = meant_opt(MEANT_OPT_DETERMINISTIC) != 0;
  MEANT_REQUIRE((lddy % 8) == 0 && (ldx % 8) == 0 && meant_aligned16(dY) && meant_aligned16(X), MEANT_ERR_ARG,
                "gemm_bf16_tn: operands must be 16-byte aligned with row strides that are multiples of 8");
  MEANT_REQUIRE(N >= 8 && K >= 8, MEANT_ERR_UNSUPPORTED, "gemm_bf16_tn: N and K must be >= 8");
  const int64_t Mtail = M % TN_BKM;
  if (Mtail) {
    const int64_t Mmain = M - Mtail;
    int rc = tn_tail(dY + Mmain * lddy, lddy, X + Mmain * ldx, ldx, dW, dbias, Mtail, N, K, stream);
    if (rc || Mmain == 0) return rc;
    M = Mmain;
  }
  if (tn256_ok(M, N, K)) {
    const int64_t ntn2 = N / 256, ntk2 = K / 256;
    int64_t splits2, rows2;
    tn256_geometry(M, N, K, splits2, rows2);
    const dim3 grid((unsigned)(ntn2 * ntk2 * splits2));
    const bool pp = meant_opt(MEANT_OPT_TN_PP) != 0;     // ping-pong form of the 256 x 256 dW kernel
    if (det) {
      const size_t need = (size_t)(splits2 * N * K + splits2 * ntk2 * 4 * N) * sizeof(float);
      MEANT_REQUIRE(ws && ws_bytes >= need && meant_aligned16(ws), MEANT_ERR_WORKSPACE,
                    "linear_bwd_dw (deterministic): workspace of %zu bytes needed (meant_linear_bwd_dw_ws), got %zu", need, ws_bytes);
      float* part = (float*)ws;
      float* pbias = part + splits2 * N * K;
      meant_route_hit(ROUTE_TN256_DET);
      if (pp) {
        MEANT_RAISE_LDS(gemm_bf16_tn256p_kernel<true>, RING * T2_BYTES);
        hipLaunchKernelGGL(gemm_bf16_tn256p_kernel<true>, grid, dim3(512), RING * T2_BYTES, stream, dY, lddy, X, ldx, dW, dbias, M, N, K, (int)ntn2,
                           (int)ntk2, rows2, part, pbias);
      } else {
        MEANT_RAISE_LDS(gemm_bf16_tn256_kernel<true>, 4 * TN2_TILE);
        hipLaunchKernelGGL(gemm_bf16_tn256_kernel<true>, grid, dim3(512), 4 * TN2_TILE, stream, dY, lddy, X, ldx, dW, dbias, M, N, K, (int)ntn2,
                           (int)ntk2, rows2, part, pbias);
      }
      MEANT_LAUNCH_CHECK("gemm_bf16_tn256<det>");
      hipLaunchKernelGGL(tn256_reduce_kernel, dim3((unsigned)ceil_div(N * K / 4, 256)), dim3(256), 0, stream, part, dbias ? pbias : nullptr, dW, dbias,
                         N, K, (int)ntk2, (int)(ntn2 * ntk2), (int)splits2);
      MEANT_LAUNCH_CHECK("tn256_reduce");
      return MEANT_OK;
    }
    meant_route_hit(ROUTE_TN256);
    if (pp) {
      MEANT_RAISE_LDS(gemm_bf16_tn256p_kernel<false>, RING * T2_BYTES);
      hipLaunchKernelGGL(gemm_bf16_tn256p_kernel<false>, grid, dim3(512), RING * T2_BYTES, stream, dY, lddy, X, ldx, dW, dbias, M, N, K, (int)ntn2,
                         (int)ntk2, rows2, (float*)nullptr, (float*)nullptr);
    } else {
      MEANT_RAISE_LDS(gemm_bf16_tn256_kernel<false>, 4 * TN2_TILE);
      hipLaunchKernelGGL(gemm_bf16_tn256_kernel<false>, grid, dim3(512), 4 * TN2_TILE, stream, dY, lddy, X, ldx, dW, dbias, M, N, K, (int)ntn2,
                         (int)ntk2, rows2, (float*)nullptr, (float*)nullptr);
    }
    MEANT_LAUNCH_CHECK("gemm_bf16_tn256");
    return MEANT_OK;
  }
  const int64_t ntn = ceil_div(N, 128), ntk = ceil_div(K, 128);
  // split the token axis so that the launch has >= ~4 blocks per CU; each split is a multiple of 64 rows
  int64_t splits = ceil_div(1024, ntn * ntk);
  const int64_t max_splits = ceil_div(M, 256);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1 || det) splits = 1;                 // deterministic: every output element has exactly one writer
  int64_t rows_per = ceil_div(ceil_div(M, splits), TN_BKM) * TN_BKM;
  splits = ceil_div(M, rows_per);
  MEANT_RAISE_LDS(gemm_bf16_tn_kernel, 4 * TN_TILE_BYTES);
  meant_route_hit(ROUTE_TN128);
  hipLaunchKernelGGL(gemm_bf16_tn_kernel, dim3((unsigned)(ntn * ntk * splits)), dim3(256), 4 * TN_TILE_BYTES, stream, dY, lddy, X,
                     ldx, dW, M, N, K, (int)ntn, (int)ntk, rows_per);
  MEANT_LAUNCH_CHECK("gemm_bf16_tn");
  if (dbias) return colsum_launch(dY, lddy, dbias, M, N, MEANT_BF16, 1, stream);
  return MEANT_OK;
}
