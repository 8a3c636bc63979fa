// Attention core for SHORT sequences (S <= 16), bf16 tier, head dims 64 / 96 / 128: one wave per (group, head), no workgroup
// barrier, no tile loop.  This is the time half of the divided space-time attention (src/meant/timesformer_pytorch.py:121-131:
// `b (f n) d -> (b n) f d`, 196 groups of 1 + 12 tokens per video): on the 128-query x 64-key tiles of attn_bf16.hip a
// 13 x 13 score block uses 1 % of the tile, and 75 k (group, head) items per step of 32 videos cost 2 ms per layer.
// Semantics are those of attn_bf16.hip (the reference's eager chain: scale, causal -inf, additive (1 - mask) * -1e9 key
// padding on the fp32 score, softmax, PV); the row statistics are the same (m, log2 l) pairs.
//
// Shapes and MFMA layouts (frow = lane & 15, fk = lane >> 4):
//   row fragments X[frow][32 ks + 8 fk .. +7] are the A and B operands of v_mfma_f32_16x16x32_bf16 as they come from
//   global memory (16 bytes per lane), so both S^T = K Q^T and S = Q K^T (and dP^T = V dO^T, dP = dO V^T) are one MFMA per
//   32 head columns with the two operands swapped; the result holds [4 fk + e][frow], e = 0..3.
//   That accumulator layout IS the B operand of v_mfma_f32_16x16x16_bf16 (k = 4 fk + e, column frow): P^T, dS^T, dS and P
//   go into the second products (O^T = V^T P^T, dQ^T = K^T dS^T, dK^T = Q^T dS, dV^T = dO^T P) straight from
//   registers.  Their A operands are TRANSPOSED 16 x 16 blocks (X[4 fk + e][16 db + frow]): the row fragments are parked in a
//   per-wave LDS image (rows of HD * 2 + 16 bytes: the four row groups of a 2-byte column read land on disjoint banks) and read
//   back with ds_read_u16.  A wave orders its own LDS traffic with a wavefront fence; nothing is shared between waves.
// Roofline: HBM.  Forward moves (3 + 1) * S * HD * 2 bytes per item, backward (3 + 1 + 3) * S * HD * 2; the arithmetic is
// noise (44 k MACs per item forward).
#include "internal.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr float LOG2E = 1.4426950408889634f;
constexpr float PADL2 = 1e9f * 1.4426950408889634f;   // (1 - mask) * -1e9 in log2 units (attn_bf16.hip)

struct ShortArgs {
  const bf16* qkv; bf16* o; float* lse; const float* key_mask;
  const bf16* dout; bf16* dqkv; RotTables rot;
  int64_t items;                                   // G * H
  int S, H; float scale; int causal;
};

template <int HD> struct ShortGeom {
  static constexpr int KS = HD / 32, DB = HD / 16, ROWB = HD * 2 + 16, MATB = 16 * ROWB;
};

// park a matrix's row fragments in the wave's LDS image: row frow, columns 32 ks + 8 fk .. +7
template <int HD>
__device__ __forceinline__ void park(char* img, const bf16x8 (&f)[HD / 32], int frow, int fk) {
#pragma unroll
  for (int ks = 0; ks < HD / 32; ++ks)
    *reinterpret_cast<bf16x8*>(img + frow * ShortGeom<HD>::ROWB + (32 * ks + 8 * fk) * 2) = f[ks];
}

// transposed 16 x 16 block db of a parked matrix as the A operand of the 16x16x16 MFMA: X[4 fk + e][16 db + frow]
template <int HD>
__device__ __forceinline__ s16x4 tr_block(const char* img, int db, int frow, int fk) {
  s16x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    v[e] = *reinterpret_cast<const short*>(img + (4 * fk + e) * ShortGeom<HD>::ROWB + (16 * db + frow) * 2);
  return v;
}

__device__ __forceinline__ s16x4 to_bf16x4(const float (&p)[4]) {
  bf16x4 b;
#pragma unroll
  for (int e = 0; e < 4; ++e) b[e] = (bf16)p[e];
  return __builtin_bit_cast(s16x4, b);
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float quad_max(float v) {   // over the four lanes that share frow
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// additive bias (log2 units) of key `key` of group g: -inf beyond the sequence, the padding term otherwise
__device__ __forceinline__ float key_bias(const float* km, int64_t g, int S, int key) {
  if (key >= S) return -INFINITY;
  return km ? fmaf(km[g * S + key], PADL2, -PADL2) : 0.f;
}

template <int HD>
__global__ __launch_bounds__(256) void attn_short_fwd_kernel(ShortArgs a) {
  using Gm = ShortGeom<HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t item = (int64_t)blockIdx.x * 4 + wave;
  if (item >= a.items) return;                      // whole wave; no workgroup barrier anywhere in this kernel
  const int64_t g = item / a.H;
  const int h = (int)(item - g * a.H);
  const int S = a.S, D = a.H * HD, frow = lane & 15, fk = lane >> 4;
  const int64_t ld = 3 * (int64_t)D;
  char* img = smem + wave * Gm::MATB;

  const int rowc = frow < S ? frow : S - 1;          // rows beyond the sequence re-read the last one (finite, never used)
  const bf16* rp = a.qkv + ((int64_t)g * S + rowc) * ld + h * HD + 8 * fk;
  bf16x8 qf[Gm::KS], kf[Gm::KS], vf[Gm::KS];
#pragma unroll
  for (int ks = 0; ks < Gm::KS; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8*>(rp + 32 * ks);
    kf[ks] = *reinterpret_cast<const bf16x8*>(rp + D + 32 * ks);
    vf[ks] = *reinterpret_cast<const bf16x8*>(rp + 2 * D + 32 * ks);
  }
  park<HD>(img, vf, frow, fk);

  f32x4 st = {0.f, 0.f, 0.f, 0.f};                   // S^T[key = 4 fk + e][query = frow]
#pragma unroll
  for (int ks = 0; ks < Gm::KS; ++ks) st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qf[ks], st, 0, 0, 0);

  const float c1 = a.scale * LOG2E;
  float t[4], m = -INFINITY;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int key = 4 * fk + e;
    float b = key_bias(a.key_mask, g, S, key);
    if (a.causal && key > frow) b = -INFINITY;
    t[e] = fmaf(st[e], c1, b);
    m = fmaxf(m, t[e]);
  }
  m = quad_max(m);                                   // finite for every real query: key 0 exists, is visible, has a finite bias
  float p[4], l = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { p[e] = __builtin_amdgcn_exp2f(t[e] - m); l += p[e]; }
  l = quad_sum(l);
  const s16x4 pt = to_bf16x4(p);
  const float inv_l = 1.0f / l;

  if (fk == 0 && frow < S) {
    float* lp = a.lse + (((int64_t)g * a.H + h) * S + frow) * 2;
    lp[0] = m;
    lp[1] = __log2f(l);
  }
  wave_lds_sync();
  bf16* op = a.o + ((int64_t)g * S + frow) * D + h * HD + 4 * fk;
#pragma unroll
  for (int db = 0; db < Gm::DB; ++db) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 ot = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(tr_block<HD>(img, db, frow, fk), pt, z, 0, 0, 0);   // O^T[16 db + 4 fk + e][frow]
    bf16x4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = (bf16)(ot[e] * inv_l);
    if (frow < S) *reinterpret_cast<bf16x4*>(op + 16 * db) = ov;
  }
}

// adjoint of the rotary map on four consecutive head columns d .. d+3 of one row (pairs (2j, 2j+1); attn_bf16.hip
// rotary_adjoint_regs): dt[2j] = A[2j] d[2j] + B[2j+1] d[2j+1];  dt[2j+1] = A[2j+1] d[2j+1] - B[2j] d[2j]
__device__ __forceinline__ f32x4 rot_adj4(f32x4 v, const float* A, const float* B, int R, int pos, int d) {
  if (d >= R) return v;                              // R % 8 == 0 and d % 4 == 0: the four columns are in or out together
  const f32x4 av = *reinterpret_cast<const f32x4*>(A + (int64_t)pos * R + d);
  const f32x4 bv = *reinterpret_cast<const f32x4*>(B + (int64_t)pos * R + d);
  f32x4 r;
  r[0] = av[0] * v[0] + bv[1] * v[1];
  r[1] = av[1] * v[1] - bv[0] * v[0];
  r[2] = av[2] * v[2] + bv[3] * v[3];
  r[3] = av[3] * v[3] - bv[2] * v[2];
  return r;
}

template <int HD>
__global__ __launch_bounds__(256) void attn_short_bwd_kernel(ShortArgs a) {
  using Gm = ShortGeom<HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t item = (int64_t)blockIdx.x * 4 + wave;
  if (item >= a.items) return;
  const int64_t g = item / a.H;
  const int h = (int)(item - g * a.H);
  const int S = a.S, D = a.H * HD, frow = lane & 15, fk = lane >> 4;
  const int64_t ld = 3 * (int64_t)D;
  char* imgK = smem + wave * 3 * Gm::MATB;
  char* imgQ = imgK + Gm::MATB;
  char* imgO = imgQ + Gm::MATB;

  const int rowc = frow < S ? frow : S - 1;
  const bf16* rp = a.qkv + ((int64_t)g * S + rowc) * ld + h * HD + 8 * fk;
  const bf16* dp_ = a.dout + ((int64_t)g * S + rowc) * D + h * HD + 8 * fk;
  bf16x8 qf[Gm::KS], kf[Gm::KS], vf[Gm::KS], gf[Gm::KS];
#pragma unroll
  for (int ks = 0; ks < Gm::KS; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8*>(rp + 32 * ks);
    kf[ks] = *reinterpret_cast<const bf16x8*>(rp + D + 32 * ks);
    vf[ks] = *reinterpret_cast<const bf16x8*>(rp + 2 * D + 32 * ks);
    gf[ks] = *reinterpret_cast<const bf16x8*>(dp_ + 32 * ks);
  }
  park<HD>(imgK, kf, frow, fk);
  park<HD>(imgQ, qf, frow, fk);
  park<HD>(imgO, gf, frow, fk);

  // row statistics: (m, log2 l) of query frow (for the transposed products) and of queries 4 fk + e (for the others)
  const float* lbase = a.lse + ((int64_t)g * a.H + h) * S * 2;
  const float mT = lbase[2 * rowc], llT = lbase[2 * rowc + 1];
  float mq[4], llq[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int q = 4 * fk + e < S ? 4 * fk + e : S - 1;
    mq[e] = lbase[2 * q];
    llq[e] = lbase[2 * q + 1];
  }

  f32x4 st = {0.f, 0.f, 0.f, 0.f}, sn = st, dpt = st, dpn = st;
#pragma unroll
  for (int ks = 0; ks < Gm::KS; ++ks) {
    st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qf[ks], st, 0, 0, 0);     // S^T [key 4fk+e][query frow]
    sn = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ks], kf[ks], sn, 0, 0, 0);     // S   [query 4fk+e][key frow]
    dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[ks], gf[ks], dpt, 0, 0, 0);   // dP^T[key 4fk+e][query frow]
    dpn = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[ks], vf[ks], dpn, 0, 0, 0);   // dP  [query 4fk+e][key frow]
  }

  const float c1 = a.scale * LOG2E;
  // transposed side: P^T, delta of query frow, dS^T
  float pT[4], dT = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int key = 4 * fk + e;
    float b = key_bias(a.key_mask, g, S, key);
    if (a.causal && key > frow) b = -INFINITY;
    pT[e] = __builtin_amdgcn_exp2f((fmaf(st[e], c1, b) - mT) - llT);   // the reference's arithmetic order (fully padded rows stay uniform)
    dT += pT[e] * dpt[e];
  }
  dT = quad_sum(dT);                                 // delta[query frow] = sum_key P dP
  float dsT[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) dsT[e] = frow < S ? pT[e] * (dpt[e] - dT) * a.scale : 0.f;
  // the other side: P and dS with the query on the register index; delta comes from the lane that owns that query
  const float bn = key_bias(a.key_mask, g, S, frow);
  float pn[4], dsn[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int q = 4 * fk + e;
    const float dq_ = __shfl(dT, q, 64);             // lane q has frow == q
    float b = bn;
    if (a.causal && frow > q) b = -INFINITY;
    float p = __builtin_amdgcn_exp2f((fmaf(sn[e], c1, b) - mq[e]) - llq[e]);
    p = q < S ? p : 0.f;                             // queries beyond the sequence contribute nothing to dK / dV
    pn[e] = p;
    dsn[e] = p * (dpn[e] - dq_) * a.scale;
  }
  const s16x4 dsT_b = to_bf16x4(dsT), dsn_b = to_bf16x4(dsn), pn_b = to_bf16x4(pn);

  wave_lds_sync();
  bf16* gp = a.dqkv + ((int64_t)g * S + frow) * ld + h * HD + 4 * fk;
  const bool rot = a.rot.qa != nullptr;
#pragma unroll
  for (int db = 0; db < Gm::DB; ++db) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 dq = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(tr_block<HD>(imgK, db, frow, fk), dsT_b, z, 0, 0, 0);   // dQ^T[16db+4fk+e][query frow]
    f32x4 dk = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(tr_block<HD>(imgQ, db, frow, fk), dsn_b, z, 0, 0, 0);   // dK^T[..][key frow]
    f32x4 dv = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(tr_block<HD>(imgO, db, frow, fk), pn_b, z, 0, 0, 0);    // dV^T[..][key frow]
    if (frow < S) {
      const int d = 16 * db + 4 * fk;
      if (rot) {
        dq = rot_adj4(dq, a.rot.qa, a.rot.qb, a.rot.R, frow, d);
        dk = rot_adj4(dk, a.rot.ka, a.rot.kb, a.rot.R, frow, d);
      }
      bf16x4 o0, o1, o2;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o0[e] = (bf16)dq[e]; o1[e] = (bf16)dk[e]; o2[e] = (bf16)dv[e]; }
      *reinterpret_cast<bf16x4*>(gp + 16 * db) = o0;
      *reinterpret_cast<bf16x4*>(gp + D + 16 * db) = o1;
      *reinterpret_cast<bf16x4*>(gp + 2 * D + 16 * db) = o2;
    }
  }
}

template <int HD> int launch_fwd(const ShortArgs& a, hipStream_t stream) {
  const unsigned grid = (unsigned)ceil_div(a.items, 4);
  hipLaunchKernelGGL(attn_short_fwd_kernel<HD>, dim3(grid), dim3(256), 4 * ShortGeom<HD>::MATB, stream, a);
  MEANT_LAUNCH_CHECK("attn_short_fwd");
  return MEANT_OK;
}
template <int HD> int launch_bwd(const ShortArgs& a, hipStream_t stream) {
  const unsigned grid = (unsigned)ceil_div(a.items, 4);
  hipLaunchKernelGGL(attn_short_bwd_kernel<HD>, dim3(grid), dim3(256), 4 * 3 * ShortGeom<HD>::MATB, stream, a);
  MEANT_LAUNCH_CHECK("attn_short_bwd");
  return MEANT_OK;
}

}  // namespace

bool attn_short_ok(int64_t S, int Dh) {
  return S >= 1 && S <= 16 && (Dh == 64 || Dh == 96 || Dh == 128) && meant_opt(MEANT_OPT_ATTN_SHORT) != 0;
}

int attn_short_fwd(const bf16* qkv, bf16* o, float* lse, const float* key_mask, int64_t G, int64_t S, int H, int Dh, float scale,
                   int causal, hipStream_t stream) {
  MEANT_REQUIRE(attn_short_ok(S, Dh), MEANT_ERR_UNSUPPORTED, "attn_short_fwd: S=%lld Dh=%d", (long long)S, Dh);
  MEANT_REQUIRE(meant_aligned16(qkv) && meant_aligned16(o), MEANT_ERR_ARG, "attn_short_fwd: 16-byte alignment");
  MEANT_REQUIRE(G * (int64_t)H < (1LL << 33), MEANT_ERR_UNSUPPORTED, "attn_short_fwd: G*H exceeds the grid limit");
  if (G == 0) return MEANT_OK;
  meant_route_hit(ROUTE_ATTN_SHORT);
  ShortArgs a{qkv, o, lse, key_mask, nullptr, nullptr, RotTables{nullptr, nullptr, nullptr, nullptr, 0}, G * (int64_t)H, (int)S, H, scale, causal};
  return Dh == 64 ? launch_fwd<64>(a, stream) : Dh == 96 ? launch_fwd<96>(a, stream) : launch_fwd<128>(a, stream);
}

int attn_short_bwd(const bf16* qkv, const bf16* dout, const float* lse, const float* key_mask, bf16* dqkv, int64_t G, int64_t S, int H,
                   int Dh, float scale, int causal, RotTables rot, hipStream_t stream) {
  MEANT_REQUIRE(attn_short_ok(S, Dh), MEANT_ERR_UNSUPPORTED, "attn_short_bwd: S=%lld Dh=%d", (long long)S, Dh);
  MEANT_REQUIRE(!rot.qa || (rot.R % 8 == 0 && rot.R <= Dh), MEANT_ERR_UNSUPPORTED, "attn_short_bwd: rotary dim must be a multiple of 8 and <= Dh");
  MEANT_REQUIRE(meant_aligned16(qkv) && meant_aligned16(dout) && meant_aligned16(dqkv), MEANT_ERR_ARG, "attn_short_bwd: 16-byte alignment");
  MEANT_REQUIRE(G * (int64_t)H < (1LL << 33), MEANT_ERR_UNSUPPORTED, "attn_short_bwd: G*H exceeds the grid limit");
  if (G == 0) return MEANT_OK;
  meant_route_hit(ROUTE_ATTN_SHORT);
  ShortArgs a{qkv, nullptr, const_cast<float*>(lse), key_mask, dout, dqkv, rot, G * (int64_t)H, (int)S, H, scale, causal};
  return Dh == 64 ? launch_bwd<64>(a, stream) : Dh == 96 ? launch_bwd<96>(a, stream) : launch_bwd<128>(a, stream);
}
