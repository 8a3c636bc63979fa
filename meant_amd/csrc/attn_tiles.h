// Shared pieces of the bf16 attention kernels (attn_bf16.hip: flash forward + two-pass backward; attn_bwd1.hip: single-pass
// backward): the swizzled 64 x 64 tile image in LDS, its DMA staging, row / transposed MFMA fragments, the per-wave
// transposed store, the rotary adjoint on accumulators, the backward argument block.
#pragma once
#include "internal.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

constexpr int DH = 64;
constexpr int KV_TILE = 64;
constexpr int TILE_B = KV_TILE * DH * 2;          // 8 KiB
constexpr float LOG2E = 1.4426950408889634f;
constexpr float PADL2 = 1e9f * 1.4426950408889634f;  // the reference's (1-mask) * -1e9 key-padding term, in log2 units

// tell the compiler a condition is wave-uniform (it cannot prove it for values that came through vector loads or
// ballots): a branch on a "divergent" bool is compiled to exec masking, and every accumulator live across it gets
// copied at the join (the dK/dV loop carried 128 v_mov_b64 per tile for that reason)
__device__ __forceinline__ bool uni(bool x) { return __builtin_amdgcn_readfirstlane((int)x) != 0; }

// max over lanes l and l ^ 32 on the VALU (gfx950's v_permlane32_swap: the upper half of the first operand trades places with
// the lower half of the second) instead of a ds_bpermute_b32 round trip in the softmax's dependent chain
__device__ __forceinline__ float xor32_max(float v) {
  const unsigned b = __builtin_bit_cast(unsigned, v);
  const auto sw = __builtin_amdgcn_permlane32_swap(b, b, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
}

__device__ __forceinline__ int swz(int r) { return ((r & 2) << 1) | ((r >> 2) & 3); }

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}

__device__ __forceinline__ void glds4(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 4, 0, 0);
}

// Work item of a workgroup in a 1-D launch over nx * H * G items (x = 128-row block of the sequence, fastest).  Workgroups are
// dealt round-robin over the 8 XCDs (block b runs on XCD b % 8), so with the natural order block -> (x, h, g) and nx = 4 the
// causal text shape gave XCDs 0 and 4 every heaviest block and XCDs 3 and 7 every lightest one: the launch lasted as long
// as two of the eight XCDs needed (dK/dV: 1.16 waves per SIMD resident on average out of 2).  Here XCD x walks the
// contiguous item range [x * chunk, (x + 1) * chunk): every XCD gets the same mix, and the blocks of one (g, h) -- which
// share their K / V or Q / dO tiles -- meet in one L2.
__device__ __forceinline__ bool attn_item(int nx, int H, int G, int& x, int& h, int& g) {
  const unsigned N = (unsigned)nx * H * G, id = blockIdx.x, chunk = (N + 7) / 8;
  const unsigned j = id >> 3, item = (id & 7) * chunk + j;
  if (j >= chunk || item >= N) return false;
  x = (int)(item % nx);
  const unsigned gh = item / nx;
  h = (int)(gh % H);
  g = (int)(gh / H);
  return true;
}
__host__ inline unsigned attn_grid(int64_t nx, int H, int64_t G) { return (unsigned)(8 * ((nx * H * G + 7) / 8)); }

// stage a [64 rows][64 cols] bf16 tile: rows row0.. (clamped to nrows-1) of a matrix with row stride ld.
// 8 pieces of 1 KiB (8 rows each); wave w issues pieces 2w, 2w+1.  A lane's row-in-tile and swizzled 16-byte column
// never change, so its two element offsets are computed once (StageOff); per tile only the uniform tile origin moves.
// Rows past the end of the sequence exist only in a ragged last tile, which gets its own clamped pair.
struct StageOff { unsigned full[2], last[2]; };
// ncols < 64 (head dim 96: the second sub-tile has 32 real columns): slots beyond the head re-read one of its first
// ncols / 8 chunks instead of running into the next head (or, for the last head of the last row, out of the buffer);
// nothing reads them
__device__ __forceinline__ StageOff make_stage_off(int64_t ld, int nrows, int wave, int lane, int ncols = 64) {
  StageOff o;
  const int last0 = ((nrows - 1) / KV_TILE) * KV_TILE;              // first row of the last tile
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 8 + (lane >> 3);
    const int c = ((lane & 7) ^ swz(r)) & (ncols / 8 - 1);
    const int rl = last0 + r < nrows ? r : nrows - 1 - last0;
    o.full[i] = (unsigned)(r * ld + c * 8);
    o.last[i] = (unsigned)(rl * ld + c * 8);
  }
  return o;
}
__device__ __forceinline__ void stage64(const bf16* __restrict__ g, int64_t ld, int row0, int nrows, char* tile, int wave, int lane,
                                        const StageOff& so) {
  const bf16* origin = g + (int64_t)row0 * ld;                       // uniform
  const bool ragged = row0 + KV_TILE > nrows;                        // uniform
#pragma unroll
  for (int i = 0; i < 2; ++i) glds16(origin + (ragged ? so.last[i] : so.full[i]), tile + (wave * 2 + i) * 1024);
}

// A/B fragment of mfma_32x32x16 from a row-major tile: lane (r = lane&31, h = lane>>5) gets
// tile[row0 + r][16*ks + 8h .. +7]
__device__ __forceinline__ bf16x8 frag_row(const char* tile, int row0, int ks, int lane) {
  const int r = row0 + (lane & 31);
  const int c = 2 * ks + (lane >> 5);
  return *reinterpret_cast<const bf16x8*>(tile + r * 128 + ((c ^ swz(r)) << 4));
}

// transposed fragment: lane (col = lane&31, h = lane>>5) gets, for j = 0..7,
// tile[krow0 + 8*(j>>2) + 4h + (j&3)][col0 + col]   -- the k order of an accumulator reused as an operand.
// Issued as inline-asm ds_read_b64_tr_b16 (see common.h: the builtin makes hipcc drain the DMA queue).
// Because the swizzle XORs the chunk index with bits of the row, the per-lane address is precomputed for
// u = j>>2 in {0,1} and the 32-column block b in {0,1}; krow0 (a multiple of 16 rows) is an immediate.
struct TrOff { unsigned o[2][2]; };               // [u][b], byte offsets inside a tile
__device__ __forceinline__ TrOff make_troff(int lane) {
  TrOff t;
  const int h = lane >> 5, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int r = 8 * u + 4 * h + q;
      const int col = 32 * b + ((lane >> 4) & 1) * 16 + 4 * p;
      t.o[u][b] = (unsigned)(r * 128 + ((((col >> 3)) ^ swz(r)) << 4) + (col & 7) * 2);
    }
  return t;
}
template <int KROW0>
__device__ __forceinline__ void tr_issue(unsigned tile_addr, const TrOff& t, int b, u32x2& lo, u32x2& hi) {
  lo = lds_read_tr16<KROW0 * 128>(tile_addr + t.o[0][b]);
  hi = lds_read_tr16<KROW0 * 128>(tile_addr + t.o[1][b]);
}

// row fragment by inline-asm ds_read_b128 (immediate offset): issued where the source says, in a batch, so that the LDS
// latency is paid once per batch instead of once per MFMA (hipcc pairs every builtin LDS load with an lgkmcnt(0) right in
// front of its consumer).  The consumer must sit behind lds_wait_frags() on the same registers.
template <int OFF>
__device__ __forceinline__ bf16x8 lds_frag(unsigned addr) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return __builtin_bit_cast(bf16x8, r);
}
// wait until at most N younger LDS operations are outstanding; names the fragments so that their consumers depend on it
template <int N>
__device__ __forceinline__ void lds_wait_frags(bf16x8& a, bf16x8& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

// accumulator (32 keys x 32 queries, fp32) -> two bf16 B-operand fragments (k-steps of 16 keys)
__device__ __forceinline__ void acc_to_frags(const f32x16& a, bf16x8& f0, bf16x8& f1) {
#pragma unroll
  for (int j = 0; j < 8; ++j) { f0[j] = (bf16)a[j]; f1[j] = (bf16)a[8 + j]; }
}

__device__ __forceinline__ int acc_row(int e, int lane) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }

// write a wave's transposed result (acc[b][e]: row = dh 32b + acc_row(e), col = lane&31 = token) to global
// rows of 64 bf16 through a per-wave LDS patch so that stores are 16 bytes per lane, 128 bytes per row.
// nblk = 1: only acc[0] / the first 32 columns exist (second half of a 96-wide head)
__device__ __forceinline__ void store_transposed(const f32x16 (&acc)[2], float mult, char* patch /* 32 x 144 B */, bf16* __restrict__ dst,
                                                  int64_t ld, int tok0, int ntok, int lane, int nblk = 2) {
  const int t = lane & 31;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    if (b >= nblk) break;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      bf16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (bf16)(acc[b][g4 * 4 + e] * mult);
      const int dh = 32 * b + 8 * g4 + 4 * (lane >> 5);
      *reinterpret_cast<bf16x4*>(patch + t * 144 + dh * 2) = v;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = i * 64 + lane;                 // 32 rows x 8 chunks
    const int r = idx >> 3, c = idx & 7;
    const u32x4 v = *reinterpret_cast<const u32x4*>(patch + r * 144 + c * 16);
    if (tok0 + r < ntok && c < 4 * nblk) *reinterpret_cast<u32x4*>(dst + (int64_t)(tok0 + r) * ld + c * 8) = v;
  }
}

// adjoint of the rotary map on a transposed gradient tile, in registers and fp32 (so the result is rounded once):
// acc[b][e] holds d/d(rotated)[token = this lane's][dh = 32b + acc_row(e)]; pairs (2j, 2j+1) sit in adjacent
// registers.  d t[2j] = A[2j] d[2j] + B[2j+1] d[2j+1] ;  d t[2j+1] = A[2j+1] d[2j+1] - B[2j] d[2j]  for dh < R.
__device__ __forceinline__ void rotary_adjoint_regs(f32x16 (&acc)[2], const float* __restrict__ A, const float* __restrict__ B, int R,
                                                    int pos, int lane) {
  const float* Ar = A + (int64_t)pos * R;
  const float* Br = B + (int64_t)pos * R;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int dh = 32 * b + 8 * g4 + 4 * (lane >> 5);
      if (dh < R) {                                   // R % 8 == 0 and dh % 4 == 0: the 4 lanes are in or out together
        const f32x4 av = *reinterpret_cast<const f32x4*>(Ar + dh);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(Br + dh);
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const float d0 = acc[b][g4 * 4 + e], d1 = acc[b][g4 * 4 + e + 1];
          acc[b][g4 * 4 + e] = av[e] * d0 + bv[e + 1] * d1;
          acc[b][g4 * 4 + e + 1] = av[e + 1] * d1 - bv[e] * d0;
        }
      }
    }
}

__device__ __forceinline__ void sload_masks(const uint64_t* p, uint64_t& m0, uint64_t& m1) {
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  u4 v;
  asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  m0 = ((uint64_t)v[1] << 32) | v[0];
  m1 = ((uint64_t)v[3] << 32) | v[2];
}

// backward argument block (both backward forms)
struct BwdArgs {
  const bf16* qkv; const bf16* o; const bf16* dout; const float* lse; const float* bias2; const int* flags; bf16* dqkv; float* delta;
  const uint64_t* masks;                             // packed tile masks per group (null when S > 4096)
  int64_t plane;                                     // G*H*S: `delta` holds two planes, -(m + log2 l) and -delta, per (g, h, query)
  int S, H; float scale; int causal;
  int G, nblk;                                       // groups, 128-row blocks per sequence (work items: ceil(nblk / 2) * H * G pairs)
  RotTables rot;                                     // adjoint rotary on dq / dk when rot.qa != null
};

}  // namespace
