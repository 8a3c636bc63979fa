// K3/K4: fused (flash-style) attention core for the bf16 tier, head dims 64 and 128, gfx950 MFMA 32x32x16.
// Replaces the reference's eager scores->softmax->PV chain (meant/attention.py:43-57,
// meant/xPosAttention.py:41-63) and its flash-attn dependency (meant/flash_attention.py:42,
// meant/xPosAttention_flash.py:40) with eager semantics: scale 1/sqrt(dim), causal -inf, additive
// (1-mask)*-1e9 key padding applied to the fp32 score before the softmax.
//
// Layout: q|k|v packed [G*S, 3*H*64] (rotary already applied), heads are 64-column slices; O [G*S, H*64].
//
// Forward / dQ kernels ("query on the lane"): workgroup = 4 waves x 32 queries; K/V tiles of 64 keys
//   stream HBM -> LDS by global_load_lds (double-buffered, one barrier per tile).
//   S^T = K Q^T is computed with the KEY as the MFMA row, so a lane owns one query column: softmax
//   statistics are per-lane scalars, and the exponentiated accumulator is directly the B operand of the
//   next product (O^T += V^T P^T, dQ^T += K^T dS^T) -- no LDS round trip for P.  The V^T / K^T operands come
//   from ds_read_b64_tr_b16 (hardware transpose) of the row-major tiles.
// dK/dV kernel ("key on the lane"): workgroup = 4 waves x 32 keys, K and V fragments live in registers,
//   Q / dO tiles of 64 queries stream through LDS and are read by rows (S, dP) and transposed (dV^T, dK^T).
//   dQ is produced by its own pass (recomputing S and dP) instead of float atomics: deterministic, and
//   the atomic rate (1.3 TB/s) would otherwise bound the kernel.
// One LDS image serves row reads and transposed reads: 128-byte rows, 16-byte chunk c of row r stored at
//   slot c ^ f(r), f(r) = ((r&2)<<1) | ((r>>2)&3)  (conflict-free for both access shapes).
// Row statistics are the pair (m, log l): see include/meant_hip.h.
// Roofline: per (g,h) the kernel moves 4*S*64*2 bytes and does 4*S*S*64 FLOPs (fwd); at S=512 / 196 the
// arithmetic intensity (256 / 98 FLOP/B) is below the machine balance, and the exp() work per score keeps
// the VALU, not the MFMA, on the critical path for Dh=64.
#include "attn_tiles.h"
#include <type_traits>

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

namespace {


// ------------------------------------------------------------------------------------------------
// key-padding mask -> additive bias in log2 units, bias2[g][key] = (1 - mask) * -1e9 * log2e (0 for real keys,
// -inf for the padding keys >= S of the last tile), and flags[g][tile]:
//   bit 0: the tile needs the bias or the tail treatment at all.  Padding is a suffix in practice, so most tiles
//          take the bias-free path.
//   bit 1: the tile can be skipped outright: all of its 64 keys are dead (bias <= -1e8, so exp2(t - m) underflows
//          to exactly 0 against any live key) AND every query row of the group is guaranteed a live visible key
//          (key 0 under the causal mask, any key otherwise) -- rows with no live key keep the reference's uniform
//          softmax over the padding and therefore need every tile.
constexpr float DEAD_BIAS = -1e8f, LIVE_BIAS = -1e6f;
__global__ __launch_bounds__(64) void attn_prep_mask_kernel(const float* __restrict__ km, float* __restrict__ bias2,
                                                             int* __restrict__ flags, int S, int nt, int causal, unsigned* __restrict__ ctr) {
  const int g = blockIdx.y, t = blockIdx.x, lane = threadIdx.x;
  if (ctr && g == 0 && t == 0 && lane < 8) ctr[lane * 16] = 0u;   // the single-pass backward's per-XCD item counters
  const int key = t * KV_TILE + lane;
  float b = 0.f;
  if (key >= S) b = -INFINITY;
  else if (km) b = fmaf(km[(int64_t)g * S + key], PADL2, -PADL2);
  bias2[((int64_t)g * nt + t) * KV_TILE + lane] = b;
  const int any = __any(b != 0.f);
  int skip = 0;
  if (km && __all(b <= DEAD_BIAS)) {
    const float* kg = km + (int64_t)g * S;
    if (causal) skip = fmaf(kg[0], PADL2, -PADL2) >= LIVE_BIAS;
    else {
      bool live = false;
      for (int k = lane; k < S; k += 64) live |= fmaf(kg[k], PADL2, -PADL2) >= LIVE_BIAS;
      skip = __any(live);
    }
  }
  if (lane == 0) flags[(int64_t)g * nt + t] = any | (skip << 1);
}

// packed per-group tile masks: masks[g] = {bit t: tile t needs the bias / tail path,
// bit t: tile t can be skipped} (ntile <= 64).  Two 64-bit words per group that a workgroup fetches with ONE scalar load
// when it picks up a work item, instead of a vector load + ballots whose latency would sit in the item's start-up chain.
__global__ __launch_bounds__(64) void attn_pack_flags_kernel(const int* __restrict__ flags, uint64_t* __restrict__ masks, int nt) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const int f = lane < nt ? flags[(int64_t)g * nt + lane] : 0;
  const uint64_t sp = __ballot(f & 1), sk = __ballot(f & 2);
  if (lane == 0) { masks[2 * (int64_t)g] = sp; masks[2 * (int64_t)g + 1] = sk; }
}

// ------------------------------------------------------------------------------------------------
// forward
struct FwdArgs {
  const bf16* qkv; bf16* o; float* lse; const float* bias2; const int* flags; const uint64_t* masks;
  int S, H; float scale; int causal; int G, nqb;
};


// Work item of the forward, dQ and dK/dV kernels: a PAIR of 128-row blocks (p, nblk - 1 - p) of one (g, h), processed one after the other.
// Under the causal mask block x has work proportional to x + 1 (dQ) or nblk - x (dK/dV), so single blocks last anything from
// zero (dead padding) to 8 tiles and the launch kept 1.2 of 2 waves per SIMD resident; pairs are all about equally long
// and there are half as many launches.  (Persistent forms were tried for both directions: a work queue around the backward
// bodies costs ~80 spilled SGPRs and the dQ kernel's third wave per SIMD; the persistent forward of round 1, which streamed
// K/V tiles across item boundaries, measures 1.57 ms on the causal text shape against 1.45 ms for pairs, and was removed.)
#define ATTN_PAIR_LOOP(NBLK, FIRST_HEAVY_IS_LAST)                                                              \
  {                                                                                                       \
    int p__, h__, g__;                                                                                    \
    const int nblk__ = (NBLK);                                                                            \
    const int npair__ = a.causal ? (nblk__ + 1) / 2 : nblk__;      /* without the mask all blocks are alike */       \
    if (!attn_item(npair__, H, a.G, p__, h__, g__)) return;                                               \
    const int hi__ = a.causal ? nblk__ - 1 - p__ : p__;                                                   \
    process((FIRST_HEAVY_IS_LAST) ? hi__ : p__, h__, g__);                                                \
    if (hi__ != p__) {                                                                                    \
      __syncthreads();                                                                                    \
      process((FIRST_HEAVY_IS_LAST) ? p__ : hi__, h__, g__);                                              \
    }                                                                                                     \
  }

// ------------------------------------------------------------------------------------------------
// forward
template <int HD_>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // two buffers of [K: NH sub-tiles | V: NH sub-tiles] (a sub-tile = 64 keys x 64 of the head's 64 NH columns), then
  // bias[2][64] floats, then 4 per-wave patches of 32x144 B (NH = 2: the patches reuse the tile space after the loop)
  constexpr int HD = HD_, NH = (HD + 63) / 64, KS = HD / 16, OB = HD / 32, BUF_B = 2 * NH * TILE_B;   // sub-tiles, k-steps, 32-column output blocks
  float* bias_s = reinterpret_cast<float*>(smem + 2 * BUF_B);
  char* patches = NH == 1 ? smem + 2 * BUF_B + 2 * 64 * 4 : smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S = a.S, H = a.H, D = H * HD;
  const int64_t ld = 3 * (int64_t)D;
  auto process = [&](int qb, int h, int g) __attribute__((always_inline)) {
  const int q0 = qb * 128 + wave * 32;             // this wave's first query
  const bf16* base = a.qkv + (int64_t)g * S * ld + h * HD;
  const bf16* Kg = base + D;
  const bf16* Vg = base + 2 * D;
  const int ntile = (S + KV_TILE - 1) / KV_TILE;
  const float* b2g = a.bias2 + (int64_t)g * ntile * KV_TILE;     // this group's bias tiles
  const int* flg = a.flags + (int64_t)g * ntile;

  // everything staged inside the loop goes by DMA (a plain LDS store would make hipcc drain the DMA queue).  The first
  // tile is requested before anything else: its round trip is the longest link of the workgroup's start-up chain.
  const StageOff soff = make_stage_off(ld, S, wave, lane);
  const StageOff soff2 = make_stage_off(ld, S, wave, lane, HD - 64 * (NH - 1));   // last sub-tile: 32 real columns when HD = 96
  auto stage = [&](int t, int buf) {
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
      stage64(Kg + DH * hf, ld, t * KV_TILE, S, smem + buf * BUF_B + hf * TILE_B, wave, lane, hf == NH - 1 ? soff2 : soff);
      stage64(Vg + DH * hf, ld, t * KV_TILE, S, smem + buf * BUF_B + (NH + hf) * TILE_B, wave, lane, hf == NH - 1 ? soff2 : soff);
    }
    if (wave == 0) glds4(b2g + t * KV_TILE + lane, bias_s + buf * 64);   // the tile's 64 bias values
  };
  stage(0, 0);

  // Q fragments (B operand): lane (query = lane&31, half) holds Q[q][16ks + 8*half .. +7]
  bf16x8 qf[KS];
  {
    int qrow = q0 + (lane & 31);
    qrow = qrow < S ? qrow : S - 1;
    const bf16* qp = base + (int64_t)qrow * ld + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
  }
  const int myq = q0 + (lane & 31);

  int kend = S;                                      // keys needed by this block
  if (a.causal) { const int lastq = qb * 128 + 127; kend = lastq + 1 < S ? lastq + 1 : S; }
  int nt = (kend + KV_TILE - 1) / KV_TILE;
  // The per-tile flags are fetched ONCE, into two wave-uniform 64-bit masks.  (A flag load inside the tile loop sits
  // behind the DMA just issued for the next tile in the in-order vmcnt queue: waiting for the flag would drain the
  // DMA and serialise every tile's load with its math.)
  uint64_t special_mask = 0, skip_mask = 0;
  const bool masks_ok = a.masks != nullptr;
  if (masks_ok) sload_masks(a.masks + 2 * (int64_t)g, special_mask, skip_mask);   // one scalar load, not a vector load + ballots
  auto tile_flag = [&](int t) -> int {
    return masks_ok ? (int)((special_mask >> t) & 1) | ((int)((skip_mask >> t) & 1) << 1) : flg[t];
  };
  while (nt > 1 && (tile_flag(nt - 1) & 2)) --nt;    // trailing all-padding tiles contribute exactly 0

  f32x16 oacc[2 * NH];
#pragma unroll
  for (int b = 0; b < OB; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[b][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c1 = a.scale * LOG2E;

  const TrOff troff = make_troff(lane);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) stage(t + 1, buf ^ 1);
    const int k0 = t * KV_TILE;
    const int flag = tile_flag(t);
    const bool active = !(flag & 2) && (!a.causal || (k0 <= q0 + 31));   // wave-uniform: tile not entirely above the diagonal / all padding
    if (active && q0 < S) {
      const char* Kt = smem + buf * BUF_B;
      const char* Vt = Kt + NH * TILE_B;
      // S^T (2 sub-tiles of 32 keys)
      f32x16 sacc[2];
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[sb][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          sacc[sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(Kt + (ks >> 2) * TILE_B, 32 * sb, ks & 3, lane), qf[ks], sacc[sb], 0, 0, 0);
      }
      // softmax in the log2 domain: t = s * (scale*log2e) (+ bias), p = exp2(t - m).  Tiles that touch neither the
      // diagonal, nor padding, nor the end of the sequence (most of them) take the mask-free path: one multiply and
      // one max per score.  O and l are rescaled only on tiles where some lane's running max actually moves.
      const bool diag = a.causal && (k0 + KV_TILE - 1 > q0);   // tile touches the diagonal for some query of this wave
      const bool special = diag || (flag & 1);
      float tmx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      if (!special) {                                  // raw scores: the scale is folded into the exponent below
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
          for (int e = 0; e < 16; ++e) tmx[e & 3] = fmaxf(tmx[e & 3], sacc[sb][e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) tmx[e] *= c1;
      } else {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int kl0 = 32 * sb + 8 * g4 + 4 * (lane >> 5);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + buf * 64 + kl0);
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
              float t = fmaf(sacc[sb][g4 * 4 + e4], c1, bv[e4]);
              if (diag && (k0 + kl0 + e4 > myq)) t = -INFINITY;
              sacc[sb][g4 * 4 + e4] = t;
              tmx[e4] = fmaxf(tmx[e4], t);
            }
          }
      }
      float tmax = fmaxf(fmaxf(tmx[0], tmx[1]), fmaxf(tmx[2], tmx[3]));
      tmax = xor32_max(tmax);
      // m_run stays finite for every real query: key 0 is never above the diagonal and biases are finite for key < S
      if (!__all(tmax <= m_run)) {
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
        m_run = m_new;
#pragma unroll
        for (int b = 0; b < OB; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) oacc[b][e] *= alpha;
      }
      float ps[4] = {0.f, 0.f, 0.f, 0.f};
      const float ec = special ? 1.0f : c1;            // special tiles hold scaled + biased scores already
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float p = __builtin_amdgcn_exp2f(fmaf(sacc[sb][e], ec, -m_run));
          sacc[sb][e] = p;
          ps[e & 3] += p;
        }
      l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);      // this lane's keys only: the two halves of a query's lanes meet after the loop
      // O^T += V^T P^T
      const unsigned vaddr = lds_addr(Vt);
      auto pv = [&](auto SB) {
        constexpr int sb = decltype(SB)::value;
        bf16x8 pf[2];
        acc_to_frags(sacc[sb], pf[0], pf[1]);
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
          u32x2 lo[2][2], hi[2][2];
          tr_issue<32 * sb>(vaddr + hf * TILE_B, troff, 0, lo[0][0], hi[0][0]);
          tr_issue<32 * sb>(vaddr + hf * TILE_B, troff, 1, lo[0][1], hi[0][1]);
          tr_issue<32 * sb + 16>(vaddr + hf * TILE_B, troff, 0, lo[1][0], hi[1][0]);
          tr_issue<32 * sb + 16>(vaddr + hf * TILE_B, troff, 1, lo[1][1], hi[1][1]);
          lds_wait_all();
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int b = 0; b < 2; ++b)
              if (2 * hf + b < OB)
                oacc[2 * hf + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack_tr(lo[s2][b], hi[s2][b]), pf[s2], oacc[2 * hf + b], 0, 0, 0);
        }
      };
      pv(std::integral_constant<int, 0>{});
      pv(std::integral_constant<int, 1>{});
    }
    __syncthreads();
  }

  if (q0 >= S) return;
  l_run += __shfl_xor(l_run, 32, 64);                  // lanes l and l + 32 hold the two key halves of one query; both saw the same m_run
  const float inv_l = 1.0f / l_run;
  if (lane < 32 && myq < S) {
    float* lp = a.lse + (((int64_t)g * H + h) * S + myq) * 2;
    lp[0] = m_run;                                  // log2-domain statistics (tier-internal layout)
    lp[1] = __log2f(l_run);
  }
  // 1/l differs per lane (query): scale per lane, then transpose through the wave's LDS patch
#pragma unroll
  for (int b = 0; b < OB; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[b][e] *= inv_l;
#pragma unroll
  for (int hf = 0; hf < NH; ++hf) {
    if (hf) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    store_transposed(reinterpret_cast<const f32x16(&)[2]>(oacc[2 * hf]), 1.0f, patches + wave * (32 * 144),
                     a.o + (int64_t)g * S * D + h * HD + DH * hf, D, q0, S, lane, OB - 2 * hf >= 2 ? 2 : 1);
  }
  };   // process
  ATTN_PAIR_LOOP(a.nqb, true)
}


// ------------------------------------------------------------------------------------------------
// backward, pass 1: dQ (and delta).  Same geometry as the forward.

template <int HD_>
__global__ __launch_bounds__(256, HD_ == 64 ? 3 : 2) void attn_bwd_dq_kernel(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HD = HD_, NH = (HD + 63) / 64, KS = HD / 16, OB = HD / 32, BUF_B = 2 * NH * TILE_B;   // sub-tiles, k-steps, 32-column output blocks    // as in the forward
  float* bias_s = reinterpret_cast<float*>(smem + 2 * BUF_B);
  char* patches = NH == 1 ? smem + 2 * BUF_B + 2 * 64 * 4 : smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S = a.S, H = a.H, D = H * HD;
  const int64_t ld = 3 * (int64_t)D;
  auto process = [&](int qb, int h, int g) __attribute__((always_inline)) {
  const int q0 = qb * 128 + wave * 32;
  const bf16* base = a.qkv + (int64_t)g * S * ld + h * HD;
  const bf16* Kg = base + D;
  const bf16* Vg = base + 2 * D;
  const int ntile = (S + KV_TILE - 1) / KV_TILE;
  const float* b2g = a.bias2 + (int64_t)g * ntile * KV_TILE;     // this group's bias tiles
  const int* flg = a.flags + (int64_t)g * ntile;
  const int myq = q0 + (lane & 31);
  const int qrow = myq < S ? myq : S - 1;

  // everything staged inside the loop goes by DMA (a plain LDS store would make hipcc drain the DMA queue).  The first
  // tile is requested before anything else: its round trip is the longest link of the workgroup's start-up chain.
  const StageOff soff = make_stage_off(ld, S, wave, lane);
  const StageOff soff2 = make_stage_off(ld, S, wave, lane, HD - 64 * (NH - 1));   // last sub-tile: 32 real columns when HD = 96
  auto stage = [&](int t, int buf) {
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
      stage64(Kg + DH * hf, ld, t * KV_TILE, S, smem + buf * BUF_B + hf * TILE_B, wave, lane, hf == NH - 1 ? soff2 : soff);
      stage64(Vg + DH * hf, ld, t * KV_TILE, S, smem + buf * BUF_B + (NH + hf) * TILE_B, wave, lane, hf == NH - 1 ? soff2 : soff);
    }
    if (wave == 0) glds4(b2g + t * KV_TILE + lane, bias_s + buf * 64);   // the tile's 64 bias values
  };
  stage(0, 0);

  bf16x8 qf[KS], dof[KS];
  float delta = 0.f;
  {
    const bf16* qp = base + (int64_t)qrow * ld + 8 * (lane >> 5);
    const bf16* dop = a.dout + ((int64_t)g * S + qrow) * D + h * HD + 8 * (lane >> 5);
    const bf16* op = a.o + ((int64_t)g * S + qrow) * D + h * HD + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
      dof[ks] = *reinterpret_cast<const bf16x8*>(dop + 16 * ks);
      const bf16x8 ov = *reinterpret_cast<const bf16x8*>(op + 16 * ks);
#pragma unroll
      for (int j = 0; j < 8; ++j) delta += (float)dof[ks][j] * (float)ov[j];
    }
    delta += __shfl_xor(delta, 32, 64);
  }
  const float c1 = a.scale * LOG2E;
  float m_q, logl_q;                                 // log2-domain (m, log2 l) written by the forward
  {
    const float* lp = a.lse + (((int64_t)g * H + h) * S + qrow) * 2;
    m_q = lp[0];
    logl_q = lp[1];
  }
  const float nml = -(m_q + logl_q);                 // used on tiles without padding, where |m| is small and the sum is exact enough
  // per-query planes for the dK/dV pass: -(m + log2 l) and -delta (it starts its dP accumulators from the latter)
  if (lane < 32 && myq < S) {
    a.delta[((int64_t)g * H + h) * S + myq] = nml;
    a.delta[a.plane + ((int64_t)g * H + h) * S + myq] = -delta;
  }

  int kend = S;
  if (a.causal) { const int lastq = qb * 128 + 127; kend = lastq + 1 < S ? lastq + 1 : S; }
  int nt = (kend + KV_TILE - 1) / KV_TILE;
  // The per-tile flags are fetched ONCE, into two wave-uniform 64-bit masks.  (A flag load inside the tile loop sits
  // behind the DMA just issued for the next tile in the in-order vmcnt queue: waiting for the flag would drain the
  // DMA and serialise every tile's load with its math.)
  uint64_t special_mask = 0, skip_mask = 0;
  const bool masks_ok = a.masks != nullptr;
  if (masks_ok) sload_masks(a.masks + 2 * (int64_t)g, special_mask, skip_mask);   // one scalar load, not a vector load + ballots
  auto tile_flag = [&](int t) -> int {
    return masks_ok ? (int)((special_mask >> t) & 1) | ((int)((skip_mask >> t) & 1) << 1) : flg[t];
  };
  while (nt > 1 && (tile_flag(nt - 1) & 2)) --nt;    // trailing all-padding tiles contribute exactly 0

  f32x16 dqacc[2 * NH];
#pragma unroll
  for (int b = 0; b < OB; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) dqacc[b][e] = 0.f;

  const TrOff troff = make_troff(lane);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) stage(t + 1, buf ^ 1);
    const int k0 = t * KV_TILE;
    const int flag = tile_flag(t);
    const bool active = !(flag & 2) && (!a.causal || (k0 <= q0 + 31));
    if (active && q0 < S) {
      const char* Kt = smem + buf * BUF_B;
      const char* Vt = Kt + NH * TILE_B;
      const bool diag = a.causal && (k0 + KV_TILE - 1 > q0);
      const bool special = diag || (flag & 1);
      const unsigned kaddr = lds_addr(Kt);
      auto body = [&](auto SB) {
        constexpr int sb = decltype(SB)::value;
        f32x16 sacc, dpacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sacc[e] = 0.f; dpacc[e] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(Kt + (ks >> 2) * TILE_B, 32 * sb, ks & 3, lane), qf[ks], sacc, 0, 0, 0);
          dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(Vt + (ks >> 2) * TILE_B, 32 * sb, ks & 3, lane), dof[ks], dpacc, 0, 0, 0);
        }
        if (!special) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float p = __builtin_amdgcn_exp2f(fmaf(sacc[e], c1, nml));      // exp2(t - m - log2 l), plain rows
            sacc[e] = p * (dpacc[e] - delta);                                    // dS^T / scale
          }
        } else {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int kl0 = 32 * sb + 8 * g4 + 4 * (lane >> 5);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + buf * 64 + kl0);
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
              const int e = g4 * 4 + e4;
              const float t = fmaf(sacc[e], c1, bv[e4]);
              float p = __builtin_amdgcn_exp2f((t - m_q) - logl_q);
              if (diag && (k0 + kl0 + e4 > myq)) p = 0.f;
              sacc[e] = p * (dpacc[e] - delta);
            }
          }
        }
        bf16x8 dsf[2];
        acc_to_frags(sacc, dsf[0], dsf[1]);
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
          u32x2 lo[2][2], hi[2][2];
          tr_issue<32 * sb>(kaddr + hf * TILE_B, troff, 0, lo[0][0], hi[0][0]);
          tr_issue<32 * sb>(kaddr + hf * TILE_B, troff, 1, lo[0][1], hi[0][1]);
          tr_issue<32 * sb + 16>(kaddr + hf * TILE_B, troff, 0, lo[1][0], hi[1][0]);
          tr_issue<32 * sb + 16>(kaddr + hf * TILE_B, troff, 1, lo[1][1], hi[1][1]);
          lds_wait_all();
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int b = 0; b < 2; ++b)
              if (2 * hf + b < OB)
                dqacc[2 * hf + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack_tr(lo[s2][b], hi[s2][b]), dsf[s2], dqacc[2 * hf + b], 0, 0, 0);
        }
      };
      body(std::integral_constant<int, 0>{});
      body(std::integral_constant<int, 1>{});
    }
    __syncthreads();
  }
  if (q0 >= S) return;
  // rotary lanes are the first rot.R <= 64 columns of a head: the first half only
  if (a.rot.qa) rotary_adjoint_regs(reinterpret_cast<f32x16(&)[2]>(dqacc[0]), a.rot.qa, a.rot.qb, a.rot.R, qrow, lane);
#pragma unroll
  for (int hf = 0; hf < NH; ++hf) {
    if (hf) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    store_transposed(reinterpret_cast<const f32x16(&)[2]>(dqacc[2 * hf]), a.scale, patches + wave * (32 * 144),
                     a.dqkv + (int64_t)g * S * ld + h * HD + DH * hf, ld, q0, S, lane, OB - 2 * hf >= 2 ? 2 : 1);
  }
  };   // process
  ATTN_PAIR_LOOP(a.nblk, true)
}

// ------------------------------------------------------------------------------------------------
// backward, pass 2: dK, dV.  workgroup = 4 waves x 32 keys; Q / dO tiles of 64 queries stream through LDS.
//
// Per 32-query x 32-key block a lane holds 16 scores (its key, 16 query rows).  Two code paths, chosen per block by
// wave-uniform conditions, so that the common one is branch-free and as short as the arithmetic allows:
//   plain   -- every key of the wave is live (bias 0) and the block touches neither the diagonal nor the end of the
//              sequence: p = exp2(s * c1 + nml[q]) with nml = -(m + log2 l) precomputed per query by the dQ pass (exact
//              enough here: a query that sees a live key has a small m), dS = p * dP' with the dP accumulator STARTED at
//              -delta[q] (loaded from LDS straight into the accumulator registers), i.e. fma, exp, mul per score;
//   careful -- padding keys, the causal diagonal or the ragged last query tile: the reference's arithmetic step by step,
//              exp2((s * c1 + bias - m) - log2 l) (m ~ -1e9 on fully padded rows, where m + log2 l is not representable),
//              masked by selects.
// The previous form evaluated the careful path's conditions per element behind divergent branches: 1209 instructions per
// 64-query tile and wave for 32 MFMAs (485 of them scalar mask bookkeeping); this one issues ~330.
// ------------------------------------------------------------------------------------------------
// backward, pass 2: dK, dV.  workgroup = 4 waves x 32 keys; Q / dO tiles of 64 queries stream through LDS.
//
// Per 32-query x 32-key block a lane holds 16 scores (its key, 16 query rows).  The body of the kernel exists in two
// variants and each workgroup (128 keys of one (g, h)) runs exactly one of them from its first load to its last store:
//   PLAIN   -- all 128 keys are live (bias 0): p = exp2(s * c1 + nml[q]) with nml = -(m + log2 l) precomputed per query by
//              the dQ pass (exact enough here: a query that sees a live key has a small m), dS = p * dP' with the dP
//              accumulator STARTED at -delta[q] (loaded from LDS straight into the accumulator registers), i.e. fma, exp,
//              mul per score.  A quarter..half of the text blocks, all of the vision ones.
//   careful -- some key is padding or lies past the end of the sequence: the reference's arithmetic step by step,
//              exp2((s * c1 + bias - m) - log2 l) (m ~ -1e9 on fully padded rows, where m + log2 l is not representable).
//              Blocks made of dead key tiles only (flags bit 1) write their zeros here and leave.
// Why two whole bodies and not a branch inside the loop: a 64-query tile is ONE basic block -- scores of both 32-query blocks, weights of block
// 0, products of block 0, weights of block 1, products of block 1 -- so that hipcc's scheduler can put one block's softmax
// arithmetic between the other block's MFMAs (an MFMA occupies the wave's issue port for 8 of its 32 cycles; ~5 vector
// instructions fit in each gap).  Every wave-uniform branch inside would end the basic block, and a branch around an MFMA
// costs copies of the accumulators it touches.  The only branch left goes around the selects that zero the weights of
// invisible / non-existent queries on the causal diagonal and in a ragged last query tile (a block entirely above the
// diagonal is masked to zero, not skipped).
// History: the first form evaluated every mask condition per element behind divergent branches: 1209 instructions per
// tile and wave for 32 MFMAs (485 of them scalar mask bookkeeping); this one issues ~330.
#ifdef ATTN_LAB_STAMP
// lab build only (tools/lab): per-workgroup cycle stamps of the dK/dV kernel
constexpr int64_t LAB_MAX_WG = 80000;
__device__ unsigned long long g_lab_stamp[LAB_MAX_WG * 4];
__device__ __forceinline__ unsigned long long lab_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#endif
template <int HD_>
__global__ __launch_bounds__(256, HD_ <= 96 ? 2 : 1) void attn_bwd_dkv_kernel(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef ATTN_LAB_STAMP
  const unsigned long long lab_t0 = lab_now();
  unsigned long long lab_t1 = 0, lab_t2 = 0;
#endif
  // two buffers of [Q: NH sub-tiles | dO: NH sub-tiles], stats[2][4][64] floats, 4 patches (NH = 2: in the tile space)
  constexpr int HD = HD_, NH = (HD + 63) / 64, KS = HD / 16, OB = HD / 32, BUF_B = 2 * NH * TILE_B;   // sub-tiles, k-steps, 32-column output blocks
  float* stats = reinterpret_cast<float*>(smem + 2 * BUF_B);
  char* patches = NH == 1 ? smem + 2 * BUF_B + 2 * 4 * 64 * 4 : smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = a.S, H = a.H, D = H * HD;
  const int64_t ld = 3 * (int64_t)D;
  auto process = [&](int kb, int h, int g) __attribute__((always_inline)) {
  const int kb0 = kb * 128;                          // block's first key
  const int key0 = kb0 + wave * 32;                  // wave's first key
  const int mykey = key0 + (lane & 31);
  const int krow = mykey < S ? mykey : S - 1;
  const bf16* base = a.qkv + (int64_t)g * S * ld + h * HD;
  const bf16* dO = a.dout + (int64_t)g * S * D + h * HD;
  const float* lse = a.lse + ((int64_t)g * H + h) * S * 2;
  const float* nmlp = a.delta + ((int64_t)g * H + h) * S;
  const float* ndp = nmlp + a.plane;

  // which kernel serves this block: flags bit 0 of a 64-key tile = some key of it carries a bias (padding, or past S);
  // bit 1 = the whole tile is dead padding (exactly zero dK / dV)
  const int ntile = (S + KV_TILE - 1) / KV_TILE;
  const int* flg = a.flags + (int64_t)g * ntile;
  const int kt_lo = kb0 / KV_TILE, kt_hi = (kb0 + 64 < S) ? kt_lo + 1 : kt_lo;
  bool wave_dead, block_dead, block_plain;
  if (a.masks) {                                       // scalar load: everything below depends on it
    uint64_t sp, sk;
    sload_masks(a.masks + 2 * (int64_t)g, sp, sk);
    wave_dead = key0 < S && ((sk >> (key0 / KV_TILE)) & 1);
    block_dead = ((sk >> kt_lo) & 1) && ((sk >> kt_hi) & 1);
    block_plain = !((sp >> kt_lo) & 1) && !((sp >> kt_hi) & 1);
  } else {
    wave_dead = key0 < S && (flg[key0 / KV_TILE] & 2);
    block_dead = (flg[kt_lo] & 2) && (flg[kt_hi] & 2);
    block_plain = !(flg[kt_lo] & 1) && !(flg[kt_hi] & 1);
  }
  wave_dead = uni(wave_dead);
  block_dead = uni(block_dead);
  block_plain = uni(block_plain && kb0 + 128 <= S);
  if (block_dead) {
    bf16* dkp = a.dqkv + (int64_t)g * S * ld + D + h * HD;
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int idx = tid; idx < 128 * 2 * (HD / 8); idx += 256) {      // 128 keys x (dK | dV) x HD/8 16-byte chunks
      const int row = idx / (2 * (HD / 8)), c = idx % (2 * (HD / 8));
      if (kb0 + row < S) *reinterpret_cast<u32x4*>(dkp + (int64_t)(kb0 + row) * ld + (c / (HD / 8)) * D + (c % (HD / 8)) * 8) = z;
    }
    return;
  }

  // the rest of the kernel exists twice, once per variant: the two never join again, so no accumulator is copied
  auto body = [&](auto PLAIN_T) {
  constexpr bool PLAIN = decltype(PLAIN_T)::value;
  bf16x8 kf[KS], vf[KS];
  {
    const bf16* kp = base + (int64_t)krow * ld + D + 8 * (lane >> 5);
    const bf16* vp = base + (int64_t)krow * ld + 2 * D + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(kp + 16 * ks);
      vf[ks] = *reinterpret_cast<const bf16x8*>(vp + 16 * ks);
    }
  }
  const float c1 = a.scale * LOG2E;
  float bkey = 0.f;                                    // key bias (log2 units; -inf past S); zero by construction when PLAIN
  if (!PLAIN) bkey = mykey < ntile * KV_TILE ? a.bias2[(int64_t)g * ntile * KV_TILE + mykey] : -INFINITY;

  f32x16 dkacc[2 * NH], dvacc[2 * NH];
#pragma unroll
  for (int b = 0; b < OB; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) { dkacc[b][e] = 0.f; dvacc[b][e] = 0.f; }

  const int nt = (S + 63) / 64;
  const int t0 = a.causal ? kb0 / 64 : 0;              // first query tile that can see this block's keys

  // per-tile statistics by DMA too: st[0..127] = interleaved (m, log2 l) pairs of the 64 queries (careful kernel),
  // st[128..191] = -(m + log2 l) (plain kernel), st[192..255] = -delta
  const StageOff soff_q = make_stage_off(ld, S, wave, lane), soff_do = make_stage_off(D, S, wave, lane);
  constexpr int LASTC = HD - 64 * (NH - 1);            // real columns of the last sub-tile (32 when HD = 96)
  const StageOff soff_q2 = make_stage_off(ld, S, wave, lane, LASTC), soff_do2 = make_stage_off(D, S, wave, lane, LASTC);
  auto stage = [&](int t, int buf) {
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
      stage64(base + DH * hf, ld, t * 64, S, smem + buf * BUF_B + hf * TILE_B, wave, lane, hf == NH - 1 ? soff_q2 : soff_q);
      stage64(dO + DH * hf, D, t * 64, S, smem + buf * BUF_B + (NH + hf) * TILE_B, wave, lane, hf == NH - 1 ? soff_do2 : soff_do);
    }
    if (wave == 0) {
      float* st = stats + buf * 256;
      int qd = t * 64 + lane;
      qd = qd < S ? qd : S - 1;
      if (PLAIN) {
        glds4(nmlp + qd, st + 128);
      } else {
        const int last = 2 * S - 1;
        int i0 = t * 128 + lane, i1 = t * 128 + 64 + lane;
        i0 = i0 < last ? i0 : last;
        i1 = i1 < last ? i1 : last;
        glds4(lse + i0, st);
        glds4(lse + i1, st + 64);
      }
      glds4(ndp + qd, st + 192);
    }
  };
  const TrOff troff = make_troff(lane);
  // row-fragment offsets inside a tile for the four k-steps (the XOR swizzle makes them lane-dependent; the 32-row block
  // and the tile are added as uniform terms)
  unsigned foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = (unsigned)((lane & 31) * 128 + (((2 * ks + (lane >> 5)) ^ swz(lane & 31)) << 4));
  const bool wave_live = uni(key0 < S && !wave_dead);

  // one 64-query tile of this wave's 32 keys
  auto tile = [&](int qt0, int buf) {
    const float* st = stats + buf * 256;
    const char* Qb = smem + buf * BUF_B;
    const char* dOb = Qb + NH * TILE_B;
    const int qrow0 = 4 * (lane >> 5);                           // + 32 sq + 8 g4 + e4: local query row of element e
    f32x16 sacc[2], dpacc[2];
    // row fragments of Q and dO for one 32-query block: KS + KS ds_read_b128, issued as one batch
    unsigned fbase[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) fbase[i] = lds_addr(Qb) + foff[i];
    auto frags = [&](auto SQ, bf16x8 (&qv)[KS], bf16x8 (&dv)[KS]) {
      constexpr int sq = decltype(SQ)::value;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (HD != 64) {                                    // wider heads: compiler-managed loads (a batch would cost registers they lack)
          qv[ks] = *reinterpret_cast<const bf16x8*>(Qb + sq * 4096 + (ks >> 2) * TILE_B + foff[ks & 3]);
          dv[ks] = *reinterpret_cast<const bf16x8*>(dOb + sq * 4096 + (ks >> 2) * TILE_B + foff[ks & 3]);
        } else if (ks < 4) {
          qv[ks] = lds_frag<sq * 4096>(fbase[ks & 3]);
          dv[ks] = lds_frag<sq * 4096 + NH * TILE_B>(fbase[ks & 3]);
        } else {
          qv[ks] = lds_frag<sq * 4096 + TILE_B>(fbase[ks & 3]);
          dv[ks] = lds_frag<sq * 4096 + TILE_B + NH * TILE_B>(fbase[ks & 3]);
        }
      }
    };
    // YOUNGER = LDS reads issued after this block's fragments that may still be in flight when its MFMAs start
    auto scores = [&](auto SQ, auto YOUNGER, bf16x8 (&qv)[KS], bf16x8 (&dv)[KS]) {
      constexpr int sq = decltype(SQ)::value;
      constexpr int younger = decltype(YOUNGER)::value;
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[sq][e] = 0.f;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 nd = *reinterpret_cast<const f32x4*>(st + 192 + 32 * sq + qrow0 + 8 * g4);
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) dpacc[sq][g4 * 4 + e4] = nd[e4];
      }
      auto kstep = [&](auto KSI) {
        constexpr int ks = decltype(KSI)::value;
        if constexpr (ks < KS) {
          // fragments were issued in the order q0 d0 q1 d1 ...: when at most `younger` + 2 (KS - 1 - ks) reads are outstanding,
          // this k-step's pair has landed
          if constexpr (HD == 64) lds_wait_frags<(younger + 2 * (KS - 1 - ks) > 15 ? 15 : younger + 2 * (KS - 1 - ks))>(qv[ks], dv[ks]);
          sacc[sq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qv[ks], kf[ks], sacc[sq], 0, 0, 0);
          dpacc[sq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dv[ks], vf[ks], dpacc[sq], 0, 0, 0);
        }
      };
      kstep(std::integral_constant<int, 0>{}); kstep(std::integral_constant<int, 1>{}); kstep(std::integral_constant<int, 2>{});
      kstep(std::integral_constant<int, 3>{}); kstep(std::integral_constant<int, 4>{}); kstep(std::integral_constant<int, 5>{});
      kstep(std::integral_constant<int, 6>{}); kstep(std::integral_constant<int, 7>{});
    };
    auto weights = [&](auto SQ, bf16x8 (&pf)[2], bf16x8 (&dsf)[2]) {
      constexpr int sq = decltype(SQ)::value;
      if (PLAIN) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 nm = *reinterpret_cast<const f32x4*>(st + 128 + 32 * sq + qrow0 + 8 * g4);
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) sacc[sq][g4 * 4 + e4] = __builtin_amdgcn_exp2f(fmaf(sacc[sq][g4 * 4 + e4], c1, nm[e4]));
        }
      } else {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 ml0 = *reinterpret_cast<const f32x4*>(st + 2 * (32 * sq + qrow0 + 8 * g4));       // m0 l0 m1 l1
          const f32x4 ml1 = *reinterpret_cast<const f32x4*>(st + 2 * (32 * sq + qrow0 + 8 * g4) + 4);   // m2 l2 m3 l3
          const float mv[4] = {ml0[0], ml0[2], ml1[0], ml1[2]};
          const float lv[4] = {ml0[1], ml0[3], ml1[1], ml1[3]};
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            const float tv = fmaf(sacc[sq][g4 * 4 + e4], c1, bkey);
            sacc[sq][g4 * 4 + e4] = __builtin_amdgcn_exp2f((tv - mv[e4]) - lv[e4]);
          }
        }
      }
      // the block touches (or lies above) this wave's stretch of the causal diagonal, or runs past the last query: zero the
      // weights of invisible / non-existent queries.  (A branch around vector instructions only: no accumulator is live
      // across it that an MFMA writes.)
      if (uni(((a.causal != 0) & (qt0 + 32 * sq < key0 + 31)) | (qt0 + 32 * sq + 32 > S))) {   // '&', '|': no short-circuit branches
        const int qvis = a.causal ? mykey : -1;          // a query q sees this lane's key iff q >= qvis (and q < S)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int q = qt0 + 32 * sq + qrow0 + 8 * (e >> 2) + (e & 3);
          sacc[sq][e] = (q < qvis || q >= S) ? 0.f : sacc[sq][e];
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) dpacc[sq][e] *= sacc[sq][e];    // dS / scale  (dP was started at -delta)
      acc_to_frags(sacc[sq], pf[0], pf[1]);
      acc_to_frags(dpacc[sq], dsf[0], dsf[1]);
    };
    auto products = [&](auto SQ, const bf16x8 (&pf)[2], const bf16x8 (&dsf)[2]) {
      constexpr int sq = decltype(SQ)::value;
      const unsigned qaddr = lds_addr(Qb) + sq * 4096, doaddr = lds_addr(dOb) + sq * 4096;
#pragma unroll
      for (int hf = 0; hf < NH; ++hf) {
        u32x2 dlo[2][2], dhi[2][2], qlo[2][2], qhi[2][2];
        tr_issue<0>(doaddr + hf * TILE_B, troff, 0, dlo[0][0], dhi[0][0]);
        tr_issue<0>(doaddr + hf * TILE_B, troff, 1, dlo[0][1], dhi[0][1]);
        tr_issue<16>(doaddr + hf * TILE_B, troff, 0, dlo[1][0], dhi[1][0]);
        tr_issue<16>(doaddr + hf * TILE_B, troff, 1, dlo[1][1], dhi[1][1]);
        tr_issue<0>(qaddr + hf * TILE_B, troff, 0, qlo[0][0], qhi[0][0]);
        tr_issue<0>(qaddr + hf * TILE_B, troff, 1, qlo[0][1], qhi[0][1]);
        tr_issue<16>(qaddr + hf * TILE_B, troff, 0, qlo[1][0], qhi[1][0]);
        tr_issue<16>(qaddr + hf * TILE_B, troff, 1, qlo[1][1], qhi[1][1]);
        // the wait names the sixteen results as in/out operands: their consumers depend on it, and nothing else does -- no
        // scheduling fence, so the other block's softmax arithmetic may be placed between the MFMAs that follow
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(dlo[0][0]), "+v"(dhi[0][0]), "+v"(dlo[0][1]), "+v"(dhi[0][1]), "+v"(dlo[1][0]), "+v"(dhi[1][0]), "+v"(dlo[1][1]),
                       "+v"(dhi[1][1]), "+v"(qlo[0][0]), "+v"(qhi[0][0]), "+v"(qlo[0][1]), "+v"(qhi[0][1]), "+v"(qlo[1][0]), "+v"(qhi[1][0]),
                       "+v"(qlo[1][1]), "+v"(qhi[1][1]));
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            if (2 * hf + b < OB) {
              dvacc[2 * hf + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack_tr(dlo[s2][b], dhi[s2][b]), pf[s2], dvacc[2 * hf + b], 0, 0, 0);
              dkacc[2 * hf + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack_tr(qlo[s2][b], qhi[s2][b]), dsf[s2], dkacc[2 * hf + b], 0, 0, 0);
            }
      }
    };
    const std::integral_constant<int, 0> B0{};
    const std::integral_constant<int, 1> B1{};
    bf16x8 pf0[2], dsf0[2], pf1[2], dsf1[2];
    // scheduling fences bound what hipcc may overlap (without them it hoists every LDS read of the tile to the top and spills
    // a hundred registers): [scores 1 | weights 0] and [products 0 | weights 1] are the two regions where one block's vector
    // arithmetic is meant to sit between the other block's MFMAs
    const std::integral_constant<int, 0> NONE{};
    const std::integral_constant<int, 2 * KS> BLOCK{};
    bf16x8 q0[KS], d0[KS], q1[KS], d1[KS];
    if (HD == 64) {
      frags(B0, q0, d0);                                   // every row fragment of the tile up front: one LDS latency per tile
      frags(B1, q1, d1);
      scores(B0, BLOCK, q0, d0);
      __builtin_amdgcn_sched_barrier(0);
      scores(B1, NONE, q1, d1);
      weights(B0, pf0, dsf0);
      __builtin_amdgcn_sched_barrier(0);
      products(B0, pf0, dsf0);
      weights(B1, pf1, dsf1);
      __builtin_amdgcn_sched_barrier(0);
      products(B1, pf1, dsf1);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      // wider heads: one block at a time -- the second block's score / dP accumulators in flight would cost the second
      // wave per SIMD (96) or spill (128)
      frags(B0, q0, d0);
      scores(B0, NONE, q0, d0);
      weights(B0, pf0, dsf0);
      products(B0, pf0, dsf0);
      __builtin_amdgcn_sched_barrier(0);
      frags(B1, q1, d1);
      scores(B1, NONE, q1, d1);
      weights(B1, pf1, dsf1);
      products(B1, pf1, dsf1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

#ifdef ATTN_LAB_STAMP
  lab_t1 = lab_now();
#endif
  if (t0 < nt) {
    stage(t0, 0);
    __syncthreads();
  }
  for (int t = t0; t < nt; ++t) {
    const int buf = (t - t0) & 1;
#ifndef ATTN_LAB_NODMA
    if (t + 1 < nt) stage(t + 1, buf ^ 1);
#endif
    const int qt0 = t * 64;
    if (wave_live) tile(qt0, buf);
#ifndef ATTN_LAB_NOBARRIER
    __syncthreads();
#endif
  }
#ifdef ATTN_LAB_STAMP
  lab_t2 = lab_now();
  auto lab_out = [&]() {
    if (tid == 0) {
      const int64_t wg = blockIdx.x;
      if (wg < LAB_MAX_WG) {
        unsigned long long* o = g_lab_stamp + wg * 4;
        o[0] = lab_t1 - lab_t0; o[1] = lab_t2 - lab_t1; o[2] = lab_now() - lab_t2; o[3] = (unsigned long long)(t0 < nt ? nt - t0 : 0);
      }
    }
  };
  if (key0 >= S) { lab_out(); return; }
#endif
  if (key0 >= S) return;
  // dV first: it needs no rotary tables, so its LDS round trip and stores go out while dK's table rows are on their way
  char* patch = patches + wave * (32 * 144);
#pragma unroll
  for (int hf = 0; hf < NH; ++hf) {
    if (hf) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    store_transposed(reinterpret_cast<const f32x16(&)[2]>(dvacc[2 * hf]), 1.0f, patch, a.dqkv + (int64_t)g * S * ld + 2 * D + h * HD + DH * hf,
                     ld, key0, S, lane, OB - 2 * hf >= 2 ? 2 : 1);
  }
  if (a.rot.ka) rotary_adjoint_regs(reinterpret_cast<f32x16(&)[2]>(dkacc[0]), a.rot.ka, a.rot.kb, a.rot.R, krow, lane);
#pragma unroll
  for (int hf = 0; hf < NH; ++hf) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    store_transposed(reinterpret_cast<const f32x16(&)[2]>(dkacc[2 * hf]), a.scale, patch, a.dqkv + (int64_t)g * S * ld + D + h * HD + DH * hf,
                     ld, key0, S, lane, OB - 2 * hf >= 2 ? 2 : 1);
  }
#ifdef ATTN_LAB_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lab_out();
#endif
  };
  if (block_plain) body(std::true_type{});
  else body(std::false_type{});
  };   // process
  ATTN_PAIR_LOOP(a.nblk, false)
}

constexpr int BWD_DKV_LDS = 4 * TILE_B + 2 * 4 * 64 * 4 + 4 * 32 * 144;
constexpr int FWD_LDS = 4 * TILE_B + 2 * 64 * 4 + 4 * 32 * 144;
// head dim 128 (NH = 2): twice the tile space, the output patches live in it (2 workgroups per CU by LDS)
constexpr int BWD_DKV_LDS2 = 8 * TILE_B + 2 * 4 * 64 * 4;
constexpr int FWD_LDS2 = 8 * TILE_B + 2 * 64 * 4;
__host__ inline bool native_dh(int Dh) { return Dh == 64 || Dh == 96 || Dh == 128; }

}  // namespace

#ifdef ATTN_LAB_STAMP
extern "C" int meant_lab_stamps(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_lab_stamp), bytes < sizeof(g_lab_stamp) ? bytes : sizeof(g_lab_stamp));
}
#endif

// Head dims 64 and 128 run on the MFMA kernels above (128 as two 64-column halves of every tile).  The host side pads
// other head dims below 128 up to 128 with zero columns (meant_amd/ops.py: the reference's default 8 heads -> Dh = 96),
// so this is what they run on too.  Anything else handed to the C ABI directly takes a widening detour: bf16 -> f32
// copies in the workspace, the fp32 attention core, f32 -> bf16.  Correct for any Dh, not fast.
// workspace of the Dh=64 path: [delta: G*H*S floats | bias2: G*nt*64 floats | flags: G*nt ints | masks: G * 2 u64]
static size_t ws_delta_bytes(int64_t G, int64_t S, int H) { return align256((size_t)G * H * S * 2 * sizeof(float)); }   // two planes
static size_t ws_bias_bytes(int64_t G, int64_t S) { return align256((size_t)G * ceil_div(S, KV_TILE) * KV_TILE * sizeof(float)); }
static size_t ws_flag_bytes(int64_t G, int64_t S) { return align256((size_t)G * ceil_div(S, KV_TILE) * sizeof(int)); }
static size_t ws_mask_bytes(int64_t G) { return align256((size_t)G * 2 * sizeof(uint64_t)); }

size_t attn_bf16_ws(int64_t G, int64_t S, int H, int Dh) {
  if (native_dh(Dh)) return ws_delta_bytes(G, S, H) + ws_bias_bytes(G, S) + ws_flag_bytes(G, S) + ws_mask_bytes(G) + attn_bwd1_ws(G, S, H, Dh);
  return 2 * align256((size_t)G * S * 3 * H * Dh * 4) + 2 * align256((size_t)G * S * H * Dh * 4) + attn_f32_ws(G, S, H, Dh);
}

// the forward's own workspace: [bias2 | flags | masks] -- neither the backward's delta planes nor the single-pass backward's
// partial-dQ scratch (1.2 GB at the text shape) belong to it
size_t attn_bf16_fwd_ws(int64_t G, int64_t S, int H, int Dh) {
  if (native_dh(Dh)) return ws_bias_bytes(G, S) + ws_flag_bytes(G, S) + ws_mask_bytes(G);
  return attn_bf16_ws(G, S, H, Dh);
}

static int cast_async(const void* src, int sd, void* dst, int dd, int64_t n, hipStream_t st) { return meant_cast(src, sd, dst, dd, n, st); }

static size_t generic_ws(int64_t G, int64_t S, int H, int Dh) {
  const size_t T = (size_t)G * S, D = (size_t)H * Dh;
  return 2 * align256(T * 3 * D * 4) + 2 * align256(T * D * 4) + attn_f32_ws(G, S, H, Dh);
}

static int attn_bf16_generic(bool backward, const bf16* qkv, const bf16* o, const bf16* dout, bf16* o_out, float* lse, const float* key_mask,
                             bf16* dqkv, int64_t G, int64_t S, int H, int Dh, float scale, int causal, void* ws, size_t ws_bytes,
                             hipStream_t stream, float drop_p = 0.f, uint64_t seed = 0) {
  meant_route_hit(ROUTE_ATTN_GENERIC);
  MEANT_REQUIRE(ws && ws_bytes >= generic_ws(G, S, H, Dh), MEANT_ERR_WORKSPACE, "attn(bf16, Dh=%d): workspace too small", Dh);
  const int64_t T = G * S, D = (int64_t)H * Dh;
  char* w = (char*)ws;
  float* qkv32 = (float*)w; w += align256((size_t)T * 3 * D * 4);
  float* dqkv32 = (float*)w; w += align256((size_t)T * 3 * D * 4);
  float* o32 = (float*)w; w += align256((size_t)T * D * 4);
  float* do32 = (float*)w; w += align256((size_t)T * D * 4);
  const size_t rest = ws_bytes - (size_t)(w - (char*)ws);
  int rc;
  if ((rc = cast_async(qkv, MEANT_BF16, qkv32, MEANT_F32, T * 3 * D, stream))) return rc;
  if (!backward) {
    if ((rc = attn_f32_fwd(qkv32, o32, lse, key_mask, G, S, H, Dh, scale, causal, w, rest, stream, drop_p, seed))) return rc;
    return cast_async(o32, MEANT_F32, o_out, MEANT_BF16, T * D, stream);
  }
  if ((rc = cast_async(dout, MEANT_BF16, do32, MEANT_F32, T * D, stream))) return rc;
  if ((rc = attn_f32_bwd(qkv32, o32, do32, lse, key_mask, dqkv32, G, S, H, Dh, scale, causal, w, rest, stream, drop_p, seed))) return rc;
  (void)o;
  return cast_async(dqkv32, MEANT_F32, dqkv, MEANT_BF16, T * 3 * D, stream);
}

size_t attn_drop_ws(int64_t G, int64_t S, int H, int Dh, int dtype) {
  return dtype == MEANT_F32 ? attn_f32_ws(G, S, H, Dh) : generic_ws(G, S, H, Dh);
}

int attn_drop_bf16(bool backward, const bf16* qkv, const bf16* o, const bf16* dout, bf16* o_out, float* lse, const float* key_mask, bf16* dqkv,
                   int64_t G, int64_t S, int H, int Dh, float scale, int causal, float drop_p, uint64_t seed, void* ws, size_t ws_bytes,
                   hipStream_t stream) {
  return attn_bf16_generic(backward, qkv, o, dout, o_out, lse, key_mask, dqkv, G, S, H, Dh, scale, causal, ws, ws_bytes, stream, drop_p, seed);
}

static int attn_bf16_check(const char* name, int64_t G, int64_t S, int H, int Dh) {
  (void)Dh;
  MEANT_REQUIRE(G <= 65535 && H <= 65535 && S < (1 << 24) && ceil_div(S, 128) * H * G < (1LL << 31) - 8, MEANT_ERR_UNSUPPORTED,
                "%s: G=%lld / H=%d / S=%lld exceed the grid limits", name, (long long)G, H, (long long)S);
  return MEANT_OK;
}

int attn_bf16_fwd(const bf16* qkv, bf16* o, float* lse, const float* key_mask, int64_t G, int64_t S, int H, int Dh, float scale,
                  int causal, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!native_dh(Dh)) return attn_bf16_generic(false, qkv, nullptr, nullptr, o, lse, key_mask, nullptr, G, S, H, Dh, scale, causal, ws, ws_bytes, stream);
  if (attn_short_ok(S, Dh)) return attn_short_fwd(qkv, o, lse, key_mask, G, S, H, Dh, scale, causal, stream);
  int rc = attn_bf16_check("attn_fwd", G, S, H, Dh);
  if (rc) return rc;
  MEANT_REQUIRE(meant_aligned16(qkv) && meant_aligned16(o), MEANT_ERR_ARG, "attn_fwd: 16-byte alignment");
  MEANT_REQUIRE(ws && ws_bytes >= attn_bf16_fwd_ws(G, S, H, Dh), MEANT_ERR_WORKSPACE, "attn_fwd: workspace too small");
  const int nt = (int)ceil_div(S, KV_TILE);
  float* bias2 = (float*)ws;
  int* flags = (int*)((char*)bias2 + ws_bias_bytes(G, S));
  hipLaunchKernelGGL(attn_prep_mask_kernel, dim3((unsigned)nt, (unsigned)G), dim3(64), 0, stream, key_mask, bias2, flags, (int)S, nt, causal, (unsigned*)nullptr);
  MEANT_LAUNCH_CHECK("attn_prep_mask");
  uint64_t* masks = nullptr;
  if (nt <= 64) {
    masks = (uint64_t*)((char*)flags + ws_flag_bytes(G, S));
    hipLaunchKernelGGL(attn_pack_flags_kernel, dim3((unsigned)G), dim3(64), 0, stream, flags, masks, nt);
    MEANT_LAUNCH_CHECK("attn_pack_flags");
  }
  const int nqb = (int)ceil_div(S, 128);
  FwdArgs a{qkv, o, lse, bias2, flags, masks, (int)S, H, scale, causal, (int)G, nqb};
  MEANT_RAISE_LDS(attn_fwd_kernel<64>, FWD_LDS);
  MEANT_RAISE_LDS(attn_fwd_kernel<96>, FWD_LDS2);
  MEANT_RAISE_LDS(attn_fwd_kernel<128>, FWD_LDS2);
  if (Dh != DH) {                                      // 96 and 128: two 64-column sub-tiles per operand tile
    const dim3 grid2(attn_grid(causal ? (nqb + 1) / 2 : nqb, H, G));
    if (Dh == 96) {
      meant_route_hit(ROUTE_ATTN_FWD_D96);
      hipLaunchKernelGGL(attn_fwd_kernel<96>, grid2, dim3(256), FWD_LDS2, stream, a);
    } else {
      meant_route_hit(ROUTE_ATTN_FWD_D128);
      hipLaunchKernelGGL(attn_fwd_kernel<128>, grid2, dim3(256), FWD_LDS2, stream, a);
    }
    MEANT_LAUNCH_CHECK("attn_fwd");
    return MEANT_OK;
  }
  meant_route_hit(ROUTE_ATTN_FWD);
  hipLaunchKernelGGL(attn_fwd_kernel<64>, dim3(attn_grid(causal ? (nqb + 1) / 2 : nqb, H, G)), dim3(256), FWD_LDS, stream, a);
  MEANT_LAUNCH_CHECK("attn_fwd");
  return MEANT_OK;
}

int attn_bf16_bwd(const bf16* qkv, const bf16* o, const bf16* dout, const float* lse, const float* key_mask, bf16* dqkv, int64_t G,
                  int64_t S, int H, int Dh, float scale, int causal, RotTables rot, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!native_dh(Dh)) {
    int rcg = attn_bf16_generic(true, qkv, o, dout, nullptr, const_cast<float*>(lse), key_mask, dqkv, G, S, H, Dh, scale, causal, ws, ws_bytes, stream);
    if (rcg || !rot.qa) return rcg;
    return meant_rotary_qk(dqkv, G * S, S, H, Dh, rot.R, rot.qa, rot.qb, rot.ka, rot.kb, 1, MEANT_BF16, stream);
  }
  if (attn_short_ok(S, Dh) && (!rot.qa || (rot.R % 8 == 0 && rot.R <= DH)))
    return attn_short_bwd(qkv, dout, lse, key_mask, dqkv, G, S, H, Dh, scale, causal, rot, stream);
  MEANT_REQUIRE(!rot.qa || (rot.R % 8 == 0 && rot.R <= DH), MEANT_ERR_UNSUPPORTED, "attn_bwd: rotary dim must be a multiple of 8 and <= 64");
  int rc = attn_bf16_check("attn_bwd", G, S, H, Dh);
  if (rc) return rc;
  MEANT_REQUIRE(ws && ws_bytes >= attn_bf16_ws(G, S, H, Dh), MEANT_ERR_WORKSPACE, "attn_bwd: workspace too small");
  MEANT_REQUIRE(meant_aligned16(qkv) && meant_aligned16(o) && meant_aligned16(dout) && meant_aligned16(dqkv), MEANT_ERR_ARG,
                "attn_bwd: 16-byte alignment");
  const int nt = (int)ceil_div(S, KV_TILE);
  float* bias2 = (float*)((char*)ws + ws_delta_bytes(G, S, H));
  int* flags = (int*)((char*)bias2 + ws_bias_bytes(G, S));
  const bool one_pass = nt <= 64 && attn_bwd1_ok(S, Dh, causal);   // S and dP computed once, dQ through LDS (attn_bwd1.hip)
  void* ws1 = (char*)flags + ws_flag_bytes(G, S) + ws_mask_bytes(G);   // its item counters (zeroed by the mask kernel) and partial dQ blocks
  hipLaunchKernelGGL(attn_prep_mask_kernel, dim3((unsigned)nt, (unsigned)G), dim3(64), 0, stream, key_mask, bias2, flags, (int)S, nt, causal,
                     one_pass ? (unsigned*)ws1 : (unsigned*)nullptr);
  MEANT_LAUNCH_CHECK("attn_prep_mask");
  uint64_t* masks = nullptr;
  if (nt <= 64) {
    masks = (uint64_t*)((char*)flags + ws_flag_bytes(G, S));
    hipLaunchKernelGGL(attn_pack_flags_kernel, dim3((unsigned)G), dim3(64), 0, stream, flags, masks, nt);
    MEANT_LAUNCH_CHECK("attn_pack_flags");
  }
  if (one_pass) return attn_bwd1_launch(qkv, o, dout, lse, bias2, flags, masks, dqkv, ws1, G, S, H, scale, causal, rot, stream);
  BwdArgs a{qkv, o, dout, lse, bias2, flags, dqkv, (float*)ws, masks, G * (int64_t)H * S, (int)S, H, scale, causal, (int)G, (int)ceil_div(S, 128), rot};
  MEANT_RAISE_LDS(attn_bwd_dq_kernel<64>, FWD_LDS);
  MEANT_RAISE_LDS(attn_bwd_dkv_kernel<64>, BWD_DKV_LDS);
  MEANT_RAISE_LDS(attn_bwd_dq_kernel<96>, FWD_LDS2);
  MEANT_RAISE_LDS(attn_bwd_dkv_kernel<96>, BWD_DKV_LDS2);
  MEANT_RAISE_LDS(attn_bwd_dq_kernel<128>, FWD_LDS2);
  MEANT_RAISE_LDS(attn_bwd_dkv_kernel<128>, BWD_DKV_LDS2);
  const int64_t nblk = ceil_div(S, 128);
  const dim3 grid(attn_grid(causal ? (nblk + 1) / 2 : nblk, H, G));   // causal: one workgroup per pair of 128-row blocks
  meant_route_hit(ROUTE_ATTN_BWD);
  if (Dh == 96) {
    meant_route_hit(ROUTE_ATTN_BWD_D96);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<96>, grid, dim3(256), FWD_LDS2, stream, a);
    MEANT_LAUNCH_CHECK("attn_bwd_dq");
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<96>, grid, dim3(256), BWD_DKV_LDS2, stream, a);
    MEANT_LAUNCH_CHECK("attn_bwd_dkv");
    return MEANT_OK;
  }
  if (Dh == 128) {
    meant_route_hit(ROUTE_ATTN_BWD_D128);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<128>, grid, dim3(256), FWD_LDS2, stream, a);
    MEANT_LAUNCH_CHECK("attn_bwd_dq");
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<128>, grid, dim3(256), BWD_DKV_LDS2, stream, a);
    MEANT_LAUNCH_CHECK("attn_bwd_dkv");
    return MEANT_OK;
  }
  hipLaunchKernelGGL(attn_bwd_dq_kernel<64>, grid, dim3(256), FWD_LDS, stream, a);
  MEANT_LAUNCH_CHECK("attn_bwd_dq");
  hipLaunchKernelGGL(attn_bwd_dkv_kernel<64>, grid, dim3(256), BWD_DKV_LDS, stream, a);
  MEANT_LAUNCH_CHECK("attn_bwd_dkv");
  return MEANT_OK;
}
