// K4': single-pass attention backward for the bf16 tier, head dim 64, sequences one workgroup can own
// (S <= 256, or S <= 512 under the causal mask).  Same semantics as the two-pass form in attn_bf16.hip
// (meant/attention.py:43-57, meant/xPosAttention.py:41-63: eager softmax(QK^T * scale + mask) V, differentiated), same
// layouts, same statistics, same rotary adjoint in the epilogue.
//
// Why: the two-pass form computes S = QK^T and dP = dO V^T twice (once per pass: 7 products of S x S x 64 and two
// exponentials per score where the algebra needs 5 and one), because the weights come out of the MFMA with either the key
// or the query on the lane, and dK/dV want the one, dQ the other.  Here they are computed once, key on the lane, feed dV and
// dK from registers as before, and dS additionally goes through LDS (written as [key][query] rows, read back by
// ds_read_b64_tr_b16, i.e. transposed by the hardware) into the dQ product.
//
// One workgroup of 8 waves (one per CU) is persistent: it draws (group, head) items from its XCD's counter and works on an item in
// STAGES of one key half (256 keys; wave w: keys 32 w .. 32 w + 31 of the half, K and V fragments and the dK / dV accumulators in
// registers for the whole stage), a stage in CHUNKS of 128 queries, a chunk as two 64-query TILES that come HBM -> LDS by DMA:
//   phase 1a / 1b   every wave: its 32 keys x the tile in ring slot 0, then slot 1: S, dP, P, dS, dV^T += dO^T P, dK^T += Q^T dS (the
//                   tile body of the two-pass dK/dV kernel), and dS as bf16 into the wave's own 8 KiB region of the panel [256 keys][128
//                   queries].  Barrier M after 1a (slot 0 is refilled behind it: next chunk's first tile), barrier Y after 1b;
//   phase 2         wave w: the 32 x 32 block (queries 32 (w >> 1).., head columns 32 (w & 1)..) of dQ^T = K^T dS^T over the half's
//                   live key blocks: both operands by transposed reads, K from its LDS image, dS from the panel; one asm statement per
//                   pair of key blocks (see there why).  Then the statistics of the next tile and the request for the one after;
//   barrier X'; dQ block: rotary adjoint (rows requested at the start of phase 2), scale, bf16, stored through the wave's panel region.
// Every tile arrives as [Q | dO | O], each wave requesting its own 1 KiB piece of each, so that the rows' statistics (delta = rowsum(dO o
// O), -(m + log2 l): there is no dQ pass that could leave them behind) are formed from LDS by the wave that requested those rows, behind
// nothing but its own vmcnt wait.  Requests go out one phase before their data is needed and are waited for (vmcnt(0)) just before
// the next ones go out: the counter is in order, a young request must never be waited for behind an old one.
// Causal sequences longer than 256: the upper key half (which only queries >= 256 see) goes first and leaves its dQ blocks
// as fp32 in a scratch buffer, in accumulator order, which the same wave adds when the lower half reaches that chunk -- 64 KiB per
// (group, head), written and read back by the same CU within microseconds.  While a stage is in its last chunk and its epilogues, the
// next stage's K half and first two tiles are already on their way.
// Per wave the variant of the arithmetic (PLAIN: all 32 keys live, or careful: the reference's step by step) is fixed for
// a stage, as in the two-pass kernel it was per workgroup.
// LDS: K half 32 KiB | two ring slots of [Q | dO | O] 48 KiB | panel 64 KiB | statistics 2 KiB | item mailbox = 146 KiB: one workgroup
// per CU, two waves per SIMD (256 registers each: the tile body is at that limit, which is why per-lane offsets pass through opaque
// copies at their use sites -- anything the compiler can hoist out of the chunk loop it will, and spill).
#include "attn_tiles.h"
#include <type_traits>

static size_t align256_(size_t x) { return (x + 255) & ~(size_t)255; }

namespace {

constexpr int B1_K = 0;                       // K half: 4 tiles of 64 keys
constexpr int B1_RING = 4 * TILE_B;           // two slots of one 64-query tile each: [Q | dO | O]
constexpr int B1_SLOT = 3 * TILE_B;
constexpr int B1_PANEL = 10 * TILE_B;         // dS panel: 8 wave regions of [query sub-tile 0: 32 keys x 128 B][sub-tile 1]
constexpr int B1_STATS = 18 * TILE_B;         // [slot][0..127: (m, log2 l) pairs | 128..191: -(m + log2 l) | 192..255: -delta]
constexpr int B1_JBOX = 18 * TILE_B + 2 * 256 * 4; // two words: the next item's index, handed from wave 0 to the others
constexpr int B1_LDS = B1_JBOX + 64;
constexpr int B1_REGION = 8192;
#ifndef B1_PIPE
#define B1_PIPE 2
#endif
#ifndef B1_PAIR
#define B1_PAIR 1
#endif
#ifndef B1_PRIO
#define B1_PRIO 1
#endif

struct Bwd1Args { BwdArgs b; float* part; unsigned* ctr; };   // ctr: per-XCD item counters (16 words apart), zero at launch
#ifdef ATTN_LAB_STAMP
// lab build only (tools/lab/stamp_bwd1.py): per-workgroup s_memtime sums of wave 0:
// prologue | phase 1 | wait at the barrier after it | phase 2 | wait at the barrier after it | dQ epilogue | dK dV epilogue | chunks
constexpr int64_t LAB1_MAX_WG = 20000;
__device__ unsigned long long g_lab1_stamp[LAB1_MAX_WG * 16];
__device__ __forceinline__ unsigned long long lab1_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define LAB1(i) { const unsigned long long n__ = lab1_now(); lab_acc[i] += n__ - lab_t; lab_t = n__; }
#else
#define LAB1(i)
#endif

// one 32 (head columns) x 32 (tokens) block, transposed accumulator -> global rows of 32 bf16 through a per-wave patch of 32 x 80 B
__device__ __forceinline__ void store_transposed1(const f32x16& acc, float mult, char* patch, bf16* __restrict__ dst, int64_t ld, int tok0,
                                                   int ntok, int lane) {
  const int t = lane & 31;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    bf16x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (bf16)(acc[g4 * 4 + e] * mult);
    const int dh = 8 * g4 + 4 * (lane >> 5);
    *reinterpret_cast<bf16x4*>(patch + t * 80 + dh * 2) = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = i * 64 + lane;                 // 32 rows x 4 chunks
    const int r = idx >> 2, c = idx & 3;
    const u32x4 v = *reinterpret_cast<const u32x4*>(patch + r * 80 + c * 16);
    if (tok0 + r < ntok) *reinterpret_cast<u32x4*>(dst + (int64_t)(tok0 + r) * ld + c * 8) = v;
  }
}

// rotary adjoint (see rotary_adjoint_regs) on one 32-column block starting at head column dh0
__device__ __forceinline__ void rotary_adjoint_1(f32x16& acc, const float* __restrict__ A, const float* __restrict__ B, int R, int pos,
                                                 int lane, int dh0) {
  const float* Ar = A + (int64_t)pos * R;
  const float* Br = B + (int64_t)pos * R;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int dh = dh0 + 8 * g4 + 4 * (lane >> 5);
    if (dh < R) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(Ar + dh);
      const f32x4 bv = *reinterpret_cast<const f32x4*>(Br + dh);
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const float d0 = acc[g4 * 4 + e], d1 = acc[g4 * 4 + e + 1];
        acc[g4 * 4 + e] = av[e] * d0 + bv[e + 1] * d1;
        acc[g4 * 4 + e + 1] = av[e + 1] * d1 - bv[e] * d0;
      }
    }
  }
}

// one 1 KiB piece (8 rows) of a 64 x 64 tile: the lane's row-in-tile and swizzled 16-byte column (see make_stage_off, which
// does the same for a wave's pair of pieces)
struct PieceOff { unsigned full, last; };
__device__ __forceinline__ PieceOff make_piece_off(int64_t ld, int nrows, int piece, int lane) {
  const int last0 = ((nrows - 1) / KV_TILE) * KV_TILE;
  const int r = piece * 8 + (lane >> 3);
  const int c = (lane & 7) ^ swz(r);
  const int rl = min(r, nrows - 1 - last0);                        // (v_min with a scalar operand: a select would park the scalar in a VGPR)
  return {(unsigned)(r * ld + c * 8), (unsigned)(rl * ld + c * 8)};
}
__device__ __forceinline__ void stage_piece(const bf16* __restrict__ g, int64_t ld, int row0, int nrows, char* tile, int piece,
                                            const PieceOff& po) {
  const bf16* origin = g + (int64_t)row0 * ld;                       // uniform
  const bool ragged = row0 + KV_TILE > nrows;                        // uniform
  glds16(origin + (ragged ? po.last : po.full), tile + piece * 1024);
}
// workgroup barrier with the waits chosen by the caller (the compiler's own __syncthreads() waits for vmcnt(0), which would
// drain a tile requested a moment ago); the "memory" clobber keeps the compiler from moving LDS / global accesses across it
#define B1_BARRIER(WAITS) asm volatile("s_waitcnt " WAITS "\n\ts_barrier" ::: "memory")

__global__ __launch_bounds__(512) void attn_bwd1_kernel(Bwd1Args args) {
  const BwdArgs& a = args.b;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = 4;
  float* stats = reinterpret_cast<float*>(smem + B1_STATS);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = a.S, H = a.H, D = H * DH;
  const int64_t ld = 3 * (int64_t)D;
  const int ntile = (S + KV_TILE - 1) / KV_TILE, nhalf = (S + 255) >> 8, nchunk = (S + 127) >> 7;
  const float c1 = a.scale * LOG2E;
  const unsigned troff0 = make_troff(lane).o[0][0];         // the other three: constant bit flips of this one (see tile())
  const unsigned foff0 = (unsigned)((lane & 31) * 128 + (((lane >> 5) ^ swz(lane & 31)) << 4));
  char* myreg = smem + B1_PANEL + wave * B1_REGION;        // this wave's panel region, and its patch once the panel is done with
  const unsigned prow = lds_addr(myreg) + (lane & 31) * 128 + 8 * (lane >> 5);
  const unsigned pswz = (unsigned)swz(lane & 31) << 4;
  // A workgroup is persistent: it walks its share of the (group, head) items of its XCD (XCD x owns the contiguous item range
  // [x * chunk, (x + 1) * chunk), see attn_item), one STAGE = one key half of one item after the other, so that everything the
  // next stage's first phase needs is requested while the current stage is in its last chunk and its epilogues.
  struct Stg {                                               // all wave-uniform.  Element offsets into the kernel's buffers, not pointers:
    int64_t qo, oo, lo, bo, po;                              // a pointer that has been through this struct is a generic one to the compiler
    uint64_t sp, sk;                                         // (flat_load / flat_store, which also count in lgkmcnt)
    int hk; unsigned j; int valid;                           // sp / sk: per 64-key tile: needs the bias / tail path; dead padding
  };
  // Items are handed out dynamically, per XCD: wave 0 draws the index of the item AFTER the next one from its XCD's counter while
  // the current item is being worked on (so the atomic's round trip is never waited for), and passes the next one's to the other
  // waves through an LDS word published by the stage-start barrier.
  const unsigned nitem = (unsigned)H * a.G, ichunk = (nitem + 7) / 8, xcd = blockIdx.x & 7;
  unsigned* const ctr = args.ctr + xcd * 16;
  unsigned* const jbox = reinterpret_cast<unsigned*>(smem + B1_JBOX);   // (not volatile: a volatile access stays a flat one; the barriers' memory clobbers order these)
  unsigned jpend = 0;                                        // wave 0, lane 0: the item index drawn ahead
  unsigned nit = 0;                                          // items opened so far (parity selects the jbox word)
  auto draw = [&]() { return __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto open_item = [&](unsigned j) -> Stg {
    Stg t{0, 0, 0, 0, 0, 0, 0, 0, j, 0};
    const unsigned item = xcd * ichunk + j;
    t.valid = j < ichunk && item < nitem;
    if (t.valid) {
      const int g = (int)(item / H), h = (int)(item % H);
      t.qo = (int64_t)g * S * ld + h * DH;                   // q|k|v and dq|dk|dv: Q columns of this head; K at + D, V at + 2 D
      t.oo = (int64_t)g * S * D + h * DH;                    // o and do
      t.lo = ((int64_t)g * H + h) * S * 2;                   // (m, log2 l) pairs
      t.bo = (int64_t)g * ntile * KV_TILE;                   // key bias
      t.po = (((int64_t)g * H + h) * 2 * 8 + wave) * 1024;   // partial dQ blocks of this wave: + (c - 2) * 8192 + i * 256 + 4 lane
      sload_masks(a.masks + 2 * (int64_t)g, t.sp, t.sk);
      t.hk = nhalf - 1;
    }
    return t;
  };
  auto upper_dead = [&](uint64_t sk) {                       // every tile of the upper key half is dead padding
    bool dead = nhalf == 2;
    for (int t = 4; t < 8 && t < ntile; ++t) dead = dead && ((sk >> t) & 1);
    return dead;
  };
  // A dead upper half is not a stage: it leaves no partial dQ, and its dK / dV rows are zeroed by the lower half's stage.
  // (The lower half always runs: it is the one that writes dQ.)
  // (every field of a stage descriptor is wave-uniform by construction; saying so keeps the address arithmetic on the scalar unit)
  auto uni64 = [](uint64_t v) {
    return ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
  };
  auto uni_stage = [&](Stg t) {
    t.qo = (int64_t)uni64((uint64_t)t.qo); t.oo = (int64_t)uni64((uint64_t)t.oo); t.lo = (int64_t)uni64((uint64_t)t.lo);
    t.bo = (int64_t)uni64((uint64_t)t.bo); t.po = (int64_t)uni64((uint64_t)t.po); t.sp = uni64(t.sp); t.sk = uni64(t.sk);
    t.hk = __builtin_amdgcn_readfirstlane(t.hk); t.j = (unsigned)__builtin_amdgcn_readfirstlane((int)t.j); t.valid = __builtin_amdgcn_readfirstlane(t.valid);
    return t;
  };
  auto open_next = [&](unsigned j) -> Stg {
    Stg t = open_item(j);
    if (t.valid && upper_dead(t.sk)) t.hk = 0;               // (the lower-half stage zeroes the dead half's dK / dV rows on its way out)
    return uni_stage(t);
  };
#ifdef ATTN_LAB_STAMP
  unsigned long long lab_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long lab_t = lab1_now();
#endif

  // request the tq-th 64-query tile of chunk c into ring slot tq: this wave's piece (rows 8 w .. 8 w + 7) of Q, dO and O, and
  // the (m, log2 l) pairs of those rows.  The lane-derived offsets are recomputed here on purpose (from a value the compiler
  // cannot see through): hoisted out of the chunk loop they would sit in registers across phase 1, which has none to spare.
  auto issue_tile = [&](const Stg& P, int c, int tq) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int row0 = c * 128 + tq * 64;
    char* slot = smem + B1_RING + tq * B1_SLOT;
    const PieceOff pq = make_piece_off(ld, S, wave, ln), pd = make_piece_off(D, S, wave, ln);
    stage_piece(a.qkv + P.qo, ld, row0, S, slot, wave, pq);
    stage_piece(a.dout + P.oo, D, row0, S, slot + TILE_B, wave, pd);
    stage_piece(a.o + P.oo, D, row0, S, slot + 2 * TILE_B, wave, pd);
    if (ln < 16) {
      int i = 2 * (row0 + 8 * wave) + ln;
      i = min(i, 2 * S - 1);
      glds4(a.lse + P.lo + i, stats + tq * 256 + 16 * wave);
    }
  };
  // K half of a stage: piece w of each of its (up to) four tiles
  auto issue_K = [&](const Stg& P) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const PieceOff pk = make_piece_off(ld, S, wave, ln);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
      if (P.hk * 256 + 64 * kt < S) stage_piece(a.qkv + P.qo + D, ld, P.hk * 256 + 64 * kt, S, smem + B1_K + kt * TILE_B, wave, pk);
  };
  // K / V fragments (B operands of the score products) and the key bias of a stage
  auto load_kv = [&](const Stg& P, bf16x8 (&kf)[KS], bf16x8 (&vf)[KS], float& bkey) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int mykey = P.hk * 256 + wave * 32 + (ln & 31);
    const bf16* kp = a.qkv + P.qo + (int64_t)min(mykey, S - 1) * ld + D + 8 * (ln >> 5);
    const bf16* vp = kp + D;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(kp + 16 * ks);
      vf[ks] = *reinterpret_cast<const bf16x8*>(vp + 16 * ks);
    }
    bkey = mykey < ntile * KV_TILE ? a.bias2[P.bo + mykey] : -INFINITY;   // log2 units; -inf past S; zero for every key of a PLAIN wave
  };
  // everything the first phase of a stage needs
  auto request_stage = [&](const Stg& P) {
    const int c0 = a.causal ? 2 * P.hk : 0;
    issue_K(P);
    issue_tile(P, c0, 0);
    if (c0 * 128 + 64 < S) issue_tile(P, c0, 1);
  };
  // statistics of the tile in slot tq, from LDS, for the 8 rows this wave requested itself (its own vmcnt wait covers them):
  // delta = rowsum(dO o O) (8 lanes per row), -(m + log2 l)
  auto stat_step = [&](int tq) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const char* slot = smem + B1_RING + tq * B1_SLOT;
    float* st = stats + tq * 256;
    const int r = 8 * wave + (ln >> 3);
    const int off = r * 128 + (((ln & 7) ^ swz(r)) << 4);
    const bf16x8 dv = *reinterpret_cast<const bf16x8*>(slot + TILE_B + off);
    const bf16x8 ov = *reinterpret_cast<const bf16x8*>(slot + 2 * TILE_B + off);
    const float2 ml = *reinterpret_cast<const float2*>(st + 2 * r);      // (every lane of the row: one LDS round trip, not two)
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) d += (float)dv[j] * (float)ov[j];
    // sum over the row's 8 lanes in the DPP network (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror), not through ds_bpermute
    d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0xB1, 0xf, 0xf, true));
    d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x4E, 0xf, 0xf, true));
    d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x141, 0xf, 0xf, true));
    if ((ln & 7) == 0) {
      st[128 + r] = -(ml.x + ml.y);
      st[192 + r] = -d;
    }
  };

  auto half = [&](const Stg& cur, Stg& nxt, auto PLAIN_T) {
    constexpr bool PLAIN = decltype(PLAIN_T)::value;
    bf16* dqkv = a.dqkv + cur.qo;
    float* part_w = args.part + cur.po;
    const uint64_t sk = cur.sk;
    const int hk = cur.hk;
    const bool part_valid = hk == 0 && nhalf == 2 && !upper_dead(sk);
    const int kh0 = hk * 256, key0 = kh0 + wave * 32, mykey = key0 + (lane & 31);
    const bool wave_live = uni(key0 < S && !((sk >> (key0 >> 6)) & 1));
    unsigned deadm = 0;                                      // bit kb: keys kh0 + 32 kb .. + 31 contribute nothing (dead padding / past S)
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      const int k0 = kh0 + 32 * kb;
      if (k0 >= S || ((sk >> (k0 >> 6)) & 1)) deadm |= 1u << kb;
    }
    const int c0 = a.causal ? 2 * hk : 0, c1l = nchunk - 1;
    bf16x8 kf[KS], vf[KS];                                   // K / V fragments of this wave's 32 keys (B operands of the score products)
    float bkey;
    load_kv(cur, kf, vf, bkey);
    f32x16 dkacc[2], dvacc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) { dkacc[b][e] = 0.f; dvacc[b][e] = 0.f; }

    // ---- phase 1 body: the 64-query tile in ring slot tq against this wave's 32 keys ----
    auto tile = [&](int qt0, int tq) {
      // the per-lane offsets pass through an opaque copy at every use site: what is derived from them (some 50 LDS addresses
      // per tile, each one add or xor away) stays inside the loop instead of being hoisted to the top of the kernel and spilled
      unsigned pswz_l = pswz, prow_l = prow, fo0 = foff0, tr0 = troff0;
      asm volatile("" : "+v"(pswz_l), "+v"(prow_l), "+v"(fo0), "+v"(tr0));
      // the other three row-fragment offsets and transposed-read offsets differ from the first by constant bit flips (the
      // swizzle is an XOR of address bits 4-6 and every tile starts on a 128-byte boundary): one register each, not four
      const TrOff troff = {{{tr0, tr0 ^ 64u}, {tr0 ^ (1024u | 32u), tr0 ^ (1024u | 32u | 64u)}}};
      const float* st = stats + tq * 256;
      const char* Qb = smem + B1_RING + tq * B1_SLOT;
      const char* dOb = Qb + TILE_B;
      const unsigned pbase = prow_l + tq * 4096;
      const unsigned pswz = pswz_l;
      const int qrow0 = 4 * (lane >> 5);
      f32x16 sacc[2], dpacc[2];
      const unsigned fb0 = lds_addr(Qb) + fo0;
      const unsigned fbase[4] = {fb0, fb0 ^ 32u, fb0 ^ 64u, fb0 ^ 96u};
      auto frags = [&](auto SQ, bf16x8 (&qv)[KS], bf16x8 (&dv)[KS]) {
        constexpr int sq = decltype(SQ)::value;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          qv[ks] = lds_frag<sq * 4096>(fbase[ks]);
          dv[ks] = lds_frag<sq * 4096 + TILE_B>(fbase[ks]);
        }
      };
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
          lds_wait_frags<(younger + 2 * (KS - 1 - ks) > 15 ? 15 : younger + 2 * (KS - 1 - ks))>(qv[ks], dv[ks]);
          sacc[sq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qv[ks], kf[ks], sacc[sq], 0, 0, 0);
          dpacc[sq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dv[ks], vf[ks], dpacc[sq], 0, 0, 0);
        };
        kstep(std::integral_constant<int, 0>{}); kstep(std::integral_constant<int, 1>{});
        kstep(std::integral_constant<int, 2>{}); kstep(std::integral_constant<int, 3>{});
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
            const f32x4 ml0 = *reinterpret_cast<const f32x4*>(st + 2 * (32 * sq + qrow0 + 8 * g4));
            const f32x4 ml1 = *reinterpret_cast<const f32x4*>(st + 2 * (32 * sq + qrow0 + 8 * g4) + 4);
            const float mv[4] = {ml0[0], ml0[2], ml1[0], ml1[2]};
            const float lv[4] = {ml0[1], ml0[3], ml1[1], ml1[3]};
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
              const float tv = fmaf(sacc[sq][g4 * 4 + e4], c1, bkey);
              sacc[sq][g4 * 4 + e4] = __builtin_amdgcn_exp2f((tv - mv[e4]) - lv[e4]);
            }
          }
        }
        if (uni(((a.causal != 0) & (qt0 + 32 * sq < key0 + 31)) | (qt0 + 32 * sq + 32 > S))) {
          const int qvis = a.causal ? mykey : -1;
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
        // dS^T rows of this lane's key into the panel: 4 x 4 consecutive queries (8 bytes each)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const u32x4 w = __builtin_bit_cast(u32x4, dsf[g4 >> 1]);
          const u32x2 v = {w[2 * (g4 & 1)], w[2 * (g4 & 1) + 1]};
          const unsigned addr = pbase + ((unsigned)((4 * sq + g4) << 4) ^ pswz);
          asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
        }
      };
      auto products = [&](auto SQ, const bf16x8 (&pf)[2], const bf16x8 (&dsf)[2]) {
        constexpr int sq = decltype(SQ)::value;
        const unsigned qaddr = lds_addr(Qb) + sq * 4096, doaddr = lds_addr(dOb) + sq * 4096;
        u32x2 dlo[2][2], dhi[2][2], qlo[2][2], qhi[2][2];
        tr_issue<0>(doaddr, troff, 0, dlo[0][0], dhi[0][0]);
        tr_issue<0>(doaddr, troff, 1, dlo[0][1], dhi[0][1]);
        tr_issue<16>(doaddr, troff, 0, dlo[1][0], dhi[1][0]);
        tr_issue<16>(doaddr, troff, 1, dlo[1][1], dhi[1][1]);
        tr_issue<0>(qaddr, troff, 0, qlo[0][0], qhi[0][0]);
        tr_issue<0>(qaddr, troff, 1, qlo[0][1], qhi[0][1]);
        tr_issue<16>(qaddr, troff, 0, qlo[1][0], qhi[1][0]);
        tr_issue<16>(qaddr, troff, 1, qlo[1][1], qhi[1][1]);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(dlo[0][0]), "+v"(dhi[0][0]), "+v"(dlo[0][1]), "+v"(dhi[0][1]), "+v"(dlo[1][0]), "+v"(dhi[1][0]), "+v"(dlo[1][1]),
                       "+v"(dhi[1][1]), "+v"(qlo[0][0]), "+v"(qhi[0][0]), "+v"(qlo[0][1]), "+v"(qhi[0][1]), "+v"(qlo[1][0]), "+v"(qhi[1][0]),
                       "+v"(qlo[1][1]), "+v"(qhi[1][1]));
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            dvacc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack_tr(dlo[s2][b], dhi[s2][b]), pf[s2], dvacc[b], 0, 0, 0);
            dkacc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack_tr(qlo[s2][b], qhi[s2][b]), dsf[s2], dkacc[b], 0, 0, 0);
          }
      };
      const std::integral_constant<int, 0> B0{};
      const std::integral_constant<int, 1> B1{};
      const std::integral_constant<int, 0> NONE{};
      const std::integral_constant<int, 2 * KS> BLOCK{};
      bf16x8 pf0[2], dsf0[2], pf1[2], dsf1[2];
      bf16x8 q0[KS], d0[KS], q1[KS], d1[KS];
#if B1_PIPE == 2
      frags(B0, q0, d0);
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
#elif B1_PIPE == 1
      frags(B0, q0, d0);
      scores(B0, NONE, q0, d0);
      __builtin_amdgcn_sched_barrier(0);
      frags(B1, q1, d1);
      scores(B1, NONE, q1, d1);
      weights(B0, pf0, dsf0);
      __builtin_amdgcn_sched_barrier(0);
      products(B0, pf0, dsf0);
      weights(B1, pf1, dsf1);
      __builtin_amdgcn_sched_barrier(0);
      products(B1, pf1, dsf1);
      __builtin_amdgcn_sched_barrier(0);
#else
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
#endif
    };

    LAB1(8)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // K half, the first chunk's tiles, the fragments: requested by request_stage
    LAB1(9)                                                  // or by the previous stage's last chunk
    // naming the fragments here makes the compiler place its own wait for their loads HERE: otherwise it cannot rule out that
    // they are still pending at the loop head and puts a vmcnt(0) in front of the first MFMA of every tile, which drains the
    // tile requested a moment before
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(kf[ks]), "+v"(vf[ks]));
    asm volatile("" : "+v"(bkey));
    stat_step(0);
    LAB1(10)
    if (hk == 0 && wave == 0 && lane == 0) jbox[(nit + 1) & 1] = jpend;   // the item after this one (drawn while the previous one ran)
    B1_BARRIER("lgkmcnt(0)");                                // K half, tile (c0, 0) and its statistics: in LDS; the next item's index
    // the stage after this one (nxt.valid == 0: none): the other key half of the same item, or the first live half of the next item
    nxt = cur;
    if (hk > 0) nxt.hk = hk - 1;
    else {
      const unsigned jn = (unsigned)__builtin_amdgcn_readfirstlane((int)jbox[(nit + 1) & 1]);
      nit += 1;
      if (wave == 0 && lane == 0) jpend = draw();
      nxt = open_next(jn);
    }
    const int c0n = a.causal ? 2 * nxt.hk : 0;               // first chunk of the next stage
    LAB1(0)
#pragma unroll 1
    for (int c = c0; c <= c1l; ++c) {
      const bool have_b = c * 128 + 64 < S, more = c < c1l, tail = !more && nxt.valid;  // uniform
      // the slot index goes through an opaque scalar: with a literal 0 / 1 every LDS address of the tile body is loop-invariant,
      // hipcc hoists ~50 of them out of the chunk loop and spills them (and K / V fragments) across phase 1
      int slot_a = 0, slot_b = 1;
      asm volatile("" : "+s"(slot_a), "+s"(slot_b));
      // ---- phase 1, first tile (slot 0) ----
      if (wave_live && (!a.causal || c * 128 + 63 >= key0)) tile(c * 128, slot_a);
      LAB1(1)
      if (have_b) {                                          // the second tile was requested a phase ago: its statistics
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LAB1(12)
        stat_step(1);
        LAB1(13)
      }
      B1_BARRIER("vmcnt(0) lgkmcnt(0)");                     // slot 0 free; slot 1 and its statistics visible
      LAB1(2)
      if (more) issue_tile(cur, c + 1, 0);
      else if (tail) issue_tile(nxt, c0n, 0);              // the next stage's first tile: slot 0 is free from here on
      // ---- phase 1, second tile (slot 1) ----
      if (have_b && wave_live && (!a.causal || c * 128 + 127 >= key0)) tile(c * 128 + 64, slot_b);
      LAB1(1)
      B1_BARRIER("lgkmcnt(0)");                              // panel complete; slot 1 free
      LAB1(11)
      // ---- phase 2 ----
      const int qt = wave >> 1, bh = wave & 1;
      const int q0b = c * 128 + 32 * qt;
      const bool p2 = q0b < S;                               // uniform
      const bool take_part = p2 && hk == 0 && part_valid && c >= 2;
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      f32x4 pv[4];                                           // the upper half's block of this chunk (added at the end: no wait here)
      f32x4 rav[4], rbv[4];                                  // rotary rows of this lane's query, for the epilogue
      int qpos = q0b + (lane & 31);
      qpos = min(qpos, S - 1);
      if (take_part) {
        const float* pp = part_w + (c - 2) * 8192 + lane * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) pv[i] = *reinterpret_cast<const f32x4*>(pp + i * 256);
      }
      const bool rot_q = p2 && hk == 0 && a.rot.qa != nullptr;
      {
        // loaded on every path (from the start of the q|k|v buffer when there is nothing to rotate) and named after the wait
        // below on every path: a load under one branch and its use under another would leave the compiler unable to tell that
        // the wait has happened, and it would put a vmcnt(0) in front of the epilogue, behind the request for the next tile
        const int Rr = rot_q ? a.rot.R : 4;
        const float* Ar = rot_q ? a.rot.qa + (int64_t)qpos * Rr : reinterpret_cast<const float*>(a.qkv);
        const float* Br = rot_q ? a.rot.qb + (int64_t)qpos * Rr : reinterpret_cast<const float*>(a.qkv);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int dh = 32 * bh + 8 * g4 + 4 * (lane >> 5);
          const int dhc = dh < Rr ? dh : 0;                  // columns past the rotary width: loaded from column 0, not used
          rav[g4] = *reinterpret_cast<const f32x4*>(Ar + dhc);
          rbv[g4] = *reinterpret_cast<const f32x4*>(Br + dhc);
        }
      }
      if (p2) {
        const unsigned kaddr = lds_addr(smem + B1_K), paddr = lds_addr(smem + B1_PANEL) + (qt >> 1) * 4096;
        const int b = qt & 1;
        unsigned livem = ~deadm & 0xffu;
        if (a.causal) {
          const int kbmax = (q0b + 31 - kh0) >> 5;           // last 32-key block any query of this block sees
          if (kbmax < 7) livem &= (2u << kbmax) - 1u;
        }
        // per-lane offsets of the transposed reads for this wave's column blocks (selects, not indexed: an indexed TrOff lives in scratch)
        unsigned tr0 = troff0;
        asm volatile("" : "+v"(tr0));
        const unsigned ka0 = kaddr + (tr0 ^ (bh ? 64u : 0u)), ka1 = ka0 ^ (1024u | 32u);
        const unsigned pa0 = paddr + (tr0 ^ (b ? 64u : 0u)), pa1 = pa0 ^ (1024u | 32u);
        // One asm statement per live key block: its eight transposed reads into FIXED registers, the wait, the two products.
        // Reads and wait must not be separate statements: a value the compiler can see between them it may copy (to form the
        // MFMA's register tuple, or at a loop edge), and a copy made before the read has landed carries the old content --
        // measured: one 32 x 32 block of dQ in ~3 M wrong, now and then, when this loop kept two blocks' reads in flight.
        // Two key blocks per statement where there are two left (sixteen reads in flight, the second block's landing under the
        // first one's products), one for an odd last one.
        unsigned m = livem;
        while (m) {
          const int kb = __builtin_ctz(m);
          m &= m - 1;
          // rows 32 (kb & 1).. of K tile kb >> 1 = byte kb * 4096 of the K image; panel region kb
          const unsigned ka = ka0 + kb * 4096, kc = ka1 + kb * 4096, pa = pa0 + kb * B1_REGION, pc = pa1 + kb * B1_REGION;
          if (B1_PAIR && m) {
            const int kb2 = __builtin_ctz(m);
            m &= m - 1;
            const unsigned ka2 = ka0 + kb2 * 4096, kc2 = ka1 + kb2 * 4096, pa2 = pa0 + kb2 * B1_REGION, pc2 = pa1 + kb2 * B1_REGION;
            asm volatile(
                "ds_read_b64_tr_b16 v[240:241], %1\n\t"
                "ds_read_b64_tr_b16 v[242:243], %2\n\t"
                "ds_read_b64_tr_b16 v[244:245], %3\n\t"
                "ds_read_b64_tr_b16 v[246:247], %4\n\t"
                "ds_read_b64_tr_b16 v[248:249], %1 offset:2048\n\t"
                "ds_read_b64_tr_b16 v[250:251], %2 offset:2048\n\t"
                "ds_read_b64_tr_b16 v[252:253], %3 offset:2048\n\t"
                "ds_read_b64_tr_b16 v[254:255], %4 offset:2048\n\t"
                "ds_read_b64_tr_b16 v[224:225], %5\n\t"
                "ds_read_b64_tr_b16 v[226:227], %6\n\t"
                "ds_read_b64_tr_b16 v[228:229], %7\n\t"
                "ds_read_b64_tr_b16 v[230:231], %8\n\t"
                "s_waitcnt lgkmcnt(8)\n\t"          // (never more than 12 in flight: the counter has 4 bits)
                "v_mfma_f32_32x32x16_bf16 %0, v[240:243], v[244:247], %0\n\t"
                "ds_read_b64_tr_b16 v[232:233], %5 offset:2048\n\t"
                "ds_read_b64_tr_b16 v[234:235], %6 offset:2048\n\t"
                "ds_read_b64_tr_b16 v[236:237], %7 offset:2048\n\t"
                "ds_read_b64_tr_b16 v[238:239], %8 offset:2048\n\t"
                "s_waitcnt lgkmcnt(8)\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, v[248:251], v[252:255], %0\n\t"
                "s_waitcnt lgkmcnt(4)\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, v[224:227], v[228:231], %0\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, v[232:235], v[236:239], %0\n\t"
                "s_nop 15"
                : "+v"(acc)
                : "v"(ka), "v"(kc), "v"(pa), "v"(pc), "v"(ka2), "v"(kc2), "v"(pa2), "v"(pc2)
                : "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239",
                  "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
          } else {
            asm volatile(
                "ds_read_b64_tr_b16 v[240:241], %1\n\t"
                "ds_read_b64_tr_b16 v[242:243], %2\n\t"
                "ds_read_b64_tr_b16 v[244:245], %3\n\t"
                "ds_read_b64_tr_b16 v[246:247], %4\n\t"
                "ds_read_b64_tr_b16 v[248:249], %1 offset:2048\n\t"
                "ds_read_b64_tr_b16 v[250:251], %2 offset:2048\n\t"
                "ds_read_b64_tr_b16 v[252:253], %3 offset:2048\n\t"
                "ds_read_b64_tr_b16 v[254:255], %4 offset:2048\n\t"
                "s_waitcnt lgkmcnt(4)\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, v[240:243], v[244:247], %0\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, v[248:251], v[252:255], %0\n\t"
                "s_nop 15"
                : "+v"(acc)
                : "v"(ka), "v"(kc), "v"(pa), "v"(pc)
                : "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
          }
        }
        // (each statement ends in the wait states an MFMA result needs before a vector instruction may read it: the compiler does
        // not know that `acc` comes out of the matrix pipe, and a register copy it places right behind a statement -- it did, once the
        // loop had two statements to merge -- would read rows the pipe has not written yet)
      }
      LAB1(3)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // tile (c + 1, 0), requested a phase ago; the partial block; the rotary rows
      LAB1(14)
      // (named here so that the compiler's own waits for these loads land here, not behind the request for tile (c + 1, 1) below)
      if (take_part) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(pv[i]));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[4 * i + e] += pv[i][e];
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) asm volatile("" : "+v"(rav[g4]), "+v"(rbv[g4]));
      if (p2 && hk != 0) {                                   // upper key half: leave the block for the lower half's pass over this chunk
        float* pp = part_w + (c - 2) * 8192 + lane * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4 v = {acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]};
          *reinterpret_cast<f32x4*>(pp + i * 256) = v;
        }
      }
      if (more) {
        stat_step(0);
        if ((c + 1) * 128 + 64 < S) issue_tile(cur, c + 1, 1);
      } else if (tail) {                                     // phase 1 of this stage is over: slot 1 and the K / V fragment registers
        if (c0n * 128 + 64 < S) issue_tile(nxt, c0n, 1);     // are free for the next stage
      }
      LAB1(15)
      B1_BARRIER("lgkmcnt(0)");                              // panel free; tile (c + 1, 0) and its statistics visible
      LAB1(4)
      if (tail) issue_K(nxt);                                // nobody reads this stage's K image any more
      if (p2 && hk == 0) {
        if (rot_q) {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int dh = 32 * bh + 8 * g4 + 4 * (lane >> 5);
            if (dh < a.rot.R) {
#pragma unroll
              for (int e = 0; e < 4; e += 2) {
                const float d0 = acc[g4 * 4 + e], d1 = acc[g4 * 4 + e + 1];
                acc[g4 * 4 + e] = rav[g4][e] * d0 + rbv[g4][e + 1] * d1;
                acc[g4 * 4 + e + 1] = rav[g4][e + 1] * d1 - rbv[g4][e] * d0;
              }
            }
          }
        }
        int ln = lane;
        asm volatile("" : "+v"(ln));
        store_transposed1(acc, a.scale, myreg, dqkv + 32 * bh, ld, q0b, S, ln);
      }
      LAB1(5)
#ifdef ATTN_LAB_STAMP
      lab_acc[7] += 1;
#endif
    }
    // ---- dV, dK of this wave's 32 keys ----
    if (wave_live) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int krow_e = min(key0 + (ln & 31), S - 1);     // recomputed: nothing per-lane of the epilogue lives across the chunk loop
      store_transposed(dvacc, 1.0f, myreg, dqkv + 2 * D, ld, key0, S, ln, 2);
      if (a.rot.ka) rotary_adjoint_regs(dkacc, a.rot.ka, a.rot.kb, a.rot.R, krow_e, ln);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      store_transposed(dkacc, a.scale, myreg, dqkv + D, ld, key0, S, ln, 2);
    } else if (key0 < S) {                                   // dead padding: exact zeros
      // (the lane index through an empty asm: otherwise hipcc forms the eight per-lane row offsets of these stores once at kernel
      // entry, spills them (the kernel sits at 256 registers) and reloads each one from scratch in front of its store -- a
      // scratch reload is a vector-memory load, and the vmcnt(0) behind it also waits for the PREVIOUS store's acknowledgement:
      // the zero rows of a dead key half went out one store round trip at a time.  Round 4, found in the ISA)
      int ln = lane;
      unsigned zz = 0;                                       // (likewise the zeros: as a constant they are parked in scratch too)
      asm volatile("" : "+v"(ln), "+v"(zz));
      const u32x4 z = {zz, zz, zz, zz};
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int idx = it * 64 + ln;
        const int row = idx >> 4, cc = idx & 15;
        if (key0 + row < S) *reinterpret_cast<u32x4*>(dqkv + (int64_t)(key0 + row) * ld + (1 + (cc >> 3)) * D + (cc & 7) * 8) = z;
      }
    }
    if (hk == 0 && upper_dead(sk)) {                         // the upper key half of this item was dead padding: exact zeros
      int ln = lane;
      unsigned zz = 0;
      asm volatile("" : "+v"(ln), "+v"(zz));
      const u32x4 z = {zz, zz, zz, zz};
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int idx = it * 64 + ln;
        const int row = 256 + wave * 32 + (idx >> 4), cc = idx & 15;
        if (row < S) *reinterpret_cast<u32x4*>(dqkv + (int64_t)row * ld + (1 + (cc >> 3)) * D + (cc & 7) * 8) = z;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    LAB1(6)
  };

#if B1_PRIO
  // the second-dispatched half of the workgroup loses every arbitration on its SIMD otherwise (MI355X_MICROARCH.md, two waves per
  // SIMD, item 4): vision 3 % faster, text even
  if (wave >= 4) __builtin_amdgcn_s_setprio(B1_PRIO);
#endif
  if (wave == 0 && lane == 0) {
    jbox[0] = draw();
    jpend = draw();
  }
  B1_BARRIER("vmcnt(0) lgkmcnt(0)");
  Stg cur = open_next((unsigned)__builtin_amdgcn_readfirstlane((int)jbox[0]));
  if (!cur.valid) return;
  request_stage(cur);
#pragma unroll 1
  while (cur.valid) {
    Stg nxt = cur;
    const int key0 = cur.hk * 256 + wave * 32;
    const bool wave_plain = uni(key0 + 32 <= S && !((cur.sp >> (key0 >> 6)) & 1) && !((cur.sk >> (key0 >> 6)) & 1));
    if (wave_plain) half(cur, nxt, std::true_type{});
    else half(cur, nxt, std::false_type{});
    cur = nxt;
  }
#ifdef ATTN_LAB_STAMP
  if (tid == 0 && blockIdx.x < LAB1_MAX_WG)
    for (int i = 0; i < 16; ++i) g_lab1_stamp[blockIdx.x * 16 + i] = lab_acc[i];
#endif
}

}  // namespace

#ifdef ATTN_LAB_STAMP
extern "C" int meant_lab_stamps1(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_lab1_stamp), bytes < sizeof(g_lab1_stamp) ? bytes : sizeof(g_lab1_stamp));
}
#endif

// workspace: [item counters: 8 x 16 words | partial dQ blocks of the upper key half (256 < S <= 512 only)]
size_t attn_bwd1_ws(int64_t G, int64_t S, int H, int Dh) {
  if (Dh != DH || S > 512) return 0;
  return 512 + (S > 256 ? align256_((size_t)G * H * 2 * 8 * 1024 * sizeof(float)) : 0);
}

bool attn_bwd1_ok(int64_t S, int Dh, int causal) {
  return meant_opt(MEANT_OPT_ATTN_BWD1) != 0 && Dh == DH && S > 0 && (S <= 256 || (causal && S <= 512));
}

// masks: the packed tile masks of attn_pack_flags_kernel (always present here: S <= 512 is at most 8 tiles); ws1: attn_bwd1_ws bytes
int attn_bwd1_launch(const bf16* qkv, const bf16* o, const bf16* dout, const float* lse, const float* bias2, const int* flags,
                     const uint64_t* masks, bf16* dqkv, void* ws1, int64_t G, int64_t S, int H, float scale, int causal, RotTables rot,
                     hipStream_t stream) {
  MEANT_REQUIRE(masks != nullptr, MEANT_ERR_ARG, "attn_bwd1: packed tile masks missing");
  MEANT_REQUIRE(ws1 != nullptr, MEANT_ERR_WORKSPACE, "attn_bwd1: workspace missing");
  unsigned* ctr = (unsigned*)ws1;                            // zeroed by attn_prep_mask_kernel, which attn_bf16_bwd launches first
  float* part = S > 256 ? (float*)((char*)ws1 + 512) : nullptr;
  Bwd1Args a{{qkv, o, dout, lse, bias2, flags, dqkv, nullptr, masks, 0, (int)S, H, scale, causal, (int)G, (int)ceil_div(S, 128), rot}, part, ctr};
  MEANT_RAISE_LDS(attn_bwd1_kernel, B1_LDS);
  meant_route_hit(ROUTE_ATTN_BWD1);
  // persistent: one workgroup per CU (it takes 146 KiB of LDS) walking its XCD's share of the items; never more workgroups than items
  int ncu = meant_num_cus() & ~7;
  if (ncu < 8) ncu = 8;
  const unsigned full = attn_grid(1, H, G);
  hipLaunchKernelGGL(attn_bwd1_kernel, dim3(full < (unsigned)ncu ? full : (unsigned)ncu), dim3(512), B1_LDS, stream, a);
  MEANT_LAUNCH_CHECK("attn_bwd1");
  return MEANT_OK;
}
