// LZ77 level-1 ("lazy2") encoder -- the codec behind zpaqfranz -m1.
// Reference: LZBuffer::LZBuffer/fill/write_literal/write_match/putb/flush, ZSFX/libzpaq.cpp:6140-6552
// (code format :6211-6222).  The output is bit-identical to LZBuffer's.  What makes that possible:
//  * Every input position is inserted into the hash table whatever the parse chose
//    (ZSFX/libzpaq.cpp:6432-6447); slot and value are pure functions of the input and the value
//    carries the position in its high bits, so "the table as LZBuffer sees it at position X" is the
//    per-slot maximum over all positions < X -- buildable with atomicMax, in any order.  The rolling
//    hash h1 has a finite window ((5<<shift1)^minMatch == 0 mod table size): recomputed from bytes.
//  * A wave walks its range in windows of 64 positions: all lanes look their position up at once
//    (vector table loads, candidate compares batched 16+16 bytes), inserts of earlier lanes of the
//    same window are forwarded through an LDS collision mask, and each lane precomputes the
//    reference's decision (:6396-6421) for both values of the (lit>0) score term.  Only the greedy
//    chain "take the match and skip blen, or emit literals" is serial: a wave-uniform loop over
//    v_readlane'd results that jumps over literal runs with one ballot.  Positions whose candidates
//    hit the 32-byte compare cap are re-evaluated exactly with 512-bytes-per-step whole-wave compares.
//  * What touches the table, what evaluates candidates and what chains tokens are three WAVES of one workgroup
//    (lz77_waves.inc: producer | evaluator | chain, rings in LDS): only the chain depends on the parse, so one table
//    serves three instruction streams.  lz_walk below is the same walk in one wave (seams and re-walks: short).
//  * Blocks are cut into segments of 2 MiB.  The table state at every segment start is built up
//    front (copy + atomicMax scatter), every segment is parsed speculatively from its own start
//    (lz77_spec3_kernel, one workgroup per segment), and one wave per block walks the true chain
//    (lz77_stitch_kernel): it re-parses from where the previous segment really ended until one of
//    its matches ends exactly where a speculative match ends -- both chains are then in the same
//    state (lit == 0), so the rest of that segment's speculative tokens are adopted verbatim.
//    With hundreds of blocks there is no speculation: one workgroup per block parses and writes the code
//    stream itself, the emission in a wave of its own (lz77_direct4_kernel).
//  * Tokens -> bits: a workgroup scan gives every token its bit offset; code bits and literal
//    bytes are OR-ed into the zeroed output in parallel (LSB-first, :6171-6186).
// Integer/byte work on random table slots: latency bound, no MFMA.  Algorithmic traffic per block of
// n bytes: n read + r*n written (r = LZ ratio); the hash tables are implementation traffic.
// tests/cpp/walk_emu.cpp compiles this file up to the token-move kernel for the HOST (ZPQ_EMU_WALK_ONLY: 64 lanes as
// fibres in lockstep, the wave intrinsics emulated) and checks the walk and the segment speculation against the oracle on the CPU.
#ifndef ZPQ_EMU_WALK_ONLY
#include <algorithm>
#include <mutex>
#include <stdlib.h>

#include "zpq_internal.h"
#define ZPQ_WAIT_VMCNT0 asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

namespace {

constexpr u32 kMaxMatch = (1u << 14) * 3;   // ZSFX/libzpaq.cpp:6258 (BUFSIZE*3)
constexpr u32 kMaxLiteral = (1u << 14) / 4;  // :6259
constexpr u32 kCap = 32;                     // speculative compare cap (bytes)
constexpr u32 kNoCand = 0xffffffffu;
constexpr u32 kSegMin = 2u << 20;            // smallest speculation segment: 15 table states per 16 MiB block instead of 31, and the three-wave parse of
                                             // 2 MiB (~140 ms) stays below the block's checksum chain (215 ms); measured 98 against 109 ms per headline step
constexpr u32 kMaxSeg = 64;                  // segments per block (64 MiB blocks at most)

struct LzCfg {
  const u8* in;
  u32 n;
  u32 minMatch, bucket, htbits, checkbits, shift1, rb;
  u32 upd_limit;   // positions < upd_limit are inserted (i + minMatchBoth < n)
  u32 level;       // 1: bit codes, 2: byte-aligned codes -- a match must then be 1 / 2 bytes longer to pay for a 3- / 4-byte offset
};
// the length a match at distance off must reach to be taken (ZSFX/libzpaq.cpp:6415-6416)
__device__ __forceinline__ u32 lz_need(const LzCfg& C, u32 off) {
  return C.minMatch + (C.level == 2 ? (u32)(off >= (1u << 16)) + (u32)(off >= (1u << 24)) : 0u);
}

// per (block, segment)
struct LzSegDev {
  LzCfg c;
  u32 x0, x1;        // segment range [x0, x1)
  u32* work;         // table as of x0, mutated by the speculative parse
  u32* pristine;     // table as of x0, mutated by the stitcher's re-parse
  u32* tpos; u32* tlen; u32* toff;  // speculative tokens
  u32 tcap;
  u32* state;        // [0]=ntok [1]=end cur [2]=end lit [3]=overflow
  u32* qpos; u32* qlen; u32* qoff;  // tokens of the seam walk INTO this segment (lz77_seam_kernel)
  u32* seam;         // [0]=ntok [1]=sync index into the speculative list (or ~0) [2]=end cur [3]=end lit
                     // [4]=assumed entry cur [5]=assumed entry lit [6]=overflow [7]=valid
};

typedef zpq_lzjob_dev LzJobDev;   // per block (also what the pack kernels read): zpq_internal.h

__device__ __forceinline__ int lg32(u32 x) { return x ? 32 - __builtin_clz(x) : 0; }  // lg(), :6224-6233
// Input, hash tables and token lists are addressed as GLOBAL memory, the collision masks as LDS: with generic
// pointers every access becomes a flat_* instruction, and a wait for one kind then waits for the other too --
// this parse is bound by exactly those waits.
typedef __attribute__((address_space(1))) const u8 g_cu8;
typedef __attribute__((address_space(1))) u32 g_u32;
typedef __attribute__((address_space(1))) const u32 g_cu32;
typedef __attribute__((address_space(1))) const u64 g_cu64;
typedef __attribute__((address_space(1))) const u32x4 g_cu32x4;
typedef __attribute__((address_space(1))) const u64_u g_cu64_u;
typedef __attribute__((address_space(1))) const u32x4_u g_cu32x4_u;
typedef __attribute__((address_space(3))) unsigned long long l_u64;
__device__ __forceinline__ u64 load8(g_cu8* p) { return *(g_cu64_u*)p; }
__device__ __forceinline__ u32x4 load16(g_cu8* p) { return *(g_cu32x4_u*)p; }

// h1 as LZBuffer holds it when it reaches position q (:6444): the rolling hash over the last
// minMatch update steps, i.e. over in[U..U+minMatch-1] with U = min(q, upd_limit); partial for U < minMatch.
__device__ __forceinline__ u32 hash_at(const LzCfg& C, u32 q) {
  const u32 U = q < C.upd_limit ? q : C.upd_limit;
  const u32 mm = C.minMatch;
  const u32 F = 5u << C.shift1;
  u32 h = 0;
  const u32 t0 = U > mm ? U - mm : 0;
  g_cu8* in = (g_cu8*)C.in;
  for (u32 t = t0; t < U; ++t) h = h * F + (in[t + mm] + 1u) * 123456791u;
  return h & ((1u << C.htbits) - 1u);
}

// Same value from 8 bytes already in registers (qb = in[q..q+7]); valid for minMatch <= q <= upd_limit, minMatch <= 8.
__device__ __forceinline__ u32 hash_fast(const LzCfg& C, u64 qb) {
  const u32 F = 5u << C.shift1;
  u32 h = 0;
  for (u32 j = 0; j < C.minMatch; ++j) h = h * F + ((u32)((qb >> (8 * j)) & 255u) + 1u) * 123456791u;
  return h & ((1u << C.htbits) - 1u);
}

// first differing byte index of two 16-byte vectors (16 if equal)
__device__ __forceinline__ u32 mismatch16(u32x4 a, u32x4 b) {
  const u64 x0 = ((u64)(a.y ^ b.y) << 32) | (u64)(a.x ^ b.x);
  const u64 x1 = ((u64)(a.w ^ b.w) << 32) | (u64)(a.z ^ b.z);
  if (x0) return (u32)(__builtin_ctzll(x0) >> 3);
  if (x1) return 8u + (u32)(__builtin_ctzll(x1) >> 3);
  return 16u;
}

// byte idx (0..31) of the 32 bytes held in two 16-byte vectors
__device__ __forceinline__ u32 byte32(const u32x4& lo, const u32x4& hi, u32 idx) {
  const u32 w = idx >> 2;
  const u32 a = w & 1 ? (w & 2 ? lo.w : lo.y) : (w & 2 ? lo.z : lo.x);
  const u32 b = w & 1 ? (w & 2 ? hi.w : hi.y) : (w & 2 ? hi.z : hi.x);
  return ((w & 4 ? b : a) >> ((idx & 3) * 8)) & 255u;
}

// event counters of the walk (tools/emu/lz_walk_counts.py builds the emulated engine with -DZPQ_LZ_COUNT): [0] windows looked
// up, [1] windows that only insert, [2] tokens, [3] tokens re-evaluated exactly (a candidate reached the 32-byte cap),
// [4] extension rounds of those, [5] whole-wave compare rounds, [6] pipeline restarts, [7] windows of the pipelined walk
#ifdef ZPQ_LZ_COUNT
__device__ unsigned long long g_lzcount[8];
#define LZ_C(i) ((void)(lane_id() == 0 ? g_lzcount[i] += 1 : 0ull))
extern "C" void zpq_debug_lzcount(unsigned long long out[8], int reset) { for (int i = 0; i < 8; ++i) { out[i] = g_lzcount[i]; if (reset) g_lzcount[i] = 0; } }
#else
#define LZ_C(i) ((void)0)
#endif
// Whole-wave compare for long matches: 512 bytes per step.  All arguments wave-uniform.
__device__ __forceinline__ u32 coop_match_len(g_cu8* in, u32 p, u32 q, u32 limit, u32 from = 0) {
  const u32 lane = (u32)lane_id();
  u32 base = from;
  while (base < limit) {
    LZ_C(5);
    u32 o = base + lane * 8;
    u64 x = o < limit ? (load8(in + p + o) ^ load8(in + q + o)) : 1ull;
    unsigned long long m = __ballot(x != 0);
    if (m) {
      int fl = __builtin_ctzll(m);
      u64 xf = (u64)__shfl((unsigned long long)x, fl);
      u32 of = base + (u32)fl * 8;
      u32 l = of < limit ? of + (u32)(__builtin_ctzll(xf) >> 3) : limit;
      return l < limit ? l : limit;
    }
    base += 512;
  }
  return limit;
}

template <int NB> struct GroupLoad;
template <> struct GroupLoad<1> { static __device__ __forceinline__ void ld(g_cu32* p, u32 (&e)[1]) { e[0] = __builtin_nontemporal_load(p); } };
template <> struct GroupLoad<2> { static __device__ __forceinline__ void ld(g_cu32* p, u32 (&e)[2]) {
  u64 v = __builtin_nontemporal_load((g_cu64*)p); e[0] = (u32)v; e[1] = (u32)(v >> 32); } };
template <> struct GroupLoad<4> { static __device__ __forceinline__ void ld(g_cu32* p, u32 (&e)[4]) {
  u32x4 v = __builtin_nontemporal_load((g_cu32x4*)p); e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w; } };
template <> struct GroupLoad<8> { static __device__ __forceinline__ void ld(g_cu32* p, u32 (&e)[8]) {
  u32x4 v = __builtin_nontemporal_load((g_cu32x4*)p), w = __builtin_nontemporal_load((g_cu32x4*)p + 1);
  e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w; e[4] = w.x; e[5] = w.y; e[6] = w.z; e[7] = w.w; } };

struct TokSink { u32* pos; u32* len; u32* off; u32 cap; u32 n; };

#ifdef ZPQ_LZ_PROFILE
__device__ unsigned long long g_lzprof[8];
#define LZ_T(i) do { const u64 t_ = __builtin_readcyclecounter(); prof[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define LZ_T(i) do {} while (0)
#endif

// Direct emitter: with one workgroup per block (no segment speculation) the parse order IS the output order, so the chain
// wave writes the code stream as it goes -- LZBuffer::write_literal / write_match / putb (ZSFX/libzpaq.cpp:6166-6184,
// 6455-6520) -- and no token list exists.  Control is wave-uniform; a literal run's bytes are spread over the lanes,
// each shifting in the bits its left neighbour spills.
struct BitSink {
  __attribute__((address_space(1))) u8* out; u32 out_cap;
  __attribute__((address_space(1))) const u8* in;
  u64 acc; u32 accbits;        // pending bits, fewer than 8 between calls
  u32 bytepos;                 // bytes written
  u32 gap_start;               // first position not yet emitted
  u32 rb, overflow;
  // the block's bytes [lit_lo, lit_hi) as the emitting wave keeps them in LDS (a ring of lit_mask + 1 bytes, position &
  // lit_mask, filled 256 bytes at a time with the next 256 already requested): a literal run is a few bytes long and starts
  // where the last match ended, so its bytes come from there instead of two dependent global loads per run
  // (lit == nullptr: no such cache)
  __attribute__((address_space(3))) u8* lit; u32 lit_lo, lit_hi, lit_mask;
  u32 in_n;                    // bytes of `in` (the ring is filled up to the end of the block, four bytes per lane)
  u32 pf, pf_at;               // the next 256 bytes' word of this lane, and where they start (~0: none requested)

  __device__ __forceinline__ u32 chunk_word(u32 at, u32 lane) const {
    return at + 4u * lane < in_n ? *(__attribute__((address_space(1))) const u32_u*)(in + at + 4u * lane) : 0u;   // (buffers are padded)
  }
  __device__ __forceinline__ void ensure(u32 from, u32 upto, u32 lane) {   // [from, upto) into the ring; upto - from <= 512
    if (from < lit_lo || from > lit_hi + 512u) { lit_lo = lit_hi = from & ~255u; pf_at = 0xffffffffu; }
    while (lit_hi < upto) {
      const u32 w = pf_at == lit_hi ? pf : chunk_word(lit_hi, lane);
      ((__attribute__((address_space(3))) u32*)lit)[((lit_hi & lit_mask) >> 2) + lane] = w;
      lit_hi += 256u;
      if (lit_hi - lit_lo > lit_mask + 1u) lit_lo = lit_hi - (lit_mask + 1u);
      pf_at = lit_hi; pf = chunk_word(lit_hi, lane);
    }
    __builtin_amdgcn_wave_barrier();
  }

  __device__ __forceinline__ void put(u64 v, u32 k, u32 lane) {           // k <= 56 bits, LSB first
    acc |= v << accbits;
    accbits += k;
    const u32 nb = accbits >> 3;
    if (lane < nb) {
      if (bytepos + lane < out_cap) out[bytepos + lane] = (u8)(acc >> (8 * lane)); else overflow = 1;
    }
    bytepos += nb;
    acc = nb >= 8 ? 0ull : acc >> (8 * nb);
    accbits &= 7u;
  }
  __device__ __forceinline__ void literal_run(u32 from, u32 len, u32 lane) {   // 00, gamma(len), bytes (:6455-6478)
    u64 v = 0; u32 k = 2;
    for (int b = lg32(len) - 2; b >= 0; --b) { v |= 1ull << k; ++k; v |= (u64)((len >> b) & 1u) << k; ++k; }
    ++k;
    put(v, k, lane);
    const u32 sh = accbits;                  // 0..7 bits already in the byte the run starts in
    if (lit && len <= 512u) {
      ensure(from, from + len, lane);
      for (u32 j = lane; j < len; j += 64) {
        const u32 c = lit[(from + j) & lit_mask];
        const u32 left = lit[(from + j - 1u) & lit_mask];                // (unused for j == 0)
        const u32 carry = j ? (sh ? left >> (8 - sh) : 0u) : (u32)acc;
        if (bytepos + j < out_cap) out[bytepos + j] = (u8)((c << sh) | carry); else overflow = 1;
      }
      bytepos += len;
      acc = sh ? (u64)((u32)lit[(from + len - 1u) & lit_mask] >> (8 - sh)) : 0ull;
      return;
    }
    for (u32 j = lane; j < len; j += 64) {
      const u32 c = in[from + j];
      const u32 left = j ? (u32)in[from + j - 1] : 0u;
      const u32 carry = j ? (sh ? left >> (8 - sh) : 0u) : (u32)acc;
      if (bytepos + j < out_cap) out[bytepos + j] = (u8)((c << sh) | carry); else overflow = 1;
    }
    bytepos += len;
    acc = sh ? (u64)((u32)in[from + len - 1] >> (8 - sh)) : 0ull;
  }
  __device__ __forceinline__ void literals(u32 upto, u32 lane) {          // everything in [gap_start, upto) as literal runs
    u32 g = upto - gap_start, from = gap_start;
    while (g) {
      const u32 r = g < kMaxLiteral ? g : kMaxLiteral;
      literal_run(from, r, lane);
      from += r; g -= r;
    }
    gap_start = upto;
  }
  __device__ __forceinline__ void match(u32 pos, u32 len, u32 off, u32 lane) {   // :6481-6520, level 1
    literals(pos, lane);
    const u32 o = off + (1u << rb) - 1u;
    const u32 lo = (u32)lg32(o) - 1u - rb;
    u64 v = ((lo + 8u) >> 3) | ((u64)(lo & 7u) << 2);
    u32 k = 5;
    for (int b = lg32(len) - 2; b >= 2; --b) { v |= 1ull << k; ++k; v |= (u64)((len >> b) & 1u) << k; ++k; }
    ++k;                                    // terminating 0
    v |= (u64)(len & 3u) << k; k += 2;      // k <= 34
    put(v, k, lane);
    const u64 tail = (u64)(o & ((1u << rb) - 1u)) | ((u64)((o >> rb) & ((1u << lo) - 1u)) << rb);
    if (rb + lo) put(tail, rb + lo, lane);
    gap_start = pos + len;
  }
  __device__ __forceinline__ u32 finish(u32 n, u32 lane) {                // trailing literals, last partial byte (flush, :6181-6184)
    literals(n, lane);
    if (accbits) {
      if (lane == 0) { if (bytepos < out_cap) out[bytepos] = (u8)acc; else overflow = 1; }
      ++bytepos; acc = 0; accbits = 0;
    }
    return bytepos;
  }
};

// Speculative tokens of one segment, consulted by the stitcher after each of its own tokens.
struct SpecList { const u32* pos; const u32* len; u32 n; u32 j; };

// Walks positions [wbase, x1) of a block in windows of 64: inserts every position into `ht`
// (which must hold exactly the inserts of all positions < wbase) and continues the greedy parse
// from (cur, lit).  Tokens go to `sink`.  With a SpecList the walk stops as soon as one of its
// matches ends where a speculative match ends and returns that token's index (else -1).
// This is the walk in ONE wave: what the seam and stitch kernels call (a few windows each).  Whole segments and blocks are
// parsed by three waves that split the same work between them (lz77_waves.inc).
template <int NB>
__device__ int lz_walk(const LzCfg& C, u32* __restrict__ ht_generic, u32 wbase, u32 x1, u32& cur, u32& lit, TokSink& sink,
                       SpecList* spec, unsigned long long* T_generic) {
  const u32 lane = (u32)lane_id();
  g_cu8* in = (g_cu8*)C.in;
  g_u32* ht = (g_u32*)ht_generic;
  g_u32* const sink_pos = (g_u32*)sink.pos; g_u32* const sink_len = (g_u32*)sink.len; g_u32* const sink_off = (g_u32*)sink.off;
  l_u64* T = (l_u64*)T_generic;
  const u32 n = C.n;
  const u32 mask = (1u << C.checkbits) - 1u;
  const u32 mm = C.minMatch;
  volatile l_u64* Tv = T;
  const bool fast_hash = mm <= 8;
  const u32 hfrozen = hash_at(C, C.upd_limit);   // h1 once updates have stopped (q > upd_limit)
#ifdef ZPQ_LZ_PROFILE
  u64 prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}; u64 tlast = __builtin_readcyclecounter();
#endif

  for (u32 base = wbase; base < x1 && base < n; base += 64) {
    const u32 q = base + lane;
    const bool inb = q < n && q < x1;
    u32 wend = base + 64 < x1 ? base + 64 : x1;
    if (wend > n) wend = n;
    // ---- per-position hash, slot, value --------------------------------------------------------
    const u64 qb = load8(in + (q < n ? q : 0));           // in[q..q+7] (buffers are padded)
    u32 h;
    if (fast_hash && q >= mm && q <= C.upd_limit) h = hash_fast(C, qb);
    else if (q > C.upd_limit) h = hfrozen;
    else h = inb ? hash_at(C, q) : 0u;
    const bool ins = inb && q < C.upd_limit;
    const u32 ih = ((q * 1234547u) >> 19) & C.bucket;                     // :6435
    const u32 slot = h ^ ih;
    const u32 b3 = (inb && q + 3 < n) ? (u32)((qb >> 24) & 255u) : 0u;
    const u32 val = (q << C.checkbits) | (b3 & mask);                      // :6436
    const u32 grp = h & ~C.bucket;
    const bool look = inb && cur < wend;   // windows swallowed by a match only insert
    LZ_C(cur < wend ? 0 : 1);

    LZ_T(0);
    u32 ent[NB];
    if (look) GroupLoad<NB>::ld((g_cu32*)(ht + (size_t)grp), ent);   // bypasses L1: the table is rewritten by this wave
    else {
#pragma unroll
      for (int j = 0; j < NB; ++j) ent[j] = 0;
    }
    // ---- forward inserts of earlier lanes in this window; find superseded stores ---------------
    bool superseded = false;
    {
    const u32 tk = (grp >> 3) & 255u;
    if (inb) __hip_atomic_fetch_or(&T[tk], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __builtin_amdgcn_wave_barrier();
    unsigned long long cm = inb ? Tv[tk] : 0ull;
    cm &= ~(1ull << lane);
    while (__ballot(cm != 0)) {
      const int k = cm ? __builtin_ctzll(cm) : 0;  // ascending: later lanes overwrite earlier ones
      const u32 sk = __shfl(slot, k), vk = __shfl(val, k);
      const bool ik = __shfl((int)ins, k) != 0;
      if (cm) {
        if (ik && (u32)k < lane && (sk & ~C.bucket) == grp) {
#pragma unroll
          for (int j = 0; j < NB; ++j)
            if ((sk & C.bucket) == (u32)j) ent[j] = vk;
        }
        if (ik && (u32)k > lane && sk == slot) superseded = true;
        cm &= cm - 1;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (inb) Tv[tk] = 0ull;
  }
    LZ_T(1);
    // ---- reorder the group into probe order ht[h1^k], k = 0..bucket (:6397) ---------------------
    {
    const u32 hb = h & C.bucket;
#pragma unroll
    for (int bit = 1; bit < NB; bit <<= 1) {
      const bool sw = (hb & (u32)bit) != 0;
#pragma unroll
      for (int j = 0; j < NB; ++j)
        if (!(j & bit)) {
          u32 a = ent[j], b = ent[j | bit];
          ent[j] = sw ? b : a;
          ent[j | bit] = sw ? a : b;
        }
    }
  }
    // ---- speculative candidate evaluation (lanes at or after the chain head) -------------------
    // All first 16-byte loads are issued before any is used: one memory round trip for the group.
    u32 cp[NB], cl[NB];
    bool slow = false;
    const u32 limit = inb ? (n - q < kMaxMatch ? n - q : kMaxMatch) : 0u;
    const bool evalp = look && q >= cur;
    u32x4 ca[NB], cb[NB];
    const u32x4 qa = load16(in + (q < n ? q : 0)), qc = load16(in + (q < n ? q : 0) + 16);
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      cp[k] = kNoCand; cl[k] = 0;
      const u32 e = ent[k];
      if (evalp && e && q + 3 < n && (e & mask) == (b3 & mask)) {           // :6398
        const u32 p = e >> C.checkbits;
        if (p < q) cp[k] = p;
      }
      g_cu8* src = in + (cp[k] != kNoCand ? cp[k] : 0u);
      ca[k] = load16(src); cb[k] = load16(src + 16);
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      if (cp[k] != kNoCand) {
        u32 l = mismatch16(ca[k], qa);
        if (l == 16) l = 16 + mismatch16(cb[k], qc);
        if (l >= limit) l = limit;                       // never beyond the input / maxMatch
        else if (l == kCap) slow = true;                 // may extend further: resolve exactly when reached
        cl[k] = l;
      }
    }
    if (__ballot(cl[0] == 12345u)) LZ_T(7);   // forces the candidate results before the timestamp
    LZ_T(2);
    // ---- the reference's decision for both values of (lit>0) (:6396-6421) -----------------------
    u32 rlen[2] = {0, 0}, roff[2] = {0, 0};
    if (evalp && !slow) {
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        u32 blen = mm - 1, bp = 0; int bscore = 0;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          if (blen < 128 && cp[k] != kNoCand) {
            const u32 p = cp[k], l = cl[k];
            bool ok = false;
            if (q + blen <= n) {
              const u32 idx = blen - 1;
              if (idx < l) ok = true;
              else if (idx == l) ok = false;  // first mismatch (l < limit here because q+blen<=n)
              else ok = byte32(ca[k], cb[k], idx) == byte32(qa, qc, idx);   // idx < 32 here (no capped candidate)
            }
            if (ok) {
              const int score = (int)(l * 8) - lg32(q - p) - 2 * f - 11;
              if (score > bscore) { blen = l; bp = p; bscore = score; }
            }
          }
        }
        const u32 off = q - bp;
        if (off > 0 && bscore > 0 && blen >= lz_need(C, off)) { rlen[f] = blen; roff[f] = off; }
      }
    }
    // ---- serial greedy chain over this window (wave-uniform) -----------------------------------
    const unsigned long long slowmask = __ballot(slow);
    LZ_T(3);
    const unsigned long long stop1 = __ballot(rlen[1] != 0) | slowmask;  // where a lit>0 run must stop
    while (cur < wend) {
      const u32 j = cur - base;
      if (lit > 0) {                                   // inside a literal run: jump to its end
        const unsigned long long m = stop1 >> j;
        u32 skip = m ? (u32)__builtin_ctzll(m) : 64u;
        if (skip > wend - cur) skip = wend - cur;
        if (skip) {
          if (lit + skip >= kMaxLiteral) { skip = kMaxLiteral - lit; lit = 0; }   // forced flush (:6450-6451)
          else lit += skip;
          cur += skip;
          continue;
        }
      }
      const u32 f = lit > 0 ? 1u : 0u;
      u32 tlen, toff;
      if ((slowmask >> j) & 1ull) {
        // exact re-evaluation of position cur (:6396-6408).  Lane j already holds every candidate's
        // position and its length up to the 32-byte cap; capped candidates are extended for ALL
        // candidates at once (64/NB lanes each, 8 bytes per lane and step), anything still
        // unresolved after 4 steps by whole-wave compares.
        const u32 i = cur;
        LZ_C(3);
        const u32 lim_i = n - i < kMaxMatch ? n - i : kMaxMatch;
        u32 xp[NB], xl[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) { xp[k] = __builtin_amdgcn_readlane(cp[k], j); xl[k] = __builtin_amdgcn_readlane(cl[k], j); }
        {
          constexpr u32 W = 64 / NB;                 // lanes per candidate
          const u32 g = lane / W, sub = lane % W;
          u32 myp = kNoCand; bool open = false;
#pragma unroll
          for (int k = 0; k < NB; ++k) if (g == (u32)k) { myp = xp[k]; open = xp[k] != kNoCand && xl[k] == kCap && kCap < lim_i; }
          u32 found = 0xffffffffu;                   // exact length once known (per candidate group)
          for (u32 step = 0; step < 4 && __ballot(open); ++step) {
            LZ_C(4);
            const u32 o = kCap + step * 8 * W + 8 * sub;
            u32 mine = 0xffffffffu;
            if (open) {
              if (o >= lim_i) mine = lim_i;
              else {
                const u64 x = load8(in + myp + o) ^ load8(in + i + o);
                if (x) { const u32 l = o + (u32)(__builtin_ctzll(x) >> 3); mine = l < lim_i ? l : lim_i; }
              }
            }
#pragma unroll
            for (u32 d = 1; d < W; d <<= 1) { const u32 y = __shfl_xor(mine, (int)d); mine = mine < y ? mine : y; }
            if (open && mine != 0xffffffffu) { found = mine; open = false; }
          }
#pragma unroll
          for (int k = 0; k < NB; ++k) {
            const u32 fk = __builtin_amdgcn_readlane(found, k * W);
            const bool still = __builtin_amdgcn_readlane((u32)open, k * W) != 0;
            if (fk != 0xffffffffu) xl[k] = fk;
            else if (still) xl[k] = coop_match_len(in, xp[k], i, lim_i, kCap + 4 * 8 * W);
          }
        }
        u32 blen = mm - 1, bp = 0; int bscore = 0;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          if (blen < 128 && xp[k] != kNoCand) {
            const u32 p = xp[k], l = xl[k];
            bool ok = false;
            if (i + blen <= n) {
              const u32 idx = blen - 1;
              if (idx < l) ok = true;
              else if (idx == l) ok = false;
              else ok = in[p + idx] == in[i + idx];
            }
            if (ok) {
              const int score = (int)(l * 8) - lg32(i - p) - 2 * (int)f - 11;
              if (score > bscore) { blen = l; bp = p; bscore = score; }
            }
          }
        }
        const u32 off = i - bp;
        const bool take = off > 0 && bscore > 0 && blen >= lz_need(C, off);
        tlen = take ? blen : 0u; toff = off;
      } else {
        const u32 l0 = __builtin_amdgcn_readlane(rlen[0], j), l1 = __builtin_amdgcn_readlane(rlen[1], j);
        const u32 o0 = __builtin_amdgcn_readlane(roff[0], j), o1 = __builtin_amdgcn_readlane(roff[1], j);
        tlen = f ? l1 : l0; toff = f ? o1 : o0;
      }
      if (tlen) {
        LZ_C(2);
        if (lane == 0 && sink.n < sink.cap) { sink_pos[sink.n] = cur; sink_len[sink.n] = tlen; sink_off[sink.n] = toff; }
        ++sink.n;
        lit = 0;
        cur += tlen;
        if (spec) {
          // does this match end where a speculative match of the segment ends?
          const u32 E = cur;
          int found = -1;
          for (;;) {
            const u32 jj = spec->j + lane;
            const u32 e = jj < spec->n ? ((g_cu32*)spec->pos)[jj] + ((g_cu32*)spec->len)[jj] : 0xffffffffu;
            const unsigned long long lt = __ballot(e < E), eq = __ballot(e == E);
            if (eq) { found = (int)(spec->j + (u32)__builtin_ctzll(eq)); break; }
            const u32 adv = (u32)__builtin_popcountll(lt);
            spec->j += adv;
            if (adv < 64) break;
          }
          if (found >= 0) {
            // positions of this window not yet inserted: the caller's table is not used again
            return found;
          }
        }
      } else {
        ++lit; ++cur;
        if (lit >= kMaxLiteral) lit = 0;  // forced literal flush (:6450-6451); runs are positional
      }
    }
    LZ_T(4);
    // ---- insert this window's positions (latest writer of a slot wins) --------------------------
    {
    if (ins && !superseded) ht[slot] = val;
    ZPQ_WAIT_VMCNT0;
  }
    LZ_T(5);
  }
#ifdef ZPQ_LZ_PROFILE
  if (lane == 0 && !spec) { for (int i = 0; i < 8; ++i) atomicAdd(&g_lzprof[i], (unsigned long long)prof[i]); }
#endif
  return -1;
}


// ---- table state at every segment start --------------------------------------------------------------
struct CopyJob { const u32* src; u32* dst; u32 words; };   // src == nullptr: zero fill

__global__ __launch_bounds__(256) void lz77_table_copy_kernel(const CopyJob* __restrict__ jobs) {
  const CopyJob J = jobs[blockIdx.y];
  const u32 nvec = J.words >> 2;
  u32x4* d = (u32x4*)J.dst;
  const u32x4* s = (const u32x4*)J.src;
  const u32x4 z = {0, 0, 0, 0};
  for (u32 i = blockIdx.x * 256u + threadIdx.x; i < nvec; i += gridDim.x * 256u) d[i] = s ? s[i] : z;
}

struct ScatterJob { LzCfg c; u32 x0, x1; u32* dst; };

// dst[slot] = max(dst[slot], value) for every inserted position of [x0, x1): the value carries the
// position in its high bits, so the maximum is the latest insert -- what LZBuffer's table holds.
__global__ __launch_bounds__(256) void lz77_table_scatter_kernel(const ScatterJob* __restrict__ jobs) {
  const ScatterJob J = jobs[blockIdx.y];
  const LzCfg& C = J.c;
  const u32 mask = (1u << C.checkbits) - 1u;
  const u32 hi = J.x1 < C.upd_limit ? J.x1 : C.upd_limit;
  for (u32 q = J.x0 + blockIdx.x * 256u + threadIdx.x; q < hi; q += gridDim.x * 256u) {
    const u64 qb = load8((g_cu8*)C.in + q);
    const u32 h = (C.minMatch <= 8 && q >= C.minMatch) ? hash_fast(C, qb) : hash_at(C, q);
    const u32 ih = ((q * 1234547u) >> 19) & C.bucket;
    const u32 val = (q << C.checkbits) | ((u32)((qb >> 24) & 255u) & mask);
    if (val) atomicMax(J.dst + (h ^ ih), val);
  }
}

// ---- seams: one wave per segment boundary -----------------------------------------------------------------
// Seam k continues the parse from where segment k-1's SPECULATION ended (true whenever segment k-1 got back
// in step, which the stitch kernel verifies) into segment k, on the pristine table of x0, until one of its
// matches ends where a speculative match of segment k ends.
template <int NB>
__global__ __launch_bounds__(64) void lz77_seam_kernel(const LzSegDev* __restrict__ segs, const u32* __restrict__ list) {
  __builtin_amdgcn_s_setprio(3);
  const u32 si = list[blockIdx.x];
  const LzSegDev S = segs[si];
  const u32 lane = (u32)lane_id();
  if (S.x0 == 0) { if (lane == 0) S.seam[7] = 0; return; }          // first segment of a block: no seam
  __shared__ unsigned long long T[256];
  T[lane] = 0; T[lane + 64] = 0; T[lane + 128] = 0; T[lane + 192] = 0;
  __builtin_amdgcn_wave_barrier();
  const LzSegDev Pv = segs[si - 1];
  u32 cur = Pv.state[1], lit = Pv.state[2];
  const u32 ecur = cur, elit = lit;
  TokSink sink{S.qpos, S.qlen, S.qoff, S.tcap, 0};
  int hit = -1;
  if (cur == S.x0 && lit == 0) hit = -2;                           // in step from the first position
  else if (cur < S.x1) {
    SpecList sl{S.tpos, S.tlen, S.state[0], 0};
    hit = lz_walk<NB>(S.c, S.pristine, S.x0, S.x1, cur, lit, sink, &sl, T);
  }
  if (lane == 0) {
    S.seam[0] = sink.n < sink.cap ? sink.n : sink.cap;
    S.seam[1] = hit == -2 ? 0u : hit >= 0 ? (u32)hit + 1u : 0xffffffffu;    // adopt speculative tokens from here
    S.seam[2] = cur; S.seam[3] = lit; S.seam[4] = ecur; S.seam[5] = elit;
    S.seam[6] = sink.n > sink.cap; S.seam[7] = 1;
  }
}

// ---- true chain: one wave per block ----------------------------------------------------------------------
// Glues [speculative tokens of segment 0] [seam 1] [speculative tokens of segment 1 from its sync index] ...
// A seam is used only if it started from the state the true chain really is in; otherwise the segment is
// re-walked here on work[k-1] (after the speculative pass that table holds exactly the inserts < x0).
template <int NB>
__global__ __launch_bounds__(64) void lz77_stitch_kernel(const LzJobDev* __restrict__ jobs, const LzSegDev* __restrict__ segs,
                                                         const u32* __restrict__ list) {
  __builtin_amdgcn_s_setprio(3);
  const LzJobDev J = jobs[list[blockIdx.x]];
  __shared__ unsigned long long T[256];
  const u32 lane = (u32)lane_id();
  T[lane] = 0; T[lane + 64] = 0; T[lane + 128] = 0; T[lane + 192] = 0;
  __builtin_amdgcn_wave_barrier();
  TokSink out{J.tok_pos, J.tok_len, J.tok_off, J.tok_cap, 0};
  u32 cur = 0, lit = 0, overflow = 0;
  // adopted token ranges are only RECORDED here; lz77_move_tokens_kernel copies them with the whole chip
  u32 kcur = 0, slot = 0;
  auto append = [&](u32 which, u32 from, u32 to) {
    if (lane == 0) {
      u32* q = J.plan + (kcur * 2 + slot) * 4;
      q[0] = which; q[1] = from; q[2] = to > from ? to : from; q[3] = out.n;
    }
    ++slot;
    if (to > from) out.n += to - from;
  };
  for (u32 t = lane; t < J.nseg * 8; t += 64) J.plan[t] = 0;
  __builtin_amdgcn_wave_barrier();
  for (u32 k = 0; k < J.nseg; ++k) {
    const LzSegDev S = segs[J.seg0 + k];
    const u32 ns = S.state[0];
    overflow |= S.state[3];
    if (cur >= S.x1) continue;                          // a match swallowed the whole segment
    kcur = k; slot = 0;
    if (cur == S.x0 && lit == 0) {                       // the speculation started in the true state
      append(0, 0, ns);
      cur = S.state[1]; lit = S.state[2];
      continue;
    }
    if (k > 0 && S.seam[7] && S.seam[4] == cur && S.seam[5] == lit) {     // the seam continued exactly this chain
      overflow |= S.seam[6];
      append(1, 0, S.seam[0]);
      if (S.seam[1] != 0xffffffffu) { append(0, S.seam[1], ns); cur = S.state[1]; lit = S.state[2]; }
      else { cur = S.seam[2]; lit = S.seam[3]; }
      continue;
    }
    // exact re-walk (rare: the previous segment never got back in step)
    SpecList sl{S.tpos, S.tlen, ns, 0};
    u32* table = k ? segs[J.seg0 + k - 1].work : S.work;
    const int hit = lz_walk<NB>(S.c, table, S.x0, S.x1, cur, lit, out, &sl, T);
    T[lane] = 0; T[lane + 64] = 0; T[lane + 128] = 0; T[lane + 192] = 0;
    __builtin_amdgcn_wave_barrier();
    if (hit >= 0) { append(0, (u32)hit + 1, ns); cur = S.state[1]; lit = S.state[2]; }
  }
  if (lane == 0) {
    J.result[0] = out.n < out.cap ? out.n : out.cap;
    if (out.n > out.cap || overflow) J.result[2] = 1;
  }
}

// Moves the recorded token ranges into the block's final token list: one workgroup per (segment, range, chunk).
__global__ __launch_bounds__(256) void lz77_move_tokens_kernel(const LzJobDev* __restrict__ jobs, const LzSegDev* __restrict__ segs,
                                                              const u32* __restrict__ seg_job) {
  const u32 si = blockIdx.y >> 1, r = blockIdx.y & 1;
  const LzJobDev J = jobs[seg_job[si]];
  const LzSegDev S = segs[si];
  const u32* q = J.plan + ((si - J.seg0) * 2 + r) * 4;
  const u32 which = q[0], from = q[1], to = q[2], dst = q[3];
  const u32* ps = which ? S.qpos : S.tpos; const u32* ln = which ? S.qlen : S.tlen; const u32* of = which ? S.qoff : S.toff;
  for (u32 t = from + blockIdx.x * 256u + threadIdx.x; t < to; t += gridDim.x * 256u) {
    const u32 o = dst + (t - from);
    if (o < J.tok_cap) { J.tok_pos[o] = ps[t]; J.tok_len[o] = ln[t]; J.tok_off[o] = of[t]; }
  }
}

#include "lz77_waves.inc"        // three / four waves on one table: lz77_spec3_kernel, lz77_direct4_kernel

#if defined(ZPQ_EMU_WALK_ONLY) && !defined(ZPQ_EMU_FULL)
}  // namespace (host emulation, tests/cpp/walk_emu.cpp: the parse kernels up to here; nothing behind them is compiled.
   //            tests/cpp/lz77_full_emu.cpp -- ZPQ_EMU_FULL -- takes the whole file, host code included, over a stand-in HIP runtime)
#else
// ---- bit costs ---------------------------------------------------------------------------------
__device__ __forceinline__ u32 lit_run_header_bits(u32 len) { return 3u + 2u * (u32)(lg32(len) - 1); }  // :6464-6476
__device__ __forceinline__ u64 lit_gap_bits(u32 g) {
  const u32 full = g / kMaxLiteral, r = g % kMaxLiteral;
  u64 bits = (u64)full * (lit_run_header_bits(kMaxLiteral) + 8ull * kMaxLiteral);
  if (r) bits += lit_run_header_bits(r) + 8ull * r;
  return bits;
}
__device__ __forceinline__ u32 match_bits(u32 len, u32 off, u32 rb) {  // :6494-6516
  const u32 o = off + (1u << rb) - 1u;
  const u32 lo = (u32)lg32(o) - 1u - rb;
  return 5u + 2u * (u32)(lg32(len) - 3) + 1u + 2u + rb + lo;
}

__device__ __forceinline__ void or_bits(u32* out, u64 bitpos, u64 value, u32 nbits) {
  // value occupies the low nbits (<= 57); LSB-first packing (putb, :6171-6179)
  if (!nbits) return;
  const u64 w = bitpos >> 5; const u32 sh = (u32)(bitpos & 31);
  const u64 lo = value << sh;
  if ((u32)lo) atomicOr(out + w, (u32)lo);
  if ((u32)(lo >> 32)) atomicOr(out + w + 1, (u32)(lo >> 32));
  if (sh && nbits + sh > 64) { const u32 hi = (u32)(value >> (64 - sh)); if (hi) atomicOr(out + w + 2, hi); }
}

__device__ __forceinline__ void put_lit_header(u32* out, u64 bitpos, u32 len) {
  // 00, then the bits of len below its leading one each preceded by a 1, then 0 (:6469-6476)
  u64 v = 0; u32 k = 2;
  for (int b = lg32(len) - 2; b >= 0; --b) { v |= 1ull << k; ++k; v |= (u64)((len >> b) & 1u) << k; ++k; }
  ++k;
  or_bits(out, bitpos, v, k);
}

__device__ __forceinline__ void put_match(u32* out, u64 bitpos, u32 len, u32 off, u32 rb) {
  const u32 o = off + (1u << rb) - 1u;
  const u32 lo = (u32)lg32(o) - 1u - rb;
  u64 v = ((lo + 8u) >> 3) | ((u64)(lo & 7u) << 2);
  u32 k = 5;
  for (int b = lg32(len) - 2; b >= 2; --b) { v |= 1ull << k; ++k; v |= (u64)((len >> b) & 1u) << k; ++k; }
  ++k;                                    // terminating 0
  v |= (u64)(len & 3u) << k; k += 2;
  or_bits(out, bitpos, v, k);             // k <= 5 + 26 + 1 + 2 = 34
  const u64 tail = (u64)(o & ((1u << rb) - 1u)) | ((u64)((o >> rb) & ((1u << lo) - 1u)) << rb);
  or_bits(out, bitpos + k, tail, rb + lo);
}

// One workgroup per block: scans token costs, records each token's start bit and writes run
// headers, match codes and the literal bytes: a gap of up to kShortGap bytes by the thread that owns the token behind it
// (seven bytes per OR), a longer one by the whole workgroup once the tile's tokens are placed.
constexpr u32 kShortGap = 56;
__global__ __launch_bounds__(1024) void lz77_pack_tokens_kernel(const LzJobDev* __restrict__ jobs) {
  const LzJobDev J = jobs[blockIdx.x];
  const u32 ntok = J.result[0];
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ u64 wsum[16];
  __shared__ u64 carry_s;
  __shared__ u32 lg_from[1024], lg_len[1024], lg_n;
  __shared__ u64 lg_bit[1024];
  if (tid == 0) { carry_s = 0; lg_n = 0; }
  __syncthreads();
  u32* out32 = (u32*)J.out;
  // items 0..ntok-1 = (gap before match t, match t); item ntok = trailing literal gap
  for (u32 t0 = 0; t0 <= ntok; t0 += 1024) {
    const u32 t = t0 + tid;
    u64 cost = 0; u32 gap = 0, gstart = 0, pos = 0, len = 0, off = 0;
    if (t <= ntok) {
      gstart = t ? J.tok_pos[t - 1] + J.tok_len[t - 1] : 0u;
      if (t < ntok) { pos = J.tok_pos[t]; len = J.tok_len[t]; off = J.tok_off[t]; } else pos = J.n;
      gap = pos - gstart;
      cost = lit_gap_bits(gap) + (t < ntok ? match_bits(len, off, J.rb) : 0u);
    }
    // inclusive scan within the workgroup
    u64 x = cost;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { u64 y = __shfl_up((unsigned long long)x, d); if (lane >= (u32)d) x += y; }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    u64 wbase = 0;
    for (u32 w = 0; w < wave; ++w) wbase += wsum[w];
    const u64 carry = carry_s;
    const u64 start = carry + wbase + x - cost;
    if (t <= ntok) {
      if (t < ntok) J.tok_bit[t] = (u32)start; else J.tok_bit[ntok] = (u32)start;
      // literal run headers of the gap
      u64 bp = start; u32 g = gap;
      while (g) {
        const u32 r = g < kMaxLiteral ? g : kMaxLiteral;
        put_lit_header(out32, bp, r);
        bp += lit_run_header_bits(r) + 8ull * r; g -= r;
      }
      if (t < ntok) put_match(out32, bp, len, off, J.rb);
      if (gap && gap <= kShortGap) {           // (one run: kShortGap < kMaxLiteral)
        const u64 lb = start + lit_run_header_bits(gap);
        for (u32 j = 0; j < gap; j += 7) {
          const u32 c = gap - j < 7 ? gap - j : 7u;
          or_bits(out32, lb + 8ull * j, load8((g_cu8*)J.in + gstart + j) & ((1ull << (8 * c)) - 1ull), 8 * c);
        }
      } else if (gap) {
        const u32 k = atomicAdd(&lg_n, 1u);
        lg_from[k] = gstart; lg_len[k] = gap; lg_bit[k] = start;
      }
    }
    __syncthreads();
    if (tid == 1023) carry_s = carry + wbase + x;
    // the long gaps of this tile, one after the other, a byte per thread and round (the run structure as the headers above)
    const u32 nlong = lg_n;
    for (u32 e = 0; e < nlong; ++e) {
      const u32 from = lg_from[e], g = lg_len[e]; const u64 gb = lg_bit[e];
      for (u32 r = tid; r < g; r += 1024) {
        const u32 run = r / kMaxLiteral, within = r % kMaxLiteral;
        const u32 runlen = (run < g / kMaxLiteral) ? kMaxLiteral : g % kMaxLiteral;
        const u64 bpos = gb + (u64)run * (lit_run_header_bits(kMaxLiteral) + 8ull * kMaxLiteral) + lit_run_header_bits(runlen) + 8ull * within;
        or_bits(out32, bpos, J.in[from + r], 8);
      }
    }
    __syncthreads();
    if (tid == 0) lg_n = 0;
    __syncthreads();
  }
  if (tid == 0) {
    const u64 bits = carry_s;
    const u64 bytes = (bits + 7) >> 3;
    J.result[1] = (u32)bytes;
    if (bytes > J.out_cap) J.result[2] = 1;
  }
}


// ---- the hash-table finder with a SECOND, higher-order context and lookahead (args[3] = minMatch2 > 0, args[6]) -----------------
// LZBuffer::fill's hash branch in full (ZSFX/libzpaq.cpp:6373-6447): the bucket of the order-(minMatch2 + lookahead) hash h2 is
// searched first -- a match is counted from `lookahead` bytes behind the position, then extended backwards, the bytes in front of
// it become leading literals -- and the order-minMatch bucket only if that found nothing of minMatch2 bytes; both hashes index
// the SAME table and every position is inserted under both.  No built-in level asks for it ("x" methods only), so it gets the
// plain formulation: ONE WAVE PER BLOCK walks the positions the parse visits, lanes = the 2 x (bucket + 1) candidates of a
// position (table word, match length by 8-byte compares, backward extension), the reference's in-order selection on
// `v_readlane`d results, the code stream written as the parse goes (LZBuffer::write_literal / write_match / putb / put,
// :6166-6184, :6452-6547 -- bits or, level 2, bytes), the skipped positions inserted 64 at a time (hashes computed from the
// bytes, "latest position wins" by atomic maximum).  Two or three dependent memory round trips per visited position: a
// fallback's speed (about a megabyte per second and block), many blocks side by side.
struct GenericJob {
  const u8* in; u32 n;
  u32 minMatch, minMatch2, lookahead, bucket, htbits, checkbits, shift1, shift2, rb, level, minMatchBoth;
  u32* ht;             // 2^htbits words, zeroed
  u8* out; u32 out_cap;
  u32* result;         // [0] matches, [1] bytes written, [2] overflow
};

struct ByteSink {      // wave-uniform state; lane 0 stores
  __attribute__((address_space(1))) u8* out; u32 cap, pos, overflow;
  u64 acc; u32 nbits;
  __device__ __forceinline__ void byte(u32 b, u32 lane) {
    if (pos < cap) { if (lane == 0) out[pos] = (u8)b; } else overflow = 1;
    ++pos;
  }
  __device__ __forceinline__ void putb(u32 x, u32 k, u32 lane) {      // k <= 24 bits, LSB first (:6171-6179)
    acc |= (u64)(x & ((1u << k) - 1u)) << nbits;
    nbits += k;
    while (nbits >= 8) { byte((u32)(acc & 255u), lane); acc >>= 8; nbits -= 8; }
  }
  __device__ __forceinline__ void flush(u32 lane) { if (nbits) { byte((u32)(acc & 255u), lane); acc = 0; nbits = 0; } }
};

// h1 / h2 as LZBuffer holds them when it reaches position q (the rolling updates of :6438-6444 folded from the bytes; `la` = 0 and
// the order-minMatch constants for h1, `lookahead` and the order-minMatch2 constants for h2); updates stop at upd_limit
__device__ __forceinline__ u32 generic_hash(g_cu8* in, u32 q, u32 upd_limit, u32 order, u32 la, u32 mulc, u32 shift, u32 addc, u32 htmask) {
  const u32 U = q < upd_limit ? q : upd_limit;
  const u32 F = mulc << shift;
  u32 h = 0;
  for (u32 t = U > order ? U - order : 0; t < U; ++t) h = h * F + ((u32)in[t + order + la] + 1u) * addc;
  return h & htmask;
}

__device__ __forceinline__ u32 generic_match_from(g_cu8* in, u32 p, u32 i, u32 from, u32 limit) {   // first l >= from with in[p+l] != in[i+l], capped at limit
  u32 l = from;
  while (l + 8 <= limit) {
    const u64 x = load8(in + p + l) ^ load8(in + i + l);
    if (x) return l + (u32)(__builtin_ctzll(x) >> 3);
    l += 8;
  }
  while (l < limit && in[p + l] == in[i + l]) ++l;
  return l;
}

__global__ __launch_bounds__(64) void lz77_generic_kernel(const GenericJob* __restrict__ jobs) {
  const GenericJob J = jobs[blockIdx.x];
  const u32 lane = (u32)lane_id();
  g_cu8* in = (g_cu8*)J.in;
  g_u32* ht = (g_u32*)J.ht;
  const u32 n = J.n, mm = J.minMatch, mm2 = J.minMatch2, la = J.lookahead, NB = J.bucket + 1u;
  const u32 htmask = (1u << J.htbits) - 1u, mask = (1u << J.checkbits) - 1u;
  const u32 upd_limit = n > J.minMatchBoth ? n - J.minMatchBoth : 0u;          // positions i with i + minMatchBoth < n are inserted
  ByteSink bs{(__attribute__((address_space(1))) u8*)J.out, J.out_cap, 0u, 0u, 0ull, 0u};
  auto at = [&](u32 x) -> u32 { return x < n ? (u32)in[x] : 0u; };             // (bytes behind the block read as 0)
  auto write_literal = [&](u32 i, u32& lit) {                                  // :6452-6485
    if (J.level == 1) {
      if (lit < 1) return;
      int ll = lg32(lit);
      bs.putb(0, 2, lane);
      --ll;
      while (--ll >= 0) { bs.putb(1, 1, lane); bs.putb((lit >> ll) & 1u, 1, lane); }
      bs.putb(0, 1, lane);
      while (lit) { bs.putb((u32)in[i - lit], 8, lane); --lit; }
    } else {
      while (lit > 0) {
        const u32 lit1 = lit > 64 ? 64u : lit;
        bs.byte(lit1 - 1, lane);
        for (u32 j = i - lit; j < i - lit + lit1; ++j) bs.byte((u32)in[j], lane);
        lit -= lit1;
      }
    }
  };
  auto write_match = [&](u32 len, u32 off) {                                   // :6488-6547
    if (J.level == 1) {
      int ll = lg32(len) - 1;
      off += (1u << J.rb) - 1u;
      const int lo = lg32(off) - 1 - (int)J.rb;
      bs.putb((u32)(lo + 8) >> 3, 2, lane);
      bs.putb((u32)lo & 7u, 3, lane);
      while (--ll >= 2) { bs.putb(1, 1, lane); bs.putb((len >> ll) & 1u, 1, lane); }
      bs.putb(0, 1, lane);
      bs.putb(len & 3u, 2, lane);
      bs.putb(off, J.rb, lane);
      if (lo > 0) bs.putb(off >> J.rb, (u32)lo, lane);
    } else {
      --off;
      while (len > 0) {
        const u32 len1 = len > mm * 2 + 63 ? mm + 63 : len > mm + 63 ? len - mm : len;
        if (off < (1u << 16)) { bs.byte(64 + len1 - mm, lane); bs.byte(off >> 8, lane); bs.byte(off, lane); }
        else if (off < (1u << 24)) { bs.byte(128 + len1 - mm, lane); bs.byte(off >> 16, lane); bs.byte(off >> 8, lane); bs.byte(off, lane); }
        else { bs.byte(192 + len1 - mm, lane); bs.byte(off >> 24, lane); bs.byte(off >> 16, lane); bs.byte(off >> 8, lane); bs.byte(off, lane); }
        len -= len1;
      }
    }
  };
  u32 i = 0, lit = 0, nmatch = 0;
  while (i < n) {
    // ---- the candidates of position i: lanes [0, NB) the h2 bucket, [NB, 2 NB) the h1 bucket, in probe order
    const u32 h1 = generic_hash(in, i, upd_limit, mm, 0, 5u, J.shift1, 123456791u, htmask);
    const u32 h2 = mm2 ? generic_hash(in, i, upd_limit, mm2, la, 9u, J.shift2, 23456789u, htmask) : 0u;
    const bool grp2 = lane < NB, grp1 = lane >= NB && lane < 2 * NB;
    u32 p = 0; bool valid = false;
    u32 lo = 0, hi = 0;                     // in[p + x] == in[i + x] for x in [lo, hi); a mismatch (or a limit) at hi and at lo - 1
    if ((grp2 && mm2) || grp1) {
      const u32 e = __builtin_nontemporal_load(ht + ((grp2 ? h2 : h1) ^ (grp2 ? lane : lane - NB)));
      const bool chk = grp2 ? true : i + 3 < n;                                   // (:6376 / :6398: only the lower order tests i + 3 < n)
      if (e && chk && (e & mask) == (at(i + 3) & mask)) { p = e >> J.checkbits; valid = p < i; }
    }
    if (valid) {
      const u32 limit = n - i < kMaxMatch ? n - i : kMaxMatch;                   // i + l < n && l < maxMatch
      const u32 from = grp2 ? la : 0u;
      hi = from <= limit ? generic_match_from(in, p, i, from, limit) : from;
      lo = from;
      while (lo > 0 && in[p + lo - 1] == in[i + lo - 1]) --lo;                  // (h2 only: back from the lookahead)
    }
    // ---- the reference's selection, candidate after candidate (:6373-6409)
    u32 blen = mm - 1, bp = 0, blit = 0; int bscore = 0;
    for (u32 k = 0; k < 2 * NB; ++k) {
      if (k < NB && !mm2) continue;
      if (k == NB && mm2 && blen >= mm2) break;                                   // "if (!minMatch2 || blen<minMatch2)"
      const bool v = __shfl((int)valid, (int)k) != 0;
      if (v) {
        const u32 pk = (u32)__shfl((int)p, (int)k), lok = (u32)__shfl((int)lo, (int)k), hik = (u32)__shfl((int)hi, (int)k);
        bool ok = i + blen <= n;
        if (ok) {                                                                 // in[p+blen-1] == in[i+blen-1]
          const u32 idx = blen - 1;
          if (idx >= lok && idx < hik) ok = true;
          else ok = in[pk + idx] == in[i + idx];                                    // (both inside the block: i + blen <= n, p < i)
        }
        if (ok) {
          if (k < NB) {
            const u32 l = hik;                                                    // counted from the lookahead
            if (l >= mm2 + la) {
              const u32 l1 = lok;
              const int score = (int)(l - l1) * 8 - lg32(i - pk) - 8 * (int)(lit == 0 && l1 > 0) - 11;
              if (score > bscore) { blen = l; bp = pk; blit = l1; bscore = score; }
            }
          } else {
            const u32 l = hik;
            const int score = (int)l * 8 - lg32(i - pk) - 2 * (int)(lit > 0) - 11;
            if (score > bscore) { blen = l; bp = pk; blit = 0; bscore = score; }
          }
        }
      }
      if (blen >= 128) break;
    }
    // ---- take the match or count a literal (:6411-6426)
    const u32 off = i - bp;
    const u32 need = mm + (J.level == 2 ? (u32)(off >= (1u << 16)) + (u32)(off >= (1u << 24)) : 0u);
    if (off > 0 && bscore > 0 && blen - blit >= need) {
      lit += blit;
      write_literal(i + blit, lit);
      write_match(blen - blit, off);
      ++nmatch;
    } else {
      blen = 1;
      ++lit;
    }
    // ---- insert the blen positions from i on, under both hashes (:6430-6447); "latest position wins" = the maximum
    for (u32 q0 = i; q0 < i + blen; q0 += 64) {
      const u32 q = q0 + lane;
      if (q < i + blen && q < upd_limit) {
        const u32 ih = ((q * 1234547u) >> 19) & J.bucket;
        const u32 val = (q << J.checkbits) | (at(q + 3) & mask);
        if (val) {
          if (mm2) atomicMax((u32*)J.ht + (generic_hash(in, q, upd_limit, mm2, la, 9u, J.shift2, 23456789u, htmask) ^ ih), val);
          atomicMax((u32*)J.ht + (generic_hash(in, q, upd_limit, mm, 0, 5u, J.shift1, 123456791u, htmask) ^ ih), val);
        }
      }
    }
    ZPQ_WAIT_VMCNT0;
    i += blen;
    if (lit >= kMaxLiteral) write_literal(i, lit);
  }
  write_literal(n, lit);
  bs.flush(lane);
  if (lane == 0) { J.result[0] = nmatch; J.result[1] = bs.pos; J.result[2] = bs.overflow; }
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------
extern "C" size_t zpq_lz77_bound(size_t n) { return n + n / 64 + 64; }   // level 2 spends a byte per 64 literals, level 1 15 bits per 4096

#ifdef ZPQ_LZ_PROFILE
extern "C" int zpq_debug_lzprof2(unsigned long long out[24], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lzprof2), 192) != hipSuccess) return -1;
  if (reset) { unsigned long long z[24] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lzprof2), z, 192); }
  return 0;
}
extern "C" int zpq_debug_lzprof(unsigned long long out[8], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lzprof), 64) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lzprof), z, 64); }
  return 0;
}
#endif

int zpq_lz77_pack_launch(zpq_ctx* ctx, const zpq_lzjob_dev* d_jobs, size_t nj, u32 max_n) {
  hipStream_t st = ctx->stream;
  ZPQ_LAUNCH(ctx, "lz77_pack_tokens_kernel", st, lz77_pack_tokens_kernel, dim3((unsigned)nj), dim3(1024), d_jobs);
  ZPQ_HIP(ctx, hipGetLastError());
  (void)max_n;
  return ZPQ_OK;
}

static bool uses_suffix_array(const int32_t a[9]) { return a[5] - a[0] >= 21; }
static bool uses_second_context(const int32_t a[9]) { return !uses_suffix_array(a) && (a[3] != 0 || a[6] != 0); }

static int check_args(zpq_ctx* ctx, const int32_t a[9], u32 n) {
  const int lvl = a[1] & 3;
  if (a[1] < 1 || a[1] > 7 || a[1] == 4 || lvl == 3 || lvl == 0)
    return zpq_fail(ctx, ZPQ_ERR_METHOD, "LZ77 pre-processor %d not implemented (levels 1 and 2 are)", a[1]);
  if (uses_suffix_array(a)) {        // LZ77-SA (methods 2..4): lz77_sa.hip
    if (a[0] < 0 || a[0] > 6 || a[5] > 31) return zpq_fail(ctx, ZPQ_ERR_METHOD, "suffix-array LZ77: block size 2^%d out of range", 20 + a[0]);
    if (lvl == 1 ? (a[2] < 4 || a[2] > 255) : (a[2] < 1 || a[2] > 64)) return zpq_fail(ctx, ZPQ_ERR_METHOD, "min match %d out of range", a[2]);
    if (a[4] < 0 || a[4] > 12) return zpq_fail(ctx, ZPQ_ERR_METHOD, "2^%d neighbours not implemented", a[4]);
    if (a[6] < 0 || a[6] > 1) return zpq_fail(ctx, ZPQ_ERR_METHOD, "lookahead %d not implemented (0 and 1 are)", a[6]);
    if ((u64)n > (1ull << (20 + a[0]))) return zpq_fail(ctx, ZPQ_ERR_ARG, "block of %u bytes exceeds 2^%d", n, 20 + a[0]);
    return ZPQ_OK;
  }
  // a second, higher-order context and lookahead (lz77_generic_kernel): both up to 64 bytes
  if (a[3] < 0 || a[3] > 64 || a[6] < 0 || a[6] > 64) return zpq_fail(ctx, ZPQ_ERR_METHOD, "secondary context %d / lookahead %d out of range", a[3], a[6]);
  if (a[2] < 4 || a[2] > 31) return zpq_fail(ctx, ZPQ_ERR_METHOD, "min match %d out of range", a[2]);
  if (a[4] < 0 || a[4] > 3) return zpq_fail(ctx, ZPQ_ERR_METHOD, "bucket 2^%d not implemented", a[4]);
  if (a[0] < 0 || a[0] > 6 || a[5] - a[0] >= 21 || a[5] < 4 || a[5] > 26 || a[5] <= a[4])
    return zpq_fail(ctx, ZPQ_ERR_METHOD, "hash table 2^%d out of range", a[5]);
  if ((u64)n > (1ull << (20 + a[0]))) return zpq_fail(ctx, ZPQ_ERR_ARG, "block of %u bytes exceeds 2^%d", n, 20 + a[0]);
  return ZPQ_OK;
}

template <typename T>
static T* carve(u8*& p, size_t count) {
  T* r = (T*)p;
  p += (count * sizeof(T) + 255) & ~(size_t)255;
  return r;
}

// HBM a job needs while it is parsed with segments of kSegBytes: 2*segments-1 hash tables plus the token lists
static size_t job_bytes_direct(const zpq_lz77_job& z) { return ((size_t)4 << z.args[5]) + 4096; }
static size_t job_bytes(const zpq_lz77_job& z, u32 kSegBytes) {
  const u32 nseg = std::max<u32>(1, (u32)(((u64)z.n + kSegBytes - 1) / kSegBytes));
  // tokens are matches of at least args[2] bytes that do not overlap: n / minMatch of them at most.  Final list 4 words
  // per token, speculative list 3 per segment, seam list 3 for every segment but the first
  const size_t mm = (size_t)(z.args[2] >= 4 ? z.args[2] : 4);
  const size_t toks = (size_t)z.n / mm + 3 + 3 * nseg;
  return ((size_t)4 << z.args[5]) * (2 * (size_t)nseg - 1) + toks * (16 + 12 + (nseg > 1 ? 12 : 0));
}

// Three-wave kernels (lz77_waves.inc): producer | evaluator | chain, one workgroup per segment / block
template <int NB>
static void launch_spec3(zpq_ctx* ctx, hipStream_t st, dim3 grid, const LzSegDev* d_segs, const u32* sl) {
  ZPQ_LAUNCH(ctx, "lz77_spec_kernel", st, lz77_spec3_kernel<NB>, grid, dim3(192), d_segs, sl);
}
template <int NB>
static void launch_direct4(zpq_ctx* ctx, hipStream_t st, dim3 grid, const LzJobDev* d_jobs, const LzSegDev* d_segs, const u32* jl) {
#ifdef ZPQ_LZ_ORACLE
  {   // the experiment of lz77_waves.inc: the first launch records the visited positions, the later ones use them
    static u32* map = nullptr; static size_t map_jobs = 0; static int launches = 0;
    const size_t nj = grid.x;
    if (!map || nj > map_jobs) {
      if (map) (void)hipFree(map);
      (void)hipMalloc((void**)&map, nj * kOracleStride * 4); map_jobs = nj; launches = 0;
    }
    const u32 mode = launches == 0 ? 1u : 2u;
    if (mode == 1u) (void)hipMemsetAsync(map, 0, nj * kOracleStride * 4, st);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lz_oracle_map), &map, sizeof map);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lz_oracle_mode), &mode, sizeof mode);
    fprintf(stderr, "[lz oracle] launch %d of %zu blocks: mode %u\n", launches, nj, mode);
    ++launches;
  }
#endif
  ZPQ_LAUNCH(ctx, "lz77_direct_kernel", st, lz77_direct4_kernel<NB>, grid, dim3(256), d_jobs, d_segs, jl);
}

// Encodes jobs[lo..hi) in one batch (their tables fit the memory budget together).
static int encode_batch(zpq_ctx* ctx, zpq_lz77_job* jobs, size_t lo, size_t hi, const u32 kSegBytes, const bool direct) {
  hipStream_t st = ctx->stream;
  const size_t nj = hi - lo;
  std::vector<LzJobDev> hj(nj);
  std::vector<LzSegDev> hs;
  size_t table_words = 0, tok_words = 0;
  u32 max_n = 0, max_seg = 1;
  for (size_t i = 0; i < nj; ++i) {
    const zpq_lz77_job& z = jobs[lo + i];
    const u32 nseg = std::max<u32>(1, (z.n + kSegBytes - 1) / kSegBytes);
    const size_t words = (size_t)1 << z.args[5];
    table_words += words * (2 * (size_t)nseg - 1);
    const u32 mmt = (u32)(z.args[2] >= 4 ? z.args[2] : 4);
    if (!direct) tok_words += ((size_t)z.n / mmt + 3) * 4;  // final pos/len/off/bit
    for (u32 k = 0; k < nseg && !direct; ++k) {
      const u32 x0 = k * kSegBytes, x1 = std::min<u64>((u64)x0 + kSegBytes, z.n);
      tok_words += ((size_t)(x1 - x0) / mmt + 3) * (k ? 6 : 3);   // speculative (+ seam, except for the first segment) token lists
    }
    max_seg = std::max(max_seg, nseg);
  }
  size_t nseg_total = 0;
  for (size_t i = 0; i < nj; ++i) nseg_total += std::max<u32>(1, (jobs[lo + i].n + kSegBytes - 1) / kSegBytes);
  u32* d_tab = (u32*)zpq_scratch(ctx, 0, table_words * 4 + 256);
  u32* d_tok = (u32*)zpq_scratch(ctx, 1, tok_words * 4 + 256);
  const size_t meta_bytes = nj * (sizeof(LzJobDev) + 16 + 4 * 4) + 256 + nseg_total * (sizeof(LzSegDev) + 48 + 36 + 4 * 4 + sizeof(CopyJob) * 2 + sizeof(ScatterJob)) + 8192;
  u8* d_meta = (u8*)zpq_scratch(ctx, 2, meta_bytes);
  if (!d_tab || !d_tok || !d_meta) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "lz77 scratch (%zu MiB of tables)", table_words >> 18);
  u8* mp = d_meta;
  LzJobDev* d_jobs = carve<LzJobDev>(mp, nj);
  LzSegDev* d_segs = carve<LzSegDev>(mp, nseg_total);
  u32* d_res = carve<u32>(mp, nj * 4);
  u32* d_state = carve<u32>(mp, nseg_total * 12);            // state[4] + seam[8] per segment
  u32* d_plan = carve<u32>(mp, nseg_total * 8);
  u32* d_segjob = carve<u32>(mp, nseg_total);
  u32* d_lists = carve<u32>(mp, (nj + nseg_total) * 4);      // per bucket width: job list, segment list
  CopyJob* d_copy = carve<CopyJob>(mp, nseg_total * 2);
  ScatterJob* d_scat = carve<ScatterJob>(mp, nseg_total);
  ZPQ_HIP(ctx, hipMemsetAsync(d_res, 0, nj * 16, st));

  size_t to = 0, tabo = 0;
  std::vector<std::vector<CopyJob>> copy_step(max_seg);      // step k: build pristine[k]
  std::vector<std::vector<ScatterJob>> scat_step(max_seg);
  std::vector<CopyJob> copy_work;
  for (size_t i = 0; i < nj; ++i) {
    zpq_lz77_job& z = jobs[lo + i];
    const int32_t* a = z.args;
    LzCfg c;
    c.in = z.d_in; c.n = z.n; c.minMatch = a[2]; c.bucket = (1u << a[4]) - 1; c.htbits = a[5]; c.checkbits = 12 - a[0];
    c.shift1 = (a[5] - 1) / a[2] + 1; c.rb = a[0] > 4 ? a[0] - 4 : 0; c.level = (u32)(a[1] & 3);
    const u32 mmb = a[2] + 4;
    c.upd_limit = z.n > mmb ? z.n - mmb : 0;
    const u32 nseg = std::max<u32>(1, (z.n + kSegBytes - 1) / kSegBytes);
    const size_t words = (size_t)1 << a[5];
    LzJobDev& J = hj[i];
    J.in = z.d_in; J.n = z.n; J.rb = c.rb; J.nseg = nseg; J.seg0 = (u32)hs.size();
    const u32 mmt = (u32)(a[2] >= 4 ? a[2] : 4);
    const u32 cap = direct ? 1u : z.n / mmt + 3;
    J.tok_pos = d_tok + to; J.tok_len = J.tok_pos + cap; J.tok_off = J.tok_len + cap; J.tok_bit = J.tok_off + cap;
    J.tok_cap = cap - 1; to += direct ? 0 : (size_t)cap * 4;
    J.result = d_res + 4 * i; J.out = z.d_out; J.out_cap = z.out_cap;
    J.plan = d_plan + 8 * (size_t)hs.size();
    if (!direct) ZPQ_HIP(ctx, hipMemsetAsync(J.out, 0, J.out_cap, st));     // (the direct emitter writes whole bytes)
    // tables: work[0..nseg-1], pristine[1..nseg-1]
    u32* work0 = d_tab + tabo;
    u32* prist0 = work0 + words * nseg - words;   // pristine[k] = prist0 + k*words, k >= 1
    tabo += words * (2 * (size_t)nseg - 1);
    copy_step[0].push_back({nullptr, work0, (u32)words});
    for (u32 k = 0; k < nseg; ++k) {
      LzSegDev S;
      S.c = c; S.x0 = k * kSegBytes; S.x1 = (u32)std::min<u64>((u64)S.x0 + kSegBytes, z.n);
      S.work = work0 + words * k;
      S.pristine = k ? prist0 + words * k : nullptr;
      const u32 scap = direct ? 0u : (S.x1 - S.x0) / mmt + 3;
      S.tpos = d_tok + to; S.tlen = S.tpos + scap; S.toff = S.tlen + scap; S.tcap = scap ? scap - 1 : 0; to += (size_t)scap * 3;
      if (k) { S.qpos = d_tok + to; S.qlen = S.qpos + scap; S.qoff = S.qlen + scap; to += (size_t)scap * 3; }
      else { S.qpos = S.tpos; S.qlen = S.tlen; S.qoff = S.toff; }       // no seam walk enters a first segment (never written, never read)
      S.state = d_state + 12 * hs.size(); S.seam = S.state + 4;
      if (k) {
        copy_step[k].push_back({k == 1 ? nullptr : prist0 + words * (k - 1), S.pristine, (u32)words});
        scat_step[k].push_back({c, (k - 1) * kSegBytes, k * kSegBytes, S.pristine});
        copy_work.push_back({S.pristine, S.work, (u32)words});
      }
      hs.push_back(S);
    }
    max_n = std::max(max_n, z.n);
  }
  ZPQ_HIP(ctx, hipMemcpyAsync(d_jobs, hj.data(), nj * sizeof(LzJobDev), hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipMemcpyAsync(d_segs, hs.data(), hs.size() * sizeof(LzSegDev), hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  // 1. table states at the segment starts: pristine[k] = pristine[k-1] + inserts of segment k-1
  {
    std::vector<CopyJob> cj; std::vector<ScatterJob> sj;
    std::vector<std::pair<size_t, size_t>> crange(max_seg), srange(max_seg);
    for (u32 k = 0; k < max_seg; ++k) {
      crange[k] = {cj.size(), copy_step[k].size()}; cj.insert(cj.end(), copy_step[k].begin(), copy_step[k].end());
      srange[k] = {sj.size(), scat_step[k].size()}; sj.insert(sj.end(), scat_step[k].begin(), scat_step[k].end());
    }
    const size_t work_at = cj.size();
    cj.insert(cj.end(), copy_work.begin(), copy_work.end());
    if (!cj.empty()) ZPQ_HIP(ctx, hipMemcpyAsync(d_copy, cj.data(), cj.size() * sizeof(CopyJob), hipMemcpyHostToDevice, st));
    if (!sj.empty()) ZPQ_HIP(ctx, hipMemcpyAsync(d_scat, sj.data(), sj.size() * sizeof(ScatterJob), hipMemcpyHostToDevice, st));
    ZPQ_HIP(ctx, hipStreamSynchronize(st));
    for (u32 k = 0; k < max_seg; ++k) {
      if (crange[k].second) {
        ZPQ_LAUNCH(ctx, "lz77_table_copy_kernel", st, lz77_table_copy_kernel, dim3(256, (unsigned)crange[k].second), dim3(256),
                   d_copy + crange[k].first);
        ZPQ_HIP(ctx, hipGetLastError());
      }
      if (srange[k].second) {
        ZPQ_LAUNCH(ctx, "lz77_table_scatter_kernel", st, lz77_table_scatter_kernel, dim3(std::min<u32>(kSegBytes / 1024, 4096), (unsigned)srange[k].second),
                   dim3(256), d_scat + srange[k].first);
        ZPQ_HIP(ctx, hipGetLastError());
      }
    }
    if (!copy_work.empty()) {
      ZPQ_LAUNCH(ctx, "lz77_table_copy_kernel", st, lz77_table_copy_kernel, dim3(256, (unsigned)copy_work.size()), dim3(256),
                 d_copy + work_at);
      ZPQ_HIP(ctx, hipGetLastError());
    }
  }
  // 2. speculative parse of every segment, 3. the true chain per block -- one launch per bucket width
  std::vector<u32> lists;
  struct Rng { size_t joff, jn, soff, sn; } rng[4];
  for (int nb = 0; nb <= 3; ++nb) {
    rng[nb].joff = lists.size();
    for (size_t i = 0; i < nj; ++i) if (jobs[lo + i].args[4] == nb) lists.push_back((u32)i);
    rng[nb].jn = lists.size() - rng[nb].joff;
    rng[nb].soff = lists.size();
    for (size_t i = 0; i < nj; ++i)
      if (jobs[lo + i].args[4] == nb) for (u32 k = 0; k < hj[i].nseg; ++k) lists.push_back(hj[i].seg0 + k);
    rng[nb].sn = lists.size() - rng[nb].soff;
  }
  ZPQ_HIP(ctx, hipMemcpyAsync(d_lists, lists.data(), lists.size() * 4, hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  {
    // who gets issue priority where waves share a SIMD: the block checksum chains (default) or the segment parse (ZPQ_PRIO=lz)
    static const u32 lz_prio = [] { const char* e = getenv("ZPQ_PRIO"); return e && !strcmp(e, "lz") ? 1u : 0u; }();
    ZPQ_HIP(ctx, hipMemcpyToSymbolAsync(HIP_SYMBOL(g_lz_prio), &lz_prio, sizeof lz_prio, 0, hipMemcpyHostToDevice, st));
  }
  // three waves per block / segment on one table (lz77_waves.inc): producer | evaluator | chain
  if (direct) {
    for (int nb = 0; nb <= 3; ++nb) {
      if (!rng[nb].jn) continue;
      dim3 gj((unsigned)rng[nb].jn);
      const u32* jl = d_lists + rng[nb].joff;
      switch (nb) {
        case 0: launch_direct4<1>(ctx, st, gj, d_jobs, d_segs, jl); break;
        case 1: launch_direct4<2>(ctx, st, gj, d_jobs, d_segs, jl); break;
        case 2: launch_direct4<4>(ctx, st, gj, d_jobs, d_segs, jl); break;
        default: launch_direct4<8>(ctx, st, gj, d_jobs, d_segs, jl); break;
      }
      ZPQ_HIP(ctx, hipGetLastError());
    }
    std::vector<u32> res(nj * 4);
    ZPQ_HIP(ctx, hipMemcpyAsync(res.data(), d_res, nj * 16, hipMemcpyDeviceToHost, st));
    ZPQ_HIP(ctx, hipStreamSynchronize(st));
    for (size_t i = 0; i < nj; ++i) {
      jobs[lo + i].n_matches = res[4 * i];
      jobs[lo + i].out_len = res[4 * i + 1];
      if (res[4 * i + 2]) return zpq_fail(ctx, ZPQ_ERR_CAPACITY, "job %zu: output capacity exceeded", lo + i);
    }
    return ZPQ_OK;
  }
  for (int nb = 0; nb <= 3; ++nb) {
    if (!rng[nb].sn) continue;
    const dim3 gs((unsigned)rng[nb].sn), gj((unsigned)rng[nb].jn), blk(64);
    const u32* sl = d_lists + rng[nb].soff; const u32* jl = d_lists + rng[nb].joff;
    switch (nb) {
      case 0: launch_spec3<1>(ctx, st, gs, d_segs, sl);
              ZPQ_LAUNCH(ctx, "lz77_seam_kernel", st, lz77_seam_kernel<1>, gs, blk, d_segs, sl);
              ZPQ_LAUNCH(ctx, "lz77_stitch_kernel", st, lz77_stitch_kernel<1>, gj, blk, d_jobs, d_segs, jl); break;
      case 1: launch_spec3<2>(ctx, st, gs, d_segs, sl);
              ZPQ_LAUNCH(ctx, "lz77_seam_kernel", st, lz77_seam_kernel<2>, gs, blk, d_segs, sl);
              ZPQ_LAUNCH(ctx, "lz77_stitch_kernel", st, lz77_stitch_kernel<2>, gj, blk, d_jobs, d_segs, jl); break;
      case 2: launch_spec3<4>(ctx, st, gs, d_segs, sl);
              ZPQ_LAUNCH(ctx, "lz77_seam_kernel", st, lz77_seam_kernel<4>, gs, blk, d_segs, sl);
              ZPQ_LAUNCH(ctx, "lz77_stitch_kernel", st, lz77_stitch_kernel<4>, gj, blk, d_jobs, d_segs, jl); break;
      default: launch_spec3<8>(ctx, st, gs, d_segs, sl);
               ZPQ_LAUNCH(ctx, "lz77_seam_kernel", st, lz77_seam_kernel<8>, gs, blk, d_segs, sl);
               ZPQ_LAUNCH(ctx, "lz77_stitch_kernel", st, lz77_stitch_kernel<8>, gj, blk, d_jobs, d_segs, jl); break;
    }
    ZPQ_HIP(ctx, hipGetLastError());
  }
  {
    std::vector<u32> segjob(nseg_total);
    for (size_t i = 0; i < nj; ++i) for (u32 k = 0; k < hj[i].nseg; ++k) segjob[hj[i].seg0 + k] = (u32)i;
    ZPQ_HIP(ctx, hipMemcpyAsync(d_segjob, segjob.data(), nseg_total * 4, hipMemcpyHostToDevice, st));
    ZPQ_HIP(ctx, hipStreamSynchronize(st));
    ZPQ_LAUNCH(ctx, "lz77_move_tokens_kernel", st, lz77_move_tokens_kernel, dim3(32, (unsigned)(nseg_total * 2)), dim3(256), d_jobs, d_segs, d_segjob);
    ZPQ_HIP(ctx, hipGetLastError());
  }
  // 4. tokens -> bits (level 1) / bytes (level 2: lz77_sa.hip's pack kernels)
  {
    std::vector<zpq_lzjob_dev> j1, j2; std::vector<u32> mm2;
    u32 max1 = 0, max2 = 0;
    for (size_t i = 0; i < nj; ++i) {
      if ((jobs[lo + i].args[1] & 3) == 2) { j2.push_back(hj[i]); mm2.push_back((u32)jobs[lo + i].args[2]); max2 = std::max(max2, jobs[lo + i].n); }
      else { j1.push_back(hj[i]); max1 = std::max(max1, jobs[lo + i].n); }
    }
    if (j2.empty()) { int rc = zpq_lz77_pack_launch(ctx, d_jobs, nj, max_n); if (rc) return rc; }
    else {
      if (!j1.empty()) {
        ZPQ_HIP(ctx, hipMemcpyAsync(d_jobs, j1.data(), j1.size() * sizeof(LzJobDev), hipMemcpyHostToDevice, st));      // (the kernels above are enqueued: the records may be replaced)
        ZPQ_HIP(ctx, hipStreamSynchronize(st));
        int rc = zpq_lz77_pack_launch(ctx, d_jobs, j1.size(), max1); if (rc) return rc;
      }
      int rc = zpq_lz77_pack2_launch(ctx, j2.data(), mm2.data(), j2.size(), max2); if (rc) return rc;
    }
  }
  std::vector<u32> res(nj * 4);
  ZPQ_HIP(ctx, hipMemcpyAsync(res.data(), d_res, nj * 16, hipMemcpyDeviceToHost, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  for (size_t i = 0; i < nj; ++i) {
    jobs[lo + i].n_matches = res[4 * i];
    jobs[lo + i].out_len = res[4 * i + 1];
    if (res[4 * i + 2]) return zpq_fail(ctx, ZPQ_ERR_CAPACITY, "job %zu: token or output capacity exceeded", lo + i);
  }
  return ZPQ_OK;
}

// what compressBlock asks before it queues a block for the encoder: a refusal belongs to that block, not to the whole call
int zpq_lz77_check_args(zpq_ctx* ctx, const int32_t args[9], u32 n) { return check_args(ctx, args, n); }

// Jobs whose hash-table finder has a second context / lookahead: one wave per block (lz77_generic_kernel), a table each, in
// batches that fit the table arena.
static int encode_second_context(zpq_ctx* ctx, zpq_lz77_job* jobs, const size_t* which, size_t nj_all) {
  hipStream_t st = ctx->stream;
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  const size_t have_b = free_b + ctx->scratch_cap[0], budget = std::max<size_t>((size_t)1 << 30, have_b / 2);
  for (size_t lo = 0; lo < nj_all;) {
    size_t hi = lo, words = 0;
    while (hi < nj_all) {
      const size_t w = (size_t)1 << jobs[which[hi]].args[5];
      if (hi > lo && (words + w) * 4 > budget) break;
      words += w; ++hi;
    }
    const size_t nj = hi - lo;
    u32* d_tab = (u32*)zpq_scratch(ctx, 0, words * 4 + 256);
    u8* d_meta = (u8*)zpq_scratch(ctx, 2, nj * (sizeof(GenericJob) + 16) + 512);
    if (!d_tab || !d_meta) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "lz77 scratch (%zu MiB of tables)", words >> 18);
    GenericJob* d_jobs = (GenericJob*)d_meta;
    u32* d_res = (u32*)(d_meta + ((nj * sizeof(GenericJob) + 255) & ~(size_t)255));
    std::vector<GenericJob> hj(nj);
    size_t at = 0;
    for (size_t k = 0; k < nj; ++k) {
      const zpq_lz77_job& z = jobs[which[lo + k]];
      const int32_t* a = z.args;
      GenericJob& G = hj[k];
      G.in = z.d_in; G.n = z.n;
      G.minMatch = (u32)a[2]; G.minMatch2 = (u32)a[3]; G.lookahead = (u32)a[6]; G.bucket = (1u << a[4]) - 1u; G.htbits = (u32)a[5];
      G.checkbits = (u32)(12 - a[0]); G.shift1 = (u32)((a[5] - 1) / a[2] + 1); G.shift2 = a[3] > 0 ? (u32)((a[5] - 1) / a[3] + 1) : 0u;
      G.rb = a[0] > 4 ? (u32)(a[0] - 4) : 0u; G.level = (u32)(a[1] & 3);
      G.minMatchBoth = (u32)std::max(a[2], a[3] + a[6]) + 4u;                   // :6283
      G.ht = d_tab + at; at += (size_t)1 << a[5];
      G.out = z.d_out; G.out_cap = z.out_cap; G.result = d_res + 4 * k;
    }
    ZPQ_HIP(ctx, hipMemsetAsync(d_tab, 0, words * 4, st));
    ZPQ_HIP(ctx, hipMemsetAsync(d_res, 0, nj * 16, st));
    ZPQ_HIP(ctx, hipMemcpyAsync(d_jobs, hj.data(), nj * sizeof(GenericJob), hipMemcpyHostToDevice, st));
    ZPQ_HIP(ctx, hipStreamSynchronize(st));
    ZPQ_LAUNCH(ctx, "lz77_generic_kernel", st, lz77_generic_kernel, dim3((unsigned)nj), dim3(64), d_jobs);
    ZPQ_HIP(ctx, hipGetLastError());
    std::vector<u32> res(nj * 4);
    ZPQ_HIP(ctx, hipMemcpyAsync(res.data(), d_res, nj * 16, hipMemcpyDeviceToHost, st));
    ZPQ_HIP(ctx, hipStreamSynchronize(st));
    for (size_t k = 0; k < nj; ++k) {
      jobs[which[lo + k]].n_matches = res[4 * k];
      jobs[which[lo + k]].out_len = res[4 * k + 1];
      if (res[4 * k + 2]) return zpq_fail(ctx, ZPQ_ERR_CAPACITY, "job %zu: output capacity exceeded", which[lo + k]);
    }
    lo = hi;
  }
  return ZPQ_OK;
}

extern "C" int zpq_lz77_encode_dev(zpq_ctx* ctx, zpq_lz77_job* jobs, size_t njobs) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (njobs == 0) return ZPQ_OK;
  for (size_t i = 0; i < njobs; ++i) {
    int rc = check_args(ctx, jobs[i].args, jobs[i].n);
    if (rc) return rc;
    if (jobs[i].out_cap < zpq_lz77_bound(jobs[i].n)) return zpq_fail(ctx, ZPQ_ERR_CAPACITY, "job %zu: out_cap too small", i);
    if (((uintptr_t)jobs[i].d_out & 3) != 0) return zpq_fail(ctx, ZPQ_ERR_ARG, "job %zu: d_out must be 4-byte aligned", i);
  }
  {   // suffix-array jobs take their own path; the rest are parsed with hash tables below
    std::vector<size_t> sa_jobs;
    for (size_t i = 0; i < njobs; ++i) if (uses_suffix_array(jobs[i].args)) sa_jobs.push_back(i);
    if (!sa_jobs.empty()) {
      int rc = zpq_lz77_sa_encode(ctx, jobs, sa_jobs.data(), sa_jobs.size());
      if (rc) return rc;
      if (sa_jobs.size() == njobs) return ZPQ_OK;
      std::vector<zpq_lz77_job> rest;
      std::vector<size_t> at;
      for (size_t i = 0; i < njobs; ++i) if (!uses_suffix_array(jobs[i].args)) { rest.push_back(jobs[i]); at.push_back(i); }
      rc = zpq_lz77_encode_dev(ctx, rest.data(), rest.size());
      for (size_t k = 0; k < rest.size(); ++k) jobs[at[k]] = rest[k];
      return rc;
    }
  }
  {   // ... and so do the hash-table jobs with a second context or lookahead (one wave per block, lz77_generic_kernel)
    std::vector<size_t> g_jobs;
    for (size_t i = 0; i < njobs; ++i) if (uses_second_context(jobs[i].args)) g_jobs.push_back(i);
    if (!g_jobs.empty()) {
      int rc = encode_second_context(ctx, jobs, g_jobs.data(), g_jobs.size());
      if (rc) return rc;
      if (g_jobs.size() == njobs) return ZPQ_OK;
      std::vector<zpq_lz77_job> rest;
      std::vector<size_t> at;
      for (size_t i = 0; i < njobs; ++i) if (!uses_second_context(jobs[i].args)) { rest.push_back(jobs[i]); at.push_back(i); }
      rc = zpq_lz77_encode_dev(ctx, rest.data(), rest.size());
      for (size_t k = 0; k < rest.size(); ++k) jobs[at[k]] = rest[k];
      return rc;
    }
  }
  // A wave parses one segment on its own copy of the hash table (64 MiB for -m1), so HBM bounds the waves in flight.
  // Segment size: 1 MiB while everything fits one batch; with many blocks the segments grow (up to one per block:
  // "one wavefront per ZPAQ block") until the batch fits or there are too few waves left to fill the chip, and what
  // still does not fit runs in batches.
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  // what is free now plus what this path already holds, less a reserve for everything else that allocates later
  const size_t have_b = free_b + ctx->scratch_cap[0] + ctx->scratch_cap[1], reserve_b = std::max<size_t>((size_t)8 << 30, total_b / 10);
  size_t budget = std::max<size_t>((size_t)2 << 30, have_b > reserve_b ? have_b - reserve_b : 0);
  if (const char* e = getenv("ZPQ_LZ_BUDGET_MB")) budget = (size_t)strtoull(e, 0, 10) << 20;
  u32 seg = kSegMin, max_n = 0;
  if (const char* e = getenv("ZPQ_LZ_SEG")) seg = std::max<u32>(1u << 16, (u32)strtoul(e, 0, 10));
  else {
    for (size_t i = 0; i < njobs; ++i) max_n = std::max(max_n, jobs[i].n);
    for (;;) {
      size_t bytes = 0, nseg = 0;
      for (size_t i = 0; i < njobs; ++i) { bytes += job_bytes(jobs[i], seg); nseg += std::max<u32>(1, (u32)(((u64)jobs[i].n + seg - 1) / seg)); }
      if (bytes <= budget || seg >= max_n || nseg <= 2048 || seg >= (1u << 30)) break;
      seg <<= 1;
    }
  }
  // Two ways to run a batch.  Speculative segments: many waves per block, 2*segments-1 tables and token lists per block
  // (few blocks: 13 blocks become 212 waves).  Direct: one wave per block that parses and emits in one go, one table per
  // block and nothing else -- with hundreds of blocks there are waves enough without speculation and HBM is what bounds
  // the number in flight.  A wave parses at a rate that does not depend on the mode, so the estimate of either is
  // (number of batches) x (bytes one wave walks); the smaller wins.
  size_t all_bytes = 0, all_direct = 0;
  for (size_t i = 0; i < njobs; ++i) { all_bytes += job_bytes(jobs[i], seg); all_direct += job_bytes_direct(jobs[i]); }
  for (size_t i = 0; i < njobs; ++i) max_n = std::max(max_n, jobs[i].n);
  const size_t nbatch = (all_bytes + budget - 1) / budget, nbatch_d = (all_direct + budget - 1) / budget;
  bool direct = (double)nbatch_d * (double)max_n < (double)nbatch * (double)std::min<u32>(seg, max_n ? max_n : 1);
  if (const char* e = getenv("ZPQ_LZ_DIRECT")) direct = atoi(e) != 0;
  for (size_t i = 0; i < njobs; ++i) if ((jobs[i].args[1] & 3) == 2) direct = false;      // (the chain wave writes bit codes only: byte codes go through the token lists)
  // batches of about equal size (a last batch of a few blocks would leave the chip idle behind its slowest wave)
  const size_t total = direct ? all_direct : all_bytes, nb_ = direct ? nbatch_d : nbatch;
  const size_t target = nb_ > 1 ? std::min(budget, total / nb_ + (total / nb_) / 16) : budget;
  size_t lo = 0;
  while (lo < njobs) {
    size_t hi = lo, bytes = 0;
    while (hi < njobs) {
      const size_t b = direct ? job_bytes_direct(jobs[hi]) : job_bytes(jobs[hi], seg);
      if (hi > lo && bytes + b > target) break;
      bytes += b; ++hi;
    }
    int rc = encode_batch(ctx, jobs, lo, hi, direct ? (1u << 30) : seg, direct);
    if (rc) return rc;
    lo = hi;
  }
  return ZPQ_OK;
}
#endif  // ZPQ_EMU_WALK_ONLY
