// LZ77 level-1 ("lazy2") encoder and decoder -- the codec behind zpaqfranz -m1.
// Reference: LZBuffer::LZBuffer/fill/write_literal/write_match/putb/flush, ZSFX/libzpaq.cpp:6140-6552
// (code format :6211-6222); the decoder restates what the level-1 PCOMP program (SURVEY.md
// Appendix D) computes when PostProcessor (ZSFX/libzpaq.cpp:2178-2233) runs it per byte.
//
// The output is bit-identical to LZBuffer's.  What makes that possible on a GPU:
//  * Every input position is inserted into the hash table whatever the parse chose
//    (ZSFX/libzpaq.cpp:6432-6447) and slot/value are pure functions of the input, so the table
//    contents seen at position i do not depend on the parse.  The rolling hash h1 has a finite
//    window ((5<<shift1)^minMatch == 0 mod table size), so it is recomputed from the input bytes.
//  * One wave walks a block in windows of 64 positions.  All 64 lanes look their position up at
//    once (table loads + candidate compares, capped at 32 bytes), inserts made by earlier lanes of
//    the same window are forwarded through an LDS collision mask, and each lane precomputes the
//    reference's decision (:6396-6421) for both values of the (lit>0) score term.
//  * Only the greedy chain "take the match and skip blen, or emit one literal" is serial: a
//    wave-uniform loop over v_readlane'd per-lane results.  Positions whose candidates hit the
//    32-byte cap are re-evaluated exactly with a cooperative 512-bytes-per-step compare.
//  * The parse emits match tokens; bit offsets come from a workgroup scan, then code bits and
//    literal bytes are OR-ed into the zeroed output in parallel (LSB-first, :6171-6186).
// Integer/byte work on random table slots: latency bound, no MFMA.  Traffic per block of n bytes:
// n read + (2^args[5]*4) table zero/update + r*n written (r = LZ ratio).
#include "zpq_internal.h"

namespace {

constexpr u32 kMaxMatch = (1u << 14) * 3;   // ZSFX/libzpaq.cpp:6258 (BUFSIZE*3)
constexpr u32 kMaxLiteral = (1u << 14) / 4;  // :6259
constexpr u32 kCap = 32;                     // speculative compare cap (bytes)
constexpr u32 kNoCand = 0xffffffffu;

struct LzJobDev {
  const u8* in;
  u32 n;
  u32 minMatch, bucket, htbits, checkbits, shift1, rb;
  u32 upd_limit;   // positions < upd_limit are inserted (i + minMatchBoth < n)
  u32* ht;         // 2^htbits entries, zeroed
  u32* tok_pos; u32* tok_len; u32* tok_off; u32* tok_bit;
  u32 tok_cap;
  u32* result;     // [0]=ntok, [1]=out_len bytes, [2]=overflow flag
  u8* out; u32 out_cap;
};

__device__ __forceinline__ int lg32(u32 x) { return x ? 32 - __builtin_clz(x) : 0; }  // lg(), :6224-6233

__device__ __forceinline__ u64 load8(const u8* p) { return *(const u64_u*)p; }

// h1 as LZBuffer holds it when it reaches position q (:6444): the rolling hash over the last
// minMatch update steps, i.e. over in[U..U+minMatch-1] with U = min(q, upd_limit); partial for U < minMatch.
__device__ __forceinline__ u32 hash_at(const LzJobDev& J, u32 q) {
  const u32 U = q < J.upd_limit ? q : J.upd_limit;
  const u32 mm = J.minMatch;
  const u32 F = 5u << J.shift1;
  u32 h = 0;
  const u32 t0 = U > mm ? U - mm : 0;
  for (u32 t = t0; t < U; ++t) h = h * F + (J.in[t + mm] + 1u) * 123456791u;
  return h & ((1u << J.htbits) - 1u);
}

// Match length of in[p..] vs in[q..], at most `limit`, 8 bytes per step (buffers are padded).
__device__ __forceinline__ u32 match_len(const u8* in, u32 p, u32 q, u32 limit) {
  u32 l = 0;
  while (l < limit) {
    u64 x = load8(in + p + l) ^ load8(in + q + l);
    if (x) { l += (u32)(__builtin_ctzll(x) >> 3); break; }
    l += 8;
  }
  return l < limit ? l : limit;
}

// Whole-wave compare for long matches: 512 bytes per step.  All arguments wave-uniform.
__device__ __forceinline__ u32 coop_match_len(const u8* in, u32 p, u32 q, u32 limit) {
  const u32 lane = (u32)lane_id();
  u32 base = 0;
  while (base < limit) {
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

template <int NB>
__global__ __launch_bounds__(64) void lz77_parse_kernel(const LzJobDev* __restrict__ jobs) {
  const LzJobDev J = jobs[blockIdx.x];
  const u32 lane = (u32)lane_id();
  const u8* in = J.in;
  const u32 n = J.n;
  const u32 mask = (1u << J.checkbits) - 1u;
  const u32 mm = J.minMatch;
  __shared__ unsigned long long T[256];  // collision masks keyed by (group & 255)
  T[lane] = 0; T[lane + 64] = 0; T[lane + 128] = 0; T[lane + 192] = 0;
  __builtin_amdgcn_wave_barrier();
  volatile unsigned long long* Tv = T;

  u32 cur = 0, lit = 0, ntok = 0;
  for (u32 base = 0; base < n; base += 64) {
    const u32 q = base + lane;
    const bool inb = q < n;
    // ---- per-position hash, slot, value --------------------------------------------------------
    const u32 h = inb ? hash_at(J, q) : 0u;
    const bool ins = inb && q < J.upd_limit;
    const u32 ih = ((q * 1234547u) >> 19) & J.bucket;                     // :6435
    const u32 slot = h ^ ih;
    const u32 b3 = (inb && q + 3 < n) ? in[q + 3] : 0u;
    const u32 val = (q << J.checkbits) | (b3 & mask);                      // :6436
    const u32 grp = h & ~J.bucket;
    const bool look = inb && cur < base + 64;  // windows swallowed by a match only insert

    u32 ent[NB];
    // ---- table group load (bypasses L1: the table is rewritten by this wave) --------------------
    if (look) {
#pragma unroll
      for (int j = 0; j < NB; ++j) ent[j] = __builtin_nontemporal_load(J.ht + grp + j);
    } else {
#pragma unroll
      for (int j = 0; j < NB; ++j) ent[j] = 0;
    }
    // ---- forward inserts of earlier lanes in this window; find superseded stores ---------------
    bool superseded = false;
    {
      const u32 tk = (grp >> 3) & 255u;  // NB <= 8: groups are at least 1 slot wide
      if (inb) atomicOr((unsigned long long*)&T[tk], 1ull << lane);
      __builtin_amdgcn_wave_barrier();
      unsigned long long cm = inb ? Tv[tk] : 0ull;
      cm &= ~(1ull << lane);
      while (__ballot(cm != 0)) {
        const int k = cm ? __builtin_ctzll(cm) : 0;  // ascending: later lanes overwrite earlier ones
        const u32 sk = __shfl(slot, k), vk = __shfl(val, k);
        const bool ik = __shfl((int)ins, k) != 0;
        if (cm) {
          if (ik && (u32)k < lane && (sk & ~J.bucket) == grp) {
#pragma unroll
            for (int j = 0; j < NB; ++j)
              if ((sk & J.bucket) == (u32)j) ent[j] = vk;
          }
          if (ik && (u32)k > lane && sk == slot) superseded = true;
          cm &= cm - 1;
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (inb) Tv[tk] = 0ull;
    }
    // ---- reorder the group into probe order ht[h1^k], k = 0..bucket (:6397) ---------------------
    {
      const u32 hb = h & J.bucket;
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
    u32 cp[NB], cl[NB];
    bool slow = false;
    const u32 limit = inb ? (n - q < kMaxMatch ? n - q : kMaxMatch) : 0u;
    const bool evalp = look && q >= cur;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      cp[k] = kNoCand; cl[k] = 0;
      const u32 e = ent[k];
      if (evalp && e && q + 3 < n && (e & mask) == (b3 & mask)) {           // :6398
        const u32 p = e >> J.checkbits;
        if (p < q) {
          const u32 lim = limit < kCap ? limit : kCap;
          const u32 l = match_len(in, p, q, lim);
          cp[k] = p; cl[k] = l;
          if (l == kCap && limit > kCap) slow = true;
        }
      }
    }
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
              else ok = in[p + idx] == in[q + idx];
            }
            if (ok) {
              const int score = (int)(l * 8) - lg32(q - p) - 2 * f - 11;
              if (score > bscore) { blen = l; bp = p; bscore = score; }
            }
          }
        }
        const u32 off = q - bp;
        if (off > 0 && bscore > 0 && blen >= mm) { rlen[f] = blen; roff[f] = off; }
      }
    }
    // ---- serial greedy chain over this window (wave-uniform) -----------------------------------
    const unsigned long long slowmask = __ballot(slow);
    const u32 wend = base + 64 < n ? base + 64 : n;
    while (cur < wend) {
      const u32 j = cur - base;
      const u32 f = lit > 0 ? 1u : 0u;
      u32 tlen, toff;
      if ((slowmask >> j) & 1ull) {
        // exact re-evaluation of position cur, replicating :6396-6408 with whole-wave compares
        const u32 i = cur;
        const u32 lim_i = n - i < kMaxMatch ? n - i : kMaxMatch;
        const u32 bi3 = in[i + 3];
        u32 blen = mm - 1, bp = 0; int bscore = 0;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          const u32 e = __builtin_amdgcn_readlane(ent[k], j);
          if (blen < 128 && e && i + 3 < n && (e & mask) == (bi3 & mask)) {
            const u32 p = e >> J.checkbits;
            if (p < i && i + blen <= n && in[p + blen - 1] == in[i + blen - 1]) {
              const u32 l = coop_match_len(in, p, i, lim_i);
              const int score = (int)(l * 8) - lg32(i - p) - 2 * (int)f - 11;
              if (score > bscore) { blen = l; bp = p; bscore = score; }
            }
          }
        }
        const u32 off = i - bp;
        const bool take = off > 0 && bscore > 0 && blen >= mm;
        tlen = take ? blen : 0u; toff = off;
      } else {
        const u32 l0 = __builtin_amdgcn_readlane(rlen[0], j), l1 = __builtin_amdgcn_readlane(rlen[1], j);
        const u32 o0 = __builtin_amdgcn_readlane(roff[0], j), o1 = __builtin_amdgcn_readlane(roff[1], j);
        tlen = f ? l1 : l0; toff = f ? o1 : o0;
      }
      if (tlen) {
        if (lane == 0) {
          if (ntok < J.tok_cap) { J.tok_pos[ntok] = cur; J.tok_len[ntok] = tlen; J.tok_off[ntok] = toff; }
        }
        ++ntok;
        lit = 0;
        cur += tlen;
      } else {
        ++lit; ++cur;
        if (lit >= kMaxLiteral) lit = 0;  // forced literal flush (:6450-6451); runs are positional
      }
    }
    // ---- insert this window's positions (latest writer of a slot wins) --------------------------
    if (ins && !superseded) J.ht[slot] = val;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (lane == 0) {
    J.result[0] = ntok < J.tok_cap ? ntok : J.tok_cap;
    if (ntok > J.tok_cap) J.result[2] = 1;
  }
}

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
// headers and match codes.  Literal bytes are written by lz77_pack_literals_kernel.
__global__ __launch_bounds__(1024) void lz77_pack_tokens_kernel(const LzJobDev* __restrict__ jobs) {
  const LzJobDev J = jobs[blockIdx.x];
  const u32 ntok = J.result[0];
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ u64 wsum[16];
  __shared__ u64 carry_s;
  if (tid == 0) carry_s = 0;
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
    }
    __syncthreads();
    if (tid == 1023) carry_s = carry + wbase + x;
    __syncthreads();
  }
  if (tid == 0) {
    const u64 bits = carry_s;
    const u64 bytes = (bits + 7) >> 3;
    J.result[1] = (u32)bytes;
    if (bytes > J.out_cap) J.result[2] = 1;
  }
}

// Literal bytes: one thread per input position.
__global__ __launch_bounds__(256) void lz77_pack_literals_kernel(const LzJobDev* __restrict__ jobs) {
  const LzJobDev J = jobs[blockIdx.y];
  const u32 x = blockIdx.x * 256u + threadIdx.x;
  const u32 ntok = J.result[0];
  // wave-uniform lower bound for the first position of this wave, then a short per-lane search
  const u32 x0 = __builtin_amdgcn_readfirstlane(x);
  if (x0 >= J.n) return;
  u32 lo = 0, hi = ntok;  // first token with tok_pos > x0
  while (lo < hi) { u32 mid = (lo + hi) >> 1; if (J.tok_pos[mid] > x0) hi = mid; else lo = mid + 1; }
  if (x >= J.n) return;
  u32 t = lo;
  while (t < ntok && J.tok_pos[t] <= x) ++t;  // <= 16 steps: matches are >= 4 bytes apart
  const u32 gstart = t ? J.tok_pos[t - 1] + J.tok_len[t - 1] : 0u;
  if (x < gstart) return;  // inside match t-1
  const u32 gend = t < ntok ? J.tok_pos[t] : J.n;
  const u32 g = gend - gstart, r = x - gstart;
  const u32 run = r / kMaxLiteral, within = r % kMaxLiteral;
  const u32 runlen = (run < g / kMaxLiteral) ? kMaxLiteral : g % kMaxLiteral;
  const u64 bp = (u64)J.tok_bit[t] + (u64)run * (lit_run_header_bits(kMaxLiteral) + 8ull * kMaxLiteral) +
                 lit_run_header_bits(runlen) + 8ull * within;
  or_bits((u32*)J.out, bp, J.in[x], 8);
}

// ---- decoder ---------------------------------------------------------------------------------------
struct LzDecDev {
  const u8* in; u32 n; u32 rb;
  u8* out; u32 out_cap;
  u32* result;  // [0]=out_len, [1]=status
};

constexpr u32 kRing = 1u << 16;

// One wave per block.  The bit parser is wave-uniform; copies are spread over the lanes.  The last
// 64 KiB of output live in an LDS ring so that near matches never wait on HBM stores.
__global__ __launch_bounds__(64) void lz77_decode_kernel(const LzDecDev* __restrict__ jobs) {
  const LzDecDev J = jobs[blockIdx.x];
  __shared__ u8 ring[kRing];
  const u32 lane = (u32)lane_id();
  const u8* in = J.in;
  const u64 nbits = (u64)J.n * 8;
  u64 bp = 0; u32 op = 0; int status = ZPQ_OK;
#define PEEK() (load8(in + (bp >> 3)) >> (bp & 7))
  for (;;) {
    if (bp + 2 > nbits) break;
    u64 w = PEEK();
    const u32 mmv = (u32)(w & 3);
    u64 used = 2; w >>= 2;
    if (mmv == 0) {                                   // literal run: gamma length then bytes
      u32 len = 1; bool trunc = false;
      for (;;) {
        if (bp + used + 1 > nbits) { trunc = true; break; }
        const u32 b = (u32)(w & 1); w >>= 1; ++used;
        if (!b) break;
        if (bp + used + 1 > nbits) { trunc = true; break; }
        len = len * 2 + (u32)(w & 1); w >>= 1; ++used;
      }
      if (trunc) break;
      bp += used;
      const u64 avail = (nbits - bp) >> 3;
      const bool cutoff = avail < len;
      if (cutoff) len = (u32)avail;                    // stream ends inside the run
      if (op + len > J.out_cap) { status = ZPQ_ERR_CAPACITY; break; }
      for (u32 j = lane; j < len; j += 64) {
        const u64 b = bp + 8ull * j;
        const u32 two = (u32)in[b >> 3] | ((u32)in[(b >> 3) + 1] << 8);
        const u8 c = (u8)(two >> (b & 7));
        J.out[op + j] = c;
        ring[(op + j) & (kRing - 1)] = c;
      }
      __builtin_amdgcn_wave_barrier();
      op += len; bp += 8ull * len;
      if (cutoff) break;
    } else {                                          // match
      if (bp + 5 > nbits) break;
      const u32 lo = (mmv - 1) * 8 + (u32)(w & 7); w >>= 3; used += 3;
      u32 len = 1; bool trunc = false;
      for (;;) {
        if (bp + used + 1 > nbits) { trunc = true; break; }
        const u32 b = (u32)(w & 1); w >>= 1; ++used;
        if (!b) break;
        if (bp + used + 1 > nbits) { trunc = true; break; }
        len = len * 2 + (u32)(w & 1); w >>= 1; ++used;
      }
      if (trunc || bp + used + 2 > nbits) break;
      len = len * 4 + (u32)(w & 3); used += 2;
      bp += used;                                     // used <= 2+3+2*15+1+2 = 38 bits
      if (bp + J.rb + lo > nbits) break;
      w = PEEK();
      const u32 r = (u32)(w & ((1ull << J.rb) - 1)); w >>= J.rb;
      const u32 qv = (u32)(w & ((1ull << lo) - 1)) | (1u << lo);
      bp += J.rb + lo;
      const u32 off = ((qv << J.rb) | r) - ((1u << J.rb) - 1u);
      if (off > op) { status = ZPQ_ERR_FORMAT; break; }
      if (op + len > J.out_cap) { status = ZPQ_ERR_CAPACITY; break; }
      const u32 src0 = op - off;
      if (off + 64 <= kRing) {                        // source inside the LDS ring
        for (u32 c0 = 0; c0 < len; c0 += 64) {
          const u32 j = c0 + lane;
          u8 c = 0;
          // off >= 64: out[op+j-off] was written before this chunk; off < 64: periodic extension
          if (j < len) c = ring[(off >= 64 ? op + j - off : src0 + (j % off)) & (kRing - 1)];
          __builtin_amdgcn_wave_barrier();
          if (j < len) { J.out[op + j] = c; ring[(op + j) & (kRing - 1)] = c; }
          __builtin_amdgcn_wave_barrier();
        }
      } else {                                        // far match: bytes written >= 64 KiB ago, len < off
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (u32 j = lane; j < len; j += 64) {
          const u8 c = __builtin_nontemporal_load(J.out + src0 + j);
          J.out[op + j] = c;
          ring[(op + j) & (kRing - 1)] = c;
        }
        __builtin_amdgcn_wave_barrier();
      }
      op += len;
    }
  }
#undef PEEK
  if (lane == 0) { J.result[0] = op; J.result[1] = (u32)status; }
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------
extern "C" size_t zpq_lz77_bound(size_t n) { return n + n / 512 + 64; }

static int check_args(zpq_ctx* ctx, const int32_t a[9], u32 n) {
  if ((a[1] & 3) != 1 || a[1] > 5) return zpq_fail(ctx, ZPQ_ERR_METHOD, "LZ77 level %d not implemented", a[1]);
  if (a[3] != 0 || a[6] != 0) return zpq_fail(ctx, ZPQ_ERR_METHOD, "secondary context not implemented");
  if (a[2] < 4 || a[2] > 31) return zpq_fail(ctx, ZPQ_ERR_METHOD, "min match %d out of range", a[2]);
  if (a[4] < 0 || a[4] > 3) return zpq_fail(ctx, ZPQ_ERR_METHOD, "bucket 2^%d not implemented", a[4]);
  if (a[0] < 0 || a[0] > 6 || a[5] - a[0] >= 21 || a[5] < 4 || a[5] > 26 || a[5] <= a[4])
    return zpq_fail(ctx, ZPQ_ERR_METHOD, "hash table 2^%d out of range", a[5]);
  if ((u64)n > (1ull << (20 + a[0]))) return zpq_fail(ctx, ZPQ_ERR_ARG, "block of %u bytes exceeds 2^%d", n, 20 + a[0]);
  return ZPQ_OK;
}

extern "C" int zpq_lz77_encode_dev(zpq_ctx* ctx, zpq_lz77_job* jobs, size_t njobs) {
  if (njobs == 0) return ZPQ_OK;
  hipStream_t st = ctx->stream;
  // scratch: per job hash table + tokens (4 arrays) + result[4]
  std::vector<LzJobDev> h(njobs);
  size_t ht_total = 0, tok_total = 0;
  for (size_t i = 0; i < njobs; ++i) {
    int rc = check_args(ctx, jobs[i].args, jobs[i].n);
    if (rc) return rc;
    if (jobs[i].out_cap < zpq_lz77_bound(jobs[i].n)) return zpq_fail(ctx, ZPQ_ERR_CAPACITY, "job %zu: out_cap too small", i);
    if (((uintptr_t)jobs[i].d_out & 3) != 0) return zpq_fail(ctx, ZPQ_ERR_ARG, "job %zu: d_out must be 4-byte aligned", i);
    ht_total += (size_t)1 << jobs[i].args[5];
    tok_total += (size_t)jobs[i].n / 4 + 2;
  }
  u32* d_ht = (u32*)zpq_scratch(ctx, 0, ht_total * 4);
  u32* d_tok = (u32*)zpq_scratch(ctx, 1, tok_total * 16);
  u8* d_meta = (u8*)zpq_scratch(ctx, 2, njobs * (sizeof(LzJobDev) + 16) + 64);
  if (!d_ht || !d_tok || !d_meta) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "lz77 scratch");
  LzJobDev* d_jobs = (LzJobDev*)d_meta;
  u32* d_res = (u32*)(d_meta + njobs * sizeof(LzJobDev));
  ZPQ_HIP(ctx, hipMemsetAsync(d_ht, 0, ht_total * 4, st));
  ZPQ_HIP(ctx, hipMemsetAsync(d_res, 0, njobs * 16, st));
  size_t ho = 0, to = 0;
  u32 max_n = 0;
  for (size_t i = 0; i < njobs; ++i) {
    const int32_t* a = jobs[i].args;
    LzJobDev& J = h[i];
    J.in = jobs[i].d_in; J.n = jobs[i].n;
    J.minMatch = a[2]; J.bucket = (1u << a[4]) - 1; J.htbits = a[5]; J.checkbits = 12 - a[0];
    J.shift1 = (a[5] - 1) / a[2] + 1; J.rb = a[0] > 4 ? a[0] - 4 : 0;
    const u32 mmb = a[2] + 4;
    J.upd_limit = J.n > mmb ? J.n - mmb : 0;
    J.ht = d_ht + ho; ho += (size_t)1 << a[5];
    const u32 cap = J.n / 4 + 2;
    J.tok_pos = d_tok + to; J.tok_len = J.tok_pos + cap; J.tok_off = J.tok_len + cap; J.tok_bit = J.tok_off + cap;
    J.tok_cap = cap - 1; to += (size_t)cap * 4;
    J.result = d_res + 4 * i;
    J.out = jobs[i].d_out; J.out_cap = jobs[i].out_cap;
    ZPQ_HIP(ctx, hipMemsetAsync(J.out, 0, J.out_cap, st));
    if (J.n > max_n) max_n = J.n;
  }
  ZPQ_HIP(ctx, hipMemcpyAsync(d_jobs, h.data(), njobs * sizeof(LzJobDev), hipMemcpyHostToDevice, st));
  // one launch per bucket width present
  for (int nbits = 0; nbits <= 3; ++nbits) {
    std::vector<u32> idx;
    for (size_t i = 0; i < njobs; ++i) if (jobs[i].args[4] == nbits) idx.push_back((u32)i);
    if (idx.empty()) continue;
    // jobs of one width are launched over a contiguous copy of their descriptors
    std::vector<LzJobDev> sub(idx.size());
    for (size_t k = 0; k < idx.size(); ++k) sub[k] = h[idx[k]];
    LzJobDev* d_sub = (LzJobDev*)zpq_scratch(ctx, 8, njobs * sizeof(LzJobDev) * 4) + (size_t)nbits * njobs;
    if (!d_sub) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "lz77 scratch");
    ZPQ_HIP(ctx, hipMemcpyAsync(d_sub, sub.data(), sub.size() * sizeof(LzJobDev), hipMemcpyHostToDevice, st));
    ZPQ_HIP(ctx, hipStreamSynchronize(st));  // `sub` is pageable host memory
    dim3 grid((unsigned)idx.size()), blk(64);
    switch (nbits) {
      case 0: ZPQ_LAUNCH(ctx, "lz77_parse_kernel", st, lz77_parse_kernel<1>, grid, blk, d_sub); break;
      case 1: ZPQ_LAUNCH(ctx, "lz77_parse_kernel", st, lz77_parse_kernel<2>, grid, blk, d_sub); break;
      case 2: ZPQ_LAUNCH(ctx, "lz77_parse_kernel", st, lz77_parse_kernel<4>, grid, blk, d_sub); break;
      default: ZPQ_LAUNCH(ctx, "lz77_parse_kernel", st, lz77_parse_kernel<8>, grid, blk, d_sub); break;
    }
    ZPQ_HIP(ctx, hipGetLastError());
  }
  ZPQ_LAUNCH(ctx, "lz77_pack_tokens_kernel", st, lz77_pack_tokens_kernel, dim3((unsigned)njobs), dim3(1024), d_jobs);
  ZPQ_HIP(ctx, hipGetLastError());
  if (max_n) {
    ZPQ_LAUNCH(ctx, "lz77_pack_literals_kernel", st, lz77_pack_literals_kernel, dim3((max_n + 255) / 256, (unsigned)njobs), dim3(256), d_jobs);
    ZPQ_HIP(ctx, hipGetLastError());
  }
  std::vector<u32> res(njobs * 4);
  ZPQ_HIP(ctx, hipMemcpyAsync(res.data(), d_res, njobs * 16, hipMemcpyDeviceToHost, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  for (size_t i = 0; i < njobs; ++i) {
    jobs[i].n_matches = res[4 * i];
    jobs[i].out_len = res[4 * i + 1];
    if (res[4 * i + 2]) return zpq_fail(ctx, ZPQ_ERR_CAPACITY, "job %zu: token or output capacity exceeded", i);
  }
  return ZPQ_OK;
}

extern "C" int zpq_lz77_decode_dev(zpq_ctx* ctx, zpq_lz77_dec_job* jobs, size_t njobs) {
  if (njobs == 0) return ZPQ_OK;
  hipStream_t st = ctx->stream;
  u8* d_meta = (u8*)zpq_scratch(ctx, 2, njobs * (sizeof(LzDecDev) + 8) + 64);
  if (!d_meta) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "lz77 scratch");
  LzDecDev* d_jobs = (LzDecDev*)d_meta;
  u32* d_res = (u32*)(d_meta + njobs * sizeof(LzDecDev));
  std::vector<LzDecDev> h(njobs);
  for (size_t i = 0; i < njobs; ++i) {
    if (jobs[i].rb > 8) return zpq_fail(ctx, ZPQ_ERR_ARG, "rb out of range");
    h[i].in = jobs[i].d_in; h[i].n = jobs[i].n; h[i].rb = jobs[i].rb;
    h[i].out = jobs[i].d_out; h[i].out_cap = jobs[i].out_cap; h[i].result = d_res + 2 * i;
  }
  ZPQ_HIP(ctx, hipMemcpyAsync(d_jobs, h.data(), njobs * sizeof(LzDecDev), hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  ZPQ_LAUNCH(ctx, "lz77_decode_kernel", st, lz77_decode_kernel, dim3((unsigned)njobs), dim3(64), d_jobs);
  ZPQ_HIP(ctx, hipGetLastError());
  std::vector<u32> res(njobs * 2);
  ZPQ_HIP(ctx, hipMemcpyAsync(res.data(), d_res, njobs * 8, hipMemcpyDeviceToHost, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  for (size_t i = 0; i < njobs; ++i) { jobs[i].out_len = res[2 * i]; jobs[i].status = (int32_t)res[2 * i + 1]; }
  return ZPQ_OK;
}
