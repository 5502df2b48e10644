// LZ77 level-1 decoder: what the 302-byte level-1 PCOMP program (code format
// ZSFX/libzpaq.cpp:6211-6222) computes when PostProcessor (ZSFX/libzpaq.cpp:2178-2233) runs it per byte.
//
// One wave per block.  The bit parser is wave-uniform and never waits on memory: the code stream is held
// in registers as a sliding window (each lane keeps 8 bytes of the current 512-byte chunk and of the next
// one; 64 bits at any bit position are two v_readlane pairs), the interleaved Elias-gamma lengths are
// decoded with one count-trailing-zeros and a bit compaction instead of a bit loop.  Copies are spread over
// the lanes; the last 64 KiB of output live in an LDS ring so that near matches never wait on HBM stores.
//
// The stream is untrusted (archives come from anywhere): lengths are capped (<= 24 gamma doublings, far
// beyond what any encoder emits), capacity is checked in 64 bits, offsets must point into produced output,
// and far matches longer than their offset are copied in pieces that only read what is already written.
#include <stdlib.h>

#include "zpq_internal.h"

namespace {

// The code stream and the output are addressed as GLOBAL memory, the ring and the staged chunks as LDS.  With generic
// pointers every access is a flat_* instruction, which counts on BOTH the vector-memory and the LDS counters: each LDS
// read of a match copy then waits for the HBM stores of the token before it (measured: 2600 cycles per token).
typedef __attribute__((address_space(1))) const u8 g_cu8;
typedef __attribute__((address_space(1))) u8 g_u8;
typedef __attribute__((address_space(1))) const u64_u g_cu64_u;
typedef __attribute__((address_space(3))) u8 l_u8;
typedef __attribute__((address_space(3))) u64 l_u64;
__device__ __forceinline__ u64 load8(g_cu8* p) { return *(g_cu64_u*)p; }

typedef zpq_lzdec_dev LzDecDev;

constexpr u32 kRing = 1u << 16;

__device__ __forceinline__ u64 readlane64(u64 v, u32 l) {
  const u32 lo = __builtin_amdgcn_readlane((u32)v, l), hi = __builtin_amdgcn_readlane((u32)(v >> 32), l);
  return (u64)lo | (u64)hi << 32;
}

// Interleaved Elias gamma, LSB first: pairs (1, bit) and a closing 0; the value has an implied leading 1 and
// the first bit read is the most significant.  w = the bits at the code; returns the value, nb = bits used
// (2k+1); k > 24 (or no terminator in the window) sets nb = 0xffffffff.
__device__ __forceinline__ u32 gamma_decode(u64 w, u32& nb) {
  const u32 w4 = (u32)w & 7u;
  if (!(w4 & 1u)) { nb = 1; return 1u; }               // "0": value 1
  if (!(w4 & 4u)) { nb = 3; return 2u | ((w4 >> 1) & 1u); }   // "1 b 0": value 2 + b
  const u64 z = ~w & 0x5555555555555555ull;            // a 0 flag at an even position ends the code
  const u32 p = z ? (u32)__builtin_ctzll(z) : 64u;
  if (p > 48) { nb = 0xffffffffu; return 1; }
  nb = p + 1;
  const u32 k = p >> 1;
  u64 x = (w >> 1) & 0x5555555555555555ull & ((1ull << p) - 1);   // data bits, now at even positions
  x = (x | (x >> 1)) & 0x3333333333333333ull;
  x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
  x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
  x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
  x = (x | (x >> 16)) & 0xffffffffull;
  const u32 y = (u32)x;                                 // bit i = i-th data bit read
  return (1u << k) | (k ? __builtin_bitreverse32(y) >> (32 - k) : 0u);
}

__global__ __launch_bounds__(64) void lz77_decode_kernel(const LzDecDev* __restrict__ jobs) {
  const LzDecDev J = jobs[blockIdx.x];
  __shared__ u8 ring_mem[kRing];
  __shared__ __attribute__((aligned(8))) u8 sbuf_mem[1024];    // the two code-stream chunks of the window, for literal runs
  l_u8* const ring = (l_u8*)ring_mem;
  l_u8* const sbuf = (l_u8*)sbuf_mem;
  g_u8* const out = (g_u8*)J.out;
  const u32 lane = (u32)lane_id();
  g_cu8* in = (g_cu8*)J.in;
  const u32 n = J.n;
  const u64 nbits = (u64)n * 8;
  u64 bp = 0; u32 op = 0; int status = ZPQ_OK;
  // code-stream window: chunk c = bytes [512c, 512c+512), 8 per lane; bytes at or past n read as 0
  auto load_chunk = [&](u32 c) -> u64 { const u64 o = (u64)c * 512 + (u64)lane * 8; return o < n ? load8(in + o) : 0ull; };
  u32 chunk = 0;
  u64 cur = load_chunk(0), nxt = load_chunk(1);
  *(l_u64*)(sbuf + lane * 8) = cur;
  *(l_u64*)(sbuf + 512 + lane * 8) = nxt;
  __builtin_amdgcn_wave_barrier();
  auto peek = [&](u64 b) -> u64 {                       // 64 bits of the stream from bit b on (b < nbits)
    const u32 byte = (u32)(b >> 3);
    const u32 c = __builtin_amdgcn_readfirstlane(byte >> 9);
    if (c != chunk) {
      const bool step = c == chunk + 1;
      cur = step ? nxt : load_chunk(c);
      nxt = load_chunk(c + 1);
      chunk = c;
      __builtin_amdgcn_wave_barrier();
      if (!step) *(l_u64*)(sbuf + (c & 1u) * 512 + lane * 8) = cur;
      *(l_u64*)(sbuf + ((c + 1) & 1u) * 512 + lane * 8) = nxt;
      __builtin_amdgcn_wave_barrier();
    }
    const u32 idx = byte & 511u;
    const u32 L = __builtin_amdgcn_readfirstlane(idx >> 3);
    const u32 sh = ((idx & 7u) << 3) | (u32)(b & 7);
    const u64 lo = readlane64(cur, L);
    const u64 hi = L == 63 ? readlane64(nxt, 0) : readlane64(cur, L + 1);
    return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
  };
  for (;;) {
    if (bp + 2 > nbits) break;
    u64 w = peek(bp);
    const u32 mmv = (u32)(w & 3);
    u32 used = 2; w >>= 2;
    if (mmv == 0) {                                   // literal run: gamma length then bytes
      u32 nb;
      u32 len = gamma_decode(w, nb);
      if (nb == 0xffffffffu) { if (bp + used + 50 <= nbits) status = ZPQ_ERR_FORMAT; break; }
      used += nb;
      if (bp + used > nbits) break;                    // stream ends inside the length code
      bp += used;
      const u64 avail = (nbits - bp) >> 3;
      const bool cutoff = avail < len;
      if (cutoff) len = (u32)avail;                    // stream ends inside the run
      if ((u64)op + len > J.out_cap) { status = ZPQ_ERR_CAPACITY; break; }
      const u32 byte0 = (u32)(bp >> 3), sh = (u32)(bp & 7);
      if (byte0 >= chunk * 512u && (u64)byte0 + len + 1 <= (u64)chunk * 512u + 1024u) {
        // the whole run (and the byte after it, for the bit shift) sits in the two chunks held in LDS
        for (u32 j = lane; j < len; j += 64) {
          const u32 q = byte0 + j;
          const u32 two = (u32)sbuf[q & 1023u] | ((u32)sbuf[(q + 1) & 1023u] << 8);
          const u8 c = (u8)(two >> sh);
          out[op + j] = c;
          ring[(op + j) & (kRing - 1)] = c;
        }
      } else {
        for (u32 j = lane; j < len; j += 64) {
          const u32 two = (u32)in[byte0 + j] | (sh ? (u32)in[byte0 + j + 1] << 8 : 0u);   // byte-aligned run: nothing past its end
          const u8 c = (u8)(two >> sh);
          out[op + j] = c;
          ring[(op + j) & (kRing - 1)] = c;
        }
      }
      __builtin_amdgcn_wave_barrier();
      op += len; bp += 8ull * len;
      if (cutoff) break;
    } else {                                          // match
      if (bp + 5 > nbits) break;
      const u32 lo = (mmv - 1) * 8 + (u32)(w & 7); w >>= 3; used += 3;
      u32 nb;
      u32 len = gamma_decode(w, nb);
      if (nb == 0xffffffffu) { if (bp + used + 50 <= nbits) status = ZPQ_ERR_FORMAT; break; }
      w >>= nb; used += nb;
      if (bp + used + 2 > nbits) break;
      len = len * 4 + (u32)(w & 3); used += 2;         // <= 2^27
      w >>= 2;
      bp += used;                                      // used <= 2+3+49+2 = 56 bits
      if (bp + J.rb + lo > nbits) break;
      if (used + J.rb + lo > 64) w = peek(bp);         // (rare: the 64-bit window already holds the offset bits otherwise)
      const u32 r = (u32)(w & ((1ull << J.rb) - 1)); w >>= J.rb;
      const u32 qv = (u32)(w & ((1ull << lo) - 1)) | (1u << lo);
      bp += J.rb + lo;
      const u32 off = ((qv << J.rb) | r) - ((1u << J.rb) - 1u);
      if (off == 0 || off > op) { status = ZPQ_ERR_FORMAT; break; }
      if ((u64)op + len > J.out_cap) { status = ZPQ_ERR_CAPACITY; break; }
      const u32 src0 = op - off;
      if (off + 64 <= kRing) {                        // source inside the LDS ring
        for (u32 c0 = 0; c0 < len; c0 += 64) {
          const u32 j = c0 + lane;
          u8 c = 0;
          // off >= 64: out[op+j-off] was written before this chunk; off < 64: periodic extension
          if (j < len) c = ring[(off >= 64 ? op + j - off : src0 + (j % off)) & (kRing - 1)];
          __builtin_amdgcn_wave_barrier();
          if (j < len) { out[op + j] = c; ring[(op + j) & (kRing - 1)] = c; }
          __builtin_amdgcn_wave_barrier();
        }
      } else {                                        // far match: bytes written >= 64 KiB ago
        // pieces of at most 32 KiB (< off): every piece only reads what earlier pieces have stored.  The first piece
        // reads bytes at least 32 KiB behind the write frontier: hundreds of store instructions ago, long complete (at
        // most 63 memory operations are ever outstanding), so only later pieces of a self-overlapping match wait.
        // Loads bypass the L1 (a line cached while it was only partly written would be stale): L2 is where stores land.
        for (u32 c0 = 0; c0 < len; c0 += 32768u) {
          const u32 pl = len - c0 < 32768u ? len - c0 : 32768u;
          if (c0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          u32 j = lane;
          for (; j + 192 < pl; j += 256) {             // four loads in flight per lane
            g_cu8* sp = (g_cu8*)out + src0 + c0 + j;
            const u8 a0 = __builtin_nontemporal_load(sp), a1 = __builtin_nontemporal_load(sp + 64),
                     a2 = __builtin_nontemporal_load(sp + 128), a3 = __builtin_nontemporal_load(sp + 192);
            g_u8* dp = out + op + c0 + j;
            dp[0] = a0; dp[64] = a1; dp[128] = a2; dp[192] = a3;
            const u32 r0 = op + c0 + j;
            ring[r0 & (kRing - 1)] = a0; ring[(r0 + 64) & (kRing - 1)] = a1; ring[(r0 + 128) & (kRing - 1)] = a2; ring[(r0 + 192) & (kRing - 1)] = a3;
          }
          for (; j < pl; j += 64) {
            const u8 c = __builtin_nontemporal_load((g_cu8*)out + src0 + c0 + j);
            out[op + c0 + j] = c;
            ring[(op + c0 + j) & (kRing - 1)] = c;
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
      op += len;
    }
  }
  if (lane == 0) { J.result[0] = op; J.result[1] = (u32)status; }
}


// ---- byte-aligned codes (LZBuffer level 2, ZSFX/libzpaq.cpp:6221-6224, :6519-6547): what methods 3 and 4 put in front of
// their models; the inverse is the level-2 post-processor program of makeConfig.  00xxxxxx = x + 1 literals follow;
// yyxxxxxx (yy > 0) = match of x + minMatch bytes, yy + 1 offset bytes follow (offset - 1, most significant first).
// One wave per block: the parse is a few scalar instructions per token, literal runs (<= 64 bytes) and matches
// (<= minMatch + 63 bytes) are copied by the lanes -- through the 64 KiB LDS ring when the source is near, from HBM
// (past the L1, which may hold lines cached while they were only partly written) when it is far.  J.rb carries minMatch.
__global__ __launch_bounds__(64) void lz2_decode_kernel(const LzDecDev* __restrict__ jobs) {
  const LzDecDev J = jobs[blockIdx.x];
  __shared__ u8 ring_mem[kRing];
  l_u8* const ring = (l_u8*)ring_mem;
  g_u8* const out = (g_u8*)J.out;
  g_cu8* const in = (g_cu8*)J.in;
  const u32 lane = (u32)lane_id();
  const u32 n = J.n, mm = J.rb & 255u;
  u32 ip = 0, op = 0; int status = ZPQ_OK;
  // code-stream window: 64 x 8 bytes from a 8-byte aligned position; bytes at or past n read as 0
  u32 wbase = 0;
  u64 win = (u64)lane * 8 < n ? load8(in + (u64)lane * 8) : 0ull;
  auto peek5 = [&](u32 p) -> u64 {                     // the 8 bytes at p (p < n)
    if (p < wbase || p + 8 > wbase + 504) {
      wbase = p & ~7u;
      const u64 o = (u64)wbase + (u64)lane * 8;
      win = o < n ? load8(in + o) : 0ull;
    }
    const u32 L = __builtin_amdgcn_readfirstlane((p - wbase) >> 3), sh = ((p - wbase) & 7u) << 3;
    const u64 lo = readlane64(win, L), hi = readlane64(win, L + 1);
    return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
  };
  while (ip < n) {
    const u64 w = peek5(ip);
    const u32 code = (u32)w & 255u;
    if (code < 64) {                                  // literals
      u32 len = code + 1;
      const u32 avail = n - ip - 1;
      const bool cutoff = avail < len;
      if (cutoff) len = avail;                         // the stream ends inside the run
      if ((u64)op + len > J.out_cap) { status = ZPQ_ERR_CAPACITY; break; }
      if (lane < len) { const u8 c = in[ip + 1 + lane]; out[op + lane] = c; ring[(op + lane) & (kRing - 1)] = c; }
      __builtin_amdgcn_wave_barrier();
      op += len; ip += 1 + len;
      if (cutoff) break;
    } else {
      const u32 nb = (code >> 6) + 1;
      if (ip + 1 + nb > n) break;                      // the stream ends inside the code
      u32 o1 = 0;
      for (u32 k = 0; k < nb; ++k) o1 = o1 << 8 | ((u32)(w >> (8 * (k + 1))) & 255u);
      const u32 len = (code & 63u) + mm;
      ip += 1 + nb;
      if (o1 >= op) { status = ZPQ_ERR_FORMAT; break; }                  // offset = o1 + 1 must not reach before the block
      const u32 off = o1 + 1;
      if (len == 0) continue;
      if ((u64)op + len > J.out_cap) { status = ZPQ_ERR_CAPACITY; break; }
      const u32 src0 = op - off;
      if (off + 128 <= kRing) {                        // source inside the LDS ring
        for (u32 c0 = 0; c0 < len; c0 += 64) {
          const u32 j = c0 + lane;
          u8 c = 0;
          if (j < len) c = ring[(off >= 64 ? op + j - off : src0 + (j % off)) & (kRing - 1)];
          __builtin_amdgcn_wave_barrier();
          if (j < len) { out[op + j] = c; ring[(op + j) & (kRing - 1)] = c; }
          __builtin_amdgcn_wave_barrier();
        }
      } else {                                         // far: written at least 64 KiB - 128 bytes ago, long complete
        for (u32 j = lane; j < len; j += 64) {
          const u8 c = __builtin_nontemporal_load((g_cu8*)out + src0 + j);
          out[op + j] = c;
          ring[(op + j) & (kRing - 1)] = c;
        }
        __builtin_amdgcn_wave_barrier();
      }
      op += len;
    }
  }
  if (lane == 0) { J.result[0] = op; J.result[1] = (u32)status; }
}

// ---- parse off the chain ----------------------------------------------------------------------------------------------
// The wave above spends ~250 scalar instructions per token on PARSING; the copies are cheap.  The parse state between
// two tokens is one number (the bit position), so a parse started at any bit merges with the true one at their first
// common token start and follows it from there -- and on real streams a garbage parse lands on a true token start
// within a few dozen tokens.  Lanes parse 4 KiB segments of the code stream from 256 bytes before their segment
// (lzdec_spec_kernel: marks the token starts it visits inside its segment, records where it leaves), one lane per block
// follows the true chain through those marks (lzdec_stitch_kernel: first true token start of every segment), the lanes
// then parse their segment again from that start and write the token list (count, scan, emit), and one wave per block
// replays the list: literal runs and match copies, nothing else (lz77_copy_kernel).  Same results, same error behaviour
// (a bad code becomes an error token at its place in the list; capacity and offset checks stay with the copies).
struct LzParDev {
  const u8* in; u32 n; u32 rb;
  u8* out; u32 out_cap; u32* result;          // result[0] = out_len, result[1] = status
  u32 nseg, seg0;
  u32* visit;                                  // bit per stream bit: token start seen by the segment's lane
  u32* sexit; u32* entry; u32* cnt; u32* dst;  // per segment
  u64* tok; u32* ntok;                         // token list (len | kind << 31, offset or literal bit position)
};
constexpr u32 kSegBits = 32768, kWarmBits = 2048;   // 4 KiB segments (the stitcher pays per segment), 256 bytes of run-up
constexpr u32 kEnd = 0xffffffffu, kDead = 0xfffffffeu, kNone = 0xffffffffu;
constexpr u32 kTokErr = 0xffffffffu;

struct Tok { u32 kind; u32 len; u32 x; u32 next; };    // kind 0 literal (x = bit position of the bytes), 1 match (x = offset), 2 end, 3 bad code

__device__ __forceinline__ u64 peek_g(g_cu8* in, u32 n, u32 b) {      // 64 bits of the stream from bit b on; bytes past n read as 0
  const u32 byte = b >> 3, sh = b & 7;
  u64 lo; u32 hi;
  if (byte + 16 <= n) {                                   // two aligned dwords pairs around the position (no unaligned vector access)
    const u32 al = byte & ~3u, sh8 = (byte & 3u) * 8;
    const __attribute__((address_space(1))) const u32* q = (const __attribute__((address_space(1))) const u32*)(in + al);
    const u64 a0 = q[0], a1 = q[1], a2 = q[2];
    const u64 w01 = a0 | (a1 << 32);
    lo = sh8 ? (w01 >> sh8) | (a2 << (64 - sh8)) : w01;
    hi = (u32)((sh8 ? (a2 >> sh8) : a2) & 0xff);
  }
  else if (byte + 9 <= n) { lo = load8(in + byte); hi = in[byte + 8]; }
  else {
    lo = 0; hi = 0;
    for (u32 j = 0; j < 8; ++j) if (byte + j < n) lo |= (u64)in[byte + j] << (8 * j);
    if (byte + 8 < n) hi = in[byte + 8];
  }
  return sh ? (lo >> sh) | ((u64)hi << (64 - sh)) : lo;
}

// one token at bit position bp: exactly the parse of lz77_decode_kernel
__device__ __forceinline__ Tok parse_token(g_cu8* in, u32 n, u32 rb, u32 bp0) {
  Tok T; T.kind = 2; T.len = 0; T.x = 0; T.next = kEnd;
  const u64 nbits = (u64)n * 8;
  u64 bp = bp0;
  if (bp + 2 > nbits) return T;
  u64 w = peek_g(in, n, bp0);
  const u32 mmv = (u32)(w & 3);
  u32 used = 2; w >>= 2;
  if (mmv == 0) {
    u32 nb;
    u32 len = gamma_decode(w, nb);
    if (nb == 0xffffffffu) { if (bp + used + 50 <= nbits) T.kind = 3; return T; }
    used += nb;
    if (bp + used > nbits) return T;
    bp += used;
    const u64 avail = (nbits - bp) >> 3;
    const bool cutoff = avail < len;
    if (cutoff) len = (u32)avail;
    T.kind = 0; T.len = len; T.x = (u32)bp;
    T.next = cutoff ? kEnd : (u32)(bp + 8ull * len);
    return T;
  }
  if (bp + 5 > nbits) return T;
  const u32 lo = (mmv - 1) * 8 + (u32)(w & 7); w >>= 3; used += 3;
  u32 nb;
  u32 len = gamma_decode(w, nb);
  if (nb == 0xffffffffu) { if (bp + used + 50 <= nbits) T.kind = 3; return T; }
  w >>= nb; used += nb;
  if (bp + used + 2 > nbits) return T;
  len = len * 4 + (u32)(w & 3); used += 2;
  w >>= 2;
  bp += used;
  if (bp + rb + lo > nbits) return T;
  if (used + rb + lo > 64) w = peek_g(in, n, (u32)bp);
  const u32 r = (u32)(w & ((1ull << rb) - 1)); w >>= rb;
  const u32 qv = (u32)(w & ((1ull << lo) - 1)) | (1u << lo);
  bp += rb + lo;
  T.kind = 1; T.len = len; T.x = ((qv << rb) | r) - ((1u << rb) - 1u); T.next = (u32)bp;
  return T;
}

__global__ __launch_bounds__(256) void lzdec_fill_kernel(u32* __restrict__ p, u32 v, size_t words) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) p[i] = v;
}

// block that owns global segment g: blocks[] is ordered by seg0
__device__ __forceinline__ u32 block_of(const LzParDev* __restrict__ blocks, u32 nblocks, u32 g) {
  u32 lo = 0, hi = nblocks;              // last block with seg0 <= g
  while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (blocks[mid].seg0 <= g) lo = mid; else hi = mid; }
  return lo;
}

__global__ __launch_bounds__(256) void lzdec_spec_kernel(const LzParDev* __restrict__ blocks, u32 nblocks, u32 nseg_total) {
  const u32 g = blockIdx.x * 256u + threadIdx.x;
  if (g >= nseg_total) return;
  const LzParDev B = blocks[block_of(blocks, nblocks, g)];
  const u32 k = g - B.seg0;
  const u32 nbits = B.n * 8u;
  const u32 beg = k * kSegBits, end = nbits - beg < kSegBits ? nbits : beg + kSegBits;
  g_cu8* in = (g_cu8*)B.in;
  u32 p = k ? beg - kWarmBits : 0u;
  while (p < end) {
    if (p >= beg) atomicOr(B.visit + (p >> 5), 1u << (p & 31));
    const Tok T = parse_token(in, B.n, B.rb, p);
    if (T.kind == 3) { p = kDead; break; }
    p = T.next;                                   // kEnd after the last token
  }
  B.sexit[k] = p;
}

__global__ __launch_bounds__(64) void lzdec_stitch_kernel(const LzParDev* __restrict__ blocks) {
  if (threadIdx.x) return;
  const LzParDev B = blocks[blockIdx.x];
  g_cu8* in = (g_cu8*)B.in;
  const u32 nbits = B.n * 8u;
  u32 p = 0;
  while (p < nbits) {
    const u32 k = p / kSegBits;
    if (B.entry[k] == kNone) B.entry[k] = p;
    if ((B.visit[p >> 5] >> (p & 31)) & 1u) { p = B.sexit[k]; continue; }   // kEnd / kDead end the loop (both >= nbits)
    const Tok T = parse_token(in, B.n, B.rb, p);
    if (T.kind >= 2) break;
    p = T.next;
  }
}

// walks the true chain through segment k from its entry; F(token) for every token produced there
template <class F>
__device__ __forceinline__ void walk_segment(const LzParDev& B, u32 k, F f) {
  u32 p = ((__attribute__((address_space(1))) const u32*)B.entry)[k];
  if (p == kNone) return;
  const u32 nbits = B.n * 8u;
  const u32 beg = k * kSegBits, end = nbits - beg < kSegBits ? nbits : beg + kSegBits;
  g_cu8* in = (g_cu8*)B.in;
  while (p < end) {
    const Tok T = parse_token(in, B.n, B.rb, p);
    if (T.kind == 2) break;
    f(T);
    if (T.kind == 3) break;
    p = T.next;
  }
}

__global__ __launch_bounds__(256) void lzdec_count_kernel(const LzParDev* __restrict__ blocks, u32 nblocks, u32 nseg_total) {
  const u32 g = blockIdx.x * 256u + threadIdx.x;
  if (g >= nseg_total) return;
  const LzParDev B = blocks[block_of(blocks, nblocks, g)];
  u32 c = 0;
  walk_segment(B, g - B.seg0, [&](const Tok& T) { c += (T.kind == 3 || T.len) ? 1u : 0u; });
  B.cnt[g - B.seg0] = c;
}

__global__ __launch_bounds__(1024) void lzdec_scan_kernel(const LzParDev* __restrict__ blocks) {
  const LzParDev B = blocks[blockIdx.x];
  __shared__ u32 wsum[16];
  __shared__ u32 carry_s;
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (u32 k0 = 0; k0 < B.nseg; k0 += 1024) {
    const u32 k = k0 + tid;
    const u32 cnt = k < B.nseg ? B.cnt[k] : 0u;
    u32 x = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 y = __shfl_up(x, d); if (lane >= (u32)d) x += y; }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    u32 wbase = 0;
    for (u32 w = 0; w < wave; ++w) wbase += wsum[w];
    const u32 carry = carry_s;
    if (k < B.nseg) B.dst[k] = carry + wbase + x - cnt;
    __syncthreads();
    if (tid == 1023) carry_s = carry + wbase + x;
    __syncthreads();
  }
  if (tid == 0) *B.ntok = carry_s;
}

__global__ __launch_bounds__(256) void lzdec_emit_kernel(const LzParDev* __restrict__ blocks, u32 nblocks, u32 nseg_total) {
  const u32 g = blockIdx.x * 256u + threadIdx.x;
  if (g >= nseg_total) return;
  const LzParDev B = blocks[block_of(blocks, nblocks, g)];
  typedef __attribute__((address_space(1))) u64 g_u64;
  typedef __attribute__((address_space(1))) const u32 g_cu32;
  g_u64* const tok = (g_u64*)B.tok;
  u32 o = ((g_cu32*)B.dst)[g - B.seg0];                    // index, not a running pointer
  walk_segment(B, g - B.seg0, [&](const Tok& T) {
    if (T.kind == 3) { tok[o] = (u64)kTokErr; ++o; }
    else if (T.len) { tok[o] = (u64)(T.len | (T.kind == 0 ? 0x80000000u : 0u)) | ((u64)T.x << 32); ++o; }
  });
}

// Replays a token list.  64 tokens per step, one per lane: output positions by a wave prefix sum, then
//  * short tokens (<= 32 bytes) whose source lies entirely before the step's output are copied by their own lane, all at
//    once -- from the code stream (literals), from the LDS ring holding the last 64 KiB of output, or from HBM when the
//    source is further back;
//  * short matches that read what this step writes (including self-overlapping ones) follow in order, one lane at a time;
//  * a long token is copied by the whole wave (the loops of lz77_decode_kernel) and splits the step, so that the ring
//    always receives the output in order.
// Offsets and capacity are checked per token in order: the first offending token ends the replay with its status and
// nothing of it is written.
constexpr u32 kShort = 32;

// up to 32 source bytes into registers -- every load is issued before the first one is waited for -- then out and ring
struct Bytes32 { u64 v[4]; };
typedef __attribute__((address_space(1))) u64_u g_u64_u;
__device__ __forceinline__ Bytes32 gather_lit(g_cu8* in, u32 n, u32 bit, u32 len) {
  Bytes32 r;
#pragma unroll
  for (u32 k = 0; k < 4; ++k) r.v[k] = 8 * k < len ? peek_g(in, n, bit + 64 * k) : 0ull;
  return r;
}
__device__ __forceinline__ Bytes32 gather_far(g_cu8* src, u32 len) {      // src .. src+len is at least 60 KiB behind the write frontier:
  Bytes32 r;                                                               // the up to 7 bytes read past src+len are inside the buffer too
#pragma unroll
  for (u32 k = 0; k < 4; ++k) r.v[k] = 8 * k < len ? __builtin_nontemporal_load((g_cu64_u*)(src + 8 * k)) : 0ull;
  return r;
}
// The ring is accessed 8 / 4 / 2 / 1 bytes at a time at any byte address (gfx950 has unaligned LDS access: one ds_read_b64 /
// ds_write_b64 each); only an access that would run over the end of the ring goes byte by byte.
typedef __attribute__((address_space(3))) u64_u l_u64_u;
typedef __attribute__((address_space(3))) u32_u l_u32_u;
typedef u16 __attribute__((aligned(1))) u16_u;
typedef __attribute__((address_space(3))) u16_u l_u16_u;
typedef __attribute__((address_space(1))) u32_u g_u32_u;
typedef __attribute__((address_space(1))) u16_u g_u16_u;
__device__ __forceinline__ u64 ring_read8(l_u8* ring, u32 at) {
  const u32 i = at & (kRing - 1);
  if (i <= kRing - 8) return *(l_u64_u*)(ring + i);
  u64 w = 0;
  for (u32 k = 0; k < 8; ++k) w |= (u64)ring[(i + k) & (kRing - 1)] << (8 * k);
  return w;
}
__device__ __forceinline__ Bytes32 gather_ring(l_u8* ring, u32 src, u32 len) {
  Bytes32 r;
#pragma unroll
  for (u32 k = 0; k < 4; ++k) r.v[k] = 8 * k < len ? ring_read8(ring, src + 8 * k) : 0ull;    // (bytes past len are read and ignored)
  return r;
}
// m (1..8) bytes of w to out[pos..] and to the ring
__device__ __forceinline__ void put_bytes(g_u8* out, l_u8* ring, u32 pos, u64 w, u32 m) {
  const u32 i = pos & (kRing - 1);
  if (i > kRing - 8) {                                    // over the end of the ring: byte by byte (1 in 8192)
    for (u32 k = 0; k < m; ++k) { const u8 c = (u8)(w >> (8 * k)); out[pos + k] = c; ring[(i + k) & (kRing - 1)] = c; }
    return;
  }
  if (m == 8) { *(g_u64_u*)(out + pos) = w; *(l_u64_u*)(ring + i) = w; return; }
  u32 o = 0;
  if (m & 4) { *(g_u32_u*)(out + pos) = (u32)w; *(l_u32_u*)(ring + i) = (u32)w; o = 4; w >>= 32; }
  if (m & 2) { *(g_u16_u*)(out + pos + o) = (u16)w; *(l_u16_u*)(ring + i + o) = (u16)w; o += 2; w >>= 16; }
  if (m & 1) { out[pos + o] = (u8)w; ring[i + o] = (u8)w; }
}
__device__ __forceinline__ void scatter(g_u8* out, l_u8* ring, u32 pos, u32 len, const Bytes32& r) {
#pragma unroll
  for (u32 k = 0; k < 4; ++k) {
    if (8 * k >= len) break;
    put_bytes(out, ring, pos + 8 * k, r.v[k], len - 8 * k < 8 ? len - 8 * k : 8);
  }
}
__device__ __forceinline__ void copy_ring_bytewise(g_u8* out, l_u8* ring, u32 off, u32 len, u32 op) {     // self-overlapping: in order
  for (u32 j = 0; j < len; ++j) { const u8 c = ring[(op + j - off) & (kRing - 1)]; out[op + j] = c; ring[(op + j) & (kRing - 1)] = c; }
}

__global__ __launch_bounds__(64) void lz77_copy_kernel(const LzParDev* __restrict__ blocks) {
  const LzParDev J = blocks[blockIdx.x];
  __shared__ u8 ring_mem[kRing];
  l_u8* const ring = (l_u8*)ring_mem;
  g_u8* const out = (g_u8*)J.out;
  g_cu8* const in = (g_cu8*)J.in;
  const u32 lane = (u32)lane_id();
  const u32 ntok = *J.ntok;
  u64 op_base = 0; int status = ZPQ_OK;
  // Software pipeline.  A step would otherwise open with dependent memory round trips (tokens -> stream bytes of the
  // literals / far match sources).  Tokens are loaded two steps ahead; one step ahead their lengths are scanned into
  // output positions and every short token's source that is not in the LDS ring is fetched into registers: stream
  // bytes for literals, HBM bytes for matches reaching further back than the ring -- provided that source was complete
  // when the CURRENT step began (it lies before this step's output), which the drain at the top of each step makes true.
  auto load_tok = [&](u32 t) -> u64 { return t + lane < ntok ? J.tok[t + lane] : 0ull; };
  struct Step { u64 tk; u64 incl; Bytes32 pre; bool have; };      // `have`: pre holds this lane's source bytes
  auto scan_len = [&](u64 tkn, bool valid_) -> u64 {
    const u32 a_ = (u32)tkn;
    u64 incl_ = (valid_ && a_ != kTokErr) ? (a_ & 0x7fffffffu) : 0u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u64 y = __shfl_up((unsigned long long)incl_, d); if (lane >= (u32)d) incl_ += y; }
    return incl_;
  };
  // sources of the step whose tokens are tkn, given where its output starts and what is known to be stored (< done)
  auto prefetch = [&](u64 tkn, u64 incl_, u64 base, u64 done, bool valid_, Bytes32& pre, bool& have) {
    const u32 a_ = (u32)tkn, x_ = (u32)(tkn >> 32), l_ = a_ & 0x7fffffffu;
    pre.v[0] = pre.v[1] = pre.v[2] = pre.v[3] = 0; have = false;
    if (!valid_ || a_ == kTokErr || l_ == 0 || l_ > kShort) return;
    const u64 p_ = base + incl_ - l_;
    if (a_ >> 31) { pre = gather_lit(in, J.n, x_, l_); have = true; return; }
    if (x_ > kRing - 4096u && (u64)x_ <= p_ && p_ - x_ + l_ <= done && p_ + l_ <= (u64)J.out_cap) {
      pre = gather_far((g_cu8*)out + (p_ - x_), l_); have = true;
    }
  };
  Step A, B;
  A.tk = load_tok(0); B.tk = load_tok(64);
  A.incl = scan_len(A.tk, lane < ntok);
  prefetch(A.tk, A.incl, 0, 0, lane < ntok, A.pre, A.have);
  for (u32 t0 = 0; t0 < ntok && status == ZPQ_OK; t0 += 64) {
    const u32 cnt = ntok - t0 < 64 ? ntok - t0 : 64;
    const bool valid = lane < cnt;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // everything earlier steps stored is in memory (and what they prefetched has arrived)
    const u64 tkC = load_tok(t0 + 128);                   // two steps ahead
    const u64 totalA = (u64)__builtin_amdgcn_readlane((u32)A.incl, 63) | ((u64)__builtin_amdgcn_readlane((u32)(A.incl >> 32), 63) << 32);
    const bool validB = t0 + 64 + lane < ntok;
    B.incl = scan_len(B.tk, validB);
    prefetch(B.tk, B.incl, op_base + totalA, op_base, validB, B.pre, B.have);    // (if this step ends early nothing of B is used)
    const u64 tk = A.tk, incl = A.incl;
    const Bytes32 pre = A.pre;
    const bool have = A.have;
    A = B; B.tk = tkC;
    const u32 a = (u32)tk, x = (u32)(tk >> 32);
    const bool is_err = valid && a == kTokErr;
    const bool is_lit = !is_err && (a >> 31);
    const u32 len = (valid && !is_err) ? (a & 0x7fffffffu) : 0u;
    const u64 pos64 = op_base + incl - len;
    const bool bad_fmt = is_err || (valid && !is_lit && (x == 0 || (u64)x > pos64));
    const bool bad_cap = valid && pos64 + len > (u64)J.out_cap;
    const u64 errmask = __ballot(bad_fmt || bad_cap);
    const u32 nproc = errmask ? (u32)__builtin_ctzll(errmask) : cnt;
    if (errmask) {
      const bool f = __builtin_amdgcn_readlane((u32)bad_fmt, nproc) != 0;
      status = f ? ZPQ_ERR_FORMAT : ZPQ_ERR_CAPACITY;
    }
    const u32 pos = (u32)pos64;                           // lanes below nproc are within the capacity: 32 bits
    const bool act = lane < nproc;
    const u64 longmask = __ballot(act && len > kShort);
    u32 cur = 0;
    while (cur < nproc) {
      const u64 lm = longmask & ~((1ull << cur) - 1ull);
      const u32 L = lm ? (u32)__builtin_ctzll(lm) : nproc;
      if (L > cur) {                                      // short tokens [cur, L)
        // rounds: a token is ready when it is a literal, when its source ends before the output of the first token
        // still pending, or when it is that first token itself (only then may it read its own output)
        u64 pend = __ballot(act && lane >= cur && lane < L);
        while (pend) {
          const u32 f = (u32)__builtin_ctzll(pend);
          const u32 pos_f = __builtin_amdgcn_readlane(pos, f);
          const bool mine = (pend >> lane) & 1ull;
          const u32 src = pos - x;                          // (matches only)
          const bool ready = mine && (is_lit || lane == f || src + len <= pos_f);
          const bool far = ready && !is_lit && x > kRing - 4096u;
          if (__ballot(far && !have)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (rare: a source inside this or the previous step's long token)
          if (ready) {
            if (have) scatter(out, ring, pos, len, pre);           // literal bytes / far source, fetched a step ago
            else if (is_lit) scatter(out, ring, pos, len, gather_lit(in, J.n, x, len));
            else if (x < len) copy_ring_bytewise(out, ring, x, len, pos);
            else if (far) scatter(out, ring, pos, len, gather_far((g_cu8*)out + src, len));
            else scatter(out, ring, pos, len, gather_ring(ring, src, len));
          }
          __builtin_amdgcn_wave_barrier();
          pend &= ~__ballot(ready);
        }
      }
      if (L < nproc) {                                    // one long token, whole wave
        const u32 tl = __builtin_amdgcn_readlane(len, L), tx = __builtin_amdgcn_readlane(x, L), op = __builtin_amdgcn_readlane(pos, L);
        const bool lit = (__builtin_amdgcn_readlane(a, L) >> 31) != 0;
        if (lit) {
          const u32 byte0 = tx >> 3, sh = tx & 7;
          for (u32 j = lane; j < tl; j += 64) {
            const u32 two = (u32)in[byte0 + j] | (sh ? (u32)in[byte0 + j + 1] << 8 : 0u);
            const u8 c = (u8)(two >> sh);
            out[op + j] = c;
            ring[(op + j) & (kRing - 1)] = c;
          }
          __builtin_amdgcn_wave_barrier();
        } else {
          const u32 off = tx, src0 = op - off;
          if (off + 64 <= kRing) {
            for (u32 c0 = 0; c0 < tl; c0 += 64) {
              const u32 j = c0 + lane;
              u8 c = 0;
              if (j < tl) c = ring[(off >= 64 ? op + j - off : src0 + (j % off)) & (kRing - 1)];
              __builtin_amdgcn_wave_barrier();
              if (j < tl) { out[op + j] = c; ring[(op + j) & (kRing - 1)] = c; }
              __builtin_amdgcn_wave_barrier();
            }
          } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (u32 c0 = 0; c0 < tl; c0 += 32768u) {     // see lz77_decode_kernel: pieces that only read what is stored
              const u32 pl = tl - c0 < 32768u ? tl - c0 : 32768u;
              if (c0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              u32 j = lane;
              for (; j + 192 < pl; j += 256) {
                g_cu8* sp = (g_cu8*)out + src0 + c0 + j;
                const u8 a0 = __builtin_nontemporal_load(sp), a1 = __builtin_nontemporal_load(sp + 64),
                         a2 = __builtin_nontemporal_load(sp + 128), a3 = __builtin_nontemporal_load(sp + 192);
                g_u8* dp = out + op + c0 + j;
                dp[0] = a0; dp[64] = a1; dp[128] = a2; dp[192] = a3;
                const u32 r0 = op + c0 + j;
                ring[r0 & (kRing - 1)] = a0; ring[(r0 + 64) & (kRing - 1)] = a1; ring[(r0 + 128) & (kRing - 1)] = a2; ring[(r0 + 192) & (kRing - 1)] = a3;
              }
              for (; j < pl; j += 64) {
                const u8 c = __builtin_nontemporal_load((g_cu8*)out + src0 + c0 + j);
                out[op + c0 + j] = c;
                ring[(op + c0 + j) & (kRing - 1)] = c;
              }
              __builtin_amdgcn_wave_barrier();
            }
          }
        }
      }
      cur = L + 1;
    }
    if (nproc) op_base += (u64)__builtin_amdgcn_readlane((u32)incl, nproc - 1) | ((u64)__builtin_amdgcn_readlane((u32)(incl >> 32), nproc - 1) << 32);
  }
  if (lane == 0) { J.result[0] = (u32)op_base; J.result[1] = (u32)status; }
}

}  // namespace

// Launch only (no host round trip): results land in result[0] = out_len, result[1] = status of every job.  h_jobs is the
// host copy of d_jobs (sizes for the scratch layout).
static int decode_group(zpq_ctx* ctx, hipStream_t st, const zpq_lzdec_dev* h_jobs, const zpq_lzdec_dev* d_jobs, size_t njobs);

int zpq_lz77_decode_launch(zpq_ctx* ctx, hipStream_t st, const zpq_lzdec_dev* h_jobs, const zpq_lzdec_dev* d_jobs, size_t njobs) {
  // Jobs are decoded in groups of at most 2^15 stream segments (128 MiB of code stream) per set of launches: bounded
  // scratch per group, and a margin around a code-generation problem met here -- lzdec_emit_kernel first wrote its tokens
  // through a running pointer captured by the walk's lambda ("*o++ = ..."); with more than 2^16 lanes in one launch the
  // lists of the lanes beyond 2^16 then held wrong tokens (the lanes computed the right ones: a hashing copy of the kernel
  // agreed with a host parse, every slot had one writer).  Indexed stores ("tok[o] = ...; ++o") are exact at any size
  // (tests/lzdec_big_roundtrip.py runs 79 000 lanes; ZPQ_LZDEC_GROUP lifts the grouping).
  size_t maxseg = 32768;
  if (const char* e = getenv("ZPQ_LZDEC_GROUP")) maxseg = (size_t)strtoull(e, 0, 10);
  size_t lo = 0;
  while (lo < njobs) {
    size_t hi = lo, segs = 0;
    while (hi < njobs) {
      const size_t ns = ((size_t)h_jobs[hi].n * 8 + kSegBits - 1) / kSegBits + 1;
      if (hi > lo && segs + ns > maxseg) break;
      segs += ns; ++hi;
    }
    int rc = decode_group(ctx, st, h_jobs + lo, d_jobs + lo, hi - lo);
    if (rc) return rc;
    lo = hi;
  }
  return ZPQ_OK;
}

static int decode_group(zpq_ctx* ctx, hipStream_t st, const zpq_lzdec_dev* h_jobs, const zpq_lzdec_dev* d_jobs, size_t njobs) {
  bool par = true;
  if (const char* e = getenv("ZPQ_LZDEC_SERIAL")) par = atoi(e) == 0;
  size_t nseg_total = 0, bytes = 0;
  for (size_t i = 0; i < njobs && par; ++i) {
    if (h_jobs[i].n >= (1u << 29)) par = false;          // bit positions are 32-bit in the token path
    const size_t nseg = ((size_t)h_jobs[i].n * 8 + kSegBits - 1) / kSegBits + 1;
    nseg_total += nseg;
    bytes += (((size_t)h_jobs[i].n + 8 + 255) & ~(size_t)255) + nseg * 16 + 256 + (((size_t)h_jobs[i].n + 2) * 8 + 255 & ~(size_t)255);
  }
  u8* work = nullptr; u8* meta = nullptr;
  if (par) {
    work = (u8*)zpq_scratch(ctx, 25, bytes + 4096);
    meta = (u8*)zpq_scratch(ctx, 26, njobs * (sizeof(LzParDev) + 16) + 4096);
    if (!work || !meta) par = false;                       // not enough memory for the token lists: the one-wave decoder needs none
  }
  if (!par) {
    ZPQ_LAUNCH(ctx, "lz77_decode_kernel", st, lz77_decode_kernel, dim3((unsigned)njobs), dim3(64), d_jobs);
    ZPQ_HIP(ctx, hipGetLastError());
    return ZPQ_OK;
  }
  std::vector<LzParDev> hb(njobs);
  LzParDev* d_blocks = (LzParDev*)meta;
  u32* d_ntok = (u32*)(meta + ((njobs * sizeof(LzParDev) + 255) & ~(size_t)255));
  u8* p = work;
  u32 seg0 = 0;
  for (size_t i = 0; i < njobs; ++i) {
    LzParDev& B = hb[i];
    const zpq_lzdec_dev& z = h_jobs[i];
    const size_t nseg = ((size_t)z.n * 8 + kSegBits - 1) / kSegBits + 1;
    B.in = z.in; B.n = z.n; B.rb = z.rb; B.out = z.out; B.out_cap = z.out_cap; B.result = z.result;
    B.nseg = (u32)nseg; B.seg0 = seg0;
    const size_t vbytes = ((size_t)z.n + 8 + 255) & ~(size_t)255;
    B.visit = (u32*)p; p += vbytes;
    B.sexit = (u32*)p; B.entry = B.sexit + nseg; B.cnt = B.entry + nseg; B.dst = B.cnt + nseg; p += (nseg * 16 + 255) & ~(size_t)255;
    B.tok = (u64*)p; p += (((size_t)z.n + 2) * 8 + 255) & ~(size_t)255;
    B.ntok = d_ntok + i;
    hipLaunchKernelGGL(lzdec_fill_kernel, dim3(1024), dim3(256), 0, st, B.visit, 0u, vbytes / 4);
    hipLaunchKernelGGL(lzdec_fill_kernel, dim3(64), dim3(256), 0, st, B.sexit, 0xffffffffu, nseg * 2);     // sexit = end, entry = none
    seg0 += (u32)nseg;
  }
  ZPQ_HIP(ctx, hipMemcpyAsync(d_blocks, hb.data(), njobs * sizeof(LzParDev), hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));                  // hb is a stack-lifetime host buffer
  const dim3 gs((unsigned)((nseg_total + 255) / 256)), blk(64), blk4(256);
  ZPQ_LAUNCH(ctx, "lzdec_spec_kernel", st, lzdec_spec_kernel, gs, blk4, d_blocks, (u32)njobs, (u32)nseg_total);
  ZPQ_LAUNCH(ctx, "lzdec_stitch_kernel", st, lzdec_stitch_kernel, dim3((unsigned)njobs), blk, d_blocks);
  ZPQ_LAUNCH(ctx, "lzdec_count_kernel", st, lzdec_count_kernel, gs, blk4, d_blocks, (u32)njobs, (u32)nseg_total);
  ZPQ_LAUNCH(ctx, "lzdec_scan_kernel", st, lzdec_scan_kernel, dim3((unsigned)njobs), dim3(1024), d_blocks);
  ZPQ_LAUNCH(ctx, "lzdec_emit_kernel", st, lzdec_emit_kernel, gs, blk4, d_blocks, (u32)njobs, (u32)nseg_total);
  ZPQ_LAUNCH(ctx, "lz77_copy_kernel", st, lz77_copy_kernel, dim3((unsigned)njobs), blk, d_blocks);
  ZPQ_HIP(ctx, hipGetLastError());
  return ZPQ_OK;
}

extern "C" int zpq_lz77_decode_dev(zpq_ctx* ctx, zpq_lz77_dec_job* jobs, size_t njobs) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (njobs == 0) return ZPQ_OK;
  hipStream_t st = ctx->stream;
  u8* d_meta = (u8*)zpq_scratch(ctx, 2, njobs * (sizeof(LzDecDev) + 8) + 64);
  if (!d_meta) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "lz77 scratch");
  LzDecDev* d_jobs = (LzDecDev*)d_meta;
  u32* d_res = (u32*)(d_meta + njobs * sizeof(LzDecDev));
  std::vector<LzDecDev> h(njobs);
  // rb with bit 31 set: byte-aligned codes (level 2), minimum match length in the low byte; those records go last
  size_t n1 = 0;
  for (int pass = 0; pass < 2; ++pass)
    for (size_t i = 0, k = pass ? n1 : 0; i < njobs; ++i) {
      const bool l2 = (jobs[i].rb >> 31) != 0;
      if (l2 != (pass == 1)) continue;
      if (!l2 && jobs[i].rb > 8) return zpq_fail(ctx, ZPQ_ERR_ARG, "rb out of range");
      h[k].in = jobs[i].d_in; h[k].n = jobs[i].n; h[k].rb = jobs[i].rb & 0x7fffffffu;
      h[k].out = jobs[i].d_out; h[k].out_cap = jobs[i].out_cap; h[k].result = d_res + 2 * i;
      ++k;
      if (!pass) n1 = k;
    }
  ZPQ_HIP(ctx, hipMemcpyAsync(d_jobs, h.data(), njobs * sizeof(LzDecDev), hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  int rc = n1 ? zpq_lz77_decode_launch(ctx, st, h.data(), d_jobs, n1) : ZPQ_OK;
  if (rc) return rc;
  if (njobs > n1) {
    ZPQ_LAUNCH(ctx, "lz2_decode_kernel", st, lz2_decode_kernel, dim3((unsigned)(njobs - n1)), dim3(64), d_jobs + n1);
    ZPQ_HIP(ctx, hipGetLastError());
  }
  std::vector<u32> res(njobs * 2);
  ZPQ_HIP(ctx, hipMemcpyAsync(res.data(), d_res, njobs * 8, hipMemcpyDeviceToHost, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  for (size_t i = 0; i < njobs; ++i) { jobs[i].out_len = res[2 * i]; jobs[i].status = (int32_t)res[2 * i + 1]; }
  return ZPQ_OK;
}
