// SHA-1 / SHA-256 over many extents: one Merkle-Damgard chain per LANE, persistent lanes that pull
// the next extent from a device counter when theirs is finished (fragments are 4 KiB..508 KiB, so a
// static extent->lane map would idle most of a wave behind its longest fragment).
//
// Replaces libzpaq::SHA1 / SHA256 (reference ZSFX/libzpaq.h:934-979, ZSFX/libzpaq.cpp:96-304) as used
// for fragment ids (ZSFX/zsfx.cpp:1463-1500, 1811-1834), segment checksums (ZSFX/libzpaq.cpp:2338-2366)
// and file verification.  Integer-only (u32 adds, rotates, boolean functions): no MFMA applies.
// Bound: VALU issue (~14 ops per input byte); traffic = each input byte read once.
#include <algorithm>
#include <stdlib.h>

#include <atomic>

#include "zpq_internal.h"

namespace {

struct Sha1State { u32 a, b, c, d, e; };

// gfx950 three-operand integer ops; hipcc does not form them reliably from C, and the SHA round
// count is what bounds these kernels (VALU issue), so they are spelled out.
__device__ __forceinline__ u32 add3(u32 a, u32 b, u32 c) { u32 r; asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// gfx950 has no v_xor3_b32 but it has the generic three-input LUT op: one instruction for a^b^c and
// for the majority function (both symmetric, so the operand order cannot matter).
__device__ __forceinline__ u32 xor3(u32 a, u32 b, u32 c) { u32 r; asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ u32 maj3(u32 a, u32 b, u32 c) { u32 r; asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe8" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// (x & m) | (y & ~m)
__device__ __forceinline__ u32 bfi(u32 m, u32 x, u32 y) { u32 r; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(x), "v"(y)); return r; }

#define SHA1_W(t) (w[(t) & 15] = rotl32(xor3(w[((t) + 13) & 15], w[((t) + 8) & 15], w[((t) + 2) & 15]) ^ w[(t) & 15], 1))
#define SHA1_R(f, k, wt)                                          \
  {                                                               \
    const u32 t1 = add3(e, (k), (wt));                            \
    const u32 tmp = add3(rotl32(a, 5), (f), t1);                  \
    e = d; d = c; c = rotl32(b, 30); b = a; a = tmp;              \
  }
#define SHA1_80_ROUNDS(WT0, WT)                                                       \
  _Pragma("unroll") for (int t = 0; t < 16; ++t) SHA1_R(bfi(b, c, d), K0, WT0(t))     \
  _Pragma("unroll") for (int t = 16; t < 20; ++t) SHA1_R(bfi(b, c, d), K0, WT(t))     \
  _Pragma("unroll") for (int t = 20; t < 40; ++t) SHA1_R(xor3(b, c, d), K1, WT(t))    \
  _Pragma("unroll") for (int t = 40; t < 60; ++t) SHA1_R(maj3(b, c, d), K2, WT(t)) \
  _Pragma("unroll") for (int t = 60; t < 80; ++t) SHA1_R(xor3(b, c, d), K3, WT(t))

__device__ __forceinline__ void sha1_rounds(u32 (&w)[16], Sha1State& s) {
  u32 a = s.a, b = s.b, c = s.c, d = s.d, e = s.e;
  // round constants in VGPRs: v_add3_u32 takes one literal/SGPR at most, registers are cheaper
  u32 K0 = 0x5A827999u, K1 = 0x6ED9EBA1u, K2 = 0x8F1BBCDCu, K3 = 0xCA62C1D6u;
  asm volatile("" : "+v"(K0), "+v"(K1), "+v"(K2), "+v"(K3));
#define WT0(t) w[t]
  SHA1_80_ROUNDS(WT0, SHA1_W)
#undef WT0
  s.a += a; s.b += b; s.c += c; s.d += d; s.e += e;
}

// Builds the (at most two) closing blocks of a message: `rem` (<64) trailing data bytes at p, the
// 0x80 marker unless already emitted, zero fill, and the 64-bit big-endian bit count when it fits.
// Returns true when this block is the last one of the message.
__device__ __forceinline__ bool tail_block(u32 (&w)[16], const u8* p, u32 rem, bool& marker_done, u64 total_bytes) {
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    u32 x = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      u32 j = 4 * t + b, v = 0;
      if (j < rem) v = p[j];
      else if (j == rem && !marker_done) v = 0x80;
      x = (x << 8) | v;
    }
    w[t] = x;
  }
  bool fits = marker_done ? true : rem <= 55;
  marker_done = true;
  if (fits) {
    u64 bits = total_bytes * 8;
    w[14] = (u32)(bits >> 32);
    w[15] = (u32)bits;
  }
  return fits;
}

__global__ __launch_bounds__(256) void sha1_extents_kernel(const u8* __restrict__ base, const u64* __restrict__ off,
                                                            const u32* __restrict__ len, u32 n, u8* __restrict__ digests,
                                                            u32* __restrict__ counter, const u32* __restrict__ order) {
  // One flat loop: a lane that finishes its extent immediately pulls the next one while its
  // neighbours keep hashing (no per-extent inner loop to re-converge on).  The next 64 input bytes
  // are always in flight while the 80 rounds of the current block run (a 16 MiB block checksum is
  // ONE chain on ONE lane: without the prefetch every block pays a full HBM round trip).
  bool have = false, marker = false;
  u32 idx = 0;
  const u8* p = nullptr;
  u64 total = 0, rem = 0;
  u32 pre = 0;                       // whole 64-byte blocks already in registers (0..2)
  Sha1State s = {0, 0, 0, 0, 0};
  u32x4 n0 = {0, 0, 0, 0}, n1 = n0, n2 = n0, n3 = n0, m0 = n0, m1 = n0, m2 = n0, m3 = n0;
  for (;;) {
    if (!have) {
      idx = atomicAdd(counter, 1u);
      if (idx >= n) return;
      if (order) idx = order[idx];      // longest extents first (see extent_order)
      p = base + off[idx];
      total = rem = len[idx];
      s = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};
      marker = false;
      have = true;
      pre = 0;
    }
    u32 w[16];
    bool last = false;
    if (rem >= 64) {
      // two blocks (one 128-byte line) ahead: n = next block, m = the one after
      if (pre == 0) {
        const u32x4_u* q = (const u32x4_u*)p; n0 = q[0]; n1 = q[1]; n2 = q[2]; n3 = q[3]; pre = 1;
      }
      const u32x4 v0 = n0, v1 = n1, v2 = n2, v3 = n3;
      p += 64; rem -= 64;
      if (pre == 2) { n0 = m0; n1 = m1; n2 = m2; n3 = m3; pre = 1; } else pre = 0;
      if (pre == 0 && rem >= 64) {
        const u32x4_u* q = (const u32x4_u*)p; n0 = q[0]; n1 = q[1]; n2 = q[2]; n3 = q[3]; pre = 1;
      }
      if (pre == 1 && rem >= 128) {
        const u32x4_u* q = (const u32x4_u*)(p + 64); m0 = q[0]; m1 = q[1]; m2 = q[2]; m3 = q[3]; pre = 2;
      }
      w[0] = bswap32(v0.x); w[1] = bswap32(v0.y); w[2] = bswap32(v0.z); w[3] = bswap32(v0.w);
      w[4] = bswap32(v1.x); w[5] = bswap32(v1.y); w[6] = bswap32(v1.z); w[7] = bswap32(v1.w);
      w[8] = bswap32(v2.x); w[9] = bswap32(v2.y); w[10] = bswap32(v2.z); w[11] = bswap32(v2.w);
      w[12] = bswap32(v3.x); w[13] = bswap32(v3.y); w[14] = bswap32(v3.z); w[15] = bswap32(v3.w);
    } else {
      last = tail_block(w, p, (u32)rem, marker, total);
      p += rem; rem = 0;
    }
    sha1_rounds(w, s);
    if (last) {
      u32* o = (u32*)(digests + (size_t)idx * 20);  // 20*idx is 4-byte aligned
      o[0] = bswap32(s.a); o[1] = bswap32(s.b); o[2] = bswap32(s.c); o[3] = bswap32(s.d); o[4] = bswap32(s.e);
      have = false;
    }
  }
}

// ---- the same, input staged through LDS -------------------------------------------------------------------------
// Lane-per-extent makes every load instruction of a wave touch 64 unrelated places (64 pages, 64 cache lines, 16 bytes
// of each): the address translation, not the data, is what the memory side is busy with (UTCL2 busy 88 % of the
// kernel, profiles/r02_pmc_sq.json).  Here the WAVE fetches for its lanes: a load instruction reads the next 128 bytes
// of EIGHT lanes' streams (8 lanes x 16 bytes each: whole cache lines, 8 pages per instruction instead of 64, an
// eighth of the instructions) and parks them in LDS; a lane then takes its 64-byte blocks from its own LDS row.
// The loop runs in trips of two blocks per lane; the rows for trip T+1 are requested at the start of trip T and
// written to the other half of the LDS buffer between its two blocks, so a trip's worth of rounds hides the fetch.
// Pieces that are not wholly inside their extent are not loaded at all (nothing is read outside the extents); the
// last < 64 bytes of an extent come from memory directly, as before.  A lane that takes a new extent sits out the
// rest of its trip (3 block slots of ~1250 per average fragment).
constexpr u32 kRowDwords = 36;                   // 128 bytes + 16 of padding: rows 8 lanes apart share banks, neighbours do not
constexpr u32 kHalfDwords = 64 * kRowDwords;

__global__ __launch_bounds__(64) void sha1_extents_staged_kernel(const u8* __restrict__ base, const u64* __restrict__ off,
                                                                  const u32* __restrict__ len, u32 n, u8* __restrict__ digests,
                                                                  u32* __restrict__ counter, const u32* __restrict__ order) {
  __shared__ u32x4 stage_raw[2 * kHalfDwords / 4];
  u32* const stage = (u32*)stage_raw;
  const u32 lane = (u32)lane_id();
  bool have = false, fresh = false, marker = false, drained = false;
  u32 idx = 0;
  const u8* p = base;                // next byte to hash
  u32 total = 0, rem = 0;
  Sha1State s = {0, 0, 0, 0, 0};
  u32 cur = 0;                       // LDS half holding this trip's rows
  const u32 src_sub = (lane >> 3) << 2;      // bpermute byte index of (lane >> 3); + 32 * i selects row 8 i + (lane >> 3)
  const u32 piece = (lane & 7) * 16;
  for (;;) {
    // (A) lanes without work take the next extent
    if (!have && !drained) {
      idx = atomicAdd(counter, 1u);
      if (idx >= n) drained = true;
      else {
        if (order) idx = order[idx];
        p = base + off[idx];
        total = rem = len[idx];
        s = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};
        marker = false;
        have = true;
        fresh = true;                // nothing staged for it yet: it starts consuming next trip
      }
    }
    if (__ballot(have) == 0ull) return;
    // (B) request the rows of the NEXT trip: a fresh lane's first 128 bytes, a streaming lane's bytes 128..255 from here
    const bool ahead = have && !fresh && rem >= 128;
    const u8* c = fresh ? p : p + 128;
    const u32 crem = !have ? 0u : fresh ? rem : ahead ? rem - 128 : 0u;
    const u64 coff = (u64)(c - base);        // offsets from the kernel argument keep the loads in the global address space
    const u32 c_lo = (u32)coff, c_hi = (u32)(coff >> 32);
    u32x4 row[8];
    u32 a_lo[8], a_hi[8], a_rem[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {            // all 24 exchanges in flight before the first address is needed
      const int src = (int)(src_sub + 32 * i);
      a_lo[i] = (u32)__builtin_amdgcn_ds_bpermute(src, (int)c_lo);
      a_hi[i] = (u32)__builtin_amdgcn_ds_bpermute(src, (int)c_hi);
      a_rem[i] = (u32)__builtin_amdgcn_ds_bpermute(src, (int)crem);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      row[i] = u32x4{0, 0, 0, 0};
      if (piece + 16 <= a_rem[i]) row[i] = *(const u32x4_u*)(base + ((((u64)a_hi[i] << 32) | a_lo[i]) + piece));
    }
    const u32* mine = stage + cur * kHalfDwords + lane * kRowDwords;
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
      if (slot == 1) {
        // (D) the requested rows go to the other half (nobody reads it during this trip)
        u32* dst = stage + (cur ^ 1) * kHalfDwords + (lane >> 3) * kRowDwords + (lane & 7) * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) *(u32x4*)(dst + 8 * i * kRowDwords) = row[i];
      }
      // (C)/(E) one block per lane
      if (have && !fresh) {
        u32 w[16];
        bool last = false;
        if (rem >= 64) {
          const u32x4* q = (const u32x4*)(mine + 16 * slot);
          const u32x4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
          p += 64; rem -= 64;
          w[0] = bswap32(v0.x); w[1] = bswap32(v0.y); w[2] = bswap32(v0.z); w[3] = bswap32(v0.w);
          w[4] = bswap32(v1.x); w[5] = bswap32(v1.y); w[6] = bswap32(v1.z); w[7] = bswap32(v1.w);
          w[8] = bswap32(v2.x); w[9] = bswap32(v2.y); w[10] = bswap32(v2.z); w[11] = bswap32(v2.w);
          w[12] = bswap32(v3.x); w[13] = bswap32(v3.y); w[14] = bswap32(v3.z); w[15] = bswap32(v3.w);
        } else {
          last = tail_block(w, p, rem, marker, total);
          p += rem; rem = 0;
        }
        sha1_rounds(w, s);
        if (last) {
          u32* o = (u32*)(digests + (size_t)idx * 20);
          o[0] = bswap32(s.a); o[1] = bswap32(s.b); o[2] = bswap32(s.c); o[3] = bswap32(s.d); o[4] = bswap32(s.e);
          have = false;
        }
      }
    }
    fresh = false;
    cur ^= 1;
    __builtin_amdgcn_wave_barrier();
  }
}

__constant__ u32 K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__device__ __forceinline__ void sha256_rounds(u32 (&w)[16], u32 (&s)[8]) {
  u32 a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
#pragma unroll
  for (int t = 0; t < 64; ++t) {
    u32 wt;
    if (t < 16) wt = w[t];
    else {
      u32 w15 = w[(t + 1) & 15], w2 = w[(t + 14) & 15];
      u32 s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
      u32 s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
      wt = w[t & 15] = w[t & 15] + s0 + w[(t + 9) & 15] + s1;
    }
    const u32 t1 = add3(add3(h, xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25)), bfi(e, f, g)), K256[t], wt);
    const u32 na = add3(t1, xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22)), maj3(a, b, c));
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = na;
  }
  s[0] += a; s[1] += b; s[2] += c; s[3] += d; s[4] += e; s[5] += f; s[6] += g; s[7] += h;
}

__global__ __launch_bounds__(256) void sha256_extents_kernel(const u8* __restrict__ base, const u64* __restrict__ off,
                                                              const u64* __restrict__ len, u32 n, u8* __restrict__ digests,
                                                              u32* __restrict__ counter, const u32* __restrict__ order,
                                                              u64 chain_min_len) {
  // same flat loop as sha1_extents_kernel: the next block (and the one after) are in flight while the 64 rounds run
  bool have = false, marker = false;
  u32 idx = 0;
  const u8* p = nullptr;
  u64 total = 0, rem = 0;
  u32 pre = 0;
  u32 s[8];
  u32x4 n0 = {0, 0, 0, 0}, n1 = n0, n2 = n0, n3 = n0, m0 = n0, m1 = n0, m2 = n0, m3 = n0;
  for (;;) {
    if (!have) {
      idx = atomicAdd(counter, 1u);
      if (idx >= n) return;
      if (order) idx = order[idx];
      if (len[idx] >= chain_min_len) continue;     // long extents: one wave each (sha256_chain_kernel)
      p = base + off[idx];
      total = rem = len[idx];
      s[0] = 0x6a09e667; s[1] = 0xbb67ae85; s[2] = 0x3c6ef372; s[3] = 0xa54ff53a;
      s[4] = 0x510e527f; s[5] = 0x9b05688c; s[6] = 0x1f83d9ab; s[7] = 0x5be0cd19;
      marker = false;
      have = true;
      pre = 0;
    }
    u32 w[16];
    bool last = false;
    if (rem >= 64) {
      if (pre == 0) {
        const u32x4_u* q = (const u32x4_u*)p; n0 = q[0]; n1 = q[1]; n2 = q[2]; n3 = q[3]; pre = 1;
      }
      const u32x4 v0 = n0, v1 = n1, v2 = n2, v3 = n3;
      p += 64; rem -= 64;
      if (pre == 2) { n0 = m0; n1 = m1; n2 = m2; n3 = m3; pre = 1; } else pre = 0;
      if (pre == 0 && rem >= 64) {
        const u32x4_u* q = (const u32x4_u*)p; n0 = q[0]; n1 = q[1]; n2 = q[2]; n3 = q[3]; pre = 1;
      }
      if (pre == 1 && rem >= 128) {
        const u32x4_u* q = (const u32x4_u*)(p + 64); m0 = q[0]; m1 = q[1]; m2 = q[2]; m3 = q[3]; pre = 2;
      }
      w[0] = bswap32(v0.x); w[1] = bswap32(v0.y); w[2] = bswap32(v0.z); w[3] = bswap32(v0.w);
      w[4] = bswap32(v1.x); w[5] = bswap32(v1.y); w[6] = bswap32(v1.z); w[7] = bswap32(v1.w);
      w[8] = bswap32(v2.x); w[9] = bswap32(v2.y); w[10] = bswap32(v2.z); w[11] = bswap32(v2.w);
      w[12] = bswap32(v3.x); w[13] = bswap32(v3.y); w[14] = bswap32(v3.z); w[15] = bswap32(v3.w);
    } else {
      last = tail_block(w, p, (u32)rem, marker, total);
      p += rem; rem = 0;
    }
    sha256_rounds(w, s);
    if (last) {
      u32* o = (u32*)(digests + (size_t)idx * 32);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = bswap32(s[i]);
      have = false;
    }
  }
}

// ---- one long chain per WAVE ------------------------------------------------------------------------
// A ZPAQ segment checksum covers a whole block (up to 16 MiB and more): a single Merkle-Damgard
// chain that no amount of lanes can split.  What can be taken off the chain is everything that does
// not depend on the running state: the 64 lanes of a wave load, byte-swap and expand the message
// schedule of the next 64 blocks in parallel (80 words per lane), and the 80 serial rounds of each
// block then run on wave-uniform values -- the state lives in SGPRs, W[t] arrives by v_readlane.
// All lanes run the 80 rounds on their own block's schedule from the same (uniform) input state;
// only lane b holds the true successor state, which is then broadcast with five v_readlane.
__device__ __forceinline__ void sha1_rounds_lane(const u32 (&w)[80], int blk, Sha1State& s) {
  u32 a = s.a, b = s.b, c = s.c, d = s.d, e = s.e;
  u32 K0 = 0x5A827999u, K1 = 0x6ED9EBA1u, K2 = 0x8F1BBCDCu, K3 = 0xCA62C1D6u;
  asm volatile("" : "+v"(K0), "+v"(K1), "+v"(K2), "+v"(K3));
#define WT(t) w[t]
  SHA1_80_ROUNDS(WT, WT)
#undef WT
  s.a += __builtin_amdgcn_readlane(a, blk); s.b += __builtin_amdgcn_readlane(b, blk);
  s.c += __builtin_amdgcn_readlane(c, blk); s.d += __builtin_amdgcn_readlane(d, blk);
  s.e += __builtin_amdgcn_readlane(e, blk);
}

__device__ __forceinline__ void sha1_chain_one(const u8* __restrict__ base, const u64* __restrict__ off,
                                               const u32* __restrict__ len, u8* __restrict__ digests, const u32 idx) {
  const int lane = lane_id();
  const u8* p = base + off[idx];
  const u64 total = len[idx];
  const u64 nfull = total >> 6;   // whole 64-byte blocks
  Sha1State s = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};
  for (u64 b0 = 0; b0 < nfull; b0 += 64) {
    const u64 mine = b0 + (u64)lane;
    u32 w[80];
    if (mine < nfull) {
      const u32x4_u* q = (const u32x4_u*)(p + mine * 64);
      const u32x4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
      w[0] = bswap32(v0.x); w[1] = bswap32(v0.y); w[2] = bswap32(v0.z); w[3] = bswap32(v0.w);
      w[4] = bswap32(v1.x); w[5] = bswap32(v1.y); w[6] = bswap32(v1.z); w[7] = bswap32(v1.w);
      w[8] = bswap32(v2.x); w[9] = bswap32(v2.y); w[10] = bswap32(v2.z); w[11] = bswap32(v2.w);
      w[12] = bswap32(v3.x); w[13] = bswap32(v3.y); w[14] = bswap32(v3.z); w[15] = bswap32(v3.w);
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t) w[t] = 0;
    }
#pragma unroll
    for (int t = 16; t < 80; ++t) w[t] = rotl32(xor3(w[t - 3], w[t - 8], w[t - 14]) ^ w[t - 16], 1);
    const int cnt = nfull - b0 < 64 ? (int)(nfull - b0) : 64;
    for (int b = 0; b < cnt; ++b) sha1_rounds_lane(w, b, s);
  }
  // closing block(s): every lane computes the same thing, lane 0 stores
  {
    const u8* q = p + nfull * 64;
    u32 rem = (u32)(total & 63);
    bool marker = false, last = false;
    while (!last) {
      u32 w[16];
      last = tail_block(w, q, rem, marker, total);
      q += rem; rem = 0;
      sha1_rounds(w, s);
    }
  }
  if (lane == 0) {
    u32* o = (u32*)(digests + (size_t)idx * 20);
    o[0] = bswap32(s.a); o[1] = bswap32(s.b); o[2] = bswap32(s.c); o[3] = bswap32(s.d); o[4] = bswap32(s.e);
  }
}

// prio: the chains ask for issue priority where they share a SIMD (a job is as long as its longest chain: ZPQ_PRIO, lz77_enc.hip)
__global__ __launch_bounds__(64) void sha1_chain_kernel(const u8* __restrict__ base, const u64* __restrict__ off,
                                                        const u32* __restrict__ len, u8* __restrict__ digests, const u32 prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);
  sha1_chain_one(base, off, len, digests, blockIdx.x);
}

// The same over a queue of extents with surplus workgroups: a wave keeps a SIMD to itself where it can (zpq_internal.h,
// cooperative wave placement)
__global__ __launch_bounds__(64) void sha1_chain_placed_kernel(const u8* __restrict__ base, const u64* __restrict__ off,
                                                               const u32* __restrict__ len, u8* __restrict__ digests, const zpq_place P) {
  u32 key; bool polite;
  u32 item = zpq_place_begin(P, key, polite);
  while (item != 0xffffffffu) {
    sha1_chain_one(base, off, len, digests, item);
    item = zpq_place_next(P, key, polite);
  }
}

// SHA-256 of a long extent (a whole file on the verify side): same shape as sha1_chain_kernel.  The 64 lanes load,
// byte-swap and expand the schedules of the next 64 blocks (W[t] + K[t], 64 words per lane); the 64 serial rounds of
// each block then run on every lane's own schedule from the uniform state and lane b's result is broadcast with
// eight v_readlane.  A wave issues one instruction per ~4 cycles whatever its dependencies, so what counts is the
// instruction count on the chain: 14 per round (the schedule's 10 per word are off the chain).
__device__ __forceinline__ void sha256_rounds_lane(const u32 (&wk)[64], int blk, u32 (&s)[8]) {
  u32 a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
#pragma unroll
  for (int t = 0; t < 64; ++t) {
    const u32 t1 = add3(h, xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25)), bfi(e, f, g)) + wk[t];
    const u32 na = add3(t1, xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22)), maj3(a, b, c));
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = na;
  }
  s[0] += __builtin_amdgcn_readlane(a, blk); s[1] += __builtin_amdgcn_readlane(b, blk);
  s[2] += __builtin_amdgcn_readlane(c, blk); s[3] += __builtin_amdgcn_readlane(d, blk);
  s[4] += __builtin_amdgcn_readlane(e, blk); s[5] += __builtin_amdgcn_readlane(f, blk);
  s[6] += __builtin_amdgcn_readlane(g, blk); s[7] += __builtin_amdgcn_readlane(h, blk);
}

// persistent waves: wave w takes the extents w, w + waves, ... whose length is at least `min_len`.  Workgroups of four
// waves: a workgroup then owns the four SIMDs of a compute unit, and the chains leave WHOLE compute units to whatever
// else runs beside them (the lane-wise kernel) instead of one SIMD here and there.
__global__ __launch_bounds__(256) void sha256_chain_kernel(const u8* __restrict__ base, const u64* __restrict__ off,
                                                           const u64* __restrict__ len, u32 n, u64 min_len,
                                                           u8* __restrict__ digests, const u32* __restrict__ list, u32* __restrict__ queue,
                                                           u32 rot) {
  const int lane = lane_id();
  const u32 waves = gridDim.x * 4u;
  // with a queue the waves take the (longest-first) list entries as they get free; without, in strides.  The first
  // entry of every wave is fixed, rotated by `rot`: several jobs in flight (each with its own launch of this kernel)
  // then put their longest chains on different compute units instead of all on the first ones to start.
  bool first = true;
  for (u32 k = blockIdx.x * 4u + (threadIdx.x >> 6);; k += waves) {
    if (queue) {
      if (first) k = (k + rot) % waves;
      else { u32 q = 0; if (lane == 0) q = atomicAdd(queue, 1u); k = __builtin_amdgcn_readfirstlane(q) + waves; }
      first = false;
      if (k >= n) { if (k < waves) continue; break; }
    }
    if (k >= n) break;
    const u32 idx = list ? list[k] : k;
    const u64 total = len[idx];
    if (total < min_len) continue;
    const u8* p = base + off[idx];
    const u64 nfull = total >> 6;
    u32 s[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    for (u64 b0 = 0; b0 < nfull; b0 += 64) {
      const u64 mine = b0 + (u64)lane;
      u32 w[64];
      if (mine < nfull) {
        const u32x4_u* q = (const u32x4_u*)(p + mine * 64);
        const u32x4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
        w[0] = bswap32(v0.x); w[1] = bswap32(v0.y); w[2] = bswap32(v0.z); w[3] = bswap32(v0.w);
        w[4] = bswap32(v1.x); w[5] = bswap32(v1.y); w[6] = bswap32(v1.z); w[7] = bswap32(v1.w);
        w[8] = bswap32(v2.x); w[9] = bswap32(v2.y); w[10] = bswap32(v2.z); w[11] = bswap32(v2.w);
        w[12] = bswap32(v3.x); w[13] = bswap32(v3.y); w[14] = bswap32(v3.z); w[15] = bswap32(v3.w);
      } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) w[t] = 0;
      }
#pragma unroll
      for (int t = 16; t < 64; ++t) {
        const u32 w15 = w[t - 15], w2 = w[t - 2];
        w[t] = add3(w[t - 16], xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3), w[t - 7]) + xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
      }
#pragma unroll
      for (int t = 0; t < 64; ++t) w[t] += K256[t];
      const int cnt = nfull - b0 < 64 ? (int)(nfull - b0) : 64;
      for (int b = 0; b < cnt; ++b) sha256_rounds_lane(w, b, s);
    }
    {
      const u8* q = p + nfull * 64;
      u32 rem = (u32)(total & 63);
      bool marker = false, last = false;
      while (!last) {
        u32 w[16];
        last = tail_block(w, q, rem, marker, total);
        q += rem; rem = 0;
        sha256_rounds(w, s);
      }
    }
    if (lane == 0) {
      u32* o = (u32*)(digests + (size_t)idx * 32);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = bswap32(s[i]);
    }
  }
}

// ---- several long chains per wave ---------------------------------------------------------------------------------
// sha256_chain_kernel spends a whole wave on one chain: 64 lanes run the same 64 rounds and one of them is right.  That is
// the fastest a single chain goes (~915 instructions per 64-byte block on the chain, 36 MB/s), but a SIMD issues one wave64
// integer instruction per four cycles whoever it comes from, so 3072 restored files of 2-51 MB cost the chip 3072 x that.
// Here a wave carries G = 64 / S chains, S lanes each: group g loads and expands the schedules of the next S blocks of ITS
// chain (lane j: block b0 + j), then all groups run the rounds of their blocks 0 .. S-1 one after the other, every lane on
// its own schedule from its group's state (VGPRs now, uniform within the group); lane g S + b holds the true successor
// state and ds_bpermute hands it to the group.  The chain is ~(915 + 544 / S + 30) instructions per block -- 4-15 % slower
// than the wave-wide form for S = 16 .. 4 -- and the wave's instructions serve G chains: the SHA-256 of Jidac's extract
// (every restored file, ZSFX/libzpaq.cpp:171-304 byte for byte) drops from ~1.5 s of the whole chip to ~1.5 s of a
// quarter (S = 16) or a sixteenth (S = 4) of it, which is what several extract jobs in flight need.
// Groups are persistent: a group whose chain ends takes the next entry of the (longest-first) list off the queue.
template <int S>
__global__ __launch_bounds__(256) void sha256_group_kernel(const u8* __restrict__ base, const u64* __restrict__ off,
                                                           const u64* __restrict__ len, u32 n, u8* __restrict__ digests,
                                                           const u32* __restrict__ list, u32* __restrict__ queue, const u32 prio) {
  // (a job is as long as its longest chain: the chains ask for issue priority over whatever else shares their SIMDs, as the
  //  block checksum chains do)
  if (prio) __builtin_amdgcn_s_setprio(3);
  const int lane = lane_id();
  const int g0 = lane & ~(S - 1), j = lane & (S - 1);          // first lane of my group, my block slot
  bool have = false, done = false;
  u32 idx = 0;
  const u8* p = nullptr;
  u64 total = 0, nfull = 0, b0 = 0;
  u32 s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (;;) {
    if (__any(!have && !done)) {
      u32 k = 0xffffffffu;
      if (!have && !done && j == 0) k = atomicAdd(queue, 1u);
      k = (u32)__builtin_amdgcn_ds_bpermute(g0 << 2, (int)k);
      if (!have && !done) {
        if (k < n) {
          idx = list ? list[k] : k;
          total = len[idx]; nfull = total >> 6; b0 = 0;
          p = base + off[idx];
          s[0] = 0x6a09e667u; s[1] = 0xbb67ae85u; s[2] = 0x3c6ef372u; s[3] = 0xa54ff53au;
          s[4] = 0x510e527fu; s[5] = 0x9b05688cu; s[6] = 0x1f83d9abu; s[7] = 0x5be0cd19u;
          have = true;
        } else done = true;
      }
    }
    if (!__any(have)) return;
    const u64 left = have ? nfull - b0 : 0;
    const u32 cnt = left < (u64)S ? (u32)left : (u32)S;
    if (__any(cnt != 0)) {
      u32 w[64];
      if ((u32)j < cnt) {
        const u32x4_u* q = (const u32x4_u*)(p + (b0 + (u64)j) * 64);
        const u32x4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
        w[0] = bswap32(v0.x); w[1] = bswap32(v0.y); w[2] = bswap32(v0.z); w[3] = bswap32(v0.w);
        w[4] = bswap32(v1.x); w[5] = bswap32(v1.y); w[6] = bswap32(v1.z); w[7] = bswap32(v1.w);
        w[8] = bswap32(v2.x); w[9] = bswap32(v2.y); w[10] = bswap32(v2.z); w[11] = bswap32(v2.w);
        w[12] = bswap32(v3.x); w[13] = bswap32(v3.y); w[14] = bswap32(v3.z); w[15] = bswap32(v3.w);
      } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) w[t] = 0;
      }
#pragma unroll
      for (int t = 16; t < 64; ++t) {
        const u32 w15 = w[t - 15], w2 = w[t - 2];
        w[t] = add3(w[t - 16], xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3), w[t - 7]) + xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
      }
#pragma unroll
      for (int t = 0; t < 64; ++t) w[t] += K256[t];
      for (u32 b = 0; b < (u32)S; ++b) {
        if (!__any(b < cnt)) break;
        u32 a = s[0], bb = s[1], c = s[2], d = s[3], e = s[4], f = s[5], gg = s[6], h = s[7];
#pragma unroll
        for (int t = 0; t < 64; ++t) {
          const u32 t1 = add3(h, xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25)), bfi(e, f, gg)) + w[t];
          const u32 na = add3(t1, xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22)), maj3(a, bb, c));
          h = gg; gg = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = na;
        }
        const int src = (g0 + (int)b) << 2;
        const u32 r0 = (u32)__builtin_amdgcn_ds_bpermute(src, (int)a), r1 = (u32)__builtin_amdgcn_ds_bpermute(src, (int)bb);
        const u32 r2 = (u32)__builtin_amdgcn_ds_bpermute(src, (int)c), r3 = (u32)__builtin_amdgcn_ds_bpermute(src, (int)d);
        const u32 r4 = (u32)__builtin_amdgcn_ds_bpermute(src, (int)e), r5 = (u32)__builtin_amdgcn_ds_bpermute(src, (int)f);
        const u32 r6 = (u32)__builtin_amdgcn_ds_bpermute(src, (int)gg), r7 = (u32)__builtin_amdgcn_ds_bpermute(src, (int)h);
        if (b < cnt) { s[0] += r0; s[1] += r1; s[2] += r2; s[3] += r3; s[4] += r4; s[5] += r5; s[6] += r6; s[7] += r7; }
      }
      b0 += cnt;
    }
    if (have && b0 >= nfull) {
      // closing block(s): every lane of the group computes the same thing, its first lane stores
      const u8* q = p + nfull * 64;
      u32 rem = (u32)(total & 63);
      bool marker = false, last = false;
      while (!last) {
        u32 w16[16];
        last = tail_block(w16, q, rem, marker, total);
        q += rem; rem = 0;
        sha256_rounds(w16, s);
      }
      if (j == 0) {
        u32* o = (u32*)(digests + (size_t)idx * 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = bswap32(s[i]);
      }
      have = false;
    }
  }
}

// ---- longest-first work order ------------------------------------------------------------------------
// A lane hashes ~13-25 MB/s, so one 508 KiB fragment is tens of milliseconds of serial work: handed
// out in input order, the last long extents would finish long after everything else (measured: the
// tail was most of the kernel).  Extents are therefore dealt longest first -- a counting sort on
// len/4096 (256 classes, descending), which is all the order LPT scheduling needs.
constexpr u32 kLenClasses = 256;
template <class L>
__device__ __forceinline__ u32 len_class(L len) {
  const u64 c = (u64)len >> 12;
  return (kLenClasses - 1) - (u32)(c < kLenClasses - 1 ? c : kLenClasses - 1);   // class 0 = longest
}

template <class L>
__global__ __launch_bounds__(256) void extent_hist_kernel(const L* __restrict__ len, u32 n, u32* __restrict__ hist) {
  __shared__ u32 h[kLenClasses];
  h[threadIdx.x] = 0;
  __syncthreads();
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) atomicAdd(&h[len_class(len[i])], 1u);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// hist[c] -> exclusive prefix (class 0 first), in place
__global__ __launch_bounds__(256) void extent_scan_kernel(u32* __restrict__ hist) {
  __shared__ u32 h[kLenClasses];
  h[threadIdx.x] = hist[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 acc = 0;
    for (u32 c = 0; c < kLenClasses; ++c) { const u32 v = h[c]; h[c] = acc; acc += v; }
  }
  __syncthreads();
  hist[threadIdx.x] = h[threadIdx.x];
}

// Each workgroup claims a contiguous run per class for its slice, then places its items (the order
// inside a class is irrelevant, so nothing downstream depends on which workgroup came first).
constexpr u32 kOrderSlice = 256 * 16;
template <class L>
__global__ __launch_bounds__(256) void extent_scatter_kernel(const L* __restrict__ len, u32 n, u32* __restrict__ cursor,
                                                             u32* __restrict__ order) {
  __shared__ u32 cnt[kLenClasses], base[kLenClasses];
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const u32 lo = blockIdx.x * kOrderSlice;
  const u32 hi = lo + kOrderSlice < n ? lo + kOrderSlice : n;
  for (u32 i = lo + threadIdx.x; i < hi; i += 256) atomicAdd(&cnt[len_class(len[i])], 1u);
  __syncthreads();
  base[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], cnt[threadIdx.x]) : 0u;
  __syncthreads();
  for (u32 i = lo + threadIdx.x; i < hi; i += 256) order[atomicAdd(&base[len_class(len[i])], 1u)] = i;
}

// Persistent launch shape: `waves` resident waves per SIMD.  Two are enough to keep the VALU issuing
// (the rounds are one dependent chain per lane) and halve the time a single long extent needs
// compared with four.
int persistent_grid(zpq_ctx* ctx, size_t n, int waves) {
  if (const char* e = getenv("ZPQ_SHA_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= 8) waves = v; }
  size_t blocks = (n + 255) / 256, cap = (size_t)ctx->cu_count * (size_t)waves;
  return (int)(blocks < cap ? (blocks ? blocks : 1) : cap);
}

// Builds the longest-first order of n extents in scratch slot 8 (main stream only); nullptr = keep
// the input order (few extents, or scratch unavailable).
template <class L>
const u32* extent_order(zpq_ctx* ctx, hipStream_t s, const L* d_len, size_t n, int grid) {
  if (s != ctx->stream || n <= (size_t)grid * 256) return nullptr;
  if (getenv("ZPQ_SHA_NO_ORDER")) return nullptr;
  u32* buf = (u32*)zpq_scratch(ctx, 8, (n + kLenClasses) * 4 + 256);
  if (!buf) return nullptr;
  u32* hist = buf;
  u32* order = buf + kLenClasses;
  if (hipMemsetAsync(hist, 0, kLenClasses * 4, s) != hipSuccess) return nullptr;
  const unsigned hb = (unsigned)std::min<size_t>((n + 255) / 256, 1024);
  hipLaunchKernelGGL(extent_hist_kernel<L>, dim3(hb), dim3(256), 0, s, d_len, (u32)n, hist);
  hipLaunchKernelGGL(extent_scan_kernel, dim3(1), dim3(256), 0, s, hist);
  hipLaunchKernelGGL(extent_scatter_kernel<L>, dim3((unsigned)((n + kOrderSlice - 1) / kOrderSlice)), dim3(256), 0, s, d_len,
                     (u32)n, hist, order);
  return order;
}

}  // namespace

int zpq_sha1_chains_on(zpq_ctx* ctx, hipStream_t s, const u8* d_base, const u64* d_off, const u32* d_len, size_t n,
                       u8* d_digests, const char* prof_name) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (n == 0) return ZPQ_OK;
  // Each chain wants a SIMD to itself (it is bound by dependent-issue latency; a co-resident wave of a
  // concurrent kernel would stretch it, or be stretched by it).  With ONE job in the process, asking for most of a
  // CU's LDS keeps other workgroups that use LDS off the CU (a 16 MiB block: 215 ms instead of ~240).  With several
  // jobs in flight (several engine contexts) that is the wrong trade: 13 chains per job x 6 jobs would take 78 of the
  // 256 CUs away from the LDS-using passes of the other jobs (fragmenter, staged SHA-1) -- measured 172 -> 144 ms
  // per step on the headline without the reservation -- so it is only made when this is the only context.
  static bool attr_set = false;
  const unsigned hog = 163840 - 512;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)sha1_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hog);
    attr_set = true;
  }
  u32 chain_prio = 0;
  {
    // few long chains (block checksums): issue priority where they share a SIMD with other jobs' waves, unless ZPQ_PRIO=lz
    static const u32 lz_first = [] { const char* e = getenv("ZPQ_PRIO"); return e && !strcmp(e, "lz") ? 1u : 0u; }();
    chain_prio = (n <= 64 && !lz_first) ? 1u : 0u;
  }
  {
    ZpqProfScope prof_scope_(ctx, prof_name, s);
    const char* hog_env = getenv("ZPQ_CHAIN_HOG");
    const bool reserve = n <= 64 && (hog_env ? atoi(hog_env) != 0 : zpq_live_contexts() == 1);
    u32* tab = (n <= 1024 && zpq_place_enabled()) ? zpq_simd_table(ctx) : nullptr;
    u32* counter = tab ? (u32*)zpq_scratch(ctx, 7, 256) : nullptr;
    if (tab && counter) {
      // few long chains beside other jobs: surplus workgroups, one chain per free SIMD (the last n drain what is left)
      counter += s == ctx->stream2 ? 41 : 40;
      ZPQ_HIP(ctx, hipMemsetAsync(counter, 0, 4, s));
      const zpq_place P{counter, tab, (u32)n, (u32)(4 * n + 64)};
      hipLaunchKernelGGL(sha1_chain_placed_kernel, dim3((unsigned)(5 * n + 64)), dim3(64), 0, s, d_base, d_off, d_len, d_digests, P);
    } else {
      hipLaunchKernelGGL(sha1_chain_kernel, dim3((unsigned)n), dim3(64), reserve ? hog : 0, s, d_base, d_off, d_len, d_digests, chain_prio);
    }
  }
  ZPQ_HIP(ctx, hipGetLastError());
  return ZPQ_OK;
}

int zpq_sha1_extents_on(zpq_ctx* ctx, hipStream_t s, const u8* d_base, const u64* d_off, const u32* d_len, size_t n,
                        u8* d_digests, const char* prof_name) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (n == 0) return ZPQ_OK;
  if (n > 0xfffffff0u) return zpq_fail(ctx, ZPQ_ERR_ARG, "too many extents");
  // one counter per stream so that the two streams never share it
  u32* counter = (u32*)zpq_scratch(ctx, 7, 256);
  if (!counter) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "scratch");
  counter += s == ctx->stream2 ? 16 : 0;
  ZPQ_HIP(ctx, hipMemsetAsync(counter, 0, 4, s));
  const int grid = persistent_grid(ctx, n, 2);
  const u32* order = extent_order(ctx, s, d_len, n, grid);
  static const int staged = [] { const char* e = getenv("ZPQ_SHA1_STAGED"); return e ? atoi(e) : 1; }();
  if (staged && n > 4096) {
    // many extents: the wave-fetched form (one wave per workgroup, 18 KiB of LDS each)
    // Four waves per compute unit (round 5, profiles/r05f_sweep_sha1_waves.txt): 24.1 ms for the 54 GB of config 2 against 27.1 ms with
    // eight -- 64 lanes of a wave read 64 pages, and the address translation is what more waves in flight make worse -- and
    // 72 KiB of every unit's LDS stay free for the other jobs.  (Parking the staging rows a block later, so that a fetch has two
    // blocks' worth of rounds to arrive, was measured too: no difference, 27.05 against 27.12 ms.)
    const int grid4 = persistent_grid(ctx, n, 1);
    ZPQ_LAUNCH(ctx, prof_name, s, sha1_extents_staged_kernel, dim3(grid4 * 4), dim3(64), d_base, d_off, d_len, (u32)n, d_digests, counter,
               order);
    ZPQ_HIP(ctx, hipGetLastError());
    return ZPQ_OK;
  }
  ZPQ_LAUNCH(ctx, prof_name, s, sha1_extents_kernel, dim3(grid), dim3(256), d_base, d_off, d_len, (u32)n, d_digests, counter,
             order);
  ZPQ_HIP(ctx, hipGetLastError());
  return ZPQ_OK;
}

extern "C" {

int zpq_sha1_extents_dev(zpq_ctx* ctx, const uint8_t* d_base, const uint64_t* d_off, const uint32_t* d_len, size_t n,
                         uint8_t* d_digests) {
  return zpq_sha1_extents_on(ctx, ctx->stream, d_base, d_off, d_len, n, d_digests);
}

int zpq_sha256_extents_dev(zpq_ctx* ctx, const uint8_t* d_base, const uint64_t* d_off, const uint64_t* d_len, size_t n,
                           uint8_t* d_digests) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (n == 0) return ZPQ_OK;
  if (n > 0xfffffff0u) return zpq_fail(ctx, ZPQ_ERR_ARG, "too many extents");
  u32* counter = (u32*)zpq_scratch(ctx, 7, 256);
  if (!counter) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "scratch");
  counter += 32;
  hipStream_t st = ctx->stream;
  ZPQ_HIP(ctx, hipMemsetAsync(counter, 0, 8, st));
  // A chain is one wave's worth of dependent instructions whichever way it is laid out (about 4.3 cycles each):
  //   one LANE per extent: 64 extents share a wave's instructions (schedule + rounds, ~1380 per 64-byte block);
  //   one WAVE per extent: the schedule moves off the chain (~950 per block: 1.45x faster), 63 lanes idle.
  // Waves on one SIMD share its issue slots, so the wave-wide form only pays while it has a SIMD to itself: the
  // longest extents (at most one per SIMD, 1 MiB and more) get a wave each on the second stream, everything else
  // goes lane-wise, longest first, on the main stream; both run at once.
  u64 chain_min = 1u << 20;
  size_t max_chains = (size_t)ctx->cu_count * 4;
  if (const char* e = getenv("ZPQ_SHA256_CHAIN_MIN")) chain_min = strtoull(e, 0, 10);
  if (const char* e = getenv("ZPQ_SHA256_CHAINS")) max_chains = strtoull(e, 0, 10);
  if (n <= (1u << 18)) {
    std::vector<u64> hl(n);
    ZPQ_HIP(ctx, hipMemcpyAsync(hl.data(), d_len, n * 8, hipMemcpyDeviceToHost, st));
    ZPQ_HIP(ctx, hipStreamSynchronize(st));
    std::vector<u32> ord(n);
    for (size_t i = 0; i < n; ++i) ord[i] = (u32)i;
    std::stable_sort(ord.begin(), ord.end(), [&](u32 a, u32 b) { return hl[a] > hl[b]; });
    // How many of the longest extents get a wave: one persistent wave per SIMD takes chains off a longest-first queue
    // (36 MB/s each), the rest goes lane-wise (11 MB/s per lane, 64 lanes per wave: 20x the throughput per SIMD, a
    // third of the speed per extent).  K minimises the later of the two finishes:
    //   chains: max(longest / 36, bytes of the K longest / (SIMDs x 36));  lanes: extent K / 11.
    size_t kc = 0;
    if (!getenv("ZPQ_SHA256_CHAIN_MIN") && !getenv("ZPQ_SHA256_CHAINS")) {
      const double rw = 36e6, rl = 11e6, simds = (double)ctx->cu_count * 4;
      double best = 1e300, sum = 0;
      for (size_t k = 0; k <= n; ++k) {
        if (k && hl[ord[k - 1]] < chain_min) break;       // short extents never get a wave
        const double tc = k ? std::max((double)hl[ord[0]] / rw, sum / (simds * rw)) : 0.0;
        const double tl = k < n ? (double)hl[ord[k]] / (k >= (size_t)simds ? rl / 2 : rl) : 0.0;   // lanes beside a chain on every SIMD get half the issue slots
        const double t = std::max(tc, tl);
        if (t < best * 0.98) { best = t; kc = k; }        // (a later K must be clearly better: lanes are the cheaper way)
        if (k < n) sum += (double)hl[ord[k]];
      }
    } else {
      while (kc < n && kc < max_chains && hl[ord[kc]] >= chain_min) ++kc;
    }
    u32* d_ord = (u32*)zpq_scratch(ctx, 8, n * 4 + 256);
    if (!d_ord) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "scratch");
    ZPQ_HIP(ctx, hipMemcpyAsync(d_ord, ord.data(), n * 4, hipMemcpyHostToDevice, st));
    ZPQ_HIP(ctx, hipStreamSynchronize(st));              // ord is a local
    // More long extents than SIMDs (an extract's restored files: thousands of megabytes-long chains): several chains per
    // wave, S lanes each (sha256_group_kernel) -- every long extent at (nearly) chain speed from the start, on a fraction
    // of the chip.  S = 8 (4 beyond eight extents per SIMD): measured on config 5 with four extract jobs in flight and twelve
    // timed jobs (profiles/r06f_extract_lanes_per_chain_twelve_jobs.txt): 827 / 702 / 722 ms per job at S = 16 / 8 / 4 against
    // 1237 ms for the wave-wide + lane-wise split -- a job alone is fastest at 16 (2.01 / 2.10 / 2.29 s), but the jobs in flight
    // share the SIMDs' issue slots and fewer waves per job is what they need.  ZPQ_SHA256_GROUP=S forces it (0 = never: the
    // split of before).
    size_t nl = 0;
    while (nl < n && hl[ord[nl]] >= chain_min) ++nl;
    int S = 0;
    {
      const size_t simds = (size_t)ctx->cu_count * 4;
      if (nl > simds) S = nl <= simds * 8 ? 8 : 4;
      if (const char* e = getenv("ZPQ_SHA256_GROUP")) { const int v = atoi(e); S = (v == 4 || v == 8 || v == 16 || v == 32) && nl ? v : 0; }
    }
    if (S) {
      kc = nl;
      ZPQ_HIP(ctx, hipEventRecord(ctx->ev2, st));
      ZPQ_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev2, 0));
      const size_t groups = 64 / (size_t)S, waves = (nl + groups - 1) / groups;
      const dim3 grid((unsigned)std::min<size_t>((waves + 3) / 4, (size_t)ctx->cu_count * 2));
      static const u32 prio = [] { const char* e = getenv("ZPQ_SHA256_PRIO"); return e ? (u32)atoi(e) : 0u; }();
      switch (S) {
        case 4: ZPQ_LAUNCH(ctx, "sha256_group_kernel", ctx->stream2, sha256_group_kernel<4>, grid, dim3(256), d_base, d_off, d_len, (u32)nl, d_digests, (const u32*)d_ord, counter + 1, prio); break;
        case 8: ZPQ_LAUNCH(ctx, "sha256_group_kernel", ctx->stream2, sha256_group_kernel<8>, grid, dim3(256), d_base, d_off, d_len, (u32)nl, d_digests, (const u32*)d_ord, counter + 1, prio); break;
        case 16: ZPQ_LAUNCH(ctx, "sha256_group_kernel", ctx->stream2, sha256_group_kernel<16>, grid, dim3(256), d_base, d_off, d_len, (u32)nl, d_digests, (const u32*)d_ord, counter + 1, prio); break;
        default: ZPQ_LAUNCH(ctx, "sha256_group_kernel", ctx->stream2, sha256_group_kernel<32>, grid, dim3(256), d_base, d_off, d_len, (u32)nl, d_digests, (const u32*)d_ord, counter + 1, prio); break;
      }
      ZPQ_HIP(ctx, hipGetLastError());
      ZPQ_HIP(ctx, hipEventRecord(ctx->ev, ctx->stream2));
    } else if (kc) {
      ZPQ_HIP(ctx, hipEventRecord(ctx->ev2, st));
      ZPQ_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev2, 0));
      const size_t cw = std::min(kc, max_chains);         // persistent waves, one per SIMD at most
      static std::atomic<u32> calls{0};
      const u32 nw = (u32)((cw + 3) / 4) * 4u;
      const u32 rot = (calls.fetch_add(1) * 341u) % nw;
      ZPQ_LAUNCH(ctx, "sha256_chain_kernel", ctx->stream2, sha256_chain_kernel, dim3((unsigned)((cw + 3) / 4)), dim3(256), d_base, d_off, d_len, (u32)kc, (u64)0,
                 d_digests, (const u32*)d_ord, counter + 1, rot);
      ZPQ_HIP(ctx, hipGetLastError());
      ZPQ_HIP(ctx, hipEventRecord(ctx->ev, ctx->stream2));
    }
    if (n > kc) {
      const int grid = persistent_grid(ctx, n - kc, 2);
      ZPQ_LAUNCH(ctx, "sha256_extents_kernel", st, sha256_extents_kernel, dim3(grid), dim3(256), d_base, d_off, d_len, (u32)(n - kc),
                 d_digests, counter, (const u32*)(d_ord + kc), ~(u64)0);
      ZPQ_HIP(ctx, hipGetLastError());
    }
    if (kc) ZPQ_HIP(ctx, hipStreamWaitEvent(st, ctx->ev, 0));
    return ZPQ_OK;
  }
  // very many extents: lengths stay on the device (class order), long ones picked out by the kernels themselves
  const int grid = persistent_grid(ctx, n, 2);
  const u32* order = extent_order(ctx, st, d_len, n, grid);
  ZPQ_LAUNCH(ctx, "sha256_extents_kernel", st, sha256_extents_kernel, dim3(grid), dim3(256), d_base, d_off, d_len, (u32)n,
             d_digests, counter, order, chain_min);
  ZPQ_HIP(ctx, hipGetLastError());
  {
    const unsigned waves = (unsigned)std::min<size_t>(n, max_chains);
    ZPQ_LAUNCH(ctx, "sha256_chain_kernel", st, sha256_chain_kernel, dim3((waves + 3) / 4), dim3(256), d_base, d_off, d_len, (u32)n, chain_min,
               d_digests, (const u32*)nullptr, (u32*)nullptr, 0u);
    ZPQ_HIP(ctx, hipGetLastError());
  }
  return ZPQ_OK;
}

// Host convenience wrappers: pack the buffers into one staging area, run the device path.
static int many(zpq_ctx* ctx, const uint8_t* const* bufs, const size_t* lens, size_t n, uint8_t* digests, int dsz) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (n == 0) return ZPQ_OK;
  size_t total = 0;
  for (size_t i = 0; i < n; ++i) total += lens[i];
  u8* d_data = (u8*)zpq_scratch(ctx, 0, total + 64);
  u8* d_meta = (u8*)zpq_scratch(ctx, 1, n * (8 + 8 + 32) + 64);
  if (!d_data || !d_meta) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "scratch");
  u64* d_off = (u64*)d_meta;
  u64* d_len64 = d_off + n;
  u8* d_dig = (u8*)(d_len64 + n);
  std::vector<u64> off(n), len64(n);
  std::vector<u32> len32(n);
  size_t o = 0;
  for (size_t i = 0; i < n; ++i) {
    off[i] = o; len64[i] = lens[i]; len32[i] = (u32)lens[i];
    if (dsz == 20 && lens[i] > 0xffffffffu) return zpq_fail(ctx, ZPQ_ERR_ARG, "SHA-1 extent > 4 GiB");
    if (lens[i]) ZPQ_HIP(ctx, hipMemcpyAsync(d_data + o, bufs[i], lens[i], hipMemcpyHostToDevice, ctx->stream));
    o += lens[i];
  }
  ZPQ_HIP(ctx, hipMemcpyAsync(d_off, off.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
  int rc;
  if (dsz == 20) {
    ZPQ_HIP(ctx, hipMemcpyAsync(d_len64, len32.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = zpq_sha1_extents_dev(ctx, d_data, d_off, (const u32*)d_len64, n, d_dig);
  } else {
    ZPQ_HIP(ctx, hipMemcpyAsync(d_len64, len64.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = zpq_sha256_extents_dev(ctx, d_data, d_off, d_len64, n, d_dig);
  }
  if (rc) return rc;
  ZPQ_HIP(ctx, hipMemcpyAsync(digests, d_dig, n * dsz, hipMemcpyDeviceToHost, ctx->stream));
  ZPQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZPQ_OK;
}

int zpq_sha1_many(zpq_ctx* ctx, const uint8_t* const* bufs, const size_t* lens, size_t n, uint8_t* digests) {
  return many(ctx, bufs, lens, n, digests, 20);
}
int zpq_sha256_many(zpq_ctx* ctx, const uint8_t* const* bufs, const size_t* lens, size_t n, uint8_t* digests) {
  return many(ctx, bufs, lens, n, digests, 32);
}

}  // extern "C"
