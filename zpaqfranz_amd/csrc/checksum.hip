// File-level checksums zpaqfranz stores next to every file in the i blocks and checks on extract/test
// (reference README.md:95-105: "triple-check with chunked SHA-1, XXHASH64 and CRC-32",
// optional BLAKE3; the golden i blocks of AUTOTEST/sha256.zpaq carry XXHASH64 + CRC-32 of 256 files, SURVEY.md
// Appendix B.4).  The hash code itself lives in the reference's missing zpaqfranz.cpp (third-party modules named in
// man/zpaqfranz.pod:197-206: Brumme's Crc32, Collet's xxHash, the BLAKE3 team's code), so the published algorithms
// are restated here and pinned on the golden attributes and the published known answers (tests).
//
// All integer work, no MFMA.  How each one is made parallel:
//   CRC-32   linear over GF(2): every 4 KiB chunk is an independent CRC (one lane each, slicing-by-4 tables in LDS);
//            a wave per file then xors the chunk CRCs multiplied by x^(8 * bytes that follow) mod P.
//   BLAKE3   a Merkle tree by design: one lane per 1 KiB chunk (16 compressions), a wave per file folds the chaining
//            values pairwise, level by level, the odd one carried up -- which is BLAKE3's left-full tree.
//   XXH64    not splittable (non-linear accumulators): its four lanes are four GPU lanes per file; long files are
//            bound by the serial multiply-rotate chain (about 48 cycles per 32-byte stripe).
#include <algorithm>

#include "zpq_internal.h"

namespace {

// file index of global chunk `c`: largest f with base[f] <= c
__device__ __forceinline__ u32 file_of(const u32* __restrict__ base, u32 nfiles, u32 c) {
  u32 lo = 0, hi = nfiles;            // base[lo] <= c < base[hi]
  while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (base[mid] <= c) lo = mid; else hi = mid; }
  return lo;
}

// ---- CRC-32 (reflected 0xEDB88320, init and final xor ~0: zlib / IEEE 802.3) -------------------------------
constexpr u32 kCrcPoly = 0xEDB88320u;
constexpr u32 kCrcChunk = 4096;

__global__ __launch_bounds__(256) void crc32_chunks_kernel(const u8* __restrict__ data, const u64* __restrict__ file_off,
                                                           const u32* __restrict__ chunk_base, u32 nfiles, u32 nchunks,
                                                           u32* __restrict__ chunk_crc) {
  __shared__ u32 T[4][256];
  {
    u32 c = threadIdx.x;
    for (int k = 0; k < 8; ++k) c = c & 1 ? (c >> 1) ^ kCrcPoly : c >> 1;
    T[0][threadIdx.x] = c;
  }
  __syncthreads();
  {
    u32 c = T[0][threadIdx.x];
    for (int t = 1; t < 4; ++t) { c = (c >> 8) ^ T[0][c & 255]; T[t][threadIdx.x] = c; }
  }
  __syncthreads();
  for (u32 ch = blockIdx.x * 256u + threadIdx.x; ch < nchunks; ch += gridDim.x * 256u) {
    const u32 f = file_of(chunk_base, nfiles, ch);
    const u64 lo = file_off[f] + (u64)(ch - chunk_base[f]) * kCrcChunk, end = file_off[f + 1];
    const u32 n = end - lo < kCrcChunk ? (u32)(end - lo) : kCrcChunk;
    const u8* p = data + lo;
    u32 crc = 0xffffffffu;
    u32 i = 0;
    for (; i + 16 <= n; i += 16) {
      const u32x4 v = *(const u32x4_u*)(p + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        crc ^= v[k];
        crc = T[3][crc & 255] ^ T[2][(crc >> 8) & 255] ^ T[1][(crc >> 16) & 255] ^ T[0][crc >> 24];
      }
    }
    for (; i < n; ++i) crc = (crc >> 8) ^ T[0][(crc ^ p[i]) & 255];
    chunk_crc[ch] = ~crc;
  }
}

// a(x) * b(x) mod P in the reflected representation (bit 31 = x^0)
__device__ __forceinline__ u32 gf2_mulmod(u32 a, u32 b) {
  u32 p = 0;
#pragma unroll 4
  for (int k = 31; k >= 0; --k) {
    p ^= (a >> k) & 1 ? b : 0u;
    b = b & 1 ? (b >> 1) ^ kCrcPoly : b >> 1;
  }
  return p;
}

// x^(8 * nbytes) mod P; x2n[k] = x^(2^k) mod P
__device__ __forceinline__ u32 gf2_xpow_bytes(const u32* __restrict__ x2n, u64 nbytes) {
  u32 p = 1u << 31;     // x^0
  u32 k = 3;
  while (nbytes) {
    if (nbytes & 1) p = gf2_mulmod(x2n[k & 31], p);
    nbytes >>= 1; ++k;
  }
  return p;
}

// one wave per file: crc(A1 || ... || Ak) = xor_i crc(Ai) * x^(8 * bytes after Ai)
__global__ __launch_bounds__(64) void crc32_combine_kernel(const u64* __restrict__ file_off, const u32* __restrict__ chunk_base,
                                                           const u32* __restrict__ chunk_crc, const u32* __restrict__ x2n_g,
                                                           u32* __restrict__ out) {
  __shared__ u32 x2n[32];
  if (threadIdx.x < 32) x2n[threadIdx.x] = x2n_g[threadIdx.x];
  __syncthreads();
  const u32 f = blockIdx.x;
  const u64 len = file_off[f + 1] - file_off[f];
  const u32 c0 = chunk_base[f], nc = chunk_base[f + 1] - c0;
  // lane j owns a contiguous run of chunks: Horner with the fixed x^(8*4096), then one shift for what follows the run
  const u32 per = (nc + 63) / 64;
  const u32 a = threadIdx.x * per, b = a + per < nc ? a + per : nc;
  u32 acc = 0;
  if (a < b) {
    const u32 xc = gf2_xpow_bytes(x2n, kCrcChunk);
    for (u32 k = a; k < b; ++k) {
      const u64 lo = (u64)k * kCrcChunk;
      const u32 n = len - lo < kCrcChunk ? (u32)(len - lo) : kCrcChunk;
      acc = (n == kCrcChunk ? gf2_mulmod(xc, acc) : gf2_mulmod(gf2_xpow_bytes(x2n, n), acc)) ^ chunk_crc[c0 + k];
    }
    const u64 after = len - ((u64)b * kCrcChunk < len ? (u64)b * kCrcChunk : len);
    if (after) acc = gf2_mulmod(gf2_xpow_bytes(x2n, after), acc);
  }
  for (int o = 32; o; o >>= 1) acc ^= __shfl_xor(acc, o, 64);
  if (threadIdx.x == 0) out[f] = acc;
}

// ---- XXH64 (seed 0) -----------------------------------------------------------------------------------------
constexpr u64 XP1 = 11400714785074694791ull, XP2 = 14029467366897019727ull, XP3 = 1609587929392839161ull,
              XP4 = 9650029242287828579ull, XP5 = 2870177450012600261ull;
__device__ __forceinline__ u64 rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ u64 xxh_round(u64 acc, u64 in) { return rotl64(acc + in * XP2, 31) * XP1; }
__device__ __forceinline__ u64 xxh_merge(u64 acc, u64 v) { return (acc ^ xxh_round(0, v)) * XP1 + XP4; }

// four lanes per file (the algorithm's four accumulators), eight stripes in flight per lane
__global__ __launch_bounds__(256) void xxh64_kernel(const u8* __restrict__ data, const u64* __restrict__ file_off, u32 nfiles,
                                                    u64* __restrict__ out) {
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  const u32 f = t >> 2, a = t & 3;
  const bool live = f < nfiles;
  const u64 lo = live ? file_off[f] : 0, len = live ? file_off[f + 1] - lo : 0;
  const u8* p = data + lo;
  u64 acc = a == 0 ? XP1 + XP2 : a == 1 ? XP2 : a == 2 ? 0 : 0 - XP1;
  const u64 stripes = len >> 5;
  u64 s = 0;
  for (; s + 8 <= stripes; s += 8) {
    u64 in[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) in[k] = *(const u64_u*)(p + ((s + k) << 5) + 8 * a);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc = xxh_round(acc, in[k]);
  }
  for (; s < stripes; ++s) acc = xxh_round(acc, *(const u64_u*)(p + (s << 5) + 8 * a));
  // the group's four accumulators meet in its first lane
  const u64 v1 = __shfl(acc, (threadIdx.x & 60) + 0, 64), v2 = __shfl(acc, (threadIdx.x & 60) + 1, 64);
  const u64 v3 = __shfl(acc, (threadIdx.x & 60) + 2, 64), v4 = __shfl(acc, (threadIdx.x & 60) + 3, 64);
  if (!live || a != 0) return;
  u64 h;
  if (len >= 32) {
    h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
    h = xxh_merge(h, v1); h = xxh_merge(h, v2); h = xxh_merge(h, v3); h = xxh_merge(h, v4);
  } else h = XP5;
  h += len;
  const u8* q = p + (stripes << 5);
  u32 rem = (u32)(len & 31);
  while (rem >= 8) { h ^= xxh_round(0, *(const u64_u*)q); h = rotl64(h, 27) * XP1 + XP4; q += 8; rem -= 8; }
  if (rem >= 4) { h ^= (u64)(*(const u32_u*)q) * XP1; h = rotl64(h, 23) * XP2 + XP3; q += 4; rem -= 4; }
  while (rem) { h ^= (u64)(*q) * XP5; h = rotl64(h, 11) * XP1; ++q; --rem; }
  h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
  out[f] = h;
}

// ---- BLAKE3 (default hash mode, 32-byte output) ----------------------------------------------------------------
constexpr u32 B3_CHUNK_START = 1, B3_CHUNK_END = 2, B3_PARENT = 4, B3_ROOT = 8;
#define B3_IV0 0x6A09E667u
#define B3_IV1 0xBB67AE85u
#define B3_IV2 0x3C6EF372u
#define B3_IV3 0xA54FF53Au
#define B3_IV4 0x510E527Fu
#define B3_IV5 0x9B05688Cu
#define B3_IV6 0x1F83D9ABu
#define B3_IV7 0x5BE0CD19u

#define B3_G(a, b, c, d, mx, my)                  \
  a = a + b + (mx); d = rotr32(d ^ a, 16);        \
  c = c + d;        b = rotr32(b ^ c, 12);        \
  a = a + b + (my); d = rotr32(d ^ a, 8);         \
  c = c + d;        b = rotr32(b ^ c, 7);

// message word order of round r is the r-fold application of the permutation 2 6 3 10 7 0 4 13 1 11 12 5 9 14 15 8
#define B3_ROUND(m, i0, i1, i2, i3, i4, i5, i6, i7, i8, i9, i10, i11, i12, i13, i14, i15) \
  B3_G(s0, s4, s8, s12, m[i0], m[i1]) B3_G(s1, s5, s9, s13, m[i2], m[i3])                  \
  B3_G(s2, s6, s10, s14, m[i4], m[i5]) B3_G(s3, s7, s11, s15, m[i6], m[i7])                \
  B3_G(s0, s5, s10, s15, m[i8], m[i9]) B3_G(s1, s6, s11, s12, m[i10], m[i11])              \
  B3_G(s2, s7, s8, s13, m[i12], m[i13]) B3_G(s3, s4, s9, s14, m[i14], m[i15])

// cv <- first eight words of compress(cv, m, counter, block_len, flags)
__device__ __forceinline__ void b3_compress(u32 (&cv)[8], const u32 (&m)[16], u64 counter, u32 block_len, u32 flags) {
  u32 s0 = cv[0], s1 = cv[1], s2 = cv[2], s3 = cv[3], s4 = cv[4], s5 = cv[5], s6 = cv[6], s7 = cv[7];
  u32 s8 = B3_IV0, s9 = B3_IV1, s10 = B3_IV2, s11 = B3_IV3, s12 = (u32)counter, s13 = (u32)(counter >> 32), s14 = block_len, s15 = flags;
  B3_ROUND(m, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
  B3_ROUND(m, 2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8)
  B3_ROUND(m, 3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1)
  B3_ROUND(m, 10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6)
  B3_ROUND(m, 12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4)
  B3_ROUND(m, 9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7)
  B3_ROUND(m, 11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13)
  cv[0] = s0 ^ s8; cv[1] = s1 ^ s9; cv[2] = s2 ^ s10; cv[3] = s3 ^ s11;
  cv[4] = s4 ^ s12; cv[5] = s5 ^ s13; cv[6] = s6 ^ s14; cv[7] = s7 ^ s15;
}

// one lane per 1 KiB chunk; cvs[chunk] = chaining value (the digest itself for a one-chunk file: ROOT is set there)
__global__ __launch_bounds__(256) void blake3_chunks_kernel(const u8* __restrict__ data, const u64* __restrict__ file_off,
                                                            const u32* __restrict__ chunk_base, u32 nfiles, u32 nchunks,
                                                            u32* __restrict__ cvs) {
  for (u32 ch = blockIdx.x * 256u + threadIdx.x; ch < nchunks; ch += gridDim.x * 256u) {
    const u32 f = file_of(chunk_base, nfiles, ch);
    const u32 ci = ch - chunk_base[f];
    const u64 lo = file_off[f] + (u64)ci * 1024u, end = file_off[f + 1];
    const u32 n = end - lo < 1024u ? (u32)(end - lo) : 1024u;
    const bool only = chunk_base[f + 1] - chunk_base[f] == 1;
    const u8* p = data + lo;
    u32 cv[8] = {B3_IV0, B3_IV1, B3_IV2, B3_IV3, B3_IV4, B3_IV5, B3_IV6, B3_IV7};
    const u32 nblk = n ? (n + 63) / 64 : 1;
    for (u32 b = 0; b < nblk; ++b) {
      u32 m[16];
      const u32 bl = n - b * 64 < 64 ? n - b * 64 : 64;
      if (bl == 64) {
        const u32x4_u* q = (const u32x4_u*)(p + b * 64);
        const u32x4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
        m[0] = v0.x; m[1] = v0.y; m[2] = v0.z; m[3] = v0.w; m[4] = v1.x; m[5] = v1.y; m[6] = v1.z; m[7] = v1.w;
        m[8] = v2.x; m[9] = v2.y; m[10] = v2.z; m[11] = v2.w; m[12] = v3.x; m[13] = v3.y; m[14] = v3.z; m[15] = v3.w;
      } else {
#pragma unroll
        for (int w = 0; w < 16; ++w) {
          u32 x = 0;
#pragma unroll
          for (int k = 3; k >= 0; --k) { const u32 j = 4 * w + k; x = (x << 8) | (j < bl ? p[b * 64 + j] : 0u); }
          m[w] = x;
        }
      }
      u32 flags = (b == 0 ? B3_CHUNK_START : 0) | (b + 1 == nblk ? B3_CHUNK_END | (only ? B3_ROOT : 0) : 0);
      b3_compress(cv, m, ci, bl, flags);
    }
    u32x4* o = (u32x4*)(cvs + (size_t)ch * 8);
    o[0] = u32x4{cv[0], cv[1], cv[2], cv[3]};
    o[1] = u32x4{cv[4], cv[5], cv[6], cv[7]};
  }
}

// one wave per file folds its chaining values in place: level by level node j = parent(2j, 2j+1), an odd last node is
// carried up.  Batches of 64 parents read [128b, 128b+128) and write [64b, 64b+64): writes never reach unread input.
__global__ __launch_bounds__(64) void blake3_tree_kernel(const u32* __restrict__ chunk_base, u32* __restrict__ cvs,
                                                         u8* __restrict__ out) {
  const u32 f = blockIdx.x;
  u32* c = cvs + (size_t)chunk_base[f] * 8;
  u32 cnt = chunk_base[f + 1] - chunk_base[f];
  while (cnt > 1) {
    const u32 parents = cnt >> 1, next = (cnt + 1) >> 1;
    const bool root = next == 1;
    for (u32 b0 = 0; b0 < next; b0 += 64) {
      const u32 j = b0 + threadIdx.x;
      u32 m[16];
      u32 cv[8] = {B3_IV0, B3_IV1, B3_IV2, B3_IV3, B3_IV4, B3_IV5, B3_IV6, B3_IV7};
      if (j < next) {
        const u32x4* q = (const u32x4*)(c + (size_t)j * 16);
        const u32x4 v0 = q[0], v1 = q[1];
        m[0] = v0.x; m[1] = v0.y; m[2] = v0.z; m[3] = v0.w; m[4] = v1.x; m[5] = v1.y; m[6] = v1.z; m[7] = v1.w;
        if (j < parents) {
          const u32x4 v2 = q[2], v3 = q[3];
          m[8] = v2.x; m[9] = v2.y; m[10] = v2.z; m[11] = v2.w; m[12] = v3.x; m[13] = v3.y; m[14] = v3.z; m[15] = v3.w;
          b3_compress(cv, m, 0, 64, B3_PARENT | (root ? B3_ROOT : 0));
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) cv[k] = m[k];       // odd one out: carried up unchanged
        }
      }
      __syncthreads();
      if (j < next) {
        u32x4* o = (u32x4*)(c + (size_t)j * 8);
        o[0] = u32x4{cv[0], cv[1], cv[2], cv[3]};
        o[1] = u32x4{cv[4], cv[5], cv[6], cv[7]};
      }
      __syncthreads();
    }
    cnt = next;
  }
  if (threadIdx.x < 8) {
    const u32 w = c[threadIdx.x];
    u8* o = out + (size_t)f * 32 + 4 * threadIdx.x;
    o[0] = (u8)w; o[1] = (u8)(w >> 8); o[2] = (u8)(w >> 16); o[3] = (u8)(w >> 24);
  }
}

u32 host_mulmod(u32 a, u32 b) {
  u32 p = 0;
  for (int k = 31; k >= 0; --k) { if ((a >> k) & 1) p ^= b; b = b & 1 ? (b >> 1) ^ kCrcPoly : b >> 1; }
  return p;
}

}  // namespace

extern "C" int zpq_file_checksums_dev(zpq_ctx* ctx, const uint8_t* d_base, const uint64_t* file_off, size_t nfiles,
                                      uint32_t* crc32, uint64_t* xxh64, uint8_t* blake3) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (nfiles == 0) return ZPQ_OK;
  if (nfiles > 0x3fffffffu) return zpq_fail(ctx, ZPQ_ERR_ARG, "too many files");
  hipStream_t st = ctx->stream;
  // chunk prefix tables: CRC-32 chunks of 4 KiB (none for an empty file), BLAKE3 chunks of 1 KiB (one for an empty file)
  std::vector<u32> cb(nfiles + 1), bb(nfiles + 1);
  u64 cc = 0, bc = 0;
  for (size_t f = 0; f < nfiles; ++f) {
    if (file_off[f + 1] < file_off[f]) return zpq_fail(ctx, ZPQ_ERR_ARG, "file offsets must ascend");
    const u64 len = file_off[f + 1] - file_off[f];
    cb[f] = (u32)cc; bb[f] = (u32)bc;
    cc += (len + kCrcChunk - 1) / kCrcChunk;
    bc += len ? (len + 1023) / 1024 : 1;
    if (cc > 0xfffffff0ull || bc > 0xfffffff0ull) return zpq_fail(ctx, ZPQ_ERR_ARG, "input too large for one checksum call");
  }
  cb[nfiles] = (u32)cc; bb[nfiles] = (u32)bc;
  const size_t off_bytes = (nfiles + 1) * 8, tab_bytes = (nfiles + 1) * 4;
  const size_t meta_bytes = off_bytes + 2 * ((tab_bytes + 7) & ~(size_t)7) + 128 + nfiles * (4 + 8 + 32) + 64;
  u8* d_meta = (u8*)zpq_scratch(ctx, 18, meta_bytes);
  const size_t work_bytes = std::max<size_t>(crc32 ? cc * 4 : 0, blake3 ? bc * 32 : 0) + 256;
  u8* d_work = (u8*)zpq_scratch(ctx, 19, work_bytes);
  if (!d_meta || !d_work) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "checksum scratch");
  u64* d_off = (u64*)d_meta;
  u32* d_cb = (u32*)(d_meta + off_bytes);
  u32* d_bb = (u32*)((u8*)d_cb + ((tab_bytes + 7) & ~(size_t)7));
  u32* d_x2n = (u32*)((u8*)d_bb + ((tab_bytes + 7) & ~(size_t)7));
  u64* d_xxh = (u64*)(d_x2n + 32);
  u8* d_b3 = (u8*)(d_xxh + nfiles);
  u32* d_crc = (u32*)(d_b3 + nfiles * 32);
  ZPQ_HIP(ctx, hipMemcpyAsync(d_off, file_off, off_bytes, hipMemcpyHostToDevice, st));
  const int cu = ctx->cu_count;
  if (crc32) {
    u32 x2n[32];
    u32 p = 1u << 30;                       // x^1
    x2n[0] = p;
    for (int k = 1; k < 32; ++k) x2n[k] = p = host_mulmod(p, p);
    ZPQ_HIP(ctx, hipMemcpyAsync(d_cb, cb.data(), tab_bytes, hipMemcpyHostToDevice, st));
    ZPQ_HIP(ctx, hipMemcpyAsync(d_x2n, x2n, sizeof x2n, hipMemcpyHostToDevice, st));
    ZPQ_HIP(ctx, hipStreamSynchronize(st));        // x2n is a stack array
    if (cc) {
      const unsigned grid = (unsigned)std::min<u64>((cc + 255) / 256, (u64)cu * 8);
      ZPQ_LAUNCH(ctx, "crc32_chunks_kernel", st, crc32_chunks_kernel, dim3(grid), dim3(256), d_base, d_off, d_cb, (u32)nfiles, (u32)cc,
                 (u32*)d_work);
      ZPQ_HIP(ctx, hipGetLastError());
    }
    ZPQ_LAUNCH(ctx, "crc32_combine_kernel", st, crc32_combine_kernel, dim3((unsigned)nfiles), dim3(64), d_off, d_cb, (const u32*)d_work,
               d_x2n, d_crc);
    ZPQ_HIP(ctx, hipGetLastError());
    ZPQ_HIP(ctx, hipMemcpyAsync(crc32, d_crc, nfiles * 4, hipMemcpyDeviceToHost, st));
  }
  if (xxh64) {
    ZPQ_LAUNCH(ctx, "xxh64_kernel", st, xxh64_kernel, dim3((unsigned)((nfiles * 4 + 255) / 256)), dim3(256), d_base, d_off, (u32)nfiles, d_xxh);
    ZPQ_HIP(ctx, hipGetLastError());
    ZPQ_HIP(ctx, hipMemcpyAsync(xxh64, d_xxh, nfiles * 8, hipMemcpyDeviceToHost, st));
  }
  if (blake3) {
    ZPQ_HIP(ctx, hipMemcpyAsync(d_bb, bb.data(), tab_bytes, hipMemcpyHostToDevice, st));
    const unsigned grid = (unsigned)std::min<u64>((bc + 255) / 256, (u64)cu * 8);
    ZPQ_LAUNCH(ctx, "blake3_chunks_kernel", st, blake3_chunks_kernel, dim3(grid), dim3(256), d_base, d_off, d_bb, (u32)nfiles, (u32)bc,
               (u32*)d_work);
    ZPQ_HIP(ctx, hipGetLastError());
    ZPQ_LAUNCH(ctx, "blake3_tree_kernel", st, blake3_tree_kernel, dim3((unsigned)nfiles), dim3(64), d_bb, (u32*)d_work, d_b3);
    ZPQ_HIP(ctx, hipGetLastError());
    ZPQ_HIP(ctx, hipMemcpyAsync(blake3, d_b3, nfiles * 32, hipMemcpyDeviceToHost, st));
  }
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  return ZPQ_OK;
}
