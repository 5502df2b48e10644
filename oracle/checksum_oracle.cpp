// oracle/checksum_oracle.cpp -- CPU ORACLE for the file-level checksums.  TEST INFRASTRUCTURE ONLY (see the header of
// zpaq_oracle.cpp: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it).
//
// zpaqfranz stores XXHASH64 + CRC-32 (default) or BLAKE3 / SHA-256 ... of every file in the i blocks
// (README.md:95-105, man/zpaqfranz.pod:44).  The code that computes them sits in the missing zpaqfranz.cpp; the
// third-party modules it vendors are named in man/zpaqfranz.pod:197-206 (Brumme's Crc32, Collet's xxHash, the
// BLAKE3 reference).  The published algorithms are restated here, serially and as plainly as possible:
//   CRC-32   ISO 3309 / zlib: reflected polynomial 0xEDB88320, init ~0, final xor ~0, bit by bit.
//   XXH64    xxHash specification (XXH64, seed 0).
//   BLAKE3   BLAKE3 specification section 2: 1 KiB chunks, 64-byte blocks, 7 rounds, left-full binary tree,
//            written as the spec's recursive definition (not as the GPU's level-by-level fold).
// Pinned by: the XXHASH64 / CRC-32 attributes of the 256 files in the reference's golden archive
// AUTOTEST/sha256.zpaq (tests/test_checksum_cpu.py), zlib.crc32 and the xxhash module of the image, and the
// published known answers of the three functions.
#include <stdint.h>
#include <string.h>

typedef uint8_t U8;
typedef uint32_t U32;
typedef uint64_t U64;

extern "C" U32 orc_crc32(const U8* p, long n) {
  U32 c = 0xffffffffu;
  for (long i = 0; i < n; ++i) {
    c ^= p[i];
    for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
  }
  return ~c;
}

namespace {
const U64 P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P3 = 1609587929392839161ull,
          P4 = 9650029242287828579ull, P5 = 2870177450012600261ull;
inline U64 rd64(const U8* p) { U64 v; memcpy(&v, p, 8); return v; }   // little-endian hosts only (x86-64 here)
inline U32 rd32(const U8* p) { U32 v; memcpy(&v, p, 4); return v; }
inline U64 rol64(U64 x, int r) { return (x << r) | (x >> (64 - r)); }
inline U64 xround(U64 acc, U64 in) { acc += in * P2; acc = rol64(acc, 31); return acc * P1; }
inline U64 xmerge(U64 acc, U64 v) { acc ^= xround(0, v); return acc * P1 + P4; }
}  // namespace

extern "C" U64 orc_xxh64(const U8* p, long n, U64 seed) {
  const U8* end = p + n;
  U64 h;
  if (n >= 32) {
    U64 v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
    do {
      v1 = xround(v1, rd64(p)); v2 = xround(v2, rd64(p + 8)); v3 = xround(v3, rd64(p + 16)); v4 = xround(v4, rd64(p + 24));
      p += 32;
    } while (p + 32 <= end);
    h = rol64(v1, 1) + rol64(v2, 7) + rol64(v3, 12) + rol64(v4, 18);
    h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
  } else h = seed + P5;
  h += (U64)n;
  while (p + 8 <= end) { h ^= xround(0, rd64(p)); h = rol64(h, 27) * P1 + P4; p += 8; }
  if (p + 4 <= end) { h ^= (U64)rd32(p) * P1; h = rol64(h, 23) * P2 + P3; p += 4; }
  while (p < end) { h ^= (U64)(*p) * P5; h = rol64(h, 11) * P1; ++p; }
  h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
  return h;
}

// ---- BLAKE3 --------------------------------------------------------------------------------------------------
namespace {
const U32 IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
const int PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
enum { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };
inline U32 ror(U32 x, int r) { return (x >> r) | (x << (32 - r)); }
inline void g(U32* s, int a, int b, int c, int d, U32 mx, U32 my) {
  s[a] = s[a] + s[b] + mx; s[d] = ror(s[d] ^ s[a], 16);
  s[c] = s[c] + s[d];      s[b] = ror(s[b] ^ s[c], 12);
  s[a] = s[a] + s[b] + my; s[d] = ror(s[d] ^ s[a], 8);
  s[c] = s[c] + s[d];      s[b] = ror(s[b] ^ s[c], 7);
}
// out = first 8 words of the compression function
void compress(const U32 cv[8], const U32 block[16], U64 counter, U32 block_len, U32 flags, U32 out[8]) {
  U32 s[16], m[16], t[16];
  for (int i = 0; i < 8; ++i) s[i] = cv[i];
  for (int i = 0; i < 4; ++i) s[8 + i] = IV[i];
  s[12] = (U32)counter; s[13] = (U32)(counter >> 32); s[14] = block_len; s[15] = flags;
  memcpy(m, block, sizeof m);
  for (int r = 0; r < 7; ++r) {
    g(s, 0, 4, 8, 12, m[0], m[1]); g(s, 1, 5, 9, 13, m[2], m[3]); g(s, 2, 6, 10, 14, m[4], m[5]); g(s, 3, 7, 11, 15, m[6], m[7]);
    g(s, 0, 5, 10, 15, m[8], m[9]); g(s, 1, 6, 11, 12, m[10], m[11]); g(s, 2, 7, 8, 13, m[12], m[13]); g(s, 3, 4, 9, 14, m[14], m[15]);
    for (int i = 0; i < 16; ++i) t[i] = m[PERM[i]];
    memcpy(m, t, sizeof m);
  }
  for (int i = 0; i < 8; ++i) out[i] = s[i] ^ s[i + 8];
}
void words(const U8* p, long n, U32 w[16]) {           // up to 64 bytes, little-endian, zero padded
  U8 b[64];
  memset(b, 0, 64);
  if (n > 0) memcpy(b, p, (size_t)n);
  for (int i = 0; i < 16; ++i) w[i] = (U32)b[4 * i] | (U32)b[4 * i + 1] << 8 | (U32)b[4 * i + 2] << 16 | (U32)b[4 * i + 3] << 24;
}
void chunk_cv(const U8* p, long n, U64 index, bool root, U32 out[8]) {
  U32 cv[8];
  memcpy(cv, IV, sizeof cv);
  const long nblk = n ? (n + 63) / 64 : 1;
  for (long b = 0; b < nblk; ++b) {
    U32 w[16];
    const long bl = n - 64 * b < 64 ? n - 64 * b : 64;
    words(p + 64 * b, bl, w);
    U32 flags = (b == 0 ? CHUNK_START : 0) | (b == nblk - 1 ? CHUNK_END | (root ? ROOT : 0) : 0);
    compress(cv, w, index, (U32)bl, flags, cv);
  }
  memcpy(out, cv, sizeof cv);
}
// subtree over the bytes p[0..n) whose first chunk has index `first`: section 2.1 -- the left child takes the
// largest power-of-two number of chunks that leaves at least one byte for the right child
void subtree_cv(const U8* p, long n, U64 first, bool root, U32 out[8]) {
  if (n <= 1024) { chunk_cv(p, n, first, root, out); return; }
  long left = 1024;
  while (left * 2 < n) left *= 2;
  U32 blk[16];
  subtree_cv(p, left, first, false, blk);
  subtree_cv(p + left, n - left, first + (U64)(left / 1024), false, blk + 8);
  compress(IV, blk, 0, 64, PARENT | (root ? ROOT : 0), out);
}
}  // namespace

extern "C" void orc_blake3(const U8* p, long n, U8 out[32]) {
  U32 h[8];
  subtree_cv(p, n, 0, true, h);
  for (int i = 0; i < 8; ++i) { out[4 * i] = (U8)h[i]; out[4 * i + 1] = (U8)(h[i] >> 8); out[4 * i + 2] = (U8)(h[i] >> 16); out[4 * i + 3] = (U8)(h[i] >> 24); }
}

// building blocks, exported so that tests can model other evaluation orders of the same tree
extern "C" void orc_blake3_chunk_cv(const U8* p, long n, U64 index, int root, U32 out[8]) { chunk_cv(p, n, index, root != 0, out); }
extern "C" void orc_blake3_parent(const U32 left[8], const U32 right[8], int root, U32 out[8]) {
  U32 blk[16];
  memcpy(blk, left, 32); memcpy(blk + 8, right, 32);
  compress(IV, blk, 0, 64, PARENT | (root ? ROOT : 0), out);
}
