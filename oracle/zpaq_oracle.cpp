// oracle/zpaq_oracle.cpp -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// A plain, serial C++ restatement of the algorithms on the zpaqfranz block-compress hot
// path.  It exists so that the HIP kernels can be checked bit for bit;
// it is NOT part of the product: only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load it.  The product (zpaqfranz_amd/) never links or calls it.
//
// Parity status: every function here is pinned either against the real reference code
// compiled in place (oracle/_ref/libzpaqref.so, see ref_shim.cpp) or against the
// reference's golden archive AUTOTEST/sha256.zpaq; tests/test_oracle_*.py hold the checks.
// Items whose reference source is absent from the snapshot AND that no fixture pins are
// marked "parity unpinned" at the function.
//
// Each function cites the reference location it follows, relative to /root/reference.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

typedef uint8_t U8;
typedef uint32_t U32;
typedef uint64_t U64;

// ------------------------------------------------------------------------------------------
// SHA-1 (FIPS 180-4).  Reference: libzpaq::SHA1, ZSFX/libzpaq.h:934-954,
// ZSFX/libzpaq.cpp:96-167 (big-endian word packing, 80 rounds, length in bits appended).
// ------------------------------------------------------------------------------------------
namespace {

inline U32 rol(U32 x, int k) { return (x << k) | (x >> (32 - k)); }

struct Sha1 {
  U32 h[5]; U8 blk[64]; U64 nbytes; int fill;
  Sha1() { reset(); }
  void reset() {
    h[0] = 0x67452301u; h[1] = 0xEFCDAB89u; h[2] = 0x98BADCFEu; h[3] = 0x10325476u; h[4] = 0xC3D2E1F0u;
    nbytes = 0; fill = 0;
  }
  void block(const U8* p) {
    U32 w[80];
    for (int t = 0; t < 16; ++t) w[t] = (U32)p[4 * t] << 24 | (U32)p[4 * t + 1] << 16 | (U32)p[4 * t + 2] << 8 | p[4 * t + 3];
    for (int t = 16; t < 80; ++t) w[t] = rol(w[t - 3] ^ w[t - 8] ^ w[t - 14] ^ w[t - 16], 1);
    U32 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
    for (int t = 0; t < 80; ++t) {
      U32 f, k;
      if (t < 20) f = (b & c) | (~b & d), k = 0x5A827999u;
      else if (t < 40) f = b ^ c ^ d, k = 0x6ED9EBA1u;
      else if (t < 60) f = (b & c) | (b & d) | (c & d), k = 0x8F1BBCDCu;
      else f = b ^ c ^ d, k = 0xCA62C1D6u;
      U32 tmp = rol(a, 5) + f + e + k + w[t];
      e = d; d = c; c = rol(b, 30); b = a; a = tmp;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e;
  }
  void update(const U8* p, size_t n) {
    nbytes += n;
    if (fill) {
      while (n && fill < 64) blk[fill++] = *p++, --n;
      if (fill == 64) block(blk), fill = 0;
    }
    while (n >= 64) block(p), p += 64, n -= 64;
    while (n) blk[fill++] = *p++, --n;
  }
  void final(U8 out[20]) {
    U64 bits = nbytes * 8;
    U8 pad = 0x80; update(&pad, 1);
    U8 z = 0; while (fill != 56) update(&z, 1);
    U8 len[8]; for (int i = 0; i < 8; ++i) len[i] = (U8)(bits >> (56 - 8 * i));
    update(len, 8);
    for (int i = 0; i < 5; ++i) { out[4 * i] = h[i] >> 24; out[4 * i + 1] = h[i] >> 16; out[4 * i + 2] = h[i] >> 8; out[4 * i + 3] = h[i]; }
    reset();
  }
};

// ------------------------------------------------------------------------------------------
// SHA-256 (FIPS 180-4).  Reference: libzpaq::SHA256, ZSFX/libzpaq.h:960-979,
// ZSFX/libzpaq.cpp:171-304.
// ------------------------------------------------------------------------------------------
const U32 K256[64] = {
  0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
  0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
  0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
  0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
  0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
  0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
  0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
  0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

inline U32 ror(U32 x, int k) { return (x >> k) | (x << (32 - k)); }

void sha256_buf(const U8* p, size_t n, U8 out[32]) {
  U32 s[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  std::vector<U8> m(p, p + n);
  m.push_back(0x80);
  while (m.size() % 64 != 56) m.push_back(0);
  U64 bits = (U64)n * 8;
  for (int i = 0; i < 8; ++i) m.push_back((U8)(bits >> (56 - 8 * i)));
  for (size_t o = 0; o < m.size(); o += 64) {
    U32 w[64];
    for (int t = 0; t < 16; ++t) w[t] = (U32)m[o + 4 * t] << 24 | (U32)m[o + 4 * t + 1] << 16 | (U32)m[o + 4 * t + 2] << 8 | m[o + 4 * t + 3];
    for (int t = 16; t < 64; ++t) {
      U32 s0 = ror(w[t - 15], 7) ^ ror(w[t - 15], 18) ^ (w[t - 15] >> 3);
      U32 s1 = ror(w[t - 2], 17) ^ ror(w[t - 2], 19) ^ (w[t - 2] >> 10);
      w[t] = w[t - 16] + s0 + w[t - 7] + s1;
    }
    U32 a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
    for (int t = 0; t < 64; ++t) {
      U32 t1 = h + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K256[t] + w[t];
      U32 t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    s[0] += a; s[1] += b; s[2] += c; s[3] += d; s[4] += e; s[5] += f; s[6] += g; s[7] += h;
  }
  for (int i = 0; i < 8; ++i) { out[4 * i] = s[i] >> 24; out[4 * i + 1] = s[i] >> 16; out[4 * i + 2] = s[i] >> 8; out[4 * i + 3] = s[i]; }
}

// floor(log2(x))+1, 0 for x==0.  Reference: lg(), ZSFX/libzpaq.cpp:6224-6233.
inline int lg(U32 x) { int r = 0; while (x) ++r, x >>= 1; return r; }

}  // namespace

// ------------------------------------------------------------------------------------------
// Content-defined fragmenter.  The add loop lives in the missing zpaqfranz.cpp; this follows
// SURVEY.md Appendix C.4 (zpaq 7.15 Jidac::add), pinned by the 388 fragment sizes + SHA-1s
// in the h-block of AUTOTEST/sha256.zpaq (tests/test_oracle_fixture.py).
//   per fragment: h=0, c1=0, o1[256]=0
//   per byte c  : h=(h+c+1)*(c==o1[c1] ? 314159265 : 271828182); o1[c1]=c; c1=c
//   cut when sz>=MAX, or (fragment<=22 && h < 2^(22-fragment) && sz>=MIN), or at EOF
//   MAX = min(8128<<fragment, blocksize-12) = 520192, MIN = min(64<<fragment, MAX) = 4096 at
//   the defaults (fragment=6, 16 MiB blocks); the caller passes MIN/MAX explicitly.
// Whether o1[] resets per fragment or per file is not distinguishable with the fixture
// (every fixture file is a single 37000-byte unit cut once): "parity unpinned" for that
// detail; per-fragment reset is what zpaq 7.15 does.
// ------------------------------------------------------------------------------------------
extern "C" long orc_chunk(const U8* buf, long n, int fragment, U32 min_frag, U32 max_frag,
                          U32* lens, long cap) {
  long nf = 0, i = 0;
  while (i < n) {
    U32 h = 0, sz = 0; U8 o1[256]; memset(o1, 0, 256); unsigned c1 = 0;
    while (i < n) {
      unsigned c = buf[i++];
      if (c == o1[c1]) h = (h + c + 1) * 314159265u; else h = (h + c + 1) * 271828182u;
      o1[c1] = (U8)c; c1 = c; ++sz;
      if (sz >= max_frag || (fragment <= 22 && h < (1u << (22 - fragment)) && sz >= min_frag)) break;
    }
    if (nf < cap) lens[nf] = sz;
    ++nf;
  }
  return nf;
}

extern "C" void orc_sha1(const U8* buf, long n, U8 out[20]) { Sha1 s; s.update(buf, (size_t)n); s.final(out); }
extern "C" void orc_sha256(const U8* buf, long n, U8 out[32]) { sha256_buf(buf, (size_t)n, out); }

// ------------------------------------------------------------------------------------------
// E8E9 transform and inverse.  Forward: e8e9(), ZSFX/libzpaq.cpp:6117-6126 (backward scan so
// that already-transformed operands are never re-read).  Inverse: the forward scan the E8E9
// PCOMP performs at end of block (ZSFX/libzpaq.h:263-270 documents it).
// ------------------------------------------------------------------------------------------
extern "C" void orc_e8e9(U8* buf, long n) {
  for (long i = n - 5; i >= 0; --i)
    if ((buf[i] & 254) == 0xe8 && ((buf[i + 4] + 1) & 254) == 0) {
      U32 a = (buf[i + 1] | buf[i + 2] << 8 | buf[i + 3] << 16) + (U32)i;
      buf[i + 1] = a; buf[i + 2] = a >> 8; buf[i + 3] = a >> 16;
    }
}
extern "C" void orc_e8e9_inverse(U8* buf, long n) {
  for (long i = 0; i + 4 < n; ++i)
    if ((buf[i] & 254) == 0xe8 && ((buf[i + 4] + 1) & 254) == 0) {
      U32 a = (buf[i + 1] | buf[i + 2] << 8 | buf[i + 3] << 16) - (U32)i;
      buf[i + 1] = a; buf[i + 2] = a >> 8; buf[i + 3] = a >> 16;
    }  // no skip: undoing in ascending order replays the forward steps last-in-first-out
}

// ------------------------------------------------------------------------------------------
// LZ77 encoder with the bucketed hash-table match finder: level 1 (variable-length codes) and level 2 (byte-aligned
// codes, where a far match must be 1 / 2 bytes longer to be taken, :6415-6416).
// Reference: LZBuffer::LZBuffer / fill / write_literal / write_match / putb / flush,
// ZSFX/libzpaq.cpp:6140-6552; code formats documented at :6211-6222.
// args[] as in ZSFX/libzpaq.cpp:6128-6138.  The hash table (args[5]-args[0] < 21) without a secondary context
// (args[3]==0, args[6]==0) is restated: that is every configuration compressBlock emits for method 1x (SURVEY.md
// Appendix C.3), and "x<N>,2,..." / "x<N>,6,..." with a small N6.
// ------------------------------------------------------------------------------------------
namespace {
struct BitSink {
  std::vector<U8>& v; U32 acc; int cnt;
  explicit BitSink(std::vector<U8>& out) : v(out), acc(0), cnt(0) {}
  void put(U32 x, int k) {  // LSB first, ZSFX/libzpaq.cpp:6171-6179
    if (k == 0) return;
    x &= (k >= 32) ? 0xffffffffu : ((1u << k) - 1);
    acc |= x << cnt; cnt += k;
    while (cnt > 7) v.push_back((U8)acc), acc >>= 8, cnt -= 8;
  }
  void flush() { if (cnt > 0) v.push_back((U8)acc); acc = 0; cnt = 0; }  // :6182-6186
};

void emit_literals(BitSink& bs, const U8* in, U32 end, U32 lit) {  // write_literal, :6464-6478
  if (lit < 1) return;
  int ll = lg(lit);
  bs.put(0, 2);
  for (int b = ll - 2; b >= 0; --b) { bs.put(1, 1); bs.put((lit >> b) & 1, 1); }
  bs.put(0, 1);
  for (U32 j = end - lit; j < end; ++j) bs.put(in[j], 8);
}

void emit_match(BitSink& bs, U32 len, U32 off, int rb) {  // write_match level 1, :6494-6516
  int ll = lg(len) - 1;
  off += (1u << rb) - 1;
  int lo = lg(off) - 1 - rb;
  bs.put((lo + 8) >> 3, 2);
  bs.put(lo & 7, 3);
  for (int b = ll - 1; b >= 2; --b) { bs.put(1, 1); bs.put((len >> b) & 1, 1); }
  bs.put(0, 1);
  bs.put(len & 3, 2);
  bs.put(off, rb);
  bs.put(off >> rb, lo);
}
}  // namespace

// Optional token trace for kernel debugging: each match as (pos,len,off).
extern "C" long orc_lz77_encode(const U8* in, long n_, const int args[9], U8* out, long cap,
                                U32* trace, long trace_cap, long* ntrace) {
  const U32 n = (U32)n_;
  const int level = args[1] & 3;
  if (args[3] < 0 || args[6] < 0 || (level != 1 && level != 2) || args[5] - args[0] >= 21 || args[2] < 4) return -10;
  const int checkbits = 12 - args[0];                 // :6253
  const U32 htsize = 1u << args[5];                   // :6250
  const U32 minMatch = args[2], maxMatch = (1u << 14) * 3, maxLiteral = (1u << 14) / 4;  // :6258-6262
  const U32 bucket = (1u << args[4]) - 1;             // :6265
  const int shift1 = (args[5] - 1) / minMatch + 1;    // :6266
  const U32 minMatch2 = args[3], lookahead = args[6]; // :6257, :6263: the second (higher-order) context and how far in front of it a match may start
  const int shift2 = minMatch2 > 0 ? (args[5] - 1) / minMatch2 + 1 : 0;   // :6267
  const U32 minMatchBoth = (minMatch > minMatch2 + lookahead ? minMatch : minMatch2 + lookahead) + 4;   // :6268
  const int rb = args[0] > 4 ? args[0] - 4 : 0;       // :6269
  const U32 mask = (1u << checkbits) - 1;
  std::vector<U32> ht(htsize, 0);
  std::vector<U8> v; v.reserve(n / 2 + 16);
  BitSink bs(v);
  auto write_literal = [&](U32 end, U32 lit) {        // :6464-6489
    if (level == 1) { emit_literals(bs, in, end, lit); return; }
    while (lit > 0) {                                 // level 2: 00xxxxxx then x+1 bytes
      const U32 lit1 = lit > 64 ? 64 : lit;
      v.push_back((U8)(lit1 - 1));
      for (U32 j = end - lit; j < end - lit + lit1; ++j) v.push_back(in[j]);
      lit -= lit1;
    }
  };
  auto write_match = [&](U32 len, U32 off) {          // :6494-6549
    if (level == 1) { emit_match(bs, len, off, rb); return; }
    --off;
    while (len > 0) {                                 // pieces of minMatch .. minMatch+63 bytes
      const U32 len1 = len > minMatch * 2 + 63 ? minMatch + 63 : len > minMatch + 63 ? len - minMatch : len;
      if (off < (1u << 16)) { v.push_back((U8)(64 + len1 - minMatch)); v.push_back((U8)(off >> 8)); v.push_back((U8)off); }
      else if (off < (1u << 24)) { v.push_back((U8)(128 + len1 - minMatch)); v.push_back((U8)(off >> 16)); v.push_back((U8)(off >> 8)); v.push_back((U8)off); }
      else { v.push_back((U8)(192 + len1 - minMatch)); v.push_back((U8)(off >> 24)); v.push_back((U8)(off >> 16)); v.push_back((U8)(off >> 8)); v.push_back((U8)off); }
      len -= len1;
    }
  };
  U32 i = 0, h1 = 0, h2 = 0, lit = 0; long nt = 0;
  auto byte_at = [&](U32 x) -> U32 { return x < n ? in[x] : 0u; };   // (the reference reads in[i+3] behind the block in the last 3 positions of the h2 search: taken as 0)
  while (i < n) {                                     // fill(), :6329-6453
    U32 blen = minMatch - 1, bp = 0, blit = 0; int bscore = 0;
    if (minMatch2 > 0) {                              // :6373-6393: the higher order first
      for (U32 k = 0; k <= bucket; ++k) {
        U32 p = ht[h2 ^ k];
        if (p && (p & mask) == (byte_at(i + 3) & mask)) {
          p >>= checkbits;
          if (p < i && i + blen <= n && in[p + blen - 1] == in[i + blen - 1]) {
            U32 l = lookahead;                        // counted from the lookahead on ...
            while (i + l < n && l < maxMatch && in[p + l] == in[i + l]) ++l;
            if (l >= minMatch2 + lookahead) {
              int l1 = (int)lookahead;                // ... then back: what is left in front are leading literals
              while (l1 > 0 && in[p + l1 - 1] == in[i + l1 - 1]) --l1;
              int score = (int)(l - l1) * 8 - lg(i - p) - 8 * (lit == 0 && l1 > 0) - 11;
              if (score > bscore) blen = l, bp = p, blit = (U32)l1, bscore = score;
            }
          }
        }
        if (blen >= 128) break;
      }
    }
    if (!minMatch2 || blen < minMatch2)
    for (U32 k = 0; k <= bucket; ++k) {               // :6396-6408
      U32 p = ht[h1 ^ k];
      if (p && i + 3 < n && (p & mask) == (in[i + 3] & mask)) {
        p >>= checkbits;
        if (p < i && i + blen <= n && in[p + blen - 1] == in[i + blen - 1]) {
          U32 l = 0;
          while (i + l < n && l < maxMatch && in[p + l] == in[i + l]) ++l;
          int score = (int)(l * 8) - lg(i - p) - 2 * (lit > 0) - 11;
          if (score > bscore) blen = l, bp = p, blit = 0, bscore = score;
        }
      }
      if (blen >= 128) break;
    }
    const U32 off = i - bp;                           // :6413-6421
    if (off > 0 && bscore > 0 && blen - blit >= minMatch + (level == 2) * ((off >= (1u << 16)) + (off >= (1u << 24)))) {
      lit += blit;
      write_literal(i + blit, lit); lit = 0;
      write_match(blen - blit, off);
      if (trace && nt < trace_cap) { trace[3 * nt] = i + blit; trace[3 * nt + 1] = blen - blit; trace[3 * nt + 2] = off; }
      ++nt;
    } else { blen = 1; ++lit; }
    while (blen--) {                                  // :6432-6447
      if (i + minMatchBoth < n) {
        U32 ih = ((i * 1234547u) >> 19) & bucket;
        const U32 pv = (i << checkbits) | (in[i + 3] & mask);
        if (minMatch2) {
          ht[h2 ^ ih] = pv;
          h2 = (((h2 * 9) << shift2) + (in[i + minMatch2 + lookahead] + 1) * 23456789u) & (htsize - 1);
        }
        ht[h1 ^ ih] = pv;
        h1 = (((h1 * 5) << shift1) + (in[i + minMatch] + 1) * 123456791u) & (htsize - 1);
      }
      ++i;
    }
    if (lit >= maxLiteral) { write_literal(i, lit); lit = 0; }  // :6450-6451
  }
  write_literal(n, lit);                              // :6456-6460
  if (level == 1) bs.flush();
  if (ntrace) *ntrace = nt;
  if ((long)v.size() > cap) return -2;
  if (!v.empty()) memcpy(out, v.data(), v.size());
  return (long)v.size();
}

// ------------------------------------------------------------------------------------------
// Suffix array.  The reference calls divsufsort (ZSFX/libzpaq.cpp:6047-6072, body :4334-6040); the
// suffix array of a string is unique, so any correct construction is a restatement of its RESULT.
// This one is plain prefix doubling with std::sort (O(n log^2 n)): slow, obviously right, and pinned
// against the real divsufsort (oracle/_ref, ref_divsufsort) by tests/test_sa_cpu.py.
// ------------------------------------------------------------------------------------------
extern "C" long orc_suffix_array(const U8* in, long n_, U32* sa) {
  const U32 n = (U32)n_;
  if (!n) return 0;
  std::vector<U32> rank(n), tmp(n);
  for (U32 i = 0; i < n; ++i) sa[i] = i, rank[i] = in[i];
  for (U32 h = 1;; h *= 2) {
    auto key2 = [&](U32 i) -> long { return i + h < n ? (long)rank[i + h] : -1; };   // the shorter suffix sorts first
    auto less = [&](U32 a, U32 b) { return rank[a] != rank[b] ? rank[a] < rank[b] : key2(a) < key2(b); };
    std::sort(sa, sa + n, less);
    tmp[sa[0]] = 0;
    for (U32 j = 1; j < n; ++j) tmp[sa[j]] = tmp[sa[j - 1]] + (less(sa[j - 1], sa[j]) ? 1 : 0);
    rank.swap(tmp);
    if (rank[sa[n - 1]] == n - 1 || h >= n) break;
  }
  return n;
}

// ------------------------------------------------------------------------------------------
// LZ77 with the suffix-array match finder (args[5]-args[0] >= 21): what compressBlock selects for
// method 2 ("x<N>,1,4,0,7,<21+N>,1": minMatch 4, 127 neighbours either side, one byte of lookahead).
// Reference: LZBuffer::LZBuffer :6260-6311 (sa, windowed isa), fill :6329-6372 (candidate search)
// and :6412-6453 (decision, literals), write_literal / write_match :6463-6550 (levels 1 and 2).
// The windowed inverse (isa[] holds one 2^(17+args[0]) window, rebuilt when the parse enters a new
// one) is kept as the reference has it, because it decides when the lookahead search is skipped.
// `sa_in` may be null (the suffix array is then built here).
// ------------------------------------------------------------------------------------------
extern "C" long orc_lz77_sa_encode(const U8* in, long n_, const int args[9], const U32* sa_in, U8* out, long cap,
                                   U32* trace, long trace_cap, long* ntrace) {
  const U32 n = (U32)n_;
  const int level = args[1] & 3;
  if (args[5] - args[0] < 21 || (level != 1 && level != 2)) return -10;
  const U32 minMatch = args[2];
  if ((minMatch < 4 && level == 1) || (minMatch < 1 && level == 2)) return -10;
  const int checkbits = 17 + args[0];                  // :6253
  const U32 maxMatch = (1u << 14) * 3, maxLiteral = (1u << 14) / 4;
  const U32 lookahead = args[6];
  const U32 bucket = (1u << args[4]) - 1;
  const int rb = args[0] > 4 ? args[0] - 4 : 0;
  const U32 mask = (1u << checkbits) - 1;
  std::vector<U32> sav;
  if (!sa_in) { sav.resize(n ? n : 1); orc_suffix_array(in, n, sav.data()); }
  const U32* sa = sa_in ? sa_in : sav.data();
  std::vector<U32> isa((size_t)1 << checkbits, 0);      // Array<unsigned> is zero-filled
  std::vector<U8> v; v.reserve(n / 2 + 16);
  BitSink bs(v);
  auto write_literal = [&](U32 i, U32& lit) {
    if (level == 1) { emit_literals(bs, in, i, lit); lit = 0; return; }
    while (lit > 0) {                                    // level 2: 00xxxxxx then x+1 bytes
      U32 lit1 = lit > 64 ? 64 : lit;
      v.push_back((U8)(lit1 - 1));
      for (U32 j = i - lit; j < i - lit + lit1; ++j) v.push_back(in[j]);
      lit -= lit1;
    }
  };
  auto write_match = [&](U32 len, U32 off) {
    if (level == 1) { emit_match(bs, len, off, rb); return; }
    --off;
    while (len > 0) {
      const U32 len1 = len > minMatch * 2 + 63 ? minMatch + 63 : len > minMatch + 63 ? len - minMatch : len;
      if (off < (1u << 16)) { v.push_back((U8)(64 + len1 - minMatch)); v.push_back((U8)(off >> 8)); v.push_back((U8)off); }
      else if (off < (1u << 24)) { v.push_back((U8)(128 + len1 - minMatch)); v.push_back((U8)(off >> 16)); v.push_back((U8)(off >> 8)); v.push_back((U8)off); }
      else { v.push_back((U8)(192 + len1 - minMatch)); v.push_back((U8)(off >> 24)); v.push_back((U8)(off >> 16)); v.push_back((U8)(off >> 8)); v.push_back((U8)off); }
      len -= len1;
    }
  };
  U32 i = 0, lit = 0; long nt = 0;
  while (i < n) {
    U32 blen = minMatch - 1, bp = 0, blit = 0; int bscore = 0;
    if (sa[isa[i & mask]] != i)                          // :6341-6345
      for (U32 j = 0; j < n; ++j)
        if ((sa[j] & ~mask) == (i & ~mask)) isa[sa[j] & mask] = j;
    for (U32 h = 0; h <= lookahead; ++h) {               // :6346-6371
      const U32 q = isa[(h + i) & mask];
      if (sa[q] != h + i) continue;
      for (int j = -1; j <= 1; j += 2) {
        for (U32 k = 1; k <= bucket; ++k) {
          U32 p;
          const U32 qq = q + (U32)(j * (int)k);          // unsigned wrap, as in the reference
          if (qq < n && (p = sa[qq] - h) < i) {
            U32 l, l1;
            for (l = h; i + l < n && l < maxMatch && in[p + l] == in[i + l]; ++l) {}
            for (l1 = h; l1 > 0 && in[p + l1 - 1] == in[i + l1 - 1]; --l1) {}
            int score = (int)(l - l1) * 8 - lg(i - p) - 4 * (lit == 0 && l1 > 0) - 11;
            for (U32 a = 0; a < h; ++a) score = score * 5 / 8;
            if (score > bscore) blen = l, bp = p, blit = l1, bscore = score;
            if (l < blen || l < minMatch || l > 255) break;
          }
        }
      }
      if (bscore <= 0 || blen < minMatch) break;
    }
    const U32 off = i - bp;                              // :6414-6427
    if (off > 0 && bscore > 0 && blen - blit >= minMatch + (level == 2) * ((off >= (1u << 16)) + (off >= (1u << 24)))) {
      lit += blit;
      write_literal(i + blit, lit);
      write_match(blen - blit, off);
      if (trace && nt < trace_cap) { trace[3 * nt] = i + blit; trace[3 * nt + 1] = blen - blit; trace[3 * nt + 2] = off; }
      ++nt;
    } else { blen = 1; ++lit; }
    i += blen;                                           // :6430-6431
    if (lit >= maxLiteral) write_literal(i, lit);        // :6450-6452
  }
  write_literal(n, lit);
  if (level == 1) bs.flush();
  if (ntrace) *ntrace = nt;
  if ((long)v.size() > cap) return -2;
  if (!v.empty()) memcpy(out, v.data(), v.size());
  return (long)v.size();
}

// BWT output of LZBuffer (level 3, ZSFX/libzpaq.cpp:6317-6326): n+5 bytes -- the last input byte, then for every suffix
// in order the byte before it (255 where the suffix is the whole block, whose 1-based rank goes to the last four bytes,
// LSB first).  `sa_in` may be null.
extern "C" long orc_bwt_encode(const U8* in, long n_, const U32* sa_in, U8* out, long cap) {
  const U32 n = (U32)n_;
  if (cap < (long)n + 5) return -2;
  std::vector<U32> sav;
  if (!sa_in) { sav.resize(n ? n : 1); orc_suffix_array(in, n, sav.data()); }
  const U32* sa = sa_in ? sa_in : sav.data();
  U32 idx = 0;
  for (U32 i = 0; i < n + 5; ++i) {
    if (i == 0) out[i] = n > 0 ? in[n - 1] : 255;
    else if (i > n) { out[i] = (U8)(idx & 255); idx >>= 8; }
    else if (sa[i - 1] == 0) { idx = i; out[i] = 255; }
    else out[i] = in[sa[i - 1] - 1];
  }
  return (long)n + 5;
}

// The candidate search of the suffix-array parse as a FUNCTION of (position, lit == 0): what LZBuffer::fill computes at
// :6339-6372 and decides at :6414-6417 when it stands at position i with or without pending literals.  The search reads
// only the input, SA and the (windowed) inverse, never the parse so far -- which is what lets the GPU evaluate every
// position up front (lz77_sa.hip) and leaves the chain "take the match and skip, or count a literal" as the only serial
// part.  rec[2*i + (lit>0)] = 0 when no match is taken, else blen | blit << 16 | (u64)offset << 32 (blen includes blit).
// tests/test_sa_chain_cpu.py replays the chain over these records and must get orc_lz77_sa_encode's tokens.
extern "C" long orc_lz77_sa_decisions(const U8* in, long n_, const int args[9], const U32* sa, U64* rec) {
  const U32 n = (U32)n_;
  const int level = args[1] & 3;
  if (args[5] - args[0] < 21 || (level != 1 && level != 2)) return -10;
  const U32 minMatch = args[2];
  const int checkbits = 17 + args[0];
  const U32 maxMatch = (1u << 14) * 3;
  const U32 lookahead = args[6];
  const U32 bucket = (1u << args[4]) - 1;
  std::vector<U32> isa(n ? n : 1);
  for (U32 j = 0; j < n; ++j) isa[sa[j]] = j;
  for (U32 i = 0; i < n; ++i)
    for (int state = 0; state < 2; ++state) {
      const bool lit0 = state == 0;
      U32 blen = minMatch - 1, bp = 0, blit = 0; int bscore = 0;
      for (U32 h = 0; h <= lookahead; ++h) {
        // the reference's isa[] holds the window of i only: a position in the next window is not found (:6347-6349)
        if (h + i >= n || ((h + i) >> checkbits) != (i >> checkbits)) continue;
        const U32 q = isa[h + i];
        for (int j = -1; j <= 1; j += 2) {
          for (U32 k = 1; k <= bucket; ++k) {
            U32 p;
            const U32 qq = q + (U32)(j * (int)k);
            if (qq < n && (p = sa[qq] - h) < i) {
              U32 l, l1;
              for (l = h; i + l < n && l < maxMatch && in[p + l] == in[i + l]; ++l) {}
              for (l1 = h; l1 > 0 && in[p + l1 - 1] == in[i + l1 - 1]; --l1) {}
              int score = (int)(l - l1) * 8 - lg(i - p) - 4 * (lit0 && l1 > 0) - 11;
              for (U32 a = 0; a < h; ++a) score = score * 5 / 8;
              if (score > bscore) blen = l, bp = p, blit = l1, bscore = score;
              if (l < blen || l < minMatch || l > 255) break;
            }
          }
        }
        if (bscore <= 0 || blen < minMatch) break;
      }
      const U32 off = i - bp;
      const bool take = off > 0 && bscore > 0 && blen - blit >= minMatch + (level == 2) * ((off >= (1u << 16)) + (off >= (1u << 24)));
      rec[2 * (size_t)i + state] = take ? (U64)blen | ((U64)blit << 16) | ((U64)off << 32) : 0;
    }
  return n;
}

// LZ77 level 1 decoder: a native restatement of what the level-1 PCOMP program does
// (SURVEY.md Appendix D disassembly; code format ZSFX/libzpaq.cpp:6211-6222).  The PCOMP is a
// byte-at-a-time state machine; codes that are cut short by end of input are dropped, as the
// program's EOS branch (a>255 -> reset) does.  Offsets index a 2^pm ring in the VM; for a
// valid stream (offset <= bytes produced) that equals plain back-references.
extern "C" long orc_lz77_decode(const U8* in, long n, int rb, U8* out, long cap) {
  U64 acc = 0; int cnt = 0; long ip = 0, op = 0;
  auto need = [&](int k) -> bool {
    while (cnt < k) { if (ip >= n) return false; acc |= (U64)in[ip++] << cnt; cnt += 8; }
    return true;
  };
  auto take = [&](int k) -> U32 { U32 x = (U32)(acc & ((1ull << k) - 1)); acc >>= k; cnt -= k; return x; };
  while (true) {
    if (!need(2)) break;
    U32 mm = take(2);
    if (mm == 0) {  // literal run
      U32 len = 1;
      while (true) {
        if (!need(1)) return op;
        if (!take(1)) break;
        if (!need(1)) return op;
        len = len * 2 + take(1);
      }
      for (U32 j = 0; j < len; ++j) {
        if (!need(8)) return op;
        if (op >= cap) return -2;
        out[op++] = (U8)take(8);
      }
    } else {        // match
      if (!need(3)) break;
      int lo = (int)((mm - 1) * 8 + take(3));
      U32 len = 1;
      while (true) {
        if (!need(1)) return op;
        if (!take(1)) break;
        if (!need(1)) return op;
        len = len * 2 + take(1);
      }
      if (!need(2)) return op;
      len = len * 4 + take(2);
      U32 r = 0;
      if (rb) { if (!need(rb)) return op; r = take(rb); }
      if (!need(lo)) return op;
      U32 q = (lo ? take(lo) : 0) | (1u << lo);
      U32 off = ((q << rb) | r) - ((1u << rb) - 1);
      if ((long)off > op) return -3;  // reference would read zero-initialised ring memory
      if (op + (long)len > cap) return -2;
      for (U32 j = 0; j < len; ++j) { out[op] = out[op - off]; ++op; }
    }
  }
  return op;
}

// ------------------------------------------------------------------------------------------
// Block framing for n=0 ("stored") blocks: Compressor::writeTag/startBlock/startSegment/
// postProcess/compress/endSegment/endBlock and Encoder stored mode.  The method bodies are
// absent from the snapshot (ZSFX/libzpaq.cpp:2383-2385; declarations ZSFX/libzpaq.h:1273-1286,
// 1340-1371); the byte layout is pinned by the reader, Decompresser, ZSFX/libzpaq.cpp:2239-2366,
// Decoder::decompress stored branch :2139-2146, PostProcessor::write :2185-2226, and by all six
// blocks of AUTOTEST/sha256.zpaq.
// ------------------------------------------------------------------------------------------
namespace {
const U8 kTag[13] = {0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3};

// LZ77 level-1 post-processor, rb=0, no E8E9: golden bytes from the i-blocks of
// AUTOTEST/sha256.zpaq, WITHOUT the 2-byte length prefix.
const U8 kPcompLz1[302] = {
  0xef,0xff,0x2f,0x0d,0x04,0x0c,0x14,0x1c,0x37,0x01,0x37,0x02,0x37,0x03,0x37,0x04,0x38,0xcb,0x82,0x50,0x47,0x08,0x83,0x58,0x07,0x01,0xdf,0x00,0x2f,0x33,0x47,0x01,0x37,0x02,0x42,0xaf,0x03,0xef,0x00,0x2f,0x1e,0x02,0xcf,0x03,0x37,0x03,
  0x42,0xd7,0x02,0x50,0x0f,0x03,0xaf,0x07,0x81,0x37,0x03,0x42,0xd7,0x03,0x50,0x43,0x8f,0x05,0x58,0x47,0x01,0x37,0x01,0x3f,0x0a,0x42,0xd7,0x02,0x50,0x1a,0x1a,0x47,0x03,0x37,0x01,0x07,0x01,0xdf,0x01,0x2f,0x3d,0x43,0xef,0x02,0x2f,0x38,0x42,0xaf,
  0x01,0xdf,0x01,0x2f,0x15,0x42,0xd7,0x01,0x50,0x0f,0x02,0x42,0xaf,0x01,0x81,0x81,0x37,0x02,0x42,0xd7,0x01,0x50,0x1a,0x1a,0x3f,0x1a,0x42,0xd7,0x01,0x50,0x07,0x02,0xcf,0x02,0x48,0x42,0xaf,0x03,0x81,0x37,0x02,0x42,0xd7,0x02,0x50,0x1a,0x1a,0x1a,
  0x47,0x02,0x37,0x01,0x3f,0xbd,0x07,0x01,0xdf,0x02,0x2f,0x39,0x07,0x03,0xeb,0x27,0x34,0x42,0x37,0x06,0x43,0x37,0x07,0x0f,0x03,0x47,0x01,0xc9,0x58,0x02,0xaa,0x83,0x58,0x0f,0x04,0x41,0x8b,0x50,0x1f,0x02,0x43,0xef,0x00,0x2f,0x08,0x1a,0x45,0x60,
  0x11,0x09,0x39,0x3f,0xf3,0x41,0x37,0x04,0x07,0x06,0x0f,0x03,0xd1,0x50,0x07,0x07,0x89,0x58,0x04,0x37,0x01,0x07,0x01,0xdf,0x03,0x2f,0x2b,0x43,0xef,0x01,0x2f,0x26,0x42,0xaf,0x01,0xdf,0x01,0x2f,0x14,0x42,0xd7,0x01,0x50,0x0f,0x02,0xaf,0x01,0x81,
  0x81,0x37,0x02,0x42,0xd7,0x01,0x50,0x1a,0x1a,0x3f,0x09,0x42,0xd7,0x01,0x50,0x1a,0x47,0x04,0x37,0x01,0x3f,0xcf,0x07,0x01,0xdf,0x04,0x2f,0x22,0x43,0xef,0x07,0x2f,0x1d,0x0f,0x04,0x42,0x60,0x39,0x09,0x41,0x37,0x04,0x42,0xd7,0x08,0x50,0x43,0x8f,
  0x08,0x58,0x07,0x02,0x02,0x37,0x02,0xdf,0x00,0x2f,0x03,0x04,0x37,0x01,0x38,0x00};

struct Out {
  std::vector<U8> v;
  void put(int c) { v.push_back((U8)c); }
  void write(const void* p, size_t n) { const U8* q = (const U8*)p; v.insert(v.end(), q, q + n); }
};

// Encoder stored mode: sub-blocks of at most 65536 bytes, each
// preceded by its 4-byte big-endian length; the zero-length terminator is written by
// endSegment as the four 0 bytes.
void stored_subblocks(Out& o, const std::vector<U8>& payload) {
  size_t i = 0;
  while (i < payload.size()) {
    size_t k = payload.size() - i; if (k > 65536) k = 65536;
    o.put((int)(k >> 24)); o.put((int)(k >> 16) & 255); o.put((int)(k >> 8) & 255); o.put((int)k & 255);
    o.write(&payload[i], k); i += k;
  }
}
}  // namespace

// Header bytes (hsize[2] hh hm ph pm n COMP 0 HCOMP 0) for the n=0 configurations makeConfig
// produces:
//   kind 0: method "0"          -> "comp 0 0 0 0 0 hcomp end"                  (pinned: c/h blocks)
//   kind 1: method "x<a0>,0"    -> "comp 9 16 0 0 0 hcomp c-- *c=a a+= 255 d=a *d=c halt end"
//                                                                             (parity unpinned)
//   kind 2: LZ77 level 1        -> "comp 9 16 0 $1+20 0 hcomp ... halt pcomp lazy2 3 ; ..."
//                                                                             (pinned: i blocks, arg0=0)
static void header_bytes(Out& o, int kind, int arg0) {
  if (kind == 0) { const U8 h[9] = {7, 0, 0, 0, 0, 0, 0, 0, 0}; o.write(h, 9); return; }
  const U8 h[16] = {0x0e, 0, 9, 16, 0, (U8)(kind == 2 ? 20 + arg0 : 0), 0, 0, 0x12, 0x68, 0x87, 0xff, 0x58, 0x72, 0x38, 0};
  o.write(h, 16);
}

// compressBlock() for the stored/LZ77-level-1 family.  Reference declaration
// ZSFX/libzpaq.h:1505, usage doc :286-294 and :73-84 (comment = decimal size [+ " " + comment]).
//   method  : "0", "0<...>", "1", "1<B>[,R,t]" or "x<N1>,0" / "x<N1>,1,<mm>,0,<b>,<h>"
//   returns : framed block length, or <0 (-11 = method outside the restated family)
// E8E9 variants (args[1]=5) and rb>0 (arg0>4) need PCOMP programs that no fixture pins and
// are refused here.
extern "C" long orc_compress_block(const U8* in, long n, const char* method, const char* filename,
                                   const char* comment, int dosha1, U8* out, long cap, int args_out[9]) {
  std::string m(method);
  int args[9] = {0};
  const int arg0 = lg((U32)n + 4095) - 20 > 0 ? lg((U32)n + 4095) - 20 : 0;
  int kind;
  if (m[0] >= '0' && m[0] <= '9' && !(m[0] == '0' && m.size() == 1)) {
    // digit method "LB,R,t" -> type
    int commas = 0, a[4] = {0};
    for (size_t i = 1; i < m.size() && commas < 4; ++i) {
      if (m[i] == ',' || m[i] == '.') ++commas;
      else if (m[i] >= '0' && m[i] <= '9') a[commas] = a[commas] * 10 + m[i] - '0';
    }
    unsigned type = commas == 0 ? 512 : a[1] * 4 + a[2];
    int level = m[0] - '0';
    int htsz = 19 + arg0 + (arg0 <= 6);
    char b[64];
    if (level == 0) snprintf(b, sizeof b, "0%d,0", arg0);
    else if (level == 1) {
      if (type & 2) return -11;  // E8E9: unpinned PCOMP
      if (type < 40) snprintf(b, sizeof b, "x%d,0", arg0);
      else if (type < 80) snprintf(b, sizeof b, "x%d,1,4,0,1,15", arg0);
      else if (type < 128) snprintf(b, sizeof b, "x%d,1,4,0,2,16", arg0);
      else if (type < 256) snprintf(b, sizeof b, "x%d,1,4,0,2,%d", arg0, htsz);
      else if (type < 960) snprintf(b, sizeof b, "x%d,1,5,0,3,%d", arg0, htsz);
      else snprintf(b, sizeof b, "x%d,1,6,0,3,%d", arg0, htsz);
    } else return -11;
    m = b;
  }
  {  // makeConfig argument scan: "{x|0}N1,N2,..." -> args[0..8]
    const char* p = m.c_str() + 1; int i = 0;
    while (i < 9 && ((*p >= '0' && *p <= '9') || *p == ',' || *p == '.')) {
      if (*p >= '0' && *p <= '9') args[i] = args[i] * 10 + *p - '0'; else if (++i < 9) args[i] = 0;
      ++p;
    }
    if (*p) return -11;  // component list follows: context-mixing method, not restated here
  }
  if (m[0] == '0') kind = 0;
  else if (m[0] == 'x' && args[1] == 0) kind = 1;
  else if (m[0] == 'x' && args[1] == 1 && args[0] <= 4) kind = 2;
  else return -11;
  if (args_out) memcpy(args_out, args, sizeof(args));

  U8 digest[20];
  if (dosha1) orc_sha1(in, n, digest);

  std::vector<U8> payload;  // what the Encoder sees: postProcess preamble + data
  if (kind == 2) {
    payload.push_back(1); payload.push_back(302 & 255); payload.push_back(302 >> 8);
    payload.insert(payload.end(), kPcompLz1, kPcompLz1 + 302);
    std::vector<U8> lz((size_t)n + n / 8 + 1024);
    long k = orc_lz77_encode(in, n, args, lz.data(), (long)lz.size(), 0, 0, 0);
    if (k < 0) return k;
    payload.insert(payload.end(), lz.begin(), lz.begin() + k);
  } else {
    payload.push_back(0);
    payload.insert(payload.end(), in, in + n);
  }

  Out o;
  o.write(kTag, 13);                                   // writeTag
  o.put('z'); o.put('P'); o.put('Q'); o.put(2); o.put(1);  // startBlock: level 2 because n==0
  header_bytes(o, kind, args[0]);
  o.put(1);                                            // startSegment
  if (filename) o.write(filename, strlen(filename));
  o.put(0);
  char sz[32]; snprintf(sz, sizeof sz, "%ld", n);
  o.write(sz, strlen(sz));
  if (comment) { o.put(' '); o.write(comment, strlen(comment)); }
  o.put(0); o.put(0);
  stored_subblocks(o, payload);                        // compress()
  o.put(0); o.put(0); o.put(0); o.put(0);              // endSegment
  if (dosha1) { o.put(253); o.write(digest, 20); } else o.put(254);
  o.put(255);                                          // endBlock
  if ((long)o.v.size() > cap) return -2;
  memcpy(out, o.v.data(), o.v.size());
  return (long)o.v.size();
}

// Inverse of the above for one n=0 block at arc[0]: parses the framing as Decompresser does
// (ZSFX/libzpaq.cpp:2239-2366), concatenates stored sub-blocks, strips the post-processor
// preamble, and undoes LZ77 level 1 when the embedded PCOMP equals the golden program.
// meta: [0]=bytes consumed, [1]=1 stored SHA-1 matched / 0 mismatch / 2 absent, [2]=kind(0 pass,2 lz1)
extern "C" long orc_decompress_block(const U8* arc, long n, U8* out, long cap, long meta[3]) {
  long p = 0;
  if (n < 13 + 5 || memcmp(arc, kTag, 13)) return -20;
  p = 13;
  if (arc[p] != 'z' || arc[p + 1] != 'P' || arc[p + 2] != 'Q' || arc[p + 4] != 1) return -21;
  p += 5;
  int hsize = arc[p] | arc[p + 1] << 8;
  if (arc[p + 6] != 0) return -22;  // n components > 0: not a stored block
  int pm = arc[p + 5]; (void)pm;
  p += 2 + hsize;
  if (arc[p++] != 1) return -23;
  while (arc[p]) ++p; ++p;          // filename
  while (arc[p]) ++p; ++p;          // comment
  if (arc[p++] != 0) return -24;    // reserved
  std::vector<U8> payload;
  while (true) {
    U32 k = (U32)arc[p] << 24 | (U32)arc[p + 1] << 16 | (U32)arc[p + 2] << 8 | arc[p + 3];
    p += 4;
    if (!k) break;
    payload.insert(payload.end(), arc + p, arc + p + k); p += k;
  }
  long r;
  if (payload.empty()) return -25;
  if (payload[0] == 0) {
    r = (long)payload.size() - 1;
    if (r > cap) return -2;
    memcpy(out, payload.data() + 1, r); meta[2] = 0;
  } else {
    int psize = payload[1] | payload[2] << 8;
    if (psize != 302 || memcmp(&payload[3], kPcompLz1, 302)) return -26;
    r = orc_lz77_decode(&payload[3 + psize], (long)payload.size() - 3 - psize, 0, out, cap);
    if (r < 0) return r;
    meta[2] = 2;
  }
  meta[1] = 2;
  if (arc[p] == 253) {
    U8 d[20]; orc_sha1(out, r, d);
    meta[1] = memcmp(d, arc + p + 1, 20) == 0; p += 21;
  } else if (arc[p] == 254) ++p; else return -27;
  if (arc[p++] != 255) return -28;
  meta[0] = p;
  return r;
}

// The fragment loop of Jidac::add in one call: fragment buf[0..n) and SHA-1
// every fragment, entirely in C so that a thread pool can run it without Python in the way
// (bench.py cpu_baseline).  Returns the fragment count; *digest_xor receives the XOR of the first 8
// digest bytes (keeps the work observable).
extern "C" long orc_fragment_and_hash(const U8* buf, long n, int fragment, U32 min_frag, U32 max_frag, U64* digest_xor) {
  long nf = 0, i = 0; U64 acc = 0;
  while (i < n) {
    U32 h = 0, sz = 0; U8 o1[256]; memset(o1, 0, 256); unsigned c1 = 0;
    const long start = i;
    Sha1 s;
    while (i < n) {
      unsigned c = buf[i++];
      if (c == o1[c1]) h = (h + c + 1) * 314159265u; else h = (h + c + 1) * 271828182u;
      o1[c1] = (U8)c; c1 = c; ++sz;
      if (sz >= max_frag || (fragment <= 22 && h < (1u << (22 - fragment)) && sz >= min_frag)) break;
    }
    s.update(buf + start, (size_t)(i - start));
    U8 d[20]; s.final(d);
    U64 v; memcpy(&v, d, 8); acc ^= v;
    ++nf;
  }
  if (digest_xor) *digest_xor = acc;
  return nf;
}
