// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" window onto the REAL reference code.  The reference translation unit
// /root/reference/ZSFX/libzpaq.cpp is compiled *where it lies* by including it here
// (LZBuffer, e8e9, lg and divsufsort are file-local to it, so a separate TU cannot reach
// them).  Nothing from the reference is copied into this repository; the build product
// goes to oracle/_ref/ (git-ignored, but shipped to the GPU box by gpurun).
//
// Build: see oracle/Makefile (target _ref/libzpaqref.so).  Requires /root/reference.
#define private public     // test shim only: lets ref_tables() read the Predictor's lookup tables
#include "libzpaq.cpp"   // resolved through -I/root/reference/ZSFX
#undef private

#include <stdexcept>
#include <string>
#include <vector>

namespace libzpaq {
// ZSFX/libzpaq.h:54-59,858: the application supplies error(); it must not return.
void error(const char* msg) { throw std::runtime_error(msg ? msg : "libzpaq error"); }
}

namespace {
struct MemReader : libzpaq::Reader {
  const unsigned char* p; size_t n, i;
  MemReader(const void* q, size_t len) : p((const unsigned char*)q), n(len), i(0) {}
  int get() { return i < n ? p[i++] : -1; }
  int read(char* buf, int k) {
    size_t r = n - i; if ((size_t)k < r) r = k;
    memcpy(buf, p + i, r); i += r; return (int)r;
  }
};
struct VecWriter : libzpaq::Writer {
  std::vector<unsigned char> v;
  void put(int c) { v.push_back((unsigned char)c); }
  void write(const char* buf, int n) { v.insert(v.end(), buf, buf + n); }
};
thread_local std::string g_err;
template <class F> long guarded(F f) {
  try { return f(); } catch (std::exception& e) { g_err = e.what(); return -1; }
}
long emit(const std::vector<unsigned char>& v, unsigned char* out, long cap) {
  if ((long)v.size() > cap) { g_err = "output buffer too small"; return -2; }
  if (!v.empty()) memcpy(out, v.data(), v.size());
  return (long)v.size();
}
}  // namespace

extern "C" {

const char* ref_last_error() { return g_err.c_str(); }

// libzpaq::SHA1::write + result (ZSFX/libzpaq.cpp:96-167)
void ref_sha1(const unsigned char* buf, long n, unsigned char out20[20]) {
  libzpaq::SHA1 s; s.write((const char*)buf, n); memcpy(out20, s.result(), 20);
}
// libzpaq::SHA256::put + result (ZSFX/libzpaq.cpp:171-304)
void ref_sha256(const unsigned char* buf, long n, unsigned char out32[32]) {
  libzpaq::SHA256 s; for (long i = 0; i < n; ++i) s.put(buf[i]); memcpy(out32, s.result(), 32);
}
// e8e9 (ZSFX/libzpaq.cpp:6117-6126), in place
void ref_e8e9(unsigned char* buf, int n) { libzpaq::e8e9(buf, n); }
int ref_lg(unsigned x) { return libzpaq::lg(x); }

// LZBuffer (ZSFX/libzpaq.cpp:6140-6552): run the reference LZ77/BWT front end over in[0..n)
// with args[0..8] exactly as compressBlock would, and return the raw code stream.
long ref_lzbuffer(const unsigned char* in, long n, const int args9[9], unsigned char* out, long cap) {
  return guarded([&]() -> long {
    libzpaq::StringBuffer sb;
    sb.write((const char*)in, (int)n);
    int args[9]; memcpy(args, args9, sizeof(args));
    libzpaq::LZBuffer lz(sb, args);
    std::vector<unsigned char> v; v.reserve(n / 2 + 64);
    char tmp[1 << 14]; int r;
    while ((r = lz.read(tmp, sizeof(tmp))) > 0) v.insert(v.end(), tmp, tmp + r);
    return emit(v, out, cap);
  });
}

// divsufsort (ZSFX/libzpaq.cpp:6047-6072): the suffix array LZBuffer builds for LZ77-SA / BWT
long ref_divsufsort(const unsigned char* in, long n, int* sa) {
  return guarded([&]() -> long { return n > 0 ? (long)libzpaq::divsufsort(in, sa, (int)n) : 0; });
}

// libzpaq::decompress (ZSFX/libzpaq.cpp:2368-2381): all blocks/segments concatenated.
long ref_decompress(const unsigned char* arc, long n, unsigned char* out, long cap) {
  return guarded([&]() -> long {
    MemReader in(arc, n); VecWriter w;
    libzpaq::decompress(&in, &w);
    return emit(w.v, out, cap);
  });
}

// Decompress ONE block starting the search at arc[0]; reports what the reference parsed.
// Follows the loop of decompressThread (ZSFX/zsfx.cpp:1783-1834) with the reference
// Decompresser (ZSFX/libzpaq.cpp:2239-2366).
// meta: [0]=bytes consumed from arc, [1]=#segments, [2]=1 if every stored SHA-1 matched
//       (0 if any mismatch, 2 if no segment carried a checksum), [3]=filename length,
//       [4]=comment length.  fn/cm receive the FIRST segment's filename/comment.
long ref_decompress_block(const unsigned char* arc, long n, unsigned char* out, long cap,
                          long meta[5], char* fn, long fncap, char* cm, long cmcap,
                          unsigned char sha1_first[21]) {
  return guarded([&]() -> long {
    MemReader in(arc, n); VecWriter w;
    libzpaq::Decompresser d; d.setInput(&in);
    if (!d.findBlock()) { g_err = "no block"; return -3; }
    long segs = 0; int ok = 2;
    libzpaq::StringBuffer f, c;
    while (true) {
      libzpaq::StringBuffer f1, c1;
      if (!d.findFilename(&f1)) break;
      d.readComment(&c1);
      if (segs == 0) { f.swap(f1); c.swap(c1); }
      libzpaq::SHA1 sha; d.setSHA1(&sha); d.setOutput(&w);
      d.decompress();
      char s[21]; d.readSegmentEnd(s);
      if (segs == 0 && sha1_first) memcpy(sha1_first, s, 21);
      if (s[0]) { const char* r = sha.result(); bool m = memcmp(s + 1, r, 20) == 0;
                  if (!m) ok = 0; else if (ok == 2) ok = 1; }
      ++segs;
    }
    meta[0] = (long)(in.i - d.buffered()); meta[1] = segs; meta[2] = ok;
    meta[3] = (long)f.size(); meta[4] = (long)c.size();
    if ((long)f.size() < fncap) { memcpy(fn, f.c_str() ? f.c_str() : "", f.size()); fn[f.size()] = 0; }
    if ((long)c.size() < cmcap) { memcpy(cm, c.c_str() ? c.c_str() : "", c.size()); cm[c.size()] = 0; }
    return emit(w.v, out, cap);
  });
}

// Compiler (ZSFX/libzpaq.cpp:2430-2706): ZPAQL source -> COMP/HCOMP header bytes (as
// ZPAQL::write(out,false) serialises them, :858-876) and PCOMP bytes (write(out,true)).
long ref_compile(const char* config, const int args9[9], unsigned char* hcomp, long hcap,
                 long* hlen, unsigned char* pcomp, long pcap, long* plen) {
  return guarded([&]() -> long {
    int args[9]; memcpy(args, args9, sizeof(args));
    libzpaq::ZPAQL hz, pz; VecWriter cmd;
    libzpaq::Compiler comp(config, args, hz, pz, &cmd);
    VecWriter h, p;
    hz.write(&h, false);
    bool has_p = pz.write(&p, true);
    long a = emit(h.v, hcomp, hcap); if (a < 0) return a; *hlen = a;
    *plen = 0;
    if (has_p) { long b = emit(p.v, pcomp, pcap); if (b < 0) return b; *plen = b; }
    return 0;
  });
}

// Run a PCOMP program (bytes as stored in a block: psize[2] + code) over a decoded byte
// stream with the reference ZPAQL VM via PostProcessor (ZSFX/libzpaq.cpp:2178-2233).
// stream must begin with the 0 (PASS) / 1 psize pcomp preamble exactly as decoded.
long ref_postprocess(const unsigned char* stream, long n, int ph, int pm, unsigned char* out, long cap) {
  return guarded([&]() -> long {
    libzpaq::PostProcessor pp; VecWriter w;
    pp.init(ph, pm); pp.setOutput(&w); pp.setSHA1(0);
    for (long i = 0; i < n; ++i) pp.write(stream[i]);
    pp.write(-1);
    return emit(w.v, out, cap);
  });
}

// Context-mixing encode with the REFERENCE Predictor (ZSFX/libzpaq.cpp:1715-2080, JIT or
// interpreter) driven by the mirror of Decoder::decode (ZSFX/libzpaq.cpp:2096-2147).
// header = block header bytes starting at hsize (as ZPAQL::read expects, :879-921).
// Emits the arithmetic-coded payload including the 4 trailing zero bytes' worth of flush
// exactly as Decoder expects them (EOS symbol then 0 0 0 0 written by endSegment).
long ref_cm_encode(const unsigned char* header, long hlen, const unsigned char* data, long n,
                   unsigned char* out, long cap) {
  return guarded([&]() -> long {
    MemReader hr(header, hlen);
    libzpaq::ZPAQL z; z.read(&hr);
    libzpaq::Predictor pr(z); pr.init();
    std::vector<unsigned char> v; v.reserve(n / 2 + 64);
    libzpaq::U32 low = 1, high = 0xFFFFFFFFu;
    auto encode = [&](int y, int p) {
      libzpaq::U32 mid = low + libzpaq::U32((libzpaq::U64(high - low) * libzpaq::U32(p)) >> 16);
      if (y) high = mid; else low = mid + 1;
      while ((high ^ low) < 0x1000000u) {
        v.push_back((unsigned char)(high >> 24));
        high = high << 8 | 255; low = low << 8; low += (low == 0);
      }
    };
    for (long i = 0; i < n; ++i) {
      encode(0, 0);
      int c = data[i];
      for (int b = 7; b >= 0; --b) {
        int p = pr.predict() * 2 + 1; int y = (c >> b) & 1;
        encode(y, p); pr.update(y);
      }
    }
    encode(1, 0);
    v.push_back(0); v.push_back(0); v.push_back(0); v.push_back(0);
    return emit(v, out, cap);
  });
}

// Several segments of ONE block (Compressor::startSegment ... endSegment more than once before endBlock): the real Predictor
// carries on from segment to segment, the coder ends each with its end-of-segment symbol and the four 0 bytes (the byte
// layout Decompresser::decompress / Decoder::decompress read back, ZSFX/libzpaq.cpp:2116-2137, 2307-2337).  data = the
// segments' bytes back to back, seg_len[s] bytes each; out_end[s] = coded bytes up to and including segment s.
long ref_cm_encode_segments(const unsigned char* header, long hlen, const unsigned char* data, const unsigned* seg_len, long nseg,
                            unsigned char* out, long cap, unsigned* out_end) {
  return guarded([&]() -> long {
    MemReader hr(header, hlen);
    libzpaq::ZPAQL z; z.read(&hr);
    libzpaq::Predictor pr(z); pr.init();
    std::vector<unsigned char> v;
    libzpaq::U32 low = 1, high = 0xFFFFFFFFu;
    auto encode = [&](int y, int p) {
      libzpaq::U32 mid = low + libzpaq::U32((libzpaq::U64(high - low) * libzpaq::U32(p)) >> 16);
      if (y) high = mid; else low = mid + 1;
      while ((high ^ low) < 0x1000000u) {
        v.push_back((unsigned char)(high >> 24));
        high = high << 8 | 255; low = low << 8; low += (low == 0);
      }
    };
    long at = 0;
    for (long s = 0; s < nseg; ++s) {
      for (long i = 0; i < (long)seg_len[s]; ++i) {
        encode(0, 0);
        int c = data[at + i];
        for (int b = 7; b >= 0; --b) {
          int p = pr.predict() * 2 + 1; int y = (c >> b) & 1;
          encode(y, p); pr.update(y);
        }
      }
      at += seg_len[s];
      encode(1, 0);
      v.push_back(0); v.push_back(0); v.push_back(0); v.push_back(0);
      out_end[s] = (unsigned)v.size();
    }
    return emit(v, out, cap);
  });
}

// The reverse with the real Decoder: coded = the segments' coded streams back to back; one Decoder (init once, as
// Decompresser::decompress does for the first segment only) decodes segment after segment.
long ref_cm_decode_segments(const unsigned char* header, long hlen, const unsigned char* coded, long n, long nseg,
                            unsigned char* out, long cap, unsigned* out_end) {
  return guarded([&]() -> long {
    MemReader hr(header, hlen);
    libzpaq::ZPAQL z; z.read(&hr);
    libzpaq::Decoder dec(z);
    MemReader in(coded, n);
    dec.in = &in;
    dec.init();
    std::vector<unsigned char> v;
    for (long s = 0; s < nseg; ++s) {
      int c;
      while ((c = dec.decompress()) >= 0) v.push_back((unsigned char)c);
      out_end[s] = (unsigned)v.size();
    }
    return emit(v, out, cap);
  });
}

// The model-independent lookup tables exactly as the reference Predictor holds them after init()
// (ZSFX/libzpaq.cpp:1724-1742, sources :718-847 and :1264-1695): squash[4096], stretch[32768],
// dt[1024], dt2k[256], and the bit-history next-state table ns[1024].
long ref_tables(unsigned short* squash, short* stretch, int* dt, int* dt2k, unsigned char* ns) {
  return guarded([&]() -> long {
    const unsigned char hdr[12] = {10, 0, 0, 0, 0, 0, 1, 1, 128, 0, 56, 0};   // comp 0 0 0 0 1 (cons 128) hcomp halt
    MemReader hr(hdr, sizeof hdr);
    libzpaq::ZPAQL z; z.read(&hr);
    libzpaq::Predictor pr(z); pr.init();
    memcpy(squash, pr.squasht, sizeof pr.squasht);
    memcpy(stretch, pr.stretcht, sizeof pr.stretcht);
    memcpy(dt, pr.dt, sizeof pr.dt);
    memcpy(dt2k, pr.dt2k, sizeof pr.dt2k);
    memcpy(ns, pr.st.ns, 1024);
    return 0;
  });
}

// CPU baseline of the add path's first stage, one file: the fragment loop (the reference's loop lives in the missing
// zpaqfranz.cpp; restated from SURVEY.md Appendix C.4, as in zpaq_oracle.cpp orc_chunk) feeding the REAL libzpaq::SHA1
// (ZSFX/libzpaq.cpp:96-167) byte by byte exactly as Jidac::add does.  Returns the fragment count; *digest_xor = xor of
// the first 8 bytes of every fragment SHA-1 (so that the work cannot be optimised away and runs can be compared).
long ref_fragment_sha1(const unsigned char* buf, long n, int fragment, unsigned min_frag, unsigned max_frag, unsigned long long* digest_xor) {
  long nf = 0; unsigned long long x = 0;
  long i = 0;
  while (i < n) {
    libzpaq::SHA1 sha;
    unsigned h = 0, c1 = 0; unsigned char o1[256]; memset(o1, 0, sizeof o1);
    unsigned sz = 0;
    while (i < n) {
      const unsigned c = buf[i++];
      if (c == o1[c1]) h = (h + c + 1) * 314159265u; else h = (h + c + 1) * 271828182u;
      o1[c1] = (unsigned char)c; c1 = c; sha.put((int)c); ++sz;
      if (sz >= max_frag || (fragment <= 22 && h < (1u << (22 - fragment)) && sz >= min_frag)) break;
    }
    unsigned long long d; memcpy(&d, sha.result(), 8); x ^= d; ++nf;
  }
  if (digest_xor) *digest_xor = x;
  return nf;
}

// CPU baseline of compressBlock("14"-class): the REAL LZBuffer over the block plus the REAL SHA1 of the block (what
// libzpaq::compressBlock costs for the stored-LZ77 family; the stored framing itself is a memcpy).  Returns the
// size of the code stream.
long ref_lz1_block_cost(const unsigned char* in, long n, const int args9[9]) {
  return guarded([&]() -> long {
    libzpaq::SHA1 sha; sha.write((const char*)in, n); (void)sha.result();
    libzpaq::StringBuffer sb;
    sb.write((const char*)in, (int)n);
    int args[9]; memcpy(args, args9, sizeof(args));
    libzpaq::LZBuffer lz(sb, args);
    long total = 0; char tmp[1 << 14]; int r;
    while ((r = lz.read(tmp, sizeof(tmp))) > 0) total += r;
    return total;
  });
}

// Context-mixing DECODE of one coded segment with the reference Decoder/Predictor: header as for
// ref_cm_encode, coded = arithmetic-coded bytes including the four trailing zero bytes.
long ref_cm_decode(const unsigned char* header, long hlen, const unsigned char* coded, long n,
                   unsigned char* out, long cap) {
  return guarded([&]() -> long {
    MemReader hr(header, hlen);
    libzpaq::ZPAQL z; z.read(&hr);
    libzpaq::Decoder dec(z);
    MemReader in(coded, n);
    dec.in = &in;
    dec.init();
    std::vector<unsigned char> v;
    int c;
    while ((c = dec.decompress()) >= 0) v.push_back((unsigned char)c);
    return emit(v, out, cap);
  });
}

}  // extern "C"
