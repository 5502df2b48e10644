// libzpaq_gpu.h -- the reference's own libzpaq API (namespace libzpaq), served by the MI355X engine.
//
// Same names, argument meaning, ownership and error behaviour as ZSFX/libzpaq.h of the reference:
//   error()          ZSFX/libzpaq.h:858   application-defined, must not return
//   Reader / Writer  ZSFX/libzpaq.h:864-876
//   SHA1 / SHA256    ZSFX/libzpaq.h:934-979 (put/write/size/usize/result; result() resets)
//   StringBuffer     ZSFX/libzpaq.h:1377-1494
//   compressBlock()  ZSFX/libzpaq.h:1505, compress() :1501, decompress() :1268
//   Decompresser     ZSFX/libzpaq.h:1243-1264
//   Compressor       ZSFX/libzpaq.h:1340-1371 (streaming: startBlock / startSegment / compress(n) / endSegment / endBlock)
// so that a Jidac-style caller (one compressBlock per block per worker thread; one Decompresser per block on the
// extract side, ZSFX/zsfx.cpp:1783-1801) links against this header unchanged.
//
// What runs where: the byte work (LZ77 levels 1 / 2, BWT, E8E9, the context-mixing coder, stored framing, block
// SHA-1 and their inverses) runs in HIP kernels behind the C ABI of include/zpaqhip.h.  Method strings outside the
// family the engine implements (a secondary LZ77 context, BWT+E8E9 above 16 MiB, more than 255
// components) end in libzpaq::error("...") -- there is no CPU fallback compiled into this library; the host keeps
// its CPU libzpaq for those if it wants.
//
// Batching: compressBlock(), Decompresser::decompress() and SHA1/SHA256::result() block the calling thread like
// the reference's do.  Calls made concurrently from N worker threads (zpaqfranz -tN) are coalesced by small
// batchers into one zpq_compress_blocks() / zpq_decompress_blocks() / zpq_sha*_many() launch each, which is how a
// per-block API feeds a GPU.  A caller that can hand over many buffers at once uses sha1_many() below.
#ifndef LIBZPAQ_GPU_H
#define LIBZPAQ_GPU_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

namespace libzpaq {

typedef uint8_t U8;
typedef uint16_t U16;
typedef uint32_t U32;
typedef uint64_t U64;

extern void error(const char* msg);  // supplied by the application (ZSFX/zsfx.cpp:129-135)

class Reader {
 public:
  virtual int get() = 0;
  virtual int read(char* buf, int n);
  virtual ~Reader() {}
};

class Writer {
 public:
  virtual void put(int c) = 0;
  virtual void write(const char* buf, int n);
  virtual ~Writer() {}
};

// SHA-1 / SHA-256 accumulate on the host side and hash on the GPU at result() time (the hot use in
// zpaqfranz is whole fragments/blocks; byte-at-a-time put() only buffers).
class SHA1 {
 public:
  SHA1() : p(0), n(0), cap(0) {}
  ~SHA1() { free(p); }
  void put(int c) { grow(1); p[n++] = (unsigned char)c; }
  void write(const char* buf, int64_t len) { if (len > 0) { grow((size_t)len); memcpy(p + n, buf, (size_t)len); n += (size_t)len; } }
  double size() const { return (double)n; }
  uint64_t usize() const { return n; }
  const char* result();  // 20 bytes; resets
 private:
  void grow(size_t k) { if (n + k > cap) { cap = (n + k) * 2 + 64; p = (unsigned char*)realloc(p, cap); if (!p) error("Out of memory"); } }
  unsigned char* p; size_t n, cap; char hbuf[20];
  SHA1(const SHA1&); void operator=(const SHA1&);
};

class SHA256 {
 public:
  SHA256() : p(0), n(0), cap(0) {}
  ~SHA256() { free(p); }
  void put(int c) { if (n + 1 > cap) { cap = (n + 1) * 2 + 64; p = (unsigned char*)realloc(p, cap); if (!p) error("Out of memory"); } p[n++] = (unsigned char)c; }
  double size() const { return (double)n; }
  uint64_t usize() const { return n; }
  const char* result();  // 32 bytes; resets
 private:
  unsigned char* p; size_t n, cap; char hbuf[32];
  SHA256(const SHA256&); void operator=(const SHA256&);
};

class StringBuffer : public Reader, public Writer {
  unsigned char* p; size_t al, wpos, rpos, limit; const size_t init;
  void reserve(size_t a) { if (a <= al) return; unsigned char* q = (unsigned char*)(p ? realloc(p, a) : malloc(a)); if (a > 0 && !q) error("Out of memory"); p = q; al = a; }
  void lengthen(size_t n) { if (wpos + n > limit || wpos + n < wpos) error("StringBuffer overflow"); if (wpos + n <= al) return; size_t a = al; while (wpos + n >= a) a = a * 2 + init; reserve(a); }
  void operator=(const StringBuffer&); StringBuffer(const StringBuffer&);
 public:
  unsigned char* data() { return p; }
  StringBuffer(size_t n = 0) : p(0), al(0), wpos(0), rpos(0), limit(size_t(-1)), init(n > 128 ? n : 128) {}
  void setLimit(size_t n) { limit = n; }
  ~StringBuffer() { if (p) free(p); }
  size_t size() const { return wpos; }
  size_t remaining() const { return wpos - rpos; }
  void reset() { if (p) free(p); p = 0; al = rpos = wpos = 0; }
  void put(int c) { lengthen(1); p[wpos++] = (unsigned char)c; }
  void write(const char* buf, int n) { if (n < 1) return; lengthen(n); if (buf) memcpy(p + wpos, buf, n); wpos += n; }
  int get() { return rpos < wpos ? p[rpos++] : -1; }
  int read(char* buf, int n) { if (rpos + n > wpos) n = (int)(wpos - rpos); if (n > 0 && buf) memcpy(buf, p + rpos, n); rpos += n; return n; }
  const char* c_str() const { return (const char*)p; }
  void resize(size_t i) { wpos = i; if (rpos > wpos) rpos = wpos; }
  void swap(StringBuffer& s) { unsigned char* t = p; p = s.p; s.p = t; size_t x; x = al; al = s.al; s.al = x; x = wpos; wpos = s.wpos; s.wpos = x; x = rpos; rpos = s.rpos; s.rpos = x; x = limit; limit = s.limit; s.limit = x; }
};

// Compress in to out as ONE block (ZSFX/libzpaq.h:1505).  in is emptied, as the reference does.
void compressBlock(StringBuffer* in, Writer* out, const char* method, const char* filename = 0,
                   const char* comment = 0, bool dosha1 = true);

// Compress in to out in multiple blocks of the size the method names (ZSFX/libzpaq.h:1501):
// filename on the first block only, comment = decimal size + comment.
void compress(Reader* in, Writer* out, const char* method, const char* filename = 0, const char* comment = 0,
              bool dosha1 = true);

// Decompress every block of in to out (ZSFX/libzpaq.h:1268); stored SHA-1s are verified.
void decompress(Reader* in, Writer* out);

// Block-at-a-time reader with the reference's interface and call sequence (ZSFX/libzpaq.h:1243-1264; used as in
// ZSFX/zsfx.cpp:1783-1801: setInput, setOutput, findBlock, {findFilename, readComment, decompress(n)*, readSegmentEnd}*).
// The segment is decoded by the engine on the first decompress() call and then handed out n bytes at a time.
// Blocks with several segments (streaming archives with more than one file per block; the journaling format zpaqfranz
// writes has one segment per block): without a model and without a post-processor program every segment is a copy of
// its stored bytes; otherwise the later segments continue the first one's model / PCOMP machine (ZSFX/libzpaq.cpp:
// 2307-2337), so the WHOLE block is read from the Reader and decoded by one device job when its first segment is
// decompressed, and the later segments are handed out from that result (decompressing one after skipping the first
// ends in error("decompression after skipped segment"), as in the reference).
class Decompresser {
 public:
  Decompresser();
  ~Decompresser();
  void setInput(Reader* in);
  bool findBlock(double* memptr = 0);       // false at end of input
  void hcomp(Writer* out2);                 // the block header (hsize .. HCOMP END), as stored
  bool findFilename(Writer* filename = 0);  // false at the end of the block
  void readComment(Writer* comment = 0);
  void setOutput(Writer* out) { out_ = out; }
  void setSHA1(SHA1* sha1ptr) { sha_ = sha1ptr; }
  bool decompress(int n = -1);              // n bytes (-1 = all); false once the segment is exhausted
  bool pcomp(Writer* out2);                 // the PCOMP section (size, program); behind a context model the head of the stream is decoded for it
  void readSegmentEnd(char* sha1string = 0);// [0] = 1 if a SHA-1 follows in [1..20], else 0
  int stat(int) { return 0; }               // the reference reports predictor statistics here (debug builds only)
  int buffered();                           // bytes read from the Reader and not consumed yet (the look-ahead of decompress())
 private:
  struct Impl;
  Impl* d_;
  Reader* in_; Writer* out_; SHA1* sha_;
  Decompresser(const Decompresser&); void operator=(const Decompresser&);
};

// Streaming writer with the reference's interface (ZSFX/libzpaq.h:1340-1371, contract :426-531): what
// libzpaq::compress() and callers with their own block structure use.  Without a model a segment's bytes are written
// as stored sub-blocks when it ends.  With a model the block's segments are collected on the host and coded by ONE
// device job at endBlock() -- a later segment continues the first one's model (the Encoder's Predictor is initialised
// by startBlock only) -- so nothing of such a block reaches the Writer before endBlock(); then it is byte for byte what
// the reference's Compressor writes.
class Compressor {
 public:
  Compressor();
  ~Compressor();
  void setOutput(Writer* out) { out_ = out; }
  void writeTag();
  void startBlock(int level);                 // libzpaq's built-in models: 1 = min.cfg, 2 = mid.cfg, 3 = max.cfg (zpq_builtin_model)
  void startBlock(const char* hcomp);         // ZPAQL byte code, starting at hsize[2]
  void startBlock(const char* config, int* args, Writer* pcomp_cmd = 0);   // ZPAQL source
  void setVerify(bool) {}
  void hcomp(Writer* out2);
  bool pcomp(Writer* out2);
  void startSegment(const char* filename = 0, const char* comment = 0);
  void setInput(Reader* i) { in_ = i; }
  void postProcess(const char* pcomp = 0, int len = 0);
  bool compress(int n = -1);                  // n bytes, -1 = all; true until the input is exhausted
  void endSegment(const char* sha1string = 0);
  char* endSegmentChecksum(int64_t* size = 0, bool dosha1 = true);
  int64_t getSize() { return (int64_t)sha1_.usize(); }
  const char* getChecksum() { return sha1_.result(); }
  void endBlock();
  int stat(int) { return 0; }
 private:
  struct Impl;
  Impl* d_;
  Writer* out_; Reader* in_;
  SHA1 sha1_;
  char sha1result_[20];
  Compressor(const Compressor&); void operator=(const Compressor&);
};

// Many independent buffers -> many SHA-1s in one launch (not in the reference: the three-line change a caller makes in
// the fragment verification loop of decompressThread, ZSFX/zsfx.cpp:1811-1834).
void sha1_many(const char* const* bufs, const size_t* lens, size_t n, char* digests /* 20*n */);

// Engine plumbing (not in the reference): which GPU this process uses; call before first use.
void setDevice(int ordinal);

}  // namespace libzpaq
#endif
