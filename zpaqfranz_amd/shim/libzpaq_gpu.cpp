// libzpaq_gpu.cpp -- see libzpaq_gpu.h.  Host-side C++ above the C ABI (include/zpaqhip.h).
#include "libzpaq_gpu.h"

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "zpaqhip.h"

namespace libzpaq {

int Reader::read(char* buf, int n) { int i = 0, c; while (i < n && (c = get()) >= 0) buf[i++] = (char)c; return i; }
void Writer::write(const char* buf, int n) { for (int i = 0; i < n; ++i) put(U8(buf[i])); }

namespace {

int g_device = 0;
const uint8_t kTag[13] = {0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3};

struct EngineHolder {
  zpq_ctx* ctx = nullptr;
  std::mutex mu;   // a context serves one thread at a time (include/zpaqhip.h)
  zpq_ctx* get() {
    if (!ctx) {
      int rc = zpq_create(g_device, &ctx);
      if (rc != ZPQ_OK) { std::string m = std::string("zpaqhip: ") + zpq_strerror(rc); error(m.c_str()); }
    }
    return ctx;
  }
};
EngineHolder& engine() { static EngineHolder e; return e; }

void fail(zpq_ctx* ctx, int rc, const char* what) {
  std::string m = std::string(what) + ": " + zpq_strerror(rc);
  if (ctx) { m += " ("; m += zpq_last_error(ctx); m += ")"; }
  error(m.c_str());
}

// Coalesces blocking calls from concurrent worker threads into one engine call: the first caller becomes the leader,
// runs the batch that has gathered (its own job included) and wakes the others; callers that arrive meanwhile form
// the next batch.  Run(ctx, jobs, n) is the zpq_* batch entry point.
template <class Job, int (*Run)(zpq_ctx*, Job*, size_t)>
struct Batcher {
  struct Item { Job job; bool done; int rc; };
  std::mutex mu;
  std::condition_variable cv;
  std::vector<Item*> queue;
  bool leader_active = false;

  void submit(Item* it) {
    std::unique_lock<std::mutex> lk(mu);
    queue.push_back(it);
    if (leader_active) { cv.wait(lk, [&] { return it->done; }); return; }
    leader_active = true;
    while (!queue.empty()) {
      std::vector<Item*> batch;
      batch.swap(queue);
      lk.unlock();
      std::this_thread::yield();   // let sibling threads that are about to submit join the next batch
      run(batch);
      lk.lock();
      for (Item* b : batch) b->done = true;
      cv.notify_all();
    }
    leader_active = false;
  }
  void run(std::vector<Item*>& batch) {
    EngineHolder& e = engine();
    std::lock_guard<std::mutex> g(e.mu);
    zpq_ctx* ctx = e.get();
    std::vector<Job> jobs(batch.size());
    for (size_t i = 0; i < batch.size(); ++i) jobs[i] = batch[i]->job;
    const int rc = Run(ctx, jobs.data(), jobs.size());   // per-job status carries the outcome where the job has one
    for (size_t i = 0; i < batch.size(); ++i) { batch[i]->job = jobs[i]; batch[i]->rc = rc; }
  }
};

int run_compress(zpq_ctx* c, zpq_block_job* j, size_t n) { return zpq_compress_blocks(c, j, n); }
int run_decompress(zpq_ctx* c, zpq_unblock_job* j, size_t n) { return zpq_decompress_blocks(c, j, n, 0); }   // callers verify through setSHA1 / their own table
struct HashJob { const uint8_t* p; size_t n; uint8_t* out; };
int run_sha1(zpq_ctx* c, HashJob* j, size_t n) {
  std::vector<const uint8_t*> b(n); std::vector<size_t> l(n); std::vector<uint8_t> d(n * 20);
  for (size_t i = 0; i < n; ++i) { b[i] = j[i].p; l[i] = j[i].n; }
  const int rc = zpq_sha1_many(c, b.data(), l.data(), n, d.data());
  for (size_t i = 0; i < n; ++i) memcpy(j[i].out, &d[20 * i], 20);
  return rc;
}
int run_sha256(zpq_ctx* c, HashJob* j, size_t n) {
  std::vector<const uint8_t*> b(n); std::vector<size_t> l(n); std::vector<uint8_t> d(n * 32);
  for (size_t i = 0; i < n; ++i) { b[i] = j[i].p; l[i] = j[i].n; }
  const int rc = zpq_sha256_many(c, b.data(), l.data(), n, d.data());
  for (size_t i = 0; i < n; ++i) memcpy(j[i].out, &d[32 * i], 32);
  return rc;
}
typedef Batcher<zpq_block_job, run_compress> CompressBatcher;
typedef Batcher<zpq_unblock_job, run_decompress> DecompressBatcher;
typedef Batcher<HashJob, run_sha1> Sha1Batcher;
typedef Batcher<HashJob, run_sha256> Sha256Batcher;
CompressBatcher& compress_batcher() { static CompressBatcher b; return b; }
DecompressBatcher& decompress_batcher() { static DecompressBatcher b; return b; }
Sha1Batcher& sha1_batcher() { static Sha1Batcher b; return b; }
Sha256Batcher& sha256_batcher() { static Sha256Batcher b; return b; }

// decimal size at the start of a comment (compressBlock's contract, ZSFX/libzpaq.h:73-84); 0 when absent
size_t comment_size(const uint8_t* blk, size_t n) {
  size_t q = 13 + 5;
  if (q + 2 > n) return 0;
  q += 2 + (blk[q] | (size_t)blk[q + 1] << 8) + 1;
  while (q < n && blk[q]) ++q;
  ++q;
  size_t v = 0;
  while (q < n && blk[q] >= '0' && blk[q] <= '9') { v = v * 10 + (size_t)(blk[q] - '0'); ++q; }
  return v;
}

}  // namespace

void setDevice(int ordinal) { g_device = ordinal; }

const char* SHA1::result() {
  Sha1Batcher::Item it;
  it.job.p = p ? p : (const uint8_t*)""; it.job.n = n; it.job.out = (uint8_t*)hbuf; it.done = false; it.rc = 0;
  sha1_batcher().submit(&it);
  if (it.rc != ZPQ_OK) fail(nullptr, it.rc, "SHA1");
  n = 0;
  return hbuf;
}

const char* SHA256::result() {
  Sha256Batcher::Item it;
  it.job.p = p ? p : (const uint8_t*)""; it.job.n = n; it.job.out = (uint8_t*)hbuf; it.done = false; it.rc = 0;
  sha256_batcher().submit(&it);
  if (it.rc != ZPQ_OK) fail(nullptr, it.rc, "SHA256");
  n = 0;
  return hbuf;
}

void sha1_many(const char* const* bufs, const size_t* lens, size_t n, char* digests) {
  EngineHolder& e = engine();
  std::lock_guard<std::mutex> g(e.mu);
  zpq_ctx* ctx = e.get();
  const int rc = zpq_sha1_many(ctx, (const uint8_t* const*)bufs, lens, n, (uint8_t*)digests);
  if (rc != ZPQ_OK) fail(ctx, rc, "sha1_many");
}

void compressBlock(StringBuffer* in, Writer* out, const char* method, const char* filename, const char* comment,
                   bool dosha1) {
  if (!in || !out || !method || !method[0]) error("compressBlock: bad arguments");
  const size_t n = in->size();
  if (n > 0xffffffffu) error("compressBlock: block too large");
  std::vector<uint8_t> framed(zpq_block_bound(n, filename, comment));
  CompressBatcher::Item it;
  memset(&it.job, 0, sizeof it.job);
  it.done = false; it.rc = 0;
  it.job.in = in->data() ? in->data() : (const uint8_t*)"";
  it.job.n = (uint32_t)n;
  it.job.method = method;
  it.job.filename = filename;
  it.job.comment = comment;
  it.job.dosha1 = dosha1 ? 1 : 0;
  it.job.out = framed.data();
  it.job.out_cap = (uint32_t)framed.size();
  compress_batcher().submit(&it);
  if (it.rc != ZPQ_OK || it.job.status != ZPQ_OK) {      // a batch can fail before it touches its jobs (status stays 0)
    std::string m = std::string("compressBlock(\"") + method + "\"): " + zpq_strerror(it.job.status ? it.job.status : it.rc);
    error(m.c_str());
  }
  out->write((const char*)framed.data(), (int)it.job.out_len);
  in->resize(0);
}

void compress(Reader* in, Writer* out, const char* method, const char* filename, const char* comment, bool dosha1) {
  // block size from the method string, as the reference documents (ZSFX/libzpaq.h:86-135):
  // "LB..." or "xB..." -> 2^(20+B) - 4096 bytes per block, default B = 4; filename and comment go
  // on the first block only; empty input produces no block.
  int bs = 4;
  if (method && method[0] && method[1] >= '0' && method[1] <= '9') {
    bs = method[1] - '0';
    if (method[2] >= '0' && method[2] <= '9') bs = bs * 10 + method[2] - '0';
    if (bs > 11) bs = 11;
  }
  const size_t block = ((size_t)0x100000 << bs) - 4096;
  StringBuffer sb(block);
  std::vector<char> tmp(1 << 20);
  while (in) {
    sb.resize(0);
    while (sb.size() < block) {
      const int want = (int)std::min(tmp.size(), block - sb.size());
      const int r = in->read(tmp.data(), want);
      if (r <= 0) break;
      sb.write(tmp.data(), r);
    }
    if (sb.size() == 0) break;
    const bool last = sb.size() < block;
    compressBlock(&sb, out, method, filename, comment, dosha1);
    filename = 0;
    comment = 0;
    if (last) break;
  }
}

// ---- Decompresser ------------------------------------------------------------------------------------------
struct Decompresser::Impl {
  std::vector<uint8_t> head;      // tag, "zPQ", level, type, header
  std::vector<uint8_t> seg;       // 1 filename 0 comment 0 0
  std::vector<uint8_t> payload;   // coded data
  std::vector<uint8_t> plain;     // decoded segment
  size_t given = 0;
  uint32_t ncomp = 0;
  int state = 0;                  // 0 no block, 1 expecting segment/end, 2 expecting comment, 3 data, 4 data read (marker known)
  bool decoded = false, have_marker = false;
  uint8_t marker[21] = {0};       // [0] = 1 if SHA-1 present
  unsigned segments = 0;
  bool first_was_pass = false;    // the block's first segment opened with 0 (no post-processor program)
  uint64_t usize_hint = 0;
  // A block whose later segments continue the first one's model or post-processor (ZSFX/libzpaq.cpp:2307-2337) is read to
  // its end and decoded in ONE device job when its first segment is asked for: `ahead` holds the bytes read past the first
  // segment (findFilename / readComment / readSegmentEnd take them from there), seg_plain the decoded segments.
  std::deque<uint8_t> ahead;
  std::vector<std::vector<uint8_t>> seg_plain;
  bool block_decoded = false;
  int next(Reader* in) { if (!ahead.empty()) { const int c = ahead.front(); ahead.pop_front(); return c; } return in->get(); }
  int next_read(Reader* in, char* buf, int k) {
    int got = 0;
    while (got < k && !ahead.empty()) { buf[got++] = (char)ahead.front(); ahead.pop_front(); }
    if (got < k) { const int r = in->read(buf + got, k - got); if (r > 0) got += r; }
    return got;
  }
};

Decompresser::Decompresser() : d_(new Impl), in_(0), out_(0), sha_(0) {}
Decompresser::~Decompresser() { delete d_; }
// What was read from the Reader and not yet consumed (ZSFX/libzpaq.h:1256; callers compute archive offsets as
// in.tell() - d.buffered(), ZSFX/zsfx.cpp:1434, 1592, 1597): the bytes the look-ahead of decompress() put back.
int Decompresser::buffered() { return (int)d_->ahead.size(); }
void Decompresser::setInput(Reader* in) { in_ = in; d_->ahead.clear(); }     // (bytes read ahead belong to the Reader they came from)

bool Decompresser::findBlock(double* memptr) {
  if (!in_) error("Decompresser: no input");
  Impl& d = *d_;
  // ZSFX/libzpaq.cpp:2239-2262: four rolling hashes over the last 16 bytes (their multipliers 12, 20, 28, 44 all hold a
  // factor 4, so the sixteenth power is 0 mod 2^32 and older bytes drop out), looked at after every byte, equal to those of
  // the 13-byte tag followed by "zPQ".  They start as if the tag had just been read: a stream that BEGINS with "zPQ" is a
  // block (Compressor::writeTag() is optional).  A tag that is not followed by "zPQ" is just bytes: the scan goes on.
  uint32_t h1 = 0x3D49B113u, h2 = 0x29EB7F93u, h3 = 0x2614BE13u, h4 = 0x3828EB13u;
  int c;
  while ((c = d.next(in_)) >= 0) {
    h1 = h1 * 12 + (uint32_t)c; h2 = h2 * 20 + (uint32_t)c; h3 = h3 * 28 + (uint32_t)c; h4 = h4 * 44 + (uint32_t)c;
    if (h1 == 0xB16B88F1u && h2 == 0xFF5376F1u && h3 == 0x72AC5BF1u && h4 == 0x2F909AF1u) break;
  }
  if (c < 0) return false;
  // (the engine's block jobs carry the tag: what is handed on is tag + "zPQ" whether or not the stream had the tag)
  d.head.assign(kTag, kTag + 13);
  uint8_t h[7] = {'z', 'P', 'Q', 0, 0, 0, 0};
  for (int i = 3; i < 7; ++i) {
    const int b = d.next(in_);
    if (i == 3 && b != 1 && b != 2) error("unsupported ZPAQ level");
    if (i == 4 && b != 1) error("unsupported ZPAQL type");
    if (b < 0) error("unexpected end of block header");
    h[i] = (uint8_t)b;
  }
  d.head.insert(d.head.end(), h, h + 7);
  const size_t hsize = h[5] | (size_t)h[6] << 8;
  for (size_t i = 0; i < hsize; ++i) { const int c = d.next(in_); if (c < 0) error("unexpected end of block header"); d.head.push_back((uint8_t)c); }
  if (hsize < 7) error("block header too short");
  const uint8_t* z = &d.head[18];              // hsize[2] hh hm ph pm n ...
  d.ncomp = z[6];
  if (h[3] == 1 && d.ncomp == 0) error("ZPAQ level 1 requires at least 1 component");     // ZSFX/libzpaq.cpp:2258-2259
  if (memptr) {                                 // ZPAQL::memory(), ZSFX/libzpaq.cpp:1001-1030
    double mem = 4.0 * (1u << z[2]) + (double)(1u << z[3]) + 4.0 * (1u << z[4]) + (double)(1u << z[5]) + (double)hsize + 512;
    size_t cp = 7;
    static const int sz[10] = {0, 2, 3, 2, 3, 4, 6, 6, 3, 5};
    for (uint32_t i = 0; i < d.ncomp && cp + 1 < hsize + 2; ++i) {
      const int t = z[cp]; const double s = (double)(1ull << (z[cp + 1] & 31));
      switch (t) {
        case 2: mem += 4 * s; break;
        case 3: mem += 64 * s + 1024; break;
        case 4: mem += 4 * s + (double)(1ull << (z[cp + 2] & 31)); break;
        case 6: mem += 2 * s; break;
        case 7: mem += 4 * s * z[cp + 3]; break;
        case 8: mem += 64 * s + 2048; break;
        case 9: mem += 128 * s; break;
        default: break;
      }
      if (t < 1 || t > 9) break;
      cp += sz[t];
    }
    *memptr = mem;
  }
  d.state = 1; d.segments = 0; d.first_was_pass = false;
  d.block_decoded = false; d.seg_plain.clear();
  return true;
}

void Decompresser::hcomp(Writer* out2) {
  if (out2 && d_->head.size() > 18) out2->write((const char*)&d_->head[18], (int)(d_->head.size() - 18));
}

bool Decompresser::findFilename(Writer* filename) {
  Impl& d = *d_;
  if (d.state != 1) error("findFilename: not at a segment boundary");
  const int c = d.next(in_);
  if (c == 255) { d.state = 0; return false; }                 // end of block
  if (c != 1) error("missing segment or end of block");
  ++d.segments;
  d.seg.assign(1, 1);
  for (;;) {
    const int b = d.next(in_);
    if (b < 0) error("unexpected end of input");
    d.seg.push_back((uint8_t)b);
    if (b == 0) break;
    if (filename) filename->put(b);
  }
  d.state = 2;
  return true;
}

void Decompresser::readComment(Writer* comment) {
  Impl& d = *d_;
  if (d.state != 2) error("readComment: no segment open");
  d.usize_hint = 0; bool digits = true;
  for (;;) {
    const int b = d.next(in_);
    if (b < 0) error("unexpected end of input");
    d.seg.push_back((uint8_t)b);
    if (b == 0) break;
    if (digits && b >= '0' && b <= '9' && d.usize_hint < (1ull << 40)) d.usize_hint = d.usize_hint * 10 + (uint64_t)(b - '0'); else digits = false;
    if (comment) comment->put(b);
  }
  const int r = d.next(in_);
  if (r != 0) error("missing reserved byte");
  d.seg.push_back(0);
  d.payload.clear(); d.plain.clear(); d.given = 0; d.decoded = false; d.have_marker = false;
  d.state = 3;
}

namespace {
// reads the coded bytes of the open segment and the 253/254 record that closes it (the same walk
// Decompresser::decompress / Decoder::skip do, ZSFX/libzpaq.cpp:2139-2160, 2339-2366)
template <class Next, class NextRead>
const char* try_read_payload(Next next, NextRead next_read, uint32_t ncomp, std::vector<uint8_t>& pay, uint8_t (&marker)[21]) {
  // returns 0, or what is wrong with the stream (the caller decides whether that is an error() now or later)
  static const char* const kEof = "unexpected end of compressed data";
  int c;
#define ZPQ_GET() do { c = next(); if (c < 0) return kEof; } while (0)
  if (ncomp) {
    uint32_t curr = 0;
    while (curr == 0) { ZPQ_GET(); pay.push_back((uint8_t)c); curr = (uint32_t)c; }
    while (curr) { ZPQ_GET(); pay.push_back((uint8_t)c); curr = curr << 8 | (uint32_t)c; }
    for (;;) { ZPQ_GET(); if (c != 0) break; pay.push_back(0); }        // the coder's own last byte may be 0 as well
  } else {
    for (;;) {
      uint32_t k = 0;
      for (int i = 0; i < 4; ++i) { ZPQ_GET(); pay.push_back((uint8_t)c); k = k << 8 | (uint32_t)c; }
      if (!k) break;
      const size_t at = pay.size();
      pay.resize(at + k);
      const int got = next_read((char*)&pay[at], (int)k);
      if (got != (int)k) { pay.resize(at + (got > 0 ? (size_t)got : 0)); return kEof; }
    }
    ZPQ_GET();
  }
  if (c == 253) {
    marker[0] = 1;
    for (int i = 1; i <= 20; ++i) {
      c = next();
      if (c < 0) { pay.push_back(253); pay.insert(pay.end(), marker + 1, marker + i); return kEof; }     // (kept, as below)
      marker[i] = (uint8_t)c;
    }
  }
  else if (c == 254) marker[0] = 0;
  else { pay.push_back((uint8_t)c); return "missing end of segment marker"; }     // (kept: a soft caller puts every byte read back)
#undef ZPQ_GET
  return 0;
}
template <class Next, class NextRead>
void read_payload(Next next, NextRead next_read, uint32_t ncomp, std::vector<uint8_t>& pay, uint8_t (&marker)[21]) {
  if (const char* what = try_read_payload(next, next_read, ncomp, pay, marker)) error(what);
}
}  // namespace

namespace {
// one framed block through the batched device call; seg_ends: room for the block's segments (result: where each ends in out)
void decode_block(const std::vector<uint8_t>& blk_padded, size_t cap, std::vector<uint8_t>& plain, std::vector<uint32_t>* seg_ends) {
  const size_t kCapMax = 0xffffffffu - 128;       // out_cap is a uint32_t (+ 64 bytes of padding)
  if (cap > kCapMax) cap = kCapMax;               // (a size hint out of a comment is whatever the archive says)
  for (;;) {
    plain.resize(cap + 64);
    DecompressBatcher::Item it;
    memset(&it.job, 0, sizeof it.job);
    it.done = false; it.rc = 0;
    it.job.in = blk_padded.data(); it.job.n = (uint32_t)(blk_padded.size() - 64);
    it.job.out = plain.data(); it.job.out_cap = (uint32_t)plain.size();
    if (seg_ends) { it.job.seg_cap = (uint32_t)seg_ends->size(); it.job.seg_out_end = seg_ends->data(); }
    decompress_batcher().submit(&it);                             // N decompressThreads -> one launch
    const int st = it.job.status ? it.job.status : it.rc;
    // (no usable size in the comments: the guess was too small; a job's capacity is 32 bits wide, so is the last try)
    if (st == ZPQ_ERR_CAPACITY && cap < kCapMax) { cap = cap > kCapMax / 4 ? kCapMax : cap * 4; continue; }
    if (st != ZPQ_OK) {                                           // a batch can fail before it touches its jobs
      std::string m = std::string("Decompresser: ") + zpq_strerror(st);
      error(m.c_str());
    }
    plain.resize(it.job.out_len);
    if (seg_ends && it.job.nseg != seg_ends->size()) error("Decompresser: segment count of the block differs from its framing");
    return;
  }
}
}  // namespace

bool Decompresser::decompress(int n) {
  Impl& d = *d_;
  if (d.state != 3 && d.state != 4) error("decompress: no segment open");
  auto nx = [&]() { return d.next(in_); };
  auto nr = [&](char* buf, int k) { return d.next_read(in_, buf, k); };
  if (!d.decoded) {
    if (!d.have_marker) { read_payload(nx, nr, d.ncomp, d.payload, d.marker); d.have_marker = true; d.state = 4; }
    if (d.segments > 1) {
      if (d.block_decoded) {
        if (d.segments > d.seg_plain.size()) error("Decompresser: segment beyond the decoded block");
        d.plain = d.seg_plain[d.segments - 1];
      } else {
        // without a model and without a post-processor program a later segment is a plain copy of its stored bytes; anything
        // else continues the first segment's decoder, which was not run: the reference refuses that as well
        // (ZSFX/libzpaq.cpp:2309: "decompression after skipped segment")
        if (d.ncomp || !d.first_was_pass) error("decompression after skipped segment");
        d.plain.clear();
        for (size_t p = 0; p + 4 <= d.payload.size();) {
          const size_t k = (size_t)d.payload[p] << 24 | (size_t)d.payload[p + 1] << 16 | (size_t)d.payload[p + 2] << 8 | d.payload[p + 3];
          p += 4;
          if (!k) break;
          d.plain.insert(d.plain.end(), d.payload.begin() + p, d.payload.begin() + p + k);
          p += k;
        }
      }
    } else {
      std::vector<uint8_t> blk(d.head);
      blk.insert(blk.end(), d.seg.begin(), d.seg.end());
      blk.insert(blk.end(), d.payload.begin(), d.payload.end());
      if (d.marker[0]) { blk.push_back(253); blk.insert(blk.end(), d.marker + 1, d.marker + 21); } else blk.push_back(254);
      d.first_was_pass = d.ncomp == 0 && !d.payload.empty() && d.payload.size() > 4 && d.payload[4] == 0;
      // Does another segment follow that depends on this one (a model, or a post-processor program: both carry on,
      // ZSFX/libzpaq.cpp:2312-2317)?  Then the whole block is read now and decoded by one device job.
      std::vector<uint8_t> rest;                                    // the bytes after this segment, up to and including 255
      bool cut = false; size_t cut_at = 0;                          // the framing behind the first segment is damaged at rest[cut_at]
      size_t nseg = 1, hint_sum = d.usize_hint, coded = d.payload.size();
      bool hints = d.usize_hint != 0;
      if (d.ncomp || !d.first_was_pass) {
        // Nothing wrong with the framing BEHIND the first segment is an error of this call: the reference delivers the segment
        // and fails at the next findFilename() (ZSFX/libzpaq.cpp:2269-2289).  So the look-ahead only collects: every byte it takes
        // goes into `rest`, and where the stream ends or stops making sense the block is cut after its last whole segment -- the
        // bytes go back for findFilename() / readComment() / readSegmentEnd() to read again and to report.
        size_t whole = 0;                                           // bytes of `rest` that are whole later segments
        size_t hint_ok = hint_sum, coded_ok = coded; bool hints_ok = hints;
        bool bad = false;
        auto take = [&]() -> int { const int b = nx(); if (b < 0) bad = true; else rest.push_back((uint8_t)b); return b; };
        int c = take();
        while (!bad && c == 1) {
          for (int b = take(); !bad && b != 0; b = take()) {}       // file name
          size_t v = 0; bool digits = true, any = false;            // comment: the decimal size in front, if there is one
          for (int b = bad ? 0 : take(); !bad && b != 0; b = take()) { if (digits && b >= '0' && b <= '9' && v < ((size_t)1 << 40)) { v = v * 10 + (size_t)(b - '0'); any = true; } else digits = false; }
          if (bad || take() != 0) { bad = true; break; }            // (the reserved byte)
          if (any) hint_sum += v; else hints = false;
          std::vector<uint8_t> pay; uint8_t mk[21] = {0};
          const char* what = try_read_payload(nx, nr, d.ncomp, pay, mk);
          coded += pay.size();
          rest.insert(rest.end(), pay.begin(), pay.end());
          if (what) { bad = true; break; }
          if (mk[0]) { rest.push_back(253); rest.insert(rest.end(), mk + 1, mk + 21); } else rest.push_back(254);
          ++nseg;
          whole = rest.size(); hint_ok = hint_sum; coded_ok = coded; hints_ok = hints;
          c = take();
        }
        if (bad || c != 255) {
          // cut after the last whole segment: the device job gets those (closed with a 255 of its own), the stream keeps the rest
          hint_sum = hint_ok; coded = coded_ok; hints = hints_ok;
          cut_at = whole ? whole : 0;
          cut = true;
        }
      }
      if (nseg == 1) {
        for (size_t q = rest.size(); q-- > 0;) d.ahead.push_front(rest[q]);       // (the 255 goes back for findFilename)
        blk.push_back(255);
        blk.resize(blk.size() + 64);                                  // readable padding (include/zpaqhip.h)
        decode_block(blk, d.usize_hint ? (size_t)d.usize_hint : d.payload.size() * 64 + 65536, d.plain, nullptr);
      } else {
        // rest = [1 segment]... then 255, or -- cut -- whole segments followed by whatever the stream holds there
        if (cut) { blk.insert(blk.end(), rest.begin(), rest.begin() + cut_at); blk.push_back(255); }
        else blk.insert(blk.end(), rest.begin(), rest.end());
        blk.resize(blk.size() + 64);
        std::vector<uint8_t> all;
        std::vector<uint32_t> ends(nseg, 0);
        decode_block(blk, hints ? hint_sum + 64 : coded * 64 + 65536, all, &ends);
        d.seg_plain.assign(nseg, std::vector<uint8_t>());
        for (size_t q = 0; q < nseg; ++q) {
          const size_t from = q ? ends[q - 1] : 0;
          if (ends[q] < from || ends[q] > all.size()) error("Decompresser: segment ends out of order");
          d.seg_plain[q].assign(all.begin() + from, all.begin() + ends[q]);
        }
        d.block_decoded = true;
        d.plain = d.seg_plain[0];
        for (size_t q = rest.size(); q-- > 0;) d.ahead.push_front(rest[q]);       // the later segments are served from here
      }
    }
    d.decoded = true; d.given = 0;
  }
  size_t k = d.plain.size() - d.given;
  if (n >= 0 && (size_t)n < k) k = (size_t)n;
  if (k) {
    if (out_) out_->write((const char*)&d.plain[d.given], (int)k);
    if (sha_) sha_->write((const char*)&d.plain[d.given], (int64_t)k);
    d.given += k;
  }
  return d.given < d.plain.size();
}

// The PCOMP section of the block (ZSFX/libzpaq.h:1254: pp.z.write(out2, true) -- two size bytes, then the program), valid once the
// first segment has been read.  Blocks without a context model keep it in their stored sub-blocks, where it is read from
// here; behind a model it is the head of the coded stream, which the device decodes as far as the section reaches
// (zpq_cm_decode_dev stops at out_cap with the bytes produced so far valid): three bytes for the size, then the program.
bool Decompresser::pcomp(Writer* out2) {
  Impl& d = *d_;
  if (!d.have_marker || d.segments != 1) return false;
  std::vector<uint8_t> first;                                  // the first bytes of the decoded stream: 1 lo hi program...
  if (d.ncomp == 0) {
    for (size_t p = 0; p + 4 <= d.payload.size();) {
      const size_t k = (size_t)d.payload[p] << 24 | (size_t)d.payload[p + 1] << 16 | (size_t)d.payload[p + 2] << 8 | d.payload[p + 3];
      p += 4;
      if (!k || p + k > d.payload.size()) break;
      first.insert(first.end(), d.payload.begin() + p, d.payload.begin() + p + k);
      p += k;
      if (first.size() >= 3 && first.size() >= 3 + ((size_t)first[1] | (size_t)first[2] << 8)) break;
    }
  } else {
    if (d.head.size() < 20 || d.payload.empty()) return false;
    EngineHolder& e = engine();
    std::lock_guard<std::mutex> g(e.mu);
    zpq_ctx* ctx = e.get();
    void *d_in = nullptr, *d_out = nullptr;
    const size_t cap_max = 3 + 65535 + 64;
    int rc = zpq_dev_alloc(ctx, d.payload.size() + 64, &d_in);
    if (rc == ZPQ_OK) rc = zpq_dev_alloc(ctx, cap_max, &d_out);
    if (rc == ZPQ_OK) rc = zpq_h2d(ctx, d_in, d.payload.data(), d.payload.size());
    for (size_t want = 3; rc == ZPQ_OK;) {
      zpq_cm_job j;
      memset(&j, 0, sizeof j);
      j.header = d.head.data() + 18; j.header_len = (uint32_t)(d.head.size() - 18);       // hsize[2] hh hm ph pm n COMP 0 HCOMP 0
      j.d_in = (const uint8_t*)d_in; j.n = (uint32_t)d.payload.size();
      j.d_out = (uint8_t*)d_out; j.out_cap = (uint32_t)want;
      rc = zpq_cm_decode_dev(ctx, &j, 1);
      if (rc != ZPQ_OK && rc != ZPQ_ERR_CAPACITY) break;
      if (j.status != ZPQ_OK && j.status != ZPQ_ERR_CAPACITY) { rc = j.status; break; }
      first.resize(j.out_len < want ? j.out_len : want);
      rc = first.empty() ? ZPQ_OK : zpq_d2h(ctx, first.data(), d_out, first.size());
      if (rc != ZPQ_OK || first.size() < 3 || first[0] != 1 || want > 3) break;
      want = 3 + ((size_t)first[1] | (size_t)first[2] << 8);
    }
    if (d_in) zpq_dev_free(ctx, d_in);
    if (d_out) zpq_dev_free(ctx, d_out);
    if (rc != ZPQ_OK && rc != ZPQ_ERR_CAPACITY) fail(ctx, rc, "Decompresser::pcomp");
  }
  if (first.size() < 3 || first[0] != 1) return false;
  const size_t n = (size_t)first[1] | (size_t)first[2] << 8;
  if (first.size() < 3 + n) return false;
  if (out2) { out2->put(first[1]); out2->put(first[2]); out2->write((const char*)&first[3], (int)n); }
  return true;
}

void Decompresser::readSegmentEnd(char* sha1string) {
  Impl& d = *d_;
  if (d.state != 3 && d.state != 4) error("readSegmentEnd: no segment open");
  if (!d.have_marker) {                                                            // segment skipped undecoded
    read_payload([&]() { return d.next(in_); }, [&](char* buf, int k) { return d.next_read(in_, buf, k); }, d.ncomp, d.payload, d.marker); d.have_marker = true;
    if (d.segments == 1) d.first_was_pass = d.ncomp == 0 && d.payload.size() > 4 && d.payload[4] == 0;
  }
  if (sha1string) memcpy(sha1string, d.marker, 21);
  d.state = 1;
}

void decompress(Reader* in, Writer* out) {
  // the reference's loop (ZSFX/libzpaq.cpp:2368-2381): every block, every segment, SHA-1s verified
  Decompresser d;
  d.setInput(in);
  d.setOutput(out);
  while (d.findBlock()) {
    while (d.findFilename()) {
      d.readComment();
      SHA1 sha;
      d.setSHA1(&sha);
      d.decompress();
      char rec[21];
      d.readSegmentEnd(rec);
      if (rec[0] && memcmp(rec + 1, sha.result(), 20) != 0) error("decompress: checksum mismatch");
    }
  }
}

// ---- Compressor ----------------------------------------------------------------------------------------------
struct Compressor::Impl {
  std::vector<uint8_t> header;    // hsize[2] hh hm ph pm n COMP 0 HCOMP 0
  std::vector<uint8_t> pcomp;     // compiled post-processor byte code (with its closing 0), empty = none
  std::vector<uint8_t> data;      // what the Encoder sees in the open segment
  int state = 0;                  // 0 INIT, 1 BLOCK1, 2 SEG1, 3 BLOCK2, 4 SEG2 (ZSFX/libzpaq.h:1369)
  unsigned segments = 0;
  bool pp_done = false;
  // A block with a context model is coded by ONE device job when it is closed: a later segment continues the first one's
  // model (the Encoder's Predictor is initialised by startBlock only), so the segments' headers, bytes and end markers
  // wait here until endBlock.  (Blocks without a model are written as they come.)
  struct Seg { std::vector<uint8_t> head, data; uint8_t marker[21]; };
  std::vector<Seg> segs;
};

Compressor::Compressor() : d_(new Impl), out_(0), in_(0) {}
Compressor::~Compressor() { delete d_; }

void Compressor::writeTag() {
  if (!out_) error("Compressor: no output");
  out_->write((const char*)kTag, 13);
}

// level = 1, 2, 3: libzpaq's built-in models (min / mid / max.cfg), ZSFX/libzpaq.h:1346
void Compressor::startBlock(int level) {
  if (level < 1) error("compression level must be at least 1");
  if (level > 3) error("compression level too high");
  uint8_t h[256];
  size_t hl = 0;
  const int rc = zpq_builtin_model(level, h, sizeof h, &hl);
  if (rc != ZPQ_OK) fail(nullptr, rc, "Compressor::startBlock(level)");
  startBlock((const char*)h);
}

void Compressor::startBlock(const char* hcomp) {
  if (!out_) error("Compressor: no output");
  if (d_->state != 0 && d_->state != 3) error("startBlock: a block is already open");
  const uint8_t* h = (const uint8_t*)hcomp;
  const size_t len = (h[0] | (size_t)h[1] << 8) + 2;
  if (len < 9) error("startBlock: header too short");
  d_->header.assign(h, h + len);
  d_->pcomp.clear();
  out_->put('z'); out_->put('P'); out_->put('Q'); out_->put(1 + (d_->header[6] == 0)); out_->put(1);
  out_->write((const char*)d_->header.data(), (int)d_->header.size());
  d_->state = 1; d_->segments = 0; d_->pp_done = false; d_->segs.clear();
}

void Compressor::startBlock(const char* config, int* args, Writer* pcomp_cmd) {
  if (!out_) error("Compressor: no output");
  if (d_->state != 0 && d_->state != 3) error("startBlock: a block is already open");
  int32_t a[9] = {0};
  if (args) for (int i = 0; i < 9; ++i) a[i] = args[i];
  std::vector<uint8_t> h(70000), p(70000);
  size_t hl = 0, pl = 0;
  const int rc = zpq_compile_config(nullptr, config, a, h.data(), h.size(), &hl, p.data(), p.size(), &pl);
  if (rc != ZPQ_OK) fail(nullptr, rc, "Compressor::startBlock: config does not compile");
  (void)pcomp_cmd;                 // the "pcomp <command> ;" text is only used by external pre-processors
  d_->header.assign(h.begin(), h.begin() + hl);
  d_->pcomp.assign(p.begin(), p.begin() + pl);
  out_->put('z'); out_->put('P'); out_->put('Q'); out_->put(1 + (d_->header[6] == 0)); out_->put(1);
  out_->write((const char*)d_->header.data(), (int)d_->header.size());
  d_->state = 1; d_->segments = 0; d_->pp_done = false; d_->segs.clear();
}

void Compressor::hcomp(Writer* out2) { if (out2) out2->write((const char*)d_->header.data(), (int)d_->header.size()); }
bool Compressor::pcomp(Writer* out2) {
  if (d_->pcomp.empty()) return false;
  if (out2) { out2->put((int)(d_->pcomp.size() & 255)); out2->put((int)(d_->pcomp.size() >> 8)); out2->write((const char*)d_->pcomp.data(), (int)d_->pcomp.size()); }
  return true;
}

void Compressor::startSegment(const char* filename, const char* comment) {
  if (d_->state != 1 && d_->state != 3) error("startSegment: no block open");
  if (d_->header[6]) {                        // with a model: kept until endBlock
    d_->segs.emplace_back();
    std::vector<uint8_t>& h = d_->segs.back().head;
    h.push_back(1);
    if (filename) h.insert(h.end(), filename, filename + strlen(filename));
    h.push_back(0);
    if (comment) h.insert(h.end(), comment, comment + strlen(comment));
    h.push_back(0);
    h.push_back(0);
  } else {
    out_->put(1);
    if (filename) out_->write(filename, (int)strlen(filename));
    out_->put(0);
    if (comment) out_->write(comment, (int)strlen(comment));
    out_->put(0);
    out_->put(0);
  }
  d_->data.clear();
  ++d_->segments;
  d_->state = d_->state == 1 ? 2 : 4;
}

void Compressor::postProcess(const char* pcomp, int len) {
  Impl& d = *d_;
  if (d.state == 4 || d.pp_done) return;
  if (d.state != 2) error("postProcess: no first segment open");
  const uint8_t* pc = (const uint8_t*)pcomp;
  if (!pc) { len = (int)d.pcomp.size(); pc = len ? d.pcomp.data() : nullptr; }
  else if (len == 0) { len = pc[0] | pc[1] << 8; pc += 2; }
  if (len > 0) {
    d.data.push_back(1); d.data.push_back((uint8_t)(len & 255)); d.data.push_back((uint8_t)((len >> 8) & 255));
    d.data.insert(d.data.end(), pc, pc + len);
  } else d.data.push_back(0);
  d.pp_done = true;
}

bool Compressor::compress(int n) {
  Impl& d = *d_;
  if (d.state != 2 && d.state != 4) error("compress: no segment open");
  if (d.state == 2 && !d.pp_done) postProcess();
  if (!in_) error("compress: no input");
  char buf[1 << 16];
  while (n != 0) {
    int want = (int)sizeof buf;
    if (n > 0 && n < want) want = n;
    const int r = in_->read(buf, want);
    if (r <= 0) return false;
    d.data.insert(d.data.end(), buf, buf + r);
    sha1_.write(buf, r);
    if (n > 0) n -= r;
  }
  return true;
}

void Compressor::endSegment(const char* sha1string) {
  Impl& d = *d_;
  if (d.state != 2 && d.state != 4) error("endSegment: no segment open");
  if (d.state == 2 && !d.pp_done) postProcess();
  const uint32_t ncomp = d.header[6];
  if (ncomp == 0) {
    // Encoder stored mode: sub-blocks of at most 64 KiB with big-endian lengths, then a zero length
    for (size_t p = 0; p < d.data.size();) {
      const size_t k = std::min<size_t>(65536, d.data.size() - p);
      out_->put((int)(k >> 24)); out_->put((int)(k >> 16) & 255); out_->put((int)(k >> 8) & 255); out_->put((int)k & 255);
      out_->write((const char*)&d.data[p], (int)k);
      p += k;
    }
    out_->put(0); out_->put(0); out_->put(0); out_->put(0);
  } else {
    Impl::Seg& sg = d.segs.back();
    sg.data.swap(d.data);
    sg.marker[0] = sha1string ? 1 : 0;
    if (sha1string) memcpy(sg.marker + 1, sha1string, 20);
    d.data.clear();
    d.state = 3;
    return;
  }
  if (sha1string) { out_->put(253); out_->write(sha1string, 20); }
  else out_->put(254);
  d.data.clear();
  d.state = 3;
}

char* Compressor::endSegmentChecksum(int64_t* size, bool dosha1) {
  if (size) *size = (int64_t)sha1_.usize();
  if (dosha1) {
    memcpy(sha1result_, sha1_.result(), 20);
    endSegment(sha1result_);
    return sha1result_;
  }
  (void)sha1_.result();
  endSegment(0);
  return 0;
}

void Compressor::endBlock() {
  Impl& d = *d_;
  if (d.state != 3) error("endBlock: no block to close");
  if (!d.segs.empty()) {
    // every segment of the block through the context-model coder in one job: the model carries on from one to the next, the
    // coder ends each with its end-of-segment symbol and the four 0 bytes
    EngineHolder& e = engine();
    std::lock_guard<std::mutex> g(e.mu);
    zpq_ctx* ctx = e.get();
    size_t n = 0;
    std::vector<uint32_t> lens(d.segs.size()), ends(d.segs.size(), 0);
    for (size_t q = 0; q < d.segs.size(); ++q) { lens[q] = (uint32_t)d.segs[q].data.size(); n += d.segs[q].data.size(); }
    if (n > 0xfff00000ull) error("Compressor: block too large for one device job");
    const size_t cap = n + n / 8 + 4096 + 16 * d.segs.size();
    std::vector<uint8_t> all(n);
    for (size_t q = 0, at = 0; q < d.segs.size(); at += d.segs[q].data.size(), ++q)
      if (!d.segs[q].data.empty()) memcpy(&all[at], d.segs[q].data.data(), d.segs[q].data.size());
    void *d_in = nullptr, *d_out = nullptr;
    int rc = zpq_dev_alloc(ctx, n + 64, &d_in);
    if (rc == ZPQ_OK) rc = zpq_dev_alloc(ctx, cap + 64, &d_out);
    if (rc != ZPQ_OK) { if (d_in) zpq_dev_free(ctx, d_in); fail(ctx, rc, "Compressor"); }
    if (n) rc = zpq_h2d(ctx, d_in, all.data(), n);
    zpq_cm_job j;
    memset(&j, 0, sizeof j);
    j.header = d.header.data(); j.header_len = (uint32_t)d.header.size();
    j.d_in = (const uint8_t*)d_in; j.n = (uint32_t)n; j.d_out = (uint8_t*)d_out; j.out_cap = (uint32_t)cap;
    if (d.segs.size() > 1) { j.nseg = (uint32_t)d.segs.size(); j.seg_len = lens.data(); j.seg_out_end = ends.data(); }
    if (rc == ZPQ_OK) rc = zpq_cm_encode_dev(ctx, &j, 1);
    std::vector<uint8_t> coded;
    if (rc == ZPQ_OK && j.status == ZPQ_OK) { coded.resize(j.out_len); if (j.out_len) rc = zpq_d2h(ctx, coded.data(), d_out, j.out_len); }
    zpq_dev_free(ctx, d_in); zpq_dev_free(ctx, d_out);
    if (rc != ZPQ_OK || j.status != ZPQ_OK) fail(ctx, rc ? rc : j.status, "Compressor: context-model coder");
    if (d.segs.size() == 1) ends[0] = j.out_len;
    for (size_t q = 0; q < d.segs.size(); ++q) {
      const Impl::Seg& sg = d.segs[q];
      const size_t from = q ? ends[q - 1] : 0;
      if (ends[q] < from || ends[q] > coded.size()) error("Compressor: coded segments out of order");
      out_->write((const char*)sg.head.data(), (int)sg.head.size());
      out_->write((const char*)coded.data() + from, (int)(ends[q] - from));      // includes the end-of-segment symbol and the four 0 bytes
      if (sg.marker[0]) { out_->put(253); out_->write((const char*)sg.marker + 1, 20); } else out_->put(254);
    }
    d.segs.clear();
  }
  out_->put(255);
  d.state = 0;
}

}  // namespace libzpaq
