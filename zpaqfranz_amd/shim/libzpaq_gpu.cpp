// libzpaq_gpu.cpp -- see libzpaq_gpu.h.  Host-side C++ above the C ABI (include/zpaqhip.h).
#include "libzpaq_gpu.h"

#include <algorithm>
#include <condition_variable>
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

// Coalesces compressBlock() calls from concurrent worker threads into one zpq_compress_blocks launch.
struct Batcher {
  struct Item { zpq_block_job job; bool done; };
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
    std::vector<zpq_block_job> jobs(batch.size());
    for (size_t i = 0; i < batch.size(); ++i) jobs[i] = batch[i]->job;
    zpq_compress_blocks(ctx, jobs.data(), jobs.size());   // per-job status carries the outcome
    for (size_t i = 0; i < batch.size(); ++i) batch[i]->job = jobs[i];
  }
};
Batcher& batcher() { static Batcher b; return b; }

}  // namespace

void setDevice(int ordinal) { g_device = ordinal; }

const char* SHA1::result() {
  EngineHolder& e = engine();
  std::lock_guard<std::mutex> g(e.mu);
  zpq_ctx* ctx = e.get();
  const uint8_t* bufs[1] = {p ? p : (const uint8_t*)""};
  size_t lens[1] = {n};
  int rc = zpq_sha1_many(ctx, bufs, lens, 1, (uint8_t*)hbuf);
  if (rc != ZPQ_OK) fail(ctx, rc, "SHA1");
  n = 0;
  return hbuf;
}

const char* SHA256::result() {
  EngineHolder& e = engine();
  std::lock_guard<std::mutex> g(e.mu);
  zpq_ctx* ctx = e.get();
  const uint8_t* bufs[1] = {p ? p : (const uint8_t*)""};
  size_t lens[1] = {n};
  int rc = zpq_sha256_many(ctx, bufs, lens, 1, (uint8_t*)hbuf);
  if (rc != ZPQ_OK) fail(ctx, rc, "SHA256");
  n = 0;
  return hbuf;
}

void compressBlock(StringBuffer* in, Writer* out, const char* method, const char* filename, const char* comment,
                   bool dosha1) {
  if (!in || !out || !method || !method[0]) error("compressBlock: bad arguments");
  const size_t n = in->size();
  if (n > 0xffffffffu) error("compressBlock: block too large");
  std::vector<uint8_t> framed(zpq_block_bound(n, filename, comment));
  Batcher::Item it;
  memset(&it.job, 0, sizeof it.job);
  it.done = false;
  it.job.in = in->data() ? in->data() : (const uint8_t*)"";
  it.job.n = (uint32_t)n;
  it.job.method = method;
  it.job.filename = filename;
  it.job.comment = comment;
  it.job.dosha1 = dosha1 ? 1 : 0;
  it.job.out = framed.data();
  it.job.out_cap = (uint32_t)framed.size();
  batcher().submit(&it);
  if (it.job.status != ZPQ_OK) {
    std::string m = std::string("compressBlock(\"") + method + "\"): " + zpq_strerror(it.job.status);
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

void decompress(Reader* in, Writer* out) {
  // slurp the archive, then hand every block to the engine (blocks are independent)
  std::vector<uint8_t> arc;
  {
    char buf[1 << 16];
    int r;
    while ((r = in->read(buf, sizeof buf)) > 0) arc.insert(arc.end(), buf, buf + r);
  }
  static const uint8_t tag[13] = {0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3};
  EngineHolder& e = engine();
  size_t pos = 0;
  while (pos + 13 <= arc.size()) {
    // findBlock (ZSFX/libzpaq.cpp:2239-2262): scan for the 13-byte tag
    size_t at = pos;
    while (at + 13 <= arc.size() && memcmp(&arc[at], tag, 13) != 0) ++at;
    if (at + 13 > arc.size()) break;
    // output size: the comment begins with the decimal size (compressBlock contract); fall back to a generous bound
    size_t cap = 0;
    {
      size_t q = at + 13 + 5;
      if (q + 2 <= arc.size()) {
        q += 2 + (arc[q] | arc[q + 1] << 8) + 1;
        while (q < arc.size() && arc[q]) ++q;
        ++q;
        size_t v = 0; bool any = false;
        while (q < arc.size() && arc[q] >= '0' && arc[q] <= '9') { v = v * 10 + (arc[q] - '0'); ++q; any = true; }
        if (any) cap = v;
      }
    }
    if (cap == 0) cap = (arc.size() - at) * 64 + 65536;
    std::vector<uint8_t> outbuf(cap + 64);
    zpq_unblock_job j;
    memset(&j, 0, sizeof j);
    j.in = &arc[at]; j.n = (uint32_t)std::min<size_t>(arc.size() - at, 0xffffffffu);
    j.out = outbuf.data(); j.out_cap = (uint32_t)outbuf.size();
    int rc;
    {
      std::lock_guard<std::mutex> g(e.mu);
      rc = zpq_decompress_blocks(e.get(), &j, 1, 1);
    }
    if (rc != ZPQ_OK || j.status != ZPQ_OK) {
      std::string m = std::string("decompress: ") + zpq_strerror(j.status ? j.status : rc);
      error(m.c_str());
    }
    out->write((const char*)outbuf.data(), (int)j.out_len);
    pos = at + j.consumed;
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
  uint64_t usize_hint = 0;
};

Decompresser::Decompresser() : d_(new Impl), in_(0), out_(0), sha_(0) {}
Decompresser::~Decompresser() { delete d_; }

bool Decompresser::findBlock(double* memptr) {
  static const uint8_t tag[13] = {0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3};
  if (!in_) error("Decompresser: no input");
  Impl& d = *d_;
  // ZSFX/libzpaq.cpp:2239-2262: the block starts right after the 13-byte tag, wherever that is
  uint8_t win[13]; size_t have = 0;
  for (;;) {
    const int c = in_->get();
    if (c < 0) return false;
    if (have < 13) win[have++] = (uint8_t)c;
    else { memmove(win, win + 1, 12); win[12] = (uint8_t)c; }
    if (have == 13 && memcmp(win, tag, 13) == 0) break;
  }
  d.head.assign(tag, tag + 13);
  uint8_t h[7];
  for (int i = 0; i < 7; ++i) { const int c = in_->get(); if (c < 0) error("unexpected end of block header"); h[i] = (uint8_t)c; }
  if (h[0] != 'z' || h[1] != 'P' || h[2] != 'Q' || (h[3] != 1 && h[3] != 2) || h[4] != 1) error("unsupported ZPAQ level or type");
  d.head.insert(d.head.end(), h, h + 7);
  const size_t hsize = h[5] | (size_t)h[6] << 8;
  for (size_t i = 0; i < hsize; ++i) { const int c = in_->get(); if (c < 0) error("unexpected end of block header"); d.head.push_back((uint8_t)c); }
  if (hsize < 7) error("block header too short");
  const uint8_t* z = &d.head[18];              // hsize[2] hh hm ph pm n ...
  d.ncomp = z[6];
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
  d.state = 1; d.segments = 0;
  return true;
}

void Decompresser::hcomp(Writer* out2) {
  if (out2 && d_->head.size() > 18) out2->write((const char*)&d_->head[18], (int)(d_->head.size() - 18));
}

bool Decompresser::findFilename(Writer* filename) {
  Impl& d = *d_;
  if (d.state != 1) error("findFilename: not at a segment boundary");
  const int c = in_->get();
  if (c == 255) { d.state = 0; return false; }                 // end of block
  if (c != 1) error("missing segment or end of block");
  if (d.segments++ > 0) error("blocks with more than one segment are not supported by the GPU engine");
  d.seg.assign(1, 1);
  for (;;) {
    const int b = in_->get();
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
    const int b = in_->get();
    if (b < 0) error("unexpected end of input");
    d.seg.push_back((uint8_t)b);
    if (b == 0) break;
    if (digits && b >= '0' && b <= '9') d.usize_hint = d.usize_hint * 10 + (uint64_t)(b - '0'); else digits = false;
    if (comment) comment->put(b);
  }
  const int r = in_->get();
  if (r != 0) error("missing reserved byte");
  d.seg.push_back(0);
  d.payload.clear(); d.plain.clear(); d.given = 0; d.decoded = false; d.have_marker = false;
  d.state = 3;
}

namespace {
// reads the coded bytes of the open segment and the 253/254 record that closes it (the same walk
// Decompresser::decompress / Decoder::skip do, ZSFX/libzpaq.cpp:2139-2160, 2339-2366)
void read_payload(Reader* in, uint32_t ncomp, std::vector<uint8_t>& pay, uint8_t (&marker)[21]) {
  auto get = [&]() -> int { const int c = in->get(); if (c < 0) error("unexpected end of compressed data"); return c; };
  int c;
  if (ncomp) {
    uint32_t curr = 0;
    while (curr == 0) { c = get(); pay.push_back((uint8_t)c); curr = (uint32_t)c; }
    while (curr) { c = get(); pay.push_back((uint8_t)c); curr = curr << 8 | (uint32_t)c; }
    for (c = get(); c == 0; c = get()) pay.push_back(0);        // the coder's own last byte may be 0 as well
  } else {
    for (;;) {
      uint32_t k = 0;
      for (int i = 0; i < 4; ++i) { c = get(); pay.push_back((uint8_t)c); k = k << 8 | (uint32_t)c; }
      if (!k) break;
      const size_t at = pay.size();
      pay.resize(at + k);
      if (in->read((char*)&pay[at], (int)k) != (int)k) error("unexpected end of compressed data");
    }
    c = get();
  }
  if (c == 253) { marker[0] = 1; for (int i = 1; i <= 20; ++i) marker[i] = (uint8_t)get(); }
  else if (c == 254) marker[0] = 0;
  else error("missing end of segment marker");
}
}  // namespace

bool Decompresser::decompress(int n) {
  Impl& d = *d_;
  if (d.state != 3 && d.state != 4) error("decompress: no segment open");
  if (!d.decoded) {
    if (!d.have_marker) { read_payload(in_, d.ncomp, d.payload, d.marker); d.have_marker = true; d.state = 4; }
    std::vector<uint8_t> blk(d.head);
    blk.insert(blk.end(), d.seg.begin(), d.seg.end());
    blk.insert(blk.end(), d.payload.begin(), d.payload.end());
    if (d.marker[0]) { blk.push_back(253); blk.insert(blk.end(), d.marker + 1, d.marker + 21); } else blk.push_back(254);
    blk.push_back(255);
    blk.resize(blk.size() + 64);                                  // readable padding (include/zpaqhip.h)
    size_t cap = d.usize_hint ? (size_t)d.usize_hint : d.payload.size() * 64 + 65536;
    d.plain.resize(cap + 64);
    zpq_unblock_job j;
    memset(&j, 0, sizeof j);
    j.in = blk.data(); j.n = (uint32_t)(blk.size() - 64);
    j.out = d.plain.data(); j.out_cap = (uint32_t)d.plain.size();
    EngineHolder& e = engine();
    int rc;
    {
      std::lock_guard<std::mutex> g(e.mu);
      rc = zpq_decompress_blocks(e.get(), &j, 1, 0);             // the caller verifies through setSHA1 / its own table
    }
    if (rc != ZPQ_OK || j.status != ZPQ_OK) {
      std::string m = std::string("Decompresser: ") + zpq_strerror(j.status ? j.status : rc);
      error(m.c_str());
    }
    d.plain.resize(j.out_len);
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

bool Decompresser::pcomp(Writer*) { return false; }

void Decompresser::readSegmentEnd(char* sha1string) {
  Impl& d = *d_;
  if (d.state != 3 && d.state != 4) error("readSegmentEnd: no segment open");
  if (!d.have_marker) { read_payload(in_, d.ncomp, d.payload, d.marker); d.have_marker = true; }   // segment skipped undecoded
  if (sha1string) memcpy(sha1string, d.marker, 21);
  d.state = 1;
}

}  // namespace libzpaq
