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

}  // namespace libzpaq
