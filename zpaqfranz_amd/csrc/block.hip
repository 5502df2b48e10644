// compressBlock / Decompresser for the stored + LZ77-level-1 method family (SURVEY.md rows a5, a10,
// a11, a15-a17).  Reference: libzpaq::compressBlock (declaration ZSFX/libzpaq.h:1505, contract
// :73-84, :286-294), Compressor framing (ZSFX/libzpaq.h:1340-1371; byte layout pinned by the reader
// Decompresser, ZSFX/libzpaq.cpp:2239-2366), Encoder stored mode (mirror of Decoder::decompress
// stored branch, ZSFX/libzpaq.cpp:2139-2146), PostProcessor preamble (:2185-2226).
//
// Host code decides the configuration (method string -> args, header bytes) exactly as
// compressBlock/makeConfig do; the bytes themselves never leave HBM: LZ77 (lz77.hip), the block
// SHA-1 chain (sha.hip, on the second stream so it overlaps the parse) and the framing copy
// (frame_kernel below: header, 64 KiB stored sub-blocks with big-endian lengths, trailer).
#include <stdlib.h>

#include <map>
#include <mutex>

#include "zpq_internal.h"

namespace {

const u8 kTag[13] = {0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3};

// LZ77 level-1 post-processor program for rb = 0 without E8E9: the 302 bytes every "-m1" block of
// up to 16 MiB carries (golden: i-blocks of the reference's AUTOTEST/sha256.zpaq; SURVEY.md
// Appendix D).  Stored after the 2-byte little-endian length 0x012e.
}  // namespace
const u8 zpq_pcomp_lz1[302] = {
    0xef, 0xff, 0x2f, 0x0d, 0x04, 0x0c, 0x14, 0x1c, 0x37, 0x01, 0x37, 0x02, 0x37, 0x03, 0x37, 0x04, 0x38, 0xcb, 0x82,
    0x50, 0x47, 0x08, 0x83, 0x58, 0x07, 0x01, 0xdf, 0x00, 0x2f, 0x33, 0x47, 0x01, 0x37, 0x02, 0x42, 0xaf, 0x03, 0xef,
    0x00, 0x2f, 0x1e, 0x02, 0xcf, 0x03, 0x37, 0x03, 0x42, 0xd7, 0x02, 0x50, 0x0f, 0x03, 0xaf, 0x07, 0x81, 0x37, 0x03,
    0x42, 0xd7, 0x03, 0x50, 0x43, 0x8f, 0x05, 0x58, 0x47, 0x01, 0x37, 0x01, 0x3f, 0x0a, 0x42, 0xd7, 0x02, 0x50, 0x1a,
    0x1a, 0x47, 0x03, 0x37, 0x01, 0x07, 0x01, 0xdf, 0x01, 0x2f, 0x3d, 0x43, 0xef, 0x02, 0x2f, 0x38, 0x42, 0xaf, 0x01,
    0xdf, 0x01, 0x2f, 0x15, 0x42, 0xd7, 0x01, 0x50, 0x0f, 0x02, 0x42, 0xaf, 0x01, 0x81, 0x81, 0x37, 0x02, 0x42, 0xd7,
    0x01, 0x50, 0x1a, 0x1a, 0x3f, 0x1a, 0x42, 0xd7, 0x01, 0x50, 0x07, 0x02, 0xcf, 0x02, 0x48, 0x42, 0xaf, 0x03, 0x81,
    0x37, 0x02, 0x42, 0xd7, 0x02, 0x50, 0x1a, 0x1a, 0x1a, 0x47, 0x02, 0x37, 0x01, 0x3f, 0xbd, 0x07, 0x01, 0xdf, 0x02,
    0x2f, 0x39, 0x07, 0x03, 0xeb, 0x27, 0x34, 0x42, 0x37, 0x06, 0x43, 0x37, 0x07, 0x0f, 0x03, 0x47, 0x01, 0xc9, 0x58,
    0x02, 0xaa, 0x83, 0x58, 0x0f, 0x04, 0x41, 0x8b, 0x50, 0x1f, 0x02, 0x43, 0xef, 0x00, 0x2f, 0x08, 0x1a, 0x45, 0x60,
    0x11, 0x09, 0x39, 0x3f, 0xf3, 0x41, 0x37, 0x04, 0x07, 0x06, 0x0f, 0x03, 0xd1, 0x50, 0x07, 0x07, 0x89, 0x58, 0x04,
    0x37, 0x01, 0x07, 0x01, 0xdf, 0x03, 0x2f, 0x2b, 0x43, 0xef, 0x01, 0x2f, 0x26, 0x42, 0xaf, 0x01, 0xdf, 0x01, 0x2f,
    0x14, 0x42, 0xd7, 0x01, 0x50, 0x0f, 0x02, 0xaf, 0x01, 0x81, 0x81, 0x37, 0x02, 0x42, 0xd7, 0x01, 0x50, 0x1a, 0x1a,
    0x3f, 0x09, 0x42, 0xd7, 0x01, 0x50, 0x1a, 0x47, 0x04, 0x37, 0x01, 0x3f, 0xcf, 0x07, 0x01, 0xdf, 0x04, 0x2f, 0x22,
    0x43, 0xef, 0x07, 0x2f, 0x1d, 0x0f, 0x04, 0x42, 0x60, 0x39, 0x09, 0x41, 0x37, 0x04, 0x42, 0xd7, 0x08, 0x50, 0x43,
    0x8f, 0x08, 0x58, 0x07, 0x02, 0x02, 0x37, 0x02, 0xdf, 0x00, 0x2f, 0x03, 0x04, 0x37, 0x01, 0x38, 0x00};
// The level-1 post-processor programs this engine decodes natively: what its own makeConfig + compiler emit for
// rb = 0..7 raw offset bits (blocks of 16 MiB << rb) with and without the E8E9 inverse.  rb = 0 / no E8E9 is the
// golden 302-byte program; the others are pinned by decode parity (the reference's LZBuffer stream under these
// programs is restored by the reference's own PostProcessor: tests/test_config_cpu.py).
const std::vector<u8>& zpq_known_pcomp(u32 rb, bool e8) {
  struct Tab { std::vector<u8> t[8][2]; };
  // function-local static with an initialiser: built once, also when several contexts decode for the first time at
  // the same moment (calls may come from any host thread)
  static const Tab tab = [] {
    Tab T;
    for (u32 r = 0; r < 8; ++r)
      for (int e = 0; e < 2; ++e) {
        char m[64];
        snprintf(m, sizeof m, "x%u,%d,5,0,3,24", 4 + r, e ? 5 : 1);
        std::string xm; int args[9]; std::vector<u8> hdr;
        if (zpq_build_config(nullptr, m, nullptr, 0, &xm, args, &hdr, &T.t[r][e]) != ZPQ_OK) T.t[r][e].clear();
      }
    return T;
  }();
  static const std::vector<u8> none;
  return rb < 8 ? tab.t[rb][e8 ? 1 : 0] : none;
}

namespace {
// The post-processor programs of levels 2 (byte-aligned LZ77), 3 (BWT) and of E8E9 alone, as this engine's makeConfig
// emits them (decode-pinned: the reference's PostProcessor restores the input under them, tests/test_pcomp_variants_cpu.py):
// the decode side recognises them byte for byte and runs native kernels instead of the ZPAQL machine.
struct KnownPre { int kind; bool e8; u32 mm; };     // kind: 2 = LZ77 level 2, 3 = BWT, 4 = E8E9 only
std::vector<u8> pcomp_of(const std::string& method) {
  static std::mutex mu;
  static std::map<std::string, std::vector<u8>> cache;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(method);
  if (it != cache.end()) return it->second;
  std::string xm; int args[9]; std::vector<u8> hdr, pc;
  if (zpq_build_config(nullptr, method.c_str(), nullptr, 0, &xm, args, &hdr, &pc) != ZPQ_OK) pc.clear();
  return cache[method] = pc;
}
bool match_known_pre(const std::vector<u8>& pc, u32 ph, u32 pm, KnownPre* out) {
  if (pm < 20 || pm > 31) return false;
  const std::string a0 = std::to_string(pm - 20);
  for (int e8 = 0; e8 < 2; ++e8) {
    if (ph == 0) {
      if (e8 && pc == pcomp_of("x" + a0 + ",4c0")) { *out = {4, true, 0}; return true; }
      const std::string tail = ",0,7," + std::to_string(21 + (int)pm - 20) + ",1c0";
      const std::vector<u8> p1 = pcomp_of("x" + a0 + "," + std::to_string(e8 ? 6 : 2) + ",1" + tail), p2 = pcomp_of("x" + a0 + "," + std::to_string(e8 ? 6 : 2) + ",2" + tail);
      if (p1.size() == pc.size() && p1.size() == p2.size() && !p1.empty()) {
        size_t at = 0, ndiff = 0;
        for (size_t i = 0; i < p1.size(); ++i) if (p1[i] != p2[i]) { at = i; ++ndiff; }
        if (ndiff == 1) {
          const u32 mm = pc[at];
          if (pc == pcomp_of("x" + a0 + "," + std::to_string(e8 ? 6 : 2) + "," + std::to_string(mm) + tail)) { *out = {2, e8 != 0, mm}; return true; }
        }
      }
    } else if (ph == pm) {
      if (pc == pcomp_of("x" + a0 + "," + std::to_string(e8 ? 7 : 3) + "c0")) { *out = {3, e8 != 0, 0}; return true; }
    }
  }
  return false;
}
#define kPcompLz1 zpq_pcomp_lz1

enum Kind { KIND_STORE0 = 0, KIND_STOREX = 1, KIND_LZ1 = 2 /* LZ77 codes, bit packed (level 1) or byte aligned (level 2) */, KIND_BWT = 3 };

struct Config {
  Kind kind;           // what produces the bytes the Encoder sees after the preamble
  int args[9];
  std::vector<u8> header;   // hsize[2] hh hm ph pm n COMP 0 HCOMP 0 (config.hip, == libzpaq::Compiler)
  std::vector<u8> pcomp;    // post-processor bytecode, empty = PASS
  u32 ncomp;                // > 0: the Encoder is the arithmetic coder over the context-mixing model
  bool e8;                  // E8E9 applied to the block before LZ77 (args[1] = 5); the post-processor undoes it
  std::string xmethod;
};

// Method string -> configuration: compressBlock's digit expansion, makeConfig and the ZPAQL compiler
// (config.hip).  `host_data` is only needed for levels 5..9 (search for periodic structure).
int parse_method(zpq_ctx* ctx, const char* method, const u8* host_data, u32 n, Config* cfg) {
  if (!method || !method[0]) return zpq_fail(ctx, ZPQ_ERR_ARG, "empty method");
  int rc = zpq_build_config(ctx, method, host_data, n, &cfg->xmethod, cfg->args, &cfg->header, &cfg->pcomp);
  if (rc) return rc;
  const std::string& m = cfg->xmethod;
  cfg->ncomp = cfg->header.size() > 6 ? cfg->header[6] : 0;
  const int pre = cfg->args[1], level = pre & 3;
  cfg->e8 = m[0] != '0' && pre >= 4 && pre <= 7;
  if (m[0] == '0') cfg->kind = KIND_STORE0;
  else if (pre > 7) return zpq_fail(ctx, ZPQ_ERR_METHOD, "method '%s': pre-processor %d does not exist", m.c_str(), pre);
  else if (level == 0) cfg->kind = KIND_STOREX;                                      // model only (x,0) or E8E9 + model (x,4)
  else if (level == 1 && cfg->args[0] <= 6) cfg->kind = KIND_LZ1;                    // blocks up to 64 MiB (rb = 0..2)
  else if (level == 2 && cfg->args[0] <= 6) cfg->kind = KIND_LZ1;                    // byte-aligned codes: over the suffix array (N6 - N1 >= 21) or the hash table
  else if (level == 3) cfg->kind = KIND_BWT;
  else return zpq_fail(ctx, ZPQ_ERR_METHOD, "method '%s' not implemented", m.c_str());
  if (cfg->kind != KIND_STORE0 && (u64)n > (1ull << (20 + cfg->args[0])))
    return zpq_fail(ctx, ZPQ_ERR_ARG, "block larger than 2^%d", 20 + cfg->args[0]);
  if (cfg->kind == KIND_LZ1) return zpq_lz77_check_args(ctx, cfg->args, n);      // (e.g. a secondary context: refused for THIS block)
  return ZPQ_OK;
}

// tag, "zPQ", level (2 when there are no components: such headers need a level-2 reader), type 1,
// header, segment start (Compressor::startBlock / startSegment as read back by Decompresser,
// ZSFX/libzpaq.cpp:2239-2330)
void build_prefix(std::vector<u8>& o, const Config& c, const char* filename, const char* comment, u32 n) {
  o.insert(o.end(), kTag, kTag + 13);
  o.push_back('z'); o.push_back('P'); o.push_back('Q'); o.push_back(c.ncomp ? 1 : 2); o.push_back(1);
  o.insert(o.end(), c.header.begin(), c.header.end());
  o.push_back(1);
  if (filename) o.insert(o.end(), filename, filename + strlen(filename));
  o.push_back(0);
  char sz[32];
  snprintf(sz, sizeof sz, "%u", n);
  o.insert(o.end(), sz, sz + strlen(sz));
  if (comment) { o.push_back(' '); o.insert(o.end(), comment, comment + strlen(comment)); }
  o.push_back(0);
  o.push_back(0);
}

struct FrameDev {
  u8* out;            // framed block
  const u8* prefix;   // device copy of prefix + preamble
  u32 prefix_len;     // bytes before the first sub-block
  u32 pre_len;        // post-processor preamble bytes that open the payload
  const u8* data;     // payload body (raw input or LZ stream)
  u32 data_len;
  const u8* digest;   // device, 20 bytes, or null
};

// payload byte j lands at prefix_len + 4*(j/65536 + 1) + j; thread j also writes the length field
// of the sub-block it opens and, for j == P-1, the terminator, checksum record and end-of-block.
__global__ __launch_bounds__(256) void frame_kernel(const FrameDev* __restrict__ jobs) {
  const FrameDev F = jobs[blockIdx.y];
  const u32 P = F.pre_len + F.data_len;
  if (blockIdx.x == 0)
    for (u32 i = threadIdx.x; i < F.prefix_len; i += 256) F.out[i] = F.prefix[i];
  const u32 j = blockIdx.x * 256u + threadIdx.x;
  if (j >= P) return;
  const u32 sb = j >> 16;
  u8* o = F.out + F.prefix_len + 4u * (sb + 1) + j;
  *o = j < F.pre_len ? F.prefix[F.prefix_len + j] : F.data[j - F.pre_len];
  if ((j & 65535u) == 0) {
    const u32 k = P - j < 65536u ? P - j : 65536u;
    o[-4] = (u8)(k >> 24); o[-3] = (u8)(k >> 16); o[-2] = (u8)(k >> 8); o[-1] = (u8)k;
  }
  if (j == P - 1) {
    u8* t = o + 1;
    t[0] = t[1] = t[2] = t[3] = 0;
    if (F.digest) { t[4] = 253; for (int i = 0; i < 20; ++i) t[5 + i] = F.digest[i]; t[25] = 255; }
    else { t[4] = 254; t[5] = 255; }
  }
}

u32 framed_size(u32 prefix_len, u32 P, bool sha) {
  const u32 nsb = (P + 65535u) / 65536u;
  return prefix_len + P + 4u * nsb + 4u + (sha ? 21u : 1u) + 1u;
}

}  // namespace

extern "C" size_t zpq_block_bound(size_t n, const char* filename, const char* comment) {
  // stored/LZ77 framing, or an arithmetic-coded stream (which may grow a little on incompressible input)
  size_t p = zpq_lz77_bound(n) + 3 + 302 + n / 16 + 1024;
  return 13 + 5 + 512 /* header: makeConfig's largest is 257 */ + 1 + (filename ? strlen(filename) : 0) + 1 + 24 + (comment ? strlen(comment) + 1 : 0) + 2 + p +
         4 * (p / 65536 + 2) + 4 + 21 + 1 + 64;
}

extern "C" int zpq_compress_blocks_dev(zpq_ctx* ctx, zpq_block_job* jobs, size_t njobs) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (njobs == 0) return ZPQ_OK;
  hipStream_t st = ctx->stream;
  std::vector<Config> cfg(njobs);
  std::vector<std::vector<u8>> prefix(njobs);
  std::vector<zpq_lz77_job> lz;
  std::vector<size_t> lz_of(njobs, (size_t)-1);
  size_t lz_out_total = 0, prefix_total = 0;
  int first_err = ZPQ_OK;
  std::vector<u8> hostbuf;
  std::vector<size_t> e8_jobs;          // jobs whose input is E8E9-transformed (on a copy) before LZ77
  size_t e8_total = 0;
  std::vector<size_t> cm_jobs;          // jobs whose Encoder is the context-mixing coder
  size_t encin_total = 0;
  std::vector<size_t> bwt_jobs;         // jobs whose body is the Burrows-Wheeler transform of the input (level 3)
  size_t bwt_total = 0;
  std::vector<const u8*> body_ptr(njobs, nullptr);     // what the Encoder sees after the preamble
  std::vector<u32> body_len(njobs, 0);
  for (size_t i = 0; i < njobs; ++i) {
    jobs[i].out_len = 0;
    body_ptr[i] = jobs[i].in; body_len[i] = jobs[i].n;
    const u8* host_data = nullptr;
    if (jobs[i].method && jobs[i].method[0] >= '5' && jobs[i].method[0] <= '9' && jobs[i].n) {
      hostbuf.resize(jobs[i].n);        // level 5 looks at the data to pick periodic models (host logic in libzpaq too)
      ZPQ_HIP(ctx, hipMemcpyAsync(hostbuf.data(), jobs[i].in, jobs[i].n, hipMemcpyDeviceToHost, st));
      ZPQ_HIP(ctx, hipStreamSynchronize(st));
      host_data = hostbuf.data();
    }
    jobs[i].status = parse_method(ctx, jobs[i].method, host_data, jobs[i].n, &cfg[i]);
    if (jobs[i].status == ZPQ_OK && jobs[i].out_cap + 512 < zpq_block_bound(jobs[i].n, jobs[i].filename, jobs[i].comment) + cfg[i].header.size())
      jobs[i].status = zpq_fail(ctx, ZPQ_ERR_CAPACITY, "job %zu: out_cap below zpq_block_bound", i);
    if (jobs[i].status != ZPQ_OK) { if (!first_err) first_err = jobs[i].status; continue; }
    build_prefix(prefix[i], cfg[i], jobs[i].filename, jobs[i].comment, jobs[i].n);
    const u32 plen = (u32)prefix[i].size();
    if (!cfg[i].pcomp.empty()) {   // postProcess(): 1, psize lo, psize hi, pcomp
      const size_t ps = cfg[i].pcomp.size();
      prefix[i].push_back(1); prefix[i].push_back((u8)(ps & 255)); prefix[i].push_back((u8)(ps >> 8));
      prefix[i].insert(prefix[i].end(), cfg[i].pcomp.begin(), cfg[i].pcomp.end());
    } else {
      prefix[i].push_back(0);        // PASS
    }
    size_t body_cap = jobs[i].n;
    if (cfg[i].e8) { e8_jobs.push_back(i); e8_total += (((size_t)jobs[i].n + 64 + 63) & ~(size_t)63) + ((((size_t)jobs[i].n + 31) / 32 * 4 + 63) & ~(size_t)63); }
    if (cfg[i].kind == KIND_BWT) {
      bwt_jobs.push_back(i);
      body_cap = (size_t)jobs[i].n + 5;
      bwt_total += (body_cap + 64 + 63) & ~(size_t)63;
    }
    if (cfg[i].kind == KIND_LZ1) {
      lz_of[i] = lz.size();
      zpq_lz77_job j;
      memset(&j, 0, sizeof j);
      j.d_in = jobs[i].in; j.n = jobs[i].n;
      for (int k = 0; k < 9; ++k) j.args[k] = cfg[i].args[k];
      j.out_cap = (u32)((zpq_lz77_bound(jobs[i].n) + 15) & ~(size_t)15);
      lz_out_total += j.out_cap;
      body_cap = j.out_cap;
      lz.push_back(j);
    }
    if (cfg[i].ncomp) {
      cm_jobs.push_back(i);
      encin_total += (prefix[i].size() - plen + body_cap + 64 + 15) & ~(size_t)15;
    }
    prefix[i].push_back((u8)(plen & 255)); prefix[i].push_back((u8)(plen >> 8));  // trailer: prefix_len (host bookkeeping)
    prefix_total += (prefix[i].size() + 15) & ~(size_t)15;
  }
  // block SHA-1 chains on the second stream (one lane per block; overlaps the LZ77 parse)
  std::vector<u64> sha_off; std::vector<u32> sha_len; std::vector<size_t> sha_job;
  for (size_t i = 0; i < njobs; ++i)
    if (jobs[i].status == ZPQ_OK && jobs[i].dosha1) { sha_off.push_back((u64)(uintptr_t)jobs[i].in); sha_len.push_back(jobs[i].n); sha_job.push_back(i); }
  u8* d_aux = (u8*)zpq_scratch(ctx, 3, prefix_total + njobs * (sizeof(FrameDev) + 8 + 4 + 20) + 512);
  u8* d_lz = (u8*)zpq_scratch(ctx, 4, lz_out_total + 64);
  if (!d_aux || !d_lz) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "block scratch");
  u64* d_sha_off = (u64*)d_aux;
  u32* d_sha_len = (u32*)(d_sha_off + njobs);
  u8* d_dig = (u8*)(d_sha_len + ((njobs + 3) & ~(size_t)3));
  FrameDev* d_frames = (FrameDev*)(d_dig + ((njobs * 20 + 15) & ~(size_t)15));
  u8* d_prefix = (u8*)(d_frames + njobs);
  if (!sha_job.empty()) {
    // everything the caller enqueued on the context stream (gathers, memsets that produce the block inputs)
    // happens before the checksum chains read them on the second stream
    ZPQ_HIP(ctx, hipEventRecord(ctx->ev2, st));
    ZPQ_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev2, 0));
    ZPQ_HIP(ctx, hipMemcpyAsync(d_sha_off, sha_off.data(), sha_off.size() * 8, hipMemcpyHostToDevice, ctx->stream2));
    ZPQ_HIP(ctx, hipMemcpyAsync(d_sha_len, sha_len.data(), sha_len.size() * 4, hipMemcpyHostToDevice, ctx->stream2));
    ZPQ_HIP(ctx, hipStreamSynchronize(ctx->stream2));
    int rc = zpq_sha1_chains_on(ctx, ctx->stream2, (const u8*)0, d_sha_off, d_sha_len, sha_job.size(), d_dig);
    if (rc) return rc;
    ZPQ_HIP(ctx, hipEventRecord(ctx->ev, ctx->stream2));
  }
  // E8E9 front end: the transform runs on a copy (the block checksum and the caller see the original bytes)
  if (!e8_jobs.empty()) {
    u8* d_e8 = (u8*)zpq_scratch(ctx, 21, e8_total + 64);
    if (!d_e8) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "e8e9 scratch");
    size_t o = 0;
    for (size_t i : e8_jobs) {
      u8* copy = d_e8 + o; o += ((size_t)jobs[i].n + 64 + 63) & ~(size_t)63;
      u32* bits = (u32*)(d_e8 + o); o += (((size_t)jobs[i].n + 31) / 32 * 4 + 63) & ~(size_t)63;
      if (jobs[i].n) ZPQ_HIP(ctx, hipMemcpyAsync(copy, jobs[i].in, jobs[i].n, hipMemcpyDeviceToDevice, st));
      ZPQ_HIP(ctx, hipMemsetAsync(copy + jobs[i].n, 0, 64, st));
      int rc = zpq_e8e9_forward_launch(ctx, st, copy, jobs[i].n, bits);
      if (rc) return rc;
      if (cfg[i].kind == KIND_LZ1) lz[lz_of[i]].d_in = copy;
      body_ptr[i] = copy;                 // x,4: the transformed bytes go to the model as they are
    }
  }
  // LZ77 streams
  {
    size_t o = 0;
    for (auto& j : lz) { j.d_out = d_lz + o; o += j.out_cap; }
    int rc = zpq_lz77_encode_dev(ctx, lz.data(), lz.size());
    if (rc) return rc;
    for (size_t i = 0; i < njobs; ++i)
      if (jobs[i].status == ZPQ_OK && cfg[i].kind == KIND_LZ1) { body_ptr[i] = lz[lz_of[i]].d_out; body_len[i] = lz[lz_of[i]].out_len; }
  }
  // Burrows-Wheeler transforms (LZBuffer level 3: suffix array -> last column + index)
  if (!bwt_jobs.empty()) {
    u8* d_bwt = (u8*)zpq_scratch(ctx, 23, bwt_total + 64);
    if (!d_bwt) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "BWT scratch");
    size_t o = 0;
    for (size_t i : bwt_jobs) {
      u8* dst = d_bwt + o; o += ((size_t)jobs[i].n + 5 + 64 + 63) & ~(size_t)63;
      int rc = zpq_bwt_dev(ctx, body_ptr[i], jobs[i].n, dst);
      if (rc) return rc;
      body_ptr[i] = dst; body_len[i] = jobs[i].n + 5;
    }
  }
  // framing
  std::vector<FrameDev> fr(njobs);
  std::vector<u8> pre_all(prefix_total);
  size_t po = 0, nfr = 0;
  u32 maxP = 0;
  std::vector<size_t> fr_job;
  std::vector<size_t> sha_slot(njobs, (size_t)-1);
  for (size_t i = 0, s = 0; i < njobs; ++i)
    if (jobs[i].status == ZPQ_OK && jobs[i].dosha1) sha_slot[i] = s++;
  for (size_t i = 0; i < njobs; ++i) {
    if (jobs[i].status != ZPQ_OK) continue;
    std::vector<u8>& pv = prefix[i];
    const u32 plen = pv[pv.size() - 2] | (u32)pv[pv.size() - 1] << 8;
    pv.resize(pv.size() - 2);
    if (cfg[i].ncomp) continue;         // arithmetic-coded: assembled below
    memcpy(&pre_all[po], pv.data(), pv.size());
    FrameDev F;
    F.out = jobs[i].out; F.prefix = d_prefix + po; F.prefix_len = plen; F.pre_len = (u32)pv.size() - plen;
    F.data = body_ptr[i]; F.data_len = body_len[i];
    F.digest = jobs[i].dosha1 ? d_dig + 20 * sha_slot[i] : nullptr;
    const u32 P = F.pre_len + F.data_len;
    jobs[i].out_len = framed_size(plen, P, jobs[i].dosha1 != 0);
    if (jobs[i].out_len > jobs[i].out_cap) { jobs[i].status = ZPQ_ERR_CAPACITY; jobs[i].out_len = 0; if (!first_err) first_err = ZPQ_ERR_CAPACITY; continue; }
    if (P > maxP) maxP = P;
    po += (pv.size() + 15) & ~(size_t)15;
    fr[nfr++] = F;
    fr_job.push_back(i);
  }
  if (nfr) {
    ZPQ_HIP(ctx, hipMemcpyAsync(d_prefix, pre_all.data(), po, hipMemcpyHostToDevice, st));
    ZPQ_HIP(ctx, hipMemcpyAsync(d_frames, fr.data(), nfr * sizeof(FrameDev), hipMemcpyHostToDevice, st));
    if (!sha_job.empty()) ZPQ_HIP(ctx, hipStreamWaitEvent(st, ctx->ev, 0));
    ZPQ_LAUNCH(ctx, "frame_kernel", st, frame_kernel, dim3((maxP + 255) / 256, (unsigned)nfr), dim3(256), d_frames);
    ZPQ_HIP(ctx, hipGetLastError());
  }
  // context-mixing blocks: the Encoder sees preamble + body; its output goes straight behind the prefix
  if (!cm_jobs.empty()) {
    u8* d_encin = (u8*)zpq_scratch(ctx, 11, encin_total + 64);
    if (!d_encin) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "encoder input scratch");
    std::vector<zpq_cm_job> cj(cm_jobs.size());
    std::vector<u32> plens(cm_jobs.size());
    size_t eo = 0;
    for (size_t k = 0; k < cm_jobs.size(); ++k) {
      const size_t i = cm_jobs[k];
      const std::vector<u8>& pv = prefix[i];         // already without the bookkeeping trailer
      u32 plen = 0;
      {   // prefix length = up to and including the reserved 0 after the comment: recompute from the pieces
        std::vector<u8> tmp; build_prefix(tmp, cfg[i], jobs[i].filename, jobs[i].comment, jobs[i].n); plen = (u32)tmp.size();
      }
      plens[k] = plen;
      const u32 pre_len = (u32)pv.size() - plen;
      const u8* body = body_ptr[i];
      const u32 body_len_i = body_len[i];
      u8* e = d_encin + eo;
      ZPQ_HIP(ctx, hipMemcpyAsync(e, pv.data() + plen, pre_len, hipMemcpyHostToDevice, st));
      if (body_len_i) ZPQ_HIP(ctx, hipMemcpyAsync(e + pre_len, body, body_len_i, hipMemcpyDeviceToDevice, st));
      ZPQ_HIP(ctx, hipMemcpyAsync(jobs[i].out, pv.data(), plen, hipMemcpyHostToDevice, st));
      memset(&cj[k], 0, sizeof cj[k]);
      cj[k].header = cfg[i].header.data(); cj[k].header_len = (u32)cfg[i].header.size();
      cj[k].d_in = e; cj[k].n = pre_len + body_len_i;
      cj[k].d_out = jobs[i].out + plen; cj[k].out_cap = jobs[i].out_cap - plen - 32;
      eo += (pre_len + body_len_i + 64 + 15) & ~(size_t)15;
    }
    ZPQ_HIP(ctx, hipStreamSynchronize(st));
    int rc = zpq_cm_encode_dev(ctx, cj.data(), cj.size());
    if (rc && !first_err) first_err = rc;
    if (!sha_job.empty()) ZPQ_HIP(ctx, hipStreamWaitEvent(st, ctx->ev, 0));
    for (size_t k = 0; k < cm_jobs.size(); ++k) {
      const size_t i = cm_jobs[k];
      if (cj[k].status != ZPQ_OK) { jobs[i].status = cj[k].status; jobs[i].out_len = 0; if (!first_err) first_err = cj[k].status; continue; }
      u8* t = jobs[i].out + plens[k] + cj[k].out_len;       // the coder already wrote the four 0 bytes
      static const u8 m253 = 253, m254 = 254, m255 = 255;
      if (jobs[i].dosha1) {
        ZPQ_HIP(ctx, hipMemcpyAsync(t, &m253, 1, hipMemcpyHostToDevice, st));
        ZPQ_HIP(ctx, hipMemcpyAsync(t + 1, d_dig + 20 * sha_slot[i], 20, hipMemcpyDeviceToDevice, st));
        ZPQ_HIP(ctx, hipMemcpyAsync(t + 21, &m255, 1, hipMemcpyHostToDevice, st));
        jobs[i].out_len = plens[k] + cj[k].out_len + 22;
      } else {
        ZPQ_HIP(ctx, hipMemcpyAsync(t, &m254, 1, hipMemcpyHostToDevice, st));
        ZPQ_HIP(ctx, hipMemcpyAsync(t + 1, &m255, 1, hipMemcpyHostToDevice, st));
        jobs[i].out_len = plens[k] + cj[k].out_len + 2;
      }
    }
  }
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  return first_err;
}

extern "C" int zpq_compress_blocks(zpq_ctx* ctx, zpq_block_job* jobs, size_t njobs) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (njobs == 0) return ZPQ_OK;
  // stage host inputs into one device arena, run the device path, copy framed blocks back
  size_t in_total = 0, out_total = 0;
  for (size_t i = 0; i < njobs; ++i) {
    in_total += ((size_t)jobs[i].n + 31) & ~(size_t)15;
    out_total += ((size_t)jobs[i].out_cap + 15) & ~(size_t)15;
  }
  u8* d_in = (u8*)zpq_scratch(ctx, 5, in_total + 64);
  u8* d_out = (u8*)zpq_scratch(ctx, 6, out_total + 64);
  if (!d_in || !d_out) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "block staging");
  std::vector<zpq_block_job> dj(jobs, jobs + njobs);
  size_t io = 0, oo = 0;
  for (size_t i = 0; i < njobs; ++i) {
    dj[i].in = d_in + io; dj[i].out = d_out + oo;
    if (jobs[i].n) ZPQ_HIP(ctx, hipMemcpyAsync(d_in + io, jobs[i].in, jobs[i].n, hipMemcpyHostToDevice, ctx->stream));
    io += ((size_t)jobs[i].n + 31) & ~(size_t)15;
    oo += ((size_t)jobs[i].out_cap + 15) & ~(size_t)15;
  }
  ZPQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  int rc = zpq_compress_blocks_dev(ctx, dj.data(), njobs);
  for (size_t i = 0; i < njobs; ++i) {
    jobs[i].status = dj[i].status; jobs[i].out_len = dj[i].out_len;
    if (dj[i].status == ZPQ_OK)
      ZPQ_HIP(ctx, hipMemcpyAsync(jobs[i].out, dj[i].out, dj[i].out_len, hipMemcpyDeviceToHost, ctx->stream));
  }
  ZPQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return rc;
}

// ---- decode side ----------------------------------------------------------------------------------------
// Host-parsed path: jobs[].in are host pointers.  unblock.hip routes here the blocks its device-side parser does
// not take (context-model coded data, PCOMP programs other than the level-1 LZ77 one, odd framing).
int zpq_decompress_hostparsed(zpq_ctx* ctx, zpq_unblock_job* jobs, size_t njobs, int verify, bool out_dev) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (njobs == 0) return ZPQ_OK;
  hipStream_t st = ctx->stream;
  // kind: 0 stored+PASS, 2 stored + the known LZ77-L1 PCOMP (native decoder), 3 generic (context-model
  // coded and/or an arbitrary PCOMP: cm.hip decoder + ZPAQL interpreter)
  // A block of several segments (ZSFX/libzpaq.cpp:2307-2337: the decoder and the post-processor are initialised for the
  // first one only) is always kind 3: payload = the segments' coded streams (or stored bytes) back to back, seg_len = the
  // bytes of each, seg_dec_end / seg_out_end = where each one's decoded stream / output ends, seg_sha = has_sha + SHA-1 each.
  struct Parsed { u32 kind; bool e8; u32 pay_off, pay_len; int has_sha; u8 sha[20]; u32 rb; std::vector<u8> payload;
                  std::vector<u8> header; u32 ncomp, ph, pm;
                  u32 nseg; std::vector<u32> seg_len, seg_dec_end, seg_out_end; std::vector<u8> seg_sha; };
  std::vector<Parsed> ps(njobs);
  int first_err = ZPQ_OK;
  size_t in_total = 0, out_total = 0;
  // 1. host: parse the framing exactly as Decompresser::findBlock/findFilename/readComment/
  //    decompress(stored)/readSegmentEnd do (ZSFX/libzpaq.cpp:2239-2366)
  for (size_t i = 0; i < njobs; ++i) {
    zpq_unblock_job& j = jobs[i];
    j.out_len = 0; j.consumed = 0; j.status = ZPQ_OK;
    const u8* a = j.in; const u32 n = j.n;
    auto bad = [&](int code, const char* why) { j.status = zpq_fail(ctx, code, "block %zu: %s", i, why); if (!first_err) first_err = j.status; };
    u32 p = 0;
    if (n < 13 + 5 + 2 || memcmp(a, kTag, 13) != 0) { bad(ZPQ_ERR_FORMAT, "no block tag"); continue; }
    p = 13;
    if (a[p] != 'z' || a[p + 1] != 'P' || a[p + 2] != 'Q' || (a[p + 3] != 1 && a[p + 3] != 2) || a[p + 4] != 1) { bad(ZPQ_ERR_FORMAT, "bad block header"); continue; }
    p += 5;
    const u32 hsize = a[p] | (u32)a[p + 1] << 8;
    if (p + 2 + hsize > n || hsize < 7) { bad(ZPQ_ERR_FORMAT, "truncated header"); continue; }
    Parsed& P = ps[i];
    P.ncomp = a[p + 6]; P.ph = a[p + 4]; P.pm = a[p + 5];
    P.header.assign(a + p, a + p + 2 + hsize);
    const u32 pm = a[p + 5];
    p += 2 + hsize;
    // every segment of the block: 1 filename 0 comment 0 0, the coded data, 253 sha1[20] | 254; then 255
    P.nseg = 0;
    const char* why = nullptr; int why_code = ZPQ_ERR_FORMAT;
    for (;;) {
      if (p >= n || a[p] != 1) { why = "missing segment"; break; }
      ++p;
      while (p < n && a[p]) ++p; ++p;
      while (p < n && a[p]) ++p; ++p;
      if (p >= n || a[p] != 0) { why = "bad segment header"; break; }
      ++p;
      const size_t before = P.payload.size();
      if (P.ncomp) {
        // arithmetic-coded data ends with four 0 bytes (Decoder::skip, ZSFX/libzpaq.cpp:2150-2160)
        u32 q = p, curr = 0;
        while (curr == 0 && q < n) curr = a[q++];
        while (curr && q < n) curr = curr << 8 | a[q++];
        if (curr) { why = "unterminated coded data"; break; }
        // the coder's last byte (Encoder::flush) may itself be 0: then the scan stopped one short of the
        // real terminator.  What follows the terminator is 253 or 254, never 0.
        while (q < n && a[q] == 0) ++q;
        P.payload.insert(P.payload.end(), a + p, a + q);
        p = q;
      } else {
        bool ok = true;
        for (;;) {
          if (p + 4 > n) { ok = false; break; }
          const u32 k = (u32)a[p] << 24 | (u32)a[p + 1] << 16 | (u32)a[p + 2] << 8 | a[p + 3];
          p += 4;
          if (!k) break;
          if ((u64)p + k > n) { ok = false; break; }
          P.payload.insert(P.payload.end(), a + p, a + p + k);
          p += k;
        }
        if (!ok) { why = "truncated stored data"; break; }
      }
      if (P.nseg == 0 && P.payload.empty()) { why = "truncated stored data"; break; }      // the first segment holds at least the post-processor byte
      P.seg_len.push_back((u32)(P.payload.size() - before));
      P.seg_sha.resize(21 * (size_t)(P.nseg + 1), 0);
      if (p < n && a[p] == 253 && p + 21 <= n) { P.seg_sha[21 * (size_t)P.nseg] = 1; memcpy(&P.seg_sha[21 * (size_t)P.nseg + 1], a + p + 1, 20); p += 21; }
      else if (p < n && a[p] == 254) ++p;
      else { why = "missing segment end"; break; }
      ++P.nseg;
      if (p < n && a[p] == 255) break;
      if (P.nseg >= 65536) { why = "more than 65536 segments in a block"; why_code = ZPQ_ERR_METHOD; break; }
    }
    if (why) { bad(why_code, why); continue; }
    P.has_sha = P.seg_sha[0]; memcpy(P.sha, &P.seg_sha[1], 20);
    j.nseg = P.nseg;
    j.consumed = p + 1;
    P.rb = pm > 24 ? pm - 24 : 0; P.e8 = false;
    if (P.ncomp || P.nseg > 1) { P.kind = 3; P.pay_off = 0; }
    else if (P.payload[0] == 0) { P.kind = 0; P.pay_off = 1; }
    else {
      if (P.payload.size() < 3) { bad(ZPQ_ERR_FORMAT, "truncated PCOMP"); continue; }
      const u32 psize = P.payload[1] | (u32)P.payload[2] << 8;
      const std::vector<u8>& plain = zpq_known_pcomp(P.rb, false);
      if (!plain.empty() && psize == plain.size() && P.payload.size() >= 3 + psize && memcmp(&P.payload[3], plain.data(), psize) == 0) { P.kind = 2; P.pay_off = 3 + psize; }
      else { P.kind = 3; P.pay_off = 0; }   // (E8E9 variants: generic path below recognises them)
    }
    P.pay_len = (u32)P.payload.size() - P.pay_off;
    in_total += ((size_t)P.pay_len + 31) & ~(size_t)15;
    out_total += ((size_t)j.out_cap + 31) & ~(size_t)15;
  }
  u8* d_in = (u8*)zpq_scratch(ctx, 5, in_total + 64);
  u8* d_out = (u8*)zpq_scratch(ctx, 4, out_total + 64);
  if (!d_in || !d_out) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "decode staging");
  // 2. device: undo LZ77 (or pass), SHA-1 of the result
  std::vector<zpq_lz77_dec_job> dj;
  std::vector<size_t> dj_job;
  std::vector<u8*> outp(njobs, nullptr);
  size_t io = 0, oo = 0;
  for (size_t i = 0; i < njobs; ++i) {
    if (jobs[i].status != ZPQ_OK) continue;
    Parsed& P = ps[i];
    outp[i] = d_out + oo;
    if (P.kind == 3) {
      oo += ((size_t)jobs[i].out_cap + 31) & ~(size_t)15;
      continue;                       // handled below, one block at a time
    }
    if (P.kind == 0) {
      if (P.pay_len > jobs[i].out_cap) { jobs[i].status = ZPQ_ERR_CAPACITY; if (!first_err) first_err = ZPQ_ERR_CAPACITY; continue; }
      if (P.pay_len) ZPQ_HIP(ctx, hipMemcpyAsync(outp[i], &P.payload[P.pay_off], P.pay_len, hipMemcpyHostToDevice, st));
      jobs[i].out_len = P.pay_len;
    } else {
      if (P.pay_len) ZPQ_HIP(ctx, hipMemcpyAsync(d_in + io, &P.payload[P.pay_off], P.pay_len, hipMemcpyHostToDevice, st));
      zpq_lz77_dec_job d;
      memset(&d, 0, sizeof d);
      d.d_in = d_in + io; d.n = P.pay_len; d.rb = P.rb; d.d_out = outp[i]; d.out_cap = jobs[i].out_cap;
      dj.push_back(d); dj_job.push_back(i);
      io += ((size_t)P.pay_len + 31) & ~(size_t)15;
    }
    oo += ((size_t)jobs[i].out_cap + 31) & ~(size_t)15;
  }
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  if (!dj.empty()) {
    int rc = zpq_lz77_decode_dev(ctx, dj.data(), dj.size());
    if (rc) return rc;
    for (size_t k = 0; k < dj.size(); ++k) {
      jobs[dj_job[k]].out_len = dj[k].out_len;
      if (dj[k].status != ZPQ_OK) { jobs[dj_job[k]].status = dj[k].status; jobs[dj_job[k]].out_len = 0; if (!first_err) first_err = dj[k].status; }
    }
  }
  // generic blocks: [context-model decode] -> post-processor preamble -> PASS / LZ77 fast path / ZPAQL VM.
  // Every arithmetic-coded block of the call is decoded by ONE zpq_cm_decode_dev call: blocks sharing a header share
  // one specialised kernel and run side by side, a wave each (cm.hip).
  std::vector<u8*> dec_at(njobs, nullptr);
  std::vector<u32> dec_len_of(njobs, 0);
  {
    size_t total = 0;
    std::vector<size_t> off(njobs, 0);
    for (size_t i = 0; i < njobs; ++i) {
      if (jobs[i].status != ZPQ_OK || ps[i].kind != 3) continue;
      const size_t dcap = (size_t)jobs[i].out_cap + 65536 + 64;
      off[i] = total;
      total += ((dcap + 255) & ~(size_t)255) + ((ps[i].payload.size() + 128 + 255) & ~(size_t)255);
    }
    u8* d_all = total ? (u8*)zpq_scratch(ctx, 10, total + 256) : nullptr;
    if (total && !d_all) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "decode staging");
    std::vector<zpq_cm_job> cj; std::vector<size_t> cj_job;
    for (size_t i = 0; i < njobs; ++i) {
      if (jobs[i].status != ZPQ_OK || ps[i].kind != 3) continue;
      Parsed& P = ps[i];
      const size_t dcap = (size_t)jobs[i].out_cap + 65536 + 64;
      dec_at[i] = d_all + off[i];
      if (P.ncomp) {
        u8* d_coded = dec_at[i] + ((dcap + 255) & ~(size_t)255);
        ZPQ_HIP(ctx, hipMemcpyAsync(d_coded, P.payload.data(), P.payload.size(), hipMemcpyHostToDevice, st));
        zpq_cm_job c;
        memset(&c, 0, sizeof c);
        c.header = P.header.data(); c.header_len = (u32)P.header.size();
        c.d_in = d_coded; c.n = (u32)P.payload.size(); c.d_out = dec_at[i]; c.out_cap = (u32)dcap;
        if (P.nseg > 1) { P.seg_dec_end.assign(P.nseg, 0); c.nseg = P.nseg; c.seg_len = P.seg_len.data(); c.seg_out_end = P.seg_dec_end.data(); }
        cj.push_back(c); cj_job.push_back(i);
      } else if (P.payload.size() > dcap) {
        jobs[i].status = ZPQ_ERR_CAPACITY; jobs[i].out_len = 0; if (!first_err) first_err = ZPQ_ERR_CAPACITY;
      } else {
        ZPQ_HIP(ctx, hipMemcpyAsync(dec_at[i], P.payload.data(), P.payload.size(), hipMemcpyHostToDevice, st));
        dec_len_of[i] = (u32)P.payload.size();
        if (P.nseg > 1) { P.seg_dec_end.assign(P.nseg, 0); u32 e = 0; for (u32 q = 0; q < P.nseg; ++q) P.seg_dec_end[q] = (e += P.seg_len[q]); }
      }
    }
    if (!cj.empty()) {
      ZPQ_HIP(ctx, hipStreamSynchronize(st));
      const int rc = zpq_cm_decode_dev(ctx, cj.data(), cj.size());
      for (size_t k = 0; k < cj.size(); ++k) {
        const size_t i = cj_job[k];
        if (cj[k].status || (rc && rc != ZPQ_ERR_FORMAT && rc != ZPQ_ERR_CAPACITY)) {
          jobs[i].status = cj[k].status ? cj[k].status : rc; jobs[i].out_len = 0; if (!first_err) first_err = jobs[i].status;
        } else dec_len_of[i] = cj[k].out_len;
      }
    }
  }
  for (size_t i = 0; i < njobs; ++i) {
    if (jobs[i].status != ZPQ_OK || ps[i].kind != 3) continue;
    Parsed& P = ps[i];
    auto fail_job = [&](int code) { jobs[i].status = code; jobs[i].out_len = 0; if (!first_err) first_err = code; };
    u8* d_dec = dec_at[i];
    const u32 dec_len = dec_len_of[i];
    if (dec_len < 1) { fail_job(ZPQ_ERR_FORMAT); continue; }
    u8 pre[3] = {0, 0, 0};
    ZPQ_HIP(ctx, hipMemcpyAsync(pre, d_dec, dec_len < 3 ? dec_len : 3, hipMemcpyDeviceToHost, st));
    ZPQ_HIP(ctx, hipStreamSynchronize(st));
    if (P.nseg > 1) {
      // several segments: PASS copies each one's decoded stream; a program runs over all of them on ONE machine, with the
      // end-of-segment input after each (the programs compressBlock writes would also do segment by segment on the native
      // decoders, but what carries over is for the program to say: the ZPAQL machine is the definition)
      if (P.seg_dec_end.size() != P.nseg || P.seg_dec_end[P.nseg - 1] != dec_len) { fail_job(ZPQ_ERR_FORMAT); continue; }
      bool ends_ok = true;                 // what the coder reported must ascend (the SHA-1 extents are cut from it)
      for (u32 q = 1; q < P.nseg; ++q) ends_ok &= P.seg_dec_end[q] >= P.seg_dec_end[q - 1];
      if (!ends_ok) { fail_job(ZPQ_ERR_FORMAT); continue; }
      P.seg_out_end.assign(P.nseg, 0);
      if (pre[0] == 0) {
        // the post-processor's type byte lies in the FIRST segment (the reference: "Unexpected EOS", PostProcessor::write state 0,
        // ZSFX/libzpaq.cpp:2187-2192); a first segment of 0 decoded bytes would wrap every segment end below
        if (P.seg_dec_end[0] < 1) { fail_job(ZPQ_ERR_FORMAT); continue; }
        const u32 len = dec_len - 1;
        if (len > jobs[i].out_cap) { fail_job(ZPQ_ERR_CAPACITY); continue; }
        if (len) ZPQ_HIP(ctx, hipMemcpyAsync(outp[i], d_dec + 1, len, hipMemcpyDeviceToDevice, st));
        for (u32 q = 0; q < P.nseg; ++q) P.seg_out_end[q] = P.seg_dec_end[q] - 1;
        jobs[i].out_len = len;
      } else if (pre[0] == 1 && dec_len >= 3) {
        const u32 psize = pre[1] | (u32)pre[2] << 8;
        if (psize < 1 || 3 + psize > P.seg_dec_end[0]) { fail_job(ZPQ_ERR_FORMAT); continue; }      // (the program lies in the first segment)
        std::vector<u8> pc(psize);
        ZPQ_HIP(ctx, hipMemcpyAsync(pc.data(), d_dec + 3, psize, hipMemcpyDeviceToHost, st));
        ZPQ_HIP(ctx, hipStreamSynchronize(st));
        std::vector<u32> lens(P.nseg);
        for (u32 q = 0; q < P.nseg; ++q) lens[q] = P.seg_dec_end[q] - (q ? P.seg_dec_end[q - 1] : 3 + psize);
        u32 olen = 0;
        int rc = zpq_pcomp_run_segments_dev(ctx, pc.data(), psize, P.ph, P.pm, d_dec + 3 + psize, dec_len - 3 - psize, lens.data(), P.nseg, outp[i],
                                            jobs[i].out_cap, P.seg_out_end.data(), &olen);
        if (rc) { fail_job(rc); continue; }
        jobs[i].out_len = olen;
      } else { fail_job(ZPQ_ERR_FORMAT); continue; }
      // the segment ends cut the SHA-1 extents below: ascending and inside the output, whoever computed them
      bool out_ok = P.seg_out_end[P.nseg - 1] <= jobs[i].out_len;
      for (u32 q = 1; q < P.nseg; ++q) out_ok &= P.seg_out_end[q] >= P.seg_out_end[q - 1];
      if (!out_ok) { fail_job(ZPQ_ERR_FORMAT); continue; }
      for (u32 q = 0; q < P.nseg && q < jobs[i].seg_cap; ++q) if (jobs[i].seg_out_end) jobs[i].seg_out_end[q] = P.seg_out_end[q];
    } else if (pre[0] == 0) {                           // PASS
      const u32 len = dec_len - 1;
      if (len > jobs[i].out_cap) { fail_job(ZPQ_ERR_CAPACITY); continue; }
      if (len) ZPQ_HIP(ctx, hipMemcpyAsync(outp[i], d_dec + 1, len, hipMemcpyDeviceToDevice, st));
      jobs[i].out_len = len;
    } else if (pre[0] == 1 && dec_len >= 3) {
      const u32 psize = pre[1] | (u32)pre[2] << 8;
      if (psize < 1 || 3 + psize > dec_len) { fail_job(ZPQ_ERR_FORMAT); continue; }
      std::vector<u8> pc(psize);
      ZPQ_HIP(ctx, hipMemcpyAsync(pc.data(), d_dec + 3, psize, hipMemcpyDeviceToHost, st));
      ZPQ_HIP(ctx, hipStreamSynchronize(st));
      const u8* d_data = d_dec + 3 + psize; const u32 dlen = dec_len - 3 - psize;
      const std::vector<u8>& k0 = zpq_known_pcomp(P.rb, false);
      const std::vector<u8>& k1 = zpq_known_pcomp(P.rb, true);
      const bool is0 = !k0.empty() && psize == k0.size() && memcmp(pc.data(), k0.data(), psize) == 0;
      const bool is1 = !is0 && !k1.empty() && psize == k1.size() && memcmp(pc.data(), k1.data(), psize) == 0;
      if (is0 || is1) {
        zpq_lz77_dec_job d;
        memset(&d, 0, sizeof d);
        u8* d_tmp = nullptr;
        if (is1) {                      // LZ77 into a temporary, E8E9 inverse into the output
          d_tmp = (u8*)zpq_scratch(ctx, 22, (size_t)jobs[i].out_cap + 128);
          if (!d_tmp) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "decode staging");
        }
        d.d_in = d_data; d.n = dlen; d.rb = P.rb; d.d_out = is1 ? d_tmp : outp[i]; d.out_cap = jobs[i].out_cap;
        int rc = zpq_lz77_decode_dev(ctx, &d, 1);
        if (rc || d.status) { fail_job(d.status ? d.status : rc); continue; }
        if (is1) {
          ZPQ_HIP(ctx, hipMemsetAsync(d_tmp + d.out_len, 0, 64, st));
          rc = zpq_e8e9_inverse_dev(ctx, d_tmp, outp[i], d.out_len);
          if (rc) { fail_job(rc); continue; }
        }
        jobs[i].out_len = d.out_len;
      } else if (KnownPre kp; !getenv("ZPQ_PCOMP_GENERIC") && match_known_pre(pc, P.ph, P.pm, &kp)) {
        // byte-aligned LZ77 / BWT / E8E9 alone, undone by native kernels (the stage output goes to a temporary when an
        // E8E9 inverse follows)
        u8* d_tmp = nullptr;
        if (kp.e8 && kp.kind != 4) {
          d_tmp = (u8*)zpq_scratch(ctx, 22, (size_t)jobs[i].out_cap + 128);
          if (!d_tmp) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "decode staging");
        }
        u8* stage_out = d_tmp ? d_tmp : outp[i];
        u32 olen = 0;
        int rc = ZPQ_OK;
        if (kp.kind == 2) {
          zpq_lz77_dec_job d;
          memset(&d, 0, sizeof d);
          d.d_in = d_data; d.n = dlen; d.rb = 0x80000000u | kp.mm; d.d_out = stage_out; d.out_cap = jobs[i].out_cap;
          rc = zpq_lz77_decode_dev(ctx, &d, 1);
          if (!rc) rc = d.status;
          olen = d.out_len;
        } else if (kp.kind == 3) {
          rc = zpq_ibwt_dev(ctx, d_data, dlen, stage_out, jobs[i].out_cap, &olen);
        } else {
          if (dlen > jobs[i].out_cap) rc = ZPQ_ERR_CAPACITY;
          olen = dlen;
        }
        if (rc) { fail_job(rc); continue; }
        if (kp.e8) {
          const u8* src = kp.kind == 4 ? d_data : d_tmp;
          if (kp.kind != 4) ZPQ_HIP(ctx, hipMemsetAsync(d_tmp + olen, 0, 64, st));
          rc = zpq_e8e9_inverse_dev(ctx, src, outp[i], olen);
          if (rc) { fail_job(rc); continue; }
        }
        jobs[i].out_len = olen;
      } else {
        u32 olen = 0;
        int rc = zpq_pcomp_run_dev(ctx, pc.data(), psize, P.ph, P.pm, d_data, dlen, outp[i], jobs[i].out_cap, &olen);
        if (rc) { fail_job(rc); continue; }
        jobs[i].out_len = olen;
      }
    } else { fail_job(ZPQ_ERR_FORMAT); continue; }
  }
  for (size_t i = 0; i < njobs; ++i)
    if (jobs[i].status == ZPQ_OK && ps[i].nseg == 1 && jobs[i].seg_out_end && jobs[i].seg_cap) jobs[i].seg_out_end[0] = jobs[i].out_len;
  // SHA-1 of every segment's bytes (one wave per chain), compared with the stored one of that segment
  std::vector<u64> so; std::vector<u32> sl; std::vector<size_t> sj; std::vector<u32> ss;
  for (size_t i = 0; i < njobs; ++i)
    if (jobs[i].status == ZPQ_OK) {
      const Parsed& P = ps[i];
      if (P.nseg > 1)
        for (u32 q = 0; q < P.nseg; ++q) {
          const u32 from = q ? P.seg_out_end[q - 1] : 0;
          so.push_back((u64)(uintptr_t)(outp[i] + from)); sl.push_back(P.seg_out_end[q] - from); sj.push_back(i); ss.push_back(q);
        }
      else { so.push_back((u64)(uintptr_t)outp[i]); sl.push_back(jobs[i].out_len); sj.push_back(i); ss.push_back(0); }
    }
  if (!sj.empty()) {
    u8* d_sha = (u8*)zpq_scratch(ctx, 3, sj.size() * 32 + 256);
    if (!d_sha) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "decode staging");
    u64* d_so = (u64*)d_sha; u32* d_sl = (u32*)(d_so + sj.size()); u8* d_dg = (u8*)(d_sl + sj.size());
    ZPQ_HIP(ctx, hipMemcpyAsync(d_so, so.data(), so.size() * 8, hipMemcpyHostToDevice, st));
    ZPQ_HIP(ctx, hipMemcpyAsync(d_sl, sl.data(), sl.size() * 4, hipMemcpyHostToDevice, st));
    ZPQ_HIP(ctx, hipStreamSynchronize(st));
    int rc = zpq_sha1_chains_on(ctx, st, (const u8*)0, d_so, d_sl, sj.size(), d_dg);   // one wave per checksum
    if (rc) return rc;
    std::vector<u8> dg(sj.size() * 20);
    ZPQ_HIP(ctx, hipMemcpyAsync(dg.data(), d_dg, dg.size(), hipMemcpyDeviceToHost, st));
    for (size_t k = 0; k < sj.size(); ++k) {
      zpq_unblock_job& j = jobs[sj[k]];
      if (ss[k] == 0 && j.out_len) ZPQ_HIP(ctx, hipMemcpyAsync(j.out, outp[sj[k]], j.out_len, out_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    }
    ZPQ_HIP(ctx, hipStreamSynchronize(st));
    for (size_t k = 0; k < sj.size(); ++k) {
      zpq_unblock_job& j = jobs[sj[k]];
      const Parsed& P = ps[sj[k]];
      if (ss[k] == 0) memcpy(j.sha1, &dg[20 * k], 20);
      if (verify && P.seg_sha[21 * (size_t)ss[k]] && memcmp(&dg[20 * k], &P.seg_sha[21 * (size_t)ss[k] + 1], 20) != 0) { j.status = ZPQ_ERR_CHECKSUM; if (!first_err) first_err = ZPQ_ERR_CHECKSUM; }
    }
  }
  return first_err;
}
