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

#include "zpq_internal.h"

namespace {

const u8 kTag[13] = {0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3};

// LZ77 level-1 post-processor program for rb = 0 without E8E9: the 302 bytes every "-m1" block of
// up to 16 MiB carries (golden: i-blocks of the reference's AUTOTEST/sha256.zpaq; SURVEY.md
// Appendix D).  Stored after the 2-byte little-endian length 0x012e.
const u8 kPcompLz1[302] = {
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

int lg_host(u32 x) { int r = 0; while (x) ++r, x >>= 1; return r; }

enum Kind { KIND_STORE0 = 0, KIND_STOREX = 1, KIND_LZ1 = 2 };

struct Config {
  Kind kind;
  int args[9];
};

// Method string -> configuration, following compressBlock's digit expansion and makeConfig's
// argument scan (level-1 row pinned by the fixture for type 512).
int parse_method(zpq_ctx* ctx, const char* method, u32 n, Config* cfg) {
  if (!method || !method[0]) return zpq_fail(ctx, ZPQ_ERR_ARG, "empty method");
  std::string m(method);
  const int arg0 = std::max(lg_host(n + 4095) - 20, 0);
  if (m[0] >= '0' && m[0] <= '9' && m != "0") {
    int commas = 0, a[4] = {0, 0, 0, 0};
    for (size_t i = 1; i < m.size() && commas < 4; ++i) {
      if (m[i] == ',' || m[i] == '.') ++commas;
      else if (m[i] >= '0' && m[i] <= '9') a[commas] = a[commas] * 10 + m[i] - '0';
    }
    const unsigned type = commas == 0 ? 512u : (unsigned)(a[1] * 4 + a[2]);
    const int level = m[0] - '0';
    const int htsz = 19 + arg0 + (arg0 <= 6);
    char b[64];
    if (level == 0) snprintf(b, sizeof b, "0%d,0", arg0);
    else if (level == 1) {
      if (type & 2) return zpq_fail(ctx, ZPQ_ERR_METHOD, "E8E9 variant (exe hint) not implemented");
      if (type < 40) snprintf(b, sizeof b, "x%d,0", arg0);
      else if (type < 80) snprintf(b, sizeof b, "x%d,1,4,0,1,15", arg0);
      else if (type < 128) snprintf(b, sizeof b, "x%d,1,4,0,2,16", arg0);
      else if (type < 256) snprintf(b, sizeof b, "x%d,1,4,0,2,%d", arg0, htsz);
      else if (type < 960) snprintf(b, sizeof b, "x%d,1,5,0,3,%d", arg0, htsz);
      else snprintf(b, sizeof b, "x%d,1,6,0,3,%d", arg0, htsz);
    } else return zpq_fail(ctx, ZPQ_ERR_METHOD, "method level %d not implemented (levels 0 and 1 only)", level);
    m = b;
  }
  memset(cfg->args, 0, sizeof cfg->args);
  const char* p = m.c_str() + 1;
  int i = 0;
  while (i < 9 && ((*p >= '0' && *p <= '9') || *p == ',' || *p == '.')) {
    if (*p >= '0' && *p <= '9') cfg->args[i] = cfg->args[i] * 10 + *p - '0';
    else if (++i < 9) cfg->args[i] = 0;
    ++p;
  }
  if (*p) return zpq_fail(ctx, ZPQ_ERR_METHOD, "context-model components ('%s') not implemented", p);
  if (m[0] == '0') cfg->kind = KIND_STORE0;
  else if (m[0] == 'x' && cfg->args[1] == 0) cfg->kind = KIND_STOREX;
  else if (m[0] == 'x' && cfg->args[1] == 1 && cfg->args[0] <= 4) cfg->kind = KIND_LZ1;
  else return zpq_fail(ctx, ZPQ_ERR_METHOD, "method '%s' not implemented", m.c_str());
  if (cfg->kind != KIND_STORE0 && (u64)n > (1ull << (20 + cfg->args[0])))
    return zpq_fail(ctx, ZPQ_ERR_ARG, "block larger than 2^%d", 20 + cfg->args[0]);
  return ZPQ_OK;
}

// tag, "zPQ", level 2 (n == 0 components), type 1, header, segment start
void build_prefix(std::vector<u8>& o, const Config& c, const char* filename, const char* comment, u32 n) {
  o.insert(o.end(), kTag, kTag + 13);
  o.push_back('z'); o.push_back('P'); o.push_back('Q'); o.push_back(2); o.push_back(1);
  if (c.kind == KIND_STORE0) {
    const u8 h[9] = {7, 0, 0, 0, 0, 0, 0, 0, 0};                      // comp 0 0 0 0 0 hcomp end
    o.insert(o.end(), h, h + 9);
  } else {
    const u8 h[16] = {0x0e, 0, 9, 16, 0, (u8)(c.kind == KIND_LZ1 ? 20 + c.args[0] : 0), 0, 0,
                      0x12, 0x68, 0x87, 0xff, 0x58, 0x72, 0x38, 0};   // comp 9 16 0 pm 0 hcomp c-- *c=a a+= 255 d=a *d=c halt
    o.insert(o.end(), h, h + 16);
  }
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
  size_t p = zpq_lz77_bound(n) + 3 + 302;
  return 13 + 5 + 16 + 1 + (filename ? strlen(filename) : 0) + 1 + 24 + (comment ? strlen(comment) + 1 : 0) + 2 + p +
         4 * (p / 65536 + 2) + 4 + 21 + 1 + 64;
}

extern "C" int zpq_compress_blocks_dev(zpq_ctx* ctx, zpq_block_job* jobs, size_t njobs) {
  if (njobs == 0) return ZPQ_OK;
  hipStream_t st = ctx->stream;
  std::vector<Config> cfg(njobs);
  std::vector<std::vector<u8>> prefix(njobs);
  std::vector<zpq_lz77_job> lz;
  std::vector<size_t> lz_of(njobs, (size_t)-1);
  size_t lz_out_total = 0, prefix_total = 0;
  int first_err = ZPQ_OK;
  for (size_t i = 0; i < njobs; ++i) {
    jobs[i].out_len = 0;
    jobs[i].status = parse_method(ctx, jobs[i].method, jobs[i].n, &cfg[i]);
    if (jobs[i].status == ZPQ_OK && jobs[i].out_cap < zpq_block_bound(jobs[i].n, jobs[i].filename, jobs[i].comment))
      jobs[i].status = zpq_fail(ctx, ZPQ_ERR_CAPACITY, "job %zu: out_cap below zpq_block_bound", i);
    if (jobs[i].status != ZPQ_OK) { if (!first_err) first_err = jobs[i].status; continue; }
    build_prefix(prefix[i], cfg[i], jobs[i].filename, jobs[i].comment, jobs[i].n);
    const u32 plen = (u32)prefix[i].size();
    if (cfg[i].kind == KIND_LZ1) {   // postProcess(): 1, psize lo, psize hi, pcomp
      prefix[i].push_back(1); prefix[i].push_back(302 & 255); prefix[i].push_back(302 >> 8);
      prefix[i].insert(prefix[i].end(), kPcompLz1, kPcompLz1 + 302);
      lz_of[i] = lz.size();
      zpq_lz77_job j;
      memset(&j, 0, sizeof j);
      j.d_in = jobs[i].in; j.n = jobs[i].n;
      for (int k = 0; k < 9; ++k) j.args[k] = cfg[i].args[k];
      j.out_cap = (u32)((zpq_lz77_bound(jobs[i].n) + 15) & ~(size_t)15);
      lz_out_total += j.out_cap;
      lz.push_back(j);
    } else {
      prefix[i].push_back(0);        // PASS
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
    ZPQ_HIP(ctx, hipMemcpyAsync(d_sha_off, sha_off.data(), sha_off.size() * 8, hipMemcpyHostToDevice, ctx->stream2));
    ZPQ_HIP(ctx, hipMemcpyAsync(d_sha_len, sha_len.data(), sha_len.size() * 4, hipMemcpyHostToDevice, ctx->stream2));
    ZPQ_HIP(ctx, hipStreamSynchronize(ctx->stream2));
    int rc = zpq_sha1_chains_on(ctx, ctx->stream2, (const u8*)0, d_sha_off, d_sha_len, sha_job.size(), d_dig);
    if (rc) return rc;
    ZPQ_HIP(ctx, hipEventRecord(ctx->ev, ctx->stream2));
  }
  // LZ77 streams
  {
    size_t o = 0;
    for (auto& j : lz) { j.d_out = d_lz + o; o += j.out_cap; }
    int rc = zpq_lz77_encode_dev(ctx, lz.data(), lz.size());
    if (rc) return rc;
  }
  // framing
  std::vector<FrameDev> fr(njobs);
  std::vector<u8> pre_all(prefix_total);
  size_t po = 0, nfr = 0;
  u32 maxP = 0;
  std::vector<size_t> fr_job;
  for (size_t i = 0, s = 0; i < njobs; ++i) {
    if (jobs[i].status != ZPQ_OK) continue;
    std::vector<u8>& pv = prefix[i];
    const u32 plen = pv[pv.size() - 2] | (u32)pv[pv.size() - 1] << 8;
    pv.resize(pv.size() - 2);
    memcpy(&pre_all[po], pv.data(), pv.size());
    FrameDev F;
    F.out = jobs[i].out; F.prefix = d_prefix + po; F.prefix_len = plen; F.pre_len = (u32)pv.size() - plen;
    if (cfg[i].kind == KIND_LZ1) { F.data = lz[lz_of[i]].d_out; F.data_len = lz[lz_of[i]].out_len; }
    else { F.data = jobs[i].in; F.data_len = jobs[i].n; }
    F.digest = nullptr;
    if (jobs[i].dosha1) { F.digest = d_dig + 20 * s; ++s; }
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
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  return first_err;
}

extern "C" int zpq_compress_blocks(zpq_ctx* ctx, zpq_block_job* jobs, size_t njobs) {
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
extern "C" int zpq_decompress_blocks(zpq_ctx* ctx, zpq_unblock_job* jobs, size_t njobs, int verify) {
  if (njobs == 0) return ZPQ_OK;
  hipStream_t st = ctx->stream;
  // kind: 0 stored+PASS, 2 stored + the known LZ77-L1 PCOMP (native decoder), 3 generic (context-model
  // coded and/or an arbitrary PCOMP: cm.hip decoder + ZPAQL interpreter)
  struct Parsed { u32 kind; u32 pay_off, pay_len; int has_sha; u8 sha[20]; u32 rb; std::vector<u8> payload;
                  std::vector<u8> header; u32 ncomp, ph, pm; };
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
    if (p >= n || a[p] != 1) { bad(ZPQ_ERR_FORMAT, "missing segment"); continue; }
    ++p;
    while (p < n && a[p]) ++p; ++p;
    while (p < n && a[p]) ++p; ++p;
    if (p >= n || a[p] != 0) { bad(ZPQ_ERR_FORMAT, "bad segment header"); continue; }
    ++p;
    bool ok = true;
    if (P.ncomp) {
      // arithmetic-coded data ends with four 0 bytes (Decoder::skip, ZSFX/libzpaq.cpp:2150-2160)
      u32 q = p, curr = 0;
      while (curr == 0 && q < n) curr = a[q++];
      while (curr && q < n) curr = curr << 8 | a[q++];
      if (curr) { bad(ZPQ_ERR_FORMAT, "unterminated coded data"); continue; }
      P.payload.assign(a + p, a + q);
      p = q;
    } else
    for (;;) {
      if (p + 4 > n) { ok = false; break; }
      const u32 k = (u32)a[p] << 24 | (u32)a[p + 1] << 16 | (u32)a[p + 2] << 8 | a[p + 3];
      p += 4;
      if (!k) break;
      if (p + k > n) { ok = false; break; }
      P.payload.insert(P.payload.end(), a + p, a + p + k);
      p += k;
    }
    if (!ok || P.payload.empty()) { bad(ZPQ_ERR_FORMAT, "truncated stored data"); continue; }
    if (p < n && a[p] == 253 && p + 21 <= n) { P.has_sha = 1; memcpy(P.sha, a + p + 1, 20); p += 21; }
    else if (p < n && a[p] == 254) { P.has_sha = 0; ++p; }
    else { bad(ZPQ_ERR_FORMAT, "missing segment end"); continue; }
    if (p >= n || a[p] != 255) { bad(ZPQ_ERR_METHOD, "multi-segment block"); continue; }
    j.consumed = p + 1;
    P.rb = pm > 24 ? pm - 24 : 0;
    if (P.ncomp) { P.kind = 3; P.pay_off = 0; }
    else if (P.payload[0] == 0) { P.kind = 0; P.pay_off = 1; }
    else {
      if (P.payload.size() < 3) { bad(ZPQ_ERR_FORMAT, "truncated PCOMP"); continue; }
      const u32 psize = P.payload[1] | (u32)P.payload[2] << 8;
      if (psize == 302 && P.payload.size() >= 3 + 302 && memcmp(&P.payload[3], kPcompLz1, 302) == 0) { P.kind = 2; P.pay_off = 3 + 302; }
      else { P.kind = 3; P.pay_off = 0; }
    }
    P.pay_len = (u32)P.payload.size() - P.pay_off;
    in_total += ((size_t)P.pay_len + 31) & ~(size_t)15;
    out_total += ((size_t)j.out_cap + 31) & ~(size_t)15;
  }
  u8* d_in = (u8*)zpq_scratch(ctx, 5, in_total + 64);
  u8* d_out = (u8*)zpq_scratch(ctx, 4, out_total + 64);
  u8* d_aux = (u8*)zpq_scratch(ctx, 3, njobs * 32 + 256);
  if (!d_in || !d_out || !d_aux) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "decode staging");
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
  // generic blocks: [context-model decode] -> post-processor preamble -> PASS / LZ77 fast path / ZPAQL VM
  for (size_t i = 0; i < njobs; ++i) {
    if (jobs[i].status != ZPQ_OK || ps[i].kind != 3) continue;
    Parsed& P = ps[i];
    auto fail_job = [&](int code) { jobs[i].status = code; jobs[i].out_len = 0; if (!first_err) first_err = code; };
    const size_t dcap = (size_t)jobs[i].out_cap + 65536 + 64;
    u8* d_dec = (u8*)zpq_scratch(ctx, 10, dcap + (P.ncomp ? P.payload.size() + 128 : 0) + 256);
    if (!d_dec) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "decode staging");
    u32 dec_len = 0;
    if (P.ncomp) {
      u8* d_coded = d_dec + ((dcap + 255) & ~(size_t)255);
      ZPQ_HIP(ctx, hipMemcpyAsync(d_coded, P.payload.data(), P.payload.size(), hipMemcpyHostToDevice, st));
      ZPQ_HIP(ctx, hipStreamSynchronize(st));
      zpq_cm_job cj;
      memset(&cj, 0, sizeof cj);
      cj.header = P.header.data(); cj.header_len = (u32)P.header.size();
      cj.d_in = d_coded; cj.n = (u32)P.payload.size(); cj.d_out = d_dec; cj.out_cap = (u32)dcap;
      int rc = zpq_cm_decode_dev(ctx, &cj, 1);
      if (rc || cj.status) { fail_job(cj.status ? cj.status : rc); continue; }
      dec_len = cj.out_len;
    } else {
      if (P.payload.size() > dcap) { fail_job(ZPQ_ERR_CAPACITY); continue; }
      ZPQ_HIP(ctx, hipMemcpyAsync(d_dec, P.payload.data(), P.payload.size(), hipMemcpyHostToDevice, st));
      dec_len = (u32)P.payload.size();
    }
    if (dec_len < 1) { fail_job(ZPQ_ERR_FORMAT); continue; }
    u8 pre[3] = {0, 0, 0};
    ZPQ_HIP(ctx, hipMemcpyAsync(pre, d_dec, dec_len < 3 ? dec_len : 3, hipMemcpyDeviceToHost, st));
    ZPQ_HIP(ctx, hipStreamSynchronize(st));
    if (pre[0] == 0) {                                  // PASS
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
      if (psize == 302 && memcmp(pc.data(), kPcompLz1, 302) == 0) {
        zpq_lz77_dec_job d;
        memset(&d, 0, sizeof d);
        d.d_in = d_data; d.n = dlen; d.rb = P.rb; d.d_out = outp[i]; d.out_cap = jobs[i].out_cap;
        int rc = zpq_lz77_decode_dev(ctx, &d, 1);
        if (rc || d.status) { fail_job(d.status ? d.status : rc); continue; }
        jobs[i].out_len = d.out_len;
      } else {
        u32 olen = 0;
        int rc = zpq_pcomp_run_dev(ctx, pc.data(), psize, P.ph, P.pm, d_data, dlen, outp[i], jobs[i].out_cap, &olen);
        if (rc) { fail_job(rc); continue; }
        jobs[i].out_len = olen;
      }
    } else { fail_job(ZPQ_ERR_FORMAT); continue; }
  }
  std::vector<u64> so; std::vector<u32> sl; std::vector<size_t> sj;
  for (size_t i = 0; i < njobs; ++i)
    if (jobs[i].status == ZPQ_OK) { so.push_back((u64)(uintptr_t)outp[i]); sl.push_back(jobs[i].out_len); sj.push_back(i); }
  if (!sj.empty()) {
    u64* d_so = (u64*)d_aux; u32* d_sl = (u32*)(d_so + njobs); u8* d_dg = (u8*)(d_sl + njobs);
    ZPQ_HIP(ctx, hipMemcpyAsync(d_so, so.data(), so.size() * 8, hipMemcpyHostToDevice, st));
    ZPQ_HIP(ctx, hipMemcpyAsync(d_sl, sl.data(), sl.size() * 4, hipMemcpyHostToDevice, st));
    ZPQ_HIP(ctx, hipStreamSynchronize(st));
    int rc = zpq_sha1_extents_on(ctx, st, (const u8*)0, d_so, d_sl, sj.size(), d_dg);
    if (rc) return rc;
    std::vector<u8> dg(sj.size() * 20);
    ZPQ_HIP(ctx, hipMemcpyAsync(dg.data(), d_dg, dg.size(), hipMemcpyDeviceToHost, st));
    for (size_t k = 0; k < sj.size(); ++k) {
      zpq_unblock_job& j = jobs[sj[k]];
      if (j.out_len) ZPQ_HIP(ctx, hipMemcpyAsync(j.out, outp[sj[k]], j.out_len, hipMemcpyDeviceToHost, st));
    }
    ZPQ_HIP(ctx, hipStreamSynchronize(st));
    for (size_t k = 0; k < sj.size(); ++k) {
      zpq_unblock_job& j = jobs[sj[k]];
      memcpy(j.sha1, &dg[20 * k], 20);
      if (verify && ps[sj[k]].has_sha && memcmp(j.sha1, ps[sj[k]].sha, 20) != 0) { j.status = ZPQ_ERR_CHECKSUM; if (!first_err) first_err = ZPQ_ERR_CHECKSUM; }
    }
  }
  return first_err;
}
