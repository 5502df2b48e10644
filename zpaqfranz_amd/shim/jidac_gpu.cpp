// jidac_gpu.cpp -- the journaling archive engine (zpaqfranz `a` / `x` for the -m0/-m1 family) on top of the
// C ABI: SURVEY.md row a19 and section 8f-1.  Host C++ like the reference's Jidac; every byte-level operation
// (fragmenting, SHA-1, dedup, LZ77, framing, decoding, verification) is one of the zpq_* GPU calls.
//
// Format (reference reader: Jidac::read_archive, ZSFX/zsfx.cpp:1283-1627):
//   every block is a ZPAQ block named jDC<yyyymmddhhmmss><c|d|h|i><10 digits> with comment "<usize> jDC\x01"
//   c: csize[8]            total size of the d blocks that follow (ZSFX/zsfx.cpp:1432-1461); method "0"
//   d: fragment bytes, then usize[4] per fragment, 0[4], count[4] (:1468-1500 reads it back); method as given
//   h: bsize[4] + {sha1[20], usize[4]} per fragment of one d block, num = first fragment id (:1463-1500); "0"
//   i: {date[8], name, 0, na[4], attr[na], ni[4], ptr[ni][4]}... (:1506-1541); method "1"
// The add-side rules that live only in the missing zpaqfranz.cpp (block cut, R/t hints, attr extension,
// i-block flush threshold) follow zpaq 7.15 as far as recalled: "parity unpinned" (DESIGN.md section 2).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <map>
#include <new>
#include <stdexcept>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

#include "jidac_gpu.h"

namespace {

typedef std::vector<uint8_t> Bytes;

const uint8_t kTag[13] = {0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3};
const uint32_t kBlockLimit = (1u << 24) - 4096;
// d block size for a method "LB...": 2^(20+B) - 4096 with B the digits after the level; without them zpaq uses B = 4
// for levels 0 and 1 and B = 6 (64 MiB) above ("x"/"s" methods: the number after the letter).  Capped at what one
// compressBlock call accepts here (2^26).
uint32_t block_limit_for(const char* method) {
  if (!method || !method[0]) return kBlockLimit;
  const char* q = method + 1;
  int b = -1;
  if (*q >= '0' && *q <= '9') { b = 0; while (*q >= '0' && *q <= '9') b = b * 10 + (*q++ - '0'); }
  if (b < 0) b = (method[0] == '0' || method[0] == '1' || method[0] == 'x' || method[0] == 's') ? 4 : 6;
  if (b > 6) b = 6;
  return (1u << (20 + b)) - 4096;
}

void put32(Bytes& b, uint32_t x) { for (int i = 0; i < 4; ++i) b.push_back((uint8_t)(x >> (8 * i))); }
void put64(Bytes& b, uint64_t x) { for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(x >> (8 * i))); }
uint32_t get32(const uint8_t* p) { return p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
uint64_t get64(const uint8_t* p) { return get32(p) | (uint64_t)get32(p + 4) << 32; }

std::string block_name(int64_t date, char type, uint32_t num) {
  char b[40];
  snprintf(b, sizeof b, "jDC%014lld%c%010u", (long long)date, type, num);
  return b;
}

struct Sha1Key {
  uint8_t d[20];
  bool operator==(const Sha1Key& o) const { return memcmp(d, o.d, 20) == 0; }
};
struct Sha1Hash { size_t operator()(const Sha1Key& k) const { size_t h; memcpy(&h, k.d, sizeof h); return h; } };

int compress_host(zpq_ctx* ctx, const Bytes& in, const char* method, const std::string& name, Bytes& out) {
  zpq_block_job j;
  memset(&j, 0, sizeof j);
  Bytes buf(zpq_block_bound(in.size(), name.c_str(), "jDC\x01"));
  static const uint8_t empty[1] = {0};
  j.in = in.empty() ? empty : in.data(); j.n = (uint32_t)in.size();
  j.method = method; j.filename = name.c_str(); j.comment = "jDC\x01"; j.dosha1 = 1;
  j.out = buf.data(); j.out_cap = (uint32_t)buf.size();
  int rc = zpq_compress_blocks(ctx, &j, 1);
  if (rc) return rc;
  out.insert(out.end(), buf.begin(), buf.begin() + j.out_len);
  return ZPQ_OK;
}

// Several small blocks (the c / h / i blocks of one add) through ONE zpq_compress_blocks call: a call costs launches and
// host round trips whatever its size, and an add of thousands of files writes hundreds of index blocks.
struct HostBlock { Bytes in; std::string method, name; };
int compress_host_many(zpq_ctx* ctx, const std::vector<HostBlock>& hb, std::vector<Bytes>& outs) {
  outs.assign(hb.size(), Bytes());
  if (hb.empty()) return ZPQ_OK;
  std::vector<zpq_block_job> jobs(hb.size());
  static const uint8_t empty[1] = {0};
  for (size_t k = 0; k < hb.size(); ++k) {
    zpq_block_job& j = jobs[k];
    memset(&j, 0, sizeof j);
    outs[k].resize(zpq_block_bound(hb[k].in.size(), hb[k].name.c_str(), "jDC\x01"));
    j.in = hb[k].in.empty() ? empty : hb[k].in.data(); j.n = (uint32_t)hb[k].in.size();
    j.method = hb[k].method.c_str(); j.filename = hb[k].name.c_str(); j.comment = "jDC\x01"; j.dosha1 = 1;
    j.out = outs[k].data(); j.out_cap = (uint32_t)outs[k].size();
  }
  const int rc = zpq_compress_blocks(ctx, jobs.data(), jobs.size());
  if (rc) return rc;
  for (size_t k = 0; k < hb.size(); ++k) { if (jobs[k].status) return jobs[k].status; outs[k].resize(jobs[k].out_len); }
  return ZPQ_OK;
}

// ---- archive index as read back (HT / DT / Block of ZSFX/zsfx.cpp:651-698) ----------------------------------
struct Frag { Sha1Key sha1; uint32_t usize; };
struct DBlock { size_t offset; uint32_t csize; uint32_t first_frag; uint32_t nfrag; uint64_t usize; };
struct FileRec { int64_t date; std::string attr; std::vector<uint32_t> ptr; };
struct Index {
  std::vector<Frag> ht;                 // 1-based: ht[0] unused
  std::vector<DBlock> blocks;
  std::map<std::string, FileRec> files; // latest version wins; date 0 = deleted
  int versions;
  Index() : ht(1), versions(0) {}
};

// Parses one block's framing: name, comment, payload extent; returns total block size or 0.
struct RawBlock { std::string name, comment; size_t size; };
bool parse_block(const uint8_t* a, size_t n, RawBlock& rb) {
  if (n < 13 + 5 + 2 + 7 || memcmp(a, kTag, 13)) return false;
  size_t p = 13;
  if (a[p] != 'z' || a[p + 1] != 'P' || a[p + 2] != 'Q') return false;
  p += 5;
  const uint32_t hsize = a[p] | (uint32_t)a[p + 1] << 8;
  if (hsize < 7 || p + 2 + hsize > n) return false;           // the header must be whole before any of it is read
  const uint32_t ncomp = a[p + 6];
  p += 2 + hsize;
  if (p >= n || a[p] != 1) return false;
  ++p;
  size_t e = p; while (e < n && a[e]) ++e;
  rb.name.assign((const char*)a + p, e - p); p = e + 1;
  e = p; while (e < n && a[e]) ++e;
  rb.comment.assign((const char*)a + p, e - p); p = e + 1;
  if (p >= n || a[p] != 0) return false;
  ++p;
  if (ncomp) {
    uint32_t curr = 0;
    while (curr == 0 && p < n) curr = a[p++];
    while (curr && p < n) curr = curr << 8 | a[p++];
    while (p < n && a[p] == 0) ++p;      // the coder's own last byte may be 0 (the marker that follows never is)
  } else {
    for (;;) {
      if (p + 4 > n) return false;
      const uint32_t k = (uint32_t)a[p] << 24 | (uint32_t)a[p + 1] << 16 | (uint32_t)a[p + 2] << 8 | a[p + 3];
      p += 4;
      if (!k) break;
      p += k;
      if (p > n) return false;
    }
  }
  if (p < n && a[p] == 253) p += 21; else if (p < n && a[p] == 254) ++p; else return false;
  if (p >= n || a[p] != 255) return false;
  rb.size = p + 1;
  return true;
}

int decompress_host(zpq_ctx* ctx, const uint8_t* blk, size_t n, size_t usize, Bytes& out) {
  out.resize(usize + 64);
  zpq_unblock_job j;
  memset(&j, 0, sizeof j);
  j.in = blk; j.n = (uint32_t)n; j.out = out.data(); j.out_cap = (uint32_t)out.size();
  int rc = zpq_decompress_blocks(ctx, &j, 1, 1);
  if (rc) return rc;
  if (j.status) return j.status;
  out.resize(j.out_len);
  return ZPQ_OK;
}

// read_archive (ZSFX/zsfx.cpp:1283-1627), journaling blocks only.
// fetch != nullptr: `arc` is a host SHADOW of an archive that lies in HBM at d_arc -- fetch(lo, hi) makes [lo, hi) of it valid --
// and the d blocks are never looked at: after a c block the walk jumps over them by the size the c block holds, as the
// reference's own read_archive does (ZSFX/zsfx.cpp:1432-1461).
// Two passes (round 6): the walk over the framing decodes only the c blocks (eight bytes each: the jump), every h and i block of
// the archive is then decoded by ONE batched call -- an archive of Silesia x256 has 13 h and ~200 i blocks, and one device call
// per block (a launch, a sync, two copies each) was 80 ms of every extract -- and the tables are filled in archive order.
int read_index(zpq_ctx* ctx, const uint8_t* arc, size_t n, Index& ix, const std::function<int(size_t, size_t)>* fetch = nullptr,
               const uint8_t* d_arc = nullptr) {
  struct Pending { char type; uint32_t num; size_t pos, size, usize, data_offset; };
  std::vector<Pending> pend;
  size_t pos = 0;
  bool rest_fetched = false;
  while (pos < n) {
    RawBlock rb;
    if (fetch && !rest_fetched) { const int rc = (*fetch)(pos, std::min(n, pos + 4096)); if (rc) return rc; }
    if (!parse_block(arc + pos, n - pos, rb)) return ZPQ_ERR_FORMAT;
    if (rb.name.size() != 28 || rb.name.compare(0, 3, "jDC") != 0 || rb.comment.size() < 4 ||
        rb.comment.compare(rb.comment.size() - 4, 4, "jDC\x01") != 0)
      return ZPQ_ERR_FORMAT;
    const char type = rb.name[17];
    const uint64_t num64 = strtoull(rb.name.c_str() + 18, 0, 10);
    const uint64_t usize64 = strtoull(rb.comment.c_str(), 0, 10);
    // sizes come from the archive: bound them before they size anything (the reference checks num < 1 and
    // num + n > 0xffffffff at ZSFX/zsfx.cpp:1470-1480; blocks are < 4 GiB by format)
    if (usize64 > 0xffffffffull - 4096 || num64 > 0xffffffffull) return ZPQ_ERR_FORMAT;
    const uint32_t num = (uint32_t)num64;
    const size_t usize = (size_t)usize64;
    if (type == 'c') {
      Bytes os;
      int rc = decompress_host(ctx, arc + pos, rb.size, usize, os);
      if (rc) return rc;
      if (os.size() != usize || os.size() < 8) return ZPQ_ERR_FORMAT;
      const int64_t jmp = (int64_t)get64(os.data());
      if (jmp < 0) break;                       // incomplete transaction: roll back (ZSFX/zsfx.cpp:1436-1443)
      pend.push_back({'c', num, pos, rb.size, usize, pos + rb.size});
      if (fetch) {
        if ((uint64_t)jmp > n - pos - rb.size) return ZPQ_ERR_FORMAT;
        pos += (size_t)jmp;                      // (+ rb.size below): the first h block of this version
        if (!rest_fetched) {                     // everything behind the first version's d blocks: index blocks (and later versions)
          const int rc2 = (*fetch)(pos + rb.size, n); if (rc2) return rc2;
          rest_fetched = true;
        }
      }
    } else if (type == 'h' || type == 'i') {
      pend.push_back({type, num, pos, rb.size, usize, 0});
    }
    pos += rb.size;
  }
  // ---- every h and i block in one call
  std::vector<size_t> off(pend.size(), 0);
  size_t total = 0, nj = 0;
  for (size_t k = 0; k < pend.size(); ++k) if (pend[k].type != 'c') { off[k] = total; total += (pend[k].usize + 64 + 63) & ~(size_t)63; ++nj; }
  Bytes plain(total ? total : 1);
  if (nj) {
    std::vector<zpq_unblock_job> jobs(nj);
    struct DevBuf { zpq_ctx* c; void* p; ~DevBuf() { if (p) zpq_dev_free_pooled(c, p); } } dbuf{ctx, nullptr};
    if (d_arc) { const int rc = zpq_dev_alloc_pooled(ctx, total + 64, &dbuf.p); if (rc) return rc; }
    size_t j = 0;
    for (size_t k = 0; k < pend.size(); ++k) {
      if (pend[k].type == 'c') continue;
      memset(&jobs[j], 0, sizeof jobs[j]);
      jobs[j].in = (d_arc ? d_arc : arc) + pend[k].pos; jobs[j].n = (uint32_t)pend[k].size;
      jobs[j].out = (d_arc ? (uint8_t*)dbuf.p : plain.data()) + off[k]; jobs[j].out_cap = (uint32_t)pend[k].usize + 64;
      ++j;
    }
    // (device form: the blocks are decoded where they lie in HBM, one copy brings all of them over)
    int rc = d_arc ? zpq_decompress_blocks_dev(ctx, jobs.data(), nj, 1) : zpq_decompress_blocks(ctx, jobs.data(), nj, 1);
    j = 0;
    for (size_t k = 0; k < pend.size(); ++k) {
      if (pend[k].type == 'c') continue;
      if (jobs[j].status) return jobs[j].status;
      if (!rc && jobs[j].out_len != pend[k].usize) return ZPQ_ERR_FORMAT;
      ++j;
    }
    if (rc) return rc;
    if (d_arc && (rc = zpq_d2h(ctx, plain.data(), dbuf.p, total))) return rc;
  }
  // ---- the tables, in archive order
  size_t data_offset = 0;
  for (size_t k = 0; k < pend.size(); ++k) {
    const Pending& P = pend[k];
    const uint8_t* os = plain.data() + off[k];
    const size_t osz = P.usize;
    if (P.type == 'c') {
      ++ix.versions;
      data_offset = P.data_offset;
    } else if (P.type == 'h') {
      if (osz % 24 != 4) return ZPQ_ERR_FORMAT;
      const uint32_t nf = (uint32_t)((osz - 4) / 24), bsize = get32(os);
      if (P.num < 1 || (uint64_t)P.num + nf > 0xffffffffull) return ZPQ_ERR_FORMAT;
      DBlock b; b.offset = data_offset; b.csize = bsize; b.first_frag = P.num; b.nfrag = nf; b.usize = 8;
      if (ix.ht.size() < (size_t)P.num + nf) ix.ht.resize((size_t)P.num + nf);
      for (uint32_t i = 0; i < nf; ++i) {
        memcpy(ix.ht[P.num + i].sha1.d, os + 4 + 24 * i, 20);
        ix.ht[P.num + i].usize = get32(os + 24 + 24 * i);
        b.usize += (uint64_t)ix.ht[P.num + i].usize + 4u;      // 64 bits: a hostile size must not wrap
      }
      ix.blocks.push_back(b);
      data_offset += bsize;
    } else {
      const uint8_t* s = os; const uint8_t* end = s + osz;
      while (s + 9 <= end) {
        FileRec fr; fr.date = (int64_t)get64(s); s += 8;
        const uint8_t* z = (const uint8_t*)memchr(s, 0, end - s);
        if (!z) return ZPQ_ERR_FORMAT;
        std::string fn((const char*)s, z - s); s = z + 1;
        if (fr.date) {
          if (s + 4 > end) return ZPQ_ERR_FORMAT;
          const uint32_t na = get32(s); s += 4;
          if (s + na > end) return ZPQ_ERR_FORMAT;
          fr.attr.assign((const char*)s, na); s += na;
          if (s + 4 > end) return ZPQ_ERR_FORMAT;
          const uint32_t ni = get32(s); s += 4;
          if ((size_t)(end - s) / 4 < ni) return ZPQ_ERR_FORMAT;
          fr.ptr.resize(ni);
          for (uint32_t i = 0; i < ni; ++i) { fr.ptr[i] = get32(s); s += 4; }
        }
        ix.files[fn] = fr;
      }
    }
  }
  return ZPQ_OK;
}

// ZPQJ_TIMING=1: the phases of an add on stderr (milliseconds of host wall clock; debugging aid)
struct PhaseClock {
  bool on; std::chrono::steady_clock::time_point t0, last; std::string line;
  PhaseClock() : on(getenv("ZPQJ_TIMING") != nullptr), t0(std::chrono::steady_clock::now()), last(t0) {}
  void mark(const char* what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    char b[96]; snprintf(b, sizeof b, " %s %.1f", what, std::chrono::duration<double, std::milli>(now - last).count());
    line += b; last = now;
  }
  ~PhaseClock() { if (on) fprintf(stderr, "[zpqj add %.1f ms]%s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), line.c_str()); }
};

// ---- add ----------------------------------------------------------------------------------------------------------
// One context per GPU.  Files (name order) are cut into contiguous ranges of about equal size, one per context;
// every context fragments and hashes its range; the tables are concatenated in range order (what an RCCL all-gather
// of the per-GPU tables yields in a multi-process run); first-occurrence dedup and the block
// packer run once over the global table, so the archive does not depend on the number of GPUs; a d block is
// compressed by the context that holds its first fragment, fragments that live on another GPU are fetched peer to
// peer.  With one context this is the whole of zpaqfranz `a` for the -m0/-m1 family.
struct Shard {
  zpq_ctx* ctx = nullptr;
  size_t f0 = 0, f1 = 0;                  // files [f0, f1) in name order
  std::vector<uint64_t> off;              // file offsets inside this shard's buffer
  void* d_data = nullptr;
  std::vector<void*> dev;                 // everything to free
  std::vector<uint64_t> foff; std::vector<uint32_t> flen, ffile; Bytes dig;
  std::vector<uint32_t> crc; std::vector<uint64_t> xxh;      // per file, when asked for
  size_t nf = 0;
  int rc = ZPQ_OK;
  ~Shard() { for (void* q : dev) if (q) zpq_dev_free_pooled(ctx, q); }
};

// ext_base / ext_off: the files are already in HBM (zpqj_add_dev): file order[k] lies at ext_base + ext_off[k] .. ext_off[k + 1],
// back to back in name order, the caller's buffer -- nothing is copied and nothing of it is freed here.
int shard_fragment(Shard& S, const uint8_t* const* datas, const uint64_t* sizes, const std::vector<size_t>& order, bool checksums,
                   const uint8_t* ext_base = nullptr, const uint64_t* ext_off = nullptr, bool no_twins = false) {
  zpq_ctx* ctx = S.ctx;
  const size_t nfiles = S.f1 - S.f0;
  int rc;
  if (ext_base && ext_off) {
    S.off.assign(ext_off + S.f0, ext_off + S.f1 + 1);
    S.d_data = (void*)ext_base;
  } else if (ext_base) {          // (zpqj_add_sharded_dev: this rank's range lies back to back at ext_base, sizes come from the caller)
    S.off.assign(nfiles + 1, 0);
    for (size_t k = 0; k < nfiles; ++k) S.off[k + 1] = S.off[k] + sizes[order[S.f0 + k]];
    S.d_data = (void*)ext_base;
  } else {
    S.off.assign(nfiles + 1, 0);
    for (size_t k = 0; k < nfiles; ++k) S.off[k + 1] = S.off[k] + sizes[order[S.f0 + k]];
    const uint64_t total = S.off[nfiles];
    if ((rc = zpq_dev_alloc_pooled(ctx, total + 64, &S.d_data))) return rc;
    S.dev.push_back(S.d_data);
    for (size_t k = 0; k < nfiles; ++k)
      if (sizes[order[S.f0 + k]] && (rc = zpq_h2d(ctx, (uint8_t*)S.d_data + S.off[k], datas[order[S.f0 + k]], sizes[order[S.f0 + k]]))) return rc;
    if ((rc = zpq_dev_memset(ctx, (uint8_t*)S.d_data + total, 0, 64))) return rc;
  }
  zpq_fragment_params fp;
  zpq_fragment_params_default(&fp);
  const size_t cap = std::max<size_t>(1, zpq_fragment_capacity(S.off.data(), nfiles, &fp));
  void *d_foff, *d_flen, *d_ffile, *d_dig;
  if ((rc = zpq_dev_alloc_pooled(ctx, cap * 8, &d_foff))) return rc; S.dev.push_back(d_foff);
  if ((rc = zpq_dev_alloc_pooled(ctx, cap * 4, &d_flen))) return rc; S.dev.push_back(d_flen);
  if ((rc = zpq_dev_alloc_pooled(ctx, cap * 4, &d_ffile))) return rc; S.dev.push_back(d_ffile);
  if ((rc = zpq_dev_alloc_pooled(ctx, cap * 20 + 64, &d_dig))) return rc; S.dev.push_back(d_dig);
  size_t nf = 0;
  // fragment loop + SHA-1 of every fragment; files whose bytes equal an earlier file of this range (compared on the device,
  // csrc/twins.hip) are not walked again: their records are the earlier file's, moved
  if (nfiles && (rc = zpq_fragment_sha1_dev(ctx, (const uint8_t*)S.d_data, S.off.data(), nfiles, &fp, (uint64_t*)d_foff, (uint32_t*)d_flen,
                                            (uint32_t*)d_ffile, (uint8_t*)d_dig, cap, &nf, no_twins ? ZPQ_FS_NO_TWINS : 0u, nullptr, nullptr))) return rc;
  if (checksums && nfiles) {       // zpaqfranz stores XXHASH64 + CRC-32 of every file in its i-block attribute: same pass over HBM
    S.crc.resize(nfiles); S.xxh.resize(nfiles);
    if ((rc = zpq_file_checksums_dev(ctx, (const uint8_t*)S.d_data, S.off.data(), nfiles, S.crc.data(), S.xxh.data(), nullptr))) return rc;
  }
  S.nf = nf;
  S.foff.resize(nf); S.flen.resize(nf); S.ffile.resize(nf); S.dig.resize(nf * 20);
  if (nf) {
    if ((rc = zpq_d2h(ctx, S.foff.data(), d_foff, nf * 8)) || (rc = zpq_d2h(ctx, S.flen.data(), d_flen, nf * 4)) ||
        (rc = zpq_d2h(ctx, S.ffile.data(), d_ffile, nf * 4)) || (rc = zpq_d2h(ctx, S.dig.data(), d_dig, nf * 20))) return rc;
  }
  return ZPQ_OK;
}

// Contiguous ranges (in name order) of about equal bytes: range r is order[edge[r] .. edge[r+1]).
void shard_plan(const std::vector<size_t>& order, const uint64_t* sizes, size_t nctx, std::vector<size_t>& edge) {
  const size_t nfiles = order.size();
  uint64_t total = 0;
  for (size_t i = 0; i < nfiles; ++i) total += sizes[i];
  edge.assign(nctx + 1, 0);
  size_t f = 0; uint64_t acc = 0;
  for (size_t r = 0; r < nctx; ++r) {
    edge[r] = f;
    const uint64_t goal = total / nctx * (r + 1) + (r + 1 == nctx ? total : 0);
    while (f < nfiles && (r + 1 == nctx || acc + sizes[order[f]] / 2 <= goal)) acc += sizes[order[f++]];
    edge[r + 1] = f;
  }
}

// Process-sharded runs: the ranges of the plan above live in `world` processes (one GPU each); what the in-process form
// reads from its neighbours' Shard objects travels through ONE caller-supplied primitive, an all-gather of byte strings
// (RCCL / MPI / anything): fragment tables, the few fragments a block needs from another rank, the compressed blocks.
struct Xchg { int rank, world; zpqj_allgatherv_fn fn; void* user; zpqj_allgatherv_dev_fn dev_fn; };
int xchg_all(const Xchg& X, const Bytes& send, std::vector<Bytes>& got) {
  std::vector<void*> rp(X.world, nullptr); std::vector<size_t> rl(X.world, 0);
  const int rc = X.fn(X.user, send.data(), send.size(), rp.data(), rl.data());
  if (rc) return ZPQ_ERR_ARG;
  got.assign(X.world, Bytes());
  for (int r = 0; r < X.world; ++r) if (rl[r]) got[r].assign((const uint8_t*)rp[r], (const uint8_t*)rp[r] + rl[r]);
  return ZPQ_OK;
}

int add_impl(zpq_ctx* const* ctxs, size_t nctx, const uint8_t* archive, size_t archive_len, const char* const* names, const uint8_t* const* datas,
             const uint64_t* sizes, const int64_t* dates, size_t nfiles, int64_t version_date, const char* method,
             uint8_t** out, size_t* out_len, uint64_t stats[6], uint32_t flags = 0, const Xchg* X = nullptr,
             const uint8_t* ext_base = nullptr, const uint64_t* ext_off = nullptr) {
  *out = nullptr; *out_len = 0;
  if (nctx == 0 || !ctxs || !ctxs[0]) return ZPQ_ERR_ARG;
  if (ext_base && (X ? (ext_off != nullptr || !sizes) : (nctx != 1 || !ext_off))) return ZPQ_ERR_ARG;
  const size_t me = X ? (size_t)X->rank : 0;
  if (X) { if (X->world < 1 || X->rank < 0 || X->rank >= X->world || !X->fn) return ZPQ_ERR_ARG; nctx = (size_t)X->world; }
  const bool checksums = (flags & ZPQJ_FILE_CHECKSUMS) != 0;
  bool hint = (flags & ZPQJ_METHOD_HINT) != 0;
  for (const char* q = method; hint && q && *q; ++q) if (*q < '0' || *q > '9') hint = false;     // only for "LB" methods
  zpq_ctx* ctx = ctxs[0];
  PhaseClock clk;
  Index ix;
  if (archive && archive_len) { int rc = read_index(ctx, archive, archive_len, ix); if (rc) return rc; }
  std::unordered_map<Sha1Key, uint32_t, Sha1Hash> known;
  for (size_t i = 1; i < ix.ht.size(); ++i) known.emplace(ix.ht[i].sha1, (uint32_t)i);
  // files in name order (the fixture's i blocks are name ordered)
  std::vector<size_t> order(nfiles);
  for (size_t i = 0; i < nfiles; ++i) order[i] = i;
  std::vector<uint64_t> ext_sizes;
  if (ext_base) {
    // device-resident files lie back to back in the order they are processed in: STRICTLY ascending names (checked name by
    // name -- a sort is not stable, equal names would pass or fail by accident: ADVICE round 5), offsets that ascend from 0,
    // a base the kernels' 16-byte reads can use (ZPQ_ERR_ARG otherwise)
    if (((uintptr_t)ext_base & 15) != 0) return ZPQ_ERR_ARG;
    for (size_t i = 0; i < nfiles; ++i) {
      if (!names[i] || (ext_off && ext_off[i + 1] < ext_off[i])) return ZPQ_ERR_ARG;
      if (i && strcmp(names[i - 1], names[i]) >= 0) return ZPQ_ERR_ARG;
    }
    if (ext_off) {
      ext_sizes.resize(nfiles);
      for (size_t i = 0; i < nfiles; ++i) ext_sizes[i] = ext_off[i + 1] - ext_off[i];
      sizes = ext_sizes.data();
    }
  } else {
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return strcmp(names[a], names[b]) < 0; });
  }
  // 1. contiguous file ranges of about equal bytes, one per context; fragment + hash per range (one host thread each)
  std::vector<Shard> sh(nctx);
  {
    std::vector<size_t> edge;
    shard_plan(order, sizes, nctx, edge);
    for (size_t r = 0; r < nctx; ++r) { sh[r].ctx = X ? (r == me ? ctxs[0] : nullptr) : ctxs[r]; sh[r].f0 = edge[r]; sh[r].f1 = edge[r + 1]; }
  }
  {
    std::vector<std::thread> th;
    const bool no_twins = (flags & ZPQJ_NO_TWINS) != 0;
    if (nctx == 1) sh[0].rc = shard_fragment(sh[0], datas, sizes, order, checksums, ext_base, ext_off, no_twins);      // (no thread for one range)
    else for (size_t r = 0; r < nctx; ++r) if (!X || r == me) th.emplace_back([&, r] { sh[r].rc = shard_fragment(sh[r], datas, sizes, order, checksums, X ? ext_base : nullptr, nullptr, no_twins); });
    for (auto& t : th) t.join();
    for (size_t r = 0; r < nctx; ++r) if (sh[r].rc) return sh[r].rc;
  }
  clk.mark("fragment+sha1");
  if (X) {
    // exchange 1: every rank's fragment table (lengths, owning files, SHA-1s) and file checksums
    Bytes snd; std::vector<Bytes> got;
    const Shard& L = sh[me];
    put64(snd, L.nf);
    snd.insert(snd.end(), (const uint8_t*)L.flen.data(), (const uint8_t*)L.flen.data() + L.nf * 4);
    snd.insert(snd.end(), (const uint8_t*)L.ffile.data(), (const uint8_t*)L.ffile.data() + L.nf * 4);
    snd.insert(snd.end(), L.dig.begin(), L.dig.begin() + L.nf * 20);
    put64(snd, L.crc.size());
    snd.insert(snd.end(), (const uint8_t*)L.crc.data(), (const uint8_t*)L.crc.data() + L.crc.size() * 4);
    snd.insert(snd.end(), (const uint8_t*)L.xxh.data(), (const uint8_t*)L.xxh.data() + L.xxh.size() * 8);
    int rcx = xchg_all(*X, snd, got);
    if (rcx) return rcx;
    for (size_t r = 0; r < nctx; ++r) {
      if (r == me) continue;
      const Bytes& g = got[r];
      if (g.size() < 16) return ZPQ_ERR_FORMAT;
      const uint64_t n = get64(g.data());
      if (n > (g.size() - 16) / 28) return ZPQ_ERR_FORMAT;                 // (no product that could wrap)
      Shard& S = sh[r];
      S.nf = (size_t)n; S.flen.resize(n); S.ffile.resize(n); S.dig.resize(n * 20);
      memcpy(S.flen.data(), g.data() + 8, n * 4); memcpy(S.ffile.data(), g.data() + 8 + n * 4, n * 4); memcpy(S.dig.data(), g.data() + 8 + n * 8, n * 20);
      const uint8_t* q = g.data() + 8 + n * 28;
      const uint64_t nc = get64(q); q += 8;
      // a peer sends one checksum pair per file of its range (none without ZPQJ_FILE_CHECKSUMS): the i blocks index sh[r].xxh / crc by file
      if (nc > (uint64_t)(g.data() + g.size() - q) / 12 || nc != (checksums ? (uint64_t)(S.f1 - S.f0) : 0ull)) return ZPQ_ERR_FORMAT;
      S.crc.resize(nc); S.xxh.resize(nc);
      memcpy(S.crc.data(), q, nc * 4); memcpy(S.xxh.data(), q + nc * 4, nc * 8);
    }
  }
  // 2. the global fragment table: ranges in order (the "all-gather"), global first-occurrence dedup on context 0
  std::vector<size_t> base(nctx + 1, 0);
  for (size_t r = 0; r < nctx; ++r) base[r + 1] = base[r] + sh[r].nf;
  const size_t nf = base[nctx];
  std::vector<uint32_t> flen(nf), ffile(nf), first(nf), shard_of(nf); Bytes dig(nf * 20);
  for (size_t r = 0; r < nctx; ++r)
    for (size_t i = 0; i < sh[r].nf; ++i) {
      flen[base[r] + i] = sh[r].flen[i]; ffile[base[r] + i] = (uint32_t)(sh[r].f0 + sh[r].ffile[i]); shard_of[base[r] + i] = (uint32_t)r;
    }
  for (size_t r = 0; r < nctx; ++r) if (sh[r].nf) memcpy(&dig[base[r] * 20], sh[r].dig.data(), sh[r].nf * 20);
  int rc;
  if (nf) {
    void *d_dig = nullptr, *d_first = nullptr;
    if ((rc = zpq_dev_alloc_pooled(ctx, nf * 20 + 64, &d_dig))) return rc;
    if ((rc = zpq_dev_alloc_pooled(ctx, nf * 4, &d_first))) { zpq_dev_free_pooled(ctx, d_dig); return rc; }
    rc = zpq_h2d(ctx, d_dig, dig.data(), nf * 20);
    if (!rc) rc = zpq_dedup_dev(ctx, (const uint8_t*)d_dig, nf, (uint32_t*)d_first);
    if (!rc) rc = zpq_d2h(ctx, first.data(), d_first, nf * 4);
    zpq_dev_free_pooled(ctx, d_dig); zpq_dev_free_pooled(ctx, d_first);
    if (rc) return rc;
  }
  clk.mark("tables+dedup");
  // fragment ids: known from earlier versions, else new (first occurrence in this batch)
  const uint32_t first_new_id = (uint32_t)ix.ht.size();
  std::vector<uint32_t> id(nf, 0), newfrags;
  for (size_t i = 0; i < nf; ++i) {
    if (first[i] != i) { id[i] = id[first[i]]; continue; }
    Sha1Key k; memcpy(k.d, &dig[20 * i], 20);
    auto it = known.find(k);
    if (it != known.end()) { id[i] = it->second; continue; }
    id[i] = first_new_id + (uint32_t)newfrags.size();
    newfrags.push_back((uint32_t)i);
  }
  // 3. pack new fragments into d blocks; a block belongs to the context holding its first fragment
  std::vector<std::pair<size_t, size_t>> blocks;   // [begin, end) into newfrags
  const uint32_t block_limit = block_limit_for(method);
  for (size_t b = 0; b < newfrags.size();) {
    size_t e = b; uint64_t bytes = 8;
    while (e < newfrags.size() && (e == b || bytes + flen[newfrags[e]] + 4 <= block_limit)) { bytes += flen[newfrags[e]] + 4; ++e; }
    blocks.push_back({b, e}); b = e;
  }
  std::vector<Bytes> dblock(blocks.size());
  std::vector<int> brc(nctx, ZPQ_OK);
  // Who compresses a d block, and where it reads the block's fragments (round 6).  Until now: the context that holds the block's
  // FIRST fragment, every fragment read where it first occurred.  With ONE tree of copies over several GPUs -- the shape of
  // the BASELINE metric -- every new fragment first occurs on the first GPU, which then compressed every block while the others
  // only fragmented and hashed.  Now the blocks are DEALT OUT in equal contiguous ranges (block b to context b n / B) wherever
  // the context dealt to holds the whole block, and a context reads a fragment from its OWN data whenever it holds a copy of it -- the global table says which of a context's fragments have that
  // first occurrence, every rank derives the same answer for every rank -- so only fragments a context really lacks travel
  // (exchange 2 / peer copies).  The archive's bytes do not depend on any of this.
  const size_t nnew = newfrags.size();
  std::vector<uint32_t> owner_of(blocks.size(), 0);
  for (size_t b = 0; b < blocks.size(); ++b) owner_of[b] = shard_of[newfrags[blocks[b].first]];
  // copy_at[r * nnew + k]: a fragment of context r with the content of new fragment k (its global index), or ~0
  std::vector<uint32_t> copy_at;
  if (nctx > 1) {
    std::vector<uint32_t> new_rank(nf, 0xffffffffu);
    for (size_t k = 0; k < nnew; ++k) new_rank[newfrags[k]] = (uint32_t)k;
    copy_at.assign(nctx * nnew, 0xffffffffu);
    for (size_t i = 0; i < nf; ++i) {
      const uint32_t k = new_rank[first[i]];
      if (k == 0xffffffffu) continue;                       // (known from an earlier version: not stored again)
      uint32_t& c = copy_at[(size_t)shard_of[i] * nnew + k];
      if (c == 0xffffffffu) c = (uint32_t)i;
    }
    // a block goes to the context it is dealt to only if that context holds EVERY fragment of it (then nothing of the block
    // travels); otherwise it stays with the context of its first fragment, as before (a tree of different files per GPU:
    // shipping whole blocks' worth of fragments would cost more than the compressor it spreads)
    for (size_t b = 0; b < blocks.size(); ++b) {
      const size_t dealt = b * nctx / blocks.size();
      bool all = true;
      for (size_t k = blocks[b].first; k < blocks[b].second && all; ++k) all = copy_at[dealt * nnew + k] != 0xffffffffu;
      if (all) owner_of[b] = (uint32_t)dealt;
    }
    if (getenv("ZPQJ_TRACE_OWNERS")) {        // debugging aid: who compresses which block
      std::string line;
      for (size_t b = 0; b < blocks.size(); ++b) line += " " + std::to_string(owner_of[b]) + (owner_of[b] == shard_of[newfrags[blocks[b].first]] ? "" : "*");
      fprintf(stderr, "[zpqj add] %zu contexts, d block owners (* = dealt away from its first fragment's context):%s\n", nctx, line.c_str());
    }
  }
  // the global index fragment k of a block is read from by context r: its own copy if it has one, else the first occurrence
  auto source_of = [&](size_t r, size_t k) -> uint32_t {
    if (nctx > 1) { const uint32_t c = copy_at[r * nnew + k]; if (c != 0xffffffffu) return c; }
    return newfrags[k];
  };
  clk.mark("ids+pack");
  // exchange 2 (process-sharded): the fragments a block takes from a rank other than its owner (the packing crosses a range
  // edge at most once per edge, so this is a few fragments per rank).  Every rank derives the same list from the global
  // table; a sender gathers its part on the device, and xoff[k] is where fragment newfrags[k] sits in its sender's string.
  std::vector<uint64_t> xoff;
  std::vector<Bytes> xgot;
  if (X) {
    xoff.assign(newfrags.size(), 0);
    std::vector<uint64_t> cur(nctx, 0), so, dso; std::vector<uint32_t> sl;
    for (size_t b = 0; b < blocks.size(); ++b) {
      const size_t owner = owner_of[b];
      for (size_t k = blocks[b].first; k < blocks[b].second; ++k) {
        const uint32_t f = source_of(owner, k); const size_t src = shard_of[f];
        if (src == owner) continue;
        xoff[k] = cur[src];
        if (src == me) { so.push_back(sh[me].foff[f - base[me]]); sl.push_back(flen[f]); dso.push_back(cur[src]); }
        cur[src] += flen[f];
      }
    }
    Bytes snd((size_t)cur[me]);
    if (!so.empty() && cur[me]) {
      zpq_ctx* c = ctxs[0];
      struct Dev { zpq_ctx* c; std::vector<void*> p; ~Dev() { for (void* q : p) zpq_dev_free_pooled(c, q); } } dev{c, {}};
      void *d_st, *d_so, *d_sl, *d_dso;
      if ((rc = zpq_dev_alloc_pooled(c, cur[me] + 64, &d_st))) return rc; dev.p.push_back(d_st);
      if ((rc = zpq_dev_alloc_pooled(c, so.size() * 8, &d_so))) return rc; dev.p.push_back(d_so);
      if ((rc = zpq_dev_alloc_pooled(c, so.size() * 4, &d_sl))) return rc; dev.p.push_back(d_sl);
      if ((rc = zpq_dev_alloc_pooled(c, so.size() * 8, &d_dso))) return rc; dev.p.push_back(d_dso);
      if ((rc = zpq_h2d(c, d_so, so.data(), so.size() * 8)) || (rc = zpq_h2d(c, d_sl, sl.data(), sl.size() * 4)) ||
          (rc = zpq_h2d(c, d_dso, dso.data(), dso.size() * 8))) return rc;
      if ((rc = zpq_gather_dev(c, (const uint8_t*)sh[me].d_data, (const uint64_t*)d_so, (const uint32_t*)d_sl, (const uint64_t*)d_dso, so.size(), (uint8_t*)d_st))) return rc;
      if ((rc = zpq_d2h(c, snd.data(), d_st, snd.size()))) return rc;
    }
    if ((rc = xchg_all(*X, snd, xgot))) return rc;
    for (size_t r = 0; r < nctx; ++r) if (r != me && xgot[r].size() != cur[r]) return ZPQ_ERR_FORMAT;
  }
  // One context and no exchange (zpqj_add, zpqj_add_dev): the compressed d blocks wait in HBM until the index blocks are
  // compressed too and the archive's layout is known; then they are packed back to back by one gather and come over in ONE
  // copy, straight to their place in the buffer that is returned (no per-block copies, no intermediate strings).
  const bool one = nctx == 1 && !X;
  // ... and with a collective that takes DEVICE memory (zpqj_add_sharded_dev) this rank's compressed blocks stay in HBM as well:
  // they are packed there and go HBM -> RCCL -> HBM; what the host sees of exchange 3 is the table of sizes
  const bool xdev = X && X->dev_fn;
  const bool keep_dev = one || xdev;
  struct Packed { zpq_ctx* c = nullptr; void* d_out = nullptr; std::vector<uint64_t> off; std::vector<uint32_t> len; std::vector<size_t> mine;
                  ~Packed() { if (d_out) zpq_dev_free_pooled(c, d_out); } } packed;
  packed.c = ctxs[0];
  auto compress_owned = [&](size_t r) -> int {
    zpq_ctx* c = X ? ctxs[0] : ctxs[r];
    std::vector<size_t> mine;
    for (size_t b = 0; b < blocks.size(); ++b) if (owner_of[b] == r) mine.push_back(b);
    if (mine.empty()) return ZPQ_OK;
    struct Dev { zpq_ctx* c; std::vector<void*> p; ~Dev() { for (void* q : p) zpq_dev_free_pooled(c, q); } } dev{c, {}};
    std::vector<uint64_t> so, dso; std::vector<uint32_t> sl; std::vector<uint64_t> boff(mine.size()); std::vector<uint32_t> bn(mine.size());
    struct Remote { size_t src_shard; uint64_t src_off; uint64_t dst_off; uint32_t len; };    // src_off: in the peer's buffer, or in its exchanged string
    std::vector<Remote> remote;
    uint64_t pos = 0;
    for (size_t m = 0; m < mine.size(); ++m) {
      const size_t b = mine[m];
      boff[m] = pos; uint64_t q = pos;
      for (size_t k = blocks[b].first; k < blocks[b].second; ++k) {
        const uint32_t f = source_of(r, k); const size_t s = shard_of[f];
        const uint64_t src = (X && s != r) ? xoff[k] : sh[s].foff[f - base[s]];
        if (s == r) { so.push_back(src); sl.push_back(flen[f]); dso.push_back(q); }
        else remote.push_back({s, src, q, flen[f]});
        q += flen[f];
      }
      const uint32_t cnt = (uint32_t)(blocks[b].second - blocks[b].first);
      bn[m] = (uint32_t)(q - pos) + 4 * cnt + 8;
      pos += (bn[m] + 127) & ~(uint64_t)63;
    }
    int rc;
    void *d_blk, *d_so = nullptr, *d_sl = nullptr, *d_dso = nullptr;
    if ((rc = zpq_dev_alloc_pooled(c, pos + 64, &d_blk))) return rc; dev.p.push_back(d_blk);
    if (!so.empty()) {
      if ((rc = zpq_dev_alloc_pooled(c, so.size() * 8, &d_so))) return rc; dev.p.push_back(d_so);
      if ((rc = zpq_dev_alloc_pooled(c, so.size() * 4, &d_sl))) return rc; dev.p.push_back(d_sl);
      if ((rc = zpq_dev_alloc_pooled(c, so.size() * 8, &d_dso))) return rc; dev.p.push_back(d_dso);
      if ((rc = zpq_h2d(c, d_so, so.data(), so.size() * 8)) || (rc = zpq_h2d(c, d_sl, sl.data(), sl.size() * 4)) ||
          (rc = zpq_h2d(c, d_dso, dso.data(), dso.size() * 8))) return rc;
      if ((rc = zpq_gather_dev(c, (const uint8_t*)sh[r].d_data, (const uint64_t*)d_so, (const uint32_t*)d_sl, (const uint64_t*)d_dso, so.size(), (uint8_t*)d_blk))) return rc;
    }
    for (const Remote& R : remote) {     // fragments of this block that another GPU holds: peer to peer, or from the exchange
      if (X) {
        if (R.src_off + R.len > xgot[R.src_shard].size()) return ZPQ_ERR_FORMAT;
        if (R.len && (rc = zpq_h2d(c, (uint8_t*)d_blk + R.dst_off, xgot[R.src_shard].data() + R.src_off, R.len))) return rc;
        continue;
      }
      if (R.len && (rc = zpq_copy_peer(c, (uint8_t*)d_blk + R.dst_off, sh[R.src_shard].ctx, (const uint8_t*)sh[R.src_shard].d_data + R.src_off, R.len))) return rc;
    }
    if (nctx == 1) clk.mark("gather");
    std::vector<zpq_block_job> jobs(mine.size());
    std::vector<std::string> nm(mine.size());
    uint64_t opos = 0; std::vector<uint64_t> ooff(mine.size());
    for (size_t m = 0; m < mine.size(); ++m) {
      const size_t b = mine[m];
      Bytes tr;
      for (size_t k = blocks[b].first; k < blocks[b].second; ++k) put32(tr, flen[newfrags[k]]);
      put32(tr, 0); put32(tr, (uint32_t)(blocks[b].second - blocks[b].first));
      if ((rc = zpq_h2d(c, (uint8_t*)d_blk + boff[m] + bn[m] - tr.size(), tr.data(), tr.size()))) return rc;
      nm[m] = block_name(version_date, 'd', first_new_id + (uint32_t)blocks[b].first);
      ooff[m] = opos; opos += (zpq_block_bound(bn[m], nm[m].c_str(), "jDC\x01") + 63) & ~(size_t)63;
    }
    void* d_out;
    if ((rc = zpq_dev_alloc_pooled(c, opos + 64, &d_out))) return rc;
    if (keep_dev) { packed.c = c; packed.d_out = d_out; } else dev.p.push_back(d_out);
    // "method,R,t" per block (zpaq's add(); ZSFX/libzpaq.h:86-135): R from the order-1 hits of its fragments, t from the
    // text / exe votes -- one lane per fragment over the assembled blocks
    std::vector<std::string> mth(mine.size(), std::string(method));
    if (hint) {
      std::vector<uint64_t> fo; std::vector<uint32_t> fl; std::vector<size_t> fb;
      for (size_t m = 0; m < mine.size(); ++m) {
        uint64_t q = boff[m];
        for (size_t k = blocks[mine[m]].first; k < blocks[mine[m]].second; ++k) { fo.push_back(q); fl.push_back(flen[newfrags[k]]); fb.push_back(m); q += flen[newfrags[k]]; }
      }
      void *d_fo, *d_fl, *d_st;
      if ((rc = zpq_dev_alloc_pooled(c, fo.size() * 8 + 64, &d_fo))) return rc; dev.p.push_back(d_fo);
      if ((rc = zpq_dev_alloc_pooled(c, fo.size() * 4 + 64, &d_fl))) return rc; dev.p.push_back(d_fl);
      if ((rc = zpq_dev_alloc_pooled(c, fo.size() * 16 + 64, &d_st))) return rc; dev.p.push_back(d_st);
      if ((rc = zpq_h2d(c, d_fo, fo.data(), fo.size() * 8)) || (rc = zpq_h2d(c, d_fl, fl.data(), fl.size() * 4))) return rc;
      if ((rc = zpq_fragment_stats_dev(c, (const uint8_t*)d_blk, (const uint64_t*)d_fo, (const uint32_t*)d_fl, fo.size(), (uint32_t*)d_st))) return rc;
      std::vector<uint32_t> st(fo.size() * 4);
      if ((rc = zpq_d2h(c, st.data(), d_st, st.size() * 4))) return rc;
      std::vector<uint64_t> hits(mine.size(), 0), bytes(mine.size(), 0); std::vector<uint32_t> nf(mine.size(), 0), tx(mine.size(), 0), ex(mine.size(), 0);
      for (size_t i = 0; i < fo.size(); ++i) { const size_t m = fb[i]; hits[m] += st[4 * i]; tx[m] += st[4 * i + 1]; ex[m] += st[4 * i + 2]; bytes[m] += st[4 * i + 3]; ++nf[m]; }
      for (size_t m = 0; m < mine.size(); ++m) {
        uint64_t R = hits[m] / (bytes[m] / 256 + 1);             // zpaq's add(): redundancy / (size / 256 + 1)
        if (R > 255) R = 255;
        const unsigned t = (ex[m] * 8 > nf[m] ? 2u : 0u) + (tx[m] * 4 > nf[m] ? 1u : 0u);
        mth[m] += "," + std::to_string(R) + "," + std::to_string(t);
      }
    }
    for (size_t m = 0; m < mine.size(); ++m) {
      zpq_block_job& j = jobs[m];
      memset(&j, 0, sizeof j);
      j.in = (uint8_t*)d_blk + boff[m]; j.n = bn[m]; j.method = mth[m].c_str(); j.filename = nm[m].c_str(); j.comment = "jDC\x01"; j.dosha1 = 1;
      j.out = (uint8_t*)d_out + ooff[m]; j.out_cap = (uint32_t)zpq_block_bound(bn[m], nm[m].c_str(), "jDC\x01");
    }
    if ((rc = zpq_sync(c))) return rc;
    if (nctx == 1) clk.mark("trailers");
    if ((rc = zpq_compress_blocks_dev(c, jobs.data(), jobs.size()))) return rc;
    if (nctx == 1) clk.mark("compressBlock");
    if (keep_dev) {          // (one context: mine = every block, in order)
      packed.off = ooff; packed.len.resize(mine.size()); packed.mine = mine;
      for (size_t m = 0; m < mine.size(); ++m) { if (jobs[m].status) return jobs[m].status; packed.len[m] = jobs[m].out_len; }
      return ZPQ_OK;
    }
    for (size_t m = 0; m < mine.size(); ++m) {
      dblock[mine[m]].resize(jobs[m].out_len);
      if ((rc = zpq_d2h(c, dblock[mine[m]].data(), jobs[m].out, jobs[m].out_len))) return rc;
    }
    return ZPQ_OK;
  };
  {
    std::vector<std::thread> th;
    if (nctx == 1) brc[0] = compress_owned(0);
    else for (size_t r = 0; r < nctx; ++r) if (!X || r == me) th.emplace_back([&, r] { brc[r] = compress_owned(r); });
    for (auto& t : th) t.join();
    for (size_t r = 0; r < nctx; ++r) if (brc[r]) return brc[r];
  }
  std::vector<uint32_t> xsize;                      // (device exchange) compressed size of every d block, whoever owns it
  std::vector<void*> xrecv; std::vector<size_t> xrlen;
  struct DevFree { zpq_ctx* c; void* p; ~DevFree() { if (p) zpq_dev_free_pooled(c, p); } } xpack{ctxs[0], nullptr};
  if (xdev) {
    // exchange 3, device form.  3a (host, a few bytes): which blocks this rank compressed and how long they came out;
    // 3b (device): the blocks themselves, packed back to back in HBM, HBM -> collective -> HBM; the archive is assembled from the
    // receive buffers by one copy per run of blocks of the same owner
    Bytes snd; std::vector<Bytes> got;
    uint64_t mylen = 0;
    for (size_t m = 0; m < packed.mine.size(); ++m) { put32(snd, (uint32_t)packed.mine[m]); put32(snd, packed.len[m]); mylen += packed.len[m]; }
    if ((rc = xchg_all(*X, snd, got))) return rc;
    xsize.assign(blocks.size(), 0);
    std::vector<uint64_t> rtotal(nctx, 0);
    for (size_t r = 0; r < nctx; ++r) {
      const Bytes& g = r == me ? snd : got[r];
      if (g.size() % 8) return ZPQ_ERR_FORMAT;
      uint32_t prev = 0;
      for (size_t q = 0; q < g.size(); q += 8) {
        const uint32_t b = get32(&g[q]), n = get32(&g[q + 4]);
        if (b >= blocks.size() || owner_of[b] != r || xsize[b] || !n || (q && b <= prev)) return ZPQ_ERR_FORMAT;
        xsize[b] = n; rtotal[r] += n; prev = b;
      }
    }
    for (size_t b = 0; b < blocks.size(); ++b) if (!xsize[b]) return ZPQ_ERR_FORMAT;
    zpq_ctx* c = ctxs[0];
    if (mylen) {
      struct Dev { zpq_ctx* c; std::vector<void*> p; ~Dev() { for (void* q : p) zpq_dev_free_pooled(c, q); } } dev{c, {}};
      const size_t nm_ = packed.mine.size();
      std::vector<uint64_t> dso(nm_); uint64_t q = 0;
      for (size_t m = 0; m < nm_; ++m) { dso[m] = q; q += packed.len[m]; }
      void *d_so, *d_sl, *d_dso;
      if ((rc = zpq_dev_alloc_pooled(c, mylen + 64, &xpack.p))) return rc;
      if ((rc = zpq_dev_alloc_pooled(c, nm_ * 8, &d_so))) return rc; dev.p.push_back(d_so);
      if ((rc = zpq_dev_alloc_pooled(c, nm_ * 4, &d_sl))) return rc; dev.p.push_back(d_sl);
      if ((rc = zpq_dev_alloc_pooled(c, nm_ * 8, &d_dso))) return rc; dev.p.push_back(d_dso);
      if ((rc = zpq_h2d(c, d_so, packed.off.data(), nm_ * 8)) || (rc = zpq_h2d(c, d_sl, packed.len.data(), nm_ * 4)) ||
          (rc = zpq_h2d(c, d_dso, dso.data(), nm_ * 8))) return rc;
      if ((rc = zpq_gather_dev(c, (const uint8_t*)packed.d_out, (const uint64_t*)d_so, (const uint32_t*)d_sl, (const uint64_t*)d_dso, nm_, (uint8_t*)xpack.p))) return rc;
      if ((rc = zpq_sync(c))) return rc;
    }
    xrecv.assign(nctx, nullptr); xrlen.assign(nctx, 0);
    if (X->dev_fn(X->user, xpack.p, (size_t)mylen, xrecv.data(), xrlen.data())) return ZPQ_ERR_ARG;
    for (size_t r = 0; r < nctx; ++r) if (xrlen[r] != rtotal[r] || (rtotal[r] && !xrecv[r])) return ZPQ_ERR_FORMAT;
  } else if (X) {
    // exchange 3: the compressed d blocks; afterwards every rank assembles the same archive
    Bytes snd; std::vector<Bytes> got;
    for (size_t b = 0; b < blocks.size(); ++b) if (owner_of[b] == me) {
      put32(snd, (uint32_t)b); put32(snd, (uint32_t)dblock[b].size()); snd.insert(snd.end(), dblock[b].begin(), dblock[b].end());
    }
    if ((rc = xchg_all(*X, snd, got))) return rc;
    for (size_t r = 0; r < nctx; ++r) {
      if (r == me) continue;
      const Bytes& g = got[r];
      for (size_t q = 0; q < g.size();) {
        if (g.size() - q < 8) return ZPQ_ERR_FORMAT;
        const uint32_t b = get32(&g[q]), n = get32(&g[q + 4]); q += 8;
        if (b >= blocks.size() || owner_of[b] != r || g.size() - q < n) return ZPQ_ERR_FORMAT;
        dblock[b].assign(g.begin() + q, g.begin() + q + n); q += n;
      }
    }
    for (size_t b = 0; b < blocks.size(); ++b) if (dblock[b].empty()) return ZPQ_ERR_FORMAT;
  }
  clk.mark("d blocks");
  Bytes dpart;                                      // the d blocks, in block order whoever compressed them (one context: they are still in HBM)
  std::vector<uint32_t> dsize(blocks.size());
  uint64_t dtotal = 0;
  for (size_t b = 0; b < blocks.size(); ++b) {
    dsize[b] = one ? packed.len[b] : xdev ? xsize[b] : (uint32_t)dblock[b].size();
    dtotal += dsize[b];
    if (!keep_dev) dpart.insert(dpart.end(), dblock[b].begin(), dblock[b].end());
  }
  // 4. c block, d blocks, h blocks, i blocks: the index blocks are put together first and compressed by ONE call
  std::vector<HostBlock> hb;
  {
    Bytes t8; put64(t8, dtotal);
    hb.push_back({t8, "0", block_name(version_date, 'c', first_new_id)});
  }
  for (size_t b = 0; b < blocks.size(); ++b) {
    Bytes tmp;
    tmp.reserve(4 + 24 * (blocks[b].second - blocks[b].first));
    put32(tmp, dsize[b]);
    for (size_t k = blocks[b].first; k < blocks[b].second; ++k) { const uint32_t f = newfrags[k]; tmp.insert(tmp.end(), &dig[20 * f], &dig[20 * f] + 20); put32(tmp, flen[f]); }
    hb.push_back({std::move(tmp), "0", block_name(version_date, 'h', first_new_id + (uint32_t)blocks[b].first)});
  }
  Bytes tmp;
  uint32_t inum = 1;
  size_t fi = 0;
  for (size_t k = 0; k < nfiles; ++k) {
    put64(tmp, (uint64_t)dates[order[k]]);
    const char* nmz = names[order[k]];
    tmp.insert(tmp.end(), nmz, nmz + strlen(nmz) + 1);
    if (checksums) {
      // the attribute as zpaqfranz writes it in the fixture's i blocks: 58 bytes, attribute tag and value
      // in the first 8, XXHASH64 as 16 hex digits at [16,32), CRC-32 as 8 hex digits at [49,57)
      size_t r = 0; while (r + 1 < nctx && k >= sh[r].f1) ++r;
      char a[58]; memset(a, 0, sizeof a);
      a[0] = 'u'; a[1] = (char)0xa4; a[2] = (char)0x81;                              // unix mode 0100644
      snprintf(a + 16, 17, "%016llX", (unsigned long long)sh[r].xxh[k - sh[r].f0]);
      char c8[9]; snprintf(c8, sizeof c8, "%08X", (unsigned)sh[r].crc[k - sh[r].f0]);
      memcpy(a + 49, c8, 8);
      a[32] = 0;
      put32(tmp, 58); tmp.insert(tmp.end(), a, a + 58);
    } else {
      put32(tmp, 3); tmp.push_back('u'); tmp.push_back(0xa4); tmp.push_back(0x81);   // unix mode 0100644
    }
    size_t f1 = fi;
    while (f1 < nf && ffile[f1] == k) ++f1;
    put32(tmp, (uint32_t)(f1 - fi));
    {                                                  // ni pointers, 4 bytes each, low byte first
      const size_t at = tmp.size();
      tmp.resize(at + 4 * (f1 - fi));
      uint8_t* q = tmp.data() + at;
      for (; fi < f1; ++fi, q += 4) { const uint32_t v = id[fi]; q[0] = (uint8_t)v; q[1] = (uint8_t)(v >> 8); q[2] = (uint8_t)(v >> 16); q[3] = (uint8_t)(v >> 24); }
    }
    if (tmp.size() > 16000 || k + 1 == nfiles) {       // zpaq flushes the index every ~16 KB
      hb.push_back({std::move(tmp), "1", block_name(version_date, 'i', inum++)});
      tmp = Bytes();
    }
  }
  clk.mark("index built");
  std::vector<Bytes> hout;
  if ((rc = compress_host_many(ctx, hb, hout))) return rc;
  clk.mark("index compressed");
  size_t need = (size_t)dtotal;
  for (const Bytes& o : hout) need += o.size();
  uint8_t* ob = (uint8_t*)malloc(need ? need : 1);
  if (!ob) return ZPQ_ERR_NOMEM;
  struct Guard { uint8_t* p; ~Guard() { free(p); } } guard{ob};          // (released to the caller at the end)
  size_t at = 0;
  memcpy(ob + at, hout[0].data(), hout[0].size()); at += hout[0].size();
  if (one && dtotal) {
    // pack on the device (the blocks sit at their capacity bounds), then one copy to where the d blocks belong
    zpq_ctx* c = packed.c;
    struct Dev { zpq_ctx* c; std::vector<void*> p; ~Dev() { for (void* q : p) zpq_dev_free_pooled(c, q); } } dev{c, {}};
    std::vector<uint64_t> dso(blocks.size());
    uint64_t q = 0;
    for (size_t b = 0; b < blocks.size(); ++b) { dso[b] = q; q += dsize[b]; }
    void *d_pack, *d_so, *d_sl, *d_dso;
    if ((rc = zpq_dev_alloc_pooled(c, dtotal + 64, &d_pack))) return rc; dev.p.push_back(d_pack);
    if ((rc = zpq_dev_alloc_pooled(c, blocks.size() * 8, &d_so))) return rc; dev.p.push_back(d_so);
    if ((rc = zpq_dev_alloc_pooled(c, blocks.size() * 4, &d_sl))) return rc; dev.p.push_back(d_sl);
    if ((rc = zpq_dev_alloc_pooled(c, blocks.size() * 8, &d_dso))) return rc; dev.p.push_back(d_dso);
    if ((rc = zpq_h2d(c, d_so, packed.off.data(), blocks.size() * 8)) || (rc = zpq_h2d(c, d_sl, dsize.data(), blocks.size() * 4)) ||
        (rc = zpq_h2d(c, d_dso, dso.data(), blocks.size() * 8))) return rc;
    if ((rc = zpq_gather_dev(c, (const uint8_t*)packed.d_out, (const uint64_t*)d_so, (const uint32_t*)d_sl, (const uint64_t*)d_dso, blocks.size(), (uint8_t*)d_pack))) return rc;
    if ((rc = zpq_d2h(c, ob + at, d_pack, (size_t)dtotal))) return rc;
  } else if (xdev && dtotal) {
    // the d blocks, in block order, out of the ranks' receive buffers: one copy per run of blocks with the same owner
    std::vector<uint64_t> cur(nctx, 0);
    size_t w = at;
    for (size_t b = 0; b < blocks.size();) {
      const size_t r = owner_of[b];
      uint64_t run = 0; size_t e = b;
      while (e < blocks.size() && owner_of[e] == r) run += dsize[e++];
      if ((rc = zpq_d2h(ctxs[0], ob + w, (const uint8_t*)xrecv[r] + cur[r], (size_t)run))) return rc;
      cur[r] += run; w += (size_t)run; b = e;
    }
  } else if (dtotal) memcpy(ob + at, dpart.data(), dpart.size());
  at += (size_t)dtotal;
  for (size_t k = 1; k < hout.size(); ++k) { memcpy(ob + at, hout[k].data(), hout[k].size()); at += hout[k].size(); }
  clk.mark("archive");
  guard.p = nullptr;
  *out = ob;
  *out_len = need;
  if (stats) {
    uint64_t ub = 0; for (uint32_t f : newfrags) ub += flen[f];
    stats[0] = nf; stats[1] = newfrags.size(); stats[2] = blocks.size(); stats[3] = ub; stats[4] = dtotal; stats[5] = need;
  }
  return ZPQ_OK;
}

// ---- extract --------------------------------------------------------------------------------------------------------
// Jidac::extract (ZSFX/zsfx.cpp:2018-2281 + decompressThread :1731-1994) with the archive's d blocks staged in HBM:
// one zpq_decompress_blocks_dev over all of them, every fragment's SHA-1 compared with the h table on the device,
// the files assembled by one gather, one copy back.
// zpaqfranz's i-block attribute extension as the fixture archives carry it (8 bytes of zpaq attribute,
// then 50 bytes holding the file's XXHASH64 as 16 hex digits at [16,32) and its CRC-32 as 8 hex digits at [49,57)).
// Returns false when the attribute has another shape (then there is nothing stored to compare with).
bool stored_checksums(const std::string& attr, uint64_t* xxh, uint32_t* crc) {
  if (attr.size() != 58) return false;
  auto hex = [&](size_t at, size_t n, uint64_t* v) {
    uint64_t r = 0;
    for (size_t i = 0; i < n; ++i) {
      const unsigned char c = (unsigned char)attr[at + i];
      int d = c >= '0' && c <= '9' ? c - '0' : c >= 'A' && c <= 'F' ? c - 'A' + 10 : c >= 'a' && c <= 'f' ? c - 'a' + 10 : -1;
      if (d < 0) return false;
      r = r << 4 | (uint64_t)d;
    }
    *v = r; return true;
  };
  uint64_t c = 0;
  if (!hex(16, 16, xxh) || !hex(49, 8, &c)) return false;
  *crc = (uint32_t)c;
  return true;
}

// verify != nullptr: nothing is returned to the host; the files are assembled in HBM and their stored XXHASH64 / CRC-32
// (when the attributes carry them) recomputed there.  verify[0..6] = files, fragments checked, bytes restored, files
// with stored checksums, XXHASH64 mismatches, CRC-32 mismatches, d blocks.
// dv != nullptr (zpqj_extract_dev): `archive` is a DEVICE pointer, the files stay in HBM (dv->d_out) and, if asked for, their
// SHA-256 is computed there; what comes back to the host is the index: names and file offsets.
struct DevExtract { uint8_t* d_out; size_t out_cap; uint8_t* d_sha256; size_t sha_cap; uint32_t flags; uint64_t** file_off; uint64_t* stats; };
int extract_impl(zpq_ctx* ctx, const uint8_t* archive, size_t archive_len, uint8_t** data, uint64_t** sizes, char** names,
                 size_t* nfiles, uint64_t* verify = nullptr, const DevExtract* dv = nullptr) {
  Index ix;
  int rc;
  struct Shadow { uint8_t* p; ~Shadow() { free(p); } } shadow{nullptr};
  const uint8_t* d_archive = nullptr;
  if (dv) {
    // the host reads the index blocks only: a zero-filled shadow of the archive (calloc: pages that are never written are never
    // backed), filled where read_index asks -- the c block in front, then everything behind the first version's d blocks
    d_archive = archive;
    shadow.p = (uint8_t*)calloc(archive_len + 64, 1);
    if (!shadow.p) return ZPQ_ERR_NOMEM;
    const std::function<int(size_t, size_t)> fetch = [&](size_t lo, size_t hi) { return hi > lo ? zpq_d2h(ctx, shadow.p + lo, d_archive + lo, hi - lo) : (int)ZPQ_OK; };
    rc = read_index(ctx, shadow.p, archive_len, ix, &fetch, d_archive);
    archive = shadow.p;
  } else {
    rc = read_index(ctx, archive, archive_len, ix);
  }
  if (rc) return rc;
  struct Dev { zpq_ctx* c; std::vector<void*> p; ~Dev() { for (void* q : p) zpq_dev_free_pooled(c, q); } } dev{ctx, {}};
  const size_t nb = ix.blocks.size();
  // d blocks -> HBM
  std::vector<uint64_t> aoff(nb), poff(nb);
  uint64_t apos = 0, ppos = 0;
  for (size_t b = 0; b < nb; ++b) {
    const DBlock& B = ix.blocks[b];
    if (B.offset + B.csize > archive_len || B.usize > 0xffffffffull - 4096) return ZPQ_ERR_FORMAT;
    aoff[b] = apos; apos += ((uint64_t)B.csize + 64 + 63) & ~(uint64_t)63;
    poff[b] = ppos; ppos += (B.usize + 64 + 63) & ~(uint64_t)63;
  }
  // (plan only -- zpqj_extract_dev without an output buffer: the index is all that is wanted)
  const bool plan_only = dv && !dv->d_out;
  void *d_arc = nullptr, *d_plain = nullptr;
  if (!dv) {
    if ((rc = zpq_dev_alloc_pooled(ctx, apos + 64, &d_arc))) return rc; dev.p.push_back(d_arc);
    if ((rc = zpq_dev_memset(ctx, d_arc, 0, apos + 64))) return rc;
  }
  if (!plan_only) { if ((rc = zpq_dev_alloc_pooled(ctx, ppos + 64, &d_plain))) return rc; dev.p.push_back(d_plain); }
  std::vector<zpq_unblock_job> jobs(nb);
  for (size_t b = 0; b < nb && !plan_only; ++b) {
    const DBlock& B = ix.blocks[b];
    memset(&jobs[b], 0, sizeof jobs[b]);
    if (dv) {
      // the d blocks are decoded where they lie (the caller guarantees 64 readable bytes behind the archive)
      jobs[b].in = d_archive + B.offset;
    } else {
      if ((rc = zpq_h2d(ctx, (uint8_t*)d_arc + aoff[b], archive + B.offset, B.csize))) return rc;
      jobs[b].in = (uint8_t*)d_arc + aoff[b];
    }
    jobs[b].n = B.csize;
    jobs[b].out = (uint8_t*)d_plain + poff[b]; jobs[b].out_cap = (uint32_t)B.usize + 64;
  }
  if (nb && !plan_only) {
    rc = zpq_decompress_blocks_dev(ctx, jobs.data(), nb, 1);
    for (size_t b = 0; b < nb; ++b) {
      if (jobs[b].status) return jobs[b].status;
      if (jobs[b].out_len != ix.blocks[b].usize) return ZPQ_ERR_FORMAT;
    }
    if (rc) return rc;
  }
  // fragment id -> offset in d_plain; SHA-1 of every stored fragment against the h table (ZSFX/zsfx.cpp:1811-1834)
  std::vector<uint64_t> where(ix.ht.size(), ~(uint64_t)0);
  std::vector<uint64_t> voff; std::vector<uint32_t> vlen; Bytes want;
  for (size_t b = 0; b < nb; ++b) {
    uint64_t o = poff[b];
    for (uint32_t i = 0; i < ix.blocks[b].nfrag; ++i) {
      const uint32_t f = ix.blocks[b].first_frag + i;
      where[f] = o;
      // the h table's sizes must stay inside what the d block decoded to (less its own size table)
      if (o + ix.ht[f].usize > poff[b] + ix.blocks[b].usize - 8 - 4ull * ix.blocks[b].nfrag) return ZPQ_ERR_FORMAT;
      voff.push_back(o); vlen.push_back(ix.ht[f].usize); want.insert(want.end(), ix.ht[f].sha1.d, ix.ht[f].sha1.d + 20);
      o += ix.ht[f].usize;
    }
  }
  if (!voff.empty() && !plan_only) {
    void *d_voff, *d_vlen, *d_want, *d_got;
    const size_t nv = voff.size();
    if ((rc = zpq_dev_alloc_pooled(ctx, nv * 8, &d_voff))) return rc; dev.p.push_back(d_voff);
    if ((rc = zpq_dev_alloc_pooled(ctx, nv * 4, &d_vlen))) return rc; dev.p.push_back(d_vlen);
    if ((rc = zpq_dev_alloc_pooled(ctx, nv * 20 + 64, &d_want))) return rc; dev.p.push_back(d_want);
    if ((rc = zpq_dev_alloc_pooled(ctx, nv * 20 + 64, &d_got))) return rc; dev.p.push_back(d_got);
    if ((rc = zpq_h2d(ctx, d_voff, voff.data(), nv * 8)) || (rc = zpq_h2d(ctx, d_vlen, vlen.data(), nv * 4)) || (rc = zpq_h2d(ctx, d_want, want.data(), nv * 20))) return rc;
    if ((rc = zpq_sha1_extents_dev(ctx, (const uint8_t*)d_plain, (const uint64_t*)d_voff, (const uint32_t*)d_vlen, nv, (uint8_t*)d_got))) return rc;
    uint64_t mism = 0, firstbad = 0;
    if ((rc = zpq_digest_compare_dev(ctx, (const uint8_t*)d_got, (const uint8_t*)d_want, nv, 20, &mism, &firstbad))) return rc;
    if (mism) return ZPQ_ERR_CHECKSUM;
  }
  // files: every pointer becomes one copy extent into the blob
  std::vector<uint64_t> so, dso; std::vector<uint32_t> sl;
  std::vector<uint64_t> sz; std::string nm;
  std::vector<const FileRec*> recs;
  uint64_t blob_len = 0;
  for (auto& kv : ix.files) {
    if (!kv.second.date) continue;
    recs.push_back(&kv.second);
    uint64_t len = 0;
    for (uint32_t q : kv.second.ptr) {
      if (q == 0 || q >= ix.ht.size() || where[q] == ~(uint64_t)0) return ZPQ_ERR_FORMAT;
      so.push_back(where[q]); sl.push_back(ix.ht[q].usize); dso.push_back(blob_len + len);
      len += ix.ht[q].usize;
    }
    blob_len += len;
    sz.push_back(len); nm += kv.first; nm.push_back('\0');
  }
  if (dv) {
    // the index goes back: names and offsets; the files stay where the gather below puts them
    uint64_t* st = dv->stats;
    if (st) { st[0] = sz.size(); st[1] = voff.size(); st[2] = blob_len; st[3] = st[4] = st[5] = 0; st[6] = nb; }
    *dv->file_off = (uint64_t*)malloc((sz.size() + 1) * 8);
    *names = (char*)malloc(nm.size() ? nm.size() : 1);
    if (!*dv->file_off || !*names) return ZPQ_ERR_NOMEM;
    (*dv->file_off)[0] = 0;
    for (size_t i = 0; i < sz.size(); ++i) (*dv->file_off)[i + 1] = (*dv->file_off)[i] + sz[i];
    memcpy(*names, nm.data(), nm.size());
    *nfiles = sz.size();
    if (plan_only) return ZPQ_OK;
    if (blob_len + 64 > dv->out_cap || (dv->d_sha256 && sz.size() > dv->sha_cap)) return ZPQ_ERR_ARG;
  } else if (verify) {
    verify[0] = sz.size(); verify[1] = voff.size(); verify[2] = blob_len; verify[3] = verify[4] = verify[5] = 0; verify[6] = nb;
  } else {
    *data = (uint8_t*)malloc(blob_len ? blob_len : 1);
    *sizes = (uint64_t*)malloc((sz.size() ? sz.size() : 1) * 8);
    *names = (char*)malloc(nm.size() ? nm.size() : 1);
    if (!*data || !*sizes || !*names) return ZPQ_ERR_NOMEM;
  }
  void* d_blob = nullptr;
  if (dv) d_blob = dv->d_out;
  else { if ((rc = zpq_dev_alloc_pooled(ctx, blob_len + 64, &d_blob))) return rc; dev.p.push_back(d_blob); }
  if (!so.empty()) {
    void *d_so, *d_sl, *d_dso;
    if ((rc = zpq_dev_alloc_pooled(ctx, so.size() * 8, &d_so))) return rc; dev.p.push_back(d_so);
    if ((rc = zpq_dev_alloc_pooled(ctx, so.size() * 4, &d_sl))) return rc; dev.p.push_back(d_sl);
    if ((rc = zpq_dev_alloc_pooled(ctx, so.size() * 8, &d_dso))) return rc; dev.p.push_back(d_dso);
    if ((rc = zpq_h2d(ctx, d_so, so.data(), so.size() * 8)) || (rc = zpq_h2d(ctx, d_sl, sl.data(), sl.size() * 4)) ||
        (rc = zpq_h2d(ctx, d_dso, dso.data(), dso.size() * 8))) return rc;
    if ((rc = zpq_gather_dev(ctx, (const uint8_t*)d_plain, (const uint64_t*)d_so, (const uint32_t*)d_sl, (const uint64_t*)d_dso, so.size(), (uint8_t*)d_blob))) return rc;
    if (!verify && !dv && blob_len && (rc = zpq_d2h(ctx, *data, d_blob, blob_len))) return rc;
  }
  if (dv) {
    if (dv->d_sha256 && !sz.empty()) {
      // every restored file's SHA-256 (what `x` / `t` verify against the originals' -- zpaqfranz's -sha256 file hash), in HBM
      if (dv->flags & ZPQJ_X_TWINS) {
        uint64_t ts[4] = {0, 0, 0, 0};       // twins, twin bytes, files compared, bytes compared
        if ((rc = zpq_sha256_files_dev(ctx, (const uint8_t*)d_blob, *dv->file_off, sz.size(), dv->d_sha256, 0, ts))) return rc;
        if (dv->stats) { dv->stats[3] = ts[0]; dv->stats[4] = ts[1]; dv->stats[5] = ts[3]; }
      } else {
        void *d_fo, *d_fl;
        if ((rc = zpq_dev_alloc_pooled(ctx, sz.size() * 8, &d_fo))) return rc; dev.p.push_back(d_fo);
        if ((rc = zpq_dev_alloc_pooled(ctx, sz.size() * 8, &d_fl))) return rc; dev.p.push_back(d_fl);
        if ((rc = zpq_h2d(ctx, d_fo, *dv->file_off, sz.size() * 8)) || (rc = zpq_h2d(ctx, d_fl, sz.data(), sz.size() * 8))) return rc;
        if ((rc = zpq_sha256_extents_dev(ctx, (const uint8_t*)d_blob, (const uint64_t*)d_fo, (const uint64_t*)d_fl, sz.size(), dv->d_sha256))) return rc;
      }
    }
    return zpq_sync(ctx);
  }
  if (verify && !sz.empty()) {
    // the `t` command: per-file XXHASH64 + CRC-32 of the assembled bytes against what the i blocks store
    std::vector<uint64_t> foff(sz.size() + 1, 0);
    for (size_t i = 0; i < sz.size(); ++i) foff[i + 1] = foff[i] + sz[i];
    std::vector<uint32_t> crc(sz.size()); std::vector<uint64_t> xx(sz.size());
    if ((rc = zpq_file_checksums_dev(ctx, (const uint8_t*)d_blob, foff.data(), sz.size(), crc.data(), xx.data(), nullptr))) return rc;
    for (size_t i = 0; i < sz.size(); ++i) {
      uint64_t wx = 0; uint32_t wc = 0;
      if (!stored_checksums(recs[i]->attr, &wx, &wc)) continue;
      ++verify[3];
      if (wx != xx[i]) ++verify[4];
      if (wc != crc[i]) ++verify[5];
    }
    if (verify[4] || verify[5]) return ZPQ_ERR_CHECKSUM;
  }
  if (verify) return ZPQ_OK;
  memcpy(*sizes, sz.data(), sz.size() * 8); memcpy(*names, nm.data(), nm.size());
  *nfiles = sz.size();
  return ZPQ_OK;
}

// nothing thrown inside (bad_alloc from a hostile size, length_error ...) may cross the C boundary
template <class F>
int guarded(F f) {
  try { return f(); }
  catch (const std::bad_alloc&) { return ZPQ_ERR_NOMEM; }
  catch (const std::exception&) { return ZPQ_ERR_FORMAT; }
}

}  // namespace

extern "C" {

void zpqj_free(void* p) { free(p); }

// Adds one version holding `nfiles` files to `archive` (may be NULL/0 for a new archive) and returns
// the NEW bytes to append (malloc'd; release with zpqj_free).  Fragments already stored by earlier
// versions, and repeats inside this batch, become pointers (dedup).  stats[0..5] = fragments, new
// fragments, d blocks, unique bytes, d-block bytes written, total bytes written.
int zpqj_add(zpq_ctx* ctx, const uint8_t* archive, size_t archive_len, const char* const* names, const uint8_t* const* datas,
             const uint64_t* sizes, const int64_t* dates, size_t nfiles, int64_t version_date, const char* method,
             uint8_t** out, size_t* out_len, uint64_t stats[6]) {
  return guarded([&] { return add_impl(&ctx, 1, archive, archive_len, names, datas, sizes, dates, nfiles, version_date, method, out, out_len, stats); });
}

// zpqj_add_multi with options: ZPQJ_FILE_CHECKSUMS stores every file's XXHASH64 + CRC-32 (computed on the device in
// the pass that fragments the files) in its i-block attribute, as zpaqfranz does by default; zpqj_verify checks them.
int zpqj_add_opts(zpq_ctx* const* ctxs, size_t nctx, const uint8_t* archive, size_t archive_len, const char* const* names,
                  const uint8_t* const* datas, const uint64_t* sizes, const int64_t* dates, size_t nfiles, int64_t version_date,
                  const char* method, uint32_t flags, uint8_t** out, size_t* out_len, uint64_t stats[6]) {
  return guarded([&] { return add_impl(ctxs, nctx, archive, archive_len, names, datas, sizes, dates, nfiles, version_date, method, out, out_len, stats, flags); });
}

// The same over several GPUs of one node (one context each): the files are sharded across them, the archive
// bytes are identical to the single-GPU result whatever nctx is.
int zpqj_add_multi(zpq_ctx* const* ctxs, size_t nctx, const uint8_t* archive, size_t archive_len, const char* const* names,
                   const uint8_t* const* datas, const uint64_t* sizes, const int64_t* dates, size_t nfiles, int64_t version_date,
                   const char* method, uint8_t** out, size_t* out_len, uint64_t stats[6]) {
  return guarded([&] { return add_impl(ctxs, nctx, archive, archive_len, names, datas, sizes, dates, nfiles, version_date, method, out, out_len, stats); });
}

// The same across PROCESSES (one GPU each; ranks of an MPI / torch.distributed job, or a C++ Jidac forked per GPU): every
// rank passes the same names / sizes / dates / archive, but data only for the files zpqj_shard_files marks as its own;
// `allgatherv` is the one collective needed (called three times, by every rank, in the same order).  Every rank returns
// the same bytes -- the bytes zpqj_add returns for the whole batch on one GPU.
int zpqj_add_sharded(zpq_ctx* ctx, int rank, int world, zpqj_allgatherv_fn allgatherv, void* user, const uint8_t* archive, size_t archive_len,
                     const char* const* names, const uint8_t* const* datas, const uint64_t* sizes, const int64_t* dates, size_t nfiles,
                     int64_t version_date, const char* method, uint32_t flags, uint8_t** out, size_t* out_len, uint64_t stats[6]) {
  const Xchg X{rank, world, allgatherv, user, nullptr};
  return guarded([&] { return add_impl(&ctx, 1, archive, archive_len, names, datas, sizes, dates, nfiles, version_date, method, out, out_len, stats, flags, &X); });
}

// zpqj_add_sharded with this rank's files ALREADY IN HBM: `names` (all files of the batch, every rank the same list) ascend
// strictly, `sizes` holds every file's size, and the files zpqj_shard_files marks for this rank lie back to back in that order at
// d_base (16-byte aligned, 64 readable bytes behind the last).  allgatherv_dev (may be NULL): the same collective over DEVICE
// memory -- send and receive buffers in HBM, receive pointers valid until its next call -- used for the one exchange that is
// large, the compressed d blocks: they then go HBM -> collective -> HBM and only the finished archive crosses PCIe
// (shim/rccl_gather.h: zpqr_allgatherv_dev).  Every rank returns the bytes zpqj_add returns for the whole batch on one GPU.
int zpqj_add_sharded_dev(zpq_ctx* ctx, int rank, int world, zpqj_allgatherv_fn allgatherv, zpqj_allgatherv_dev_fn allgatherv_dev, void* user,
                         const uint8_t* archive, size_t archive_len, const char* const* names, const uint8_t* d_base, const uint64_t* sizes,
                         const int64_t* dates, size_t nfiles, int64_t version_date, const char* method, uint32_t flags, uint8_t** out,
                         size_t* out_len, uint64_t stats[6]) {
  if (!out || !out_len) return ZPQ_ERR_ARG;
  if (!ctx || !d_base || !sizes || !names || !dates) { *out = nullptr; *out_len = 0; return ZPQ_ERR_ARG; }
  const Xchg X{rank, world, allgatherv, user, allgatherv_dev};
  return guarded([&] { return add_impl(&ctx, 1, archive, archive_len, names, nullptr, sizes, dates, nfiles, version_date, method, out, out_len,
                                       stats, flags, &X, d_base, nullptr); });
}

// mine[k] = 1 where rank `rank` of `world` must supply datas[k] to zpqj_add_sharded, else 0.
// zpqj_add with the files already in HBM (device-resident extents in, c/d/h/i archive out): `names` ascend (strcmp) and file k
// lies at d_base + file_off[k] .. file_off[k + 1], the files back to back in that order -- the order Jidac::add walks them in.
// d_base: 16-byte aligned, 64 readable bytes behind the last file.  Nothing of the input crosses PCIe; what comes back is
// the archive (host memory, zpqj_free).  One context; several calls on several contexts may be in flight at once.
int zpqj_add_dev(zpq_ctx* ctx, const uint8_t* archive, size_t archive_len, const char* const* names, const uint8_t* d_base,
                 const uint64_t* file_off, const int64_t* dates, size_t nfiles, int64_t version_date, const char* method,
                 uint32_t flags, uint8_t** out, size_t* out_len, uint64_t stats[6]) {
  if (!out || !out_len) return ZPQ_ERR_ARG;
  if (!ctx || !d_base || !file_off || !names || !dates) { *out = nullptr; *out_len = 0; return ZPQ_ERR_ARG; }
  return guarded([&] { return add_impl(&ctx, 1, archive, archive_len, names, nullptr, nullptr, dates, nfiles, version_date, method, out, out_len,
                                       stats, flags, nullptr, d_base, file_off); });
}

int zpqj_shard_files(const char* const* names, const uint64_t* sizes, size_t nfiles, int world, int rank, uint8_t* mine) {
  if (world < 1 || rank < 0 || rank >= world || (nfiles && (!names || !sizes || !mine))) return ZPQ_ERR_ARG;
  return guarded([&] {
    std::vector<size_t> order(nfiles), edge;
    for (size_t i = 0; i < nfiles; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return strcmp(names[a], names[b]) < 0; });
    shard_plan(order, sizes, (size_t)world, edge);
    for (size_t i = 0; i < nfiles; ++i) mine[i] = 0;
    for (size_t f = edge[rank]; f < edge[rank + 1]; ++f) mine[order[f]] = 1;
    return (int)ZPQ_OK;
  });
}

// Extracts the latest version of every file: decompresses the d blocks on the GPU, verifies every
// fragment's SHA-1 against the h table (ZSFX/zsfx.cpp:1811-1834) and returns one malloc'd blob holding
// the files back to back in name order plus their names/sizes (names: NUL-separated, malloc'd).
int zpqj_extract(zpq_ctx* ctx, const uint8_t* archive, size_t archive_len, uint8_t** data, uint64_t** sizes, char** names,
                 size_t* nfiles) {
  *data = nullptr; *sizes = nullptr; *names = nullptr; *nfiles = 0;
  const int rc = guarded([&] { return extract_impl(ctx, archive, archive_len, data, sizes, names, nfiles); });
  if (rc) { free(*data); free(*sizes); free(*names); *data = nullptr; *sizes = nullptr; *names = nullptr; *nfiles = 0; }
  return rc;
}

// Jidac::extract (ZSFX/zsfx.cpp:2018-2281, decompressThread :1731-1994) with the archive ALREADY IN HBM and the restored files LEFT
// in HBM: the host reads the index (c / h / i blocks: a few megabytes cross PCIe, the d blocks are jumped over as the reference's
// read_archive does, :1432-1461), the d blocks are decoded where they lie (stored SHA-1s checked), every fragment's SHA-1 is
// compared with the h table, the files are assembled back to back in name order in d_out, and with ZPQJ_X_SHA256 every file's
// SHA-256 is left in d_sha256 (32 bytes per file).  *file_off (nfiles + 1 offsets into d_out) and *names (NUL separated) are
// malloc'd: zpqj_free.  d_out == NULL: plan only -- the index is read, *file_off / *names / *nfiles / stats come back and
// nothing is decoded (the caller sizes d_out = file_off[nfiles] + 64 and d_sha256 from it).  d_archive must be readable for 64
// bytes behind archive_len.  stats[0..6] = files, fragments checked, bytes restored, twin files, twin bytes, bytes compared
// (ZPQJ_X_TWINS, else 0), d blocks.
int zpqj_extract_dev(zpq_ctx* ctx, const uint8_t* d_archive, size_t archive_len, uint8_t* d_out, size_t out_cap, uint8_t* d_sha256,
                     size_t sha256_cap_files, uint32_t flags, uint64_t** file_off, char** names, size_t* nfiles, uint64_t stats[7]) {
  if (!file_off || !names || !nfiles) return ZPQ_ERR_ARG;
  *file_off = nullptr; *names = nullptr; *nfiles = 0;
  if (!ctx || !d_archive) return ZPQ_ERR_ARG;
  if (!(flags & ZPQJ_X_SHA256)) d_sha256 = nullptr;
  const DevExtract dv{d_out, out_cap, d_sha256, sha256_cap_files, flags, file_off, stats};
  const int rc = guarded([&] { return extract_impl(ctx, d_archive, archive_len, nullptr, nullptr, names, nfiles, nullptr, &dv); });
  if (rc) { free(*file_off); free(*names); *file_off = nullptr; *names = nullptr; *nfiles = 0; }
  return rc;
}

// The `t` (test) command: everything zpqj_extract does except handing the files to the host -- every d block decoded with
// its stored SHA-1, every fragment's SHA-1 against the h table, the files assembled in HBM -- plus, where the i blocks
// carry zpaqfranz's per-file XXHASH64 / CRC-32 attribute, those recomputed on the device and compared.
// stats[0..6] = files, fragments checked, bytes restored, files with stored checksums, XXHASH64 mismatches, CRC-32
// mismatches, d blocks.  ZPQ_ERR_CHECKSUM if anything differs.
int zpqj_verify(zpq_ctx* ctx, const uint8_t* archive, size_t archive_len, uint64_t stats[7]) {
  for (int i = 0; i < 7; ++i) stats[i] = 0;
  return guarded([&] { return extract_impl(ctx, archive, archive_len, nullptr, nullptr, nullptr, nullptr, stats); });
}

}  // extern "C"
