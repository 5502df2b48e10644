// Decompresser, device-resident path: framed ZPAQ blocks in HBM in, original
// bytes in HBM out, many blocks per call.  Reference: Decompresser::findBlock / findFilename / readComment /
// decompress / readSegmentEnd (ZSFX/libzpaq.cpp:2239-2366) as driven per block by decompressThread
// (ZSFX/zsfx.cpp:1731-1834).
//
//   unframe_walk_kernel   one lane per block walks the framing where it lies: tag, zPQ header, segment name and
//                         comment, the stored sub-blocks ([len BE32][bytes]... 0), the 253/254 record, 255.  It
//                         emits (a) a fixed-size record per block for the host (sizes, kind, stored SHA-1) and
//                         (b) one copy extent per sub-block that strips the framing and the post-processor
//                         preamble: PASS payloads go straight to the caller's output, LZ77 level-1 streams to a
//                         contiguous staging area.
//   gather_kernel         (dedup.hip) moves the extents.
//   lz77_decode_kernel    (lz77_dec.hip) one wave per block.
//   sha1_chain_kernel     (sha.hip) one wave per block checksum; unblock_finish_kernel compares with the stored
//                         SHA-1 on the device.
// The host sees one small D2H of the per-block records and one of the results.  Blocks the walker does not take
// (context-model coded data whose end cannot be found without decoding, PCOMP programs other than the level-1
// LZ77 one, framing with more sub-blocks than the table holds) are copied to the host and go through the
// host-parsed path of block.hip -- still decoded on the GPU.
#include "zpq_internal.h"

namespace {

struct WalkJob {
  const u8* in; u32 n;
  u8* out; u32 out_cap;
  u8* stage;          // contiguous staging for an LZ77 stream (>= n bytes)
  u32 tbase, tcap;    // this block's slice of the extent table
};

struct WalkInfo {
  i32 status;         // ZPQ_OK, a negative zpq_status, or 1 = "host path"
  u32 kind;           // 0 PASS, 2 LZ77 level 1, 4 LZ77 level 1 + E8E9 inverse
  u32 hdr_off, hsize, ncomp, ph, pm;
  u32 pay_len;        // stored payload bytes including the post-processor preamble
  u32 skip;           // preamble bytes stripped (1 or 305)
  u32 nsub;
  u32 has_sha;
  u32 consumed;
  u8 sha[20];
};

__constant__ u8 c_tag[13] = {0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3};

// progs: u32 off[16], u32 len[16], then the bytes of the known level-1 post-processor programs; index 2*rb + e8
__global__ __launch_bounds__(64) void unframe_walk_kernel(const WalkJob* __restrict__ jobs, u32 njobs, const u8* __restrict__ progs,
                                                          WalkInfo* __restrict__ info, u64* __restrict__ t_src,
                                                          u64* __restrict__ t_dst, u32* __restrict__ t_len) {
  const u32 i = blockIdx.x * 64u + threadIdx.x;
  if (i >= njobs) return;
  const WalkJob J = jobs[i];
  WalkInfo I;
  memset(&I, 0, sizeof I);
  for (u32 k = 0; k < J.tcap; ++k) t_len[J.tbase + k] = 0;
  const u8* a = J.in; const u32 n = J.n;
  u32 ns = 0;                                        // extents emitted so far
  // every exit that is not ZPQ_OK takes its extents back: the gather that follows runs over the whole table, and a
  // truncated or hostile block must not have it write anything (for PASS blocks the target is the caller's buffer)
  auto done = [&](i32 st) {
    if (st != ZPQ_OK) for (u32 q = 0; q < ns; ++q) t_len[J.tbase + q] = 0;
    I.status = st; info[i] = I;
  };
  if (n < 13 + 5 + 2 + 7) return done(ZPQ_ERR_FORMAT);
  for (int k = 0; k < 13; ++k) if (a[k] != c_tag[k]) return done(ZPQ_ERR_FORMAT);
  u32 p = 13;
  if (a[p] != 'z' || a[p + 1] != 'P' || a[p + 2] != 'Q' || (a[p + 3] != 1 && a[p + 3] != 2) || a[p + 4] != 1) return done(ZPQ_ERR_FORMAT);
  p += 5;
  const u32 hsize = a[p] | (u32)a[p + 1] << 8;
  if ((u64)p + 2 + hsize > n || hsize < 7) return done(ZPQ_ERR_FORMAT);
  I.hdr_off = p; I.hsize = hsize; I.ph = a[p + 4]; I.pm = a[p + 5]; I.ncomp = a[p + 6];
  p += 2 + hsize;
  if (p >= n || a[p] != 1) return done(ZPQ_ERR_FORMAT);
  ++p;
  while (p < n && a[p]) ++p;
  ++p;
  while (p < n && a[p]) ++p;
  ++p;
  if (p >= n || a[p] != 0) return done(ZPQ_ERR_FORMAT);
  ++p;
  if (I.ncomp) return done(1);                       // arithmetic-coded: the end is only found by decoding
  // first sub-block decides the kind
  if ((u64)p + 4 > n) return done(ZPQ_ERR_FORMAT);
  u32 k0 = bswap32(*(const u32_u*)(a + p));
  if (k0 == 0 || (u64)p + 4 + k0 > n) return done(ZPQ_ERR_FORMAT);
  const u8* f = a + p + 4;
  u32 skip;
  if (f[0] == 0) { I.kind = 0; skip = 1; }
  else if (f[0] == 1) {
    if (k0 < 3) return done(1);
    const u32 psize = f[1] | (u32)f[2] << 8;
    if (k0 < 3 + psize) return done(1);              // preamble split over sub-blocks: host path
    const u32 rb = I.pm > 24 ? I.pm - 24 : 0;
    if (rb > 7) return done(1);
    const u32* poff = (const u32*)progs; const u32* plen = poff + 16;
    u32 which = 2;
    for (u32 e = 0; e < 2 && which == 2; ++e) {
      const u32 ix = 2 * rb + e;
      if (plen[ix] == 0 || plen[ix] != psize) continue;
      const u8* q = progs + poff[ix];
      bool same = true;
      for (u32 k = 0; k < psize; ++k) same &= f[3 + k] == q[k];
      if (same) which = e;
    }
    if (which == 2) return done(1);                  // some other program: ZPAQL machine on the host-parsed path
    I.kind = which ? 4 : 2; skip = 3 + psize;
  } else return done(ZPQ_ERR_FORMAT);
  I.skip = skip;
  u8* dst = I.kind == 0 ? J.out : J.stage;
  u64 acc = 0;
  for (;;) {
    if ((u64)p + 4 > n) return done(ZPQ_ERR_FORMAT);
    const u32 k = bswap32(*(const u32_u*)(a + p));
    p += 4;
    if (!k) break;
    if ((u64)p + k > n) return done(ZPQ_ERR_FORMAT);
    if (ns >= J.tcap) return done(1);
    u32 so = 0, ln = k;
    if (ns == 0) { so = skip; ln = k - skip; }
    if (I.kind == 0 && acc + k - skip > J.out_cap) return done(ZPQ_ERR_CAPACITY);   // before the extent exists
    t_src[J.tbase + ns] = (u64)(uintptr_t)(a + p + so);
    t_dst[J.tbase + ns] = (u64)(uintptr_t)(dst + (acc ? acc - skip : 0));
    t_len[J.tbase + ns] = ln;
    acc += k; ++ns; p += k;
  }
  I.nsub = ns; I.pay_len = (u32)acc;
  if (p < n && a[p] == 253 && (u64)p + 21 <= n) { I.has_sha = 1; for (int k = 0; k < 20; ++k) I.sha[k] = a[p + 1 + k]; p += 21; }
  else if (p < n && a[p] == 254) { I.has_sha = 0; ++p; }
  else return done(ZPQ_ERR_FORMAT);
  if (p >= n) return done(ZPQ_ERR_FORMAT);
  if (a[p] != 255) return done(1);                   // another segment follows: host path (its framing is parsed there)
  I.consumed = p + 1;
  done(ZPQ_OK);
}

struct FinishJob {
  u32 kind;            // 0 PASS (len known), 2 LZ77 (len/status from the decoder), 0xffffffff = not on this path
  u32 pass_len;
  u32 lz_slot;         // index into the decoder's result array
  u32 has_sha;
  u8 sha[20];
};
struct FinishOut { u32 out_len; i32 status; u8 sha1[20]; };

// lengths for the checksum chains: out_len of every block (0 for failed / foreign blocks)
__global__ __launch_bounds__(256) void unblock_lens_kernel(const FinishJob* __restrict__ fj, u32 n, const u32* __restrict__ lzres,
                                                           u32* __restrict__ lens, FinishOut* __restrict__ fo) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const FinishJob F = fj[i];
  u32 len = 0; i32 st = ZPQ_OK;
  if (F.kind == 0) len = F.pass_len;
  else if (F.kind == 2) { len = lzres[2 * F.lz_slot]; st = (i32)lzres[2 * F.lz_slot + 1]; if (st) len = 0; }
  else st = 1;
  lens[i] = len;
  fo[i].out_len = len; fo[i].status = st;
}

__global__ __launch_bounds__(256) void unblock_finish_kernel(const FinishJob* __restrict__ fj, u32 n, const u8* __restrict__ dig,
                                                             int verify, FinishOut* __restrict__ fo) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  bool same = true;
  for (int k = 0; k < 20; ++k) { const u8 d = dig[20 * (size_t)i + k]; fo[i].sha1[k] = d; same &= d == fj[i].sha[k]; }
  if (fo[i].status == ZPQ_OK && verify && fj[i].has_sha && !same) fo[i].status = ZPQ_ERR_CHECKSUM;
}

__global__ __launch_bounds__(256) void digest_compare_kernel(const u8* __restrict__ a, const u8* __restrict__ b, u32 n, u32 dsz,
                                                             u32* __restrict__ result) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  bool same = true;
  for (u32 k = 0; k < dsz; ++k) same &= a[(size_t)i * dsz + k] == b[(size_t)i * dsz + k];
  if (!same) { atomicAdd(&result[0], 1u); atomicMin(&result[1], i); }
}

}  // namespace

// The level-1 post-processor programs decoded natively (diagnostic / tests; host only, ctx may be NULL).
extern "C" int zpq_known_pcomp_bytes(uint32_t rb, int e8e9, uint8_t* out, size_t cap, size_t* len) {
  const std::vector<u8>& pc = zpq_known_pcomp(rb, e8e9 != 0);
  if (len) *len = pc.size();
  if (pc.empty()) return ZPQ_ERR_ARG;
  if (pc.size() > cap) return ZPQ_ERR_CAPACITY;
  memcpy(out, pc.data(), pc.size());
  return ZPQ_OK;
}

extern "C" int zpq_digest_compare_dev(zpq_ctx* ctx, const uint8_t* d_a, const uint8_t* d_b, size_t n, uint32_t digest_size,
                                      uint64_t* mismatches, uint64_t* first_mismatch) {
  if (ctx) (void)hipSetDevice(ctx->device);
  if (mismatches) *mismatches = 0;
  if (first_mismatch) *first_mismatch = 0;
  if (n == 0) return ZPQ_OK;
  if (n > 0xfffffff0u || digest_size == 0 || digest_size > 64) return zpq_fail(ctx, ZPQ_ERR_ARG, "digest compare: bad arguments");
  u32* d_res = (u32*)zpq_scratch(ctx, 17, 64);
  if (!d_res) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "scratch");
  const u32 init[2] = {0, 0xffffffffu};
  ZPQ_HIP(ctx, hipMemcpyAsync(d_res, init, 8, hipMemcpyHostToDevice, ctx->stream));
  ZPQ_LAUNCH(ctx, "digest_compare_kernel", ctx->stream, digest_compare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), d_a, d_b,
             (u32)n, digest_size, d_res);
  ZPQ_HIP(ctx, hipGetLastError());
  u32 res[2];
  ZPQ_HIP(ctx, hipMemcpyAsync(res, d_res, 8, hipMemcpyDeviceToHost, ctx->stream));
  ZPQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (mismatches) *mismatches = res[0];
  if (first_mismatch) *first_mismatch = res[0] ? res[1] : 0;
  return ZPQ_OK;
}

extern "C" int zpq_decompress_blocks_dev(zpq_ctx* ctx, zpq_unblock_job* jobs, size_t njobs, int verify) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (njobs == 0) return ZPQ_OK;
  if (njobs > 0x0fffffffu) return zpq_fail(ctx, ZPQ_ERR_ARG, "too many blocks");
  hipStream_t st = ctx->stream;
  // 1. extent table slices and staging areas
  std::vector<WalkJob> wj(njobs);
  size_t stage_total = 0, tab_total = 0;
  for (size_t i = 0; i < njobs; ++i) {
    jobs[i].out_len = 0; jobs[i].consumed = 0; jobs[i].status = ZPQ_OK; memset(jobs[i].sha1, 0, 20);
    wj[i].in = jobs[i].in; wj[i].n = jobs[i].n; wj[i].out = jobs[i].out; wj[i].out_cap = jobs[i].out_cap;
    wj[i].tbase = (u32)tab_total; wj[i].tcap = jobs[i].n / 65536u + 8u;
    tab_total += wj[i].tcap;
    stage_total += ((size_t)jobs[i].n + 64 + 63) & ~(size_t)63;
  }
  if (tab_total > 0x7fffffffu) return zpq_fail(ctx, ZPQ_ERR_ARG, "too many sub-blocks");
  // one arena for the per-block records: 8-byte aligned arrays first
  const size_t per_block = sizeof(WalkJob) + sizeof(zpq_lzdec_dev) + 8 + sizeof(WalkInfo) + sizeof(FinishJob) + sizeof(FinishOut) + 8 + 4 + 20;
  u8* d_meta = (u8*)zpq_scratch(ctx, 12, njobs * per_block + 1024);
  u8* d_tab = (u8*)zpq_scratch(ctx, 13, tab_total * 20 + 256);
  u8* d_stage = (u8*)zpq_scratch(ctx, 14, stage_total + 64);
  u8* d_lz1 = (u8*)zpq_scratch(ctx, 15, 16384);
  if (!d_meta || !d_tab || !d_stage || !d_lz1) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "decode scratch");
  static_assert(sizeof(WalkJob) % 8 == 0 && sizeof(zpq_lzdec_dev) % 8 == 0, "record arrays must keep 8-byte alignment");
  static_assert(sizeof(WalkInfo) % 4 == 0 && sizeof(FinishJob) % 4 == 0 && sizeof(FinishOut) % 4 == 0, "record arrays must keep 4-byte alignment");
  WalkJob* d_wj = (WalkJob*)d_meta;
  zpq_lzdec_dev* d_lzj = (zpq_lzdec_dev*)(d_wj + njobs);
  u64* d_shaoff = (u64*)(d_lzj + njobs);
  WalkInfo* d_info = (WalkInfo*)(d_shaoff + njobs);
  FinishJob* d_fj = (FinishJob*)(d_info + njobs);
  FinishOut* d_fo = (FinishOut*)(d_fj + njobs);
  u32* d_lzres = (u32*)(d_fo + njobs);
  u32* d_shalen = d_lzres + 2 * njobs;
  u8* d_dig = (u8*)(d_shalen + njobs);
  u64* t_src = (u64*)d_tab;
  u64* t_dst = t_src + tab_total;
  u32* t_len = (u32*)(t_dst + tab_total);
  {
    size_t so = 0;
    for (size_t i = 0; i < njobs; ++i) { wj[i].stage = d_stage + so; so += ((size_t)jobs[i].n + 64 + 63) & ~(size_t)63; }
  }
  ZPQ_HIP(ctx, hipMemcpyAsync(d_wj, wj.data(), njobs * sizeof(WalkJob), hipMemcpyHostToDevice, st));
  {
    static const std::vector<u8> progs = [] {        // built once per process (thread-safe initialisation): table + bytes
      std::vector<u8> g(128, 0);
      for (u32 r = 0; r < 8; ++r)
        for (u32 e = 0; e < 2; ++e) {
          const std::vector<u8>& pc = zpq_known_pcomp(r, e != 0);
          const u32 off = (u32)g.size(), len = (u32)pc.size();
          memcpy(&g[4 * (2 * r + e)], &off, 4); memcpy(&g[64 + 4 * (2 * r + e)], &len, 4);
          g.insert(g.end(), pc.begin(), pc.end());
        }
      return g;
    }();
    if (progs.size() > 16384) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "post-processor table");
    ZPQ_HIP(ctx, hipMemcpyAsync(d_lz1, progs.data(), progs.size(), hipMemcpyHostToDevice, st));
  }
  ZPQ_LAUNCH(ctx, "unframe_walk_kernel", st, unframe_walk_kernel, dim3((unsigned)((njobs + 63) / 64)), dim3(64), d_wj, (u32)njobs, d_lz1,
             d_info, t_src, t_dst, t_len);
  ZPQ_HIP(ctx, hipGetLastError());
  std::vector<WalkInfo> info(njobs);
  ZPQ_HIP(ctx, hipMemcpyAsync(info.data(), d_info, njobs * sizeof(WalkInfo), hipMemcpyDeviceToHost, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  // 2. strip the framing (one gather over every sub-block of every block), undo LZ77
  int first_err = ZPQ_OK;
  std::vector<FinishJob> fj(njobs);
  std::vector<zpq_lzdec_dev> lzj;
  std::vector<size_t> lzj_job;
  std::vector<u64> shaoff(njobs);
  std::vector<size_t> host_path;
  std::vector<size_t> e8_jobs;                      // LZ77 output goes to a temporary, the E8E9 inverse writes the caller's buffer
  size_t e8_total = 0, e8_nstate = 0; u32 e8_maxcap = 0;
  for (size_t i = 0; i < njobs; ++i) {
    const WalkInfo& I = info[i];
    memset(&fj[i], 0, sizeof fj[i]);
    fj[i].kind = 0xffffffffu;
    shaoff[i] = (u64)(uintptr_t)jobs[i].out;
    if (I.status == 1) { host_path.push_back(i); continue; }
    if (I.status != ZPQ_OK) {
      jobs[i].status = zpq_fail(ctx, I.status, "block %zu: %s", i, zpq_strerror(I.status));
      if (!first_err) first_err = jobs[i].status;
      continue;
    }
    jobs[i].consumed = I.consumed;
    fj[i].has_sha = I.has_sha; memcpy(fj[i].sha, I.sha, 20);
    if (I.kind == 0) { fj[i].kind = 0; fj[i].pass_len = I.pay_len - I.skip; }
    else {
      fj[i].kind = 2; fj[i].lz_slot = (u32)lzj.size();
      zpq_lzdec_dev d;
      d.in = wj[i].stage; d.n = I.pay_len - I.skip; d.rb = I.pm > 24 ? I.pm - 24 : 0;
      d.out = jobs[i].out; d.out_cap = jobs[i].out_cap; d.result = d_lzres + 2 * lzj.size();
      if (d.rb > 8) { jobs[i].status = ZPQ_ERR_FORMAT; fj[i].kind = 0xffffffffu; if (!first_err) first_err = ZPQ_ERR_FORMAT; continue; }
      if (I.kind == 4) {
        e8_jobs.push_back(i);
        e8_total += ((size_t)jobs[i].out_cap + 128 + 63) & ~(size_t)63;
        e8_nstate += (size_t)jobs[i].out_cap / 1024 + 2;
        if (jobs[i].out_cap > e8_maxcap) e8_maxcap = jobs[i].out_cap;
      }
      lzj.push_back(d); lzj_job.push_back(i);
    }
  }
  u8* d_e8 = nullptr; u8* d_e8meta = nullptr;
  std::vector<zpq_e8inv_job> e8j;
  if (!e8_jobs.empty()) {
    d_e8 = (u8*)zpq_scratch(ctx, 22, e8_total + 64);
    d_e8meta = (u8*)zpq_scratch(ctx, 20, e8_jobs.size() * sizeof(zpq_e8inv_job) + (2 * e8_nstate + 16) * 4 + 64);
    if (!d_e8 || !d_e8meta) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "e8e9 staging");
    size_t o = 0, so = 0, q = 0;
    for (size_t k = 0; k < lzj.size(); ++k) {
      const size_t i = lzj_job[k];
      if (q >= e8_jobs.size() || e8_jobs[q] != i) continue;
      zpq_e8inv_job e;
      e.in = d_e8 + o; e.out = jobs[i].out; e.len = d_lzres + 2 * k; e.cap = jobs[i].out_cap; e.st_base = (u32)so;
      lzj[k].out = d_e8 + o;
      o += ((size_t)jobs[i].out_cap + 128 + 63) & ~(size_t)63; so += (size_t)jobs[i].out_cap / 1024 + 2; ++q;
      e8j.push_back(e);
    }
  }
  int rc = zpq_gather_dev(ctx, (const u8*)0, t_src, t_len, t_dst, tab_total, (u8*)0);
  if (rc) return rc;
  if (!lzj.empty()) {
    ZPQ_HIP(ctx, hipMemcpyAsync(d_lzj, lzj.data(), lzj.size() * sizeof(zpq_lzdec_dev), hipMemcpyHostToDevice, st));
    if ((rc = zpq_lz77_decode_launch(ctx, st, lzj.data(), d_lzj, lzj.size()))) return rc;
  }
  if (!e8j.empty()) {
    zpq_e8inv_job* d_e8j = (zpq_e8inv_job*)d_e8meta;
    u32* d_state = (u32*)(d_e8meta + ((e8j.size() * sizeof(zpq_e8inv_job) + 63) & ~(size_t)63));
    ZPQ_HIP(ctx, hipMemcpyAsync(d_e8j, e8j.data(), e8j.size() * sizeof(zpq_e8inv_job), hipMemcpyHostToDevice, st));
    ZPQ_HIP(ctx, hipMemsetAsync(d_state + 2 * e8_nstate, 0, 4, st));
    if ((rc = zpq_e8e9_inverse_launch(ctx, st, d_e8j, e8j.size(), e8_maxcap, d_state, e8_nstate))) return rc;
  }
  // 3. checksums of the results, compared on the device
  ZPQ_HIP(ctx, hipMemcpyAsync(d_fj, fj.data(), njobs * sizeof(FinishJob), hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipMemcpyAsync(d_shaoff, shaoff.data(), njobs * 8, hipMemcpyHostToDevice, st));
  const unsigned g256 = (unsigned)((njobs + 255) / 256);
  ZPQ_LAUNCH(ctx, "unblock_lens_kernel", st, unblock_lens_kernel, dim3(g256), dim3(256), d_fj, (u32)njobs, d_lzres, d_shalen, d_fo);
  ZPQ_HIP(ctx, hipGetLastError());
  if ((rc = zpq_sha1_chains_on(ctx, st, (const u8*)0, d_shaoff, d_shalen, njobs, d_dig))) return rc;
  ZPQ_LAUNCH(ctx, "unblock_finish_kernel", st, unblock_finish_kernel, dim3(g256), dim3(256), d_fj, (u32)njobs, d_dig, verify, d_fo);
  ZPQ_HIP(ctx, hipGetLastError());
  std::vector<FinishOut> fo(njobs);
  ZPQ_HIP(ctx, hipMemcpyAsync(fo.data(), d_fo, njobs * sizeof(FinishOut), hipMemcpyDeviceToHost, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  for (size_t i = 0; i < njobs; ++i) {
    if (fj[i].kind == 0xffffffffu) continue;
    jobs[i].out_len = fo[i].out_len; jobs[i].status = fo[i].status; memcpy(jobs[i].sha1, fo[i].sha1, 20);
    jobs[i].nseg = 1;
    if (jobs[i].seg_out_end && jobs[i].seg_cap) jobs[i].seg_out_end[0] = jobs[i].out_len;
    if (jobs[i].status && !first_err) first_err = jobs[i].status;
  }
  // 4. the rest: host-parsed (still GPU-decoded), one copy of the block to the host each
  if (!host_path.empty()) {
    std::vector<std::vector<u8>> hb(host_path.size());
    std::vector<zpq_unblock_job> hj(host_path.size());
    for (size_t k = 0; k < host_path.size(); ++k) {
      const size_t i = host_path[k];
      hb[k].resize((size_t)jobs[i].n + 64);
      ZPQ_HIP(ctx, hipMemcpyAsync(hb[k].data(), jobs[i].in, jobs[i].n, hipMemcpyDeviceToHost, st));
      hj[k] = jobs[i];
      hj[k].in = hb[k].data();
    }
    ZPQ_HIP(ctx, hipStreamSynchronize(st));
    rc = zpq_decompress_hostparsed(ctx, hj.data(), hj.size(), verify, true);
    for (size_t k = 0; k < host_path.size(); ++k) {
      const size_t i = host_path[k];
      const u8* in = jobs[i].in;
      jobs[i] = hj[k];
      jobs[i].in = in;
      if (jobs[i].status && !first_err) first_err = jobs[i].status;
    }
    if (rc && !first_err) first_err = rc;
  }
  return first_err;
}

// Host buffers in and out: staged through HBM, decoded by the device path.
extern "C" int zpq_decompress_blocks(zpq_ctx* ctx, zpq_unblock_job* jobs, size_t njobs, int verify) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (njobs == 0) return ZPQ_OK;
  hipStream_t st = ctx->stream;
  size_t in_total = 0, out_total = 0;
  for (size_t i = 0; i < njobs; ++i) {
    in_total += ((size_t)jobs[i].n + 64 + 63) & ~(size_t)63;
    out_total += ((size_t)jobs[i].out_cap + 64 + 63) & ~(size_t)63;
  }
  u8* d_in = (u8*)zpq_scratch(ctx, 16, in_total + 64);
  u8* d_out = (u8*)zpq_scratch(ctx, 6, out_total + 64);
  if (!d_in || !d_out) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "decode staging");
  std::vector<zpq_unblock_job> dj(jobs, jobs + njobs);
  size_t io = 0, oo = 0;
  for (size_t i = 0; i < njobs; ++i) {
    dj[i].in = d_in + io; dj[i].out = d_out + oo;
    if (jobs[i].n) ZPQ_HIP(ctx, hipMemcpyAsync(d_in + io, jobs[i].in, jobs[i].n, hipMemcpyHostToDevice, st));
    io += ((size_t)jobs[i].n + 64 + 63) & ~(size_t)63;
    oo += ((size_t)jobs[i].out_cap + 64 + 63) & ~(size_t)63;
  }
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  const int rc = zpq_decompress_blocks_dev(ctx, dj.data(), njobs, verify);
  for (size_t i = 0; i < njobs; ++i) {
    const u8* in = jobs[i].in; u8* out = jobs[i].out;
    jobs[i] = dj[i];
    jobs[i].in = in; jobs[i].out = out;
    if (dj[i].out_len) ZPQ_HIP(ctx, hipMemcpyAsync(out, dj[i].out, dj[i].out_len, hipMemcpyDeviceToHost, st));
  }
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  return rc;
}
