// Whole-file duplicates found by COMPARING bytes, before anything is hashed ("twin files").
//
// Jidac::add learns that two files hold the same data only after it has fragmented and SHA-1'd both of them
// (the fragment ids meet in the index, ZSFX/zsfx.cpp:651-659).  Fragment boundaries and fragment
// ids are functions of a file's bytes alone -- the rolling hash, o1[] and the size counter start afresh at every file
// -- so a file whose bytes EQUAL those of an earlier file has that file's fragment list, offset by the distance between
// the two, and that file's ids.  Establishing equality costs two streaming reads and an XOR per 16 bytes (HBM bound);
// fragmenting and hashing cost ~24 integer instructions per byte (issue bound, ~6x slower on this chip).  So:
//   1. twin_probe_kernel: files whose length occurs more than once get a 64-bit fingerprint of eight 16-byte samples;
//      the first file of every (length, fingerprint) class is the class's representative;
//   2. twin_compare_kernel: every other file of a class is compared with its representative, byte for byte.  Work items
//      are (chunk of the representative, member) in chunk-major order, so the workgroups that run at one time all read
//      the same few representative chunks (L2 / Infinity Cache resident) and HBM carries each member's bytes once;
//   3. a member that matched is a twin: the callers fragment and hash the representatives only and replicate the records.
// Nothing here is a heuristic: a twin is declared only after all of its bytes were compared, a file that differs in one
// bit is fragmented and hashed like any other, and with no equal lengths in a call nothing is launched at all.
#include <algorithm>
#include <unordered_map>

#include "zpq_internal.h"

namespace {

constexpr u32 kChunk = 256u << 10;     // bytes of one (chunk, member) work item: 16 rounds of 4 x 16 B per thread
constexpr u32 kProbes = 8;

struct TwinGroup { u64 item_base, rep_off, len; u32 mem_base, nmem; };

__device__ __forceinline__ u64 probe_mix(u64 h, const u32x4 v) {
  h = (h ^ ((u64)v.x | (u64)v.y << 32)) * 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  h = (h ^ ((u64)v.z | (u64)v.w << 32)) * 0xC2B2AE3D27D4EB4Full;
  return h ^ (h >> 32);
}

__global__ __launch_bounds__(256) void twin_probe_kernel(const u8* __restrict__ data, const u64* __restrict__ off,
                                                          const u64* __restrict__ len, u32 n, u64* __restrict__ fp) {
  const u32 f = blockIdx.x * 256 + threadIdx.x;
  if (f >= n) return;
  const u64 L = len[f];
  u64 h = L * 0x9E3779B97F4A7C15ull + 1;
  if (L >= 16) {
    const u8* p = data + off[f];
    const u64 step = (L - 16) / (kProbes - 1);
    u32x4 v[kProbes];
#pragma unroll
    for (u32 i = 0; i < kProbes; ++i) v[i] = *(const u32x4_u*)(p + (i + 1 == kProbes ? L - 16 : step * i));
#pragma unroll
    for (u32 i = 0; i < kProbes; ++i) h = probe_mix(h, v[i]);
  }
  fp[f] = h;
}

__device__ __forceinline__ u32 xor_or(const u32x4 a, const u32x4 b) { return (a.x ^ b.x) | (a.y ^ b.y) | (a.z ^ b.z) | (a.w ^ b.w); }

// One (chunk, member) item per workgroup round.  The member side -- the stream HBM has to deliver -- is read with
// 16-byte ALIGNED non-temporal loads (up to 15 head bytes of the file are compared on their own), the representative
// side with whatever alignment the distance between the two files leaves (cache resident).
__global__ __launch_bounds__(256) void twin_compare_kernel(const u8* __restrict__ data, const TwinGroup* __restrict__ grp, u32 ngrp,
                                                            const u64* __restrict__ mem_off, u64 nitems, u32* __restrict__ bad) {
  const u32 t = threadIdx.x;
  for (u64 item = blockIdx.x; item < nitems; item += gridDim.x) {
    u32 lo = 0, hi = ngrp;
    while (hi - lo > 1) {
      const u32 mid = (lo + hi) >> 1;
      if (grp[mid].item_base <= item) lo = mid; else hi = mid;
    }
    const TwinGroup G = grp[lo];
    const u64 local = item - G.item_base;
    const u64 c = local / G.nmem;
    const u32 m = G.mem_base + (u32)(local - c * G.nmem);
    if (__hip_atomic_load(bad + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;   // already known to differ
    const u8* A = data + mem_off[m];
    const u8* B = data + G.rep_off;
    const u32 head = (u32)((16u - (u32)((uintptr_t)A & 15u)) & 15u);
    const u64 start = (u64)head + c * (u64)kChunk;
    if (start >= G.len && !(c == 0 && head)) continue;
    u32 diff = 0;
    if (start < G.len) {
      const u64 end = start + kChunk < G.len ? start + kChunk : G.len;
      const u32 nbytes = (u32)(end - start), nwords = nbytes >> 4, tail = nbytes & 15u;
      const u32x4* Aw = (const u32x4*)(A + start);
      const u8* Bp = B + start;
      u32 w = t;
      for (; w + 768 < nwords; w += 1024) {
        const u32x4 a0 = __builtin_nontemporal_load(Aw + w), a1 = __builtin_nontemporal_load(Aw + w + 256),
                    a2 = __builtin_nontemporal_load(Aw + w + 512), a3 = __builtin_nontemporal_load(Aw + w + 768);
        const u32x4 b0 = *(const u32x4_u*)(Bp + 16ull * w), b1 = *(const u32x4_u*)(Bp + 16ull * (w + 256)),
                    b2 = *(const u32x4_u*)(Bp + 16ull * (w + 512)), b3 = *(const u32x4_u*)(Bp + 16ull * (w + 768));
        diff |= xor_or(a0, b0) | xor_or(a1, b1) | xor_or(a2, b2) | xor_or(a3, b3);
      }
      for (; w < nwords; w += 256) diff |= xor_or(Aw[w], *(const u32x4_u*)(Bp + 16ull * w));
      if (t < tail) diff |= (u32)(A[start + 16ull * nwords + t] ^ Bp[16ull * nwords + t]);
    }
    if (c == 0 && t < head) diff |= (u32)(A[t] ^ B[t]);
    if (__any(diff != 0) && (t & 63u) == 0) __hip_atomic_store(bad + m, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// digest of file f = digest of its representative (W 32-bit words each)
template <int W>
__global__ __launch_bounds__(256) void twin_digest_spread_kernel(u32 nfiles, const u32* __restrict__ uidx, const u32* __restrict__ udig,
                                                                  u32* __restrict__ dig) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nfiles * W) return;
  const u32 f = i / W, w = i - f * W;
  dig[i] = udig[(size_t)uidx[f] * W + w];
}

}  // namespace

// rep[f] = the earliest file holding the same bytes as file f (established by comparing all of them), else f.
// off / len are HOST arrays of n extents in d_base.  Files shorter than min_bytes are left alone.  Synchronous on st.
// stats (may be null): [0] twins found, [1] their bytes, [2] files compared, [3] bytes compared.
int zpq_twins_find(zpq_ctx* ctx, hipStream_t st, const u8* d_base, const u64* off, const u64* len, size_t n, u64 min_bytes, u32* rep,
                   u64 stats[4]) {
  if (ctx) (void)hipSetDevice(ctx->device);
  for (size_t f = 0; f < n; ++f) rep[f] = (u32)f;
  if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
  if (n < 2) return ZPQ_OK;
  if (n > 0x7fffffffu) return zpq_fail(ctx, ZPQ_ERR_ARG, "too many files");
  if (min_bytes < 16) min_bytes = 16;
  // 1. only lengths that occur more than once can have twins
  std::unordered_map<u64, u32> seen;
  seen.reserve(n * 2);
  for (size_t f = 0; f < n; ++f)
    if (len[f] >= min_bytes) ++seen[len[f]];
  std::vector<u32> pidx;
  for (size_t f = 0; f < n; ++f)
    if (len[f] >= min_bytes && seen[len[f]] > 1) pidx.push_back((u32)f);
  const size_t np = pidx.size();
  if (np < 2) return ZPQ_OK;
  // 2. fingerprints
  std::vector<u64> h(np * 2);
  for (size_t k = 0; k < np; ++k) { h[k] = off[pidx[k]]; h[np + k] = len[pidx[k]]; }
  u64* d_p = (u64*)zpq_scratch(ctx, 28, np * 24 + 256);
  if (!d_p) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "twin scratch");
  ZPQ_HIP(ctx, hipMemcpyAsync(d_p, h.data(), np * 16, hipMemcpyHostToDevice, st));
  ZPQ_LAUNCH(ctx, "twin_probe_kernel", st, twin_probe_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), d_base, d_p, d_p + np, (u32)np,
             d_p + 2 * np);
  ZPQ_HIP(ctx, hipGetLastError());
  std::vector<u64> fp(np);
  ZPQ_HIP(ctx, hipMemcpyAsync(fp.data(), d_p + 2 * np, np * 8, hipMemcpyDeviceToHost, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  // 3. classes of (length, fingerprint): first member = representative
  struct Key { u64 l, f; bool operator==(const Key& o) const { return l == o.l && f == o.f; } };
  struct KeyHash { size_t operator()(const Key& k) const { return (size_t)(k.l * 0x9E3779B97F4A7C15ull ^ k.f); } };
  std::unordered_map<Key, u32, KeyHash> cls;      // -> group number
  cls.reserve(np * 2);
  std::vector<u32> grp_rep;                       // representative file of every class
  std::vector<std::vector<u32>> grp_mem;          // its other files
  for (size_t k = 0; k < np; ++k) {
    const Key key{len[pidx[k]], fp[k]};
    auto it = cls.find(key);
    if (it == cls.end()) { cls.emplace(key, (u32)grp_rep.size()); grp_rep.push_back(pidx[k]); grp_mem.emplace_back(); }
    else grp_mem[it->second].push_back(pidx[k]);
  }
  std::vector<TwinGroup> G;
  std::vector<u64> mem_off;
  std::vector<u32> mem_file;
  u64 nitems = 0, cmp_bytes = 0;
  for (size_t g = 0; g < grp_rep.size(); ++g) {
    if (grp_mem[g].empty()) continue;
    TwinGroup tg;
    tg.item_base = nitems; tg.rep_off = off[grp_rep[g]]; tg.len = len[grp_rep[g]];
    tg.mem_base = (u32)mem_off.size(); tg.nmem = (u32)grp_mem[g].size();
    for (u32 f : grp_mem[g]) { mem_off.push_back(off[f]); mem_file.push_back(f); rep[f] = grp_rep[g]; }
    nitems += (tg.len + kChunk - 1) / kChunk * (u64)tg.nmem;
    cmp_bytes += tg.len * tg.nmem;
    G.push_back(tg);
  }
  const size_t nm = mem_off.size();
  if (nm == 0) return ZPQ_OK;
  // 4. compare
  const size_t gbytes = (G.size() * sizeof(TwinGroup) + 63) & ~(size_t)63, mbytes = (nm * 8 + 63) & ~(size_t)63;
  u8* d_c = (u8*)zpq_scratch(ctx, 29, gbytes + mbytes + nm * 4 + 256);
  if (!d_c) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "twin scratch");
  TwinGroup* d_grp = (TwinGroup*)d_c;
  u64* d_mem = (u64*)(d_c + gbytes);
  u32* d_bad = (u32*)(d_c + gbytes + mbytes);
  ZPQ_HIP(ctx, hipMemcpyAsync(d_grp, G.data(), G.size() * sizeof(TwinGroup), hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipMemcpyAsync(d_mem, mem_off.data(), nm * 8, hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipMemsetAsync(d_bad, 0, nm * 4, st));
  int wg_per_cu = 8;
  if (const char* e = getenv("ZPQ_TWIN_WG")) { const int v = atoi(e); if (v >= 1 && v <= 16) wg_per_cu = v; }
  const u64 grid = std::min<u64>(nitems, (u64)ctx->cu_count * (u64)wg_per_cu);
  ZPQ_LAUNCH(ctx, "twin_compare_kernel", st, twin_compare_kernel, dim3((unsigned)grid), dim3(256), d_base, d_grp, (u32)G.size(), d_mem, nitems,
             d_bad);
  ZPQ_HIP(ctx, hipGetLastError());
  std::vector<u32> bad(nm);
  ZPQ_HIP(ctx, hipMemcpyAsync(bad.data(), d_bad, nm * 4, hipMemcpyDeviceToHost, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  u64 ntw = 0, tw_bytes = 0;
  for (size_t k = 0; k < nm; ++k) {
    if (bad[k]) rep[mem_file[k]] = mem_file[k];
    else { ++ntw; tw_bytes += len[mem_file[k]]; }
  }
  if (stats) { stats[0] = ntw; stats[1] = tw_bytes; stats[2] = nm; stats[3] = cmp_bytes; }
  return ZPQ_OK;
}

extern "C" {

int zpq_file_twins_dev(zpq_ctx* ctx, const uint8_t* d_base, const uint64_t* file_off, size_t nfiles, uint64_t min_bytes, uint32_t* file_rep,
                       uint64_t stats[4]) {
  if (!ctx || !file_off || !file_rep) return ZPQ_ERR_ARG;
  std::vector<u64> len(nfiles);
  for (size_t f = 0; f < nfiles; ++f) {
    if (file_off[f + 1] < file_off[f]) return zpq_fail(ctx, ZPQ_ERR_ARG, "file_off not monotone");
    len[f] = file_off[f + 1] - file_off[f];
  }
  return zpq_twins_find(ctx, ctx->stream, d_base, file_off, len.data(), nfiles, min_bytes, file_rep, stats);
}

// SHA-256 of every file (the per-file check of extract, ZSFX/zsfx.cpp:2018-2281 / AUTOTEST) with the twin fold in front:
// a restored file whose bytes equal an earlier restored file's -- every byte compared -- has that file's digest.
int zpq_sha256_files_dev(zpq_ctx* ctx, const uint8_t* d_base, const uint64_t* file_off, size_t nfiles, uint8_t* d_digests, uint32_t flags,
                         uint64_t stats[4]) {
  if (!ctx || !file_off || !d_digests) return ZPQ_ERR_ARG;
  (void)hipSetDevice(ctx->device);
  if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
  if (nfiles == 0) return ZPQ_OK;
  if (nfiles > 0x7fffffffu) return zpq_fail(ctx, ZPQ_ERR_ARG, "too many files");
  if (((uintptr_t)d_digests & 3) != 0) return zpq_fail(ctx, ZPQ_ERR_ARG, "d_digests must be 4-byte aligned");
  hipStream_t st = ctx->stream;
  std::vector<u64> len(nfiles);
  for (size_t f = 0; f < nfiles; ++f) {
    if (file_off[f + 1] < file_off[f]) return zpq_fail(ctx, ZPQ_ERR_ARG, "file_off not monotone");
    len[f] = file_off[f + 1] - file_off[f];
  }
  std::vector<u32> rep(nfiles);
  static const bool env_off = [] { const char* e = getenv("ZPQ_TWINS"); return e && atoi(e) == 0; }();
  u64 tst[4] = {0, 0, 0, 0};
  if (!(flags & ZPQ_FS_NO_TWINS) && !env_off) {
    const int rc = zpq_twins_find(ctx, st, d_base, file_off, len.data(), nfiles, 4096, rep.data(), tst);
    if (rc) return rc;
  } else {
    for (size_t f = 0; f < nfiles; ++f) rep[f] = (u32)f;
  }
  if (stats) memcpy(stats, tst, sizeof tst);
  // the representatives, back to back: [off | len] u64, then uidx u32 per file, then their digests
  std::vector<u64> h;
  std::vector<u32> uidx(nfiles);
  std::vector<u64> uoff, ulen;
  for (size_t f = 0; f < nfiles; ++f)
    if (rep[f] == f) { uidx[f] = (u32)uoff.size(); uoff.push_back(file_off[f]); ulen.push_back(len[f]); }
  for (size_t f = 0; f < nfiles; ++f) uidx[f] = uidx[rep[f]];
  const size_t nu = uoff.size();
  const bool fold = nu != nfiles;
  const size_t o_len = nu * 8, o_idx = o_len + nu * 8, o_dig = (o_idx + nfiles * 4 + 63) & ~(size_t)63;
  u8* d_t = (u8*)zpq_scratch(ctx, 28, o_dig + nu * 32 + 256);
  if (!d_t) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "twin scratch");
  ZPQ_HIP(ctx, hipMemcpyAsync(d_t, uoff.data(), nu * 8, hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipMemcpyAsync(d_t + o_len, ulen.data(), nu * 8, hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipMemcpyAsync(d_t + o_idx, uidx.data(), nfiles * 4, hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));      // the tables above are locals
  int rc = zpq_sha256_extents_dev(ctx, d_base, (const u64*)d_t, (const u64*)(d_t + o_len), nu, fold ? d_t + o_dig : d_digests);
  if (rc || !fold) return rc;
  ZPQ_LAUNCH(ctx, "twin_digest_spread_kernel", st, twin_digest_spread_kernel<8>, dim3((unsigned)((nfiles * 8 + 255) / 256)), dim3(256), (u32)nfiles,
             (const u32*)(d_t + o_idx), (const u32*)(d_t + o_dig), (u32*)d_digests);
  ZPQ_HIP(ctx, hipGetLastError());
  return ZPQ_OK;
}

}  // extern "C"
