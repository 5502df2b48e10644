// Dedup index: "a fragment is new iff its SHA-1 was never seen before"
// (reference: HT{sha1[20],usize} ZSFX/zsfx.cpp:651-659; the add-side index lives in the missing
// zpaqfranz.cpp).  Open-addressing table in HBM keyed by the 20-byte digest; every slot converges to
// the SMALLEST fragment index of its digest class (atomicMin), so first[] is the first occurrence
// in fragment order no matter how the lanes were scheduled -- the property that keeps the archive
// independent of GPU count.
#include "zpq_internal.h"

namespace {

constexpr u32 kEmpty = 0xffffffffu;

__device__ __forceinline__ bool same_digest(const u8* __restrict__ dig, u32 i, u32 j) {
  const u32* a = (const u32*)(dig + (size_t)i * 20);
  const u32* b = (const u32*)(dig + (size_t)j * 20);
  return a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3] && a[4] == b[4];
}

__device__ __forceinline__ u32 slot_of(const u8* __restrict__ dig, u32 i, u32 mask) {
  const u32* a = (const u32*)(dig + (size_t)i * 20);
  return (a[0] * 2654435761u ^ a[1]) & mask;  // SHA-1 output is already uniform
}

__global__ __launch_bounds__(256) void dedup_insert_kernel(const u8* __restrict__ dig, u32 n, u32* __restrict__ table,
                                                           u32 mask) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  u32 h = slot_of(dig, i, mask);
  for (;;) {
    u32 cur = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == kEmpty) {
      u32 old = atomicCAS(&table[h], kEmpty, i);
      if (old == kEmpty) return;
      cur = old;
    }
    if (same_digest(dig, cur, i)) { atomicMin(&table[h], i); return; }  // a slot never changes class
    h = (h + 1) & mask;
  }
}

__global__ __launch_bounds__(256) void dedup_lookup_kernel(const u8* __restrict__ dig, u32 n,
                                                           const u32* __restrict__ table, u32 mask,
                                                           u32* __restrict__ first) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  u32 h = slot_of(dig, i, mask);
  for (;;) {
    u32 cur = table[h];
    if (same_digest(dig, cur, i)) { first[i] = cur; return; }
    h = (h + 1) & mask;
  }
}

}  // namespace

extern "C" int zpq_dedup_dev(zpq_ctx* ctx, const uint8_t* d_digests, size_t n, uint32_t* d_first) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (n == 0) return ZPQ_OK;
  if (n > 0x7fffffffu) return zpq_fail(ctx, ZPQ_ERR_ARG, "too many fragments");
  u32 size = 1024;
  while (size < 2 * n) size <<= 1;
  u32* table = (u32*)zpq_scratch(ctx, 5, (size_t)size * 4);
  if (!table) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "dedup table");
  ZPQ_HIP(ctx, hipMemsetAsync(table, 0xff, (size_t)size * 4, ctx->stream));
  const unsigned grid = (unsigned)((n + 255) / 256);
  ZPQ_LAUNCH(ctx, "dedup_insert_kernel", ctx->stream, dedup_insert_kernel, dim3(grid), dim3(256), d_digests, (u32)n, table, size - 1);
  ZPQ_HIP(ctx, hipGetLastError());
  ZPQ_LAUNCH(ctx, "dedup_lookup_kernel", ctx->stream, dedup_lookup_kernel, dim3(grid), dim3(256), d_digests, (u32)n, table, size - 1,
                     d_first);
  ZPQ_HIP(ctx, hipGetLastError());
  return ZPQ_OK;
}

// ---- extent gather: the data movement of the block packer (row a4) ------------------------------------
namespace {
// One workgroup per (extent, piece slot): slot p copies the 16 KiB pieces p, p+P, p+2P, ... of its extent, so
// extents of any length are covered; the grid is persistent over the (extent, slot) items.  16-byte vector copies when source and destination are mutually aligned, bytes
// otherwise.  HBM-bound copy: traffic = 2 x bytes moved.
__global__ __launch_bounds__(256) void gather_kernel(const u8* __restrict__ src_base, const u64* __restrict__ src_off,
                                                     const u32* __restrict__ len, const u64* __restrict__ dst_off,
                                                     u8* __restrict__ dst_base, u32 pieces_per_extent, u64 items) {
 for (u64 w = blockIdx.x; w < items; w += gridDim.x) {
  const u32 e = (u32)(w / pieces_per_extent), piece = (u32)(w % pieces_per_extent);
  const u32 n = len[e];
  for (u64 lo = (u64)piece * 16384u; lo < n; lo += (u64)pieces_per_extent * 16384u) {
    const u32 cnt = n - lo < 16384u ? (u32)(n - lo) : 16384u;
    const u8* s = src_base + src_off[e] + lo;
    u8* d = dst_base + dst_off[e] + lo;
    u32 head = (u32)((16 - ((uintptr_t)d & 15)) & 15);
    if (head > cnt) head = cnt;
    for (u32 i = threadIdx.x; i < head; i += 256) d[i] = s[i];
    const u32 body = (cnt - head) & ~15u;
    for (u32 i = threadIdx.x * 16; i < body; i += 256 * 16)
      *(u32x4*)(d + head + i) = *(const u32x4_u*)(s + head + i);
    for (u32 i = head + body + threadIdx.x; i < cnt; i += 256) d[i] = s[i];
  }
 }
}
}  // namespace

extern "C" int zpq_gather_dev(zpq_ctx* ctx, const uint8_t* d_src_base, const uint64_t* d_src_off, const uint32_t* d_len,
                              const uint64_t* d_dst_off, size_t n, uint8_t* d_dst_base) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (n == 0) return ZPQ_OK;
  // piece slots per extent (longer extents loop inside the kernel); with very many extents (the extract side
  // scatters every fragment of every file) fewer slots and a persistent grid keep the launch small
  const u32 pieces = n >= 16384 ? 4 : 64;
  const u64 items = (u64)n * pieces;
  const unsigned grid = (unsigned)(items < (1u << 20) ? items : (1u << 20));   // one workgroup per item while that is a sane grid
  ZPQ_LAUNCH(ctx, "gather_kernel", ctx->stream, gather_kernel, dim3(grid), dim3(256), d_src_base,
             d_src_off, d_len, d_dst_off, d_dst_base, pieces, items);
  ZPQ_HIP(ctx, hipGetLastError());
  return ZPQ_OK;
}

// ---- per-fragment statistics for the block method hint (row a4) -----------------------------------------------------------
// zpaq's add() passes "method,R,t" to compressBlock (ZSFX/libzpaq.h:86-135): R = 0..255 redundancy of the block, t bit 0
// = text, bit 1 = x86.  R comes from the order-1 prediction hits the fragment loop counts anyway (c == o1[c1], table reset
// per fragment -- SURVEY.md Appendix C.4); the text / exe detectors live in the missing zpaqfranz.cpp, so the ones here are
// this engine's own (PARITY UNPINNED): a fragment votes "text" when at least 15/16 of its bytes are
// letters, digits, blanks or common punctuation and none is a control byte below 9, and "exe" when at least one byte in
// 48 is 0x8B (mov reg, r/m) or an E8/E9 followed four bytes later by 00/FF.
// One lane per fragment (unique fragments only: a few hundred MB at most per job), o1[] in LDS, 16 bytes per load.
namespace {
__global__ __launch_bounds__(64) void fragment_stats_kernel(const u8* __restrict__ base, const u64* __restrict__ off, const u32* __restrict__ len,
                                                            u32 n, u32* __restrict__ stats) {
  __shared__ u8 tab[256 * 64];
  const u32 lane = (u32)lane_id();
  const u32 f = blockIdx.x * 64u + lane;
  for (u32 v = 0; v < 256; ++v) tab[v * 64 + lane] = 0;
  __builtin_amdgcn_wave_barrier();
  if (f >= n) return;
  const u8* p = base + off[f];
  const u32 L = len[f];
  u32 hits = 0, text = 0, ctrl = 0, x86 = 0, c1 = 0, w4 = 0;     // w4: the last four bytes (oldest in the high byte)
  for (u32 i = 0; i < L; i += 16) {
    const u32x4 d = *(const u32x4_u*)(p + i);                    // (readable 64 bytes past the end: include/zpaqhip.h)
    const u32 wv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (i + j >= L) break;
      const u32 c = (wv[j >> 2] >> (8 * (j & 3))) & 255u;
      const u32 a = c1 * 64 + lane;
      hits += tab[a] == c;
      tab[a] = (u8)c;
      c1 = c;
      const bool alnum = (c - 'a' < 26u) || (c - 'A' < 26u) || (c - '0' < 10u);
      text += alnum || c == ' ' || c == '.' || c == ',' || c == '\n' || c == '\r' || c == '\t' || c == ';' || c == ':' || c == '\'' || c == '"' || c == '-' || c >= 128;
      ctrl += c < 9;
      const u32 old = w4 >> 24;                                  // the byte four positions back
      x86 += c == 0x8b || ((old & 0xfe) == 0xe8 && (c == 0 || c == 255));
      w4 = w4 << 8 | c;
    }
  }
  stats[4 * f + 0] = hits;
  stats[4 * f + 1] = (L >= 16 && text * 16 >= L * 15 && ctrl == 0) ? 1u : 0u;
  stats[4 * f + 2] = (L >= 48 && x86 * 48 >= L) ? 1u : 0u;
  stats[4 * f + 3] = L;
}
}  // namespace

extern "C" int zpq_fragment_stats_dev(zpq_ctx* ctx, const uint8_t* d_base, const uint64_t* d_off, const uint32_t* d_len, size_t n, uint32_t* d_stats) {
  if (!ctx) return ZPQ_ERR_ARG;
  (void)hipSetDevice(ctx->device);
  if (n == 0) return ZPQ_OK;
  if (n > 0xffffffffu) return zpq_fail(ctx, ZPQ_ERR_ARG, "too many fragments");
  ZPQ_LAUNCH(ctx, "fragment_stats_kernel", ctx->stream, fragment_stats_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), d_base, d_off, d_len, (u32)n, d_stats);
  ZPQ_HIP(ctx, hipGetLastError());
  return ZPQ_OK;
}
