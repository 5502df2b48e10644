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
