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
  if (n == 0) return ZPQ_OK;
  if (n > 0x7fffffffu) return zpq_fail(ctx, ZPQ_ERR_ARG, "too many fragments");
  u32 size = 1024;
  while (size < 2 * n) size <<= 1;
  u32* table = (u32*)zpq_scratch(ctx, 5, (size_t)size * 4);
  if (!table) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "dedup table");
  ZPQ_HIP(ctx, hipMemsetAsync(table, 0xff, (size_t)size * 4, ctx->stream));
  const unsigned grid = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(dedup_insert_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_digests, (u32)n, table, size - 1);
  ZPQ_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(dedup_lookup_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_digests, (u32)n, table, size - 1,
                     d_first);
  ZPQ_HIP(ctx, hipGetLastError());
  return ZPQ_OK;
}
