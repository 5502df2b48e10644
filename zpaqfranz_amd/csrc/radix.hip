// LSD radix sort of (u64 key, u32 value) pairs, stable, 8 bits per pass -- hand-written, for the two places the engine
// sorts: the suffix-array construction of -m2 (lz77_sa.hip, keys of up to 64 bits) and the candidate tables of the LZ77
// hash-table parse (lz77_enc.hip, 48-bit keys).  EXPERIMENTAL (end of round 3): selected with ZPQ_SORT=own, rocPRIM stays the
// default until this has been run and timed on the hardware; what could be checked without a GPU was the kernel source
// itself -- tests/cpp/radix_emu.cpp compiles it for the host and runs it on emulated workgroups (tests/test_radix_emu_cpu.py).
//
// A pass over digit d = (key >> shift) & 255, tiles of 4096 pairs (256 threads x 16):
//   rs_hist_kernel     per tile: digit histogram in LDS -> counts[digit][tile]
//   rs_totals_kernel   per digit: sum of its row; rs_scan_kernel: exclusive scan of every row, offset by the digits below
//   rs_scatter_kernel  per tile: every pair's rank among the tile's pairs with the same digit, in input order (stability):
//                      a wave owns a contiguous quarter of the tile and takes it 64 pairs at a time; lanes with equal
//                      digits find each other with eight ballots, their rank is "pairs of my digit this wave has seen"
//                      (a per-wave LDS counter, advanced by the first lane of each set) + "equal lanes below me"; the
//                      waves' counters are then prefixed per digit; the tile is laid out sorted by digit in LDS (48 KiB)
//                      and written from there, so neighbouring threads write neighbouring pairs of a digit's run to
//                      counts[digit][tile] + (place in the run).
// Memory per pass: keys read twice, pairs written once, in runs.
#if !defined(ZPQ_EMU_RADIX_ONLY) && !defined(ZPQ_EMU_FULL)      // (the host emulations of tests/cpp bring their own vocabulary)
#include <algorithm>
#include <stdlib.h>

#include "zpq_internal.h"
#endif

namespace {

constexpr u32 kRsThreads = 256, kRsItems = 16, kRsTile = kRsThreads * kRsItems;

__global__ __launch_bounds__(256) void rs_hist_kernel(const u64* __restrict__ keys, u64 n, u32 shift, u32 dmask, u32* __restrict__ counts, u32 ntiles) {
  __shared__ u32 h[256];
  const u32 t = threadIdx.x;
  h[t] = 0;
  __syncthreads();
  const u64 base = (u64)blockIdx.x * kRsTile;
#pragma unroll
  for (u32 i = 0; i < kRsItems; ++i) {
    const u64 idx = base + (u64)i * kRsThreads + t;
    if (idx < n) atomicAdd(&h[(u32)(keys[idx] >> shift) & dmask], 1u);
  }
  __syncthreads();
  counts[(size_t)t * ntiles + blockIdx.x] = h[t];
}

// block-wide exclusive prefix of one u32 per thread (256 threads); returns the prefix, *total the sum
__device__ __forceinline__ u32 rs_block_exclusive(u32 x, u32* total) {
  __shared__ u32 wsum[4];
  const u32 t = threadIdx.x, lane = t & 63u, w = t >> 6;
  u32 s = x;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const u32 y = __shfl_up(s, d); if (lane >= (u32)d) s += y; }
  if (lane == 63) wsum[w] = s;
  __syncthreads();
  u32 below = 0, all = 0;
#pragma unroll
  for (u32 k = 0; k < 4; ++k) { const u32 v = wsum[k]; if (k < w) below += v; all += v; }
  __syncthreads();                          // wsum may be reused by the next call
  *total = all;
  return below + s - x;
}

__global__ __launch_bounds__(256) void rs_totals_kernel(const u32* __restrict__ counts, u32 ntiles, u32* __restrict__ totals) {
  const u32* row = counts + (size_t)blockIdx.x * ntiles;
  u32 s = 0;
  for (u32 i = threadIdx.x; i < ntiles; i += kRsThreads) s += row[i];
  u32 all;
  (void)rs_block_exclusive(s, &all);
  if (threadIdx.x == 0) totals[blockIdx.x] = all;
}

__global__ __launch_bounds__(256) void rs_scan_kernel(u32* __restrict__ counts, u32 ntiles, const u32* __restrict__ totals) {
  // pairs with a smaller digit come first
  u32 all;
  const u32 below_me = rs_block_exclusive(totals[threadIdx.x], &all);
  __shared__ u32 dbase;
  if (threadIdx.x == blockIdx.x) dbase = below_me;
  __syncthreads();
  u32* row = counts + (size_t)blockIdx.x * ntiles;
  // every thread a contiguous piece of the row
  const u32 per = (ntiles + kRsThreads - 1) / kRsThreads;
  const u32 lo = threadIdx.x * per < ntiles ? threadIdx.x * per : ntiles;
  const u32 hi = lo + per < ntiles ? lo + per : ntiles;
  u32 s = 0;
  for (u32 i = lo; i < hi; ++i) s += row[i];
  u32 run = dbase + rs_block_exclusive(s, &all);
  for (u32 i = lo; i < hi; ++i) { const u32 c = row[i]; row[i] = run; run += c; }
}

__global__ __launch_bounds__(256) void rs_scatter_kernel(const u64* __restrict__ keys, const u32* __restrict__ vals, u64 n, u32 shift, u32 dmask,
                                                         const u32* __restrict__ counts, u32 ntiles, u64* __restrict__ keys_out,
                                                         u32* __restrict__ vals_out) {
  __shared__ u32 wcnt[4][256];     // pairs of each digit every wave holds (running while the wave walks its quarter)
  __shared__ u32 gbase[256];       // where this tile's pairs of each digit start in the output
  const u32 t = threadIdx.x, lane = t & 63u, w = t >> 6;
#pragma unroll
  for (u32 k = 0; k < 4; ++k) wcnt[k][t] = 0;
  gbase[t] = counts[(size_t)t * ntiles + blockIdx.x];
  __syncthreads();
  const u64 base = (u64)blockIdx.x * kRsTile + (u64)w * (64u * kRsItems);
  const unsigned long long below = (1ull << lane) - 1ull;
  u64 k[kRsItems];
  u32 v[kRsItems], rk[kRsItems];
#pragma unroll
  for (u32 r = 0; r < kRsItems; ++r) {
    const u64 idx = base + (u64)r * 64u + lane;
    const bool valid = idx < n;
    k[r] = valid ? keys[idx] : 0;
    v[r] = valid ? vals[idx] : 0;
    const u32 d = (u32)(k[r] >> shift) & dmask;
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (u32 b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(valid && bit);
      m &= bit ? bal : ~bal;
    }
    const u32 rin = (u32)__builtin_popcountll(m & below);
    const u32 pre = valid ? wcnt[w][d] : 0u;
    __builtin_amdgcn_wave_barrier();
    if (valid && rin == 0) wcnt[w][d] = pre + (u32)__builtin_popcountll(m);
    __builtin_amdgcn_wave_barrier();
    rk[r] = pre + rin;
  }
  __syncthreads();
  // pairs of digit t in the waves before each wave, and in the whole tile
  u32 mine = 0;
#pragma unroll
  for (u32 q = 0; q < 4; ++q) { const u32 c = wcnt[q][t]; wcnt[q][t] = mine; mine += c; }
  u32 all;
  const u32 tstart = rs_block_exclusive(mine, &all);     // where digit t starts in the tile once it is sorted by digit
  __shared__ u32 dstart[256];
  dstart[t] = tstart;
  __syncthreads();
  // the tile sorted by digit in LDS (ranks keep the input order inside a digit) ...
  __shared__ u64 sk[kRsTile];
  __shared__ u32 sv[kRsTile];
#pragma unroll
  for (u32 r = 0; r < kRsItems; ++r) {
    const u64 idx = base + (u64)r * 64u + lane;
    if (idx < n) {
      const u32 d = (u32)(k[r] >> shift) & dmask;
      const u32 at = dstart[d] + wcnt[w][d] + rk[r];
      sk[at] = k[r]; sv[at] = v[r];
    }
  }
  __syncthreads();
  // ... goes out in that order: neighbouring threads write neighbouring pairs of a digit's run
  const u32 count = all;                                 // pairs in this tile
#pragma unroll
  for (u32 i = 0; i < kRsItems; ++i) {
    const u32 j = i * kRsThreads + t;
    if (j < count) {
      const u64 key = sk[j];
      const u32 d = (u32)(key >> shift) & dmask;
      const u64 dst = (u64)gbase[d] + (j - dstart[d]);
      keys_out[dst] = key;
      vals_out[dst] = sv[j];
    }
  }
}

}  // namespace

#ifndef ZPQ_EMU_RADIX_ONLY
// Stable sort of n pairs by key bits [begin_bit, end_bit).  keys_in / vals_in are used as the second buffer (clobbered); the
// result is in keys_out / vals_out.  scratch: zpq_radix_scratch_words(n) u32.  n < 2^32 (counts and ranks are 32-bit).
size_t zpq_radix_scratch_words(size_t n) { return ((n + kRsTile - 1) / kRsTile) * 256 + 256 + 64; }

int zpq_radix_sort_pairs(zpq_ctx* ctx, hipStream_t st, u64* keys_in, u64* keys_out, u32* vals_in, u32* vals_out, size_t n, u32 begin_bit, u32 end_bit,
                         u32* scratch) {
  if (n >= 0xffffffffull) return zpq_fail(ctx, ZPQ_ERR_ARG, "radix sort: too many pairs");
  if (end_bit > 64 || begin_bit > end_bit) return zpq_fail(ctx, ZPQ_ERR_ARG, "radix sort: bad bit range");
  u64* ka = keys_in; u64* kb = keys_out; u32* va = vals_in; u32* vb = vals_out;
  if (n) {
    const u32 ntiles = (u32)((n + kRsTile - 1) / kRsTile);
    u32* counts = scratch;
    u32* totals = scratch + (size_t)ntiles * 256;
    for (u32 shift = begin_bit; shift < end_bit; shift += 8) {
      const u32 bits = std::min<u32>(8, end_bit - shift), dmask = (1u << bits) - 1u;
      ZPQ_LAUNCH(ctx, "rs_hist_kernel", st, rs_hist_kernel, dim3(ntiles), dim3(256), (const u64*)ka, (u64)n, shift, dmask, counts, ntiles);
      ZPQ_LAUNCH(ctx, "rs_totals_kernel", st, rs_totals_kernel, dim3(256), dim3(256), (const u32*)counts, ntiles, totals);
      ZPQ_LAUNCH(ctx, "rs_scan_kernel", st, rs_scan_kernel, dim3(256), dim3(256), counts, ntiles, (const u32*)totals);
      ZPQ_LAUNCH(ctx, "rs_scatter_kernel", st, rs_scatter_kernel, dim3(ntiles), dim3(256), (const u64*)ka, (const u32*)va, (u64)n, shift, dmask,
                 (const u32*)counts, ntiles, kb, vb);
      ZPQ_HIP(ctx, hipGetLastError());
      std::swap(ka, kb); std::swap(va, vb);
    }
  }
  if (ka != keys_out && n) {            // an even number of passes (or none) left the result in the first buffer
    ZPQ_HIP(ctx, hipMemcpyAsync(keys_out, ka, n * 8, hipMemcpyDeviceToDevice, st));
    ZPQ_HIP(ctx, hipMemcpyAsync(vals_out, va, n * 4, hipMemcpyDeviceToDevice, st));
  }
  return ZPQ_OK;
}
#endif  // ZPQ_EMU_RADIX_ONLY
