// Test infrastructure: zpaqfranz_amd/csrc/lz77_cand.inc -- the device source of the candidate-table kernels -- compiled for
// the HOST.  The macros the include is written against are bound to a serial SIMT shim (one "thread" after the other),
// std::stable_sort stands in for the radix sort; cand_host() then runs keys -> sort -> sweep exactly as cand_build() in
// lz77_enc.hip launches them.  tests/test_lz_cand_cpu.py compares the table with the oracle's.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <vector>

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

namespace {
struct Dim { u32 x, y; };
thread_local Dim g_tid, g_bid, g_gdim;
}
#define ZPQ_CAND_KERNEL(bounds) static
#define ZPQ_CAND_DEV static inline
#define ZPQ_CAND_TID (g_tid.x)
#define ZPQ_CAND_BID_X (g_bid.x)
#define ZPQ_CAND_BID_Y (g_bid.y)
#define ZPQ_CAND_GDIM_X (g_gdim.x)
#define ZPQ_CAND_GLOBAL
#define ZPQ_CAND_ATOMIC_INC(p) ((*(p))++)
#define __restrict__
#include "lz77_cand.inc"

template <class F>
static void launch(u32 gx, u32 gy, u32 threads, F&& body) {
  g_gdim = {gx, gy};
  for (u32 by = 0; by < gy; ++by)
    for (u32 bx = 0; bx < gx; ++bx)
      for (u32 t = 0; t < threads; ++t) { g_bid = {bx, by}; g_tid = {t, 0}; body(); }
}

// blocks: nblocks inputs back to back in `in` at in_off[b] (n[b] bytes each, 8 readable bytes behind the last one);
// cand: n[b] << args[4] words per block, back to back.  Returns 0.
extern "C" int cand_host(const u8* in, const u64* in_off, const u32* n, u32 nblocks, const int32_t args[9], u32* cand) {
  std::vector<CandJob> jobs(nblocks);
  u64 pos0 = 0, cw = 0;
  u32 max_n = 0;
  for (u32 b = 0; b < nblocks; ++b) {
    LzCfg& c = jobs[b].c;
    c.in = in + in_off[b]; c.n = n[b]; c.minMatch = args[2]; c.bucket = (1u << args[4]) - 1; c.htbits = args[5]; c.checkbits = 12 - args[0];
    c.shift1 = (args[5] - 1) / args[2] + 1; c.rb = args[0] > 4 ? args[0] - 4 : 0;
    const u32 mmb = args[2] + 4;
    c.upd_limit = n[b] > mmb ? n[b] - mmb : 0;
    jobs[b].pos0 = pos0; jobs[b].cand = cand + cw; jobs[b].lb = (u32)args[4]; jobs[b].pad = 0;
    pos0 += n[b]; cw += (u64)n[b] << args[4];
    max_n = std::max(max_n, n[b]);
  }
  const u64 total = pos0;
  if (!total) return 0;
  std::vector<u64> k0(total), k1(total);
  std::vector<u32> v0(total), v1(total);
  launch(std::min<u32>((max_n + 255) / 256, 7), nblocks, 256, [&] { lz77_cand_keys_kernel(jobs.data(), k0.data(), v0.data()); });   // a small grid: the kernel strides
  std::vector<u64> idx(total);
  std::iota(idx.begin(), idx.end(), 0);
  std::stable_sort(idx.begin(), idx.end(), [&](u64 a, u64 b) { return k0[a] < k0[b]; });
  for (u64 i = 0; i < total; ++i) { k1[i] = k0[idx[i]]; v1[i] = v0[idx[i]]; }
  const u32 grid = (u32)((total + 63) / 64);
  switch (args[4]) {
    case 0: launch(grid, 1, 64, [&] { lz77_cand_sweep_kernel<1>(jobs.data(), k1.data(), v1.data(), total, nullptr, 0, 0); }); break;
    case 1: launch(grid, 1, 64, [&] { lz77_cand_sweep_kernel<2>(jobs.data(), k1.data(), v1.data(), total, nullptr, 0, 0); }); break;
    case 2: launch(grid, 1, 64, [&] { lz77_cand_sweep_kernel<4>(jobs.data(), k1.data(), v1.data(), total, nullptr, 0, 0); }); break;
    default: launch(grid, 1, 64, [&] { lz77_cand_sweep_kernel<8>(jobs.data(), k1.data(), v1.data(), total, nullptr, 0, 0); }); break;
  }
  return 0;
}
