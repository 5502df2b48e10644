// Test infrastructure: the radix sort kernels of zpaqfranz_amd/csrc/radix.hip compiled for the HOST and run on emulated
// workgroups (simt_emu.h: 256 fibres per workgroup, wave operations and __syncthreads as rendezvous).  radix_emu() runs the
// passes exactly as zpq_radix_sort_pairs() launches them.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "simt_emu.h"

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(x)
#define __shared__ static
#define __restrict__
struct EmuIdx { u32 x, y, z; };
static EmuIdx blockIdx, gridDim;
static inline EmuIdx emu_thread_idx() { return EmuIdx{(u32)emu::g_tid, 0, 0}; }
#define threadIdx (emu_thread_idx())
#define __ballot(p) emu::ballot((p), __LINE__)
#define __shfl_up(v, d) emu::shfl((v), emu::lane() >= (int)(d) ? emu::lane() - (int)(d) : emu::lane(), __LINE__)
#define __builtin_amdgcn_wave_barrier() ((void)emu::wave_rendezvous(0, __LINE__))
#define __syncthreads() emu::block_barrier(__LINE__)
template <class T> static inline T emu_atomic_add(T* p, T v) { const T o = *p; *p = o + v; return o; }
#define atomicAdd(p, v) emu_atomic_add((p), (v))

#define ZPQ_EMU_RADIX_ONLY
#include "radix.hip"

namespace {
struct Args { const u64* ka; const u32* va; u64 n; u32 shift, dmask; u32* counts; u32 ntiles; u32* totals; u64* kb; u32* vb; };
Args A;
void hist_body() { rs_hist_kernel(A.ka, A.n, A.shift, A.dmask, A.counts, A.ntiles); }
void totals_body() { rs_totals_kernel(A.counts, A.ntiles, A.totals); }
void scan_body() { rs_scan_kernel(A.counts, A.ntiles, A.totals); }
void scatter_body() { rs_scatter_kernel(A.ka, A.va, A.n, A.shift, A.dmask, A.counts, A.ntiles, A.kb, A.vb); }
const char* launch(void (*body)(), u32 grid) {
  gridDim = {grid, 1, 1};
  for (u32 b = 0; b < grid; ++b) {
    blockIdx = {b, 0, 0};
    if (const char* e = emu::run_block(body, 256)) return e;
  }
  return nullptr;
}
}  // namespace

// keys / vals: n pairs, sorted in place (stable) by key bits [begin_bit, end_bit).  Returns 0, or -1 with a text in err.
extern "C" int radix_emu(u64* keys, u32* vals, u64 n, u32 begin_bit, u32 end_bit, char* err, u32 err_cap) {
  if (!n) return 0;
  std::vector<u64> k2(n);
  std::vector<u32> v2(n);
  const u32 ntiles = (u32)((n + kRsTile - 1) / kRsTile);
  std::vector<u32> counts((size_t)ntiles * 256 + 256, 0);
  u64* ka = keys; u64* kb = k2.data(); u32* va = vals; u32* vb = v2.data();
  for (u32 shift = begin_bit; shift < end_bit; shift += 8) {
    const u32 bits = std::min<u32>(8, end_bit - shift);
    A = Args{ka, va, n, shift, (1u << bits) - 1u, counts.data(), ntiles, counts.data() + (size_t)ntiles * 256, kb, vb};
    const char* e = launch(hist_body, ntiles);
    if (!e) e = launch(totals_body, 256);
    if (!e) e = launch(scan_body, 256);
    if (!e) e = launch(scatter_body, ntiles);
    if (e) { if (err && err_cap) { strncpy(err, e, err_cap - 1); err[err_cap - 1] = 0; } return -1; }
    std::swap(ka, kb); std::swap(va, vb);
  }
  if (ka != keys) { memcpy(keys, ka, n * 8); memcpy(vals, va, n * 4); }
  return 0;
}
