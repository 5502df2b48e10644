// Test infrastructure: the fragmenter kernels of zpaqfranz_amd/csrc/fragment.hip compiled for the HOST and run on the fibre
// emulator (simt_emu.h), launched as fragment_run() launches them: fragment_spec_kernel (persistent lanes, LDS o1[] tables),
// its RESUME form for the parked crossing walks, fragment_stitch_kernel (a wave per file, four per workgroup: the exact
// wave evaluator with its DPP scan lives here), fragment_emit_kernel -- with or without a representative table (twins).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "simt_emu.h"

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_u;
typedef u64 __attribute__((aligned(1))) u64_u;
typedef u32 __attribute__((aligned(1))) u32_u;

#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
struct EmuIdx { u32 x, y, z; };
static EmuIdx blockIdx, gridDim;
static inline EmuIdx emu_thread_idx() { return EmuIdx{(u32)emu::g_tid, 0, 0}; }
#define threadIdx (emu_thread_idx())
static inline int lane_id() { return emu::lane(); }
#define __ballot(p) emu::ballot((p), __LINE__)
#define __any(p) emu::any((p), __LINE__)
#define __all(p) emu::all((p), __LINE__)
#define __shfl(v, src) emu::shfl((v), (int)(src), __LINE__)
#define __builtin_amdgcn_readlane(v, l) emu::shfl((v), (int)(l), __LINE__)
#define __builtin_amdgcn_wave_barrier() ((void)emu::wave_rendezvous(0, __LINE__))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rows, banks, bc) emu::dpp((old), (src), (ctrl), (rows), __LINE__)
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or((p), (v), __ATOMIC_RELAXED)
#define __HIP_MEMORY_SCOPE_WAVEFRONT 0
template <class T, class V> static inline T emu_atomic_add(T* p, V v) { const T o = *p; *p = o + (T)v; return o; }
#define atomicAdd(p, v) emu_atomic_add((p), (v))

#define ZPQ_EMU_FRAGMENT_ONLY
#include "fragment.hip"

namespace {
struct Run {
  const u8* data; u64 readable; const u64* file_off; u32 nfiles; const u32* seg_file; const u64* seg_base; u64 nseg; FragP P; u32 spec_cap; u32* spec_rel;
  u32* spec_cnt; CrossOut* cross; unsigned long long* counters; Parked* parked; u64 budget; const u64* cut_base; u64* cuts; u32* cut_cnt; const u32* rep;
  const u64* frag_base; u64* frag_off; u32* frag_len; u32* frag_file;
};
Run R;
void spec_body() { fragment_spec_kernel<false>(R.data, R.readable, R.file_off, R.seg_file, R.seg_base, R.nseg, R.P, R.spec_cap, R.spec_rel, R.spec_cnt, R.cross, R.counters, R.parked, R.budget); }
void resume_body() { fragment_spec_kernel<true>(R.data, R.readable, R.file_off, R.seg_file, R.seg_base, R.nseg, R.P, R.spec_cap, R.spec_rel, R.spec_cnt, R.cross, R.counters, R.parked, R.budget); }
void stitch_body() { fragment_stitch_kernel(R.data, R.readable, R.file_off, R.nfiles, R.seg_base, R.P, R.spec_cap, R.spec_rel, R.spec_cnt, R.cross, R.cut_base, R.cuts, R.cut_cnt, R.rep); }
void emit_body() { fragment_emit_kernel(R.file_off, R.nfiles, R.cut_base, R.cuts, R.cut_cnt, R.frag_base, R.frag_off, R.frag_len, R.frag_file, R.rep); }
const char* launch(void (*body)(), u32 grid, int threads) {
  gridDim = {grid, 1, 1};
  for (u32 b = 0; b < grid; ++b) { blockIdx = {b, 0, 0}; if (const char* e = emu::run_block(body, threads)) return e; }
  return nullptr;
}
}  // namespace

// data: all files back to back (file_off[nfiles+1]) with >= 64 readable bytes behind; rep: null or the twin table
// (rep[f] = earliest equal file); out_*: room for cap records.  waves: spec waves to launch (few: lanes then pull several
// segments).  Returns the number of fragments, -1 on an emulation error, -2 when cap is too small.
extern "C" long frag_emu(const u8* data, const u64* file_off, u32 nfiles, u32 log2frag, u32 minf, u32 maxf, u64 seg, u32 waves, u64 budget, const u32* rep,
                         u64* out_off, u32* out_len, u32* out_file, u64 cap, char* err, u32 err_cap) {
  FragP P;
  P.minf = minf; P.maxf = maxf; P.thresh = log2frag <= 22 ? 1u << (22 - log2frag) : 0u; P.pad = 0; P.seg = seg;
  const u64 all_bytes = file_off[nfiles];
  std::vector<u64> seg_base(nfiles + 1), cut_base(nfiles + 1);
  u64 nseg = 0, ncut = 0;
  for (u32 f = 0; f < nfiles; ++f) {
    const bool walked = !rep || rep[f] == f;
    const u64 len = walked ? file_off[f + 1] - file_off[f] : 0;
    seg_base[f] = nseg; cut_base[f] = ncut;
    nseg += (len + seg - 1) / seg;
    ncut += walked ? len / P.minf + 1 : 0;
  }
  seg_base[nfiles] = nseg; cut_base[nfiles] = ncut;
  if (!nseg) return 0;
  std::vector<u32> seg_file(nseg);
  for (u32 f = 0; f < nfiles; ++f) for (u64 s = seg_base[f]; s < seg_base[f + 1]; ++s) seg_file[s] = f;
  const u32 spec_cap = (u32)(seg / P.minf + 2);
  std::vector<u32> spec_rel(nseg * spec_cap + 2, 0), spec_cnt(nseg, 0), cut_cnt(nfiles, 0);
  std::vector<CrossOut> cross(nseg);
  std::vector<Parked> parked(nseg);
  std::vector<u64> cuts(ncut + 1, 0);
  unsigned long long counters[3] = {0, 0, 0};
  memset(cross.data(), 0, nseg * sizeof(CrossOut));
  R = Run{data, all_bytes, file_off, nfiles, seg_file.data(), seg_base.data(), nseg, P, spec_cap, spec_rel.data(), spec_cnt.data(), cross.data(), counters,
          parked.data(), budget, cut_base.data(), cuts.data(), cut_cnt.data(), rep, nullptr, out_off, out_len, out_file};
  const u32 want = (u32)((nseg + 63) / 64);
  const char* e = launch(spec_body, std::min(want, std::max(1u, waves)), 64);
  if (!e) e = launch(resume_body, std::min(want, 2u), 64);
  R.readable = (all_bytes + 3) & ~3ull;
  if (!e) e = launch(stitch_body, (nfiles + 3) / 4, 256);
  if (e) { if (err && err_cap) { strncpy(err, e, err_cap - 1); err[err_cap - 1] = 0; } return -1; }
  std::vector<u64> frag_base(nfiles + 1);
  u64 nf = 0;
  for (u32 f = 0; f < nfiles; ++f) { frag_base[f] = nf; nf += cut_cnt[rep ? rep[f] : f]; }
  frag_base[nfiles] = nf;
  if (nf > cap) return -2;
  R.frag_base = frag_base.data();
  e = launch(emit_body, (nfiles + 3) / 4, 256);
  if (e) { if (err && err_cap) { strncpy(err, e, err_cap - 1); err[err_cap - 1] = 0; } return -1; }
  return (long)nf;
}
