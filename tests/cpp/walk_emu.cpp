// Test infrastructure: the LZ77 hash-table parse kernels of zpaqfranz_amd/csrc/lz77_enc.hip compiled for the HOST and run as
// emulated waves (simt_emu.h: 64 lanes = 64 fibres of one thread, every wave-level operation a rendezvous).  What lockstep
// gives for free on the GPU -- every lane's stores of a window are issued before any lane's loads of the next -- is a
// rendezvous here too (ZPQ_WAIT_VMCNT0).  Built with the ROCm clang++ as a host compiler (address spaces, ext vectors).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <algorithm>
#include <vector>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_u;
typedef u64 __attribute__((aligned(1))) u64_u;
typedef u32 __attribute__((aligned(1))) u32_u;

#include "simt_emu.h"

// ---- what lz77_enc.hip expects from the device environment ------------------------------------------------------------
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
struct EmuDim { u32 x, y, z; };
static EmuDim threadIdx, blockIdx, gridDim;
static bool g_in_wave = false;
static inline int lane_id() { return g_in_wave ? emu::lane() : (int)(threadIdx.x & 63u); }
static inline unsigned long long emu_ballot(bool p, int op) { return emu::ballot(p, op); }
template <class T> static inline T emu_shfl(T v, int src, int op) { return emu::shfl(v, src, op); }
#define __ballot(p) emu_ballot((p), __LINE__)
#define __shfl(v, src) emu_shfl((v), (int)(src), __LINE__)
#define __shfl_xor(v, m) emu_shfl((v), lane_id() ^ (int)(m), __LINE__)
#define __shfl_up(v, d) emu_shfl((v), lane_id() >= (int)(d) ? lane_id() - (int)(d) : lane_id(), __LINE__)
#define __builtin_amdgcn_readlane(v, l) emu_shfl((v), (int)(l), __LINE__)
#define __builtin_amdgcn_readfirstlane(v) emu_shfl((v), 0, __LINE__)
#define __builtin_amdgcn_wave_barrier() ((void)emu::wave_rendezvous(0, __LINE__))
#define __builtin_amdgcn_s_setprio(x) ((void)0)

#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or((p), (v), __ATOMIC_RELAXED)
#define __hip_atomic_fetch_max(p, v, order, scope) __atomic_fetch_max((p), (v), __ATOMIC_RELAXED)
#define __ATOMIC_RELAXED_HIP 0
#define __HIP_MEMORY_SCOPE_WAVEFRONT 0
#define __syncthreads() emu::block_barrier(__LINE__)
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_RELAXED)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_RELAXED)
#define __builtin_amdgcn_s_sleep(x) ((void)(emu::g_active ? (emu::to_main(), 0) : 0))      /* a sleeping wave lets the other wave of the workgroup run */
#define LZ_DUO_TID() ((u32)emu::g_tid)                 /* lz77_waves.inc: two-wave workgroups (threadIdx is not per fibre in this harness) */
#define LZ_DUO_WAVE_ID() ((u32)emu::wave())
#define ZPQ_WAIT_VMCNT0 ((void)emu::wave_rendezvous(0, __LINE__))      /* lockstep: every lane's stores before anybody's next loads */
// serial stand-ins for the global atomics of the thread-independent kernels (one thread runs after the other)
template <class T> static inline T emu_atomic_max(T* p, T v) { const T o = *p; if (v > o) *p = v; return o; }
#define atomicMax(p, v) emu_atomic_max((p), (v))
template <class T> static inline T emu_atomic_add(T* p, T v) { const T o = *p; *p = o + v; return o; }
#define atomicAdd(p, v) emu_atomic_add((p), (v))
// zpq_internal.h: the plain-launch half of the cooperative placement helpers (tab == nullptr)
struct zpq_place { u32* queue; u32* tab; u32 n; u32 polite; };
static inline u32 zpq_place_begin(const zpq_place& P, u32& key, bool& polite) { key = 0; polite = true; return blockIdx.x < P.n ? blockIdx.x : 0xffffffffu; }
static inline u32 zpq_place_next(const zpq_place&, u32, bool) { return 0xffffffffu; }
struct zpq_lzjob_dev {                       // (zpq_internal.h; only named by a typedef in front of lz_walk)
  const u8* in; u32 n; u32 rb; u32 nseg, seg0; u32* tok_pos; u32* tok_len; u32* tok_off; u32* tok_bit; u32 tok_cap; u32* result; u8* out; u32 out_cap; u32* plan;
};

#define ZPQ_EMU_WALK_ONLY
#include "lz77_enc.hip"

// ---- one wave walks a whole block (lz_walk: what the seam and stitch kernels call) ----------------------------------------
namespace {
struct WalkArgs { LzCfg c; u32* table; u32* tpos; u32* tlen; u32* toff; u32 tcap; u32 ntok; u32 end_cur, end_lit; };
WalkArgs g_w;
unsigned long long g_T[256];

template <int NB>
void walk_body() {
  TokSink sink{g_w.tpos, g_w.tlen, g_w.toff, g_w.tcap, 0};
  u32 cur = 0, lit = 0;
  lz_walk<NB>(g_w.c, g_w.table, 0, g_w.c.n, cur, lit, sink, nullptr, g_T);
  if (lane_id() == 0) { g_w.ntok = sink.n; g_w.end_cur = cur; g_w.end_lit = lit; }
}
LzCfg make_cfg(const u8* in, u32 n, const int32_t args[9]) {
  LzCfg c;
  c.in = in; c.n = n; c.minMatch = args[2]; c.bucket = (1u << args[4]) - 1; c.htbits = args[5]; c.checkbits = 12 - args[0];
  c.shift1 = (args[5] - 1) / args[2] + 1; c.rb = args[0] > 4 ? args[0] - 4 : 0; c.level = (u32)(args[1] & 3);
  const u32 mmb = args[2] + 4;
  c.upd_limit = n > mmb ? n - mmb : 0;
  return c;
}
}  // namespace

// in: n bytes + >= 64 readable bytes behind; table: zeroed hash table of 2^args[5] words, 16-byte aligned.
// tok: 3 * cap words (pos | len | off).  Returns the token count, or < 0.
extern "C" long walk_emu(const u8* in, u32 n, const int32_t args[9], u32* table, u32* tok, u32 cap, char* err, u32 err_cap) {
  g_w.c = make_cfg(in, n, args);
  g_w.table = table; g_w.tpos = tok; g_w.tlen = tok + cap; g_w.toff = tok + 2 * (size_t)cap; g_w.tcap = cap; g_w.ntok = 0;
  memset(g_T, 0, sizeof g_T);
  void (*body)() = nullptr;
  switch (args[4]) {
    case 0: body = walk_body<1>; break;
    case 1: body = walk_body<2>; break;
    case 2: body = walk_body<4>; break;
    default: body = walk_body<8>; break;
  }
  g_in_wave = true;
  const char* e = emu::run_block(body);
  g_in_wave = false;
  if (e) { if (err && err_cap) { strncpy(err, e, err_cap - 1); err[err_cap - 1] = 0; } return -1; }
  return (long)g_w.ntok;
}

// ---- the segment speculation of one block, as encode_batch() lays it out -----------------------------------------------------
// table states (copy + scatter kernels), then lz77_spec3_kernel per segment (a workgroup of three waves: producer | evaluator |
// chain, lz77_waves.inc), lz77_seam_kernel per segment, lz77_stitch_kernel, lz77_move_tokens_kernel: the kernels themselves,
// the wave kernels as emulated waves, the thread-independent ones thread by thread.
namespace {
struct SpecRun { const LzSegDev* segs; const u32* list; const LzJobDev* jobs; u32 nseg; };
SpecRun g_s;
template <int NB> void spec_body() { lz77_spec3_kernel<NB>(g_s.segs, g_s.list); }
template <int NB> void seam_body() { lz77_seam_kernel<NB>(g_s.segs, g_s.list); }
template <int NB> void stitch_body() { lz77_stitch_kernel<NB>(g_s.jobs, g_s.segs, g_s.list); }
typedef void (*Body)();
template <int NB> void bodies(Body& sp, Body& se, Body& st) { sp = spec_body<NB>; se = seam_body<NB>; st = stitch_body<NB>; }
const char* wave(Body b, u32 bx, int threads = 64) { blockIdx = {bx, 0, 0}; g_in_wave = true; const char* e = emu::run_block(b, threads); g_in_wave = false; return e; }
template <class F> void serial(u32 gx, u32 gy, u32 threads, F&& f) {
  gridDim = {gx, gy, 1};
  for (u32 by = 0; by < gy; ++by) for (u32 bx = 0; bx < gx; ++bx) for (u32 t = 0; t < threads; ++t) { blockIdx = {bx, by, 0}; threadIdx = {t, 0, 0}; f(); }
}
}  // namespace

// tok: 3 * cap words (pos | len | off) of the final list.  Returns the token count, < 0 on an emulation error, -2 on overflow.
extern "C" long spec_emu(const u8* in, u32 n, const int32_t args[9], u32 seg_bytes, u32* tok, u32 cap, char* err, u32 err_cap) {
  const LzCfg c = make_cfg(in, n, args);
  const u32 nseg = std::max<u32>(1, (u32)(((u64)n + seg_bytes - 1) / seg_bytes));
  const size_t words = (size_t)1 << args[5];
  const u32 mmt = (u32)(args[2] >= 4 ? args[2] : 4);
  std::vector<u32> tables(words * (2 * (size_t)nseg - 1) + 16, 0);                      // work[0..nseg-1], pristine[1..nseg-1]
  u32* tab0 = (u32*)(((uintptr_t)tables.data() + 15) & ~(uintptr_t)15);
  u32* prist0 = tab0 + words * nseg - words;
  const u32 fcap = n / mmt + 3;
  std::vector<u32> ftok((size_t)fcap * 4, 0), state((size_t)nseg * 12, 0), plan((size_t)nseg * 8, 0), result(4, 0);
  std::vector<std::vector<u32>> lists(nseg * 2);
  std::vector<LzSegDev> segs(nseg);
  LzJobDev J;
  J.in = in; J.n = n; J.rb = c.rb; J.nseg = nseg; J.seg0 = 0;
  J.tok_pos = ftok.data(); J.tok_len = J.tok_pos + fcap; J.tok_off = J.tok_len + fcap; J.tok_bit = J.tok_off + fcap; J.tok_cap = fcap - 1;
  J.result = result.data(); J.out = nullptr; J.out_cap = 0; J.plan = plan.data();
  for (u32 k = 0; k < nseg; ++k) {
    LzSegDev& S = segs[k];
    S.c = c; S.x0 = k * seg_bytes; S.x1 = (u32)std::min<u64>((u64)S.x0 + seg_bytes, n);
    S.work = tab0 + words * k;
    S.pristine = k ? prist0 + words * k : nullptr;
    const u32 scap = (S.x1 - S.x0) / mmt + 3;
    lists[2 * k].assign((size_t)scap * 3, 0);
    S.tpos = lists[2 * k].data(); S.tlen = S.tpos + scap; S.toff = S.tlen + scap; S.tcap = scap - 1;
    if (k) { lists[2 * k + 1].assign((size_t)scap * 3, 0); S.qpos = lists[2 * k + 1].data(); S.qlen = S.qpos + scap; S.qoff = S.qlen + scap; }
    else { S.qpos = S.tpos; S.qlen = S.tlen; S.qoff = S.toff; }
    S.state = state.data() + 12 * (size_t)k; S.seam = S.state + 4;
  }
  for (u32 k = 1; k < nseg; ++k) {      // table states: pristine[k] = pristine[k-1] + inserts of segment k-1; work[k] = pristine[k]
    CopyJob cj{k == 1 ? nullptr : prist0 + words * (k - 1), segs[k].pristine, (u32)words};
    serial(8, 1, 256, [&] { lz77_table_copy_kernel(&cj); });
    ScatterJob sj{c, (k - 1) * seg_bytes, k * seg_bytes, segs[k].pristine};
    serial(8, 1, 256, [&] { lz77_table_scatter_kernel(&sj); });
    CopyJob cw{segs[k].pristine, segs[k].work, (u32)words};
    serial(8, 1, 256, [&] { lz77_table_copy_kernel(&cw); });
  }
  std::vector<u32> seglist(nseg), joblist(1, 0);
  for (u32 k = 0; k < nseg; ++k) seglist[k] = k;
  g_s = SpecRun{segs.data(), seglist.data(), &J, nseg};
  Body sp = nullptr, se = nullptr, st = nullptr;
  switch (args[4]) {
    case 0: bodies<1>(sp, se, st); break;
    case 1: bodies<2>(sp, se, st); break;
    case 2: bodies<4>(sp, se, st); break;
    default: bodies<8>(sp, se, st); break;
  }
  const char* e = nullptr;
  gridDim = {nseg, 1, 1};
  for (u32 k = 0; k < nseg && !e; ++k) e = wave(sp, k, 192);
  for (u32 k = 0; k < nseg && !e; ++k) e = wave(se, k);
  g_s.list = joblist.data();
  gridDim = {1, 1, 1};
  if (!e) e = wave(st, 0);
  if (e) { if (err && err_cap) { strncpy(err, e, err_cap - 1); err[err_cap - 1] = 0; } return -1; }
  std::vector<u32> segjob(nseg, 0);
  serial(4, nseg * 2, 256, [&] { lz77_move_tokens_kernel(&J, segs.data(), segjob.data()); });
  if (result[2]) return -2;
  const u32 nt = result[0];
  if (nt > cap) return -2;
  for (u32 i = 0; i < nt; ++i) { tok[i] = J.tok_pos[i]; tok[cap + i] = J.tok_len[i]; tok[2 * (size_t)cap + i] = J.tok_off[i]; }
  return (long)nt;
}

// ---- one workgroup per block, parsing and emitting in one go (lz77_direct4_kernel) -------------------------------------------
namespace {
template <int NB> void direct_body() { lz77_direct4_kernel<NB>(g_s.jobs, g_s.segs, g_s.list); }
}
// out: the code stream (out_cap bytes, zeroed by the caller).  Returns its length, < 0 on an emulation error, -2 on overflow.
extern "C" long direct_emu(const u8* in, u32 n, const int32_t args[9], u32* table, u8* out, u32 out_cap, char* err, u32 err_cap) {
  LzSegDev S;
  memset(&S, 0, sizeof S);
  S.c = make_cfg(in, n, args); S.x0 = 0; S.x1 = n; S.work = table; S.pristine = table;
  u32 result[4] = {0, 0, 0, 0};
  LzJobDev J;
  memset(&J, 0, sizeof J);
  J.in = in; J.n = n; J.rb = S.c.rb; J.nseg = 1; J.seg0 = 0; J.result = result; J.out = out; J.out_cap = out_cap;
  u32 list0 = 0;
  g_s = SpecRun{&S, &list0, &J, 1};
  Body b = nullptr;
  switch (args[4]) {
    case 0: b = direct_body<1>; break;
    case 1: b = direct_body<2>; break;
    case 2: b = direct_body<4>; break;
    default: b = direct_body<8>; break;
  }
  gridDim = {1, 1, 1};
  const char* e = wave(b, 0, 256);
  if (e) { if (err && err_cap) { strncpy(err, e, err_cap - 1); err[err_cap - 1] = 0; } return -1; }
  if (result[2]) return -2;
  return (long)result[1];
}
