// Test infrastructure: lz_walk -- the heart of the LZ77 hash-table parse, zpaqfranz_amd/csrc/lz77_enc.hip -- compiled for
// the HOST and run as one emulated wave: 64 lanes are 64 fibres (ucontext) of one thread that take turns; every wave-level
// operation (ballot, shuffle, readlane, wave barrier) is a rendezvous at which each lane deposits its value, yields, and
// reads everybody's when its turn comes again.  Lanes of a real wave run in lockstep, so all of them reach the same
// rendezvous in the same order; a lane that does not is reported (operation ids are compared).  What lockstep gives for
// free on the GPU -- every lane's stores of a window are issued before any lane's loads of the next -- is a rendezvous
// here too (ZPQ_WAIT_VMCNT0).  Built with the ROCm clang++ as a host compiler (address spaces, ext vectors).
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

// ---- the emulated wave ---------------------------------------------------------------------------------------------
namespace emu {
constexpr int W = 64;
ucontext_t g_main, g_ctx[W];
int g_lane = 0;
bool g_done[W];
u64 g_buf[2][W];
int g_op[2][W];
u32 g_phase[W];
const char* g_error = nullptr;
void (*g_body)() = nullptr;

inline void yield() { swapcontext(&g_ctx[g_lane], &g_main); }
// deposits v, lets every other lane reach the same rendezvous, returns all 64 values
inline const u64* rendezvous(u64 v, int op) {
  const int me = g_lane, p = (int)(g_phase[me]++ & 1u);
  g_buf[p][me] = v; g_op[p][me] = op;
  yield();
  for (int i = 0; i < W; ++i)
    if (g_op[p][i] != op && !g_error) g_error = "lanes reached different wave operations (divergent intrinsic)";
  return g_buf[p];
}
void trampoline() { g_body(); g_done[g_lane] = true; swapcontext(&g_ctx[g_lane], &g_main); }
// runs body() on 64 lanes in lockstep; returns nullptr or an error text
const char* run_wave(void (*body)()) {
  static std::vector<char> stacks((size_t)W * (256 << 10));
  g_body = body; g_error = nullptr;
  for (int i = 0; i < W; ++i) {
    g_done[i] = false; g_phase[i] = 0;
    getcontext(&g_ctx[i]);
    g_ctx[i].uc_stack.ss_sp = stacks.data() + (size_t)i * (256 << 10);
    g_ctx[i].uc_stack.ss_size = 256 << 10;
    g_ctx[i].uc_link = &g_main;
    makecontext(&g_ctx[i], trampoline, 0);
  }
  for (;;) {
    bool any = false;
    for (int i = 0; i < W; ++i)
      if (!g_done[i]) { any = true; g_lane = i; swapcontext(&g_main, &g_ctx[i]); }
    if (!any) break;
    bool all = true, none = true;
    for (int i = 0; i < W; ++i) { all = all && g_done[i]; none = none && !g_done[i]; }
    if (!all && !none && !g_error) g_error = "some lanes left the kernel while others wait at a wave operation";
    if (g_error && !all) return g_error;      // (the fibres are abandoned)
  }
  return g_error;
}
}  // namespace emu

// ---- what lz77_enc.hip expects from the device environment ------------------------------------------------------------
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(x)
#define __shared__ static
static inline int lane_id() { return emu::g_lane; }
struct EmuDim { u32 x, y, z; };
static EmuDim threadIdx, blockIdx, gridDim;
static inline unsigned long long emu_ballot(bool p, int op) {
  const u64* a = emu::rendezvous(p ? 1 : 0, op);
  unsigned long long m = 0;
  for (int i = 0; i < 64; ++i) m |= (unsigned long long)(a[i] & 1) << i;
  return m;
}
template <class T> static inline T emu_shfl(T v, int src, int op) {
  u64 bits = 0; memcpy(&bits, &v, sizeof v);
  const u64* a = emu::rendezvous(bits, op);
  T r; memcpy(&r, &a[src & 63], sizeof r);
  return r;
}
#define __ballot(p) emu_ballot((p), __LINE__)
#define __shfl(v, src) emu_shfl((v), (int)(src), __LINE__)
#define __shfl_xor(v, m) emu_shfl((v), lane_id() ^ (int)(m), __LINE__)
#define __builtin_amdgcn_readlane(v, l) emu_shfl((v), (int)(l), __LINE__)
#define __builtin_amdgcn_readfirstlane(v) emu_shfl((v), 0, __LINE__)
#define __builtin_amdgcn_wave_barrier() ((void)emu::rendezvous(0, __LINE__))
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or((p), (v), __ATOMIC_RELAXED)
#define __ATOMIC_RELAXED_HIP 0
#define __HIP_MEMORY_SCOPE_WAVEFRONT 0
#define ZPQ_WAIT_VMCNT0 ((void)emu::rendezvous(0, __LINE__))      /* lockstep: every lane's stores before anybody's next loads */
struct zpq_lzjob_dev {                       // (zpq_internal.h; only named by a typedef in front of lz_walk)
  const u8* in; u32 n; u32 rb; u32 nseg, seg0; u32* tok_pos; u32* tok_len; u32* tok_off; u32* tok_bit; u32 tok_cap; u32* result; u8* out; u32 out_cap; u32* plan;
};

#define ZPQ_EMU_WALK_ONLY
#include "lz77_enc.hip"

// ---- one wave walks a whole block ------------------------------------------------------------------------------------------
namespace {
struct WalkArgs { LzCfg c; u32* table; int nb; bool cand; u32* tpos; u32* tlen; u32* toff; u32 tcap; u32 ntok; u32 end_cur, end_lit; };
WalkArgs g_w;
unsigned long long g_T[256];

template <int NB, bool CAND>
void walk_body() {
  TokSink sink{g_w.tpos, g_w.tlen, g_w.toff, g_w.tcap, 0};
  u32 cur = 0, lit = 0;
  lz_walk<NB, false, CAND>(g_w.c, g_w.table, 0, g_w.c.n, cur, lit, sink, nullptr, g_T);
  if (lane_id() == 0) { g_w.ntok = sink.n; g_w.end_cur = cur; g_w.end_lit = lit; }
}
}  // namespace

// in: n bytes + >= 64 readable bytes behind; table: zeroed hash table of 2^args[5] words (cand == 0) or the candidate table
// of n << args[4] words (cand != 0), 16-byte aligned.  tok: 3 * cap words (pos | len | off).  Returns the token count, or < 0.
extern "C" long walk_emu(const u8* in, u32 n, const int32_t args[9], u32* table, int cand, u32* tok, u32 cap, char* err, u32 err_cap) {
  LzCfg& c = g_w.c;
  c.in = in; c.n = n; c.minMatch = args[2]; c.bucket = (1u << args[4]) - 1; c.htbits = args[5]; c.checkbits = 12 - args[0];
  c.shift1 = (args[5] - 1) / args[2] + 1; c.rb = args[0] > 4 ? args[0] - 4 : 0;
  const u32 mmb = args[2] + 4;
  c.upd_limit = n > mmb ? n - mmb : 0;
  g_w.table = table; g_w.cand = cand != 0; g_w.tpos = tok; g_w.tlen = tok + cap; g_w.toff = tok + 2 * (size_t)cap; g_w.tcap = cap; g_w.ntok = 0;
  memset(g_T, 0, sizeof g_T);
  void (*body)() = nullptr;
  switch (args[4]) {
    case 0: body = cand ? walk_body<1, true> : walk_body<1, false>; break;
    case 1: body = cand ? walk_body<2, true> : walk_body<2, false>; break;
    case 2: body = cand ? walk_body<4, true> : walk_body<4, false>; break;
    default: body = cand ? walk_body<8, true> : walk_body<8, false>; break;
  }
  const char* e = emu::run_wave(body);
  if (e) { if (err && err_cap) { strncpy(err, e, err_cap - 1); err[err_cap - 1] = 0; } return -1; }
  return (long)g_w.ntok;
}
