// Test infrastructure: HIP device source on the CPU.  A workgroup is a set of fibres (ucontext) of ONE host thread that take
// turns; lane l of wave w is fibre 64 w + l.  A wave-level operation (ballot, shuffle, readlane, wave barrier) is a
// rendezvous of the 64 fibres of a wave, __syncthreads one of all fibres of the workgroup: a fibre deposits its value, and
// unless it is the last to arrive it yields until the rendezvous is complete; then it reads everybody's values.  Lanes of a
// real wave run in lockstep, so all of them reach the same rendezvous in the same order -- operation ids are compared and a
// lane that strays is reported.  No memory model, no timing: this finds logic errors, the GPU tests find the rest.
// The including file binds the HIP vocabulary (__ballot, __shfl, lane_id(), threadIdx ...) to these functions.
#pragma once
#include <stdint.h>
#include <string.h>
#include <ucontext.h>

#include <vector>

namespace emu {
constexpr int kMaxThreads = 1024, kWave = 64;
struct Scope {
  int size = 0, arrived = 0;
  uint32_t gen = 0;
  uint64_t buf[2][kMaxThreads];
  int op[2][kMaxThreads];
};
inline ucontext_t g_main, g_ctx[kMaxThreads];
inline int g_nthreads = 64, g_tid = 0;          // fibres of the workgroup, the running fibre
inline bool g_done[kMaxThreads];
inline Scope g_wave[kMaxThreads / kWave], g_block;
inline const char* g_error = nullptr;
inline void (*g_body)() = nullptr;
inline bool g_active = false;                   // inside run_block (else: plain serial execution)
inline unsigned long long g_progress = 0;       // rendezvous completed + fibres finished (deadlock detection)

inline int lane() { return g_tid & (kWave - 1); }
inline int wave() { return g_tid / kWave; }
inline void yield() { swapcontext(&g_ctx[g_tid], &g_main); }

// deposits v at slot idx of scope S, waits for the other members, returns all values of this rendezvous
inline const uint64_t* rendezvous(Scope& S, int idx, uint64_t v, int op) {
  const int p = (int)(S.gen & 1u);
  const uint32_t my_gen = S.gen;
  S.buf[p][idx] = v; S.op[p][idx] = op;
  if (++S.arrived == S.size) { S.arrived = 0; ++S.gen; ++g_progress; }
  else while (S.gen == my_gen && !g_error) yield();
  for (int i = 0; i < S.size; ++i)
    if (S.op[p][i] != op && !g_error) g_error = "threads reached different wave / block operations (divergent intrinsic)";
  return S.buf[p];
}
inline const uint64_t* wave_rendezvous(uint64_t v, int op) { return rendezvous(g_wave[wave()], lane(), v, op); }
inline void block_barrier(int op) { (void)rendezvous(g_block, g_tid, 0, op); }

inline void trampoline() { g_body(); g_done[g_tid] = true; ++g_progress; swapcontext(&g_ctx[g_tid], &g_main); }

// runs body() on `nthreads` fibres (a multiple of 64); returns nullptr or an error text
inline const char* run_block(void (*body)(), int nthreads = 64) {
  static std::vector<char> stacks;
  const size_t stack = 128 << 10;
  if (stacks.size() < (size_t)nthreads * stack) stacks.resize((size_t)nthreads * stack);
  g_body = body; g_error = nullptr; g_nthreads = nthreads; g_active = true;
  g_block.size = nthreads; g_block.arrived = 0; g_block.gen = 0;
  for (int w = 0; w < nthreads / kWave; ++w) { g_wave[w].size = kWave; g_wave[w].arrived = 0; g_wave[w].gen = 0; }
  for (int i = 0; i < nthreads; ++i) {
    g_done[i] = false;
    getcontext(&g_ctx[i]);
    g_ctx[i].uc_stack.ss_sp = stacks.data() + (size_t)i * stack;
    g_ctx[i].uc_stack.ss_size = stack;
    g_ctx[i].uc_link = &g_main;
    makecontext(&g_ctx[i], trampoline, 0);
  }
  long idle_rounds = 0;
  for (;;) {
    int live = 0;
    const unsigned long long before = g_progress;
    for (int i = 0; i < nthreads; ++i) {
      if (g_done[i]) continue;
      ++live; g_tid = i; swapcontext(&g_main, &g_ctx[i]);
    }
    if (!live) break;
    if (g_error) break;                               // (the fibres are abandoned)
    // a whole round in which no rendezvous completed and nobody finished: e.g. a fibre has left the kernel while its
    // wave still waits for it (divergent exit)
    if (g_progress == before) { if (++idle_rounds > 4) { g_error = "no progress: some threads wait at a wave / block operation the others never reach"; break; } }
    else idle_rounds = 0;
  }
  g_active = false;
  return g_error;
}

// ---- the operations ----------------------------------------------------------------------------------------------------
inline unsigned long long ballot(bool p, int op) {
  const uint64_t* a = wave_rendezvous(p ? 1 : 0, op);
  unsigned long long m = 0;
  for (int i = 0; i < kWave; ++i) m |= (unsigned long long)(a[i] & 1) << i;
  return m;
}
template <class T> inline T shfl(T v, int src, int op) {
  static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes");
  uint64_t bits = 0; memcpy(&bits, &v, sizeof v);
  const uint64_t* a = wave_rendezvous(bits, op);
  T r; memcpy(&r, &a[src & (kWave - 1)], sizeof r);
  return r;
}
inline bool any(bool p, int op) { return ballot(p, op) != 0; }
inline bool all(bool p, int op) { return ballot(!p, op) == 0; }
// __builtin_amdgcn_update_dpp(old, src, ctrl, row_mask, bank_mask = 0xf, bound_ctrl = false): the controls this engine uses --
// wave_shr:1 (0x138), row_shr:1..15 (0x111..0x11f), row_bcast:15 (0x142), row_bcast:31 (0x143).  A lane whose row is not in
// row_mask, or whose source does not exist, keeps `old`.
inline int dpp(int old, int src, int ctrl, int row_mask, int op) {
  const int l = lane(), row = l >> 4;
  int from = -1;
  if (ctrl == 0x138) from = l >= 1 ? l - 1 : -1;
  else if (ctrl >= 0x111 && ctrl <= 0x11f) { const int k = ctrl - 0x110; from = (l & 15) >= k ? l - k : -1; }
  else if (ctrl == 0x142) from = row >= 1 ? (row - 1) * 16 + 15 : -1;
  else if (ctrl == 0x143) from = l >= 32 ? 31 : -1;
  else g_error = "unsupported DPP control";
  const int got = shfl(src, from >= 0 ? from : l, op);
  if (!((row_mask >> row) & 1) || from < 0) return old;
  return got;
}
}  // namespace emu
