// Test infrastructure: HIP device source on the CPU.  A workgroup is a set of fibres of ONE host thread that take turns; lane l
// of wave w is fibre 64 w + l.  A wave-level operation (ballot, shuffle, readlane, wave barrier) is a rendezvous of the
// live fibres of a wave, __syncthreads one of all live fibres of the workgroup: a fibre deposits its value, and unless it is
// the last to arrive it waits until the rendezvous is complete; then it reads everybody's values.  Lanes of a real wave run
// in lockstep, so all of them reach the same rendezvous in the same order -- operation ids are compared and a lane that
// strays is reported.  A fibre that leaves the kernel no longer takes part (an exited lane: ballot bit 0).  No memory model,
// no timing: this finds logic errors, the GPU tests find the rest.
// The including file binds the HIP vocabulary (__ballot, __shfl, lane_id(), threadIdx ...) to these functions.
// Fibres switch with a few instructions on x86-64 (callee-saved registers and the stack pointer), ucontext elsewhere.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#if !defined(__x86_64__) || defined(EMU_UCONTEXT)
#include <ucontext.h>
#endif

#if defined(EMU_EXTERN_STATE)
#define EMU_STATE extern __attribute__((visibility("default")))
#else
#define EMU_STATE inline __attribute__((visibility("default")))
#endif

namespace emu {
constexpr int kMaxThreads = 1024, kWave = 64;
struct Scope {
  int size, arrived;                 // live members, members that have arrived at the current rendezvous
  uint32_t gen;
  unsigned long long mask[2];        // bit i = low bit of member i's value (wave scopes: the ballot)
  uint64_t buf[2][kMaxThreads];
  int op[2][kMaxThreads];
  bool dead[kMaxThreads];            // members that have left the kernel (or never existed)
  uint32_t rel[kMaxThreads];         // how often a member has been released from a rendezvous
  unsigned char relpar[kMaxThreads]; // ... and the parity of the generation that released it last
  unsigned long long members[2];     // the lanes of the released group
};
EMU_STATE int g_nthreads, g_tid;                  // fibres of the workgroup, the running fibre
EMU_STATE bool g_done[kMaxThreads];
EMU_STATE Scope g_wave[kMaxThreads / kWave], g_block;
EMU_STATE Scope* g_wait[kMaxThreads];             // the scope a fibre waits in (slot g_wait_idx, release count g_wait_rel), or null
EMU_STATE uint32_t g_wait_rel[kMaxThreads];
EMU_STATE int g_wait_idx[kMaxThreads];
EMU_STATE const char* g_error;
EMU_STATE void (*g_body)();
EMU_STATE bool g_active;                          // inside run_block (else: plain serial execution)
EMU_STATE unsigned long long g_progress;          // rendezvous completed + fibres finished (deadlock detection)
EMU_STATE char* g_stacks;
EMU_STATE int g_rescan;                           // >= 0: a rendezvous has completed, the scheduler goes back to this fibre
#if defined(__x86_64__) && !defined(EMU_UCONTEXT)
EMU_STATE void* g_sp[kMaxThreads];
EMU_STATE void* g_main_sp;
// saves the callee-saved registers and the stack pointer at *save, continues on the stack `load` (assembled at file scope, so
// that the compiler sees an opaque call: everything in memory may have changed when it returns)
extern "C" void emu_fibre_switch(void** save, void* load);
asm(".text\n.weak emu_fibre_switch\n.type emu_fibre_switch,@function\nemu_fibre_switch:\n"
    "pushq %rbp\n pushq %rbx\n pushq %r12\n pushq %r13\n pushq %r14\n pushq %r15\n"
    "movq %rsp, (%rdi)\n movq %rsi, %rsp\n"
    "popq %r15\n popq %r14\n popq %r13\n popq %r12\n popq %rbx\n popq %rbp\n ret\n"
    ".size emu_fibre_switch, .-emu_fibre_switch\n");
static inline void fibre_switch(void** save, void* load) { emu_fibre_switch(save, load); }
inline void to_main() { fibre_switch(&g_sp[g_tid], g_main_sp); }
inline void to_fibre(int i) { g_tid = i; fibre_switch(&g_main_sp, g_sp[i]); }
#else
EMU_STATE ucontext_t g_main, g_ctx[kMaxThreads];
inline void to_main() { swapcontext(&g_ctx[g_tid], &g_main); }
inline void to_fibre(int i) { g_tid = i; swapcontext(&g_main, &g_ctx[i]); }
#endif

inline int lane() { return g_tid & (kWave - 1); }
inline int wave() { return g_tid / kWave; }

// Every live member of the scope has arrived at SOME operation (or the last one missing has left the kernel).  Normally it is
// the same operation for all: everybody is released.  When it is not -- lanes inside a branch or a loop the others have left
// use a wave operation, which the hardware executes for the active lanes only, the others waiting where the paths join --
// the members at the operation with the LOWEST source line are released as a group of their own (ballots and shuffles see
// only them); the others stay, their deposit moves to the next generation.  (Lowest line first: the inside of a branch, a
// loop body or a callee comes before what follows it in the source.  A heuristic -- the structure of the control flow is not
// known here.)  __syncthreads reached by part of a workgroup is an error.
inline void settle(Scope& S) {
  const bool block = &S == &g_block;
  const int n = block ? g_nthreads : kWave;
  const int p = (int)(S.gen & 1u);
  int low = 0x7fffffff, first = -2;
  bool mixed = false;
  for (int i = 0; i < n; ++i) {
    const int o = S.op[p][i];
    if (o == -1) continue;
    if (first == -2) first = o; else if (o != first) mixed = true;
    if (o < low) low = o;
  }
  if (mixed && block && !g_error) {
    static char text[160];
    snprintf(text, sizeof text, "__syncthreads (source line %d) reached by part of the workgroup only", low);
    g_error = text;
  }
  unsigned long long m = 0, members = 0;
  int carried = 0;
  for (int i = 0; i < n; ++i) {
    const int o = S.op[p][i];
    if (S.dead[i] || o == -1) { S.buf[p ^ 1][i] = 0; S.op[p ^ 1][i] = -1; continue; }   // departed: reads 0 from now on
    if (o == low || block) {
      if (i < 64) { m |= (unsigned long long)(S.buf[p][i] & 1u) << i; members |= 1ull << i; }
      ++S.rel[i]; S.relpar[i] = (unsigned char)p;
    } else {
      S.buf[p ^ 1][i] = S.buf[p][i]; S.op[p ^ 1][i] = o; S.buf[p][i] = 0; ++carried;
    }
  }
  S.mask[p] = m; S.members[p] = members;
  S.arrived = carried; ++S.gen; ++g_progress;
  // lockstep: after an operation the lanes go on TOGETHER.  The nearest a fibre scheduler gets to that is lane order:
  // the last to arrive does not run ahead of the others, everybody resumes from the lowest lane up ("all lanes read the
  // counter, the highest lane of the group moves it on" works as on the hardware)
  g_rescan = block ? 0 : (int)(&S - g_wave) * kWave;
}
// deposits v at slot idx of scope S, waits until the members at this operation are released, returns the values of that
// rendezvous (0 for lanes that are not part of it)
inline const uint64_t* rendezvous(Scope& S, int idx, uint64_t v, int op) {
  const int p = (int)(S.gen & 1u);
  const uint32_t my_rel = S.rel[idx];
  S.buf[p][idx] = v; S.op[p][idx] = op;
  if (++S.arrived >= S.size) {
    settle(S);
    if (S.rel[idx] != my_rel) { to_main(); return S.buf[S.relpar[idx]]; }
  }
  g_wait[g_tid] = &S; g_wait_idx[g_tid] = idx; g_wait_rel[g_tid] = my_rel;
  while (S.rel[idx] == my_rel && !g_error) to_main();
  g_wait[g_tid] = nullptr;
  return S.buf[S.relpar[idx]];
}
inline const uint64_t* wave_rendezvous(uint64_t v, int op) { return rendezvous(g_wave[wave()], lane(), v, op); }
inline void block_barrier(int op) { (void)rendezvous(g_block, g_tid, 0, op); }

// a fibre has left the kernel: it no longer counts; a rendezvous the others wait at may be complete now
inline void retire(Scope& S, int idx) {
  const int p = (int)(S.gen & 1u);             // the rendezvous being collected: this member is not part of it
  S.buf[p][idx] = 0; S.op[p][idx] = -1; S.dead[idx] = true;
  --S.size;
  if (S.size > 0 && S.arrived >= S.size) settle(S);
}
inline void fibre_entry() {
  g_body();
  retire(g_wave[wave()], lane());
  retire(g_block, g_tid);
  g_done[g_tid] = true; ++g_progress;
  for (;;) to_main();
}

// runs body() on `nthreads` fibres; returns nullptr or an error text
inline const char* run_block(void (*body)(), int nthreads = 64) {
  const size_t stack = 256 << 10;
  if (!g_stacks) {
    g_stacks = (char*)mmap(nullptr, (size_t)kMaxThreads * stack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (g_stacks == (char*)MAP_FAILED) { g_stacks = nullptr; return "no memory for the fibre stacks"; }
  }
  if (nthreads < 1 || nthreads > kMaxThreads) return "workgroup size out of range";
  g_body = body; g_error = nullptr; g_nthreads = nthreads; g_active = true; g_rescan = -1;
  g_block.size = nthreads; g_block.arrived = 0; g_block.gen = 0;
  memset(g_block.dead, 0, sizeof g_block.dead);
  const int nwaves = (nthreads + kWave - 1) / kWave;
  for (int w = 0; w < nwaves; ++w) {
    Scope& S = g_wave[w];
    S.size = nthreads - w * kWave < kWave ? nthreads - w * kWave : kWave; S.arrived = 0; S.gen = 0;
    memset(S.dead, 0, sizeof S.dead);
    for (int l = S.size; l < kWave; ++l) { S.buf[0][l] = S.buf[1][l] = 0; S.op[0][l] = S.op[1][l] = -1; S.dead[l] = true; }      // lanes that do not exist
  }
  for (int i = 0; i < nthreads; ++i) {
    g_done[i] = false; g_wait[i] = nullptr;
    char* top = g_stacks + (size_t)(i + 1) * stack;
#if defined(__x86_64__) && !defined(EMU_UCONTEXT)
    void** sp = (void**)top - 8;
    for (int k = 0; k < 6; ++k) sp[k] = nullptr;
    sp[6] = (void*)&fibre_entry; sp[7] = nullptr;
    g_sp[i] = sp;
#else
    getcontext(&g_ctx[i]);
    g_ctx[i].uc_stack.ss_sp = top - stack;
    g_ctx[i].uc_stack.ss_size = stack;
    g_ctx[i].uc_link = &g_main;
    makecontext(&g_ctx[i], fibre_entry, 0);
#endif
  }
  long idle_rounds = 0;
  for (;;) {
    int live = 0;
    const unsigned long long before = g_progress;
    for (int i = 0; i < nthreads; ++i) {
      if (g_done[i]) continue;
      ++live;
      if (g_wait[i] && g_wait[i]->rel[g_wait_idx[i]] == g_wait_rel[i]) continue;       // still waiting for the others
      to_fibre(i);
      if (g_rescan >= 0) { i = g_rescan - 1; g_rescan = -1; }
    }
    if (!live) break;
    if (g_error) break;                               // (the fibres are abandoned)
    // a whole round in which no rendezvous completed and nobody finished: e.g. lanes wait at an operation inside a branch
    // the others did not take
    if (g_progress == before) { if (++idle_rounds > 2) { g_error = "no progress: some threads wait at a wave / block operation the others never reach"; break; } }
    else idle_rounds = 0;
  }
  g_active = false;
  return g_error;
}

// ---- the operations ----------------------------------------------------------------------------------------------------
inline unsigned long long ballot(bool p, int op) {
  Scope& S = g_wave[wave()];
  (void)rendezvous(S, lane(), p ? 1 : 0, op);
  return S.mask[S.relpar[lane()]];
}
template <class T> inline T shfl(T v, int src, int op) {
  static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes");
  uint64_t bits = 0; memcpy(&bits, &v, sizeof v);
  const uint64_t* a = wave_rendezvous(bits, op);
  T r; memcpy(&r, &a[src & (kWave - 1)], sizeof r);
  return r;
}
// the value of the lowest lane that takes part (readfirstlane reads the first ACTIVE lane)
template <class T> inline T first(T v, int op) {
  static_assert(sizeof(T) <= 8, "at most 8 bytes");
  uint64_t bits = 0; memcpy(&bits, &v, sizeof v);
  Scope& S = g_wave[wave()];
  const uint64_t* a = rendezvous(S, lane(), bits, op);
  const unsigned long long mem = S.members[S.relpar[lane()]];
  T r; memcpy(&r, &a[mem ? __builtin_ctzll(mem) : 0], sizeof r);
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
