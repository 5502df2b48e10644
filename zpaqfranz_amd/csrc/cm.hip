// Context-mixing codec for ZPAQ blocks with n > 0 components: the Predictor
// (CONS/CM/ICM/MATCH/AVG/MIX2/MIX/ISSE/SSE), the 32-bit binary arithmetic coder, and the ZPAQL virtual
// machine that computes the component contexts (HCOMP) and post-processes decoded data (PCOMP).
// Reference: Predictor::init/predict0/update0/find/train (ZSFX/libzpaq.cpp:1715-2080, ZSFX/libzpaq.h:1151-1185),
// Decoder::decode/decompress (:2096-2137), Encoder (declaration ZSFX/libzpaq.h:1273-1286; mirror of the
// decoder), ZPAQL::run0/execute (:1019-1254), PostProcessor::write (:2185-2226).
//
// The lookup tables are GENERATED (squash/stretch from their defining formulas, quoted in the reference
// at :1733 and :1739; the bit-history state table from the ZPAQ specification's num_states/next_state
// rules); tests compare every entry with the reference's literal tables.
//
// One block = one serial chain of bit decisions.  cm_wave_kernel gives a block a wave and every component a
// lane (state in registers, tables in LDS, one load round per bit); cm_code_kernel is the plain one-lane
// walk, kept for models with more than 64 components or more mixers/SSE stages than the wave kernel has
// register slots for.  Measured split and next steps: DESIGN.md section 7-2.
#include <math.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <map>

#include "zpq_internal.h"

namespace {

enum { NONE = 0, CONS, CM, ICM, MATCH, AVG, MIX2, MIX, ISSE, SSE };
const int kCompSize[10] = {0, 2, 3, 2, 3, 4, 6, 6, 3, 5};   // ZSFX/libzpaq.cpp:706

struct Tables {
  u16 squash[4096];
  short stretch[32768];
  int dt[1024];
  int dt2k[256];
  u8 ns[1024];
};

// ---- table generation (host) ------------------------------------------------------------------------
int num_states(int n0, int n1) {
  const int B = 6;
  const int bound[B] = {20, 48, 15, 8, 6, 5};
  if (n0 < n1) return num_states(n1, n0);
  if (n0 < 0 || n1 < 0 || n1 >= B || n0 > bound[n1]) return 0;
  return 1 + (n1 > 0 && n0 + n1 <= 17);
}
void discount(int& n0) { n0 = (n0 >= 1) + (n0 >= 2) + (n0 >= 3) + (n0 >= 4) + (n0 >= 5) + (n0 >= 7) + (n0 >= 8); }
void next_state(int& n0, int& n1, int y) {
  if (n0 < n1) { next_state(n1, n0, 1 - y); return; }
  if (y) { ++n1; discount(n0); } else { ++n0; discount(n1); }
  while (!num_states(n0, n1)) {
    if (n1 < 2) --n0;
    else { n0 = (n0 * (n1 - 1) + (n1 / 2)) / n1; --n1; }
  }
}

void make_tables(Tables& T) {
  for (int i = 0; i < 4096; ++i) {                       // squash(x) = floor(32768/(1+e^(-x/64))), x = i-2048
    int v = (int)(32768.0 / (1 + exp((i - 2048) * (-1.0 / 64))));
    T.squash[i] = (u16)(v < 0 ? 0 : v > 32767 ? 32767 : v);
  }
  for (int i = 16384; i < 32768; ++i)                    // stretch = ln(p/(1-p)) in 1/64 units, odd symmetric
    T.stretch[i] = (short)((int)(log((i + 0.5) / (32767.5 - i)) * 64 + 0.5 + 100000) - 100000);
  for (int i = 0; i < 16384; ++i) T.stretch[i] = (short)-T.stretch[32767 - i];
  for (int i = 0; i < 1024; ++i) T.dt[i] = (1 << 17) / (i * 2 + 3) * 2;
  T.dt2k[0] = 0;
  for (int i = 1; i < 256; ++i) T.dt2k[i] = 2048 / i;
  // bit-history states ordered by n0+n1, then n1 (ZPAQ specification)
  const int N = 50;
  static u8 t[N][N][2];
  memset(t, 0, sizeof t);
  int state = 0;
  for (int i = 0; i < N; ++i)
    for (int n1 = 0; n1 <= i; ++n1) {
      const int n0 = i - n1, n = num_states(n0, n1);
      if (n) { t[n0][n1][0] = (u8)state; t[n0][n1][1] = (u8)(state + n - 1); state += n; }
    }
  memset(T.ns, 0, sizeof T.ns);
  for (int n0 = 0; n0 < N; ++n0)
    for (int n1 = 0; n1 < N; ++n1)
      for (int y = 0; y < num_states(n0, n1); ++y) {
        const int s = t[n0][n1][y];
        int s0 = n0, s1 = n1;
        next_state(s0, s1, 0); T.ns[s * 4 + 0] = t[s0][s1][0];
        s0 = n0; s1 = n1;
        next_state(s0, s1, 1); T.ns[s * 4 + 1] = t[s0][s1][1];
        T.ns[s * 4 + 2] = (u8)n0; T.ns[s * 4 + 3] = (u8)n1;
      }
}

const Tables& host_tables() {
  static Tables T;
  static bool done = false;
  if (!done) { make_tables(T); done = true; }
  return T;
}

// ---- device-side model -----------------------------------------------------------------------------------
struct Comp {
  u32 type, a1, a2, a3, a4, a5;   // component type and its (up to 5) header arguments
  u32 limit, cxt, a, b, c;        // Component scalars (ZSFX/libzpaq.h:1084-1111)
  u32* cm; u32 cm_mask;           // cm[] and size-1
  u8* ht; u32 ht_mask;
  u16* a16; u32 a16_mask;
};

struct Vm {                       // one ZPAQL machine (HCOMP or PCOMP)
  const u8* prog; u32 plen;       // bytecode, execution starts at 0
  u32* H; u32 hmask; u8* M; u32 mmask; u32* R;
  u32 a, b, c, d, f;
  u8* out; u32 out_cap, out_len;  // OUT instruction target (PCOMP only)
  int err;
  u32 in_lds;                     // bit 0: prog points into LDS, bit 1: H does (set by the kernel that put them there)
  // interpreted instructions a call may take: `budget` free per call (0: 2^30, a post-processor's one long call at the end of a
  // segment) plus what is left of `credit`, which the whole block has once (HCOMP: 2^20 per byte + 2^24 once -- a long
  // initialisation loop passes, an endless loop is refused after ~seven seconds of one lane instead of 2^30 steps = minutes)
  u32 budget, credit;
};

struct CmJobDev {
  u32 n;                          // components
  unsigned long long dep;         // bit i: component i's prediction depends on earlier ones (n <= 64)
  Comp* comp;
  int* p;                         // p[256]
  u32* h;                         // h[256]
  Vm vm;                          // HCOMP machine
  const Tables* T;
  const u8* in; u32 in_len;
  u8* out; u32 out_cap;
  u32* result;                    // [0]=bytes produced, [1]=status
  u32* seg; u32 nseg;             // segments of the block (null: one): u32[nseg] input bytes each, then u32[nseg] (result) output end of each
};

__device__ __forceinline__ int clamp2k(int x) { return x < -2048 ? -2048 : x > 2047 ? 2047 : x; }
__device__ __forceinline__ int clamp512k(int x) { return x < -(1 << 19) ? -(1 << 19) : x >= (1 << 19) ? (1 << 19) - 1 : x; }

// ZPAQL interpreter (ZSFX/libzpaq.cpp:1033-1254).  Operand encodings are regular: op&7 selects
// A B C D *B *C *D N for the two-operand groups.
template <class ProgPtr, class HPtr>
__device__ void vm_body(Vm& z, u32 input, ProgPtr P, HPtr H) {
  // everything the loop touches is copied to locals first: a byte store into M[] may alias *z as far as the
  // compiler can tell, and it would otherwise re-load the pointers and masks from HBM after every store
  const u32 plen = z.plen, hmask = z.hmask, mmask = z.mmask;
  typedef __attribute__((address_space(1))) u8 g_u8;
  typedef __attribute__((address_space(1))) u32 g_u32;
  g_u8* const M = (g_u8*)z.M; g_u32* const R = (g_u32*)z.R;      // always in HBM: global_*, not flat_* instructions
  u32 out_len = z.out_len; const u32 out_cap = z.out_cap; u8* const outp = z.out;
  int err = 0;
  u32 pc = 0, a = input, b = z.b, c = z.c, d = z.d, f = z.f;
  // (2^30 interpreted instructions per call: a program that is still running then is refused -- err = 2, ZPQ_ERR_LIMIT -- where
  //  the reference would go on, ZSFX/libzpaq.cpp:1033-1254 has no limit; until round 6 the call simply ended as if it had halted)
  const u32 free_steps = z.budget ? z.budget : (1u << 30);
  const u32 max_steps = free_steps + z.credit;                  // (credit <= 2^24: no wrap)
  u32 guard = 0;
  for (;; ++guard) {
    if (guard >= max_steps) { err = 2; break; }
    // The machine runs on one lane (or on lanes in identical states): telling the compiler that the
    // opcode, the program counter and the flag are wave-uniform turns the dispatch below into scalar
    // branches instead of a tree of exec-mask splits.
    pc = (u32)__builtin_amdgcn_readfirstlane((int)pc);
    if (pc >= plen) { err = 1; break; }
    const u32 op = (u32)__builtin_amdgcn_readfirstlane((int)(u32)P[pc++]);
    if (op == 56) break;                                     // HALT
    if (op >= 64 && op < 240 && (op < 120 || op >= 128)) {
      const u32 sel = op & 7, grp = op >> 3;
      u32 v;
      switch (sel) {
        case 0: v = a; break; case 1: v = b; break; case 2: v = c; break; case 3: v = d; break;
        case 4: v = M[b & mmask]; break; case 5: v = M[c & mmask]; break; case 6: v = H[d & hmask]; break;
        default: v = P[pc++]; break;
      }
      switch (grp) {
        case 8: a = v; break; case 9: b = v; break; case 10: c = v; break; case 11: d = v; break;
        case 12: M[b & mmask] = (u8)v; break; case 13: M[c & mmask] = (u8)v; break; case 14: H[d & hmask] = v; break;
        case 16: a += v; break; case 17: a -= v; break; case 18: a *= v; break;
        case 19: a = v ? a / v : 0; break; case 20: a = v ? a % v : 0; break;
        case 21: a &= v; break; case 22: a &= ~v; break; case 23: a |= v; break; case 24: a ^= v; break;
        case 25: a <<= (v & 31); break; case 26: a >>= (v & 31); break;
        case 27: f = a == v; break; case 28: f = a < v; break; case 29: f = a > v; break;
        default: err = 1; break;
      }
      if (err) break;
      continue;
    }
    switch (op) {
      case 1: ++a; break; case 2: --a; break; case 3: a = ~a; break; case 4: a = 0; break;
      case 7: a = R[P[pc++]]; break;
      case 8: { u32 t = a; a = b; b = t; } break; case 9: ++b; break; case 10: --b; break; case 11: b = ~b; break; case 12: b = 0; break;
      case 15: b = R[P[pc++]]; break;
      case 16: { u32 t = a; a = c; c = t; } break; case 17: ++c; break; case 18: --c; break; case 19: c = ~c; break; case 20: c = 0; break;
      case 23: c = R[P[pc++]]; break;
      case 24: { u32 t = a; a = d; d = t; } break; case 25: ++d; break; case 26: --d; break; case 27: d = ~d; break; case 28: d = 0; break;
      case 31: d = R[P[pc++]]; break;
      // a byte of M swaps with the LOW byte of A only (swap(U8&), ZSFX/libzpaq.h:1073)
      case 32: { auto& x = M[b & mmask]; u32 t = x; x = (u8)a; a = (a & 0xffffff00u) | t; } break;
      case 33: ++M[b & mmask]; break; case 34: --M[b & mmask]; break;
      case 35: M[b & mmask] = ~M[b & mmask]; break; case 36: M[b & mmask] = 0; break;
      case 39: if (__builtin_amdgcn_readfirstlane((int)f)) pc += ((P[pc] + 128) & 255) - 127; else ++pc; break;          // JT
      case 40: { auto& x = M[c & mmask]; u32 t = x; x = (u8)a; a = (a & 0xffffff00u) | t; } break;
      case 41: ++M[c & mmask]; break; case 42: --M[c & mmask]; break;
      case 43: M[c & mmask] = ~M[c & mmask]; break; case 44: M[c & mmask] = 0; break;
      case 47: if (!__builtin_amdgcn_readfirstlane((int)f)) pc += ((P[pc] + 128) & 255) - 127; else ++pc; break;         // JF
      case 48: { auto& x = H[d & hmask]; u32 t = x; x = a; a = t; } break;
      case 49: ++H[d & hmask]; break; case 50: --H[d & hmask]; break;
      case 51: H[d & hmask] = ~H[d & hmask]; break; case 52: H[d & hmask] = 0; break;
      case 55: R[P[pc++]] = a; break;
      case 57: if (outp) { if (out_len < out_cap) outp[out_len] = (u8)a; ++out_len; } break;   // OUT
      case 59: a = (a + M[b & mmask] + 512) * 773; break;                        // HASH
      case 60: H[d & hmask] = (H[d & hmask] + a + 512) * 773; break;         // HASHD
      case 63: pc += ((P[pc] + 128) & 255) - 127; break;                             // JMP
      case 255: { u32 t = P[pc] + 256u * P[pc + 1]; if (t >= plen) { err = 1; } pc = t; } break;   // LJ
      default: err = 1; break;
    }
    if (err) break;
  }
  z.a = a; z.b = b; z.c = c; z.d = d; z.f = f; z.out_len = out_len;
  if (err) z.err = err;
  if (guard > free_steps) z.credit = guard - free_steps > z.credit ? 0u : z.credit - (guard - free_steps);
}

// The wave coder keeps the HCOMP program and a small H[] in LDS: give those the ds_* path (a flat access
// also waits for every outstanding global access, and the other way round).
__device__ void vm_run(Vm& z, u32 input) {
  typedef __attribute__((address_space(3))) const u8 l_cu8;
  typedef __attribute__((address_space(3))) u32 l_u32;
  typedef __attribute__((address_space(1))) const u8 g_cu8;
  typedef __attribute__((address_space(1))) u32 g_u32;
  const bool pl = z.in_lds & 1, hl = z.in_lds & 2;
  if (pl && hl) vm_body(z, input, (l_cu8*)z.prog, (l_u32*)z.H);
  else if (!pl && !hl) vm_body(z, input, (g_cu8*)z.prog, (g_u32*)z.H);
  else vm_body(z, input, z.prog, z.H);
}

// find(): ZSFX/libzpaq.cpp:2064-2080
__device__ u32 cm_find(u8* ht, u32 ht_size, int sizebits, u32 cxt) {
  const u32 chk = (cxt >> sizebits) & 255;
  const u32 h0 = (cxt * 16) & (ht_size - 16);
  if (ht[h0] == chk) return h0;
  const u32 h1 = h0 ^ 16;
  if (ht[h1] == chk) return h1;
  const u32 h2 = h0 ^ 32;
  if (ht[h2] == chk) return h2;
  u32 r;
  if (ht[h0 + 1] <= ht[h1 + 1] && ht[h0 + 1] <= ht[h2 + 1]) r = h0;
  else if (ht[h1 + 1] < ht[h2 + 1]) r = h1;
  else r = h2;
  for (int i = 0; i < 16; ++i) ht[r + i] = 0;
  ht[r] = (u8)chk;
  return r;
}

struct Pred {
  const CmJobDev& J;
  u32 c8, hmap4;
  __device__ Pred(const CmJobDev& j) : J(j), c8(1), hmap4(1) {}
  __device__ int squash(int x) const { return J.T->squash[x + 2048]; }
  __device__ int stretch(u32 x) const { return J.T->stretch[x]; }

  __device__ int predict() {          // predict0, ZSFX/libzpaq.cpp:1846-1943
    int* p = J.p; const u32* h = J.h;
    const u32 n = J.n;
    for (u32 i = 0; i < n; ++i) {
      Comp& cr = J.comp[i];
      switch (cr.type) {
        case CONS: break;
        case CM:
          cr.cxt = h[i] ^ hmap4;
          p[i] = stretch(cr.cm[cr.cxt & cr.cm_mask] >> 17);
          break;
        case ICM:
          if (c8 == 1 || (c8 & 0xf0) == 16) cr.c = cm_find(cr.ht, cr.ht_mask + 1, cr.a1 + 2, h[i] + 16 * c8);
          cr.cxt = cr.ht[cr.c + (hmap4 & 15)];
          p[i] = stretch(cr.cm[cr.cxt & cr.cm_mask] >> 8);
          break;
        case MATCH:
          if (cr.a == 0) p[i] = 0;
          else {
            cr.c = (cr.ht[(cr.limit - cr.b) & cr.ht_mask] >> (7 - cr.cxt)) & 1;
            p[i] = stretch((u32)(J.T->dt2k[cr.a] * ((int)cr.c * -2 + 1)) & 32767u);
          }
          break;
        case AVG:
          p[i] = (p[cr.a1] * (int)cr.a3 + p[cr.a2] * (256 - (int)cr.a3)) >> 8;
          break;
        case MIX2: {
          cr.cxt = (h[i] + (c8 & cr.a5)) & (cr.c - 1);
          const int w = cr.a16[cr.cxt];
          p[i] = (w * p[cr.a2] + (65536 - w) * p[cr.a3]) >> 16;
        } break;
        case MIX: {
          const int m = (int)cr.a3;
          cr.cxt = h[i] + (c8 & cr.a5);
          cr.cxt = (cr.cxt & (cr.c - 1)) * m;
          const int* wt = (const int*)&cr.cm[cr.cxt];
          int s = 0;
          for (int j = 0; j < m; ++j) s += (wt[j] >> 8) * p[cr.a2 + j];
          p[i] = clamp2k(s >> 8);
        } break;
        case ISSE: {
          if (c8 == 1 || (c8 & 0xf0) == 16) cr.c = cm_find(cr.ht, cr.ht_mask + 1, cr.a1 + 2, h[i] + 16 * c8);
          cr.cxt = cr.ht[cr.c + (hmap4 & 15)];
          const int* wt = (const int*)&cr.cm[cr.cxt * 2];
          p[i] = clamp2k((wt[0] * p[cr.a2] + wt[1] * 64) >> 16);
        } break;
        case SSE: {
          cr.cxt = (h[i] + c8) * 32;
          int pq = p[cr.a2] + 992;
          if (pq < 0) pq = 0;
          if (pq > 1983) pq = 1983;
          const int wt = pq & 63;
          pq >>= 6;
          cr.cxt += pq;
          p[i] = stretch(((cr.cm[cr.cxt & cr.cm_mask] >> 10) * (64 - wt) + (cr.cm[(cr.cxt + 1) & cr.cm_mask] >> 10) * wt) >> 13);
          cr.cxt += wt >> 5;
        } break;
        default: break;
      }
    }
    return squash(p[n - 1]);
  }

  __device__ void train(Comp& cr, int y) {     // ZSFX/libzpaq.h:1151-1157; the product wraps in 32 bits
    u32& pn = cr.cm[cr.cxt & cr.cm_mask];
    const u32 count = pn & 0x3ff;
    const int error = y * 32767 - (int)(pn >> 17);
    pn += ((u32)error * (u32)J.T->dt[count] & 0xfffffc00u) + (count < cr.limit);
  }

  __device__ void update(int y) {              // update0, ZSFX/libzpaq.cpp:1946-2058
    int* p = J.p; u32* h = J.h;
    const u8* ns = J.T->ns;
    const u32 n = J.n;
    for (u32 i = 0; i < n; ++i) {
      Comp& cr = J.comp[i];
      switch (cr.type) {
        case CM: train(cr, y); break;
        case ICM: {
          u8& bh = cr.ht[cr.c + (hmap4 & 15)];
          bh = ns[bh * 4 + y];
          u32& pn = cr.cm[cr.cxt & cr.cm_mask];
          pn += (u32)((int)(y * 32767 - (int)(pn >> 8)) >> 2);
        } break;
        case MATCH: {
          if ((int)cr.c != y) cr.a = 0;
          u8& cur = cr.ht[cr.limit & cr.ht_mask];
          cur = (u8)(cur + cur + y);
          if (++cr.cxt == 8) {
            cr.cxt = 0;
            ++cr.limit;
            cr.limit &= (1u << cr.a2) - 1;
            if (cr.a == 0) {
              cr.b = cr.limit - cr.cm[h[i] & cr.cm_mask];
              if (cr.b & cr.ht_mask)
                while (cr.a < 255 && cr.ht[(cr.limit - cr.a - 1) & cr.ht_mask] == cr.ht[(cr.limit - cr.a - cr.b - 1) & cr.ht_mask]) ++cr.a;
            } else cr.a += cr.a < 255;
            cr.cm[h[i] & cr.cm_mask] = cr.limit;
          }
        } break;
        case MIX2: {
          const int err = (y * 32767 - squash(p[i])) * (int)cr.a4 >> 5;
          int w = cr.a16[cr.cxt];
          w += (err * (p[cr.a2] - p[cr.a3]) + (1 << 12)) >> 13;
          if (w < 0) w = 0;
          if (w > 65535) w = 65535;
          cr.a16[cr.cxt] = (u16)w;
        } break;
        case MIX: {
          const int m = (int)cr.a3;
          const int err = (y * 32767 - squash(p[i])) * (int)cr.a4 >> 4;
          int* wt = (int*)&cr.cm[cr.cxt];
          for (int j = 0; j < m; ++j) wt[j] = clamp512k(wt[j] + ((err * p[cr.a2 + j] + (1 << 12)) >> 13));
        } break;
        case ISSE: {
          const int err = y * 32767 - squash(p[i]);
          int* wt = (int*)&cr.cm[cr.cxt * 2];
          wt[0] = clamp512k(wt[0] + ((err * p[cr.a2] + (1 << 12)) >> 13));
          wt[1] = clamp512k(wt[1] + ((err + 16) >> 5));
          cr.ht[cr.c + (hmap4 & 15)] = ns[cr.cxt * 4 + y];
        } break;
        case SSE: train(cr, y); break;
        default: break;
      }
    }
    c8 += c8 + y;
    if (c8 >= 256) {
      Vm& z = const_cast<Vm&>(J.vm);
      vm_run(z, c8 - 256);
      hmap4 = 1;
      c8 = 1;
      for (u32 i = 0; i < n; ++i) h[i] = z.H[i & z.hmask];
    } else if (c8 >= 16 && c8 < 32) hmap4 = (hmap4 & 0xf) << 5 | y << 4 | 1;
    else hmap4 = (hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2 + y) & 0xf);
  }
};

__global__ __launch_bounds__(64) void cm_code_kernel(CmJobDev* jobs, int encode) {
  if (threadIdx.x != 0) return;
  CmJobDev& J = jobs[blockIdx.x];
  Pred pr(J);
  u32 low = 1, high = 0xffffffffu, op = 0;
  int status = ZPQ_OK;
  if (encode) {
    auto put = [&](u32 c) { if (op < J.out_cap) J.out[op] = (u8)c; ++op; };
    auto enc = [&](int y, u32 p) {
      const u32 mid = low + (u32)(((u64)(high - low) * p) >> 16);
      if (y) high = mid; else low = mid + 1;
      while ((high ^ low) < 0x1000000u) { put(high >> 24); high = high << 8 | 255; low <<= 8; low += (low == 0); }
    };
    // a block's segments one after the other: the model carries on, the coder ends each with its end-of-segment symbol
    const u32 nseg = J.seg ? J.nseg : 1u;
    u32 i = 0;
    for (u32 s = 0; s < nseg && !J.vm.err; ++s) {
      const u32 send = J.seg ? i + J.seg[s] : J.in_len;
      for (; i < send && i < J.in_len && !J.vm.err; ++i) {
        const u32 c = J.in[i];
        enc(0, 0);
        for (int b = 7; b >= 0; --b) {
          const u32 p = (u32)pr.predict() * 2 + 1;
          const int y = (c >> b) & 1;
          enc(y, p);
          pr.update(y);
        }
      }
      enc(1, 0);
      put(0); put(0); put(0); put(0);
      if (J.seg) J.seg[nseg + s] = op;
    }
    if (op > J.out_cap) status = ZPQ_ERR_CAPACITY;
  } else {
    u32 ip = 0, curr = 0, send = J.in_len;
    bool bad = false;
    auto get = [&]() -> u32 { if (ip < send) return J.in[ip++]; bad = true; return 0; };
    auto dec = [&](u32 p) -> int {               // Decoder::decode, ZSFX/libzpaq.cpp:2096-2114
      if (curr < low || curr > high) { bad = true; return 1; }
      const u32 mid = low + (u32)(((u64)(high - low) * p) >> 16);
      int y;
      if (curr <= mid) { y = 1; high = mid; } else { y = 0; low = mid + 1; }
      while ((high ^ low) < 0x1000000u) { high = high << 8 | 255; low <<= 8; low += (low == 0); curr = curr << 8 | get(); }
      return y;
    };
    const u32 nseg = J.seg ? J.nseg : 1u;
    for (u32 s = 0; s < nseg; ++s) {             // (Decoder::decompress: `curr == 0` starts a segment, ZSFX/libzpaq.cpp:2122-2126)
      if (J.seg) { send = ip + J.seg[s]; if (send > J.in_len) send = J.in_len; }
      curr = 0;
      for (int i = 0; i < 4; ++i) curr = curr << 8 | get();
      while (!bad && !J.vm.err) {
        if (dec(0)) { if (curr != 0) bad = true; break; }
        u32 c = 1;
        while (c < 256) {
          const u32 p = (u32)pr.predict() * 2 + 1;
          c += c + (u32)dec(p);
          pr.update((int)(c & 1));
        }
        if (op >= J.out_cap) { status = ZPQ_ERR_CAPACITY; break; }   // caller asked for a prefix only
        J.out[op++] = (u8)(c - 256);
      }
      if (bad || J.vm.err || status != ZPQ_OK) break;
      if (J.seg) { if (ip != send) { bad = true; break; } J.seg[nseg + s] = op; }     // a segment uses exactly its own bytes
    }
    if (bad) status = ZPQ_ERR_FORMAT;
  }
  if (J.vm.err) status = J.vm.err == 2 ? ZPQ_ERR_LIMIT : ZPQ_ERR_FORMAT;
  J.result[0] = op;
  J.result[1] = (u32)status;
}

// ---- wave-per-block coder: lanes = components ------------------------------------------------------
// The bit decisions of a block form one serial chain, but inside one decision most of the work is
// independent per component: computing the context slot, fetching its state from HBM (a random
// access each), and -- after the bit is known -- training it.  Lane i owns component i (its
// descriptor and scalars live in registers); one pass serves every table lookup of a bit at once,
// so a decision costs one memory round trip instead of one per component.  What is sequential by
// the model's definition stays sequential, in component order, on wave-uniform values: ISSE/AVG/MIX2
// refine an earlier prediction, MIX is a dot product over lanes (DPP reduction), SSE interpolates.
// The coder registers (low/high/curr) and c8/hmap4 are uniform; lane 0 runs the ZPAQL HCOMP program
// once per byte.  Blocks with more than 64 components take the one-lane kernel above.
template <int CTRL, int ROWS>
__device__ __forceinline__ int dpp_add_step(int v) {      // v += neighbour (0 where there is none)
  return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROWS, 0xf, false);
}
__device__ __forceinline__ int wave_sum_to_last(int v) {   // lane 63 ends up with the sum of all lanes
  v = dpp_add_step<0x111, 0xf>(v); v = dpp_add_step<0x112, 0xf>(v); v = dpp_add_step<0x114, 0xf>(v);
  v = dpp_add_step<0x118, 0xf>(v); v = dpp_add_step<0x142, 0xa>(v); v = dpp_add_step<0x143, 0xc>(v);
  return v;
}
__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ u32 rlu(u32 v, int lane) { return (u32)__builtin_amdgcn_readlane((int)v, lane); }

// Table memory is addressed through address-space qualified pointers: global_* for the component
// arrays in HBM, ds_* for the model tables in LDS.  A flat_* access would make every wait a wait for
// both kinds.
typedef __attribute__((address_space(1))) u32 g_u32;
typedef __attribute__((address_space(1))) u16 g_u16;
typedef __attribute__((address_space(1))) u8 g_u8;
typedef __attribute__((address_space(3))) const Tables lds_tables;

// find(): ZSFX/libzpaq.cpp:2064-2080
__device__ u32 cm_find_g(g_u8* ht, u32 ht_size, int sizebits, u32 cxt) {
  const u32 chk = (cxt >> sizebits) & 255;
  const u32 h0 = (cxt * 16) & (ht_size - 16);
  if (ht[h0] == chk) return h0;
  const u32 h1 = h0 ^ 16;
  if (ht[h1] == chk) return h1;
  const u32 h2 = h0 ^ 32;
  if (ht[h2] == chk) return h2;
  u32 r;
  if (ht[h0 + 1] <= ht[h1 + 1] && ht[h0 + 1] <= ht[h2 + 1]) r = h0;
  else if (ht[h1 + 1] < ht[h2 + 1]) r = h1;
  else r = h2;
  for (int i = 0; i < 16; ++i) ht[r + i] = 0;
  ht[r] = (u8)chk;
  return r;
}

// All memory a bit decision needs is fetched in ONE round: every address is computed first, then every
// lane issues its loads back to back with no branch in between (a load behind a divergent branch is
// waited for before the next branch may issue its own) -- the table entry of a CM/ICM, the weight pair
// of an ISSE, the weight of a MIX2, the weight each lane contributes to each of the (up to kMaxMix)
// mixers, its entry of the 33-entry row of each of the (up to kMaxSse) SSE stages.  Bit-history buckets
// (ICM/ISSE) are fetched once per nibble -- the three candidate rows of find() in one round -- and the
// chosen 16-byte row then lives in registers.  After the bit is known nothing is loaded: every new
// value is computed from registers and stored without waiting.
constexpr int kMaxMix = 4, kMaxSse = 2;
// Cycle-counter statistics per phase (find / addresses+loads / leaves / dependents / update / vm / total / bytes):
// compiled in with -DZPQ_CM_PROFILE (ZPQ_EXTRA_FLAGS), printed when ZPQ_CM_STATS is set.  Off by default: every
// s_memtime is a scalar memory operation the wave has to wait for.
__device__ unsigned long long g_cm_prof[8];
#ifdef ZPQ_CM_PROFILE
#define CM_TICK(var) unsigned long long var = __builtin_readcyclecounter()
#define CM_TOCK(k, var) { const unsigned long long t_ = __builtin_readcyclecounter(); prof[k] += t_ - var; var = t_; }
#else
#define CM_TICK(var)
#define CM_TOCK(k, var)
#endif
typedef __attribute__((address_space(1))) u64_u g_u64_u;
typedef __attribute__((address_space(1))) u32x4_u g_u32x4_u;

struct WavePred {
  lds_tables* T;
  Comp C;              // this lane's component (type NONE beyond n)
  g_u32* cm; g_u8* ht; g_u16* a16;     // C.cm / C.ht / C.a16 as global pointers
  int p;               // this lane's stretched prediction p[i]
  u32 h;               // this lane's context H[i]
  u32 c8, hmap4;
  u32 va, vb;          // round A: table entry (CM, ICM), weights (ISSE: va, vb), weight (MIX2)
  g_u8* pa;            // where va came from
  u32 row[4];          // ICM/ISSE: the bucket of this nibble (ht[C.c .. C.c+15])
  u32 st;              // ICM/ISSE: bit-history state of this bit
  u32 mbyte;           // MATCH: the predicted byte
  u32 mcur;            // MATCH: the byte being assembled
  int wm[kMaxMix]; g_u32* pm[kMaxMix];        // this lane's weight in mixer slot k, and its address
  u32 se[kMaxSse];                            // lane l holds entry l of SSE slot k's row
  unsigned long long dep;   // components whose prediction depends on earlier ones, in order
  g_u8* dummy;         // any readable 64 bytes in HBM
  int lane;
  u32 n_last;          // index of the last component (the model's output)
  int vmerr;           // HCOMP machine fault (uniform)
  unsigned long long prof[6];

  __device__ __forceinline__ int squash(int x) const { return T->squash[x + 2048]; }
  __device__ __forceinline__ int stretch(u32 x) const { return T->stretch[x]; }
  __device__ __forceinline__ static u32 row_get(const u32 (&r)[4], u32 idx) {
    const u32 w = idx < 4 ? r[0] : idx < 8 ? r[1] : idx < 12 ? r[2] : r[3];
    return (w >> (8 * (idx & 3))) & 255u;
  }
  __device__ __forceinline__ static void row_set(u32 (&r)[4], u32 idx, u32 v) {
    // value selects only: a select between ADDRESSES of the four words would push the whole predictor
    // out of registers into scratch memory
    const u32 sh = 8 * (idx & 3), m = ~(255u << sh), x = v << sh, k = idx >> 2;
    const u32 n0 = (r[0] & m) | x, n1 = (r[1] & m) | x, n2 = (r[2] & m) | x, n3 = (r[3] & m) | x;
    r[0] = k == 0 ? n0 : r[0]; r[1] = k == 1 ? n1 : r[1]; r[2] = k == 2 ? n2 : r[2]; r[3] = k == 3 ? n3 : r[3];
  }
  __device__ __forceinline__ static g_u32* lane_ptr(g_u32* q, int i) {      // lane i's pointer, on every lane
    return (g_u32*)(((u64)rlu((u32)((u64)q >> 32), i) << 32) | rlu((u32)(u64)q, i));
  }

  // find() (ZSFX/libzpaq.cpp:2064-2080) for every ICM/ISSE lane at once: the three candidate buckets
  // arrive together, the choice and the replacement are register work, a replaced bucket is stored
  __device__ __forceinline__ void find_rows() {
    const bool mine = C.type == ICM || C.type == ISSE;
    const u32 cxt = h + 16 * c8;
    const u32 chk = (cxt >> (C.a1 + 2)) & 255;
    const u32 h0 = (cxt * 16) & (C.ht_mask + 1 - 16), h1 = h0 ^ 16, h2 = h0 ^ 32;
    g_u8* base = mine ? ht : dummy;
    const u32x4 r0 = *(const g_u32x4_u*)(base + (mine ? h0 : 0));
    const u32x4 r1 = *(const g_u32x4_u*)(base + (mine ? h1 : 16));
    const u32x4 r2 = *(const g_u32x4_u*)(base + (mine ? h2 : 32));
    if (!mine) return;
    u32 sel;
    bool fresh = false;
    if ((r0.x & 255) == chk) sel = 0;
    else if ((r1.x & 255) == chk) sel = 1;
    else if ((r2.x & 255) == chk) sel = 2;
    else {
      const u32 q0 = (r0.x >> 8) & 255, q1 = (r1.x >> 8) & 255, q2 = (r2.x >> 8) & 255;
      sel = (q0 <= q1 && q0 <= q2) ? 0 : (q1 < q2 ? 1 : 2);
      fresh = true;
    }
    C.c = sel == 0 ? h0 : sel == 1 ? h1 : h2;
    if (fresh) {
      row[0] = chk; row[1] = row[2] = row[3] = 0;
      u32x4 z; z.x = chk; z.y = z.z = z.w = 0;
      *(g_u32x4_u*)(ht + C.c) = z;
    } else {
      const u32x4 r = sel == 0 ? r0 : sel == 1 ? r1 : r2;
      row[0] = r.x; row[1] = r.y; row[2] = r.z; row[3] = r.w;
    }
  }

  __device__ __forceinline__ int predict() {            // predict0, ZSFX/libzpaq.cpp:1846-1943
    CM_TICK(tq);
    if (c8 == 1 || (c8 & 0xf0) == 16) find_rows();
    CM_TOCK(0, tq)
    // ---- every address first ---------------------------------------------------------------------
    pa = dummy;
    switch (C.type) {
      case CM: C.cxt = h ^ hmap4; pa = (g_u8*)&cm[C.cxt & C.cm_mask]; break;
      case ICM: st = row_get(row, hmap4 & 15); C.cxt = st; pa = (g_u8*)&cm[st & C.cm_mask]; break;
      case ISSE: st = row_get(row, hmap4 & 15); C.cxt = st; pa = (g_u8*)&cm[st * 2]; break;
      case MIX2: C.cxt = (h + (c8 & C.a5)) & (C.c - 1); pa = (g_u8*)&a16[C.cxt]; break;
      case MIX: C.cxt = ((h + (c8 & C.a5)) & (C.c - 1)) * C.a3; break;
      case SSE: C.cxt = (h + c8) * 32; break;
      default: break;
    }
    g_u32* ps[kMaxSse];
#pragma unroll
    for (int s = 0; s < kMaxMix; ++s) pm[s] = (g_u32*)dummy;
#pragma unroll
    for (int s = 0; s < kMaxSse; ++s) ps[s] = (g_u32*)dummy;
    {
      int kmix = 0, ksse = 0;
      for (unsigned long long m = dep; m; m &= m - 1) {
        const int i = __builtin_ctzll(m);
        const u32 t = rlu(C.type, i);
        if (t == MIX) {
          const int j = (int)rlu(C.a2, i), m_in = (int)rlu(C.a3, i);
          const int k = lane - j;
          g_u32* q = lane_ptr(cm, i) + rlu(C.cxt, i) + (u32)((k >= 0 && k < m_in) ? k : 0);
#pragma unroll
          for (int s = 0; s < kMaxMix; ++s) if (s == kmix) pm[s] = q;
          ++kmix;
        } else if (t == SSE) {
          g_u32* q = lane_ptr(cm, i) + ((rlu(C.cxt, i) + (u32)(lane < 33 ? lane : 32)) & rlu(C.cm_mask, i));
#pragma unroll
          for (int s = 0; s < kMaxSse; ++s) if (s == ksse) ps[s] = q;
          ++ksse;
        }
      }
    }
    // ---- round A: every load of this bit, back to back ---------------------------------------------
    {
      const u64 two = *(const g_u64_u*)pa;
      u32 t0[kMaxMix], t1[kMaxSse];
#pragma unroll
      for (int s = 0; s < kMaxMix; ++s) t0[s] = *pm[s];
#pragma unroll
      for (int s = 0; s < kMaxSse; ++s) t1[s] = *ps[s];
      va = (u32)two; vb = (u32)(two >> 32);
#pragma unroll
      for (int s = 0; s < kMaxMix; ++s) wm[s] = (int)t0[s];
#pragma unroll
      for (int s = 0; s < kMaxSse; ++s) se[s] = t1[s];
    }
    CM_TOCK(1, tq)
    // ---- leaves --------------------------------------------------------------------------------------
    switch (C.type) {
      case CM: p = stretch(va >> 17); break;
      case ICM: p = stretch(va >> 8); break;
      case MATCH:
        if (C.a == 0) p = 0;
        else {
          C.c = (mbyte >> (7 - C.cxt)) & 1;
          p = stretch((u32)(T->dt2k[C.a] * ((int)C.c * -2 + 1)) & 32767u);
        }
        break;
      case MIX2: va &= 0xffffu; break;    // 2-byte entry (the load started at its address)
      default: break;
    }
    CM_TOCK(2, tq)
    // ---- dependent components, in index order (registers only) ---------------------------------------
    int kmix = 0, ksse = 0;
    for (unsigned long long m = dep; m; m &= m - 1) {
      const int i = __builtin_ctzll(m);
      const u32 t = rlu(C.type, i);
      const int j = (int)rlu(C.a2, i);
      if (t == ISSE) {
        const int r = clamp2k(((int)rlu(va, i) * rl(p, j) + (int)rlu(vb, i) * 64) >> 16);
        if (lane == i) p = r;
      } else if (t == MIX) {
        const int m_in = (int)rlu(C.a3, i);
        int w = 0;
#pragma unroll
        for (int s = 0; s < kMaxMix; ++s) if (s == kmix) w = wm[s];
        ++kmix;
        const int term = (lane >= j && lane < j + m_in) ? (w >> 8) * p : 0;
        const int sum = rl(wave_sum_to_last(term), 63);
        if (lane == i) p = clamp2k(sum >> 8);
      } else if (t == MIX2) {
        const int w = (int)rlu(va, i);
        const int r = (w * rl(p, j) + (65536 - w) * rl(p, (int)rlu(C.a3, i))) >> 16;
        if (lane == i) p = r;
      } else if (t == AVG) {
        const int a1 = (int)rlu(C.a1, i), wgt = (int)rlu(C.a3, i);
        const int r = (rl(p, a1) * wgt + rl(p, j) * (256 - wgt)) >> 8;
        if (lane == i) p = r;
      } else if (t == SSE) {
        u32 e = 0;
#pragma unroll
        for (int s = 0; s < kMaxSse; ++s) if (s == ksse) e = se[s];
        ++ksse;
        int pq = rl(p, j) + 992;
        if (pq < 0) pq = 0;
        if (pq > 1983) pq = 1983;
        const int wt = pq & 63;
        pq >>= 6;
        const u32 e0 = rlu(e, pq), e1 = rlu(e, pq + 1);
        if (lane == i) {
          C.cxt += (u32)pq;
          p = stretch(((e0 >> 10) * (u32)(64 - wt) + (e1 >> 10) * (u32)wt) >> 13);
          C.cxt += (u32)(wt >> 5);
          va = (wt >> 5) ? e1 : e0;                    // the entry train() will touch
        }
      }
    }
    CM_TOCK(3, tq)
    return squash(rl(p, (int)n_last));
  }

  __device__ __forceinline__ u32 trained(u32 cur, int y, u32 limit) const {   // train(), ZSFX/libzpaq.h:1151-1157 (wraps in 32 bits)
    const u32 count = cur & 0x3ff;
    const int error = y * 32767 - (int)(cur >> 17);
    return cur + (((u32)error * (u32)T->dt[count] & 0xfffffc00u) + (count < limit));
  }

  __device__ __forceinline__ void update(int y, Vm& z) {   // update0, ZSFX/libzpaq.cpp:1946-2058 -- stores only
    const __attribute__((address_space(3))) u8* ns = T->ns;
    CM_TICK(tq);
    int kmix = 0;
    for (unsigned long long m = dep; m; m &= m - 1) {
      const int i = __builtin_ctzll(m);
      if (rlu(C.type, i) != MIX) continue;
      const int j = (int)rlu(C.a2, i), m_in = (int)rlu(C.a3, i);
      const int err = ((y * 32767 - squash(rl(p, i))) * (int)rlu(C.a4, i)) >> 4;
      int w = 0; g_u32* q = (g_u32*)dummy;
#pragma unroll
      for (int s = 0; s < kMaxMix; ++s) if (s == kmix) { w = wm[s]; q = pm[s]; }
      ++kmix;
      if (lane >= j && lane < j + m_in) *q = (u32)clamp512k(w + ((err * p + (1 << 12)) >> 13));
    }
    // predictions of this lane's input components, fetched with every lane active
    const int pa2 = __shfl(p, (int)(C.a2 & 63)), pa3 = __shfl(p, (int)(C.a3 & 63));
    switch (C.type) {
      case CM: *(g_u32*)pa = trained(va, y, C.limit); break;
      case ICM: {
        const u32 nst = ns[st * 4 + y];
        row_set(row, hmap4 & 15, nst);
        ht[C.c + (hmap4 & 15)] = (u8)nst;
        *(g_u32*)pa = va + (u32)((int)(y * 32767 - (int)(va >> 8)) >> 2);
      } break;
      case MATCH: {
        if ((int)C.c != y) C.a = 0;
        mcur = (mcur + mcur + (u32)y) & 255u;
        if (++C.cxt == 8) {
          ht[C.limit & C.ht_mask] = (u8)mcur;
          mcur = 0;
          C.cxt = 0;
          ++C.limit;
          C.limit &= (1u << C.a2) - 1;
          if (C.a == 0) {
            C.b = C.limit - cm[h & C.cm_mask];
            if (C.b & C.ht_mask)
              while (C.a < 255 && ht[(C.limit - C.a - 1) & C.ht_mask] == ht[(C.limit - C.a - C.b - 1) & C.ht_mask]) ++C.a;
          } else C.a += C.a < 255;
          cm[h & C.cm_mask] = C.limit;
          mbyte = ht[(C.limit - C.b) & C.ht_mask];
        }
      } break;
      case MIX2: {
        const int err = ((y * 32767 - squash(p)) * (int)C.a4) >> 5;
        int w = (int)va;
        w += (err * (pa2 - pa3) + (1 << 12)) >> 13;
        if (w < 0) w = 0;
        if (w > 65535) w = 65535;
        a16[C.cxt] = (u16)w;
      } break;
      case ISSE: {
        const int err = y * 32767 - squash(p);
        g_u32* wt = (g_u32*)pa;
        wt[0] = (u32)clamp512k((int)va + ((err * pa2 + (1 << 12)) >> 13));
        wt[1] = (u32)clamp512k((int)vb + ((err + 16) >> 5));
        const u32 nst = ns[st * 4 + y];
        row_set(row, hmap4 & 15, nst);
        ht[C.c + (hmap4 & 15)] = (u8)nst;
      } break;
      case SSE: cm[C.cxt & C.cm_mask] = trained(va, y, C.limit); break;
      default: break;
    }
    CM_TOCK(4, tq)
    c8 += c8 + (u32)y;
    if (c8 >= 256) {
      if (lane == 0) vm_run(z, c8 - 256);
      __builtin_amdgcn_wave_barrier();
      __threadfence_block();
      vmerr = __builtin_amdgcn_readfirstlane(z.err);
      hmap4 = 1;
      c8 = 1;
      h = ((volatile u32*)z.H)[(u32)lane & z.hmask];
      CM_TOCK(5, tq)
    } else if (c8 >= 16 && c8 < 32) hmap4 = (hmap4 & 0xf) << 5 | (u32)y << 4 | 1;
    else hmap4 = (hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2 + (u32)y) & 0xf);
  }
};

__global__ __launch_bounds__(64) void cm_wave_kernel(CmJobDev* jobs, int encode) {
  CmJobDev& J = jobs[blockIdx.x];
  const int lane = (int)threadIdx.x;
  WavePred pr;
  pr.lane = lane; pr.c8 = 1; pr.hmap4 = 1; pr.va = 0; pr.vb = 0; pr.h = 0;
  pr.dep = J.dep; pr.n_last = J.n - 1;
  pr.dummy = (g_u8*)J.T; pr.pa = pr.dummy; pr.st = 0; pr.mcur = 0;
  pr.row[0] = pr.row[1] = pr.row[2] = pr.row[3] = 0;
#pragma unroll
  for (int k = 0; k < kMaxMix; ++k) { pr.wm[k] = 0; pr.pm[k] = (g_u32*)pr.dummy; }
#pragma unroll
  for (int k = 0; k < kMaxSse; ++k) pr.se[k] = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) pr.prof[k] = 0;
  CM_TICK(tk_);
  if ((u32)lane < J.n) { pr.C = J.comp[lane]; pr.p = J.p[lane]; }
  else { memset(&pr.C, 0, sizeof pr.C); pr.p = 0; }
  pr.cm = (g_u32*)pr.C.cm; pr.ht = (g_u8*)pr.C.ht; pr.a16 = (g_u16*)pr.C.a16;
  pr.mbyte = pr.C.type == MATCH ? pr.ht[0] : 0;      // (limit - b) & mask == 0 at the start
  // the HCOMP machine runs once per byte on lane 0: its program and (when small) its H[] live in LDS
  __shared__ u8 s_prog[4096];
  __shared__ u32 s_H[1024];
  __shared__ Tables s_T;          // squash/stretch/dt/dt2k/ns: 86 KiB, read several times per component and bit
  {
    const u32* src = (const u32*)J.T; u32* dst = (u32*)&s_T;
    for (u32 i = (u32)lane; i < sizeof(Tables) / 4; i += 64) dst[i] = src[i];
    pr.T = (lds_tables*)&s_T;
  }
  Vm z = J.vm;
  if (z.plen <= sizeof s_prog) {
    for (u32 i = (u32)lane; i < z.plen; i += 64) s_prog[i] = z.prog[i];
    z.prog = s_prog; z.in_lds |= 1;
  }
  if (z.hmask < 1024) {
    for (u32 i = (u32)lane; i <= z.hmask; i += 64) s_H[i] = 0;      // H[] starts zeroed (ZPAQL::init)
    z.H = s_H; z.in_lds |= 2;
  }
  __syncthreads();
  pr.vmerr = 0;
  u32 low = 1, high = 0xffffffffu, op = 0;
  int status = ZPQ_OK;
  if (encode) {
    auto put = [&](u32 c) { if (lane == 0 && op < J.out_cap) J.out[op] = (u8)c; ++op; };
    auto enc = [&](int y, u32 p) {
      const u32 mid = low + (u32)(((u64)(high - low) * p) >> 16);
      if (y) high = mid; else low = mid + 1;
      while ((high ^ low) < 0x1000000u) { put(high >> 24); high = high << 8 | 255; low <<= 8; low += (low == 0); }
    };
    const u32 nseg = J.seg ? J.nseg : 1u;         // segments: as in cm_code_kernel
    u32 i = 0;
    for (u32 s = 0; s < nseg && !pr.vmerr; ++s) {
      const u32 send = J.seg ? i + J.seg[s] : J.in_len;
      for (; i < send && i < J.in_len && !pr.vmerr; ++i) {
        const u32 c = J.in[i];
        enc(0, 0);
        for (int b = 7; b >= 0; --b) {
          const u32 p = (u32)pr.predict() * 2 + 1;
          const int y = (int)((c >> b) & 1);
          enc(y, p);
          pr.update(y, z);
        }
      }
      enc(1, 0);
      put(0); put(0); put(0); put(0);
      if (J.seg && lane == 0) J.seg[nseg + s] = op;
    }
    if (op > J.out_cap) status = ZPQ_ERR_CAPACITY;
  } else {
    u32 ip = 0, curr = 0, send = J.in_len;
    bool bad = false;
    auto get = [&]() -> u32 { if (ip < send) return J.in[ip++]; bad = true; return 0; };
    auto dec = [&](u32 p) -> int {               // Decoder::decode, ZSFX/libzpaq.cpp:2096-2114
      if (curr < low || curr > high) { bad = true; return 1; }
      const u32 mid = low + (u32)(((u64)(high - low) * p) >> 16);
      int y;
      if (curr <= mid) { y = 1; high = mid; } else { y = 0; low = mid + 1; }
      while ((high ^ low) < 0x1000000u) { high = high << 8 | 255; low <<= 8; low += (low == 0); curr = curr << 8 | get(); }
      return y;
    };
    const u32 nseg = J.seg ? J.nseg : 1u;         // segments: as in cm_code_kernel
    for (u32 s = 0; s < nseg; ++s) {
      if (J.seg) { send = ip + J.seg[s]; if (send > J.in_len) send = J.in_len; }
      curr = 0;
      for (int i = 0; i < 4; ++i) curr = curr << 8 | get();
      while (!bad && !pr.vmerr) {
        if (dec(0)) { if (curr != 0) bad = true; break; }
        u32 c = 1;
        while (c < 256) {
          const u32 p = (u32)pr.predict() * 2 + 1;
          c += c + (u32)dec(p);
          pr.update((int)(c & 1), z);
        }
        if (op >= J.out_cap) { status = ZPQ_ERR_CAPACITY; break; }   // caller asked for a prefix only
        if (lane == 0) J.out[op] = (u8)(c - 256);
        ++op;
      }
      if (bad || pr.vmerr || status != ZPQ_OK) break;
      if (J.seg) { if (ip != send) { bad = true; break; } if (lane == 0) J.seg[nseg + s] = op; }
    }
    if (bad) status = ZPQ_ERR_FORMAT;
  }
  if (pr.vmerr) status = pr.vmerr == 2 ? ZPQ_ERR_LIMIT : ZPQ_ERR_FORMAT;
  if (lane == 0) {
    J.result[0] = op; J.result[1] = (u32)status;
#ifdef ZPQ_CM_PROFILE
#pragma unroll
    for (int k = 0; k < 6; ++k) atomicAdd(&g_cm_prof[k], pr.prof[k]);
    atomicAdd(&g_cm_prof[6], __builtin_readcyclecounter() - tk_);
    atomicAdd(&g_cm_prof[7], (unsigned long long)(encode ? J.in_len : op));
#endif
  }
}

// Component array initialisation (Predictor::init, ZSFX/libzpaq.cpp:1757-1845), parallel over elements.
struct InitJob { u32 kind; u32* cm; u32 count; u32 arg; const Tables* T; u16* a16; };
__global__ __launch_bounds__(256) void cm_init_kernel(const InitJob* jobs) {
  const InitJob J = jobs[blockIdx.y];
  for (u32 j = blockIdx.x * 256u + threadIdx.x; j < J.count; j += gridDim.x * 256u) {
    switch (J.kind) {
      case CM: J.cm[j] = 0x80000000u; break;
      case ICM: { const u8* ns = J.T->ns; J.cm[j] = (u32)(((ns[j * 4 + 3] * 2 + 1) << 22) / (ns[j * 4 + 2] + ns[j * 4 + 3] + 1)); } break;
      case MATCH: ((u8*)J.cm)[j] = 1; break;            // cr.ht(0) = 1
      case MIX2: J.a16[j] = 32768; break;
      case MIX: J.cm[j] = 65536u / J.arg; break;
      case ISSE: {
        const u8* ns = J.T->ns; const u32 s = j >> 1;
        if (j & 1) {
          const int ci = ((ns[s * 4 + 3] * 2 + 1) << 22) / (ns[s * 4 + 2] + ns[s * 4 + 3] + 1);
          J.cm[j] = (u32)clamp512k(J.T->stretch[ci >> 8] * 1024);
        } else J.cm[j] = 1u << 15;
      } break;
      case SSE: J.cm[j] = (u32)J.T->squash[((j & 31) * 64 - 992) + 2048] << 17 | J.arg; break;
      default: break;
    }
  }
}

// Generic post-processor: runs a PCOMP program once per decoded byte and once with 2^32-1 at the end
// (PostProcessor::write state 5, ZSFX/libzpaq.cpp:2221-2224).
// seg (null: one segment of n bytes): u32[nseg] input bytes per segment of the block, then u32[nseg] (result) the output end
// of each: the machine keeps its state from segment to segment (Decompresser::decompress initialises it for the first only).
__global__ __launch_bounds__(64) void pcomp_run_kernel(Vm* vms, const u8* in, u32 n, u32* result, u32* seg, u32 nseg) {
  if (threadIdx.x != 0) return;
  Vm& z = vms[0];
  const u32 ns = seg ? nseg : 1u;
  u32 i = 0;
  for (u32 s = 0; s < ns && !z.err; ++s) {
    const u32 send = seg ? i + seg[s] : n;
    for (; i < send && i < n && !z.err; ++i) vm_run(z, in[i]);
    if (!z.err) vm_run(z, 0xffffffffu);
    if (seg) seg[nseg + s] = z.out_len;
  }
  result[0] = z.out_len;
  result[1] = z.err == 2 ? (u32)ZPQ_ERR_LIMIT : z.err ? (u32)ZPQ_ERR_FORMAT : (z.out_len > z.out_cap ? (u32)ZPQ_ERR_CAPACITY : 0u);
}

typedef zpq_cm_header ParsedHeader;
#define parse_header zpq_cm_parse_header

const Tables* device_tables(zpq_ctx* ctx) {
  Tables* d = (Tables*)zpq_scratch(ctx, 9, sizeof(Tables));
  if (!d) return nullptr;
  // (re)upload every time the slot is fresh; cheap (86 KiB)
  if (hipMemcpyAsync(d, &host_tables(), sizeof(Tables), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return nullptr;
  return d;
}

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// Bytes of model state one block needs: what Predictor::init allocates (ZSFX/libzpaq.cpp:1757-1845) plus the HCOMP
// machine's H and M.  0 + status when a component is larger than this engine supports.
int model_bytes(zpq_ctx* ctx, const ParsedHeader& P, size_t* big) {
  size_t bytes = al256((size_t)4 << P.hh) + al256((size_t)1 << P.hm);
  for (auto& c : P.comps) {
    const u32 sb = c.size() > 1 ? c[1] : 0;
    switch (c[0]) {
      case CM: if (sb > 28) return zpq_fail(ctx, ZPQ_ERR_METHOD, "CM 2^%u too large", sb); bytes += al256((size_t)4 << sb); break;
      case ICM: if (sb > 24) return zpq_fail(ctx, ZPQ_ERR_METHOD, "ICM 2^%u too large", sb); bytes += al256(1024) + al256((size_t)64 << sb); break;
      case MATCH: if (sb > 28 || c[2] > 30) return zpq_fail(ctx, ZPQ_ERR_METHOD, "MATCH too large"); bytes += al256((size_t)4 << sb) + al256((size_t)1 << c[2]); break;
      case MIX2: if (sb > 28) return zpq_fail(ctx, ZPQ_ERR_METHOD, "MIX2 too large"); bytes += al256((size_t)2 << sb); break;
      case MIX: if (sb > 24) return zpq_fail(ctx, ZPQ_ERR_METHOD, "MIX too large"); bytes += al256(((size_t)4 << sb) * c[3]); break;
      case ISSE: if (sb > 24) return zpq_fail(ctx, ZPQ_ERR_METHOD, "ISSE too large"); bytes += al256(2048) + al256((size_t)64 << sb); break;
      case SSE: if (sb > 24) return zpq_fail(ctx, ZPQ_ERR_METHOD, "SSE too large"); bytes += al256((size_t)128 << sb); break;
      default: break;
    }
  }
  *big = bytes;
  return ZPQ_OK;
}
// small per-block records (uploaded in one piece): component descriptors for both kernels, p[], h[], R[], the program
size_t meta_bytes(const ParsedHeader& P) {
  return al256(P.n * sizeof(Comp)) + al256(P.n * sizeof(zpq_spec_comp)) + al256(256 * 4) * 3 + al256(P.hcomp.size() + 8);
}

// One batch: blocks idx[0..nb) of jobs[], all resident at once.
int run_cm_batch(zpq_ctx* ctx, zpq_cm_job* jobs, const std::vector<ParsedHeader>& ph, const size_t* idx, size_t nb, int encode,
                 const Tables* dT) {
  hipStream_t st = ctx->stream;
  const bool trace = getenv("ZPQ_CM_TRACE") != nullptr;
  auto now = [] { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
  const double t_begin = now();
  auto stage = [&](const char* what) { if (trace) { (void)hipStreamSynchronize(st); fprintf(stderr, "[cm trace] %-28s %.3f s\n", what, now() - t_begin); } };
  size_t ncomp_all = 0;
  for (size_t k = 0; k < nb; ++k) ncomp_all += ph[idx[k]].n;
  const bool want_prof = getenv("ZPQ_CM_PROF") != nullptr;
  size_t nseg_all = 0;                                // segment tables of the blocks that have more than one
  for (size_t k = 0; k < nb; ++k) {
    const zpq_cm_job& j = jobs[idx[k]];
    if (j.nseg > 1) {
      if (!j.seg_len || !j.seg_out_end) return zpq_fail(ctx, ZPQ_ERR_ARG, "job %zu: %u segments without seg_len / seg_out_end", idx[k], j.nseg);
      u64 sum = 0;
      for (u32 s = 0; s < j.nseg; ++s) sum += j.seg_len[s];
      if (sum != j.n) return zpq_fail(ctx, ZPQ_ERR_ARG, "job %zu: the segments' lengths do not add up to n", idx[k]);
      nseg_all += j.nseg;
    }
  }
  size_t meta = al256(nb * sizeof(CmJobDev)) + al256(nb * sizeof(zpq_spec_job)) + al256(nb * 8) + al256(ncomp_all * sizeof(InitJob)) + al256(nb * 4) + al256(nb * 64) +
                al256(nseg_all * 8);
  size_t big = 0;
  for (size_t k = 0; k < nb; ++k) {
    size_t b = 0;
    int rc = model_bytes(ctx, ph[idx[k]], &b);
    if (rc) return rc;
    big += b; meta += meta_bytes(ph[idx[k]]);
  }
  u8* arena = (u8*)zpq_scratch(ctx, 0, meta + big + 4096);
  if (!arena) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "cm model memory (%zu MiB)", (meta + big) >> 20);
  stage("arena");
  std::vector<u8> hm(meta, 0);                       // host image of the metadata region
  ZPQ_HIP(ctx, hipMemsetAsync(arena + meta, 0, big, st));
  stage("memset");
  size_t mo = 0;
  u8* bp = arena + meta;
  auto take_meta = [&](size_t x) { const size_t r = mo; mo += al256(x); return r; };
  auto take_big = [&](size_t x) { u8* r = bp; bp += al256(x); return r; };
  const size_t o_jobs = take_meta(nb * sizeof(CmJobDev)), o_sjobs = take_meta(nb * sizeof(zpq_spec_job)), o_res = take_meta(nb * 8),
               o_init = take_meta(ncomp_all * sizeof(InitJob)), o_cnt = take_meta(nb * 4), o_prof = take_meta(nb * 64), o_seg = take_meta(nseg_all * 8);
  size_t seg_at = 0;                                  // u32 words into the segment region
  std::vector<size_t> seg_of(nb, 0);
  CmJobDev* hj = (CmJobDev*)(hm.data() + o_jobs);
  zpq_spec_job* hs = (zpq_spec_job*)(hm.data() + o_sjobs);
  InitJob* hinit = (InitJob*)(hm.data() + o_init);
  size_t ninit = 0;
  u32* d_res = (u32*)(arena + o_res);
  for (size_t k = 0; k < nb; ++k) {
    const size_t i = idx[k];
    const ParsedHeader& P = ph[i];
    CmJobDev& J = hj[k];
    zpq_spec_job& S = hs[k];
    J.n = P.n; J.T = dT;
    const size_t o_comp = take_meta(P.n * sizeof(Comp)), o_scomp = take_meta(P.n * sizeof(zpq_spec_comp)), o_p = take_meta(256 * 4),
                 o_h = take_meta(256 * 4), o_R = take_meta(256 * 4), o_prog = take_meta(P.hcomp.size() + 8);
    J.comp = (Comp*)(arena + o_comp); J.p = (int*)(arena + o_p); J.h = (u32*)(arena + o_h);
    J.vm.R = (u32*)(arena + o_R);
    J.vm.H = (u32*)take_big((size_t)4 << P.hh); J.vm.hmask = (1u << P.hh) - 1;
    J.vm.M = take_big((size_t)1 << P.hm); J.vm.mmask = (1u << P.hm) - 1;
    J.vm.prog = arena + o_prog; J.vm.plen = (u32)P.hcomp.size();
    J.vm.budget = 1u << 20; J.vm.credit = 1u << 24;
    if (!P.hcomp.empty()) memcpy(hm.data() + o_prog, P.hcomp.data(), P.hcomp.size());
    Comp* hc = (Comp*)(hm.data() + o_comp);
    zpq_spec_comp* sc = (zpq_spec_comp*)(hm.data() + o_scomp);
    int* hp = (int*)(hm.data() + o_p);
    for (u32 q = 0; q < P.n; ++q) {
      const std::vector<u8>& c = P.comps[q];
      Comp& C = hc[q];
      C.type = c[0];
      C.a1 = c.size() > 1 ? c[1] : 0; C.a2 = c.size() > 2 ? c[2] : 0; C.a3 = c.size() > 3 ? c[3] : 0;
      C.a4 = c.size() > 4 ? c[4] : 0; C.a5 = c.size() > 5 ? c[5] : 0;
      InitJob in; memset(&in, 0, sizeof in); in.T = dT;
      switch (c[0]) {
        case CONS: hp[q] = ((int)c[1] - 128) * 4; break;
        case CM:
          C.cm = (u32*)take_big((size_t)4 << c[1]); C.cm_mask = (1u << c[1]) - 1; C.limit = c[2] * 4;
          in.kind = CM; in.cm = C.cm; in.count = 1u << c[1];
          break;
        case ICM:
          C.limit = 1023;
          C.cm = (u32*)take_big(1024); C.cm_mask = 255;
          C.ht = take_big((size_t)64 << c[1]); C.ht_mask = (64u << c[1]) - 1;
          in.kind = ICM; in.cm = C.cm; in.count = 256;
          break;
        case MATCH:
          C.cm = (u32*)take_big((size_t)4 << c[1]); C.cm_mask = (1u << c[1]) - 1;
          C.ht = take_big((size_t)1 << c[2]); C.ht_mask = (1u << c[2]) - 1;
          in.kind = MATCH; in.cm = (u32*)C.ht; in.count = 1;                // cr.ht(0)=1
          break;
        case AVG:
          if (c[1] >= q || c[2] >= q) return zpq_fail(ctx, ZPQ_ERR_FORMAT, "AVG input out of range");
          break;
        case MIX2:
          if (c[2] >= q || c[3] >= q) return zpq_fail(ctx, ZPQ_ERR_FORMAT, "MIX2 input out of range");
          C.c = 1u << c[1];
          C.a16 = (u16*)take_big((size_t)2 << c[1]); C.a16_mask = (1u << c[1]) - 1;
          in.kind = MIX2; in.count = 1u << c[1]; in.a16 = C.a16;
          break;
        case MIX:
          if (c[2] >= q || c[3] < 1 || c[3] > q - c[2]) return zpq_fail(ctx, ZPQ_ERR_FORMAT, "MIX inputs out of range");
          C.c = 1u << c[1];
          C.cm = (u32*)take_big(((size_t)4 << c[1]) * c[3]); C.cm_mask = 0xffffffffu;
          in.kind = MIX; in.cm = C.cm; in.count = (1u << c[1]) * c[3]; in.arg = c[3];
          break;
        case ISSE:
          if (c[2] >= q) return zpq_fail(ctx, ZPQ_ERR_FORMAT, "ISSE input out of range");
          C.ht = take_big((size_t)64 << c[1]); C.ht_mask = (64u << c[1]) - 1;
          C.cm = (u32*)take_big(2048); C.cm_mask = 511;
          in.kind = ISSE; in.cm = C.cm; in.count = 512;
          break;
        case SSE:
          if (c[2] >= q || c[3] > c[4] * 4) return zpq_fail(ctx, ZPQ_ERR_FORMAT, "SSE arguments out of range");
          C.cm = (u32*)take_big((size_t)128 << c[1]); C.cm_mask = (32u << c[1]) - 1; C.limit = c[4] * 4;
          in.kind = SSE; in.cm = C.cm; in.count = 32u << c[1]; in.arg = c[3];
          break;
        default: break;
      }
      if (in.kind) hinit[ninit++] = in;
      zpq_spec_comp& Z = sc[q];
      Z.cm = (u64)(uintptr_t)(c[0] == MIX2 ? (void*)C.a16 : (void*)C.cm); Z.ht = (u64)(uintptr_t)C.ht;
      Z.type = C.type; Z.a1 = C.a1; Z.a2 = C.a2; Z.a3 = C.a3; Z.a4 = C.a4; Z.a5 = C.a5;
      Z.limit = C.limit; Z.cm_mask = C.cm_mask; Z.ht_mask = C.ht_mask; Z.csize = C.c;
    }
    J.dep = 0;
    if (P.n <= 64)
      for (u32 q = 0; q < P.n; ++q) {
        const u32 t = P.comps[q][0];
        if (t == AVG || t == MIX2 || t == MIX || t == ISSE || t == SSE) J.dep |= 1ull << q;
      }
    J.in = jobs[i].d_in; J.in_len = jobs[i].n;
    J.out = jobs[i].d_out; J.out_cap = jobs[i].out_cap;
    J.result = d_res + 2 * k;
    S.comp = (u64)(uintptr_t)(arena + o_scomp); S.p0 = (u64)(uintptr_t)J.p; S.H = (u64)(uintptr_t)J.vm.H; S.M = (u64)(uintptr_t)J.vm.M;
    S.R = (u64)(uintptr_t)J.vm.R; S.in = (u64)(uintptr_t)J.in; S.out = (u64)(uintptr_t)J.out; S.result = (u64)(uintptr_t)J.result;
    S.in_len = J.in_len; S.out_cap = J.out_cap;
    S.prof = want_prof ? (u64)(uintptr_t)(arena + o_prof + 64 * k) : 0;
    J.seg = nullptr; J.nseg = 0; S.seg = 0; S.nseg = 0;
    if (jobs[i].nseg > 1) {
      u32* hseg = (u32*)(hm.data() + o_seg) + seg_at;
      for (u32 s = 0; s < jobs[i].nseg; ++s) hseg[s] = jobs[i].seg_len[s];
      J.seg = (u32*)(arena + o_seg) + seg_at; J.nseg = jobs[i].nseg;
      S.seg = (u64)(uintptr_t)J.seg; S.nseg = J.nseg;
      seg_of[k] = seg_at;
      seg_at += 2 * (size_t)jobs[i].nseg;
    }
  }
  // which kernel codes which block: blocks sharing a header share one specialised kernel; the rest take the generic ones
  std::vector<int> group(nb, -1);
  std::vector<zpq_cm_spec*> gk;
  std::vector<std::vector<size_t>> members;
  const bool want_spec = getenv("ZPQ_CM_GENERIC") == nullptr;
  {
    std::map<std::string, int> by_header;
    for (size_t k = 0; k < nb; ++k) {
      const size_t i = idx[k];
      if (!want_spec || ph[i].n > 64) continue;
      const std::string key((const char*)jobs[i].header, (size_t)(2 + (jobs[i].header[0] | jobs[i].header[1] << 8)));
      auto it = by_header.find(key);
      if (it == by_header.end()) {
        zpq_cm_spec* sk = nullptr;
        int g = -1;
        if (zpq_cm_spec_get(ctx, ph[i], encode != 0, &sk) == ZPQ_OK && sk) { g = (int)gk.size(); gk.push_back(sk); members.emplace_back(); }
        else {
          static bool warned = false;
          if (!warned) { warned = true; fprintf(stderr, "[zpaqhip] specialised context-mixing coder unavailable (%s): generic kernel used\n", ctx->err.c_str()); }
        }
        it = by_header.insert({key, g}).first;
      }
      group[k] = it->second;
      if (group[k] >= 0) members[group[k]].push_back(k);
    }
  }
  // the specialised kernels take their blocks from a queue: order each group's records contiguously
  std::vector<zpq_spec_job> ordered;
  std::vector<size_t> gstart(gk.size() + 1, 0);
  for (size_t g = 0; g < gk.size(); ++g) {
    gstart[g] = ordered.size();
    // longest first: the last block to finish decides the launch
    std::stable_sort(members[g].begin(), members[g].end(), [&](size_t x, size_t y) { return hs[x].in_len > hs[y].in_len; });
    for (size_t k : members[g]) ordered.push_back(hs[k]);
  }
  gstart[gk.size()] = ordered.size();
  if (!ordered.empty()) memcpy(hs, ordered.data(), ordered.size() * sizeof(zpq_spec_job));
  std::vector<CmJobDev> generic;
  for (size_t k = 0; k < nb; ++k) if (group[k] < 0) generic.push_back(hj[k]);
  if (!generic.empty()) memcpy(hj, generic.data(), generic.size() * sizeof(CmJobDev));
  stage("kernels ready");
  ZPQ_HIP(ctx, hipMemcpyAsync(arena, hm.data(), meta, hipMemcpyHostToDevice, st));
  for (size_t q = 0; q < ninit; q += 32768) {
    const size_t m = ninit - q < 32768 ? ninit - q : 32768;
    ZPQ_LAUNCH(ctx, "cm_init_kernel", st, cm_init_kernel, dim3(64, (unsigned)m), dim3(256), (const InitJob*)(arena + o_init) + q);
    ZPQ_HIP(ctx, hipGetLastError());
  }
  stage("tables initialised");
  for (size_t g = 0; g < gk.size(); ++g) {
    int rc = zpq_cm_spec_launch(ctx, gk[g], st, arena + o_sjobs + gstart[g] * sizeof(zpq_spec_job), (u32)(gstart[g + 1] - gstart[g]),
                                (u32*)(arena + o_cnt) + g, dT, encode);
    if (rc) return rc;
  }
  if (!generic.empty()) {
    bool wave_ok = getenv("ZPQ_CM_ONE_LANE") == nullptr;
    for (size_t k = 0; k < nb; ++k) {
      if (group[k] >= 0) continue;
      int nm = 0, nsse = 0;
      for (auto& c : ph[idx[k]].comps) { nm += c[0] == MIX; nsse += c[0] == SSE; }
      if (ph[idx[k]].n > 64 || nm > kMaxMix || nsse > kMaxSse) wave_ok = false;
    }
    CmJobDev* d_jobs = (CmJobDev*)(arena + o_jobs);
    if (wave_ok) ZPQ_LAUNCH(ctx, "cm_wave_kernel", st, cm_wave_kernel, dim3((unsigned)generic.size()), dim3(64), d_jobs, encode);
    else ZPQ_LAUNCH(ctx, "cm_code_kernel", st, cm_code_kernel, dim3((unsigned)generic.size()), dim3(64), d_jobs, encode);
    ZPQ_HIP(ctx, hipGetLastError());
  }
  std::vector<u32> res(nb * 2);
  if (const char* wd = getenv("ZPQ_CM_WATCHDOG")) {
    // diagnostic: give the coders wd seconds, then report how far each block got (the kernels store their byte
    // position in result[0] as they go when built with progress marks) and give up on the process
    const double limit = atof(wd), t0 = now();
    while (hipStreamQuery(st) == hipErrorNotReady && now() - t0 < limit) usleep(20000);
    if (hipStreamQuery(st) == hipErrorNotReady) {
      hipStream_t s2; (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
      (void)hipMemcpyAsync(res.data(), d_res, nb * 8, hipMemcpyDeviceToHost, s2);
      (void)hipStreamSynchronize(s2);
      fprintf(stderr, "[cm watchdog] still running after %.1f s; progress words:", limit);
      for (size_t k = 0; k < nb && k < 16; ++k) fprintf(stderr, " %u/%u", res[2 * k], res[2 * k + 1]);
      fprintf(stderr, "\n");
      fflush(stderr);
      _exit(3);
    }
  }
  ZPQ_HIP(ctx, hipMemcpyAsync(res.data(), d_res, nb * 8, hipMemcpyDeviceToHost, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  if (trace) fprintf(stderr, "[cm trace] %zu blocks (%zu specialised groups, %zu generic), %zu MiB of models: coded after %.3f s\n", nb, gk.size(), generic.size(), big >> 20, now() - t_begin);
  if (want_prof) {
    std::vector<unsigned long long> pr(nb * 8);
    ZPQ_HIP(ctx, hipMemcpy(pr.data(), arena + o_prof, nb * 64, hipMemcpyDeviceToHost));
    unsigned long long t[8] = {0}, bytes = 0;
    for (size_t k = 0; k < nb; ++k) { for (int q = 0; q < 8; ++q) t[q] += pr[8 * k + q]; bytes += encode ? jobs[idx[k]].n : res[2 * k]; }
    if (bytes) fprintf(stderr, "[cm prof] cycles per byte: find %.0f | addresses+loads %.0f | leaves %.0f | chain %.0f | update %.0f | byte boundary %.0f | squash+coder %.0f | total %.0f (%llu bytes)\n",
            (double)t[0] / bytes, (double)t[1] / bytes, (double)t[2] / bytes, (double)t[3] / bytes, (double)t[4] / bytes, (double)t[5] / bytes, (double)t[6] / bytes, (double)t[7] / bytes, bytes);
  }
  if (getenv("ZPQ_CM_STATS")) {
    unsigned long long c[8] = {0};
    (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(g_cm_prof), sizeof c);
    fprintf(stderr, "[cm stats] cumulative cycles: find=%llu loads=%llu leaves=%llu dependents=%llu update=%llu vm=%llu total=%llu bytes=%llu\n",
            c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
  }
  std::vector<u32> segs(2 * nseg_all);
  if (nseg_all) ZPQ_HIP(ctx, hipMemcpy(segs.data(), arena + o_seg, nseg_all * 8, hipMemcpyDeviceToHost));
  for (size_t k = 0; k < nb; ++k) {
    zpq_cm_job& j = jobs[idx[k]];
    j.out_len = res[2 * k];
    j.status = (int32_t)res[2 * k + 1];
    if (j.nseg > 1)                                   // (a block that stopped early leaves the later entries at the bytes produced)
      for (u32 s = 0; s < j.nseg; ++s) { const u32 e = segs[seg_of[k] + j.nseg + s]; j.seg_out_end[s] = j.status == ZPQ_OK || e ? e : j.out_len; }
  }
  return ZPQ_OK;
}

int run_cm(zpq_ctx* ctx, zpq_cm_job* jobs, size_t njobs, int encode) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (njobs == 0) return ZPQ_OK;
  const Tables* dT = device_tables(ctx);
  if (!dT) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "cm tables");
  std::vector<ParsedHeader> ph(njobs);
  std::vector<size_t> need(njobs);
  for (size_t i = 0; i < njobs; ++i) {
    int rc = parse_header(ctx, jobs[i].header, jobs[i].header_len, ph[i]);
    if (rc) return rc;
    if (ph[i].n == 0) return zpq_fail(ctx, ZPQ_ERR_ARG, "job %zu: block has no components (stored mode)", i);
    size_t b = 0;
    rc = model_bytes(ctx, ph[i], &b);
    if (rc) return rc;
    need[i] = b + meta_bytes(ph[i]) + 4096;
  }
  // as many blocks side by side as the models fit into what is free now (less a reserve), batch after batch
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  size_t budget = free_b + ctx->scratch_cap[0];
  budget = budget > total_b / 10 ? budget - total_b / 10 : budget / 2;
  if (const char* e = getenv("ZPQ_CM_BUDGET_MB")) budget = (size_t)atoll(e) << 20;
  std::vector<size_t> order(njobs);
  for (size_t i = 0; i < njobs; ++i) order[i] = i;
  size_t at = 0;
  while (at < njobs) {
    size_t sum = 0, nb = 0;
    while (at + nb < njobs && nb < 16384 && (nb == 0 || sum + need[order[at + nb]] <= budget)) sum += need[order[at + nb++]];
    int rc = run_cm_batch(ctx, jobs, ph, order.data() + at, nb, encode, dT);
    if (rc) return rc;
    at += nb;
  }
  int first = ZPQ_OK;
  for (size_t i = 0; i < njobs; ++i) if (jobs[i].status && !first) first = jobs[i].status;
  return first;
}

}  // namespace

// Block header bytes starting at hsize[2] (ZPAQL::read, ZSFX/libzpaq.cpp:879-921)
int zpq_cm_parse_header(zpq_ctx* ctx, const u8* h, u32 len, zpq_cm_header& P) {
  static const int kCompSz[10] = {0, 2, 3, 2, 3, 4, 6, 6, 3, 5};   // ZSFX/libzpaq.cpp:706
  if (len < 9) return zpq_fail(ctx, ZPQ_ERR_FORMAT, "header too short");
  const u32 hsize = h[0] | (u32)h[1] << 8;
  if (hsize + 2 > len) return zpq_fail(ctx, ZPQ_ERR_FORMAT, "header truncated");
  P.hh = h[2]; P.hm = h[3]; P.ph = h[4]; P.pm = h[5]; P.n = h[6];
  P.comps.clear(); P.hcomp.clear();
  if (P.hh > 24 || P.hm > 28) return zpq_fail(ctx, ZPQ_ERR_METHOD, "H/M of 2^%u/2^%u too large for this engine", P.hh, P.hm);
  u32 p = 7;
  for (u32 i = 0; i < P.n; ++i) {
    if (p >= hsize + 2) return zpq_fail(ctx, ZPQ_ERR_FORMAT, "COMP overflows header");
    const u32 t = h[p];
    if (t < 1 || t > 9) return zpq_fail(ctx, ZPQ_ERR_FORMAT, "invalid component type %u", t);
    if (p + kCompSz[t] > hsize + 2) return zpq_fail(ctx, ZPQ_ERR_FORMAT, "COMP overflows header");
    P.comps.push_back(std::vector<u8>(h + p, h + p + kCompSz[t]));
    p += kCompSz[t];
  }
  if (p + 1 >= hsize + 2) return zpq_fail(ctx, ZPQ_ERR_FORMAT, "COMP fills the header: no COMP END / HCOMP END");
  if (h[p++] != 0) return zpq_fail(ctx, ZPQ_ERR_FORMAT, "missing COMP END");
  if (hsize + 2 < p + 1 || h[hsize + 1] != 0) return zpq_fail(ctx, ZPQ_ERR_FORMAT, "missing HCOMP END");
  P.hcomp.assign(h + p, h + hsize + 1);
  return ZPQ_OK;
}

extern "C" {

int zpq_cm_tables(uint16_t* squash, int16_t* stretch, int32_t* dt, int32_t* dt2k, uint8_t* ns) {
  const Tables& T = host_tables();
  memcpy(squash, T.squash, sizeof T.squash);
  memcpy(stretch, T.stretch, sizeof T.stretch);
  memcpy(dt, T.dt, sizeof T.dt);
  memcpy(dt2k, T.dt2k, sizeof T.dt2k);
  memcpy(ns, T.ns, sizeof T.ns);
  return ZPQ_OK;
}

int zpq_cm_encode_dev(zpq_ctx* ctx, zpq_cm_job* jobs, size_t njobs) { return run_cm(ctx, jobs, njobs, 1); }
int zpq_cm_decode_dev(zpq_ctx* ctx, zpq_cm_job* jobs, size_t njobs) { return run_cm(ctx, jobs, njobs, 0); }

int zpq_pcomp_run_dev(zpq_ctx* ctx, const uint8_t* pcomp, uint32_t psize, uint32_t ph, uint32_t pm, const uint8_t* d_in,
                      uint32_t n, uint8_t* d_out, uint32_t out_cap, uint32_t* out_len) {
  return zpq_pcomp_run_segments_dev(ctx, pcomp, psize, ph, pm, d_in, n, nullptr, 0, d_out, out_cap, nullptr, out_len);
}

int zpq_pcomp_run_segments_dev(zpq_ctx* ctx, const uint8_t* pcomp, uint32_t psize, uint32_t ph, uint32_t pm, const uint8_t* d_in,
                               uint32_t n, const uint32_t* seg_len, uint32_t nseg, uint8_t* d_out, uint32_t out_cap,
                               uint32_t* seg_out_end, uint32_t* out_len) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (ph > 24 || pm > 30) return zpq_fail(ctx, ZPQ_ERR_METHOD, "PCOMP memory 2^%u/2^%u too large", ph, pm);
  if (nseg > 1) {
    if (!seg_len || !seg_out_end) return zpq_fail(ctx, ZPQ_ERR_ARG, "%u segments without seg_len / seg_out_end", nseg);
    u64 sum = 0;
    for (u32 s = 0; s < nseg; ++s) sum += seg_len[s];
    if (sum != n) return zpq_fail(ctx, ZPQ_ERR_ARG, "the segments' lengths do not add up to n");
  } else nseg = 0;
  hipStream_t st = ctx->stream;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t bytes = al(sizeof(Vm)) + al(8) + al(1024) + al((size_t)4 << ph) + al((size_t)1 << pm) + al(psize + 8) + al(8 * (size_t)nseg);
  u8* arena = (u8*)zpq_scratch(ctx, 0, bytes + 1024);
  if (!arena) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "pcomp memory");
  ZPQ_HIP(ctx, hipMemsetAsync(arena, 0, bytes, st));
  u8* ap = arena;
  auto take = [&](size_t x) { u8* r = ap; ap += al(x); return r; };
  Vm* d_vm = (Vm*)take(sizeof(Vm));
  u32* d_res = (u32*)take(8);
  Vm v;
  memset(&v, 0, sizeof v);
  v.R = (u32*)take(1024);
  v.H = (u32*)take((size_t)4 << ph); v.hmask = (1u << ph) - 1;
  v.M = take((size_t)1 << pm); v.mmask = (1u << pm) - 1;
  u8* prog = take(psize + 8);
  u32* d_seg = nseg ? (u32*)take(8 * (size_t)nseg) : nullptr;
  v.prog = prog; v.plen = psize;
  v.out = d_out; v.out_cap = out_cap;
  ZPQ_HIP(ctx, hipMemcpyAsync(prog, pcomp, psize, hipMemcpyHostToDevice, st));
  if (nseg) ZPQ_HIP(ctx, hipMemcpyAsync(d_seg, seg_len, 4 * (size_t)nseg, hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipMemcpyAsync(d_vm, &v, sizeof v, hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  // the program translated to machine code (cm_jit.hip); the interpreter only when that is not to be had
  // (the closing 0 of the stored program is not part of it)
  const u32 plen = psize && pcomp[psize - 1] == 0 ? psize - 1 : psize;
  if (getenv("ZPQ_CM_GENERIC") || zpq_pcomp_spec_run(ctx, st, pcomp, plen, ph, pm, d_in, n, d_out, out_cap, v.H, v.M, v.R, d_res, d_seg, nseg) != ZPQ_OK) {
    ZPQ_LAUNCH(ctx, "pcomp_run_kernel", st, pcomp_run_kernel, dim3(1), dim3(64), d_vm, d_in, n, d_res, d_seg, nseg);
    ZPQ_HIP(ctx, hipGetLastError());
  }
  u32 res[2];
  ZPQ_HIP(ctx, hipMemcpyAsync(res, d_res, 8, hipMemcpyDeviceToHost, st));
  if (nseg) ZPQ_HIP(ctx, hipMemcpyAsync(seg_out_end, d_seg + nseg, 4 * (size_t)nseg, hipMemcpyDeviceToHost, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  *out_len = res[0];
  if (res[1]) return zpq_fail(ctx, (int)res[1], "PCOMP run failed");
  return ZPQ_OK;
}

}  // extern "C"
