// Content-defined fragmenter: the per-byte loop of Jidac::add
//     h = (h + c + 1) * (c == o1[c1] ? 314159265 : 271828182);  o1[c1] = c;  c1 = c;
//     cut when sz >= MAX || (h < 2^(22-fragment) && sz >= MIN) || EOF       (state reset per fragment)
// (reference: zpaqfranz.cpp is absent from the snapshot; algorithm per SURVEY.md Appendix C.4, its
// output records are read back at ZSFX/zsfx.cpp:1463-1500 and pinned by AUTOTEST/sha256.zpaq).
//
// MI355X formulation -- exact, no heuristics.  The recurrence is serial per fragment and "cut k
// decides where fragment k+1 starts" is serial per file, so the parallelism is speculative:
//  1. fragment_spec_kernel: files are split into 256 KiB segments and persistent LANES pull segments
//     from a device counter; a lane fragments its segment from the segment start, as if a fragment
//     began there, and then walks on across the segment end until the crossing fragment is closed.
//     Lane-serial is the instruction-efficient shape for this loop (~12 VALU per byte-step of 64
//     lanes): each lane owns a 256-entry o1[] table in LDS (bank = lane, conflict free) and streams
//     its segment 16 bytes per load, three 64-byte register buffers deep; the 16 LDS read/write pairs
//     of a group are issued back to back, the multiply chain runs on registers, and only a group
//     whose minimum hash falls under the threshold is re-walked byte by byte.
//  2. fragment_stitch_kernel: one wave per file glues the pieces: cuts of segment k from the index
//     where the true chain fell in step with it, its crossing cut(s), then segment k' from the index
//     of that cut in its list ... -- a hand-over happens only at a cut BOTH chains made (same reset
//     state from there), and anything that does not line up (never-synchronising data such as runs
//     of zeros) is re-evaluated exactly by the wave-parallel evaluator below (64 bytes per step:
//     in-window o1 forwarding through 64-bit LDS masks, affine-map scan for the hash).
// Integer-only byte work; traffic = input read ~1.25x (crossing fragments); bound by VALU/LDS issue.
// (tests/cpp/frag_emu.cpp compiles the kernels of this file for the host -- ZPQ_EMU_FRAGMENT_ONLY -- and runs them on the
// fibre emulator against the oracle's cuts)
#ifndef ZPQ_EMU_FRAGMENT_ONLY
#include <algorithm>
#include <stdlib.h>

#include "zpq_internal.h"
#endif

namespace {

// The speculation segment (independent of the fragment size limits) is chosen per call so that the
// resident lanes get one segment each: a lane walks ~10-40 MB/s, so a second, partly filled round of
// segments would cost as much again as the first.
constexpr u64 kSegMin = 1ull << 18, kSegMax = 1ull << 22, kSegGrain = 1ull << 14;   // (64 KiB segments: 19 ms of per-file stitching on a 212 MB call, 4 ms with 256 KiB)
constexpr u64 kNone = ~0ull;

struct FragP {
  u32 minf, maxf, thresh;  // thresh = 2^(22-fragment) or 0 when fragment > 22
  u32 pad;
  u64 seg;                 // speculation segment size of this call
};

struct WaveLds {
  unsigned long long M[256];  // per predecessor value: mask of lanes (this window) having it
  u32 O[64];                  // o1[] packed 4 entries per word
};

template <int CTRL, int ROWS>
__device__ __forceinline__ u32 dpp_mov(u32 old, u32 src) {   // lanes without a source keep `old`
  return (u32)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROWS, 0xf, false);
}

// LDS is addressed through address_space(3) pointers throughout: a generic pointer would turn every
// table access into a flat_* instruction, which also waits on the outstanding global prefetches.
typedef __attribute__((address_space(3))) volatile u8 lds_u8;
typedef __attribute__((address_space(3))) volatile u32 lds_u32;
typedef __attribute__((address_space(3))) volatile unsigned long long lds_u64;

// Wave-parallel exact evaluator: the fragment that starts at S (fresh state); returns the offset E of
// its last byte.  Used where speculation cannot help (chains that never fall in step: periodic or
// constant data, where every byte is predicted and the hash never forgets its start).  256 bytes per
// iteration:
//   * predictions, 64 positions at a time (lane = position): the o1[] entry a position sees is the
//     byte after the nearest earlier position with the same predecessor -- inside the window that is
//     "highest lower lane with my predecessor" (one 64-bit lane mask per predecessor value, built
//     with LDS atomics), before the window it is the table itself;
//   * the hash: byte j is the affine map h -> m_j*h + m_j*(c_j+1); every lane composes the maps of
//     its four consecutive bytes, an inclusive DPP scan composes across the wave, and each lane
//     replays its four positions from its exclusive prefix to test the cut condition.
__device__ __forceinline__ u64 eval_fragment(const u8* __restrict__ data, u64 readable, lds_u64* M, lds_u32* O32, u64 S,
                                             u64 file_end, const FragP P) {
  const u32 lane = (u32)lane_id();
  lds_u8* O = (lds_u8*)O32;
  O32[lane] = 0;                       // fresh o1[]
  __builtin_amdgcn_wave_barrier();
  const u64 lastb = readable ? readable - 1 : 0, lastw = readable >= 4 ? readable - 4 : 0;
  auto ldb = [&](u64 q) -> u32 { return data[q < lastb ? q : lastb]; };
  auto ldw = [&](u64 q) -> u32 { return *(const u32_u*)(data + (q < lastw ? q : lastw)); };
  u64 pos = S;
  u32 hin = 0, c1in = 0;
  u32 nb0 = ldb(pos + lane), nb1 = ldb(pos + 64 + lane), nb2 = ldb(pos + 128 + lane), nb3 = ldb(pos + 192 + lane);
  u32 nw = ldw(pos + 4 * lane);
  for (;;) {
    const u32 bb[4] = {nb0, nb1, nb2, nb3};
    const u32 w = nw;
    nb0 = ldb(pos + 256 + lane); nb1 = ldb(pos + 320 + lane); nb2 = ldb(pos + 384 + lane); nb3 = ldb(pos + 448 + lane);
    nw = ldw(pos + 256 + 4 * lane);
    const u64 q0 = pos + 4ull * lane;
    // Fast path: every byte of the tile is what the table already predicts.  Then the table does not
    // change (each write repeats its entry), in-window forwarding cannot differ from the table, and
    // all flags are "predicted".  This is the steady state of exactly the data that ends up here.
    bool steady = true;
    {
      u32 pp = dpp_mov<0x138, 0xf>(c1in, w >> 24);           // byte before this lane's four
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32 c = (w >> (8 * j)) & 255u;
        if (q0 + j < file_end && (u32)O[pp] != c) steady = false;
        pp = c;
      }
    }
    unsigned long long F[4];
    if (__all(steady)) {
      F[0] = F[1] = F[2] = F[3] = ~0ull;
      c1in = (u32)__builtin_amdgcn_readlane((int)(w >> 24), 63);
    } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const u64 q = pos + 64u * r + lane;
      const bool valid = q < file_end;
      const u32 c = valid ? bb[r] : 0u;
      const u32 p = dpp_mov<0x138, 0xf>(c1in, c);          // wave_shr:1 -- lane 0 keeps the byte before the window
      if (valid) __hip_atomic_fetch_or((__attribute__((address_space(3))) unsigned long long*)(M + p), 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __builtin_amdgcn_wave_barrier();
      const unsigned long long mask = valid ? M[p] : 0ull;
      const unsigned long long lower = mask & ((1ull << lane) - 1ull);
      const int k = lower ? 63 - __builtin_clzll(lower) : 0;
      const u32 pc = (u32)__shfl((int)c, k);
      const u32 pred = lower ? pc : (u32)O[p];
      F[r] = __ballot(valid && c == pred);
      if (valid) M[p] = 0ull;                                // leave the mask table clean for the next window
      if (valid && (mask >> lane) == 1ull) O[p] = (u8)c;    // latest occurrence of p in the window
      __builtin_amdgcn_wave_barrier();
      c1in = (u32)__builtin_amdgcn_readlane((int)c, 63);
    }
    }
    // blocked layout: lane l owns positions pos + 4l .. pos + 4l + 3 (window l >> 4, bits 4(l & 15)...)
    const u32 win = lane >> 4;
    const unsigned long long Fm = win == 0 ? F[0] : win == 1 ? F[1] : win == 2 ? F[2] : F[3];
    const u32 bits = (u32)(Fm >> ((4u * lane) & 63u)) & 15u;
    u32 a[4], b[4];
    u32 ca = 1u, cb = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 c = (w >> (8 * j)) & 255u;
      const u32 m = ((bits >> j) & 1u) ? 314159265u : 271828182u;
      if (q0 + j < file_end) { cb = m * (cb + c + 1u); ca = m * ca; }
      a[j] = ca; b[j] = cb;
    }
    // inclusive scan of (ca, cb) over the wave: hi o lo = (a_hi * a_lo, a_hi * b_lo + b_hi)
#define ZPQ_SCAN_STEP(CTRL, ROWS) { const u32 alo = dpp_mov<CTRL, ROWS>(1u, ca), blo = dpp_mov<CTRL, ROWS>(0u, cb); cb = ca * blo + cb; ca = ca * alo; }
    ZPQ_SCAN_STEP(0x111, 0xf) ZPQ_SCAN_STEP(0x112, 0xf) ZPQ_SCAN_STEP(0x114, 0xf) ZPQ_SCAN_STEP(0x118, 0xf)   // row_shr 1,2,4,8
    ZPQ_SCAN_STEP(0x142, 0xa) ZPQ_SCAN_STEP(0x143, 0xc)                                                        // row_bcast 15, 31
#undef ZPQ_SCAN_STEP
    const u32 ax = dpp_mov<0x138, 0xf>(1u, ca), bx = dpp_mov<0x138, 0xf>(0u, cb);   // exclusive prefix
    const u32 h0 = ax * hin + bx;
    u64 firstq = kNone;
    u32 hlast = h0;
#pragma unroll
    for (int j = 3; j >= 0; --j) {
      const u64 q = q0 + j;
      const u32 h = a[j] * h0 + b[j];
      if (j == 3) hlast = h;
      const u64 sz = q - S + 1;
      if (q < file_end && (sz >= P.maxf || (h < P.thresh && sz >= P.minf) || q + 1 == file_end)) firstq = q;
    }
    const unsigned long long cm = __ballot(firstq != kNone);
    if (cm) {
      const int l0 = __builtin_ctzll(cm);
      const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)firstq, l0);
      const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(firstq >> 32), l0);
      return ((u64)hi << 32) | lo;
    }
    hin = (u32)__builtin_amdgcn_readlane((int)hlast, 63);
    pos += 256;
  }
}

// ---- lane-serial evaluator ------------------------------------------------------------------------
// o1[] of lane l lives at byte ((v>>2)*2 + (l>>5))*128 + (l&31)*4 + (v&3) of a 16 KiB per-wave block:
// every lane of a 32-lane half hits its own bank whatever v is.
struct LaneO1 {
  lds_u8* t;
  u32 lanebase;
  __device__ __forceinline__ u32 addr(u32 v) const { return lanebase + (((v << 6) & 0xFF00u) | (v & 3u)); }
  __device__ __forceinline__ void clear() {
    lds_u32* w = (lds_u32*)(t + lanebase);
#pragma unroll
    for (int a = 0; a < 64; ++a) w[a * 64] = 0;
  }
};
struct LaneState { u32 h, c1, sz; };

__device__ __forceinline__ u32 byte_of(const u32x4& d, int j) {
  const u32 w = j < 4 ? d.x : j < 8 ? d.y : j < 12 ? d.z : d.w;
  return (w >> (8 * (j & 3))) & 255u;
}

// One 16-byte group.  Returns true when a cut happened (pos then points behind the cut and the
// state is reset); otherwise pos advances by 16.
template <class OnCut>
__device__ __forceinline__ bool lane_group(const u32x4 d, u64& pos, const u64 file_end, const FragP& P, LaneO1& o,
                                           LaneState& s, OnCut&& on_cut) {
  u32 c[16], pr[16];
  u32 prev = s.c1;
#pragma unroll
  for (int j = 0; j < 16; ++j) {          // 16 in-order LDS read/write pairs, nothing waits in between
    c[j] = byte_of(d, j);
    const u32 a = o.addr(prev);
    pr[j] = o.t[a];
    o.t[a] = (u8)c[j];
    prev = c[j];
  }
  u32 h = s.h, hmin = 0xffffffffu;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    h = (h + c[j] + 1u) * (c[j] == pr[j] ? 314159265u : 271828182u);
    hmin = hmin < h ? hmin : h;
  }
  const bool maybe = (hmin < P.thresh && s.sz + 16 >= P.minf) || s.sz + 16 >= P.maxf || pos + 16 == file_end;
  if (maybe) {                            // rare: locate the first cut of this group exactly
    u32 hh = s.h, sz = s.sz; int cutj = -1; bool trig = false;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      hh = (hh + c[j] + 1u) * (c[j] == pr[j] ? 314159265u : 271828182u);
      ++sz;
      if (cutj < 0 && (sz >= P.maxf || (hh < P.thresh && sz >= P.minf) || pos + j + 1 == file_end)) {
        cutj = j;
        trig = hh < P.thresh && sz >= P.minf;
      }
    }
    if (cutj >= 0) {
      const u64 E = pos + (u64)cutj;
      on_cut(E, trig);
      o.clear();          // also wipes what the bytes after the cut wrote: they are re-walked from E+1
      s.h = 0; s.c1 = 0; s.sz = 0;
      pos = E + 1;
      return true;
    }
  }
  s.h = h; s.c1 = prev; s.sz += 16; pos += 16;
  return false;
}

// A lane's input stream: the 64 bytes being walked plus the next 128 already in flight (these kernels
// run at 2-3 waves per SIMD -- LDS bound -- so HBM latency has to be hidden by hand).  Three register
// buffers take turns (phase PH walks buffer PH and refills it with the bytes 192 further on), so a
// loaded value is never copied: a register copy of a buffer would make the compiler wait for the load
// it was just issued for.  The steady-state refill is UNCONDITIONAL (clamped address, executed by
// every lane every step): a load behind a branch would force a full drain (s_waitcnt vmcnt(0))
// before older data could be touched.
struct LaneStream {
  u64 at;          // stream offset the buffer of the coming phase starts at; ~0 when nothing usable is loaded
  u64 last;        // highest offset a 16-byte load may start at (readable - 16)
  u32x4 a[4], b[4], c[4];
  __device__ __forceinline__ u32x4 ld(const u8* data, u64 off) const {
    return *(const u32x4_u*)(data + (off < last ? off : last));
  }
};

// Advances this lane's stream by 64 bytes (16 or 1 near the end of [pos, lim)).  Must be called by
// every lane of the wave every step, with PH cycling 0,1,2 (finished lanes pass pos >= lim and only
// take part in the refill).  on_cut(E, trig) is called for every cut (trig: the hash fired, as opposed to
// a cut forced by the size limit or the end of the file).
template <int PH, class OnCut>
__device__ __forceinline__ void lane_step(const u8* __restrict__ data, LaneStream& ls, u64& pos, const u64 lim,
                                          const u64 file_end, const FragP& P, LaneO1& o, LaneState& s, OnCut&& on_cut) {
  u32x4 (&cur)[4] = PH == 0 ? ls.a : PH == 1 ? ls.b : ls.c;
  u32x4 (&nx1)[4] = PH == 0 ? ls.b : PH == 1 ? ls.c : ls.a;
  u32x4 (&nx2)[4] = PH == 0 ? ls.c : PH == 1 ? ls.a : ls.b;
  if (pos + 64 <= lim) {
    if (ls.at != pos) {                               // rare: first step, or right after a cut
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        cur[i] = ls.ld(data, pos + 16 * i); nx1[i] = ls.ld(data, pos + 64 + 16 * i); nx2[i] = ls.ld(data, pos + 128 + 16 * i);
      }
    }
    bool cut = false;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if (!cut) cut = lane_group(cur[g], pos, file_end, P, o, s, on_cut);
    ls.at = cut ? ~0ull : pos;
  } else {
    ls.at = ~0ull;
    if (pos + 16 <= lim) {
      lane_group(*(const u32x4_u*)(data + pos), pos, file_end, P, o, s, on_cut);
    } else if (pos < lim) {   // fewer than 16 bytes left: one byte per call
      const u32 c = data[pos];
      const u32 a = o.addr(s.c1);
      const u32 pr = o.t[a];
      o.t[a] = (u8)c;
      s.h = (s.h + c + 1u) * (c == pr ? 314159265u : 271828182u);
      s.c1 = c; ++s.sz;
      if (s.sz >= P.maxf || (s.h < P.thresh && s.sz >= P.minf) || pos + 1 == file_end) {
        on_cut(pos, s.h < P.thresh && s.sz >= P.minf);
        o.clear();
        s.h = 0; s.c1 = 0; s.sz = 0;
      }
      ++pos;
    }
  }
  // refill the buffer this phase walked with bytes [pos+128, pos+192) of the advanced stream
  // (meaningless but harmless when the stream is out of step; the next call reloads all three)
  const u64 nxt = (ls.at == pos ? pos : 0) + 128;
#pragma unroll
  for (int i = 0; i < 4; ++i) cur[i] = ls.ld(data, nxt + 16 * i);
}

// ---- 1. speculative pass: persistent LANES, each pulling 256 KiB segments from a device counter -------
// A lane fragments its segment from the segment start as if a fragment began there, records the cuts
// that fall inside the segment, and then keeps walking -- its state at the segment end is exactly what
// the seam needs -- until the crossing fragment ends at a place where the next lane could have cut
// too (a hash-triggered cut >= min_fragment behind the boundary; cuts forced by the size limit are walked
// through, at most four cuts).  Crossing fragments are ~64 KiB on
// average but exponentially distributed, hence the dynamic hand-out: a wave never idles behind its
// slowest lane.
constexpr u32 kCrossMax = 4;
struct CrossOut { u64 x[kCrossMax]; u32 n; u32 pad; };
__device__ unsigned long long g_frag_stats[8];   // ZPQ_FRAG_STATS=1: [0] seams in step, [1] cross unusable, [2] lookups missed, [3] exact evals, [4] exact bytes

// A lane whose crossing walk exceeds `budget` bytes parks its state (o1[] table, hash, position) and takes the
// next segment; a second launch (RESUME) of a few waves picks the parked walks up again.  Long crossings are
// rare (exponential tail, plus fragments forced by the size limit) but a lane walks only 10-40 MB/s: left in
// place they would keep nearly every wave of the first launch resident with two or three live lanes.
struct Parked { u64 s, pos, x[kCrossMax]; u32 h, c1, sz, nx, cnt, pad; u32 tab[64]; };

template <bool RESUME>
__global__ __launch_bounds__(64) void fragment_spec_kernel(const u8* __restrict__ data, u64 readable, const u64* __restrict__ file_off,
                                                            const u32* __restrict__ seg_file,
                                                            const u64* __restrict__ seg_base, u64 nseg, FragP P,
                                                            u32 spec_cap, u32* __restrict__ spec_rel,
                                                            u32* __restrict__ spec_cnt, CrossOut* __restrict__ cross,
                                                            unsigned long long* __restrict__ counters, Parked* __restrict__ parked,
                                                            u64 budget) {
  __shared__ u8 tab[16384];
  const u32 lane = (u32)lane_id();
  LaneO1 o{(lds_u8*)tab, (lane >> 5) * 128u + (lane & 31u) * 4u};
  lds_u32* const orow = (lds_u32*)(o.t + o.lanebase);      // this lane's 64 table words sit 256 bytes apart
  o.clear();
  const u64 nwork = RESUME ? counters[1] : nseg;           // counters: [0] segments handed out, [1] parked, [2] resumed
  bool active = false, exhausted = false;
  u64 s = 0, pos = 0, lim = 0, fe = 0, g = 0, segend = 0, x0 = 0, x1 = 0, x2 = 0, x3 = 0;
  u32 cnt = 0, nx = 0;
  u32* out = spec_rel;
  LaneState st{0, 0, 0};
  LaneStream ls; ls.at = ~0ull; ls.last = readable >= 16 ? readable - 16 : 0;
#define ZPQ_FRAG_STEP(PH)                                                                                     \
  {                                                                                                           \
    if (!active && !exhausted) {                                                                              \
      const u64 w = atomicAdd(&counters[RESUME ? 2 : 0], 1ull);                                               \
      if (w < nwork) {                                                                                        \
        if (RESUME) {                                                                                         \
          const Parked* pk = parked + w;                                                                      \
          s = pk->s; pos = pk->pos; x0 = pk->x[0]; x1 = pk->x[1]; x2 = pk->x[2]; x3 = pk->x[3];               \
          st.h = pk->h; st.c1 = pk->c1; st.sz = pk->sz; nx = pk->nx; cnt = pk->cnt;                           \
          _Pragma("unroll") for (int a = 0; a < 64; ++a) orow[a * 64] = pk->tab[a];                           \
        } else {                                                                                              \
          s = w;                                                                                              \
        }                                                                                                     \
        const u32 f = seg_file[s];                                                                            \
        const u64 fs = file_off[f];                                                                           \
        fe = file_off[f + 1];                                                                                 \
        g = fs + (s - seg_base[f]) * P.seg;                                                                   \
        segend = g + P.seg < fe ? g + P.seg : fe;                                                             \
        const u64 far = segend + (u64)P.minf + (u64)kCrossMax * P.maxf + 64; /* kCrossMax crossing fragments */ \
        lim = far < fe ? far : fe;                                                                            \
        if (!RESUME) { pos = g; cnt = 0; nx = 0; x0 = x1 = x2 = x3 = 0; }                                     \
        out = spec_rel + s * (u64)spec_cap;                                                                   \
        ls.at = ~0ull;                                                                                        \
        active = true; /* fresh segments: o1[] and the hash state are clean, lanes stop on a cut */           \
      } else {                                                                                                \
        exhausted = true;                                                                                     \
      }                                                                                                       \
    }                                                                                                         \
    if (!__any(active)) break;                                                                                \
    bool done = false;                                                                                        \
    /* every lane steps every iteration; idle ones have pos >= lim and only take part in the refill */        \
    lane_step<PH>(data, ls, pos, lim, fe, P, o, st, [&](u64 E, bool trig) {                                   \
      if (E < segend) {                                                                                       \
        if (cnt < spec_cap) out[cnt] = (u32)(E - g);                                                          \
        ++cnt;                                                                                                \
        if (E + 1 == segend) done = true; /* cut on the boundary: the next lane starts in step */             \
      } else {                                                                                                \
        if (nx == 0) x0 = E; else if (nx == 1) x1 = E; else if (nx == 2) x2 = E; else x3 = E;                 \
        ++nx;                                                                                                 \
        /* a cut the size limit forced is one no speculating lane made: keep walking (this lane IS the    */ \
        /* true chain) until the hash fires where the lane of that segment could have cut as well         */ \
        if ((trig && E + 1 >= segend + (u64)P.minf) || nx == kCrossMax) done = true;                          \
      }                                                                                                       \
      if (E + 1 == fe) done = true;                                                                           \
    });                                                                                                       \
    if (active && (done || pos >= lim)) {                                                                     \
      if (!RESUME) spec_cnt[s] = cnt < spec_cap ? cnt : spec_cap;                                             \
      CrossOut co; co.x[0] = x0; co.x[1] = x1; co.x[2] = x2; co.x[3] = x3;                                    \
      co.n = (done && cnt <= spec_cap) ? nx : 0xffffffffu; co.pad = 0;                                        \
      cross[s] = co;                                                                                          \
      active = false;                                                                                         \
      lim = pos;                                                                                              \
    } else if (!RESUME && active && pos >= segend + budget) {                                                 \
      /* park: the segment's own list is complete, only the crossing walk is left */                          \
      spec_cnt[s] = cnt < spec_cap ? cnt : spec_cap;                                                          \
      Parked* pk = parked + atomicAdd(&counters[1], 1ull);                                                    \
      pk->s = s; pk->pos = pos; pk->x[0] = x0; pk->x[1] = x1; pk->x[2] = x2; pk->x[3] = x3;                   \
      pk->h = st.h; pk->c1 = st.c1; pk->sz = st.sz; pk->nx = nx; pk->cnt = cnt; pk->pad = 0;                  \
      _Pragma("unroll") for (int a = 0; a < 64; ++a) pk->tab[a] = orow[a * 64];                               \
      o.clear();                                                                                              \
      st.h = 0; st.c1 = 0; st.sz = 0;                                                                         \
      active = false;                                                                                         \
      lim = pos;                                                                                              \
    }                                                                                                         \
  }
  for (;;) {
    ZPQ_FRAG_STEP(0)
    ZPQ_FRAG_STEP(1)
    ZPQ_FRAG_STEP(2)
  }
#undef ZPQ_FRAG_STEP
}

// ---- 2. exact chain: one wave per file ---------------------------------------------------------------
__global__ __launch_bounds__(256) void fragment_stitch_kernel(const u8* __restrict__ data, u64 readable,
                                                               const u64* __restrict__ file_off, u32 nfiles,
                                                               const u64* __restrict__ seg_base, FragP P, u32 spec_cap,
                                                               const u32* __restrict__ spec_rel,
                                                               const u32* __restrict__ spec_cnt,
                                                               const CrossOut* __restrict__ cross,
                                                               const u64* __restrict__ cut_base,
                                                               u64* __restrict__ cuts, u32* __restrict__ cut_cnt,
                                                               const u32* __restrict__ rep) {
  __shared__ WaveLds lds[4];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const u32 f = blockIdx.x * 4 + wave;
  lds[wave].M[lane] = 0; lds[wave].M[lane + 64] = 0; lds[wave].M[lane + 128] = 0; lds[wave].M[lane + 192] = 0;
  __builtin_amdgcn_wave_barrier();
  if (f >= nfiles) return;
  if (rep && rep[f] != f) {          // a twin (twins.hip): its list is its representative's, shifted (fragment_emit_kernel)
    if (lane == 0) cut_cnt[f] = 0;
    return;
  }
  const u64 fs = file_off[f], fe = file_off[f + 1];
  const u64 sb = seg_base[f];
  const u64 kSegBytes = P.seg;
  const u32 nsegf = (u32)((fe - fs + kSegBytes - 1) / kSegBytes);
  u64* out = cuts + cut_base[f];
  u32 cnt = 0;
  u64 S = fs;                 // start of the next fragment of the true chain
  bool synced = true;         // true chain == speculation of segment k from its cut index `from`
  u32 k = 0, from = 0;
  while (S < fe) {
    u64 E = 0;                // last cut emitted in this round
    if (synced) {
      if (k >= nsegf) break;
      const u64 sidx = sb + k;
      const u64 g = fs + (u64)k * kSegBytes;
      const u64 segend = g + kSegBytes < fe ? g + kSegBytes : fe;
      const u32 nk = spec_cnt[sidx];
      const u32* rel = spec_rel + sidx * (u64)spec_cap;
      for (u32 j = from + lane; j < nk; j += 64) out[cnt + (j - from)] = g + rel[j];
      if (nk > from) { S = g + rel[nk - 1] + 1; cnt += nk - from; }
      if (S >= fe) break;
      if (S == segend) { ++k; from = 0; continue; }     // cut on the boundary
      // The lane that produced this list was in the true state from cut `from` on (or from the segment
      // start when from == 0), so the fragment(s) it walked across the boundary are true as well.
      const CrossOut co = cross[sidx];
      if (co.n == 0xffffffffu || co.n == 0 || co.n > kCrossMax) {
        if (lane == 0) atomicAdd(&g_frag_stats[1], 1ull);
        synced = false;       // nothing usable: go on exactly from S
        continue;
      }
      if ((u32)lane < co.n) out[cnt + lane] = co.x[lane];
      cnt += co.n;
      E = co.x[co.n - 1];
      S = E + 1;
      if (S >= fe) break;
    } else {
      // exact evaluation of one fragment.  (No alignment test on S here: a segment without any
      // speculative cut hands over at its own start and must make progress.)
      E = eval_fragment(data, readable, (lds_u64*)lds[wave].M, (lds_u32*)lds[wave].O, S, fe, P);
      if (lane == 0) { atomicAdd(&g_frag_stats[3], 1ull); atomicAdd(&g_frag_stats[4], E + 1 - S); }
      if (lane == 0) out[cnt] = E;
      ++cnt;
      S = E + 1;
      if (S >= fe) break;
    }
    // is the chain back in step with a speculation?  Only a cut that the speculating lane made too
    // puts both in the same (reset) state.
    synced = false;
    if ((S - fs) % kSegBytes == 0) { synced = true; k = (u32)((S - fs) / kSegBytes); from = 0; continue; }
    const u32 k2 = (u32)((E - fs) / kSegBytes);
    const u64 sidx2 = sb + k2;
    const u32 n2 = spec_cnt[sidx2];
    const u32* rel2 = spec_rel + sidx2 * (u64)spec_cap;
    const u32 want = (u32)(E - (fs + (u64)k2 * kSegBytes));
    u32 hit = 0xffffffffu;
    for (u32 j = lane; j < n2; j += 64) if (rel2[j] == want) hit = j;
    const unsigned long long hm = __ballot(hit != 0xffffffffu);
    if (hm) { synced = true; k = k2; from = (u32)__shfl((int)hit, __builtin_ctzll(hm)) + 1; if (lane == 0) atomicAdd(&g_frag_stats[0], 1ull); }
    else if (lane == 0) atomicAdd(&g_frag_stats[2], 1ull);
  }
  if (lane == 0) cut_cnt[f] = cnt;
}

// ---- compaction: per-file cut lists -> global fragment records ----------------------------------
__global__ __launch_bounds__(256) void fragment_emit_kernel(const u64* __restrict__ file_off, u32 nfiles,
                                                             const u64* __restrict__ cut_base,
                                                             const u64* __restrict__ cuts,
                                                             const u32* __restrict__ cut_cnt,
                                                             const u64* __restrict__ frag_base,
                                                             u64* __restrict__ frag_off, u32* __restrict__ frag_len,
                                                             u32* __restrict__ frag_file, const u32* __restrict__ rep) {
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const u32 f = blockIdx.x * 4 + wave;
  if (f >= nfiles) return;
  const u32 r = rep ? rep[f] : f;            // a twin takes its representative's cuts, moved by the distance between the two
  const u64* c = cuts + cut_base[r];
  const u32 n = cut_cnt[r];
  const u64 fb = frag_base[f];
  const u64 shift = file_off[f] - file_off[r];
  for (u32 j = lane; j < n; j += 64) {
    const u64 start = (j ? c[j - 1] + 1 : file_off[r]) + shift;
    frag_off[fb + j] = start;
    frag_len[fb + j] = (u32)(c[j] + 1 + shift - start);
    frag_file[fb + j] = f;
  }
}

// ---- twins: fragment ids of the representatives hashed once, then handed to every file --------------------------
// compact: the (offset, length) records of the representatives' fragments back to back (the SHA-1 pass runs over these)
__global__ __launch_bounds__(256) void twin_compact_kernel(const u32* __restrict__ ufile, u32 nu, const u64* __restrict__ frag_base,
                                                            const u64* __restrict__ ubase, const u64* __restrict__ frag_off,
                                                            const u32* __restrict__ frag_len, u64* __restrict__ uoff,
                                                            u32* __restrict__ ulen) {
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const u32 u = blockIdx.x * 4 + wave;
  if (u >= nu) return;
  const u32 f = ufile[u];
  const u64 fb = frag_base[f], n = frag_base[f + 1] - fb, ub = ubase[f];
  for (u64 j = lane; j < n; j += 64) { uoff[ub + j] = frag_off[fb + j]; ulen[ub + j] = frag_len[fb + j]; }
}
// spread: every file's ids = its representative's (20 bytes = 5 words each; digest arrays are 4-byte aligned)
__global__ __launch_bounds__(256) void twin_spread_kernel(u32 nfiles, const u32* __restrict__ rep, const u64* __restrict__ frag_base,
                                                           const u64* __restrict__ ubase, const u32* __restrict__ udig,
                                                           u32* __restrict__ dig) {
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const u32 f = blockIdx.x * 4 + wave;
  if (f >= nfiles) return;
  const u64 fb = frag_base[f], n = (frag_base[f + 1] - fb) * 5, ub = ubase[rep[f]];
  const u32* src = udig + ub * 5;
  u32* dst = dig + fb * 5;
  for (u64 j = lane; j < n; j += 64) dst[j] = src[j];
}

}  // namespace

#ifndef ZPQ_EMU_FRAGMENT_ONLY
namespace {
// files f with rep && rep[f] != f are twins of file rep[f] (twins.hip): they are not walked, their records are their
// representative's moved by the distance between the two.  frag_base_out (may be null): per-file record prefix, nfiles + 1.
int fragment_run(zpq_ctx* ctx, const uint8_t* d_base, const uint64_t* file_off, size_t nfiles,
                 const zpq_fragment_params* p, uint64_t* d_frag_off, uint32_t* d_frag_len, uint32_t* d_frag_file,
                 size_t frag_cap, size_t* nfrags, const u32* rep, std::vector<u64>* frag_base_out) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (!ctx || !file_off || !p || !nfrags) return ZPQ_ERR_ARG;
  *nfrags = 0;
  if (nfiles == 0) return ZPQ_OK;
  if (nfiles > 0x7fffffffu) return zpq_fail(ctx, ZPQ_ERR_ARG, "too many files");
  if (((uintptr_t)d_base & 15) != 0) return zpq_fail(ctx, ZPQ_ERR_ARG, "d_base must be 16-byte aligned");
  if (p->min_fragment == 0 || p->max_fragment < p->min_fragment) return zpq_fail(ctx, ZPQ_ERR_ARG, "bad fragment limits");
  FragP P;
  P.minf = p->min_fragment; P.maxf = p->max_fragment;
  P.thresh = p->fragment_log2 <= 22 ? 1u << (22 - p->fragment_log2) : 0u;
  const u64 all_bytes = file_off[nfiles];
  u64 total = 0;                             // bytes that are walked: twins are not
  for (size_t f = 0; f < nfiles; ++f)
    if (!rep || rep[f] == f) total += file_off[f + 1] - file_off[f];
  const u64 readable = (all_bytes + 3) & ~3ull;  // callers pad allocations by >= 16 bytes (see header)
  // Every resident lane owns 256 B of LDS: 10 waves per CU would fill the 160 KiB.  Six do better (round 5, profiles/
  // r05c_sweep_fragment_waves_per_cu.txt: the kernel alone 43.7 ms against 55.8 for 54 GB, the step with twelve jobs in flight 109
  // against 117 ms): fewer lanes means longer segments -- a smaller share of crossing walks and of steps a wave spends behind
  // its slowest lane -- and 64 KiB of every unit's LDS stay free for the other jobs' kernels; with four the SIMDs run dry.
  int waves_per_cu = 6;
  if (const char* e = getenv("ZPQ_FRAG_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= 10) waves_per_cu = v; }
  u64 cap_waves = (u64)ctx->cu_count * (u64)waves_per_cu;
  {
    const u64 lanes = cap_waves * 64;
    u64 seg = (total + lanes - lanes / 32 - 1) / (lanes - lanes / 32);         // ~3% slack for the ragged file ends
    seg = (seg + kSegGrain - 1) / kSegGrain * kSegGrain;
    seg = std::min(std::max(seg, kSegMin), kSegMax);
    if (const char* e = getenv("ZPQ_FRAG_SEG")) { const long long v = atoll(e); if (v >= 4096) seg = (u64)v; }   // tests
    P.seg = seg; P.pad = 0;
  }
  const u64 kSegBytes = P.seg;

  // host-side segment and capacity tables
  std::vector<u64> seg_base(nfiles + 1), cut_base(nfiles + 1);
  u64 nseg = 0, ncut = 0;
  for (size_t f = 0; f < nfiles; ++f) {
    if (file_off[f + 1] < file_off[f]) return zpq_fail(ctx, ZPQ_ERR_ARG, "file_off not monotone");
    if (rep && (rep[f] > f || rep[rep[f]] != rep[f])) return zpq_fail(ctx, ZPQ_ERR_ARG, "bad representative table");
    const bool walked = !rep || rep[f] == f;
    u64 len = walked ? file_off[f + 1] - file_off[f] : 0;
    seg_base[f] = nseg; cut_base[f] = ncut;
    nseg += (len + kSegBytes - 1) / kSegBytes;
    ncut += walked ? len / P.minf + 1 : 0;
  }
  seg_base[nfiles] = nseg; cut_base[nfiles] = ncut;
  if (nseg == 0) { if (frag_base_out) frag_base_out->assign(nfiles + 1, 0); return ZPQ_OK; }
  std::vector<u32> seg_file(nseg);
  for (size_t f = 0; f < nfiles; ++f)
    for (u64 s = seg_base[f]; s < seg_base[f + 1]; ++s) seg_file[s] = (u32)f;
  const u32 spec_cap = (u32)(kSegBytes / P.minf + 2);

  // device scratch: [file_off | seg_base | cut_base | frag_base | seg_file | spec_cnt | cut_cnt] , spec_rel, cuts
  const size_t nf1 = nfiles + 1;
  size_t meta_bytes = nf1 * 8 * 4 + nseg * 4 * 2 + nfiles * 4 * 2 + 64 + 256;
  u8* meta = (u8*)zpq_scratch(ctx, 2, meta_bytes);
  u32* d_spec_rel = (u32*)zpq_scratch(ctx, 3, nseg * (size_t)spec_cap * 4 + nseg * (sizeof(CrossOut) + sizeof(Parked)) + 512);
  u64* d_cuts = (u64*)zpq_scratch(ctx, 4, ncut * 8);
  if (!meta || !d_spec_rel || !d_cuts) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "fragment scratch");
  CrossOut* d_cross = (CrossOut*)(d_spec_rel + nseg * (size_t)spec_cap + ((nseg * (size_t)spec_cap) & 1));
  Parked* d_parked = (Parked*)(d_cross + nseg);
  u64* d_file_off = (u64*)meta;
  u64* d_seg_base = d_file_off + nf1;
  u64* d_cut_base = d_seg_base + nf1;
  u64* d_frag_base = d_cut_base + nf1;
  u32* d_seg_file = (u32*)(d_frag_base + nf1);
  u32* d_spec_cnt = d_seg_file + nseg;
  u32* d_cut_cnt = d_spec_cnt + nseg;
  unsigned long long* d_counter = (unsigned long long*)(((uintptr_t)(d_cut_cnt + nfiles) + 7) & ~(uintptr_t)7);
  u32* d_rep = rep ? (u32*)(d_counter + 4) : nullptr;
  hipStream_t st = ctx->stream;
  ZPQ_HIP(ctx, hipMemsetAsync(d_counter, 0, 24, st));
  if (rep) ZPQ_HIP(ctx, hipMemcpyAsync(d_rep, rep, nfiles * 4, hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipMemcpyAsync(d_file_off, file_off, nf1 * 8, hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipMemcpyAsync(d_seg_base, seg_base.data(), nf1 * 8, hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipMemcpyAsync(d_cut_base, cut_base.data(), nf1 * 8, hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipMemcpyAsync(d_seg_file, seg_file.data(), nseg * 4, hipMemcpyHostToDevice, st));

  u64 want_waves = (nseg + 63) / 64;
  if (const char* e = getenv("ZPQ_FRAG_MAX_WAVES")) { const int v = atoi(e); if (v >= 1) cap_waves = std::min<u64>(cap_waves, (u64)v); }  // tests
  u64 budget = 256 << 10;      // bytes a lane may walk past its segment before it parks the walk
  if (const char* e = getenv("ZPQ_FRAG_BUDGET")) { const long long v = atoll(e); budget = v > 0 ? (u64)v : ~0ull >> 1; }
  ZPQ_LAUNCH(ctx, "fragment_spec_kernel", st, fragment_spec_kernel<false>, dim3((unsigned)std::min(want_waves, cap_waves)), dim3(64),
             d_base, all_bytes, d_file_off, d_seg_file, d_seg_base, nseg, P, spec_cap, d_spec_rel, d_spec_cnt, d_cross, d_counter,
             d_parked, budget);
  ZPQ_HIP(ctx, hipGetLastError());
  // parked walks: a few waves, every lane live (waves that find nothing exit at once)
  int resume_waves_per_cu = 2;
  if (const char* e = getenv("ZPQ_FRAG_RESUME_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= 10) resume_waves_per_cu = v; }
  ZPQ_LAUNCH(ctx, "fragment_resume_kernel", st, fragment_spec_kernel<true>,
             dim3((unsigned)std::min<u64>(want_waves, std::min<u64>(cap_waves, (u64)ctx->cu_count * resume_waves_per_cu))), dim3(64), d_base, all_bytes,
             d_file_off, d_seg_file, d_seg_base, nseg, P, spec_cap, d_spec_rel, d_spec_cnt, d_cross, d_counter, d_parked, budget);
  ZPQ_HIP(ctx, hipGetLastError());
  ZPQ_LAUNCH(ctx, "fragment_stitch_kernel", st, fragment_stitch_kernel, dim3((unsigned)((nfiles + 3) / 4)), dim3(256), d_base,
             readable, d_file_off, (u32)nfiles, d_seg_base, P, spec_cap, d_spec_rel, d_spec_cnt, d_cross, d_cut_base, d_cuts,
             d_cut_cnt, (const u32*)d_rep);
  ZPQ_HIP(ctx, hipGetLastError());

  if (getenv("ZPQ_FRAG_STATS")) {
    unsigned long long st8[8] = {0};
    (void)hipStreamSynchronize(st);
    (void)hipMemcpyFromSymbol(st8, HIP_SYMBOL(g_frag_stats), sizeof(st8));
    fprintf(stderr, "[frag stats] nseg=%llu in_step=%llu cross_unusable=%llu lookups_missed=%llu exact_evals=%llu exact_bytes=%llu\n",
            (unsigned long long)nseg, st8[0], st8[1], st8[2], st8[3], st8[4]);
  }
  // per-file counts -> exclusive prefix on the host (nfiles words; the data never leaves HBM)
  std::vector<u32> cnt(nfiles);
  ZPQ_HIP(ctx, hipMemcpyAsync(cnt.data(), d_cut_cnt, nfiles * 4, hipMemcpyDeviceToHost, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  std::vector<u64> frag_base(nf1);
  u64 nf = 0;
  for (size_t f = 0; f < nfiles; ++f) { frag_base[f] = nf; nf += cnt[rep ? rep[f] : f]; }
  frag_base[nfiles] = nf;
  if (frag_base_out) *frag_base_out = frag_base;
  *nfrags = (size_t)nf;
  if (nf > frag_cap) return zpq_fail(ctx, ZPQ_ERR_CAPACITY, "fragment capacity %zu < %llu", frag_cap, (unsigned long long)nf);
  ZPQ_HIP(ctx, hipMemcpyAsync(d_frag_base, frag_base.data(), nf1 * 8, hipMemcpyHostToDevice, st));
  ZPQ_LAUNCH(ctx, "fragment_emit_kernel", st, fragment_emit_kernel, dim3((unsigned)((nfiles + 3) / 4)), dim3(256), d_file_off,
                     (u32)nfiles, d_cut_base, d_cuts, d_cut_cnt, d_frag_base, d_frag_off, d_frag_len, d_frag_file, (const u32*)d_rep);
  ZPQ_HIP(ctx, hipGetLastError());
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  return ZPQ_OK;
}
}  // namespace

extern "C" {

void zpq_fragment_params_default(zpq_fragment_params* p) {
  p->fragment_log2 = 6;
  p->min_fragment = 64u << 6;
  p->max_fragment = 8128u << 6;
}

size_t zpq_fragment_capacity(const uint64_t* file_off, size_t nfiles, const zpq_fragment_params* p) {
  size_t cap = 0;
  const u64 minf = p->min_fragment ? p->min_fragment : 1;
  for (size_t f = 0; f < nfiles; ++f) {
    u64 len = file_off[f + 1] - file_off[f];
    cap += (size_t)(len / minf + 1);
  }
  return cap;
}

int zpq_fragment_dev(zpq_ctx* ctx, const uint8_t* d_base, const uint64_t* file_off, size_t nfiles,
                     const zpq_fragment_params* p, uint64_t* d_frag_off, uint32_t* d_frag_len, uint32_t* d_frag_file,
                     size_t frag_cap, size_t* nfrags) {
  return fragment_run(ctx, d_base, file_off, nfiles, p, d_frag_off, d_frag_len, d_frag_file, frag_cap, nfrags, nullptr, nullptr);
}

// Fragment + SHA-1 of every fragment in one call, with the twin-file fold (twins.hip) in front: files whose bytes equal an
// earlier file's are found by comparison, only the representatives are fragmented and hashed, and the records of the
// twins are their representative's, moved.  The result is what zpq_fragment_dev + zpq_sha1_extents_dev give.
int zpq_fragment_sha1_dev(zpq_ctx* ctx, const uint8_t* d_base, const uint64_t* file_off, size_t nfiles,
                          const zpq_fragment_params* p, uint64_t* d_frag_off, uint32_t* d_frag_len, uint32_t* d_frag_file,
                          uint8_t* d_digests, size_t frag_cap, size_t* nfrags, uint32_t flags, uint32_t* file_rep, uint64_t stats[4]) {
  if (ctx) (void)hipSetDevice(ctx->device);
  if (!ctx || !file_off || !p || !nfrags) return ZPQ_ERR_ARG;
  *nfrags = 0;
  if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
  if (((uintptr_t)d_digests & 3) != 0) return zpq_fail(ctx, ZPQ_ERR_ARG, "d_digests must be 4-byte aligned");
  hipStream_t st = ctx->stream;
  std::vector<u32> rep_local;
  u32* rep = file_rep;
  bool twins = false;
  static const bool env_off = [] { const char* e = getenv("ZPQ_TWINS"); return e && atoi(e) == 0; }();
  if (!(flags & ZPQ_FS_NO_TWINS) && !env_off && nfiles >= 2) {
    if (!rep) { rep_local.resize(nfiles); rep = rep_local.data(); }
    std::vector<u64> len(nfiles);
    for (size_t f = 0; f < nfiles; ++f) {
      if (file_off[f + 1] < file_off[f]) return zpq_fail(ctx, ZPQ_ERR_ARG, "file_off not monotone");
      len[f] = file_off[f + 1] - file_off[f];
    }
    u64 tst[4];
    const int rc = zpq_twins_find(ctx, st, d_base, file_off, len.data(), nfiles, p->min_fragment, rep, tst);
    if (rc) return rc;
    if (stats) memcpy(stats, tst, sizeof tst);
    twins = tst[0] != 0;
  } else if (rep) {
    for (size_t f = 0; f < nfiles; ++f) rep[f] = (u32)f;
  }
  std::vector<u64> frag_base;
  int rc = fragment_run(ctx, d_base, file_off, nfiles, p, d_frag_off, d_frag_len, d_frag_file, frag_cap, nfrags, twins ? rep : nullptr,
                        &frag_base);
  if (rc) return rc;
  const size_t nf = *nfrags;
  if (nf == 0) return ZPQ_OK;
  if (!twins) return zpq_sha1_extents_on(ctx, st, d_base, d_frag_off, d_frag_len, nf, d_digests);
  // ids of the representatives' fragments, then every file takes its representative's
  std::vector<u32> ufile;
  std::vector<u64> ubase(nfiles, 0);
  u64 nu = 0;
  for (size_t f = 0; f < nfiles; ++f)
    if (rep[f] == f) { ufile.push_back((u32)f); ubase[f] = nu; nu += frag_base[f + 1] - frag_base[f]; }
  const size_t o_ubase = 0, o_fbase = o_ubase + nfiles * 8, o_uoff = o_fbase + (nfiles + 1) * 8, o_rep = o_uoff + nu * 8,
               o_ufile = o_rep + nfiles * 4, o_ulen = o_ufile + ufile.size() * 4, o_udig = (o_ulen + nu * 4 + 63) & ~(size_t)63;
  u8* d_t = (u8*)zpq_scratch(ctx, 29, o_udig + nu * 20 + 256);
  if (!d_t) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "twin scratch");
  ZPQ_HIP(ctx, hipMemcpyAsync(d_t + o_ubase, ubase.data(), nfiles * 8, hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipMemcpyAsync(d_t + o_fbase, frag_base.data(), (nfiles + 1) * 8, hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipMemcpyAsync(d_t + o_rep, rep, nfiles * 4, hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipMemcpyAsync(d_t + o_ufile, ufile.data(), ufile.size() * 4, hipMemcpyHostToDevice, st));
  ZPQ_LAUNCH(ctx, "twin_compact_kernel", st, twin_compact_kernel, dim3((unsigned)((ufile.size() + 3) / 4)), dim3(256), (const u32*)(d_t + o_ufile),
             (u32)ufile.size(), (const u64*)(d_t + o_fbase), (const u64*)(d_t + o_ubase), (const u64*)d_frag_off, (const u32*)d_frag_len,
             (u64*)(d_t + o_uoff), (u32*)(d_t + o_ulen));
  ZPQ_HIP(ctx, hipGetLastError());
  // Few fragments (what is left of a tree of copies): one WAVE per fragment -- a lane hashes ~25 MB/s, so lane-wise the
  // pass lasts as long as its longest fragment (20 ms for 508 KiB) however few there are; wave-wise (schedule off the
  // chain, 78 MB/s) it is 6.5 ms, at 39x the issue slots per byte: worth it below ~0.5 GB.
  u64 ubytes = 0;
  for (u32 f : ufile) ubytes += file_off[f + 1] - file_off[f];
  static const int wave_ids = [] { const char* e = getenv("ZPQ_TWIN_WAVE_IDS"); return e ? atoi(e) : 1; }();
  if (wave_ids && nu <= 65535 && ubytes <= (512ull << 20))
    rc = zpq_sha1_chains_on(ctx, st, d_base, (const u64*)(d_t + o_uoff), (const u32*)(d_t + o_ulen), (size_t)nu, d_t + o_udig, "sha1_fragment_waves_kernel");
  else
    rc = zpq_sha1_extents_on(ctx, st, d_base, (const u64*)(d_t + o_uoff), (const u32*)(d_t + o_ulen), (size_t)nu, d_t + o_udig);
  if (rc) return rc;
  ZPQ_LAUNCH(ctx, "twin_spread_kernel", st, twin_spread_kernel, dim3((unsigned)((nfiles + 3) / 4)), dim3(256), (u32)nfiles, (const u32*)(d_t + o_rep),
             (const u64*)(d_t + o_fbase), (const u64*)(d_t + o_ubase), (const u32*)(d_t + o_udig), (u32*)d_digests);
  ZPQ_HIP(ctx, hipGetLastError());
  ZPQ_HIP(ctx, hipStreamSynchronize(st));      // the host tables above are locals
  return ZPQ_OK;
}

}  // extern "C"
#endif  // ZPQ_EMU_FRAGMENT_ONLY
