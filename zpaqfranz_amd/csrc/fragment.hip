// Content-defined fragmenter: the per-byte loop of Jidac::add
//     h = (h + c + 1) * (c == o1[c1] ? 314159265 : 271828182);  o1[c1] = c;  c1 = c;
//     cut when sz >= MAX || (h < 2^(22-fragment) && sz >= MIN) || EOF       (state reset per fragment)
// (reference: zpaqfranz.cpp is absent from the snapshot; algorithm per SURVEY.md Appendix C.4, its
// output records are read back at ZSFX/zsfx.cpp:1463-1500 and pinned by AUTOTEST/sha256.zpaq).
//
// MI355X formulation -- exact, no heuristics.  The recurrence is serial per fragment and "cut k
// decides where fragment k+1 starts" is serial per file, so the parallelism is speculative:
//  1. fragment_spec_kernel: files are split into 1 MiB segments and EVERY LANE fragments one segment
//     from its own start, as if a fragment began there.  Lane-serial is the instruction-efficient
//     shape for this loop (~12 VALU per byte-step of 64 lanes): each lane owns a 256-entry o1[] table
//     in LDS (bank = lane, conflict free) and streams its segment 16 bytes per load; the 16 LDS
//     read/write pairs of a group are issued back to back, the multiply chain runs on registers, and
//     only a group whose minimum hash falls under the threshold is re-walked byte by byte.
//  2. fragment_seam_kernel: one lane per segment boundary continues from the last speculative cut of
//     segment k across the boundary until one of its cuts coincides with a speculative cut of the
//     next segment: from there both chains are in the same (reset) state.
//  3. fragment_stitch_kernel: one wave per file glues the pieces: speculative cuts, seam cuts,
//     speculative cuts ... -- every hand-over is verified (the seam must start where the previous
//     piece ended), and anything that does not line up (never-synchronising data such as runs of
//     zeros) is re-evaluated exactly by the wave-parallel evaluator below (64 bytes per step:
//     in-window o1 forwarding through 64-bit LDS masks, affine-map scan for the hash).
// Integer-only byte work; traffic = input read ~1.1x (seam re-reads); bound by VALU/LDS issue.
#include <algorithm>

#include "zpq_internal.h"

namespace {

constexpr u64 kSegBytes = 1ull << 18;   // speculation segment (independent of the fragment size limits)
constexpr u64 kNone = ~0ull;

struct FragP {
  u32 minf, maxf, thresh;  // thresh = 2^(22-fragment) or 0 when fragment > 22
};

struct WaveLds {
  unsigned long long M[256];  // per predecessor value: mask of lanes (this window) having it
  u32 O[64];                  // o1[] packed 4 entries per word
};

// Streams a byte range through three 256-byte register chunks (one aligned dword per lane each),
// prefetched two chunks ahead; get() hands lane l the byte at pos+l.
struct ByteReader {
  const u8* data;
  u64 readable;  // bytes of `data` that may be touched (multiple of 4)
  u64 cb;        // offset of chunk r0 (multiple of 4)
  u32 r0, r1, r2;
  __device__ __forceinline__ u32 load(u64 off) const {
    u64 a = off + 4u * (u32)lane_id();
    return a + 4 <= readable ? *(const u32*)(data + a) : 0u;
  }
  __device__ __forceinline__ void init(u64 pos) {
    cb = pos & ~3ull;
    r0 = load(cb); r1 = load(cb + 256); r2 = load(cb + 512);
  }
  __device__ __forceinline__ void advance_to(u64 pos) {
    while (pos >= cb + 256) { r0 = r1; r1 = r2; cb += 256; r2 = load(cb + 512); }
  }
  __device__ __forceinline__ u32 get(u64 pos) const {  // requires cb <= pos < cb+256
    u32 idx = (u32)(pos - cb) + (u32)lane_id();
    u32 src = idx >> 2;
    u32 v0 = __shfl(r0, (int)(src & 63)), v1 = __shfl(r1, (int)(src & 63));
    u32 v = src < 64 ? v0 : v1;
    return (v >> ((idx & 3) * 8)) & 255u;
  }
};

__device__ __forceinline__ void reset_o1(WaveLds& L) {
  L.O[lane_id()] = 0;
  __builtin_amdgcn_wave_barrier();
}

// Evaluates the fragment that starts at S (fresh state) and returns the offset E of its last byte,
// or kNone if no cut was found before `stop` (speculative callers stop at their segment end).
__device__ u64 eval_fragment(ByteReader& rd, WaveLds& L, u64 S, u64 file_end, u64 stop, const FragP P) {
  const int lane = lane_id();
  volatile unsigned long long* M = L.M;
  volatile u8* O = (volatile u8*)L.O;
  reset_o1(L);
  u64 pos = S;
  u32 hin = 0, c1in = 0;
  for (;;) {
    if (pos >= stop) return kNone;
    rd.advance_to(pos);
    const u64 q = pos + (u64)lane;
    const bool valid = q < file_end;
    const u32 craw = rd.get(pos);  // shuffles inside: must run with the whole wave active
    const u32 c = valid ? craw : 0u;
    u32 p = __shfl_up(c, 1);
    if (lane == 0) p = c1in;
    if (valid) atomicOr((unsigned long long*)&L.M[p], 1ull << lane);
    __builtin_amdgcn_wave_barrier();
    const unsigned long long mask = valid ? M[p] : 0ull;
    const unsigned long long lower = mask & ((1ull << lane) - 1ull);
    const int k = lower ? 63 - __builtin_clzll(lower) : 0;
    const u32 pc = __shfl(c, k);
    const u32 pred = lower ? pc : (u32)O[p];
    const u32 m = (valid && c == pred) ? 314159265u : 271828182u;
    // affine map of this byte: h -> a*h + b; invalid lanes are the identity
    u32 a = valid ? m : 1u, b = valid ? (c + 1u) * m : 0u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      u32 alo = __shfl_up(a, d), blo = __shfl_up(b, d);
      if (lane >= d) { b = a * blo + b; a = a * alo; }
    }
    const u32 h = a * hin + b;
    const u64 sz = q - S + 1;
    const bool cut = valid && (sz >= P.maxf || (h < P.thresh && sz >= P.minf) || q + 1 == file_end);
    const unsigned long long cm = __ballot(cut);
    if (valid) M[p] = 0ull;  // leave the mask table clean for the next window
    if (cm) {
      __builtin_amdgcn_wave_barrier();
      u64 E = pos + (u64)__builtin_ctzll(cm);
      return E < stop ? E : kNone;
    }
    if (valid && (mask >> lane) == 1ull) O[p] = (u8)c;  // latest occurrence of p in the window
    __builtin_amdgcn_wave_barrier();
    hin = __shfl(h, 63);
    c1in = __shfl(c, 63);
    pos += 64;
  }
}


// ---- lane-serial evaluator ------------------------------------------------------------------------
// o1[] of lane l lives at byte ((v>>2)*2 + (l>>5))*128 + (l&31)*4 + (v&3) of a 16 KiB per-wave block:
// every lane of a 32-lane half hits its own bank whatever v is.
typedef __attribute__((address_space(3))) volatile u8 lds_u8;     // keeps the accesses ds_* (not flat_*)
typedef __attribute__((address_space(3))) volatile u32 lds_u32;
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
    u32 hh = s.h, sz = s.sz; int cutj = -1;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      hh = (hh + c[j] + 1u) * (c[j] == pr[j] ? 314159265u : 271828182u);
      ++sz;
      if (cutj < 0 && (sz >= P.maxf || (hh < P.thresh && sz >= P.minf) || pos + j + 1 == file_end)) cutj = j;
    }
    if (cutj >= 0) {
      const u64 E = pos + (u64)cutj;
      on_cut(E);
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
// run at 2-3 waves per SIMD -- LDS bound -- so HBM latency has to be hidden by hand).  The steady-state
// prefetch is UNCONDITIONAL (clamped address, executed by every lane every step): a load behind a branch
// would force the compiler to drain the whole queue (s_waitcnt vmcnt(0)) before touching older data.
struct LaneStream {
  u64 at;          // stream offset of a[0]; ~0 when nothing usable is loaded
  u64 last;        // highest offset a 16-byte load may start at (readable - 16)
  u32x4 a[4], b[4], c[4];
  __device__ __forceinline__ u32x4 ld(const u8* data, u64 off) const {
    return *(const u32x4_u*)(data + (off < last ? off : last));
  }
  __device__ __forceinline__ void prime(const u8* data, u64 pos) {
    at = pos;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = ld(data, pos + 16 * i); b[i] = ld(data, pos + 64 + 16 * i); c[i] = ld(data, pos + 128 + 16 * i); }
  }
};

// Advances this lane's stream by 64 bytes (16 or 1 near the end of [pos, lim)).  Must be called by
// every lane of the wave every step (finished lanes pass pos >= lim and only take part in the
// prefetch).  on_cut(E) is called for every cut.
template <class OnCut>
__device__ __forceinline__ void lane_step(const u8* __restrict__ data, LaneStream& ls, u64& pos, const u64 lim,
                                          const u64 file_end, const FragP& P, LaneO1& o, LaneState& s, OnCut&& on_cut) {
  if (pos + 64 <= lim) {
    if (ls.at != pos) ls.prime(data, pos);            // rare: first step, or right after a cut
    bool cut = false;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if (!cut) cut = lane_group(ls.a[g], pos, file_end, P, o, s, on_cut);
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
        on_cut(pos);
        o.clear();
        s.h = 0; s.c1 = 0; s.sz = 0;
      }
      ++pos;
    }
  }
  // steady-state rotation: a <- b <- c <- bytes [pos+128, pos+192) (meaningless but harmless when the
  // stream is out of step; the next call primes)
  const u64 nxt = (ls.at == pos ? pos : 0) + 128;
#pragma unroll
  for (int i = 0; i < 4; ++i) { ls.a[i] = ls.b[i]; ls.b[i] = ls.c[i]; ls.c[i] = ls.ld(data, nxt + 16 * i); }
}

// ---- 1. speculative pass: one LANE per 1 MiB segment ----------------------------------------------
__global__ __launch_bounds__(64) void fragment_spec_kernel(const u8* __restrict__ data, u64 readable, const u64* __restrict__ file_off,
                                                            const u32* __restrict__ seg_file,
                                                            const u64* __restrict__ seg_base, u64 nseg, FragP P,
                                                            u32 spec_cap, u32* __restrict__ spec_rel,
                                                            u32* __restrict__ spec_cnt) {
  __shared__ u8 tab[16384];
  const u32 lane = (u32)lane_id();
  const u64 s = (u64)blockIdx.x * 64 + lane;
  LaneO1 o{(lds_u8*)tab, (lane >> 5) * 128u + (lane & 31u) * 4u};
  o.clear();
  bool active = s < nseg;
  u64 pos = 0, lim = 0, fe = 0, g = 0;
  if (active) {
    const u32 f = seg_file[s];
    const u64 fs = file_off[f];
    fe = file_off[f + 1];
    g = fs + (s - seg_base[f]) * kSegBytes;
    lim = g + kSegBytes < fe ? g + kSegBytes : fe;
    pos = g;
  }
  LaneState st{0, 0, 0};
  LaneStream ls; ls.at = ~0ull; ls.last = readable >= 16 ? readable - 16 : 0;
  u32 cnt = 0;
  u32* out = spec_rel + s * (u64)spec_cap;
  while (__any(active)) {      // finished lanes keep stepping (pos >= lim: prefetch only)
    lane_step(data, ls, pos, lim, fe, P, o, st, [&](u64 E) { if (cnt < spec_cap) out[cnt] = (u32)(E - g); ++cnt; });
    active = pos < lim;
  }
  if (s < nseg) spec_cnt[s] = cnt < spec_cap ? cnt : spec_cap;
}

// ---- 2. seams: one LANE per segment boundary ---------------------------------------------------------
// seam k of a file continues from the last speculative cut of segment k until it is back in step with a
// later segment's speculation (or has run past segment k+1 without meeting it).
struct SeamOut { u64 start; u64 cont; u32 cnt; u32 sync_seg; u32 sync_from; u32 pad; };
constexpr u32 kNoSync = 0xffffffffu;

__global__ __launch_bounds__(64) void fragment_seam_kernel(const u8* __restrict__ data, u64 readable, const u64* __restrict__ file_off,
                                                            const u32* __restrict__ seg_file,
                                                            const u64* __restrict__ seg_base, u64 nseg, FragP P,
                                                            u32 spec_cap, const u32* __restrict__ spec_rel,
                                                            const u32* __restrict__ spec_cnt,
                                                            SeamOut* __restrict__ seam, u32* __restrict__ seam_rel) {
  __shared__ u8 tab[16384];
  const u32 lane = (u32)lane_id();
  const u64 s = (u64)blockIdx.x * 64 + lane;      // seam after segment s (within the same file)
  LaneO1 o{(lds_u8*)tab, (lane >> 5) * 128u + (lane & 31u) * 4u};
  o.clear();
  bool active = false;
  u64 pos = 0, lim = 0, fe = 0, fs = 0, g = 0, sb = 0;
  u32 k = 0, nsegf = 0;
  SeamOut so{~0ull, 0, 0, kNoSync, 0, 0};
  if (s < nseg) {
    const u32 f = seg_file[s];
    fs = file_off[f]; fe = file_off[f + 1]; sb = seg_base[f];
    k = (u32)(s - sb);
    nsegf = (u32)((fe - fs + kSegBytes - 1) / kSegBytes);
    g = fs + (u64)k * kSegBytes;
    const u32 nk = spec_cnt[s];
    if (k + 1 < nsegf && nk > 0 && nk < spec_cap) {
      so.start = g + spec_rel[s * (u64)spec_cap + nk - 1] + 1;
      pos = so.start;
      lim = g + 9 * kSegBytes < fe ? g + 9 * kSegBytes : fe;     // give up 8 segments further on (fragments may span several)
      active = true;
      if (pos == g + kSegBytes) { so.sync_seg = k + 1; so.sync_from = 0; active = false; }   // cut on the boundary
    }
  }
  LaneState st{0, 0, 0};
  LaneStream ls; ls.at = ~0ull; ls.last = readable >= 16 ? readable - 16 : 0;
  u32* out = seam_rel + s * (u64)spec_cap;
  while (__any(active)) {      // finished lanes keep stepping with pos >= lim (prefetch only)
    {
      bool synced = false;
      lane_step(data, ls, pos, lim, fe, P, o, st, [&](u64 E) {
        if (so.cnt < spec_cap) out[so.cnt] = (u32)(E - g);
        ++so.cnt;
        if (E + 1 >= fe) { so.sync_seg = nsegf; so.sync_from = 0; synced = true; return; }   // reached EOF
        const u32 k2 = (u32)((E - fs) / kSegBytes);
        if ((E + 1 - fs) % kSegBytes == 0) { so.sync_seg = k2 + 1; so.sync_from = 0; synced = true; return; }
        if (k2 > k) {                           // is E a speculative cut of its segment?
          const u64 s2 = sb + k2;
          const u32* rel = spec_rel + s2 * (u64)spec_cap;
          const u32 want = (u32)(E - (fs + (u64)k2 * kSegBytes));
          u32 lo = 0, hi = spec_cnt[s2];
          while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (rel[mid] < want) lo = mid + 1; else hi = mid; }
          if (lo < spec_cnt[s2] && rel[lo] == want) { so.sync_seg = k2; so.sync_from = lo + 1; synced = true; }
        }
      });
      if (synced) lim = pos;                    // done: no further bytes for this lane
      active = pos < lim;
    }
  }
  if (s < nseg) {
    so.cont = pos;
    if (so.cnt > spec_cap) { so.start = ~0ull; }      // overflow: never trusted
    seam[s] = so;
  }
}

// ---- 3. exact chain: one wave per file ---------------------------------------------------------------
__global__ __launch_bounds__(256) void fragment_stitch_kernel(const u8* __restrict__ data, u64 readable,
                                                               const u64* __restrict__ file_off, u32 nfiles,
                                                               const u64* __restrict__ seg_base, FragP P, u32 spec_cap,
                                                               const u32* __restrict__ spec_rel,
                                                               const u32* __restrict__ spec_cnt,
                                                               const SeamOut* __restrict__ seam,
                                                               const u32* __restrict__ seam_rel,
                                                               const u64* __restrict__ cut_base,
                                                               u64* __restrict__ cuts, u32* __restrict__ cut_cnt) {
  __shared__ WaveLds lds[4];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const u32 f = blockIdx.x * 4 + wave;
  lds[wave].M[lane] = 0; lds[wave].M[lane + 64] = 0; lds[wave].M[lane + 128] = 0; lds[wave].M[lane + 192] = 0;
  __builtin_amdgcn_wave_barrier();
  if (f >= nfiles) return;
  const u64 fs = file_off[f], fe = file_off[f + 1];
  const u64 sb = seg_base[f];
  const u32 nsegf = (u32)((fe - fs + kSegBytes - 1) / kSegBytes);
  u64* out = cuts + cut_base[f];
  u32 cnt = 0;
  ByteReader rd{data, readable, 0, 0, 0, 0};
  u64 S = fs;                 // start of the next fragment of the true chain
  bool synced = true;         // true chain == speculation of segment k from its cut index `from`
  u32 k = 0, from = 0;
  while (S < fe) {
    if (synced) {
      if (k >= nsegf) break;
      const u64 sidx = sb + k;
      const u64 g = fs + (u64)k * kSegBytes;
      const u32 nk = spec_cnt[sidx];
      const u32* rel = spec_rel + sidx * (u64)spec_cap;
      for (u32 j = from + lane; j < nk; j += 64) out[cnt + (j - from)] = g + rel[j];
      if (nk > from) { S = g + rel[nk - 1] + 1; cnt += nk - from; }
      if (S >= fe) break;
      const SeamOut so = seam[sidx];
      if (so.start == S) {                      // the seam continued exactly this chain
        const u32* srel = seam_rel + sidx * (u64)spec_cap;
        for (u32 j = lane; j < so.cnt; j += 64) out[cnt + j] = g + srel[j];
        if (so.cnt) S = g + srel[so.cnt - 1] + 1;
        cnt += so.cnt;
        if (so.sync_seg != kNoSync) { k = so.sync_seg; from = so.sync_from; continue; }
        // else: the seam ran through segment k+1 without meeting its speculation; go on exactly from S
      }
      synced = false;
      rd.init(S);
    }
    // exact evaluation of one fragment, then look for the speculation again.  (No alignment test on
    // S here: a segment without any speculative cut hands over at its own start and must make progress.)
    const u64 E = eval_fragment(rd, lds[wave], S, fe, kNone, P);
    if (lane == 0) out[cnt] = E;
    ++cnt;
    S = E + 1;
    if (S >= fe) break;
    if ((S - fs) % kSegBytes == 0) { synced = true; k = (u32)((S - fs) / kSegBytes); from = 0; continue; }
    const u32 k2 = (u32)((E - fs) / kSegBytes);
    const u64 sidx2 = sb + k2;
    const u32 n2 = spec_cnt[sidx2];
    const u32* rel2 = spec_rel + sidx2 * (u64)spec_cap;
    const u32 want = (u32)(E - (fs + (u64)k2 * kSegBytes));
    u32 hit = 0xffffffffu;
    for (u32 j = lane; j < n2; j += 64) if (rel2[j] == want) hit = j;
    const unsigned long long hm = __ballot(hit != 0xffffffffu);
    if (hm) { synced = true; k = k2; from = (u32)__shfl((int)hit, __builtin_ctzll(hm)) + 1; }
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
                                                             u32* __restrict__ frag_file) {
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const u32 f = blockIdx.x * 4 + wave;
  if (f >= nfiles) return;
  const u64* c = cuts + cut_base[f];
  const u32 n = cut_cnt[f];
  const u64 fb = frag_base[f];
  for (u32 j = lane; j < n; j += 64) {
    const u64 start = j ? c[j - 1] + 1 : file_off[f];
    frag_off[fb + j] = start;
    frag_len[fb + j] = (u32)(c[j] + 1 - start);
    frag_file[fb + j] = f;
  }
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
  if (!ctx || !file_off || !p || !nfrags) return ZPQ_ERR_ARG;
  *nfrags = 0;
  if (nfiles == 0) return ZPQ_OK;
  if (nfiles > 0x7fffffffu) return zpq_fail(ctx, ZPQ_ERR_ARG, "too many files");
  if (((uintptr_t)d_base & 15) != 0) return zpq_fail(ctx, ZPQ_ERR_ARG, "d_base must be 16-byte aligned");
  if (p->min_fragment == 0 || p->max_fragment < p->min_fragment) return zpq_fail(ctx, ZPQ_ERR_ARG, "bad fragment limits");
  FragP P;
  P.minf = p->min_fragment; P.maxf = p->max_fragment;
  P.thresh = p->fragment_log2 <= 22 ? 1u << (22 - p->fragment_log2) : 0u;
  const u64 total = file_off[nfiles];
  const u64 readable = (total + 3) & ~3ull;  // callers pad allocations by >= 16 bytes (see header)

  // host-side segment and capacity tables
  std::vector<u64> seg_base(nfiles + 1), cut_base(nfiles + 1);
  u64 nseg = 0, ncut = 0;
  for (size_t f = 0; f < nfiles; ++f) {
    if (file_off[f + 1] < file_off[f]) return zpq_fail(ctx, ZPQ_ERR_ARG, "file_off not monotone");
    u64 len = file_off[f + 1] - file_off[f];
    seg_base[f] = nseg; cut_base[f] = ncut;
    nseg += (len + kSegBytes - 1) / kSegBytes;
    ncut += len / P.minf + 1;
  }
  seg_base[nfiles] = nseg; cut_base[nfiles] = ncut;
  if (nseg == 0) return ZPQ_OK;
  std::vector<u32> seg_file(nseg);
  for (size_t f = 0; f < nfiles; ++f)
    for (u64 s = seg_base[f]; s < seg_base[f + 1]; ++s) seg_file[s] = (u32)f;
  const u32 spec_cap = (u32)(kSegBytes / P.minf + 2);

  // device scratch: [file_off | seg_base | cut_base | frag_base | seg_file | spec_cnt | cut_cnt] , spec_rel, cuts
  const size_t nf1 = nfiles + 1;
  size_t meta_bytes = nf1 * 8 * 4 + nseg * 4 * 2 + nfiles * 4 + 256;
  u8* meta = (u8*)zpq_scratch(ctx, 2, meta_bytes);
  u32* d_spec_rel = (u32*)zpq_scratch(ctx, 3, nseg * (size_t)spec_cap * 8 + nseg * sizeof(SeamOut) + 256);
  u64* d_cuts = (u64*)zpq_scratch(ctx, 4, ncut * 8);
  if (!meta || !d_spec_rel || !d_cuts) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "fragment scratch");
  u32* d_seam_rel = d_spec_rel + nseg * (size_t)spec_cap;
  SeamOut* d_seam = (SeamOut*)(d_seam_rel + nseg * (size_t)spec_cap + ((nseg * (size_t)spec_cap) & 1));
  u64* d_file_off = (u64*)meta;
  u64* d_seg_base = d_file_off + nf1;
  u64* d_cut_base = d_seg_base + nf1;
  u64* d_frag_base = d_cut_base + nf1;
  u32* d_seg_file = (u32*)(d_frag_base + nf1);
  u32* d_spec_cnt = d_seg_file + nseg;
  u32* d_cut_cnt = d_spec_cnt + nseg;
  hipStream_t st = ctx->stream;
  ZPQ_HIP(ctx, hipMemcpyAsync(d_file_off, file_off, nf1 * 8, hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipMemcpyAsync(d_seg_base, seg_base.data(), nf1 * 8, hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipMemcpyAsync(d_cut_base, cut_base.data(), nf1 * 8, hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipMemcpyAsync(d_seg_file, seg_file.data(), nseg * 4, hipMemcpyHostToDevice, st));

  ZPQ_LAUNCH(ctx, "fragment_spec_kernel", st, fragment_spec_kernel, dim3((unsigned)((nseg + 63) / 64)), dim3(64), d_base,
             total, d_file_off, d_seg_file, d_seg_base, nseg, P, spec_cap, d_spec_rel, d_spec_cnt);
  ZPQ_HIP(ctx, hipGetLastError());
  ZPQ_LAUNCH(ctx, "fragment_seam_kernel", st, fragment_seam_kernel, dim3((unsigned)((nseg + 63) / 64)), dim3(64), d_base,
             total, d_file_off, d_seg_file, d_seg_base, nseg, P, spec_cap, d_spec_rel, d_spec_cnt, d_seam, d_seam_rel);
  ZPQ_HIP(ctx, hipGetLastError());
  ZPQ_LAUNCH(ctx, "fragment_stitch_kernel", st, fragment_stitch_kernel, dim3((unsigned)((nfiles + 3) / 4)), dim3(256), d_base,
             readable, d_file_off, (u32)nfiles, d_seg_base, P, spec_cap, d_spec_rel, d_spec_cnt, d_seam, d_seam_rel,
             d_cut_base, d_cuts, d_cut_cnt);
  ZPQ_HIP(ctx, hipGetLastError());

  // per-file counts -> exclusive prefix on the host (nfiles words; the data never leaves HBM)
  std::vector<u32> cnt(nfiles);
  ZPQ_HIP(ctx, hipMemcpyAsync(cnt.data(), d_cut_cnt, nfiles * 4, hipMemcpyDeviceToHost, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  std::vector<u64> frag_base(nf1);
  u64 nf = 0;
  for (size_t f = 0; f < nfiles; ++f) { frag_base[f] = nf; nf += cnt[f]; }
  frag_base[nfiles] = nf;
  *nfrags = (size_t)nf;
  if (nf > frag_cap) return zpq_fail(ctx, ZPQ_ERR_CAPACITY, "fragment capacity %zu < %llu", frag_cap, (unsigned long long)nf);
  ZPQ_HIP(ctx, hipMemcpyAsync(d_frag_base, frag_base.data(), nf1 * 8, hipMemcpyHostToDevice, st));
  ZPQ_LAUNCH(ctx, "fragment_emit_kernel", st, fragment_emit_kernel, dim3((unsigned)((nfiles + 3) / 4)), dim3(256), d_file_off,
                     (u32)nfiles, d_cut_base, d_cuts, d_cut_cnt, d_frag_base, d_frag_off, d_frag_len, d_frag_file);
  ZPQ_HIP(ctx, hipGetLastError());
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  return ZPQ_OK;
}

}  // extern "C"
