// CPU model, stage 2 (see README.md): the crossing-free fragmenter in the shape the GPU kernels will have.
//
//   pass P  page summaries: per 4 KiB page the set of context bytes present (256 bits) and the last successor of
//           each context inside the page (256 B).  Store-only, no dependence between pages.
//   pass M  per file: running merge of the page tables -> the exact global o1[] table at every segment start
//           (here: at every page boundary on demand).
//   pass S  one lane per segment, equal work, no crossing: starts WARM bytes early with the exact table and h = 0,
//           walks with NO resets, and records the positions where the (converged) global hash is below the threshold.
//   pass T  one chain per file: fragment by fragment, page by page.  While the true chain is in step with the global
//           one, a page without a new context costs one mask test and a lookup in the trigger list; a page with a
//           new context is scanned; a first occurrence whose fresh-table prediction differs from the global one opens
//           a disturbance window that is evaluated exactly (global hash re-derived from the 32 unpredicted bytes
//           before it) until 32 unpredicted bytes have passed.
//
// Everything pass T looks up is either in the page summaries or found by a bounded backward scan -- nothing per
// position is stored.  gcc -O2 -o frag_pages frag_pages.c ; ./frag_pages file [seg_bytes]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef uint8_t u8; typedef uint32_t u32; typedef uint64_t u64; typedef int64_t i64;
#define MA 314159265u
#define MB 271828182u
#define PG 4096
#define WARM 8192

static u32 minf = 4096, maxf = 520192, T = 1u << 16;
static const u8* d; static i64 n;
typedef struct { u64 m[4]; u8 last[256]; } Page;
static Page* pages; static i64 npages;
static i64* trig; static i64 ntrig;             // global triggers, ascending (concatenation of the lanes' lists)
static u64 st_scan, st_exact, st_events, st_disturb, st_back;

static inline int gctx(i64 p) { return p ? d[p - 1] : 0; }
static inline int has(const u64* m, int v) { return (int)((m[v >> 6] >> (v & 63)) & 1); }
static inline void add(u64* m, int v) { m[v >> 6] |= 1ull << (v & 63); }

static size_t serial(i64* cuts) {
  size_t nc = 0; u8 o1[256]; memset(o1, 0, 256); u32 h = 0, c1 = 0, sz = 0;
  for (i64 p = 0; p < n; ++p) {
    const u32 c = d[p];
    h = (h + c + 1) * (c == o1[c1] ? MA : MB); o1[c1] = (u8)c; c1 = c; ++sz;
    if (sz >= maxf || (h < T && sz >= minf) || p + 1 == n) { cuts[nc++] = p; memset(o1, 0, 256); h = 0; c1 = 0; sz = 0; }
  }
  return nc;
}

// ---- pass P -----------------------------------------------------------------------------------------
static void pass_pages(void) {
  npages = (n + PG - 1) / PG;
  pages = (Page*)calloc((size_t)npages + 1, sizeof(Page));
  for (i64 j = 0; j < npages; ++j) {
    const i64 hi = (j + 1) * PG < n ? (j + 1) * PG : n;
    for (i64 p = j * PG; p < hi; ++p) { const int v = gctx(p); add(pages[j].m, v); pages[j].last[v] = d[p]; }
  }
}
// ---- pass M: global table before page boundary x*PG --------------------------------------------------------
static void table_before_page(i64 x, u8* tab) {     // (lanes ask in ascending order: the merge is kept running)
  static u8 run[256]; static i64 upto = 0;
  if (x < upto) { memset(run, 0, 256); upto = 0; }
  for (; upto < x; ++upto) for (int v = 0; v < 256; ++v) if (has(pages[upto].m, v)) run[v] = pages[upto].last[v];
  memcpy(tab, run, 256);
}
// ---- pass S ---------------------------------------------------------------------------------------------
static void pass_lanes(i64 seg) {
  trig = (i64*)malloc((size_t)(n + 1) * sizeof(i64)); ntrig = 0;
  for (i64 g = 0; g < n; g += seg) {
    const i64 e = g + seg < n ? g + seg : n;
    i64 w = g - WARM; if (w < 0) w = 0; w = w / PG * PG;
    u8 tab[256]; table_before_page(w / PG, tab);
    u32 h = 0, c1 = (u32)gctx(w), misp = 0;
    int conv = w == 0;                       // from the file start the chain IS the global chain
    for (i64 p = w; p < e; ++p) {
      const u32 c = d[p];
      const int f = c == tab[c1];
      h = (h + c + 1) * (f ? MA : MB); tab[c1] = (u8)c; c1 = c;
      if (!f && ++misp >= 32) conv = 1;
      if (p >= g) {
        if (!conv) { fprintf(stderr, "lane at %lld: hash not converged at its segment start (all-predicted warm-up): "
                                     "the GPU version marks the zone for exact evaluation; the model stops here\n", (long long)g); exit(3); }
        if (h < T) trig[ntrig++] = p;
      }
    }
  }
}

// ---- lookups of pass T ------------------------------------------------------------------------------------
// successor of the last position q in [lo, p) with gctx(q)==v (true-scope callers pass lo = S+1), or -1
static int prev_succ(int v, i64 lo, i64 p) {
  if (p <= lo) return -1;
  i64 j = (p - 1) / PG;
  // partial page: scan backwards inside page j
  for (i64 q = p - 1; q >= j * PG && q >= lo; --q) { ++st_back; if (gctx(q) == v) return d[q]; }
  for (--j; j >= 0 && (j + 1) * PG > lo; --j) {
    if (!has(pages[j].m, v)) continue;
    if (j * PG >= lo) return pages[j].last[v];                 // whole page in scope: its summary answers
    for (i64 q = (j + 1) * PG - 1; q >= lo; --q) { ++st_back; if (gctx(q) == v) return d[q]; }   // page that contains lo
    return -1;
  }
  return -1;
}
static u32 pred_global(i64 p) { const int r = p ? prev_succ(gctx(p), 0, p) : -1; return r < 0 ? 0u : (u32)r; }
static u32 pred_true(i64 S, i64 p) {
  if (p == S) return 0;
  const int v = d[p - 1];
  const int r = prev_succ(v, S + 1, p);
  if (r >= 0) return (u32)r;
  return v == 0 ? d[S] : 0u;                 // the byte at S was recorded under context 0
}
// global hash at position p (p >= 0), re-derived from the 32 unpredicted bytes before it
static u32 hash_global_at(i64 p) {
  i64 t = p; u32 misp = 0;
  while (t >= 0 && misp < 32) { if (d[t] != pred_global(t)) ++misp; --t; }
  u32 h = 0;
  for (i64 q = t + 1; q <= p; ++q) { const u32 c = d[q]; h = (h + c + 1) * (c == pred_global(q) ? MA : MB); }
  return h;
}
static i64 next_trigger(i64 from) {          // first global trigger >= from, or a huge value
  i64 lo = 0, hi = ntrig;
  while (lo < hi) { i64 m = (lo + hi) / 2; if (trig[m] < from) lo = m + 1; else hi = m; }
  return lo < ntrig ? trig[lo] : (i64)1 << 62;
}

// ---- pass T -------------------------------------------------------------------------------------------------
static size_t pass_stitch(i64* cuts) {
  size_t nc = 0; i64 S = 0;
  while (S < n) {
    const i64 minpos = S + minf - 1, maxpos = S + maxf - 1, endp = maxpos < n - 1 ? maxpos : n - 1;
    u64 seen[4] = {0, 0, 0, 0};            // true contexts seen since S
    i64 p = S, cut = -1; u32 h = 0; int exact = 1, since = 0;
    while (cut < 0) {
      if (exact) {
        const int v = p == S ? 0 : d[p - 1];
        const u32 pt = pred_true(S, p), c = d[p];
        const int f = c == pt;
        int disturbed = p == S;
        if (!has(seen, v)) { add(seen, v); ++st_events; if (p != S && f != (c == pred_global(p))) disturbed = 1; }
        h = (h + c + 1) * (f ? MA : MB); ++st_exact;
        if (disturbed) { since = 0; ++st_disturb; } else if (!f) ++since;
        if (p >= maxpos || (h < T && p >= minpos) || p + 1 == n) { cut = p; break; }
        ++p;
        if (since >= 32) exact = 0;
      } else {
        // in step: advance to the end of this page (or the fragment), looking for a new context or a trigger
        const i64 j = p / PG;
        i64 lim = (j + 1) * PG - 1; if (lim > endp) lim = endp;
        int fresh = 0;
        if (p == j * PG) {                   // whole page ahead: its summary tells whether anything is new
          for (int k = 0; k < 4; ++k) if (pages[j].m[k] & ~seen[k]) fresh = 1;
        } else fresh = 1;                    // partial page: look
        i64 e = -1;
        if (fresh) { for (i64 q = p; q <= lim; ++q) { ++st_scan; if (!has(seen, d[q - 1])) { e = q; break; } } }
        const i64 upto = e >= 0 ? e - 1 : lim;
        const i64 tr = next_trigger(p > minpos ? p : minpos);
        if (tr <= upto) { cut = tr; break; }
        if (e < 0) {
          if (lim == endp) { cut = endp; break; }
          p = lim + 1; continue;             // (nothing new in [p, lim]: `seen` is unchanged)
        }
        // first occurrence of context d[e-1] since S
        const int v = d[e - 1]; add(seen, v); ++st_events;
        const u32 c = d[e];
        const u32 tp = v == 0 ? d[S] : 0u;   // fresh table (only the byte at S was recorded, under context 0)
        const int tf = c == tp, gf = c == pred_global(e);
        if (tf == gf) {                      // same multiplier: still in step
          if (e >= endp) { cut = endp; break; }     // (a trigger at e itself was tested above only up to e-1)
          const i64 tr2 = next_trigger(e > minpos ? e : minpos);
          if (tr2 == e) { cut = e; break; }
          p = e + 1; continue;
        }
        ++st_disturb;
        h = hash_global_at(e - 1);           // true hash == global hash up to e-1
        h = (h + c + 1) * (tf ? MA : MB); ++st_exact;
        since = 0;
        if (e >= maxpos || (h < T && e >= minpos) || e + 1 == n) { cut = e; break; }
        p = e + 1; exact = 1;
      }
    }
    cuts[nc++] = cut;
    S = cut + 1;
  }
  return nc;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: frag_pages file [seg_bytes [minf maxf log2T]]\n"); return 2; }
  const i64 seg = argc > 2 ? atoll(argv[2]) : (1 << 18);
  if (argc > 5) { minf = (u32)atoi(argv[3]); maxf = (u32)atoi(argv[4]); T = 1u << atoi(argv[5]); }
  FILE* f = fopen(argv[1], "rb"); if (!f) return 3;
  fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
  u8* buf = (u8*)malloc((size_t)n + 1); if (fread(buf, 1, (size_t)n, f) != (size_t)n) return 4; fclose(f); d = buf;
  i64* c1 = (i64*)malloc(((size_t)n / minf + 2) * sizeof(i64) * 2 + 64); i64* c2 = c1 + n / minf + 2;
  const size_t n1 = serial(c1);
  pass_pages();
  pass_lanes(seg);
  const size_t n2 = pass_stitch(c2);
  const int ok = n1 == n2 && memcmp(c1, c2, n1 * sizeof(i64)) == 0;
  printf("%s: %lld bytes, %zu fragments, %s; stitch touched: page scans %.2f%%, exact %.2f%%, backward scans %.2f%% of the bytes; "
         "%.1f first occurrences and %.2f disturbances per fragment; %lld global triggers\n", argv[1], (long long)n, n1,
         ok ? "IDENTICAL" : "MISMATCH", 100.0 * st_scan / (n ? n : 1), 100.0 * st_exact / (n ? n : 1), 100.0 * st_back / (n ? n : 1),
         (double)st_events / (n1 ? n1 : 1), (double)st_disturb / (n1 ? n1 : 1), (long long)ntrig);
  if (!ok) for (size_t i = 0; i < n1 && i < n2; ++i) if (c1[i] != c2[i]) { printf("first difference at fragment %zu: %lld vs %lld\n", i, (long long)c1[i], (long long)c2[i]); break; }
  return ok ? 0 : 1;
}
