// CPU model, stage 3 (see README.md): the per-file chain of the crossing-free fragmenter as a DUAL-TABLE WALKER -- no
// backward scans.  Inputs are what the lane passes deliver for the never-reset ("global") chain of a file:
//   * per 4 KiB page: the set of context bytes present and the last successor of each (pass P),
//   * the positions where the global hash is below the threshold (triggers) and the global hash at every page start
//     (pass S: equal-work lanes, exact o1[] table at their segment start from the running merge of pass P, hash warmed up).
// The walker goes through the file fragment by fragment keeping two 256-entry tables current at its position: TRUE (reset
// at the fragment start) and GLOBAL (never reset).  D = contexts on which they differ; D only shrinks inside a fragment.
//   exact mode: byte by byte from the fragment start (true hash from 0), flags of both tables compared; after 32
//               unpredicted bytes without a difference the true hash equals the global one ("in step").
//   skip mode:  at a page boundary, in step, and the page holds no context of D: the page cannot make the chains differ --
//               look up the next global trigger in it, else merge its summary into both tables and go on.
//               A page that holds a context of D is walked exactly from its start (hash = stored global hash there).
// gcc -O2 -o frag_dual frag_dual.c ; ./frag_dual file [seg_bytes [minf maxf log2T]]   -> IDENTICAL + how much was walked
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
static u32* hpage;                               // global hash BEFORE the first byte of every page
static i64* trig; static i64 ntrig;
static u64 st_exact, st_pages_skipped, st_pages_walked;

static inline int gctx(i64 p) { return p ? d[p - 1] : 0; }
static inline int has(const u64* m, int v) { return (int)((m[v >> 6] >> (v & 63)) & 1); }
static inline void add(u64* m, int v) { m[v >> 6] |= 1ull << (v & 63); }
static inline void del(u64* m, int v) { m[v >> 6] &= ~(1ull << (v & 63)); }

static size_t serial(i64* cuts) {
  size_t nc = 0; u8 o1[256]; memset(o1, 0, 256); u32 h = 0, c1 = 0, sz = 0;
  for (i64 p = 0; p < n; ++p) {
    const u32 c = d[p];
    h = (h + c + 1) * (c == o1[c1] ? MA : MB); o1[c1] = (u8)c; c1 = c; ++sz;
    if (sz >= maxf || (h < T && sz >= minf) || p + 1 == n) { cuts[nc++] = p; memset(o1, 0, 256); h = 0; c1 = 0; sz = 0; }
  }
  return nc;
}

static void pass_pages(void) {
  npages = (n + PG - 1) / PG;
  pages = (Page*)calloc((size_t)npages + 1, sizeof(Page));
  for (i64 j = 0; j < npages; ++j) {
    const i64 hi = (j + 1) * PG < n ? (j + 1) * PG : n;
    for (i64 p = j * PG; p < hi; ++p) { const int v = gctx(p); add(pages[j].m, v); pages[j].last[v] = d[p]; }
  }
}
static void table_before_page(i64 x, u8* tab) {
  static u8 run[256]; static i64 upto = 0;
  if (x < upto) { memset(run, 0, 256); upto = 0; }
  for (; upto < x; ++upto) for (int v = 0; v < 256; ++v) if (has(pages[upto].m, v)) run[v] = pages[upto].last[v];
  memcpy(tab, run, 256);
}
// lanes: global triggers and the global hash at page starts.  A lane whose hash has not converged by its segment start
// (an all-predicted warm-up) cannot vouch for its triggers: the model marks those pages "unknown" (hpage valid flag off)
// and the walker treats them as pages to walk exactly -- the GPU version does the same.
static u8* hvalid;
static void pass_lanes(i64 seg) {
  trig = (i64*)malloc((size_t)(n + 1) * sizeof(i64)); ntrig = 0;
  hpage = (u32*)calloc((size_t)npages + 1, 4); hvalid = (u8*)calloc((size_t)npages + 1, 1);
  for (i64 g = 0; g < n; g += seg) {
    const i64 e = g + seg < n ? g + seg : n;
    i64 w = g - WARM; if (w < 0) w = 0; w = w / PG * PG;
    u8 tab[256]; table_before_page(w / PG, tab);
    u32 h = 0, c1 = (u32)gctx(w), misp = 0;
    int conv = w == 0;
    for (i64 p = w; p < e; ++p) {
      if (p >= g && p % PG == 0) { hpage[p / PG] = h; hvalid[p / PG] = (u8)conv; }
      const u32 c = d[p];
      const int f = c == tab[c1];
      h = (h + c + 1) * (f ? MA : MB); tab[c1] = (u8)c; c1 = c;
      if (!f && ++misp >= 32) conv = 1;
      if (p >= g && conv && h < T) trig[ntrig++] = p;
      if (p >= g && !conv) hvalid[p / PG] = 0;            // triggers of this page are not all known
    }
  }
}
static i64 next_trigger(i64 from) {
  i64 lo = 0, hi = ntrig;
  while (lo < hi) { i64 m = (lo + hi) / 2; if (trig[m] < from) lo = m + 1; else hi = m; }
  return lo < ntrig ? trig[lo] : (i64)1 << 62;
}

static size_t pass_walker(i64* cuts) {
  size_t nc = 0; i64 S = 0;
  u8 G[256]; memset(G, 0, 256);                  // global table at the walker's position
  u32 hg = 0;                                    // global hash at the walker's position (tracked in exact mode only)
  while (S < n) {
    const i64 minpos = S + minf - 1, maxpos = S + maxf - 1, endp = maxpos < n - 1 ? maxpos : n - 1;
    u8 Tt[256]; memset(Tt, 0, 256);
    u64 D[4] = {0, 0, 0, 0};
    for (int v = 0; v < 256; ++v) if (G[v]) add(D, v);
    i64 p = S, cut = -1; u32 h = 0; int since = 0, exact = 1, hg_known = 0;
    u32 ct = 0;                                   // true context: 0 at the fragment start
    (void)hg_known;
    while (cut < 0) {
      if (exact) {
        const u32 c = d[p];
        const u32 cg = (u32)gctx(p);
        const int ft = c == Tt[ct], fg = c == G[cg];
        h = (h + c + 1) * (ft ? MA : MB);
        if (hg_known) hg = (hg + c + 1) * (fg ? MA : MB);
        Tt[ct] = (u8)c; G[cg] = (u8)c;
        if (ct == cg) del(D, (int)ct); else { if (Tt[cg] != G[cg]) add(D, (int)cg); }   // (only at p == S the contexts differ)
        if (ft != fg || p == S) since = 0; else if (!ft) ++since;
        ++st_exact;
        if (p >= maxpos || (h < T && p >= minpos) || p + 1 == n) { cut = p; break; }
        ct = c; ++p;
        if (since >= 32 && p % PG == 0 && hvalid[p / PG]) exact = 0;   // in step at a page boundary a lane vouches for: try to skip
      } else {
        const i64 j = p / PG;                                // p is a page start
        // (hvalid[j] holds here.)  The page after this one must be vouched for as well, or this page is walked exactly so
        // that the walker arrives there with a hash of its own
        int clash = (j + 1 < npages && !hvalid[j + 1]);
        for (int k = 0; k < 4 && !clash; ++k) if (pages[j].m[k] & D[k]) clash = 1;
        const i64 lim = (j + 1) * PG - 1 < endp ? (j + 1) * PG - 1 : endp;
        if (!clash) {
          const i64 tr = next_trigger(p > minpos ? p : minpos);
          if (tr <= lim) { cut = tr; }
          else if (lim == endp) { cut = endp; }
          if (cut >= 0) {                                    // the global table must be current up to the cut for the next fragment
            for (i64 q = p; q <= cut; ++q) G[gctx(q)] = d[q];
            break;
          }
          for (int v = 0; v < 256; ++v) if (has(pages[j].m, v)) { G[v] = pages[j].last[v]; Tt[v] = pages[j].last[v]; }
          ++st_pages_skipped;
          p = lim + 1; ct = d[p - 1];
        } else {                                             // walk this page exactly, true hash = global hash at its start
          h = hpage[j];
          ++st_pages_walked;
          exact = 1; since = 32;                             // in step on entry; any flag difference resets it
        }
      }
    }
    cuts[nc++] = cut;
    S = cut + 1;
    (void)hg;
  }
  return nc;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: frag_dual file [seg_bytes [minf maxf log2T]]\n"); return 2; }
  const i64 seg = argc > 2 ? atoll(argv[2]) : (1 << 18);
  if (argc > 5) { minf = (u32)atoi(argv[3]); maxf = (u32)atoi(argv[4]); T = 1u << atoi(argv[5]); }
  FILE* f = fopen(argv[1], "rb"); if (!f) return 3;
  fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
  u8* buf = (u8*)malloc((size_t)n + 1); if (fread(buf, 1, (size_t)n, f) != (size_t)n) return 4; fclose(f); d = buf;
  i64* c1 = (i64*)malloc(((size_t)n / minf + 2) * sizeof(i64) * 2 + 64); i64* c2 = c1 + n / minf + 2;
  const size_t n1 = serial(c1);
  pass_pages();
  pass_lanes(seg);
  const size_t n2 = pass_walker(c2);
  const int ok = n1 == n2 && memcmp(c1, c2, n1 * sizeof(i64)) == 0;
  printf("%s: %lld bytes, %zu fragments, %s; walker: %.2f%% of the bytes walked exactly, %llu pages skipped, %llu pages entered exactly\n", argv[1],
         (long long)n, n1, ok ? "IDENTICAL" : "MISMATCH", 100.0 * st_exact / (n ? n : 1), (unsigned long long)st_pages_skipped, (unsigned long long)st_pages_walked);
  if (!ok) for (size_t i = 0; i < n1 && i < n2; ++i) if (c1[i] != c2[i]) { printf("first difference at fragment %zu: %lld vs %lld\n", i, (long long)c1[i], (long long)c2[i]); break; }
  return ok ? 0 : 1;
}
