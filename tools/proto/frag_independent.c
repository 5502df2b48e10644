// CPU prototype of the start-independent fragmenter: validates, against the plain serial
// loop, that cut positions can be derived from
//   (1) quantities that do NOT depend on where fragments start -- the "global" chain: o1[] never reset, hash never
//       reset; its trigger positions (hg < T) are what a uniform, crossing-free GPU pass would produce per segment
//       (the global hash depends on the last 32 mispredicted bytes only, so any lane can compute it after a warm-up);
//   (2) a per-fragment correction: the true chain differs from the global one only in "disturbance windows" -- after
//       the fragment start (hash reset) and after the first occurrence of each context byte inside the fragment whose
//       true prediction (fresh table) differs from the global one -- each window closing after 32 mispredictions.
// Not product code: a model of the stitch logic, run on the CPU against the oracle's loop.   gcc -O2 -o frag_proto ...
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef uint8_t u8; typedef uint32_t u32; typedef uint64_t u64; typedef int64_t i64;
#define MA 314159265u
#define MB 271828182u

static u32 minf = 4096, maxf = 520192, T = 1u << 16;

// reference loop: fragment ends (offset of last byte) of a file
static size_t serial(const u8* d, size_t n, i64* cuts) {
  size_t nc = 0; u8 o1[256]; memset(o1, 0, 256); u32 h = 0, c1 = 0, sz = 0;
  for (size_t p = 0; p < n; ++p) {
    const u32 c = d[p];
    h = (h + c + 1) * (c == o1[c1] ? MA : MB); o1[c1] = (u8)c; c1 = c; ++sz;
    if (sz >= maxf || (h < T && sz >= minf) || p + 1 == n) { cuts[nc++] = (i64)p; memset(o1, 0, 256); h = 0; c1 = 0; sz = 0; }
  }
  return nc;
}

typedef struct { const u8* d; size_t n; u32* hg; u8* gflag; u32* M; i64* lo; /* lo[p] = previous q<p with d[q-1]==d[p-1] (p>=1), else -1 */
                 i64* trig; size_t ntrig; i64* occ[256]; size_t nocc[256]; } Glob;

static void global_pass(Glob* G) {
  const u8* d = G->d; const size_t n = G->n;
  u8 o1[256]; i64 last[256]; memset(o1, 0, 256); for (int i = 0; i < 256; ++i) last[i] = -1;
  u32 h = 0, c1 = 0, m = 0; G->ntrig = 0;
  for (size_t p = 0; p < n; ++p) {
    const u32 c = d[p];
    const int f = c == o1[c1];
    h = (h + c + 1) * (f ? MA : MB); G->hg[p] = h; G->gflag[p] = (u8)f; m += !f; G->M[p] = m;
    G->lo[p] = p ? last[c1] : -1;      // position 0 has no real context
    if (p) last[c1] = (i64)p;
    o1[c1] = (u8)c; c1 = c;
    if (h < T) G->trig[G->ntrig++] = (i64)p;
  }
  // occurrence lists: occ[v] = positions p>=1 with d[p-1]==v, ascending
  size_t cnt[256] = {0};
  for (size_t p = 1; p < n; ++p) ++cnt[d[p - 1]];
  for (int v = 0; v < 256; ++v) { G->occ[v] = (i64*)malloc((cnt[v] + 1) * sizeof(i64)); G->nocc[v] = 0; }
  for (size_t p = 1; p < n; ++p) { const int v = d[p - 1]; G->occ[v][G->nocc[v]++] = (i64)p; }
}

static i64 lower_bound(const i64* a, size_t n, i64 x) { size_t lo = 0, hi = n; while (lo < hi) { size_t m = (lo + hi) / 2; if (a[m] < x) lo = m + 1; else hi = m; } return (i64)lo; }

// true prediction at position p of the fragment that starts at S (p > S): successor of the last position q in (S, p)
// with the same context; the byte at S was recorded under context 0
static u32 true_pred(const Glob* G, i64 S, i64 p) {
  const u8* d = G->d;
  const int v = d[p - 1];
  i64 q = G->lo[p];                         // previous position with context v (real context), or -1
  if (q > S) return d[q];
  // q <= S: position S itself (if d[S-1]==v) was recorded under context 0, not v
  if (v == 0) {
    // candidates with true context 0: position S, and positions q in (S,p) with d[q-1]==0 -- none of the latter (lo<=S)
    return d[S];
  }
  return 0;
}

static size_t stitched(const Glob* G, i64* cuts, u64* exact_bytes, u64* events_total) {
  const u8* d = G->d; const i64 n = (i64)G->n;
  size_t nc = 0; i64 S = 0;
  while (S < n) {
    const i64 minpos = S + (i64)minf - 1, maxpos = S + (i64)maxf - 1;
    // disturbance events of this fragment: first occurrence after S of every context value whose true flag differs
    i64 ev[257]; int nev = 0;
    for (int v = 0; v < 256; ++v) {
      const i64 k = lower_bound(G->occ[v], G->nocc[v], S + 1);
      if ((size_t)k >= G->nocc[v]) continue;
      const i64 e = G->occ[v][k];
      if (e > maxpos) continue;
      const u32 tp = (v == 0) ? d[S] : 0u;
      const int tf = d[e] == tp;
      if (tf != G->gflag[e]) ev[nev++] = e;
    }
    // sort events
    for (int i = 1; i < nev; ++i) { i64 x = ev[i]; int j = i - 1; while (j >= 0 && ev[j] > x) { ev[j + 1] = ev[j]; --j; } ev[j + 1] = x; }
    *events_total += (u64)nev;
    i64 cut = -1;
    // Nothing before minpos can cut, so the chain only has to be right from minpos on.  Walk back from the last
    // disturbance before minpos (the fragment start counts as one) while the previous window had not closed yet
    // (fewer than 32 mispredictions in between: between two consecutive events the true flags ARE the global ones,
    // so the global misprediction counter M measures it).  Exact evaluation starts at that event, from the global
    // hash just before it; if even the last window closes before minpos, no evaluation is needed at all.
    i64 evs[258]; int nes = 0; evs[nes++] = S;
    for (int i = 0; i < nev; ++i) if (ev[i] > S) evs[nes++] = ev[i];
    int last = 0;
    while (last + 1 < nes && evs[last + 1] < minpos && evs[last + 1] < n) ++last;
    i64 p = S; u32 h = 0; int since = 0; int exact = 1; int ei = 0;
    {
      const i64 upto = (minpos < n ? minpos : n) - 1;           // last position that cannot cut
      int j = last;
      if (upto >= evs[j] && G->M[upto] - G->M[evs[j]] >= 32) {
        // in step with the global chain at minpos already
        exact = 0; p = upto + 1;
        while (ei < nev && ev[ei] <= upto) ++ei;
      } else {
        while (j > 0 && G->M[evs[j] - 1] - G->M[evs[j - 1]] < 32) --j;
        p = evs[j]; h = j ? G->hg[p - 1] : 0u;
        while (ei < nev && ev[ei] < p) ++ei;
      }
    }
    while (cut < 0) {
      if (exact) {
        // evaluate position p exactly
        u32 pred;
        if (p == S) pred = 0; else pred = true_pred(G, S, p);
        const u32 c = d[p];
        const int f = c == pred;
        h = (h + c + 1) * (f ? MA : MB);
        ++*exact_bytes;
        int disturbed = p == S;
        while (ei < nev && ev[ei] <= p) { if (ev[ei] == p) disturbed = 1; ++ei; }
        if (disturbed) since = 0;            // the window is counted AFTER the disturbed position
        else if (!f) ++since;
        if (p >= maxpos || (h < T && p >= minpos) || p + 1 == n) { cut = p; break; }
        if (since >= 32) {
          if (h != G->hg[p]) { fprintf(stderr, "model violated: window closed at %lld but h != hg\n", (long long)p); exit(2); }
          exact = 0;
        }
        ++p;
      } else {
        // in step with the global chain from p on: next interesting position = next event or next global trigger >= minpos
        const i64 ne = ei < nev ? ev[ei] : (i64)1 << 62;
        i64 from = p > minpos ? p : minpos;
        const i64 k = lower_bound(G->trig, G->ntrig, from);
        i64 tr = (size_t)k < G->ntrig ? G->trig[k] : (i64)1 << 62;
        i64 endp = maxpos < n - 1 ? maxpos : n - 1;
        if (tr < ne && tr <= endp) { cut = tr; break; }
        if (ne > endp) { cut = endp; break; }
        // resume exact evaluation at the event, from the global hash just before it
        p = ne; h = G->hg[p - 1]; exact = 1; since = 0;
      }
    }
    cuts[nc++] = cut;
    S = cut + 1;
  }
  return nc;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: frag_proto file [minf maxf log2T]\n"); return 2; }
  if (argc >= 5) { minf = (u32)atoi(argv[2]); maxf = (u32)atoi(argv[3]); T = 1u << atoi(argv[4]); }
  FILE* f = fopen(argv[1], "rb"); if (!f) return 3;
  fseek(f, 0, SEEK_END); size_t n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
  u8* d = (u8*)malloc(n + 1); if (fread(d, 1, n, f) != n) return 4; fclose(f);
  i64* c1 = (i64*)malloc((n / minf + 2) * sizeof(i64) * 2 + 64); i64* c2 = c1 + n / minf + 2;
  size_t n1 = serial(d, n, c1);
  Glob G; G.d = d; G.n = n; G.hg = (u32*)malloc(4 * n + 4); G.gflag = (u8*)malloc(n + 1); G.M = (u32*)malloc(4 * n + 4);
  G.lo = (i64*)malloc(8 * n + 8); G.trig = (i64*)malloc(8 * n + 8);
  global_pass(&G);
  u64 exact = 0, events = 0;
  size_t n2 = stitched(&G, c2, &exact, &events);
  int ok = n1 == n2 && memcmp(c1, c2, n1 * sizeof(i64)) == 0;
  printf("%s: %zu bytes, %zu fragments, %s; exact evaluation %.3f%% of bytes, %.2f disturbance events per fragment, %zu global triggers\n",
         argv[1], n, n1, ok ? "IDENTICAL" : "MISMATCH", 100.0 * (double)exact / (double)(n ? n : 1), (double)events / (double)(n1 ? n1 : 1), G.ntrig);
  if (!ok) { for (size_t i = 0; i < n1 && i < n2; ++i) if (c1[i] != c2[i]) { printf("first difference at fragment %zu: %lld vs %lld\n", i, (long long)c1[i], (long long)c2[i]); break; } }
  return ok ? 0 : 1;
}
