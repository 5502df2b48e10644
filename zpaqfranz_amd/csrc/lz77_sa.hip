// Suffix array and the LZ77 parse that uses it -- what zpaqfranz -m2 runs (SA branch).
// Reference: divsufsort (ZSFX/libzpaq.cpp:6047-6072, body :4334-6040) called from LZBuffer::LZBuffer (:6296-6310), and
// the candidate search of LZBuffer::fill (:6339-6372) with the decision / literal bookkeeping at :6412-6453.
//
// The reference is a serial program twice over: induced sorting, then a greedy parse that asks the suffix array for
// neighbours at every position it visits.  Neither maps to a GPU as written; what is kept is the RESULT of each:
//
//  * The suffix array of a string is unique, so it is built by prefix doubling with discarding: suffixes are sorted by
//    their first 8 bytes (one 64-bit radix sort of (key, position) pairs), then in rounds h = 8, 16, ... only the
//    suffixes whose group is still ambiguous are re-sorted by (group, rank of the suffix h bytes on).  A suffix that
//    runs past the end in a round gets its length as second key, below every in-range rank: the shorter one first.
//    Singletons are final and leave the working set, which on text shrinks ~3x per round.  The sort is rocPRIM's
//    device radix sort (the only library call on the path); the keys, heads, compaction and rank scatter are here.
//    When the last group splits, rank[] is the inverse suffix array.
//  * LCP of neighbouring suffixes, capped at 65535, by Kasai's recurrence over chunks of 128 text positions per lane
//    (a chunk restarts from 0; the cap bounds what a restart can cost).  With it the match length against the k-th
//    neighbour of a suffix is a running minimum -- no byte compares, however repetitive the block is.
//  * The candidate search of fill() reads only the input, SA and ISA and one bit of parse state (lit == 0, which
//    changes a score term when a lookahead match begins with a literal).  So every position's decision is computed
//    up front, for both values of that bit, one lane per position (lz77_sa_candidates_kernel), exactly as :6346-6371
//    orders, breaks and scores the candidates.  isa[] there is a window of 2^(17+args[0]) positions rebuilt on entry:
//    the one observable effect -- no lookahead search from the last position of a window -- is reproduced.
//  * The serial part that remains is the chain "take the match and skip, or count a literal": one wave per block walks
//    the precomputed decisions (LDS-staged, 64 positions per register window, v_readlane per step) and writes the
//    token list; tokens become bits in the pack kernels of lz77_enc.hip.
// All of it is integer / byte work on HBM-resident arrays (≈48 B of working set per input byte during the sort).
#include <algorithm>
#include <cstring>
#include <stdlib.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "zpq_internal.h"

namespace {

constexpr u32 kMaxMatch = (1u << 14) * 3;    // ZSFX/libzpaq.cpp:6258
constexpr u32 kMaxLiteral = (1u << 14) / 4;  // :6259
constexpr u32 kLcpCap = 65535;
constexpr u32 kLcpChunk = 128;
constexpr u32 kWalkWin = 2048;               // positions staged in LDS per refill of the walk

__device__ __forceinline__ int lg32(u32 x) { return x ? 32 - __builtin_clz(x) : 0; }  // lg(), :6233-6242
u32 lg32_host(u64 x) { u32 r = 0; while (x) { ++r; x >>= 1; } return r; }
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

// ---- suffix array -----------------------------------------------------------------------------------------------------

// round 0 keys: the first 8 bytes, big-endian, zero-padded past the end
__global__ __launch_bounds__(256) void sa_keys0_kernel(const u8* __restrict__ in, u32 n, u64* __restrict__ key, u32* __restrict__ val,
                                                       u32* __restrict__ pos) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  u64 k = 0;
  if (i + 8 <= n) k = __builtin_bswap64(*(const u64_u*)(in + i));
  else for (u32 j = 0; j < 8; ++j) k = (k << 8) | (i + j < n ? in[i + j] : 0u);
  key[i] = k; val[i] = i; pos[i] = i;
}

// group heads of the sorted working set: headpos = slot of the element when it opens a group (for the max-scan that
// gives every element its group's first slot), kg = (kept, kept head) counters for the sum-scan
__global__ __launch_bounds__(256) void sa_heads_kernel(const u64* __restrict__ key, const u32* __restrict__ pos, u32 m, u32* __restrict__ headpos,
                                                       u64* __restrict__ kg) {
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  if (t >= m) return;
  const u64 k = key[t];
  const bool head = t == 0 || key[t - 1] != k;
  const bool next_head = t + 1 == m || key[t + 1] != k;
  const bool keep = !(head && next_head);
  headpos[t] = head ? pos[t] : 0u;
  kg[t] = (keep ? 1ull : 0ull) | ((keep && head) ? 1ull << 32 : 0ull);
}

// writes the order and the ranks of this round, and compacts the elements whose group is still ambiguous
__global__ __launch_bounds__(256) void sa_apply_kernel(const u64* __restrict__ key, const u32* __restrict__ val, const u32* __restrict__ pos, u32 m,
                                                       const u32* __restrict__ gstart, const u64* __restrict__ kg_incl, u32* __restrict__ sa,
                                                       u32* __restrict__ rank, u32* __restrict__ val2, u32* __restrict__ pos2,
                                                       u32* __restrict__ gid2) {
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  if (t >= m) return;
  const u32 v = val[t], p = pos[t];
  sa[p] = v;
  rank[v] = gstart[t];
  const u64 k = key[t];
  const bool head = t == 0 || key[t - 1] != k;
  const bool next_head = t + 1 == m || key[t + 1] != k;
  if (head && next_head) return;
  const u64 c = kg_incl[t];
  const u32 idx = (u32)c - 1u;                 // inclusive count of kept elements, this one included
  val2[idx] = v; pos2[idx] = p; gid2[idx] = (u32)(c >> 32) - 1u;
}

// keys of a doubling round: (dense group number, rank of the suffix h bytes on), or the length for a suffix that ends first
__global__ __launch_bounds__(256) void sa_keys_kernel(const u32* __restrict__ val, const u32* __restrict__ gid, u32 m, u32 n, u32 h, u32 b2,
                                                      const u32* __restrict__ rank, u64* __restrict__ key) {
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  if (t >= m) return;
  const u32 i = val[t];
  const u64 ih = (u64)i + h;
  const u32 k2 = ih < n ? rank[ih] + h + 1u : n - i;
  key[t] = ((u64)gid[t] << b2) | k2;
}

struct SaWork {
  u64* key[2]; u32* val[2]; u32* pos[2]; u32* gid; u32* headpos; u64* kg; void* tmp; size_t tmp_bytes;
};

size_t sa_temp_bytes(u32 n) {
  size_t a = 0, b = 0, c = 0;
  (void)rocprim::radix_sort_pairs(nullptr, a, (u64*)nullptr, (u64*)nullptr, (u32*)nullptr, (u32*)nullptr, (size_t)n, 0u, 64u, (hipStream_t)0);
  (void)rocprim::inclusive_scan(nullptr, b, (u32*)nullptr, (u32*)nullptr, (size_t)n, rocprim::maximum<u32>(), (hipStream_t)0);
  (void)rocprim::inclusive_scan(nullptr, c, (u64*)nullptr, (u64*)nullptr, (size_t)n, rocprim::plus<u64>(), (hipStream_t)0);
  return std::max(a, std::max(b, c)) + 256;
}

size_t sa_work_bytes(u32 n) {
  const size_t m = ((size_t)n + 63) & ~(size_t)63;
  return m * (8 * 2 + 4 * 2 + 4 * 2 + 4 + 4 + 8) + sa_temp_bytes(n) + 16 * 256;
}

template <typename T>
T* carve(u8*& p, size_t count) {
  T* r = (T*)p;
  p += (count * sizeof(T) + 255) & ~(size_t)255;
  return r;
}

// d_sa[n], d_rank[n] (the inverse on return); work = sa_work_bytes(n) of device memory
int build_suffix_array(zpq_ctx* ctx, hipStream_t st, const u8* d_in, u32 n, u32* d_sa, u32* d_rank, u8* work, u32* rounds_out) {
  if (rounds_out) *rounds_out = 0;
  if (n == 0) return ZPQ_OK;
  SaWork W;
  u8* p = work;
  W.key[0] = carve<u64>(p, n); W.key[1] = carve<u64>(p, n);
  W.val[0] = carve<u32>(p, n); W.val[1] = carve<u32>(p, n);
  W.pos[0] = carve<u32>(p, n); W.pos[1] = carve<u32>(p, n);
  W.gid = carve<u32>(p, n); W.headpos = carve<u32>(p, n); W.kg = carve<u64>(p, n);
  W.tmp = p; W.tmp_bytes = sa_temp_bytes(n);
  u64* pinned = (u64*)zpq_pinned(ctx, 64);
  if (!pinned) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "pinned staging");

  u32 m = n, h = 8, bits = 64, rounds = 0;
  int kb = 0, vb = 0, pb = 0;      // buffers holding the unsorted keys / values / slots of this round
  ZPQ_LAUNCH(ctx, "sa_keys0_kernel", st, sa_keys0_kernel, dim3((n + 255) / 256), dim3(256), d_in, n, W.key[0], W.val[0], W.pos[0]);
  for (;;) {
    const dim3 grid((m + 255) / 256), blk(256);
    {
      ZpqProfScope prof_scope_(ctx, "sa_radix_sort_pairs", st);
      size_t tb = W.tmp_bytes;
      ZPQ_HIP(ctx, rocprim::radix_sort_pairs(W.tmp, tb, W.key[kb], W.key[kb ^ 1], W.val[vb], W.val[vb ^ 1], (size_t)m, 0u, bits, st));
    }
    kb ^= 1; vb ^= 1;
    ZPQ_LAUNCH(ctx, "sa_heads_kernel", st, sa_heads_kernel, grid, blk, W.key[kb], W.pos[pb], m, W.headpos, W.kg);
    {
      ZpqProfScope prof_scope_(ctx, "sa_scans", st);
      size_t tb = W.tmp_bytes;
      ZPQ_HIP(ctx, rocprim::inclusive_scan(W.tmp, tb, W.headpos, W.headpos, (size_t)m, rocprim::maximum<u32>(), st));
      tb = W.tmp_bytes;
      ZPQ_HIP(ctx, rocprim::inclusive_scan(W.tmp, tb, W.kg, W.kg, (size_t)m, rocprim::plus<u64>(), st));
    }
    ZPQ_LAUNCH(ctx, "sa_apply_kernel", st, sa_apply_kernel, grid, blk, W.key[kb], W.val[vb], W.pos[pb], m, W.headpos, W.kg, d_sa, d_rank,
               W.val[vb ^ 1], W.pos[pb ^ 1], W.gid);
    ZPQ_HIP(ctx, hipGetLastError());
    ZPQ_HIP(ctx, hipMemcpyAsync(pinned, W.kg + (m - 1), 8, hipMemcpyDeviceToHost, st));
    ZPQ_HIP(ctx, hipStreamSynchronize(st));
    ++rounds;
    const u32 kept = (u32)pinned[0], groups = (u32)(pinned[0] >> 32);
    if (kept == 0) break;
    if (h >= (1u << 30)) return zpq_fail(ctx, ZPQ_ERR_HIP, "suffix array did not converge");
    vb ^= 1; pb ^= 1; m = kept;
    const u32 b2 = (u32)lg32_host((u64)n + h + 1);
    const u32 bg = groups > 1 ? (u32)lg32_host(groups - 1) : 0u;
    bits = b2 + bg;
    ZPQ_LAUNCH(ctx, "sa_keys_kernel", st, sa_keys_kernel, dim3((m + 255) / 256), dim3(256), W.val[vb], W.gid, m, n, h, b2, d_rank, W.key[kb]);
    h *= 2;
  }
  if (rounds_out) *rounds_out = rounds;
  return ZPQ_OK;
}

// ---- BWT output (LZBuffer level 3, ZSFX/libzpaq.cpp:6317-6326) ----------------------------------------------------------
// out[0] = last byte, out[j+1] = the byte before suffix sa[j] (255 for the suffix that is the whole block; its 1-based rank
// goes LSB first into the four bytes after the n+1 transformed ones)
__global__ __launch_bounds__(256) void bwt_output_kernel(const u8* __restrict__ in, u32 n, const u32* __restrict__ sa, u8* __restrict__ out) {
  const u32 j = blockIdx.x * 256u + threadIdx.x;
  if (j == 0) out[0] = n ? in[n - 1] : 255;
  if (j >= n) return;
  const u32 s = sa[j];
  if (s == 0) {
    out[j + 1] = 255;
    const u32 idx = j + 1;
    out[n + 1] = (u8)idx; out[n + 2] = (u8)(idx >> 8); out[n + 3] = (u8)(idx >> 16); out[n + 4] = (u8)(idx >> 24);
  } else out[j + 1] = in[s - 1];
}

// ---- LCP of neighbouring suffixes (Kasai), capped ------------------------------------------------------------------
// lcp[q] = min(kLcpCap, longest common prefix of suffixes sa[q-1] and sa[q]); lcp[0] = 0
__global__ __launch_bounds__(256) void sa_lcp_kernel(const u8* __restrict__ in, u32 n, const u32* __restrict__ sa, const u32* __restrict__ isa,
                                                     u16* __restrict__ lcp) {
  const u32 c = blockIdx.x * 256u + threadIdx.x;
  const u64 i0 = (u64)c * kLcpChunk;
  if (i0 >= n) return;
  const u32 i1 = (u32)std::min<u64>(i0 + kLcpChunk, n);
  u32 l = 0;
  for (u32 i = (u32)i0; i < i1; ++i) {
    const u32 q = isa[i];
    if (q == 0) { lcp[0] = 0; l = 0; continue; }
    const u32 j = sa[q - 1];
    const u32 lim = std::min(kLcpCap, n - std::max(i, j));
    while (l + 8 <= lim) {
      const u64 x = *(const u64_u*)(in + i + l) ^ *(const u64_u*)(in + j + l);
      if (x) { l += (u32)__builtin_ctzll(x) >> 3; goto done; }
      l += 8;
    }
    while (l < lim && in[i + l] == in[j + l]) ++l;
  done:
    lcp[q] = (u16)l;
    if (l) --l;
  }
}

// ---- per-position decisions ------------------------------------------------------------------------------------------
struct SaCfg {
  const u8* in; u32 n;
  u32 minMatch, bucket, lookahead, checkbits, level;
};

// final decision record: bit 63 take, bit 62 blit (leading literal), bits 32..47 blen (blit included), bits 0..31 offset
__device__ __forceinline__ u64 make_rec(bool take, u32 blen, u32 blit, u32 off) {
  return take ? (1ull << 63) | ((u64)blit << 62) | ((u64)blen << 32) | off : 0ull;
}

struct Best { u32 blen, bp, blit; int bscore; };

constexpr u32 kTileR = 128;                   // neighbours either side held in LDS (method 2 looks at 127)
constexpr u32 kTile = 256 + 2 * kTileR;

// SA / LCP / preceding byte around the 256 slots of a workgroup, one 8-byte LDS word per slot x:
// sa[x] | lcp[x] << 32 | lcp[x+1] << 48 -- a scan step in either direction is a single ds_read_b64 (the byte before the
// suffix, needed by the lookahead pass only, sits in its own array).  The neighbour scans of adjacent slots overlap
// almost entirely, so the tile is read from HBM once per 256 slots.  INTILE: every neighbour the scan can reach is in
// the tile (bucket < kTileR); otherwise slots outside it are fetched from the global arrays.
struct Tile {
  const u32* sa_g; const u16* lcp_g; const u8* bw_g;
  u32 t0, tn, n;                              // slots [t0, t0+tn) are in LDS
  const u64* pk; const u8* bw_l;
  template <bool INTILE> __device__ __forceinline__ u64 word(u32 qq) const {
    const u32 r = qq - t0;
    if (INTILE || r < tn) return pk[r];
    return (u64)sa_g[qq] | ((u64)lcp_g[qq] << 32) | ((u64)(qq + 1 < n ? lcp_g[qq + 1] : 0) << 48);
  }
  template <bool INTILE> __device__ __forceinline__ u32 bw(u32 qq) const { const u32 r = qq - t0; return (INTILE || r < tn) ? bw_l[r] : bw_g[qq]; }
};

template <bool BW>
__device__ __forceinline__ void load_tile(Tile& T, const u32* sa, const u16* lcp, const u8* bw, u32 n, u32 q0, u64* pk, u8* bw_l) {
  T.sa_g = sa; T.lcp_g = lcp; T.bw_g = bw; T.pk = pk; T.bw_l = bw_l; T.n = n;
  T.t0 = q0 >= kTileR ? q0 - kTileR : 0u;
  const u32 end = (u32)std::min<u64>((u64)q0 + 256 + kTileR, n);
  T.tn = end - T.t0;
  for (u32 t = threadIdx.x; t < T.tn; t += 256) {
    const u32 x = T.t0 + t;
    pk[t] = (u64)sa[x] | ((u64)lcp[x] << 32) | ((u64)(x + 1 < n ? lcp[x + 1] : 0) << 48);
    if (BW) bw_l[t] = bw[x];
  }
  __syncthreads();
}

// One direction of the neighbour scan around SA slot q without lookahead (:6350-6365, h = 0).  The match length against
// the k-th neighbour is the running minimum of lcp[] between the two slots; neighbours at or after i are passed over.
template <int DIR, bool INTILE>
__device__ __forceinline__ void scan_dir0(const SaCfg& C, const Tile& T, u32 q, u32 i, Best& B) {
  u32 run = kLcpCap;
  const u32 total = DIR < 0 ? std::min(C.bucket, q) : std::min(C.bucket, C.n - 1 - q);
#pragma unroll 4
  for (u32 k = 1; k <= total; ++k) {
    const u32 qq = DIR < 0 ? q - k : q + k;
    const u64 w = T.word<INTILE>(qq);
    run = std::min<u32>(run, DIR < 0 ? (u32)(w >> 48) : (u32)(w >> 32) & 0xffffu);      // lcp[qq+1] going down, lcp[qq] going up
    const u32 p = (u32)w;
    if (p >= i) continue;
    const u32 l = std::min(run, kMaxMatch);      // never past the end of the block: lcp <= n - i
    const int score = (int)l * 8 - lg32(i - p) - 11;
    if (score > B.bscore) { B.blen = l; B.bp = p; B.blit = 0; B.bscore = score; }
    if (l < B.blen || l < C.minMatch || l > 255) break;
  }
}

// The lookahead scan (h = 1) around the slot of position i+1, for both parse states at once: the candidates, their
// lengths and the leading literal are the same, only the score term 4*(lit==0 && l1>0) and with it each state's best
// and break point differ.
template <int DIR, bool INTILE>
__device__ __forceinline__ void scan_dir1(const SaCfg& C, const Tile& T, u32 q, u32 i, u32 my_bw, Best& A, Best& B) {
  u32 run = kLcpCap;
  bool actA = true, actB = true;
  const u32 total = DIR < 0 ? std::min(C.bucket, q) : std::min(C.bucket, C.n - 1 - q);
#pragma unroll 4
  for (u32 k = 1; k <= total; ++k) {
    const u32 qq = DIR < 0 ? q - k : q + k;
    const u64 w = T.word<INTILE>(qq);
    run = std::min<u32>(run, DIR < 0 ? (u32)(w >> 48) : (u32)(w >> 32) & 0xffffu);
    const u32 s = (u32)w;
    if (s < 1 || s - 1 >= i) continue;           // p = s - 1 must exist and lie before i
    const u32 p = s - 1;
    const u32 l = std::min(run + 1u, kMaxMatch); // counted from the lookahead point; lcp <= n - (i+1) keeps it inside the block
    const u32 l1 = T.bw<INTILE>(qq) == my_bw ? 0u : 1u;   // in[p] against in[i]
    const int base = (int)(l - l1) * 8 - lg32(i - p) - 11;
    const int sB = base * 5 / 8;
    const bool stop = l < C.minMatch || l > 255;
    // inside a plateau of equal lengths almost no candidate beats the best: sA <= sB, so nothing can change unless sB does
    if (sB > A.bscore || sB > B.bscore) {
      const int sA = (base - 4 * (int)l1) * 5 / 8;
      if (actA && sA > A.bscore) { A.blen = l; A.bp = p; A.blit = l1; A.bscore = sA; }
      if (actB && sB > B.bscore) { B.blen = l; B.bp = p; B.blit = l1; B.bscore = sB; }
    }
    if (l < A.blen || stop) actA = false;
    if (l < B.blen || stop) actB = false;
    if (!actA && !actB) break;
  }
}

// pass 1, one lane per SA slot: the search without lookahead (state independent).  rec[2i] = blen:bp, rec[2i+1] = bscore;
// bw[q] = the byte before suffix sa[q] (what pass 2 compares instead of fetching in[p] per candidate)
template <bool INTILE>
__global__ __launch_bounds__(256) void lz77_sa_cand0_kernel(SaCfg C, const u32* __restrict__ sa, const u16* __restrict__ lcp, u64* __restrict__ rec,
                                                            u8* __restrict__ bw) {
  __shared__ u64 pk[kTile];
  Tile T;
  const u32 q0 = blockIdx.x * 256u;
  load_tile<false>(T, sa, lcp, nullptr, C.n, q0, pk, nullptr);
  const u32 q = q0 + threadIdx.x;
  if (q >= C.n) return;
  const u32 i = (u32)T.word<true>(q);
  bw[q] = i ? C.in[i - 1] : 0;
  Best B{C.minMatch - 1, 0u, 0u, 0};
  scan_dir0<-1, INTILE>(C, T, q, i, B);
  scan_dir0<+1, INTILE>(C, T, q, i, B);
  rec[2 * (size_t)i] = ((u64)B.blen << 32) | B.bp;
  rec[2 * (size_t)i + 1] = (u64)(u32)B.bscore;
}

// pass 2, one lane per SA slot q1 = slot of position i+1: the lookahead search for both values of (lit == 0), then the
// decision (:6414-6417).  The lane whose suffix is position 0 finishes position n-1, which has no successor.
template <bool INTILE>
__global__ __launch_bounds__(256) void lz77_sa_cand1_kernel(SaCfg C, const u32* __restrict__ sa, const u16* __restrict__ lcp, const u8* __restrict__ bw,
                                                            u64* __restrict__ rec) {
  __shared__ u64 pk[kTile];
  __shared__ u8 bw_l[kTile];
  Tile T;
  const u32 q0 = blockIdx.x * 256u;
  load_tile<true>(T, sa, lcp, bw, C.n, q0, pk, bw_l);
  const u32 q1 = q0 + threadIdx.x;
  if (q1 >= C.n) return;
  const u32 j = (u32)T.word<true>(q1);
  const u32 i = j ? j - 1 : C.n - 1;
  const u64 r0 = rec[2 * (size_t)i], r1 = rec[2 * (size_t)i + 1];
  Best B0{(u32)(r0 >> 32), (u32)r0, 0u, (int)(u32)r1};
  Best BA = B0, BB = B0;                                   // lit == 0 / lit > 0
  // isa[] of the reference holds one window of 2^checkbits positions (:6341-6348): position i+1 is visible from i only inside it
  if (j && C.lookahead >= 1 && !(B0.bscore <= 0 || B0.blen < C.minMatch) && (j >> C.checkbits) == (i >> C.checkbits)) {
    const u32 my_bw = T.bw<true>(q1);                      // in[i]
    scan_dir1<-1, INTILE>(C, T, q1, i, my_bw, BA, BB);
    scan_dir1<+1, INTILE>(C, T, q1, i, my_bw, BA, BB);
  }
  const u32 offA = i - BA.bp, offB = i - BB.bp;
  // :6414-6417 -- at level 2 a far match must be one / two bytes longer to pay for its longer offset
  const u32 needA = C.minMatch + (C.level == 2 ? (u32)(offA >= (1u << 16)) + (u32)(offA >= (1u << 24)) : 0u);
  const u32 needB = C.minMatch + (C.level == 2 ? (u32)(offB >= (1u << 16)) + (u32)(offB >= (1u << 24)) : 0u);
  const bool takeA = offA > 0 && BA.bscore > 0 && BA.blen - BA.blit >= needA;
  const bool takeB = offB > 0 && BB.bscore > 0 && BB.blen - BB.blit >= needB;
  rec[2 * (size_t)i] = make_rec(takeA, BA.blen, BA.blit, offA);
  rec[2 * (size_t)i + 1] = make_rec(takeB, BB.blen, BB.blit, offB);
}

// one bit per position: the decision with lit > 0 takes a match (lets the chain jump over literal runs)
__global__ __launch_bounds__(256) void lz77_sa_takeb_kernel(const u64* __restrict__ rec, u32 n, u64* __restrict__ takeb) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  const bool t = i < n && (rec[2 * (size_t)i + 1] >> 63);
  const u64 m = __ballot(t);
  if ((threadIdx.x & 63) == 0 && i < n) takeb[i >> 6] = m;
}

// ---- the chain -------------------------------------------------------------------------------------------------------
// Between two positions reached with lit == 0 ("nodes": the block start, the end of a match, the flush of a 4096-byte
// literal run, :6451) the parse is a function of the first one alone: take the lit==0 decision there, or else the first
// lit>0 decision that takes within 4095 further positions, or else flush.  So chains started anywhere merge for good at
// their first common node.  Every 8 KiB segment is walked from its first position by one lane (lz77_sa_spec_kernel);
// one lane per block then follows the true chain (lz77_sa_stitch_kernel): wherever it lands on a node the segment's
// lane has visited it adopts that lane's tokens up to the segment's exit, otherwise it steps itself.
struct SaBlockDev {
  const u64* rec; const u64* takeb; u32 n;
  u32 seg, nseg, seg0;                // segment bytes, segments of this block, index of the first in the per-segment arrays
  u32 tcap;                           // tokens a segment list can hold
  u32* visit;                         // bit per position: node of a speculative chain (own segment only)
  u32* stok; u32* otok;               // per segment tcap x {node, pos, len, off}: speculative / stitcher's own tokens
  u32* scnt; u32* ocnt; u32* sexit; u32* join;   // per segment
  u32* tok_pos; u32* tok_len; u32* tok_off; u32 tok_cap;   // final list
  u32* result;                        // [0] tokens [2] overflow
};

struct Step { u32 next; bool tok; u32 pos, len, off; };

// one node of the chain (see above); p < n
__device__ __forceinline__ Step next_node(const u64* __restrict__ rec, const u64* __restrict__ takeb, u32 n, u32 p) {
  Step S; S.tok = false;
  u64 r = rec[2 * (size_t)p];
  u32 m = p;
  if (!(r >> 63)) {
    // first position in (p, p+4096) whose lit>0 decision takes
    const u32 lim = (u32)std::min<u64>((u64)p + kMaxLiteral, n);     // exclusive
    m = p + 1;
    bool found = false;
    while (m < lim) {
      u64 w = takeb[m >> 6] >> (m & 63);
      if (w) { m += (u32)__builtin_ctzll(w); found = m < lim; break; }
      m = (m | 63u) + 1u;
    }
    if (!found) { S.next = lim; return S; }                          // flush (or the end of the block)
    r = rec[2 * (size_t)m + 1];
  }
  const u32 blit = (u32)(r >> 62) & 1u, blen = (u32)(r >> 32) & 0xffffu;
  S.tok = true; S.pos = m + blit; S.len = blen - blit; S.off = (u32)r; S.next = m + blen;
  return S;
}

__global__ __launch_bounds__(64) void lz77_sa_spec_kernel(const SaBlockDev* __restrict__ blocks, const u32* __restrict__ seg_block, u32 nseg_total) {
  const u32 g = blockIdx.x * 64u + threadIdx.x;
  if (g >= nseg_total) return;
  const SaBlockDev B = blocks[seg_block[g]];
  const u32 k = g - B.seg0;
  const u32 s = k * B.seg, e = (u32)std::min<u64>((u64)s + B.seg, B.n);
  u32* tk = B.stok + (size_t)k * B.tcap * 4;
  u32 p = s, nt = 0;
  while (p < e) {
    atomicOr(B.visit + (p >> 5), 1u << (p & 31));
    const Step S = next_node(B.rec, B.takeb, B.n, p);
    if (S.tok) {
      if (nt < B.tcap) { tk[4 * nt] = p; tk[4 * nt + 1] = S.pos; tk[4 * nt + 2] = S.len; tk[4 * nt + 3] = S.off; }
      ++nt;
    }
    p = S.next;
  }
  B.scnt[k] = nt; B.sexit[k] = p;
  if (nt > B.tcap) B.result[2] = 1;
}

__global__ __launch_bounds__(64) void lz77_sa_stitch_kernel(const SaBlockDev* __restrict__ blocks) {
  if (threadIdx.x) return;
  const SaBlockDev B = blocks[blockIdx.x];
  u32 p = 0;
  while (p < B.n) {
    const u32 k = p / B.seg;
    if (B.join[k] == 0xffffffffu && ((B.visit[p >> 5] >> (p & 31)) & 1u)) { B.join[k] = p; p = B.sexit[k]; continue; }
    const Step S = next_node(B.rec, B.takeb, B.n, p);
    if (S.tok) {
      const u32 c = B.ocnt[k];
      if (c < B.tcap) { u32* tk = B.otok + ((size_t)k * B.tcap + c) * 4; tk[0] = p; tk[1] = S.pos; tk[2] = S.len; tk[3] = S.off; }
      else B.result[2] = 1;
      B.ocnt[k] = c + 1;
    }
    p = S.next;
  }
}

// tokens of a segment on the true chain: the stitcher's own, then the speculative ones from the join node on
__device__ __forceinline__ u32 spec_from(const SaBlockDev& B, u32 k) {     // index of the first speculative token at or after the join node
  const u32 j = B.join[k];
  const u32 c = std::min(B.scnt[k], B.tcap);
  if (j == 0xffffffffu) return c;
  const u32* tk = B.stok + (size_t)k * B.tcap * 4;
  u32 lo = 0, hi = c;
  while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (tk[4 * mid] >= j) hi = mid; else lo = mid + 1; }
  return lo;
}

// per block: segment token counts -> offsets (one workgroup per block, segments in chunks of 1024)
__global__ __launch_bounds__(1024) void lz77_sa_count_kernel(const SaBlockDev* __restrict__ blocks, u32* __restrict__ seg_from, u32* __restrict__ seg_dst) {
  const SaBlockDev B = blocks[blockIdx.x];
  __shared__ u32 wsum[16];
  __shared__ u32 carry_s;
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (u32 k0 = 0; k0 < B.nseg; k0 += 1024) {
    const u32 k = k0 + tid;
    u32 cnt = 0, from = 0;
    if (k < B.nseg) {
      from = spec_from(B, k);
      cnt = std::min(B.ocnt[k], B.tcap) + (std::min(B.scnt[k], B.tcap) - from);
      seg_from[B.seg0 + k] = from;
    }
    u32 x = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 y = __shfl_up(x, d); if (lane >= (u32)d) x += y; }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    u32 wbase = 0;
    for (u32 w = 0; w < wave; ++w) wbase += wsum[w];
    const u32 carry = carry_s;
    if (k < B.nseg) seg_dst[B.seg0 + k] = carry + wbase + x - cnt;
    __syncthreads();
    if (tid == 1023) carry_s = carry + wbase + x;
    __syncthreads();
  }
  if (tid == 0) {
    const u32 total = carry_s;
    B.result[0] = std::min(total, B.tok_cap);
    if (total > B.tok_cap) B.result[2] = 1;
  }
}

__global__ __launch_bounds__(64) void lz77_sa_move_kernel(const SaBlockDev* __restrict__ blocks, const u32* __restrict__ seg_block,
                                                         const u32* __restrict__ seg_from, const u32* __restrict__ seg_dst) {
  const u32 g = blockIdx.x;
  const SaBlockDev B = blocks[seg_block[g]];
  const u32 k = g - B.seg0;
  const u32 no = std::min(B.ocnt[k], B.tcap), from = seg_from[g], ns = std::min(B.scnt[k], B.tcap) - from;
  const u32 dst = seg_dst[g];
  const u32* ot = B.otok + (size_t)k * B.tcap * 4;
  const u32* st = B.stok + ((size_t)k * B.tcap + from) * 4;
  for (u32 t = threadIdx.x; t < no + ns; t += 64) {
    const u32* tk = t < no ? ot + 4 * t : st + 4 * (t - no);
    const u32 o = dst + t;
    if (o < B.tok_cap) { B.tok_pos[o] = tk[1]; B.tok_len[o] = tk[2]; B.tok_off[o] = tk[3]; }
  }
}


// ---- level 2: byte-aligned codes (write_literal / write_match, ZSFX/libzpaq.cpp:6481-6489, 6518-6549) ------------------
// literals: runs of at most 64, a byte (run length - 1) in front of each; a match: pieces of minMatch .. minMatch+63 bytes,
// each a byte 64/128/192 + (piece - minMatch) followed by 2/3/4 bytes of (offset - 1), most significant first
struct Pack2Dev { const u8* in; u32 n; u32 minMatch; const u32* tok_pos; const u32* tok_len; const u32* tok_off; u32* tok_at; u32* result; u8* out; u32 out_cap; };
__device__ __forceinline__ u32 off_bytes(u32 off) { const u32 o = off - 1; return o < (1u << 16) ? 2u : o < (1u << 24) ? 3u : 4u; }
__device__ __forceinline__ u32 piece_len(u32 len, u32 mm) { return len > mm * 2 + 63 ? mm + 63 : len > mm + 63 ? len - mm : len; }
__device__ __forceinline__ u32 match_bytes2(u32 len, u32 off, u32 mm) {
  u32 np = 0;
  while (len) { len -= piece_len(len, mm); ++np; }
  return np * (1u + off_bytes(off));
}
__device__ __forceinline__ u32 gap_bytes2(u32 g) { return g + (g + 63) / 64; }

__global__ __launch_bounds__(1024) void lz77_pack2_tokens_kernel(const Pack2Dev* __restrict__ jobs) {
  const Pack2Dev J = jobs[blockIdx.x];
  const u32 ntok = J.result[0];
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ u32 wsum[16];
  __shared__ u32 carry_s, over_s;
  if (tid == 0) { carry_s = 0; over_s = 0; }
  __syncthreads();
  for (u32 t0 = 0; t0 <= ntok; t0 += 1024) {              // items 0..ntok-1 = (gap before match t, match t); item ntok = trailing literals
    const u32 t = t0 + tid;
    u32 cost = 0, gap = 0, gstart = 0, pos = 0, len = 0, off = 0;
    if (t <= ntok) {
      gstart = t ? J.tok_pos[t - 1] + J.tok_len[t - 1] : 0u;
      if (t < ntok) { pos = J.tok_pos[t]; len = J.tok_len[t]; off = J.tok_off[t]; } else pos = J.n;
      gap = pos - gstart;
      cost = gap_bytes2(gap) + (t < ntok ? match_bytes2(len, off, J.minMatch) : 0u);
    }
    u32 x = cost;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 y = __shfl_up(x, d); if (lane >= (u32)d) x += y; }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    u32 wbase = 0;
    for (u32 w = 0; w < wave; ++w) wbase += wsum[w];
    const u32 carry = carry_s;
    const u32 start = carry + wbase + x - cost;
    if (t <= ntok) {
      J.tok_at[t] = start;
      if ((u64)start + cost > J.out_cap) over_s = 1;
      else {
        u8* o = J.out + start;
        for (u32 r = 0; r < gap; r += 64) o[r + r / 64] = (u8)((gap - r < 64 ? gap - r : 64) - 1);    // run headers; the bytes come from lz77_pack2_literals_kernel
        o += gap_bytes2(gap);
        if (t < ntok) {
          const u32 ov = off - 1, nb = off_bytes(off);
          while (len) {
            const u32 l1 = piece_len(len, J.minMatch);
            *o++ = (u8)((nb - 1) * 64 + l1 - J.minMatch);
            if (nb == 4) *o++ = (u8)(ov >> 24);
            if (nb >= 3) *o++ = (u8)(ov >> 16);
            *o++ = (u8)(ov >> 8); *o++ = (u8)ov;
            len -= l1;
          }
        }
      }
    }
    __syncthreads();
    if (tid == 1023) carry_s = carry + wbase + x;
    __syncthreads();
  }
  if (tid == 0) { J.result[1] = carry_s; if (over_s || carry_s > J.out_cap) J.result[2] = 1; }
}

__global__ __launch_bounds__(256) void lz77_pack2_literals_kernel(const Pack2Dev* __restrict__ jobs) {
  const Pack2Dev J = jobs[blockIdx.y];
  const u32 x = blockIdx.x * 256u + threadIdx.x;
  const u32 ntok = J.result[0];
  if (J.result[2]) return;
  const u32 x0 = __builtin_amdgcn_readfirstlane(x);
  if (x0 >= J.n) return;
  u32 lo = 0, hi = ntok;                                   // first token with tok_pos > x0
  while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (J.tok_pos[mid] > x0) hi = mid; else lo = mid + 1; }
  if (x >= J.n) return;
  u32 t = lo;
  while (t < ntok && J.tok_pos[t] <= x) ++t;
  const u32 gstart = t ? J.tok_pos[t - 1] + J.tok_len[t - 1] : 0u;
  if (x < gstart) return;                                  // inside match t-1
  const u32 r = x - gstart;
  J.out[J.tok_at[t] + r + r / 64 + 1] = J.in[x];
}
}  // namespace

// tokens -> byte-aligned codes for nj blocks whose token lists are final (result[0] = token count, out zeroed); also what
// the hash-table parse of lz77_enc.hip calls for its level-2 jobs
int zpq_lz77_pack2_launch(zpq_ctx* ctx, const zpq_lzjob_dev* h_jobs, const u32* min_match, size_t nj, u32 max_n) {
  if (!nj) return ZPQ_OK;
  hipStream_t st = ctx->stream;
  std::vector<Pack2Dev> j2(nj);
  for (size_t i = 0; i < nj; ++i) {
    Pack2Dev& P = j2[i];
    P.in = h_jobs[i].in; P.n = h_jobs[i].n; P.minMatch = min_match[i]; P.tok_pos = h_jobs[i].tok_pos; P.tok_len = h_jobs[i].tok_len; P.tok_off = h_jobs[i].tok_off;
    P.tok_at = h_jobs[i].tok_bit; P.result = h_jobs[i].result; P.out = h_jobs[i].out; P.out_cap = h_jobs[i].out_cap;
  }
  Pack2Dev* d_p2 = (Pack2Dev*)zpq_scratch(ctx, 30, nj * sizeof(Pack2Dev) + 256);
  if (!d_p2) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "level-2 pack records");
  ZPQ_HIP(ctx, hipMemcpyAsync(d_p2, j2.data(), nj * sizeof(Pack2Dev), hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  ZPQ_LAUNCH(ctx, "lz77_pack2_tokens_kernel", st, lz77_pack2_tokens_kernel, dim3((unsigned)nj), dim3(1024), d_p2);
  if (max_n) ZPQ_LAUNCH(ctx, "lz77_pack2_literals_kernel", st, lz77_pack2_literals_kernel, dim3((max_n + 255) / 256, (unsigned)nj), dim3(256), d_p2);
  ZPQ_HIP(ctx, hipGetLastError());
  return ZPQ_OK;
}


// ---- host side -------------------------------------------------------------------------------------------------------

// divsufsort's result on the device (ZSFX/libzpaq.cpp:6047): d_sa[n] and, when asked for, its inverse d_isa[n]
extern "C" int zpq_suffix_array_dev(zpq_ctx* ctx, const void* d_in, size_t n, uint32_t* d_sa, uint32_t* d_isa) {
  if (!ctx) return ZPQ_ERR_ARG;
  (void)hipSetDevice(ctx->device);
  if (n == 0) return ZPQ_OK;
  if (n >= (1ull << 31)) return zpq_fail(ctx, ZPQ_ERR_ARG, "suffix array of %zu bytes: blocks are below 2 GiB", n);
  if (!d_in || !d_sa) return zpq_fail(ctx, ZPQ_ERR_ARG, "null buffer");
  const size_t wb = sa_work_bytes((u32)n) + (d_isa ? 0 : (n * 4 + 256));
  u8* work = (u8*)zpq_scratch(ctx, 0, wb);
  if (!work) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "suffix array scratch (%zu MiB)", wb >> 20);
  u32* rank = d_isa;
  if (!rank) { rank = (u32*)work; work += (n * 4 + 255) & ~(size_t)255; }
  int rc = build_suffix_array(ctx, ctx->stream, (const u8*)d_in, (u32)n, d_sa, rank, work, nullptr);
  if (rc) return rc;
  ZPQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZPQ_OK;
}

// The Burrows-Wheeler transform as LZBuffer emits it for (args[1] & 3) == 3 (ZSFX/libzpaq.cpp:6317-6326): n+5 bytes.
extern "C" int zpq_bwt_dev(zpq_ctx* ctx, const void* d_in, size_t n, uint8_t* d_out) {
  if (!ctx) return ZPQ_ERR_ARG;
  (void)hipSetDevice(ctx->device);
  if (n >= (1ull << 31)) return zpq_fail(ctx, ZPQ_ERR_ARG, "BWT of %zu bytes: blocks are below 2 GiB", n);
  if (!d_out || (n && !d_in)) return zpq_fail(ctx, ZPQ_ERR_ARG, "null buffer");
  hipStream_t st = ctx->stream;
  u32* d_sa = nullptr;
  if (n) {
    const size_t wb = sa_work_bytes((u32)n) + (n * 4 + 256) * 2;
    u8* work = (u8*)zpq_scratch(ctx, 0, wb);
    if (!work) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "suffix array scratch (%zu MiB)", wb >> 20);
    d_sa = (u32*)work; work += (n * 4 + 255) & ~(size_t)255;
    u32* rank = (u32*)work; work += (n * 4 + 255) & ~(size_t)255;
    int rc = build_suffix_array(ctx, st, (const u8*)d_in, (u32)n, d_sa, rank, work, nullptr);
    if (rc) return rc;
  } else {
    const u8 tail[5] = {255, 0, 0, 0, 0};
    ZPQ_HIP(ctx, hipMemcpyAsync(d_out, tail, 5, hipMemcpyHostToDevice, st));
    ZPQ_HIP(ctx, hipStreamSynchronize(st));
    return ZPQ_OK;
  }
  ZPQ_LAUNCH(ctx, "bwt_output_kernel", st, bwt_output_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), (const u8*)d_in, (u32)n, d_sa, d_out);
  ZPQ_HIP(ctx, hipGetLastError());
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  return ZPQ_OK;
}

namespace {
struct SaSizes { size_t rec, takeb, visit, tok, segtok, nseg; u32 tok_cap, tcap; };
SaSizes sa_sizes(u32 n, u32 mm, u32 seg) {
  SaSizes z;
  const size_t nn = ((size_t)n + 255) & ~(size_t)255;
  z.nseg = std::max<size_t>(1, ((size_t)n + seg - 1) / seg);
  z.tok_cap = n / mm + 3;
  z.tcap = seg / mm + 2;
  z.rec = nn * 16; z.takeb = (nn / 64 + 8) * 8; z.visit = (nn / 32 + 8) * 4;
  z.tok = ((size_t)z.tok_cap * 16 + 255) & ~(size_t)255;
  z.segtok = ((size_t)z.nseg * z.tcap * 16 + 255) & ~(size_t)255;
  return z;
}
size_t sa_job_bytes(u32 n, u32 mm, u32 seg) {
  const SaSizes z = sa_sizes(n, mm, seg);
  return z.rec + z.takeb + z.visit + z.tok + 2 * z.segtok + z.nseg * 32 + 4096;
}
}  // namespace

// LZ77 jobs whose match finder is the suffix array (args[5]-args[0] >= 21).  Suffix array, LCP and the two candidate
// passes run block after block (each fills the chip and they share one work area); the chains of all blocks of a batch
// are then walked, stitched and packed together.
int zpq_lz77_sa_encode(zpq_ctx* ctx, zpq_lz77_job* jobs, const size_t* which, size_t nj_all) {
  hipStream_t st = ctx->stream;
  u32 seg = 8192;
  if (const char* e = getenv("ZPQ_SA_SEG")) seg = std::max<u32>(16, (u32)strtoul(e, 0, 10));   // tests
  u32 max_n_all = 0;
  for (size_t w = 0; w < nj_all; ++w) max_n_all = std::max(max_n_all, jobs[which[w]].n);
  const size_t nn_max = ((size_t)max_n_all + 255) & ~(size_t)255;
  const size_t shared_bytes = nn_max * 4 * 2 + (nn_max + 256) * 2 + nn_max + 2048 + sa_work_bytes(max_n_all);
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  // what is free now plus what this path already holds, less a reserve for everything else that allocates later
  const size_t have_b = free_b + ctx->scratch_cap[0] + ctx->scratch_cap[24], reserve_b = std::max<size_t>((size_t)8 << 30, total_b / 10);
  size_t budget = std::max<size_t>((size_t)2 << 30, have_b > reserve_b ? have_b - reserve_b : 0);
  if (const char* e = getenv("ZPQ_LZ_BUDGET_MB")) budget = (size_t)strtoull(e, 0, 10) << 20;
  budget = budget > shared_bytes ? budget - shared_bytes : 0;

  size_t lo = 0;
  while (lo < nj_all) {
    size_t hi = lo, bytes = 0, nseg_total = 0;
    while (hi < nj_all) {
      const zpq_lz77_job& z = jobs[which[hi]];
      const size_t b = sa_job_bytes(z.n, (u32)z.args[2], seg);
      if (hi > lo && bytes + b > budget) break;
      bytes += b; nseg_total += sa_sizes(z.n, (u32)z.args[2], seg).nseg; ++hi;
    }
    const size_t nj = hi - lo;
    u8* shared = (u8*)zpq_scratch(ctx, 0, shared_bytes);
    u8* per = (u8*)zpq_scratch(ctx, 24, bytes + 4096);
    const size_t meta_bytes = nj * (sizeof(zpq_lzjob_dev) + sizeof(SaBlockDev) + sizeof(Pack2Dev) + 16 + 768) + nseg_total * 16 + 4096;
    u8* d_meta = (u8*)zpq_scratch(ctx, 2, meta_bytes);
    if (!shared || !per || !d_meta) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "lz77 suffix-array scratch (%zu MiB)", (shared_bytes + bytes) >> 20);
    u8* sp = shared;
    u32* d_sa = carve<u32>(sp, nn_max);
    u32* d_isa = carve<u32>(sp, nn_max);
    u16* d_lcp = carve<u16>(sp, nn_max + 256);
    u8* d_bw = carve<u8>(sp, nn_max);
    u8* sort_work = sp;
    u8* mp = d_meta;
    zpq_lzjob_dev* d_jobs = carve<zpq_lzjob_dev>(mp, nj);
    SaBlockDev* d_blocks = carve<SaBlockDev>(mp, nj);
    u32* d_res = carve<u32>(mp, nj * 4);
    u32* d_segblock = carve<u32>(mp, nseg_total);
    u32* d_segfrom = carve<u32>(mp, nseg_total);
    u32* d_segdst = carve<u32>(mp, nseg_total);
    u8* d_pack2 = (u8*)carve<Pack2Dev>(mp, nj);
    std::vector<zpq_lzjob_dev> hj(nj);
    std::vector<SaBlockDev> hb(nj);
    std::vector<u32> segblock(nseg_total);
    ZPQ_HIP(ctx, hipMemsetAsync(d_res, 0, nj * 16, st));
    u8* pp = per;
    u32 seg0 = 0, max_n = 0;
    for (size_t i = 0; i < nj; ++i) {
      zpq_lz77_job& z = jobs[which[lo + i]];
      const SaSizes S = sa_sizes(z.n, (u32)z.args[2], seg);
      zpq_lzjob_dev& J = hj[i];
      memset(&J, 0, sizeof J);
      SaBlockDev& B = hb[i];
      memset(&B, 0, sizeof B);
      u64* rec = (u64*)pp; pp += S.rec;
      u64* takeb = (u64*)pp; pp += S.takeb;
      u32* visit = (u32*)pp; pp += S.visit;
      u32* tok = (u32*)pp; pp += S.tok;
      u32* stok = (u32*)pp; pp += S.segtok;
      u32* otok = (u32*)pp; pp += S.segtok;
      u32* segw = (u32*)pp; pp += (S.nseg * 16 + 255) & ~(size_t)255;
      J.in = (const u8*)z.d_in; J.n = z.n; J.rb = z.args[0] > 4 ? (u32)z.args[0] - 4 : 0; J.nseg = 1; J.seg0 = 0;
      J.tok_pos = tok; J.tok_len = tok + S.tok_cap; J.tok_off = J.tok_len + S.tok_cap; J.tok_bit = J.tok_off + S.tok_cap;
      J.tok_cap = S.tok_cap - 1; J.result = d_res + 4 * i; J.out = (u8*)z.d_out; J.out_cap = z.out_cap; J.plan = nullptr;
      B.rec = rec; B.takeb = takeb; B.n = z.n; B.seg = seg; B.nseg = (u32)S.nseg; B.seg0 = seg0; B.tcap = S.tcap;
      B.visit = visit; B.stok = stok; B.otok = otok;
      B.scnt = segw; B.ocnt = segw + S.nseg; B.sexit = segw + 2 * S.nseg; B.join = segw + 3 * S.nseg;
      B.tok_pos = J.tok_pos; B.tok_len = J.tok_len; B.tok_off = J.tok_off; B.tok_cap = J.tok_cap; B.result = J.result;
      for (size_t k = 0; k < S.nseg; ++k) segblock[seg0 + k] = (u32)i;
      seg0 += (u32)S.nseg;
      max_n = std::max(max_n, z.n);
      ZPQ_HIP(ctx, hipMemsetAsync(J.out, 0, J.out_cap, st));
      ZPQ_HIP(ctx, hipMemsetAsync(visit, 0, S.visit, st));
      ZPQ_HIP(ctx, hipMemsetAsync(segw, 0, S.nseg * 12, st));                  // scnt, ocnt, sexit
      ZPQ_HIP(ctx, hipMemsetAsync(segw + 3 * S.nseg, 0xff, S.nseg * 4, st));   // join = none
      if (z.n) {
        int rc = build_suffix_array(ctx, st, J.in, z.n, d_sa, d_isa, sort_work, nullptr);
        if (rc) return rc;
        const u32 n = z.n;
        ZPQ_LAUNCH(ctx, "sa_lcp_kernel", st, sa_lcp_kernel, dim3(((n + kLcpChunk - 1) / kLcpChunk + 255) / 256), dim3(256), J.in, n, d_sa, d_isa, d_lcp);
        SaCfg C;
        C.in = J.in; C.n = n; C.minMatch = (u32)z.args[2]; C.bucket = (1u << z.args[4]) - 1; C.lookahead = (u32)z.args[6]; C.checkbits = 17 + (u32)z.args[0]; C.level = (u32)z.args[1] & 3;
        if (C.bucket < kTileR) {         // every neighbour a scan reaches is inside the workgroup's tile
          ZPQ_LAUNCH(ctx, "lz77_sa_cand0_kernel", st, lz77_sa_cand0_kernel<true>, dim3((n + 255) / 256), dim3(256), C, d_sa, d_lcp, rec, d_bw);
          ZPQ_LAUNCH(ctx, "lz77_sa_cand1_kernel", st, lz77_sa_cand1_kernel<true>, dim3((n + 255) / 256), dim3(256), C, d_sa, d_lcp, d_bw, rec);
        } else {
          ZPQ_LAUNCH(ctx, "lz77_sa_cand0_kernel", st, lz77_sa_cand0_kernel<false>, dim3((n + 255) / 256), dim3(256), C, d_sa, d_lcp, rec, d_bw);
          ZPQ_LAUNCH(ctx, "lz77_sa_cand1_kernel", st, lz77_sa_cand1_kernel<false>, dim3((n + 255) / 256), dim3(256), C, d_sa, d_lcp, d_bw, rec);
        }
        ZPQ_LAUNCH(ctx, "lz77_sa_takeb_kernel", st, lz77_sa_takeb_kernel, dim3((n + 255) / 256), dim3(256), rec, n, takeb);
        ZPQ_HIP(ctx, hipGetLastError());
      }
    }
    ZPQ_HIP(ctx, hipMemcpyAsync(d_jobs, hj.data(), nj * sizeof(zpq_lzjob_dev), hipMemcpyHostToDevice, st));
    ZPQ_HIP(ctx, hipMemcpyAsync(d_blocks, hb.data(), nj * sizeof(SaBlockDev), hipMemcpyHostToDevice, st));
    ZPQ_HIP(ctx, hipMemcpyAsync(d_segblock, segblock.data(), nseg_total * 4, hipMemcpyHostToDevice, st));
    ZPQ_LAUNCH(ctx, "lz77_sa_spec_kernel", st, lz77_sa_spec_kernel, dim3((unsigned)((nseg_total + 63) / 64)), dim3(64), d_blocks, d_segblock, (u32)nseg_total);
    ZPQ_LAUNCH(ctx, "lz77_sa_stitch_kernel", st, lz77_sa_stitch_kernel, dim3((unsigned)nj), dim3(64), d_blocks);
    ZPQ_LAUNCH(ctx, "lz77_sa_count_kernel", st, lz77_sa_count_kernel, dim3((unsigned)nj), dim3(1024), d_blocks, d_segfrom, d_segdst);
    ZPQ_LAUNCH(ctx, "lz77_sa_move_kernel", st, lz77_sa_move_kernel, dim3((unsigned)nseg_total), dim3(64), d_blocks, d_segblock, d_segfrom, d_segdst);
    ZPQ_HIP(ctx, hipGetLastError());
    {   // tokens -> codes: level 1 (bits) for the jobs that ask for it, level 2 (bytes) for the others
      std::vector<zpq_lzjob_dev> j1; std::vector<Pack2Dev> j2;
      u32 max1 = 0, max2 = 0;
      for (size_t i = 0; i < nj; ++i) {
        const zpq_lz77_job& z = jobs[which[lo + i]];
        if ((z.args[1] & 3) == 2) {
          Pack2Dev P; P.in = hj[i].in; P.n = hj[i].n; P.minMatch = (u32)z.args[2]; P.tok_pos = hj[i].tok_pos; P.tok_len = hj[i].tok_len; P.tok_off = hj[i].tok_off;
          P.tok_at = hj[i].tok_bit; P.result = hj[i].result; P.out = hj[i].out; P.out_cap = hj[i].out_cap;
          j2.push_back(P); max2 = std::max(max2, z.n);
        } else { j1.push_back(hj[i]); max1 = std::max(max1, z.n); }
      }
      // (the records go behind the ones uploaded above: d_jobs has room for nj, the level-2 ones use the segment map's tail)
      if (!j1.empty()) {
        ZPQ_HIP(ctx, hipMemcpyAsync(d_jobs, j1.data(), j1.size() * sizeof(zpq_lzjob_dev), hipMemcpyHostToDevice, st));
        ZPQ_HIP(ctx, hipStreamSynchronize(st));
        int rc = zpq_lz77_pack_launch(ctx, d_jobs, j1.size(), max1);
        if (rc) return rc;
      }
      if (!j2.empty()) {
        Pack2Dev* d_p2 = (Pack2Dev*)d_pack2;
        ZPQ_HIP(ctx, hipMemcpyAsync(d_p2, j2.data(), j2.size() * sizeof(Pack2Dev), hipMemcpyHostToDevice, st));
        ZPQ_HIP(ctx, hipStreamSynchronize(st));
        ZPQ_LAUNCH(ctx, "lz77_pack2_tokens_kernel", st, lz77_pack2_tokens_kernel, dim3((unsigned)j2.size()), dim3(1024), d_p2);
        if (max2) ZPQ_LAUNCH(ctx, "lz77_pack2_literals_kernel", st, lz77_pack2_literals_kernel, dim3((max2 + 255) / 256, (unsigned)j2.size()), dim3(256), d_p2);
        ZPQ_HIP(ctx, hipGetLastError());
      }
    }
    std::vector<u32> res(nj * 4);
    ZPQ_HIP(ctx, hipMemcpyAsync(res.data(), d_res, nj * 16, hipMemcpyDeviceToHost, st));
    ZPQ_HIP(ctx, hipStreamSynchronize(st));     // (also keeps hj/hb/segblock alive until their uploads are done)
    for (size_t i = 0; i < nj; ++i) {
      zpq_lz77_job& z = jobs[which[lo + i]];
      z.n_matches = res[4 * i];
      z.out_len = res[4 * i + 1];
      if (res[4 * i + 2]) return zpq_fail(ctx, ZPQ_ERR_CAPACITY, "job %zu: token or output capacity exceeded", which[lo + i]);
    }
    lo = hi;
  }
  return ZPQ_OK;
}
