// Inverse Burrows-Wheeler transform (decode side of methods 3 / "x,3": what the BWT post-processor program of
// makeConfig does on the reference's ZPAQL machine, PostProcessor::write ZSFX/libzpaq.cpp:2185-2226; the stream is what
// LZBuffer emits at level 3, :6317-6326): B[0] = last input byte, B[i] = the byte before suffix sa[i-1] (255 where that
// suffix is the whole block: row idx), B[n+1..n+4] = idx, LSB first.
//
// The program inverts through a linked list (one dependent random read per output byte: 1.5 us each on HBM, 25 s
// for a 16 MiB block on one lane).  Here:
//   1. the list IS a stable sort of the row numbers by their byte (one 8-bit counting pass: per-chunk histograms,
//      a scan, a stable scatter -- ranks among equal bytes of a 64-row step come from 8 ballots);
//   2. following it is list ranking: every 2^k-th row (and row idx) is a splitter; a lane per splitter walks to the next
//      splitter (lengths), one lane ranks the <= 8192 splitters in LDS (start offsets), and the lanes walk again writing
//      their stretch of the output.  A block of n bytes costs ~2 n / 8192 dependent reads of latency instead of n.
#include "zpq_internal.h"

namespace {

constexpr u32 kChunk = 4096;       // rows per histogram / scatter chunk
constexpr u32 kMaxSplit = 8192;    // splitters ranked in LDS

__global__ __launch_bounds__(256) void ibwt_hist_kernel(const u8* __restrict__ L, u32 rows, u32 idx, u32* __restrict__ hist) {
  __shared__ u32 h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const u32 base = blockIdx.x * kChunk;
  for (u32 k = threadIdx.x; k < kChunk; k += 256) {
    const u32 r = base + k;
    if (r < rows && r != idx) atomicAdd(&h[L[r]], 1u);
  }
  __syncthreads();
  hist[(size_t)blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
}

// hist[chunk][c] -> first list position of byte c in that chunk (position 0 is the end-of-string row)
__global__ __launch_bounds__(256) void ibwt_scan_kernel(u32* __restrict__ hist, u32 nchunks) {
  __shared__ u32 tot[256];
  const u32 c = threadIdx.x;
  u32 run = 0;
  for (u32 k = 0; k < nchunks; ++k) { const u32 v = hist[(size_t)k * 256 + c]; hist[(size_t)k * 256 + c] = run; run += v; }
  tot[c] = run;
  __syncthreads();
  u32 before = 1;
  for (u32 q = 0; q < c; ++q) before += tot[q];
  for (u32 k = 0; k < nchunks; ++k) hist[(size_t)k * 256 + c] += before;
}

__global__ __launch_bounds__(64) void ibwt_scatter_kernel(const u8* __restrict__ L, u32 rows, u32 idx, const u32* __restrict__ hist,
                                                          u32* __restrict__ T) {
  __shared__ u32 cnt[256];
  const u32 lane = (u32)lane_id();
  for (u32 q = lane; q < 256; q += 64) cnt[q] = hist[(size_t)blockIdx.x * 256 + q];
  __builtin_amdgcn_wave_barrier();
  const u32 base = blockIdx.x * kChunk;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (u32 s = 0; s < kChunk; s += 64) {
    const u32 r = base + s + lane;
    const bool valid = r < rows && r != idx;
    const u32 c = valid ? L[r] : 0u;
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned long long b = __ballot(valid && ((c >> k) & 1u));
      m &= ((c >> k) & 1u) ? b : ~b;
    }
    const u32 rank = (u32)__popcll(m & lt);
    if (valid) {
      T[cnt[c] + rank] = r;
      if ((m >> lane) == 1ull) cnt[c] += (u32)__popcll(m);        // the highest lane of its byte moves the counter on
    }
    __builtin_amdgcn_wave_barrier();
  }
}

struct WalkP { u32 rows, idx, shift, nsplit; };       // splitter s < nsplit-1 is row s << shift; the last one is row idx
__device__ __forceinline__ bool is_split(u32 r, const WalkP& P) { return (r & ((1u << P.shift) - 1u)) == 0 || r == P.idx; }
__device__ __forceinline__ u32 split_of(u32 r, const WalkP& P) { return r == P.idx ? P.nsplit - 1 : r >> P.shift; }

// lengths: from every splitter to the next one along the list
__global__ __launch_bounds__(64) void ibwt_len_kernel(const u32* __restrict__ T, WalkP P, u32* __restrict__ nxt, u32* __restrict__ len) {
  const u32 s = blockIdx.x * 64u + threadIdx.x;
  if (s >= P.nsplit) return;
  const u32 row = s == P.nsplit - 1 ? P.idx : s << P.shift;
  if (row == 0 || row >= P.rows || (s != P.nsplit - 1 && row == P.idx)) { nxt[s] = 0xffffffffu; len[s] = 0; return; }   // row 0 ends the list
  u32 r = row, k = 0;
  do { r = T[r]; ++k; } while (r < P.rows && !is_split(r, P) && k <= P.rows);
  nxt[s] = (r < P.rows && is_split(r, P)) ? split_of(r, P) : 0xfffffffeu;
  len[s] = k;
}

// start offsets: one lane follows the splitters from row idx to row 0
__global__ __launch_bounds__(64) void ibwt_rank_kernel(const u32* __restrict__ nxt, const u32* __restrict__ len, WalkP P, u32* __restrict__ off,
                                                       u32* __restrict__ result) {
  __shared__ u32 sn[kMaxSplit + 1], sl[kMaxSplit + 1];
  for (u32 q = threadIdx.x; q < P.nsplit; q += 64) { sn[q] = nxt[q]; sl[q] = len[q]; off[q] = 0xffffffffu; }
  __syncthreads();
  if (threadIdx.x) return;
  u32 s = P.nsplit - 1, pos = 0, steps = 0;
  bool ok = true;
  while (s != 0) {
    if (s >= P.nsplit || ++steps > P.nsplit) { ok = false; break; }
    off[s] = pos;
    pos += sl[s];
    s = sn[s];
  }
  result[0] = pos;                                   // bytes the walk produces: rows - 1 when the stream is sound
  result[1] = (ok && pos + 1 == P.rows) ? 0u : (u32)ZPQ_ERR_FORMAT;
}

__global__ __launch_bounds__(64) void ibwt_emit_kernel(const u8* __restrict__ L, const u32* __restrict__ T, WalkP P, const u32* __restrict__ off,
                                                       const u32* __restrict__ len, const u32* __restrict__ result, u8* __restrict__ out, u32 out_cap) {
  const u32 s = blockIdx.x * 64u + threadIdx.x;
  if (s >= P.nsplit || result[1] != 0) return;
  const u32 o = off[s];
  if (o == 0xffffffffu) return;
  u32 r = s == P.nsplit - 1 ? P.idx : s << P.shift;
  const u32 n = len[s];
  for (u32 j = 0; j < n; ++j) {
    r = T[r];
    if (o + j < out_cap) out[o + j] = L[r];
  }
}

}  // namespace

// d_bwt[0..m): the level-3 stream (m = n + 5).  Writes the n original bytes to d_out; *out_len = n.  Synchronous.
int zpq_ibwt_dev(zpq_ctx* ctx, const u8* d_bwt, u32 m, u8* d_out, u32 out_cap, u32* out_len) {
  hipStream_t st = ctx->stream;
  *out_len = 0;
  if (m < 5) return zpq_fail(ctx, ZPQ_ERR_FORMAT, "BWT stream shorter than its index");
  u8 tail[4];
  ZPQ_HIP(ctx, hipMemcpyAsync(tail, d_bwt + m - 4, 4, hipMemcpyDeviceToHost, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  const u32 idx = tail[0] | (u32)tail[1] << 8 | (u32)tail[2] << 16 | (u32)tail[3] << 24;
  const u32 rows = m - 4;                             // n + 1
  const u32 n = rows - 1;
  if (n == 0) return idx == 0 ? ZPQ_OK : zpq_fail(ctx, ZPQ_ERR_FORMAT, "BWT index out of range");
  if (idx == 0 || idx >= rows) return zpq_fail(ctx, ZPQ_ERR_FORMAT, "BWT index out of range");
  if (n > out_cap) return zpq_fail(ctx, ZPQ_ERR_CAPACITY, "inverse BWT needs %u bytes", n);
  const u32 nchunks = (rows + kChunk - 1) / kChunk;
  u32 shift = 0;
  while (((rows + (1u << shift) - 1) >> shift) > kMaxSplit - 1) ++shift;
  WalkP P; P.rows = rows; P.idx = idx; P.shift = shift; P.nsplit = ((rows + (1u << shift) - 1) >> shift) + 1;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t bytes = al((size_t)rows * 4) + al((size_t)nchunks * 1024) + al((size_t)P.nsplit * 4) * 3 + al(8);
  u8* w = (u8*)zpq_scratch(ctx, 27, bytes + 256);
  if (!w) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "inverse BWT scratch (%zu MiB)", bytes >> 20);
  u32* T = (u32*)w; w += al((size_t)rows * 4);
  u32* hist = (u32*)w; w += al((size_t)nchunks * 1024);
  u32* nxt = (u32*)w; w += al((size_t)P.nsplit * 4);
  u32* len = (u32*)w; w += al((size_t)P.nsplit * 4);
  u32* off = (u32*)w; w += al((size_t)P.nsplit * 4);
  u32* res = (u32*)w;
  ZPQ_LAUNCH(ctx, "ibwt_hist_kernel", st, ibwt_hist_kernel, dim3(nchunks), dim3(256), d_bwt, rows, idx, hist);
  ZPQ_LAUNCH(ctx, "ibwt_scan_kernel", st, ibwt_scan_kernel, dim3(1), dim3(256), hist, nchunks);
  ZPQ_LAUNCH(ctx, "ibwt_scatter_kernel", st, ibwt_scatter_kernel, dim3(nchunks), dim3(64), d_bwt, rows, idx, hist, T);
  const u32 idx_word = idx;
  ZPQ_HIP(ctx, hipMemcpyAsync(T, &idx_word, 4, hipMemcpyHostToDevice, st));      // list position 0 is the end-of-string row
  ZPQ_LAUNCH(ctx, "ibwt_len_kernel", st, ibwt_len_kernel, dim3((P.nsplit + 63) / 64), dim3(64), T, P, nxt, len);
  ZPQ_LAUNCH(ctx, "ibwt_rank_kernel", st, ibwt_rank_kernel, dim3(1), dim3(64), nxt, len, P, off, res);
  ZPQ_LAUNCH(ctx, "ibwt_emit_kernel", st, ibwt_emit_kernel, dim3((P.nsplit + 63) / 64), dim3(64), d_bwt, T, P, off, len, res, d_out, out_cap);
  ZPQ_HIP(ctx, hipGetLastError());
  u32 r[2];
  ZPQ_HIP(ctx, hipMemcpyAsync(r, res, 8, hipMemcpyDeviceToHost, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  if (r[1]) return zpq_fail(ctx, ZPQ_ERR_FORMAT, "BWT stream is not a permutation cycle (%u of %u bytes reachable)", r[0], n);
  *out_len = n;
  return ZPQ_OK;
}
