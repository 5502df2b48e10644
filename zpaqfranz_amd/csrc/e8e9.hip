// E8E9 pre-processor: libzpaq's e8e9() (ZSFX/libzpaq.cpp:6117-6126), the x86
// CALL/JMP filter compressBlock applies when the exe bit of the type hint is set -- relative 24-bit targets
// after an E8/E9 opcode whose fourth operand byte is 00 or FF are made absolute (+ offset).
//
//   for (i = n-5; i >= 0; --i)
//     if ((buf[i] & 254) == 0xe8 && ((buf[i+4] + 1) & 254) == 0) { a = buf[i+1..i+3] + i; store a; }
//
// The loop runs downwards and a rewrite at i touches bytes i+1..i+3 only, so:
//   * an opcode byte is never modified before it is tested (writes land above the running index),
//   * the test byte buf[i+4] and the operand can only have been modified by candidates at i+1..i+3.
// Opcode positions closer than 4 bytes therefore form chains that must be replayed in order, and chains
// are independent of each other: the highest candidate q of the chain below p reads at most q+4 <= p,
// an opcode byte, which nobody writes.  One thread per position finds the chain heads (a candidate
// without another candidate in the 3 bytes above it); the head's thread replays its chain downwards in
// place.  Chains are a few bytes long in real code; a run of E8 bytes degenerates to one serial walk,
// still exact.
#include <stdlib.h>

#include "zpq_internal.h"

namespace {

// Candidacy is a property of the ORIGINAL bytes (an operand byte may turn into E8 once it is rewritten, and an
// E8 that is another call's operand is tested before that call rewrites it): one bit per position, taken
// before anything is written.
__global__ __launch_bounds__(256) void e8e9_mark_kernel(const u8* __restrict__ buf, u64 n, u32* __restrict__ bits) {
  const u64 w = (u64)blockIdx.x * 256 + threadIdx.x;          // word w covers positions 32w .. 32w+31
  if (w * 32 >= n) return;
  u32 m = 0;
  for (u32 j = 0; j < 32; ++j) {
    const u64 k = w * 32 + j;
    if (k + 5 <= n && (buf[k] & 254) == 0xe8) m |= 1u << j;
  }
  bits[w] = m;
}

__global__ __launch_bounds__(256) void e8e9_forward_kernel(u8* __restrict__ buf, u64 n, const u32* __restrict__ bits) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i + 5 > n) return;
  auto cand = [&](u64 k) { return k + 5 <= n && ((bits[k >> 5] >> (k & 31)) & 1u) != 0; };
  if (!cand(i) || cand(i + 1) || cand(i + 2) || cand(i + 3)) return;            // not a chain head
  u64 k = i;
  for (;;) {
    if (((buf[k + 4] + 1) & 254) == 0) {
      const u32 a = ((u32)buf[k + 1] | (u32)buf[k + 2] << 8 | (u32)buf[k + 3] << 16) + (u32)k;
      buf[k + 1] = (u8)a; buf[k + 2] = (u8)(a >> 8); buf[k + 3] = (u8)(a >> 16);
    }
    // next candidate of this chain: the nearest one within 3 bytes below
    if (k >= 1 && cand(k - 1)) k -= 1;
    else if (k >= 2 && cand(k - 2)) k -= 2;
    else if (k >= 3 && cand(k - 3)) k -= 3;
    else break;
  }
}

// ---- inverse (what the E8E9 variants of the level-1 post-processor do at the end of a segment, config.hip
// lazy2_source(rb, true): "for b = 0..d-1: if b+4 < d and (*b & 254) == 232 and (b[4] + 1) & 254 == 0: operand -= b") ----
// The loop runs UPWARDS and is the exact mirror of e8e9(): whether position i is an opcode is decided on its restored
// byte, its test byte and operand are read as the opcodes below left them.  All of that state is the current value of
// the three bytes ahead of the running index, so the walk can start anywhere once those three bytes are known: every
// lane walks one segment after a short warm-up that assumes the bytes are still as stored (true unless a dependency
// chain of rewrites is longer than the warm-up, which real code does not do and dense E8 runs do); it records the
// state it assumed at its segment start and the state it leaves.  One lane per block then compares neighbouring
// states in order and re-walks, from the true state, the segments whose assumption was wrong.  Exact in every case.
typedef zpq_e8inv_job E8InvJob;
constexpr u32 kE8Seg = 1024;

// walks [from, to) with `cur` = current values of bytes from..from+7; writes out[i] for i >= wr_from; returns the state at `to`
__device__ __forceinline__ u32 e8_walk(const u8* __restrict__ T, u8* __restrict__ out, u32 n, u32 from, u32 to, u32 wr_from, u64 cur,
                                       u32 seg_start, u32* in_state) {
  u32 i = from;
  while (i < to) {
    // the next 8 stored bytes (positions i+8 .. i+15), zero past the end
    u64 nxt = 0;
    if (i + 8 < n) {
      nxt = *(const u64_u*)(T + i + 8);
      const u32 valid = n - (i + 8);
      if (valid < 8) nxt &= (1ull << (8 * valid)) - 1;
    }
    const u32 stop = to - i < 8 ? to - i : 8;
    for (u32 k = 0; k < stop; ++k, ++i) {
      if (i == seg_start && in_state) *in_state = (u32)cur & 0xffffffu;
      const u32 b0 = (u32)cur & 255u;
      if (i + 4 < n && (b0 & 254u) == 0xe8u && ((((u32)(cur >> 32) & 255u) + 1u) & 254u) == 0) {
        const u32 a = (((u32)(cur >> 8)) & 0xffffffu) - i;
        cur = (cur & ~0xffffff00ull) | ((u64)(a & 0xffffffu) << 8);
      }
      if (i >= wr_from) out[i] = (u8)b0;
      cur = (cur >> 8) | ((nxt & 255ull) << 56);
      nxt >>= 8;
    }
  }
  return (u32)cur & 0xffffffu;
}

__device__ __forceinline__ u64 e8_window(const u8* __restrict__ T, u32 n, u32 at) {
  u64 w = 0;
  if (at < n) {
    w = *(const u64_u*)(T + at);
    const u32 valid = n - at;
    if (valid < 8) w &= (1ull << (8 * valid)) - 1;
  }
  return w;
}

__global__ __launch_bounds__(256) void e8e9_inverse_walk_kernel(const E8InvJob* __restrict__ jobs, u32 warmup, u32* __restrict__ st_in,
                                                                u32* __restrict__ st_out) {
  const E8InvJob J = jobs[blockIdx.y];
  const u32 n = *J.len;
  const u32 seg = blockIdx.x * 256u + threadIdx.x;
  const u64 s64 = (u64)seg * kE8Seg;
  if (s64 >= n) return;
  const u32 s = (u32)s64, e = n - s < kE8Seg ? n : s + kE8Seg;
  const u32 from = s >= warmup ? s - warmup : 0;
  u32 in_state = 0;
  const u32 o = e8_walk(J.in, J.out, n, from, e, s, e8_window(J.in, n, from), s, &in_state);
  st_in[J.st_base + seg] = in_state;
  st_out[J.st_base + seg] = o;
}

__global__ __launch_bounds__(64) void e8e9_inverse_fix_kernel(const E8InvJob* __restrict__ jobs, u32 njobs, const u32* __restrict__ st_in,
                                                              u32* __restrict__ st_out, u32* __restrict__ refixed) {
  const u32 j = blockIdx.x * 64u + threadIdx.x;
  if (j >= njobs) return;
  const E8InvJob J = jobs[j];
  const u32 n = *J.len;
  const u32 nseg = (u32)(((u64)n + kE8Seg - 1) / kE8Seg);
  for (u32 k = 1; k < nseg; ++k) {
    const u32 truth = st_out[J.st_base + k - 1];
    if (st_in[J.st_base + k] == truth) continue;
    const u32 s = k * kE8Seg, e = n - s < kE8Seg ? n : s + kE8Seg;
    const u64 cur = (e8_window(J.in, n, s) & ~0xffffffull) | truth;
    st_out[J.st_base + k] = e8_walk(J.in, J.out, n, s, e, s, cur, 0xffffffffu, nullptr);
    atomicAdd(refixed, 1u);
  }
}

}  // namespace

// d_jobs: device array of {in, out, len*, cap, st_base}; host copy tells the grid shape.  No host round trip.
int zpq_e8e9_inverse_launch(zpq_ctx* ctx, hipStream_t st, const void* d_jobs, size_t njobs, u32 max_cap, u32* d_state, size_t nstate) {
  if (njobs == 0) return ZPQ_OK;
  u32 warmup = 64;
  if (const char* e = getenv("ZPQ_E8_WARMUP")) warmup = (u32)atoi(e);
  u32* st_in = d_state; u32* st_out = d_state + nstate; u32* refixed = d_state + 2 * nstate;
  const unsigned gx = (unsigned)((((u64)max_cap + kE8Seg - 1) / kE8Seg + 255) / 256);
  ZPQ_LAUNCH(ctx, "e8e9_inverse_walk_kernel", st, e8e9_inverse_walk_kernel, dim3(gx ? gx : 1, (unsigned)njobs), dim3(256), (const E8InvJob*)d_jobs,
             warmup, st_in, st_out);
  ZPQ_HIP(ctx, hipGetLastError());
  ZPQ_LAUNCH(ctx, "e8e9_inverse_fix_kernel", st, e8e9_inverse_fix_kernel, dim3((unsigned)((njobs + 63) / 64)), dim3(64), (const E8InvJob*)d_jobs,
             (u32)njobs, st_in, st_out, refixed);
  ZPQ_HIP(ctx, hipGetLastError());
  return ZPQ_OK;
}

// forward transform without the host synchronisation (compressBlock's E8E9 front end); bits: (n+31)/32 words of scratch
int zpq_e8e9_forward_launch(zpq_ctx* ctx, hipStream_t st, u8* d_buf, size_t n, u32* bits) {
  if (n < 5) return ZPQ_OK;
  const size_t words = (n + 31) / 32;
  ZPQ_LAUNCH(ctx, "e8e9_mark_kernel", st, e8e9_mark_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), d_buf, (u64)n, bits);
  ZPQ_LAUNCH(ctx, "e8e9_forward_kernel", st, e8e9_forward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), d_buf, (u64)n, bits);
  ZPQ_HIP(ctx, hipGetLastError());
  return ZPQ_OK;
}

// Host-visible inverse over one buffer (out of place): d_out[0..n) = inverse of d_in[0..n).
extern "C" int zpq_e8e9_inverse_dev(zpq_ctx* ctx, const uint8_t* d_in, uint8_t* d_out, size_t n) {
  if (ctx) (void)hipSetDevice(ctx->device);
  if (!ctx) return ZPQ_ERR_ARG;
  if (n == 0) return ZPQ_OK;
  if (n > 0xfffffff0u) return zpq_fail(ctx, ZPQ_ERR_ARG, "e8e9: offsets are 32-bit in the reference");
  const size_t nseg = (n + kE8Seg - 1) / kE8Seg;
  u8* d = (u8*)zpq_scratch(ctx, 20, sizeof(E8InvJob) + 64 + (2 * nseg + 16) * 4);
  if (!d) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "e8e9 scratch");
  u32* d_len = (u32*)(d + sizeof(E8InvJob));
  u32* d_state = (u32*)(d + sizeof(E8InvJob) + 64);
  E8InvJob j; j.in = d_in; j.out = d_out; j.len = d_len; j.cap = (u32)n; j.st_base = 0;
  const u32 n32 = (u32)n;
  ZPQ_HIP(ctx, hipMemcpyAsync(d, &j, sizeof j, hipMemcpyHostToDevice, ctx->stream));
  ZPQ_HIP(ctx, hipMemcpyAsync(d_len, &n32, 4, hipMemcpyHostToDevice, ctx->stream));
  ZPQ_HIP(ctx, hipMemsetAsync(d_state + 2 * nseg, 0, 4, ctx->stream));
  ZPQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  int rc = zpq_e8e9_inverse_launch(ctx, ctx->stream, d, 1, (u32)n, d_state, nseg);
  if (rc) return rc;
  ZPQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZPQ_OK;
}

extern "C" int zpq_e8e9_dev(zpq_ctx* ctx, uint8_t* d_buf, size_t n) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (!ctx) return ZPQ_ERR_ARG;
  if (n < 5) return ZPQ_OK;
  if (n > 0xffffffffu) return zpq_fail(ctx, ZPQ_ERR_ARG, "e8e9: offsets are 32-bit in the reference (n <= 2^32-1)");
  const size_t words = (n + 31) / 32;
  u32* bits = (u32*)zpq_scratch(ctx, 1, words * 4 + 256);
  if (!bits) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "e8e9 scratch");
  ZPQ_LAUNCH(ctx, "e8e9_mark_kernel", ctx->stream, e8e9_mark_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), d_buf, (u64)n,
             bits);
  ZPQ_LAUNCH(ctx, "e8e9_forward_kernel", ctx->stream, e8e9_forward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), d_buf,
             (u64)n, bits);
  ZPQ_HIP(ctx, hipGetLastError());
  ZPQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZPQ_OK;
}
