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

}  // namespace

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
