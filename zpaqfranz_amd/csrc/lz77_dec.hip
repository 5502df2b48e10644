// LZ77 level-1 decoder: what the 302-byte level-1 PCOMP program (code format
// ZSFX/libzpaq.cpp:6211-6222) computes when PostProcessor (ZSFX/libzpaq.cpp:2178-2233) runs it per byte.
// One wave per block: the bit parser is wave-uniform, copies are spread over the lanes, and the last
// 64 KiB of output live in an LDS ring so that near matches never wait on HBM stores.
#include "zpq_internal.h"

namespace {

__device__ __forceinline__ u64 load8(const u8* p) { return *(const u64_u*)p; }

// ---- decoder ---------------------------------------------------------------------------------------
struct LzDecDev {
  const u8* in; u32 n; u32 rb;
  u8* out; u32 out_cap;
  u32* result;  // [0]=out_len, [1]=status
};

constexpr u32 kRing = 1u << 16;

// One wave per block.  The bit parser is wave-uniform; copies are spread over the lanes.  The last
// 64 KiB of output live in an LDS ring so that near matches never wait on HBM stores.
__global__ __launch_bounds__(64) void lz77_decode_kernel(const LzDecDev* __restrict__ jobs) {
  const LzDecDev J = jobs[blockIdx.x];
  __shared__ u8 ring[kRing];
  const u32 lane = (u32)lane_id();
  const u8* in = J.in;
  const u64 nbits = (u64)J.n * 8;
  u64 bp = 0; u32 op = 0; int status = ZPQ_OK;
#define PEEK() (load8(in + (bp >> 3)) >> (bp & 7))
  for (;;) {
    if (bp + 2 > nbits) break;
    u64 w = PEEK();
    const u32 mmv = (u32)(w & 3);
    u64 used = 2; w >>= 2;
    if (mmv == 0) {                                   // literal run: gamma length then bytes
      u32 len = 1; bool trunc = false;
      for (;;) {
        if (bp + used + 1 > nbits) { trunc = true; break; }
        const u32 b = (u32)(w & 1); w >>= 1; ++used;
        if (!b) break;
        if (bp + used + 1 > nbits) { trunc = true; break; }
        len = len * 2 + (u32)(w & 1); w >>= 1; ++used;
      }
      if (trunc) break;
      bp += used;
      const u64 avail = (nbits - bp) >> 3;
      const bool cutoff = avail < len;
      if (cutoff) len = (u32)avail;                    // stream ends inside the run
      if (op + len > J.out_cap) { status = ZPQ_ERR_CAPACITY; break; }
      for (u32 j = lane; j < len; j += 64) {
        const u64 b = bp + 8ull * j;
        const u32 two = (u32)in[b >> 3] | ((u32)in[(b >> 3) + 1] << 8);
        const u8 c = (u8)(two >> (b & 7));
        J.out[op + j] = c;
        ring[(op + j) & (kRing - 1)] = c;
      }
      __builtin_amdgcn_wave_barrier();
      op += len; bp += 8ull * len;
      if (cutoff) break;
    } else {                                          // match
      if (bp + 5 > nbits) break;
      const u32 lo = (mmv - 1) * 8 + (u32)(w & 7); w >>= 3; used += 3;
      u32 len = 1; bool trunc = false;
      for (;;) {
        if (bp + used + 1 > nbits) { trunc = true; break; }
        const u32 b = (u32)(w & 1); w >>= 1; ++used;
        if (!b) break;
        if (bp + used + 1 > nbits) { trunc = true; break; }
        len = len * 2 + (u32)(w & 1); w >>= 1; ++used;
      }
      if (trunc || bp + used + 2 > nbits) break;
      len = len * 4 + (u32)(w & 3); used += 2;
      bp += used;                                     // used <= 2+3+2*15+1+2 = 38 bits
      if (bp + J.rb + lo > nbits) break;
      w = PEEK();
      const u32 r = (u32)(w & ((1ull << J.rb) - 1)); w >>= J.rb;
      const u32 qv = (u32)(w & ((1ull << lo) - 1)) | (1u << lo);
      bp += J.rb + lo;
      const u32 off = ((qv << J.rb) | r) - ((1u << J.rb) - 1u);
      if (off > op) { status = ZPQ_ERR_FORMAT; break; }
      if (op + len > J.out_cap) { status = ZPQ_ERR_CAPACITY; break; }
      const u32 src0 = op - off;
      if (off + 64 <= kRing) {                        // source inside the LDS ring
        for (u32 c0 = 0; c0 < len; c0 += 64) {
          const u32 j = c0 + lane;
          u8 c = 0;
          // off >= 64: out[op+j-off] was written before this chunk; off < 64: periodic extension
          if (j < len) c = ring[(off >= 64 ? op + j - off : src0 + (j % off)) & (kRing - 1)];
          __builtin_amdgcn_wave_barrier();
          if (j < len) { J.out[op + j] = c; ring[(op + j) & (kRing - 1)] = c; }
          __builtin_amdgcn_wave_barrier();
        }
      } else {                                        // far match: bytes written >= 64 KiB ago, len < off
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (u32 j = lane; j < len; j += 64) {
          const u8 c = __builtin_nontemporal_load(J.out + src0 + j);
          J.out[op + j] = c;
          ring[(op + j) & (kRing - 1)] = c;
        }
        __builtin_amdgcn_wave_barrier();
      }
      op += len;
    }
  }
#undef PEEK
  if (lane == 0) { J.result[0] = op; J.result[1] = (u32)status; }
}

}  // namespace

extern "C" int zpq_lz77_decode_dev(zpq_ctx* ctx, zpq_lz77_dec_job* jobs, size_t njobs) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (njobs == 0) return ZPQ_OK;
  hipStream_t st = ctx->stream;
  u8* d_meta = (u8*)zpq_scratch(ctx, 2, njobs * (sizeof(LzDecDev) + 8) + 64);
  if (!d_meta) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "lz77 scratch");
  LzDecDev* d_jobs = (LzDecDev*)d_meta;
  u32* d_res = (u32*)(d_meta + njobs * sizeof(LzDecDev));
  std::vector<LzDecDev> h(njobs);
  for (size_t i = 0; i < njobs; ++i) {
    if (jobs[i].rb > 8) return zpq_fail(ctx, ZPQ_ERR_ARG, "rb out of range");
    h[i].in = jobs[i].d_in; h[i].n = jobs[i].n; h[i].rb = jobs[i].rb;
    h[i].out = jobs[i].d_out; h[i].out_cap = jobs[i].out_cap; h[i].result = d_res + 2 * i;
  }
  ZPQ_HIP(ctx, hipMemcpyAsync(d_jobs, h.data(), njobs * sizeof(LzDecDev), hipMemcpyHostToDevice, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  ZPQ_LAUNCH(ctx, "lz77_decode_kernel", st, lz77_decode_kernel, dim3((unsigned)njobs), dim3(64), d_jobs);
  ZPQ_HIP(ctx, hipGetLastError());
  std::vector<u32> res(njobs * 2);
  ZPQ_HIP(ctx, hipMemcpyAsync(res.data(), d_res, njobs * 8, hipMemcpyDeviceToHost, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  for (size_t i = 0; i < njobs; ++i) { jobs[i].out_len = res[2 * i]; jobs[i].status = (int32_t)res[2 * i + 1]; }
  return ZPQ_OK;
}
