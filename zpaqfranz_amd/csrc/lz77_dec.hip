// LZ77 level-1 decoder: what the 302-byte level-1 PCOMP program (code format
// ZSFX/libzpaq.cpp:6211-6222) computes when PostProcessor (ZSFX/libzpaq.cpp:2178-2233) runs it per byte.
//
// One wave per block.  The bit parser is wave-uniform and never waits on memory: the code stream is held
// in registers as a sliding window (each lane keeps 8 bytes of the current 512-byte chunk and of the next
// one; 64 bits at any bit position are two v_readlane pairs), the interleaved Elias-gamma lengths are
// decoded with one count-trailing-zeros and a bit compaction instead of a bit loop.  Copies are spread over
// the lanes; the last 64 KiB of output live in an LDS ring so that near matches never wait on HBM stores.
//
// The stream is untrusted (archives come from anywhere): lengths are capped (<= 24 gamma doublings, far
// beyond what any encoder emits), capacity is checked in 64 bits, offsets must point into produced output,
// and far matches longer than their offset are copied in pieces that only read what is already written.
#include "zpq_internal.h"

namespace {

// The code stream and the output are addressed as GLOBAL memory, the ring and the staged chunks as LDS.  With generic
// pointers every access is a flat_* instruction, which counts on BOTH the vector-memory and the LDS counters: each LDS
// read of a match copy then waits for the HBM stores of the token before it (measured: 2600 cycles per token).
typedef __attribute__((address_space(1))) const u8 g_cu8;
typedef __attribute__((address_space(1))) u8 g_u8;
typedef __attribute__((address_space(1))) const u64_u g_cu64_u;
typedef __attribute__((address_space(3))) u8 l_u8;
typedef __attribute__((address_space(3))) u64 l_u64;
__device__ __forceinline__ u64 load8(g_cu8* p) { return *(g_cu64_u*)p; }

typedef zpq_lzdec_dev LzDecDev;

constexpr u32 kRing = 1u << 16;

__device__ __forceinline__ u64 readlane64(u64 v, u32 l) {
  const u32 lo = __builtin_amdgcn_readlane((u32)v, l), hi = __builtin_amdgcn_readlane((u32)(v >> 32), l);
  return (u64)lo | (u64)hi << 32;
}

// Interleaved Elias gamma, LSB first: pairs (1, bit) and a closing 0; the value has an implied leading 1 and
// the first bit read is the most significant.  w = the bits at the code; returns the value, nb = bits used
// (2k+1); k > 24 (or no terminator in the window) sets nb = 0xffffffff.
__device__ __forceinline__ u32 gamma_decode(u64 w, u32& nb) {
  const u32 w4 = (u32)w & 7u;
  if (!(w4 & 1u)) { nb = 1; return 1u; }               // "0": value 1
  if (!(w4 & 4u)) { nb = 3; return 2u | ((w4 >> 1) & 1u); }   // "1 b 0": value 2 + b
  const u64 z = ~w & 0x5555555555555555ull;            // a 0 flag at an even position ends the code
  const u32 p = z ? (u32)__builtin_ctzll(z) : 64u;
  if (p > 48) { nb = 0xffffffffu; return 1; }
  nb = p + 1;
  const u32 k = p >> 1;
  u64 x = (w >> 1) & 0x5555555555555555ull & ((1ull << p) - 1);   // data bits, now at even positions
  x = (x | (x >> 1)) & 0x3333333333333333ull;
  x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
  x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
  x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
  x = (x | (x >> 16)) & 0xffffffffull;
  const u32 y = (u32)x;                                 // bit i = i-th data bit read
  return (1u << k) | (k ? __builtin_bitreverse32(y) >> (32 - k) : 0u);
}

__global__ __launch_bounds__(64) void lz77_decode_kernel(const LzDecDev* __restrict__ jobs) {
  const LzDecDev J = jobs[blockIdx.x];
  __shared__ u8 ring_mem[kRing];
  __shared__ __attribute__((aligned(8))) u8 sbuf_mem[1024];    // the two code-stream chunks of the window, for literal runs
  l_u8* const ring = (l_u8*)ring_mem;
  l_u8* const sbuf = (l_u8*)sbuf_mem;
  g_u8* const out = (g_u8*)J.out;
  const u32 lane = (u32)lane_id();
  g_cu8* in = (g_cu8*)J.in;
  const u32 n = J.n;
  const u64 nbits = (u64)n * 8;
  u64 bp = 0; u32 op = 0; int status = ZPQ_OK;
  // code-stream window: chunk c = bytes [512c, 512c+512), 8 per lane; bytes at or past n read as 0
  auto load_chunk = [&](u32 c) -> u64 { const u64 o = (u64)c * 512 + (u64)lane * 8; return o < n ? load8(in + o) : 0ull; };
  u32 chunk = 0;
  u64 cur = load_chunk(0), nxt = load_chunk(1);
  *(l_u64*)(sbuf + lane * 8) = cur;
  *(l_u64*)(sbuf + 512 + lane * 8) = nxt;
  __builtin_amdgcn_wave_barrier();
  auto peek = [&](u64 b) -> u64 {                       // 64 bits of the stream from bit b on (b < nbits)
    const u32 byte = (u32)(b >> 3);
    const u32 c = __builtin_amdgcn_readfirstlane(byte >> 9);
    if (c != chunk) {
      const bool step = c == chunk + 1;
      cur = step ? nxt : load_chunk(c);
      nxt = load_chunk(c + 1);
      chunk = c;
      __builtin_amdgcn_wave_barrier();
      if (!step) *(l_u64*)(sbuf + (c & 1u) * 512 + lane * 8) = cur;
      *(l_u64*)(sbuf + ((c + 1) & 1u) * 512 + lane * 8) = nxt;
      __builtin_amdgcn_wave_barrier();
    }
    const u32 idx = byte & 511u;
    const u32 L = __builtin_amdgcn_readfirstlane(idx >> 3);
    const u32 sh = ((idx & 7u) << 3) | (u32)(b & 7);
    const u64 lo = readlane64(cur, L);
    const u64 hi = L == 63 ? readlane64(nxt, 0) : readlane64(cur, L + 1);
    return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
  };
  for (;;) {
    if (bp + 2 > nbits) break;
    u64 w = peek(bp);
    const u32 mmv = (u32)(w & 3);
    u32 used = 2; w >>= 2;
    if (mmv == 0) {                                   // literal run: gamma length then bytes
      u32 nb;
      u32 len = gamma_decode(w, nb);
      if (nb == 0xffffffffu) { if (bp + used + 50 <= nbits) status = ZPQ_ERR_FORMAT; break; }
      used += nb;
      if (bp + used > nbits) break;                    // stream ends inside the length code
      bp += used;
      const u64 avail = (nbits - bp) >> 3;
      const bool cutoff = avail < len;
      if (cutoff) len = (u32)avail;                    // stream ends inside the run
      if ((u64)op + len > J.out_cap) { status = ZPQ_ERR_CAPACITY; break; }
      const u32 byte0 = (u32)(bp >> 3), sh = (u32)(bp & 7);
      if (byte0 >= chunk * 512u && (u64)byte0 + len + 1 <= (u64)chunk * 512u + 1024u) {
        // the whole run (and the byte after it, for the bit shift) sits in the two chunks held in LDS
        for (u32 j = lane; j < len; j += 64) {
          const u32 q = byte0 + j;
          const u32 two = (u32)sbuf[q & 1023u] | ((u32)sbuf[(q + 1) & 1023u] << 8);
          const u8 c = (u8)(two >> sh);
          out[op + j] = c;
          ring[(op + j) & (kRing - 1)] = c;
        }
      } else {
        for (u32 j = lane; j < len; j += 64) {
          const u32 two = (u32)in[byte0 + j] | ((u32)in[byte0 + j + 1] << 8);
          const u8 c = (u8)(two >> sh);
          out[op + j] = c;
          ring[(op + j) & (kRing - 1)] = c;
        }
      }
      __builtin_amdgcn_wave_barrier();
      op += len; bp += 8ull * len;
      if (cutoff) break;
    } else {                                          // match
      if (bp + 5 > nbits) break;
      const u32 lo = (mmv - 1) * 8 + (u32)(w & 7); w >>= 3; used += 3;
      u32 nb;
      u32 len = gamma_decode(w, nb);
      if (nb == 0xffffffffu) { if (bp + used + 50 <= nbits) status = ZPQ_ERR_FORMAT; break; }
      w >>= nb; used += nb;
      if (bp + used + 2 > nbits) break;
      len = len * 4 + (u32)(w & 3); used += 2;         // <= 2^27
      w >>= 2;
      bp += used;                                      // used <= 2+3+49+2 = 56 bits
      if (bp + J.rb + lo > nbits) break;
      if (used + J.rb + lo > 64) w = peek(bp);         // (rare: the 64-bit window already holds the offset bits otherwise)
      const u32 r = (u32)(w & ((1ull << J.rb) - 1)); w >>= J.rb;
      const u32 qv = (u32)(w & ((1ull << lo) - 1)) | (1u << lo);
      bp += J.rb + lo;
      const u32 off = ((qv << J.rb) | r) - ((1u << J.rb) - 1u);
      if (off == 0 || off > op) { status = ZPQ_ERR_FORMAT; break; }
      if ((u64)op + len > J.out_cap) { status = ZPQ_ERR_CAPACITY; break; }
      const u32 src0 = op - off;
      if (off + 64 <= kRing) {                        // source inside the LDS ring
        for (u32 c0 = 0; c0 < len; c0 += 64) {
          const u32 j = c0 + lane;
          u8 c = 0;
          // off >= 64: out[op+j-off] was written before this chunk; off < 64: periodic extension
          if (j < len) c = ring[(off >= 64 ? op + j - off : src0 + (j % off)) & (kRing - 1)];
          __builtin_amdgcn_wave_barrier();
          if (j < len) { out[op + j] = c; ring[(op + j) & (kRing - 1)] = c; }
          __builtin_amdgcn_wave_barrier();
        }
      } else {                                        // far match: bytes written >= 64 KiB ago
        // pieces of at most 32 KiB (< off): every piece only reads what earlier pieces have stored.  The first piece
        // reads bytes at least 32 KiB behind the write frontier: hundreds of store instructions ago, long complete (at
        // most 63 memory operations are ever outstanding), so only later pieces of a self-overlapping match wait.
        // Loads bypass the L1 (a line cached while it was only partly written would be stale): L2 is where stores land.
        for (u32 c0 = 0; c0 < len; c0 += 32768u) {
          const u32 pl = len - c0 < 32768u ? len - c0 : 32768u;
          if (c0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          u32 j = lane;
          for (; j + 192 < pl; j += 256) {             // four loads in flight per lane
            g_cu8* sp = (g_cu8*)out + src0 + c0 + j;
            const u8 a0 = __builtin_nontemporal_load(sp), a1 = __builtin_nontemporal_load(sp + 64),
                     a2 = __builtin_nontemporal_load(sp + 128), a3 = __builtin_nontemporal_load(sp + 192);
            g_u8* dp = out + op + c0 + j;
            dp[0] = a0; dp[64] = a1; dp[128] = a2; dp[192] = a3;
            const u32 r0 = op + c0 + j;
            ring[r0 & (kRing - 1)] = a0; ring[(r0 + 64) & (kRing - 1)] = a1; ring[(r0 + 128) & (kRing - 1)] = a2; ring[(r0 + 192) & (kRing - 1)] = a3;
          }
          for (; j < pl; j += 64) {
            const u8 c = __builtin_nontemporal_load((g_cu8*)out + src0 + c0 + j);
            out[op + c0 + j] = c;
            ring[(op + c0 + j) & (kRing - 1)] = c;
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
      op += len;
    }
  }
  if (lane == 0) { J.result[0] = op; J.result[1] = (u32)status; }
}

}  // namespace

// Launch only (no host round trip): results land in d_res[2*i] = out_len, d_res[2*i+1] = status.
int zpq_lz77_decode_launch(zpq_ctx* ctx, hipStream_t st, const zpq_lzdec_dev* d_jobs, size_t njobs) {
  ZPQ_LAUNCH(ctx, "lz77_decode_kernel", st, lz77_decode_kernel, dim3((unsigned)njobs), dim3(64), d_jobs);
  ZPQ_HIP(ctx, hipGetLastError());
  return ZPQ_OK;
}

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
  int rc = zpq_lz77_decode_launch(ctx, st, d_jobs, njobs);
  if (rc) return rc;
  std::vector<u32> res(njobs * 2);
  ZPQ_HIP(ctx, hipMemcpyAsync(res.data(), d_res, njobs * 8, hipMemcpyDeviceToHost, st));
  ZPQ_HIP(ctx, hipStreamSynchronize(st));
  for (size_t i = 0; i < njobs; ++i) { jobs[i].out_len = res[2 * i]; jobs[i].status = (int32_t)res[2 * i + 1]; }
  return ZPQ_OK;
}
