// Internal declarations shared by the HIP translation units behind include/zpaqhip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "zpaqhip.h"

struct zpq_ctx {
  int device;
  hipStream_t stream;
  hipStream_t stream2;      // second stream: block SHA-1 chains overlap the LZ77 parse
  hipEvent_t ev;
  int cu_count;
  std::string err;
  // grow-only device scratch arena (avoids hipMalloc in the steady state)
  void* scratch[12];
  size_t scratch_cap[12];
  void* pinned;             // pinned host staging
  size_t pinned_cap;
  // optional per-kernel event timing
  bool profiling;
  struct ProfRec { const char* name; hipEvent_t a, b; };
  std::vector<ProfRec> prof;
};

// Brackets one kernel launch with events on its stream when profiling is on.
struct ZpqProfScope {
  zpq_ctx* c; hipStream_t s; hipEvent_t a, b; bool on;
  ZpqProfScope(zpq_ctx* ctx, const char* name, hipStream_t st) : c(ctx), s(st), on(ctx->profiling) {
    if (on) {
      (void)hipEventCreate(&a); (void)hipEventCreate(&b);
      (void)hipEventRecord(a, s);
      c->prof.push_back({name, a, b});
    }
  }
  ~ZpqProfScope() { if (on) (void)hipEventRecord(b, s); }
};
#define ZPQ_LAUNCH(ctx, name, st, kernel, grid, block, ...)                        \
  do {                                                                             \
    ZpqProfScope prof_scope_((ctx), (name), (st));                                 \
    hipLaunchKernelGGL(kernel, grid, block, 0, (st), __VA_ARGS__);                 \
  } while (0)

// Returns a device scratch buffer of at least `bytes` in slot `slot` (grow-only).
void* zpq_scratch(zpq_ctx* ctx, int slot, size_t bytes);
void* zpq_pinned(zpq_ctx* ctx, size_t bytes);
int zpq_fail(zpq_ctx* ctx, int status, const char* fmt, ...);

#define ZPQ_HIP(ctx, call)                                                                  \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return zpq_fail((ctx), ZPQ_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                      __FILE__, __LINE__);                                                  \
  } while (0)

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

// unaligned vector loads: gfx950 runs with unaligned access mode on, hipcc emits a single
// global_load_dwordx4 / dwordx2 for these.
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_u;
typedef u64 __attribute__((aligned(1))) u64_u;
typedef u32 __attribute__((aligned(1))) u32_u;

static __device__ __forceinline__ u32 bswap32(u32 x) { return __builtin_bswap32(x); }
static __device__ __forceinline__ u32 rotl32(u32 x, int k) { return __builtin_rotateleft32(x, k); }
static __device__ __forceinline__ u32 rotr32(u32 x, int k) { return __builtin_rotateright32(x, k); }
static __device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// internal cross-TU entry points
// method string -> expanded x/0 method, $1..$9, block header bytes (hsize..HCOMP 0) and PCOMP bytecode
// (config.hip; host_data is only read for level 5)
int zpq_build_config(zpq_ctx* ctx, const char* method, const u8* host_data, u32 n, std::string* xmethod, int args[9],
                     std::vector<u8>* header, std::vector<u8>* pcomp);
// one WAVE per extent (long chains: block checksums); zpq_sha1_extents_on uses one LANE per extent
int zpq_sha1_chains_on(zpq_ctx* ctx, hipStream_t s, const u8* d_base, const u64* d_off, const u32* d_len, size_t n,
                       u8* d_digests);
int zpq_sha1_extents_on(zpq_ctx* ctx, hipStream_t s, const u8* d_base, const u64* d_off,
                        const u32* d_len, size_t n, u8* d_digests, const char* prof_name = "sha1_extents_kernel");
