// Test infrastructure: a stand-in for the HIP runtime and for zpq_internal.h, so that a WHOLE engine source file -- kernels
// AND host code -- compiles for the host: device memory is host memory, a kernel launch runs every workgroup on the fibre
// emulator (simt_emu.h), copies are memcpy, streams do nothing.  With it zpq_lz77_encode_dev() itself runs on the CPU.
#pragma once
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "simt_emu.h"
#include "zpaqhip.h"

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_u;
typedef u64 __attribute__((aligned(1))) u64_u;
typedef u32 __attribute__((aligned(1))) u32_u;

// ---- device vocabulary ----------------------------------------------------------------------------------------------------
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
struct EmuIdx { u32 x, y, z; };
static EmuIdx blockIdx, gridDim;
static inline EmuIdx emu_thread_idx() { return EmuIdx{(u32)emu::g_tid, 0, 0}; }
#define threadIdx (emu_thread_idx())
static inline int lane_id() { return emu::lane(); }
#define __ballot(p) emu::ballot((p), __LINE__)
#define __any(p) emu::any((p), __LINE__)
#define __all(p) emu::all((p), __LINE__)
#define __shfl(v, src) emu::shfl((v), (int)(src), __LINE__)
#define __shfl_xor(v, m) emu::shfl((v), emu::lane() ^ (int)(m), __LINE__)
#define __shfl_up(v, d) emu::shfl((v), emu::lane() >= (int)(d) ? emu::lane() - (int)(d) : emu::lane(), __LINE__)
#define __builtin_amdgcn_readlane(v, l) emu::shfl((v), (int)(l), __LINE__)
#define __builtin_amdgcn_readfirstlane(v) emu::shfl((v), 0, __LINE__)
#define __builtin_amdgcn_wave_barrier() ((void)emu::wave_rendezvous(0, __LINE__))
#define __builtin_amdgcn_s_setprio(x) ((void)0)

#define __syncthreads() emu::block_barrier(__LINE__)
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or((p), (v), __ATOMIC_RELAXED)
#define __hip_atomic_fetch_max(p, v, order, scope) __atomic_fetch_max((p), (v), __ATOMIC_RELAXED)
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_RELAXED)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_RELAXED)
#define __builtin_amdgcn_s_sleep(x) ((void)(emu::g_active ? (emu::to_main(), 0) : 0))      /* a sleeping wave lets the other waves of the workgroup run */
#define __HIP_MEMORY_SCOPE_WAVEFRONT 0
#define ZPQ_WAIT_VMCNT0 ((void)emu::wave_rendezvous(0, __LINE__))      /* lockstep: every lane's stores before anybody's next loads */
template <class T, class V> static inline T emu_atomic_add(T* p, V v) { const T o = *p; *p = o + (T)v; return o; }
template <class T, class V> static inline T emu_atomic_max(T* p, V v) { const T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class V> static inline T emu_atomic_or(T* p, V v) { const T o = *p; *p = o | (T)v; return o; }
#define atomicAdd(p, v) emu_atomic_add((p), (v))
#define atomicMax(p, v) emu_atomic_max((p), (v))
#define atomicOr(p, v) emu_atomic_or((p), (v))

// ---- runtime ------------------------------------------------------------------------------------------------------------------
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
struct dim3 { u32 x, y, z; dim3(u32 a = 1, u32 b = 1, u32 c = 1) : x(a), y(b), z(c) {} };
static const char* g_emu_launch_error = nullptr;
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
typedef void* hipEvent_t;
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = malloc(8); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) & ~(size_t)255); return *p ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
#define HIP_SYMBOL(x) (&(x))
template <class T> static inline hipError_t hipMemcpyToSymbolAsync(T* sym, const void* s, size_t n, size_t off, int, hipStream_t) { memcpy((char*)sym + off, s, n); return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return g_emu_launch_error ? hipErrorUnknown : hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return g_emu_launch_error ? g_emu_launch_error : "emulated HIP error"; }
static inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)48 << 30; *t = (size_t)64 << 30; return hipSuccess; }

static std::function<void()> g_emu_fn;
static inline void emu_fn_trampoline() { g_emu_fn(); }
template <class F> static inline void emu_launch(dim3 grid, dim3 block, F&& f) {
  g_emu_fn = f;
  gridDim = {grid.x, grid.y, grid.z};
  for (u32 z = 0; z < grid.z; ++z) for (u32 y = 0; y < grid.y; ++y) for (u32 x = 0; x < grid.x; ++x) {
    blockIdx = {x, y, z};
    if (const char* e = emu::run_block(emu_fn_trampoline, (int)((block.x + 63) / 64 * 64))) { if (!g_emu_launch_error) g_emu_launch_error = e; return; }
  }
}

// ---- zpq_internal.h ---------------------------------------------------------------------------------------------------------
#define ZPQ_SCRATCH_SLOTS 32
struct zpq_ctx {
  int device = 0; hipStream_t stream = nullptr, stream2 = nullptr; int cu_count = 256; std::string err;
  void* scratch[ZPQ_SCRATCH_SLOTS] = {nullptr}; size_t scratch_cap[ZPQ_SCRATCH_SLOTS] = {0}; bool profiling = false;
  ~zpq_ctx() { for (void* p : scratch) free(p); }
};
static inline void* zpq_scratch(zpq_ctx* ctx, int slot, size_t bytes) {
  if (bytes <= ctx->scratch_cap[slot] && ctx->scratch[slot]) return ctx->scratch[slot];
  free(ctx->scratch[slot]);
  const size_t cap = (bytes + bytes / 4 + 4096 + 255) & ~(size_t)255;
  ctx->scratch[slot] = aligned_alloc(256, cap);
  ctx->scratch_cap[slot] = ctx->scratch[slot] ? cap : 0;
  return ctx->scratch[slot];
}
static inline int zpq_fail(zpq_ctx* ctx, int status, const char* fmt, ...) {
  char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  if (ctx) ctx->err = buf;
  return status;
}
struct ZpqProfScope { ZpqProfScope(zpq_ctx*, const char*, hipStream_t) {} };
#define ZPQ_LAUNCH(ctx, name, st, kernel, grid, block, ...) emu_launch((grid), (block), [&] { kernel(__VA_ARGS__); })
#define ZPQ_HIP(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return zpq_fail((ctx), ZPQ_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)
struct zpq_lzjob_dev {
  const u8* in; u32 n; u32 rb; u32 nseg, seg0; u32* tok_pos; u32* tok_len; u32* tok_off; u32* tok_bit; u32 tok_cap; u32* result; u8* out; u32 out_cap; u32* plan;
};
struct zpq_place { u32* queue; u32* tab; u32 n; u32 polite; };
static inline u32 zpq_place_begin(const zpq_place& P, u32& key, bool& polite) { key = 0; polite = true; return blockIdx.x < P.n ? blockIdx.x : 0xffffffffu; }
static inline u32 zpq_place_next(const zpq_place&, u32, bool) { return 0xffffffffu; }
static inline bool zpq_place_enabled() { return false; }
static inline u32* zpq_simd_table(zpq_ctx*) { return nullptr; }
int zpq_lz77_sa_encode(zpq_ctx*, zpq_lz77_job*, const size_t*, size_t);
int zpq_lz77_pack_launch(zpq_ctx* ctx, const zpq_lzjob_dev* d_jobs, size_t nj, u32 max_n);
int zpq_lz77_pack2_launch(zpq_ctx* ctx, const zpq_lzjob_dev* h_jobs, const u32* min_match, size_t nj, u32 max_n);

// ---- rocPRIM: what the engine calls, by std::stable_sort ----------------------------------------------------------------------
namespace rocprim {
template <class K, class V>
hipError_t radix_sort_pairs(void* tmp, size_t& bytes, const K* kin, K* kout, const V* vin, V* vout, size_t n, unsigned b0, unsigned b1, hipStream_t) {
  if (!tmp) { bytes = 256; return hipSuccess; }
  std::vector<size_t> idx(n);
  for (size_t i = 0; i < n; ++i) idx[i] = i;
  const K mask = b1 - b0 >= 64 ? ~(K)0 : (((K)1 << (b1 - b0)) - 1);
  std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return ((kin[a] >> b0) & mask) < ((kin[b] >> b0) & mask); });
  for (size_t i = 0; i < n; ++i) { kout[i] = kin[idx[i]]; vout[i] = vin[idx[i]]; }
  return hipSuccess;
}
}  // namespace rocprim
