// Test infrastructure (see ../simt_state.h): <hip/hip_runtime.h> for the host build of the engine's sources.
#pragma once
#include <dlfcn.h>
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

#include "../../simt_emu.h"
#include "../simt_state.h"

// ---- device vocabulary ----------------------------------------------------------------------------------------------------
#define __device__
#define __host__
#define __global__
#define __constant__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static __attribute__((aligned(64)))
#define HIP_SYMBOL(x) (&(x))
static inline EmuIdx emu_thread_idx() { return EmuIdx{(unsigned)emu::g_tid, 0, 0}; }
#define threadIdx (emu_thread_idx())
#define warpSize 64
#define __ballot(p) emu::ballot((p), __LINE__)
#define __builtin_amdgcn_ballot_w64(p) emu::ballot((p), __LINE__)
#define __any(p) emu::any((p), __LINE__)
#define __all(p) emu::all((p), __LINE__)
#define __shfl(v, src, ...) emu::shfl((v), (int)(src), __LINE__)
#define __shfl_xor(v, m, ...) emu::shfl((v), emu::lane() ^ (int)(m), __LINE__)
#define __shfl_up(v, d, ...) emu::shfl((v), emu::lane() >= (int)(d) ? emu::lane() - (int)(d) : emu::lane(), __LINE__)
#define __shfl_down(v, d, ...) emu::shfl((v), emu::lane() + (int)(d) < 64 ? emu::lane() + (int)(d) : emu::lane(), __LINE__)
#define __builtin_amdgcn_readlane(v, l) emu::shfl((v), (int)(l), __LINE__)
#define __builtin_amdgcn_readfirstlane(v) emu::first((v), __LINE__)
#define __builtin_amdgcn_ds_bpermute(addr, v) emu::shfl((v), (int)((unsigned)(addr) >> 2), __LINE__)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rows, banks, bc) emu::dpp((old), (src), (ctrl), (rows), __LINE__)
#define __builtin_amdgcn_wave_barrier() ((void)emu::wave_rendezvous(0, __LINE__))
#define __builtin_amdgcn_s_waitcnt(x) ((void)emu::wave_rendezvous(0, __LINE__))       /* lockstep: everybody's memory operations up to here, then on */
#define EMU_WAIT_VMCNT0 ((void)emu::wave_rendezvous(0, __LINE__))
#define __builtin_amdgcn_fence(...) ((void)0)
// a sleeping wave lets the others run: a spin loop that waits for ANOTHER wave of the workgroup must let that wave's fibres have their turn
#define __builtin_amdgcn_s_sleep(x) ((void)(emu::g_active ? (emu::to_main(), 0) : 0))
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_getreg(x) (0u)
#define __syncthreads() emu::block_barrier(__LINE__)
#define __threadfence_block() ((void)0)
#define __threadfence() ((void)0)
#define __popcll(x) __builtin_popcountll(x)
#define __popc(x) __builtin_popcount(x)
#define __HIP_MEMORY_SCOPE_WAVEFRONT 0
#define __HIP_MEMORY_SCOPE_WORKGROUP 1
#define __HIP_MEMORY_SCOPE_AGENT 2
#define __HIP_MEMORY_SCOPE_SYSTEM 3
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or((p), (v), __ATOMIC_RELAXED)
#define __hip_atomic_fetch_max(p, v, order, scope) __atomic_fetch_max((p), (v), __ATOMIC_RELAXED)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), __ATOMIC_RELAXED)
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_RELAXED)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_RELAXED)
// one workgroup runs at a time and its fibres take turns: plain read-modify-write
template <class T, class V> static inline T atomicAdd(T* p, V v) { const T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class V> static inline T atomicSub(T* p, V v) { const T o = *p; *p = (T)(o - (T)v); return o; }
template <class T, class V> static inline T atomicMax(T* p, V v) { const T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class V> static inline T atomicMin(T* p, V v) { const T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class V> static inline T atomicOr(T* p, V v) { const T o = *p; *p = (T)(o | (T)v); return o; }
template <class T, class V> static inline T atomicAnd(T* p, V v) { const T o = *p; *p = (T)(o & (T)v); return o; }
template <class T, class V> static inline T atomicExch(T* p, V v) { const T o = *p; *p = (T)v; return o; }
template <class T, class U, class V> static inline T atomicCAS(T* p, U cmp, V v) { const T o = *p; if (o == (T)cmp) *p = (T)v; return o; }

// ---- runtime ------------------------------------------------------------------------------------------------------------------
typedef struct EmuStream* hipStream_t;
typedef struct EmuEvent* hipEvent_t;
typedef void* hipModule_t;
typedef void (*hipFunction_t)(void**);
typedef int hipError_t;
typedef int hipMemcpyKind;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorUnknown = 999 };
enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct hipDeviceProp_t {
  char name[256]; char gcnArchName[256]; int multiProcessorCount, clockRate, memoryClockRate, memoryBusWidth, l2CacheSize; size_t totalGlobalMem;
};
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
template <class T> static inline hipError_t hipMemcpyFromSymbol(void* d, T* sym, size_t n) { memcpy(d, (const void*)sym, n); return hipSuccess; }
template <class T> static inline hipError_t hipMemcpyToSymbol(T* sym, const void* s, size_t n) { memcpy((void*)sym, s, n); return hipSuccess; }
template <class T> static inline hipError_t hipMemcpyToSymbolAsync(T* sym, const void* s, size_t n, size_t off, int, hipStream_t) {
  if (getenv("EMU_TRACE")) fprintf(stderr, "[emu] %zu bytes to a device symbol, first word %u\n", n, n >= 4 ? *(const unsigned*)s : 0u);
  memcpy((char*)sym + off, s, n); return hipSuccess;
}
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
static inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, unsigned, const unsigned*) { *s = (hipStream_t)malloc(8); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 0; return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof *p);
  strcpy(p->name, "fibre emulator (tests/cpp/emu_rt)"); strcpy(p->gcnArchName, "gfx950:emulated");
  p->multiProcessorCount = 8; p->clockRate = 2400000; p->memoryClockRate = 2000000; p->memoryBusWidth = 8192; p->l2CacheSize = 4 << 20;
  p->totalGlobalMem = (size_t)16 << 30;
  return hipSuccess;
}
static inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)6 << 30; *t = (size_t)16 << 30; return hipSuccess; }
// device memory: host memory with a canary behind it (checked when it is freed); in the AddressSanitizer build the exact
// size, so that the first byte behind an allocation is a red zone
#if defined(EMU_ASAN)
static inline hipError_t hipMalloc(void** p, size_t n) { return posix_memalign(p, 256, n ? n : 1) == 0 ? hipSuccess : hipErrorOutOfMemory; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
#else
static inline hipError_t hipMalloc(void** p, size_t n) {
  char* b = (char*)aligned_alloc(256, ((n + 255) & ~(size_t)255) + 512);
  if (!b) return hipErrorOutOfMemory;
  memcpy(b, &n, sizeof n);
  memset(b + 256 + n, 0xA5, ((n + 255) & ~(size_t)255) - n + 256);
  *p = b + 256;
  return hipSuccess;
}
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  char* b = (char*)p - 256;
  size_t n; memcpy(&n, b, sizeof n);
  const size_t tail = ((n + 255) & ~(size_t)255) - n + 256;
  for (size_t i = 0; i < tail; ++i) if ((unsigned char)b[256 + n + i] != 0xA5) { fprintf(stderr, "[emu] write behind a device allocation of %zu bytes (+%zu)\n", n, i); abort(); }
  free(b);
  return hipSuccess;
}
#endif
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
EMU_VAR const char* g_emu_last_error_text;
static inline hipError_t hipGetLastError() {        // (reports a failed launch once, like the real call)
  if (!g_emu_launch_error) return hipSuccess;
  g_emu_last_error_text = g_emu_launch_error; g_emu_launch_error = nullptr;
  return hipErrorUnknown;
}
static inline const char* hipGetErrorString(hipError_t) { return g_emu_last_error_text ? g_emu_last_error_text : "emulated HIP error"; }
template <class F> static inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }

// a launch: every workgroup, one after the other, on the fibre emulator
EMU_VAR std::function<void()>* g_emu_fn;
EMU_VAR std::mutex g_emu_mu;                 // host threads (the shim's workers) launch one after the other
static inline void emu_fn_trampoline() { (*g_emu_fn)(); }
template <class F> static inline void emu_launch(dim3 grid, dim3 block, F&& f) {
  std::lock_guard<std::mutex> emu_lock(g_emu_mu);
  if (g_emu_launch_error) return;
  std::function<void()> fn = f;
  g_emu_fn = &fn;
  gridDim = {grid.x, grid.y, grid.z};
  blockDim = {block.x, block.y, block.z};
  ++g_emu_launches;
  const int threads = (int)(block.x * block.y * block.z);
  for (unsigned z = 0; z < grid.z; ++z) for (unsigned y = 0; y < grid.y; ++y) for (unsigned x = 0; x < grid.x; ++x) {
    blockIdx = {x, y, z};
    ++g_emu_blocks;
    if (const char* e = emu::run_block(emu_fn_trampoline, threads)) { g_emu_launch_error = e; fprintf(stderr, "[emu] %s\n", e); return; }
  }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_launch((grid), (block), [&] { kernel(__VA_ARGS__); })
static inline hipError_t hipModuleLaunchKernel(hipFunction_t f, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz, unsigned, hipStream_t,
                                               void** args, void**) {
  emu_launch(dim3(gx, gy, gz), dim3(bx, by, bz), [&] { f(args); });
  return hipSuccess;
}
// a "code object" is the path of a shared object the stand-in hiprtc has built (hiprtc.h)
static inline hipError_t hipModuleLoadData(hipModule_t* m, const void* image) {
  *m = dlopen((const char*)image, RTLD_NOW | RTLD_LOCAL);
  if (!*m) { fprintf(stderr, "[emu] dlopen: %s\n", dlerror()); return hipErrorUnknown; }
  return hipSuccess;
}
static inline hipError_t hipModuleGetFunction(hipFunction_t* f, hipModule_t m, const char* name) {
  const std::string s = std::string(name) + "__emu";
  *f = (hipFunction_t)dlsym(m, s.c_str());
  return *f ? hipSuccess : hipErrorUnknown;
}
static inline hipError_t hipModuleUnload(hipModule_t m) { if (m) dlclose(m); return hipSuccess; }
