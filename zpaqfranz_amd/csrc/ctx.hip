// Context, error reporting and device-memory helpers of the C ABI (include/zpaqhip.h).
#include <stdarg.h>

#include <algorithm>
#include <atomic>
#include <vector>

#include "zpq_internal.h"

int zpq_fail(zpq_ctx* ctx, int status, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return status;
}

// ---- making room: every context keeps up to 6 GiB of idle pooled blocks, twelve contexts run beside each other in
// the bench, and until round 6 only the pool's own allocator ever gave them back.  Now any device allocation of the engine that
// fails first releases the idle blocks of its own context, then those of every other live context on the device, and tries again.
static std::mutex g_ctx_mu;
static std::vector<zpq_ctx*> g_ctxs;

static size_t pool_trim(zpq_ctx* c) {
  std::lock_guard<std::mutex> g(c->pool_mu);
  size_t freed = 0;
  for (size_t i = c->pool.size(); i-- > 0;)
    if (!c->pool[i].in_use) {
      // (an idle block's last user has finished or is ordered before this on the device: hipFree waits for the device)
      freed += c->pool[i].cap;
      (void)hipFree(c->pool[i].p);
      c->pool.erase(c->pool.begin() + (long)i);
    }
  return freed;
}

hipError_t zpq_device_malloc(zpq_ctx* ctx, void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess) return e;
  (void)hipGetLastError();                       // (a failed hipMalloc must not surface as the "last error" of the next launch)
  if (pool_trim(ctx)) {
    if ((e = hipMalloc(p, bytes)) == hipSuccess) return e;
    (void)hipGetLastError();
  }
  size_t freed = 0;
  {
    std::lock_guard<std::mutex> g(g_ctx_mu);
    for (zpq_ctx* c : g_ctxs)
      if (c != ctx && c->device == ctx->device) freed += pool_trim(c);
  }
  if (freed) {
    if ((e = hipMalloc(p, bytes)) == hipSuccess) return e;
    (void)hipGetLastError();
  }
  return e;
}

void* zpq_scratch(zpq_ctx* ctx, int slot, size_t bytes) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (bytes <= ctx->scratch_cap[slot] && ctx->scratch[slot]) return ctx->scratch[slot];
  if (ctx->scratch[slot]) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamSynchronize(ctx->stream2); (void)hipFree(ctx->scratch[slot]); }
  // grow-only with some slack against re-allocating for every slightly larger request -- a quarter, but never more than 256 MiB:
  // the context-mixing models of a batch are sized to what HBM holds (hundreds of GB), and a quarter on top of that was what
  // failed, not the request (round 6: 2816 blocks x 84 MiB = 248 GB asked for 310)
  size_t cap = bytes + std::min<size_t>(bytes / 4, (size_t)256 << 20) + 4096;
  void* p = nullptr;
  if (zpq_device_malloc(ctx, &p, cap) != hipSuccess) { ctx->scratch[slot] = nullptr; ctx->scratch_cap[slot] = 0; return nullptr; }
  ctx->scratch[slot] = p;
  ctx->scratch_cap[slot] = cap;
  return p;
}

void* zpq_pinned(zpq_ctx* ctx, size_t bytes) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  if (bytes <= ctx->pinned_cap && ctx->pinned) return ctx->pinned;
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  size_t cap = bytes + bytes / 4 + 4096;
  void* p = nullptr;
  if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) { ctx->pinned = nullptr; ctx->pinned_cap = 0; return nullptr; }
  ctx->pinned = p;
  ctx->pinned_cap = cap;
  return p;
}

// ---- cooperative wave placement (zpq_internal.h) -----------------------------------------------------------------
u32* zpq_simd_table(zpq_ctx* ctx) {
  static std::mutex mu;
  static u32* tab[64] = {nullptr};
  std::lock_guard<std::mutex> g(mu);
  const int d = ctx->device & 63;
  if (!tab[d]) {
    (void)hipSetDevice(ctx->device);
    void* p = nullptr;
    if (hipMalloc(&p, 65536 * 4) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 65536 * 4) != hipSuccess) { (void)hipFree(p); return nullptr; }
    tab[d] = (u32*)p;
  }
  return tab[d];
}
static std::atomic<int> g_live_contexts{0};
bool zpq_place_enabled() {
  static const int mode = [] { const char* e = getenv("ZPQ_PLACE"); return e ? atoi(e) : 0; }();   // 0 off, 1 with several contexts, 2 always
  return mode == 2 || (mode == 1 && g_live_contexts.load(std::memory_order_relaxed) > 1);
}

// ZPQ_PLACE_DEBUG: how many distinct SIMD keys a chip-wide launch sees (1024 expected on MI355X)
__global__ __launch_bounds__(64) void place_probe_kernel(u32* __restrict__ out) {
  if ((threadIdx.x & 63u) == 0) out[blockIdx.x] = zpq_simd_key();
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}
static void place_probe(zpq_ctx* c) {
  const u32 n = 8192;
  u32* d = nullptr;
  if (hipMalloc((void**)&d, n * 4) != hipSuccess) return;
  hipLaunchKernelGGL(place_probe_kernel, dim3(n), dim3(64), 0, c->stream, d);
  std::vector<u32> h(n);
  (void)hipMemcpyAsync(h.data(), d, n * 4, hipMemcpyDeviceToHost, c->stream);
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  std::vector<u32> u(h); std::sort(u.begin(), u.end()); u.erase(std::unique(u.begin(), u.end()), u.end());
  u32 xcc = 0; for (u32 k : u) xcc |= 1u << (k >> 12);
  fprintf(stderr, "[zpaqhip] placement probe: %u workgroups landed on %zu distinct SIMD keys, XCC mask %#x, first keys %#x %#x %#x %#x\n", n, u.size(), xcc,
          h[0], h[1], h[8], h[9]);
}

int zpq_live_contexts() { return g_live_contexts.load(std::memory_order_relaxed); }

extern "C" {

int zpq_create(int device_ordinal, zpq_ctx** out) {
  if (!out) return ZPQ_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ZPQ_ERR_NO_DEVICE;
  if (device_ordinal < 0 || device_ordinal >= ndev) return ZPQ_ERR_ARG;
  if (hipSetDevice(device_ordinal) != hipSuccess) return ZPQ_ERR_HIP;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) return ZPQ_ERR_HIP;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ZPQ_ERR_NO_DEVICE;  // kernels are gfx950-only
  zpq_ctx* c = new zpq_ctx();
  c->device = device_ordinal;
  c->cu_count = prop.multiProcessorCount;
  for (int i = 0; i < ZPQ_SCRATCH_SLOTS; ++i) c->scratch[i] = nullptr, c->scratch_cap[i] = 0;
  c->pinned = nullptr;
  c->pinned_cap = 0;
  c->profiling = false;
  // The serial chains (block SHA-1 / SHA-256: one wave each, bound by dependent issue) run on stream2.  A chain shares
  // its SIMD's issue slots with whatever else is resident there, and with several jobs in flight the chip-wide passes of
  // the other jobs stretch it 2.5x (215 -> 550 ms for a 16 MiB block).  ZPQ_CHAIN_CUS=N gives the chains N compute units
  // of their own: stream2 is created with a CU mask of the first N units, the main stream with the complement.
  int chain_cus = 0;
  if (const char* e = getenv("ZPQ_CHAIN_CUS")) chain_cus = atoi(e);
  bool masked = false;
  if (chain_cus > 0 && chain_cus < c->cu_count) {
    const uint32_t words = (uint32_t)((c->cu_count + 31) / 32);
    std::vector<uint32_t> m_chain(words, 0), m_main(words, 0);
    for (int i = 0; i < c->cu_count; ++i) (i < chain_cus ? m_chain : m_main)[i / 32] |= 1u << (i % 32);
    masked = hipExtStreamCreateWithCUMask(&c->stream, words, m_main.data()) == hipSuccess;
    if (masked && hipExtStreamCreateWithCUMask(&c->stream2, words, m_chain.data()) != hipSuccess) {
      (void)hipStreamDestroy(c->stream);
      masked = false;
    }
    if (!masked) (void)hipGetLastError();
  }
  if ((!masked && (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
                   hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess)) ||
      hipEventCreateWithFlags(&c->ev, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev2, hipEventDisableTiming) != hipSuccess) {
    delete c;
    return ZPQ_ERR_HIP;
  }
  if (getenv("ZPQ_PLACE_DEBUG")) place_probe(c);
  g_live_contexts.fetch_add(1, std::memory_order_relaxed);
  { std::lock_guard<std::mutex> g(g_ctx_mu); g_ctxs.push_back(c); }
  *out = c;
  return ZPQ_OK;
}

void zpq_destroy(zpq_ctx* ctx) {
  if (!ctx) return;
  g_live_contexts.fetch_sub(1, std::memory_order_relaxed);
  { std::lock_guard<std::mutex> g(g_ctx_mu); g_ctxs.erase(std::remove(g_ctxs.begin(), g_ctxs.end(), ctx), g_ctxs.end()); }
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipStreamSynchronize(ctx->stream2);
  for (int i = 0; i < ZPQ_SCRATCH_SLOTS; ++i)
    if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  for (const zpq_ctx::PoolBlock& b : ctx->pool) (void)hipFree(b.p);
  (void)hipEventDestroy(ctx->ev);
  (void)hipEventDestroy(ctx->ev2);
  (void)hipStreamDestroy(ctx->stream);
  (void)hipStreamDestroy(ctx->stream2);
  delete ctx;
}

const char* zpq_strerror(int status) {
  switch (status) {
    case ZPQ_OK: return "ok";
    case ZPQ_ERR_NO_DEVICE: return "no gfx950 HIP device";
    case ZPQ_ERR_HIP: return "HIP runtime error";
    case ZPQ_ERR_ARG: return "invalid argument";
    case ZPQ_ERR_CAPACITY: return "output capacity too small";
    case ZPQ_ERR_METHOD: return "method not implemented by this engine";
    case ZPQ_ERR_FORMAT: return "malformed ZPAQ block";
    case ZPQ_ERR_CHECKSUM: return "SHA-1 mismatch";
    case ZPQ_ERR_NOMEM: return "out of memory";
    case ZPQ_ERR_LIMIT: return "a ZPAQL program of the block used up the engine's loop budget";
    default: return "unknown status";
  }
}

const char* zpq_last_error(const zpq_ctx* ctx) { return ctx ? ctx->err.c_str() : ""; }

int zpq_sync(zpq_ctx* ctx) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  ZPQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ZPQ_HIP(ctx, hipStreamSynchronize(ctx->stream2));
  return ZPQ_OK;
}

void* zpq_stream(zpq_ctx* ctx) { return (void*)ctx->stream; }

int zpq_device_info(zpq_ctx* ctx, int64_t info[6], char* name, size_t name_cap) {
  hipDeviceProp_t prop;
  ZPQ_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
  info[0] = prop.multiProcessorCount;
  info[1] = prop.clockRate;
  info[2] = prop.memoryClockRate;
  info[3] = prop.memoryBusWidth;
  info[4] = prop.l2CacheSize;
  info[5] = (int64_t)(prop.totalGlobalMem >> 20);
  if (name && name_cap) snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
  return ZPQ_OK;
}

int zpq_profile_enable(zpq_ctx* ctx, int on) {
  ctx->profiling = on != 0;
  return ZPQ_OK;
}

int zpq_profile_report(zpq_ctx* ctx, char* buf, size_t cap) {
  ZPQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ZPQ_HIP(ctx, hipStreamSynchronize(ctx->stream2));
  struct Acc { const char* name; int count; double ms; };
  std::vector<Acc> acc;
  for (auto& r : ctx->prof) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    bool found = false;
    for (auto& a : acc) if (strcmp(a.name, r.name) == 0) { a.count++; a.ms += ms; found = true; break; }
    if (!found) acc.push_back({r.name, 1, ms});
  }
  ctx->prof.clear();
  std::string out;
  char line[256];
  for (auto& a : acc) { snprintf(line, sizeof line, "%s %d %.6f\n", a.name, a.count, a.ms); out += line; }
  if (buf && cap) { size_t k = out.size() < cap - 1 ? out.size() : cap - 1; memcpy(buf, out.data(), k); buf[k] = 0; }
  return ZPQ_OK;
}

int zpq_dev_alloc(zpq_ctx* ctx, size_t bytes, void** dptr) {
  ZPQ_HIP(ctx, hipSetDevice(ctx->device));
  if (zpq_device_malloc(ctx, dptr, bytes ? bytes : 1) != hipSuccess) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "hipMalloc(%zu)", bytes);
  return ZPQ_OK;
}
// The idle blocks of the context's pool go back to the driver (*freed: their bytes).  Allocations of the engine that fail do this
// themselves, for every context of the device; a host that is about to allocate on its own (torch, another library) calls it.
int zpq_pool_trim(zpq_ctx* ctx, size_t* freed) {
  if (!ctx) return ZPQ_ERR_ARG;
  ZPQ_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n = pool_trim(ctx);
  if (freed) *freed = n;
  return ZPQ_OK;
}
int zpq_dev_free(zpq_ctx* ctx, void* dptr) {
  ZPQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ZPQ_HIP(ctx, hipFree(dptr));
  return ZPQ_OK;
}
// Device memory that goes back to the context instead of the driver: hipFree waits for the whole device and hipMalloc maps
// pages, which a job that runs twelve at a time beside others cannot afford per call.  Blocks up to 1 GiB are kept (at most 6 GiB
// per context, the largest evicted first when that is exceeded) and handed out again to requests they fit without wasting
// more than half; zpq_destroy releases them.  Called by one job at a time per context, like everything else of a context.
int zpq_dev_alloc_pooled(zpq_ctx* ctx, size_t bytes, void** dptr) {
  if (!ctx || !dptr) return ZPQ_ERR_ARG;
  ZPQ_HIP(ctx, hipSetDevice(ctx->device));
  if (!bytes) bytes = 1;
  {
    std::lock_guard<std::mutex> g(ctx->pool_mu);
    size_t best = ctx->pool.size();
    for (size_t i = 0; i < ctx->pool.size(); ++i) {
      const zpq_ctx::PoolBlock& b = ctx->pool[i];
      if (!b.in_use && b.cap >= bytes && b.cap / 2 <= bytes + (1u << 20) && (best == ctx->pool.size() || b.cap < ctx->pool[best].cap)) best = i;
    }
    if (best != ctx->pool.size()) { ctx->pool[best].in_use = true; *dptr = ctx->pool[best].p; return ZPQ_OK; }
  }
  const size_t cap = bytes <= ((size_t)1 << 30) ? ((bytes + bytes / 8 + 65535) & ~(size_t)65535) : bytes;
  // (makes room -- this context's idle blocks first, then the other contexts' -- before it gives up)
  if (zpq_device_malloc(ctx, dptr, cap) != hipSuccess) return zpq_fail(ctx, ZPQ_ERR_NOMEM, "hipMalloc(%zu)", cap);
  if (bytes <= ((size_t)1 << 30)) { std::lock_guard<std::mutex> g(ctx->pool_mu); ctx->pool.push_back({*dptr, cap, true}); }
  return ZPQ_OK;
}
int zpq_dev_free_pooled(zpq_ctx* ctx, void* dptr) {
  if (!ctx) return ZPQ_ERR_ARG;
  if (!dptr) return ZPQ_OK;
  size_t idle = 0;
  bool mine = false;
  std::unique_lock<std::mutex> g(ctx->pool_mu);
  for (zpq_ctx::PoolBlock& b : ctx->pool) {
    if (b.p == dptr) { b.in_use = false; mine = true; }
    if (!b.in_use) idle += b.cap;
  }
  if (!mine) { g.unlock(); return zpq_dev_free(ctx, dptr); }          // larger than the pool keeps
  while (idle > ((size_t)6 << 30)) {                    // over the limit: the largest idle block goes back to the driver
    size_t big = ctx->pool.size();
    for (size_t i = 0; i < ctx->pool.size(); ++i)
      if (!ctx->pool[i].in_use && (big == ctx->pool.size() || ctx->pool[i].cap > ctx->pool[big].cap)) big = i;
    if (big == ctx->pool.size()) break;
    ZPQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    idle -= ctx->pool[big].cap;
    (void)hipFree(ctx->pool[big].p);
    ctx->pool.erase(ctx->pool.begin() + (long)big);
  }
  return ZPQ_OK;
}
int zpq_h2d(zpq_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  ZPQ_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
  ZPQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZPQ_OK;
}
int zpq_d2h(zpq_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  ZPQ_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  ZPQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZPQ_OK;
}
int zpq_copy_peer(zpq_ctx* dst_ctx, void* dst_dev, zpq_ctx* src_ctx, const void* src_dev, size_t bytes) {
  if (!dst_ctx || !src_ctx) return ZPQ_ERR_ARG;
  if (bytes == 0) return ZPQ_OK;
  (void)hipSetDevice(src_ctx->device);
  ZPQ_HIP(src_ctx, hipStreamSynchronize(src_ctx->stream));       // the source bytes are final
  (void)hipSetDevice(dst_ctx->device);
  if (dst_ctx->device == src_ctx->device) {
    ZPQ_HIP(dst_ctx, hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, dst_ctx->stream));
  } else {
    int can = 0;
    (void)hipDeviceCanAccessPeer(&can, dst_ctx->device, src_ctx->device);
    if (can) (void)hipDeviceEnablePeerAccess(src_ctx->device, 0);   // already enabled is not an error worth reporting
    (void)hipGetLastError();
    ZPQ_HIP(dst_ctx, hipMemcpyPeerAsync(dst_dev, dst_ctx->device, src_dev, src_ctx->device, bytes, dst_ctx->stream));
  }
  ZPQ_HIP(dst_ctx, hipStreamSynchronize(dst_ctx->stream));
  return ZPQ_OK;
}
int zpq_dev_memset(zpq_ctx* ctx, void* dst_dev, int value, size_t bytes) {
  if (ctx) (void)hipSetDevice(ctx->device);      // calls may come from any host thread: make the context's device current
  ZPQ_HIP(ctx, hipMemsetAsync(dst_dev, value, bytes, ctx->stream));
  return ZPQ_OK;
}

}  // extern "C"
