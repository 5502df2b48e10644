// Internal declarations shared by the HIP translation units behind include/zpaqhip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>

#include "zpaqhip.h"

// grow-only device scratch arenas; slot numbers are fixed per use (see the zpq_scratch callers):
// 0-11 compress side / hashing, 12-17 the device-resident decode path (unblock.hip), 18-19 checksums, 20-22 E8E9,
// 24 per-block arrays of the suffix-array LZ77 path, 28-29 twin files (twins.hip, fragment.hip), 30 level-2 pack records
#define ZPQ_SCRATCH_SLOTS 32

struct zpq_ctx {
  int device;
  hipStream_t stream;
  hipStream_t stream2;      // second stream: block SHA-1 chains overlap the LZ77 parse
  hipEvent_t ev;
  hipEvent_t ev2;           // main stream -> second stream ordering (work the caller enqueued before a call)
  int cu_count;
  std::string err;
  // grow-only device scratch arena (avoids hipMalloc in the steady state)
  void* scratch[ZPQ_SCRATCH_SLOTS];
  size_t scratch_cap[ZPQ_SCRATCH_SLOTS];
  void* pinned;             // pinned host staging
  size_t pinned_cap;
  // device blocks handed back by zpq_dev_free_pooled, kept for the next zpq_dev_alloc_pooled (ctx.hip)
  struct PoolBlock { void* p; size_t cap; bool in_use; };
  std::vector<PoolBlock> pool;
  std::mutex pool_mu;       // the owner's alloc / free against another context's zpq_device_malloc making room (ctx.hip)
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

// hipMalloc that makes room before it gives up: on failure the idle pool blocks of THIS context, then of every other live
// context of the device, go back to the driver and the allocation is tried again (ctx.hip).  hipSuccess or the last error.
hipError_t zpq_device_malloc(zpq_ctx* ctx, void** p, size_t bytes);
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
typedef int32_t i32;

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

// ---- cooperative wave placement ("one hungry wave per SIMD") ------------------------------------------------------
// With several jobs in flight the long few-wave kernels (13 checksum chains, 212 LZ77 segment waves per job) land on
// SIMDs at the dispatcher's whim: ~550 such waves on 1024 SIMDs, yet a kernel lasts as long as its unluckiest wave --
// the one that shares its SIMD's issue slots with two others (measured: chains 215 -> 313 ms, segment parse 148 -> 250 ms
// with six jobs).  So these kernels are launched with SURPLUS single-wave workgroups over a queue of work items; a wave
// reads where it has landed (HW_ID / XCC_ID), registers in a process-wide table of SIMD loads, and a "polite" wave that
// finds its SIMD taken gives the slot back and exits -- the dispatcher tries another SIMD with the next workgroup.  The
// last n workgroups are not polite: they drain whatever is left, so every item is processed whatever the placement.
struct zpq_place { u32* queue; u32* tab; u32 n; u32 polite; };     // tab == nullptr: plain launch, item = blockIdx.x
u32* zpq_simd_table(zpq_ctx* ctx);                                  // 65536 zeroed counters per device, shared by every context of the process
bool zpq_place_enabled();                                           // several contexts alive and ZPQ_PLACE != 0
static __device__ __forceinline__ u32 zpq_simd_key() {
  const u32 hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);         // HW_ID: simd [5:4], pipe [7:6], cu [11:8], sh [12], se [15:13]
  const u32 xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);        // XCC_ID [3:0]
  return ((xcc & 15u) << 12) | ((hw >> 4) & 0xFFFu);
}
// wave-uniform; every lane executes the atomics (adding 0 except lane 0): nothing lane-dependent next to a loop edge
static __device__ __forceinline__ u32 zpq_wave_add(u32* p, u32 v) {
  const u32 r = atomicAdd(p, (threadIdx.x & 63u) == 0 ? v : 0u);
  return (u32)__builtin_amdgcn_readfirstlane((int)r);
}
// -> first item of this workgroup (0xffffffff: exit now); key/held are the caller's to pass to zpq_place_next / _end
static __device__ __forceinline__ u32 zpq_place_begin(const zpq_place& P, u32& key, bool& polite) {
  key = 0; polite = true;
  if (!P.tab) return blockIdx.x < P.n ? blockIdx.x : 0xffffffffu;
  key = zpq_simd_key();
  polite = blockIdx.x < P.polite;
  const u32 prev = zpq_wave_add(P.tab + key, 1u);
  if (polite && prev != 0) { (void)zpq_wave_add(P.tab + key, 0xffffffffu); return 0xffffffffu; }
  const u32 item = zpq_wave_add(P.queue, 1u);
  if (item >= P.n) { (void)zpq_wave_add(P.tab + key, 0xffffffffu); return 0xffffffffu; }
  return item;
}
// -> next item for a wave that must drain the queue (0xffffffff: done; the table entry has been given back)
static __device__ __forceinline__ u32 zpq_place_next(const zpq_place& P, u32 key, bool polite) {
  if (!P.tab) return 0xffffffffu;
  if (!polite) {
    const u32 item = zpq_wave_add(P.queue, 1u);
    if (item < P.n) return item;
  }
  (void)zpq_wave_add(P.tab + key, 0xffffffffu);
  return 0xffffffffu;
}

// device-side job record of the LZ77 level-1 decoder (lz77_dec.hip); result[0] = out_len, result[1] = status
struct zpq_lzdec_dev {
  const u8* in; u32 n; u32 rb;
  u8* out; u32 out_cap;
  u32* result;
};
// device-side record of one LZ77 block as the token -> bit pack kernels read it (lz77_enc.hip); also the job record of
// the hash-table parse
struct zpq_lzjob_dev {
  const u8* in;
  u32 n;
  u32 rb;
  u32 nseg, seg0;    // segments seg0 .. seg0+nseg-1 in the segment array
  u32* tok_pos; u32* tok_len; u32* tok_off; u32* tok_bit;   // final token list
  u32 tok_cap;
  u32* result;       // [0]=ntok, [1]=out_len bytes, [2]=overflow flag
  u8* out; u32 out_cap;
  u32* plan;         // per segment 2 x {which list (0 spec / 1 seam), from, to, dst}: token ranges to move
};
// tokens -> code bits for nj records (out must be zeroed, result[0] = token count); max_n = longest block
int zpq_lz77_pack_launch(zpq_ctx* ctx, const zpq_lzjob_dev* d_jobs, size_t nj, u32 max_n);
int zpq_lz77_check_args(zpq_ctx* ctx, const int32_t args[9], u32 n);     // ZPQ_OK or the status zpq_lz77_encode_dev would return for these arguments
// the same for level 2 (byte-aligned codes; lz77_sa.hip): HOST records, minMatch per record
int zpq_lz77_pack2_launch(zpq_ctx* ctx, const zpq_lzjob_dev* h_jobs, const u32* min_match, size_t nj, u32 max_n);
// the jobs jobs[which[0..nj)] whose match finder is the suffix array (lz77_sa.hip)
int zpq_lz77_sa_encode(zpq_ctx* ctx, zpq_lz77_job* jobs, const size_t* which, size_t nj);
// decodes every record on `st`; no host round trip for results (h_jobs = host copy of d_jobs, for the scratch layout)
int zpq_lz77_decode_launch(zpq_ctx* ctx, hipStream_t st, const zpq_lzdec_dev* h_jobs, const zpq_lzdec_dev* d_jobs, size_t njobs);
// the 302-byte LZ77 level-1 post-processor program (rb = 0, no E8E9): golden, AUTOTEST/sha256.zpaq i blocks
extern const u8 zpq_pcomp_lz1[302];
// the level-1 post-processor programs decoded natively: rb = 0..7 raw offset bits, with / without the E8E9 inverse
const std::vector<u8>& zpq_known_pcomp(u32 rb, bool e8);
// inverse Burrows-Wheeler transform of a level-3 stream (ibwt.hip); synchronous
int zpq_live_contexts();       // engine contexts alive in this process (several = several jobs in flight)
int zpq_ibwt_dev(zpq_ctx* ctx, const u8* d_bwt, u32 m, u8* d_out, u32 out_cap, u32* out_len);
// decode path for blocks that need host parsing (context-model coded data, arbitrary PCOMP programs): jobs[].in are
// HOST pointers; jobs[].out are device pointers when out_dev, else host pointers (block.hip)
int zpq_decompress_hostparsed(zpq_ctx* ctx, zpq_unblock_job* jobs, size_t njobs, int verify, bool out_dev);

// E8E9 (e8e9.hip): launches without host round trips.  Inverse jobs: device array of zpq_e8inv_job; d_state holds
// 2*nstate+1 words (assumed/left states per 1 KiB segment, and a counter of re-walked segments that must be zeroed).
struct zpq_e8inv_job { const u8* in; u8* out; const u32* len; u32 cap; u32 st_base; };
int zpq_e8e9_inverse_launch(zpq_ctx* ctx, hipStream_t st, const void* d_jobs, size_t njobs, u32 max_cap, u32* d_state, size_t nstate);
int zpq_e8e9_forward_launch(zpq_ctx* ctx, hipStream_t st, u8* d_buf, size_t n, u32* bits);

// internal cross-TU entry points
// method string -> expanded x/0 method, $1..$9, block header bytes (hsize..HCOMP 0) and PCOMP bytecode
// (config.hip; host_data is only read for level 5)
int zpq_build_config(zpq_ctx* ctx, const char* method, const u8* host_data, u32 n, std::string* xmethod, int args[9],
                     std::vector<u8>* header, std::vector<u8>* pcomp);
// ---- context mixing: parsed block header and the run-time specialised coder (cm.hip, cm_jit.hip) -----------------
struct zpq_cm_header {
  u32 hh, hm, ph, pm, n;
  std::vector<std::vector<u8>> comps;   // type + arguments per component
  std::vector<u8> hcomp;                // program bytes (without the closing 0)
};
// header = hsize[2] hh hm ph pm n COMP 0 HCOMP 0 (ZPAQL::read, ZSFX/libzpaq.cpp:879-921); ctx may be NULL
int zpq_cm_parse_header(zpq_ctx* ctx, const u8* h, u32 len, zpq_cm_header& P);
struct zpq_cm_spec;                     // kernels compiled for one header on one device
int zpq_cm_spec_source(const zpq_cm_header& P, std::string* src, std::string* why);
int zpq_cm_spec_get(zpq_ctx* ctx, const zpq_cm_header& P, bool own_config, zpq_cm_spec** out);
u32 zpq_cm_spec_waves(const zpq_cm_spec* k);
int zpq_cm_spec_launch(zpq_ctx* ctx, zpq_cm_spec* k, hipStream_t st, const void* d_jobs, u32 njobs, u32* d_counter, const void* d_tables,
                       int encode);
// any PCOMP program translated to device code and run over d_in (cm_jit.hip); H/M/R: zeroed device arrays; d_result: [0] bytes
// produced, [1] status
int zpq_pcomp_spec_run(zpq_ctx* ctx, hipStream_t st, const u8* pcomp, u32 psize, u32 ph, u32 pm, const u8* d_in, u32 n, u8* d_out, u32 out_cap,
                       u32* d_H, u8* d_M, u32* d_R, u32* d_result, u32* d_seg = nullptr, u32 nseg = 0);
// device-side records of the specialised coder (layout shared with cm_spec_src.inc)
struct zpq_spec_comp { u64 cm, ht; u32 type, a1, a2, a3, a4, a5, limit, cm_mask, ht_mask, csize, pad0, pad1; };
struct zpq_spec_job { u64 comp, p0, H, M, R, in, out, result; u32 in_len, out_cap; u64 prof; u64 seg; u32 nseg, pad; };   // seg: u32[2 * nseg], input bytes per segment then the output end of each (0: one segment)
static_assert(sizeof(zpq_spec_comp) == 64 && sizeof(zpq_spec_job) == 96, "layout shared with the generated kernels");

// twin files (twins.hip): rep[f] = earliest extent with the same bytes (compared), else f; off / len are host arrays
int zpq_twins_find(zpq_ctx* ctx, hipStream_t st, const u8* d_base, const u64* off, const u64* len, size_t n, u64 min_bytes, u32* rep,
                   u64 stats[4]);
// one WAVE per extent (long chains: block checksums); zpq_sha1_extents_on uses one LANE per extent
int zpq_sha1_chains_on(zpq_ctx* ctx, hipStream_t s, const u8* d_base, const u64* d_off, const u32* d_len, size_t n,
                       u8* d_digests, const char* prof_name = "sha1_chain_kernel");
int zpq_sha1_extents_on(zpq_ctx* ctx, hipStream_t s, const u8* d_base, const u64* d_off,
                        const u32* d_len, size_t n, u8* d_digests, const char* prof_name = "sha1_extents_kernel");
