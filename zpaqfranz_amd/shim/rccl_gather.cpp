// rccl_gather.cpp -- an all-gather of byte strings over RCCL (rccl_gather.h): what zpqj_add_sharded[_dev] calls per add.
// Host code; the bytes travel HBM to HBM over xGMI on a stream the communicator owns.  zpqr_allgatherv takes and returns HOST strings
// (fragment tables, block sizes: staged through pinned memory at both ends); zpqr_allgatherv_dev takes and returns DEVICE
// memory (the compressed d blocks of zpqj_add_sharded_dev: no staging at all).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "rccl_gather.h"

struct zpqr_comm {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int device = 0, rank = 0, world = 1;
  void *d_send = nullptr, *d_recv = nullptr, *d_len = nullptr;     // grow-only device staging
  size_t send_cap = 0, recv_cap = 0;
  uint8_t* h_pin = nullptr; size_t pin_cap = 0;                    // pinned host staging of the string this rank sends
  uint8_t* h_out = nullptr; size_t out_cap = 0;                    // pinned: what recv[] points into (the D2H copies land here directly)
  std::string err;
};

namespace {
int fail(zpqr_comm* c, int code, const char* what, const char* detail) {
  if (c) c->err = std::string(what) + ": " + detail;
  return code;
}
#define ZR_HIP(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail((c), ZPQ_ERR_HIP, #call, hipGetErrorString(e_)); } while (0)
#define ZR_NCCL(c, call) do { ncclResult_t e_ = (call); if (e_ != ncclSuccess) return fail((c), ZPQ_ERR_HIP, #call, ncclGetErrorString(e_)); } while (0)
int grow(zpqr_comm* c, void** p, size_t* cap, size_t need) {
  if (need <= *cap) return ZPQ_OK;
  if (*p) { ZR_HIP(c, hipStreamSynchronize(c->stream)); ZR_HIP(c, hipFree(*p)); *p = nullptr; *cap = 0; }
  const size_t n = need + need / 4 + 4096;
  ZR_HIP(c, hipMalloc(p, n));
  *cap = n;
  return ZPQ_OK;
}
int grow_pin(zpqr_comm* c, uint8_t** p, size_t* cap, size_t need) {
  if (need <= *cap && *p) return ZPQ_OK;
  if (*p) { ZR_HIP(c, hipStreamSynchronize(c->stream)); ZR_HIP(c, hipHostFree(*p)); *p = nullptr; *cap = 0; }
  const size_t n = need + need / 4 + 4096;
  ZR_HIP(c, hipHostMalloc((void**)p, n, hipHostMallocDefault));
  *cap = n;
  return ZPQ_OK;
}
}  // namespace

extern "C" {

int zpqr_unique_id(uint8_t id[ZPQR_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) == ZPQR_ID_BYTES, "ncclUniqueId is 128 bytes");
  if (!id) return ZPQ_ERR_ARG;
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return ZPQ_ERR_HIP;
  memcpy(id, &u, sizeof u);
  return ZPQ_OK;
}

int zpqr_create(zpq_ctx* ctx, int rank, int world, const uint8_t id[ZPQR_ID_BYTES], zpqr_comm** out) {
  if (!out) return ZPQ_ERR_ARG;
  *out = nullptr;
  if (!ctx || !id || world < 1 || world > ZPQR_MAX_RANKS || rank < 0 || rank >= world) return ZPQ_ERR_ARG;
  zpqr_comm* c = new zpqr_comm;
  c->rank = rank; c->world = world;
  // A stream of its own (round 6).  Every call here is synchronous for its caller -- what is sent is complete before the call
  // (zpqj_add_sharded[_dev] synchronises its context first), what is received is complete when it returns -- so nothing has to
  // be ordered with the context's stream; on that stream every synchronisation below waited for whatever kernels the context's
  // CURRENT add had in flight, and with several adds in flight sharing one communicator every collective of every add queued
  // behind one context's fragment pass: 272 ms per step through the product call at world size 1 against 99 ms without
  // collectives (profiles/r06d_bench.json / bench_rccl1).
  int rc = ZPQ_OK;
  do {
    if (hipStreamGetDevice((hipStream_t)zpq_stream(ctx), &c->device) != hipSuccess || hipSetDevice(c->device) != hipSuccess) { rc = ZPQ_ERR_HIP; break; }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { c->stream = nullptr; rc = ZPQ_ERR_HIP; break; }
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    if (ncclCommInitRank(&c->comm, world, u, rank) != ncclSuccess) { rc = ZPQ_ERR_HIP; break; }
    if (hipMalloc(&c->d_len, 8 * (size_t)(world + 1)) != hipSuccess) { rc = ZPQ_ERR_NOMEM; break; }
  } while (0);
  if (rc) { zpqr_destroy(c); return rc; }
  *out = c;
  return ZPQ_OK;
}

namespace {
// a rank that cannot go on (out of memory between the two collectives of a call) must not leave the others waiting in the
// second one for ever: the communicator is aborted, which fails the peers' pending and later calls
int give_up(zpqr_comm* c, int rc) {
  if (c->comm) { (void)ncclCommAbort(c->comm); c->comm = nullptr; }
  return rc;
}
// 1. everybody's length; returns the longest and the total
int exchange_lengths(zpqr_comm* c, size_t send_len, unsigned long long* lens, size_t* mx, size_t* total) {
  const size_t W = (size_t)c->world;
  if (!c->comm) return fail(c, ZPQ_ERR_HIP, "zpqr_allgatherv", "the communicator was aborted by an earlier failure");
  unsigned long long mine = send_len;
  uint64_t* d_len = (uint64_t*)c->d_len;
  ZR_HIP(c, hipMemcpyAsync(d_len + W, &mine, 8, hipMemcpyHostToDevice, c->stream));
  ZR_NCCL(c, ncclAllGather(d_len + W, d_len, 8, ncclChar, c->comm, c->stream));
  ZR_HIP(c, hipMemcpyAsync(lens, d_len, 8 * W, hipMemcpyDeviceToHost, c->stream));
  ZR_HIP(c, hipStreamSynchronize(c->stream));
  *mx = 0; *total = 0;
  for (size_t r = 0; r < W; ++r) { *mx = lens[r] > *mx ? (size_t)lens[r] : *mx; *total += (size_t)lens[r]; }
  return ZPQ_OK;
}
// 2. the strings with their exact lengths: a grouped send / receive per pair of ranks (xGMI is point to point: every pair has
// its own link), rank r's string at d_recv + off[r]; a rank's own string is a device copy
int exchange_exact(zpqr_comm* c, const void* d_send, const unsigned long long* lens, const size_t* off) {
  const size_t W = (size_t)c->world, me = (size_t)c->rank;
  if (lens[me]) ZR_HIP(c, hipMemcpyAsync((uint8_t*)c->d_recv + off[me], d_send, (size_t)lens[me], hipMemcpyDeviceToDevice, c->stream));
  if (W > 1) {
    ZR_NCCL(c, ncclGroupStart());
    for (size_t r = 0; r < W; ++r) {
      if (r == me) continue;
      if (lens[me]) ZR_NCCL(c, ncclSend(d_send, (size_t)lens[me], ncclChar, (int)r, c->comm, c->stream));
      if (lens[r]) ZR_NCCL(c, ncclRecv((uint8_t*)c->d_recv + off[r], (size_t)lens[r], ncclChar, (int)r, c->comm, c->stream));
    }
    ZR_NCCL(c, ncclGroupEnd());
  }
  return ZPQ_OK;
}
}  // namespace

int zpqr_allgatherv(void* user, const void* send, size_t send_len, void** recv, size_t* recv_len) {
  zpqr_comm* c = (zpqr_comm*)user;
  if (!c || !recv || !recv_len || (send_len && !send)) return ZPQ_ERR_ARG;
  const size_t W = (size_t)c->world;
  ZR_HIP(c, hipSetDevice(c->device));
  unsigned long long lens[ZPQR_MAX_RANKS];
  size_t mx = 0, total = 0;
  int rc;
  if ((rc = exchange_lengths(c, send_len, lens, &mx, &total))) return rc;
  if (mx) {
    // strings of about equal length: one ncclAllGather of slots padded to the longest (ring over xGMI: every link carries
    // (W-1)/W of the padded total); lengths that differ by more than 2x (one rank holds all the d blocks, say): exact sends
    const bool exact = mx * W > 2 * total;
    const size_t slot = (mx + 15) & ~(size_t)15;
    size_t off[ZPQR_MAX_RANKS], at = 0;
    for (size_t r = 0; r < W; ++r) { off[r] = exact ? at : r * slot; at += ((size_t)lens[r] + 15) & ~(size_t)15; }
    if ((rc = grow(c, &c->d_send, &c->send_cap, slot)) || (rc = grow(c, &c->d_recv, &c->recv_cap, exact ? at : slot * W)) ||
        (rc = grow_pin(c, &c->h_pin, &c->pin_cap, slot)) || (rc = grow_pin(c, &c->h_out, &c->out_cap, total)))
      return give_up(c, rc);
    if (send_len) {
      memcpy(c->h_pin, send, send_len);
      ZR_HIP(c, hipMemcpyAsync(c->d_send, c->h_pin, send_len, hipMemcpyHostToDevice, c->stream));
    }
    if (exact) { if ((rc = exchange_exact(c, c->d_send, lens, off))) return rc; }
    else ZR_NCCL(c, ncclAllGather(c->d_send, c->d_recv, slot, ncclChar, c->comm, c->stream));
    // only the bytes that are strings come back, rank by rank, straight into the pinned buffer recv[] points into
    at = 0;
    for (size_t r = 0; r < W; ++r) {
      if (lens[r]) ZR_HIP(c, hipMemcpyAsync(c->h_out + at, (const uint8_t*)c->d_recv + off[r], (size_t)lens[r], hipMemcpyDeviceToHost, c->stream));
      at += (size_t)lens[r];
    }
    ZR_HIP(c, hipStreamSynchronize(c->stream));
  }
  size_t at = 0;
  static uint8_t nothing;
  for (size_t r = 0; r < W; ++r) { recv[r] = c->h_out ? (void*)(c->h_out + at) : (void*)&nothing; recv_len[r] = (size_t)lens[r]; at += (size_t)lens[r]; }
  return ZPQ_OK;
}

// The same over DEVICE memory (jidac_gpu.h: zpqj_allgatherv_dev_fn): d_send in HBM, d_recv[r] point into an HBM buffer the
// communicator owns until its next call.  No host staging at either end: HBM -> xGMI -> HBM, exact lengths.
int zpqr_allgatherv_dev(void* user, const void* d_send, size_t send_len, void** d_recv, size_t* recv_len) {
  zpqr_comm* c = (zpqr_comm*)user;
  if (!c || !d_recv || !recv_len || (send_len && !d_send)) return ZPQ_ERR_ARG;
  const size_t W = (size_t)c->world;
  ZR_HIP(c, hipSetDevice(c->device));
  unsigned long long lens[ZPQR_MAX_RANKS];
  size_t mx = 0, total = 0;
  int rc;
  if ((rc = exchange_lengths(c, send_len, lens, &mx, &total))) return rc;
  size_t off[ZPQR_MAX_RANKS], at = 0;
  for (size_t r = 0; r < W; ++r) { off[r] = at; at += ((size_t)lens[r] + 15) & ~(size_t)15; }
  if (mx) {
    if ((rc = grow(c, &c->d_recv, &c->recv_cap, at))) return give_up(c, rc);
    if ((rc = exchange_exact(c, d_send, lens, off))) return rc;
    ZR_HIP(c, hipStreamSynchronize(c->stream));
  }
  for (size_t r = 0; r < W; ++r) { d_recv[r] = lens[r] ? (void*)((uint8_t*)c->d_recv + off[r]) : nullptr; recv_len[r] = (size_t)lens[r]; }
  return ZPQ_OK;
}

const char* zpqr_last_error(const zpqr_comm* c) { return c ? c->err.c_str() : ""; }

void zpqr_destroy(zpqr_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  if (c->d_send) (void)hipFree(c->d_send);
  if (c->d_recv) (void)hipFree(c->d_recv);
  if (c->d_len) (void)hipFree(c->d_len);
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  if (c->h_out) (void)hipHostFree(c->h_out);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

}  // extern "C"
