// rccl_gather.cpp -- an all-gather of byte strings over RCCL (rccl_gather.h): what zpqj_add_sharded calls three times per add.
// Host code; the bytes travel HBM to HBM over xGMI (ncclAllGather on the context's stream), staged through pinned memory
// at both ends because the caller's strings (fragment tables, d blocks) are host data, as in the reference.
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
  uint8_t* h_pin = nullptr; size_t pin_cap = 0;                    // pinned host staging (send and receive side)
  std::vector<uint8_t> host;                                       // what recv[] points into
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
int grow_pin(zpqr_comm* c, size_t need) {
  if (need <= c->pin_cap) return ZPQ_OK;
  if (c->h_pin) { ZR_HIP(c, hipStreamSynchronize(c->stream)); ZR_HIP(c, hipHostFree(c->h_pin)); c->h_pin = nullptr; c->pin_cap = 0; }
  const size_t n = need + need / 4 + 4096;
  ZR_HIP(c, hipHostMalloc((void**)&c->h_pin, n, hipHostMallocDefault));
  c->pin_cap = n;
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
  if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return ZPQ_ERR_ARG;
  zpqr_comm* c = new zpqr_comm;
  c->rank = rank; c->world = world;
  c->stream = (hipStream_t)zpq_stream(ctx);                        // the collectives are ordered with the context's own work
  int rc = ZPQ_OK;
  do {
    if (hipStreamGetDevice(c->stream, &c->device) != hipSuccess || hipSetDevice(c->device) != hipSuccess) { rc = ZPQ_ERR_HIP; break; }
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    if (ncclCommInitRank(&c->comm, world, u, rank) != ncclSuccess) { rc = ZPQ_ERR_HIP; break; }
    if (hipMalloc(&c->d_len, 8 * (size_t)(world + 1)) != hipSuccess) { rc = ZPQ_ERR_NOMEM; break; }
  } while (0);
  if (rc) { zpqr_destroy(c); return rc; }
  *out = c;
  return ZPQ_OK;
}

int zpqr_allgatherv(void* user, const void* send, size_t send_len, void** recv, size_t* recv_len) {
  zpqr_comm* c = (zpqr_comm*)user;
  if (!c || !recv || !recv_len || (send_len && !send)) return ZPQ_ERR_ARG;
  const size_t W = (size_t)c->world;
  ZR_HIP(c, hipSetDevice(c->device));
  // 1. everybody's length
  unsigned long long mine = send_len, lens[1024];
  if (W > 1024) return fail(c, ZPQ_ERR_ARG, "zpqr_allgatherv", "more than 1024 ranks");
  uint64_t* d_len = (uint64_t*)c->d_len;
  ZR_HIP(c, hipMemcpyAsync(d_len + W, &mine, 8, hipMemcpyHostToDevice, c->stream));
  ZR_NCCL(c, ncclAllGather(d_len + W, d_len, 8, ncclChar, c->comm, c->stream));
  ZR_HIP(c, hipMemcpyAsync(lens, d_len, 8 * W, hipMemcpyDeviceToHost, c->stream));
  ZR_HIP(c, hipStreamSynchronize(c->stream));
  size_t mx = 0, total = 0;
  for (size_t r = 0; r < W; ++r) { mx = lens[r] > mx ? (size_t)lens[r] : mx; total += (size_t)lens[r]; }
  c->host.resize(total ? total : 1);
  if (mx) {
    // 2. the strings, padded to the longest: one all-gather (ring over xGMI: every link carries (W-1)/W of the padded total)
    const size_t slot = (mx + 15) & ~(size_t)15;
    int rc;
    if ((rc = grow(c, &c->d_send, &c->send_cap, slot)) || (rc = grow(c, &c->d_recv, &c->recv_cap, slot * W)) || (rc = grow_pin(c, slot * W))) return rc;
    if (send_len) {
      memcpy(c->h_pin, send, send_len);
      ZR_HIP(c, hipMemcpyAsync(c->d_send, c->h_pin, send_len, hipMemcpyHostToDevice, c->stream));
    }
    ZR_NCCL(c, ncclAllGather(c->d_send, c->d_recv, slot, ncclChar, c->comm, c->stream));
    ZR_HIP(c, hipStreamSynchronize(c->stream));                    // (h_pin is reused for the way back)
    // only the bytes that are strings come back, rank by rank
    size_t at = 0;
    for (size_t r = 0; r < W; ++r) {
      if (lens[r]) ZR_HIP(c, hipMemcpyAsync(c->h_pin + at, (const uint8_t*)c->d_recv + r * slot, (size_t)lens[r], hipMemcpyDeviceToHost, c->stream));
      at += (size_t)lens[r];
    }
    ZR_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(c->host.data(), c->h_pin, total);
  }
  size_t at = 0;
  for (size_t r = 0; r < W; ++r) { recv[r] = c->host.data() + at; recv_len[r] = (size_t)lens[r]; at += (size_t)lens[r]; }
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
  delete c;
}

}  // extern "C"
