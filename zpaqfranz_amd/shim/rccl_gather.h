/* rccl_gather.h -- the ONE collective zpqj_add_sharded needs (jidac_gpu.h: zpqj_allgatherv_fn), over RCCL: an all-gather of byte
 * strings between the ranks of a node (one process per GPU, xGMI underneath).  No torch, no MPI: plain rccl.h.
 *
 *   rank 0:            zpqr_unique_id(id);   ... hand the 128 bytes to every rank (a file, a socket, the launcher's environment) ...
 *   every rank:        zpqr_create(ctx, rank, world, id, &comm);
 *                      zpqj_add_sharded(ctx, rank, world, zpqr_allgatherv, comm, ...);
 *                      zpqr_destroy(comm);
 *
 * zpqr_allgatherv: lengths first (one ncclAllGather of 8 bytes per rank), then the strings -- padded to the longest in one
 * ncclAllGather, or, when the lengths differ by more than 2x, with their exact lengths by grouped ncclSend / ncclRecv; recv[r]
 * points into a pinned host buffer the communicator owns until its next call.  A rank that fails between the two collectives
 * aborts the communicator (its peers' calls fail instead of waiting for it).  Counterpart in the reference:
 * none -- Jidac::add is one process; this is the exchange step of DESIGN.md section 6 (fragment tables, seam fragments, d blocks). */
#ifndef ZPQ_RCCL_GATHER_H
#define ZPQ_RCCL_GATHER_H
#include <stddef.h>
#include <stdint.h>
#include "zpaqhip.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct zpqr_comm zpqr_comm;
#define ZPQR_ID_BYTES 128
#define ZPQR_MAX_RANKS 1024      /* zpqr_create refuses a larger world */
int zpqr_unique_id(uint8_t id[ZPQR_ID_BYTES]);
int zpqr_create(zpq_ctx* ctx, int rank, int world, const uint8_t id[ZPQR_ID_BYTES], zpqr_comm** out);
/* matches zpqj_allgatherv_fn with user = the zpqr_comm*; returns 0 or a negative ZPQ_ERR_* */
int zpqr_allgatherv(void* comm, const void* send, size_t send_len, void** recv, size_t* recv_len);
/* matches zpqj_allgatherv_dev_fn: d_send and d_recv[r] are DEVICE pointers (d_recv[r] into a buffer the communicator owns until its
 * next call); exact lengths by grouped ncclSend / ncclRecv, no host staging */
int zpqr_allgatherv_dev(void* comm, const void* d_send, size_t send_len, void** d_recv, size_t* recv_len);
const char* zpqr_last_error(const zpqr_comm* comm);
void zpqr_destroy(zpqr_comm* comm);
#ifdef __cplusplus
}
#endif
#endif
