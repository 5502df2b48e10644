/* jidac_gpu.h -- journaling archive add/extract on the MI355X engine (see jidac_gpu.cpp).  C linkage so
 * that any host can call it; memory returned through out-pointers is malloc'd: release with zpqj_free. */
#ifndef JIDAC_GPU_H
#define JIDAC_GPU_H
#include <stddef.h>
#include <stdint.h>
#include "zpaqhip.h"
#ifdef __cplusplus
extern "C" {
#endif
void zpqj_free(void* p);
int zpqj_add(zpq_ctx* ctx, const uint8_t* archive, size_t archive_len, const char* const* names,
             const uint8_t* const* datas, const uint64_t* sizes, const int64_t* dates, size_t nfiles,
             int64_t version_date, const char* method, uint8_t** out, size_t* out_len, uint64_t stats[6]);
/* zpqj_add over several GPUs of one node, one context each (files sharded by contiguous ranges in name order, global
 * dedup, blocks compressed by the GPU that holds their first fragment, other fragments fetched peer to peer).  The
 * bytes returned do not depend on nctx. */
int zpqj_add_multi(zpq_ctx* const* ctxs, size_t nctx, const uint8_t* archive, size_t archive_len, const char* const* names,
                   const uint8_t* const* datas, const uint64_t* sizes, const int64_t* dates, size_t nfiles,
                   int64_t version_date, const char* method, uint8_t** out, size_t* out_len, uint64_t stats[6]);
/* zpqj_add_multi with options */
#define ZPQJ_FILE_CHECKSUMS 1u   /* store XXHASH64 + CRC-32 of every file in its i-block attribute (zpaqfranz's default) */
#define ZPQJ_METHOD_HINT 2u      /* method "LB" (digits only): every d block gets "LB,R,t" from its fragments' statistics
                                  * (zpq_fragment_stats_dev), as zpaq's add() does; the detectors are unpinned */
#define ZPQJ_NO_TWINS 4u         /* every file through the fragment loop and SHA-1, also one whose bytes equal an earlier file's
                                  * (default: such files are found by comparing every byte on the device, csrc/twins.hip) */
int zpqj_add_opts(zpq_ctx* const* ctxs, size_t nctx, const uint8_t* archive, size_t archive_len, const char* const* names,
                  const uint8_t* const* datas, const uint64_t* sizes, const int64_t* dates, size_t nfiles,
                  int64_t version_date, const char* method, uint32_t flags, uint8_t** out, size_t* out_len, uint64_t stats[6]);
/* zpqj_add_opts on one context with the files ALREADY IN HBM: `names` ascend (strcmp) and file k lies at d_base + file_off[k] ..
 * file_off[k + 1], back to back in that order (ZPQ_ERR_ARG otherwise); d_base is 16-byte aligned with 64 readable bytes behind
 * the last file.  The archive comes back in host memory (zpqj_free).  Jidac::add's loop over files -> fragments -> blocks
 * (the missing zpaqfranz.cpp; its product is read back by ZSFX/zsfx.cpp:1384-1542) as one call that never moves the input. */
int zpqj_add_dev(zpq_ctx* ctx, const uint8_t* archive, size_t archive_len, const char* const* names, const uint8_t* d_base,
                 const uint64_t* file_off, const int64_t* dates, size_t nfiles, int64_t version_date, const char* method,
                 uint32_t flags, uint8_t** out, size_t* out_len, uint64_t stats[6]);
/* The add across processes, one GPU each.  The caller supplies ONE collective: an all-gather of byte strings over its
 * ranks (MPI_Allgatherv, torch.distributed.all_gather_object, RCCL all_gather on padded buffers ...).  It must fill
 * recv[r] / recv_len[r] for every rank r (memory it owns, valid until its next call or the return of zpqj_add_sharded;
 * recv[rank] may repeat `send`) and return 0.  zpqj_add_sharded calls it exactly three times on every rank: fragment
 * tables (28 bytes per fragment), the fragments that a d block takes across a range edge (at most one block per edge),
 * the compressed d blocks.  Every rank passes the same archive / names / sizes / dates / method / flags and the data of
 * the files zpqj_shard_files marks for it (other datas[k] are not read) and gets back the same bytes: those zpqj_add
 * returns for the whole batch on one GPU. */
typedef int (*zpqj_allgatherv_fn)(void* user, const void* send, size_t send_len, void** recv, size_t* recv_len);
int zpqj_add_sharded(zpq_ctx* ctx, int rank, int world, zpqj_allgatherv_fn allgatherv, void* user, const uint8_t* archive,
                     size_t archive_len, const char* const* names, const uint8_t* const* datas, const uint64_t* sizes,
                     const int64_t* dates, size_t nfiles, int64_t version_date, const char* method, uint32_t flags,
                     uint8_t** out, size_t* out_len, uint64_t stats[6]);
/* zpqj_add_sharded with this rank's files ALREADY IN HBM (the sharded counterpart of zpqj_add_dev): `names` ascend strictly (strcmp)
 * and `sizes` holds every file's size -- the same lists on every rank --, the files zpqj_shard_files marks for this rank lie back
 * to back in that order at d_base (16-byte aligned, 64 readable bytes behind the last file).  allgatherv_dev (may be NULL) is the
 * same collective over DEVICE memory: d_send / d_recv[r] are HBM pointers (d_recv[r] valid until its next call).  It carries the one
 * exchange that is large, the compressed d blocks, HBM -> collective -> HBM; the other exchanges (fragment tables, the sizes of
 * the blocks) stay host strings.  shim/rccl_gather.h has both collectives over RCCL. */
typedef int (*zpqj_allgatherv_dev_fn)(void* user, const void* d_send, size_t send_len, void** d_recv, size_t* recv_len);
int zpqj_add_sharded_dev(zpq_ctx* ctx, int rank, int world, zpqj_allgatherv_fn allgatherv, zpqj_allgatherv_dev_fn allgatherv_dev, void* user,
                         const uint8_t* archive, size_t archive_len, const char* const* names, const uint8_t* d_base, const uint64_t* sizes,
                         const int64_t* dates, size_t nfiles, int64_t version_date, const char* method, uint32_t flags, uint8_t** out,
                         size_t* out_len, uint64_t stats[6]);
int zpqj_shard_files(const char* const* names, const uint64_t* sizes, size_t nfiles, int world, int rank, uint8_t* mine);
int zpqj_extract(zpq_ctx* ctx, const uint8_t* archive, size_t archive_len, uint8_t** data, uint64_t** sizes,
                 char** names, size_t* nfiles);
/* Jidac::extract (ZSFX/zsfx.cpp:2018-2281; decompressThread :1731-1994; read_archive's jump over the d blocks :1432-1461) with the
 * archive ALREADY IN HBM (d_archive, readable for 64 bytes behind archive_len) and the restored files LEFT IN HBM: back to back
 * in name order in d_out (out_cap >= file_off[nfiles] + 64).  *file_off (nfiles + 1 entries) and *names (NUL separated) come back
 * malloc'd (zpqj_free).  ZPQJ_X_SHA256: every restored file's SHA-256 into d_sha256 (32 bytes per file, room for
 * sha256_cap_files); with ZPQJ_X_TWINS a file whose bytes equal an earlier restored file's (every byte compared on the device)
 * takes that file's digest.  d_out == NULL: plan only (index read, sizes returned, nothing decoded).
 * stats[0..6] = files, fragments checked, bytes restored, twin files, twin bytes, bytes compared (ZPQJ_X_TWINS), d blocks. */
#define ZPQJ_X_SHA256 1u
#define ZPQJ_X_TWINS 2u
int zpqj_extract_dev(zpq_ctx* ctx, const uint8_t* d_archive, size_t archive_len, uint8_t* d_out, size_t out_cap, uint8_t* d_sha256,
                     size_t sha256_cap_files, uint32_t flags, uint64_t** file_off, char** names, size_t* nfiles, uint64_t stats[7]);
/* zpaqfranz t: decode + verify everything on the device (block SHA-1s, fragment SHA-1s against the h table, and the
 * per-file XXHASH64 / CRC-32 the i blocks carry, where they do); nothing but stats[0..6] comes back: files, fragments
 * checked, bytes restored, files with stored checksums, XXHASH64 mismatches, CRC-32 mismatches, d blocks. */
int zpqj_verify(zpq_ctx* ctx, const uint8_t* archive, size_t archive_len, uint64_t stats[7]);
#ifdef __cplusplus
}
#endif
#endif
