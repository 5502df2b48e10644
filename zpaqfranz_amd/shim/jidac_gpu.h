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
int zpqj_add_opts(zpq_ctx* const* ctxs, size_t nctx, const uint8_t* archive, size_t archive_len, const char* const* names,
                  const uint8_t* const* datas, const uint64_t* sizes, const int64_t* dates, size_t nfiles,
                  int64_t version_date, const char* method, uint32_t flags, uint8_t** out, size_t* out_len, uint64_t stats[6]);
int zpqj_extract(zpq_ctx* ctx, const uint8_t* archive, size_t archive_len, uint8_t** data, uint64_t** sizes,
                 char** names, size_t* nfiles);
/* zpaqfranz t: decode + verify everything on the device (block SHA-1s, fragment SHA-1s against the h table, and the
 * per-file XXHASH64 / CRC-32 the i blocks carry, where they do); nothing but stats[0..6] comes back: files, fragments
 * checked, bytes restored, files with stored checksums, XXHASH64 mismatches, CRC-32 mismatches, d blocks. */
int zpqj_verify(zpq_ctx* ctx, const uint8_t* archive, size_t archive_len, uint64_t stats[7]);
#ifdef __cplusplus
}
#endif
#endif
