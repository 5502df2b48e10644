/* zpaqhip.h -- C ABI of the MI355X (gfx950) engine for zpaqfranz's block compress/decompress
 * hot path.  Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * What it replaces (paths relative to the reference tree /root/reference):
 *   - the fragment loop of Jidac::add (rolling hash + SHA-1 per fragment; source file
 *     zpaqfranz.cpp is absent from the snapshot, algorithm in SURVEY.md Appendix C.4, the
 *     records it produces are parsed back at ZSFX/zsfx.cpp:1463-1500)      -> zpq_fragment_*
 *   - libzpaq::SHA1 / SHA256 (ZSFX/libzpaq.h:934-979, ZSFX/libzpaq.cpp:96-304)   -> zpq_sha1_*, zpq_sha256_*
 *   - the dedup index over HT{sha1,usize} (ZSFX/zsfx.cpp:651-659)                 -> zpq_dedup_*
 *   - libzpaq::compressBlock (ZSFX/libzpaq.h:1505) for the stored / LZ77 level-1 family that
 *     "-m0" and "-m1" expand to: LZBuffer (ZSFX/libzpaq.cpp:6140-6552), Encoder stored mode and
 *     Compressor framing (ZSFX/libzpaq.h:1273-1286,1340-1371)                     -> zpq_lz77_*, zpq_compress_blocks*
 *   - libzpaq::Decompresser for the same family (ZSFX/libzpaq.cpp:2239-2366; the LZ77 inverse is
 *     the 302-byte PCOMP program run by PostProcessor, :2178-2233)                -> zpq_decompress_blocks*
 *
 * Conventions
 *   - Every function returns ZPQ_OK (0) or a negative zpq_status; nothing throws or longjmps
 *     across the ABI.  zpq_last_error(ctx) gives a human-readable detail string.
 *   - "_dev" entry points take DEVICE pointers (HBM-resident data, the fast path); the others
 *     take host pointers and stage through the context's stream.
 *   - All work is enqueued on the context's HIP stream; functions that return results to the
 *     host synchronise that stream before returning.
 *   - A context is bound to one GPU and must be used by one host thread at a time (the
 *     reference's rule: one Compressor/Decompresser per thread, ZSFX/libzpaq.h:57-59).
 *   - There is NO CPU fallback: if no gfx950 device is present zpq_create fails with
 *     ZPQ_ERR_NO_DEVICE.
 */
#ifndef ZPAQHIP_H
#define ZPAQHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum zpq_status {
  ZPQ_OK = 0,
  ZPQ_ERR_NO_DEVICE = -1,   /* no HIP device / not gfx950 */
  ZPQ_ERR_HIP = -2,         /* a HIP runtime call failed (see zpq_last_error) */
  ZPQ_ERR_ARG = -3,         /* invalid argument */
  ZPQ_ERR_CAPACITY = -4,    /* caller-provided output capacity too small */
  ZPQ_ERR_METHOD = -5,      /* method string outside the family this engine implements */
  ZPQ_ERR_FORMAT = -6,      /* malformed ZPAQ block on the decode side */
  ZPQ_ERR_CHECKSUM = -7,    /* stored SHA-1 does not match the decoded data */
  ZPQ_ERR_NOMEM = -8,       /* device or host allocation failed */
  ZPQ_ERR_LIMIT = -9        /* a ZPAQL program of the block (HCOMP / PCOMP) used up the engine's loop budget: a hard refusal of a
                             * block that may be valid -- the reference's interpreter has no limit (ZSFX/libzpaq.cpp:1033-1254);
                             * 2^24 backward jumps per byte + 2^28 once per block (HCOMP), 2^22 + 64 per byte (PCOMP) */
} zpq_status;

typedef struct zpq_ctx zpq_ctx;

/* ---- context ------------------------------------------------------------------------------ */
int zpq_create(int device_ordinal, zpq_ctx** out);
void zpq_destroy(zpq_ctx* ctx);
const char* zpq_strerror(int status);
const char* zpq_last_error(const zpq_ctx* ctx);
int zpq_sync(zpq_ctx* ctx);
/* The context's hipStream_t (as void*), so a host that owns device buffers (e.g. torch) can
 * order its own work against the engine's. */
void* zpq_stream(zpq_ctx* ctx);
/* Device properties the benchmark reports: [0]=CU count, [1]=max clock kHz, [2]=memory clock
 * kHz, [3]=memory bus width bits, [4]=L2 bytes, [5]=total HBM bytes (low 32 bits in MiB). */
int zpq_device_info(zpq_ctx* ctx, int64_t info[6], char* name, size_t name_cap);

/* ---- per-kernel timing (HIP events on the launch stream) ---------------------------------- */
/* When enabled every kernel launch is bracketed by hipEvents on the stream it is launched on.
 * zpq_profile_report synchronises, writes one line per kernel "name count total_ms" into buf
 * (NUL-terminated, truncated to cap) and clears the records. */
int zpq_profile_enable(zpq_ctx* ctx, int on);
int zpq_profile_report(zpq_ctx* ctx, char* buf, size_t cap);

/* ---- device memory helpers (for hosts without their own allocator) ------------------------ */
int zpq_dev_alloc(zpq_ctx* ctx, size_t bytes, void** dptr);
int zpq_dev_free(zpq_ctx* ctx, void* dptr);
/* The same, but the block goes back to the CONTEXT and is handed out again by the next call it fits (blocks up to 1 GiB,
 * at most 6 GiB idle per context; zpq_destroy releases them): hipFree waits for the whole device, which a job running
 * beside eleven others cannot afford per call -- the reference's counterpart is the StringBuffer a compress job keeps
 * (ZSFX/libzpaq.h:1377-1494).  One job at a time per context, like every call on a context. */
int zpq_dev_alloc_pooled(zpq_ctx* ctx, size_t bytes, void** dptr);
int zpq_dev_free_pooled(zpq_ctx* ctx, void* dptr);
/* Gives the context's idle pooled blocks back to the driver (*freed, optional: their bytes).  Every device allocation of the
 * engine that fails does this by itself -- for the calling context, then for every other live context of the device -- and
 * tries again before it reports ZPQ_ERR_NOMEM; a host about to allocate on its own calls it. */
int zpq_pool_trim(zpq_ctx* ctx, size_t* freed);
int zpq_h2d(zpq_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int zpq_d2h(zpq_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int zpq_dev_memset(zpq_ctx* ctx, void* dst_dev, int value, size_t bytes);
/* Device-to-device copy between two contexts (possibly on different GPUs of the node: peer to peer over xGMI when
 * the devices can reach each other, staged by the runtime otherwise).  Ordered on dst_ctx's stream; returns when
 * the bytes have arrived. */
int zpq_copy_peer(zpq_ctx* dst_ctx, void* dst_dev, zpq_ctx* src_ctx, const void* src_dev, size_t bytes);

/* ---- SHA-1 / SHA-256 over many extents (rows a2, a18) ------------------------------------- */
/* digest[i] = SHA-1(base[off[i] .. off[i]+len[i])), 20 bytes each (32 for SHA-256).
 * Every input buffer handed to this library (here and below) must stay readable for 64 bytes past
 * its last byte (pad the allocation): the kernels use 8/16-byte vector reads. */
int zpq_sha1_extents_dev(zpq_ctx* ctx, const uint8_t* d_base, const uint64_t* d_off,
                         const uint32_t* d_len, size_t n, uint8_t* d_digests);
int zpq_sha256_extents_dev(zpq_ctx* ctx, const uint8_t* d_base, const uint64_t* d_off,
                           const uint64_t* d_len, size_t n, uint8_t* d_digests);
/* Host convenience: n independent host buffers -> n digests. */
int zpq_sha1_many(zpq_ctx* ctx, const uint8_t* const* bufs, const size_t* lens, size_t n,
                  uint8_t* digests /* n*20 */);
int zpq_sha256_many(zpq_ctx* ctx, const uint8_t* const* bufs, const size_t* lens, size_t n,
                    uint8_t* digests /* n*32 */);

/* ---- content-defined fragmenter (row a1) --------------------------------------------------- */
typedef struct zpq_fragment_params {
  uint32_t fragment_log2;  /* zpaqfranz -fragment N, default 6: cut when hash < 2^(22-N) */
  uint32_t min_fragment;   /* default 64<<N   = 4096   */
  uint32_t max_fragment;   /* default 8128<<N = 520192 (capped at blocksize-12 by the caller) */
} zpq_fragment_params;
void zpq_fragment_params_default(zpq_fragment_params* p);

/* Files are the extents [file_off[f], file_off[f+1]) of one device buffer (file_off has
 * nfiles+1 entries, HOST pointer).  Produces, in file order then position order, one record per
 * fragment: absolute offset into d_base, length, owning file.  The three output arrays are
 * DEVICE pointers with room for frag_cap records; *nfrags receives the count (host).
 * frag_cap >= zpq_fragment_capacity(...) always suffices. */
size_t zpq_fragment_capacity(const uint64_t* file_off, size_t nfiles, const zpq_fragment_params* p);
int zpq_fragment_dev(zpq_ctx* ctx, const uint8_t* d_base, const uint64_t* file_off, size_t nfiles,
                     const zpq_fragment_params* p, uint64_t* d_frag_off, uint32_t* d_frag_len,
                     uint32_t* d_frag_file, size_t frag_cap, size_t* nfrags);

/* ---- twin files: whole-file duplicates found by comparison (rows a1-a3) -------------- */
/* Jidac::add discovers that two files hold the same data through their fragment ids (the index lookup of
 * ZSFX/zsfx.cpp:651-659 over ids made by the fragment loop + SHA1).  Both are functions of the file's bytes alone, so a
 * file that EQUALS an earlier one has that file's fragments and ids; comparing is HBM-bound streaming, hashing is not.
 * file_rep[f] (HOST, nfiles entries) = earliest file with the same bytes as file f, established by comparing every
 * byte, else f.  Only files of at least min_bytes whose length occurs more than once are looked at.
 * stats (HOST, may be NULL): twins found, their bytes, files compared, bytes compared. */
int zpq_file_twins_dev(zpq_ctx* ctx, const uint8_t* d_base, const uint64_t* file_off, size_t nfiles, uint64_t min_bytes,
                       uint32_t* file_rep, uint64_t stats[4]);
/* zpq_fragment_dev + zpq_sha1_extents_dev over the same files in one call (d_digests: DEVICE, 20 bytes per record,
 * 4-byte aligned), with the twin fold in front: representatives are fragmented and hashed, twins receive their
 * representative's records moved to their own offsets.  Output identical to the two separate calls for any input.
 * flags: ZPQ_FS_NO_TWINS = skip the fold.  file_rep (HOST, may be NULL) and stats as above. */
#define ZPQ_FS_NO_TWINS 1u
int zpq_fragment_sha1_dev(zpq_ctx* ctx, const uint8_t* d_base, const uint64_t* file_off, size_t nfiles,
                          const zpq_fragment_params* p, uint64_t* d_frag_off, uint32_t* d_frag_len, uint32_t* d_frag_file,
                          uint8_t* d_digests, size_t frag_cap, size_t* nfrags, uint32_t flags, uint32_t* file_rep,
                          uint64_t stats[4]);

/* SHA-256 of every file [file_off[f], file_off[f+1]) (HOST offsets) into d_digests (DEVICE, 32 bytes per file, 4-byte
 * aligned): zpq_sha256_extents_dev over whole files with the twin fold in front -- what extract's per-file check needs
 * when a tree holds copies (ZSFX/zsfx.cpp:2018-2281).  flags / stats as for zpq_fragment_sha1_dev. */
int zpq_sha256_files_dev(zpq_ctx* ctx, const uint8_t* d_base, const uint64_t* file_off, size_t nfiles, uint8_t* d_digests,
                         uint32_t flags, uint64_t stats[4]);

/* ---- file-level checksums (section 8f-2) ------------------------------------------------------ */
/* What zpaqfranz stores per file in the i blocks and re-checks on extract / test (README.md:95-105; attribute
 * layout: SURVEY.md Appendix B.4): CRC-32 (zlib polynomial), XXH64 (seed 0) and, optionally, BLAKE3 (32 bytes).
 * Files are the extents [file_off[f], file_off[f+1]) of one device buffer (file_off: HOST array, nfiles+1
 * entries, as for zpq_fragment_dev).  Results go to HOST arrays; any of them may be NULL to skip that hash. */
int zpq_file_checksums_dev(zpq_ctx* ctx, const uint8_t* d_base, const uint64_t* file_off, size_t nfiles,
                           uint32_t* crc32, uint64_t* xxh64, uint8_t* blake3 /* 32*nfiles */);

/* ---- dedup index (row a3) ------------------------------------------------------------------ */
/* first[i] = smallest j <= i with digest[j] == digest[i] (20-byte SHA-1 keys): fragment i is new
 * iff first[i] == i.  Deterministic regardless of scheduling. */
int zpq_dedup_dev(zpq_ctx* ctx, const uint8_t* d_digests, size_t n, uint32_t* d_first);

/* ---- block method hint (row a4) ------------------------------------------------------------- */
/* What zpaq's add() appends to the method it hands to compressBlock, "method,R,t" (ZSFX/libzpaq.h:86-135), comes from
 * per-fragment statistics: d_stats[4*i+0] = order-1 prediction hits of fragment i (c == o1[c1] with the table reset at
 * the fragment start: the count the fragment loop keeps), [1] = 1 if the fragment looks like
 * text, [2] = 1 if it looks like x86 code, [3] = its length.  R = 256 * hits / bytes over a block's fragments (capped
 * at 255), t = (exe fragments > 1/8) * 2 + (text fragments > 1/4).  The text / exe detectors are this engine's own
 * (zpaqfranz.cpp is not in the snapshot): parity unpinned.  Extents as for zpq_sha1_extents_dev. */
int zpq_fragment_stats_dev(zpq_ctx* ctx, const uint8_t* d_base, const uint64_t* d_off, const uint32_t* d_len, size_t n,
                           uint32_t* d_stats);

/* ---- block packer data movement (row a4) ---------------------------------------------------- */
/* Copies n extents: d_dst_base[dst_off[i] .. +len[i]) = d_src_base[src_off[i] .. +len[i]).
 * The host-side packer (which fragments go to which block, in which order) stays host logic as in
 * the reference; only the bytes move, HBM to HBM.  All index arrays are DEVICE pointers. */
int zpq_gather_dev(zpq_ctx* ctx, const uint8_t* d_src_base, const uint64_t* d_src_off,
                   const uint32_t* d_len, const uint64_t* d_dst_off, size_t n, uint8_t* d_dst_base);

/* ---- LZ77 level-1 code stream (row a8) ----------------------------------------------------- */
/* One job = one ZPAQ block's input.  args[9] are LZBuffer's (ZSFX/libzpaq.cpp:6128-6138):
 * args[0]=log2 MiB, args[1]=1, args[2]=min match 4..31, args[3]=0, args[4]=log2 bucket 0..3,
 * args[5]=log2 hash size <= 26 (and < args[0]+21), args[6]=0.  Output: the exact byte stream
 * LZBuffer::read() yields.  d_out capacity per job must be >= zpq_lz77_bound(n). */
typedef struct zpq_lz77_job {
  const uint8_t* d_in;   /* device; readable for 64 bytes past n */
  uint32_t n;            /* <= 2^(20+args[0]) */
  int32_t args[9];
  uint8_t* d_out;        /* device */
  uint32_t out_cap;
  uint32_t out_len;      /* result */
  uint32_t n_matches;    /* result (diagnostic) */
} zpq_lz77_job;
size_t zpq_lz77_bound(size_t n);
int zpq_lz77_encode_dev(zpq_ctx* ctx, zpq_lz77_job* jobs, size_t njobs);
/* Inverse (what the level-1 PCOMP does): d_in/n = code stream, rb = max(args[0]-4,0).  With bit 31 of rb set the stream
 * holds LZBuffer's byte-aligned codes (level 2, ZSFX/libzpaq.cpp:6221-6224) and the low byte of rb is the minimum match
 * length (args[2]): what the level-2 post-processor of methods 3 and 4 undoes. */
typedef struct zpq_lz77_dec_job {
  const uint8_t* d_in;
  uint32_t n;
  uint32_t rb;
  uint8_t* d_out;
  uint32_t out_cap;
  uint32_t out_len;      /* result */
  int32_t status;        /* result: ZPQ_OK / ZPQ_ERR_FORMAT / ZPQ_ERR_CAPACITY */
} zpq_lz77_dec_job;
int zpq_lz77_decode_dev(zpq_ctx* ctx, zpq_lz77_dec_job* jobs, size_t njobs);

/* ---- compressBlock / Decompresser for the stored + LZ77 family (rows a5, a10, a11, a15-a17) - */
/* Mirrors libzpaq::compressBlock(in, out, method, filename, comment, dosha1)
 * (ZSFX/libzpaq.h:1505): method is "0", "1", "LB,R,t" with L in {0,1}, or an explicit
 * "x<N1>,0" / "x<N1>,1,<minmatch>,0,<bucket>,<hashbits>".  The framed block (13-byte tag, zPQ
 * header, segment, stored sub-blocks, SHA-1 trailer, 255) is written to out. */
typedef struct zpq_block_job {
  const uint8_t* in;      /* block input: device pointer for *_dev, host pointer otherwise */
  uint32_t n;
  const char* method;     /* host C string */
  const char* filename;   /* host C string or NULL */
  const char* comment;    /* host C string or NULL (the decimal size is always prepended) */
  int32_t dosha1;
  uint8_t* out;           /* device (dev variant) or host */
  uint32_t out_cap;       /* >= zpq_block_bound(n, filename, comment) */
  uint32_t out_len;       /* result */
  int32_t status;         /* result, per block */
} zpq_block_job;
size_t zpq_block_bound(size_t n, const char* filename, const char* comment);
int zpq_compress_blocks_dev(zpq_ctx* ctx, zpq_block_job* jobs, size_t njobs);
int zpq_compress_blocks(zpq_ctx* ctx, zpq_block_job* jobs, size_t njobs);

/* One framed block in, original bytes out; status ZPQ_ERR_CHECKSUM if a stored SHA-1 does not match
 * and verify != 0. */
typedef struct zpq_unblock_job {
  const uint8_t* in;      /* framed block bytes (device for *_dev, else host) */
  uint32_t n;
  uint8_t* out;
  uint32_t out_cap;
  uint32_t out_len;       /* result */
  uint32_t consumed;      /* result: bytes of `in` that made up the block */
  int32_t status;         /* result */
  uint8_t sha1[20];       /* result: SHA-1 of the decoded bytes (of the first segment's, if the block has several) */
  /* A block of several segments (a streaming archive with more than one file per block): the decoder's model and the
   * post-processor carry on from segment to segment (Decompresser::decompress, ZSFX/libzpaq.cpp:2307-2337); out receives the
   * segments' bytes back to back, every stored SHA-1 is compared with its own segment's when verify != 0. */
  uint32_t nseg;          /* result: segments in the block */
  uint32_t seg_cap;       /* entries of seg_out_end the caller provides (0: none) */
  uint32_t* seg_out_end;  /* host, optional, result: bytes of out up to and including each segment */
} zpq_unblock_job;
int zpq_decompress_blocks(zpq_ctx* ctx, zpq_unblock_job* jobs, size_t njobs, int verify);
/* The same with framed blocks and outputs resident in HBM (in/out are DEVICE pointers; every `in` readable for
 * 64 bytes past n): the extract-side counterpart of zpq_compress_blocks_dev.  The framing is walked on the
 * device, all blocks of the call are decoded by one launch each of the gather / LZ77 / checksum kernels, and
 * the stored SHA-1 is compared on the device; only the per-block records cross PCIe.  This is what a
 * Jidac::extract over an archive staged in HBM calls once per batch of d blocks (decompressThread,
 * ZSFX/zsfx.cpp:1731-1834). */
int zpq_decompress_blocks_dev(zpq_ctx* ctx, zpq_unblock_job* jobs, size_t njobs, int verify);
/* The post-processor programs the decode side runs natively instead of interpreting: the level-1 LZ77 program for
 * rb = 0..7 raw offset bits (blocks of 16 MiB << rb), with and without the E8E9 inverse.  rb = 0 / no E8E9 is the
 * golden 302-byte program of the reference's fixture.  Host call, no GPU needed. */
int zpq_known_pcomp_bytes(uint32_t rb, int e8e9, uint8_t* out, size_t cap, size_t* len);
/* Compares two device arrays of n digests of digest_size bytes each (fragment SHA-1s of an extracted block
 * against the h table, ZSFX/zsfx.cpp:1811-1834; file checksums against the originals): *mismatches = number
 * of differing entries, *first_mismatch = smallest differing index. */
int zpq_digest_compare_dev(zpq_ctx* ctx, const uint8_t* d_a, const uint8_t* d_b, size_t n, uint32_t digest_size,
                           uint64_t* mismatches, uint64_t* first_mismatch);

/* ---- context-mixing blocks, n > 0 components (rows a11-a16) ---------------------------------- */
/* header = the block header exactly as stored in the archive, starting at hsize[2]
 * (hsize hh hm ph pm n COMP 0 HCOMP 0; ZPAQL::read, ZSFX/libzpaq.cpp:879-921), host pointer.
 * encode: d_in = the bytes the Encoder sees (post-processor preamble + data); the output is the
 *         arithmetic-coded stream including the end-of-segment symbol and the four 0 bytes that
 *         follow it (what sits between the segment header and the 253/254 marker).
 * decode: d_in = that stream; output = the decoded bytes (preamble included).  Decoding stops at
 *         out_cap bytes with status ZPQ_ERR_CAPACITY (the bytes produced so far are valid). */
typedef struct zpq_cm_job {
  const uint8_t* header;  /* host */
  uint32_t header_len;
  const uint8_t* d_in;    /* device, readable 64 bytes past n */
  uint32_t n;
  uint8_t* d_out;         /* device */
  uint32_t out_cap;
  uint32_t out_len;       /* result */
  int32_t status;         /* result */
  /* Segments of one block (Compressor::startSegment ... endSegment more than once before endBlock; Decompresser::decompress
   * after a second findFilename, ZSFX/libzpaq.cpp:2307-2337): the predictor and the HCOMP machine carry on from segment to
   * segment, only the arithmetic coder starts afresh.  nseg <= 1: one segment, as above.  nseg > 1: d_in holds the segments'
   * inputs back to back (encode: the bytes of each; decode: each one's coded stream with its end-of-segment symbol and four
   * 0 bytes), seg_len[s] (host) = input bytes of segment s (their sum = n), and seg_out_end[s] (host, result) = output bytes
   * produced up to and including segment s (the outputs lie back to back in d_out as well). */
  uint32_t nseg;
  const uint32_t* seg_len;
  uint32_t* seg_out_end;
} zpq_cm_job;
int zpq_cm_encode_dev(zpq_ctx* ctx, zpq_cm_job* jobs, size_t njobs);
int zpq_cm_decode_dev(zpq_ctx* ctx, zpq_cm_job* jobs, size_t njobs);
/* The model-independent tables the predictor uses (generated, not stored): squash[4096],
 * stretch[32768], dt[1024], dt2k[256], ns[1024] -- exported so that tests can compare them with
 * the reference's literal tables (ZSFX/libzpaq.cpp:718-847, 1264-1695). Host call, no GPU needed. */
int zpq_cm_tables(uint16_t* squash, int16_t* stretch, int32_t* dt, int32_t* dt2k, uint8_t* ns);
/* Generic post-processor (rows a14, a16): runs the PCOMP program `pcomp[0..psize)` (host pointer,
 * bytecode without the 2-byte length) with H = 2^ph words, M = 2^pm bytes once per input byte and
 * once with 2^32-1 at the end (PostProcessor::write, ZSFX/libzpaq.cpp:2185-2226); OUT bytes -> d_out. */
int zpq_pcomp_run_dev(zpq_ctx* ctx, const uint8_t* pcomp, uint32_t psize, uint32_t ph, uint32_t pm,
                      const uint8_t* d_in, uint32_t n, uint8_t* d_out, uint32_t out_cap, uint32_t* out_len);
/* The same over the segments of one block: d_in holds the decoded streams of the segments back to back (the first without its
 * post-processor preamble), seg_len[s] (host) bytes each; the machine keeps its state from segment to segment and runs once
 * with 2^32-1 at the end of each (Decompresser::decompress initialises the post-processor for the first segment only,
 * ZSFX/libzpaq.cpp:2312-2317); seg_out_end[s] (host, result) = output bytes up to and including segment s. */
int zpq_pcomp_run_segments_dev(zpq_ctx* ctx, const uint8_t* pcomp, uint32_t psize, uint32_t ph, uint32_t pm,
                               const uint8_t* d_in, uint32_t n, const uint32_t* seg_len, uint32_t nseg, uint8_t* d_out,
                               uint32_t out_cap, uint32_t* seg_out_end, uint32_t* out_len);

/* ---- E8E9 pre-processor (row a7) ------------------------------------------------------------- */
/* libzpaq's e8e9() (ZSFX/libzpaq.cpp:6117-6126) over d_buf[0..n) in place: the x86 CALL/JMP filter
 * compressBlock applies before modelling when the type hint has the exe bit. */
int zpq_e8e9_dev(zpq_ctx* ctx, uint8_t* d_buf, size_t n);
/* The inverse (what the E8E9 variants of the post-processor do when a segment ends), out of place:
 * d_out[0..n) = original bytes of the transformed d_in[0..n).  d_in must be readable 64 bytes past n. */
int zpq_e8e9_inverse_dev(zpq_ctx* ctx, const uint8_t* d_in, uint8_t* d_out, size_t n);

/* ---- suffix array (row a9) -------------------------------------------------------------------- */
/* divsufsort(T, SA, n) (ZSFX/libzpaq.cpp:6047-6072, called at :6304 for LZ77-SA and BWT): d_sa[0..n) = the
 * positions of d_in[0..n) in suffix order (a shorter suffix sorts before a longer one it is a prefix of).
 * d_isa, when not NULL, receives the inverse (d_isa[d_sa[j]] = j).  n < 2^31.  Jobs of zpq_lz77_encode_dev
 * with args[5]-args[0] >= 21 (method 2: "x<N>,1,4,0,7,<21+N>,1"; with args[1] & 3 == 2 the byte-aligned codes of
 * methods 3 and 4, e.g. "x<N>,2,12,0,7,<21+N>,1") build it internally. */
int zpq_suffix_array_dev(zpq_ctx* ctx, const void* d_in, size_t n, uint32_t* d_sa, uint32_t* d_isa);
/* What LZBuffer emits for (args[1] & 3) == 3, the BWT front end of methods 3 and 4 (ZSFX/libzpaq.cpp:6317-6326):
 * d_out[0] = last input byte, d_out[1..n] = the byte before every suffix in suffix order (255 for the suffix that is
 * the whole block), d_out[n+1..n+4] = that suffix's 1-based rank, LSB first.  d_out holds n+5 bytes.  (compressBlock
 * serves these methods since round 3 -- level 3 for text, "x<N>,3..." -- with the post-processor program decode-pinned by
 * the reference's own PostProcessor) */
int zpq_bwt_dev(zpq_ctx* ctx, const void* d_in, size_t n, uint8_t* d_out);

/* ---- block configuration on the host (rows a5, a6); no GPU needed, ctx may be NULL ------------- */
/* compressBlock()'s expansion of "0".."5"[B][,R,t] into the x/0 method it stands for
 * (libzpaq 7.15 compressBlock; the snapshot's ZSFX/libzpaq.cpp ends before it, see config.hip).
 * `data` (n bytes, host) is only read for level 5. */
int zpq_expand_method(zpq_ctx* ctx, const char* method, const uint8_t* data, size_t n, char* out, size_t cap);
/* makeConfig(): "x..."/"0..." method -> ZPAQ config source (NUL terminated) and $1..$9 in args. */
int zpq_make_config(zpq_ctx* ctx, const char* method, int32_t args[9], char* out, size_t cap, size_t* out_len);
/* libzpaq::Compiler (ZSFX/libzpaq.h:1373-1420, ZSFX/libzpaq.cpp:2500-2706): config source ->
 * header = hsize[2] hh hm ph pm n COMP 0 HCOMP 0 (what Compressor::startBlock writes after "zPQ" level
 * type), pcomp = post-processor bytecode with its closing 0 (empty when the config has none). */
int zpq_compile_config(zpq_ctx* ctx, const char* source, const int32_t* args, uint8_t* header, size_t header_cap,
                       size_t* header_len, uint8_t* pcomp, size_t pcomp_cap, size_t* pcomp_len);
/* The block headers libzpaq's Compressor::startBlock(int level) selects, level 1..3 = min.cfg / mid.cfg / max.cfg of the
 * ZPAQ distribution (ZSFX/libzpaq.h:1346; libzpaq keeps them as a byte array in the half of libzpaq.cpp the snapshot
 * lacks): hsize[2] hh hm ph pm n COMP 0 HCOMP 0, 28 / 71 / 198 bytes.  zpq_builtin_model_source gives the config text. */
int zpq_builtin_model(int level, uint8_t* header, size_t header_cap, size_t* header_len);
const char* zpq_builtin_model_source(int level);

#ifdef __cplusplus
}
#endif
#endif /* ZPAQHIP_H */
