"""ctypes mirror of include/zpaqhip.h.  Names, argument meaning and error behaviour follow the C ABI
one to one; nothing here computes anything on the CPU."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    return os.path.join(_HERE, "libzpaqhip.so")


class ZpqError(RuntimeError):
    def __init__(self, status, detail):
        super().__init__("zpaqhip status %d: %s" % (status, detail))
        self.status = status


class FragmentParams(C.Structure):
    _fields_ = [("fragment_log2", C.c_uint32), ("min_fragment", C.c_uint32), ("max_fragment", C.c_uint32)]


class Lz77Job(C.Structure):
    _fields_ = [("d_in", C.c_void_p), ("n", C.c_uint32), ("args", C.c_int32 * 9), ("d_out", C.c_void_p),
                ("out_cap", C.c_uint32), ("out_len", C.c_uint32), ("n_matches", C.c_uint32)]


class Lz77DecJob(C.Structure):
    _fields_ = [("d_in", C.c_void_p), ("n", C.c_uint32), ("rb", C.c_uint32), ("d_out", C.c_void_p),
                ("out_cap", C.c_uint32), ("out_len", C.c_uint32), ("status", C.c_int32)]


class BlockJob(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("n", C.c_uint32), ("method", C.c_char_p), ("filename", C.c_char_p),
                ("comment", C.c_char_p), ("dosha1", C.c_int32), ("out", C.c_void_p), ("out_cap", C.c_uint32),
                ("out_len", C.c_uint32), ("status", C.c_int32)]


class UnblockJob(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("n", C.c_uint32), ("out", C.c_void_p), ("out_cap", C.c_uint32),
                ("out_len", C.c_uint32), ("consumed", C.c_uint32), ("status", C.c_int32), ("sha1", C.c_uint8 * 20),
                ("nseg", C.c_uint32), ("seg_cap", C.c_uint32), ("seg_out_end", C.POINTER(C.c_uint32))]


class CmJob(C.Structure):
    _fields_ = [("header", C.c_char_p), ("header_len", C.c_uint32), ("d_in", C.c_void_p), ("n", C.c_uint32),
                ("d_out", C.c_void_p), ("out_cap", C.c_uint32), ("out_len", C.c_uint32), ("status", C.c_int32),
                ("nseg", C.c_uint32), ("seg_len", C.POINTER(C.c_uint32)), ("seg_out_end", C.POINTER(C.c_uint32))]


_lib = None


def load():
    """Loads libzpaqhip.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise ImportError("libzpaqhip.so is missing: run `python -m zpaqfranz_amd.build` (hipcc, gfx950)")
    L = C.CDLL(p)
    L.zpq_strerror.restype = C.c_char_p
    L.zpq_last_error.restype = C.c_char_p
    L.zpq_last_error.argtypes = [C.c_void_p]
    L.zpq_stream.restype = C.c_void_p
    L.zpq_stream.argtypes = [C.c_void_p]
    for name in ("zpq_fragment_capacity", "zpq_lz77_bound", "zpq_block_bound"):
        getattr(L, name).restype = C.c_size_t
    L.zpq_lz77_bound.argtypes = [C.c_size_t]
    L.zpq_block_bound.argtypes = [C.c_size_t, C.c_char_p, C.c_char_p]
    L.zpq_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.zpq_destroy.argtypes = [C.c_void_p]
    L.zpq_destroy.restype = None
    L.zpq_sync.argtypes = [C.c_void_p]
    L.zpq_dev_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.zpq_dev_free.argtypes = [C.c_void_p, C.c_void_p]
    L.zpq_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.zpq_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.zpq_dev_memset.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]
    L.zpq_fragment_stats_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.zpq_device_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_char_p, C.c_size_t]
    L.zpq_sha1_extents_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.zpq_sha256_extents_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.zpq_sha1_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.zpq_sha256_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.zpq_fragment_params_default.argtypes = [C.POINTER(FragmentParams)]
    L.zpq_fragment_params_default.restype = None
    L.zpq_fragment_capacity.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(FragmentParams)]
    L.zpq_fragment_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(FragmentParams), C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zpq_file_twins_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p, C.c_void_p]
    L.zpq_fragment_sha1_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(FragmentParams), C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_uint32, C.c_void_p,
                                        C.c_void_p]
    L.zpq_sha256_files_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_void_p]
    L.zpq_dedup_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.zpq_gather_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.zpq_profile_enable.argtypes = [C.c_void_p, C.c_int]
    L.zpq_profile_report.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.zpq_cm_encode_dev.argtypes = [C.c_void_p, C.POINTER(CmJob), C.c_size_t]
    L.zpq_cm_decode_dev.argtypes = [C.c_void_p, C.POINTER(CmJob), C.c_size_t]
    L.zpq_cm_tables.argtypes = [C.c_void_p] * 5
    L.zpq_pcomp_run_dev.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                    C.c_uint32, C.POINTER(C.c_uint32)]
    L.zpq_lz77_encode_dev.argtypes = [C.c_void_p, C.POINTER(Lz77Job), C.c_size_t]
    L.zpq_lz77_decode_dev.argtypes = [C.c_void_p, C.POINTER(Lz77DecJob), C.c_size_t]
    L.zpq_compress_blocks_dev.argtypes = [C.c_void_p, C.POINTER(BlockJob), C.c_size_t]
    L.zpq_compress_blocks.argtypes = [C.c_void_p, C.POINTER(BlockJob), C.c_size_t]
    L.zpq_decompress_blocks.argtypes = [C.c_void_p, C.POINTER(UnblockJob), C.c_size_t, C.c_int]
    L.zpq_decompress_blocks_dev.argtypes = [C.c_void_p, C.POINTER(UnblockJob), C.c_size_t, C.c_int]
    L.zpq_digest_compare_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.zpq_file_checksums_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.zpq_e8e9_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.zpq_suffix_array_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.zpq_bwt_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.zpq_expand_method.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.zpq_make_config.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int32), C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zpq_compile_config.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int32), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                     C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    _lib = L
    return L


class ConfigRefused(ValueError):
    """The method needs a pre-processor no fixture pins (status ZPQ_ERR_METHOD) or the source does not compile."""


def expand_method(method, data=b""):
    """compressBlock's "0".."5"[B][,R,t] -> the x/0 method it stands for (host only, no GPU)."""
    L = load()
    out = C.create_string_buffer(4096)
    rc = L.zpq_expand_method(None, method.encode(), bytes(data), len(data), out, 4096)
    if rc:
        raise ConfigRefused("zpq_expand_method(%r) -> %d" % (method, rc))
    return out.value.decode()


def make_config(method):
    """makeConfig: x/0 method -> (config source, [$1..$9]) (host only, no GPU)."""
    L = load()
    args = (C.c_int32 * 9)()
    out = C.create_string_buffer(1 << 16)
    n = C.c_size_t(0)
    rc = L.zpq_make_config(None, method.encode(), args, out, 1 << 16, C.byref(n))
    if rc:
        raise ConfigRefused("zpq_make_config(%r) -> %d" % (method, rc))
    return out.value.decode(), list(args)


def compile_config(source, args=()):
    """libzpaq::Compiler: config source -> (header bytes hsize..HCOMP 0, pcomp bytecode) (host only, no GPU)."""
    L = load()
    a = (C.c_int32 * 9)(*(list(args) + [0] * 9)[:9])
    h = (C.c_ubyte * 70000)()
    p = (C.c_ubyte * 70000)()
    hl, pl = C.c_size_t(0), C.c_size_t(0)
    rc = L.zpq_compile_config(None, source.encode(), a, h, 70000, C.byref(hl), p, 70000, C.byref(pl))
    if rc:
        raise ConfigRefused("zpq_compile_config -> %d" % rc)
    return bytes(h[:hl.value]), bytes(p[:pl.value])


class DevBuf:
    """A device allocation owned through zpq_dev_alloc (tests use this; bench.py uses torch tensors)."""

    def __init__(self, eng, nbytes):
        self.eng, self.nbytes = eng, nbytes
        p = C.c_void_p()
        eng._ck(eng.L.zpq_dev_alloc(eng.ctx, nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, data, offset=0):
        b = bytes(data)
        if b:
            self.eng._ck(self.eng.L.zpq_h2d(self.eng.ctx, self.ptr + offset, b, len(b)))
        return self

    def download(self, nbytes=None, offset=0):
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        out = C.create_string_buffer(max(1, nbytes))
        if nbytes:
            self.eng._ck(self.eng.L.zpq_d2h(self.eng.ctx, out, self.ptr + offset, nbytes))
        return out.raw[:nbytes]

    def free(self):
        if self.ptr:
            self.eng.L.zpq_dev_free(self.eng.ctx, self.ptr)
            self.ptr = None


class Engine:
    PAD = 64  # every data buffer is padded so that the kernels' 8/16-byte reads stay in bounds

    def __init__(self, device=0):
        self.L = load()
        ctx = C.c_void_p()
        rc = self.L.zpq_create(device, C.byref(ctx))
        if rc != 0:
            raise ZpqError(rc, self.L.zpq_strerror(rc).decode())
        self.ctx = ctx

    def close(self):
        if self.ctx:
            self.L.zpq_destroy(self.ctx)
            self.ctx = None

    def _ck(self, rc):
        if rc != 0:
            raise ZpqError(rc, "%s (%s)" % (self.L.zpq_strerror(rc).decode(), self.L.zpq_last_error(self.ctx).decode()))

    def sync(self):
        self._ck(self.L.zpq_sync(self.ctx))

    def stream(self):
        return self.L.zpq_stream(self.ctx)

    def device_info(self):
        info = (C.c_int64 * 6)()
        name = C.create_string_buffer(256)
        self._ck(self.L.zpq_device_info(self.ctx, info, name, 256))
        return dict(name=name.value.decode(), cu=info[0], clock_khz=info[1], mem_clock_khz=info[2], bus_bits=info[3],
                    l2_bytes=info[4], hbm_mib=info[5])

    def alloc(self, nbytes):
        return DevBuf(self, nbytes + self.PAD)

    def upload(self, data):
        return self.alloc(len(data)).upload(data)

    # ---- hashing ------------------------------------------------------------------------------
    def e8e9(self, data):
        """libzpaq e8e9() on the device (host convenience used by the tests)."""
        d = self.upload(data)
        self._ck(self.L.zpq_e8e9_dev(self.ctx, d.ptr, len(data)))
        return d.download(len(data))

    def sha1_many(self, bufs):
        return self._many(bufs, 20, self.L.zpq_sha1_many)

    def sha256_many(self, bufs):
        return self._many(bufs, 32, self.L.zpq_sha256_many)

    def _many(self, bufs, dsz, fn):
        n = len(bufs)
        keep = [C.create_string_buffer(bytes(b), max(1, len(b))) for b in bufs]
        ptrs = (C.c_void_p * max(1, n))(*[C.cast(k, C.c_void_p).value for k in keep])
        lens = (C.c_size_t * max(1, n))(*[len(b) for b in bufs])
        out = C.create_string_buffer(max(1, n * dsz))
        self._ck(fn(self.ctx, ptrs, lens, n, out))
        return [out.raw[i * dsz:(i + 1) * dsz] for i in range(n)]

    def file_checksums_dev(self, d_base, file_off, crc32=True, xxh64=True, blake3=True):
        """CRC-32, XXH64 and BLAKE3 of the files [file_off[f], file_off[f+1]) of a device buffer -> three lists (or None)."""
        n = len(file_off) - 1
        arr = (C.c_uint64 * len(file_off))(*file_off)
        c = (C.c_uint32 * max(1, n))() if crc32 else None
        x = (C.c_uint64 * max(1, n))() if xxh64 else None
        b = (C.c_ubyte * max(1, 32 * n))() if blake3 else None
        self._ck(self.L.zpq_file_checksums_dev(self.ctx, d_base, arr, n, c, x, b))
        return (list(c[:n]) if crc32 else None, list(x[:n]) if xxh64 else None,
                [bytes(b[32 * i:32 * i + 32]) for i in range(n)] if blake3 else None)

    def file_checksums(self, files, **kw):
        """Host convenience used by the tests: list of bytes -> (crc32 list, xxh64 list, blake3 list)."""
        file_off = [0]
        for f in files:
            file_off.append(file_off[-1] + len(f))
        d = self.upload(b"".join(files))
        try:
            return self.file_checksums_dev(d.ptr, file_off, **kw)
        finally:
            d.free()

    def digest_compare_dev(self, d_a, d_b, n, digest_size):
        """(number of differing digests, index of the first one) of two device digest arrays."""
        m, f = C.c_uint64(0), C.c_uint64(0)
        self._ck(self.L.zpq_digest_compare_dev(self.ctx, d_a, d_b, n, digest_size, C.byref(m), C.byref(f)))
        return m.value, f.value

    def sha1_extents_dev(self, d_base, d_off, d_len, n, d_digests):
        self._ck(self.L.zpq_sha1_extents_dev(self.ctx, d_base, d_off, d_len, n, d_digests))

    def sha256_extents_dev(self, d_base, d_off, d_len, n, d_digests):
        self._ck(self.L.zpq_sha256_extents_dev(self.ctx, d_base, d_off, d_len, n, d_digests))

    def sha256_files_dev(self, d_base, file_off, d_digests, twins=True):
        """SHA-256 of every file of one device buffer, twin files folded first (digest of the earlier equal file); -> twin stats"""
        arr = (C.c_uint64 * len(file_off))(*file_off)
        st = (C.c_uint64 * 4)()
        self._ck(self.L.zpq_sha256_files_dev(self.ctx, d_base, arr, len(file_off) - 1, d_digests, 0 if twins else 1, st))
        return dict(twins=st[0], twin_bytes=st[1], compared=st[2], compared_bytes=st[3])

    # ---- fragmenter -----------------------------------------------------------------------------
    def fragment_params(self, fragment=6, min_fragment=None, max_fragment=None):
        p = FragmentParams()
        p.fragment_log2 = fragment
        p.min_fragment = (64 << fragment) if min_fragment is None else min_fragment
        p.max_fragment = (8128 << fragment) if max_fragment is None else max_fragment
        return p

    def fragment_capacity(self, file_off, params):
        arr = (C.c_uint64 * len(file_off))(*file_off)
        return self.L.zpq_fragment_capacity(arr, len(file_off) - 1, C.byref(params))

    def fragment_dev(self, d_base, file_off, params, d_frag_off, d_frag_len, d_frag_file, cap):
        arr = (C.c_uint64 * len(file_off))(*file_off)
        n = C.c_size_t(0)
        self._ck(self.L.zpq_fragment_dev(self.ctx, d_base, arr, len(file_off) - 1, C.byref(params), d_frag_off, d_frag_len,
                                         d_frag_file, cap, C.byref(n)))
        return n.value

    def file_twins_dev(self, d_base, file_off, min_bytes=4096):
        """-> (rep, stats): rep[f] = earliest file with the same bytes as file f (all bytes compared on the device), else f."""
        arr = (C.c_uint64 * len(file_off))(*file_off)
        nfiles = len(file_off) - 1
        rep = (C.c_uint32 * max(1, nfiles))()
        st = (C.c_uint64 * 4)()
        self._ck(self.L.zpq_file_twins_dev(self.ctx, d_base, arr, nfiles, min_bytes, rep, st))
        return list(rep[:nfiles]), dict(twins=st[0], twin_bytes=st[1], compared=st[2], compared_bytes=st[3])

    def fragment_sha1_dev(self, d_base, file_off, params, d_frag_off, d_frag_len, d_frag_file, d_digests, cap, twins=True, want_rep=False):
        """fragment_dev + sha1_extents_dev in one call, twin files folded first (include/zpaqhip.h); -> n or (n, rep, stats)."""
        arr = (C.c_uint64 * len(file_off))(*file_off)
        nfiles = len(file_off) - 1
        n = C.c_size_t(0)
        rep = (C.c_uint32 * max(1, nfiles))() if want_rep else None
        st = (C.c_uint64 * 4)()
        self._ck(self.L.zpq_fragment_sha1_dev(self.ctx, d_base, arr, nfiles, C.byref(params), d_frag_off, d_frag_len, d_frag_file,
                                              d_digests, cap, C.byref(n), 0 if twins else 1, rep, st))
        self.last_twin_stats = dict(twins=st[0], twin_bytes=st[1], compared=st[2], compared_bytes=st[3])
        if want_rep:
            return n.value, list(rep[:nfiles]), self.last_twin_stats
        return n.value

    def fragment_files(self, files, params=None):
        """Host convenience used by the tests: list of bytes -> [(file, offset_in_file, length)]."""
        import struct
        params = params or self.fragment_params()
        file_off = [0]
        for f in files:
            file_off.append(file_off[-1] + len(f))
        data = self.upload(b"".join(files))
        cap = max(1, self.fragment_capacity(file_off, params))
        fo, fl, ff = self.alloc(cap * 8), self.alloc(cap * 4), self.alloc(cap * 4)
        try:
            n = self.fragment_dev(data.ptr, file_off, params, fo.ptr, fl.ptr, ff.ptr, cap)
            offs = struct.unpack("<%dQ" % n, fo.download(n * 8))
            lens = struct.unpack("<%dI" % n, fl.download(n * 4))
            fil = struct.unpack("<%dI" % n, ff.download(n * 4))
        finally:
            for b in (data, fo, fl, ff):
                b.free()
        return [(fil[i], offs[i] - file_off[fil[i]], lens[i]) for i in range(n)]

    def gather_dev(self, d_src_base, d_src_off, d_len, d_dst_off, n, d_dst_base):
        self._ck(self.L.zpq_gather_dev(self.ctx, d_src_base, d_src_off, d_len, d_dst_off, n, d_dst_base))

    def profile(self, on):
        self._ck(self.L.zpq_profile_enable(self.ctx, int(on)))

    def profile_report(self):
        buf = C.create_string_buffer(1 << 16)
        self._ck(self.L.zpq_profile_report(self.ctx, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            nm, cnt, ms = line.split()
            out[nm] = (int(cnt), float(ms))
        return out

    def dedup_dev(self, d_digests, n, d_first):
        self._ck(self.L.zpq_dedup_dev(self.ctx, d_digests, n, d_first))

    # ---- LZ77 -----------------------------------------------------------------------------------
    def lz77_bound(self, n):
        return self.L.zpq_lz77_bound(n)

    def bwt(self, data):
        """LZBuffer's level-3 output for `data` (n + 5 bytes)."""
        n = len(data)
        d_in = self.upload(data) if n else self.alloc(16)
        d_out = self.alloc(n + 16)
        try:
            self._ck(self.L.zpq_bwt_dev(self.ctx, d_in.ptr if n else None, n, d_out.ptr))
            return d_out.download(n + 5)
        finally:
            d_in.free(); d_out.free()

    def suffix_array(self, data, inverse=False):
        """divsufsort's result for `data` (numpy uint32), optionally with the inverse."""
        import numpy as np
        n = len(data)
        d_in = self.upload(data) if n else self.alloc(16)
        d_sa = self.alloc(max(16, 4 * n))
        d_isa = self.alloc(max(16, 4 * n)) if inverse else None
        try:
            self._ck(self.L.zpq_suffix_array_dev(self.ctx, d_in.ptr, n, d_sa.ptr, d_isa.ptr if inverse else None))
            sa = np.frombuffer(d_sa.download(4 * n), dtype=np.uint32).copy() if n else np.zeros(0, np.uint32)
            if inverse:
                isa = np.frombuffer(d_isa.download(4 * n), dtype=np.uint32).copy() if n else np.zeros(0, np.uint32)
                return sa, isa
            return sa
        finally:
            d_in.free(); d_sa.free()
            if d_isa is not None:
                d_isa.free()

    def lz77_encode(self, blocks, argsets):
        """Host convenience: list of bytes + list of args[<=9] -> list of code streams."""
        n = len(blocks)
        jobs = (Lz77Job * max(1, n))()
        ins, outs = [], []
        for i, (b, a) in enumerate(zip(blocks, argsets)):
            d_in = self.upload(b)
            cap = (self.lz77_bound(len(b)) + 15) & ~15
            d_out = self.alloc(cap)
            ins.append(d_in); outs.append(d_out)
            jobs[i].d_in, jobs[i].n, jobs[i].d_out, jobs[i].out_cap = d_in.ptr, len(b), d_out.ptr, cap
            for k, v in enumerate((list(a) + [0] * 9)[:9]):
                jobs[i].args[k] = v
        try:
            self._ck(self.L.zpq_lz77_encode_dev(self.ctx, jobs, n))
            res = [outs[i].download(jobs[i].out_len) for i in range(n)]
            self.last_matches = [jobs[i].n_matches for i in range(n)]
        finally:
            for b in ins + outs:
                b.free()
        return res

    def lz77_decode(self, streams, out_caps, rb=0):
        n = len(streams)
        jobs = (Lz77DecJob * max(1, n))()
        ins, outs = [], []
        for i, s in enumerate(streams):
            d_in = self.upload(s)
            d_out = self.alloc(out_caps[i])
            ins.append(d_in); outs.append(d_out)
            jobs[i].d_in, jobs[i].n, jobs[i].rb, jobs[i].d_out, jobs[i].out_cap = d_in.ptr, len(s), rb, d_out.ptr, out_caps[i]
        try:
            self._ck(self.L.zpq_lz77_decode_dev(self.ctx, jobs, n))
            res = [(jobs[i].status, outs[i].download(jobs[i].out_len)) for i in range(n)]
        finally:
            for b in ins + outs:
                b.free()
        return res

    # ---- context mixing ---------------------------------------------------------------------------
    def cm_code(self, headers, inputs, out_caps, encode):
        """Host convenience: arithmetic-code (encode=True) or decode each input under its block header."""
        n = len(inputs)
        jobs = (CmJob * max(1, n))()
        ins, outs = [], []
        for i, b in enumerate(inputs):
            d_in = self.upload(b); d_out = self.alloc(out_caps[i])
            ins.append(d_in); outs.append(d_out)
            jobs[i].header, jobs[i].header_len = bytes(headers[i]), len(headers[i])
            jobs[i].d_in, jobs[i].n, jobs[i].d_out, jobs[i].out_cap = d_in.ptr, len(b), d_out.ptr, out_caps[i]
        try:
            fn = self.L.zpq_cm_encode_dev if encode else self.L.zpq_cm_decode_dev
            rc = fn(self.ctx, jobs, n)
            if rc != 0 and all(jobs[i].status == 0 for i in range(n)):
                self._ck(rc)
            res = [(jobs[i].status, outs[i].download(min(jobs[i].out_len, out_caps[i]))) for i in range(n)]
        finally:
            for b in ins + outs:
                b.free()
        return res

    def cm_code_segments(self, header, segments, out_cap, encode):
        """One block of several segments (the model carries on from one to the next): segments = the bytes of each
        (encode) or each one's coded stream (decode) -> (status, [output of each segment])."""
        job = (CmJob * 1)()
        data = b"".join(segments)
        d_in = self.upload(data); d_out = self.alloc(out_cap)
        k = len(segments)
        lens = (C.c_uint32 * k)(*[len(x) for x in segments]); ends = (C.c_uint32 * k)()
        job[0].header, job[0].header_len = bytes(header), len(header)
        job[0].d_in, job[0].n, job[0].d_out, job[0].out_cap = d_in.ptr, len(data), d_out.ptr, out_cap
        job[0].nseg, job[0].seg_len, job[0].seg_out_end = k, lens, ends
        try:
            fn = self.L.zpq_cm_encode_dev if encode else self.L.zpq_cm_decode_dev
            rc = fn(self.ctx, job, 1)
            if rc != 0 and job[0].status == 0:
                self._ck(rc)
            out = d_out.download(min(job[0].out_len, out_cap))
            cuts = [0] + [min(int(e), len(out)) for e in ends] if k > 1 else [0, len(out)]
            return job[0].status, [out[cuts[i]:cuts[i + 1]] for i in range(k)]
        finally:
            d_in.free(); d_out.free()

    def pcomp_run(self, pcomp, ph, pm, data, out_cap):
        d_in = self.upload(data); d_out = self.alloc(out_cap)
        n = C.c_uint32(0)
        try:
            self._ck(self.L.zpq_pcomp_run_dev(self.ctx, bytes(pcomp), len(pcomp), ph, pm, d_in.ptr, len(data), d_out.ptr, out_cap, C.byref(n)))
            return d_out.download(n.value)
        finally:
            d_in.free(); d_out.free()

    # ---- compressBlock / Decompresser -------------------------------------------------------------
    def block_bound(self, n, filename=None, comment=None):
        return self.L.zpq_block_bound(n, filename, comment)

    def compress_blocks(self, blocks, methods, filenames=None, comments=None, dosha1=True):
        """Mirror of libzpaq::compressBlock for a batch of host buffers -> list of (status, framed bytes)."""
        n = len(blocks)
        jobs = (BlockJob * max(1, n))()
        keep_in, keep_out = [], []
        for i, b in enumerate(blocks):
            fn = filenames[i] if filenames else None
            cm = comments[i] if comments else None
            fn = fn.encode("latin1") if isinstance(fn, str) else fn
            cm = cm.encode("latin1") if isinstance(cm, str) else cm
            cap = self.block_bound(len(b), fn, cm)
            ib = C.create_string_buffer(bytes(b), max(1, len(b)))
            ob = C.create_string_buffer(cap)
            keep_in.append(ib); keep_out.append(ob)
            jobs[i].in_ = C.cast(ib, C.c_void_p).value
            jobs[i].n = len(b)
            jobs[i].method = methods[i].encode()
            jobs[i].filename, jobs[i].comment = fn, cm
            jobs[i].dosha1 = int(dosha1)
            jobs[i].out = C.cast(ob, C.c_void_p).value
            jobs[i].out_cap = cap
        rc = self.L.zpq_compress_blocks(self.ctx, jobs, n)
        if rc != 0 and all(jobs[i].status == 0 for i in range(n)):
            self._ck(rc)
        return [(jobs[i].status, keep_out[i].raw[:jobs[i].out_len]) for i in range(n)]

    def compress_blocks_dev(self, jobs, n):
        rc = self.L.zpq_compress_blocks_dev(self.ctx, jobs, n)
        if rc != 0:
            self._ck(rc)

    def decompress_blocks(self, framed, out_caps, verify=True):
        n = len(framed)
        jobs = (UnblockJob * max(1, n))()
        keep_in, keep_out = [], []
        for i, b in enumerate(framed):
            ib = C.create_string_buffer(bytes(b), max(1, len(b)))
            ob = C.create_string_buffer(max(1, out_caps[i]))
            keep_in.append(ib); keep_out.append(ob)
            jobs[i].in_, jobs[i].n = C.cast(ib, C.c_void_p).value, len(b)
            jobs[i].out, jobs[i].out_cap = C.cast(ob, C.c_void_p).value, out_caps[i]
        self.L.zpq_decompress_blocks(self.ctx, jobs, n, int(verify))
        return [dict(status=jobs[i].status, data=keep_out[i].raw[:jobs[i].out_len], consumed=jobs[i].consumed,
                     sha1=bytes(jobs[i].sha1)) for i in range(n)]


    def decompress_blocks_dev(self, jobs, n, verify=True):
        """zpq_decompress_blocks_dev on a prepared UnblockJob array (device pointers); per-job status carries the outcome."""
        return self.L.zpq_decompress_blocks_dev(self.ctx, jobs, n, int(verify))

    def decompress_blocks_resident(self, framed, out_caps, verify=True):
        """Host convenience used by the tests: uploads the framed blocks, decodes them device to device in ONE call,
        downloads the results (same return shape as decompress_blocks)."""
        n = len(framed)
        jobs = (UnblockJob * max(1, n))()
        ins = [self.upload(b) for b in framed]
        outs = [self.alloc(max(1, c)) for c in out_caps]
        try:
            for i in range(n):
                jobs[i].in_, jobs[i].n = ins[i].ptr, len(framed[i])
                jobs[i].out, jobs[i].out_cap = outs[i].ptr, out_caps[i]
            self.decompress_blocks_dev(jobs, n, verify)
            return [dict(status=jobs[i].status, data=outs[i].download(jobs[i].out_len), consumed=jobs[i].consumed,
                         sha1=bytes(jobs[i].sha1)) for i in range(n)]
        finally:
            for b in ins + outs:
                b.free()


# ---- journaling archives (zpaqfranz_amd/shim/jidac_gpu.cpp) ---------------------------------------------
_shim = None


def load_shim():
    global _shim
    if _shim is None:
        load()
        p = os.path.join(_HERE, "libzpaq_jidac.so")
        if not os.path.exists(p):
            raise ImportError("libzpaq_jidac.so is missing: run `python -m zpaqfranz_amd.build`")
        S = C.CDLL(p)
        S.zpqj_free.argtypes = [C.c_void_p]
        S.zpqj_free.restype = None
        S.zpqj_add.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                               C.c_int64, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p]
        S.zpqj_add_multi.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_int64, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p]
        S.zpqj_extract.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        S.zpqj_verify.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p]
        S.zpqj_extract_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32,
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p]
        S.zpqj_add_opts.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                    C.c_int64, C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p]
        S.zpqj_add_sharded.argtypes = [C.c_void_p, C.c_int, C.c_int, ALLGATHERV, C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_size_t), C.c_void_p]
        S.zpqj_shard_files.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        S.zpqj_add_sharded_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, ALLGATHERV, ALLGATHERV, C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_size_t), C.c_void_p]
        S.zpqj_add_dev.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64,
                                   C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p]
        _shim = S
    return _shim


ALLGATHERV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t))


def dist_allgather_bytes(group=None):
    """An all-gather of byte strings over torch.distributed (RCCL on the GPU box, gloo on CPU): lengths first, then the
    strings padded to the longest.  Returns f(bytes) -> [bytes per rank]."""
    import torch
    import torch.distributed as dist

    def f(b):
        world = dist.get_world_size(group)
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        n = torch.tensor([len(b)], dtype=torch.int64, device=dev)
        ns = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(ns, n, group=group)
        ns = [int(x.item()) for x in ns]
        m = max(1, max(ns))
        t = torch.zeros(m, dtype=torch.uint8, device=dev)
        if b:
            t[:len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        ts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(ts, t, group=group)
        return [bytes(ts[r][:ns[r]].cpu().numpy().tobytes()) for r in range(world)]
    return f


def dist_allgather_dev(eng, group=None):
    """The device-memory form of dist_allgather_bytes for jidac_add_sharded_dev over any torch.distributed backend: f(device
    pointer, length) -> [(device pointer, length) of every rank].  A functional stand-in (D2H, all-gather, H2D) for tests over gloo;
    the product's device collective is RcclGather (shim/rccl_gather.cpp: zpqr_allgatherv_dev, HBM -> xGMI -> HBM)."""
    ag = dist_allgather_bytes(group)
    held = []

    def f(d_ptr, n):
        mine = b""
        if n:
            buf = C.create_string_buffer(n)
            eng._ck(eng.L.zpq_d2h(eng.ctx, buf, C.c_void_p(d_ptr), n))
            mine = buf.raw
        got = ag(mine)
        for d in held:
            d.free()
        del held[:]
        out = []
        for g in got:
            if g:
                d = eng.upload(g); held.append(d); out.append((d.ptr, len(g)))
            else:
                out.append((0, 0))
        return out
    return f


class RcclGather:
    """The in-tree collective for jidac_add_sharded (shim/rccl_gather.cpp: rccl.h, no torch): an all-gather of byte strings over
    RCCL on the engine's own stream.  `uid`: the 128 bytes RcclGather.unique_id() returned on rank 0, handed to every rank by
    whatever the launcher offers (a file, the environment, a socket)."""

    @staticmethod
    def lib():
        p = os.path.join(_HERE, "libzpaq_rccl.so")
        if not os.path.exists(p):
            raise ImportError("libzpaq_rccl.so is missing: run `python -m zpaqfranz_amd.build`")
        load()
        R = C.CDLL(p)
        R.zpqr_unique_id.argtypes = [C.c_char_p]
        R.zpqr_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
        R.zpqr_allgatherv.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        R.zpqr_allgatherv_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        R.zpqr_last_error.argtypes = [C.c_void_p]
        R.zpqr_last_error.restype = C.c_char_p
        R.zpqr_destroy.argtypes = [C.c_void_p]
        R.zpqr_destroy.restype = None
        return R

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        rc = RcclGather.lib().zpqr_unique_id(buf)
        if rc != 0:
            raise ZpqError(rc, "zpqr_unique_id")
        return buf.raw

    def __init__(self, eng, rank, world, uid):
        self.R = RcclGather.lib()
        self.comm = C.c_void_p()
        self.rank, self.world = rank, world
        rc = self.R.zpqr_create(eng.ctx, rank, world, uid, C.byref(self.comm))
        if rc != 0:
            raise ZpqError(rc, "zpqr_create (rank %d of %d)" % (rank, world))
        self.fn = C.cast(self.R.zpqr_allgatherv, ALLGATHERV)      # the C function itself goes to zpqj_add_sharded: no Python in between
        self.fn_dev = C.cast(self.R.zpqr_allgatherv_dev, ALLGATHERV)   # ... and its device-memory form (same signature, HBM pointers)

    def gather_dev(self, d_send, n):
        """zpqr_allgatherv_dev from Python: device pointer + length -> [(device pointer, length) of every rank]"""
        recv = (C.c_void_p * self.world)()
        rlen = (C.c_size_t * self.world)()
        rc = self.R.zpqr_allgatherv_dev(self.comm, C.c_void_p(d_send), n, recv, rlen)
        if rc != 0:
            raise ZpqError(rc, "zpqr_allgatherv_dev: " + self.R.zpqr_last_error(self.comm).decode())
        return [(recv[r] or 0, rlen[r]) for r in range(self.world)]

    def __call__(self, b):
        """the same collective from Python: bytes -> [bytes of every rank]"""
        recv = (C.c_void_p * self.world)()
        rlen = (C.c_size_t * self.world)()
        buf = C.create_string_buffer(bytes(b), max(1, len(b)))
        rc = self.R.zpqr_allgatherv(self.comm, buf, len(b), recv, rlen)
        if rc != 0:
            raise ZpqError(rc, "zpqr_allgatherv: " + self.R.zpqr_last_error(self.comm).decode())
        return [C.string_at(recv[r], rlen[r]) if rlen[r] else b"" for r in range(self.world)]

    def close(self):
        if self.comm:
            self.R.zpqr_destroy(self.comm)
            self.comm = C.c_void_p()


def jidac_shard_files(names, sizes, world, rank):
    """Which files rank `rank` of `world` supplies to jidac_add_sharded: a list of bools in the order given."""
    S = load_shim()
    n = len(names)
    nm = (C.c_char_p * max(1, n))(*[x.encode() for x in names])
    sz = (C.c_uint64 * max(1, n))(*sizes)
    mine = (C.c_uint8 * max(1, n))()
    rc = S.zpqj_shard_files(nm, sz, n, world, rank, mine)
    if rc != 0:
        raise ZpqError(rc, "zpqj_shard_files")
    return [bool(mine[i]) for i in range(n)]


def jidac_add_sharded(eng, rank, world, allgather, archive, files, version_date, method="14", dates=None, checksums=False, hint=False):
    """zpqj_add_sharded: one PROCESS per GPU.  files: list of (name, bytes-or-None, size) -- bytes only for the files
    jidac_shard_files marks for this rank.  allgather: f(bytes) -> [bytes of every rank] (e.g. dist_allgather_bytes()).
    Every rank returns the same (archive bytes, stats)."""
    S = load_shim()
    n = len(files)
    names = (C.c_char_p * max(1, n))(*[f[0].encode() for f in files])
    keep = [C.create_string_buffer(bytes(f[1]), max(1, len(f[1]))) if f[1] is not None else None for f in files]
    datas = (C.c_void_p * max(1, n))(*[C.cast(k, C.c_void_p).value if k is not None else None for k in keep])
    sizes = (C.c_uint64 * max(1, n))(*[f[2] for f in files])
    dts = (C.c_int64 * max(1, n))(*(dates or [version_date] * n))
    out, out_len = C.c_void_p(), C.c_size_t(0)
    stats = (C.c_uint64 * 6)()
    held, err = [], []

    def cb(user, send, send_len, recv, recv_len):
        try:
            got = allgather(C.string_at(send, send_len) if send_len else b"")
            del held[:]
            for r in range(world):
                buf = C.create_string_buffer(got[r], max(1, len(got[r])))
                held.append(buf)
                recv[r] = C.cast(buf, C.c_void_p).value
                recv_len[r] = len(got[r])
            return 0
        except Exception as e:          # an exception cannot cross the C frames: report it after the call
            err.append(e)
            return 1
    native = isinstance(allgather, RcclGather)          # the in-tree RCCL collective: zpqj_add_sharded calls the C function directly
    rc = S.zpqj_add_sharded(eng.ctx, rank, world, allgather.fn if native else ALLGATHERV(cb), allgather.comm if native else None,
                            bytes(archive) if archive else None, len(archive) if archive else 0,
                            names, datas, sizes, dts, n, version_date, method.encode(), (1 if checksums else 0) | (2 if hint else 0),
                            C.byref(out), C.byref(out_len), stats)
    if err:
        raise err[0]
    if rc != 0:
        raise ZpqError(rc, "%s (%s)" % (eng.L.zpq_strerror(rc).decode(), eng.L.zpq_last_error(eng.ctx).decode()))
    data = C.string_at(out.value, out_len.value)
    S.zpqj_free(out)
    keys = ("fragments", "new_fragments", "d_blocks", "unique_bytes", "d_bytes", "bytes_written")
    return data, dict(zip(keys, [int(x) for x in stats]))


def jidac_add_sharded_dev(eng, rank, world, allgather, allgather_dev, archive, names, sizes, d_base, version_date, method="14", dates=None,
                          checksums=False, hint=False, twins=True, wrap=None):
    """zpqj_add_sharded_dev: one PROCESS per GPU, this rank's files (jidac_shard_files) already in HBM back to back at d_base.
    names / sizes: ALL files of the batch (names ascending), the same on every rank.  allgather: f(bytes) -> [bytes of every
    rank], or an RcclGather (then the C collectives are called directly: no Python between the engine and RCCL);
    allgather_dev: None, or f(device pointer, length) -> [(device pointer, length) of every rank] (ignored with an RcclGather:
    its own device form is used).  wrap: optional f(section, thunk) -> thunk(): every collective of the call goes through it
    (bench.py keeps several adds in flight and orders their collectives with it).  Every rank returns (archive bytes, stats)."""
    S = load_shim()
    n = len(names)
    c_names = (C.c_char_p * max(1, n))(*[x.encode() for x in names])
    c_sizes = (C.c_uint64 * max(1, n))(*sizes)
    dts = (C.c_int64 * max(1, n))(*(dates or [version_date] * n))
    out, out_len = C.c_void_p(), C.c_size_t(0)
    stats = (C.c_uint64 * 6)()
    held, err, sec = [], [], [0]
    native = isinstance(allgather, RcclGather)

    def run(thunk):
        k = sec[0]; sec[0] += 1
        return wrap(k, thunk) if wrap else thunk()

    def cb(user, send, send_len, recv, recv_len):
        try:
            if native:
                return run(lambda: allgather.R.zpqr_allgatherv(allgather.comm, send, send_len, recv, recv_len))
            got = run(lambda: allgather(C.string_at(send, send_len) if send_len else b""))
            del held[:]
            for r in range(world):
                buf = C.create_string_buffer(got[r], max(1, len(got[r])))
                held.append(buf)
                recv[r] = C.cast(buf, C.c_void_p).value
                recv_len[r] = len(got[r])
            return 0
        except Exception as e:          # an exception cannot cross the C frames: report it after the call
            err.append(e)
            return 1

    def cb_dev(user, send, send_len, recv, recv_len):
        try:
            if native:
                return run(lambda: allgather.R.zpqr_allgatherv_dev(allgather.comm, send, send_len, recv, recv_len))
            got = run(lambda: allgather_dev(send or 0, send_len))
            for r in range(world):
                recv[r] = got[r][0] or None
                recv_len[r] = got[r][1]
            return 0
        except Exception as e:
            err.append(e)
            return 1
    direct = native and wrap is None          # nothing to order: the C functions themselves
    fn = allgather.fn if direct else ALLGATHERV(cb)
    fn_dev = allgather.fn_dev if direct else (ALLGATHERV(cb_dev) if (native or allgather_dev is not None) else C.cast(None, ALLGATHERV))
    flags = (1 if checksums else 0) | (2 if hint else 0) | (0 if twins else 4)
    rc = S.zpqj_add_sharded_dev(eng.ctx, rank, world, fn, fn_dev, allgather.comm if direct else None,
                                bytes(archive) if archive else None, len(archive) if archive else 0,
                                c_names, C.c_void_p(d_base), c_sizes, dts, n, version_date, method.encode(), flags,
                                C.byref(out), C.byref(out_len), stats)
    if err:
        raise err[0]
    if rc != 0:
        raise ZpqError(rc, "%s (%s)" % (eng.L.zpq_strerror(rc).decode(), eng.L.zpq_last_error(eng.ctx).decode()))
    data = C.string_at(out.value, out_len.value)
    S.zpqj_free(out)
    keys = ("fragments", "new_fragments", "d_blocks", "unique_bytes", "d_bytes", "bytes_written")
    return data, dict(zip(keys, [int(x) for x in stats]))


def jidac_add(eng, archive, files, version_date, method="14", dates=None, checksums=False, hint=False):
    """files: list of (name, bytes).  Returns (bytes to append to the archive, stats dict).  `eng` may be a list of
    engines (one per GPU): the files are then sharded across them (zpqj_add_multi), with identical output."""
    S = load_shim()
    engs = list(eng) if isinstance(eng, (list, tuple)) else None
    if engs:
        eng = engs[0]
    n = len(files)
    names = (C.c_char_p * max(1, n))(*[f[0].encode() for f in files])
    keep = [C.create_string_buffer(bytes(f[1]), max(1, len(f[1]))) for f in files]
    datas = (C.c_void_p * max(1, n))(*[C.cast(k, C.c_void_p).value for k in keep])
    sizes = (C.c_uint64 * max(1, n))(*[len(f[1]) for f in files])
    dts = (C.c_int64 * max(1, n))(*(dates or [version_date] * n))
    out, out_len = C.c_void_p(), C.c_size_t(0)
    stats = (C.c_uint64 * 6)()
    if checksums or hint:
        es = engs or [eng]
        ctxs = (C.c_void_p * len(es))(*[e.ctx.value for e in es])
        rc = S.zpqj_add_opts(ctxs, len(es), bytes(archive) if archive else None, len(archive) if archive else 0, names, datas, sizes, dts, n,
                             version_date, method.encode(), (1 if checksums else 0) | (2 if hint else 0), C.byref(out), C.byref(out_len), stats)
    elif engs:
        ctxs = (C.c_void_p * len(engs))(*[e.ctx.value for e in engs])
        rc = S.zpqj_add_multi(ctxs, len(engs), bytes(archive) if archive else None, len(archive) if archive else 0, names, datas, sizes, dts, n,
                              version_date, method.encode(), C.byref(out), C.byref(out_len), stats)
    else:
        rc = S.zpqj_add(eng.ctx, bytes(archive) if archive else None, len(archive) if archive else 0, names, datas, sizes, dts, n,
                        version_date, method.encode(), C.byref(out), C.byref(out_len), stats)
    if rc != 0:
        raise ZpqError(rc, "%s (%s)" % (eng.L.zpq_strerror(rc).decode(), eng.L.zpq_last_error(eng.ctx).decode()))
    data = C.string_at(out.value, out_len.value)
    S.zpqj_free(out)
    keys = ("fragments", "new_fragments", "d_blocks", "unique_bytes", "d_bytes", "bytes_written")
    return data, dict(zip(keys, [int(x) for x in stats]))


class DevFiles:
    """The per-call tables of jidac_add_dev, built once for a set of files that stays in HBM (names, offsets, dates)."""

    def __init__(self, names, file_off, dates=None, version_date=0):
        n = len(names)
        assert len(file_off) == n + 1
        self.n = n
        self.names = (C.c_char_p * max(1, n))(*[x.encode() for x in names])
        self.off = (C.c_uint64 * (n + 1))(*file_off)
        self.dates = (C.c_int64 * max(1, n))(*(dates or [version_date] * n))


def jidac_add_dev(eng, archive, d_base, files, version_date, method="14", checksums=False, hint=False, twins=True, raw=False):
    """zpqj_add_dev: the files are already in HBM -- `files` is a DevFiles (names ascending, file k at d_base + off[k] .. off[k + 1],
    back to back).  Returns (bytes to append to the archive, stats dict); raw=True: (address, length, stats), the caller frees the
    address with load_shim().zpqj_free (no copy of the archive into a Python bytes object)."""
    S = load_shim()
    out, out_len = C.c_void_p(), C.c_size_t(0)
    stats = (C.c_uint64 * 6)()
    flags = (1 if checksums else 0) | (2 if hint else 0) | (0 if twins else 4)
    rc = S.zpqj_add_dev(eng.ctx, bytes(archive) if archive else None, len(archive) if archive else 0, files.names, C.c_void_p(d_base), files.off,
                        files.dates, files.n, version_date, method.encode(), flags, C.byref(out), C.byref(out_len), stats)
    if rc != 0:
        raise ZpqError(rc, "%s (%s)" % (eng.L.zpq_strerror(rc).decode(), eng.L.zpq_last_error(eng.ctx).decode()))
    keys = ("fragments", "new_fragments", "d_blocks", "unique_bytes", "d_bytes", "bytes_written")
    st = dict(zip(keys, [int(x) for x in stats]))
    if raw:
        return out.value, out_len.value, st
    data = C.string_at(out.value, out_len.value)
    S.zpqj_free(out)
    return data, st


def jidac_verify(eng, archive):
    """zpaqfranz t: (status, stats dict).  status 0 = every block, fragment and stored file checksum verified."""
    S = load_shim()
    st = (C.c_uint64 * 7)()
    rc = S.zpqj_verify(eng.ctx, bytes(archive), len(archive), st)
    keys = ("files", "fragments", "bytes", "files_with_checksums", "xxh64_mismatches", "crc32_mismatches", "d_blocks")
    return rc, dict(zip(keys, [int(x) for x in st]))


def jidac_extract_dev(eng, d_archive, archive_len, d_out=0, out_cap=0, d_sha256=0, sha256_cap=0, sha256=True, twins=False):
    """zpqj_extract_dev: archive resident in HBM in, restored files left in HBM (d_out) + their SHA-256 (d_sha256); returns
    (names, file_off, stats).  d_out = 0: plan only (the index: names and offsets)."""
    S = load_shim()
    fo, names, n = C.c_void_p(), C.c_void_p(), C.c_size_t(0)
    st = (C.c_uint64 * 7)()
    flags = (1 if sha256 and d_sha256 else 0) | (2 if twins else 0)
    rc = S.zpqj_extract_dev(eng.ctx, C.c_void_p(d_archive), archive_len, C.c_void_p(d_out) if d_out else None, out_cap,
                            C.c_void_p(d_sha256) if d_sha256 else None, sha256_cap, flags, C.byref(fo), C.byref(names), C.byref(n), st)
    if rc != 0:
        raise ZpqError(rc, "%s (%s)" % (eng.L.zpq_strerror(rc).decode(), eng.L.zpq_last_error(eng.ctx).decode()))
    off = list((C.c_uint64 * (n.value + 1)).from_address(fo.value))
    nm, p = [], names.value
    for _ in range(n.value):
        s_ = C.string_at(p)
        p += len(s_) + 1
        nm.append(s_.decode())
    S.zpqj_free(fo); S.zpqj_free(names)
    keys = ("files", "fragments", "bytes", "twins", "twin_bytes", "compared_bytes", "d_blocks")
    return nm, off, {k: int(v) for k, v in zip(keys, st)}


def jidac_extract(eng, archive):
    """Returns {name: bytes} of the latest version of every file."""
    S = load_shim()
    data, sizes, names, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_size_t(0)
    rc = S.zpqj_extract(eng.ctx, bytes(archive), len(archive), C.byref(data), C.byref(sizes), C.byref(names), C.byref(n))
    if rc != 0:
        raise ZpqError(rc, "%s (%s)" % (eng.L.zpq_strerror(rc).decode(), eng.L.zpq_last_error(eng.ctx).decode()))
    sz = list((C.c_uint64 * n.value).from_address(sizes.value)) if n.value else []
    blob = C.string_at(data.value, sum(sz))
    out, off, p = {}, 0, names.value
    for k in range(n.value):
        nm = C.string_at(p)
        p += len(nm) + 1
        out[nm.decode()] = blob[off:off + sz[k]]
        off += sz[k]
    for q in (data, sizes, names):
        S.zpqj_free(q)
    return out
