"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when it has been built, the real
reference compiled in place (oracle/_ref/libzpaqref.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _build():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("zpaq_oracle.cpp", "checksum_oracle.cpp")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "liboracle.so"])
    return so


_L = C.CDLL(_build())
_u8p = C.POINTER(C.c_ubyte)
for name in ("orc_suffix_array", "orc_bwt_encode", "orc_lz77_sa_encode", "orc_lz77_sa_decisions", "orc_chunk", "orc_lz77_encode", "orc_lz77_decode", "orc_compress_block", "orc_decompress_block", "orc_fragment_and_hash"):
    getattr(_L, name).restype = C.c_long


def _buf(b):
    return (C.c_ubyte * max(1, len(b))).from_buffer_copy(bytes(b) + (b"\0" if len(b) == 0 else b""))


def sha1(b):
    out = (C.c_ubyte * 20)()
    _L.orc_sha1(_buf(b), C.c_long(len(b)), out)
    return bytes(out)


def sha256(b):
    out = (C.c_ubyte * 32)()
    _L.orc_sha256(_buf(b), C.c_long(len(b)), out)
    return bytes(out)


_L.orc_crc32.restype = C.c_uint32
_L.orc_xxh64.restype = C.c_uint64
_L.orc_xxh64.argtypes = [C.c_void_p, C.c_long, C.c_uint64]


def crc32(b):
    return int(_L.orc_crc32(_buf(b), C.c_long(len(b))))


def xxh64(b, seed=0):
    return int(_L.orc_xxh64(C.cast(_buf(b), C.c_void_p), len(b), seed))


def blake3(b):
    out = (C.c_ubyte * 32)()
    _L.orc_blake3(_buf(b), C.c_long(len(b)), out)
    return bytes(out)


def chunk(b, fragment=6, min_frag=4096, max_frag=520192):
    cap = len(b) // max(1, min_frag) + 2
    lens = (C.c_uint32 * cap)()
    n = _L.orc_chunk(_buf(b), C.c_long(len(b)), fragment, C.c_uint32(min_frag), C.c_uint32(max_frag), lens, C.c_long(cap))
    assert n <= cap
    return list(lens[:n])


def fragment_and_hash_view(mem, offset, n, fragment=6, min_frag=4096, max_frag=520192):
    """Fragment + SHA-1 of mem[offset:offset+n] where mem is a ctypes array created once by the caller
    (no per-call copy; the C call releases the GIL)."""
    x = C.c_uint64(0)
    nf = _L.orc_fragment_and_hash(C.byref(mem, offset), C.c_long(n), fragment, C.c_uint32(min_frag), C.c_uint32(max_frag), C.byref(x))
    return nf, x.value


def compress_block_view(mem, offset, n, method="14"):
    """compressBlock of mem[offset:offset+n] without copying the input (output size only)."""
    cap = n + n // 8 + 4096
    out = (C.c_ubyte * cap)()
    args = (C.c_int * 9)()
    r = _L.orc_compress_block(C.byref(mem, offset), C.c_long(n), method.encode(), b"jDC20240101000000d0000000001", b"jDC\x01", 1, out, C.c_long(cap), args)
    if r < 0:
        raise RuntimeError("orc_compress_block failed: %d" % r)
    return r


def e8e9(b):
    x = _buf(b)
    _L.orc_e8e9(x, C.c_long(len(b)))
    return bytes(x)[: len(b)]


def e8e9_inverse(b):
    x = _buf(b)
    _L.orc_e8e9_inverse(x, C.c_long(len(b)))
    return bytes(x)[: len(b)]


def lz77_encode(b, args, trace=False):
    a = (C.c_int * 9)(*(list(args) + [0] * 9)[:9])
    cap = len(b) + len(b) // 8 + 1024
    out = (C.c_ubyte * cap)()
    tcap = len(b) // 4 + 16 if trace else 0
    tr = (C.c_uint32 * (3 * tcap))() if trace else None
    nt = C.c_long(0)
    r = _L.orc_lz77_encode(_buf(b), C.c_long(len(b)), a, out, C.c_long(cap), tr, C.c_long(tcap), C.byref(nt))
    if r < 0:
        raise RuntimeError("orc_lz77_encode failed: %d" % r)
    if trace:
        t = [(tr[3 * i], tr[3 * i + 1], tr[3 * i + 2]) for i in range(nt.value)]
        return bytes(out[:r]), t
    return bytes(out[:r])


def suffix_array(b):
    import numpy as np
    sa = np.zeros(max(1, len(b)), dtype=np.uint32)
    _L.orc_suffix_array(_buf(b), C.c_long(len(b)), sa.ctypes.data_as(C.POINTER(C.c_uint32)))
    return sa[: len(b)]


def lz77_sa_encode(b, args, sa=None, trace=False):
    """LZ77 with the suffix-array match finder (levels 1 and 2); `sa` (numpy uint32) is optional."""
    a = (C.c_int * 9)(*(list(args) + [0] * 9)[:9])
    cap = len(b) + len(b) // 8 + 1024
    out = (C.c_ubyte * cap)()
    tcap = len(b) + 16 if trace else 0
    tr = (C.c_uint32 * (3 * tcap))() if trace else None
    nt = C.c_long(0)
    sap = sa.ctypes.data_as(C.POINTER(C.c_uint32)) if sa is not None else None
    r = _L.orc_lz77_sa_encode(_buf(b), C.c_long(len(b)), a, sap, out, C.c_long(cap), tr, C.c_long(tcap), C.byref(nt))
    if r < 0:
        raise RuntimeError("orc_lz77_sa_encode failed: %d" % r)
    if trace:
        return bytes(out[:r]), [(tr[3 * i], tr[3 * i + 1], tr[3 * i + 2]) for i in range(nt.value)]
    return bytes(out[:r])


def bwt_encode(b):
    out = (C.c_ubyte * (len(b) + 5))()
    r = _L.orc_bwt_encode(_buf(b), C.c_long(len(b)), None, out, C.c_long(len(b) + 5))
    if r < 0:
        raise RuntimeError("orc_bwt_encode failed: %d" % r)
    return bytes(out[:r])


def lz77_sa_decisions(b, args, sa):
    """numpy uint64[2n]: the decision at every position for lit == 0 / lit > 0 (0 = literal, else blen | blit<<16 | off<<32)."""
    import numpy as np
    a = (C.c_int * 9)(*(list(args) + [0] * 9)[:9])
    rec = np.zeros(max(1, 2 * len(b)), dtype=np.uint64)
    r = _L.orc_lz77_sa_decisions(_buf(b), C.c_long(len(b)), a, sa.ctypes.data_as(C.POINTER(C.c_uint32)), rec.ctypes.data_as(C.POINTER(C.c_uint64)))
    if r < 0:
        raise RuntimeError("orc_lz77_sa_decisions failed: %d" % r)
    return rec[: 2 * len(b)]


def lz77_decode(b, cap, rb=0):
    out = (C.c_ubyte * max(1, cap))()
    r = _L.orc_lz77_decode(_buf(b), C.c_long(len(b)), rb, out, C.c_long(cap))
    if r < 0:
        raise RuntimeError("orc_lz77_decode failed: %d" % r)
    return bytes(out[:r])


def compress_block(b, method, filename=None, comment=None, dosha1=True):
    cap = len(b) + len(b) // 8 + 4096
    out = (C.c_ubyte * cap)()
    args = (C.c_int * 9)()
    r = _L.orc_compress_block(_buf(b), C.c_long(len(b)), method.encode(), filename.encode() if filename is not None else None,
                              comment if comment is None else (comment if isinstance(comment, bytes) else comment.encode()),
                              int(dosha1), out, C.c_long(cap), args)
    if r < 0:
        raise RuntimeError("orc_compress_block failed: %d" % r)
    return bytes(out[:r]), list(args)


def decompress_block(arc, cap):
    out = (C.c_ubyte * max(1, cap))()
    meta = (C.c_long * 3)()
    r = _L.orc_decompress_block(_buf(arc), C.c_long(len(arc)), out, C.c_long(cap), meta)
    if r < 0:
        raise RuntimeError("orc_decompress_block failed: %d" % r)
    return bytes(out[:r]), list(meta)


# ---------------------------------------------------------------------------------------------
# the real reference (optional)
# ---------------------------------------------------------------------------------------------
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libzpaqref.so")
_R = None
if os.path.exists(REF_SO):
    _R = C.CDLL(REF_SO)
    for name in ("ref_divsufsort", "ref_fragment_sha1", "ref_lz1_block_cost", "ref_lzbuffer", "ref_decompress", "ref_decompress_block", "ref_compile", "ref_postprocess", "ref_cm_encode", "ref_cm_decode", "ref_tables"):
        getattr(_R, name).restype = C.c_long
    _R.ref_last_error.restype = C.c_char_p


def have_ref():
    return _R is not None


def ref_sha1(b):
    out = (C.c_ubyte * 20)()
    _R.ref_sha1(_buf(b), C.c_long(len(b)), out)
    return bytes(out)


def ref_sha256(b):
    out = (C.c_ubyte * 32)()
    _R.ref_sha256(_buf(b), C.c_long(len(b)), out)
    return bytes(out)


def ref_e8e9(b):
    x = _buf(b)
    _R.ref_e8e9(x, C.c_int(len(b)))
    return bytes(x)[: len(b)]


def ref_divsufsort(b):
    import numpy as np
    sa = np.zeros(max(1, len(b)), dtype=np.int32)
    r = _R.ref_divsufsort(_buf(b), C.c_long(len(b)), sa.ctypes.data_as(C.POINTER(C.c_int)))
    if r < 0:
        raise RuntimeError("ref_divsufsort: %s" % _R.ref_last_error())
    return sa[: len(b)].astype(np.uint32)


def ref_lzbuffer(b, args):
    a = (C.c_int * 9)(*(list(args) + [0] * 9)[:9])
    cap = len(b) + len(b) // 8 + 1024
    out = (C.c_ubyte * cap)()
    r = _R.ref_lzbuffer(_buf(b), C.c_long(len(b)), a, out, C.c_long(cap))
    if r < 0:
        raise RuntimeError("ref_lzbuffer: %s" % _R.ref_last_error())
    return bytes(out[:r])


def ref_decompress(arc, cap):
    out = (C.c_ubyte * max(1, cap))()
    r = _R.ref_decompress(_buf(arc), C.c_long(len(arc)), out, C.c_long(cap))
    if r < 0:
        raise RuntimeError("ref_decompress: %s" % _R.ref_last_error())
    return bytes(out[:r])


def ref_decompress_block(arc, cap):
    out = (C.c_ubyte * max(1, cap))()
    meta = (C.c_long * 5)()
    fn = C.create_string_buffer(4096)
    cm = C.create_string_buffer(4096)
    sh = (C.c_ubyte * 21)()
    r = _R.ref_decompress_block(_buf(arc), C.c_long(len(arc)), out, C.c_long(cap), meta, fn, C.c_long(4096), cm, C.c_long(4096), sh)
    if r < 0:
        raise RuntimeError("ref_decompress_block: %s" % _R.ref_last_error())
    return dict(data=bytes(out[:r]), consumed=meta[0], segments=meta[1], sha1_ok=meta[2],
                filename=fn.raw[: meta[3]], comment=cm.raw[: meta[4]], sha1=bytes(sh))


def ref_compile(config, args):
    a = (C.c_int * 9)(*(list(args) + [0] * 9)[:9])
    h = (C.c_ubyte * 70000)()
    p = (C.c_ubyte * 70000)()
    hl, pl = C.c_long(0), C.c_long(0)
    r = _R.ref_compile(config.encode(), a, h, C.c_long(70000), C.byref(hl), p, C.c_long(70000), C.byref(pl))
    if r < 0:
        raise RuntimeError("ref_compile: %s" % _R.ref_last_error())
    return bytes(h[: hl.value]), bytes(p[: pl.value])


def ref_postprocess(stream, ph, pm, cap):
    out = (C.c_ubyte * max(1, cap))()
    r = _R.ref_postprocess(_buf(stream), C.c_long(len(stream)), ph, pm, out, C.c_long(cap))
    if r < 0:
        raise RuntimeError("ref_postprocess: %s" % _R.ref_last_error())
    return bytes(out[:r])


def ref_cm_encode(header, data):
    cap = len(data) + len(data) // 2 + 4096
    out = (C.c_ubyte * cap)()
    r = _R.ref_cm_encode(_buf(header), C.c_long(len(header)), _buf(data), C.c_long(len(data)), out, C.c_long(cap))
    if r < 0:
        raise RuntimeError("ref_cm_encode: %s" % _R.ref_last_error())
    return bytes(out[:r])


def ref_cm_decode(header, coded, cap):
    out = (C.c_ubyte * max(1, cap))()
    r = _R.ref_cm_decode(_buf(header), C.c_long(len(header)), _buf(coded), C.c_long(len(coded)), out, C.c_long(cap))
    if r < 0:
        raise RuntimeError("ref_cm_decode: %s" % _R.ref_last_error())
    return bytes(out[:r])


def ref_cm_encode_segments(header, segments):
    """the real Predictor over several segments of one block -> the coded stream of each"""
    data = b"".join(segments)
    k = len(segments)
    cap = len(data) + len(data) // 2 + 4096 + 16 * k
    out = (C.c_ubyte * cap)()
    lens = (C.c_uint * k)(*[len(x) for x in segments]); ends = (C.c_uint * k)()
    r = _R.ref_cm_encode_segments(_buf(header), C.c_long(len(header)), _buf(data), lens, C.c_long(k), out, C.c_long(cap), ends)
    if r < 0:
        raise RuntimeError("ref_cm_encode_segments: %s" % _R.ref_last_error())
    cuts = [0] + [int(e) for e in ends]
    return [bytes(out[cuts[i]:cuts[i + 1]]) for i in range(k)]


def ref_cm_decode_segments(header, coded_segments, cap):
    """the real Decoder (initialised once) over the coded streams of a block's segments -> the bytes of each"""
    coded = b"".join(coded_segments)
    k = len(coded_segments)
    out = (C.c_ubyte * max(1, cap))()
    ends = (C.c_uint * k)()
    r = _R.ref_cm_decode_segments(_buf(header), C.c_long(len(header)), _buf(coded), C.c_long(len(coded)), C.c_long(k), out, C.c_long(cap), ends)
    if r < 0:
        raise RuntimeError("ref_cm_decode_segments: %s" % _R.ref_last_error())
    cuts = [0] + [int(e) for e in ends]
    return [bytes(out[cuts[i]:cuts[i + 1]]) for i in range(k)]


def ref_tables():
    sq = (C.c_uint16 * 4096)(); st = (C.c_int16 * 32768)(); dt = (C.c_int * 1024)(); d2 = (C.c_int * 256)(); ns = (C.c_ubyte * 1024)()
    if _R.ref_tables(sq, st, dt, d2, ns) != 0:
        raise RuntimeError("ref_tables: %s" % _R.ref_last_error())
    return list(sq), list(st), list(dt), list(d2), bytes(ns)
