"""Rows a5 / f4: compressBlock at level 3 and the explicit x,2 / x,3 / x,4 methods -- byte-aligned LZ77 over the suffix
array, the Burrows-Wheeler transform and E8E9 in front of a context model.  Their post-processor programs have no byte
fixture in the reference tree; they are pinned by DECODE parity: the REAL reference Decompresser (oracle/_ref,
ZSFX/libzpaq.cpp:2239-2366 with its PostProcessor / ZPAQL machine) restores every block this engine writes, stored SHA-1
verified; the pre-processed stream inside equals the REAL LZBuffer's (:6140-6552); and the engine's own decode side
(native level-2 decoder, inverse BWT by list ranking, E8E9 inverse, or the translated program) gives the input back."""
import os

import numpy as np
import pytest

import datagen
import orc

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref/libzpaqref.so not available")


@pytest.fixture(scope="module")
def eng():
    from zpaqfranz_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def exe_like(n, seed):
    a = bytearray(datagen.binary_like(n, seed))
    rng = np.random.default_rng(seed)
    for p in rng.integers(0, n - 8, n // 40):
        a[p] = 0xE8 if p & 2 else 0xE9
        a[p + 4] = 0 if p & 1 else 0xFF
    return bytes(a)


METHODS = ["34", "34,200,1", "34,170,2", "34,30,0", "x4,2,12,0,7,25,1c0,0,511i2", "x4,6,12,0,7,25,1c0,0,511i2", "x4,3ci1", "x4,7ci1",
           "x4,4ci1,1,1,1,2a", "44,160,2", "x0,3ci1", "x6,3ci1", "x5,7ci1",          # (x5,7: BWT + E8E9 above 16 MiB -- the streaming E8E9 stage of round 6)
           "x4,2,8,0,3,22,0c0,0,511i2", "x4,6,6,0,2,20,0c0,0,511", "x4,2,5,0,3,24,0"]     # byte-aligned codes from the HASH-TABLE finder (round 4)


def _coded_payload(framed):
    k = framed.index(b"zPQ"); p = k + 5
    header = framed[p: p + 2 + (framed[p] | framed[p + 1] << 8)]
    p += len(header) + 1
    p = framed.index(b"\0", p) + 1; p = framed.index(b"\0", p) + 1; p += 1
    return header, framed[p: len(framed) - 22]


@needs_ref
@pytest.mark.parametrize("method", METHODS)
def test_reference_decompresser_restores_and_stream_equals_lzbuffer(eng, method):
    from zpaqfranz_amd import engine as E
    n = 120000
    exe = ",2" in method[2:] and method[0] in "34" or method.startswith(("x4,6", "x4,7", "x4,4", "x5,7"))
    blocks = [exe_like(n, 3) if exe else datagen.text_like(n, 4), datagen.mixed(n // 2, 5), b"", b"q"]
    res = eng.compress_blocks(blocks, [method] * len(blocks), ["f%d" % i for i in range(len(blocks))], ["c"] * len(blocks), True)
    for b, (st, framed) in zip(blocks, res):
        assert st == 0
        r = orc.ref_decompress_block(framed, len(b) + 64)                 # the REAL Decompresser + PostProcessor
        assert r["data"] == b and r["sha1_ok"] == 1, (method, len(b))
        xm = E.expand_method(method, b)
        src, args = E.make_config(xm)
        header, coded = _coded_payload(framed)
        if header[6]:                                                     # context model in front: the Encoder's input
            seen = orc.ref_cm_decode(header, coded, len(b) * 2 + 4096)
            lvl = args[1] & 3
            if lvl in (2, 3):
                pre = 3 + (seen[1] | seen[2] << 8)
                assert seen[pre:] == orc.ref_lzbuffer(b, args), (method, len(b))
    back = eng.decompress_blocks([f for _, f in res], [len(b) + 64 for b in blocks])
    for b, r in zip(blocks, back):
        assert r["status"] == 0 and r["data"] == b and r["sha1"] == orc.sha1(b), method


def test_inverse_bwt_and_level2_decoder_on_larger_blocks(eng, monkeypatch):
    """1.25 MiB through both native decode kernels, and the same blocks through the translated ZPAQL programs
    (ZPQ_PCOMP_GENERIC=1): identical output."""
    data = datagen.text_like(1 << 20, 11) + datagen.binary_like(1 << 18, 12)
    for method in ("x4,3c0", "x4,2,8,0,7,25,1c0"):
        (st, framed), = eng.compress_blocks([data], [method], ["f"], None, True)
        assert st == 0
        r, = eng.decompress_blocks([framed], [len(data) + 64])
        assert r["status"] == 0 and r["data"] == data
    small = data[:200000]
    monkeypatch.setenv("ZPQ_PCOMP_GENERIC", "1")
    for method in ("x4,3c0", "x4,2,8,0,7,25,1c0", "x4,7c0"):
        (st, framed), = eng.compress_blocks([small], [method], ["f"], None, True)
        assert st == 0
        r, = eng.decompress_blocks([framed], [len(small) + 64])
        assert r["status"] == 0 and r["data"] == small, method


def test_hostile_bwt_streams_are_refused(eng):
    """A BWT stream whose list does not run through every row (damaged bytes / index) must give a format error, not a
    hang: the reference's program would walk a cycle forever."""
    data = datagen.text_like(50000, 13)
    (st, framed), = eng.compress_blocks([data], ["x0,3"], ["f"], None, False)
    assert st == 0
    bad = bytearray(framed)
    k = len(bad) // 2
    bad[k] ^= 0x55; bad[k + 1] ^= 0x33
    r, = eng.decompress_blocks([bytes(bad)], [len(data) + 64], verify=False)
    assert r["status"] in (-6, 0)
    if r["status"] == 0:
        assert len(r["data"]) == len(data)


def _o1_hits(b):
    o1 = [0] * 256; c1 = 0; hits = 0
    for c in b:
        hits += o1[c1] == c
        o1[c1] = c; c1 = c
    return hits


def test_fragment_statistics_and_method_hint(eng):
    """Row a4: the order-1 hit count of a fragment is the one the fragment loop defines (table
    reset per fragment) -- checked against the plain loop; with ZPQJ_METHOD_HINT every d block is compressed with
    "1B,R,t" derived from it, so text, x86-like and incompressible blocks get different LZ77 variants (E8E9, stored),
    and the archive still extracts (the REAL reference decoder walks it too)."""
    import ctypes as C
    import numpy as np
    from zpaqfranz_amd import engine as E
    frs = [datagen.text_like(30000, 1), exe_like(40000, 2), datagen.random_bytes(20000, 3), b"", b"a" * 5000, datagen.mixed(70001, 4)]
    blob = b"".join(frs)
    off = np.cumsum([0] + [len(f) for f in frs[:-1]]).astype(np.uint64)
    ln = np.array([len(f) for f in frs], dtype=np.uint32)
    d = eng.upload(blob); d_off = eng.upload(off.tobytes()); d_len = eng.upload(ln.tobytes()); d_st = eng.alloc(16 * len(frs))
    try:
        eng._ck(eng.L.zpq_fragment_stats_dev(eng.ctx, d.ptr, d_off.ptr, d_len.ptr, len(frs), d_st.ptr))
        eng.sync()
        st = np.frombuffer(d_st.download(16 * len(frs)), dtype=np.uint32).reshape(-1, 4)
    finally:
        for b in (d, d_off, d_len, d_st):
            b.free()
    assert st[:, 0].tolist() == [_o1_hits(f) for f in frs]
    assert st[:, 3].tolist() == [len(f) for f in frs]
    assert st[0, 1] == 1 and st[0, 2] == 0 and st[1, 2] == 1 and st[2, 1] == 0 and st[2, 2] == 0
    # an archive whose blocks differ in kind: 1 MiB blocks ("10"), hint on
    files = [("text.txt", datagen.text_like(900000, 5)), ("prog.exe", exe_like(900000, 6)), ("noise.bin", datagen.random_bytes(900000, 7))]
    arc, stats = E.jidac_add(eng, b"", files, 20240101000000, method="10", hint=True)
    assert E.jidac_extract(eng, arc) == dict(files)
    methods = set()
    pos = 0
    while pos < len(arc):
        r = orc.ref_decompress_block(arc[pos:], 1 << 21)
        if r["filename"][17:18] == b"d":
            k = arc.index(b"zPQ", pos) + 5
            methods.add(bytes(arc[k: k + 2 + (arc[k] | arc[k + 1] << 8)]))
        assert r["sha1_ok"] == 1
        pos += r["consumed"]
    assert len(methods) >= 2          # not one header for everything: the hint chose per block


def test_shim_decompresser_pcomp_behind_a_context_model(eng, tmp_path):
    """libzpaq::Decompresser::pcomp() (ZSFX/libzpaq.h:1254): for a block whose post-processor section sits inside the
    arithmetic-coded stream the shim decodes the head of that stream on the device -- the section it hands back must be the
    program compressBlock put there (two size bytes in front); a block coded by a model alone has none."""
    import subprocess
    from zpaqfranz_amd import build, engine as E
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    here = build.HERE                      # (the product's directory; the emulated engine's when the suite runs on the CPU)
    drv = str(tmp_path / "pcomp_driver")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(root, "zpaqfranz_amd", "shim"), os.path.join(root, "tests", "cpp", "pcomp_driver.cpp"),
                           "-L" + here, "-lzpaq_gpu", "-lzpaqhip", "-Wl,-rpath," + here, "-o", drv])
    data = datagen.text_like(60000, 9)
    methods = ["x4,6,12,0,7,25,1c0,0,511i2", "x4,3ci1", "x4,0ci1", "x4,2,8,0,3,22,0c0,0,511i2"]
    res = eng.compress_blocks([data] * len(methods), methods, ["f%d" % i for i in range(len(methods))], ["c"] * len(methods), True)
    arc = tmp_path / "m.zpaq"
    arc.write_bytes(b"".join(f for st, f in res))
    assert all(st == 0 for st, _ in res)
    r = subprocess.run([drv, str(arc)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [l.split("|") for l in r.stdout.strip().splitlines()]
    assert [l[0] for l in lines] == ["f%d" % i for i in range(len(methods))]
    for m, l in zip(methods, lines):
        src, args = E.make_config(E.expand_method(m, data))
        _, pc = E.compile_config(src, args)
        want = (bytes([len(pc) & 255, len(pc) >> 8]) + pc).hex() if pc else ""
        assert l[1] == want, (m, l[1][:40], want[:40])
