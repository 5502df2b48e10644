"""libzpaq::Decompresser::pcomp() of the shim (ZSFX/libzpaq.h:1254): for the reference's own fixture the three LZ77-coded
i blocks must hand back the 302-byte level-1 post-processor program the reference wrote (two size bytes in front), the
stored c / h blocks nothing; the d block is coded behind a context model, where the head of the stream has to be decoded on the
device (here, without one: an error, not a wrong answer; tests/test_gpu_m3.py asks the chip).  Host-side parsing only: runs
without a GPU."""
import ctypes as C
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_pcomp_of_the_fixture_blocks(tmp_path):
    from zpaqfranz_amd import build, engine
    build.build(verbose=False)
    here = os.path.join(ROOT, "zpaqfranz_amd")
    drv = str(tmp_path / "pcomp_driver")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(here, "shim"), os.path.join(ROOT, "tests", "cpp", "pcomp_driver.cpp"),
                           "-L" + here, "-lzpaq_gpu", "-lzpaqhip", "-Wl,-rpath," + here, "-o", drv])
    r = subprocess.run([drv, os.path.join(G, "sha256.zpaq")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [l.split("|") for l in r.stdout.strip().splitlines()]
    blocks = json.load(open(os.path.join(G, "blocks.json")))
    assert [l[0] for l in lines] == [b["filename"] for b in blocks]
    L = engine.load()
    L.zpq_known_pcomp_bytes.argtypes = [C.c_uint32, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    buf = C.create_string_buffer(1024); n = C.c_size_t(0)
    assert L.zpq_known_pcomp_bytes(0, 0, buf, 1024, C.byref(n)) == 0 and n.value == 302
    want = bytes([302 & 255, 302 >> 8]) + buf.raw[:302]
    kinds = {b["filename"][17]: l[1] for b, l in zip(blocks, lines)}            # jDC<14 digits><c|d|h|i><10 digits>
    assert bytes.fromhex(kinds["i"]) == want                                    # LZ77 level 1: the golden program
    assert kinds["c"] == "" and kinds["h"] == ""                                 # stored blocks: no PCOMP section
    assert kinds["d"] == "" or kinds["d"].startswith("error:")                   # behind a context model: needs the device (GPU test below the -m gpu marker)
