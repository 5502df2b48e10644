"""zpq_lz77_encode_dev() ITSELF on the CPU: tests/cpp/lz77_full_emu.cpp compiles zpaqfranz_amd/csrc/lz77_enc.hip whole -- host
code and every kernel -- over a stand-in HIP runtime (tests/cpp/fake_hip.h: device memory is host memory, a launch runs every
workgroup on the fibre emulator).  The code stream must be the oracle's, byte for byte, on every path the environment selects:
table states with speculative segments (the default; one and several segments per block, with the seam and stitch walks and the
token packers behind them) and one workgroup of four waves per block writing the stream itself.
The switches are read once per process, hence one process per setting."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang++ not found")

SCRIPT = r"""
import ctypes as C, sys
sys.path.insert(0, %(root)r + "/tests")
import numpy as np, datagen, orc
L = C.CDLL(%(so)r)
L.lz77_full_emu.restype = C.c_long
L.lz77_full_emu.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_int32), C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32]
scale = %(scale)d
inputs = {"text": datagen.text_like(6000 * scale, 1), "mixed": datagen.mixed(9000 * scale, 3), "runs": bytes(1500 * scale) + b"ab" * (900 * scale) + datagen.random_bytes(500, 4),
          "tiny": b"abcabcabcabc", "one": b"x", "empty": b""}
bad = 0
only = %(only)r
for args in %(argsets)r:
    for name, b in inputs.items():
        if only and name not in only:
            continue
        n = len(b); cap = (n + n // 8 + 1024 + 15) & ~15
        out = np.zeros(cap, dtype=np.uint8); err = C.create_string_buffer(400)
        r = L.lz77_full_emu(b + bytes(64), n, (C.c_int32 * 9)(*(args + [0] * 9)[:9]), out.ctypes.data, cap, err, 400)
        if r < 0 or bytes(out[:r]) != orc.lz77_encode(b, args):
            bad += 1
            print("MISMATCH", args, name, r, err.value.decode())
print("mismatches:", bad)
sys.exit(1 if bad else 0)
"""


@pytest.fixture(scope="module")
def so(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("full") / "lz77_full_emu.so")
    subprocess.check_call([CLANG, "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-Wno-unused", "-Wno-unused-value",
                           "-I" + os.path.join(ROOT, "zpaqfranz_amd", "csrc"), "-I" + os.path.join(ROOT, "tests", "cpp"), "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "lz77_full_emu.cpp"), "-o", path])
    return path


A3 = [[4, 1, 5, 0, 3, 15], [0, 1, 4, 0, 1, 14], [5, 1, 4, 0, 2, 15]]


SETTINGS = [
    ("default", {}, 2, A3[:2], None),                                                                 # table states, 2 MiB segments (one per block here)
    ("table-segments", {"ZPQ_LZ_SEG": "65536"}, 6, A3, None),                                         # table states, several speculative segments
    ("table-direct", {"ZPQ_LZ_DIRECT": "1"}, 4, A3, None),                                            # one workgroup per block writes the stream itself
]


def test_the_encoder_entry_point_on_the_cpu_gives_the_oracles_stream(so):
    """every setting in a process of its own (the switches are read once per process), all of them at the same time"""
    from concurrent.futures import ThreadPoolExecutor

    def one(setting):
        name, env, scale, argsets, only = setting
        e = {k: v for k, v in os.environ.items() if not k.startswith("ZPQ_")}
        e.update(env)
        r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "so": so, "scale": scale, "argsets": argsets, "only": only}], capture_output=True, text=True, env=e,
                           timeout=900)
        return name, r.returncode, r.stdout[-1500:], r.stderr[-1500:]
    with ThreadPoolExecutor(max_workers=min(len(SETTINGS), os.cpu_count() or 2)) as ex:
        results = list(ex.map(one, SETTINGS))
    bad = [r for r in results if r[1] != 0]
    assert not bad, bad
