"""The hand-written radix sort (zpaqfranz_amd/csrc/radix.hip, experimental: ZPQ_SORT=own) on the CPU: tests/cpp/radix_emu.cpp
compiles the kernels' DEVICE source for the host and runs every workgroup on the fibre emulator (256 threads, ballots,
shuffles, wave barriers and __syncthreads as rendezvous), pass by pass as zpq_radix_sort_pairs() launches them.  The result
must be numpy's stable sort on the selected key bits: order of the keys AND of the values (stability)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang++ not found")


@pytest.fixture(scope="module")
def radix(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("radix") / "radix_emu.so")
    subprocess.check_call([CLANG, "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-Wno-unused", "-I" + os.path.join(ROOT, "zpaqfranz_amd", "csrc"),
                           "-I" + os.path.join(ROOT, "tests", "cpp"), os.path.join(ROOT, "tests", "cpp", "radix_emu.cpp"), "-o", so])
    L = C.CDLL(so)
    L.radix_emu.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint32]

    def run(keys, b0, b1):
        n = len(keys)
        k, v = keys.copy(), np.arange(n, dtype=np.uint32)
        err = C.create_string_buffer(256)
        rc = L.radix_emu(k.ctypes.data, v.ctypes.data, n, b0, b1, err, 256)
        assert rc == 0, err.value.decode()
        return k, v
    return run


def _keys(n, kind, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.integers(0, 1 << 62, size=n, dtype=np.uint64)
    if kind == "few":                                   # seven distinct keys: long runs of equal digits, stability matters
        return rng.integers(0, 7, size=n, dtype=np.uint64) * np.uint64(0x0101010101)
    if kind == "sorted":
        return np.sort(rng.integers(0, 1 << 40, size=n, dtype=np.uint64))
    if kind == "reverse":
        return np.sort(rng.integers(0, 1 << 40, size=n, dtype=np.uint64))[::-1].copy()
    return (rng.zipf(1.3, size=n).astype(np.uint64) * np.uint64(2654435761)) & np.uint64((1 << 40) - 1)   # skewed


@pytest.mark.parametrize("n,b0,b1,kind", [(1, 0, 8, "uniform"), (63, 0, 8, "uniform"), (64, 0, 8, "few"), (4096, 0, 8, "uniform"), (4097, 0, 16, "uniform"),
                                          (10000, 3, 20, "few"), (20000, 0, 24, "zipf"), (9000, 8, 8, "uniform"), (16384, 40, 62, "uniform"),
                                          (12345, 0, 13, "uniform"), (8192, 0, 24, "sorted"), (8191, 0, 24, "reverse"), (5000, 26, 48, "zipf")])
def test_emulated_radix_sort_is_the_stable_sort(radix, n, b0, b1, kind):
    keys = _keys(n, kind, n + b0 + b1)
    k, v = radix(keys, b0, b1)
    dig = (keys >> np.uint64(b0)) & np.uint64((1 << (b1 - b0)) - 1) if b1 > b0 else np.zeros(n, dtype=np.uint64)
    order = np.argsort(dig, kind="stable")
    assert np.array_equal(v, order.astype(np.uint32)), "values (= input positions) are not in stable order"
    assert np.array_equal(k, keys[order])
