"""The start-independent fragmenter model (tools/proto/frag_independent.c) must give exactly the
serial loop's cuts: global (never reset) hash triggers + per-fragment disturbance windows.  CPU only; this is the
design check for the crossing-free GPU pass, not product code."""
import os
import subprocess

import pytest

import datagen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def proto(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("proto") / "frag_proto")
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tools", "proto", "frag_independent.c")])
    return exe


CASES = {
    "text": lambda: datagen.text_like(3 << 20, 1),
    "binary": lambda: datagen.binary_like(3 << 20, 2),
    "mixed": lambda: datagen.mixed(4 << 20, 3),
    "random": lambda: datagen.random_bytes(2 << 20, 4),
    "zeros": lambda: bytes(3 << 20),
    "period": lambda: bytes(range(256)) * 6000,
    "runs": lambda: b"".join(bytes([i % 256]) * 70000 + datagen.text_like(50000, i) for i in range(12)),
    "tiny": lambda: b"abc",
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("params", [(), ("512", "65024", "19")])
def test_model_reproduces_the_serial_cuts(proto, tmp_path, name, params):
    f = tmp_path / "in.bin"
    f.write_bytes(CASES[name]())
    r = subprocess.run([proto, str(f), *params], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "IDENTICAL" in r.stdout, r.stdout + r.stderr


@pytest.fixture(scope="module")
def proto_pages(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("proto2") / "frag_pages")
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tools", "proto", "frag_pages.c")])
    return exe


@pytest.mark.parametrize("name", ["text", "binary", "mixed", "random", "tiny"])
@pytest.mark.parametrize("seg", ["65536", "262144", "344064"])
def test_page_model_reproduces_the_serial_cuts(proto_pages, tmp_path, name, seg):
    """Stage 2: page summaries + equal-work lanes + page-by-page chain (tools/proto/frag_pages.c)."""
    f = tmp_path / "in.bin"
    f.write_bytes(CASES[name]())
    r = subprocess.run([proto_pages, str(f), seg], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "IDENTICAL" in r.stdout, r.stdout + r.stderr


@pytest.fixture(scope="module")
def proto_dual(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("proto3") / "frag_dual")
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tools", "proto", "frag_dual.c")])
    return exe


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("params", [("262144",), ("65536",), ("344064", "512", "65024", "19"), ("100000", "1024", "8192", "10")])
def test_dual_table_walker_reproduces_the_serial_cuts(proto_dual, tmp_path, name, params):
    """Stage 3: the per-file chain as a dual-table walker (tools/proto/frag_dual.c) -- true and global o1[] tables kept
    current at the walker's position, pages skipped while they hold no context on which the tables differ, no backward
    scans; also on the inputs stage 2 cannot finish (fully predictable data: the walker simply never leaves exact mode)."""
    f = tmp_path / "in.bin"
    f.write_bytes(CASES[name]())
    r = subprocess.run([proto_dual, str(f), *params], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "IDENTICAL" in r.stdout, r.stdout + r.stderr
