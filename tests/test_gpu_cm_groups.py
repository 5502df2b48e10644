"""EXPERIMENTAL, not part of the round-3 record: several blocks per wave in the run-time generated context-mixing coder
(cm_spec_src.inc with ZS < 64, ZPQ_CM_GROUPS=1) was written at the end of round 3 without GPU time left to run it; on the CPU
emulator it passes.  These tests are what it has to pass on hardware before it may become the default; they only run with
ZPQ_TEST_EXPERIMENTAL=1.  The switch is read once per process, so the cases run in processes of their own."""
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("ZPQ_TEST_EXPERIMENTAL") != "1", reason="experimental path: set ZPQ_TEST_EXPERIMENTAL=1")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import hashlib, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import datagen, orc
from zpaqfranz_amd import Engine
eng = Engine(0)
sizes = [1, 7, 300, 4097, 9000, 2500, 0, 12000, 64, 5000, 3333]          # ragged: the groups of a wave finish at different times
out = []
for method in ("34", "44", "x4,0ci1,1m", "54", "x0,3ci1"):
    blocks = [(datagen.text_like(n, 10 + i) if i %% 2 else datagen.mixed(n, 20 + i)) if n else b"" for i, n in enumerate(sizes)]
    if method == "54":
        blocks = blocks[:6]
    res = eng.compress_blocks(blocks, [method] * len(blocks), ["f"] * len(blocks), None, True)
    for (st, fr), b in zip(res, blocks):
        assert st == 0, (method, st)
        if orc.have_ref():
            r = orc.ref_decompress_block(fr, len(b) + 64)                  # the REAL Decompresser
            assert r["data"] == b and r["sha1_ok"] == 1, (method, len(b))
    back = eng.decompress_blocks([f for _, f in res], [len(b) + 64 for b in blocks])
    for b, r in zip(blocks, back):
        assert r["status"] == 0 and r["data"] == b, (method, len(b), r["status"])
    out.append(hashlib.sha1(b"".join(f for _, f in res)).hexdigest())
print("DIGESTS", " ".join(out))
"""


def _run(extra):
    e = dict(os.environ, **extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, env=e, timeout=1500)
    assert r.returncode == 0, (extra, r.stdout[-1500:], r.stderr[-2500:])
    return [ln for ln in r.stdout.splitlines() if ln.startswith("DIGESTS")][-1]


def test_blocks_coded_in_lane_groups_equal_the_one_block_per_wave_coder():
    """Ragged batches of blocks through methods 3 / 4 / 5 and two explicit models, coded several to a wave: the real
    reference Decompresser restores every block, the grouped decoder restores every block, and the coded bytes are those of
    the default coder."""
    want = _run({})
    assert _run({"ZPQ_CM_GROUPS": "1"}) == want
    assert _run({"ZPQ_CM_GROUPS": "1", "ZPQ_CM_WAVES": "16"}) == want


def test_the_coder_tests_pass_with_lane_groups():
    e = dict(os.environ, ZPQ_CM_GROUPS="1")
    # (without the 9.4 MB fixture block in both directions: minutes, and one block at a time exercises no grouping)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_cm_spec.py"), "-x", "-q", "-p", "no:cacheprovider",
                        "--deselect", "tests/test_gpu_cm_spec.py::test_reference_archive_in_full_both_directions"],
                       capture_output=True, text=True, env=e, timeout=3000, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:]
