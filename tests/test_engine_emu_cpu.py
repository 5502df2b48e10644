"""The GPU tests on the CPU.  tests/emu_build.py compiles the engine's own sources -- zpaqfranz_amd/csrc/*.hip, host code AND
kernels, unchanged but for inline assembly -- over a stand-in HIP runtime (tests/cpp/emu_rt) whose kernel launch runs every
workgroup on the fibre emulator (tests/cpp/simt_emu.h: lanes are fibres, wave and workgroup operations are rendezvous); the
context-mixing coder's run-time generated kernels are compiled by the host compiler where hiprtc would.  The GPU test files
(tests/test_gpu_*.py) then run UNCHANGED in child processes whose loader points at that library (ZPQ_TEST_EMU=1: conftest.py
and tests/emu_site/usercustomize.py), i.e. the same assertions against the oracle, the reference and the golden fixtures.

What this is: a logic check of every kernel and every host path of the engine at each commit, without a GPU.  What it is
not: the memory system, real concurrency between workgroups, timing -- and it is never the product (zpaqfranz_amd loads
libzpaqhip.so and fails without a gfx950 device).

All GPU tests pass on the emulator (the experimental paths' too) except the five that need torch on a GPU (tools/emu/run_gpu_tests_on_cpu.sh runs
everything, ~25 minutes on 8 cores).  Here: those that finish in seconds -- the others are deselected by name below -- plus
the experimental switches' own tests and the two-process sharded add (about 170 tests, ~2 minutes on 8 cores)."""
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import pytest

import emu_build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not emu_build.available(), reason="ROCm clang++ not found")

# need torch with a GPU (bench.py / tools children)
NEEDS_TORCH = [
    "test_gpu_lzdec.py::test_more_than_65536_stream_segments_in_one_call",
    "test_gpu_round2.py::test_sha1_extents_staged_and_direct_forms_agree_with_hashlib",
    "test_gpu_sa.py::test_text_m2_two_ranks",
    "test_gpu_parity.py::test_two_rank_add_is_bit_identical_to_serial",
    "test_gpu_parity.py::test_two_rank_shared_corpus_equals_the_single_gpu_archive",
]
# pass on the emulator, but take minutes there (megabytes through the context-mixing coder, 64 MiB blocks ...)
SLOW = [
    "test_gpu_lzdec.py::test_many_streams_in_one_call",
    "test_gpu_lzdec.py::test_truncated_streams",
    "test_gpu_lzdec.py::test_token_path_blocks_with_raw_offset_bits",
    "test_gpu_lzdec.py::test_damaged_and_random_streams_agree",
    "test_gpu_lzdec.py::test_token_path_equals_wave_decoder_on_real_streams",
    "test_gpu_m3.py::test_inverse_bwt_and_level2_decoder_on_larger_blocks",
    "test_gpu_m3.py::test_reference_decompresser_restores_and_stream_equals_lzbuffer",
    "test_gpu_round2.py::test_lz77_encoder_segment_size_never_changes_the_stream",
    "test_gpu_round2.py::test_compress_block_e8e9_and_big_blocks[x5,5,6,0,3,25-17838137]",
    "test_gpu_round2.py::test_jidac_add_multi_is_identical_to_single_context",
    "test_gpu_round2.py::test_many_blocks_one_call",
    "test_gpu_round2.py::test_resident_decode_roundtrip_mixed_batch",
    "test_gpu_sa.py::test_block_64mib_properties",
    "test_gpu_sa.py::test_jidac_add_with_method_2",
    "test_gpu_sa.py::test_sa_parse_equals_reference_lzbuffer_8mib",
    "test_gpu_sa.py::test_suffix_array_equals_divsufsort_4mib",
    "test_gpu_verify.py::test_verify_plain_archive_has_nothing_stored_to_compare",
    "test_gpu_verify.py::test_verify_reports_a_wrong_stored_checksum",
    "test_gpu_parity.py::test_libzpaq_shim_decompresser_class_reads_fixture_archives",
    "test_gpu_parity.py::test_compress_block_level5_reproduces_fixture_archive",
    "test_gpu_parity.py::test_compress_block_cm_methods_equal_reference_coder",
    "test_gpu_parity.py::test_lz77_full_16mib_block",
    "test_gpu_parity.py::test_compress_block_level5_prefix_of_second_fixture",
    "test_gpu_parity.py::test_sha_more_extents_than_lanes_longest_first",
    "test_gpu_parity.py::test_cm_encode_decode_equal_reference[alltypes]",
    "test_gpu_parity.py::test_cm_encode_decode_equal_reference[mid]",
    "test_gpu_parity.py::test_lz77_streams_bit_identical[4,1,4,0,2,24]",
    "test_gpu_parity.py::test_lz77_streams_bit_identical[4,1,5,0,3,24]",
    "test_gpu_parity.py::test_lz77_streams_bit_identical[0,1,4,0,2,20]",
    "test_gpu_parity.py::test_lz77_streams_bit_identical[4,1,4,0,2,16]",
    "test_gpu_parity.py::test_lz77_streams_bit_identical[0,1,4,0,2,16]",
    "test_gpu_parity.py::test_lz77_streams_bit_identical[4,1,4,0,1,15]",
    "test_gpu_parity.py::test_lz77_streams_bit_identical[0,1,5,0,3,20]",
    "test_gpu_cm_spec.py::test_many_blocks_several_headers_one_call",
    "test_gpu_cm_spec.py::test_builtin_models_of_startblock_level_equal_reference[2]",
    "test_gpu_cm_spec.py::test_specialised_kernel_equals_generic_kernel",
    "test_gpu_cm_spec.py::test_reference_archive_in_full_both_directions",
    "test_gpu_cm_spec.py::test_methods_3_4_5_models_equal_reference",
    "test_gpu_cm_spec.py::test_builtin_models_of_startblock_level_equal_reference[3]",
]
FILES = ["test_gpu_twins.py", "test_gpu_verify.py", "test_gpu_lzdec.py", "test_gpu_m3.py", "test_gpu_round2.py", "test_gpu_sa.py", "test_gpu_parity.py",
         "test_gpu_cm_spec.py", "test_gpu_segments.py"]


def emu_env():
    env = dict(os.environ, ZPQ_TEST_EMU="1")
    env["PYTHONPATH"] = os.path.join(ROOT, "tests", "emu_site") + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    env.pop("ZPQ_TEST_EXPERIMENTAL", None)
    return env


def run_file(job):
    name, extra, timeout = job[:3]
    more_env = job[3] if len(job) > 3 else {}
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", name), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "--durations=5"] + extra
    for d in NEEDS_TORCH + SLOW:
        if d.startswith(name + "::"):
            cmd += ["--deselect", "tests/" + d]
    t0 = time.time()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(emu_env(), **more_env), timeout=timeout, cwd=ROOT)
        return name, r.returncode, time.time() - t0, r.stdout[-3000:] + r.stderr[-1500:]
    except subprocess.TimeoutExpired as e:
        return name, -1, time.time() - t0, "timeout: " + str(e.stdout)[-2000:]


def test_the_gpu_test_files_pass_on_the_emulated_engine():
    emu_build.build()
    jobs = [(f, [], 900) for f in FILES]
    # the two largest files in halves, so that no worker is the long pole
    jobs = [j for j in jobs if j[0] not in ("test_gpu_parity.py", "test_gpu_round2.py")]
    jobs += [("test_gpu_parity.py", ["-k", "fragmenter or sha or dedup or e8e9 or shim"], 900),
             ("test_gpu_parity.py", ["-k", "journaling or lz77"], 900),
             ("test_gpu_parity.py", ["-k", "not (fragmenter or sha or dedup or e8e9 or shim or journaling or lz77)"], 900),
             ("test_gpu_round2.py", ["-k", "shim or jidac or resident"], 900), ("test_gpu_round2.py", ["-k", "not (shim or jidac or resident)"], 900)]
    # row (e): the journaling add sharded over two PROCESSES (gloo, world size 2), each with its own emulated engine, gives
    # the single-GPU archive
    # (zpqj_add_sharded_dev -- `-k resident_in_hbm` -- passes here too, 80 s per variant: left to the GPU suite and to
    #  tools/emu/run_gpu_tests_on_cpu.sh to keep this suite's time)
    jobs.append(("test_sharded_add.py", ["-k", "flags0 or flags2 or failing"], 900))
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 2)) as ex:
        res = list(ex.map(run_file, jobs))
    report = "\n".join("%s rc=%d %.0f s: %s" % (n, rc, t, out.strip().splitlines()[-1] if out.strip() else "") for n, rc, t, out in res)
    print(report)
    bad = [(n, rc, out) for n, rc, t, out in res if rc != 0]
    assert not bad, "\n\n".join("%s rc=%d\n%s" % b for b in bad)
    passed = sum(int(out.split(" passed")[0].split()[-1]) for _, _, _, out in res if " passed" in out)
    assert passed >= 115, report
