#!/bin/bash
# Every GPU test file against the emulated engine (tests/emu_build.py), one pytest process per file, in parallel.
# ~25-40 minutes on 8 cores; the five tests that need torch on a GPU fail by construction (listed in
# tests/test_engine_emu_cpu.py NEEDS_TORCH), three context-mixing tests over megabytes take hours and are deselected.
# usage: tools/emu/run_gpu_tests_on_cpu.sh [outdir]   (ZPQ_TEST_EXPERIMENTAL=1 adds the experimental paths)
R=$(cd "$(dirname "$0")/../.." && pwd)
O=${1:-/tmp/emu_runs}
mkdir -p "$O"; : > "$O/done.txt"
cd "$R"
python tests/emu_build.py || exit 1
export ZPQ_TEST_EMU=1 PYTHONPATH="$R/tests/emu_site${PYTHONPATH:+:$PYTHONPATH}"
for f in tests/test_gpu_*.py; do
  n=$(basename "$f" .py)
  ( timeout 3000 python -m pytest "$f" -m gpu -q --durations=10 -p no:cacheprovider \
      --deselect tests/test_gpu_cm_spec.py::test_reference_archive_in_full_both_directions \
      --deselect tests/test_gpu_parity.py::test_libzpaq_shim_decompresser_class_reads_fixture_archives \
      --deselect tests/test_gpu_parity.py::test_compress_block_level5_reproduces_fixture_archive > "$O/$n.log" 2>&1
    echo "$n rc=$? $(tail -1 "$O/$n.log")" >> "$O/done.txt" ) &
done
wait
cat "$O/done.txt"
