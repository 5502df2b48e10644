"""Test infrastructure: on PYTHONPATH of the child processes of tests/test_engine_emu_cpu.py.  With ZPQ_TEST_EMU=1 every python
process they start (pytest runs, `python -c` helpers of the GPU tests) loads the emulated engine (tests/emu_build.py)."""
import os
import sys

if os.environ.get("ZPQ_TEST_EMU") == "1":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import emu_build
    emu_build.activate()
