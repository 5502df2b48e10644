import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


# ZPQ_TEST_EMU=1 (set by tests/test_engine_emu_cpu.py for its child processes, never by the driver): the GPU test files run
# against tests/_emu/libzpaqhip.so -- the engine's own sources compiled for the host over the fibre emulator
# (tests/emu_build.py).  The product knows nothing of it: the test process points the loader at the other directory
# (tests/emu_site/usercustomize.py does the same for the python processes the tests start).
if os.environ.get("ZPQ_TEST_EMU") == "1":
    import emu_build
    emu_build.activate()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libzpaqref.so (the reference compiled in place)")


def pytest_collection_modifyitems(config, items):
    import orc
    if not orc.have_ref():
        skip = pytest.mark.skip(reason="oracle/_ref/libzpaqref.so not built (no /root/reference here)")
        for it in items:
            if "ref" in it.keywords:
                it.add_marker(skip)
