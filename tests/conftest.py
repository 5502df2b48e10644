import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


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
