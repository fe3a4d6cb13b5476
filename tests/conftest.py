import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    return oracle_binding.load()


def pytest_sessionfinish(session, exitstatus):
    """tests/test_gpu_labels.py runs the parity tests once more in a child process against the label-counting build of the DP kernel
    (MM2C_LIB_PATH = minimap2-fpga_amd/variants/labelcount.so) and asks for the table of label hits here, when that run ends."""
    path = os.environ.get("MM2C_LABEL_TABLE")
    if not path:
        return
    import ctypes as C
    import json
    import mm2chain
    from mm2chain import _native as N
    mm2chain.init()
    hits = (C.c_ulonglong * 256)()
    N.check(N.load().mm2c_debug_label_hits(hits, 0), "mm2c_debug_label_hits")
    json.dump({"lib": N.LIB_PATH, "exitstatus": int(exitstatus), "hits": [int(v) for v in hits]}, open(path, "w"))
    mm2chain.shutdown()
