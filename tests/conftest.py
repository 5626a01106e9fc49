import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import etl_amd  # noqa: E402,F401  (before anything initialises HIP: the package sets the process's hardware-queue default)


# Every context the suites create checks the batch-state invariants of the ASYNC chain at its entry points (host_orchestrate.inc:
# check_invariants aborts the process with the violated rule): on the MI355X and under the CPU emulator alike. bench.py does not set it.
os.environ.setdefault("ETLG_DEBUG_INVARIANTS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) on a box without a device when selected by accident.
    if os.environ.get("ETLG_SIMT_RUN") == "1":
        return   # tests/test_simt_emulation.py re-runs the parity files against the SIMT emulator build (tests/simt)
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
