import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter):
    """SpMV parity: how many rows the suite compared against the oracle and how many of them needed the looser form of the
    bound (tests/util.py::assert_spmv_close) -- also appended to gpurun_out/parity_buckets.jsonl on the GPU box."""
    util = sys.modules.get("util")  # the module object the tests imported (tests/ is on sys.path: rootdir conftest)
    if util is None or not hasattr(util, "PARITY_STATS"):
        return
    st = util.PARITY_STATS
    if not st["comparisons"]:
        return
    import json
    line = "spmv parity: %d comparisons, %d rows against the oracle, literal bound |y_gpu - y_oracle| <= tol * sum|a x| held on all but %d rows (fallback rows); worst literal ratio %s" % (
        st["comparisons"], st["rows"], st["fallback_rows"], json.dumps(st["worst_literal"]))
    terminalreporter.write_line(line)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        try:
            with open(os.path.join(out_dir, "parity_buckets.jsonl"), "a") as f:
                f.write(json.dumps({"summary": st}) + "\n")
        except OSError:
            pass


def _torch_first():
    """This image's PyTorch bundles its own HIP runtime (ROCm 7.0) while hipcc links the system one (7.2):
    when both live in one process PyTorch must initialise first (the other order leaves torch without a
    device).  bench.py does the same."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


_torch_first()  # at conftest import: before any test module can load libsparsemat_hip.so (collection-time skipifs do)


def _gpu_count():
    try:
        import ctypes as C
        import sparsemat_amd
        n = C.c_int(0)
        sparsemat_amd.lib().smh_device_count(C.byref(n))
        return n.value
    except Exception:
        return 0


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must run the HIP path; a missing device or library is a FAILURE, not a skip."""
    import sparsemat_amd
    assert os.path.exists(sparsemat_amd.LIB_PATH), "libsparsemat_hip.so missing: run `python -m sparsemat_amd.build`"
    n = _gpu_count()
    assert n > 0, "no HIP device visible: -m gpu tests need a GPU (there is no CPU fallback)"
    return n
