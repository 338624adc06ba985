import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
