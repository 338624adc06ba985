"""GPU: the C++ mirror of the reference's interface (include/sparsemat.hpp) over the C ABI, replaying the
reference's own unit tests (tests/cpp/test_trait_surface.cpp)."""
import os
import subprocess

import pytest

import sparsemat_amd as sm

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_trait_surface(gpu, tmp_path):
    exe = str(tmp_path / "test_trait_surface")
    libdir = os.path.dirname(sm.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_trait_surface.cpp"), "-o", exe,
                           "-L", libdir, "-lsparsemat_hip", "-Wl,-rpath," + libdir])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok (0 failures)" in r.stdout, r.stdout + r.stderr
