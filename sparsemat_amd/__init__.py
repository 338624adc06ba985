"""sparsemat_amd -- MI355X (gfx950) drop-in for the CSR SpMV / BLAS-1 / CG hot path of the Rust
crate lostinc0de/sparsemat.

The product is ``libsparsemat_hip.so`` (hand-written HIP kernels behind the C ABI of
``include/sparsemat_hip.h``).  This package is the host-side mirror of the reference's interface
for that path -- same names as the crate root re-exports (``SparseMatCRS``, ``DenseVec``) and its
``linearsolver`` / ``sparsemat_par`` modules (``SparseMatPar`` = ``SparseMatParLocal``: ``smh_par_*``) -- used by the tests and
the benchmark.

There is no CPU fallback: importing works anywhere (so the C ABI can be inspected), computing
needs a HIP device.
"""
from ._lib import SparseMatPanic, lib, LIB_PATH  # noqa: F401
from .densevec import DenseVec  # noqa: F401
from .sparsemat_crs import SparseMatCRS  # noqa: F401
from .linearsolver import ConjugateGradient, JacobiConjugateGradient  # noqa: F401
from . import synth  # noqa: F401
from .sparsemat_par_local import Comm, ParVec, SparseMatParLocal  # noqa: F401

SparseMatPar = SparseMatParLocal  # the reference's name (sparsemat_par.rs:12) for the ONE implementation: csrc/par.hip behind smh_par_*

__all__ = ["SparseMatCRS", "DenseVec", "ConjugateGradient", "SparseMatPar", "SparseMatParLocal", "ParVec", "Comm", "SparseMatPanic", "synth",
           "lib", "LIB_PATH"]
