"""ConjugateGradient: host-side mirror of the reference's ``LinearSolver`` / ``ConjugateGradient``
(linearsolver.rs:6-61).

* ``ConjugateGradient``     -- one GPU: the whole iteration runs device-resident inside the HIP library
                               (``smh_cg_solve*``: fused kernels, scalars in HBM, hipGraph replay).
* ``ParConjugateGradient``  -- the same recurrence over a row-partitioned ``SparseMatPar`` (one process per
                               GPU): per iteration one halo exchange of p, the local SpMV, and TWO scalar
                               all-reduces (p.Ap and r.r) on device-resident 1-element tensors; the vector
                               updates are the library's kernels with device-resident scalars
                               (``smh_blas_*_dev``), so no scalar visits the host except the stop test.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib
from .densevec import DenseVec


class ConjugateGradient:
    """``ConjugateGradient::default()`` == ``ConjugateGradient()`` (tol 1e-12, iter_max 10000,
    linearsolver.rs:17-24).  The reference keeps both fields private; ``new(tol, iter_max)`` is the
    documented addition."""

    def __init__(self, tol=1e-12, iter_max=10_000, variant="auto", check_every=0):
        self.tol = float(tol)
        self.iter_max = int(iter_max)
        self.variant = variant
        self.check_every = int(check_every)
        self.iterations = None  # loop bodies entered by the last solve
        self.r_norm_squared = None  # last r.r (as f64)

    @classmethod
    def default(cls):
        return cls()

    @classmethod
    def new(cls, tol, iter_max):
        return cls(tol, iter_max)

    def solve(self, mat, b, x):
        """LinearSolver::solve(&self, mat, b, x): x is updated in place (linearsolver.rs:27-61).
        Raises SparseMatPanic("Matrix is not symmetric") / ("Matrix and vector size mismatch")."""
        iters, rr = C.c_size_t(), C.c_double()
        var = _lib.VARIANTS[self.variant]
        if isinstance(b, DenseVec) and isinstance(x, DenseVec):
            check(lib().smh_cg_solve_vec(mat._h, b._h, x._h, self.tol, self.iter_max, var, self.check_every,
                                         C.byref(iters), C.byref(rr)))
        else:
            if not (isinstance(x, np.ndarray) and x.dtype == mat.dtype and x.flags.c_contiguous):
                raise TypeError("x must be a contiguous numpy array of the matrix dtype (updated in place)")
            bb = np.ascontiguousarray(b, dtype=mat.dtype)
            check(lib().smh_cg_solve(mat._h, bb.ctypes.data if bb.size else None, bb.size,
                                     x.ctypes.data if x.size else None, x.size, self.tol, self.iter_max, var,
                                     C.byref(iters), C.byref(rr)))
        self.iterations, self.r_norm_squared = iters.value, rr.value
        return x


class JacobiConjugateGradient:
    """Jacobi-preconditioned CG -- an EXTENSION (the reference has no preconditioner): the
    recurrence of ``ConjugateGradient::solve`` with z = r / diag(A); same guards, stop rule and panics
    (``smh_pcg_jacobi_solve``).  Host vectors; x (numpy array of the matrix dtype) is updated in place."""

    def __init__(self, tol=1e-12, iter_max=10_000, variant="auto"):
        self.tol = float(tol)
        self.iter_max = int(iter_max)
        self.variant = variant
        self.iterations = None
        self.r_norm_squared = None

    def solve(self, mat, b, x):
        if not (isinstance(x, np.ndarray) and x.dtype == mat.dtype and x.flags.c_contiguous):
            raise TypeError("x must be a contiguous numpy array of the matrix dtype (updated in place)")
        bb = np.ascontiguousarray(b, dtype=mat.dtype)
        iters, rr = C.c_size_t(), C.c_double()
        check(lib().smh_pcg_jacobi_solve(mat._h, bb.ctypes.data if bb.size else None, bb.size, x.ctypes.data if x.size else None,
                                         x.size, self.tol, self.iter_max, _lib.VARIANTS[self.variant], C.byref(iters), C.byref(rr)))
        self.iterations, self.r_norm_squared = iters.value, rr.value
        return x
