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


class HipVectorOps:
    """DenseVec kernels of the library on torch CUDA tensors, scalars read from device memory."""

    def __init__(self, like):
        import torch
        self._torch = torch
        self.dtype_code = _lib.SMH_F64 if like.dtype == torch.float64 else _lib.SMH_F32
        self.scratch = torch.empty(lib().smh_blas_dot_scratch_bytes(), dtype=torch.uint8, device=like.device)

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream().cuda_stream)

    def dot(self, x, y, out):
        check(lib().smh_blas_dot_dev(self.dtype_code, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), x.numel(),
                                     C.c_void_p(out.data_ptr()), C.c_void_p(self.scratch.data_ptr()), self._stream()))

    def axpy(self, y, a, x):  # y += round(a*x)
        check(lib().smh_blas_axpy_dev(self.dtype_code, C.c_void_p(y.data_ptr()), C.c_void_p(a.data_ptr()),
                                      C.c_void_p(x.data_ptr()), y.numel(), self._stream()))

    def xpby(self, p, b, r):  # p = round(b*p) + r
        check(lib().smh_blas_xpby_dev(self.dtype_code, C.c_void_p(p.data_ptr()), C.c_void_p(b.data_ptr()),
                                      C.c_void_p(r.data_ptr()), p.numel(), self._stream()))


class ParConjugateGradient:
    """ConjugateGradient::solve (linearsolver.rs:27-61) over a SparseMatPar: rank b holds rows
    [b*R, (b+1)*R) of A and the matching slices of b and x.  Same update order and roundings as the
    reference; the two reductions are local deterministic trees + an all-reduce (sum) over the ranks."""

    def __init__(self, tol=1e-12, iter_max=10_000, ops=None):
        self.tol, self.iter_max, self.ops = float(tol), int(iter_max), ops
        self.iterations = None
        self.r_norm_squared = None

    def solve(self, par, b_local, x_local):
        import torch
        import torch.distributed as dist
        if par.n_rows() != par.n_cols():
            raise _lib.SparseMatPanic(_lib.SMH_ERR_NOT_SQUARE, "Matrix is not symmetric")            # :30-32
        rows = par.end - par.begin
        if b_local.numel() != rows or x_local.numel() != rows:
            raise _lib.SparseMatPanic(_lib.SMH_ERR_DIM_MISMATCH, "Matrix and vector size mismatch")  # :33-36
        ops = self.ops or HipVectorOps(x_local)
        multi = par.n_blocks > 1
        if par._plan is None:
            par.setup_window_exchange(x_local)

        def allreduce(t):
            if multi:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=par.group)

        n = par.n_rows()
        dev, dt = x_local.device, x_local.dtype
        full = torch.zeros(n, dtype=dt, device=dev)        # holds x, then p: own slice + exchanged window
        mine = full[par.begin:par.end]
        ap = torch.empty(rows, dtype=dt, device=dev)
        rr, rr_prev, pap, alpha, neg_alpha, beta = (torch.zeros(1, dtype=dt, device=dev) for _ in range(6))
        # r = b - A x (:38); p = r (:39); rr = r.r (:40)
        mine.copy_(x_local)
        par.exchange_window(full)
        par.mvp_local(full, ap)
        r = b_local - ap
        mine.copy_(r)                                      # p lives in `full`
        ops.dot(r, r, rr)
        allreduce(rr)
        iters = 0
        for _k in range(self.iter_max):
            iters += 1
            par.exchange_window(full)                      # the entries of p this block references
            par.mvp_local(full, ap)                        # :43
            ops.dot(mine, ap, pap)
            allreduce(pap)
            torch.div(rr, pap, out=alpha)                  # :45
            ops.axpy(x_local, alpha, mine)                 # *x += p * alpha          :47
            torch.neg(alpha, out=neg_alpha)
            ops.axpy(r, neg_alpha, ap)                     # r -= mat_p * alpha       :49  (r + round(-alpha*Ap))
            rr_prev.copy_(rr)
            ops.dot(r, r, rr)                              # :51
            allreduce(rr)
            if float(rr.item()) ** 0.5 < self.tol:         # :52-54, before the beta update
                break
            torch.div(rr, rr_prev, out=beta)               # :56
            ops.xpby(mine, beta, r)                        # p = beta*p + r           :58-59
        self.iterations = iters
        self.r_norm_squared = float(rr.item())
        return x_local
