"""SparseMatParLocal: ``SparseMatPar<SparseMatCRS<T,u32>>`` (sparsemat_par.rs:12-35, 86-140) driven by ONE process --
the C-ABI form (``smh_par_*``, ``csrc/par.hip``) a single-process host like the reference's binds: one row block per
device (several blocks may share a device), blocks run concurrently, the CG exchanges halos device to device.

The one-process-per-GPU form over ``torch.distributed`` (RCCL) is ``sparsemat_par.SparseMatPar``.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


class SparseMatParLocal:
    def __init__(self, handle, dtype):
        self._h = handle
        self._dtype = np.dtype(dtype).type

    @classmethod
    def with_sub_matrices(cls, n_blocks, n_rows, n_cols, offset_rows, columns, values, device_ids=None, validate=True):
        """``SparseMatPar::with_sub_matrices(n_blocks, max_n_rows)`` (sparsemat_par.rs:20-28) filled from a global CRS:
        block b = rows [b R, (b+1) R), R = n_rows // n_blocks, the last block taking the remainder."""
        values = np.ascontiguousarray(values)
        if values.dtype not in (np.float32, np.float64):
            raise TypeError("the HIP path handles f32/f64 values only (got %s)" % values.dtype)
        off = np.ascontiguousarray(offset_rows, dtype=np.uint32)
        col = np.ascontiguousarray(columns, dtype=np.uint32)
        if len(off) != n_rows + 1 or len(col) != len(values):
            raise _lib.SparseMatPanic(_lib.SMH_ERR_INVALID, "malformed CRS arrays")
        dev = None
        if device_ids is not None:
            dev = (C.c_int * n_blocks)(*[int(d) for d in device_ids])
        h = C.c_void_p()
        check(lib().smh_par_create(_lib.dtype_code(values.dtype), n_blocks, dev, n_rows, n_cols, off.ctypes.data,
                                   col.ctypes.data if len(col) else None, values.ctypes.data if len(values) else None,
                                   1 if validate else 0, C.byref(h)))
        return cls(h, values.dtype)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().smh_par_destroy(h)
            except Exception:
                pass

    def n_blocks(self):
        return lib().smh_par_n_blocks(self._h)

    def n_rows(self):
        return lib().smh_par_n_rows(self._h)

    def n_cols(self):  # sparsemat_par.rs:109-115
        return lib().smh_par_n_cols(self._h)

    def n_non_zero_entries(self):  # :117-123
        return lib().smh_par_nnz(self._h)

    def rows_per_block(self):
        return lib().smh_par_rows_per_block(self._h)

    def block(self, b):
        """(row_begin, row_end, device, n_non_zero_entries) of block b."""
        crs, r0, r1, dev = C.c_void_p(), C.c_size_t(), C.c_size_t(), C.c_int()
        check(lib().smh_par_block(self._h, b, C.byref(crs), C.byref(r0), C.byref(r1), C.byref(dev)))
        return r0.value, r1.value, dev.value, lib().smh_crs_nnz(crs)

    def get_block_and_row_id(self, row):  # :31-35 (clamped to the last block)
        b, r = C.c_size_t(), C.c_size_t()
        check(lib().smh_par_get_block_and_row_id(self._h, row, C.byref(b), C.byref(r)))
        return b.value, r.value

    def scale(self, a):  # :135-139
        check(lib().smh_par_scale(self._h, float(a)))

    def mvp(self, rhs, variant="auto"):
        """``SparseMatrix::mvp`` through the blocks; returns a new vector with ``n_rows`` entries."""
        x = np.ascontiguousarray(rhs, dtype=self._dtype)
        y = np.zeros(self.n_rows(), self._dtype)
        check(lib().smh_par_spmv(self._h, x.ctypes.data if len(x) else None, len(x), y.ctypes.data, _lib.VARIANTS[variant]))
        return y

    def cg_solve(self, b, x, tol=1e-12, iter_max=10_000, variant="auto"):
        """``ConjugateGradient::solve(&par, &b, &mut x)``: x (numpy array of the matrix's dtype) is updated in place;
        returns (iterations, r.r)."""
        b = np.ascontiguousarray(b, dtype=self._dtype)
        if x.dtype != self._dtype or not x.flags["C_CONTIGUOUS"]:
            raise TypeError("x must be a contiguous %s array (it is updated in place)" % np.dtype(self._dtype).name)
        iters, rr = C.c_size_t(), C.c_double()
        check(lib().smh_par_cg_solve(self._h, b.ctypes.data if len(b) else None, len(b), x.ctypes.data if len(x) else None, len(x),
                                     float(tol), int(iter_max), _lib.VARIANTS[variant], C.byref(iters), C.byref(rr)))
        return iters.value, rr.value
