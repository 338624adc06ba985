"""Synthetic workloads born in HBM (bench / test support; spec in DESIGN.md "Synthetic inputs").

Thin wrappers over the ``smh_synth_*`` entry points.  Device memory comes from ``DeviceBuffer``
(``smh_dev_alloc``), so nothing here needs torch.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib
from .sparsemat_crs import SparseMatCRS

PATTERN_BANDED, PATTERN_UNIFORM, PATTERN_DIAG, PATTERN_WINDOW = 0, 1, 2, 3
SEED_MATRIX, SEED_X = 0x5EED0001, 0x5EED0002


class DeviceBuffer:
    """A hipMalloc'ed byte range owned by Python."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        check(lib().smh_dev_alloc(max(int(nbytes), 16), C.byref(p)))
        self.ptr, self.nbytes = p.value, int(nbytes)

    def __del__(self):
        p, self.ptr = getattr(self, "ptr", None), None
        if p:
            try:
                lib().smh_dev_free(C.c_void_p(p))
            except Exception:
                pass

    def upload(self, array):
        a = np.ascontiguousarray(array)
        assert a.nbytes <= max(self.nbytes, 16)
        check(lib().smh_dev_upload(C.c_void_p(self.ptr), a.ctypes.data, a.nbytes))

    def download(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        check(lib().smh_dev_download(out.ctypes.data, C.c_void_p(self.ptr), out.nbytes))
        return out


def gen_x(seed, n, dtype=np.float32, begin=0, ptr=None):
    """x[j] = unit(hash(seed, begin+j)); returns (DeviceBuffer or None, ptr)."""
    buf = None
    if ptr is None:
        buf = DeviceBuffer(n * np.dtype(dtype).itemsize)
        ptr = buf.ptr
    check(lib().smh_synth_x(_lib.dtype_code(dtype), seed, begin, n, C.c_void_p(ptr), None))
    check(lib().smh_device_synchronize())
    return buf, ptr


def crs_fixed(seed, pattern, n, k, dtype=np.float32, row_begin=0, row_end=None):
    """k entries per row, rows [row_begin,row_end) of an n x n matrix -> SparseMatCRS (device born)."""
    row_end = n if row_end is None else row_end
    rows = row_end - row_begin
    nnz = rows * k
    off = DeviceBuffer((rows + 1) * 4)
    col = DeviceBuffer((nnz + 4) * 4)
    val = DeviceBuffer((nnz + 4) * np.dtype(dtype).itemsize)
    check(lib().smh_synth_fixed(_lib.dtype_code(dtype), seed, pattern, n, k, row_begin, row_end,
                                C.c_void_p(off.ptr), C.c_void_p(col.ptr), C.c_void_p(val.ptr), None))
    check(lib().smh_device_synchronize())
    return SparseMatCRS.from_device_parts(rows, n, nnz, off.ptr, col.ptr, val.ptr, dtype, keep=(off, col, val))


def powerlaw_offsets(seed, n_rows, kmax=2048, alpha=1.52, row_begin=0, row_end=None):
    row_end = n_rows if row_end is None else row_end
    cdf = np.empty(kmax, dtype=np.uint32)
    check(lib().smh_synth_powerlaw_cdf(kmax, alpha, cdf.ctypes.data))
    lengths = np.empty(row_end - row_begin, dtype=np.uint32)
    check(lib().smh_synth_powerlaw_lengths(seed, row_begin, row_end, kmax, cdf.ctypes.data, lengths.ctypes.data))
    off = np.zeros(len(lengths) + 1, dtype=np.uint64)
    np.cumsum(lengths, out=off[1:])
    if off[-1] >= 0xFFFFFFFF:
        raise _lib.SparseMatPanic(_lib.SMH_ERR_CAPACITY, "Maximum number of 4294967295 entries reached")
    return off.astype(np.uint32)


def crs_powerlaw(seed, n_rows, n_cols, dtype=np.float64, kmax=2048, alpha=1.52, row_begin=0, row_end=None):
    """Row lengths 1..kmax with P(k) ~ k^-alpha, uniform columns -> SparseMatCRS (device born)."""
    row_end = n_rows if row_end is None else row_end
    off_h = powerlaw_offsets(seed, n_rows, kmax, alpha, row_begin, row_end)
    rows, nnz = row_end - row_begin, int(off_h[-1])
    off = DeviceBuffer((rows + 1) * 4)
    off.upload(off_h)
    col = DeviceBuffer((nnz + 4) * 4)
    val = DeviceBuffer((nnz + 4) * np.dtype(dtype).itemsize)
    check(lib().smh_synth_fill(_lib.dtype_code(dtype), seed, n_cols, row_begin, row_end, C.c_void_p(off.ptr),
                               C.c_void_p(col.ptr), C.c_void_p(val.ptr), None))
    check(lib().smh_device_synchronize())
    return SparseMatCRS.from_device_parts(rows, n_cols, nnz, off.ptr, col.ptr, val.ptr, dtype, keep=(off, col, val))


def laplace3d_nnz(nx, ny, nz, row_begin=0, row_end=None):
    row_end = nx * ny * nz if row_end is None else row_end
    out = C.c_size_t()
    check(lib().smh_synth_laplace3d(_lib.SMH_F32, nx, ny, nz, row_begin, row_end, None, None, None, C.byref(out), None))
    return out.value


def crs_laplace3d(nx, ny, nz, dtype=np.float32, row_begin=0, row_end=None):
    """7-point Laplacian (diag 6, off-diag -1, natural ordering) -> SparseMatCRS (device born)."""
    n = nx * ny * nz
    row_end = n if row_end is None else row_end
    rows = row_end - row_begin
    nnz = laplace3d_nnz(nx, ny, nz, row_begin, row_end)
    off = DeviceBuffer((rows + 1) * 4)
    col = DeviceBuffer((nnz + 4) * 4)
    val = DeviceBuffer((nnz + 4) * np.dtype(dtype).itemsize)
    out = C.c_size_t()
    check(lib().smh_synth_laplace3d(_lib.dtype_code(dtype), nx, ny, nz, row_begin, row_end, C.c_void_p(off.ptr),
                                    C.c_void_p(col.ptr), C.c_void_p(val.ptr), C.byref(out), None))
    check(lib().smh_device_synchronize())
    return SparseMatCRS.from_device_parts(rows, n, nnz, off.ptr, col.ptr, val.ptr, dtype, keep=(off, col, val))
