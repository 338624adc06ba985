"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

ctypes/numpy front end of ``oracle/liboracle.so`` (the plain-C restatement of the
reference crate's SpMV / BLAS-1 / CG arithmetic, see ``sparsemat_oracle.h``).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; ``sparsemat_amd`` never does.

Parity status: pinned by the reference's own known-answer tests
(``tests/golden/``, ``tests/test_oracle_golden.py``).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ORC_OK, ORC_ERR_INDEX_OOB, ORC_ERR_NOT_SQUARE, ORC_ERR_SIZE_MISMATCH = 0, 1, 2, 3


class OraclePanic(RuntimeError):
    """Stands for a Rust panic of the reference; .code is the ORC_ERR_* value."""

    MESSAGES = {
        ORC_ERR_INDEX_OOB: "index out of bounds",  # densevec.rs:40-42
        ORC_ERR_NOT_SQUARE: "Matrix is not symmetric",  # linearsolver.rs:31
        ORC_ERR_SIZE_MISMATCH: "Matrix and vector size mismatch",  # linearsolver.rs:35
    }

    def __init__(self, code, message=None):
        super().__init__(message or self.MESSAGES.get(code, "oracle error %d" % code))
        self.code = code


def build(force=False):
    """Compile liboracle.so with the committed Makefile (gcc only)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "sparsemat_oracle.c")
    hdr = os.path.join(_HERE, "sparsemat_oracle.h")
    stale = (not os.path.exists(so)) or any(
        os.path.getmtime(f) > os.path.getmtime(so) for f in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("ORACLE_SO") or build()
        _LIB = C.CDLL(so)
        _declare(_LIB)
    return _LIB


_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_sz = C.c_size_t


def _declare(L):
    for suf, fp, ft in (("f32", _f32p, C.c_float), ("f64", _f64p, C.c_double)):
        getattr(L, "orc_spmv_" + suf).argtypes = [_sz, _u32p, _u32p, fp, fp, _sz, fp]
        getattr(L, "orc_spmv_rows_" + suf).argtypes = [_sz, _sz, _u32p, _u32p, fp, fp, _sz, fp]
        getattr(L, "orc_spmv_omp_" + suf).argtypes = [_sz, _u32p, _u32p, fp, fp, _sz, fp, C.c_int, C.POINTER(C.c_int)]
        getattr(L, "orc_spmv_abs_" + suf).argtypes = [_sz, _u32p, _u32p, fp, fp, _f64p]
        getattr(L, "orc_spmv_abs_" + suf).restype = None
        getattr(L, "orc_mat_inner_prod_" + suf).argtypes = [_sz, _u32p, _u32p, fp, fp, fp]
        getattr(L, "orc_mat_inner_prod_" + suf).restype = ft
        for op in ("add", "sub"):
            getattr(L, "orc_vec_%s_%s" % (op, suf)).argtypes = [fp, _sz, fp, _sz]
        getattr(L, "orc_vec_scale_" + suf).argtypes = [fp, _sz, ft]
        getattr(L, "orc_vec_scale_" + suf).restype = None
        getattr(L, "orc_vec_axpy_" + suf).argtypes = [fp, ft, fp, _sz]
        getattr(L, "orc_vec_axpy_" + suf).restype = None
        getattr(L, "orc_vec_xpby_" + suf).argtypes = [fp, ft, fp, _sz]
        getattr(L, "orc_vec_xpby_" + suf).restype = None
        getattr(L, "orc_dot_" + suf).argtypes = [fp, fp, _sz]
        getattr(L, "orc_dot_" + suf).restype = ft
        getattr(L, "orc_norm_squared_" + suf).argtypes = [fp, _sz]
        getattr(L, "orc_norm_squared_" + suf).restype = ft
        getattr(L, "orc_norm_" + suf).argtypes = [fp, _sz]
        getattr(L, "orc_norm_" + suf).restype = C.c_double
        getattr(L, "orc_cg_" + suf).argtypes = [_sz, _sz, _u32p, _u32p, fp, fp, _sz, fp, _sz,
                                                 C.c_double, _sz, C.POINTER(_sz), _f64p]
        getattr(L, "orc_pcg_jacobi_" + suf).argtypes = [_sz, _sz, _u32p, _u32p, fp, fp, _sz, fp, _sz,
                                                         C.c_double, _sz, C.POINTER(_sz), _f64p]
        getattr(L, "orc_gen_x_" + suf).argtypes = [C.c_uint64, _sz, _sz, fp]
        getattr(L, "orc_gen_x_" + suf).restype = None
        getattr(L, "orc_gen_fixed_" + suf).argtypes = [C.c_uint64, C.c_int, _sz, C.c_uint32, _sz,
                                                        _sz, _u32p, _u32p, fp]
        getattr(L, "orc_gen_fixed_" + suf).restype = None
        getattr(L, "orc_gen_fill_" + suf).argtypes = [C.c_uint64, _sz, _sz, _sz, _u32p, _u32p, fp]
        getattr(L, "orc_gen_fill_" + suf).restype = None
        getattr(L, "orc_laplace2d_" + suf).argtypes = [_sz, _sz, _u32p, _u32p, fp]
        getattr(L, "orc_laplace2d_" + suf).restype = _sz
        getattr(L, "orc_laplace3d_" + suf).argtypes = [_sz, _sz, _sz, _u32p, _u32p, fp]
        getattr(L, "orc_laplace3d_" + suf).restype = _sz
        getattr(L, "orc_laplace3d_rows_" + suf).argtypes = [_sz, _sz, _sz, _sz, _sz, _u32p, _u32p, fp]
        getattr(L, "orc_laplace3d_rows_" + suf).restype = _sz
        getattr(L, "orc_assemble_" + suf).argtypes = [_sz, _u32p, _u32p, fp, C.POINTER(C.c_uint8), C.POINTER(_sz),
                                                       C.POINTER(_sz), C.POINTER(_sz), _u32p, _u32p, fp]
        getattr(L, "orc_crs_replay_" + suf).argtypes = [_sz, _u32p, _u32p, fp, C.POINTER(C.c_uint8), C.POINTER(_sz),
                                                         C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz), _u32p, _u32p, fp]
        getattr(L, "orc_crs_prod_ops_" + suf).argtypes = [_sz, _sz, _u32p, _u32p, fp, _sz, _sz, _u32p, _u32p, fp, _sz,
                                                           C.POINTER(_sz), _u32p, _u32p, fp]
        getattr(L, "orc_crs_sort_rows_" + suf).argtypes = [_sz, _u32p, _u32p, fp]
        getattr(L, "orc_crs_sort_rows_" + suf).restype = None
    L.orc_par_rows_per_block.argtypes = [_sz, _sz]
    L.orc_par_rows_per_block.restype = _sz
    L.orc_par_block_and_row.argtypes = [_sz, _sz, _sz, C.POINTER(_sz), C.POINTER(_sz)]
    L.orc_par_block_and_row.restype = None
    L.orc_splitmix64.argtypes = [C.c_uint64]
    L.orc_splitmix64.restype = C.c_uint64
    L.orc_powerlaw_cdf.argtypes = [C.c_uint32, C.c_double, _u32p]
    L.orc_powerlaw_cdf.restype = None
    L.orc_gen_powerlaw_lengths.argtypes = [C.c_uint64, _sz, _sz, C.c_uint32, _u32p, _u32p]
    L.orc_gen_powerlaw_lengths.restype = None
    L.orc_merge_path_search.argtypes = [_sz, _sz, _u32p, _sz, _u64p, _u32p, _u32p]
    L.orc_merge_path_search.restype = None


# --------------------------------------------------------------------------- helpers
def _suf(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32", _f32p
    if dtype == np.float64:
        return "f64", _f64p
    raise TypeError("oracle handles f32/f64 only, got %s" % dtype)


def _p(a, ptr):
    return a.ctypes.data_as(ptr) if a is not None else None


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _check(rc):
    if rc != ORC_OK:
        raise OraclePanic(rc)


# --------------------------------------------------------------------------- SpMV
def spmv(offset_rows, columns, values, x, rows=None):
    """y = A.x as the reference computes it (sparsematrix.rs:146-158). len(y)==n_rows."""
    values = np.ascontiguousarray(values)
    suf, fp = _suf(values.dtype)
    off, col = _c(offset_rows, np.uint32), _c(columns, np.uint32)
    x = _c(x, values.dtype)
    n_rows = len(off) - 1
    y = np.zeros(n_rows, dtype=values.dtype)
    if rows is None:
        rc = getattr(lib(), "orc_spmv_" + suf)(n_rows, _p(off, _u32p), _p(col, _u32p), _p(values, fp),
                                               _p(x, fp), len(x), _p(y, fp))
    else:
        rc = getattr(lib(), "orc_spmv_rows_" + suf)(rows[0], rows[1], _p(off, _u32p), _p(col, _u32p),
                                                    _p(values, fp), _p(x, fp), len(x), _p(y, fp))
    _check(rc)
    return y


def spmv_omp(offset_rows, columns, values, x, threads=0):
    """orc_spmv's loop with the rows spread over all host cores -- NOT reference behaviour (the reference is serial);
    the second CPU row of bench.py.  Returns (y, threads used); y is bit-identical to spmv()."""
    values = np.ascontiguousarray(values)
    suf, fp = _suf(values.dtype)
    off, col = _c(offset_rows, np.uint32), _c(columns, np.uint32)
    x = _c(x, values.dtype)
    n_rows = len(off) - 1
    y = np.zeros(n_rows, dtype=values.dtype)
    nt = C.c_int(0)
    _check(getattr(lib(), "orc_spmv_omp_" + suf)(n_rows, _p(off, _u32p), _p(col, _u32p), _p(values, fp), _p(x, fp), len(x),
                                                 _p(y, fp), int(threads), C.byref(nt)))
    return y, nt.value


def spmv_abs(offset_rows, columns, values, x):
    """Per-row sum_j |a_ij x_j| in f64 (scale of the componentwise parity bound)."""
    values = np.ascontiguousarray(values)
    suf, fp = _suf(values.dtype)
    off, col = _c(offset_rows, np.uint32), _c(columns, np.uint32)
    x = _c(x, values.dtype)
    out = np.zeros(len(off) - 1, dtype=np.float64)
    getattr(lib(), "orc_spmv_abs_" + suf)(len(off) - 1, _p(off, _u32p), _p(col, _u32p), _p(values, fp),
                                          _p(x, fp), _p(out, _f64p))
    return out


def mat_inner_prod(offset_rows, columns, values, lhs, rhs):
    values = np.ascontiguousarray(values)
    suf, fp = _suf(values.dtype)
    off, col = _c(offset_rows, np.uint32), _c(columns, np.uint32)
    lhs, rhs = _c(lhs, values.dtype), _c(rhs, values.dtype)
    return values.dtype.type(getattr(lib(), "orc_mat_inner_prod_" + suf)(
        len(off) - 1, _p(off, _u32p), _p(col, _u32p), _p(values, fp), _p(lhs, fp), _p(rhs, fp)))


# --------------------------------------------------------------------------- DenseVec
def vec_add(x, y):
    x = np.array(x, copy=True)
    suf, fp = _suf(x.dtype)
    y = _c(y, x.dtype)
    rc = getattr(lib(), "orc_vec_add_" + suf)(_p(x, fp), len(x), _p(y, fp), len(y))
    if rc:
        raise OraclePanic(rc, "Dimension mismatch")  # densevec.rs:53
    return x


def vec_sub(x, y):
    x = np.array(x, copy=True)
    suf, fp = _suf(x.dtype)
    y = _c(y, x.dtype)
    rc = getattr(lib(), "orc_vec_sub_" + suf)(_p(x, fp), len(x), _p(y, fp), len(y))
    if rc:
        raise OraclePanic(rc, "Dimension mismatch")  # densevec.rs:62
    return x


def vec_scale(x, a):
    x = np.array(x, copy=True)
    suf, fp = _suf(x.dtype)
    getattr(lib(), "orc_vec_scale_" + suf)(_p(x, fp), len(x), x.dtype.type(a))
    return x


def vec_axpy(y, a, x):
    y = np.array(y, copy=True)
    suf, fp = _suf(y.dtype)
    x = _c(x, y.dtype)
    getattr(lib(), "orc_vec_axpy_" + suf)(_p(y, fp), y.dtype.type(a), _p(x, fp), len(y))
    return y


def vec_xpby(p, b, r):
    p = np.array(p, copy=True)
    suf, fp = _suf(p.dtype)
    r = _c(r, p.dtype)
    getattr(lib(), "orc_vec_xpby_" + suf)(_p(p, fp), p.dtype.type(b), _p(r, fp), len(p))
    return p


def dot(x, y):
    x = np.ascontiguousarray(x)
    suf, fp = _suf(x.dtype)
    y = _c(y, x.dtype)
    n = min(len(x), len(y))  # zip truncates (vector.rs:52)
    return x.dtype.type(getattr(lib(), "orc_dot_" + suf)(_p(x, fp), _p(y, fp), n))


def norm_squared(x):
    x = np.ascontiguousarray(x)
    suf, fp = _suf(x.dtype)
    return x.dtype.type(getattr(lib(), "orc_norm_squared_" + suf)(_p(x, fp), len(x)))


def norm(x):
    x = np.ascontiguousarray(x)
    suf, fp = _suf(x.dtype)
    return float(getattr(lib(), "orc_norm_" + suf)(_p(x, fp), len(x)))


# --------------------------------------------------------------------------- CG
def cg(n_rows, n_cols, offset_rows, columns, values, b, x0, tol=1e-12, iter_max=10_000):
    """ConjugateGradient::solve (linearsolver.rs:27-61). Returns (x, iters, rr)."""
    values = np.ascontiguousarray(values)
    suf, fp = _suf(values.dtype)
    off, col = _c(offset_rows, np.uint32), _c(columns, np.uint32)
    b = _c(b, values.dtype)
    x = np.array(x0, dtype=values.dtype, copy=True)
    iters, rr = _sz(0), C.c_double(0.0)
    rc = getattr(lib(), "orc_cg_" + suf)(n_rows, n_cols, _p(off, _u32p), _p(col, _u32p), _p(values, fp),
                                         _p(b, fp), len(b), _p(x, fp), len(x), tol, iter_max,
                                         C.byref(iters), C.byref(rr))
    _check(rc)
    return x, iters.value, rr.value


def pcg_jacobi(n_rows, n_cols, offset_rows, columns, values, b, x0, tol=1e-12, iter_max=10_000):
    """Jacobi-preconditioned CG -- an EXTENSION, not in the reference: ConjugateGradient::solve
    with z = r / diag(A).  Returns (x, iters, rr); a zero / absent diagonal raises OraclePanic(ORC_ERR_ZERO_DIAGONAL)."""
    values = np.ascontiguousarray(values)
    suf, fp = _suf(values.dtype)
    off, col = _c(offset_rows, np.uint32), _c(columns, np.uint32)
    b = _c(b, values.dtype)
    x = np.array(x0, dtype=values.dtype, copy=True)
    iters, rr = _sz(0), C.c_double(0.0)
    _check(getattr(lib(), "orc_pcg_jacobi_" + suf)(n_rows, n_cols, _p(off, _u32p), _p(col, _u32p), _p(values, fp),
                                                   _p(b, fp), len(b), _p(x, fp), len(x), tol, iter_max,
                                                   C.byref(iters), C.byref(rr)))
    return x, iters.value, rr.value


# --------------------------------------------------------------------------- SparseMatPar
def par_rows_per_block(n_blocks, max_n_rows):
    return lib().orc_par_rows_per_block(n_blocks, max_n_rows)


def par_block_and_row(n_blocks, rows_per_block, row):
    b, r = _sz(0), _sz(0)
    lib().orc_par_block_and_row(n_blocks, rows_per_block, row, C.byref(b), C.byref(r))
    return b.value, r.value


# --------------------------------------------------------------------------- synthetic inputs
def splitmix64(z):
    return lib().orc_splitmix64(z & 0xFFFFFFFFFFFFFFFF)


def gen_x(seed, n, dtype=np.float32, begin=0):
    suf, fp = _suf(dtype)
    x = np.empty(n, dtype=dtype)
    getattr(lib(), "orc_gen_x_" + suf)(seed, begin, n, _p(x, fp))
    return x


PATTERN_BANDED, PATTERN_UNIFORM, PATTERN_DIAG, PATTERN_WINDOW = 0, 1, 2, 3


def gen_fixed(seed, pattern, n, k, dtype=np.float32, row_begin=0, row_end=None):
    """k nnz per row; returns (offset_rows, columns, values) of rows [row_begin,row_end)."""
    suf, fp = _suf(dtype)
    row_end = n if row_end is None else row_end
    rows = row_end - row_begin
    off = np.empty(rows + 1, dtype=np.uint32)
    col = np.empty(rows * k, dtype=np.uint32)
    val = np.empty(rows * k, dtype=dtype)
    getattr(lib(), "orc_gen_fixed_" + suf)(seed, pattern, n, k, row_begin, row_end, _p(off, _u32p),
                                           _p(col, _u32p), _p(val, fp))
    return off, col, val


def powerlaw_cdf(kmax=2048, alpha=1.52):
    cdf = np.empty(kmax, dtype=np.uint32)
    lib().orc_powerlaw_cdf(kmax, alpha, _p(cdf, _u32p))
    return cdf


def gen_powerlaw(seed, n_rows, n_cols, dtype=np.float64, kmax=2048, alpha=1.52, row_begin=0,
                 row_end=None):
    suf, fp = _suf(dtype)
    row_end = n_rows if row_end is None else row_end
    rows = row_end - row_begin
    cdf = powerlaw_cdf(kmax, alpha)
    lengths = np.empty(rows, dtype=np.uint32)
    lib().orc_gen_powerlaw_lengths(seed, row_begin, row_end, kmax, _p(cdf, _u32p), _p(lengths, _u32p))
    off64 = np.zeros(rows + 1, dtype=np.uint64)
    np.cumsum(lengths, out=off64[1:])
    assert off64[-1] < 0xFFFFFFFF
    off = off64.astype(np.uint32)
    nnz = int(off[-1])
    col = np.empty(nnz, dtype=np.uint32)
    val = np.empty(nnz, dtype=dtype)
    getattr(lib(), "orc_gen_fill_" + suf)(seed, n_cols, row_begin, row_end, _p(off, _u32p),
                                          _p(col, _u32p), _p(val, fp))
    return off, col, val


def laplace2d(nx, ny, dtype=np.float32):
    suf, fp = _suf(dtype)
    f = getattr(lib(), "orc_laplace2d_" + suf)
    nnz = f(nx, ny, None, None, None)
    off = np.empty(nx * ny + 1, dtype=np.uint32)
    col = np.empty(nnz, dtype=np.uint32)
    val = np.empty(nnz, dtype=dtype)
    f(nx, ny, _p(off, _u32p), _p(col, _u32p), _p(val, fp))
    return off, col, val


def laplace3d(nx, ny, nz, dtype=np.float32):
    suf, fp = _suf(dtype)
    f = getattr(lib(), "orc_laplace3d_" + suf)
    nnz = f(nx, ny, nz, None, None, None)
    off = np.empty(nx * ny * nz + 1, dtype=np.uint32)
    col = np.empty(nnz, dtype=np.uint32)
    val = np.empty(nnz, dtype=dtype)
    f(nx, ny, nz, _p(off, _u32p), _p(col, _u32p), _p(val, fp))
    return off, col, val


def laplace3d_rows(nx, ny, nz, row_begin, row_end, dtype=np.float32):
    """Rows [row_begin, row_end) of laplace3d(nx, ny, nz): offsets rebased to 0, global columns (a sample of BASELINE C4)."""
    suf, fp = _suf(dtype)
    f = getattr(lib(), "orc_laplace3d_rows_" + suf)
    nnz = f(nx, ny, nz, row_begin, row_end, None, None, None)
    off = np.empty(row_end - row_begin + 1, dtype=np.uint32)
    col = np.empty(nnz, dtype=np.uint32)
    val = np.empty(nnz, dtype=dtype)
    f(nx, ny, nz, row_begin, row_end, _p(off, _u32p), _p(col, _u32p), _p(val, fp))
    return off, col, val


def merge_path_search(offset_rows, nnz, diagonals):
    off = _c(offset_rows, np.uint32)
    d = _c(diagonals, np.uint64)
    rows = np.empty(len(d), dtype=np.uint32)
    nz = np.empty(len(d), dtype=np.uint32)
    lib().orc_merge_path_search(len(off) - 1, nnz, _p(off, _u32p), len(d), _p(d, _u64p), _p(rows, _u32p),
                                _p(nz, _u32p))
    return rows, nz


def assemble(rows, cols, vals, ops=None):
    """add_to / set stream on a SparseMatIndexList followed by to_crs() (sparsemat_indexlist.rs:61-63,158-164;
    sparsemat_crs.rs:24-50).  ops[k]: 0 add_to, 1 set (None: all add_to).
    Returns (n_rows, n_cols, offset_rows, columns, values)."""
    vals = np.ascontiguousarray(vals)
    suf, fp = _suf(vals.dtype)
    rows, cols = _c(rows, np.uint32), _c(cols, np.uint32)
    n = len(vals)
    assert len(rows) == n and len(cols) == n
    ops_a = None if ops is None else _c(ops, np.uint8)
    max_rows = int(rows.max()) + 1 if n else 0
    off = np.zeros(max_rows + 1, np.uint32)
    col = np.zeros(max(n, 1), np.uint32)
    val = np.zeros(max(n, 1), vals.dtype)
    nr, nc, nnz = _sz(), _sz(), _sz()
    _check(getattr(lib(), "orc_assemble_" + suf)(
        n, _p(rows, _u32p), _p(cols, _u32p), _p(vals, fp),
        None if ops_a is None else ops_a.ctypes.data_as(C.POINTER(C.c_uint8)),
        C.byref(nr), C.byref(nc), C.byref(nnz), _p(off, _u32p), _p(col, _u32p), _p(val, fp)))
    return nr.value, nc.value, off[:nr.value + 1].copy(), col[:nnz.value].copy(), val[:nnz.value].copy()


def crs_replay(rows, cols, vals, ops=None):
    """The same stream replayed on a SparseMatCRS (sparsemat_crs.rs:54-92,143-149), first-push quirk included.
    Returns (n_rows, n_cols, offset_rows[n_rows+1], columns, values, stored): columns / values hold what iter_row
    reaches (offset_rows[n_rows] entries); `stored` = columns.len() of the reference (an orphaned first entry counts)."""
    vals = np.ascontiguousarray(vals)
    suf, fp = _suf(vals.dtype)
    rows, cols = _c(rows, np.uint32), _c(cols, np.uint32)
    n = len(vals)
    assert len(rows) == n and len(cols) == n
    ops_a = None if ops is None else _c(ops, np.uint8)
    off = np.zeros((int(rows.max()) + 2) if n else 1, np.uint32)
    col = np.zeros(max(n, 1), np.uint32)
    val = np.zeros(max(n, 1), vals.dtype)
    nr, nc, nnz, stored = _sz(), _sz(), _sz(), _sz()
    _check(getattr(lib(), "orc_crs_replay_" + suf)(
        n, _p(rows, _u32p), _p(cols, _u32p), _p(vals, fp),
        None if ops_a is None else ops_a.ctypes.data_as(C.POINTER(C.c_uint8)),
        C.byref(nr), C.byref(nc), C.byref(nnz), C.byref(stored), _p(off, _u32p), _p(col, _u32p), _p(val, fp)))
    return nr.value, nc.value, off[:nr.value + 1].copy(), col[:nnz.value].copy(), val[:nnz.value].copy(), stored.value


def transpose(offset_rows, columns, values):
    """SparseMatrix::transpose (sparsematrix.rs:174-184) of a SparseMatCRS: `ret.set(j, i, val)` for every entry in
    row-major storage order, into a fresh SparseMatCRS.  Same return as crs_replay."""
    off = _c(offset_rows, np.uint32)
    n_rows = len(off) - 1
    src_rows = np.repeat(np.arange(n_rows, dtype=np.uint32), np.diff(off.astype(np.int64)))
    nnz = int(off[-1]) if n_rows else 0
    return crs_replay(_c(columns, np.uint32)[:nnz], src_rows, np.ascontiguousarray(values)[:nnz], np.ones(nnz, np.uint8))


def prod(a, b, cap=None):
    """SparseMatrix::prod (sparsematrix.rs:186-210) for SparseMatCRS operands a, b = (n_rows, n_cols, offset_rows,
    columns, values): the reference's loops produce the call stream ret.set(i, j, sum); replayed on a SparseMatCRS.
    Same return as crs_replay.  Raises OraclePanic(ORC_ERR_SIZE_MISMATCH) for the reference's Err("Dimension mismatch")."""
    a_rows, a_cols, a_off, a_col, a_val = a
    b_rows, b_cols, b_off, b_col, b_val = b
    a_val = np.ascontiguousarray(a_val)
    suf, fp = _suf(a_val.dtype)
    b_val = _c(b_val, a_val.dtype)
    a_off, a_col, b_off, b_col = (_c(v, np.uint32) for v in (a_off, a_col, b_off, b_col))
    cap = int(cap if cap is not None else min(a_rows * max(b_cols, 1), 50_000_000))
    o_rows, o_cols, o_vals = np.zeros(max(cap, 1), np.uint32), np.zeros(max(cap, 1), np.uint32), np.zeros(max(cap, 1), a_val.dtype)
    n = _sz()
    _check(getattr(lib(), "orc_crs_prod_ops_" + suf)(
        a_rows, a_cols, _p(a_off, _u32p), _p(a_col, _u32p), _p(a_val, fp), b_rows, b_cols, _p(b_off, _u32p), _p(b_col, _u32p),
        _p(b_val, fp), cap, C.byref(n), _p(o_rows, _u32p), _p(o_cols, _u32p), _p(o_vals, fp)))
    k = n.value
    return crs_replay(o_rows[:k], o_cols[:k], o_vals[:k], np.ones(k, np.uint8))


def column_info(offset_rows, columns, n_cols):
    """ColumnIter::assemble_column_info (sparsemat_crs.rs:180-191): rows[k] = row of entry k, and per column the entry
    indices in storage order (what IndexList::iter_row of indexlist_col yields, indexlist.rs:62-83), as a CSC-like
    (col_ptr[n_cols+1], entries[nnz]) pair.  iter_col(j) = [(rows[e], values[e]) for e in entries[col_ptr[j]:col_ptr[j+1]]]."""
    off = _c(offset_rows, np.uint32)
    n_rows = len(off) - 1
    rows = np.repeat(np.arange(n_rows, dtype=np.uint32), np.diff(off.astype(np.int64)))
    nnz = len(rows)
    cols = _c(columns, np.uint32)[:nnz]
    lists = [[] for _ in range(n_cols)]
    for k in range(nnz):  # indexlist_col.push(j) in storage order
        lists[int(cols[k])].append(k)
    col_ptr = np.zeros(n_cols + 1, np.uint32)
    col_ptr[1:] = np.cumsum([len(l) for l in lists], dtype=np.uint64).astype(np.uint32)
    entries = np.array([e for l in lists for e in l], np.uint32)
    return rows, col_ptr, entries


def crs_sort_rows(offset_rows, columns, values):
    """Sortable::sort_row (sparsemat_crs.rs:163-172) on every row; returns sorted copies (stable by column)."""
    values = np.array(values, copy=True)
    suf, fp = _suf(values.dtype)
    off = _c(offset_rows, np.uint32)
    col = np.array(columns, dtype=np.uint32, copy=True)
    getattr(lib(), "orc_crs_sort_rows_" + suf)(len(off) - 1, _p(off, _u32p), _p(col, _u32p), _p(values, fp))
    return col, values
