"""Helpers shared by the tests: seeded CRS builders and the parity bounds of SURVEY.md 8(d)."""
import numpy as np

import oracle

EPS = {np.dtype(np.float32): 2.0 ** -24, np.dtype(np.float64): 2.0 ** -53}
# north_star: f32 accumulations within 1e-5 relative; 1e-12 analogue for f64
REL_TOL = {np.dtype(np.float32): 1e-5, np.dtype(np.float64): 1e-12}


def random_crs(rng, n_rows, n_cols, row_lengths, dtype, sort_rows=False, dup=False):
    """CRS with the given per-row lengths; columns in draw order (unsorted, duplicates possible)."""
    row_lengths = np.asarray(row_lengths, dtype=np.int64)
    assert len(row_lengths) == n_rows
    off = np.zeros(n_rows + 1, dtype=np.uint32)
    np.cumsum(row_lengths, out=off[1:])
    nnz = int(off[-1])
    col = rng.integers(0, n_cols, size=nnz, dtype=np.uint32)
    if dup and nnz > 1:
        col[1::3] = col[0:-1:3][:len(col[1::3])]
    if sort_rows:
        for i in range(n_rows):
            col[off[i]:off[i + 1]].sort()
    val = rng.uniform(-1.0, 1.0, size=nnz).astype(dtype)
    return off, col, val


def assert_spmv_close(y, off, col, val, x, what=""):
    """normwise and componentwise parity bounds against the oracle (storage-order sequential)."""
    dt = np.dtype(val.dtype)
    y_ref = oracle.spmv(off, col, val, x)
    assert y.shape == y_ref.shape, what
    scale = oracle.spmv_abs(off, col, val, x)
    lens = np.diff(off.astype(np.int64))
    # the reference's own rounding grows with the row length: allow max(REL_TOL, 2*L*eps)
    rel = np.maximum(REL_TOL[dt], 2.0 * lens * EPS[dt])
    err = np.abs(y.astype(np.float64) - y_ref.astype(np.float64))
    bad = err > rel * scale + np.finfo(dt).tiny
    assert not bad.any(), "%s: %d rows out of bound, worst %g (row %d, len %d)" % (
        what, bad.sum(), (err / np.maximum(scale, 1e-300)).max(), int(np.argmax(err / np.maximum(scale, 1e-300))),
        int(lens[np.argmax(err / np.maximum(scale, 1e-300))]))
    denom = np.abs(y_ref).max() if len(y_ref) else 0.0
    if denom > 0:
        assert err.max() / denom <= max(REL_TOL[dt], 2.0 * lens.max() * EPS[dt]), what
    return y_ref
