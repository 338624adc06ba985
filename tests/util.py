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


def exact_row_sums(off, col, val, x):
    """Row sums of a_ij * x_j far more exactly than either side of the comparison: f32 data in f64 (the products are exact
    there, 24 + 24 bits, and a sum of L <= 2048 terms errs by L * 2^-53), f64 data in the 80-bit long double of this host
    (2^-64 per operation).  Returned in that wide type."""
    wide = np.float64 if np.dtype(val.dtype) == np.dtype(np.float32) else np.longdouble
    n_rows = len(off) - 1
    out = np.zeros(n_rows, dtype=wide)
    if len(val) == 0:
        return out
    prod = val.astype(wide) * np.asarray(x)[col].astype(wide)
    lens = np.diff(off.astype(np.int64))
    nonempty = lens > 0
    # (the non-empty rows' starts, in order, cut the entry stream into exactly their segments)
    out[nonempty] = np.add.reduceat(prod, off[:-1].astype(np.int64)[nonempty])
    return out


BUCKETS = ((1, 8), (9, 32), (33, 128), (129, 512), (513, 1 << 31))


def _log_buckets(what, dt, lens, excess):
    """worst (|y_gpu - exact| - |y_oracle - exact|) / sum|a x| per row-length bucket -> the test log and, on the GPU box,
    gpurun_out/parity_buckets.jsonl (summarised under profiles/)."""
    import json
    import os
    rec = {"what": what, "dtype": dt.name, "rows": int(len(lens)), "tol": REL_TOL[dt], "buckets": {}}
    for lo, hi in BUCKETS:
        sel = (lens >= lo) & (lens <= hi)
        if sel.any():
            rec["buckets"]["%d-%s" % (lo, hi if hi < (1 << 31) else "")] = {"rows": int(sel.sum()), "worst_excess": float(excess[sel].max())}
    print("parity buckets", json.dumps(rec))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        try:
            with open(os.path.join(out_dir, "parity_buckets.jsonl"), "a") as f:
                f.write(json.dumps(rec) + "\n")
        except OSError:
            pass


def assert_spmv_close(y, off, col, val, x, what=""):
    """north_star's bound without a row-length allowance: against an (effectively) exact row sum, the device result may be
    no further off than the reference's own storage-order sequential sum (the oracle) plus 1e-5 * sum_j |a_ij x_j|
    (1e-12 for f64).  So a kernel is never blamed for the reference's rounding of a 2048-entry row, and never gets
    credit for it either: |y_gpu - exact| <= |y_oracle - exact| + tol * sum|a x| per row, and the same normwise."""
    dt = np.dtype(val.dtype)
    y_ref = oracle.spmv(off, col, val, x)
    assert y.shape == y_ref.shape, what
    scale = oracle.spmv_abs(off, col, val, x)
    lens = np.diff(off.astype(np.int64))
    exact = exact_row_sums(off, col, val, x)
    wide = exact.dtype
    err_gpu = np.abs(y.astype(wide) - exact).astype(np.float64)
    err_ref = np.abs(y_ref.astype(wide) - exact).astype(np.float64)
    excess = (err_gpu - err_ref) / np.maximum(scale, 1e-300)
    _log_buckets(what, dt, lens, excess)
    bad = err_gpu > err_ref + REL_TOL[dt] * scale + np.finfo(dt).tiny
    assert not bad.any(), "%s: %d rows out of bound, worst excess %g * sum|a x| (row %d, len %d)" % (
        what, bad.sum(), excess.max(), int(np.argmax(excess)), int(lens[np.argmax(excess)]))
    denom = float(np.abs(exact).max()) if len(exact) else 0.0
    if denom > 0:
        assert err_gpu.max() / denom <= err_ref.max() / denom + REL_TOL[dt], what
    return y_ref
