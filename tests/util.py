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


# what the suite reports at its end (tests/conftest.py): rows compared, rows that needed the looser form, worst literal ratio
PARITY_STATS = {"comparisons": 0, "rows": 0, "fallback_rows": 0, "worst_literal": {}}


def assert_spmv_close(y, off, col, val, x, what=""):
    """SURVEY 8(d)'s componentwise gate, literally: |y_gpu - y_oracle| <= tol * sum_j |a_ij x_j| per row (tol 1e-5 for f32,
    1e-12 for f64), the oracle being the reference's storage-order sequential sum.  Only a row in which THE ORACLE ITSELF is
    further than tol * sum|a x| from the (effectively) exact row sum -- the reference's own rounding of a very long row --
    may instead satisfy |y_gpu - exact| <= |y_oracle - exact| + tol * sum|a x|: there the device is held to being no worse
    than the reference, not to reproducing its error.  Such rows are counted (PARITY_STATS; the suite prints the total) --
    none is expected at these tolerances.  The same normwise."""
    dt = np.dtype(val.dtype)
    y_ref = oracle.spmv(off, col, val, x)
    assert y.shape == y_ref.shape, what
    scale = oracle.spmv_abs(off, col, val, x)
    lens = np.diff(off.astype(np.int64))
    exact = exact_row_sums(off, col, val, x)
    wide = exact.dtype
    tol = REL_TOL[dt]
    tiny = np.finfo(dt).tiny
    literal = np.abs(y.astype(wide) - y_ref.astype(wide)).astype(np.float64)
    err_gpu = np.abs(y.astype(wide) - exact).astype(np.float64)
    err_ref = np.abs(y_ref.astype(wide) - exact).astype(np.float64)
    excess = (err_gpu - err_ref) / np.maximum(scale, 1e-300)
    _log_buckets(what, dt, lens, excess)
    fails_literal = literal > tol * scale + tiny
    oracle_is_far = err_ref > tol * scale + tiny          # the only rows the looser form is for
    fallback = fails_literal & oracle_is_far
    PARITY_STATS["comparisons"] += 1
    PARITY_STATS["rows"] += int(len(lens))
    PARITY_STATS["fallback_rows"] += int(fallback.sum())
    if len(lens):
        w = float((literal / np.maximum(scale, 1e-300)).max())
        PARITY_STATS["worst_literal"][dt.name] = max(PARITY_STATS["worst_literal"].get(dt.name, 0.0), w)
    bad = fails_literal & ~oracle_is_far
    assert not bad.any(), "%s: %d rows beyond |y_gpu - y_oracle| <= %g * sum|a x|, worst %g (row %d, len %d)" % (
        what, bad.sum(), tol, float((literal / np.maximum(scale, 1e-300))[bad].max()), int(np.argmax(bad)), int(lens[np.argmax(bad)]))
    bad = fallback & (err_gpu > err_ref + tol * scale + tiny)
    assert not bad.any(), "%s: %d rows out of the excess bound, worst excess %g * sum|a x| (row %d, len %d)" % (
        what, bad.sum(), excess.max(), int(np.argmax(excess)), int(lens[np.argmax(excess)]))
    denom = float(np.abs(y_ref.astype(np.float64)).max()) if len(y_ref) else 0.0
    if denom > 0:  # normwise, literal: max |y_gpu - y_oracle| <= tol * max |y_oracle| (else: no worse than the oracle against exact)
        if literal.max() / denom > tol:
            denom = float(np.abs(exact).max())
            assert err_gpu.max() / denom <= err_ref.max() / denom + tol, what
    return y_ref


def build_mock_rccl():
    """tests/mock_rccl/libmock_rccl.so (the stand-in for RCCL that lets several ranks share one device), rebuilt when stale."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "mock_rccl", "mock_rccl.cpp")
    so = os.path.join(root, "tests", "mock_rccl", "libmock_rccl.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", so,
                        "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return so
