"""Randomised sweep over shapes near every tiling boundary of the kernels (64-row wave tiles, 256-row K1s tiles,
2048-item merge tiles, 4-entry chunks, borrowed/unpadded arrays), all kernel families, both dtypes."""
import numpy as np
import pytest

import oracle
import sparsemat_amd as sm
from sparsemat_amd import synth
from util import assert_spmv_close

pytestmark = pytest.mark.gpu

ROWS = [1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1000, 2047, 2048, 2049, 4100]


def _matrix(rng, n_rows, dtype):
    n_cols = int(rng.integers(1, 3 * n_rows + 5))
    style = rng.integers(0, 5)
    if style == 0:
        lens = rng.integers(0, 6, n_rows)
    elif style == 1:
        lens = rng.integers(0, 40, n_rows)
    elif style == 2:
        lens = np.full(n_rows, int(rng.integers(1, 34)))
    elif style == 3:
        lens = rng.integers(0, 4, n_rows)
        lens[rng.integers(0, n_rows)] = int(rng.integers(100, 5000))
    else:
        lens = (rng.pareto(1.2, n_rows) * 3).astype(np.int64).clip(0, 3000)
    off = np.zeros(n_rows + 1, dtype=np.uint32)
    np.cumsum(lens, out=off[1:])
    nnz = int(off[-1])
    if rng.random() < 0.5:  # banded-ish
        rows = np.repeat(np.arange(n_rows), lens)
        col = np.clip(rows * n_cols // max(n_rows, 1) + rng.integers(-50, 51, nnz), 0, n_cols - 1).astype(np.uint32)
    else:
        col = rng.integers(0, n_cols, nnz, dtype=np.uint32)
    val = rng.uniform(-1, 1, nnz).astype(dtype)
    x = rng.uniform(-1, 1, n_cols).astype(dtype)
    return n_cols, off, col, val, x


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_fuzz_all_kernels(gpu, dtype):
    rng = np.random.default_rng(20260101)
    bits = np.uint32 if dtype == np.float32 else np.uint64
    for trial, n_rows in enumerate(ROWS * 2):
        n_cols, off, col, val, x = _matrix(rng, n_rows, dtype)
        what = "trial %d: %d x %d, nnz %d" % (trial, n_rows, n_cols, len(val))
        y_ref = oracle.spmv(off, col, val, x)
        # owned (padded) copy and a borrowed, unpadded device copy of the same arrays
        owned = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
        bo, bc, bv = synth.DeviceBuffer(off.nbytes), synth.DeviceBuffer(max(col.nbytes, 4)), synth.DeviceBuffer(max(val.nbytes, 8))
        bo.upload(off)
        if len(val):
            bc.upload(col)
            bv.upload(val)
        borrowed = sm.SparseMatCRS.from_device_parts(n_rows, n_cols, len(val), bo.ptr, bc.ptr, bv.ptr, dtype, keep=(bo, bc, bv))
        for m, tag in ((owned, "owned"), (borrowed, "borrowed")):
            for variant in ("seq", "stream"):  # bit-exact kernels
                y = m.mvp(x, variant=variant)
                assert np.array_equal(y.view(bits), y_ref.view(bits)), (what, tag, variant)
            assert_spmv_close(m.mvp(x, variant="merge"), off, col, val, x, "%s %s merge" % (what, tag))
            assert_spmv_close(m.mvp(x, variant="auto"), off, col, val, x, "%s %s auto" % (what, tag))
            for lanes in (1, 4, 8, 32):
                m.set_vector_lanes(lanes)
                for ring in ((1, 0) if lanes <= 8 else (0,)):
                    m.set_ring(ring)
                    assert_spmv_close(m.mvp(x, variant="vector"), off, col, val, x,
                                      "%s %s vector lanes %d ring %d" % (what, tag, lanes, ring))
            m.set_vector_lanes(0)
            m.set_ring(-1)
            # column-blocked K2c with a random block width (1 .. ~30 blocks)
            shift = int(rng.integers(max(1, int(np.log2(max(n_cols, 2))) - 4), int(np.log2(max(n_cols, 2))) + 2))
            m.set_colblock_shift(shift)
            if (n_cols + (1 << shift) - 1) >> shift <= 128:
                assert_spmv_close(m.mvp(x, variant="colblock"), off, col, val, x, "%s %s colblock 2^%d" % (what, tag, shift))
            m.set_colblock_shift(0)


def test_fuzz_replay_transpose_prod_against_the_container_restatement(gpu):
    """Many small random cases for the widened rows: streams replayed on a SparseMatCRS, transpose
    and prod, each bit-exact against the literal C restatement of the reference's container -- shapes chosen so that
    every form of the first-push quirk, duplicates, empty rows and exact cancellations come up."""
    import oracle
    rng = np.random.default_rng(2024)

    def same(m, e):
        off, col, val = m.raw_parts()
        assert (m.n_rows(), m.n_cols(), len(col)) == (e[0], e[1], len(e[3]))
        assert np.array_equal(off[:e[0] + 1], e[2]) and np.array_equal(col, e[3]) and val.tobytes() == e[4].tobytes()

    quirks = 0
    for t in range(120):
        dt = np.float32 if t % 2 else np.float64
        # replay
        n = int(rng.integers(0, 120))
        n_r, n_c = int(rng.integers(1, 12)), int(rng.integers(1, 9))
        rows, cols = rng.integers(0, n_r, n), rng.integers(0, n_c, n)
        vals = rng.integers(-2, 3, n).astype(dt) if t % 3 == 0 else rng.uniform(-1, 1, n).astype(dt)
        ops = None if t % 4 == 0 else rng.integers(0, 2, n).astype(np.uint8)
        e = oracle.crs_replay(rows, cols, vals, ops)
        same(sm.SparseMatCRS.from_triplets(rows, cols, vals, ops, into_crs=True), e)
        quirks += e[5] != len(e[3])
        # transpose and prod of small random matrices (unsorted rows, duplicates, empty rows)
        a_rows, a_cols = int(rng.integers(1, 14)), int(rng.integers(1, 14))
        lens = rng.integers(0, 6, a_rows)
        lens[-1] = max(lens[-1], 1)
        off = np.zeros(a_rows + 1, np.uint32)
        np.cumsum(lens, out=off[1:])
        col = rng.integers(0, a_cols, int(off[-1])).astype(np.uint32)
        col[-1] = a_cols - 1
        val = rng.integers(-2, 3, len(col)).astype(dt) if t % 3 == 0 else rng.uniform(-1, 1, len(col)).astype(dt)
        a = sm.SparseMatCRS.from_raw_parts(a_rows, a_cols, off, col, val)
        et = oracle.transpose(off, col, val)
        same(a.transpose(), et)
        quirks += et[5] != len(et[3])
        # b with the dimensions the reference's prod demands: b.n_rows == a.n_cols, b.n_cols == a.n_rows
        lens_b = rng.integers(0, 6, a_cols)
        lens_b[-1] = max(lens_b[-1], 1)
        off_b = np.zeros(a_cols + 1, np.uint32)
        np.cumsum(lens_b, out=off_b[1:])
        col_b = rng.integers(0, a_rows, int(off_b[-1])).astype(np.uint32)
        col_b[-1] = a_rows - 1
        val_b = rng.integers(-2, 3, len(col_b)).astype(dt) if t % 3 == 0 else rng.uniform(-1, 1, len(col_b)).astype(dt)
        b = sm.SparseMatCRS.from_raw_parts(a_cols, a_rows, off_b, col_b, val_b)
        same(a.prod(b), oracle.prod((a_rows, a_cols, off, col, val), (a_cols, a_rows, off_b, col_b, val_b)))
    assert quirks > 5  # orphaned first entries did occur


def test_fuzz_column_tables_and_predicates(gpu):
    """Random small matrices (unsorted rows, duplicate columns, partly symmetric by construction): the column tables
    equal the oracle's restatement of assemble_column_info, is_sorted / is_symmetric equal the trait defaults written
    out in Python (get = first match in storage order, zero when absent: sparsemat_crs.rs:54-67,136-142)."""
    import oracle
    rng = np.random.default_rng(77)
    seen = {"sym": 0, "asym": 0, "sorted": 0, "unsorted": 0}
    for t in range(150):
        n = int(rng.integers(1, 10))
        dense = np.zeros((n, n))
        if t % 3 == 0:  # symmetric pattern and values
            m = rng.integers(-2, 3, (n, n)).astype(np.float64) * (rng.random((n, n)) < 0.4)
            dense = np.triu(m) + np.triu(m, 1).T
        else:
            dense = rng.integers(-2, 3, (n, n)).astype(np.float64) * (rng.random((n, n)) < 0.35)
        rows_l, cols_l, vals_l = [], [], []
        off = [0]
        for i in range(n):
            js = [j for j in range(n) if dense[i, j] != 0 or rng.random() < 0.05]  # a few stored zeros
            if t % 2:
                js = list(rng.permutation(js))  # storage order
            if js and rng.random() < 0.2:
                js.append(js[0])  # a duplicate column; get() sees the first one only
            for j in js:
                cols_l.append(int(j))
                vals_l.append(dense[i, j])
            off.append(len(cols_l))
        off = np.array(off, np.uint32)
        col = np.array(cols_l, np.uint32)
        val = np.array(vals_l, np.float64)
        a = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val, validate=False)

        def get(i, j):
            if i >= n:
                return 0.0
            for q in range(off[i], off[i + 1]):
                if col[q] == j:
                    return val[q]
            return 0.0
        want_sym = all(get(int(col[q]), i) == val[q] for i in range(n) for q in range(off[i], off[i + 1]))
        want_sorted = all(col[q] >= col[q - 1] for i in range(n) for q in range(off[i] + 1, off[i + 1]))
        assert a.is_symmetric() == want_sym and a.is_sorted() == want_sorted
        seen["sym" if want_sym else "asym"] += 1
        seen["sorted" if want_sorted else "unsorted"] += 1
        rows, col_ptr, entries = a.column_info()
        e_rows, e_ptr, e_entries = oracle.column_info(off, col, n)
        assert np.array_equal(rows, e_rows) and np.array_equal(col_ptr, e_ptr) and np.array_equal(entries, e_entries)
    assert min(seen.values()) > 10, seen


def _structured(rng, dtype):
    """A matrix with local structure: row r takes len_r distinct columns from a window around slope * r (plus, sometimes, a few
    far diagonals like a stencil), stored shuffled; some rows empty, sometimes one very long row or dense tile."""
    n_rows = int(rng.choice([1, 2, 255, 256, 257, 1000, 4097, 20_000, 70_001]))
    slope = float(rng.choice([0.03, 0.5, 1.0, 1.0, 3.0, 9.0]))
    w = int(rng.choice([0, 1, 7, 40, 300]))
    max_len = int(rng.choice([1, 3, 8, 33, 70]))
    n_cols = int(slope * n_rows) + w + 2
    width = 2 * w + 1
    lens = rng.integers(0, min(max_len, width) + 1, n_rows)
    if rng.integers(0, 3) == 0:
        lens[rng.integers(0, n_rows)] = min(width, int(rng.integers(1, 600)))
    order = np.argsort(rng.random((n_rows, width)), axis=1)          # a random subset of the window per row, in random order
    keep = np.arange(width)[None, :] < lens[:, None]
    rows_idx = np.repeat(np.arange(n_rows), lens)
    cols = (np.floor(slope * np.arange(n_rows))[:, None].astype(np.int64) + order - w)[keep]
    far = int(rng.choice([0, 0, 17, 4096])) if n_rows > 300 else 0
    cols = np.where((far > 0) & (rng.random(len(cols)) < 0.2), cols + far * rng.integers(-1, 2, len(cols)), cols)
    cols = np.clip(cols, 0, n_cols - 1)
    off = np.zeros(n_rows + 1, np.uint32)
    np.cumsum(lens, out=off[1:])
    # clipping and the far diagonals may have produced a repeated (row, column) pair: that is a legal input (general route)
    val = rng.uniform(-1, 1, len(cols)).astype(dtype)
    return n_rows, n_cols, off, cols.astype(np.uint32), val, rows_idx


def test_fuzz_transpose_routes_agree(gpu):
    """150 structured matrices: the bucketed transposition and the general route give the same arrays bit for bit, whichever
    route the first call took, and so do the column tables by buckets and by the stable sort; small ones also against the
    literal restatement of the reference's container."""
    import os
    rng = np.random.default_rng(20261004)
    taken = {"bucketed": 0, "general": 0}
    for it in range(150):
        dtype = np.float32 if it % 2 else np.float64
        n_rows, n_cols, off, col, val, _ = _structured(rng, dtype)
        if len(col) == 0:
            continue
        m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
        os.environ.pop("SMH_TRANSPOSE_BUCKETED", None)
        t = m.transpose()
        route = sm.SparseMatCRS.last_transpose_route()
        taken[route] += 1
        os.environ["SMH_TRANSPOSE_BUCKETED"] = "0"
        try:
            g = m.transpose()
        finally:
            os.environ.pop("SMH_TRANSPOSE_BUCKETED", None)
        what = "case %d: %d x %d, %d entries, route %s" % (it, n_rows, n_cols, len(col), route)
        assert (t.n_rows(), t.n_cols(), t.n_non_zero_entries(), t.orphans()) == (g.n_rows(), g.n_cols(), g.n_non_zero_entries(), g.orphans()), what
        for a, b in zip(t.raw_parts(), g.raw_parts()):
            assert a.tobytes() == b.tobytes(), what
        # the column tables: the bucketed passes (keys = entry indices, ascending) against the device-wide stable sort
        info = m.column_info()
        os.environ["SMH_COLUMN_INFO_BUCKETED"] = "0"
        try:
            info_sorted = m.column_info()
        finally:
            os.environ.pop("SMH_COLUMN_INFO_BUCKETED", None)
        for a, b in zip(info, info_sorted):
            assert np.array_equal(a, b), what
        if len(col) <= 6000:
            e_rows, e_ptr, e_entries = oracle.column_info(off, col, n_cols)
            assert np.array_equal(info[0], e_rows) and np.array_equal(info[1], e_ptr) and np.array_equal(info[2], e_entries), what
        if len(col) <= 6000:
            e = oracle.transpose(off, col, val)
            t_off, t_col, t_val = t.raw_parts()
            assert (t.n_rows(), t.n_cols()) == (e[0], e[1]) and np.array_equal(t_off[:e[0] + 1], e[2]) and np.array_equal(t_col, e[3]), what
            assert t_val.tobytes() == e[4].tobytes(), what
    assert taken["bucketed"] >= 40 and taken["general"] >= 10, taken


def _short_row_matrix(rng, n_rows, dtype):
    """Rows of 0..9 entries (a few longer), columns in up to four clusters around the diagonal: the shapes K1s XS / XD stage x for."""
    n_cols = n_rows + int(rng.integers(0, 50))
    lens = rng.integers(0, int(rng.integers(2, 10)), n_rows)
    if rng.random() < 0.5:
        lens[rng.integers(0, n_rows, max(1, n_rows // 300))] = rng.integers(9, 60)   # rows beyond the eight masked adds
    if rng.random() < 0.3:
        lens[int(rng.integers(0, n_rows)):][:int(rng.integers(1, 700))] = 0               # a run of empty rows (whole tiles of them)
    off = np.zeros(n_rows + 1, dtype=np.uint32)
    np.cumsum(lens, out=off[1:])
    nnz = int(off[-1])
    rows = np.repeat(np.arange(n_rows), lens)
    arms = np.array([0] + [int(v) for v in rng.integers(-3000, 3000, int(rng.integers(0, 4)))])
    col = np.clip(rows + arms[rng.integers(0, len(arms), nnz)] + rng.integers(-40, 41, nnz), 0, n_cols - 1).astype(np.uint32)
    return n_cols, off, col, rng.uniform(-1, 1, nnz).astype(dtype), rng.uniform(-1, 1, n_cols).astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_fuzz_stream_with_x_staged(gpu, dtype):
    """K1s with x staged in LDS, both bodies (column codes / stage offsets), forced on random short-row matrices around the tile
    boundaries: bit-exact against the oracle whatever the layout decides (a matrix whose tiles do not fit a stage keeps the gathers)."""
    rng = np.random.default_rng(777)
    bits = np.uint32 if dtype == np.float32 else np.uint64
    trials = int(__import__("os").environ.get("SMH_FUZZ_TRIALS", "40"))
    staged = 0
    for trial in range(trials):
        n_rows = int(rng.choice([255, 256, 257, 511, 513, 1000, 4099, 20000, 70001]))
        n_cols, off, col, val, x = _short_row_matrix(rng, n_rows, dtype)
        what = "trial %d: %d x %d, nnz %d" % (trial, n_rows, n_cols, len(val))
        y_ref = oracle.spmv(off, col, val, x)
        m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
        lhs = rng.uniform(-1, 1, n_rows).astype(dtype)
        ip_plain = m.inner_prod(lhs, x, variant="stream")
        m.set_stream_xs(1)
        for direct in (0, 1, -1):
            m.set_stream_direct(direct)
            y = m.mvp(x, variant="stream")
            assert np.array_equal(y.view(bits), y_ref.view(bits)), (what, direct, m.stream_layout(), m.stream_direct())
            assert m.inner_prod(lhs, x, variant="stream") == ip_plain, (what, direct)
        staged += m.stream_layout()["xs_chunks"] != 0
    assert staged >= trials // 2   # most of these matrices are what the staged bodies are for


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_fuzz_tiled(gpu, dtype):
    """K2t on random shapes over 1..6 column slices (row lengths from empty to thousands, banded or scattered columns, duplicates):
    within the parity bound, and the same bits on a second launch."""
    rng = np.random.default_rng(4242)
    bits = np.uint32 if dtype == np.float32 else np.uint64
    trials = int(__import__("os").environ.get("SMH_FUZZ_TRIALS", "24"))
    for trial in range(trials):
        n_rows = int(rng.choice([1, 63, 257, 1000, 4100, 9001, 30000]))
        n_cols, off, col, val, x = _matrix(rng, n_rows, dtype)
        # stretch the columns over several slices (the generator stays below 3 n_rows + 5 columns)
        stretch = int(rng.integers(1, 1 + max(1, (6 * 16384) // max(n_cols, 1))))
        n_cols2 = n_cols * stretch
        col2 = (col.astype(np.int64) * stretch + rng.integers(0, stretch, len(col))).astype(np.uint32)
        x2 = rng.uniform(-1, 1, n_cols2).astype(dtype)
        what = "trial %d: %d x %d, nnz %d" % (trial, n_rows, n_cols2, len(val))
        m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols2, off, col2, val)
        y = m.mvp(x2, variant="tiled")
        assert_spmv_close(y, off, col2, val, x2, what + " tiled")
        assert np.array_equal(y.view(bits), m.mvp(x2, variant="tiled").view(bits)), what
