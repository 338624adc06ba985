"""SURVEY 8f rank 4, the part next to the hot path: operation streams replayed on a SparseMatCRS itself (smh_crs_replay:
rows in reverse order of first appearance, first-push quirk), SparseMatrix::transpose (sparsematrix.rs:174-184) and the
column tables of ColumnIter (sparsemat_crs.rs:180-204) -- all on the device, BIT-EXACT against the literal C restatement
of the reference's container (oracle.crs_replay, pinned in test_oracle_golden.py to the reference's own CRS fixture
and to the pure-Python model of the container)."""
import json
import os
import zlib

import numpy as np
import pytest
import scipy.sparse as sp

import oracle
import sparsemat_amd as sm
from sparsemat_amd import _lib

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")


def same_crs(m, expect):
    n_rows, n_cols, off, col, val = expect[:5]
    assert (m.n_rows(), m.n_cols(), m.n_non_zero_entries()) == (n_rows, n_cols, len(col))
    g_off, g_col, g_val = m.raw_parts()
    assert np.array_equal(g_off[:n_rows + 1], off)
    assert np.array_equal(g_col, col)
    assert g_val.tobytes() == val.tobytes()


def test_reference_crs_kat_stream_on_device(gpu):
    """src/lib.rs:114-154: add_to calls on a SparseMatCRS -> its arrays (push prepends) and assert_eq!(mvp.get(0), 20.16)."""
    with open(GOLDEN) as f:
        case = [c for c in json.load(f)["cases"] if c["name"] == "check_sparsemat_crs"][0]
    ops = case["ops"]
    vals = np.array([np.float32(float(o[3])) for o in ops], np.float32)
    m = sm.SparseMatCRS.from_triplets([o[1] for o in ops], [o[2] for o in ops], vals, [1 if o[0] == "set" else 0 for o in ops],
                                      into_crs=True)
    crs = case["crs"]
    val = np.array([int(b, 16) for b in crs["values_bits"]], np.uint32).view(np.float32)
    same_crs(m, (crs["n_rows"], crs["n_cols"], np.array(crs["offset_rows"], np.uint32), np.array(crs["columns"], np.uint32), val))
    x = np.array([np.float32(float(s)) for s in case["x"]], np.float32)
    y = m.mvp(x, variant="stream")
    for i, lit in case["expect_mvp"]:
        assert y[i] == np.float32(float(lit))
    # src/lib.rs:137-148: sp_crs.iter_col(2) after assemble_column_info(), iter_row(0), iter_row(5); :153 density
    info = m.column_info()
    for j, want in case["expect_iter_col"]:
        assert m.iter_col(j, info) == [(r, float(np.float32(float(lit)))) for r, lit in want]
    for i, want in case["expect_iter_row"]:
        assert [(int(c), float(v)) for c, v in m.iter_row(i)] == [(c, float(np.float32(float(lit)))) for c, lit in want]
    num, den = case["expect_density"]
    assert m.density() == num / den


def stream(rng, n, n_r, n_c, dtype, head):
    rows, cols = rng.integers(0, n_r, n), rng.integers(0, n_c, n)
    if head == "second_row_smaller" and n >= 2:      # Vec::resize truncates: the first entry is orphaned
        rows[0], rows[1] = n_r - 1, 0
    elif head == "twin" and n >= 2:                  # the first operation keeps an entry of its own
        rows[1], cols[1] = rows[0], cols[0]
    elif head == "same_row_other_column" and n >= 2:
        rows[1], cols[1] = rows[0], (cols[0] + 1) % n_c
        if n_c == 1:
            rows[1] = rows[0]
    elif head == "second_row_larger" and n >= 2:
        rows[0], rows[1] = 0, n_r - 1
    vals = rng.uniform(-1, 1, n).astype(dtype)
    vals[rng.random(n) < 0.05] = dtype(-0.0)
    ops = (rng.random(n) < 0.4).astype(np.uint8)
    return rows, cols, vals, ops


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("head", ["random", "second_row_smaller", "twin", "same_row_other_column", "second_row_larger"])
@pytest.mark.parametrize("shape", ["short_rows", "long_rows"])
def test_replay_into_crs_bit_exact(gpu, dtype, head, shape):
    rng = np.random.default_rng(zlib.crc32((head + shape).encode()))
    if shape == "short_rows":
        n, n_r, n_c = 6000, 500, 60      # one thread replays a row
    else:
        n, n_r, n_c = 9000, 3, 40        # > 2048 operations per row: the segmented-sort route
    rows, cols, vals, ops = stream(rng, n, n_r, n_c, dtype, head)
    for kinds in (ops, None):
        m = sm.SparseMatCRS.from_triplets(rows, cols, vals, kinds, into_crs=True)
        expect = oracle.crs_replay(rows, cols, vals, kinds)
        same_crs(m, expect)
        x = rng.uniform(-1, 1, expect[1]).astype(dtype)
        assert m.mvp(x, variant="stream").tobytes() == oracle.spmv(expect[2], expect[3], expect[4], x).tobytes()


def test_replay_into_crs_tiny_streams(gpu):
    f = np.float32
    # one operation: n_rows stays 0 (sparsemat_crs.rs:75-76) -- nothing reachable, the column count is kept
    m = sm.SparseMatCRS.from_triplets([5], [3], np.array([1.5], f), into_crs=True)
    e = oracle.crs_replay([5], [3], np.array([1.5], f))
    assert (m.n_rows(), m.n_cols(), m.n_non_zero_entries()) == (0, 4, 0) == (e[0], e[1], len(e[3])) and e[5] == 1
    for rows, cols in (([5, 2], [3, 1]), ([2, 5], [3, 1]), ([2, 2], [3, 3]), ([2, 2], [3, 1]), ([0, 0, 0], [0, 0, 0])):
        for ops in (None, [1] * len(rows), [0, 1, 0][:len(rows)]):
            vals = np.array([1.5, -0.0, 2.25][:len(rows)], f)
            m = sm.SparseMatCRS.from_triplets(rows, cols, vals, ops, into_crs=True)
            same_crs(m, oracle.crs_replay(rows, cols, vals, ops))
    m = sm.SparseMatCRS.from_triplets([], [], np.array([], f), into_crs=True)
    assert (m.n_rows(), m.n_cols(), m.n_non_zero_entries()) == (0, 0, 0)


def random_crs(rng, n_rows, n_cols, max_len, dtype, sorted_unique):
    lens = rng.integers(0, max_len + 1, n_rows)
    off = np.zeros(n_rows + 1, np.uint32)
    np.cumsum(lens, out=off[1:])
    col = np.empty(int(off[-1]), np.uint32)
    for r in range(n_rows):
        a, b = int(off[r]), int(off[r + 1])
        if sorted_unique:
            col[a:b] = np.sort(rng.choice(n_cols, b - a, replace=False))
        else:
            col[a:b] = rng.integers(0, n_cols, b - a)  # unsorted, duplicates
    val = rng.uniform(-1, 1, len(col)).astype(dtype)
    return off, col, val


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("kind", ["sorted_unique", "storage_order_with_duplicates", "first_row_descending"])
def test_transpose_bit_exact(gpu, dtype, kind):
    rng = np.random.default_rng({"sorted_unique": 1, "storage_order_with_duplicates": 2, "first_row_descending": 3}[kind])
    n_rows, n_cols = 300, 170
    off, col, val = random_crs(rng, n_rows, n_cols, 24, dtype, kind != "storage_order_with_duplicates")
    if kind == "first_row_descending":  # the second stored entry has the smaller column: the quirk drops the first
        off = np.concatenate([np.zeros(1, np.uint32), off + np.uint32(3)])
        col = np.concatenate([np.array([9, 4, 1], np.uint32), col])
        val = np.concatenate([rng.uniform(-1, 1, 3).astype(dtype), val])
        n_rows += 1
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    t = m.transpose()
    expect = oracle.transpose(off, col, val)
    same_crs(t, expect)
    # the reference's n_non_zero_entries() (= columns.len()) counts an orphaned first entry: nnz + orphans on this side
    assert t.n_non_zero_entries() + t.orphans() == expect[5]
    if kind == "first_row_descending":
        assert expect[5] == len(expect[3]) + 1 and t.orphans() == 1  # the orphan the reference still counts
    else:
        assert t.orphans() == (1 if expect[5] != len(expect[3]) else 0)
    # (A^T) x through the hot path, bit-exact
    x = rng.uniform(-1, 1, n_rows).astype(dtype)
    assert t.mvp(x, variant="stream").tobytes() == oracle.spmv(expect[2], expect[3], expect[4], x).tobytes()


def test_transpose_empty_and_rowless(gpu):
    m = sm.SparseMatCRS.from_raw_parts(3, 4, np.zeros(4, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.float32))
    t = m.transpose()
    assert (t.n_rows(), t.n_cols(), t.n_non_zero_entries()) == (0, 0, 0)  # with_capacity + no set call


def test_transpose_at_scale_matches_scipy_structure(gpu):
    """2 M rows x 12 entries, sorted unique columns: the values are moved, never combined, so sort_rows(A^T) must equal
    scipy's transpose bit for bit; the rows of A^T itself list the source rows in descending order."""
    rng = np.random.default_rng(5)
    n, k = 2_000_000, 12
    base = (np.arange(n, dtype=np.int64)[:, None] * 7 + np.arange(k)[None, :] * 1009) % n
    base.sort(axis=1)
    keep = np.ones_like(base, bool)
    keep[:, 1:] = base[:, 1:] != base[:, :-1]
    lens = keep.sum(1)
    off = np.zeros(n + 1, np.uint32)
    np.cumsum(lens, out=off[1:])
    col = base[keep].astype(np.uint32)
    val = rng.uniform(-1, 1, len(col)).astype(np.float32)
    m = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val)
    t = m.transpose()
    ref = sp.csr_matrix((val, col, off), shape=(n, n)).T.tocsr()
    ref.sort_indices()
    assert (t.n_rows(), t.n_cols(), t.n_non_zero_entries()) == (int(col.max()) + 1, n, len(col))
    t_off, t_col, t_val = t.raw_parts()
    n_t = t.n_rows()
    assert np.array_equal(t_off, ref.indptr[:n_t + 1].astype(np.uint32))
    # descending inside every row
    inner = np.ones(len(t_col), bool)
    inner[t_off[1:-1][t_off[1:-1] < len(t_col)]] = False
    inner[0] = False
    assert np.all(t_col[1:][inner[1:]] < t_col[:-1][inner[1:]])
    t.sort_rows()
    _, s_col, s_val = t.raw_parts()
    assert np.array_equal(s_col, ref.indices.astype(np.uint32))
    assert s_val.tobytes() == ref.data.astype(np.float32).tobytes()
    # transposing twice gives A with every row reversed -- minus the entry the first-push quirk orphans: row 0 of A^T is
    # stored descending, so the second `set` goes to a smaller row than the first and offset_rows is truncated
    # (sparsemat_crs.rs:75-81); the lost entry is A^T's first stored one, (0, first) = A's (first, 0)
    assert t_off[1] >= 2 and t_col[1] < t_col[0]
    first = int(t_col[0])
    lost = int(off[first])
    assert col[lost] == 0
    e_off = off.copy()
    e_off[first + 1:] -= 1
    tt = m.transpose().transpose()
    tt.sort_rows()
    same_crs(tt, (n, n, e_off, np.delete(col, lost), np.delete(val, lost)))


@pytest.mark.parametrize("kind", ["sorted_unique", "storage_order_with_duplicates"])
def test_column_info_matches_restatement(gpu, kind):
    rng = np.random.default_rng(11)
    n_rows, n_cols = 400, 90
    off, col, val = random_crs(rng, n_rows, n_cols, 30, np.float64, kind == "sorted_unique")
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    rows, col_ptr, entries = m.column_info()
    e_rows, e_ptr, e_entries = oracle.column_info(off, col, n_cols)
    assert np.array_equal(rows, e_rows) and np.array_equal(col_ptr, e_ptr) and np.array_equal(entries, e_entries)
    info = (rows, col_ptr, entries)
    for j in (0, 17, n_cols - 1):
        want = [(int(e_rows[e]), float(val[e])) for e in e_entries[e_ptr[j]:e_ptr[j + 1]]]
        assert m.iter_col(j, info, val) == want


def test_column_info_at_scale_and_errors(gpu):
    rng = np.random.default_rng(12)
    n_rows, n_cols, nnz_row = 1_000_000, 300_000, 9
    off = (np.arange(n_rows + 1, dtype=np.uint64) * nnz_row).astype(np.uint32)
    col = rng.integers(0, n_cols, n_rows * nnz_row, dtype=np.uint32)
    val = np.ones(len(col), np.float32)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    rows, col_ptr, entries = m.column_info()
    assert np.array_equal(rows, np.repeat(np.arange(n_rows, dtype=np.uint32), nnz_row))
    assert np.array_equal(entries, np.argsort(col, kind="stable").astype(np.uint32))
    assert np.array_equal(col_ptr, np.searchsorted(np.sort(col), np.arange(n_cols + 1)).astype(np.uint32))
    # a column beyond n_cols has no list to go to
    bad = sm.SparseMatCRS.from_raw_parts(2, 3, np.array([0, 1, 2], np.uint32), np.array([1, 7], np.uint32),
                                         np.ones(2, np.float32), validate=False)
    with pytest.raises(sm.SparseMatPanic) as e:
        bad.column_info()
    assert e.value.status == _lib.SMH_ERR_INDEX_RANGE


# ---- the bucketed route (csrc/transpose_bucket.hip) -------------------------------------------------------------------------
def band_crs(rng, n_rows, n_cols, slope, below, above, max_len, dtype, shuffle):
    """rows with up to max_len distinct columns from [slope r - below, slope r + above] (clipped), empty rows included"""
    lens = rng.integers(0, max_len + 1, n_rows)
    cols, off = [], np.zeros(n_rows + 1, np.uint32)
    for r in range(n_rows):
        c0, c1 = max(0, int(slope * r) - below), min(n_cols, int(slope * r) + above + 1)
        k = min(int(lens[r]), max(0, c1 - c0))
        c = np.sort(rng.choice(np.arange(c0, c1), k, replace=False)) if k else np.zeros(0, np.int64)
        if shuffle:
            rng.shuffle(c)
        cols.append(c)
        off[r + 1] = off[r] + k
    col = np.concatenate(cols).astype(np.uint32)
    val = rng.uniform(-1, 1, len(col)).astype(dtype)
    return off, col, val


def transposed_both_routes(m):
    os.environ.pop("SMH_TRANSPOSE_BUCKETED", None)
    t = m.transpose()
    route = sm.SparseMatCRS.last_transpose_route()
    os.environ["SMH_TRANSPOSE_BUCKETED"] = "0"
    try:
        g = m.transpose()
        assert sm.SparseMatCRS.last_transpose_route() == "general"
    finally:
        os.environ.pop("SMH_TRANSPOSE_BUCKETED", None)
    assert (t.n_rows(), t.n_cols(), t.n_non_zero_entries(), t.orphans()) == (g.n_rows(), g.n_cols(), g.n_non_zero_entries(), g.orphans())
    for a, b in zip(t.raw_parts(), g.raw_parts()):
        assert a.tobytes() == b.tobytes()
    return t, route


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("shape", ["band", "band_storage_order", "tall", "wide", "steep_long_result_rows", "result_row_beyond_a_bucket", "repeat", "second_entry_smaller"])
def test_transpose_bucketed_route_bit_exact(gpu, dtype, shape):
    """Matrices with local structure take the two bucketed passes; the result is the general route's and the oracle's bit for
    bit.  Repeated pairs, buckets beyond the LDS capacity and the first-push quirk fall back."""
    rng = np.random.default_rng(zlib.crc32(shape.encode()))
    expect_route = "bucketed"
    if shape in ("band", "band_storage_order", "repeat", "second_entry_smaller"):
        n_rows, n_cols = 5000, 5100
        off, col, val = band_crs(rng, n_rows, n_cols, 1.0, 40, 60, 24, dtype, shape != "band")
        if shape == "repeat":  # one (row, column) pair twice: `set` keeps one entry with the later value
            r = 2500
            while off[r + 1] - off[r] < 2:
                r += 1
            col[off[r] + 1] = col[off[r]]
            expect_route = "general"
        if shape == "second_entry_smaller":
            r = 0
            while off[r + 1] - off[r] == 0:
                r += 1
            k = int(off[r])  # the first two stored entries (of one row or of two): the second gets the smaller column
            a, b = int(col[k]), int(col[k + 1])
            col[k], col[k + 1] = max(a, b) + (1 if a == b else 0), min(a, b)
            expect_route = "general"
    elif shape == "tall":      # result rows of ~100 entries: several entries per ranking lane
        n_rows, n_cols = 30000, 1000
        off, col, val = band_crs(rng, n_rows, n_cols, 1 / 30, 4, 4, 6, dtype, True)
        assert np.bincount(col).max() > 64
    elif shape == "wide":
        n_rows, n_cols = 1500, 12000
        off, col, val = band_crs(rng, n_rows, n_cols, 8.0, 30, 30, 40, dtype, True)
    elif shape == "steep_long_result_rows":   # 6000 rows into 12 columns: result rows of ~750 entries, buckets of 8 rows
        n_rows, n_cols = 6000, 12
        off, col, val = band_crs(rng, n_rows, n_cols, 12 / 6000, 1, 1, 3, dtype, True)
        assert np.bincount(col).max() > 512
    else:                      # 60000 rows into 3 columns: one result row alone is larger than a bucket may be
        n_rows, n_cols = 60000, 3
        off, col, val = band_crs(rng, n_rows, n_cols, 3 / 60000, 1, 1, 2, dtype, True)
        assert np.bincount(col).max() > 9728
        expect_route = "general"
    if shape != "second_entry_smaller" and col[1] < col[0]:  # (keep the container's first-push quirk out of the other shapes)
        col[0], col[1] = col[1], col[0]
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    t, route = transposed_both_routes(m)
    assert route == expect_route
    expect = oracle.transpose(off, col, val)
    same_crs(t, expect)
    assert t.n_non_zero_entries() + t.orphans() == expect[5]


def test_transpose_bucketed_at_scale(gpu):
    """1 M x 32 banded-stratified and window patterns (BASELINE C2's generators) and a 3-D Laplacian: both routes agree bit
    for bit; bands and stencils take the bucketed route, uniformly random columns (a tile reaches thousands of buckets) do not."""
    from sparsemat_amd import synth
    for pattern, dtype in ((synth.PATTERN_BANDED, np.float32), (synth.PATTERN_WINDOW, np.float64)):
        m = synth.crs_fixed(synth.SEED_MATRIX, pattern, 1_000_000, 32, dtype)
        t, route = transposed_both_routes(m)
        assert route == "bucketed"
        off, col, val = m.raw_parts()
        ref = sp.csr_matrix((val, col, off), shape=(m.n_rows(), m.n_cols())).T.tocsr()
        ref.sort_indices()
        t_off, t_col, t_val = t.raw_parts()
        assert np.array_equal(t_off, ref.indptr.astype(np.uint32))
        t.sort_rows()
        _, s_col, s_val = t.raw_parts()
        assert np.array_equal(s_col, ref.indices.astype(np.uint32)) and s_val.tobytes() == ref.data.astype(dtype).tobytes()
        rows, col_ptr, entries = m.column_info()  # (by buckets) = the stable order of the entry indices by column
        assert np.array_equal(entries, np.argsort(col, kind="stable").astype(np.uint32))
        assert np.array_equal(col_ptr, ref.indptr.astype(np.uint32)) and np.array_equal(rows, np.repeat(np.arange(m.n_rows(), dtype=np.uint32), 32))
    lap = synth.crs_laplace3d(96, 96, 96, np.float32)   # a stencil: seven diagonals, a handful of buckets per tile
    _, route = transposed_both_routes(lap)
    assert route == "bucketed"
    uni = synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_UNIFORM, 400_000, 16, np.float32)  # columns without locality
    _, route = transposed_both_routes(uni)
    assert route == "general"
