"""GPU parity of the SpMV kernels (K1 vector, K2 merge, SEQ) against the CPU oracle, through the C ABI.

Bar: integer structures bit-exact; SEQ variant bit-exact on values (same order, two roundings);
VECTOR / MERGE within 1e-5 (f32) / 1e-12 (f64) of sum_j |a_ij x_j| per row.
"""
import json
import os
import zlib

import numpy as np
import pytest

import oracle
import sparsemat_amd as sm
from sparsemat_amd import _lib
from util import assert_spmv_close, random_crs

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")
ALL_LANES = [1, 2, 4, 8, 16, 32, 64]


def _kats():
    with open(GOLDEN) as f:
        return [c for c in json.load(f)["cases"] if "expect_mvp" in c]


@pytest.mark.parametrize("case", _kats(), ids=lambda c: c["name"])
def test_reference_known_answers(gpu, case):
    """The reference's own asserted SpMV results (src/lib.rs:80-82,150-152,174-176,198-200)."""
    crs = case["crs"]
    dt = np.float32
    val = np.array([int(b, 16) for b in crs["values_bits"]], dtype=np.uint32).view(dt)
    x = np.array([dt(float(s)) for s in case["x"]], dtype=dt)
    m = sm.SparseMatCRS.from_raw_parts(crs["n_rows"], crs["n_cols"], crs["offset_rows"], crs["columns"], val)
    y_seq = m.mvp(x, variant="seq")
    assert len(y_seq) == crs["n_rows"]
    for i, lit in case["expect_mvp"]:
        assert y_seq[i] == dt(float(lit)), "SEQ must reproduce the reference literal exactly"
    y_ref = oracle.spmv(crs["offset_rows"], crs["columns"], val, x)
    assert np.array_equal(y_seq.view(np.uint32), y_ref.view(np.uint32))
    for variant in ("vector", "merge", "auto"):
        y = m.mvp(x, variant=variant)
        assert_spmv_close(y, np.array(crs["offset_rows"], np.uint32), np.array(crs["columns"], np.uint32), val, x,
                          "%s/%s" % (case["name"], variant))
        for i, lit in case["expect_mvp"]:
            assert abs(float(y[i]) - float(lit)) <= 1e-5 * abs(float(lit))
    # `A * v` sugar (sparsematrix.rs:435-443) == mvp
    assert np.array_equal(m * x, m.mvp(x))


def _lengths(rng, kind, n_rows):
    if kind == "uniform32":
        return np.full(n_rows, 32)
    if kind == "short":
        return rng.integers(0, 9, size=n_rows)
    if kind == "empty_heavy":
        l = rng.integers(0, 5, size=n_rows)
        l[rng.random(n_rows) < 0.6] = 0
        return l
    if kind == "skewed":
        l = rng.integers(1, 20, size=n_rows)
        l[rng.integers(0, n_rows, size=max(1, n_rows // 50))] = rng.integers(65, 300, size=max(1, n_rows // 50))
        l[rng.integers(0, n_rows)] = 2049 + 513
        l[rng.integers(0, n_rows)] = 5000
        return l
    if kind == "ragged":
        return rng.integers(0, 70, size=n_rows)
    raise ValueError(kind)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("kind", ["uniform32", "short", "empty_heavy", "skewed", "ragged"])
def test_random_matrices_all_variants(gpu, dtype, kind):
    rng = np.random.default_rng(zlib.crc32((kind + np.dtype(dtype).name).encode()))
    n_rows, n_cols = 3001, 2777  # n_rows != n_cols, not multiples of anything
    off, col, val = random_crs(rng, n_rows, n_cols, _lengths(rng, kind, n_rows), dtype, dup=True)
    x = rng.uniform(-1, 1, size=n_cols).astype(dtype)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    assert (m.n_rows(), m.n_cols(), m.n_non_zero_entries()) == (n_rows, n_cols, len(val))
    assert m.max_row_len() == int(np.diff(off.astype(np.int64)).max())
    # round trip of the integer structure is bit-exact
    off2, col2, val2 = m.raw_parts()
    assert np.array_equal(off, off2) and np.array_equal(col, col2) and np.array_equal(val, val2)
    # SEQ: bit-exact
    y_ref = oracle.spmv(off, col, val, x)
    y_seq = m.mvp(x, variant="seq")
    assert np.array_equal(y_seq.view(np.uint8), y_ref.view(np.uint8)), "SEQ kernel must be bit-exact"
    # empty rows give exactly 0
    assert np.all(y_seq[np.diff(off.astype(np.int64)) == 0] == 0)
    # MERGE and every VECTOR width
    assert_spmv_close(m.mvp(x, variant="merge"), off, col, val, x, "merge/" + kind)
    for lanes in ALL_LANES:
        m.set_vector_lanes(lanes)
        assert_spmv_close(m.mvp(x, variant="vector"), off, col, val, x, "vector%d/%s" % (lanes, kind))
    m.set_vector_lanes(0)
    assert_spmv_close(m.mvp(x, variant="auto"), off, col, val, x, "auto/" + kind)


def test_merge_is_bitwise_reproducible_and_table_matches_oracle(gpu):
    rng = np.random.default_rng(7)
    n_rows = 20011
    lens = rng.integers(0, 40, size=n_rows)
    lens[::997] = 3000
    off, col, val = random_crs(rng, n_rows, n_rows, lens, np.float64)
    x = rng.uniform(-1, 1, size=n_rows)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_rows, off, col, val)
    rows, nz, items = m.merge_table()
    diag = np.minimum(np.arange(len(rows), dtype=np.uint64) * np.uint64(items), np.uint64(n_rows + len(val)))
    rows_ref, nz_ref = oracle.merge_path_search(off, len(val), diag)
    assert np.array_equal(rows, rows_ref) and np.array_equal(nz, nz_ref), "merge-path coordinates must be bit-exact"
    y1 = m.mvp(x, variant="merge")
    y2 = m.mvp(x, variant="merge")
    assert np.array_equal(y1.view(np.uint64), y2.view(np.uint64)), "no float atomics: run-to-run bitwise equal"
    assert_spmv_close(y1, off, col, val, x, "merge skew")


def test_edge_shapes(gpu):
    f = np.float32
    # empty matrix
    m = sm.SparseMatCRS.from_raw_parts(0, 0, [], [], np.array([], f))
    assert m.mvp(np.array([], f)).shape == (0,)
    # rows but no entries
    m = sm.SparseMatCRS.from_raw_parts(5, 3, [0, 0, 0, 0, 0, 0], [], np.array([], f))
    for v in ("vector", "merge", "seq"):
        assert np.array_equal(m.mvp(np.ones(3, f), variant=v), np.zeros(5, f))
    # single entry, nnz not a multiple of 4 (array tail chunk), x longer than n_cols
    m = sm.SparseMatCRS.from_raw_parts(1, 2, [0, 1], [1], np.array([2.5], f))
    for v in ("vector", "merge", "seq"):
        assert np.array_equal(m.mvp(np.array([1, 3, 9], f), variant=v), np.array([7.5], f))
    # one very long row (> 2048) next to empty rows: wavefront-per-row (lanes 64) and merge
    rng = np.random.default_rng(3)
    lens = np.array([0, 0, 9001, 0, 1, 0])
    off, col, val = random_crs(rng, 6, 10000, lens, f)
    x = rng.uniform(-1, 1, 10000).astype(f)
    m = sm.SparseMatCRS.from_raw_parts(6, 10000, off, col, val)
    assert m.resolved_variant()[0] == "merge"
    for v in ("vector", "merge"):
        assert_spmv_close(m.mvp(x, variant=v), off, col, val, x, v)


def test_panics_of_the_reference(gpu):
    f = np.float32
    m = sm.SparseMatCRS.from_raw_parts(2, 4, [0, 1, 2], [0, 3], np.array([1, 2], f))
    # densevec.rs:41: x.dim() <= a column index -> index out of bounds
    with pytest.raises(sm.SparseMatPanic) as e:
        m.mvp(np.ones(3, f))
    assert e.value.status == _lib.SMH_ERR_INDEX_RANGE and "index out of bounds" in str(e.value)
    m.mvp(np.ones(4, f))
    # malformed structure is refused at create
    with pytest.raises(sm.SparseMatPanic):
        sm.SparseMatCRS.from_raw_parts(2, 4, [0, 2, 1], [0, 3], np.array([1, 2], f))
    with pytest.raises(sm.SparseMatPanic):
        sm.SparseMatCRS.from_raw_parts(2, 2, [0, 1, 2], [0, 3], np.array([1, 2], f))


def test_scale_and_update_values(gpu):
    rng = np.random.default_rng(11)
    off, col, val = random_crs(rng, 100, 100, rng.integers(0, 12, 100), np.float32)
    x = rng.uniform(-1, 1, 100).astype(np.float32)
    m = sm.SparseMatCRS.from_raw_parts(100, 100, off, col, val)
    m.scale(0.3)  # sparsemat_crs.rs:153-157: one rounding per value
    scaled = (val * np.float32(0.3)).astype(np.float32)
    assert np.array_equal(m.raw_parts()[2], scaled)
    assert np.array_equal(m.mvp(x, variant="seq"), oracle.spmv(off, col, scaled, x))
    m.update_values(val)
    assert np.array_equal(m.mvp(x, variant="seq"), oracle.spmv(off, col, val, x))


def test_device_vector_path_and_laplacians(gpu):
    # C1-shaped plumbing case: 5-point Laplacian (small grid), device-resident vectors
    off, col, val = oracle.laplace2d(37, 23, np.float32)
    n = 37 * 23
    m = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val)
    x = oracle.gen_x(sm.synth.SEED_X, n, np.float32)
    xv = sm.DenseVec.from_vec(x)
    for v in ("vector", "merge", "seq", "auto"):
        yv = m.mvp(xv, variant=v)
        assert yv.dim() == n
        assert_spmv_close(yv.to_numpy(), off, col, val, x, "laplace2d/" + v)
    off, col, val = oracle.laplace3d(9, 7, 5, np.float64)
    n = 9 * 7 * 5
    m = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val)
    x = oracle.gen_x(sm.synth.SEED_X, n, np.float64)
    assert_spmv_close(m.mvp(x), off, col, val, x, "laplace3d")
