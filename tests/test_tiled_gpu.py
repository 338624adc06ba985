"""K2t (spmv_tiled.hip): y = A x in two streaming passes over a 2-D tiled copy -- products against slices of x held in LDS,
folded per (row, slice) inside pass 1, then one wavefront per block of rows adding its tiles slice by slice.  Checked four ways:
the plan's INTEGER STRUCTURE (chunks, continuation bits, product slots, row blocks, row codes, tile table) array by array
against a numpy restatement of the build (tests/tiled_model.py); the result within the parity bound of the storage-order
oracle (the reference's loop, sparsematrix.rs:146-158); pass 1's products and y BIT-EXACT against the numpy restatement of
the kernels' own summation order; bitwise reproducible from launch to launch."""
import numpy as np
import pytest

import oracle
import sparsemat_amd as sm
from tiled_model import SLICE, TiledModel
from util import assert_spmv_close, random_crs

pytestmark = pytest.mark.gpu


def bits(a):
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


def check_against_model(m, off, col, val, x, n_rows, n_cols, what, structure=True):
    lay = m.tiled_layout(arrays=True)
    mod = TiledModel(off, col, val, n_rows, n_cols)
    if structure:
        assert (lay["n_slices"], lay["slice_columns"], lay["n_row_blocks"], lay["rows_per_block"]) == (mod.n_cb, SLICE, mod.n_rb, mod.R), what
        assert lay["copy_entries"] == mod.n_chunks * mod.CH and lay["n_products"] == mod.n_prod
        assert np.array_equal(lay["slice_chunks"], mod.slice_chunks) and np.array_equal(lay["chunks"], mod.chunks)
        assert np.array_equal(lay["codes"], mod.codes) and np.array_equal(bits(lay["values"]), bits(mod.values))
        assert np.array_equal(lay["row_block_start"], mod.rb_start) and np.array_equal(lay["product_rows"], mod.product_rows)
        assert np.array_equal(lay["tile_start"], mod.tile_start)
    y = m.mvp(x, variant="tiled")
    prod = mod.products(val, x)
    got = m.tiled_layout(arrays=True)["products"]
    assert np.array_equal(bits(got[mod.real]), bits(prod[mod.real])), what + ": pass 1's product sums"
    assert np.array_equal(bits(y), bits(mod.reduce(prod))), what + ": y in the kernels' own order"
    assert np.array_equal(bits(y), bits(m.mvp(x, variant="tiled"))), what + ": run-to-run"
    return y, lay


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("kind", ["short", "uniform32", "ragged", "skewed", "empty_heavy", "dense_rows"])
def test_tiled_structure_products_and_result(gpu, dtype, kind):
    rng = np.random.default_rng({"short": 30, "uniform32": 31, "ragged": 32, "skewed": 33, "empty_heavy": 34, "dense_rows": 38}[kind])
    n_rows, n_cols = 6007, 5 * SLICE - 331  # 5 column slices, the last one short
    if kind == "short":
        lens = rng.integers(0, 5, n_rows)
    elif kind == "uniform32":
        lens = np.full(n_rows, 32)                 # ~6 entries per (row, slice): every row is folded inside pass 1
    elif kind == "ragged":
        lens = rng.integers(0, 70, n_rows)
    elif kind == "skewed":
        lens = rng.integers(0, 6, n_rows)
        lens[17] = 9000                            # ~1800 entries of ONE row per slice: cut by 7 chunk boundaries, merged again in pass 2
        lens[4000:4100] = 300                      # ~60 per slice: runs longer than the snap distance
    elif kind == "dense_rows":
        lens = np.zeros(n_rows, dtype=np.int64)    # every tile is a handful of long runs (equal neighbours in every round of pass 2)
        lens[::97] = 2500
    else:
        lens = rng.integers(0, 5, n_rows)
        lens[rng.random(n_rows) < 0.8] = 0
    off, col, val = random_crs(rng, n_rows, n_cols, lens, dtype, dup=True)
    x = rng.uniform(-1, 1, n_cols).astype(dtype)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    y, lay = check_against_model(m, off, col, val, x, n_rows, n_cols, "tiled " + kind)
    assert_spmv_close(y, off, col, val, x, "tiled " + kind)
    assert lay["n_slices"] == 5 and lay["rows_per_block"] <= (3328 if dtype == np.float32 else 1664)
    assert lay["n_products"] <= len(col) + 3 * len(lay["chunks"])  # never more product slots than entries (+ the 16-byte padding)
    assert m.resolved_variant()[0] != "tiled"  # x is tiny: AUTO never picks it here


def test_tiled_follows_value_updates_and_edge_shapes(gpu):
    f = np.float64
    rng = np.random.default_rng(36)
    n_rows, n_cols = 1500, 3 * SLICE + 5
    off, col, val = random_crs(rng, n_rows, n_cols, rng.integers(0, 20, n_rows), f)
    x = rng.uniform(-1, 1, n_cols).astype(f)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    assert_spmv_close(m.mvp(x, variant="tiled"), off, col, val, x, "before")
    mod = TiledModel(off, col, val, n_rows, n_cols)
    m.scale(-0.5)
    assert np.array_equal(bits(m.mvp(x, variant="tiled")), bits(mod.reduce(mod.products(val * -0.5, x))))
    val2 = rng.uniform(-1, 1, len(val)).astype(f)
    m.update_values(val2)
    want = mod.reduce(mod.products(val2, x))
    assert np.array_equal(bits(m.mvp(x, variant="tiled")), bits(want))
    # an x longer than n_cols is fine, a shorter one only if it covers the largest column (densevec.rs:40-42: the gather panics)
    xl = np.concatenate([x, np.ones(7, f)])
    assert np.array_equal(bits(m.mvp(xl, variant="tiled")), bits(want))
    with pytest.raises(sm.SparseMatPanic) as e:
        m.mvp(x[:int(col.max())], variant="tiled")
    assert "index out of bounds" in str(e.value)
    # no entries at all / a single entry in the last column / one row
    m = sm.SparseMatCRS.from_raw_parts(5, 300, [0, 0, 0, 0, 0, 0], [], np.array([], f))
    assert np.array_equal(m.mvp(np.ones(300, f), variant="tiled"), np.zeros(5, f))
    m = sm.SparseMatCRS.from_raw_parts(2, 2 * SLICE, [0, 0, 1], [2 * SLICE - 1], np.array([2.0], f))
    xx = np.arange(2 * SLICE, dtype=f)
    assert np.array_equal(m.mvp(xx, variant="tiled"), np.array([0.0, 2.0 * (2 * SLICE - 1)], f))
    cols = np.arange(0, 4 * SLICE, 97, dtype=np.uint32)
    m = sm.SparseMatCRS.from_raw_parts(1, 4 * SLICE, [0, len(cols)], cols, np.ones(len(cols), f))
    xx = rng.uniform(-1, 1, 4 * SLICE).astype(f)
    one = TiledModel(np.array([0, len(cols)]), cols, np.ones(len(cols), f), 1, 4 * SLICE)
    assert np.array_equal(bits(m.mvp(xx, variant="tiled")), bits(one.reduce(one.products(np.ones(len(cols), f), xx))))


def test_tiled_many_slices_and_row_blocks(gpu):
    """40 / 67 / 129 slices walk the reduce pass's 64-tile table window and its 4-tile batches through their edges; short rows
    make row blocks of thousands of rows (the cap) whose tiles hold a few dozen products."""
    rng = np.random.default_rng(37)
    for n_slices, dtype in ((40, np.float32), (67, np.float64), (129, np.float32)):
        n_rows, n_cols = 20011, n_slices * SLICE - 3
        off, col, val = random_crs(rng, n_rows, n_cols, rng.integers(0, 9, n_rows), dtype)
        x = rng.uniform(-1, 1, n_cols).astype(dtype)
        m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
        y, lay = check_against_model(m, off, col, val, x, n_rows, n_cols, "tiled %d slices" % n_slices)
        assert lay["n_slices"] == n_slices and lay["rows_per_block"] > 64
        assert_spmv_close(y, off, col, val, x, "tiled %d slices" % n_slices)


def test_tiled_tiles_of_several_rounds(gpu):
    """Row blocks hold target x n_slices products; with every column inside the FIRST of four slices all of them land in one tile:
    ~700 (f32) / ~240 (f64) products = 3 / 2 rounds of 64 E -- the reduce pass's later rounds (direct loads, masks at both ends)."""
    rng = np.random.default_rng(39)
    n_rows, n_cols = 9000, 4 * SLICE
    lens = rng.integers(1, 4, n_rows)
    for dtype in (np.float32, np.float64):
        off, col, val = random_crs(rng, n_rows, SLICE - 7, lens, dtype)
        x = rng.uniform(-1, 1, n_cols).astype(dtype)
        m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
        y, lay = check_against_model(m, off, col, val, x, n_rows, n_cols, "long tiles")
        tiles = np.diff(lay["tile_start"].astype(np.int64), axis=0)
        assert tiles[:, 0].max() > (256 if dtype == np.float32 else 128) and not tiles[:, 1:].any()
        assert_spmv_close(y, off, col, val, x, "long tiles")


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("chunks", [1009, 1013, 1024])
def test_tiled_wavefront_runs_of_64_chunks(gpu, dtype, chunks):
    """One slice of a 16-slice matrix holds every entry: pass 1 takes it as ONE part of 1009..1024 chunks, i.e. 64 chunks per
    wavefront -- the most a wavefront's descriptor register holds.  With 3 chunks in flight the loop's last trip looks at slots
    64 and 65 of the run, which must be empty (round 3's kernel took the run's first two descriptors for them and overwrote their
    product slots with the NEXT wavefront's chunks)."""
    rng = np.random.default_rng(400 + chunks)
    n_rows, n_cols = 8000, 16 * SLICE
    total = 240 * chunks - 57
    lens = np.full(n_rows, total // n_rows)
    lens[: total - lens.sum()] += 1
    off, col, val = random_crs(rng, n_rows, SLICE - 3, lens, dtype)
    col = (col + 3 * SLICE).astype(np.uint32)  # all of them in slice 3
    x = rng.uniform(-1, 1, n_cols).astype(dtype)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    y, lay = check_against_model(m, off, col, val, x, n_rows, n_cols, "64-chunk runs (%d)" % chunks)
    per_slice = np.diff(lay["slice_chunks"].astype(np.int64))
    assert per_slice.max() == per_slice.sum() and (per_slice.max() + 15) // 16 == 64, per_slice.max()
    assert_spmv_close(y, off, col, val, x, "64-chunk runs")


def test_tiled_refuses_columns_beyond_n_cols(gpu):
    """A handle made without validation may hold a column >= n_cols (smh_crs_create_dev's default); the tiled / column-blocked
    builders size their tables from n_cols and must refuse it instead of walking past them."""
    f = np.float32
    off = np.array([0, 2, 3], dtype=np.uint32)
    col = np.array([1, 70000, 5], dtype=np.uint32)  # n_cols says 100
    val = np.ones(3, f)
    m = sm.SparseMatCRS.from_raw_parts(2, 100, off, col, val, validate=False)
    x = np.ones(70001, f)
    for variant in ("tiled", "colblock", "colfused", "colsplit"):
        with pytest.raises(sm.SparseMatPanic) as e:
            m.mvp(x, variant=variant)
        assert ">= n_cols" in str(e.value), variant
    assert np.array_equal(m.mvp(x, variant="seq"), np.array([2.0, 1.0], f))  # the row-major kernels only need x to reach
