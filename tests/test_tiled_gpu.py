"""K2t (spmv_tiled.hip): y = A x in two streaming passes over a 2-D tiled copy -- products against slices of x held in
LDS, then one wavefront per block of rows folding its tiles slice by slice.  Checked three ways: within the parity bound
of the storage-order oracle (the reference's loop, sparsematrix.rs:146-158); BIT-EXACT against a numpy restatement of the
kernel's own summation order (per row: slices ascending, storage order inside a slice, each slice's sum added to the row's
running sum); bitwise reproducible from launch to launch."""
import numpy as np
import pytest

import oracle
import sparsemat_amd as sm
from util import assert_spmv_close, random_crs

pytestmark = pytest.mark.gpu

SLICE = 16384


def bits(a):
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


def tiled_reference(off, col, val, x, n_rows):
    """The kernel's order in the value type: prod = round(val * x[col]); per (row, slice) a left fold from the first product;
    y = ((0 + S_slice_a) + S_slice_b) + ... over the slices the row touches, ascending."""
    dt = val.dtype.type
    y = np.zeros(n_rows, dtype=val.dtype)
    prod = (val * x[col]).astype(val.dtype)
    sl = (col // SLICE).astype(np.int64)
    for i in range(n_rows):
        a, e = int(off[i]), int(off[i + 1])
        if a == e:
            continue
        s_i, p_i = sl[a:e], prod[a:e]
        acc = dt(0)
        for b in np.unique(s_i):
            part = p_i[s_i == b]
            s = part[0]
            for v in part[1:]:
                s = dt(s + v)
            acc = dt(acc + s)
        y[i] = acc
    return y


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("kind", ["short", "uniform32", "ragged", "skewed", "empty_heavy"])
def test_tiled_product_in_its_own_order_and_within_the_bound(gpu, dtype, kind):
    rng = np.random.default_rng({"short": 30, "uniform32": 31, "ragged": 32, "skewed": 33, "empty_heavy": 34}[kind])
    n_rows, n_cols = 6007, 5 * SLICE - 331  # 5 column slices, the last one short
    if kind == "short":
        lens = rng.integers(0, 5, n_rows)          # tiles of ~50 entries: one round per tile
    elif kind == "uniform32":
        lens = np.full(n_rows, 32)                 # 64-row blocks hold ~400 entries per tile: several rounds, runs across them
    elif kind == "ragged":
        lens = rng.integers(0, 70, n_rows)
    elif kind == "skewed":
        lens = rng.integers(0, 6, n_rows)
        lens[17] = 9000                            # ~1800 entries of ONE row in every tile it touches
        lens[4000:4100] = 300
    else:
        lens = rng.integers(0, 5, n_rows)
        lens[rng.random(n_rows) < 0.8] = 0
    off, col, val = random_crs(rng, n_rows, n_cols, lens, dtype, dup=True)
    x = rng.uniform(-1, 1, n_cols).astype(dtype)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    lay = m.tiled_layout()
    assert lay["n_slices"] == 5 and lay["slice_columns"] == SLICE
    # row blocks of equal entry counts (~48 per tile): never fewer blocks than the largest one would give
    assert lay["n_row_blocks"] >= -(-n_rows // lay["rows_per_block"]) and lay["rows_per_block"] <= 3072
    if len(col):
        assert 0.5 < len(col) / (lay["n_row_blocks"] * 5 * 48.0) < 1.5 or kind in ("skewed", "empty_heavy")
    cnt = np.bincount((col // SLICE).astype(np.int64), minlength=5)
    assert lay["copy_entries"] == int(((cnt + 7) // 8 * 8).sum())
    y = m.mvp(x, variant="tiled")
    assert_spmv_close(y, off, col, val, x, "tiled " + kind)
    assert np.array_equal(bits(y), bits(tiled_reference(off, col, val, x, n_rows)))
    assert np.array_equal(bits(y), bits(m.mvp(x, variant="tiled")))  # run-to-run reproducible
    assert m.resolved_variant()[0] != "tiled"  # x is tiny: AUTO never picks it here


def test_tiled_single_slice_equals_the_reference_order(gpu):
    """One column slice: the kernel's order IS the reference's (storage order from the first entry) -> the oracle's bits."""
    rng = np.random.default_rng(35)
    n_rows, n_cols = 3001, 9000
    off, col, val = random_crs(rng, n_rows, n_cols, rng.integers(0, 40, n_rows), np.float32, dup=True)
    x = rng.uniform(-1, 1, n_cols).astype(np.float32)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    assert m.tiled_layout()["n_slices"] == 1
    y = m.mvp(x, variant="tiled")
    # the oracle starts from +0 and adds the first product (0 + p = p exactly, except that -0 becomes +0)
    ref = oracle.spmv(off, col, val, x)
    assert np.array_equal(bits(y + np.float32(0)), bits(ref + np.float32(0)))


def test_tiled_follows_value_updates_and_edge_shapes(gpu):
    f = np.float64
    rng = np.random.default_rng(36)
    n_rows, n_cols = 1500, 3 * SLICE + 5
    off, col, val = random_crs(rng, n_rows, n_cols, rng.integers(0, 20, n_rows), f)
    x = rng.uniform(-1, 1, n_cols).astype(f)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    assert_spmv_close(m.mvp(x, variant="tiled"), off, col, val, x, "before")
    m.scale(-0.5)
    y = m.mvp(x, variant="tiled")
    assert np.array_equal(bits(y), bits(tiled_reference(off, col, (val * -0.5), x, n_rows)))
    val2 = rng.uniform(-1, 1, len(val)).astype(f)
    m.update_values(val2)
    assert np.array_equal(bits(m.mvp(x, variant="tiled")), bits(tiled_reference(off, col, val2, x, n_rows)))
    # an x longer than n_cols is fine, a shorter one only if it covers the largest column (densevec.rs:40-42: the gather panics)
    xl = np.concatenate([x, np.ones(7, f)])
    assert np.array_equal(bits(m.mvp(xl, variant="tiled")), bits(tiled_reference(off, col, val2, x, n_rows)))
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
    assert np.array_equal(bits(m.mvp(xx, variant="tiled")), bits(tiled_reference(np.array([0, len(cols)]), cols, np.ones(len(cols), f), xx, 1)))


def test_tiled_rows_whose_block_is_not_a_power_of_two_and_many_slices(gpu):
    """Rows per block follow the density (here ~48 entries per tile -> a few hundred rows, not a power of two); 40 slices
    walk the reduce pass's 64-tile table window and its 8-tile batches through their edges (40 = 5 batches)."""
    rng = np.random.default_rng(37)
    for n_slices, dtype in ((40, np.float32), (67, np.float64), (129, np.float32)):
        n_rows, n_cols = 20011, n_slices * SLICE - 3
        off, col, val = random_crs(rng, n_rows, n_cols, rng.integers(0, 9, n_rows), dtype)
        x = rng.uniform(-1, 1, n_cols).astype(dtype)
        m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
        lay = m.tiled_layout()
        assert lay["n_slices"] == n_slices and lay["rows_per_block"] > 64
        y = m.mvp(x, variant="tiled")
        assert_spmv_close(y, off, col, val, x, "tiled %d slices" % n_slices)
        assert np.array_equal(bits(y), bits(tiled_reference(off, col, val, x, n_rows)))
