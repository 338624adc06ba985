"""K1r (vector kernel with the LDS x-ring): plan invariants (integer structure) and value parity."""
import numpy as np
import pytest

import oracle
import sparsemat_amd as sm
from sparsemat_amd import synth
from util import assert_spmv_close, random_crs

pytestmark = pytest.mark.gpu
RING_ENTRIES = 16384


def check_plan(m, off, col):
    """Every row is covered exactly once, in order, by its block's phases; a ring phase's window
    (what the ring holds after its load) contains every column its rows reference."""
    n_rows = len(off) - 1
    ring = m.ring_entries()  # 16384 (64 KiB of f32, 128 KiB of f64) or, for f32 rows that need it, 32768
    assert ring in (RING_ENTRIES, 2 * RING_ENTRIES) and (ring == RING_ENTRIES or m.dtype == np.float32)
    nb, frac, active, ptr, ph = m.ring_plan()
    assert nb % 8 == 0 and len(ptr) == nb + 1 and ptr[0] == 0 and ptr[-1] == len(ph)
    assert np.all(np.diff(ptr.astype(np.int64)) >= 0)
    next_row = 0
    ring_rows = 0
    for b in range(nb):
        lo = hi = 0
        for p in range(ptr[b], ptr[b + 1]):
            rb, re, llo, lhi, use = (int(v) for v in ph[p])
            assert rb == next_row and re > rb and rb % 64 == 0
            next_row = re
            if llo < lhi:
                assert lhi - llo <= ring
                if llo == hi and hi > lo:  # continuation of the window
                    hi = lhi
                    lo = max(lo, hi - ring)
                else:  # restart
                    lo, hi = llo, lhi
            if use:
                ring_rows += re - rb
                cols = col[off[rb]:off[re]]
                if len(cols):
                    assert cols.min() >= lo and cols.max() < hi, (b, p, cols.min(), cols.max(), lo, hi)
                    assert hi - lo <= ring
    assert next_row == n_rows
    assert abs(frac - ring_rows / max(n_rows, 1)) < 1e-12
    return frac, active


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("pattern,n,k", [(synth.PATTERN_BANDED, 200_000, 32), (synth.PATTERN_DIAG, 70_001, 32),
                                          (synth.PATTERN_UNIFORM, 50_000, 32), (synth.PATTERN_BANDED, 3000, 7)])
def test_ring_plan_and_parity_generated(gpu, dtype, pattern, n, k):
    m = synth.crs_fixed(synth.SEED_MATRIX, pattern, n, k, dtype)
    off, col, val = m.raw_parts()
    x = oracle.gen_x(synth.SEED_X, n, dtype)
    frac, active = check_plan(m, off, col)
    if pattern != synth.PATTERN_UNIFORM:
        assert frac == 1.0 and active
    if pattern == synth.PATTERN_UNIFORM:
        assert frac == 0.0  # span of every tile exceeds the ring: all phases gather from L2
    for lanes in (1, 2, 4, 8):
        m.set_vector_lanes(lanes)
        for ring in (1, 0):
            m.set_ring(ring)
            assert_spmv_close(m.mvp(x, variant="vector"), off, col, val, x, "lanes%d ring%d" % (lanes, ring))


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_ring_mixed_phases_random(gpu, dtype):
    """Blocks of narrow-band rows interleaved with wide rows, empty rows and a window that jumps
    backwards: ring phases, restarts and global-gather phases in one matrix."""
    rng = np.random.default_rng(99)
    n_rows, n_cols = 40_000, 300_000
    lens = rng.integers(0, 40, size=n_rows)
    lens[5000:5600] = 0
    off = np.zeros(n_rows + 1, dtype=np.uint32)
    np.cumsum(lens, out=off[1:])
    col = np.empty(int(off[-1]), dtype=np.uint32)
    centers = np.linspace(0, n_cols - 1, n_rows)
    centers[20_000:30_000] = np.linspace(100_000, 0, 10_000)  # runs backwards
    for i in range(n_rows):
        a, b = off[i], off[i + 1]
        if 12_000 <= i < 12_128 or i % 1777 == 0:  # wide rows: span the whole vector
            col[a:b] = rng.integers(0, n_cols, size=b - a)
        else:
            lo = int(max(0, centers[i] - 1500))
            col[a:b] = rng.integers(lo, min(n_cols, lo + 3000), size=b - a)
    val = rng.uniform(-1, 1, size=len(col)).astype(dtype)
    x = rng.uniform(-1, 1, size=n_cols).astype(dtype)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    frac, _ = check_plan(m, off, col)
    assert 0.5 < frac < 1.0
    for lanes in (2, 8):
        m.set_vector_lanes(lanes)
        m.set_ring(1)
        y = m.mvp(x, variant="vector")
        assert_spmv_close(y, off, col, val, x, "mixed lanes%d" % lanes)
        assert np.array_equal(y, m.mvp(x, variant="vector")), "ring kernel is bitwise reproducible"
        # ring phases stream 16-bit columns (low halves; columns here go up to 300 000): the very same arithmetic
        import os
        try:
            os.environ["SMH_RING_COL16"] = "0"
            y32 = m.mvp(x, variant="vector")
            os.environ["SMH_RING_COL16"] = "1"
            y16 = m.mvp(x, variant="vector")
        finally:
            os.environ.pop("SMH_RING_COL16", None)
        assert np.array_equal(y32, y) and np.array_equal(y16, y)
    # sort_rows reorders the columns: the 16-bit copy must follow
    m.sort_rows()
    s_col, s_val = oracle.crs_sort_rows(off, col, val)
    assert_spmv_close(m.mvp(x, variant="vector"), off, s_col, s_val, x, "after sort_rows")


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_wide_ring_for_f32_bands_beyond_16384_columns(gpu, dtype):
    """Rows whose columns span +-10000 around the diagonal do not fit the 16384-column ring: an f32 matrix takes the
    32768-column ring (128 KiB of LDS), an f64 matrix keeps the narrow plan (global gathers) -- parity either way,
    and the plan invariants hold for the ring size the plan reports."""
    rng = np.random.default_rng(123)
    n = 150_000
    lens = rng.integers(20, 45, n)
    off = np.zeros(n + 1, np.uint32)
    np.cumsum(lens, out=off[1:])
    centers = np.repeat(np.arange(n), lens)
    col = np.clip(centers + rng.integers(-10_000, 10_000, len(centers)), 0, n - 1).astype(np.uint32)
    val = rng.uniform(-1, 1, len(col)).astype(dtype)
    x = rng.uniform(-1, 1, n).astype(dtype)
    m = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val)
    frac, _ = check_plan(m, off, col)
    if dtype == np.float32:
        assert m.ring_entries() == 2 * RING_ENTRIES and frac > 0.9
    else:
        assert m.ring_entries() == RING_ENTRIES and frac < 0.1
    for lanes in (4, 8, 16):
        m.set_vector_lanes(lanes)
        y = m.mvp(x, variant="vector")
        assert_spmv_close(y, off, col, val, x, "wide ring lanes %d" % lanes)
        assert np.array_equal(y, m.mvp(x, variant="vector"))


def test_ring_small_and_edge_shapes(gpu):
    f = np.float32
    rng = np.random.default_rng(4)
    for n_rows, n_cols in [(1, 1), (63, 10), (64, 64), (65, 17_000), (129, 5)]:
        lens = rng.integers(0, 70, size=n_rows)
        off, col, val = random_crs(rng, n_rows, n_cols, lens, f, dup=True)
        x = rng.uniform(-1, 1, n_cols).astype(f)
        m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
        check_plan(m, off, col)
        m.set_ring(1)
        for lanes in (1, 2, 4, 8):
            m.set_vector_lanes(lanes)
            assert_spmv_close(m.mvp(x, variant="vector"), off, col, val, x, "%dx%d lanes%d" % (n_rows, n_cols, lanes))
