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
    bands, intervals = m.ring_bands(intervals=True)
    assert bands in (1, 4)
    S = ring // bands  # columns per band
    nb, frac, active, ptr, ph = m.ring_plan()
    assert nb % 8 == 0 and len(ptr) == nb + 1 and ptr[0] == 0 and ptr[-1] == len(ph)
    assert np.all(np.diff(ptr.astype(np.int64)) >= 0)
    next_row = 0
    ring_rows = 0
    for b in range(nb):
        win = [[0, 0] for _ in range(4)]  # band k holds columns [lo, hi): what its slots contain after the loads so far
        for p in range(ptr[b], ptr[b + 1]):
            row = [int(v) for v in ph[p]]
            rb, re, use = row[0], row[1], row[4]
            loads = [(row[2], row[3])] + [(row[5 + k], row[8 + k]) for k in range(3)]
            assert rb == next_row and re > rb and rb % 64 == 0
            next_row = re
            for k, (llo, lhi) in enumerate(loads):
                if llo >= lhi:
                    continue
                assert k == 0 or bands == 4
                assert lhi - llo <= S
                lo, hi = win[k]
                if llo == hi and hi > lo:  # continuation of the band's window: older slots get overwritten
                    win[k] = [max(lo, lhi - S), lhi]
                else:  # restart
                    win[k] = [llo, lhi]
            if use:
                ring_rows += re - rb
                for t in range(rb // 64, (re + 63) // 64):
                    r0, r1 = t * 64, min(n_rows, t * 64 + 64)
                    cols = col[off[r0]:off[r1]]
                    if not len(cols):
                        continue
                    if bands == 1:
                        assert cols.min() >= win[0][0] and cols.max() < win[0][1], (b, p, t)
                    else:  # every column of the tile lies in one of its intervals, interval k inside band k's window
                        iv = [(int(a), int(e)) for a, e in intervals[t] if e > a]
                        assert iv and all(iv[i][1] <= iv[i + 1][0] for i in range(len(iv) - 1))
                        inside = np.zeros(len(cols), bool)
                        for k, (a, e) in enumerate(iv):
                            assert e - a <= S and a >= win[k][0] and e <= win[k][1], (b, p, t, k, (a, e), win[k])
                            inside |= (cols >= a) & (cols < e)
                        assert inside.all(), (b, p, t)
                for k in range(4):
                    assert win[k][1] - win[k][0] <= S
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


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_banded_ring_for_stencils(gpu, dtype):
    """Rows that reference a few narrow column intervals far apart (7-point stencil on a 200 x 200 x 30 grid: the
    planes are 40 000 columns apart, more than any single window) take the BANDED ring: 4 bands of a quarter of the
    ring, band k sliding over the tiles' k-th interval.  Plan invariants against the actual columns (incl. the
    boundary planes, where tiles have 2 instead of 3 intervals) and value parity for every lane width; the same
    for a 27-point stencil assembled from an element stream (unsorted rows) and after sort_rows."""
    g = (200, 200, 30)
    n = g[0] * g[1] * g[2]
    off, col, val = oracle.laplace3d(*g, dtype)
    rng = np.random.default_rng(5)
    val = (val * rng.uniform(0.5, 1.5, len(val))).astype(dtype)  # no special structure in the values
    x = rng.uniform(-1, 1, n).astype(dtype)
    m = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val)
    frac, _ = check_plan(m, off, col)
    assert m.ring_bands() == 4 and frac > 0.95
    for lanes in (2, 4, 8):
        m.set_vector_lanes(lanes)
        y = m.mvp(x, variant="vector")
        assert_spmv_close(y, off, col, val, x, "banded ring lanes %d" % lanes)
        assert np.array_equal(y, m.mvp(x, variant="vector"))
    # 27-point stencil from a hexahedral element stream: rows in first-appearance order
    ge = 60
    nodes = np.arange((ge + 1) ** 3, dtype=np.uint32).reshape(ge + 1, ge + 1, ge + 1)
    corners = np.stack([nodes[dx:ge + dx, dy:ge + dy, dz:ge + dz].ravel()
                        for dx in (0, 1) for dy in (0, 1) for dz in (0, 1)], axis=1)
    t_rows = np.repeat(corners, 8, axis=1).ravel()
    t_cols = np.tile(corners, (1, 8)).ravel()
    t_vals = rng.uniform(-1, 1, len(t_rows)).astype(dtype)
    m27 = sm.SparseMatCRS.from_triplets(t_rows, t_cols, t_vals)
    off, col, val = m27.raw_parts()
    n27 = m27.n_rows()
    x = rng.uniform(-1, 1, n27).astype(dtype)
    assert m27.resolved_variant()[0] == "vector"
    check_plan(m27, off, col)
    assert_spmv_close(m27.mvp(x), off, col, val, x, "27-point")
    m27.sort_rows()
    off, col, val = m27.raw_parts()
    assert_spmv_close(m27.mvp(x), off, col, val, x, "27-point, rows sorted")


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_banded_ring_random_clusters(gpu, dtype):
    """Adversarial input for the banded plan: every 64-row tile draws 1-6 column clusters (more than 4: not
    describable -> global-gather phase), cluster positions jump forwards and backwards from tile to tile, some
    tiles are empty, some clusters are wider than a band.  The simulated ring contents must cover every ring
    phase's columns and the product must match the oracle."""
    rng = np.random.default_rng(2024)
    n_rows, n_cols = 64 * 700 + 13, 3_000_000
    lens = rng.integers(0, 25, n_rows)
    for t in range(5, (n_rows + 63) // 64, 37):
        lens[t * 64:t * 64 + 64] = 0  # tiles without entries
    off = np.zeros(n_rows + 1, np.uint32)
    np.cumsum(lens, out=off[1:])
    col = np.empty(int(off[-1]), np.uint32)
    base = 0
    for t in range((n_rows + 63) // 64):
        r0, r1 = t * 64, min(n_rows, t * 64 + 64)
        a, b = off[r0], off[r1]
        n_clusters = int(rng.integers(1, 7))
        base = (base + int(rng.integers(-40_000, 90_000))) % (n_cols - 600_000)  # wanders both ways
        centres = base + np.sort(rng.choice(500_000, n_clusters, replace=False))
        width = 6000 if t % 53 == 7 else int(rng.integers(50, 900))  # now and then wider than a 4096-column band
        pick = rng.integers(0, n_clusters, b - a)
        col[a:b] = centres[pick] + rng.integers(0, width, b - a)
    val = rng.uniform(-1, 1, len(col)).astype(dtype)
    x = rng.uniform(-1, 1, n_cols).astype(dtype)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    frac, _ = check_plan(m, off, col)
    assert m.ring_bands() == 4 and 0.5 <= frac < 1.0  # the > 4-cluster and the wide tiles stay outside the ring
    for lanes in (4, 8):
        m.set_vector_lanes(lanes)
        y = m.mvp(x, variant="vector")
        assert_spmv_close(y, off, col, val, x, "random clusters lanes %d" % lanes)
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


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_k1_is_k1r_bit_for_bit(gpu, dtype):
    """The plain lane-group kernel (VECTOR with the ring off) and the LDS-ring kernel share lanes, chunk grid, lane layout (f64: two
    16-byte pieces per lane and pass in both) and the order of their FMAs: their results are the same bits.  The partition relies on
    it: a block's few boundary rows go through K1 (a hundred small workgroups) while the interior goes through K1r
    (spmv_enqueue_rows_short, capi.hip)."""
    rng = np.random.default_rng(5)
    u = np.uint32 if dtype == np.float32 else np.uint64
    cases = [synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_WINDOW, 400_000, 32, dtype),
             synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, 200_000, 7, dtype)]
    n_rows = 60_000
    lens = rng.integers(0, 90, n_rows)
    off = np.zeros(n_rows + 1, np.uint32)
    np.cumsum(lens, out=off[1:])
    centers = np.repeat(np.arange(n_rows), lens)
    col = np.clip(centers + rng.integers(-3000, 3000, len(centers)), 0, n_rows - 1).astype(np.uint32)
    cases.append(sm.SparseMatCRS.from_raw_parts(n_rows, n_rows, off, col, rng.uniform(-1, 1, len(col)).astype(dtype)))
    # a stencil (the banded ring: four bands) and rows of 64 entries in a +-8000 band (the wide ring)
    off, col, val = oracle.laplace3d(48, 40, 36, dtype)
    cases.append(sm.SparseMatCRS.from_raw_parts(48 * 40 * 36, 48 * 40 * 36, off, col, val))
    n_rows = 120_000
    off = (np.arange(n_rows + 1, dtype=np.uint64) * 64).astype(np.uint32)
    col = np.clip(np.repeat(np.arange(n_rows), 64) + rng.integers(-8000, 8000, n_rows * 64), 0, n_rows - 1).astype(np.uint32)
    cases.append(sm.SparseMatCRS.from_raw_parts(n_rows, n_rows, off, col, rng.uniform(-1, 1, len(col)).astype(dtype)))
    for m in cases:
        x = rng.uniform(-1, 1, m.n_cols()).astype(dtype)
        for lanes in (0, 4, 8, 16):
            m.set_vector_lanes(lanes)
            m.set_ring(1)
            assert m.ring_plan()[2]  # the ring is in use
            y1 = m.mvp(x, variant="vector")
            m.set_ring(0)
            y0 = m.mvp(x, variant="vector")
            assert np.array_equal(y1.view(u), y0.view(u)), lanes
