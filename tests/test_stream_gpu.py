"""K1s (CSR-stream for short rows): BIT-EXACT against the oracle (rounded products, storage-order fold), on
every shape incl. tiles that overflow the LDS stage, unsorted/duplicate columns and unpadded borrowed arrays."""
import numpy as np
import pytest

import oracle
import sparsemat_amd as sm
from sparsemat_amd import synth
from util import random_crs

pytestmark = pytest.mark.gpu


def bits(a):
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("kind", ["short", "len8", "empty_heavy", "overflow_tile", "ragged"])
def test_stream_bit_exact(gpu, dtype, kind):
    rng = np.random.default_rng({"short": 1, "len8": 2, "empty_heavy": 3, "overflow_tile": 4, "ragged": 5}[kind])
    n_rows, n_cols = 5003, 4001
    if kind == "short":
        lens = rng.integers(0, 10, n_rows)
    elif kind == "len8":
        lens = np.full(n_rows, 8)  # power-of-two stride in LDS: the skewed index must keep it conflict-free AND right
    elif kind == "empty_heavy":
        lens = rng.integers(0, 4, n_rows)
        lens[rng.random(n_rows) < 0.7] = 0
    elif kind == "overflow_tile":
        lens = rng.integers(0, 9, n_rows)
        lens[300:420] = 60  # tile 1 holds > 4096 entries: folded from global memory
        lens[1000] = 5000   # one very long row
    else:
        lens = rng.integers(0, 40, n_rows)
    off, col, val = random_crs(rng, n_rows, n_cols, lens, dtype, dup=True)
    x = rng.uniform(-1, 1, n_cols).astype(dtype)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    y = m.mvp(x, variant="stream")
    y_ref = oracle.spmv(off, col, val, x)
    assert np.array_equal(bits(y), bits(y_ref)), "K1s must be bit-exact (%d rows differ)" % (bits(y) != bits(y_ref)).sum()
    if kind in ("short", "len8", "empty_heavy"):
        assert m.resolved_variant()[0] == "stream"
        assert np.array_equal(bits(m.mvp(x)), bits(y_ref))  # AUTO == K1s here
    else:
        assert m.resolved_variant()[0] != "stream"


def test_stream_on_laplacians_and_borrowed_unpadded_arrays(gpu):
    for dims in [(37, 23, 1), (9, 7, 5), (64, 64, 4)]:
        off, col, val = oracle.laplace3d(*dims, np.float32)
        n = dims[0] * dims[1] * dims[2]
        x = oracle.gen_x(synth.SEED_X, n, np.float32)
        m = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val)
        assert m.resolved_variant()[0] == "stream"
        assert np.array_equal(bits(m.mvp(x)), bits(oracle.spmv(off, col, val, x)))
    # device-born Laplacian: borrowed arrays whose nnz is not a multiple of 4 (tail chunk read entry by entry)
    row_end = next(re for re in range(300, 310) if synth.laplace3d_nnz(11, 7, 5, 13, re) % 4 != 0)
    m = synth.crs_laplace3d(11, 7, 5, np.float32, 13, row_end)
    off, col, val = m.raw_parts()
    assert len(val) % 4 != 0
    x = oracle.gen_x(synth.SEED_X, 11 * 7 * 5, np.float32)
    assert np.array_equal(bits(m.mvp(x, variant="stream")), bits(oracle.spmv(off, col, val, x)))


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_stream_bit_exact_on_clustered_and_scattered_columns(gpu, dtype):
    """(Round 1-2 also carried K1s-w, a per-tile LDS window of x filled with 4-byte loads -- measured slower and removed in round 3;
    what staging x in LDS became is K1s XS, tested below.)"""
    # 7-point Laplacian: 3 planes -> 3 column intervals per tile
    nx, ny, nz = 40, 36, 9
    off, col, val = oracle.laplace3d(nx, ny, nz, dtype)
    n = nx * ny * nz
    x = oracle.gen_x(synth.SEED_X, n, dtype)
    m = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val)
    y_ref = oracle.spmv(off, col, val, x)
    assert np.array_equal(bits(m.mvp(x, variant="stream")), bits(y_ref))
    # mixed: narrow clusters, 5 clusters (too many intervals -> widest gaps win, may or may not fit), random rows
    rng = np.random.default_rng(21)
    n_rows, n_cols = 4000, 2_000_000
    lens = rng.integers(0, 9, n_rows)
    off = np.zeros(n_rows + 1, np.uint32)
    np.cumsum(lens, out=off[1:])
    col = np.empty(int(off[-1]), np.uint32)
    for i in range(n_rows):
        a, b = off[i], off[i + 1]
        t = i // 256
        if t % 4 == 0:    # one cluster
            col[a:b] = 5000 * t + rng.integers(0, 900, b - a)
        elif t % 4 == 1:  # three far clusters
            col[a:b] = rng.choice([1000, 700_000, 1_900_000], b - a) + rng.integers(0, 600, b - a)
        elif t % 4 == 2:  # six clusters: cannot be described by 4 short intervals
            col[a:b] = rng.choice([0, 300_000, 600_000, 900_000, 1_200_000, 1_500_000], b - a) + rng.integers(0, 700, b - a)
        else:             # anywhere
            col[a:b] = rng.integers(0, n_cols, b - a)
    val = rng.uniform(-1, 1, len(col)).astype(dtype)
    x = rng.uniform(-1, 1, n_cols).astype(dtype)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    y_ref = oracle.spmv(off, col, val, x)
    assert np.array_equal(bits(m.mvp(x, variant="stream")), bits(y_ref))


def test_stream_edge_shapes(gpu):
    f = np.float32
    m = sm.SparseMatCRS.from_raw_parts(5, 3, [0, 0, 0, 0, 0, 0], [], np.array([], f))
    assert np.array_equal(m.mvp(np.ones(3, f), variant="stream"), np.zeros(5, f))
    m = sm.SparseMatCRS.from_raw_parts(1, 2, [0, 1], [1], np.array([2.5], f))
    assert np.array_equal(m.mvp(np.array([1, 3, 9], f), variant="stream"), np.array([7.5], f))
    rng = np.random.default_rng(8)
    for n_rows in (255, 256, 257, 513):
        off, col, val = random_crs(rng, n_rows, 100, rng.integers(0, 7, n_rows), f)
        x = rng.uniform(-1, 1, 100).astype(f)
        m = sm.SparseMatCRS.from_raw_parts(n_rows, 100, off, col, val)
        assert np.array_equal(bits(m.mvp(x, variant="stream")), bits(oracle.spmv(off, col, val, x)))


def _with_codes(flag, fn):
    import os
    try:
        os.environ["SMH_STREAM_C16"] = flag
        return fn()
    finally:
        os.environ.pop("SMH_STREAM_C16", None)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_stream_16bit_column_codes_bit_exact(gpu, dtype):
    """When every tile's columns fall into <= 4 intervals of <= 16384 the kernel streams 16-bit codes (interval << 14
    | offset) instead of the u32 columns; the rebuilt column is the same number, so the product stays bit-exact --
    stencils (3 intervals per tile, columns far beyond 65535), bands with empty rows and dense multi-pass tiles, a
    matrix where some tiles do not qualify (u32 columns stay in use), borrowed arrays ending inside a chunk."""
    rng = np.random.default_rng(77)
    cases = []
    g = (70, 66, 40)  # 184 800 rows: the +-plane neighbours are 4 620 columns away
    off, col, val = oracle.laplace3d(*g, dtype)
    cases.append(("laplace3d", g[0] * g[1] * g[2], g[0] * g[1] * g[2], off, col, val))
    n = 120_000   # band of +-3000 around the diagonal, unsorted with duplicates, empty rows, a dense stretch
    lens = rng.integers(0, 30, n)
    lens[5000:5600] = 0
    lens[70_000:70_300] = 90
    off = np.zeros(n + 1, np.uint32)
    np.cumsum(lens, out=off[1:])
    centers = np.repeat(np.arange(n), lens)
    col = np.clip(centers + rng.integers(-3000, 3000, len(centers)), 0, n - 1).astype(np.uint32)
    cases.append(("band", n, n, off, col, rng.uniform(-1, 1, len(col)).astype(dtype)))
    n_rows, n_cols = 30_000, 5_000_000   # only some tiles qualify
    lens = rng.integers(0, 12, n_rows)
    off = np.zeros(n_rows + 1, np.uint32)
    np.cumsum(lens, out=off[1:])
    col = np.empty(int(off[-1]), np.uint32)
    for i in range(n_rows):
        a, b = off[i], off[i + 1]
        if (i // 256) % 5 == 4:
            col[a:b] = rng.integers(0, n_cols, b - a)
        else:
            col[a:b] = 150 * i + rng.choice([0, 70_000, 2_000_000], b - a) + rng.integers(0, 3000, b - a)
    col = np.minimum(col, n_cols - 1).astype(np.uint32)
    cases.append(("mixed", n_rows, n_cols, off, col, rng.uniform(-1, 1, len(col)).astype(dtype)))
    for name, n_rows, n_cols, off, col, val in cases:
        x = rng.uniform(-1, 1, n_cols).astype(dtype)
        y_ref = oracle.spmv(off, col, val, x)
        m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
        y16 = _with_codes("1", lambda: m.mvp(x, variant="stream"))
        y32 = _with_codes("0", lambda: m.mvp(x, variant="stream"))
        assert np.array_equal(bits(y16), bits(y_ref)), name
        assert np.array_equal(bits(y32), bits(y_ref)), name
        m.sort_rows()  # reorders the columns: the codes must follow
        s_col, s_val = oracle.crs_sort_rows(off, col, val)
        assert np.array_equal(bits(m.mvp(x, variant="stream")), bits(oracle.spmv(off, s_col, s_val, x))), name
    # device-born Laplacian block whose arrays are borrowed and end inside a chunk
    row_end = next(re for re in range(300, 310) if synth.laplace3d_nnz(11, 7, 5, 13, re) % 4 != 0)
    m = synth.crs_laplace3d(11, 7, 5, np.float32, 13, row_end)
    off, col, val = m.raw_parts()
    x = oracle.gen_x(synth.SEED_X, 11 * 7 * 5, np.float32)
    assert np.array_equal(bits(_with_codes("1", lambda: m.mvp(x, variant="stream"))), bits(oracle.spmv(off, col, val, x)))
    # ... and through K1s XD: its buffer descriptors end with the tile's entries, so nothing past the borrowed arrays' end is read
    m.set_stream_xs(1)
    m.set_stream_direct(1)
    assert m.stream_direct()
    assert np.array_equal(bits(m.mvp(x, variant="stream")), bits(oracle.spmv(off, col, val, x)))


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_stream_x_staged_in_lds_bit_exact(gpu, dtype):
    """K1s XS: for coded single-pass tiles with byte row lengths whose column intervals span <= 2048 (or 4096) entries of x, the
    intervals are copied to LDS with 16-byte loads and the gathers become LDS reads.  AUTO keeps that for x of 8 MB and more, so
    the small cases force it (set_stream_xs(1)); same bits as the oracle, as the kernel without the stage, and through the dot
    epilogue; x that is too short for the last aligned chunk or not 16-byte aligned falls back to gathers from memory."""
    rng = np.random.default_rng(99)
    cases = []
    for g in ((40, 40, 40), (300, 300, 1), (1000, 30, 3), (7, 5, 3), (257, 1, 1)):   # 3-D, 2-D, wide planes (4096-entry stage), tiny, 1-D
        off, col, val = oracle.laplace3d(*g, dtype)
        cases.append(("laplace %dx%dx%d" % g, g[0] * g[1] * g[2], off, col, val))
    n = 50_001   # a narrow band stored in storage order with duplicates and empty rows
    lens = rng.integers(0, 8, n)
    lens[1000:1300] = 0
    off = np.zeros(n + 1, np.uint32); np.cumsum(lens, out=off[1:])
    centers = np.repeat(np.arange(n), lens)
    col = np.clip(centers + rng.integers(-300, 300, len(centers)), 0, n - 1).astype(np.uint32)
    cases.append(("band", n, off, col, rng.uniform(-1, 1, len(col)).astype(dtype)))
    seen = set()
    for name, n, off, col, val in cases:
        x = rng.uniform(-1, 1, n).astype(dtype)
        y_ref = oracle.spmv(off, col, val, x)
        m = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val)
        assert m.stream_layout()["xs_chunks"] == 0, name          # AUTO: x is below 1 MB here
        y_plain = m.mvp(x, variant="stream")
        m.set_stream_xs(1)
        lay = m.stream_layout()
        assert lay["coded"] and lay["byte_lengths"] and lay["small_tiles"] and lay["xs_chunks"] in (2, 4), (name, lay)
        seen.add(lay["xs_chunks"])
        y_xs = m.mvp(x, variant="stream")
        assert np.array_equal(bits(y_xs), bits(y_ref)) and np.array_equal(bits(y_plain), bits(y_ref)), name
        # lhs^T A x through the epilogue of the staged kernel = through the plain one (same partial sums, same fold)
        lhs = rng.uniform(-1, 1, n).astype(dtype)
        ip_xs = m.inner_prod(lhs, x, variant="stream")
        m.set_stream_xs(0)
        assert m.stream_layout()["xs_chunks"] == 0 and m.inner_prod(lhs, x, variant="stream") == ip_xs, name
        m.set_stream_xs(1)
        # device vectors: x one entry longer than the matrix needs (aligned end inside), and a pointer that is not 16-byte aligned
        xbuf = synth.DeviceBuffer((n + 9) * x.itemsize)
        xbuf.upload(np.concatenate([np.zeros(1, dtype), x, np.zeros(8, dtype)]))
        ybuf = synth.DeviceBuffer(n * x.itemsize)
        m.mvp_dev(xbuf.ptr + x.itemsize, n, ybuf.ptr, "stream")   # misaligned x: the fallback
        sm.lib().smh_device_synchronize()
        assert np.array_equal(bits(ybuf.download(dtype, n)), bits(y_ref)), name
    assert seen == {2, 4}   # both stage sizes ran
    # a matrix whose tiles' intervals do not fit any stage keeps the gathers from memory
    # (a tile's columns are described as <= 4 intervals cut at their widest gaps: five clusters 5000 apart leave one interval of
    # 5256 columns, 6024 in total -- beyond the 4096-entry stage; a 3000 x 40 grid, three clusters of 256, fits since round 3)
    n = 120_000
    off = np.arange(n + 1, dtype=np.uint32) * 5
    col = np.clip(np.arange(n)[:, None] + np.array([-10_000, -5000, 0, 5000, 10_000]), 0, n - 1).astype(np.uint32).reshape(-1)
    val = rng.uniform(-1, 1, 5 * n).astype(dtype)
    m = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val)
    m.set_stream_xs(1)
    assert m.stream_layout()["coded"] and m.stream_layout()["xs_chunks"] == 0 and not m.stream_direct()
    x = rng.uniform(-1, 1, n).astype(dtype)
    assert np.array_equal(bits(m.mvp(x, variant="stream")), bits(oracle.spmv(off, col, val, x)))
    off, col, val = oracle.laplace3d(3000, 40, 1, dtype)   # rows 3000 apart: three clusters of 256 columns
    m = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val)
    m.set_stream_xs(1)
    assert m.stream_layout()["xs_chunks"] == 2
    assert np.array_equal(bits(m.mvp(x, variant="stream")), bits(oracle.spmv(off, col, val, x)))


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_stream_xd_stage_offsets_bit_exact(gpu, dtype):
    """K1s XD (spmv_stream_xd.hip): the XS kernel with the code array holding byte offsets into the tile's LDS stage of x, an unskewed
    product stage written 16 bytes at a time, buffer loads, a row's first eight products under a lane mask.  Automatic where x is
    staged and most rows have an odd length; same bits as the oracle, as the classic XS body and as the plain kernel -- for the
    product, for lhs . (A x) through the dot epilogue and for an x that cannot be staged (launches by runs of rows:
    tests/test_par_overlap_gpu.py)."""
    rng = np.random.default_rng(2024)
    cases = []
    for g in ((40, 40, 40), (300, 300, 1), (1000, 30, 3), (7, 5, 3), (257, 1, 1), (48, 20, 9)):
        off, col, val = oracle.laplace3d(*g, dtype)
        cases.append(("laplace %dx%dx%d" % g, g[0] * g[1] * g[2], off, col, val, bool(2 * int((np.diff(off) & 1).sum()) >= len(off) - 1)))
    n = 20_000   # rows of 5 entries with one of 23 / 40 now and then: the loop that follows the eight masked adds
    lens = np.full(n, 5)
    lens[::64], lens[37::640] = 23, 40
    off = np.zeros(n + 1, np.uint32); np.cumsum(lens, out=off[1:])
    col = np.clip(np.repeat(np.arange(n), lens) + rng.integers(-250, 250, int(off[-1])), 0, n - 1).astype(np.uint32)
    cases.append(("long rows among short ones", n, off, col, rng.uniform(-1, 1, len(col)).astype(dtype), True))
    n = 30_000   # rows of EVEN length only (4 and 6), a band: automatic keeps the skewed stage, forced runs (conflicts, same bits)
    lens = rng.choice([4, 6], n)
    off = np.zeros(n + 1, np.uint32); np.cumsum(lens, out=off[1:])
    col = np.clip(np.repeat(np.arange(n), lens) + rng.integers(-200, 200, int(off[-1])), 0, n - 1).astype(np.uint32)
    cases.append(("even rows", n, off, col, rng.uniform(-1, 1, len(col)).astype(dtype), False))
    n = 50_001   # storage order with duplicates, empty rows, a ragged last tile
    lens = rng.integers(0, 8, n)
    lens[1000:1300] = 0
    off = np.zeros(n + 1, np.uint32); np.cumsum(lens, out=off[1:])
    col = np.clip(np.repeat(np.arange(n), lens) + rng.integers(-300, 300, int(off[-1])), 0, n - 1).astype(np.uint32)
    cases.append(("band", n, off, col, rng.uniform(-1, 1, len(col)).astype(dtype), None))
    for name, n, off, col, val, auto_direct in cases:
        x = rng.uniform(-1, 1, n).astype(dtype)
        lhs = rng.uniform(-1, 1, n).astype(dtype)
        y_ref = oracle.spmv(off, col, val, x)
        m = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val)
        assert not m.stream_direct(), name                        # x is not staged by default at these sizes
        y_plain, ip_plain = m.mvp(x, variant="stream"), m.inner_prod(lhs, x, variant="stream")
        m.set_stream_xs(1)
        assert m.stream_layout()["xs_chunks"] in (2, 4), name
        if auto_direct is not None:
            assert m.stream_direct() == auto_direct, name
        m.set_stream_direct(0)
        assert not m.stream_direct(), name
        y_xs, ip_xs = m.mvp(x, variant="stream"), m.inner_prod(lhs, x, variant="stream")
        m.set_stream_direct(1)
        assert m.stream_direct(), name
        y_xd, ip_xd = m.mvp(x, variant="stream"), m.inner_prod(lhs, x, variant="stream")
        for y in (y_plain, y_xs, y_xd):
            assert np.array_equal(bits(y), bits(y_ref)), name
        assert ip_plain == ip_xs == ip_xd, name
        # a pointer that is not 16-byte aligned: x cannot be staged, the stage offsets mean nothing -> the u32 columns, same bits
        xbuf = synth.DeviceBuffer((n + 9) * x.itemsize)
        xbuf.upload(np.concatenate([np.zeros(1, dtype), x, np.zeros(8, dtype)]))
        ybuf = synth.DeviceBuffer(n * x.itemsize)
        m.mvp_dev(xbuf.ptr + x.itemsize, n, ybuf.ptr, "stream")
        sm.lib().smh_device_synchronize()
        assert np.array_equal(bits(ybuf.download(dtype, n)), bits(y_ref)), name
        # back and forth once more: the code array is rewritten in place
        m.set_stream_direct(0)
        assert np.array_equal(bits(m.mvp(x, variant="stream")), bits(y_ref)), name
        m.set_stream_direct(-1)
        assert np.array_equal(bits(m.mvp(x, variant="stream")), bits(y_ref)), name


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_stream_value_dictionary_bit_exact(gpu, dtype):
    """K1s XD-V: with at most 32 distinct values (16 with the 4096-entry stage) the spare bits of the stage-offset codes name each entry's
    value in a dictionary and the kernel does not read the value array.  The dictionary is exact (bit patterns, -0.0 and +0.0 apart),
    sorted by pattern, and vanishes the moment the values do not allow it; update_values looks again, scale scales it; the product, the
    dot epilogue and the solver stay bit for bit what they were."""
    rng = np.random.default_rng(77)
    u = np.uint32 if dtype == np.float32 else np.uint64

    def expect(vals):  # distinct bit patterns, sorted as unsigned integers
        return np.sort(np.unique(np.asarray(vals, dtype).view(u))).view(dtype)

    def check(m, off, col, val, n, name, want_n):
        x, lhs = rng.uniform(-1, 1, n).astype(dtype), rng.uniform(-1, 1, n).astype(dtype)
        d = m.stream_value_dict()
        assert len(d) == want_n, (name, len(d), want_n)
        if want_n:  # (sorted by bit pattern when it is built; smh_crs_scale scales it in place, which may reorder the patterns)
            assert np.array_equal(np.sort(bits(d)), bits(expect(val))), name
        assert np.array_equal(bits(m.mvp(x, variant="stream")), bits(oracle.spmv(off, col, val, x))), name
        m.set_stream_value_dict(0)   # the same kernel reading the value array: same bits, same dot
        assert len(m.stream_value_dict()) == 0
        ip0 = m.inner_prod(lhs, x, variant="stream")
        assert np.array_equal(bits(m.mvp(x, variant="stream")), bits(oracle.spmv(off, col, val, x))), name
        m.set_stream_value_dict(-1)
        assert len(m.stream_value_dict()) == want_n and m.inner_prod(lhs, x, variant="stream") == ip0, name

    # the stencils: two values; the 2048-entry stage and (wide planes) the 4096-entry one
    for g in ((40, 40, 40), (1000, 30, 3)):
        off, col, val = oracle.laplace3d(*g, dtype)
        n = g[0] * g[1] * g[2]
        m = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val)
        m.set_stream_xs(1)
        m.set_stream_direct(1)   # (automatic only where most rows are of odd length: not so on the three-plane grid)
        assert m.stream_direct() and (m.stream_layout()["xs_chunks"] == 4) == (g[0] == 1000)
        check(m, off, col, val, n, "laplace %s" % (g,), 2)
        # scale: the dictionary is scaled with the values
        m.scale(-0.375)
        check(m, off, col, (val * dtype(-0.375)).astype(dtype), n, "scaled", 2)
        # new values, arbitrary: the form goes away; back to a few: it returns
        v2 = rng.uniform(-1, 1, len(val)).astype(dtype)
        m.update_values(v2)
        check(m, off, col, v2, n, "arbitrary values", 0)
        few = np.array([0.0, -0.0, 1.5, -2.25, 3.0], dtype)
        v3 = few[rng.integers(0, 5, len(val))]
        m.update_values(v3)
        check(m, off, col, v3, n, "five values, both zeros", 5)
        # the solver on top (fused p.Ap epilogue) against the oracle
        m.update_values(val)   # other values, the same NUMBER of them as before the last update is not the same dictionary
        check(m, off, col, val, n, "the first values again", 2)
        m.update_values(v3)
        m.update_values((v3 * dtype(2)).astype(dtype))
        check(m, off, col, (v3 * dtype(2)).astype(dtype), n, "five values doubled", 5)
        if g[0] == 40:
            m.update_values(val)
            b = oracle.spmv(off, col, val, np.ones(n, dtype))
            tol = 2e-2 if dtype == np.float32 else 1e-9  # (f32: well above what the type attains on this matrix, cf. bench.py's C4 gate)
            xs = np.zeros(n, dtype)
            cg = sm.ConjugateGradient(tol, 500)
            cg.solve(m, b, xs)
            x_ref, it_ref, _ = oracle.cg(n, n, off, col, val, b, np.zeros(n, dtype), tol=tol, iter_max=500)
            assert len(m.stream_value_dict()) == 2 and abs(cg.iterations - it_ref) <= 1 and np.abs(xs - x_ref).max() <= 10 * tol, (cg.iterations, it_ref)
    # capacity: 32 distinct values with the 2048-entry stage (33: no), 16 with the 4096-entry one (17: no)
    for g, cap in (((40, 40, 40), 32), ((1000, 30, 3), 16)):
        off, col, _ = oracle.laplace3d(*g, dtype)
        n = g[0] * g[1] * g[2]
        for k in (cap, cap + 1):
            pool = (np.arange(k) * 0.37 - 3.0).astype(dtype)
            val = pool[rng.integers(0, k, len(col))]
            val[:k] = pool   # every value occurs
            m = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val)
            m.set_stream_xs(1)
            m.set_stream_direct(1)
            check(m, off, col, val, n, "%d values" % k, k if k <= cap else 0)
    # without stage offsets (x not staged) there are no spare bits: nothing changes
    off, col, val = oracle.laplace3d(12, 12, 12, dtype)
    m = sm.SparseMatCRS.from_raw_parts(1728, 1728, off, col, val)
    assert not m.stream_direct() and len(m.stream_value_dict()) == 0


@pytest.mark.parametrize("persist", ["0", "1"])
def test_stream_value_dictionary_both_launch_forms(gpu, persist):
    """K1s XD-V has two launch forms -- one workgroup per tile, and persistent workgroups that fetch the next tile while they multiply
    (k_spmv_stream_xdp; the default for f64 only) -- chosen once per process by SMH_STREAM_PERSIST.  Both must give the oracle's bits
    in both types: the dictionary test above and BASELINE C1's product (3907 tiles: several per persistent workgroup, a ragged last
    one) run again in a fresh process under each setting."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.join(here, "test_stream_gpu.py"),
                        os.path.join(here, "test_c1_gpu.py"), "-k", "test_stream_value_dictionary_bit_exact or test_stream_value_dictionary_rows_beyond_eight_entries or test_c1_product"],
                       env=dict(os.environ, SMH_STREAM_PERSIST=persist), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_stream_value_dictionary_rows_beyond_eight_entries(gpu, dtype):
    """K1s XD folds a row's first eight products from registers and the rest in a loop: a 5-point Laplacian in which every 97th row
    carries 6 or 8 more entries (11 / 13 in all; duplicates of its own columns, values from the matrix's small set), so that the loop
    runs in the dictionary kernels of both launch forms (f32: a workgroup per tile, f64: persistent workgroups) -- bit for bit the
    oracle, with and without the dictionary, the dot epilogue too."""
    nx, ny = 1000, 60
    off, col, val = oracle.laplace2d(nx, ny, dtype)
    n = nx * ny
    rng = np.random.default_rng(97)
    lens = np.diff(off).astype(np.int64)
    extra = np.zeros(n, np.int64)
    extra[::97] = np.where(np.arange(0, n, 97) % 2 == 0, 6, 8)
    new_off = np.zeros(n + 1, np.uint32)
    np.cumsum(lens + extra, out=new_off[1:])
    new_col = np.empty(int(new_off[-1]), np.uint32)
    new_val = np.empty(int(new_off[-1]), dtype)
    pool = np.array([4.0, -1.0, 0.5], dtype)
    for i in range(n):
        a, b = int(off[i]), int(off[i + 1])
        o = int(new_off[i])
        new_col[o:o + b - a] = col[a:b]
        new_val[o:o + b - a] = val[a:b]
        e = int(extra[i])
        if e:
            new_col[o + b - a:o + b - a + e] = col[a:b][rng.integers(0, b - a, e)]
            new_val[o + b - a:o + b - a + e] = pool[rng.integers(0, 3, e)]
    m = sm.SparseMatCRS.from_raw_parts(n, n, new_off, new_col, new_val)
    m.set_stream_xs(1)
    m.set_stream_direct(1)
    assert m.stream_direct() and len(m.stream_value_dict()) == 3 and m.max_row_len() == 13
    x, lhs = rng.uniform(-1, 1, n).astype(dtype), rng.uniform(-1, 1, n).astype(dtype)
    want = oracle.spmv(new_off, new_col, new_val, x)
    assert np.array_equal(bits(m.mvp(x, variant="stream")), bits(want))
    ip = m.inner_prod(lhs, x, variant="stream")
    m.set_stream_value_dict(0)
    assert np.array_equal(bits(m.mvp(x, variant="stream")), bits(want)) and m.inner_prod(lhs, x, variant="stream") == ip
