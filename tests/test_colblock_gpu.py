"""K2c (column-blocked CSR for columns without locality): the split is integer work -> bit-exact against a numpy
stable partition; the product is within the parity tolerance of the storage-order oracle AND bit-exact against
the oracle applied block by block (each block is a K1s launch: rounded products, storage-order fold, one rounded
add into y per block)."""
import numpy as np
import pytest

import oracle
import sparsemat_amd as sm
from sparsemat_amd import synth
from util import assert_spmv_close, random_crs

pytestmark = pytest.mark.gpu


def bits(a):
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


def split_reference(off, col, val, n_cols, shift):
    """Stable partition of the entries by (column block, row): offsets[B, n_rows+1] absolute, columns, values."""
    n_rows = len(off) - 1
    n_blocks = max(1, (n_cols + (1 << shift) - 1) >> shift)
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(off.astype(np.int64)))
    blk = (col >> np.uint32(shift)).astype(np.int64)
    order = np.argsort(blk * n_rows + rows, kind="stable")
    cnt = np.zeros((n_blocks, n_rows + 1), dtype=np.int64)
    np.add.at(cnt, (blk, rows), 1)
    flat = cnt.ravel()
    offs = np.concatenate(([0], np.cumsum(flat)[:-1])).astype(np.uint32).reshape(n_blocks, n_rows + 1)
    return n_blocks, offs, col[order], val[order]


def blockwise_reference(offs, col2, val2, x, n_rows):
    """y = A_0 x; y = y + A_b x ... with the oracle's storage-order fold per block."""
    y = None
    for b in range(offs.shape[0]):
        lo = int(offs[b, 0])
        o = (offs[b].astype(np.int64) - lo).astype(np.uint32)
        hi = int(offs[b, -1])
        part = oracle.spmv(o, col2[lo:hi], val2[lo:hi], x)
        y = part if y is None else (y + part).astype(part.dtype)
    return y


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("kind", ["uniform32", "ragged", "skewed", "empty_heavy"])
def test_colblock_split_and_product(gpu, dtype, kind):
    rng = np.random.default_rng({"uniform32": 11, "ragged": 12, "skewed": 13, "empty_heavy": 14}[kind])
    n_rows, n_cols, shift = 6007, 5001, 8  # 20 column blocks of 256 columns
    if kind == "uniform32":
        lens = np.full(n_rows, 32)
    elif kind == "ragged":
        lens = rng.integers(0, 70, n_rows)
    elif kind == "skewed":
        lens = rng.integers(0, 6, n_rows)
        lens[17] = 9000      # far more than a tile's LDS stage, in every block
        lens[4000:4100] = 300
    else:
        lens = rng.integers(0, 5, n_rows)
        lens[rng.random(n_rows) < 0.8] = 0
    off, col, val = random_crs(rng, n_rows, n_cols, lens, dtype, dup=True)
    x = rng.uniform(-1, 1, n_cols).astype(dtype)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    m.set_colblock_shift(shift)
    cb = m.colblock()
    n_blocks, offs, col2, val2 = split_reference(off, col, val, n_cols, shift)
    assert cb["shift"] == shift and cb["n_blocks"] == n_blocks and cb["rows_per_thread"] in (1, 2, 4, 8)
    assert np.array_equal(cb["offsets"], offs)
    assert np.array_equal(cb["columns"], col2)
    assert np.array_equal(bits(cb["values"]), bits(val2))
    y = m.mvp(x, variant="colblock")
    assert_spmv_close(y, off, col, val, x, "colblock " + kind)
    assert np.array_equal(bits(y), bits(blockwise_reference(offs, col2, val2, x, n_rows)))
    # run-to-run bitwise reproducible
    assert np.array_equal(bits(y), bits(m.mvp(x, variant="colblock")))


def test_colblock_single_block_is_stream_bit_exact(gpu):
    rng = np.random.default_rng(15)
    n_rows, n_cols = 3001, 900
    off, col, val = random_crs(rng, n_rows, n_cols, rng.integers(0, 12, n_rows), np.float32)
    x = rng.uniform(-1, 1, n_cols).astype(np.float32)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    cb = m.colblock(arrays=False)
    assert cb["n_blocks"] == 1 and cb["shift"] == 19
    assert np.array_equal(bits(m.mvp(x, variant="colblock")), bits(oracle.spmv(off, col, val, x)))
    assert m.resolved_variant()[0] != "colblock"  # x is tiny: AUTO never picks K2c


def test_colblock_follows_value_updates_and_edge_shapes(gpu):
    f = np.float64
    rng = np.random.default_rng(16)
    n_rows, n_cols = 1500, 2000
    off, col, val = random_crs(rng, n_rows, n_cols, rng.integers(0, 20, n_rows), f)
    x = rng.uniform(-1, 1, n_cols).astype(f)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    m.set_colblock_shift(7)
    assert_spmv_close(m.mvp(x, variant="colblock"), off, col, val, x, "before")
    m.scale(-0.5)
    assert_spmv_close(m.mvp(x, variant="colblock"), off, col, (val * -0.5), x, "scaled")
    val2 = rng.uniform(-1, 1, len(val)).astype(f)
    m.update_values(val2)
    assert_spmv_close(m.mvp(x, variant="colblock"), off, col, val2, x, "updated")
    # no entries at all / a single entry
    m = sm.SparseMatCRS.from_raw_parts(5, 300, [0, 0, 0, 0, 0, 0], [], np.array([], f))
    m.set_colblock_shift(6)
    assert np.array_equal(m.mvp(np.ones(300, f), variant="colblock"), np.zeros(5, f))
    m = sm.SparseMatCRS.from_raw_parts(2, 300, [0, 0, 1], [299], np.array([2.0], f))
    m.set_colblock_shift(6)
    assert np.array_equal(m.mvp(np.arange(300, dtype=f), variant="colblock"), np.array([0.0, 598.0], f))


def test_auto_picks_colblock_only_without_column_locality(gpu):
    """x of 12 MB (f32): uniform columns -> column-blocked (K2f); banded columns -> K1r as before.  Parity on both."""
    rows, n, k = 200_000, 3_000_000, 16
    m_u = synth.crs_fixed(synth.SEED_MATRIX, 1, n, k, np.float32, 0, rows)
    assert m_u.resolved_variant()[0] == "tiled"  # f32, ~48 entries per (slice, row block) tile: the 2-D tiled passes (K2t) ...
    y_f = m_u.mvp(oracle.gen_x(synth.SEED_X, n, np.float32), variant="colfused")  # ... the one-sweep blocking (K2f) otherwise
    cb = m_u.colblock(arrays=False)
    assert cb["n_blocks"] == 6 and cb["span_fraction"] > 0.9 and m_u.colfused(arrays=False)["n_blocks"] == 12
    off, col, val = m_u.raw_parts()
    x = oracle.gen_x(synth.SEED_X, n, np.float32)
    y = m_u.mvp(x)
    assert_spmv_close(y, off, col, val, x, "auto tiled")
    assert_spmv_close(y_f, off, col, val, x, "colfused")
    for pattern in (0, 2):  # banded: K1r as before
        m_b = synth.crs_fixed(synth.SEED_MATRIX, pattern, n, k, np.float32, 1_000_000, 1_000_000 + rows)
        assert m_b.resolved_variant()[0] == "vector", pattern
        assert m_b.colblock(arrays=False)["span_fraction"] < 0.05


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
def test_cg_on_a_matrix_without_locality_runs_colblock(gpu, dtype):
    """A randomly permuted 7-point Laplacian (SPD, 1.33 M rows: 10.6 MB of f64 x, columns all over it) resolves to
    the 2-D tiled passes, K2t (before it: K2f); the device-resident CG (hipGraph replay of those launches)
    follows the oracle's iterates (f32: within the drift of 40 unconverged iterations whose sums are ordered differently)."""
    g = 110
    n = g ** 3
    off, col, val = oracle.laplace3d(g, g, g, dtype)
    perm = np.random.default_rng(4).permutation(n).astype(np.uint32)   # new index of old row/column
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    lens = np.diff(off.astype(np.int64))
    lens_p = lens[inv]                                                   # row r of P A P^T is old row inv[r]
    off_p = np.zeros(n + 1, np.uint32)
    np.cumsum(lens_p, out=off_p[1:])
    src = (off[:-1].astype(np.int64)[inv])
    gather = np.repeat(src - off_p[:-1].astype(np.int64), lens_p) + np.arange(int(off_p[-1]))
    col_p, val_p = perm[col[gather]], val[gather]
    m = sm.SparseMatCRS.from_raw_parts(n, n, off_p, col_p, val_p)
    assert m.resolved_variant()[0] == "tiled" and m.tiled_layout()["n_slices"] == 82  # before K2t: K2f ...
    if dtype == np.float64:
        assert m.colfused(arrays=False)["n_blocks"] == 6  # ... over 2^18 columns of f64 each
    b = oracle.spmv(off_p, col_p, val_p, np.ones(n, dtype))
    iters = 40
    x_ref, it_ref, rr_ref = oracle.cg(n, n, off_p, col_p, val_p, b, np.zeros(n, dtype), tol=1e-30, iter_max=iters)
    xd, bd = sm.DenseVec.from_vec(np.zeros(n, dtype)), sm.DenseVec.from_vec(b)
    cg = sm.ConjugateGradient(1e-30, iters)
    cg.solve(m, bd, xd)
    assert cg.iterations == it_ref == iters
    if dtype == np.float64:
        assert abs(cg.r_norm_squared - rr_ref) <= 1e-9 * rr_ref
        np.testing.assert_allclose(xd.to_numpy(), x_ref, rtol=0, atol=1e-10)
    else:
        # the oracle's sequential f32 dots over 1.3 M elements drift by themselves (9 % in r.r here): the device solver with the
        # bit-exact K1s product and the same tree-shaped dots is the closer yardstick for what changing the product's order does
        xs = sm.DenseVec.from_vec(np.zeros(n, dtype))
        cg_s = sm.ConjugateGradient(1e-30, iters, variant="stream")
        cg_s.solve(m, bd, xs)
        assert cg_s.iterations == iters and abs(cg.r_norm_squared - cg_s.r_norm_squared) <= 0.02 * cg_s.r_norm_squared
        np.testing.assert_allclose(xd.to_numpy(), xs.to_numpy(), rtol=0, atol=2e-3 * np.abs(x_ref).max())
        np.testing.assert_allclose(xd.to_numpy(), x_ref, rtol=0, atol=5e-2 * np.abs(x_ref).max())


# ---- K2f: the same blocking in one sweep over y (spmv_colfused.hip) ---------------------------------------------------
def fused_tiles(off, rt):
    """The greedy tiling: in row order, a tile takes 64 * h rows with the largest h <= rt that keeps it within 1.25x the
    entries of a mean full-height tile (h = 1 if even 64 rows exceed that); the last tile takes what is left."""
    n_rows, nnz = len(off) - 1, int(off[-1])
    target = int(1.25 * (nnz * 64.0 * rt / n_rows)) + 1 if n_rows else 0
    o = off.astype(np.int64)
    tiles, r = [], 0
    while r < n_rows:
        tiles.append(r)
        h = 1
        for cand in range(rt, 0, -1):
            if o[min(n_rows, r + 64 * cand)] - o[r] <= target:
                h = cand
                break
        r = min(n_rows, r + 64 * h)
    return np.array(tiles + [n_rows], np.int64)


def fused_reference(off, col, val, n_cols, shift, rt):
    """The K2f copy restated in numpy: entries sorted by (tile, column block, row, storage order); one u32 per
    (tile, block) -- where its entries start -- and one byte per (row, block)."""
    n_rows = len(off) - 1
    lens = np.diff(off.astype(np.int64))
    n_blocks = max(1, -(-n_cols // (1 << shift)))
    tile_rows = 64 * rt
    tiles = fused_tiles(off, rt)
    n_tiles = len(tiles) - 1
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), lens)
    blk = (col >> np.uint32(shift)).astype(np.int64)
    tile_of_row = np.searchsorted(tiles, np.arange(n_rows), side="right") - 1
    h_of_tile = (np.diff(tiles) + 63) // 64
    tile = tile_of_row[rows]
    order = np.lexsort((rows, blk, tile))  # stable: ties keep the storage order
    cnt = np.zeros(n_tiles * n_blocks * tile_rows, np.int64)
    inside = rows - tiles[tile]
    h = h_of_tile[tile]
    np.add.at(cnt, ((tile * n_blocks + blk) * 64 + inside // h) * rt + inside % h, 1)
    per_seg = cnt.reshape(n_tiles * n_blocks, tile_rows).sum(axis=1) if n_tiles else np.zeros(0, np.int64)
    seg = np.concatenate(([0], np.cumsum(per_seg))).astype(np.uint32)
    return n_blocks, tiles.astype(np.uint32), seg, cnt, col[order], val[order]


def block_ordered_reference(off, col, val, x, shift):
    """A row's entries taken column block by column block (storage order inside a block), folded sequentially like the
    reference folds a row: the order in which K2f adds."""
    lens = np.diff(off.astype(np.int64))
    rows = np.repeat(np.arange(len(off) - 1, dtype=np.int64), lens)
    order = np.lexsort(((col >> np.uint32(shift)).astype(np.int64), rows))
    return oracle.spmv(off, col[order], val[order], x)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("kind", ["uniform32", "ragged", "long_rows", "empty_heavy", "one_tile"])
def test_colfused_split_and_product(gpu, dtype, kind):
    rng = np.random.default_rng({"uniform32": 21, "ragged": 22, "long_rows": 23, "empty_heavy": 24, "one_tile": 25}[kind])
    n_rows, n_cols, shift = 6007, 5001, 8  # 20 column blocks of 256 columns; 6 tiles of 1024 rows, the last one ragged
    if kind == "uniform32":
        lens = np.full(n_rows, 32)
    elif kind == "ragged":
        lens = rng.integers(0, 70, n_rows)
    elif kind == "long_rows":
        lens = rng.integers(0, 6, n_rows)
        lens[17] = 3000      # 150 per block: several passes of its wave's LDS stage in every block
        lens[4000:4100] = 300
    elif kind == "one_tile":
        n_rows = 700
        lens = rng.integers(0, 40, n_rows)
    else:
        lens = rng.integers(0, 5, n_rows)
        lens[rng.random(n_rows) < 0.8] = 0
    off, col, val = random_crs(rng, n_rows, n_cols, lens, dtype, dup=True)
    x = rng.uniform(-1, 1, n_cols).astype(dtype)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    m.set_colblock_shift(shift)
    cf = m.colfused()
    assert cf["fits"] and cf["shift"] == shift and cf["rows_per_lane"] == 16
    n_blocks, tiles, seg, cnt, col2, val2 = fused_reference(off, col, val, n_cols, shift, cf["rows_per_lane"])
    assert (cf["n_blocks"], cf["n_tiles"]) == (n_blocks, len(tiles) - 1) and np.array_equal(cf["tile_rows"], tiles)
    if kind == "long_rows":
        assert np.diff(tiles.astype(np.int64)).min() < 64 * cf["rows_per_lane"]  # the stretch of long rows got shorter tiles
    assert np.array_equal(cf["segments"], seg) and np.array_equal(cf["counts"], cnt.astype(np.uint8))
    assert np.array_equal(cf["columns"], col2) and np.array_equal(bits(cf["values"]), bits(val2))
    y = m.mvp(x, variant="colfused")
    assert_spmv_close(y, off, col, val, x, "colfused " + kind)
    assert np.array_equal(bits(y), bits(block_ordered_reference(off, col, val, x, shift)))  # bit-exact in ITS order of additions
    assert np.array_equal(bits(y), bits(m.mvp(x, variant="colfused")))                      # run-to-run reproducible
    m.scale(-0.5)   # the copy follows value changes
    assert np.array_equal(bits(m.mvp(x, variant="colfused")), bits(block_ordered_reference(off, col, (val * dtype(-0.5)).astype(dtype), x, shift)))
    val3 = rng.uniform(-1, 1, len(val)).astype(dtype)
    m.update_values(val3)
    assert np.array_equal(bits(m.mvp(x, variant="colfused")), bits(block_ordered_reference(off, col, val3, x, shift)))


def test_colfused_falls_back_when_a_row_block_pair_overflows_the_byte_table(gpu):
    """More than 255 entries of one row in one column block: no K2f copy; the variant then runs K2c, bit for bit."""
    rng = np.random.default_rng(31)
    n_rows, n_cols, shift = 3000, 2000, 8
    lens = rng.integers(0, 10, n_rows)
    lens[5] = 4000  # 500 per block
    off, col, val = random_crs(rng, n_rows, n_cols, lens, np.float32)
    x = rng.uniform(-1, 1, n_cols).astype(np.float32)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    m.set_colblock_shift(shift)
    assert not m.colfused(arrays=False)["fits"]
    y = m.mvp(x, variant="colfused")
    assert np.array_equal(bits(y), bits(m.mvp(x, variant="colblock")))
    assert_spmv_close(y, off, col, val, x, "colfused -> colblock")


def test_colfused_edge_shapes(gpu):
    f = np.float64
    m = sm.SparseMatCRS.from_raw_parts(5, 300, np.zeros(6, np.uint32), np.zeros(0, np.uint32), np.zeros(0, f))
    m.set_colblock_shift(6)
    assert np.array_equal(m.mvp(np.ones(300, f), variant="colfused"), np.zeros(5, f))
    m = sm.SparseMatCRS.from_raw_parts(2, 300, [0, 0, 2], [299, 299], np.array([1.0, 1.0], f))
    m.set_colblock_shift(6)
    assert np.array_equal(m.mvp(np.arange(300, dtype=f), variant="colfused"), np.array([0.0, 598.0], f))
    # one block only: the order of additions is the storage order -> bit-exact against the reference loop itself
    rng = np.random.default_rng(2)
    off, col, val = random_crs(rng, 2500, 900, rng.integers(0, 50, 2500), np.float32)
    x = rng.uniform(-1, 1, 900).astype(np.float32)
    m = sm.SparseMatCRS.from_raw_parts(2500, 900, off, col, val)
    assert m.colfused(arrays=False)["n_blocks"] == 1
    assert np.array_equal(bits(m.mvp(x, variant="colfused")), bits(oracle.spmv(off, col, val, x)))


# ---- K2s: a skewed matrix as a long-row and a short-row matrix (spmv_colsplit.hip) -----------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_colsplit_structure_and_product(gpu, dtype):
    rng = np.random.default_rng(41)
    n_rows, n_cols = 9001, 7003
    lens = rng.integers(0, 12, n_rows)
    heavy = rng.random(n_rows) < 0.04           # a minority of long rows holding most of the entries
    lens[heavy] = rng.integers(64, 700, heavy.sum())
    lens[11] = 63                                # just below the threshold: stays in the short part
    lens[12] = 64                                # exactly the threshold: long
    off, col, val = random_crs(rng, n_rows, n_cols, lens, dtype, dup=True)
    x = rng.uniform(-1, 1, n_cols).astype(dtype)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    m.set_colblock_shift(9)
    cs = m.colsplit()
    assert cs["split"] and cs["min_long"] == 64
    is_long = lens >= 64
    rows_long = np.nonzero(is_long)[0]
    assert cs["n_long"] == len(rows_long) and np.array_equal(cs["long_rows"], rows_long.astype(np.uint32))
    ent_long = np.repeat(is_long, lens)
    off_l = np.zeros(len(rows_long) + 1, np.uint32)
    np.cumsum(lens[rows_long], out=off_l[1:])
    off_s = np.zeros(n_rows + 1, np.uint32)
    np.cumsum(np.where(is_long, 0, lens), out=off_s[1:])
    for name, want in (("long", (len(rows_long), off_l, col[ent_long], val[ent_long])), ("short", (n_rows, off_s, col[~ent_long], val[~ent_long]))):
        got = cs[name]
        assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]), name
        assert np.array_equal(bits(got[3]), bits(want[3])), name
    y = m.mvp(x, variant="colsplit")
    assert_spmv_close(y, off, col, val, x, "colsplit")
    assert np.array_equal(bits(y), bits(m.mvp(x, variant="colsplit")))  # run-to-run reproducible
    m.scale(-2.0)                                                        # both parts follow value changes
    assert_spmv_close(m.mvp(x, variant="colsplit"), off, col, (val * dtype(-2.0)).astype(dtype), x, "colsplit scaled")
    val2 = rng.uniform(-1, 1, len(val)).astype(dtype)
    m.update_values(val2)
    assert_spmv_close(m.mvp(x, variant="colsplit"), off, col, val2, x, "colsplit updated")


def test_colsplit_declines_when_it_is_not_worth_it(gpu):
    """Rows of similar length (no long minority): no split is kept; the variant runs K2c, bit for bit."""
    rng = np.random.default_rng(43)
    off, col, val = random_crs(rng, 4000, 3000, rng.integers(0, 40, 4000), np.float32)
    x = rng.uniform(-1, 1, 3000).astype(np.float32)
    m = sm.SparseMatCRS.from_raw_parts(4000, 3000, off, col, val)
    m.set_colblock_shift(8)
    assert not m.colsplit()["split"]
    assert np.array_equal(bits(m.mvp(x, variant="colsplit")), bits(m.mvp(x, variant="colblock")))
