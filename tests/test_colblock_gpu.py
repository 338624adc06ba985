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
    """x of 12 MB (f32): uniform columns -> K2c; banded columns -> K1r as before.  Parity on both."""
    rows, n, k = 200_000, 3_000_000, 16
    m_u = synth.crs_fixed(synth.SEED_MATRIX, 1, n, k, np.float32, 0, rows)
    assert m_u.resolved_variant()[0] == "colblock"
    cb = m_u.colblock(arrays=False)
    assert cb["n_blocks"] == 6 and cb["span_fraction"] > 0.9
    off, col, val = m_u.raw_parts()
    x = oracle.gen_x(synth.SEED_X, n, np.float32)
    y = m_u.mvp(x)
    assert_spmv_close(y, off, col, val, x, "auto colblock")
    for pattern in (0, 2):  # banded: K1r as before
        m_b = synth.crs_fixed(synth.SEED_MATRIX, pattern, n, k, np.float32, 1_000_000, 1_000_000 + rows)
        assert m_b.resolved_variant()[0] == "vector", pattern
        assert m_b.colblock(arrays=False)["span_fraction"] < 0.05


def test_cg_on_a_matrix_without_locality_runs_colblock(gpu):
    """A randomly permuted 7-point Laplacian (SPD, 1.33 M rows: 10.6 MB of f64 x, columns all over it) resolves to
    K2c; the device-resident CG (hipGraph replay of K2c launches) follows the oracle's iterates."""
    g, dtype = 110, np.float64
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
    assert m.resolved_variant()[0] == "colblock" and m.colblock(arrays=False)["n_blocks"] == 3
    b = oracle.spmv(off_p, col_p, val_p, np.ones(n, dtype))
    iters = 40
    x_ref, it_ref, rr_ref = oracle.cg(n, n, off_p, col_p, val_p, b, np.zeros(n, dtype), tol=1e-30, iter_max=iters)
    xd, bd = sm.DenseVec.from_vec(np.zeros(n, dtype)), sm.DenseVec.from_vec(b)
    cg = sm.ConjugateGradient(1e-30, iters)
    cg.solve(m, bd, xd)
    assert cg.iterations == it_ref == iters
    assert abs(cg.r_norm_squared - rr_ref) <= 1e-9 * rr_ref
    np.testing.assert_allclose(xd.to_numpy(), x_ref, rtol=0, atol=1e-10)
