"""The exchange beside the product (csrc/par.hip; SURVEY 5 / 8e "overlap the gather with the next block of rows"): a block's rows
other blocks reference are multiplied first, the window exchange runs on a second stream per block, the block's INTERIOR rows are
multiplied meanwhile.  The reference's blocks are independent (sparsemat_par.rs:54-64, results at b R), so the order of a block's
rows is free of semantics -- and here it is free of arithmetic too: the same kernels take the same rows with the same lanes, so y
and every CG iterate must be BIT-IDENTICAL with the overlap on and off, and equal to the oracle where the kernel is bit-exact."""
import numpy as np
import pytest

import oracle
import sparsemat_amd as sm
from sparsemat_amd import synth

pytestmark = pytest.mark.gpu


def banded(rng, n, k, band, dtype):
    off = np.arange(n + 1, dtype=np.uint32) * k
    base = np.arange(n, dtype=np.int64)[:, None] + rng.integers(-band, band + 1, (n, k))
    col = np.clip(base, 0, n - 1).astype(np.uint32).reshape(-1)
    return off, col, rng.uniform(-1, 1, n * k).astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("n_blocks", [2, 4, 7])
def test_interior_rows_and_bit_identical_products(gpu, dtype, n_blocks):
    rng = np.random.default_rng(40 + n_blocks)
    n, k, band = 120_011, 16, 700
    off, col, val = banded(rng, n, k, band, dtype)
    m = sm.SparseMatParLocal.with_sub_matrices(n_blocks, n, n, off, col, val, device_ids=[0] * n_blocks)
    assert m.exchange_mode("auto")[0] == "window"
    r = n // n_blocks
    cols2d = col.reshape(n, k).astype(np.int64)
    for variant in ("stream", "vector"):
        for b in range(n_blocks):
            r0, r1 = b * r, (n if b == n_blocks - 1 else (b + 1) * r)
            a, e = m.interior(b, variant)
            assert 0 < e - a < r1 - r0, (variant, b, a, e)        # there is an interior, and there are boundary rows
            inner = cols2d[r0 + a:r0 + e]
            assert inner.min() >= r0 and inner.max() < r1           # interior rows reference the block's own slice only
            others = np.concatenate([cols2d[:r0].reshape(-1), cols2d[r1:].reshape(-1)])
            assert not ((others >= r0 + a) & (others < r0 + e)).any()  # ... and no other block references them
            assert e - a > (r1 - r0) - 2 * (2 * band + 2048)        # not much more than the band is held back
    with pytest.raises(sm.SparseMatPanic):
        m.interior(n_blocks, "stream")
    assert m.interior(0, "merge") == (0, 0)                         # a kernel that cannot be launched by runs of rows: no split
    x_host = rng.uniform(-1, 1, n).astype(dtype)
    want = oracle.spmv(off, col, val, x_host)
    want2 = oracle.spmv(off, col, val, want)
    for variant in ("stream", "vector", "auto", "merge"):
        got = {}
        for on in (True, False):
            m.set_overlap(on)
            x, y, z = m.vec(host=x_host), m.vec(), m.vec()
            m.mvp_dev(x, y, variant=variant)
            m.mvp_dev(y, z, variant=variant)   # the exchanged y IS the next x: every block must hold what it references
            m.synchronize()
            got[on] = (y.download(), z.download(), [z.download_block(b) for b in range(n_blocks)])
        assert got[True][0].tobytes() == got[False][0].tobytes() and got[True][1].tobytes() == got[False][1].tobytes(), variant
        for b in range(n_blocks):  # ... including the halo every block received
            r0, r1 = b * r, (n if b == n_blocks - 1 else (b + 1) * r)
            lo, hi = int(cols2d[r0:r1].min()), int(cols2d[r0:r1].max()) + 1
            assert got[True][2][b][lo:hi].tobytes() == got[False][2][b][lo:hi].tobytes(), (variant, b)
        if variant == "stream":  # K1s: the reference's order of additions
            assert got[True][0].tobytes() == want.tobytes() and got[True][1].tobytes() == want2.tobytes()
    m.set_overlap(True)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_cg_iterates_are_bit_identical_with_and_without_the_overlap(gpu, dtype):
    g = 40
    off, col, val = oracle.laplace3d(g, g, g, dtype)
    n = g * g * g
    b_host = oracle.spmv(off, col, val, np.ones(n, dtype))
    for n_blocks in (2, 5):
        m = sm.SparseMatParLocal.with_sub_matrices(n_blocks, n, n, off, col, val, device_ids=[0] * n_blocks)
        assert all(m.interior(b)[1] > m.interior(b)[0] for b in range(n_blocks))  # AUTO = K1s: 256-row tiles
        res = {}
        for on in (True, False):
            m.set_overlap(on)
            for iters in (1, 7, 400):
                b, x = m.vec(host=b_host), m.vec()
                res[(on, iters)] = (m.cg_solve_vec(b, x, tol=1e-6 if dtype == np.float32 else 1e-10, iter_max=iters, check_every=3), x.download())
        for iters in (1, 7, 400):
            (it_a, rr_a), xa = res[(True, iters)]
            (it_b, rr_b), xb = res[(False, iters)]
            assert it_a == it_b and rr_a == rr_b and xa.tobytes() == xb.tobytes(), (n_blocks, iters)
        assert res[(True, 400)][0][0] < 400 and np.abs(res[(True, 400)][1] - 1).max() < 1e-3


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_issuing_threads_change_no_bit(gpu, dtype):
    """One issuing host thread per block (smh_par_set_threads(1)) against ONE thread issuing
    block after block: the product step and every CG iterate are bit for bit the same -- streams, events, kernels and their order
    per block do not depend on who issues them -- with the overlap on and off, window and all-gather exchange, 3 and 8 blocks."""
    g = 40
    off, col, val = oracle.laplace3d(g, g, g, dtype)
    n = g * g * g
    rng = np.random.default_rng(12)
    x_host = rng.uniform(-1, 1, n).astype(dtype)
    b_host = oracle.spmv(off, col, val, np.ones(n, dtype))
    want = oracle.spmv(off, col, val, x_host)
    for n_blocks in (3, 8):
        m = sm.SparseMatParLocal.with_sub_matrices(n_blocks, n, n, off, col, val, device_ids=[0] * n_blocks)
        res = {}
        for threads in (1, 0):
            m.set_threads(threads)
            for overlap in (True, False):
                m.set_overlap(overlap)
                for exch in ("window", "allgather"):
                    x, y, z = m.vec(host=x_host), m.vec(), m.vec()
                    m.mvp_dev(x, y, variant="stream", exchange=exch)
                    m.mvp_dev(y, z, variant="stream", exchange=exch)
                    m.synchronize()
                    assert y.download().tobytes() == want.tobytes(), (threads, overlap, exch)
                    res[(threads, overlap, exch)] = z.download().tobytes()
                for iters in (1, 9, 300):
                    b, xs = m.vec(host=b_host), m.vec()
                    res[(threads, overlap, iters)] = (m.cg_solve_vec(b, xs, tol=1e-6 if dtype == np.float32 else 1e-10, iter_max=iters, check_every=4),
                                                      xs.download().tobytes())
        for key in [k for k in res if k[0] == 1]:
            assert res[key] == res[(0,) + key[1:]], (n_blocks, key)
        assert res[(1, True, 300)][0][0] < 300
        m.set_threads(-1)
        m.set_overlap(True)


def test_blocks_without_an_interior_still_multiply(gpu):
    """Scattered columns: every row references other blocks -- AUTO's exchange is the all-gather (no overlap), and a forced window
    finds no interior: the call is then simply not split."""
    rng = np.random.default_rng(44)
    n, k = 30_000, 8
    off = np.arange(n + 1, dtype=np.uint32) * k
    col = rng.integers(0, n, n * k).astype(np.uint32)
    val = rng.uniform(-1, 1, n * k).astype(np.float32)
    m = sm.SparseMatParLocal.with_sub_matrices(3, n, n, off, col, val, device_ids=[0, 0, 0])
    assert m.exchange_mode("auto")[0] == "allgather" and all(m.interior(b, "stream") == (0, 0) for b in range(3))
    x_host = rng.uniform(-1, 1, n).astype(np.float32)
    x, y = m.vec(host=x_host), m.vec()
    m.mvp_dev(x, y, variant="stream", exchange="window")
    m.synchronize()
    assert y.download().tobytes() == oracle.spmv(off, col, val, x_host).tobytes()


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_k1s_xd_blocks_launched_by_runs_of_rows(gpu, dtype):
    """Blocks whose STREAM kernel is K1s XD (x staged in LDS, stage offsets; forced here, automatic from 8 MB of x): the overlapped
    product launches it over a block's boundary tiles and its interior tiles separately, the CG with the dot epilogue on top --
    same bits as the oracle / as without the overlap."""
    g, n_blocks = 40, 3
    off, col, val = oracle.laplace3d(g, g, g, dtype)
    n = g * g * g
    r = n // n_blocks
    blocks = []
    for b in range(n_blocks):
        r0, r1 = b * r, (n if b == n_blocks - 1 else (b + 1) * r)
        k0, k1 = int(off[r0]), int(off[r1])
        blk = sm.SparseMatCRS.from_raw_parts(r1 - r0, n, (off[r0:r1 + 1] - off[r0]).astype(np.uint32), col[k0:k1], val[k0:k1])
        blk.set_stream_xs(1)
        assert blk.stream_direct()
        blocks.append(blk)
    m = sm.SparseMatParLocal.adopt(blocks, n)
    assert all(m.interior(b, "stream")[1] > m.interior(b, "stream")[0] for b in range(n_blocks))
    rng = np.random.default_rng(8)
    x_host = rng.uniform(-1, 1, n).astype(dtype)
    want = oracle.spmv(off, col, val, x_host)
    b_host = oracle.spmv(off, col, val, np.ones(n, dtype))
    res = {}
    for on in (True, False):
        m.set_overlap(on)
        x, y = m.vec(host=x_host), m.vec()
        m.mvp_dev(x, y, variant="stream", exchange="window")
        m.synchronize()
        assert y.download().tobytes() == want.tobytes(), on
        b, xs = m.vec(host=b_host), m.vec()
        res[on] = (m.cg_solve_vec(b, xs, tol=1e-6 if dtype == np.float32 else 1e-10, iter_max=60, check_every=3), xs.download())
    assert res[True][0] == res[False][0] and res[True][1].tobytes() == res[False][1].tobytes()
    # ... and the same iterates as with the classic bodies (the dot epilogue folds the same partial sums in the same order)
    for blk in blocks:
        blk.set_stream_direct(0)
    b, xs = m.vec(host=b_host), m.vec()
    again = (m.cg_solve_vec(b, xs, tol=1e-6 if dtype == np.float32 else 1e-10, iter_max=60, check_every=3), xs.download())
    assert again[0] == res[True][0] and again[1].tobytes() == res[True][1].tobytes()


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_value_dictionary_blocks_of_many_tiles_by_runs_of_rows(gpu, dtype):
    """BASELINE C1's matrix (5-point Laplacian 1000 x 1000: two distinct values -> K1s XD-V) cut into three ragged blocks of ~1300 tiles:
    in f64 the kernel runs on persistent workgroups that walk SEVERAL tiles each, here over the runs of rows the overlapped product
    launches separately (boundary tiles, interior tiles: tile ranges that do not start at 0).  Same bits as the oracle with the overlap
    on and off, the solver's iterates identical either way."""
    g, n_blocks = 1000, 3
    off, col, val = oracle.laplace2d(g, g, dtype)
    n = g * g
    cuts = [0, 333_411, 667_003, n]
    blocks = []
    for b in range(n_blocks):
        r0, r1 = cuts[b], cuts[b + 1]
        k0, k1 = int(off[r0]), int(off[r1])
        blk = sm.SparseMatCRS.from_raw_parts(r1 - r0, n, (off[r0:r1 + 1] - off[r0]).astype(np.uint32), col[k0:k1], val[k0:k1])
        blk.set_stream_xs(1)
        assert blk.stream_direct() and len(blk.stream_value_dict()) == 2
        blocks.append(blk)
    m = sm.SparseMatParLocal.adopt(blocks, n, split_rows=cuts)
    rng = np.random.default_rng(18)
    x_host = rng.uniform(-1, 1, n).astype(dtype)
    want = oracle.spmv(off, col, val, x_host)
    b_host = oracle.spmv(off, col, val, np.ones(n, dtype))
    res = {}
    for on in (True, False):
        m.set_overlap(on)
        x, y = m.vec(host=x_host), m.vec()
        m.mvp_dev(x, y, variant="stream", exchange="window")
        m.synchronize()
        assert y.download().tobytes() == want.tobytes(), on
        b, xs = m.vec(host=b_host), m.vec()
        res[on] = (m.cg_solve_vec(b, xs, tol=1e-30, iter_max=12, check_every=4), xs.download())
    assert res[True][0] == res[False][0] and res[True][1].tobytes() == res[False][1].tobytes()
