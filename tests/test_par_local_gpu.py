"""SparseMatPar behind the C ABI, one process (smh_par_*, csrc/par.hip): row blocks on devices, y = A x with
all blocks running concurrently, ConjugateGradient::solve with device-to-device halo exchange.  On the one-GPU test box
every block lives on device 0 -- the partition, the window logic, the peer copies and the host-side folds are the same
code that places block b on device b."""
import json
import os

import numpy as np
import pytest

import oracle
import sparsemat_amd as sm
from sparsemat_amd import _lib

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")


def sm_device_count():
    import ctypes as C
    n = C.c_int(0)
    _lib.check(sm.lib().smh_device_count(C.byref(n)))
    return n.value


def test_reference_par_kat(gpu):
    """src/lib.rs:180-202: SparseMatPar::with_sub_matrices(4, 16), the add_to/set calls of the other container tests,
    assert_eq!(mvp.get(0), 34.544)."""
    with open(GOLDEN) as f:
        case = [c for c in json.load(f)["cases"] if c["name"] == "check_sparsemat_par"][0]
    crs, par = case["crs"], case["par"]
    n_rows = par["max_n_rows"]  # rows 3..15 stay empty; the reference's n_rows() reports 3 (:95-107)
    off = np.array(crs["offset_rows"] + [crs["offset_rows"][-1]] * (n_rows - crs["n_rows"]), np.uint32)
    val = np.array([int(b, 16) for b in crs["values_bits"]], np.uint32).view(np.float32)
    m = sm.SparseMatParLocal.with_sub_matrices(par["n_blocks"], n_rows, crs["n_cols"], off, crs["columns"], val, device_ids=[0] * 4)
    assert (m.n_blocks(), m.rows_per_block(), m.n_cols(), m.n_non_zero_entries()) == (4, par["rows_per_block"], 3, 6)
    assert m.block(0)[:2] == (0, 4) and m.block(3)[:2] == (12, 16) and m.block(0)[3] == 6 and m.block(1)[3] == 0
    x = np.array([np.float32(float(s)) for s in case["x"]], np.float32)
    y = m.mvp(x, variant="stream")
    for i, lit in case["expect_mvp"]:
        assert y[i] == np.float32(float(lit))
    assert np.all(y[crs["n_rows"]:] == 0)
    # device_ids = NULL: block b on device b mod device count
    auto = sm.SparseMatParLocal.with_sub_matrices(par["n_blocks"], n_rows, crs["n_cols"], off, crs["columns"], val)
    assert [auto.block(b)[2] for b in range(4)] == [b % max(1, sm_device_count()) for b in range(4)]
    assert auto.mvp(x, variant="stream").tobytes() == y.tobytes()
    # get_block_and_row_id (sparsemat_par.rs:31-35); beyond n_blocks * R the LAST block is taken (the reference clamps to
    # n_blocks and panics)
    assert m.get_block_and_row_id(0) == (0, 0) and m.get_block_and_row_id(7) == (1, 3) and m.get_block_and_row_id(15) == (3, 3)
    assert m.get_block_and_row_id(17) == (3, 5)


def random_crs(rng, n_rows, n_cols, max_len, dtype, band=None):
    lens = rng.integers(0, max_len + 1, n_rows)
    off = np.zeros(n_rows + 1, np.uint32)
    np.cumsum(lens, out=off[1:])
    col = np.empty(int(off[-1]), np.uint32)
    for r in range(n_rows):
        a, b = int(off[r]), int(off[r + 1])
        if band is None:
            col[a:b] = rng.integers(0, n_cols, b - a)
        else:
            lo, hi = max(0, r - band), min(n_cols, r + band + 1)
            col[a:b] = rng.integers(lo, hi, b - a)
    return off, col, rng.uniform(-1, 1, len(col)).astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("n_blocks", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("shape", ["banded", "scattered"])
def test_par_spmv_bit_exact(gpu, dtype, n_blocks, shape):
    rng = np.random.default_rng(n_blocks * 10 + (shape == "banded"))
    n_rows, n_cols = 5003, 4801  # neither divisible by the block counts: the last block takes the remainder
    off, col, val = random_crs(rng, n_rows, n_cols, 20, dtype, band=300 if shape == "banded" else None)
    m = sm.SparseMatParLocal.with_sub_matrices(n_blocks, n_rows, n_cols, off, col, val, device_ids=[0] * n_blocks)
    r = n_rows // n_blocks
    assert m.rows_per_block() == r and m.block(n_blocks - 1)[:2] == ((n_blocks - 1) * r, n_rows)
    assert m.n_non_zero_entries() == len(col)
    x = rng.uniform(-1, 1, n_cols).astype(dtype)
    want = oracle.spmv(off, col, val, x)
    assert m.mvp(x, variant="stream").tobytes() == want.tobytes()   # K1s: the reference's order of additions
    got = m.mvp(x)                                                    # whatever AUTO picks per block
    scale = oracle.spmv_abs(off, col, val, x)
    assert np.all(np.abs(got.astype(np.float64) - want) <= (1e-5 if dtype == np.float32 else 1e-12) * scale + 1e-300)
    m.scale(0.5)
    assert m.mvp(x, variant="stream").tobytes() == oracle.spmv(off, col, (val * dtype(0.5)).astype(dtype), x).tobytes()


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("n_blocks", [1, 2, 4, 7])
def test_par_cg_matches_oracle(gpu, dtype, n_blocks):
    g = 14
    off, col, val = oracle.laplace3d(g, g, g, dtype)
    n = g ** 3
    rng = np.random.default_rng(3)
    x_true = rng.uniform(-1, 1, n).astype(dtype)
    b = oracle.spmv(off, col, val, x_true)
    tol = 1e-4 if dtype == np.float32 else 1e-10
    m = sm.SparseMatParLocal.with_sub_matrices(n_blocks, n, n, off, col, val, device_ids=[0] * n_blocks)
    x = np.zeros(n, dtype)
    iters, rr = m.cg_solve(b, x, tol=tol, iter_max=500)
    o_x, o_iters, o_rr = oracle.cg(n, n, off, col, val, b, np.zeros(n, dtype), tol=tol, iter_max=500)
    assert abs(iters - o_iters) <= 1 and np.sqrt(rr) < tol   # measured: the oracle's count for 1..7 blocks
    err = np.max(np.abs(x.astype(np.float64) - o_x.astype(np.float64)))
    assert err < 10 * tol, err                                # measured <= 0.03 tol
    # the single-device solver on the same system
    a = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val)
    cg = sm.ConjugateGradient(tol, 500)
    x1 = sm.DenseVec.zeros(n, dtype)
    cg.solve(a, sm.DenseVec.from_vec(b), x1)
    assert abs(cg.iterations - iters) <= 1
    # iter_max is honoured, x is the iterate reached
    x2 = np.zeros(n, dtype)
    it2, _ = m.cg_solve(b, x2, tol=tol, iter_max=3)
    o_x2, o_it2, _ = oracle.cg(n, n, off, col, val, b, np.zeros(n, dtype), tol=tol, iter_max=3)
    assert it2 == 3 == o_it2
    assert np.max(np.abs(x2.astype(np.float64) - o_x2.astype(np.float64))) < (1e-4 if dtype == np.float32 else 1e-11)


def test_par_errors_mirror_the_reference(gpu):
    f = np.float64
    off, col, val = oracle.laplace3d(4, 4, 4, f)
    m = sm.SparseMatParLocal.with_sub_matrices(3, 64, 64, off, col, val, device_ids=[0, 0, 0])
    with pytest.raises(sm.SparseMatPanic) as e:   # linearsolver.rs:33-36
        m.cg_solve(np.ones(63, f), np.zeros(64, f))
    assert e.value.status == _lib.SMH_ERR_DIM_MISMATCH and "Matrix and vector size mismatch" in str(e.value)
    with pytest.raises(sm.SparseMatPanic) as e:   # densevec.rs:41: rhs.get(j) beyond the vector
        m.mvp(np.ones(10, f))
    assert e.value.status == _lib.SMH_ERR_INDEX_RANGE
    rect = sm.SparseMatParLocal.with_sub_matrices(2, 64, 70, off, col, val, device_ids=[0, 0])
    with pytest.raises(sm.SparseMatPanic) as e:   # linearsolver.rs:30-32
        rect.cg_solve(np.ones(64, f), np.zeros(64, f))
    assert e.value.status == _lib.SMH_ERR_NOT_SQUARE and "Matrix is not symmetric" in str(e.value)
    with pytest.raises(sm.SparseMatPanic) as e:   # more blocks than rows: R = 0 (the reference divides by zero)
        sm.SparseMatParLocal.with_sub_matrices(65, 64, 64, off, col, val, device_ids=[0] * 65)
    assert e.value.status == _lib.SMH_ERR_INVALID
    with pytest.raises(sm.SparseMatPanic):        # no such device
        sm.SparseMatParLocal.with_sub_matrices(2, 64, 64, off, col, val, device_ids=[0, 99])
