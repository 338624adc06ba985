"""Jacobi-preconditioned CG (smh_pcg_jacobi_solve) -- an EXTENSION: the reference has no
preconditioner, so parity here is against the oracle's restatement of the same recurrence (ConjugateGradient::solve,
linearsolver.rs:27-61, with z = r / diag(A)), itself checked against scipy on the CPU side of this file's first test;
the guards, the stop rule and the panics are the reference's."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
import sparsemat_amd as sm
from sparsemat_amd import _lib

pytestmark = pytest.mark.gpu


def scaled_laplacian(g, dtype, spread):
    """S A S with A the 7-point Laplacian and S = diag(10^u), u uniform in [-spread/2, spread/2]: SPD, badly scaled --
    what a diagonal preconditioner exists for.  Rows stay in storage order."""
    off, col, val = oracle.laplace3d(g, g, g, np.float64)
    n = g ** 3
    rng = np.random.default_rng(1)
    s = 10.0 ** rng.uniform(-spread / 2, spread / 2, n)
    rows = np.repeat(np.arange(n), np.diff(off.astype(np.int64)))
    v = (val * s[rows] * s[col]).astype(dtype)
    return n, off, col, v


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_pcg_matches_the_restatement_and_beats_plain_cg(gpu, dtype):
    n, off, col, val = scaled_laplacian(16, dtype, 3.0)
    rng = np.random.default_rng(2)
    x_true = rng.uniform(-1, 1, n).astype(dtype)
    b = oracle.spmv(off, col, val, x_true)
    tol = float(np.linalg.norm(b)) * (1e-5 if dtype == np.float32 else 1e-11)
    a = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val)
    x = np.zeros(n, dtype)
    pcg = sm.JacobiConjugateGradient(tol, 5000)
    pcg.solve(a, b, x)
    o_x, o_iters, o_rr = oracle.pcg_jacobi(n, n, off, col, val, b, np.zeros(n, dtype), tol=tol, iter_max=5000)
    assert np.sqrt(pcg.r_norm_squared) < tol and np.sqrt(o_rr) < tol
    assert abs(pcg.iterations - o_iters) <= max(3, o_iters // 20), (pcg.iterations, o_iters)
    # both reach the solution of the system (the true residual is what tol bounds)
    m = sp.csr_matrix((val.astype(np.float64), col, off), shape=(n, n))
    for sol in (x, o_x):
        assert np.linalg.norm(m @ sol.astype(np.float64) - b) < 20 * tol
    # ... and the preconditioner pays: the reference's plain CG needs several times the iterations on this matrix
    cg = sm.ConjugateGradient(tol, 5000)
    x_plain = np.zeros(n, dtype)
    cg.solve(a, b, x_plain)
    assert cg.iterations > 2 * pcg.iterations, (cg.iterations, pcg.iterations)
    # a fixed number of iterations gives the same iterate as the restatement (tolerance of a regrouped reduction)
    x3 = np.zeros(n, dtype)
    p3 = sm.JacobiConjugateGradient(0.0, 3)
    p3.solve(a, b, x3)
    o3, it3, _ = oracle.pcg_jacobi(n, n, off, col, val, b, np.zeros(n, dtype), tol=0.0, iter_max=3)
    assert p3.iterations == 3 == it3
    scale = np.max(np.abs(o3))
    assert np.max(np.abs(x3.astype(np.float64) - o3.astype(np.float64))) <= (2e-5 if dtype == np.float32 else 1e-12) * scale


def test_pcg_identity_scaling_equals_cg_iterates(gpu):
    """diag(A) = 1: z = r exactly, so the preconditioned recurrence is the reference's CG (same iteration count)."""
    g, dtype = 10, np.float64
    off, col, val = oracle.laplace3d(g, g, g, dtype)
    n = g ** 3
    val = val / 6.0  # diagonal 1, off-diagonals -1/6
    b = np.ones(n, dtype)
    a = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val)
    x1, x2 = np.zeros(n, dtype), np.zeros(n, dtype)
    pcg, cg = sm.JacobiConjugateGradient(1e-10, 1000), sm.ConjugateGradient(1e-10, 1000)
    pcg.solve(a, b, x1)
    cg.solve(a, b, x2)
    assert pcg.iterations == cg.iterations
    assert np.max(np.abs(x1 - x2)) < 1e-10


def test_pcg_guards(gpu):
    f = np.float64
    off, col, val = oracle.laplace3d(4, 4, 4, f)
    a = sm.SparseMatCRS.from_raw_parts(64, 64, off, col, val)
    pcg = sm.JacobiConjugateGradient()
    with pytest.raises(sm.SparseMatPanic) as e:   # linearsolver.rs:33-36
        pcg.solve(a, np.ones(63, f), np.zeros(64, f))
    assert e.value.status == _lib.SMH_ERR_DIM_MISMATCH and "Matrix and vector size mismatch" in str(e.value)
    rect = sm.SparseMatCRS.from_raw_parts(64, 70, off, col, val)
    with pytest.raises(sm.SparseMatPanic) as e:   # linearsolver.rs:30-32
        pcg.solve(rect, np.ones(64, f), np.zeros(64, f))
    assert e.value.status == _lib.SMH_ERR_NOT_SQUARE
    v0 = val.copy()
    v0[int(np.flatnonzero(col[off[5]:off[6]] == 5)[0]) + int(off[5])] = 0.0
    with pytest.raises(sm.SparseMatPanic) as e:   # nothing to divide by
        pcg.solve(sm.SparseMatCRS.from_raw_parts(64, 64, off, col, v0), np.ones(64, f), np.zeros(64, f))
    assert e.value.status == _lib.SMH_ERR_INVALID and "row 5" in str(e.value)
    with pytest.raises(oracle.OraclePanic):
        oracle.pcg_jacobi(64, 64, off, col, v0, np.ones(64, f), np.zeros(64, f))
