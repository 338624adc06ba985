"""SparseMatrix::prod (sparsematrix.rs:186-210) and the predicates is_symmetric / is_sorted (:212-222, :251-271) on the
device, BIT-EXACT against the literal C restatement of the reference's loops (oracle.prod: column lists of rhs, stably
sorted rows of self, `sum += val * val_rhs` in list order, sums != 0 replayed with `set` on a SparseMatCRS); at scale
against scipy's SpGEMM (structure exact, values to rounding)."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
import sparsemat_amd as sm
from sparsemat_amd import _lib

pytestmark = pytest.mark.gpu


def same_crs(m, expect):
    n_rows, n_cols, off, col, val = expect[:5]
    assert (m.n_rows(), m.n_cols(), m.n_non_zero_entries()) == (n_rows, n_cols, len(col))
    g_off, g_col, g_val = m.raw_parts()
    assert np.array_equal(g_off[:n_rows + 1], off)
    assert np.array_equal(g_col, col)
    assert g_val.tobytes() == val.tobytes()


def random_crs(rng, n_rows, n_cols, max_len, dtype, duplicates, small_ints=False):
    """Rows in storage order (unsorted); with `duplicates` a column may repeat inside a row.  The last row / some entry
    reaches the last column so that n_cols is what a reference container would report."""
    lens = rng.integers(0, max_len + 1, n_rows)
    lens[-1] = max(lens[-1], 1)
    off = np.zeros(n_rows + 1, np.uint32)
    np.cumsum(lens, out=off[1:])
    col = np.empty(int(off[-1]), np.uint32)
    for r in range(n_rows):
        a, b = int(off[r]), int(off[r + 1])
        col[a:b] = rng.integers(0, n_cols, b - a) if duplicates else rng.permutation(n_cols)[:b - a]
    col[-1] = n_cols - 1
    if small_ints:  # exact cancellations: sums that come out as 0 must be dropped
        val = rng.integers(-2, 3, len(col)).astype(dtype)
    else:
        val = rng.uniform(-1, 1, len(col)).astype(dtype)
    return n_rows, n_cols, off, col, val


def device(m):
    return sm.SparseMatCRS.from_raw_parts(m[0], m[1], m[2], m[3], m[4])


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("kind", ["plain", "duplicates_both_sides", "cancellations", "rectangular"])
def test_prod_bit_exact(gpu, dtype, kind):
    rng = np.random.default_rng({"plain": 1, "duplicates_both_sides": 2, "cancellations": 3, "rectangular": 4}[kind])
    n, k = (120, 120) if kind != "rectangular" else (150, 70)
    dup = kind == "duplicates_both_sides"
    a = random_crs(rng, n, k, 9, dtype, dup, small_ints=kind == "cancellations")
    b = random_crs(rng, k, n, 9, dtype, dup, small_ints=kind == "cancellations")
    c = device(a).prod(device(b))
    expect = oracle.prod(a, b)
    same_crs(c, expect)
    if kind == "cancellations":  # some sums did come out as zero and were dropped (sparsematrix.rs:203)
        pattern = (sp.csr_matrix((np.ones_like(a[4]), a[3], a[2]), shape=(n, k))
                   @ sp.csr_matrix((np.ones_like(b[4]), b[3], b[2]), shape=(k, n))).tocsr()
        assert len(expect[3]) < pattern.nnz
    # descending columns inside every row
    off, col, _ = c.raw_parts()
    for r in range(c.n_rows()):
        assert np.all(np.diff(col[off[r]:off[r + 1]].astype(np.int64)) < 0)


def test_reference_prod_kat_on_device(gpu):
    """src/lib.rs:99-101: `let mp = sp_crs.prod(&sp).unwrap(); assert_eq!(mp.get(1, 2), 17.9632);` -- the matrix of
    check_sparsemat_indexlist times itself, through the device."""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")) as f:
        case = [c for c in json.load(f)["cases"] if c["name"] == "check_sparsemat_indexlist"][0]
    crs = case["crs"]
    val = np.array([int(b, 16) for b in crs["values_bits"]], np.uint32).view(np.float32)
    a = (crs["n_rows"], crs["n_cols"], np.array(crs["offset_rows"], np.uint32), np.array(crs["columns"], np.uint32), val)
    mp = device(a).prod(device(a))
    same_crs(mp, oracle.prod(a, a))
    off, col, v = mp.raw_parts()
    for i, j, lit in case["expect_prod_self"]:
        got = [v[q] for q in range(off[i], off[i + 1]) if col[q] == j]   # mp.get(i, j)
        assert got == [np.float32(float(lit))]


def test_prod_in_several_batches(gpu):
    """The products of a batch of rows are bounded (SMH_PROD_BATCH, read once per process, default 2^27): with a
    budget of 500 the same product goes through ~20 batches, in a child process, and is still bit-exact."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, oracle, sparsemat_amd as sm\n"
        "import torch; torch.cuda.init()\n"
        "rng = np.random.default_rng(7)\n"
        "def rc(n, k):\n"
        "    lens = rng.integers(0, 8, n); lens[-1] = 3\n"
        "    off = np.zeros(n + 1, np.uint32); np.cumsum(lens, out=off[1:])\n"
        "    col = rng.integers(0, k, int(off[-1])).astype(np.uint32); col[-1] = k - 1\n"
        "    return n, k, off, col, rng.uniform(-1, 1, len(col)).astype(np.float32)\n"
        "a, b = rc(400, 300), rc(300, 400)\n"
        "c = sm.SparseMatCRS.from_raw_parts(*a).prod(sm.SparseMatCRS.from_raw_parts(*b))\n"
        "e = oracle.prod(a, b)\n"
        "off, col, val = c.raw_parts()\n"
        "assert (c.n_rows(), c.n_cols()) == (e[0], e[1]) and np.array_equal(off, e[2]) and np.array_equal(col, e[3])\n"
        "assert val.tobytes() == e[4].tobytes()\n"
        "print('ok', len(col))\n")
    env = dict(os.environ, SMH_PROD_BATCH="500", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr


def test_prod_dimension_rule_and_tiny_results(gpu):
    f = np.float32
    a = device((2, 3, np.array([0, 1, 2], np.uint32), np.array([0, 2], np.uint32), np.array([1.0, 2.0], f)))
    b = device((3, 3, np.array([0, 1, 1, 2], np.uint32), np.array([0, 2], np.uint32), np.array([1.0, 1.0], f)))
    with pytest.raises(sm.SparseMatPanic) as e:   # Err(SparseMatError::new("Dimension mismatch")), sparsematrix.rs:188-190
        a.prod(b)
    assert e.value.status == _lib.SMH_ERR_DIM_MISMATCH and "Dimension mismatch" in str(e.value)
    # exactly one non-zero sum: the single `set` leaves n_rows == 0 (sparsemat_crs.rs:75-76)
    a1 = (2, 2, np.array([0, 1, 1], np.uint32), np.array([1], np.uint32), np.array([3.0], f))
    b1 = (2, 2, np.array([0, 0, 1], np.uint32), np.array([1], np.uint32), np.array([0.5], f))
    c = device(a1).prod(device(b1))
    exp = oracle.prod(a1, b1)
    assert (c.n_rows(), c.n_cols(), c.n_non_zero_entries()) == (0, 2, 0) == (exp[0], exp[1], len(exp[3])) and exp[5] == 1
    # no non-zero sum at all
    b0 = (2, 2, np.array([0, 1, 1], np.uint32), np.array([1], np.uint32), np.array([0.5], f))
    c = device(a1).prod(device(b0))
    assert (c.n_rows(), c.n_cols(), c.n_non_zero_entries()) == (0, 0, 0)
    same_crs(c, oracle.prod(a1, b0))


def test_prod_at_scale_against_scipy(gpu):
    """A = 7-point Laplacian 96^3 (f64): A.A has the 25-point pattern; integer values keep every sum exact, so the
    result must equal scipy's to the bit, and no sum cancels to zero."""
    g = 96
    a = sm.synth.crs_laplace3d(g, g, g, np.float64)
    off, col, val = a.raw_parts()
    n = g ** 3
    c = a.prod(a)
    ref = sp.csr_matrix((val, col, off), shape=(n, n))
    ref = (ref @ ref).tocsr()
    ref.sort_indices()
    assert ref.count_nonzero() == ref.nnz
    c.sort_rows()
    c_off, c_col, c_val = c.raw_parts()
    assert (c.n_rows(), c.n_cols()) == (n, n)
    assert np.array_equal(c_off, ref.indptr.astype(np.uint32)) and np.array_equal(c_col, ref.indices.astype(np.uint32))
    assert c_val.tobytes() == ref.data.tobytes()


def test_is_symmetric_and_is_sorted(gpu):
    a = sm.synth.crs_laplace3d(20, 20, 20, np.float32)
    assert a.is_symmetric() and a.is_sorted()
    off, col, val = a.raw_parts()
    n = a.n_rows()
    v2 = val.copy()
    v2[int(off[n // 2])] += np.float32(0.5)       # one off-diagonal entry differs from its mirror
    assert col[int(off[n // 2])] != n // 2
    assert not sm.SparseMatCRS.from_raw_parts(n, n, off, col, v2).is_symmetric()
    c2 = col.copy()
    a0, b0 = int(off[7]), int(off[8])
    c2[a0:b0] = c2[a0:b0][::-1]
    v3 = val.copy()
    v3[a0:b0] = v3[a0:b0][::-1]
    m = sm.SparseMatCRS.from_raw_parts(n, n, off, c2, v3)
    assert not m.is_sorted() and m.is_symmetric()  # storage order does not matter to get(j, i)
    # a stored entry whose mirror is absent (get -> zero), a stored zero whose mirror is absent (0 == 0: symmetric)
    f = np.float64
    assert not sm.SparseMatCRS.from_raw_parts(2, 2, np.array([0, 1, 1], np.uint32), np.array([1], np.uint32), np.array([2.0], f)).is_symmetric()
    assert sm.SparseMatCRS.from_raw_parts(2, 2, np.array([0, 1, 1], np.uint32), np.array([1], np.uint32), np.array([0.0], f)).is_symmetric()
    # column beyond the rows: get(j, i) with j >= n_rows is zero
    assert not sm.SparseMatCRS.from_raw_parts(1, 3, np.array([0, 1], np.uint32), np.array([2], np.uint32), np.array([1.0], f)).is_symmetric()
    # duplicates: get takes the FIRST match in storage order (sparsemat_crs.rs:54-67)
    dup = sm.SparseMatCRS.from_raw_parts(2, 2, np.array([0, 2, 3], np.uint32), np.array([1, 1, 0], np.uint32), np.array([5.0, 7.0, 5.0], f))
    assert not dup.is_symmetric()   # entry (0,1)=7 vs get(1,0)=5
    nan = sm.SparseMatCRS.from_raw_parts(1, 1, np.array([0, 1], np.uint32), np.array([0], np.uint32), np.array([np.nan], f))
    assert not nan.is_symmetric()   # NaN != NaN
    empty = sm.SparseMatCRS.from_raw_parts(3, 3, np.zeros(4, np.uint32), np.zeros(0, np.uint32), np.zeros(0, f))
    assert empty.is_symmetric() and empty.is_sorted()


def test_new_entry_points_reject_bad_arguments(gpu):
    """NULL handles / outputs and mismatched value types come back as statuses, never as crashes."""
    import ctypes as C
    L = sm.lib()
    h = C.c_void_p()
    out_i = C.c_int()
    assert L.smh_crs_transpose(None, C.byref(h)) == _lib.SMH_ERR_INVALID
    assert L.smh_crs_prod(None, None, C.byref(h)) == _lib.SMH_ERR_INVALID
    assert L.smh_crs_is_symmetric(None, C.byref(out_i)) == _lib.SMH_ERR_INVALID
    assert L.smh_crs_is_sorted(None, C.byref(out_i)) == _lib.SMH_ERR_INVALID
    assert L.smh_crs_column_info(None, None, None, None) == _lib.SMH_ERR_INVALID
    assert L.smh_crs_replay(0, 3, None, None, None, None, C.byref(h)) == _lib.SMH_ERR_INVALID
    assert L.smh_par_create(0, 2, None, 4, 4, None, None, None, 1, C.byref(h)) == _lib.SMH_ERR_INVALID
    assert L.smh_par_spmv(None, None, 0, None, 0) == _lib.SMH_ERR_INVALID
    assert L.smh_par_cg_solve(None, None, 0, None, 0, 1e-6, 10, 0, None, None) == _lib.SMH_ERR_INVALID
    assert L.smh_pcg_jacobi_solve(None, None, 0, None, 0, 1e-6, 10, 0, None, None) == _lib.SMH_ERR_INVALID
    assert L.smh_par_destroy(None) == 0 and L.smh_par_n_blocks(None) == 0
    a32 = device((2, 2, np.array([0, 1, 2], np.uint32), np.array([0, 1], np.uint32), np.array([1.0, 2.0], np.float32)))
    a64 = device((2, 2, np.array([0, 1, 2], np.uint32), np.array([0, 1], np.uint32), np.array([1.0, 2.0], np.float64)))
    with pytest.raises(sm.SparseMatPanic) as e:
        a32.prod(a64)
    assert e.value.status == _lib.SMH_ERR_INVALID
    # an empty left operand, an operand without entries
    z = device((2, 2, np.zeros(3, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.float32)))
    c = z.prod(a32)
    assert (c.n_rows(), c.n_cols(), c.n_non_zero_entries()) == (0, 0, 0)
    c = a32.prod(z)
    assert (c.n_rows(), c.n_cols(), c.n_non_zero_entries()) == (0, 0, 0)
    assert z.is_symmetric() and z.is_sorted()
    t = z.transpose()
    assert (t.n_rows(), t.n_cols(), t.n_non_zero_entries()) == (0, 0, 0)
