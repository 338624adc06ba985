"""CPU: the oracle pinned to the reference's own known-answer tests (tests/golden/reference_kats.json),
the assembly restatement's storage order, and independent cross-checks (scipy)."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import oracle
from oracle import assembly

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")
with open(GOLDEN) as f:
    CASES = json.load(f)["cases"]


def _values(case):
    crs = case["crs"]
    if case["dtype"] == "f32":
        return np.array([int(b, 16) for b in crs["values_bits"]], dtype=np.uint32).view(np.float32)
    return np.array([int(b, 16) for b in crs["values_bits"]], dtype=np.uint64).view(np.float64)


@pytest.mark.parametrize("case", [c for c in CASES if "expect_mvp" in c], ids=lambda c: c["name"])
def test_oracle_spmv_reproduces_reference_asserts(case):
    """assert_eq!(mvp.get(0), 34.544) / 20.16 (src/lib.rs:81,151,175,199): exact f32 equality."""
    crs, dt = case["crs"], np.float32
    val = _values(case)
    x = np.array([dt(float(s)) for s in case["x"]], dtype=dt)
    y = oracle.spmv(crs["offset_rows"], crs["columns"], val, x)
    assert len(y) == crs["n_rows"]  # dim == n_rows (vector.rs:40-42 growth)
    for i, lit in case["expect_mvp"]:
        assert y[i] == dt(float(lit))
    # a sorted-column accumulation does NOT reproduce 34.544: the oracle must keep storage order
    if case["name"] == "check_sparsemat_indexlist":
        off, col = crs["offset_rows"], np.array(crs["columns"])
        order = np.argsort(col[off[0]:off[1]], kind="stable")
        s = dt(0)
        for k in order:
            s = dt(s + dt(x[col[k]] * val[k]))
        assert s != dt(34.544)


@pytest.mark.parametrize("case", [c for c in CASES if "ops" in c], ids=lambda c: c["name"])
def test_assembly_restatement_storage_order(case):
    """Replaying the reference tests' call sequences yields the committed CRS arrays and the iteration
    orders the reference asserts (src/lib.rs:67-71, 122-128)."""
    dt = np.float32 if case["dtype"] == "f32" else np.float64
    name = case["name"]
    if name == "check_sparsemat_crs":
        m = assembly.CrsPushMatrix(dt)
    elif name == "check_sparsemat_par":
        m = assembly.ParMatrix(case["par"]["n_blocks"], case["par"]["max_n_rows"], dt)
    else:
        m = assembly.IndexListMatrix(dt)
    for op, i, j, lit in case["ops"]:
        getattr(m, op)(i, j, dt(float(lit)))
    n_rows, n_cols, off, col, val = m.to_crs_arrays()
    crs = case["crs"]
    assert (n_rows, n_cols) == (crs["n_rows"], crs["n_cols"])
    assert list(off) == crs["offset_rows"] and list(col) == crs["columns"]
    assert np.array_equal(val, _values(case))
    flat = [(i, int(col[k]), val[k]) for i in range(n_rows) for k in range(off[i], off[i + 1])]
    for key in ("iter_prefix", "iter_full"):
        if key in case:
            for (i, j, lit), got in zip(case[key], flat):
                assert (i, j) == got[:2] and got[2] == dt(float(lit))
            if key == "iter_full":
                assert len(flat) == len(case[key])
    for i, j, lit in case.get("expect_get", []):
        assert m.get(i, j) == dt(float(lit))
    for i, dense in case.get("expect_row_dense", []):
        row = np.zeros(n_cols, dt)
        for k in range(off[i], off[i + 1]):
            row[col[k]] = val[k]
        assert [("%s" % v) for v in row] == ["0.0" if d == "0" else d for d in dense]
    if "expect_density" in case:
        num, den = case["expect_density"]
        assert len(val) / (n_rows * n_cols) == num / den


def _ops_arrays(case, dt):
    ops = case["ops"]
    return ([o[1] for o in ops], [o[2] for o in ops], np.array([dt(float(o[3])) for o in ops], dtype=dt),
            [1 if o[0] == "set" else 0 for o in ops])


@pytest.mark.parametrize("case", [c for c in CASES if "ops" in c and "SparseMatIndexList" in c["container"]
                                  and "Par" not in c["container"]], ids=lambda c: c["name"])
def test_c_assembly_oracle_reproduces_reference_crs(case):
    """orc_assemble (the C restatement of the add_to/set stream + to_crs) against the CRS arrays the
    reference's own tests pin (src/lib.rs:54-98, 36-52), bit for bit."""
    dt = np.float32 if case["dtype"] == "f32" else np.float64
    rows, cols, vals, ops = _ops_arrays(case, dt)
    n_rows, n_cols, off, col, val = oracle.assemble(rows, cols, vals, ops)
    crs = case["crs"]
    assert (n_rows, n_cols) == (crs["n_rows"], crs["n_cols"])
    assert list(off) == crs["offset_rows"] and list(col) == crs["columns"]
    assert val.tobytes() == _values(case).tobytes()


def test_c_crs_replay_oracle_reproduces_reference_crs_and_container_restatement():
    """orc_crs_replay (the stream applied to a SparseMatCRS itself: push prepends, first-push quirk) against the
    arrays src/lib.rs:114-154 pins, and against the step-by-step Python container (CrsPushMatrix) on random streams,
    orphaned first entries included; transpose / column_info restatements against a dense walk."""
    case = [c for c in CASES if c["name"] == "check_sparsemat_crs"][0]
    rows, cols, vals, ops = _ops_arrays(case, np.float32)
    n_rows, n_cols, off, col, val, stored = oracle.crs_replay(rows, cols, vals, ops)
    crs = case["crs"]
    assert (n_rows, n_cols, stored) == (crs["n_rows"], crs["n_cols"], len(crs["columns"]))
    assert list(off) == crs["offset_rows"] and list(col) == crs["columns"]
    assert val.tobytes() == _values(case).tobytes()
    rng = np.random.default_rng(6)
    orphans = 0
    for t in range(300):
        dt = np.float32 if t % 2 else np.float64
        n, n_r, n_c = int(rng.integers(0, 60)), int(rng.integers(1, 8)), int(rng.integers(1, 6))
        rows, cols = rng.integers(0, n_r, n), rng.integers(0, n_c, n)
        vals = rng.uniform(-1, 1, n).astype(dt)
        vals[rng.random(n) < 0.1] = dt(-0.0)
        ops = rng.integers(0, 2, n) if t % 3 else None
        m = assembly.CrsPushMatrix(dt)
        for k in range(n):
            (m.set if ops is not None and ops[k] else m.add_to)(int(rows[k]), int(cols[k]), vals[k])
        n_rows, n_cols, off, col, val, stored = oracle.crs_replay(rows, cols, vals, ops)
        e_rows, e_cols, e_off, e_col, e_val = m.to_crs_arrays()
        reach = int(e_off[-1]) if e_rows else 0
        assert (n_rows, n_cols, stored) == (e_rows, e_cols, len(e_col))
        assert np.array_equal(off[:n_rows + 1] if n_rows else off[:0], e_off[:n_rows + 1] if n_rows else e_off[:0])
        assert np.array_equal(col, e_col[:reach]) and val.tobytes() == e_val[:reach].tobytes()
        orphans += stored - reach
    assert orphans > 0  # the quirk was exercised
    # transpose = the set(j, i, val) replay; column_info = per-column entry lists in storage order
    off = np.array([0, 2, 2, 5], np.uint32)
    col = np.array([1, 3, 0, 3, 1], np.uint32)
    val = np.array([1.0, 2.0, 3.0, 4.0, 5.0], np.float32)
    t_rows, t_cols, t_off, t_col, t_val, t_stored = oracle.transpose(off, col, val)
    assert (t_rows, t_cols, t_stored) == (4, 3, 5)
    assert list(t_off) == [0, 1, 3, 3, 5] and list(t_col) == [2, 2, 0, 2, 0] and list(t_val) == [3.0, 5.0, 1.0, 4.0, 2.0]
    rows, col_ptr, entries = oracle.column_info(off, col, 4)
    assert list(rows) == [0, 0, 2, 2, 2] and list(col_ptr) == [0, 1, 3, 3, 5] and list(entries) == [2, 0, 4, 1, 3]


def _crs_of(case):
    crs = case["crs"]
    return (crs["n_rows"], crs["n_cols"], np.array(crs["offset_rows"], np.uint32), np.array(crs["columns"], np.uint32),
            _values(case))


def test_prod_and_column_iterator_oracles_reproduce_reference_asserts():
    """The reference's own assertions on the widened rows: `sp_crs.prod(&sp).unwrap().get(1, 2) == 17.9632`
    (src/lib.rs:99-101) and the column iterators (src/lib.rs:85-90 index list: insertion order; :137-142 CRS: storage
    order)."""
    case = [c for c in CASES if c["name"] == "check_sparsemat_indexlist"][0]
    a = _crs_of(case)
    n_rows, n_cols, off, col, val, _ = oracle.prod(a, a)
    for i, j, lit in case["expect_prod_self"]:
        got = [val[q] for q in range(off[i], off[i + 1]) if col[q] == j]
        assert got == [np.float32(float(lit))]
    # the index-list matrix lists a column in insertion order: first appearance of (row, j) in the call sequence
    for j, want in case["expect_iter_col_insertion_order"]:
        seen = []
        for _, r, c, _v in case["ops"]:
            if c == j and r not in seen:
                seen.append(r)
        assert seen == [w[0] for w in want]
        for r, lit in want:
            q = [q for q in range(a[2][r], a[2][r + 1]) if a[3][q] == j][0]
            assert a[4][q] == np.float32(float(lit))
    case = [c for c in CASES if c["name"] == "check_sparsemat_crs"][0]
    n_rows, n_cols, off, col, val = _crs_of(case)
    rows, col_ptr, entries = oracle.column_info(off, col, n_cols)
    for j, want in case["expect_iter_col"]:
        got = [(int(rows[e]), val[e]) for e in entries[col_ptr[j]:col_ptr[j + 1]]]
        assert got == [(r, np.float32(float(lit))) for r, lit in want]
    num, den = case["expect_density"]
    assert len(col) / (n_rows * n_cols) == num / den  # density(), sparsematrix.rs:237-241


def test_c_prod_oracle_matches_the_loops_written_out():
    """orc_crs_prod_ops + orc_crs_replay against SparseMatrix::prod (sparsematrix.rs:186-210) written out in plain
    Python over the step-by-step CRS container: column lists of rhs in storage order, the row stably sorted,
    take_while(col <= row), `sum += val * val_rhs` in the value type, `if sum != 0 { ret.set(i, j, sum) }`."""
    rng = np.random.default_rng(8)
    for t in range(25):
        dt = np.float32 if t % 2 else np.float64
        n, k = int(rng.integers(1, 9)), int(rng.integers(1, 9))

        def rand(n_rows, n_cols):
            lens = rng.integers(0, 5, n_rows)
            lens[-1] = max(lens[-1], 1)
            off = np.zeros(n_rows + 1, np.uint32)
            np.cumsum(lens, out=off[1:])
            col = rng.integers(0, n_cols, int(off[-1])).astype(np.uint32)  # unsorted, duplicates
            col[-1] = n_cols - 1
            val = rng.integers(-2, 3, len(col)).astype(dt) if t % 3 == 0 else rng.uniform(-1, 1, len(col)).astype(dt)
            return n_rows, n_cols, off, col, val
        a, b = rand(n, k), rand(k, n)
        # rhs.assemble_column_info(): per column the (row, value) pairs in storage order
        b_cols = [[] for _ in range(b[1])]
        for r in range(b[0]):
            for q in range(b[2][r], b[2][r + 1]):
                b_cols[b[3][q]].append((r, b[4][q]))
        ret = assembly.CrsPushMatrix(dt)
        for i in range(a[0]):
            row = sorted(((int(a[3][q]), a[4][q]) for q in range(a[2][i], a[2][i + 1])), key=lambda cv: cv[0])  # stable
            for j in range(b[1]):
                s = dt(0)
                for r, v_rhs in b_cols[j]:
                    for c, v in row:
                        if c > r:
                            break
                        if c == r:
                            s = dt(s + dt(v * v_rhs))
                if s != 0:
                    ret.set(i, j, s)
        e_rows, e_cols, e_off, e_col, e_val = ret.to_crs_arrays()
        reach = int(e_off[-1]) if e_rows else 0
        n_rows, n_cols, off, col, val, stored = oracle.prod(a, b)
        assert (n_rows, n_cols, stored) == (e_rows, e_cols, len(e_col))
        assert np.array_equal(off[:n_rows + 1] if n_rows else off[:0], e_off[:n_rows + 1] if n_rows else e_off[:0])
        assert np.array_equal(col, e_col[:reach]) and val.tobytes() == e_val[:reach].tobytes()
    with pytest.raises(oracle.OraclePanic):  # Err("Dimension mismatch"), :188-190
        oracle.prod(a, (b[0], b[1] + 1, b[2], b[3], b[4]))


def test_c_assembly_oracle_matches_container_restatement():
    """Random add_to/set streams: the C oracle equals the step-by-step Python container (IndexListMatrix),
    incl. gaps (empty rows), -0.0 values and the empty stream; sort_rows equals a stable numpy sort."""
    rng = np.random.default_rng(5)
    for t in range(60):
        dt = np.float32 if t % 2 else np.float64
        n, n_r, n_c = int(rng.integers(0, 300)), int(rng.integers(1, 15)), int(rng.integers(1, 10))
        rows, cols = rng.integers(0, n_r, n), rng.integers(0, n_c, n)
        vals = rng.uniform(-1, 1, n).astype(dt)
        vals[rng.random(n) < 0.1] = dt(-0.0)
        ops = rng.integers(0, 2, n) if t % 3 else None
        m = assembly.IndexListMatrix(dt)
        for k in range(n):
            (m.set if ops is not None and ops[k] else m.add_to)(int(rows[k]), int(cols[k]), vals[k])
        n_rows, n_cols, off, col, val = oracle.assemble(rows, cols, vals, ops)
        if n == 0:
            assert (n_rows, n_cols, len(col)) == (0, 0, 0)
            continue
        e_rows, e_cols, e_off, e_col, e_val = m.to_crs_arrays()
        assert (n_rows, n_cols) == (e_rows, e_cols)
        assert np.array_equal(off, e_off) and np.array_equal(col, e_col)
        assert val.tobytes() == e_val.tobytes()
        s_col, s_val = oracle.crs_sort_rows(off, col, val)
        for i in range(n_rows):
            a, b = off[i], off[i + 1]
            order = np.argsort(col[a:b], kind="stable")
            assert np.array_equal(s_col[a:b], col[a:b][order])
            assert s_val[a:b].tobytes() == val[a:b][order].tobytes()


def test_oracle_cg_reproduces_reference_assert():
    """check_cg (src/lib.rs:36-52)."""
    case = [c for c in CASES if c["name"] == "check_cg"][0]
    crs = case["crs"]
    b = np.array([float(s) for s in case["b"]])
    x0 = np.array([float(s) for s in case["x0"]])
    x, iters, rr = oracle.cg(crs["n_rows"], crs["n_cols"], crs["offset_rows"], crs["columns"], _values(case), b, x0,
                             tol=case["cg"]["tol"], iter_max=case["cg"]["iter_max"])
    for i, lit in case["expect_floor_1e4"]:
        assert np.floor(x[i] * 1e4) / 1e4 == float(lit)
    assert iters == 2 and np.sqrt(rr) < 1e-12
    np.testing.assert_allclose(x, [1 / 11, 7 / 11], rtol=1e-14)


def test_oracle_jacobi_pcg_extension_solves_and_reduces_to_cg():
    """orc_pcg_jacobi (an extension: the reference has no preconditioner): with diag(A) = 1 it is the reference's CG
    step for step (same iterates: z = r exactly); on a badly scaled SPD matrix it reaches scipy's solution in far fewer
    iterations than the plain recurrence."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    g = 8
    off, col, val = oracle.laplace3d(g, g, g, np.float64)
    n = g ** 3
    unit = val / 6.0
    b = np.linspace(-1, 1, n)
    x_p, it_p, _ = oracle.pcg_jacobi(n, n, off, col, unit, b, np.zeros(n), tol=1e-12, iter_max=1000)
    x_c, it_c, _ = oracle.cg(n, n, off, col, unit, b, np.zeros(n), tol=1e-12, iter_max=1000)
    assert it_p == it_c and np.array_equal(x_p, x_c)
    rng = np.random.default_rng(4)
    s = 10.0 ** rng.uniform(-1.5, 1.5, n)
    rows = np.repeat(np.arange(n), np.diff(off.astype(np.int64)))
    v = val * s[rows] * s[col]
    m = sp.csr_matrix((v, col, off), shape=(n, n))
    x_ref = spla.spsolve(m.tocsc(), b)
    tol = np.linalg.norm(b) * 1e-12
    x_p, it_p, _ = oracle.pcg_jacobi(n, n, off, col, v, b, np.zeros(n), tol=tol, iter_max=5000)
    _, it_c, _ = oracle.cg(n, n, off, col, v, b, np.zeros(n), tol=tol, iter_max=5000)
    assert np.max(np.abs(x_p - x_ref)) <= 1e-8 * np.max(np.abs(x_ref))
    assert it_p * 3 < it_c


def test_oracle_panics_mirror_the_reference():
    f = np.float64
    with pytest.raises(oracle.OraclePanic) as e:  # densevec.rs:41
        oracle.spmv([0, 1], [3], np.array([1.0]), np.ones(3))
    assert e.value.code == oracle.ORC_ERR_INDEX_OOB
    with pytest.raises(oracle.OraclePanic) as e:  # linearsolver.rs:30-32
        oracle.cg(2, 3, [0, 1, 2], [0, 2], np.array([1, 2], f), np.ones(2), np.zeros(2))
    assert str(e.value) == "Matrix is not symmetric"
    with pytest.raises(oracle.OraclePanic) as e:  # linearsolver.rs:33-36
        oracle.cg(2, 2, [0, 1, 2], [0, 1], np.array([1, 2], f), np.ones(3), np.zeros(2))
    assert str(e.value) == "Matrix and vector size mismatch"
    with pytest.raises(oracle.OraclePanic):  # densevec.rs:52-54
        oracle.vec_add(np.ones(2), np.ones(3))
    assert np.array_equal(oracle.vec_add(np.ones(3), np.ones(2)), [2, 2, 1])  # zip truncation


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_oracle_against_scipy(dtype):
    """Third-party sanity (tolerance only): scipy CSR product and CG on an SPD Laplacian."""
    rng = np.random.default_rng(1)
    n_rows, n_cols = 500, 400
    lens = rng.integers(0, 50, n_rows)
    off = np.zeros(n_rows + 1, np.uint32)
    np.cumsum(lens, out=off[1:])
    col = rng.integers(0, n_cols, off[-1]).astype(np.uint32)
    val = rng.uniform(-1, 1, off[-1]).astype(dtype)
    x = rng.uniform(-1, 1, n_cols).astype(dtype)
    a = sp.csr_matrix((val.astype(np.float64), col, off), shape=(n_rows, n_cols))
    np.testing.assert_allclose(oracle.spmv(off, col, val, x), a @ x.astype(np.float64), rtol=0,
                               atol=60 * np.finfo(dtype).eps * 50)
    off, col, val = oracle.laplace3d(6, 5, 4, np.float64)
    n = 120
    a = sp.csr_matrix((val, col, off), shape=(n, n))
    assert (abs(a - a.T)).nnz == 0
    b = rng.uniform(-1, 1, n)
    xs, info = spla.cg(a, b, rtol=1e-12, atol=0)
    xo, iters, rr = oracle.cg(n, n, off, col, val, b, np.zeros(n), tol=1e-11)
    assert info == 0 and iters < 200
    np.testing.assert_allclose(xo, xs, rtol=1e-8, atol=1e-10)


def test_oracle_blas1_and_reductions():
    rng = np.random.default_rng(2)
    a, b = rng.uniform(-1, 1, 1000).astype(np.float32), rng.uniform(-1, 1, 1000).astype(np.float32)
    s = np.float32(0)
    for u, v in zip(a, b):
        s = np.float32(s + np.float32(u * v))  # left fold, two roundings (vector.rs:50-53)
    assert oracle.dot(a, b) == s
    assert oracle.norm(a) == np.sqrt(np.float64(oracle.norm_squared(a)))
    al = np.float32(0.37)
    assert np.array_equal(oracle.vec_axpy(a, al, b), (a + (b * al)).astype(np.float32))
    assert np.array_equal(oracle.vec_xpby(a, al, b), ((a * al) + b).astype(np.float32))


def test_oracle_par_arithmetic_keeps_the_reference_clamp():
    """sparsemat_par.rs:21 (integer division) and :31-35 (clamp to n_blocks, the reference's off-by-one)."""
    assert oracle.par_rows_per_block(4, 16) == 4 and oracle.par_rows_per_block(4, 18) == 4
    assert oracle.par_block_and_row(4, 4, 0) == (0, 0)
    assert oracle.par_block_and_row(4, 4, 15) == (3, 3)
    assert oracle.par_block_and_row(4, 4, 17) == (4, 1)  # block id 4 of 4: out of bounds in the reference


def test_generators_are_deterministic_and_in_range():
    off, col, val = oracle.gen_fixed(1, oracle.PATTERN_BANDED, 10_000, 32, np.float32, 100, 400)
    assert off[0] == 0 and off[-1] == 300 * 32 and col.max() < 10_000
    c = col.reshape(-1, 32).astype(np.int64)
    assert np.all(np.diff(c, axis=1) > 0)
    rows = np.arange(100, 400)[:, None]
    assert np.all(np.abs(c - rows) <= 8192)
    off2, col2, val2 = oracle.gen_fixed(1, oracle.PATTERN_BANDED, 10_000, 32, np.float32, 100, 400)
    assert np.array_equal(col, col2) and np.array_equal(val, val2)
    assert val.min() >= -1 and val.max() < 1
    # SURVEY 8(d)'s "banded" to the letter: k distinct columns without replacement from [i - 4096, i + 4096], ascending
    for n, k, rb, re in ((50_000, 32, 0, 50_000), (50_000, 32, 45_000, 50_000), (40, 32, 0, 40), (9000, 64, 100, 1100)):
        off, col, val = oracle.gen_fixed(3, oracle.PATTERN_WINDOW, n, k, np.float64, rb, re)
        c = col.reshape(-1, k).astype(np.int64)
        rows = np.arange(rb, re)[:, None]
        assert off[-1] == (re - rb) * k and np.all(np.diff(c, axis=1) > 0) and c.min() >= 0 and c.max() < n
        assert np.all(np.abs(c - rows) <= 4096)
        sub = oracle.gen_fixed(3, oracle.PATTERN_WINDOW, n, k, np.float64, rb + 7, re - 3)  # a row's draw is its own
        assert np.array_equal(sub[1], col[7 * k:(re - rb - 3) * k]) and np.array_equal(sub[2], val[7 * k:(re - rb - 3) * k])
    off, col, val = oracle.gen_fixed(3, oracle.PATTERN_WINDOW, 2_000_000, 32, np.float32, 1_000_000, 1_020_000)
    d = (col.reshape(-1, 32).astype(np.int64) - np.arange(1_000_000, 1_020_000)[:, None]).ravel()
    hist = np.histogram(d, bins=16, range=(-4096, 4097))[0]
    assert hist.min() > 0.9 * hist.mean() and hist.max() < 1.1 * hist.mean()  # uniform over the window, not stratified
    assert np.diff(col.reshape(-1, 32), axis=1).max() > 1500                  # (strata of 256 columns bound a gap by 511)
    cdf = oracle.powerlaw_cdf()
    assert np.all(np.diff(cdf.astype(np.int64)) >= 0) and cdf[-1] == 0xFFFFFFFF
    off, col, val = oracle.gen_powerlaw(5, 20_000, 20_000)
    lens = np.diff(off.astype(np.int64))
    assert lens.min() >= 1 and lens.max() <= 2048 and 15 < lens.mean() < 60
    off, col, val = oracle.laplace2d(1000, 1000)
    assert len(val) == 4_996_000  # BASELINE C1: nnz of the 1000x1000 5-point Laplacian
