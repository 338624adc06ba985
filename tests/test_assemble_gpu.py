"""On-device assembly (smh_crs_assemble: add_to/set stream -> CRS) and sort_rows: BIT-EXACT
against the C restatement of the reference's containers (oracle.assemble, itself pinned to the reference's
own CRS fixtures in test_oracle_golden.py) -- structure and values -- plus the reference's KAT streams
themselves, and SpMV on the assembled matrix."""
import json
import os

import numpy as np
import pytest

import oracle
import sparsemat_amd as sm

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")
with open(GOLDEN) as f:
    CASES = [c for c in json.load(f)["cases"] if "ops" in c and "SparseMatIndexList" in c["container"]
             and "Par" not in c["container"]]


def same_crs(m, expect):
    n_rows, n_cols, off, col, val = expect
    assert (m.n_rows(), m.n_cols(), m.n_non_zero_entries()) == (n_rows, n_cols, len(col))
    g_off, g_col, g_val = m.raw_parts()
    assert np.array_equal(g_off[:n_rows + 1], off)
    assert np.array_equal(g_col, col)
    assert g_val.tobytes() == val.tobytes()  # bit for bit, signed zeros included


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_reference_kat_streams_on_device(gpu, case):
    """The call sequences of the reference's own tests (src/lib.rs:54-98, 36-52) give its CRS arrays."""
    dt = np.float32 if case["dtype"] == "f32" else np.float64
    ops = case["ops"]
    rows, cols = [o[1] for o in ops], [o[2] for o in ops]
    vals = np.array([dt(float(o[3])) for o in ops], dtype=dt)
    kinds = [1 if o[0] == "set" else 0 for o in ops]
    m = sm.SparseMatCRS.from_triplets(rows, cols, vals, kinds)
    crs = case["crs"]
    bits = np.uint32 if dt == np.float32 else np.uint64
    val = np.array([int(b, 16) for b in crs["values_bits"]], dtype=bits).view(dt)
    same_crs(m, (crs["n_rows"], crs["n_cols"], np.array(crs["offset_rows"], np.uint32),
                 np.array(crs["columns"], np.uint32), val))
    if "expect_mvp" in case:  # assert_eq!(mvp.get(0), 34.544): through the assembled device matrix
        x = np.array([dt(float(s)) for s in case["x"]], dtype=dt)
        y = m.mvp(x, variant="stream")
        for i, lit in case["expect_mvp"]:
            assert y[i] == dt(float(lit))


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("kind", ["add_only", "mixed", "heavy_duplicates", "gaps", "one_entry"])
def test_assemble_bit_exact(gpu, dtype, kind):
    rng = np.random.default_rng({"add_only": 1, "mixed": 2, "heavy_duplicates": 3, "gaps": 4, "one_entry": 5}[kind])
    n = 40_000
    if kind == "heavy_duplicates":
        n_r, n_c = 40, 25          # ~40 operations per entry
    elif kind == "gaps":
        n_r, n_c = 200_000, 5_000  # most rows never touched
    elif kind == "one_entry":
        n_r, n_c = 1, 1            # a single run of 40 000 operations
    else:
        n_r, n_c = 3_000, 2_000
    rows, cols = rng.integers(0, n_r, n), rng.integers(0, n_c, n)
    vals = rng.uniform(-1, 1, n).astype(dtype)
    vals[rng.random(n) < 0.02] = dtype(-0.0)
    ops = None if kind == "add_only" else (rng.random(n) < 0.3).astype(np.uint8)
    m = sm.SparseMatCRS.from_triplets(rows, cols, vals, ops)
    expect = oracle.assemble(rows, cols, vals, ops)
    same_crs(m, expect)
    # the assembled matrix feeds the hot path: bit-exact SpMV through K1s
    n_rows, n_cols, off, col, val = expect
    x = rng.uniform(-1, 1, n_cols).astype(dtype)
    y = m.mvp(x, variant="stream")
    assert y.tobytes() == oracle.spmv(off, col, val, x).tobytes()
    # sort_row on every row
    m.sort_rows()
    s_col, s_val = oracle.crs_sort_rows(off, col, val)
    same_crs(m, (n_rows, n_cols, off, s_col, s_val))


def test_sort_rows_is_stable_on_duplicate_columns(gpu):
    """A CRS adopted with duplicate columns inside a row (legal input): sort_by is stable."""
    rng = np.random.default_rng(9)
    n_rows, n_cols = 500, 12
    lens = rng.integers(0, 40, n_rows)
    off = np.zeros(n_rows + 1, np.uint32)
    np.cumsum(lens, out=off[1:])
    col = rng.integers(0, n_cols, int(off[-1]), dtype=np.uint32)
    val = rng.uniform(-1, 1, len(col)).astype(np.float32)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    m.sort_rows()
    s_col, s_val = oracle.crs_sort_rows(off, col, val)
    same_crs(m, (n_rows, n_cols, off, s_col, s_val))
    x = rng.uniform(-1, 1, n_cols).astype(np.float32)
    assert m.mvp(x, variant="stream").tobytes() == oracle.spmv(off, s_col, s_val, x).tobytes()


def test_assemble_edge_cases(gpu):
    f = np.float32
    m = sm.SparseMatCRS.from_triplets([], [], np.array([], f))
    assert (m.n_rows(), m.n_cols(), m.n_non_zero_entries()) == (0, 0, 0)  # SparseMatCRS::new()
    m = sm.SparseMatCRS.from_triplets([7], [3], np.array([2.5], f))
    same_crs(m, oracle.assemble([7], [3], np.array([2.5], f)))
    assert (m.n_rows(), m.n_cols()) == (8, 4)
    # 0 + (-0) = +0 for add_to, set keeps -0
    m = sm.SparseMatCRS.from_triplets([0, 0], [0, 1], np.array([-0.0, -0.0], f), [0, 1])
    v = m.raw_parts()[2]
    assert v.tobytes() == np.array([0.0, -0.0], f).tobytes()
    with pytest.raises(sm.SparseMatPanic):
        sm.SparseMatCRS.from_triplets([0, 1], [0], np.array([1.0, 2.0], f))


def test_assemble_fem_like_stream(gpu):
    """Element-by-element assembly of a 3-D hexahedral mesh (8 x 8 local matrices: every entry of the 27-point
    stencil receives up to 8 contributions) -- the call pattern the reference's add_to exists for."""
    g = 24
    nodes = np.arange((g + 1) ** 3).reshape(g + 1, g + 1, g + 1)
    corners = np.stack([nodes[dx:g + dx, dy:g + dy, dz:g + dz].ravel()
                        for dx in (0, 1) for dy in (0, 1) for dz in (0, 1)], axis=1)  # [cells, 8]
    rows = np.repeat(corners, 8, axis=1).ravel()
    cols = np.tile(corners, (1, 8)).ravel()
    rng = np.random.default_rng(3)
    local = rng.uniform(-1, 1, (8, 8)).astype(np.float64)
    vals = np.tile(local.ravel(), len(corners))
    vals = (vals * rng.uniform(0.5, 1.5, len(vals))).astype(np.float64)
    m = sm.SparseMatCRS.from_triplets(rows, cols, vals)
    expect = oracle.assemble(rows, cols, vals)
    same_crs(m, expect)
    assert m.n_rows() == (g + 1) ** 3 and m.max_row_len() == 27


def test_assemble_then_solve_on_device(gpu):
    """The pipeline the reference's containers exist for: add_to stream -> to_crs -> ConjugateGradient::solve,
    entirely behind the C ABI.  7-point Laplacian assembled edge by edge (every diagonal entry is the sum of up to
    6 contributions of 1.0: exact in any order), solved for x* = 1."""
    g, dtype = 20, np.float64
    idx = np.arange(g ** 3).reshape(g, g, g)
    pairs = []
    for axis in range(3):
        a = np.take(idx, np.arange(g - 1), axis=axis).ravel()
        b = np.take(idx, np.arange(1, g), axis=axis).ravel()
        pairs.append((a, b))
    a = np.concatenate([p[0] for p in pairs])
    b = np.concatenate([p[1] for p in pairs])
    # per edge (a,b): A[a,a] += 1, A[b,b] += 1, A[a,b] -= 1, A[b,a] -= 1 ; plus a unit mass on the diagonal (SPD)
    rows = np.concatenate([a, b, a, b, idx.ravel()])
    cols = np.concatenate([a, b, b, a, idx.ravel()])
    vals = np.concatenate([np.ones(len(a)), np.ones(len(a)), -np.ones(len(a)), -np.ones(len(a)), np.ones(g ** 3)]).astype(dtype)
    order = np.random.default_rng(2).permutation(len(vals))  # element loops do not come sorted
    rows, cols, vals = rows[order], cols[order], vals[order]
    m = sm.SparseMatCRS.from_triplets(rows, cols, vals)
    n_rows, n_cols, off, col, val = oracle.assemble(rows, cols, vals)
    same_crs(m, (n_rows, n_cols, off, col, val))
    n = g ** 3
    bvec = oracle.spmv(off, col, val, np.ones(n, dtype))
    x_ref, it_ref, _ = oracle.cg(n, n, off, col, val, bvec, np.zeros(n, dtype), tol=1e-10, iter_max=500)
    xd, bd = sm.DenseVec.from_vec(np.zeros(n, dtype)), sm.DenseVec.from_vec(bvec)
    cg = sm.ConjugateGradient(1e-10, 500)
    cg.solve(m, bd, xd)
    assert abs(cg.iterations - it_ref) <= 2
    np.testing.assert_allclose(xd.to_numpy(), np.ones(n), rtol=0, atol=1e-8)
    np.testing.assert_allclose(xd.to_numpy(), x_ref, rtol=0, atol=1e-8)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("shape", ["short_rows", "one_long_row", "heavy_duplicates", "gaps_and_plateaus"])
@pytest.mark.parametrize("into_crs", [False, True], ids=["to_crs", "replay"])
def test_streams_in_row_order_skip_the_sort_by_row(gpu, dtype, shape, into_crs):
    """A stream whose rows never decrease (a row-major assembly loop, the products of SparseMatrix::prod) is in the order the
    stable sort by row would give: the assembly skips that sort.  Same arrays as with the sort (SMH_ASSEMBLE_PRESORTED=0)
    and as the restatement of the reference's containers, on the row-wise and on the long-row route."""
    rng = np.random.default_rng({"short_rows": 21, "one_long_row": 22, "heavy_duplicates": 23, "gaps_and_plateaus": 24}[shape])
    n = 30_000
    if shape == "short_rows":
        rows, n_c = np.sort(rng.integers(0, 4_000, n)), 900
    elif shape == "one_long_row":          # beyond the row-wise replay: the segmented-sort route
        rows, n_c = np.sort(np.concatenate([rng.integers(0, 300, n - 9000), np.full(9000, 123)])), 5_000
    elif shape == "heavy_duplicates":
        rows, n_c = np.sort(rng.integers(0, 50, n)), 20
    else:
        rows, n_c = np.sort(rng.choice([3, 4, 4000, 70_000, 70_001, 150_000], n)), 60
    cols = rng.integers(0, n_c, n)
    vals = rng.uniform(-1, 1, n).astype(dtype)
    vals[rng.random(n) < 0.02] = dtype(-0.0)
    ops = (rng.random(n) < 0.3).astype(np.uint8)
    if into_crs and rows[1] == rows[0] and cols[1] == cols[0]:
        cols[1] = (cols[0] + 1) % n_c      # (keep the first-push twin quirk out of this test)
    expect = (oracle.crs_replay if into_crs else oracle.assemble)(rows, cols, vals, ops)[:5]
    got = sm.SparseMatCRS.from_triplets(rows, cols, vals, ops, into_crs=into_crs)
    same_crs(got, expect)
    os.environ["SMH_ASSEMBLE_PRESORTED"] = "0"
    try:
        sorted_route = sm.SparseMatCRS.from_triplets(rows, cols, vals, ops, into_crs=into_crs)
    finally:
        os.environ.pop("SMH_ASSEMBLE_PRESORTED", None)
    same_crs(sorted_route, expect)
    # ... and a stream that is NOT in row order still is sorted (one operation moved to the front)
    rows2, cols2, vals2, ops2 = np.roll(rows, 1), np.roll(cols, 1), np.roll(vals, 1), np.roll(ops, 1)
    if rows2[0] > rows2[1]:
        e2 = (oracle.crs_replay if into_crs else oracle.assemble)(rows2, cols2, vals2, ops2)[:5]
        same_crs(sm.SparseMatCRS.from_triplets(rows2, cols2, vals2, ops2, into_crs=into_crs), e2)
