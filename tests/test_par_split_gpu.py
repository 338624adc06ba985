"""nnz-balanced split points ("as an option for skewed matrices"; csrc/par.hip smh_par_create_split / _adopt_split):
blocks of equal ENTRY counts instead of the reference's equal row counts (sparsemat_par.rs:21).  The partition becomes a table of
row boundaries; block lookup, the window plan, the interior rows, both exchanges and the solver follow it -- results against the
oracle on the global matrix, bit for bit with the bit-exact kernel."""
import numpy as np
import pytest

import oracle
import sparsemat_amd as sm
from sparsemat_amd import sparsemat_par_local

pytestmark = pytest.mark.gpu


def skewed(rng, n, dtype, band):
    # long rows crowd the END of the matrix (i.i.d. lengths would give equal rows equal entries): equal row counts are badly balanced
    lens = np.minimum(3000, ((rng.pareto(1.1, n) + 1) * (1 + 12 * (np.arange(n) / n) ** 3)).astype(np.int64))
    off = np.zeros(n + 1, np.uint32)
    np.cumsum(lens, out=off[1:])
    col = np.clip(np.repeat(np.arange(n), lens) + rng.integers(-band, band + 1, int(off[-1])), 0, n - 1).astype(np.uint32)
    return off, col, rng.uniform(-1, 1, len(col)).astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("n_blocks", [2, 4, 7])
def test_nnz_split_products_lookup_and_plan(gpu, dtype, n_blocks):
    rng = np.random.default_rng(60 + n_blocks)
    n = 60_000
    off, col, val = skewed(rng, n, dtype, 900)
    nnz = int(off[-1])
    m = sm.SparseMatParLocal.with_sub_matrices(n_blocks, n, n, off, col, val, device_ids=[0] * n_blocks, split="nnz")
    cut = m.split()
    # block k starts at the first row whose entries begin at or beyond k nnz / n_blocks
    assert cut == [0] + [int(np.searchsorted(off, nnz * k // n_blocks, side="left")) for k in range(1, n_blocks)] + [n]
    per = [int(off[cut[k + 1]]) - int(off[cut[k]]) for k in range(n_blocks)]
    assert max(per) - min(per) <= 2 * 3000 and max(np.diff(cut)) > 1.5 * min(np.diff(cut))  # equal entries, unequal rows
    rows = sm.SparseMatParLocal.with_sub_matrices(n_blocks, n, n, off, col, val, device_ids=[0] * n_blocks)  # the reference's cut
    per_rows = [int(off[c1]) - int(off[c0]) for c0, c1 in zip(rows.split()[:-1], rows.split()[1:])]
    assert max(per) / (nnz / n_blocks) < max(per_rows) / (nnz / n_blocks)  # ... and better balanced than equal rows
    for k in range(n_blocks):
        assert m.block(k)[:2] == (cut[k], cut[k + 1])
        assert m.get_block_and_row_id(cut[k]) == (k, 0) and m.get_block_and_row_id(cut[k + 1] - 1) == (k, cut[k + 1] - 1 - cut[k])
    # the plan arithmetic of the library with the split table against a direct restatement
    needs = np.ones(n_blocks, np.uint8)
    lo = np.array([col[off[cut[k]]:off[cut[k + 1]]].min() for k in range(n_blocks)], np.uint32)
    hi = np.array([col[off[cut[k]]:off[cut[k + 1]]].max() for k in range(n_blocks)], np.uint32)
    for b in range(n_blocks):
        recv, send, mode, worst = sparsemat_par_local.plan(n_blocks, n, needs, lo, hi, b, split_rows=cut)
        for q in range(n_blocks):
            a, e = max(int(lo[b]), cut[q]), min(int(hi[b]) + 1, cut[q + 1])
            assert recv[q] == ((a, e) if a < e and q != b else (0, 0))
            a, e = max(int(lo[q]), cut[b]), min(int(hi[q]) + 1, cut[b + 1])
            assert send[q] == ((a, e) if a < e and q != b else (0, 0))
    x_host = rng.uniform(-1, 1, n).astype(dtype)
    want = oracle.spmv(off, col, val, x_host)
    want2 = oracle.spmv(off, col, val, want)
    for exch in ("auto", "window", "allgather"):
        for overlap in (True, False):
            m.set_overlap(overlap)
            x, y, z = m.vec(host=x_host), m.vec(), m.vec()
            m.mvp_dev(x, y, variant="stream", exchange=exch)
            m.mvp_dev(y, z, variant="stream", exchange=exch)
            m.synchronize()
            assert y.download().tobytes() == want.tobytes() and z.download().tobytes() == want2.tobytes(), (exch, overlap)
    assert m.mvp(x_host, variant="stream").tobytes() == want.tobytes()  # the host-vector product


def test_adopted_blocks_with_a_split_table_and_the_solver(gpu):
    g = 24
    off, col, val = oracle.laplace3d(g, g, g, np.float64)
    n = g ** 3
    cut = [0, 1000, 1001, 7000, n]  # arbitrary boundaries, a one-row block among them
    blocks = []
    for k in range(4):
        o = off[cut[k]:cut[k + 1] + 1].astype(np.int64)
        blocks.append(sm.SparseMatCRS.from_raw_parts(cut[k + 1] - cut[k], n, (o - o[0]).astype(np.uint32), col[o[0]:o[-1]], val[o[0]:o[-1]]))
    m = sm.SparseMatParLocal.adopt(blocks, n, split_rows=cut)
    assert m.split() == cut and m.get_block_and_row_id(1000) == (1, 0) and m.get_block_and_row_id(1001) == (2, 0) and m.get_block_and_row_id(n - 1) == (3, n - 1 - 7000)
    b_host = oracle.spmv(off, col, val, np.ones(n))
    b, x = m.vec(host=b_host), m.vec()
    iters, rr = m.cg_solve_vec(b, x, tol=1e-10, iter_max=500)
    o_x, o_iters, _ = oracle.cg(n, n, off, col, val, b_host, np.zeros(n), tol=1e-10, iter_max=500)
    assert abs(iters - o_iters) <= 1 and np.abs(x.download() - o_x).max() < 1e-9
    with pytest.raises(sm.SparseMatPanic):  # a table that does not end at n_rows
        sm.SparseMatParLocal.adopt(blocks, n, split_rows=[0, 1000, 1001, 7000, n - 1])
    with pytest.raises(sm.SparseMatPanic):  # blocks that do not have the table's row counts
        sm.SparseMatParLocal.adopt(blocks, n, split_rows=[0, 999, 1001, 7000, n])
