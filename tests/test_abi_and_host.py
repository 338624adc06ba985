"""CPU: the C-ABI library loads and exports every symbol include/sparsemat_hip.h declares, fails loudly
without a device (no CPU fallback), the product never touches oracle/, and the SparseMatPar host logic
(partition arithmetic, gather layout) incl. a world_size-2 gloo run."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import sparsemat_amd as sm
from sparsemat_amd import _lib
import par_reference as sparsemat_par

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sparsemat_hip.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = sm.lib()
    names = _declared_symbols()
    assert len(names) >= 50
    for name in names:
        assert hasattr(L, name), "libsparsemat_hip.so lacks %s" % name
        assert name in _lib.SIGNATURES, "python binding lacks %s" % name
    assert sorted(_lib.SIGNATURES) == names, "binding declares symbols the header does not"
    assert L.smh_abi_version() == 3
    out = subprocess.run(["nm", "-D", "--defined-only", sm.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (smh_[a-z0-9_]+)", out))
    assert exported == set(names), exported ^ set(names)


def test_library_is_built_for_gfx950_only():
    """Every device code object in the offload bundle targets gfx950 (rocPRIM's host-side architecture
    name table mentions other names as plain strings; those are not code objects)."""
    import re
    blob = open(sm.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}
    for other in (b"sm_80", b"sm_90", b"nvptx"):
        assert other not in blob


def _no_gpu():
    n = C.c_int(0)
    sm.lib().smh_device_count(C.byref(n))
    return n.value == 0


@pytest.mark.skipif(not _no_gpu(), reason="only meaningful on a box without a GPU")
def test_compute_fails_loudly_without_a_device():
    f = np.float32
    with pytest.raises(sm.SparseMatPanic) as e:
        sm.SparseMatCRS.from_raw_parts(1, 1, [0, 1], [0], np.array([1.0], f))
    assert e.value.status == _lib.SMH_ERR_NO_DEVICE and "no CPU fallback" in str(e.value)
    with pytest.raises(sm.SparseMatPanic):
        sm.DenseVec.from_vec(np.ones(3, f))
    # argument errors are reported before the device is needed, with the reference's wording
    h = C.c_void_p()
    rc = sm.lib().smh_crs_create(0, 1, 1, 0xFFFFFFFF, None, None, None, 1, C.byref(h))
    assert rc == _lib.SMH_ERR_CAPACITY and b"Maximum number of 4294967295 entries reached" in sm.lib().smh_last_error()
    assert sm.lib().smh_status_string(_lib.SMH_ERR_NOT_SQUARE) == b"Matrix is not symmetric"
    assert sm.lib().smh_status_string(_lib.SMH_ERR_DIM_MISMATCH) == b"Dimension mismatch"


def test_product_never_uses_the_oracle():
    """oracle/ is test infrastructure: nothing under sparsemat_amd/, include/, rust/ or tools/ may name it (the timing
    scripts that run the oracle as checker / CPU baseline live under tests/bench/)."""
    for base in ("sparsemat_amd", "include", "rust", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    text = open(os.path.join(dirpath, fn), errors="replace").read()
                    assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), fn
                    assert "liboracle" not in text, fn
                    assert not re.search(r"#\s*include\s*[<\"][^>\"]*oracle", text), fn


def test_product_package_has_one_partition_and_no_torch():
    """a8 / a9 exist once in the product: csrc/par.hip behind smh_par_* (sparsemat_amd.SparseMatPar = SparseMatParLocal).  Round 1's
    torch.distributed form lives under tests/ (par_reference.py: plan restatement + gloo harness); the package imports no torch."""
    assert sm.SparseMatPar is sm.SparseMatParLocal and not hasattr(sm, "sparsemat_par")
    for fn in os.listdir(os.path.join(ROOT, "sparsemat_amd")):
        if fn.endswith(".py"):
            text = open(os.path.join(ROOT, "sparsemat_amd", fn)).read()
            assert not re.search(r"^(import|from)\s+torch\b", text, flags=re.M), fn  # (DenseVec.from_torch borrows a tensor's memory, lazily)
    assert not os.path.exists(os.path.join(ROOT, "sparsemat_amd", "sparsemat_par.py"))


def test_laplace3d_closed_form_offsets_match_the_oracle():
    """Host part of the device generator (size query needs no GPU): prefix counts are bit-exact."""
    import oracle
    for dims in [(1, 1, 1), (2, 1, 1), (1, 3, 1), (1, 1, 4), (3, 4, 5), (7, 7, 7)]:
        off, col, val = oracle.laplace3d(*dims)
        n = dims[0] * dims[1] * dims[2]
        for rb in range(0, n + 1):
            assert sm.synth.laplace3d_nnz(*dims, 0, rb) == off[rb]
    assert sm.synth.laplace3d_nnz(512, 512, 512) == 937_951_232  # BASELINE C4


def test_powerlaw_host_generator_matches_the_oracle():
    import oracle
    off = sm.synth.powerlaw_offsets(sm.synth.SEED_MATRIX, 50_000)
    off_ref, _, _ = oracle.gen_powerlaw(sm.synth.SEED_MATRIX, 50_000, 50_000)
    assert np.array_equal(off, off_ref)


# ---- SparseMatPar host logic --------------------------------------------------------------------------
def test_partition_arithmetic():
    import oracle
    for n_blocks, n_rows in [(4, 16), (4, 18), (8, 80_000_000), (3, 10), (2, 5), (1, 7)]:
        r = sparsemat_par.rows_per_block(n_blocks, n_rows)
        assert r == oracle.par_rows_per_block(n_blocks, n_rows) == n_rows // n_blocks  # sparsemat_par.rs:21
        covered = []
        for b in range(n_blocks):
            lo, hi = sparsemat_par.block_range(n_blocks, n_rows, b)
            assert lo == b * r  # results are placed at b * R (:64)
            covered += list(range(lo, hi)) if n_rows < 100 else []
        if n_rows < 100:
            assert covered == list(range(n_rows))
            for row in range(n_rows):
                b, lr = sparsemat_par.block_and_row(n_blocks, n_rows, row)
                ob, olr = oracle.par_block_and_row(n_blocks, r, row)
                if ob < n_blocks:  # where the reference's clamp is in range the two agree
                    assert (b, lr) == (ob, olr)
                lo, hi = sparsemat_par.block_range(n_blocks, n_rows, b)
                assert lo <= row < hi and lr == row - lo


def test_split_crs_rebases_offsets_and_keeps_global_columns():
    import oracle
    rng = np.random.default_rng(0)
    n_rows, n_cols = 103, 77
    lens = rng.integers(0, 9, n_rows)
    off = np.zeros(n_rows + 1, np.uint32)
    np.cumsum(lens, out=off[1:])
    col = rng.integers(0, n_cols, off[-1]).astype(np.uint32)
    val = rng.uniform(-1, 1, off[-1]).astype(np.float32)
    x = rng.uniform(-1, 1, n_cols).astype(np.float32)
    y = oracle.spmv(off, col, val, x)
    parts = []
    for b in range(4):
        lo, lc, lv = sparsemat_par.split_crs(n_rows, off, col, val, 4, b)
        assert lo[0] == 0 and lo[-1] == len(lc) == len(lv)
        parts.append(oracle.spmv(lo, lc, lv, x))
    assert np.array_equal(np.concatenate(parts), y)  # bit-exact: row sums do not depend on the split


WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + '/tests')
import numpy as np, torch, torch.distributed as dist
import oracle
import par_reference as sparsemat_par
from par_reference import SparseMatPar

class CheckerBlock:  # test-only local product (the CPU oracle); the product's block is HipBlock
    def __init__(self, off, col, val):
        self.off, self.col, self.val, self.n_rows = off, col, val, len(off) - 1
    def col_range(self):
        return (int(self.col.min()), int(self.col.max()) + 1) if len(self.col) else (0, 0)
    def mvp_into(self, x, y):
        y.copy_(torch.from_numpy(oracle.spmv(self.off, self.col, self.val, x.numpy())))

def make(n_rows, banded, rng):
    lens = rng.integers(0, 12, n_rows)
    off = np.zeros(n_rows + 1, np.uint32); np.cumsum(lens, out=off[1:])
    if banded:  # columns within +-3 of the row: a block references its slice + a 3-entry halo
        rows = np.repeat(np.arange(n_rows), lens)
        col = np.clip(rows + rng.integers(-3, 4, off[-1]), 0, n_rows - 1).astype(np.uint32)
    else:
        col = rng.integers(0, n_rows, off[-1]).astype(np.uint32)
    val = rng.uniform(-0.3, 0.3, off[-1]).astype(np.float32)
    return off, col, val

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
for n_rows in (64 * world, 64 * world + 37):  # even split (in-place gather) and ragged last block
    for banded in (False, True):
        rng = np.random.default_rng(42)
        off, col, val = make(n_rows, banded, rng)
        x = rng.uniform(-1, 1, n_rows).astype(np.float32)
        lo, lc, lv = sparsemat_par.split_crs(n_rows, off, col, val, world, rank)
        par = SparseMatPar.with_sub_matrices(world, n_rows, n_rows, rank, CheckerBlock(lo, lc, lv))
        # (1) mvp: every rank ends with the full vector; iterate (the gathered vector feeds the next product)
        y_ref = oracle.spmv(off, col, val, x)
        xt = torch.from_numpy(x.copy())
        for it in range(3):
            yt = par.mvp(xt)
            assert yt.shape == (n_rows,)
            assert np.array_equal(yt.numpy(), y_ref), (rank, n_rows, it)
            xt = yt.clone(); y_ref = oracle.spmv(off, col, val, y_ref)
        # (2) mvp_window: own slice + the referenced interval only
        mode = par.setup_window_exchange(xt)
        assert mode == ("halo" if banded else "allgather"), (mode, banded)
        clo, chi = par.local.col_range()
        y_ref = oracle.spmv(off, col, val, x)
        xt = torch.from_numpy(x.copy())
        for it in range(3):
            yt = par.mvp_window(xt)
            got, want = yt.numpy(), y_ref
            assert np.array_equal(got[par.begin:par.end], want[par.begin:par.end]), (rank, n_rows, banded, it)
            assert np.array_equal(got[clo:chi], want[clo:chi]), ("window", rank, n_rows, banded, it)
            xt = yt.clone(); y_ref = oracle.spmv(off, col, val, y_ref)
        if banded:  # the plan really is neighbour-only
            _, send, recv, ssz, rsz = par._plan
            assert sum(rsz) <= 6 and all(sz == 0 for q, sz in enumerate(rsz) if abs(q - rank) > 1)
dist.barrier()
dist.destroy_process_group()
print("rank %%d ok" %% rank)
'''


@pytest.mark.parametrize("world", [2, 3])
def test_sparsemat_par_gloo(tmp_path, world):
    """N>1 path on CPU: `world` processes, gloo, local products by the checker block; the all-gather
    layout (even and ragged) and the referenced-window exchange (neighbour plan and its all-gather fallback)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29531 + world), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("rank %d ok" % r) in o, o


def test_library_plan_matches_the_python_plan():
    """smh_par_plan (the plan arithmetic csrc/par.hip exchanges by: pure host code, no device needed) against the Python
    restatement the gloo tests run on, over random partitions incl. ragged last blocks, blocks without entries and
    intervals that reach across several neighbours."""
    from sparsemat_amd import sparsemat_par_local
    rng = np.random.default_rng(7)
    for _ in range(200):
        n_blocks = int(rng.integers(1, 9))
        n_rows = int(rng.integers(n_blocks, 400))
        needs = rng.integers(0, 4, n_blocks) > 0
        lo = rng.integers(0, n_rows, n_blocks)
        hi = np.minimum(n_rows - 1, lo + rng.integers(0, n_rows, n_blocks))  # inclusive
        py_needs = [(int(lo[q]), int(hi[q]) + 1) if needs[q] else (0, 0) for q in range(n_blocks)]
        worst = 0
        for block in range(n_blocks):
            recv, send, mode, max_recv = sparsemat_par_local.plan(n_blocks, n_rows, needs, lo, hi, block)
            py_send, py_recv = sparsemat_par.exchange_plan(n_blocks, n_rows, py_needs, block)
            assert recv == py_recv and send == py_send, (n_blocks, n_rows, block)
            worst = max(worst, sum(b - a for a, b in py_recv))
            for q in range(n_blocks):  # what I send to q is what q expects from me
                assert send[q] == sparsemat_par_local.plan(n_blocks, n_rows, needs, lo, hi, q)[0][block]
        assert max_recv == worst
        assert mode == ("none" if n_blocks == 1 else ("window" if worst * 2 < n_rows else "allgather"))
    with pytest.raises(sm.SparseMatPanic):  # R == 0 (sparsemat_par.rs:21,32)
        sparsemat_par_local.plan(5, 4, [1] * 5, [0] * 5, [3] * 5, 0)


def test_library_plan_with_a_split_table():
    """smh_par_plan_split (pure host code): the window plan of a partition cut at arbitrary row boundaries -- SURVEY 8e's nnz-balanced
    option -- against a direct restatement: block b receives of block q's rows [split[q], split[q+1]) what its columns [lo, hi]
    reference, and sends what q's reference of its own; with the reference's boundaries it equals smh_par_plan."""
    from sparsemat_amd import sparsemat_par_local
    rng = np.random.default_rng(8)
    for _ in range(200):
        n_blocks = int(rng.integers(1, 9))
        n_rows = int(rng.integers(n_blocks, 500))
        cut = [0] + sorted(int(v) for v in rng.integers(0, n_rows + 1, n_blocks - 1)) + [n_rows]  # empty blocks allowed
        needs = rng.integers(0, 4, n_blocks) > 0
        lo = rng.integers(0, n_rows, n_blocks)
        hi = np.minimum(n_rows - 1, lo + rng.integers(0, n_rows, n_blocks))
        worst = 0
        for b in range(n_blocks):
            recv, send, mode, max_recv = sparsemat_par_local.plan(n_blocks, n_rows, needs, lo, hi, b, split_rows=cut)
            got = 0
            for q in range(n_blocks):
                a, e = max(int(lo[b]), cut[q]), min(int(hi[b]) + 1, cut[q + 1])
                want = (a, e) if (needs[b] and q != b and a < e) else (0, 0)
                assert recv[q] == want, (cut, b, q)
                got += want[1] - want[0]
                assert send[q] == sparsemat_par_local.plan(n_blocks, n_rows, needs, lo, hi, q, split_rows=cut)[0][b]
            worst = max(worst, got)
        assert max_recv == max(worst, max(sum(e - a for a, e in sparsemat_par_local.plan(n_blocks, n_rows, needs, lo, hi, q, split_rows=cut)[0])
                                            for q in range(n_blocks)))
        r = n_rows // n_blocks
        ref_cut = [k * r for k in range(n_blocks)] + [n_rows]
        for b in range(n_blocks):
            assert sparsemat_par_local.plan(n_blocks, n_rows, needs, lo, hi, b, split_rows=ref_cut) == sparsemat_par_local.plan(n_blocks, n_rows, needs, lo, hi, b)
    with pytest.raises(sm.SparseMatPanic):
        sparsemat_par_local.plan(3, 10, [1] * 3, [0] * 3, [9] * 3, 0, split_rows=[0, 5, 4, 10])  # a block ending before it begins


def test_exchange_plan_arithmetic():
    n_blocks, n_rows = 4, 103  # R = 25, last block 28 rows
    needs = [(0, 30), (20, 55), (45, 80), (70, 103)]
    for rank in range(n_blocks):
        send, recv = sparsemat_par.exchange_plan(n_blocks, n_rows, needs, rank)
        b, e = sparsemat_par.block_range(n_blocks, n_rows, rank)
        for q in range(n_blocks):
            sa, sb = send[q]
            qs, qr = sparsemat_par.exchange_plan(n_blocks, n_rows, needs, q)
            assert (sa, sb) == qr[rank]  # what I send to q is what q expects from me
            if sa < sb:
                assert b <= sa and sb <= e  # a piece of MY slice
        got = sorted((a, bb) for a, bb in recv if a < bb)
        covered = set(range(b, e))
        for a, bb in got:
            covered |= set(range(a, bb))
        assert set(range(*needs[rank])) <= covered  # slice + received pieces cover the referenced interval


WORKER_CG = r'''
import os, sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + '/tests')
import numpy as np, torch, torch.distributed as dist
import oracle
import par_reference as sparsemat_par
from par_reference import SparseMatPar, ParConjugateGradient

class CheckerBlock:
    def __init__(self, off, col, val):
        self.off, self.col, self.val, self.n_rows = off, col, val, len(off) - 1
    def col_range(self):
        return (int(self.col.min()), int(self.col.max()) + 1) if len(self.col) else (0, 0)
    def mvp_into(self, x, y):
        y.copy_(torch.from_numpy(oracle.spmv(self.off, self.col, self.val, x.numpy())))

class CheckerOps:  # test-only vector engine: the oracle's DenseVec arithmetic on CPU tensors
    def dot(self, x, y, out):
        out[0] = float(oracle.dot(x.numpy(), y.numpy()))
    def axpy(self, y, a, x):
        y.copy_(torch.from_numpy(oracle.vec_axpy(y.numpy(), a.numpy()[0], x.numpy())))
    def xpby(self, p, b, r):
        p.copy_(torch.from_numpy(oracle.vec_xpby(p.numpy(), b.numpy()[0], r.numpy())))

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
nx, ny, nz = 9, 8, 7
n = nx * ny * nz
off, col, val = oracle.laplace3d(nx, ny, nz, np.float64)
rng = np.random.default_rng(3)
b = rng.uniform(-1, 1, n)
x_ref, it_ref, rr_ref = oracle.cg(n, n, off, col, val, b, np.zeros(n), tol=1e-10, iter_max=400)
lo, lc, lv = sparsemat_par.split_crs(n, off, col, val, world, rank)
par = SparseMatPar.with_sub_matrices(world, n, n, rank, CheckerBlock(lo, lc, lv))
assert par.setup_window_exchange(torch.zeros(1, dtype=torch.float64)) == "halo"  # 7-point stencil: neighbours only
bl = torch.from_numpy(b[par.begin:par.end].copy())
xl = torch.zeros(par.end - par.begin, dtype=torch.float64)
cg = ParConjugateGradient(1e-10, 400, ops=CheckerOps())
cg.solve(par, bl, xl)
assert abs(cg.iterations - it_ref) <= 2, (cg.iterations, it_ref)
assert np.sqrt(cg.r_norm_squared) < 1e-10
np.testing.assert_allclose(xl.numpy(), x_ref[par.begin:par.end], rtol=0, atol=1e-8)
dist.barrier()
dist.destroy_process_group()
print("rank %%d ok" %% rank)
'''


@pytest.mark.parametrize("world", [2, 3])
def test_par_cg_gloo(tmp_path, world):
    """Row-partitioned CG on CPU: halo exchange of p + two scalar all-reduces per iteration (gloo), local
    arithmetic by the oracle; the result matches the oracle's single-process solve."""
    script = tmp_path / "worker_cg.py"
    script.write_text(WORKER_CG % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29551 + world), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("rank %d ok" % r) in o, o
