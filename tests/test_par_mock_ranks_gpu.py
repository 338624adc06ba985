"""The one-process-per-GPU code of the library (csrc/par.hip with a communicator from smh_comm_create) with MORE THAN ONE RANK
on the one GPU of this build: RCCL refuses two ranks on one device, so the rank processes run with tests/mock_rccl (a stand-in
for the dozen RCCL entry points par.hip calls, bytes through POSIX shared memory, LD_PRELOADed -- test infrastructure, see its
header).  What runs is the library's own rank logic, unchanged: the plan table all-gather of smh_par_create_rank, the window
exchange's send / receive pairing, the in-place all-gather with a ragged last block, the CG folds -- checked against the oracle
on the GLOBAL matrix, bit for bit where the kernel is bit-exact.  bench.py's launcher path runs the same way with three ranks."""
import json
import os
import secrets
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mock_so(gpu):
    from util import build_mock_rccl
    return build_mock_rccl()


RANK = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import oracle
import sparsemat_amd as sm
from sparsemat_amd import _lib, synth

rank, world = int(os.environ["SMH_T_RANK"]), int(os.environ["SMH_T_WORLD"])
_lib.check(sm.lib().smh_set_device(0))
comm = sm.Comm(bytes.fromhex(os.environ["SMH_T_UID"]), world, rank)
assert (comm.size(), comm.rank()) == (world, rank)
comm.barrier()
assert comm.max(float(rank)) == float(world - 1)


def my_block(n, off, col, val, dtype):
    r = n // world
    r0, r1 = rank * r, (n if rank == world - 1 else (rank + 1) * r)   # sparsemat_par.rs:21; the last block takes the remainder
    o = off[r0:r1 + 1].astype(np.int64)
    blk = sm.SparseMatCRS.from_raw_parts(r1 - r0, n, (o - o[0]).astype(np.uint32), col[o[0]:o[-1]], val[o[0]:o[-1]])
    return blk, r0, r1


def check_products(n, off, col, val, dtype, want_mode, tag):
    blk, r0, r1 = my_block(n, off, col, val, dtype)
    par = sm.SparseMatParLocal.for_rank(comm, n, blk)
    assert par.backend() == "rccl" and par.n_local_blocks() == 1
    mode, _ = par.exchange_mode("auto")
    assert mode == want_mode, (tag, mode)
    x_host = oracle.gen_x(synth.SEED_X, n, dtype)
    y_ref = oracle.spmv(off, col, val, x_host)
    z_ref = oracle.spmv(off, col, val, y_ref)
    x, y, z = par.vec(host=x_host), par.vec(), par.vec()
    for exch in ("auto", "allgather") + (("window",) if want_mode == "window" else ()):
        par.mvp_dev(x, y, "stream", exch)      # y slices, then the exchange
        par.mvp_dev(y, z, "stream", exch)      # reads what the exchange delivered
        par.synchronize()
        got_y, got_z = y.download_block(0), z.download_block(0)
        assert got_y[r0:r1].tobytes() == y_ref[r0:r1].tobytes(), (tag, exch, "own slice of y")
        assert got_z[r0:r1].tobytes() == z_ref[r0:r1].tobytes(), (tag, exch, "z = A y from the exchanged y")
        if exch == "allgather" or mode == "allgather":
            assert got_y.tobytes() == y_ref.tobytes(), (tag, exch, "gathered y")
        elif blk.n_non_zero_entries():
            lo, hi = blk.col_range()
            assert got_y[lo:hi + 1].tobytes() == y_ref[lo:hi + 1].tobytes(), (tag, exch, "window of y")
    return par


rng = np.random.default_rng(5)
# 1. BASELINE C5's shape in small, ragged last block: window exchange with both neighbours
n = 50_003
check_products(n, *oracle.gen_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, n, 32, np.float32), np.float32, "window", "banded f32")
# 2. columns everywhere: the all-gather (in place at b R) + the ragged tail's broadcast
n = 9_001
check_products(n, *oracle.gen_fixed(synth.SEED_MATRIX, synth.PATTERN_UNIFORM, n, 9, np.float64), np.float64, "allgather", "uniform f64")
# 3. one-directional coupling (row i takes columns i - 700 .. i - 690 and itself): every rank receives from below only; rank 0 nothing
n = 12_000
lens = np.full(n, 12, np.int64)
off = np.zeros(n + 1, np.uint32); np.cumsum(lens, out=off[1:])
col = np.clip(np.arange(n)[:, None] - 700 + np.arange(12)[None, :], 0, None)
col[:, 11] = np.arange(n)
col = col.astype(np.uint32).ravel()
val = rng.uniform(-1, 1, len(col)).astype(np.float32)
check_products(n, off, col, val, np.float32, "window", "one-directional")
# 4. block diagonal: nobody needs anybody -- every rank's group is empty
n = 8_000
r = n // world
col = ((np.arange(n) // r).clip(0, world - 1) * r)[:, None] + rng.integers(0, r, (n, 5))
off = (np.arange(n + 1) * 5).astype(np.uint32)
val = rng.uniform(-1, 1, n * 5).astype(np.float64)
check_products(n, off, col.astype(np.uint32).ravel(), val, np.float64, "window", "block diagonal")
# 4b. a rank whose block holds no entry at all (it needs nothing; the others still read its slice of the vector, which is
#     zero there), random columns elsewhere
n = 6_000
r = n // world
lens = rng.integers(0, 7, n)
lens[r:2 * r if world > 2 else n] = 0          # rank 1's rows (the last rank's for two ranks) are empty
off = np.zeros(n + 1, np.uint32); np.cumsum(lens, out=off[1:])
col = rng.integers(0, n, int(off[-1]), dtype=np.uint32)
val = rng.uniform(-1, 1, len(col)).astype(np.float64)
check_products(n, off, col, val, np.float64, "allgather", "one empty block")
# 5. the solver: ConjugateGradient::solve with M = SparseMatPar, scalars folded across the ranks on the devices
for dtype, tol in ((np.float64, 1e-10), (np.float32, 1e-4)):
    g = 14
    off, col, val = oracle.laplace3d(g, g, g, dtype)
    n = g ** 3
    b_host = oracle.spmv(off, col, val, np.random.default_rng(11).uniform(-1, 1, n).astype(dtype))
    blk, r0, r1 = my_block(n, off, col, val, dtype)
    par = sm.SparseMatParLocal.for_rank(comm, n, blk)
    b, x = par.vec(host=b_host), par.vec()
    iters, rr = par.cg_solve_vec(b, x, tol=tol, iter_max=500, check_every=3)
    o_x, o_iters, o_rr = oracle.cg(n, n, off, col, val, b_host, np.zeros(n, dtype), tol=tol, iter_max=500)
    assert abs(iters - o_iters) <= 1 and np.sqrt(rr) < tol, (dtype, iters, o_iters, rr)
    got = x.download_block(0)
    assert np.max(np.abs(got[r0:r1].astype(np.float64) - o_x[r0:r1])) < 10 * tol
    all_iters = comm.max(float(iters))
    assert all_iters == iters   # every rank took the same decision
# 6. SURVEY 8e's option for skewed matrices: blocks of equal ENTRY counts (power-law row lengths; every rank passes its first row),
#    the plan, the exchanges (the all-gather becomes a group of broadcasts: unequal slices) and the solver's folds follow the split table
n = 40_000
lens = np.minimum(2000, ((rng.pareto(1.2, n) + 1) * (1 + 12 * (np.arange(n) / n) ** 3)).astype(np.int64))  # long rows crowd the end
off = np.zeros(n + 1, np.uint32); np.cumsum(lens, out=off[1:])
band = rng.integers(-600, 601, int(off[-1]))
col = np.clip(np.repeat(np.arange(n), lens) + band, 0, n - 1).astype(np.uint32)
val = rng.uniform(-1, 1, len(col)).astype(np.float64)
nnz = int(off[-1])
cut = [0] + [int(np.searchsorted(off, nnz * k // world, side="left")) for k in range(1, world)] + [n]
r0, r1 = cut[rank], cut[rank + 1]
o = off[r0:r1 + 1].astype(np.int64)
blk = sm.SparseMatCRS.from_raw_parts(r1 - r0, n, (o - o[0]).astype(np.uint32), col[o[0]:o[-1]], val[o[0]:o[-1]])
par = sm.SparseMatParLocal.for_rank(comm, n, blk, row_begin=r0)
assert par.split() == cut and par.get_block_and_row_id(r0) == (rank, 0)
assert max(cut[k + 1] - cut[k] for k in range(world)) > 1.2 * min(cut[k + 1] - cut[k] for k in range(world))  # rows differ, entries do not
x_host = oracle.gen_x(synth.SEED_X, n, np.float64)
y_ref = oracle.spmv(off, col, val, x_host)
x, y = par.vec(host=x_host), par.vec()
for exch in ("auto", "window", "allgather"):
    par.mvp_dev(x, y, "stream", exch)
    par.synchronize()
    got = y.download_block(0)
    assert got[r0:r1].tobytes() == y_ref[r0:r1].tobytes(), ("nnz split", exch)
    if exch == "allgather":
        assert got.tobytes() == y_ref.tobytes()
    else:
        lo, hi = blk.col_range()
        assert got[lo:hi + 1].tobytes() == y_ref[lo:hi + 1].tobytes(), ("nnz split window", exch)
comm.barrier()
print("RANK %%d OK" %% rank)
"""


@pytest.mark.parametrize("world", [2, 3, 4])
def test_ranks_on_one_device_through_the_mock(gpu, mock_so, world):
    uid = ("smh_mock_t_%s" % secrets.token_hex(8)).encode().ljust(128, b"\0").hex()
    procs = []
    for rank in range(world):
        env = dict(os.environ, LD_PRELOAD=mock_so, SMH_T_RANK=str(rank), SMH_T_WORLD=str(world), SMH_T_UID=uid)
        procs.append(subprocess.Popen([sys.executable, "-c", RANK % {"root": ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for rank, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("RANK %d OK" % rank) in out, "rank %d of %d:\n%s\n%s" % (rank, world, out[-1500:], err[-3000:])


def test_bench_under_the_launcher_with_three_ranks(gpu, mock_so):
    """bench.py exactly as the driver starts it for N > 1 (torch.distributed.run, one process per rank), three ranks sharing the
    device through the mock: the file rendezvous of the communicator id, smh_par_create_rank, the window exchange, the max over
    ranks, the exchange self-check and the all-gather leg -- the line says the exchange delivered."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1", "--master-port", "29643",
           os.path.join(ROOT, "bench.py"), "--gpus", "3", "--rows", "300000", "--steps", "3", "--warmup", "1"]
    env = dict(os.environ, LD_PRELOAD=mock_so, SMH_BENCH_SHARE_DEVICES="1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["config"]["exchange"] == "window" and d["config"]["exchange_backend"] == "rccl"
    assert d["config"]["launch"].startswith("torch.distributed.run")
    assert d["exchange_check"]["ok"] is True and d["exchange_check"]["rows_per_rank"] in (64, 128)
    assert d["allgather_leg"]["exchange_check"]["ok"] is True


BAD_SPLIT = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import sparsemat_amd as sm
from sparsemat_amd import _lib
rank, case = int(os.environ["SMH_T_RANK"]), os.environ["SMH_T_CASE"]
_lib.check(sm.lib().smh_set_device(0))
comm = sm.Comm(bytes.fromhex(os.environ["SMH_T_UID"]), 2, rank)
n = 20
# rank 0 / rank 1: (first row, rows) -- "gap": rows 10, 11 belong to nobody; "overlap": rows 8, 9 to both; "short": the reference
# partition (no row_begin) with a block of 9 instead of 10 rows on rank 1
first, rows = {"gap": ((0, 10), (12, 8)), "overlap": ((0, 10), (8, 12)), "short": ((None, 10), (None, 9))}[case][rank]
off = (np.arange(rows + 1) * 1).astype(np.uint32)
col = np.arange(rows, dtype=np.uint32)
blk = sm.SparseMatCRS.from_raw_parts(rows, n, off, col, np.ones(rows, np.float32))
try:
    if first is None:
        sm.SparseMatParLocal.for_rank(comm, n, blk)
    else:
        sm.SparseMatParLocal.for_rank(comm, n, blk, row_begin=first)
    print("RANK %%d CREATED" %% rank)
except sm.SparseMatPanic as e:
    print("RANK %%d REFUSED: %%s" %% (rank, e))
comm.barrier()   # both ranks are still in step: nobody was left inside the table's all-gather
print("RANK %%d DONE" %% rank)
"""


@pytest.mark.parametrize("case,needle", [("gap", "must tile the rows"), ("overlap", "must tile the rows"), ("short", "failed its local checks")])
def test_a_bad_partition_is_refused_on_every_rank_together(gpu, mock_so, case, needle):
    """smh_par_create_rank[_split]: a gap, an overlap, or one rank holding the wrong number of rows.  Round 3 let the rank that saw the
    problem return while its peer got a handle (and would have hung in its first collective), or return BEFORE the plan table's
    all-gather while its peer blocked inside it.  Now the local verdicts travel with the table and every rank validates the whole
    table: both refuse, both reach the barrier that follows."""
    uid = ("smh_mock_b_%s" % secrets.token_hex(8)).encode().ljust(128, b"\0").hex()
    procs = []
    for rank in range(2):
        env = dict(os.environ, LD_PRELOAD=mock_so, SMH_T_RANK=str(rank), SMH_T_UID=uid, SMH_T_CASE=case)
        procs.append(subprocess.Popen([sys.executable, "-c", BAD_SPLIT % {"root": ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=200) for p in procs]
    for rank, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("RANK %d DONE" % rank) in out, (rank, out[-800:], err[-2000:])
        assert ("RANK %d REFUSED" % rank) in out and "CREATED" not in out, (rank, out)
    assert any(needle in o[0] for o in outs), [o[0] for o in outs]


NEGATIVE = r"""
import ctypes as C, os, sys
sys.path.insert(0, %(root)r)
import sparsemat_amd as sm
from sparsemat_amd import _lib, synth
rank, case = int(os.environ["SMH_T_RANK"]), os.environ["SMH_T_CASE"]
_lib.check(sm.lib().smh_set_device(0))
mock = C.CDLL(os.environ["LD_PRELOAD"])
class Uid(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
uid = Uid()
uid.internal = bytes.fromhex(os.environ["SMH_T_UID"]).rstrip(b"\0")
mock.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
for f in (mock.ncclSend, mock.ncclRecv):
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
comm = C.c_void_p()
assert mock.ncclCommInitRank(C.byref(comm), 2, uid, rank) == 0
buf = synth.DeviceBuffer(4096)
F32 = 7
assert mock.ncclGroupStart() == 0
if case == "recv_without_send":
    if rank == 0:
        assert mock.ncclRecv(buf.ptr, 16, F32, 1, comm, None) == 0
else:  # counts disagree
    if rank == 0:
        assert mock.ncclRecv(buf.ptr, 16, F32, 1, comm, None) == 0
    else:
        assert mock.ncclSend(buf.ptr, 8, F32, 0, comm, None) == 0
rc = mock.ncclGroupEnd()
print("RANK %%d rc %%d" %% (rank, rc))
"""


@pytest.mark.parametrize("case,needle", [("recv_without_send", "without a matching send"), ("count_mismatch", "disagree on the count")])
def test_the_mock_fails_where_rccl_would_hang(gpu, mock_so, case, needle):
    """The checker checked: a receive that no peer sends to, and a send / receive pair with different counts, are reported by
    the stand-in (real RCCL would hang or corrupt) -- so a green run above means the library's pairing really matched."""
    uid = ("smh_mock_n_%s" % secrets.token_hex(8)).encode().ljust(128, b"\0").hex()
    procs = []
    for rank in range(2):
        env = dict(os.environ, LD_PRELOAD=mock_so, SMH_T_RANK=str(rank), SMH_T_UID=uid, SMH_T_CASE=case)
        procs.append(subprocess.Popen([sys.executable, "-c", NEGATIVE % {"root": ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=200) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    assert "RANK 0 rc 0" not in outs[0][0] and needle in outs[0][1], outs[0]
