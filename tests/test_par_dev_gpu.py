"""The device-resident multi-GPU path behind the C ABI (smh_par_spmv_dev / smh_par_exchange / smh_par_cg_solve_vec /
smh_comm_*, csrc/par.hip): the reference's intended mvp_par (sparsemat_par.rs:37-68) -- every block
multiplies against the shared vector, results at b * R, ONE exchange -- with the exchange inside the library.

On the one-GPU box the blocks share device 0, so the PEER backend (pull kernel) carries the one-process tests; the RCCL
call sequence runs with a lone rank (ncclCommInitRank, in-place all-gather, empty send/receive group, 1-element gathers of
the CG folds; SMH_PAR_EXCHANGE_SINGLE=1).  The tests at the end need more than one device and run wherever there is one."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
import sparsemat_amd as sm
from sparsemat_amd import _lib, synth

from test_par_local_gpu import random_crs, sm_device_count

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def block_rows(n_blocks, n_rows, b):
    r = n_rows // n_blocks
    return b * r, (n_rows if b == n_blocks - 1 else (b + 1) * r)


def col_interval(off, col, r0, r1):
    c = col[off[r0]:off[r1]]
    return (int(c.min()), int(c.max()) + 1) if len(c) else (0, 0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("n_blocks", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("shape", ["banded", "scattered"])
def test_spmv_dev_then_exchange(gpu, dtype, n_blocks, shape):
    rng = np.random.default_rng(100 + n_blocks * 2 + (shape == "banded"))
    n = 5003  # not divisible by the block counts: the last block takes the remainder
    off, col, val = random_crs(rng, n, n, 20, dtype, band=300 if shape == "banded" else None)
    m = sm.SparseMatParLocal.with_sub_matrices(n_blocks, n, n, off, col, val, device_ids=[0] * n_blocks)
    assert m.backend() == "peer" and m.n_local_blocks() == n_blocks
    x_host = rng.uniform(-1, 1, n).astype(dtype)
    y_want = oracle.spmv(off, col, val, x_host)
    z_want = oracle.spmv(off, col, val, y_want)
    # what AUTO resolves to: the window when no block receives half of the vector or more
    worst = 0
    for b in range(n_blocks):
        r0, r1 = block_rows(n_blocks, n, b)
        lo, hi = col_interval(off, col, r0, r1)
        worst = max(worst, (hi - lo) - max(0, min(hi, r1) - max(lo, r0)))
    mode, max_recv = m.exchange_mode("auto")
    assert max_recv == worst and mode == ("none" if n_blocks == 1 else ("window" if worst * 2 < n else "allgather"))
    for exchange in ("allgather", "window", "auto"):
        x, y, z = m.vec(host=x_host), m.vec(), m.vec()
        m.mvp_dev(x, y, variant="stream", exchange=exchange)   # K1s: the reference's order of additions, bit-exact
        m.mvp_dev(y, z, variant="stream", exchange=exchange)   # the exchanged y IS the next x
        m.synchronize()
        assert y.download().tobytes() == y_want.tobytes() and z.download().tobytes() == z_want.tobytes()
        resolved = m.exchange_mode(exchange)[0]
        for b in range(n_blocks):
            got = y.download_block(b)
            if resolved == "allgather":  # every block holds the whole vector
                assert got.tobytes() == y_want.tobytes(), (exchange, b)
            elif resolved == "window":   # its own slice and everything its columns reference
                r0, r1 = block_rows(n_blocks, n, b)
                lo, hi = col_interval(off, col, r0, r1)
                assert got[r0:r1].tobytes() == y_want[r0:r1].tobytes() and got[lo:hi].tobytes() == y_want[lo:hi].tobytes()


def test_exchange_on_its_own_and_errors(gpu):
    rng = np.random.default_rng(5)
    n, nb = 1000, 4
    off, col, val = random_crs(rng, n, n, 8, np.float32, band=40)
    m = sm.SparseMatParLocal.with_sub_matrices(nb, n, n, off, col, val, device_ids=[0] * nb)
    v = m.vec(host=np.arange(n, dtype=np.float32))  # replicated: every owned slice is right
    for b in range(nb):  # spoil everything a block does not own, then gather it back
        r0, r1 = block_rows(nb, n, b)
        keep = np.full(n, -1, np.float32)
        keep[r0:r1] = np.arange(r0, r1)
        _lib.check(sm.lib().smh_dev_upload(v.ptr(b), keep.ctypes.data, keep.nbytes))
    m.exchange(v, "allgather")
    m.synchronize()
    for b in range(nb):
        assert np.array_equal(v.download_block(b), np.arange(n, dtype=np.float32))
    with pytest.raises(sm.SparseMatPanic) as e:   # a vector of another length cannot be exchanged by rows
        m.exchange(m.vec(n=n + 1), "allgather")
    assert e.value.status == _lib.SMH_ERR_DIM_MISMATCH
    with pytest.raises(sm.SparseMatPanic):        # x and y must differ
        m.mvp_dev(v, v)
    with pytest.raises(sm.SparseMatPanic) as e:   # densevec.rs:41: a column beyond x
        m.mvp_dev(m.vec(n=10), m.vec())
    assert e.value.status == _lib.SMH_ERR_INDEX_RANGE
    with pytest.raises(sm.SparseMatPanic):        # RCCL needs a device per block
        m.set_backend("rccl")
    other = sm.SparseMatParLocal.with_sub_matrices(2, n, n, off, col, val, device_ids=[0, 0])
    with pytest.raises(sm.SparseMatPanic):        # a vector of another partition
        m.mvp_dev(other.vec(), m.vec())
    rect = sm.SparseMatParLocal.with_sub_matrices(2, n, n + 7, off, col, val, device_ids=[0, 0])
    assert rect.exchange_mode("auto")[0] == "allgather"  # a window is addressed by column and owned by row: square only
    with pytest.raises(sm.SparseMatPanic):
        rect.exchange(rect.vec(), "window")


@pytest.mark.parametrize("n_blocks", [2, 4])
def test_adopted_device_born_blocks(gpu, n_blocks):
    """BASELINE C5's shape in small: blocks generated on the device (rows [b R, (b+1) R) of the banded-stratified matrix,
    global columns), adopted, multiplied, halo-exchanged; the oracle regenerates the matrix on the host."""
    n, k = 40_000, 32
    r = n // n_blocks
    blocks = [synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, n, k, np.float32, b * r, (b + 1) * r) for b in range(n_blocks)]
    m = sm.SparseMatParLocal.adopt(blocks, n)
    mode, max_recv = m.exchange_mode("auto")
    assert mode == "window" and 0 < max_recv <= 2 * 4096
    off, col, val = oracle.gen_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, n, k, np.float32)
    x_host = oracle.gen_x(synth.SEED_X, n, np.float32)
    x, y, z = m.vec(host=x_host), m.vec(), m.vec()
    m.mvp_dev(x, y, variant="stream")
    m.mvp_dev(y, z, variant="stream")
    m.synchronize()
    y_want = oracle.spmv(off, col, val, x_host)
    assert y.download().tobytes() == y_want.tobytes()
    assert z.download().tobytes() == oracle.spmv(off, col, val, y_want).tobytes()
    from util import assert_spmv_close
    m.mvp_dev(x, y)  # AUTO: the ring kernel per block
    m.synchronize()
    assert_spmv_close(y.download(), off, col, val, x_host, "adopted blocks, AUTO")
    with pytest.raises(sm.SparseMatPanic) as e:  # a block of the wrong height
        sm.SparseMatParLocal.adopt(blocks[:-1] + [blocks[0]], n + 5)
    assert e.value.status == _lib.SMH_ERR_DIM_MISMATCH


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("check_every", [1, 3, 16])
def test_cg_vec_device_scalars(gpu, dtype, check_every):
    """The solver's scalars never visit the host between polls: whatever the polling period, the iteration count and the
    iterate are those of the reference's loop (stop test before the beta update, linearsolver.rs:52-54)."""
    g, nb = 12, 3
    off, col, val = oracle.laplace3d(g, g, g, dtype)
    n = g ** 3
    rng = np.random.default_rng(11)
    b_host = oracle.spmv(off, col, val, rng.uniform(-1, 1, n).astype(dtype))
    tol = 1e-4 if dtype == np.float32 else 1e-10
    m = sm.SparseMatParLocal.with_sub_matrices(nb, n, n, off, col, val, device_ids=[0] * nb)
    b, x = m.vec(host=b_host), m.vec()
    iters, rr = m.cg_solve_vec(b, x, tol=tol, iter_max=400, check_every=check_every)
    o_x, o_iters, o_rr = oracle.cg(n, n, off, col, val, b_host, np.zeros(n, dtype), tol=tol, iter_max=400)
    assert abs(iters - o_iters) <= 1 and np.sqrt(rr) < tol
    assert np.max(np.abs(x.download().astype(np.float64) - o_x)) < 10 * tol
    # bitwise reproducible: the folds are fixed trees, whatever the polling period
    x2 = m.vec()
    it2, rr2 = m.cg_solve_vec(b, x2, tol=tol, iter_max=400, check_every=7)
    assert (it2, rr2) == (iters, rr) and x2.download().tobytes() == x.download().tobytes()
    for cap in (0, 1, 5):  # iter_max is honoured exactly
        x3 = m.vec()
        it3, _ = m.cg_solve_vec(b, x3, tol=tol, iter_max=cap, check_every=check_every)
        o_x3, o_it3, _ = oracle.cg(n, n, off, col, val, b_host, np.zeros(n, dtype), tol=tol, iter_max=cap)
        assert it3 == o_it3 == cap
        assert np.max(np.abs(x3.download().astype(np.float64) - o_x3)) < (1e-4 if dtype == np.float32 else 1e-11)


LONE_RANK = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import oracle
import sparsemat_amd as sm
from sparsemat_amd import _lib, synth
_lib.check(sm.lib().smh_set_device(0))
comm = sm.Comm(sm.Comm.unique_id(), 1, 0)          # ncclGetUniqueId + ncclCommInitRank
assert (comm.size(), comm.rank()) == (1, 0)
comm.barrier()
assert comm.max(3.25) == 3.25
n, k = 30_000, 32
blk = synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, n, k, np.float32)
m = sm.SparseMatParLocal.for_rank(comm, n, blk)
assert m.backend() == "rccl" and (m.n_blocks(), m.n_local_blocks()) == (1, 1)
off, col, val = oracle.gen_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, n, k, np.float32)
xh = oracle.gen_x(synth.SEED_X, n, np.float32)
want = oracle.spmv(off, col, val, xh)
for exchange in ("allgather", "window", "auto"):   # a lone rank through the RCCL calls (SMH_PAR_EXCHANGE_SINGLE=1)
    x, y = m.vec(host=xh), m.vec()
    m.mvp_dev(x, y, variant="stream", exchange=exchange)
    m.synchronize()
    assert y.download().tobytes() == want.tobytes(), exchange
g = 10
off, col, val = oracle.laplace3d(g, g, g, np.float64)
nn = g ** 3
lap = sm.SparseMatCRS.from_raw_parts(nn, nn, off, col, val)
mc = sm.SparseMatParLocal.for_rank(comm, nn, lap)
bh = oracle.spmv(off, col, val, np.ones(nn))
b, x = mc.vec(host=bh), mc.vec()
iters, rr = mc.cg_solve_vec(b, x, tol=1e-10, iter_max=300)   # the folds go through 1-element ncclAllGathers
ox, oit, _ = oracle.cg(nn, nn, off, col, val, bh, np.zeros(nn), tol=1e-10, iter_max=300)
assert abs(iters - oit) <= 1 and np.abs(x.download() - ox).max() < 1e-9
try:
    m.set_backend("peer")
    raise SystemExit("a per-rank handle accepted the PEER backend")
except sm.SparseMatPanic:
    pass
mc.close(); m.close(); comm.close()
print("lone rank ok", iters, oit)
"""


def test_lone_rank_runs_the_rccl_call_sequence(gpu, tmp_path):
    """One process per GPU with a world of one: smh_comm_* (ncclCommInitRank), smh_par_create_rank, and -- with the test
    knob -- the in-place ncclAllGather / the grouped send-receive / the 1-element gathers of the CG folds.  A separate
    process because the knob is read once."""
    script = tmp_path / "lone_rank.py"
    script.write_text(LONE_RANK % {"root": ROOT})
    env = dict(os.environ, SMH_PAR_EXCHANGE_SINGLE="1")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "lone rank ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ---- more than one device: runs on a multi-GPU node only ---------------------------------------------------------------
needs_two = pytest.mark.skipif(sm_device_count() < 2, reason="needs at least two GPUs (one-process blocks on distinct devices)")


@needs_two
@pytest.mark.parametrize("backend", ["rccl", "peer"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_blocks_on_distinct_devices(gpu, backend, dtype):
    """Peer access, cross-device events, ncclCommInitAll, both exchanges, the f64 ring kernel's 128 KiB LDS opt-in on
    every device (function attributes are per device), the CG folds across devices."""
    nd = min(sm_device_count(), 8)
    rng = np.random.default_rng(21)
    n = 60_011
    off, col, val = random_crs(rng, n, n, 40, dtype, band=2000)
    m = sm.SparseMatParLocal.with_sub_matrices(nd, n, n, off, col, val, device_ids=list(range(nd)))
    assert m.backend() == "rccl"  # AUTO: a device per block
    m.set_backend(backend)
    xh = rng.uniform(-1, 1, n).astype(dtype)
    want = oracle.spmv(off, col, val, xh)
    from util import assert_spmv_close
    for exchange in ("allgather", "window"):
        x, y, z = m.vec(host=xh), m.vec(), m.vec()
        m.mvp_dev(x, y, variant="stream", exchange=exchange)
        m.mvp_dev(y, z, variant="stream", exchange=exchange)
        m.synchronize()
        assert y.download().tobytes() == want.tobytes() and z.download().tobytes() == oracle.spmv(off, col, val, want).tobytes()
        if exchange == "allgather":
            for b in range(nd):
                assert y.download_block(b).tobytes() == want.tobytes()
        m.mvp_dev(x, y, variant="vector", exchange=exchange)  # K1r on every device (f64: dynamic LDS above 64 KiB)
        m.synchronize()
        assert_spmv_close(y.download(), off, col, val, xh, "distinct devices, K1r")
    g = 16
    off, col, val = oracle.laplace3d(g, g, g, dtype)
    nn = g ** 3
    mc = sm.SparseMatParLocal.with_sub_matrices(nd, nn, nn, off, col, val, device_ids=list(range(nd)))
    mc.set_backend(backend)
    bh = oracle.spmv(off, col, val, np.ones(nn, dtype))
    tol = 1e-4 if dtype == np.float32 else 1e-10
    xs = np.zeros(nn, dtype)
    iters, rr = mc.cg_solve(bh, xs, tol=tol, iter_max=500)
    ox, oit, _ = oracle.cg(nn, nn, off, col, val, bh, np.zeros(nn, dtype), tol=tol, iter_max=500)
    assert abs(iters - oit) <= 1 and np.abs(xs.astype(np.float64) - ox).max() < 10 * tol


def test_full_size_c5_eight_blocks_on_one_device(gpu):
    """BASELINE configs[4] at its size as far as one GPU goes: 80 M rows x 32 = 2.56e9 entries in 8 row blocks of 10 M
    (global columns, sparsemat_par.rs:20-35), all adopted on device 0 (PEER backend), x born on the device, y = A x, the
    window exchange, z = A y.  Checked: the plan (window, <= 2 x 4096 received entries per block), every block's copy of y
    on the column interval it references against the owners' slices (bit for bit: the exchange moves bits), y and z on row blocks
    that straddle every block boundary against the oracle (K1r: the north-star bound; z from the exchanged y)."""
    from util import assert_spmv_close
    nb, r, k = 8, 10_000_000, 32
    n = nb * r
    blocks = [synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, n, k, np.float32, b * r, (b + 1) * r) for b in range(nb)]
    m = sm.SparseMatParLocal.adopt(blocks, n)
    assert m.n_non_zero_entries() == n * k > 2 ** 31
    assert m.backend() == "peer"
    mode, max_recv = m.exchange_mode("auto")
    assert mode == "window" and 0 < max_recv <= 2 * 4096
    x, y, z = m.vec(), m.vec(), m.vec()
    m.synchronize()
    for b in range(nb):
        synth.gen_x(synth.SEED_X, n, np.float32, ptr=x.ptr(b))
    sm.lib().smh_device_synchronize()
    m.mvp_dev(x, y)      # AUTO kernel per block, then the exchange (AUTO: window)
    m.mvp_dev(y, z)
    m.synchronize()
    assert blocks[0].resolved_variant()[0] == "vector"
    x_host = x.download_block(0)
    y_host, z_host = y.download(), z.download()
    for b in range(nb):  # the exchange: block b's copy holds the owners' bits on the columns it references
        lo, hi = blocks[b].col_range()
        hi += 1
        assert max(0, b * r - 4096) <= lo <= max(0, b * r - 3840) and min(n, (b + 1) * r + 3840) <= hi <= min(n, (b + 1) * r + 4096)
        mine = y.download_block(b)
        assert np.array_equal(mine[lo:hi].view(np.uint32), y_host[lo:hi].view(np.uint32)), b
        del mine
    for edge in range(nb + 1):  # rows on both sides of every block boundary
        rb, re = max(0, edge * r - 1500), min(n, edge * r + 1500)
        off, col, val = oracle.gen_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, n, k, np.float32, rb, re)
        assert_spmv_close(y_host[rb:re], off, col, val, x_host, "C5 y rows %d.." % rb)
        assert_spmv_close(z_host[rb:re], off, col, val, y_host, "C5 z rows %d.." % rb)
    # the bit-exact kernel on the same partition agrees with the oracle on those rows bit for bit
    m.mvp_dev(x, y, variant="stream")
    m.synchronize()
    y_st = y.download()
    for edge in (0, 3, nb):
        rb, re = max(0, edge * r - 1500), min(n, edge * r + 1500)
        off, col, val = oracle.gen_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, n, k, np.float32, rb, re)
        assert np.array_equal(y_st[rb:re].view(np.uint32), oracle.spmv(off, col, val, x_host).view(np.uint32))
    assert np.abs(y_st.astype(np.float64) - y_host).max() < 5e-5


@needs_two
@pytest.mark.parametrize("backend", ["rccl", "peer"])
def test_distinct_devices_ragged_split_overlap_and_threads(gpu, backend):
    nd = min(sm_device_count(), 8)
    _ragged_split_overlap_and_threads(nd, list(range(nd)), backend)


def test_ragged_split_overlap_and_threads_on_one_device(gpu):
    """The body of the distinct-device test above with its blocks on ONE device (peer backend), so that what a first two-GPU box
    runs has itself been run."""
    _ragged_split_overlap_and_threads(3, [0, 0, 0], "peer")


def _ragged_split_overlap_and_threads(nd, device_ids, backend):
    """Everything else section 5 of DESIGN.md describes, so that the first box with two GPUs exercises it in one run: a split table
    of unequal blocks (nnz-balanced: the all-gather becomes a group of broadcasts), the window exchange beside the interior rows'
    product (cross-device events on the side streams) against the same step with the overlap off, the issuing thread per block
    (a device per thread) against one issuing thread, and the solver on top -- all bit for bit the same, the K1s products bit for
    bit the oracle's."""
    dtype = np.float64
    rng = np.random.default_rng(23)
    n = 90_000
    lens = np.minimum(1500, ((rng.pareto(1.3, n) + 1) * (1 + 10 * (np.arange(n) / n) ** 3)).astype(np.int64))  # long rows crowd the end
    off = np.zeros(n + 1, np.uint32)
    np.cumsum(lens, out=off[1:])
    col = np.clip(np.repeat(np.arange(n), lens) + rng.integers(-900, 901, int(off[-1])), 0, n - 1).astype(np.uint32)
    val = rng.uniform(-1, 1, len(col)).astype(dtype)
    m = sm.SparseMatParLocal.with_sub_matrices(nd, n, n, off, col, val, device_ids=device_ids, split="nnz")
    cut = m.split()
    assert max(np.diff(cut)) > 1.2 * min(np.diff(cut))  # ragged by construction
    m.set_backend(backend)
    xh = rng.uniform(-1, 1, n).astype(dtype)
    want = oracle.spmv(off, col, val, xh)
    want2 = oracle.spmv(off, col, val, want)
    res = {}
    for threads in (0, 1):
        m.set_threads(threads)
        for overlap in (True, False):
            m.set_overlap(overlap)
            for exchange in ("window", "allgather"):
                x, y, z = m.vec(host=xh), m.vec(), m.vec()
                m.mvp_dev(x, y, variant="stream", exchange=exchange)
                m.mvp_dev(y, z, variant="stream", exchange=exchange)
                m.synchronize()
                assert y.download().tobytes() == want.tobytes() and z.download().tobytes() == want2.tobytes(), (threads, overlap, exchange)
    g = 24
    off, col, val = oracle.laplace3d(g, g, g, dtype)
    nn = g ** 3
    mc = sm.SparseMatParLocal.with_sub_matrices(nd, nn, nn, off, col, val, device_ids=device_ids)
    mc.set_backend(backend)
    bh = oracle.spmv(off, col, val, np.ones(nn, dtype))
    for threads in (0, 1):
        mc.set_threads(threads)
        for overlap in (True, False):
            mc.set_overlap(overlap)
            b, xs = mc.vec(host=bh), mc.vec()
            res[(threads, overlap)] = (mc.cg_solve_vec(b, xs, tol=1e-10, iter_max=400, check_every=5), xs.download().tobytes())
    assert len(set(res.values())) == 1, "iterates differ between the schedules"
    ox, oit, _ = oracle.cg(nn, nn, off, col, val, bh, np.zeros(nn, dtype), tol=1e-10, iter_max=400)
    (iters, rr), xb = res[(0, True)]
    assert abs(iters - oit) <= 1 and np.abs(np.frombuffer(xb, dtype) - ox).max() < 1e-9
