"""Handle lifetime: every lazily built workspace (merge table, ring plans incl. the wide and the banded ring, 16-bit
column arrays, K1s windows and codes, K2c blocked copy, assembly scratch) is released with its handle -- device
memory in use returns to where it started after many create / use / destroy rounds."""
import gc

import numpy as np
import pytest
import torch

import oracle
import sparsemat_amd as sm
from util import random_crs

pytestmark = pytest.mark.gpu


def _used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return total - free


def _used_without_the_pool():
    import ctypes as C
    sm._lib.check(sm.lib().smh_pool_trim())
    kept, live = C.c_size_t(), C.c_size_t()
    sm._lib.check(sm.lib().smh_pool_stats(C.byref(kept), C.byref(live)))
    assert kept.value == 0
    return _used(), live.value


def _exercise(rng, kind):
    if kind == 0:    # banded rows: ring + 16-bit columns
        n = 60_000
        lens = rng.integers(10, 40, n)
        off = np.zeros(n + 1, np.uint32)
        np.cumsum(lens, out=off[1:])
        centers = np.repeat(np.arange(n), lens)
        col = np.clip(centers + rng.integers(-2000, 2000, len(centers)), 0, n - 1).astype(np.uint32)
        val = rng.uniform(-1, 1, len(col)).astype(np.float32)
        n_cols = n
    elif kind == 1:  # stencil: K1s codes, banded ring when the vector family is forced
        g = (120, 110, 12)
        off, col, val = oracle.laplace3d(*g, np.float64)
        n = n_cols = g[0] * g[1] * g[2]
    else:            # random columns: merge, K2c with a small block width
        n, n_cols = 30_000, 50_000
        off, col, val = random_crs(rng, n, n_cols, rng.integers(0, 30, n), np.float32)
    m = sm.SparseMatCRS.from_raw_parts(n, n_cols, off, col, val)
    x = rng.uniform(-1, 1, n_cols).astype(val.dtype)
    m.set_colblock_shift(12)  # at most 39 column blocks for these sizes
    for variant in ("auto", "vector", "merge", "stream", "colblock", "colfused", "colsplit", "tiled", "seq"):
        m.mvp(x, variant=variant)
    m.set_vector_lanes(4)
    m.mvp(x, variant="vector")
    m.inner_prod(np.ones(n, val.dtype), x)
    m.sort_rows()
    m.mvp(x, variant="auto")
    t = sm.SparseMatCRS.from_triplets(rng.integers(0, 500, 5000), rng.integers(0, 400, 5000),
                                      rng.uniform(-1, 1, 5000).astype(np.float32))
    t.sort_rows()
    del m, t


def test_no_device_memory_is_left_behind(gpu):
    rng = np.random.default_rng(0)
    for kind in range(3):   # first round: one-time allocations (code objects, rocPRIM state, thread-local scratch)
        _exercise(rng, kind)
    gc.collect()
    # the library keeps the device memory it frees (csrc/pool.hip): what it KEEPS depends on what ran before in this process, so both
    # readings are taken with the pool returned to the runtime, and the bytes the library has handed out are compared exactly
    before, live_before = _used_without_the_pool()
    for rep in range(8):
        for kind in range(3):
            _exercise(rng, kind)
    gc.collect()
    after, live_after = _used_without_the_pool()
    assert live_after == live_before, "the library still holds %d bytes more than before" % (live_after - live_before)
    assert after - before < 8 << 20, "device memory in use grew by %.1f MiB over 24 create/use/destroy rounds" % ((after - before) / 2 ** 20)
