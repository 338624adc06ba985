#!/usr/bin/env python3
"""What a step costs ONE rank of a real job beyond its product (development aid; one GPU): a block of 10 M rows of the headline matrix
with a neighbour block on each side (SMH_RANK_LIKE_NEIGHBOUR rows, default 8192: all of them boundary; 1 000 000: neighbours with an interior of their own, as on a node), one process, window exchange -- the big block's product is
split into boundary rows + interior rows and exchanges its halos exactly as a rank between two neighbours would; the neighbours' own
work is negligible.  Prints the step time with the overlap on and off against the same 10 M rows as a matrix of their own."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sparsemat_amd as sm  # noqa: E402
from sparsemat_amd import _lib, synth  # noqa: E402
from bench import Events, stats  # noqa: E402

lib, check = sm.lib(), _lib.check
rows, tiny, k = 10_000_000, int(os.environ.get("SMH_RANK_LIKE_NEIGHBOUR", "8192")), 32
n = rows + 2 * tiny
cuts = [0, tiny, tiny + rows, n]
blocks = [synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_WINDOW, n, k, np.float32, cuts[b], cuts[b + 1]) for b in range(3)]
par = sm.SparseMatParLocal.adopt(blocks, n, split_rows=cuts)
x, y = par.vec(host=np.ones(n, np.float32)), par.vec()
steps = 50
trace = os.environ.get("SMH_RANK_LIKE_TRACE") == "1"  # (under rocprofv3 --kernel-trace: the overlapped form only, a few steps)
if trace:
    steps = 5
for on in ((True,) if trace else (True, False, True, False)):
    par.set_overlap(on)
    for _ in range(10):
        par.mvp_dev(x, y, variant="auto", exchange="window")
    par.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(steps):
        par.mvp_dev(x, y, variant="auto", exchange="window")
    par.synchronize()
    print("three blocks (%d | 10 M | %d rows), overlap %-5s: %.4f ms per step" % (tiny, tiny, on, (time.perf_counter() - t0) / steps * 1e3), flush=True)
if trace:
    sys.exit(0)
del par, blocks, x, y
rows = n  # (all three blocks' rows as one matrix)
m = synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_WINDOW, rows, k, np.float32)
xb, xp = synth.gen_x(synth.SEED_X, rows, np.float32)
yb = synth.DeviceBuffer(rows * 4)
s = C.c_void_p()
check(lib.smh_stream_create(C.byref(s)))
for _ in range(10):
    m.mvp_dev(xp, rows, yb.ptr, "auto", stream=s.value)
check(lib.smh_stream_synchronize(s))
import time
t0 = time.perf_counter()
for _ in range(steps):
    m.mvp_dev(xp, rows, yb.ptr, "auto", stream=s.value)
check(lib.smh_stream_synchronize(s))
print("all %d rows as ONE matrix: %.4f ms per product" % (rows, 0) if False else "all the rows as ONE matrix: %.4f ms per product" % ((time.perf_counter() - t0) / steps * 1e3), flush=True)
