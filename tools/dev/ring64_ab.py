#!/usr/bin/env python3
"""f64 on the headline shape (10 M rows x 32, window / stratified columns): K1r kernel time for the library SPARSEMAT_HIP_LIB names
and SMH_RING_BLOCKS_PER_CU (development aid)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sparsemat_amd as sm  # noqa: E402
from sparsemat_amd import _lib, synth  # noqa: E402
from bench import Events, stats  # noqa: E402

lib, check = sm.lib(), _lib.check
n = 10_000_000
s = C.c_void_p()
check(lib.smh_stream_create(C.byref(s)))
for name, pat, dtype in (("window64", synth.PATTERN_WINDOW, np.float64), ("banded64", synth.PATTERN_BANDED, np.float64), ("window32", synth.PATTERN_WINDOW, np.float32)):
    m = synth.crs_fixed(synth.SEED_MATRIX, pat, n, 32, dtype)
    vs = np.dtype(dtype).itemsize
    xb, xp = synth.gen_x(synth.SEED_X, n, dtype)
    yb = synth.DeviceBuffer(n * vs)
    for _ in range(3):
        m.mvp_dev(xp, n, yb.ptr, "auto", stream=s.value)
    check(lib.smh_stream_synchronize(s))
    ev = Events(lib, check, 20)
    for i in range(20):
        ev.start(i, s.value)
        m.mvp_dev(xp, n, yb.ptr, "auto", stream=s.value)
        ev.stop(i, s.value)
    check(lib.smh_stream_synchronize(s))
    t = stats(ev.times_ms())
    B = m.n_non_zero_entries() * (vs + 4) + (n + 1) * 4 + 2 * n * vs
    nb, frac, act, _, ph = m.ring_plan()
    print("%s lib=%s per_cu=%s: %s ring blocks %d phases %d: median %.4f ms min %.4f -> %.3f of 8 TB/s" % (
        name, os.path.basename(sm.LIB_PATH), os.environ.get("SMH_RING_BLOCKS_PER_CU", "default"), m.resolved_variant(), nb, len(ph), t["median"], t["min"],
        B / t["median"] / 1e6 / 8000), flush=True)
    del m, xb, yb
