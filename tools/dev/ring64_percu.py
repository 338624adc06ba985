#!/usr/bin/env python3
"""K1r row ranges per CU (SMH_RING_BLOCKS_PER_CU, read when a matrix's plan is built) A/B inside ONE process, interleaved, f64 and f32 on
the headline shape (development aid)."""
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
for dtype in (np.float64, np.float32):
    vs = np.dtype(dtype).itemsize
    xb, xp = synth.gen_x(synth.SEED_X, n, dtype)
    yb = synth.DeviceBuffer(n * vs)
    for per_cu in (6, 8, 7, 10, 6, 8, 9, 8, 6, 12, 16):
        os.environ["SMH_RING_BLOCKS_PER_CU"] = str(per_cu)
        m = synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_WINDOW, n, 32, dtype)
        for _ in range(3):
            m.mvp_dev(xp, n, yb.ptr, "auto", stream=s.value)
        check(lib.smh_stream_synchronize(s))
        ev = Events(lib, check, 30)
        for i in range(30):
            ev.start(i, s.value)
            m.mvp_dev(xp, n, yb.ptr, "auto", stream=s.value)
            ev.stop(i, s.value)
        check(lib.smh_stream_synchronize(s))
        t = stats(ev.times_ms())
        print("%s per_cu %2d: median %.4f ms  min %.4f  max %.4f" % (np.dtype(dtype).name, per_cu, t["median"], t["min"], t["max"]), flush=True)
        del m
