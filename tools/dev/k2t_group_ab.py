#!/usr/bin/env python3
"""K2t: row blocks cut from product counts per 1 / 32 / 64 rows (SMH_TILED_GROUP, read at build), interleaved in one process, C3 and
C2-uniform (development aid)."""
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
for case in os.environ.get("CASES", "c3,uniform").split(","):
    dtype = np.float64 if case == "c3" else np.float32
    vs = np.dtype(dtype).itemsize
    xb, xp = synth.gen_x(synth.SEED_X, n, dtype)
    yb = synth.DeviceBuffer(n * vs)
    for group in [int(v) for v in os.environ.get("ROWGROUPS", "1,32,1,32,8,64,1,32").split(",")]:
        os.environ["SMH_TILED_GROUP"] = str(group)
        m = synth.crs_powerlaw(synth.SEED_MATRIX, n, n, dtype) if case == "c3" else synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_UNIFORM, n, 32, dtype)
        lay = m.tiled_layout()
        for _ in range(3):
            m.mvp_dev(xp, n, yb.ptr, "tiled", stream=s.value)
        check(lib.smh_stream_synchronize(s))
        ev = Events(lib, check, 20)
        for i in range(20):
            ev.start(i, s.value)
            m.mvp_dev(xp, n, yb.ptr, "tiled", stream=s.value)
            ev.stop(i, s.value)
        check(lib.smh_stream_synchronize(s))
        t = stats(ev.times_ms())
        print("%s group %3d: %d row blocks (largest %d rows): median %.4f ms min %.4f" % (case, group, lay["n_row_blocks"], lay["rows_per_block"], t["median"], t["min"]), flush=True)
        del m
