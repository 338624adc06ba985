#!/usr/bin/env python3
"""K2t on C3 (and C2-uniform with CASES=uniform): products per tile (SMH_TILED_TILE) and parts per slice (SMH_TILED_PARTS), both read when
the copy is built / the product is launched, swept in one process on one box (development aid).  SETTINGS="TILE=60;TILE=64,PARTS=12;..." """
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
settings = os.environ.get("SETTINGS", "default;TILE=52;TILE=56;default;TILE=64;TILE=68;PARTS=8;PARTS=12;default").split(";")
for case in os.environ.get("CASES", "c3").split(","):
    dtype = np.float64 if case == "c3" else np.float32
    vs = np.dtype(dtype).itemsize
    xb, xp = synth.gen_x(synth.SEED_X, n, dtype)
    yb = synth.DeviceBuffer(n * vs)
    for st in settings:
        for k in ("SMH_TILED_TILE", "SMH_TILED_PARTS", "SMH_TILED_CAP"):
            os.environ.pop(k, None)
        if st != "default":
            for kv in st.split(","):
                k, v = kv.split("=")
                os.environ["SMH_TILED_" + k] = v
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
        print("%s %-18s: %d row blocks (largest %d rows): median %.4f ms min %.4f" % (case, st, lay["n_row_blocks"], lay["rows_per_block"], t["median"], t["min"]), flush=True)
        del m
