#!/usr/bin/env python3
"""Where does staging x in LDS (K1s XS) start to pay?  7-point Laplacians of growing size, the stage forced on / off."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sparsemat_amd as sm
from sparsemat_amd import synth
L = sm.lib()
for dtype in (np.float32, np.float64):
    for g in (100, 128, 160, 200, 256, 320):
        m = synth.crs_laplace3d(g, g, g, dtype)
        n = m.n_rows()
        xbuf, xptr = synth.gen_x(synth.SEED_X, n, dtype)
        ybuf = synth.DeviceBuffer(n * np.dtype(dtype).itemsize)
        res = {}
        for mode in (0, 1):
            m.set_stream_xs(mode)
            for _ in range(5): m.mvp_dev(xptr, n, ybuf.ptr, "stream")
            L.smh_device_synchronize()
            reps = 40
            t0 = time.perf_counter()
            for _ in range(reps): m.mvp_dev(xptr, n, ybuf.ptr, "stream")
            L.smh_device_synchronize()
            res[mode] = (time.perf_counter() - t0) / reps * 1e3
        print("%s %d^3 (x = %.1f MB): gathers %.4f ms, x staged %.4f ms (%+.1f %%)" % (np.dtype(dtype).name, g, n * np.dtype(dtype).itemsize / 1e6,
              res[0], res[1], 100 * (res[1] / res[0] - 1)), flush=True)
