#!/usr/bin/env python3
"""Where does staging x in LDS start to pay?  7-point / 5-point Laplacians of growing size through K1s with gathers from memory,
with x staged (XS, column codes) and as K1s XD (stage offsets), forced each way."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sparsemat_amd as sm
from sparsemat_amd import synth
L = sm.lib()
grids = [(g, g, g) for g in (64, 80, 100, 128, 160, 200, 256, 320)] + [(1000, 1000, 1), (2000, 2000, 1), (1024, 1024, 8), (1024, 64, 64)]
for dtype in (np.float32, np.float64):
    for g in grids:
        m = synth.crs_laplace3d(g[0], g[1], g[2], dtype)
        n = m.n_rows()
        xbuf, xptr = synth.gen_x(synth.SEED_X, n, dtype)
        ybuf = synth.DeviceBuffer(n * np.dtype(dtype).itemsize)
        res, lay = {}, {}
        for name, xs, direct in (("gathers", 0, 0), ("xs", 1, 0), ("xd", 1, 1)):
            m.set_stream_xs(xs)
            m.set_stream_direct(direct)
            lay[name] = (m.stream_layout()["xs_chunks"], m.stream_direct())
            for _ in range(5): m.mvp_dev(xptr, n, ybuf.ptr, "stream")
            L.smh_device_synchronize()
            reps = 60
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(reps): m.mvp_dev(xptr, n, ybuf.ptr, "stream")
                L.smh_device_synchronize()
                best = min(best, (time.perf_counter() - t0) / reps * 1e3)
            res[name] = best
        m.set_stream_xs(-1); m.set_stream_direct(-1)
        auto = (m.stream_layout()["xs_chunks"], m.stream_direct())
        print("%s %dx%dx%d (x = %.1f MB): gathers %.4f ms, xs%d %.4f ms (%+.1f %%), xd %.4f ms (%+.1f %%); automatic: xs%d direct=%s" % (
              np.dtype(dtype).name, g[0], g[1], g[2], n * np.dtype(dtype).itemsize / 1e6, res["gathers"], lay["xs"][0], res["xs"],
              100 * (res["xs"] / res["gathers"] - 1), res["xd"], 100 * (res["xd"] / res["gathers"] - 1), auto[0], auto[1]), flush=True)
