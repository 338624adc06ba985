#!/usr/bin/env python3
"""K1s with / without the LDS stage of x (SMH_STREAM_XS) on a grid whose planes are 1024 wide (x window 2818 entries: the
4096-entry stage), f32 and f64; bit-exactness against the SEQ kernel on all rows."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sparsemat_amd as sm
from sparsemat_amd import synth
L = sm.lib()
for dtype, dims in ((np.float32, (1024, 1024, 128)), (np.float64, (1024, 512, 128))):
    m = synth.crs_laplace3d(*dims, dtype)
    n = m.n_rows()
    xbuf, xptr = synth.gen_x(synth.SEED_X, n, dtype)
    ybuf = synth.DeviceBuffer(n * np.dtype(dtype).itemsize)
    zbuf = synth.DeviceBuffer(n * np.dtype(dtype).itemsize)
    for _ in range(3): m.mvp_dev(xptr, n, ybuf.ptr, "stream")
    L.smh_device_synchronize()
    t0 = time.perf_counter()
    for _ in range(20): m.mvp_dev(xptr, n, ybuf.ptr, "stream")
    L.smh_device_synchronize()
    t = (time.perf_counter() - t0) / 20
    m.mvp_dev(xptr, n, zbuf.ptr, "seq")
    L.smh_device_synchronize()
    same = np.array_equal(ybuf.download(dtype, n).view(np.uint8), zbuf.download(dtype, n).view(np.uint8))
    print("grid %s %s XS=%s: K1s %.3f ms per product, bit-exact vs SEQ: %s" % (dims, np.dtype(dtype).name, os.environ.get("SMH_STREAM_XS", "1"), t * 1e3, same), flush=True)
