#!/usr/bin/env python3
"""How long do hipMalloc / hipFree of GB-sized buffers take on this box, and how do repeated transpositions of C2 behave
call by call (a transposition allocates ~10 GB of scratch and 2.6 GB of result, and frees them)?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sparsemat_amd as sm
from sparsemat_amd import synth

L = sm.lib()
def now():
    L.smh_device_synchronize()
    return time.perf_counter()

for gb in (0.04, 1.28, 3.84):
    ta, tf = [], []
    for _ in range(12):
        t0 = now(); b = synth.DeviceBuffer(int(gb * 1e9)); t1 = now()
        L.smh_dev_memset(b.ptr, 0, b.nbytes, None) if hasattr(L, "smh_dev_memset") else None
        t2 = now(); del b; t3 = now()
        ta.append((t1 - t0) * 1e3); tf.append((t3 - t2) * 1e3)
    print("%.2f GB: alloc ms %s | free ms %s" % (gb, " ".join("%.2f" % t for t in ta), " ".join("%.2f" % t for t in tf)), flush=True)

a = synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, 10_000_000, 32, np.float32)
ts = []
for i in range(10):
    t0 = now(); t = a.transpose(); t1 = now(); del t; t2 = now()
    ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
print("transpose ms (call, then destroy of the result): " + " ".join("%.1f/%.1f" % p for p in ts), flush=True)
