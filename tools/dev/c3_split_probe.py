#!/usr/bin/env python3
"""Experiment: BASELINE C3 split by row length -- long rows (>= L entries) and short rows as two matrices, each through K2c
with its own block width.  Hypothesis: the long-row part has almost no y / offset sweeps (few rows), the short part few
gathers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparsemat_amd as sm
from sparsemat_amd import synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from quick_bench import time_variant

n = 10_000_000
m = synth.crs_powerlaw(synth.SEED_MATRIX, n, n, np.float64)
off, col, val = m.raw_parts()
lens = np.diff(off.astype(np.int64))
xbuf, xptr = synth.gen_x(synth.SEED_X, n, np.float64)
ybuf = synth.DeviceBuffer(n * 8)
print("whole: K2c", time_variant(m, xptr, n, ybuf.ptr, "colblock", reps=8), flush=True)
del m
for L in (64, 128, 256, 512):
    is_long = lens >= L
    rows_long = np.nonzero(is_long)[0]
    ent_long = np.repeat(is_long, lens)
    off_l = np.zeros(len(rows_long) + 1, np.uint32); np.cumsum(lens[rows_long], out=off_l[1:])
    ml = sm.SparseMatCRS.from_raw_parts(len(rows_long), n, off_l, col[ent_long], val[ent_long])
    lens_s = np.where(is_long, 0, lens)
    off_s = np.zeros(n + 1, np.uint32); np.cumsum(lens_s, out=off_s[1:])
    ms = sm.SparseMatCRS.from_raw_parts(n, n, off_s, col[~ent_long], val[~ent_long])
    print("L=%d: long rows %d entries %d (%.1f%%) | short rows entries %d mean %.2f" % (L, len(rows_long), int(off_l[-1]), 100.0 * off_l[-1] / off[-1], int(off_s[-1]), off_s[-1] / n), flush=True)
    for shift in (17, 18, 19):
        ml.set_colblock_shift(shift)
        print("   long  K2c 2^%d: median %.3f ms" % (shift, time_variant(ml, xptr, n, ybuf.ptr, "colblock", reps=8)[0]), flush=True)
    for var in ("merge", "stream"):
        print("   long  %s: median %.3f ms" % (var, time_variant(ml, xptr, n, ybuf.ptr, var, reps=5)[0]), flush=True)
    for shift in (19, 20, 21):
        ms.set_colblock_shift(shift)
        print("   short K2c 2^%d: median %.3f ms" % (shift, time_variant(ms, xptr, n, ybuf.ptr, "colblock", reps=8)[0]), flush=True)
        ms.set_colblock_shift(shift)
        cf = ms.colfused(arrays=False)
        if cf["fits"]:
            print("   short K2f 2^%d: median %.3f ms" % (shift, time_variant(ms, xptr, n, ybuf.ptr, "colfused", reps=8)[0]), flush=True)
    for var in ("stream", "merge"):
        print("   short %s: median %.3f ms" % (var, time_variant(ms, xptr, n, ybuf.ptr, var, reps=5)[0]), flush=True)
    del ml, ms
