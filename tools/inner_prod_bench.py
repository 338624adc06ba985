#!/usr/bin/env python3
"""SparseMatrix::inner_prod (lhs^T A rhs, sparsematrix.rs:161-171) next to the plain product on the same handle:
C2 banded (K1r: SpMV + dot) and the 512^3 Laplacian (K1s: the dot rides the SpMV epilogue -- no y, no second pass).
Both calls are the synchronous vector API (smh_crs_spmv_vec / smh_crs_inner_prod_vec), timed on the host."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsemat_amd as sm  # noqa: E402
from sparsemat_amd import synth  # noqa: E402


def timed(fn, reps=20):
    fn()
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3


for name, make in (("C2 banded 10M x 32 f32 (K1r)", lambda: synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, 10_000_000, 32, np.float32)),
                   ("C4 Laplacian 512^3 f32 (K1s)", lambda: synth.crs_laplace3d(512, 512, 512, np.float32))):
    m = make()
    n = m.n_rows()
    buf, ptr = synth.gen_x(synth.SEED_X, n, np.float32)
    x = sm.DenseVec.from_device_ptr(ptr, n, np.float32, keep=buf)
    buf2, ptr2 = synth.gen_x(synth.SEED_X + 1, n, np.float32)
    lhs = sm.DenseVec.from_device_ptr(ptr2, n, np.float32, keep=buf2)
    y = sm.DenseVec.zeros(n, np.float32)
    def spmv():
        sm._lib.check(sm.lib().smh_crs_spmv_vec(m._h, x._h, y._h, 0))
    t_spmv = timed(spmv)
    t_ip = timed(lambda: m.inner_prod(lhs, x))
    val = m.inner_prod(lhs, x)
    ref = float(np.dot(lhs.to_numpy().astype(np.float64), y.to_numpy().astype(np.float64)))
    print("%-32s %s: SpMV %.3f ms, inner_prod %.3f ms (%.3fx); value %.6g vs dot(lhs, A rhs) in f64 %.6g" % (
        name, m.resolved_variant(), t_spmv, t_ip, t_ip / t_spmv, val, ref), flush=True)
    del m, x, lhs, y
