#!/usr/bin/env python3
"""BASELINE configs[0] (5-point Laplacian 1000 x 1000): ConjugateGradient::solve on the device against oracle.cg -- iteration
counts and distances for the tolerances tests/test_c1_gpu.py asserts with (development aid; prints, asserts nothing)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # noqa: E402
import sparsemat_amd as sm  # noqa: E402
from sparsemat_amd import synth  # noqa: E402

G = 1000
N = G * G
for dtype, tols in ((np.float32, (1e-1, 1e-2)), (np.float64, (1e-4,))):
    off, col, val = oracle.laplace2d(G, G, dtype)
    m = sm.SparseMatCRS.from_raw_parts(N, N, off, col, val)
    xstar = oracle.gen_x(synth.SEED_X, N, dtype)
    b = oracle.spmv(off, col, val, xstar)
    for tol in tols:
        t0 = time.time()
        x_ref, it_ref, rr_ref = oracle.cg(N, N, off, col, val, b, np.zeros(N, dtype), tol=tol, iter_max=5000)
        t_or = time.time() - t0
        for variant in ("auto", "seq", "vector"):
            x = np.zeros(N, dtype)
            cg = sm.ConjugateGradient(tol, 5000, variant=variant)
            t0 = time.time()
            cg.solve(m, b, x)
            print("%s tol %g %-6s: iterations device %d oracle %d | rnorm device %.4g oracle %.4g | max|x - x_oracle| %.4g  max|x - x*| %.4g (oracle %.4g) | %.2f s (oracle %.1f s)"
                  % (np.dtype(dtype).name, tol, variant, cg.iterations, it_ref, np.sqrt(cg.r_norm_squared), np.sqrt(rr_ref),
                     np.abs(x.astype(np.float64) - x_ref).max(), np.abs(x.astype(np.float64) - xstar).max(),
                     np.abs(x_ref.astype(np.float64) - xstar).max(), time.time() - t0, t_or), flush=True)
    x1 = np.zeros(N, dtype)
    sm.ConjugateGradient(1e-30, 1, variant="seq").solve(m, b, x1)
    x1_ref, _, _ = oracle.cg(N, N, off, col, val, b, np.zeros(N, dtype), tol=1e-30, iter_max=1)
    print("%s one iteration (seq): max rel diff %.3g, bit-identical %s" % (np.dtype(dtype).name, float(np.max(np.abs(x1 - x1_ref) / np.maximum(np.abs(x1_ref), 1e-300))),
                                                                          x1.tobytes() == x1_ref.tobytes()), flush=True)
