#!/usr/bin/env python3
"""Ad-hoc timing of SparseMatrix::prod (smh_crs_prod) on one GPU: A.A for the 7-point Laplacian g^3 (49 products per
row, 25-point result) and A.A^T-like products of the C2 banded generator at a reduced row length, both operands born in
HBM.  Prints ms, products/s, and -- on a bounded sample -- the rate of the reference algorithm restated in C
(oracle.prod: O(n_rows * n_cols) loop trips on one core) with a bit-exact comparison; scipy's SpGEMM time on the host
for orientation (third party, different summation order)."""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # noqa: E402  (checker / CPU baseline only)
import sparsemat_amd as sm  # noqa: E402
from sparsemat_amd import synth  # noqa: E402


def timed(fn, reps):
    ts = []
    for _ in range(reps + 1):
        sm.lib().smh_device_synchronize()
        t0 = time.perf_counter()
        out = fn()
        sm.lib().smh_device_synchronize()
        ts.append(time.perf_counter() - t0)
        shape = (out.n_rows(), out.n_cols(), out.n_non_zero_entries())
        del out
    ts = sorted(ts[1:])
    return ts[len(ts) // 2], ts[0], shape


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=192)
    ap.add_argument("--rows", type=int, default=4_000_000)
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--cpu-grid", type=int, default=24)
    ap.add_argument("--scipy", action="store_true")
    args = ap.parse_args()
    torch.cuda.init()
    g = args.grid
    cases = [("7-pt Laplacian %d^3 f64, A.A" % g, lambda: synth.crs_laplace3d(g, g, g, np.float64), 49),
             ("banded f32 %.0fM x %d, A.A" % (args.rows / 1e6, args.k),
              lambda: synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, args.rows, args.k, np.float32), args.k * args.k)]
    for name, make, per_row in cases:
        a = make()
        med, best, shape = timed(lambda: a.prod(a), args.reps)
        products = a.n_rows() * per_row
        print("%s: %d entries, ~%d products | prod median %.1f ms (min %.1f) = %.2f G products/s; result rows %d cols %d entries %d"
              % (name, a.n_non_zero_entries(), products, med * 1e3, best * 1e3, products / med / 1e9, *shape), flush=True)
        if args.scipy:
            off, col, val = a.raw_parts()
            m = sp.csr_matrix((val, col, off), shape=(a.n_rows(), a.n_cols()))
            t0 = time.perf_counter()
            r = m @ m
            print("   scipy (host, 1 core, for orientation): %.1f ms, %d entries" % ((time.perf_counter() - t0) * 1e3, r.nnz), flush=True)
        del a
    # the reference algorithm on one core, bounded sample, with parity
    cg = args.cpu_grid
    small = synth.crs_laplace3d(cg, cg, cg, np.float64)
    off, col, val = small.raw_parts()
    n = cg ** 3
    t0 = time.perf_counter()
    e = oracle.prod((n, n, off, col, val), (n, n, off, col, val))
    cpu = time.perf_counter() - t0
    c = small.prod(small)
    g_off, g_col, g_val = c.raw_parts()
    ok = (c.n_rows(), c.n_cols()) == (e[0], e[1]) and np.array_equal(g_off, e[2]) and np.array_equal(g_col, e[3]) \
        and g_val.tobytes() == e[4].tobytes()
    print("cpu oracle (reference algorithm, 1 core): %d^3 = %d rows in %.2f s = %.6f G products/s; device result bit-exact: %s"
          % (cg, n, cpu, n * 49 / cpu / 1e9, ok), flush=True)
    assert ok


if __name__ == "__main__":
    main()
