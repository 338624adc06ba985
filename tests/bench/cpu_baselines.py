#!/usr/bin/env python3
"""SURVEY 8(d)'s CPU rows for the configs bench.py does not time: the reference algorithm restated in C (the oracle), ONE
thread like the reference, and -- labelled NOT reference behaviour -- the same per-row loop on the host's usable cores with
OpenMP, on the GPU box's host.  C3 (f64 power law, 10 M rows) and C4 (7-point Laplacian 512^3 f32: SpMV, and CG iterations
of linearsolver.rs:27-61) at full size, medians; everything generated on the host by the oracle's own generators."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # noqa: E402  (the CPU baseline IS the oracle here)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import usable_cores  # noqa: E402


def med(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def spmv_rows(name, off, col, val, x, reps, threads):
    n_rows, nnz, vs = len(off) - 1, len(col), val.dtype.itemsize
    b = nnz * (vs + 4) + (n_rows + 1) * 4 + n_rows * vs + len(x) * vs
    t1 = med(lambda: oracle.spmv(off, col, val, x), reps)
    print("%s: SpMV, reference algorithm, 1 core: %.1f ms = %.1f GB/s (algorithmic %.3f GB), median of %d" % (name, t1 * 1e3, b / t1 / 1e9, b / 1e9, reps), flush=True)
    tn = med(lambda: oracle.spmv_omp(off, col, val, x, threads), reps)
    print("%s: SpMV, same loop on %d cores with OpenMP (NOT reference behaviour): %.1f ms = %.1f GB/s" % (name, threads, tn * 1e3, b / tn / 1e9), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cg-iters", type=int, default=3)
    args = ap.parse_args()
    threads = usable_cores()
    print("host: %d usable cores" % threads, flush=True)
    t0 = time.perf_counter()
    off, col, val = oracle.gen_powerlaw(0x5EED0001, args.rows, args.rows, np.float64)
    x = oracle.gen_x(0x5EED0002, args.rows, np.float64)
    print("C3 generated on the host in %.1f s: %d rows, %d entries" % (time.perf_counter() - t0, args.rows, len(col)), flush=True)
    spmv_rows("C3 f64 power law", off, col, val, x, args.reps, threads)
    del off, col, val, x
    t0 = time.perf_counter()
    g = args.grid
    off, col, val = oracle.laplace3d(g, g, g, np.float32)
    n = g ** 3
    x = oracle.gen_x(0x5EED0002, n, np.float32)
    print("C4 generated on the host in %.1f s: %d rows, %d entries" % (time.perf_counter() - t0, n, len(col)), flush=True)
    spmv_rows("C4 7-pt Laplacian %d^3 f32" % g, off, col, val, x, args.reps, threads)
    b = oracle.spmv(off, col, val, np.ones(n, np.float32))
    t0 = time.perf_counter()
    _, iters, rr = oracle.cg(n, n, off, col, val, b, np.zeros(n, np.float32), tol=0.0, iter_max=args.cg_iters)
    t = (time.perf_counter() - t0) / max(iters, 1)
    vs = 4
    b_iter = len(col) * (vs + 4) + (n + 1) * 4 + 2 * n * vs + 9 * n * vs
    print("C4 CG (linearsolver.rs:27-61), reference algorithm, 1 core: %.2f s per iteration over %d iterations = %.1f GB/s against the %.2f GB minimum-traffic figure"
          % (t, iters, b_iter / t / 1e9, b_iter / 1e9), flush=True)


if __name__ == "__main__":
    main()
