#!/usr/bin/env python3
"""BASELINE C4: CG on the 7-point Laplacian (default 512^3, f32), device-resident, fixed iteration count.
Prints ms/iteration and GB/s against the minimum-traffic figure of SURVEY 8(d) (B_spmv + 9*n*sizeof T)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsemat_amd as sm  # noqa: E402
from sparsemat_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--variant", default="auto")
    args = ap.parse_args()
    dtype = np.float32 if args.dtype == "f32" else np.float64
    vs = np.dtype(dtype).itemsize
    g = args.grid
    n = g * g * g
    t0 = time.time()
    a = synth.crs_laplace3d(g, g, g, dtype)
    nnz = a.n_non_zero_entries()
    print("laplace3d %d^3: n=%d nnz=%d built in %.2fs; auto=%s ring=%s" % (g, n, nnz, time.time() - t0, a.resolved_variant(),
                                                                         a.ring_plan()[1:3]), flush=True)
    ones = sm.DenseVec.from_vec(np.ones(n, dtype))
    b = a.mvp(ones, variant=args.variant)  # b = A.1 so that x* = 1
    x = sm.DenseVec.zeros(n, dtype)
    # SpMV alone
    sm.lib().smh_device_synchronize()
    y = sm.DenseVec.zeros(n, dtype)
    for variant in ("vector", "merge", "auto"):
        a.mvp_dev(ones.data_ptr(), n, y.data_ptr(), variant)
        sm.lib().smh_device_synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            a.mvp_dev(ones.data_ptr(), n, y.data_ptr(), variant)
        sm.lib().smh_device_synchronize()
        dt = (time.perf_counter() - t0) / reps
        b_spmv = nnz * (vs + 4) + (n + 1) * 4 + 2 * n * vs
        print("  SpMV %-6s %.3f ms  %.0f GB/s (%.1f%% of 8 TB/s)" % (variant, dt * 1e3, b_spmv / dt / 1e9, b_spmv / dt / 8e10), flush=True)
    cg = sm.ConjugateGradient(0.0, args.iters, variant=args.variant, check_every=args.iters)
    cg.solve(a, b, x)  # warm-up (allocations, plans)
    x = sm.DenseVec.zeros(n, dtype)
    t0 = time.perf_counter()
    cg.solve(a, b, x)
    dt = time.perf_counter() - t0
    per_it = dt / cg.iterations
    b_min = b_spmv + 9 * n * vs
    b_unfused = b_spmv + 12 * n * vs
    err = float(np.abs(x.to_numpy() - 1).max())
    out = {"config": "CG 7-pt Laplacian %d^3 %s" % (g, args.dtype), "iterations": cg.iterations, "ms_per_iteration": per_it * 1e3,
           "GBps_vs_min_traffic": b_min / per_it / 1e9, "pct_of_8TBps": b_min / per_it / 8e10,
           "GBps_vs_reference_op_sequence": b_unfused / per_it / 1e9, "r_norm": float(np.sqrt(cg.r_norm_squared)),
           "max_abs_err_vs_ones": err}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
