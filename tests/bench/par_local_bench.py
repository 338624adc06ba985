#!/usr/bin/env python3
"""Ad-hoc: cost of the single-process SparseMatPar path (smh_par_*) on ONE GPU -- every block on device 0, so what is
measured is the partition machinery itself (per-block launches on separate streams, halo copies between the blocks'
buffers, host folds of the dot products, unfused vector updates) against the fused single-matrix solver on the same
system.  7-point Laplacian g^3 f32, fixed iteration count (tol = 0)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # noqa: E402  (host generator only)
import sparsemat_amd as sm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--iters", type=int, default=100)
    args = ap.parse_args()
    torch.cuda.init()
    g, dtype = args.grid, np.float32
    off, col, val = oracle.laplace3d(g, g, g, dtype)
    n = g ** 3
    b = np.ones(n, dtype)
    a = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val)
    x = sm.DenseVec.zeros(n, dtype)
    bv = sm.DenseVec.from_vec(b)
    cg = sm.ConjugateGradient(0.0, args.iters)
    cg.solve(a, bv, x)  # warm-up (plans, graph)
    x = sm.DenseVec.zeros(n, dtype)
    t0 = time.perf_counter()
    cg.solve(a, bv, x)
    sm.lib().smh_device_synchronize()
    t_single = time.perf_counter() - t0
    x_single = x.to_numpy()
    print("laplace3d %d^3 f32, %d CG iterations: single matrix (fused, hipGraph) %.3f ms per iteration" %
          (g, cg.iterations, t_single * 1e3 / cg.iterations), flush=True)
    del a
    for n_blocks in (1, 2, 4, 8):
        m = sm.SparseMatParLocal.with_sub_matrices(n_blocks, n, n, off, col, val, device_ids=[0] * n_blocks)
        xp = np.zeros(n, dtype)
        m.cg_solve(b, xp, tol=0.0, iter_max=3)  # warm-up
        xp = np.zeros(n, dtype)
        t0 = time.perf_counter()
        iters, rr = m.cg_solve(b, xp, tol=0.0, iter_max=args.iters)
        t = time.perf_counter() - t0
        err = float(np.max(np.abs(xp.astype(np.float64) - x_single.astype(np.float64))) / np.max(np.abs(x_single)))
        print("  %d block(s) on device 0: %.3f ms per iteration (incl. upload of b, x and download of x: %.1f ms total), "
              "max |x - x_single| / max |x| = %.2e (unconverged f32 iterates: rounding order differs)"
              % (n_blocks, t * 1e3 / iters, t * 1e3, err), flush=True)
        del m


if __name__ == "__main__":
    main()
