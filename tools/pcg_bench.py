#!/usr/bin/env python3
"""Ad-hoc: Jacobi-preconditioned CG (smh_pcg_jacobi_solve, an extension) per iteration on the 7-point Laplacian g^3 f32,
next to the plain CG of the same library.  The entry point takes host vectors, so the per-iteration time is the
difference quotient of two iteration counts (upload / download cancel)."""
import argparse
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
    args = ap.parse_args()
    g, dtype = args.grid, np.float32
    n = g ** 3
    a = synth.crs_laplace3d(g, g, g, dtype)
    b = np.ones(n, dtype)
    vs = 4
    spmv_bytes = a.n_non_zero_entries() * (vs + 4) + (n + 1) * 4 + 2 * n * vs

    def run(solver, iters):
        x = np.zeros(n, dtype)
        t0 = time.perf_counter()
        solver(iters).solve(a, b, x)
        return time.perf_counter() - t0

    # vector streams of the tail (what the kernels move; p.Ap rides the K1s epilogue): PCG = update 4 (r, Ap, d in; r out) +
    # x/p sweep 6 (p, x, r, d in; p, x out); CG = update 3 + x/p sweep 5 (DESIGN K5)
    for name, make, streams in (("Jacobi PCG", lambda it: sm.JacobiConjugateGradient(0.0, it), 10),
                                ("CG (host vectors)", lambda it: sm.ConjugateGradient(0.0, it), 8)):
        bytes_iter = spmv_bytes + streams * n * vs
        run(make, 3)  # warm-up (plans)
        t_a, t_b = run(make, 20), run(make, 80)
        per = (t_b - t_a) / 60
        print("laplace3d %d^3 f32 %-18s %.3f ms per iteration (20 iterations %.1f ms, 80 iterations %.1f ms); %.0f GB/s against %.2f GB "
              "per iteration" % (g, name, per * 1e3, t_a * 1e3, t_b * 1e3, bytes_iter / per / 1e9, bytes_iter / 1e9), flush=True)


if __name__ == "__main__":
    main()
