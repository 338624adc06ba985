#!/usr/bin/env python3
"""Ad-hoc timing of SparseMatrix::transpose (smh_crs_transpose) and the column tables (smh_crs_column_info_dev) on one
GPU: BASELINE C2 (10M x 32, banded) and the 7-point Laplacian (default 256^3), matrices generated in HBM.  Prints ms
and entries/s (host-synchronous calls, median), SMH_ASSEMBLE_TIMING-style stage times on request, and -- on a bounded
sample -- the rate of the reference algorithm restated in C (oracle.transpose: Vec::insert per entry, one core) with a
bit-exact comparison."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # noqa: E402  (checker / CPU baseline only)
import sparsemat_amd as sm  # noqa: E402
from sparsemat_amd import synth  # noqa: E402


def med(fn, reps):
    ts = []
    for _ in range(reps + 1):
        sm.lib().smh_device_synchronize()
        t0 = time.perf_counter()
        out = fn()
        sm.lib().smh_device_synchronize()
        ts.append(time.perf_counter() - t0)
        del out
    ts = sorted(ts[1:])
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu-rows", type=int, default=20_000, help="rows of the CPU sample (O(n * nnz) algorithm)")
    args = ap.parse_args()
    torch.cuda.init()
    cases = [("C2 banded f32 %dM x 32" % (args.rows // 1_000_000), lambda: synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, args.rows, 32, np.float32)),
             ("7-pt Laplacian %d^3 f32" % args.grid, lambda: synth.crs_laplace3d(args.grid, args.grid, args.grid, np.float32))]
    for name, make in cases:
        a = make()
        nnz = a.n_non_zero_entries()
        t_med, t_min = med(a.transpose, args.reps)
        t = a.transpose()
        print("%s: %d entries | transpose median %.1f ms (min %.1f) = %.2f G entries/s; A^T: rows %d cols %d entries %d" %
              (name, nnz, t_med * 1e3, t_min * 1e3, nnz / t_med / 1e9, t.n_rows(), t.n_cols(), t.n_non_zero_entries()), flush=True)
        del t
        rows = torch.empty(nnz, dtype=torch.int32, device="cuda")
        ents = torch.empty(nnz, dtype=torch.int32, device="cuda")
        ptr = torch.empty(a.n_cols() + 1, dtype=torch.int32, device="cuda")

        def info():
            sm._lib.check(sm.lib().smh_crs_column_info_dev(a._h, rows.data_ptr(), ptr.data_ptr(), ents.data_ptr()))
        c_med, c_min = med(info, args.reps)
        print("%s: column_info median %.1f ms (min %.1f) = %.2f G entries/s" % (name, c_med * 1e3, c_min * 1e3, nnz / c_med / 1e9),
              flush=True)
        del a, rows, ents, ptr
    # the reference algorithm on one core, bounded sample, with parity
    n = args.cpu_rows
    small = synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, n, 32, np.float32)
    off, col, val = small.raw_parts()
    t0 = time.perf_counter()
    e = oracle.transpose(off, col, val)
    cpu = time.perf_counter() - t0
    t = small.transpose()
    g_off, g_col, g_val = t.raw_parts()
    ok = (t.n_rows(), t.n_cols()) == (e[0], e[1]) and np.array_equal(g_off, e[2]) and np.array_equal(g_col, e[3]) \
        and g_val.tobytes() == e[4].tobytes()
    print("cpu oracle (reference algorithm, 1 core): %d rows x 32 = %d entries in %.2f s = %.4f G entries/s; device result bit-exact: %s"
          % (n, len(col), cpu, len(col) / cpu / 1e9, ok), flush=True)
    assert ok


if __name__ == "__main__":
    main()
