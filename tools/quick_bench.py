#!/usr/bin/env python3
"""Ad-hoc kernel timing on one GPU (development aid, not the contract bench): times every SpMV
variant on the BASELINE C2/C3 shapes with HIP events and prints GB/s against 8 TB/s."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsemat_amd as sm  # noqa: E402
from sparsemat_amd import synth  # noqa: E402


def time_variant(m, xptr, n_x, yptr, variant, reps=20, warm=3):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(warm):
        m.mvp_dev(xptr, n_x, yptr, variant, stream=st)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        m.mvp_dev(xptr, n_x, yptr, variant, stream=st)
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def algo_bytes(m, n_x):
    vs = np.dtype(m.dtype).itemsize
    return m.n_non_zero_entries() * (vs + 4) + (m.n_rows() + 1) * 4 + m.n_rows() * vs + n_x * vs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--cases", default="banded,uniform,powerlaw")
    args = ap.parse_args()
    torch.cuda.init()
    n = args.rows
    for case in args.cases.split(","):
        if case == "powerlaw":
            dtype = np.float64
            m = synth.crs_powerlaw(synth.SEED_MATRIX, n, n, dtype)
        else:
            dtype = np.float32
            pat = synth.PATTERN_BANDED if case == "banded" else synth.PATTERN_UNIFORM
            m = synth.crs_fixed(synth.SEED_MATRIX, pat, n, 32, dtype)
        xbuf, xptr = synth.gen_x(synth.SEED_X, n, dtype)
        ybuf = synth.DeviceBuffer(n * np.dtype(dtype).itemsize)
        B = algo_bytes(m, n)
        print("== %s: rows %d nnz %d dtype %s auto=%s max_row %d bytes %.3f GB" % (
            case, n, m.n_non_zero_entries(), np.dtype(dtype).name, m.resolved_variant(), m.max_row_len(), B / 1e9), flush=True)
        for lanes in (2, 4, 8, 16, 32, 64):
            m.set_vector_lanes(lanes)
            med, mn = time_variant(m, xptr, n, ybuf.ptr, "vector")
            print("  vector lanes=%-2d  median %.3f ms  min %.3f ms  %.0f GB/s (%.1f%% of 8 TB/s)" % (
                lanes, med, mn, B / med / 1e6, B / med / 1e6 / 80), flush=True)
        m.set_vector_lanes(0)
        med, mn = time_variant(m, xptr, n, ybuf.ptr, "merge")
        print("  merge            median %.3f ms  min %.3f ms  %.0f GB/s (%.1f%% of 8 TB/s)" % (
            med, mn, B / med / 1e6, B / med / 1e6 / 80), flush=True)
        med, mn = time_variant(m, xptr, n, ybuf.ptr, "seq", reps=3, warm=1)
        print("  seq              median %.3f ms" % med, flush=True)
        del m, xbuf, ybuf


if __name__ == "__main__":
    main()
