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


def report(tag, B, med, mn):
    print("  %-22s median %.3f ms  min %.3f ms  %.0f GB/s (%.1f%% of 8 TB/s)" % (
        tag, med, mn, B / med / 1e6, B / med / 1e6 / 80), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--k", type=int, default=32, help="entries per row of the fixed-length cases")
    ap.add_argument("--cases", default="banded,diag,uniform,powerlaw")
    ap.add_argument("--lanes", default="4,8,16")
    ap.add_argument("--only-blocked", action="store_true", help="time only the column-blocked variants (K2c / K2f) and AUTO")
    ap.add_argument("--cb-shifts", default="", help="K2c column-block widths to time (log2 columns), e.g. 18,19,20")
    ap.add_argument("--tiled", default="uniform,powerlaw", help="cases that also time the 2-D tiled variant (K2t)")
    args = ap.parse_args()
    torch.cuda.init()
    n = args.rows
    for case in args.cases.split(","):
        if case in ("powerlaw", "powerlaw32"):
            dtype = np.float64 if case == "powerlaw" else np.float32
            m = synth.crs_powerlaw(synth.SEED_MATRIX, n, n, dtype)
        elif case.startswith("lap"):
            dtype = np.float32
            g = int(case[3:])
            m = synth.crs_laplace3d(g, g, g, dtype)
        else:
            dtype = np.float64 if case.endswith("64") else np.float32
            pat = {"banded": synth.PATTERN_BANDED, "uniform": synth.PATTERN_UNIFORM, "diag": synth.PATTERN_DIAG, "window": synth.PATTERN_WINDOW}[
                case.replace("64", "")]
            m = synth.crs_fixed(synth.SEED_MATRIX, pat, n, args.k, dtype)
        nr = m.n_rows()
        xbuf, xptr = synth.gen_x(synth.SEED_X, nr, dtype)
        ybuf = synth.DeviceBuffer(nr * np.dtype(dtype).itemsize)
        B = algo_bytes(m, nr)
        nb, frac, act, ptr, ph = m.ring_plan()
        print("== %s: rows %d nnz %d dtype %s auto=%s max_row %d bytes %.3f GB | ring: blocks %d phases %d fraction %.3f active %s" % (
            case, nr, m.n_non_zero_entries(), np.dtype(dtype).name, m.resolved_variant(), m.max_row_len(), B / 1e9,
            nb, len(ph), frac, act), flush=True)
        for spec in ([] if args.only_blocked else args.lanes.split(",")):  # "lanes" or "lanes:chunks"
            lanes, chunks = (int(v) for v in (spec + ":0").split(":")[:2])
            m.set_vector_lanes(lanes)
            m.set_vector_chunks(chunks)
            for ring in ((0, 1) if chunks == 0 else (1,)):
                m.set_ring(ring)
                med, mn = time_variant(m, xptr, nr, ybuf.ptr, "vector")
                report("vector lanes=%s %s" % (spec, "K1r" if ring else "K1"), B, med, mn)
        m.set_vector_lanes(0)
        m.set_vector_chunks(0)
        m.set_ring(-1)
        if not args.only_blocked:
            med, mn = time_variant(m, xptr, nr, ybuf.ptr, "merge")
            report("merge", B, med, mn)
        if not args.only_blocked:
            med, mn = time_variant(m, xptr, nr, ybuf.ptr, "stream", reps=8 if case != "lap512" and not case.startswith("lap") else 20)
            report("stream (K1s)", B, med, mn)
        for shift in [int(v) for v in args.cb_shifts.split(",") if v]:
            m.set_colblock_shift(shift)
            try:
                cb = m.colblock(arrays=False)
            except sm.SparseMatPanic as e:  # more than 128 column blocks
                print("  colblock 2^%d: %s" % (shift, e), flush=True)
                continue
            med, mn = time_variant(m, xptr, nr, ybuf.ptr, "colblock", reps=8)
            report("colblock 2^%d x%d rpt%d" % (cb["shift"], cb["n_blocks"], cb["rows_per_thread"]), B, med, mn)
            cf = m.colfused(arrays=False)
            if cf["fits"]:
                med, mn = time_variant(m, xptr, nr, ybuf.ptr, "colfused", reps=8)
                report("colfused 2^%d x%d rt%d" % (cf["shift"], cf["n_blocks"], cf["rows_per_lane"]), B, med, mn)
        m.set_colblock_shift(0)
        if args.cb_shifts and m.colsplit_flag():
            med, mn = time_variant(m, xptr, nr, ybuf.ptr, "colsplit", reps=8)
            report("colsplit (long rows K2c + short rows K2f)", B, med, mn)
        if args.tiled and case in args.tiled.split(","):
            import time
            t0 = time.perf_counter()
            lay = m.tiled_layout()
            t_plan = time.perf_counter() - t0
            med, mn = time_variant(m, xptr, nr, ybuf.ptr, "tiled", reps=8)
            report("tiled (K2t) %d slices x %d row blocks of %d (copy built in %.0f ms)" % (
                lay["n_slices"], lay["n_row_blocks"], lay["rows_per_block"], t_plan * 1e3), B, med, mn)
        med, mn = time_variant(m, xptr, nr, ybuf.ptr, "auto")
        report("auto", B, med, mn)
        del m, xbuf, ybuf


if __name__ == "__main__":
    main()
