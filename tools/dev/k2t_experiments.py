#!/usr/bin/env python3
"""Round-4 experiments on K2t (spmv_tiled.hip), BASELINE C3 (f64 power law) and C2-uniform (f32); development aid, prints only.
  full      the whole matrix with the library SPARSEMAT_HIP_LIB names (A/B of builds: slice width / workgroup size per value type)
  subset K  rows [0, n/K) of the same matrix as a matrix of its own (all columns): its product stream is 1/K of the whole one's and
            stays in the 256 MiB Infinity Cache between the passes -- K x its time is what a row-super-block schedule of the two
            passes could reach at best (x is staged K times, pass 2 has 1/K of the row blocks), for several tile targets."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sparsemat_amd as sm  # noqa: E402
from sparsemat_amd import _lib, synth  # noqa: E402
from bench import Events, stats  # noqa: E402

lib, check = sm.lib(), _lib.check


def timed(m, xptr, n_x, yptr, stream, reps=12, warm=3, variant="tiled"):
    for _ in range(warm):
        m.mvp_dev(xptr, n_x, yptr, variant, stream=stream.value)
    check(lib.smh_stream_synchronize(stream))
    ev = Events(lib, check, reps)
    for i in range(reps):
        ev.start(i, stream.value)
        m.mvp_dev(xptr, n_x, yptr, variant, stream=stream.value)
        ev.stop(i, stream.value)
    check(lib.smh_stream_synchronize(stream))
    return stats(ev.times_ms())


def make(case, n, r0, r1):
    if case == "c3":
        return synth.crs_powerlaw(synth.SEED_MATRIX, n, n, np.float64, row_begin=r0, row_end=r1), np.float64
    return synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_UNIFORM, n, 32, np.float32, r0, r1), np.float32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="c3,uniform")
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--subsets", default="", help="comma list of K")
    ap.add_argument("--tiles", default="", help="comma list of SMH_TILED_TILE values for the subsets (default: the library's)")
    args = ap.parse_args()
    check(lib.smh_set_device(0))
    import ctypes as C
    s = C.c_void_p()
    check(lib.smh_stream_create(C.byref(s)))
    n = args.rows
    print("library:", sm.LIB_PATH, flush=True)
    for case in args.cases.split(","):
        m, dtype = make(case, n, 0, n)
        vs = np.dtype(dtype).itemsize
        xb, xp = synth.gen_x(synth.SEED_X, n, dtype)
        yb = synth.DeviceBuffer(n * vs)
        nnz = m.n_non_zero_entries()
        B = nnz * (vs + 4) + (n + 1) * 4 + 2 * n * vs
        lay = m.tiled_layout()
        t = timed(m, xp, n, yb.ptr, s)
        print("%s full: %d entries, %d slices of %d columns x %d row blocks, %d products (%.3f per entry): median %.4f ms (min %.4f) -> %.3f of 8 TB/s"
              % (case, nnz, lay["n_slices"], lay["slice_columns"], lay["n_row_blocks"], lay["n_products"], lay["n_products"] / nnz, t["median"], t["min"],
                 B / t["median"] / 1e6 / 8000), flush=True)
        del m
        for K in [int(v) for v in args.subsets.split(",") if v]:
            for tile in ([None] + [v for v in args.tiles.split(",") if v]):
                if tile is None:
                    os.environ.pop("SMH_TILED_TILE", None)
                else:
                    os.environ["SMH_TILED_TILE"] = tile
                sub, _ = make(case, n, 0, n // K)
                lay = sub.tiled_layout()
                ysub = synth.DeviceBuffer((n // K) * vs)
                t = timed(sub, xp, n, ysub.ptr, s)
                print("%s rows [0, n/%d): tile target %s -> %d row blocks (rows per block %d), %d products = %.0f MB of stream: median %.4f ms  x %d = %.4f ms"
                      % (case, K, tile or "default", lay["n_row_blocks"], lay["rows_per_block"], lay["n_products"], lay["n_products"] * (vs + 2) / 1e6,
                         t["median"], K, K * t["median"]), flush=True)
                del sub, ysub
        os.environ.pop("SMH_TILED_TILE", None)
        del xb, yb


if __name__ == "__main__":
    main()
