#!/usr/bin/env python3
"""Ad-hoc: SpMV variants on a 27-point-stencil matrix (trilinear hexahedra on a g^3-cell grid), assembled on the
device from its element stream -- rows of up to 27 entries in first-appearance order (unsorted), three planes of
columns per row."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsemat_amd as sm  # noqa: E402
from sparsemat_amd import synth  # noqa: E402
from assemble_bench import hex_stream  # noqa: E402
from quick_bench import algo_bytes, report, time_variant  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=160)
    ap.add_argument("--sort", action="store_true", help="sort_rows() before timing")
    args = ap.parse_args()
    torch.cuda.init()
    rows, cols, vals = hex_stream(args.grid, np.float32, np.random.default_rng(1))
    m = sm.SparseMatCRS.from_triplets(rows, cols, vals)
    del rows, cols, vals
    if args.sort:
        m.sort_rows()
    n = m.n_rows()
    xbuf, xptr = synth.gen_x(synth.SEED_X, n, np.float32)
    ybuf = synth.DeviceBuffer(n * 4)
    B = algo_bytes(m, n)
    print("27-point stencil, grid %d^3 cells: rows %d nnz %d (mean %.1f) auto=%s ring entries %d fraction %.3f bytes %.3f GB%s" % (
        args.grid, n, m.n_non_zero_entries(), m.n_non_zero_entries() / n, m.resolved_variant(), m.ring_entries(),
        m.ring_plan()[1], B / 1e9, " (rows sorted)" if args.sort else ""), flush=True)
    for lanes in (4, 8):
        m.set_vector_lanes(lanes)
        med, mn = time_variant(m, xptr, n, ybuf.ptr, "vector")
        report("vector lanes=%d K1r" % lanes, B, med, mn)
    m.set_vector_lanes(0)
    for variant in ("merge", "stream", "auto"):
        med, mn = time_variant(m, xptr, n, ybuf.ptr, variant)
        report(variant, B, med, mn)


if __name__ == "__main__":
    main()
