#!/usr/bin/env python3
"""Ad-hoc timing of the on-device assembly (smh_crs_assemble_dev) on one GPU: element-by-element stream of a
g^3-cell hexahedral mesh (64 add_to calls per cell; every entry of the 27-point stencil receives up to 8
contributions), operations already resident in HBM.  Prints operations/s, the CPU oracle's rate on a bounded
sample of the same stream (the reference algorithm: one core, linked-list walk per call) and checks parity on
that sample."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # noqa: E402  (checker / CPU baseline only)
import sparsemat_amd as sm  # noqa: E402


def hex_stream(g, dtype, rng):
    nodes = np.arange((g + 1) ** 3, dtype=np.uint32).reshape(g + 1, g + 1, g + 1)
    corners = np.stack([nodes[dx:g + dx, dy:g + dy, dz:g + dz].ravel()
                        for dx in (0, 1) for dy in (0, 1) for dz in (0, 1)], axis=1)
    rows = np.repeat(corners, 8, axis=1).ravel()
    cols = np.tile(corners, (1, 8)).ravel()
    vals = rng.uniform(-1, 1, len(rows)).astype(dtype)
    return rows, cols, vals


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=128, help="cells per edge (128: 2.1M rows, 134M operations)")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu-grid", type=int, default=48, help="cells per edge of the CPU oracle sample")
    args = ap.parse_args()
    dtype = np.float32 if args.dtype == "f32" else np.float64
    torch.cuda.init()
    rng = np.random.default_rng(1)
    rows, cols, vals = hex_stream(args.grid, dtype, rng)
    n = len(vals)
    d_rows = torch.from_numpy(rows.view(np.int32)).cuda()
    d_cols = torch.from_numpy(cols.view(np.int32)).cuda()
    d_vals = torch.from_numpy(vals).cuda()
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps + 1):
        t0 = time.perf_counter()
        m = sm.SparseMatCRS.from_device_triplets(n, d_rows.data_ptr(), d_cols.data_ptr(), d_vals.data_ptr(), dtype)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        shape = (m.n_rows(), m.n_cols(), m.n_non_zero_entries(), m.max_row_len())
        del m
    ts = sorted(ts[1:])
    med = ts[len(ts) // 2]
    print("device: grid %d^3 cells, %d operations -> rows %d cols %d entries %d max row %d | median %.1f ms (min %.1f) "
          "= %.2f G operations/s" % (args.grid, n, *shape, med * 1e3, ts[0] * 1e3, n / med / 1e9), flush=True)
    t0 = time.perf_counter()
    ms = sm.SparseMatCRS.from_device_triplets(n, d_rows.data_ptr(), d_cols.data_ptr(), d_vals.data_ptr(), dtype)
    ms.sort_rows()
    torch.cuda.synchronize()
    print("device: assemble + sort_rows %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    # CPU oracle on a smaller mesh of the same kind + parity
    rows_c, cols_c, vals_c = hex_stream(args.cpu_grid, dtype, np.random.default_rng(2))
    t0 = time.perf_counter()
    expect = oracle.assemble(rows_c, cols_c, vals_c)
    t_cpu = time.perf_counter() - t0
    mc = sm.SparseMatCRS.from_triplets(rows_c, cols_c, vals_c)
    off, col, val = mc.raw_parts()
    ok = (mc.n_rows(), mc.n_cols()) == expect[:2] and np.array_equal(off, expect[2]) and np.array_equal(col, expect[3]) \
        and val.tobytes() == expect[4].tobytes()
    print("cpu oracle (reference algorithm, 1 core): grid %d^3, %d operations in %.2f s = %.4f G operations/s; "
          "device result bit-exact: %s" % (args.cpu_grid, len(vals_c), t_cpu, len(vals_c) / t_cpu / 1e9, ok), flush=True)
    if not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
