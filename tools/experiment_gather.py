#!/usr/bin/env python3
"""Experiment: random-gather rate of the SpMV kernels as a function of the x footprint (does an L2-resident
x block lift the ~59 G gathers/s of the uniform-column case?).  rows x 32 entries, columns uniform in [0, n_cols)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsemat_amd as sm  # noqa: E402
from sparsemat_amd import synth  # noqa: E402


def main():
    torch.cuda.init()
    rows, k = 4_000_000, 32
    rng = np.random.default_rng(1)
    off = (np.arange(rows + 1, dtype=np.uint64) * k).astype(np.uint32)
    val = rng.uniform(-1, 1, rows * k).astype(np.float32)
    for n_cols in (65_536, 262_144, 524_288, 1_048_576, 2_097_152, 4_194_304, 16_777_216):
        col = rng.integers(0, n_cols, rows * k, dtype=np.uint32)
        m = sm.SparseMatCRS.from_raw_parts(rows, n_cols, off, col, val)
        xbuf, xptr = synth.gen_x(synth.SEED_X, n_cols, np.float32)
        ybuf = synth.DeviceBuffer(rows * 4)
        st = torch.cuda.current_stream().cuda_stream
        out = []
        for variant in ("vector", "merge", "colblock"):
            for _ in range(3):
                m.mvp_dev(xptr, n_cols, ybuf.ptr, variant, stream=st)
            torch.cuda.synchronize()
            ts = []
            for _ in range(10):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); m.mvp_dev(xptr, n_cols, ybuf.ptr, variant, stream=st); b.record(); b.synchronize()
                ts.append(a.elapsed_time(b))
            ts.sort()
            out.append("%s %.3f ms (%.0f G gathers/s)" % (variant, ts[5], rows * k / ts[5] / 1e6))
        print("x = %6.1f MB (%9d cols), ring fraction %.2f: %s" % (n_cols * 4 / 1e6, n_cols, m.ring_plan()[1], "; ".join(out)), flush=True)
        del m


if __name__ == "__main__":
    main()
