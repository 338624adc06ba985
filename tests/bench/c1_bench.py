#!/usr/bin/env python3
"""BASELINE config C1 (the reference's own CPU-sized case: 5-point Laplacian on a 1000 x 1000 grid) on one GPU:
SpMV per launch and CG per iteration (hipGraph replay), next to the CPU oracle on the same host.  At 45 MB the
problem fits the L2/MALL: this measures launch and latency overheads, not HBM."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # noqa: E402  (checker / CPU baseline only)
import sparsemat_amd as sm  # noqa: E402


def main():
    torch.cuda.init()
    for dtype in (np.float32, np.float64):
        g = 1000
        n = g * g
        off, col, val = oracle.laplace2d(g, g, dtype)
        m = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val)
        x = np.ones(n, dtype)
        xd = torch.from_numpy(x).cuda()
        yd = torch.zeros(n, dtype=xd.dtype, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            m.mvp_dev(xd.data_ptr(), n, yd.data_ptr(), "auto", stream=st)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            m.mvp_dev(xd.data_ptr(), n, yd.data_ptr(), "auto", stream=st)
        b.record()
        b.synchronize()
        t_spmv = a.elapsed_time(b) / 200
        y_ref = oracle.spmv(off, col, val, x)
        exact = bool(np.array_equal(yd.cpu().numpy().view(np.uint8), y_ref.view(np.uint8)))
        t0 = time.perf_counter()
        for _ in range(20):
            oracle.spmv(off, col, val, x)
        t_cpu = (time.perf_counter() - t0) / 20 * 1e3
        bvec = oracle.spmv(off, col, val, np.ones(n, dtype))
        iters = 400
        xs, bd = sm.DenseVec.from_vec(np.zeros(n, dtype)), sm.DenseVec.from_vec(bvec)
        cg = sm.ConjugateGradient(1e-30, iters, check_every=50)
        cg.solve(m, bd, xs)  # warm (plans, graph instantiation)
        xs = sm.DenseVec.from_vec(np.zeros(n, dtype))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cg.solve(m, bd, xs)
        torch.cuda.synchronize()
        t_cg = (time.perf_counter() - t0) / cg.iterations * 1e3
        nbytes = len(val) * (val.itemsize + 4) + (n + 1) * 4 + 2 * n * val.itemsize
        print("C1 5-pt Laplacian 1000x1000 %s: auto=%s  SpMV %.4f ms/launch (%.0f GB/s, bit-exact vs oracle: %s; oracle 1 core "
              "%.2f ms)  CG %.4f ms/iteration over %d iterations" % (np.dtype(dtype).name, m.resolved_variant(), t_spmv,
                                                                    nbytes / t_spmv / 1e6, exact, t_cpu, t_cg, cg.iterations), flush=True)


if __name__ == "__main__":
    main()
