#!/usr/bin/env python3
"""Ad-hoc: DenseVec kernels (K3 element-wise, K4 reductions) on 2^27-element vectors (512 MiB of f32): time per call
(host-synchronous API, median of 20) and GB/s against the bytes each operation must move."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsemat_amd as sm  # noqa: E402


def med(fn, reps=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        sm.lib().smh_device_synchronize()
        t0 = time.perf_counter()
        fn()
        sm.lib().smh_device_synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3


def main():
    torch.cuda.init()
    n = 1 << 27
    for dtype in (np.float32, np.float64):
        vs = np.dtype(dtype).itemsize
        x = sm.DenseVec.from_vec(np.ones(n, dtype))
        y = sm.DenseVec.from_vec(np.full(n, 0.5, dtype))
        ops = [("add  x += y", lambda: x.add(y), 3), ("sub  x -= y", lambda: x.sub(y), 3), ("scale x *= a", lambda: x.scale(1.0000001), 2),
               ("axpy x += a*y", lambda: x.axpy(1e-9, y), 3), ("xpby x = b*x + y", lambda: x.xpby(0.999, y), 3),
               ("dot  x.y", lambda: x.inner_prod(y), 2), ("norm_squared", lambda: x.norm_squared(), 1)]
        for name, fn, streams in ops:
            ms = med(fn)
            print("%s n=2^27 %-18s %.3f ms  %.0f GB/s (%.1f%% of 8 TB/s)" % (np.dtype(dtype).name, name, ms, streams * n * vs / ms / 1e6,
                                                                           streams * n * vs / ms / 1e6 / 80), flush=True)


if __name__ == "__main__":
    main()
