#!/usr/bin/env python3
"""Where the one-process partitioned CG spends its time (development aid): host issue time against drain time per batch of
iterations (SMH_PAR_TRACE=1), n blocks on device 0, 7-point Laplacian g^3 f32."""
import os
import sys

import numpy as np

os.environ["SMH_PAR_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # noqa: E402  (host generator only)
import sparsemat_amd as sm  # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
check_every = int(sys.argv[3]) if len(sys.argv) > 3 else 25
off, col, val = oracle.laplace3d(g, g, g, np.float32)
n = g ** 3
m = sm.SparseMatParLocal.with_sub_matrices(n_blocks, n, n, off, col, val, device_ids=[0] * n_blocks)
b, x = m.vec(host=np.ones(n, np.float32)), m.vec()
m.cg_solve_vec(b, x, tol=0.0, iter_max=5, check_every=5)
sys.stderr.write("== measured\n")
b, x = m.vec(host=np.ones(n, np.float32)), m.vec()
m.cg_solve_vec(b, x, tol=0.0, iter_max=100, check_every=check_every)
