import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, oracle, sparsemat_amd as sm
for g, dtype, tol, itmax in ((12, np.float64, 1e-9, 500), (12, np.float32, 1e-3, 500), (64, np.float64, 1e-9, 2000), (14, np.float32, 1e-4, 500), (14, np.float64, 1e-10, 500)):
    n = g ** 3
    off, col, val = oracle.laplace3d(g, g, g, dtype)
    b = oracle.spmv(off, col, val, np.ones(n, dtype))
    x_ref, it_ref, rr_ref = oracle.cg(n, n, off, col, val, b, np.zeros(n, dtype), tol=tol, iter_max=itmax)
    m = sm.SparseMatCRS.from_raw_parts(n, n, off, col, val)
    for variant in ("seq", "stream", "auto", "merge", "vector"):
        xd, bd = sm.DenseVec.from_vec(np.zeros(n, dtype)), sm.DenseVec.from_vec(b)
        cg = sm.ConjugateGradient(tol, itmax, variant=variant)
        cg.solve(m, bd, xd)
        x = xd.to_numpy().astype(np.float64)
        print("g=%d %s %-6s iters %d (oracle %d)  |x-x_ref|/tol %.2f  |x-1|/tol %.2f  |x_ref-1|/tol %.2f" % (
            g, np.dtype(dtype).name, variant, cg.iterations, it_ref, np.abs(x - x_ref).max() / tol, np.abs(x - 1).max() / tol, np.abs(x_ref.astype(np.float64) - 1).max() / tol), flush=True)
    for nb in (1, 2, 4, 7):
        mp = sm.SparseMatParLocal.with_sub_matrices(nb, n, n, off, col, val, device_ids=[0] * nb)
        x = np.zeros(n, dtype)
        iters, rr = mp.cg_solve(b, x, tol=tol, iter_max=itmax)
        print("   par nb=%d iters %d (oracle %d) |x-x_ref|/tol %.2f" % (nb, iters, it_ref, np.abs(x.astype(np.float64) - x_ref).max() / tol), flush=True)
