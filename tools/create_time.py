import time, numpy as np, torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import sparsemat_amd as sm
from sparsemat_amd import synth
torch.cuda.init()
for name, fn in (("lap512", lambda: synth.crs_laplace3d(512, 512, 512, np.float32)),
                 ("C2 banded", lambda: synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, 10_000_000, 32, np.float32)),
                 ("C2 uniform", lambda: synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_UNIFORM, 10_000_000, 32, np.float32))):
    for rep in range(2):
        t0 = time.perf_counter(); m = fn(); sm.lib().smh_device_synchronize(); t1 = time.perf_counter()
        v = m.resolved_variant()
        t2 = time.perf_counter(); m.prepare(); sm.lib().smh_device_synchronize(); t3 = time.perf_counter()
        print("%s: generate+create %.1f ms, prepare(%s) %.1f ms" % (name, (t1 - t0) * 1e3, v[0], (t3 - t2) * 1e3), flush=True)
        del m
