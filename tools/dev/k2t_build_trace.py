#!/usr/bin/env python3
"""Stage times of K2t's build (SMH_TILED_BUILD_TRACE=1) on C2-uniform and C3, plus smh_crs_prepare_stats; development aid."""
import os
import sys

import numpy as np

os.environ["SMH_TILED_BUILD_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sparsemat_amd as sm  # noqa: E402
from sparsemat_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
for rep in range(2):
    for case in ("uniform", "c3"):
        if case == "c3":
            m = synth.crs_powerlaw(synth.SEED_MATRIX, n, n, np.float64)
        else:
            m = synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_UNIFORM, n, 32, np.float32)
        sys.stderr.write("== %s (pass %d)\n" % (case, rep))
        ms, nb = m.prepare_stats("tiled")
        sys.stderr.write("== %s: prepare_stats %.2f ms, %.2f GB derived\n" % (case, ms, nb / 1e9))
        del m
