#!/usr/bin/env python3
"""Is the plain lane-group kernel (K1: VECTOR with the ring off) bit for bit the LDS-ring kernel (K1r)?  (development aid)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sparsemat_amd as sm  # noqa: E402
from sparsemat_amd import synth  # noqa: E402

rng = np.random.default_rng(5)
for dtype in (np.float32, np.float64):
    u = np.uint32 if dtype == np.float32 else np.uint64
    cases = [("window 1M x 32", synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_WINDOW, 1_000_000, 32, dtype)),
             ("banded 300k x 32", synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, 300_000, 32, dtype)),
             ("banded 200k x 7", synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, 200_000, 7, dtype))]
    n_rows = 60_000
    lens = rng.integers(0, 90, n_rows)
    off = np.zeros(n_rows + 1, np.uint32)
    np.cumsum(lens, out=off[1:])
    centers = np.repeat(np.arange(n_rows), lens)
    col = np.clip(centers + rng.integers(-3000, 3000, len(centers)), 0, n_rows - 1).astype(np.uint32)
    cases.append(("ragged rows 0..89, band +-3000", sm.SparseMatCRS.from_raw_parts(n_rows, n_rows, off, col, rng.uniform(-1, 1, len(col)).astype(dtype))))
    for name, m in cases:
        n = m.n_cols()
        x = rng.uniform(-1, 1, n).astype(dtype)
        for lanes in (0, 4, 8, 16):
            m.set_vector_lanes(lanes)
            m.set_ring(1)
            y1 = m.mvp(x, variant="vector")
            m.set_ring(0)
            y0 = m.mvp(x, variant="vector")
            d = int((y1.view(u) != y0.view(u)).sum())
            print("%s %-32s lanes %2d: %d of %d rows differ" % (np.dtype(dtype).name, name, lanes, d, len(y1)), flush=True)
