#!/usr/bin/env python3
"""Which Laplacians get the staged-x bodies when forced (diagnostic)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sparsemat_amd as sm
from sparsemat_amd import synth
for g in ((40, 40, 40), (48, 48, 48), (50, 50, 50), (64, 64, 9), (64, 64, 64), (80, 80, 80), (100, 100, 100), (128, 128, 128), (128, 128, 9)):
    m = synth.crs_laplace3d(g[0], g[1], g[2], np.float32)
    m.set_stream_xs(1)
    print(g, m.n_rows(), m.stream_layout(), m.stream_direct(), flush=True)
