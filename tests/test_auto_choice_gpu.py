"""AUTO (the variant selection of capi.hip: thresholds on create-time statistics) on shapes it was NOT tuned on: its
choice must be within 30 % of the fastest kernel family for the matrix (HIP events through the C ABI, median of 9
launches).  The shapes differ from BASELINE's in size, row length, band width and value type; the bound is deliberately
loose -- it catches a wrong family (typically 2-5x), not a tuning nuance."""
import ctypes as C

import numpy as np
import pytest

import sparsemat_amd as sm
from sparsemat_amd import _lib, synth

pytestmark = pytest.mark.gpu


def time_variant(m, xptr, n_x, yptr, variant, stream, reps=9):
    lib, check = sm.lib(), _lib.check
    a, b = C.c_void_p(), C.c_void_p()
    check(lib.smh_event_create(C.byref(a)))
    check(lib.smh_event_create(C.byref(b)))
    for _ in range(2):
        m.mvp_dev(xptr, n_x, yptr, variant, stream=stream)
    ts = []
    for _ in range(reps):
        check(lib.smh_event_record(a, C.c_void_p(stream)))
        m.mvp_dev(xptr, n_x, yptr, variant, stream=stream)
        check(lib.smh_event_record(b, C.c_void_p(stream)))
        ms = C.c_float()
        check(lib.smh_event_elapsed_ms(a, b, C.byref(ms)))
        ts.append(ms.value)
    lib.smh_event_destroy(a)
    lib.smh_event_destroy(b)
    return sorted(ts)[len(ts) // 2]


SHAPES = {
    "banded_f32_3M_x20": lambda: synth.crs_fixed(synth.SEED_MATRIX + 7, synth.PATTERN_BANDED, 3_000_000, 20, np.float32),
    "band_contiguous_f64_2M_x48": lambda: synth.crs_fixed(synth.SEED_MATRIX + 8, synth.PATTERN_DIAG, 2_000_000, 48, np.float64),
    "uniform_f32_6M_x12": lambda: synth.crs_fixed(synth.SEED_MATRIX + 9, synth.PATTERN_UNIFORM, 6_000_000, 12, np.float32),
    "laplace_f64_200": lambda: synth.crs_laplace3d(200, 200, 200, np.float64),
    "powerlaw_f32_3M": lambda: synth.crs_powerlaw(synth.SEED_MATRIX + 10, 3_000_000, 3_000_000, np.float32),
}


@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_auto_is_close_to_the_best_family(gpu, shape):
    m = SHAPES[shape]()
    n = m.n_cols()
    dtype = m.dtype
    xbuf, xptr = synth.gen_x(synth.SEED_X, n, dtype)
    ybuf = synth.DeviceBuffer(m.n_rows() * np.dtype(dtype).itemsize)
    s = C.c_void_p()
    _lib.check(sm.lib().smh_stream_create(C.byref(s)))
    families = ["vector", "merge", "stream"]
    if n * np.dtype(dtype).itemsize >= 4 << 20:
        families += ["colblock", "colfused", "tiled"]
    times = {}
    for v in families + ["auto"]:
        m.prepare(v)
        times[v] = time_variant(m, xptr, n, ybuf.ptr, v, s.value)
    sm.lib().smh_stream_destroy(s)
    best = min(times[v] for v in families)
    print("AUTO choice %s on %s: %s" % (m.resolved_variant(), shape, {k: round(v, 4) for k, v in times.items()}))
    assert times["auto"] <= 1.30 * best, (shape, m.resolved_variant(), times)
