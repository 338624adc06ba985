"""Randomised sweep over shapes near every tiling boundary of the kernels (64-row wave tiles, 256-row K1s tiles,
2048-item merge tiles, 4-entry chunks, borrowed/unpadded arrays), all kernel families, both dtypes."""
import numpy as np
import pytest

import oracle
import sparsemat_amd as sm
from sparsemat_amd import synth
from util import assert_spmv_close

pytestmark = pytest.mark.gpu

ROWS = [1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1000, 2047, 2048, 2049, 4100]


def _matrix(rng, n_rows, dtype):
    n_cols = int(rng.integers(1, 3 * n_rows + 5))
    style = rng.integers(0, 5)
    if style == 0:
        lens = rng.integers(0, 6, n_rows)
    elif style == 1:
        lens = rng.integers(0, 40, n_rows)
    elif style == 2:
        lens = np.full(n_rows, int(rng.integers(1, 34)))
    elif style == 3:
        lens = rng.integers(0, 4, n_rows)
        lens[rng.integers(0, n_rows)] = int(rng.integers(100, 5000))
    else:
        lens = (rng.pareto(1.2, n_rows) * 3).astype(np.int64).clip(0, 3000)
    off = np.zeros(n_rows + 1, dtype=np.uint32)
    np.cumsum(lens, out=off[1:])
    nnz = int(off[-1])
    if rng.random() < 0.5:  # banded-ish
        rows = np.repeat(np.arange(n_rows), lens)
        col = np.clip(rows * n_cols // max(n_rows, 1) + rng.integers(-50, 51, nnz), 0, n_cols - 1).astype(np.uint32)
    else:
        col = rng.integers(0, n_cols, nnz, dtype=np.uint32)
    val = rng.uniform(-1, 1, nnz).astype(dtype)
    x = rng.uniform(-1, 1, n_cols).astype(dtype)
    return n_cols, off, col, val, x


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_fuzz_all_kernels(gpu, dtype):
    rng = np.random.default_rng(20260101)
    bits = np.uint32 if dtype == np.float32 else np.uint64
    for trial, n_rows in enumerate(ROWS * 2):
        n_cols, off, col, val, x = _matrix(rng, n_rows, dtype)
        what = "trial %d: %d x %d, nnz %d" % (trial, n_rows, n_cols, len(val))
        y_ref = oracle.spmv(off, col, val, x)
        # owned (padded) copy and a borrowed, unpadded device copy of the same arrays
        owned = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
        bo, bc, bv = synth.DeviceBuffer(off.nbytes), synth.DeviceBuffer(max(col.nbytes, 4)), synth.DeviceBuffer(max(val.nbytes, 8))
        bo.upload(off)
        if len(val):
            bc.upload(col)
            bv.upload(val)
        borrowed = sm.SparseMatCRS.from_device_parts(n_rows, n_cols, len(val), bo.ptr, bc.ptr, bv.ptr, dtype, keep=(bo, bc, bv))
        for m, tag in ((owned, "owned"), (borrowed, "borrowed")):
            for variant in ("seq", "stream"):  # bit-exact kernels
                y = m.mvp(x, variant=variant)
                assert np.array_equal(y.view(bits), y_ref.view(bits)), (what, tag, variant)
            assert_spmv_close(m.mvp(x, variant="merge"), off, col, val, x, "%s %s merge" % (what, tag))
            assert_spmv_close(m.mvp(x, variant="auto"), off, col, val, x, "%s %s auto" % (what, tag))
            for lanes in (1, 4, 8, 32):
                m.set_vector_lanes(lanes)
                for ring in ((1, 0) if lanes <= 8 else (0,)):
                    m.set_ring(ring)
                    assert_spmv_close(m.mvp(x, variant="vector"), off, col, val, x,
                                      "%s %s vector lanes %d ring %d" % (what, tag, lanes, ring))
            m.set_vector_lanes(0)
            m.set_ring(-1)
            # column-blocked K2c with a random block width (1 .. ~30 blocks)
            shift = int(rng.integers(max(1, int(np.log2(max(n_cols, 2))) - 4), int(np.log2(max(n_cols, 2))) + 2))
            m.set_colblock_shift(shift)
            if (n_cols + (1 << shift) - 1) >> shift <= 128:
                assert_spmv_close(m.mvp(x, variant="colblock"), off, col, val, x, "%s %s colblock 2^%d" % (what, tag, shift))
            m.set_colblock_shift(0)
