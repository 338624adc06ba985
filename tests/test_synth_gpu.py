"""Device generators vs the oracle's independent host generators: bit-exact (integer structure and
exactly representable values), plus full-size (BASELINE C2) property checks of the SpMV kernels."""
import numpy as np
import pytest

import oracle
import sparsemat_amd as sm
from sparsemat_amd import synth
from util import assert_spmv_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("pattern", [synth.PATTERN_BANDED, synth.PATTERN_UNIFORM, synth.PATTERN_WINDOW], ids=["banded", "uniform", "window"])
@pytest.mark.parametrize("n,k,rb,re", [(5000, 32, 0, 5000), (40, 32, 0, 40), (100_000, 32, 30_000, 41_234), (9000, 7, 8000, 9000)])
def test_fixed_generator_bit_exact(gpu, dtype, pattern, n, k, rb, re):
    m = synth.crs_fixed(synth.SEED_MATRIX, pattern, n, k, dtype, rb, re)
    off, col, val = m.raw_parts()
    off_r, col_r, val_r = oracle.gen_fixed(synth.SEED_MATRIX, pattern, n, k, dtype, rb, re)
    assert np.array_equal(off, off_r) and np.array_equal(col, col_r)
    assert np.array_equal(val.view(np.uint8), val_r.view(np.uint8))
    assert col.max() < n
    if pattern in (synth.PATTERN_BANDED, synth.PATTERN_WINDOW):  # ascending and distinct inside every row
        c = col.reshape(-1, k).astype(np.int64)
        assert np.all(np.diff(c, axis=1) > 0)
    if pattern == synth.PATTERN_WINDOW:
        assert np.all(np.abs(c - np.arange(rb, re)[:, None]) <= 4096)
        with pytest.raises(sm.SparseMatPanic):
            synth.crs_fixed(synth.SEED_MATRIX, pattern, 1000, 65, dtype)


def test_x_and_powerlaw_and_laplace_generators_bit_exact(gpu):
    for dtype in (np.float32, np.float64):
        buf, _ = synth.gen_x(synth.SEED_X, 10_007, dtype, begin=123)
        x = buf.download(dtype, 10_007)
        assert np.array_equal(x.view(np.uint8), oracle.gen_x(synth.SEED_X, 10_007, dtype, begin=123).view(np.uint8))
        assert x.min() >= -1 and x.max() < 1
    m = synth.crs_powerlaw(synth.SEED_MATRIX, 30_000, 30_000, np.float64)
    off, col, val = m.raw_parts()
    off_r, col_r, val_r = oracle.gen_powerlaw(synth.SEED_MATRIX, 30_000, 30_000, np.float64)
    assert np.array_equal(off, off_r) and np.array_equal(col, col_r) and np.array_equal(val, val_r)
    lens = np.diff(off.astype(np.int64))
    assert lens.min() >= 1 and lens.max() <= 2048 and 20 < lens.mean() < 45
    assert m.resolved_variant()[0] == "merge"
    m = synth.crs_laplace3d(11, 7, 5, np.float32, 13, 300)
    off, col, val = m.raw_parts()
    off_f, col_f, val_f = oracle.laplace3d(11, 7, 5, np.float32)
    lo, hi = off_f[13], off_f[300]
    assert np.array_equal(off, off_f[13:301] - lo) and np.array_equal(col, col_f[lo:hi]) and np.array_equal(val, val_f[lo:hi])


@pytest.mark.parametrize("dtype,n_rows", [(np.float64, 200_000), (np.float32, 150_000)], ids=["f64", "f32"])
def test_powerlaw_merge_parity(gpu, dtype, n_rows):
    """BASELINE C3 shape at a size the oracle finishes in seconds."""
    m = synth.crs_powerlaw(synth.SEED_MATRIX, n_rows, n_rows, dtype)
    off, col, val = m.raw_parts()
    x = oracle.gen_x(synth.SEED_X, n_rows, dtype)
    for variant in ("merge", "vector", "auto"):
        assert_spmv_close(m.mvp(x, variant=variant), off, col, val, x, variant)
    assert np.array_equal(m.mvp(x, variant="seq").view(np.uint8), oracle.spmv(off, col, val, x).view(np.uint8))


def test_full_size_c2_properties(gpu):
    """BASELINE C2 (10M rows x 32, f32) at full size through size-independent properties:
    sampled row blocks vs the oracle (which regenerates those rows itself), SEQ bit-exact on the
    samples, the two independent kernels agree everywhere, and linearity."""
    n, k = 10_000_000, 32
    m = synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, n, k, np.float32)
    assert m.n_non_zero_entries() == n * k and m.resolved_variant() == ("vector", 8)
    xbuf, xptr = synth.gen_x(synth.SEED_X, n, np.float32)
    x = xbuf.download(np.float32, n)
    ys = {}
    for variant in ("vector", "merge", "seq"):
        ybuf = synth.DeviceBuffer(n * 4)
        m.mvp_dev(xptr, n, ybuf.ptr, variant)
        sm.lib().smh_device_synchronize()
        ys[variant] = ybuf.download(np.float32, n)
    # (a) sampled row blocks against the oracle on rows it regenerates independently
    for rb in (0, 1, 4095, 4_999_000, n - 3000):
        re = min(n, rb + 3000)
        off, col, val = oracle.gen_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, n, k, np.float32, rb, re)
        y_ref = oracle.spmv(off, col, val, x)
        assert np.array_equal(ys["seq"][rb:re].view(np.uint32), y_ref.view(np.uint32))
        assert_spmv_close(ys["vector"][rb:re], off, col, val, x, "vector rows %d.." % rb)
        assert_spmv_close(ys["merge"][rb:re], off, col, val, x, "merge rows %d.." % rb)
    # (b) independent kernels agree on every row (|y| <= 32, sum|a x| ~ 8)
    assert np.abs(ys["vector"].astype(np.float64) - ys["seq"]).max() < 5e-5
    assert np.abs(ys["merge"].astype(np.float64) - ys["seq"]).max() < 5e-5
    # (c) linearity: A(2x) == 2 A x exactly (power-of-two scaling commutes with every rounding)
    x2 = synth.DeviceBuffer(n * 4)
    x2.upload(x * np.float32(2))
    y2 = synth.DeviceBuffer(n * 4)
    m.mvp_dev(x2.ptr, n, y2.ptr, "vector")
    sm.lib().smh_device_synchronize()
    assert np.array_equal(y2.download(np.float32, n), ys["vector"] * np.float32(2))


def _run(m, xptr, n_x, variant, dtype):
    ybuf = synth.DeviceBuffer(m.n_rows() * np.dtype(dtype).itemsize)
    m.mvp_dev(xptr, n_x, ybuf.ptr, variant)
    sm.lib().smh_device_synchronize()
    return ybuf.download(dtype, m.n_rows())


def test_full_size_c2_window_without_replacement(gpu):
    """SURVEY 8(d)'s primary pattern to the letter -- 32 distinct columns drawn WITHOUT REPLACEMENT from [i - 4096, i + 4096],
    ascending -- at BASELINE C2's size (the headline generator is the stratified subset of it).  AUTO takes the same ring
    kernel (a 64-row tile references 8 256 columns: inside the 16 384-column ring); sampled row blocks against the oracle
    under the north-star bound, the bit-exact kernels bit for bit, every kernel against every other on all rows."""
    n, k = 10_000_000, 32
    m = synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_WINDOW, n, k, np.float32)
    assert m.n_non_zero_entries() == n * k
    assert m.resolved_variant()[0] == "vector"
    _, ring_frac, ring_active, _, _ = m.ring_plan()
    assert ring_active and ring_frac == 1.0
    xbuf, xptr = synth.gen_x(synth.SEED_X, n, np.float32)
    x = xbuf.download(np.float32, n)
    y_auto = _run(m, xptr, n, "auto", np.float32)
    y_st = _run(m, xptr, n, "stream", np.float32)
    y_mg = _run(m, xptr, n, "merge", np.float32)
    for rb in (0, 4_999_000, n - 3000):
        re = min(n, rb + 3000)
        off, col, val = oracle.gen_fixed(synth.SEED_MATRIX, synth.PATTERN_WINDOW, n, k, np.float32, rb, re)
        assert np.array_equal(y_st[rb:re].view(np.uint32), oracle.spmv(off, col, val, x).view(np.uint32))
        assert_spmv_close(y_auto[rb:re], off, col, val, x, "K1r rows %d.." % rb)
        assert_spmv_close(y_mg[rb:re], off, col, val, x, "merge rows %d.." % rb)
    assert np.abs(y_auto.astype(np.float64) - y_st).max() < 5e-5
    assert np.abs(y_mg.astype(np.float64) - y_st).max() < 5e-5


def test_more_than_2_to_the_31_entries_in_one_matrix(gpu):
    """Maximum sizes: 70 M rows x 32 = 2.24e9 entries in ONE handle (> 2^31, below the reference's limit of u32::MAX,
    sparsemat_crs.rs:82-84) -- the entry count of BASELINE C5 on a single GPU.  Every 32-bit index in the kernels
    must be unsigned: sampled row blocks (incl. the rows whose entries straddle 2^31) against the oracle, bit-exact
    for K1s and SEQ, tolerance for K1r and merge-path; the kernels agree on every row."""
    n, k = 70_000_000, 32
    m = synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, n, k, np.float32)
    assert m.n_non_zero_entries() == n * k > 2 ** 31
    xbuf, xptr = synth.gen_x(synth.SEED_X, n, np.float32)
    x = xbuf.download(np.float32, n)
    y_ring = _run(m, xptr, n, "auto", np.float32)
    y_st = _run(m, xptr, n, "stream", np.float32)
    y_mg = _run(m, xptr, n, "merge", np.float32)
    y_seq = _run(m, xptr, n, "seq", np.float32)
    r31 = (2 ** 31) // k  # the row holding entry 2^31
    for rb in (0, r31 - 1500, n - 3000):
        re = min(n, rb + 3000)
        off, col, val = oracle.gen_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, n, k, np.float32, rb, re)
        y_ref = oracle.spmv(off, col, val, x)
        assert np.array_equal(y_st[rb:re].view(np.uint32), y_ref.view(np.uint32))
        assert np.array_equal(y_seq[rb:re].view(np.uint32), y_ref.view(np.uint32))
        assert_spmv_close(y_ring[rb:re], off, col, val, x, "K1r rows %d.." % rb)
        assert_spmv_close(y_mg[rb:re], off, col, val, x, "merge rows %d.." % rb)
    assert np.array_equal(y_st.view(np.uint32), y_seq.view(np.uint32))
    assert np.abs(y_ring.astype(np.float64) - y_st).max() < 5e-5
    assert np.abs(y_mg.astype(np.float64) - y_st).max() < 5e-5


def test_entry_count_just_below_u32_max(gpu):
    """The largest matrix the reference can hold (sparsemat_crs.rs:82-84: fewer than u32::MAX entries): 134 217 000
    rows x 32 = 4 294 944 000 entries (43 GB of arrays) in one handle; the last rows' offsets sit 23 295 below 2^32.
    Sampled row blocks at both ends against the oracle, all kernels agree on every row; one more row of 32 entries
    would still fit, 729 more rows are refused with the reference's panic text."""
    n, k = 134_217_000, 32
    m = synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, n, k, np.float32)
    assert m.n_non_zero_entries() == n * k and 2 ** 32 - 1 - n * k == 23_295
    xbuf, xptr = synth.gen_x(synth.SEED_X, n, np.float32)
    x = xbuf.download(np.float32, n)
    y_ring = _run(m, xptr, n, "auto", np.float32)
    y_st = _run(m, xptr, n, "stream", np.float32)
    y_mg = _run(m, xptr, n, "merge", np.float32)
    for rb in (0, n // 2, n - 2500):
        re = min(n, rb + 2500)
        off, col, val = oracle.gen_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, n, k, np.float32, rb, re)
        y_ref = oracle.spmv(off, col, val, x)
        assert np.array_equal(y_st[rb:re].view(np.uint32), y_ref.view(np.uint32))
        assert_spmv_close(y_ring[rb:re], off, col, val, x, "K1r rows %d.." % rb)
        assert_spmv_close(y_mg[rb:re], off, col, val, x, "merge rows %d.." % rb)
    assert np.abs(y_ring.astype(np.float64) - y_st).max() < 5e-5
    assert np.abs(y_mg.astype(np.float64) - y_st).max() < 5e-5
    del m
    with pytest.raises(sm.SparseMatPanic) as e:
        synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, n + 729, k, np.float32)
    assert "Maximum number of 4294967295 entries reached" in str(e.value)


def test_full_size_c2_uniform_colblock_properties(gpu):
    """C2 stress variant (10M rows x 32 uniform columns, f32) at full size: AUTO = the 2-D tiled passes (K2t); sampled row
    blocks against the oracle (rows regenerated independently), the bit-exact K1s/SEQ pair agrees with it on the
    samples bit for bit, K2t and K2f agree with K1s on every row within the parity bound, linearity under x -> 2x."""
    n, k = 10_000_000, 32
    m = synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_UNIFORM, n, k, np.float32)
    assert m.resolved_variant()[0] == "tiled"  # two streaming passes over the 2-D tiled copy (K2t); before it: K2f
    lay = m.tiled_layout()
    # ~174 products per (slice, row block) tile; one product per (row, slice) pair: ~2.5 % of the entries fold into a neighbour
    assert lay["n_slices"] == 611 and lay["rows_per_block"] <= 3328 and 140 <= lay["n_products"] / lay["n_slices"] / lay["n_row_blocks"] <= 200
    assert 0.95 * n * k < lay["n_products"] < 1.01 * n * k
    cf = m.colfused(arrays=False)
    assert cf["fits"] and cf["n_blocks"] == 39 and cf["shift"] == 18
    xbuf, xptr = synth.gen_x(synth.SEED_X, n, np.float32)
    x = xbuf.download(np.float32, n)
    y_cb = _run(m, xptr, n, "auto", np.float32)
    y_st = _run(m, xptr, n, "stream", np.float32)
    y_k2f = _run(m, xptr, n, "colfused", np.float32)
    assert np.abs(y_k2f.astype(np.float64) - y_st).max() < 5e-5
    for rb in (0, 2047, 5_000_000, n - 2500):
        re = min(n, rb + 2500)
        off, col, val = oracle.gen_fixed(synth.SEED_MATRIX, synth.PATTERN_UNIFORM, n, k, np.float32, rb, re)
        y_ref = oracle.spmv(off, col, val, x)
        assert np.array_equal(y_st[rb:re].view(np.uint32), y_ref.view(np.uint32))  # multi-pass K1s: bit-exact
        assert_spmv_close(y_cb[rb:re], off, col, val, x, "colblock rows %d.." % rb)
    assert np.abs(y_cb.astype(np.float64) - y_st).max() < 5e-5  # sum|a x| ~ 8 per row
    x2 = synth.DeviceBuffer(n * 4)
    x2.upload(x * np.float32(2))
    assert np.array_equal(_run(m, x2.ptr, n, "auto", np.float32), y_cb * np.float32(2))


def test_full_size_c3_powerlaw_properties(gpu):
    """BASELINE C3 (f64, 10M rows, power-law lengths 1..2048, uniform columns) at full size: K2t (AUTO), the column-blocked
    K2c / K2f / K2s, merge-path K2 and the bit-exact K1s agree on every row to 1e-12 of the row's scale; sampled rows against
    the oracle."""
    n = 10_000_000
    m = synth.crs_powerlaw(synth.SEED_MATRIX, n, n, np.float64)
    assert m.resolved_variant()[0] == "tiled" and m.max_row_len() == 2048  # K2t: row blocks cut by entries; before it K2s, below
    lay = m.tiled_layout()
    assert lay["n_slices"] == 611 and lay["rows_per_block"] <= 1664 and 40 <= lay["n_products"] / 611 / lay["n_row_blocks"] <= 70
    # the entries of a long row that share a slice are folded inside pass 1: a third fewer products than entries
    assert 0.55 * m.n_non_zero_entries() < lay["n_products"] < 0.80 * m.n_non_zero_entries()
    cs_flag = m.colsplit()
    assert cs_flag["split"] and 700_000 < cs_flag["n_long"] < 800_000 and cs_flag["long_variant"] == "colblock" and cs_flag["short_variant"] == "colfused"
    xbuf, xptr = synth.gen_x(synth.SEED_X, n, np.float64)
    x = xbuf.download(np.float64, n)
    y_cb = _run(m, xptr, n, "auto", np.float64)
    y_mg = _run(m, xptr, n, "merge", np.float64)
    y_st = _run(m, xptr, n, "stream", np.float64)
    y_k2f = _run(m, xptr, n, "colfused", np.float64)  # the one-sweep form and the per-block launches must agree too
    y_k2c = _run(m, xptr, n, "colblock", np.float64)
    y_k2s = _run(m, xptr, n, "colsplit", np.float64)  # taken apart by row length (AUTO's choice before K2t)
    assert np.abs(y_k2f - y_st).max() < 2.1e-9 and np.abs(y_k2c - y_st).max() < 2.1e-9 and np.abs(y_k2s - y_st).max() < 2.1e-9
    # |row| <= 2048 entries of magnitude < 1: sum|a x| <= 2048; bound 1e-12 * 2048 covers every row
    assert np.abs(y_cb - y_st).max() < 2.1e-9 and np.abs(y_mg - y_st).max() < 2.1e-9
    off_all = synth.powerlaw_offsets(synth.SEED_MATRIX, n)
    for rb in (0, 3_333_333, n - 4000):
        re = min(n, rb + 4000)
        off, col, val = oracle.gen_powerlaw(synth.SEED_MATRIX, n, n, np.float64, row_begin=rb, row_end=re)
        assert np.array_equal(off, (off_all[rb:re + 1].astype(np.int64) - int(off_all[rb])).astype(np.uint32))
        y_ref = oracle.spmv(off, col, val, x)
        assert np.array_equal(y_st[rb:re].view(np.uint64), y_ref.view(np.uint64))
        assert_spmv_close(y_cb[rb:re], off, col, val, x, "colblock rows %d.." % rb)
        assert_spmv_close(y_mg[rb:re], off, col, val, x, "merge rows %d.." % rb)


def test_full_size_c4_laplacian_512_spmv_and_cg(gpu):
    """BASELINE configs[3] at its size: the 7-point Laplacian on a 512^3 grid (134,217,728 rows, 937,951,232 entries, f32).
    SpMV: AUTO = K1s with 16-bit column codes and byte row lengths; it must equal the SEQ kernel (one lane per row, the
    reference's loop) on EVERY row bit for bit, and the oracle on sampled row blocks.  CG (linearsolver.rs:27-61): three
    iterations from x0 = 0, b = A.1.  Iteration 1 is exact on both sides (b is integer valued and its sums stay below
    2^24): bit-identical x.  From then on the reference's SEQUENTIAL f32 folds over 1.3e8 elements (vector.rs:50-58) are
    themselves inaccurate at this size -- partial sums beyond 2^24 swallow the low bits of every further term -- so its
    alpha / beta drift from the exact quotients by ~1e-3, while the device folds with a fixed tree.  As for the SpMV rows
    (tests/util.py), the device is therefore measured against an (effectively) exact restatement -- the same recurrence, the
    same f32 element-wise roundings, the two dot products per iteration accumulated in f64 -- and must be no further from
    it than the oracle is: measured in r02, device 2.1e-6 of max|x| against the oracle's 2.9e-3."""
    g = 512
    n = g ** 3
    m = synth.crs_laplace3d(g, g, g, np.float32)
    assert (m.n_rows(), m.n_non_zero_entries()) == (n, 937_951_232) and m.resolved_variant()[0] == "stream"
    xbuf, xptr = synth.gen_x(synth.SEED_X, n, np.float32)
    y_auto = _run(m, xptr, n, "auto", np.float32)
    y_seq = _run(m, xptr, n, "seq", np.float32)
    assert np.array_equal(y_auto.view(np.uint32), y_seq.view(np.uint32))  # all 134 M rows, bit for bit
    del y_seq
    off, col, val = oracle.laplace3d(g, g, g, np.float32)  # the oracle's own generator (host, ~8 GB)
    x = xbuf.download(np.float32, n)
    for rb in (0, g * g - 300, n // 2 + 12345, n - 70_000):
        re = min(n, rb + 70_000)
        y_ref = oracle.spmv(off, col, val, x, rows=(rb, re))
        assert np.array_equal(y_auto[rb:re].view(np.uint32), y_ref[rb:re].view(np.uint32)), rb
    del y_auto, x, xbuf
    b = oracle.spmv(off, col, val, np.ones(n, np.float32))
    x_ref, it_ref, rr_ref = oracle.cg(n, n, off, col, val, b, np.zeros(n, np.float32), tol=1e-12, iter_max=3)
    xd = np.zeros(n, np.float32)
    cg = sm.ConjugateGradient(1e-12, 3)
    cg.solve(m, b, xd)
    assert cg.iterations == it_ref == 3

    def dot64(u, v):  # blockwise f64 accumulation (no 1 GB temporaries)
        return float(sum(np.dot(u[i:i + (1 << 24)].astype(np.float64), v[i:i + (1 << 24)].astype(np.float64)) for i in range(0, n, 1 << 24)))

    xe, r = np.zeros(n, np.float32), b.copy()   # r = b - A.0
    p = r.copy()
    rr = dot64(r, r)
    for _ in range(3):
        ap = oracle.spmv(off, col, val, p)       # storage-order f32 rows: what K1s computes bit for bit
        alpha = np.float32(rr) / np.float32(dot64(p, ap))
        xe += p * alpha                          # round(p * alpha), then add   (linearsolver.rs:47)
        r -= ap * alpha                          #                               (:49)
        rr_new = dot64(r, r)
        beta = np.float32(rr_new) / np.float32(rr)
        p *= beta                                # p.scale(beta); p.add(&r)      (:58-59)
        p += r
        rr = rr_new
    scale = float(np.abs(xe).max())
    d_dev = float(np.abs(xd.astype(np.float64) - xe).max()) / scale
    d_orc = float(np.abs(x_ref.astype(np.float64) - xe).max()) / scale
    print("C4 full size, 3 CG iterations: max |x - x_exactdots| / max|x|: device %.3g, oracle %.3g; rr device %.6g oracle %.6g exact %.6g"
          % (d_dev, d_orc, cg.r_norm_squared, rr_ref, rr))
    assert d_dev <= d_orc + 1e-5 and d_dev < 1e-4
    assert abs(cg.r_norm_squared - rr) <= 1e-4 * rr
    # one iteration is exact on both sides (see the docstring): bit-identical x
    x1_ref, _, _ = oracle.cg(n, n, off, col, val, b, np.zeros(n, np.float32), tol=1e-12, iter_max=1)
    x1 = np.zeros(n, np.float32)
    sm.ConjugateGradient(1e-12, 1).solve(m, b, x1)
    assert np.array_equal(x1.view(np.uint32), x1_ref.view(np.uint32))
