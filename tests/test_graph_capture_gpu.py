"""smh_crs_prepare + smh_crs_spmv_dev inside a caller-owned hipGraph (captured through torch's stream): after
`prepare` no launch of any kernel family allocates or synchronises, and the replayed product equals the direct one
bit for bit (every kernel is deterministic)."""
import numpy as np
import pytest
import torch

import sparsemat_amd as sm
from sparsemat_amd import synth
from util import assert_spmv_close, random_crs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", ["vector", "merge", "stream", "colblock", "colfused", "colsplit", "tiled", "seq"])
def test_spmv_dev_replays_from_a_captured_graph(gpu, variant):
    rng = np.random.default_rng(31)
    n_rows, n_cols = 20_011, 17_003
    off, col, val = random_crs(rng, n_rows, n_cols, rng.integers(0, 40, n_rows), np.float32)
    m = sm.SparseMatCRS.from_raw_parts(n_rows, n_cols, off, col, val)
    if variant in ("colblock", "colfused", "colsplit"):
        m.set_colblock_shift(12)  # 5 column blocks
    m.prepare(variant)
    x = torch.from_numpy(rng.uniform(-1, 1, n_cols).astype(np.float32)).cuda()
    y = torch.zeros(n_rows, dtype=torch.float32, device="cuda")
    y_direct = torch.zeros_like(y)
    m.mvp_dev(x.data_ptr(), n_cols, y_direct.data_ptr(), variant, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            m.mvp_dev(x.data_ptr(), n_cols, y.data_ptr(), variant, stream=side.cuda_stream)
    for scale in (1.0, -2.0):  # replay twice with different x contents
        x.mul_(scale)
        y.zero_()
        g.replay()
        torch.cuda.synchronize()
        m.mvp_dev(x.data_ptr(), n_cols, y_direct.data_ptr(), variant, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(y, y_direct)
        assert_spmv_close(y.cpu().numpy(), off, col, val, x.cpu().numpy(), variant)


def test_prepare_on_the_headline_shapes(gpu):
    for pattern, expect in ((synth.PATTERN_BANDED, "vector"), (synth.PATTERN_UNIFORM, "tiled")):
        m = synth.crs_fixed(synth.SEED_MATRIX, pattern, 2_000_000, 16, np.float32)
        assert m.resolved_variant()[0] == expect
        m.prepare()
        if expect == "tiled":  # f32 rows dense enough for ~48 entries per tile: the two streaming passes (K2t)
            assert m.tiled_layout()["n_slices"] == 123 and m.colfused(arrays=False)["n_blocks"] == 8
