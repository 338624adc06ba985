"""SparseMatPar: row-block partitioned SpMV across the GPUs of one node.

Mirrors the reference's ``SparseMatPar<M>`` (sparsemat_par.rs:12-35, 71-140): ``n_blocks``
sub-matrices of ``R = max_n_rows / n_blocks`` local rows each (:21), LOCAL row ids and GLOBAL
column ids, every block multiplied against the full right-hand side and the results placed at
``b * R`` -- the design its commented-out ``mvp_par`` sketches (:37-68).  Here: one process per
GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI), block b lives on rank b, the local
SpMV is the HIP kernel, and the one exchange step is an all-gather of the y slices so that every
rank holds the full vector for the next product.

Deviation (documented in DESIGN.md): the reference's ``get_block_and_row_id`` clamps the block
id to ``n_blocks`` (an out-of-bounds index, sparsemat_par.rs:32) and so cannot address rows
``>= n_blocks*R``; here the remainder rows belong to the LAST block.

The partition arithmetic and the gather layout are pure host logic, independent of where the
local product runs; ``local`` is any object with ``n_rows`` and ``mvp_into(x, y)``.
``HipBlock`` (the product) wraps a device-resident ``SparseMatCRS``; the CPU/gloo tests plug in
their own checker block.
"""
import numpy as np


def rows_per_block(n_blocks, max_n_rows):
    """R = max_n_rows / n_blocks (integer division), sparsemat_par.rs:21."""
    if n_blocks <= 0:
        raise ValueError("n_blocks must be positive")
    return max_n_rows // n_blocks


def block_range(n_blocks, n_rows, b):
    """Global rows [begin, end) of block b: b*R .. (b+1)*R, the last block takes the remainder."""
    r = rows_per_block(n_blocks, n_rows)
    begin = b * r
    end = n_rows if b == n_blocks - 1 else (b + 1) * r
    return begin, end


def block_and_row(n_blocks, n_rows, row):
    """(block, local row) of a global row -- get_block_and_row_id (sparsemat_par.rs:31-35) with
    the clamp moved to n_blocks-1."""
    r = rows_per_block(n_blocks, n_rows)
    b = min(row // r, n_blocks - 1) if r > 0 else n_blocks - 1
    return b, row - b * r


def split_crs(n_rows, offset_rows, columns, values, n_blocks, b):
    """Block b of a global CRS as an independent CRS: offsets rebased to 0 (each sub-matrix is
    its own SparseMatCRS, sparsemat_par.rs:15,22), columns stay global."""
    begin, end = block_range(n_blocks, n_rows, b)
    off = np.asarray(offset_rows)
    lo, hi = int(off[begin]), int(off[end])
    local_off = (off[begin:end + 1].astype(np.int64) - lo).astype(np.uint32)
    return local_off, np.asarray(columns)[lo:hi], np.asarray(values)[lo:hi]


class HipBlock:
    """One rank's sub-matrix on its GPU: a SparseMatCRS plus torch-tensor plumbing."""

    def __init__(self, mat, variant="auto"):
        self.mat = mat
        self.variant = variant
        self.n_rows = mat.n_rows()

    def mvp_into(self, x, y):
        import torch
        assert x.is_cuda and y.is_cuda and y.numel() == self.n_rows
        self.mat.mvp_dev(x.data_ptr(), x.numel(), y.data_ptr(), self.variant,
                         stream=torch.cuda.current_stream().cuda_stream)


class SparseMatPar:
    """``SparseMatPar::with_sub_matrices(n_blocks, max_n_rows)`` for ``n_blocks`` = world size."""

    def __init__(self, n_blocks, n_rows, n_cols, rank, local, group=None):
        self.n_blocks, self._n_rows, self._n_cols, self.rank = n_blocks, n_rows, n_cols, rank
        self.local = local
        self.group = group
        self.n_rows_sub_matrix = rows_per_block(n_blocks, n_rows)
        self.begin, self.end = block_range(n_blocks, n_rows, rank)
        if local.n_rows != self.end - self.begin:
            raise ValueError("local block has %d rows, partition expects %d" % (local.n_rows, self.end - self.begin))
        last_begin, last_end = block_range(n_blocks, n_rows, n_blocks - 1)
        self._pad = last_end - last_begin  # rows of the largest (= last) block
        self._ragged = self._pad != self.n_rows_sub_matrix
        self._gather_buf = None
        self._y_local = None

    @classmethod
    def with_sub_matrices(cls, n_blocks, max_n_rows, n_cols, rank, local, group=None):
        return cls(n_blocks, max_n_rows, n_cols, rank, local, group)

    def n_rows(self):
        return self._n_rows

    def n_cols(self):
        return self._n_cols

    def get_block_and_row_id(self, row):
        return block_and_row(self.n_blocks, self._n_rows, row)

    def mvp_local(self, x, y_local):
        """This rank's slice: y[begin:end] = A_b . x (no communication)."""
        self.local.mvp_into(x, y_local)

    def mvp(self, x, out=None):
        """y = A.x on every rank: local SpMV + all-gather of the slices into the full vector."""
        import torch
        import torch.distributed as dist
        if out is None:
            out = torch.empty(self._n_rows, dtype=x.dtype, device=x.device)
        if self.n_blocks == 1:
            self.local.mvp_into(x, out)
            return out
        if not self._ragged:
            # in-place layout: rank b's slice already sits at b*R of the gathered vector
            self.local.mvp_into(x, out[self.begin:self.end])
            dist.all_gather_into_tensor(out, out[self.begin:self.end], group=self.group)
            return out
        # ragged last block: gather equal, padded counts, then compact
        if self._gather_buf is None or self._gather_buf.dtype != x.dtype:
            self._gather_buf = torch.zeros(self.n_blocks * self._pad, dtype=x.dtype, device=x.device)
            self._y_local = torch.zeros(self._pad, dtype=x.dtype, device=x.device)
        self.local.mvp_into(x, self._y_local[:self.end - self.begin])
        dist.all_gather_into_tensor(self._gather_buf, self._y_local, group=self.group)
        r = self.n_rows_sub_matrix
        for b in range(self.n_blocks):
            bb, be = block_range(self.n_blocks, self._n_rows, b)
            out[bb:be] = self._gather_buf[b * self._pad:b * self._pad + (be - bb)]
        return out
