"""TEST INFRASTRUCTURE (moved out of the product package in round 3): a Python / torch.distributed form of the row-block
partition, kept for the CPU (gloo, world size 2-3) tests of the plan arithmetic and of the exchange layout, and as the
independent restatement `smh_par_plan` (csrc/par.hip) is compared with.  The PRODUCT's partition is csrc/par.hip behind
`smh_par_*` (sparsemat_amd.SparseMatParLocal); nothing under sparsemat_amd/ imports this file.

SparseMatPar: row-block partitioned SpMV across the GPUs of one node.

Mirrors the reference's ``SparseMatPar<M>`` (sparsemat_par.rs:12-35, 71-140): ``n_blocks``
sub-matrices of ``R = max_n_rows / n_blocks`` local rows each (:21), LOCAL row ids and GLOBAL
column ids, every block multiplied against the right-hand side and the results placed at
``b * R`` -- the design its commented-out ``mvp_par`` sketches (:37-68).  Here: one process per
GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI), block b lives on rank b, the local
SpMV is the HIP kernel, and there is ONE exchange step per product:

* ``mvp``            -- all-gather of the y slices: every rank ends with the full vector (what the
                        reference's sketch returns).
* ``mvp_window``     -- every rank ends with its own slice plus exactly the part of the vector its
                        block REFERENCES (its column interval [cmin, cmax]) -- all a following
                        product needs.  For a banded matrix that is a few KiB from the two
                        neighbours instead of (N-1)/N of the vector over per-link-bound xGMI rings;
                        for a block that references everything it degenerates to the all-gather.
                        One ``all_to_all_single`` with split sizes (zero for non-neighbours).

Deviation (documented in DESIGN.md): the reference's ``get_block_and_row_id`` clamps the block
id to ``n_blocks`` (an out-of-bounds index, sparsemat_par.rs:32) and so cannot address rows
``>= n_blocks*R``; here the remainder rows belong to the LAST block.

The partition arithmetic, the exchange plan and the gather layout are pure host logic, independent
of where the local product runs; ``local`` is any object with ``n_rows``, ``col_range()`` and
``mvp_into(x, y)``.  ``HipBlock`` (the product) wraps a device-resident ``SparseMatCRS``; the
CPU/gloo tests plug in their own checker block.
"""
import os

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


def exchange_plan(n_blocks, n_rows, needs, rank):
    """Who sends what for ``mvp_window``.  ``needs[q] = (lo, hi)``: rank q references vector entries
    [lo, hi) (empty when lo >= hi).  Returns (send, recv): lists over peers q of the global range
    [a, b) this rank sends to q (a piece of ITS OWN slice) / receives from q (a piece of q's slice)."""
    my_b, my_e = block_range(n_blocks, n_rows, rank)
    send, recv = [], []
    for q in range(n_blocks):
        qb, qe = block_range(n_blocks, n_rows, q)
        if q == rank:
            send.append((0, 0))
            recv.append((0, 0))
            continue
        lo, hi = needs[q]
        a, b = max(lo, my_b), min(hi, my_e)  # what q needs of my slice
        send.append((a, b) if a < b else (0, 0))
        lo, hi = needs[rank]
        a, b = max(lo, qb), min(hi, qe)  # what I need of q's slice
        recv.append((a, b) if a < b else (0, 0))
    return send, recv


class HipBlock:
    """One rank's sub-matrix on its GPU: a SparseMatCRS plus torch-tensor plumbing."""

    def __init__(self, mat, variant="auto"):
        self.mat = mat
        self.variant = variant
        self.n_rows = mat.n_rows()

    def col_range(self):
        lo, hi = self.mat.col_range()
        return (lo, hi + 1) if self.mat.n_non_zero_entries() else (0, 0)

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
        self._plan = None  # (mode, send, recv, send_sizes, recv_sizes)
        # halo exchange as grouped point-to-point operations on slices of the vector itself (no staging); False: one
        # all_to_all_single through packed send / receive buffers (SMH_HALO_A2A=1 selects it)
        self.halo_p2p = os.environ.get("SMH_HALO_A2A", "0") != "1"

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

    # ---- exchange 1: all-gather of the slices (every rank gets the full vector) -----------------
    def allgather(self, v):
        """v holds this rank's slice at [begin,end): fill the rest from the other ranks."""
        import torch
        import torch.distributed as dist
        if self.n_blocks == 1:
            return v
        if not self._ragged:
            # in-place layout: rank b's slice already sits at b*R of the gathered vector
            dist.all_gather_into_tensor(v, v[self.begin:self.end], group=self.group)
            return v
        # ragged last block: gather equal, padded counts, then compact
        if self._gather_buf is None or self._gather_buf.dtype != v.dtype:
            self._gather_buf = torch.zeros(self.n_blocks * self._pad, dtype=v.dtype, device=v.device)
            self._y_local = torch.zeros(self._pad, dtype=v.dtype, device=v.device)
        self._y_local[:self.end - self.begin] = v[self.begin:self.end]
        dist.all_gather_into_tensor(self._gather_buf, self._y_local, group=self.group)
        for b in range(self.n_blocks):
            bb, be = block_range(self.n_blocks, self._n_rows, b)
            v[bb:be] = self._gather_buf[b * self._pad:b * self._pad + (be - bb)]
        return v

    def mvp(self, x, out=None):
        """y = A.x on every rank: local SpMV + all-gather of the slices into the full vector."""
        import torch
        if out is None:
            out = torch.empty(self._n_rows, dtype=x.dtype, device=x.device)
        self.local.mvp_into(x, out[self.begin:self.end])
        return self.allgather(out)

    # ---- exchange 2: only what each block references -------------------------------------------------
    def setup_window_exchange(self, like, mode="auto"):
        """Agree on the exchange plan (collective, once): every rank publishes the column interval its
        block references.  mode "auto": neighbour pieces when they are less than half of the vector,
        else the all-gather (a block that references everything gains nothing from a plan)."""
        import torch
        import torch.distributed as dist
        lo, hi = self.local.col_range()
        if self.n_blocks == 1:
            self._plan = ("none", None, None, None, None)
            return self._plan[0]
        mine = torch.tensor([lo, hi], dtype=torch.int64, device=like.device)
        everyone = torch.empty(2 * self.n_blocks, dtype=torch.int64, device=like.device)
        dist.all_gather_into_tensor(everyone, mine, group=self.group)
        needs = [(int(a), int(b)) for a, b in everyone.cpu().view(-1, 2).tolist()]
        send, recv = exchange_plan(self.n_blocks, self._n_rows, needs, self.rank)
        worst = 0
        for q in range(self.n_blocks):  # the plan is global: every rank computes the same decision
            _, rq = exchange_plan(self.n_blocks, self._n_rows, needs, q)
            worst = max(worst, sum(b - a for a, b in rq))
        use_halo = mode == "halo" or (mode == "auto" and worst * 2 < self._n_rows)
        self._plan = ("halo" if use_halo else "allgather", send, recv,
                      [b - a for a, b in send], [b - a for a, b in recv])
        return self._plan[0]

    def exchange_window(self, v):
        """v holds this rank's slice at [begin,end): fetch the referenced entries owned by other ranks."""
        import torch
        import torch.distributed as dist
        if self._plan is None:
            raise RuntimeError("call setup_window_exchange() first")
        mode, send, recv, send_sizes, recv_sizes = self._plan
        if mode == "none":
            return v
        if mode == "allgather":
            return self.allgather(v)
        if self.halo_p2p:
            # Zero-copy: every piece is a contiguous slice of v itself -- sends read this rank's slice, receives land in
            # the neighbours' slices (disjoint from it) -- so the exchange is ONE grouped launch of point-to-point
            # operations (ncclGroupStart/End under RCCL) with no staging buffer, no pack and no unpack kernel.
            ops = []
            for q, (a, b) in enumerate(recv):
                if a < b:
                    ops.append(dist.P2POp(dist.irecv, v[a:b], self._global_rank(q), group=self.group))
            for q, (a, b) in enumerate(send):
                if a < b:
                    ops.append(dist.P2POp(dist.isend, v[a:b], self._global_rank(q), group=self.group))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            return v
        pieces = [v[a:b] for (a, b) in send if a < b]
        sendbuf = torch.cat(pieces) if pieces else v.new_empty(0)
        recvbuf = v.new_empty(sum(recv_sizes))
        dist.all_to_all_single(recvbuf, sendbuf, output_split_sizes=recv_sizes, input_split_sizes=send_sizes,
                               group=self.group)
        pos = 0
        for (a, b) in recv:
            if a < b:
                v[a:b] = recvbuf[pos:pos + (b - a)]
                pos += b - a
        return v

    def _global_rank(self, q):
        import torch.distributed as dist
        return q if self.group is None else dist.get_global_rank(self.group, q)

    def mvp_window(self, x, out=None):
        """Local SpMV into this rank's slice of ``out`` + exchange of the referenced entries: afterwards
        ``out`` is valid on [begin,end) and on the block's column interval -- ready to be the next x."""
        import torch
        if out is None:
            out = torch.zeros(self._n_rows, dtype=x.dtype, device=x.device)
        self.local.mvp_into(x, out[self.begin:self.end])
        return self.exchange_window(out)


# ---- the row-partitioned CG recurrence on torch tensors (round 1's solver; the product's is smh_par_cg_solve) ----------------
import ctypes as C  # noqa: E402

from sparsemat_amd import _lib  # noqa: E402
from sparsemat_amd._lib import check, lib  # noqa: E402


class HipVectorOps:
    """DenseVec kernels of the library on torch CUDA tensors, scalars read from device memory."""

    def __init__(self, like):
        import torch
        self._torch = torch
        self.dtype_code = _lib.SMH_F64 if like.dtype == torch.float64 else _lib.SMH_F32
        self.scratch = torch.empty(lib().smh_blas_dot_scratch_bytes(), dtype=torch.uint8, device=like.device)

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream().cuda_stream)

    def dot(self, x, y, out):
        check(lib().smh_blas_dot_dev(self.dtype_code, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), x.numel(),
                                     C.c_void_p(out.data_ptr()), C.c_void_p(self.scratch.data_ptr()), self._stream()))

    def axpy(self, y, a, x):  # y += round(a*x)
        check(lib().smh_blas_axpy_dev(self.dtype_code, C.c_void_p(y.data_ptr()), C.c_void_p(a.data_ptr()),
                                      C.c_void_p(x.data_ptr()), y.numel(), self._stream()))

    def xpby(self, p, b, r):  # p = round(b*p) + r
        check(lib().smh_blas_xpby_dev(self.dtype_code, C.c_void_p(p.data_ptr()), C.c_void_p(b.data_ptr()),
                                      C.c_void_p(r.data_ptr()), p.numel(), self._stream()))


class ParConjugateGradient:
    """ConjugateGradient::solve (linearsolver.rs:27-61) over a SparseMatPar: rank b holds rows
    [b*R, (b+1)*R) of A and the matching slices of b and x.  Same update order and roundings as the
    reference; the two reductions are local deterministic trees + an all-reduce (sum) over the ranks."""

    def __init__(self, tol=1e-12, iter_max=10_000, ops=None):
        self.tol, self.iter_max, self.ops = float(tol), int(iter_max), ops
        self.iterations = None
        self.r_norm_squared = None

    def solve(self, par, b_local, x_local):
        import torch
        import torch.distributed as dist
        if par.n_rows() != par.n_cols():
            raise _lib.SparseMatPanic(_lib.SMH_ERR_NOT_SQUARE, "Matrix is not symmetric")            # :30-32
        rows = par.end - par.begin
        if b_local.numel() != rows or x_local.numel() != rows:
            raise _lib.SparseMatPanic(_lib.SMH_ERR_DIM_MISMATCH, "Matrix and vector size mismatch")  # :33-36
        ops = self.ops or HipVectorOps(x_local)
        multi = par.n_blocks > 1
        if par._plan is None:
            par.setup_window_exchange(x_local)

        def allreduce(t):
            if multi:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=par.group)

        n = par.n_rows()
        dev, dt = x_local.device, x_local.dtype
        full = torch.zeros(n, dtype=dt, device=dev)        # holds x, then p: own slice + exchanged window
        mine = full[par.begin:par.end]
        ap = torch.empty(rows, dtype=dt, device=dev)
        rr, rr_prev, pap, alpha, neg_alpha, beta = (torch.zeros(1, dtype=dt, device=dev) for _ in range(6))
        # r = b - A x (:38); p = r (:39); rr = r.r (:40)
        mine.copy_(x_local)
        par.exchange_window(full)
        par.mvp_local(full, ap)
        r = b_local - ap
        mine.copy_(r)                                      # p lives in `full`
        ops.dot(r, r, rr)
        allreduce(rr)
        iters = 0
        for _k in range(self.iter_max):
            iters += 1
            par.exchange_window(full)                      # the entries of p this block references
            par.mvp_local(full, ap)                        # :43
            ops.dot(mine, ap, pap)
            allreduce(pap)
            torch.div(rr, pap, out=alpha)                  # :45
            ops.axpy(x_local, alpha, mine)                 # *x += p * alpha          :47
            torch.neg(alpha, out=neg_alpha)
            ops.axpy(r, neg_alpha, ap)                     # r -= mat_p * alpha       :49  (r + round(-alpha*Ap))
            rr_prev.copy_(rr)
            ops.dot(r, r, rr)                              # :51
            allreduce(rr)
            if float(rr.item()) ** 0.5 < self.tol:         # :52-54, before the beta update
                break
            torch.div(rr, rr_prev, out=beta)               # :56
            ops.xpby(mine, beta, r)                        # p = beta*p + r           :58-59
        self.iterations = iters
        self.r_norm_squared = float(rr.item())
        return x_local
