"""SparseMatParLocal / ParVec / Comm: ``SparseMatPar<SparseMatCRS<T,u32>>`` (sparsemat_par.rs:12-35, 86-140) through the C ABI
(``smh_par_*``, ``smh_comm_*``; ``csrc/par.hip``) -- the partition, the exchange of the dense vector (RCCL all-gather /
window send-receive, or direct peer reads) and the device-resident CG all live in the library:

* one process, all blocks: ``with_sub_matrices`` (split host arrays) or ``adopt`` (blocks already on their devices);
* one process per GPU: ``Comm`` (``ncclCommInitRank``) + ``for_rank`` around the rank's own block.

(Round 1's form over ``torch.distributed`` is test infrastructure now: ``tests/par_reference.py``, the restatement the plan
arithmetic of ``smh_par_plan`` is checked against in the CPU / gloo tests.)
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


class Comm:
    """One rank of an RCCL communicator (``smh_comm_*``).  ``Comm.unique_id()`` on one rank, the 128 bytes to the others
    by any host-side means, then ``Comm(id, n_ranks, rank)`` on every rank (after ``smh_set_device``)."""

    def __init__(self, uid, n_ranks, rank):
        assert len(uid) == _lib.COMM_ID_BYTES
        self._buf = C.create_string_buffer(bytes(uid), _lib.COMM_ID_BYTES)
        h = C.c_void_p()
        check(lib().smh_comm_create(self._buf, int(n_ranks), int(rank), C.byref(h)))
        self._h = h

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        check(lib().smh_comm_unique_id(buf))
        return bytes(buf.raw)

    def size(self):
        return lib().smh_comm_size(self._h)

    def rank(self):
        return lib().smh_comm_rank(self._h)

    def ranks_seen(self):
        """(ncclCommCount, ncclCommCuDevice) of this rank's communicator -- what RCCL itself says (``smh_comm_ranks_seen``)."""
        n, d = C.c_int(), C.c_int()
        check(lib().smh_comm_ranks_seen(self._h, C.byref(n), C.byref(d)))
        return n.value, d.value

    @staticmethod
    def rccl_version():
        v = C.c_int()
        check(lib().smh_rccl_version(C.byref(v)))
        return v.value

    def barrier(self):
        check(lib().smh_comm_barrier(self._h))

    def max(self, value):
        v = C.c_double(float(value))
        check(lib().smh_comm_max_f64(self._h, C.byref(v)))
        return v.value

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().smh_comm_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ParVec:
    """Distributed DenseVec (``smh_par_vec``): one full-length device buffer per local block."""

    def __init__(self, par, n):
        self.par = par
        h = C.c_void_p()
        check(lib().smh_par_vec_create(par._h, int(n), C.byref(h)))
        self._h = h
        self.n = int(n)

    def upload(self, host):
        a = np.ascontiguousarray(host, dtype=self.par._dtype)
        assert len(a) == self.n
        check(lib().smh_par_vec_upload(self._h, a.ctypes.data if len(a) else None))
        return self

    def download(self):
        """owned slices of the local blocks at their global offsets (other entries zero)."""
        out = np.zeros(self.n, self.par._dtype)
        check(lib().smh_par_vec_download(self._h, out.ctypes.data))
        return out

    def download_block(self, local_block):
        out = np.zeros(self.n, self.par._dtype)
        check(lib().smh_par_vec_download_block(self._h, local_block, out.ctypes.data))
        return out

    def ptr(self, local_block=0):
        p = C.c_void_p()
        check(lib().smh_par_vec_ptr(self._h, local_block, C.byref(p)))
        return p.value

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and getattr(self.par, "_h", None):
            lib().smh_par_vec_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SparseMatParLocal:
    def __init__(self, handle, dtype, keep=None):
        self._h = handle
        self._dtype = np.dtype(dtype).type
        self._keep = keep  # adopted blocks / communicator must outlive the handle

    @classmethod
    def with_sub_matrices(cls, n_blocks, n_rows, n_cols, offset_rows, columns, values, device_ids=None, validate=True, split="rows"):
        """``SparseMatPar::with_sub_matrices(n_blocks, max_n_rows)`` (sparsemat_par.rs:20-28) filled from a global CRS:
        block b = rows [b R, (b+1) R), R = n_rows // n_blocks, the last block taking the remainder (``split="rows"``); or blocks of
        equal entry counts (``split="nnz"``, ``smh_par_create_split``: SURVEY 8e's option for skewed matrices)."""
        values = np.ascontiguousarray(values)
        if values.dtype not in (np.float32, np.float64):
            raise TypeError("the HIP path handles f32/f64 values only (got %s)" % values.dtype)
        off = np.ascontiguousarray(offset_rows, dtype=np.uint32)
        col = np.ascontiguousarray(columns, dtype=np.uint32)
        if len(off) != n_rows + 1 or len(col) != len(values):
            raise _lib.SparseMatPanic(_lib.SMH_ERR_INVALID, "malformed CRS arrays")
        dev = None
        if device_ids is not None:
            dev = (C.c_int * n_blocks)(*[int(d) for d in device_ids])
        h = C.c_void_p()
        check(lib().smh_par_create_split(_lib.dtype_code(values.dtype), n_blocks, dev, n_rows, n_cols, off.ctypes.data,
                                         col.ctypes.data if len(col) else None, values.ctypes.data if len(values) else None,
                                         1 if validate else 0, {"rows": 0, "nnz": 1}[split], C.byref(h)))
        return cls(h, values.dtype)

    @classmethod
    def adopt(cls, blocks, n_rows, split_rows=None):
        """Blocks (``SparseMatCRS``) that already live on their devices: block b = rows [b R, (b+1) R), or -- ``split_rows``, the
        n_blocks + 1 row boundaries -- rows [split_rows[b], split_rows[b + 1])."""
        arr = (C.c_void_p * len(blocks))(*[b._h for b in blocks])
        h = C.c_void_p()
        table = None if split_rows is None else np.ascontiguousarray(split_rows, dtype=np.uintp)
        check(lib().smh_par_adopt_split(len(blocks), arr, int(n_rows), None if table is None else table.ctypes.data, C.byref(h)))
        return cls(h, blocks[0].dtype, keep=list(blocks))

    @classmethod
    def for_rank(cls, comm, n_rows, block, row_begin=None):
        """One process per GPU: this rank's block; n_blocks = comm size, block id = comm rank.  Collective.  ``row_begin``: the
        rank's first row when the blocks were not cut by the reference's arithmetic (all ranks pass one, or none does)."""
        h = C.c_void_p()
        check(lib().smh_par_create_rank_split(comm._h, int(n_rows), block._h, C.c_size_t(-1).value if row_begin is None else int(row_begin),
                                              C.byref(h)))
        return cls(h, block.dtype, keep=[comm, block])

    def split(self):
        """The n_blocks + 1 row boundaries of the partition (``smh_par_split``)."""
        out = np.zeros(self.n_blocks() + 1, dtype=np.uintp)
        check(lib().smh_par_split(self._h, out.ctypes.data))
        return [int(v) for v in out]

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().smh_par_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def n_blocks(self):
        return lib().smh_par_n_blocks(self._h)

    def n_local_blocks(self):
        return lib().smh_par_n_local_blocks(self._h)

    def n_rows(self):
        return lib().smh_par_n_rows(self._h)

    def n_cols(self):  # sparsemat_par.rs:109-115
        return lib().smh_par_n_cols(self._h)

    def n_non_zero_entries(self):  # :117-123
        return lib().smh_par_nnz(self._h)

    def rows_per_block(self):
        return lib().smh_par_rows_per_block(self._h)

    def block(self, b):
        """(row_begin, row_end, device, n_non_zero_entries) of local block b."""
        crs, r0, r1, dev = C.c_void_p(), C.c_size_t(), C.c_size_t(), C.c_int()
        check(lib().smh_par_block(self._h, b, C.byref(crs), C.byref(r0), C.byref(r1), C.byref(dev)))
        return r0.value, r1.value, dev.value, lib().smh_crs_nnz(crs)

    def block_stream(self, b=0):
        s = C.c_void_p()
        check(lib().smh_par_block_stream(self._h, b, C.byref(s)))
        return s.value

    def get_block_and_row_id(self, row):  # :31-35 (clamped to the last block)
        b, r = C.c_size_t(), C.c_size_t()
        check(lib().smh_par_get_block_and_row_id(self._h, row, C.byref(b), C.byref(r)))
        return b.value, r.value

    def scale(self, a):  # :135-139
        check(lib().smh_par_scale(self._h, float(a)))

    def set_backend(self, name):
        check(lib().smh_par_set_backend(self._h, _lib.PAR_BACKENDS[name]))

    def backend(self):
        return {v: k for k, v in _lib.PAR_BACKENDS.items()}[lib().smh_par_backend(self._h)]

    def set_threads(self, mode):
        """One issuing host thread per local block (``smh_par_set_threads``): -1 automatic (off unless SMH_PAR_THREADS=1), 0 off, 1 on.  Results do not
        depend on it."""
        check(lib().smh_par_set_threads(self._h, int(mode)))

    def set_overlap(self, on):
        """Window exchanges beside the interior rows' product (``smh_par_set_overlap``; on by default)."""
        check(lib().smh_par_set_overlap(self._h, 1 if on else 0))

    def interior(self, local_block, variant="auto"):
        """Local rows [begin, end) of a local block that its kernel for ``variant`` multiplies beside the exchange (equal: none)."""
        a, e = C.c_size_t(), C.c_size_t()
        check(lib().smh_par_interior(self._h, local_block, _lib.VARIANTS[variant], C.byref(a), C.byref(e)))
        return a.value, e.value

    def exchange_mode(self, mode="auto"):
        """(what ``mode`` resolves to, largest number of entries any block receives in a window exchange)."""
        m, worst = C.c_int(), C.c_size_t()
        check(lib().smh_par_exchange_mode(self._h, _lib.EXCHANGES[mode], C.byref(m), C.byref(worst)))
        return _lib.EXCHANGE_NAMES[m.value], worst.value

    def vec(self, n=None, host=None):
        v = ParVec(self, self.n_rows() if n is None else n)
        if host is not None:
            v.upload(host)
        return v

    def mvp_dev(self, x, y, variant="auto", exchange="auto"):
        """The intended ``mvp_par`` (:37-68), device resident: y slices = A_b x, then one exchange of y.  Asynchronous."""
        check(lib().smh_par_spmv_dev(self._h, x._h, y._h, _lib.VARIANTS[variant], _lib.EXCHANGES[exchange]))

    def exchange(self, v, mode="auto"):
        check(lib().smh_par_exchange(self._h, v._h, _lib.EXCHANGES[mode]))

    def synchronize(self):
        check(lib().smh_par_synchronize(self._h))

    def mvp(self, rhs, variant="auto"):
        """``SparseMatrix::mvp`` through the blocks; returns a new vector with ``n_rows`` entries."""
        x = np.ascontiguousarray(rhs, dtype=self._dtype)
        y = np.zeros(self.n_rows(), self._dtype)
        check(lib().smh_par_spmv(self._h, x.ctypes.data if len(x) else None, len(x), y.ctypes.data, _lib.VARIANTS[variant]))
        return y

    def cg_solve_vec(self, b, x, tol=1e-12, iter_max=10_000, variant="auto", check_every=0):
        iters, rr = C.c_size_t(), C.c_double()
        check(lib().smh_par_cg_solve_vec(self._h, b._h, x._h, float(tol), int(iter_max), _lib.VARIANTS[variant], int(check_every),
                                         C.byref(iters), C.byref(rr)))
        return iters.value, rr.value

    def cg_solve(self, b, x, tol=1e-12, iter_max=10_000, variant="auto"):
        """``ConjugateGradient::solve(&par, &b, &mut x)``: x (numpy array of the matrix's dtype) is updated in place;
        returns (iterations, r.r)."""
        b = np.ascontiguousarray(b, dtype=self._dtype)
        if x.dtype != self._dtype or not x.flags["C_CONTIGUOUS"]:
            raise TypeError("x must be a contiguous %s array (it is updated in place)" % np.dtype(self._dtype).name)
        iters, rr = C.c_size_t(), C.c_double()
        check(lib().smh_par_cg_solve(self._h, b.ctypes.data if len(b) else None, len(b), x.ctypes.data if len(x) else None, len(x),
                                     float(tol), int(iter_max), _lib.VARIANTS[variant], C.byref(iters), C.byref(rr)))
        return iters.value, rr.value


def plan(n_blocks, n_rows, needs, lo, hi, block, split_rows=None):
    """``smh_par_plan[_split]``: the window-exchange plan of one block (pure host arithmetic in the library; no device).
    Returns (recv, send, auto_mode, max_recv): recv[q] / send[q] = (begin, end) global ranges."""
    needs = np.ascontiguousarray(needs, dtype=np.uint8)
    lo = np.ascontiguousarray(lo, dtype=np.uint32)
    hi = np.ascontiguousarray(hi, dtype=np.uint32)
    arrs = [np.zeros(n_blocks, dtype=np.uintp) for _ in range(4)]
    mode, worst = C.c_int(), C.c_size_t()
    table = None if split_rows is None else np.ascontiguousarray(split_rows, dtype=np.uintp)
    check(lib().smh_par_plan_split(n_blocks, n_rows, None if table is None else table.ctypes.data, needs.ctypes.data, lo.ctypes.data,
                                   hi.ctypes.data, block, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data,
                                   C.byref(mode), C.byref(worst)))
    recv = [(int(a), int(b)) for a, b in zip(arrs[0], arrs[1])]
    send = [(int(a), int(b)) for a, b in zip(arrs[2], arrs[3])]
    return recv, send, _lib.EXCHANGE_NAMES[mode.value], worst.value
