"""SparseMatCRS: host-side mirror of the reference's ``SparseMatCRS<T,u32>`` (sparsemat_crs.rs) for
the SpMV hot path, backed by a device-resident ``smh_crs``.

The host owns the CRS arrays (``offset_rows``, ``columns``, ``values``); ``from_raw_parts`` copies
them to HBM once.  ``mvp`` / ``*`` run the hand-written HIP kernels.  Assembly
(``add_to``/``set``/``get_mut``) stays with the reference's own containers -- out of scope here.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib
from .densevec import DenseVec


class SparseMatCRS:
    def __init__(self, handle, dtype, keep=None):
        self._h = handle
        self._dtype = np.dtype(dtype)
        self._keep = keep

    # ---- constructors ------------------------------------------------------------------------
    @classmethod
    def from_raw_parts(cls, n_rows, n_cols, offset_rows, columns, values, validate=True):
        """Fills the gap the reference leaves (its only bulk constructor, from_sparsemat_index
        sparsemat_crs.rs:24-50, is pub(crate)): adopt CRS arrays as they are -- rows in storage
        order, unsorted, duplicates allowed."""
        values = np.ascontiguousarray(values)
        if values.dtype not in (np.float32, np.float64):
            raise TypeError("the HIP path handles f32/f64 values only (got %s)" % values.dtype)
        off = np.ascontiguousarray(offset_rows, dtype=np.uint32)
        col = np.ascontiguousarray(columns, dtype=np.uint32)
        if n_rows and len(off) != n_rows + 1:
            raise _lib.SparseMatPanic(_lib.SMH_ERR_INVALID, "offset_rows must have n_rows+1 entries")
        if len(col) != len(values):
            raise _lib.SparseMatPanic(_lib.SMH_ERR_INVALID, "columns and values differ in length")
        h = C.c_void_p()
        check(lib().smh_crs_create(_lib.dtype_code(values.dtype), n_rows, n_cols, len(values),
                                   off.ctypes.data if len(off) else None,
                                   col.ctypes.data if len(col) else None,
                                   values.ctypes.data if len(values) else None,
                                   1 if validate else 0, C.byref(h)))
        return cls(h, values.dtype)

    @classmethod
    def from_device_parts(cls, n_rows, n_cols, nnz, off_ptr, col_ptr, val_ptr, dtype, keep=None,
                          validate=False):
        """Borrow CRS arrays already resident in HBM (raw device pointers); ``keep`` is held alive."""
        h = C.c_void_p()
        check(lib().smh_crs_create_dev(_lib.dtype_code(dtype), n_rows, n_cols, nnz, C.c_void_p(off_ptr),
                                       C.c_void_p(col_ptr), C.c_void_p(val_ptr), 1 if validate else 0,
                                       C.byref(h)))
        return cls(h, dtype, keep)

    @classmethod
    def from_triplets(cls, rows, cols, values, ops=None, into_crs=False):
        """The CRS the reference ends up with after the call stream ``mat.add_to(rows[k], cols[k], values[k])``
        (``ops[k] == 1``: ``mat.set(...)``) on a SparseMatIndexList and ``to_crs()``
        (sparsemat_indexlist.rs:61-63,158-164; sparsemat_crs.rs:24-50) -- built on the device, bit-exact:
        entries of a row in order of first appearance, duplicates folded in stream order.
        ``into_crs=True``: the stream is applied to a SparseMatCRS itself (sparsemat_crs.rs:54-92,143-149): rows in
        reverse order of first appearance, first-push quirk included (see ``smh_crs_replay`` in the header)."""
        values = np.ascontiguousarray(values)
        if values.dtype not in (np.float32, np.float64):
            raise TypeError("the HIP path handles f32/f64 values only (got %s)" % values.dtype)
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        n = len(values)
        if len(rows) != n or len(cols) != n:
            raise _lib.SparseMatPanic(_lib.SMH_ERR_INVALID, "rows, cols and values differ in length")
        ops_a = None if ops is None else np.ascontiguousarray(ops, dtype=np.uint8)
        if ops_a is not None and len(ops_a) != n:
            raise _lib.SparseMatPanic(_lib.SMH_ERR_INVALID, "ops and values differ in length")
        h = C.c_void_p()
        fn = lib().smh_crs_replay if into_crs else lib().smh_crs_assemble
        check(fn(_lib.dtype_code(values.dtype), n, rows.ctypes.data if n else None, cols.ctypes.data if n else None,
                 values.ctypes.data if n else None, ops_a.ctypes.data if (ops_a is not None and n) else None, C.byref(h)))
        return cls(h, values.dtype)

    @classmethod
    def from_device_triplets(cls, n_ops, rows_ptr, cols_ptr, vals_ptr, dtype, ops_ptr=None, into_crs=False):
        """``from_triplets`` over operation arrays that already live in HBM (raw device pointers)."""
        h = C.c_void_p()
        fn = lib().smh_crs_replay_dev if into_crs else lib().smh_crs_assemble_dev
        check(fn(_lib.dtype_code(dtype), n_ops, C.c_void_p(rows_ptr), C.c_void_p(cols_ptr), C.c_void_p(vals_ptr),
                 C.c_void_p(ops_ptr or 0), C.byref(h)))
        return cls(h, dtype)

    def transpose(self):
        """SparseMatrix::transpose (sparsematrix.rs:174-184): ``ret.set(j, i, val)`` for every entry in row-major
        storage order into a fresh SparseMatCRS -- on the device, bit-exact including the storage order the
        reference's CRS container produces (rows reversed, first-push quirk)."""
        h = C.c_void_p()
        check(lib().smh_crs_transpose(self._h, C.byref(h)))
        return type(self)(h, self.dtype)

    @staticmethod
    def last_transpose_route():
        """"general" (stable sort by target row) or "bucketed" (two bucketed passes, csrc/transpose_bucket.hip): how this
        thread's last ``transpose`` was carried out."""
        return ("general", "bucketed")[lib().smh_last_transpose_route()]

    def column_info(self):
        """ColumnIter::assemble_column_info (sparsemat_crs.rs:180-191) as arrays ``(rows, col_ptr, entries)``:
        ``rows[k]`` = row of entry k; column j's entries in storage order are ``entries[col_ptr[j]:col_ptr[j+1]]``."""
        nnz, n_cols = self.n_non_zero_entries(), self.n_cols()
        rows = np.zeros(max(nnz, 1), np.uint32)
        col_ptr = np.zeros(n_cols + 1, np.uint32)
        entries = np.zeros(max(nnz, 1), np.uint32)
        check(lib().smh_crs_column_info(self._h, rows.ctypes.data, col_ptr.ctypes.data, entries.ctypes.data))
        return rows[:nnz], col_ptr, entries[:nnz]

    def iter_col(self, col, info=None, values=None):
        """ColumnIter::iter_col (sparsemat_crs.rs:193-204): the (row, value) pairs of a column in storage order.
        ``info`` = a ``column_info()`` result and ``values`` the downloaded values, to avoid repeating both."""
        rows, col_ptr, entries = info if info is not None else self.column_info()
        if values is None:
            values = self.raw_parts()[2]
        e = entries[col_ptr[col]:col_ptr[col + 1]] if col < len(col_ptr) - 1 else entries[:0]
        return list(zip(rows[e].tolist(), values[e].tolist()))

    def prod(self, rhs):
        """SparseMatrix::prod (sparsematrix.rs:186-210): the SparseMatCRS the reference's loops build for
        ``self.prod(&rhs)`` -- same sums (order of additions included), zero sums dropped, rows in descending column
        order.  Err("Dimension mismatch") becomes a SparseMatPanic with that text."""
        h = C.c_void_p()
        check(lib().smh_crs_prod(self._h, rhs._h, C.byref(h)))
        return type(self)(h, self.dtype)

    def is_symmetric(self):  # sparsematrix.rs:212-222
        out = C.c_int(0)
        check(lib().smh_crs_is_symmetric(self._h, C.byref(out)))
        return bool(out.value)

    def is_sorted(self):  # sparsematrix.rs:263-271
        out = C.c_int(0)
        check(lib().smh_crs_is_sorted(self._h, C.byref(out)))
        return bool(out.value)

    def sort_rows(self):
        """Sortable::sort_row (sparsemat_crs.rs:163-172) on every row: ascending columns, stable."""
        check(lib().smh_crs_sort_rows(self._h))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().smh_crs_destroy(h)
            except Exception:
                pass

    # ---- SparseMatrix accessors (sparsemat_crs.rs:124-134) ------------------------------------
    @property
    def dtype(self):
        return self._dtype

    def n_rows(self):
        return lib().smh_crs_n_rows(self._h)

    def n_cols(self):
        return lib().smh_crs_n_cols(self._h)

    def n_non_zero_entries(self):
        """Entries the handle's arrays hold.  The reference's ``n_non_zero_entries()`` (``columns.len()``) additionally counts
        an entry orphaned by the first-push quirk of a replay / transpose / prod: add ``orphans()``."""
        return lib().smh_crs_nnz(self._h)

    def orphans(self):
        """0 or 1: the entry the reference's container would still hold although no row reaches it (``smh_crs_orphans``)."""
        return lib().smh_crs_orphans(self._h)

    def empty(self):  # sparsematrix.rs:119-121
        return self.n_rows() == 0

    def density(self):  # sparsematrix.rs:237-241
        return self.n_non_zero_entries() / (self.n_rows() * self.n_cols())

    def raw_parts(self):
        off = np.empty(self.n_rows() + 1, dtype=np.uint32)
        col = np.empty(self.n_non_zero_entries(), dtype=np.uint32)
        val = np.empty(self.n_non_zero_entries(), dtype=self._dtype)
        check(lib().smh_crs_download(self._h, off.ctypes.data, col.ctypes.data, val.ctypes.data))
        return off, col, val

    def iter_row(self, row):  # sparsemat_crs.rs:102-110 (row >= n_rows -> empty)
        if row >= self.n_rows():
            return []
        off, col, val = self.raw_parts()
        return list(zip(col[off[row]:off[row + 1]], val[off[row]:off[row + 1]]))

    def scale(self, a):  # sparsemat_crs.rs:153-157
        check(lib().smh_crs_scale(self._h, float(a)))

    def update_values(self, values):
        values = np.ascontiguousarray(values, dtype=self._dtype)
        assert len(values) == self.n_non_zero_entries()
        check(lib().smh_crs_update_values(self._h, values.ctypes.data))

    # ---- kernel selection ----------------------------------------------------------------------
    def resolved_variant(self):
        v, lanes = C.c_int(), C.c_int()
        check(lib().smh_crs_resolved_variant(self._h, C.byref(v), C.byref(lanes)))
        return {1: "vector", 2: "merge", 3: "seq", 4: "stream", 5: "colblock", 6: "colfused", 7: "colsplit", 8: "tiled"}[v.value], lanes.value

    def set_colblock_shift(self, shift):
        """K2c: column blocks of 2**shift columns (0: automatic, 2 MiB of x)."""
        check(lib().smh_crs_set_colblock_shift(self._h, shift))

    def tiled_layout(self, arrays=False):
        """K2t (``smh_crs_tiled_layout``; builds the 2-D tiled copy on first use): dict with n_slices, slice_columns,
        rows_per_block, n_row_blocks, copy_entries, n_products; with ``arrays`` also the plan's arrays
        (``smh_crs_tiled_array``): slice_chunks, chunks (n x 2: first product slot, entries), codes, values, product_rows,
        row_block_start, tile_start ((n_row_blocks + 1) x n_slices), products (of the last launch)."""
        a, b, c, d, e, f = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_size_t(), C.c_size_t()
        check(lib().smh_crs_tiled_layout(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e)))
        check(lib().smh_crs_tiled_products(self._h, C.byref(f)))
        out = {"n_slices": a.value, "slice_columns": b.value, "rows_per_block": c.value, "n_row_blocks": d.value, "copy_entries": e.value,
               "n_products": f.value}
        if arrays:
            kinds = (("slice_chunks", np.uint32), ("chunks", np.uint32), ("codes", np.uint16), ("values", self.dtype), ("product_rows", np.uint16),
                     ("row_block_start", np.uint32), ("tile_start", np.uint32), ("products", self.dtype))
            for which, (name, dt) in enumerate(kinds):
                n = C.c_size_t()
                check(lib().smh_crs_tiled_array(self._h, which, None, 0, C.byref(n)))
                buf = np.empty(n.value // np.dtype(dt).itemsize, dtype=dt)
                check(lib().smh_crs_tiled_array(self._h, which, buf.ctypes.data, buf.nbytes, None))
                out[name] = buf
            out["chunks"] = out["chunks"].reshape(-1, 2)
            out["tile_start"] = out["tile_start"].reshape(d.value + 1, a.value)
        return out

    def colsplit_flag(self):
        """True when the handle keeps a row-length split (builds it on first use)."""
        flag = C.c_int()
        check(lib().smh_crs_colsplit(self._h, C.byref(flag), None, None, None, None, None))
        return bool(flag.value)

    def colsplit(self):
        """The row-length split (``smh_crs_colsplit``): dict with split (bool), min_long, n_long, long_rows and the two
        sub-matrices' raw parts ``long`` / ``short`` = (n_rows, offsets, columns, values)."""
        flag, ml, nl = C.c_int(), C.c_uint32(), C.c_size_t()
        check(lib().smh_crs_colsplit(self._h, C.byref(flag), C.byref(ml), C.byref(nl), None, None, None))
        out = {"split": bool(flag.value), "min_long": ml.value, "n_long": nl.value}
        if out["split"]:
            rows = np.zeros(nl.value, np.uint32)
            hl, hs = C.c_void_p(), C.c_void_p()
            check(lib().smh_crs_colsplit(self._h, None, None, None, rows.ctypes.data, C.byref(hl), C.byref(hs)))
            out["long_rows"] = rows
            for name, h in (("long", hl), ("short", hs)):
                nr, nnz = lib().smh_crs_n_rows(h), lib().smh_crs_nnz(h)
                off, col, val = np.zeros(nr + 1, np.uint32), np.zeros(nnz, np.uint32), np.zeros(nnz, self._dtype)
                check(lib().smh_crs_download(h, off.ctypes.data, col.ctypes.data if nnz else None, val.ctypes.data if nnz else None))
                var, lanes = C.c_int(), C.c_int()
                check(lib().smh_crs_resolved_variant(h, C.byref(var), C.byref(lanes)))
                out[name] = (nr, off, col, val)
                out[name + "_variant"] = {1: "vector", 2: "merge", 3: "seq", 4: "stream", 5: "colblock", 6: "colfused", 7: "colsplit", 8: "tiled"}[var.value]
        return out

    def colfused(self, arrays=True):
        """The K2f copy (``smh_crs_colfused``): dict with fits, shift, n_blocks, rows_per_lane, n_tiles and -- with
        ``arrays`` -- tile_rows, segments, counts, columns, values (entries sorted by (tile, block, row, storage order))."""
        fits, sh, nb, rt, nt = C.c_int(), C.c_uint32(), C.c_size_t(), C.c_uint32(), C.c_size_t()
        check(lib().smh_crs_colfused(self._h, C.byref(fits), C.byref(sh), C.byref(nb), C.byref(rt), C.byref(nt), None, None, None, None, None))
        out = {"fits": bool(fits.value), "shift": sh.value, "n_blocks": nb.value, "rows_per_lane": rt.value, "n_tiles": nt.value}
        if arrays and out["fits"]:
            nnz = self.n_non_zero_entries()
            tiles = np.zeros(nt.value + 1, np.uint32)
            seg = np.zeros(nt.value * nb.value + 1, np.uint32)
            cnt = np.zeros(nt.value * nb.value * 64 * rt.value, np.uint8)
            col = np.zeros(nnz, np.uint32)
            val = np.zeros(nnz, self._dtype)
            check(lib().smh_crs_colfused(self._h, None, None, None, None, None, tiles.ctypes.data, seg.ctypes.data,
                                         cnt.ctypes.data if len(cnt) else None, col.ctypes.data if nnz else None,
                                         val.ctypes.data if nnz else None))
            out.update(tile_rows=tiles, segments=seg, counts=cnt, columns=col, values=val)
        return out

    def colblock(self, arrays=True):
        """K2c column-blocked copy: dict(shift, n_blocks, rows_per_thread, span_fraction[, offsets[B, n_rows+1],
        columns, values])."""
        sh, nb, rpt, span = C.c_uint32(), C.c_size_t(), C.c_int(), C.c_double()
        check(lib().smh_crs_colblock(self._h, C.byref(sh), C.byref(nb), C.byref(rpt), C.byref(span), None, None, None))
        out = dict(shift=sh.value, n_blocks=nb.value, rows_per_thread=rpt.value, span_fraction=span.value)
        if arrays:
            nnz = self.n_non_zero_entries()
            off = np.zeros((nb.value, self.n_rows() + 1), dtype=np.uint32)
            col = np.zeros(nnz, dtype=np.uint32)
            val = np.zeros(nnz, dtype=self._dtype)
            check(lib().smh_crs_colblock(self._h, None, None, None, None, off.ctypes.data,
                                         col.ctypes.data if nnz else None, val.ctypes.data if nnz else None))
            out.update(offsets=off, columns=col, values=val)
        return out

    def set_vector_lanes(self, lanes):
        check(lib().smh_crs_set_vector_lanes(self._h, lanes))

    def set_stream_xs(self, mode):
        """K1s XS (the tile's column intervals of x staged in LDS): -1 automatic, 0 never, 1 whenever the tiles allow."""
        check(lib().smh_crs_set_stream_xs(self._h, mode))

    def set_stream_direct(self, mode):
        """K1s XD (stage offsets instead of column codes, unskewed product stage): -1 automatic, 0 never, 1 whenever x is staged."""
        check(lib().smh_crs_set_stream_direct(self._h, mode))

    def stream_direct(self):
        """Does a STREAM launch of this matrix run as K1s XD?"""
        a = C.c_int()
        check(lib().smh_crs_stream_direct(self._h, C.byref(a)))
        return bool(a.value)

    def stream_value_dict(self):
        """K1s XD-V (``smh_crs_stream_value_dict``): the dictionary of the matrix's distinct values the CSR-stream kernel multiplies
        with instead of reading the value array -- a numpy array of its entries, empty when the form is not active (more than 32 / 16
        distinct values, or no stage-offset codes)."""
        n = C.c_int()
        buf = np.zeros(32, dtype=self.dtype)
        check(lib().smh_crs_stream_value_dict(self._h, C.byref(n), buf.ctypes.data))
        return buf[:n.value].copy()

    def set_stream_value_dict(self, mode):
        """-1 automatic (whenever the values allow), 0 never (``smh_crs_set_stream_value_dict``)."""
        check(lib().smh_crs_set_stream_value_dict(self._h, int(mode)))

    def stream_layout(self):
        """What a STREAM launch of this matrix uses: dict(coded, byte_lengths, small_tiles, xs_chunks)."""
        a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(lib().smh_crs_stream_layout(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"coded": bool(a.value), "byte_lengths": bool(b.value), "small_tiles": bool(c.value), "xs_chunks": d.value}

    def set_vector_chunks(self, chunks):
        check(lib().smh_crs_set_vector_chunks(self._h, chunks))

    def set_ring(self, mode):
        """K1r (LDS x-ring) for the vector family: -1 automatic, 0 off, 1 on."""
        check(lib().smh_crs_set_ring(self._h, mode))

    def ring_entries(self):
        """Columns the LDS ring of this matrix's K1r plan holds (16384, or 32768 for f32 matrices that need it)."""
        out = C.c_uint32()
        check(lib().smh_crs_ring_entries(self._h, C.byref(out)))
        return out.value

    def ring_bands(self, intervals=False):
        """1 (one sliding window) or 4 (banded ring); with ``intervals`` also the per-tile column intervals
        [n_tiles64, 4, 2] of a banded plan (None otherwise)."""
        bands = C.c_uint32()
        check(lib().smh_crs_ring_bands(self._h, C.byref(bands), None))
        if not intervals:
            return bands.value
        if bands.value != 4:
            return bands.value, None
        table = np.zeros(((self.n_rows() + 63) // 64, 4, 2), dtype=np.uint32)
        check(lib().smh_crs_ring_bands(self._h, C.byref(bands), table.ctypes.data))
        return bands.value, table

    def ring_plan(self):
        """(n_blocks, ring_fraction, active, phase_ptr, phases[n,11]) of the K1r plan; a phase row is
        (row_begin, row_end, load_lo, load_hi, use_ring, lo1, lo2, lo3, hi1, hi2, hi3)."""
        nb, nph, frac, act = C.c_uint32(), C.c_size_t(), C.c_double(), C.c_int()
        check(lib().smh_crs_ring_plan(self._h, C.byref(nb), C.byref(nph), C.byref(frac), C.byref(act), None, None))
        ptr = np.zeros(nb.value + 1, dtype=np.uint32)
        ph = np.zeros((nph.value, 11), dtype=np.uint32)
        check(lib().smh_crs_ring_plan(self._h, None, None, None, None, ptr.ctypes.data,
                                      ph.ctypes.data if nph.value else None))
        return nb.value, frac.value, bool(act.value), ptr, ph

    def max_row_len(self):
        out = C.c_uint32()
        check(lib().smh_crs_max_row_len(self._h, C.byref(out)))
        return out.value

    def col_range(self):
        """(smallest, largest) stored column index."""
        lo, hi = C.c_uint32(), C.c_uint32()
        check(lib().smh_crs_col_range(self._h, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def merge_table(self):
        n = lib().smh_crs_merge_tiles(self._h)
        rows = np.zeros(n + 1, dtype=np.uint32)
        nz = np.zeros(n + 1, dtype=np.uint32)
        check(lib().smh_crs_merge_table(self._h, rows.ctypes.data, nz.ctypes.data))
        return rows, nz, lib().smh_crs_merge_tile_items(self._h)

    # ---- SparseMatrix::mvp (sparsematrix.rs:146-158) and `A * v` (:435-443) --------------------
    def mvp(self, rhs, variant="auto"):
        """y = A.rhs; returns a NEW vector with dim == n_rows.  ``rhs`` is a DenseVec (device
        resident) or anything array-like (host: uploaded, result returned as numpy)."""
        var = _lib.VARIANTS[variant]
        if isinstance(rhs, DenseVec):
            ret = DenseVec.zeros(self.n_rows(), self._dtype)
            check(lib().smh_crs_spmv_vec(self._h, rhs._h, ret._h, var))
            return ret
        x = np.ascontiguousarray(rhs, dtype=self._dtype)
        y = np.zeros(self.n_rows(), dtype=self._dtype)
        check(lib().smh_crs_spmv(self._h, x.ctypes.data if x.size else None, x.size,
                                 y.ctypes.data if y.size else None, var))
        return y

    def __mul__(self, rhs):
        return self.mvp(rhs)

    def inner_prod(self, lhs, rhs, variant="auto"):
        """SparseMatrix::inner_prod (sparsematrix.rs:161-171): lhs^T A rhs as a Python float."""
        out = C.c_double()
        var = _lib.VARIANTS[variant]
        if isinstance(lhs, DenseVec) and isinstance(rhs, DenseVec):
            check(lib().smh_crs_inner_prod_vec(self._h, lhs._h, rhs._h, var, C.byref(out)))
            return out.value
        a = np.ascontiguousarray(lhs, dtype=self._dtype)
        b = np.ascontiguousarray(rhs, dtype=self._dtype)
        check(lib().smh_crs_inner_prod(self._h, a.ctypes.data if a.size else None, a.size,
                                       b.ctypes.data if b.size else None, b.size, var, C.byref(out)))
        return out.value

    def prepare(self, variant="auto"):
        """Build the lazily created workspaces of ``variant`` now (before capturing mvp_dev into a hipGraph)."""
        check(lib().smh_crs_prepare(self._h, _lib.VARIANTS[variant]))

    def prepare_stats(self, variant="auto"):
        """``smh_crs_prepare_stats``: (prepare_ms, derived_bytes) -- what the inspectors of ``variant`` cost: host wall time of
        the create-time inspection plus the plan's build, and the device memory they hold beside the CRS arrays.  The reference
        has no set-up step (sparsemat_crs.rs:102-110)."""
        ms, nb = C.c_double(), C.c_size_t()
        check(lib().smh_crs_prepare_stats(self._h, _lib.VARIANTS[variant], C.byref(ms), C.byref(nb)))
        return ms.value, nb.value

    def mvp_dev(self, x_ptr, x_len, y_ptr, variant="auto", stream=None):
        """Asynchronous y = A.x on raw device pointers (stream: a hipStream_t value or None)."""
        check(lib().smh_crs_spmv_dev(self._h, C.c_void_p(x_ptr), x_len, C.c_void_p(y_ptr),
                                     _lib.VARIANTS[variant], C.c_void_p(stream or 0)))
