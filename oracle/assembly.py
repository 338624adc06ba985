"""Assembly semantics of the reference, restated in pure Python -- TEST INFRASTRUCTURE ONLY.

Small cases only (the reference's own unit tests build <= 4x4 matrices).  What matters
to the SpMV hot path is the STORAGE ORDER these containers hand to
``SparseMatrix::mvp`` (sparsematrix.rs:146-158), because the reference accumulates in
storage order and its tests assert exact f32 results (src/lib.rs:80-82, 150-152).

* ``IndexListMatrix``  -- sparsemat_indexlist.rs: entries appended, a row is iterated
  in insertion order (indexlist.rs:62-83 walks to the row's tail); ``to_crs`` keeps that
  order (sparsemat_crs.rs:24-50).
* ``CrsPushMatrix``    -- sparsemat_crs.rs:71-92: ``push`` inserts at the START of the
  row, including the quirk that the very first push leaves ``n_rows == 0`` (:75-81).
* ``ParMatrix``        -- sparsemat_par.rs:20-35, 86-107 over IndexListMatrix blocks.

Arithmetic on stored values follows the value dtype (np.float32 / np.float64), one
rounding per operation, like the Rust scalar ops.
"""
import numpy as np

UNSET = 0xFFFFFFFF  # SparseMatrix::UNSET for Index = u32 (sparsematrix.rs:68)


class _Base:
    def __init__(self, dtype):
        self.dtype = np.dtype(dtype).type

    # sparsematrix.rs:226-228 / :231-233
    def set(self, i, j, val):
        k = self._get_mut(i, j)
        self._vals()[k] = self.dtype(val)

    def add_to(self, i, j, val):
        k = self._get_mut(i, j)
        self._vals()[k] = self.dtype(self._vals()[k] + self.dtype(val))

    def get(self, i, j):
        k = self._find_index(i, j)
        return self.dtype(0) if k is None else self._vals()[k]


class IndexListMatrix(_Base):
    def __init__(self, dtype=np.float32):
        super().__init__(dtype)
        self.n_cols = 0
        self.columns, self.values = [], []
        self.rows = []  # rows[i] = list of entry indices in insertion order

    def _vals(self):
        return self.values

    def n_rows(self):
        return len(self.rows)  # indexlist.rs:58-60 (pos_start.len())

    def _find_index(self, i, j):  # sparsemat_indexlist.rs:29-42
        if i < self.n_rows():
            for k in self.rows[i]:
                if self.columns[k] == j:
                    return k
        return None

    def _push(self, i, j, val):  # sparsemat_indexlist.rs:45-53 + indexlist.rs:62-83
        if j >= self.n_cols:
            self.n_cols = j + 1
        while i >= len(self.rows):
            self.rows.append([])
        k = len(self.columns)
        assert k != UNSET
        self.rows[i].append(k)
        self.columns.append(j)
        self.values.append(self.dtype(val))
        return k

    def _get_mut(self, i, j):  # sparsemat_indexlist.rs:158-164
        k = self._find_index(i, j)
        return self._push(i, j, 0) if k is None else k

    def iter_row(self, i):
        if i >= self.n_rows():
            return []
        return [(self.columns[k], self.values[k]) for k in self.rows[i]]

    def to_crs_arrays(self):  # to_crs :61-63 -> sparsemat_crs.rs:24-50
        off, col, val = [], [], []
        for i in range(self.n_rows()):
            off.append(len(col))
            for c, v in self.iter_row(i):
                col.append(c)
                val.append(v)
        off.append(len(col))
        return (self.n_rows(), self.n_cols, np.array(off, np.uint32), np.array(col, np.uint32),
                np.array(val, self.dtype))


class CrsPushMatrix(_Base):
    def __init__(self, dtype=np.float32):
        super().__init__(dtype)
        self._n_rows = 0
        self.n_cols = 0
        self.values, self.columns, self.offset_rows = [], [], []

    def _vals(self):
        return self.values

    def n_rows(self):
        return self._n_rows

    def _find_index(self, i, j):  # sparsemat_crs.rs:54-67
        if i < self._n_rows:
            for k in range(self.offset_rows[i], self.offset_rows[i + 1]):
                if self.columns[k] == j:
                    return k
        return None

    def _push(self, i, j, val):  # sparsemat_crs.rs:71-92
        if j >= self.n_cols:
            self.n_cols = j + 1
        if len(self.offset_rows) == 0:
            self.offset_rows = [0] * (i + 2)  # n_rows is NOT updated here (:75-76)
        elif i >= self._n_rows:
            last = self.offset_rows[-1]
            if i + 2 >= len(self.offset_rows):
                self.offset_rows += [last] * (i + 2 - len(self.offset_rows))
            else:
                self.offset_rows = self.offset_rows[:i + 2]  # Vec::resize truncates
            self._n_rows = i + 1
        if self.offset_rows[i] == UNSET:
            raise OverflowError("Maximum number of %d entries reached" % UNSET)
        k = self.offset_rows[i]
        self.columns.insert(k, j)
        self.values.insert(k, self.dtype(val))
        for r in range(i + 1, len(self.offset_rows)):
            self.offset_rows[r] += 1
        return k

    def _get_mut(self, i, j):  # sparsemat_crs.rs:143-149
        k = self._find_index(i, j)
        return self._push(i, j, 0) if k is None else k

    def iter_row(self, i):  # sparsemat_crs.rs:102-110
        if i < self._n_rows:
            s, e = self.offset_rows[i], self.offset_rows[i + 1]
            return list(zip(self.columns[s:e], self.values[s:e]))
        return []

    def to_crs_arrays(self):
        return (self._n_rows, self.n_cols, np.array(self.offset_rows[:self._n_rows + 1], np.uint32),
                np.array(self.columns, np.uint32), np.array(self.values, self.dtype))


class ParMatrix(_Base):
    """SparseMatPar<SparseMatIndexList> (sparsemat_par.rs)."""

    def __init__(self, n_blocks, max_n_rows, dtype=np.float32):
        super().__init__(dtype)
        self.n_blocks = n_blocks
        self.rows_per_block = max_n_rows // n_blocks  # :21
        self.blocks = [IndexListMatrix(dtype) for _ in range(n_blocks)]

    def block_and_row(self, row):  # :31-35 (clamps to n_blocks: the reference's off-by-one)
        b = min(row // self.rows_per_block, self.n_blocks)
        return b, row - b * self.rows_per_block

    def set(self, i, j, val):
        b, r = self.block_and_row(i)
        self.blocks[b].set(r, j, val)

    def add_to(self, i, j, val):
        b, r = self.block_and_row(i)
        self.blocks[b].add_to(r, j, val)

    def get(self, i, j):
        b, r = self.block_and_row(i)
        return self.blocks[b].get(r, j)

    def n_rows(self):  # :95-107
        last = 0
        for b, m in enumerate(self.blocks):
            if m.n_rows() == 0:
                break
            last = b
        return last * self.rows_per_block + self.blocks[last].n_rows()

    def n_cols(self):
        return max(m.n_cols for m in self.blocks)

    def iter_row(self, i):  # :86-89
        b, r = self.block_and_row(i)
        return self.blocks[b].iter_row(r)

    def to_crs_arrays(self):
        """Global CRS as the default mvp walks it (rows 0..n_rows through iter_row)."""
        off, col, val = [], [], []
        for i in range(self.n_rows()):
            off.append(len(col))
            for c, v in self.iter_row(i):
                col.append(c)
                val.append(v)
        off.append(len(col))
        return (self.n_rows(), self.n_cols(), np.array(off, np.uint32), np.array(col, np.uint32),
                np.array(val, self.dtype))
