"""numpy restatement of K2t (sparsemat_amd/csrc/spmv_tiled.hip) -- test infrastructure, not product code.

It restates (a) the BUILD: the copy by column slice, its chunks (starts snapped to row boundaries), the continuation bits, the
product slots, the row blocks cut by product counts, the row codes and the tile table -- integer work, compared array by array
with what the library built (smh_crs_tiled_array); and (b) the two passes' SUMMATION ORDER in the value type: the in-lane left
fold, the Kogge-Stone segmented scan over the 64 lanes (row_shr 1, 2, 4, 8, row_bcast 15, row_bcast 31), the carry into a
lane's first segment, and pass 2's per-slice adds (equal neighbours merged by the same fold) -- compared bit for bit with the
device's products and its y.  What the model is pinned to is the kernel's own documented order; parity with the REFERENCE
(sparsematrix.rs:146-158) is the oracle's job (tests/util.py::assert_spmv_close)."""
import numpy as np

SLICE = 16384
SNAP = 16
GROUP = 32  # row blocks begin at multiples of this many rows
CONT = 0x8000
TARGET = {np.dtype(np.float32): 174.0, np.dtype(np.float64): 60.0}
CAP = {np.dtype(np.float32): 3328, np.dtype(np.float64): 1664}


E1 = 4  # entries per lane in pass 1 (a chunk = 64 E1 slots)


def lanes_per(dtype):
    """Products per lane in pass 2 (16 bytes): a round = 64 of these; a chunk's share of the product stream is padded to it."""
    return 4 if np.dtype(dtype) == np.dtype(np.float32) else 2


def seg_scan(v, stop):
    """v, stop: [B, 64].  Inclusive segmented scan as the DPP steps do it: every lane reads the OLD values of its source."""
    v = v.copy()
    stop = stop.copy()
    lane = np.arange(64)

    def step(src, has):
        nonlocal v, stop
        s = np.where(has, src, 0)
        v_in, s_in = v[:, s], stop[:, s]
        add = has[None, :] & ~stop
        v = np.where(add, v + v_in, v).astype(v.dtype)
        stop = stop | (has[None, :] & s_in)

    for d in (1, 2, 4, 8):
        step(lane - d, (lane % 16) >= d)
    row = lane // 16
    step(16 * row - 1, (row == 1) | (row == 3))   # row_bcast:15, rows 1 and 3
    step(np.full(64, 31), row >= 2)               # row_bcast:31, rows 2 and 3
    return v


def fold_runs(p, cont):
    """p: [B, 64, E] products, cont: [B, 64, E] bool (entry continues the run of the entry before it).  Returns the running sums
    (the value at a run's last entry is the run's sum) and the mask of last entries."""
    p = p.copy()
    B, _, E = p.shape
    through = cont[:, :, 0].copy()
    first = np.zeros_like(cont)
    first[:, :, 0] = through
    for k in range(1, E):
        p[:, :, k] = np.where(cont[:, :, k], p[:, :, k - 1] + p[:, :, k], p[:, :, k])
        through = through & cont[:, :, k]
        first[:, :, k] = through
    run = seg_scan(p[:, :, E - 1], ~through)
    carry = np.zeros_like(run)
    carry[:, 1:] = run[:, :-1]
    for k in range(E):
        p[:, :, k] = np.where(first[:, :, k], carry + p[:, :, k], p[:, :, k])
    tail = np.ones_like(cont)
    tail[:, :, :E - 1] = ~cont[:, :, 1:]
    tail[:, :-1, E - 1] = ~cont[:, 1:, 0]
    return p, tail


class TiledModel:
    def __init__(self, off, col, val, n_rows, n_cols, target=None, cap=None):
        dt = np.dtype(val.dtype)
        self.dt, self.E = dt, lanes_per(dt)
        E = self.E
        self.CH, self.STRIDE = 64 * E1, 64 * E1 - SNAP
        CH, STRIDE = self.CH, self.STRIDE
        off = np.asarray(off, dtype=np.int64)
        col = np.asarray(col, dtype=np.int64)
        self.n_rows, self.n_cols = n_rows, n_cols
        self.n_cb = max(1, -(-n_cols // SLICE))
        rows = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(off))
        sl = col // SLICE
        order = np.argsort(sl, kind="stable")  # inside a slice: (row, storage order)
        cnt = np.bincount(sl, minlength=self.n_cb)
        start = np.concatenate([[0], np.cumsum(cnt)])
        srow, scol = rows[order], col[order]
        self.order = order
        # chunks
        slice_chunks = [0]
        c_first, c_len, c_slice = [], [], []
        for s in range(self.n_cb):
            n = int(cnt[s])
            r = srow[start[s]:start[s + 1]]
            nch = -(-n // STRIDE)
            starts = []
            for c in range(nch):
                nominal = c * STRIDE
                at = min(nominal, n)
                for j in range(SNAP):
                    q = nominal + j
                    if q >= n:
                        at = n
                        break
                    if q == 0 or r[q] != r[q - 1]:
                        at = q
                        break
                starts.append(at)
            for c in range(nch):
                a, b = starts[c], (starts[c + 1] if c + 1 < nch else n)
                c_first.append(int(start[s]) + a)
                c_len.append(b - a)
                c_slice.append(s)
            slice_chunks.append(slice_chunks[-1] + nch)
        self.slice_chunks = np.array(slice_chunks, dtype=np.uint32)
        self.n_chunks = len(c_first)
        nC = self.n_chunks
        self.c_first, self.c_len = np.array(c_first, dtype=np.int64), np.array(c_len, dtype=np.int64)
        # the copy: [n_chunks, CH] slots
        slot = np.arange(CH)[None, :]
        have = slot < self.c_len[:, None]
        src = np.where(have, self.c_first[:, None] + slot, 0)
        self.have = have
        self.src_entry = np.where(have, order[src] if len(order) else 0, 0)  # CSR index of every slot's entry
        row_s = np.where(have, srow[src] if len(srow) else 0, -1)
        cont = have.copy()
        cont[:, 0] = False
        cont[:, 1:] &= row_s[:, 1:] == row_s[:, :-1]
        self.cont, self.row_s = cont, row_s
        code = np.where(have, (scol[src] if len(scol) else 0) - np.array(c_slice, dtype=np.int64)[:, None] * SLICE, 0)
        self.codes = (code | np.where(cont, CONT, 0)).astype(np.uint16).reshape(-1)
        self.values = np.where(have, np.asarray(val)[self.src_entry] if len(val) else 0, 0).astype(dt).reshape(-1)
        # product slots: one per run, each chunk's share padded to E
        head = have & ~cont
        runs = head.sum(axis=1)
        share = (runs + E - 1) // E * E
        obase = np.concatenate([[0], np.cumsum(share)])
        self.obase, self.n_prod = obase, int(obase[-1])
        self.chunks = np.stack([obase[:-1], self.c_len], axis=1).astype(np.uint32)
        prow = np.zeros(self.n_prod, dtype=np.int64)
        real = np.zeros(self.n_prod, dtype=bool)
        for c in range(nC):
            rr = row_s[c][head[c]]
            prow[obase[c]:obase[c] + len(rr)] = rr
            real[obase[c]:obase[c] + len(rr)] = True
            if len(rr):
                prow[obase[c] + len(rr):obase[c + 1]] = row_s[c][self.c_len[c] - 1]
        self.prow, self.real = prow, real
        # a chunk whose last run goes on in the next chunk of its slice: its padding slots carry that row (and zeros), so that
        # pass 2 sees the parts as neighbours
        counts = real.copy()
        for c in range(nC - 1):
            if c_slice[c + 1] == c_slice[c] and self.c_len[c] and self.c_len[c + 1] and row_s[c + 1][0] == row_s[c][self.c_len[c] - 1]:
                counts[obase[c]:obase[c + 1]] = True
        self.counts = counts  # slots that take part in pass 2 (real ones + the zero padding of cut pairs)
        # row blocks of equal product counts, cut at multiples of GROUP rows (the build counts products per group of rows)
        rcount = np.bincount(prow[real], minlength=n_rows) if self.n_prod else np.zeros(n_rows, dtype=np.int64)
        n_groups = -(-n_rows // GROUP)
        gcount = np.add.reduceat(rcount, np.arange(0, n_rows, GROUP)) if n_rows else np.zeros(0, dtype=np.int64)
        target = TARGET[dt] if target is None else target
        cap = CAP[dt] if cap is None else cap
        cap_g = max(1, cap // GROUP)
        per_block = int(target * self.n_cb)
        rb = []
        r = 0
        while r < n_groups:
            rb.append(r * GROUP)
            have_n, e = 0, r
            while e < n_groups and e - r < cap_g and (e == r or have_n + gcount[e] <= per_block):
                have_n += gcount[e]
                e += 1
            r = e
        if not rb:
            rb.append(0)
        rb.append(n_rows)
        self.rb_start = np.array(rb, dtype=np.uint32)
        self.n_rb = len(rb) - 1
        self.R = int(max(1, np.diff(self.rb_start.astype(np.int64)).max()))
        blk = np.searchsorted(self.rb_start[1:].astype(np.int64), prow, side="right")
        blk = np.minimum(blk, self.n_rb - 1)
        # (the byte offset of the row's sum in pass 2's LDS: slot 0 is the dump slot)
        self.product_rows = np.where(counts, (prow - self.rb_start[blk].astype(np.int64) + 1) * dt.itemsize, 0).astype(np.uint16)
        # tile table
        pb = obase[self.slice_chunks.astype(np.int64)]
        ts = np.zeros((self.n_rb + 1, self.n_cb), dtype=np.uint32)
        for s in range(self.n_cb):
            seg = prow[pb[s]:pb[s + 1]]
            ts[:, s] = pb[s] + np.searchsorted(seg, self.rb_start.astype(np.int64), side="left")
        self.tile_start = ts

    def products(self, val, x):
        """Pass 1 in the value type: the product slots' values (padding slots: zero)."""
        dt, E, CH = self.dt, self.E, self.CH
        nC = self.n_chunks
        out = np.zeros(self.n_prod, dtype=dt)
        if nC == 0:
            return out
        v = np.where(self.have, np.asarray(val, dtype=dt)[self.src_entry], 0).astype(dt)
        colv = (self.codes.reshape(nC, CH) & 0x3FFF).astype(np.int64)
        base = (np.repeat(np.arange(self.n_cb), np.diff(self.slice_chunks.astype(np.int64))) * SLICE)[:, None]
        xi = np.minimum(base + colv, len(x) - 1)
        p = (v * np.asarray(x, dtype=dt)[xi]).astype(dt)
        p = np.where(self.have, p, 0).astype(dt)  # (what the empty slots multiply is never part of a run's sum)
        ps, tail = fold_runs(p.reshape(nC, 64, E1), self.cont.reshape(nC, 64, E1))
        tail = tail.reshape(nC, CH) & self.have
        ps = ps.reshape(nC, CH)
        for c in range(nC):
            vals = ps[c][tail[c]]
            out[self.obase[c]:self.obase[c] + len(vals)] = vals
        return out

    def reduce(self, prod):
        """Pass 2 in the value type, tile by tile, round by round."""
        dt, E, R = self.dt, self.E, self.R
        RND = 64 * E
        y = np.zeros(self.n_rows, dtype=dt)
        pad = np.concatenate([prod, np.zeros(2 * RND, dtype=dt)])
        rc = np.concatenate([self.product_rows, np.zeros(2 * RND, dtype=np.uint16)]).astype(np.int64) // dt.itemsize
        for rb in range(self.n_rb):
            r0 = int(self.rb_start[rb])
            acc = np.zeros(R + 1, dtype=dt)  # [0]: the dump slot
            for s in range(self.n_cb):
                lo, hi = int(self.tile_start[rb, s]), int(self.tile_start[rb + 1, s])
                a = lo & ~(E - 1)
                while a < hi:  # (an empty tile's single round adds nothing)
                    idx = np.arange(a, a + RND)
                    ok = (idx >= lo) & (idx < hi)
                    r = np.where(ok, rc[idx], 0)
                    p = np.where(ok, pad[idx], 0).astype(dt)
                    prev = np.concatenate([[0], r[:-1]])
                    cont = (r == prev) & (r != 0)
                    if cont.any():
                        ps, tail = fold_runs(p.reshape(1, 64, E), cont.reshape(1, 64, E))
                        p = ps.reshape(-1)
                        r = np.where(tail.reshape(-1), r, 0)
                    sel = r != 0
                    acc[r[sel]] = acc[r[sel]] + p[sel]
                    a += RND
            n = int(self.rb_start[rb + 1]) - r0
            y[r0:r0 + n] = acc[1:n + 1]
        return y
