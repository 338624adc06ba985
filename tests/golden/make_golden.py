#!/usr/bin/env python3
"""Regenerates tests/golden/reference_kats.json.

The reference (Rust) cannot be built or run in the build image (no cargo/rustc), so the
golden vectors are the known-answer DATA of the reference's own unit tests
(/root/reference/src/lib.rs), transcribed here as (call sequence, input vector, asserted
value) triples with their file:line.  The CRS arrays are produced by replaying the call
sequence through ``oracle.assembly`` (a pure-Python restatement of the reference's
assembly containers, which fixes the storage order the SpMV accumulates in); the
``expect`` values are the literals the reference asserts with ``assert_eq!``.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys
from fractions import Fraction

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import assembly  # noqa: E402


def f32_exact(lit):
    """Nearest f32 to a decimal literal, computed exactly (what rustc does for `4.2f32`)."""
    v = np.float32(float(lit))
    # verify there was no double rounding: |lit - v| must be <= |lit - neighbours|
    q = Fraction(lit)
    lo, hi = np.nextafter(v, np.float32(-np.inf)), np.nextafter(v, np.float32(np.inf))
    d = abs(q - Fraction(float(v)))
    assert d <= abs(q - Fraction(float(lo))) and d <= abs(q - Fraction(float(hi))), lit
    return v


# --- call sequences of the reference's tests (DATA, with file:line) ------------------------
SEQ_INDEXLIST = [  # src/lib.rs:57-64 (also :158-165 rowvec, :183-190 par)
    ["add_to", 0, 1, "4.2"], ["add_to", 1, 2, "4.12"], ["add_to", 2, 2, "2.12"],
    ["add_to", 1, 1, "1.12"], ["add_to", 1, 1, "1.12"],  # *get_mut(1,1) += 1.12
    ["add_to", 0, 2, "0.12"],                            # *get_mut(0,2) += 0.12
    ["set", 0, 0, "8.12"],                               # *get_mut(0,0) = 8.12
    ["set", 0, 0, "7.12"],
]
SEQ_CRS = [  # src/lib.rs:116-121
    ["add_to", 0, 1, "4.2"], ["add_to", 2, 2, "2.12"], ["add_to", 1, 2, "4.12"],
    ["add_to", 3, 2, "1.12"], ["add_to", 3, 3, "5.12"],
]
SEQ_CG = [  # src/lib.rs:38-42
    ["set", 0, 0, "4.0"], ["set", 0, 1, "1.0"], ["set", 1, 0, "1.0"], ["set", 1, 1, "3.0"],
]


def replay(mat, seq, conv):
    for op, i, j, lit in seq:
        getattr(mat, op)(i, j, conv(lit))
    return mat


def crs_dict(mat, dtype):
    n_rows, n_cols, off, col, val = mat.to_crs_arrays()
    bits = val.view(np.uint32 if dtype == "f32" else np.uint64)
    return {
        "n_rows": int(n_rows), "n_cols": int(n_cols),
        "offset_rows": [int(v) for v in off], "columns": [int(v) for v in col],
        "values": [float(v) for v in val],  # decimal, shortest round-trip of the stored value
        "values_bits": ["0x%x" % int(b) for b in bits],  # exact bit patterns
    }


def main():
    cases = []

    m = replay(assembly.IndexListMatrix(np.float32), SEQ_INDEXLIST, f32_exact)
    cases.append({
        "name": "check_sparsemat_indexlist", "ref": "src/lib.rs:54-82,94-98", "dtype": "f32",
        "container": "SparseMatIndexList (also its to_crs())", "ops": SEQ_INDEXLIST,
        "crs": crs_dict(m, "f32"),
        # src/lib.rs:67-71: iteration order asserted by the reference
        "iter_prefix": [[0, 1, "4.2"], [0, 2, "0.12"], [0, 0, "7.12"], [1, 2, "4.12"]],
        "x": ["2.0", "4.8", "1.2"],                       # src/lib.rs:79
        "expect_mvp": [[0, "34.544"]],                     # src/lib.rs:81
        "expect_get": [[0, 0, "7.12"]],                    # src/lib.rs:65
        "expect_row_dense": [[1, ["0", "2.24", "4.12"]]],  # src/lib.rs:95-98 ("0 2.24 4.12 ")
        # src/lib.rs:85-90: sp.iter_col(2) after assemble_column_info() -- the index-list matrix lists a column's
        # entries in INSERTION order ((1,2) was added before (2,2) before (0,2))
        "expect_iter_col_insertion_order": [[2, [[1, "4.12"], [2, "2.12"], [0, "0.12"]]]],
        # src/lib.rs:99-101: let mp = sp_crs.prod(&sp).unwrap(); assert_eq!(mp.get(1, 2), 17.9632)
        # (row 1 of self = {2: 4.12, 1: 2.24}; column 2 of rhs in list order: 2.24*4.12 first, then 4.12*2.12)
        "expect_prod_self": [[1, 2, "17.9632"]],
    })

    m = replay(assembly.CrsPushMatrix(np.float32), SEQ_CRS, f32_exact)
    cases.append({
        "name": "check_sparsemat_crs", "ref": "src/lib.rs:114-154", "dtype": "f32",
        "container": "SparseMatCRS (direct add_to: push prepends to the row)", "ops": SEQ_CRS,
        "crs": crs_dict(m, "f32"),
        # src/lib.rs:122-128: full iteration order incl. (3,3) BEFORE (3,2)
        "iter_full": [[0, 1, "4.2"], [1, 2, "4.12"], [2, 2, "2.12"], [3, 3, "5.12"], [3, 2, "1.12"]],
        "x": ["2.0", "4.8", "1.2", "3.4"],                 # src/lib.rs:149
        "expect_mvp": [[0, "20.16"]],                      # src/lib.rs:151
        "expect_density": [5, 16],                         # src/lib.rs:153
        # src/lib.rs:137-142: sp_crs.iter_col(2) after assemble_column_info(): row-major storage order
        "expect_iter_col": [[2, [[1, "4.12"], [2, "2.12"], [3, "1.12"]]]],
        "expect_iter_row": [[0, [[1, "4.2"]]], [5, []]],   # src/lib.rs:144-148 (row 5 >= n_rows: empty)
    })

    m = replay(assembly.IndexListMatrix(np.float32), SEQ_INDEXLIST, f32_exact)
    cases.append({
        "name": "check_sparsemat_rowvec", "ref": "src/lib.rs:156-178", "dtype": "f32",
        "container": "SparseMatRowVec (append order, sparsemat_rowvec.rs:35-48)", "ops": SEQ_INDEXLIST,
        "crs": crs_dict(m, "f32"),
        "iter_prefix": [[0, 1, "4.2"], [0, 2, "0.12"], [0, 0, "7.12"], [1, 2, "4.12"]],
        "x": ["2.0", "4.8", "1.2"],                        # src/lib.rs:173
        "expect_mvp": [[0, "34.544"]],                     # src/lib.rs:175
    })

    m = replay(assembly.ParMatrix(4, 16, np.float32), SEQ_INDEXLIST, f32_exact)
    cases.append({
        "name": "check_sparsemat_par", "ref": "src/lib.rs:180-202", "dtype": "f32",
        "container": "SparseMatPar<SparseMatIndexList>::with_sub_matrices(4, 16)", "ops": SEQ_INDEXLIST,
        "par": {"n_blocks": 4, "max_n_rows": 16, "rows_per_block": 4},
        "crs": crs_dict(m, "f32"),
        "iter_prefix": [[0, 1, "4.2"], [0, 2, "0.12"], [0, 0, "7.12"], [1, 2, "4.12"]],
        "x": ["2.0", "4.8", "1.2"],                        # src/lib.rs:197
        "expect_mvp": [[0, "34.544"]],                     # src/lib.rs:199
    })

    m = replay(assembly.IndexListMatrix(np.float64), SEQ_CG, float)
    cases.append({
        "name": "check_cg", "ref": "src/lib.rs:36-52", "dtype": "f64",
        "container": "SparseMatIndexList<f64,u32>", "ops": SEQ_CG,
        "crs": crs_dict(m, "f64"),
        "b": ["1.0", "2.0"], "x0": ["2.0", "1.0"],         # src/lib.rs:43-48
        "cg": {"tol": 1e-12, "iter_max": 10000},            # linearsolver.rs:17-24 (Default)
        "expect_floor_1e4": [[0, "0.0909"]],               # src/lib.rs:51
    })

    out = {
        "_about": "Known-answer data of the reference's own unit tests (lostinc0de/sparsemat "
                  "src/lib.rs); regenerate with tests/golden/make_golden.py. Literals are "
                  "strings so that f32 parsing is exact; crs.values_bits are the exact bits.",
        "cases": cases,
    }
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
