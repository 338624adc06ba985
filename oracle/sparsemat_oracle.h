/*
 * sparsemat_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, single thread, -ffp-contract=off) of the reference
 * crate's arithmetic on the CSR SpMV / BLAS-1 / CG hot path.  Nothing in the
 * shipped product (sparsemat_amd/, include/) may include, link or call this:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and
 * only as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED.  The reference is Rust and cannot be built here (no
 * cargo/rustc in the image), so the oracle is pinned against every known-answer
 * vector the reference's own tests hold for this path (src/lib.rs:36-52, 80-82,
 * 122-128, 150-152, 174-176, 198-200) -- see tests/golden/ and
 * tests/test_oracle_golden.py.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * the reference repo root).
 */
#ifndef SPARSEMAT_ORACLE_H
#define SPARSEMAT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes: 0 ok; the others mirror the reference's panics */
#define ORC_OK 0
#define ORC_ERR_INDEX_OOB 1     /* densevec.rs:40-42  values[i] out of bounds      */
#define ORC_ERR_NOT_SQUARE 2    /* linearsolver.rs:30-32 "Matrix is not symmetric" */
#define ORC_ERR_SIZE_MISMATCH 3 /* linearsolver.rs:33-36 / densevec.rs:52-54,61-63 */
#define ORC_ERR_CAPACITY 4      /* sparsemat_crs.rs:82-84 "Maximum number of {} entries reached" */
#define ORC_ERR_ZERO_DIAGONAL 5 /* the Jacobi extension only: get(i, i) == 0 */

/* ---- SpMV: sparsematrix.rs:146-158 over sparsemat_crs.rs:102-110 ------- */
int orc_spmv_f32(size_t n_rows, const uint32_t *offset_rows, const uint32_t *columns,
                 const float *values, const float *x, size_t x_len, float *y);
int orc_spmv_f64(size_t n_rows, const uint32_t *offset_rows, const uint32_t *columns,
                 const double *values, const double *x, size_t x_len, double *y);
/* rows [row_begin,row_end) only (used by the bounded cpu_baseline sample) */
int orc_spmv_rows_f32(size_t row_begin, size_t row_end, const uint32_t *offset_rows,
                      const uint32_t *columns, const float *values, const float *x,
                      size_t x_len, float *y);
int orc_spmv_rows_f64(size_t row_begin, size_t row_end, const uint32_t *offset_rows,
                      const uint32_t *columns, const double *values, const double *x,
                      size_t x_len, double *y);
/* the same per-row loop with rows spread over all host cores (OpenMP) -- NOT reference behaviour (the
 * reference is serial); bench.py's second CPU row.  y is bit-identical to orc_spmv's. */
int orc_spmv_omp_f32(size_t n_rows, const uint32_t *offset_rows, const uint32_t *columns,
                     const float *values, const float *x, size_t x_len, float *y, int threads, int *threads_out);
int orc_spmv_omp_f64(size_t n_rows, const uint32_t *offset_rows, const uint32_t *columns,
                     const double *values, const double *x, size_t x_len, double *y, int threads, int *threads_out);
/* sum_j |a_ij * x_j| per row in f64: the scale of the componentwise parity bound */
void orc_spmv_abs_f32(size_t n_rows, const uint32_t *offset_rows, const uint32_t *columns,
                      const float *values, const float *x, double *out);
void orc_spmv_abs_f64(size_t n_rows, const uint32_t *offset_rows, const uint32_t *columns,
                      const double *values, const double *x, double *out);
/* x^T A y: sparsematrix.rs:161-171 */
float orc_mat_inner_prod_f32(size_t n_rows, const uint32_t *offset_rows, const uint32_t *columns,
                             const float *values, const float *lhs, const float *rhs);
double orc_mat_inner_prod_f64(size_t n_rows, const uint32_t *offset_rows, const uint32_t *columns,
                              const double *values, const double *lhs, const double *rhs);

/* ---- DenseVec element-wise ops: densevec.rs:51-73 ---------------------- */
int orc_vec_add_f32(float *x, size_t nx, const float *y, size_t ny);
int orc_vec_sub_f32(float *x, size_t nx, const float *y, size_t ny);
void orc_vec_scale_f32(float *x, size_t n, float a);
int orc_vec_add_f64(double *x, size_t nx, const double *y, size_t ny);
int orc_vec_sub_f64(double *x, size_t nx, const double *y, size_t ny);
void orc_vec_scale_f64(double *x, size_t n, double a);
/* y += round(a*x): `*x += p.clone() * alpha`  linearsolver.rs:47 (Mul densevec.rs:121-130, AddAssign :76-81) */
void orc_vec_axpy_f32(float *y, float a, const float *x, size_t n);
void orc_vec_axpy_f64(double *y, double a, const double *x, size_t n);
/* p = round(b*p) + r: linearsolver.rs:58-59 */
void orc_vec_xpby_f32(float *p, float b, const float *r, size_t n);
void orc_vec_xpby_f64(double *p, double b, const double *r, size_t n);

/* ---- reductions: vector.rs:50-58 (Iterator::sum == left fold from 0) --- */
float orc_dot_f32(const float *x, const float *y, size_t n);
double orc_dot_f64(const double *x, const double *y, size_t n);
float orc_norm_squared_f32(const float *x, size_t n);
double orc_norm_squared_f64(const double *x, size_t n);
double orc_norm_f32(const float *x, size_t n); /* vector.rs:61-63 */
double orc_norm_f64(const double *x, size_t n);

/* ---- ConjugateGradient::solve: linearsolver.rs:27-61 ------------------- */
/* iters_out = number of loop bodies entered (a body that breaks counts);
 * rr_out = last r.norm_squared() widened to f64. */
int orc_cg_f32(size_t n_rows, size_t n_cols, const uint32_t *offset_rows,
               const uint32_t *columns, const float *values, const float *b, size_t b_len,
               float *x, size_t x_len, double tol, size_t iter_max, size_t *iters_out,
               double *rr_out);
int orc_cg_f64(size_t n_rows, size_t n_cols, const uint32_t *offset_rows,
               const uint32_t *columns, const double *values, const double *b, size_t b_len,
               double *x, size_t x_len, double tol, size_t iter_max, size_t *iters_out,
               double *rr_out);
/* Jacobi-preconditioned CG: an EXTENSION (not in the reference) -- the recurrence above with
 * z = r / diag(A), diag_i = get(i, i); same guards, stop rule and arithmetic conventions. */
int orc_pcg_jacobi_f32(size_t n_rows, size_t n_cols, const uint32_t *offset_rows,
                       const uint32_t *columns, const float *values, const float *b, size_t b_len,
                       float *x, size_t x_len, double tol, size_t iter_max, size_t *iters_out,
                       double *rr_out);
int orc_pcg_jacobi_f64(size_t n_rows, size_t n_cols, const uint32_t *offset_rows,
                       const uint32_t *columns, const double *values, const double *b, size_t b_len,
                       double *x, size_t x_len, double tol, size_t iter_max, size_t *iters_out,
                       double *rr_out);

/* ---- SparseMatPar row-block arithmetic: sparsemat_par.rs:20-35 --------- */
/* R = max_n_rows / n_blocks (:21).  block/row of a global row as the
 * reference computes it (:31-35, clamp to n_blocks -- its off-by-one kept). */
size_t orc_par_rows_per_block(size_t n_blocks, size_t max_n_rows);
void orc_par_block_and_row(size_t n_blocks, size_t rows_per_block, size_t row,
                           size_t *block_out, size_t *row_out);

/* ---- synthetic workloads (spec in DESIGN.md "Synthetic inputs") ---- */
uint64_t orc_splitmix64(uint64_t z);
void orc_gen_x_f32(uint64_t seed, size_t begin, size_t n, float *x);
void orc_gen_x_f64(uint64_t seed, size_t begin, size_t n, double *x);
/* fixed `k` nnz per row, rows [row_begin,row_end) of an n x n matrix; offsets
 * are rebased to the block (offset_rows[0]==0), columns stay global.
 * pattern 0 = banded-stratified (ascending), 1 = uniform (draw order),
 * 2 = contiguous band (k consecutive columns around the diagonal). */
void orc_gen_fixed_f32(uint64_t seed, int pattern, size_t n, uint32_t k, size_t row_begin,
                       size_t row_end, uint32_t *offset_rows, uint32_t *columns, float *values);
void orc_gen_fixed_f64(uint64_t seed, int pattern, size_t n, uint32_t k, size_t row_begin,
                       size_t row_end, uint32_t *offset_rows, uint32_t *columns, double *values);
/* power-law row lengths in [1,kmax], P(k) ~ k^-alpha; cdf has kmax u32 entries */
void orc_powerlaw_cdf(uint32_t kmax, double alpha, uint32_t *cdf);
void orc_gen_powerlaw_lengths(uint64_t seed, size_t row_begin, size_t row_end, uint32_t kmax,
                              const uint32_t *cdf, uint32_t *lengths);
/* given offsets (len rows+1, rebased), fill uniform columns + values */
void orc_gen_fill_f32(uint64_t seed, size_t n_cols, size_t row_begin, size_t row_end,
                      const uint32_t *offset_rows, uint32_t *columns, float *values);
void orc_gen_fill_f64(uint64_t seed, size_t n_cols, size_t row_begin, size_t row_end,
                      const uint32_t *offset_rows, uint32_t *columns, double *values);
/* 5-point (nx*ny) / 7-point (nx*ny*nz) Laplacians, natural ordering, columns
 * ascending, diag 4 / 6, off-diag -1.  Return nnz; arrays may be NULL to count. */
size_t orc_laplace2d_f32(size_t nx, size_t ny, uint32_t *offset_rows, uint32_t *columns, float *values);
size_t orc_laplace3d_f32(size_t nx, size_t ny, size_t nz, uint32_t *offset_rows, uint32_t *columns, float *values);
size_t orc_laplace2d_f64(size_t nx, size_t ny, uint32_t *offset_rows, uint32_t *columns, double *values);
size_t orc_laplace3d_f64(size_t nx, size_t ny, size_t nz, uint32_t *offset_rows, uint32_t *columns, double *values);
/* rows [row_begin, row_end) of the 7-point matrix: offsets rebased to 0, global columns */
size_t orc_laplace3d_rows_f32(size_t nx, size_t ny, size_t nz, size_t row_begin, size_t row_end, uint32_t *offset_rows, uint32_t *columns, float *values);
size_t orc_laplace3d_rows_f64(size_t nx, size_t ny, size_t nz, size_t row_begin, size_t row_end, uint32_t *offset_rows, uint32_t *columns, double *values);

/* ---- assembly: add_to/set stream on a SparseMatIndexList, then to_crs() ----
 * sparsemat_indexlist.rs:29-53,61-63,158-164; indexlist.rs:62-83; sparsematrix.rs:226-233;
 * sparsemat_crs.rs:24-50.  ops[k] 0 = add_to, 1 = set (NULL: all add_to).  offset_rows needs
 * max(row)+2 entries, columns / values n entries. */
int orc_assemble_f32(size_t n, const uint32_t *rows, const uint32_t *cols, const float *vals,
                     const uint8_t *ops, size_t *n_rows_out, size_t *n_cols_out, size_t *nnz_out,
                     uint32_t *offset_rows, uint32_t *columns, float *values);
int orc_assemble_f64(size_t n, const uint32_t *rows, const uint32_t *cols, const double *vals,
                     const uint8_t *ops, size_t *n_rows_out, size_t *n_cols_out, size_t *nnz_out,
                     uint32_t *offset_rows, uint32_t *columns, double *values);
/* The same stream replayed on a SparseMatCRS (the container Trait::transpose / prod fill through `set`,
 * sparsematrix.rs:174-210): sparsemat_crs.rs:54-92,143-149 literally, incl. the first-push quirk.
 * offset_rows needs max(row)+2 entries, columns / values n entries; *nnz_out = entries reachable through
 * iter_row, *stored_out = columns.len() (an orphaned first entry stays at the tail).  O(n * nnz). */
int orc_crs_replay_f32(size_t n, const uint32_t *rows, const uint32_t *cols, const float *vals,
                       const uint8_t *ops, size_t *n_rows_out, size_t *n_cols_out, size_t *nnz_out,
                       size_t *stored_out, uint32_t *offset_rows, uint32_t *columns, float *values);
int orc_crs_replay_f64(size_t n, const uint32_t *rows, const uint32_t *cols, const double *vals,
                       const uint8_t *ops, size_t *n_rows_out, size_t *n_cols_out, size_t *nnz_out,
                       size_t *stored_out, uint32_t *offset_rows, uint32_t *columns, double *values);
/* SparseMatrix::prod (sparsematrix.rs:186-210), literally, up to the container: writes the call stream
 * ret.set(i, j, sum) (sums != 0, i then j ascending) to ops_* (capacity cap; ORC_ERR_CAPACITY beyond), to be
 * replayed with orc_crs_replay.  Dimension rule :188-190 -> ORC_ERR_SIZE_MISMATCH.  O(n_rows * n_cols). */
int orc_crs_prod_ops_f32(size_t a_rows, size_t a_cols, const uint32_t *a_off, const uint32_t *a_col,
                         const float *a_val, size_t b_rows, size_t b_cols, const uint32_t *b_off,
                         const uint32_t *b_col, const float *b_val, size_t cap, size_t *n_out,
                         uint32_t *ops_rows, uint32_t *ops_cols, float *ops_vals);
int orc_crs_prod_ops_f64(size_t a_rows, size_t a_cols, const uint32_t *a_off, const uint32_t *a_col,
                         const double *a_val, size_t b_rows, size_t b_cols, const uint32_t *b_off,
                         const uint32_t *b_col, const double *b_val, size_t cap, size_t *n_out,
                         uint32_t *ops_rows, uint32_t *ops_cols, double *ops_vals);
/* Sortable::sort_row (sparsemat_crs.rs:163-172) on every row: stable by column */
void orc_crs_sort_rows_f32(size_t n_rows, const uint32_t *offset_rows, uint32_t *columns, float *values);
void orc_crs_sort_rows_f64(size_t n_rows, const uint32_t *offset_rows, uint32_t *columns, double *values);

/* ---- merge-path coordinates (restates the build's own search) ---- */
/* For diagonal d over (row_end_offsets = offset_rows+1 [n_rows]) x (0..nnz):
 * returns the row coordinate; the nnz coordinate is d - row. */
void orc_merge_path_search(size_t n_rows, size_t nnz, const uint32_t *offset_rows,
                           size_t n_diagonals, const uint64_t *diagonals, uint32_t *row_out,
                           uint32_t *nnz_out);

#ifdef __cplusplus
}
#endif
#endif
