/*
 * sparsemat_oracle.c -- TEST INFRASTRUCTURE ONLY (see sparsemat_oracle.h).
 *
 * Plain-C, single-thread restatement of the reference crate's arithmetic for
 * the CSR SpMV / BLAS-1 / CG path.  Build with -ffp-contract=off: the reference
 * (rustc, no fast-math) rounds every multiply and every add separately.
 * Parity: pinned by the reference's own known-answer tests (tests/golden/).
 */
#include "sparsemat_oracle.h"

#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================= *
 * SpMV.  sparsematrix.rs:146-158: for each row i, sum = T::zero(); for each
 * (col,val) of iter_row(i) in STORAGE order: sum += rhs.get(col) * val; then
 * ret.set(i,sum).  iter_row = sparsemat_crs.rs:102-110 (slice
 * [offset_rows[i], offset_rows[i+1])).  rhs.get = densevec.rs:40-42 (bounds
 * checked -> panic; here ORC_ERR_INDEX_OOB).  u32 -> usize: types.rs:32-36.
 * ======================================================================= */
#define DEF_SPMV(SUF, T)                                                                    \
    int orc_spmv_rows_##SUF(size_t row_begin, size_t row_end, const uint32_t *offset_rows,  \
                            const uint32_t *columns, const T *values, const T *x,           \
                            size_t x_len, T *y) {                                           \
        for (size_t i = row_begin; i < row_end; ++i) {                                      \
            size_t start = (size_t)offset_rows[i];                                          \
            size_t end = (size_t)offset_rows[i + 1];                                        \
            T sum = (T)0;                                                                   \
            for (size_t k = start; k < end; ++k) {                                          \
                size_t j = (size_t)columns[k];                                              \
                if (j >= x_len) return ORC_ERR_INDEX_OOB;                                   \
                T prod = x[j] * values[k]; /* rounded product (rhs.get(j) * val) */         \
                sum = sum + prod;          /* rounded add (sum += ...)            */         \
            }                                                                               \
            y[i] = sum;                                                                     \
        }                                                                                   \
        return ORC_OK;                                                                      \
    }                                                                                       \
    int orc_spmv_##SUF(size_t n_rows, const uint32_t *offset_rows, const uint32_t *columns, \
                       const T *values, const T *x, size_t x_len, T *y) {                   \
        return orc_spmv_rows_##SUF(0, n_rows, offset_rows, columns, values, x, x_len, y);   \
    }                                                                                       \
    /* NOT reference behaviour (the reference's mvp is serial; its mvp_par is commented out,   \
     * sparsemat_par.rs:37-68): the same per-row loop with the rows spread over all host     \
     * cores.  Reported by bench.py beside the one-core figure so that the GPU/CPU ratio is   \
     * not inflated by the reference being single-threaded.  Per row the       \
     * arithmetic is identical, so y equals orc_spmv's bit for bit.  threads <= 0: OpenMP's   \
     * default; the count actually used comes back through *threads_out. */                                                              \
    int orc_spmv_omp_##SUF(size_t n_rows, const uint32_t *offset_rows,                      \
                           const uint32_t *columns, const T *values, const T *x,            \
                           size_t x_len, T *y, int threads, int *threads_out) {             \
        int bad = 0, nt = 1;                                                                \
        if (threads <= 0) threads = omp_get_max_threads();                                  \
        _Pragma("omp parallel num_threads(threads)")                                        \
        {                                                                                   \
            _Pragma("omp single") nt = omp_get_num_threads();                               \
            _Pragma("omp for schedule(static) reduction(|:bad)")                            \
            for (size_t i = 0; i < n_rows; ++i)                                             \
                bad |= orc_spmv_rows_##SUF(i, i + 1, offset_rows, columns, values, x,       \
                                           x_len, y) != ORC_OK;                             \
        }                                                                                   \
        if (threads_out) *threads_out = nt;                                                 \
        return bad ? ORC_ERR_INDEX_OOB : ORC_OK;                                            \
    }                                                                                       \
    void orc_spmv_abs_##SUF(size_t n_rows, const uint32_t *offset_rows,                     \
                            const uint32_t *columns, const T *values, const T *x,           \
                            double *out) {                                                  \
        for (size_t i = 0; i < n_rows; ++i) {                                               \
            double s = 0.0;                                                                 \
            for (size_t k = offset_rows[i]; k < (size_t)offset_rows[i + 1]; ++k)            \
                s += fabs((double)x[columns[k]] * (double)values[k]);                       \
            out[i] = s;                                                                     \
        }                                                                                   \
    }                                                                                       \
    /* sparsematrix.rs:161-171: sum += lhs.get(i) * val * rhs.get(j), left to right */     \
    T orc_mat_inner_prod_##SUF(size_t n_rows, const uint32_t *offset_rows,                  \
                               const uint32_t *columns, const T *values, const T *lhs,      \
                               const T *rhs) {                                              \
        T sum = (T)0;                                                                       \
        for (size_t i = 0; i < n_rows; ++i)                                                 \
            for (size_t k = offset_rows[i]; k < (size_t)offset_rows[i + 1]; ++k) {          \
                T t = lhs[i] * values[k];                                                   \
                t = t * rhs[columns[k]];                                                    \
                sum = sum + t;                                                              \
            }                                                                               \
        return sum;                                                                         \
    }

DEF_SPMV(f32, float)
DEF_SPMV(f64, double)

/* ======================================================================= *
 * DenseVec ops.  densevec.rs:51-58 add, :60-67 sub (panic "Dimension
 * mismatch" iff self.dim() < rhs.dim(); zip truncates to the shorter), :69-73
 * scale.  axpy / xpby restate the composite updates of linearsolver.rs:47,49
 * and :58-59 with the reference's rounding order.
 * ======================================================================= */
#define DEF_VEC(SUF, T)                                                        \
    int orc_vec_add_##SUF(T *x, size_t nx, const T *y, size_t ny) {            \
        if (nx < ny) return ORC_ERR_SIZE_MISMATCH;                             \
        for (size_t i = 0; i < ny; ++i) x[i] = x[i] + y[i];                    \
        return ORC_OK;                                                         \
    }                                                                          \
    int orc_vec_sub_##SUF(T *x, size_t nx, const T *y, size_t ny) {            \
        if (nx < ny) return ORC_ERR_SIZE_MISMATCH;                             \
        for (size_t i = 0; i < ny; ++i) x[i] = x[i] - y[i];                    \
        return ORC_OK;                                                         \
    }                                                                          \
    void orc_vec_scale_##SUF(T *x, size_t n, T a) {                            \
        for (size_t i = 0; i < n; ++i) x[i] = x[i] * a;                        \
    }                                                                          \
    void orc_vec_axpy_##SUF(T *y, T a, const T *x, size_t n) {                 \
        for (size_t i = 0; i < n; ++i) {                                       \
            T t = x[i] * a; /* p.clone() * alpha */                            \
            y[i] = y[i] + t; /* *x += ...         */                           \
        }                                                                      \
    }                                                                          \
    void orc_vec_xpby_##SUF(T *p, T b, const T *r, size_t n) {                 \
        for (size_t i = 0; i < n; ++i) {                                       \
            T t = p[i] * b; /* p.scale(beta) */                                \
            p[i] = t + r[i]; /* p.add(&r)    */                                \
        }                                                                      \
    }                                                                          \
    /* vector.rs:50-53: zip().map(x*y).sum() -- left fold from zero */         \
    T orc_dot_##SUF(const T *x, const T *y, size_t n) {                        \
        T s = (T)0;                                                            \
        for (size_t i = 0; i < n; ++i) {                                       \
            T t = x[i] * y[i];                                                 \
            s = s + t;                                                         \
        }                                                                      \
        return s;                                                              \
    }                                                                          \
    /* vector.rs:56-58 */                                                      \
    T orc_norm_squared_##SUF(const T *x, size_t n) {                           \
        T s = (T)0;                                                            \
        for (size_t i = 0; i < n; ++i) {                                       \
            T t = x[i] * x[i];                                                 \
            s = s + t;                                                         \
        }                                                                      \
        return s;                                                              \
    }                                                                          \
    /* vector.rs:61-63: f64::sqrt(self.norm_squared().into()) */               \
    double orc_norm_##SUF(const T *x, size_t n) {                              \
        return sqrt((double)orc_norm_squared_##SUF(x, n));                     \
    }

DEF_VEC(f32, float)
DEF_VEC(f64, double)

/* ======================================================================= *
 * ConjugateGradient::solve.  linearsolver.rs:27-61.
 *   :30-32 n_rows != n_cols           -> "Matrix is not symmetric"
 *   :33-36 n_rows != b.dim()/x.dim()  -> "Matrix and vector size mismatch"
 *   :38    r = b - A x ; :39 p = r ; :40 rr = r.r
 *   :41-60 loop: Ap (:43), alpha = rr / p.Ap (:45), x += p*alpha (:47),
 *          r -= Ap*alpha (:49), rr (:50-51), stop if sqrt(f64(rr)) < tol
 *          (:52-54, BEFORE the beta update), beta = rr/rr_prev (:56),
 *          p = beta*p + r (:58-59).
 * ======================================================================= */
#define DEF_CG(SUF, T)                                                                       \
    int orc_cg_##SUF(size_t n_rows, size_t n_cols, const uint32_t *offset_rows,              \
                     const uint32_t *columns, const T *values, const T *b, size_t b_len,     \
                     T *x, size_t x_len, double tol, size_t iter_max, size_t *iters_out,     \
                     double *rr_out) {                                                       \
        if (n_rows != n_cols) return ORC_ERR_NOT_SQUARE;                                     \
        if (n_rows != b_len || n_rows != x_len) return ORC_ERR_SIZE_MISMATCH;                \
        size_t n = n_rows;                                                                   \
        T *r = (T *)malloc((n ? n : 1) * sizeof(T));                                         \
        T *p = (T *)malloc((n ? n : 1) * sizeof(T));                                         \
        T *ap = (T *)malloc((n ? n : 1) * sizeof(T));                                        \
        int rc = orc_spmv_##SUF(n, offset_rows, columns, values, x, x_len, ap);              \
        size_t iters = 0;                                                                    \
        T rr = (T)0;                                                                         \
        if (rc == ORC_OK) {                                                                  \
            for (size_t i = 0; i < n; ++i) r[i] = b[i] - ap[i];                              \
            memcpy(p, r, n * sizeof(T));                                                     \
            rr = orc_norm_squared_##SUF(r, n);                                               \
            for (size_t k = 0; k < iter_max; ++k) {                                          \
                ++iters;                                                                     \
                rc = orc_spmv_##SUF(n, offset_rows, columns, values, p, n, ap);              \
                if (rc != ORC_OK) break;                                                     \
                T alpha = rr / orc_dot_##SUF(p, ap, n);                                      \
                orc_vec_axpy_##SUF(x, alpha, p, n);                                          \
                for (size_t i = 0; i < n; ++i) {                                             \
                    T t = ap[i] * alpha; /* mat_p * alpha */                                 \
                    r[i] = r[i] - t;     /* r -= ...      */                                 \
                }                                                                            \
                T rr_prev = rr;                                                              \
                rr = orc_norm_squared_##SUF(r, n);                                           \
                if (sqrt((double)rr) < tol) break;                                           \
                T beta = rr / rr_prev;                                                       \
                orc_vec_xpby_##SUF(p, beta, r, n);                                           \
            }                                                                                \
        }                                                                                    \
        if (iters_out) *iters_out = iters;                                                   \
        if (rr_out) *rr_out = (double)rr;                                                    \
        free(r);                                                                             \
        free(p);                                                                             \
        free(ap);                                                                            \
        return rc;                                                                           \
    }

DEF_CG(f32, float)
DEF_CG(f64, double)

/* ======================================================================= *
 * SparseMatPar.  sparsemat_par.rs:20-28 (R = max_n_rows / n_blocks) and
 * :31-35 (block = min(row / R, n_blocks), row_id = row - block*R).
 * ======================================================================= */
size_t orc_par_rows_per_block(size_t n_blocks, size_t max_n_rows) { return max_n_rows / n_blocks; }

void orc_par_block_and_row(size_t n_blocks, size_t rows_per_block, size_t row, size_t *block_out,
                           size_t *row_out) {
    size_t b = row / rows_per_block;
    if (b > n_blocks) b = n_blocks; /* the reference clamps to n_blocks, not n_blocks-1 */
    *block_out = b;
    *row_out = row - b * rows_per_block;
}

/* ======================================================================= *
 * Synthetic workloads (the build's own; spec in DESIGN.md).  Counter based so
 * the device generator reproduces them bit for bit.
 * ======================================================================= */
#define ORC_GOLD 0x9E3779B97F4A7C15ull

uint64_t orc_splitmix64(uint64_t z) {
    z += ORC_GOLD;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline uint64_t rowkey(uint64_t seed, uint64_t row) { return orc_splitmix64(seed ^ (row * ORC_GOLD)); }
static inline float h2f32(uint64_t h) { return (float)(h >> 40) * 0x1p-23f - 1.0f; }
static inline double h2f64(uint64_t h) { return (double)(h >> 11) * 0x1p-52 - 1.0; }

void orc_gen_x_f32(uint64_t seed, size_t begin, size_t n, float *x) {
    for (size_t j = 0; j < n; ++j) x[j] = h2f32(rowkey(seed, begin + j));
}
void orc_gen_x_f64(uint64_t seed, size_t begin, size_t n, double *x) {
    for (size_t j = 0; j < n; ++j) x[j] = h2f64(rowkey(seed, begin + j));
}

#define DEF_GEN(SUF, T, H2V)                                                                  \
    void orc_gen_fixed_##SUF(uint64_t seed, int pattern, size_t n, uint32_t k,                \
                             size_t row_begin, size_t row_end, uint32_t *offset_rows,         \
                             uint32_t *columns, T *values) {                                  \
        uint64_t s = n / k;                                                                   \
        if (s < 1) s = 1;                                                                     \
        if (s > 256) s = 256;                                                                 \
        uint64_t w = s * k;                                                                   \
        _Pragma("omp parallel for schedule(static)")                                          \
        for (size_t row = row_begin; row < row_end; ++row) {                                  \
            uint64_t rk = rowkey(seed, row);                                                  \
            size_t o = (row - row_begin) * (size_t)k;                                         \
            offset_rows[row - row_begin] = (uint32_t)o;                                       \
            int64_t base = (int64_t)row - (int64_t)(w / 2);                                   \
            if (base > (int64_t)n - (int64_t)w) base = (int64_t)n - (int64_t)w;               \
            if (base < 0) base = 0;                                                           \
            if (pattern == 3) { /* SURVEY 8(d) "banded" to the letter: k distinct columns drawn */ \
                /* without replacement from [row - 4096, row + 4096] within [0, n), stored ascending */ \
                uint64_t lo = row > 4096 ? row - 4096 : 0;                                    \
                uint64_t hi = row + 4096 < n - 1 ? row + 4096 : n - 1;                        \
                uint64_t ww = hi - lo + 1;                                                    \
                uint32_t *c = columns + o;                                                    \
                uint32_t got = 0;                                                             \
                for (uint64_t t = 0; got < k; ++t) { /* draw t: accepted unless already drawn */ \
                    uint32_t cand = (uint32_t)(lo + orc_splitmix64(rk + 2ull * t) % ww);      \
                    uint32_t q = 0;                                                           \
                    while (q < got && c[q] != cand) ++q;                                      \
                    if (q == got) c[got++] = cand;                                            \
                }                                                                             \
                for (uint32_t a = 1; a < k; ++a) { /* ascending (insertion sort) */           \
                    uint32_t v = c[a];                                                        \
                    uint32_t b = a;                                                           \
                    while (b > 0 && c[b - 1] > v) { c[b] = c[b - 1]; --b; }                   \
                    c[b] = v;                                                                 \
                }                                                                             \
                for (uint32_t j = 0; j < k; ++j) values[o + j] = H2V(orc_splitmix64(rk + 2ull * j + 1ull)); \
                continue;                                                                     \
            }                                                                                 \
            for (uint32_t j = 0; j < k; ++j) {                                                \
                uint64_t hc = orc_splitmix64(rk + 2ull * j);                                  \
                uint64_t hv = orc_splitmix64(rk + 2ull * j + 1ull);                           \
                uint64_t c = pattern == 0 ? (uint64_t)base + j * s + hc % s : hc % n;         \
                if (pattern == 2) {                                                           \
                    int64_t b2 = (int64_t)row - (int64_t)(k / 2);                             \
                    if (b2 > (int64_t)n - (int64_t)k) b2 = (int64_t)n - (int64_t)k;           \
                    if (b2 < 0) b2 = 0;                                                       \
                    c = (uint64_t)b2 + j;                                                     \
                }                                                                             \
                columns[o + j] = (uint32_t)c;                                                 \
                values[o + j] = H2V(hv);                                                      \
            }                                                                                 \
        }                                                                                     \
        offset_rows[row_end - row_begin] = (uint32_t)((row_end - row_begin) * (size_t)k);     \
    }                                                                                         \
    void orc_gen_fill_##SUF(uint64_t seed, size_t n_cols, size_t row_begin, size_t row_end,   \
                            const uint32_t *offset_rows, uint32_t *columns, T *values) {      \
        _Pragma("omp parallel for schedule(dynamic, 4096)")                                   \
        for (size_t row = row_begin; row < row_end; ++row) {                                  \
            uint64_t rk = rowkey(seed, row);                                                  \
            size_t o = offset_rows[row - row_begin];                                          \
            uint32_t len = offset_rows[row - row_begin + 1] - offset_rows[row - row_begin];   \
            for (uint32_t j = 0; j < len; ++j) {                                              \
                uint64_t hc = orc_splitmix64(rk + 2ull * j);                                  \
                uint64_t hv = orc_splitmix64(rk + 2ull * j + 1ull);                           \
                columns[o + j] = (uint32_t)(hc % n_cols);                                     \
                values[o + j] = H2V(hv);                                                      \
            }                                                                                 \
        }                                                                                     \
    }

DEF_GEN(f32, float, h2f32)
DEF_GEN(f64, double, h2f64)

void orc_powerlaw_cdf(uint32_t kmax, double alpha, uint32_t *cdf) {
    double z = 0.0;
    for (uint32_t k = 1; k <= kmax; ++k) z += pow((double)k, -alpha);
    double acc = 0.0;
    for (uint32_t k = 1; k <= kmax; ++k) {
        acc += pow((double)k, -alpha);
        double f = acc / z * 4294967296.0;
        cdf[k - 1] = f >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)f;
    }
    cdf[kmax - 1] = 0xFFFFFFFFu;
}

void orc_gen_powerlaw_lengths(uint64_t seed, size_t row_begin, size_t row_end, uint32_t kmax,
                              const uint32_t *cdf, uint32_t *lengths) {
    for (size_t row = row_begin; row < row_end; ++row) {
        uint32_t u = (uint32_t)(orc_splitmix64(rowkey(seed, row) ^ 0xA5A5A5A5A5A5A5A5ull) >> 32);
        /* count of entries cdf[i] <= u (upper bound), clamped to kmax-1 */
        uint32_t lo = 0, hi = kmax;
        while (lo < hi) {
            uint32_t mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        if (lo > kmax - 1) lo = kmax - 1;
        lengths[row - row_begin] = 1u + lo;
    }
}

#define DEF_LAPLACE(SUF, T)                                                                   \
    size_t orc_laplace2d_##SUF(size_t nx, size_t ny, uint32_t *offset_rows, uint32_t *columns,\
                               T *values) {                                                   \
        size_t nnz = 0;                                                                       \
        for (size_t j = 0; j < ny; ++j)                                                       \
            for (size_t i = 0; i < nx; ++i) {                                                 \
                size_t row = j * nx + i;                                                      \
                if (offset_rows) offset_rows[row] = (uint32_t)nnz;                            \
                long long nb[5] = {j > 0 ? (long long)(row - nx) : -1,                        \
                                   i > 0 ? (long long)(row - 1) : -1, (long long)row,         \
                                   i + 1 < nx ? (long long)(row + 1) : -1,                    \
                                   j + 1 < ny ? (long long)(row + nx) : -1};                  \
                for (int t = 0; t < 5; ++t)                                                   \
                    if (nb[t] >= 0) {                                                         \
                        if (columns) columns[nnz] = (uint32_t)nb[t];                          \
                        if (values) values[nnz] = t == 2 ? (T)4 : (T)-1;                      \
                        ++nnz;                                                                \
                    }                                                                         \
            }                                                                                 \
        if (offset_rows) offset_rows[nx * ny] = (uint32_t)nnz;                                \
        return nnz;                                                                           \
    }                                                                                         \
    size_t orc_laplace3d_##SUF(size_t nx, size_t ny, size_t nz, uint32_t *offset_rows,        \
                               uint32_t *columns, T *values) {                                \
        size_t nnz = 0;                                                                       \
        size_t nxy = nx * ny;                                                                 \
        for (size_t k = 0; k < nz; ++k)                                                       \
            for (size_t j = 0; j < ny; ++j)                                                   \
                for (size_t i = 0; i < nx; ++i) {                                             \
                    size_t row = k * nxy + j * nx + i;                                        \
                    if (offset_rows) offset_rows[row] = (uint32_t)nnz;                        \
                    long long nb[7] = {k > 0 ? (long long)(row - nxy) : -1,                   \
                                       j > 0 ? (long long)(row - nx) : -1,                    \
                                       i > 0 ? (long long)(row - 1) : -1, (long long)row,     \
                                       i + 1 < nx ? (long long)(row + 1) : -1,                \
                                       j + 1 < ny ? (long long)(row + nx) : -1,               \
                                       k + 1 < nz ? (long long)(row + nxy) : -1};             \
                    for (int t = 0; t < 7; ++t)                                               \
                        if (nb[t] >= 0) {                                                     \
                            if (columns) columns[nnz] = (uint32_t)nb[t];                      \
                            if (values) values[nnz] = t == 3 ? (T)6 : (T)-1;                  \
                            ++nnz;                                                            \
                        }                                                                     \
                }                                                                             \
        if (offset_rows) offset_rows[nxy * nz] = (uint32_t)nnz;                               \
        return nnz;                                                                           \
    }                                                                                         \
    /* rows [row_begin, row_end) of the same matrix: offsets rebased to 0 (row_end -          \
     * row_begin + 1 of them), GLOBAL columns -- a sample of BASELINE C4 without its 8 GB */  \
    size_t orc_laplace3d_rows_##SUF(size_t nx, size_t ny, size_t nz, size_t row_begin,        \
                                    size_t row_end, uint32_t *offset_rows, uint32_t *columns, \
                                    T *values) {                                              \
        size_t nnz = 0;                                                                       \
        size_t nxy = nx * ny;                                                                 \
        for (size_t row = row_begin; row < row_end; ++row) {                                  \
            size_t k = row / nxy, j = row % nxy / nx, i = row % nx;                           \
            if (offset_rows) offset_rows[row - row_begin] = (uint32_t)nnz;                    \
            long long nb[7] = {k > 0 ? (long long)(row - nxy) : -1,                           \
                               j > 0 ? (long long)(row - nx) : -1,                            \
                               i > 0 ? (long long)(row - 1) : -1, (long long)row,             \
                               i + 1 < nx ? (long long)(row + 1) : -1,                        \
                               j + 1 < ny ? (long long)(row + nx) : -1,                       \
                               k + 1 < nz ? (long long)(row + nxy) : -1};                     \
            for (int t = 0; t < 7; ++t)                                                       \
                if (nb[t] >= 0) {                                                             \
                    if (columns) columns[nnz] = (uint32_t)nb[t];                              \
                    if (values) values[nnz] = t == 3 ? (T)6 : (T)-1;                          \
                    ++nnz;                                                                    \
                }                                                                             \
        }                                                                                     \
        if (offset_rows) offset_rows[row_end - row_begin] = (uint32_t)nnz;                    \
        return nnz;                                                                           \
    }

DEF_LAPLACE(f32, float)
DEF_LAPLACE(f64, double)

/* ======================================================================= *
 * Merge-path diagonal search (restates the build's K2 partitioner, not a
 * reference function): list A = row END offsets (offset_rows[1..n_rows]),
 * list B = the naturals 0..nnz-1.  On diagonal d find the split (row, d-row)
 * with A[row-1] <= B[d-row] ... by binary search on "A[mid] <= d-1-mid".
 * ======================================================================= */
void orc_merge_path_search(size_t n_rows, size_t nnz, const uint32_t *offset_rows,
                           size_t n_diagonals, const uint64_t *diagonals, uint32_t *row_out,
                           uint32_t *nnz_out) {
    const uint32_t *row_end = offset_rows + 1;
    for (size_t t = 0; t < n_diagonals; ++t) {
        uint64_t d = diagonals[t];
        if (d > (uint64_t)n_rows + nnz) d = (uint64_t)n_rows + nnz;
        uint64_t lo = d > nnz ? d - nnz : 0;
        uint64_t hi = d < n_rows ? d : n_rows;
        while (lo < hi) {
            uint64_t mid = (lo + hi) >> 1;
            if ((uint64_t)row_end[mid] <= d - 1 - mid) lo = mid + 1; else hi = mid;
        }
        row_out[t] = (uint32_t)lo;
        nnz_out[t] = (uint32_t)(d - lo);
    }
}

/* ======================================================================= *
 * Assembly: a stream of add_to / set calls on a SparseMatIndexList, then
 * to_crs().  Follows the reference's containers step by step:
 *   get_mut(i,j)  sparsemat_indexlist.rs:158-164: find_index, else push(i,j,0)
 *   find_index    sparsemat_indexlist.rs:29-42: walk the row's list, first match
 *   push          sparsemat_indexlist.rs:45-53 + IndexList::push indexlist.rs:62-83
 *                 (pos_start grows to row+1, the new entry is appended to the
 *                 END of the row's list; n_cols = max(n_cols, j+1))
 *   add_to / set  sparsematrix.rs:231-233 / :226-228 (`+=` / `=` on the entry)
 *   to_crs        sparsemat_indexlist.rs:61-63 -> sparsemat_crs.rs:24-50: rows
 *                 0..n_rows, each row in list (= first-insertion) order; an
 *                 index list without entries gives SparseMatCRS::new() (no rows).
 * The reference walks to the tail of the row's list on every push
 * (indexlist.rs:73-79); a per-row tail pointer gives the same list.
 * ops[k]: 0 = add_to, 1 = set (NULL: all add_to).  Outputs: offset_rows needs
 * max(row)+2 entries, columns/values n entries.  Returns ORC_OK.
 * ======================================================================= */
#define ORC_UNSET 0xFFFFFFFFu
#define DEF_ASSEMBLE(SUF, T)                                                                        \
    int orc_assemble_##SUF(size_t n, const uint32_t *rows, const uint32_t *cols, const T *vals,     \
                           const uint8_t *ops, size_t *n_rows_out, size_t *n_cols_out,              \
                           size_t *nnz_out, uint32_t *offset_rows, uint32_t *columns, T *values) {  \
        size_t n_rows = 0, n_cols = 0, n_entries = 0;                                               \
        for (size_t k = 0; k < n; ++k)                                                              \
            if ((size_t)rows[k] + 1 > n_rows) n_rows = (size_t)rows[k] + 1;                         \
        uint32_t *pos_start = (uint32_t *)malloc((n_rows + 1) * sizeof(uint32_t));                 \
        uint32_t *tail = (uint32_t *)malloc((n_rows + 1) * sizeof(uint32_t));                       \
        uint32_t *next = (uint32_t *)malloc((n + 1) * sizeof(uint32_t));                            \
        uint32_t *ecol = (uint32_t *)malloc((n + 1) * sizeof(uint32_t));                            \
        T *eval = (T *)malloc((n + 1) * sizeof(T));                                                 \
        if (!pos_start || !tail || !next || !ecol || !eval) {                                       \
            free(pos_start); free(tail); free(next); free(ecol); free(eval);                        \
            return ORC_ERR_SIZE_MISMATCH;                                                           \
        }                                                                                           \
        for (size_t r = 0; r < n_rows; ++r) pos_start[r] = ORC_UNSET;                               \
        for (size_t k = 0; k < n; ++k) {                                                            \
            const uint32_t i = rows[k], j = cols[k];                                                \
            uint32_t index = ORC_UNSET;                                                             \
            for (uint32_t it = pos_start[i]; it != ORC_UNSET; it = next[it])                        \
                if (ecol[it] == j) { index = it; break; }                                           \
            if (index == ORC_UNSET) { /* push(i, j, T::zero()) */                                   \
                if ((size_t)j + 1 > n_cols) n_cols = (size_t)j + 1;                                 \
                index = (uint32_t)n_entries++;                                                      \
                next[index] = ORC_UNSET;                                                            \
                if (pos_start[i] == ORC_UNSET) pos_start[i] = index; else next[tail[i]] = index;    \
                tail[i] = index;                                                                    \
                ecol[index] = j;                                                                    \
                eval[index] = (T)0;                                                                 \
            }                                                                                       \
            if (ops && ops[k]) eval[index] = vals[k]; else eval[index] = eval[index] + vals[k];     \
        }                                                                                           \
        size_t nnz = 0;                                                                             \
        if (n_entries > 0) {                                                                        \
            for (size_t r = 0; r < n_rows; ++r) {                                                   \
                offset_rows[r] = (uint32_t)nnz;                                                     \
                for (uint32_t it = pos_start[r]; it != ORC_UNSET; it = next[it]) {                  \
                    columns[nnz] = ecol[it];                                                        \
                    values[nnz] = eval[it];                                                         \
                    ++nnz;                                                                          \
                }                                                                                   \
            }                                                                                       \
            offset_rows[n_rows] = (uint32_t)nnz;                                                    \
        } else {                                                                                    \
            n_rows = 0;                                                                             \
        }                                                                                           \
        *n_rows_out = n_rows; *n_cols_out = n_cols; *nnz_out = nnz;                                 \
        free(pos_start); free(tail); free(next); free(ecol); free(eval);                            \
        return ORC_OK;                                                                              \
    }

DEF_ASSEMBLE(f32, float)
DEF_ASSEMBLE(f64, double)

/* Sortable::sort_row (sparsemat_crs.rs:163-172) applied to every row: the row's (column, value) pairs are
 * sorted by column with slice::sort_by, a STABLE sort -- duplicates of a column keep their storage order.
 * Insertion sort is stable and rows are short. */
#define DEF_SORT_ROWS(SUF, T)                                                                       \
    void orc_crs_sort_rows_##SUF(size_t n_rows, const uint32_t *offset_rows, uint32_t *columns,     \
                                 T *values) {                                                       \
        for (size_t r = 0; r < n_rows; ++r) {                                                       \
            const size_t a = offset_rows[r], b = offset_rows[r + 1];                                \
            for (size_t k = a + 1; k < b; ++k) {                                                    \
                const uint32_t c = columns[k];                                                      \
                const T v = values[k];                                                              \
                size_t p = k;                                                                       \
                while (p > a && columns[p - 1] > c) {                                               \
                    columns[p] = columns[p - 1];                                                    \
                    values[p] = values[p - 1];                                                      \
                    --p;                                                                            \
                }                                                                                   \
                columns[p] = c;                                                                     \
                values[p] = v;                                                                      \
            }                                                                                       \
        }                                                                                           \
    }

DEF_SORT_ROWS(f32, float)
DEF_SORT_ROWS(f64, double)

/* The same operation stream replayed on a SparseMatCRS itself (the container Trait::transpose and prod fill
 * through `set`, sparsematrix.rs:174-184,186-210): get_mut sparsemat_crs.rs:143-149 -> find_index :54-67 (first
 * match in storage order, only rows < n_rows) / push :71-92, restated literally:
 *   - push inserts at the START of the row (index = offset_rows[i]) and bumps every later offset;
 *   - the very first push sizes offset_rows to i + 2 but leaves n_rows == 0 (:75-76), so the second operation never
 *     finds anything and pushes again; Vec::resize(i + 2, last) then TRUNCATES offset_rows when its row is smaller
 *     than the first one's -- the first entry stays in columns / values (n_non_zero_entries counts it, :132-134)
 *     but no row addresses it any more;
 *   - n_cols = largest pushed column + 1.
 * Outputs: offset_rows[0..n_rows] (needs max(row) + 2 entries), columns / values [stored] in storage order (need n
 * entries); *nnz_out = offset_rows[n_rows] (what iter_row reaches), *stored_out = columns.len().  O(n * nnz): small
 * cases only. */
#define DEF_CRS_REPLAY(SUF, T)                                                                      \
    int orc_crs_replay_##SUF(size_t n, const uint32_t *rows, const uint32_t *cols, const T *vals,   \
                             const uint8_t *ops, size_t *n_rows_out, size_t *n_cols_out,            \
                             size_t *nnz_out, size_t *stored_out, uint32_t *offset_rows,            \
                             uint32_t *columns, T *values) {                                        \
        size_t n_rows = 0, n_cols = 0, len_off = 0, stored = 0;                                     \
        for (size_t k = 0; k < n; ++k) {                                                            \
            const size_t i = rows[k], j = cols[k];                                                  \
            size_t index = ORC_UNSET;                                                               \
            if (i < n_rows)                                                                         \
                for (size_t q = offset_rows[i]; q < offset_rows[i + 1]; ++q)                        \
                    if (columns[q] == j) { index = q; break; }                                      \
            if (index == ORC_UNSET) { /* push(i, j, T::zero()) */                                   \
                if (j >= n_cols) n_cols = j + 1;                                                    \
                if (len_off == 0) {                                                                 \
                    for (size_t q = 0; q < i + 2; ++q) offset_rows[q] = 0;                          \
                    len_off = i + 2;                                                                \
                } else if (i >= n_rows) {                                                           \
                    const uint32_t last = offset_rows[len_off - 1];                                 \
                    for (size_t q = len_off; q < i + 2; ++q) offset_rows[q] = last;                 \
                    len_off = i + 2; /* shrinks when i + 2 < len: Vec::resize truncates */          \
                    n_rows = i + 1;                                                                 \
                }                                                                                   \
                if (offset_rows[i] == ORC_UNSET) return ORC_ERR_CAPACITY;                           \
                index = offset_rows[i];                                                             \
                memmove(columns + index + 1, columns + index, (stored - index) * sizeof(uint32_t)); \
                memmove(values + index + 1, values + index, (stored - index) * sizeof(T));          \
                columns[index] = (uint32_t)j;                                                       \
                values[index] = (T)0;                                                               \
                ++stored;                                                                           \
                for (size_t q = i + 1; q < len_off; ++q) offset_rows[q] += 1;                       \
            }                                                                                       \
            if (ops && ops[k]) values[index] = vals[k]; else values[index] = values[index] + vals[k]; \
        }                                                                                           \
        *n_rows_out = n_rows; *n_cols_out = n_cols; *stored_out = stored;                           \
        *nnz_out = n_rows ? offset_rows[n_rows] : 0;                                                \
        return ORC_OK;                                                                              \
    }

DEF_CRS_REPLAY(f32, float)
DEF_CRS_REPLAY(f64, double)

/* SparseMatrix::prod (sparsematrix.rs:186-210) for Self = SparseMatCRS, rhs = SparseMatCRS with its column tables
 * (assemble_column_info sparsemat_crs.rs:180-191: per column the entries in storage order), restated literally up
 * to the container: for every row i (its entries stably sorted by column, :194-195) and every column j of rhs in
 * ascending order, sum = 0; for every (row, val_rhs) of rhs' column j in list order, for every (col, val) of the
 * sorted row while col <= row: if col == row { sum += val * val_rhs } (multiply and add rounded separately); the
 * sums that compare != 0 become the call stream ret.set(i, j, sum) -- written to ops_* here (capacity cap), to be
 * replayed on a SparseMatCRS by orc_crs_replay.  Dimension rule of :188-190 -> ORC_ERR_SIZE_MISMATCH.
 * O(n_rows * n_cols): small cases only. */
#define DEF_PROD(SUF, T)                                                                            \
    int orc_crs_prod_ops_##SUF(size_t a_rows, size_t a_cols, const uint32_t *a_off,                 \
                               const uint32_t *a_col, const T *a_val, size_t b_rows, size_t b_cols, \
                               const uint32_t *b_off, const uint32_t *b_col, const T *b_val,        \
                               size_t cap, size_t *n_out, uint32_t *ops_rows, uint32_t *ops_cols,   \
                               T *ops_vals) {                                                       \
        if (a_rows != b_cols || a_cols != b_rows) return ORC_ERR_SIZE_MISMATCH;                     \
        const size_t b_nnz = b_rows ? b_off[b_rows] : 0;                                            \
        /* column tables of rhs: counting sort by column keeps the storage order inside a column */ \
        uint32_t *ptr = (uint32_t *)calloc(b_cols + 2, sizeof(uint32_t));                           \
        uint32_t *ent = (uint32_t *)malloc((b_nnz + 1) * sizeof(uint32_t));                         \
        uint32_t *row_of = (uint32_t *)malloc((b_nnz + 1) * sizeof(uint32_t));                      \
        for (size_t r = 0; r < b_rows; ++r)                                                         \
            for (size_t q = b_off[r]; q < b_off[r + 1]; ++q) { row_of[q] = (uint32_t)r; ptr[b_col[q] + 2]++; } \
        for (size_t j = 0; j < b_cols; ++j) ptr[j + 2] += ptr[j + 1];                               \
        for (size_t q = 0; q < b_nnz; ++q) ent[ptr[b_col[q] + 1]++] = (uint32_t)q;                  \
        size_t n = 0, max_len = 0;                                                                  \
        for (size_t i = 0; i < a_rows; ++i)                                                         \
            if ((size_t)(a_off[i + 1] - a_off[i]) > max_len) max_len = a_off[i + 1] - a_off[i];     \
        uint32_t *sc = (uint32_t *)malloc((max_len + 1) * sizeof(uint32_t));                        \
        T *sv = (T *)malloc((max_len + 1) * sizeof(T));                                             \
        int rc = ORC_OK;                                                                            \
        for (size_t i = 0; i < a_rows && rc == ORC_OK; ++i) {                                       \
            const size_t len = a_off[i + 1] - a_off[i];                                             \
            for (size_t q = 0; q < len; ++q) { /* stable insertion sort by column */                \
                const uint32_t c = a_col[a_off[i] + q];                                             \
                const T v = a_val[a_off[i] + q];                                                    \
                size_t p = q;                                                                       \
                while (p > 0 && sc[p - 1] > c) { sc[p] = sc[p - 1]; sv[p] = sv[p - 1]; --p; }       \
                sc[p] = c; sv[p] = v;                                                               \
            }                                                                                       \
            for (size_t j = 0; j < b_cols; ++j) {                                                   \
                T sum = (T)0;                                                                       \
                for (size_t t = ptr[j]; t < ptr[j + 1]; ++t) {                                      \
                    const uint32_t row = row_of[ent[t]];                                            \
                    const T val_rhs = b_val[ent[t]];                                                \
                    for (size_t q = 0; q < len && sc[q] <= row; ++q)                                \
                        if (sc[q] == row) { const T prod = sv[q] * val_rhs; sum = sum + prod; }     \
                }                                                                                   \
                if (sum != (T)0) {                                                                  \
                    if (n >= cap) { rc = ORC_ERR_CAPACITY; break; }                                 \
                    ops_rows[n] = (uint32_t)i; ops_cols[n] = (uint32_t)j; ops_vals[n] = sum; ++n;   \
                }                                                                                   \
            }                                                                                       \
        }                                                                                           \
        *n_out = n;                                                                                 \
        free(ptr); free(ent); free(row_of); free(sc); free(sv);                                     \
        return rc;                                                                                  \
    }

DEF_PROD(f32, float)
DEF_PROD(f64, double)

/* Jacobi-preconditioned CG -- an EXTENSION: the reference has no preconditioner, so there is
 * nothing of its own to pin this against; it is ConjugateGradient::solve (linearsolver.rs:27-61) with the same guards,
 * the same stop rule (sqrt(f64(r.r)) < tol, tested after the update of r and before beta) and the same arithmetic
 * conventions (sequential left folds, one rounding per operation), plus z = r / diag(A) with diag_i = get(i, i) (first
 * match in storage order, sparsemat_crs.rs:54-67,136-142):
 *   r = b - A x; p = z; rz = r.z;  loop: alpha = rz / (p.Ap); x += p*alpha; r -= Ap*alpha; rr = r.r; stop test;
 *   z = r / d; rz' = r.z; beta = rz' / rz; p = p*beta + z.
 * A zero or absent diagonal entry is ORC_ERR_ZERO_DIAGONAL. */
#define DEF_PCG(SUF, T)                                                                     \
    int orc_pcg_jacobi_##SUF(size_t n_rows, size_t n_cols, const uint32_t *offset_rows,      \
                             const uint32_t *columns, const T *values, const T *b,           \
                             size_t b_len, T *x, size_t x_len, double tol, size_t iter_max,  \
                             size_t *iters_out, double *rr_out) {                            \
        if (n_rows != n_cols) return ORC_ERR_NOT_SQUARE;                                     \
        if (n_rows != b_len || n_rows != x_len) return ORC_ERR_SIZE_MISMATCH;                \
        size_t n = n_rows;                                                                   \
        T *r = (T *)malloc((n ? n : 1) * sizeof(T));                                         \
        T *p = (T *)malloc((n ? n : 1) * sizeof(T));                                         \
        T *ap = (T *)malloc((n ? n : 1) * sizeof(T));                                        \
        T *d = (T *)malloc((n ? n : 1) * sizeof(T));                                         \
        int rc = ORC_OK;                                                                     \
        for (size_t i = 0; i < n && rc == ORC_OK; ++i) {                                     \
            d[i] = (T)0;                                                                     \
            for (size_t q = offset_rows[i]; q < offset_rows[i + 1]; ++q)                     \
                if (columns[q] == i) { d[i] = values[q]; break; }                            \
            if (d[i] == (T)0) rc = ORC_ERR_ZERO_DIAGONAL;                                    \
        }                                                                                    \
        if (rc == ORC_OK) rc = orc_spmv_##SUF(n, offset_rows, columns, values, x, x_len, ap); \
        size_t iters = 0;                                                                    \
        T rr = (T)0;                                                                         \
        if (rc == ORC_OK) {                                                                  \
            T rz = (T)0;                                                                     \
            for (size_t i = 0; i < n; ++i) r[i] = b[i] - ap[i];                              \
            rr = orc_norm_squared_##SUF(r, n);                                               \
            for (size_t i = 0; i < n; ++i) { p[i] = r[i] / d[i]; T t = r[i] * p[i]; rz = rz + t; } \
            for (size_t k = 0; k < iter_max; ++k) {                                          \
                ++iters;                                                                     \
                rc = orc_spmv_##SUF(n, offset_rows, columns, values, p, n, ap);              \
                if (rc != ORC_OK) break;                                                     \
                T alpha = rz / orc_dot_##SUF(p, ap, n);                                      \
                orc_vec_axpy_##SUF(x, alpha, p, n);                                          \
                for (size_t i = 0; i < n; ++i) {                                             \
                    T t = ap[i] * alpha;                                                     \
                    r[i] = r[i] - t;                                                         \
                }                                                                            \
                rr = orc_norm_squared_##SUF(r, n);                                           \
                if (sqrt((double)rr) < tol) break;                                           \
                T rz_new = (T)0;                                                             \
                for (size_t i = 0; i < n; ++i) { T z = r[i] / d[i]; T t = r[i] * z; rz_new = rz_new + t; } \
                T beta = rz_new / rz;                                                        \
                rz = rz_new;                                                                 \
                for (size_t i = 0; i < n; ++i) { T z = r[i] / d[i]; T t = p[i] * beta; p[i] = t + z; } \
            }                                                                                \
        }                                                                                    \
        if (iters_out) *iters_out = iters;                                                   \
        if (rr_out) *rr_out = (double)rr;                                                    \
        free(r); free(p); free(ap); free(d);                                                 \
        return rc;                                                                           \
    }

DEF_PCG(f32, float)
DEF_PCG(f64, double)
