// sparsemat.hpp -- C++ host-side mirror of the reference crate's interface for the hot path, header-only
// over the C ABI of libsparsemat_hip.so (include/sparsemat_hip.h).
//
// The reference is a Rust crate; this image has no Rust toolchain, so the tested host side above the C
// ABI is this C++ mirror (same names, argument meaning and failure behaviour as the reference) plus the
// Python mirror used by the tests/bench (sparsemat_amd/).  The Rust shim a maintainer would add is in
// rust/ and INTEGRATION.md.
//
//   sparsemat::DenseVec<T>            <- densevec.rs:5-140, vector.rs:5-64
//   sparsemat::SparseMatCRS<T>        <- sparsemat_crs.rs (Index = u32), SparseMatrix::mvp sparsematrix.rs:146-158
//   sparsemat::ConjugateGradient      <- linearsolver.rs:6-61
//   sparsemat::Panic                  <- the reference's panic!() texts (linearsolver.rs:31,35; densevec.rs:53,62)
//
// T is float or double.  Everything computes on the device; there is no CPU fallback.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "sparsemat_hip.h"

namespace sparsemat {

struct Panic : std::runtime_error {
    int status;
    Panic(int st, const std::string &msg) : std::runtime_error(msg), status(st) {}
};

namespace detail {
inline void check(int status) {
    if (status != SMH_OK) {
        std::string msg = smh_last_error();
        if (msg.empty()) msg = smh_status_string(status);
        throw Panic(status, msg);
    }
}
template <typename T> struct dtype_of;
template <> struct dtype_of<float> { static constexpr smh_dtype value = SMH_F32; };
template <> struct dtype_of<double> { static constexpr smh_dtype value = SMH_F64; };
}  // namespace detail

template <typename T>
class DenseVec {
  public:
    using Value = T;
    DenseVec() { detail::check(smh_vec_create(detail::dtype_of<T>::value, 0, &h_)); }
    explicit DenseVec(size_t n) { detail::check(smh_vec_create(detail::dtype_of<T>::value, n, &h_)); }
    // Vector::from_vec (densevec.rs:30-34)
    static DenseVec from_vec(const std::vector<T> &v) {
        DenseVec r(nullptr);
        detail::check(smh_vec_from_host(detail::dtype_of<T>::value, v.size(), v.data(), &r.h_));
        return r;
    }
    DenseVec(const DenseVec &o) : DenseVec(o.dim()) { detail::check(smh_vec_copy(h_, o.h_)); }  // Clone
    DenseVec(DenseVec &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    DenseVec &operator=(DenseVec o) { std::swap(h_, o.h_); return *this; }
    ~DenseVec() { smh_vec_destroy(h_); }

    size_t dim() const { return smh_vec_dim(h_); }  // densevec.rs:36-38
    std::vector<T> to_vec() const {                 // iter().collect()
        std::vector<T> v(dim());
        detail::check(smh_vec_download(h_, v.data()));
        return v;
    }
    T get(size_t i) const {                         // densevec.rs:40-42 (bounds checked)
        if (i >= dim()) throw Panic(SMH_ERR_INDEX_RANGE, "index out of bounds");
        return to_vec()[i];
    }
    void add(const DenseVec &rhs) { detail::check(smh_vec_add(h_, rhs.h_)); }   // :51-58
    void sub(const DenseVec &rhs) { detail::check(smh_vec_sub(h_, rhs.h_)); }   // :60-67
    void scale(T a) { detail::check(smh_vec_scale(h_, (double)a)); }            // :69-73
    T inner_prod(const DenseVec &rhs) const {                                   // vector.rs:50-53
        double out = 0;
        detail::check(smh_vec_dot(h_, rhs.h_, &out));
        return (T)out;
    }
    T norm_squared() const {                                                    // vector.rs:56-58
        double out = 0;
        detail::check(smh_vec_norm_squared(h_, &out));
        return (T)out;
    }
    double norm() const { return std::sqrt((double)norm_squared()); }           // vector.rs:61-63
    // operator sugar (densevec.rs:76-140)
    DenseVec &operator+=(const DenseVec &r) { add(r); return *this; }
    DenseVec &operator-=(const DenseVec &r) { sub(r); return *this; }
    DenseVec &operator*=(T a) { scale(a); return *this; }
    friend DenseVec operator+(DenseVec l, const DenseVec &r) { l += r; return l; }
    friend DenseVec operator-(DenseVec l, const DenseVec &r) { l -= r; return l; }
    friend DenseVec operator*(DenseVec l, T a) { l *= a; return l; }
    friend T operator*(const DenseVec &l, const DenseVec &r) { return l.inner_prod(r); }

    smh_vec *handle() const { return h_; }

  private:
    explicit DenseVec(std::nullptr_t) : h_(nullptr) {}
    smh_vec *h_ = nullptr;
};

template <typename T>
class SparseMatCRS {
  public:
    using Value = T;
    using Index = uint32_t;
    // The reference has no public raw-array constructor (from_sparsemat_index is pub(crate),
    // sparsemat_crs.rs:24-50): this is the documented addition.  Arrays are borrowed for the call.
    static SparseMatCRS from_raw_parts(size_t n_rows, size_t n_cols, const std::vector<uint32_t> &offset_rows,
                                       const std::vector<uint32_t> &columns, const std::vector<T> &values) {
        if (n_rows && offset_rows.size() != n_rows + 1) throw Panic(SMH_ERR_INVALID, "offset_rows must have n_rows+1 entries");
        if (columns.size() != values.size()) throw Panic(SMH_ERR_INVALID, "columns and values differ in length");
        SparseMatCRS m;
        detail::check(smh_crs_create(detail::dtype_of<T>::value, n_rows, n_cols, values.size(), offset_rows.data(),
                                     columns.data(), values.data(), 1, &m.h_));
        return m;
    }
    SparseMatCRS(SparseMatCRS &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    SparseMatCRS &operator=(SparseMatCRS &&o) noexcept { std::swap(h_, o.h_); return *this; }
    SparseMatCRS(const SparseMatCRS &) = delete;
    ~SparseMatCRS() { smh_crs_destroy(h_); }

    size_t n_rows() const { return smh_crs_n_rows(h_); }              // sparsemat_crs.rs:124-126
    size_t n_cols() const { return smh_crs_n_cols(h_); }              // :128-130
    size_t n_non_zero_entries() const { return smh_crs_nnz(h_); }     // :132-134
    bool empty() const { return n_rows() == 0; }                      // sparsematrix.rs:119-121
    double density() const {                                           // sparsematrix.rs:237-241
        return (double)n_non_zero_entries() / (double)(n_rows() * n_cols());
    }
    double sparsity() const { return 1.0 - density(); }                // sparsematrix.rs:244-246
    void scale(T a) { detail::check(smh_crs_scale(h_, (double)a)); }  // sparsemat_crs.rs:153-157

    // SparseMatrix::mvp (sparsematrix.rs:146-158): returns a NEW vector with dim == n_rows
    DenseVec<T> mvp(const DenseVec<T> &rhs, int variant = SMH_SPMV_AUTO) const {
        DenseVec<T> ret(n_rows());
        detail::check(smh_crs_spmv_vec(h_, rhs.handle(), ret.handle(), variant));
        return ret;
    }
    std::vector<T> mvp(const std::vector<T> &rhs, int variant = SMH_SPMV_AUTO) const {
        std::vector<T> y(n_rows());
        detail::check(smh_crs_spmv(h_, rhs.data(), rhs.size(), y.data(), variant));
        return y;
    }
    // SparseMatrix::inner_prod (sparsematrix.rs:161-171): lhs^T A rhs
    T inner_prod(const DenseVec<T> &lhs, const DenseVec<T> &rhs) const {
        double out = 0.0;
        detail::check(smh_crs_inner_prod_vec(h_, lhs.handle(), rhs.handle(), SMH_SPMV_AUTO, &out));
        return (T)out;
    }
    // `A * v` (sparsemat_ops! Mul<DenseVec>, sparsematrix.rs:435-443)
    friend DenseVec<T> operator*(const SparseMatCRS &a, const DenseVec<T> &v) { return a.mvp(v); }

    // Sortable::sort_row (sparsemat_crs.rs:163-172) for every row: ascending columns, stable
    void sort_rows() { detail::check(smh_crs_sort_rows(h_)); }
    // SparseMatrix::transpose (sparsematrix.rs:174-184): `ret.set(j, i, val)` over the entries in storage order, into a
    // fresh SparseMatCRS -- same arrays as the reference's container ends up with (rows reversed, first-push quirk)
    SparseMatCRS transpose() const {
        SparseMatCRS t;
        detail::check(smh_crs_transpose(h_, &t.h_));
        return t;
    }
    // SparseMatrix::prod (sparsematrix.rs:186-210): self * rhs as the reference's loops build it (same order of
    // additions, zero sums dropped, rows in descending column order); Err("Dimension mismatch") becomes a Panic
    SparseMatCRS prod(const SparseMatCRS &rhs) const {
        SparseMatCRS c;
        detail::check(smh_crs_prod(h_, rhs.h_, &c.h_));
        return c;
    }
    bool is_symmetric() const {  // sparsematrix.rs:212-222
        int out = 0;
        detail::check(smh_crs_is_symmetric(h_, &out));
        return out != 0;
    }
    bool is_sorted() const {  // sparsematrix.rs:263-271
        int out = 0;
        detail::check(smh_crs_is_sorted(h_, &out));
        return out != 0;
    }
    // ColumnIter::assemble_column_info (sparsemat_crs.rs:180-191) as arrays: rows[k] = row of entry k; iter_col(j)
    // (:193-204) walks entries[col_ptr[j] .. col_ptr[j+1]) and yields (rows[e], values[e])
    void column_info(std::vector<uint32_t> &rows, std::vector<uint32_t> &col_ptr, std::vector<uint32_t> &entries) const {
        rows.assign(n_non_zero_entries() ? n_non_zero_entries() : 1, 0u);
        entries.assign(rows.size(), 0u);
        col_ptr.assign(n_cols() + 1, 0u);
        detail::check(smh_crs_column_info(h_, rows.data(), col_ptr.data(), entries.data()));
        rows.resize(n_non_zero_entries());
        entries.resize(n_non_zero_entries());
    }
    // the CRS arrays as stored (offset_rows, columns, values)
    void raw_parts(std::vector<uint32_t> &offset_rows, std::vector<uint32_t> &columns, std::vector<T> &values) const {
        offset_rows.assign(n_rows() + 1, 0u);
        columns.assign(n_non_zero_entries(), 0u);
        values.assign(n_non_zero_entries(), T(0));
        detail::check(smh_crs_download(h_, offset_rows.data(), columns.data(), values.data()));
    }

    smh_crs *handle() const { return h_; }

  private:
    template <typename U> friend class SparseMatIndexList;
    SparseMatCRS() = default;
    smh_crs *h_ = nullptr;
};

// Assembly with the call surface of the reference's SparseMatIndexList (sparsemat_indexlist.rs): `add_to` /
// `set` (sparsematrix.rs:231-233 / :226-228) are RECORDED (O(1) each, no list walk) and `to_crs()` (:61-63 ->
// sparsemat_crs.rs:24-50) builds the identical CRS on the device with smh_crs_assemble: rows in order of first
// appearance, duplicates folded in call order, bit for bit.
template <typename T>
class SparseMatIndexList {
  public:
    void add_to(size_t i, size_t j, T val) { record(i, j, val, 0); }
    void set(size_t i, size_t j, T val) { record(i, j, val, 1); }
    // SparseMatrix::get (sparsemat_indexlist.rs:148-155): value of (i, j), zero when absent.  Folds the
    // recorded calls (O(calls)): meant for spot checks, as in the reference's tests.
    T get(size_t i, size_t j) const {
        T acc = T(0);
        for (size_t k = 0; k < vals_.size(); ++k)
            if (rows_[k] == i && cols_[k] == j) acc = ops_[k] ? vals_[k] : T(acc + vals_[k]);
        return acc;
    }
    size_t n_rows() const { return n_rows_; }  // indexlist.rs:58-60
    size_t n_cols() const { return n_cols_; }  // sparsemat_indexlist.rs:46-48
    SparseMatCRS<T> to_crs() const {
        SparseMatCRS<T> m;
        detail::check(smh_crs_assemble(detail::dtype_of<T>::value, vals_.size(), rows_.data(), cols_.data(), vals_.data(),
                                       ops_.data(), &m.h_));
        return m;
    }
    // The same calls made on a SparseMatCRS directly (get_mut sparsemat_crs.rs:143-149 -> push :71-92 inserts at the
    // START of the row, first-push quirk included): the matrix src/lib.rs:114-154 builds.  smh_crs_replay.
    SparseMatCRS<T> as_direct_crs() const {
        SparseMatCRS<T> m;
        detail::check(smh_crs_replay(detail::dtype_of<T>::value, vals_.size(), rows_.data(), cols_.data(), vals_.data(),
                                     ops_.data(), &m.h_));
        return m;
    }

  private:
    void record(size_t i, size_t j, T val, uint8_t op) {
        if (i >= 0xFFFFFFFFull || j >= 0xFFFFFFFFull) throw Panic(SMH_ERR_CAPACITY, "index exceeds u32");
        rows_.push_back((uint32_t)i); cols_.push_back((uint32_t)j); vals_.push_back(val); ops_.push_back(op);
        if (i + 1 > n_rows_) n_rows_ = i + 1;
        if (j + 1 > n_cols_) n_cols_ = j + 1;
    }
    std::vector<uint32_t> rows_, cols_;
    std::vector<T> vals_;
    std::vector<uint8_t> ops_;
    size_t n_rows_ = 0, n_cols_ = 0;
};

// SparseMatPar<SparseMatCRS<T,u32>> (sparsemat_par.rs:12-35, 86-140) driven by one process: block b = rows
// [b R, (b+1) R), R = n_rows / n_blocks (the last block takes the remainder), on device devices[b] (empty: block b on
// device b mod device count).  Filled from a global CRS; host vectors in, host vectors out.
template <typename T>
class SparseMatPar {
  public:
    static SparseMatPar with_sub_matrices(size_t n_blocks, size_t n_rows, size_t n_cols, const std::vector<uint32_t> &offset_rows,
                                          const std::vector<uint32_t> &columns, const std::vector<T> &values,
                                          const std::vector<int> &devices = {}) {
        if (offset_rows.size() != n_rows + 1) throw Panic(SMH_ERR_INVALID, "offset_rows must have n_rows+1 entries");
        if (columns.size() != values.size()) throw Panic(SMH_ERR_INVALID, "columns and values differ in length");
        if (!devices.empty() && devices.size() != n_blocks) throw Panic(SMH_ERR_INVALID, "one device per block");
        SparseMatPar m;
        detail::check(smh_par_create(detail::dtype_of<T>::value, n_blocks, devices.empty() ? nullptr : devices.data(), n_rows, n_cols,
                                     offset_rows.data(), columns.data(), values.data(), 1, &m.h_));
        return m;
    }
    SparseMatPar(SparseMatPar &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    SparseMatPar &operator=(SparseMatPar &&o) noexcept { std::swap(h_, o.h_); return *this; }
    SparseMatPar(const SparseMatPar &) = delete;
    ~SparseMatPar() { smh_par_destroy(h_); }

    size_t n_rows() const { return smh_par_n_rows(h_); }
    size_t n_cols() const { return smh_par_n_cols(h_); }                       // sparsemat_par.rs:109-115
    size_t n_non_zero_entries() const { return smh_par_nnz(h_); }              // :117-123
    size_t n_blocks() const { return smh_par_n_blocks(h_); }
    void scale(T a) { detail::check(smh_par_scale(h_, (double)a)); }           // :135-139
    std::pair<size_t, size_t> get_block_and_row_id(size_t row) const {         // :31-35, clamped to the last block
        size_t b = 0, r = 0;
        detail::check(smh_par_get_block_and_row_id(h_, row, &b, &r));
        return {b, r};
    }
    std::vector<T> mvp(const std::vector<T> &rhs, int variant = SMH_SPMV_AUTO) const {
        std::vector<T> y(n_rows());
        detail::check(smh_par_spmv(h_, rhs.data(), rhs.size(), y.data(), variant));
        return y;
    }
    smh_par *handle() const { return h_; }

    // A DenseVec distributed over the blocks: one full-length device buffer per block, block b owns [b R, (b+1) R).
    class ParVec {
      public:
        ParVec(const SparseMatPar &m, size_t n) { detail::check(smh_par_vec_create(m.h_, n, &v_)); }
        ParVec(const SparseMatPar &m, const std::vector<T> &host) : ParVec(m, host.size()) {
            detail::check(smh_par_vec_upload(v_, host.data()));
        }
        ParVec(ParVec &&o) noexcept : v_(o.v_) { o.v_ = nullptr; }
        ParVec(const ParVec &) = delete;
        ~ParVec() { smh_par_vec_destroy(v_); }
        size_t dim() const { return smh_par_vec_dim(v_); }
        std::vector<T> to_vec() const {  // the owned slices of all blocks = the whole vector
            std::vector<T> out(dim());
            detail::check(smh_par_vec_download(v_, out.data()));
            return out;
        }
        smh_par_vec *handle() const { return v_; }

      private:
        smh_par_vec *v_ = nullptr;
    };
    // The reference's intended mvp_par (sparsemat_par.rs:37-68), device resident: every block multiplies against the
    // shared x, its results go to offset b * R of y, then ONE exchange inside the library (RCCL all-gather, or only the
    // entries each block references) makes y usable as the next x.  Asynchronous: synchronize() waits.
    void mvp_par(const ParVec &x, ParVec &y, int variant = SMH_SPMV_AUTO, int exchange = SMH_EXCHANGE_AUTO) const {
        detail::check(smh_par_spmv_dev(h_, x.handle(), y.handle(), variant, exchange));
    }
    void synchronize() const { detail::check(smh_par_synchronize(h_)); }

  private:
    SparseMatPar() = default;
    smh_par *h_ = nullptr;
};

// linearsolver.rs:12-61.  The reference keeps tol / iter_max private with only Default (1e-12, 10000);
// the two-argument constructor is the documented addition.
class ConjugateGradient {
  public:
    ConjugateGradient() = default;
    ConjugateGradient(double tol, size_t iter_max) : tol_(tol), iter_max_(iter_max) {}
    // LinearSolver::solve(&self, mat, b, x): x is updated in place; panics become sparsemat::Panic with
    // the reference's text ("Matrix is not symmetric", "Matrix and vector size mismatch").
    template <typename T>
    void solve(const SparseMatCRS<T> &mat, const DenseVec<T> &b, DenseVec<T> &x) {
        detail::check(smh_cg_solve_vec(mat.handle(), b.handle(), x.handle(), tol_, iter_max_, SMH_SPMV_AUTO, 0,
                                       &iterations_, &r_norm_squared_));
    }
    // Jacobi-preconditioned variant -- an extension, not in the reference: z = r / diag(A); host vectors
    template <typename T>
    void solve_jacobi(const SparseMatCRS<T> &mat, const std::vector<T> &b, std::vector<T> &x) {
        detail::check(smh_pcg_jacobi_solve(mat.handle(), b.data(), b.size(), x.data(), x.size(), tol_, iter_max_, SMH_SPMV_AUTO,
                                           &iterations_, &r_norm_squared_));
    }
    // the same solve on a row-partitioned matrix (M = SparseMatPar): host vectors, x updated in place
    template <typename T>
    void solve(const SparseMatPar<T> &mat, const std::vector<T> &b, std::vector<T> &x) {
        detail::check(smh_par_cg_solve(mat.handle(), b.data(), b.size(), x.data(), x.size(), tol_, iter_max_, SMH_SPMV_AUTO,
                                       &iterations_, &r_norm_squared_));
    }
    size_t iterations() const { return iterations_; }
    double r_norm_squared() const { return r_norm_squared_; }

  private:
    double tol_ = 1e-12;
    size_t iter_max_ = 10000;
    size_t iterations_ = 0;
    double r_norm_squared_ = 0.0;
};

}  // namespace sparsemat
