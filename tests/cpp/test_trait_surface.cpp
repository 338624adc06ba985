// Replays the reference's own SpMV / CG unit tests (lostinc0de/sparsemat src/lib.rs:36-52, 114-154,
// 54-82) through the C++ mirror of its interface (include/sparsemat.hpp) on the GPU.
// Built and run by tests/test_cpp_surface_gpu.py.
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "sparsemat.hpp"

using namespace sparsemat;

static int failures = 0;
#define CHECK(cond)                                                         \
    do {                                                                    \
        if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

int main() {
    // check_sparsemat_crs (src/lib.rs:114-154): CRS as add_to leaves it (row 3 stored as (3,3),(3,2))
    {
        auto a = SparseMatCRS<float>::from_raw_parts(4, 4, {0, 1, 2, 3, 5}, {1, 2, 2, 3, 2}, {4.2f, 4.12f, 2.12f, 5.12f, 1.12f});
        auto v = DenseVec<float>::from_vec({2.0f, 4.8f, 1.2f, 3.4f});
        auto y = a * v;                      // sp_crs.clone() * v
        CHECK(y.dim() == 4);
        CHECK(y.get(0) == 20.16f);           // assert_eq!(mvp.get(0), 20.16)
        CHECK(a.n_non_zero_entries() == 5 && a.n_rows() == 4 && a.n_cols() == 4);
        auto ys = a.mvp(v, SMH_SPMV_SEQ).to_vec();
        CHECK(ys[0] == 20.16f && ys[3] == 3.4f * 5.12f + 1.2f * 1.12f);
    }
    // check_sparsemat_indexlist -> to_crs (src/lib.rs:54-82): row 0 stored as (0,1),(0,2),(0,0)
    {
        auto a = SparseMatCRS<float>::from_raw_parts(3, 3, {0, 3, 5, 6}, {1, 2, 0, 2, 1, 2},
                                                     {4.2f, 0.12f, 7.12f, 4.12f, 2.24f, 2.12f});
        auto y = a.mvp(std::vector<float>{2.0f, 4.8f, 1.2f}, SMH_SPMV_SEQ);
        CHECK(y[0] == 34.544f);              // assert_eq!(mvp.get(0), 34.544)
        auto y2 = a.mvp(std::vector<float>{2.0f, 4.8f, 1.2f});
        CHECK(std::fabs(y2[0] - 34.544f) <= 1e-5f * 34.544f);
    }
    // check_sparsemat_indexlist (src/lib.rs:54-98), the call sequence itself: assembled on the device
    {
        SparseMatIndexList<float> sp;
        sp.add_to(0, 1, 4.2f);
        sp.add_to(1, 2, 4.12f);
        sp.add_to(2, 2, 2.12f);
        sp.add_to(1, 1, 1.12f);
        sp.add_to(1, 1, 1.12f);
        sp.add_to(0, 2, 0.12f);
        sp.set(0, 0, 8.12f);
        sp.set(0, 0, 7.12f);
        CHECK(sp.get(0, 0) == 7.12f);        // assert_eq!(sp.get(0, 0), 7.12)
        CHECK(sp.n_rows() == 3 && sp.n_cols() == 3);
        auto a = sp.to_crs();
        std::vector<uint32_t> off, col;
        std::vector<float> val;
        a.raw_parts(off, col, val);
        CHECK((off == std::vector<uint32_t>{0, 3, 5, 6}));
        CHECK((col == std::vector<uint32_t>{1, 2, 0, 2, 1, 2}));  // row 0 iterates (0,1),(0,2),(0,0): lib.rs:67-71
        CHECK((val == std::vector<float>{4.2f, 0.12f, 7.12f, 4.12f, 2.24f, 2.12f}));
        auto y = a.mvp(std::vector<float>{2.0f, 4.8f, 1.2f}, SMH_SPMV_STREAM);
        CHECK(y[0] == 34.544f);              // assert_eq!(mvp.get(0), 34.544)
        a.sort_rows();                       // sp.sort_row(i) for every row (lib.rs:94-98: "0 2.24 4.12 ")
        a.raw_parts(off, col, val);
        CHECK((col == std::vector<uint32_t>{0, 1, 2, 1, 2, 2}));
        CHECK((val == std::vector<float>{7.12f, 4.2f, 0.12f, 2.24f, 4.12f, 2.12f}));
    }
    // check_sparsemat_crs (src/lib.rs:114-154): add_to on a SparseMatCRS itself -- push prepends to the row
    {
        SparseMatIndexList<float> calls;     // (records the calls; as_direct_crs() applies them to a SparseMatCRS)
        calls.add_to(0, 1, 4.2f);
        calls.add_to(2, 2, 2.12f);
        calls.add_to(1, 2, 4.12f);
        calls.add_to(3, 2, 1.12f);
        calls.add_to(3, 3, 5.12f);
        auto sp_crs = calls.as_direct_crs();
        std::vector<uint32_t> off, col;
        std::vector<float> val;
        sp_crs.raw_parts(off, col, val);
        CHECK((off == std::vector<uint32_t>{0, 1, 2, 3, 5}));
        CHECK((col == std::vector<uint32_t>{1, 2, 2, 3, 2}));     // lib.rs:122-128: (3,3) before (3,2)
        CHECK((val == std::vector<float>{4.2f, 4.12f, 2.12f, 5.12f, 1.12f}));
        std::vector<uint32_t> rows, col_ptr, entries;              // lib.rs:137-142: iter_col(2) = (1,4.12),(2,2.12),(3,1.12)
        sp_crs.column_info(rows, col_ptr, entries);
        std::vector<std::pair<uint32_t, float>> col2;
        for (uint32_t e = col_ptr[2]; e < col_ptr[3]; ++e) col2.push_back({rows[entries[e]], val[entries[e]]});
        CHECK((col2 == std::vector<std::pair<uint32_t, float>>{{1, 4.12f}, {2, 2.12f}, {3, 1.12f}}));
        auto y = sp_crs.mvp(std::vector<float>{2.0f, 4.8f, 1.2f, 3.4f}, SMH_SPMV_STREAM);
        CHECK(y[0] == 20.16f);               // assert_eq!(mvp.get(0), 20.16)
        CHECK(sp_crs.density() == 5.0 / 16.0);   // assert_eq!(sp_crs.density(), 5.0 / 16.0), lib.rs:153
        CHECK(sp_crs.sparsity() == 1.0 - 5.0 / 16.0);
    }
    // SparseMatrix::transpose (sparsematrix.rs:174-184) + ColumnIter tables (sparsemat_crs.rs:180-204)
    {
        auto a = SparseMatCRS<float>::from_raw_parts(3, 4, {0, 2, 2, 5}, {1, 3, 0, 3, 1}, {1.0f, 2.0f, 3.0f, 4.0f, 5.0f});
        auto t = a.transpose();
        std::vector<uint32_t> off, col;
        std::vector<float> val;
        t.raw_parts(off, col, val);
        CHECK(t.n_rows() == 4 && t.n_cols() == 3);
        CHECK((off == std::vector<uint32_t>{0, 1, 3, 3, 5}));
        CHECK((col == std::vector<uint32_t>{2, 2, 0, 2, 0}));     // a SparseMatCRS prepends: latest source row first
        CHECK((val == std::vector<float>{3.0f, 5.0f, 1.0f, 4.0f, 2.0f}));
        std::vector<uint32_t> rows, col_ptr, entries;
        a.column_info(rows, col_ptr, entries);
        CHECK((rows == std::vector<uint32_t>{0, 0, 2, 2, 2}));
        CHECK((col_ptr == std::vector<uint32_t>{0, 1, 3, 3, 5}));
        CHECK((entries == std::vector<uint32_t>{2, 0, 4, 1, 3}));
        CHECK(!a.is_sorted() && !a.is_symmetric());
        // prod (sparsematrix.rs:186-210): needs a.n_rows == b.n_cols and a.n_cols == b.n_rows
        auto b = SparseMatCRS<float>::from_raw_parts(4, 3, {0, 1, 2, 2, 4}, {0, 2, 1, 2}, {2.0f, 0.5f, 4.0f, -1.0f});
        auto c = a.prod(b);
        c.raw_parts(off, col, val);
        // row 0 of a: (1, 1.0), (3, 2.0) -> b row 1: (2, .5), b row 3: (1, 4), (2, -1): c(0,1) = 8, c(0,2) = .5 + (-2)
        // row 2 of a: (0, 3.0), (3, 4.0), (1, 5.0) -> c(2,0) = 6, c(2,1) = 16, c(2,2) = 2.5 + (-4); columns descending
        CHECK(c.n_rows() == 3 && c.n_cols() == 3);
        CHECK((off == std::vector<uint32_t>{0, 2, 2, 5}));
        CHECK((col == std::vector<uint32_t>{2, 1, 2, 1, 0}));
        CHECK((val == std::vector<float>{-1.5f, 8.0f, -1.5f, 16.0f, 6.0f}));
        bool threw = false;
        try { a.prod(a); } catch (const Panic &p) { threw = std::string(p.what()).find("Dimension mismatch") != std::string::npos; }
        CHECK(threw);
    }
    // check_sparsemat_par (src/lib.rs:180-202): with_sub_matrices(4, 16) holding the matrix of the other container tests
    {
        std::vector<uint32_t> off = {0, 3, 5, 6};
        off.resize(17, 6u);  // rows 3..15 empty
        auto par = SparseMatPar<float>::with_sub_matrices(4, 16, 3, off, {1, 2, 0, 2, 1, 2},
                                                          {4.2f, 0.12f, 7.12f, 4.12f, 2.24f, 2.12f}, {0, 0, 0, 0});
        CHECK(par.n_blocks() == 4 && par.n_non_zero_entries() == 6 && par.n_cols() == 3);
        auto y = par.mvp(std::vector<float>{2.0f, 4.8f, 1.2f}, SMH_SPMV_STREAM);
        CHECK(y[0] == 34.544f);              // assert_eq!(mvp.get(0), 34.544)
        CHECK((par.get_block_and_row_id(7) == std::pair<size_t, size_t>{1, 3}));
        // the intended mvp_par (sparsemat_par.rs:37-68), device resident: results at b * R, one exchange, y = the next x
        {
            std::vector<uint32_t> o2 = {0, 2, 4, 6, 8}, c2 = {0, 1, 1, 2, 2, 3, 3, 0};
            std::vector<float> v2 = {2.0f, 1.0f, 3.0f, 1.0f, 4.0f, 1.0f, 5.0f, 1.0f};
            auto sq = SparseMatPar<float>::with_sub_matrices(2, 4, 4, o2, c2, v2, {0, 0});
            SparseMatPar<float>::ParVec xv(sq, std::vector<float>{1.0f, 2.0f, 3.0f, 4.0f}), yv(sq, 4), zv(sq, 4);
            sq.mvp_par(xv, yv, SMH_SPMV_STREAM);
            sq.mvp_par(yv, zv, SMH_SPMV_STREAM);
            sq.synchronize();
            CHECK((yv.to_vec() == std::vector<float>{4.0f, 9.0f, 16.0f, 21.0f}));     // A x
            CHECK((zv.to_vec() == std::vector<float>{17.0f, 43.0f, 85.0f, 109.0f}));   // A (A x): the exchanged y was the next x
        }
        // ConjugateGradient::solve on a partitioned SPD matrix (two blocks on device 0)
        auto spd = SparseMatPar<double>::with_sub_matrices(2, 2, 2, {0, 2, 4}, {0, 1, 0, 1}, {4.0, 1.0, 1.0, 3.0}, {0, 0});
        std::vector<double> b = {1.0, 2.0}, x = {2.0, 1.0};
        ConjugateGradient cg;
        cg.solve(spd, b, x);
        CHECK(std::floor(x[0] * 10000.0) / 10000.0 == 0.0909);   // src/lib.rs:49-51 through the partitioned path
        CHECK(cg.iterations() == 2);
    }
    // check_cg (src/lib.rs:36-52)
    {
        auto a = SparseMatCRS<double>::from_raw_parts(2, 2, {0, 2, 4}, {0, 1, 0, 1}, {4.0, 1.0, 1.0, 3.0});
        auto b = DenseVec<double>::from_vec({1.0, 2.0});
        auto x = DenseVec<double>::from_vec({2.0, 1.0});
        ConjugateGradient cg;                // ::default()
        cg.solve(a, b, x);
        CHECK(std::floor(x.get(0) * 10000.0) / 10000.0 == 0.0909);
        CHECK(cg.iterations() == 2);
        // the Jacobi-preconditioned extension reaches the same solution
        std::vector<double> bj = {1.0, 2.0}, xj = {2.0, 1.0};
        ConjugateGradient pcg;
        pcg.solve_jacobi(a, bj, xj);
        CHECK(std::fabs(xj[0] - 1.0 / 11.0) < 1e-12 && std::fabs(xj[1] - 7.0 / 11.0) < 1e-12);
    }
    // DenseVec operators (densevec.rs:76-140)
    {
        auto p = DenseVec<float>::from_vec({1.0f, 2.0f, 3.0f});
        auto q = DenseVec<float>::from_vec({0.5f, 0.25f, 4.0f});
        CHECK((p + q).to_vec() == (std::vector<float>{1.5f, 2.25f, 7.0f}));
        CHECK((p - q).to_vec() == (std::vector<float>{0.5f, 1.75f, -1.0f}));
        CHECK((p * 2.0f).to_vec() == (std::vector<float>{2.0f, 4.0f, 6.0f}));
        CHECK(p * q == 13.0f && p.norm_squared() == 14.0f && p.norm() == std::sqrt(14.0));
        CHECK(p.to_vec() == (std::vector<float>{1.0f, 2.0f, 3.0f}));  // operands untouched
    }
    // panics of the reference
    {
        auto a = SparseMatCRS<double>::from_raw_parts(2, 3, {0, 1, 2}, {0, 2}, {1.0, 2.0});
        auto b = DenseVec<double>::from_vec({1.0, 2.0});
        auto x = DenseVec<double>::from_vec({0.0, 0.0});
        try { ConjugateGradient().solve(a, b, x); CHECK(false); }
        catch (const Panic &e) { CHECK(std::string(e.what()) == "Matrix is not symmetric"); }   // linearsolver.rs:31
        auto sq = SparseMatCRS<double>::from_raw_parts(2, 2, {0, 1, 2}, {0, 1}, {1.0, 2.0});
        auto b3 = DenseVec<double>::from_vec({1.0, 2.0, 3.0});
        try { ConjugateGradient().solve(sq, b3, x); CHECK(false); }
        catch (const Panic &e) { CHECK(std::string(e.what()) == "Matrix and vector size mismatch"); }  // :35
        try { x.add(b3); CHECK(false); }
        catch (const Panic &e) { CHECK(std::string(e.what()) == "Dimension mismatch"); }          // densevec.rs:53
        try { a.mvp(std::vector<double>{1.0, 2.0}); CHECK(false); }                              // densevec.rs:41
        catch (const Panic &e) { CHECK(e.status == SMH_ERR_INDEX_RANGE); }
    }
    std::printf(failures ? "FAILED (%d)\n" : "ok (%d failures)\n", failures);
    return failures ? 1 : 0;
}
