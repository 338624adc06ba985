// tools/vendor_rocsparse.cpp -- COMPARISON ROW ONLY (a vendor sparse library may appear in
// benchmark output, never in the implementation).  Times rocSPARSE's CSR SpMV algorithms (rocsparse_spmv: default,
// adaptive, rowsplit, LRB, nnzsplit; analysis done once, outside the timed region) on the BASELINE shapes next to
// this library's AUTO kernel, same device arrays, HIP events, median of 20 single launches.
//
//   hipcc -O2 --offload-arch=gfx950 -Iinclude tools/vendor_rocsparse.cpp -o /tmp/vendor_rocsparse \
//         -Lsparsemat_amd -lsparsemat_hip -lrocsparse -Wl,-rpath,$PWD/sparsemat_amd -Wno-deprecated-declarations
//   /tmp/vendor_rocsparse banded uniform lap512 powerlaw
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sparsemat_hip.h"

#define HIPCHK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); std::exit(2); } } while (0)
#define SMH(c) do { int r_ = (c); if (r_ != SMH_OK) { std::printf("smh error %d (%s) at %d\n", r_, smh_last_error(), __LINE__); std::exit(2); } } while (0)
#define RS(c) do { rocsparse_status s_ = (c); if (s_ != rocsparse_status_success) { std::printf("rocsparse status %d at %d\n", (int)s_, __LINE__); return -1.0; } } while (0)

template <typename F>
static double median_ms(F &&launch, int reps = 20, int warm = 3) {
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    for (int i = 0; i < warm; ++i) launch();
    HIPCHK(hipDeviceSynchronize());
    std::vector<float> t(reps);
    for (int i = 0; i < reps; ++i) {
        HIPCHK(hipEventRecord(a, nullptr));
        launch();
        HIPCHK(hipEventRecord(b, nullptr));
        HIPCHK(hipEventSynchronize(b));
        HIPCHK(hipEventElapsedTime(&t[i], a, b));
    }
    std::sort(t.begin(), t.end());
    HIPCHK(hipEventDestroy(a));
    HIPCHK(hipEventDestroy(b));
    return t[reps / 2];
}

static double time_rocsparse(rocsparse_handle h, rocsparse_spmv_alg alg, size_t n_rows, size_t n_cols, size_t nnz, uint32_t *off,
                             uint32_t *col, void *val, void *x, void *y, bool f64) {
    const rocsparse_datatype dt = f64 ? rocsparse_datatype_f64_r : rocsparse_datatype_f32_r;
    rocsparse_spmat_descr A;
    rocsparse_dnvec_descr X, Y;
    RS(rocsparse_create_csr_descr(&A, (int64_t)n_rows, (int64_t)n_cols, (int64_t)nnz, off, col, val, rocsparse_indextype_i32,
                                  rocsparse_indextype_i32, rocsparse_index_base_zero, dt));
    RS(rocsparse_create_dnvec_descr(&X, (int64_t)n_cols, x, dt));
    RS(rocsparse_create_dnvec_descr(&Y, (int64_t)n_rows, y, dt));
    const double a64 = 1.0, b64 = 0.0;
    const float a32 = 1.0f, b32 = 0.0f;
    const void *alpha = f64 ? (const void *)&a64 : (const void *)&a32, *beta = f64 ? (const void *)&b64 : (const void *)&b32;
    size_t bytes = 0;
    RS(rocsparse_spmv(h, rocsparse_operation_none, alpha, A, X, beta, Y, dt, alg, rocsparse_spmv_stage_buffer_size, &bytes, nullptr));
    void *buf = nullptr;
    HIPCHK(hipMalloc(&buf, bytes ? bytes : 16));
    RS(rocsparse_spmv(h, rocsparse_operation_none, alpha, A, X, beta, Y, dt, alg, rocsparse_spmv_stage_preprocess, &bytes, buf));
    HIPCHK(hipDeviceSynchronize());
    bool ok = true;
    const double ms = median_ms([&]() {
        if (rocsparse_spmv(h, rocsparse_operation_none, alpha, A, X, beta, Y, dt, alg, rocsparse_spmv_stage_compute, &bytes, buf) !=
            rocsparse_status_success)
            ok = false;
    });
    HIPCHK(hipFree(buf));
    rocsparse_destroy_spmat_descr(A);
    rocsparse_destroy_dnvec_descr(X);
    rocsparse_destroy_dnvec_descr(Y);
    return ok ? ms : -1.0;
}

int main(int argc, char **argv) {
    rocsparse_handle h;
    if (rocsparse_create_handle(&h) != rocsparse_status_success) { std::printf("no rocsparse handle\n"); return 2; }
    for (int ai = 1; ai < argc; ++ai) {
        const std::string c = argv[ai];
        const bool f64 = c == "powerlaw";
        const smh_dtype dt = f64 ? SMH_F64 : SMH_F32;
        const size_t vs = f64 ? 8 : 4;
        size_t n = 10000000, nnz = 0;
        uint32_t *off = nullptr, *col = nullptr;
        void *val = nullptr;
        if (c == "lap512") {
            n = (size_t)512 * 512 * 512;
            SMH(smh_synth_laplace3d(dt, 512, 512, 512, 0, n, nullptr, nullptr, nullptr, &nnz, nullptr));
            HIPCHK(hipMalloc((void **)&off, (n + 1) * 4)); HIPCHK(hipMalloc((void **)&col, (nnz + 4) * 4)); HIPCHK(hipMalloc(&val, (nnz + 4) * vs));
            SMH(smh_synth_laplace3d(dt, 512, 512, 512, 0, n, off, col, val, &nnz, nullptr));
        } else if (c == "powerlaw") {
            std::vector<uint32_t> cdf(2048), len(n), offs(n + 1, 0);
            SMH(smh_synth_powerlaw_cdf(2048, 1.52, cdf.data()));
            SMH(smh_synth_powerlaw_lengths(0x5EED0001ull, 0, n, 2048, cdf.data(), len.data()));
            for (size_t i = 0; i < n; ++i) offs[i + 1] = offs[i] + len[i];
            nnz = offs[n];
            HIPCHK(hipMalloc((void **)&off, (n + 1) * 4)); HIPCHK(hipMalloc((void **)&col, (nnz + 4) * 4)); HIPCHK(hipMalloc(&val, (nnz + 4) * vs));
            HIPCHK(hipMemcpy(off, offs.data(), (n + 1) * 4, hipMemcpyHostToDevice));
            SMH(smh_synth_fill(dt, 0x5EED0001ull, n, 0, n, off, col, val, nullptr));
        } else {
            const int pattern = c == "uniform" ? 1 : c == "diag" ? 2 : 0;
            nnz = n * 32;
            HIPCHK(hipMalloc((void **)&off, (n + 1) * 4)); HIPCHK(hipMalloc((void **)&col, (nnz + 4) * 4)); HIPCHK(hipMalloc(&val, (nnz + 4) * vs));
            SMH(smh_synth_fixed(dt, 0x5EED0001ull, pattern, n, 32, 0, n, off, col, val, nullptr));
        }
        void *x = nullptr, *y = nullptr, *y2 = nullptr;
        HIPCHK(hipMalloc(&x, n * vs)); HIPCHK(hipMalloc(&y, n * vs)); HIPCHK(hipMalloc(&y2, n * vs));
        SMH(smh_synth_x(dt, 0x5EED0002ull, 0, n, x, nullptr));
        HIPCHK(hipDeviceSynchronize());
        const double B = (double)nnz * (vs + 4) + (double)(n + 1) * 4 + 2.0 * (double)n * vs;
        std::printf("== %s: rows %zu nnz %zu %s, algorithmic bytes %.3f GB\n", c.c_str(), n, nnz, f64 ? "f64" : "f32", B / 1e9);
        smh_crs *m = nullptr;
        SMH(smh_crs_create_dev(dt, n, n, nnz, off, col, val, 0, &m));
        SMH(smh_crs_prepare(m, SMH_SPMV_AUTO));
        const double ours = median_ms([&]() { smh_crs_spmv_dev(m, x, n, y, SMH_SPMV_AUTO, nullptr); });
        std::printf("  %-34s %8.3f ms  %6.0f GB/s (%.1f%% of 8 TB/s)\n", "this library (AUTO)", ours, B / ours / 1e6, B / ours / 1e6 / 80);
        const struct { rocsparse_spmv_alg alg; const char *name; } algs[] = {
            {rocsparse_spmv_alg_default, "rocsparse_spmv default"}, {rocsparse_spmv_alg_csr_adaptive, "rocsparse_spmv csr_adaptive"},
            {rocsparse_spmv_alg_csr_rowsplit, "rocsparse_spmv csr_rowsplit"}, {rocsparse_spmv_alg_csr_lrb, "rocsparse_spmv csr_lrb"},
            {rocsparse_spmv_alg_csr_nnzsplit, "rocsparse_spmv csr_nnzsplit"}};
        for (const auto &a : algs) {
            const double ms = time_rocsparse(h, a.alg, n, n, nnz, off, col, val, x, y2, f64);
            if (ms < 0) { std::printf("  %-34s not available\n", a.name); continue; }
            std::printf("  %-34s %8.3f ms  %6.0f GB/s (%.1f%% of 8 TB/s)\n", a.name, ms, B / ms / 1e6, B / ms / 1e6 / 80);
        }
        // agreement of the two results (last vendor algorithm that ran)
        std::vector<char> h1(n * vs), h2(n * vs);
        HIPCHK(hipMemcpy(h1.data(), y, n * vs, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(h2.data(), y2, n * vs, hipMemcpyDeviceToHost));
        double worst = 0.0;
        for (size_t i = 0; i < n; ++i) {
            const double d = f64 ? ((double *)h1.data())[i] - ((double *)h2.data())[i] : (double)((float *)h1.data())[i] - ((float *)h2.data())[i];
            worst = std::max(worst, d < 0 ? -d : d);
        }
        std::printf("  max |y_vendor - y_ours| = %.3g\n", worst);
        smh_crs_destroy(m);
        HIPCHK(hipFree(off)); HIPCHK(hipFree(col)); HIPCHK(hipFree(val)); HIPCHK(hipFree(x)); HIPCHK(hipFree(y)); HIPCHK(hipFree(y2));
    }
    rocsparse_destroy_handle(h);
    return 0;
}
