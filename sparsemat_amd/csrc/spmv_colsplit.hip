// spmv_colsplit.hip -- K2s: a skewed matrix without column locality as TWO column-blocked matrices (gfx950).
//
// BASELINE C3 (f64, 10M rows, power-law row lengths 1..2048, uniform columns) is where column blocking pays its sweeps
// worst: K2c re-reads and rewrites all of y and one offset per row for EVERY column block, for rows of which most hold
// three entries -- 14.2 GB of traffic for 4.0 GB of CSR (profiles/r02_pmc_k2c_powerlaw.json) -- while K2f (one sweep) is
// thrown off by the few rows of up to 2048 entries (spmv_colfused.hip).  The two problems sit in DIFFERENT rows
// (profiles/r02_c3_split_probe.log): the 742 k rows of 64 entries and more hold 83 % of the entries, the 9.3 M short
// rows the rest.  So the handle keeps two sub-matrices (built once, on the device):
//   * LONG  : the rows of >= 64 entries, compacted (row i of it = row long_rows[i] of the matrix): few rows, so K2c's
//             sweeps of y and the offsets cost next to nothing and it runs at the L2-hit gather rate (2^18-column blocks);
//   * SHORT : all rows, the long ones emptied: rows of similar length again -> K2f, one sweep over y;
// and y = SHORT x (every row; zero for the emptied ones), then y[long_rows[i]] = (LONG x)[i].  Measured as two handles:
// 1.93 + 0.82 = 2.75 ms against K2c's 3.25 ms.  A row's sum is formed as in K2c / K2f (block by block): tolerance parity,
// deterministic.  The split is integer work, checked against a numpy restatement in the tests.
#include "internal.hpp"

namespace smh {

int device_exclusive_scan_u32(uint32_t *data, uint64_t n, hipStream_t s, uint64_t *total_out);  // spmv_colblock.hip

// flag[r] = row r is long; lshort[r] = its length if it is not, else 0   (r < n_rows; entry n_rows = 0: scans give totals)
__global__ void __launch_bounds__(kBlock)
k_split_lengths(const uint32_t *__restrict__ off, uint64_t n_rows, uint32_t min_long, uint32_t *__restrict__ flag, uint32_t *__restrict__ lshort) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_rows; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t len = r < n_rows ? off[r + 1] - off[r] : 0u;
        const bool is_long = r < n_rows && len >= min_long;
        flag[r] = is_long ? 1u : 0u;
        lshort[r] = is_long ? 0u : len;
    }
}

// pos = exclusive scan of flag: long row r is row pos[r] of LONG
__global__ void __launch_bounds__(kBlock)
k_split_long_rows(const uint32_t *__restrict__ off, const uint32_t *__restrict__ pos, uint64_t n_rows, uint32_t *__restrict__ long_rows,
                  uint32_t *__restrict__ llen /* n_long + 1, last = 0 */) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t i = pos[r];
        if (pos[r + 1] != i) {
            long_rows[i] = (uint32_t)r;
            llen[i] = off[r + 1] - off[r];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) llen[pos[n_rows]] = 0u;
}

// SHORT: one thread copies one (short) row; a long row has no entries there
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_split_copy_short(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const T *__restrict__ val, const uint32_t *__restrict__ off_s,
                   uint64_t n_rows, uint32_t *__restrict__ col_s, T *__restrict__ val_s) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t d0 = off_s[r], n = off_s[r + 1] - d0;
        const uint64_t s0 = off[r];
        for (uint32_t k = 0; k < n; ++k) { col_s[d0 + k] = col[s0 + k]; val_s[d0 + k] = val[s0 + k]; }
    }
}

// LONG: one wave copies one long row
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_split_copy_long(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const T *__restrict__ val, const uint32_t *__restrict__ long_rows,
                  const uint32_t *__restrict__ off_l, uint64_t n_long, uint32_t *__restrict__ col_l, T *__restrict__ val_l) {
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave, n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    for (uint64_t i = wave; i < n_long; i += n_waves) {
        const uint64_t s0 = off[long_rows[i]];
        const uint32_t d0 = off_l[i], n = off_l[i + 1] - d0;
        for (uint32_t k = lane; k < n; k += kWave) { col_l[d0 + k] = col[s0 + k]; val_l[d0 + k] = val[s0 + k]; }
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_split_scatter(const uint32_t *__restrict__ long_rows, const T *__restrict__ y_long, uint64_t n_long, T *__restrict__ y) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_long; i += (uint64_t)gridDim.x * blockDim.x) y[long_rows[i]] = y_long[i];
}

static unsigned sp_grid(uint64_t n) {
    uint64_t b = (n + kBlock - 1) / kBlock;
    if (b > 8192) b = 8192;
    return (unsigned)(b ? b : 1);
}

// Outputs (device, owned by the caller): long_rows [n_long]; LONG = (off_l [n_long + 1], col_l, val_l [nnz_long + 4]);
// SHORT = (off_s [n_rows + 1], col_s, val_s [nnz_short + 4]).  Entry arrays are padded with 4 zero entries (16-byte chunks).
int build_colsplit(int dtype, const uint32_t *off, const uint32_t *col, const void *val, size_t n_rows, size_t nnz, uint32_t min_long,
                   size_t *n_long_out, size_t *nnz_long_out, uint32_t **long_rows_out, uint32_t **off_l_out, uint32_t **col_l_out, void **val_l_out,
                   uint32_t **off_s_out, uint32_t **col_s_out, void **val_s_out, hipStream_t s) {
    const size_t vs = dtype_size(dtype);
    uint32_t *pos = nullptr, *off_s = nullptr, *long_rows = nullptr, *off_l = nullptr, *col_l = nullptr, *col_s = nullptr;
    void *val_l = nullptr, *val_s = nullptr;
    uint64_t n_long = 0, nnz_short = 0, nnz_long = 0;
    auto body = [&]() -> int {
        SMH_HIP(hipMalloc((void **)&pos, (n_rows + 1) * sizeof(uint32_t)));
        SMH_HIP(hipMalloc((void **)&off_s, (n_rows + 1) * sizeof(uint32_t)));
        hipLaunchKernelGGL(k_split_lengths, dim3(sp_grid(n_rows + 1)), dim3(kBlock), 0, s, off, (uint64_t)n_rows, min_long, pos, off_s);
        SMH_HIP(hipGetLastError());
        SMH_TRY(device_exclusive_scan_u32(pos, n_rows + 1, s, &n_long));
        SMH_TRY(device_exclusive_scan_u32(off_s, n_rows + 1, s, &nnz_short));
        nnz_long = nnz - nnz_short;
        SMH_HIP(hipMalloc((void **)&long_rows, (n_long ? n_long : 1) * sizeof(uint32_t)));
        SMH_HIP(hipMalloc((void **)&off_l, (n_long + 1) * sizeof(uint32_t)));
        hipLaunchKernelGGL(k_split_long_rows, dim3(sp_grid(n_rows)), dim3(kBlock), 0, s, off, pos, (uint64_t)n_rows, long_rows, off_l);
        SMH_HIP(hipGetLastError());
        uint64_t check_long = 0;
        SMH_TRY(device_exclusive_scan_u32(off_l, n_long + 1, s, &check_long));
        if (check_long != nnz_long) return fail(SMH_ERR_INVALID, "row-length split: entry counts disagree (%llu + %llu != %zu)",
                                                (unsigned long long)check_long, (unsigned long long)nnz_short, nnz);
        SMH_HIP(hipMalloc((void **)&col_l, (nnz_long + 4) * sizeof(uint32_t)));
        SMH_HIP(hipMalloc(&val_l, (nnz_long + 4) * vs));
        SMH_HIP(hipMalloc((void **)&col_s, (nnz_short + 4) * sizeof(uint32_t)));
        SMH_HIP(hipMalloc(&val_s, (nnz_short + 4) * vs));
        SMH_HIP(hipMemsetAsync(col_l + nnz_long, 0, 4 * sizeof(uint32_t), s));
        SMH_HIP(hipMemsetAsync((char *)val_l + nnz_long * vs, 0, 4 * vs, s));
        SMH_HIP(hipMemsetAsync(col_s + nnz_short, 0, 4 * sizeof(uint32_t), s));
        SMH_HIP(hipMemsetAsync((char *)val_s + nnz_short * vs, 0, 4 * vs, s));
        if (dtype == SMH_F64) {
            hipLaunchKernelGGL(k_split_copy_short<double>, dim3(sp_grid(n_rows)), dim3(kBlock), 0, s, off, col, (const double *)val, off_s,
                               (uint64_t)n_rows, col_s, (double *)val_s);
            hipLaunchKernelGGL(k_split_copy_long<double>, dim3(sp_grid(n_long * kWave)), dim3(kBlock), 0, s, off, col, (const double *)val, long_rows,
                               off_l, n_long, col_l, (double *)val_l);
        } else {
            hipLaunchKernelGGL(k_split_copy_short<float>, dim3(sp_grid(n_rows)), dim3(kBlock), 0, s, off, col, (const float *)val, off_s,
                               (uint64_t)n_rows, col_s, (float *)val_s);
            hipLaunchKernelGGL(k_split_copy_long<float>, dim3(sp_grid(n_long * kWave)), dim3(kBlock), 0, s, off, col, (const float *)val, long_rows,
                               off_l, n_long, col_l, (float *)val_l);
        }
        SMH_HIP(hipGetLastError());
        SMH_HIP(hipStreamSynchronize(s));
        return SMH_OK;
    };
    const int rc = body();
    (void)hipFree(pos);
    if (rc != SMH_OK) {
        (void)hipFree(off_s); (void)hipFree(long_rows); (void)hipFree(off_l); (void)hipFree(col_l); (void)hipFree(val_l); (void)hipFree(col_s); (void)hipFree(val_s);
        return rc;
    }
    *n_long_out = (size_t)n_long; *nnz_long_out = (size_t)nnz_long;
    *long_rows_out = long_rows; *off_l_out = off_l; *col_l_out = col_l; *val_l_out = val_l;
    *off_s_out = off_s; *col_s_out = col_s; *val_s_out = val_s;
    return SMH_OK;
}

int launch_split_scatter(int dtype, const uint32_t *long_rows, const void *y_long, size_t n_long, void *y, hipStream_t s) {
    if (n_long == 0) return SMH_OK;
    if (dtype == SMH_F64) hipLaunchKernelGGL(k_split_scatter<double>, dim3(sp_grid(n_long)), dim3(kBlock), 0, s, long_rows, (const double *)y_long, (uint64_t)n_long, (double *)y);
    else hipLaunchKernelGGL(k_split_scatter<float>, dim3(sp_grid(n_long)), dim3(kBlock), 0, s, long_rows, (const float *)y_long, (uint64_t)n_long, (float *)y);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

}  // namespace smh
