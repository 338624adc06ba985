// synth.hip -- counter-based synthetic workload generators (bench / test support, not the hot path).
//
// The spec (DESIGN.md "Synthetic inputs") is implemented twice, independently: here for the
// device (so 10M..80M-row matrices are born in HBM) and in oracle/sparsemat_oracle.c for the CPU
// checker; tests compare the two bit for bit (integer + exactly-representable float work).
#include <cmath>

#include "internal.hpp"

namespace smh {

constexpr uint64_t kGold = 0x9E3779B97F4A7C15ull;

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += kGold;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t rowkey(uint64_t seed, uint64_t row) { return splitmix64(seed ^ (row * kGold)); }

template <typename T> __host__ __device__ __forceinline__ T hash_to_unit(uint64_t h);
template <> __host__ __device__ __forceinline__ float hash_to_unit<float>(uint64_t h) {
    return (float)(h >> 40) * 0x1p-23f - 1.0f;  // 24 bits: exact
}
template <> __host__ __device__ __forceinline__ double hash_to_unit<double>(uint64_t h) {
    return (double)(h >> 11) * 0x1p-52 - 1.0;  // 53 bits: exact
}

template <typename T>
__global__ void k_synth_x(uint64_t seed, uint64_t begin, uint64_t n, T *__restrict__ x) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x)
        x[j] = hash_to_unit<T>(rowkey(seed, begin + j));
}

// one thread per ENTRY (coalesced stores); k entries per row
template <typename T>
__global__ void k_synth_fixed(uint64_t seed, int pattern, uint64_t n, uint32_t k, uint64_t row_begin,
                              uint64_t row_end, uint32_t *__restrict__ off, uint32_t *__restrict__ col,
                              T *__restrict__ val) {
    uint64_t s = n / k;
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    const uint64_t w = s * k;
    const uint64_t rows = row_end - row_begin;
    const uint64_t total = rows * k;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t lr = e / k;
        const uint32_t j = (uint32_t)(e - lr * k);
        const uint64_t row = row_begin + lr;
        const uint64_t rk = rowkey(seed, row);
        const uint64_t hc = splitmix64(rk + 2ull * j);
        const uint64_t hv = splitmix64(rk + 2ull * j + 1ull);
        int64_t base = (int64_t)row - (int64_t)(w / 2);
        if (base > (int64_t)n - (int64_t)w) base = (int64_t)n - (int64_t)w;
        if (base < 0) base = 0;
        uint64_t c = pattern == 0 ? (uint64_t)base + j * s + hc % s : hc % n;
        if (pattern == 2) {  // contiguous band of k columns centred on the diagonal
            int64_t b2 = (int64_t)row - (int64_t)(k / 2);
            if (b2 > (int64_t)n - (int64_t)k) b2 = (int64_t)n - (int64_t)k;
            if (b2 < 0) b2 = 0;
            c = (uint64_t)b2 + j;
        }
        col[e] = (uint32_t)c;
        val[e] = hash_to_unit<T>(hv);
        if (j == 0) off[lr] = (uint32_t)e;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) off[rows] = (uint32_t)total;
}

// pattern 3 (SURVEY 8(d) "banded" to the letter): k <= kWindowMaxK distinct columns drawn without replacement from
// [row - 4096, row + 4096] within [0, n), stored ascending.  One thread per row: the draws of a row depend on each other.
constexpr uint32_t kWindowMaxK = 64;
template <typename T>
__global__ void k_synth_window(uint64_t seed, uint64_t n, uint32_t k, uint64_t row_begin, uint64_t row_end,
                               uint32_t *__restrict__ off, uint32_t *__restrict__ col, T *__restrict__ val) {
    const uint64_t rows = row_end - row_begin;
    for (uint64_t lr = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; lr < rows; lr += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t row = row_begin + lr;
        const uint64_t rk = rowkey(seed, row);
        const uint64_t lo = row > 4096 ? row - 4096 : 0;
        const uint64_t hi = row + 4096 < n - 1 ? row + 4096 : n - 1;
        const uint64_t ww = hi - lo + 1;
        uint32_t c[kWindowMaxK];
        uint32_t got = 0;
        for (uint64_t t = 0; got < k; ++t) {  // draw t: accepted unless already drawn
            const uint32_t cand = (uint32_t)(lo + splitmix64(rk + 2ull * t) % ww);
            uint32_t q = 0;
            while (q < got && c[q] != cand) ++q;
            if (q == got) c[got++] = cand;
        }
        for (uint32_t a = 1; a < k; ++a) {  // ascending
            const uint32_t v = c[a];
            uint32_t b = a;
            while (b > 0 && c[b - 1] > v) { c[b] = c[b - 1]; --b; }
            c[b] = v;
        }
        const uint64_t o = lr * k;
        off[lr] = (uint32_t)o;
        for (uint32_t j = 0; j < k; ++j) {
            col[o + j] = c[j];
            val[o + j] = hash_to_unit<T>(splitmix64(rk + 2ull * j + 1ull));
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) off[rows] = (uint32_t)(rows * k);
}

// one (sub)wave-strided loop per row: rows given by offsets
template <typename T>
__global__ void k_synth_fill(uint64_t seed, uint64_t n_cols, uint64_t row_begin, uint64_t rows,
                             const uint32_t *__restrict__ off, uint32_t *__restrict__ col, T *__restrict__ val) {
    // 8 lanes per row
    const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const uint32_t lane = threadIdx.x & 7u;
    const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) >> 3;
    for (uint64_t lr = gid; lr < rows; lr += ngroups) {
        const uint64_t rk = rowkey(seed, row_begin + lr);
        const uint64_t o = off[lr];
        const uint32_t len = off[lr + 1] - off[lr];
        for (uint32_t j = lane; j < len; j += 8) {
            const uint64_t hc = splitmix64(rk + 2ull * j);
            const uint64_t hv = splitmix64(rk + 2ull * j + 1ull);
            col[o + j] = (uint32_t)(hc % n_cols);
            val[o + j] = hash_to_unit<T>(hv);
        }
    }
}

// 7-point Laplacian rows [row_begin,row_end): closed-form offsets (prefix of neighbour counts)
__host__ __device__ __forceinline__ uint64_t lap3d_prefix(uint64_t row, uint64_t nx, uint64_t ny, uint64_t nz) {
    // number of entries in rows [0,row): 7*row minus missing neighbours
    const uint64_t nxy = nx * ny;
    const uint64_t k = row / nxy, rem = row - k * nxy, j = rem / nx, i = rem - j * nx;
    // -x faces missing: one per line start among rows < row
    const uint64_t lines_before = k * ny + j;                 // complete lines
    uint64_t miss = 0;
    miss += lines_before + (i > 0 ? 1 : 0);                   // i == 0 cells (no -x)
    miss += lines_before;                                     // i == nx-1 cells (no +x) in complete lines
    // -y missing: cells with j == 0: per complete plane nx, plus partial plane
    miss += k * nx + (j > 0 ? nx : i);
    // +y missing: cells with j == ny-1
    miss += k * nx + (j == ny - 1 ? i : 0);
    // -z missing: cells with k == 0
    miss += k > 0 ? nxy : rem;
    // +z missing: cells with k == nz-1
    miss += k >= nz ? nxy : (k == nz - 1 ? rem : 0);
    return 7 * row - miss;
}

template <typename T>
__global__ void k_synth_laplace3d(uint64_t nx, uint64_t ny, uint64_t nz, uint64_t row_begin, uint64_t rows,
                                  uint32_t *__restrict__ off, uint32_t *__restrict__ col, T *__restrict__ val) {
    const uint64_t nxy = nx * ny;
    const uint64_t base = lap3d_prefix(row_begin, nx, ny, nz);
    for (uint64_t lr = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; lr <= rows;
         lr += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t row = row_begin + lr;
        uint64_t o = lap3d_prefix(row, nx, ny, nz) - base;
        off[lr] = (uint32_t)o;
        if (lr == rows) continue;
        const uint64_t k = row / nxy, rem = row - k * nxy, j = rem / nx, i = rem - j * nx;
        if (k > 0) { col[o] = (uint32_t)(row - nxy); val[o] = T(-1); ++o; }
        if (j > 0) { col[o] = (uint32_t)(row - nx); val[o] = T(-1); ++o; }
        if (i > 0) { col[o] = (uint32_t)(row - 1); val[o] = T(-1); ++o; }
        col[o] = (uint32_t)row; val[o] = T(6); ++o;
        if (i + 1 < nx) { col[o] = (uint32_t)(row + 1); val[o] = T(-1); ++o; }
        if (j + 1 < ny) { col[o] = (uint32_t)(row + nx); val[o] = T(-1); ++o; }
        if (k + 1 < nz) { col[o] = (uint32_t)(row + nxy); val[o] = T(-1); ++o; }
    }
}

static inline unsigned gen_grid(uint64_t work) {
    uint64_t b = (work + kBlock - 1) / kBlock;
    if (b > 4096) b = 4096;
    if (b == 0) b = 1;
    return (unsigned)b;
}

int synth_x(int dtype, uint64_t seed, size_t begin, size_t n, void *x, hipStream_t s) {
    if (n == 0) return SMH_OK;
    if (dtype == SMH_F64) hipLaunchKernelGGL(k_synth_x<double>, dim3(gen_grid(n)), dim3(kBlock), 0, s, seed, (uint64_t)begin, (uint64_t)n, (double *)x);
    else hipLaunchKernelGGL(k_synth_x<float>, dim3(gen_grid(n)), dim3(kBlock), 0, s, seed, (uint64_t)begin, (uint64_t)n, (float *)x);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

int synth_fixed(int dtype, uint64_t seed, int pattern, size_t n, uint32_t k, size_t row_begin, size_t row_end,
                uint32_t *off, uint32_t *col, void *val, hipStream_t s) {
    const uint64_t total = (uint64_t)(row_end - row_begin) * k;
    if (pattern < 0 || pattern > 3) return fail(SMH_ERR_INVALID, "unknown generator pattern %d", pattern);
    if (pattern == 3) {
        if (k > kWindowMaxK) return fail(SMH_ERR_INVALID, "the window pattern draws at most %u columns per row (%u asked)", kWindowMaxK, k);
        const uint64_t rows = row_end - row_begin;
        if (dtype == SMH_F64)
            hipLaunchKernelGGL(k_synth_window<double>, dim3(gen_grid(rows ? rows : 1)), dim3(kBlock), 0, s, seed, (uint64_t)n, k,
                               (uint64_t)row_begin, (uint64_t)row_end, off, col, (double *)val);
        else
            hipLaunchKernelGGL(k_synth_window<float>, dim3(gen_grid(rows ? rows : 1)), dim3(kBlock), 0, s, seed, (uint64_t)n, k,
                               (uint64_t)row_begin, (uint64_t)row_end, off, col, (float *)val);
        SMH_HIP(hipGetLastError());
        return SMH_OK;
    }
    if (dtype == SMH_F64)
        hipLaunchKernelGGL(k_synth_fixed<double>, dim3(gen_grid(total)), dim3(kBlock), 0, s, seed, pattern, (uint64_t)n, k,
                           (uint64_t)row_begin, (uint64_t)row_end, off, col, (double *)val);
    else
        hipLaunchKernelGGL(k_synth_fixed<float>, dim3(gen_grid(total)), dim3(kBlock), 0, s, seed, pattern, (uint64_t)n, k,
                           (uint64_t)row_begin, (uint64_t)row_end, off, col, (float *)val);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

int synth_fill(int dtype, uint64_t seed, size_t n_cols, size_t row_begin, size_t row_end, const uint32_t *off,
               uint32_t *col, void *val, hipStream_t s) {
    const uint64_t rows = row_end - row_begin;
    if (rows == 0) return SMH_OK;
    if (dtype == SMH_F64)
        hipLaunchKernelGGL(k_synth_fill<double>, dim3(gen_grid(rows * 8)), dim3(kBlock), 0, s, seed, (uint64_t)n_cols,
                           (uint64_t)row_begin, rows, off, col, (double *)val);
    else
        hipLaunchKernelGGL(k_synth_fill<float>, dim3(gen_grid(rows * 8)), dim3(kBlock), 0, s, seed, (uint64_t)n_cols,
                           (uint64_t)row_begin, rows, off, col, (float *)val);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

size_t synth_laplace3d_nnz(size_t nx, size_t ny, size_t nz, size_t row_begin, size_t row_end) {
    return (size_t)(lap3d_prefix(row_end, nx, ny, nz) - lap3d_prefix(row_begin, nx, ny, nz));
}

int synth_laplace3d(int dtype, size_t nx, size_t ny, size_t nz, size_t row_begin, size_t row_end, uint32_t *off,
                    uint32_t *col, void *val, hipStream_t s) {
    const uint64_t rows = row_end - row_begin;
    if (dtype == SMH_F64)
        hipLaunchKernelGGL(k_synth_laplace3d<double>, dim3(gen_grid(rows + 1)), dim3(kBlock), 0, s, (uint64_t)nx, (uint64_t)ny,
                           (uint64_t)nz, (uint64_t)row_begin, rows, off, col, (double *)val);
    else
        hipLaunchKernelGGL(k_synth_laplace3d<float>, dim3(gen_grid(rows + 1)), dim3(kBlock), 0, s, (uint64_t)nx, (uint64_t)ny,
                           (uint64_t)nz, (uint64_t)row_begin, rows, off, col, (float *)val);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

// host-side pieces of the power-law generator (same arithmetic, same order as the spec)
void synth_powerlaw_cdf(uint32_t kmax, double alpha, uint32_t *cdf) {
    double z = 0.0;
    for (uint32_t k = 1; k <= kmax; ++k) z += std::pow((double)k, -alpha);
    double acc = 0.0;
    for (uint32_t k = 1; k <= kmax; ++k) {
        acc += std::pow((double)k, -alpha);
        const double f = acc / z * 4294967296.0;
        cdf[k - 1] = f >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)f;
    }
    cdf[kmax - 1] = 0xFFFFFFFFu;
}

void synth_powerlaw_lengths(uint64_t seed, size_t row_begin, size_t row_end, uint32_t kmax, const uint32_t *cdf,
                            uint32_t *lengths) {
    for (size_t row = row_begin; row < row_end; ++row) {
        const uint32_t u = (uint32_t)(splitmix64(rowkey(seed, row) ^ 0xA5A5A5A5A5A5A5A5ull) >> 32);
        uint32_t lo = 0, hi = kmax;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        if (lo > kmax - 1) lo = kmax - 1;
        lengths[row - row_begin] = 1u + lo;
    }
}

}  // namespace smh
