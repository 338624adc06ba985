// spmv_vector.hip -- K1: (sub-)wavefront-per-row CSR SpMV for gfx950, plus the SEQ checker kernel.
//
// Replaces the loop of SparseMatrix::mvp (reference sparsematrix.rs:146-158) over
// SparseMatCRS::iter_row (sparsemat_crs.rs:102-110).
//
// K1 layout: LANES lanes (1..64, power of two; 64 = the literal one-wavefront-per-row kernel)
// own one row.  Each lane streams 16-B-ALIGNED chunks of 4 consecutive entries of
// columns[] (one global_load_dwordx4) and values[] (one / two dwordx4) -- the chunk grid is
// anchored at element 0 of the arrays, not at the row start, so every wave instruction reads
// whole, aligned 16-B pieces whatever offset_rows[] says; entries of a chunk that belong to
// a neighbouring row are masked out.  x[col] is gathered from L2 / Infinity Cache; partial
// sums are combined with a 64-lane-wave butterfly (__shfl_xor).  HBM-bound: no MFMA.
//
// blockIdx -> rows mapping is XCD-aware: blocks b, b+8, b+16 ... share an XCD (round-robin
// dispatch), so XCD g = b%8 sweeps the g-th contiguous eighth of the rows and its private
// 4 MiB L2 only ever holds that eighth's window of x.  Placement changes speed only.
#include "internal.hpp"

namespace smh {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <typename T>
__device__ __forceinline__ void load_vals4(const T *__restrict__ val, uint64_t k, T (&v)[4]);

template <>
__device__ __forceinline__ void load_vals4<float>(const float *__restrict__ val, uint64_t k, float (&v)[4]) {
    f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(val + k));
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <>
__device__ __forceinline__ void load_vals4<double>(const double *__restrict__ val, uint64_t k, double (&v)[4]) {
    f64x2 a = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + k));
    f64x2 b = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + k + 2));
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}

// One 4-entry chunk starting at the 4-aligned element k.  The last chunk of the ARRAYS may
// reach past nnz (borrowed device arrays carry no padding): that one is read entry by entry.
template <typename T>
__device__ __forceinline__ void load_chunk(const uint32_t *__restrict__ col, const T *__restrict__ val,
                                           uint64_t k, uint64_t nnz, uint32_t (&c)[4], T (&v)[4]) {
    if (k + 4 <= nnz) {
        u32x4 cc = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(col + k));
        c[0] = cc.x; c[1] = cc.y; c[2] = cc.z; c[3] = cc.w;
        load_vals4<T>(val, k, v);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bool in = k + e < nnz;
            c[e] = in ? col[k + e] : 0u;
            v[e] = in ? val[k + e] : T(0);
        }
    }
}

// Two consecutive entries starting at the EVEN element k (f64's 16-byte piece); past the arrays' end entry by entry.
typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ void load_pair(const uint32_t *__restrict__ col, const T *__restrict__ val, uint64_t k, uint64_t nnz,
                                          uint32_t &c0, uint32_t &c1, T &v0, T &v1) {
    static_assert(sizeof(T) == 8, "the two-entry piece is f64's");
    if (k + 2 <= nnz) {
        const u32x2v cc = __builtin_nontemporal_load(reinterpret_cast<const u32x2v *>(col + k));
        const f64x2 a = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + k));
        c0 = cc.x; c1 = cc.y; v0 = a.x; v1 = a.y;
    } else {
        const bool in0 = k < nnz;
        c0 = in0 ? col[k] : 0u; v0 = in0 ? val[k] : T(0);
        c1 = 0u; v1 = T(0);
    }
}

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename T, int LANES, int UNROLL>
__global__ void __launch_bounds__(kBlock)
k_spmv_vector(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const T *__restrict__ val,
              const T *__restrict__ x, T *__restrict__ y, uint64_t n_rows, uint64_t nnz,
              uint64_t rows_per_xcd) {
    constexpr int GROUPS = kBlock / LANES;  // rows a block covers per step
    const uint32_t lane = threadIdx.x % LANES;
    const uint32_t group = threadIdx.x / LANES;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t local = blockIdx.x >> 3;
    const uint32_t blocks_per_xcd = gridDim.x >> 3;

    const uint64_t xcd_begin = (uint64_t)xcd * rows_per_xcd;
    uint64_t xcd_end = xcd_begin + rows_per_xcd;
    if (xcd_end > n_rows) xcd_end = n_rows;
    const uint64_t step = (uint64_t)blocks_per_xcd * GROUPS;  // rows between two steps of this block

    // block-uniform loop (every lane of a wave stays in it so the shuffles are well defined)
    for (uint64_t base = xcd_begin + (uint64_t)local * GROUPS; base < xcd_end; base += step * UNROLL) {
        uint32_t start[UNROLL], end[UNROLL];
        uint64_t row[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            row[u] = base + (uint64_t)u * step + group;
            bool valid = row[u] < xcd_end;
            start[u] = valid ? off[row[u]] : 0u;
            end[u] = valid ? off[row[u] + 1] : 0u;
        }
        T sum[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            sum[u] = T(0);
            const uint64_t s = start[u], e = end[u];
            if constexpr (sizeof(T) == 8) {
                // f64 (round 4): a lane's 32 bytes of values per pass are TWO 16-byte pieces -- entries 2 lane, 2 lane + 1 and 2 LANES +
                // 2 lane, 2 LANES + 2 lane + 1 of the pass -- so that each load instruction of the lane group covers contiguous bytes
                // (K1r's lesson, spmv_ring2.hip lane_pos): the same slots, the same order of FMAs per lane as K1r, hence its bits.
                for (uint64_t kb = s & ~uint64_t(3); kb + 2u * lane < e; kb += 4u * LANES) {
                    uint32_t c[4];
                    T v[4];
                    const uint64_t i0 = kb + 2u * lane, i1 = kb + 2u * LANES + 2u * lane;
                    load_pair<T>(col, val, i0, nnz, c[0], c[1], v[0], v[1]);
                    load_pair<T>(col, val, i1, nnz, c[2], c[3], v[2], v[3]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint64_t idx = (j < 2 ? i0 : i1 - 2u) + (uint64_t)j;
                        if (idx >= s && idx < e) sum[u] = fma_t(v[j], x[c[j]], sum[u]);
                    }
                }
            } else {
                for (uint64_t k = (s & ~uint64_t(3)) + 4u * lane; k < e; k += 4u * LANES) {
                    uint32_t c[4];
                    T v[4];
                    load_chunk<T>(col, val, k, nnz, c, v);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint64_t idx = k + j;
                        if (idx >= s && idx < e) sum[u] = fma_t(v[j], x[c[j]], sum[u]);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int o = LANES / 2; o > 0; o >>= 1) sum[u] += __shfl_xor(sum[u], o, kWave);
            if (lane == 0 && row[u] < xcd_end) y[row[u]] = sum[u];
        }
    }
}

// SEQ: one lane per row, STORAGE order, product rounded then added (two roundings, no FMA):
// bit-for-bit the reference's `sum += rhs.get(j) * val` (sparsematrix.rs:151-154).
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_spmv_seq(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const T *__restrict__ val,
           const T *__restrict__ x, T *__restrict__ y, uint64_t n_rows) {
    for (uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; row < n_rows;
         row += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t s = off[row], e = off[row + 1];
        T sum = T(0);
        for (uint64_t k = s; k < e; ++k) {
            T prod;
            if constexpr (sizeof(T) == 4) {
                prod = __fmul_rn(x[col[k]], val[k]);
                sum = __fadd_rn(sum, prod);
            } else {
                prod = __dmul_rn(x[col[k]], val[k]);
                sum = __dadd_rn(sum, prod);
            }
        }
        y[row] = sum;
    }
}

template <typename T, int LANES>
static int launch_vector_t(const uint32_t *off, const uint32_t *col, const T *val, const T *x, T *y,
                           size_t n_rows, size_t nnz, hipStream_t s) {
    constexpr int UNROLL = 2;
    constexpr int GROUPS = kBlock / LANES;
    // rows per XCD: a multiple of GROUPS so XCD ranges start on a block-step boundary
    uint64_t rows_per_xcd = (n_rows + 7) / 8;
    rows_per_xcd = (rows_per_xcd + GROUPS - 1) / GROUPS * GROUPS;
    uint64_t steps = (rows_per_xcd + (uint64_t)GROUPS * UNROLL - 1) / ((uint64_t)GROUPS * UNROLL);
    uint64_t blocks_per_xcd = steps < 256 ? steps : 256;  // 256 CUs x 8 blocks fill the chip
    if (blocks_per_xcd == 0) blocks_per_xcd = 1;
    dim3 grid((unsigned)(blocks_per_xcd * 8));
    hipLaunchKernelGGL((k_spmv_vector<T, LANES, UNROLL>), grid, dim3(kBlock), 0, s, off, col, val, x, y,
                       (uint64_t)n_rows, (uint64_t)nnz, rows_per_xcd);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

template <typename T>
static int launch_vector_lanes(int lanes, const uint32_t *off, const uint32_t *col, const T *val, const T *x,
                               T *y, size_t n_rows, size_t nnz, hipStream_t s) {
    switch (lanes) {
        case 1: return launch_vector_t<T, 1>(off, col, val, x, y, n_rows, nnz, s);
        case 2: return launch_vector_t<T, 2>(off, col, val, x, y, n_rows, nnz, s);
        case 4: return launch_vector_t<T, 4>(off, col, val, x, y, n_rows, nnz, s);
        case 8: return launch_vector_t<T, 8>(off, col, val, x, y, n_rows, nnz, s);
        case 16: return launch_vector_t<T, 16>(off, col, val, x, y, n_rows, nnz, s);
        case 32: return launch_vector_t<T, 32>(off, col, val, x, y, n_rows, nnz, s);
        case 64: return launch_vector_t<T, 64>(off, col, val, x, y, n_rows, nnz, s);
        default: return fail(SMH_ERR_INVALID, "vector kernel: lanes per row must be 1,2,4,...,64 (got %d)", lanes);
    }
}

int launch_spmv_vector(int dtype, int lanes, const uint32_t *off, const uint32_t *col, const void *val,
                       const void *x, void *y, size_t n_rows, size_t nnz, hipStream_t s) {
    if (n_rows == 0) return SMH_OK;
    if (dtype == SMH_F64)
        return launch_vector_lanes<double>(lanes, off, col, (const double *)val, (const double *)x, (double *)y,
                                           n_rows, nnz, s);
    return launch_vector_lanes<float>(lanes, off, col, (const float *)val, (const float *)x, (float *)y, n_rows,
                                      nnz, s);
}

int launch_spmv_seq(int dtype, const uint32_t *off, const uint32_t *col, const void *val, const void *x, void *y,
                    size_t n_rows, hipStream_t s) {
    if (n_rows == 0) return SMH_OK;
    size_t blocks = (n_rows + kBlock - 1) / kBlock;
    if (blocks > 8192) blocks = 8192;
    if (dtype == SMH_F64)
        hipLaunchKernelGGL(k_spmv_seq<double>, dim3((unsigned)blocks), dim3(kBlock), 0, s, off, col,
                           (const double *)val, (const double *)x, (double *)y, (uint64_t)n_rows);
    else
        hipLaunchKernelGGL(k_spmv_seq<float>, dim3((unsigned)blocks), dim3(kBlock), 0, s, off, col,
                           (const float *)val, (const float *)x, (float *)y, (uint64_t)n_rows);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

}  // namespace smh
