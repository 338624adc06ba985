// spmv_stream.hip -- K1s: CSR-stream SpMV for SHORT rows (stencils, FEM: mean row <= ~12 entries), gfx950.
//
// Same product as the other kernels (reference sparsematrix.rs:146-158 over sparsemat_crs.rs:102-110).
// Why another mapping: with short rows the row-per-lane-group kernels (K1/K1r) are bound by the vector
// memory address path, not by HBM -- rocprofv3 on the 512^3 7-point Laplacian (profiles/r01_pmc_lap512.json):
// TA busy 85 % of the kernel, ~2 clocks per distinct cache line a wave instruction touches, and a lane
// group that covers 16 entry slots for 7 entries makes every load/gather instruction touch 2-3x the lines
// it needs.  K1s streams the entries DENSELY instead:
//
//   * one 256-thread block per tile of 256 consecutive rows (XCD-aware tile order);
//   * the tile's entries [off[r0], off[r1]) are read as one dense run of 16-B aligned chunks (every lane
//     of every load instruction carries 4 useful entries), multiplied by the gathered x[col] and the
//     ROUNDED products parked in LDS (skewed index: no bank conflicts for power-of-two row lengths);
//   * after one barrier, thread r folds the products of row r0+r from LDS SEQUENTIALLY, in storage
//     order, with a rounded add per entry, and the 256 results leave as coalesced stores.
//
// Product rounded, then added in storage order: this is exactly the reference's `sum += rhs.get(j) * val`
// -- K1s is BIT-EXACT against the reference loop (like the SEQ checker), not merely within tolerance.
// A tile with more entries than the LDS stage holds (kStreamCap) is folded straight from global memory
// by the same threads (correct for any matrix; AUTO only picks K1s when no tile overflows).
#include "internal.hpp"

namespace smh {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t skew(uint32_t i) { return i + (i >> 5); }  // +1 word every 32: breaks 2^k strides

__device__ __forceinline__ float st_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ double st_mul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ float st_add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double st_add(double a, double b) { return __dadd_rn(a, b); }

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_spmv_stream(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const T *__restrict__ val,
              const T *__restrict__ x, T *__restrict__ y, uint64_t n_rows, uint64_t nnz, uint64_t nnz_readable,
              uint64_t n_tiles) {
    __shared__ T s_prod[kStreamCap + kStreamCap / 32 + 8];
    // bijective XCD-aware remap: XCD g (= blockIdx % 8) walks a contiguous run of tiles
    const uint64_t q = n_tiles >> 3, rm = n_tiles & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const uint64_t tile = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + idx;
    const uint64_t r0 = tile * kStreamRows;
    const uint64_t r1 = r0 + kStreamRows < n_rows ? r0 + kStreamRows : n_rows;
    const uint32_t tid = threadIdx.x;
    const uint64_t r = r0 + tid;
    const uint32_t o0 = off[r < r1 ? r : r1];
    const uint32_t o1 = off[r + 1 < r1 ? r + 1 : r1];
    const uint32_t k0 = off[r0], k1 = off[r1];  // tile-uniform: scalar loads
    T sum = T(0);
    if (k1 - k0 <= (uint32_t)kStreamCap) {
        // dense run of aligned chunks; the arrays' last partial chunk is read entry by entry when the
        // arrays are not padded (nnz_readable = nnz rounded up for padded arrays)
        for (uint64_t k = (uint64_t)(k0 & ~3u) + 4u * tid; k < k1; k += 4u * kBlock) {
            uint32_t c[4];
            T v[4];
            if (k + 4 <= nnz_readable) {
                const u32x4 cc = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(col + k));
                c[0] = cc.x; c[1] = cc.y; c[2] = cc.z; c[3] = cc.w;
                if constexpr (sizeof(T) == 4) {
                    const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(val + k));
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
                } else {
                    const f64x2 a = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + k));
                    const f64x2 b = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + k + 2));
                    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = k + e < nnz;
                    c[e] = in ? col[k + e] : 0u;
                    v[e] = in ? val[k + e] : T(0);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint64_t i = k + e;
                if (i >= k0 && i < k1) s_prod[skew((uint32_t)(i - k0))] = st_mul(x[c[e]], v[e]);
            }
        }
        __syncthreads();
        // row r: storage order, one rounded add per entry (reference: sum += product)
        uint32_t i = o0 - k0;
        const uint32_t iend = o1 - k0;
        for (; i + 4 <= iend; i += 4) {
            const T p0 = s_prod[skew(i)], p1 = s_prod[skew(i + 1)], p2 = s_prod[skew(i + 2)], p3 = s_prod[skew(i + 3)];
            sum = st_add(sum, p0);
            sum = st_add(sum, p1);
            sum = st_add(sum, p2);
            sum = st_add(sum, p3);
        }
        for (; i < iend; ++i) sum = st_add(sum, s_prod[skew(i)]);
    } else {
        // oversize tile: fold from global memory (same order, same roundings)
        for (uint64_t k = o0; k < o1; ++k) sum = st_add(sum, st_mul(x[col[k]], val[k]));
    }
    if (r < r1) y[r] = sum;
}

int launch_spmv_stream(int dtype, const uint32_t *off, const uint32_t *col, const void *val, const void *x, void *y,
                       size_t n_rows, size_t nnz, bool padded, hipStream_t s) {
    if (n_rows == 0) return SMH_OK;
    const uint64_t n_tiles = (n_rows + kStreamRows - 1) / kStreamRows;
    const uint64_t readable = padded ? ((nnz + 3) & ~uint64_t(3)) : nnz;
    if (dtype == SMH_F64)
        hipLaunchKernelGGL(k_spmv_stream<double>, dim3((unsigned)n_tiles), dim3(kBlock), 0, s, off, col,
                           (const double *)val, (const double *)x, (double *)y, (uint64_t)n_rows, (uint64_t)nnz, readable,
                           n_tiles);
    else
        hipLaunchKernelGGL(k_spmv_stream<float>, dim3((unsigned)n_tiles), dim3(kBlock), 0, s, off, col,
                           (const float *)val, (const float *)x, (float *)y, (uint64_t)n_rows, (uint64_t)nnz, readable,
                           n_tiles);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

// largest number of entries in any 256-row tile (decides whether AUTO may use K1s)
__global__ void __launch_bounds__(kBlock)
k_stream_max_tile(const uint32_t *__restrict__ off, uint64_t n_rows, uint64_t n_tiles, uint32_t *__restrict__ out) {
    uint32_t m = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_tiles; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r0 = t * kStreamRows, r1 = r0 + kStreamRows < n_rows ? r0 + kStreamRows : n_rows;
        const uint32_t e = off[r1] - off[r0];
        m = e > m ? e : m;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_down(m, o, kWave));
    if ((threadIdx.x & (kWave - 1)) == 0) atomicMax(out, m);  // integer max: exact, order independent
}

int launch_stream_max_tile(const uint32_t *off, size_t n_rows, uint32_t *d_out, hipStream_t s) {
    SMH_HIP(hipMemsetAsync(d_out, 0, sizeof(uint32_t), s));
    if (n_rows == 0) return SMH_OK;
    const uint64_t n_tiles = (n_rows + kStreamRows - 1) / kStreamRows;
    uint64_t blocks = (n_tiles + kBlock - 1) / kBlock;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_stream_max_tile, dim3((unsigned)blocks), dim3(kBlock), 0, s, off, (uint64_t)n_rows, n_tiles, d_out);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

}  // namespace smh
