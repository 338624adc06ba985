// spmv_merge.hip -- K2: merge-path CSR SpMV for skewed row lengths (gfx950).
//
// Same product as K1 (reference sparsematrix.rs:146-158 over sparsemat_crs.rs:102-110), load
// balanced for rows of 1..2048+ entries: the "merge" of the row-end offsets
// (offset_rows[1..n_rows]) with the entry indices 0..nnz-1 is cut into tiles of kMergeTile
// items, so a tile never holds more than kMergeTile rows + entries however skewed the matrix.
//
//   create time   k_merge_table: one diagonal binary search per tile -> (row, entry) start
//                 coordinates (integer structure, checked bit-exact against the CPU oracle).
//   per SpMV      k_spmv_merge: one 256-thread block per tile.
//                   1. products values[k]*x[columns[k]] of the tile's entry range are streamed
//                      with 16-B aligned coalesced chunks and staged in LDS; the tile's row-end
//                      offsets are staged beside them;
//                   2. every thread finds its own diagonal in LDS and consumes 8 merge items
//                      sequentially (add a product / finish a row);
//                   3. rows that span threads are stitched by a segmented scan (64-lane
//                      __shfl_up inside a wave, LDS across the 4 waves);
//                   4. the partial sum of the row left open at the tile's end goes to a carry
//                      slot per tile.
//                 k_merge_fixup: adds the carries to y in tile order (no float atomics: the
//                 result is bitwise reproducible run to run).
#include "internal.hpp"

namespace smh {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// split of diagonal d: smallest row r with row_end[r] > d-1-r  (row_end = off+1)
__device__ __forceinline__ uint64_t merge_search_global(const uint32_t *__restrict__ off, uint64_t n_rows,
                                                        uint64_t nnz, uint64_t d) {
    uint64_t lo = d > nnz ? d - nnz : 0;
    uint64_t hi = d < n_rows ? d : n_rows;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if ((uint64_t)off[mid + 1] <= d - 1 - mid) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ void k_merge_table(const uint32_t *__restrict__ off, uint64_t n_rows, uint64_t nnz, uint64_t n_tiles,
                              uint32_t *__restrict__ tile_row, uint32_t *__restrict__ tile_nz) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    uint64_t d = t * (uint64_t)kMergeTile;
    if (d > n_rows + nnz) d = n_rows + nnz;
    uint64_t r = merge_search_global(off, n_rows, nnz, d);
    tile_row[t] = (uint32_t)r;
    tile_nz[t] = (uint32_t)(d - r);
}

__device__ __forceinline__ uint64_t xcd_tile(uint64_t bid, uint64_t n_tiles) {
    // bijective XCD-aware remap (blocks b, b+8, ... share an XCD): XCD g gets a contiguous run of tiles
    const uint64_t q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <typename T>
__device__ __forceinline__ void stage_products(const uint32_t *__restrict__ col, const T *__restrict__ val,
                                               const T *__restrict__ x, uint64_t nz0, uint64_t nz1, uint64_t nnz,
                                               T *s_prod) {
    for (uint64_t k = (nz0 & ~uint64_t(3)) + 4u * threadIdx.x; k < nz1; k += 4u * kBlock) {
        uint32_t c[4];
        T v[4];
        if (k + 4 <= nnz) {
            u32x4 cc = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(col + k));
            c[0] = cc.x; c[1] = cc.y; c[2] = cc.z; c[3] = cc.w;
            if constexpr (sizeof(T) == 4) {
                f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(val + k));
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
                f64x2 a = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + k));
                f64x2 b = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + k + 2));
                v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bool in = k + e < nnz;
                c[e] = in ? col[k + e] : 0u;
                v[e] = in ? val[k + e] : T(0);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint64_t idx = k + e;
            if (idx >= nz0 && idx < nz1) s_prod[idx - nz0] = v[e] * x[c[e]];
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_spmv_merge(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const T *__restrict__ val,
             const T *__restrict__ x, T *__restrict__ y, uint64_t n_rows, uint64_t nnz, uint64_t n_tiles,
             const uint32_t *__restrict__ tile_row, const uint32_t *__restrict__ tile_nz,
             uint32_t *__restrict__ carry_row, T *__restrict__ carry_val) {
    __shared__ T s_prod[kMergeTile];
    __shared__ uint32_t s_row_end[kMergeTile + 1];
    __shared__ T s_wave_val[kBlock / kWave];
    __shared__ int s_wave_flag[kBlock / kWave];

    const uint64_t tile = xcd_tile(blockIdx.x, n_tiles);
    const uint32_t tid = threadIdx.x;
    const uint64_t row0 = tile_row[tile], row1 = tile_row[tile + 1];
    const uint64_t nz0 = tile_nz[tile], nz1 = tile_nz[tile + 1];
    const uint32_t tile_rows = (uint32_t)(row1 - row0);
    const uint32_t tile_nnz = (uint32_t)(nz1 - nz0);
    const uint32_t tile_items = tile_rows + tile_nnz;

    // 1. stage row-end offsets (one extra: the row left open at the tile's end) and the products
    for (uint32_t i = tid; i <= tile_rows; i += kBlock) {
        const uint64_t r = row0 + i;
        s_row_end[i] = r < n_rows ? off[r + 1] : 0xFFFFFFFFu;
    }
    stage_products<T>(col, val, x, nz0, nz1, nnz, s_prod);
    __syncthreads();

    // 2. this thread's diagonal inside the tile
    uint32_t d0 = tid * kMergeItemsPerThread;
    if (d0 > tile_items) d0 = tile_items;
    uint32_t d1 = d0 + kMergeItemsPerThread;
    if (d1 > tile_items) d1 = tile_items;
    uint32_t lo = d0 > tile_nnz ? d0 - tile_nnz : 0u;
    uint32_t hi = d0 < tile_rows ? d0 : tile_rows;
    const uint32_t nz0_32 = (uint32_t)nz0;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_row_end[mid] <= nz0_32 + (d0 - 1u - mid)) lo = mid + 1; else hi = mid;
    }
    uint32_t tx = lo, ty = d0 - lo;

    // consume up to 8 merge items: either add the next product or finish row tx
    T running = T(0);
    T first_val = T(0);
    uint32_t first_row = 0;
    int emitted = 0;
    uint32_t row_end = s_row_end[tx];
    for (uint32_t it = d0; it < d1; ++it) {
        if (nz0_32 + ty < row_end) {
            running += s_prod[ty];
            ++ty;
        } else {
            if (!emitted) {
                first_val = running;  // may still miss the part accumulated by earlier threads
                first_row = tx;
                emitted = 1;
            } else {
                y[row0 + tx] = running;
            }
            running = T(0);
            ++tx;
            row_end = s_row_end[tx];
        }
    }

    // 3. segmented inclusive scan of the open-row partials ("emitted" starts a new segment)
    const uint32_t lane = tid & (kWave - 1), wave = tid / kWave;
    T sv = running;
    int sf = emitted;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const T pv = __shfl_up(sv, o, kWave);
        const int pf = __shfl_up(sf, o, kWave);
        if ((int)lane >= o) {
            if (!sf) sv = pv + sv;
            sf |= pf;
        }
    }
    if (lane == kWave - 1) {
        s_wave_val[wave] = sv;
        s_wave_flag[wave] = sf;
    }
    __syncthreads();
    T prefix = T(0);  // combined partial of all earlier waves
    for (uint32_t w = 0; w < wave; ++w) prefix = s_wave_flag[w] ? s_wave_val[w] : prefix + s_wave_val[w];
    const T incl = sf ? sv : prefix + sv;
    T carry_in = __shfl_up(incl, 1, kWave);
    if (lane == 0) carry_in = prefix;

    if (emitted) y[row0 + first_row] = carry_in + first_val;

    // 4. the row still open at the end of the tile
    if (tid == kBlock - 1) {
        carry_row[tile] = row1 < n_rows ? (uint32_t)row1 : 0xFFFFFFFFu;
        carry_val[tile] = incl;
    }
}

template <typename T>
__global__ void k_merge_fixup(T *__restrict__ y, uint64_t n_tiles, const uint32_t *__restrict__ carry_row,
                              const T *__restrict__ carry_val) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const uint32_t r = carry_row[t];
    if (r == 0xFFFFFFFFu) return;
    if (t > 0 && carry_row[t - 1] == r) return;  // only the first tile of a run of equal rows works
    T acc = T(0);
    for (uint64_t u = t; u < n_tiles && carry_row[u] == r; ++u) acc += carry_val[u];
    y[r] = acc + y[r];  // earlier tiles' parts first, then the part of the tile the row ends in
}

int launch_merge_table(const uint32_t *off, size_t n_rows, size_t nnz, size_t n_tiles, uint32_t *tile_row,
                       uint32_t *tile_nz, hipStream_t s) {
    size_t blocks = (n_tiles + 1 + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_merge_table, dim3((unsigned)blocks), dim3(kBlock), 0, s, off, (uint64_t)n_rows,
                       (uint64_t)nnz, (uint64_t)n_tiles, tile_row, tile_nz);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

template <typename T>
static int launch_merge_t(const uint32_t *off, const uint32_t *col, const T *val, const T *x, T *y, size_t n_rows,
                          size_t nnz, size_t n_tiles, const uint32_t *tile_row, const uint32_t *tile_nz,
                          uint32_t *carry_row, T *carry_val, hipStream_t s) {
    hipLaunchKernelGGL(k_spmv_merge<T>, dim3((unsigned)n_tiles), dim3(kBlock), 0, s, off, col, val, x, y,
                       (uint64_t)n_rows, (uint64_t)nnz, (uint64_t)n_tiles, tile_row, tile_nz, carry_row, carry_val);
    SMH_HIP(hipGetLastError());
    size_t blocks = (n_tiles + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_merge_fixup<T>, dim3((unsigned)blocks), dim3(kBlock), 0, s, y, (uint64_t)n_tiles,
                       carry_row, carry_val);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

int launch_spmv_merge(int dtype, const uint32_t *off, const uint32_t *col, const void *val, const void *x, void *y,
                      size_t n_rows, size_t nnz, size_t n_tiles, const uint32_t *tile_row, const uint32_t *tile_nz,
                      uint32_t *carry_row, void *carry_val, hipStream_t s) {
    if (n_rows == 0 || n_tiles == 0) return SMH_OK;
    if (dtype == SMH_F64)
        return launch_merge_t<double>(off, col, (const double *)val, (const double *)x, (double *)y, n_rows, nnz,
                                      n_tiles, tile_row, tile_nz, carry_row, (double *)carry_val, s);
    return launch_merge_t<float>(off, col, (const float *)val, (const float *)x, (float *)y, n_rows, nnz, n_tiles,
                                 tile_row, tile_nz, carry_row, (float *)carry_val, s);
}

}  // namespace smh
