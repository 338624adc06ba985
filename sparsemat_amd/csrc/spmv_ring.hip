// spmv_ring.hip -- K1r: (sub-)wavefront-per-row CSR SpMV with an LDS-resident window of x (gfx950).
//
// Same product as K1 (reference sparsematrix.rs:146-158 over sparsemat_crs.rs:102-110).  K1's limit
// on banded matrices is not HBM but the vector L1: every x[col] gather of a wave instruction touches
// its own cache line (measured: 27 % of HBM peak on the stratified band vs 63 % when the gathers
// coalesce).  K1r moves the gathers to LDS:
//
//   * persistent blocks (512 threads, 2 per CU, 64 KiB of LDS each) own CONTIGUOUS row ranges, and
//     blockIdx -> range is XCD-aware (blocks b, b+8, .. share an XCD and get neighbouring ranges);
//   * an inspector (create time) records the column span of every 64-row tile; the host turns that
//     into a per-block PLAN of phases {rows, x-range to load, ring or not}: the block keeps a
//     sliding window of x in an LDS ring (index = column mod ring size) and each phase only loads
//     the part of its window that the ring does not hold yet, so x is read from L2/HBM about once
//     per block; a tile whose span does not fit the ring falls back to global gathers (uniform
//     branch), so the kernel is correct for ANY matrix -- the plan only changes speed;
//   * a wave owns 64 consecutive rows per step: LANES lanes per row, all 16-B aligned chunk loads of
//     the 64 rows are issued back to back (up to 16 KiB in flight per wave), partial sums are folded
//     with a 64-lane butterfly (__shfl_xor) and transposed in-register so that the 64 results leave
//     in ONE coalesced 256-B store.
#include <algorithm>
#include <vector>

#include "internal.hpp"

namespace smh {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int kRingThreads = 512;
constexpr int kRingBytes = 65536;
constexpr int kRingWaves = kRingThreads / kWave;
constexpr int kTileRows = 64;  // rows a wave handles per step; also the inspector's granularity

// ---- inspector: column span of every 64-row tile ---------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_tile_span(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, uint64_t n_rows, uint64_t n_tiles,
            uint32_t *__restrict__ cmin, uint32_t *__restrict__ cmax) {
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    for (uint64_t t = wave; t < n_tiles; t += n_waves) {
        const uint64_t r0 = t * kTileRows;
        const uint64_t r1 = r0 + kTileRows < n_rows ? r0 + kTileRows : n_rows;
        const uint64_t k0 = off[r0], k1 = off[r1];
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
        for (uint64_t k = k0 + lane; k < k1; k += kWave) {
            const uint32_t c = col[k];
            lo = min(lo, c);
            hi = max(hi, c);
        }
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            lo = min(lo, (uint32_t)__shfl_xor(lo, o, kWave));
            hi = max(hi, (uint32_t)__shfl_xor(hi, o, kWave));
        }
        if (lane == 0) { cmin[t] = lo; cmax[t] = hi; }
    }
}

int launch_tile_span(const uint32_t *off, const uint32_t *col, size_t n_rows, size_t n_tiles, uint32_t *cmin,
                     uint32_t *cmax, hipStream_t s) {
    if (n_tiles == 0) return SMH_OK;
    uint64_t blocks = (n_tiles * kWave + kBlock - 1) / kBlock;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_tile_span, dim3((unsigned)blocks), dim3(kBlock), 0, s, off, col, (uint64_t)n_rows,
                       (uint64_t)n_tiles, cmin, cmax);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

// ---- host: phase plan -------------------------------------------------------------------------------
// Greedy: a phase takes as many consecutive tiles as keep the union of their column spans within the
// ring; what the ring already holds from the previous phase is not reloaded.
void build_ring_plan(size_t n_rows, size_t elem_size, const uint32_t *cmin, const uint32_t *cmax, size_t n_blocks,
                     std::vector<uint32_t> &phase_ptr, std::vector<RingPhase> &phases, double *ring_row_fraction) {
    const uint64_t ring = kRingBytes / elem_size;
    const size_t n_tiles = (n_rows + kTileRows - 1) / kTileRows;
    const size_t tiles_per_block = (n_tiles + n_blocks - 1) / n_blocks;
    phase_ptr.assign(n_blocks + 1, 0);
    phases.clear();
    uint64_t ring_rows = 0;
    for (size_t b = 0; b < n_blocks; ++b) {
        phase_ptr[b] = (uint32_t)phases.size();
        size_t t = std::min(b * tiles_per_block, n_tiles);
        const size_t t_end = std::min(t + tiles_per_block, n_tiles);
        uint64_t lo = 0, hi = 0;  // ring holds columns [lo, hi)
        while (t < t_end) {
            // tiles without entries join any phase
            uint64_t umin = cmin[t], umax = cmax[t];
            const bool empty0 = cmin[t] > cmax[t];
            size_t e = t + 1;
            if (!empty0 && umax + 1 - umin > ring) {
                // span wider than the ring: one global-gather phase over the run of such tiles
                while (e < t_end && cmin[e] <= cmax[e] && (uint64_t)cmax[e] + 1 - cmin[e] > ring) ++e;
                RingPhase p{(uint32_t)(t * kTileRows), (uint32_t)std::min<uint64_t>((uint64_t)e * kTileRows, n_rows), 0, 0, 0};
                phases.push_back(p);
                t = e;
                continue;
            }
            bool any = !empty0;
            while (e < t_end) {
                if (cmin[e] > cmax[e]) { ++e; continue; }
                const uint64_t nmin = any ? std::min<uint64_t>(umin, cmin[e]) : cmin[e];
                const uint64_t nmax = any ? std::max<uint64_t>(umax, cmax[e]) : cmax[e];
                if (nmax + 1 - nmin > ring) break;
                umin = nmin; umax = nmax; any = true;
                ++e;
            }
            RingPhase p{(uint32_t)(t * kTileRows), (uint32_t)std::min<uint64_t>((uint64_t)e * kTileRows, n_rows), 0, 0, 1};
            if (any) {
                if (umin >= lo && umin < hi) {           // continues the window: load only the new part
                    if (umax + 1 > hi) { p.load_lo = (uint32_t)hi; p.load_hi = (uint32_t)(umax + 1); hi = umax + 1; }
                    lo = std::max<uint64_t>(lo, hi > ring ? hi - ring : 0);
                } else {                                 // disjoint (or behind): restart the window
                    p.load_lo = (uint32_t)umin; p.load_hi = (uint32_t)(umax + 1);
                    lo = umin; hi = umax + 1;
                }
            }
            ring_rows += p.row_end - p.row_begin;
            phases.push_back(p);
            t = e;
        }
    }
    phase_ptr[n_blocks] = (uint32_t)phases.size();
    if (ring_row_fraction) *ring_row_fraction = n_rows ? (double)ring_rows / (double)n_rows : 0.0;
}

// ---- kernel -----------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void ring_load_vals4(const T *__restrict__ val, uint64_t k, T (&v)[4]);
template <>
__device__ __forceinline__ void ring_load_vals4<float>(const float *__restrict__ val, uint64_t k, float (&v)[4]) {
    f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(val + k));
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <>
__device__ __forceinline__ void ring_load_vals4<double>(const double *__restrict__ val, uint64_t k, double (&v)[4]) {
    f64x2 a = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + k));
    f64x2 b = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + k + 2));
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}

template <typename T>
__device__ __forceinline__ void ring_load_chunk(const uint32_t *__restrict__ col, const T *__restrict__ val,
                                                uint64_t k, uint64_t nnz, uint32_t (&c)[4], T (&v)[4]) {
    if (k + 4 <= nnz) {
        u32x4 cc = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(col + k));
        c[0] = cc.x; c[1] = cc.y; c[2] = cc.z; c[3] = cc.w;
        ring_load_vals4<T>(val, k, v);
    } else {  // last, partial chunk of the arrays: entry by entry (borrowed arrays carry no padding)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool in = k + e < nnz;
            c[e] = in ? col[k + e] : 0u;
            v[e] = in ? val[k + e] : T(0);
        }
    }
}

__device__ __forceinline__ float ring_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double ring_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// One wave, 64 consecutive rows starting at tile_row: STEPS = LANES steps of 64/LANES rows.
template <typename T, int LANES, bool RING>
__device__ __forceinline__ void ring_tile(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col,
                                          const T *__restrict__ val, const T *__restrict__ x, const T *ring,
                                          T *__restrict__ y, uint64_t tile_row, uint64_t row_end, uint64_t nnz) {
    constexpr int RPS = kWave / LANES;  // rows per step
    constexpr int STEPS = LANES;
    constexpr uint32_t MASK = kRingBytes / sizeof(T) - 1;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t g = lane / LANES, j = lane % LANES;

    // steps are issued in batches of SB (all first chunks of a batch back to back: SB x 2 KiB in flight
    // per wave) -- SB = 4 keeps the kernel under 128 VGPRs, i.e. 4 waves per SIMD / 2 blocks per CU
    constexpr int SB = STEPS < 4 ? STEPS : 4;
    T out = T(0);
#pragma unroll
    for (int tb = 0; tb < STEPS; tb += SB) {
        uint32_t s[SB], e[SB];
#pragma unroll
        for (int t = 0; t < SB; ++t) {
            const uint64_t row = tile_row + (uint64_t)(tb + t) * RPS + g;
            const bool valid = row < row_end;
            s[t] = valid ? off[row] : 0u;
            e[t] = valid ? off[row + 1] : 0u;
        }
        uint32_t c[SB][4];
        T v[SB][4];
        uint64_t k0[SB];
#pragma unroll
        for (int t = 0; t < SB; ++t) {
            k0[t] = ((uint64_t)s[t] & ~uint64_t(3)) + 4u * j;
            if (k0[t] < e[t]) ring_load_chunk<T>(col, val, k0[t], nnz, c[t], v[t]);
        }
#pragma unroll
        for (int t = 0; t < SB; ++t) {
            T sum = T(0);
            const uint64_t ss = s[t], ee = e[t];
            if (k0[t] < ee) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint64_t idx = k0[t] + q;
                    if (idx >= ss && idx < ee) sum = ring_fma(v[t][q], RING ? ring[c[t][q] & MASK] : x[c[t][q]], sum);
                }
                // rows longer than one pass of the lane group
                for (uint64_t k = k0[t] + 4u * LANES; k < ee; k += 4u * LANES) {
                    uint32_t cc[4];
                    T vv[4];
                    ring_load_chunk<T>(col, val, k, nnz, cc, vv);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (k + q < ee) sum = ring_fma(vv[q], RING ? ring[cc[q] & MASK] : x[cc[q]], sum);
                }
            }
#pragma unroll
            for (int o = LANES / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, kWave);
            // in-register transpose: lane L must end up with the sum of row tile_row + L.  Every lane of
            // group gg holds the sum of row step*RPS + gg; lane L fetches group (L % RPS)'s value and keeps
            // it in the step == L / RPS.
            const T got = __shfl(sum, (int)((lane % RPS) * LANES), kWave);
            if ((int)(lane / RPS) == tb + t) out = got;
        }
    }
    const uint64_t row = tile_row + lane;
    if (row < row_end) y[row] = out;
}

template <typename T, int LANES>
__global__ void __launch_bounds__(kRingThreads, 4)  // 4 waves per SIMD = two 512-thread blocks per CU
k_spmv_ring(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const T *__restrict__ val,
            const T *__restrict__ x, T *__restrict__ y, uint64_t nnz, const uint32_t *__restrict__ phase_ptr,
            const RingPhase *__restrict__ phases) {
    __shared__ T ring[kRingBytes / sizeof(T)];
    constexpr uint32_t MASK = kRingBytes / sizeof(T) - 1;
    // XCD-aware: blocks sharing an XCD (b % 8) take neighbouring row ranges
    const uint32_t per_xcd = gridDim.x >> 3;
    const uint32_t lb = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    const uint32_t wave = threadIdx.x / kWave;
    const uint32_t p0 = phase_ptr[lb], p1 = phase_ptr[lb + 1];
    for (uint32_t p = p0; p < p1; ++p) {
        const RingPhase ph = phases[p];
        if (ph.load_hi > ph.load_lo) {
            __syncthreads();  // the previous phase's gathers are done before its slots are overwritten
            for (uint64_t cidx = (uint64_t)ph.load_lo + threadIdx.x; cidx < ph.load_hi; cidx += kRingThreads)
                ring[cidx & MASK] = x[cidx];
            __syncthreads();
        }
        const uint64_t rb = ph.row_begin, re = ph.row_end;
        if (ph.use_ring) {
            for (uint64_t tr = rb + (uint64_t)wave * kTileRows; tr < re; tr += (uint64_t)kRingWaves * kTileRows)
                ring_tile<T, LANES, true>(off, col, val, x, ring, y, tr, re, nnz);
        } else {
            for (uint64_t tr = rb + (uint64_t)wave * kTileRows; tr < re; tr += (uint64_t)kRingWaves * kTileRows)
                ring_tile<T, LANES, false>(off, col, val, x, ring, y, tr, re, nnz);
        }
    }
}

template <typename T>
static int launch_ring_t(int lanes, const uint32_t *off, const uint32_t *col, const T *val, const T *x, T *y,
                         size_t nnz, unsigned n_blocks, const uint32_t *phase_ptr, const RingPhase *phases,
                         hipStream_t s) {
    dim3 grid(n_blocks), block(kRingThreads);
    switch (lanes) {
        case 1: hipLaunchKernelGGL((k_spmv_ring<T, 1>), grid, block, 0, s, off, col, val, x, y, (uint64_t)nnz, phase_ptr, phases); break;
        case 2: hipLaunchKernelGGL((k_spmv_ring<T, 2>), grid, block, 0, s, off, col, val, x, y, (uint64_t)nnz, phase_ptr, phases); break;
        case 4: hipLaunchKernelGGL((k_spmv_ring<T, 4>), grid, block, 0, s, off, col, val, x, y, (uint64_t)nnz, phase_ptr, phases); break;
        case 8: hipLaunchKernelGGL((k_spmv_ring<T, 8>), grid, block, 0, s, off, col, val, x, y, (uint64_t)nnz, phase_ptr, phases); break;
        default: return fail(SMH_ERR_INVALID, "ring kernel: lanes per row must be 1, 2, 4 or 8 (got %d)", lanes);
    }
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

int launch_spmv_ring(int dtype, int lanes, const uint32_t *off, const uint32_t *col, const void *val, const void *x,
                     void *y, size_t nnz, unsigned n_blocks, const uint32_t *phase_ptr, const RingPhase *phases,
                     hipStream_t s) {
    if (dtype == SMH_F64)
        return launch_ring_t<double>(lanes, off, col, (const double *)val, (const double *)x, (double *)y, nnz, n_blocks,
                                     phase_ptr, phases, s);
    return launch_ring_t<float>(lanes, off, col, (const float *)val, (const float *)x, (float *)y, nnz, n_blocks,
                                phase_ptr, phases, s);
}

}  // namespace smh
