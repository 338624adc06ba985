// spmv_colfused.hip -- K2f: column-blocked SpMV in ONE sweep over y (columns without locality), gfx950.
//
// Same product as the other kernels (reference sparsematrix.rs:146-158 over sparsemat_crs.rs:102-110).
// K2c (spmv_colblock.hip) keeps the gathered part of x L2-resident by running one launch per column block,
// `y += A_b x`: every launch sweeps the per-block row offsets and reads and rewrites all of y.  rocprofv3 on BASELINE
// C3 (f64 power law, 20 blocks of 2^19 columns; profiles/r02_pmc_k2c_powerlaw.json): 629 MB fetched + 80 MB written per
// launch, 14.2 GB per product for 4.0 GB of algorithmic bytes -- and with 4 MiB of f64 x per block (a whole L2) one gather
// in six misses; halving the blocks doubles the sweeps (5.08 ms).  K2f removes the sweeps instead of trading them:
//
//   * a WAVE owns a tile of 64 x RT consecutive rows (lane l: rows l*RT .. l*RT+RT-1 of the tile) and keeps their RT
//     running sums in REGISTERS while it walks the column blocks 0, 1, ..., B-1; y is written once, at the end;
//   * all waves of a launch walk the blocks in the same order, so at any time the gathers of an XCD fall into a few
//     neighbouring blocks of x (1-2 MiB each), which its L2 holds -- the launch is sized to the waves the chip holds at once (a "round";
//     a matrix with more rows than that takes several rounds, each re-reading x through the Infinity Cache);
//   * the device copy is laid out for exactly this walk: entries sorted by (tile, column block, row, storage order), so a
//     wave streams ONE contiguous run of aligned 16-byte chunks; per (tile, block) one u32 (where its entries start) and
//     per (row, block) one BYTE (how many entries), instead of K2c's u32 offset per (row, block);
//   * the products of a pass are parked in a wave-private LDS stage and each lane folds the entries of its rows in
//     storage order (wave-private: no block barrier anywhere in the kernel).
//
// BALANCED TILES.  A tile is 64 x h consecutive rows with h <= RT chosen per tile (greedy, in row order, on the host at
// build time) so that it holds at most 1.25x the entries of a mean full-height tile: a stretch of long rows gets shorter
// tiles, a regular matrix gets full-height tiles throughout.
// WHERE IT WORKS (measured, profiles/r02_k2f_sweep.log).  L2 locality only holds while the waves of an XCD stay within a
// block or two of each other.  Rows of similar length keep them together by themselves: C2-uniform (10M x 32, f32) runs in
// 1.98-2.05 ms against K2c's 2.19 ms, at the L2-hit gather rate (~160 G gathers/s).  Skewed rows do not: on BASELINE C3
// (f64, power law 1..2048 -- two thirds of the entries sit in rows of 256 and more) the waves drift over all blocks at
// once (5.3-5.6 ms, K2c 3.25 ms): a lane folds its rows sequentially, so a wave with a 2048-entry row falls behind, misses
// more, and falls further behind.  (A lock step between the waves of an XCD -- per-(XCD, block) counters, a wave entering block b
// only when all waves of its XCD had left block b - lag -- was built and measured in round 2 and cost more than it recovered: a
// barrier aligns the phases of all waves, everyone streams, then everyone gathers, and the two stop overlapping: C2-uniform 3.75 ms
// with lag 1, 2.37 with lag 3; C3 10.7 / 5.2 ms.  Removed in round 3; profiles/r02_k2f_sweep.log.)  AUTO takes K2f only for matrices
// whose longest row is within 2x the mean and leaves the rest to K2c / K2t.
//
// A row's sum is formed block by block (ascending column block, storage order inside a block): not the reference's
// order, so tolerance parity like K1r/K2/K2c -- and bit-exact against the oracle applied to the permuted rows, which the
// tests check.  Deterministic, bitwise reproducible.  The split itself is integer work, checked bit-exact in the tests.
// A (row, block) pair with more than 255 entries cannot be described by the byte table: the builder reports it and the
// caller stays with K2c.
#include "internal.hpp"

#include <vector>

namespace smh {

int device_exclusive_scan_u32(uint32_t *data, uint64_t n, hipStream_t s, uint64_t *total_out);  // spmv_colblock.hip

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float cf_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ double cf_mul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ float cf_add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double cf_add(double a, double b) { return __dadd_rn(a, b); }

// tile of row r: the last t with tile_row[t] <= r (tile_row[n_tiles] = n_rows)
__device__ __forceinline__ uint32_t cf_tile_of(const uint32_t *__restrict__ tile_row, uint32_t n_tiles, uint64_t r) {
    uint32_t lo = 0, hi = n_tiles;  // invariant: tile_row[lo] <= r < tile_row[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((uint64_t)tile_row[mid] <= r) lo = mid; else hi = mid;
    }
    return lo;
}

// index of (row r, column block 0) in the count table / the scan array, whose order (tile, block, lane, j) is the storage
// order of the K2f copy; block b is b * 64 * rt further.  Tile t holds rows [tile_row[t], tile_row[t+1]) = 64 lanes x h
// rows, lane = (r - first) / h, j = (r - first) % h
__device__ __forceinline__ uint64_t cf_index0(const uint32_t *__restrict__ tile_row, uint32_t n_tiles, uint64_t r, uint32_t n_blocks,
                                              uint32_t rt) {
    const uint32_t t = cf_tile_of(tile_row, n_tiles, r);
    const uint32_t first = tile_row[t], h = (tile_row[t + 1] - first + 63u) / 64u;
    const uint32_t in = (uint32_t)r - first;
    return ((uint64_t)t * n_blocks * 64ull + in / h) * rt + in % h;
}

// Greedy tiling in row order (host side, on a copy of offset_rows: a few thousand short searches, once per matrix): tile t
// starts at tile_row[t] and takes the largest h in 1..rt whose 64 h rows hold at most `target` entries (h = 1 if even 64
// rows exceed it); the last tile takes what is left (its row count need not be a multiple of 64: h = ceil(rows / 64)).
static void cf_tiles_host(const uint32_t *off, uint64_t n_rows, uint32_t rt, uint64_t target, std::vector<uint32_t> &tile_row) {
    tile_row.clear();
    uint64_t r = 0;
    while (r < n_rows) {
        tile_row.push_back((uint32_t)r);
        const uint64_t base = off[r];
        uint32_t lo = 1, hi = rt;  // entries(h) grows with h: bisect for the largest admissible h
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            const uint64_t e = r + 64ull * mid < n_rows ? r + 64ull * mid : n_rows;
            if ((uint64_t)off[e] - base <= target) lo = mid; else hi = mid - 1;
        }
        r = r + 64ull * lo < n_rows ? r + 64ull * lo : n_rows;
    }
    tile_row.push_back((uint32_t)n_rows);
}

// cur[index(r, b)] = entries of row r in column block b (u32, scanned afterwards); *overflow |= 1 when one exceeds 255
__global__ void __launch_bounds__(kBlock)
k_cf_count(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, uint64_t n_rows, uint32_t shift,
           uint32_t n_blocks, uint32_t rt, const uint32_t *__restrict__ tile_row, uint32_t n_tiles, uint32_t *__restrict__ cur,
           uint32_t *__restrict__ overflow) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k1 = off[r + 1];
        bool over = false;
        uint32_t *mine = cur + cf_index0(tile_row, n_tiles, r, n_blocks, rt);  // row r is this thread's
        for (uint64_t k = off[r]; k < k1; ++k) {
            uint32_t *slot = mine + (uint64_t)(col[k] >> shift) * 64ull * rt;
            const uint32_t c = *slot + 1u;
            *slot = c;
            over |= c > 255u;
        }
        if (over) atomicOr(overflow, 1u);
    }
}

// cnt8[i] = (u8) cur[i]   (before the scan turns the counts into positions)
__global__ void __launch_bounds__(kBlock)
k_cf_narrow(const uint32_t *__restrict__ cur, uint64_t n, uint8_t *__restrict__ cnt8) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        cnt8[i] = (uint8_t)cur[i];
}

// seg[t * B + b] = position of the first entry of (tile t, block b) = scanned cur at (t, b, lane 0, j 0)
__global__ void __launch_bounds__(kBlock)
k_cf_segments(const uint32_t *__restrict__ cur, uint64_t n_segments, uint32_t rt, uint32_t nnz, uint32_t *__restrict__ seg) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n_segments; i += (uint64_t)gridDim.x * blockDim.x)
        seg[i] = i < n_segments ? cur[i * 64ull * rt] : nnz;
}

// entry k of row r goes to cur[index(r, block(col[k]))]++ : storage order kept inside a (row, block) pair
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_cf_scatter(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const T *__restrict__ val, uint64_t n_rows,
             uint32_t shift, uint32_t n_blocks, uint32_t rt, const uint32_t *__restrict__ tile_row, uint32_t n_tiles,
             uint32_t *__restrict__ cur, uint32_t *__restrict__ col2, T *__restrict__ val2) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k1 = off[r + 1];
        uint32_t *mine = cur + cf_index0(tile_row, n_tiles, r, n_blocks, rt);
        for (uint64_t k = off[r]; k < k1; ++k) {
            const uint32_t c = col[k];
            uint32_t *slot = mine + (uint64_t)(c >> shift) * 64ull * rt;
            const uint32_t pos = *slot;
            *slot = pos + 1u;
            col2[pos] = c;
            val2[pos] = val[k];
        }
    }
}

// ---- the product ------------------------------------------------------------------------------------------------------
// RT rows per lane at most, NIT 16-byte chunks per lane and pass (a pass stages 256 * NIT entries per wave).
template <typename T, int RT, int NIT>
__global__ void __launch_bounds__(kBlock)
k_spmv_colfused(const uint32_t *__restrict__ tile_row, const uint32_t *__restrict__ seg, const uint8_t *__restrict__ cnt,
                const uint32_t *__restrict__ col, const T *__restrict__ val, const T *__restrict__ x, T *__restrict__ y,
                uint32_t n_blocks, uint64_t tile_begin, uint64_t tile_end, uint64_t nnz_readable) {
    static_assert(RT == 8 || RT == 16, "counts of a lane are read as one 8- or 16-byte word");
    constexpr uint32_t P = 4u * kWave * NIT;  // entries per pass
    __shared__ T s_prod[kBlock / kWave][P + 8];
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    // XCD-aware: blockIdx % 8 is the XCD; the tiles of an XCD are a contiguous run (its share of y and of the stream).
    // gridDim.x is a multiple of 8 (launcher): every XCD has q waves, of which those with tile < tile_end take part.
    const uint64_t q = (uint64_t)(gridDim.x >> 3) * (kBlock / kWave), xcd = blockIdx.x & 7;
    const uint64_t in_xcd = (uint64_t)(blockIdx.x >> 3) * (kBlock / kWave) + wave;
    const uint64_t tile = tile_begin + xcd * q + in_xcd;
    if (tile >= tile_end) return;  // wave-uniform; the kernel has no block-wide barrier
    T *stage = s_prod[wave];
    T acc[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) acc[j] = T(0);
    const uint32_t row_first = tile_row[tile];
    const uint32_t h = (tile_row[tile + 1] - row_first + 63u) / 64u;  // rows per lane of this tile (<= RT)
    // (build_colfused verifies the tile table before any kernel sees it; a tile of more than 64 RT rows -- the inconsistency behind
    // round 2's memory fault during bring-up, DESIGN.md "K2f" -- would make the lane ranges below walk past the count table)
    if (h > (uint32_t)RT) return;
    const uint32_t *seg_t = seg + tile * n_blocks;
    const uint8_t *cnt_t = cnt + (tile * n_blocks * kWave + lane) * RT;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        const uint32_t s0 = seg_t[b], s1 = seg_t[b + 1];  // wave-uniform (scalar loads)
        if (s1 != s0) {
            // this lane's RT counts, its total, and where its entries start (exclusive wave scan of the totals)
            uint32_t cw[RT / 4];
            if constexpr (RT == 16) {
                const u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(cnt_t + (uint64_t)b * kWave * RT));  // (read once: leave L2 to x)
                cw[0] = w.x; cw[1] = w.y; cw[2] = w.z; cw[3] = w.w;
            } else {
                const u32x2 w = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(cnt_t + (uint64_t)b * kWave * RT));
                cw[0] = w.x; cw[1] = w.y;
            }
            uint32_t total = 0;
#pragma unroll
            for (int wd = 0; wd < RT / 4; ++wd)
                total += (cw[wd] & 0xFFu) + ((cw[wd] >> 8) & 0xFFu) + ((cw[wd] >> 16) & 0xFFu) + (cw[wd] >> 24);
            uint32_t incl = total;
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, o, kWave);
                if ((int)lane >= o) incl += up;
            }
            const uint32_t t0 = s0 + incl - total, t1 = s0 + incl;  // this lane's entries [t0, t1)
            for (uint32_t ps = s0; ps < s1;) {
                const uint32_t pa = ps & ~3u;                       // aligned start of the pass
                const uint32_t pe = s1 - pa > P ? pa + P : s1;     // entries [ps, pe) are staged at stage[i - pa]
                const uint32_t *__restrict__ colp = col + pa;
                const T *__restrict__ valp = val + pa;
                uint32_t c[NIT][4];
                T v[NIT][4];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const uint32_t jj = 4u * lane + (uint32_t)it * (4u * kWave);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { c[it][e] = 0u; v[it][e] = T(0); }
                    if (pa + jj < pe) {
                        if ((uint64_t)pa + jj + 4 <= nnz_readable) {
                            const u32x4 cc = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(colp + jj));
                            c[it][0] = cc.x; c[it][1] = cc.y; c[it][2] = cc.z; c[it][3] = cc.w;
                            if constexpr (sizeof(T) == 4) {
                                const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(valp + jj));
                                v[it][0] = a.x; v[it][1] = a.y; v[it][2] = a.z; v[it][3] = a.w;
                            } else {
                                const f64x2 a0 = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(valp + jj));
                                const f64x2 a1 = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(valp + jj + 2));
                                v[it][0] = a0.x; v[it][1] = a0.y; v[it][2] = a1.x; v[it][3] = a1.y;
                            }
                        } else {  // (never taken: the copy is padded to whole chunks; kept for arrays that are not)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if ((uint64_t)pa + jj + e < nnz_readable) { c[it][e] = colp[jj + e]; v[it][e] = valp[jj + e]; }
                        }
                    }
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const uint32_t jj = 4u * lane + (uint32_t)it * (4u * kWave);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t i = pa + jj + e;
                        if (i >= ps && i < pe) stage[jj + e] = cf_mul(x[c[it][e]], v[it][e]);  // rounded product (rhs.get(j) * val)
                    }
                }
                __builtin_amdgcn_wave_barrier();
                // fold: the part of this lane's entries that the pass holds, row by row, in storage order
                if (t0 < pe && t1 > ps) {
                    uint32_t e0 = t0;
#pragma unroll
                    for (int j = 0; j < RT; ++j) {
                        const uint32_t cj = (cw[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                        const uint32_t lo = e0 > ps ? e0 : ps, hi = e0 + cj < pe ? e0 + cj : pe;
                        T a = acc[j];
                        for (uint32_t i = lo; i < hi; ++i) a = cf_add(a, stage[i - pa]);  // rounded add (sum += ...)
                        acc[j] = a;
                        e0 += cj;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                ps = pe;
            }
        }
    }
    const uint64_t r0 = (uint64_t)row_first + (uint64_t)lane * h, r_end = tile_row[tile + 1];
#pragma unroll
    for (int j = 0; j < RT; ++j)
        if ((uint32_t)j < h && r0 + j < r_end) y[r0 + j] = acc[j];
}

// ---- host side ------------------------------------------------------------------------------------------------------------
static unsigned cf_rows_grid(uint64_t n) {
    uint64_t b = (n + kBlock - 1) / kBlock;
    if (b > 8192) b = 8192;
    if (b == 0) b = 1;
    return (unsigned)b;
}

// Build the K2f copy.  Outputs (device, owned by the caller): tile_row [n_tiles + 1], seg [n_tiles * n_blocks + 1],
// cnt [n_tiles * n_blocks * 64 * rt] bytes, col2 / val2 [nnz + 4].  *fits_out = false (and nothing allocated) when a
// (row, block) pair exceeds 255 entries.
int build_colfused(int dtype, const uint32_t *off, const uint32_t *col, const void *val, size_t n_rows, size_t nnz, uint32_t shift,
                   size_t n_blocks, uint32_t rt, size_t *n_tiles_out, uint32_t **tile_row_out, uint32_t **seg_out, uint8_t **cnt_out,
                   uint32_t **col2_out, void **val2_out, bool *fits_out, hipStream_t s) {
    const uint64_t tile_rows = 64ull * rt;
    const size_t vs = dtype_size(dtype);
    *fits_out = false;
    *n_tiles_out = 0;
    uint32_t *cur = nullptr, *seg = nullptr, *col2 = nullptr, *d_over = nullptr, *tile_row = nullptr;
    uint8_t *cnt8 = nullptr;
    void *val2 = nullptr;
    bool fits = false;
    uint64_t n_tiles = 0;
    auto body = [&]() -> int {
        // tiles of at most 1.25x the entries of a mean full-height tile
        SMH_HIP(hipMalloc((void **)&d_over, sizeof(uint32_t)));
        SMH_HIP(hipMemsetAsync(d_over, 0, sizeof(uint32_t), s));
        const double mean_tile = n_rows ? (double)nnz * (double)tile_rows / (double)n_rows : 0.0;
        uint64_t target = (uint64_t)(1.25 * mean_tile) + 1;
        if (const char *e = getenv("SMH_COLFUSED_BALANCE")) {  // tuning knob: 0 = full-height tiles whatever they hold
            if (atoi(e) == 0) target = ~uint64_t(0);
        }
        std::vector<uint32_t> h_off(n_rows + 1), h_tiles;
        SMH_HIP(hipMemcpyAsync(h_off.data(), off, (n_rows + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        SMH_HIP(hipStreamSynchronize(s));
        cf_tiles_host(h_off.data(), n_rows, rt, target, h_tiles);
        n_tiles = h_tiles.size() - 1;
        // (what every kernel below relies on: strictly increasing starts, at most 64 rt rows per tile, all rows covered)
        for (uint64_t t = 0; t < n_tiles; ++t)
            if (h_tiles[t + 1] <= h_tiles[t] || (uint64_t)h_tiles[t + 1] - h_tiles[t] > tile_rows)
                return fail(SMH_ERR_INVALID, "fused column-blocked copy: malformed tile table at tile %llu", (unsigned long long)t);
        if (h_tiles[0] != 0 || h_tiles[n_tiles] != n_rows) return fail(SMH_ERR_INVALID, "fused column-blocked copy: tile table does not cover the rows");
        SMH_HIP(hipMalloc((void **)&tile_row, (n_tiles + 1) * sizeof(uint32_t)));
        SMH_HIP(hipMemcpyAsync(tile_row, h_tiles.data(), (n_tiles + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        SMH_HIP(hipStreamSynchronize(s));  // (h_tiles is a local)
        const uint64_t n_seg = n_tiles * n_blocks, total = n_seg * tile_rows;
        if (total >= (1ull << 34)) return fail(SMH_ERR_OOM, "fused column-blocked copy: count table too large");
        SMH_HIP(hipMalloc((void **)&cur, (total ? total : 1) * sizeof(uint32_t)));
        SMH_HIP(hipMemsetAsync(cur, 0, (total ? total : 1) * sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_cf_count, dim3(cf_rows_grid(n_rows)), dim3(kBlock), 0, s, off, col, (uint64_t)n_rows, shift,
                           (uint32_t)n_blocks, rt, tile_row, (uint32_t)n_tiles, cur, d_over);
        SMH_HIP(hipGetLastError());
        uint32_t h_over = 0;
        SMH_HIP(hipMemcpyAsync(&h_over, d_over, sizeof h_over, hipMemcpyDeviceToHost, s));
        SMH_HIP(hipStreamSynchronize(s));
        if (h_over) return SMH_OK;  // a (row, block) pair with more than 255 entries: not describable
        fits = true;
        SMH_HIP(hipMalloc((void **)&cnt8, total ? total : 1));
        hipLaunchKernelGGL(k_cf_narrow, dim3(cf_rows_grid(total)), dim3(kBlock), 0, s, cur, total, cnt8);
        SMH_HIP(hipGetLastError());
        SMH_TRY(device_exclusive_scan_u32(cur, total, s, nullptr));
        SMH_HIP(hipMalloc((void **)&seg, (n_seg + 1) * sizeof(uint32_t)));
        hipLaunchKernelGGL(k_cf_segments, dim3(cf_rows_grid(n_seg + 1)), dim3(kBlock), 0, s, cur, n_seg, rt, (uint32_t)nnz, seg);
        SMH_HIP(hipGetLastError());
        SMH_HIP(hipMalloc((void **)&col2, (nnz + 4) * sizeof(uint32_t)));
        SMH_HIP(hipMalloc(&val2, (nnz + 4) * vs));
        SMH_HIP(hipMemsetAsync(col2 + nnz, 0, 4 * sizeof(uint32_t), s));
        SMH_HIP(hipMemsetAsync((char *)val2 + nnz * vs, 0, 4 * vs, s));
        if (dtype == SMH_F64)
            hipLaunchKernelGGL(k_cf_scatter<double>, dim3(cf_rows_grid(n_rows)), dim3(kBlock), 0, s, off, col, (const double *)val,
                               (uint64_t)n_rows, shift, (uint32_t)n_blocks, rt, tile_row, (uint32_t)n_tiles, cur, col2, (double *)val2);
        else
            hipLaunchKernelGGL(k_cf_scatter<float>, dim3(cf_rows_grid(n_rows)), dim3(kBlock), 0, s, off, col, (const float *)val,
                               (uint64_t)n_rows, shift, (uint32_t)n_blocks, rt, tile_row, (uint32_t)n_tiles, cur, col2, (float *)val2);
        SMH_HIP(hipGetLastError());
        SMH_HIP(hipStreamSynchronize(s));
        return SMH_OK;
    };
    const int rc = body();
    (void)hipFree(cur);
    (void)hipFree(d_over);
    if (rc != SMH_OK || !fits) {
        (void)hipFree(seg); (void)hipFree(cnt8); (void)hipFree(col2); (void)hipFree(val2); (void)hipFree(tile_row);
        return rc;
    }
    *n_tiles_out = (size_t)n_tiles;
    *tile_row_out = tile_row; *seg_out = seg; *cnt_out = cnt8; *col2_out = col2; *val2_out = val2;
    *fits_out = true;
    return SMH_OK;
}

template <typename T, int RT, int NIT>
static int cf_launch(const uint32_t *tile_row, size_t n_tiles, const uint32_t *seg, const uint8_t *cnt, const uint32_t *col, const T *val,
                     const T *x, T *y, size_t nnz, uint32_t n_blocks, int device, hipStream_t s) {
    // a round = the waves the chip holds at once (so that all of them walk the column blocks together)
    static int per_cu_cache[64] = {0};  // per device; (racing threads compute the same value)
    static int cus_cache[64] = {0};
    int per_cu = per_cu_cache[device & 63], cus = cus_cache[device & 63];
    if (per_cu == 0) {
        SMH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_spmv_colfused<T, RT, NIT>, kBlock, 0));
        if (per_cu < 1) per_cu = 1;
        hipDeviceProp_t prop;
        cus = hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        per_cu_cache[device & 63] = per_cu;
        cus_cache[device & 63] = cus;
    }
    static const int per_cu_env = getenv("SMH_COLFUSED_BLOCKS_PER_CU") ? atoi(getenv("SMH_COLFUSED_BLOCKS_PER_CU")) : 0;  // tuning knob
    if (per_cu_env >= 1 && per_cu_env <= per_cu) per_cu = per_cu_env;
    const uint64_t wpb = kBlock / kWave;
    uint64_t cap_blocks = ((uint64_t)per_cu * (uint64_t)cus) & ~uint64_t(7);
    if (cap_blocks < 8) cap_blocks = 8;
    const uint64_t nnz_readable = (nnz + 3) & ~uint64_t(3);
    for (uint64_t t = 0; t < n_tiles;) {
        uint64_t blocks = (n_tiles - t + wpb - 1) / wpb;
        // even rounds: spread what is left over the rounds still needed instead of full rounds and a small one
        const uint64_t rounds_left = (blocks + cap_blocks - 1) / cap_blocks;
        blocks = (blocks + rounds_left - 1) / rounds_left;
        blocks = (blocks + 7) & ~uint64_t(7);  // a multiple of 8: every XCD gets the same number of waves
        if (blocks > cap_blocks) blocks = cap_blocks;
        const uint64_t t_end = t + blocks * wpb < n_tiles ? t + blocks * wpb : n_tiles;
        hipLaunchKernelGGL((k_spmv_colfused<T, RT, NIT>), dim3((unsigned)blocks), dim3(kBlock), 0, s, tile_row, seg, cnt, col, val, x, y,
                           n_blocks, t, t_end, nnz_readable);
        SMH_HIP(hipGetLastError());
        t = t_end;
    }
    return SMH_OK;
}

int launch_spmv_colfused(int dtype, uint32_t rt, const uint32_t *tile_row, size_t n_tiles, const uint32_t *seg, const uint8_t *cnt,
                         const uint32_t *col, const void *val, const void *x, void *y, size_t n_rows, size_t nnz, uint32_t n_blocks,
                         int device, hipStream_t s) {
    if (n_rows == 0 || n_tiles == 0) return SMH_OK;
    if (dtype == SMH_F64) {
        if (rt == 16) return cf_launch<double, 16, 2>(tile_row, n_tiles, seg, cnt, col, (const double *)val, (const double *)x, (double *)y, nnz, n_blocks, device, s);
        if (rt == 8) return cf_launch<double, 8, 2>(tile_row, n_tiles, seg, cnt, col, (const double *)val, (const double *)x, (double *)y, nnz, n_blocks, device, s);
    } else {
        if (rt == 16) return cf_launch<float, 16, 4>(tile_row, n_tiles, seg, cnt, col, (const float *)val, (const float *)x, (float *)y, nnz, n_blocks, device, s);
        if (rt == 8) return cf_launch<float, 8, 4>(tile_row, n_tiles, seg, cnt, col, (const float *)val, (const float *)x, (float *)y, nnz, n_blocks, device, s);
    }
    return fail(SMH_ERR_INVALID, "K2f: rows per lane must be 8 or 16 (got %u)", rt);
}

}  // namespace smh
