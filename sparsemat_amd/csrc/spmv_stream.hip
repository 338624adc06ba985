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
//
// x STAGED IN LDS (K1s XS; its successor K1s XD lives in spmv_stream_xd.hip).  The gathers are what keeps the address path busy (7
// stencil arms interleaved over the lanes: ~14 cache lines per gather instruction).  An inspector (plan time, one wave per tile)
// describes the columns a tile references as up to 4 disjoint intervals (a stencil tile: the planes it touches; a span beyond 1024
// columns is cut at its widest gaps); the 16-bit column codes are relative to them, and when the intervals of every tile fit a
// 2048- or 4096-entry stage the kernel copies them to LDS with 16-byte loads issued BEFORE the tile's own chunk loads and gathers
// from LDS.  (Round 1's K1s-w staged x with 4-byte loads in a phase of its own and lost by 0.9 ms; it left the library in round 3.)
//
// ANY ROW LENGTH.  A tile with more entries than the LDS product stage holds (4096) is taken in several passes
// over the stage; a row that straddles passes carries its accumulator, so the adds stay in storage order:
// K1s is correct AND bit-exact for every matrix, the tables and the tile height only change speed.
#include "internal.hpp"

namespace smh {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t skew(uint32_t i) { return i + (i >> 5); }  // +1 word every 32: breaks 2^k strides

__device__ __forceinline__ float st_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ double st_mul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ float st_add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double st_add(double a, double b) { return __dadd_rn(a, b); }

// ---- inspector: up to 4 column intervals per tile -------------------------------------------------------
// win[8*t + 2k], win[8*t + 2k + 1] = [lo, hi) of interval k (sorted, disjoint; hi == lo: unused).  All zero:
// the tile has no window (gathers go to L2).  Limits: a tile qualifies when it holds at most max_entries entries,
// every interval is at most max_width columns wide and the intervals hold at most max_total columns together
// (the 16-bit column codes: 16384 per interval, nothing else; K1r's banded ring: a quarter of the ring per interval; a tile WITHOUT
// entries counts as described).
__global__ void __launch_bounds__(kBlock)
k_stream_windows(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, uint64_t n_rows, uint64_t n_tiles,
                 uint64_t tile_rows, uint64_t max_entries, uint64_t max_width, uint64_t max_total, uint64_t split_above,
                 bool count_empty, uint32_t *__restrict__ win, uint32_t *__restrict__ n_windowed) {
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    for (uint64_t t = wave; t < n_tiles; t += n_waves) {
        const uint64_t r0 = t * tile_rows, r1 = r0 + tile_rows < n_rows ? r0 + tile_rows : n_rows;
        const uint64_t k0 = off[r0], k1 = off[r1];
        uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
        bool ok = k1 > k0 && k1 - k0 <= max_entries;
        if (ok) {
            uint32_t cmin = 0xFFFFFFFFu, cmax = 0u;
            for (uint64_t k = k0 + lane; k < k1; k += kWave) {
                const uint32_t c = col[k];
                cmin = min(cmin, c);
                cmax = max(cmax, c);
            }
#pragma unroll
            for (int o = kWave / 2; o > 0; o >>= 1) {
                cmin = min(cmin, (uint32_t)__shfl_xor(cmin, o, kWave));
                cmax = max(cmax, (uint32_t)__shfl_xor(cmax, o, kWave));
            }
            const uint64_t span = (uint64_t)cmax - cmin + 1;
            // one interval when the columns' span allows it -- unless it is wider than split_above: then the clusters inside it are
            // looked for first (the column codes do not care, but what a tile stages of x in LDS is the intervals' total: a 64^3
            // Laplacian's tile spans 8448 columns and references 896 of them), and the span stays the fallback
            const bool single_ok = span <= max_width && span <= max_total;
            if (single_ok && span <= split_above) {
                lo[0] = cmin; hi[0] = cmax + 1;
            } else {
                // occupancy of 64 equal buckets over [cmin, cmax]; the 3 widest empty runs split the columns
                const uint64_t w = (span + 63) / 64;
                unsigned long long occ = 0ull;
                for (uint64_t k = k0 + lane; k < k1; k += kWave) occ |= 1ull << ((col[k] - cmin) / w);
#pragma unroll
                for (int o = kWave / 2; o > 0; o >>= 1) occ |= (unsigned long long)__shfl_xor((long long)occ, o, kWave);
                // every lane runs the same scalar scan (64 steps): top-3 gaps by length
                int gs[3] = {-1, -1, -1}, gl[3] = {0, 0, 0};
                int run_start = -1;
                for (int b = 0; b <= 64; ++b) {
                    const bool empty = b < 64 && !((occ >> b) & 1ull);
                    if (empty) { if (run_start < 0) run_start = b; continue; }
                    if (run_start >= 0) {
                        int s0 = run_start, l0 = b - run_start;
                        run_start = -1;
#pragma unroll
                        for (int g = 0; g < 3; ++g)
                            if (l0 > gl[g]) { const int ts = gs[g], tl = gl[g]; gs[g] = s0; gl[g] = l0; s0 = ts; l0 = tl; }
                    }
                }
                // bucket boundaries of the (up to) 4 segments, in ascending order
                int cut_s[3], cut_e[3], nc = 0;
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    if (gl[g] > 0) { cut_s[nc] = gs[g]; cut_e[nc] = gs[g] + gl[g]; ++nc; }
                for (int a = 0; a < nc; ++a)          // tiny insertion sort by start
                    for (int b2 = a + 1; b2 < nc; ++b2)
                        if (cut_s[b2] < cut_s[a]) { int ts = cut_s[a], te = cut_e[a]; cut_s[a] = cut_s[b2]; cut_e[a] = cut_e[b2]; cut_s[b2] = ts; cut_e[b2] = te; }
                int seg_b[4], seg_e[4];  // segments in bucket units [seg_b, seg_e)
                int nseg = 0, cur = 0;
                for (int a = 0; a < nc; ++a) { seg_b[nseg] = cur; seg_e[nseg] = cut_s[a]; ++nseg; cur = cut_e[a]; }
                seg_b[nseg] = cur; seg_e[nseg] = 64; ++nseg;
                // exact column bounds per segment
                uint32_t smin[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, smax[4] = {0, 0, 0, 0};
                for (uint64_t k = k0 + lane; k < k1; k += kWave) {
                    const uint32_t c = col[k];
                    const int b = (int)((c - cmin) / w);
#pragma unroll
                    for (int sgm = 0; sgm < 4; ++sgm)
                        if (sgm < nseg && b >= seg_b[sgm] && b < seg_e[sgm]) { smin[sgm] = min(smin[sgm], c); smax[sgm] = max(smax[sgm], c); }
                }
                uint64_t total = 0, widest = 0;
#pragma unroll
                for (int sgm = 0; sgm < 4; ++sgm) {
#pragma unroll
                    for (int o = kWave / 2; o > 0; o >>= 1) {
                        smin[sgm] = min(smin[sgm], (uint32_t)__shfl_xor(smin[sgm], o, kWave));
                        smax[sgm] = max(smax[sgm], (uint32_t)__shfl_xor(smax[sgm], o, kWave));
                    }
                    if (sgm < nseg && smin[sgm] <= smax[sgm]) {
                        lo[sgm] = smin[sgm]; hi[sgm] = smax[sgm] + 1;
                        const uint64_t wd = (uint64_t)hi[sgm] - lo[sgm];
                        total += wd;
                        widest = wd > widest ? wd : widest;
                    }
                }
                ok = total <= max_total && widest <= max_width;
                if (!ok && single_ok) {
                    ok = true;
                    lo[0] = cmin; hi[0] = cmax + 1;
                    lo[1] = lo[2] = lo[3] = hi[1] = hi[2] = hi[3] = 0;
                }
            }
        }
        if (lane < 8) {
            const int k = lane >> 1;
            // count_empty mode distinguishes "no entries" (all zero) from "not describable" (interval 0 = [1, 0))
            const uint32_t none = (count_empty && k1 > k0 && lane == 0) ? 1u : 0u;
            win[8 * t + lane] = ok ? ((lane & 1) ? hi[k] : lo[k]) : none;
        }
        if (lane == 0 && (ok || (count_empty && k1 == k0))) atomicAdd(n_windowed, 1u);  // integer count: exact
    }
}

// ---- kernel ----------------------------------------------------------------------------------------------
// RPT = rows per thread: a tile is RPT*256 consecutive rows (2 when every 512-row tile fits the LDS stage)
// DOT: also leave dot_partials[tile] = sum over the tile's rows of dot_lhs[row] * y[row]: with dot_lhs = x (square
// matrices) the p.Ap of a CG iteration falls out of the SpMV epilogue, with any lhs it is SparseMatrix::inner_prod
// (sparsematrix.rs:161-171; y == NULL then: nothing is stored).  Fixed order: bitwise reproducible.
// ACC: y += A x (the column-blocked variant K2c accumulates one column block per launch).
// MULTI: tiles may hold more entries than the LDS stage (pass loop).  MULTI = false is the host's promise that no
// tile does; the body is then loop-free and needs 70 instead of 90 VGPRs (f32: 7 instead of 5 waves per SIMD).
// A/B knob: waves per SIMD the register allocator must allow (0: no constraint).  Measured on the 512^3 Laplacian,
// same box: unconstrained (68 VGPRs, 7 waves) 1.79 / 1.82 ms; 8 waves (64 VGPRs + 12 B/lane of scratch) 1.96 / 1.95 ms.
// Two more experiments on that matrix, both dropped: (a) the tile's entry range from a compact, L2-resident boundary
// table instead of off[] (1.845 / 1.869 vs 1.881 / 1.879 ms: within noise); (b) a plane-interleaved tile order per
// XCD (tiles at one in-plane position of all planes back to back, so that the three uses of an x entry coincide):
// 1.95 ms vs 1.82 ms -- the concurrently active tiles then stream from 64 distant regions per array.  rocprofv3
// (profiles/r01_pmc_lap512_stream.json): L2 fetches 9.9 GB for 8.6 GB algorithmic reads, 87 % of wave cycles waiting.
// 16-bit column codes (C16 below) DO pay -- 1.57 ms against 1.70-1.86 ms on one box (profiles/r01_ab_stream_col16_codes.log)
// -- but only with the decode deferred to the gather phase: the first attempt decoded right after each chunk load,
// which put a wait between the chunk loads, and lost (1.88-1.93 ms against 1.63-1.69 ms,
// profiles/r01_ab_stream_col16_codes_dropped.log).
#ifndef SMH_STREAM_MIN_WAVES
#define SMH_STREAM_MIN_WAVES 0
#endif
#ifndef SMH_STREAM_NT_STORE  // y leaves with non-temporal stores: 1.543-1.552 vs 1.563-1.572 ms on the 512^3 Laplacian
#define SMH_STREAM_NT_STORE 1
#endif
#if SMH_STREAM_MIN_WAVES > 0
#define SMH_STREAM_BOUNDS __launch_bounds__(kBlock, SMH_STREAM_MIN_WAVES)
#else
#define SMH_STREAM_BOUNDS __launch_bounds__(kBlock)
#endif
// CAP: entries of the LDS product stage (a pass).  (A CAP = 2048 body for sparse tiles -- 44 instead of 68 VGPRs, 8
// waves per SIMD, half the LDS -- was measured on the 512^3 Laplacian: 1.84 / 1.79 ms against 1.82 / 1.66 ms: occupancy
// is not what limits this kernel either.)
// C16 (only when EVERY tile with entries has a code-table description): the kernel streams 16-bit column codes
// (`code`: interval << 14 | offset; intervals in `cwin`, 8 u32 per 256-row tile) instead of the u32 columns -- 2 of
// 8 bytes per f32 entry less from HBM.  The raw code words stay in registers through the load phase (decoding right
// after each load would put a wait between the chunk loads) and are turned into columns in the gather phase.
__device__ __forceinline__ uint32_t decode_col(uint32_t cd, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) {
    const uint32_t q = cd >> 14;
    return (q == 0u ? b0 : q == 1u ? b1 : q == 2u ? b2 : b3) + (cd & 16383u);
}

// L8 (with C16, when no row holds more than 255 entries): the row boundaries come from one BYTE per row (its length) and
// one u32 per tile (where its entries start) instead of the u32 offset_rows stream -- 1 instead of 4 bytes per row from HBM
// (512^3 Laplacian: 537 -> 136 MB of the 7.2 GB a product moves).  The in-tile prefix sum of the lengths is a wave scan
// whose cross-wave part rides on the barrier the kernel has anyway.
// (Round 3 tried a bank skew of the staged x -- entry i at i + 4 (i / 128 + i / 1024), so that the neighbours i - N and i + N of a
// row, which share a bank on grids with N a multiple of 32 and are read by ONE wavefront instruction, stop colliding: 28 % of this
// kernel's LDS cycles are conflicts.  It measured SLOWER, 1.34-1.36 against 1.29-1.31 ms on the 512^3 Laplacian
// (profiles/r03_ab_k1s_xs_skew_negative.log): the body is bound by vector-ALU issue, and the three extra instructions per entry
// cost more than the conflicts they remove.)
template <typename T, int RPT, bool DOT, bool ACC = false, bool MULTI = true, int CAP = kStreamCap, bool C16 = false, bool L8 = false,
          int XS = 0 /* 16-byte chunks of x per thread staged in LDS: 0 (none), 2 or 4 */>
__global__ void SMH_STREAM_BOUNDS
k_spmv_stream(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const T *__restrict__ val,
              const T *__restrict__ x, T *__restrict__ y, uint64_t n_rows, uint64_t nnz, uint64_t nnz_readable,
              uint64_t n_tiles, T *__restrict__ dot_partials,
              const uint16_t *__restrict__ code, const uint32_t *__restrict__ cwin, const uint8_t *__restrict__ len8,
              const uint32_t *__restrict__ tbase, const T *__restrict__ dot_lhs, uint64_t tile0) {
    static_assert(!C16 || RPT == 1, "the code table describes 256-row tiles");
    static_assert(!L8 || (C16 && !MULTI), "row lengths as bytes: single-pass 256-row tiles with column codes");
    static_assert(!XS || L8, "x staged in LDS: the coded single-pass body");
    // XS: the tile's column intervals of x (the code table's <= 4 intervals, <= 2048 or 4096 entries in 16-byte chunks) are
    // copied to LDS with 16-byte loads issued BEFORE the tile's chunk loads, and the gathers become LDS reads: two vector-memory
    // instructions per thread instead of eight, and no second, dependent trip to memory.
    constexpr int kXsCap = XS ? XS * kBlock * 4 : 4;  // entries of x the stage holds (2048 / 4096)
    __shared__ __attribute__((aligned(16))) T s_xs[kXsCap];
    __shared__ uint32_t s_wtot[L8 ? kBlock / kWave : 1];
    __shared__ T s_prod[CAP + CAP / 32 + 8];
    // bijective XCD-aware remap: XCD g (= blockIdx % 8) walks a contiguous run of tiles
    const uint64_t q = n_tiles >> 3, rm = n_tiles & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    // (n_tiles = the tiles of THIS launch, tile0 = the first of them: a launch may cover a run of the matrix's tiles only -- the
    // partitioned product multiplies a block's boundary rows before its interior ones, par.hip)
    const uint64_t tile = tile0 + (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + idx;
    constexpr uint64_t TILE_ROWS = (uint64_t)kStreamRows * RPT;
    const uint64_t r0 = tile * TILE_ROWS;
    const uint64_t r1 = r0 + TILE_ROWS < n_rows ? r0 + TILE_ROWS : n_rows;
    const uint32_t tid = threadIdx.x;
    uint32_t o0[RPT], o1[RPT];
    uint32_t k0, k1, my_len = 0, my_excl = 0;
    if constexpr (L8) {
        k0 = tbase[tile];      // tile-uniform: scalar loads
        k1 = tbase[tile + 1];
        my_len = len8[r0 + tid];  // (the byte array is padded to whole tiles with zeros)
        uint32_t incl = my_len;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o, kWave);
            if ((int)(tid & (kWave - 1)) >= o) incl += up;
        }
        if ((tid & (kWave - 1)) == kWave - 1) s_wtot[tid / kWave] = incl;  // read after the barrier below
        my_excl = incl - my_len;
        o0[0] = o1[0] = 0;
    } else {
#pragma unroll
        for (int rr = 0; rr < RPT; ++rr) {
            const uint64_t r = r0 + (uint64_t)rr * kBlock + tid;
            o0[rr] = off[r < r1 ? r : r1];  // (non-temporal loads here: no measurable difference)
            o1[rr] = off[r + 1 < r1 ? r + 1 : r1];
        }
        k0 = off[r0];  // tile-uniform: scalar loads
        k1 = off[r1];
    }
    T sum[RPT];
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) sum[rr] = T(0);
    // DOT: this thread's lhs entries, requested NOW so that their latency hides behind the whole tile (fetched in the
    // epilogue they sat on the critical path of every block: 1.69 against 1.58 ms on the 512^3 Laplacian)
    T dl[DOT ? RPT : 1];
    if constexpr (DOT) {
#pragma unroll
        for (int rr = 0; rr < RPT; ++rr) {
            const uint64_t r = r0 + (uint64_t)rr * kBlock + tid;
            dl[rr] = r < r1 ? dot_lhs[r] : T(0);
        }
    }
    // The tile's entries [k0, k1) are taken in passes of <= kStreamCap entries (stencil/FEM tiles: one pass).  Per
    // pass: a dense run of aligned chunks (the arrays' last partial chunk is read entry by entry when the arrays
    // are not padded; nnz_readable = nnz rounded up for padded arrays).  A pass holds at most NIT chunks per
    // thread: ALL of a thread's chunk loads are issued before the first gather (they sit in the memory queue
    // together), then all gathers.  A row that straddles passes keeps its accumulator: the order of the adds is
    // the storage order whatever the number of passes.
    constexpr int NIT = (CAP + 3 + 4 * kBlock - 1) / (4 * kBlock);  // chunks per thread that cover CAP entries from an aligned start
    uint32_t cb0 = 0, cb1 = 0, cb2 = 0, cb3 = 0;  // C16: starts of the tile's column intervals (tile-uniform)
    if constexpr (C16) {
        const uint32_t *w = cwin + 8 * tile;  // scalar loads
        cb0 = w[0]; cb1 = w[2]; cb2 = w[4]; cb3 = w[6];
    }
    uint32_t sb0 = 0, sb1 = 0, sb2 = 0, sb3 = 0;  // XS: where column cb_q sits in s_xs
    T xr[XS ? XS : 1][4];
    uint32_t xs_tot = 0;
    if constexpr (XS) {
        const uint32_t *w = cwin + 8 * tile;  // scalar loads (the starts are cb0..cb3 already)
        const uint32_t e0 = w[1], e1 = w[3], e2 = w[5], e3 = w[7];
        const uint32_t al0 = cb0 & ~3u, al1 = cb1 & ~3u, al2 = cb2 & ~3u, al3 = cb3 & ~3u;
        const uint32_t n0 = e0 > cb0 ? (e0 - al0 + 3u) >> 2 : 0u, n1 = e1 > cb1 ? (e1 - al1 + 3u) >> 2 : 0u;
        const uint32_t n2 = e2 > cb2 ? (e2 - al2 + 3u) >> 2 : 0u, n3 = e3 > cb3 ? (e3 - al3 + 3u) >> 2 : 0u;
        const uint32_t p1 = n0, p2 = p1 + n1, p3 = p2 + n2;
        xs_tot = p3 + n3;  // (<= XS * kBlock chunks and inside x: checked by the host, smh_crs::stream_xs_*)
        sb0 = cb0 & 3u; sb1 = 4u * p1 + (cb1 & 3u); sb2 = 4u * p2 + (cb2 & 3u); sb3 = 4u * p3 + (cb3 & 3u);
#pragma unroll
        for (int u = 0; u < XS; ++u) {
            const uint32_t j = tid + (uint32_t)u * kBlock;
            xr[u][0] = xr[u][1] = xr[u][2] = xr[u][3] = T(0);
            if (j < xs_tot) {
                const uint32_t q = (uint32_t)(j >= p1) + (uint32_t)(j >= p2) + (uint32_t)(j >= p3);
                const uint32_t pq = q == 0u ? 0u : q == 1u ? p1 : q == 2u ? p2 : p3;
                const uint32_t al = q == 0u ? al0 : q == 1u ? al1 : q == 2u ? al2 : al3;
                const T *g = x + ((uint64_t)al + 4u * (j - pq));
                if constexpr (sizeof(T) == 4) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(g);
                    xr[u][0] = a.x; xr[u][1] = a.y; xr[u][2] = a.z; xr[u][3] = a.w;
                } else {
                    const f64x2 a = *reinterpret_cast<const f64x2 *>(g), b = *reinterpret_cast<const f64x2 *>(g + 2);
                    xr[u][0] = a.x; xr[u][1] = a.y; xr[u][2] = b.x; xr[u][3] = b.y;
                }
            }
        }
    }
    uint32_t ps = k0;
    do {
        const uint32_t pe = MULTI && k1 - ps > (uint32_t)CAP ? ps + (uint32_t)CAP : k1;
        // everything per lane is a 32-bit position relative to the pass's aligned start `pa` (tile-uniform)
        const uint64_t pa = (uint64_t)(ps & ~3u);
        const uint32_t *__restrict__ colp = col + pa;
        const T *__restrict__ valp = val + pa;
        const uint32_t lo = ps & 3u, hi = pe - (uint32_t)pa;
        const uint64_t rd64 = nnz_readable - pa, nn64 = nnz - pa;
        const uint32_t rd = rd64 > 0x10000u ? 0x10000u : (uint32_t)rd64, nn = nn64 > 0x10000u ? 0x10000u : (uint32_t)nn64;
        uint32_t c[NIT][4] = {};  // (initialised: no values carried around the pass loop)
        T v[NIT][4] = {};
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const uint32_t j = 4u * tid + (uint32_t)it * (4u * kBlock);
            if (j < hi) {
                if (j + 4 <= rd) {
                    if constexpr (C16) {  // two packed words, untouched until the gather phase
                        const u32x2 cw = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(code + pa + j));
                        c[it][0] = cw.x; c[it][1] = cw.y;
                    } else {
                        const u32x4 cc = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(colp + j));
                        c[it][0] = cc.x; c[it][1] = cc.y; c[it][2] = cc.z; c[it][3] = cc.w;
                    }
                    if constexpr (sizeof(T) == 4) {
                        const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(valp + j));
                        v[it][0] = a.x; v[it][1] = a.y; v[it][2] = a.z; v[it][3] = a.w;
                    } else {
                        const f64x2 a = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(valp + j));
                        const f64x2 b = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(valp + j + 2));
                        v[it][0] = a.x; v[it][1] = a.y; v[it][2] = b.x; v[it][3] = b.y;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool in = j + e < nn;
                        if constexpr (C16) {
                            const uint32_t cd = in ? (uint32_t)code[pa + j + e] : 0u;
                            c[it][e >> 1] = (e & 1) ? (c[it][e >> 1] | (cd << 16)) : cd;
                        } else {
                            c[it][e] = in ? colp[j + e] : 0u;
                        }
                        v[it][e] = in ? valp[j + e] : T(0);
                    }
                }
            }
        }
        if constexpr (XS) {  // (the x chunks were requested before the tile's own: they are here first)
#pragma unroll
            for (int u = 0; u < XS; ++u) {
                const uint32_t j = tid + (uint32_t)u * kBlock;
                if (j < xs_tot) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) s_xs[4u * j + e] = xr[u][e];
                }
            }
            __syncthreads();
        }
        if constexpr (XS) {
            // branch-free: every slot reads some x from the LDS stage (codes of slots outside the tile decode to a clamped index)
            // and writes its product -- to its place in the stage, or to a dump slot past the stage's end.  (The guarded form
            // below costs a compare pair, an exec save / restore and a branch per slot; this body is VALU-bound.)
            constexpr uint32_t kDump = (uint32_t)(CAP + CAP / 32 + 7);
            const uint32_t span = hi - lo;
            // the four interval offsets (< 2^12 each) in one 64-bit word: one shift selects where a compare / select chain of six did
            const uint64_t sbp = (uint64_t)sb0 | ((uint64_t)sb1 << 16) | ((uint64_t)sb2 << 32) | ((uint64_t)sb3 << 48);
            static_assert((kXsCap & (kXsCap - 1)) == 0, "the stage index is clamped with a mask");
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const uint32_t j = 4u * tid + (uint32_t)it * (4u * kBlock);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t cd = (e & 1) ? (c[it][e >> 1] >> 16) : (c[it][e >> 1] & 0xFFFFu);
                    const uint32_t idx = ((uint32_t)(sbp >> ((cd >> 14) << 4)) & 0xFFFFu) + (cd & 16383u);
                    const T xe = s_xs[idx & (uint32_t)(kXsCap - 1)];  // (slots outside the tile carry other tiles' codes)
                    const uint32_t rel = j + (uint32_t)e - lo;  // (wraps below lo: fails the test too)
                    s_prod[rel < span ? skew(rel) : kDump] = st_mul(xe, v[it][e]);
                }
            }
        }
        T xv[NIT][4];
#pragma unroll
        for (int it = 0; !XS && it < NIT; ++it) {
            const uint32_t j = 4u * tid + (uint32_t)it * (4u * kBlock);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t i = j + e;
                xv[it][e] = T(0);
                if (i >= lo && i < hi) {
                    if constexpr (C16) {
                        const uint32_t cd = (e & 1) ? (c[it][e >> 1] >> 16) : (c[it][e >> 1] & 0xFFFFu);
                        if constexpr (XS) xv[it][e] = s_xs[decode_col(cd, sb0, sb1, sb2, sb3)];
                        else xv[it][e] = x[decode_col(cd, cb0, cb1, cb2, cb3)];
                    } else {
                        xv[it][e] = x[c[it][e]];
                    }
                }
            }
        }
#pragma unroll
        for (int it = 0; !XS && it < NIT; ++it) {
            const uint32_t j = 4u * tid + (uint32_t)it * (4u * kBlock);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t i = j + e;
                if (i >= lo && i < hi) s_prod[skew(i - lo)] = st_mul(xv[it][e], v[it][e]);
            }
        }
        __syncthreads();
        if constexpr (L8) {  // this row's entries: the tile's start + the lengths of the rows before it
            uint32_t base = k0;
            for (uint32_t w = 0; w < tid / kWave; ++w) base += s_wtot[w];
            o0[0] = base + my_excl;
            o1[0] = o0[0] + my_len;
        }
        // each row: storage order, one rounded add per entry (reference: sum += product)
#pragma unroll
        for (int rr = 0; rr < RPT; ++rr) {
            uint32_t i = min(max(o0[rr], ps), pe) - ps;
            const uint32_t iend = min(max(o1[rr], ps), pe) - ps;
            T acc = sum[rr];
            for (; i + 4 <= iend; i += 4) {
                const T p0 = s_prod[skew(i)], p1 = s_prod[skew(i + 1)], p2 = s_prod[skew(i + 2)], p3 = s_prod[skew(i + 3)];
                acc = st_add(acc, p0);
                acc = st_add(acc, p1);
                acc = st_add(acc, p2);
                acc = st_add(acc, p3);
            }
            for (; i < iend; ++i) acc = st_add(acc, s_prod[skew(i)]);
            sum[rr] = acc;
        }
        ps = pe;
        if (MULTI && ps < k1) __syncthreads();  // the next pass overwrites the stage
    } while (MULTI && ps < k1);
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const uint64_t r = r0 + (uint64_t)rr * kBlock + tid;
#if SMH_STREAM_NT_STORE  // A/B: y leaves with a non-temporal store (it is not read again by this kernel)
        if (r < r1 && (!DOT || y)) {  // (DOT with y == NULL: only lhs . (A x) is wanted -- SparseMatrix::inner_prod)
            if constexpr (ACC) y[r] = st_add(y[r], sum[rr]);
            else __builtin_nontemporal_store(sum[rr], &y[r]);
        }
#else
        if (r < r1 && (!DOT || y)) y[r] = ACC ? st_add(y[r], sum[rr]) : sum[rr];
#endif
    }
    if constexpr (DOT) {
        __shared__ T s_red[kBlock / kWave];
        T d = T(0);
#pragma unroll
        for (int rr = 0; rr < RPT; ++rr) {
            const uint64_t r = r0 + (uint64_t)rr * kBlock + tid;
            if (r < r1) d += dl[rr] * sum[rr];
        }
        d = wave_sum_to_lane63(d);  // (lanes by the DPP scan network, then the wavefronts in index order; K1s XD adds in the same order)
        if ((tid & (kWave - 1)) == kWave - 1) s_red[tid / kWave] = d;
        __syncthreads();
        if (tid == 0) {
            T t = T(0);
#pragma unroll
            for (int w = 0; w < kBlock / kWave; ++w) t += s_red[w];
            dot_partials[tile] = t;
        }
    }
}

// single_pass: the caller knows that no tile holds more than kStreamCap entries (statistic taken at create time)
template <typename T>
static int launch_stream_t(const uint32_t *off, const uint32_t *col, const T *val, const T *x, T *y, size_t n_rows,
                           size_t nnz, bool padded, int rpt, bool single_pass, T *dot_partials,
                           const uint16_t *code, const uint32_t *cwin, const uint8_t *len8, const uint32_t *tbase, const T *dot_lhs,
                           hipStream_t s, bool small_tiles, int xs, uint64_t tile_begin, uint64_t tile_end) {
    const uint64_t readable = padded ? ((nnz + 3) & ~uint64_t(3)) : nnz;
    const uint64_t all_tiles = stream_tiles(n_rows, rpt);
    // tiles [tile_begin, tile_end) of the matrix (default: all of them)
    const uint64_t tile0 = tile_begin < all_tiles ? tile_begin : all_tiles, tile1 = tile_end < all_tiles ? tile_end : all_tiles;
    if (tile1 <= tile0) return SMH_OK;
    const uint64_t n_tiles = tile1 - tile0;
    const dim3 grid((unsigned)n_tiles), block(kBlock);
    // experiment knob: unused dynamic LDS per block, to cap the blocks a CU holds at once (0: whatever fits).  Unlike the
    // element-wise kernels (fewer, fatter grids stream faster) this kernel wants every block it can get: 512^3 Laplacian,
    // one box, 1.54 ms unconstrained, 1.64 / 2.10 / 2.62 / 3.64 ms with 8 / 16 / 24 / 36 KiB of padding
    // (profiles/r01_k1s_blocks_per_cu.log)
    static const unsigned lds_pad = getenv("SMH_STREAM_LDS_PAD") ? (unsigned)atoi(getenv("SMH_STREAM_LDS_PAD")) : 0u;
#define SMH_ST_LAUNCH(R, D, M, C)                                                                                        \
    hipLaunchKernelGGL((k_spmv_stream<T, R, D, false, M, kStreamCap, C>), grid, block, lds_pad, s, off, col, val, x, y, \
                       (uint64_t)n_rows, (uint64_t)nnz, readable, n_tiles, dot_partials, code, cwin,                   \
                       (const uint8_t *)nullptr, (const uint32_t *)nullptr, dot_lhs, tile0)
#define SMH_ST_PICK(R, C)                                                         \
    do {                                                                          \
        if (single_pass) {                                                        \
            if (dot_partials) SMH_ST_LAUNCH(R, true, false, C); else SMH_ST_LAUNCH(R, false, false, C); \
        } else {                                                                  \
            if (dot_partials) SMH_ST_LAUNCH(R, true, true, C); else SMH_ST_LAUNCH(R, false, true, C);   \
        }                                                                         \
    } while (0)
    if (rpt == 2) SMH_ST_PICK(2, false);
#define SMH_ST_XS(D, P)                                                                                                                     \
    hipLaunchKernelGGL((k_spmv_stream<T, 1, D, false, false, kStreamCapSmall, true, true, P>), grid, block, lds_pad, s, off, col, val, x, y, \
                       (uint64_t)n_rows, (uint64_t)nnz, readable, n_tiles, dot_partials, code, cwin, len8, tbase, dot_lhs, tile0)
    else if (code && cwin && len8 && tbase && single_pass && small_tiles && xs == 2) {  // ... and x staged in LDS (2048 entries)
        if (dot_partials) SMH_ST_XS(true, 2); else SMH_ST_XS(false, 2);
    }
    else if (code && cwin && len8 && tbase && single_pass && small_tiles && xs == 4) {  // ... (4096 entries: wider grids)
        if (dot_partials) SMH_ST_XS(true, 4); else SMH_ST_XS(false, 4);
    }
#undef SMH_ST_XS
    else if (code && cwin && len8 && tbase && single_pass && small_tiles) {
        // ... and no tile beyond kStreamCapSmall entries (stencils: 7 x 256 = 1792): two chunk slots per thread instead of five.
        // 32-34 VGPRs instead of 67-70 (f32), 46-50 instead of ~106 (f64: 4 -> 8 waves per SIMD).  Measured, one box: 400^3 f64
        // 1.412 -> 1.330 ms; 512^3 f32 unchanged (1.503 / 1.515 ms) -- neither registers nor instruction count bound the f32
        // kernel: rocprofv3 shows the texture addresser busy 70 % of the time (8 gather instructions per thread and tile)
        if (dot_partials)
            hipLaunchKernelGGL((k_spmv_stream<T, 1, true, false, false, kStreamCapSmall, true, true>), grid, block, lds_pad, s, off, col, val,
                               x, y, (uint64_t)n_rows, (uint64_t)nnz, readable, n_tiles, dot_partials, code, cwin, len8, tbase, dot_lhs, tile0);
        else
            hipLaunchKernelGGL((k_spmv_stream<T, 1, false, false, false, kStreamCapSmall, true, true>), grid, block, lds_pad, s, off, col, val,
                               x, y, (uint64_t)n_rows, (uint64_t)nnz, readable, n_tiles, dot_partials, code, cwin, len8, tbase, dot_lhs, tile0);
    }
    else if (code && cwin && len8 && tbase && single_pass) {  // column codes + byte row lengths
        if (dot_partials)
            hipLaunchKernelGGL((k_spmv_stream<T, 1, true, false, false, kStreamCap, true, true>), grid, block, lds_pad, s, off, col, val,
                               x, y, (uint64_t)n_rows, (uint64_t)nnz, readable, n_tiles, dot_partials, code, cwin, len8, tbase, dot_lhs, tile0);
        else
            hipLaunchKernelGGL((k_spmv_stream<T, 1, false, false, false, kStreamCap, true, true>), grid, block, lds_pad, s, off, col, val,
                               x, y, (uint64_t)n_rows, (uint64_t)nnz, readable, n_tiles, dot_partials, code, cwin, len8, tbase, dot_lhs, tile0);
    }
    else if (code && cwin) SMH_ST_PICK(1, true);  // 16-bit column codes (every tile described)
    else SMH_ST_PICK(1, false);
#undef SMH_ST_PICK
#undef SMH_ST_LAUNCH
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

// K2c building block: one column block of the column-blocked copy, y (+)= A_b x.  `off` holds ABSOLUTE entry
// positions into col/val (the blocks share one pair of arrays, padded to a multiple of 4 entries).
template <typename T>
static int launch_stream_block_t(const uint32_t *off, const uint32_t *col, const T *val, const T *x, T *y, size_t n_rows,
                                 size_t nnz_total, int rpt, bool single_pass, bool acc, hipStream_t s) {
    const size_t rows = (size_t)kStreamRows * rpt;
    const uint64_t n_tiles = (n_rows + rows - 1) / rows;
    const uint64_t readable = (nnz_total + 3) & ~uint64_t(3);
    const dim3 grid((unsigned)n_tiles), block(kBlock);
#define SMH_SB_LAUNCH(R, A, M)                                                                                          \
    hipLaunchKernelGGL((k_spmv_stream<T, R, false, A, M>), grid, block, 0, s, off, col, val, x, y, (uint64_t)n_rows, \
                       (uint64_t)nnz_total, readable, n_tiles, (T *)nullptr,         \
                       (const uint16_t *)nullptr, (const uint32_t *)nullptr, (const uint8_t *)nullptr, (const uint32_t *)nullptr,  \
                       (const T *)nullptr, (uint64_t)0)
#define SMH_SB_PICK(R)                                                                      \
    do {                                                                                     \
        if (single_pass) { if (acc) SMH_SB_LAUNCH(R, true, false); else SMH_SB_LAUNCH(R, false, false); } \
        else             { if (acc) SMH_SB_LAUNCH(R, true, true);  else SMH_SB_LAUNCH(R, false, true);  } \
    } while (0)
    switch (rpt) {
        case 1: SMH_SB_PICK(1); break;
        case 2: SMH_SB_PICK(2); break;
        case 4: SMH_SB_PICK(4); break;
        case 8: SMH_SB_PICK(8); break;
        default: return fail(SMH_ERR_INVALID, "K2c: rows per thread must be 1, 2, 4 or 8");
    }
#undef SMH_SB_PICK
#undef SMH_SB_LAUNCH
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

int launch_spmv_stream_block(int dtype, const uint32_t *off, const uint32_t *col, const void *val, const void *x, void *y,
                             size_t n_rows, size_t nnz_total, int rpt, bool single_pass, bool acc, hipStream_t s) {
    if (n_rows == 0) return SMH_OK;
    if (dtype == SMH_F64)
        return launch_stream_block_t<double>(off, col, (const double *)val, (const double *)x, (double *)y, n_rows, nnz_total,
                                             rpt, single_pass, acc, s);
    return launch_stream_block_t<float>(off, col, (const float *)val, (const float *)x, (float *)y, n_rows, nnz_total, rpt,
                                        single_pass, acc, s);
}

// tiles (= blocks = dot partials) of a K1s launch
size_t stream_tiles(size_t n_rows, int rpt) {
    const size_t rows = (size_t)kStreamRows * (rpt == 2 ? 2 : 1);
    return (n_rows + rows - 1) / rows;
}

int launch_spmv_stream(int dtype, const uint32_t *off, const uint32_t *col, const void *val, const void *x, void *y,
                       size_t n_rows, size_t nnz, bool padded, int rpt, bool single_pass,
                       void *dot_partials, const uint16_t *code, const uint32_t *cwin, const uint8_t *len8, const uint32_t *tbase,
                       const void *dot_lhs, hipStream_t s, bool small_tiles, int xs, uint64_t tile_begin, uint64_t tile_end) {
    if (n_rows == 0) return SMH_OK;
    if (dot_partials && !dot_lhs) dot_lhs = x;  // CG's p.Ap
    if (!dot_partials && !y) return fail(SMH_ERR_INVALID, "K1s: no output");
    if (dtype == SMH_F64)
        return launch_stream_t<double>(off, col, (const double *)val, (const double *)x, (double *)y, n_rows, nnz, padded,
                                       rpt, single_pass, (double *)dot_partials, code, cwin, len8, tbase, (const double *)dot_lhs, s, small_tiles, xs, tile_begin, tile_end);
    return launch_stream_t<float>(off, col, (const float *)val, (const float *)x, (float *)y, n_rows, nnz, padded, rpt,
                                  single_pass, (float *)dot_partials, code, cwin, len8, tbase, (const float *)dot_lhs, s, small_tiles, xs, tile_begin, tile_end);
}

// row lengths as bytes (rows padded to whole 256-row tiles with zeros) and the tiles' first entries (n_tiles + 1 values);
// the caller guarantees max_row_len <= 255
__global__ void __launch_bounds__(kBlock)
k_stream_len8(const uint32_t *__restrict__ off, uint64_t n_rows, uint64_t n_padded, uint64_t n_tiles, uint8_t *__restrict__ len8,
              uint32_t *__restrict__ tbase) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t r = tid; r < n_padded; r += nthreads) len8[r] = r < n_rows ? (uint8_t)(off[r + 1] - off[r]) : (uint8_t)0;
    for (uint64_t t = tid; t <= n_tiles; t += nthreads) {
        const uint64_t r = t * kStreamRows;
        tbase[t] = off[r < n_rows ? r : n_rows];
    }
}

int launch_stream_len8(const uint32_t *off, size_t n_rows, uint8_t *len8, uint32_t *tbase, hipStream_t s) {
    const uint64_t n_tiles = (n_rows + kStreamRows - 1) / kStreamRows;
    if (n_tiles == 0) return SMH_OK;
    const uint64_t n_padded = n_tiles * kStreamRows;
    uint64_t blocks = (n_padded + kBlock - 1) / kBlock;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_stream_len8, dim3((unsigned)blocks), dim3(kBlock), 0, s, off, (uint64_t)n_rows, n_padded, n_tiles, len8, tbase);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

// the intervals the 16-bit column codes are relative to: 256-row tiles of any size, <= 16384 columns per interval, tiles without
// entries count as described; a span beyond 1024 columns is split where it has gaps (what K1s XS / XD stage of x is the
// intervals' total)
int launch_stream_windows(const uint32_t *off, const uint32_t *col, size_t n_rows, uint32_t *win, uint32_t *d_count, hipStream_t s) {
    const uint64_t n_tiles = (n_rows + kStreamRows - 1) / kStreamRows;
    SMH_HIP(hipMemsetAsync(d_count, 0, sizeof(uint32_t), s));
    if (n_tiles == 0) return SMH_OK;
    uint64_t blocks = (n_tiles * kWave + kBlock - 1) / kBlock;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_stream_windows, dim3((unsigned)blocks), dim3(kBlock), 0, s, off, col, (uint64_t)n_rows, n_tiles,
                       (uint64_t)kStreamRows, ~uint64_t(0), (uint64_t)kStreamCodeWidth, ~uint64_t(0), (uint64_t)1024, true, win, d_count);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

// the same inspector over tiles of `tile_rows` rows with intervals of at most max_width columns each (K1r's banded
// ring: 64-row tiles, a quarter of the ring per interval); tiles without entries count as described
int launch_tile_intervals(const uint32_t *off, const uint32_t *col, size_t n_rows, size_t tile_rows, size_t max_width,
                          uint32_t *win, uint32_t *d_count, hipStream_t s) {
    const uint64_t n_tiles = (n_rows + tile_rows - 1) / tile_rows;
    SMH_HIP(hipMemsetAsync(d_count, 0, sizeof(uint32_t), s));
    if (n_tiles == 0) return SMH_OK;
    uint64_t blocks = (n_tiles * kWave + kBlock - 1) / kBlock;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_stream_windows, dim3((unsigned)blocks), dim3(kBlock), 0, s, off, col, (uint64_t)n_rows, n_tiles,
                       (uint64_t)tile_rows, ~uint64_t(0), (uint64_t)max_width, ~uint64_t(0), (uint64_t)max_width, true, win, d_count);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

// 16-bit column codes: code = interval << 14 | (column - interval start), intervals from the tile's table entry
__global__ void __launch_bounds__(kBlock)
k_stream_codes(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const uint32_t *__restrict__ win,
               uint64_t n_rows, uint64_t n_tiles, uint16_t *__restrict__ code) {
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint32_t *w = win + 8 * t;
        const uint32_t a0 = w[0], a1 = w[2], e1 = w[3], a2 = w[4], e2 = w[5], a3 = w[6], e3 = w[7];
        const uint64_t r0 = t * kStreamRows, r1 = r0 + kStreamRows < n_rows ? r0 + kStreamRows : n_rows;
        const uint64_t k0 = off[r0], k1 = off[r1];
        for (uint64_t k = k0 + threadIdx.x; k < k1; k += kBlock) {
            const uint32_t c = col[k];
            // the used intervals are a prefix of the table, sorted, disjoint, and contain every column of the tile
            const uint32_t q = (uint32_t)(e1 > a1 && c >= a1) + (uint32_t)(e2 > a2 && c >= a2) + (uint32_t)(e3 > a3 && c >= a3);
            const uint32_t base = q == 3u ? a3 : q == 2u ? a2 : q == 1u ? a1 : a0;
            code[k] = (uint16_t)((q << 14) | (c - base));
        }
    }
}

int launch_stream_codes(const uint32_t *off, const uint32_t *col, const uint32_t *win, size_t n_rows, uint16_t *code,
                        hipStream_t s) {
    const uint64_t n_tiles = (n_rows + kStreamRows - 1) / kStreamRows;
    if (n_tiles == 0) return SMH_OK;
    const uint64_t blocks = n_tiles < 16384 ? n_tiles : 16384;
    hipLaunchKernelGGL(k_stream_codes, dim3((unsigned)blocks), dim3(kBlock), 0, s, off, col, win, (uint64_t)n_rows, n_tiles, code);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

// XS eligibility: out[0] = most 16-byte chunks any tile's column intervals need (from aligned starts), out[1] = the largest
// x index + 1 those chunks touch
__global__ void __launch_bounds__(kBlock)
k_stream_xs_stats(const uint32_t *__restrict__ win, uint64_t n_tiles, uint32_t *__restrict__ out) {
    uint32_t mc = 0, me = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_tiles; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t *w = win + 8 * t;
        uint32_t chunks = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t a = w[2 * q], e = w[2 * q + 1];
            if (e > a) {
                const uint32_t al = a & ~3u, n = (e - al + 3u) >> 2;
                chunks += n;
                me = max(me, al + 4u * n);
            }
        }
        mc = max(mc, chunks);
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        mc = max(mc, (uint32_t)__shfl_down(mc, o, kWave));
        me = max(me, (uint32_t)__shfl_down(me, o, kWave));
    }
    if ((threadIdx.x & (kWave - 1)) == 0) { atomicMax(&out[0], mc); atomicMax(&out[1], me); }
}

int launch_stream_xs_stats(const uint32_t *win, size_t n_tiles, uint32_t *d_out2, hipStream_t s) {
    SMH_HIP(hipMemsetAsync(d_out2, 0, 2 * sizeof(uint32_t), s));
    if (n_tiles == 0) return SMH_OK;
    uint64_t blocks = (n_tiles + kBlock - 1) / kBlock;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_stream_xs_stats, dim3((unsigned)blocks), dim3(kBlock), 0, s, win, (uint64_t)n_tiles, d_out2);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

// largest number of entries in any 256-row tile (decides whether AUTO may use K1s)
__global__ void __launch_bounds__(kBlock)
k_stream_max_tile(const uint32_t *__restrict__ off, uint64_t n_rows, uint64_t n_tiles, uint64_t tile_rows,
                  uint32_t *__restrict__ out) {
    uint32_t m = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_tiles; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r0 = t * tile_rows, r1 = r0 + tile_rows < n_rows ? r0 + tile_rows : n_rows;
        const uint32_t e = off[r1] - off[r0];
        m = e > m ? e : m;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_down(m, o, kWave));
    if ((threadIdx.x & (kWave - 1)) == 0) atomicMax(out, m);  // integer max: exact, order independent
}

int launch_stream_max_tile(const uint32_t *off, size_t n_rows, size_t tile_rows, uint32_t *d_out, hipStream_t s) {
    SMH_HIP(hipMemsetAsync(d_out, 0, sizeof(uint32_t), s));
    if (n_rows == 0) return SMH_OK;
    const uint64_t n_tiles = (n_rows + tile_rows - 1) / tile_rows;
    uint64_t blocks = (n_tiles + kBlock - 1) / kBlock;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_stream_max_tile, dim3((unsigned)blocks), dim3(kBlock), 0, s, off, (uint64_t)n_rows, n_tiles,
                       (uint64_t)tile_rows, d_out);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

}  // namespace smh
