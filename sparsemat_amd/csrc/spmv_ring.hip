// spmv_ring.hip -- K1r inspector and phase plan (the kernel itself is in spmv_ring2.hip).
//
// K1's limit on banded matrices is not HBM but the vector L1 / address path: every x[col] gather of a wave
// instruction touches its own cache line (measured: 27 % of HBM peak on the stratified band vs 63 % when
// the gathers coalesce).  K1r moves the gathers to LDS:
//
//   * blocks own CONTIGUOUS row ranges, and blockIdx -> range is XCD-aware (blocks b, b+8, .. share an
//     XCD and get neighbouring ranges);
//   * an inspector (create time, one wave per 64-row tile, one read of columns[]) records the column span of
//     every tile; the host turns that into a per-block PLAN of phases {rows, x-range to load, ring or not}:
//     the block keeps a sliding window of x in an LDS ring (index = column mod ring size: 16384 entries --
//     64 KiB of f32, 128 KiB of f64) and each phase only loads the part of its window that the ring does
//     not hold yet; a run of tiles whose span does not fit the ring is one phase with global gathers
//     (uniform branch), so the kernel is correct for ANY matrix -- the plan only changes speed.
#include <algorithm>
#include <vector>

#include "internal.hpp"

namespace smh {

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
// ring (ring_entries columns); what the ring already holds from the previous phase is not reloaded.
// noring_mode: gather mode of phases that cannot use the ring (0 cached, 2 L1-bypassing global gathers).
void build_ring_plan(size_t n_rows, size_t ring_entries, const uint32_t *cmin, const uint32_t *cmax, size_t n_blocks,
                     uint32_t noring_mode, std::vector<uint32_t> &phase_ptr, std::vector<RingPhase> &phases,
                     double *ring_row_fraction) {
    const uint64_t ring = ring_entries;
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
                RingPhase p{(uint32_t)(t * kTileRows), (uint32_t)std::min<uint64_t>((uint64_t)e * kTileRows, n_rows), 0, 0,
                            noring_mode};
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

// ---- banded ring: 4 bands of ring_entries / 4 columns each ----------------------------------------------------
// For matrices whose rows reference a few narrow column intervals far apart (stencils on structured grids: the plane
// below, the own plane, the plane above): `win` holds up to 4 sorted, disjoint intervals per 64-row tile (8 u32 per
// tile, all zero: tile without entries, interval 0 == [1, 0): not describable).  Band k of the ring keeps a sliding
// window over the tiles' k-th intervals, slot = k * S + (column mod S), S = ring_entries / 4; the bands slide
// independently, with the same greedy rule as the single ring per band.
void build_ring_plan_banded(size_t n_rows, size_t ring_entries, const uint32_t *win, size_t n_blocks, uint32_t noring_mode,
                            std::vector<uint32_t> &phase_ptr, std::vector<RingPhase> &phases, double *ring_row_fraction) {
    const uint64_t S = ring_entries / 4;
    const size_t n_tiles = (n_rows + kTileRows - 1) / kTileRows;
    const size_t tiles_per_block = (n_tiles + n_blocks - 1) / n_blocks;
    phase_ptr.assign(n_blocks + 1, 0);
    phases.clear();
    uint64_t ring_rows = 0;
    auto lo_of = [&](size_t t, int k) { return (uint64_t)win[8 * t + 2 * k]; };
    auto hi_of = [&](size_t t, int k) { return (uint64_t)win[8 * t + 2 * k + 1]; };
    auto undescribed = [&](size_t t) { return lo_of(t, 0) > hi_of(t, 0); };
    for (size_t b = 0; b < n_blocks; ++b) {
        phase_ptr[b] = (uint32_t)phases.size();
        size_t t = std::min(b * tiles_per_block, n_tiles);
        const size_t t_end = std::min(t + tiles_per_block, n_tiles);
        uint64_t wlo[4] = {0, 0, 0, 0}, whi[4] = {0, 0, 0, 0};  // band k of the ring holds columns [wlo, whi)
        while (t < t_end) {
            size_t e = t + 1;
            if (undescribed(t)) {  // one global-gather phase over the run of such tiles
                while (e < t_end && undescribed(e)) ++e;
                RingPhase p{(uint32_t)(t * kTileRows), (uint32_t)std::min<uint64_t>((uint64_t)e * kTileRows, n_rows), 0, 0,
                            noring_mode};
                phases.push_back(p);
                t = e;
                continue;
            }
            uint64_t umin[4], umax[4];  // union of interval k over the phase's tiles (umin > umax: none yet)
            for (int k = 0; k < 4; ++k) {
                const bool used = hi_of(t, k) > lo_of(t, k);
                umin[k] = used ? lo_of(t, k) : 1;
                umax[k] = used ? hi_of(t, k) - 1 : 0;
            }
            while (e < t_end && !undescribed(e)) {
                uint64_t nmin[4], nmax[4];
                bool fits = true;
                for (int k = 0; k < 4; ++k) {
                    nmin[k] = umin[k]; nmax[k] = umax[k];
                    if (hi_of(e, k) > lo_of(e, k)) {
                        const bool any = umin[k] <= umax[k];
                        nmin[k] = any ? std::min<uint64_t>(umin[k], lo_of(e, k)) : lo_of(e, k);
                        nmax[k] = any ? std::max<uint64_t>(umax[k], hi_of(e, k) - 1) : hi_of(e, k) - 1;
                        if (nmax[k] + 1 - nmin[k] > S) fits = false;
                    }
                }
                if (!fits) break;
                for (int k = 0; k < 4; ++k) { umin[k] = nmin[k]; umax[k] = nmax[k]; }
                ++e;
            }
            RingPhase p{(uint32_t)(t * kTileRows), (uint32_t)std::min<uint64_t>((uint64_t)e * kTileRows, n_rows), 0, 0, 1};
            for (int k = 0; k < 4; ++k) {
                if (umin[k] > umax[k]) continue;
                uint32_t l_lo = 0, l_hi = 0;
                if (umin[k] >= wlo[k] && umin[k] < whi[k]) {  // continues the band's window: load only the new part
                    if (umax[k] + 1 > whi[k]) { l_lo = (uint32_t)whi[k]; l_hi = (uint32_t)(umax[k] + 1); whi[k] = umax[k] + 1; }
                    wlo[k] = std::max<uint64_t>(wlo[k], whi[k] > S ? whi[k] - S : 0);
                } else {                                      // disjoint (or behind): restart the band
                    l_lo = (uint32_t)umin[k]; l_hi = (uint32_t)(umax[k] + 1);
                    wlo[k] = umin[k]; whi[k] = umax[k] + 1;
                }
                if (k == 0) { p.load_lo = l_lo; p.load_hi = l_hi; }
                else { p.band_lo[k - 1] = l_lo; p.band_hi[k - 1] = l_hi; }
            }
            ring_rows += p.row_end - p.row_begin;
            phases.push_back(p);
            t = e;
        }
    }
    phase_ptr[n_blocks] = (uint32_t)phases.size();
    if (ring_row_fraction) *ring_row_fraction = n_rows ? (double)ring_rows / (double)n_rows : 0.0;
}

// 16-bit ring slots of the banded ring: code = k * S + (column mod S), k = the tile interval holding the column
// (tiles that are not describable keep zeros: their phases gather from global memory with the u32 columns)
__global__ void __launch_bounds__(kBlock)
k_ring_band_codes(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const uint32_t *__restrict__ win,
                  uint64_t n_rows, uint64_t n_tiles, uint32_t S, uint16_t *__restrict__ code) {
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / kWave;
    for (uint64_t t = wave; t < n_tiles; t += n_waves) {
        const uint32_t *w = win + 8 * t;
        const uint32_t a0 = w[0], e0 = w[1], a1 = w[2], e1 = w[3], a2 = w[4], e2 = w[5], a3 = w[6], e3 = w[7];
        if (e0 <= a0) continue;  // no entries, or not describable
        const uint64_t r0 = t * kTileRows, r1 = r0 + kTileRows < n_rows ? r0 + kTileRows : n_rows;
        const uint64_t k0 = off[r0], k1 = off[r1];
        for (uint64_t k = k0 + lane; k < k1; k += kWave) {
            const uint32_t c = col[k];
            const uint32_t q = (uint32_t)(e1 > a1 && c >= a1) + (uint32_t)(e2 > a2 && c >= a2) + (uint32_t)(e3 > a3 && c >= a3);
            code[k] = (uint16_t)(q * S + (c & (S - 1u)));
        }
    }
}

int launch_ring_band_codes(const uint32_t *off, const uint32_t *col, const uint32_t *win, size_t n_rows, uint32_t S,
                           uint16_t *code, hipStream_t s) {
    const uint64_t n_tiles = (n_rows + kTileRows - 1) / kTileRows;
    if (n_tiles == 0) return SMH_OK;
    uint64_t blocks = (n_tiles * kWave + kBlock - 1) / kBlock;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_ring_band_codes, dim3((unsigned)blocks), dim3(kBlock), 0, s, off, col, win, (uint64_t)n_rows, n_tiles, S, code);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

}  // namespace smh
