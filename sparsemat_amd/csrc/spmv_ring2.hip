// spmv_ring2.hip -- K1r, software-pipelined body (same plan, same result as spmv_ring.hip).
//
// Profile of the first K1r body (profiles/r01_pmc_sq_ring_v1.json): waves are parked on memory 81 % of
// their cycles and every wave alternates "issue 8 KiB of loads -> wait -> reduce", so the bytes in
// flight per CU sag while a wave computes.  This body keeps every wave's memory queue primed:
//
//   unit  = SB steps x (64/LANES) rows of one wave (LANES 8: 4 steps x 8 rows = 32 rows, 8 KiB of
//           columns+values)
//   per iteration, in program order (vmcnt retires in order, so older loads must be the ones needed
//   first):   offsets(unit+2)  ->  chunk loads(unit+1)  ->  consume(unit)
//   i.e. row offsets are fetched two units ahead (one coalesced load per unit, fanned out to the lane
//   groups with ds_bpermute), the 16-B column/value chunks one unit ahead, and the LDS gathers / FMAs /
//   butterfly / transposed coalesced store of the current unit run under the next unit's HBM latency.
//   Steady state is straight-line code (no branch between issue and consume), so hipcc's counted
//   s_waitcnt vmcnt(N) leaves exactly the next unit's loads in flight.
//   All chunk loads are unconditional and branch-free: addresses are clamped to the last whole chunk
//   that is safe to read, entries outside the row are masked with selects (never multiplied in: LDS
//   garbage may be NaN), and the <= 3 entries of an unpadded array's final partial chunk are added by
//   a one-thread fix-up kernel (in storage order: they are the last entries of their rows).
#include "internal.hpp"

#include <atomic>

namespace smh {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// The ring holds kRingEntries columns for either value type (the plan does not depend on T):
//   f32: 64 KiB  -> 512-thread blocks, two resident per CU;
//   f64: 128 KiB -> 1024-thread blocks, one resident per CU (same 16 waves per CU, <= 128 VGPRs).
// LDS is dynamic (above the 64 KiB static limit for f64).
// RING = columns the ring holds: 16384 by default; f32 matrices whose rows do not fit that but fit 32768 take the
// 128 KiB / 1024-thread configuration too (chosen by the plan builder's caller).
#ifndef SMH_RING2_THREADS64
#define SMH_RING2_THREADS64 1024
#endif
template <typename T, int RING> struct Ring2Cfg {
    static constexpr bool kBig = (size_t)RING * sizeof(T) > 65536;  // 128 KiB of LDS: one workgroup per CU
    static constexpr int kThreads = kBig ? (sizeof(T) == 8 ? SMH_RING2_THREADS64 : 1024) : 512;
    // wavefronts per SIMD the register budget is sized for: the CU's resident workgroups' wavefronts over its 4 SIMDs
    static constexpr int kWavesPerSimd = (kBig ? 1 : 2) * (kThreads / 64) / 4;
};
#ifndef SMH_RING2_SB
#define SMH_RING2_SB 2
#endif
constexpr int kRing2SB = SMH_RING2_SB;
#ifndef SMH_RING2_SB64
#define SMH_RING2_SB64 1
#endif
// steps per unit: two units x SB x 32 B per lane live in VGPRs
// Addressing A/B (same box, interleaved runs): 32-bit byte offsets (saddr form, would also need a < 2^30
// entries-per-phase guard) gave 0.465 vs 0.465 ms on C2 and 2.89 vs 2.76 ms on the 512^3 Laplacian against
// 64-bit offsets -- no gain, so the guard-free 64-bit form is the default.
#ifndef SMH_RING2_SADDR
#define SMH_RING2_SADDR 0
#endif
#if SMH_RING2_SADDR
#define SMH_R2_OFF(rel, size) ((size_t)((rel) * (size)))
#else
#define SMH_R2_OFF(rel, size) ((size_t)(rel) * (size))
#endif

template <typename T, int NCH>
struct Unit {
    uint32_t o0, o1;  // lane L: off[base+L], off[base+L+1] (rows clamped to the phase end)
    uint32_t c[NCH][4];
    T v[NCH][4];
};

// WHICH entries of a pass (4 LANES consecutive entry slots of a row, starting on the 4-entry grid) lane j of the row's group takes.
//   f32: slots 4j .. 4j+3 -- 16 bytes of values per lane, one load instruction covers 16 LANES contiguous bytes per row.
//   f64: slots 2j, 2j+1 and 2 LANES + 2j, 2 LANES + 2j + 1 -- TWO 16-byte loads per lane, each of which covers 16 LANES contiguous bytes
//        per row (LANES 8: a whole 128-byte line).  Round 3 gave an f64 lane 32 contiguous bytes: each of its two load instructions
//        then touched every line of the row but used half of it, and the kernel ran at 4.9 TB/s where the f32 one reaches 6.2; with
//        whole lines per instruction 0.67-0.70 -> 0.58-0.60 ms on the headline shape (profiles/r04_k1r_f64_line_loads.log).
template <typename T, int LANES>
__device__ __forceinline__ uint32_t lane_pos(uint32_t j, int q) {
    if constexpr (sizeof(T) == 8) return q < 2 ? 2u * j + (uint32_t)q : 2u * LANES + 2u * j + (uint32_t)(q - 2);
    else return 4u * j + (uint32_t)q;
}

// colp/valp are wave-uniform (SGPR) base pointers of the phase's first chunk, pass_rel the 32-bit element offset of the pass from
// it: hipcc emits the saddr form `global_load_dwordx4 v, v_off, s[base]`, one address VGPR per load.  Every address is clamped to the
// last piece that is safe to read (last_rel: the last whole 4-entry chunk of the arrays): lanes without work re-read a valid piece,
// which the caller masks.
// C16: colp points into the 16-bit column array (the low halves of the columns: all a ring phase needs, since the
// ring slot of a column is `column mod 16384`) -- 4 columns are then 8 bytes (f32: one load; f64: two of 4 bytes)
template <typename T, bool C16, int LANES>
__device__ __forceinline__ void load_chunk_nb(const void *__restrict__ colp, const T *__restrict__ valp, uint32_t pass_rel, uint32_t j,
                                              uint32_t last_rel, uint32_t (&c)[4], T (&v)[4]) {
    if constexpr (sizeof(T) == 4) {
        uint32_t rel = pass_rel + 4u * j;
        rel = rel < last_rel ? rel : last_rel;
        if constexpr (C16) {  // kept packed: c[0], c[1] hold two columns each (unpacked where they are used, see ring_slot)
            const u32x2 cc = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(
                reinterpret_cast<const char *>(colp) + SMH_R2_OFF(rel, 2u)));
            c[0] = cc.x; c[1] = cc.y;  // (unpacking here instead measured the same: 0.343-0.345 ms either way)
        } else {
            const u32x4 cc = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(
                reinterpret_cast<const char *>(colp) + SMH_R2_OFF(rel, 4u)));
            c[0] = cc.x; c[1] = cc.y; c[2] = cc.z; c[3] = cc.w;
        }
        const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(
            reinterpret_cast<const char *>(valp) + SMH_R2_OFF(rel, 4u)));
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    } else {
        // two pieces of 2 entries (even positions; the last readable one is last_rel + 2)
        uint32_t e0 = pass_rel + 2u * j, e1 = pass_rel + 2u * LANES + 2u * j;
        e0 = e0 < last_rel + 2u ? e0 : last_rel + 2u;
        e1 = e1 < last_rel + 2u ? e1 : last_rel + 2u;
        if constexpr (C16) {
            c[0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(colp) + SMH_R2_OFF(e0, 2u)));
            c[1] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(colp) + SMH_R2_OFF(e1, 2u)));
        } else {
            const u32x2 c0 = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(reinterpret_cast<const char *>(colp) + SMH_R2_OFF(e0, 4u)));
            const u32x2 c1 = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(reinterpret_cast<const char *>(colp) + SMH_R2_OFF(e1, 4u)));
            c[0] = c0.x; c[1] = c0.y; c[2] = c1.x; c[3] = c1.y;
        }
        const f64x2 a = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(reinterpret_cast<const char *>(valp) + SMH_R2_OFF(e0, 8u)));
        const f64x2 b = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(reinterpret_cast<const char *>(valp) + SMH_R2_OFF(e1, 8u)));
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    }
}

// ring slot of entry q of a chunk: column mod kRingEntries, from the u32 columns or from the packed 16-bit pairs
template <bool C16, int RING>
__device__ __forceinline__ uint32_t ring_slot(const uint32_t (&c)[4], int q) {
    constexpr uint32_t MASK = RING - 1;
    if constexpr (C16) return (q & 1) ? ((c[q >> 1] >> 16) & MASK) : (c[q >> 1] & MASK);
    else return c[q] & MASK;
}

__device__ __forceinline__ float r2_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double r2_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename T, int NCH>
__device__ __forceinline__ void load_offsets(Unit<T, NCH> &u, const uint32_t *__restrict__ off, uint64_t base,
                                             uint64_t row_end, uint32_t lane) {
    uint64_t r0 = base + lane, r1 = base + lane + 1;
    r0 = r0 < row_end ? r0 : row_end;
    r1 = r1 < row_end ? r1 : row_end;
    u.o0 = off[r0];
    u.o1 = off[r1];
}

// row bounds of (step t, this lane's group), limited to the entries the kernel may touch
template <int LANES>
__device__ __forceinline__ void row_bounds(uint32_t o0, uint32_t o1, int t, uint32_t lane, uint32_t nnz_lim,
                                           uint32_t &s, uint32_t &e) {
    if constexpr (LANES == 1) {  // one lane per row: the lane already holds its own row's offsets
        s = o0;
        e = o1;
    } else {
        constexpr int RPS = kWave / LANES;
        const int src = t * RPS + (int)(lane / LANES);
        s = (uint32_t)__shfl((int)o0, src, kWave);
        e = (uint32_t)__shfl((int)o1, src, kWave);
    }
    s = s < nnz_lim ? s : nnz_lim;
    e = e < nnz_lim ? e : nnz_lim;
}

// A lane group covers 4*LANES*CH entry slots of its row per pass: chunk (ch, j) = slots [4*(ch*LANES+j), +4)
template <typename T, int LANES, int CH, int SB, bool C16>
__device__ __forceinline__ void issue_unit(Unit<T, SB * CH> &u, const void *__restrict__ colp,
                                           const T *__restrict__ valp, uint32_t kb, uint32_t nnz_lim, uint32_t last_rel,
                                           uint32_t lane) {
    const uint32_t j = lane % LANES;
#pragma unroll
    for (int t = 0; t < SB; ++t) {
        uint32_t s, e;
        row_bounds<LANES>(u.o0, u.o1, t, lane, nnz_lim, s, e);
        s = s > kb ? s : kb;  // (only rows past the arrays' readable end are below kb: they are empty)
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            // (lanes without work re-read a valid piece: clamped inside, masked when consumed)
            load_chunk_nb<T, C16, LANES>(colp, valp, (s & ~3u) - kb + 4u * (uint32_t)(ch * LANES), j, last_rel, u.c[t * CH + ch], u.v[t * CH + ch]);
        }
    }
}

// DOT (SparseMatrix::inner_prod, sparsematrix.rs:161-171): `y` then holds lhs, nothing is stored, and every lane adds
// lhs[row] * (A x)[row] of the rows it would have stored to its *dacc
template <typename T, int LANES, int CH, int SB, int GM, bool C16, int RING, bool DOT = false>
__device__ __forceinline__ void consume_unit(const Unit<T, SB * CH> &u, uint64_t base, uint64_t row_end,
                                             const void *__restrict__ colp, const T *__restrict__ valp,
                                             const T *__restrict__ x, const T *ring, T *__restrict__ y, uint32_t kb,
                                             uint32_t nnz_lim, uint32_t last_rel, uint32_t lane, T *dacc = nullptr) {
    constexpr int RPS = kWave / LANES;
    const uint32_t j = lane % LANES;
    T out = T(0);
#pragma unroll
    for (int t = 0; t < SB; ++t) {
        uint32_t s, e;
        row_bounds<LANES>(u.o0, u.o1, t, lane, nnz_lim, s, e);
        s = s > kb ? s : kb;
        e = e > s ? e : s;
        const uint32_t sa = s & ~3u;      // chunk grid is anchored at element 0
        const uint32_t lo = s - sa;       // 0..3: entries of the first chunk that belong to the previous row
        const uint32_t len = e - sa;      // row end relative to the aligned start
        T sum = T(0);
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t rel = 4u * (uint32_t)(ch * LANES) + lane_pos<T, LANES>(j, q);
                const bool in = rel >= lo && rel < len;
                T xv;
                // (round 4, a build that is wrong on purpose: every lane gathering its OWN slot -- no bank conflicts -- ran exactly as
                // fast, 0.699 against 0.698 ms on f64 and 0.353 against 0.353 on f32: the random LDS gathers are not what bounds K1r)
                if constexpr (GM == 1) xv = ring[ring_slot<C16, RING>(u.c[t * CH + ch], q)];
                else if constexpr (GM == 2) xv = __builtin_nontemporal_load(&x[in ? u.c[t * CH + ch][q] : 0u]);
                else xv = x[in ? u.c[t * CH + ch][q] : 0u];
                const T f = r2_fma(u.v[t * CH + ch][q], xv, sum);
                sum = in ? f : sum;
            }
        }
        // rows longer than one pass of the lane group (rare; not pipelined)
        for (uint32_t pass = 4u * (uint32_t)(CH * LANES); pass + lane_pos<T, LANES>(j, 0) < len; pass += 4u * LANES) {
            uint32_t cc[4];
            T vv[4];
            load_chunk_nb<T, C16, LANES>(colp, valp, sa - kb + pass, j, last_rel, cc, vv);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool in = pass + lane_pos<T, LANES>(j, q) < len;
                T xv;
                if constexpr (GM == 1) xv = ring[ring_slot<C16, RING>(cc, q)];
                else if constexpr (GM == 2) xv = __builtin_nontemporal_load(&x[in ? cc[q] : 0u]);
                else xv = x[in ? cc[q] : 0u];
                const T f = r2_fma(vv[q], xv, sum);
                sum = in ? f : sum;
            }
        }
        if constexpr (LANES == 1) {
            out = sum;
        } else {
#pragma unroll
            for (int o = LANES / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, kWave);
            // transpose: lane L keeps the sum of the unit's row L (held by every lane of group L % RPS in step L / RPS)
            const T got = __shfl(sum, (int)((lane % RPS) * LANES), kWave);
            out = ((int)(lane / RPS) == t) ? got : out;
        }
    }
    const uint64_t row = base + lane;
    if constexpr (DOT) {
        if (lane < (uint32_t)(SB * RPS) && row < row_end) *dacc = r2_fma(y[row], out, *dacc);
    } else {
        if (lane < (uint32_t)(SB * RPS) && row < row_end) y[row] = out;
    }
}

// col: the 32-bit column array, or (C16) the 16-bit one
template <typename T, int LANES, int CH, int GM, bool C16, int RING, bool DOT = false>
__device__ __forceinline__ void phase_rows(const uint32_t *__restrict__ off, const void *__restrict__ col,
                                           const T *__restrict__ val, const T *__restrict__ x, const T *ring,
                                           T *__restrict__ y, uint64_t rb, uint64_t re, uint32_t nnz_lim,
                                           uint64_t last_chunk, uint32_t wave, uint32_t lane, T *dacc = nullptr) {
    constexpr int STEPS = LANES;
    constexpr int SBMAX = sizeof(T) == 8 ? SMH_RING2_SB64 : kRing2SB;  // f64 chunks take 12 VGPRs: one step per unit (two: A/B in round 4)
    constexpr int SB = STEPS < SBMAX ? STEPS : SBMAX;
    constexpr int RU = SB * (kWave / LANES);                 // rows per unit
    constexpr uint64_t STRIDE = (uint64_t)(Ring2Cfg<T, RING>::kThreads / kWave) * RU;  // rows between two units of a wave
    uint64_t base = rb + (uint64_t)wave * RU;
    if (base >= re) return;
    // 32-bit addressing inside the phase: everything is relative to the phase's first (aligned) entry
    uint32_t kb = __builtin_amdgcn_readfirstlane(off[rb]) & ~3u;
    kb = kb < (uint32_t)last_chunk ? kb : (uint32_t)last_chunk;
    const uint32_t last_rel = (uint32_t)last_chunk - kb;
    const void *colp = reinterpret_cast<const char *>(col) + (size_t)kb * (C16 ? 2u : 4u);
    const T *valp = val + kb;
    Unit<T, SB * CH> A, B, N;  // N: only its offsets are used (the unit after next)
    load_offsets(A, off, base, re, lane);
    load_offsets(B, off, base + STRIDE, re, lane);
    issue_unit<T, LANES, CH, SB, C16>(A, colp, valp, kb, nnz_lim, last_rel, lane);
    for (;;) {
        if (base + STRIDE >= re) {
            consume_unit<T, LANES, CH, SB, GM, C16, RING, DOT>(A, base, re, colp, valp, x, ring, y, kb, nnz_lim, last_rel, lane, dacc);
            break;
        }
        // program order = age order: offsets(+2) older than chunks(+1); both stay in flight under consume
        load_offsets(N, off, base + 2 * STRIDE, re, lane);
        issue_unit<T, LANES, CH, SB, C16>(B, colp, valp, kb, nnz_lim, last_rel, lane);
        consume_unit<T, LANES, CH, SB, GM, C16, RING, DOT>(A, base, re, colp, valp, x, ring, y, kb, nnz_lim, last_rel, lane, dacc);
        A.o0 = N.o0; A.o1 = N.o1;
        base += STRIDE;
        if (base + STRIDE >= re) {
            consume_unit<T, LANES, CH, SB, GM, C16, RING, DOT>(B, base, re, colp, valp, x, ring, y, kb, nnz_lim, last_rel, lane, dacc);
            break;
        }
        load_offsets(N, off, base + 2 * STRIDE, re, lane);
        issue_unit<T, LANES, CH, SB, C16>(A, colp, valp, kb, nnz_lim, last_rel, lane);
        consume_unit<T, LANES, CH, SB, GM, C16, RING, DOT>(B, base, re, colp, valp, x, ring, y, kb, nnz_lim, last_rel, lane, dacc);
        B.o0 = N.o0; B.o1 = N.o1;
        base += STRIDE;
    }
}

// C16: ring phases stream the 16-bit column array `col16` (6 instead of 8 bytes per f32 entry); phases with global
// gathers need whole columns and keep reading `col`
// DOT: y holds lhs (read only) and dot_partials[blockIdx.x] = this block's share of lhs . (A x); nothing else is stored
template <typename T, int LANES, int CH, bool C16, int RING, bool DOT = false>
__global__ void __launch_bounds__((Ring2Cfg<T, RING>::kThreads), (Ring2Cfg<T, RING>::kWavesPerSimd))
k_spmv_ring2(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const uint16_t *__restrict__ col16,
             const T *__restrict__ val, const T *__restrict__ x, T *__restrict__ y, uint32_t nnz_lim, uint64_t last_chunk,
             const uint32_t *__restrict__ phase_ptr, const RingPhase *__restrict__ phases, uint32_t bands,
             T *__restrict__ dot_partials, uint32_t lb0, uint32_t lb_n) {
    T dacc_v = T(0);
    T *dacc = DOT ? &dacc_v : nullptr;
    extern __shared__ __attribute__((aligned(16))) unsigned char ring_raw[];  // RING * sizeof(T), dynamic
    T *ring = reinterpret_cast<T *>(ring_raw);
    // bands == 1: one window, slot = column mod RING.  bands == 4 (banded plan, C16 only): band k owns slots
    // [k * S, (k + 1) * S), S = RING / 4, slot = k * S + column mod S -- which is what col16 then holds
    const uint32_t MASK = (bands == 4u ? (uint32_t)RING / 4u : (uint32_t)RING) - 1u;
    constexpr int kRing2Threads = Ring2Cfg<T, RING>::kThreads;
    const uint32_t per_xcd = gridDim.x >> 3;
    // XCD-aware: neighbours share an L2.  A launch covers the plan's row ranges [lb0, lb0 + lb_n) (all of them, or -- the
    // partitioned product, par.hip -- a block's boundary / interior ranges); the grid is lb_n rounded up to a multiple of 8
    const uint32_t lb_rel = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (lb_rel >= lb_n) return;  // (whole workgroup, before any barrier)
    const uint32_t lb = lb0 + lb_rel;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t p0 = phase_ptr[lb], p1 = phase_ptr[lb + 1];
    for (uint32_t p = p0; p < p1; ++p) {
        const RingPhase ph = phases[p];
        const bool more = bands == 4u && (ph.band_hi[0] > ph.band_lo[0] || ph.band_hi[1] > ph.band_lo[1] || ph.band_hi[2] > ph.band_lo[2]);
        if (ph.load_hi > ph.load_lo || more) {
            __syncthreads();  // the previous phase's gathers are done before its slots are overwritten
            for (uint64_t cidx = (uint64_t)ph.load_lo + threadIdx.x; cidx < ph.load_hi; cidx += kRing2Threads)
                ring[cidx & MASK] = x[cidx];
            if (more) {
#pragma unroll
                for (uint32_t k = 0; k < 3; ++k)
                    for (uint64_t cidx = (uint64_t)ph.band_lo[k] + threadIdx.x; cidx < ph.band_hi[k]; cidx += kRing2Threads)
                        ring[(k + 1u) * (MASK + 1u) + (cidx & MASK)] = x[cidx];
            }
            __syncthreads();
        }
        // gather mode of the phase: 1 = LDS ring, 0 = L1/L2-cached global gathers, 2 = L1-bypassing (nt) global
        // gathers for phases whose columns have no locality to keep in the 32 KiB L1
        if (ph.use_ring == 1)
            phase_rows<T, LANES, CH, 1, C16, RING, DOT>(off, C16 ? (const void *)col16 : (const void *)col, val, x, ring, y, ph.row_begin,
                                                        ph.row_end, nnz_lim, last_chunk, wave, lane, dacc);
        else if (ph.use_ring == 2)
            phase_rows<T, LANES, CH, 2, false, RING, DOT>(off, col, val, x, ring, y, ph.row_begin, ph.row_end, nnz_lim, last_chunk, wave,
                                                          lane, dacc);
        else
            phase_rows<T, LANES, CH, 0, false, RING, DOT>(off, col, val, x, ring, y, ph.row_begin, ph.row_end, nnz_lim, last_chunk, wave,
                                                          lane, dacc);
    }
    if constexpr (DOT) {  // fixed order: lanes (butterfly), waves (index order) -- bitwise reproducible
        __shared__ T s_dot[kRing2Threads / kWave];
        T d = dacc_v;
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) d += __shfl_down(d, o, kWave);
        if (lane == 0) s_dot[wave] = d;
        __syncthreads();
        if (threadIdx.x == 0) {
            T t = T(0);
            for (int w = 0; w < kRing2Threads / kWave; ++w) t += s_dot[w];
            dot_partials[blockIdx.x] = t;
        }
    }
}

// the <= 3 entries of an unpadded array's last partial chunk, appended to their rows in storage order
template <typename T>
__global__ void k_ring2_tail(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col,
                             const T *__restrict__ val, const T *__restrict__ x, T *__restrict__ y, uint64_t n_rows,
                             uint64_t k_begin, uint64_t nnz) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint64_t r = n_rows - 1;
    while (r > 0 && (uint64_t)off[r] > k_begin) --r;  // row holding entry k_begin
    for (uint64_t k = k_begin; k < nnz; ++k) {
        while ((uint64_t)off[r + 1] <= k) ++r;
        y[r] = r2_fma(val[k], x[col[k]], y[r]);
    }
}

// ... and for the DOT form: *out = sum over those entries of lhs[row] * val * x[col] (their share of lhs . (A x))
template <typename T>
__global__ void k_ring2_tail_dot(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const T *__restrict__ val,
                                 const T *__restrict__ x, const T *__restrict__ lhs, T *__restrict__ out, uint64_t n_rows, uint64_t k_begin,
                                 uint64_t nnz) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint64_t r = n_rows - 1;
    while (r > 0 && (uint64_t)off[r] > k_begin) --r;
    T acc = T(0);
    for (uint64_t k = k_begin; k < nnz; ++k) {
        while ((uint64_t)off[r + 1] <= k) ++r;
        acc = r2_fma(lhs[r], val[k] * x[col[k]], acc);
    }
    *out = acc;
}

// col16[k] = low half of col[k] (k < nnz), zero padding up to the next multiple of 4 entries and one chunk beyond
__global__ void __launch_bounds__(kBlock)
k_narrow_columns(const uint32_t *__restrict__ col, uint64_t nnz, uint64_t n_out, uint16_t *__restrict__ col16) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_out; k += (uint64_t)gridDim.x * blockDim.x)
        col16[k] = k < nnz ? (uint16_t)col[k] : (uint16_t)0;
}

int launch_narrow_columns(const uint32_t *col, size_t nnz, uint16_t *col16, size_t n_out, hipStream_t s) {
    if (n_out == 0) return SMH_OK;
    uint64_t blocks = (n_out + kBlock - 1) / kBlock;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_narrow_columns, dim3((unsigned)blocks), dim3(kBlock), 0, s, col, (uint64_t)nnz, (uint64_t)n_out, col16);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

// dot_partials != NULL: the DOT form -- `y` is lhs (read only), dot_partials[0..n_blocks] receive the blocks' shares of
// lhs . (A x) plus, in slot n_blocks, that of an unpadded array's last partial chunk
template <typename T, int RING>
static int launch_ring2_t(int lanes, int chunks, const uint32_t *off, const uint32_t *col, const uint16_t *col16, const T *val,
                          const T *x, T *y, size_t n_rows, size_t nnz, bool padded, unsigned n_blocks,
                          const uint32_t *phase_ptr, const RingPhase *phases, uint32_t bands, T *dot_partials, hipStream_t s,
                          unsigned block_begin, unsigned block_end) {
    if (bands == 4u && !col16) return fail(SMH_ERR_INVALID, "banded ring plan without the 16-bit slot array");
    // the plan's row ranges [lb0, lb1) (default: all of them)
    const unsigned lb0 = block_begin < n_blocks ? block_begin : n_blocks, lb1 = block_end < n_blocks ? block_end : n_blocks;
    const bool whole = lb0 == 0 && lb1 == n_blocks;
    if (!whole && dot_partials) return fail(SMH_ERR_INVALID, "ring kernel: the DOT form takes the whole plan");
    if (lb1 <= lb0) return SMH_OK;
    // entries the streaming kernel may touch: everything when the arrays are padded to a multiple of 4,
    // else only whole chunks (the rest goes to k_ring2_tail)
    const uint64_t nnz_lim = padded ? nnz : (nnz & ~uint64_t(3));
    if (dot_partials) SMH_HIP(hipMemsetAsync(dot_partials, 0, ((size_t)n_blocks + 1) * sizeof(T), s));
    if (nnz_lim == 0) {
        if (!dot_partials && lb1 == n_blocks) SMH_HIP(hipMemsetAsync(y, 0, n_rows * sizeof(T), s));  // (<= 3 entries: the tail kernel below writes them)
    } else {
        const uint64_t last_chunk = (nnz_lim - 1) & ~uint64_t(3);
        dim3 grid(((lb1 - lb0) + 7u) & ~7u), block(Ring2Cfg<T, RING>::kThreads);
        constexpr size_t lds_bytes = (size_t)RING * sizeof(T);
        // dynamic LDS above 64 KiB must be allowed per kernel (idempotent, cheap)
#define SMH_R2_LAUNCH2(L, C, N, D)                                                                                       \
    do {                                                                                                                 \
        /* once per (instantiation, DEVICE): function attributes are per device, and smh_par_* places blocks on       \
           several devices.  Bit d of the mask = device d has the opt-in; a second thread racing on the same bit only   \
           repeats an idempotent call.  (Set by the first launch, so never inside a later stream capture.) */          \
        static std::atomic<uint64_t> attr_mask{0};                                                                       \
        const uint64_t dev_bit = 1ull << (unsigned)(current_device() & 63);                                              \
        if (!(attr_mask.load(std::memory_order_acquire) & dev_bit)) {                                                    \
            SMH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spmv_ring2<T, L, C, N, RING, D>),               \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));                    \
            attr_mask.fetch_or(dev_bit, std::memory_order_release);                                                      \
        }                                                                                                                \
        hipLaunchKernelGGL((k_spmv_ring2<T, L, C, N, RING, D>), grid, block, lds_bytes, s, off, col, col16, val, x, y,   \
                           (uint32_t)nnz_lim, last_chunk, phase_ptr, phases, bands, dot_partials, (uint32_t)lb0,         \
                           (uint32_t)(lb1 - lb0));                                                                       \
    } while (0)
#define SMH_R2_LAUNCH1(L, C, N)                                                            \
    do {                                                                                   \
        if (dot_partials) SMH_R2_LAUNCH2(L, C, N, true); else SMH_R2_LAUNCH2(L, C, N, false); \
    } while (0)
#define SMH_R2_LAUNCH(L, C)                                            \
    do {                                                               \
        if (col16) SMH_R2_LAUNCH1(L, C, true); else SMH_R2_LAUNCH1(L, C, false); \
    } while (0)
        switch (lanes * 16 + chunks) {
            case 1 * 16 + 1: SMH_R2_LAUNCH(1, 1); break;
            case 1 * 16 + 2: SMH_R2_LAUNCH(1, 2); break;
            case 1 * 16 + 3: SMH_R2_LAUNCH(1, 3); break;
            case 2 * 16 + 1: SMH_R2_LAUNCH(2, 1); break;
            case 2 * 16 + 2: SMH_R2_LAUNCH(2, 2); break;
            case 4 * 16 + 1: SMH_R2_LAUNCH(4, 1); break;
            case 8 * 16 + 1: SMH_R2_LAUNCH(8, 1); break;
            case 16 * 16 + 1: SMH_R2_LAUNCH(16, 1); break;
            case 32 * 16 + 1: SMH_R2_LAUNCH(32, 1); break;
            case 64 * 16 + 1: SMH_R2_LAUNCH(64, 1); break;
            default:
                return fail(SMH_ERR_INVALID, "ring kernel: unsupported (lanes per row, chunks per lane) = (%d, %d)", lanes, chunks);
        }
#undef SMH_R2_LAUNCH
#undef SMH_R2_LAUNCH1
#undef SMH_R2_LAUNCH2
        SMH_HIP(hipGetLastError());
    }
    if (nnz_lim != nnz && lb1 == n_blocks) {  // (the last <= 3 entries belong to the last rows: with the launch that covers them)
        if (dot_partials)
            hipLaunchKernelGGL(k_ring2_tail_dot<T>, dim3(1), dim3(64), 0, s, off, col, val, x, (const T *)y, dot_partials + n_blocks,
                               (uint64_t)n_rows, nnz_lim, (uint64_t)nnz);
        else
            hipLaunchKernelGGL(k_ring2_tail<T>, dim3(1), dim3(64), 0, s, off, col, val, x, y, (uint64_t)n_rows, nnz_lim, (uint64_t)nnz);
        SMH_HIP(hipGetLastError());
    }
    return SMH_OK;
}

// ring_entries: what the phase plan was built for (kRingEntries; kRingEntriesWide for f32 matrices that need it).
// dot_partials != NULL (n_blocks + 1 values): the DOT form, `y` = lhs (see launch_ring2_t).
int launch_spmv_ring2(int dtype, int lanes, int chunks, const uint32_t *off, const uint32_t *col, const uint16_t *col16,
                      const void *val, const void *x, void *y, size_t n_rows, size_t nnz, bool padded, unsigned n_blocks,
                      const uint32_t *phase_ptr, const RingPhase *phases, unsigned ring_entries, unsigned bands,
                      hipStream_t s, void *dot_partials, unsigned block_begin, unsigned block_end) {
    if (n_rows == 0) return SMH_OK;
    if (bands != 1u && bands != 4u) return fail(SMH_ERR_INVALID, "ring kernel: %u bands", bands);
    if (dtype == SMH_F64) {
        if (ring_entries != (unsigned)kRingEntries) return fail(SMH_ERR_INVALID, "f64 ring kernel: ring of %u columns", ring_entries);
        return launch_ring2_t<double, kRingEntries>(lanes, chunks, off, col, col16, (const double *)val, (const double *)x,
                                                    (double *)y, n_rows, nnz, padded, n_blocks, phase_ptr, phases, bands,
                                                    (double *)dot_partials, s, block_begin, block_end);
    }
    if (ring_entries == (unsigned)kRingEntriesWide)
        return launch_ring2_t<float, kRingEntriesWide>(lanes, chunks, off, col, col16, (const float *)val, (const float *)x,
                                                       (float *)y, n_rows, nnz, padded, n_blocks, phase_ptr, phases, bands,
                                                       (float *)dot_partials, s, block_begin, block_end);
    if (ring_entries != (unsigned)kRingEntries) return fail(SMH_ERR_INVALID, "ring kernel: ring of %u columns", ring_entries);
    return launch_ring2_t<float, kRingEntries>(lanes, chunks, off, col, col16, (const float *)val, (const float *)x, (float *)y,
                                               n_rows, nnz, padded, n_blocks, phase_ptr, phases, bands, (float *)dot_partials, s, block_begin, block_end);
}

}  // namespace smh
