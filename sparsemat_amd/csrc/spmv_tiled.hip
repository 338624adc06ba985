// spmv_tiled.hip -- K2t: y = A x in two streaming passes over a 2-D tiled copy of A, for matrices whose columns have no
// locality (BASELINE C2 "uniform", C3).  Replaces the reference loop sparsematrix.rs:146-158 (over sparsemat_crs.rs:102-110)
// for those matrices.  The sum of a row is formed slice by slice (ascending column slices; inside a slice the order written
// down under "ORDER" below), so the result agrees with the reference within the rounding bound of DESIGN.md section 2, not
// bit for bit -- like K1r / K2 / K2c / K2f.  It can return -0.0 where the reference returns +0.0 (a row whose products are all
// -0.0: the reference starts from +0.0, a run here starts from its first product) -- equal in value, documented in
// include/sparsemat_hip.h.
//
// Why two passes: every row-major kernel gathers x[col] through the vector L1, and a gather that misses it costs a cache line
// and an L2 round trip -- 150-190 G gathers/s however the work is arranged ("gather wall").  Here no gather
// leaves the CU:
//   pass 1 "expand":  the entries are stored by column slice (16384 columns), inside a slice by (row, storage order), cut into
//                     CHUNKS of at most 256 entries that one wavefront takes at a time, E = 4 consecutive entries per lane (one
//                     16-byte load of f32 values, two of f64).  A workgroup stages its slice of x in LDS, multiplies (16-bit column codes),
//                     and FOLDS the entries of a chunk that belong to one row (adjacent after the sort) with a segmented scan over
//                     the wavefront; the sums -- one per (row, slice, chunk) -- leave compacted, 16 bytes per lane.  Chunks start
//                     at row boundaries (the build snaps every chunk start forward to the next one, up to 16 entries), so a
//                     (row, slice) pair almost always yields ONE product whatever the row's length.
//   pass 2 "reduce":  inside a slice the products are ordered by row, so those of a block of consecutive rows are one contiguous
//                     TILE per slice.  The row blocks are cut (host, once) so that each holds the same number of PRODUCTS.  One
//                     wavefront per row block walks its tiles slice by slice, 16 bytes of products per lane (4 f32 / 2 f64), and adds them
//                     into wave-private sums in LDS; the rows of a tile are distinct (checked per tile with one ballot; else a
//                     segmented scan merges equal neighbours first), so the adds of a tile are independent: no atomics, no
//                     barriers between them, a fixed order -- bitwise reproducible.  The loads of the next tiles are in flight meanwhile.
//                     The wavefronts of up to 12 ADJACENT row blocks share a workgroup and meet at a barrier before every batch of
//                     tiles: their tiles are neighbours in memory, and reading them at one time is worth 22 % of the pass.
//   pass 1 runs on persistent workgroups (the next item's slice of x travels into registers while the current one is in use).
// Round 2's form (one product per entry, one entry per lane in pass 2: 16 / 28 B per entry, 42 VALU per 48-entry tile) is in
// the history of this file; what changed and what it bought: DESIGN.md section 4, "K2t".
//
// ORDER (what tests/test_tiled_gpu.py restates bit for bit).  prod = round(val * x[col]).  Inside a chunk: lane l holds entries
// E l .. E l + E - 1; s_0 = p_0, s_k = continues_k ? s_{k-1} + p_k : p_k (left fold inside the lane); the lanes' last running sums
// are combined by a Kogge-Stone segmented scan in the order row_shr 1, 2, 4, 8, row_bcast 15, row_bcast 31 (v = stop ? v : v + v_in);
// an entry of a lane's first segment adds the carry of the lanes before it as carry + s_k.  The value at the last entry of a
// run is its product sum.  Pass 2 adds the product sums of a row to its running sum in slice order (a (row, slice) pair cut by a
// chunk boundary has its parts merged first by the same scan, inside one round of 64 E products).
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <algorithm>
#include <chrono>
#include <vector>

#include "internal.hpp"

namespace smh {

int device_exclusive_scan_u32(uint32_t *data, uint64_t n, hipStream_t s, uint64_t *total_out);  // spmv_colblock.hip

// columns per slice and threads per pass-1 workgroup, per value type (A/B builds override them: sparsemat_amd/build.py extra_flags).
// f32: 16384 columns = 64 KiB of x + a 16 KiB stage -> two 1024-thread workgroups per CU.  f64: 16384 columns = 128 KiB + 32 KiB: ONE
// workgroup per CU; the alternative that gives two (8192 columns, 512 threads: 64 + 16 KiB) was measured in round 4 and lost what it
// gained to the product stream, which grows by 18 % on BASELINE C3 when a long row's entries spread over twice as many slices
#ifndef SMH_T3_SLICE_F32
#define SMH_T3_SLICE_F32 16384
#endif
#ifndef SMH_T3_SLICE_F64
#define SMH_T3_SLICE_F64 16384
#endif
#ifndef SMH_T3_THREADS_F32
#define SMH_T3_THREADS_F32 1024
#endif
#ifndef SMH_T3_THREADS_F64
#define SMH_T3_THREADS_F64 1024
#endif
constexpr uint32_t kT3Snap = 16;         // a chunk start moves forward by up to this many entries to the next row boundary
constexpr int kT3Batch = 4;              // tiles whose loads are in flight together, per wavefront
constexpr int kT3Ahead = 3;              // pass 1: chunks whose loads are in flight per wavefront
constexpr uint32_t kT3RowGroup = 32;      // row blocks begin at multiples of this many rows (the build counts products per group)
constexpr uint32_t kT3Cont = 0x8000u;    // code bit 15: same row as the entry before (never set on a chunk's first entry)
constexpr uint32_t kT3ColMask = 0x3FFFu;

template <typename T> struct T3;
template <> struct T3<float> {
    static constexpr int E1 = 4, E2 = 4;                     // entries per lane: pass 1 (a chunk = 64 E1), pass 2 (a round = 64 E2)
    typedef float V1 __attribute__((ext_vector_type(4)));
    typedef float V2 __attribute__((ext_vector_type(4)));
    typedef uint32_t C2 __attribute__((ext_vector_type(2))); // 4 x u16
    static constexpr uint32_t kSlice = SMH_T3_SLICE_F32;
    static constexpr int kThreads = SMH_T3_THREADS_F32;      // pass 1: 16 wavefronts, one chunk each per step
    static constexpr uint32_t kCapRows = 3328;               // most rows of a row block: 3329 sums = 13 KiB of LDS per (one-wavefront) workgroup; measured, profiles/r03_k2t_rewrite_sweep1.log
};
template <> struct T3<double> {
    static constexpr int E1 = 4, E2 = 2;                     // (pass 1: two 16-byte loads of values per lane; pass 2: one)
    typedef double V1 __attribute__((ext_vector_type(4)));
    typedef double V2 __attribute__((ext_vector_type(2)));
    typedef uint32_t C2;                                     // 2 x u16
    static constexpr uint32_t kSlice = SMH_T3_SLICE_F64;
    static constexpr int kThreads = SMH_T3_THREADS_F64;
    static constexpr uint32_t kCapRows = 1664;               // 1665 x 8 B = 13 KiB
};
typedef uint32_t T3C1 __attribute__((ext_vector_type(2)));   // pass 1: the 4 codes of a lane
typedef uint32_t t3_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t t3_u8 __attribute__((ext_vector_type(8)));
constexpr int kT3Rsrc = 0x00020000;                          // dword 3 of a raw buffer descriptor on gfx9 (32-bit data format)
template <typename T> constexpr uint32_t t3_chunk() { return 64u * T3<T>::E1; }           // slots of a chunk: 256
template <typename T> constexpr uint32_t t3_stride() { return t3_chunk<T>() - kT3Snap; }  // nominal entries per chunk: 240

static double t3_tile_target(int dtype) {
    // mean products per tile: a round takes 64 E2 of them (256 / 128); beyond ~0.7 of that too many tiles need a second round
    double v = dtype == SMH_F64 ? 60.0 : 174.0;
    if (const char *e = getenv("SMH_TILED_TILE")) {  // tuning knob
        const double w = atof(e);
        if (w >= 8.0 && w <= 256.0) v = w;
    }
    return v;
}
static uint32_t t3_slice(int dtype) { return dtype == SMH_F64 ? T3<double>::kSlice : T3<float>::kSlice; }
static_assert(T3<float>::kSlice <= 16384 && T3<double>::kSlice <= 16384 && T3<float>::kSlice % T3<float>::kThreads == 0 &&
              T3<double>::kSlice % T3<double>::kThreads == 0, "14-bit column codes; whole values of x per thread");
static uint32_t t3_cap_rows(int dtype) {
    uint32_t cap = dtype == SMH_F64 ? T3<double>::kCapRows : T3<float>::kCapRows;
    if (const char *e = getenv("SMH_TILED_CAP")) {  // tuning knob: most rows of a row block
        const int v = atoi(e);
        if (v >= 1 && v <= 16384) cap = (uint32_t)v;
    }
    return cap;
}

// ---- DPP helpers ------------------------------------------------------------------------------------------------------
template <int CTRL, int ROWS> __device__ __forceinline__ uint32_t t3_dpp(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROWS, 0xF, false);
}
template <int CTRL, int ROWS> __device__ __forceinline__ float t3_dpp(float old, float v) {
    return __uint_as_float(t3_dpp<CTRL, ROWS>(__float_as_uint(old), __float_as_uint(v)));
}
template <int CTRL, int ROWS> __device__ __forceinline__ double t3_dpp(double old, double v) {
    const uint64_t o = (uint64_t)__double_as_longlong(old), b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = t3_dpp<CTRL, ROWS>((uint32_t)o, (uint32_t)b), hi = t3_dpp<CTRL, ROWS>((uint32_t)(o >> 32), (uint32_t)(b >> 32));
    return __longlong_as_double((long long)((uint64_t)hi << 32 | lo));
}
// one step of the segmented inclusive scan: (v, stop) of this lane takes (v_in, stop_in) of the lane the control names; a lane
// without a source (row start, rows not in ROWS) keeps its values (v_in = 0 is never added there: stop_in = 1 is not set either,
// so the add is guarded by `has`)
template <int CTRL, int ROWS, typename T> __device__ __forceinline__ void t3_scan_step(T &v, uint32_t &stop) {
    const T v_in = t3_dpp<CTRL, ROWS>(T(0), v);
    const uint32_t in = t3_dpp<CTRL, ROWS>(2u, stop);  // 2: no source lane
    if (in != 2u && !stop) v = v + v_in;
    stop |= (in & 1u);
}
// R_l = the running sum at the end of lane l, where lane l passes what comes from the left on only if stop_l == 0
template <typename T> __device__ __forceinline__ T t3_seg_scan(T v, uint32_t stop) {
    t3_scan_step<0x111, 0xF>(v, stop);  // row_shr:1
    t3_scan_step<0x112, 0xF>(v, stop);  // row_shr:2
    t3_scan_step<0x114, 0xF>(v, stop);  // row_shr:4
    t3_scan_step<0x118, 0xF>(v, stop);  // row_shr:8
    t3_scan_step<0x142, 0xA>(v, stop);  // row_bcast:15 into rows 1 and 3
    t3_scan_step<0x143, 0xC>(v, stop);  // row_bcast:31 into rows 2 and 3
    return v;
}

// The fold of one chunk / one round held E entries per lane: p[k] products, cont bit k = entry k continues the run of the entry
// before it (bit 0 of lane 0 must be clear).  On return p[k] is the running sum of its run up to entry k (so the value at a
// run's LAST entry is the run's sum) and the return value has bit k set where entry k is the last of its run -- as far as `cont`
// of the following entry tells (the caller masks entries that do not exist).
template <typename T, int E> __device__ __forceinline__ uint32_t t3_fold_runs(T (&p)[E], uint32_t cont) {
    uint32_t through = cont & 1u;  // all entries up to k continue: they belong to the run that enters the lane
    uint32_t first_seg = through;  // bit k: entry k is part of that run
#pragma unroll
    for (int k = 1; k < E; ++k) {
        if (cont >> k & 1u) p[k] = p[k - 1] + p[k];
        through &= cont >> k;
        first_seg |= (through & 1u) << k;
    }
    // the lanes' last running sums, combined over the wavefront; a lane stops what comes from the left unless all of its entries
    // continue -- and when no lane does (the usual case: a run rarely spans a whole lane) the scan has nothing to combine: the
    // running sum at the end of a lane is its own last sum (the scan below returns exactly that for all-stop flags)
    T run = p[E - 1];
    if (__ballot(through & 1u)) run = t3_seg_scan<T>(run, (through & 1u) ^ 1u);
    const T carry = t3_dpp<0x138, 0xF>(T(0), run);  // wave_shr:1 -- the running sum at the end of the lane before
#pragma unroll
    for (int k = 0; k < E; ++k)
        if (first_seg >> k & 1u) p[k] = carry + p[k];
    const uint32_t next0 = t3_dpp<0x130, 0xF>(0u, cont & 1u);  // wave_shl:1 -- does the next lane's first entry continue?  (lane 63: no)
    return (~(cont >> 1 | next0 << (E - 1))) & ((1u << E) - 1u);
}

// ---- pass 1 -----------------------------------------------------------------------------------------------------------
// chunk c: slots [c * CH, c * CH + len) of val / code hold its entries (the rest of the CH slots is zero), its product sums
// go to prod[obase ...), 16 bytes per lane (the last piece may be padded: the build gives those slots the dump row).
struct T3Chunk { uint32_t obase, len; };

template <typename T>
struct T3Slot {
    typename T3<T>::V1 v;
    T3C1 cd;
    uint32_t ob, ln;
};

template <typename T, int AHEAD>
__global__ __launch_bounds__(T3<T>::kThreads) void k_t3_expand(const T *__restrict__ x, uint64_t x_len, const T *__restrict__ val,
                                                                 const uint16_t *__restrict__ code, const uint32_t *__restrict__ cptr,
                                                                 const T3Chunk *__restrict__ chunk, T *__restrict__ prod, uint32_t parts,
                                                                 uint32_t n_items, uint32_t per_xcd) {
    using V = typename T3<T>::V1;
    using V2 = typename T3<T>::V2;
    constexpr int E = T3<T>::E1, E2 = T3<T>::E2;
    static_assert(sizeof(V2) == 16, "16 bytes per store");
    constexpr uint32_t CH = 64u * E;
    constexpr uint32_t kT3Slice = T3<T>::kSlice;
    constexpr int kT3ExpandThreads = T3<T>::kThreads;
    extern __shared__ __attribute__((aligned(16))) char t3_smem[];
    T *xs = (T *)t3_smem;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    T *stage = xs + kT3Slice + wave * CH;
    // PERSISTENT workgroups (the grid is what the chip holds at once): workgroups b, b + 8, ... share an XCD (round-robin dispatch);
    // an XCD gets a contiguous run of `per_xcd` (slice, part) items -- the parts of a slice then stage its x from one L2 instead of
    // fetching it into eight -- and its workgroups deal them out among themselves.  While an item's chunks are folded the NEXT item's
    // slice of x is already on its way into registers (16 values per thread), so a workgroup pays the latency of staging x once,
    // not once per item: without the staging pass 1 ran 5-13 % faster, f64 -- one workgroup per CU, nobody to overlap with --
    // the most.  (per_xcd == 0, the knob's other setting: the items are dealt round-robin over all workgroups.)
    uint32_t g, g_step, g_end;
    if (per_xcd) {
        const uint32_t xcd = blockIdx.x & 7u;
        g = xcd * per_xcd + (blockIdx.x >> 3);
        g_step = gridDim.x >> 3;
        g_end = (xcd + 1u) * per_xcd < n_items ? (xcd + 1u) * per_xcd : n_items;
    } else {
        g = blockIdx.x;
        g_step = gridDim.x;
        g_end = n_items;
    }
    if (g >= g_end) return;  // (whole workgroup, before any barrier)
    constexpr int NP = (int)(kT3Slice / kT3ExpandThreads);  // values of x per thread and slice
    T pre[NP];
    // (buffer loads whose descriptor ends with x -- or holds nothing when there is no next item --: no branch around a load, see below)
    auto fetch = [&](uint32_t item, bool any) {  // the slice of `item` into registers (columns past x_len: zero)
        const uint64_t c0 = (uint64_t)(item / parts) * kT3Slice;
        const uint64_t left = any && c0 < x_len ? x_len - c0 : 0;
        const uint32_t have = (uint32_t)__builtin_amdgcn_readfirstlane((int)(left < kT3Slice ? (uint32_t)left : kT3Slice));
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)(x + (have ? c0 : 0)), 0, (int)(have * sizeof(T)), kT3Rsrc);
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const uint32_t at = (threadIdx.x + (uint32_t)u * kT3ExpandThreads) * (uint32_t)sizeof(T);
            if constexpr (sizeof(T) == 4) pre[u] = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)at, 0, 0));
            else pre[u] = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(rx, (int)at, 0, 0));
        }
    };
    fetch(g, true);
  for (;;) {
#pragma unroll
    for (int u = 0; u < NP; ++u) xs[threadIdx.x + (uint32_t)u * kT3ExpandThreads] = pre[u];
    __syncthreads();
    const bool more = g + g_step < g_end;  // (workgroup-uniform)
    const uint32_t cb = g / parts, part = g % parts;
    const uint32_t c_lo = cptr[cb], c_hi = cptr[cb + 1];
    const uint32_t per = (c_hi - c_lo + parts - 1) / parts;
    const uint32_t k0 = c_lo + part * per, k1 = k0 + per < c_hi ? k0 + per : c_hi;
    // every wavefront takes a contiguous run of the part's chunks (their descriptors are then one coalesced load per 64 chunks,
    // handed out with v_readlane), AHEAD chunks' loads in flight while one is folded
    constexpr uint32_t W = kT3ExpandThreads / 64;
    const uint32_t n_w = (k1 > k0 ? k1 - k0 + W - 1 : 0u) / W;
    // (wave-uniform by construction; said to the compiler so that the buffer descriptors below live in scalar registers)
    const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(k0 + wave * n_w < k1 ? k0 + wave * n_w : k1));
    const uint32_t w1 = w0 + n_w < k1 ? w0 + n_w : k1;
    const uint32_t pos = E * lane;
    // lane l: the descriptor of the wavefront's l-th chunk (the launch sizes the parts so that a wavefront never has more than 64)
    uint32_t my_ob = 0, my_ln = 0;
    if (w0 + lane < w1) {
        const T3Chunk d = chunk[w0 + lane];
        my_ob = d.obase;
        my_ln = d.len;
    }
    // the next item's slice: requested AFTER the descriptors (loads return in order: the chunk loop must not wait for the slice to
    // get its first descriptor) and before this item's chunks, whose latency it shares
    __builtin_amdgcn_sched_barrier(0);
    fetch(g + g_step, more);
    __builtin_amdgcn_sched_barrier(0);
    // Loads and stores go through BUFFER instructions whose descriptor covers exactly the chunk's pieces: a lane past them reads
    // zeros / stores nothing, without a branch -- a branch around a load or store makes the compiler drain every load in flight
    // before the next use (it cannot count them any more), which is what kept the first form of this loop at one chunk in flight.
    auto issue = [&](T3Slot<T> &S, uint32_t i) {  // the i-th chunk of this wavefront (beyond its run: an empty one)
        // (i can reach 64 and 65 when the run holds 64 chunks and AHEAD does not divide 64: `i & 63` would then name the run's FIRST
        // chunks again and their product slots would be overwritten -- a slot past the run must be empty whatever lane it maps to)
        const bool mine = w0 + i < w1;  // (wave-uniform)
        S.ob = mine ? (uint32_t)__builtin_amdgcn_readlane((int)my_ob, (int)(i & 63u)) : 0u;
        S.ln = mine ? (uint32_t)__builtin_amdgcn_readlane((int)my_ln, (int)(i & 63u)) : 0u;
        const uint64_t at = (uint64_t)(w0 + i) * CH;
        const uint32_t pieces = (S.ln + E - 1) & ~(uint32_t)(E - 1);  // whole pieces (the rest of the chunk's slots is zero)
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(val + at), 0, (int)(pieces * sizeof(T)), kT3Rsrc);
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)(code + at), 0, (int)(pieces * 2u), kT3Rsrc);
        if constexpr (sizeof(T) == 4) {
            S.v = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rv, (int)(pos * 4u), 0, 2 /* nt */));
        } else {
            const t3_u4 a = __builtin_amdgcn_raw_buffer_load_b128(rv, (int)(pos * 8u), 0, 2);
            const t3_u4 b = __builtin_amdgcn_raw_buffer_load_b128(rv, (int)(pos * 8u + 16u), 0, 2);
            const t3_u8 ab = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            S.v = __builtin_bit_cast(V, ab);
        }
        S.cd = __builtin_bit_cast(T3C1, __builtin_amdgcn_raw_buffer_load_b64(rc, (int)(pos * 2u), 0, 2));
    };
    auto process = [&](const T3Slot<T> &S) {
        T p[E];
        p[0] = S.v.x * xs[S.cd.x & kT3ColMask]; p[1] = S.v.y * xs[S.cd.x >> 16 & kT3ColMask];
        p[2] = S.v.z * xs[S.cd.y & kT3ColMask]; p[3] = S.v.w * xs[S.cd.y >> 16 & kT3ColMask];
        const uint32_t cont = (S.cd.x >> 15 & 1u) | (S.cd.x >> 31) << 1 | (S.cd.y >> 15 & 1u) << 2 | (S.cd.y >> 31) << 3;
        uint32_t tail = t3_fold_runs<T, E>(p, cont);
        // entries that exist: pos + k < len (the slots after them hold zeros without continuation bits: they reach no run's sum,
        // and no run end is taken from them)
        const uint32_t have = pos >= S.ln ? 0u : (S.ln - pos >= (uint32_t)E ? (1u << E) - 1u : (1u << (S.ln - pos)) - 1u);
        tail &= have;
        // where each run's sum goes: the number of run ends before it
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const uint64_t b = __ballot(tail >> k & 1u);
            before += __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
            total += (uint32_t)__popcll(b);
        }
#pragma unroll
        for (int k = 0; k < E; ++k) {
            if (tail >> k & 1u) stage[before] = p[k];
            before += tail >> k & 1u;
        }
        // the padding of the last 16-byte piece is zero: where a chunk boundary cuts a (row, slice) pair the build gives these
        // slots the pair's row, so that pass 2 sees its parts as neighbours and merges them (else they carry the dump row)
        const uint32_t total_r = (total + E2 - 1) & ~(uint32_t)(E2 - 1);
        if (lane < (uint32_t)E2 && total + lane < total_r) stage[total + lane] = T(0);
        __builtin_amdgcn_wave_barrier();
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)(prod + S.ob), 0, (int)(total_r * sizeof(T)), kT3Rsrc);
#pragma unroll
        for (int h = 0; h < E / E2; ++h)  // 16 bytes per store; a lane past the chunk's products stores nothing
            __builtin_amdgcn_raw_buffer_store_b128(*(const t3_u4 *)(stage + pos + h * E2), rp, (int)((pos + h * E2) * sizeof(T)), 0, 0);
        __builtin_amdgcn_wave_barrier();
    };
    T3Slot<T> S[AHEAD];
    // (in THIS order: left alone the scheduler issued the prologue's loads last chunk first, and the loop -- which has to be right
    // for its first trip too -- then waited for every load in flight before each chunk)
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) {
        issue(S[u], (uint32_t)u);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (uint32_t i = 0; w0 + i < w1; i += AHEAD) {
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) {
            process(S[u]);
            issue(S[u], i + u + AHEAD);
        }
    }
    if (!more) break;
    g += g_step;
    __syncthreads();  // every wavefront is done with this slice before the next one overwrites it
  }
}

// ---- pass 2 -----------------------------------------------------------------------------------------------------------
template <typename T, int NB>
struct T3Tiles {
    typename T3<T>::V2 pv[NB];
    typename T3<T>::C2 rv[NB];
    uint32_t bs[NB], ln[NB];
};

// tstart[rb * n_cb + cb] = index (into prod / rowc) of the first product of tile (cb, rb); row n_rb of the table holds the ends of
// the last row block's tiles.  rowc: (the product's row relative to its block + 1) * sizeof(T) -- the byte offset of the row's sum
// in the wavefront's LDS -- or 0, the dump slot, for padding (so a lane past the tile, which reads zeros, lands there too).
// One wavefront per row block, its (R + 1) sums in LDS at `stride` bytes per wavefront; the blockDim / 64 wavefronts of a workgroup
// own ADJACENT row blocks and meet at a barrier before every batch of tiles, so that the workgroup's loads of a slice -- neighbours
// in memory -- are issued together (launch_t says what that is worth).  Nothing else is shared: the sums, the adds and their order are
// each wavefront's own.  DUPS: the copy has (row, slice) pairs cut by a chunk boundary, i.e. a tile may hold equal neighbours
// (checked per round; merged first); without them the rows of a round are distinct by construction.
template <typename T, bool DUPS, int NB>
__global__ __launch_bounds__(1024) void k_t3_reduce(const T *__restrict__ prod, const uint16_t *__restrict__ rowc,
                                                   const uint32_t *__restrict__ tstart, uint32_t n_cb, uint32_t n_rb,
                                                   const uint32_t *__restrict__ rb_start, T *__restrict__ y, uint32_t xcd_map, uint32_t stride) {
    using V = typename T3<T>::V2;
    using C = typename T3<T>::C2;
    constexpr int E = T3<T>::E2;
    constexpr uint32_t RND = 64u * E;
    extern __shared__ __attribute__((aligned(16))) char t3_smem[];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t gb = blockIdx.x;  // neighbouring row blocks' tiles share cache lines: neighbours on one XCD (one L2)
    if (xcd_map) gb = (blockIdx.x & 7u) * ((gridDim.x + 7u) / 8u) + (blockIdx.x >> 3);
    const uint32_t rbq = gb * (blockDim.x >> 6) + wv;
    const bool live = rbq < n_rb;  // (a wavefront without a row block still keeps every barrier: it walks empty tiles)
    const uint32_t rb = live ? rbq : 0u;
    char *accb = t3_smem + (size_t)wv * stride;
    T *acc = (T *)accb;  // [0]: dump; [1 + i]: row r0 + i
    const uint32_t r0 = rb_start[rb], rows = live ? rb_start[rb + 1] - r0 : 0u;
    for (uint32_t i = lane; i <= rows; i += 64) acc[i] = T(0);
    const uint32_t *ts0 = tstart + (size_t)rb * n_cb, *ts1 = ts0 + n_cb;
    auto table = [&](uint32_t t, uint32_t &base, uint32_t &len) {  // lane l: tile t + l
        const uint32_t cbl = t + lane;
        base = 0;
        len = 0;
        if (cbl < n_cb && live) {
            base = ts0[cbl];
            len = ts1[cbl] - base;
        }
    };
    uint32_t cur_base, nxt_base, cur_len, nxt_len, win = 0;  // cur_*: tiles [win, win + 64), nxt_*: the 64 after them
    table(0, cur_base, cur_len);
    table(64, nxt_base, nxt_len);
    const uint32_t pos = E * lane;
    auto issue = [&](T3Tiles<T, NB> &B, uint32_t j0) {
        if (j0 >= win + 64) {
            cur_base = nxt_base;
            cur_len = nxt_len;
            win += 64;
            table(win + 64, nxt_base, nxt_len);
        }
#pragma unroll
        for (int d = 0; d < NB; ++d) {
            const int j = (int)((j0 + d) & 63);
            B.bs[d] = (uint32_t)__builtin_amdgcn_readlane((int)cur_base, j);
            B.ln[d] = (uint32_t)__builtin_amdgcn_readlane((int)cur_len, j);  // 0 past the last slice
        }
#pragma unroll
        for (int d = 0; d < NB; ++d) {
            // aligned 16-byte BUFFER loads whose descriptor ends with the tile: the lanes past it fetch nothing and read zeros -- the
            // dump row -- (a tile of 174 products fills 44 lanes; plain loads had the other 20 fetch the neighbours' products), and no
            // branch surrounds the loads
            const uint32_t a0 = B.bs[d] & ~(uint32_t)(E - 1);
            const uint32_t n = (B.bs[d] + B.ln[d] - a0 + E - 1) & ~(uint32_t)(E - 1);
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)(prod + a0), 0, (int)(n * sizeof(T)), kT3Rsrc);
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *)(rowc + a0), 0, (int)(n * 2u), kT3Rsrc);
            B.pv[d] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rp, (int)(pos * sizeof(T)), 0, 0));
            if constexpr (E == 4) B.rv[d] = __builtin_bit_cast(C, __builtin_amdgcn_raw_buffer_load_b64(rr, (int)(pos * 2u), 0, 0));
            else B.rv[d] = __builtin_amdgcn_raw_buffer_load_b32(rr, (int)(pos * 2u), 0, 0);
        }
    };
    // one round: up to 64 E consecutive products, E per lane, the first `skip` of the round (lane 0's) and those from `stop` on (the
    // last piece's, when the tile ends inside it and its neighbour's products follow) belonging to other tiles
    auto round = [&](const V &pv, const C &rv, uint32_t skip, uint32_t stop) {
        T p[E];
        uint32_t r[E];
        if constexpr (E == 4) {
            p[0] = pv.x; p[1] = pv.y; p[2] = pv.z; p[3] = pv.w;
            r[0] = rv.x & 0xFFFFu; r[1] = rv.x >> 16; r[2] = rv.y & 0xFFFFu; r[3] = rv.y >> 16;
        } else {
            p[0] = pv.x; p[1] = pv.y;
            r[0] = rv & 0xFFFFu; r[1] = rv >> 16;
        }
#pragma unroll
        for (int k = 0; k < E; ++k)
            if (pos + k < skip || pos + k >= stop) r[k] = 0;  // (the dump slot: what is added there is never read)
        if constexpr (DUPS) {
            // equal neighbours (a (row, slice) pair cut by a chunk boundary; the dump row never counts)
            const uint32_t prev_r = t3_dpp<0x138, 0xF>(0u, r[E - 1]);  // wave_shr:1
            uint32_t cont = (uint32_t)(r[0] == prev_r && r[0] != 0u);
#pragma unroll
            for (int k = 1; k < E; ++k) cont |= (uint32_t)(r[k] == r[k - 1] && r[k] != 0u) << k;
            if (__ballot(cont != 0)) {
                // merge them first (same fold as pass 1), then only the last entry of each run adds
                const uint32_t tail = t3_fold_runs<T, E>(p, cont);
#pragma unroll
                for (int k = 0; k < E; ++k)
                    if (!(tail >> k & 1u)) r[k] = 0;
            }
        }
        // the rows of the round are distinct (sorted, no equal neighbours): independent read-add-write sequences, all reads first.
        // Code 0 (padding, other tiles' products, lanes past the tile) is the dump slot, whose value is never read.  (LDS float atomics
        // -- one ds_add_f32 instead of read, add, write -- measured 4x SLOWER here: 1.77 ms against 0.42 for C2-uniform's pass 2,
        // ~150 cycles per wavefront instruction; ds_add_f64 no faster than the three instructions it replaces.)
        T s[E];
#pragma unroll
        for (int k = 0; k < E; ++k) s[k] = *(const T *)(accb + r[k]);
#pragma unroll
        for (int k = 0; k < E; ++k) *(T *)(accb + r[k]) = s[k] + p[k];  // (unconditional: a branch per write cost 20 %)
    };
    auto fold = [&](T3Tiles<T, NB> &B) {
#pragma unroll
        for (int d = 0; d < NB; ++d) {
            const uint32_t bs = B.bs[d], end = bs + B.ln[d];
            const uint32_t a0 = bs & ~(uint32_t)(E - 1);
            round(B.pv[d], B.rv[d], bs - a0, end - a0);
            for (uint32_t a = a0 + RND; a < end; a += RND) {  // rare: a tile of more than one round
                V pv = V(0);
                C rv = C(0);
                if (a + pos < end) {
                    pv = *(const V *)(prod + (uint64_t)a + pos);
                    rv = *(const C *)(rowc + (uint64_t)a + pos);
                }
                round(pv, rv, 0u, end - a);
            }
        }
    };
    T3Tiles<T, NB> A, B;
    issue(A, 0);
    __builtin_amdgcn_sched_barrier(0);
    for (uint32_t j0 = 0; j0 < n_cb; j0 += 2 * NB) {
        __syncthreads();  // lock step with the workgroup's other row blocks (the trip count is the same for all: n_cb)
        issue(B, j0 + NB);
        fold(A);
        issue(A, j0 + 2 * NB);
        fold(B);
    }
    for (uint32_t i = lane; i < rows; i += 64) y[(uint64_t)r0 + i] = acc[i + 1];
}

// ---- plan -------------------------------------------------------------------------------------------------------------
// The build sorts the ENTRIES THEMSELVES -- {row, column, value} travels as the sort's payload -- by column slice (stable: inside a
// slice the (row, storage) order of the CRS arrays stays).  Round 3 sorted entry indices and gathered values / columns / rows through
// them afterwards: 320 M random 4- and 8-byte reads plus a binary search per entry for its row, 21 of the build's 54 ms on BASELINE
// C2-uniform; now every later kernel streams the sorted entries.
template <typename T> struct T3Entry { uint32_t row, col; T val; };
template <typename T> struct T3LoadEntry {  // entry i of the CRS arrays (rows_e: the row of every entry, assemble.hip::expand_rows)
    const uint32_t *rows_e, *col;
    const T *val;
    __device__ __forceinline__ T3Entry<T> operator()(uint64_t i) const { return T3Entry<T>{rows_e[i], col[i], val[i]}; }
};
struct T3LoadKey {
    const uint32_t *col;
    uint32_t slice;
    __device__ __forceinline__ uint32_t operator()(uint64_t i) const { return col[i] / slice; }
};

// start[b] = first position of the sorted keys holding a value >= b (b = 0 .. n_cb)
__global__ __launch_bounds__(kBlock) void k_t3_bounds(const uint32_t *__restrict__ key_s, uint64_t nnz, uint32_t n_cb, uint64_t *__restrict__ start) {
    const uint32_t b = blockIdx.x * kBlock + threadIdx.x;
    if (b > n_cb) return;
    uint64_t lo = 0, hi = nnz;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) / 2;
        if (key_s[mid] < b) lo = mid + 1; else hi = mid;
    }
    start[b] = lo;
}

// the slice of chunk c: the last s with cptr[s] <= c
__device__ __forceinline__ uint32_t t3_slice_of(const uint32_t *__restrict__ cptr, uint32_t n_cb, uint32_t c) {
    uint32_t lo = 0, hi = n_cb;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (cptr[mid] <= c) lo = mid; else hi = mid;
    }
    return lo;
}

// cstart[c] = where chunk c begins among its slice's sorted entries: nominally at (c - cptr[s]) * stride, moved forward to the
// next row boundary when there is one within kT3Snap entries (so that a (row, slice) pair is not cut); never past the slice's end
template <typename T>
__global__ __launch_bounds__(kBlock) void k_t3_chunk_starts(const uint32_t *__restrict__ cptr, uint32_t n_cb, uint32_t n_chunks,
                                                             const uint64_t *__restrict__ start, const T3Entry<T> *__restrict__ ps,
                                                             uint32_t stride, uint32_t *__restrict__ cstart) {
    const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= n_chunks) return;
    const uint32_t s = t3_slice_of(cptr, n_cb, c);
    const uint64_t q0 = start[s], cnt = start[s + 1] - q0;
    const uint64_t nominal = (uint64_t)(c - cptr[s]) * stride;
    uint64_t at = nominal < cnt ? nominal : cnt;
    for (uint32_t j = 0; j < kT3Snap; ++j) {
        const uint64_t q = nominal + j;
        if (q >= cnt) { at = cnt; break; }
        if (q == 0 || ps[q0 + q].row != ps[q0 + q - 1].row) { at = q; break; }
    }
    cstart[c] = (uint32_t)at;
}

// one wavefront per chunk: the chunk's slots of val / code (continuation bits included) and the number of its runs
template <typename T>
__global__ __launch_bounds__(kBlock) void k_t3_fill(const uint32_t *__restrict__ cptr, uint32_t n_cb, uint32_t n_chunks, const uint64_t *__restrict__ start,
                                                     const uint32_t *__restrict__ cstart, const T3Entry<T> *__restrict__ ps,
                                                     T *__restrict__ val_a, uint16_t *__restrict__ code_a, uint32_t *__restrict__ clen,
                                                     uint32_t *__restrict__ ntails) {
    constexpr uint32_t CH = t3_chunk<T>();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const uint32_t s = t3_slice_of(cptr, n_cb, c);
    const uint64_t q0 = start[s], cnt = start[s + 1] - q0;
    const uint32_t a = cstart[c], b = c + 1 < cptr[s + 1] ? cstart[c + 1] : (uint32_t)cnt;
    const uint32_t len = b - a;  // <= CH - 1
    uint32_t runs = 0;
    for (uint32_t j = lane; j < CH; j += 64) {
        T v = T(0);
        uint32_t cd = 0;
        if (j < len) {
            const uint64_t q = q0 + a + j;
            const T3Entry<T> e = ps[q];
            v = e.val;
            cd = e.col - s * T3<T>::kSlice;
            if (j > 0 && e.row == ps[q - 1].row) cd |= kT3Cont;
            else ++runs;
        }
        val_a[(uint64_t)c * CH + j] = v;
        code_a[(uint64_t)c * CH + j] = (uint16_t)cd;
    }
    for (int o = 32; o; o >>= 1) runs += (uint32_t)__shfl_down((int)runs, o, 64);
    if (lane == 0) {
        clen[c] = len;
        ntails[c] = (runs + (uint32_t)T3<T>::E2 - 1u) & ~((uint32_t)T3<T>::E2 - 1u);  // the chunk's share of prod: whole 16-byte pieces
    }
}

// one wavefront per chunk: prow[obase + j] = row of the chunk's j-th run (the padding of the last piece repeats the last row, so
// that the array stays sorted inside a slice); rcount[row] += 1 per run; desc[c] = {obase, len}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_t3_prow(const uint32_t *__restrict__ cptr, uint32_t n_cb, uint32_t n_chunks, const uint64_t *__restrict__ start,
                                                     const uint32_t *__restrict__ cstart, const uint32_t *__restrict__ clen,
                                                     const uint32_t *__restrict__ obase, const T3Entry<T> *__restrict__ ps,
                                                     const uint16_t *__restrict__ code_a, uint32_t *__restrict__ prow, uint32_t *__restrict__ preal,
                                                     uint32_t *__restrict__ gcount, uint32_t group, T3Chunk *__restrict__ desc, uint32_t *__restrict__ any_cut) {
    constexpr uint32_t CH = t3_chunk<T>();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const uint32_t s = t3_slice_of(cptr, n_cb, c);
    const uint64_t q0 = start[s] + cstart[c];
    const uint32_t len = clen[c], ob = obase[c], slots = obase[c + 1] - ob;
    if (lane == 0) desc[c] = T3Chunk{ob, len};
    uint32_t done = 0, last_row = 0;
    for (uint32_t j0 = 0; j0 < CH; j0 += 64) {  // run starts in order: the k-th one names product slot k
        const uint32_t j = j0 + lane;
        const bool head = j < len && !(code_a[(uint64_t)c * CH + j] & kT3Cont);
        const uint64_t m = __ballot(head);
        const uint32_t rank = done + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (head) {
            const uint32_t row = ps[q0 + j].row;
            prow[ob + rank] = row;
            atomicAdd(gcount + row / group, 1u);  // products per group of rows: what the row blocks are cut by
        }
        done += (uint32_t)__popcll(m);
    }
    if (len) last_row = ps[q0 + len - 1].row;
    for (uint32_t k = done + lane; k < slots; k += 64) prow[ob + k] = last_row;
    // does the chunk's last run go on in the next chunk (a (row, slice) pair longer than the snap distance, cut here)?
    bool cut = false;
    if (len && c + 1 < cptr[s + 1] && clen[c + 1]) cut = ps[start[s] + cstart[c + 1]].row == last_row;
    if (lane == 0) {
        preal[c] = done | (cut ? 0x80000000u : 0u);
        if (cut) atomicOr(any_cut, 1u);
    }
}

// rowc[i] = (the row of product slot i relative to its row block + 1) * value_bytes: where the row's sum lives in pass 2's LDS;
// the padding slots of a chunk's last piece: 0 (the dump slot), or the row of the chunk's last run when that run goes on in the
// next chunk
__global__ __launch_bounds__(kBlock) void k_t3_rowcode(uint32_t n_chunks, const uint32_t *__restrict__ obase, const uint32_t *__restrict__ preal,
                                                        const uint32_t *__restrict__ prow, const uint32_t *__restrict__ rb_start, uint32_t n_rb,
                                                        uint32_t value_bytes, uint16_t *__restrict__ rowc) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const uint32_t ob = obase[c], slots = obase[c + 1] - ob, real = preal[c] & 0x7FFFFFFFu;
    const bool cut = preal[c] >> 31;  // the padding then belongs to the cut pair's row (its value is zero): see k_t3_expand
    for (uint32_t k = lane; k < slots; k += 64) {
        uint32_t code = 0;
        if (k < real || cut) {
            const uint32_t row = prow[ob + k];
            uint32_t bl = 0, bh = n_rb;  // the row block: the last one with rb_start[b] <= row
            while (bl + 1 < bh) {
                const uint32_t mid = (bl + bh) / 2;
                if (rb_start[mid] <= row) bl = mid; else bh = mid;
            }
            code = (row - rb_start[bl] + 1u) * value_bytes;
        }
        rowc[ob + k] = (uint16_t)code;
    }
}

// tstart[rb * n_cb + cb], rb = 0 .. n_rb: the first product slot of slice cb whose row is >= rb_start[rb] (= n_rows for rb = n_rb)
__global__ __launch_bounds__(kBlock) void k_t3_table(const uint32_t *__restrict__ prow, const uint32_t *__restrict__ cptr, const uint32_t *__restrict__ obase,
                                                      uint32_t n_cb, uint32_t n_rb, const uint32_t *__restrict__ rb_start, uint32_t *__restrict__ tstart) {
    const uint64_t total = (uint64_t)(n_rb + 1) * n_cb;
    for (uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (uint64_t)gridDim.x * kBlock) {
        const uint32_t rb = (uint32_t)(t / n_cb), cb = (uint32_t)(t % n_cb);
        const uint32_t first_row = rb_start[rb];
        uint32_t lo = obase[cptr[cb]], hi = obase[cptr[cb + 1]];
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (prow[mid] < first_row) lo = mid + 1; else hi = mid;
        }
        tstart[t] = lo;
    }
}

static unsigned t3_bits_for(uint64_t v) {
    unsigned b = 1;
    while (b < 64 && (v >> b)) ++b;
    return b;
}

struct T3Scratch {
    void *p[16] = {};
    int n = 0;
    template <typename U> int alloc(U **out, size_t count) {
        SMH_HIP(hipMalloc((void **)out, (count ? count : 1) * sizeof(U)));
        p[n++] = *out;
        return SMH_OK;
    }
    ~T3Scratch() { for (int i = 0; i < n; ++i) (void)hipFree(p[i]); }
};

// the geometry for rows of equal length and no two entries of a row in one slice (AUTO's estimate; the build cuts the row blocks
// by the products the rows really have)
void tiled_geometry(size_t n_rows, size_t n_cols, size_t nnz, int dtype, uint32_t *n_cb, uint32_t *R, uint32_t *n_rb) {
    const uint32_t kT3Slice = t3_slice(dtype);
    const uint64_t cb = ((uint64_t)n_cols + kT3Slice - 1) / kT3Slice;
    *n_cb = (uint32_t)(cb ? cb : 1);
    const double per_row_and_slice = n_rows ? (double)nnz / (double)n_rows / (double)*n_cb : 0.0;
    const uint32_t cap = t3_cap_rows(dtype);
    double r = per_row_and_slice > 0.0 ? t3_tile_target(dtype) / per_row_and_slice : (double)cap;
    if (r > (double)cap) r = (double)cap;
    if (r < 1.0) r = 1.0;
    *R = (uint32_t)r;
    const uint64_t rb = ((uint64_t)n_rows + *R - 1) / *R;
    *n_rb = (uint32_t)(rb ? rb : 1);
}

template <typename T>
static int build_t(::smh_crs *m) {
    hipStream_t s = m->stream;
    const uint64_t nnz = m->nnz;
    constexpr uint32_t CH = t3_chunk<T>(), STRIDE = t3_stride<T>(), kT3Slice = T3<T>::kSlice;
    constexpr int kT3ExpandThreads = T3<T>::kThreads;
    const uint64_t n_cb64 = ((uint64_t)m->n_cols + kT3Slice - 1) / kT3Slice;
    const uint32_t n_cb = (uint32_t)(n_cb64 ? n_cb64 : 1);
    if (nnz + nnz / 64 + 1024 >= (1ull << 32)) return fail(SMH_ERR_INVALID, "tiled variant: %llu entries are too many for its 32-bit product index", (unsigned long long)nnz);
    // SMH_TILED_BUILD_TRACE=1 (development aid): the stages' wall times, each behind a stream synchronisation, on stderr
    static const bool trace = getenv("SMH_TILED_BUILD_TRACE") && atoi(getenv("SMH_TILED_BUILD_TRACE")) != 0;
    auto t_last = std::chrono::steady_clock::now();
    auto stage = [&](const char *what) {
        if (!trace) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[k2t build] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    T3Scratch tmp;
    uint32_t *rows_e = nullptr, *key_s = nullptr;
    T3Entry<T> *ps = nullptr;  // the entries sorted by slice (inside a slice: row, storage order)
    uint64_t *d_start = nullptr;
    SMH_TRY(tmp.alloc(&rows_e, nnz));
    SMH_TRY(tmp.alloc(&key_s, nnz));
    SMH_TRY(tmp.alloc(&ps, nnz));
    SMH_TRY(tmp.alloc(&d_start, (size_t)n_cb + 1));
    const unsigned grid = 2048;
    if (nnz) {
        SMH_TRY(expand_rows(m->d_off, m->n_rows, rows_e, s));
        stage("scratch + rows of entries");
        // keys and payload are READ through iterators over the CRS arrays (nothing is materialised before the first pass)
        auto keys_in = rocprim::make_transform_iterator(rocprim::counting_iterator<uint64_t>(0), T3LoadKey{m->d_col, kT3Slice});
        auto vals_in = rocprim::make_transform_iterator(rocprim::counting_iterator<uint64_t>(0), T3LoadEntry<T>{rows_e, m->d_col, (const T *)m->d_val});
        size_t bytes = 0;
        void *ws = nullptr;
        SMH_HIP(rocprim::radix_sort_pairs(ws, bytes, keys_in, key_s, vals_in, ps, (size_t)nnz, 0u, t3_bits_for(n_cb - 1), s));
        SMH_HIP(hipMalloc(&ws, bytes ? bytes : 16));
        const hipError_t e1 = rocprim::radix_sort_pairs(ws, bytes, keys_in, key_s, vals_in, ps, (size_t)nnz, 0u, t3_bits_for(n_cb - 1), s);
        const hipError_t e2 = hipStreamSynchronize(s);
        (void)hipFree(ws);
        SMH_HIP(e1);
        SMH_HIP(e2);
        stage("radix sort (entries as payload)");
    }
    hipLaunchKernelGGL(k_t3_bounds, dim3((n_cb + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, key_s, nnz, n_cb, d_start);
    SMH_HIP(hipGetLastError());
    std::vector<uint64_t> start((size_t)n_cb + 1);
    SMH_HIP(hipMemcpyAsync(start.data(), d_start, start.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    SMH_HIP(hipStreamSynchronize(s));
    stage("slice bounds");
    // chunks: ceil(entries of the slice / stride) each (a chunk whose start was moved past the slice's end stays empty)
    std::vector<uint32_t> cptr((size_t)n_cb + 1);
    cptr[0] = 0;
    for (uint32_t b = 0; b < n_cb; ++b) {
        const uint64_t cnt = start[b + 1] - start[b];
        if (cnt >= (1ull << 32)) return fail(SMH_ERR_INVALID, "tiled variant: a column slice holds %llu entries", (unsigned long long)cnt);
        cptr[b + 1] = cptr[b] + (uint32_t)((cnt + STRIDE - 1) / STRIDE);
    }
    const uint32_t n_chunks = cptr[n_cb];
    uint32_t max_slice_chunks = 0;
    for (uint32_t b = 0; b < n_cb; ++b) max_slice_chunks = std::max(max_slice_chunks, cptr[b + 1] - cptr[b]);
    const uint64_t slots = (uint64_t)n_chunks * CH;
    uint32_t *cstart = nullptr, *clen = nullptr, *obase = nullptr, *preal = nullptr, *gcount = nullptr, *prow = nullptr;
    uint32_t group = kT3RowGroup;
    if (const char *e = getenv("SMH_TILED_GROUP")) { const int v = atoi(e); if (v >= 1 && v <= 1024) group = (uint32_t)v; }  // tuning knob (1: a count per row, round 3's cut)
    const size_t n_groups = (m->n_rows + group - 1) / group;
    SMH_TRY(tmp.alloc(&cstart, (size_t)n_chunks));
    SMH_TRY(tmp.alloc(&clen, (size_t)n_chunks));
    SMH_TRY(tmp.alloc(&obase, (size_t)n_chunks + 1));
    SMH_TRY(tmp.alloc(&preal, (size_t)n_chunks));
    SMH_TRY(tmp.alloc(&gcount, n_groups));
    uint32_t *any_cut = nullptr;  // does any chunk boundary cut a (row, slice) pair?
    SMH_TRY(tmp.alloc(&any_cut, 1));
    SMH_HIP(hipMemsetAsync(any_cut, 0, sizeof(uint32_t), s));
    SMH_HIP(hipMalloc(&m->d_t2_val, (slots + CH) * sizeof(T)));
    SMH_HIP(hipMalloc((void **)&m->d_t2_code, (slots + CH) * sizeof(uint16_t)));
    SMH_HIP(hipMalloc((void **)&m->d_t3_cptr, ((size_t)n_cb + 1) * sizeof(uint32_t)));
    SMH_HIP(hipMalloc((void **)&m->d_t3_chunk, ((size_t)n_chunks + 1) * sizeof(T3Chunk)));
    SMH_HIP(hipMemcpyAsync(m->d_t3_cptr, cptr.data(), cptr.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    SMH_HIP(hipMemsetAsync(gcount, 0, (n_groups ? n_groups : 1) * sizeof(uint32_t), s));
    SMH_HIP(hipMemsetAsync(obase, 0, ((size_t)n_chunks + 1) * sizeof(uint32_t), s));
    const unsigned wgrid = (n_chunks + kBlock / 64 - 1) / (kBlock / 64);
    if (n_chunks) {
        hipLaunchKernelGGL(k_t3_chunk_starts<T>, dim3((n_chunks + kBlock - 1) / kBlock), dim3(kBlock), 0, s, m->d_t3_cptr, n_cb, n_chunks, d_start, ps, STRIDE, cstart);
        SMH_HIP(hipGetLastError());
    stage("allocs + chunk starts");
        hipLaunchKernelGGL(k_t3_fill<T>, dim3(wgrid), dim3(kBlock), 0, s, m->d_t3_cptr, n_cb, n_chunks, d_start, cstart, ps,
                           (T *)m->d_t2_val, m->d_t2_code, clen, obase);
        SMH_HIP(hipGetLastError());
    stage("fill");
    }
    uint64_t n_prod = 0;
    SMH_TRY(device_exclusive_scan_u32(obase, (uint64_t)n_chunks + 1, s, &n_prod));  // obase[c] = products before chunk c; obase[n_chunks] = all
    stage("scan");
    if (n_prod >= (1ull << 32) - 4 * CH) return fail(SMH_ERR_INVALID, "tiled variant: %llu products are too many for its 32-bit index", (unsigned long long)n_prod);
    SMH_TRY(tmp.alloc(&prow, (size_t)n_prod));
    if (n_chunks) {
        hipLaunchKernelGGL(k_t3_prow<T>, dim3(wgrid), dim3(kBlock), 0, s, m->d_t3_cptr, n_cb, n_chunks, d_start, cstart, clen, obase, ps, m->d_t2_code, prow, preal,
                           gcount, group, (T3Chunk *)m->d_t3_chunk, any_cut);
        SMH_HIP(hipGetLastError());
    stage("prow");
    }
    uint32_t h_cut = 0;
    SMH_HIP(hipMemcpyAsync(&h_cut, any_cut, sizeof h_cut, hipMemcpyDeviceToHost, s));  // (synchronised with the row counts below)
    // row blocks of equal PRODUCT counts (a tile = one slice of a block: ~target products whatever the row lengths), at most `cap`
    // rows each (their sums share the LDS); greedy on the host over the product counts of GROUPS of 32 rows -- 1.25 MB to fetch for
    // 10 M rows, where round 3 fetched a count per row and walked them all (15 of the build's 54 ms).  Row blocks therefore begin at
    // multiples of 32 rows; which rows share a block changes nothing in any row's sum.
    std::vector<uint32_t> rb_start;
    uint32_t n_rb = 0, R = 1;
    {
        const uint32_t cap_g = std::max(1u, t3_cap_rows(m->dtype) / group);
        const uint64_t per_block = (uint64_t)(t3_tile_target(m->dtype) * (double)n_cb);
        std::vector<uint32_t> h_cnt(n_groups);
        if (n_groups) SMH_HIP(hipMemcpyAsync(h_cnt.data(), gcount, h_cnt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        SMH_HIP(hipStreamSynchronize(s));
        auto cut = [&](uint64_t target, std::vector<uint32_t> &out) {
            out.clear();
            size_t r = 0;
            while (r < n_groups) {
                out.push_back((uint32_t)(r * group));
                uint64_t have = 0;
                size_t e = r;
                while (e < n_groups && e - r < cap_g && (e == r || have + h_cnt[e] <= target)) have += h_cnt[e++];
                r = e;
            }
        };
        cut(per_block, rb_start);
        // Pass 2 runs one wavefront per row block, W of them per workgroup, as many workgroups at once as the LDS of the CUs holds;
        // a wavefront's walk over its tiles is bound by latency, so a last round with a handful of row blocks lasts as long as a
        // full one: BASELINE C3 with 6145 row blocks (two rounds of 3072 and ONE block) ran 1.57 ms where 6122-6144 blocks run
        // 1.44 (profiles/r04_k2t_row_block_rounds.log).  When the last round would be less than a quarter full and slightly larger
        // tiles (up to +30 % products) make it unnecessary, the blocks are cut for one round less.
        {
            hipDeviceProp_t prop;
            const uint64_t cus = hipGetDeviceProperties(&prop, m->device) == hipSuccess && prop.multiProcessorCount > 0 ? (uint64_t)prop.multiProcessorCount : 256;
            const uint64_t stride = (((uint64_t)t3_cap_rows(m->dtype) + 1) * sizeof(T) + 15) & ~(uint64_t)15;
            const uint64_t slots = cus * std::max<uint64_t>(1, std::min<uint64_t>(16, ((uint64_t)160 << 10) / stride));  // row blocks per round
            const uint64_t n0 = rb_start.size();
            const uint64_t q = (n0 + slots - 1) / slots;
            static const bool rounds_off = getenv("SMH_TILED_ROUNDS") && atoi(getenv("SMH_TILED_ROUNDS")) == 0;  // tuning knob
            if (!rounds_off && q >= 2 && n0 - (q - 1) * slots < slots / 4) {
                uint64_t lo = per_block, hi = per_block + per_block * 3 / 10;  // smallest target in (lo, hi] whose cut fits q - 1 rounds
                const uint64_t goal = (q - 1) * slots - slots / 128;  // (a little air: rounds filled to the last slot ran 2 % slower than 99 % full ones)
                std::vector<uint32_t> trial;
                cut(hi, trial);
                if (trial.size() <= goal) {
                    while (hi - lo > 1) {
                        const uint64_t mid = lo + (hi - lo) / 2;
                        cut(mid, trial);
                        if (trial.size() <= goal) hi = mid; else lo = mid;
                    }
                    cut(hi, rb_start);
                }
            }
        }
        if (rb_start.empty()) rb_start.push_back(0);
        rb_start.push_back((uint32_t)m->n_rows);
        n_rb = (uint32_t)(rb_start.size() - 1);
        for (uint32_t b = 0; b < n_rb; ++b) R = std::max(R, rb_start[b + 1] - rb_start[b]);
    }
    stage("group counts to host + greedy");
    const uint64_t table_entries = (uint64_t)(n_rb + 1) * n_cb;
    if (table_entries * 4 > (4ull << 30))
        return fail(SMH_ERR_INVALID, "tiled variant: %u column slices x %u row blocks need a tile table beyond 4 GiB", n_cb, n_rb);
    // (+ a round of slack: a tile's loads cover whole rounds whatever its length)
    SMH_HIP(hipMalloc(&m->d_t2_prod, (n_prod + 2 * CH) * sizeof(T)));
    SMH_HIP(hipMalloc((void **)&m->d_t2_row, (n_prod + 2 * CH) * sizeof(uint16_t)));
    SMH_HIP(hipMalloc((void **)&m->d_t2_tstart, table_entries * sizeof(uint32_t)));
    SMH_HIP(hipMalloc((void **)&m->d_t2_rbstart, rb_start.size() * sizeof(uint32_t)));
    SMH_HIP(hipMemcpyAsync(m->d_t2_rbstart, rb_start.data(), rb_start.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    SMH_HIP(hipMemsetAsync(m->d_t2_prod, 0, (n_prod + 2 * CH) * sizeof(T), s));
    SMH_HIP(hipMemsetAsync(m->d_t2_row, 0, (n_prod + 2 * CH) * sizeof(uint16_t), s));  // (slack: the dump slot)
    if (n_chunks) {
        hipLaunchKernelGGL(k_t3_rowcode, dim3(wgrid), dim3(kBlock), 0, s, n_chunks, obase, preal, prow, m->d_t2_rbstart, n_rb, (uint32_t)sizeof(T), m->d_t2_row);
        SMH_HIP(hipGetLastError());
    stage("allocs + rowcode");
    }
    // (obase of a slice's first chunk = where its products begin; empty slices have none)
    hipLaunchKernelGGL(k_t3_table, dim3(grid), dim3(kBlock), 0, s, prow, m->d_t3_cptr, obase, n_cb, n_rb, m->d_t2_rbstart, m->d_t2_tstart);
    SMH_HIP(hipGetLastError());
    SMH_HIP(hipStreamSynchronize(s));
    stage("table");
    // 128 KiB and more of dynamic LDS need the attribute on every device the kernel runs on: set with each build, on the matrix's device
    for (const void *f : {reinterpret_cast<const void *>(k_t3_expand<T, 2>), reinterpret_cast<const void *>(k_t3_expand<T, 3>), reinterpret_cast<const void *>(k_t3_expand<T, 4>)})
        SMH_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((kT3Slice + (kT3ExpandThreads / 64) * CH) * sizeof(T))));
    // (pass 2: several wavefronts' sums per workgroup)
    for (const void *f : {reinterpret_cast<const void *>(k_t3_reduce<T, false, 2>), reinterpret_cast<const void *>(k_t3_reduce<T, false, 4>),
                          reinterpret_cast<const void *>(k_t3_reduce<T, false, 8>), reinterpret_cast<const void *>(k_t3_reduce<T, true, 2>),
                          reinterpret_cast<const void *>(k_t3_reduce<T, true, 4>), reinterpret_cast<const void *>(k_t3_reduce<T, true, 8>)})
        SMH_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10));
    // (the row codes are byte offsets into a wavefront's sums and must fit 16 bits)
    if (((uint64_t)R + 1) * sizeof(T) > 0xFFFFu) return fail(SMH_ERR_INVALID, "tiled variant: row blocks of %u rows do not fit the 16-bit row codes", R);
    m->t2_n_cb = n_cb;
    m->t2_n_rb = n_rb;
    m->t2_R = R;
    m->t2_tot = slots;
    m->t3_n_chunks = n_chunks;
    m->t3_max_slice_chunks = max_slice_chunks;
    m->t3_n_prod = n_prod;
    m->t3_dups = h_cut != 0;
    return SMH_OK;
}

uint32_t tiled_slice_columns(int dtype) { return t3_slice(dtype); }

void tiled_free(::smh_crs *m) {
    (void)hipFree(m->d_t2_val); (void)hipFree(m->d_t2_prod); (void)hipFree(m->d_t2_code); (void)hipFree(m->d_t2_row);
    (void)hipFree(m->d_t3_cptr); (void)hipFree(m->d_t3_chunk); (void)hipFree(m->d_t2_tstart); (void)hipFree(m->d_t2_rbstart);
    m->d_t2_val = m->d_t2_prod = nullptr;
    m->d_t2_code = m->d_t2_row = nullptr;
    m->d_t3_cptr = nullptr;
    m->d_t3_chunk = nullptr;
    m->d_t2_tstart = m->d_t2_rbstart = nullptr;
    m->t2_built = m->t2_ok = false;
}

int tiled_build(::smh_crs *m) {
    if (m->t2_built) return m->t2_ok ? SMH_OK : fail(SMH_ERR_INVALID, "the tiled copy could not be built for this matrix");
    SMH_TRY(columns_within_n_cols(m, "tiled variant"));  // (the slice tables are sized from n_cols)
    m->t2_built = true;
    const int rc = m->dtype == SMH_F64 ? build_t<double>(m) : build_t<float>(m);
    if (rc != SMH_OK) {
        const std::string keep = smh_last_error();
        tiled_free(m);
        m->t2_built = true;  // do not try again
        return fail(rc, "%s", keep.c_str());
    }
    m->t2_ok = true;
    return SMH_OK;
}

// the plan's integer structure (and the copy) for inspection: `which` as in include/sparsemat_hip.h (smh_crs_tiled_array)
int tiled_array(::smh_crs *m, int which, void *out, size_t capacity_bytes, size_t *bytes_out) {
    const size_t vs = dtype_size(m->dtype), chunk_slots = m->dtype == SMH_F64 ? t3_chunk<double>() : t3_chunk<float>();
    const void *src = nullptr;
    size_t bytes = 0;
    switch (which) {
        case 0: src = m->d_t3_cptr; bytes = ((size_t)m->t2_n_cb + 1) * 4; break;
        case 1: src = m->d_t3_chunk; bytes = (size_t)m->t3_n_chunks * sizeof(T3Chunk); break;
        case 2: src = m->d_t2_code; bytes = (size_t)m->t3_n_chunks * chunk_slots * 2; break;
        case 3: src = m->d_t2_val; bytes = (size_t)m->t3_n_chunks * chunk_slots * vs; break;
        case 4: src = m->d_t2_row; bytes = (size_t)m->t3_n_prod * 2; break;
        case 5: src = m->d_t2_rbstart; bytes = ((size_t)m->t2_n_rb + 1) * 4; break;
        case 6: src = m->d_t2_tstart; bytes = ((size_t)m->t2_n_rb + 1) * m->t2_n_cb * 4; break;
        case 7: src = m->d_t2_prod; bytes = (size_t)m->t3_n_prod * vs; break;
        default: return fail(SMH_ERR_INVALID, "smh_crs_tiled_array: unknown array %d", which);
    }
    if (bytes_out) *bytes_out = bytes;
    if (!out) return SMH_OK;
    if (capacity_bytes < bytes) return fail(SMH_ERR_INVALID, "smh_crs_tiled_array: %zu bytes needed, %zu given", bytes, capacity_bytes);
    if (bytes) {
        SMH_HIP(hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, m->stream));
        SMH_HIP(hipStreamSynchronize(m->stream));
    }
    return SMH_OK;
}

template <typename T>
static int launch_t(::smh_crs *m, const void *x, size_t x_len, void *y, hipStream_t s) {
    constexpr uint32_t CH = t3_chunk<T>(), kT3Slice = T3<T>::kSlice;
    constexpr int kT3ExpandThreads = T3<T>::kThreads;
    const size_t lds1 = ((size_t)kT3Slice + (kT3ExpandThreads / 64) * CH) * sizeof(T), lds2 = ((size_t)m->t2_R + 1) * sizeof(T);
    static const uint32_t xcd_map = getenv("SMH_TILED_XCD") ? (uint32_t)atoi(getenv("SMH_TILED_XCD")) : 3u;  // tuning knob: bit 0 pass 1, bit 1 pass 2
    if (m->t3_n_chunks) {
        // a workgroup pays for staging its slice of x (16384 entries), so it should fold several times as many entries: ~65 000
        // (round 2's measurement, profiles/r02_tiled_pass1.log) = 16 chunks per wavefront
        const uint64_t per_slice = (uint64_t)m->t3_n_chunks * CH / (m->t2_n_cb ? m->t2_n_cb : 1);
        uint32_t parts = (uint32_t)((per_slice + 32768) / 65536);
        parts = parts < 1 ? 1 : (parts > 64 ? 64 : parts);
        // (a wavefront keeps the descriptors of its chunks in one register: at most 64 chunks each, 1024 per 16-wavefront workgroup)
        constexpr uint32_t per_wg = 64u * (uint32_t)(kT3ExpandThreads / 64);
        const uint32_t need = (m->t3_max_slice_chunks + per_wg - 1u) / per_wg;
        if (parts < need) parts = need;
        static const int ahead = getenv("SMH_TILED_AHEAD") ? atoi(getenv("SMH_TILED_AHEAD")) : kT3Ahead;  // tuning knob: chunks in flight per wavefront
        auto *exp = ahead == 2 ? k_t3_expand<T, 2> : ahead == 4 ? k_t3_expand<T, 4> : k_t3_expand<T, 3>;
        // persistent workgroups: as many as the device holds at once (a multiple of 8: every XCD the same number), never more than items
        static int held_cache[64][2] = {};  // per device and value type; (racing threads compute the same value)
        int &held = held_cache[m->device & 63][sizeof(T) == 8];
        if (held == 0) {
            int per_cu = 0;
            SMH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, exp, kT3ExpandThreads, lds1));
            hipDeviceProp_t prop;
            const int cus = hipGetDeviceProperties(&prop, m->device) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            held = (per_cu < 1 ? 1 : per_cu) * cus;
        }
        uint32_t wgs = (uint32_t)held & ~7u;
        if (wgs < 8) wgs = 8;
        // the items are dealt out statically, so the pass lasts as long as the workgroup with the most of them: among the part counts
        // near the one above take the one whose items divide most evenly (611 slices x 8 parts over 512 workgroups: 10 against 9.5 on
        // average, 5 % lost; x 10 parts: 12 against 11.9 -- measured 572 against 586-601 us on f32, 1015-1044 against 1051 on f64)
        auto longest = [&](uint32_t c) {  // items of the busiest workgroup
            const uint64_t items = (uint64_t)m->t2_n_cb * c;
            return (xcd_map & 1u) ? ((items + 7) / 8 + wgs / 8 - 1) / (wgs / 8) : (items + wgs - 1) / wgs;
        };
        {
            const uint32_t lo = std::max(need, std::max(1u, parts * 5 / 8)), hi = std::min(64u, parts + parts / 2);
            uint32_t best = parts;
            for (uint32_t c = lo; c <= hi; ++c)  // cost = longest(c) / c, compared as cross products; ties: the nearer to the target
                if (longest(c) * best < longest(best) * c ||
                    (longest(c) * best == longest(best) * c && (c > parts ? c - parts : parts - c) < (best > parts ? best - parts : parts - best)))
                    best = c;
            parts = best;
        }
        if (const char *e = getenv("SMH_TILED_PARTS")) { const int v = atoi(e); if (v >= (int)need && v <= 64) parts = (uint32_t)v; }  // tuning knob
        const uint32_t g1 = m->t2_n_cb * parts;  // (slice, part) items
        const uint32_t per_xcd = (xcd_map & 1u) ? (g1 + 7u) / 8u : 0u;
        if ((xcd_map & 1u) ? wgs / 8u > per_xcd : wgs > g1) wgs = (xcd_map & 1u) ? (per_xcd ? per_xcd * 8u : 8u) : (g1 ? g1 : 1u);
        hipLaunchKernelGGL(exp, dim3(wgs), dim3(kT3ExpandThreads), lds1, s, (const T *)x, (uint64_t)x_len, (const T *)m->d_t2_val, m->d_t2_code,
                           m->d_t3_cptr, (const T3Chunk *)m->d_t3_chunk, (T *)m->d_t2_prod, parts, g1, per_xcd);
        SMH_HIP(hipGetLastError());
    }
    // pass 2: one wavefront per row block, and the wavefronts of ADJACENT row blocks share a workgroup and walk the slices in lock step
    // (a barrier per batch of tiles): their tiles are neighbours in memory, so the workgroup reads one piece of W x ~1 KiB per slice
    // instead of W pieces at W different times -- the pass is bound by exactly those reads (profiles/r03_k2t_pass2_bound_experiments.log;
    // W = 1 / 2 / 4 / 6 / 12: 472 / 429 / 406 / 395 / 368 us on C2-uniform, 536 / 502 / 477 / 471 / 451 on C3, r03_k2t_pass2_lockstep.log).
    // W = the largest count (<= 16: 1024 threads) that packs the CU's LDS and still leaves a workgroup for every CU
    static const int waves2_env = getenv("SMH_TILED_WAVES") ? atoi(getenv("SMH_TILED_WAVES")) : 0;  // tuning knob: 0 = automatic
    const uint32_t stride2 = (uint32_t)((lds2 + 15) & ~(size_t)15);
    static int cus_cache[64] = {};  // per device
    int &cus = cus_cache[m->device & 63];
    if (cus == 0) {
        hipDeviceProp_t prop;
        cus = hipGetDeviceProperties(&prop, m->device) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const uint32_t fit = std::max<uint32_t>(1u, (uint32_t)(((size_t)160 << 10) / stride2));  // wavefronts a CU's LDS holds sums for
    uint32_t waves2 = 1;
    for (uint32_t w = std::min(fit, 16u); w > 1; --w) {
        const bool packs = (uint64_t)(fit / w) * w * 10 >= (uint64_t)fit * 9;                      // workgroups of w leave at most a tenth of that unused
        const bool spreads = ((uint64_t)m->t2_n_rb + w - 1) / w >= (uint64_t)cus * 9 / 10;          // ... and there is a workgroup for (nearly) every CU
        if (packs && spreads) { waves2 = w; break; }
    }
    if (waves2_env >= 1 && (uint32_t)waves2_env <= std::min(fit, 16u)) waves2 = (uint32_t)waves2_env;
    const uint32_t g2 = (m->t2_n_rb + waves2 - 1u) / waves2, g2r = (xcd_map & 2u) ? (g2 + 7u) & ~7u : g2;
    static const int batch = getenv("SMH_TILED_BATCH") ? atoi(getenv("SMH_TILED_BATCH")) : kT3Batch;  // tuning knob: 2, 4 or 8 tiles per batch
    auto *red = batch == 8 ? (m->t3_dups ? k_t3_reduce<T, true, 8> : k_t3_reduce<T, false, 8>)
                : batch == 2 ? (m->t3_dups ? k_t3_reduce<T, true, 2> : k_t3_reduce<T, false, 2>)
                             : (m->t3_dups ? k_t3_reduce<T, true, 4> : k_t3_reduce<T, false, 4>);
    hipLaunchKernelGGL(red, dim3(g2r), dim3(64 * waves2), (size_t)waves2 * stride2, s, (const T *)m->d_t2_prod, m->d_t2_row, m->d_t2_tstart, m->t2_n_cb, m->t2_n_rb, m->d_t2_rbstart,
                       (T *)y, xcd_map >> 1 & 1u, stride2);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

int launch_spmv_tiled(::smh_crs *m, const void *x, size_t x_len, void *y, hipStream_t s) {
    if (m->n_rows == 0) return SMH_OK;
    return m->dtype == SMH_F64 ? launch_t<double>(m, x, x_len, y, s) : launch_t<float>(m, x, x_len, y, s);
}

}  // namespace smh
