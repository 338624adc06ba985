// spmv_tiled.hip -- K2t: y = A x in two streaming passes over a 2-D tiled copy of A, for matrices whose columns have no
// locality (BASELINE C2 "uniform", C3).  Replaces the reference loop sparsematrix.rs:146-158 for those matrices; the
// sum of a row is taken slice by slice (ascending column slices, storage order within a slice), so the result agrees
// with the reference within the rounding bound of DESIGN.md section 2, not bit for bit -- like K2c / K2f.
//
// Why: every other kernel family gathers x[col] through the vector L1, and a gather that misses it costs one cache line
// and one L2 round trip -- 150-190 G gathers/s however the work is arranged ("gather wall").  Here no
// gather leaves the CU:
//   pass 1 "expand":  the entries are stored by column slice (C = 16384 columns); a workgroup stages its slice of x in
//                     LDS and computes prod[i] = val[i] * x_lds[code[i]] for its part of the slice's entries -- a pure
//                     streaming map (16-bit column codes; 16-byte loads and stores), the gathers hit LDS.
//   pass 2 "reduce":  within a slice the entries are ordered by row, so the entries of a block of consecutive rows form
//                     one contiguous TILE per slice.  The row blocks are cut so that each holds the same number of entries
//                     (~48 per tile: one entry per lane; skewed matrices get short blocks around their long rows).  One
//                     wavefront per row block walks its tiles slice by slice, adds runs of equal rows with two
//                     ballots and one-lane DPP shifts, and accumulates into R wave-private sums in LDS: no atomics,
//                     no barriers, a fixed order -- bitwise reproducible.  The loads of the next 8 tiles are in flight
//                     while 8 are folded.
// Traffic per entry: pass 1 reads sizeof(T)+2 and writes sizeof(T), pass 2 reads sizeof(T)+2 -- 16 B (f32) / 28 B (f64)
// against CSR's 8 / 12, all of it streaming.  Memory: the copy (sizeof(T)+4 per entry) plus the product buffer.
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <vector>

#include "internal.hpp"

namespace smh {

constexpr uint32_t kT2Slice = 16384;   // columns per slice: 64 KiB (f32) / 128 KiB (f64) of x in LDS
constexpr int kT2ExpandThreads = 1024;
constexpr int kT2Waves = 4;            // wavefronts (row blocks) per workgroup of the reduce pass
constexpr int kT2Batch = 8;            // tiles whose loads are in flight together
constexpr double kT2TileTarget = 48.0; // mean entries per tile (one per lane; a longer tile takes a slow second round)
// tuning knob SMH_TILED_TILE: another target (16..64)
static double t2_tile_target() {
    if (const char *e = getenv("SMH_TILED_TILE")) {
        const double v = atof(e);
        if (v >= 16.0 && v <= 64.0) return v;
    }
    return kT2TileTarget;
}

typedef float t2_f4 __attribute__((ext_vector_type(4)));
typedef double t2_d2 __attribute__((ext_vector_type(2)));
typedef uint32_t t2_u2 __attribute__((ext_vector_type(2)));
// one lane's 16 bytes of values and the 16-bit codes that go with them
template <typename T> struct T2Lane;
template <> struct T2Lane<float> {
    using V = t2_f4;
    using C = t2_u2;
    static constexpr int kEntries = 4, kUnroll = 2;
    static __device__ __forceinline__ V mul(V v, C c, const float *xs) {
        V p;
        p.x = v.x * xs[c.x & 0xFFFF]; p.y = v.y * xs[c.x >> 16];
        p.z = v.z * xs[c.y & 0xFFFF]; p.w = v.w * xs[c.y >> 16];
        return p;
    }
};
template <> struct T2Lane<double> {
    using V = t2_d2;
    using C = uint32_t;
    static constexpr int kEntries = 2, kUnroll = 2;
    static __device__ __forceinline__ V mul(V v, C c, const double *xs) {
        V p;
        p.x = v.x * xs[c & 0xFFFF]; p.y = v.y * xs[c >> 16];
        return p;
    }
};

// ---- pass 1 ---------------------------------------------------------------------------------------------------------
// grid = slices * parts; cb_ptr[b] = first entry of slice b in the copy (multiples of 8 entries: segments are padded with
// zero-valued entries of code 0).  Every load and store instruction of a wavefront covers ONE contiguous kilobyte (a lane
// takes 16 bytes of values -- 4 f32 / 2 f64 entries -- and their codes, the next lane the next 16; two such pieces in flight): with a lane owning 8
// consecutive entries instead, a 16-byte load used a quarter (f64) or half (f32) of every line it touched and the read
// side alone ran at 3.6 TB/s on f64 (profiles/r02_t2d_probe.log)
template <typename T, int U>
__global__ __launch_bounds__(kT2ExpandThreads) void k_t2_expand(const T *__restrict__ x, uint64_t x_len, const T *__restrict__ val,
                                                                 const uint16_t *__restrict__ code, const uint64_t *__restrict__ cb_ptr,
                                                                 T *__restrict__ prod, uint32_t parts, uint32_t n_items, uint32_t xcd_map) {
    extern __shared__ __attribute__((aligned(16))) char t2_smem[];
    T *xs = (T *)t2_smem;
    // workgroups b, b + 8, ... share an XCD (round-robin dispatch): give an XCD a contiguous run of (slice, part) pairs, so
    // that the parts of a slice stage its x from one L2 instead of fetching it into eight (a speed hint only)
    uint32_t g = blockIdx.x;
    if (xcd_map) {
        const uint32_t per = (gridDim.x + 7u) / 8u;
        g = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    }
    if (g >= n_items) return;  // (the remapped grid is rounded up to a multiple of 8; whole workgroup, before any barrier)
    const uint32_t cb = g / parts, part = g % parts;
    const uint64_t c0 = (uint64_t)cb * kT2Slice;
    for (uint32_t i = threadIdx.x; i < kT2Slice; i += kT2ExpandThreads) xs[i] = c0 + i < x_len ? x[c0 + i] : T(0);
    __syncthreads();
    using L = T2Lane<T>;
    using V = typename L::V;
    using C = typename L::C;
    constexpr int E = L::kEntries;
    const uint64_t a0 = cb_ptr[cb], a1 = cb_ptr[cb + 1];
    const uint64_t chunks = (a1 - a0) / E;  // 16-byte pieces of the slice's values
    const uint64_t per = (chunks + parts - 1) / parts;
    const uint64_t k0 = (uint64_t)part * per, k1 = k0 + per < chunks ? k0 + per : chunks;
    const V *vv = (const V *)(val + a0);
    const C *cc = (const C *)(code + a0);
    V *pp = (V *)(prod + a0);
    for (uint64_t k = k0 + threadIdx.x; k < k1; k += (uint64_t)kT2ExpandThreads * U) {
        V v[U];
        C c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {  // unconditional loads from a clamped index; the store decides
            const uint64_t q = k + (uint64_t)u * kT2ExpandThreads;
            const uint64_t qe = q < k1 ? q : k1 - 1;
            v[u] = __builtin_nontemporal_load(vv + qe);
            c[u] = __builtin_nontemporal_load(cc + qe);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t q = k + (uint64_t)u * kT2ExpandThreads;
            // plain stores: non-temporal ones cost 30 % here on f64 (profiles/r02_t2d_probe.log)
            if (q < k1) pp[q] = L::mul(v[u], c[u], xs);
        }
    }
}

// ---- pass 2 ---------------------------------------------------------------------------------------------------------
__device__ inline float t2_shl1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xF, 0xF, false)); }
__device__ inline double t2_shl1(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x130, 0xF, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x130, 0xF, 0xF, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

template <typename T>
struct T2Batch {
    T pv[kT2Batch];
    uint32_t rv[kT2Batch], ln[kT2Batch];
    uint64_t bs[kT2Batch];
};

// tstart[rb * n_cb + cb] = first entry of tile (cb, rb) relative to cb_ptr[cb]; row n_rb of the table holds the ends of the
// last row block's tiles.  Row block rb = rows [rb_start[rb], rb_start[rb+1]); rowc: the entry's row relative to its block.
// LDS: kT2Waves * R sums.
template <typename T>
__global__ __launch_bounds__(kT2Waves * 64) void k_t2_reduce(const T *__restrict__ prod, const uint16_t *__restrict__ rowc,
                                                              const uint64_t *__restrict__ cb_ptr, const uint32_t *__restrict__ tstart, uint32_t n_cb,
                                                              uint32_t n_rb, const uint32_t *__restrict__ rb_start, uint32_t R, T *__restrict__ y, uint32_t xcd_map) {
    extern __shared__ __attribute__((aligned(16))) char t2_smem[];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // neighbouring row blocks' tiles share cache lines: keep neighbours on one XCD (one L2)
    uint32_t g = blockIdx.x;
    if (xcd_map) {
        const uint32_t per = (gridDim.x + 7u) / 8u;
        g = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    }
    const uint32_t rb = g * kT2Waves + w;
    if (rb >= n_rb) return;  // whole wavefronts; no barrier below
    T *acc = (T *)t2_smem + (size_t)w * R;  // R = the largest row block
    const uint32_t r0 = rb_start[rb], rows = rb_start[rb + 1] - r0;
    for (uint32_t i = lane; i < rows; i += 64) acc[i] = T(0);
    const uint32_t *ts0 = tstart + (size_t)rb * n_cb, *ts1 = ts0 + n_cb;
    auto table = [&](uint32_t g, uint64_t &base, uint32_t &len) {  // lane l: tile g + l
        const uint32_t cbl = g + lane;
        base = 0;
        len = 0;
        if (cbl < n_cb) {
            const uint32_t s = ts0[cbl], e = ts1[cbl];
            base = cb_ptr[cbl] + s;
            len = e - s;
        }
    };
    uint64_t cur_base, nxt_base;
    uint32_t cur_len, nxt_len, win = 0;  // cur_*: tiles [win, win + 64), nxt_*: the 64 after them
    table(0, cur_base, cur_len);
    table(64, nxt_base, nxt_len);
    auto issue = [&](T2Batch<T> &B, uint32_t j0) {
        if (j0 >= win + 64) {
            cur_base = nxt_base;
            cur_len = nxt_len;
            win += 64;
            table(win + 64, nxt_base, nxt_len);
        }
#pragma unroll
        for (int d = 0; d < kT2Batch; ++d) {
            const int j = (int)((j0 + d) & 63);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cur_base, j);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cur_base >> 32), j);
            B.bs[d] = (uint64_t)hi << 32 | lo;
            B.ln[d] = (uint32_t)__builtin_amdgcn_readlane((int)cur_len, j);  // 0 past the last slice
        }
#pragma unroll
        for (int d = 0; d < kT2Batch; ++d) {  // unconditional loads (a tile's first entry is always a valid address), masked when folded
            const uint32_t idx = lane < B.ln[d] ? lane : 0;
            B.pv[d] = prod[B.bs[d] + idx];
            B.rv[d] = rowc[B.bs[d] + idx];
        }
    };
    // one round = up to 64 consecutive entries of a tile, one per lane (p, r: this lane's product and row; r = ~0 past the end)
    auto round = [&](uint64_t bs, uint32_t len, uint32_t t0, T p, uint32_t r, uint32_t prev0) {
        const uint32_t t = t0 + lane;
        uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)r, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
        if (lane == 0) prev = prev0;
        const bool valid = t < len;
        const bool head = valid && prev != r;
        // run lengths from two ballots: m = heads whose run is longer than k; the k-th neighbour's product arrives by k one-lane shifts
        const uint64_t nh = __ballot(valid && !head);
        uint64_t m = __ballot(head);
        T s = p, q = p;
        for (uint32_t k = 1; k < 64; ++k) {  // (k = 64 would shift a 64-bit mask by its width)
            m &= nh >> k;
            if (!m) break;
            q = t2_shl1(q);
            if ((m >> lane) & 1) s += q;
        }
        const uint32_t r63 = (uint32_t)__builtin_amdgcn_readlane((int)r, 63);
        if (head) {
            const uint32_t nx = t0 + 64;  // the run may go on past this round (tiles of more than 64 entries only)
            if (r63 == r && nx < len)
                for (uint32_t k = nx; k < len && rowc[bs + k] == r; ++k) s += prod[bs + k];
            acc[r] += s;
        }
    };
    // the first round of every tile works on the prefetched registers and issues no load, so the wait in front of it can leave
    // the next batch's loads in flight (a load inside the common path would force vmcnt(0): they return in order)
    auto fold = [&](T2Batch<T> &B) {
#pragma unroll
        for (int d = 0; d < kT2Batch; ++d) {
            const uint32_t len = B.ln[d];
            const uint64_t bs = B.bs[d];
            round(bs, len, 0, lane < len ? B.pv[d] : T(0), lane < len ? B.rv[d] : 0xFFFFFFFFu, 0xFFFFFFFFu);
            for (uint32_t t0 = 64; t0 < len; t0 += 64) {  // rare: a tile of more than 64 entries
                const uint32_t t = t0 + lane;
                T p = T(0);
                uint32_t r = 0xFFFFFFFFu;
                if (t < len) { p = prod[bs + t]; r = rowc[bs + t]; }
                round(bs, len, t0, p, r, (uint32_t)rowc[bs + t0 - 1]);
            }
        }
    };
    T2Batch<T> A, B;
    issue(A, 0);
    for (uint32_t j0 = 0; j0 < n_cb; j0 += 2 * kT2Batch) {
        issue(B, j0 + kT2Batch);
        fold(A);
        issue(A, j0 + 2 * kT2Batch);
        fold(B);
    }
    for (uint32_t i = lane; i < rows; i += 64) y[(uint64_t)r0 + i] = acc[i];
}

// ---- plan -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_t2_keys(const uint32_t *__restrict__ col, uint64_t nnz, uint32_t *__restrict__ key, uint32_t *__restrict__ idx) {
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nnz; i += (uint64_t)gridDim.x * kBlock) {
        key[i] = col[i] / kT2Slice;
        idx[i] = (uint32_t)i;
    }
}

// start[b] = first position of the sorted keys holding a value >= b (b = 0 .. n_cb)
__global__ __launch_bounds__(kBlock) void k_t2_bounds(const uint32_t *__restrict__ key_s, uint64_t nnz, uint32_t n_cb, uint64_t *__restrict__ start) {
    const uint32_t b = blockIdx.x * kBlock + threadIdx.x;
    if (b > n_cb) return;
    uint64_t lo = 0, hi = nnz;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) / 2;
        if (key_s[mid] < b) lo = mid + 1; else hi = mid;
    }
    start[b] = lo;
}

// the copy in slice-major order: position of sorted rank q = cb_ptr[slice] + (q - start[slice])
template <typename T>
__global__ __launch_bounds__(kBlock) void k_t2_fill(const uint32_t *__restrict__ off, uint64_t n_rows, const uint32_t *__restrict__ col,
                                                     const T *__restrict__ val, const uint32_t *__restrict__ key_s, const uint32_t *__restrict__ perm,
                                                     uint64_t nnz, const uint64_t *__restrict__ start, const uint64_t *__restrict__ cb_ptr,
                                                     const uint32_t *__restrict__ rb_start, uint32_t n_rb,
                                                     T *__restrict__ val_a, uint16_t *__restrict__ code_a, uint16_t *__restrict__ row_a,
                                                     uint32_t *__restrict__ row_full) {
    for (uint64_t q = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q < nnz; q += (uint64_t)gridDim.x * kBlock) {
        const uint32_t cb = key_s[q], i = perm[q];
        const uint64_t p = cb_ptr[cb] + (q - start[cb]);
        uint64_t lo = 0, hi = n_rows;  // the row of entry i: the last one with off[row] <= i
        while (lo < hi) {
            const uint64_t mid = (lo + hi) / 2;
            if (off[mid + 1] <= i) lo = mid + 1; else hi = mid;
        }
        const uint32_t row = (uint32_t)lo;
        val_a[p] = val[i];
        code_a[p] = (uint16_t)(col[i] - cb * kT2Slice);
        uint32_t bl = 0, bh = n_rb;  // the row block: the last one with rb_start[b] <= row
        while (bl + 1 < bh) {
            const uint32_t mid = (bl + bh) / 2;
            if (rb_start[mid] <= row) bl = mid; else bh = mid;
        }
        row_a[p] = (uint16_t)(row - rb_start[bl]);
        row_full[p] = row;
    }
}

// tstart[rb * n_cb + cb], rb = 0 .. n_rb: first entry of slice cb (relative) whose row is >= rb_start[rb] (= n_rows for rb = n_rb)
__global__ __launch_bounds__(kBlock) void k_t2_table(const uint32_t *__restrict__ row_full, const uint64_t *__restrict__ start,
                                                      const uint64_t *__restrict__ cb_ptr, uint32_t n_cb, uint32_t n_rb,
                                                      const uint32_t *__restrict__ rb_start, uint32_t *__restrict__ tstart) {
    const uint64_t total = (uint64_t)(n_rb + 1) * n_cb;
    for (uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (uint64_t)gridDim.x * kBlock) {
        const uint32_t rb = (uint32_t)(t / n_cb), cb = (uint32_t)(t % n_cb);
        const uint64_t cnt = start[cb + 1] - start[cb], first_row = rb_start[rb];
        const uint32_t *seg = row_full + cb_ptr[cb];
        uint64_t lo = 0, hi = cnt;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) / 2;
            if (seg[mid] < first_row) lo = mid + 1; else hi = mid;
        }
        tstart[t] = (uint32_t)lo;
    }
}

static unsigned t2_bits_for(uint64_t v) {
    unsigned b = 1;
    while (b < 64 && (v >> b)) ++b;
    return b;
}

struct T2Scratch {
    void *p[8] = {};
    int n = 0;
    template <typename U> int alloc(U **out, size_t count) {
        SMH_HIP(hipMalloc((void **)out, (count ? count : 1) * sizeof(U)));
        p[n++] = *out;
        return SMH_OK;
    }
    ~T2Scratch() { for (int i = 0; i < n; ++i) (void)hipFree(p[i]); }
};

// the geometry for rows of equal length (AUTO's estimate; the build cuts the row blocks by entries, see row_blocks())
void tiled_v1_geometry(size_t n_rows, size_t n_cols, size_t nnz, int dtype, uint32_t *n_cb, uint32_t *R, uint32_t *n_rb) {
    const uint64_t cb = ((uint64_t)n_cols + kT2Slice - 1) / kT2Slice;
    *n_cb = (uint32_t)(cb ? cb : 1);
    // rows per block: a tile (one slice x one row block) should hold ~kT2TileTarget entries; the sums of kT2Waves blocks
    // share 48 KiB of LDS
    const double per_row_and_slice = n_rows ? (double)nnz / (double)n_rows / (double)*n_cb : 0.0;
    const uint32_t cap = dtype == SMH_F64 ? 1280u : 3072u;  // rows whose sums one wavefront keeps in LDS (see build_t)
    double r = per_row_and_slice > 0.0 ? t2_tile_target() / per_row_and_slice : (double)cap;
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
    uint32_t n_cb, R, n_rb;
    tiled_v1_geometry(m->n_rows, m->n_cols, m->nnz, m->dtype, &n_cb, &R, &n_rb);
    // row blocks of equal ENTRY counts (a tile = one slice of a block: ~kT2TileTarget entries whatever the row lengths), at most
    // `cap` rows each (their sums share the LDS); greedy over the row offsets, on the host
    std::vector<uint32_t> rb_start;
    {
        // f32: 3072 rows (12 KiB of sums per wavefront); f64: 1280 (10 KiB: 16 wavefronts per CU) -- C3 2.31 / 2.15 / 2.19 / 2.18 ms
        // with 1536 / 1280 / 1024 / 768, 10 M x 16 uniform 1.11 / 1.02 / 1.13 / 1.21 (profiles/r02_tiled_cap.log)
        uint32_t cap = sizeof(T) == 8 ? 1280u : 3072u;
        if (const char *e = getenv("SMH_TILED_CAP")) {  // tuning knob: most rows of a row block
            const int v = atoi(e);
            if (v >= 1 && v <= 8192) cap = (uint32_t)v;
        }
        const uint64_t per_block = (uint64_t)(t2_tile_target() * (double)n_cb);
        std::vector<uint32_t> h_off(m->n_rows + 1);
        SMH_HIP(hipMemcpyAsync(h_off.data(), m->d_off, h_off.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        SMH_HIP(hipStreamSynchronize(s));
        rb_start.reserve((size_t)n_rb + 16);
        size_t r = 0;
        while (r < m->n_rows) {
            rb_start.push_back((uint32_t)r);
            const size_t hi = r + cap < m->n_rows ? r + cap : m->n_rows;
            // the first row boundary in (r, hi] at which the block holds per_block entries or more
            const uint64_t want = (uint64_t)h_off[r] + per_block;
            size_t e = (size_t)(std::lower_bound(h_off.begin() + r + 1, h_off.begin() + hi + 1, want,
                                                 [](uint32_t a, uint64_t b) { return (uint64_t)a < b; }) - h_off.begin());
            if (e > hi) e = hi;
            r = e;
        }
        if (rb_start.empty()) rb_start.push_back(0);
        rb_start.push_back((uint32_t)m->n_rows);
        n_rb = (uint32_t)(rb_start.size() - 1);
        R = 1;
        for (uint32_t b = 0; b < n_rb; ++b) R = std::max(R, rb_start[b + 1] - rb_start[b]);
    }
    const uint64_t table_entries = (uint64_t)(n_rb + 1) * n_cb;
    if (table_entries * 4 > (4ull << 30))
        return fail(SMH_ERR_INVALID, "tiled variant: %u column slices x %u row blocks need a tile table beyond 4 GiB", n_cb, n_rb);
    T2Scratch tmp;
    uint32_t *key = nullptr, *key_s = nullptr, *idx = nullptr, *perm = nullptr, *row_full = nullptr;
    uint64_t *d_start = nullptr;
    SMH_TRY(tmp.alloc(&key, nnz));
    SMH_TRY(tmp.alloc(&key_s, nnz));
    SMH_TRY(tmp.alloc(&idx, nnz));
    SMH_TRY(tmp.alloc(&perm, nnz));
    SMH_TRY(tmp.alloc(&d_start, (size_t)n_cb + 1));
    const unsigned grid = 2048;
    if (nnz) {
        hipLaunchKernelGGL(k_t2_keys, dim3(grid), dim3(kBlock), 0, s, m->d_col, nnz, key, idx);
        SMH_HIP(hipGetLastError());
        {
            size_t bytes = 0;
            void *ws = nullptr;
            SMH_HIP(rocprim::radix_sort_pairs(ws, bytes, key, key_s, idx, perm, (size_t)nnz, 0u, t2_bits_for(n_cb - 1), s));
            SMH_HIP(hipMalloc(&ws, bytes ? bytes : 16));
            const hipError_t e1 = rocprim::radix_sort_pairs(ws, bytes, key, key_s, idx, perm, (size_t)nnz, 0u, t2_bits_for(n_cb - 1), s);
            const hipError_t e2 = hipStreamSynchronize(s);
            (void)hipFree(ws);
            SMH_HIP(e1);
            SMH_HIP(e2);
        }
    }
    hipLaunchKernelGGL(k_t2_bounds, dim3((n_cb + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, key_s, nnz, n_cb, d_start);
    SMH_HIP(hipGetLastError());
    std::vector<uint64_t> start((size_t)n_cb + 1), cb_ptr((size_t)n_cb + 1);
    SMH_HIP(hipMemcpyAsync(start.data(), d_start, start.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    SMH_HIP(hipStreamSynchronize(s));
    cb_ptr[0] = 0;
    for (uint32_t b = 0; b < n_cb; ++b) {
        const uint64_t cnt = start[b + 1] - start[b];
        if (cnt >= (1ull << 32)) return fail(SMH_ERR_INVALID, "tiled variant: a column slice holds %llu entries", (unsigned long long)cnt);
        cb_ptr[b + 1] = cb_ptr[b] + ((cnt + 7) & ~7ull);
    }
    const uint64_t tot = cb_ptr[n_cb];
    SMH_TRY(tmp.alloc(&row_full, tot));
    // the plan's own buffers (+8 entries of slack: a tile past the last entry still has a valid first address)
    SMH_HIP(hipMalloc(&m->d_t2_val, (tot + 8) * sizeof(T)));
    SMH_HIP(hipMalloc(&m->d_t2_prod, (tot + 8) * sizeof(T)));
    SMH_HIP(hipMalloc((void **)&m->d_t2_code, (tot + 8) * sizeof(uint16_t)));
    SMH_HIP(hipMalloc((void **)&m->d_t2_row, (tot + 8) * sizeof(uint16_t)));
    SMH_HIP(hipMalloc((void **)&m->d_t2_cbptr, ((size_t)n_cb + 1) * sizeof(uint64_t)));
    SMH_HIP(hipMalloc((void **)&m->d_t2_tstart, table_entries * sizeof(uint32_t)));
    SMH_HIP(hipMalloc((void **)&m->d_t2_rbstart, rb_start.size() * sizeof(uint32_t)));
    SMH_HIP(hipMemcpyAsync(m->d_t2_rbstart, rb_start.data(), rb_start.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    SMH_HIP(hipMemsetAsync(m->d_t2_val, 0, (tot + 8) * sizeof(T), s));
    SMH_HIP(hipMemsetAsync(m->d_t2_prod, 0, (tot + 8) * sizeof(T), s));
    SMH_HIP(hipMemsetAsync(m->d_t2_code, 0, (tot + 8) * sizeof(uint16_t), s));
    SMH_HIP(hipMemsetAsync(m->d_t2_row, 0, (tot + 8) * sizeof(uint16_t), s));
    SMH_HIP(hipMemcpyAsync(m->d_t2_cbptr, cb_ptr.data(), cb_ptr.size() * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    if (nnz) {
        hipLaunchKernelGGL(k_t2_fill<T>, dim3(grid), dim3(kBlock), 0, s, m->d_off, (uint64_t)m->n_rows, m->d_col, (const T *)m->d_val, key_s, perm, nnz,
                           d_start, m->d_t2_cbptr, m->d_t2_rbstart, n_rb, (T *)m->d_t2_val, m->d_t2_code, m->d_t2_row, row_full);
        SMH_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_t2_table, dim3(grid), dim3(kBlock), 0, s, row_full, d_start, m->d_t2_cbptr, n_cb, n_rb, m->d_t2_rbstart, m->d_t2_tstart);
    SMH_HIP(hipGetLastError());
    SMH_HIP(hipStreamSynchronize(s));
    // 128 KiB of dynamic LDS (f64) need the attribute on every device the kernel runs on: set with each build, on the matrix's device
    SMH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_t2_expand<T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kT2Slice * sizeof(T))));
    SMH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_t2_expand<T, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kT2Slice * sizeof(T))));
    SMH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_t2_expand<T, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kT2Slice * sizeof(T))));
    m->t2_n_cb = n_cb;
    m->t2_n_rb = n_rb;
    m->t2_R = R;
    m->t2_tot = tot;
    return SMH_OK;
}

void tiled_v1_free(::smh_crs *m) {
    (void)hipFree(m->d_t2_val); (void)hipFree(m->d_t2_prod); (void)hipFree(m->d_t2_code); (void)hipFree(m->d_t2_row);
    (void)hipFree(m->d_t2_cbptr); (void)hipFree(m->d_t2_tstart); (void)hipFree(m->d_t2_rbstart);
    m->d_t2_val = m->d_t2_prod = nullptr;
    m->d_t2_code = m->d_t2_row = nullptr;
    m->d_t2_cbptr = nullptr;
    m->d_t2_tstart = m->d_t2_rbstart = nullptr;
    m->t2_built = m->t2_ok = false;
}

int tiled_v1_build(::smh_crs *m) {
    if (m->t2_built) return SMH_OK;
    SMH_TRY(columns_within_n_cols(m, "tiled variant"));  // (the slice tables are sized from n_cols)
    m->t2_built = true;
    const int rc = m->dtype == SMH_F64 ? build_t<double>(m) : build_t<float>(m);
    if (rc != SMH_OK) {
        const std::string keep = smh_last_error();
        tiled_v1_free(m);
        m->t2_built = true;  // do not try again
        return fail(rc, "%s", keep.c_str());
    }
    m->t2_ok = true;
    return SMH_OK;
}

template <typename T>
static int launch_t(::smh_crs *m, const void *x, size_t x_len, void *y, hipStream_t s) {
    const size_t lds1 = (size_t)kT2Slice * sizeof(T), lds2 = (size_t)kT2Waves * m->t2_R * sizeof(T);
    // a workgroup pays for staging its slice of x (16384 entries), so it should multiply several times as many entries: ~65 000 per
    // workgroup, two 16-byte pieces per thread in flight (profiles/r02_tiled_pass1.log: C2-uniform / C3 with 8 parts per slice 1.14 /
    // 2.17 ms against 1.22 / 2.32 with 5 parts and 8 pieces; 2 M rows x 16 f64 with 2 / 4 / 8 / 32 parts 0.208 / 0.216 / 0.227 / 0.319)
    const uint64_t per_slice = m->t2_tot / (m->t2_n_cb ? m->t2_n_cb : 1);
    uint32_t parts = (uint32_t)((per_slice + 32768) / 65536);
    int unroll = T2Lane<T>::kUnroll;
    if (const char *e = getenv("SMH_TILED_PARTS")) { const int v = atoi(e); if (v >= 1) parts = (uint32_t)v; }      // tuning knobs
    if (const char *e = getenv("SMH_TILED_UNROLL")) { const int v = atoi(e); if (v == 2 || v == 4 || v == 8) unroll = v; }
    parts = parts < 1 ? 1 : (parts > 32 ? 32 : parts);
    auto *kern = unroll == 4 ? k_t2_expand<T, 4> : unroll == 8 ? k_t2_expand<T, 8> : k_t2_expand<T, 2>;
    static const uint32_t xcd_map = getenv("SMH_TILED_XCD") ? (uint32_t)atoi(getenv("SMH_TILED_XCD")) : 3u;  // tuning knob: bit 0 pass 1, bit 1 pass 2
    // (the remapped grid is rounded up to a multiple of 8 so that every XCD's run has the same length)
    const uint32_t g1 = m->t2_n_cb * parts, g1r = (xcd_map & 1u) ? (g1 + 7u) & ~7u : g1;
    hipLaunchKernelGGL(kern, dim3(g1r), dim3(kT2ExpandThreads), lds1, s, (const T *)x, (uint64_t)x_len, (const T *)m->d_t2_val,
                       m->d_t2_code, m->d_t2_cbptr, (T *)m->d_t2_prod, parts, g1, xcd_map & 1u);
    SMH_HIP(hipGetLastError());
    const uint32_t g2 = (m->t2_n_rb + kT2Waves - 1) / kT2Waves, g2r = (xcd_map & 2u) ? (g2 + 7u) & ~7u : g2;
    hipLaunchKernelGGL(k_t2_reduce<T>, dim3(g2r), dim3(kT2Waves * 64), lds2, s, (const T *)m->d_t2_prod, m->d_t2_row,
                       m->d_t2_cbptr, m->d_t2_tstart, m->t2_n_cb, m->t2_n_rb, m->d_t2_rbstart, m->t2_R, (T *)y, xcd_map >> 1 & 1u);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

int launch_spmv_tiled_v1(::smh_crs *m, const void *x, size_t x_len, void *y, hipStream_t s) {
    if (m->n_rows == 0) return SMH_OK;
    return m->dtype == SMH_F64 ? launch_t<double>(m, x, x_len, y, s) : launch_t<float>(m, x, x_len, y, s);
}

}  // namespace smh
