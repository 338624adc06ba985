// transpose_win.hip -- SparseMatrix::transpose (sparsematrix.rs:174-184) without a sort, for matrices whose columns move with
// their rows (bands, block structures, FEM orderings): a counting placement through LDS windows.
//
// The general route (capi.hip: smh_crs_transpose -> assemble.hip) sorts all (target row, source row, value) triples by
// target row with a radix sort: three passes over 12-byte pairs, 8.3 of 16 ms on BASELINE C2.  When the source tiles'
// column spans are monotone in the tile index, a tile of T consecutive TARGET rows receives entries only from a short,
// known range of SOURCE rows, and T counters fit in LDS:
//   S  k_tw_spans   smallest / largest column of every 256-row source tile (one pass over columns[]), last non-empty row;
//      host         source range [s_lo, s_hi) of every target tile from the spans' monotone envelopes; the re-read factor must stay <= 4 (1.4 at 10 M rows)
//   A  k_tw_count   per target tile: LDS histogram of the columns that fall into it -> entries per result row
//      scan         -> offset_rows of the result
//   B  k_tw_place   per target tile: the same walk; an LDS cursor per result row (atomic add) hands every entry its slot in
//                   the row -- in whatever order the lanes arrive
//   D  k_tw_order   per result row: rank the entries by source row, DESCENDING -- the order `set` leaves behind, since
//                   SparseMatCRS::push prepends (sparsemat_crs.rs:85-87) and the source rows arrive ascending -- and write
//                   them in place of the sort's output.  Two entries of a result row with the same source row are a
//                   repeated (row, column) pair of the source: the caller then takes the general route, which knows what
//                   `set` does with repeats.
// The result is bit for bit what the general route gives (tests/test_transpose_gpu.py runs both on every shape);
// rows of the result longer than kTwMaxRow, a re-read factor above 4 (columns that do not follow the rows) and the container's
// first-push quirks (decided by the caller) decline the route.
#include <algorithm>
#include <vector>

#include "internal.hpp"

namespace smh {

int device_exclusive_scan_u32(uint32_t *data, uint64_t n, hipStream_t s, uint64_t *total_out);  // spmv_colblock.hip

namespace {

constexpr uint32_t kTwSrcRows = 256;   // rows of a source tile
constexpr uint32_t kTwMaxRow = 512;    // longest result row k_tw_order ranks
constexpr uint32_t kTwGroup = 32;      // lanes that rank one result row
constexpr uint32_t kTwThreads = 512;   // threads of a counting / placing block
constexpr uint32_t kTwMaxTile = 19456; // target rows per tile at most (76 KiB of counters: two blocks per CU)

struct TwScalars {
    uint32_t last_row_plus1;  // last source row that holds an entry, + 1
    uint32_t max_count;       // longest result row
    uint32_t repeats;         // != 0: some (row, column) pair occurs twice in the source
};

__global__ void __launch_bounds__(kBlock)
k_tw_spans(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, uint64_t n_rows, uint32_t *__restrict__ lo,
           uint32_t *__restrict__ hi, TwScalars *__restrict__ sc) {
    __shared__ uint32_t s_lo[kBlock / kWave], s_hi[kBlock / kWave], s_last[kBlock / kWave];
    const uint64_t r0 = (uint64_t)blockIdx.x * kTwSrcRows;
    const uint64_t r1 = r0 + kTwSrcRows < n_rows ? r0 + kTwSrcRows : n_rows;
    const uint64_t e0 = off[r0], e1 = off[r1];
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
    constexpr int U = 8;  // loads in flight per thread
    for (uint64_t k = e0 + threadIdx.x; k < e1; k += (uint64_t)U * kBlock) {
        uint32_t c[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const uint64_t kk = k + (uint64_t)j * kBlock;
            c[j] = col[kk < e1 ? kk : e1 - 1];  // (a repeat of the last entry changes neither bound)
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            mn = c[j] < mn ? c[j] : mn;
            mx = c[j] > mx ? c[j] : mx;
        }
    }
    const uint64_t r = r0 + threadIdx.x;  // kTwSrcRows == kBlock: one row per thread
    uint32_t last = (r < r1 && off[r + 1] > off[r]) ? (uint32_t)r + 1u : 0u;
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        const uint32_t a = (uint32_t)__shfl_down((int)mn, o, kWave), b = (uint32_t)__shfl_down((int)mx, o, kWave);
        const uint32_t l = (uint32_t)__shfl_down((int)last, o, kWave);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
        last = l > last ? l : last;
    }
    if ((threadIdx.x & (kWave - 1)) == 0) { s_lo[threadIdx.x / kWave] = mn; s_hi[threadIdx.x / kWave] = mx; s_last[threadIdx.x / kWave] = last; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kBlock / kWave; ++w) {
            mn = s_lo[w] < mn ? s_lo[w] : mn;
            mx = s_hi[w] > mx ? s_hi[w] : mx;
            last = s_last[w] > last ? s_last[w] : last;
        }
        lo[blockIdx.x] = mn;  // (empty tile: lo = ~0, hi = 0)
        hi[blockIdx.x] = mx;
        if (last) atomicMax(&sc->last_row_plus1, last);
    }
}

constexpr int kTwUnroll = 8;  // entries a thread has in flight in the counting / placing walks

// entries per result row of target tile blockIdx.x (rows [t0, t0 + T)), counted in LDS
__global__ void __launch_bounds__(kTwThreads)
k_tw_count(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, uint64_t n_rows, uint64_t n_t, uint32_t T,
           const uint32_t *__restrict__ s_lo, const uint32_t *__restrict__ s_hi, uint32_t *__restrict__ cnt_t, TwScalars *__restrict__ sc) {
    extern __shared__ uint32_t cnt[];
    const uint64_t t0 = (uint64_t)blockIdx.x * T;
    for (uint32_t i = threadIdx.x; i < T; i += kTwThreads) cnt[i] = 0;
    __syncthreads();
    const uint64_t ra = (uint64_t)s_lo[blockIdx.x] * kTwSrcRows, rb = (uint64_t)s_hi[blockIdx.x] * kTwSrcRows;
    if (ra < rb) {
        const uint64_t e0 = off[ra], e1 = off[rb < n_rows ? rb : n_rows];
        for (uint64_t k = e0 + threadIdx.x; k < e1; k += (uint64_t)kTwUnroll * kTwThreads) {
            uint32_t d[kTwUnroll];
#pragma unroll
            for (int j = 0; j < kTwUnroll; ++j) {
                const uint64_t kk = k + (uint64_t)j * kTwThreads;
                d[j] = kk < e1 ? __builtin_nontemporal_load(col + kk) - (uint32_t)t0 : 0xFFFFFFFFu;  // (wraps below the tile: fails the test too)
            }
#pragma unroll
            for (int j = 0; j < kTwUnroll; ++j)
                if (d[j] < T) atomicAdd(&cnt[d[j]], 1u);
        }
    }
    __syncthreads();
    uint32_t mx = 0;
    for (uint32_t i = threadIdx.x; i < T && t0 + i < n_t; i += kTwThreads) {
        const uint32_t c = cnt[i];
        cnt_t[t0 + i] = c;
        mx = c > mx ? c : mx;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        const uint32_t a = (uint32_t)__shfl_down((int)mx, o, kWave);
        mx = a > mx ? a : mx;
    }
    if ((threadIdx.x & (kWave - 1)) == 0 && mx) atomicMax(&sc->max_count, mx);
}

// what k_tw_place leaves for k_tw_order: (source row, value) of an entry, in its result row, in arrival order
template <typename V> struct TwPair;
template <> struct __attribute__((aligned(8))) TwPair<float> { uint32_t row; float val; };
template <> struct __attribute__((aligned(16))) TwPair<double> { uint32_t row; uint32_t pad; double val; };

// every entry of the tile's source range that falls into the tile takes the next slot of its result row
template <typename V>
__global__ void __launch_bounds__(kTwThreads)
k_tw_place(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const V *__restrict__ val, uint64_t n_rows, uint64_t n_t,
           uint32_t T, const uint32_t *__restrict__ s_lo, const uint32_t *__restrict__ s_hi, const uint32_t *__restrict__ off_t,
           TwPair<V> *__restrict__ tmp) {
    extern __shared__ uint32_t cur[];         // T cursors, then the current source tile's kTwSrcRows + 1 offsets
    uint32_t *s_off = cur + T;
    const uint64_t t0 = (uint64_t)blockIdx.x * T;
    for (uint32_t i = threadIdx.x; i < T; i += kTwThreads) cur[i] = t0 + i < n_t ? off_t[t0 + i] : 0u;
    for (uint32_t st = s_lo[blockIdx.x]; st < s_hi[blockIdx.x]; ++st) {
        const uint64_t r0 = (uint64_t)st * kTwSrcRows;
        __syncthreads();  // (cursors initialised / the previous tile's offsets no longer read)
        for (uint32_t i = threadIdx.x; i <= kTwSrcRows; i += kTwThreads) s_off[i] = off[r0 + i < n_rows ? r0 + i : n_rows];
        __syncthreads();
        const uint64_t e0 = s_off[0], e1 = s_off[kTwSrcRows];
        for (uint64_t k = e0 + threadIdx.x; k < e1; k += (uint64_t)kTwUnroll * kTwThreads) {
            uint32_t d[kTwUnroll];
            V v[kTwUnroll];
#pragma unroll
            for (int j = 0; j < kTwUnroll; ++j) {  // all loads first: the walk is latency-bound otherwise
                const uint64_t kk = k + (uint64_t)j * kTwThreads;
                const bool in = kk < e1;
                d[j] = in ? __builtin_nontemporal_load(col + kk) - (uint32_t)t0 : 0xFFFFFFFFu;
                v[j] = in ? __builtin_nontemporal_load(val + kk) : V(0);
            }
            uint32_t a[kTwUnroll], b[kTwUnroll];
#pragma unroll
            for (int j = 0; j < kTwUnroll; ++j) { a[j] = 0; b[j] = kTwSrcRows; }
#pragma unroll
            for (int step = 0; step < 8; ++step) {  // kTwSrcRows = 2^8: the row of entry kk, s_off[a] <= kk < s_off[a + 1]
#pragma unroll
                for (int j = 0; j < kTwUnroll; ++j) {
                    const uint32_t mid = (a[j] + b[j]) >> 1;
                    const bool up = (uint64_t)s_off[mid] <= k + (uint64_t)j * kTwThreads;
                    a[j] = up ? mid : a[j];
                    b[j] = up ? b[j] : mid;
                }
            }
#pragma unroll
            for (int j = 0; j < kTwUnroll; ++j) {
                if (d[j] >= T) continue;
                const uint32_t slot = atomicAdd(&cur[d[j]], 1u);
                TwPair<V> p;
                p.row = (uint32_t)r0 + a[j];
                p.val = v[j];
                if constexpr (sizeof(V) == 8) p.pad = 0;
                tmp[slot] = p;
            }
        }
    }
}

// result row j: its entries ranked by source row, descending; kTwGroup lanes per row
template <typename V>
__global__ void __launch_bounds__(kBlock)
k_tw_order(const uint32_t *__restrict__ off_t, const TwPair<V> *__restrict__ tmp, uint64_t n_t, uint32_t *__restrict__ out_col,
           V *__restrict__ out_val, TwScalars *__restrict__ sc) {
    __shared__ __attribute__((aligned(16))) uint32_t keys[kBlock / kTwGroup][kTwMaxRow];
    const uint32_t g = threadIdx.x / kTwGroup, lane = threadIdx.x % kTwGroup;
    const uint64_t groups = (uint64_t)gridDim.x * (kBlock / kTwGroup);
    bool repeat = false;
    for (uint64_t j = (uint64_t)blockIdx.x * (kBlock / kTwGroup) + g; j < n_t; j += groups) {
        const uint32_t base = off_t[j], len = off_t[j + 1] - base;  // (len <= kTwMaxRow: checked by the host)
        const uint32_t padded = (len + 3u) & ~3u;
        for (uint32_t e = lane; e < padded; e += kTwGroup) keys[g][e] = e < len ? tmp[base + e].row : 0xFFFFFFFFu;
        // (same wave, LDS in program order: the reads below see the writes above.)  Rank = entries with a larger source row;
        // the padding (~0) counts as larger for everybody and is taken off again; equal keys are a repeated pair.
        for (uint32_t e = lane; e < len; e += kTwGroup) {
            const TwPair<V> mine = tmp[base + e];
            uint32_t gt = 0, ge = 0;
            for (uint32_t q = 0; q < padded; q += 4) {
                const uint4 o = *reinterpret_cast<const uint4 *>(&keys[g][q]);
                gt += (o.x > mine.row) + (o.y > mine.row) + (o.z > mine.row) + (o.w > mine.row);
                ge += (o.x >= mine.row) + (o.y >= mine.row) + (o.z >= mine.row) + (o.w >= mine.row);
            }
            repeat |= ge != gt + 1u;
            const uint32_t rank = gt - (padded - len);
            out_col[base + rank] = mine.row;
            out_val[base + rank] = mine.val;
        }
    }
    if (repeat) atomicOr(&sc->repeats, 1u);
}

template <typename V>
int run(const uint32_t *off, const uint32_t *col, const V *val, size_t n_rows, size_t nnz, uint32_t max_col, uint32_t **off_out, uint32_t **col_out,
        V **val_out, size_t *n_cols_out, bool *done, hipStream_t s) {
    *done = false;
    const uint64_t n_t = (uint64_t)max_col + 1;  // rows of the result
    const uint64_t n_st = (n_rows + kTwSrcRows - 1) / kTwSrcRows;
    int device = 0, cus = 256;
    SMH_HIP(hipGetDevice(&device));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    // target rows per tile: one round of two blocks per CU where the counters allow it
    uint64_t T = (n_t + 2ull * cus - 1) / (2ull * cus);
    T = (T + 63) & ~63ull;
    T = T < 4096 ? 4096 : (T > kTwMaxTile ? kTwMaxTile : T);
    const uint64_t n_tt = (n_t + T - 1) / T;

    uint32_t *d_lo = nullptr, *d_hi = nullptr, *d_slo = nullptr, *d_shi = nullptr, *d_offt = nullptr, *d_col = nullptr;
    TwPair<V> *d_tmp = nullptr;
    V *d_val = nullptr;
    TwScalars *d_sc = nullptr;
    auto cleanup = [&](bool keep_result) {
        (void)hipFree(d_lo); (void)hipFree(d_slo); (void)hipFree(d_tmp);
        if (!keep_result) { (void)hipFree(d_offt); (void)hipFree(d_col); (void)hipFree(d_val); }
    };
    auto go = [&]() -> int {
        SMH_HIP(hipMalloc((void **)&d_lo, (2 * n_st + 4) * sizeof(uint32_t)));  // spans and the scalars in one block
        d_hi = d_lo + n_st;
        d_sc = reinterpret_cast<TwScalars *>(d_hi + n_st);
        SMH_HIP(hipMemsetAsync(d_sc, 0, sizeof(TwScalars), s));
        hipLaunchKernelGGL(k_tw_spans, dim3((unsigned)n_st), dim3(kBlock), 0, s, off, col, (uint64_t)n_rows, d_lo, d_hi, d_sc);
        SMH_HIP(hipGetLastError());
        std::vector<uint32_t> lo(n_st), hi(n_st);
        TwScalars sc;
        SMH_HIP(hipMemcpyAsync(lo.data(), d_lo, n_st * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        SMH_HIP(hipMemcpyAsync(hi.data(), d_hi, n_st * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        SMH_HIP(hipMemcpyAsync(&sc, d_sc, sizeof sc, hipMemcpyDeviceToHost, s));
        SMH_HIP(hipStreamSynchronize(s));
        // Which source tiles can reach into target tile [t0, t1)?  Those with hi >= t0 and lo < t1.  With the running
        // maximum of hi (tiles up to s) and the running minimum of lo (tiles from s on) -- both non-decreasing in s, both on
        // the safe side of the tile's own span -- that is one contiguous range found by bisection; for a band the envelopes
        // ARE the spans up to their jitter, for scattered columns the range is everything and the route is declined below.
        for (uint64_t t = 1; t < n_st; ++t) hi[t] = hi[t] > hi[t - 1] ? hi[t] : hi[t - 1];
        for (uint64_t t = n_st - 1; t-- > 0;) lo[t] = lo[t] < lo[t + 1] ? lo[t] : lo[t + 1];
        std::vector<uint32_t> slo(n_tt), shi(n_tt);
        uint64_t walked = 0;
        for (uint64_t tt = 0; tt < n_tt; ++tt) {
            const uint64_t t0 = tt * T, t1 = t0 + T;
            const size_t a = std::lower_bound(hi.begin(), hi.end(), (uint32_t)t0) - hi.begin();                                   // first with hi >= t0
            const size_t b = t1 > 0xFFFFFFFFull ? n_st : (size_t)(std::lower_bound(lo.begin(), lo.end(), (uint32_t)t1) - lo.begin());  // first with lo >= t1
            if (a < b) { slo[tt] = (uint32_t)a; shi[tt] = (uint32_t)b; walked += b - a; }
            else { slo[tt] = shi[tt] = 0; }
        }
        if (walked > n_st * 4 + n_tt) return SMH_OK;  // the source would be re-read more than 4 times: declined
        SMH_HIP(hipMalloc((void **)&d_slo, 2 * n_tt * sizeof(uint32_t)));
        d_shi = d_slo + n_tt;
        SMH_HIP(hipMemcpyAsync(d_slo, slo.data(), n_tt * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        SMH_HIP(hipMemcpyAsync(d_shi, shi.data(), n_tt * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        SMH_HIP(hipMalloc((void **)&d_offt, (n_t + 1) * sizeof(uint32_t)));
        SMH_HIP(hipMemsetAsync(d_offt + n_t, 0, sizeof(uint32_t), s));
        const size_t lds_a = (size_t)T * sizeof(uint32_t), lds_b = ((size_t)T + kTwSrcRows + 1) * sizeof(uint32_t);
        SMH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tw_count), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
        SMH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tw_place<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
        hipLaunchKernelGGL(k_tw_count, dim3((unsigned)n_tt), dim3(kTwThreads), lds_a, s, off, col, (uint64_t)n_rows, n_t, (uint32_t)T, d_slo, d_shi, d_offt, d_sc);
        SMH_HIP(hipGetLastError());
        SMH_HIP(hipMemcpyAsync(&sc, d_sc, sizeof sc, hipMemcpyDeviceToHost, s));
        SMH_HIP(hipStreamSynchronize(s));
        if (sc.max_count > kTwMaxRow) return SMH_OK;  // a result row too long for the ranking kernel: declined
        uint64_t total = 0;
        SMH_TRY(device_exclusive_scan_u32(d_offt, n_t + 1, s, &total));
        if (total != nnz) return fail(SMH_ERR_INVALID, "windowed transposition counted %llu of %zu entries", (unsigned long long)total, nnz);
        SMH_HIP(hipMalloc((void **)&d_tmp, (nnz + 4) * sizeof(TwPair<V>)));
        SMH_HIP(hipMalloc((void **)&d_col, (nnz + 4) * sizeof(uint32_t)));
        SMH_HIP(hipMalloc((void **)&d_val, (nnz + 4) * sizeof(V)));
        SMH_HIP(hipMemsetAsync(d_col + nnz, 0, 4 * sizeof(uint32_t), s));
        SMH_HIP(hipMemsetAsync(d_val + nnz, 0, 4 * sizeof(V), s));
        hipLaunchKernelGGL((k_tw_place<V>), dim3((unsigned)n_tt), dim3(kTwThreads), lds_b, s, off, col, val, (uint64_t)n_rows, n_t, (uint32_t)T, d_slo, d_shi,
                           d_offt, d_tmp);
        SMH_HIP(hipGetLastError());
        const uint64_t row_blocks = (n_t + (kBlock / kTwGroup) - 1) / (kBlock / kTwGroup);
        hipLaunchKernelGGL((k_tw_order<V>), dim3((unsigned)(row_blocks < 16384 ? row_blocks : 16384)), dim3(kBlock), 0, s, d_offt, d_tmp, n_t, d_col,
                           d_val, d_sc);
        SMH_HIP(hipGetLastError());
        SMH_HIP(hipMemcpyAsync(&sc, d_sc, sizeof sc, hipMemcpyDeviceToHost, s));
        SMH_HIP(hipStreamSynchronize(s));
        if (sc.repeats) return SMH_OK;  // a repeated (row, column) pair: `set` semantics live in the general route
        *n_cols_out = sc.last_row_plus1;
        *done = true;
        return SMH_OK;
    };
    const int rc = go();
    cleanup(rc == SMH_OK && *done);
    if (rc == SMH_OK && *done) { *off_out = d_offt; *col_out = d_col; *val_out = d_val; }
    return rc;
}

}  // namespace

// *done == false with SMH_OK: the matrix does not qualify, nothing was produced
int transpose_windowed(int dtype, const uint32_t *off, const uint32_t *col, const void *val, size_t n_rows, size_t nnz, uint32_t max_col,
                       uint32_t **off_out, uint32_t **col_out, void **val_out, size_t *n_rows_out, size_t *n_cols_out, bool *done, hipStream_t s) {
    *done = false;
    if (n_rows == 0 || nnz == 0) return SMH_OK;
    *n_rows_out = (size_t)max_col + 1;
    if (dtype == SMH_F64)
        return run<double>(off, col, (const double *)val, n_rows, nnz, max_col, off_out, col_out, (double **)val_out, n_cols_out, done, s);
    return run<float>(off, col, (const float *)val, n_rows, nnz, max_col, off_out, col_out, (float **)val_out, n_cols_out, done, s);
}

}  // namespace smh
