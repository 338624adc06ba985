// transpose_bucket.hip -- SparseMatrix::transpose (sparsematrix.rs:174-184) without the device-wide sort, for matrices with
// local structure (bands, stencils, block and FEM orderings): two bucketed passes instead of a radix sort over all entries.
//
// The general route (capi.hip: smh_crs_transpose -> assemble.hip) sorts all (target row, source row, value) triples by
// target row with a device-wide radix sort: three passes over 12-byte pairs, 8.3 of 16 ms on BASELINE C2.  Here the target
// rows are cut into BUCKETS of 2^shift rows (256 unless the result's rows are long) and
//   H  k_tb_hist  every 256-row source tile counts its entries per bucket in an LDS hash table and adds the counts to the
//                 buckets' totals -- one global atomic per (tile, bucket) pair, not per entry.  A tile that reaches more
//                 than kTbMaxDistinct buckets (columns without locality) declines the route.
//      scan       -> where every bucket starts; the largest bucket must fit the LDS of pass S
//   P  k_tb_part  the same tiles again, in chunks of 4096 entries: a chunk is ordered by bucket in LDS, reserves a run in each
//                 of its buckets (one global atomic per pair) and writes its entries there as (source row, target row inside
//                 the bucket, value) -- contiguous runs, in no particular order inside the bucket
//   S  k_tb_sort  one block per bucket, a thread keeps its <= 19 entries in registers from the single read to the final stores:
//                 entries per result row (-> offset_rows of the result, no separate counting pass), the source rows grouped
//                 by result row in LDS, and every entry ranked inside its row by counting the LARGER source rows -- descending
//                 source row is the order `set` leaves behind, since SparseMatCRS::push prepends (sparsemat_crs.rs:85-87) and
//                 the source rows arrive ascending.  Two entries on one position (same result row, same source row) are a
//                 repeated (row, column) pair of the source: the caller then takes the general route, which knows what `set`
//                 does with repeats.  The column lists of ColumnIter (column_lists_bucketed) are the same passes with the
//                 entry index as key, ascending, without values.
// The order in which tiles reserve their runs is arbitrary; pass S orders every bucket completely, so the result is
// deterministic and bit for bit what the general route gives (tests/test_transpose_gpu.py runs both on every shape).
// A first attempt walked per-target-tile source windows with LDS cursors and scattered the entries straight to their rows
// (correct, 13.7 ms on C2: 320 M scattered 8-byte stores into a 2 MB window per block, 512 blocks at once -- every store
// its own HBM transaction); the bucket pass exists to make those writes runs.
#include <algorithm>
#include <vector>

#include "internal.hpp"

namespace smh {

int device_exclusive_scan_u32(uint32_t *data, uint64_t n, hipStream_t s, uint64_t *total_out);  // spmv_colblock.hip

namespace {

constexpr uint32_t kTbSrcRows = 256;      // rows of a source tile (= threads of its block)
constexpr uint32_t kTbHash = 1024;        // slots of a tile's bucket table
constexpr uint32_t kTbMaxDistinct = 640;  // buckets a tile may reach
constexpr uint32_t kTbCap = 9728;         // entries of a bucket at most (76 KiB of LDS in pass S: two blocks per CU)
constexpr uint32_t kTbSortThreads = 512;
constexpr uint32_t kTbEmpty = 0xFFFFFFFFu;
constexpr int kTbUnroll = 8;              // loads a thread keeps in flight

struct TbScalars {
    uint32_t last_row_plus1;  // last source row that holds an entry, + 1
    uint32_t max_bucket;      // entries of the largest bucket
    uint32_t overflow;        // != 0: a tile reached more than kTbMaxDistinct buckets
    uint32_t repeats;         // != 0: some (row, column) pair occurs twice in the source
};

__device__ __forceinline__ uint32_t tb_hash(uint32_t b) { return (b * 2654435761u) >> 22; }  // 10 bits

// slot of bucket b in the tile's table, inserting it if absent; kTbEmpty when the table is too full
__device__ __forceinline__ uint32_t tb_insert(uint32_t *keys, uint32_t *n_used, uint32_t b) {
    uint32_t h = tb_hash(b);
    for (uint32_t probe = 0; probe < kTbHash; ++probe, h = (h + 1) & (kTbHash - 1)) {
        const uint32_t seen = __atomic_load_n(&keys[h], __ATOMIC_RELAXED);
        if (seen == b) return h;
        if (seen == kTbEmpty) {
            const uint32_t old = atomicCAS(&keys[h], kTbEmpty, b);
            if (old == kTbEmpty) {
                if (atomicAdd(n_used, 1u) >= kTbMaxDistinct) return kTbEmpty;
                return h;
            }
            if (old == b) return h;
        }
    }
    return kTbEmpty;
}

// H: bucket totals
__global__ void __launch_bounds__(kTbSrcRows)
k_tb_hist(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, uint64_t n_rows, uint32_t shift, uint32_t *__restrict__ total,
          TbScalars *__restrict__ sc) {
    __shared__ uint32_t keys[kTbHash], cnt[kTbHash], n_used, s_last[kTbSrcRows / kWave];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < kTbHash; i += kTbSrcRows) { keys[i] = kTbEmpty; cnt[i] = 0; }
    if (tid == 0) n_used = 0;
    __syncthreads();
    const uint64_t r0 = (uint64_t)blockIdx.x * kTbSrcRows;
    const uint64_t r1 = r0 + kTbSrcRows < n_rows ? r0 + kTbSrcRows : n_rows;
    const uint64_t e0 = off[r0], e1 = off[r1];
    bool lost = false;
    for (uint64_t k = e0 + tid; k < e1; k += (uint64_t)kTbUnroll * kTbSrcRows) {
        uint32_t b[kTbUnroll];
#pragma unroll
        for (int j = 0; j < kTbUnroll; ++j) {
            const uint64_t kk = k + (uint64_t)j * kTbSrcRows;
            b[j] = kk < e1 ? col[kk] >> shift : kTbEmpty;
        }
#pragma unroll
        for (int j = 0; j < kTbUnroll; ++j) {
            if (b[j] == kTbEmpty) continue;
            const uint32_t h = tb_insert(keys, &n_used, b[j]);
            if (h == kTbEmpty) lost = true; else atomicAdd(&cnt[h], 1u);
        }
    }
    const uint64_t r = r0 + tid;
    uint32_t last = (r < r1 && off[r + 1] > off[r]) ? (uint32_t)r + 1u : 0u;
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        const uint32_t l = (uint32_t)__shfl_down((int)last, o, kWave);
        last = l > last ? l : last;
    }
    if ((tid & (kWave - 1)) == 0) s_last[tid / kWave] = last;
    __syncthreads();
    if (lost) atomicOr(&sc->overflow, 1u);
    for (uint32_t i = tid; i < kTbHash; i += kTbSrcRows)
        if (keys[i] != kTbEmpty && cnt[i]) atomicAdd(&total[keys[i]], cnt[i]);
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < (int)(kTbSrcRows / kWave); ++w) last = s_last[w] > last ? s_last[w] : last;
        if (last) atomicMax(&sc->last_row_plus1, last);
    }
}

__global__ void __launch_bounds__(kBlock)
k_tb_max(const uint32_t *__restrict__ total, uint64_t n_b, TbScalars *__restrict__ sc) {
    uint32_t mx = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_b; i += (uint64_t)gridDim.x * kBlock) mx = total[i] > mx ? total[i] : mx;
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        const uint32_t a = (uint32_t)__shfl_down((int)mx, o, kWave);
        mx = a > mx ? a : mx;
    }
    if ((threadIdx.x & (kWave - 1)) == 0 && mx) atomicMax(&sc->max_bucket, mx);
}

// P: entries into their buckets.  A tile is taken in chunks of kTbChunk entries; a chunk is ordered by bucket in LDS first,
// so that what goes to a bucket leaves as ONE contiguous run per array (a counting sort by bucket: table counts, prefix,
// placement), and the tile's blocks do not litter L2 with partially written lines.
constexpr uint32_t kTbChunk = 4096;
constexpr uint32_t kTbPartThreads = 512;  // (8 entries per thread and chunk: 16 cost 207-220 VGPRs, two waves per SIMD)

template <typename V, bool INFO>  // INFO: the column tables (keys = entry indices, no values); else the transposition
__global__ void __launch_bounds__(kTbPartThreads)
k_tb_part(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const V *__restrict__ val, uint64_t n_rows, uint32_t shift,
          uint32_t *__restrict__ cursor, uint32_t *__restrict__ bk_src, uint8_t *__restrict__ bk_t, V *__restrict__ bk_val) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tb_lds[];
    V *st_val = reinterpret_cast<V *>(tb_lds);
    uint32_t *st_src = reinterpret_cast<uint32_t *>(st_val + kTbChunk);
    uint32_t *keys = st_src + kTbChunk, *cnt = keys + kTbHash, *lpre = cnt + kTbHash, *start = lpre + kTbHash, *s_off = start + kTbHash;
    uint32_t *s_wsum = s_off + kTbSrcRows + 1;  // the waves' partial sums (8), then the table's occupancy
    uint16_t *st_h = reinterpret_cast<uint16_t *>(s_wsum + 12);
    uint8_t *st_t = reinterpret_cast<uint8_t *>(st_h + kTbChunk);
    const uint32_t tid = threadIdx.x;
    const uint64_t r0 = (uint64_t)blockIdx.x * kTbSrcRows;
    for (uint32_t i = tid; i <= kTbSrcRows; i += kTbPartThreads) s_off[i] = off[r0 + i < n_rows ? r0 + i : n_rows];
    __syncthreads();
    const uint64_t e0 = s_off[0], e1 = s_off[kTbSrcRows];
    const uint32_t low = (1u << shift) - 1u;
    constexpr int kPer = kTbChunk / kTbPartThreads;       // 8 entries per thread and chunk
    constexpr int kSlots = kTbHash / kTbPartThreads;      // 2 table slots per thread in the prefix
    for (uint64_t c0 = e0; c0 < e1; c0 += kTbChunk) {
        const uint32_t n = (uint32_t)(e1 - c0 < kTbChunk ? e1 - c0 : kTbChunk);
        for (uint32_t i = tid; i < kTbHash; i += kTbPartThreads) { keys[i] = kTbEmpty; cnt[i] = 0; }
        if (tid == 0) s_wsum[8] = 0;
        __syncthreads();
        uint32_t c[kPer], hr[kPer];  // column; row inside the tile << 24 | table slot << 12 | rank inside the chunk's share of the bucket
        V v[kPer];
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const uint32_t i = tid + (uint32_t)j * kTbPartThreads;
            const bool in = i < n;
            c[j] = in ? col[c0 + i] : kTbEmpty;
            if constexpr (INFO) v[j] = V(0); else v[j] = in ? __builtin_nontemporal_load(val + c0 + i) : V(0);
        }
        if constexpr (INFO) {
#pragma unroll
            for (int j = 0; j < kPer; ++j) hr[j] = 0;
        } else {   // the rows of the chunk's entries (bisection in the tile's offsets, the lookups of a thread interleaved)
            uint32_t a[kPer], b[kPer];
#pragma unroll
            for (int j = 0; j < kPer; ++j) { a[j] = 0; b[j] = kTbSrcRows; }
#pragma unroll
            for (int step = 0; step < 8; ++step) {  // kTbSrcRows = 2^8: s_off[a] <= entry < s_off[a + 1]
#pragma unroll
                for (int j = 0; j < kPer; ++j) {
                    const uint32_t mid = (a[j] + b[j]) >> 1;
                    const bool up = (uint64_t)s_off[mid] <= c0 + tid + (uint64_t)j * kTbPartThreads;
                    a[j] = up ? mid : a[j];
                    b[j] = up ? b[j] : mid;
                }
            }
#pragma unroll
            for (int j = 0; j < kPer; ++j) hr[j] = a[j] << 24;
        }
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            if (tid + (uint32_t)j * kTbPartThreads >= n) continue;
            const uint32_t h = tb_insert(keys, &s_wsum[8], c[j] >> shift);  // (k_tb_hist has checked that the table holds the tile)
            hr[j] |= (h << 12) | atomicAdd(&cnt[h], 1u);                     // (h < 2^10 slots, rank < kTbChunk = 2^12)
        }
        __syncthreads();
        // exclusive prefix over the table's counts, and a run in every bucket the chunk reaches
        {
            uint32_t q[kSlots], run = 0;
#pragma unroll
            for (int t = 0; t < kSlots; ++t) { q[t] = cnt[kSlots * tid + t]; run += q[t]; }
            uint32_t incl = run;
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, o, kWave);
                if ((int)(tid & (kWave - 1)) >= o) incl += up;
            }
            if ((tid & (kWave - 1)) == kWave - 1) s_wsum[tid / kWave] = incl;
            __syncthreads();
            uint32_t acc = incl - run;
            for (uint32_t w = 0; w < tid / kWave; ++w) acc += s_wsum[w];
#pragma unroll
            for (int t = 0; t < kSlots; ++t) {
                lpre[kSlots * tid + t] = acc;
                acc += q[t];
                if (q[t]) start[kSlots * tid + t] = atomicAdd(&cursor[keys[kSlots * tid + t]], q[t]);
            }
        }
        __syncthreads();  // (lpre complete)
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            if (tid + (uint32_t)j * kTbPartThreads >= n) continue;
            const uint32_t h = (hr[j] >> 12) & 0x3FFu, p = lpre[h] + (hr[j] & 0xFFFu);
            if constexpr (INFO) st_src[p] = (uint32_t)(c0 + tid + (uint64_t)j * kTbPartThreads);  // the entry's index
            else { st_src[p] = (uint32_t)r0 + (hr[j] >> 24); st_val[p] = v[j]; }
            st_h[p] = (uint16_t)h;
            st_t[p] = (uint8_t)(c[j] & low);
        }
        __syncthreads();
        for (uint32_t p = tid; p < n; p += kTbPartThreads) {  // neighbours in p are neighbours in their bucket's run
            const uint32_t h = st_h[p];
            const uint32_t gpos = start[h] + (p - lpre[h]);
            // (plain stores on purpose: a run is written by many wave-instructions, 4 bytes and 1 byte per lane, and relies on L2
            // to merge them into lines -- with non-temporal stores the C2 transposition took 27 ms instead of 7)
            bk_src[gpos] = st_src[p];
            bk_t[gpos] = st_t[p];
            if constexpr (!INFO) bk_val[gpos] = st_val[p];
        }
        __syncthreads();  // (the table and the stage are reset by the next chunk)
    }
}

template <typename V> constexpr size_t tb_part_lds() {
    return kTbChunk * (sizeof(V) + 4 + 2 + 1) + (4 * kTbHash + kTbSrcRows + 1 + 12) * 4 + 16;
}

// S: one bucket, ordered completely.  A thread keeps its (at most kTbPer) entries in registers from the single read of the bucket
// to the final stores; LDS holds the source rows grouped by result row (what the ranking compares) and little else.
constexpr uint32_t kTbPer = (kTbCap + kTbSortThreads - 1) / kTbSortThreads;  // 19 entries per thread at most
constexpr uint32_t kTbSlots = kTbCap + 3 * 256;  // LDS slots: every result row starts at a multiple of four (16-byte reads)
constexpr size_t kTbSortLds = (kTbSlots + kTbCap / 32 + 256 + 257 + 257 + 3) * 4 + 16;

template <typename V, bool INFO>
__global__ void __launch_bounds__(kTbSortThreads)
k_tb_sort(const uint32_t *__restrict__ bucket_off, const uint32_t *__restrict__ bk_src, const uint8_t *__restrict__ bk_t, const V *__restrict__ bk_val,
          uint64_t n_t, uint32_t shift, uint32_t *__restrict__ off_t, uint32_t *__restrict__ out_col, V *__restrict__ out_val, TbScalars *__restrict__ sc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tb_lds[];
    uint32_t *keys = reinterpret_cast<uint32_t *>(tb_lds);  // source row + 1 of the entry in slot s (slots grouped by result row; 0: padding)
    uint32_t *taken = keys + kTbSlots;                      // one bit per output position: two entries on one position = a repeated pair
    uint32_t *rcnt = taken + kTbCap / 32, *rofs = rcnt + 256, *rpad = rofs + 257;
    const uint32_t tid = threadIdx.x;
    const uint32_t brows = 1u << shift;
    const uint32_t b0 = bucket_off[blockIdx.x], n = bucket_off[blockIdx.x + 1] - b0;  // (n <= kTbCap: checked by the host)
    const uint64_t row0 = (uint64_t)blockIdx.x << shift;
    // the bucket, once: entry i = tid + j * threads lives in register set j
    uint32_t t[kTbPer], src[kTbPer];
    V v[kTbPer];
#pragma unroll
    for (uint32_t j = 0; j < kTbPer; ++j) {
        const uint32_t i = tid + j * kTbSortThreads;
        const bool in = i < n;
        t[j] = in ? bk_t[b0 + i] : 256u;
        src[j] = in ? bk_src[b0 + i] : 0u;
        if constexpr (INFO) v[j] = V(0); else v[j] = in ? __builtin_nontemporal_load(bk_val + b0 + i) : V(0);
    }
    if (tid < 256) rcnt[tid] = 0;
    for (uint32_t i = tid; i < kTbCap / 32; i += kTbSortThreads) taken[i] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < kTbPer; ++j)
        if (t[j] < 256u) atomicAdd(&rcnt[t[j]], 1u);
    __syncthreads();
    if (tid < kWave) {  // exclusive prefixes over the bucket's (at most 256) result rows, four per lane: entries, and slots (rows padded to 4)
        uint32_t c[4], run = 0, runp = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { c[q] = rcnt[4 * tid + q]; run += c[q]; runp += (c[q] + 3u) & ~3u; }
        uint32_t incl = run, inclp = runp;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o, kWave), upp = (uint32_t)__shfl_up((int)inclp, o, kWave);
            if ((int)tid >= o) { incl += up; inclp += upp; }
        }
        uint32_t acc = incl - run, accp = inclp - runp;
#pragma unroll
        for (int q = 0; q < 4; ++q) { rofs[4 * tid + q] = acc; rpad[4 * tid + q] = accp; acc += c[q]; accp += (c[q] + 3u) & ~3u; }
        if (tid == kWave - 1) { rofs[256] = acc; rpad[256] = accp; }
    }
    __syncthreads();
    const uint32_t n_slots = rpad[256];
    if (tid < 256) {
        rcnt[tid] = 0;  // (now the rows' cursors)
        if (tid < brows && row0 + tid < n_t) off_t[row0 + tid] = b0 + rofs[tid];
    }
    for (uint32_t i = tid; i < n_slots; i += kTbSortThreads) keys[i] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < kTbPer; ++j) {
        if (t[j] >= 256u) continue;
        // (source rows are < 2^32 - 1: + 1 does not wrap; 0 stays the padding, below every key)
        keys[rpad[t[j]] + atomicAdd(&rcnt[t[j]], 1u)] = src[j] + 1u;
    }
    __syncthreads();
    // every entry: its rank inside its result row = the row's entries with a LARGER source row
    bool repeat = false;
#pragma unroll
    for (uint32_t j = 0; j < kTbPer; ++j) {
        if (t[j] >= 256u) continue;
        const uint32_t base = rpad[t[j]], len = rpad[t[j] + 1] - base;  // (len: a multiple of four)
        const uint32_t me = src[j] + 1u;
        uint32_t gt = 0;
        for (uint32_t q = 0; q < len; q += 4) {
            const uint4 o = *reinterpret_cast<const uint4 *>(keys + base + q);
            if constexpr (INFO) gt += (o.x && o.x < me) + (o.y && o.y < me) + (o.z && o.z < me) + (o.w && o.w < me);  // ascending: the SMALLER keys (0 = padding)
            else gt += (o.x > me) + (o.y > me) + (o.z > me) + (o.w > me);
        }
        const uint32_t pos = rofs[t[j]] + gt;
        out_col[b0 + pos] = src[j];
        if constexpr (!INFO) {
            const uint32_t bit = 1u << (pos & 31u);
            repeat |= (atomicOr(&taken[pos >> 5], bit) & bit) != 0u;  // equal source rows rank equal
            out_val[b0 + pos] = v[j];
        }
    }
    if (repeat) atomicOr(&sc->repeats, 1u);
}

// INFO: col_ptr / entries are the caller's device arrays (*off_out / *col_out on entry, at least max_col + 2 and nnz elements)
template <typename V, bool INFO>
int run(const uint32_t *off, const uint32_t *col, const V *val, size_t n_rows, size_t nnz, uint32_t max_col, uint32_t **off_out, uint32_t **col_out,
        V **val_out, size_t *n_cols_out, bool *done, hipStream_t s) {
    *done = false;
    const uint64_t n_t = (uint64_t)max_col + 1;  // rows of the result
    const uint64_t n_st = (n_rows + kTbSrcRows - 1) / kTbSrcRows;
    // rows per bucket: 256, fewer when the result's rows are long (a bucket of average rows should hold <= 4096 entries:
    // dense stretches -- the clamped windows at the ends of BASELINE C2 hold twice the average -- then still fit)
    uint32_t shift = 8;
    while (shift > 0 && ((uint64_t)nnz << shift) / n_t > 4096) --shift;
    uint64_t n_b = 0;

    uint32_t *d_total = nullptr, *d_cursor = nullptr, *d_src = nullptr, *d_offt = nullptr, *d_col = nullptr;
    uint8_t *d_t = nullptr;
    V *d_bval = nullptr, *d_val = nullptr;
    TbScalars *d_sc = nullptr;
    auto cleanup = [&](bool keep_result) {
        (void)hipFree(d_total); (void)hipFree(d_cursor); (void)hipFree(d_src); (void)hipFree(d_t); (void)hipFree(d_bval);
        if (!keep_result && !INFO) { (void)hipFree(d_offt); (void)hipFree(d_col); (void)hipFree(d_val); }
    };
    auto go = [&]() -> int {
        TbScalars sc;
        for (int attempt = 0;; ++attempt) {
            n_b = (n_t + (1ull << shift) - 1) >> shift;
            (void)hipFree(d_total);
            d_total = nullptr;
            SMH_HIP(hipMalloc((void **)&d_total, (n_b + 1 + 4) * sizeof(uint32_t)));  // the buckets' totals / starts, then the scalars
            d_sc = reinterpret_cast<TbScalars *>(d_total + n_b + 1);
            SMH_HIP(hipMemsetAsync(d_total, 0, (n_b + 1 + 4) * sizeof(uint32_t), s));
            hipLaunchKernelGGL(k_tb_hist, dim3((unsigned)n_st), dim3(kTbSrcRows), 0, s, off, col, (uint64_t)n_rows, shift, d_total, d_sc);
            SMH_HIP(hipGetLastError());
            hipLaunchKernelGGL(k_tb_max, dim3((unsigned)std::min<uint64_t>((n_b + kBlock - 1) / kBlock, 1024)), dim3(kBlock), 0, s, d_total, n_b, d_sc);
            SMH_HIP(hipGetLastError());
            SMH_HIP(hipMemcpyAsync(&sc, d_sc, sizeof sc, hipMemcpyDeviceToHost, s));
            SMH_HIP(hipStreamSynchronize(s));
            if (sc.overflow) return SMH_OK;  // columns without locality: declined
            if (sc.max_bucket <= kTbCap) break;
            // a bucket too large for the LDS of pass S (dense stretches of columns): halve the buckets' rows and count again
            if (shift == 0 || attempt == 3) return SMH_OK;
            --shift;
        }
        uint64_t total = 0;
        SMH_TRY(device_exclusive_scan_u32(d_total, n_b + 1, s, &total));
        if (total != nnz) return fail(SMH_ERR_INVALID, "bucketed transposition counted %llu of %zu entries", (unsigned long long)total, nnz);
        SMH_HIP(hipMalloc((void **)&d_cursor, n_b * sizeof(uint32_t)));
        SMH_HIP(hipMemcpyAsync(d_cursor, d_total, n_b * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        SMH_HIP(hipMalloc((void **)&d_src, (nnz + 4) * sizeof(uint32_t)));
        SMH_HIP(hipMalloc((void **)&d_t, nnz + 16));
        if (!INFO) SMH_HIP(hipMalloc((void **)&d_bval, (nnz + 4) * sizeof(V)));
        SMH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tb_part<V, INFO>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tb_part_lds<V>()));
        SMH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tb_sort<V, INFO>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTbSortLds));
        hipLaunchKernelGGL((k_tb_part<V, INFO>), dim3((unsigned)n_st), dim3(kTbPartThreads), tb_part_lds<V>(), s, off, col, val, (uint64_t)n_rows, shift, d_cursor, d_src, d_t, d_bval);
        SMH_HIP(hipGetLastError());
        if (INFO) {
            d_offt = *off_out;
            d_col = *col_out;
        } else {
            SMH_HIP(hipMalloc((void **)&d_offt, (n_t + 1) * sizeof(uint32_t)));
            SMH_HIP(hipMalloc((void **)&d_col, (nnz + 4) * sizeof(uint32_t)));
            SMH_HIP(hipMalloc((void **)&d_val, (nnz + 4) * sizeof(V)));
            SMH_HIP(hipMemsetAsync(d_col + nnz, 0, 4 * sizeof(uint32_t), s));
            SMH_HIP(hipMemsetAsync(d_val + nnz, 0, 4 * sizeof(V), s));
        }
        const uint32_t nnz32 = (uint32_t)nnz;
        SMH_HIP(hipMemcpyAsync(d_offt + n_t, &nnz32, sizeof nnz32, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL((k_tb_sort<V, INFO>), dim3((unsigned)n_b), dim3(kTbSortThreads), kTbSortLds, s, d_total, d_src, d_t, d_bval, n_t, shift, d_offt, d_col, d_val, d_sc);
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
    if (rc == SMH_OK && *done && !INFO) { *off_out = d_offt; *col_out = d_col; *val_out = d_val; }
    return rc;
}

}  // namespace

// *done == false with SMH_OK: the matrix does not qualify, nothing was produced
int transpose_bucketed(int dtype, const uint32_t *off, const uint32_t *col, const void *val, size_t n_rows, size_t nnz, uint32_t max_col,
                       uint32_t **off_out, uint32_t **col_out, void **val_out, size_t *n_rows_out, size_t *n_cols_out, bool *done, hipStream_t s) {
    *done = false;
    if (n_rows == 0 || nnz == 0) return SMH_OK;
    *n_rows_out = (size_t)max_col + 1;
    if (dtype == SMH_F64)
        return run<double, false>(off, col, (const double *)val, n_rows, nnz, max_col, off_out, col_out, (double **)val_out, n_cols_out, done, s);
    return run<float, false>(off, col, (const float *)val, n_rows, nnz, max_col, off_out, col_out, (float **)val_out, n_cols_out, done, s);
}

__global__ void __launch_bounds__(kBlock) k_tb_fill(uint32_t *__restrict__ p, uint64_t n, uint32_t v) {
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) p[i] = v;
}

// ColumnIter::assemble_column_info's per-column lists (sparsemat_crs.rs:180-191) by the same passes: the entry indices grouped
// by column, ascending inside a column (= storage order: the stable sort of assemble.hip::column_info).  col_ptr [n_cols + 1]
// and entries [nnz] are the caller's device arrays; *done == false: not applicable (columns without locality), nothing valid
// was written.
int column_lists_bucketed(const uint32_t *off, const uint32_t *col, size_t n_rows, size_t n_cols, size_t nnz, uint32_t max_col, uint32_t *col_ptr,
                          uint32_t *entries, bool *done, hipStream_t s) {
    *done = false;
    if (n_rows == 0 || nnz == 0 || (size_t)max_col >= n_cols) return SMH_OK;
    size_t dummy_cols = 0;
    float *no_val = nullptr;
    SMH_TRY((run<float, true>(off, col, nullptr, n_rows, nnz, max_col, &col_ptr, &entries, &no_val, &dummy_cols, done, s)));
    if (*done && n_cols > (size_t)max_col + 1) {  // columns beyond the last one that occurs: empty lists at the end
        const uint64_t n = n_cols - ((size_t)max_col + 1);
        hipLaunchKernelGGL(k_tb_fill, dim3((unsigned)std::min<uint64_t>((n + kBlock - 1) / kBlock, 4096)), dim3(kBlock), 0, s, col_ptr + (size_t)max_col + 2, n,
                           (uint32_t)nnz);
        SMH_HIP(hipGetLastError());
        SMH_HIP(hipStreamSynchronize(s));
    }
    return SMH_OK;
}

}  // namespace smh
