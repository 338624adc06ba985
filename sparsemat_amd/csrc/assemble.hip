// assemble.hip -- on-device CRS construction, gfx950.
//
// What it replaces.  The reference assembles a matrix by a stream of `add_to(i, j, v)` / `set(i, j, v)` calls
// on a SparseMatIndexList (sparsematrix.rs:226-233 -> get_mut sparsemat_indexlist.rs:158-164 -> find_index
// :29-42 / push :45-53 over IndexList::push indexlist.rs:62-83) and converts it with `to_crs()`
// (sparsemat_indexlist.rs:61-63 -> SparseMatCRS::from_sparsemat_index sparsemat_crs.rs:24-50).  Every call
// walks the row's linked list on one core: O(nnz * row length) pointer chasing.  The resulting CRS is fully
// determined by the stream:
//   * n_rows = largest row + 1, n_cols = largest column + 1 (indexlist.rs:63-65, sparsemat_indexlist.rs:46-48);
//   * a row holds one entry per distinct column, in order of the column's FIRST appearance in the stream;
//   * the entry's value is the left fold of its operations in stream order, starting from T::zero():
//     add_to: acc = acc + v, set: acc = v (one rounding per add).
// Device formulation (integer structure bit-exact, values bit-exact -- the fold is sequential per entry):
//   1. STABLE radix sort (rocPRIM) of the operations by row only (as few 8-bit passes as the row count needs),
//      payload = column << 32 | stream position: a row's operations become contiguous, still in stream order;
//   2a. SHORT ROWS (no row with more than 2048 operations -- every assembly stream: 27 x 8 operations per row for
//      trilinear hexahedra): one thread REPLAYS one row exactly as the reference does -- find the column in the
//      row's list (first 32 entries in LDS), else append -- so the list comes out in first-appearance order with
//      the folded values and nothing else needs sorting; the per-row counts are scanned into the CRS offsets and
//      the lists moved to their places with coalesced stores;
//   2b. LONG ROWS (any stream is handled): segmented sort of the payload inside every row (rocPRIM takes long
//      segments in several passes), run heads -> exclusive scan -> dense entry ids, one thread per run folds it
//      sequentially, then a segmented sort of the entries of a row by first stream position.
// Measured on 134 M operations of a 128^3-cell mesh (profiles/r01_assemble_bench.log): first version -- global
// sorts on row << 32 | column and row << 32 | first position, 13 radix passes over 64-bit keys -- 19.8 ms; this
// one 3 radix passes + the row replay (2.3 ms).
// Also here: Sortable::sort_row (sparsemat_crs.rs:163-172; slice::sort_by is stable) for all rows at once --
// one segmented stable sort by column over the CRS offsets.
// The same pipeline serves streams replayed on a SparseMatCRS itself (smh_crs_replay, transpose, prod: `reverse_rows`
// emits every row's list backwards because push inserts at the row's start, sparsemat_crs.rs:85-87; the container's
// first-push quirk is resolved by the caller from the first two operations, capi.hip), with two shortcuts for
// transpose: `all_set` (no ops array) and `repeats_adjacent` (no (row, column) pair repeats -> the sorted operations
// ARE the entries: k_asm_direct_emit).  Row expansion (k_expand_rows), the column tables of
// ColumnIter::assemble_column_info (one stable sort of the entry indices by column) and append_to_row (the twin entry
// of the quirk) live here too.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include <chrono>

#include "internal.hpp"

namespace smh {

int device_exclusive_scan_u32(uint32_t *data, uint64_t n, hipStream_t s, uint64_t *total_out);  // spmv_colblock.hip

static unsigned grid_for(uint64_t n) {
    uint64_t b = (n + kBlock - 1) / kBlock;
    if (b > 16384) b = 16384;
    return (unsigned)(b ? b : 1);
}

// sort key (row), payload (column << 32 | stream position) + the matrix dimensions (integer max: exact)
__global__ void __launch_bounds__(kBlock)
k_asm_keys(const uint32_t *__restrict__ rows, const uint32_t *__restrict__ cols, uint64_t n, uint32_t *__restrict__ row_key,
           uint64_t *__restrict__ cp, uint32_t *__restrict__ dims /* [0] max row, [1] max column, [2] != 0: the rows are not non-decreasing */) {
    uint32_t mr = 0, mc = 0;
    bool unsorted = false;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
    const uint64_t n4 = (((uintptr_t)rows | (uintptr_t)cols) & 15u) ? 0 : n / 4;  // 16-B loads need aligned arrays
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (uint64_t)gridDim.x * blockDim.x) {
        const u32x4 r = reinterpret_cast<const u32x4 *>(rows)[q], c = reinterpret_cast<const u32x4 *>(cols)[q];
        const uint64_t k = 4 * q;
        reinterpret_cast<u32x4 *>(row_key)[q] = r;
        u64x2 a, b;
        a.x = ((uint64_t)c.x << 32) | (uint32_t)k;       a.y = ((uint64_t)c.y << 32) | (uint32_t)(k + 1);
        b.x = ((uint64_t)c.z << 32) | (uint32_t)(k + 2); b.y = ((uint64_t)c.w << 32) | (uint32_t)(k + 3);
        reinterpret_cast<u64x2 *>(cp)[2 * q] = a;
        reinterpret_cast<u64x2 *>(cp)[2 * q + 1] = b;
        mr = max(max(mr, r.x), max(max(r.y, r.z), r.w));
        mc = max(max(mc, c.x), max(max(c.y, c.z), c.w));
        unsorted |= (r.x > r.y) | (r.y > r.z) | (r.z > r.w) | (q > 0 && rows[k - 1] > r.x);
    }
    for (uint64_t k = 4 * n4 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = rows[k], c = cols[k];
        row_key[k] = r;
        cp[k] = ((uint64_t)c << 32) | (uint32_t)k;
        mr = max(mr, r);
        mc = max(mc, c);
        unsorted |= k > 0 && rows[k - 1] > r;
    }
    if (unsorted) dims[2] = 1u;  // (every writer stores the same value)
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        mr = max(mr, (uint32_t)__shfl_down(mr, o, kWave));
        mc = max(mc, (uint32_t)__shfl_down(mc, o, kWave));
    }
    // one pair of atomics per block (131 072 same-address atomics, one pair per wave, cost 1.5 ms on their own)
    __shared__ uint32_t s_mr[kBlock / kWave], s_mc[kBlock / kWave];
    if ((threadIdx.x & (kWave - 1)) == 0) { s_mr[threadIdx.x / kWave] = mr; s_mc[threadIdx.x / kWave] = mc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(&dims[0], max(max(s_mr[0], s_mr[1]), max(s_mr[2], s_mr[3])));
        atomicMax(&dims[1], max(max(s_mc[0], s_mc[1]), max(s_mc[2], s_mc[3])));
    }
}

// seg[r] = first position whose (sorted) row is >= r, r = 0..n_rows; seg[n_rows] = n
__global__ void __launch_bounds__(kBlock)
k_asm_segments(const uint32_t *__restrict__ row_s, uint64_t n, uint64_t n_rows, uint32_t *__restrict__ seg) {
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = row_s[p];
        const uint64_t prev = p ? (uint64_t)row_s[p - 1] : ~uint64_t(0);
        if (p == 0 || prev != r)
            for (uint64_t q = p ? prev + 1 : 0; q <= r; ++q) seg[q] = (uint32_t)p;
        if (p + 1 == n)
            for (uint64_t q = r + 1; q <= n_rows; ++q) seg[q] = (uint32_t)n;
    }
}

__device__ __forceinline__ bool run_head(const uint32_t *row_s, const uint64_t *cp_s, uint64_t k) {
    return k == 0 || row_s[k] != row_s[k - 1] || (uint32_t)(cp_s[k] >> 32) != (uint32_t)(cp_s[k - 1] >> 32);
}

// operations brought into sorted order (coalesced writes; the fold then reads runs sequentially) + run heads
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_asm_heads(const uint32_t *__restrict__ row_s, const uint64_t *__restrict__ cp_s, const T *__restrict__ vals,
            const uint8_t *__restrict__ ops, uint64_t n, T *__restrict__ vals_s, uint8_t *__restrict__ ops_s,
            uint32_t *__restrict__ head) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t i = (uint32_t)cp_s[k];
        vals_s[k] = vals[i];
        if (ops) ops_s[k] = ops[i];
        head[k] = run_head(row_s, cp_s, k) ? 1u : 0u;
    }
}

// one thread per run: sequential fold in stream order (sparsematrix.rs:226-233 on the entry get_mut returned,
// which push created as T::zero(): sparsemat_indexlist.rs:160-162)
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_asm_fold(const uint32_t *__restrict__ row_s, const uint64_t *__restrict__ cp_s, const T *__restrict__ vals_s,
           const uint8_t *__restrict__ ops_s, const uint32_t *__restrict__ entry_id /* exclusive scan of heads */,
           uint64_t n, bool reverse, bool all_set, uint32_t *__restrict__ first_pos, uint32_t *__restrict__ uidx,
           uint32_t *__restrict__ ucol, T *__restrict__ uval) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        if (!run_head(row_s, cp_s, k)) continue;
        T acc = T(0);
        uint64_t j = k;
        do {
            const T v = vals_s[j];
            if (all_set || (ops_s && ops_s[j])) acc = v;
            else if constexpr (sizeof(T) == 4) acc = __fadd_rn(acc, v);
            else acc = __dadd_rn(acc, v);
            ++j;
        } while (j < n && !run_head(row_s, cp_s, j));
        const uint32_t u = entry_id[k];
        // the run is in stream order: its head is the first appearance (reverse: latest first, as a SparseMatCRS stores)
        first_pos[u] = reverse ? (uint32_t)(n - 1) - (uint32_t)cp_s[k] : (uint32_t)cp_s[k];
        uidx[u] = u;
        ucol[u] = (uint32_t)(cp_s[k] >> 32);
        uval[u] = acc;
    }
}

// CRS offsets: entries before row r = run heads before the row's first operation
__global__ void __launch_bounds__(kBlock)
k_asm_offsets(const uint32_t *__restrict__ seg, const uint32_t *__restrict__ entry_id, uint64_t n, uint64_t n_rows,
              uint32_t n_entries, uint32_t *__restrict__ off) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_rows; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p = seg[r];
        off[r] = p < n ? entry_id[p] : n_entries;
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_asm_emit(const uint32_t *__restrict__ uidx_s, const uint32_t *__restrict__ ucol, const T *__restrict__ uval,
           uint64_t n_entries, uint32_t *__restrict__ col, T *__restrict__ val) {
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_entries; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t u = uidx_s[p];
        col[p] = ucol[u];
        val[p] = uval[u];
    }
}

// ---- short rows: one thread replays one row -------------------------------------------------------------------
// After the sort by row, a row's operations are contiguous and in stream order.  One thread walks them exactly
// like the reference's get_mut (find the column in the row's list, else append): the list comes out in
// first-appearance order with the folded values -- no further sorting.  The first kRowCap entries of the list
// live in LDS (entry-major: conflict-free), longer lists continue in the row's own stretch of the global scratch
// (a row never has more entries than operations).  Used when no row has more than kRowwiseMaxOps operations.
template <typename T>
__device__ __forceinline__ T asm_add(T a, T b) {  // one rounding, never contracted
    if constexpr (sizeof(T) == 4) return __fadd_rn(a, b);
    else return __dadd_rn(a, b);
}

constexpr int kRowBlock = 64;
constexpr int kRowCap = 32;
constexpr uint32_t kRowwiseMaxOps = 2048;

template <typename T>
__global__ void __launch_bounds__(kRowBlock)
k_asm_rowwise(const uint32_t *__restrict__ seg, const uint64_t *__restrict__ cp_s, const T *__restrict__ vals,
              const uint8_t *__restrict__ ops, bool all_set, uint64_t n_rows, uint32_t *lcol, T *lval,
              uint32_t *__restrict__ counts) {
    __shared__ uint32_t s_col[kRowCap][kRowBlock];
    __shared__ T s_val[kRowCap][kRowBlock];
    const uint32_t t = threadIdx.x;
    const uint64_t r = (uint64_t)blockIdx.x * kRowBlock + t;
    if (r >= n_rows) return;  // (no barriers in this kernel)
    const uint64_t a = seg[r], b = seg[r + 1];
    uint32_t cnt = 0;
    for (uint64_t j = a; j < b; ++j) {
        const uint64_t e = cp_s[j];
        const uint32_t c = (uint32_t)(e >> 32), i = (uint32_t)e;
        const T v = vals[i];
        const bool set = all_set || (ops && ops[i]);
        uint32_t f = cnt;  // find_index (sparsemat_indexlist.rs:29-42): first match in list order
        const uint32_t lim = cnt < (uint32_t)kRowCap ? cnt : (uint32_t)kRowCap;
        for (uint32_t q = 0; q < lim; ++q)
            if (s_col[q][t] == c) { f = q; break; }
        if (f == cnt && cnt > (uint32_t)kRowCap)
            for (uint32_t q = kRowCap; q < cnt; ++q)
                if (lcol[a + q] == c) { f = q; break; }
        if (f < cnt) {  // add_to: `+=`, set: `=` (sparsematrix.rs:226-233)
            if (f < (uint32_t)kRowCap) s_val[f][t] = set ? v : asm_add(s_val[f][t], v);
            else lval[a + f] = set ? v : asm_add(lval[a + f], v);
        } else {  // push(i, j, T::zero()) then the operation (sparsemat_indexlist.rs:158-164)
            const T nv = set ? v : asm_add(T(0), v);
            if (cnt < (uint32_t)kRowCap) { s_col[cnt][t] = c; s_val[cnt][t] = nv; }
            else { lcol[a + cnt] = c; lval[a + cnt] = nv; }
            ++cnt;
        }
    }
    const uint32_t lim = cnt < (uint32_t)kRowCap ? cnt : (uint32_t)kRowCap;
    for (uint32_t q = 0; q < lim; ++q) { lcol[a + q] = s_col[q][t]; lval[a + q] = s_val[q][t]; }
    counts[r] = cnt;
    if (r + 1 == n_rows) counts[n_rows] = 0;
}

// row r's list (at seg[r] in the scratch) -> its place in the CRS arrays
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_asm_row_emit(const uint32_t *__restrict__ seg, const uint32_t *__restrict__ off, const uint32_t *__restrict__ lcol,
               const T *__restrict__ lval, uint64_t n_rows, bool reverse, uint32_t *__restrict__ col, T *__restrict__ val) {
    // a block moves the lists of 256 consecutive rows: the output range is walked densely (coalesced stores), the
    // row of an output position found by bisection of the block's 257 offsets held in LDS
    __shared__ uint32_t s_off[kBlock + 1], s_seg[kBlock];
    const uint64_t n_groups = (n_rows + kBlock - 1) / kBlock;
    for (uint64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint64_t r0 = g * kBlock, r1 = r0 + kBlock < n_rows ? r0 + kBlock : n_rows;
        const uint32_t nr = (uint32_t)(r1 - r0);
        __syncthreads();  // (the previous group's tables are no longer read)
        if (threadIdx.x < nr) s_seg[threadIdx.x] = seg[r0 + threadIdx.x];
        if (threadIdx.x <= nr) s_off[threadIdx.x] = off[r0 + threadIdx.x];
        if (threadIdx.x == 0 && nr == (uint32_t)kBlock) s_off[kBlock] = off[r1];  // (off has n_rows + 1 entries)
        __syncthreads();
        const uint32_t base = s_off[0], total = s_off[nr] - base;
        for (uint32_t p = threadIdx.x; p < total; p += kBlock) {
            uint32_t lo = 0, hi = nr;  // last row with s_off[row] - base <= p
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_off[mid] - base <= p) lo = mid; else hi = mid;
            }
            const uint32_t idx = p - (s_off[lo] - base);  // reverse: a SparseMatCRS prepends (sparsemat_crs.rs:85-87)
            const uint64_t src = (uint64_t)s_seg[lo] + (reverse ? s_off[lo + 1] - s_off[lo] - 1 - idx : idx);
            col[(uint64_t)base + p] = lcol[src];
            val[(uint64_t)base + p] = lval[src];
        }
    }
}

// ---- streams without repeated (row, column) pairs: nothing to fold, nothing to search --------------------------
// The caller may guarantee that operations on the same (row, column) are ADJACENT in the row's list (transpose: all
// entries of a source row are consecutive in the stream).  Then one pass over neighbours tells whether any pair
// repeats at all; if none does, every operation is its own entry: offsets = the row segments, and the entries are the
// sorted operations themselves (reversed per row for a SparseMatCRS target).
__global__ void __launch_bounds__(kBlock)
k_asm_adjacent_repeats(const uint32_t *__restrict__ row_s, const uint64_t *__restrict__ cp_s, uint64_t n, uint32_t *flag) {
    for (uint64_t k = 1 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x)
        if (row_s[k] == row_s[k - 1] && (uint32_t)(cp_s[k] >> 32) == (uint32_t)(cp_s[k - 1] >> 32)) *flag = 1u;
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_asm_direct_emit(const uint32_t *__restrict__ seg, const uint64_t *__restrict__ cp_s, const T *__restrict__ vals,
                  const uint8_t *__restrict__ ops, bool all_set, uint64_t n_rows, bool reverse, uint32_t *__restrict__ col,
                  T *__restrict__ val) {
    __shared__ uint32_t s_off[kBlock + 1];
    const uint64_t n_groups = (n_rows + kBlock - 1) / kBlock;
    for (uint64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint64_t r0 = g * kBlock, r1 = r0 + kBlock < n_rows ? r0 + kBlock : n_rows;
        const uint32_t nr = (uint32_t)(r1 - r0);
        __syncthreads();
        if (threadIdx.x <= nr) s_off[threadIdx.x] = seg[r0 + threadIdx.x];
        if (threadIdx.x == 0 && nr == (uint32_t)kBlock) s_off[kBlock] = seg[r1];
        __syncthreads();
        const uint32_t base = s_off[0], total = s_off[nr] - base;
        for (uint32_t p = threadIdx.x; p < total; p += kBlock) {
            uint32_t lo = 0, hi = nr;  // last row with s_off[row] - base <= p
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_off[mid] - base <= p) lo = mid; else hi = mid;
            }
            const uint32_t idx = p - (s_off[lo] - base);
            const uint64_t e = cp_s[reverse ? (uint64_t)s_off[lo + 1] - 1 - idx : (uint64_t)s_off[lo] + idx];
            const uint32_t i = (uint32_t)e;
            const T v = vals[i];
            col[(uint64_t)base + p] = (uint32_t)(e >> 32);
            val[(uint64_t)base + p] = (all_set || (ops && ops[i])) ? v : asm_add(T(0), v);  // push(.., zero) then `=` or `+=`
        }
    }
}

// ---- the value rides the sort (uniform `set` streams without repeats: SparseMatrix::transpose) --------------------------
// Payload of the stable sort by row = (column, VALUE) instead of (column, stream position): the sorted payload then IS the
// result and nothing is gathered afterwards.  (k_asm_direct_emit fetches every value through its stream position: one
// 4-byte value per 128-byte line, 41 GB of line traffic on the C2 transpose, 6.4 of its 21.6 ms.)
template <typename T>
struct AsmPay {
    uint32_t col;
    T val;
};

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_asm_keys_val(const uint32_t *__restrict__ rows, const uint32_t *__restrict__ cols, const T *__restrict__ vals, uint64_t n,
               uint32_t *__restrict__ row_key, AsmPay<T> *__restrict__ pay, uint32_t *__restrict__ dims) {
    uint32_t mr = 0, mc = 0;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = rows[k], c = cols[k];
        row_key[k] = r;
        AsmPay<T> e;
        e.col = c;
        e.val = vals[k];
        pay[k] = e;
        mr = max(mr, r);
        mc = max(mc, c);
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        mr = max(mr, (uint32_t)__shfl_down(mr, o, kWave));
        mc = max(mc, (uint32_t)__shfl_down(mc, o, kWave));
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {  // integer max: exact, order independent
        atomicMax(&dims[0], mr);
        atomicMax(&dims[1], mc);
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_asm_adjacent_repeats_val(const uint32_t *__restrict__ row_s, const AsmPay<T> *__restrict__ pay_s, uint64_t n, uint32_t *flag) {
    for (uint64_t k = 1 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x)
        if (row_s[k] == row_s[k - 1] && pay_s[k].col == pay_s[k - 1].col) *flag = 1u;
}

// sorted operation k of row r goes to position k (to_crs order) or to the mirrored place of its row's range (reverse: a
// SparseMatCRS filled by `set` keeps the entries of a row in reverse order of arrival, sparsemat_crs.rs:85-87)
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_asm_emit_val(const uint32_t *__restrict__ row_s, const uint32_t *__restrict__ seg, const AsmPay<T> *__restrict__ pay_s, uint64_t n,
               bool reverse, uint32_t *__restrict__ col, T *__restrict__ val) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const AsmPay<T> e = pay_s[k];
        uint64_t dst = k;
        if (reverse) {
            const uint32_t r = row_s[k];
            dst = (uint64_t)seg[r] + ((uint64_t)seg[r + 1] - 1 - k);
        }
        col[dst] = e.col;
        val[dst] = e.val;
    }
}

static unsigned bits_for(uint64_t v) {  // bits needed to hold v
    unsigned b = 1;
    while (b < 64 && (v >> b)) ++b;
    return b;
}

// rocPRIM calls: query the temporary storage, allocate, run, synchronise, free
#define SMH_ROCPRIM(call_with_tmp)                                     \
    do {                                                               \
        size_t bytes = 0;                                              \
        void *tmp = nullptr;                                           \
        SMH_HIP(call_with_tmp);                                        \
        SMH_HIP(hipMalloc(&tmp, bytes ? bytes : 16));                  \
        const hipError_t e1 = (call_with_tmp);                         \
        const hipError_t e2 = hipStreamSynchronize(s);                 \
        (void)hipFree(tmp);                                            \
        SMH_HIP(e1);                                                   \
        SMH_HIP(e2);                                                   \
    } while (0)

// device buffers freed on scope exit
struct Scratch {
    void *p[20] = {};
    int n = 0;
    template <typename U> int alloc(U **out, size_t count) {
        SMH_HIP(hipMalloc((void **)out, (count ? count : 1) * sizeof(U)));
        p[n++] = *out;
        return SMH_OK;
    }
    ~Scratch() { for (int i = 0; i < n; ++i) (void)hipFree(p[i]); }
};

template <typename T>
static int assemble_t(uint64_t n, const uint32_t *rows, const uint32_t *cols, const T *vals, const uint8_t *ops, bool reverse,
                      bool all_set, bool repeats_adjacent,
                      size_t *n_rows_out, size_t *n_cols_out, size_t *nnz_out, uint32_t **off_out, uint32_t **col_out,
                      T **val_out, hipStream_t s) {
    // SMH_ASSEMBLE_TIMING=1: wall time of every stage on stderr (development aid)
    static const bool timing = getenv("SMH_ASSEMBLE_TIMING") && atoi(getenv("SMH_ASSEMBLE_TIMING")) != 0;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[assemble] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    if (all_set) ops = nullptr;  // every operation is `set`: no array to consult
    static const bool allow_value_payload = !(getenv("SMH_ASSEMBLE_VALUE_PAYLOAD") && atoi(getenv("SMH_ASSEMBLE_VALUE_PAYLOAD")) == 0);
    if (all_set && repeats_adjacent && allow_value_payload && n > 0) {
        // Transposition-shaped streams: sort (row) -> (column, value) and emit -- unless two neighbours turn out to name
        // the same (row, column), in which case the general route below starts over with stream positions.
        Scratch bufs;
        uint32_t *dims = nullptr, *row_key = nullptr, *row_s = nullptr, *seg = nullptr;
        AsmPay<T> *pay = nullptr, *pay_s = nullptr;
        SMH_TRY(bufs.alloc(&dims, 2));
        SMH_HIP(hipMemsetAsync(dims, 0, 2 * sizeof(uint32_t), s));
        SMH_TRY(bufs.alloc(&row_key, n));
        SMH_TRY(bufs.alloc(&row_s, n));
        SMH_TRY(bufs.alloc(&pay, n));
        SMH_TRY(bufs.alloc(&pay_s, n));
        hipLaunchKernelGGL((k_asm_keys_val<T>), dim3(grid_for(n) < 4096u ? grid_for(n) : 4096u), dim3(kBlock), 0, s, rows, cols, vals, n, row_key,
                           pay, dims);
        SMH_HIP(hipGetLastError());
        uint32_t h_dims[2];
        SMH_HIP(hipMemcpyAsync(h_dims, dims, sizeof h_dims, hipMemcpyDeviceToHost, s));
        SMH_HIP(hipStreamSynchronize(s));
        lap("keys (value payload)");
        const uint64_t n_rows = (uint64_t)h_dims[0] + 1, n_cols = (uint64_t)h_dims[1] + 1;
        SMH_ROCPRIM(rocprim::radix_sort_pairs(tmp, bytes, row_key, row_s, pay, pay_s, (size_t)n, 0u, bits_for(h_dims[0]), s));
        lap("sort by row (value payload)");
        SMH_TRY(bufs.alloc(&seg, n_rows + 1));
        hipLaunchKernelGGL(k_asm_segments, dim3(grid_for(n)), dim3(kBlock), 0, s, row_s, n, n_rows, seg);
        uint32_t repeats = 0;
        SMH_HIP(hipMemsetAsync(dims, 0, sizeof(uint32_t), s));
        hipLaunchKernelGGL((k_asm_adjacent_repeats_val<T>), dim3(grid_for(n)), dim3(kBlock), 0, s, row_s, pay_s, n, dims);
        SMH_HIP(hipGetLastError());
        SMH_HIP(hipMemcpyAsync(&repeats, dims, sizeof repeats, hipMemcpyDeviceToHost, s));
        SMH_HIP(hipStreamSynchronize(s));
        lap("segments, repeats");
        if (!repeats) {
            uint32_t *off = nullptr, *col = nullptr;
            T *val = nullptr;
            auto go = [&]() -> int {
                SMH_HIP(hipMalloc((void **)&off, (n_rows + 1) * sizeof(uint32_t)));
                SMH_HIP(hipMalloc((void **)&col, (n + 4) * sizeof(uint32_t)));
                SMH_HIP(hipMalloc((void **)&val, (n + 4) * sizeof(T)));
                SMH_HIP(hipMemcpyAsync(off, seg, (n_rows + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
                SMH_HIP(hipMemsetAsync(col + n, 0, 4 * sizeof(uint32_t), s));
                SMH_HIP(hipMemsetAsync(val + n, 0, 4 * sizeof(T), s));
                hipLaunchKernelGGL((k_asm_emit_val<T>), dim3(grid_for(n)), dim3(kBlock), 0, s, row_s, seg, pay_s, n, reverse, col, val);
                SMH_HIP(hipGetLastError());
                SMH_HIP(hipStreamSynchronize(s));
                lap("emit (value payload)");
                return SMH_OK;
            };
            const int rc = go();
            if (rc != SMH_OK) {
                (void)hipFree(off); (void)hipFree(col); (void)hipFree(val);
                return rc;
            }
            *n_rows_out = (size_t)n_rows; *n_cols_out = (size_t)n_cols; *nnz_out = (size_t)n;
            *off_out = off; *col_out = col; *val_out = val;
            return SMH_OK;
        }
    }
    Scratch tmp_bufs;
    uint32_t *dims = nullptr, *row_key = nullptr, *row_s = nullptr, *seg = nullptr, *head = nullptr;
    uint64_t *cp = nullptr, *cp_s = nullptr;
    T *vals_s = nullptr;
    uint8_t *ops_s = nullptr;
    SMH_TRY(tmp_bufs.alloc(&dims, 3));
    SMH_HIP(hipMemsetAsync(dims, 0, 3 * sizeof(uint32_t), s));
    SMH_TRY(tmp_bufs.alloc(&row_key, n));
    SMH_TRY(tmp_bufs.alloc(&cp, n));
    lap("scratch allocation");
    hipLaunchKernelGGL(k_asm_keys, dim3(grid_for(n) < 4096u ? grid_for(n) : 4096u), dim3(kBlock), 0, s, rows, cols, n, row_key, cp, dims);
    SMH_HIP(hipGetLastError());
    lap("keys");
    uint32_t h_dims[3];
    SMH_HIP(hipMemcpyAsync(h_dims, dims, sizeof h_dims, hipMemcpyDeviceToHost, s));
    SMH_HIP(hipStreamSynchronize(s));
    const uint64_t n_rows = (uint64_t)h_dims[0] + 1, n_cols = (uint64_t)h_dims[1] + 1;
    // 1. by row (stable): cp_s holds the rows' operations in stream order.  A stream whose rows never decrease -- a row-major
    // assembly loop, the products of SparseMatrix::prod -- is in that order already: nothing to sort (3 of 7 ms per prod batch)
    const bool allow_presorted = !(getenv("SMH_ASSEMBLE_PRESORTED") && atoi(getenv("SMH_ASSEMBLE_PRESORTED")) == 0);
    if (allow_presorted && !h_dims[2]) {
        row_s = row_key;
        cp_s = cp;
        lap("sort by row (skipped: rows non-decreasing)");
    } else {
        SMH_TRY(tmp_bufs.alloc(&row_s, n));
        SMH_TRY(tmp_bufs.alloc(&cp_s, n));
        SMH_ROCPRIM(rocprim::radix_sort_pairs(tmp, bytes, row_key, row_s, cp, cp_s, (size_t)n, 0u, bits_for(h_dims[0]), s));
        lap("sort by row");
    }
    // 2. inside every row by (column, stream position); output back into `cp`
    SMH_TRY(tmp_bufs.alloc(&seg, n_rows + 1));
    hipLaunchKernelGGL(k_asm_segments, dim3(grid_for(n)), dim3(kBlock), 0, s, row_s, n, n_rows, seg);
    SMH_HIP(hipGetLastError());
    // longest row (in operations) decides the route
    uint32_t max_ops = 0;
    SMH_TRY(launch_stream_max_tile(seg, n_rows, 1, dims, s));
    SMH_HIP(hipMemcpyAsync(&max_ops, dims, sizeof max_ops, hipMemcpyDeviceToHost, s));
    SMH_HIP(hipStreamSynchronize(s));
    static const bool allow_rowwise = !(getenv("SMH_ASSEMBLE_ROWWISE") && atoi(getenv("SMH_ASSEMBLE_ROWWISE")) == 0);
    static const bool allow_direct = !(getenv("SMH_ASSEMBLE_DIRECT") && atoi(getenv("SMH_ASSEMBLE_DIRECT")) == 0);
    if (repeats_adjacent && allow_direct) {
        uint32_t repeats = 0;
        SMH_HIP(hipMemsetAsync(dims, 0, sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_asm_adjacent_repeats, dim3(grid_for(n)), dim3(kBlock), 0, s, row_s, cp_s, n, dims);
        SMH_HIP(hipGetLastError());
        SMH_HIP(hipMemcpyAsync(&repeats, dims, sizeof repeats, hipMemcpyDeviceToHost, s));
        SMH_HIP(hipStreamSynchronize(s));
        lap("segments, longest row, repeats");
        if (!repeats) {  // every operation is an entry: offsets = segments, entries = the sorted operations
            uint32_t *off = nullptr, *col = nullptr;
            T *val = nullptr;
            auto go = [&]() -> int {
                SMH_HIP(hipMalloc((void **)&off, (n_rows + 1) * sizeof(uint32_t)));
                SMH_HIP(hipMalloc((void **)&col, (n + 4) * sizeof(uint32_t)));
                SMH_HIP(hipMalloc((void **)&val, (n + 4) * sizeof(T)));
                SMH_HIP(hipMemcpyAsync(off, seg, (n_rows + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
                SMH_HIP(hipMemsetAsync(col + n, 0, 4 * sizeof(uint32_t), s));
                SMH_HIP(hipMemsetAsync(val + n, 0, 4 * sizeof(T), s));
                hipLaunchKernelGGL((k_asm_direct_emit<T>), dim3(grid_for(n_rows)), dim3(kBlock), 0, s, seg, cp_s, vals, ops, all_set, n_rows,
                                   reverse, col, val);
                SMH_HIP(hipGetLastError());
                SMH_HIP(hipStreamSynchronize(s));
                lap("direct emit");
                return SMH_OK;
            };
            const int rc = go();
            if (rc != SMH_OK) {
                (void)hipFree(off); (void)hipFree(col); (void)hipFree(val);
                return rc;
            }
            *n_rows_out = (size_t)n_rows; *n_cols_out = (size_t)n_cols; *nnz_out = (size_t)n;
            *off_out = off; *col_out = col; *val_out = val;
            return SMH_OK;
        }
    }
    if (allow_rowwise && max_ops <= kRowwiseMaxOps) {
        // short rows: one thread replays a row; lists land in `row_key` (columns) / a value scratch at seg[r]
        uint32_t *lcol = row_key;  // dead after the sort
        T *lval = nullptr;
        SMH_TRY(tmp_bufs.alloc(&lval, n));
        uint32_t *off = nullptr, *col = nullptr;
        T *val = nullptr;
        uint64_t n_entries = 0;
        lap("segments, longest row");
        auto go = [&]() -> int {
            SMH_HIP(hipMalloc((void **)&off, (n_rows + 1) * sizeof(uint32_t)));
            hipLaunchKernelGGL((k_asm_rowwise<T>), dim3((unsigned)((n_rows + kRowBlock - 1) / kRowBlock)), dim3(kRowBlock), 0, s, seg, cp_s,
                               vals, ops, all_set, n_rows, lcol, lval, off);
            SMH_HIP(hipGetLastError());
            lap("row replay");
            SMH_TRY(device_exclusive_scan_u32(off, n_rows + 1, s, &n_entries));  // counts -> CRS offsets
            SMH_HIP(hipMalloc((void **)&col, (n_entries + 4) * sizeof(uint32_t)));
            SMH_HIP(hipMalloc((void **)&val, (n_entries + 4) * sizeof(T)));
            SMH_HIP(hipMemsetAsync(col + n_entries, 0, 4 * sizeof(uint32_t), s));
            SMH_HIP(hipMemsetAsync(val + n_entries, 0, 4 * sizeof(T), s));
            lap("offsets, result allocation");
            hipLaunchKernelGGL((k_asm_row_emit<T>), dim3(grid_for(n_rows)), dim3(kBlock), 0, s, seg, off, lcol, lval, n_rows, reverse, col, val);
            SMH_HIP(hipGetLastError());
            SMH_HIP(hipStreamSynchronize(s));
            lap("emit");
            return SMH_OK;
        };
        const int rc = go();
        if (rc != SMH_OK) {
            (void)hipFree(off); (void)hipFree(col); (void)hipFree(val);
            return rc;
        }
        *n_rows_out = (size_t)n_rows; *n_cols_out = (size_t)n_cols; *nnz_out = (size_t)n_entries;
        *off_out = off; *col_out = col; *val_out = val;
        return SMH_OK;
    }
    // long rows: segmented sorts (rocPRIM takes long segments in several passes)
    uint64_t *runs_buf = cp;
    if (cp_s == cp) SMH_TRY(tmp_bufs.alloc(&runs_buf, n));  // (presorted stream: cp is the sort's input, not a free buffer)
    SMH_ROCPRIM(rocprim::segmented_radix_sort_keys(tmp, bytes, cp_s, runs_buf, (unsigned)n, (unsigned)n_rows, seg, seg + 1, 0u,
                                                   32u + bits_for(h_dims[1]), s));
    const uint64_t *runs = runs_buf;  // (row_s[k], runs[k]) ascending in (row, column, stream position)
    // 3. runs -> entries
    SMH_TRY(tmp_bufs.alloc(&head, n));
    SMH_TRY(tmp_bufs.alloc(&vals_s, n));
    if (ops) SMH_TRY(tmp_bufs.alloc(&ops_s, n));
    hipLaunchKernelGGL((k_asm_heads<T>), dim3(grid_for(n)), dim3(kBlock), 0, s, row_s, runs, vals, ops, n, vals_s, ops_s, head);
    SMH_HIP(hipGetLastError());
    uint64_t n_entries = 0;
    SMH_TRY(device_exclusive_scan_u32(head, n, s, &n_entries));
    uint32_t *first_pos = nullptr, *first_pos_s = nullptr, *uidx = nullptr, *uidx_s = nullptr, *ucol = nullptr;
    T *uval = nullptr;
    SMH_TRY(tmp_bufs.alloc(&first_pos, n_entries));
    SMH_TRY(tmp_bufs.alloc(&first_pos_s, n_entries));
    SMH_TRY(tmp_bufs.alloc(&uidx, n_entries));
    SMH_TRY(tmp_bufs.alloc(&uidx_s, n_entries));
    SMH_TRY(tmp_bufs.alloc(&ucol, n_entries));
    SMH_TRY(tmp_bufs.alloc(&uval, n_entries));
    hipLaunchKernelGGL((k_asm_fold<T>), dim3(grid_for(n)), dim3(kBlock), 0, s, row_s, runs, vals_s, ops_s, head, n, reverse, all_set, first_pos,
                       uidx, ucol, uval);
    SMH_HIP(hipGetLastError());
    // result arrays (owned by the caller; padded like smh_crs_create's)
    uint32_t *off = nullptr, *col = nullptr;
    T *val = nullptr;
    auto finish = [&]() -> int {
        SMH_HIP(hipMalloc((void **)&off, (n_rows + 1) * sizeof(uint32_t)));
        SMH_HIP(hipMalloc((void **)&col, (n_entries + 4) * sizeof(uint32_t)));
        SMH_HIP(hipMalloc((void **)&val, (n_entries + 4) * sizeof(T)));
        SMH_HIP(hipMemsetAsync(col + n_entries, 0, 4 * sizeof(uint32_t), s));
        SMH_HIP(hipMemsetAsync(val + n_entries, 0, 4 * sizeof(T), s));
        hipLaunchKernelGGL(k_asm_offsets, dim3(grid_for(n_rows + 1)), dim3(kBlock), 0, s, seg, head, n, n_rows, (uint32_t)n_entries, off);
        SMH_HIP(hipGetLastError());
        // 4. first-appearance order inside every row
        SMH_ROCPRIM(rocprim::segmented_radix_sort_pairs(tmp, bytes, first_pos, first_pos_s, uidx, uidx_s, (unsigned)n_entries,
                                                        (unsigned)n_rows, off, off + 1, 0u, bits_for(n - 1), s));
        hipLaunchKernelGGL((k_asm_emit<T>), dim3(grid_for(n_entries)), dim3(kBlock), 0, s, uidx_s, ucol, uval, n_entries, col, val);
        SMH_HIP(hipGetLastError());
        SMH_HIP(hipStreamSynchronize(s));
        return SMH_OK;
    };
    const int rc = finish();
    if (rc != SMH_OK) {
        (void)hipFree(off); (void)hipFree(col); (void)hipFree(val);
        return rc;
    }
    *n_rows_out = (size_t)n_rows; *n_cols_out = (size_t)n_cols; *nnz_out = (size_t)n_entries;
    *off_out = off; *col_out = col; *val_out = val;
    return SMH_OK;
}

// rows/cols/vals/ops: DEVICE arrays of n operations (ops may be null: all add_to).  n >= 1.
// all_set: every operation is `set` (ops ignored).  repeats_adjacent: the caller guarantees that operations on the same (row,
// column) are neighbours in the row's list (transpose) -- streams without any repeat then skip the replay.
// reverse_rows: every row in REVERSE order of first appearance -- the layout the same stream leaves in a SparseMatCRS, whose push
// inserts at the start of the row (sparsemat_crs.rs:85-87); the first-push quirk is the caller's business (capi.hip).
int assemble_triplets(int dtype, size_t n, const uint32_t *rows, const uint32_t *cols, const void *vals, const uint8_t *ops,
                      bool reverse_rows, bool all_set, bool repeats_adjacent, size_t *n_rows_out, size_t *n_cols_out, size_t *nnz_out,
                      uint32_t **off_out, uint32_t **col_out,
                      void **val_out, hipStream_t s) {
    if (dtype == SMH_F64)
        return assemble_t<double>(n, rows, cols, (const double *)vals, ops, reverse_rows, all_set, repeats_adjacent, n_rows_out, n_cols_out, nnz_out, off_out, col_out,
                                  (double **)val_out, s);
    return assemble_t<float>(n, rows, cols, (const float *)vals, ops, reverse_rows, all_set, repeats_adjacent, n_rows_out, n_cols_out, nnz_out, off_out, col_out,
                             (float **)val_out, s);
}

// ---- sort_row for every row --------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_iota(uint32_t *__restrict__ idx, uint64_t n) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) idx[k] = (uint32_t)k;
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_sortrows_gather(const uint32_t *__restrict__ idx_s, const T *__restrict__ val, uint64_t nnz, T *__restrict__ val_s) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (uint64_t)gridDim.x * blockDim.x)
        val_s[k] = val[idx_s[k]];
}

int sort_rows(int dtype, const uint32_t *off, uint32_t *col, void *val, size_t n_rows, size_t nnz, uint32_t max_col,
              hipStream_t s) {
    if (nnz == 0 || n_rows == 0) return SMH_OK;
    Scratch tmp_bufs;
    uint32_t *col_s = nullptr, *idx = nullptr, *idx_s = nullptr;
    char *val_s = nullptr;
    const size_t vs = dtype_size(dtype);
    SMH_TRY(tmp_bufs.alloc(&col_s, nnz));
    SMH_TRY(tmp_bufs.alloc(&idx, nnz));
    SMH_TRY(tmp_bufs.alloc(&idx_s, nnz));
    SMH_TRY(tmp_bufs.alloc(&val_s, nnz * vs));
    hipLaunchKernelGGL(k_iota, dim3(grid_for(nnz)), dim3(kBlock), 0, s, idx, (uint64_t)nnz);
    SMH_HIP(hipGetLastError());
    // the CRS offsets are the segments; stable, so duplicates of a column keep their storage order
    SMH_ROCPRIM(rocprim::segmented_radix_sort_pairs(tmp, bytes, col, col_s, idx, idx_s, (unsigned)nnz, (unsigned)n_rows, off, off + 1,
                                                    0u, bits_for(max_col), s));
    if (dtype == SMH_F64)
        hipLaunchKernelGGL((k_sortrows_gather<double>), dim3(grid_for(nnz)), dim3(kBlock), 0, s, idx_s, (const double *)val,
                           (uint64_t)nnz, (double *)val_s);
    else
        hipLaunchKernelGGL((k_sortrows_gather<float>), dim3(grid_for(nnz)), dim3(kBlock), 0, s, idx_s, (const float *)val,
                           (uint64_t)nnz, (float *)val_s);
    SMH_HIP(hipGetLastError());
    SMH_HIP(hipMemcpyAsync(col, col_s, nnz * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    SMH_HIP(hipMemcpyAsync(val, val_s, nnz * vs, hipMemcpyDeviceToDevice, s));
    SMH_HIP(hipStreamSynchronize(s));
    return SMH_OK;
}

// ---- row of every entry, transpose, column tables ------------------------------------------------------------
// rows[k] = row of entry k (what assemble_column_info pushes, sparsemat_crs.rs:186-189).  A block takes 256
// consecutive rows and walks their entries densely (coalesced stores), bisecting its 257 offsets in LDS.
__global__ void __launch_bounds__(kBlock)
k_expand_rows(const uint32_t *__restrict__ off, uint64_t n_rows, uint32_t *__restrict__ rows) {
    __shared__ uint32_t s_off[kBlock + 1];
    const uint64_t n_groups = (n_rows + kBlock - 1) / kBlock;
    for (uint64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint64_t r0 = g * kBlock, r1 = r0 + kBlock < n_rows ? r0 + kBlock : n_rows;
        const uint32_t nr = (uint32_t)(r1 - r0);
        __syncthreads();
        if (threadIdx.x <= nr) s_off[threadIdx.x] = off[r0 + threadIdx.x];
        if (threadIdx.x == 0 && nr == (uint32_t)kBlock) s_off[kBlock] = off[r1];
        __syncthreads();
        const uint32_t base = s_off[0], total = s_off[nr] - base;
        for (uint32_t p = threadIdx.x; p < total; p += kBlock) {
            uint32_t lo = 0, hi = nr;  // last row with s_off[row] - base <= p
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_off[mid] - base <= p) lo = mid; else hi = mid;
            }
            rows[(uint64_t)base + p] = (uint32_t)(r0 + lo);
        }
    }
}

int expand_rows(const uint32_t *off, size_t n_rows, uint32_t *rows_out, hipStream_t s) {
    if (n_rows == 0) return SMH_OK;
    hipLaunchKernelGGL(k_expand_rows, dim3(grid_for(n_rows)), dim3(kBlock), 0, s, off, (uint64_t)n_rows, rows_out);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

__global__ void __launch_bounds__(kBlock)
k_bump_offsets(uint32_t *__restrict__ off, uint64_t first, uint64_t last) {  // off[first..last] += 1
    for (uint64_t r = first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= last; r += (uint64_t)gridDim.x * blockDim.x) off[r] += 1u;
}

// One more entry at the END of row `row` (new arrays, the old ones are freed): the place the first-push quirk of a
// SparseMatCRS leaves the first operation's entry in when the second operation names the same (row, column)
// (sparsemat_crs.rs:75-81; capi.hip::assemble_common).
int append_to_row(int dtype, uint32_t *off, uint32_t **col, void **val, size_t n_rows, size_t *nnz, size_t row, uint32_t column,
                  const void *value_host, hipStream_t s) {
    const size_t vs = dtype_size(dtype), n = *nnz;
    uint32_t p = 0;
    SMH_HIP(hipMemcpyAsync(&p, off + row + 1, sizeof p, hipMemcpyDeviceToHost, s));
    SMH_HIP(hipStreamSynchronize(s));
    uint32_t *ncol = nullptr;
    char *nval = nullptr;
    SMH_HIP(hipMalloc((void **)&ncol, (n + 1 + 4) * sizeof(uint32_t)));
    if (hipMalloc((void **)&nval, (n + 1 + 4) * vs) != hipSuccess) { (void)hipFree(ncol); return fail(SMH_ERR_OOM, "hipMalloc failed"); }
    auto go = [&]() -> int {
        SMH_HIP(hipMemsetAsync(ncol + n + 1, 0, 4 * sizeof(uint32_t), s));
        SMH_HIP(hipMemsetAsync(nval + (n + 1) * vs, 0, 4 * vs, s));
        if (p) {
            SMH_HIP(hipMemcpyAsync(ncol, *col, (size_t)p * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
            SMH_HIP(hipMemcpyAsync(nval, *val, (size_t)p * vs, hipMemcpyDeviceToDevice, s));
        }
        SMH_HIP(hipMemcpyAsync(ncol + p, &column, sizeof column, hipMemcpyHostToDevice, s));
        SMH_HIP(hipMemcpyAsync(nval + (size_t)p * vs, value_host, vs, hipMemcpyHostToDevice, s));
        if (n > p) {
            SMH_HIP(hipMemcpyAsync(ncol + p + 1, *col + p, (n - p) * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
            SMH_HIP(hipMemcpyAsync(nval + ((size_t)p + 1) * vs, (const char *)*val + (size_t)p * vs, (n - p) * vs, hipMemcpyDeviceToDevice, s));
        }
        hipLaunchKernelGGL(k_bump_offsets, dim3(grid_for(n_rows - row)), dim3(kBlock), 0, s, off, (uint64_t)row + 1, (uint64_t)n_rows);
        SMH_HIP(hipGetLastError());
        SMH_HIP(hipStreamSynchronize(s));
        return SMH_OK;
    };
    const int rc = go();
    if (rc != SMH_OK) { (void)hipFree(ncol); (void)hipFree(nval); return rc; }
    (void)hipFree(*col); (void)hipFree(*val);
    *col = ncol; *val = nval; *nnz = n + 1;
    return SMH_OK;
}

// ColumnIter::assemble_column_info (sparsemat_crs.rs:180-191) as arrays: rows[k] = row of entry k; the per-column
// lists the reference keeps as an IndexList (entry indices in storage order: indexlist.rs:62-83 appends at the
// list's tail) = a STABLE sort of the entry indices by column, col_ptr[j]..col_ptr[j+1] delimiting column j.
// All three outputs are device arrays: rows[nnz], col_ptr[n_cols + 1], entries[nnz].  Columns must be < n_cols.
int column_info(const uint32_t *off, const uint32_t *col, size_t n_rows, size_t n_cols, size_t nnz, uint32_t max_col, uint32_t *rows,
                uint32_t *col_ptr, uint32_t *entries, hipStream_t s) {
    if (nnz == 0) {
        SMH_HIP(hipMemsetAsync(col_ptr, 0, (n_cols + 1) * sizeof(uint32_t), s));
        SMH_HIP(hipStreamSynchronize(s));
        return SMH_OK;
    }
    SMH_TRY(expand_rows(off, n_rows, rows, s));
    // matrices with local structure: the lists by two bucketed passes instead of the device-wide sort (transpose_bucket.hip;
    // the same arrays bit for bit, tests/test_transpose_gpu.py runs both); SMH_COLUMN_INFO_BUCKETED=0 keeps the sort
    if (!(getenv("SMH_COLUMN_INFO_BUCKETED") && atoi(getenv("SMH_COLUMN_INFO_BUCKETED")) == 0)) {
        bool done = false;
        SMH_TRY(column_lists_bucketed(off, col, n_rows, n_cols, nnz, max_col, col_ptr, entries, &done, s));
        if (done) {
            SMH_HIP(hipStreamSynchronize(s));
            return SMH_OK;
        }
    }
    Scratch tmp_bufs;
    uint32_t *col_s = nullptr, *idx = nullptr;
    SMH_TRY(tmp_bufs.alloc(&col_s, nnz));
    SMH_TRY(tmp_bufs.alloc(&idx, nnz));
    hipLaunchKernelGGL(k_iota, dim3(grid_for(nnz)), dim3(kBlock), 0, s, idx, (uint64_t)nnz);
    SMH_HIP(hipGetLastError());
    SMH_ROCPRIM(rocprim::radix_sort_pairs(tmp, bytes, col, col_s, idx, entries, nnz, 0u, bits_for(max_col), s));
    hipLaunchKernelGGL(k_asm_segments, dim3(grid_for(nnz)), dim3(kBlock), 0, s, col_s, (uint64_t)nnz, (uint64_t)n_cols, col_ptr);
    SMH_HIP(hipGetLastError());
    SMH_HIP(hipStreamSynchronize(s));
    return SMH_OK;
}

}  // namespace smh
