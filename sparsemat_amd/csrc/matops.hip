// matops.hip -- structure predicates and SparseMatrix::prod on the device, gfx950.
//
// prod (sparsematrix.rs:186-210) in the reference: for every row i of self (entries stably sorted by column) and every
// column j of rhs in ascending order, `sum` runs over rhs' column j in list order (= storage order, ascending entry
// index: assemble_column_info sparsemat_crs.rs:180-191) and, for each of its (row, val_rhs), over the entries of the
// sorted row with col == row: sum += val * val_rhs; sums != 0 go through ret.set(i, j, sum) into a fresh SparseMatCRS.
// O(n_rows * n_cols) loop trips whatever the sparsity.
// Device formulation, bit-exact (same products, same order of additions for every (i, j)):
//   * row-wise expansion: for row i walk its column groups k ascending (a group = the duplicates of one column, in
//     storage order), for every entry e of rhs' row k in storage order, for every member d of the group: one product
//     val_d * val_e (rounded) for output column col_e.  For a fixed (i, j) this lists the products ordered by rhs entry
//     index, then by position in the sorted row -- the order the reference adds them in;
//   * the products of a batch of rows (<= 2^27 by default) are an add_to stream for the assembly machinery
//     (assemble.hip): stable sort by row, sequential fold per (row, column) from zero (0 + p = p, and -0 becomes +0 as
//     in `sum = zero; sum += p`);
//   * rows sorted by column, sums == 0 dropped (also -0 and exact cancellations; NaN stays: NaN != 0), every row
//     reversed (ascending set calls into a SparseMatCRS leave descending columns: push prepends, sparsemat_crs.rs:85-87),
//     n_rows / n_cols = largest row / column with a kept sum + 1; a single kept sum leaves a matrix without rows (first
//     push quirk :75-76), the quirk's other two forms cannot occur for ascending (i, j).
#include "internal.hpp"

#include <chrono>
#include <memory>
#include <new>
#include <vector>

namespace smh {

int device_exclusive_scan_u32(uint32_t *data, uint64_t n, hipStream_t s, uint64_t *total_out);  // spmv_colblock.hip

static unsigned mat_grid(uint64_t n) {
    uint64_t b = (n + kBlock - 1) / kBlock;
    if (b > 16384) b = 16384;
    return (unsigned)(b ? b : 1);
}

// ---- predicates ------------------------------------------------------------------------------------------------
// is_sorted (sparsematrix.rs:251-271): no column smaller than its predecessor in the row
__global__ void __launch_bounds__(kBlock)
k_is_sorted(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, uint64_t n_rows, uint32_t *flag) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t a = off[r], b = off[r + 1];
        for (uint64_t q = a + 1; q < b; ++q)
            if (col[q] < col[q - 1]) { *flag = 0u; break; }
    }
}

// is_symmetric (sparsematrix.rs:212-222): for every stored (i, j, val): get(j, i) != val -> false; get = first match in
// row j's storage order, zero when absent or j >= n_rows (sparsemat_crs.rs:54-67,136-142).  NaN != NaN: not symmetric.
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_is_symmetric(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const T *__restrict__ val, uint64_t n_rows,
               uint32_t *flag) {
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
            uint32_t lo = 0, hi = nr;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_off[mid] - base <= p) lo = mid; else hi = mid;
            }
            const uint64_t i = r0 + lo, e = (uint64_t)base + p;
            const uint64_t j = col[e];
            T w = T(0);
            if (j < n_rows)
                for (uint64_t q = off[j]; q < off[j + 1]; ++q)
                    if (col[q] == i) { w = val[q]; break; }
            if (w != val[e]) *flag = 0u;
        }
    }
}

int crs_is_sorted(const uint32_t *off, const uint32_t *col, size_t n_rows, int *out, hipStream_t s) {
    uint32_t *flag = nullptr, h = 1;
    SMH_HIP(hipMalloc((void **)&flag, sizeof(uint32_t)));
    auto go = [&]() -> int {
        SMH_HIP(hipMemcpyAsync(flag, &h, sizeof h, hipMemcpyHostToDevice, s));
        if (n_rows) hipLaunchKernelGGL(k_is_sorted, dim3(mat_grid(n_rows)), dim3(kBlock), 0, s, off, col, (uint64_t)n_rows, flag);
        SMH_HIP(hipGetLastError());
        SMH_HIP(hipMemcpyAsync(&h, flag, sizeof h, hipMemcpyDeviceToHost, s));
        SMH_HIP(hipStreamSynchronize(s));
        return SMH_OK;
    };
    const int rc = go();
    (void)hipFree(flag);
    *out = (int)h;
    return rc;
}

int crs_is_symmetric(int dtype, const uint32_t *off, const uint32_t *col, const void *val, size_t n_rows, int *out, hipStream_t s) {
    uint32_t *flag = nullptr, h = 1;
    SMH_HIP(hipMalloc((void **)&flag, sizeof(uint32_t)));
    auto go = [&]() -> int {
        SMH_HIP(hipMemcpyAsync(flag, &h, sizeof h, hipMemcpyHostToDevice, s));
        if (n_rows) {
            if (dtype == SMH_F64)
                hipLaunchKernelGGL((k_is_symmetric<double>), dim3(mat_grid(n_rows)), dim3(kBlock), 0, s, off, col, (const double *)val,
                                   (uint64_t)n_rows, flag);
            else
                hipLaunchKernelGGL((k_is_symmetric<float>), dim3(mat_grid(n_rows)), dim3(kBlock), 0, s, off, col, (const float *)val,
                                   (uint64_t)n_rows, flag);
        }
        SMH_HIP(hipGetLastError());
        SMH_HIP(hipMemcpyAsync(&h, flag, sizeof h, hipMemcpyDeviceToHost, s));
        SMH_HIP(hipStreamSynchronize(s));
        return SMH_OK;
    };
    const int rc = go();
    (void)hipFree(flag);
    *out = (int)h;
    return rc;
}

// ---- prod ------------------------------------------------------------------------------------------------------
// products one entry of the (sorted) left operand generates = length of the right operand's row it names
__global__ void __launch_bounds__(kBlock)
k_prod_counts(const uint32_t *__restrict__ a_col, uint64_t a_nnz, const uint32_t *__restrict__ b_off, uint64_t b_rows,
              uint32_t *__restrict__ cnt) {
    for (uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < a_nnz; d += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k = a_col[d];
        cnt[d] = k < b_rows ? b_off[k + 1] - b_off[k] : 0u;
    }
}

__global__ void __launch_bounds__(kBlock)
k_prod_row_sums(const uint32_t *__restrict__ a_off, const uint32_t *__restrict__ cnt, uint64_t n_rows, uint32_t *__restrict__ row_sum) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t sum = 0;
        for (uint64_t q = a_off[r]; q < a_off[r + 1]; ++q) sum += cnt[q];
        row_sum[r] = sum < 0xFFFFFFFFull ? (uint32_t)sum : 0xFFFFFFFFu;  // (saturated: refused by the host, a row must stay below 2^31)
    }
}

template <typename T>
__device__ __forceinline__ T prod_mul(T a, T b) {  // one rounding, never contracted with the later add
    if constexpr (sizeof(T) == 4) return __fmul_rn(a, b);
    else return __dmul_rn(a, b);
}

// one thread per entry d of the batch: its products, interleaved with those of the other members of its column group
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_prod_expand(uint64_t e0, uint64_t n_e, const uint32_t *__restrict__ row_of, const uint32_t *__restrict__ a_off,
              const uint32_t *__restrict__ a_col, const T *__restrict__ a_val, const uint32_t *__restrict__ b_off,
              const uint32_t *__restrict__ b_col, const T *__restrict__ b_val, uint64_t b_rows,
              const uint32_t *__restrict__ base /* exclusive scan of the batch's counts */, uint32_t r0, uint32_t *__restrict__ p_row,
              uint32_t *__restrict__ p_col, T *__restrict__ p_val) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_e; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t d = e0 + t;
        const uint32_t i = row_of[d], k = a_col[d];
        if (k >= b_rows) continue;
        const uint64_t lo = a_off[i], hi = a_off[i + 1];
        uint64_t gs = d, ge = d + 1;
        while (gs > lo && a_col[gs - 1] == k) --gs;
        while (ge < hi && a_col[ge] == k) ++ge;
        const uint64_t g = ge - gs, bs = b_off[k], len = b_off[k + 1] - bs;
        const uint64_t pos0 = (uint64_t)base[gs - e0] + (d - gs);
        const T a = a_val[d];
        for (uint64_t e = 0; e < len; ++e) {
            const uint64_t pos = pos0 + e * g;
            p_row[pos] = i - r0;
            p_col[pos] = b_col[bs + e];
            p_val[pos] = prod_mul(a, b_val[bs + e]);
        }
    }
}

// kept sums (sum != 0, sparsematrix.rs:203): entry-parallel.  flag[k] = entry k is kept (flag[nnz] = 0); its exclusive
// scan `rank` numbers the kept entries in (row, ascending column) order, so rank[off[r]] is row r's offset in the compacted
// arrays and rank[off[r + 1]] - rank[off[r]] its kept count; a kept entry of row r goes to the MIRRORED place inside the row's
// range -- descending columns, what ascending `set` calls leave in a SparseMatCRS (push prepends, sparsemat_crs.rs:85-87).
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_prod_flags(const T *__restrict__ val, uint64_t nnz, uint32_t *__restrict__ flag) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= nnz; k += (uint64_t)gridDim.x * blockDim.x)
        flag[k] = (k < nnz && val[k] != T(0)) ? 1u : 0u;
}

__global__ void __launch_bounds__(kBlock)
k_prod_row_kept(const uint32_t *__restrict__ off, const uint32_t *__restrict__ rank, uint64_t n_rows, uint32_t *__restrict__ kept) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * blockDim.x)
        kept[r] = rank[off[r + 1]] - rank[off[r]];
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_prod_compact(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const T *__restrict__ val, uint64_t n_rows,
               const uint32_t *__restrict__ rank, uint32_t r0, uint32_t *__restrict__ out_col, T *__restrict__ out_val,
               uint32_t *dims /* [0] largest kept row + 1, [1] largest kept column + 1 */) {
    __shared__ uint32_t s_off[kBlock + 1];
    uint32_t mr = 0, mc = 0;
    const uint64_t n_groups = (n_rows + kBlock - 1) / kBlock;
    for (uint64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint64_t g0 = g * kBlock, g1 = g0 + kBlock < n_rows ? g0 + kBlock : n_rows;
        const uint32_t nr = (uint32_t)(g1 - g0);
        __syncthreads();
        if (threadIdx.x <= nr) s_off[threadIdx.x] = off[g0 + threadIdx.x];
        if (threadIdx.x == 0 && nr == (uint32_t)kBlock) s_off[kBlock] = off[g1];
        __syncthreads();
        const uint32_t base = s_off[0], total = s_off[nr] - base;
        for (uint32_t p = threadIdx.x; p < total; p += kBlock) {
            const uint64_t e = (uint64_t)base + p;
            const T v = val[e];
            if (!(v != T(0))) continue;  // same test as the flags (NaN != 0: kept; -0 == 0: dropped)
            uint32_t lo = 0, hi = nr;  // last row with s_off[row] - base <= p
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_off[mid] - base <= p) lo = mid; else hi = mid;
            }
            const uint32_t c = col[e];
            const uint64_t w = (uint64_t)rank[s_off[lo]] + rank[s_off[lo + 1]] - 1u - rank[e];
            out_col[w] = c;
            out_val[w] = v;
            mc = max(mc, c + 1u);
            mr = max(mr, (uint32_t)(g0 + lo) + r0 + 1u);
        }
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        mr = max(mr, (uint32_t)__shfl_down(mr, o, kWave));
        mc = max(mc, (uint32_t)__shfl_down(mc, o, kWave));
    }
    if ((threadIdx.x & (kWave - 1)) == 0 && (mr | mc)) {
        atomicMax(&dims[0], mr);
        atomicMax(&dims[1], mc);
    }
}

struct DevBufs {  // device buffers freed on scope exit
    std::vector<void *> p;
    template <typename U> int alloc(U **out, size_t count) {
        SMH_HIP(hipMalloc((void **)out, (count ? count : 1) * sizeof(U)));
        p.push_back(*out);
        return SMH_OK;
    }
    void release(void *q) {
        for (auto &x : p)
            if (x == q) { (void)hipFree(x); x = nullptr; }
    }
    ~DevBufs() { for (void *x : p) (void)hipFree(x); }
};

template <typename T>
static int prod_t(const uint32_t *a_off, const uint32_t *a_col, const T *a_val, size_t a_rows, size_t a_nnz, uint32_t a_max_col,
                  const uint32_t *b_off, const uint32_t *b_col, const T *b_val, size_t b_rows, size_t *n_rows_out, size_t *n_cols_out,
                  size_t *nnz_out, uint32_t **off_out, uint32_t **col_out, T **val_out, hipStream_t s) {
    const int dtype = sizeof(T) == 8 ? SMH_F64 : SMH_F32;
    static const uint64_t budget = getenv("SMH_PROD_BATCH") ? (uint64_t)atoll(getenv("SMH_PROD_BATCH")) : (uint64_t)1 << 27;
    // SMH_PROD_TIMING=1: wall time of every stage on stderr (development aid)
    static const bool timing = getenv("SMH_PROD_TIMING") && atoi(getenv("SMH_PROD_TIMING")) != 0;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[prod] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    DevBufs bufs;
    *n_rows_out = *n_cols_out = *nnz_out = 0;
    *off_out = *col_out = nullptr;
    *val_out = nullptr;
    // left operand with rows stably sorted by column (sparsematrix.rs:194-195); a copy only when not sorted already
    int sorted = 1;
    SMH_TRY(crs_is_sorted(a_off, a_col, a_rows, &sorted, s));
    const uint32_t *as_col = a_col;
    const T *as_val = a_val;
    if (!sorted) {
        uint32_t *c = nullptr;
        T *v = nullptr;
        SMH_TRY(bufs.alloc(&c, a_nnz));
        SMH_TRY(bufs.alloc(&v, a_nnz));
        SMH_HIP(hipMemcpyAsync(c, a_col, a_nnz * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        SMH_HIP(hipMemcpyAsync(v, a_val, a_nnz * sizeof(T), hipMemcpyDeviceToDevice, s));
        SMH_TRY(sort_rows(dtype, a_off, c, v, a_rows, a_nnz, a_max_col, s));
        as_col = c; as_val = v;
    }
    uint32_t *row_of = nullptr, *cnt = nullptr, *kept = nullptr, *dims = nullptr;
    uint32_t *row_sum = nullptr;
    SMH_TRY(bufs.alloc(&row_of, a_nnz));
    SMH_TRY(bufs.alloc(&cnt, a_nnz));
    SMH_TRY(bufs.alloc(&row_sum, a_rows));
    SMH_TRY(bufs.alloc(&kept, a_rows + 1));
    SMH_TRY(bufs.alloc(&dims, 2));
    SMH_HIP(hipMemsetAsync(kept, 0, (a_rows + 1) * sizeof(uint32_t), s));
    SMH_HIP(hipMemsetAsync(dims, 0, 2 * sizeof(uint32_t), s));
    SMH_TRY(expand_rows(a_off, a_rows, row_of, s));
    if (a_nnz) hipLaunchKernelGGL(k_prod_counts, dim3(mat_grid(a_nnz)), dim3(kBlock), 0, s, as_col, (uint64_t)a_nnz, b_off, (uint64_t)b_rows, cnt);
    SMH_HIP(hipGetLastError());
    if (a_rows) hipLaunchKernelGGL(k_prod_row_sums, dim3(mat_grid(a_rows)), dim3(kBlock), 0, s, a_off, cnt, (uint64_t)a_rows, row_sum);
    SMH_HIP(hipGetLastError());
    // products per row on the host (4 bytes a row, no zero-filled staging); the entry offsets of the few batch borders are
    // fetched one by one below
    std::unique_ptr<uint32_t[]> h_sum(new (std::nothrow) uint32_t[a_rows ? a_rows : 1]);
    if (!h_sum) return fail(SMH_ERR_OOM, "host allocation failed");
    if (a_rows) SMH_HIP(hipMemcpyAsync(h_sum.get(), row_sum, a_rows * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    SMH_HIP(hipStreamSynchronize(s));
    bufs.release(row_sum);
    lap("sorted check, counts, row sums");
    // batches of whole rows with at most `budget` products (a longer single row goes alone)
    struct Batch { size_t r0, r1; uint64_t products; uint32_t e0, e1; };
    std::vector<Batch> batches;
    uint64_t max_products = 0, max_entries = 0;
    for (size_t r = 0; r < a_rows;) {
        size_t r1 = r;
        uint64_t p = 0;
        while (r1 < a_rows && (r1 == r || p + h_sum[r1] <= budget)) p += h_sum[r1++];
        if (p >= 0x7FFFFFFFull) return fail(SMH_ERR_CAPACITY, "prod: row %zu alone generates %llu products", r, (unsigned long long)p);
        if (p) {
            Batch bt{r, r1, p, 0, 0};
            SMH_HIP(hipMemcpy(&bt.e0, a_off + r, sizeof(uint32_t), hipMemcpyDeviceToHost));
            SMH_HIP(hipMemcpy(&bt.e1, a_off + r1, sizeof(uint32_t), hipMemcpyDeviceToHost));
            batches.push_back(bt);
            if (p > max_products) max_products = p;
            if ((uint64_t)(bt.e1 - bt.e0) > max_entries) max_entries = bt.e1 - bt.e0;
        }
        r = r1;
    }
    uint32_t *base = nullptr, *p_row = nullptr, *p_col = nullptr;
    T *p_val = nullptr;
    SMH_TRY(bufs.alloc(&base, max_entries + 1));
    SMH_TRY(bufs.alloc(&p_row, max_products));
    SMH_TRY(bufs.alloc(&p_col, max_products));
    SMH_TRY(bufs.alloc(&p_val, max_products));
    struct Piece { uint32_t *col; T *val; uint64_t n; };
    std::vector<Piece> pieces;
    uint64_t total = 0;
    for (const Batch &bt : batches) {
        const uint64_t e0 = bt.e0, n_e = bt.e1 - bt.e0;
        SMH_HIP(hipMemcpyAsync(base, cnt + e0, n_e * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        SMH_HIP(hipMemsetAsync(base + n_e, 0, sizeof(uint32_t), s));
        uint64_t products = 0;
        SMH_TRY(device_exclusive_scan_u32(base, n_e + 1, s, &products));
        if (products != bt.products) return fail(SMH_ERR_INVALID, "prod: product count changed between passes");
        hipLaunchKernelGGL((k_prod_expand<T>), dim3(mat_grid(n_e)), dim3(kBlock), 0, s, e0, n_e, row_of, a_off, as_col, as_val, b_off, b_col,
                           b_val, (uint64_t)b_rows, base, (uint32_t)bt.r0, p_row, p_col, p_val);
        SMH_HIP(hipGetLastError());
        lap("expand");
        size_t nr = 0, nc = 0, nnz_b = 0;
        uint32_t *off_b = nullptr, *col_b = nullptr;
        void *val_b = nullptr;
        SMH_TRY(assemble_triplets(dtype, products, p_row, p_col, p_val, nullptr, false, false, false, &nr, &nc, &nnz_b, &off_b, &col_b, &val_b, s));
        bufs.p.push_back(off_b); bufs.p.push_back(col_b); bufs.p.push_back(val_b);
        lap("fold (assembly)");
        SMH_TRY(sort_rows(dtype, off_b, col_b, val_b, nr, nnz_b, (uint32_t)(nc ? nc - 1 : 0), s));
        lap("sort rows by column");
        uint32_t *rank = nullptr;
        SMH_TRY(bufs.alloc(&rank, nnz_b + 1));
        hipLaunchKernelGGL((k_prod_flags<T>), dim3(mat_grid(nnz_b + 1)), dim3(kBlock), 0, s, (const T *)val_b, (uint64_t)nnz_b, rank);
        SMH_HIP(hipGetLastError());
        uint64_t n_kept = 0;
        SMH_TRY(device_exclusive_scan_u32(rank, nnz_b + 1, s, &n_kept));
        hipLaunchKernelGGL(k_prod_row_kept, dim3(mat_grid(nr)), dim3(kBlock), 0, s, off_b, rank, (uint64_t)nr, kept + bt.r0);
        SMH_HIP(hipGetLastError());
        Piece pc{nullptr, nullptr, n_kept};
        if (n_kept) {
            SMH_TRY(bufs.alloc(&pc.col, n_kept));
            SMH_TRY(bufs.alloc(&pc.val, n_kept));
            hipLaunchKernelGGL((k_prod_compact<T>), dim3(mat_grid(nr)), dim3(kBlock), 0, s, off_b, col_b, (const T *)val_b, (uint64_t)nr, rank,
                               (uint32_t)bt.r0, pc.col, pc.val, dims);
            SMH_HIP(hipGetLastError());
            pieces.push_back(pc);
            total += n_kept;
        }
        SMH_HIP(hipStreamSynchronize(s));
        bufs.release(off_b); bufs.release(col_b); bufs.release(val_b); bufs.release(rank);
        lap("drop zeros, reverse");
    }
    if (total >= 0xFFFFFFFFull) return fail(SMH_ERR_CAPACITY, "Maximum number of %u entries reached", 0xFFFFFFFFu);
    uint32_t h_dims[2] = {0, 0};
    SMH_HIP(hipMemcpyAsync(h_dims, dims, sizeof h_dims, hipMemcpyDeviceToHost, s));
    SMH_HIP(hipStreamSynchronize(s));
    // result arrays (owned by the caller, padded like smh_crs_create's)
    const bool rowless = total <= 1;  // no set call at all, or the single push that leaves n_rows == 0
    const size_t n_rows = rowless ? 0 : h_dims[0], nnz = rowless ? 0 : (size_t)total;
    uint32_t *off = nullptr, *col = nullptr;
    T *val = nullptr;
    auto finish = [&]() -> int {
        SMH_HIP(hipMalloc((void **)&off, (n_rows + 1) * sizeof(uint32_t)));
        SMH_HIP(hipMalloc((void **)&col, (nnz + 4) * sizeof(uint32_t)));
        SMH_HIP(hipMalloc((void **)&val, (nnz + 4) * sizeof(T)));
        SMH_HIP(hipMemsetAsync(col + nnz, 0, 4 * sizeof(uint32_t), s));
        SMH_HIP(hipMemsetAsync(val + nnz, 0, 4 * sizeof(T), s));
        if (rowless) {
            SMH_HIP(hipMemsetAsync(off, 0, sizeof(uint32_t), s));
        } else {
            uint64_t check = 0;
            SMH_TRY(device_exclusive_scan_u32(kept, a_rows + 1, s, &check));  // kept[a_rows] == 0: offsets of all rows
            if (check != total) return fail(SMH_ERR_INVALID, "prod: kept-entry count changed between passes");
            SMH_HIP(hipMemcpyAsync(off, kept, (n_rows + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
            uint64_t at = 0;
            for (const Piece &pc : pieces) {
                SMH_HIP(hipMemcpyAsync(col + at, pc.col, pc.n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
                SMH_HIP(hipMemcpyAsync(val + at, pc.val, pc.n * sizeof(T), hipMemcpyDeviceToDevice, s));
                at += pc.n;
            }
        }
        SMH_HIP(hipStreamSynchronize(s));
        return SMH_OK;
    };
    const int rc = finish();
    if (rc != SMH_OK) {
        (void)hipFree(off); (void)hipFree(col); (void)hipFree(val);
        return rc;
    }
    *n_rows_out = n_rows; *n_cols_out = total ? h_dims[1] : 0; *nnz_out = nnz;
    *off_out = off; *col_out = col; *val_out = val;
    return SMH_OK;
}

// Device arrays; the dimension rule of sparsematrix.rs:188-190 is the caller's (capi.hip).
int prod_crs(int dtype, const uint32_t *a_off, const uint32_t *a_col, const void *a_val, size_t a_rows, size_t a_nnz, uint32_t a_max_col,
             const uint32_t *b_off, const uint32_t *b_col, const void *b_val, size_t b_rows, size_t *n_rows_out, size_t *n_cols_out,
             size_t *nnz_out, uint32_t **off_out, uint32_t **col_out, void **val_out, hipStream_t s) {
    if (dtype == SMH_F64)
        return prod_t<double>(a_off, a_col, (const double *)a_val, a_rows, a_nnz, a_max_col, b_off, b_col, (const double *)b_val, b_rows,
                              n_rows_out, n_cols_out, nnz_out, off_out, col_out, (double **)val_out, s);
    return prod_t<float>(a_off, a_col, (const float *)a_val, a_rows, a_nnz, a_max_col, b_off, b_col, (const float *)b_val, b_rows,
                         n_rows_out, n_cols_out, nnz_out, off_out, col_out, (float **)val_out, s);
}

}  // namespace smh
