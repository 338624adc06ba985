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
//   1. key = row << 32 | column, payload = stream position; STABLE radix sort (rocPRIM): the operations of an
//      entry become one contiguous run, still in stream order;
//   2. run heads flagged, exclusive scan -> dense entry ids; one thread per run folds it sequentially and
//      records (row << 32 | first stream position, column, value);
//   3. second stable sort on row << 32 | first position: rows ascending, first-appearance order inside a row;
//   4. gather columns / values, row offsets from the sorted row ids.
// Also here: Sortable::sort_row (sparsemat_crs.rs:163-172; slice::sort_by is stable) for all rows at once --
// one stable sort on row << 32 | column.
#include <rocprim/device/device_radix_sort.hpp>

#include "internal.hpp"

namespace smh {

int device_exclusive_scan_u32(uint32_t *data, uint64_t n, hipStream_t s, uint64_t *total_out);  // spmv_colblock.hip

static unsigned grid_for(uint64_t n) {
    uint64_t b = (n + kBlock - 1) / kBlock;
    if (b > 16384) b = 16384;
    return (unsigned)(b ? b : 1);
}

// keys of the first sort + the matrix dimensions (integer max: exact, order independent)
__global__ void __launch_bounds__(kBlock)
k_asm_keys(const uint32_t *__restrict__ rows, const uint32_t *__restrict__ cols, uint64_t n, uint64_t *__restrict__ key,
           uint32_t *__restrict__ idx, uint32_t *__restrict__ dims /* [0] max row, [1] max column */) {
    uint32_t mr = 0, mc = 0;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = rows[k], c = cols[k];
        key[k] = ((uint64_t)r << 32) | c;
        idx[k] = (uint32_t)k;
        mr = max(mr, r);
        mc = max(mc, c);
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        mr = max(mr, (uint32_t)__shfl_down(mr, o, kWave));
        mc = max(mc, (uint32_t)__shfl_down(mc, o, kWave));
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        atomicMax(&dims[0], mr);
        atomicMax(&dims[1], mc);
    }
}

// operations brought into sorted order (coalesced writes; the fold then reads runs sequentially) + run heads
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_asm_heads(const uint64_t *__restrict__ key, const uint32_t *__restrict__ idx, const T *__restrict__ vals,
            const uint8_t *__restrict__ ops, uint64_t n, T *__restrict__ vals_s, uint8_t *__restrict__ ops_s,
            uint32_t *__restrict__ head) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t i = idx[k];
        vals_s[k] = vals[i];
        if (ops) ops_s[k] = ops[i];
        head[k] = (k == 0 || key[k] != key[k - 1]) ? 1u : 0u;
    }
}

// one thread per run: sequential fold in stream order (sparsematrix.rs:226-233 on the entry get_mut returned,
// which push created as T::zero(): sparsemat_indexlist.rs:160-162)
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_asm_fold(const uint64_t *__restrict__ key, const uint32_t *__restrict__ idx, const T *__restrict__ vals_s,
           const uint8_t *__restrict__ ops_s, const uint32_t *__restrict__ entry_id /* exclusive scan of heads */,
           uint64_t n, uint64_t *__restrict__ ukey, uint32_t *__restrict__ uidx, uint32_t *__restrict__ ucol,
           T *__restrict__ uval) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t me = key[k];
        if (k > 0 && key[k - 1] == me) continue;  // not a run head
        T acc = T(0);
        uint64_t j = k;
        do {
            const T v = vals_s[j];
            if (ops_s && ops_s[j]) acc = v;
            else if constexpr (sizeof(T) == 4) acc = __fadd_rn(acc, v);
            else acc = __dadd_rn(acc, v);
            ++j;
        } while (j < n && key[j] == me);
        const uint32_t u = entry_id[k];
        ukey[u] = (me & 0xFFFFFFFF00000000ull) | idx[k];  // stable sort: idx[k] is the run's first stream position
        uidx[u] = u;
        ucol[u] = (uint32_t)me;
        uval[u] = acc;
    }
}

// entries in final order + row offsets from the sorted row ids
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_asm_emit(const uint64_t *__restrict__ ukey_s, const uint32_t *__restrict__ uidx_s, const uint32_t *__restrict__ ucol,
           const T *__restrict__ uval, uint64_t n_entries, uint64_t n_rows, uint32_t *__restrict__ off,
           uint32_t *__restrict__ col, T *__restrict__ val) {
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_entries; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t u = uidx_s[p];
        col[p] = ucol[u];
        val[p] = uval[u];
        const uint64_t r = ukey_s[p] >> 32;
        const uint64_t prev = p ? (ukey_s[p - 1] >> 32) : ~uint64_t(0);  // rows (prev, r] start at p
        if (p == 0 || prev != r)
            for (uint64_t q = p ? prev + 1 : 0; q <= r; ++q) off[q] = (uint32_t)p;
        if (p + 1 == n_entries)
            for (uint64_t q = r + 1; q <= n_rows; ++q) off[q] = (uint32_t)n_entries;
    }
}

static int sort_pairs_u64_u32(const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, uint64_t n,
                              unsigned end_bit, hipStream_t s) {
    size_t bytes = 0;
    SMH_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, (size_t)n, 0u, end_bit, s));
    void *tmp = nullptr;
    SMH_HIP(hipMalloc(&tmp, bytes ? bytes : 16));
    const hipError_t e = rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, (size_t)n, 0u, end_bit, s);
    const hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    SMH_HIP(e);
    SMH_HIP(e2);
    return SMH_OK;
}

static unsigned bits_for(uint32_t v) {  // bits needed to hold v
    unsigned b = 1;
    while (b < 32 && (v >> b)) ++b;
    return b;
}

// device buffers freed on scope exit
struct Scratch {
    void *p[16] = {};
    int n = 0;
    template <typename U> int alloc(U **out, size_t count) {
        SMH_HIP(hipMalloc((void **)out, (count ? count : 1) * sizeof(U)));
        p[n++] = *out;
        return SMH_OK;
    }
    ~Scratch() { for (int i = 0; i < n; ++i) (void)hipFree(p[i]); }
};

template <typename T>
static int assemble_t(uint64_t n, const uint32_t *rows, const uint32_t *cols, const T *vals, const uint8_t *ops,
                      size_t *n_rows_out, size_t *n_cols_out, size_t *nnz_out, uint32_t **off_out, uint32_t **col_out,
                      T **val_out, hipStream_t s) {
    Scratch tmp;
    uint64_t *key = nullptr, *key_s = nullptr, *ukey = nullptr, *ukey_s = nullptr;
    uint32_t *idx = nullptr, *idx_s = nullptr, *head = nullptr, *dims = nullptr, *uidx = nullptr, *uidx_s = nullptr, *ucol = nullptr;
    T *vals_s = nullptr, *uval = nullptr;
    uint8_t *ops_s = nullptr;
    SMH_TRY(tmp.alloc(&dims, 2));
    SMH_HIP(hipMemsetAsync(dims, 0, 2 * sizeof(uint32_t), s));
    SMH_TRY(tmp.alloc(&key, n));
    SMH_TRY(tmp.alloc(&key_s, n));
    SMH_TRY(tmp.alloc(&idx, n));
    SMH_TRY(tmp.alloc(&idx_s, n));
    hipLaunchKernelGGL(k_asm_keys, dim3(grid_for(n)), dim3(kBlock), 0, s, rows, cols, n, key, idx, dims);
    SMH_HIP(hipGetLastError());
    uint32_t h_dims[2];
    SMH_HIP(hipMemcpyAsync(h_dims, dims, sizeof h_dims, hipMemcpyDeviceToHost, s));
    SMH_HIP(hipStreamSynchronize(s));
    const uint64_t n_rows = (uint64_t)h_dims[0] + 1, n_cols = (uint64_t)h_dims[1] + 1;
    const unsigned row_bits = bits_for(h_dims[0]);
    SMH_TRY(sort_pairs_u64_u32(key, key_s, idx, idx_s, n, 32 + row_bits, s));
    // runs
    SMH_TRY(tmp.alloc(&head, n));
    SMH_TRY(tmp.alloc(&vals_s, n));
    if (ops) SMH_TRY(tmp.alloc(&ops_s, n));
    hipLaunchKernelGGL((k_asm_heads<T>), dim3(grid_for(n)), dim3(kBlock), 0, s, key_s, idx_s, vals, ops, n, vals_s, ops_s, head);
    SMH_HIP(hipGetLastError());
    uint64_t n_entries = 0;
    SMH_TRY(device_exclusive_scan_u32(head, n, s, &n_entries));
    // (key, idx) are free again: reuse them for the second sort's inputs
    ukey = key;
    uidx = idx;
    SMH_TRY(tmp.alloc(&ucol, n_entries));
    SMH_TRY(tmp.alloc(&uval, n_entries));
    hipLaunchKernelGGL((k_asm_fold<T>), dim3(grid_for(n)), dim3(kBlock), 0, s, key_s, idx_s, vals_s, ops_s, head, n, ukey, uidx,
                       ucol, uval);
    SMH_HIP(hipGetLastError());
    SMH_HIP(hipStreamSynchronize(s));
    ukey_s = key_s;  // the first sort's outputs are dead after the fold
    uidx_s = idx_s;
    SMH_TRY(sort_pairs_u64_u32(ukey, ukey_s, uidx, uidx_s, n_entries, 32 + row_bits, s));
    // result arrays (owned by the caller; padded like smh_crs_create's)
    uint32_t *off = nullptr, *col = nullptr;
    T *val = nullptr;
    auto emit = [&]() -> int {
        SMH_HIP(hipMalloc((void **)&off, (n_rows + 1) * sizeof(uint32_t)));
        SMH_HIP(hipMalloc((void **)&col, (n_entries + 4) * sizeof(uint32_t)));
        SMH_HIP(hipMalloc((void **)&val, (n_entries + 4) * sizeof(T)));
        SMH_HIP(hipMemsetAsync(col + n_entries, 0, 4 * sizeof(uint32_t), s));
        SMH_HIP(hipMemsetAsync(val + n_entries, 0, 4 * sizeof(T), s));
        hipLaunchKernelGGL((k_asm_emit<T>), dim3(grid_for(n_entries)), dim3(kBlock), 0, s, ukey_s, uidx_s, ucol, uval, n_entries,
                           n_rows, off, col, val);
        SMH_HIP(hipGetLastError());
        SMH_HIP(hipStreamSynchronize(s));
        return SMH_OK;
    };
    const int rc = emit();
    if (rc != SMH_OK) {
        (void)hipFree(off); (void)hipFree(col); (void)hipFree(val);
        return rc;
    }
    *n_rows_out = (size_t)n_rows; *n_cols_out = (size_t)n_cols; *nnz_out = (size_t)n_entries;
    *off_out = off; *col_out = col; *val_out = val;
    return SMH_OK;
}

// rows/cols/vals/ops: DEVICE arrays of n operations (ops may be null: all add_to).  n >= 1.
int assemble_triplets(int dtype, size_t n, const uint32_t *rows, const uint32_t *cols, const void *vals, const uint8_t *ops,
                      size_t *n_rows_out, size_t *n_cols_out, size_t *nnz_out, uint32_t **off_out, uint32_t **col_out,
                      void **val_out, hipStream_t s) {
    if (dtype == SMH_F64)
        return assemble_t<double>(n, rows, cols, (const double *)vals, ops, n_rows_out, n_cols_out, nnz_out, off_out, col_out,
                                  (double **)val_out, s);
    return assemble_t<float>(n, rows, cols, (const float *)vals, ops, n_rows_out, n_cols_out, nnz_out, off_out, col_out,
                             (float **)val_out, s);
}

// ---- sort_row for every row --------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_sortrows_keys(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, uint64_t n_rows, uint64_t nnz,
                uint64_t *__restrict__ key, uint32_t *__restrict__ idx) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (uint64_t)gridDim.x * blockDim.x) {
        // row of entry k: the last r with off[r] <= k
        uint64_t lo = 0, hi = n_rows;
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if ((uint64_t)off[mid] <= k) lo = mid; else hi = mid;
        }
        key[k] = (lo << 32) | col[k];
        idx[k] = (uint32_t)k;
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_sortrows_gather(const uint32_t *__restrict__ idx_s, const T *__restrict__ val, uint64_t nnz, T *__restrict__ val_s) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (uint64_t)gridDim.x * blockDim.x)
        val_s[k] = val[idx_s[k]];
}

__global__ void __launch_bounds__(kBlock)
k_sortrows_cols(const uint64_t *__restrict__ key_s, uint64_t nnz, uint32_t *__restrict__ col) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (uint64_t)gridDim.x * blockDim.x)
        col[k] = (uint32_t)key_s[k];
}

int sort_rows(int dtype, const uint32_t *off, uint32_t *col, void *val, size_t n_rows, size_t nnz, hipStream_t s) {
    if (nnz == 0 || n_rows == 0) return SMH_OK;
    Scratch tmp;
    uint64_t *key = nullptr, *key_s = nullptr;
    uint32_t *idx = nullptr, *idx_s = nullptr;
    void *val_s = nullptr;
    const size_t vs = dtype_size(dtype);
    SMH_TRY(tmp.alloc(&key, nnz));
    SMH_TRY(tmp.alloc(&key_s, nnz));
    SMH_TRY(tmp.alloc(&idx, nnz));
    SMH_TRY(tmp.alloc(&idx_s, nnz));
    SMH_TRY(tmp.alloc((char **)&val_s, nnz * vs));
    hipLaunchKernelGGL(k_sortrows_keys, dim3(grid_for(nnz)), dim3(kBlock), 0, s, off, col, (uint64_t)n_rows, (uint64_t)nnz, key, idx);
    SMH_HIP(hipGetLastError());
    SMH_TRY(sort_pairs_u64_u32(key, key_s, idx, idx_s, nnz, 32 + bits_for((uint32_t)(n_rows - 1)), s));
    if (dtype == SMH_F64)
        hipLaunchKernelGGL((k_sortrows_gather<double>), dim3(grid_for(nnz)), dim3(kBlock), 0, s, idx_s, (const double *)val,
                           (uint64_t)nnz, (double *)val_s);
    else
        hipLaunchKernelGGL((k_sortrows_gather<float>), dim3(grid_for(nnz)), dim3(kBlock), 0, s, idx_s, (const float *)val,
                           (uint64_t)nnz, (float *)val_s);
    SMH_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_sortrows_cols, dim3(grid_for(nnz)), dim3(kBlock), 0, s, key_s, (uint64_t)nnz, col);
    SMH_HIP(hipGetLastError());
    SMH_HIP(hipMemcpyAsync(val, val_s, nnz * vs, hipMemcpyDeviceToDevice, s));
    SMH_HIP(hipStreamSynchronize(s));
    return SMH_OK;
}

}  // namespace smh
