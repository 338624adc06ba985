// spmv_colblock.hip -- K2c: column-blocked CSR for matrices whose columns have no locality (gfx950).
//
// Why: with columns spread over all of x every gather is an L1 and L2 miss; rocprofv3 on the uniform-column
// C2 matrix (profiles/r01_pmc_uniform.json) shows the per-CU miss path, not HBM, as the limit (~59 G
// gathers/s whatever the kernel).  tools/experiment_gather.py: the same kernels reach 180-190 G gathers/s
// when x fits the XCD's 4 MiB L2 (<= 2 MiB), 102 G/s at 8 MiB, 54 G/s at 67 MiB.  K2c therefore keeps the
// gathered part of x L2-resident: the device copy of the matrix is re-laid out, once, as B column blocks
//     A = [A_0 | A_1 | ... | A_{B-1}],   block width 2^19 columns (2 MiB of f32 x, 4 MiB of f64 x),
// each A_b a CSR over ALL rows (entries of a row keep their storage order inside a block), and
//     y = A_0 x ; y += A_1 x ; ... ; y += A_{B-1} x
// runs as B launches of the dense CSR-stream kernel K1s (its tiles widened to 2048 rows, since a row has
// only mean/B entries per block), during each of which every XCD gathers from one block of x.
// Measured: C2-uniform 5.4 -> 2.0-2.2 ms, C3 (f64 power law) 5.9 -> 3.25 ms.
// Extra traffic: B offset arrays and B read-modify-write sweeps of y (streamed, coalesced).
// The sum of a row is formed block by block, i.e. NOT in storage order: tolerance parity (like K1r/K2),
// deterministic and bitwise reproducible.  The split itself is integer work, checked bit-exact in the tests.
#include <vector>

#include "internal.hpp"

namespace smh {

constexpr int kScanChunk = 4096;  // elements per block of the scan passes (256 threads x 16)

// cnt[b*(n_rows+1) + r] = number of entries of row r whose column lies in block b
__global__ void __launch_bounds__(kBlock)
k_cb_count(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, uint64_t n_rows, uint32_t shift,
           uint32_t *__restrict__ cnt) {
    const uint64_t stride = n_rows + 1;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k1 = off[r + 1];
        for (uint64_t k = off[r]; k < k1; ++k) cnt[(uint64_t)(col[k] >> shift) * stride + r] += 1u;  // row r is this thread's
    }
}

// pass 1 of the exclusive scan: sums[c] = sum of chunk c
__global__ void __launch_bounds__(kBlock)
k_scan_sums(const uint32_t *__restrict__ in, uint64_t n, uint32_t *__restrict__ sums) {
    __shared__ uint32_t s_w[kBlock / kWave];
    const uint64_t base = (uint64_t)blockIdx.x * kScanChunk;
    uint32_t acc = 0;
    for (int i = 0; i < kScanChunk / kBlock; ++i) {
        const uint64_t k = base + (uint64_t)i * kBlock + threadIdx.x;
        if (k < n) acc += in[k];
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) acc += (uint32_t)__shfl_down((int)acc, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) s_w[threadIdx.x / kWave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// pass 2: in-place exclusive scan of every chunk, plus the chunk's base (exclusive scan of the sums)
__global__ void __launch_bounds__(kBlock)
k_scan_apply(uint32_t *__restrict__ data, uint64_t n, const uint32_t *__restrict__ bases) {
    __shared__ uint32_t s_w[kBlock / kWave];
    constexpr int PER = kScanChunk / kBlock;  // 16 consecutive elements per thread
    const uint64_t k0 = (uint64_t)blockIdx.x * kScanChunk + (uint64_t)threadIdx.x * PER;
    uint32_t v[PER];
    uint32_t tot = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        v[i] = k0 + i < n ? data[k0 + i] : 0u;
        tot += v[i];
    }
    // exclusive scan of the per-thread totals across the block
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t incl = tot;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const uint32_t p = (uint32_t)__shfl_up((int)incl, o, kWave);
        if ((int)lane >= o) incl += p;
    }
    if (lane == kWave - 1) s_w[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0;
    for (uint32_t w = 0; w < wave; ++w) wave_base += s_w[w];
    uint32_t run = bases[blockIdx.x] + wave_base + incl - tot;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        if (k0 + i < n) data[k0 + i] = run;
        run += v[i];
    }
}

// scatter: entry k of row r goes to cur[block(col[k])][r]++ (storage order kept inside a (row, block) pair)
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_cb_scatter(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const T *__restrict__ val, uint64_t n_rows,
             uint32_t shift, uint32_t *__restrict__ cur, uint32_t *__restrict__ col2, T *__restrict__ val2) {
    const uint64_t stride = n_rows + 1;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k1 = off[r + 1];
        for (uint64_t k = off[r]; k < k1; ++k) {
            const uint32_t c = col[k];
            uint32_t *slot = cur + (uint64_t)(c >> shift) * stride + r;
            const uint32_t pos = *slot;
            *slot = pos + 1u;
            col2[pos] = c;
            val2[pos] = val[k];
        }
    }
}

static unsigned rows_grid(uint64_t n_rows) {
    uint64_t b = (n_rows + kBlock - 1) / kBlock;
    if (b > 8192) b = 8192;
    if (b == 0) b = 1;
    return (unsigned)b;
}

// in-place exclusive scan of `n` u32 on the device (chunk sums folded on the host: n/4096 values)
int device_exclusive_scan_u32(uint32_t *data, uint64_t n, hipStream_t s, uint64_t *total_out) {
    if (total_out) *total_out = 0;
    if (n == 0) return SMH_OK;
    const uint64_t chunks = (n + kScanChunk - 1) / kScanChunk;
    uint32_t *d_sums = nullptr;
    SMH_HIP(hipMalloc((void **)&d_sums, chunks * sizeof(uint32_t)));
    std::vector<uint32_t> h(chunks);
    int rc = SMH_OK;
    auto body = [&]() -> int {
        hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)chunks), dim3(kBlock), 0, s, data, n, d_sums);
        SMH_HIP(hipGetLastError());
        SMH_HIP(hipMemcpyAsync(h.data(), d_sums, chunks * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        SMH_HIP(hipStreamSynchronize(s));
        uint64_t run = 0;
        for (uint64_t c = 0; c < chunks; ++c) {
            const uint32_t v = h[c];
            h[c] = (uint32_t)run;
            run += v;
        }
        if (run >= 0xFFFFFFFFull) return fail(SMH_ERR_CAPACITY, "Maximum number of %u entries reached", 0xFFFFFFFFu);
        if (total_out) *total_out = run;
        SMH_HIP(hipMemcpyAsync(d_sums, h.data(), chunks * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)chunks), dim3(kBlock), 0, s, data, n, d_sums);
        SMH_HIP(hipGetLastError());
        SMH_HIP(hipStreamSynchronize(s));
        return SMH_OK;
    };
    rc = body();
    (void)hipFree(d_sums);
    return rc;
}

// Build the column-blocked copy.  Outputs (device, owned by the caller): off2 [n_blocks*(n_rows+1)] absolute
// entry positions, col2 / val2 [nnz + 4].
int build_colblock(int dtype, const uint32_t *off, const uint32_t *col, const void *val, size_t n_rows, size_t nnz,
                   uint32_t shift, size_t n_blocks, uint32_t **off2_out, uint32_t **col2_out, void **val2_out,
                   hipStream_t s) {
    const uint64_t total = (uint64_t)n_blocks * (n_rows + 1);
    const size_t vs = dtype_size(dtype);
    uint32_t *off2 = nullptr, *cur = nullptr, *col2 = nullptr;
    void *val2 = nullptr;
    auto body = [&]() -> int {
        SMH_HIP(hipMalloc((void **)&off2, total * sizeof(uint32_t)));
        SMH_HIP(hipMemsetAsync(off2, 0, total * sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_cb_count, dim3(rows_grid(n_rows)), dim3(kBlock), 0, s, off, col, (uint64_t)n_rows, shift, off2);
        SMH_HIP(hipGetLastError());
        SMH_TRY(device_exclusive_scan_u32(off2, total, s, nullptr));
        SMH_HIP(hipMalloc((void **)&cur, total * sizeof(uint32_t)));
        SMH_HIP(hipMemcpyAsync(cur, off2, total * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        SMH_HIP(hipMalloc((void **)&col2, (nnz + 4) * sizeof(uint32_t)));
        SMH_HIP(hipMalloc(&val2, (nnz + 4) * vs));
        SMH_HIP(hipMemsetAsync(col2 + nnz, 0, 4 * sizeof(uint32_t), s));
        SMH_HIP(hipMemsetAsync((char *)val2 + nnz * vs, 0, 4 * vs, s));
        if (dtype == SMH_F64)
            hipLaunchKernelGGL(k_cb_scatter<double>, dim3(rows_grid(n_rows)), dim3(kBlock), 0, s, off, col, (const double *)val,
                               (uint64_t)n_rows, shift, cur, col2, (double *)val2);
        else
            hipLaunchKernelGGL(k_cb_scatter<float>, dim3(rows_grid(n_rows)), dim3(kBlock), 0, s, off, col, (const float *)val,
                               (uint64_t)n_rows, shift, cur, col2, (float *)val2);
        SMH_HIP(hipGetLastError());
        SMH_HIP(hipStreamSynchronize(s));
        return SMH_OK;
    };
    const int rc = body();
    (void)hipFree(cur);
    if (rc != SMH_OK) {
        (void)hipFree(off2); (void)hipFree(col2); (void)hipFree(val2);
        return rc;
    }
    *off2_out = off2; *col2_out = col2; *val2_out = val2;
    return SMH_OK;
}

}  // namespace smh
