// pcg.hip -- Jacobi-preconditioned conjugate gradient, gfx950.
//
// An EXTENSION: the reference's only solver is the unpreconditioned ConjugateGradient (linearsolver.rs:12-61).  This is
// that recurrence with z = r / diag(A), diag_i = get(i, i) (first match in storage order: sparsemat_crs.rs:54-67,
// 136-142) -- same guards (:30-36), same stop rule (sqrt(f64(r.r)) < tol after the update of r, before beta, :52-54),
// same arithmetic conventions (one rounding per operation; multiply and add never contracted):
//   r = b - A x;  p = r / d;  rz = r.(r/d)
//   loop: Ap = A p;  alpha = rz / (p.Ap);  x += p*alpha;  r -= Ap*alpha;  rr = r.r;  stop test;
//         rz' = r.(r/d);  beta = rz' / rz;  p = p*beta + r/d
// Kernels: the SpMV is the matrix's own (K1r/K1s/...); the tail is fused so that z is never stored: one sweep updates
// x and r and leaves the block partials of r.r and r.(r/d) (6 vector streams + d), one sweep rebuilds p (r, d, p).  The
// two scalars of an iteration pass through the host (p.Ap, then r.r / r.z): 2 synchronisations per iteration.
// Reductions: fixed grid, per-thread strided sums, wave butterfly, LDS across waves, one block folds the partials in
// index order -- deterministic.
#include "internal.hpp"

#include <cmath>

using namespace smh;

namespace {

constexpr int kPcgBlocks = 512;  // 2 blocks per CU (the CG tail's measured optimum)

template <typename T> __device__ __forceinline__ T p_mul(T a, T b) { if constexpr (sizeof(T) == 4) return __fmul_rn(a, b); else return __dmul_rn(a, b); }
template <typename T> __device__ __forceinline__ T p_add(T a, T b) { if constexpr (sizeof(T) == 4) return __fadd_rn(a, b); else return __dadd_rn(a, b); }
template <typename T> __device__ __forceinline__ T p_div(T a, T b) { if constexpr (sizeof(T) == 4) return __fdiv_rn(a, b); else return __ddiv_rn(a, b); }

// d[i] = get(i, i); *bad = the smallest row whose diagonal entry is zero or absent (stays ~0 when there is none)
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_pcg_diag(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const T *__restrict__ val, uint64_t n, T *__restrict__ d,
           unsigned long long *bad) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        T v = T(0);
        for (uint64_t q = off[i]; q < off[i + 1]; ++q)
            if (col[q] == i) { v = val[q]; break; }
        d[i] = v;
        if (v == T(0)) atomicMin(bad, (unsigned long long)i);
    }
}

template <typename T>
__device__ __forceinline__ void block_sums(T a, T b, T *pa, T *pb) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        a = p_add(a, (T)__shfl_down(a, o, kWave));
        b = p_add(b, (T)__shfl_down(b, o, kWave));
    }
    __shared__ T sa[kBlock / kWave], sb[kBlock / kWave];
    if ((threadIdx.x & (kWave - 1)) == 0) { sa[threadIdx.x / kWave] = a; sb[threadIdx.x / kWave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        T ta = sa[0], tb = sb[0];
        for (int w = 1; w < kBlock / kWave; ++w) { ta = p_add(ta, sa[w]); tb = p_add(tb, sb[w]); }
        pa[blockIdx.x] = ta;
        pb[blockIdx.x] = tb;
    }
}

// UPDATE: x += p*alpha, r -= ap*alpha first; always: partials of r.r and r.(r/d)
template <typename T, bool UPDATE>
__global__ void __launch_bounds__(kBlock)
k_pcg_update(T *__restrict__ x, T *__restrict__ r, const T *__restrict__ p, const T *__restrict__ ap, const T *__restrict__ d,
             uint64_t n, T alpha, T *__restrict__ part_rr, T *__restrict__ part_rz) {
    // 16 bytes per lane and array (the buffers are hipMalloc'ed: aligned); the last n % V elements one by one
    constexpr int V = 16 / sizeof(T);
    typedef T VT __attribute__((ext_vector_type(V)));
    T s_rr = T(0), s_rz = T(0);
    const uint64_t nv = n / V, tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t q = tid; q < nv; q += nthreads) {
        VT rv = reinterpret_cast<const VT *>(r)[q];
        const VT dv = reinterpret_cast<const VT *>(d)[q];
        if (UPDATE) {
            VT xv = reinterpret_cast<const VT *>(x)[q];
            const VT pv = reinterpret_cast<const VT *>(p)[q], av = reinterpret_cast<const VT *>(ap)[q];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                xv[e] = p_add(xv[e], p_mul(pv[e], alpha));
                rv[e] = p_add(rv[e], -p_mul(av[e], alpha));
            }
            reinterpret_cast<VT *>(x)[q] = xv;
            reinterpret_cast<VT *>(r)[q] = rv;
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            s_rr = p_add(s_rr, p_mul(rv[e], rv[e]));
            s_rz = p_add(s_rz, p_mul(rv[e], p_div(rv[e], dv[e])));
        }
    }
    for (uint64_t i = nv * V + tid; i < n; i += nthreads) {
        T ri = r[i];
        if (UPDATE) {
            x[i] = p_add(x[i], p_mul(p[i], alpha));
            ri = p_add(ri, -p_mul(ap[i], alpha));
            r[i] = ri;
        }
        s_rr = p_add(s_rr, p_mul(ri, ri));
        s_rz = p_add(s_rz, p_mul(ri, p_div(ri, d[i])));
    }
    block_sums(s_rr, s_rz, part_rr, part_rz);
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_pcg_fold(const T *__restrict__ part_a, const T *__restrict__ part_b, unsigned n_parts, T *__restrict__ out /* [2] */) {
    T a = T(0), b = T(0);
    for (unsigned k = threadIdx.x; k < n_parts; k += kBlock) { a = p_add(a, part_a[k]); b = p_add(b, part_b[k]); }
    block_sums(a, b, out, out + 1);  // (one block: blockIdx.x == 0)
}

// p = p*beta + r/d   (FIRST: p = r/d)
template <typename T, bool FIRST>
__global__ void __launch_bounds__(kBlock)
k_pcg_p(T *__restrict__ p, const T *__restrict__ r, const T *__restrict__ d, uint64_t n, T beta) {
    constexpr int V = 16 / sizeof(T);
    typedef T VT __attribute__((ext_vector_type(V)));
    const uint64_t nv = n / V, tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t q = tid; q < nv; q += nthreads) {
        const VT rv = reinterpret_cast<const VT *>(r)[q], dv = reinterpret_cast<const VT *>(d)[q];
        VT pv;
        if (!FIRST) pv = reinterpret_cast<const VT *>(p)[q];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const T z = p_div(rv[e], dv[e]);
            pv[e] = FIRST ? z : p_add(p_mul(pv[e], beta), z);
        }
        reinterpret_cast<VT *>(p)[q] = pv;
    }
    for (uint64_t i = nv * V + tid; i < n; i += nthreads) {
        const T z = p_div(r[i], d[i]);
        p[i] = FIRST ? z : p_add(p_mul(p[i], beta), z);
    }
}

unsigned pcg_grid(size_t n) {
    uint64_t b = (n + kBlock - 1) / kBlock;
    if (b > (uint64_t)kPcgBlocks) b = kPcgBlocks;
    return (unsigned)(b ? b : 1);
}

template <typename T>
int pcg_t(smh_crs *m, const T *b_host, T *x_host, size_t n, double tol, size_t iter_max, int variant, size_t *iters_out, double *rr_out) {
    const int dt = sizeof(T) == 8 ? SMH_F64 : SMH_F32;
    hipStream_t s = nullptr;
    T *d_x = nullptr, *d_r = nullptr, *d_p = nullptr, *d_ap = nullptr, *d_d = nullptr, *d_part = nullptr, *d_out = nullptr, *h_out = nullptr;
    unsigned long long *d_bad = nullptr;
    size_t iters = 0;
    double rr = 0.0;
    const unsigned grid = pcg_grid(n);
    auto go = [&]() -> int {
        SMH_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        const size_t vb = (n ? n : 1) * sizeof(T);
        SMH_HIP(hipMalloc((void **)&d_x, vb)); SMH_HIP(hipMalloc((void **)&d_r, vb)); SMH_HIP(hipMalloc((void **)&d_p, vb));
        SMH_HIP(hipMalloc((void **)&d_ap, vb)); SMH_HIP(hipMalloc((void **)&d_d, vb));
        SMH_HIP(hipMalloc((void **)&d_part, (2 * (size_t)kPcgBlocks + (size_t)kReducePartials + 8) * sizeof(T)));
        SMH_HIP(hipMalloc((void **)&d_out, 4 * sizeof(T)));
        SMH_HIP(hipMalloc((void **)&d_bad, sizeof(unsigned long long)));
        SMH_HIP(hipHostMalloc((void **)&h_out, 4 * sizeof(T), hipHostMallocDefault));
        T *part_rr = d_part, *part_rz = d_part + kPcgBlocks, *dot_scratch = d_part + 2 * kPcgBlocks;
        const uint32_t *off = m->d_off, *col = m->d_col;
        const void *val = m->d_val;
        SMH_HIP(hipStreamSynchronize(m->stream));
        // diag(A); a zero diagonal cannot be divided by
        unsigned long long bad = ~0ull;
        SMH_HIP(hipMemcpyAsync(d_bad, &bad, sizeof bad, hipMemcpyHostToDevice, s));
        if (n) hipLaunchKernelGGL((k_pcg_diag<T>), dim3(pcg_grid(n) * 4), dim3(kBlock), 0, s, off, col, (const T *)val, (uint64_t)n, d_d, d_bad);
        SMH_HIP(hipGetLastError());
        SMH_HIP(hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, s));
        if (n) {
            SMH_HIP(hipMemcpyAsync(d_r, b_host, n * sizeof(T), hipMemcpyHostToDevice, s));
            SMH_HIP(hipMemcpyAsync(d_x, x_host, n * sizeof(T), hipMemcpyHostToDevice, s));
        }
        SMH_HIP(hipStreamSynchronize(s));
        if (bad != ~0ull) return fail(SMH_ERR_INVALID, "Jacobi preconditioner: zero or absent diagonal entry in row %llu", bad);
        // r = b - A x; p = r / d; rr, rz
        SMH_TRY(smh_crs_spmv_dev(m, d_x, n, d_ap, variant, s));
        if (n) SMH_TRY(launch_ew(dt, Ew::Sub, d_r, d_ap, n, 0.0, nullptr, s));
        hipLaunchKernelGGL((k_pcg_update<T, false>), dim3(grid), dim3(kBlock), 0, s, d_x, d_r, d_p, d_ap, d_d, (uint64_t)n, T(0), part_rr, part_rz);
        hipLaunchKernelGGL((k_pcg_fold<T>), dim3(1), dim3(kBlock), 0, s, part_rr, part_rz, grid, d_out);
        hipLaunchKernelGGL((k_pcg_p<T, true>), dim3(grid), dim3(kBlock), 0, s, d_p, d_r, d_d, (uint64_t)n, T(0));
        SMH_HIP(hipGetLastError());
        SMH_HIP(hipMemcpyAsync(h_out, d_out, 2 * sizeof(T), hipMemcpyDeviceToHost, s));
        SMH_HIP(hipStreamSynchronize(s));
        T rr_t = h_out[0], rz = h_out[1];
        rr = (double)rr_t;
        for (size_t k = 0; k < iter_max; ++k) {
            SMH_TRY(smh_crs_spmv_dev(m, d_p, n, d_ap, variant, s));
            if (n) SMH_TRY(launch_dot(dt, d_p, d_ap, n, dot_scratch, d_out + 2, s));
            else SMH_HIP(hipMemsetAsync(d_out + 2, 0, sizeof(T), s));
            SMH_HIP(hipMemcpyAsync(h_out + 2, d_out + 2, sizeof(T), hipMemcpyDeviceToHost, s));
            SMH_HIP(hipStreamSynchronize(s));
            const T alpha = rz / h_out[2];
            hipLaunchKernelGGL((k_pcg_update<T, true>), dim3(grid), dim3(kBlock), 0, s, d_x, d_r, d_p, d_ap, d_d, (uint64_t)n, alpha, part_rr, part_rz);
            hipLaunchKernelGGL((k_pcg_fold<T>), dim3(1), dim3(kBlock), 0, s, part_rr, part_rz, grid, d_out);
            SMH_HIP(hipGetLastError());
            SMH_HIP(hipMemcpyAsync(h_out, d_out, 2 * sizeof(T), hipMemcpyDeviceToHost, s));
            SMH_HIP(hipStreamSynchronize(s));
            ++iters;
            rr_t = h_out[0];
            rr = (double)rr_t;
            if (std::sqrt(rr) < tol) break;
            const T beta = h_out[1] / rz;
            rz = h_out[1];
            hipLaunchKernelGGL((k_pcg_p<T, false>), dim3(grid), dim3(kBlock), 0, s, d_p, d_r, d_d, (uint64_t)n, beta);
            SMH_HIP(hipGetLastError());
        }
        if (n) SMH_HIP(hipMemcpyAsync(x_host, d_x, n * sizeof(T), hipMemcpyDeviceToHost, s));
        SMH_HIP(hipStreamSynchronize(s));
        return SMH_OK;
    };
    const int rc = go();
    if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    (void)hipFree(d_x); (void)hipFree(d_r); (void)hipFree(d_p); (void)hipFree(d_ap); (void)hipFree(d_d); (void)hipFree(d_part);
    (void)hipFree(d_out); (void)hipFree(d_bad);
    if (h_out) (void)hipHostFree(h_out);
    if (iters_out) *iters_out = iters;
    if (rr_out) *rr_out = rr;
    return rc;
}

}  // namespace

extern "C" int smh_pcg_jacobi_solve(smh_crs *m, const void *b_host, size_t b_len, void *x_host_inout, size_t x_len, double tol,
                                    size_t iter_max, int variant, size_t *iters_out, double *rr_out) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    const size_t n = smh_crs_n_rows(m);
    if (n != smh_crs_n_cols(m)) return fail(SMH_ERR_NOT_SQUARE, "Matrix is not symmetric");                       // linearsolver.rs:30-32
    if (n != b_len || n != x_len) return fail(SMH_ERR_DIM_MISMATCH, "Matrix and vector size mismatch");          // :33-36
    if (n && (!b_host || !x_host_inout)) return fail(SMH_ERR_INVALID, "NULL host vector");
    if (smh_crs_dtype(m) == SMH_F64)
        return pcg_t<double>(m, (const double *)b_host, (double *)x_host_inout, n, tol, iter_max, variant, iters_out, rr_out);
    return pcg_t<float>(m, (const float *)b_host, (float *)x_host_inout, n, tol, iter_max, variant, iters_out, rr_out);
}
