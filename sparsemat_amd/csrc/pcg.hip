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
// r and leaves the block partials of r.r and r.(r/d) (r, Ap, d in; r out), one sweep adds p*alpha to x and rebuilds p
// (p, x, r, d in; p, x out): 10 n values per iteration beside the SpMV (the plain CG moves 8 n; the two reads of d are
// what the preconditioner costs), p.Ap out of the SpMV epilogue when the kernel offers it.  All scalars (r.r, r.z, p.Ap,
// alpha, beta, the stop flag, the iteration count) live in device memory as in cg.hip: the host replays a hipGraph of 8
// bodies and polls one small block per batch -- no synchronisation inside an iteration.
// Reductions: fixed grid, per-thread strided sums, wave butterfly, LDS across waves, one block folds the partials in
// index order -- deterministic.
#include "internal.hpp"

#include <cmath>
#include <cstring>

using namespace smh;

namespace smh {
unsigned reduce_blocks(size_t n);  // blas1.hip
}

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

// ---- device-resident scalars (like cg.hip): nothing of an iteration passes through the host ---------------------------
template <typename T>
struct PcgScalars {
    T rr, rz, pap, alpha, beta;
    uint32_t converged;
    uint32_t active;   // this loop body runs (not converged, fewer than iter_max bodies entered)
    uint32_t entered;  // ... was entered: its x update is due even if the stop test then ends the loop
    uint32_t pad_;
    uint64_t iters;
    uint64_t iter_max;
    double tol;
};

template <typename T>
__global__ void k_pcg_init(PcgScalars<T> *sc, double tol, uint64_t iter_max) {
    sc->rr = sc->rz = sc->pap = sc->alpha = sc->beta = T(0);
    sc->converged = sc->active = sc->entered = sc->pad_ = 0;
    sc->iters = 0;
    sc->iter_max = iter_max;
    sc->tol = tol;
}

template <typename T>
__device__ __forceinline__ T block_sum1(T a) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) a = p_add(a, (T)__shfl_down(a, o, kWave));
    __shared__ T sa1[kBlock / kWave];
    if ((threadIdx.x & (kWave - 1)) == 0) sa1[threadIdx.x / kWave] = a;
    __syncthreads();
    T t = T(0);
    if (threadIdx.x == 0) {
        t = sa1[0];
        for (int w = 1; w < kBlock / kWave; ++w) t = p_add(t, sa1[w]);
    }
    return t;  // (thread 0)
}

// out[b] = sum of block b's strided share of in[0..n): first stage of folding the SpMV epilogue's per-tile partials
template <typename T>
__global__ void __launch_bounds__(kBlock) k_pcg_sum_stage1(const T *__restrict__ in, uint64_t n, T *__restrict__ out) {
    T a = T(0);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) a = p_add(a, in[i]);
    const T t = block_sum1(a);
    if (threadIdx.x == 0) out[blockIdx.x] = t;
}

// p.Ap = fold(partials); decide whether this body runs; alpha = r.z / p.Ap
template <typename T>
__global__ void __launch_bounds__(kBlock) k_pcg_alpha(PcgScalars<T> *sc, const T *__restrict__ partials, uint32_t count) {
    T a = T(0);
    for (uint32_t i = threadIdx.x; i < count; i += kBlock) a = p_add(a, partials[i]);
    const T pap = block_sum1(a);
    if (threadIdx.x == 0) {
        const bool active = !sc->converged && sc->iters < sc->iter_max;
        sc->active = sc->entered = active ? 1u : 0u;
        if (active) {
            sc->iters += 1;
            sc->pap = pap;
            sc->alpha = p_div(sc->rz, pap);
        }
    }
}

// UPDATE: r -= ap*alpha first (gated by "active"); always: partials of r.r and r.(r/d)
template <typename T, bool UPDATE>
__global__ void __launch_bounds__(kBlock)
k_pcg_update(const PcgScalars<T> *__restrict__ sc, T *__restrict__ r, const T *__restrict__ ap, const T *__restrict__ d, uint64_t n,
             T *__restrict__ part_rr, T *__restrict__ part_rz) {
    // 16 bytes per lane and array (the buffers are hipMalloc'ed: aligned); the last n % V elements one by one
    constexpr int V = 16 / sizeof(T);
    typedef T VT __attribute__((ext_vector_type(V)));
    T alpha = T(0);
    if (UPDATE) {
        if (!sc->active) return;  // block-uniform
        alpha = sc->alpha;
    }
    T s_rr = T(0), s_rz = T(0);
    const uint64_t nv = n / V, tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t q = tid; q < nv; q += nthreads) {
        VT rv = __builtin_nontemporal_load(reinterpret_cast<const VT *>(r) + q);
        const VT dv = __builtin_nontemporal_load(reinterpret_cast<const VT *>(d) + q);
        if (UPDATE) {
            const VT av = __builtin_nontemporal_load(reinterpret_cast<const VT *>(ap) + q);
#pragma unroll
            for (int e = 0; e < V; ++e) rv[e] = p_add(rv[e], -p_mul(av[e], alpha));
            __builtin_nontemporal_store(rv, reinterpret_cast<VT *>(r) + q);
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
            ri = p_add(ri, -p_mul(ap[i], alpha));
            r[i] = ri;
        }
        s_rr = p_add(s_rr, p_mul(ri, ri));
        s_rz = p_add(s_rz, p_mul(ri, p_div(ri, d[i])));
    }
    block_sums(s_rr, s_rz, part_rr, part_rz);
}

// FIRST: rr, rz of the initial residual.  Else: the stop test on r.r (before beta, linearsolver.rs:52-54), then beta = rz' / rz
template <typename T, bool FIRST>
__global__ void __launch_bounds__(kBlock)
k_pcg_beta(PcgScalars<T> *sc, const T *__restrict__ part_a, const T *__restrict__ part_b, unsigned n_parts) {
    if (!FIRST && !sc->active) return;
    T a = T(0), b = T(0);
    for (unsigned k = threadIdx.x; k < n_parts; k += kBlock) { a = p_add(a, part_a[k]); b = p_add(b, part_b[k]); }
    __shared__ T out2[2];
    block_sums(a, b, out2, out2 + 1);  // (one block: blockIdx.x == 0; thread 0 writes)
    if (threadIdx.x == 0) {
        const T rr = out2[0], rz = out2[1];
        sc->rr = rr;
        if (FIRST) {
            sc->rz = rz;
        } else if (sqrt((double)rr) < sc->tol) {
            sc->converged = 1;
            sc->active = 0;
        } else {
            sc->beta = p_div(rz, sc->rz);
            sc->rz = rz;
        }
    }
}

// x += p*alpha (the entered body's update, linearsolver.rs:47 -- carried out here because this sweep reads p anyway), then
// p = p*beta + r/d while the loop goes on.  FIRST: p = r/d.
template <typename T, bool FIRST>
__global__ void __launch_bounds__(kBlock)
k_pcg_p(const PcgScalars<T> *__restrict__ sc, T *__restrict__ p, const T *__restrict__ r, const T *__restrict__ d, T *__restrict__ x, uint64_t n) {
    constexpr int V = 16 / sizeof(T);
    typedef T VT __attribute__((ext_vector_type(V)));
    T alpha = T(0), beta = T(0);
    bool rebuild = true;
    if (!FIRST) {
        if (!sc->entered) return;
        alpha = sc->alpha;
        beta = sc->beta;
        rebuild = sc->active != 0;
    }
    const uint64_t nv = n / V, tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t q = tid; q < nv; q += nthreads) {
        VT pv;
        if (!FIRST) {
            pv = __builtin_nontemporal_load(reinterpret_cast<const VT *>(p) + q);
            VT xv = __builtin_nontemporal_load(reinterpret_cast<const VT *>(x) + q);
#pragma unroll
            for (int e = 0; e < V; ++e) xv[e] = p_add(xv[e], p_mul(pv[e], alpha));
            __builtin_nontemporal_store(xv, reinterpret_cast<VT *>(x) + q);
        }
        if (rebuild) {
            const VT rv = __builtin_nontemporal_load(reinterpret_cast<const VT *>(r) + q), dv = __builtin_nontemporal_load(reinterpret_cast<const VT *>(d) + q);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const T z = p_div(rv[e], dv[e]);
                pv[e] = FIRST ? z : p_add(p_mul(pv[e], beta), z);
            }
            __builtin_nontemporal_store(pv, reinterpret_cast<VT *>(p) + q);
        }
    }
    for (uint64_t i = nv * V + tid; i < n; i += nthreads) {
        if (!FIRST) x[i] = p_add(x[i], p_mul(p[i], alpha));
        if (rebuild) {
            const T z = p_div(r[i], d[i]);
            p[i] = FIRST ? z : p_add(p_mul(p[i], beta), z);
        }
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
    T *d_x = nullptr, *d_r = nullptr, *d_p = nullptr, *d_ap = nullptr, *d_d = nullptr, *d_part = nullptr, *d_dot = nullptr;
    PcgScalars<T> *d_sc = nullptr, *h_sc = nullptr;
    unsigned long long *d_bad = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    size_t iters = 0;
    double rr = 0.0;
    const unsigned grid = pcg_grid(n);
    auto go = [&]() -> int {
        SMH_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        const size_t vb = (n ? n : 1) * sizeof(T);
        SMH_HIP(hipMalloc((void **)&d_x, vb)); SMH_HIP(hipMalloc((void **)&d_r, vb)); SMH_HIP(hipMalloc((void **)&d_p, vb));
        SMH_HIP(hipMalloc((void **)&d_ap, vb)); SMH_HIP(hipMalloc((void **)&d_d, vb));
        SMH_HIP(hipMalloc((void **)&d_part, (2 * (size_t)kPcgBlocks + 2 * (size_t)kReducePartials + 8) * sizeof(T)));
        SMH_HIP(hipMalloc((void **)&d_sc, sizeof(PcgScalars<T>)));
        SMH_HIP(hipMalloc((void **)&d_bad, sizeof(unsigned long long)));
        SMH_HIP(hipHostMalloc((void **)&h_sc, sizeof(PcgScalars<T>), hipHostMallocDefault));
        T *part_rr = d_part, *part_rz = d_part + kPcgBlocks, *dot_scratch = d_part + 2 * kPcgBlocks, *fold_scratch = dot_scratch + kReducePartials + 8;
        // p.Ap: out of the SpMV epilogue when the kernel offers it (K1s), else a separate two-stage dot
        const size_t n_dot = spmv_fused_dot_partials(m, n, variant);
        if (n_dot) SMH_HIP(hipMalloc((void **)&d_dot, n_dot * sizeof(T)));
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
        hipLaunchKernelGGL((k_pcg_init<T>), dim3(1), dim3(1), 0, s, d_sc, tol, (uint64_t)iter_max);
        SMH_TRY(spmv_enqueue(m, d_x, n, d_ap, variant, s));
        if (n) SMH_TRY(launch_ew(dt, Ew::Sub, d_r, d_ap, n, 0.0, nullptr, s));
        hipLaunchKernelGGL((k_pcg_update<T, false>), dim3(grid), dim3(kBlock), 0, s, d_sc, d_r, d_ap, d_d, (uint64_t)n, part_rr, part_rz);
        hipLaunchKernelGGL((k_pcg_beta<T, true>), dim3(1), dim3(kBlock), 0, s, d_sc, part_rr, part_rz, grid);
        hipLaunchKernelGGL((k_pcg_p<T, true>), dim3(grid), dim3(kBlock), 0, s, d_sc, d_p, d_r, d_d, d_x, (uint64_t)n);
        SMH_HIP(hipGetLastError());
        // one loop body: SpMV (+ p.Ap), alpha, update of r + partials, stop test / beta, x and p
        auto body = [&]() -> int {
            SMH_TRY(spmv_enqueue(m, d_p, n, d_ap, variant, s, d_dot));
            if (d_dot && n_dot > (size_t)kReducePartials) {
                const unsigned fb = reduce_blocks(n_dot);
                hipLaunchKernelGGL((k_pcg_sum_stage1<T>), dim3(fb), dim3(kBlock), 0, s, d_dot, (uint64_t)n_dot, fold_scratch);
                hipLaunchKernelGGL((k_pcg_alpha<T>), dim3(1), dim3(kBlock), 0, s, d_sc, fold_scratch, fb);
            } else if (d_dot) {
                hipLaunchKernelGGL((k_pcg_alpha<T>), dim3(1), dim3(kBlock), 0, s, d_sc, d_dot, (uint32_t)n_dot);
            } else {
                if (n) SMH_TRY(launch_dot(dt, d_p, d_ap, n, dot_scratch, dot_scratch + kReducePartials, s));
                else SMH_HIP(hipMemsetAsync(dot_scratch + kReducePartials, 0, sizeof(T), s));
                hipLaunchKernelGGL((k_pcg_alpha<T>), dim3(1), dim3(kBlock), 0, s, d_sc, dot_scratch + kReducePartials, 1u);
            }
            hipLaunchKernelGGL((k_pcg_update<T, true>), dim3(grid), dim3(kBlock), 0, s, d_sc, d_r, d_ap, d_d, (uint64_t)n, part_rr, part_rz);
            hipLaunchKernelGGL((k_pcg_beta<T, false>), dim3(1), dim3(kBlock), 0, s, d_sc, part_rr, part_rz, grid);
            hipLaunchKernelGGL((k_pcg_p<T, false>), dim3(grid), dim3(kBlock), 0, s, d_sc, d_p, d_r, d_d, d_x, (uint64_t)n);
            SMH_HIP(hipGetLastError());
            return SMH_OK;
        };
        // batches of check_every bodies, captured once into a hipGraph and replayed (bodies past the stop are no-ops on the
        // device); the host polls the scalar block once per batch
        const size_t check_every = 8;
        if (iter_max > check_every && hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            int crc = SMH_OK;
            for (size_t i = 0; i < check_every && crc == SMH_OK; ++i) crc = body();
            const hipError_t ce = hipStreamEndCapture(s, &graph);
            if (crc != SMH_OK || ce != hipSuccess || !graph || hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0) != hipSuccess) {
                graph_exec = nullptr;  // plain stream launches instead
                (void)hipGetLastError();
            }
        } else {
            (void)hipGetLastError();
        }
        size_t launched = 0;
        bool converged = false;
        auto poll = [&]() -> int {
            SMH_HIP(hipMemcpyAsync(h_sc, d_sc, sizeof(PcgScalars<T>), hipMemcpyDeviceToHost, s));
            SMH_HIP(hipStreamSynchronize(s));
            converged = h_sc->converged != 0;
            iters = (size_t)h_sc->iters;
            rr = (double)h_sc->rr;
            return SMH_OK;
        };
        while (launched < iter_max) {
            size_t batch = iter_max - launched < check_every ? iter_max - launched : check_every;
            if (graph_exec) {
                SMH_HIP(hipGraphLaunch(graph_exec, s));
                batch = check_every;
            } else {
                for (size_t i = 0; i < batch; ++i) SMH_TRY(body());
            }
            launched += batch;
            SMH_TRY(poll());
            if (converged) break;
        }
        if (iter_max == 0) SMH_TRY(poll());
        if (n) SMH_HIP(hipMemcpyAsync(x_host, d_x, n * sizeof(T), hipMemcpyDeviceToHost, s));
        SMH_HIP(hipStreamSynchronize(s));
        return SMH_OK;
    };
    const int rc = go();
    char keep[512];
    strncpy(keep, smh_last_error(), sizeof keep);
    keep[sizeof keep - 1] = 0;
    if (s) (void)hipStreamSynchronize(s);
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    if (graph) (void)hipGraphDestroy(graph);
    if (s) (void)hipStreamDestroy(s);
    (void)hipFree(d_x); (void)hipFree(d_r); (void)hipFree(d_p); (void)hipFree(d_ap); (void)hipFree(d_d); (void)hipFree(d_part);
    (void)hipFree(d_dot); (void)hipFree(d_sc); (void)hipFree(d_bad);
    if (h_sc) (void)hipHostFree(h_sc);
    (void)hipGetLastError();
    if (rc != SMH_OK) return fail(rc, "%s", keep);
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
