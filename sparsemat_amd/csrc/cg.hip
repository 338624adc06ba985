// cg.hip -- K5: device-resident conjugate-gradient iteration for gfx950.
//
// Restates ConjugateGradient::solve (reference linearsolver.rs:27-61) as a stream of kernels whose
// scalars (r.r, p.Ap, alpha, beta, the stop flag, the iteration count) never leave HBM:
//
//   per iteration   SpMV  Ap = A p  (+ per-tile partials of p.Ap where the kernel can) (launched by the caller)  :43
//                   [k_sum_stage1 when more than 1024 partials came back / a two-stage dot when none did]       :45
//                   k_cg_par_update : fold the partials, alpha = rr / pAp, decide "active" -- every workgroup for
//                                     itself, workgroup 0 writes the scalars --; r -= round(Ap*alpha); partials r.r :45, :49-51
//                   k_cg_par_p      : fold, rr_prev/rr, stop if sqrt(f64(rr)) < tol, else beta (likewise);
//                                     x += round(p*alpha) (:47), then p = round(p*beta) + r                      :50-59
//   (until round 4 alpha and beta were one-workgroup kernels of their own, k_cg_alpha / k_cg_beta: two more launches per
//   iteration; the fused kernels fold in the same order, so every result kept its bits)
//
// `*x += p * alpha` (:47) is carried out by the sweep that rebuilds p: it reads the old p anyway, so x costs one read and
// one write there instead of p AND x in the update sweep -- 8 n instead of 9 n values of vector traffic per iteration,
// the same operations on the same operands (x does not feed anything else inside an iteration), bit-identical results.
// That sweep therefore runs whenever the body was ENTERED ("entered"), and only its p part is gated by "active": the
// iteration that converges still delivers its x.
//
// "active" = not converged and fewer than iter_max bodies entered; once it drops, the gated
// kernels are no-ops, so the host may enqueue iterations in batches and poll the flag lazily:
// x, r, p are exactly those of the iteration in which the reference would have left its loop.
// Element-wise updates round the product and the sum separately (bit-identical to the reference
// for equal alpha/beta); the reductions are fixed trees (bitwise reproducible).
#include "internal.hpp"

namespace smh {

template <typename T>
struct CgScalars {
    T rr, rr_prev, pap, alpha, beta;
    uint32_t converged;
    uint32_t active;
    uint32_t entered;  // the current loop body was entered: its x update is due (set with alpha)
    uint32_t pad_;
    uint64_t iters;
    uint64_t iter_max;
    double tol;
};

template <typename T> struct CgVec;
template <> struct CgVec<float> { typedef float type __attribute__((ext_vector_type(4))); static constexpr int N = 4; };
template <> struct CgVec<double> { typedef double type __attribute__((ext_vector_type(2))); static constexpr int N = 2; };

__device__ __forceinline__ float cg_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ double cg_mul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ float cg_add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double cg_add(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ float cg_sub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ double cg_sub(double a, double b) { return __dsub_rn(a, b); }

template <typename T>
__device__ __forceinline__ T cg_block_sum(T v, T *s_w) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) s_w[wave] = v;
    __syncthreads();
    T r = T(0);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) r += s_w[w];
    }
    return r;
}

// out[b] = sum of this block's strided share of in[0..n)   (first stage of folding many partial sums)
template <typename T>
__global__ void __launch_bounds__(kBlock) k_sum_stage1(const T *__restrict__ in, uint64_t n, T *__restrict__ out) {
    __shared__ T s_w[kBlock / kWave];
    T acc = T(0);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        acc += in[i];
    const T r = cg_block_sum<T>(acc, s_w);
    if (threadIdx.x == 0) out[blockIdx.x] = r;
}

template <typename T>
__global__ void k_cg_init(CgScalars<T> *sc, double tol, uint64_t iter_max) {
    sc->rr = sc->rr_prev = sc->pap = sc->alpha = sc->beta = T(0);
    sc->converged = 0;
    sc->active = 0;
    sc->entered = 0;
    sc->pad_ = 0;
    sc->iters = 0;
    sc->iter_max = iter_max;
    sc->tol = tol;
}

// rr = fold(partials)   (after the initial r.r reduction, linearsolver.rs:40)
template <typename T>
__global__ void __launch_bounds__(kBlock) k_cg_set_rr(CgScalars<T> *sc, const T *__restrict__ partials, uint32_t count) {
    __shared__ T s_w[kBlock / kWave];
    T acc = T(0);
    for (uint32_t i = threadIdx.x; i < count; i += kBlock) acc += partials[i];
    const T r = cg_block_sum<T>(acc, s_w);
    if (threadIdx.x == 0) sc->rr = r;
}

// r -= round(Ap * alpha) on [0, n), partials[blockIdx.x] = this workgroup's share of r.r   (linearsolver.rs:49-51)
template <typename T, bool VEC>
__device__ __forceinline__ void cg_update_body(T alpha, T *__restrict__ r, const T *__restrict__ ap, uint64_t n, T *__restrict__ partials, T *s_w) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    T acc = T(0);
    if constexpr (VEC) {
        typedef typename CgVec<T>::type V;
        constexpr int N = CgVec<T>::N;
        const uint64_t nv = n / N;
        V *rv = reinterpret_cast<V *>(r);
        const V *apv = reinterpret_cast<const V *>(ap);
        for (uint64_t i = tid; i < nv; i += nthreads) {
            V rr = __builtin_nontemporal_load(rv + i);  // (streams far larger than the caches: the non-temporal forms move 13 % more, blas1.hip)
            const V aa = __builtin_nontemporal_load(apv + i);
#pragma unroll
            for (int e = 0; e < N; ++e) {
                rr[e] = cg_sub(rr[e], cg_mul(aa[e], alpha));  // r -= mat_p * alpha        :49
                acc += rr[e] * rr[e];                         // r.norm_squared()          :51
            }
            __builtin_nontemporal_store(rr, rv + i);
        }
        for (uint64_t i = nv * N + tid; i < n; i += nthreads) {
            const T t = cg_sub(r[i], cg_mul(ap[i], alpha));
            r[i] = t;
            acc += t * t;
        }
    } else {
        for (uint64_t i = tid; i < n; i += nthreads) {
            const T t = cg_sub(r[i], cg_mul(ap[i], alpha));
            r[i] = t;
            acc += t * t;
        }
    }
    const T s = cg_block_sum<T>(acc, s_w);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// x += round(p * alpha) (:47) and, when `rebuild`, p = round(p * beta) + r (:58-59), on [0, n)
template <typename T, bool VEC>
__device__ __forceinline__ void cg_p_body(bool rebuild, T alpha, T beta, T *__restrict__ p, const T *__restrict__ r, T *__restrict__ x, uint64_t n) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    if constexpr (VEC) {
        typedef typename CgVec<T>::type V;
        constexpr int N = CgVec<T>::N;
        const uint64_t nv = n / N;
        V *pv = reinterpret_cast<V *>(p);
        V *xv = reinterpret_cast<V *>(x);
        const V *rv = reinterpret_cast<const V *>(r);
        if (rebuild) {
            for (uint64_t i = tid; i < nv; i += nthreads) {
                V pp = __builtin_nontemporal_load(pv + i), xx = __builtin_nontemporal_load(xv + i);
                const V rr = __builtin_nontemporal_load(rv + i);
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    xx[e] = cg_add(xx[e], cg_mul(pp[e], alpha));  // *x += p.clone() * alpha        :47
                    pp[e] = cg_add(cg_mul(pp[e], beta), rr[e]);   // p.scale(beta); p.add(&r)       :58-59
                }
                __builtin_nontemporal_store(xx, xv + i);
                __builtin_nontemporal_store(pp, pv + i);
            }
        } else {
            for (uint64_t i = tid; i < nv; i += nthreads) {
                V xx = __builtin_nontemporal_load(xv + i);
                const V pp = __builtin_nontemporal_load(pv + i);
#pragma unroll
                for (int e = 0; e < N; ++e) xx[e] = cg_add(xx[e], cg_mul(pp[e], alpha));
                __builtin_nontemporal_store(xx, xv + i);
            }
        }
        for (uint64_t i = nv * N + tid; i < n; i += nthreads) {
            x[i] = cg_add(x[i], cg_mul(p[i], alpha));
            if (rebuild) p[i] = cg_add(cg_mul(p[i], beta), r[i]);
        }
    } else {
        for (uint64_t i = tid; i < n; i += nthreads) {
            x[i] = cg_add(x[i], cg_mul(p[i], alpha));
            if (rebuild) p[i] = cg_add(cg_mul(p[i], beta), r[i]);
        }
    }
}


// ---- the two launches of an iteration's tail (single matrix: cg_iter_tail; row-partitioned: par.hip) ----------------------------
// alpha + the x / r update in one launch, beta with the stop test + the p sweep in the other: every workgroup folds the `nb` partial
// sums itself (the product's per-tile p.Ap partials or their first fold; the update's r.r partials; across row blocks: the blocks'
// values) -- in exactly the order the one-workgroup kernels of rounds 1-3 folded them, so every workgroup (of every block) gets the
// same bits -- and takes the same decision; workgroup 0 writes the scalar block.  The scalars are DOUBLE-BUFFERED: a launch reads only `in` and writes only `out`
// (the host swaps them from launch to launch), so no workgroup can see a half-updated block.  Two launches fewer per block and
// iteration (of ten).
template <typename T>
__device__ __forceinline__ T cg_fold_everywhere(const T *__restrict__ vals, uint32_t count, T *s_w, T *s_one) {
    T acc = T(0);
    for (uint32_t i = threadIdx.x; i < count; i += kBlock) acc += vals[i];
    const T r = cg_block_sum<T>(acc, s_w);  // (thread 0 holds it)
    if (threadIdx.x == 0) *s_one = r;
    __syncthreads();
    return *s_one;
}

template <typename T, bool VEC>
__global__ void __launch_bounds__(kBlock)
k_cg_par_update(const CgScalars<T> *__restrict__ in, CgScalars<T> *__restrict__ out, const T *__restrict__ pap_vals, uint32_t nb,
                T *__restrict__ r, const T *__restrict__ ap, uint64_t n, T *__restrict__ partials) {
    __shared__ T s_w[kBlock / kWave];
    __shared__ T s_one;
    const T pap = cg_fold_everywhere<T>(pap_vals, nb, s_w, &s_one);
    const bool active = !in->converged && in->iters < in->iter_max;
    const T alpha = active ? in->rr / pap : in->alpha;  // :45 (no breakdown guard, like the reference)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        CgScalars<T> o = *in;
        o.active = active ? 1u : 0u;
        o.entered = active ? 1u : 0u;
        if (active) {
            o.iters += 1;  // a loop body is entered (for _k in 0..iter_max, :41)
            o.pap = pap;
            o.alpha = alpha;
        }
        *out = o;
    }
    if (!active) return;  // (uniform over the grid: `in` is the same for every workgroup)
    cg_update_body<T, VEC>(alpha, r, ap, n, partials, s_w);
}

template <typename T, bool VEC>
__global__ void __launch_bounds__(kBlock)
k_cg_par_p(const CgScalars<T> *__restrict__ in, CgScalars<T> *__restrict__ out, const T *__restrict__ rr_vals, uint32_t nb,
           T *__restrict__ p, const T *__restrict__ r, T *__restrict__ x, uint64_t n) {
    __shared__ T s_w[kBlock / kWave];
    __shared__ T s_one;
    const bool was_active = in->active != 0, entered = in->entered != 0;
    T rr = T(0), beta = in->beta;
    bool conv = false;
    if (was_active) {
        rr = cg_fold_everywhere<T>(rr_vals, nb, s_w, &s_one);
        conv = sqrt((double)rr) < in->tol;  // :52-54, BEFORE the beta update
        if (!conv) beta = rr / in->rr;      // :56
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        CgScalars<T> o = *in;
        if (was_active) {
            o.rr_prev = in->rr;
            o.rr = rr;
            if (conv) { o.converged = 1; o.active = 0; }
            else o.beta = beta;
        }
        *out = o;
    }
    if (!entered) return;
    cg_p_body<T, VEC>(was_active && !conv, in->alpha, beta, p, r, x, n);
}

// ---- host-side driver pieces (called from capi.hip) ----------------------------------------------
unsigned reduce_blocks(size_t n);  // blas1.hip

static inline bool cg_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

size_t cg_scalars_bytes(int dtype) { return dtype == SMH_F64 ? sizeof(CgScalars<double>) : sizeof(CgScalars<float>); }

template <typename T>
static int cg_begin_t(void *sc, const T *r, size_t n, T *partials, double tol, size_t iter_max, hipStream_t s) {
    hipLaunchKernelGGL(k_cg_init<T>, dim3(1), dim3(1), 0, s, (CgScalars<T> *)sc, tol, (uint64_t)iter_max);
    SMH_HIP(hipGetLastError());
    SMH_TRY(launch_dot(sizeof(T) == 8 ? SMH_F64 : SMH_F32, r, r, n, partials, partials + kReducePartials, s));
    // launch_dot folded into partials[kReducePartials]; copy it into the scalar block
    hipLaunchKernelGGL(k_cg_set_rr<T>, dim3(1), dim3(kBlock), 0, s, (CgScalars<T> *)sc, partials + kReducePartials, 1u);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

int cg_begin(int dtype, void *sc, const void *r, size_t n, void *partials, double tol, size_t iter_max, hipStream_t s) {
    if (dtype == SMH_F64) return cg_begin_t<double>(sc, (const double *)r, n, (double *)partials, tol, iter_max, s);
    return cg_begin_t<float>(sc, (const float *)r, n, (float *)partials, tol, iter_max, s);
}

// everything of one iteration AFTER the SpMV Ap = A p
template <typename T>
static int cg_iter_tail_t(void *scv, void *sc2v, T *x, T *r, T *p, const T *ap, size_t n, T *partials, const T *dot_partials,
                          uint32_t dot_count, hipStream_t s) {
    const CgScalars<T> *sc = (const CgScalars<T> *)scv;
    CgScalars<T> *sc2 = (CgScalars<T> *)sc2v;
    // 2 blocks per CU for the two vector sweeps of the tail: 2.51-2.57 ms per C4 iteration against 2.66-2.69 with 1024 / 2048
    // blocks in the same call (profiles/r01_cg_grid_sweep.log)
    static const unsigned grid_cap = getenv("SMH_CG_BLOCKS") ? (unsigned)atoi(getenv("SMH_CG_BLOCKS")) : 512u;  // tuning knob
    unsigned rb = reduce_blocks(n);
    if (rb > grid_cap) rb = grid_cap;
    const bool vec = cg_aligned16(x) && cg_aligned16(r) && cg_aligned16(p) && cg_aligned16(ap);
    // Round 4: the two one-workgroup scalar kernels (alpha; beta with the stop test) ride the sweeps that follow them -- every workgroup
    // folds the partial sums itself, in the order the one-workgroup kernels did (same bits), the scalars double-buffered (sc -> sc2 -> sc)
    // -- so an iteration is the product + 2 launches (+ one fold when the product left more than 1024 partials) instead of + 4: what a
    // launch-bound solve (BASELINE C1: 28 us per iteration) is made of.
    // p . Ap: either the SpMV epilogue already left per-tile partials (fused), or a separate two-stage dot
    const T *pap_vals;
    uint32_t pap_count;
    if (dot_partials && dot_count > (uint32_t)kReducePartials) {
        // many tiles: fold them with a full grid first (a single block folding 5e5 values costs ~0.6 ms); into the buffer's second half:
        // the update's workgroups read these while others already write their r.r partials into the first
        const unsigned fb = reduce_blocks(dot_count);
        T *stage1 = partials + kReducePartials + 8;
        hipLaunchKernelGGL(k_sum_stage1<T>, dim3(fb), dim3(kBlock), 0, s, dot_partials, (uint64_t)dot_count, stage1);
        SMH_HIP(hipGetLastError());
        pap_vals = stage1;
        pap_count = fb;
    } else if (dot_partials) {
        pap_vals = dot_partials;
        pap_count = dot_count;
    } else {
        SMH_TRY(launch_dot(sizeof(T) == 8 ? SMH_F64 : SMH_F32, p, ap, n, partials, partials + kReducePartials, s));
        pap_vals = partials + kReducePartials;
        pap_count = 1u;
    }
    if (vec)
        hipLaunchKernelGGL((k_cg_par_update<T, true>), dim3(rb), dim3(kBlock), 0, s, sc, sc2, pap_vals, pap_count, r, ap, (uint64_t)n, partials);
    else
        hipLaunchKernelGGL((k_cg_par_update<T, false>), dim3(rb), dim3(kBlock), 0, s, sc, sc2, pap_vals, pap_count, r, ap, (uint64_t)n, partials);
    SMH_HIP(hipGetLastError());
    uint64_t pb = (n / CgVec<T>::N + kBlock) / kBlock;
    static const uint64_t p_cap = getenv("SMH_CG_P_BLOCKS") ? (uint64_t)atoll(getenv("SMH_CG_P_BLOCKS")) : 512;  // tuning knob
    if (pb > p_cap) pb = p_cap;
    if (vec)
        hipLaunchKernelGGL((k_cg_par_p<T, true>), dim3((unsigned)pb), dim3(kBlock), 0, s, (const CgScalars<T> *)sc2, (CgScalars<T> *)scv, (const T *)partials, rb, p, r, x, (uint64_t)n);
    else
        hipLaunchKernelGGL((k_cg_par_p<T, false>), dim3((unsigned)pb), dim3(kBlock), 0, s, (const CgScalars<T> *)sc2, (CgScalars<T> *)scv, (const T *)partials, rb, p, r, x, (uint64_t)n);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

int cg_iter_tail(int dtype, void *sc, void *sc2, void *x, void *r, void *p, const void *ap, size_t n, void *partials,
                 const void *dot_partials, uint32_t dot_count, hipStream_t s) {
    if (dtype == SMH_F64)
        return cg_iter_tail_t<double>(sc, sc2, (double *)x, (double *)r, (double *)p, (const double *)ap, n, (double *)partials,
                                      (const double *)dot_partials, dot_count, s);
    return cg_iter_tail_t<float>(sc, sc2, (float *)x, (float *)r, (float *)p, (const float *)ap, n, (float *)partials,
                                 (const float *)dot_partials, dot_count, s);
}

// host view of the scalar block after a poll
void cg_read_scalars(int dtype, const void *host_copy, int *converged, uint64_t *iters, double *rr) {
    if (dtype == SMH_F64) {
        const CgScalars<double> *h = (const CgScalars<double> *)host_copy;
        *converged = (int)h->converged; *iters = h->iters; *rr = (double)h->rr;
    } else {
        const CgScalars<float> *h = (const CgScalars<float> *)host_copy;
        *converged = (int)h->converged; *iters = h->iters; *rr = (double)h->rr;
    }
}

}  // namespace smh

// ---- pieces of the row-partitioned solver (par.hip) ----------------------------------------------------------------
// The same kernels as above, cut where the partitioned solver has to fold across row blocks: a block reduces its own
// rows to ONE value (cg_fold), the blocks' values meet (peer-visible slots or an RCCL all-gather, par.hip), and every
// block folds the same `nb` values in the same fixed tree -- so all blocks take identical alpha / beta / stop decisions
// without a host round trip, and the result is bitwise reproducible.
namespace smh {

template <typename T>
static int cg_fold_t(const T *partials, uint32_t count, T *out, hipStream_t s) {
    hipLaunchKernelGGL(k_sum_stage1<T>, dim3(1), dim3(kBlock), 0, s, partials, (uint64_t)count, out);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

int cg_fold(int dtype, const void *partials, uint32_t count, void *out, hipStream_t s) {
    if (dtype == SMH_F64) return cg_fold_t<double>((const double *)partials, count, (double *)out, s);
    return cg_fold_t<float>((const float *)partials, count, (float *)out, s);
}

#define SMH_CG_PAR_SCALAR(NAME, KERNEL)                                                                            \
    int NAME(int dtype, void *sc, const void *vals, uint32_t nb, hipStream_t s) {                                  \
        if (dtype == SMH_F64)                                                                                      \
            hipLaunchKernelGGL(KERNEL<double>, dim3(1), dim3(kBlock), 0, s, (CgScalars<double> *)sc, (const double *)vals, nb); \
        else                                                                                                       \
            hipLaunchKernelGGL(KERNEL<float>, dim3(1), dim3(kBlock), 0, s, (CgScalars<float> *)sc, (const float *)vals, nb);    \
        SMH_HIP(hipGetLastError());                                                                                \
        return SMH_OK;                                                                                             \
    }
SMH_CG_PAR_SCALAR(cg_par_set_rr, k_cg_set_rr)  // rr = fold(vals)                         linearsolver.rs:40
#undef SMH_CG_PAR_SCALAR

int cg_par_init(int dtype, void *sc, double tol, size_t iter_max, hipStream_t s) {
    if (dtype == SMH_F64) hipLaunchKernelGGL(k_cg_init<double>, dim3(1), dim3(1), 0, s, (CgScalars<double> *)sc, tol, (uint64_t)iter_max);
    else hipLaunchKernelGGL(k_cg_init<float>, dim3(1), dim3(1), 0, s, (CgScalars<float> *)sc, tol, (uint64_t)iter_max);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

// p.Ap = fold(pap_vals[0..nb)); alpha; "active" (:45) -- then r -= round(Ap*alpha) on this block's rows and partials[0..*count_out)
// of r.r (:49-51).  Scalars: read from sc_in, written to sc_out (see k_cg_par_update).
template <typename T>
static int cg_par_update_t(const void *sc_in, void *sc_out, const T *pap_vals, uint32_t nb, T *r, const T *ap, size_t n, T *partials,
                           uint32_t *count_out, hipStream_t s) {
    unsigned rb = reduce_blocks(n);
    if (rb > 512u) rb = 512u;
    if (cg_aligned16(r) && cg_aligned16(ap))
        hipLaunchKernelGGL((k_cg_par_update<T, true>), dim3(rb), dim3(kBlock), 0, s, (const CgScalars<T> *)sc_in, (CgScalars<T> *)sc_out, pap_vals, nb, r, ap, (uint64_t)n, partials);
    else
        hipLaunchKernelGGL((k_cg_par_update<T, false>), dim3(rb), dim3(kBlock), 0, s, (const CgScalars<T> *)sc_in, (CgScalars<T> *)sc_out, pap_vals, nb, r, ap, (uint64_t)n, partials);
    SMH_HIP(hipGetLastError());
    *count_out = rb;
    return SMH_OK;
}

int cg_par_update(int dtype, const void *sc_in, void *sc_out, const void *pap_vals, uint32_t nb, void *r, const void *ap, size_t n, void *partials,
                  uint32_t *count_out, hipStream_t s) {
    if (dtype == SMH_F64) return cg_par_update_t<double>(sc_in, sc_out, (const double *)pap_vals, nb, (double *)r, (const double *)ap, n, (double *)partials, count_out, s);
    return cg_par_update_t<float>(sc_in, sc_out, (const float *)pap_vals, nb, (float *)r, (const float *)ap, n, (float *)partials, count_out, s);
}

// r.r = fold(rr_vals[0..nb)); stop test; beta (:51-56) -- then x += round(p*alpha) (:47) and p = round(p*beta) + r (:58-59) on this
// block's rows.  Scalars: read from sc_in, written to sc_out.
template <typename T>
static int cg_par_p_t(const void *sc_in, void *sc_out, const T *rr_vals, uint32_t nb, T *p, const T *r, T *x, size_t n, hipStream_t s) {
    uint64_t pb = (n / CgVec<T>::N + kBlock) / kBlock;
    if (pb > 512) pb = 512;
    if (cg_aligned16(p) && cg_aligned16(r) && cg_aligned16(x))
        hipLaunchKernelGGL((k_cg_par_p<T, true>), dim3((unsigned)pb), dim3(kBlock), 0, s, (const CgScalars<T> *)sc_in, (CgScalars<T> *)sc_out, rr_vals, nb, p, r, x, (uint64_t)n);
    else
        hipLaunchKernelGGL((k_cg_par_p<T, false>), dim3((unsigned)pb), dim3(kBlock), 0, s, (const CgScalars<T> *)sc_in, (CgScalars<T> *)sc_out, rr_vals, nb, p, r, x, (uint64_t)n);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

int cg_par_p(int dtype, const void *sc_in, void *sc_out, const void *rr_vals, uint32_t nb, void *p, const void *r, void *x, size_t n, hipStream_t s) {
    if (dtype == SMH_F64) return cg_par_p_t<double>(sc_in, sc_out, (const double *)rr_vals, nb, (double *)p, (const double *)r, (double *)x, n, s);
    return cg_par_p_t<float>(sc_in, sc_out, (const float *)rr_vals, nb, (float *)p, (const float *)r, (float *)x, n, s);
}

}  // namespace smh
