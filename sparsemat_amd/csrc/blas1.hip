// blas1.hip -- K3/K4: DenseVec element-wise ops and deterministic reductions for gfx950.
//
// Element-wise (reference densevec.rs:51-73 and the composite updates of linearsolver.rs:47,49,
// 58-59): HBM-bound streams, 16 B per lane per access, grid capped at 256 CUs x 8 blocks with a
// grid-stride loop.  Multiply and add are issued as SEPARATE roundings (__fmul_rn/__fadd_rn, no
// FMA contraction) so that, given the same scalar, every element is bit-identical to the
// reference's `x * a` then `+=`.
//
// Reductions (vector.rs:50-58): two stages, fixed tree, no float atomics -> bitwise reproducible.
// Stage 1: <= kReducePartials blocks, per-thread strided partials in T, 64-lane __shfl_down
// butterfly, LDS across the 4 waves, one partial per block.  Stage 2: one block folds the
// partials in index order.  (The reference folds left to right in T; a tree in T is at least as
// accurate, and parity is a tolerance on |sum| scaled by sum|x_i y_i| -- see tests.)
#include "internal.hpp"

namespace smh {

template <typename T> struct VecOf;
template <> struct VecOf<float> { typedef float type __attribute__((ext_vector_type(4))); static constexpr int N = 4; };
template <> struct VecOf<double> { typedef double type __attribute__((ext_vector_type(2))); static constexpr int N = 2; };

__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ double sub_rn(double a, double b) { return __dsub_rn(a, b); }

template <Ew OP, typename T>
__device__ __forceinline__ T ew_apply(T x, T y, T a) {
    if constexpr (OP == Ew::Add) return add_rn(x, y);
    else if constexpr (OP == Ew::Sub) return sub_rn(x, y);
    else if constexpr (OP == Ew::Scale) return mul_rn(x, a);
    else if constexpr (OP == Ew::Axpy) return add_rn(x, mul_rn(y, a));   // x += round(y*a)
    else if constexpr (OP == Ew::Xpby) return add_rn(mul_rn(x, a), y);   // x = round(x*a) + y
    else return sub_rn(y, x);                                            // RSubInto: x = y - x
}

template <Ew OP, typename T, bool VEC>
__global__ void __launch_bounds__(kBlock)
k_ew(T *__restrict__ x, const T *__restrict__ y, uint64_t n, T a, const T *__restrict__ a_dev) {
    if (a_dev) a = *a_dev;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    if constexpr (VEC) {
        typedef typename VecOf<T>::type V;
        constexpr int N = VecOf<T>::N;
        const uint64_t nv = n / N;
        V *xv = reinterpret_cast<V *>(x);
        const V *yv = reinterpret_cast<const V *>(y);
        // non-temporal loads and stores: the vectors are far larger than the caches and nothing is read twice -- 0.271 instead
        // of 0.307 ms for x += y on 2^27 f32 (5.9 instead of 5.2 TB/s, 94 % of the measured copy ceiling)
        for (uint64_t i = tid; i < nv; i += nthreads) {
            V xx = __builtin_nontemporal_load(xv + i);
            V yy = OP == Ew::Scale ? xx : __builtin_nontemporal_load(yv + i);
#pragma unroll
            for (int e = 0; e < N; ++e) xx[e] = ew_apply<OP, T>(xx[e], yy[e], a);
            __builtin_nontemporal_store(xx, xv + i);
        }
        for (uint64_t i = nv * N + tid; i < n; i += nthreads)
            x[i] = ew_apply<OP, T>(x[i], OP == Ew::Scale ? x[i] : y[i], a);
    } else {
        for (uint64_t i = tid; i < n; i += nthreads)
            x[i] = ew_apply<OP, T>(x[i], OP == Ew::Scale ? x[i] : y[i], a);
    }
}

static inline unsigned stream_grid(uint64_t work_items) {
    uint64_t blocks = (work_items + kBlock - 1) / kBlock;
    // 256 CUs x 3 blocks, grid-stride the rest: measured on 2^27-element vectors, 768 blocks stream at 5.4-5.5 TB/s where
    // 2048 reach 4.8-5.1 and 256 fall far behind (profiles/r01_blas1_bench.log); whole multiples of the CU count only
    static const uint64_t cap = getenv("SMH_EW_BLOCKS") ? (uint64_t)atoll(getenv("SMH_EW_BLOCKS")) : 768;  // tuning knob
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    return (unsigned)blocks;
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <Ew OP, typename T>
static int launch_ew_t(T *x, const T *y, size_t n, double a, const T *a_dev, hipStream_t s) {
    if (n == 0) return SMH_OK;
    const bool vec = aligned16(x) && (OP == Ew::Scale || aligned16(y));
    if (vec)
        hipLaunchKernelGGL((k_ew<OP, T, true>), dim3(stream_grid(n / VecOf<T>::N + 1)), dim3(kBlock), 0, s, x, y,
                           (uint64_t)n, (T)a, a_dev);
    else
        hipLaunchKernelGGL((k_ew<OP, T, false>), dim3(stream_grid(n)), dim3(kBlock), 0, s, x, y, (uint64_t)n, (T)a,
                           a_dev);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

template <typename T>
static int launch_ew_op(Ew op, T *x, const T *y, size_t n, double a, const T *a_dev, hipStream_t s) {
    switch (op) {
        case Ew::Add: return launch_ew_t<Ew::Add, T>(x, y, n, a, a_dev, s);
        case Ew::Sub: return launch_ew_t<Ew::Sub, T>(x, y, n, a, a_dev, s);
        case Ew::Scale: return launch_ew_t<Ew::Scale, T>(x, y, n, a, a_dev, s);
        case Ew::Axpy: return launch_ew_t<Ew::Axpy, T>(x, y, n, a, a_dev, s);
        case Ew::Xpby: return launch_ew_t<Ew::Xpby, T>(x, y, n, a, a_dev, s);
        case Ew::RSubInto: return launch_ew_t<Ew::RSubInto, T>(x, y, n, a, a_dev, s);
    }
    return fail(SMH_ERR_INVALID, "unknown element-wise op");
}

int launch_ew(int dtype, Ew op, void *x, const void *y, size_t n, double a, const void *a_dev, hipStream_t s) {
    if (dtype == SMH_F64) return launch_ew_op<double>(op, (double *)x, (const double *)y, n, a, (const double *)a_dev, s);
    return launch_ew_op<float>(op, (float *)x, (const float *)y, n, a, (const float *)a_dev, s);
}

int launch_scale_values(int dtype, void *v, size_t n, double a, hipStream_t s) {
    return launch_ew(dtype, Ew::Scale, v, nullptr, n, a, nullptr, s);
}

// ---- reductions ------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T block_reduce_sum(T v, T *s_w) {
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
    return r;  // valid in thread 0
}

// stage 1: partials[b] = sum over this block's strided share of x[i]*y[i]
template <typename T, bool VEC>
__global__ void __launch_bounds__(kBlock)
k_dot_stage1(const T *__restrict__ x, const T *__restrict__ y, uint64_t n, T *__restrict__ partials) {
    __shared__ T s_w[kBlock / kWave];
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    T acc = T(0);
    if constexpr (VEC) {
        typedef typename VecOf<T>::type V;
        constexpr int N = VecOf<T>::N;
        const uint64_t nv = n / N;
        const V *xv = reinterpret_cast<const V *>(x);
        const V *yv = reinterpret_cast<const V *>(y);
        for (uint64_t i = tid; i < nv; i += nthreads) {
            const V xx = __builtin_nontemporal_load(xv + i);
            const V yy = xv == yv ? xx : __builtin_nontemporal_load(yv + i);  // (norm_squared: one read)
#pragma unroll
            for (int e = 0; e < N; ++e) acc += xx[e] * yy[e];
        }
        for (uint64_t i = nv * N + tid; i < n; i += nthreads) acc += x[i] * y[i];
    } else {
        for (uint64_t i = tid; i < n; i += nthreads) acc += x[i] * y[i];
    }
    const T r = block_reduce_sum<T>(acc, s_w);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

// stage 2: one block folds `count` partials in index order
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_reduce_stage2(const T *__restrict__ partials, uint32_t count, T *__restrict__ result) {
    __shared__ T s_w[kBlock / kWave];
    T acc = T(0);
    for (uint32_t i = threadIdx.x; i < count; i += kBlock) acc += partials[i];
    const T r = block_reduce_sum<T>(acc, s_w);
    if (threadIdx.x == 0) *result = r;
}

unsigned reduce_blocks(size_t n) {
    uint64_t blocks = (n + (uint64_t)kBlock * 8 - 1) / ((uint64_t)kBlock * 8);
    if (blocks > (uint64_t)kReducePartials) blocks = kReducePartials;
    if (blocks == 0) blocks = 1;
    return (unsigned)blocks;
}

template <typename T>
static int launch_dot_t(const T *x, const T *y, size_t n, T *partials, T *result, hipStream_t s) {
    const unsigned blocks = reduce_blocks(n);
    if (aligned16(x) && aligned16(y))
        hipLaunchKernelGGL((k_dot_stage1<T, true>), dim3(blocks), dim3(kBlock), 0, s, x, y, (uint64_t)n, partials);
    else
        hipLaunchKernelGGL((k_dot_stage1<T, false>), dim3(blocks), dim3(kBlock), 0, s, x, y, (uint64_t)n, partials);
    SMH_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_reduce_stage2<T>, dim3(1), dim3(kBlock), 0, s, partials, blocks, result);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

// *result = sum of in[0..count): the same two stages as a dot (fixed grid and tree: deterministic)
template <typename T>
__global__ void __launch_bounds__(kBlock) k_sum_stage1_b(const T *__restrict__ in, uint64_t n, T *__restrict__ partials) {
    __shared__ T s_w[kBlock / kWave];
    T acc = T(0);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) acc += in[i];
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) acc += __shfl_down(acc, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) s_w[threadIdx.x / kWave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T r = T(0);
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) r += s_w[w];
        partials[blockIdx.x] = r;
    }
}

int launch_fold2(int dtype, const void *in, size_t count, void *partials, void *result_dev, hipStream_t s) {
    const unsigned blocks = reduce_blocks(count);
    if (dtype == SMH_F64) {
        hipLaunchKernelGGL(k_sum_stage1_b<double>, dim3(blocks), dim3(kBlock), 0, s, (const double *)in, (uint64_t)count, (double *)partials);
        hipLaunchKernelGGL(k_reduce_stage2<double>, dim3(1), dim3(kBlock), 0, s, (const double *)partials, blocks, (double *)result_dev);
    } else {
        hipLaunchKernelGGL(k_sum_stage1_b<float>, dim3(blocks), dim3(kBlock), 0, s, (const float *)in, (uint64_t)count, (float *)partials);
        hipLaunchKernelGGL(k_reduce_stage2<float>, dim3(1), dim3(kBlock), 0, s, (const float *)partials, blocks, (float *)result_dev);
    }
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

int launch_dot(int dtype, const void *x, const void *y, size_t n, void *partials, void *result_dev, hipStream_t s) {
    if (dtype == SMH_F64)
        return launch_dot_t<double>((const double *)x, (const double *)y, n, (double *)partials, (double *)result_dev, s);
    return launch_dot_t<float>((const float *)x, (const float *)y, n, (float *)partials, (float *)result_dev, s);
}

// ---- CRS structure statistics / validation ---------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_crs_stats(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, uint64_t n_rows, uint64_t nnz,
            CrsStats *__restrict__ st) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    uint32_t max_len = 0, max_col = 0, min_inv = 0, bad = 0;  // min_inv = ~min column (so that 0 is the identity)
    for (uint64_t r = tid; r < n_rows; r += nthreads) {
        const uint32_t a = off[r], b = off[r + 1];
        if (b < a) bad |= 1u;
        else if (b - a > max_len) max_len = b - a;
    }
    for (uint64_t k = tid; k < nnz; k += nthreads) {
        const uint32_t c = col[k];
        if (c > max_col) max_col = c;
        if (~c > min_inv) min_inv = ~c;
    }
    if (tid == 0) {
        if (off[0] != 0u) bad |= 2u;
        if ((uint64_t)off[n_rows] != nnz) bad |= 4u;
    }
    // integer max/or reductions: order independent, atomics are exact
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        max_len = max(max_len, (uint32_t)__shfl_down(max_len, o, kWave));
        max_col = max(max_col, (uint32_t)__shfl_down(max_col, o, kWave));
        min_inv = max(min_inv, (uint32_t)__shfl_down(min_inv, o, kWave));
        bad |= (uint32_t)__shfl_down(bad, o, kWave);
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        atomicMax(&st->max_row_len, max_len);
        atomicMax(&st->max_col, max_col);
        atomicMax(&st->min_col_inv, min_inv);
        if (bad) atomicOr(&st->bad, bad);
    }
}

int launch_crs_stats(const uint32_t *off, const uint32_t *col, size_t n_rows, size_t nnz, CrsStats *d_stats,
                     hipStream_t s) {
    SMH_HIP(hipMemsetAsync(d_stats, 0, sizeof(CrsStats), s));
    uint64_t work = nnz > n_rows ? nnz : n_rows;
    hipLaunchKernelGGL(k_crs_stats, dim3(stream_grid(work)), dim3(kBlock), 0, s, off, col, (uint64_t)n_rows,
                       (uint64_t)nnz, d_stats);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

}  // namespace smh
