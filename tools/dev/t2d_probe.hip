// Development probe (not part of the library): does a two-phase, 2-D tiled product get uniform-column matrices past the
// gather wall?  Phase 1 ("expand"): entries stored column-slice-major, the slice of x staged in LDS, prod[i] = val[i] *
// x_lds[code[i]] -- a pure streaming map, gathers hit LDS.  Phase 2 ("reduce"): one wavefront per block of R rows walks
// its tiles (slice 0, 1, ...) of prod in order and accumulates into wave-private LDS sums -- deterministic, no atomics.
// Build: hipcc -O3 --offload-arch=gfx950 tools/dev/t2d_probe.hip -o tools/dev/t2d_probe     Run: tools/dev/t2d_probe [f32|f64] [rows] [R]
#include <hip/hip_runtime.h>
#ifndef NT_RED
#define NT_RED 1
#endif
#ifndef NT_EXP
#define NT_EXP 1
#endif
#ifndef EXP_PIPE
#define EXP_PIPE 0
#endif
#ifndef EXP_DIAG
#define EXP_DIAG 0
#endif
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static inline uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

template <typename T> struct V4;
typedef float nf4 __attribute__((ext_vector_type(4)));
typedef double nd4 __attribute__((ext_vector_type(4)));
typedef uint32_t nu4 __attribute__((ext_vector_type(4)));
template <> struct V4<float> { using t = nf4; };
template <> struct V4<double> { using t = nd4; };

// phase 1: grid = slices * parts; block 1024; dynamic LDS = C * sizeof(T)
template <typename T>
__global__ __launch_bounds__(1024) void k_expand(const T *__restrict__ x, uint64_t n_cols, uint32_t C, const T *__restrict__ val,
                                                  const uint16_t *__restrict__ code, const uint64_t *__restrict__ cb_ptr, T *__restrict__ prod,
                                                  uint32_t parts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *xs = (T *)smem;
    const uint32_t cb = blockIdx.x / parts, part = blockIdx.x % parts;
    const uint64_t c0 = (uint64_t)cb * C;
    for (uint32_t i = threadIdx.x; i < C; i += 1024) xs[i] = c0 + i < n_cols ? x[c0 + i] : T(0);
    __syncthreads();
    const uint64_t a0 = cb_ptr[cb], a1 = cb_ptr[cb + 1];  // multiples of 8 entries
    const uint64_t groups = (a1 - a0) / 8;
    const uint64_t per = (groups + parts - 1) / parts;
    uint64_t g0 = (uint64_t)part * per, g1 = g0 + per < groups ? g0 + per : groups;
    using W = typename V4<T>::t;
    auto mul = [&](const nu4 cd, const W v0, const W v1, uint64_t e) {
        W p0, p1;
#if EXP_DIAG == 1  // timing only: conflict-free LDS reads instead of the gathers
        const uint32_t q = (threadIdx.x * 8u + (uint32_t)(cd.x & 7u) * 0u) & (C - 1);
        p0.x = v0.x * xs[q]; p0.y = v0.y * xs[q + 1]; p0.z = v0.z * xs[q + 2]; p0.w = v0.w * xs[q + 3];
        p1.x = v1.x * xs[q + 4]; p1.y = v1.y * xs[q + 5]; p1.z = v1.z * xs[q + 6]; p1.w = v1.w * xs[q + 7 + (cd.w & 0u)];
        *(W *)(prod + e) = p0;
        *(W *)(prod + e + 4) = p1;
        return;
#endif
        p0.x = v0.x * xs[cd.x & 0xFFFF]; p0.y = v0.y * xs[cd.x >> 16];
        p0.z = v0.z * xs[cd.y & 0xFFFF]; p0.w = v0.w * xs[cd.y >> 16];
        p1.x = v1.x * xs[cd.z & 0xFFFF]; p1.y = v1.y * xs[cd.z >> 16];
        p1.z = v1.z * xs[cd.w & 0xFFFF]; p1.w = v1.w * xs[cd.w >> 16];
#if EXP_DIAG == 2  // timing only: no stores (one conditional store keeps the products alive)
        if (p0.x + p0.y + p0.z + p0.w + p1.x + p1.y + p1.z + p1.w == (T)12345.678) *(W *)(prod + e) = p0;
        return;
#endif
#if NT_EXP
        __builtin_nontemporal_store(p0, (W *)(prod + e));
        __builtin_nontemporal_store(p1, (W *)(prod + e + 4));
#else
        *(W *)(prod + e) = p0;
        *(W *)(prod + e + 4) = p1;
#endif
    };
#if EXP_PIPE
    // software pipelined: the loads of the next two groups are issued before the current two are multiplied and stored
    struct G2 { nu4 c0, c1; W a0, a1, b0, b1; };
    auto ld = [&](uint64_t g, G2 &o) {
        const uint64_t e = a0 + g * 8;
        const bool two = g + 1024 < g1;
        const uint64_t f = two ? e + 8192 : e;
        o.c0 = __builtin_nontemporal_load((const nu4 *)(code + e));
        o.a0 = __builtin_nontemporal_load((const W *)(val + e));
        o.a1 = __builtin_nontemporal_load((const W *)(val + e + 4));
        o.c1 = __builtin_nontemporal_load((const nu4 *)(code + f));
        o.b0 = __builtin_nontemporal_load((const W *)(val + f));
        o.b1 = __builtin_nontemporal_load((const W *)(val + f + 4));
    };
    uint64_t g = g0 + threadIdx.x;
    if (g < g1) {
        G2 cur, nxt;
        ld(g, cur);
        for (;;) {
            const uint64_t gn = g + 2048;
            const bool more = gn < g1;
            if (more) ld(gn, nxt);
            const uint64_t e = a0 + g * 8;
            mul(cur.c0, cur.a0, cur.a1, e);
            if (g + 1024 < g1) mul(cur.c1, cur.b0, cur.b1, e + 8192);
            if (!more) break;
            cur = nxt;
            g = gn;
        }
    }
}
#else
    uint64_t g = g0 + threadIdx.x;
    for (; g + 1024 < g1; g += 2048) {
        const uint64_t e = a0 + g * 8, f = e + 8192;
        const nu4 cd = __builtin_nontemporal_load((const nu4 *)(code + e));
        const W v0 = __builtin_nontemporal_load((const W *)(val + e));
        const W v1 = __builtin_nontemporal_load((const W *)(val + e + 4));
        const nu4 ce = __builtin_nontemporal_load((const nu4 *)(code + f));
        const W w0 = __builtin_nontemporal_load((const W *)(val + f));
        const W w1 = __builtin_nontemporal_load((const W *)(val + f + 4));
        mul(cd, v0, v1, e);
        mul(ce, w0, w1, f);
    }
    if (g < g1) {
        const uint64_t e = a0 + g * 8;
        const nu4 cd = __builtin_nontemporal_load((const nu4 *)(code + e));
        const W v0 = __builtin_nontemporal_load((const W *)(val + e));
        const W v1 = __builtin_nontemporal_load((const W *)(val + e + 4));
        mul(cd, v0, v1, e);
    }
}
#endif

// phase 2: one wavefront per block of R rows; tstart[(rb) * n_cb + cb] = first entry of tile (cb, rb), relative to cb_ptr[cb];
// row n_rb of the table holds the ends of the last row block's tiles.  block = NW wavefronts, LDS = NW * R * sizeof(T).
// Software pipelined: the loads of batch b+1 (kD tiles, one entry per lane each) are in flight while batch b is folded.
#ifndef KD
#define KD 8
#endif
#ifndef NW
#define NW 4
#endif
constexpr int kD = KD;
__device__ inline float shl1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xF, 0xF, false)); }
__device__ inline double shl1(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x130, 0xF, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x130, 0xF, 0xF, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

template <typename T>
struct Batch {
    T pv[kD];
    uint32_t rv[kD], ln[kD];
    uint64_t bs[kD];
};

template <typename T>
__global__ __launch_bounds__(NW * 64) void k_reduce(const T *__restrict__ prod, const uint16_t *__restrict__ rowc, const uint64_t *__restrict__ cb_ptr,
                                                 const uint32_t *__restrict__ tstart, uint32_t n_cb, uint32_t n_rb, uint32_t R, T *__restrict__ y,
                                                 uint64_t n_rows, unsigned long long *dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t rb = blockIdx.x * NW + w;
    if (rb >= n_rb) return;
    T *acc = (T *)smem + (size_t)w * R;
    for (uint32_t i = lane; i < R; i += 64) acc[i] = T(0);
    const uint32_t *ts0 = tstart + (size_t)rb * n_cb, *ts1 = ts0 + n_cb;
    auto table = [&](uint32_t g, uint64_t &base, uint32_t &len) {  // lane l: tile g + l
        const uint32_t cbl = g + lane;
        base = 0;
        len = 0;
        if (cbl < n_cb) {
            const uint32_t s = ts0[cbl], e = ts1[cbl];
            base = cb_ptr[cbl] + s;
            len = e - s;
        }
    };
    uint64_t cur_base, nxt_base;
    uint32_t cur_len, nxt_len, win = 0;  // cur_*: tiles [win, win + 64), nxt_*: the 64 after them
    table(0, cur_base, cur_len);
    table(64, nxt_base, nxt_len);
    auto issue = [&](Batch<T> &B, uint32_t j0) {
        if (j0 >= win + 64) {
            cur_base = nxt_base;
            cur_len = nxt_len;
            win += 64;
            table(win + 64, nxt_base, nxt_len);
        }
#pragma unroll
        for (int d = 0; d < kD; ++d) {
            const int j = (int)((j0 + d) & 63);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cur_base, j);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cur_base >> 32), j);
            B.bs[d] = (uint64_t)hi << 32 | lo;
            B.ln[d] = (uint32_t)__builtin_amdgcn_readlane((int)cur_len, j);  // 0 past the last slice
        }
#pragma unroll
        for (int d = 0; d < kD; ++d) {  // unconditional loads (a tile's first entry is always a valid address), masked when folded
            const uint32_t idx = lane < B.ln[d] ? lane : 0;
#if NT_RED
            B.pv[d] = __builtin_nontemporal_load(prod + B.bs[d] + idx);
#else
            B.pv[d] = prod[B.bs[d] + idx];
#endif
            B.rv[d] = rowc[B.bs[d] + idx];
        }
    };
    auto fold = [&](Batch<T> &B) {
#pragma unroll
        for (int d = 0; d < kD; ++d) {
            const uint32_t len = B.ln[d];
            const uint64_t bs = B.bs[d];
            uint32_t t = lane;
            T p = lane < len ? B.pv[d] : T(0);
            uint32_t r = lane < len ? B.rv[d] : 0xFFFFFFFFu;
            for (;;) {
                uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)r, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
                if (lane == 0) prev = t == 0 ? 0xFFFFFFFFu : rowc[bs + t - 1];
                const bool valid = t < len;
                const bool head = valid && prev != r;
                // run lengths from two ballots: m = heads whose run is longer than k; the k-th neighbour's product arrives by k one-lane shifts
                const uint64_t nh = __ballot(valid && !head);
                uint64_t m = __ballot(head);
                T s = p, q = p;
                for (uint32_t k = 1;; ++k) {
                    m &= nh >> k;
                    if (!m) break;
                    q = shl1(q);
                    if ((m >> lane) & 1) s += q;
                }
                const uint32_t r63 = (uint32_t)__builtin_amdgcn_readlane((int)r, 63);
                if (head) {
                    const uint32_t nx = t - lane + 64;  // the run may go on past this pass (tiles of more than 64 entries only)
                    if (r63 == r && nx < len)
                        for (uint32_t k = nx; k < len && rowc[bs + k] == r; ++k) s += prod[bs + k];
                    acc[r] += s;
                }
                t += 64;
                if (__builtin_amdgcn_readfirstlane(t - lane) >= len) break;
                p = T(0);
                r = 0xFFFFFFFFu;
                if (t < len) { p = prod[bs + t]; r = rowc[bs + t]; }
            }
        }
    };
    Batch<T> A, B;
    issue(A, 0);
    for (uint32_t j0 = 0; j0 < n_cb; j0 += 2 * kD) {
        issue(B, j0 + kD);
        fold(A);
        issue(A, j0 + 2 * kD);
        fold(B);
    }
    const uint64_t r0 = (uint64_t)rb * R;
    for (uint32_t i = lane; i < R; i += 64)
        if (r0 + i < n_rows) y[r0 + i] = acc[i];
}

template <typename T>
static int run(uint64_t n, uint32_t R_arg, bool longrows) {
    const uint32_t C = getenv("T2D_C") ? (uint32_t)atoi(getenv("T2D_C")) : (sizeof(T) == 4 ? 32768 : 16384);  // 128 KB of x per slice
    const uint32_t n_cb = (uint32_t)((n + C - 1) / C);
    // matrix: 32 entries per row, columns uniform; optionally every 997th row 2048 entries
    std::vector<uint64_t> off(n + 1);
    off[0] = 0;
    for (uint64_t i = 0; i < n; ++i) off[i + 1] = off[i] + ((longrows && i % 997 == 0) ? 2048 : (longrows ? 30 : 32));
    const uint64_t nnz = off[n];
    printf("rows %llu nnz %llu slices %u (C=%u)\n", (unsigned long long)n, (unsigned long long)nnz, n_cb, C);
    std::vector<uint32_t> col(nnz);
    std::vector<T> val(nnz), x(n);
    for (uint64_t i = 0; i < n; ++i) x[i] = (T)((double)(mix(i ^ 0x5EED0002) >> 11) / 9007199254740992.0 * 2 - 1);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i)
        for (uint64_t k = off[i]; k < off[i + 1]; ++k) {
            const uint64_t h = mix((uint64_t)i * 0x9E3779B97F4A7C15ull ^ (k - off[i]));
            col[k] = (uint32_t)(h % n);
            val[k] = (T)((double)(mix(h) >> 11) / 9007199254740992.0 * 2 - 1);
        }
    // slice-major order (stable counting sort by slice), segments padded to 8 entries
    std::vector<uint64_t> cnt(n_cb + 1, 0), cb_ptr(n_cb + 1);
    for (uint64_t k = 0; k < nnz; ++k) cnt[col[k] / C]++;
    cb_ptr[0] = 0;
    for (uint32_t b = 0; b < n_cb; ++b) cb_ptr[b + 1] = cb_ptr[b] + ((cnt[b] + 7) & ~7ull);
    const uint64_t tot = cb_ptr[n_cb];
    std::vector<T> valA(tot, T(0));
    std::vector<uint16_t> codeA(tot, 0), rowA(tot, 0);
    uint32_t R = R_arg;
    if (!R) {
        double avg_tile_per_row = (double)nnz / n / n_cb;
        R = 64;
        while (R * avg_tile_per_row < 40 && R < 8192 / sizeof(T) * 1) R *= 2;
        if (getenv("T2D_MEAN")) R = (uint32_t)(atof(getenv("T2D_MEAN")) / avg_tile_per_row);
    }
    const uint32_t n_rb = (uint32_t)((n + R - 1) / R);
    printf("R %u row blocks %u mean tile %.1f entries, tile table %.1f MB\n", R, n_rb, (double)nnz / n_cb / n_rb, (double)(n_rb + 1) * n_cb * 4 / 1e6);
    std::vector<uint32_t> tstart((size_t)(n_rb + 1) * n_cb, 0);
    {
        std::vector<uint64_t> pos(cb_ptr.begin(), cb_ptr.end() - 1);
        uint32_t cur_rb = 0;
        for (uint32_t b = 0; b < n_cb; ++b) tstart[b] = 0;
        for (uint64_t i = 0; i < n; ++i) {
            const uint32_t rb = (uint32_t)(i / R);
            if (rb != cur_rb) {
                for (uint32_t q = cur_rb + 1; q <= rb; ++q)
                    for (uint32_t b = 0; b < n_cb; ++b) tstart[(size_t)q * n_cb + b] = (uint32_t)(pos[b] - cb_ptr[b]);
                cur_rb = rb;
            }
            for (uint64_t k = off[i]; k < off[i + 1]; ++k) {
                const uint32_t b = col[k] / C;
                const uint64_t p = pos[b]++;
                valA[p] = val[k];
                codeA[p] = (uint16_t)(col[k] - b * C);
                rowA[p] = (uint16_t)(i - (uint64_t)rb * R);
            }
        }
        for (uint32_t q = cur_rb + 1; q <= n_rb; ++q)
            for (uint32_t b = 0; b < n_cb; ++b) tstart[(size_t)q * n_cb + b] = (uint32_t)(pos[b] - cb_ptr[b]);
    }
    // reference in double, CSR order
    std::vector<double> yref(n), ymag(n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        double s = 0, m = 0;
        for (uint64_t k = off[i]; k < off[i + 1]; ++k) { s += (double)val[k] * (double)x[col[k]]; m += std::fabs((double)val[k] * (double)x[col[k]]); }
        yref[i] = s; ymag[i] = m;
    }
    T *dx, *dval, *dprod, *dy;
    uint16_t *dcode, *drow;
    uint64_t *dcb;
    uint32_t *dts;
    CK(hipMalloc(&dx, n * sizeof(T))); CK(hipMalloc(&dval, tot * sizeof(T))); CK(hipMalloc(&dprod, tot * sizeof(T))); CK(hipMalloc(&dy, n * sizeof(T)));
    CK(hipMalloc(&dcode, tot * 2)); CK(hipMalloc(&drow, tot * 2)); CK(hipMalloc(&dcb, (n_cb + 1) * 8)); CK(hipMalloc(&dts, tstart.size() * 4));
    CK(hipMemcpy(dx, x.data(), n * sizeof(T), hipMemcpyHostToDevice));
    CK(hipMemcpy(dval, valA.data(), tot * sizeof(T), hipMemcpyHostToDevice));
    CK(hipMemcpy(dcode, codeA.data(), tot * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(drow, rowA.data(), tot * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcb, cb_ptr.data(), (n_cb + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dts, tstart.data(), tstart.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dprod, 0, tot * sizeof(T)));
    unsigned long long *ddbg;
    CK(hipMalloc(&ddbg, 4096));
    CK(hipMemset(ddbg, 0, 4096));
    CK(hipFuncSetAttribute((const void *)k_expand<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(C * sizeof(T))));
    const uint32_t parts = getenv("T2D_PARTS") ? (uint32_t)atoi(getenv("T2D_PARTS")) : std::max<uint32_t>(1, (uint32_t)(4096 / n_cb));
    printf("expand: %u parts per slice\n", parts);
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    std::vector<float> t1s, t2s;
    for (int it = 0; it < 12; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_expand<T>, dim3(n_cb * parts), dim3(1024), C * sizeof(T), 0, dx, n, C, dval, dcode, dcb, dprod, parts);
        CK(hipEventRecord(e1, 0));
        hipLaunchKernelGGL(k_reduce<T>, dim3((n_rb + NW - 1) / NW), dim3(NW * 64), NW * R * sizeof(T), 0, dprod, drow, dcb, dts, n_cb, n_rb, R, dy, n, ddbg);
        CK(hipEventRecord(e2, 0));
        CK(hipEventSynchronize(e2));
        CK(hipGetLastError());
        float a, b;
        CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
        if (it >= 2) { t1s.push_back(a); t2s.push_back(b); }
    }
    std::sort(t1s.begin(), t1s.end()); std::sort(t2s.begin(), t2s.end());
    std::vector<T> y(n);
    { unsigned long long h[48]; CK(hipMemcpy(h, ddbg, sizeof h, hipMemcpyDeviceToHost)); for (int q = 0; q < 6 && q * 4096 + 17 < (int)n_rb; ++q) printf("  wave of row block %d: table %llu loads %llu consume %llu total %llu start %llu (s_memtime ticks)\n", q * 4096 + 17, h[q*6], h[q*6+1], h[q*6+2], h[q*6+3], h[q*6+4]); }
    CK(hipMemcpy(y.data(), dy, n * sizeof(T), hipMemcpyDeviceToHost));
    double worst = 0;
    const double eps = sizeof(T) == 4 ? 1e-5 : 1e-12;
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const double d = std::fabs((double)y[i] - yref[i]);
        const double rel = ymag[i] > 0 ? d / ymag[i] : d;
        if (rel > worst) worst = rel;
        if (d > eps * ymag[i] + 1e-300) ++bad;
    }
    const double b1 = (double)tot * (2 * sizeof(T) + 2), b2 = (double)tot * (sizeof(T) + 2) + (double)n * sizeof(T);
    printf("expand  median %.3f ms min %.3f (%.0f GB/s of its own bytes)\n", t1s[t1s.size() / 2], t1s[0], b1 / t1s[t1s.size() / 2] / 1e6);
    printf("reduce  median %.3f ms min %.3f (%.0f GB/s of its own bytes)\n", t2s[t2s.size() / 2], t2s[0], b2 / t2s[t2s.size() / 2] / 1e6);
    printf("total   median %.3f ms; worst |dy| / sum|a x| = %.3e, rows over the bound: %llu\n", t1s[t1s.size() / 2] + t2s[t2s.size() / 2], worst,
           (unsigned long long)bad);
    return bad ? 1 : 0;
}

int main(int argc, char **argv) {
    const bool f64 = argc > 1 && !strcmp(argv[1], "f64");
    const uint64_t n = argc > 2 ? strtoull(argv[2], nullptr, 10) : 10000000ull;
    const uint32_t R = argc > 3 ? (uint32_t)atoi(argv[3]) : 0;
    const bool longrows = argc > 4 && atoi(argv[4]);
    return f64 ? run<double>(n, R, longrows) : run<float>(n, R, longrows);
}
