// spmv_stream_pipe.hip -- K1s-p: the CSR-stream kernel for stencil-like matrices on PERSISTENT blocks with a software pipeline
// over tiles (gfx950).  AN EXPERIMENT THAT LOST, kept behind SMH_STREAM_PIPE=1 ("K1s-p").
//
// Same product, same arithmetic, same order as K1s (spmv_stream.hip) -- the rounded products of a 256-row tile parked in
// LDS, thread r folding row r sequentially in storage order: bit-exact against the reference loop
// (sparsematrix.rs:146-158) -- for the case K1s serves with 16-bit column codes and byte row lengths (every tile's columns
// in <= 4 intervals, no row above 255 entries) and at most 2045 entries per tile (5-, 7-point stencils).
// The idea: a K1s block walks tile start -> chunk loads -> gathers -> LDS -> fold with three dependent trips to memory and
// nothing of its own in flight meanwhile.  Here a block owns a contiguous run of tiles and keeps five of them in flight,
// each step of the loop doing, in program order (vmcnt retires in order, so the compiler's counted waits leave the younger
// loads in flight):
//     gathers of tile t+2   (its chunks were requested three steps ago)
//     products of tile t    (its gathers were requested two steps ago) -> LDS, barrier, fold, store
//     chunk loads of tile t+5 into the registers tile t just freed
// All loads are unconditional (addresses clamped, entries masked afterwards) so the loop body is straight-line code.  The
// LDS stage is double-buffered: one barrier per tile.  With the p.Ap / inner_prod epilogue every thread accumulates
// lhs[row] * y[row] over its tiles and the block leaves ONE partial sum (fixed order: bitwise reproducible).
// Measured on the 512^3 Laplacian: 1.86 ms with three sets (97 VGPRs), 2.01 ms with these five (132 VGPRs), K1s 1.50-1.55 ms;
// the time per tile and CU does not change with the number of resident blocks -- the texture addresser, busy 70 % under K1s
// already, is the bound, and it has the same eight gather instructions per thread and tile to process here.
#include "internal.hpp"

namespace smh {

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int kPipeChunks = 2;                          // 16-byte chunks per thread and tile
constexpr uint32_t kPipeCap = 4u * kBlock * kPipeChunks;  // 2048 entries per tile at most
constexpr uint32_t kPipeWin = 64;                       // tiles whose tables a block keeps in LDS

__device__ __forceinline__ uint32_t pp_skew(uint32_t i) { return i + (i >> 5); }
__device__ __forceinline__ float pp_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ double pp_mul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ float pp_add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double pp_add(double a, double b) { return __dadd_rn(a, b); }

template <typename T>
struct PipeTile {
    uint32_t k0, k1;              // the tile's entries [k0, k1)            (block-uniform)
    uint32_t rel;                 // the tile, counted from the block's first one
    uint32_t len;                 // this thread's row length (byte table)
    uint32_t cw[kPipeChunks][2];  // packed 16-bit column codes of its chunks
    T v[kPipeChunks][4];          // values of its chunks
    T xv[kPipeChunks][4];         // x at their columns (requested one step before the tile is consumed)
    T dl;                         // DOT: lhs[row]
};

// requests everything tile t needs from memory; nothing waits here.  t is clamped by the caller (a clamped tile is loaded
// again and never consumed), so every load is unconditional.
template <typename T, bool DOT>
__device__ __forceinline__ void pipe_load(PipeTile<T> &p, uint64_t t, uint32_t rel, uint32_t slot, const uint32_t *s_tb, uint64_t n_rows,
                                          const uint8_t *__restrict__ len8, const uint16_t *__restrict__ code, const T *__restrict__ val,
                                          const T *__restrict__ dot_lhs, uint64_t last_chunk, uint32_t tid) {
    p.k0 = s_tb[slot];
    p.k1 = s_tb[slot + 1];
    p.rel = rel;
    const uint64_t r = t * kStreamRows + tid;
    p.len = len8[r];  // (padded to whole tiles)
    if constexpr (DOT) p.dl = dot_lhs[r < n_rows ? r : n_rows - 1];
    const uint64_t pa = (uint64_t)(p.k0 & ~3u);
#pragma unroll
    for (int it = 0; it < kPipeChunks; ++it) {
        uint64_t q = pa + 4u * tid + (uint64_t)it * (4u * kBlock);
        q = q < last_chunk ? q : last_chunk;  // beyond the arrays: re-read the last whole chunk (masked later)
        const u32x2 c = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(code + q));
        p.cw[it][0] = c.x; p.cw[it][1] = c.y;
        if constexpr (sizeof(T) == 4) {
            const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(val + q));
            p.v[it][0] = a.x; p.v[it][1] = a.y; p.v[it][2] = a.z; p.v[it][3] = a.w;
        } else {
            const f64x2 a = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + q));
            const f64x2 b = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + q + 2));
            p.v[it][0] = a.x; p.v[it][1] = a.y; p.v[it][2] = b.x; p.v[it][3] = b.y;
        }
    }
}

// decodes the tile's columns and requests x for them (clamped to column 0 outside the tile's entries)
template <typename T>
__device__ __forceinline__ void pipe_gather(PipeTile<T> &p, const T *__restrict__ x, const uint32_t *s_cw, uint32_t wrel, uint32_t tid) {
    const uint32_t lo = p.k0 & 3u, hi = p.k1 - (p.k0 & ~3u);
    const uint32_t *w = s_cw + 4u * (p.rel - wrel);
    const uint32_t cb0 = w[0], cb1 = w[1], cb2 = w[2], cb3 = w[3];
#pragma unroll
    for (int it = 0; it < kPipeChunks; ++it) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t i = 4u * tid + (uint32_t)it * (4u * kBlock) + e;
            const uint32_t cd = (e & 1) ? (p.cw[it][e >> 1] >> 16) : (p.cw[it][e >> 1] & 0xFFFFu);
            const uint32_t q = cd >> 14;
            const uint32_t c = (q == 0u ? cb0 : q == 1u ? cb1 : q == 2u ? cb2 : cb3) + (cd & 16383u);
            p.xv[it][e] = x[(i >= lo && i < hi) ? c : 0u];
        }
    }
}

template <typename T, bool DOT>
__global__ void __launch_bounds__(kBlock)
k_spmv_stream_pipe(const T *__restrict__ val, const T *__restrict__ x, T *__restrict__ y, uint64_t n_rows, uint64_t n_tiles, uint64_t last_chunk,
                   const uint16_t *__restrict__ code, const uint32_t *__restrict__ cwin, const uint8_t *__restrict__ len8,
                   const uint32_t *__restrict__ tbase, const T *__restrict__ dot_lhs, T *__restrict__ dot_partials) {
    __shared__ T s_prod[2][kPipeCap + kPipeCap / 32 + 8];
    __shared__ uint32_t s_wtot[2][kBlock / kWave];
    const uint32_t tid = threadIdx.x;
    // XCD-aware: blockIdx % 8 is the XCD; the blocks of an XCD own neighbouring runs of tiles
    const uint64_t per_xcd = gridDim.x >> 3;
    const uint64_t lb = (uint64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    const uint64_t nb = (uint64_t)gridDim.x;
    const uint64_t t0 = n_tiles * lb / nb, t1 = n_tiles * (lb + 1) / nb;  // this block's tiles [t0, t1)
    T dacc = T(0);
    // The per-tile tables (first entry of the tile, starts of its column intervals) reach the pipeline through an LDS window
    // of kPipeWin tiles, refilled by one coalesced load every ~60 tiles.  Fetched by scalar loads when they were needed --
    // tile start -> chunk addresses, interval starts -> gather addresses -- each step stalled on a trip to HBM (3.6 us per
    // tile, 1.85 ms per product) whatever the vector loads had in flight.
    __shared__ uint32_t s_tb[kPipeWin + 1];
    __shared__ uint32_t s_cw[kPipeWin * 4];
    uint64_t wf = t0;  // the window holds tiles [wf, wf + kPipeWin)
    auto fill = [&]() {
        for (uint32_t i = tid; i <= kPipeWin; i += kBlock) {
            const uint64_t tt = wf + i < n_tiles ? wf + i : n_tiles;
            s_tb[i] = tbase[tt];
        }
        for (uint32_t i = tid; i < kPipeWin * 4; i += kBlock) {
            const uint64_t tt = wf + i / 4 < n_tiles ? wf + i / 4 : n_tiles - 1;
            s_cw[i] = cwin[8 * tt + 2 * (i & 3u)];
        }
    };
    if (t0 < t1) {
        const uint64_t tl = t1 - 1;  // tiles beyond the run are clamped to its last one (loaded, never consumed)
        fill();
        __syncthreads();
        PipeTile<T> S0, S1, S2, S3, S4;  // tiles t .. t + 4: consumed / gathered / gather issued now / loaded / loading
#define SMH_PIPE_LOAD(SET, TT)                                                                                       \
    do {                                                                                                              \
        const uint64_t tc = (TT) < t1 ? (TT) : tl;                                                                    \
        pipe_load<T, DOT>(SET, tc, (uint32_t)(tc - t0), (uint32_t)(tc - wf), s_tb, n_rows, len8, code, val, dot_lhs, last_chunk, tid);     \
    } while (0)
        SMH_PIPE_LOAD(S0, t0);
        SMH_PIPE_LOAD(S1, t0 + 1);
        SMH_PIPE_LOAD(S2, t0 + 2);
        SMH_PIPE_LOAD(S3, t0 + 3);
        SMH_PIPE_LOAD(S4, t0 + 4);
        pipe_gather<T>(S0, x, s_cw, 0u, tid);
        pipe_gather<T>(S1, x, s_cw, 0u, tid);
        uint32_t buf = 0;
        // one step: gathers of the tile two ahead, then products / fold / store of the CURRENT one, then the chunk loads of the
        // tile five ahead into the set just freed.  Unrolled by five so that the sets keep static names.
#define SMH_PIPE_STEP(CUR, GSET, T_CUR)                                                                               \
    do {                                                                                                              \
        if ((T_CUR) + 7 > wf + kPipeWin) { /* the window runs out: refill it from the current tile on (block-uniform) */    \
            __syncthreads();                                                                                          \
            wf = (T_CUR);                                                                                             \
            fill();                                                                                                   \
            __syncthreads();                                                                                          \
        }                                                                                                             \
        pipe_gather<T>(GSET, x, s_cw, (uint32_t)(wf - t0), tid);                                                                         \
        {                                                                                                             \
            T *stage = s_prod[buf];                                                                                   \
            const uint32_t lo = CUR.k0 & 3u, hi = CUR.k1 - (CUR.k0 & ~3u);                                           \
            _Pragma("unroll") for (int it = 0; it < kPipeChunks; ++it) {                                             \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                      \
                    const uint32_t i = 4u * tid + (uint32_t)it * (4u * kBlock) + e;                                   \
                    if (i >= lo && i < hi) stage[pp_skew(i - lo)] = pp_mul(CUR.xv[it][e], CUR.v[it][e]);              \
                }                                                                                                     \
            }                                                                                                         \
            uint32_t incl = CUR.len;                                                                                  \
            _Pragma("unroll") for (int o = 1; o < kWave; o <<= 1) {                                                  \
                const uint32_t up = (uint32_t)__shfl_up((int)incl, o, kWave);                                         \
                if ((int)(tid & (kWave - 1)) >= o) incl += up;                                                        \
            }                                                                                                         \
            if ((tid & (kWave - 1)) == kWave - 1) s_wtot[buf][tid / kWave] = incl;                                    \
            __syncthreads();                                                                                          \
            uint32_t i0 = incl - CUR.len;                                                                             \
            for (uint32_t wv = 0; wv < tid / kWave; ++wv) i0 += s_wtot[buf][wv];                                      \
            const uint32_t i1 = i0 + CUR.len;                                                                         \
            T acc = T(0);                                                                                             \
            uint32_t i = i0;                                                                                          \
            for (; i + 4 <= i1; i += 4) {                                                                             \
                const T q0 = stage[pp_skew(i)], q1 = stage[pp_skew(i + 1)], q2 = stage[pp_skew(i + 2)], q3 = stage[pp_skew(i + 3)]; \
                acc = pp_add(acc, q0); acc = pp_add(acc, q1); acc = pp_add(acc, q2); acc = pp_add(acc, q3);          \
            }                                                                                                         \
            for (; i < i1; ++i) acc = pp_add(acc, stage[pp_skew(i)]);                                                 \
            const uint64_t r = (T_CUR) * kStreamRows + tid;                                                           \
            if (r < n_rows) {                                                                                         \
                if (!DOT || y) __builtin_nontemporal_store(acc, &y[r]);                                               \
                if constexpr (DOT) dacc += CUR.dl * acc;                                                              \
            }                                                                                                         \
            buf ^= 1u;                                                                                                \
        }                                                                                                             \
        SMH_PIPE_LOAD(CUR, (T_CUR) + 5);                                                                              \
    } while (0)
        for (uint64_t t = t0;;) {
            SMH_PIPE_STEP(S0, S2, t);
            if (++t >= t1) break;
            SMH_PIPE_STEP(S1, S3, t);
            if (++t >= t1) break;
            SMH_PIPE_STEP(S2, S4, t);
            if (++t >= t1) break;
            SMH_PIPE_STEP(S3, S0, t);
            if (++t >= t1) break;
            SMH_PIPE_STEP(S4, S1, t);
            if (++t >= t1) break;
        }
#undef SMH_PIPE_STEP
#undef SMH_PIPE_LOAD
    }
    if constexpr (DOT) {
        __shared__ T s_red[kBlock / kWave];
        T d = dacc;
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) d += __shfl_down(d, o, kWave);
        if ((tid & (kWave - 1)) == 0) s_red[tid / kWave] = d;
        __syncthreads();
        if (tid == 0) {
            T tsum = T(0);
#pragma unroll
            for (int w = 0; w < kBlock / kWave; ++w) tsum += s_red[w];
            dot_partials[blockIdx.x] = tsum;
        }
    }
}

uint32_t stream_pipe_cap() { return kPipeCap - 3u; }  // (a tile's first chunk may start up to 3 entries early)

// blocks of the persistent grid (= dot partials when the epilogue is on): what the chip holds at once, a multiple of 8
unsigned stream_pipe_blocks(int dtype, int device) {
    static int cache[2][64] = {};
    int &slot = cache[dtype == SMH_F64 ? 1 : 0][device & 63];
    if (slot == 0) {
        int per_cu = 0, cus = 256;
        const hipError_t e = dtype == SMH_F64
                                 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_spmv_stream_pipe<double, true>, kBlock, 0)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_spmv_stream_pipe<float, true>, kBlock, 0);
        if (e != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 2; }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        if (const char *env = getenv("SMH_STREAM_PIPE_BLOCKS_PER_CU")) {  // tuning knob
            const int v = atoi(env);
            if (v >= 1 && v <= per_cu) per_cu = v;
        }
        slot = ((per_cu * cus) + 7) & ~7;
    }
    return (unsigned)slot;
}

// Preconditions (checked by the caller, capi.hip): column codes + interval table + byte row lengths + tile starts exist, no
// tile holds more than stream_pipe_cap() entries, the arrays can be read in whole 16-byte chunks (padded, or nnz % 4 == 0).  dot_partials (optional): stream_pipe_blocks()
// values; dot_lhs NULL = x; y may be NULL with dot_partials.
int launch_spmv_stream_pipe(int dtype, const void *val, const void *x, void *y, size_t n_rows, size_t nnz, void *dot_partials,
                            const uint16_t *code, const uint32_t *cwin, const uint8_t *len8, const uint32_t *tbase, const void *dot_lhs,
                            int device, hipStream_t s) {
    if (n_rows == 0) return SMH_OK;
    if (dot_partials && !dot_lhs) dot_lhs = x;
    if (!dot_partials && !y) return fail(SMH_ERR_INVALID, "K1s-p: no output");
    const uint64_t n_tiles = (n_rows + kStreamRows - 1) / kStreamRows;
    const uint64_t readable = (nnz + 3) & ~uint64_t(3);
    const uint64_t last_chunk = readable >= 4 ? readable - 4 : 0;
    const unsigned blocks = stream_pipe_blocks(dtype, device);
#define SMH_PIPE_LAUNCH(T, D)                                                                                                   \
    hipLaunchKernelGGL((k_spmv_stream_pipe<T, D>), dim3(blocks), dim3(kBlock), 0, s, (const T *)val, (const T *)x, (T *)y, (uint64_t)n_rows, \
                       n_tiles, last_chunk, code, cwin, len8, tbase, (const T *)dot_lhs, (T *)dot_partials)
    if (dtype == SMH_F64) { if (dot_partials) SMH_PIPE_LAUNCH(double, true); else SMH_PIPE_LAUNCH(double, false); }
    else { if (dot_partials) SMH_PIPE_LAUNCH(float, true); else SMH_PIPE_LAUNCH(float, false); }
#undef SMH_PIPE_LAUNCH
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

}  // namespace smh
