// spmv_stream_xd.hip -- K1s "XD": the CSR-stream kernel for stencil-like matrices with everything a thread would have to compute
// per entry moved into the build.  Same product, same arithmetic and same ORDER as K1s (spmv_stream.hip; reference
// sparsematrix.rs:146-158 over sparsemat_crs.rs:102-110): products rounded, then added to the row's sum in storage order, one
// rounded add per entry, starting from +0 -- bit-exact against the reference loop.
//
// Why: K1s with x staged in LDS (XS) is bound by vector-ALU issue, not by bytes -- a wavefront instruction occupies its 16-lane
// SIMD for 4 cycles, a 256-row tile costs each of its four wavefronts ~400 of them, 2048 tiles per CU: 1.37 ms of issue time on
// the 512^3 Laplacian for 1.30 ms measured.  Per entry the XS body spent 6 instructions turning a 16-bit column code (interval,
// offset) into an LDS address, 7 on where the product goes (position relative to the tile, range test, bank skew, select), and
// the row sums walked a skewed stage in two loops with a dependent LDS round trip per step.  Here
//   * the code array holds the BYTE OFFSET of x[col] inside the tile's LDS stage of x (the stage's layout is a function of
//     the tile's interval table alone, so the build can know it): decode = extracting 16 bits;
//   * a thread's four products of a chunk leave as ONE 16-byte LDS store at the chunk's own position (the stage is not skewed
//     and covers both chunks of every thread, so slots before the tile's first / after its last entry need no test: they receive
//     products nobody reads);
//   * the chunk loads are buffer loads whose descriptors end with the tile's entries: no branches, no tail path;
//   * a row's first 8 products are read with 8 independent LDS loads and added under a lane mask (rows beyond 8 entries continue in
//     a loop); the row-length prefix sum is a DPP scan.
// Without the skew the row sums of rows of EVEN length collide in LDS (stride 8 -> 8 lanes per bank); the handle takes this form
// only when most rows have an odd length -- every stencil with a diagonal: 5, 7, 9, 27 points -- (capi.hip, stream_direct_choice).
#include "internal.hpp"

namespace smh {

namespace {

typedef uint32_t xd_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t xd_u2 __attribute__((ext_vector_type(2)));
typedef float xd_f4 __attribute__((ext_vector_type(4)));
typedef double xd_d2 __attribute__((ext_vector_type(2)));
constexpr int kXdRsrc = 0x00020000;  // dword 3 of a raw buffer descriptor on gfx9 (32-bit data format)

__device__ __forceinline__ float xd_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ double xd_mul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ float xd_add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double xd_add(double a, double b) { return __dadd_rn(a, b); }

template <int CTRL, int ROWS> __device__ __forceinline__ uint32_t xd_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xF, false);  // lanes without a source receive 0
}
// inclusive prefix sum over the wavefront
__device__ __forceinline__ uint32_t xd_wave_scan(uint32_t v) {
    v += xd_dpp<0x111, 0xF>(v);  // row_shr:1
    v += xd_dpp<0x112, 0xF>(v);  // row_shr:2
    v += xd_dpp<0x114, 0xF>(v);  // row_shr:4
    v += xd_dpp<0x118, 0xF>(v);  // row_shr:8
    v += xd_dpp<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v += xd_dpp<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

constexpr int kXdSlots = 2 * 4 * kBlock;  // product slots of a tile: two 4-entry chunks per thread (kStreamCapSmall + 3 <= 2048)
static_assert(kStreamCapSmall + 3 <= kXdSlots, "a tile's entries from an aligned start fit two chunks per thread");

// K1s XD-V: the VALUE DICTIONARY form.  A stage offset is a multiple of sizeof(T) below the stage's size, so a 16-bit code has bits
// to spare -- the low log2(sizeof T) ones and those above the stage: 5 bits with the 2048-entry stage (f32 and f64), 4 with the
// 4096-entry one.  When the matrix holds no more than 32 / 16 DISTINCT values (bit patterns) -- every constant-coefficient stencil,
// every unweighted graph Laplacian or adjacency matrix; BASELINE C4 has two, 6 and -1 -- those bits name the entry's value in a
// dictionary the kernel keeps in LDS, and the value array is not read at all: 2 bytes per entry leave HBM instead of 6 (f32) / 10
// (f64).  The product is x times the very same bit pattern as before, the adds are in storage order: still bit for bit the
// reference's result.  (capi.hip decides; k_value_dict_* below build the dictionary, exactly, or say that it does not exist.)
template <typename T, int XS> struct XdBits {
    static constexpr uint32_t kLow = sizeof(T) == 8 ? 3u : 2u;                 // free low bits
    static constexpr uint32_t kIdx = XS == 4 ? 12u : 11u;                      // bits of an entry index inside the stage
    static constexpr uint32_t kOfsMask = ((1u << (kLow + kIdx)) - 1u) & ~((1u << kLow) - 1u);
    static constexpr uint32_t kValues = 1u << (16u - kIdx);                    // 32 / 16 dictionary entries
    static constexpr uint32_t kHigh = 16u - (kLow + kIdx);                     // spare bits above the offset: 3 (f32, 2048-entry stage) or 2
    __device__ static __forceinline__ uint32_t ofs(uint32_t c) { return c & kOfsMask; }
    __device__ static __forceinline__ uint32_t vidx(uint32_t c) { return (c & ((1u << kLow) - 1u)) | ((c >> (kLow + kIdx)) << kLow); }
    // dictionaries of at most 2^kHigh values (every stencil) keep the index in the HIGH bits alone: one shift instead of three operations
    __device__ static __forceinline__ uint32_t vidx_high(uint32_t c) { return c >> (kLow + kIdx); }
};

// XS: 16-byte chunks of x per thread the tile's stage holds (2 or 4); VD: the codes carry value-dictionary indices (val is not read) --
// 1: in the low and the high spare bits (up to 32 / 16 values), 2: in the high ones alone (up to 8 / 4 values)
template <typename T, bool DOT, int XS, int VD = 0>
__global__ void __launch_bounds__(kBlock)
k_spmv_stream_xd(const T *__restrict__ val, const T *__restrict__ x, T *__restrict__ y, uint64_t n_rows, uint64_t n_tiles,
                 T *__restrict__ dot_partials, const uint16_t *__restrict__ scode, const uint32_t *__restrict__ cwin,
                 const uint8_t *__restrict__ len8, const uint32_t *__restrict__ tbase, const T *__restrict__ dot_lhs, uint64_t tile0,
                 const T *__restrict__ dict = nullptr) {
    constexpr int kXsCap = XS * kBlock * 4;  // entries of x the stage holds (2048 / 4096)
    __shared__ T s_dict[VD ? 32 : 1];
    // (the dictionary array has 32 entries whatever its fill.  Requested by EVERY thread, without a branch, and put into the LDS only with
    // the stage of x below: inside `if (tid < 32)` the compiler waits for the load on the spot -- a whole memory round trip for the first
    // wavefront before it has requested anything else, and the workgroup's barrier waits for that wavefront.)
    T dict_v = T(0);
    if constexpr (VD) dict_v = dict[threadIdx.x & 31u];
    __shared__ __attribute__((aligned(16))) T s_xs[kXsCap];
    __shared__ __attribute__((aligned(16))) T s_prod[kXdSlots + 8];  // (+8: a row's eight unconditional reads may pass the tile's end)
    __shared__ uint32_t s_wtot[kBlock / kWave];
    // bijective XCD-aware remap: XCD g (= blockIdx % 8) walks a contiguous run of the launch's tiles (as K1s)
    const uint64_t q8 = n_tiles >> 3, rm = n_tiles & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const uint64_t tile = tile0 + (xcd < rm ? xcd * (q8 + 1) : rm * (q8 + 1) + (xcd - rm) * q8) + idx;
    const uint64_t r0 = tile * (uint64_t)kStreamRows;
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1);
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid / kWave));
    const uint32_t k0 = tbase[tile], k1 = tbase[tile + 1];  // tile-uniform: scalar loads
    const uint32_t my_len = len8[r0 + tid];                 // (the byte array is padded to whole tiles with zeros)
    T dl = T(0);
    if constexpr (DOT) dl = r0 + tid < n_rows ? dot_lhs[r0 + tid] : T(0);  // requested now: its latency hides behind the tile
    // ---- the tile's intervals of x: 16-byte chunks, requested before the tile's own entries (they are needed first) ----
    const uint32_t *w = cwin + 8 * tile;  // scalar loads
    const uint32_t cb0 = w[0], e0 = w[1], cb1 = w[2], e1 = w[3], cb2 = w[4], e2 = w[5], cb3 = w[6], e3 = w[7];
    const uint32_t al0 = cb0 & ~3u, al1 = cb1 & ~3u, al2 = cb2 & ~3u, al3 = cb3 & ~3u;
    const uint32_t n0 = e0 > cb0 ? (e0 - al0 + 3u) >> 2 : 0u, n1 = e1 > cb1 ? (e1 - al1 + 3u) >> 2 : 0u;
    const uint32_t n2 = e2 > cb2 ? (e2 - al2 + 3u) >> 2 : 0u, n3 = e3 > cb3 ? (e3 - al3 + 3u) >> 2 : 0u;
    const uint32_t p1 = n0, p2 = p1 + n1, p3 = p2 + n2, xs_tot = p3 + n3;  // (<= XS * kBlock chunks, inside x: checked by the host)
    const uint32_t d0 = al0 >> 2, d1 = (al1 >> 2) - p1, d2 = (al2 >> 2) - p2, d3 = (al3 >> 2) - p3;  // (mod 2^32; j + d_q is a chunk of x)
    // f64: a thread's 32 bytes (of x, of the values, of the products) are TWO 16-byte pieces, and a wavefront's lanes take piece
    // l and piece 64 + l of its 128 -- each load / store instruction then covers 1 KiB of contiguous bytes.  With 32 contiguous
    // bytes per lane (rounds 2-3) each of the two instructions touched every line and used half of it: what kept K1r's f64 kernel
    // at 0.8 of its f32 rate did the same here.  Which thread moves which piece changes no arithmetic: the
    // products land in the same stage slots and are added in storage order as before (bit-exact).
    const uint32_t wbase = tid & ~(uint32_t)(kWave - 1);  // first thread of this wavefront
    T xr[XS][4];
    uint32_t xpiece[XS][2];  // f64: the two 16-byte pieces (2 entries each) of the stage this thread fills
#pragma unroll
    for (int u = 0; u < XS; ++u) {
        xr[u][0] = xr[u][1] = xr[u][2] = xr[u][3] = T(0);
        if constexpr (sizeof(T) == 4) {
            const uint32_t j = tid + (uint32_t)u * kBlock;
            xpiece[u][0] = xpiece[u][1] = 0;
            if (j < xs_tot) {
                // chunk j of the stage = chunk j + d_q of x, q = the interval j falls into (tile-uniform offsets: three selects)
                uint32_t d = d0;
                d = j >= p1 ? d1 : d;
                d = j >= p2 ? d2 : d;
                d = j >= p3 ? d3 : d;
                const T *g = x + 4ull * (uint64_t)(j + d);
                const xd_f4 a = *reinterpret_cast<const xd_f4 *>(g);
                xr[u][0] = a.x; xr[u][1] = a.y; xr[u][2] = a.z; xr[u][3] = a.w;
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t pi = 2u * ((uint32_t)u * kBlock + wbase) + (uint32_t)h * kWave + lane;  // piece of the stage
                const uint32_t j = pi >> 1;                                                            // its chunk
                xpiece[u][h] = pi;
                if (j < xs_tot) {
                    uint32_t d = d0;
                    d = j >= p1 ? d1 : d;
                    d = j >= p2 ? d2 : d;
                    d = j >= p3 ? d3 : d;
                    const xd_d2 a = *reinterpret_cast<const xd_d2 *>(x + 4ull * (uint64_t)(j + d) + 2u * (pi & 1u));
                    xr[u][2 * h] = a.x; xr[u][2 * h + 1] = a.y;
                }
            }
        }
    }
    // ---- the tile's entries: two chunks per thread from the aligned start `pa`; a descriptor that ends with the tile's last
    // entry makes every slot beyond it read as zero (code 0: the stage's first entry; value 0) ----
    const uint32_t pa = k0 & ~3u, hi = k1 - pa;  // hi <= kXdSlots (kStreamCapSmall, checked by the host)
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)(scode + pa), 0, (int)(((hi + 3u) & ~3u) * 2u), kXdRsrc);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(val + pa), 0, (int)(hi * (uint32_t)sizeof(T)), kXdRsrc);
    xd_u2 cw[2];
    T v[2][4];
    uint32_t epos[2][2];  // f64: the entry positions (relative to pa) of the thread's two 2-entry pieces
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        v[it][0] = v[it][1] = v[it][2] = v[it][3] = T(0);
        if constexpr (sizeof(T) == 4) {
            const uint32_t j = 4u * tid + (uint32_t)it * (4u * kBlock);
            epos[it][0] = epos[it][1] = 0;
            cw[it] = __builtin_bit_cast(xd_u2, __builtin_amdgcn_raw_buffer_load_b64(rc, (int)(j * 2u), 0, 2 /* nt */));
            if constexpr (!VD) {
                const xd_f4 a = __builtin_bit_cast(xd_f4, __builtin_amdgcn_raw_buffer_load_b128(rv, (int)(j * 4u), 0, 2));
                v[it][0] = a.x; v[it][1] = a.y; v[it][2] = a.z; v[it][3] = a.w;
            }
        } else {
            const uint32_t e0p = 4u * ((uint32_t)it * kBlock + wbase) + 2u * lane, e1p = e0p + 2u * kWave;
            epos[it][0] = e0p; epos[it][1] = e1p;
            cw[it].x = __builtin_amdgcn_raw_buffer_load_b32(rc, (int)(e0p * 2u), 0, 2 /* nt */);
            cw[it].y = __builtin_amdgcn_raw_buffer_load_b32(rc, (int)(e1p * 2u), 0, 2);
            if constexpr (!VD) {
                const xd_d2 a = __builtin_bit_cast(xd_d2, __builtin_amdgcn_raw_buffer_load_b128(rv, (int)(e0p * 8u), 0, 2));
                const xd_d2 b = __builtin_bit_cast(xd_d2, __builtin_amdgcn_raw_buffer_load_b128(rv, (int)(e1p * 8u), 0, 2));
                v[it][0] = a.x; v[it][1] = a.y; v[it][2] = b.x; v[it][3] = b.y;
            }
        }
    }
    // ---- row boundaries: prefix sum of the byte lengths (its cross-wave part rides on the barrier below) ----
    const uint32_t incl = xd_wave_scan(my_len);
    if (lane == kWave - 1) s_wtot[wave] = incl;
    // ---- x into its stage ----
#pragma unroll
    for (int u = 0; u < XS; ++u) {
        if constexpr (sizeof(T) == 4) {
            const uint32_t j = tid + (uint32_t)u * kBlock;
            if (j < xs_tot) {
                xd_f4 a; a.x = xr[u][0]; a.y = xr[u][1]; a.z = xr[u][2]; a.w = xr[u][3];
                *reinterpret_cast<xd_f4 *>(&s_xs[4u * j]) = a;
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if ((xpiece[u][h] >> 1) < xs_tot) {
                    xd_d2 a; a.x = xr[u][2 * h]; a.y = xr[u][2 * h + 1];
                    *reinterpret_cast<xd_d2 *>(&s_xs[2u * xpiece[u][h]]) = a;
                }
        }
    }
    if constexpr (VD) {
        if (tid < 32u) s_dict[tid] = dict_v;
    }
    __syncthreads();
    // ---- products: x from the stage by byte offset, four products per 16-byte store at the chunk's own position ----
    const char *xs_bytes = reinterpret_cast<const char *>(s_xs);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        T p[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t cword = (e >> 1) ? cw[it].y : cw[it].x;
            const uint32_t code = (e & 1) ? (cword >> 16) : (cword & 0xFFFFu);
            if constexpr (VD) {
                // (a slot past the tile reads code 0: stage entry 0 times dictionary entry 0 -- never added to any row)
                const uint32_t vi = VD == 2 ? XdBits<T, XS>::vidx_high(code) : XdBits<T, XS>::vidx(code);
                p[e] = xd_mul(*reinterpret_cast<const T *>(xs_bytes + XdBits<T, XS>::ofs(code)), s_dict[vi]);
            } else {
                p[e] = xd_mul(*reinterpret_cast<const T *>(xs_bytes + code), v[it][e]);
            }
        }
        if constexpr (sizeof(T) == 4) {
            const uint32_t j = 4u * tid + (uint32_t)it * (4u * kBlock);
            xd_f4 a; a.x = p[0]; a.y = p[1]; a.z = p[2]; a.w = p[3];
            *reinterpret_cast<xd_f4 *>(&s_prod[j]) = a;
        } else {
            xd_d2 a, b; a.x = p[0]; a.y = p[1]; b.x = p[2]; b.y = p[3];
            *reinterpret_cast<xd_d2 *>(&s_prod[epos[it][0]]) = a;
            *reinterpret_cast<xd_d2 *>(&s_prod[epos[it][1]]) = b;
        }
    }
    __syncthreads();
    // ---- row sums: storage order, one rounded add per entry (reference: sum += product) ----
    uint32_t before = 0;  // entries of the tile in the waves before this one (wave-uniform)
#pragma unroll
    for (int ww = 0; ww < kBlock / kWave - 1; ++ww) before += (uint32_t)ww < wave ? s_wtot[ww] : 0u;
    const T *pp = s_prod + ((k0 - pa) + before + (incl - my_len));
    T q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = pp[i];
    T acc = T(0);
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if ((uint32_t)i < my_len) acc = xd_add(acc, q[i]);
    for (uint32_t i = 8; i < my_len; ++i) acc = xd_add(acc, pp[i]);
    const uint64_t r = r0 + tid;
    if (r < n_rows && (!DOT || y)) __builtin_nontemporal_store(acc, &y[r]);  // (DOT with y == NULL: only lhs . (A x) is wanted)
    if constexpr (DOT) {  // fixed order: lanes (DPP scan network), waves (index order) -- bitwise reproducible, as K1s
        __shared__ T s_red[kBlock / kWave];
        T d = T(0);
        if (r < n_rows) d += dl * acc;  // (as K1s: 0 + the product)
        d = wave_sum_to_lane63(d);
        if (lane == kWave - 1) s_red[wave] = d;
        __syncthreads();
        if (tid == 0) {
            T t = T(0);
#pragma unroll
            for (int ww = 0; ww < kBlock / kWave; ++ww) t += s_red[ww];
            dot_partials[tile] = t;
        }
    }
}

// ---- K1s XD-V on PERSISTENT workgroups ------------------------------------------------------------------------------------------
// With the value array gone the kernel above is bound by LATENCY: a workgroup lives 2.85 us -- the two dependent memory round trips of
// its tile (tile table -> codes and x chunks) -- at full occupancy (profiles/r04_k1s_value_dictionary.log).  Here a workgroup walks a
// contiguous run of tiles and the NEXT tile's table entries, codes, x chunks and row lengths are on their way while the current tile is
// multiplied and summed: the same loads, the same LDS stages, the same products and the same order of additions per row (bit-exact),
// two barriers per tile as before; what a thread holds of a tile is the register set XdTileRegs, two of them alternate.
template <typename T, int XS>
struct XdTileRegs {
    uint32_t k0, pa, my_len, xs_tot;
    uint64_t r0, tile;
    T dl;
    T xr[XS][4];
    uint32_t xpiece[XS][2];
    xd_u2 cw[2];
    uint32_t epos[2][2];
};

template <typename T, bool DOT, int XS, int VD>
__global__ void __launch_bounds__(kBlock)
k_spmv_stream_xdp(const T *__restrict__ x, T *__restrict__ y, uint64_t n_rows, uint64_t n_tiles, T *__restrict__ dot_partials,
                  const uint16_t *__restrict__ scode, const uint32_t *__restrict__ cwin, const uint8_t *__restrict__ len8,
                  const uint32_t *__restrict__ tbase, const T *__restrict__ dot_lhs, uint64_t tile0, const T *__restrict__ dict) {
    static_assert(VD == 1 || VD == 2, "the persistent form exists for the value-dictionary kernels");
    constexpr int kXsCap = XS * kBlock * 4;
    __shared__ T s_dict[32];
    __shared__ __attribute__((aligned(16))) T s_xs[kXsCap];
    __shared__ __attribute__((aligned(16))) T s_prod[kXdSlots + 8];
    __shared__ uint32_t s_wtot[2][kBlock / kWave];  // (by tile parity: a fast wavefront stages the next tile while a slow one still sums)
    __shared__ T s_red[2][kBlock / kWave];          // (by tile parity too: folded one barrier later, see fold_dot)
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1);
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid / kWave));
    const uint32_t wbase = tid & ~(uint32_t)(kWave - 1);
    if (tid < 32u) s_dict[tid] = dict[tid];
    // XCD g (= blockIdx % 8) owns a contiguous eighth of the launch's tiles, its workgroups contiguous runs of those
    const uint64_t q8 = n_tiles >> 3, rm = n_tiles & 7, xcd = blockIdx.x & 7, wgx = gridDim.x >> 3, j_wg = blockIdx.x >> 3;
    const uint64_t xcd_first = xcd < rm ? xcd * (q8 + 1) : rm * (q8 + 1) + (xcd - rm) * q8, xcd_count = q8 + (xcd < rm ? 1u : 0u);
    const uint64_t per = (xcd_count + wgx - 1) / wgx;
    const uint64_t t_begin = xcd_first + j_wg * per;
    const uint64_t t_end = t_begin + per < xcd_first + xcd_count ? t_begin + per : xcd_first + xcd_count;
    if (t_begin >= t_end) return;  // (whole workgroup, before any barrier)

    auto load = [&](XdTileRegs<T, XS> &R, uint64_t t) {  // t: index inside the launch; everything of tile tile0 + t a thread needs, requested
        R.tile = tile0 + t;
        R.r0 = R.tile * (uint64_t)kStreamRows;
        const uint32_t k0 = tbase[R.tile], k1 = tbase[R.tile + 1];
        R.k0 = k0;
        R.my_len = len8[R.r0 + tid];
        R.dl = T(0);
        if constexpr (DOT) R.dl = R.r0 + tid < n_rows ? dot_lhs[R.r0 + tid] : T(0);
        const uint32_t *w = cwin + 8 * R.tile;
        const uint32_t cb0 = w[0], e0 = w[1], cb1 = w[2], e1 = w[3], cb2 = w[4], e2 = w[5], cb3 = w[6], e3 = w[7];
        const uint32_t al0 = cb0 & ~3u, al1 = cb1 & ~3u, al2 = cb2 & ~3u, al3 = cb3 & ~3u;
        const uint32_t n0 = e0 > cb0 ? (e0 - al0 + 3u) >> 2 : 0u, n1 = e1 > cb1 ? (e1 - al1 + 3u) >> 2 : 0u;
        const uint32_t n2 = e2 > cb2 ? (e2 - al2 + 3u) >> 2 : 0u, n3 = e3 > cb3 ? (e3 - al3 + 3u) >> 2 : 0u;
        const uint32_t p1 = n0, p2 = p1 + n1, p3 = p2 + n2;
        R.xs_tot = p3 + n3;
        const uint32_t d0 = al0 >> 2, d1 = (al1 >> 2) - p1, d2 = (al2 >> 2) - p2, d3 = (al3 >> 2) - p3;
#pragma unroll
        for (int u = 0; u < XS; ++u) {
            R.xr[u][0] = R.xr[u][1] = R.xr[u][2] = R.xr[u][3] = T(0);
            if constexpr (sizeof(T) == 4) {
                const uint32_t j = tid + (uint32_t)u * kBlock;
                R.xpiece[u][0] = R.xpiece[u][1] = 0;
                if (j < R.xs_tot) {
                    uint32_t d = d0;
                    d = j >= p1 ? d1 : d;
                    d = j >= p2 ? d2 : d;
                    d = j >= p3 ? d3 : d;
                    const xd_f4 a = *reinterpret_cast<const xd_f4 *>(x + 4ull * (uint64_t)(j + d));
                    R.xr[u][0] = a.x; R.xr[u][1] = a.y; R.xr[u][2] = a.z; R.xr[u][3] = a.w;
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t pi = 2u * ((uint32_t)u * kBlock + wbase) + (uint32_t)h * kWave + lane;
                    const uint32_t j = pi >> 1;
                    R.xpiece[u][h] = pi;
                    if (j < R.xs_tot) {
                        uint32_t d = d0;
                        d = j >= p1 ? d1 : d;
                        d = j >= p2 ? d2 : d;
                        d = j >= p3 ? d3 : d;
                        const xd_d2 a = *reinterpret_cast<const xd_d2 *>(x + 4ull * (uint64_t)(j + d) + 2u * (pi & 1u));
                        R.xr[u][2 * h] = a.x; R.xr[u][2 * h + 1] = a.y;
                    }
                }
            }
        }
        const uint32_t pa = k0 & ~3u, hi = k1 - pa;
        R.pa = pa;
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)(scode + pa), 0, (int)(((hi + 3u) & ~3u) * 2u), kXdRsrc);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            if constexpr (sizeof(T) == 4) {
                const uint32_t j = 4u * tid + (uint32_t)it * (4u * kBlock);
                R.epos[it][0] = j; R.epos[it][1] = j;
                R.cw[it] = __builtin_bit_cast(xd_u2, __builtin_amdgcn_raw_buffer_load_b64(rc, (int)(j * 2u), 0, 2 /* nt */));
            } else {
                const uint32_t e0p = 4u * ((uint32_t)it * kBlock + wbase) + 2u * lane, e1p = e0p + 2u * kWave;
                R.epos[it][0] = e0p; R.epos[it][1] = e1p;
                R.cw[it].x = __builtin_amdgcn_raw_buffer_load_b32(rc, (int)(e0p * 2u), 0, 2);
                R.cw[it].y = __builtin_amdgcn_raw_buffer_load_b32(rc, (int)(e1p * 2u), 0, 2);
            }
        }
    };
    // the wavefronts' shares of a tile's dot, folded in index order (as the per-tile kernel does) -- by thread 0 AFTER the next barrier the
    // workgroup passes anyway (the first one of the following tile; a closing one after the last): a barrier of its own per tile, with
    // one thread folding behind it while 255 wait at the next, cost the f32 solver 0.25 ms per iteration
    auto fold_dot = [&](uint32_t par, uint64_t tile) {
        if (tid == 0) {
            T t = T(0);
#pragma unroll
            for (int ww = 0; ww < kBlock / kWave; ++ww) t += s_red[par][ww];
            dot_partials[tile] = t;
        }
    };
    // one tile with its registers R, prefetching tile t_next into N between the two barriers
    auto tile_body = [&](XdTileRegs<T, XS> &R, XdTileRegs<T, XS> &N, uint64_t t_next, uint32_t par, bool first) {
        const uint32_t incl = xd_wave_scan(R.my_len);
        if (lane == kWave - 1) s_wtot[par][wave] = incl;
#pragma unroll
        for (int u = 0; u < XS; ++u) {
            if constexpr (sizeof(T) == 4) {
                const uint32_t j = tid + (uint32_t)u * kBlock;
                if (j < R.xs_tot) {
                    xd_f4 a; a.x = R.xr[u][0]; a.y = R.xr[u][1]; a.z = R.xr[u][2]; a.w = R.xr[u][3];
                    *reinterpret_cast<xd_f4 *>(&s_xs[4u * j]) = a;
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    if ((R.xpiece[u][h] >> 1) < R.xs_tot) {
                        xd_d2 a; a.x = R.xr[u][2 * h]; a.y = R.xr[u][2 * h + 1];
                        *reinterpret_cast<xd_d2 *>(&s_xs[2u * R.xpiece[u][h]]) = a;
                    }
            }
        }
        __syncthreads();
        if constexpr (DOT) {
            if (!first) fold_dot(par ^ 1u, R.tile - 1);  // (a workgroup's tiles are consecutive)
        }
        load(N, t_next);  // (always: no branch around the loads; the last tile is simply requested once more)
        const char *xs_bytes = reinterpret_cast<const char *>(s_xs);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            T p[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t cword = (e >> 1) ? R.cw[it].y : R.cw[it].x;
                const uint32_t code = (e & 1) ? (cword >> 16) : (cword & 0xFFFFu);
                const uint32_t vi = VD == 2 ? XdBits<T, XS>::vidx_high(code) : XdBits<T, XS>::vidx(code);
                p[e] = xd_mul(*reinterpret_cast<const T *>(xs_bytes + XdBits<T, XS>::ofs(code)), s_dict[vi]);
            }
            if constexpr (sizeof(T) == 4) {
                xd_f4 a; a.x = p[0]; a.y = p[1]; a.z = p[2]; a.w = p[3];
                *reinterpret_cast<xd_f4 *>(&s_prod[R.epos[it][0]]) = a;
            } else {
                xd_d2 a, b; a.x = p[0]; a.y = p[1]; b.x = p[2]; b.y = p[3];
                *reinterpret_cast<xd_d2 *>(&s_prod[R.epos[it][0]]) = a;
                *reinterpret_cast<xd_d2 *>(&s_prod[R.epos[it][1]]) = b;
            }
        }
        __syncthreads();
        uint32_t before = 0;
#pragma unroll
        for (int ww = 0; ww < kBlock / kWave - 1; ++ww) before += (uint32_t)ww < wave ? s_wtot[par][ww] : 0u;
        const T *pp = s_prod + ((R.k0 - R.pa) + before + (incl - R.my_len));
        T q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = pp[i];
        T acc = T(0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if ((uint32_t)i < R.my_len) acc = xd_add(acc, q[i]);
        for (uint32_t i = 8; i < R.my_len; ++i) acc = xd_add(acc, pp[i]);
        const uint64_t r = R.r0 + tid;
        if (r < n_rows && (!DOT || y)) __builtin_nontemporal_store(acc, &y[r]);
        if constexpr (DOT) {  // fixed order: lanes (DPP scan network), waves (index order) -- bitwise reproducible, as K1s
            T d = T(0);
            if (r < n_rows) d += R.dl * acc;
            d = wave_sum_to_lane63(d);
            if (lane == kWave - 1) s_red[par][wave] = d;
        }
    };
    XdTileRegs<T, XS> A, B;
    const uint64_t t_last = t_end - 1;
    load(A, t_begin);
    uint32_t last_par = 0;
    for (uint64_t t = t_begin;;) {
        tile_body(A, B, t + 1 < t_end ? t + 1 : t_last, 0u, t == t_begin);
        last_par = 0;
        if (++t >= t_end) break;
        tile_body(B, A, t + 1 < t_end ? t + 1 : t_last, 1u, false);
        last_par = 1;
        if (++t >= t_end) break;
    }
    if constexpr (DOT) {
        __syncthreads();
        fold_dot(last_par, tile0 + t_last);
    }
}

// stage offsets: code = BYTES from the start of the tile's LDS stage of x to x[column] -- the stage holds the tile's intervals
// back to back, each from its 4-aligned start in whole 16-byte chunks (exactly what the kernel's fill does)
__global__ void __launch_bounds__(kBlock)
k_stream_stage_codes(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, const uint32_t *__restrict__ win,
                     uint64_t n_rows, uint64_t n_tiles, uint32_t elem_bytes, uint16_t *__restrict__ code) {
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint32_t *w = win + 8 * t;
        const uint32_t a0 = w[0], e0 = w[1], a1 = w[2], e1 = w[3], a2 = w[4], e2 = w[5], a3 = w[6], e3 = w[7];
        const uint32_t al0 = a0 & ~3u, al1 = a1 & ~3u, al2 = a2 & ~3u, al3 = a3 & ~3u;
        const uint32_t n0 = e0 > a0 ? (e0 - al0 + 3u) >> 2 : 0u, n1 = e1 > a1 ? (e1 - al1 + 3u) >> 2 : 0u;
        const uint32_t n2 = e2 > a2 ? (e2 - al2 + 3u) >> 2 : 0u;
        const uint32_t p1 = n0, p2 = p1 + n1, p3 = p2 + n2;
        const uint64_t r0 = t * kStreamRows, r1 = r0 + kStreamRows < n_rows ? r0 + kStreamRows : n_rows;
        const uint64_t k0 = off[r0], k1 = off[r1];
        for (uint64_t k = k0 + threadIdx.x; k < k1; k += kBlock) {
            const uint32_t c = col[k];
            // the used intervals are a prefix of the table, sorted, disjoint, and contain every column of the tile
            const uint32_t q = (uint32_t)(e1 > a1 && c >= a1) + (uint32_t)(e2 > a2 && c >= a2) + (uint32_t)(e3 > a3 && c >= a3);
            const uint32_t al = q == 3u ? al3 : q == 2u ? al2 : q == 1u ? al1 : al0;
            const uint32_t pq = q == 3u ? p3 : q == 2u ? p2 : q == 1u ? p1 : 0u;
            code[k] = (uint16_t)((4u * pq + (c - al)) * elem_bytes);
        }
    }
}

// ---- the value dictionary (see K1s XD-V above) --------------------------------------------------------------------------------------
// Distinct BIT PATTERNS of val[0..nnz), at most kXdDictCap of them, exactly -- or the verdict that there are more.  Phase 1: every
// workgroup collects the patterns of its share in an LDS table (compare-and-swap into the first empty slot) and leaves the table in
// global memory; a workgroup whose table overflows raises a flag that makes everybody stop (a matrix with arbitrary values is found
// out within the first few hundred entries of every workgroup).  Phase 2: one workgroup merges the tables the same way, sorts the
// result by bit pattern (the dictionary must not depend on who came first) and publishes it.
constexpr uint32_t kXdDictCap = 32;
// (a slot is empty while it holds this pattern -- a NaN payload no computation produces; a matrix that does hold it is reported as
// "more than the dictionary holds", which costs it nothing but this optimisation)
template <typename T> struct XdPat;
template <> struct XdPat<float> { typedef uint32_t U; static constexpr U kEmpty = 0x7FC5A5A5u; };
template <> struct XdPat<double> { typedef unsigned long long U; static constexpr U kEmpty = 0x7FF8A5A55A5AA5A5ull; };

template <typename U>
__device__ __forceinline__ bool xd_dict_insert(U *tab, U pat, U empty) {  // false: the table is full and does not hold pat
    for (uint32_t i = 0; i < kXdDictCap; ++i) {
        const U seen = tab[i];
        if (seen == pat) return true;
        if (seen == empty) {
            const U old = atomicCAS(tab + i, empty, pat);
            if (old == empty || old == pat) return true;
        }
    }
    return false;
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_value_dict_collect(const T *__restrict__ val, uint64_t nnz, typename XdPat<T>::U *__restrict__ wg_tables, uint32_t *__restrict__ overflow) {
    typedef typename XdPat<T>::U U;
    __shared__ U tab[kXdDictCap];
    if (threadIdx.x < kXdDictCap) tab[threadIdx.x] = XdPat<T>::kEmpty;
    __syncthreads();
    const U *bits = reinterpret_cast<const U *>(val);
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nnz; i += (uint64_t)gridDim.x * kBlock) {
        if (*(volatile uint32_t *)overflow) break;  // somebody found a 33rd value: nothing left to learn
        const U pat = bits[i];
        if (pat == XdPat<T>::kEmpty || !xd_dict_insert<U>(tab, pat, XdPat<T>::kEmpty)) { atomicOr(overflow, 1u); break; }
    }
    __syncthreads();
    if (threadIdx.x < kXdDictCap) wg_tables[(uint64_t)blockIdx.x * kXdDictCap + threadIdx.x] = tab[threadIdx.x];
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_value_dict_merge(const typename XdPat<T>::U *__restrict__ wg_tables, uint32_t n_tables, T *__restrict__ dict, uint32_t *__restrict__ count,
                   uint32_t *__restrict__ overflow) {
    typedef typename XdPat<T>::U U;
    __shared__ U tab[kXdDictCap];
    __shared__ uint32_t s_over;
    if (threadIdx.x < kXdDictCap) tab[threadIdx.x] = XdPat<T>::kEmpty;
    if (threadIdx.x == 0) s_over = *overflow;
    __syncthreads();
    if (!s_over) {
        for (uint32_t i = threadIdx.x; i < n_tables * kXdDictCap; i += kBlock) {
            const U pat = wg_tables[i];
            if (pat != XdPat<T>::kEmpty && !xd_dict_insert<U>(tab, pat, XdPat<T>::kEmpty)) s_over = 1u;  // (benign race: any writer writes 1)
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t n = 0;
        U sorted[kXdDictCap];
        for (uint32_t i = 0; i < kXdDictCap; ++i)
            if (tab[i] != XdPat<T>::kEmpty) sorted[n++] = tab[i];
        for (uint32_t i = 1; i < n; ++i) {  // insertion sort by bit pattern: <= 32 entries, once per matrix
            const U v = sorted[i];
            uint32_t j = i;
            while (j > 0 && sorted[j - 1] > v) { sorted[j] = sorted[j - 1]; --j; }
            sorted[j] = v;
        }
        U *out = reinterpret_cast<U *>(dict);
        for (uint32_t i = 0; i < kXdDictCap; ++i) out[i] = i < n ? sorted[i] : (U)0;
        *count = s_over ? 0u : n;
        *overflow = s_over;
    }
}

// code[k] |= the dictionary index of val[k], spread over the code's spare bits (low `low` bits, then the bits from `high_shift` up)
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_value_dict_encode(const T *__restrict__ val, uint64_t nnz, const T *__restrict__ dict, uint32_t n, uint32_t low, uint32_t high_shift,
                    uint16_t *__restrict__ code) {
    typedef typename XdPat<T>::U U;
    __shared__ U tab[kXdDictCap];
    if (threadIdx.x < kXdDictCap) tab[threadIdx.x] = reinterpret_cast<const U *>(dict)[threadIdx.x];
    __syncthreads();
    const U *bits = reinterpret_cast<const U *>(val);
    for (uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x; k < nnz; k += (uint64_t)gridDim.x * kBlock) {
        const U pat = bits[k];
        uint32_t idx = 0;
        for (uint32_t i = 0; i < n; ++i) idx = tab[i] == pat ? i : idx;  // (every pattern is in the dictionary: it was built from these values)
        const uint32_t lowmask = (1u << low) - 1u;
        code[k] = (uint16_t)(code[k] | (idx & lowmask) | ((idx >> low) << high_shift));
    }
}

// rows of odd length (decides whether the unskewed stage applies)
__global__ void __launch_bounds__(kBlock)
k_stream_odd_rows(const uint8_t *__restrict__ len8, uint64_t n, unsigned long long *__restrict__ out) {
    uint32_t c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) c += len8[i] & 1u;
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) c += (uint32_t)__shfl_down((int)c, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && c) atomicAdd(out, (unsigned long long)c);  // integer count: exact, order independent
}

// workgroups of k_spmv_stream_xdp<T, DOT, XS, VD> a CU holds.  Both DOT forms are asked about at the first launch of either: the dot-fused
// one is first launched inside the solver's stream capture, where the runtime refuses the query (the fallback of 4 then halved f32's
// grid: 1.74 against 1.45 ms per CG iteration until this was found); a refused query is not remembered.
template <typename T, int XS, int VD> int xdp_resident(bool dot) {
    static int res[2] = {0, 0};
    if (res[0] == 0 || res[1] == 0) {
        int a = 0, b = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_spmv_stream_xdp<T, false, XS, VD>, kBlock, 0) == hipSuccess && a > 0) res[0] = a;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_spmv_stream_xdp<T, true, XS, VD>, kBlock, 0) == hipSuccess && b > 0) res[1] = b;
        (void)hipGetLastError();
    }
    const int r = res[dot ? 1 : 0];
    return r > 0 ? r : (sizeof(T) == 8 ? 4 : 8);
}

template <typename T>
int launch_xd_t(const T *val, const T *x, T *y, size_t n_rows, T *dot_partials, const uint16_t *scode, const uint32_t *cwin,
                const uint8_t *len8, const uint32_t *tbase, const T *dot_lhs, hipStream_t s, int xs, uint64_t tile_begin, uint64_t tile_end,
                const T *dict, bool dict_high) {
    const uint64_t all_tiles = stream_tiles(n_rows, 1);
    const uint64_t tile0 = tile_begin < all_tiles ? tile_begin : all_tiles, tile1 = tile_end < all_tiles ? tile_end : all_tiles;
    if (tile1 <= tile0) return SMH_OK;
    const uint64_t n_tiles = tile1 - tile0;
    const dim3 grid((unsigned)n_tiles), block(kBlock);
    // K1s XD-V in f64 runs on persistent workgroups (k_spmv_stream_xdp), as many as the chip holds at once.  Measured on C4 (512^3,
    // profiles/r04_k1s_xdv_persistent.log): f64 product 1.45 -> 1.15 ms, CG iteration 3.18 -> 2.99 ms; f32 product 0.73 ms either way
    // (bound by instruction issue, not latency: 8 workgroups per CU already hide the round trips) and its dot-fused form slower
    // (1.76 against 1.52 ms per CG iteration), so f32 keeps one workgroup per tile.  SMH_STREAM_PERSIST = 0 / 1 forces it off / on for both.
    static const int persist_env = getenv("SMH_STREAM_PERSIST") ? atoi(getenv("SMH_STREAM_PERSIST")) : -1;  // tuning knob
    const bool persist = persist_env < 0 ? sizeof(T) == 8 : persist_env != 0;
    if (dict && persist && (xs == 2 || xs == 4)) {
        static int cus_cache[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        int &cus = cus_cache[dev & 63];
        if (cus == 0) {
            hipDeviceProp_t prop;
            cus = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        static const int per_cu_env = getenv("SMH_STREAM_PERSIST_WGS") ? atoi(getenv("SMH_STREAM_PERSIST_WGS")) : 0;  // tuning knob
#define SMH_XDP(D, P, V)                                                                                                           \
    do {                                                                                                                           \
        const int resident = xdp_resident<T, P, V>(D); /* workgroups of this instantiation one CU holds */                         \
        uint64_t wgx = (uint64_t)(cus / 8 > 0 ? cus / 8 : 1) * (uint64_t)(per_cu_env > 0 ? per_cu_env : resident); /* per XCD */   \
        if (wgx > (n_tiles + 7) / 8) wgx = (n_tiles + 7) / 8;                                                                      \
        if (wgx < 1) wgx = 1;                                                                                                      \
        hipLaunchKernelGGL((k_spmv_stream_xdp<T, D, P, V>), dim3((unsigned)(8 * wgx)), block, 0, s, x, y, (uint64_t)n_rows, n_tiles, \
                           dot_partials, scode, cwin, len8, tbase, dot_lhs, tile0, dict);                                          \
    } while (0)
#define SMH_XDP2(P)                                                                              \
    do {                                                                                         \
        if (dict_high) { if (dot_partials) SMH_XDP(true, P, 2); else SMH_XDP(false, P, 2); }    \
        else { if (dot_partials) SMH_XDP(true, P, 1); else SMH_XDP(false, P, 1); }              \
    } while (0)
        if (xs == 2) SMH_XDP2(2); else SMH_XDP2(4);
#undef SMH_XDP2
#undef SMH_XDP
        SMH_HIP(hipGetLastError());
        return SMH_OK;
    }
#define SMH_XD(D, P, V)                                                                                                              \
    hipLaunchKernelGGL((k_spmv_stream_xd<T, D, P, V>), grid, block, 0, s, val, x, y, (uint64_t)n_rows, n_tiles, dot_partials, scode, \
                       cwin, len8, tbase, dot_lhs, tile0, dict)
#define SMH_XD2(P)                                                                                  \
    do {                                                                                            \
        if (dict && dict_high) { if (dot_partials) SMH_XD(true, P, 2); else SMH_XD(false, P, 2); } \
        else if (dict) { if (dot_partials) SMH_XD(true, P, 1); else SMH_XD(false, P, 1); }         \
        else { if (dot_partials) SMH_XD(true, P, 0); else SMH_XD(false, P, 0); }                   \
    } while (0)
    if (xs == 2) SMH_XD2(2);
    else if (xs == 4) SMH_XD2(4);
    else return fail(SMH_ERR_INVALID, "K1s XD: the stage of x holds 2 or 4 chunks per thread");
#undef SMH_XD2
#undef SMH_XD
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

}  // namespace

// dict != NULL: the codes carry value-dictionary indices (K1s XD-V); val is then not read
int launch_spmv_stream_xd(int dtype, const void *val, const void *x, void *y, size_t n_rows, void *dot_partials, const uint16_t *scode,
                          const uint32_t *cwin, const uint8_t *len8, const uint32_t *tbase, const void *dot_lhs, hipStream_t s, int xs,
                          uint64_t tile_begin, uint64_t tile_end, const void *dict, bool dict_high) {
    if (n_rows == 0) return SMH_OK;
    if (dot_partials && !dot_lhs) dot_lhs = x;  // CG's p.Ap
    if (!dot_partials && !y) return fail(SMH_ERR_INVALID, "K1s: no output");
    if (dtype == SMH_F64)
        return launch_xd_t<double>((const double *)val, (const double *)x, (double *)y, n_rows, (double *)dot_partials, scode, cwin, len8,
                                   tbase, (const double *)dot_lhs, s, xs, tile_begin, tile_end, (const double *)dict, dict_high);
    return launch_xd_t<float>((const float *)val, (const float *)x, (float *)y, n_rows, (float *)dot_partials, scode, cwin, len8, tbase,
                              (const float *)dot_lhs, s, xs, tile_begin, tile_end, (const float *)dict, dict_high);
}

// The dictionary of val's distinct bit patterns: dict_out (device, 32 entries of the value type, sorted by pattern, unused ones zero)
// and *count_out = how many (0: more than 32, or the reserved pattern occurs).  Synchronises the stream.
template <typename T>
static int value_dict_t(const T *val, size_t nnz, T *dict_out, uint32_t *count_out, hipStream_t s) {
    typedef typename XdPat<T>::U U;
    *count_out = 0;
    if (nnz == 0) return SMH_OK;
    uint64_t blocks = (nnz + (uint64_t)kBlock * 64 - 1) / ((uint64_t)kBlock * 64);
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    U *tables = nullptr;
    uint32_t *flags = nullptr;  // [0] overflow, [1] count
    SMH_HIP(hipMalloc((void **)&tables, blocks * kXdDictCap * sizeof(U)));
    hipError_t e = hipMalloc((void **)&flags, 2 * sizeof(uint32_t));
    uint32_t h[2] = {1u, 0u};
    auto go = [&]() -> int {
        SMH_HIP(e);
        SMH_HIP(hipMemsetAsync(flags, 0, 2 * sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_value_dict_collect<T>, dim3((unsigned)blocks), dim3(kBlock), 0, s, val, (uint64_t)nnz, tables, flags);
        SMH_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_value_dict_merge<T>, dim3(1), dim3(kBlock), 0, s, (const U *)tables, (uint32_t)blocks, dict_out, flags + 1, flags);
        SMH_HIP(hipGetLastError());
        SMH_HIP(hipMemcpyAsync(h, flags, sizeof h, hipMemcpyDeviceToHost, s));
        SMH_HIP(hipStreamSynchronize(s));
        return SMH_OK;
    };
    const int rc = go();
    (void)hipFree(tables);
    (void)hipFree(flags);
    SMH_TRY(rc);
    *count_out = h[0] ? 0u : h[1];
    return SMH_OK;
}

int stream_value_dict(int dtype, const void *val, size_t nnz, void *dict_out, uint32_t *count_out, hipStream_t s) {
    return dtype == SMH_F64 ? value_dict_t<double>((const double *)val, nnz, (double *)dict_out, count_out, s)
                            : value_dict_t<float>((const float *)val, nnz, (float *)dict_out, count_out, s);
}

// code[k] |= index of val[k] in the dictionary (n entries), in the spare bits of a stage-offset code for a stage of xs * 1024 entries
// do n dictionary entries fit the spare bits ABOVE the offset alone (the cheaper form for the kernel)?
bool stream_value_dict_high(int dtype, uint32_t n, int xs) {
    const uint32_t low = dtype == SMH_F64 ? 3u : 2u, high_shift = low + (xs == 4 ? 12u : 11u);
    return n <= (1u << (16u - high_shift));
}

int launch_stream_value_codes(int dtype, const void *val, size_t nnz, const void *dict, uint32_t n, int xs, uint16_t *code, hipStream_t s) {
    if (nnz == 0) return SMH_OK;
    const uint32_t high_shift = (dtype == SMH_F64 ? 3u : 2u) + (xs == 4 ? 12u : 11u);
    const uint32_t low = stream_value_dict_high(dtype, n, xs) ? 0u : (dtype == SMH_F64 ? 3u : 2u);  // (low == 0: the whole index goes above the offset)
    uint64_t blocks = (nnz + kBlock - 1) / kBlock;
    if (blocks > 16384) blocks = 16384;
    if (dtype == SMH_F64)
        hipLaunchKernelGGL(k_value_dict_encode<double>, dim3((unsigned)blocks), dim3(kBlock), 0, s, (const double *)val, (uint64_t)nnz, (const double *)dict, n, low, high_shift, code);
    else
        hipLaunchKernelGGL(k_value_dict_encode<float>, dim3((unsigned)blocks), dim3(kBlock), 0, s, (const float *)val, (uint64_t)nnz, (const float *)dict, n, low, high_shift, code);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

// how many dictionary entries the spare bits of a code can name, for a stage of xs * 1024 entries
uint32_t stream_value_dict_capacity(int xs) { return xs == 4 ? 16u : 32u; }

int launch_stream_stage_codes(const uint32_t *off, const uint32_t *col, const uint32_t *win, size_t n_rows, uint32_t elem_bytes,
                              uint16_t *code, hipStream_t s) {
    const uint64_t n_tiles = (n_rows + kStreamRows - 1) / kStreamRows;
    if (n_tiles == 0) return SMH_OK;
    const uint64_t blocks = n_tiles < 16384 ? n_tiles : 16384;
    hipLaunchKernelGGL(k_stream_stage_codes, dim3((unsigned)blocks), dim3(kBlock), 0, s, off, col, win, (uint64_t)n_rows, n_tiles, elem_bytes, code);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

// *d_out (unsigned long long, device) = rows of odd length among the n_padded byte lengths
int launch_stream_odd_rows(const uint8_t *len8, size_t n_padded, unsigned long long *d_out, hipStream_t s) {
    SMH_HIP(hipMemsetAsync(d_out, 0, sizeof(unsigned long long), s));
    if (n_padded == 0) return SMH_OK;
    uint64_t blocks = (n_padded + kBlock - 1) / kBlock;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_stream_odd_rows, dim3((unsigned)blocks), dim3(kBlock), 0, s, len8, (uint64_t)n_padded, d_out);
    SMH_HIP(hipGetLastError());
    return SMH_OK;
}

}  // namespace smh
