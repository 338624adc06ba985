// capi.hip -- extern "C" boundary of libsparsemat_hip.so (see include/sparsemat_hip.h).
//
// Host-side logic only: argument checks with the reference's panic conditions, HBM residency of
// the CRS arrays, kernel-variant selection, the CG driver loop.  No CPU compute path exists here:
// every entry point that computes needs a HIP device.
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "internal.hpp"

namespace smh {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int fail(int status, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return status;
}

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    int status = (e == hipErrorOutOfMemory) ? SMH_ERR_OOM : (e == hipErrorNoDevice ? SMH_ERR_NO_DEVICE : SMH_ERR_HIP);
    snprintf(g_err, sizeof g_err, "HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    (void)hipGetLastError();  // clear sticky state
    return status;
}

int require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(SMH_ERR_NO_DEVICE, "no HIP device visible: libsparsemat_hip has no CPU fallback");
    }
    return SMH_OK;
}

int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; }
    return d;
}

// defined in the kernel files
size_t cg_scalars_bytes(int dtype);
int cg_begin(int dtype, void *sc, const void *r, size_t n, void *partials, double tol, size_t iter_max, hipStream_t s);
int cg_iter_tail(int dtype, void *sc, void *sc2, void *x, void *r, void *p, const void *ap, size_t n, void *partials,
                 const void *dot_partials, uint32_t dot_count, hipStream_t s);
void cg_read_scalars(int dtype, const void *host_copy, int *converged, uint64_t *iters, double *rr);
int synth_x(int dtype, uint64_t seed, size_t begin, size_t n, void *x, hipStream_t s);
int synth_fixed(int dtype, uint64_t seed, int pattern, size_t n, uint32_t k, size_t row_begin, size_t row_end,
                uint32_t *off, uint32_t *col, void *val, hipStream_t s);
int synth_fill(int dtype, uint64_t seed, size_t n_cols, size_t row_begin, size_t row_end, const uint32_t *off,
               uint32_t *col, void *val, hipStream_t s);
size_t synth_laplace3d_nnz(size_t nx, size_t ny, size_t nz, size_t row_begin, size_t row_end);
int synth_laplace3d(int dtype, size_t nx, size_t ny, size_t nz, size_t row_begin, size_t row_end, uint32_t *off,
                    uint32_t *col, void *val, hipStream_t s);
void build_ring_plan(size_t n_rows, size_t ring_entries, const uint32_t *cmin, const uint32_t *cmax, size_t n_blocks,
                     uint32_t noring_mode, std::vector<uint32_t> &phase_ptr, std::vector<RingPhase> &phases,
                     double *ring_row_fraction);
void build_ring_plan_banded(size_t n_rows, size_t ring_entries, const uint32_t *win, size_t n_blocks, uint32_t noring_mode,
                            std::vector<uint32_t> &phase_ptr, std::vector<RingPhase> &phases, double *ring_row_fraction);
void synth_powerlaw_cdf(uint32_t kmax, double alpha, uint32_t *cdf);
void synth_powerlaw_lengths(uint64_t seed, size_t row_begin, size_t row_end, uint32_t kmax, const uint32_t *cdf,
                            uint32_t *lengths);

static bool valid_dtype(int dt) { return dt == SMH_F32 || dt == SMH_F64; }

static int ensure_cap(void **buf, size_t *cap, size_t bytes) {
    if (*cap >= bytes && *buf) return SMH_OK;
    if (*buf) { SMH_HIP(hipFree(*buf)); *buf = nullptr; *cap = 0; }
    size_t want = bytes < 256 ? 256 : bytes;
    SMH_HIP(hipMalloc(buf, want));
    *cap = want;
    return SMH_OK;
}

// ---- variant selection ---------------------------------------------------------------------------
static int auto_lanes(const smh_crs *m) {
    if (m->forced_lanes) return m->forced_lanes;
    const double mean = m->n_rows ? (double)m->nnz / (double)m->n_rows : 0.0;
    // one pass of a lane group covers 4*lanes entry slots: size the group to the mean row; short rows
    // start anywhere inside their first 16-B chunk, so they get 3 slots of slack (measured on the 7-point
    // Laplacian: 4 lanes 2.87 ms, 2 lanes 3.68 ms, 1 lane x 3 chunks 3.11 ms)
    const double need = mean < 16.0 ? mean + 3.0 : mean;
    int lanes = 1;
    while (lanes < 64 && 4.0 * lanes < need) lanes <<= 1;
    // Long rows in the pipelined body (contiguous-band matrices, 320 M entries, same box): rows of 128 / 256 entries
    // run in 0.313 / 0.301 ms with 16 lanes (2 / 4 passes of 64 slots) against 0.359 / 0.348 ms with one pass of 32 / 64
    // lanes; rows of 100 entries prefer one pass of 32 lanes (0.398 vs 0.500 ms).
    if (m->use_ring != 0) {
        if (mean >= 128.0) lanes = 16;
        else if (lanes > 32) lanes = 32;
        // ... and when their columns do not fit the ring (global gathers, one cache line each) 8 lanes lose least
        // (banded +-32768, rows of 256: 1.28-1.33 ms with 4-8 lanes, 1.51 ms with 16, 1.62-1.80 ms with 64)
        if (mean > 32.0 && m->ring_planned && m->ring_fraction < 0.5) lanes = 8;
    }
    return lanes;
}

// lanes sized to the mean row alone (the skew test of AUTO)
static int mean_lanes(const smh_crs *m) {
    const double mean = m->n_rows ? (double)m->nnz / (double)m->n_rows : 0.0;
    const double need = mean < 16.0 ? mean + 3.0 : mean;
    int lanes = 1;
    while (lanes < 64 && 4.0 * lanes < need) lanes <<= 1;
    return lanes;
}

// 16-B chunks each lane loads per pass (pipelined K1r body only): 4*lanes*chunks entry slots per row and pass
static int auto_chunks(const smh_crs *m) {
    const int lanes = auto_lanes(m);
    if (m->forced_chunks) {
        if (lanes == 1) return m->forced_chunks;
        if (lanes == 2) return m->forced_chunks > 2 ? 2 : m->forced_chunks;
        return 1;
    }
    return 1;
}

// K2c geometry: column blocks of 2 MiB of x
// 2^19 columns: 2 MiB of f32 x.  f64 takes the same width (4 MiB of x, a whole L2): measured on C3, 20 blocks of
// 2^19 run in 3.64 ms, 39 blocks of 2^18 in 5.08 ms -- the per-block sweeps of offsets and y (12 B + 8 B per row)
// outweigh the better hit rate (profiles/r01_colblock_sweep.log)
static uint32_t cb_shift_for(const smh_crs *m) { return m->cb_forced_shift ? m->cb_forced_shift : 19u; }
static size_t cb_blocks_for(const smh_crs *m) {
    const uint64_t w = 1ull << cb_shift_for(m);
    const uint64_t b = ((uint64_t)m->n_cols + w - 1) / w;
    return (size_t)(b ? b : 1);
}
// values changed (update_values / scale): the blocked copy is rebuilt on its next use
static void drop_colblock(smh_crs *m) {
    tiled_free(m);
    (void)hipFree(m->d_cb_off); (void)hipFree(m->d_cb_col); (void)hipFree(m->d_cb_val);
    m->d_cb_off = m->d_cb_col = nullptr;
    m->d_cb_val = nullptr;
    m->cb_built = false;
    (void)hipFree(m->d_cf_seg); (void)hipFree(m->d_cf_cnt); (void)hipFree(m->d_cf_col); (void)hipFree(m->d_cf_val);
    (void)hipFree(m->d_cf_tile_row);
    m->d_cf_seg = m->d_cf_col = m->d_cf_tile_row = nullptr;
    m->d_cf_cnt = nullptr;
    m->d_cf_val = nullptr;
    m->cf_built = m->cf_ok = false;
    (void)smh_crs_destroy(m->split_long);
    (void)smh_crs_destroy(m->split_short);
    m->split_long = m->split_short = nullptr;
    (void)hipFree(m->d_split_rows); (void)hipFree(m->d_split_y);
    m->d_split_rows = nullptr;
    m->d_split_y = nullptr;
    m->split_built = m->split_ok = false;
    m->split_n_long = 0;
    if (m->split_stream) { (void)hipStreamSynchronize(m->split_stream); (void)hipStreamDestroy(m->split_stream); m->split_stream = nullptr; }
    if (m->split_fork) { (void)hipEventDestroy(m->split_fork); m->split_fork = nullptr; }
    if (m->split_join) { (void)hipEventDestroy(m->split_join); m->split_join = nullptr; }
}
// K2f geometry: blocks of 2^18 columns (1 MiB of f32 x, 2 MiB of f64 x): its waves walk the blocks without a barrier and
// spread over a few of them, so the L2 has to hold more than one (measured on C2-uniform, f32: 2^18 2.05 ms, 2^19 2.40 ms);
// a forced width applies to both blocked variants
static uint32_t cf_shift_for(const smh_crs *m) { return m->cb_forced_shift ? m->cb_forced_shift : 18u; }
static size_t cf_blocks_for(const smh_crs *m) {
    const uint64_t w = 1ull << cf_shift_for(m);
    const uint64_t b = ((uint64_t)m->n_cols + w - 1) / w;
    return (size_t)(b ? b : 1);
}
// x too large for the L2s AND rows whose columns span a large part of it (statistic taken at create time for
// matrices with more than 8 MiB of x): gathers would miss L1 and L2 -> column-blocked execution
// (tools/experiment_gather.py, 4M rows x 32 uniform columns, f32: K1 / K2c at x = 4 MiB 0.80 / 0.68 ms, 8 MiB
// 1.25 / 0.66 ms, 16 MiB 1.76 / 0.71 ms, 64 MiB 2.35 / 1.44 ms)
constexpr size_t kColblockMinXBytes = 4u << 20;
static bool wants_colblock(const smh_crs *m) {
    const size_t b = cb_blocks_for(m);
    return m->n_cols * dtype_size(m->dtype) >= kColblockMinXBytes && m->span_fraction > 0.25 && b >= 2 && b <= 128;
}

// K2t is an option: not switched off, its copy was not refused, >= min_tile entries per (slice, row block) tile, a tile table of <= 1 GiB
static bool tiled_fits(const smh_crs *m, double min_tile = 32.0) {
    static const bool tiled_off = getenv("SMH_TILED") && atoi(getenv("SMH_TILED")) == 0;  // tuning knob
    if (tiled_off || (m->t2_built && !m->t2_ok)) return false;
    uint32_t n_cb = 0, R = 0, n_rb = 0;
    tiled_geometry(m->n_rows, m->n_cols, m->nnz, m->dtype, &n_cb, &R, &n_rb);
    const double tile = (double)m->nnz / (double)n_cb / (double)n_rb;
    return tile >= min_tile && (double)(n_rb + 1) * (double)n_cb * 4.0 <= (double)(1u << 30);
}

static int resolve_variant(const smh_crs *m, int variant) {
    if (variant != SMH_SPMV_AUTO) return variant;
    if (wants_colblock(m)) {
        // one sweep over y (K2f) unless its byte table cannot describe the matrix / it was switched off
        static const bool fused_off = getenv("SMH_COLBLOCK_FUSED") && atoi(getenv("SMH_COLBLOCK_FUSED")) == 0;  // tuning knob
        if (fused_off || (m->cf_built && !m->cf_ok) || cf_blocks_for(m) > 255) return SMH_SPMV_COLBLOCK;
        // K2f's waves keep their tiles for the whole sweep: that only works while they stay together, i.e. for rows of
        // similar length.  Skewed rows (BASELINE C3, power law 1..2048) let them drift over all column blocks at once
        // -- 5.2-5.6 ms against K2c's 3.25 ms, and a lock step costs more than it recovers (profiles/r02_k2f_sweep.log) --
        // so those stay with the per-block launches, whose tiles the dispatcher hands out dynamically.
        // (measured on parts of C3, profiles/r02_c3_split_probe.log: rows of up to 63 / 127 entries, mean 5.5 / 7.9, still run
        // best through K2f -- 0.82 / 1.15 ms against K2c's 0.97 / 1.26; up to 255 entries K2c wins, 1.61 against 1.79)
        const double mean = m->n_rows ? (double)m->nnz / (double)m->n_rows : 0.0;
        const double similar = 2.0 * mean + 8.0 > 128.0 ? 2.0 * mean + 8.0 : 128.0;
        if ((double)m->max_row_len <= similar) {
            // the two streaming passes of K2t, whose gathers stay in LDS: ahead of K2f on every shape measured (1-10 M rows x 8-64
            // entries, profiles/r02_tiled_crossover.log) -- f32 (16 B per entry against CSR's 8) by 31-66 % (C2-uniform 1.14 ms against
            // 1.90), f64 (28 B against 12) by 13-66 % (10 M columns, rows of 8 / 16 / 32 / 64: 0.71 / 1.06 / 1.99 / 3.63 ms against 1.00 /
            // 1.63 / 3.13 / 6.36; 4 M x 16: 0.44 against 0.50).  Needs >= 32 entries per tile (>= 12 measured on f64 with 39 blocks)
            const size_t blocks = cf_blocks_for(m);
            const bool pays = m->dtype == SMH_F32 ? tiled_fits(m)
                              : m->no_split       ? false  // (the parts of a K2s split stay as measured)
                                                  : tiled_fits(m, blocks >= 24 ? 12.0 : 32.0);
            if (pays) return SMH_SPMV_TILED;
            return SMH_SPMV_COLFUSED;
        }
        // skewed rows: K2t folds the entries a long row has in one slice inside its first pass and cuts its row blocks by product
        // counts, so long rows cost it nothing special.  Round 3's form is ahead on both value types: C3 (f64) 1.6 ms against K2s's 2.76
        // and K2c's 3.25; a 3M-row f32 power law (184 slices) 0.23 ms against K2c's 0.62 (tests/test_auto_choice_gpu.py)
        if (!m->no_split && tiled_fits(m, m->dtype == SMH_F64 && cf_blocks_for(m) >= 24 ? 12.0 : 32.0)) return SMH_SPMV_TILED;  // (the parts of a K2s split stay as measured)
        // ... and a matrix with a minority of long rows is taken apart by row length (K2s)
        static const bool split_off = getenv("SMH_COLBLOCK_SPLIT") && atoi(getenv("SMH_COLBLOCK_SPLIT")) == 0;  // tuning knob
        // ... when K2c's sweeps (per column block and row: one offset, y read and written) weigh as much as the entries
        // themselves: C3 (f64, 10M rows, 20 blocks) 4.0 GB of sweeps for 3.8 GB of entries -> 2.78 against 3.26 ms; a 3M-row
        // f32 power law (6 blocks) 0.22 GB for 0.76 GB -> K2c stays ahead, 0.61 against 0.75 ms
        const double vs = (double)dtype_size(m->dtype);
        const double sweeps = (double)cb_blocks_for(m) * (double)m->n_rows * (4.0 + 2.0 * vs), entries = (double)m->nnz * (4.0 + vs);
        if (!split_off && !m->no_split && !(m->split_built && !m->split_ok) && sweeps >= 0.6 * entries) return SMH_SPMV_COLSPLIT;
        return SMH_SPMV_COLBLOCK;
    }
    const int lanes = mean_lanes(m);
    // short rows (stencils, FEM): the dense CSR-stream kernel (a tile denser than its LDS stage takes several passes)
    const double mean = m->n_rows ? (double)m->nnz / (double)m->n_rows : 0.0;
    // ... except rows of 9-12 entries whose columns fit the LDS ring (plan taken at create time): there the ring kernel
    // with 4 lanes wins (banded, 320 M entries, rows of 12: 0.445 vs 0.637 ms; rows of 8: 0.572 vs 0.586 ms, a tie)
    const bool short_ring_rows = mean > 8.0 && m->ring_planned && m->ring_fraction >= 0.5 && m->use_ring != 0;
    if (mean <= 12.0 && m->max_row_len <= 64 && !short_ring_rows) return SMH_SPMV_STREAM;
    // skew test: the longest row needs >= 8 passes of a group sized for the mean row
    if ((uint64_t)m->max_row_len >= 8ull * 4ull * (uint64_t)lanes && m->max_row_len > 64) return SMH_SPMV_MERGE;
    // long rows whose columns do not fit the LDS ring (plan taken at create time): every kernel is then bound by one
    // cache line per gather; the dense stream kernel loses least up to ~128 entries per row (banded +-8192..32768,
    // 320 M entries: rows of 64 / 128: K1s 0.91 / 1.18 ms, lane-group kernels 1.24-1.42 / 1.34-1.69 ms)
    if (mean > 32.0 && mean <= 128.0 && m->ring_planned && m->ring_fraction < 0.5) return SMH_SPMV_STREAM;
    return SMH_SPMV_VECTOR;
}

static int ensure_merge_ws(smh_crs *m) {
    if (m->d_tile_row) return SMH_OK;
    const uint64_t items = (uint64_t)m->n_rows + (uint64_t)m->nnz;
    m->n_tiles = (size_t)((items + kMergeTile - 1) / kMergeTile);
    if (m->n_tiles == 0) return SMH_OK;
    SMH_HIP(hipMalloc((void **)&m->d_tile_row, (m->n_tiles + 1) * sizeof(uint32_t)));
    SMH_HIP(hipMalloc((void **)&m->d_tile_nz, (m->n_tiles + 1) * sizeof(uint32_t)));
    SMH_HIP(hipMalloc((void **)&m->d_carry_row, m->n_tiles * sizeof(uint32_t)));
    SMH_HIP(hipMalloc(&m->d_carry_val, m->n_tiles * dtype_size(m->dtype)));
    SMH_TRY(launch_merge_table(m->d_off, m->n_rows, m->nnz, m->n_tiles, m->d_tile_row, m->d_tile_nz, m->stream));
    SMH_HIP(hipStreamSynchronize(m->stream));
    return SMH_OK;
}

// K1r: inspector pass + host plan, once per matrix
// with_bands = false: the single-window plans only (what AUTO needs at create time from a matrix with rows too short
// for the lane-group kernels anyway); the banded attempt -- an inspector pass, a table readback, a host pass over the
// tiles: ~50 ms at 134 M rows -- then waits until the VECTOR family is actually used.
static int ensure_ring_plan(smh_crs *m, bool with_bands = true) {
    if (m->ring_planned && (!with_bands || m->ring_bands_tried)) return SMH_OK;
    if (m->ring_planned) {  // planned without the banded attempt: plan again, completely
        (void)hipFree(m->d_phase_ptr); (void)hipFree(m->d_phases);
        m->d_phase_ptr = nullptr;
        m->d_phases = nullptr;
        m->ring_planned = false;
    }
    m->ring_bands_tried = with_bands;
    const size_t n_tiles = (m->n_rows + 63) / 64;
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, m->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    // Two 512-thread blocks (64 KiB of LDS each) are resident per CU; the row range is cut into 3x as many
    // blocks so that the hardware dispatcher evens out the tail (round 1, on C2: 2/CU 0.4245 ms, 8/CU 0.404 ms, 24/CU 0.402 ms;
    // round 3, the final kernel: 4 / 6 / 8 / 12 / 16 per CU 0.360 / 0.357 / 0.361 / 0.364 / 0.371 ms, and 0.350-0.353 against
    // 0.356-0.359 ms for 6 against 8 on banded and window matrices, f64 level -- profiles/r03_k1r_blocks_per_cu.log)
    unsigned per_cu = 6;
    if (const char *e = getenv("SMH_RING_BLOCKS_PER_CU")) {  // tuning knob
        const int v = atoi(e);
        if (v >= 1 && v <= 64) per_cu = (unsigned)v;
    }
    unsigned blocks = per_cu * (unsigned)cus;
    blocks = (blocks + 7u) & ~7u;
    std::vector<uint32_t> cmin(n_tiles), cmax(n_tiles);
    if (n_tiles) {
        uint32_t *d_min = nullptr, *d_max = nullptr;
        SMH_HIP(hipMalloc((void **)&d_min, n_tiles * sizeof(uint32_t)));
        hipError_t e = hipMalloc((void **)&d_max, n_tiles * sizeof(uint32_t));
        int rc = e == hipSuccess ? launch_tile_span(m->d_off, m->d_col, m->n_rows, n_tiles, d_min, d_max, m->stream)
                                 : hip_fail(e, "hipMalloc", __FILE__, __LINE__);
        if (rc == SMH_OK) {
            e = hipMemcpyAsync(cmin.data(), d_min, n_tiles * sizeof(uint32_t), hipMemcpyDeviceToHost, m->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(cmax.data(), d_max, n_tiles * sizeof(uint32_t), hipMemcpyDeviceToHost, m->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
            if (e != hipSuccess) rc = hip_fail(e, "tile span readback", __FILE__, __LINE__);
        }
        (void)hipFree(d_min); (void)hipFree(d_max);
        SMH_TRY(rc);
    }
    {  // locality statistic for AUTO: mean column span of a 64-row tile relative to n_cols
        double acc = 0.0;
        size_t used = 0;
        for (size_t t = 0; t < n_tiles; ++t)
            if (cmin[t] <= cmax[t]) { acc += (double)(cmax[t] - cmin[t]) + 1.0; ++used; }
        m->span_fraction = used && m->n_cols ? acc / (double)used / (double)m->n_cols : 0.0;
    }
    std::vector<uint32_t> phase_ptr;
    std::vector<RingPhase> phases;
    // Phases without a ring gather through L1/L2.  Bypassing L1 (nontemporal gathers, mode 2) was measured
    // SLOWER on both kinds of such matrices (uniform columns 6.29 vs 5.48 ms, 512^3 Laplacian 3.77 vs 2.68 ms),
    // so it stays a tuning knob.  The uniform case is bound by the per-CU L1 miss path (rocprofv3: TA busy 78 %,
    // 71 % of wave cycles stalled on VMEM issue, ~59 G gathers/s) -- only column blocking would change that.
    uint32_t noring_mode = 0;
    if (const char *e = getenv("SMH_GATHER_NT")) noring_mode = atoi(e) ? 2u : 0u;
    build_ring_plan(m->n_rows, kRingEntries, cmin.data(), cmax.data(), blocks, noring_mode, phase_ptr, phases,
                    &m->ring_fraction);
    m->ring_entries = kRingEntries;
    // f32 rows that do not fit 16384 columns but fit 32768: the wide ring (128 KiB of LDS, one 1024-thread block per CU
    // like f64, so half as many blocks).  Rows of 64 entries in a +-8192 band: 0.90 ms (K1s) -> see DESIGN.md.
    const char *wide_env = getenv("SMH_RING_WIDE");  // tuning knob: 0 = never
    if (m->dtype == SMH_F32 && m->ring_fraction < 0.5 && !(wide_env && atoi(wide_env) == 0)) {
        std::vector<uint32_t> phase_ptr_w;
        std::vector<RingPhase> phases_w;
        double frac_w = 0.0;
        const unsigned blocks_w = ((blocks / 2) + 7u) & ~7u;
        build_ring_plan(m->n_rows, kRingEntriesWide, cmin.data(), cmax.data(), blocks_w, noring_mode, phase_ptr_w, phases_w, &frac_w);
        if (frac_w >= 0.5) {
            phase_ptr.swap(phase_ptr_w);
            phases.swap(phases_w);
            m->ring_fraction = frac_w;
            m->ring_entries = kRingEntriesWide;
            blocks = blocks_w;
        }
    }
    // Still mostly outside the ring: rows that reference a few narrow column intervals far apart (stencils on
    // structured grids) get the BANDED ring -- four bands of a quarter of the ring, one per interval of the tile.
    m->ring_bands = 1;
    const char *band_env = getenv("SMH_RING_BANDS");  // tuning knob: 0 = never
    if (with_bands && m->ring_fraction < 0.5 && n_tiles && m->nnz && !(band_env && atoi(band_env) == 0)) {
        const unsigned sizes[2] = {(unsigned)kRingEntries, (unsigned)kRingEntriesWide};
        const int n_sizes = m->dtype == SMH_F32 && !(wide_env && atoi(wide_env) == 0) ? 2 : 1;
        std::vector<uint32_t> h_win(n_tiles * 8);
        uint32_t *d_win = nullptr, *d_count = nullptr;
        SMH_HIP(hipMalloc((void **)&d_win, n_tiles * 8 * sizeof(uint32_t)));
        hipError_t e = hipMalloc((void **)&d_count, sizeof(uint32_t));
        int rc = e == hipSuccess ? SMH_OK : hip_fail(e, "hipMalloc", __FILE__, __LINE__);
        bool adopted = false;
        for (int si = 0; si < n_sizes && rc == SMH_OK && !adopted; ++si) {
            const unsigned ring = sizes[si];
            rc = launch_tile_intervals(m->d_off, m->d_col, m->n_rows, 64, ring / 4, d_win, d_count, m->stream);
            if (rc == SMH_OK) {
                e = hipMemcpyAsync(h_win.data(), d_win, n_tiles * 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, m->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
                if (e != hipSuccess) rc = hip_fail(e, "tile interval readback", __FILE__, __LINE__);
            }
            if (rc != SMH_OK) break;
            std::vector<uint32_t> phase_ptr_b;
            std::vector<RingPhase> phases_b;
            double frac_b = 0.0;
            const unsigned blocks_b = ring == (unsigned)kRingEntries || m->dtype == SMH_F64 ? blocks : ((blocks / 2) + 7u) & ~7u;
            build_ring_plan_banded(m->n_rows, ring, h_win.data(), blocks_b, noring_mode, phase_ptr_b, phases_b, &frac_b);
            if (frac_b >= 0.5 && frac_b > m->ring_fraction) {
                phase_ptr.swap(phase_ptr_b);
                phases.swap(phases_b);
                m->ring_fraction = frac_b;
                m->ring_entries = ring;
                m->ring_bands = 4;
                blocks = blocks_b;
                adopted = true;
            }
        }
        (void)hipFree(d_count);
        if (adopted) m->d_ring_win = d_win;  // kept: the 16-bit ring slots are built from it on first use
        else (void)hipFree(d_win);
        SMH_TRY(rc);
    }
    SMH_HIP(hipMalloc((void **)&m->d_phase_ptr, phase_ptr.size() * sizeof(uint32_t)));
    SMH_HIP(hipMalloc((void **)&m->d_phases, (phases.size() + 1) * sizeof(RingPhase)));
    SMH_HIP(hipMemcpy(m->d_phase_ptr, phase_ptr.data(), phase_ptr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (!phases.empty())
        SMH_HIP(hipMemcpy(m->d_phases, phases.data(), phases.size() * sizeof(RingPhase), hipMemcpyHostToDevice));
    m->ring_blocks = blocks;
    m->ring_n_phases = phases.size();
    m->ring_planned = true;
    return SMH_OK;
}

// The column-blocked / tiled builders size their tables from n_cols and index them with col >> shift (col / slice): a handle
// created without validation (smh_crs_create_dev's default) may hold columns >= n_cols, which would walk past those tables.
// Every such build starts here (max_col is a create-time statistic: no per-call cost).
int columns_within_n_cols(const smh_crs *m, const char *what) {
    if (m->nnz && (size_t)m->max_col >= m->n_cols)
        return fail(SMH_ERR_INDEX_RANGE, "%s: column index %u >= n_cols %zu", what, m->max_col, m->n_cols);
    return SMH_OK;
}

// K2c: build the column-blocked copy, once per matrix
static int ensure_colblock(smh_crs *m) {
    if (m->cb_built) return SMH_OK;
    SMH_TRY(columns_within_n_cols(m, "column-blocked variant"));
    const size_t blocks = cb_blocks_for(m);
    if (blocks == 0 || blocks > 128)
        return fail(SMH_ERR_INVALID, "column-blocked variant: %zu column blocks (supported: 1..128)", blocks);
    if ((uint64_t)blocks * (m->n_rows + 1) >= (1ull << 34)) return fail(SMH_ERR_OOM, "column-blocked offsets too large");
    m->cb_shift = cb_shift_for(m);
    SMH_TRY(build_colblock(m->dtype, m->d_off, m->d_col, m->d_val, m->n_rows, m->nnz, m->cb_shift, blocks, &m->d_cb_off,
                           &m->d_cb_col, &m->d_cb_val, m->stream));
    m->cb_blocks = blocks;
    // tile height of the K1s launches: the tallest of 2048/1024/512/256 rows whose busiest tile fits the LDS stage
    uint32_t *d_max = nullptr;
    SMH_HIP(hipMalloc((void **)&d_max, sizeof(uint32_t)));
    int rpt = 8;
    int rc = SMH_OK;
    uint32_t worst = 0;
    for (; rpt >= 1; rpt >>= 1) {
        worst = 0;
        for (size_t b = 0; b < blocks && rc == SMH_OK; ++b) {
            uint32_t h = 0;
            rc = launch_stream_max_tile(m->d_cb_off + b * (m->n_rows + 1), m->n_rows, (size_t)kStreamRows * rpt, d_max, m->stream);
            if (rc == SMH_OK) {
                hipError_t e = hipMemcpyAsync(&h, d_max, sizeof h, hipMemcpyDeviceToHost, m->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
                if (e != hipSuccess) rc = hip_fail(e, "colblock tile statistic", __FILE__, __LINE__);
            }
            worst = h > worst ? h : worst;
        }
        if (rc != SMH_OK || worst <= (uint32_t)kStreamCap || rpt == 1) break;
    }
    (void)hipFree(d_max);
    SMH_TRY(rc);
    m->cb_rpt = rpt < 1 ? 1 : rpt;
    m->cb_single_pass = worst <= (uint32_t)kStreamCap;  // else 256-row tiles, several passes where needed
    m->cb_built = true;
    return SMH_OK;
}

// K2f: build the fused column-blocked copy, once per matrix (cf_ok == false afterwards: not describable -> K2c)
static int ensure_colfused(smh_crs *m) {
    if (m->cf_built) return SMH_OK;
    SMH_TRY(columns_within_n_cols(m, "fused column-blocked variant"));
    const size_t blocks = cf_blocks_for(m);
    if (blocks > 255) return fail(SMH_ERR_INVALID, "fused column-blocked variant: %zu column blocks (supported: 1..255)", blocks);
    uint32_t rt = 16;
    if (const char *e = getenv("SMH_COLFUSED_RT")) {  // tuning knob: rows per lane, 8 or 16
        if (atoi(e) == 8) rt = 8;
    }
    bool fits = false;
    SMH_TRY(build_colfused(m->dtype, m->d_off, m->d_col, m->d_val, m->n_rows, m->nnz, cf_shift_for(m), blocks, rt, &m->cf_tiles,
                           &m->d_cf_tile_row, &m->d_cf_seg, &m->d_cf_cnt, &m->d_cf_col, &m->d_cf_val, &fits, m->stream));
    m->cf_shift = cf_shift_for(m);
    m->cf_blocks = blocks;
    m->cf_rt = rt;
    m->cf_ok = fits;
    m->cf_built = true;
    return SMH_OK;
}

// K2s: the row-length split, once per matrix (split_ok == false afterwards: not worth it / not possible -> K2c)
constexpr uint32_t kSplitMinLong = 64;  // rows of this many entries and more form the LONG part
static int finish_create(smh_crs *m, int validate);
static int ensure_split(smh_crs *m) {
    if (m->split_built) return SMH_OK;
    SMH_TRY(columns_within_n_cols(m, "row-length split"));
    m->split_built = true;
    m->split_ok = false;
    if (m->no_split || m->n_rows == 0 || m->nnz == 0) return SMH_OK;
    size_t n_long = 0, nnz_long = 0;
    uint32_t *rows = nullptr, *off_l = nullptr, *col_l = nullptr, *off_s = nullptr, *col_s = nullptr;
    void *val_l = nullptr, *val_s = nullptr;
    SMH_TRY(build_colsplit(m->dtype, m->d_off, m->d_col, m->d_val, m->n_rows, m->nnz, kSplitMinLong, &n_long, &nnz_long, &rows, &off_l, &col_l,
                           &val_l, &off_s, &col_s, &val_s, m->stream));
    auto wrap = [&](size_t n_rows, size_t nnz, uint32_t *off, uint32_t *col, void *val, uint32_t shift, smh_crs **out) -> int {
        smh_crs *p = new (std::nothrow) smh_crs();
        if (!p) { (void)hipFree(off); (void)hipFree(col); (void)hipFree(val); return fail(SMH_ERR_OOM, "host allocation failed"); }
        p->dtype = m->dtype; p->device = m->device; p->n_rows = n_rows; p->n_cols = m->n_cols; p->nnz = nnz;
        p->d_off = off; p->d_col = col; p->d_val = val; p->owns = true;
        p->no_split = true;
        p->cb_forced_shift = shift;
        const int rc = finish_create(p, 0);
        if (rc != SMH_OK) { char keep[512]; strncpy(keep, g_err, sizeof keep); keep[sizeof keep - 1] = 0; smh_crs_destroy(p); return fail(rc, "%s", keep); }
        *out = p;
        return SMH_OK;
    };
    // worth it when the long rows are a minority that holds a good part of the entries
    const bool worth = n_long > 0 && n_long * 4 <= m->n_rows && nnz_long * 4 >= m->nnz;
    if (!worth) {
        (void)hipFree(rows); (void)hipFree(off_l); (void)hipFree(col_l); (void)hipFree(val_l); (void)hipFree(off_s); (void)hipFree(col_s); (void)hipFree(val_s);
        return SMH_OK;
    }
    m->d_split_rows = rows;
    m->split_n_long = n_long;
    // LONG: K2c with 2^18-column blocks (1.93 against 2.03 ms with 2^19 on C3's long part); SHORT: its own AUTO with 2^19
    int rc = wrap(n_long, nnz_long, off_l, col_l, val_l, 18u, &m->split_long);
    if (rc == SMH_OK) rc = wrap(m->n_rows, m->nnz - nnz_long, off_s, col_s, val_s, 19u, &m->split_short);
    else { (void)hipFree(off_s); (void)hipFree(col_s); (void)hipFree(val_s); }
    if (rc == SMH_OK) {
        hipError_t e = hipMalloc(&m->d_split_y, (n_long ? n_long : 1) * dtype_size(m->dtype));
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->split_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->split_fork, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->split_join, hipEventDisableTiming);
        if (e != hipSuccess) rc = hip_fail(e, "split workspace", __FILE__, __LINE__);
    }
    if (rc != SMH_OK) {
        char keep[512]; strncpy(keep, g_err, sizeof keep); keep[sizeof keep - 1] = 0;
        (void)smh_crs_destroy(m->split_long); (void)smh_crs_destroy(m->split_short);
        m->split_long = m->split_short = nullptr;
        (void)hipFree(m->d_split_rows); m->d_split_rows = nullptr;
        return fail(rc, "%s", keep);
    }
    m->split_ok = true;
    return SMH_OK;
}

// K1s: 16-bit column codes, once per matrix.  Kept only when EVERY tile has a description (stencils, bands): the kernel
// variant then has no per-tile branch; any other matrix streams its u32 columns as before and nothing stays allocated.
static int ensure_stream_codes(smh_crs *m) {
    if (m->stream_coded) return SMH_OK;
    m->stream_coded = true;
    const size_t n_tiles = (m->n_rows + kStreamRows - 1) / kStreamRows;
    if (n_tiles == 0 || m->nnz == 0) return SMH_OK;
    uint32_t *d_count = nullptr, h_count = 0;
    SMH_HIP(hipMalloc((void **)&m->d_stream_cwin, n_tiles * 8 * sizeof(uint32_t)));
    SMH_HIP(hipMalloc((void **)&d_count, sizeof(uint32_t)));
    int rc = launch_stream_windows(m->d_off, m->d_col, m->n_rows, m->d_stream_cwin, d_count, m->stream);
    hipError_t e = hipSuccess;
    if (rc == SMH_OK) e = hipMemcpyAsync(&h_count, d_count, sizeof h_count, hipMemcpyDeviceToHost, m->stream);
    if (rc == SMH_OK && e == hipSuccess) e = hipStreamSynchronize(m->stream);
    (void)hipFree(d_count);
    if (rc == SMH_OK && e != hipSuccess) rc = hip_fail(e, "stream code table readback", __FILE__, __LINE__);
    SMH_TRY(rc);
    if ((size_t)h_count != n_tiles) {  // some tile's columns need more than 4 intervals of 16384
        (void)hipFree(m->d_stream_cwin);
        m->d_stream_cwin = nullptr;
        return SMH_OK;
    }
    const size_t n_out = ((m->nnz + 3) & ~size_t(3)) + 4;
    SMH_HIP(hipMalloc((void **)&m->d_stream_code, n_out * sizeof(uint16_t)));
    SMH_HIP(hipMemsetAsync(m->d_stream_code, 0, n_out * sizeof(uint16_t), m->stream));
    SMH_TRY(launch_stream_codes(m->d_off, m->d_col, m->d_stream_cwin, m->n_rows, m->d_stream_code, m->stream));
    m->stream_direct = false;
    // ... and with rows of at most 255 entries the row boundaries shrink from a u32 offset to a byte per row
    if (m->max_row_len <= 255u) {
        SMH_HIP(hipMalloc((void **)&m->d_stream_len8, n_tiles * kStreamRows));
        SMH_HIP(hipMalloc((void **)&m->d_stream_tbase, (n_tiles + 1) * sizeof(uint32_t)));
        SMH_TRY(launch_stream_len8(m->d_off, m->n_rows, m->d_stream_len8, m->d_stream_tbase, m->stream));
        // how much of x a tile's intervals span (decides whether the body that stages x in LDS applies)
        uint32_t *d_xs = nullptr, h_xs[2] = {0xFFFFFFFFu, 0};
        SMH_HIP(hipMalloc((void **)&d_xs, 2 * sizeof(uint32_t)));
        int rc2 = launch_stream_xs_stats(m->d_stream_cwin, n_tiles, d_xs, m->stream);
        hipError_t e2 = rc2 == SMH_OK ? hipMemcpyAsync(h_xs, d_xs, sizeof h_xs, hipMemcpyDeviceToHost, m->stream) : hipSuccess;
        if (rc2 == SMH_OK && e2 == hipSuccess) e2 = hipStreamSynchronize(m->stream);
        (void)hipFree(d_xs);
        SMH_TRY(rc2);
        if (e2 != hipSuccess) return hip_fail(e2, "stream window statistics", __FILE__, __LINE__);
        m->stream_xs_chunks = h_xs[0];
        m->stream_xs_end = h_xs[1];
        // ... and how many rows have an odd length (decides whether the unskewed product stage of K1s XD applies)
        unsigned long long *d_odd = nullptr, h_odd = 0;
        SMH_HIP(hipMalloc((void **)&d_odd, sizeof(unsigned long long)));
        int rc3 = launch_stream_odd_rows(m->d_stream_len8, n_tiles * kStreamRows, d_odd, m->stream);
        hipError_t e3 = rc3 == SMH_OK ? hipMemcpyAsync(&h_odd, d_odd, sizeof h_odd, hipMemcpyDeviceToHost, m->stream) : hipSuccess;
        if (rc3 == SMH_OK && e3 == hipSuccess) e3 = hipStreamSynchronize(m->stream);
        (void)hipFree(d_odd);
        SMH_TRY(rc3);
        if (e3 != hipSuccess) return hip_fail(e3, "stream row-length statistics", __FILE__, __LINE__);
        m->stream_odd_rows = (uint64_t)h_odd;
    }
    SMH_HIP(hipStreamSynchronize(m->stream));
    return SMH_OK;
}

static void drop_stream_codes(smh_crs *m) {
    (void)hipFree(m->d_stream_cwin); (void)hipFree(m->d_stream_code); (void)hipFree(m->d_stream_len8); (void)hipFree(m->d_stream_tbase);
    m->d_stream_cwin = nullptr;
    m->d_stream_code = nullptr;
    m->d_stream_len8 = nullptr;
    m->d_stream_tbase = nullptr;
    m->stream_coded = false;
    m->stream_direct = m->stream_vdict = false;
    m->stream_vdict_xs = 0;
}

// does the VECTOR family run as K1r (LDS x-ring) for this matrix?
static int vector_uses_ring(smh_crs *m, bool *out) {
    *out = false;
    if (m->use_ring == 0 || m->n_rows == 0) return SMH_OK;
    SMH_TRY(ensure_ring_plan(m));
    // the pipelined body also wins without the ring (its global-gather phases), so it is the default
    // whenever its lane widths apply; mode 0 keeps the plain K1 kernel selectable
    *out = true;
    // 16-bit columns for the ring phases: a ring slot is `column mod 16384`, so the low half of a column is all a
    // ring phase reads -- 6 instead of 8 bytes per f32 entry from HBM.  One extra 2-byte-per-entry array, built once.
    int want = m->use_col16;
    if (const char *e = getenv("SMH_RING_COL16")) want = atoi(e) ? 1 : 0;  // tuning knob
    // (the banded plan cannot do without: its gathers take the ring slot from that array)
    const bool use16 = m->ring_bands == 4 || want == 1 || (want < 0 && m->ring_fraction >= 0.25);
    if (use16 && !m->d_col16 && m->nnz) {
        const size_t n_out = ((m->nnz + 3) & ~size_t(3)) + 4;
        SMH_HIP(hipMalloc((void **)&m->d_col16, n_out * sizeof(uint16_t)));
        if (m->ring_bands == 4) {
            SMH_HIP(hipMemsetAsync(m->d_col16, 0, n_out * sizeof(uint16_t), m->stream));
            SMH_TRY(launch_ring_band_codes(m->d_off, m->d_col, m->d_ring_win, m->n_rows, m->ring_entries / 4, m->d_col16, m->stream));
        } else {
            SMH_TRY(launch_narrow_columns(m->d_col, m->nnz, m->d_col16, n_out, m->stream));
        }
        SMH_HIP(hipStreamSynchronize(m->stream));
    } else if (!use16 && m->d_col16) {
        (void)hipFree(m->d_col16);
        m->d_col16 = nullptr;
    }
    return SMH_OK;
}

// K1s configuration of this handle
static int stream_rpt(const smh_crs *m) {
    // rows per thread: 512-row tiles measured no better than 256-row tiles (1.86 vs 1.80 ms on the 512^3
    // Laplacian), so one row per thread unless asked (SMH_STREAM_RPT=2, tuning knob)
    int want = m->stream_rows_per_thread;
    if (const char *e = getenv("SMH_STREAM_RPT")) want = atoi(e);
    return want == 2 && m->max_tile512_entries <= (uint32_t)kStreamCap ? 2 : 1;
}

// Can y = A x also leave the partial sums of x.y (CG's p.Ap) in its epilogue?  Only the K1s kernel does;
// returns the number of partials it would write (0: not fused -- run a separate dot).
// K1s configuration the STREAM variant runs with for this handle (builds the code tables on first use)
struct StreamCfg {
    int rpt = 1;
    bool single_pass = false;
    const uint16_t *code = nullptr;
    const uint32_t *cwin = nullptr;
    const uint8_t *len8 = nullptr;
    const uint32_t *tbase = nullptr;
    bool small = false; // no tile beyond kStreamCapSmall entries: the two-chunk body
    int xs = 0;         // ... and every tile's column intervals fit an LDS stage of x: 16-byte chunks per thread (2 or 4), 0 = no
    bool direct = false;  // `code` holds byte offsets into that stage, not column codes: only K1s XD (spmv_stream_xd.hip) reads it
    const void *dict = nullptr;  // ... with value-dictionary indices in their spare bits (K1s XD-V): the kernel does not read the values
    bool dict_high = false;      // ... in the high spare bits alone (few values)
};
// recode = false (every launch path): the configuration is READ from the handle -- the code array keeps the meaning it has.
// recode = true (the first build, smh_crs_prepare, the setters): the code array is rewritten in place when the choice between
// column codes and K1s XD's stage offsets has changed -- behind a device synchronisation, because a product enqueued on any
// stream may still read it; never under a stream capture (the callers say so in the header), and a graph captured before the
// change must be captured again.
static int stream_cfg(smh_crs *m, StreamCfg *c, bool recode = false) {
    *c = StreamCfg();
    // a 512-row tiling is only chosen when every such tile fits the LDS stage; the 256-row tiling takes
    // tiles of any density (loop-free body when the create-time statistic says that none overflows)
    c->rpt = stream_rpt(m);
    c->single_pass = m->have_stats && (c->rpt == 2 || m->max_tile_entries <= (uint32_t)kStreamCap);
    // 16-bit column codes when every tile's columns fall into <= 4 intervals of <= 16384 (stencils, bands)
    const char *c16_env = getenv("SMH_STREAM_C16");  // tuning knob: 0 = always the u32 columns
    // (single-pass tiles only: on dense multi-pass tiles -- banded C2 through K1s -- the decode costs more than
    // the bytes save, 0.83 vs 0.80 ms)
    if (!(c16_env && atoi(c16_env) == 0) && c->rpt == 1 && c->single_pass) {
        const bool first = !m->stream_coded;
        SMH_TRY(ensure_stream_codes(m));
        if (first) recode = true;  // (the build itself: allocations and synchronisations anyway)
        c->code = m->d_stream_code;
        c->cwin = m->d_stream_code ? m->d_stream_cwin : nullptr;
        static const bool l8_off = getenv("SMH_STREAM_L8") && atoi(getenv("SMH_STREAM_L8")) == 0;  // tuning knob
        if (c->cwin && !l8_off) { c->len8 = m->d_stream_len8; c->tbase = m->d_stream_tbase; }
        static const bool small_off = getenv("SMH_STREAM_SMALL") && atoi(getenv("SMH_STREAM_SMALL")) == 0;  // tuning knob
        c->small = !small_off && m->have_stats && m->max_tile_entries <= (uint32_t)kStreamCapSmall;
        static const bool xs_off = getenv("SMH_STREAM_XS") && atoi(getenv("SMH_STREAM_XS")) == 0;  // tuning knob
        // worth it from ~1 MB of x on (tools/dev/xs_threshold.py, profiles/r03_xs_threshold.log: 64^3 ... 320^3 cubes and 1000^2 / 2000^2
        // grids, -11 .. -20 % with the 2048-entry stage, -18 .. -40 % as K1s XD, both dtypes; below that the launch dominates).  The
        // 4096-entry stage pays on f32 only (grid planes 1024 wide, x of 17-34 MB: -7 .. -9 %; f64, LDS-limited to three blocks per
        // CU: +3 .. +10 %).  (Round 2 had 8 MB / 32 MB here: its inspector described a tile whose columns span < 16384 as ONE interval,
        // so a 64^3 .. 100^3 cube staged its whole span or nothing.)
        const bool forced = m->use_stream_xs == 1;
        const size_t x_bytes = m->n_cols * dtype_size(m->dtype);
        const bool on2 = forced || x_bytes >= ((size_t)1 << 20), on4 = forced || (x_bytes >= ((size_t)8 << 20) && m->dtype == SMH_F32);
        c->xs = (xs_off || m->use_stream_xs == 0 || !c->small || !c->len8) ? 0
                : (m->stream_xs_chunks <= 2u * kBlock && on2) ? 2
                : (m->stream_xs_chunks <= 4u * kBlock && on4) ? 4 : 0;
        // K1s XD: the code array as stage offsets.  The unskewed product stage it goes with collides on rows of even length, so
        // automatic = most rows odd (stencils with a diagonal)
        static const bool xd_off = getenv("SMH_STREAM_XD") && atoi(getenv("SMH_STREAM_XD")) == 0;  // tuning knob
        const bool want_direct = c->code && c->xs != 0 && !xd_off && m->use_stream_direct != 0 &&
                                 (m->use_stream_direct == 1 || 2 * m->stream_odd_rows >= (uint64_t)m->n_rows);
        // K1s XD-V: the matrix's distinct values in a dictionary, their indices in the codes' spare bits (spmv_stream_xd.hip) -- when
        // the values allow it (at most 32 bit patterns; 16 with the 4096-entry stage).  Looked at once per matrix, and again after
        // smh_crs_update_values.
        static const bool vd_off = getenv("SMH_STREAM_VDICT") && atoi(getenv("SMH_STREAM_VDICT")) == 0;  // tuning knob
        bool want_vdict = want_direct && !vd_off && m->use_stream_vdict != 0 && m->stream_dict_state != 0;
        bool dict_rebuilt = false;  // (the indices in the codes belong to the dictionary they were made with)
        if (recode && want_vdict && m->stream_dict_state < 0) {
            if (!m->d_stream_dict) SMH_HIP(hipMalloc(&m->d_stream_dict, 32 * dtype_size(m->dtype)));
            SMH_HIP(hipDeviceSynchronize());  // (a product in flight may still read the dictionary)
            uint32_t n_vals = 0;
            SMH_TRY(stream_value_dict(m->dtype, m->d_val, m->nnz, m->d_stream_dict, &n_vals, m->stream));
            m->stream_dict_n = n_vals;
            m->stream_dict_state = n_vals ? 1 : 0;
            dict_rebuilt = true;
        }
        want_vdict = want_vdict && m->stream_dict_state == 1 && m->stream_dict_n <= stream_value_dict_capacity(c->xs);
        const bool form_ok = want_direct == m->stream_direct && want_vdict == m->stream_vdict && (!want_vdict || m->stream_vdict_xs == c->xs) &&
                             !(dict_rebuilt && (want_vdict || m->stream_vdict));
        if (recode && c->code && !form_ok) {
            SMH_HIP(hipDeviceSynchronize());  // (a product enqueued on any stream may still read the array)
            if (want_direct)
                SMH_TRY(launch_stream_stage_codes(m->d_off, m->d_col, m->d_stream_cwin, m->n_rows, (uint32_t)dtype_size(m->dtype), m->d_stream_code, m->stream));
            else
                SMH_TRY(launch_stream_codes(m->d_off, m->d_col, m->d_stream_cwin, m->n_rows, m->d_stream_code, m->stream));
            if (want_vdict)
                SMH_TRY(launch_stream_value_codes(m->dtype, m->d_val, m->nnz, m->d_stream_dict, m->stream_dict_n, c->xs, m->d_stream_code, m->stream));
            SMH_HIP(hipStreamSynchronize(m->stream));
            m->stream_direct = want_direct;
            m->stream_vdict = want_vdict;
            m->stream_vdict_xs = want_vdict ? c->xs : 0;
        }
        // (stage offsets without a stage -- xs == 0 after a setter that was not followed by a prepare cannot happen: the setters
        // recode; an x too short / misaligned for the stage is handled per call in stream_launch)
        c->direct = c->code && m->stream_direct;
        c->dict = c->direct && m->stream_vdict ? m->d_stream_dict : nullptr;
        c->dict_high = c->dict && stream_value_dict_high(m->dtype, m->stream_dict_n, m->stream_vdict_xs);
    }
    return SMH_OK;
}

// one K1s launch over the tiles [t0, t1) with the configuration `c`
static int stream_launch(smh_crs *m, const StreamCfg &c, const void *x, size_t x_len, void *y, hipStream_t s, void *dot_partials,
                         const void *dot_lhs, uint64_t t0, uint64_t t1) {
    // the staged chunks are groups of 4 entries of x fetched with 16-byte loads (f32: one load, f64: two, entries [g, g+2) and
    // [g+2, g+4)); with x itself 16-byte aligned a load that holds at least one valid entry may reach past x_len but never past the
    // 16-byte block (hence page, hence allocation granule) its valid entry lies in.  f32: the last chunk holds a valid entry when
    // x_len rounded up to 4 reaches stream_xs_end; f64: its SECOND load starts at stream_xs_end - 2 and must hold a valid entry
    // too (x_len >= stream_xs_end - 1), else the gathers stay global
    const bool xs_ok = ((m->dtype == SMH_F64 ? x_len + 1 : ((x_len + 3) & ~(size_t)3)) >= (size_t)m->stream_xs_end) &&
                       (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
    if (c.direct) {
        if (xs_ok && c.xs)
            return launch_spmv_stream_xd(m->dtype, m->d_val, x, y, m->n_rows, dot_partials, c.code, c.cwin, c.len8, c.tbase, dot_lhs, s,
                                         c.xs, t0, t1, c.dict, c.dict_high);
        // stage offsets mean nothing without the stage: this call streams the u32 columns (same arithmetic, same order)
        return launch_spmv_stream(m->dtype, m->d_off, m->d_col, m->d_val, x, y, m->n_rows, m->nnz, m->owns, c.rpt, c.single_pass,
                                  dot_partials, nullptr, nullptr, nullptr, nullptr, dot_lhs, s, false, 0, t0, t1);
    }
    return launch_spmv_stream(m->dtype, m->d_off, m->d_col, m->d_val, x, y, m->n_rows, m->nnz, m->owns, c.rpt, c.single_pass,
                              dot_partials, c.code, c.cwin, c.len8, c.tbase, dot_lhs, s, c.small, xs_ok ? c.xs : 0, t0, t1);
}

// any_lhs: the dot is taken with a vector of its own (n_rows entries; SparseMatrix::inner_prod) instead of x itself, so
// the matrix need not be square
size_t spmv_fused_dot_partials(smh_crs *m, size_t x_len, int variant, bool any_lhs) {
    if (const char *e = getenv("SMH_CG_FUSED_DOT")) {  // tuning knob: 0 = always the separate dot
        if (atoi(e) == 0) return 0;
    }
    if (resolve_variant(m, variant) != SMH_SPMV_STREAM) return 0;
    if (!any_lhs && (m->n_rows != m->n_cols || x_len < m->n_rows)) return 0;
    StreamCfg c;
    if (stream_cfg(m, &c) != SMH_OK) return 0;
    return stream_tiles(m->n_rows, c.rpt);
}

// enqueue y = A x on stream s (device pointers); dot_partials (optional, K1s only): see above
int spmv_enqueue(smh_crs *m, const void *x, size_t x_len, void *y, int variant, hipStream_t s, void *dot_partials, const void *dot_lhs) {
    if (m->nnz > 0 && (size_t)m->max_col >= x_len)
        return fail(SMH_ERR_INDEX_RANGE, "index out of bounds: the len is %zu but the index is %u", x_len, m->max_col);
    const int v = resolve_variant(m, variant);
    switch (v) {
        case SMH_SPMV_VECTOR: {
            bool ring = false;
            SMH_TRY(vector_uses_ring(m, &ring));
            if (ring)
                // owned arrays are padded to a multiple of 4 entries; borrowed ones may end inside a 16-B chunk
                return launch_spmv_ring2(m->dtype, auto_lanes(m), auto_chunks(m), m->d_off, m->d_col, m->d_col16, m->d_val, x, y, m->n_rows,
                                         m->nnz, m->owns || m->nnz % 4 == 0, m->ring_blocks, m->d_phase_ptr, m->d_phases,
                                         m->ring_entries, m->ring_bands, s);
            return launch_spmv_vector(m->dtype, auto_lanes(m), m->d_off, m->d_col, m->d_val, x, y, m->n_rows, m->nnz, s);
        }
        case SMH_SPMV_SEQ:
            return launch_spmv_seq(m->dtype, m->d_off, m->d_col, m->d_val, x, y, m->n_rows, s);
        case SMH_SPMV_STREAM: {
            StreamCfg c;
            SMH_TRY(stream_cfg(m, &c));
            return stream_launch(m, c, x, x_len, y, s, dot_partials, dot_lhs, 0, ~uint64_t(0));
        }
        case SMH_SPMV_COLSPLIT: {
            {
                const int rc = ensure_split(m);
                // AUTO chose this plan and its lazy build failed (scratch out of memory, ...): the plan is marked refused
                // (split_built && !split_ok) and AUTO resolves again within this call -- K2c / K2f / K1 can still serve the product
                if (rc != SMH_OK && variant == SMH_SPMV_AUTO && rc != SMH_ERR_INDEX_RANGE) {
                    m->split_built = true;
                    m->split_ok = false;
                    g_err[0] = 0;
                    return spmv_enqueue(m, x, x_len, y, variant, s, dot_partials, dot_lhs);
                }
                SMH_TRY(rc);
            }
            if (m->split_ok) {
                // The two parts lean on different resources (LONG: the L2 gather path; SHORT: HBM streams and latency), which
                // suggests running LONG on a stream of its own beside SHORT (fork and join by events).  Measured on C3, one box:
                // 2.90 ms overlapped against 2.79 ms back to back -- K2f sizes its rounds to the whole chip and the blocks of
                // x of the two parts evict each other -- so it is a knob, off by default (SMH_COLSPLIT_OVERLAP=1)
                static const bool overlap = getenv("SMH_COLSPLIT_OVERLAP") && atoi(getenv("SMH_COLSPLIT_OVERLAP")) != 0;
                hipStream_t sl = overlap ? m->split_stream : s;
                if (overlap) {
                    SMH_HIP(hipEventRecord(m->split_fork, s));
                    SMH_HIP(hipStreamWaitEvent(sl, m->split_fork, 0));
                }
                SMH_TRY(spmv_enqueue(m->split_long, x, x_len, m->d_split_y, SMH_SPMV_AUTO, sl));  // the long rows, compacted
                if (overlap) SMH_HIP(hipEventRecord(m->split_join, sl));
                SMH_TRY(spmv_enqueue(m->split_short, x, x_len, y, SMH_SPMV_AUTO, s));             // every row (0 for the long ones)
                if (overlap) SMH_HIP(hipStreamWaitEvent(s, m->split_join, 0));
                return launch_split_scatter(m->dtype, m->d_split_rows, m->d_split_y, m->split_n_long, y, s);
            }
            SMH_TRY(ensure_colblock(m));  // not worth splitting: the per-block launches
            for (size_t b = 0; b < m->cb_blocks; ++b)
                SMH_TRY(launch_spmv_stream_block(m->dtype, m->d_cb_off + b * (m->n_rows + 1), m->d_cb_col, m->d_cb_val, x, y,
                                                 m->n_rows, m->nnz, m->cb_rpt, m->cb_single_pass, b > 0, s));
            return SMH_OK;
        }
        case SMH_SPMV_COLFUSED: {
            SMH_TRY(ensure_colfused(m));
            if (m->cf_ok)
                return launch_spmv_colfused(m->dtype, m->cf_rt, m->d_cf_tile_row, m->cf_tiles, m->d_cf_seg, m->d_cf_cnt, m->d_cf_col, m->d_cf_val,
                                            x, y, m->n_rows, m->nnz, (uint32_t)m->cf_blocks, m->device, s);
        }
        [[fallthrough]];  // a (row, block) pair with more than 255 entries: the per-block launches
        case SMH_SPMV_COLBLOCK: {
            SMH_TRY(ensure_colblock(m));
            for (size_t b = 0; b < m->cb_blocks; ++b)
                SMH_TRY(launch_spmv_stream_block(m->dtype, m->d_cb_off + b * (m->n_rows + 1), m->d_cb_col, m->d_cb_val, x, y,
                                                 m->n_rows, m->nnz, m->cb_rpt, m->cb_single_pass, b > 0, s));
            return SMH_OK;
        }
        case SMH_SPMV_TILED: {
            const int rc = tiled_build(m);
            // as above: a failed lazy build (K2t needs ~5 x 4 B x nnz of scratch, a copy of the entries and a product buffer)
            // leaves t2_built && !t2_ok, which AUTO's rule reads as "does not fit" -- resolve again within this call
            if (rc != SMH_OK && variant == SMH_SPMV_AUTO && rc != SMH_ERR_INDEX_RANGE && m->t2_built && !m->t2_ok) {
                g_err[0] = 0;
                return spmv_enqueue(m, x, x_len, y, variant, s, dot_partials, dot_lhs);
            }
            SMH_TRY(rc);
            return launch_spmv_tiled(m, x, x_len, y, s);
        }
        case SMH_SPMV_MERGE:
            SMH_TRY(ensure_merge_ws(m));
            return launch_spmv_merge(m->dtype, m->d_off, m->d_col, m->d_val, x, y, m->n_rows, m->nnz, m->n_tiles,
                                     m->d_tile_row, m->d_tile_nz, m->d_carry_row, m->d_carry_val, s);
        default:
            return fail(SMH_ERR_INVALID, "unknown SpMV variant %d", variant);
    }
}

// ---- products of a run of rows (par.hip: a block's boundary rows before / after its interior ones) --------------------------
// The kernels whose launches decompose by rows WITHOUT changing any row's arithmetic: K1s (256-row tiles; bit-exact anyway) and
// K1r (the plan's row ranges; a row's lanes, chunks and order do not depend on which workgroup takes it).  *gran_out = the row
// granularity a run must respect (0: this handle's kernel for `variant` cannot be launched by parts).
int spmv_rows_granularity(smh_crs *m, int variant, size_t *gran_out) {
    *gran_out = 0;
    if (m->n_rows == 0) return SMH_OK;
    const int v = resolve_variant(m, variant);
    if (v == SMH_SPMV_STREAM) {
        StreamCfg c;
        SMH_TRY(stream_cfg(m, &c));
        *gran_out = (size_t)kStreamRows * (size_t)c.rpt;
    } else if (v == SMH_SPMV_VECTOR) {
        bool ring = false;
        SMH_TRY(vector_uses_ring(m, &ring));
        if (ring && m->ring_blocks) {
            const size_t n_tiles = (m->n_rows + 63) / 64;
            *gran_out = ((n_tiles + m->ring_blocks - 1) / m->ring_blocks) * 64;  // build_ring_plan: tiles per row range
        }
    }
    return SMH_OK;
}

// y[row0, row1) = (A x)[row0, row1): row0 a multiple of the granularity, row1 too or == n_rows.  Same kernels, same arithmetic as
// the whole product.  dot_partials: as spmv_enqueue (K1s only; the tiles of the run write their partials, the others are left alone)
int spmv_enqueue_rows(smh_crs *m, const void *x, size_t x_len, void *y, int variant, hipStream_t s, size_t row0, size_t row1,
                      void *dot_partials, const void *dot_lhs) {
    if (m->nnz > 0 && (size_t)m->max_col >= x_len)
        return fail(SMH_ERR_INDEX_RANGE, "index out of bounds: the len is %zu but the index is %u", x_len, m->max_col);
    if (row1 > m->n_rows) row1 = m->n_rows;
    if (row0 >= row1) return SMH_OK;
    size_t gran = 0;
    SMH_TRY(spmv_rows_granularity(m, variant, &gran));
    if (gran == 0 || row0 % gran || (row1 % gran && row1 != m->n_rows))
        return fail(SMH_ERR_INVALID, "rows [%zu, %zu) cannot be launched on their own (granularity %zu)", row0, row1, gran);
    const int v = resolve_variant(m, variant);
    if (v == SMH_SPMV_STREAM) {
        StreamCfg c;
        SMH_TRY(stream_cfg(m, &c));
        return stream_launch(m, c, x, x_len, y, s, dot_partials, dot_lhs, row0 / gran, (row1 + gran - 1) / gran);
    }
    if (dot_partials) return fail(SMH_ERR_INVALID, "a product by parts with the dot epilogue needs the CSR-stream kernel");
    return launch_spmv_ring2(m->dtype, auto_lanes(m), auto_chunks(m), m->d_off, m->d_col, m->d_col16, m->d_val, x, y, m->n_rows, m->nnz,
                             m->owns || m->nnz % 4 == 0, m->ring_blocks, m->d_phase_ptr, m->d_phases, m->ring_entries, m->ring_bands, s, nullptr,
                             (unsigned)(row0 / gran), (unsigned)((row1 + gran - 1) / gran));
}

// The same for a SHORT run of rows (a partition block's boundary rows: a few thousand), any row0 / row1.  One ring workgroup walks its
// ~6500 rows in ~100 us whatever else the chip does -- the ring kernel gets its rate from 512 of them at once -- so two boundary
// launches were 200 us of a rank's 340 us step (profiles/r04_par_boundary_rows_k1.log).  The plain lane-group kernel K1 (same lanes,
// same chunk grid and lane layout, same order of FMAs; x through L1 / L2 instead of the LDS ring) gives the ring kernel's bits in
// both value types (tests/test_ring_gpu.py::test_k1_is_k1r_bit_for_bit, and every overlap-on / overlap-off comparison of the
// partition tests), and a short run is a hundred small workgroups: microseconds.  Everything else goes the way of spmv_enqueue_rows.
int spmv_enqueue_rows_short(smh_crs *m, const void *x, size_t x_len, void *y, int variant, hipStream_t s, size_t row0, size_t row1) {
    static const bool off = getenv("SMH_PAR_BOUNDARY_K1") && atoi(getenv("SMH_PAR_BOUNDARY_K1")) == 0;  // tuning knob
    if (row1 > m->n_rows) row1 = m->n_rows;
    if (row0 >= row1) return SMH_OK;
    if (!off && resolve_variant(m, variant) == SMH_SPMV_VECTOR && auto_lanes(m) >= 4 && row1 - row0 <= (size_t)1 << 20) {
        bool ring = false;
        SMH_TRY(vector_uses_ring(m, &ring));
        if (ring) {
            if (m->nnz > 0 && (size_t)m->max_col >= x_len)
                return fail(SMH_ERR_INDEX_RANGE, "index out of bounds: the len is %zu but the index is %u", x_len, m->max_col);
            return launch_spmv_vector(m->dtype, auto_lanes(m), m->d_off + row0, m->d_col, m->d_val, x, (char *)y + row0 * dtype_size(m->dtype), row1 - row0,
                                      m->nnz, s);
        }
    }
    return spmv_enqueue_rows(m, x, x_len, y, variant, s, row0, row1, nullptr, nullptr);
}

static int finish_create_inner(smh_crs *m, int validate);
static int finish_create(smh_crs *m, int validate) {
    // (what the create-time inspection costs: the statistics passes and, for matrices AUTO needs it for, the K1r inspector)
    const auto t0 = std::chrono::steady_clock::now();
    const long long b0 = pool_thread_net_bytes();
    const int rc = finish_create_inner(m, validate);
    m->create_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    m->create_bytes = pool_thread_net_bytes() - b0;
    return rc;
}
static int finish_create_inner(smh_crs *m, int validate) {
    SMH_HIP(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    if (m->n_rows > 0) {
        CrsStats *d_st = nullptr, h_st;
        SMH_HIP(hipMalloc((void **)&d_st, sizeof(CrsStats)));
        int rc = launch_crs_stats(m->d_off, m->d_col, m->n_rows, m->nnz, d_st, m->stream);
        if (rc == SMH_OK) {
            hipError_t e = hipMemcpyAsync(&h_st, d_st, sizeof h_st, hipMemcpyDeviceToHost, m->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
            // (reuse the first word of the scratch for the K1s tile statistic)
            if (e == hipSuccess && !(h_st.bad & 1u)) {
                uint32_t mt = 0;
                rc = launch_stream_max_tile(m->d_off, m->n_rows, kStreamRows, &d_st->max_row_len, m->stream);
                if (rc == SMH_OK) e = hipMemcpyAsync(&mt, &d_st->max_row_len, sizeof mt, hipMemcpyDeviceToHost, m->stream);
                if (rc == SMH_OK && e == hipSuccess) e = hipStreamSynchronize(m->stream);
                m->max_tile_entries = mt;
                if (rc == SMH_OK && e == hipSuccess)
                    rc = launch_stream_max_tile(m->d_off, m->n_rows, 2 * kStreamRows, &d_st->max_row_len, m->stream);
                if (rc == SMH_OK && e == hipSuccess) e = hipMemcpyAsync(&mt, &d_st->max_row_len, sizeof mt, hipMemcpyDeviceToHost, m->stream);
                if (rc == SMH_OK && e == hipSuccess) e = hipStreamSynchronize(m->stream);
                m->max_tile512_entries = mt;
            }
            if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
            if (e != hipSuccess) rc = hip_fail(e, "crs stats readback", __FILE__, __LINE__);
        }
        (void)hipFree(d_st);
        SMH_TRY(rc);
        m->max_row_len = h_st.max_row_len;
        m->max_col = h_st.max_col;
        m->min_col = m->nnz ? ~h_st.min_col_inv : 0u;
        m->have_stats = true;
        // a malformed row structure would send the kernels out of bounds: always refused
        if (h_st.bad & 1u) return fail(SMH_ERR_INVALID, "offset_rows is not monotone non-decreasing");
        if (h_st.bad & 2u) return fail(SMH_ERR_INVALID, "offset_rows[0] != 0");
        if (h_st.bad & 4u) return fail(SMH_ERR_INVALID, "offset_rows[n_rows] != nnz");
        if (validate && m->nnz > 0 && (size_t)h_st.max_col >= m->n_cols)
            return fail(SMH_ERR_INDEX_RANGE, "column index %u >= n_cols %zu", h_st.max_col, m->n_cols);
        // x larger than the L2s: take the locality statistic AUTO needs (one pass over columns[]; it is the K1r
        // inspector, so its plan is ready too)
        // ... and rows long enough for the lane-group kernels: AUTO wants to know whether their columns fit the ring
        const bool lane_group_rows = m->n_rows && m->nnz > 8 * m->n_rows;  // (mean row > 8)
        if (m->n_cols * dtype_size(m->dtype) >= kColblockMinXBytes || lane_group_rows)
            SMH_TRY(ensure_ring_plan(m, lane_group_rows));
    }
    return SMH_OK;
}

static int check_create_args(int dtype, size_t n_rows, size_t nnz, const void *off, const void *col, const void *val,
                             smh_crs **out) {
    if (!out) return fail(SMH_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!valid_dtype(dtype)) return fail(SMH_ERR_INVALID, "dtype must be SMH_F32 or SMH_F64");
    // Index = u32: UNSET = u32::MAX is reserved, entry count must stay below it (sparsemat_crs.rs:82-84)
    if (nnz >= 0xFFFFFFFFull) return fail(SMH_ERR_CAPACITY, "Maximum number of %u entries reached", 0xFFFFFFFFu);
    if (n_rows >= 0xFFFFFFFFull) return fail(SMH_ERR_CAPACITY, "n_rows does not fit the u32 index type");
    if (n_rows > 0 && !off) return fail(SMH_ERR_INVALID, "offset_rows is NULL");
    if (nnz > 0 && (!col || !val)) return fail(SMH_ERR_INVALID, "columns/values is NULL");
    return SMH_OK;
}

static int vec_check_pair(const smh_vec *x, const smh_vec *y) {
    if (!x || !y) return fail(SMH_ERR_INVALID, "NULL vector handle");
    if (x->dtype != y->dtype) return fail(SMH_ERR_INVALID, "vector dtype mismatch");
    return SMH_OK;
}

// scratch for reductions of the vector API: per thread, per device
struct ReduceScratch { void *d = nullptr; int device = -1; };
static thread_local ReduceScratch g_red;
static int reduce_scratch(void **out) {
    const int dev = current_device();
    if (!g_red.d || g_red.device != dev) {
        // (a scratch left on another device is intentionally leaked: handles are device-bound)
        SMH_HIP(hipMalloc(&g_red.d, (kReducePartials + 8) * sizeof(double)));
        g_red.device = dev;
    }
    *out = g_red.d;
    return SMH_OK;
}

}  // namespace smh

using namespace smh;

extern "C" {

int smh_abi_version(void) { return SMH_ABI_VERSION; }

const char *smh_last_error(void) { return g_err; }

const char *smh_status_string(int status) {
    switch (status) {
        case SMH_OK: return "ok";
        case SMH_ERR_DIM_MISMATCH: return "Dimension mismatch";
        case SMH_ERR_NOT_SQUARE: return "Matrix is not symmetric";
        case SMH_ERR_INDEX_RANGE: return "index out of bounds";
        case SMH_ERR_INVALID: return "invalid argument";
        case SMH_ERR_HIP: return "HIP runtime error";
        case SMH_ERR_OOM: return "out of device memory";
        case SMH_ERR_NO_DEVICE: return "no HIP device (no CPU fallback)";
        case SMH_ERR_CAPACITY: return "Maximum number of entries reached";
        case SMH_ERR_COMM: return "RCCL error";
        default: return "unknown status";
    }
}

int smh_device_count(int *count_out) {
    if (!count_out) return fail(SMH_ERR_INVALID, "count_out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count_out = n;
    return SMH_OK;
}

int smh_set_device(int device) {
    SMH_TRY(require_device());
    SMH_HIP(hipSetDevice(device));
    return SMH_OK;
}

int smh_device_synchronize(void) {
    SMH_TRY(require_device());
    SMH_HIP(hipDeviceSynchronize());
    return SMH_OK;
}

// ---- SparseMatCRS ------------------------------------------------------------------------------------
int smh_crs_create(smh_dtype dtype, size_t n_rows, size_t n_cols, size_t nnz, const uint32_t *offset_rows,
                   const uint32_t *columns, const void *values, int validate, smh_crs **out) {
    SMH_TRY(check_create_args(dtype, n_rows, nnz, offset_rows, columns, values, out));
    SMH_TRY(require_device());
    smh_crs *m = new (std::nothrow) smh_crs();
    if (!m) return fail(SMH_ERR_OOM, "host allocation failed");
    m->dtype = dtype; m->n_rows = n_rows; m->n_cols = n_cols; m->nnz = nnz; m->owns = true;
    m->device = current_device();
    const size_t vs = dtype_size(dtype);
    int rc = SMH_OK;
    auto go = [&]() -> int {
        SMH_HIP(hipMalloc((void **)&m->d_off, (n_rows + 1) * sizeof(uint32_t)));
        SMH_HIP(hipMalloc((void **)&m->d_col, (nnz + 4) * sizeof(uint32_t)));
        SMH_HIP(hipMalloc(&m->d_val, (nnz + 4) * vs));
        if (n_rows > 0) SMH_HIP(hipMemcpy(m->d_off, offset_rows, (n_rows + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
        else SMH_HIP(hipMemset(m->d_off, 0, sizeof(uint32_t)));
        if (nnz > 0) {
            SMH_HIP(hipMemcpy(m->d_col, columns, nnz * sizeof(uint32_t), hipMemcpyHostToDevice));
            SMH_HIP(hipMemcpy(m->d_val, values, nnz * vs, hipMemcpyHostToDevice));
        }
        return finish_create(m, validate);
    };
    rc = go();
    if (rc != SMH_OK) { char keep[512]; strncpy(keep, g_err, sizeof keep); keep[sizeof keep - 1] = 0; smh_crs_destroy(m); strncpy(g_err, keep, sizeof g_err); return rc; }
    *out = m;
    return SMH_OK;
}

int smh_crs_create_dev(smh_dtype dtype, size_t n_rows, size_t n_cols, size_t nnz, const uint32_t *offset_rows_dev,
                       const uint32_t *columns_dev, const void *values_dev, int validate, smh_crs **out) {
    SMH_TRY(check_create_args(dtype, n_rows, nnz, offset_rows_dev, columns_dev, values_dev, out));
    SMH_TRY(require_device());
    if (((uintptr_t)columns_dev & 15u) || ((uintptr_t)values_dev & 15u))
        return fail(SMH_ERR_INVALID, "columns/values device pointers must be 16-byte aligned");
    smh_crs *m = new (std::nothrow) smh_crs();
    if (!m) return fail(SMH_ERR_OOM, "host allocation failed");
    m->dtype = dtype; m->n_rows = n_rows; m->n_cols = n_cols; m->nnz = nnz; m->owns = false;
    m->device = current_device();
    m->d_off = const_cast<uint32_t *>(offset_rows_dev);
    m->d_col = const_cast<uint32_t *>(columns_dev);
    m->d_val = const_cast<void *>(values_dev);
    int rc = finish_create(m, validate);
    if (rc != SMH_OK) { char keep[512]; strncpy(keep, g_err, sizeof keep); keep[sizeof keep - 1] = 0; smh_crs_destroy(m); strncpy(g_err, keep, sizeof g_err); return rc; }
    *out = m;
    return SMH_OK;
}

// add_to / set stream -> CRS (assemble.hip).  `on_device`: the arrays are device pointers.
// into_crs: the stream is replayed on a SparseMatCRS instead of a SparseMatIndexList + to_crs(): rows come out in reverse order
// of first appearance (push prepends, sparsemat_crs.rs:85-87) and the container's first-push quirk applies (:75-81: the first push
// leaves n_rows == 0, so the second operation never finds an entry):
//   * second row <  first row: Vec::resize truncates offset_rows and the first entry is orphaned -- it stays in columns / values
//     (and in n_cols) but no row reaches it; the result is the replay of operations 1.. alone.  Orphans are not materialised here;
//   * second (row, column) == first: the first operation keeps an entry of its own, the oldest of its row (= last in storage);
//   * a single operation: no rows at all (n_rows stays 0), one orphan.
static int assemble_common(smh_dtype dtype, size_t n_ops, const uint32_t *rows, const uint32_t *cols, const void *values,
                           const uint8_t *ops, bool on_device, bool into_crs, smh_crs **out, bool transposing = false) {
    // transposing: every operation is `set` (ops unused) and repeats of a (row, column) pair are neighbours in the row's list
    if (!out) return fail(SMH_ERR_INVALID, "NULL out pointer");
    if (dtype != SMH_F32 && dtype != SMH_F64) return fail(SMH_ERR_INVALID, "unknown dtype %d", (int)dtype);
    if (n_ops && (!rows || !cols || !values)) return fail(SMH_ERR_INVALID, "NULL operation array");
    if (n_ops >= 0xFFFFFFFFull) return fail(SMH_ERR_CAPACITY, "Maximum number of %u entries reached", 0xFFFFFFFFu);
    SMH_TRY(require_device());
    smh_crs *m = new (std::nothrow) smh_crs();
    if (!m) return fail(SMH_ERR_OOM, "host allocation failed");
    m->dtype = dtype; m->owns = true;
    m->device = current_device();
    const size_t vs = dtype_size(dtype);
    void *d_in[4] = {nullptr, nullptr, nullptr, nullptr};
    auto no_rows = [&](size_t n_cols) -> int {  // SparseMatCRS::new() (sparsemat_crs.rs:47-49): no rows at all
        SMH_HIP(hipMalloc((void **)&m->d_off, sizeof(uint32_t)));
        SMH_HIP(hipMemset(m->d_off, 0, sizeof(uint32_t)));
        SMH_HIP(hipMalloc((void **)&m->d_col, 4 * sizeof(uint32_t)));
        SMH_HIP(hipMalloc(&m->d_val, 4 * vs));
        m->n_cols = n_cols;
        return finish_create(m, 0);
    };
    auto go = [&]() -> int {
        if (n_ops == 0) return no_rows(0);
        // the first two operations decide the SparseMatCRS quirk
        size_t skip = 0, min_cols = 0;
        bool twin = false;
        uint32_t r01[2] = {0, 0}, c01[2] = {0, 0};
        double v0 = 0.0;  // (holds an f32 or an f64 bit pattern)
        uint8_t op0 = 0;
        if (into_crs) {
            const size_t k = n_ops < 2 ? n_ops : 2;
            const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost;
            SMH_HIP(hipMemcpy(r01, rows, k * sizeof(uint32_t), kind));
            SMH_HIP(hipMemcpy(c01, cols, k * sizeof(uint32_t), kind));
            SMH_HIP(hipMemcpy(&v0, values, vs, kind));
            if (transposing) op0 = 1;
            else if (ops) SMH_HIP(hipMemcpy(&op0, ops, 1, kind));
            if (n_ops == 1) { m->orphans = 1; return no_rows((size_t)c01[0] + 1); }
            if (r01[1] < r01[0]) { skip = 1; min_cols = (size_t)c01[0] + 1; m->orphans = 1; }
            else if (r01[1] == r01[0] && c01[1] == c01[0]) { skip = 1; twin = true; }
        }
        const uint32_t *d_rows = rows, *d_cols = cols;
        const void *d_vals = values;
        const uint8_t *d_ops = ops;
        if (!on_device) {
            SMH_HIP(hipMalloc(&d_in[0], n_ops * sizeof(uint32_t)));
            SMH_HIP(hipMalloc(&d_in[1], n_ops * sizeof(uint32_t)));
            SMH_HIP(hipMalloc(&d_in[2], n_ops * vs));
            SMH_HIP(hipMemcpy(d_in[0], rows, n_ops * sizeof(uint32_t), hipMemcpyHostToDevice));
            SMH_HIP(hipMemcpy(d_in[1], cols, n_ops * sizeof(uint32_t), hipMemcpyHostToDevice));
            SMH_HIP(hipMemcpy(d_in[2], values, n_ops * vs, hipMemcpyHostToDevice));
            if (ops) {
                SMH_HIP(hipMalloc(&d_in[3], n_ops));
                SMH_HIP(hipMemcpy(d_in[3], ops, n_ops, hipMemcpyHostToDevice));
            }
            d_rows = (const uint32_t *)d_in[0]; d_cols = (const uint32_t *)d_in[1]; d_vals = d_in[2]; d_ops = (const uint8_t *)d_in[3];
        }
        SMH_TRY(assemble_triplets(dtype, n_ops - skip, d_rows + skip, d_cols + skip, (const char *)d_vals + skip * vs,
                                  d_ops ? d_ops + skip : nullptr, into_crs, transposing, transposing, &m->n_rows, &m->n_cols, &m->nnz, &m->d_off, &m->d_col,
                                  &m->d_val, nullptr));
        if (min_cols > m->n_cols) m->n_cols = min_cols;
        if (twin) {  // the first operation's own entry: push(i, j, zero) then `=` or `+=` (sparsematrix.rs:226-233)
            if (!op0) {
                if (dtype == SMH_F64) { double v; memcpy(&v, &v0, 8); v = 0.0 + v; memcpy(&v0, &v, 8); }
                else { float v; memcpy(&v, &v0, 4); v = 0.0f + v; memcpy(&v0, &v, 4); }
            }
            SMH_TRY(append_to_row(dtype, m->d_off, &m->d_col, &m->d_val, m->n_rows, &m->nnz, r01[0], c01[0], &v0, nullptr));
        }
        return finish_create(m, 0);
    };
    const int rc = go();
    for (void *p : d_in) (void)hipFree(p);
    if (rc != SMH_OK) { char keep[512]; strncpy(keep, g_err, sizeof keep); keep[sizeof keep - 1] = 0; smh_crs_destroy(m); strncpy(g_err, keep, sizeof g_err); return rc; }
    *out = m;
    return SMH_OK;
}

int smh_crs_assemble(smh_dtype dtype, size_t n_ops, const uint32_t *rows, const uint32_t *cols, const void *values,
                     const uint8_t *ops, smh_crs **out) {
    return assemble_common(dtype, n_ops, rows, cols, values, ops, false, false, out);
}

int smh_crs_assemble_dev(smh_dtype dtype, size_t n_ops, const uint32_t *rows_dev, const uint32_t *cols_dev,
                         const void *values_dev, const uint8_t *ops_dev, smh_crs **out) {
    return assemble_common(dtype, n_ops, rows_dev, cols_dev, values_dev, ops_dev, true, false, out);
}

int smh_crs_replay(smh_dtype dtype, size_t n_ops, const uint32_t *rows, const uint32_t *cols, const void *values,
                   const uint8_t *ops, smh_crs **out) {
    return assemble_common(dtype, n_ops, rows, cols, values, ops, false, true, out);
}

int smh_crs_replay_dev(smh_dtype dtype, size_t n_ops, const uint32_t *rows_dev, const uint32_t *cols_dev,
                       const void *values_dev, const uint8_t *ops_dev, smh_crs **out) {
    return assemble_common(dtype, n_ops, rows_dev, cols_dev, values_dev, ops_dev, true, true, out);
}

static thread_local int g_transpose_route = 0;
int smh_last_transpose_route(void) { return g_transpose_route; }

// SparseMatrix::transpose (sparsematrix.rs:174-184) for Self = SparseMatCRS: `ret.set(j, i, val)` for every entry in row-major
// storage order into a fresh SparseMatCRS -- the replay above with rows = the columns array (borrowed), columns = the row of
// every entry, all operations `set`.
int smh_crs_transpose(const smh_crs *a, smh_crs **out) {
    if (!a || !out) return fail(SMH_ERR_INVALID, "NULL argument");
    SMH_HIP(hipStreamSynchronize(a->stream));
    if (a->nnz == 0) return assemble_common((smh_dtype)a->dtype, 0, nullptr, nullptr, nullptr, nullptr, true, true, out);
    g_transpose_route = 0;
    // matrices with local structure: two bucketed passes instead of the device-wide sort (transpose_bucket.hip).  The container's first-push quirks (second target row below the first: the first entry is
    // orphaned; a single operation) and repeated (row, column) pairs stay with the general route.
    const bool bucketed_allowed = !(getenv("SMH_TRANSPOSE_BUCKETED") && atoi(getenv("SMH_TRANSPOSE_BUCKETED")) == 0);
    if (bucketed_allowed && a->nnz >= 2 && a->have_stats) {
        uint32_t c01[2] = {0, 0};
        SMH_HIP(hipMemcpy(c01, a->d_col, sizeof c01, hipMemcpyDeviceToHost));
        if (c01[1] >= c01[0]) {
            uint32_t *t_off = nullptr, *t_col = nullptr;
            void *t_val = nullptr;
            size_t t_rows = 0, t_cols = 0;
            bool done = false;
            SMH_TRY(transpose_bucketed(a->dtype, a->d_off, a->d_col, a->d_val, a->n_rows, a->nnz, a->max_col, &t_off, &t_col, &t_val, &t_rows, &t_cols, &done,
                                       nullptr));
            if (done) {
                smh_crs *m = new (std::nothrow) smh_crs();
                if (!m) { (void)hipFree(t_off); (void)hipFree(t_col); (void)hipFree(t_val); return fail(SMH_ERR_OOM, "host allocation failed"); }
                m->dtype = a->dtype; m->owns = true;
                m->device = current_device();
                m->d_off = t_off; m->d_col = t_col; m->d_val = t_val;
                m->n_rows = t_rows; m->n_cols = t_cols; m->nnz = a->nnz;
                const int rc = finish_create(m, 0);
                if (rc != SMH_OK) { char keep[512]; strncpy(keep, g_err, sizeof keep); keep[sizeof keep - 1] = 0; smh_crs_destroy(m); strncpy(g_err, keep, sizeof g_err); return rc; }
                g_transpose_route = 1;
                *out = m;
                return SMH_OK;
            }
        }
    }
    uint32_t *d_rowof = nullptr;
    SMH_HIP(hipMalloc((void **)&d_rowof, a->nnz * sizeof(uint32_t)));
    auto go = [&]() -> int {
        SMH_TRY(expand_rows(a->d_off, a->n_rows, d_rowof, nullptr));
        SMH_HIP(hipStreamSynchronize(nullptr));
        return assemble_common((smh_dtype)a->dtype, a->nnz, a->d_col, d_rowof, a->d_val, nullptr, true, true, out, true);
    };
    const int rc = go();
    (void)hipFree(d_rowof);
    return rc;
}

static int column_info_common(const smh_crs *m, uint32_t *rows, uint32_t *col_ptr, uint32_t *entries, bool on_device) {
    if (!m || !rows || !col_ptr || !entries) return fail(SMH_ERR_INVALID, "NULL argument");
    if (m->nnz && (size_t)m->max_col >= m->n_cols)
        return fail(SMH_ERR_INDEX_RANGE, "column %u out of range for %zu columns", m->max_col, m->n_cols);
    SMH_HIP(hipStreamSynchronize(m->stream));
    if (on_device) return column_info(m->d_off, m->d_col, m->n_rows, m->n_cols, m->nnz, m->max_col, rows, col_ptr, entries, m->stream);
    uint32_t *d[3] = {nullptr, nullptr, nullptr};
    auto go = [&]() -> int {
        SMH_HIP(hipMalloc((void **)&d[0], (m->nnz ? m->nnz : 1) * sizeof(uint32_t)));
        SMH_HIP(hipMalloc((void **)&d[1], (m->n_cols + 1) * sizeof(uint32_t)));
        SMH_HIP(hipMalloc((void **)&d[2], (m->nnz ? m->nnz : 1) * sizeof(uint32_t)));
        SMH_TRY(column_info(m->d_off, m->d_col, m->n_rows, m->n_cols, m->nnz, m->max_col, d[0], d[1], d[2], m->stream));
        if (m->nnz) SMH_HIP(hipMemcpy(rows, d[0], m->nnz * sizeof(uint32_t), hipMemcpyDeviceToHost));
        SMH_HIP(hipMemcpy(col_ptr, d[1], (m->n_cols + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (m->nnz) SMH_HIP(hipMemcpy(entries, d[2], m->nnz * sizeof(uint32_t), hipMemcpyDeviceToHost));
        return SMH_OK;
    };
    const int rc = go();
    for (uint32_t *p : d) (void)hipFree(p);
    return rc;
}

int smh_crs_column_info(const smh_crs *m, uint32_t *rows, uint32_t *col_ptr, uint32_t *entries) {
    return column_info_common(m, rows, col_ptr, entries, false);
}

int smh_crs_column_info_dev(const smh_crs *m, uint32_t *rows_dev, uint32_t *col_ptr_dev, uint32_t *entries_dev) {
    return column_info_common(m, rows_dev, col_ptr_dev, entries_dev, true);
}

// SparseMatrix::prod (sparsematrix.rs:186-210) for SparseMatCRS operands; Err("Dimension mismatch") of :188-190 as a status
int smh_crs_prod(const smh_crs *a, const smh_crs *b, smh_crs **out) {
    if (!a || !b || !out) return fail(SMH_ERR_INVALID, "NULL argument");
    if (a->dtype != b->dtype) return fail(SMH_ERR_INVALID, "operands differ in value type");
    if (a->n_rows != b->n_cols || a->n_cols != b->n_rows) return fail(SMH_ERR_DIM_MISMATCH, "Dimension mismatch");
    if (a->nnz && (size_t)a->max_col >= a->n_cols)
        return fail(SMH_ERR_INDEX_RANGE, "column %u out of range for %zu columns", a->max_col, a->n_cols);
    SMH_HIP(hipStreamSynchronize(a->stream));
    SMH_HIP(hipStreamSynchronize(b->stream));
    smh_crs *m = new (std::nothrow) smh_crs();
    if (!m) return fail(SMH_ERR_OOM, "host allocation failed");
    m->dtype = a->dtype; m->owns = true;
    m->device = current_device();
    auto go = [&]() -> int {
        SMH_TRY(prod_crs(a->dtype, a->d_off, a->d_col, a->d_val, a->n_rows, a->nnz, a->max_col, b->d_off, b->d_col, b->d_val, b->n_rows,
                         &m->n_rows, &m->n_cols, &m->nnz, &m->d_off, &m->d_col, &m->d_val, nullptr));
        return finish_create(m, 0);
    };
    const int rc = go();
    if (rc != SMH_OK) { char keep[512]; strncpy(keep, g_err, sizeof keep); keep[sizeof keep - 1] = 0; smh_crs_destroy(m); strncpy(g_err, keep, sizeof g_err); return rc; }
    *out = m;
    return SMH_OK;
}

int smh_crs_is_symmetric(const smh_crs *m, int *out) {
    if (!m || !out) return fail(SMH_ERR_INVALID, "NULL argument");
    return crs_is_symmetric(m->dtype, m->d_off, m->d_col, m->d_val, m->n_rows, out, m->stream);
}

int smh_crs_is_sorted(const smh_crs *m, int *out) {
    if (!m || !out) return fail(SMH_ERR_INVALID, "NULL argument");
    return crs_is_sorted(m->d_off, m->d_col, m->n_rows, out, m->stream);
}

int smh_crs_sort_rows(smh_crs *m) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    SMH_TRY(sort_rows(m->dtype, m->d_off, m->d_col, m->d_val, m->n_rows, m->nnz, m->max_col, m->stream));
    drop_colblock(m);  // the blocked copy keeps storage order inside a (row, block) pair
    (void)hipFree(m->d_col16);  // the 16-bit column arrays follow the storage order too: rebuilt on next use
    m->d_col16 = nullptr;
    drop_stream_codes(m);
    return SMH_OK;
}

int smh_crs_destroy(smh_crs *m) {
    if (!m) return SMH_OK;
    if (m->stream) { (void)hipStreamSynchronize(m->stream); (void)hipStreamDestroy(m->stream); }
    if (m->owns) { (void)hipFree(m->d_off); (void)hipFree(m->d_col); (void)hipFree(m->d_val); }
    (void)hipFree(m->d_tile_row); (void)hipFree(m->d_tile_nz); (void)hipFree(m->d_carry_row); (void)hipFree(m->d_carry_val);
    (void)hipFree(m->d_phase_ptr); (void)hipFree(m->d_phases); (void)hipFree(m->d_col16); (void)hipFree(m->d_ring_win);
    (void)hipFree(m->d_stream_cwin); (void)hipFree(m->d_stream_code);
    (void)hipFree(m->d_stream_len8); (void)hipFree(m->d_stream_tbase); (void)hipFree(m->d_stream_dict);
    (void)hipFree(m->d_cb_off); (void)hipFree(m->d_cb_col); (void)hipFree(m->d_cb_val);
    (void)hipFree(m->d_cf_seg); (void)hipFree(m->d_cf_cnt); (void)hipFree(m->d_cf_col); (void)hipFree(m->d_cf_val);
    (void)hipFree(m->d_cf_tile_row);
    (void)smh_crs_destroy(m->split_long); (void)smh_crs_destroy(m->split_short);
    (void)hipFree(m->d_split_rows); (void)hipFree(m->d_split_y);
    tiled_free(m);
    (void)hipFree(m->d_x); (void)hipFree(m->d_y);
    (void)hipGetLastError();
    delete m;
    return SMH_OK;
}

int smh_crs_update_values(smh_crs *m, const void *values_host) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL argument");
    if (m->nnz && values_host) SMH_HIP(hipMemcpy(m->d_val, values_host, m->nnz * dtype_size(m->dtype), hipMemcpyHostToDevice));
    drop_colblock(m);
    // the value dictionary of K1s XD-V described the OLD values: look again now (one pass over the values: nothing beside the copy above)
    // and take the indices out of the codes, or put the new ones in
    m->stream_dict_state = -1;
    if (m->stream_coded && m->stream_direct) {
        StreamCfg c;
        SMH_TRY(stream_cfg(m, &c, true));
    }
    return SMH_OK;
}

int smh_crs_download(const smh_crs *m, uint32_t *offset_rows, uint32_t *columns, void *values) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    SMH_HIP(hipStreamSynchronize(m->stream));
    if (offset_rows) SMH_HIP(hipMemcpy(offset_rows, m->d_off, (m->n_rows + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (columns && m->nnz) SMH_HIP(hipMemcpy(columns, m->d_col, m->nnz * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (values && m->nnz) SMH_HIP(hipMemcpy(values, m->d_val, m->nnz * dtype_size(m->dtype), hipMemcpyDeviceToHost));
    return SMH_OK;
}

size_t smh_crs_n_rows(const smh_crs *m) { return m ? m->n_rows : 0; }
size_t smh_crs_n_cols(const smh_crs *m) { return m ? m->n_cols : 0; }
size_t smh_crs_nnz(const smh_crs *m) { return m ? m->nnz : 0; }
size_t smh_crs_orphans(const smh_crs *m) { return m ? m->orphans : 0; }
int smh_crs_dtype(const smh_crs *m) { return m ? m->dtype : -1; }

int smh_crs_max_row_len(const smh_crs *m, uint32_t *out) {
    if (!m || !out) return fail(SMH_ERR_INVALID, "NULL argument");
    *out = m->max_row_len;
    return SMH_OK;
}

int smh_crs_col_range(const smh_crs *m, uint32_t *min_out, uint32_t *max_out) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    if (min_out) *min_out = m->min_col;
    if (max_out) *max_out = m->max_col;
    return SMH_OK;
}

int smh_crs_scale(smh_crs *m, double a) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    SMH_TRY(launch_scale_values(m->dtype, m->d_val, m->nnz, a, m->stream));
    // (K1s XD-V: every entry is its dictionary value times a, rounded as the entry itself is: the indices in the codes stay right)
    if (m->d_stream_dict && m->stream_dict_state == 1) SMH_TRY(launch_scale_values(m->dtype, m->d_stream_dict, 32, a, m->stream));
    if (m->cb_built) SMH_TRY(launch_scale_values(m->dtype, m->d_cb_val, m->nnz, a, m->stream));
    if (m->cf_built && m->cf_ok) SMH_TRY(launch_scale_values(m->dtype, m->d_cf_val, m->nnz, a, m->stream));
    if (m->t2_built && m->t2_ok) SMH_TRY(launch_scale_values(m->dtype, m->d_t2_val, (size_t)m->t2_tot, a, m->stream));
    if (m->split_built && m->split_ok) {
        SMH_TRY(smh_crs_scale(m->split_long, a));
        SMH_TRY(smh_crs_scale(m->split_short, a));
    }
    SMH_HIP(hipStreamSynchronize(m->stream));
    return SMH_OK;
}

int smh_crs_tiled_layout(smh_crs *m, uint32_t *n_slices_out, uint32_t *slice_columns_out, uint32_t *rows_per_block_out, uint32_t *n_row_blocks_out,
                         size_t *copy_entries_out) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    SMH_TRY(tiled_build(m));
    if (!m->t2_ok) return fail(SMH_ERR_INVALID, "the tiled copy could not be built for this matrix");
    if (n_slices_out) *n_slices_out = m->t2_n_cb;
    if (slice_columns_out) *slice_columns_out = tiled_slice_columns(m->dtype);
    if (rows_per_block_out) *rows_per_block_out = m->t2_R;
    if (n_row_blocks_out) *n_row_blocks_out = m->t2_n_rb;
    if (copy_entries_out) *copy_entries_out = (size_t)m->t2_tot;
    return SMH_OK;
}

int smh_crs_tiled_products(smh_crs *m, size_t *n_products_out) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    SMH_TRY(tiled_build(m));
    if (!m->t2_ok) return fail(SMH_ERR_INVALID, "the tiled copy could not be built for this matrix");
    if (n_products_out) *n_products_out = (size_t)m->t3_n_prod;
    return SMH_OK;
}

int smh_crs_tiled_array(smh_crs *m, int which, void *out, size_t capacity_bytes, size_t *bytes_out) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    SMH_TRY(tiled_build(m));
    if (!m->t2_ok) return fail(SMH_ERR_INVALID, "the tiled copy could not be built for this matrix");
    return tiled_array(m, which, out, capacity_bytes, bytes_out);
}

int smh_crs_set_colblock_shift(smh_crs *m, uint32_t shift) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    if (shift > 31) return fail(SMH_ERR_INVALID, "column block shift must be 0 (automatic) or 1..31");
    if (shift != m->cb_forced_shift) drop_colblock(m);
    m->cb_forced_shift = shift;
    return SMH_OK;
}

int smh_crs_colblock(smh_crs *m, uint32_t *shift_out, size_t *n_blocks_out, int *rows_per_thread_out,
                     double *span_fraction_out, uint32_t *offsets_out, uint32_t *columns_out, void *values_out) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    SMH_TRY(ensure_ring_plan(m, false));  // the locality statistic
    SMH_TRY(ensure_colblock(m));
    if (shift_out) *shift_out = m->cb_shift;
    if (n_blocks_out) *n_blocks_out = m->cb_blocks;
    if (rows_per_thread_out) *rows_per_thread_out = m->cb_rpt;
    if (span_fraction_out) *span_fraction_out = m->span_fraction;
    if (offsets_out)
        SMH_HIP(hipMemcpy(offsets_out, m->d_cb_off, m->cb_blocks * (m->n_rows + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (columns_out && m->nnz) SMH_HIP(hipMemcpy(columns_out, m->d_cb_col, m->nnz * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (values_out && m->nnz) SMH_HIP(hipMemcpy(values_out, m->d_cb_val, m->nnz * dtype_size(m->dtype), hipMemcpyDeviceToHost));
    return SMH_OK;
}

int smh_crs_colfused(smh_crs *m, int *fits_out, uint32_t *shift_out, size_t *n_blocks_out, uint32_t *rows_per_lane_out,
                     size_t *n_tiles_out, uint32_t *tile_rows_out, uint32_t *segments_out, uint8_t *counts_out, uint32_t *columns_out,
                     void *values_out) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    SMH_TRY(ensure_colfused(m));
    const size_t tile_rows = (size_t)64 * m->cf_rt, n_tiles = m->cf_ok ? m->cf_tiles : 0;
    if (fits_out) *fits_out = m->cf_ok ? 1 : 0;
    if (shift_out) *shift_out = m->cf_shift;
    if (n_blocks_out) *n_blocks_out = m->cf_blocks;
    if (rows_per_lane_out) *rows_per_lane_out = m->cf_rt;
    if (n_tiles_out) *n_tiles_out = n_tiles;
    if (!m->cf_ok) return SMH_OK;
    SMH_HIP(hipStreamSynchronize(m->stream));
    if (tile_rows_out) SMH_HIP(hipMemcpy(tile_rows_out, m->d_cf_tile_row, (n_tiles + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (segments_out) SMH_HIP(hipMemcpy(segments_out, m->d_cf_seg, (n_tiles * m->cf_blocks + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (counts_out && n_tiles) SMH_HIP(hipMemcpy(counts_out, m->d_cf_cnt, n_tiles * m->cf_blocks * tile_rows, hipMemcpyDeviceToHost));
    if (columns_out && m->nnz) SMH_HIP(hipMemcpy(columns_out, m->d_cf_col, m->nnz * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (values_out && m->nnz) SMH_HIP(hipMemcpy(values_out, m->d_cf_val, m->nnz * dtype_size(m->dtype), hipMemcpyDeviceToHost));
    return SMH_OK;
}

int smh_crs_colsplit(smh_crs *m, int *split_out, uint32_t *min_long_out, size_t *n_long_out, uint32_t *long_rows_out, smh_crs **long_out,
                     smh_crs **short_out) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    SMH_TRY(ensure_split(m));
    if (split_out) *split_out = m->split_ok ? 1 : 0;
    if (min_long_out) *min_long_out = kSplitMinLong;
    if (n_long_out) *n_long_out = m->split_ok ? m->split_n_long : 0;
    if (long_out) *long_out = m->split_ok ? m->split_long : nullptr;
    if (short_out) *short_out = m->split_ok ? m->split_short : nullptr;
    if (long_rows_out && m->split_ok && m->split_n_long)
        SMH_HIP(hipMemcpy(long_rows_out, m->d_split_rows, m->split_n_long * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return SMH_OK;
}

int smh_crs_resolved_variant(const smh_crs *m, int *variant_out, int *lanes_out) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    if (variant_out) *variant_out = resolve_variant(m, SMH_SPMV_AUTO);
    if (lanes_out) *lanes_out = auto_lanes(m);
    return SMH_OK;
}

int smh_crs_set_stream_xs(smh_crs *m, int mode) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    if (mode < -1 || mode > 1) return fail(SMH_ERR_INVALID, "mode must be -1 (automatic), 0 (never) or 1 (whenever the tiles allow)");
    m->use_stream_xs = mode;
    if (m->stream_coded) {  // the choice may flip the code array's meaning: now, not inside a later launch
        StreamCfg c;
        SMH_TRY(stream_cfg(m, &c, true));
    }
    return SMH_OK;
}

int smh_crs_set_stream_direct(smh_crs *m, int mode) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    if (mode < -1 || mode > 1) return fail(SMH_ERR_INVALID, "mode must be -1 (automatic), 0 (never) or 1 (whenever x is staged)");
    m->use_stream_direct = mode;
    if (m->stream_coded) {
        StreamCfg c;
        SMH_TRY(stream_cfg(m, &c, true));
    }
    return SMH_OK;
}

int smh_crs_stream_direct(smh_crs *m, int *direct_out) {
    if (!m || !direct_out) return fail(SMH_ERR_INVALID, "NULL argument");
    StreamCfg c;
    SMH_TRY(stream_cfg(m, &c));
    *direct_out = c.direct && c.xs != 0;
    return SMH_OK;
}

int smh_crs_stream_value_dict(smh_crs *m, int *n_values_out, void *values_out) {
    if (!m || !n_values_out) return fail(SMH_ERR_INVALID, "NULL argument");
    StreamCfg c;
    SMH_TRY(stream_cfg(m, &c));
    *n_values_out = c.dict ? (int)m->stream_dict_n : 0;
    if (c.dict && values_out) {
        SMH_HIP(hipMemcpyAsync(values_out, m->d_stream_dict, (size_t)m->stream_dict_n * dtype_size(m->dtype), hipMemcpyDeviceToHost, m->stream));
        SMH_HIP(hipStreamSynchronize(m->stream));
    }
    return SMH_OK;
}

int smh_crs_set_stream_value_dict(smh_crs *m, int mode) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    if (mode < -1 || mode > 0) return fail(SMH_ERR_INVALID, "mode must be -1 (automatic: whenever the values allow) or 0 (never)");
    m->use_stream_vdict = mode;
    if (m->stream_coded) {
        StreamCfg c;
        SMH_TRY(stream_cfg(m, &c, true));
    }
    return SMH_OK;
}

int smh_crs_stream_layout(smh_crs *m, int *coded_out, int *byte_lengths_out, int *small_tiles_out, int *xs_chunks_out) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    StreamCfg c;
    SMH_TRY(stream_cfg(m, &c));
    if (coded_out) *coded_out = c.code && c.cwin;
    if (byte_lengths_out) *byte_lengths_out = c.len8 && c.tbase;
    if (small_tiles_out) *small_tiles_out = c.small;
    if (xs_chunks_out) *xs_chunks_out = c.xs;
    return SMH_OK;
}

int smh_crs_set_vector_chunks(smh_crs *m, int chunks) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    if (chunks < 0 || chunks > 3) return fail(SMH_ERR_INVALID, "chunks per lane must be 0 (automatic), 1, 2 or 3");
    m->forced_chunks = chunks;
    return SMH_OK;
}

int smh_crs_set_ring(smh_crs *m, int mode) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    if (mode < -1 || mode > 1) return fail(SMH_ERR_INVALID, "ring mode must be -1 (auto), 0 (plain K1) or 1 (K1r)");
    m->use_ring = mode;
    return SMH_OK;
}

int smh_crs_ring_plan(smh_crs *m, uint32_t *n_blocks_out, size_t *n_phases_out, double *ring_fraction_out,
                      int *active_out, uint32_t *phase_ptr_out, uint32_t *phases_out) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    SMH_TRY(ensure_ring_plan(m));
    bool ring = false;
    SMH_TRY(vector_uses_ring(m, &ring));
    if (n_blocks_out) *n_blocks_out = m->ring_blocks;
    if (n_phases_out) *n_phases_out = m->ring_n_phases;
    if (ring_fraction_out) *ring_fraction_out = m->ring_fraction;
    if (active_out) *active_out = ring ? 1 : 0;
    if (phase_ptr_out)
        SMH_HIP(hipMemcpy(phase_ptr_out, m->d_phase_ptr, (m->ring_blocks + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (phases_out && m->ring_n_phases)
        SMH_HIP(hipMemcpy(phases_out, m->d_phases, m->ring_n_phases * sizeof(RingPhase), hipMemcpyDeviceToHost));
    return SMH_OK;
}

int smh_crs_ring_entries(smh_crs *m, uint32_t *out) {
    if (!m || !out) return fail(SMH_ERR_INVALID, "NULL argument");
    SMH_TRY(ensure_ring_plan(m));
    *out = m->ring_entries;
    return SMH_OK;
}

int smh_crs_ring_bands(smh_crs *m, uint32_t *bands_out, uint32_t *intervals_out) {
    if (!m || !bands_out) return fail(SMH_ERR_INVALID, "NULL argument");
    SMH_TRY(ensure_ring_plan(m));
    *bands_out = m->ring_bands;
    if (intervals_out && m->ring_bands == 4 && m->d_ring_win)
        SMH_HIP(hipMemcpy(intervals_out, m->d_ring_win, ((m->n_rows + 63) / 64) * 8 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return SMH_OK;
}

int smh_crs_set_vector_lanes(smh_crs *m, int lanes) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    if (lanes != 0 && (lanes < 1 || lanes > 64 || (lanes & (lanes - 1))))
        return fail(SMH_ERR_INVALID, "lanes per row must be 0 or a power of two in 1..64");
    m->forced_lanes = lanes;
    return SMH_OK;
}

static int prepare_inner(smh_crs *m, int variant) {
    switch (resolve_variant(m, variant)) {
        case SMH_SPMV_VECTOR: {
            bool ring = false;
            return vector_uses_ring(m, &ring);  // builds the K1r phase plan when the ring is used
        }
        case SMH_SPMV_MERGE: return ensure_merge_ws(m);
        case SMH_SPMV_COLBLOCK: return ensure_colblock(m);
        case SMH_SPMV_COLFUSED:
            SMH_TRY(ensure_colfused(m));
            return m->cf_ok ? SMH_OK : ensure_colblock(m);
        case SMH_SPMV_COLSPLIT:
            SMH_TRY(ensure_split(m));
            if (!m->split_ok) return ensure_colblock(m);
            SMH_TRY(smh_crs_prepare(m->split_short, SMH_SPMV_AUTO));
            return smh_crs_prepare(m->split_long, SMH_SPMV_AUTO);
        case SMH_SPMV_STREAM: {
            StreamCfg c;
            return stream_cfg(m, &c, true);  // code tables; the code array in the form the current settings ask for
        }
        case SMH_SPMV_TILED: return tiled_build(m);
        case SMH_SPMV_SEQ: return SMH_OK;
        default: return fail(SMH_ERR_INVALID, "unknown SpMV variant %d", variant);
    }
}

int smh_crs_prepare(smh_crs *m, int variant) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    // what the inspectors cost (smh_crs_prepare_stats): wall time of the build, device-synchronised at both ends, and the
    // device memory it leaves allocated (pooled blocks: everything of 1 MiB and more; scratch freed inside the build nets out)
    const auto t0 = std::chrono::steady_clock::now();
    const long long b0 = pool_thread_net_bytes();
    // AUTO's plan refused by its lazy build (marked so by the builder): resolve again, as the first product would
    const int tried = resolve_variant(m, variant);
    int rc = prepare_inner(m, variant);
    if (rc != SMH_OK && variant == SMH_SPMV_AUTO && rc != SMH_ERR_INDEX_RANGE && resolve_variant(m, variant) != tried) {
        g_err[0] = 0;
        rc = prepare_inner(m, variant);
    }
    if (rc == SMH_OK && m->stream) {
        const hipError_t e = hipStreamSynchronize(m->stream);
        if (e != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__);
    }
    m->prepare_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    m->prepare_bytes += pool_thread_net_bytes() - b0;
    return rc;
}

int smh_crs_prepare_stats(smh_crs *m, int variant, double *prepare_ms_out, size_t *derived_bytes_out) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    SMH_TRY(smh_crs_prepare(m, variant));
    if (prepare_ms_out) *prepare_ms_out = m->create_ms + m->prepare_ms;
    if (derived_bytes_out) *derived_bytes_out = (size_t)(m->prepare_bytes + m->create_bytes > 0 ? m->prepare_bytes + m->create_bytes : 0);
    return SMH_OK;
}

int smh_crs_spmv_dev(smh_crs *m, const void *x_dev, size_t x_len, void *y_dev, int variant, void *stream) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    if (m->n_rows && (!y_dev || (m->nnz && !x_dev))) return fail(SMH_ERR_INVALID, "NULL device vector");
    return spmv_enqueue(m, x_dev, x_len, y_dev, variant, (hipStream_t)stream);
}

int smh_crs_spmv(smh_crs *m, const void *x_host, size_t x_len, void *y_host, int variant) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    if (m->n_rows == 0) return SMH_OK;
    if (!y_host || (x_len && !x_host)) return fail(SMH_ERR_INVALID, "NULL host vector");
    const size_t vs = dtype_size(m->dtype);
    SMH_TRY(ensure_cap(&m->d_x, &m->d_x_cap, x_len * vs));
    SMH_TRY(ensure_cap(&m->d_y, &m->d_y_cap, m->n_rows * vs));
    if (x_len) SMH_HIP(hipMemcpyAsync(m->d_x, x_host, x_len * vs, hipMemcpyHostToDevice, m->stream));
    SMH_TRY(spmv_enqueue(m, m->d_x, x_len, m->d_y, variant, m->stream));
    SMH_HIP(hipMemcpyAsync(y_host, m->d_y, m->n_rows * vs, hipMemcpyDeviceToHost, m->stream));
    SMH_HIP(hipStreamSynchronize(m->stream));
    return SMH_OK;
}

size_t smh_crs_merge_tiles(const smh_crs *m) {
    if (!m) return 0;
    return (size_t)(((uint64_t)m->n_rows + m->nnz + kMergeTile - 1) / kMergeTile);
}

size_t smh_crs_merge_tile_items(const smh_crs *) { return kMergeTile; }

int smh_crs_merge_table(smh_crs *m, uint32_t *row_out, uint32_t *nnz_out) {
    if (!m || !row_out || !nnz_out) return fail(SMH_ERR_INVALID, "NULL argument");
    SMH_TRY(ensure_merge_ws(m));
    if (m->n_tiles == 0) { row_out[0] = 0; nnz_out[0] = 0; return SMH_OK; }
    SMH_HIP(hipMemcpy(row_out, m->d_tile_row, (m->n_tiles + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    SMH_HIP(hipMemcpy(nnz_out, m->d_tile_nz, (m->n_tiles + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return SMH_OK;
}

// ---- DenseVec ----------------------------------------------------------------------------------------
int smh_vec_create(smh_dtype dtype, size_t n, smh_vec **out) {
    if (!out) return fail(SMH_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!valid_dtype(dtype)) return fail(SMH_ERR_INVALID, "dtype must be SMH_F32 or SMH_F64");
    SMH_TRY(require_device());
    smh_vec *v = new (std::nothrow) smh_vec();
    if (!v) return fail(SMH_ERR_OOM, "host allocation failed");
    v->dtype = dtype; v->n = n; v->owns = true; v->device = current_device();
    const size_t bytes = (n ? n : 1) * dtype_size(dtype);
    hipError_t e = hipMalloc(&v->d, bytes);
    if (e == hipSuccess) e = hipMemset(v->d, 0, bytes);
    if (e != hipSuccess) { delete v; return hip_fail(e, "smh_vec_create", __FILE__, __LINE__); }
    *out = v;
    return SMH_OK;
}

int smh_vec_from_host(smh_dtype dtype, size_t n, const void *host, smh_vec **out) {
    SMH_TRY(smh_vec_create(dtype, n, out));
    if (n) {
        int rc = smh_vec_upload(*out, host);
        if (rc != SMH_OK) { smh_vec_destroy(*out); *out = nullptr; return rc; }
    }
    return SMH_OK;
}

int smh_vec_wrap_dev(smh_dtype dtype, size_t n, void *dev_ptr, smh_vec **out) {
    if (!out) return fail(SMH_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!valid_dtype(dtype)) return fail(SMH_ERR_INVALID, "dtype must be SMH_F32 or SMH_F64");
    if (n && !dev_ptr) return fail(SMH_ERR_INVALID, "dev_ptr is NULL");
    SMH_TRY(require_device());
    smh_vec *v = new (std::nothrow) smh_vec();
    if (!v) return fail(SMH_ERR_OOM, "host allocation failed");
    v->dtype = dtype; v->n = n; v->d = dev_ptr; v->owns = false; v->device = current_device();
    *out = v;
    return SMH_OK;
}

int smh_vec_destroy(smh_vec *v) {
    if (!v) return SMH_OK;
    if (v->owns) { (void)hipFree(v->d); (void)hipGetLastError(); }
    delete v;
    return SMH_OK;
}

int smh_vec_upload(smh_vec *v, const void *host) {
    if (!v || (v->n && !host)) return fail(SMH_ERR_INVALID, "NULL argument");
    if (v->n) SMH_HIP(hipMemcpy(v->d, host, v->n * dtype_size(v->dtype), hipMemcpyHostToDevice));
    return SMH_OK;
}

int smh_vec_download(const smh_vec *v, void *host) {
    if (!v || (v->n && !host)) return fail(SMH_ERR_INVALID, "NULL argument");
    if (v->n) SMH_HIP(hipMemcpy(host, v->d, v->n * dtype_size(v->dtype), hipMemcpyDeviceToHost));
    return SMH_OK;
}

size_t smh_vec_dim(const smh_vec *v) { return v ? v->n : 0; }
int smh_vec_dtype(const smh_vec *v) { return v ? v->dtype : -1; }
void *smh_vec_data(const smh_vec *v) { return v ? v->d : nullptr; }

int smh_vec_copy(smh_vec *dst, const smh_vec *src) {
    SMH_TRY(vec_check_pair(dst, src));
    if (dst->n != src->n) return fail(SMH_ERR_DIM_MISMATCH, "Dimension mismatch");
    if (src->n) SMH_HIP(hipMemcpy(dst->d, src->d, src->n * dtype_size(src->dtype), hipMemcpyDeviceToDevice));
    return SMH_OK;
}

static int vec_ew(Ew op, smh_vec *x, const smh_vec *y, double a) {
    SMH_TRY(vec_check_pair(x, y));
    // densevec.rs:52-54 / :61-63: panic iff self.dim() < rhs.dim(); zip() covers rhs.dim() entries
    if (x->n < y->n) return fail(SMH_ERR_DIM_MISMATCH, "Dimension mismatch");
    SMH_TRY(launch_ew(x->dtype, op, x->d, y->d, y->n, a, nullptr, nullptr));
    SMH_HIP(hipStreamSynchronize(nullptr));
    return SMH_OK;
}

int smh_vec_add(smh_vec *x, const smh_vec *y) { return vec_ew(Ew::Add, x, y, 0.0); }
int smh_vec_sub(smh_vec *x, const smh_vec *y) { return vec_ew(Ew::Sub, x, y, 0.0); }
int smh_vec_axpy(smh_vec *y, double a, const smh_vec *x) { return vec_ew(Ew::Axpy, y, x, a); }
int smh_vec_xpby(smh_vec *p, double b, const smh_vec *r) { return vec_ew(Ew::Xpby, p, r, b); }

int smh_vec_scale(smh_vec *x, double a) {
    if (!x) return fail(SMH_ERR_INVALID, "NULL vector handle");
    SMH_TRY(launch_ew(x->dtype, Ew::Scale, x->d, nullptr, x->n, a, nullptr, nullptr));
    SMH_HIP(hipStreamSynchronize(nullptr));
    return SMH_OK;
}

int smh_vec_dot(const smh_vec *x, const smh_vec *y, double *out) {
    SMH_TRY(vec_check_pair(x, y));
    if (!out) return fail(SMH_ERR_INVALID, "out is NULL");
    const size_t n = x->n < y->n ? x->n : y->n;  // zip truncates (vector.rs:52)
    void *scratch = nullptr;
    SMH_TRY(reduce_scratch(&scratch));
    char *res = (char *)scratch + kReducePartials * sizeof(double);
    SMH_TRY(launch_dot(x->dtype, x->d, y->d, n, scratch, res, nullptr));
    if (x->dtype == SMH_F64) {
        double h = 0;
        SMH_HIP(hipMemcpy(&h, res, sizeof h, hipMemcpyDeviceToHost));
        *out = h;
    } else {
        float h = 0;
        SMH_HIP(hipMemcpy(&h, res, sizeof h, hipMemcpyDeviceToHost));
        *out = (double)h;
    }
    return SMH_OK;
}

// lhs^T (A rhs) on device vectors: y = A rhs with the matrix's own kernel into the handle's y staging, then the
// two-stage dot with lhs.  (sparsematrix.rs:161-171 sums lhs_i * a_ij * rhs_j over all entries in storage order;
// a parallel reduction regroups that sum, so parity is tolerance-level, like dot.)
static int inner_prod_dev(smh_crs *m, const void *d_lhs, size_t lhs_len, const void *d_rhs, size_t rhs_len, int variant,
                          double *out) {
    if (!out) return fail(SMH_ERR_INVALID, "out is NULL");
    *out = 0.0;
    if (m->n_rows == 0 || m->nnz == 0) return SMH_OK;
    const size_t vs = dtype_size(m->dtype);
    // lhs.get(i) is evaluated inside the entry loop (sparsematrix.rs:165-168): only rows that hold entries index lhs, so a
    // short lhs is an error only if it ends before the last non-empty row (densevec.rs:41 panics there).  The kernels read
    // lhs for every row, so a short-but-legal lhs is continued with zeros in a scratch copy.
    struct Padded {
        void *d = nullptr;
        ~Padded() { if (d) (void)hipFree(d); }
    } padded;
    if (lhs_len < m->n_rows) {
        size_t lo = 0, hi = m->n_rows;  // smallest i with offset_rows[i] == nnz; rows i.. are empty
        while (lo < hi) {
            const size_t mid = lo + (hi - lo) / 2;
            uint32_t off = 0;
            SMH_HIP(hipMemcpyAsync(&off, m->d_off + mid, sizeof off, hipMemcpyDeviceToHost, m->stream));
            SMH_HIP(hipStreamSynchronize(m->stream));
            if ((size_t)off >= m->nnz) hi = mid; else lo = mid + 1;
        }
        if (lhs_len < lo)  // row lo - 1 is the last one with entries
            return fail(SMH_ERR_INDEX_RANGE, "index out of bounds: the len is %zu but the index is %zu", lhs_len, lo - 1);
        SMH_HIP(hipMalloc(&padded.d, m->n_rows * vs));
        SMH_HIP(hipMemsetAsync(padded.d, 0, m->n_rows * vs, m->stream));
        if (lhs_len) SMH_HIP(hipMemcpyAsync(padded.d, d_lhs, lhs_len * vs, hipMemcpyDeviceToDevice, m->stream));
        d_lhs = padded.d;
    }
    void *scratch = nullptr;
    SMH_TRY(reduce_scratch(&scratch));
    char *res = (char *)scratch + kReducePartials * sizeof(double);
    const size_t n_dot = spmv_fused_dot_partials(m, rhs_len, variant, true);
    bool ring = false;
    if (!n_dot && resolve_variant(m, variant) == SMH_SPMV_VECTOR) SMH_TRY(vector_uses_ring(m, &ring));
    if (ring) {
        // K1r: the lanes that would store a row's sum multiply it by lhs[row] instead and the blocks leave partial sums
        if (m->nnz > 0 && (size_t)m->max_col >= rhs_len)
            return fail(SMH_ERR_INDEX_RANGE, "index out of bounds: the len is %zu but the index is %u", rhs_len, m->max_col);
        SMH_TRY(ensure_cap(&m->d_y, &m->d_y_cap, ((size_t)m->ring_blocks + 1) * vs));
        SMH_TRY(launch_spmv_ring2(m->dtype, auto_lanes(m), auto_chunks(m), m->d_off, m->d_col, m->d_col16, m->d_val, d_rhs, const_cast<void *>(d_lhs),
                                  m->n_rows, m->nnz, m->owns || m->nnz % 4 == 0, m->ring_blocks, m->d_phase_ptr, m->d_phases, m->ring_entries,
                                  m->ring_bands, m->stream, m->d_y));
        SMH_TRY(launch_fold2(m->dtype, m->d_y, (size_t)m->ring_blocks + 1, scratch, res, m->stream));
    } else if (n_dot) {
        // K1s: lhs_i * (A rhs)_i summed per tile in the SpMV's epilogue -- no y vector, no second pass; the tile partials
        // are folded by the two small reduction kernels
        SMH_TRY(ensure_cap(&m->d_y, &m->d_y_cap, n_dot * vs));  // (the staging buffer holds the partials here)
        SMH_TRY(spmv_enqueue(m, d_rhs, rhs_len, nullptr, variant, m->stream, m->d_y, d_lhs));
        SMH_TRY(launch_fold2(m->dtype, m->d_y, n_dot, scratch, res, m->stream));
    } else {
        SMH_TRY(ensure_cap(&m->d_y, &m->d_y_cap, m->n_rows * vs));
        SMH_TRY(spmv_enqueue(m, d_rhs, rhs_len, m->d_y, variant, m->stream));
        SMH_TRY(launch_dot(m->dtype, d_lhs, m->d_y, m->n_rows, scratch, res, m->stream));
    }
    double h64 = 0;
    float h32 = 0;
    if (m->dtype == SMH_F64) SMH_HIP(hipMemcpyAsync(&h64, res, sizeof h64, hipMemcpyDeviceToHost, m->stream));
    else SMH_HIP(hipMemcpyAsync(&h32, res, sizeof h32, hipMemcpyDeviceToHost, m->stream));
    SMH_HIP(hipStreamSynchronize(m->stream));
    *out = m->dtype == SMH_F64 ? h64 : (double)h32;
    return SMH_OK;
}

int smh_crs_inner_prod_vec(smh_crs *m, const smh_vec *lhs, const smh_vec *rhs, int variant, double *out) {
    if (!m || !lhs || !rhs) return fail(SMH_ERR_INVALID, "NULL handle");
    if (lhs->dtype != m->dtype || rhs->dtype != m->dtype) return fail(SMH_ERR_INVALID, "value types differ");
    SMH_HIP(hipDeviceSynchronize());  // vectors may have pending work on other streams
    return inner_prod_dev(m, lhs->d, lhs->n, rhs->d, rhs->n, variant, out);
}

int smh_crs_inner_prod(smh_crs *m, const void *lhs_host, size_t lhs_len, const void *rhs_host, size_t rhs_len, int variant,
                       double *out) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    if ((lhs_len && !lhs_host) || (rhs_len && !rhs_host)) return fail(SMH_ERR_INVALID, "NULL host vector");
    const size_t vs = dtype_size(m->dtype);
    // rhs goes to the x staging; lhs to a scratch allocation of its own
    SMH_TRY(ensure_cap(&m->d_x, &m->d_x_cap, rhs_len * vs));
    void *d_lhs = nullptr;
    SMH_HIP(hipMalloc(&d_lhs, (lhs_len ? lhs_len : 1) * vs));
    auto go = [&]() -> int {
        if (rhs_len) SMH_HIP(hipMemcpyAsync(m->d_x, rhs_host, rhs_len * vs, hipMemcpyHostToDevice, m->stream));
        if (lhs_len) SMH_HIP(hipMemcpyAsync(d_lhs, lhs_host, lhs_len * vs, hipMemcpyHostToDevice, m->stream));
        return inner_prod_dev(m, d_lhs, lhs_len, m->d_x, rhs_len, variant, out);
    };
    const int rc = go();
    (void)hipFree(d_lhs);
    return rc;
}

int smh_vec_norm_squared(const smh_vec *x, double *out) { return smh_vec_dot(x, x, out); }

int smh_vec_norm(const smh_vec *x, double *out) {
    SMH_TRY(smh_vec_norm_squared(x, out));
    *out = std::sqrt(*out);  // f64::sqrt(self.norm_squared().into())  vector.rs:61-63
    return SMH_OK;
}

// ---- BLAS-1 on raw device pointers with DEVICE-resident scalars (asynchronous) ------------------------
// Building blocks of a solver whose scalars never visit the host (the multi-GPU CG all-reduces them on
// the device between these calls).
int smh_blas_dot_dev(smh_dtype dtype, const void *x_dev, const void *y_dev, size_t n, void *result_dev,
                     void *scratch_dev, void *stream) {
    if (!valid_dtype(dtype)) return fail(SMH_ERR_INVALID, "dtype must be SMH_F32 or SMH_F64");
    if (!result_dev || !scratch_dev || (n && (!x_dev || !y_dev))) return fail(SMH_ERR_INVALID, "NULL device pointer");
    return launch_dot(dtype, x_dev, y_dev, n, scratch_dev, result_dev, (hipStream_t)stream);
}

size_t smh_blas_dot_scratch_bytes(void) { return (size_t)(kReducePartials + 8) * sizeof(double); }

int smh_blas_axpy_dev(smh_dtype dtype, void *y_dev, const void *a_dev, const void *x_dev, size_t n, void *stream) {
    if (!valid_dtype(dtype)) return fail(SMH_ERR_INVALID, "dtype must be SMH_F32 or SMH_F64");
    if (!a_dev || (n && (!x_dev || !y_dev))) return fail(SMH_ERR_INVALID, "NULL device pointer");
    return launch_ew(dtype, Ew::Axpy, y_dev, x_dev, n, 0.0, a_dev, (hipStream_t)stream);
}

int smh_blas_xpby_dev(smh_dtype dtype, void *p_dev, const void *b_dev, const void *r_dev, size_t n, void *stream) {
    if (!valid_dtype(dtype)) return fail(SMH_ERR_INVALID, "dtype must be SMH_F32 or SMH_F64");
    if (!b_dev || (n && (!p_dev || !r_dev))) return fail(SMH_ERR_INVALID, "NULL device pointer");
    return launch_ew(dtype, Ew::Xpby, p_dev, r_dev, n, 0.0, b_dev, (hipStream_t)stream);
}

int smh_crs_spmv_vec(smh_crs *m, const smh_vec *x, smh_vec *y, int variant) {
    if (!m || !x || !y) return fail(SMH_ERR_INVALID, "NULL handle");
    if (x->dtype != m->dtype || y->dtype != m->dtype) return fail(SMH_ERR_INVALID, "dtype mismatch");
    if (y->n != m->n_rows) return fail(SMH_ERR_DIM_MISMATCH, "Dimension mismatch");
    SMH_TRY(spmv_enqueue(m, x->d, x->n, y->d, variant, m->stream));
    SMH_HIP(hipStreamSynchronize(m->stream));
    return SMH_OK;
}

// ---- ConjugateGradient ---------------------------------------------------------------------------------
int smh_cg_solve_vec(smh_crs *m, const smh_vec *b, smh_vec *x, double tol, size_t iter_max, int variant,
                     size_t check_every, size_t *iters_out, double *rr_out) {
    if (!m || !b || !x) return fail(SMH_ERR_INVALID, "NULL handle");
    if (b->dtype != m->dtype || x->dtype != m->dtype) return fail(SMH_ERR_INVALID, "dtype mismatch");
    if (m->n_rows != m->n_cols) return fail(SMH_ERR_NOT_SQUARE, "Matrix is not symmetric");           // :30-32
    if (m->n_rows != b->n || m->n_rows != x->n)
        return fail(SMH_ERR_DIM_MISMATCH, "Matrix and vector size mismatch");                            // :33-36
    if (check_every == 0) check_every = 4;
    const size_t n = m->n_rows;
    const size_t vs = dtype_size(m->dtype);
    hipStream_t s = m->stream;
    void *r = nullptr, *p = nullptr, *ap = nullptr, *partials = nullptr, *dot_partials = nullptr, *sc = nullptr, *sc2 = nullptr, *h_sc = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int rc = SMH_OK;
    size_t iters = 0;
    double rr = 0.0;
    auto body = [&]() -> int {
        const size_t vb = (n ? n : 1) * vs;
        SMH_HIP(hipMalloc(&r, vb));
        SMH_HIP(hipMalloc(&p, vb));
        SMH_HIP(hipMalloc(&ap, vb));
        SMH_HIP(hipMalloc(&partials, (2 * kReducePartials + 16) * vs));  // (r.r partials | the dot's result | a first fold of many p.Ap partials)
        // p.Ap: left in the SpMV epilogue when the kernel can (K1s), else a separate two-stage dot
        const size_t n_dot = spmv_fused_dot_partials(m, n, variant);
        if (n_dot) SMH_HIP(hipMalloc(&dot_partials, n_dot * vs));
        SMH_HIP(hipMalloc(&sc, cg_scalars_bytes(m->dtype)));
        SMH_HIP(hipMalloc(&sc2, cg_scalars_bytes(m->dtype)));  // (the scalars are double-buffered within an iteration: cg.hip)
        SMH_HIP(hipHostMalloc(&h_sc, cg_scalars_bytes(m->dtype)));
        // r = b - A x  (:38) ; p = r.clone() (:39) ; rr = r.r (:40)
        SMH_TRY(spmv_enqueue(m, x->d, x->n, r, variant, s));
        SMH_TRY(launch_ew(m->dtype, Ew::RSubInto, r, b->d, n, 0.0, nullptr, s));
        if (n) SMH_HIP(hipMemcpyAsync(p, r, n * vs, hipMemcpyDeviceToDevice, s));
        SMH_TRY(cg_begin(m->dtype, sc, r, n, partials, tol, iter_max, s));
        size_t launched = 0;
        int converged = 0;
        // A batch of `check_every` iterations (the product + 2 to 3 launches each) is captured ONCE into a hipGraph and replayed:
        // for small systems the loop is launch-bound.  Iterations past convergence / iter_max are no-ops on the
        // device, so whole batches can always be replayed.  (All workspaces were created by the SpMV above.)
        if (iter_max > check_every && hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            int crc = SMH_OK;
            for (size_t i = 0; i < check_every && crc == SMH_OK; ++i) {
                crc = spmv_enqueue(m, p, n, ap, variant, s, dot_partials);
                if (crc == SMH_OK)
                    crc = cg_iter_tail(m->dtype, sc, sc2, x->d, r, p, ap, n, partials, dot_partials, (uint32_t)n_dot, s);
            }
            hipError_t ce = hipStreamEndCapture(s, &graph);
            if (crc != SMH_OK || ce != hipSuccess || !graph ||
                hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0) != hipSuccess) {
                graph_exec = nullptr;  // fall back to plain stream launches
                (void)hipGetLastError();
            }
        } else {
            (void)hipGetLastError();
        }
        while (launched < iter_max) {
            size_t batch = iter_max - launched < check_every ? iter_max - launched : check_every;
            if (graph_exec) {
                SMH_HIP(hipGraphLaunch(graph_exec, s));
                batch = check_every;
            } else {
                for (size_t i = 0; i < batch; ++i) {
                    SMH_TRY(spmv_enqueue(m, p, n, ap, variant, s, dot_partials));                        // :43
                    SMH_TRY(cg_iter_tail(m->dtype, sc, sc2, x->d, r, p, ap, n, partials, dot_partials, (uint32_t)n_dot, s));  // :45-59
                }
            }
            launched += batch;
            SMH_HIP(hipMemcpyAsync(h_sc, sc, cg_scalars_bytes(m->dtype), hipMemcpyDeviceToHost, s));
            SMH_HIP(hipStreamSynchronize(s));
            uint64_t it64 = 0;
            cg_read_scalars(m->dtype, h_sc, &converged, &it64, &rr);
            iters = (size_t)it64;
            if (converged) break;
        }
        if (iter_max == 0) {
            SMH_HIP(hipMemcpyAsync(h_sc, sc, cg_scalars_bytes(m->dtype), hipMemcpyDeviceToHost, s));
            SMH_HIP(hipStreamSynchronize(s));
            uint64_t it64 = 0;
            cg_read_scalars(m->dtype, h_sc, &converged, &it64, &rr);
        }
        return SMH_OK;
    };
    rc = body();
    char keep[512];
    strncpy(keep, g_err, sizeof keep); keep[sizeof keep - 1] = 0;
    (void)hipStreamSynchronize(s);
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipFree(r); (void)hipFree(p); (void)hipFree(ap); (void)hipFree(partials); (void)hipFree(dot_partials); (void)hipFree(sc); (void)hipFree(sc2);
    if (h_sc) (void)hipHostFree(h_sc);
    (void)hipGetLastError();
    strncpy(g_err, keep, sizeof g_err);
    if (rc != SMH_OK) return rc;
    if (iters_out) *iters_out = iters;
    if (rr_out) *rr_out = rr;
    return SMH_OK;
}

int smh_cg_solve(smh_crs *m, const void *b_host, size_t b_len, void *x_host_inout, size_t x_len, double tol,
                 size_t iter_max, int variant, size_t *iters_out, double *rr_out) {
    if (!m) return fail(SMH_ERR_INVALID, "NULL handle");
    if (m->n_rows != m->n_cols) return fail(SMH_ERR_NOT_SQUARE, "Matrix is not symmetric");
    if (m->n_rows != b_len || m->n_rows != x_len) return fail(SMH_ERR_DIM_MISMATCH, "Matrix and vector size mismatch");
    smh_vec *b = nullptr, *x = nullptr;
    int rc = smh_vec_from_host((smh_dtype)m->dtype, b_len, b_host, &b);
    if (rc == SMH_OK) rc = smh_vec_from_host((smh_dtype)m->dtype, x_len, x_host_inout, &x);
    if (rc == SMH_OK) rc = smh_cg_solve_vec(m, b, x, tol, iter_max, variant, 0, iters_out, rr_out);
    if (rc == SMH_OK) rc = smh_vec_download(x, x_host_inout);
    smh_vec_destroy(b);
    smh_vec_destroy(x);
    return rc;
}

// ---- synthetic workloads ---------------------------------------------------------------------------------
int smh_synth_x(smh_dtype dtype, uint64_t seed, size_t begin, size_t n, void *x_dev, void *stream) {
    SMH_TRY(require_device());
    return synth_x(dtype, seed, begin, n, x_dev, (hipStream_t)stream);
}

int smh_synth_fixed(smh_dtype dtype, uint64_t seed, int pattern, size_t n, uint32_t k, size_t row_begin,
                    size_t row_end, uint32_t *offset_rows_dev, uint32_t *columns_dev, void *values_dev, void *stream) {
    SMH_TRY(require_device());
    if (k == 0 || n < k || row_end < row_begin || row_end > n) return fail(SMH_ERR_INVALID, "bad generator arguments");
    if ((uint64_t)(row_end - row_begin) * k >= 0xFFFFFFFFull)
        return fail(SMH_ERR_CAPACITY, "Maximum number of %u entries reached", 0xFFFFFFFFu);  // sparsemat_crs.rs:82-84
    return synth_fixed(dtype, seed, pattern, n, k, row_begin, row_end, offset_rows_dev, columns_dev, values_dev,
                       (hipStream_t)stream);
}

int smh_synth_powerlaw_cdf(uint32_t kmax, double alpha, uint32_t *cdf_host) {
    if (!cdf_host || kmax == 0) return fail(SMH_ERR_INVALID, "bad arguments");
    synth_powerlaw_cdf(kmax, alpha, cdf_host);
    return SMH_OK;
}

int smh_synth_powerlaw_lengths(uint64_t seed, size_t row_begin, size_t row_end, uint32_t kmax, const uint32_t *cdf_host,
                               uint32_t *lengths_host) {
    if (!cdf_host || !lengths_host || kmax == 0) return fail(SMH_ERR_INVALID, "bad arguments");
    synth_powerlaw_lengths(seed, row_begin, row_end, kmax, cdf_host, lengths_host);
    return SMH_OK;
}

int smh_synth_fill(smh_dtype dtype, uint64_t seed, size_t n_cols, size_t row_begin, size_t row_end,
                   const uint32_t *offset_rows_dev, uint32_t *columns_dev, void *values_dev, void *stream) {
    SMH_TRY(require_device());
    if (n_cols == 0) return fail(SMH_ERR_INVALID, "n_cols == 0");
    return synth_fill(dtype, seed, n_cols, row_begin, row_end, offset_rows_dev, columns_dev, values_dev,
                      (hipStream_t)stream);
}

int smh_synth_laplace3d(smh_dtype dtype, size_t nx, size_t ny, size_t nz, size_t row_begin, size_t row_end,
                        uint32_t *offset_rows_dev, uint32_t *columns_dev, void *values_dev, size_t *nnz_out,
                        void *stream) {
    if (row_end < row_begin || row_end > nx * ny * nz) return fail(SMH_ERR_INVALID, "bad row range");
    const size_t nnz = synth_laplace3d_nnz(nx, ny, nz, row_begin, row_end);
    if (nnz_out) *nnz_out = nnz;
    if (!offset_rows_dev && !columns_dev && !values_dev) return SMH_OK;  // size query only (no device needed)
    if (nnz >= 0xFFFFFFFFull) return fail(SMH_ERR_CAPACITY, "block nnz exceeds u32");
    SMH_TRY(require_device());
    return synth_laplace3d(dtype, nx, ny, nz, row_begin, row_end, offset_rows_dev, columns_dev, values_dev,
                           (hipStream_t)stream);
}

// ---- raw device memory helpers -----------------------------------------------------------------------------
int smh_dev_alloc(size_t bytes, void **out) {
    if (!out) return fail(SMH_ERR_INVALID, "out is NULL");
    SMH_TRY(require_device());
    SMH_HIP(hipMalloc(out, bytes ? bytes : 1));
    return SMH_OK;
}
int smh_dev_free(void *p) {
    if (p) SMH_HIP(hipFree(p));
    return SMH_OK;
}
int smh_dev_upload(void *dst_dev, const void *src_host, size_t bytes) {
    if (bytes) SMH_HIP(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return SMH_OK;
}
int smh_dev_download(void *dst_host, const void *src_dev, size_t bytes) {
    if (bytes) SMH_HIP(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return SMH_OK;
}
int smh_dev_memset(void *dst_dev, int value, size_t bytes, void *stream) {
    if (bytes) SMH_HIP(hipMemsetAsync(dst_dev, value, bytes, (hipStream_t)stream));
    return SMH_OK;
}

// ---- streams and timing events for hosts without a HIP binding (bench.py: HIP events around every launch) ----
int smh_stream_create(void **stream_out) {
    if (!stream_out) return fail(SMH_ERR_INVALID, "stream_out is NULL");
    SMH_TRY(require_device());
    hipStream_t s = nullptr;
    SMH_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream_out = s;
    return SMH_OK;
}
int smh_stream_destroy(void *stream) {
    if (stream) SMH_HIP(hipStreamDestroy((hipStream_t)stream));
    return SMH_OK;
}
int smh_stream_synchronize(void *stream) {
    SMH_HIP(hipStreamSynchronize((hipStream_t)stream));
    return SMH_OK;
}
int smh_event_create(void **event_out) {
    if (!event_out) return fail(SMH_ERR_INVALID, "event_out is NULL");
    SMH_TRY(require_device());
    hipEvent_t e = nullptr;
    SMH_HIP(hipEventCreate(&e));
    *event_out = e;
    return SMH_OK;
}
int smh_event_destroy(void *event) {
    if (event) SMH_HIP(hipEventDestroy((hipEvent_t)event));
    return SMH_OK;
}
int smh_event_record(void *event, void *stream) {
    SMH_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return SMH_OK;
}
int smh_event_elapsed_ms(void *start, void *stop, float *ms_out) {
    if (!ms_out) return fail(SMH_ERR_INVALID, "ms_out is NULL");
    SMH_HIP(hipEventSynchronize((hipEvent_t)stop));
    SMH_HIP(hipEventElapsedTime(ms_out, (hipEvent_t)start, (hipEvent_t)stop));
    return SMH_OK;
}

}  // extern "C"
