// internal.hpp -- shared declarations of libsparsemat_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/sparsemat_hip.h"

namespace smh {

// ---- error plumbing: never throw across the C ABI --------------------------------------
void set_error(const char *fmt, ...);
int fail(int status, const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define SMH_HIP(call)                                                     \
    do {                                                                  \
        hipError_t e__ = (call);                                          \
        if (e__ != hipSuccess) return ::smh::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

#define SMH_TRY(expr)                 \
    do {                              \
        int rc__ = (expr);            \
        if (rc__ != SMH_OK) return rc__; \
    } while (0)

int require_device();  // SMH_ERR_NO_DEVICE when no HIP device is visible
int current_device();

inline size_t dtype_size(int dt) { return dt == SMH_F64 ? 8 : 4; }

constexpr int kWave = 64;           // gfx950 wavefront
constexpr int kBlock = 256;         // 4 waves: one per SIMD
// The sum of a value over the wavefront by DPP (row_shr 1, 2, 4, 8, row_bcast 15, 31: an inclusive scan; lane 63 ends with the total)
// -- six vector instructions in a fixed order, where the __shfl_down butterfly is a chain of six ds_bpermute round trips through
// the LDS at the very end of a workgroup's life.  Lanes without a source add +0.
template <int CTRL, int ROWS> __device__ __forceinline__ float wave_dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xF, false));
}
template <int CTRL, int ROWS> __device__ __forceinline__ double wave_dpp_add(double v) {
    const long long b = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROWS, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)((unsigned long long)b >> 32), CTRL, ROWS, 0xF, false);
    return v + __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <typename T> __device__ __forceinline__ T wave_sum_to_lane63(T v) {
    v = wave_dpp_add<0x111, 0xF>(v);
    v = wave_dpp_add<0x112, 0xF>(v);
    v = wave_dpp_add<0x114, 0xF>(v);
    v = wave_dpp_add<0x118, 0xF>(v);
    v = wave_dpp_add<0x142, 0xA>(v);
    v = wave_dpp_add<0x143, 0xC>(v);
    return v;
}
constexpr int kMergeItemsPerThread = 8;
constexpr int kMergeTile = kBlock * kMergeItemsPerThread;  // merge items (rows + nnz) per tile
constexpr int kReducePartials = 1024;  // blocks of a stage-1 reduction
constexpr int kRingEntries = 16384;    // K1r: columns of x the LDS ring holds (64 KiB f32 / 128 KiB f64)
constexpr int kRingEntriesWide = 32768;  // ... f32 only, for rows that need it: 128 KiB, one 1024-thread block per CU
constexpr int kStreamRows = kBlock;    // K1s: rows per tile (one thread folds one row)
constexpr int kStreamCap = 4096;       // K1s: entries of a tile staged in LDS
constexpr int kStreamCapSmall = 2045;  // K1s: ... when no tile holds more (two 16-byte chunks per thread from an aligned start)
constexpr int kStreamCodeWidth = 16384;  // K1s 16-bit column codes: columns per interval (14 bits) x 4 intervals

// ---- launchers (defined in the .hip files) ---------------------------------------------
// K1 / SEQ
int launch_spmv_vector(int dtype, int lanes, const uint32_t *off, const uint32_t *col, const void *val,
                       const void *x, void *y, size_t n_rows, size_t nnz, hipStream_t s);
int launch_spmv_seq(int dtype, const uint32_t *off, const uint32_t *col, const void *val, const void *x,
                    void *y, size_t n_rows, hipStream_t s);
// K2
int launch_merge_table(const uint32_t *off, size_t n_rows, size_t nnz, size_t n_tiles, uint32_t *tile_row,
                       uint32_t *tile_nz, hipStream_t s);
int launch_spmv_merge(int dtype, const uint32_t *off, const uint32_t *col, const void *val, const void *x,
                      void *y, size_t n_rows, size_t nnz, size_t n_tiles, const uint32_t *tile_row,
                      const uint32_t *tile_nz, uint32_t *carry_row, void *carry_val, hipStream_t s);
// K1s (CSR-stream for short rows)
int launch_spmv_stream(int dtype, const uint32_t *off, const uint32_t *col, const void *val, const void *x, void *y,
                       size_t n_rows, size_t nnz, bool padded, int rows_per_thread,
                       bool single_pass /* no tile holds more than kStreamCap entries */,
                       void *dot_partials /* optional: x.y per tile, stream_tiles() entries */,
                       const uint16_t *code, const uint32_t *cwin /* optional: 16-bit column codes + their interval table */,
                       const uint8_t *len8, const uint32_t *tbase /* optional (with codes): byte row lengths + tile starts */,
                       const void *dot_lhs /* with dot_partials: the vector dotted with y (NULL: x); y may then be NULL */,
                       hipStream_t s, bool small_tiles = false /* no tile holds more than kStreamCapSmall entries */,
                       int xs = 0 /* ... and the tiles' column intervals fit an LDS stage of x of xs * 1024 entries (2 or 4; stream_xs_* checked by the caller) */,
                       uint64_t tile_begin = 0, uint64_t tile_end = ~uint64_t(0) /* the matrix's tiles [begin, end) only (default: all) */);
int launch_stream_xs_stats(const uint32_t *win, size_t n_tiles, uint32_t *d_out2, hipStream_t s);
int launch_stream_len8(const uint32_t *off, size_t n_rows, uint8_t *len8, uint32_t *tbase, hipStream_t s);
size_t stream_tiles(size_t n_rows, int rows_per_thread);
int launch_stream_windows(const uint32_t *off, const uint32_t *col, size_t n_rows, uint32_t *win,
                          uint32_t *d_count, hipStream_t s);
int launch_stream_codes(const uint32_t *off, const uint32_t *col, const uint32_t *win, size_t n_rows, uint16_t *code,
                        hipStream_t s);
// K1s XD (spmv_stream_xd.hip): the code array as byte offsets into the tile's LDS stage of x, an unskewed product stage
int launch_spmv_stream_xd(int dtype, const void *val, const void *x, void *y, size_t n_rows, void *dot_partials, const uint16_t *scode,
                          const uint32_t *cwin, const uint8_t *len8, const uint32_t *tbase, const void *dot_lhs, hipStream_t s, int xs,
                          uint64_t tile_begin = 0, uint64_t tile_end = ~uint64_t(0), const void *dict = nullptr /* K1s XD-V: value dictionary */,
                          bool dict_high = false /* ... its indices sit in the codes' high spare bits alone */);
bool stream_value_dict_high(int dtype, uint32_t n, int xs);
// K1s XD-V (spmv_stream_xd.hip): the dictionary of val's distinct bit patterns (32 entries of the value type on the device; *count_out
// = 0 when there are more), the codes' spare bits filled with the entries' dictionary indices, and how many indices those bits can name
int stream_value_dict(int dtype, const void *val, size_t nnz, void *dict_out, uint32_t *count_out, hipStream_t s);
int launch_stream_value_codes(int dtype, const void *val, size_t nnz, const void *dict, uint32_t n, int xs, uint16_t *code, hipStream_t s);
uint32_t stream_value_dict_capacity(int xs);
int launch_stream_stage_codes(const uint32_t *off, const uint32_t *col, const uint32_t *win, size_t n_rows, uint32_t elem_bytes,
                              uint16_t *code, hipStream_t s);
int launch_stream_odd_rows(const uint8_t *len8, size_t n_padded, unsigned long long *d_out, hipStream_t s);
int launch_stream_max_tile(const uint32_t *off, size_t n_rows, size_t tile_rows, uint32_t *d_out, hipStream_t s);
// K2c (column-blocked CSR; each block runs the K1s kernel)
int launch_spmv_stream_block(int dtype, const uint32_t *off, const uint32_t *col, const void *val, const void *x, void *y,
                             size_t n_rows, size_t nnz_total, int rows_per_thread, bool single_pass, bool accumulate,
                             hipStream_t s);
int build_colblock(int dtype, const uint32_t *off, const uint32_t *col, const void *val, size_t n_rows, size_t nnz,
                   uint32_t shift, size_t n_blocks, uint32_t **off2_out, uint32_t **col2_out, void **val2_out,
                   hipStream_t s);
// K2f (column-blocked, one sweep over y; spmv_colfused.hip)
int build_colfused(int dtype, const uint32_t *off, const uint32_t *col, const void *val, size_t n_rows, size_t nnz, uint32_t shift,
                   size_t n_blocks, uint32_t rt, size_t *n_tiles_out, uint32_t **tile_row_out, uint32_t **seg_out, uint8_t **cnt_out,
                   uint32_t **col2_out, void **val2_out, bool *fits_out, hipStream_t s);
int launch_spmv_colfused(int dtype, uint32_t rt, const uint32_t *tile_row, size_t n_tiles, const uint32_t *seg, const uint8_t *cnt,
                         const uint32_t *col, const void *val, const void *x, void *y, size_t n_rows, size_t nnz, uint32_t n_blocks,
                         int device, hipStream_t s);
// K2s (a skewed matrix as a long-row and a short-row column-blocked matrix; spmv_colsplit.hip)
int build_colsplit(int dtype, const uint32_t *off, const uint32_t *col, const void *val, size_t n_rows, size_t nnz, uint32_t min_long,
                   size_t *n_long_out, size_t *nnz_long_out, uint32_t **long_rows_out, uint32_t **off_l_out, uint32_t **col_l_out, void **val_l_out,
                   uint32_t **off_s_out, uint32_t **col_s_out, void **val_s_out, hipStream_t s);
int launch_split_scatter(int dtype, const uint32_t *long_rows, const void *y_long, size_t n_long, void *y, hipStream_t s);
// on-device assembly (assemble.hip): add_to/set stream -> CRS; sort_row for all rows.  Device pointers.
int assemble_triplets(int dtype, size_t n, const uint32_t *rows, const uint32_t *cols, const void *vals, const uint8_t *ops,
                      bool reverse_rows, bool all_set, bool repeats_adjacent, size_t *n_rows_out, size_t *n_cols_out, size_t *nnz_out,
                      uint32_t **off_out, uint32_t **col_out, void **val_out, hipStream_t s);
int expand_rows(const uint32_t *off, size_t n_rows, uint32_t *rows_out, hipStream_t s);
// transpose_bucket.hip: the column lists of ColumnIter by the same two bucketed passes (col_ptr [n_cols + 1], entries [nnz]: device arrays)
int column_lists_bucketed(const uint32_t *off, const uint32_t *col, size_t n_rows, size_t n_cols, size_t nnz, uint32_t max_col, uint32_t *col_ptr,
                          uint32_t *entries, bool *done, hipStream_t s);
// transpose_bucket.hip: transposition by two bucketed passes; *done == false: not applicable, nothing produced
int transpose_bucketed(int dtype, const uint32_t *off, const uint32_t *col, const void *val, size_t n_rows, size_t nnz, uint32_t max_col,
                       uint32_t **off_out, uint32_t **col_out, void **val_out, size_t *n_rows_out, size_t *n_cols_out, bool *done, hipStream_t s);
int append_to_row(int dtype, uint32_t *off, uint32_t **col, void **val, size_t n_rows, size_t *nnz, size_t row, uint32_t column,
                  const void *value_host, hipStream_t s);
int column_info(const uint32_t *off, const uint32_t *col, size_t n_rows, size_t n_cols, size_t nnz, uint32_t max_col, uint32_t *rows,
                uint32_t *col_ptr, uint32_t *entries, hipStream_t s);
int sort_rows(int dtype, const uint32_t *off, uint32_t *col, void *val, size_t n_rows, size_t nnz, uint32_t max_col,
              hipStream_t s);
// structure predicates and SparseMatrix::prod (matops.hip).  Device pointers.
int crs_is_sorted(const uint32_t *off, const uint32_t *col, size_t n_rows, int *out, hipStream_t s);
int crs_is_symmetric(int dtype, const uint32_t *off, const uint32_t *col, const void *val, size_t n_rows, int *out, hipStream_t s);
int prod_crs(int dtype, const uint32_t *a_off, const uint32_t *a_col, const void *a_val, size_t a_rows, size_t a_nnz, uint32_t a_max_col,
             const uint32_t *b_off, const uint32_t *b_col, const void *b_val, size_t b_rows, size_t *n_rows_out, size_t *n_cols_out,
             size_t *nnz_out, uint32_t **off_out, uint32_t **col_out, void **val_out, hipStream_t s);
// K1r (LDS x-ring): inspector, host plan, kernel
struct RingPhase {
    uint32_t row_begin, row_end;  // rows of this phase (row_begin is a multiple of 64)
    uint32_t load_lo, load_hi;    // columns of x to add to the LDS ring before the phase (may be empty); band 0
    uint32_t use_ring;            // 0: the phase's column span exceeds the ring -> global gathers
    uint32_t band_lo[3], band_hi[3];  // banded ring (4 bands of a quarter of the ring each): loads of bands 1..3
};
constexpr int kRingPhaseWords = 11;
int launch_tile_span(const uint32_t *off, const uint32_t *col, size_t n_rows, size_t n_tiles, uint32_t *cmin,
                     uint32_t *cmax, hipStream_t s);
// col16 (optional): the low halves of the columns, padded with zeros to a multiple of 4 entries plus one chunk; ring
// phases then stream 2 instead of 4 bytes per column
int launch_spmv_ring2(int dtype, int lanes, int chunks, const uint32_t *off, const uint32_t *col, const uint16_t *col16,
                      const void *val, const void *x, void *y, size_t n_rows, size_t nnz, bool padded, unsigned n_blocks,
                      const uint32_t *phase_ptr, const RingPhase *phases, unsigned ring_entries, unsigned bands,
                      hipStream_t s, void *dot_partials = nullptr /* DOT form: y = lhs (read only), n_blocks + 1 partials of lhs . (A x) */,
                      unsigned block_begin = 0, unsigned block_end = ~0u /* the plan's row ranges [begin, end) only (default: all; not with the DOT form) */);
// banded ring (4 bands): per-tile column intervals (the K1s inspector over 64-row tiles) and the 16-bit ring slots
int launch_tile_intervals(const uint32_t *off, const uint32_t *col, size_t n_rows, size_t tile_rows, size_t max_width,
                          uint32_t *win, uint32_t *d_count, hipStream_t s);
int launch_ring_band_codes(const uint32_t *off, const uint32_t *col, const uint32_t *win, size_t n_rows, uint32_t S,
                           uint16_t *code, hipStream_t s);
int launch_narrow_columns(const uint32_t *col, size_t nnz, uint16_t *col16, size_t n_out, hipStream_t s);
// structure statistics / validation
struct CrsStats {
    uint32_t max_row_len;
    uint32_t max_col;
    uint32_t min_col_inv;  // ~(smallest column)
    uint32_t bad;  // bit0: offsets not monotone, bit1: off[0]!=0, bit2: off[n]!=nnz
};
int launch_crs_stats(const uint32_t *off, const uint32_t *col, size_t n_rows, size_t nnz, CrsStats *d_stats,
                     hipStream_t s);
// y = A x on stream s with the handle's kernel (capi.hip).  dot_partials (optional): when spmv_fused_dot_partials() > 0 the
// kernel also leaves that many partial sums of x.y there (K1s epilogue; square matrices) -- CG's / PCG's p.Ap for free
// K2t (spmv_tiled.hip)
void tiled_geometry(size_t n_rows, size_t n_cols, size_t nnz, int dtype, uint32_t *n_cb, uint32_t *rows_per_block, uint32_t *n_rb);
int tiled_build(::smh_crs *m);   // lazy; sets t2_ok
uint32_t tiled_slice_columns(int dtype);
int columns_within_n_cols(const ::smh_crs *m, const char *what);  // capi.hip: SMH_ERR_INDEX_RANGE when max_col >= n_cols
void tiled_free(::smh_crs *m);
int tiled_array(::smh_crs *m, int which, void *out, size_t capacity_bytes, size_t *bytes_out);
int launch_spmv_tiled(::smh_crs *m, const void *x, size_t x_len, void *y, hipStream_t s);
size_t spmv_fused_dot_partials(::smh_crs *m, size_t x_len, int variant, bool any_lhs = false);
int spmv_enqueue(::smh_crs *m, const void *x, size_t x_len, void *y, int variant, hipStream_t s, void *dot_partials = nullptr,
                 const void *dot_lhs = nullptr);
// products of a run of rows (capi.hip; used by par.hip to multiply a block's boundary rows before / after its interior ones)
int spmv_rows_granularity(::smh_crs *m, int variant, size_t *gran_out);
int spmv_enqueue_rows(::smh_crs *m, const void *x, size_t x_len, void *y, int variant, hipStream_t s, size_t row0, size_t row1,
                      void *dot_partials = nullptr, const void *dot_lhs = nullptr);
// ... of a SHORT run (a block's boundary rows), any row0 / row1: ring matrices through the plain lane-group kernel (same bits)
int spmv_enqueue_rows_short(::smh_crs *m, const void *x, size_t x_len, void *y, int variant, hipStream_t s, size_t row0, size_t row1);
// BLAS-1 (a_dev: scalar read from device memory when non-null, else `a`)
enum class Ew { Add, Sub, Scale, Axpy, Xpby, RSubInto };
int launch_ew(int dtype, Ew op, void *x, const void *y, size_t n, double a, const void *a_dev, hipStream_t s);
int launch_dot(int dtype, const void *x, const void *y, size_t n, void *partials, void *result_dev,
               hipStream_t s);
int launch_fold2(int dtype, const void *in, size_t count, void *partials, void *result_dev, hipStream_t s);  // *result = sum(in)
int launch_scale_values(int dtype, void *v, size_t n, double a, hipStream_t s);

}  // namespace smh

// ---- handles ------------------------------------------------------------------------------
struct smh_crs {
    int dtype = SMH_F32;
    int device = 0;
    size_t n_rows = 0, n_cols = 0, nnz = 0;
    size_t orphans = 0;  // entries the reference's container still holds but no row reaches (first-push quirk of a replay): 0 or 1
    uint32_t *d_off = nullptr;
    uint32_t *d_col = nullptr;
    void *d_val = nullptr;
    bool owns = true;
    hipStream_t stream = nullptr;
    // statistics
    uint32_t max_row_len = 0;
    uint32_t max_col = 0;
    uint32_t min_col = 0;
    uint32_t max_tile_entries = 0;  // most entries in any 256-row tile (K1s eligibility)
    uint32_t max_tile512_entries = 0;  // ... in any 512-row tile (K1s with two rows per thread)
    bool have_stats = false;
    int forced_lanes = 0;
    int forced_chunks = 0;
    // merge-path workspace (lazy)
    size_t n_tiles = 0;
    uint32_t *d_tile_row = nullptr, *d_tile_nz = nullptr, *d_carry_row = nullptr;
    void *d_carry_val = nullptr;
    int stream_rows_per_thread = 0;    // 0 automatic (2 when every 512-row tile fits), 1 force one
    // K1s 16-bit column codes (lazy; kept only when every tile has a description)
    bool stream_coded = false;         // inspected
    uint32_t *d_stream_cwin = nullptr; // 8 u32 per 256-row tile
    uint16_t *d_stream_code = nullptr; // one u16 per entry
    uint8_t *d_stream_len8 = nullptr;  // with codes and max_row_len <= 255: one byte per row (its length) ...
    uint32_t *d_stream_tbase = nullptr; // ... and one u32 per 256-row tile (its first entry)
    // K2c column-blocked copy (lazy)
    bool cb_built = false;
    uint32_t cb_forced_shift = 0;  // 0 automatic (2 MiB of x per block)
    uint32_t cb_shift = 0;       // block width = 2^cb_shift columns
    size_t cb_blocks = 0;
    int cb_rpt = 1;              // rows per thread of the K1s launches (tile = 256*cb_rpt rows)
    bool cb_single_pass = false; // no tile of any block exceeds the LDS stage
    uint32_t *d_cb_off = nullptr, *d_cb_col = nullptr;
    void *d_cb_val = nullptr;
    double span_fraction = 0.0;  // mean column span of a 64-row tile / n_cols (1: no locality at all)
    // K2s row-length split (lazy): two sub-handles (owned) + the rows of the long one
    bool split_built = false, split_ok = false;  // ok: the split was worth building (few long rows holding many entries)
    bool no_split = false;                       // this handle IS a part of a split: never split again
    smh_crs *split_long = nullptr, *split_short = nullptr;
    uint32_t *d_split_rows = nullptr;
    void *d_split_y = nullptr;
    size_t split_n_long = 0;
    hipStream_t split_stream = nullptr;  // the LONG part runs beside the SHORT one (fork / join by events)
    hipEvent_t split_fork = nullptr, split_join = nullptr;
    // K2f fused column-blocked copy (lazy)
    bool cf_built = false;       // the build was attempted (cf_ok: and the byte table could describe the matrix)
    bool cf_ok = false;
    uint32_t cf_shift = 0, cf_rt = 0;
    size_t cf_blocks = 0, cf_tiles = 0;
    uint32_t *d_cf_seg = nullptr, *d_cf_col = nullptr, *d_cf_tile_row = nullptr;
    uint8_t *d_cf_cnt = nullptr;
    void *d_cf_val = nullptr;
    // K2t 2-D tiled copy (lazy; spmv_tiled.hip): entries by column slice, within a slice by row
    bool t2_built = false, t2_ok = false;  // built: the build was attempted
    uint32_t t2_n_cb = 0, t2_n_rb = 0, t2_R = 0;  // column slices, row blocks, rows of the largest block
    uint64_t t2_tot = 0;                   // entries of the copy (slices padded to 8)
    void *d_t2_val = nullptr, *d_t2_prod = nullptr;  // values in copy order; the products of the last launch
    uint16_t *d_t2_code = nullptr, *d_t2_row = nullptr;  // column within the slice; row within the row block
    uint32_t *d_t2_tstart = nullptr;       // (n_rb + 1) x n_cb tile starts, relative to the slice
    uint32_t *d_t2_rbstart = nullptr;      // first row of each row block (n_rb + 1); blocks hold equal entry counts
    uint32_t *d_t3_cptr = nullptr;         // (round 3's form) first chunk of each slice (n_cb + 1)
    void *d_t3_chunk = nullptr;            // per chunk {where its product sums go, entries}
    uint32_t t3_n_chunks = 0, t3_max_slice_chunks = 0;
    bool t3_dups = false;                  // a chunk boundary cuts a (row, slice) pair somewhere: pass 2 checks for equal neighbours
    uint64_t t3_n_prod = 0;                // product slots (one per (row, slice, chunk), chunk shares padded to 16 bytes)
    // K1r plan (lazy)
    bool ring_planned = false;
    unsigned ring_blocks = 0;
    double ring_fraction = 0.0;  // share of rows whose gathers are served from the LDS ring
    unsigned ring_entries = smh::kRingEntries;  // ring size the plan was built for
    unsigned ring_bands = 1;                    // 1: one sliding window; 4: banded ring (needs d_col16 = ring slots)
    bool ring_bands_tried = false;              // the plan in place was built with the banded attempt allowed
    uint32_t *d_ring_win = nullptr;             // banded plan: the tiles' column intervals (8 u32 per 64-row tile)
    size_t ring_n_phases = 0;
    uint32_t *d_phase_ptr = nullptr;
    smh::RingPhase *d_phases = nullptr;
    int use_ring = -1;  // -1 automatic, 0 never, 1 always (when lanes <= 8), 2 always with the first K1r body
    int use_stream_xs = -1;  // K1s XS: -1 automatic (x beyond the L2s; the 4096-entry stage on f32 only), 0 never, 1 whenever the tiles allow
    uint32_t stream_xs_chunks = 0xFFFFFFFFu, stream_xs_end = 0;  // K1s XS: most x chunks a tile needs; largest x index + 1 they touch
    bool stream_direct = false;   // d_stream_code holds stage byte offsets (K1s XD) instead of column codes
    // K1s XD-V: ... and, in their spare bits, indices into a dictionary of the matrix's distinct values (the value array is then not read)
    bool stream_vdict = false;    // the codes carry the indices now (built for a stage of stream_vdict_xs * 1024 entries)
    int stream_vdict_xs = 0;
    int stream_dict_state = -1;   // -1 not looked at (or the values changed), 0 too many distinct values, 1 d_stream_dict / stream_dict_n are valid
    uint32_t stream_dict_n = 0;
    void *d_stream_dict = nullptr;  // 32 values
    int use_stream_vdict = -1;    // -1 automatic (whenever the values allow), 0 never
    int use_stream_direct = -1;   // K1s XD: -1 automatic (most rows of odd length), 0 never, 1 whenever x is staged
    uint64_t stream_odd_rows = 0; // rows of odd length (taken with the byte lengths)
    uint16_t *d_col16 = nullptr;  // K1r: 16-bit column array for the ring phases (lazy; null: not used)
    int use_col16 = -1;           // -1 automatic (when at least a quarter of the rows are ring rows), 0 never, 1 always
    // what the inspectors cost (smh_crs_prepare_stats): wall ms / pooled device bytes left allocated, create-time and prepare-time
    double create_ms = 0.0, prepare_ms = 0.0;
    long long create_bytes = 0, prepare_bytes = 0;
    // staging for the host-pointer API (lazy, reused)
    void *d_x = nullptr, *d_y = nullptr;
    size_t d_x_cap = 0, d_y_cap = 0;
};

struct smh_vec {
    int dtype = SMH_F32;
    int device = 0;
    size_t n = 0;
    void *d = nullptr;
    bool owns = true;
};

// ---- device memory goes through the caching layer of pool.hip (see there for why) ---------------------------------------
namespace smh {
hipError_t pool_malloc(void **out, size_t bytes);
hipError_t pool_free(void *p);
long long pool_thread_net_bytes();  // pooled bytes this thread has allocated minus freed so far (differences measure a build)
}  // namespace smh
#ifndef SMH_POOL_IMPL
#define hipMalloc(p, n) ::smh::pool_malloc((void **)(p), (n))
#define hipFree(p) ::smh::pool_free((void *)(p))
#endif
