/*
 * sparsemat_hip.h -- C ABI of libsparsemat_hip.so: the MI355X (gfx950) drop-in for the
 * CSR SpMV / BLAS-1 / CG hot path of the Rust crate lostinc0de/sparsemat.
 *
 * The host (Rust, C++ or Python) OWNS the CRS arrays; this library copies them to HBM
 * (or borrows device arrays the host already placed there) and runs hand-written HIP
 * kernels.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 * There is NO CPU fallback: without a HIP device every compute entry point fails with
 * SMH_ERR_NO_DEVICE.
 *
 * Conventions
 *   - every function returns an int status (SMH_OK == 0); a message for the calling
 *     thread's last failure is available from smh_last_error().  Nothing ever
 *     unwinds/throws across the ABI: the Rust shim turns a non-zero status into the
 *     reference's panic text.
 *   - host arrays are borrowed for the duration of the call only.
 *   - value type T is f32 or f64 (smh_dtype); index type I is u32 (the reference's
 *     SparseMatCRS<T,u32>; other index widths stay on the reference's CPU path).
 *   - `*_dev` entry points take DEVICE pointers and a hipStream_t (as void*; NULL is the
 *     HIP null stream, as everywhere in HIP) and are asynchronous; the others run on the
 *     handle's private stream and synchronise before returning.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * reference repository root).
 */
#ifndef SPARSEMAT_HIP_H
#define SPARSEMAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history.  3 (round 4): smh_crs_tiled_layout is back to its version-2 form of SIX out pointers (round 3 had appended a
 * seventh without a version bump: a caller built against the older header would have had the library write through a garbage
 * pointer) and the product count has its own entry point, smh_crs_tiled_products; REMOVED since version 2:
 * smh_crs_set_stream_windows and smh_crs_stream_windows (K1s's column windows are chosen by the inspector alone; the K1s-w kernel
 * they steered left the library); ADDED: smh_crs_tiled_products, smh_crs_prepare_stats,
 * smh_comm_ranks_seen, smh_rccl_version, smh_par_set_threads,
 * smh_crs_stream_value_dict, smh_crs_set_stream_value_dict.  A caller checks smh_abi_version() ==
 * SMH_ABI_VERSION once at load time (rust/src/lib.rs does). */
#define SMH_ABI_VERSION 3

/* ---- status codes ------------------------------------------------------------------ */
enum {
    SMH_OK = 0,
    SMH_ERR_DIM_MISMATCH = 1,  /* "Dimension mismatch" densevec.rs:53,62; "Matrix and vector size mismatch" linearsolver.rs:35 */
    SMH_ERR_NOT_SQUARE = 2,    /* "Matrix is not symmetric" linearsolver.rs:31 */
    SMH_ERR_INDEX_RANGE = 3,   /* a column index >= x.dim(): the implicit slice panic of densevec.rs:41 */
    SMH_ERR_INVALID = 4,       /* malformed CRS arrays / bad argument / misaligned device pointer */
    SMH_ERR_HIP = 5,           /* a HIP runtime call failed (message carries hipGetErrorString) */
    SMH_ERR_OOM = 6,           /* hipMalloc failed */
    SMH_ERR_NO_DEVICE = 7,     /* no HIP device: there is no CPU fallback */
    SMH_ERR_CAPACITY = 8,      /* nnz >= u32::MAX: "Maximum number of {} entries reached" sparsemat_crs.rs:82-84 */
    SMH_ERR_COMM = 9           /* an RCCL call failed (message carries ncclGetErrorString) */
};

typedef enum { SMH_F32 = 0, SMH_F64 = 1 } smh_dtype;

/* SpMV kernel families (DESIGN.md "Kernels") */
typedef enum {
    SMH_SPMV_AUTO = 0,   /* pick from the row-length statistics taken at create time        */
    SMH_SPMV_VECTOR = 1, /* K1: (sub-)wavefront per row, 16-B coalesced chunks, shfl reduce  */
    SMH_SPMV_MERGE = 2,  /* K2: merge-path tiles, LDS-staged products, deterministic fix-up  */
    SMH_SPMV_SEQ = 3,    /* one lane per row, storage order, mul then add: bit-exact vs the */
                         /* reference loop sparsematrix.rs:146-158 (checker, not fast)      */
    SMH_SPMV_STREAM = 4, /* K1s: short rows; dense entry stream, rounded products in LDS,   */
                         /* one thread folds a row in storage order: fast AND bit-exact     */
                         /* (stencil-like matrices stream 16-bit column codes, built once)  */
    SMH_SPMV_COLBLOCK = 5, /* K2c: columns without locality and x larger than an L2: the     */
                         /* device copy is split into blocks of 2^19 columns, y += A_b x    */
    SMH_SPMV_COLFUSED = 6, /* K2f: the same blocking in ONE sweep over y: a wave keeps the   */
                         /* sums of its rows in registers while all waves walk the column   */
                         /* blocks together (AUTO's choice when the rows are of similar     */
                         /* length; runs K2c when a (row, block) pair exceeds 255 entries)  */
    SMH_SPMV_COLSPLIT = 7, /* K2s: skewed rows without column locality (BASELINE C3): the rows */
                         /* of >= 64 entries as a compacted matrix through K2c, the rest     */
                         /* through K2f, y assembled from both (K2c when not worth it)      */
    SMH_SPMV_TILED = 8   /* K2t: columns without locality, two streaming passes over a 2-D   */
                         /* tiled copy: products against slices of x held in LDS, then one  */
                         /* wavefront per block of rows folds them (no gather leaves a CU)  */
} smh_spmv_variant;

typedef struct smh_crs smh_crs; /* device-resident SparseMatCRS<T,u32>  (sparsemat_crs.rs:9-17) */
typedef struct smh_vec smh_vec; /* device-resident DenseVec<T>          (densevec.rs:5-7)      */

/* ---- library / device --------------------------------------------------------------- */
int smh_abi_version(void);
const char *smh_last_error(void);            /* thread-local, never NULL */
const char *smh_status_string(int status);   /* the reference's panic text for a status */
int smh_device_count(int *count_out);        /* 0 devices is SMH_OK with *count_out == 0 */
int smh_set_device(int device);              /* device used by handles created afterwards (per thread) */
int smh_device_synchronize(void);

/* ---- SparseMatCRS<T,u32> --------------------------------------------------------------
 * smh_crs_create: replaces building a SparseMatCRS (fields sparsemat_crs.rs:9-17; the
 * only bulk constructor, from_sparsemat_index :24-50, is pub(crate)): copies
 * offset_rows[n_rows+1], columns[nnz], values[nnz] to HBM.  Columns of a row are taken in
 * storage order, unsorted and with duplicates allowed, exactly as iter_row (:102-110)
 * yields them.  Monotone offsets, offset_rows[0]==0 and offset_rows[n_rows]==nnz are always
 * checked (on the device); validate != 0 additionally requires columns < n_cols.          */
int smh_crs_create(smh_dtype dtype, size_t n_rows, size_t n_cols, size_t nnz,
                   const uint32_t *offset_rows, const uint32_t *columns, const void *values,
                   int validate, smh_crs **out);
/* Same, over arrays that already live in device memory (borrowed, not copied, must outlive
 * the handle; columns/values 16-byte aligned).  device = ordinal owning the pointers.   */
int smh_crs_create_dev(smh_dtype dtype, size_t n_rows, size_t n_cols, size_t nnz,
                       const uint32_t *offset_rows_dev, const uint32_t *columns_dev,
                       const void *values_dev, int validate, smh_crs **out);
/* ---- assembly on the device ----------------------------------------
 * smh_crs_assemble: replaces a stream of n_ops calls `mat.add_to(rows[k], cols[k], values[k])`
 * (ops == NULL or ops[k] == 0; sparsematrix.rs:231-233) / `mat.set(...)` (ops[k] == 1; :226-228)
 * on a SparseMatIndexList (get_mut sparsemat_indexlist.rs:158-164, push :45-53 over
 * indexlist.rs:62-83) followed by `to_crs()` (sparsemat_indexlist.rs:61-63 ->
 * sparsemat_crs.rs:24-50), with the identical result: n_rows = largest row + 1, n_cols =
 * largest column + 1, one entry per distinct (row, column) in order of first appearance
 * inside its row, value = left fold of the entry's operations in stream order from zero.
 * Bit-exact, including the values.  n_ops == 0 gives the empty matrix (no rows);
 * n_ops >= u32::MAX is SMH_ERR_CAPACITY (indexlist.rs:69).  The _dev form takes device arrays. */
int smh_crs_assemble(smh_dtype dtype, size_t n_ops, const uint32_t *rows, const uint32_t *cols,
                     const void *values, const uint8_t *ops, smh_crs **out);
int smh_crs_assemble_dev(smh_dtype dtype, size_t n_ops, const uint32_t *rows_dev,
                         const uint32_t *cols_dev, const void *values_dev, const uint8_t *ops_dev,
                         smh_crs **out);
/* smh_crs_replay: the same stream applied to a SparseMatCRS itself -- get_mut sparsemat_crs.rs:143-149
 * over find_index :54-67 / push :71-92 -- which is the container SparseMatrix::transpose and ::prod
 * fill through `set` (sparsematrix.rs:174-210).  Same entries and values as smh_crs_assemble, but
 * a row stores them in REVERSE order of first appearance (push inserts at the row's start, :85-87),
 * and the container's first-push quirk is honoured (:75-81: the first push leaves n_rows == 0):
 * if the second operation's row is smaller than the first one's, the first entry is orphaned
 * (Vec::resize truncates offset_rows) -- no row reaches it, so it is not part of the result
 * arrays although the reference's n_non_zero_entries() still counts it and n_cols covers it; if
 * the second operation names the same (row, column), the first keeps an entry of its own at the
 * end of the row; a single operation gives a matrix without rows.  Bit-exact. */
int smh_crs_replay(smh_dtype dtype, size_t n_ops, const uint32_t *rows, const uint32_t *cols,
                   const void *values, const uint8_t *ops, smh_crs **out);
int smh_crs_replay_dev(smh_dtype dtype, size_t n_ops, const uint32_t *rows_dev,
                       const uint32_t *cols_dev, const void *values_dev, const uint8_t *ops_dev,
                       smh_crs **out);
/* SparseMatrix::transpose (sparsematrix.rs:174-184) of a SparseMatCRS: `ret.set(j, i, val)` for
 * every entry in row-major storage order, replayed as above (so row j of the result lists the
 * source rows in DESCENDING order of visit, duplicates of (i, j) keep the last value, and the
 * quirk drops the very first entry when the second stored entry has a smaller column). */
int smh_crs_transpose(const smh_crs *a, smh_crs **out);
/* Diagnostics: how this thread's last smh_crs_transpose was carried out -- 0 the general route (stable device-wide sort
 * by target row through the replay machinery), 1 two bucketed passes (csrc/transpose_bucket.hip: matrices with local
 * structure -- bands, stencils, block orderings -- without repeated entries; the same result bit for bit, about twice
 * as fast).  SMH_TRANSPOSE_BUCKETED=0 (environment) keeps every call on the general route. */
int smh_last_transpose_route(void);
/* ColumnIter::assemble_column_info (sparsemat_crs.rs:180-191) as flat arrays: rows[k] = row of
 * entry k (the reference's `rows` vector), and its per-column IndexList (entry indices in storage
 * order, indexlist.rs:62-83) as col_ptr[n_cols + 1] / entries[nnz]: iter_col(j) (:193-204, next
 * :211-219) yields (rows[e], values[e]) for e in entries[col_ptr[j] .. col_ptr[j+1]).  A column
 * index >= n_cols is SMH_ERR_INDEX_RANGE.  The _dev form writes device arrays. */
int smh_crs_column_info(const smh_crs *m, uint32_t *rows, uint32_t *col_ptr, uint32_t *entries);
int smh_crs_column_info_dev(const smh_crs *m, uint32_t *rows_dev, uint32_t *col_ptr_dev,
                            uint32_t *entries_dev);
/* SparseMatrix::prod (sparsematrix.rs:186-210), self = a, rhs = b, both SparseMatCRS: the matrix the
 * reference's loops leave in a fresh SparseMatCRS -- entry (i, j) = the sum over rhs' column j in
 * storage order of val(i, row) * val_rhs with the same order of additions (bit-exact), sums that
 * compare == 0 dropped (:203), rows in descending column order (ascending `set` calls into a CRS),
 * n_rows / n_cols = largest kept row / column + 1; one kept sum leaves a matrix without rows (the
 * first-push quirk).  The reference's Err("Dimension mismatch") (:188-190: a.n_rows != b.n_cols or
 * a.n_cols != b.n_rows) is SMH_ERR_DIM_MISMATCH.  No column tables are needed on rhs. */
int smh_crs_prod(const smh_crs *a, const smh_crs *b, smh_crs **out);
/* is_symmetric (sparsematrix.rs:212-222: get(j, i) != val for some stored entry -> false; get takes
 * the first match in storage order) and is_sorted (:251-271): *out = 1 / 0. */
int smh_crs_is_symmetric(const smh_crs *m, int *out);
int smh_crs_is_sorted(const smh_crs *m, int *out);
/* Sortable::sort_row (sparsemat_crs.rs:163-172) for every row: ascending columns, stable
 * (duplicates of a column keep their storage order).  Sorts the handle's arrays in place. */
int smh_crs_sort_rows(smh_crs *m);
int smh_crs_destroy(smh_crs *m);
/* host mutated values: re-upload.  values_host == NULL: the device values (borrowed arrays of
 * smh_crs_create_dev) were changed in place -- derived copies (K2c) are rebuilt on next use */
int smh_crs_update_values(smh_crs *m, const void *values_host);
int smh_crs_download(const smh_crs *m, uint32_t *offset_rows, uint32_t *columns, void *values);
/* n_rows :124-126, n_cols :128-130, n_non_zero_entries :132-134 */
size_t smh_crs_n_rows(const smh_crs *m);
size_t smh_crs_n_cols(const smh_crs *m);
size_t smh_crs_nnz(const smh_crs *m);
/* entries the reference's SparseMatCRS would still hold in columns / values although no row reaches them (the first-push
 * quirk of smh_crs_replay / smh_crs_transpose / smh_crs_prod, see smh_crs_replay): 0 or 1.  The reference's
 * n_non_zero_entries() (= columns.len(), sparsemat_crs.rs:132-134) and density() count them: they equal
 * smh_crs_nnz() + smh_crs_orphans(); the arrays of this handle hold smh_crs_nnz() entries. */
size_t smh_crs_orphans(const smh_crs *m);
int smh_crs_dtype(const smh_crs *m);
int smh_crs_max_row_len(const smh_crs *m, uint32_t *out);
/* smallest / largest column index stored (0/0 for a matrix without entries): the part of x a
 * row block references -- what SparseMatPar exchanges between ranks */
int smh_crs_col_range(const smh_crs *m, uint32_t *min_out, uint32_t *max_out);
/* SparseMatrix::scale (sparsemat_crs.rs:153-157): values *= a */
int smh_crs_scale(smh_crs *m, double a);
/* variant AUTO resolves to; lanes_out = lanes per row of the VECTOR kernel */
int smh_crs_resolved_variant(const smh_crs *m, int *variant_out, int *lanes_out);
/* override the VECTOR kernel's lanes-per-row (1,2,4,...,64; 0 = automatic) */
int smh_crs_set_vector_lanes(smh_crs *m, int lanes);
/* K1s XS, the STREAM kernel for coded single-pass tiles with the tile's column intervals of x staged in LDS by 16-byte loads
 * issued before the tile's own: mode -1 = automatic (x beyond the L2s; 2048 entries of x per tile, 4096 on f32),
 * 0 = never, 1 = whenever every tile's intervals fit a stage.  smh_crs_stream_layout reports what a STREAM launch of this
 * matrix uses: 16-bit column codes, byte row lengths, the two-chunk body (no tile above 2045 entries) and the 16-byte chunks
 * of x per thread staged (0 none, 2 or 4); a launch still falls back to gathers from memory when x is not 16-byte aligned
 * (the staged pieces are aligned 16-byte blocks of x: the last one may be read up to 12 bytes past x_len, inside the block --
 * and page -- that holds x's last entry; those values are never used).  Same bits as K1s in every form. */
int smh_crs_set_stream_xs(smh_crs *m, int mode);
int smh_crs_stream_layout(smh_crs *m, int *coded_out, int *byte_lengths_out, int *small_tiles_out, int *xs_chunks_out);
/* K1s XD-V, the value dictionary of the CSR-stream kernel (spmv_stream_xd.hip): when the matrix holds at most 32 distinct values (bit
 * patterns; 16 with the 4096-entry stage of x) -- every constant-coefficient stencil, unweighted graph Laplacians, adjacency matrices --
 * and the kernel runs with stage offsets (K1s XD), the spare bits of the 16-bit codes name each entry's value in a dictionary and the
 * value array is not read by the product: 2 bytes per entry instead of 6 (f32) / 10 (f64).  Same products, same order of additions:
 * still bit for bit SparseMatrix::mvp (sparsematrix.rs:146-158).  Built by smh_crs_prepare / the first product; smh_crs_update_values
 * makes the library look again, smh_crs_scale scales the dictionary.  *n_values_out: entries of the dictionary in use, 0 = the form is
 * not active; values_out (optional): room for 32 values.  smh_crs_set_stream_value_dict: -1 automatic, 0 never (SMH_STREAM_VDICT=0). */
int smh_crs_stream_value_dict(smh_crs *m, int *n_values_out, void *values_out);
int smh_crs_set_stream_value_dict(smh_crs *m, int mode);
/* K1s XD: the XS kernel with its per-entry address arithmetic moved into the build -- the 16-bit code array holds the byte offset
 * of x[col] inside the tile's LDS stage instead of (interval, offset), the products are staged unskewed, 16 bytes per store.  Same
 * bits again.  The unskewed stage collides on rows of even length, hence mode -1 = automatic (when x is staged and most rows have
 * an odd length: stencils with a diagonal), 0 = never, 1 = whenever x is staged.  A launch whose x cannot be staged (alignment,
 * length) streams the u32 columns instead.  smh_crs_stream_direct: does a STREAM launch of this matrix run in that form? */
int smh_crs_set_stream_direct(smh_crs *m, int mode);
int smh_crs_stream_direct(smh_crs *m, int *direct_out);
/* 16-B chunks per lane and pass of the pipelined VECTOR body (1..3; 0 = automatic): a lane group
 * covers 4*lanes*chunks entry slots of its row per pass */
int smh_crs_set_vector_chunks(smh_crs *m, int chunks);
/* K1r, the pipelined VECTOR kernel with an LDS-resident sliding window of x: mode -1 =
 * automatic (always; rows whose column span exceeds the ring gather from L2),
 * 0 = plain K1, 1 = same as -1.  The ring holds 16384 columns (64 KiB of f32 with two 512-thread
 * blocks per CU, 128 KiB of f64 with one 1024-thread block per CU).  When at least a quarter of
 * the rows run in ring phases the handle also keeps a u16 array with the low halves of the
 * columns (2 bytes per entry, built on first use): ring phases only need `column mod 16384` and
 * stream that array instead of the u32 columns (same arithmetic, bitwise identical results). */
int smh_crs_set_ring(smh_crs *m, int mode);
/* the K1r phase plan (integer structure, invariants checked in tests): phase_ptr_out needs
 * n_blocks+1 entries, phases_out 11 u32 per phase {row_begin,row_end,load_lo,load_hi,use_ring,
 * lo1,lo2,lo3,hi1,hi2,hi3} (load ranges of band 0 and, in banded plans, of bands 1..3);
 * call first with NULL arrays to get the sizes.                                           */
/* columns the ring of this matrix's plan holds: 16384, or 32768 for f32 matrices whose rows need the
 * wider window (128 KiB of LDS, one 1024-thread block per CU) */
int smh_crs_ring_entries(smh_crs *m, uint32_t *out);
/* 1: one sliding window (slot = column mod ring size).  4: banded ring for rows that reference a few narrow
 * column intervals far apart (stencils): band k holds the k-th interval of the tiles, slot = k * S +
 * column mod S, S = ring size / 4.  intervals_out (optional, banded plans only): 8 u32 per 64-row tile,
 * {lo0,hi0,..,lo3,hi3} sorted and disjoint; all zero = tile without entries, [1,0) = not describable. */
int smh_crs_ring_bands(smh_crs *m, uint32_t *bands_out, uint32_t *intervals_out);
int smh_crs_ring_plan(smh_crs *m, uint32_t *n_blocks_out, size_t *n_phases_out,
                      double *ring_fraction_out, int *active_out, uint32_t *phase_ptr_out,
                      uint32_t *phases_out);

/* K2c, the column-blocked copy (built on first use; integer structure, checked bit-exact in tests):
 * block b holds the entries with column >> shift == b, as a CSR over all rows whose offsets
 * offsets_out[b*(n_rows+1) + r] are absolute positions into columns_out / values_out [nnz]; inside a
 * (row, block) pair the entries keep their storage order.  Call with NULL arrays for the sizes.
 * span_fraction_out (may be NULL): mean column span of a 64-row tile / n_cols, the locality
 * statistic AUTO uses (COLBLOCK when it exceeds 0.25 and x holds at least 4 MiB).  More than 128
 * blocks (n_cols > 2^26 at the default width) is SMH_ERR_INVALID.                              */
int smh_crs_set_colblock_shift(smh_crs *m, uint32_t shift); /* block = 2^shift columns; 0 = automatic (19) */
int smh_crs_colblock(smh_crs *m, uint32_t *shift_out, size_t *n_blocks_out, int *rows_per_thread_out,
                     double *span_fraction_out, uint32_t *offsets_out, uint32_t *columns_out,
                     void *values_out);

/* K2f, the fused column-blocked copy (built on first use; integer structure, checked bit-exact in tests): rows are
 * grouped into tiles -- tile t = rows [tile_rows[t], tile_rows[t+1]) = 64 lanes x h consecutive rows each, h <=
 * rows_per_lane chosen greedily in row order so that a tile holds at most 1.25x the entries of a mean full-height tile
 * (h = 1 when even 64 rows exceed that) -- and columns into blocks of 2^shift; the entries are sorted by (tile, block,
 * row, storage order).  segments_out[t * n_blocks + b] = position of the first entry of (tile t, block b)
 * (n_tiles * n_blocks + 1 values), counts_out[((t * n_blocks + b) * 64 + lane) * rows_per_lane + j] = entries of row
 * tile_rows[t] + lane * h + j in block b (one byte each; zero for j >= h), columns_out / values_out [nnz] the permuted
 * entries.  Call with NULL arrays for the sizes.  *fits_out == 0: a (row, block) pair holds more than 255 entries,
 * there is no such copy (SMH_SPMV_COLFUSED then runs K2c).  Block width: smh_crs_set_colblock_shift, default 2^18
 * columns; more than 255 blocks is SMH_ERR_INVALID. */
int smh_crs_colfused(smh_crs *m, int *fits_out, uint32_t *shift_out, size_t *n_blocks_out, uint32_t *rows_per_lane_out,
                     size_t *n_tiles_out, uint32_t *tile_rows_out, uint32_t *segments_out, uint8_t *counts_out,
                     uint32_t *columns_out, void *values_out);

/* K2t, the 2-D tiled copy (built on first use of SMH_SPMV_TILED; spmv_tiled.hip): the entries by column slice of
 * *slice_columns_out (16384) columns, within a slice by (row, storage order), cut into chunks of at most 64 E entries (E = 4 for
 * f32, 2 for f64: 16 bytes of values per lane) that start at row boundaries where one lies within 16 entries of the nominal
 * start; every chunk owns 64 E slots of the copy (*copy_entries_out slots in total, the unused ones zero).  Pass 1 multiplies a
 * chunk against the slice of x held in LDS and folds the entries of one row into one product sum, so a (row, slice) pair
 * yields one product (more only where a chunk boundary cuts it): *n_products_out product slots (each chunk's share padded to
 * 16 bytes).  The rows are cut into *n_row_blocks_out blocks of consecutive rows holding the same number of products -- about
 * 174 (f32) / 60 (f64) per (slice, row block) tile -- and at most *rows_per_block_out rows (the largest block); pass 2: one
 * wavefront per row block adds its tiles' products to wave-private row sums, slice by slice.  The sum of a row is therefore
 * taken per slice (slices ascending; inside a slice the fold order documented at the top of spmv_tiled.hip): within the parity
 * bound, bitwise reproducible, not bit-identical to the reference's storage-order fold.  A row's result can be -0.0 where the
 * reference (which starts every row from +0.0, sparsematrix.rs:149) returns +0.0 -- only when every product of the row is -0.0;
 * equal in value.  Memory: the copy (sizeof(T) + 2 bytes per slot), the products of the last launch (sizeof(T) + 2 bytes per
 * product slot) and a table of (n_row_blocks + 1) x n_slices u32 (beyond 4 GiB: SMH_ERR_INVALID).  Any out pointer may be NULL. */
int smh_crs_tiled_layout(smh_crs *m, uint32_t *n_slices_out, uint32_t *slice_columns_out, uint32_t *rows_per_block_out,
                         uint32_t *n_row_blocks_out, size_t *copy_entries_out);
/* ... and the number of product slots pass 1 writes / pass 2 reads (its own entry point: see the ABI history above). */
int smh_crs_tiled_products(smh_crs *m, size_t *n_products_out);

/* The arrays of that plan, for inspection (tests restate the build and the passes' summation order from them): `which` =
 * 0 first chunk of each slice (u32, n_slices + 1) | 1 per chunk {first product slot, entries} (2 x u32) | 2 the copy's 16-bit
 * codes (bit 15: same row as the entry before; bits 0-13: column within the slice) | 3 the copy's values | 4 per product slot:
 * its row relative to its row block, or rows_per_block for padding (u16) | 5 first row of each row block (u32, n_row_blocks + 1)
 * | 6 tile table: (n_row_blocks + 1) x n_slices first product slots (u32) | 7 the products of the last launch.  out == NULL: only
 * *bytes_out is set. */
int smh_crs_tiled_array(smh_crs *m, int which, void *out, size_t capacity_bytes, size_t *bytes_out);

/* K2s, the row-length split (built on first use): *split_out == 1 when the handle keeps its rows of >= *min_long_out
 * entries as a compacted sub-matrix (row i of it = row long_rows_out[i], n_long of them; global columns) and all rows
 * with those emptied as a second one -- both are ordinary handles owned by `m` (inspect, do not destroy) -- and 0 when
 * that is not worth it (long rows more than a quarter of the rows, or holding less than a quarter of the entries). */
int smh_crs_colsplit(smh_crs *m, int *split_out, uint32_t *min_long_out, size_t *n_long_out, uint32_t *long_rows_out,
                     smh_crs **long_out, smh_crs **short_out);

/* SparseMatrix::mvp (sparsematrix.rs:146-158) == `A * v` (Mul, sparsematrix.rs:435-443):
 * y[0..n_rows) = A.x.  x_len is x.dim(); a column index >= x_len is SMH_ERR_INDEX_RANGE
 * (the reference panics in densevec.rs:41).  y has exactly n_rows entries (ret.set grows
 * it one row at a time, vector.rs:40-42 / densevec.rs:44-49); empty rows give 0.        */
int smh_crs_spmv(smh_crs *m, const void *x_host, size_t x_len, void *y_host, int variant);
/* Builds now what the first product with `variant` would build lazily (K1r phase plan, merge-path
 * table, K2c blocked copy: device allocations and a synchronisation): call it before capturing
 * smh_crs_spmv_dev into a hipGraph of your own.  Never needed for correctness otherwise.      */
int smh_crs_prepare(smh_crs *m, int variant);
/* What the inspectors cost -- the reference has no set-up step at all (iter_row, sparsemat_crs.rs:102-110, is a slice zip), so an
 * honest comparison carries it.  Runs smh_crs_prepare(m, variant) (a no-op when already built) and reports, accumulated over the
 * handle's life: *prepare_ms_out = host wall time of the create-time inspection (statistics passes, K1r inspector) plus every
 * build smh_crs_prepare performed, device-synchronised; *derived_bytes_out = the device memory those builds left allocated
 * beside the CRS arrays (16-bit column arrays, code / phase tables, K2t's copy, product buffer and tile table ...; blocks below
 * 1 MiB are not counted).  A plan built lazily by a first product that was NOT preceded by smh_crs_prepare is not in the figures. */
int smh_crs_prepare_stats(smh_crs *m, int variant, double *prepare_ms_out, size_t *derived_bytes_out);
int smh_crs_spmv_dev(smh_crs *m, const void *x_dev, size_t x_len, void *y_dev, int variant,
                     void *stream);
/* SparseMatrix::inner_prod (sparsematrix.rs:161-171): lhs^T A rhs = sum over all entries of
 * lhs[i] * a_ij * rhs[j].  An lhs that ends at or before the last row holding entries, or a column
 * index >= rhs_len, is SMH_ERR_INDEX_RANGE (the reference panics in densevec.rs:41; rows without
 * entries never index lhs, so trailing empty rows may lie beyond lhs_len, as in the reference).
 * Computed as dot(lhs, A rhs): the dot rides the SpMV's epilogue where the kernel has one, then a
 * two-stage deterministic reduction (tolerance-level parity, like dot). */
int smh_crs_inner_prod(smh_crs *m, const void *lhs_host, size_t lhs_len, const void *rhs_host,
                       size_t rhs_len, int variant, double *out);
/* merge-path tile table (integer structure, checked bit-exact in tests): tile t starts at
 * (row_out[t], nnz_out[t]); arrays need smh_crs_merge_tiles()+1 entries.                 */
size_t smh_crs_merge_tiles(const smh_crs *m);
size_t smh_crs_merge_tile_items(const smh_crs *m);
int smh_crs_merge_table(smh_crs *m, uint32_t *row_out, uint32_t *nnz_out);

/* ---- DenseVec<T> ------------------------------------------------------------------------
 * Vector::with_capacity/from_vec (densevec.rs:24-34), dim (:36-38), iter (:20-22)      */
int smh_vec_create(smh_dtype dtype, size_t n, smh_vec **out);                  /* zeros */
int smh_vec_from_host(smh_dtype dtype, size_t n, const void *host, smh_vec **out);
int smh_vec_wrap_dev(smh_dtype dtype, size_t n, void *dev_ptr, smh_vec **out); /* borrowed */
int smh_vec_destroy(smh_vec *v);
int smh_vec_upload(smh_vec *v, const void *host);
int smh_vec_download(const smh_vec *v, void *host);
size_t smh_vec_dim(const smh_vec *v);
int smh_vec_dtype(const smh_vec *v);
void *smh_vec_data(const smh_vec *v); /* device pointer */
int smh_vec_copy(smh_vec *dst, const smh_vec *src);                            /* clone() */
/* Vector::add / sub / scale (densevec.rs:51-58, :60-67, :69-73; += -= *= sugar :76-96):
 * x += y, x -= y over the first y.dim() entries; SMH_ERR_DIM_MISMATCH iff x.dim() < y.dim() */
int smh_vec_add(smh_vec *x, const smh_vec *y);
int smh_vec_sub(smh_vec *x, const smh_vec *y);
int smh_vec_scale(smh_vec *x, double a);
/* y += round(a*x)  -- `*x += p.clone() * alpha` linearsolver.rs:47 (densevec.rs:121-130, :76-81) */
int smh_vec_axpy(smh_vec *y, double a, const smh_vec *x);
/* p = round(b*p) + r -- `p.scale(beta); p.add(&r)` linearsolver.rs:58-59 */
int smh_vec_xpby(smh_vec *p, double b, const smh_vec *r);
/* Vector::inner_prod (vector.rs:50-53; `v * w` densevec.rs:133-140), norm_squared (:56-58):
 * deterministic two-stage tree reduction in T, result widened to double (T::into::<f64>) */
int smh_vec_dot(const smh_vec *x, const smh_vec *y, double *out);
int smh_vec_norm_squared(const smh_vec *x, double *out);
int smh_vec_norm(const smh_vec *x, double *out); /* vector.rs:61-63 */
/* The same kernels on raw device pointers with DEVICE-resident scalars, asynchronous on `stream`:
 * the building blocks of the row-partitioned CG, whose scalars are all-reduced on the device.
 * dot: *result_dev = x.y (type T; scratch_dev needs smh_blas_dot_scratch_bytes()).
 * axpy: y += round(*a_dev * x) (linearsolver.rs:47,49).  xpby: p = round(*b_dev * p) + r (:58-59). */
int smh_blas_dot_dev(smh_dtype dtype, const void *x_dev, const void *y_dev, size_t n, void *result_dev,
                     void *scratch_dev, void *stream);
size_t smh_blas_dot_scratch_bytes(void);
int smh_blas_axpy_dev(smh_dtype dtype, void *y_dev, const void *a_dev, const void *x_dev, size_t n, void *stream);
int smh_blas_xpby_dev(smh_dtype dtype, void *p_dev, const void *b_dev, const void *r_dev, size_t n, void *stream);
/* SparseMatrix::mvp on device vectors; y is resized semantics-free: y.dim() must be n_rows */
int smh_crs_spmv_vec(smh_crs *m, const smh_vec *x, smh_vec *y, int variant);
/* SparseMatrix::inner_prod on device vectors (sparsematrix.rs:161-171) */
int smh_crs_inner_prod_vec(smh_crs *m, const smh_vec *lhs, const smh_vec *rhs, int variant, double *out);

/* ---- ConjugateGradient::solve (linearsolver.rs:27-61) -----------------------------------
 * Device-resident: SpMV + fused updates + reductions, scalars (alpha, beta, r.r) stay in
 * HBM; the stop rule `sqrt(f64(r.r)) < tol`, tested before the beta update (:52-54), is
 * evaluated on the device every iteration and polled by the host every `check_every`
 * iterations (kernels of later iterations are no-ops once it fired, so x is exactly the x
 * of the iteration that converged).  tol/iter_max: the private fields :12-15 (Default
 * 1e-12 / 10000, :17-24).  Status: SMH_ERR_NOT_SQUARE (:30-32), SMH_ERR_DIM_MISMATCH
 * (:33-36).  iters_out = loop bodies entered (a body that breaks counts), rr_out = last
 * r.norm_squared() as f64.                                                               */
int smh_cg_solve(smh_crs *m, const void *b_host, size_t b_len, void *x_host_inout, size_t x_len,
                 double tol, size_t iter_max, int variant, size_t *iters_out, double *rr_out);
int smh_cg_solve_vec(smh_crs *m, const smh_vec *b, smh_vec *x, double tol, size_t iter_max,
                     int variant, size_t check_every, size_t *iters_out, double *rr_out);

/* Jacobi-preconditioned CG -- an EXTENSION (the reference has no
 * preconditioner): the recurrence of ConjugateGradient::solve with z = r / diag(A), diag_i =
 * get(i, i) (first match in storage order); same guards, statuses, stop rule (on r.r, tested
 * before the beta update) and outputs as smh_cg_solve.  A zero or absent diagonal entry is
 * SMH_ERR_INVALID.  Host vectors; x is updated in place. */
int smh_pcg_jacobi_solve(smh_crs *m, const void *b_host, size_t b_len, void *x_host_inout,
                         size_t x_len, double tol, size_t iter_max, int variant, size_t *iters_out,
                         double *rr_out);

/* ---- SparseMatPar<SparseMatCRS<T,u32>> (sparsemat_par.rs:12-35, 86-140) over the GPUs of one node ----
 * The reference keeps n_blocks sub-matrices of R = n_rows / n_blocks rows (with_sub_matrices :20-28;
 * integer division :21); block b = global rows [b R, (b+1) R) with local row ids and GLOBAL column
 * ids; its (commented-out) mvp_par :37-68 multiplies every block against the shared rhs and places
 * the results at b * R -- a row partition plus an all-gather.  Here block b lives on a GPU, the local
 * product is the library's SpMV, and the exchange runs INSIDE the library, device to device over
 * xGMI: an in-place RCCL all-gather of the y slices (ncclAllGather at recvbuff + b R; a ragged last
 * block adds one ncclBroadcast of its tail), or -- when the blocks' column intervals say that less
 * than half of the vector is referenced -- grouped ncclSend / ncclRecv of exactly the referenced
 * entries ("window"; a banded matrix moves a halo).  Two ways to own the blocks:
 *   one process, all blocks   smh_par_create (splits host arrays) / smh_par_adopt (blocks already on
 *                             their devices): ncclCommInitAll over the blocks' devices when they are
 *                             distinct (backend RCCL), or -- backend PEER, also the only choice when
 *                             blocks share a device -- one pull kernel per block that reads the peers'
 *                             slices directly (peer access; hipMemcpyPeerAsync where there is none).
 *   one process per GPU       smh_comm_unique_id on one rank, smh_comm_create (ncclCommInitRank) on
 *                             all, smh_par_create_rank around each rank's own block: rank = block id.
 * The last block takes the remainder rows -- the reference clamps the block id to n_blocks (:32), one
 * past the last block, and panics there.  R == 0 is SMH_ERR_INVALID (the reference divides by zero in
 * :32).  Vectors of the partitioned path are smh_par_vec: one full-length device buffer per local
 * block; block b OWNS entries [b R, (b+1) R), the rest of a buffer is valid as far as the last
 * upload / exchange made it so.  All smh_par_*_dev / _vec calls are asynchronous on the blocks'
 * private streams (ordered among themselves); smh_par_synchronize waits for them.              */
typedef struct smh_par smh_par;
typedef struct smh_par_vec smh_par_vec;
typedef struct smh_comm smh_comm; /* one rank of an RCCL communicator (one process per GPU) */

enum { SMH_EXCHANGE_NONE = 0, SMH_EXCHANGE_ALLGATHER = 1, SMH_EXCHANGE_WINDOW = 2, SMH_EXCHANGE_AUTO = 3 };
enum { SMH_PAR_BACKEND_AUTO = 0, SMH_PAR_BACKEND_PEER = 1, SMH_PAR_BACKEND_RCCL = 2 };
#define SMH_COMM_ID_BYTES 128

/* ncclGetUniqueId -> id_out[SMH_COMM_ID_BYTES]: call on ONE rank, hand the bytes to the others by any
 * host-side means (a file, a socket, the launcher's store), then every rank calls smh_comm_create
 * (ncclCommInitRank) with its rank on its device (smh_set_device first).  Collective. */
int smh_comm_unique_id(void *id_out);
int smh_comm_create(const void *id, int n_ranks, int rank, smh_comm **out);
int smh_comm_destroy(smh_comm *c);
int smh_comm_size(const smh_comm *c);
int smh_comm_rank(const smh_comm *c);
/* What RCCL itself reports (a first run on a node answers "did the library see N ranks, and which RCCL" by itself):
 * *count_out = ncclCommCount of the rank's communicator, *device_out = ncclCommCuDevice (either may be NULL);
 * *version_out = ncclGetVersion (e.g. 22203; needs no communicator). */
int smh_comm_ranks_seen(const smh_comm *c, int *count_out, int *device_out);
int smh_rccl_version(int *version_out);
/* helpers for a host without a collective library of its own (bench, tests): a barrier across the
 * ranks (device work of this rank's device drained first), and max over ranks of one double */
int smh_comm_barrier(smh_comm *c);
int smh_comm_max_f64(smh_comm *c, double *value_inout);

int smh_par_create(smh_dtype dtype, size_t n_blocks, const int *device_ids, size_t n_rows,
                   size_t n_cols, const uint32_t *offset_rows, const uint32_t *columns,
                   const void *values, int validate, smh_par **out);
/* WHERE the rows are cut ("nnz-balanced split points ... as an option for skewed matrices").  SMH_SPLIT_ROWS: the
 * reference's arithmetic -- R = n_rows / n_blocks rows per block (sparsemat_par.rs:21), the remainder to the last one.
 * SMH_SPLIT_NNZ: block k starts at the first row whose entries begin at or beyond k nnz / n_blocks (every block keeps at least one
 * row): equal work per GPU for skewed matrices.  The partition is then a table of n_blocks + 1 row boundaries (smh_par_split);
 * smh_par_get_block_and_row_id searches it, the window plan and the interior rows follow it, and the all-gather exchange -- whose
 * in-place ncclAllGather needs equal counts -- becomes one grouped launch of n_blocks ncclBroadcast, each slice from its owner.
 * smh_par_adopt_split / smh_par_create_rank_split take blocks cut elsewhere: split_rows = the n_blocks + 1 boundaries (NULL: the
 * reference's partition); row_begin = this rank's first row ((size_t)-1: the reference's partition; all ranks or none). */
enum { SMH_SPLIT_ROWS = 0, SMH_SPLIT_NNZ = 1 };
int smh_par_create_split(smh_dtype dtype, size_t n_blocks, const int *device_ids, size_t n_rows,
                         size_t n_cols, const uint32_t *offset_rows, const uint32_t *columns,
                         const void *values, int validate, int split_mode, smh_par **out);
int smh_par_adopt_split(size_t n_blocks, smh_crs *const *blocks, size_t n_rows, const size_t *split_rows, smh_par **out);
int smh_par_create_rank_split(smh_comm *comm, size_t n_rows, smh_crs *block, size_t row_begin, smh_par **out);
/* rows_out[k] .. rows_out[k + 1]: the rows of block k (n_blocks + 1 values, whichever way the partition was cut) */
int smh_par_split(const smh_par *p, size_t *rows_out);
/* blocks that already live on their devices (device-born inputs): blocks[b] = rows [b R, (b+1) R) of
 * an n_rows-row matrix, all with the same n_cols and dtype.  Borrowed: they must outlive the handle. */
int smh_par_adopt(size_t n_blocks, smh_crs *const *blocks, size_t n_rows, smh_par **out);
/* one process per GPU: this rank's block (borrowed) of an n_rows-row matrix; n_blocks = comm size,
 * block id = comm rank.  Collective (the ranks exchange their column intervals). */
int smh_par_create_rank(smh_comm *comm, size_t n_rows, smh_crs *block, smh_par **out);
int smh_par_destroy(smh_par *p);
size_t smh_par_n_blocks(const smh_par *p);
size_t smh_par_n_local_blocks(const smh_par *p);  /* n_blocks, or 1 with one process per GPU */
size_t smh_par_n_rows(const smh_par *p);
size_t smh_par_n_cols(const smh_par *p);          /* :109-115 */
size_t smh_par_nnz(const smh_par *p);             /* n_non_zero_entries :117-123 (local blocks) */
size_t smh_par_rows_per_block(const smh_par *p);  /* n_rows_sub_matrix :13 */
/* local block i: its device matrix (owned by the par handle unless adopted), global row range, device */
int smh_par_block(const smh_par *p, size_t block, smh_crs **crs_out, size_t *row_begin,
                  size_t *row_end, int *device);
/* the private stream local block i's work is enqueued on (to bracket it with events of the caller's) */
int smh_par_block_stream(const smh_par *p, size_t block, void **stream_out);
/* get_block_and_row_id (:31-35), clamped to the last block */
int smh_par_get_block_and_row_id(const smh_par *p, size_t row, size_t *block_out, size_t *row_out);
int smh_par_scale(smh_par *p, double a);          /* :135-139 */
/* exchange backend of a one-process handle (AUTO: RCCL when every block has a device of its own, else
 * PEER); SMH_PAR_BACKEND=peer|rccl in the environment overrides AUTO.  RCCL with blocks sharing a
 * device is SMH_ERR_INVALID; a per-rank handle is always RCCL. */
int smh_par_set_backend(smh_par *p, int backend);
int smh_par_backend(const smh_par *p);
/* The exchange beside the product (on by default; SMH_PAR_OVERLAP=0 in the environment switches it off).  The blocks of the
 * reference's design are independent (sparsemat_par.rs:54-64: every block multiplies on its own, results at b R), so the order
 * in which a block's rows are multiplied is free of semantics.  When the exchange of a call is a WINDOW and the block's kernel
 * can be launched by runs of rows (K1s: 256-row tiles, K1r: the plan's row ranges -- same kernels, same arithmetic per row, so
 * results are bit-identical either way), smh_par_spmv_dev multiplies the rows other blocks reference on a second stream per
 * block, the exchange behind them, and the block's INTERIOR rows on its main stream meanwhile; smh_par_cg_solve_vec starts the
 * exchange of p on the second stream and behind it the rows that wait for it, and multiplies the interior rows (which reference no
 * other block's entries) on the main stream meanwhile.  (Ring matrices: the few boundary rows go through the plain lane-group
 * kernel, whose results are the ring kernel's bit for bit.)  smh_par_interior:
 * the rows [*row_begin, *row_end) (local) of local block i that its kernel for `variant` treats as interior (equal: none -- the
 * call is then not split for that block). */
int smh_par_set_overlap(smh_par *p, int on);
/* One-process handles with several local blocks: every block's runtime calls (kernel launches, event records and waits) are issued
 * by a host thread of its own -- block 0's by the caller -- instead of one thread issuing them block after block (the reference's
 * intended design is independent blocks, sparsemat_par.rs:54-64).  mode: -1 automatic (off unless SMH_PAR_THREADS=1: with the blocks
 * on ONE device, the only set-up measured so far, the runtime serialises its callers and nothing is gained), 0 one issuing thread,
 * 1 a thread per block.  Streams, events, kernels and their order per block do not depend on it: results are
 * bit for bit the same.  Synchronises the handle's streams. */
int smh_par_set_threads(smh_par *p, int mode);
int smh_par_interior(const smh_par *p, size_t local_block, int variant, size_t *row_begin, size_t *row_end);
/* what SMH_EXCHANGE_AUTO resolves to for this matrix (WINDOW when the matrix is square and no block
 * receives half of the vector or more, else ALLGATHER; NONE for one block) and the largest number of
 * entries any block receives in a window exchange */
int smh_par_exchange_mode(const smh_par *p, int mode, int *resolved_out, size_t *max_recv_out);
/* The plan arithmetic on its own (pure host code, no device): block `block` of n_blocks over n_rows
 * rows, needs[q] != 0 iff block q has entries, referencing columns [lo[q], hi[q]] (inclusive).
 * recv_begin/end[q] = the range of q's slice this block receives in a window exchange, send_begin/
 * end[q] = the range of its own slice it sends to q (empty: begin == end == 0).  auto_mode_out /
 * max_recv_out as smh_par_exchange_mode (for a square matrix). */
int smh_par_plan(size_t n_blocks, size_t n_rows, const uint8_t *needs, const uint32_t *lo,
                 const uint32_t *hi, size_t block, size_t *recv_begin, size_t *recv_end,
                 size_t *send_begin, size_t *send_end, int *auto_mode_out, size_t *max_recv_out);
/* ... for a partition given by its n_blocks + 1 row boundaries (NULL: as smh_par_plan) */
int smh_par_plan_split(size_t n_blocks, size_t n_rows, const size_t *split_rows, const uint8_t *needs, const uint32_t *lo,
                       const uint32_t *hi, size_t block, size_t *recv_begin, size_t *recv_end,
                       size_t *send_begin, size_t *send_end, int *auto_mode_out, size_t *max_recv_out);

/* distributed DenseVec: n entries in one device buffer per local block.  upload replicates a host
 * vector into every local buffer; download collects the OWNED slices (n == n_rows) of the local
 * blocks at their global offsets (with one process per GPU: this rank's rows only);
 * download_block copies local block i's whole buffer; ptr gives its device pointer. */
int smh_par_vec_create(smh_par *p, size_t n, smh_par_vec **out);
int smh_par_vec_destroy(smh_par_vec *v);
size_t smh_par_vec_dim(const smh_par_vec *v);
int smh_par_vec_upload(smh_par_vec *v, const void *host);
int smh_par_vec_download(const smh_par_vec *v, void *host);
int smh_par_vec_download_block(const smh_par_vec *v, size_t local_block, void *host);
int smh_par_vec_ptr(const smh_par_vec *v, size_t local_block, void **dev_ptr_out);
/* the intended mvp_par (:37-68), device resident: every local block writes y[b R ..) = A_b x into ITS
 * slice of y (x must be valid on the columns the block references), then ONE exchange of y
 * (SMH_EXCHANGE_*): afterwards y is valid everywhere (ALLGATHER) or on every block's own slice plus the
 * columns it references (WINDOW) -- ready to be the next x.  x != y.  Asynchronous. */
int smh_par_spmv_dev(smh_par *p, const smh_par_vec *x, smh_par_vec *y, int variant, int exchange);
/* the exchange on its own: v (n == n_rows) holds every block's owned slice */
int smh_par_exchange(smh_par *p, smh_par_vec *v, int mode);
int smh_par_synchronize(smh_par *p);
/* y = A x on host vectors (the trait-default mvp through iter_row :86-89): blocks run concurrently,
 * each uploading only x[min column .. max column] of its rows.  One-process handles only. */
int smh_par_spmv(smh_par *p, const void *x_host, size_t x_len, void *y_host, int variant);
/* ConjugateGradient::solve (linearsolver.rs:27-61) with x, r, p, Ap distributed by rows: per iteration
 * ONE window exchange of p, the local SpMV, and two cross-block folds (p.Ap, r.r) of device-resident
 * scalars -- each block reduces its rows to one value, the values meet (peer-visible slots / a
 * 1-element ncclAllGather) and every block folds the same n_blocks values in the same fixed order, so
 * all blocks take the same alpha, beta and stop decision without the host; the host polls the stop
 * flag every check_every iterations (0: default) like smh_cg_solve.  _vec: b and x hold the owned
 * slices (x in/out).  The host-vector form takes full-length vectors (with one process per GPU only
 * this rank's rows of b are read and of x written).  Same statuses and outputs as smh_cg_solve. */
int smh_par_cg_solve_vec(smh_par *p, const smh_par_vec *b, smh_par_vec *x, double tol, size_t iter_max,
                         int variant, size_t check_every, size_t *iters_out, double *rr_out);
int smh_par_cg_solve(smh_par *p, const void *b_host, size_t b_len, void *x_host_inout, size_t x_len,
                     double tol, size_t iter_max, int variant, size_t *iters_out, double *rr_out);

/* ---- synthetic workloads (bench / test support, DESIGN.md "Synthetic inputs") ------------
 * Counter-based generators writing straight into device memory so 10M..80M-row inputs never
 * cross PCIe.  pattern: 0 banded-stratified (ascending), 1 uniform (draw order),
 * 2 contiguous band (k consecutive columns around the diagonal), 3 window (k <= 64 distinct
 * columns drawn without replacement from [row - 4096, row + 4096], ascending).              */
int smh_synth_x(smh_dtype dtype, uint64_t seed, size_t begin, size_t n, void *x_dev, void *stream);
int smh_synth_fixed(smh_dtype dtype, uint64_t seed, int pattern, size_t n, uint32_t k,
                    size_t row_begin, size_t row_end, uint32_t *offset_rows_dev,
                    uint32_t *columns_dev, void *values_dev, void *stream);
int smh_synth_powerlaw_cdf(uint32_t kmax, double alpha, uint32_t *cdf_host);
int smh_synth_powerlaw_lengths(uint64_t seed, size_t row_begin, size_t row_end, uint32_t kmax,
                               const uint32_t *cdf_host, uint32_t *lengths_host);
int smh_synth_fill(smh_dtype dtype, uint64_t seed, size_t n_cols, size_t row_begin, size_t row_end,
                   const uint32_t *offset_rows_dev, uint32_t *columns_dev, void *values_dev,
                   void *stream);
/* 7-point Laplacian on an nx*ny*nz grid (natural ordering, diag 6, off-diag -1), rows
 * [row_begin,row_end), offsets rebased.  offsets need rows+1, columns/values nnz entries
 * (call with NULL arrays to get nnz_out only).                                            */
int smh_synth_laplace3d(smh_dtype dtype, size_t nx, size_t ny, size_t nz, size_t row_begin,
                        size_t row_end, uint32_t *offset_rows_dev, uint32_t *columns_dev,
                        void *values_dev, size_t *nnz_out, void *stream);

/* ---- device memory kept by the library ---------------------------------------------------
 * Everything the library allocates on a device (matrix arrays, plans, scratch of assembly /
 * transpose / prod, vectors, smh_dev_alloc) goes through a caching layer (csrc/pool.hip): freed
 * blocks of 1 MiB and more stay with the library, at most SMH_POOL_MAX_BYTES per device (environment,
 * default 16 GiB, 0 = off), because a fresh hipMalloc of a few GB takes 1.2-1.4 s every few calls
 * on this platform.  smh_pool_trim returns everything kept to the HIP runtime (call it before
 * another library in the process needs the memory); smh_pool_stats reports kept and in-use bytes
 * of pooled blocks.  Not part of the reference's interface (its Vec allocations have no such cost). */
int smh_pool_trim(void);
int smh_pool_stats(size_t *kept_bytes_out, size_t *live_bytes_out);

/* ---- raw device memory helpers for hosts without a HIP binding (the Rust shim) ---------- */
int smh_dev_alloc(size_t bytes, void **out);
int smh_dev_free(void *p);
int smh_dev_upload(void *dst_dev, const void *src_host, size_t bytes);
int smh_dev_download(void *dst_host, const void *src_dev, size_t bytes);
int smh_dev_memset(void *dst_dev, int value, size_t bytes, void *stream); /* asynchronous */
/* streams and timing events (hipStream_t / hipEvent_t as void*): what bench.py brackets every launch with */
int smh_stream_create(void **stream_out);
int smh_stream_destroy(void *stream);
int smh_stream_synchronize(void *stream);
int smh_event_create(void **event_out);
int smh_event_destroy(void *event);
int smh_event_record(void *event, void *stream);
int smh_event_elapsed_ms(void *start, void *stop, float *ms_out); /* waits for `stop` */

#ifdef __cplusplus
}
#endif
#endif /* SPARSEMAT_HIP_H */
