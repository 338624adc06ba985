//! Raw declarations of include/sparsemat_hip.h (ABI version 3) -- the subset the shim uses.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_int, c_void};

#[repr(C)] pub struct smh_crs { _private: [u8; 0] }
#[repr(C)] pub struct smh_vec { _private: [u8; 0] }
#[repr(C)] pub struct smh_par { _private: [u8; 0] }
#[repr(C)] pub struct smh_par_vec { _private: [u8; 0] }
#[repr(C)] pub struct smh_comm { _private: [u8; 0] }

pub const SMH_OK: c_int = 0;
pub const SMH_ERR_DIM_MISMATCH: c_int = 1;
pub const SMH_F32: c_int = 0;
pub const SMH_F64: c_int = 1;
pub const SMH_SPMV_AUTO: c_int = 0;
pub const SMH_EXCHANGE_NONE: c_int = 0;
pub const SMH_EXCHANGE_ALLGATHER: c_int = 1;
pub const SMH_EXCHANGE_WINDOW: c_int = 2;
pub const SMH_EXCHANGE_AUTO: c_int = 3;
pub const SMH_COMM_ID_BYTES: usize = 128;
pub const SMH_SPLIT_ROWS: c_int = 0;  // the reference's partition: R = n_rows / n_blocks rows per block (sparsemat_par.rs:21)
pub const SMH_SPLIT_NNZ: c_int = 1;   // opt-in: blocks of equal entry counts

extern "C" {
    pub fn smh_abi_version() -> c_int;
    pub fn smh_last_error() -> *const c_char;
    pub fn smh_device_count(count_out: *mut c_int) -> c_int;
    pub fn smh_pool_trim() -> c_int;  // return the device memory the library keeps (csrc/pool.hip) to the HIP runtime
    pub fn smh_pool_stats(kept_bytes_out: *mut usize, live_bytes_out: *mut usize) -> c_int;
    pub fn smh_crs_create(dtype: c_int, n_rows: usize, n_cols: usize, nnz: usize,
                          offset_rows: *const u32, columns: *const u32, values: *const c_void,
                          validate: c_int, out: *mut *mut smh_crs) -> c_int;
    pub fn smh_crs_destroy(m: *mut smh_crs) -> c_int;
    pub fn smh_crs_update_values(m: *mut smh_crs, values_host: *const c_void) -> c_int;
    pub fn smh_crs_spmv(m: *mut smh_crs, x_host: *const c_void, x_len: usize, y_host: *mut c_void,
                        variant: c_int) -> c_int;
    pub fn smh_cg_solve(m: *mut smh_crs, b_host: *const c_void, b_len: usize, x_host_inout: *mut c_void,
                        x_len: usize, tol: c_double, iter_max: usize, variant: c_int,
                        iters_out: *mut usize, rr_out: *mut c_double) -> c_int;
    // extension (not in the reference): Jacobi-preconditioned CG, same contract as smh_cg_solve
    pub fn smh_pcg_jacobi_solve(m: *mut smh_crs, b_host: *const c_void, b_len: usize, x_host_inout: *mut c_void,
                                x_len: usize, tol: c_double, iter_max: usize, variant: c_int,
                                iters_out: *mut usize, rr_out: *mut c_double) -> c_int;
    pub fn smh_crs_inner_prod(m: *mut smh_crs, lhs_host: *const c_void, lhs_len: usize, rhs_host: *const c_void,
                              rhs_len: usize, variant: c_int, out: *mut c_double) -> c_int;
    // add_to (ops[k] == 0 / ops null) or set (ops[k] == 1) stream -> the CRS `to_crs()` would return
    pub fn smh_crs_assemble(dtype: c_int, n_ops: usize, rows: *const u32, cols: *const u32, values: *const c_void,
                            ops: *const u8, out: *mut *mut smh_crs) -> c_int;
    // the same stream applied to a SparseMatCRS itself (push prepends, first-push quirk): what transpose / prod fill
    pub fn smh_crs_replay(dtype: c_int, n_ops: usize, rows: *const u32, cols: *const u32, values: *const c_void,
                          ops: *const u8, out: *mut *mut smh_crs) -> c_int;
    pub fn smh_crs_transpose(a: *const smh_crs, out: *mut *mut smh_crs) -> c_int;
    pub fn smh_crs_set_stream_xs(m: *mut smh_crs, mode: c_int) -> c_int;  // K1s with x staged in LDS: -1 auto, 0 never, 1 whenever the tiles allow
    pub fn smh_crs_set_stream_direct(m: *mut smh_crs, mode: c_int) -> c_int;  // K1s XD (stage offsets instead of column codes): -1 auto, 0 never, 1 whenever x is staged
    pub fn smh_crs_stream_direct(m: *mut smh_crs, direct_out: *mut c_int) -> c_int;
    pub fn smh_crs_stream_layout(m: *mut smh_crs, coded_out: *mut c_int, byte_lengths_out: *mut c_int, small_tiles_out: *mut c_int, xs_chunks_out: *mut c_int) -> c_int;
    pub fn smh_last_transpose_route() -> c_int;  // 0 general (device-wide sort), 1 two bucketed passes: diagnostics only
    pub fn smh_crs_orphans(m: *const smh_crs) -> usize;
    pub fn smh_crs_prod(a: *const smh_crs, b: *const smh_crs, out: *mut *mut smh_crs) -> c_int;
    pub fn smh_crs_is_symmetric(m: *const smh_crs, out: *mut c_int) -> c_int;
    pub fn smh_crs_is_sorted(m: *const smh_crs, out: *mut c_int) -> c_int;
    pub fn smh_crs_column_info(m: *const smh_crs, rows: *mut u32, col_ptr: *mut u32, entries: *mut u32) -> c_int;
    pub fn smh_crs_sort_rows(m: *mut smh_crs) -> c_int;
    // SparseMatPar<SparseMatCRS<T,u32>>, one process: block b on device device_ids[b] (null: b mod device count)
    pub fn smh_par_create(dtype: c_int, n_blocks: usize, device_ids: *const c_int, n_rows: usize, n_cols: usize,
                          offset_rows: *const u32, columns: *const u32, values: *const c_void, validate: c_int,
                          out: *mut *mut smh_par) -> c_int;
    // the same with the rows cut by entry count (SMH_SPLIT_NNZ): the partition becomes a table of n_blocks + 1 row boundaries
    pub fn smh_par_create_split(dtype: c_int, n_blocks: usize, device_ids: *const c_int, n_rows: usize, n_cols: usize,
                                offset_rows: *const u32, columns: *const u32, values: *const c_void, validate: c_int,
                                split_mode: c_int, out: *mut *mut smh_par) -> c_int;
    pub fn smh_par_split(p: *const smh_par, rows_out: *mut usize) -> c_int;
    // exchange overlapped with the product of the rows that reference no other block (default on; bit-identical either way)
    pub fn smh_par_set_overlap(p: *mut smh_par, on: c_int) -> c_int;
    pub fn smh_par_interior(p: *const smh_par, local_block: usize, variant: c_int, row_begin: *mut usize, row_end: *mut usize) -> c_int;
    pub fn smh_par_destroy(p: *mut smh_par) -> c_int;
    pub fn smh_par_spmv(p: *mut smh_par, x_host: *const c_void, x_len: usize, y_host: *mut c_void, variant: c_int) -> c_int;
    pub fn smh_par_cg_solve(p: *mut smh_par, b_host: *const c_void, b_len: usize, x_host_inout: *mut c_void, x_len: usize,
                            tol: c_double, iter_max: usize, variant: c_int, iters_out: *mut usize,
                            rr_out: *mut c_double) -> c_int;
    // device-resident form: distributed vectors (one full-length buffer per local block), y = A x + ONE exchange of y
    // inside the library (in-place ncclAllGather / grouped ncclSend+ncclRecv of the referenced window / peer reads)
    pub fn smh_par_vec_create(p: *mut smh_par, n: usize, out: *mut *mut smh_par_vec) -> c_int;
    pub fn smh_par_vec_destroy(v: *mut smh_par_vec) -> c_int;
    pub fn smh_par_vec_upload(v: *mut smh_par_vec, host: *const c_void) -> c_int;
    pub fn smh_par_vec_download(v: *const smh_par_vec, host: *mut c_void) -> c_int;
    pub fn smh_par_spmv_dev(p: *mut smh_par, x: *const smh_par_vec, y: *mut smh_par_vec, variant: c_int, exchange: c_int) -> c_int;
    pub fn smh_par_exchange(p: *mut smh_par, v: *mut smh_par_vec, mode: c_int) -> c_int;
    pub fn smh_par_synchronize(p: *mut smh_par) -> c_int;
    pub fn smh_par_cg_solve_vec(p: *mut smh_par, b: *const smh_par_vec, x: *mut smh_par_vec, tol: c_double, iter_max: usize,
                                variant: c_int, check_every: usize, iters_out: *mut usize, rr_out: *mut c_double) -> c_int;
    // one process per GPU: ncclGetUniqueId on one rank, ncclCommInitRank on all, the rank's own block
    pub fn smh_comm_unique_id(id_out: *mut c_void) -> c_int;
    pub fn smh_comm_create(id: *const c_void, n_ranks: c_int, rank: c_int, out: *mut *mut smh_comm) -> c_int;
    pub fn smh_comm_destroy(c: *mut smh_comm) -> c_int;
    pub fn smh_par_create_rank(comm: *mut smh_comm, n_rows: usize, block: *mut smh_crs, out: *mut *mut smh_par) -> c_int;
    // row_begin = this rank's first row (usize::MAX: the reference's partition; all ranks or none)
    pub fn smh_par_create_rank_split(comm: *mut smh_comm, n_rows: usize, block: *mut smh_crs, row_begin: usize, out: *mut *mut smh_par) -> c_int;
    pub fn smh_crs_n_rows(m: *const smh_crs) -> usize;
    pub fn smh_crs_n_cols(m: *const smh_crs) -> usize;
    pub fn smh_crs_nnz(m: *const smh_crs) -> usize;
    pub fn smh_crs_download(m: *const smh_crs, offset_rows: *mut u32, columns: *mut u32, values: *mut c_void) -> c_int;
}
