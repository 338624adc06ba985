//! Raw declarations of include/sparsemat_hip.h (ABI version 1) -- the subset the shim uses.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_int, c_void};

#[repr(C)] pub struct smh_crs { _private: [u8; 0] }
#[repr(C)] pub struct smh_vec { _private: [u8; 0] }

pub const SMH_OK: c_int = 0;
pub const SMH_F32: c_int = 0;
pub const SMH_F64: c_int = 1;
pub const SMH_SPMV_AUTO: c_int = 0;

extern "C" {
    pub fn smh_abi_version() -> c_int;
    pub fn smh_last_error() -> *const c_char;
    pub fn smh_device_count(count_out: *mut c_int) -> c_int;
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
}
