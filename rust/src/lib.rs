//! Drop-in glue between lostinc0de/sparsemat and libsparsemat_hip.so (MI355X / gfx950).
//!
//! NOT COMPILED HERE (no Rust toolchain in the build image).  It shows exactly what a maintainer adds
//! to the crate: a device handle cached next to the CRS arrays, an override of the trait default
//! `SparseMatrix::mvp` (src/sparsematrix.rs:146-158) inside `impl SparseMatrix for SparseMatCRS`
//! (src/sparsemat_crs.rs:95-158), and a device-resident `ConjugateGradient::solve`
//! (src/linearsolver.rs:27-61).  Non-zero statuses are re-raised as the reference's panics.
pub mod ffi;

use std::ffi::CStr;
use std::os::raw::{c_int, c_void};

/// Value types the HIP path accelerates; everything else stays on the reference's CPU loop.
pub trait HipValue: Copy {
    const DTYPE: c_int;
}
impl HipValue for f32 { const DTYPE: c_int = ffi::SMH_F32; }
impl HipValue for f64 { const DTYPE: c_int = ffi::SMH_F64; }

fn check(status: c_int) {
    if status != ffi::SMH_OK {
        // same text as the reference's panic!() ("Matrix is not symmetric", "Dimension mismatch", ...)
        let msg = unsafe { CStr::from_ptr(ffi::smh_last_error()) }.to_string_lossy().into_owned();
        panic!("{}", msg);
    }
}

/// The header this shim was written against (include/sparsemat_hip.h: `#define SMH_ABI_VERSION 3`).  Checked once per process
/// before the first handle is made: a library of another ABI version must not be called through these declarations.
pub const SMH_ABI_VERSION: c_int = 3;

fn check_abi() {
    static ONCE: std::sync::Once = std::sync::Once::new();
    ONCE.call_once(|| {
        let got = unsafe { ffi::smh_abi_version() };
        assert_eq!(got, SMH_ABI_VERSION, "libsparsemat_hip.so has ABI version {}, this shim expects {}", got, SMH_ABI_VERSION);
    });
}

/// Device-resident copy of a `SparseMatCRS<T, u32>`; the Rust side keeps owning the Vecs.
pub struct DeviceCrs {
    handle: *mut ffi::smh_crs,
    n_rows: usize,
}

impl DeviceCrs {
    /// `offset_rows`, `columns`, `values` are the private fields of SparseMatCRS
    /// (src/sparsemat_crs.rs:12-14), borrowed for the duration of the call.
    pub fn new<T: HipValue>(n_rows: usize, n_cols: usize, offset_rows: &[u32], columns: &[u32], values: &[T]) -> Self {
        assert_eq!(offset_rows.len(), n_rows + 1);
        assert_eq!(columns.len(), values.len());
        check_abi();
        let mut handle = std::ptr::null_mut();
        check(unsafe {
            ffi::smh_crs_create(T::DTYPE, n_rows, n_cols, values.len(), offset_rows.as_ptr(), columns.as_ptr(),
                                values.as_ptr() as *const c_void, 1, &mut handle)
        });
        DeviceCrs { handle, n_rows }
    }

    /// `SparseMatrix::mvp` for `V = DenseVec<T>`: returns a new Vec with `n_rows` entries.
    pub fn mvp<T: HipValue + Default>(&self, x: &[T]) -> Vec<T> {
        let mut y = vec![T::default(); self.n_rows];
        check(unsafe {
            ffi::smh_crs_spmv(self.handle, x.as_ptr() as *const c_void, x.len(), y.as_mut_ptr() as *mut c_void,
                              ffi::SMH_SPMV_AUTO)
        });
        y
    }

    /// `ConjugateGradient::solve`: `x` is updated in place; returns (iterations, r.r).
    pub fn cg_solve<T: HipValue>(&self, b: &[T], x: &mut [T], tol: f64, iter_max: usize) -> (usize, f64) {
        let (mut iters, mut rr) = (0usize, 0f64);
        check(unsafe {
            ffi::smh_cg_solve(self.handle, b.as_ptr() as *const c_void, b.len(), x.as_mut_ptr() as *mut c_void,
                              x.len(), tol, iter_max, ffi::SMH_SPMV_AUTO, &mut iters, &mut rr)
        });
        (iters, rr)
    }
}

impl DeviceCrs {
    /// `SparseMatrix::inner_prod` (src/sparsematrix.rs:161-171): lhs^T A rhs.
    pub fn inner_prod<T: HipValue>(&self, lhs: &[T], rhs: &[T]) -> f64 {
        let mut out = 0f64;
        check(unsafe {
            ffi::smh_crs_inner_prod(self.handle, lhs.as_ptr() as *const c_void, lhs.len(), rhs.as_ptr() as *const c_void,
                                    rhs.len(), ffi::SMH_SPMV_AUTO, &mut out)
        });
        out
    }

    /// `SparseMatrix::transpose` (src/sparsematrix.rs:174-184) for `Self = SparseMatCRS`: the arrays the reference's
    /// `ret.set(j, i, val)` loop leaves in a fresh SparseMatCRS (rows reversed, first-push quirk), built on the device.
    pub fn transpose(&self) -> DeviceCrs {
        let mut handle = std::ptr::null_mut();
        check(unsafe { ffi::smh_crs_transpose(self.handle, &mut handle) });
        DeviceCrs { handle, n_rows: unsafe { ffi::smh_crs_n_rows(handle) } }
    }

    /// `SparseMatrix::prod` (src/sparsematrix.rs:186-210) with the reference's `Result`: `Err("Dimension mismatch")`
    /// when `self.n_rows() != rhs.n_cols() || self.n_cols() != rhs.n_rows()`; no column tables needed on `rhs`.
    pub fn prod(&self, rhs: &DeviceCrs) -> Result<DeviceCrs, String> {
        let mut handle = std::ptr::null_mut();
        let status = unsafe { ffi::smh_crs_prod(self.handle, rhs.handle, &mut handle) };
        if status == ffi::SMH_ERR_DIM_MISMATCH {
            return Err(unsafe { CStr::from_ptr(ffi::smh_last_error()) }.to_string_lossy().into_owned());
        }
        check(status);
        Ok(DeviceCrs { handle, n_rows: unsafe { ffi::smh_crs_n_rows(handle) } })
    }

    /// `is_symmetric` (src/sparsematrix.rs:212-222) / `is_sorted` (:263-271) on the device copy.
    pub fn is_symmetric(&self) -> bool {
        let mut out: c_int = 0;
        check(unsafe { ffi::smh_crs_is_symmetric(self.handle, &mut out) });
        out != 0
    }
    pub fn is_sorted(&self) -> bool {
        let mut out: c_int = 0;
        check(unsafe { ffi::smh_crs_is_sorted(self.handle, &mut out) });
        out != 0
    }

    /// `ColumnIter::assemble_column_info` (src/sparsemat_crs.rs:180-191) as arrays `(rows, col_ptr, entries)`:
    /// `iter_col(j)` yields `(rows[e], values[e])` for `e in entries[col_ptr[j]..col_ptr[j+1]]`.
    pub fn column_info(&self) -> (Vec<u32>, Vec<u32>, Vec<u32>) {
        let nnz = unsafe { ffi::smh_crs_nnz(self.handle) };
        let n_cols = unsafe { ffi::smh_crs_n_cols(self.handle) };
        let (mut rows, mut col_ptr, mut entries) = (vec![0u32; nnz.max(1)], vec![0u32; n_cols + 1], vec![0u32; nnz.max(1)]);
        check(unsafe { ffi::smh_crs_column_info(self.handle, rows.as_mut_ptr(), col_ptr.as_mut_ptr(), entries.as_mut_ptr()) });
        rows.truncate(nnz);
        entries.truncate(nnz);
        (rows, col_ptr, entries)
    }

    /// `sort_row(i)` for every row (src/sparsemat_crs.rs:163-172), on the device copy.
    pub fn sort_rows(&mut self) {
        check(unsafe { ffi::smh_crs_sort_rows(self.handle) });
    }

    /// The CRS arrays as stored on the device: (n_cols, offset_rows, columns, values) -- what
    /// `SparseMatCRS::from_raw_parts` takes back on the Rust side.
    pub fn raw_parts<T: HipValue + Default>(&self) -> (usize, Vec<u32>, Vec<u32>, Vec<T>) {
        let nnz = unsafe { ffi::smh_crs_nnz(self.handle) };
        let (mut off, mut col, mut val) = (vec![0u32; self.n_rows + 1], vec![0u32; nnz], vec![T::default(); nnz]);
        check(unsafe { ffi::smh_crs_download(self.handle, off.as_mut_ptr(), col.as_mut_ptr(), val.as_mut_ptr() as *mut c_void) });
        (unsafe { ffi::smh_crs_n_cols(self.handle) }, off, col, val)
    }
}

/// Write-only recorder with the call surface of `SparseMatIndexList` assembly (src/sparsematrix.rs:226-233):
/// `add_to` / `set` append in O(1) (no walk of the row's list, src/sparsemat_indexlist.rs:29-42), `to_crs`
/// (src/sparsemat_indexlist.rs:61-63) assembles on the device -- same CRS, bit for bit: rows in order of first
/// appearance, duplicates folded in call order.
pub struct TripletStream<T: HipValue> {
    rows: Vec<u32>,
    cols: Vec<u32>,
    vals: Vec<T>,
    ops: Vec<u8>,
}

impl<T: HipValue> TripletStream<T> {
    pub fn with_capacity(cap: usize) -> Self {
        TripletStream { rows: Vec::with_capacity(cap), cols: Vec::with_capacity(cap), vals: Vec::with_capacity(cap), ops: Vec::with_capacity(cap) }
    }
    pub fn add_to(&mut self, i: usize, j: usize, val: T) { self.push(i, j, val, 0) }
    pub fn set(&mut self, i: usize, j: usize, val: T) { self.push(i, j, val, 1) }
    fn push(&mut self, i: usize, j: usize, val: T, op: u8) {
        self.rows.push(i as u32); self.cols.push(j as u32); self.vals.push(val); self.ops.push(op);
    }
    pub fn to_crs(&self) -> DeviceCrs {
        let mut handle = std::ptr::null_mut();
        check(unsafe {
            ffi::smh_crs_assemble(T::DTYPE, self.vals.len(), self.rows.as_ptr(), self.cols.as_ptr(),
                                  self.vals.as_ptr() as *const c_void, self.ops.as_ptr(), &mut handle)
        });
        DeviceCrs { handle, n_rows: unsafe { ffi::smh_crs_n_rows(handle) } }
    }
}

impl Drop for DeviceCrs {
    fn drop(&mut self) {
        unsafe { ffi::smh_crs_destroy(self.handle) };
    }
}

/// `SparseMatPar<SparseMatCRS<T, u32>>` (src/sparsemat_par.rs:12-35) on several devices of this process: block `b`
/// = rows `[b R, (b+1) R)`, `R = n_rows / n_blocks` (src/sparsemat_par.rs:21), on device `devices[b]`.  Built from the
/// global CRS arrays the sub-matrices concatenate to (local row ids, global column ids, as the reference stores them).
pub struct DevicePar {
    handle: *mut ffi::smh_par,
    n_rows: usize,
}

impl DevicePar {
    pub fn new<T: HipValue>(n_blocks: usize, devices: Option<&[i32]>, n_rows: usize, n_cols: usize, offset_rows: &[u32],
                            columns: &[u32], values: &[T]) -> Self {
        assert_eq!(offset_rows.len(), n_rows + 1);
        if let Some(d) = devices { assert_eq!(d.len(), n_blocks); }
        let mut handle = std::ptr::null_mut();
        check(unsafe {
            ffi::smh_par_create(T::DTYPE, n_blocks, devices.map_or(std::ptr::null(), |d| d.as_ptr()), n_rows, n_cols,
                                offset_rows.as_ptr(), columns.as_ptr(), values.as_ptr() as *const c_void, 1, &mut handle)
        });
        DevicePar { handle, n_rows }
    }

    /// Not in the reference: the same matrix cut into blocks of equal ENTRY counts (for skewed matrices; block lookup, exchanges
    /// and the solver follow the table of row boundaries `split()` returns).
    pub fn new_nnz_balanced<T: HipValue>(n_blocks: usize, devices: Option<&[i32]>, n_rows: usize, n_cols: usize, offset_rows: &[u32],
                                         columns: &[u32], values: &[T]) -> Self {
        assert_eq!(offset_rows.len(), n_rows + 1);
        if let Some(d) = devices { assert_eq!(d.len(), n_blocks); }
        let mut handle = std::ptr::null_mut();
        check(unsafe {
            ffi::smh_par_create_split(T::DTYPE, n_blocks, devices.map_or(std::ptr::null(), |d| d.as_ptr()), n_rows, n_cols,
                                      offset_rows.as_ptr(), columns.as_ptr(), values.as_ptr() as *const c_void, 1,
                                      ffi::SMH_SPLIT_NNZ, &mut handle)
        });
        DevicePar { handle, n_rows }
    }

    /// `SparseMatrix::mvp` through `iter_row` (src/sparsemat_par.rs:86-89): all blocks run concurrently.
    pub fn mvp<T: HipValue + Default>(&self, x: &[T]) -> Vec<T> {
        let mut y = vec![T::default(); self.n_rows];
        check(unsafe {
            ffi::smh_par_spmv(self.handle, x.as_ptr() as *const c_void, x.len(), y.as_mut_ptr() as *mut c_void, ffi::SMH_SPMV_AUTO)
        });
        y
    }

    /// `ConjugateGradient::solve` with `M = SparseMatPar<..>`: halos of `p` travel device to device, the two dot products are
    /// folded across the blocks on the devices (host vectors in and out).
    pub fn cg_solve<T: HipValue>(&self, b: &[T], x: &mut [T], tol: f64, iter_max: usize) -> (usize, f64) {
        let (mut iters, mut rr) = (0usize, 0f64);
        check(unsafe {
            ffi::smh_par_cg_solve(self.handle, b.as_ptr() as *const c_void, b.len(), x.as_mut_ptr() as *mut c_void, x.len(),
                                  tol, iter_max, ffi::SMH_SPMV_AUTO, &mut iters, &mut rr)
        });
        (iters, rr)
    }
}

/// A `DenseVec` distributed over the blocks of a `DevicePar`: block `b` owns entries `[b R, (b+1) R)`; every block keeps a
/// full-length device buffer, valid as far as the last upload / exchange made it.
pub struct DeviceParVec<'a> {
    handle: *mut ffi::smh_par_vec,
    par: &'a DevicePar,
}

impl<'a> DeviceParVec<'a> {
    pub fn from_slice<T: HipValue>(par: &'a DevicePar, host: &[T]) -> Self {
        let mut handle = std::ptr::null_mut();
        check(unsafe { ffi::smh_par_vec_create(par.handle, host.len(), &mut handle) });
        check(unsafe { ffi::smh_par_vec_upload(handle, host.as_ptr() as *const c_void) });
        DeviceParVec { handle, par }
    }
    pub fn zeros(par: &'a DevicePar) -> Self {
        let mut handle = std::ptr::null_mut();
        check(unsafe { ffi::smh_par_vec_create(par.handle, par.n_rows, &mut handle) });
        DeviceParVec { handle, par }
    }
    pub fn to_vec<T: HipValue + Default>(&self) -> Vec<T> {
        let mut out = vec![T::default(); self.par.n_rows];
        check(unsafe { ffi::smh_par_vec_download(self.handle, out.as_mut_ptr() as *mut c_void) });
        out
    }
}

impl<'a> Drop for DeviceParVec<'a> {
    fn drop(&mut self) {
        unsafe { ffi::smh_par_vec_destroy(self.handle) };
    }
}

impl DevicePar {
    /// The reference's intended `mvp_par` (src/sparsemat_par.rs:37-68), device resident: every block multiplies against the
    /// shared `x`, writes its slice of `y` at `b * R`, then ONE exchange inside the library (RCCL all-gather, or only the
    /// entries each block references) makes `y` usable as the next `x`.  Asynchronous; `synchronize` waits.
    pub fn mvp_dev(&self, x: &DeviceParVec, y: &mut DeviceParVec) {
        check(unsafe { ffi::smh_par_spmv_dev(self.handle, x.handle, y.handle, ffi::SMH_SPMV_AUTO, ffi::SMH_EXCHANGE_AUTO) });
    }
    pub fn synchronize(&self) {
        check(unsafe { ffi::smh_par_synchronize(self.handle) });
    }
    /// `ConjugateGradient::solve` on distributed vectors: scalars stay on the devices, folded across blocks in a fixed order.
    pub fn cg_solve_dev(&self, b: &DeviceParVec, x: &mut DeviceParVec, tol: f64, iter_max: usize) -> (usize, f64) {
        let (mut iters, mut rr) = (0usize, 0f64);
        check(unsafe { ffi::smh_par_cg_solve_vec(self.handle, b.handle, x.handle, tol, iter_max, ffi::SMH_SPMV_AUTO, 0, &mut iters, &mut rr) });
        (iters, rr)
    }
}

impl Drop for DevicePar {
    fn drop(&mut self) {
        unsafe { ffi::smh_par_destroy(self.handle) };
    }
}
