"""ctypes binding of libsparsemat_hip.so (the C ABI declared in include/sparsemat_hip.h).

The library is the product; this module only declares its signatures.  There is no
fallback: if the shared object is missing or fails to load, importing a symbol raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPARSEMAT_HIP_LIB: load an A/B build of the same ABI instead (development aid)
LIB_PATH = os.environ.get("SPARSEMAT_HIP_LIB") or os.path.join(_HERE, "libsparsemat_hip.so")

SMH_OK = 0
SMH_ERR_DIM_MISMATCH, SMH_ERR_NOT_SQUARE, SMH_ERR_INDEX_RANGE, SMH_ERR_INVALID = 1, 2, 3, 4
SMH_ERR_HIP, SMH_ERR_OOM, SMH_ERR_NO_DEVICE, SMH_ERR_CAPACITY, SMH_ERR_COMM = 5, 6, 7, 8, 9
SMH_F32, SMH_F64 = 0, 1
EXCHANGE_NONE, EXCHANGE_ALLGATHER, EXCHANGE_WINDOW, EXCHANGE_AUTO = 0, 1, 2, 3
EXCHANGES = {"none": EXCHANGE_NONE, "allgather": EXCHANGE_ALLGATHER, "window": EXCHANGE_WINDOW, "halo": EXCHANGE_WINDOW, "auto": EXCHANGE_AUTO}
EXCHANGE_NAMES = {EXCHANGE_NONE: "none", EXCHANGE_ALLGATHER: "allgather", EXCHANGE_WINDOW: "window", EXCHANGE_AUTO: "auto"}
PAR_BACKEND_AUTO, PAR_BACKEND_PEER, PAR_BACKEND_RCCL = 0, 1, 2
PAR_BACKENDS = {"auto": PAR_BACKEND_AUTO, "peer": PAR_BACKEND_PEER, "rccl": PAR_BACKEND_RCCL}
COMM_ID_BYTES = 128
SPMV_AUTO, SPMV_VECTOR, SPMV_MERGE, SPMV_SEQ, SPMV_STREAM, SPMV_COLBLOCK, SPMV_COLFUSED, SPMV_COLSPLIT, SPMV_TILED = 0, 1, 2, 3, 4, 5, 6, 7, 8
VARIANTS = {"auto": SPMV_AUTO, "vector": SPMV_VECTOR, "merge": SPMV_MERGE, "seq": SPMV_SEQ, "stream": SPMV_STREAM,
            "colblock": SPMV_COLBLOCK, "colfused": SPMV_COLFUSED, "colsplit": SPMV_COLSPLIT, "tiled": SPMV_TILED}

_sz = C.c_size_t
_vp = C.c_void_p
_u32p = C.POINTER(C.c_uint32)
_int = C.c_int

# name -> (restype, argtypes); every symbol include/sparsemat_hip.h declares
SIGNATURES = {
    "smh_abi_version": (_int, []),
    "smh_last_error": (C.c_char_p, []),
    "smh_status_string": (C.c_char_p, [_int]),
    "smh_device_count": (_int, [C.POINTER(_int)]),
    "smh_set_device": (_int, [_int]),
    "smh_device_synchronize": (_int, []),
    "smh_crs_create": (_int, [_int, _sz, _sz, _sz, _vp, _vp, _vp, _int, C.POINTER(_vp)]),
    "smh_crs_create_dev": (_int, [_int, _sz, _sz, _sz, _vp, _vp, _vp, _int, C.POINTER(_vp)]),
    "smh_crs_assemble": (_int, [_int, _sz, _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "smh_crs_assemble_dev": (_int, [_int, _sz, _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "smh_crs_replay": (_int, [_int, _sz, _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "smh_crs_replay_dev": (_int, [_int, _sz, _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "smh_crs_transpose": (_int, [_vp, C.POINTER(_vp)]),
    "smh_crs_column_info": (_int, [_vp, _vp, _vp, _vp]),
    "smh_crs_column_info_dev": (_int, [_vp, _vp, _vp, _vp]),
    "smh_crs_prod": (_int, [_vp, _vp, C.POINTER(_vp)]),
    "smh_crs_is_symmetric": (_int, [_vp, C.POINTER(_int)]),
    "smh_crs_is_sorted": (_int, [_vp, C.POINTER(_int)]),
    "smh_crs_sort_rows": (_int, [_vp]),
    "smh_crs_destroy": (_int, [_vp]),
    "smh_crs_update_values": (_int, [_vp, _vp]),
    "smh_crs_download": (_int, [_vp, _vp, _vp, _vp]),
    "smh_crs_n_rows": (_sz, [_vp]),
    "smh_crs_n_cols": (_sz, [_vp]),
    "smh_crs_nnz": (_sz, [_vp]),
    "smh_crs_orphans": (_sz, [_vp]),
    "smh_crs_dtype": (_int, [_vp]),
    "smh_crs_max_row_len": (_int, [_vp, _u32p]),
    "smh_crs_col_range": (_int, [_vp, _u32p, _u32p]),
    "smh_crs_scale": (_int, [_vp, C.c_double]),
    "smh_crs_resolved_variant": (_int, [_vp, C.POINTER(_int), C.POINTER(_int)]),
    "smh_crs_set_vector_lanes": (_int, [_vp, _int]),
    "smh_crs_set_vector_chunks": (_int, [_vp, _int]),
    "smh_crs_set_ring": (_int, [_vp, _int]),
    "smh_crs_ring_entries": (_int, [_vp, _u32p]),
    "smh_crs_ring_bands": (_int, [_vp, _u32p, _vp]),
    "smh_crs_ring_plan": (_int, [_vp, _u32p, C.POINTER(_sz), C.POINTER(C.c_double), C.POINTER(_int), _vp, _vp]),
    "smh_crs_set_colblock_shift": (_int, [_vp, C.c_uint32]),
    "smh_crs_tiled_layout": (_int, [_vp, _u32p, _u32p, _u32p, _u32p, C.POINTER(_sz)]),
    "smh_crs_tiled_products": (_int, [_vp, C.POINTER(_sz)]),
    "smh_crs_prepare_stats": (_int, [_vp, _int, C.POINTER(C.c_double), C.POINTER(_sz)]),
    "smh_crs_tiled_array": (_int, [_vp, _int, _vp, _sz, C.POINTER(_sz)]),
    "smh_crs_colblock": (_int, [_vp, _u32p, C.POINTER(_sz), C.POINTER(_int), C.POINTER(C.c_double), _vp, _vp, _vp]),
    "smh_crs_colfused": (_int, [_vp, C.POINTER(_int), _u32p, C.POINTER(_sz), _u32p, C.POINTER(_sz), _vp, _vp, _vp, _vp, _vp]),
    "smh_crs_colsplit": (_int, [_vp, C.POINTER(_int), _u32p, C.POINTER(_sz), _vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "smh_crs_spmv": (_int, [_vp, _vp, _sz, _vp, _int]),
    "smh_crs_spmv_dev": (_int, [_vp, _vp, _sz, _vp, _int, _vp]),
    "smh_crs_prepare": (_int, [_vp, _int]),
    "smh_crs_merge_tiles": (_sz, [_vp]),
    "smh_crs_merge_tile_items": (_sz, [_vp]),
    "smh_crs_merge_table": (_int, [_vp, _vp, _vp]),
    "smh_vec_create": (_int, [_int, _sz, C.POINTER(_vp)]),
    "smh_vec_from_host": (_int, [_int, _sz, _vp, C.POINTER(_vp)]),
    "smh_vec_wrap_dev": (_int, [_int, _sz, _vp, C.POINTER(_vp)]),
    "smh_vec_destroy": (_int, [_vp]),
    "smh_vec_upload": (_int, [_vp, _vp]),
    "smh_vec_download": (_int, [_vp, _vp]),
    "smh_vec_dim": (_sz, [_vp]),
    "smh_vec_dtype": (_int, [_vp]),
    "smh_vec_data": (_vp, [_vp]),
    "smh_vec_copy": (_int, [_vp, _vp]),
    "smh_vec_add": (_int, [_vp, _vp]),
    "smh_vec_sub": (_int, [_vp, _vp]),
    "smh_vec_scale": (_int, [_vp, C.c_double]),
    "smh_vec_axpy": (_int, [_vp, C.c_double, _vp]),
    "smh_vec_xpby": (_int, [_vp, C.c_double, _vp]),
    "smh_vec_dot": (_int, [_vp, _vp, C.POINTER(C.c_double)]),
    "smh_vec_norm_squared": (_int, [_vp, C.POINTER(C.c_double)]),
    "smh_vec_norm": (_int, [_vp, C.POINTER(C.c_double)]),
    "smh_blas_dot_dev": (_int, [_int, _vp, _vp, _sz, _vp, _vp, _vp]),
    "smh_blas_dot_scratch_bytes": (_sz, []),
    "smh_blas_axpy_dev": (_int, [_int, _vp, _vp, _vp, _sz, _vp]),
    "smh_blas_xpby_dev": (_int, [_int, _vp, _vp, _vp, _sz, _vp]),
    "smh_crs_spmv_vec": (_int, [_vp, _vp, _vp, _int]),
    "smh_crs_inner_prod": (_int, [_vp, _vp, _sz, _vp, _sz, _int, C.POINTER(C.c_double)]),
    "smh_crs_inner_prod_vec": (_int, [_vp, _vp, _vp, _int, C.POINTER(C.c_double)]),
    "smh_cg_solve": (_int, [_vp, _vp, _sz, _vp, _sz, C.c_double, _sz, _int, C.POINTER(_sz),
                            C.POINTER(C.c_double)]),
    "smh_cg_solve_vec": (_int, [_vp, _vp, _vp, C.c_double, _sz, _int, _sz, C.POINTER(_sz),
                                C.POINTER(C.c_double)]),
    "smh_pcg_jacobi_solve": (_int, [_vp, _vp, _sz, _vp, _sz, C.c_double, _sz, _int, C.POINTER(_sz),
                                    C.POINTER(C.c_double)]),
    "smh_comm_unique_id": (_int, [_vp]),
    "smh_comm_create": (_int, [_vp, _int, _int, C.POINTER(_vp)]),
    "smh_comm_destroy": (_int, [_vp]),
    "smh_comm_size": (_int, [_vp]),
    "smh_comm_rank": (_int, [_vp]),
    "smh_comm_ranks_seen": (_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "smh_rccl_version": (_int, [C.POINTER(C.c_int)]),
    "smh_comm_barrier": (_int, [_vp]),
    "smh_comm_max_f64": (_int, [_vp, C.POINTER(C.c_double)]),
    "smh_par_create": (_int, [_int, _sz, _vp, _sz, _sz, _vp, _vp, _vp, _int, C.POINTER(_vp)]),
    "smh_par_create_split": (_int, [_int, _sz, _vp, _sz, _sz, _vp, _vp, _vp, _int, _int, C.POINTER(_vp)]),
    "smh_par_adopt_split": (_int, [_sz, C.POINTER(_vp), _sz, _vp, C.POINTER(_vp)]),
    "smh_par_create_rank_split": (_int, [_vp, _sz, _vp, _sz, C.POINTER(_vp)]),
    "smh_par_split": (_int, [_vp, _vp]),
    "smh_par_plan_split": (_int, [_sz, _sz, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, C.POINTER(_int), C.POINTER(_sz)]),
    "smh_par_adopt": (_int, [_sz, C.POINTER(_vp), _sz, C.POINTER(_vp)]),
    "smh_par_create_rank": (_int, [_vp, _sz, _vp, C.POINTER(_vp)]),
    "smh_par_n_local_blocks": (_sz, [_vp]),
    "smh_par_set_backend": (_int, [_vp, _int]),
    "smh_par_backend": (_int, [_vp]),
    "smh_par_set_overlap": (_int, [_vp, _int]),
    "smh_par_set_threads": (_int, [_vp, _int]),
    "smh_par_interior": (_int, [_vp, _sz, _int, C.POINTER(_sz), C.POINTER(_sz)]),
    "smh_par_exchange_mode": (_int, [_vp, _int, C.POINTER(_int), C.POINTER(_sz)]),
    "smh_par_plan": (_int, [_sz, _sz, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, C.POINTER(_int), C.POINTER(_sz)]),
    "smh_par_vec_create": (_int, [_vp, _sz, C.POINTER(_vp)]),
    "smh_par_vec_destroy": (_int, [_vp]),
    "smh_par_vec_dim": (_sz, [_vp]),
    "smh_par_vec_upload": (_int, [_vp, _vp]),
    "smh_par_vec_download": (_int, [_vp, _vp]),
    "smh_par_vec_download_block": (_int, [_vp, _sz, _vp]),
    "smh_par_vec_ptr": (_int, [_vp, _sz, C.POINTER(_vp)]),
    "smh_par_spmv_dev": (_int, [_vp, _vp, _vp, _int, _int]),
    "smh_par_exchange": (_int, [_vp, _vp, _int]),
    "smh_par_synchronize": (_int, [_vp]),
    "smh_par_cg_solve_vec": (_int, [_vp, _vp, _vp, C.c_double, _sz, _int, _sz, C.POINTER(_sz), C.POINTER(C.c_double)]),
    "smh_par_destroy": (_int, [_vp]),
    "smh_par_n_blocks": (_sz, [_vp]),
    "smh_par_n_rows": (_sz, [_vp]),
    "smh_par_n_cols": (_sz, [_vp]),
    "smh_par_nnz": (_sz, [_vp]),
    "smh_par_rows_per_block": (_sz, [_vp]),
    "smh_par_block": (_int, [_vp, _sz, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_int)]),
    "smh_par_block_stream": (_int, [_vp, _sz, C.POINTER(_vp)]),
    "smh_par_get_block_and_row_id": (_int, [_vp, _sz, C.POINTER(_sz), C.POINTER(_sz)]),
    "smh_par_scale": (_int, [_vp, C.c_double]),
    "smh_par_spmv": (_int, [_vp, _vp, _sz, _vp, _int]),
    "smh_par_cg_solve": (_int, [_vp, _vp, _sz, _vp, _sz, C.c_double, _sz, _int, C.POINTER(_sz),
                                C.POINTER(C.c_double)]),
    "smh_synth_x": (_int, [_int, C.c_uint64, _sz, _sz, _vp, _vp]),
    "smh_synth_fixed": (_int, [_int, C.c_uint64, _int, _sz, C.c_uint32, _sz, _sz, _vp, _vp, _vp, _vp]),
    "smh_synth_powerlaw_cdf": (_int, [C.c_uint32, C.c_double, _vp]),
    "smh_synth_powerlaw_lengths": (_int, [C.c_uint64, _sz, _sz, C.c_uint32, _vp, _vp]),
    "smh_synth_fill": (_int, [_int, C.c_uint64, _sz, _sz, _sz, _vp, _vp, _vp, _vp]),
    "smh_synth_laplace3d": (_int, [_int, _sz, _sz, _sz, _sz, _sz, _vp, _vp, _vp, C.POINTER(_sz), _vp]),
    "smh_crs_set_stream_xs": (_int, [_vp, _int]),
    "smh_crs_set_stream_direct": (_int, [_vp, _int]),
    "smh_crs_stream_direct": (_int, [_vp, C.POINTER(_int)]),
    "smh_crs_stream_layout": (_int, [_vp, C.POINTER(_int), C.POINTER(_int), C.POINTER(_int), C.POINTER(_int)]),
    "smh_crs_stream_value_dict": (_int, [_vp, C.POINTER(C.c_int), _vp]),
    "smh_crs_set_stream_value_dict": (_int, [_vp, _int]),
    "smh_last_transpose_route": (_int, []),
    "smh_pool_trim": (_int, []),
    "smh_pool_stats": (_int, [C.POINTER(_sz), C.POINTER(_sz)]),
    "smh_dev_alloc": (_int, [_sz, C.POINTER(_vp)]),
    "smh_dev_free": (_int, [_vp]),
    "smh_dev_upload": (_int, [_vp, _vp, _sz]),
    "smh_dev_download": (_int, [_vp, _vp, _sz]),
    "smh_dev_memset": (_int, [_vp, _int, _sz, _vp]),
    "smh_stream_create": (_int, [C.POINTER(_vp)]),
    "smh_stream_destroy": (_int, [_vp]),
    "smh_stream_synchronize": (_int, [_vp]),
    "smh_event_create": (_int, [C.POINTER(_vp)]),
    "smh_event_destroy": (_int, [_vp]),
    "smh_event_record": (_int, [_vp, _vp]),
    "smh_event_elapsed_ms": (_int, [_vp, _vp, C.POINTER(C.c_float)]),
}

_LIB = None


class SparseMatPanic(RuntimeError):
    """A non-zero status of the C ABI; the message is the reference's panic text where one exists."""

    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


def lib():
    """Load libsparsemat_hip.so.  Fails loudly: there is no CPU or pure-Python fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -m sparsemat_amd.build` (hipcc, gfx950). "
                "sparsemat_amd has no CPU fallback." % LIB_PATH)
        import sys
        if "torch" in sys.modules:
            # PyTorch wheels bundle their own HIP runtime; if torch shares the process it must initialise before
            # the system runtime this library links is loaded (otherwise torch later finds no device)
            try:
                if sys.modules["torch"].cuda.is_available():
                    sys.modules["torch"].cuda.init()
            except Exception:
                pass
        # one process per GPU over RCCL shares device memory between processes through dmabuf handles; the host driver of
        # this platform supports only that mode, and the HSA runtime reads the switch when it initialises (first HIP call)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(status):
    if status != SMH_OK:
        L = lib()
        msg = L.smh_last_error().decode(errors="replace") or L.smh_status_string(status).decode()
        raise SparseMatPanic(status, msg)


def dtype_code(np_dtype):
    import numpy as np
    dt = np.dtype(np_dtype)
    if dt == np.float32:
        return SMH_F32
    if dt == np.float64:
        return SMH_F64
    raise TypeError("the HIP path handles f32/f64 values only (got %s)" % dt)


def np_dtype(code):
    import numpy as np
    return np.float64 if code == SMH_F64 else np.float32
