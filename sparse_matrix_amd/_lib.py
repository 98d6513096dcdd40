"""ctypes binding of libspam_hip.so (the C ABI in include/spam_hip.h).

There is NO CPU fallback: if the HIP library is missing or a call fails, this raises.  The oracle
(``oracle/``) is test infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libspam_hip.so")

SPAM_OK = 0
SPAM_F32, SPAM_F64, SPAM_I64 = 0, 1, 2
ERR_NAMES = {1: "SPAM_E_INVALID", 2: "SPAM_E_INDEX_RANGE", 3: "SPAM_E_OOM", 4: "SPAM_E_HIP",
             5: "SPAM_E_OVERFLOW", 6: "SPAM_E_UNSUPPORTED", 7: "SPAM_E_DTYPE", 8: "SPAM_E_COMM"}
DTYPE_SUFFIX = {0: "f32", 1: "f64", 2: "i64"}


class SpamError(RuntimeError):
    """Non-zero status from libspam_hip.so — where the Rust reference would panic."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class SpgemmStats(C.Structure):
    _fields_ = [("rows", C.c_uint64), ("inner", C.c_uint64), ("cols", C.c_uint64),
                ("nnz_a", C.c_uint64), ("nnz_b", C.c_uint64), ("nnz_c", C.c_uint64),
                ("flops_f", C.c_uint64),
                ("rows_sym_bin", C.c_uint64 * 8), ("rows_num_bin", C.c_uint64 * 8),
                ("heavy_items", C.c_uint64),
                ("heavy_dense_groups", C.c_uint64), ("heavy_hash_groups", C.c_uint64),
                ("heavy_dense_nnz", C.c_uint64), ("heavy_hash_nnz", C.c_uint64),
                ("ms_flops", C.c_float), ("ms_symbolic", C.c_float), ("ms_alloc", C.c_float),
                ("ms_numeric", C.c_float), ("ms_total", C.c_float)]

    def asdict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else v
        return d


# every symbol include/spam_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
_U64 = C.c_uint64
_PP = C.POINTER(C.c_void_p)
SYMBOLS = {
    "spam_hip_init": (C.c_int, [C.c_int]),
    "spam_hip_set_stream": (C.c_int, [_P]),
    "spam_hip_synchronize": (C.c_int, []),
    "spam_hip_trim": (C.c_int, []),
    "spam_hip_copy_gbps": (C.c_int, [_U64, C.c_int, C.POINTER(C.c_double)]),
    "spam_last_error": (C.c_char_p, []),
    "spam_version": (C.c_int, []),
    "spam_csr_upload_f64": (C.c_int, [_U64, _U64, _U64, _P, _P, _P, C.c_int, _PP]),
    "spam_csr_upload_f32": (C.c_int, [_U64, _U64, _U64, _P, _P, _P, C.c_int, _PP]),
    "spam_csr_upload_i64": (C.c_int, [_U64, _U64, _U64, _P, _P, _P, C.c_int, _PP]),
    "spam_csr_from_device": (C.c_int, [C.c_int, _U64, _U64, _U64, _P, _P, _P, C.c_int, _PP]),
    "spam_csr_from_sorted_coo": (C.c_int, [C.c_int, _U64, _U64, _U64, _P, _P, _P, C.c_int, _PP]),
    "spam_csr_row_block": (C.c_int, [_P, _U64, _U64, _PP]),
    "spam_csr_dims": (C.c_int, [_P, C.POINTER(_U64), C.POINTER(_U64), C.POINTER(_U64), C.POINTER(C.c_int),
                                C.POINTER(C.c_int)]),
    "spam_csr_device_ptrs": (C.c_int, [_P, _PP, _PP, _PP]),
    "spam_csr_download_f64": (C.c_int, [_P, _P, _P, _P]),
    "spam_csr_download_f32": (C.c_int, [_P, _P, _P, _P]),
    "spam_csr_download_i64": (C.c_int, [_P, _P, _P, _P]),
    "spam_csr_sort_rows": (C.c_int, [_P]),
    "spam_csr_invariants": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "spam_csr_add": (C.c_int, [_P, _P, C.c_int, _PP]),
    "spam_csr_free": (None, [_P]),
    "spam_spmv_f64": (C.c_int, [_P, _P, _P]),
    "spam_spmv_f32": (C.c_int, [_P, _P, _P]),
    "spam_spmv_i64": (C.c_int, [_P, _P, _P]),
    "spam_spmv_dev": (C.c_int, [_P, _P, _P]),
    "spam_spmv_prepare": (C.c_int, [_P, _P]),
    "spam_csr_values_changed": (C.c_int, [_P]),
    "spam_spmv_plan_info": (C.c_int, [_P, C.POINTER(_U64), C.POINTER(C.c_double), C.POINTER(_U64)]),
    "spam_spmv_sliced_info": (C.c_int, [_P, C.POINTER(C.c_int)] + [C.POINTER(_U64)] * 4 + [C.POINTER(C.c_double)] * 4),
    "spam_spmv_rg_info": (C.c_int, [_P, C.POINTER(C.c_int)] + [C.POINTER(_U64)] * 5),
    "spam_spgemm_symbolic": (C.c_int, [_P, _P, _PP, C.POINTER(_U64)]),
    "spam_spgemm_numeric_f64": (C.c_int, [_P, C.c_int, _P, _P, _P]),
    "spam_spgemm_numeric_f32": (C.c_int, [_P, C.c_int, _P, _P, _P]),
    "spam_spgemm_numeric_i64": (C.c_int, [_P, C.c_int, _P, _P, _P]),
    "spam_spgemm_numeric_dev": (C.c_int, [_P, C.c_int, _PP]),
    "spam_spgemm_plan_stats": (C.c_int, [_P, C.POINTER(SpgemmStats)]),
    "spam_spgemm_plan_free": (None, [_P]),
    "spam_spgemm_dev": (C.c_int, [_P, _P, C.c_int, _PP, C.POINTER(SpgemmStats)]),
    "spam_spgemm_rank_step": (C.c_int, [_P, _P, _U64, _U64, _U64, C.c_int, _PP, C.POINTER(_U64), C.POINTER(_U64),
                                        C.POINTER(SpgemmStats)]),
    "spam_rows_to_parts": (C.c_int, [_P, _P, _U64, _P, _P]),
    "spam_rows_to_parts_weighted": (C.c_int, [_P, _P, _U64, _U64, _P]),
    "spam_row_overhead_auto": (_U64, [_P]),
    "spam_row_flops_dev": (C.c_int, [_P, _P, _U64, _U64, _P]),
    "spam_rows_to_parts_from_flops_dev": (C.c_int, [_P, _U64, _U64, _U64, _P]),
}

# every symbol include/spam_hip_rccl.h declares (libspam_hip_rccl.so: links RCCL; loaded only by multi-GPU hosts)
SYMBOLS_RCCL = {
    "spam_rccl_unique_id": (C.c_int, [_P]),
    "spam_rccl_comm_create": (C.c_int, [_P, C.c_int, C.c_int, _PP]),
    "spam_rccl_comm_adopt": (C.c_int, [_P, C.c_int, C.c_int, _PP]),
    "spam_rccl_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "spam_rccl_last_exchange": (C.c_int, [_P, C.POINTER(_U64), C.POINTER(_U64)]),
    "spam_rccl_comm_free": (None, [_P]),
    "spam_csr_allgather_rccl": (C.c_int, [_P, _P, _P, _PP]),
    "spam_vector_allgather_rccl": (C.c_int, [_P, C.c_int, _P, _P, _P]),
    "spam_spgemm_rank_step_rccl": (C.c_int, [_P, _P, _P, _P, _U64, C.c_int, _PP, C.POINTER(_U64), C.POINTER(_U64),
                                             C.POINTER(SpgemmStats)]),
}
RCCL_ID_BYTES = 128
ROW_OVERHEAD_AUTO = 0xFFFFFFFFFFFFFFFF  # SPAM_ROW_OVERHEAD_AUTO: per-row partition term from the width of B
LIB_RCCL_PATH = os.path.join(_HERE, "libspam_hip_rccl.so")

_lib = None
_lib_rccl = None


def load_rccl():
    """Load libspam_hip_rccl.so (RCCL exchange behind the C ABI) and bind every declared symbol."""
    global _lib_rccl
    if _lib_rccl is None:
        load()  # libspam_hip.so first: the sibling resolves its symbols against the copy already mapped
        if not os.path.exists(LIB_RCCL_PATH):
            raise ImportError(f"{LIB_RCCL_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(LIB_RCCL_PATH)
        for name, (res, args) in SYMBOLS_RCCL.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib_rccl = lib
    return _lib_rccl


def load():
    """Load libspam_hip.so and bind every declared symbol.  Raises if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  sparse_matrix_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(status: int):
    if status != SPAM_OK:
        raise SpamError(status, load().spam_last_error().decode())


_initialised = False


def ensure_device(device: int | None = None):
    """spam_hip_init once per process (device defaults to LOCAL_RANK or 0)."""
    global _initialised
    if not _initialised or device is not None:
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        check(load().spam_hip_init(device))
        _initialised = True
