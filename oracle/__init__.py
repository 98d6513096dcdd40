"""CPU oracle for the spam_csr hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product (``sparse_matrix_amd``) never does, and fails loudly without its HIP library.

``liboracle.so`` is the C++ restatement in ``spam_oracle.cpp`` (follows /root/reference's Rust line by
line; see its header for the parity-pinning statement).  ``dok.py`` restates the reference's DOK
test-oracle (triple-loop product, zero-dropping maps, Higham bound) in numpy for small cases.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ORC_OK, ORC_E_OVERFLOW, ORC_E_OOB, ORC_E_ASSERT, ORC_E_BADKEY = range(5)


class OracleError(RuntimeError):
    """Mirrors a panic of the Rust reference (overflow unwrap, slice OOB, assert_eq!)."""

    def __init__(self, code):
        names = {1: "checked_add overflow", 2: "index out of bounds (inner dimension mismatch)",
                 3: "assert_eq!(curr, ...) failed", 4: "key == u32::MAX"}
        super().__init__(f"oracle panic: {names.get(code, code)}")
        self.code = code


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "spam_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_table_size_for.restype = C.c_uint64
        _LIB.orc_table_size_for.argtypes = [C.c_uint64]
        _LIB.orc_hash_u32.restype = C.c_uint64
        _LIB.orc_hash_u32.argtypes = [C.c_uint32]
        _LIB.orc_hashset_run.restype = C.c_uint64
        _LIB.orc_hashmap_run_f64.restype = C.c_uint64
        _LIB.orc_hardware_threads.restype = C.c_uint
    return _LIB


_DT = {np.dtype(np.float64): "f64", np.dtype(np.float32): "f32", np.dtype(np.int8): "i8",
       np.dtype(np.int64): "i64"}


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def hardware_threads() -> int:
    return int(lib().orc_hardware_threads())


def table_size_for(capacity: int) -> int:
    return int(lib().orc_table_size_for(capacity))


def hash_u32(key: int) -> int:
    return int(lib().orc_hash_u32(key))


def checked_inclusive_scan(v) -> np.ndarray:
    v = _u64(v)
    out = np.empty(v.size + 1, dtype=np.uint64)
    e = lib().orc_checked_inclusive_scan(_p(v), C.c_uint64(v.size), _p(out))
    if e:
        raise OracleError(e)
    return out


def hashset_run(keys, shrink_hint: int):
    """One symbolic row: returns (len, upper_bound, slots[:upper_bound])."""
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    cap = max(64, 8 * table_size_for(max(int(keys.size), 1)))
    slots = np.empty(cap, dtype=np.uint32)
    ub = C.c_uint64(0)
    n = lib().orc_hashset_run(_p(keys), C.c_uint64(keys.size), C.c_uint64(shrink_hint), _p(slots),
                              C.c_uint64(cap), C.byref(ub))
    return int(n), int(ub.value), slots[: ub.value].copy()


def hashmap_run_f64(keys, vals, with_capacity: int, row_nz: int):
    """One numeric row: returns drained (keys, vals) in slot order."""
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    ko = np.empty(keys.size, dtype=np.uint32)
    vo = np.empty(keys.size, dtype=np.float64)
    m = lib().orc_hashmap_run_f64(_p(keys), _p(vals), C.c_uint64(keys.size), C.c_uint64(with_capacity),
                                  C.c_uint64(row_nz), _p(ko), _p(vo))
    return ko[:m].copy(), vo[:m].copy()


def rows_to_threads(a_off, a_idx, b_off, tnum: int):
    """mul_hash.rs:38-64 -> (row_nz[rows], rows_offset[tnum+1])."""
    a_off, a_idx, b_off = _u64(a_off), _u64(a_idx), _u64(b_off)
    rows = a_off.size - 1
    row_nz = np.empty(rows, dtype=np.uint64)
    rows_offset = np.empty(tnum + 1, dtype=np.uint64)
    e = lib().orc_rows_to_threads(C.c_uint64(rows), _p(a_off), _p(a_idx), C.c_uint64(b_off.size - 1), _p(b_off),
                                  C.c_uint64(tnum), _p(row_nz), _p(rows_offset))
    if e:
        raise OracleError(e)
    return row_nz, rows_offset


def mul_hash(a, b, sort_rows: bool = False, tnum: int | None = None):
    """CsrMatrix::mul_hash::<B1, B2>  (spam_csr/src/mul_hash.rs:13-36).

    a, b: (offsets, indices, vals) in the reference's host layout.  Returns (offsets u64[rows+1],
    indices u64[nnz], vals[nnz]).  sort_rows is the const generic B2.  tnum stands for num_cpus::get().
    """
    a_off, a_idx, a_val = _u64(a[0]), _u64(a[1]), np.ascontiguousarray(a[2])
    b_off, b_idx, b_val = _u64(b[0]), _u64(b[1]), np.ascontiguousarray(b[2])
    if a_val.dtype != b_val.dtype or a_val.dtype not in _DT:
        raise TypeError(f"unsupported value dtypes {a_val.dtype}/{b_val.dtype}")
    sfx = _DT[a_val.dtype]
    if tnum is None:
        tnum = hardware_threads()
    rows, brows = a_off.size - 1, b_off.size - 1
    row_nz, rows_offset = rows_to_threads(a_off, a_idx, b_off, tnum)
    L = lib()
    e = L.orc_mul_hash_symbolic(C.c_uint64(rows), _p(a_off), _p(a_idx), C.c_uint64(brows), _p(b_off), _p(b_idx),
                                _p(row_nz), _p(rows_offset), C.c_uint64(tnum))
    if e:
        raise OracleError(e)
    nnz = int(row_nz.sum())
    offsets = np.empty(rows + 1, dtype=np.uint64)
    indices = np.empty(nnz, dtype=np.uint64)
    vals = np.empty(nnz, dtype=a_val.dtype)
    fn = getattr(L, f"orc_mul_hash_numeric_{sfx}")
    e = fn(C.c_uint64(rows), _p(a_off), _p(a_idx), _p(a_val), C.c_uint64(brows), _p(b_off), _p(b_idx), _p(b_val),
           _p(row_nz), _p(rows_offset), C.c_uint64(tnum), C.c_int(int(sort_rows)), _p(offsets), _p(indices),
           _p(vals))
    if e:
        raise OracleError(e)
    return offsets, indices, vals


def apply_elementwise(a, b, subtract: bool = False, is_sorted: bool = True):
    """impl Add / Sub for CsrMatrix (spam_csr/src/lib.rs:83-149, 276-290).  Returns (offsets, indices, vals)."""
    a_off, a_idx, a_val = _u64(a[0]), _u64(a[1]), np.ascontiguousarray(a[2])
    b_off, b_idx, b_val = _u64(b[0]), _u64(b[1]), np.ascontiguousarray(b[2])
    if a_off.size != b_off.size:
        raise OracleError(ORC_E_ASSERT)  # assert_eq!((rows, cols), ...) (lib.rs:87-91)
    sfx = _DT[a_val.dtype]
    rows = a_off.size - 1
    cap = a_idx.size + b_idx.size
    offsets = np.empty(rows + 1, dtype=np.uint64)
    indices = np.empty(cap, dtype=np.uint64)
    vals = np.empty(cap, dtype=a_val.dtype)
    fn = getattr(lib(), f"orc_apply_elementwise_{sfx}")
    fn.restype = C.c_uint64
    n = fn(C.c_uint64(rows), _p(a_off), _p(a_idx), _p(a_val), _p(b_off), _p(b_idx), _p(b_val), C.c_int(int(subtract)),
           C.c_int(int(is_sorted)), _p(offsets), _p(indices), _p(vals))
    return offsets, indices[:n].copy(), vals[:n].copy()


def spmv(a, x, nthreads: int = 1) -> np.ndarray:
    """y = A x with the summation order mul_hash gives for an n x 1 right-hand side."""
    a_off, a_idx, a_val = _u64(a[0]), _u64(a[1]), np.ascontiguousarray(a[2])
    x = np.ascontiguousarray(x, dtype=a_val.dtype)
    sfx = _DT[a_val.dtype]
    rows = a_off.size - 1
    if a_idx.size and int(a_idx.max()) >= x.size:
        raise OracleError(ORC_E_OOB)
    y = np.empty(rows, dtype=a_val.dtype)
    getattr(lib(), f"orc_spmv_{sfx}")(C.c_uint64(rows), _p(a_off), _p(a_idx), _p(a_val), _p(x), _p(y),
                                      C.c_uint(nthreads))
    return y
