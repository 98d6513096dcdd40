"""Host-side mirror of the reference's ``spam_csr::CsrMatrix<T, IS_SORTED>`` over the HIP back end.

Same names, argument meaning and error behaviour as the Rust type (spam_csr/src/lib.rs:25-32,151-197,
292-358) and the ``Matrix`` trait (spam_matrix/src/lib.rs:15-27) for everything on the SpMV / SpGEMM path;
``mul_hash`` (spam_csr/src/mul_hash.rs:13-36) and the new ``spmv`` run on the GPU through the C ABI in
include/spam_hip.h.  ``+`` / ``-`` (apply_elementwise, lib.rs:83-149) also run on the device; element get/set and transpose stay with the
reference's CPU code and are not provided here.

Element types: f32, f64 and i64 (= Wrapping<i64>) are device types.  i8 / i16 / i32 (Wrapping<i8> is what the reference's
own mul_hash test uses, tests.rs:358) run as i64 and are truncated back, which is exact; complex64 / complex128 run as
four real products (two real sums for ``+`` / ``-``) over one sparsity pattern.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import DTYPE_SUFFIX, SPAM_F32, SPAM_F64, SPAM_I64, SpamError, SpgemmStats, check

# int64 = Wrapping<i64> (the reference's integer type: tests.rs:356-371 uses Wrapping<i8>, the bench parses i64)
_NP2DT = {np.dtype(np.float32): SPAM_F32, np.dtype(np.float64): SPAM_F64, np.dtype(np.int64): SPAM_I64}
_DT2NP = {SPAM_F32: np.float32, SPAM_F64: np.float64, SPAM_I64: np.int64}
# The other element types the reference instantiates CsrMatrix with reach the three device types from the host side:
#   * Wrapping<i8> (tests.rs:358), i16, i32: Z/2^64 -> Z/2^k is a ring homomorphism, so the values are widened to i64,
#     multiplied and added with 64-bit wrapping on the device and truncated back — bit-exact, not an approximation;
#   * Complex<f32/f64> (the fuzz targets and MatrixType::Complex, lib.rs:428-430): products and sums are bilinear, so
#     C = (Ar Br - Ai Bi) + i (Ar Bi + Ai Br) from four real products that share one sparsity pattern (mul_hash's
#     pattern depends on the indices only and keeps computed zeros).  Same result as complex arithmetic up to rounding.
_WIDEN = {np.dtype(np.int8): np.int64, np.dtype(np.int16): np.int64, np.dtype(np.int32): np.int64}
_COMPLEX_PART = {np.dtype(np.complex64): np.float32, np.dtype(np.complex128): np.float64}


def _typed(name: str, dt: int):
    """The entry point of a value type: spam_<name>_{f32,f64,i64}."""
    return getattr(_lib.load(), f"spam_{name}_{DTYPE_SUFFIX[dt]}")


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


class DeviceCsr:
    """An uploaded / device-resident matrix (opaque ``spam_csr*``)."""

    def __init__(self, handle, keepalive=None):
        self._h = C.c_void_p(handle) if not isinstance(handle, C.c_void_p) else handle
        self._keep = keepalive  # parent handles / torch tensors the view borrows from

    # -- construction -------------------------------------------------------------------------
    @classmethod
    def upload(cls, rows, cols, offsets, indices, vals, is_sorted):
        _lib.ensure_device()
        L = _lib.load()
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        indices = np.ascontiguousarray(indices, dtype=np.uint64)
        vals = np.ascontiguousarray(vals)
        if vals.dtype not in _NP2DT:
            raise TypeError(f"device handles hold f32 / f64 / int64 values, not {vals.dtype} (CsrMatrix widens i8/i16/i32 and splits complex)")
        if offsets.size != rows + 1:
            raise SpamError(1, "len(offsets) != rows + 1 (invariant 2)")
        if indices.size != vals.size:
            raise SpamError(1, "len(indices) != len(vals) (invariant 1)")
        h = C.c_void_p()
        fn = _typed("csr_upload", _NP2DT[vals.dtype])
        check(fn(rows, cols, vals.size, _vp(offsets), _vp(indices), _vp(vals), int(bool(is_sorted)), C.byref(h)))
        return cls(h)

    @classmethod
    def from_device(cls, dtype, rows, cols, nnz, row_ptr_ptr, col_idx_ptr, vals_ptr, is_sorted, keepalive=None):
        """Wrap device arrays (u64 row_ptr, u32 col_idx, T vals) without copying."""
        _lib.ensure_device()
        h = C.c_void_p()
        check(_lib.load().spam_csr_from_device(dtype, rows, cols, nnz, C.c_void_p(row_ptr_ptr), C.c_void_p(col_idx_ptr),
                                               C.c_void_p(vals_ptr), int(bool(is_sorted)), C.byref(h)))
        return cls(h, keepalive)

    @classmethod
    def from_torch(cls, rows, cols, row_ptr, col_idx, vals, is_sorted):
        """torch CUDA tensors: row_ptr int64[rows+1], col_idx int32[nnz], vals f32/f64[nnz]."""
        import torch

        assert row_ptr.dtype == torch.int64 and col_idx.dtype == torch.int32
        dt = {torch.float64: SPAM_F64, torch.float32: SPAM_F32, torch.int64: SPAM_I64}[vals.dtype]
        return cls.from_device(dt, rows, cols, int(col_idx.numel()), row_ptr.data_ptr(), col_idx.data_ptr(),
                               vals.data_ptr(), is_sorted, keepalive=(row_ptr, col_idx, vals))

    @classmethod
    def from_sorted_coo_torch(cls, rows, cols, row_idx, col_idx, vals, is_sorted=True):
        """(row, col)-sorted triples as torch CUDA tensors (int32 row_idx / col_idx): row_ptr is built on the device."""
        import torch

        assert row_idx.dtype == torch.int32 and col_idx.dtype == torch.int32
        dt = {torch.float64: SPAM_F64, torch.float32: SPAM_F32, torch.int64: SPAM_I64}[vals.dtype]
        _lib.ensure_device()
        h = C.c_void_p()
        check(_lib.load().spam_csr_from_sorted_coo(dt, rows, cols, int(col_idx.numel()), C.c_void_p(row_idx.data_ptr()),
                                                   C.c_void_p(col_idx.data_ptr()), C.c_void_p(vals.data_ptr()),
                                                   int(bool(is_sorted)), C.byref(h)))
        return cls(h, keepalive=(row_idx, col_idx, vals))

    def row_block(self, lo: int, hi: int) -> "DeviceCsr":
        h = C.c_void_p()
        check(_lib.load().spam_csr_row_block(self._h, lo, hi, C.byref(h)))
        return DeviceCsr(h, keepalive=self)

    # -- queries ------------------------------------------------------------------------------
    def dims(self):
        r, c, n = C.c_uint64(), C.c_uint64(), C.c_uint64()
        dt, srt = C.c_int(), C.c_int()
        check(_lib.load().spam_csr_dims(self._h, C.byref(r), C.byref(c), C.byref(n), C.byref(dt), C.byref(srt)))
        return r.value, c.value, n.value, dt.value, bool(srt.value)

    def device_ptrs(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(_lib.load().spam_csr_device_ptrs(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def add(self, rhs: "DeviceCsr", subtract: bool = False) -> "DeviceCsr":
        """C = self + rhs / self - rhs on the device (sorted union rows, zeros kept)."""
        h = C.c_void_p()
        check(_lib.load().spam_csr_add(self._h, rhs._h, int(bool(subtract)), C.byref(h)))
        return DeviceCsr(h)

    def sort_rows(self):
        """In-place per-row sort by column (the handle becomes IS_SORTED = true)."""
        check(_lib.load().spam_csr_sort_rows(self._h))

    def invariants(self) -> bool:
        ok = C.c_int()
        check(_lib.load().spam_csr_invariants(self._h, C.byref(ok)))
        return bool(ok.value)

    def download(self):
        rows, cols, nnz, dt, _ = self.dims()
        offsets = np.empty(rows + 1, dtype=np.uint64)
        indices = np.empty(nnz, dtype=np.uint64)
        vals = np.empty(nnz, dtype=_DT2NP[dt])
        check(_typed("csr_download", dt)(self._h, _vp(offsets), _vp(indices), _vp(vals)))
        return offsets, indices, vals

    def download_row_ptr(self) -> np.ndarray:
        rows, _, _, dt, _ = self.dims()
        offsets = np.empty(rows + 1, dtype=np.uint64)
        check(_typed("csr_download", dt)(self._h, _vp(offsets), None, None))
        return offsets

    # -- products -----------------------------------------------------------------------------
    def spmv_prepare(self, x_ptr: int | None = None):
        """The SpMV analysis (row -> tile map, hot-column table, XCD-sliced structure and the timed choice between the
        two paths on the device vector at ``x_ptr``, or on ones).  Once per matrix; ``spmv_dev`` never analyses."""
        check(_lib.load().spam_spmv_prepare(self._h, C.c_void_p(x_ptr) if x_ptr else None))

    def values_changed(self):
        """Tell the handle that ``vals`` was updated in place (the prepared SpMV plan may hold copies)."""
        check(_lib.load().spam_csr_values_changed(self._h))

    def spmv_dev(self, x_ptr: int, y_ptr: int):
        check(_lib.load().spam_spmv_dev(self._h, C.c_void_p(x_ptr), C.c_void_p(y_ptr)))

    def spmv_plan_info(self) -> dict:
        k, cov, tiles = C.c_uint64(), C.c_double(), C.c_uint64()
        check(_lib.load().spam_spmv_plan_info(self._h, C.byref(k), C.byref(cov), C.byref(tiles)))
        info = {"hot_columns": k.value, "hot_cover": cov.value, "tiles": tiles.value}
        act = C.c_int()
        u = [C.c_uint64() for _ in range(4)]
        d = [C.c_double() for _ in range(4)]
        check(_lib.load().spam_spmv_sliced_info(self._h, C.byref(act), *[C.byref(v) for v in u], *[C.byref(v) for v in d]))
        info["sliced"] = {"active": bool(act.value), "long_rows": u[0].value, "nnz_long": u[1].value,
                          "table_slots_per_xcd": u[2].value, "lds_slots": u[3].value, "cover_table": d[0].value,
                          "cover_single": d[1].value, "ms_plain": d[2].value, "ms_sliced": d[3].value}
        g = [C.c_uint64() for _ in range(5)]
        check(_lib.load().spam_spmv_rg_info(self._h, C.byref(act), *[C.byref(v) for v in g]))
        info["sliced"]["row_groups"] = {"active": bool(act.value), "groups": g[0].value, "blocks": g[1].value,
                                        "accumulators": g[2].value, "stream_bytes": g[3].value, "plan_bytes": g[4].value}
        return info

    def spmv(self, x: np.ndarray) -> np.ndarray:
        rows, cols, _, dt, _ = self.dims()
        x = np.ascontiguousarray(x, dtype=_DT2NP[dt])
        if x.size != cols:
            raise SpamError(1, f"x has {x.size} entries, A has {cols} columns")
        y = np.empty(rows, dtype=_DT2NP[dt])
        check(_typed("spmv", dt)(self._h, _vp(x), _vp(y)))
        return y

    def spgemm(self, rhs: "DeviceCsr", sort_rows=False):
        """C = self * rhs, device-resident.  sort_rows: False/0 table order, True/1 sorted (B2), 2 the
        reference's linprobe slot order.  Returns (DeviceCsr, stats dict)."""
        h = C.c_void_p()
        st = SpgemmStats()
        check(_lib.load().spam_spgemm_dev(self._h, rhs._h, int(sort_rows), C.byref(h), C.byref(st)))
        return DeviceCsr(h), st.asdict()

    def spgemm_symbolic(self, rhs: "DeviceCsr"):
        """Phase 1 (flop count + symbolic + scan): returns (plan handle, nnz(C)).  Reads row_ptr / col_idx only, so it
        can run while B's values are still arriving (multi-GPU: overlap with the all-gather of vals)."""
        plan = C.c_void_p()
        nnz = C.c_uint64()
        check(_lib.load().spam_spgemm_symbolic(self._h, rhs._h, C.byref(plan), C.byref(nnz)))
        return plan, nnz.value

    @staticmethod
    def spgemm_numeric(plan, sort_rows=False):
        """Phase 2 on a plan from spgemm_symbolic: returns (DeviceCsr, stats dict); the plan is consumed and freed."""
        L = _lib.load()
        h = C.c_void_p()
        st = SpgemmStats()
        try:
            check(L.spam_spgemm_numeric_dev(plan, int(sort_rows), C.byref(h)))
            check(L.spam_spgemm_plan_stats(plan, C.byref(st)))
        finally:
            L.spam_spgemm_plan_free(plan)
        return DeviceCsr(h), st.asdict()

    def rank_step(self, rhs: "DeviceCsr", rank: int, world: int, row_overhead: int = _lib.ROW_OVERHEAD_AUTO, sort_rows=False):
        """This rank's block of C = self * rhs on `world` GPUs (spam_spgemm_rank_step): returns (C block or None,
        stats dict, (row_lo, row_hi))."""
        h = C.c_void_p()
        st = SpgemmStats()
        lo, hi = C.c_uint64(), C.c_uint64()
        check(_lib.load().spam_spgemm_rank_step(self._h, rhs._h, rank, world, row_overhead, int(sort_rows), C.byref(h),
                                                C.byref(lo), C.byref(hi), C.byref(st)))
        return (DeviceCsr(h) if h.value else None), st.asdict(), (lo.value, hi.value)

    def rows_to_parts(self, rhs: "DeviceCsr", nparts: int):
        rows = self.dims()[0]
        flops = np.empty(rows, dtype=np.uint64)
        offs = np.empty(nparts + 1, dtype=np.uint64)
        check(_lib.load().spam_rows_to_parts(self._h, rhs._h, nparts, _vp(flops), _vp(offs)))
        return flops, offs

    def balanced_row_blocks(self, rhs: "DeviceCsr", nparts: int, row_overhead: int = _lib.ROW_OVERHEAD_AUTO) -> list:
        """Contiguous row blocks balanced by F + row_overhead per non-empty row, computed on the device."""
        offs = np.empty(nparts + 1, dtype=np.uint64)
        check(_lib.load().spam_rows_to_parts_weighted(self._h, rhs._h, nparts, row_overhead, _vp(offs)))
        return [int(v) for v in offs]

    def free(self):
        if self._h:
            _lib.load().spam_csr_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class CsrMatrix:
    """``CsrMatrix<T, IS_SORTED>``: host Vecs in the reference's layout, products on the GPU."""

    def __init__(self, rows, cols, vals, indices, offsets, is_sorted: bool):
        if rows <= 0 or cols <= 0:
            raise ValueError("rows and cols are NonZeroUsize (spam_csr/src/lib.rs:27-28)")
        self._rows, self._cols = int(rows), int(cols)
        self.vals = np.ascontiguousarray(vals)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint64)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.is_sorted = bool(is_sorted)
        self._dev = None
        self._parts = None  # complex values: the (real, imaginary) CsrMatrix pair, built on first use

    # -- Matrix trait (spam_matrix/src/lib.rs:15-27) ----------------------------------------
    @classmethod
    def new(cls, shape, dtype=np.float64, is_sorted=True):
        rows, cols = shape
        return cls(rows, cols, np.empty(0, dtype), np.empty(0, np.uint64), np.zeros(rows + 1, np.uint64), is_sorted)

    @classmethod
    def new_square(cls, n, dtype=np.float64, is_sorted=True):
        return cls.new((n, n), dtype, is_sorted)

    @classmethod
    def identity(cls, n, dtype=np.float64, is_sorted=True):
        """spam_csr/src/lib.rs:177-185"""
        return cls(n, n, np.ones(n, dtype), np.arange(n, dtype=np.uint64), np.arange(n + 1, dtype=np.uint64), is_sorted)

    def rows(self):
        return self._rows

    def cols(self):
        return self._cols

    def nnz(self):
        return int(self.indices.size)

    def invariants(self) -> bool:
        """Invariants 1-7 (spam_csr/src/lib.rs:47-81) — 1 and 2 on the host, the rest on the device."""
        if self.indices.size != self.vals.size or self.offsets.size != self._rows + 1:
            return False
        try:
            return self.device().invariants()
        except SpamError as e:
            if e.code in (1, 2):
                return False
            raise

    # -- constructors from DOK (spam_csr/src/lib.rs:315-358) --------------------------------
    @classmethod
    def from_dok(cls, shape, entries: dict, dtype=np.float64, rng=None):
        """``From<DokMatrix>`` (sorted rows) or, with ``rng``, ``from_dok(dok, rng)`` (shuffled rows).

        ``entries`` maps (row, col) -> value; zeros are dropped as DokMatrix::set_element does
        (spam_dok/src/lib.rs:171-175)."""
        rows, cols = shape
        items = [(k, v) for k, v in entries.items() if v != 0]
        for (i, j), _ in items:
            if not (0 <= i < rows and 0 <= j < cols):
                raise IndexError("IndexError")  # spam_matrix::IndexError
        if rng is None:
            items.sort(key=lambda kv: kv[0])  # BTreeMap order: (row, col) lexicographic
        else:
            perm = rng.permutation(len(items))
            items = [items[p] for p in perm]
            items.sort(key=lambda kv: kv[0][0])  # stable by row only
        counts = np.zeros(rows, dtype=np.uint64)
        for (i, _), _ in items:
            counts[i] += 1
        offsets = np.zeros(rows + 1, dtype=np.uint64)
        np.cumsum(counts, out=offsets[1:])
        indices = np.array([j for (_, j), _ in items], dtype=np.uint64)
        vals = np.array([v for _, v in items], dtype=dtype)
        return cls(rows, cols, vals, indices, offsets, is_sorted=rng is None)

    def to_dok(self) -> dict:
        """``From<CsrMatrix> for DokMatrix`` (spam_csr/src/lib.rs:375-384): zeros dropped."""
        out = {}
        off = self.offsets.astype(np.int64)
        for r in range(self._rows):
            for p in range(off[r], off[r + 1]):
                v = self.vals[p]
                key = (r, int(self.indices[p]))
                if v != 0:
                    out[key] = v
                else:
                    out.pop(key, None)
        return out

    # -- device ------------------------------------------------------------------------------
    def device(self) -> DeviceCsr:
        """The uploaded copy.  Narrow integers are uploaded as i64; a complex matrix answers with the handle of its
        real part (same structure: enough for invariants() and the dimension queries)."""
        if self.vals.dtype in _COMPLEX_PART:
            return self._real_imag()[0].device()
        if self._dev is None:
            vals = self.vals.astype(_WIDEN[self.vals.dtype]) if self.vals.dtype in _WIDEN else self.vals
            self._dev = DeviceCsr.upload(self._rows, self._cols, self.offsets, self.indices, vals, self.is_sorted)
        return self._dev

    def _device_dtype(self):
        return np.dtype(_WIDEN.get(self.vals.dtype, self.vals.dtype))

    def _narrow(self, vals):
        """Device values back in this matrix's element type (i64 -> i8/i16/i32 keeps the low bits: Wrapping<T>)."""
        return vals.astype(self.vals.dtype) if self.vals.dtype in _WIDEN else vals

    def _real_imag(self):
        if self._parts is None:
            part = _COMPLEX_PART[self.vals.dtype]
            self._parts = tuple(CsrMatrix(self._rows, self._cols, np.ascontiguousarray(v, dtype=part), self.indices,
                                          self.offsets, self.is_sorted) for v in (self.vals.real, self.vals.imag))
        return self._parts

    @staticmethod
    def _same_pattern(*ms):
        o, i = ms[0].offsets, ms[0].indices
        if not all(np.array_equal(m.offsets, o) and np.array_equal(m.indices, i) for m in ms[1:]):
            raise RuntimeError("the real products of a complex operation came back with different patterns")

    # -- the hot path ------------------------------------------------------------------------
    def mul_hash(self, rhs: "CsrMatrix", sort_output: bool = False, reference_order: bool = False) -> "CsrMatrix":
        """``mul_hash::<B1, B2>`` (spam_csr/src/mul_hash.rs:13-36); ``sort_output`` is B2.
        ``reference_order`` (with B2 = false) asks for the reference's exact unsorted layout — linprobe slot
        order — instead of this library's table order (rows of at most 4096 entries).

        Two phases exactly like the reference: symbolic gives nnz, the caller-side Vecs are allocated to
        that capacity (mul_hash.rs:119), numeric fills them."""
        if self.vals.dtype != rhs.vals.dtype:
            raise TypeError(f"both operands hold the same T (mul_hash.rs:13): {self.vals.dtype} vs {rhs.vals.dtype}")
        if self.vals.dtype in _COMPLEX_PART:
            if reference_order:
                raise NotImplementedError("reference slot order is not offered for complex values")
            # four real products, every one with sorted rows so that the four value arrays line up entry by entry
            # (a sorted row is one of the orders B2 = false allows)
            (ar, ai), (br, bi) = self._real_imag(), rhs._real_imag()
            rr, ii, ri, ir = (x.mul_hash(y, sort_output=True) for x, y in ((ar, br), (ai, bi), (ar, bi), (ai, br)))
            self._same_pattern(rr, ii, ri, ir)
            vals = np.empty(rr.vals.size, dtype=self.vals.dtype)
            vals.real = rr.vals - ii.vals
            vals.imag = ri.vals + ir.vals
            return CsrMatrix(self._rows, rhs._cols, vals, rr.indices, rr.offsets, is_sorted=sort_output)
        L = _lib.load()
        a, b = self.device(), rhs.device()
        plan = C.c_void_p()
        nnz = C.c_uint64()
        check(L.spam_spgemm_symbolic(a._h, b._h, C.byref(plan), C.byref(nnz)))
        try:
            offsets = np.empty(self._rows + 1, dtype=np.uint64)
            indices = np.empty(nnz.value, dtype=np.uint64)
            vals = np.empty(nnz.value, dtype=self._device_dtype())
            fn = _typed("spgemm_numeric", _NP2DT[self._device_dtype()])
            order = 1 if sort_output else (2 if reference_order else 0)
            check(fn(plan, order, _vp(offsets), _vp(indices), _vp(vals)))
        finally:
            L.spam_spgemm_plan_free(plan)
        return CsrMatrix(self._rows, rhs._cols, self._narrow(vals), indices, offsets, is_sorted=sort_output)

    def _ewise(self, rhs: "CsrMatrix", subtract: bool) -> "CsrMatrix":
        if self.vals.dtype != rhs.vals.dtype:
            raise TypeError(f"both operands hold the same T: {self.vals.dtype} vs {rhs.vals.dtype}")
        if self.vals.dtype in _COMPLEX_PART:  # (a + ib) +- (c + id), part by part; zeros are kept, so one pattern
            (ar, ai), (br, bi) = self._real_imag(), rhs._real_imag()
            re, im = ar._ewise(br, subtract), ai._ewise(bi, subtract)
            self._same_pattern(re, im)
            vals = np.empty(re.vals.size, dtype=self.vals.dtype)
            vals.real, vals.imag = re.vals, im.vals
            return CsrMatrix(self._rows, self._cols, vals, re.indices, re.offsets, is_sorted=True)
        d = self.device().add(rhs.device(), subtract)
        o, i, v = d.download()
        return CsrMatrix(self._rows, self._cols, self._narrow(v), i, o, is_sorted=True)

    def __add__(self, rhs):
        """``impl Add for CsrMatrix`` (spam_csr/src/lib.rs:276-282)."""
        return self._ewise(rhs, False)

    def __sub__(self, rhs):
        """``impl Sub for CsrMatrix`` (spam_csr/src/lib.rs:284-290)."""
        return self._ewise(rhs, True)

    def __mul__(self, rhs):
        """``impl Mul for &CsrMatrix`` -> ``CsrMatrix<T, false>`` (spam_csr/src/lib.rs:292-297)."""
        return self.mul_hash(rhs, sort_output=False)

    __matmul__ = __mul__

    def spmv(self, x, prepare: bool = False) -> np.ndarray:
        """y = A x (absent upstream; defined as mul_hash with an n x 1 rhs).  ``prepare`` runs the
        SpMV analysis on the (cached) device copy first — worth it when the matrix is multiplied many times."""
        x = np.asarray(x)
        if self.vals.dtype in _COMPLEX_PART:  # y = (Ar xr - Ai xi) + i (Ar xi + Ai xr)
            ar, ai = self._real_imag()
            x = x.astype(self.vals.dtype)
            xr, xi = np.ascontiguousarray(x.real), np.ascontiguousarray(x.imag)
            y = np.empty(self._rows, dtype=self.vals.dtype)
            y.real = ar.spmv(xr, prepare) - ai.spmv(xi, prepare)
            y.imag = ar.spmv(xi) + ai.spmv(xr)
            return y
        d = self.device()
        if prepare:
            d.spmv_prepare()
        if self.vals.dtype in _WIDEN:
            x = x.astype(self.vals.dtype).astype(np.int64)
        return self._narrow(d.spmv(x))
