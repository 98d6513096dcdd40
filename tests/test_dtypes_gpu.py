"""The element types the reference instantiates CsrMatrix with beyond f64 / i64, through the GPU path.

* `Wrapping<i8>` is the type of the reference's own hot-path test (spam_csr/src/tests.rs:356-371): unsorted CSR operands
  built from DOK maps, `mul_hash::<false, false>`, result compared as a DOK map with the DOK triple loop — exact.  The
  host mirror runs i8 / i16 / i32 as i64 on the device and keeps the low bits (Z/2^64 -> Z/2^k is a ring homomorphism),
  so the comparison here is exact as well; the checker is oracle/dok.py's triple loop in numpy int8 arithmetic, which
  shares nothing with that route.
* `Complex<f64>` (fuzz targets, MatrixType::Complex in the bench, lib.rs:428-430) runs as four real products over one
  pattern; checked with the reference's own acceptance rule for floating point, Higham's normwise bound
  (spam_dok/src/lib.rs:56-92), restated for complex operands, and entry by entry against scipy.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from sparse_matrix_amd import synth
from helpers import csr_np

pytestmark = pytest.mark.gpu


def _rand_dok(rng, r, c, draw):
    k = int(rng.integers(0, 2 * r * c + 1))
    return {(int(rng.integers(r)), int(rng.integers(c))): draw() for _ in range(k)}


@pytest.mark.parametrize("dtype", [np.int8, np.int16, np.int32])
def test_reference_property_wrapping_narrow_ints(spam, dtype):
    """tests.rs:356-371 as written (Wrapping<i8>), and the same for i16 / i32."""
    from oracle import dok

    info = np.iinfo(dtype)
    rng = np.random.default_rng(int(info.bits))
    for case in range(60):
        l, m, n = (rng.integers(1, 25, 3) if case % 3 == 2 else rng.integers(1, 5, 3))
        draw = lambda: int(rng.integers(info.min, info.max, endpoint=True))
        A = spam.CsrMatrix.from_dok((l, m), _rand_dok(rng, l, m, draw), dtype=dtype, rng=rng)
        B = spam.CsrMatrix.from_dok((m, n), _rand_dok(rng, m, n, draw), dtype=dtype, rng=rng)
        C = A.mul_hash(B, sort_output=False)
        assert C.vals.dtype == dtype and C.invariants()
        ad = dok.to_dense(l, m, A.offsets, A.indices, A.vals)
        bd = dok.to_dense(m, n, B.offsets, B.indices, B.vals)
        want = dok.dense_to_map(dok.dok_mul(ad, bd))  # numpy intN arithmetic wraps like Wrapping<iN>
        got = dok.to_map(l, C.offsets, C.indices, C.vals)
        assert got == want, f"case {case}"
        # y = A x and A + A in the same ring
        x = np.array([draw() for _ in range(m)], dtype=dtype)
        with np.errstate(over="ignore"):
            assert np.array_equal(A.spmv(x), (ad.astype(dtype) @ x).astype(dtype))
        S = A + A
        with np.errstate(over="ignore"):
            assert dok.to_map(l, S.offsets, S.indices, S.vals) == dok.dense_to_map(ad + ad)


def test_i8_rmat_through_every_bin(spam, oracle):
    """Scale 12 (rows of C up to the heavy path): the i8 product equals the low byte of the oracle's i64 product, and
    keeps the computed zeros (a multiple of 256 is an explicit 0, as in the reference)."""
    m = synth.rmat(12)
    off, idx, _ = csr_np(m)
    v8 = np.random.default_rng(3).integers(-128, 127, idx.size, dtype=np.int8, endpoint=True)
    A = spam.CsrMatrix(m.rows, m.cols, v8, idx, off, True)
    C = A.mul_hash(A, sort_output=True)
    wo, wi, wv = oracle.mul_hash((off, idx, v8.astype(np.int64)), (off, idx, v8.astype(np.int64)), sort_rows=True)
    assert np.array_equal(C.offsets, wo) and np.array_equal(C.indices, wi)
    assert C.vals.dtype == np.int8 and np.array_equal(C.vals, wv.astype(np.int8))
    assert (C.vals == 0).any(), "the case is meant to hold explicit zeros"


def _cvals(rng, n, dtype):
    return (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(dtype)


def _scipy(M):
    return sp.csr_matrix((M.vals, M.indices.astype(np.int64), M.offsets.astype(np.int64)), shape=(M.rows(), M.cols()))


@pytest.mark.parametrize("dtype,u", [(np.complex128, 2.0 ** -53), (np.complex64, 2.0 ** -24)])
def test_complex_products_against_the_normwise_bound(spam, dtype, u):
    rng = np.random.default_rng(9)
    inf_norm = lambda d: float(np.abs(d).sum(axis=1).max()) if d.size else 0.0
    for case in range(40):
        l, m, n = rng.integers(1, 25, 3)
        draw = lambda: complex(rng.uniform(-1, 1), rng.uniform(-1, 1))
        A = spam.CsrMatrix.from_dok((l, m), _rand_dok(rng, l, m, draw), dtype=dtype, rng=rng)
        B = spam.CsrMatrix.from_dok((m, n), _rand_dok(rng, m, n, draw), dtype=dtype, rng=rng)
        C = A.mul_hash(B, sort_output=bool(case & 1))
        assert C.vals.dtype == dtype and C.invariants()
        ad, bd, cd = _scipy(A).toarray().astype(np.complex128), _scipy(B).toarray().astype(np.complex128), _scipy(C).toarray()
        # Higham (3.13) as in is_good_approx_of_mul (spam_dok/src/lib.rs:56-92); a complex product costs a few real
        # operations, covered by gamma_{n+2} and the factor sqrt(2) of |re| + |im| <= sqrt(2) |z|
        nn = float(n + 2)
        gamma = nn * u / (1 - nn * u)
        assert inf_norm(ad @ bd - cd) <= 2 * np.sqrt(2) * gamma * inf_norm(ad) * inf_norm(bd)
        # the pattern is the symbolic one: products that cancel stay as explicit entries
        pat = (np.abs(ad) != 0).astype(np.int64) @ (np.abs(bd) != 0).astype(np.int64)
        assert C.nnz() == int((pat != 0).sum())


def test_complex_rmat_spgemm_spmv_add(spam):
    m = synth.rmat(11)
    off, idx, _ = csr_np(m)
    rng = np.random.default_rng(4)
    A = spam.CsrMatrix(m.rows, m.cols, _cvals(rng, idx.size, np.complex128), idx, off, True)
    B = spam.CsrMatrix(m.rows, m.cols, _cvals(rng, idx.size, np.complex128), idx, off, True)
    sa, sb = _scipy(A), _scipy(B)
    C = A.mul_hash(B, sort_output=True)
    want = (sa @ sb).tocsr()
    want.sort_indices()
    # scipy prunes nothing either; compare through absolute products (no cancellation in the tolerance's scale)
    scale = (abs(sa) @ abs(sb)).tocsr()
    scale.sort_indices()
    assert np.array_equal(C.offsets.astype(np.int64), want.indptr) and np.array_equal(C.indices.astype(np.int64), want.indices)
    assert np.all(np.abs(C.vals - want.data) <= 1e-13 * scale.data)
    x = _cvals(rng, m.cols, np.complex128)
    y = A.spmv(x, prepare=True)
    assert np.all(np.abs(y - sa @ x) <= 1e-13 * (abs(sa) @ np.abs(x)) + 1e-300)
    for sub in (False, True):
        S = (A - B) if sub else (A + B)
        w = (sa - sb) if sub else (sa + sb)
        # A and B share one pattern: the sum keeps every position, computed zeros included (scipy prunes nothing here)
        assert np.array_equal(S.offsets, A.offsets) and np.array_equal(S.indices, A.indices)
        assert np.array_equal(S.vals, A.vals - B.vals if sub else A.vals + B.vals)
        assert abs(_scipy(S) - w).max() == 0


def test_mixed_element_types_are_refused(spam):
    A = spam.CsrMatrix.identity(4, dtype=np.complex128)
    B = spam.CsrMatrix.identity(4, dtype=np.float64)
    with pytest.raises(TypeError):
        A.mul_hash(B)
    with pytest.raises(TypeError):
        _ = A + B
