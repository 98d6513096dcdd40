"""Wrapping<i64> through the GPU path: the reference's own hot-path property with exact equality.

The reference tests mul_hash on `Wrapping<i8>` matrices built unsorted from DOK and compares the product, as a DOK map,
with the DOK triple loop — exact equality, integers have no rounding (spam_csr/src/tests.rs:356-371) — and its bench
runs `MatrixType::Integer` files as i64 (spam_csr/src/lib.rs:421-424).  Integer sums do not depend on the order of the
additions, so here the GPU result must equal the oracle BIT FOR BIT in structure and values, which the f64 path (order-
dependent rounding) cannot offer: this is the strongest parity leg available while the float oracle stays unpinned.
"""
import numpy as np
import pytest
import torch

from sparse_matrix_amd import synth
from helpers import assert_csr_exact, csr_np

pytestmark = pytest.mark.gpu


def _ivals(n, seed, small=False):
    rng = np.random.default_rng(seed)
    if small:
        return rng.integers(-3, 4, n, dtype=np.int64)  # many cancellations: explicit zeros must be kept
    v = rng.integers(np.iinfo(np.int64).min, np.iinfo(np.int64).max, n, dtype=np.int64, endpoint=True)
    v[v == 0] = 1
    return v


def _as_i64(m, seed, small=False):
    off, idx, _ = csr_np(m)
    return off, idx, _ivals(idx.size, seed, small)


def _all_orders(spam, oracle, a, b, rows, inner, cols, a_sorted=True, b_sorted=True):
    A = spam.CsrMatrix(rows, inner, a[2], a[1], a[0], a_sorted)
    B = spam.CsrMatrix(inner, cols, b[2], b[1], b[0], b_sorted)
    ws = oracle.mul_hash(a, b, sort_rows=True)
    wu = oracle.mul_hash(a, b, sort_rows=False)
    Cs = A.mul_hash(B, sort_output=True)
    assert Cs.vals.dtype == np.int64
    assert_csr_exact((Cs.offsets, Cs.indices, Cs.vals), ws, True)
    Ct = A.mul_hash(B)
    assert_csr_exact((Ct.offsets, Ct.indices, Ct.vals), wu, False)
    Cr = A.mul_hash(B, reference_order=True)  # the reference's unsorted layout: linprobe slot order, values attached
    assert_csr_exact((Cr.offsets, Cr.indices, Cr.vals), wu, True)
    return Cs


def test_reference_property_wrapping_vs_dok_triple_loop(spam, oracle):
    """tests.rs:356-371 with i64: unsorted CSR operands from DOK maps (dimensions 1..4 and up to 24), product through
    mul_hash::<false, false> on the GPU, compared as zero-dropped DOK maps with the triple loop — exact."""
    from oracle import dok

    rng = np.random.default_rng(77)
    for case in range(100):
        l, m, n = (rng.integers(1, 25, 3) if case % 3 == 2 else rng.integers(1, 5, 3))
        small = case % 2 == 0
        def rand_dok(r, c):
            k = int(rng.integers(0, 2 * r * c + 1))
            vals = _ivals(k, int(rng.integers(1 << 30)), small)
            return {(int(rng.integers(r)), int(rng.integers(c))): int(v) for v in vals}
        A = spam.CsrMatrix.from_dok((l, m), rand_dok(l, m), dtype=np.int64, rng=rng)
        B = spam.CsrMatrix.from_dok((m, n), rand_dok(m, n), dtype=np.int64, rng=rng)
        C = A.mul_hash(B, sort_output=False)
        assert C.invariants()
        ad = dok.to_dense(l, m, A.offsets, A.indices, A.vals)
        bd = dok.to_dense(m, n, B.offsets, B.indices, B.vals)
        want = dok.dense_to_map(dok.dok_mul(ad, bd))
        got = dok.to_map(l, C.offsets, C.indices, C.vals)
        assert got == want, f"case {case}"
        wu = oracle.mul_hash((A.offsets, A.indices, A.vals), (B.offsets, B.indices, B.vals), sort_rows=False)
        assert_csr_exact((C.offsets, C.indices, C.vals), wu, False)


@pytest.mark.parametrize("scale,small", [(8, False), (11, True), (13, False)])
def test_rmat_square_every_bin(spam, oracle, scale, small):
    """R-MAT A*A with full-range (wrapping) and small (cancelling) values: every symbolic / numeric bin up to the heavy
    path (scale 13: rows of C beyond 4096 entries), all three output orders, values bit-exact."""
    m = synth.rmat(scale)
    a = _as_i64(m, 5 + scale, small)
    _all_orders(spam, oracle, a, a, m.rows, m.cols, m.cols)


def test_rectangular_unsorted_operands_and_wide_b(spam, oracle):
    ma = synth.random_csr(200, 300, 0.05, seed=31, signed=False, sort_rows=False)
    mb = synth.random_csr(300, 40_000, 0.02, seed=32, signed=False, sort_rows=False)
    _all_orders(spam, oracle, _as_i64(ma, 1), _as_i64(mb, 2), 200, 300, 40_000, a_sorted=False, b_sorted=False)
    ma = synth.random_csr(6, 2000, 0.25, seed=41, signed=False)
    mb = synth.random_csr(2000, 600_000, 0.001, seed=42, signed=False)  # heavy rows over many chunks
    _all_orders(spam, oracle, _as_i64(ma, 3), _as_i64(mb, 4), 6, 2000, 600_000)


def test_spmv_exact(spam, oracle, monkeypatch):
    """y = A x in wrapping i64: exact on the plain tile path (unprepared and prepared) and on the forced XCD-sliced one."""
    for forced in ("0", "2"):
        monkeypatch.setenv("SPAM_SPMV_SLICED", forced)
        monkeypatch.setenv("SPAM_SPMV_SLICE_KB", "4")
        for m in (synth.rmat(13), synth.powerlaw(20_000, heavy_every=500, heavy_nnz=9_000, light_nnz=6, dtype=torch.float64),
                  synth.banded(5000, 5)):
            off, idx, val = _as_i64(m, 9)
            x = _ivals(m.cols, 10)
            A = spam.CsrMatrix(m.rows, m.cols, val, idx, off, True)
            ref = oracle.spmv((off, idx, val), x)
            for prepare in (False, True):
                y = A.spmv(x, prepare=prepare)
                assert y.dtype == np.int64 and np.array_equal(y, ref), (forced, prepare)
            if forced == "2" and m.rows > 6000:
                assert A.device().spmv_plan_info()["sliced"]["active"]


def test_add_sub_exact(spam, oracle):
    ma = synth.random_csr(300, 500, 0.03, seed=7, signed=False)
    mb = synth.random_csr(300, 500, 0.03, seed=8, signed=False)
    a, b = _as_i64(ma, 11, True), _as_i64(mb, 12, True)
    A = spam.CsrMatrix(300, 500, a[2], a[1], a[0], True)
    B = spam.CsrMatrix(300, 500, b[2], b[1], b[0], True)
    for sub, C in ((False, A + B), (True, A - B)):
        want = oracle.apply_elementwise(a, b, subtract=sub)
        assert_csr_exact((C.offsets, C.indices, C.vals), want, True)


def test_integer_matrix_market_runs_as_i64(spam, oracle, tmp_path):
    """MatrixType::Integer files are parsed as i64 and multiplied with wrapping arithmetic (lib.rs:421-424)."""
    from sparse_matrix_amd.mmio import parse_matrix_market

    m = synth.rmat(7)
    off, idx, val = _as_i64(m, 21)
    lines = ["%%MatrixMarket matrix coordinate integer general\n", f"{m.rows} {m.cols} {idx.size}\n"]
    o = off.astype(np.int64)
    for r in range(m.rows):
        for p in range(o[r], o[r + 1]):
            lines.append(f"{r + 1} {int(idx[p]) + 1} {int(val[p])}\n")
    f = tmp_path / "ints.mtx"
    f.write_text("".join(lines))
    kind, rows, cols, ents = parse_matrix_market(f.read_text())
    assert kind == "integer"
    A = spam.CsrMatrix.from_dok((rows, cols), ents, dtype=np.int64)
    assert np.array_equal(A.vals, val) and np.array_equal(A.indices, idx)
    C = A * A
    want = oracle.mul_hash((off, idx, val), (off, idx, val), sort_rows=False)
    assert_csr_exact((C.offsets, C.indices, C.vals), want, False)


def test_cfg2_full_size_spmv_is_bit_exact_in_i64(spam, oracle):
    """BASELINE configs[1]'s matrix (10 M rows, ~162.7 M entries) with Wrapping<i64> values of full range: the prepared
    plan (XCD-sliced, row-group form — the kernels the headline number is measured on) and the unprepared plain path
    must both equal the oracle BIT FOR BIT over all 10 M rows: integer sums do not depend on the order of the LDS
    atomics, so this is the one check of the headline path, at its full size, that has no tolerance in it."""
    from sparse_matrix_amd import DeviceCsr
    from sparse_matrix_amd.synth import mix64

    dev = torch.device("cuda", 0)
    A = synth.rmat_rect(10_000_000, 160_000_000, dtype=torch.float64, device=dev, oversample=1.30)
    vals = mix64(torch.arange(A.nnz, dtype=torch.int64, device=dev) + 12345)          # all 64 bits in use: sums wrap
    x = mix64(torch.arange(A.cols, dtype=torch.int64, device=dev) * 3 + 777)
    dA = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, vals, True)
    y_plain = torch.empty(A.rows, dtype=torch.int64, device=dev)
    y_prep = torch.empty_like(y_plain)
    dA.spmv_dev(x.data_ptr(), y_plain.data_ptr())
    dA.spmv_prepare(x.data_ptr())
    info = dA.spmv_plan_info()["sliced"]
    assert info["active"] and info["row_groups"]["active"], info
    for _ in range(2):  # (twice: nothing a product leaves behind may change the next one)
        y_prep.fill_(-1)
        dA.spmv_dev(x.data_ptr(), y_prep.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(y_prep, y_plain)
    off = A.row_ptr.cpu().numpy().astype(np.uint64)
    idx = A.col_idx.cpu().numpy().astype(np.uint64)
    ref = oracle.spmv((off, idx, vals.cpu().numpy()), x.cpu().numpy(), nthreads=oracle.hardware_threads())
    assert np.array_equal(y_prep.cpu().numpy(), ref)
