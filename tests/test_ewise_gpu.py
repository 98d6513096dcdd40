"""Element-wise A + B / A - B on the MI355X (spam_csr_add) against the oracle's apply_elementwise
(spam_csr/src/lib.rs:83-149, 276-290): bit-exact structure AND values for sorted inputs (one add per entry, no
reordering of sums), set-equality for unsorted inputs (the reference's HashMap path has no defined order)."""
import numpy as np
import pytest
import torch

from sparse_matrix_amd import synth
from helpers import assert_csr_parity, csr_np

pytestmark = pytest.mark.gpu


def _mk(spam, m, is_sorted=True):
    off, idx, val = csr_np(m)
    return spam.CsrMatrix(m.rows, m.cols, val, idx, off, is_sorted), (off, idx, val)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_add_sub_sorted_bit_exact(spam, oracle, dtype):
    ma = synth.rmat(12, seed=42, dtype=dtype)
    mb = synth.rmat(12, seed=43, dtype=dtype)          # same shape, different pattern, overlapping in the hot corner
    A, a = _mk(spam, ma)
    B, b = _mk(spam, mb)
    for sub in (False, True):
        C = (A - B) if sub else (A + B)
        wo, wi, wv = oracle.apply_elementwise(a, b, subtract=sub, is_sorted=True)
        assert np.array_equal(C.offsets, wo) and np.array_equal(C.indices, wi) and np.array_equal(C.vals, wv)
        assert C.invariants()
    D = A - A                                          # every entry cancels: structure kept, values exactly 0
    assert np.array_equal(D.indices, a[1]) and np.all(D.vals == 0) and D.to_dok() == {}


def test_add_long_rows_and_empty_rows(spam, oracle):
    ma = synth.powerlaw(30_000, heavy_every=1000, heavy_nnz=20_000, light_nnz=8, seed=1, dtype=torch.float64)
    mb = synth.powerlaw(30_000, heavy_every=700, heavy_nnz=9_000, light_nnz=3, seed=2, dtype=torch.float64)
    A, a = _mk(spam, ma)
    B, b = _mk(spam, mb)
    C = A + B
    wo, wi, wv = oracle.apply_elementwise(a, b)
    assert np.array_equal(C.offsets, wo) and np.array_equal(C.indices, wi) and np.array_equal(C.vals, wv)
    E = spam.CsrMatrix.new((30_000, 30_000))
    F = A + E
    assert np.array_equal(F.offsets, a[0]) and np.array_equal(F.indices, a[1]) and np.array_equal(F.vals, a[2])


def test_add_unsorted_inputs_set_equal(spam, oracle):
    ma = synth.random_csr(300, 400, 0.05, seed=7, signed=False, sort_rows=False)
    mb = synth.random_csr(300, 400, 0.04, seed=8, signed=False, sort_rows=False)
    A, a = _mk(spam, ma, False)
    B, b = _mk(spam, mb, False)
    C = A + B
    want = oracle.apply_elementwise(a, b, is_sorted=False)
    assert_csr_parity((C.offsets, C.indices, C.vals), want, False, 0.0)


def test_add_dimension_mismatch(spam):
    with pytest.raises(spam.SpamError) as e:
        spam.CsrMatrix.identity(4) + spam.CsrMatrix.identity(5)
    assert e.value.code == 1


def test_add_full_size_property():
    """(A + B) x == A x + B x at 10 M rows / 2 x 80 M entries, device-resident."""
    from sparse_matrix_amd import DeviceCsr

    dev = torch.device("cuda", 0)
    A = synth.rmat_rect(10_000_000, 80_000_000, seed=42, device=dev, oversample=1.3)
    B = synth.rmat_rect(10_000_000, 80_000_000, seed=99, device=dev, oversample=1.3)
    dA = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
    dB = DeviceCsr.from_torch(B.rows, B.cols, B.row_ptr, B.col_idx, B.vals, True)
    dC = dA.add(dB)
    assert dC.invariants()
    x = synth.dense_vector(A.cols, device=dev)
    ya, yb, yc = (torch.empty(A.rows, dtype=torch.float64, device=dev) for _ in range(3))
    dA.spmv_dev(x.data_ptr(), ya.data_ptr())
    dB.spmv_dev(x.data_ptr(), yb.data_ptr())
    dC.spmv_dev(x.data_ptr(), yc.data_ptr())
    torch.cuda.synchronize()
    ref = ya + yb
    nz = ref != 0
    assert ((yc - ref).abs()[nz] / ref.abs()[nz]).max().item() <= 1e-10
    assert dC.dims()[2] <= A.nnz + B.nnz
