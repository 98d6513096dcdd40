"""Host-side logic of the CsrMatrix mirror that needs no device: how the element types are routed."""
import numpy as np
import pytest

from sparse_matrix_amd.csr import CsrMatrix


def test_narrow_integers_run_as_i64_and_keep_the_low_bits():
    for dt in (np.int8, np.int16, np.int32):
        A = CsrMatrix.identity(3, dtype=dt)
        assert A._device_dtype() == np.int64
        bits = np.iinfo(dt).bits
        wide = np.array([(1 << bits) + 44, -(1 << (bits - 1)) - 1, -1, 0], dtype=np.int64)
        got = A._narrow(wide)
        assert got.dtype == dt and got.tolist() == [44, (1 << (bits - 1)) - 1, -1, 0]
    for dt in (np.float32, np.float64, np.int64):
        A = CsrMatrix.identity(3, dtype=dt)
        assert A._device_dtype() == dt and A._narrow(A.vals) is A.vals


@pytest.mark.parametrize("dt,part", [(np.complex128, np.float64), (np.complex64, np.float32)])
def test_complex_matrices_split_into_two_real_ones_over_one_pattern(dt, part):
    Z = CsrMatrix(2, 3, np.array([1 + 2j, 3 - 1j, -0.5j], dtype=dt), [0, 2, 1], [0, 2, 3], True)
    re, im = Z._real_imag()
    assert re.vals.dtype == part and im.vals.dtype == part
    assert re.vals.tolist() == [1, 3, 0] and im.vals.tolist() == [2, -1, -0.5]  # zeros of a part are kept
    for p in (re, im):
        assert np.array_equal(p.indices, Z.indices) and np.array_equal(p.offsets, Z.offsets) and p.is_sorted
    assert Z._real_imag()[0] is re


def test_operands_of_different_element_types_are_refused_before_any_device_work():
    A, B = CsrMatrix.identity(4, dtype=np.complex128), CsrMatrix.identity(4, dtype=np.float64)
    with pytest.raises(TypeError):
        A.mul_hash(B)
    with pytest.raises(TypeError):
        _ = A + B
    with pytest.raises(TypeError):
        _ = CsrMatrix.identity(4, dtype=np.int8) - CsrMatrix.identity(4, dtype=np.int64)
