"""Matrix Market I/O against the reference's grammar (spam_dok/src/lib.rs:282-490) — hand-written known
answers for every branch, plus the reference's own round-trip property (spam_dok/src/tests.rs:51-66)."""
import math

import pytest
from hypothesis import given, settings, strategies as st

from sparse_matrix_amd.mmio import (FromMatrixMarketError, into_float_matrix_market, parse_matrix_market,
                                    rust_float_display)


def test_real_general_with_comments_zero_drop_and_last_wins():
    text = ("%%MatrixMarket matrix coordinate real general\n% a comment\n%another\n3 4 5\n"
            "1 1 2.5\n3 4 -1e-3\n2 2 0.0\n1 1 7\n2 3 .5\n")
    kind, rows, cols, e = parse_matrix_market(text)
    assert (kind, rows, cols) == ("real", 3, 4)
    assert e == {(0, 0): 7.0, (2, 3): -0.001, (1, 2): 0.5}          # zero dropped, repeated (1,1) keeps the last


def test_integer_symmetric_mirrors_entries():
    kind, rows, cols, e = parse_matrix_market("%%MatrixMarket matrix coordinate integer symmetric\n3 3 2\n2 1 5\n3 3 -4\n")
    assert kind == "integer" and e == {(1, 0): 5, (0, 1): 5, (2, 2): -4}


def test_complex_general_is_never_mirrored():
    kind, _, _, e = parse_matrix_market("%%MatrixMarket matrix coordinate complex general\r\n2 2 1\r\n1 2 1.5 -2\r\n")
    assert kind == "complex" and e == {(0, 1): complex(1.5, -2.0)}


def test_parsing_stops_silently_at_the_first_non_entry_line():
    text = "%%MatrixMarket matrix coordinate real general\n2 2 9\n1 1 1.0\n1  2 3.0\n2 2 4.0\n"   # double space on line 2
    assert parse_matrix_market(text)[3] == {(0, 0): 1.0}


def test_entry_count_is_ignored_and_indices_are_not_range_checked():
    _, rows, cols, e = parse_matrix_market("%%MatrixMarket matrix coordinate real general\n2 2 0\n5 7 1.0\n")
    assert (rows, cols) == (2, 2) and e == {(4, 6): 1.0}


@pytest.mark.parametrize("bad", ["%%MatrixMarket matrix array real general\n1 1\n",
                                 "%%MatrixMarket matrix coordinate real  general\n1 1 0\n",
                                 "%%MatrixMarket matrix coordinate real general\n1 1\n"])
def test_header_and_size_errors(bad):
    with pytest.raises(FromMatrixMarketError):
        parse_matrix_market(bad)


def test_zero_dimension_is_an_error():
    with pytest.raises(FromMatrixMarketError):
        parse_matrix_market("%%MatrixMarket matrix coordinate real general\n0 3 0\n")


@pytest.mark.parametrize("hdr", ["real skew-symmetric", "complex hermitian", "pattern general"])
def test_unsupported_kinds_hit_the_todo(hdr):
    with pytest.raises(NotImplementedError):
        parse_matrix_market(f"%%MatrixMarket matrix coordinate {hdr}\n1 1 0\n")


def test_rust_float_display():
    cases = [(1.0, "1"), (-2.5, "-2.5"), (1e-7, "0.0000001"), (1e21, "1000000000000000000000"), (0.1, "0.1"),
             (123456789.125, "123456789.125"), (5e-324, "0." + "0" * 323 + "5"), (0.0, "0"), (-0.0, "-0")]
    for x, s in cases:
        assert rust_float_display(x) == s
    assert rust_float_display(float("nan")) == "NaN" and rust_float_display(float("-inf")) == "-inf"


def test_writer_layout():
    s = into_float_matrix_market(2, 3, {(1, 2): 0.5, (0, 0): 3.0})
    assert s == "%%MatrixMarket matrix coordinate real general\n2 3 2\n1 1 3\n2 3 0.5\n"


@settings(max_examples=200, deadline=None)
@given(st.integers(1, 4), st.integers(1, 4), st.data())
def test_round_trip(rows, cols, data):
    """spam_dok/src/tests.rs:51-66: write then parse gives the same DokMatrix<f64> (NaN excluded: NaN != NaN)."""
    ents = data.draw(st.dictionaries(st.tuples(st.integers(0, rows - 1), st.integers(0, cols - 1)),
                                     st.floats(allow_nan=False, allow_infinity=False).filter(lambda v: v != 0), max_size=8))
    kind, r, c, back = parse_matrix_market(into_float_matrix_market(rows, cols, ents))
    assert (kind, r, c) == ("real", rows, cols) and back == ents
    assert all(math.copysign(1, back[k]) == math.copysign(1, v) for k, v in ents.items())
