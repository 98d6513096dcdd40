"""Matrix Market coordinate I/O with the reference's grammar (host-side text handling, no GPU involved).

Restates ``parse_matrix_market`` (spam_dok/src/lib.rs:282-478) and ``into_float_matrix_market``
(spam_dok/src/lib.rs:480-490) so the reference's bench shape — for every file in ``./matrices``: parse,
``CsrMatrix::from(dok)``, time ``m.mul_hash(&m)`` (spam_csr/src/lib.rs:389-435) — can run on the GPU path
(``tools/bench_mm.py``).  Behaviour kept, including the odd corners:

* header must be exactly ``%%MatrixMarket matrix coordinate <integer|real|complex|pattern>
  <general|symmetric|skew-symmetric|hermitian>``; ``pattern``, ``skew-symmetric`` and ``hermitian`` are
  recognised and then unsupported (the reference hits ``todo!()``): ``NotImplementedError`` here;
* fields are separated by exactly one space, lines end in ``\\n`` or ``\\r\\n``;
* the entry count on the size line is ignored; parsing stops silently at the first line that is not an
  entry (``fold_many0``), whatever follows is dropped;
* indices are 1-based and NOT range-checked; zeros are dropped; a repeated coordinate keeps the LAST value;
  ``symmetric`` mirrors every entry (also for the diagonal); complex entries are never mirrored;
* a zero dimension is an error (``HasZeroDimension``).
"""
from __future__ import annotations

import re
from decimal import Decimal

_INT = r"-?[0-9]+"
# nom::number::complete::recognize_float: [+-]? (digits [. digits?]? | . digits) ([eE] [+-]? digits)?
_FLOAT = r"[+-]?(?:[0-9]+(?:\.[0-9]*)?|\.[0-9]+)(?:[eE][+-]?[0-9]+)?"
_EOL = r"\r?\n"
_HEADER = re.compile(r"%%MatrixMarket matrix coordinate (integer|real|complex|pattern) "
                     r"(general|symmetric|skew-symmetric|hermitian)" + _EOL)
_COMMENT = re.compile(r"%[^\r\n]*" + _EOL)
_SIZE = re.compile(rf"({_INT}) ({_INT}) ({_INT})" + _EOL)
_ENTRY = {
    "integer": re.compile(rf"({_INT}) ({_INT}) ({_INT})" + _EOL),
    "real": re.compile(rf"({_INT}) ({_INT}) ({_FLOAT})" + _EOL),
    "complex": re.compile(rf"({_INT}) ({_INT}) ({_FLOAT}) ({_FLOAT})" + _EOL),
}


class FromMatrixMarketError(ValueError):
    """``FromMatrixMarketError::{Nom, HasZeroDimension}`` (spam_dok/src/lib.rs:274-280)."""


def _usize(s: str) -> int:
    v = int(s)
    if v < 0:
        raise FromMatrixMarketError("parsing error: negative usize")  # str::parse::<usize> fails -> nom error
    return v


def parse_matrix_market(text: str):
    """Returns (kind, rows, cols, entries) with kind in {"integer", "real", "complex"} and entries a dict
    {(row, col): value} holding 0-based coordinates — the fields of the DokMatrix the reference builds."""
    m = _HEADER.match(text)
    if not m:
        raise FromMatrixMarketError("parsing error: header")
    kind, shape = m.group(1), m.group(2)
    if shape not in ("general", "symmetric"):
        raise NotImplementedError(f"matrix shape {shape} unsupported")  # todo!() (lib.rs:369)
    pos = m.end()
    while True:
        c = _COMMENT.match(text, pos)
        if not c:
            break
        pos = c.end()
    s = _SIZE.match(text, pos)
    if not s:
        raise FromMatrixMarketError("parsing error: size line")
    rows, cols = _usize(s.group(1)), _usize(s.group(2))
    _usize(s.group(3))
    pos = s.end()
    if kind == "pattern":
        raise NotImplementedError("entry type pattern unsupported")  # todo!() (lib.rs:447)
    entries = {}
    pat = _ENTRY[kind]
    while True:
        e = pat.match(text, pos)
        if not e:
            break  # fold_many0 stops; the remainder of the input is ignored
        try:
            r, c = _usize(e.group(1)), _usize(e.group(2))
        except FromMatrixMarketError:
            break
        if kind == "integer":
            v = int(e.group(3))
            if not -(1 << 63) <= v < (1 << 63):
                break  # i64 parse fails -> the entry parser fails -> fold stops
        elif kind == "real":
            v = float(e.group(3))
        else:
            v = complex(float(e.group(3)), float(e.group(4)))
        pos = e.end()
        if v != 0:
            if r == 0 or c == 0:
                raise OverflowError("index 0 in a 1-based file: the reference underflows r - 1")
            entries[(r - 1, c - 1)] = v
            if shape == "symmetric" and kind != "complex":
                entries[(c - 1, r - 1)] = v
    if rows == 0 or cols == 0:
        raise FromMatrixMarketError("number of rows or columns is 0")
    return kind, rows, cols, entries


def rust_float_display(x: float) -> str:
    """``format!("{}", x)`` for f64: shortest digits that round-trip, never an exponent, no trailing ``.0``."""
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "inf" if x > 0 else "-inf"
    s = format(Decimal(repr(x)), "f")
    if "." in s:
        s = s.rstrip("0").rstrip(".")
    if s in ("-0", "0"):
        return "-0" if str(x).startswith("-") else "0"
    return s


def into_float_matrix_market(rows: int, cols: int, entries: dict) -> str:
    """``into_float_matrix_market`` (spam_dok/src/lib.rs:480-490): entries in BTreeMap = (row, col) order."""
    out = ["%%MatrixMarket matrix coordinate real general\n", f"{rows} {cols} {len(entries)}\n"]
    for (i, j) in sorted(entries):
        out.append(f"{i + 1} {j + 1} {rust_float_display(float(entries[(i, j)]))}\n")
    return "".join(out)
