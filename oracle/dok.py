"""The reference's DOK test oracle restated in numpy — small cases only (TEST INFRASTRUCTURE).

* ``to_map``       CSR -> {(row, col): value} with zeros dropped: ``From<CsrMatrix> for DokMatrix``
                   (spam_csr/src/lib.rs:375-384) through ``DokMatrix::set_element`` (spam_dok/src/lib.rs:167-176).
* ``dok_mul``      the O(l*m*n) triple loop ``impl Mul for &DokMatrix`` (spam_dok/src/lib.rs:206-233), wrapping
                   arithmetic for integer dtypes (tests use Wrapping<i8>, spam_csr/src/tests.rs:358).
* ``is_good_approx_of_mul``  Higham (3.13) normwise bound (spam_dok/src/lib.rs:56-92).
* ``invariants``   CsrMatrix invariants 1-7 (spam_csr/src/lib.rs:47-81).
"""
from __future__ import annotations

import numpy as np


def to_dense(rows, cols, offsets, indices, vals):
    d = np.zeros((rows, cols), dtype=vals.dtype)
    offsets = np.asarray(offsets, dtype=np.int64)
    for r in range(rows):
        for p in range(offsets[r], offsets[r + 1]):
            d[r, int(indices[p])] = vals[p]  # set_element: later write replaces (rows hold distinct cols)
    return d


def to_map(rows, offsets, indices, vals):
    out = {}
    offsets = np.asarray(offsets, dtype=np.int64)
    for r in range(rows):
        for p in range(offsets[r], offsets[r + 1]):
            v = vals[p]
            if v != 0:
                out[(r, int(indices[p]))] = v
            else:
                out.pop((r, int(indices[p])), None)
    return out


def dok_mul(a_dense, b_dense):
    """t = t + a[i,k] * b[k,j] for k ascending (spam_dok/src/lib.rs:214-229)."""
    l, m = a_dense.shape
    m2, n = b_dense.shape
    assert m == m2, "LHS cols != RHS rows"  # spam_dok/src/lib.rs:210
    out = np.zeros((l, n), dtype=a_dense.dtype)
    with np.errstate(over="ignore"):
        for k in range(m):
            out = out + a_dense[:, k:k + 1] * b_dense[k:k + 1, :]
    return out


def dense_to_map(d):
    return {(int(i), int(j)): d[i, j] for i, j in zip(*np.nonzero(d))}


def is_good_approx_of_mul(c_dense, a_dense, b_dense) -> bool:
    """||A*B - C||_inf <= 2 * gamma_n * ||A||_inf * ||B||_inf, gamma_n = n*u/(1-n*u), n = cols(C), u = eps/2."""
    inf_norm = lambda m: float(np.abs(m).sum(axis=1).max()) if m.size else 0.0
    n = float(c_dense.shape[1])
    u = np.finfo(np.float64).eps / 2.0
    gamma = n * u / (1.0 - n * u)
    expected = dok_mul(a_dense.astype(np.float64), b_dense.astype(np.float64))
    if not np.isnan(expected).any() and np.isnan(c_dense).any():
        return False
    ln, rn = inf_norm(a_dense), inf_norm(b_dense)
    bound = 0.0 if (ln == 0.0 or rn == 0.0) else ln * rn
    return inf_norm(expected - c_dense) <= 2.0 * gamma * bound


def invariants(rows, cols, offsets, indices, vals, is_sorted: bool) -> bool:
    offsets = np.asarray(offsets).astype(np.int64)
    indices = np.asarray(indices).astype(np.int64)
    if len(indices) != len(vals):                      # 1
        return False
    if len(offsets) != rows + 1:                       # 2
        return False
    if np.any(np.diff(offsets) < 0):                   # 3
        return False
    if offsets[rows] != len(indices):                  # 4
        return False
    if len(indices) and (indices.min() < 0 or indices.max() >= cols):  # 5
        return False
    if offsets[0] != 0:                                # 7
        return False
    if len(indices):                                   # 6 (vectorised: compare neighbours inside rows)
        row_of = np.repeat(np.arange(rows), np.diff(offsets))
        if is_sorted:
            same = row_of[1:] == row_of[:-1]
            if np.any(same & (indices[1:] <= indices[:-1])):
                return False
        else:
            key = row_of * (int(cols) + 1) + indices
            if len(np.unique(key)) != len(key):
                return False
    return True
