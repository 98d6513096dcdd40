"""Shared helpers for the parity tests (oracle = checker, product = libspam_hip.so through its C ABI)."""
import numpy as np


def csr_np(m):
    """synth.Csr -> (offsets u64, indices u64, vals) in the reference's host layout."""
    return m.cpu_numpy()


def rows_as_sets(offsets, indices):
    off = np.asarray(offsets).astype(np.int64)
    idx = np.asarray(indices).astype(np.int64)
    return [frozenset(idx[off[r]:off[r + 1]].tolist()) for r in range(len(off) - 1)]


def sort_rows(offsets, indices, vals):
    """Canonical per-row column order (what B2 = true produces)."""
    off = np.asarray(offsets).astype(np.int64)
    idx = np.asarray(indices).astype(np.int64)
    row_of = np.repeat(np.arange(len(off) - 1), np.diff(off))
    order = np.lexsort((idx, row_of))
    return idx[order], np.asarray(vals)[order]


def assert_csr_parity(got, want, sorted_output, rtol):
    """got / want = (offsets, indices, vals).  row_ptr bit-exact; col_idx bit-exact when sorted, per-row
    set-equal otherwise; values within rtol relative (compared after canonical ordering)."""
    go, gi, gv = got
    wo, wi, wv = want
    assert np.array_equal(np.asarray(go, dtype=np.uint64), np.asarray(wo, dtype=np.uint64)), "row_ptr differs"
    if sorted_output:
        assert np.array_equal(np.asarray(gi, dtype=np.uint64), np.asarray(wi, dtype=np.uint64)), "col_idx differs"
    gi_s, gv_s = sort_rows(go, gi, gv)
    wi_s, wv_s = sort_rows(wo, wi, wv)
    assert np.array_equal(gi_s, wi_s), "per-row column sets differ"
    if len(wv_s):
        err = np.abs(gv_s.astype(np.float64) - wv_s.astype(np.float64))
        scale = np.maximum(np.abs(wv_s.astype(np.float64)), np.finfo(np.float64).tiny)
        assert float((err / scale).max()) <= rtol, f"max rel err {(err / scale).max():.3e} > {rtol}"


def download_view_rows(C, lo, hi):
    """Rows [lo, hi) of a device matrix as (offsets from 0, indices, vals).  A row-block view starts its arrays at
    a 16-byte boundary, i.e. up to 3 entries before the first row: those are dropped here."""
    go, gi, gv = C.row_block(lo, hi).download()
    pad = int(go[0])
    return go - go[0], gi[pad:], gv[pad:]


def spgemm_rows_vs_oracle(oracle, dA, dB, C, a_host, b_host, row_lo, target_products, rtol=1e-10, ref_order=True):
    """Rows [row_lo, r1) of C = A*B against the oracle's mul_hash on the same rows (r1 chosen so that the sample holds
    about `target_products` intermediate products): row_ptr exact, per-row column sets equal, values within rtol — and,
    recomputed with sort_rows = 2 on those rows, col_idx bit-identical to the reference's linprobe slot order.
    a_host / b_host are the host copies (offsets, indices, vals) of the operands dA / dB were built from.
    Returns (r1, products in the sample, entries of C in the sample)."""
    flops_row, _ = oracle.rows_to_threads(a_host[0], a_host[1], b_host[0], 1)
    ps = np.concatenate([[0], np.cumsum(flops_row.astype(np.float64))])
    n = len(flops_row)
    r1 = int(np.searchsorted(ps, ps[row_lo] + min(target_products, ps[-1] - ps[row_lo])))
    r1 = max(row_lo + 1, min(r1, n))
    s, e = int(a_host[0][row_lo]), int(a_host[0][r1])
    sub = (a_host[0][row_lo:r1 + 1] - a_host[0][row_lo], a_host[1][s:e], a_host[2][s:e])
    want = oracle.mul_hash(sub, b_host, sort_rows=False)
    assert_csr_parity(download_view_rows(C, row_lo, r1), want, False, rtol)
    if ref_order:
        C2, _ = dA.row_block(row_lo, r1).spgemm(dB, sort_rows=2)
        o2, i2, v2 = C2.download()
        C2.free()
        assert np.array_equal(o2, want[0]), "row_ptr (reference order run)"
        assert np.array_equal(i2, want[1]), "unsorted col_idx must equal the reference's linprobe slot order"
        if len(want[2]):
            assert float(np.max(np.abs(v2 - want[2]) / np.abs(want[2]))) <= rtol
    return r1, float(ps[r1] - ps[row_lo]), len(want[1])


def assert_csr_exact(got, want, sorted_output):
    """Integer values (Wrapping<i64>): everything bit-exact — row_ptr, col_idx (as stored when sorted or in reference
    order, per-row sets otherwise) and the values themselves (integer sums do not depend on the summation order)."""
    go, gi, gv = got
    wo, wi, wv = want
    assert np.array_equal(np.asarray(go, dtype=np.uint64), np.asarray(wo, dtype=np.uint64)), "row_ptr differs"
    if sorted_output:
        assert np.array_equal(np.asarray(gi, dtype=np.uint64), np.asarray(wi, dtype=np.uint64)), "col_idx differs"
        assert np.array_equal(gv, wv), "values differ"
    else:
        gi_s, gv_s = sort_rows(go, gi, gv)
        wi_s, wv_s = sort_rows(wo, wi, wv)
        assert np.array_equal(gi_s, wi_s), "per-row column sets differ"
        assert np.array_equal(gv_s, wv_s), "values differ"
