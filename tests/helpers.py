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
