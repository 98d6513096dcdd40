"""The synthetic generators: deterministic, valid CsrMatrix<T, true> layouts, stated sizes."""
import numpy as np
import torch

from oracle import dok
from sparse_matrix_amd import synth


def _ok(m, is_sorted=True):
    o, i, v = m.cpu_numpy()
    return dok.invariants(m.rows, m.cols, o, i, v, is_sorted)


def test_banded_counts():
    m = synth.banded(10000, 5)
    assert m.nnz == 49994 and _ok(m)
    assert float(m.vals.min()) >= 0.5 and float(m.vals.max()) < 1.5


def test_rmat_is_deterministic_and_chunk_independent():
    a = synth.rmat(10, chunk=1 << 24)
    b = synth.rmat(10, chunk=1 << 11)
    assert torch.equal(a.row_ptr, b.row_ptr) and torch.equal(a.col_idx, b.col_idx) and torch.equal(a.vals, b.vals)
    assert _ok(a) and a.rows == 1024 and 0.6 * 16 * 1024 < a.nnz <= 16 * 1024


def test_rmat_row_blocks_tile_the_matrix():
    """Each GPU generates only its row block; together they are exactly the full matrix."""
    full = synth.rmat(9)
    n = full.rows
    cols, vals, counts = [], [], []
    for lo, hi in ((0, 100), (100, 300), (300, n)):
        blk = synth.rmat(9, row_range=(lo, hi))
        assert blk.rows == hi - lo and _ok(blk)
        cols.append(blk.col_idx), vals.append(blk.vals), counts.append(blk.row_ptr[1:] - blk.row_ptr[:-1])
    assert torch.equal(torch.cat(cols), full.col_idx) and torch.equal(torch.cat(vals), full.vals)
    assert torch.equal(torch.cat(counts), full.row_ptr[1:] - full.row_ptr[:-1])


def test_rmat_rect_clips():
    m = synth.rmat_rect(3000, 40000)
    assert m.rows == m.cols == 3000 and _ok(m) and int(m.col_idx.max()) < 3000


def test_powerlaw_shape():
    m = synth.powerlaw(20000, heavy_every=1000, heavy_nnz=5000, light_nnz=8, dtype=torch.float32)
    d = (m.row_ptr[1:] - m.row_ptr[:-1]).numpy()
    assert sorted(np.unique(d).tolist()) == [8, 5000] and (d == 5000).sum() == 20
    assert m.nnz == 20 * 5000 + (20000 - 20) * 8 and _ok(m)


def test_random_csr_unsorted_rows_are_valid_unsorted():
    m = synth.random_csr(50, 60, 0.2, seed=1, sort_rows=False)
    assert _ok(m, is_sorted=False) and not _ok(m, is_sorted=True)
