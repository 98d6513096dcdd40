"""SpMV parity on the MI355X: libspam_hip.so (through the C ABI) against the CPU oracle.

The reference has no SpMV; the oracle's definition is mul_hash with an n x 1 right-hand side
(SURVEY.md §3.5).  Tolerance: 1e-10 relative for f64 (BASELINE.json north_star), 1e-5 for f32 against
the f64-accumulated oracle; inputs are positive so relative error is meaningful.
"""
import numpy as np
import pytest
import torch

from sparse_matrix_amd import synth
from helpers import csr_np

pytestmark = pytest.mark.gpu

F64_RTOL = 1e-10
F32_RTOL = 1e-5


def _check(spam, oracle, m, rtol, is_sorted=True):
    off, idx, val = csr_np(m)
    x = synth.dense_vector(m.cols, dtype=m.vals.dtype).numpy()
    A = spam.CsrMatrix(m.rows, m.cols, val, idx, off, is_sorted)
    if val.dtype == np.float32:
        ref = oracle.spmv((off, idx, val.astype(np.float64)), x.astype(np.float64))
    else:
        ref = oracle.spmv((off, idx, val), x)
    nz = ref != 0
    # a handle multiplies unprepared (plain tile path, no analysis) and prepared (spam_spmv_prepare: hot-column table,
    # XCD-sliced structure where it qualifies): both against the oracle
    for prepare in (False, True):
        y = A.spmv(x, prepare=prepare)
        assert y.shape == ref.shape
        assert np.array_equal(y[~nz], ref[~nz].astype(y.dtype)), "rows without entries must give exactly 0"
        rel = np.abs(y[nz].astype(np.float64) - ref[nz]) / np.abs(ref[nz])
        assert rel.size == 0 or rel.max() <= rtol, f"max rel err {rel.max():.3e} (prepare={prepare})"
        if not prepare:
            y0 = y
    return y0, ref


def test_banded_cfg1(spam, oracle):
    """BASELINE configs[0]: 10k x 10k banded bw=5.  A row that lies inside one nnz tile is summed by one
    lane left to right — the oracle's order — so it is bit-identical; only rows cut by a tile boundary
    (every 2048 / 4096 entries) are summed as partial + partial and may differ in the last bit."""
    m = synth.banded(10000, 5)
    y, ref = _check(spam, oracle, m, F64_RTOL)
    off = m.row_ptr.numpy()
    inside = (off[:-1] // 2048) == ((off[1:] - 1) // 2048)
    assert inside.sum() > 9900
    assert np.array_equal(y[inside], ref[inside])


@pytest.mark.parametrize("scale", [8, 12, 16])
def test_rmat_f64(spam, oracle, scale):
    _check(spam, oracle, synth.rmat(scale), F64_RTOL)


def test_rmat_rect_f64(spam, oracle):
    """Non power-of-two clip of R-MAT (the cfg2 generator) at 300k rows."""
    _check(spam, oracle, synth.rmat_rect(300_000, 4_800_000), F64_RTOL)


def test_powerlaw_f32_long_rows(spam, oracle):
    """cfg5 shape at 1/50 scale: rows of 50 000 entries span a dozen tiles (carry path) next to rows of 8."""
    _check(spam, oracle, synth.powerlaw(40_000, heavy_every=1000, heavy_nnz=30_000, light_nnz=8, dtype=torch.float32), F32_RTOL)


def test_powerlaw_f64_long_rows(spam, oracle):
    _check(spam, oracle, synth.powerlaw(60_000, heavy_every=500, heavy_nnz=50_000, light_nnz=8,
                                        dtype=torch.float64), F64_RTOL)


def test_single_dense_row_and_empty_rows(spam, oracle):
    """One row holding every entry, all other rows empty; plus trailing empty rows at a tile boundary."""
    n = 9000
    off = np.zeros(n + 1, dtype=np.uint64)
    off[5:] = n  # row 4 has n entries
    idx = np.arange(n, dtype=np.uint64)
    val = np.linspace(0.5, 1.5, n)
    A = spam.CsrMatrix(n, n, val, idx, off, True)
    x = synth.dense_vector(n).numpy()
    y = A.spmv(x)
    ref = oracle.spmv((off, idx, val), x)
    assert np.all(y[np.arange(n) != 4] == 0)
    assert abs(y[4] - ref[4]) <= F64_RTOL * abs(ref[4])


def test_rows_exactly_on_tile_boundaries(spam, oracle):
    """Rows of 4096 entries: every row starts exactly where a 4096-entry tile starts (and 2048 for NPT=8)."""
    rows, cols, L = 6, 5000, 4096
    off = (np.arange(rows + 1) * L).astype(np.uint64)
    idx = np.tile(np.arange(L, dtype=np.uint64), rows)
    val = 0.5 + np.random.default_rng(0).random(rows * L)
    A = spam.CsrMatrix(rows, cols, val, idx, off, True)
    x = synth.dense_vector(cols).numpy()
    y = A.spmv(x)
    ref = oracle.spmv((off, idx, val), x)
    assert np.abs(y - ref).max() <= F64_RTOL * np.abs(ref).max()


def test_empty_matrix(spam):
    A = spam.CsrMatrix.new((7, 3))
    assert np.array_equal(A.spmv(np.ones(3)), np.zeros(7))


def test_unsorted_rows(spam, oracle):
    _check(spam, oracle, synth.random_csr(500, 700, 0.05, seed=3, signed=False, sort_rows=False), F64_RTOL,
           is_sorted=False)


def test_row_block_view(spam, oracle):
    """A row block of A (what each GPU holds) gives the matching slice of y."""
    m = synth.rmat(12)
    off, idx, val = csr_np(m)
    A = spam.CsrMatrix(m.rows, m.cols, val, idx, off, True)
    x = synth.dense_vector(m.cols).numpy()
    ref = oracle.spmv((off, idx, val), x)
    lo, hi = 1000, 3001
    blk = A.device().row_block(lo, hi)
    y = blk.spmv(x)
    assert blk.invariants()
    assert np.abs(y - ref[lo:hi]).max() <= F64_RTOL * np.abs(ref).max()


def test_wrong_x_length_is_an_error(spam):
    A = spam.CsrMatrix.identity(4)
    with pytest.raises(spam.SpamError):
        A.spmv(np.ones(5))


def test_cfg2_full_size_linearity_and_third_party(spam, oracle):
    """BASELINE configs[1] at full size (10 M rows, ~160 M nnz, f64), device-resident:
    linearity A(x + 2z) == Ax + 2Az, agreement with torch.sparse (hipSPARSE, an independent third party),
    and the oracle on the whole matrix (0.1 s on the box's cores)."""
    from sparse_matrix_amd import DeviceCsr

    dev = torch.device("cuda", 0)
    A = synth.rmat_rect(10_000_000, 160_000_000, dtype=torch.float64, device=dev, oversample=1.30)
    assert 150_000_000 < A.nnz < 175_000_000
    dA = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
    x = synth.dense_vector(A.cols, seed=7, device=dev)
    z = synth.dense_vector(A.cols, seed=8, device=dev)
    yx, yz, yc = (torch.empty(A.rows, dtype=torch.float64, device=dev) for _ in range(3))
    # unprepared first (one-shot use: plain tile path, no analysis), then the prepared plan (XCD-sliced on this matrix)
    dA.spmv_dev(x.data_ptr(), yx.data_ptr())
    torch.cuda.synchronize()
    y_cold = yx.clone()
    assert not dA.spmv_plan_info()["sliced"]["active"]
    dA.spmv_prepare(x.data_ptr())
    assert dA.spmv_plan_info()["sliced"]["active"], dA.spmv_plan_info()
    dA.spmv_dev(x.data_ptr(), yx.data_ptr())
    torch.cuda.synchronize()
    nzc = y_cold != 0
    assert ((yx - y_cold).abs()[nzc] / y_cold.abs()[nzc]).max().item() <= F64_RTOL and torch.all(yx[~nzc] == 0)
    dA.spmv_dev(z.data_ptr(), yz.data_ptr())
    comb = x + 2 * z
    dA.spmv_dev(comb.data_ptr(), yc.data_ptr())
    torch.cuda.synchronize()
    ref = yx + 2 * yz
    nz = ref != 0
    assert ((yc - ref).abs()[nz] / ref.abs()[nz]).max().item() <= F64_RTOL
    assert torch.all(yc[~nz] == 0)
    S = torch.sparse_csr_tensor(A.row_ptr, A.col_idx.to(torch.int64), A.vals, size=(A.rows, A.cols))
    yt = torch.mv(S, x) if hasattr(torch, "mv") else S @ x
    assert ((yt - yx).abs()[nz] / yx.abs()[nz]).max().item() <= F64_RTOL
    del S, yt
    yo = oracle.spmv(A.cpu_numpy(), x.cpu().numpy(), nthreads=oracle.hardware_threads())
    yh = yx.cpu().numpy()
    m = yo != 0
    assert (np.abs(yh[m] - yo[m]) / np.abs(yo[m])).max() <= F64_RTOL and np.all(yh[~m] == 0)


def test_device_csr_from_sorted_coo(spam, oracle):
    """The device-side From<DokMatrix> (spam_csr/src/lib.rs:315-334): row_ptr from sorted (row, col) triples, with
    empty leading / inner / trailing rows; same SpMV as the host-built matrix; unsorted rows are rejected."""
    from sparse_matrix_amd import DeviceCsr

    dev = torch.device("cuda", 0)
    m = synth.rmat(11)
    rows_of = torch.repeat_interleave(torch.arange(m.rows, dtype=torch.int32), (m.row_ptr[1:] - m.row_ptr[:-1]))
    d = DeviceCsr.from_sorted_coo_torch(m.rows, m.cols, rows_of.to(dev), m.col_idx.to(dev), m.vals.to(dev))
    assert d.invariants()
    assert np.array_equal(d.download_row_ptr(), m.row_ptr.numpy().astype(np.uint64))
    x = synth.dense_vector(m.cols).numpy()
    ref = oracle.spmv(csr_np(m), x)
    assert np.abs(d.spmv(x) - ref).max() <= F64_RTOL * np.abs(ref).max()
    bad = rows_of.clone()
    bad[5], bad[6] = 9, 3
    with pytest.raises(spam.SpamError):
        DeviceCsr.from_sorted_coo_torch(m.rows, m.cols, bad.to(dev), m.col_idx.to(dev), m.vals.to(dev))


def test_cfg5_full_size_f32_powerlaw(spam, oracle):
    """BASELINE configs[4] at full size: 2 M rows, 2000 rows of 50 000 entries among rows of 8 (115.98 M nnz), f32,
    device-resident, against the f64-accumulated oracle at 1e-5 relative."""
    from sparse_matrix_amd import DeviceCsr

    dev = torch.device("cuda", 0)
    A = synth.powerlaw(dtype=torch.float32, device=dev)
    assert A.rows == 2_000_000 and A.nnz == 2000 * 50_000 + 1_998_000 * 8
    dA = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
    x = synth.dense_vector(A.cols, dtype=torch.float32, device=dev)
    y = torch.empty(A.rows, dtype=torch.float32, device=dev)
    dA.spmv_prepare(x.data_ptr())
    dA.spmv_dev(x.data_ptr(), y.data_ptr())
    torch.cuda.synchronize()
    off, idx, val = A.cpu_numpy()
    ref = oracle.spmv((off, idx, val.astype(np.float64)), x.cpu().numpy().astype(np.float64), nthreads=oracle.hardware_threads())
    yh = y.cpu().numpy().astype(np.float64)
    assert (np.abs(yh - ref) / np.abs(ref)).max() <= F32_RTOL


def test_randomized_shapes(spam, oracle):
    """80 random matrices: rows 1..30 000, cols 1..50 000, uniform / skewed row lengths incl. rows of tens of
    thousands of entries and empty rows, sorted or shuffled rows, f64 / f32 — against the oracle."""
    rng = np.random.default_rng(7)
    for case in range(80):
        rows, cols = int(rng.integers(1, 30_001)), int(rng.integers(1, 50_001))
        kind = case % 4
        if kind == 0:
            lens = rng.integers(0, 40, rows)
        elif kind == 1:
            lens = (rng.pareto(1.2, rows) * 3).astype(np.int64)
        elif kind == 2:
            lens = np.where(rng.random(rows) < 0.002, rng.integers(5000, 40_000, rows), rng.integers(0, 6, rows))
        else:
            lens = np.zeros(rows, dtype=np.int64)
            lens[rng.integers(0, rows, max(1, rows // 50))] = rng.integers(1, 3000)
        lens = np.minimum(lens, cols)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        idx = np.concatenate([rng.choice(cols, int(k), replace=False) for k in lens if k] or [np.empty(0, dtype=np.int64)])
        is_sorted = bool(case % 3)
        if is_sorted and idx.size:
            o = off.astype(np.int64)
            row_of = np.repeat(np.arange(rows), np.diff(o))
            idx = idx[np.lexsort((idx, row_of))]
        dtype = np.float32 if case % 5 == 3 else np.float64
        val = rng.uniform(0.5, 1.5, idx.size).astype(dtype)
        x = rng.uniform(0.5, 1.5, cols).astype(dtype)
        A = spam.CsrMatrix(rows, cols, val, idx.astype(np.uint64), off, is_sorted)
        y = A.spmv(x)
        ref = oracle.spmv((off, idx.astype(np.uint64), val.astype(np.float64)), x.astype(np.float64))
        nz = ref != 0
        assert np.all(y[~nz] == 0), f"case {case}"
        rtol = F64_RTOL if dtype == np.float64 else F32_RTOL
        if nz.any():
            assert (np.abs(y[nz].astype(np.float64) - ref[nz]) / np.abs(ref[nz])).max() <= rtol, f"case {case}"


def test_one_shot_is_not_an_analysis(spam, oracle):
    """The entry point a Rust shim's `spmv_hip` would call once per product must not pay the analysis: on a fresh handle
    the first spam_spmv_dev only adds the row -> tile map (one small kernel) to a plain SpMV — it stays within a few ms
    for 16 M entries — while spam_spmv_prepare (hot-column histogram, threshold bisection, XCD-sliced structure, timed
    comparison) is the expensive, explicit step."""
    import time
    from sparse_matrix_amd import DeviceCsr

    dev = torch.device("cuda", 0)
    A = synth.rmat(20, dtype=torch.float64, device=dev)
    x = synth.dense_vector(A.cols, device=dev)
    y = torch.empty(A.rows, dtype=torch.float64, device=dev)
    warm = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
    warm.spmv_dev(x.data_ptr(), y.data_ptr())  # allocator and code objects warm
    torch.cuda.synchronize()
    dA = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dA.spmv_dev(x.data_ptr(), y.data_ptr())
    torch.cuda.synchronize()
    t_cold = time.perf_counter() - t0
    t0 = time.perf_counter()
    dA.spmv_dev(x.data_ptr(), y.data_ptr())
    torch.cuda.synchronize()
    t_warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    dA.spmv_prepare(x.data_ptr())
    torch.cuda.synchronize()
    t_prep = time.perf_counter() - t0
    print(f"first SpMV {t_cold * 1e3:.2f} ms, second {t_warm * 1e3:.2f} ms, prepare {t_prep * 1e3:.1f} ms")
    assert t_cold < 0.02 and t_cold < 20 * max(t_warm, 2e-4)
    ref = oracle.spmv(A.cpu_numpy(), x.cpu().numpy(), nthreads=oracle.hardware_threads())
    yh = y.cpu().numpy()
    m = ref != 0
    assert (np.abs(yh[m] - ref[m]) / np.abs(ref[m])).max() <= F64_RTOL


def test_values_changed_refreshes_the_plan(spam, oracle, monkeypatch):
    """Wrapped device arrays may be updated in place (torch tensors, RCCL receive buffers).  The XCD-sliced plan holds
    copies of the values of the long rows and of the short-row residual: after scaling vals, spam_csr_values_changed
    gathers them again and y follows; the plain path reads the values where they are."""
    from sparse_matrix_amd import DeviceCsr

    dev = torch.device("cuda", 0)
    for forced in ("2", "0"):
        monkeypatch.setenv("SPAM_SPMV_SLICED", forced)
        monkeypatch.setenv("SPAM_SPMV_SLICE_KB", "4")
        A = synth.rmat(14, dtype=torch.float64, device=dev)
        dA = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
        dA.spmv_prepare()
        assert dA.spmv_plan_info()["sliced"]["active"] == (forced == "2")
        x = synth.dense_vector(A.cols, device=dev)
        y = torch.empty(A.rows, dtype=torch.float64, device=dev)
        dA.spmv_dev(x.data_ptr(), y.data_ptr())
        torch.cuda.synchronize()
        y1 = y.clone()
        A.vals.mul_(3.0)
        A.vals[::7] += 0.25
        dA.values_changed()
        dA.spmv_dev(x.data_ptr(), y.data_ptr())
        torch.cuda.synchronize()
        ref = oracle.spmv(A.cpu_numpy(), x.cpu().numpy())
        yh = y.cpu().numpy()
        m = ref != 0
        assert (np.abs(yh[m] - ref[m]) / np.abs(ref[m])).max() <= F64_RTOL
        assert not torch.equal(y, y1)
