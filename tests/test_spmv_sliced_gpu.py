"""The XCD-sliced SpMV path (spam_spmv.hip, second half) against the CPU oracle, through the C ABI.

The path is chosen by the analysis only for vectors larger than one XCD's L2; here it is forced
(SPAM_SPMV_SLICED=2) on small matrices with table budgets of a few KB, so that table columns, LDS-level columns,
cold columns, long / residual rows, slab tiles with carries and the work stealing are all exercised.  The
full-size cases (cfg2, cfg5) in test_spmv_gpu.py run whatever the analysis picks on the box.
Tolerances as in test_spmv_gpu.py: 1e-10 relative f64, 1e-5 f32 against the f64-accumulated oracle.
"""
import os

import numpy as np
import pytest
import torch

from sparse_matrix_amd import synth
from helpers import csr_np

pytestmark = pytest.mark.gpu

F64_RTOL = 1e-10
F32_RTOL = 1e-5

ENV_KEYS = ("SPAM_SPMV_SLICED", "SPAM_SPMV_SLICE_KB", "SPAM_SPMV_LDS_KB", "SPAM_SPMV_LONG", "SPAM_SPMV_SLICED_TPB",
            "SPAM_SPMV_SLICED_NPT", "SPAM_SPMV_SLICED_WGS", "SPAM_SPMV_L0_KB", "SPAM_SPMV_PIPE", "SPAM_SPMV_RG",
            "SPAM_SPMV_RG_EBLK", "SPAM_SPMV_RG_NT", "SPAM_SPMV_RG_K", "SPAM_SPMV_RG_ABITS", "SPAM_SPMV_RG_SHORT", "SPAM_SPMV_RG_ORDER",
            "SPAM_SPMV_RG_FMT")


@pytest.fixture(params=[1, 0], ids=["rowgroups", "tiles"])
def sliced_env(request):
    """Every test runs with the long rows in row-group form (LDS accumulators, the default) and in tile form."""
    saved = {k: os.environ.get(k) for k in ENV_KEYS}

    def set_env(**kw):
        for k in ENV_KEYS:
            os.environ.pop(k, None)
        os.environ["SPAM_SPMV_SLICED"] = "2"
        os.environ["SPAM_SPMV_RG"] = str(request.param)
        for k, v in kw.items():
            os.environ["SPAM_SPMV_" + k] = str(v)

    yield set_env
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _check(spam, oracle, off, idx, val, rows, cols, is_sorted, rtol, expect_active=True):
    x = np.random.default_rng(11).uniform(0.5, 1.5, cols).astype(val.dtype)
    A = spam.CsrMatrix(rows, cols, val, idx, off, is_sorted)
    y = A.spmv(x, prepare=True)  # the knobs in the environment are read by spam_spmv_prepare
    info = A.device().spmv_plan_info()["sliced"]
    if expect_active:
        assert info["active"], info
    ref = oracle.spmv((off, idx, val.astype(np.float64)), x.astype(np.float64))
    nz = ref != 0
    assert np.array_equal(y[~nz], np.zeros((~nz).sum(), dtype=y.dtype)), "rows without entries must give exactly 0"
    if nz.any():
        rel = np.abs(y[nz].astype(np.float64) - ref[nz]) / np.abs(ref[nz])
        assert rel.max() <= rtol, f"max rel err {rel.max():.3e}"
    y2 = A.spmv(x)
    if info["row_groups"]["active"] and val.dtype.kind == "f":
        # LDS atomics add a row's products in an order that varies: reproducible to rounding, not bit for bit
        tol = 1e-13 if val.dtype == np.float64 else 1e-6
        assert np.all(np.abs(y.astype(np.float64) - y2.astype(np.float64)) <= tol * np.abs(ref)), "repeat differs beyond rounding"
        assert np.array_equal(y == 0, y2 == 0)
    else:
        assert np.array_equal(y, y2), "the tile form is deterministic (fixed tile and slab order)"
    if os.environ.get("SPAM_SPMV_RG") == "1" and info["active"]:
        assert info["row_groups"]["active"] and info["row_groups"]["blocks"] > 0, info
    return info


@pytest.mark.parametrize("cfg", [dict(SLICE_KB=4), dict(SLICE_KB=4, LDS_KB=1), dict(SLICE_KB=16, LDS_KB=4, LONG=1),
                                 dict(SLICE_KB=2, LONG=40), dict(SLICE_KB=8, LDS_KB=8, SLICED_TPB=1024),
                                 dict(SLICE_KB=8, SLICED_TPB=512), dict(SLICE_KB=4, SLICED_WGS=1),
                                 dict(SLICE_KB=4, LDS_KB=2, SLICED_NPT=8), dict(SLICE_KB=4, L0_KB=0), dict(SLICE_KB=4, L0_KB=1), dict(SLICE_KB=4, PIPE=0),
                                 dict(SLICE_KB=4, LDS_KB=1, PIPE=0, SLICED_TPB=512)])
@pytest.mark.parametrize("scale", [9, 14])
def test_rmat_f64_forced(spam, oracle, sliced_env, cfg, scale):
    sliced_env(**cfg)
    m = synth.rmat(scale)
    off, idx, val = csr_np(m)
    info = _check(spam, oracle, off, idx, val, m.rows, m.cols, True, F64_RTOL)
    assert info["long_rows"] > 0 and info["table_slots_per_xcd"] > 0
    if cfg.get("LDS_KB"):
        assert info["lds_slots"] > 0


def test_rmat_rect_cold_columns(spam, oracle, sliced_env):
    """300k columns against 8 x 512 table slots: most columns are cold (gathered from x itself)."""
    sliced_env(SLICE_KB=4, LDS_KB=1)
    m = synth.rmat_rect(300_000, 4_800_000)
    off, idx, val = csr_np(m)
    info = _check(spam, oracle, off, idx, val, m.rows, m.cols, True, F64_RTOL)
    assert info["cover_table"] < 0.9


@pytest.mark.parametrize("dtype,rtol", [(torch.float32, F32_RTOL), (torch.float64, F64_RTOL)])
def test_powerlaw_long_rows_span_slab_tiles(spam, oracle, sliced_env, dtype, rtol):
    """Rows of 30 000 entries: ~3750 per slab, so slab tiles carry row pieces into the fix-up."""
    sliced_env(SLICE_KB=64, LDS_KB=8)
    m = synth.powerlaw(40_000, heavy_every=1000, heavy_nnz=30_000, light_nnz=8, dtype=dtype)
    off, idx, val = csr_np(m)
    info = _check(spam, oracle, off, idx, val, m.rows, m.cols, True, rtol)
    assert info["cover_table"] > 0.99  # 40 000 columns fit the eight tables


def test_banded_and_identity(spam, oracle, sliced_env):
    sliced_env(SLICE_KB=4, LONG=3)
    m = synth.banded(10000, 5)
    off, idx, val = csr_np(m)
    _check(spam, oracle, off, idx, val, m.rows, m.cols, True, F64_RTOL)
    n = 5000
    off = np.arange(n + 1, dtype=np.uint64)
    _check(spam, oracle, off, np.arange(n, dtype=np.uint64), np.full(n, 1.25), n, n, True, F64_RTOL, expect_active=False)


def test_all_rows_long_and_no_rows_long(spam, oracle, sliced_env):
    m = synth.rmat(10)
    off, idx, val = csr_np(m)
    sliced_env(SLICE_KB=4, LONG=1)       # every non-empty row goes to the slabs, the residual is empty
    info = _check(spam, oracle, off, idx, val, m.rows, m.cols, True, F64_RTOL)
    assert info["nnz_long"] == m.nnz
    sliced_env(SLICE_KB=4, LONG=1 << 30)  # no row qualifies: the plain path runs
    info = _check(spam, oracle, off, idx, val, m.rows, m.cols, True, F64_RTOL, expect_active=False)
    assert not info["active"]


def test_randomized_shapes_forced(spam, oracle, sliced_env):
    """60 random matrices (uniform / skewed / spiky row lengths, empty rows, shuffled rows, f64 / f32) with
    varying table, LDS and long-row settings."""
    rng = np.random.default_rng(23)
    sliced_env()
    n_cases = 60 if os.environ["SPAM_SPMV_RG"] == "1" else 24  # (the tile form: a third of the shapes, for the suite's run time)
    for case in range(n_cases):
        sliced_env(SLICE_KB=int(rng.choice([1, 2, 8, 32])), LDS_KB=int(rng.choice([0, 0, 1, 4])),
                   LONG=int(rng.choice([1, 2, 5, 17])), SLICED_TPB=int(rng.choice([256, 512, 1024])),
                   SLICED_NPT=int(rng.choice([8, 16])), PIPE=int(rng.choice([0, 1, 1])))
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
        if lens.sum() == 0:
            continue
        # skewed column choice so that table, LDS-level and cold columns all occur
        w = rng.pareto(1.0, cols) + 1e-3
        w /= w.sum()
        idx = np.concatenate([rng.choice(cols, int(k), replace=False, p=w if k * 20 < cols else None) for k in lens if k])
        is_sorted = bool(case % 3)
        if is_sorted:
            row_of = np.repeat(np.arange(rows), np.diff(off.astype(np.int64)))
            idx = idx[np.lexsort((idx, row_of))]
        dtype = np.float32 if case % 5 == 3 else np.float64
        val = rng.uniform(0.5, 1.5, idx.size).astype(dtype)
        _check(spam, oracle, off, idx.astype(np.uint64), val, rows, cols, is_sorted,
               F64_RTOL if dtype == np.float64 else F32_RTOL, expect_active=False)


def test_auto_mode_keeps_small_matrices_on_the_plain_path(spam, oracle, sliced_env):
    sliced_env()
    os.environ["SPAM_SPMV_SLICED"] = "1"
    m = synth.rmat(12)
    off, idx, val = csr_np(m)
    info = _check(spam, oracle, off, idx, val, m.rows, m.cols, True, F64_RTOL, expect_active=False)
    assert not info["active"]


def test_auto_mode_times_the_row_group_form_on_a_mid_size_matrix(spam, oracle, sliced_env):
    """x of R-MAT scale 18 (2 MB) fits any L2, and one table would cover it: the tile form is no candidate there, the
    row-group form is (sorted gathers, sums in LDS) and spam_spmv_prepare decides by timing both — whichever it keeps,
    the product is right; the block size follows the matrix (4 M entries: 16 k per block, hundreds of blocks)."""
    sliced_env()
    os.environ["SPAM_SPMV_SLICED"] = "1"
    m = synth.rmat(18)
    off, idx, val = csr_np(m)
    info = _check(spam, oracle, off, idx, val, m.rows, m.cols, True, F64_RTOL, expect_active=False)
    if os.environ["SPAM_SPMV_RG"] == "1":
        assert info["ms_plain"] > 0 and info["ms_sliced"] > 0, info  # both were timed
        if info["active"]:
            assert info["row_groups"]["active"] and info["row_groups"]["blocks"] >= 150, info
    else:
        assert not info["active"], info  # the tile form gains from the tables only


@pytest.mark.parametrize("kind", ["powerlaw_f32", "tall", "giant_row_i64", "banded"])
def test_auto_mode_mid_size_shapes(spam, oracle, sliced_env, kind):
    """Matrices of 3-6 M entries in AUTO mode (no knob set): whatever spam_spmv_prepare keeps after timing its candidates —
    block size by the matrix, giant rows in pieces, the plain path for the banded one — the product agrees with the oracle."""
    sliced_env()
    os.environ["SPAM_SPMV_SLICED"] = "1"
    if kind == "powerlaw_f32":
        m = synth.powerlaw(300_000, heavy_every=500, heavy_nnz=4000, light_nnz=6, dtype=torch.float32)
        off, idx, val = csr_np(m)
        info = _check(spam, oracle, off, idx, val, m.rows, m.cols, True, F32_RTOL, expect_active=False)
    elif kind == "tall":
        m = synth.random_csr(1_500_000, 200_000, 3.2e6 / (1_500_000 * 200_000), seed=3, signed=False)
        off, idx, val = csr_np(m)
        info = _check(spam, oracle, off, idx, val, m.rows, m.cols, True, F64_RTOL, expect_active=False)
    elif kind == "giant_row_i64":
        m = synth.powerlaw(400_000, heavy_every=200_000, heavy_nnz=400_000, light_nnz=7, dtype=torch.float64)
        off, idx, _ = csr_np(m)
        rng = np.random.default_rng(2)
        val = rng.integers(-2**60, 2**60, idx.size, dtype=np.int64)
        x = rng.integers(-2**60, 2**60, m.cols, dtype=np.int64)
        A = spam.CsrMatrix(m.rows, m.cols, val, idx, off, True)
        ref = oracle.spmv((off, idx, val), x)
        for k in range(3):
            assert np.array_equal(A.spmv(x, prepare=(k == 0)), ref), k
        info = A.device().spmv_plan_info()["sliced"]
    else:
        m = synth.banded(600_000, 7)
        off, idx, val = csr_np(m)
        info = _check(spam, oracle, off, idx, val, m.rows, m.cols, True, F64_RTOL, expect_active=False)
    if os.environ["SPAM_SPMV_RG"] == "1":
        assert info["ms_plain"] > 0, info  # the plain path was timed: from 3 M entries on the row-group form is a candidate
    elif m.cols * val.itemsize <= (4 << 20):
        # the tile form is a candidate only when x is larger than one L2: spam_spmv_prepare then skips the column
        # statistics and the timing of the plain path altogether (nothing could use them)
        assert info["ms_plain"] == 0 and not info["active"], info


def test_row_block_view_forced(spam, oracle, sliced_env):
    """A zero-copy row-block view (what a rank holds of a row-sharded matrix: its row_ptr starts inside the parent's
    arrays, a few pad entries in front) through the sliced path."""
    sliced_env(SLICE_KB=4, LDS_KB=1)
    m = synth.rmat(12)
    off, idx, val = csr_np(m)
    A = spam.CsrMatrix(m.rows, m.cols, val, idx, off, True)
    x = synth.dense_vector(m.cols).numpy()
    ref = oracle.spmv((off, idx, val), x)
    for lo, hi in ((1000, 3001), (0, 17), (4090, 4096)):
        blk = A.device().row_block(lo, hi)
        blk.spmv_prepare()
        assert blk.spmv_plan_info()["sliced"]["active"] or hi - lo < 100  # (a block without long rows stays plain)
        y = blk.spmv(x)
        assert np.abs(y - ref[lo:hi]).max() <= F64_RTOL * np.abs(ref).max()
        assert np.array_equal(y == 0, ref[lo:hi] == 0)
    assert A.device().row_block(1000, 3001).spmv(x).shape == (2001,)


# ---- the row-group form (spam_spmv_rg.hip) -------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [dict(RG_EBLK=1024), dict(RG_EBLK=1024, RG_ABITS=9), dict(RG_ABITS=10, RG_NT=1024, RG_K=2),
                                 dict(RG_EBLK=4096, RG_NT=256), dict(RG_EBLK=2048, RG_NT=1024, RG_K=4, LONG=1),
                                 dict(RG_EBLK=1024, RG_SHORT=0, LONG=9), dict(RG_ABITS=9, LONG=40), dict(RG_EBLK=1024, LONG=7),
                                 dict(RG_ORDER=1000, RG_EBLK=4096), dict(RG_ORDER=1), dict(RG_ORDER=4, LONG=3, RG_EBLK=1024),
                                 dict(LONG=1), dict(LONG=5, RG_EBLK=2048)])
@pytest.mark.parametrize("kind", ["rmat14", "powerlaw"])
def test_row_groups_geometries(spam, oracle, sliced_env, cfg, kind):
    """Small blocks (many groups, rows with several accumulators each), narrow accumulator fields (511 / 1023 pairs per
    block), every kernel geometry; the hub rows of the power-law matrix take the folded sub-accumulators; short rows
    as blocks of the same kernel (LONG > 1) or in their own kernel (RG_SHORT=0); all ranks in row order / by length."""
    sliced_env(SLICE_KB=8, **cfg)
    if os.environ["SPAM_SPMV_RG"] != "1":
        pytest.skip("row-group knobs")
    m = synth.rmat(14) if kind == "rmat14" else synth.powerlaw(30_000, heavy_every=700, heavy_nnz=20_000, light_nnz=6,
                                                               dtype=torch.float64)
    off, idx, val = csr_np(m)
    info = _check(spam, oracle, off, idx, val, m.rows, m.cols, True, F64_RTOL)
    g = info["row_groups"]
    assert g["active"] and g["groups"] >= 1 and g["accumulators"] >= info["long_rows"]
    if "RG_EBLK" in cfg and kind == "powerlaw":
        assert g["accumulators"] > info["long_rows"], "rows of 20 000 entries take several accumulators"


def test_row_groups_i64_exact_and_values_changed(spam, oracle, sliced_env):
    """Wrapping<i64>: the LDS atomics are integer adds, so the row-group form equals the oracle bit for bit; an
    in-place update of the values reaches the plan's copy through spam_csr_values_changed."""
    sliced_env(SLICE_KB=8, RG_EBLK=1024)
    m = synth.rmat(13)
    off, idx, _ = csr_np(m)
    rng = np.random.default_rng(5)
    val = rng.integers(-2**62, 2**62, idx.size, dtype=np.int64)
    x = rng.integers(-2**62, 2**62, m.cols, dtype=np.int64)
    A = spam.CsrMatrix(m.rows, m.cols, val, idx, off, True)
    y = A.spmv(x, prepare=True)
    ref = oracle.spmv((off, idx, val), x)
    assert np.array_equal(y, ref)
    d = A.device()
    assert d.spmv_plan_info()["sliced"]["active"]
    dev = torch.device("cuda", 0)
    tv = torch.from_numpy(val).to(dev)
    B = spam.DeviceCsr.from_torch(m.rows, m.cols, torch.from_numpy(off.astype(np.int64)).to(dev),
                                  torch.from_numpy(idx.astype(np.int32)).to(dev), tv, True)
    tx = torch.from_numpy(x).to(dev)
    ty = torch.empty(m.rows, dtype=torch.int64, device=dev)
    B.spmv_prepare(tx.data_ptr())
    B.spmv_dev(tx.data_ptr(), ty.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(ty.cpu().numpy(), ref)
    tv.mul_(3)
    B.values_changed()
    B.spmv_dev(tx.data_ptr(), ty.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(ty.cpu().numpy(), oracle.spmv((off, idx, val * 3), x))
    tv.mul_(7)  # (a second change: the value-source map built by the first one is used)
    B.values_changed()
    B.spmv_dev(tx.data_ptr(), ty.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(ty.cpu().numpy(), oracle.spmv((off, idx, val * 21), x))


@pytest.mark.parametrize("prepared", [True, False], ids=["prepared", "plain"])
def test_one_handle_from_several_host_threads(spam, oracle, sliced_env, prepared):
    """spam_hip.h: a handle may serve SpMVs from several host threads at once.  One SpMV is several launches over the
    plan's scratch (tables, queue heads, pair sums, carries): the library enqueues them as a unit.  Four threads, each
    with its own x and y on one handle (ctypes releases the GIL during the calls); i64, so any mix-up of two threads'
    tables or sums shows as a wrong integer."""
    import threading
    sliced_env(SLICE_KB=8, RG_EBLK=1024)
    m = synth.rmat(12)
    off, idx, _ = csr_np(m)
    rng = np.random.default_rng(17)
    val = rng.integers(-2**40, 2**40, idx.size, dtype=np.int64)
    dev = torch.device("cuda", 0)
    B = spam.DeviceCsr.from_torch(m.rows, m.cols, torch.from_numpy(off.astype(np.int64)).to(dev),
                                  torch.from_numpy(idx.astype(np.int32)).to(dev), torch.from_numpy(val).to(dev), True)
    nthreads, reps = 4, 25
    xs = [rng.integers(-2**20, 2**20, m.cols, dtype=np.int64) for _ in range(nthreads)]
    refs = [oracle.spmv((off, idx, val), x) for x in xs]
    txs = [torch.from_numpy(x).to(dev) for x in xs]
    tys = [torch.zeros(reps, m.rows, dtype=torch.int64, device=dev) for _ in range(nthreads)]
    if prepared:
        B.spmv_prepare(txs[0].data_ptr())
        assert B.spmv_plan_info()["sliced"]["active"]
    else:
        B.spmv_dev(txs[0].data_ptr(), tys[0][0].data_ptr())  # (the lazily built plain plan)
    torch.cuda.synchronize()
    errors = []
    start = threading.Barrier(nthreads)

    def work(t):
        try:
            start.wait()
            for r in range(reps):
                B.spmv_dev(txs[t].data_ptr(), tys[t][r].data_ptr())
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for t in range(nthreads):
        got = tys[t].cpu().numpy()
        for r in range(reps):
            assert np.array_equal(got[r], refs[t]), f"thread {t} repeat {r}"


@pytest.mark.parametrize("prepared", [True, False], ids=["prepared", "plain"])
def test_spmv_is_legal_under_stream_capture(spam, oracle, sliced_env, prepared):
    """After spam_spmv_prepare (or a first SpMV) a product is kernel launches on the library's stream and nothing else —
    no allocation, no synchronisation, no host read-back — so it can be recorded into a HIP graph and replayed; x and y
    are read / written in place at replay."""
    sliced_env(SLICE_KB=8, RG_EBLK=1024)
    from sparse_matrix_amd import _lib
    m = synth.rmat(12)
    off, idx, val = csr_np(m)
    dev = torch.device("cuda", 0)
    B = spam.DeviceCsr.from_torch(m.rows, m.cols, torch.from_numpy(off.astype(np.int64)).to(dev),
                                  torch.from_numpy(idx.astype(np.int32)).to(dev), torch.from_numpy(val).to(dev), True)
    rng = np.random.default_rng(23)
    x = torch.from_numpy(rng.uniform(0.5, 1.5, m.cols)).to(dev)
    y = torch.zeros(m.rows, dtype=torch.float64, device=dev)
    L = _lib.load()
    side = torch.cuda.Stream()
    try:
        with torch.cuda.stream(side):
            _lib.check(L.spam_hip_set_stream(side.cuda_stream))
            if prepared:
                B.spmv_prepare(x.data_ptr())
                assert B.spmv_plan_info()["sliced"]["active"]
            B.spmv_dev(x.data_ptr(), y.data_ptr())  # (plain: builds the lazily created tile map outside the capture)
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                B.spmv_dev(x.data_ptr(), y.data_ptr())
                B.spmv_dev(x.data_ptr(), y.data_ptr())
        torch.cuda.synchronize()
        for rep in range(3):
            xn = rng.uniform(0.5, 1.5, m.cols)
            x.copy_(torch.from_numpy(xn).to(dev))
            y.zero_()
            g.replay()
            torch.cuda.synchronize()
            ref1 = oracle.spmv((off, idx, val), xn)
            # (the graph holds the product twice: the second run overwrites y with the same result)
            assert np.allclose(y.cpu().numpy(), ref1, rtol=1e-12, atol=0), f"replay {rep}"
    finally:
        _lib.check(L.spam_hip_set_stream(torch.cuda.current_stream().cuda_stream))


@pytest.mark.parametrize("dtype", [np.int64, np.float64, np.float32])
def test_row_groups_giant_rows_run_as_pieces(spam, oracle, sliced_env, dtype):
    """A row longer than a whole group's share of entries sits alone in its group: its blocks are listed in the kernel's
    queue as several pieces, whose sums meet in the block's pairs by device-scope atomics; the reduce sets those pairs
    back to zero, so the second and third product of a handle give the same result as the first (i64: bit for bit)."""
    sliced_env(SLICE_KB=8, RG_EBLK=1024)
    # 1024-entry blocks: a group takes 9216 entries, a row of 30 000 sits alone and its nine blocks (13 chunks each) run in
    # four pieces; no row has fewer than three entries, so every queue entry beyond 9 per group is a piece
    m = synth.powerlaw(30000, heavy_every=10000, heavy_nnz=30000, light_nnz=5, dtype=torch.float64)
    off, idx, val = csr_np(m)
    rng = np.random.default_rng(31)
    if dtype == np.int64:
        val = rng.integers(-2**50, 2**50, idx.size, dtype=np.int64)
        x = rng.integers(-2**30, 2**30, m.cols, dtype=np.int64)
    else:
        val = val.astype(dtype)
        x = rng.uniform(0.5, 1.5, m.cols).astype(dtype)
    A = spam.CsrMatrix(m.rows, m.cols, val, idx, off, True)
    ref = oracle.spmv((off, idx, val if dtype == np.int64 else val.astype(np.float64)), x if dtype == np.int64 else x.astype(np.float64))
    ys = [A.spmv(x, prepare=(k == 0)) for k in range(3)]
    info = A.device().spmv_plan_info()["sliced"]
    if os.environ["SPAM_SPMV_RG"] == "1":
        g = info["row_groups"]
        assert g["active"] and g["blocks"] > 9 * g["groups"], ("pieces make more queue entries than (part, group) blocks", g)
    for y in ys:
        if dtype == np.int64:
            assert np.array_equal(y, ref)
        else:
            assert np.all(np.abs(y.astype(np.float64) - ref) <= (1e-12 if dtype == np.float64 else 2e-6) * np.abs(ref))


@pytest.mark.parametrize("fmt", [0, 1, 2])
@pytest.mark.parametrize("kind", ["dense_blocks", "gappy", "rmat14"])
def test_row_groups_packed_entry_words(spam, oracle, sliced_env, fmt, kind):
    """The packed chunk formats of the row-group form (16- / 24-bit entry words: accumulator + slot DELTA, slots rebuilt by
    wave prefix sums; padding entries bridge deltas that do not fit), each forced as the upper limit: f64 against the
    oracle, Wrapping<i64> bit for bit, and an in-place value update through spam_csr_values_changed (the padded value
    positions keep their sources).  dense_blocks: 40 000 rows over 3 000 columns — every block walks its table with
    deltas of 0..2 (16-bit words); gappy: the same rows plus columns spread over 200 000 (deltas of hundreds: 24-bit
    words, or padding under the 16-bit limit); rmat14: a mix, with hub rows and pieces."""
    sliced_env(SLICE_KB=8, RG_FMT=fmt, RG_EBLK=4096)
    if os.environ["SPAM_SPMV_RG"] != "1":
        pytest.skip("row-group format")
    rng = np.random.default_rng(17 + fmt)
    if kind == "rmat14":
        m = synth.rmat(14)
        off, idx, val = csr_np(m)
        rows, cols = m.rows, m.cols
    else:
        rows, cols = 40_000, (3_000 if kind == "dense_blocks" else 200_000)
        per = rng.integers(3, 24, rows)
        off = np.concatenate([[0], np.cumsum(per)]).astype(np.uint64)
        pieces = []
        for r in range(rows):
            hot = rng.choice(3_000, size=min(int(per[r]), 3_000), replace=False)
            if kind == "gappy" and r % 3 == 0:
                hot[: len(hot) // 2] = rng.choice(np.arange(3_000, cols), size=len(hot) // 2, replace=False)
            pieces.append(np.sort(np.unique(hot)))
        per = np.array([len(p) for p in pieces])
        off = np.concatenate([[0], np.cumsum(per)]).astype(np.uint64)
        idx = np.concatenate(pieces).astype(np.uint64)
        val = rng.uniform(0.5, 1.5, idx.size)
    info = _check(spam, oracle, off, idx, val, rows, cols, True, F64_RTOL)
    assert info["row_groups"]["active"]
    # Wrapping<i64>: exact, and again after the values changed in place
    vi = rng.integers(-2**62, 2**62, idx.size, dtype=np.int64)
    xi = rng.integers(-2**62, 2**62, cols, dtype=np.int64)
    dev = torch.device("cuda", 0)
    tv = torch.from_numpy(vi).to(dev)
    B = spam.DeviceCsr.from_torch(rows, cols, torch.from_numpy(off.astype(np.int64)).to(dev),
                                  torch.from_numpy(idx.astype(np.int32)).to(dev), tv, True)
    tx = torch.from_numpy(xi).to(dev)
    ty = torch.empty(rows, dtype=torch.int64, device=dev)
    B.spmv_prepare(tx.data_ptr())
    assert B.spmv_plan_info()["sliced"]["row_groups"]["active"]
    B.spmv_dev(tx.data_ptr(), ty.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(ty.cpu().numpy(), oracle.spmv((off, idx, vi), xi))
    sb0 = B.spmv_plan_info()["sliced"]["row_groups"]["stream_bytes"]
    tv.mul_(5)
    B.values_changed()
    B.spmv_dev(tx.data_ptr(), ty.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(ty.cpu().numpy(), oracle.spmv((off, idx, vi * 5), xi))
    # the first change rebuilt the plan with its value-source map (spam_spmv_prepare does not build it); a second change
    # takes the short way: every padded value position is re-gathered from its recorded source
    tv.mul_(-3)
    B.values_changed()
    B.spmv_dev(tx.data_ptr(), ty.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(ty.cpu().numpy(), oracle.spmv((off, idx, vi * (-15)), xi))
    # the packed formats must actually shrink the stream where the deltas are small
    if kind == "dense_blocks":
        os.environ["SPAM_SPMV_RG_FMT"] = "0"
        B0 = spam.DeviceCsr.from_torch(rows, cols, torch.from_numpy(off.astype(np.int64)).to(dev),
                                       torch.from_numpy(idx.astype(np.int32)).to(dev), tv, True)
        B0.spmv_prepare(tx.data_ptr())
        plain = B0.spmv_plan_info()["sliced"]["row_groups"]["stream_bytes"]
        if fmt == 2:
            assert sb0 < 0.90 * plain, (sb0, plain)
        elif fmt == 1:
            assert sb0 < 0.96 * plain, (sb0, plain)
        else:
            assert sb0 == plain
