"""SpGEMM parity on the MI355X: libspam_hip.so (through the C ABI) against the CPU oracle of
CsrMatrix::mul_hash (spam_csr/src/mul_hash.rs:13-201).

Bar (BASELINE.json north_star): row_ptr bit-exact; col_idx bit-exact with sort_rows (B2 = true) and
per-row set-equal without (the reference's own test compares order-insensitively through DokMatrix,
spam_csr/src/tests.rs:356-371); f64 values within 1e-10 relative on positive data, Higham's bound
(spam_dok/src/lib.rs:56-92) on signed data.
"""
import os

import numpy as np
import pytest
import torch

from sparse_matrix_amd import synth
from helpers import assert_csr_parity, csr_np, rows_as_sets

pytestmark = pytest.mark.gpu

F64_RTOL = 1e-10
F32_RTOL = 1e-5


def _mk(spam, m, is_sorted=True):
    off, idx, val = csr_np(m)
    return spam.CsrMatrix(m.rows, m.cols, val, idx, off, is_sorted), (off, idx, val)


def _both(spam, oracle, ma, mb, rtol, a_sorted=True, b_sorted=True):
    A, a = _mk(spam, ma, a_sorted)
    B, b = _mk(spam, mb, b_sorted)
    for sort_rows in (True, False):
        C = A.mul_hash(B, sort_output=sort_rows)
        want = oracle.mul_hash(a, b, sort_rows=sort_rows)
        assert C.rows() == ma.rows and C.cols() == mb.cols
        assert C.invariants()
        assert_csr_parity((C.offsets, C.indices, C.vals), want, sort_rows, rtol)
    return C


def test_banded_cfg1(spam, oracle):
    """BASELINE configs[0]: 10k x 10k banded, C = A*A: F = 249 950, nnz(C) = 89 980."""
    m = synth.banded(10000, 5)
    C = _both(spam, oracle, m, m, F64_RTOL)
    assert C.nnz() == 89980


@pytest.mark.parametrize("scale", [6, 10, 13])
def test_rmat_square(spam, oracle, scale):
    """R-MAT A*A: scale 13 has rows in every bin incl. the dense-chunk path (nnz(C_i) > 4096)."""
    m = synth.rmat(scale)
    _both(spam, oracle, m, m, F64_RTOL)


def test_rmat_multi_chunk(spam, oracle):
    """scale 16: 65 536 columns = 4 numeric chunks; heavy rows use B pre-split at chunk boundaries."""
    m = synth.rmat(16)
    A, a = _mk(spam, m)
    dev = A.device()
    C, st = dev.spgemm(dev, sort_rows=True)
    got = C.download()
    want = oracle.mul_hash(a, a, sort_rows=True)
    assert st["rows_num_bin"][4] > 0 and st["heavy_items"] > 0
    assert st["flops_f"] == int(oracle.rows_to_threads(a[0], a[1], a[0], 1)[0].sum())
    assert_csr_parity(got, want, True, F64_RTOL)
    assert C.invariants()


def test_wide_b_many_symbolic_chunks(spam, oracle):
    """B with 300 000 columns: 2 symbolic bitmap windows and 37 numeric chunks per heavy row."""
    ma = synth.random_csr(16, 500, 0.5, seed=11, signed=False)
    mb = synth.random_csr(500, 300_000, 0.002, seed=12, signed=False)
    _both(spam, oracle, ma, mb, F64_RTOL)


def test_very_wide_b_several_bitmap_windows(spam, oracle):
    """B with 2.5 M columns: 306 numeric chunks of 8192 and 10 symbolic bitmap windows of 2^18 columns per heavy row,
    rows of C with ~450 k entries; all three output orders."""
    ma = synth.random_csr(6, 2000, 0.25, seed=41, signed=False)
    mb = synth.random_csr(2000, 2_500_000, 0.0004, seed=42, signed=False)
    C = _both(spam, oracle, ma, mb, F64_RTOL)
    assert np.diff(C.offsets.astype(np.int64)).max() > 300_000
    _exact_unsorted(spam, oracle, ma, mb)


@pytest.mark.parametrize("cols", [20_000_000, 70_000_000])
def test_very_wide_b_beyond_2_pow_24_columns(spam, oracle, cols):
    """cols(B) > 2^24 with heavy rows (round 1 refused these: the heavy path kept its chunk list in LDS).  20 M columns =
    2442 chunks of 8192; 70 M columns exceeds 8192 such chunks and switches to the 16384-column geometry; rows of C
    with ~19 k entries spread over the whole width (most groups are span-limited), all three output orders."""
    ma = synth.random_csr(5, 1000, 0.004 * 1000 / 1000, seed=51, signed=False)
    mb = synth.random_csr(1000, cols, 5000 / cols, seed=52, signed=False)
    C = _both(spam, oracle, ma, mb, F64_RTOL)
    assert np.diff(C.offsets.astype(np.int64)).max() > 4096
    _exact_unsorted(spam, oracle, ma, mb)


@pytest.mark.parametrize("sym_geom", [0, 1, 2, 3])
@pytest.mark.parametrize("heavy_tpb", [256, 512, 1024])
def test_heavy_path_geometries(spam, oracle, sym_geom, heavy_tpb):
    """Every geometry of the symbolic heavy kernel (window width / workgroup size, SPAM_SYM_GEOM) and of the numeric one
    (SPAM_HEAVY_TPB), which the library otherwise picks from the matrix width: 3 M columns = 3 to 12 bitmap windows per
    row; A rows of 4 entries (shared head only) and of ~1000 entries (owned and streamed batches)."""
    saved = {k: os.environ.get(k) for k in ("SPAM_SYM_GEOM", "SPAM_HEAVY_TPB")}
    os.environ["SPAM_SYM_GEOM"], os.environ["SPAM_HEAVY_TPB"] = str(sym_geom), str(heavy_tpb)
    try:
        ma = synth.random_csr(4, 1000, 0.004, seed=61, signed=False)
        mb = synth.random_csr(1000, 3_000_000, 5000 / 3_000_000, seed=62, signed=False)
        C = _both(spam, oracle, ma, mb, F64_RTOL)
        assert np.diff(C.offsets.astype(np.int64)).max() > 4096
        ma = synth.random_csr(3, 2000, 0.5, seed=63, signed=False)
        mb = synth.random_csr(2000, 600_000, 30 / 600_000, seed=64, signed=False)
        C = _both(spam, oracle, ma, mb, F64_RTOL)
        assert np.diff(C.offsets.astype(np.int64)).max() > 4096
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_unsorted_b_heavy_rows_filter_path(spam, oracle):
    """Unsorted B cannot be pre-split; heavy rows then filter every product by chunk."""
    ma = synth.random_csr(8, 300, 0.6, seed=21, signed=False)
    mb = synth.random_csr(300, 40_000, 0.05, seed=22, signed=False, sort_rows=False)
    _both(spam, oracle, ma, mb, F64_RTOL, b_sorted=False)


def test_f32(spam, oracle):
    m = synth.rmat(11, dtype=torch.float32)
    _both(spam, oracle, m, m, F32_RTOL)


def test_rectangular_and_empty_rows(spam, oracle):
    ma = synth.random_csr(300, 200, 0.02, seed=1, signed=False)
    mb = synth.random_csr(200, 450, 0.03, seed=2, signed=False)
    _both(spam, oracle, ma, mb, F64_RTOL)


def test_identity_is_neutral(spam):
    """I*A = A*I = A as (row, col) -> value maps (SURVEY.md §8c algebraic KAT)."""
    m = synth.rmat(9)
    A, a = _mk(spam, m)
    I = spam.CsrMatrix.identity(m.rows)
    for C in (I.mul_hash(A, True), A.mul_hash(I, True)):
        assert np.array_equal(C.offsets, a[0]) and np.array_equal(C.indices, a[1]) and np.array_equal(C.vals, a[2])


def test_cancellation_keeps_explicit_zero(spam):
    """a*b + (-a*b): mul_hash keeps the structural entry with value 0.0 (structure = symbolic pattern);
    it disappears only in the CSR -> DOK conversion (spam_dok/src/lib.rs:171-175)."""
    A = spam.CsrMatrix.from_dok((1, 2), {(0, 0): 2.0, (0, 1): -2.0})
    B = spam.CsrMatrix.from_dok((2, 2), {(0, 0): 3.0, (1, 0): 3.0, (1, 1): 1.0})
    C = A.mul_hash(B, sort_output=True)
    assert C.indices.tolist() == [0, 1] and C.vals.tolist() == [0.0, -2.0]
    assert C.to_dok() == {(0, 1): -2.0}


def test_property_small_random_vs_dok(spam, oracle):
    """The reference's hot-path test (spam_csr/src/tests.rs:356-371) and fuzz property
    (fuzz/fuzz_targets/mul_hash.rs:28-46) on the GPU path: unsorted CSR inputs from DOK matrices of
    dimension 1..4 (and up to 40), mul_hash::<false,false>, invariants hold, CSR -> DOK equals the DOK
    triple loop within Higham's bound."""
    from oracle import dok

    rng = np.random.default_rng(1234)
    for case in range(120):
        big = case % 4 == 3
        l, m, n = (rng.integers(1, 41, 3) if big else rng.integers(1, 5, 3))
        da = {(int(rng.integers(l)), int(rng.integers(m))): float(rng.normal()) for _ in range(int(rng.integers(0, 2 * l * m + 1)))}
        db = {(int(rng.integers(m)), int(rng.integers(n))): float(rng.normal()) for _ in range(int(rng.integers(0, 2 * m * n + 1)))}
        A = spam.CsrMatrix.from_dok((l, m), da, rng=rng)
        B = spam.CsrMatrix.from_dok((m, n), db, rng=rng)
        C = A.mul_hash(B, sort_output=False)
        assert C.invariants()
        ad = dok.to_dense(l, m, A.offsets, A.indices, A.vals)
        bd = dok.to_dense(m, n, B.offsets, B.indices, B.vals)
        cd = dok.to_dense(l, n, C.offsets, C.indices, C.vals)
        assert dok.is_good_approx_of_mul(cd, ad, bd)
        # structure: exactly the symbolic pattern of the oracle
        want = oracle.mul_hash((A.offsets, A.indices, A.vals), (B.offsets, B.indices, B.vals), sort_rows=False)
        assert np.array_equal(C.offsets, want[0])
        assert rows_as_sets(C.offsets, C.indices) == rows_as_sets(want[0], want[1])


def test_dimension_mismatch_is_an_error(spam):
    """The reference panics on rhs.offsets[k + 1] (mul_hash.rs:46); the ABI reports SPAM_E_INVALID."""
    A = spam.CsrMatrix.identity(4)
    B = spam.CsrMatrix.identity(5)
    with pytest.raises(spam.SpamError) as e:
        A.mul_hash(B)
    assert e.value.code == 1


def test_rows_to_parts_matches_oracle(spam, oracle):
    """rows_to_threads with tnum = #GPUs (mul_hash.rs:38-64) + its own post-conditions (:204-224)."""
    m = synth.rmat(12)
    A, a = _mk(spam, m)
    for parts in (1, 2, 3, 8):
        flops, offs = A.device().rows_to_parts(A.device(), parts)
        rn, ro = oracle.rows_to_threads(a[0], a[1], a[0], parts)
        assert np.array_equal(flops, rn) and np.array_equal(offs, ro)
        assert np.all(np.diff(offs.astype(np.int64)) >= 0) and offs[-1] == m.rows
        from sparse_matrix_amd import dist as spdist
        assert A.device().balanced_row_blocks(A.device(), parts, 2300) == spdist.balanced_row_blocks(rn.astype(np.int64), parts, 2300)


def test_row_block_product_equals_slice(spam, oracle):
    """C[lo:hi, :] = A[lo:hi, :] * B — what each GPU computes in the row-sharded multi-GPU run."""
    m = synth.rmat(11)
    A, a = _mk(spam, m)
    dev = A.device()
    want = oracle.mul_hash(a, a, sort_rows=True)
    _, offs = dev.rows_to_parts(dev, 3)
    for lo, hi in zip(offs[:-1].tolist(), offs[1:].tolist()):
        if hi == lo:
            continue
        C, _ = dev.row_block(lo, hi).spgemm(dev, sort_rows=True)
        o, i, v = C.download()
        wo = want[0][lo:hi + 1] - want[0][lo]
        s, e = int(want[0][lo]), int(want[0][hi])
        assert_csr_parity((o, i, v), (wo, want[1][s:e], want[2][s:e]), True, F64_RTOL)


def test_rank_step_entry_point(spam, oracle):
    """spam_spgemm_rank_step (the per-rank sequence of the row-sharded multi-GPU product as ONE C call): for every rank of
    a 3-rank and an 8-rank world the returned block is rows [row_lo, row_hi) of the full product, the ranges tile the
    rows in order and equal the weighted partition (rows_to_threads on F + 2300 per non-empty row)."""
    m = synth.rmat(11)
    A, a = _mk(spam, m)
    dev = A.device()
    want = oracle.mul_hash(a, a, sort_rows=True)
    for world in (3, 8):
        offs = dev.balanced_row_blocks(dev, world, 2300)
        nxt = 0
        for rank in range(world):
            C, st, (lo, hi) = dev.rank_step(dev, rank, world, 2300, sort_rows=True)
            assert (lo, hi) == (offs[rank], offs[rank + 1]) and lo == nxt
            nxt = hi
            if hi == lo:
                assert C is None
                continue
            o, i, v = C.download()
            s, e = int(want[0][lo]), int(want[0][hi])
            assert_csr_parity((o, i, v), (want[0][lo:hi + 1] - want[0][lo], want[1][s:e], want[2][s:e]), True, F64_RTOL)
            assert st["nnz_c"] == e - s
        assert nxt == m.rows
    with pytest.raises(spam.SpamError):
        dev.rank_step(dev, 3, 3)


def test_full_size_row_sum_property(spam, oracle):
    """BASELINE configs[2] at full size (R-MAT scale 20: F = 2.09e10 products, nnz(C) = 9.7e9 > 2^32).
    (1) a check that needs no CPU product: (A*A)*x == A*(A*x) row by row for a RANDOM positive x (a permuted or
    corrupted col_idx inside a row changes the row's dot product, unlike x = 1), sorted and unsorted output, plus
    invariants 1-7 of the sorted result on the device;
    (2) two bounded row samples of the unsorted product against the oracle — 306 heavy rows from the middle of the
    matrix and a run of lighter rows near its end: row_ptr exact, column sets equal, values 1e-10, and the same rows
    in the reference's slot order bit-exact (sort_rows = 2)."""
    import torch
    from sparse_matrix_amd import DeviceCsr
    from helpers import spgemm_rows_vs_oracle

    dev = torch.device("cuda", 0)
    A = synth.rmat(20, dtype=torch.float64, device=dev)
    dA = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
    x = synth.dense_vector(A.cols, dtype=torch.float64, device=dev)
    t1, t2, t3 = (torch.empty(A.rows, dtype=torch.float64, device=dev) for _ in range(3))
    dA.spmv_dev(x.data_ptr(), t1.data_ptr())
    dA.spmv_dev(t1.data_ptr(), t2.data_ptr())
    a_host = A.cpu_numpy()
    for sort_rows in (False, True):
        C, st = dA.spgemm(dA, sort_rows=sort_rows)
        assert st["nnz_c"] > 9_000_000_000 and st["flops_f"] > 20_000_000_000
        C.spmv_dev(x.data_ptr(), t3.data_ptr())
        torch.cuda.synchronize()
        rel = ((t3 - t2).abs() / t2.abs().clamp_min(1e-300)).max().item()
        assert rel <= F64_RTOL, rel
        if sort_rows:
            assert C.invariants()
        else:
            n = A.rows
            r1, Fs, nz = spgemm_rows_vs_oracle(oracle, dA, dA, C, a_host, a_host, n // 2, 2e8)
            assert nz > 10_000_000 and r1 - n // 2 > 100
            spgemm_rows_vs_oracle(oracle, dA, dA, C, a_host, a_host, n - n // 16, 5e7)
        rp = C.download_row_ptr().astype(np.int64)
        assert rp[0] == 0 and rp[-1] == st["nnz_c"] and np.all(np.diff(rp) >= 0)
        C.free()


def test_scale22_rank_blocks_cfg4(spam, oracle):
    """BASELINE configs[3] (8 x MI355X, R-MAT scale 22: F = 1.46e11, nnz(C) = 7.2e10 = 864 GB, which only fits
    sharded) as far as one GPU can take it: the matrix is built at full size, cut into the 8 row blocks the 8-rank run
    uses (rows_to_threads on F + row_overhead per non-empty row, mul_hash.rs:52-62), and ALL EIGHT per-rank products
    C_r = A[block r] * B are run at full size on this GPU, one after another (the heaviest holds <= 137 GB of C).
    Every block is checked by row sums against A_r*(B*x) for a random x over ALL its rows (column-sensitive), its
    row_ptr is a prefix sum ending at nnz(C_r), its flop count equals the partition's, and the blocks tile C:
    sum of nnz(C_r) = nnz(A*A) = 7.2e10, sum of F_r = F.  The heaviest block and a middle one also give three bounded
    row samples each to the oracle (row_ptr exact, column sets, 1e-10, reference slot order bit-exact)."""
    import torch
    from sparse_matrix_amd import DeviceCsr, _lib
    from helpers import spgemm_rows_vs_oracle

    dev = torch.device("cuda", 0)
    A = synth.rmat(22, dtype=torch.float64, device=dev)
    n = A.rows
    dB = DeviceCsr.from_torch(n, n, A.row_ptr, A.col_idx, A.vals, True)
    offs = dB.balanced_row_blocks(dB, 8)  # SPAM_ROW_OVERHEAD_AUTO: 32 products per 8192-column chunk of B = 16 384 at scale 22
    from sparse_matrix_amd import _lib as _l
    assert int(_l.load().spam_row_overhead_auto(dB._h)) == 32 * 512
    assert offs[0] == 0 and offs[-1] == n and all(b > a for a, b in zip(offs[:-1], offs[1:]))
    b_host = A.cpu_numpy()
    x = synth.dense_vector(n, dtype=torch.float64, device=dev)
    bx = torch.empty(n, dtype=torch.float64, device=dev)
    dB.spmv_dev(x.data_ptr(), bx.data_ptr())
    flops, _ = dB.rows_to_parts(dB, 1)
    cost = [int(flops[offs[r]:offs[r + 1]].sum()) for r in range(8)]
    heaviest = int(np.argmax([int(b_host[0][offs[r + 1]] - b_host[0][offs[r]]) for r in range(8)]))  # most entries of A
    sampled = {heaviest, 4}
    total_nnz_c, total_f = 0, 0
    for r in range(8):
        lo, hi = offs[r], offs[r + 1]
        dAr = dB.row_block(lo, hi)
        C, st = dAr.spgemm(dB, sort_rows=False)
        assert st["flops_f"] == cost[r]
        total_nnz_c += st["nnz_c"]
        total_f += st["flops_f"]
        want = torch.empty(hi - lo, dtype=torch.float64, device=dev)
        got = torch.empty(hi - lo, dtype=torch.float64, device=dev)
        dAr.spmv_dev(bx.data_ptr(), want.data_ptr())
        C.spmv_dev(x.data_ptr(), got.data_ptr())
        torch.cuda.synchronize()
        rel = ((got - want).abs() / want.abs().clamp_min(1e-300)).max().item()
        assert rel <= F64_RTOL, (r, rel)
        if r in sampled:
            s, e = int(b_host[0][lo]), int(b_host[0][hi])
            a_host = (b_host[0][lo:hi + 1] - b_host[0][lo], b_host[1][s:e], b_host[2][s:e])
            for row_lo, target in (((hi - lo) // 2, 1e8), (0, 5e7), ((hi - lo) - max(1, (hi - lo) // 50), 3e7)):
                spgemm_rows_vs_oracle(oracle, dAr, dB, C, a_host, b_host, row_lo, target)
        rp = C.download_row_ptr().astype(np.int64)
        assert rp[-1] == st["nnz_c"] and np.all(np.diff(rp) >= 0)
        C.free()
        del C, dAr, want, got
        # (the output blocks go back to the library's allocator, which hands them to the next block's product when
        # they fit and returns them to the driver by itself when a larger one does not)
    _lib.check(_lib.load().spam_hip_trim())
    assert total_f == int(flops.sum()) and total_f > 140_000_000_000
    assert 71_000_000_000 < total_nnz_c < 73_000_000_000, total_nnz_c  # nnz(A*A) at scale 22 (7.2e10)


def test_matrix_market_bench_shape(spam, oracle, tmp_path):
    """The reference's bench path end to end (spam_csr/src/lib.rs:389-435): Matrix Market text -> DOK ->
    CsrMatrix::from (sorted) -> m.mul_hash(&m) unsorted, against the oracle on the same parsed matrix."""
    from sparse_matrix_amd.mmio import into_float_matrix_market, parse_matrix_market

    m = synth.rmat(8)
    off, idx, val = csr_np(m)
    ents = {(r, int(idx[p])): float(val[p]) for r in range(m.rows) for p in range(int(off[r]), int(off[r + 1]))}
    f = tmp_path / "rmat8.mtx"
    f.write_text(into_float_matrix_market(m.rows, m.cols, ents))
    kind, rows, cols, back = parse_matrix_market(f.read_text())
    assert kind == "real" and back == ents
    A = spam.CsrMatrix.from_dok((rows, cols), back)
    assert np.array_equal(A.offsets, off) and np.array_equal(A.indices, idx) and np.array_equal(A.vals, val)
    C = A * A  # impl Mul -> CsrMatrix<T, false>
    want = oracle.mul_hash((off, idx, val), (off, idx, val), sort_rows=False)
    assert_csr_parity((C.offsets, C.indices, C.vals), want, False, F64_RTOL)


def _exact_unsorted(spam, oracle, ma, mb, a_sorted=True, b_sorted=True, rtol=F64_RTOL):
    A, a = _mk(spam, ma, a_sorted)
    B, b = _mk(spam, mb, b_sorted)
    C = A.mul_hash(B, sort_output=False, reference_order=True)
    wo, wi, wv = oracle.mul_hash(a, b, sort_rows=False)
    assert np.array_equal(C.offsets, wo), "row_ptr"
    assert np.array_equal(C.indices, wi), "col_idx must equal the reference's linprobe slot order"
    if len(wv):
        assert np.max(np.abs(C.vals - wv) / np.abs(wv)) <= rtol
    return C


def test_reference_slot_order_small_kats(spam):
    """The hand-derived slot orders of tests/golden/linprobe_kat.json through the GPU: a 1 x n row times the
    identity inserts the columns in the given order."""
    import json, os

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "linprobe_kat.json")))
    for case in gold["map_drain_order"]:
        keys = case["insert"]
        uniq = list(dict.fromkeys(keys))
        if len(uniq) != len(keys):
            continue  # a CSR row holds distinct columns; repeated keys are covered by the products below
        n = max(keys) + 1
        A = spam.CsrMatrix(1, n, np.ones(len(keys)), np.array(keys, dtype=np.uint64), np.array([0, len(keys)], dtype=np.uint64), False)
        C = A.mul_hash(spam.CsrMatrix.identity(n), reference_order=True)
        assert C.indices.tolist() == case["drain"], (keys, C.indices.tolist())


def test_reference_slot_order_products(spam, oracle):
    """Bit-exact unsorted col_idx against the oracle (the reference's sequential linprobe order): banded cfg1,
    R-MAT with collision runs and repeated keys, unsorted inputs, rectangular shapes, f32."""
    m = synth.banded(10000, 5)
    _exact_unsorted(spam, oracle, m, m)
    for scale in (6, 9, 10):
        m = synth.rmat(scale)
        _exact_unsorted(spam, oracle, m, m)
    _exact_unsorted(spam, oracle, synth.random_csr(200, 300, 0.05, seed=31, signed=False, sort_rows=False),
                    synth.random_csr(300, 500, 0.04, seed=32, signed=False, sort_rows=False), a_sorted=False, b_sorted=False)
    _exact_unsorted(spam, oracle, synth.random_csr(3, 2000, 0.9, seed=33, signed=False),
                    synth.random_csr(2000, 4000, 0.01, seed=34, signed=False))       # rows of ~4000 entries: 8192-slot table
    m = synth.rmat(9, dtype=torch.float32)
    _exact_unsorted(spam, oracle, m, m, rtol=F32_RTOL)


def test_reference_slot_order_property_small_random(spam, oracle):
    """The reference's own test shape (dims 1..4 and up to 40, unsorted CSR from DOK, tests.rs:356-371): col_idx
    bit-exact for every case."""
    rng = np.random.default_rng(99)
    for case in range(80):
        l, m, n = rng.integers(1, 41, 3) if case % 2 else rng.integers(1, 5, 3)
        da = {(int(rng.integers(l)), int(rng.integers(m))): float(rng.uniform(0.5, 1.5)) for _ in range(int(rng.integers(0, 2 * l * m + 1)))}
        db = {(int(rng.integers(m)), int(rng.integers(n))): float(rng.uniform(0.5, 1.5)) for _ in range(int(rng.integers(0, 2 * m * n + 1)))}
        A = spam.CsrMatrix.from_dok((l, m), da, rng=rng)
        B = spam.CsrMatrix.from_dok((m, n), db, rng=rng)
        C = A.mul_hash(B, reference_order=True)
        wo, wi, wv = oracle.mul_hash((A.offsets, A.indices, A.vals), (B.offsets, B.indices, B.vals), sort_rows=False)
        assert np.array_equal(C.offsets, wo) and np.array_equal(C.indices, wi)
        assert C.invariants()


def test_reference_slot_order_long_rows(spam, oracle):
    """Rows beyond 4096 entries (table in HBM): R-MAT scale 13 (one chunk pass), scale 16 (65 536 columns, rows of
    up to 37 k entries, several passes, B pre-split), and an unsorted B (filter path) — col_idx bit-exact."""
    m = synth.rmat(13)
    C = _exact_unsorted(spam, oracle, m, m)
    assert np.diff(C.offsets.astype(np.int64)).max() > 4096
    m = synth.rmat(16)
    _exact_unsorted(spam, oracle, m, m)
    _exact_unsorted(spam, oracle, synth.random_csr(8, 300, 0.6, seed=21, signed=False),
                    synth.random_csr(300, 40_000, 0.05, seed=22, signed=False, sort_rows=False), b_sorted=False)


def test_bin_and_group_boundaries(spam, oracle):
    """Rows engineered to sit exactly on every internal threshold: F and nnz(C_i) of 64/65, 512/513, 2048/2049,
    4096/4097 (bin edges), 3072/3073 and 6143/6144/6145 (dense threshold and hash-group capacity of the heavy
    path), with columns straddling 16 384-column chunk edges; overlapping B rows make F > nnz.  All three orders."""
    targets = [1, 2, 63, 64, 65, 511, 512, 513, 2047, 2048, 2049, 3071, 3072, 3073, 4095, 4096, 4097, 6143, 6144, 6145,
               9000, 16383, 16384, 16385, 40000]
    cols = 70_000
    b_rows = []
    # B row 2t holds t_nnz consecutive columns starting just below a chunk edge; B row 2t+1 overlaps its first half
    for t, n in enumerate(targets):
        start = (16384 * (1 + t % 3) - n // 3) % (cols - n)
        b_rows.append(np.arange(start, start + n))
        b_rows.append(np.arange(start, start + max(1, n // 2)))
    b_off = np.concatenate([[0], np.cumsum([len(r) for r in b_rows])]).astype(np.uint64)
    b_idx = np.concatenate(b_rows).astype(np.uint64)
    rng = np.random.default_rng(5)
    b_val = rng.uniform(0.5, 1.5, b_idx.size)
    # A row t references B rows 2t and 2t+1 (in that order), plus a last row referencing everything
    a_rows = [[2 * t, 2 * t + 1] for t in range(len(targets))] + [list(range(2 * len(targets)))]
    a_off = np.concatenate([[0], np.cumsum([len(r) for r in a_rows])]).astype(np.uint64)
    a_idx = np.concatenate(a_rows).astype(np.uint64)
    a_val = rng.uniform(0.5, 1.5, a_idx.size)
    A = spam.CsrMatrix(len(a_rows), len(b_rows), a_val, a_idx, a_off, True)
    B = spam.CsrMatrix(len(b_rows), cols, b_val, b_idx, b_off, True)
    a, b = (a_off, a_idx, a_val), (b_off, b_idx, b_val)
    want_s = oracle.mul_hash(a, b, sort_rows=True)
    want_u = oracle.mul_hash(a, b, sort_rows=False)
    assert np.diff(want_s[0].astype(np.int64))[:len(targets)].tolist() == targets
    Cs = A.mul_hash(B, sort_output=True)
    assert_csr_parity((Cs.offsets, Cs.indices, Cs.vals), want_s, True, F64_RTOL)
    Ct = A.mul_hash(B, sort_output=False)
    assert_csr_parity((Ct.offsets, Ct.indices, Ct.vals), want_u, False, F64_RTOL)
    Cr = A.mul_hash(B, reference_order=True)
    assert np.array_equal(Cr.offsets, want_u[0]) and np.array_equal(Cr.indices, want_u[1])
    assert np.max(np.abs(Cr.vals - want_u[2]) / want_u[2]) <= F64_RTOL


def test_sort_rows_in_place(spam, oracle):
    """spam_csr_sort_rows: unsorted rows of 8 .. 30 000 entries (bitonic bins and the dense-window path) become
    strictly increasing with their values attached; SpMV before == after."""
    import torch as _t
    m = synth.powerlaw(20_000, heavy_every=1000, heavy_nnz=12_000, light_nnz=8, dtype=_t.float64)
    off, idx, val = csr_np(m)
    rng = np.random.default_rng(3)
    pi, pv = idx.copy(), val.copy()
    o = off.astype(np.int64)
    for r in range(m.rows):                      # shuffle inside rows
        p = rng.permutation(o[r + 1] - o[r]) + o[r]
        pi[o[r]:o[r + 1]], pv[o[r]:o[r + 1]] = idx[p], val[p]
    A = spam.CsrMatrix(m.rows, m.cols, pv, pi, off, False)
    d = A.device()
    assert d.invariants()
    d.sort_rows()
    assert d.dims()[4] is True and d.invariants()
    go, gi, gv = d.download()
    assert np.array_equal(go, off) and np.array_equal(gi, idx) and np.array_equal(gv, val)


def test_column_range_limit(spam):
    """Column keys must stay below u32::MAX (mul_hash.rs:12, linprobe/src/set.rs:110): cols = 2^32 - 1 is refused at
    upload (SPAM_E_INDEX_RANGE); cols = 2^32 - 2 with an entry in the last column is accepted."""
    off = np.array([0, 1], dtype=np.uint64)
    val = np.array([1.0])
    with pytest.raises(spam.SpamError) as e:
        spam.CsrMatrix(1, 2 ** 32 - 1, val, np.array([5], dtype=np.uint64), off, True).device()
    assert e.value.code == 2
    ok = spam.CsrMatrix(1, 2 ** 32 - 2, val, np.array([2 ** 32 - 3], dtype=np.uint64), off, True)
    assert ok.device().invariants()
    with pytest.raises(spam.SpamError):          # index >= cols (invariant 5)
        spam.CsrMatrix(1, 10, val, np.array([10], dtype=np.uint64), off, True).device()


def test_randomized_shapes_all_orders(spam, oracle):
    """120 random products (rows 1..400, inner 1..400, cols 1..60 000, densities 0.1 %..60 %, sorted or shuffled
    operands, f64 / f32): row_ptr exact; sorted col_idx exact; table order set-equal; reference order bit-exact."""
    rng = np.random.default_rng(2026)
    for case in range(120):
        l, m = int(rng.integers(1, 401)), int(rng.integers(1, 401))
        n = int(rng.choice([1, 7, 64, 500, 5000, 20000, 60000]))
        da = float(rng.choice([0.001, 0.01, 0.1, 0.6]))
        db = float(rng.choice([0.001, 0.01, 0.05, 0.3])) if n <= 5000 else float(rng.choice([0.0005, 0.003, 0.02]))
        sa, sb = bool(rng.integers(2)), bool(rng.integers(2))
        dt = torch.float32 if case % 5 == 4 else torch.float64
        ma = synth.random_csr(l, m, da, seed=1000 + case, signed=False, sort_rows=sa, dtype=dt)
        mb = synth.random_csr(m, n, db, seed=5000 + case, signed=False, sort_rows=sb, dtype=dt)
        A, a = _mk(spam, ma, sa)
        B, b = _mk(spam, mb, sb)
        rtol = F64_RTOL if dt == torch.float64 else F32_RTOL
        ws = oracle.mul_hash(a, b, sort_rows=True)
        wu = oracle.mul_hash(a, b, sort_rows=False)
        Cs = A.mul_hash(B, sort_output=True)
        assert_csr_parity((Cs.offsets, Cs.indices, Cs.vals), ws, True, rtol)
        Ct = A.mul_hash(B)
        assert_csr_parity((Ct.offsets, Ct.indices, Ct.vals), wu, False, rtol)
        Cr = A.mul_hash(B, reference_order=True)
        assert np.array_equal(Cr.offsets, wu[0]) and np.array_equal(Cr.indices, wu[1]), f"case {case}"


@pytest.mark.parametrize("n_out", [3, 40, 300, 1500, 3000, 6000])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_first_product_is_stored_not_added_to_zero(spam, oracle, n_out, dtype):
    """`or_insert(t1)` STORES the first product of a column (/root/reference/spam_csr/src/mul_hash.rs:157-161,
    /root/reference/linprobe/src/map.rs:113-116) — so a product that underflows to -0.0 comes out as -0.0, and
    (-0.0) + (-0.0) stays -0.0, while one +0.0 among them gives +0.0.  An accumulator that starts at +0.0 would flip
    those signs; the kernels start theirs at -0.0.  Known answer, hand-derived, one row per numeric bin (16-lane and
    64-lane hash tables, the two one-row-per-workgroup tables, the rank-addressed heavy path beyond 4096 entries):
    row 0 of A = [t, t] against two rows of B of n_out entries each with t*b underflowing: columns where both
    products are -0.0 keep the sign bit, columns where the second is +0.0 lose it."""
    tiny = dtype(1e-200) if dtype == np.float64 else dtype(1e-30)
    a_vals = np.array([-tiny, tiny], dtype=dtype)              # A = 1 x 2
    A = spam.CsrMatrix(1, 2, a_vals, np.array([0, 1], dtype=np.uint64), np.array([0, 2], dtype=np.uint64), True)
    cols = np.arange(n_out, dtype=np.uint64)
    b0 = np.full(n_out, tiny, dtype=dtype)                     # (-tiny) * tiny      -> -0.0 (first product of every column)
    b1 = np.where(cols % 2 == 0, -tiny, tiny).astype(dtype)    # tiny * (-/+ tiny)   -> -0.0 on even, +0.0 on odd columns
    B = spam.CsrMatrix(2, n_out, np.concatenate([b0, b1]), np.concatenate([cols, cols]),
                       np.array([0, n_out, 2 * n_out], dtype=np.uint64), True)
    want_sign = (cols % 2 == 0)                                # -0.0 + -0.0 = -0.0 ; -0.0 + +0.0 = +0.0
    host_a = (A.offsets, A.indices, A.vals)
    host_b = (B.offsets, B.indices, B.vals)
    for sort_rows in (True, False):
        C = A.mul_hash(B, sort_output=sort_rows)
        wo, wi, wv = oracle.mul_hash(host_a, host_b, sort_rows=sort_rows)
        assert C.nnz() == n_out and np.all(C.vals == 0), "cancellation / underflow keeps explicit zeros"
        order = np.argsort(C.indices, kind="stable")
        assert np.array_equal(np.signbit(C.vals[order]), want_sign), "sign of zero (hand-derived)"
        assert np.array_equal(np.signbit(wv[np.argsort(wi, kind='stable')]), want_sign), "the oracle agrees with the hand derivation"
    # a single product of -0.0 per column: the first (and only) product is what the reference stores
    A1 = spam.CsrMatrix(1, 1, np.array([-tiny], dtype=dtype), np.array([0], dtype=np.uint64), np.array([0, 1], dtype=np.uint64), True)
    B1 = spam.CsrMatrix(1, n_out, b0, cols, np.array([0, n_out], dtype=np.uint64), True)
    C1 = A1.mul_hash(B1, sort_output=True)
    assert np.all(C1.vals == 0) and np.all(np.signbit(C1.vals)), "a lone -0.0 product must stay -0.0"
