#!/usr/bin/env python3
"""Diagnostic: SpMV over a zoo of matrix shapes in AUTO mode — what spam_spmv_prepare chooses, what it costs, whether the
result agrees with an index_add reference built by torch."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_amd import DeviceCsr, _lib, synth
from sparse_matrix_amd.synth import Csr
dev = torch.device("cuda", 0); _lib.ensure_device(0)
_lib.check(_lib.load().spam_hip_set_stream(torch.cuda.current_stream().cuda_stream))


def timed(dA, x, y, reps):
    for _ in range(3):
        dA.spmv_dev(x.data_ptr(), y.data_ptr())
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        dA.spmv_dev(x.data_ptr(), y.data_ptr())
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def rect(rows, cols, per_row, dtype, seed=5):
    """every row per_row entries, columns uniform (duplicates merged)"""
    r = torch.arange(rows, dtype=torch.int64, device=dev).repeat_interleave(per_row)
    h = synth.mix64(torch.arange(r.numel(), dtype=torch.int64, device=dev) + seed * 7919)
    c = (h & 0x7FFFFFFFFFFF) % cols
    return synth._csr_from_keys((r << 32) | c, rows, cols, seed, dtype, False)


def few_giant_rows(n, giants, giant_nnz, light, dtype):
    a = synth.powerlaw(n, heavy_every=max(1, n // giants), heavy_nnz=giant_nnz, light_nnz=light, dtype=dtype, device=dev)
    return a


zoo = [
    ("banded 10M bw17 f64", lambda: synth.banded(10_000_000, 17, dtype=torch.float64, device=dev)),
    ("banded 2M bw5 f64", lambda: synth.banded(2_000_000, 5, dtype=torch.float64, device=dev)),
    ("tall 20M x 1M, 8/row f64", lambda: rect(20_000_000, 1_000_000, 8, torch.float64)),
    ("wide 1M x 20M, 64/row f64", lambda: rect(1_000_000, 20_000_000, 64, torch.float64)),
    ("long rows 100k x 4M, 1000/row f64", lambda: rect(100_000, 4_000_000, 1000, torch.float64)),
    ("4 giant rows of 4M + 8/row, 4M f64", lambda: few_giant_rows(4_000_000, 4, 4_000_000, 8, torch.float64)),
    ("power-law 1M, 1/500 rows of 20k, f32", lambda: synth.powerlaw(1_000_000, 500, 20_000, 8, dtype=torch.float32, device=dev)),
    ("rmat scale 20 f32", lambda: synth.rmat(20, dtype=torch.float32, device=dev)),
    ("rmat scale 21 i64", lambda: synth.rmat(21, dtype=torch.float64, device=dev)),
]
only = sys.argv[1:]
for name, make in zoo:
    if only and not any(o in name for o in only):
        continue
    A = make()
    if "i64" in name:
        A = Csr(A.rows, A.cols, A.row_ptr, A.col_idx, (A.vals * 1000).to(torch.int64))
    dt = A.vals.dtype
    x = synth.dense_vector(A.cols, dtype=torch.float64, device=dev)
    x = (x * 1000).to(dt) if dt == torch.int64 else x.to(dt)
    y = torch.empty(A.rows, dtype=dt, device=dev)
    lens = A.row_ptr[1:] - A.row_ptr[:-1]
    # reference: hipSPARSE through torch.sparse in f64 (index_add_ takes minutes when millions of entries share one row, and
    # torch.segment_reduce returned wrong sums for 20 M segments); i64 by wrapping prefix sums
    if dt == torch.int64:
        prod = A.vals * x[A.col_idx.to(torch.int64)]
        ps = torch.zeros(A.nnz + 1, dtype=torch.int64, device=dev)
        torch.cumsum(prod, 0, out=ps[1:])
        ref = ps[A.row_ptr[1:]] - ps[A.row_ptr[:-1]]
        del ps, prod
    else:
        S = torch.sparse_csr_tensor(A.row_ptr, A.col_idx.to(torch.int64), A.vals.to(torch.float64), size=(A.rows, A.cols))
        ref = S @ x.to(torch.float64)
        del S
        if not bool(torch.isfinite(ref).all()):  # (seen on the f32 power-law case: non-finite sums from torch.sparse)
            prod = A.vals.to(torch.float64) * x[A.col_idx.to(torch.int64)].to(torch.float64)
            ps = torch.zeros(A.nnz + 1, dtype=torch.float64, device=dev)
            torch.cumsum(prod, 0, out=ps[1:])
            ref = ps[A.row_ptr[1:]] - ps[A.row_ptr[:-1]]  # prefix sums: good to ~1e-9 relative here
            del ps, prod
            name += " (ref: prefix sums)"
    esz = A.vals.element_size()
    nbytes = A.nnz * (4 + esz) + (A.rows + 1) * 4 + A.cols * esz + A.rows * esz
    dA = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
    reps = 20
    t0 = timed(dA, x, y, reps)
    tp = time.perf_counter()
    dA.spmv_prepare(x.data_ptr())
    torch.cuda.synchronize()
    tp = time.perf_counter() - tp
    t1 = timed(dA, x, y, reps)
    assert bool(torch.isfinite(y.to(torch.float64)).all()) if dt != torch.int64 else True, "non-finite y"
    if dt == torch.int64:
        err = float((y != ref).sum())
    else:
        err = float(((y.to(torch.float64) - ref).abs() / ref.abs().clamp_min(1e-300)).max())
    info = dA.spmv_plan_info()
    sl = info["sliced"]
    mode = "row groups" if sl["row_groups"]["active"] else "sliced tiles" if sl["active"] else f"plain, hot {info['hot_columns']}"
    print(f"{name:40s} nnz {A.nnz:>10} max row {int(lens.max()):>8}: unprepared {t0 * 1e3:8.1f} us ({nbytes / t0 / 8e7:5.1f} %)  "
          f"prepared {t1 * 1e3:8.1f} us ({nbytes / t1 / 8e7:5.1f} %) [{mode}; timed plain {sl['ms_plain'] * 1e3:.0f} / sliced "
          f"{sl['ms_sliced'] * 1e3:.0f} us; prepare {tp * 1e3:.0f} ms]  err {err:.1e}", flush=True)
    del dA, A, x, y, ref
    torch.cuda.empty_cache()
