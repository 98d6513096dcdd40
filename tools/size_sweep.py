#!/usr/bin/env python3
"""Diagnostic: SpMV across matrix sizes (R-MAT scale 14..23, edge factor 16, f64): unprepared, prepared, share of the roof."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_amd import DeviceCsr, _lib, synth
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


for scale in [int(a) for a in sys.argv[1:]] or range(14, 24):
    A = synth.rmat(scale, dtype=torch.float64, device=dev)
    x = synth.dense_vector(A.cols, dtype=torch.float64, device=dev); y = torch.empty(A.rows, dtype=torch.float64, device=dev)
    nbytes = A.nnz * 12 + (A.rows + 1) * 4 + A.cols * 8 + A.rows * 8
    reps = 200 if scale < 20 else 30
    dA = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
    t0 = timed(dA, x, y, reps)
    dA.spmv_prepare(x.data_ptr())
    t1 = timed(dA, x, y, reps)
    info = dA.spmv_plan_info()
    mode = "row groups" if info["sliced"]["row_groups"]["active"] else "sliced tiles" if info["sliced"]["active"] else f"plain, hot table {info['hot_columns']}"
    print(f"scale {scale}: nnz {A.nnz:>10}  unprepared {t0 * 1e3:8.1f} us ({nbytes / t0 / 8e7:5.1f} %)  prepared {t1 * 1e3:8.1f} us "
          f"({nbytes / t1 / 8e7:5.1f} % of 8 TB/s)  [{mode}]", flush=True)
    del dA, A, x, y
