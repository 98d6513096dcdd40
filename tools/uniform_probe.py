#!/usr/bin/env python3
"""Diagnostic: SpMV on a matrix WITHOUT column skew (every row `k` entries, columns uniform) — the worst case for the
slab tables: what does spam_spmv_prepare choose, and what do the paths cost?"""
import os, sys, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_amd import DeviceCsr, _lib, synth

dev = torch.device("cuda", 0)
_lib.ensure_device(0)
_lib.check(_lib.load().spam_hip_set_stream(torch.cuda.current_stream().cuda_stream))
n, k = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 16
A = synth.powerlaw(n, heavy_every=1 << 40, heavy_nnz=k, light_nnz=k, dtype=torch.float64, device=dev)
x = synth.dense_vector(A.cols, dtype=torch.float64, device=dev)
y = torch.empty(A.rows, dtype=torch.float64, device=dev)
rowidx = torch.repeat_interleave(torch.arange(A.rows, device=dev), A.row_ptr[1:] - A.row_ptr[:-1])
ref = torch.zeros(A.rows, dtype=torch.float64, device=dev).index_add_(0, rowidx, A.vals * x[A.col_idx.to(torch.int64)])
nbytes = A.nnz * 12 + (A.rows + 1) * 4 + A.cols * 8 + A.rows * 8
print(A.stats(), "alg bytes %.3f GB" % (nbytes / 1e9), flush=True)
for env in ({"SPAM_SPMV_SLICED": "1"}, {"SPAM_SPMV_SLICED": "0"}, {"SPAM_SPMV_SLICED": "2"}, {"SPAM_SPMV_SLICED": "2", "SPAM_SPMV_RG": "0"}):
    for kk in [e for e in os.environ if e.startswith("SPAM_SPMV_")]:
        del os.environ[kk]
    os.environ.update(env)
    dA = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
    dA.spmv_prepare(x.data_ptr())
    dA.spmv_dev(x.data_ptr(), y.data_ptr())
    torch.cuda.synchronize()
    err = float(((y - ref).abs() / ref.abs().clamp_min(1e-300)).max())
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        dA.spmv_dev(x.data_ptr(), y.data_ptr())
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    info = dA.spmv_plan_info()["sliced"]
    print(json.dumps(env), "-> %.4f ms  %.0f GB/s (%.1f%%)  err %.1e  sliced %s rg %s cover8 %.3f ms_plain %.3f ms_sliced %.3f" % (
        ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80, err, info["active"], info["row_groups"]["active"], info["cover_table"],
        info["ms_plain"], info["ms_sliced"]), flush=True)
    del dA
