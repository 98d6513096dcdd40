#!/usr/bin/env python3
"""Diagnostic: time the prepared SpMV of the bench matrix with the library named by PROBE_LIB (A/B of two builds on one box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_amd import _lib
if os.environ.get("PROBE_LIB"):
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
from sparse_matrix_amd import DeviceCsr, synth
dev = torch.device("cuda", 0); _lib.ensure_device(0)
_lib.check(_lib.load().spam_hip_set_stream(torch.cuda.current_stream().cuda_stream))
A = synth.rmat_rect(10_000_000, 160_000_000, dtype=torch.float64, device=dev, oversample=1.30)
x = synth.dense_vector(A.cols, dtype=torch.float64, device=dev); y = torch.empty(A.rows, dtype=torch.float64, device=dev)
dA = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
os.environ["SPAM_SPMV_SLICED"] = "2"
dA.spmv_prepare(x.data_ptr())
res = []
for rep in range(4):
    for _ in range(3):
        dA.spmv_dev(x.data_ptr(), y.data_ptr())
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        dA.spmv_dev(x.data_ptr(), y.data_ptr())
    e.record(); torch.cuda.synchronize()
    res.append(s.elapsed_time(e) / 20)
print(os.environ.get("PROBE_LIB", "default"), " ".join(f"{r:.4f}" for r in res), "ms", flush=True)
