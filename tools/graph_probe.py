#!/usr/bin/env python3
"""Diagnostic: K prepared SpMVs of the bench matrix launched one by one vs replayed as ONE HIP graph (wall clock per step)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_amd import DeviceCsr, _lib, synth

dev = torch.device("cuda", 0)
_lib.ensure_device(0)
L = _lib.load()
_lib.check(L.spam_hip_set_stream(torch.cuda.current_stream().cuda_stream))
A = synth.rmat_rect(10_000_000, 160_000_000, dtype=torch.float64, device=dev, oversample=1.30)
x = synth.dense_vector(A.cols, dtype=torch.float64, device=dev)
y = torch.empty(A.rows, dtype=torch.float64, device=dev)
dA = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
dA.spmv_prepare(x.data_ptr())
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3):
    dA.spmv_dev(x.data_ptr(), y.data_ptr())
torch.cuda.synchronize()
yref = y.clone()


def direct():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        dA.spmv_dev(x.data_ptr(), y.data_ptr())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / K


side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    _lib.check(L.spam_hip_set_stream(side.cuda_stream))
    with torch.cuda.graph(g, stream=side):
        for _ in range(K):
            dA.spmv_dev(x.data_ptr(), y.data_ptr())
_lib.check(L.spam_hip_set_stream(torch.cuda.current_stream().cuda_stream))


def graph():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / K


for rep in range(4):
    d = direct()
    y.zero_()
    gr = graph()
    ok = bool(((y - yref).abs() <= 1e-12 * yref.abs()).all())
    print(f"rep {rep}: direct {d:.4f} ms/step, graph {gr:.4f} ms/step, graph result ok {ok}", flush=True)
