#!/usr/bin/env python3
"""Symbolic phase alone (spam_spgemm_symbolic) of C = A*A on R-MAT at a given scale.  Diagnostic only: lets a build
whose symbolic kernel is deliberately wrong (timing ablations) run without the numeric phase behind it."""
import os, sys, time
import ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_amd import DeviceCsr, _lib, synth
from sparse_matrix_amd.csr import SpgemmStats


def main():
    if os.environ.get("PROBE_LIB"):
        _lib.LIB_PATH = os.environ["PROBE_LIB"]
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda", 0)
    _lib.ensure_device(0)
    L = _lib.load()
    _lib.check(L.spam_hip_set_stream(torch.cuda.current_stream().cuda_stream))
    A = synth.rmat(scale, dtype=torch.float64, device=dev)
    dA = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
    for r in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        plan, nnz = dA.spgemm_symbolic(dA)
        torch.cuda.synchronize(); t = time.perf_counter() - t0
        st = SpgemmStats()
        _lib.check(L.spam_spgemm_plan_stats(plan, C.byref(st)))
        L.spam_spgemm_plan_free(plan)
        d = st.asdict()
        print(f"rep {r}: wall {t*1e3:.1f} ms  nnzC={nnz:.4e}  flops {d['ms_flops']:.2f} sym {d['ms_symbolic']:.2f} ms", flush=True)


main()
