#!/usr/bin/env python3
"""How well do flop-balanced row blocks balance TIME?  Runs each of the P blocks of C = A*A one after another on
one GPU and reports per-block time; max/mean is the load-balance efficiency an 8-GPU run would see (minus
communication).  Diagnostic only."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_amd import DeviceCsr, _lib, synth

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
_lib.ensure_device(0)
_lib.check(_lib.load().spam_hip_set_stream(torch.cuda.current_stream().cuda_stream))
A = synth.rmat(scale, dtype=torch.float64, device=dev)
B = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
from sparse_matrix_amd import dist as spdist
flops, offs = B.rows_to_parts(B, P)
if os.environ.get("WEIGHTED", "1") == "1":
    offs = np.array(spdist.balanced_row_blocks(flops.astype(np.int64), P, int(os.environ.get("ROW_OVERHEAD", spdist.ROW_OVERHEAD_PRODUCTS))))
print("offsets", offs.tolist())
tot = []
for rep in range(2):
    times = []
    for r in range(P):
        lo, hi = int(offs[r]), int(offs[r + 1])
        blk = B.row_block(lo, hi)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        C, st = blk.spgemm(B, sort_rows=False)
        torch.cuda.synchronize(); t = time.perf_counter() - t0
        times.append((t * 1e3, st["flops_f"], st["nnz_c"], hi - lo, st["ms_symbolic"], st["ms_numeric"]))
        C.free()
    tot = times
for r, (t, f, n, rows, ms_s, ms_n) in enumerate(tot):
    print(f"block {r}: rows {rows:8d}  F {f:.3e}  nnzC {n:.3e}  {t:7.1f} ms (sym {ms_s:.1f} num {ms_n:.1f})")
ts = np.array([t[0] for t in tot])
print(f"sum {ts.sum():.1f} ms  max {ts.max():.1f}  mean {ts.mean():.1f}  balance efficiency mean/max = {ts.mean()/ts.max():.3f}")
