#!/usr/bin/env python3
"""Per-rank fixed costs of one multi-GPU SpGEMM step (everything except the all-gather), timed on one GPU for
block r of P.  Diagnostic only."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_amd import DeviceCsr, _lib, synth, dist as spdist
scale, P, r = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0); _lib.ensure_device(0)
_lib.check(_lib.load().spam_hip_set_stream(torch.cuda.current_stream().cuda_stream))
A = synth.rmat(scale, dtype=torch.float64, device=dev)
def sync(): torch.cuda.synchronize()
for rep in range(3):
    sync(); t0 = time.perf_counter()
    B = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True); sync(); t1 = time.perf_counter()
    offs = B.balanced_row_blocks(B, P, spdist.ROW_OVERHEAD_PRODUCTS); sync(); t2 = time.perf_counter()
    blk = B.row_block(offs[r], offs[r + 1]); sync(); t3 = time.perf_counter()
    C, st = blk.spgemm(B, sort_rows=False); sync(); t4 = time.perf_counter()
    C.free(); sync(); t5 = time.perf_counter()
    print(f"rep {rep}: from_torch {1e3*(t1-t0):.2f}  partition {1e3*(t2-t1):.2f}  row_block {1e3*(t3-t2):.2f}  spgemm {1e3*(t4-t3):.2f} "
          f"(dev: flops {st['ms_flops']:.2f} sym {st['ms_symbolic']:.2f} alloc {st['ms_alloc']:.2f} num {st['ms_numeric']:.2f})  free {1e3*(t5-t4):.2f} ms")
