#!/usr/bin/env python3
"""One row block of C = A*A (block r of P at a given scale) for rocprofv3 --kernel-trace --stats.  Diagnostic."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_amd import DeviceCsr, _lib, synth, dist as spdist
scale, P, r = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0); _lib.ensure_device(0)
_lib.check(_lib.load().spam_hip_set_stream(torch.cuda.current_stream().cuda_stream))
A = synth.rmat(scale, dtype=torch.float64, device=dev)
B = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
offs = B.balanced_row_blocks(B, P, spdist.ROW_OVERHEAD_PRODUCTS)
blk = B.row_block(offs[r], offs[r + 1])
for _ in range(2):
    C, st = blk.spgemm(B, sort_rows=False)
    C.free()
print({k: st[k] for k in ("ms_flops", "ms_symbolic", "ms_numeric", "rows_sym_bin", "rows_num_bin")})
