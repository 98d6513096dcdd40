#!/usr/bin/env python3
"""Symbolic / numeric phase times of one rank block (r of P, scale s) under a list of environment variants (A/B in one
process).  usage: sym_block_probe.py 20 8 3 '[{"SPAM_SYM_GIANT_SHIFT": -2}, {}]'   Diagnostic."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_amd import DeviceCsr, _lib, synth, dist as spdist
scale, P, r = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
variants = json.loads(sys.argv[4]) if len(sys.argv) > 4 else [{}]
dev = torch.device("cuda", 0); _lib.ensure_device(0)
_lib.check(_lib.load().spam_hip_set_stream(torch.cuda.current_stream().cuda_stream))
A = synth.rmat(scale, dtype=torch.float64, device=dev)
B = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
offs = B.balanced_row_blocks(B, P, spdist.ROW_OVERHEAD_PRODUCTS)
blocks = range(P) if r < 0 else [r]
for v in variants:
    for k in [k for k in os.environ if k.startswith("SPAM_SYM_") or k.startswith("SPAM_HEAVY_")]:
        del os.environ[k]
    for k, val in v.items():
        os.environ[k] = str(val)
    for rb in blocks:
        blk = B.row_block(offs[rb], offs[rb + 1])
        ts = []
        for _ in range(4):
            C, st = blk.spgemm(B, sort_rows=False)
            C.free()
            ts.append((st["ms_symbolic"], st["ms_numeric"], st["ms_flops"]))
        a = np.median(np.array(ts[1:]), axis=0)
        print(json.dumps(v), f"block {rb}: symbolic {a[0]:.2f} ms numeric {a[1]:.2f} ms flops {a[2]:.2f} ms", flush=True)
