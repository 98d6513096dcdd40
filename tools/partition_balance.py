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
RO = int(os.environ["ROW_OVERHEAD"]) if "ROW_OVERHEAD" in os.environ else int(_lib.load().spam_row_overhead_auto(B._h))
print("row overhead", RO, "products per non-empty row", "(library default for this B)" if "ROW_OVERHEAD" not in os.environ else "(forced)")
if os.environ.get("WEIGHTED", "1") == "1":
    offs = np.array(spdist.balanced_row_blocks(flops.astype(np.int64), P, RO))
print("offsets", offs.tolist())
tot = []
for r in range(P):  # block-major, the last of three runs: the first sizes the library's allocator for this block's C
    lo, hi = int(offs[r]), int(offs[r + 1])
    blk = B.row_block(lo, hi)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        C, st = blk.spgemm(B, sort_rows=False)
        torch.cuda.synchronize(); t = time.perf_counter() - t0
        C.free()
    tot.append((t * 1e3, st["flops_f"], st["nnz_c"], hi - lo, st["ms_symbolic"], st["ms_numeric"]))
# the product in one piece, same process (t1 of the strong-scaling estimate), and the per-rank fixed work of the
# multi-GPU step (spam_spgemm_rank_step_rccl): flop count of 1/P of the rows + partition of the all-gathered counts
t1s = []
one_piece = int(os.environ.get("ONE_PIECE", "1" if scale <= 20 else "0"))  # (scale 22: C = 864 GB does not fit one GPU)
st1 = {"ms_symbolic": float("nan"), "ms_numeric": float("nan")}
for rep in range(3 if one_piece else 0):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    C, st1 = B.spgemm(B, sort_rows=False)
    torch.cuda.synchronize(); t1s.append((time.perf_counter() - t0) * 1e3)
    C.free()
t1 = float(np.median(t1s[1:])) if one_piece else float("nan")
import ctypes
L = _lib.load()
n = A.rows
dfl = torch.empty(n, dtype=torch.int64, device=dev)
offs_out = (ctypes.c_uint64 * (P + 1))()
fixed = []
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(L.spam_row_flops_dev(B._h, B._h, 0, n // P, ctypes.c_void_p(dfl.data_ptr())))
    _lib.check(L.spam_rows_to_parts_from_flops_dev(ctypes.c_void_p(dfl.data_ptr()), n, P, RO, offs_out))
    torch.cuda.synchronize(); fixed.append((time.perf_counter() - t0) * 1e3)
fixed_ms = float(np.median(fixed[1:]))
for r, (t, f, n, rows, ms_s, ms_n) in enumerate(tot):
    print(f"block {r}: rows {rows:8d}  F {f:.3e}  nnzC {n:.3e}  {t:7.1f} ms (sym {ms_s:.1f} num {ms_n:.1f})")
ts = np.array([t[0] for t in tot])
print(f"sum {ts.sum():.1f} ms  max {ts.max():.1f}  mean {ts.mean():.1f}  balance efficiency mean/max = {ts.mean()/ts.max():.3f}")
exchange_ms = 1.0  # all-gather-v of B over xGMI (scale 22: 0.72 GB received per rank over 7 links at ~150 GB/s; not measurable on one GPU)
if not one_piece:
    print(f"per-rank fixed work (flop count of rows/{P} + partition) = {fixed_ms:.2f} ms; the product in one piece does not fit one GPU "
          f"at this scale: no t1, no efficiency estimate — the load balance above (mean/max) is what this run shows")
    sys.exit(0)
eff = t1 / (P * (ts.max() + fixed_ms + exchange_ms))
print(f"one piece t1 = {t1:.1f} ms (symbolic {st1['ms_symbolic']:.1f}, numeric {st1['ms_numeric']:.1f}); per-rank fixed work "
      f"(flop count of rows/{P} + partition) = {fixed_ms:.2f} ms; assumed exchange {exchange_ms:.1f} ms")
print(f"predicted strong-scaling efficiency at {P} GPUs = t1 / ({P} * (max block + fixed + exchange)) = "
      f"{t1:.1f} / ({P} * ({ts.max():.1f} + {fixed_ms:.2f} + {exchange_ms:.1f})) = {eff:.3f}   [NOT a measurement on {P} GPUs]")
