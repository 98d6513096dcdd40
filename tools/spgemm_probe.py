#!/usr/bin/env python3
"""SpGEMM probe: C = A*A on R-MAT at a given scale, prints the library's phase timings.  Diagnostic only."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_amd import DeviceCsr, _lib, synth

def main():
    if os.environ.get("PROBE_LIB"):
        _lib.LIB_PATH = os.environ["PROBE_LIB"]  # A/B two builds of the library
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    sort_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    dt = torch.float32 if os.environ.get("PROBE_F32") else torch.float64
    dev = torch.device("cuda", 0)
    _lib.ensure_device(0)
    _lib.check(_lib.load().spam_hip_set_stream(torch.cuda.current_stream().cuda_stream))
    A = synth.rmat(scale, dtype=dt, device=dev)
    dA = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
    print("A", A.stats(), flush=True)
    for r in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        C, st = dA.spgemm(dA, sort_rows=bool(sort_rows))
        torch.cuda.synchronize(); t = time.perf_counter() - t0
        if r == reps - 1 and not os.environ.get('SPAM_DBG_HEAVY'):
            # size-independent property: C*1 == A*(A*1) row by row (uses the SpMV kernel on C and on A)
            ones = torch.ones(A.cols, dtype=dt, device=dev)
            t1 = torch.empty(A.rows, dtype=dt, device=dev); t2 = torch.empty_like(t1); t3 = torch.empty_like(t1)
            dA.spmv_dev(ones.data_ptr(), t1.data_ptr()); dA.spmv_dev(t1.data_ptr(), t2.data_ptr())
            C.spmv_dev(ones.data_ptr(), t3.data_ptr()); torch.cuda.synchronize()
            rel = ((t3 - t2).abs() / t2.abs().clamp_min(1e-300)).max().item()
            relv = ((t3 - t2).abs() / t2.abs().clamp_min(1e-300))
            bad = torch.nonzero(relv > (1e-9 if dt == torch.float64 else 1e-3)).flatten().cpu().numpy()
            rp = C.download_row_ptr().astype(np.int64); nzr = np.diff(rp)
            fl, _ = dA.rows_to_parts(dA, 1)
            print("bad rows", len(bad), "of", A.rows)
            if len(bad):
                print(" first bad", bad[:10], "nnz_row", nzr[bad[:10]], "F_row", fl[bad[:10]], "rel", relv[bad[:10]].cpu().numpy())
                print(" bad nnz_row range", nzr[bad].min(), nzr[bad].max(), " bad row_ptr range", rp[bad].min(), rp[bad].max(), " 2^32=", 2**32)
                for lo, hi in [(1,64),(65,512),(513,2048),(2049,4096),(4097,10**9)]:
                    m = (nzr[bad] >= lo) & (nzr[bad] <= hi); print(f"   bin nnz {lo}-{hi}: bad {m.sum()} / total {((nzr>=lo)&(nzr<=hi)).sum()}")
            print(f"row-sum property max rel err {rel:.3e}; invariants(sorted={bool(sort_rows)}) -> {C.invariants() if sort_rows else 'n/a'}", flush=True)
        C.free()
        print(f"rep {r}: wall {t*1e3:.1f} ms  F={st['flops_f']:.3e} nnzC={st['nnz_c']:.3e}  "
              f"flops {st['ms_flops']:.2f} sym {st['ms_symbolic']:.2f} alloc {st['ms_alloc']:.2f} num {st['ms_numeric']:.2f} ms  "
              f"-> {st['flops_f']/t/1e9:.1f} Gprod/s {st['nnz_c']/t/1e6:.0f} Mnnz/s", flush=True)
    print("heavy groups: dense", st["heavy_dense_groups"], "nnz", st["heavy_dense_nnz"], " hash", st["heavy_hash_groups"], "nnz", st["heavy_hash_nnz"])
    print("sym bins", st["rows_sym_bin"][:4], "num bins", st["rows_num_bin"][:5], "heavy items", st["heavy_items"])

main()
