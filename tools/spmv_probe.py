#!/usr/bin/env python3
"""A/B probe for the SpMV kernel variants on one GPU, interleaved rounds in ONE process
(cdna_hip_programming.md §5.4 rule 24).  Not part of the product or the tests."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_amd import _lib
if os.environ.get("PROBE_LIB"):
    _lib.LIB_PATH = os.environ["PROBE_LIB"]  # A/B two builds of the library
from sparse_matrix_amd import DeviceCsr, synth

def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "rmat"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda", 0)
    _lib.ensure_device(0)
    _lib.check(_lib.load().spam_hip_set_stream(torch.cuda.current_stream().cuda_stream))
    if which == "rmat":
        A = synth.rmat_rect(10_000_000, 160_000_000, dtype=torch.float64, device=dev, oversample=1.30)
    elif which == "banded":
        A = synth.banded(10_000_000, 17, dtype=torch.float64, device=dev)
    elif which == "powerlaw":
        A = synth.powerlaw(dtype=torch.float32, device=dev)
    x = synth.dense_vector(A.cols, dtype=A.vals.dtype, device=dev)
    y = torch.empty(A.rows, dtype=A.vals.dtype, device=dev)
    dA = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
    vs = A.vals.element_size()
    nbytes = A.nnz * (vs + 4) + (A.rows + 1) * 4 + A.cols * vs + A.rows * vs
    print(which, A.stats(), "alg bytes %.3f GB" % (nbytes / 1e9), flush=True)
    variants = [dict(SPAM_SPMV_HOT=h, SPAM_SPMV_NT=t) for h in (2, 0) for t in (1, 0)]
    extra = os.environ.get("PROBE_VARIANTS")
    if extra:
        variants = json.loads(extra)
    res = {json.dumps(v): [] for v in variants}
    for r in range(rounds):
        for v in variants:
            for k in [k for k in os.environ if k.startswith("SPAM_SPMV_")]:
                del os.environ[k]
            for k, val in v.items():
                os.environ[k] = str(val)
            dA.spmv_prepare(x.data_ptr())  # (re)build the plan with these knobs
            dA.spmv_dev(x.data_ptr(), y.data_ptr())
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                dA.spmv_dev(x.data_ptr(), y.data_ptr())
            e.record()
            torch.cuda.synchronize()
            res[json.dumps(v)].append(s.elapsed_time(e) / 10)
            if r == 0:
                print(json.dumps(v), "plan:", dA.spmv_plan_info(), flush=True)
    for k, ts in res.items():
        med = float(np.median(ts))
        print(f"{k}: median {med:.4f} ms  min {min(ts):.4f}  -> {nbytes / med / 1e6:.0f} GB/s ({nbytes / med / 1e6 / 80:.1f}% of 8 TB/s)")

main()
