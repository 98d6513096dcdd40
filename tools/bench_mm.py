#!/usr/bin/env python3
"""The reference's bench shape on the GPU path: for every Matrix Market file in a directory, parse it, build
CsrMatrix::from(dok) (sorted rows) and time m.mul_hash(&m) with unsorted output (gen_bench_mul!,
spam_csr/src/lib.rs:386-437; benches/mul_hash.rs:6-11 uses ./matrices).  Integer files run as i64 with wrapping
arithmetic, exactly as the reference benches MatrixType::Integer (lib.rs:421-424); complex files (MatrixType::Complex,
lib.rs:428-430) run as the four real products the host mirror composes a complex product from (sorted rows, so that the
four value arrays line up; the final a - b / a + b over nnz(C) values is not timed).  Prints one JSON line per file.

  python tools/bench_mm.py DIR [--reps 5]
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_amd import CsrMatrix, _lib
from sparse_matrix_amd.mmio import parse_matrix_market


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    _lib.ensure_device()
    for name in sorted(os.listdir(args.dir)):
        path = os.path.join(args.dir, name)
        if not os.path.isfile(path):
            continue  # entry.file_type()?.is_file() (lib.rs:416)
        kind, rows, cols, ents = parse_matrix_market(open(path).read())
        dtype = {"integer": np.int64, "real": np.float64, "complex": np.complex128}[kind]
        m = CsrMatrix.from_dok((rows, cols), ents, dtype=dtype)
        if kind == "complex":
            re, im = (p.device() for p in m._real_imag())
            products = [(re, re), (im, im), (re, im), (im, re)]
        else:
            products = [(m.device(), m.device())]
        ts, st = [], None
        for _ in range(args.reps + 1):
            t0 = time.perf_counter()
            outs = [x.spgemm(y, sort_rows=len(products) > 1) for x, y in products]
            _lib.check(_lib.load().spam_hip_synchronize())
            ts.append(time.perf_counter() - t0)
            st = outs[0][1]
            for C, _ in outs:
                C.free()
        t = float(np.median(ts[1:]))
        print(json.dumps({"file": name, "kind": kind, "rows": rows, "cols": cols, "nnz": m.nnz(), "flops_F": st["flops_f"],
                          "nnz_c": st["nnz_c"], "ms": round(t * 1e3, 4), "Mnnz_out_per_s": round(st["nnz_c"] / t / 1e6, 2),
                          "gflops": round(2 * st["flops_f"] / t / 1e9, 3)}))


if __name__ == "__main__":
    main()
