#!/usr/bin/env python3
"""How the products of C = A*A (R-MAT) spread over (row, column chunk) cells of different density.  Diagnostic only:
answers whether storing the symbolic phase's bitmap for dense cells (so that the numeric phase need not recompute the
pattern there) would cover enough of the work.  Sampled heavy rows, products enumerated on the host."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_amd import DeviceCsr, _lib, synth

CW = 8192
REGION = 512 * 144


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    nsample = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    dev = torch.device("cuda", 0)
    _lib.ensure_device(0)
    _lib.check(_lib.load().spam_hip_set_stream(torch.cuda.current_stream().cuda_stream))
    A = synth.rmat(scale, dtype=torch.float64, device=dev)
    dA = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
    C, st = dA.spgemm(dA, sort_rows=False)
    nzr = np.diff(C.download_row_ptr().astype(np.int64))
    C.free()
    rp = A.row_ptr.cpu().numpy().astype(np.int64)
    ci = A.col_idx.cpu().numpy().astype(np.int64)
    heavy = np.nonzero(nzr > 4096)[0]
    print(f"scale {scale}: heavy rows {len(heavy)} of {A.rows}, their entries {nzr[heavy].sum():.3e} of {nzr.sum():.3e}", flush=True)
    rng = np.random.default_rng(1)
    rows = rng.choice(heavy, size=min(nsample, len(heavy)), replace=False)
    thr = [128, 256, 512, 1024, 2048]
    nch = (A.cols + CW - 1) // CW
    tot_f = 0
    f_cell = {t: 0 for t in thr}      # products in cells with cnt >= t
    f_group = {t: 0 for t in thr}     # products in groups all of whose non-empty cells have cnt >= t
    n_cell = {t: 0 for t in thr}      # such cells (1 KiB of bitmap each)
    n_groups = 0
    for i in rows:
        ks = ci[rp[i]:rp[i + 1]]
        cols = np.concatenate([ci[rp[k]:rp[k + 1]] for k in ks])
        fch = np.bincount(cols >> 13, minlength=nch)
        cch = np.bincount(np.unique(cols) >> 13, minlength=nch)
        tot_f += fch.sum()
        for t in thr:
            m = cch >= t
            f_cell[t] += fch[m].sum()
            n_cell[t] += int(m.sum())
        # greedy groups: 8 bytes per 32 columns of span + 8 per entry <= REGION
        j = 0
        while j < nch:
            if cch[j] == 0:
                j += 1
                continue
            j0, cnt = j, 0
            while j < nch and (j + 1 - j0) * (CW // 32) * 8 + (cnt + cch[j]) * 8 <= REGION:
                cnt += cch[j]
                j += 1
            if j == j0:
                raise SystemExit("chunk does not fit")
            n_groups += 1
            sel = cch[j0:j] > 0
            for t in thr:
                if np.all(cch[j0:j][sel] >= t):
                    f_group[t] += fch[j0:j].sum()
    scale_up = len(heavy) / len(rows)
    print(f"sampled {len(rows)} rows: products {tot_f:.3e}, groups {n_groups} (x{scale_up:.0f} -> {n_groups * scale_up:.3e})")
    for t in thr:
        print(f"  cnt >= {t:5d}: products in such cells {f_cell[t] / tot_f:6.1%}, in fully covered groups {f_group[t] / tot_f:6.1%}, "
              f"bitmap pool {n_cell[t] * scale_up * CW / 8 / 2**30:.2f} GiB")


main()
