import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_amd import synth
dev = torch.device("cuda", 0)
A = synth.rmat_rect(10_000_000, 160_000_000, dtype=torch.float64, device=dev, oversample=1.30)
cnt = torch.bincount(A.col_idx.to(torch.int64), minlength=A.cols)
srt, _ = torch.sort(cnt, descending=True)
cum = torch.cumsum(srt, 0).double() / A.nnz
for k in [16384, 65536, 131072, 262144, 393216, 524288, 1048576, 2097152, 4194304]:
    print(f"top {k:8d} columns ({k*8/1e6:6.2f} MB of f64): {100*cum[k-1].item():.1f}% of gathers")
# line-granular (16 doubles per 128-B line): mass of the hottest lines
lc = cnt.view(-1, 16).sum(1) if A.cols % 16 == 0 else None
if lc is not None:
    s2, _ = torch.sort(lc, descending=True); c2 = torch.cumsum(s2, 0).double() / A.nnz
    for k in [8192, 16384, 24576, 32768]:
        print(f"hottest {k} lines ({k*128/1e6:.2f} MB): {100*c2[k-1].item():.1f}%")
