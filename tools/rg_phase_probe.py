import os, sys, torch
sys.path.insert(0, '/root/repo')
from sparse_matrix_amd import DeviceCsr, _lib, synth
dev = torch.device("cuda", 0); _lib.ensure_device(0)
A = synth.rmat_rect(10_000_000, 160_000_000, dtype=torch.float64, device=dev, oversample=1.30)
x = synth.dense_vector(A.cols, dtype=torch.float64, device=dev); y = torch.empty(A.rows, dtype=torch.float64, device=dev)
dA = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
os.environ["SPAM_SPMV_SLICED"] = "2"; os.environ["SPAM_SPMV_PROF"] = os.environ.get("RG_PROBE_PROF", "3")
dA.spmv_prepare(x.data_ptr())
for _ in range(2):
    dA.spmv_dev(x.data_ptr(), y.data_ptr()); torch.cuda.synchronize()
