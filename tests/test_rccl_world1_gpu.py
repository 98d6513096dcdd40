"""The RCCL calls of the multi-GPU path, run for real on the one GPU a test box has.

The 8-GPU run (BASELINE configs[3]) is the driver's to launch; what a 1-GPU box CAN check is that the collectives
dist.py issues on the "nccl" backend — all_reduce of the shard sizes, the all-gather-v of B as grouped send / recv
(`all_to_all` over views of the gathered arrays, asynchronous, vals overlapped with the symbolic phase) and
all_gather_into_tensor for x — are accepted by RCCL with these tensor types and shapes and deliver the data.  A process
group of ONE rank does that: every send meets its receive on the same device.  (Two ranks on one GPU are refused by
RCCL, which is why the N > 1 rehearsals in profiles/ use gloo.)  Runs in a child process: the process group is global
state.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys, socket
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.getcwd())
from sparse_matrix_amd import DeviceCsr, _lib, synth
from sparse_matrix_amd import dist as spdist

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert str(dist.get_backend()).lower() == "nccl"
_lib.ensure_device(0)
_lib.check(_lib.load().spam_hip_set_stream(torch.cuda.current_stream().cuda_stream))
for dtype in (torch.float64, torch.float32, torch.int64):
    m = synth.rmat(12, dtype=torch.float64, device=dev)
    vals = m.vals.to(dtype) if dtype != torch.int64 else (m.vals * 1000).to(torch.int64)
    n = m.rows
    g = spdist.allgather_csr(m.row_ptr, m.col_idx, vals, [0, n], 0, 1, dist, async_vals=True, force_collectives=True)
    assert g.stats["mode"].startswith("all_to_all"), g.stats
    g.wait_structure()
    B = DeviceCsr.from_torch(n, m.cols, g.row_ptr, g.col_idx, g.vals, True)
    A = DeviceCsr.from_torch(n, m.cols, m.row_ptr, m.col_idx, vals, True)
    plan, nnz = A.spgemm_symbolic(B)          # structure only: vals may still be in flight
    g.wait_values()
    torch.cuda.synchronize()
    assert torch.equal(g.row_ptr, m.row_ptr) and torch.equal(g.col_idx, m.col_idx) and torch.equal(g.vals, vals)
    C, st = DeviceCsr.spgemm_numeric(plan, sort_rows=True)
    C2, _ = A.spgemm(A, sort_rows=True)
    o1, i1, v1 = C.download(); o2, i2, v2 = C2.download()
    assert np.array_equal(o1, o2) and np.array_equal(i1, i2)
    assert np.array_equal(v1, v2) if dtype == torch.int64 else np.allclose(v1, v2, rtol=1e-5 if dtype == torch.float32 else 1e-12)
    x = synth.dense_vector(m.cols, dtype=torch.float64, device=dev).to(dtype)
    xg = spdist.allgather_vector(x, m.cols, 0, 1, dist, force_collectives=True)
    assert xg.data_ptr() != x.data_ptr() and torch.equal(xg, x)
cnt = torch.tensor([7], dtype=torch.int64, device=dev)
dist.all_reduce(cnt)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert int(cnt.item()) == 7 and float(t.item()) == 1.5
dist.destroy_process_group()
print("rccl world-1 ok")
"""


def test_rccl_collectives_in_a_world_of_one():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", CHILD], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl world-1 ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
