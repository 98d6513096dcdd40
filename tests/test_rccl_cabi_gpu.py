"""The multi-GPU exchange BEHIND THE C ABI (include/spam_hip_rccl.h, libspam_hip_rccl.so), run for real on the one GPU a
test box has: a communicator of ONE rank made with ncclCommInitRank from a 128-byte unique id, the grouped
ncclSend / ncclRecv all-gather-v of B's three arrays with the row_ptr fix-up on the device and the values still in
flight behind an event, the sharded flop count + all-gather of the per-row counts + partition (rows_to_threads,
/root/reference/spam_csr/src/mul_hash.rs:38-64 with tnum = ranks) + this rank's block of C, and the all-gather of x.  No
torch.distributed anywhere: the library does the exchange, Python only passes pointers — as the Rust host would.
(Two ranks on one GPU are refused by RCCL; N > 1 is the driver's 8-GPU run.  The partition / exchange LOGIC for N > 1 is
covered on CPU by tests/test_dist_cpu.py and by the emulated ranks below.)  Child processes: a communicator is
process-global state in RCCL.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import oracle
from sparse_matrix_amd import DeviceCsr, _lib, synth
from sparse_matrix_amd.dist import RcclComm

assert "torch.distributed" not in sys.modules or not torch.distributed.is_initialized()
dev = torch.device("cuda", 0)
_lib.ensure_device(0)
comm = RcclComm(0, 1, RcclComm.unique_id())
for dtype in (torch.float64, torch.float32, torch.int64):
    m = synth.rmat(12, dtype=torch.float64, device=dev)
    vals = m.vals.to(dtype) if dtype != torch.int64 else ((m.vals - 1.0) * 4000).to(torch.int64)
    n = m.rows
    shard = DeviceCsr.from_torch(n, m.cols, m.row_ptr, m.col_idx, vals, True)
    B, received = comm.allgather_csr(shard, [0, n])
    assert received == 0  # a world of one receives nothing from OTHER ranks; the self send / recv still ran
    # structure is usable at once: partition + symbolic phase read row_ptr / col_idx only
    flops, offs = B.rows_to_parts(B, 4)
    Cb, st, (lo, hi) = comm.rank_step(B, B, None, sort_rows=True)  # row overhead: SPAM_ROW_OVERHEAD_AUTO
    assert (lo, hi) == (0, n) and st["rows"] == n
    go, gi, gv = B.download()          # waits for the values on the device
    ho, hi_, hv = shard.download()
    assert np.array_equal(go, ho) and np.array_equal(gi, hi_) and np.array_equal(gv, hv), "gathered B != shard"
    host = (ho, hi_, hv)
    wo, wi, wv = oracle.mul_hash(host, host, sort_rows=True)
    co, ci, cv = Cb.download()
    assert np.array_equal(co, wo) and np.array_equal(ci, wi), "C structure vs oracle"
    if dtype == torch.int64:
        assert np.array_equal(cv, wv), "Wrapping<i64> values must be bit-exact"
    else:
        tol = 1e-10 if dtype == torch.float64 else 1e-5
        assert float(np.max(np.abs(cv - wv) / np.abs(wv))) <= tol
    of, _ = oracle.rows_to_threads(ho, hi_, ho, 4)
    assert np.array_equal(np.asarray(flops, dtype=np.uint64), np.asarray(of, dtype=np.uint64))
    # a gathered handle is an ordinary matrix: SpMV, add, row blocks, a second product
    x = synth.dense_vector(m.cols, dtype=torch.float64, device=dev).to(dtype)
    xg = comm.allgather_vector(x, [m.cols])
    torch.cuda.synchronize(); _lib.check(_lib.load().spam_hip_synchronize())
    assert xg.data_ptr() != x.data_ptr() and torch.equal(xg, x)
    y = B.spmv(x.cpu().numpy())
    yr = oracle.spmv(host, x.cpu().numpy())
    assert np.array_equal(y, yr) if dtype == torch.int64 else np.allclose(y, yr, rtol=1e-5 if dtype == torch.float32 else 1e-12)
    # emulated ranks: the blocks of a 3-rank world computed one after another from the gathered B tile C exactly
    pieces = []
    for r in range(3):
        blk, _, (l, h) = B.rank_step(B, r, 3, 4000, sort_rows=True)
        pieces.append((l, h, blk.download() if blk is not None else None))
    assert pieces[0][0] == 0 and pieces[-1][1] == n and all(pieces[i][1] == pieces[i + 1][0] for i in range(2))
    cat_idx = np.concatenate([p[2][1] for p in pieces if p[2] is not None])
    assert np.array_equal(cat_idx, wi)
    B.free(); shard.free(); Cb.free()
comm.free()
print("rccl c-abi world-1 ok")
"""


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


@pytest.mark.gpu
def test_exchange_and_rank_step_through_the_c_abi_in_a_world_of_one():
    r = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl c-abi world-1 ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def _build_c_host(tmp_path):
    exe = str(tmp_path / "c_rccl_host")
    lib_dir = os.path.join(ROOT, "sparse_matrix_amd")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_rccl_host.c"), "-o", exe, "-L", lib_dir, "-lspam_hip_rccl", "-lspam_hip",
                           f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"])
    return exe


def test_plain_c_host_compiles_and_links(tmp_path):
    """No GPU needed: both headers are valid C11 and every entry point the multi-GPU call sequence uses resolves."""
    assert os.path.exists(_build_c_host(tmp_path))


@pytest.mark.gpu
def test_plain_c_host_runs_the_rank_sequence(tmp_path):
    out = subprocess.run([_build_c_host(tmp_path)], capture_output=True, text=True, timeout=300, env=_env())
    assert out.returncode == 0 and "c rccl host ok" in out.stdout, out.stdout + out.stderr
