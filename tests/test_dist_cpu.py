"""The multi-GPU path's host logic on CPU: 2 ranks over gloo (no GPU, no RCCL).

What runs across ranks in bench.py is (1) the all-gather-v of B's CSR row blocks with row_ptr fix-up and
(2) the flop-balanced row split (rows_to_threads with tnum = world, mul_hash.rs:38-64).  Here the local
product is the CPU oracle (allowed in tests); on the GPU box it is libspam_hip.so.
"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sparse_matrix_amd import dist as spdist, synth

SCALE = 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    import oracle

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1 << SCALE
        full = synth.rmat(SCALE)
        # initial distribution: nnz-balanced row blocks (unequal row counts: the R-MAT head rows are the long ones)
        bucket = 16
        counts = (full.row_ptr[bucket::bucket] - full.row_ptr[:-bucket:bucket])
        offs0 = spdist.nnz_balanced_blocks(counts, bucket, n, world)
        assert offs0[0] == 0 and offs0[-1] == n and offs0[1] < n // 2, offs0
        shard_nnz = [int(full.row_ptr[offs0[r + 1]] - full.row_ptr[offs0[r]]) for r in range(world)]
        assert max(shard_nnz) <= 1.15 * (full.nnz / world), shard_nnz
        lo, hi = offs0[rank], offs0[rank + 1]
        blk = synth.rmat(SCALE, row_range=(lo, hi) if world > 1 else None)
        g = spdist.allgather_csr(blk.row_ptr, blk.col_idx, blk.vals, offs0, rank, world, dist, async_vals=True)
        g.wait_structure()
        g.wait_values()
        rp, col, val = g.wait()
        assert torch.equal(rp, full.row_ptr) and torch.equal(col, full.col_idx) and torch.equal(val, full.vals)
        # the exchange moves B once: no padding to the longest shard
        assert g.stats["bytes_received"] <= 1.15 * g.stats["bytes_unpadded"], g.stats
        other = full.nnz - blk.nnz
        assert g.stats["bytes_received"] == other * (4 + 8) + (n - (hi - lo)) * 8
        a = (rp.numpy().astype(np.uint64), col.numpy().astype(np.uint64), val.numpy())
        flops, offs_ref = oracle.rows_to_threads(a[0], a[1], a[0], world)
        offs = spdist.rows_to_parts_host(torch.from_numpy(flops.astype(np.int64)), world)
        assert offs == offs_ref.tolist()
        mylo, myhi = offs[rank], offs[rank + 1]
        sub = (a[0][mylo:myhi + 1] - a[0][mylo], a[1][int(a[0][mylo]):int(a[0][myhi])], a[2][int(a[0][mylo]):int(a[0][myhi])])
        o, i, v = oracle.mul_hash(sub, a, sort_rows=True, tnum=1)
        # every rank checks its block against the single-process product
        wo, wi, wv = oracle.mul_hash(a, a, sort_rows=True, tnum=1)
        s, e = int(wo[mylo]), int(wo[myhi])
        assert np.array_equal(o, wo[mylo:myhi + 1] - wo[mylo]) and np.array_equal(i, wi[s:e]) and np.array_equal(v, wv[s:e])
        # row-sharded SpMV: all-gather x, multiply the local row block, compare with the matching slice of A x
        xfull = synth.dense_vector(n)
        lo, hi = spdist.equal_row_block(n, rank, world)
        blk = synth.rmat(SCALE, row_range=(lo, hi))
        xl = xfull[lo:hi].clone()
        xg = spdist.allgather_vector(xl, n, rank, world, dist)
        assert torch.equal(xg, xfull)
        yb = oracle.spmv(blk.cpu_numpy(), xg.numpy())
        assert np.array_equal(yb, oracle.spmv(a, xfull.numpy())[lo:hi])
        # C's global row_ptr from one all-gather of per-rank nnz counts
        cnt = torch.zeros(world, dtype=torch.int64)
        cnt[rank] = len(i)
        dist.all_reduce(cnt)
        assert int(cnt.sum()) == len(wi)
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_two_rank_allgather_and_partition():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_partition_rule_edge_cases():
    """total = 0 and single-row cases of the partition_point rule (mul_hash.rs:55-62)."""
    assert spdist.rows_to_parts_host(torch.tensor([0, 0, 0]), 2) == [0, 3, 3]
    assert spdist.rows_to_parts_host(torch.tensor([5]), 3) == [0, 0, 0, 1]
    assert spdist.rows_to_parts_host(torch.tensor([1, 1, 1, 1]), 2) == [0, 2, 4]


def test_nnz_balanced_blocks_rule():
    """Block edges fall on bucket edges, are monotone, and split the entries evenly where the buckets allow."""
    c = torch.tensor([100, 1, 1, 1, 1, 1, 1, 1])
    assert spdist.nnz_balanced_blocks(c, 4, 32, 2) == [0, 4, 32]
    assert spdist.nnz_balanced_blocks(torch.ones(8, dtype=torch.int64), 4, 30, 4) == [0, 8, 16, 24, 30]
    assert spdist.nnz_balanced_blocks(torch.zeros(8, dtype=torch.int64), 4, 32, 3)[-1] == 32
