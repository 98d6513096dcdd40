"""Row-sharded multi-GPU SpGEMM plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI; "gloo" for CPU tests and single-GPU rehearsals).

The reference is single-process; its only partition is rows_to_threads (spam_csr/src/mul_hash.rs:38-64),
contiguous row ranges balanced by intermediate products.  The same rule cuts A across GPUs (tnum = world
size).  C's rows depend on the matching rows of A and ALL of B, so the one exchange step is an all-gather
of B's CSR arrays (row blocks are unequal: shards are padded to the longest and moved by one all-gather per array);
nothing is reduced and C stays row-sharded.  Everything here works on torch tensors of any device, so the
logic is exercised on CPU with gloo (tests/test_dist_cpu.py).
"""
from __future__ import annotations

import torch


def equal_row_block(n_rows: int, rank: int, world: int) -> tuple[int, int]:
    """Initial distribution: rank r generates / holds rows [r*n/world, (r+1)*n/world)."""
    return rank * n_rows // world, (rank + 1) * n_rows // world


def allgather_csr(row_ptr: torch.Tensor, col_idx: torch.Tensor, vals: torch.Tensor, n_rows: int, rank: int,
                  world: int, dist=None):
    """All-gather-v of equal_row_block-distributed CSR blocks into the full matrix on every rank.

    row_ptr int64[rows_local+1] (block-local offsets starting at 0), col_idx int32[nnz_local], vals[nnz_local].
    Returns (row_ptr int64[n_rows+1], col_idx, vals) with row_ptr globalised by each source's offset.
    Traffic per rank: receives nnz_total*(4+sizeof T) + n_rows*8 bytes, sends its own block once per peer.
    """
    dev = col_idx.device
    if world == 1 or dist is None:
        return row_ptr, col_idx, vals
    nnz_local = int(col_idx.numel())
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    sizes[rank] = nnz_local
    dist.all_reduce(sizes)
    sizes_h = sizes.cpu().tolist()
    starts_h = [0] * world
    for r in range(1, world):
        starts_h[r] = starts_h[r - 1] + sizes_h[r - 1]
    nnz_tot = starts_h[-1] + sizes_h[-1]
    col = torch.empty(nnz_tot, dtype=col_idx.dtype, device=dev)
    val = torch.empty(nnz_tot, dtype=vals.dtype, device=dev)
    rp = torch.empty(n_rows + 1, dtype=torch.int64, device=dev)
    lo, hi = equal_row_block(n_rows, rank, world)
    s0 = starts_h[rank]
    rp[n_rows] = nnz_tot
    try:
        # one padded all-gather per array (3 collectives: on xGMI every peer link carries one shard concurrently),
        # then each shard is copied to its place; falls back to per-source broadcasts where the backend lacks it
        maxn = max(sizes_h)
        maxr = max(equal_row_block(n_rows, r, world)[1] - equal_row_block(n_rows, r, world)[0] for r in range(world))
        pc = torch.zeros(maxn, dtype=col_idx.dtype, device=dev)
        pv = torch.zeros(maxn, dtype=vals.dtype, device=dev)
        pr = torch.zeros(maxr, dtype=torch.int64, device=dev)
        pc[:nnz_local] = col_idx
        pv[:nnz_local] = vals
        pr[:hi - lo] = row_ptr[:-1] + s0  # row_ptr fix-up by the source rank's global nnz offset
        gc = torch.empty(world * maxn, dtype=col_idx.dtype, device=dev)
        gv = torch.empty(world * maxn, dtype=vals.dtype, device=dev)
        gr = torch.empty(world * maxr, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(gc, pc)
        dist.all_gather_into_tensor(gv, pv)
        dist.all_gather_into_tensor(gr, pr)
        for r in range(world):
            r0, r1 = equal_row_block(n_rows, r, world)
            a0, a1 = starts_h[r], starts_h[r] + sizes_h[r]
            col[a0:a1] = gc[r * maxn:r * maxn + sizes_h[r]]
            val[a0:a1] = gv[r * maxn:r * maxn + sizes_h[r]]
            rp[r0:r1] = gr[r * maxr:r * maxr + (r1 - r0)]
    except (RuntimeError, NotImplementedError):
        col[s0:s0 + nnz_local] = col_idx
        val[s0:s0 + nnz_local] = vals
        rp[lo:hi] = row_ptr[:-1] + s0
        for r in range(world):  # unequal shards: one broadcast per source and array
            r0, r1 = equal_row_block(n_rows, r, world)
            a0, a1 = starts_h[r], starts_h[r] + sizes_h[r]
            if a1 > a0:
                dist.broadcast(col[a0:a1], src=r)
                dist.broadcast(val[a0:a1], src=r)
            if r1 > r0:
                dist.broadcast(rp[r0:r1], src=r)
    return rp, col, val


def rows_to_parts_host(row_flops: torch.Tensor, nparts: int) -> list[int]:
    """rows_to_threads' partition rule on a vector of per-row intermediate-product counts
    (mul_hash.rs:51-62): ps = [0, cumsum], average = ceil(total / nparts),
    offsets[t] = partition_point(ps, x <= average * t) - 1."""
    f = row_flops.to(torch.int64).cpu()
    ps = torch.zeros(f.numel() + 1, dtype=torch.int64)
    torch.cumsum(f, 0, out=ps[1:])
    total = int(ps[-1].item())
    avg = -(-total // nparts)
    offs = [0]
    for t in range(1, nparts):
        pp = int(torch.searchsorted(ps, torch.tensor([avg * t], dtype=torch.int64), right=True).item())
        offs.append(pp - 1)
    offs.append(f.numel())
    return offs


ROW_OVERHEAD_PRODUCTS = 2300  # measured on MI355X (scale-20 A*A): a non-empty row costs about as much as 2300 products


def balanced_row_blocks(row_flops, nparts: int, row_overhead: int = ROW_OVERHEAD_PRODUCTS) -> list[int]:
    """The reference's partition rule (rows_to_threads, mul_hash.rs:51-62) applied to a cost that adds a fixed
    per-row term to the intermediate-product count: blocks of many light rows otherwise run ~20 % longer than
    blocks of few heavy rows with the same F (SURVEY.md §8e "cost-weighted partition")."""
    f = torch.as_tensor(row_flops).to(torch.int64).cpu()
    return rows_to_parts_host(f + row_overhead * (f > 0).to(torch.int64), nparts)


def allgather_vector(x_local: torch.Tensor, n: int, rank: int, world: int, dist=None) -> torch.Tensor:
    """Row-sharded SpMV needs all of x: every rank owns x[equal_row_block] and receives the rest (one padded
    all-gather; 8 B * n / world per peer link)."""
    if world == 1 or dist is None:
        return x_local
    blocks = [equal_row_block(n, r, world) for r in range(world)]
    maxb = max(hi - lo for lo, hi in blocks)
    pad = torch.zeros(maxb, dtype=x_local.dtype, device=x_local.device)
    pad[: x_local.numel()] = x_local
    out = torch.empty(world * maxb, dtype=x_local.dtype, device=x_local.device)
    dist.all_gather_into_tensor(out, pad)
    x = torch.empty(n, dtype=x_local.dtype, device=x_local.device)
    for r, (lo, hi) in enumerate(blocks):
        x[lo:hi] = out[r * maxb: r * maxb + (hi - lo)]
    return x
