"""Row-sharded multi-GPU SpGEMM plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI; "gloo" for CPU tests and single-GPU rehearsals).

The reference is single-process; its only partition is rows_to_threads (spam_csr/src/mul_hash.rs:38-64),
contiguous row ranges balanced by intermediate products.  The same rule cuts A across GPUs (tnum = world
size).  C's rows depend on the matching rows of A and ALL of B, so the one exchange step is an all-gather-v
of B's three CSR arrays; nothing is reduced and C stays row-sharded.

Exchange.  Row shards are unequal, so the all-gather-v is an all-to-all in which every rank offers its shard to
every peer and receives each peer's shard at its final offset of the gathered array: on RCCL that is one
grouped ncclSend / ncclRecv per pair (xGMI is point-to-point, so every peer link carries exactly one shard,
concurrently), no padding and no copy afterwards — bytes received = the unpadded size of B.  The collectives are
issued asynchronously in the order row_ptr, col_idx, vals; the symbolic phase needs only the first two, so the
transfer of vals overlaps with it (`allgather_csr(..., async_vals=True)`).  Backends without all_to_all
(gloo on CPU) take one broadcast per source and array — decided from the backend before any collective is
issued, never by catching an error in the middle of one.

Initial distribution.  With equal row blocks an R-MAT shard holds up to 44 % of the entries (the first eighth of
the rows, a + b = 0.76 per level); `nnz_balanced_blocks` cuts the rows so that every rank starts with about the
same number of entries, which is what bounds the slowest link of the exchange.

Everything here works on torch tensors of any device, so the logic is exercised on CPU with gloo
(tests/test_dist_cpu.py).
"""
from __future__ import annotations

import ctypes as C

import torch


def equal_row_block(n_rows: int, rank: int, world: int) -> tuple[int, int]:
    """Rank r holds rows [r*n/world, (r+1)*n/world)."""
    return rank * n_rows // world, (rank + 1) * n_rows // world


def equal_row_blocks(n_rows: int, world: int) -> list[int]:
    return [r * n_rows // world for r in range(world)] + [n_rows]


def nnz_balanced_blocks(bucket_nnz, bucket_rows: int, n_rows: int, world: int) -> list[int]:
    """Row offsets [0, ..., n_rows] of `world` contiguous blocks holding about the same number of entries each;
    bucket_nnz[i] = entries in rows [i*bucket_rows, (i+1)*bucket_rows) (block edges fall on bucket edges)."""
    c = torch.as_tensor(bucket_nnz).to(torch.int64).cpu()
    ps = torch.zeros(c.numel() + 1, dtype=torch.int64)
    torch.cumsum(c, 0, out=ps[1:])
    total = int(ps[-1].item())
    offs = [0]
    for t in range(1, world):
        b = int(torch.searchsorted(ps, torch.tensor([total * t // world], dtype=torch.int64), right=False).item())
        b = max(offs[-1] // bucket_rows, min(b, c.numel()))
        offs.append(min(b * bucket_rows, n_rows))
    offs.append(n_rows)
    return offs


def _has_all_to_all(dist) -> bool:
    """Capability from the backend name, agreed by construction on every rank (no probing collective)."""
    return str(dist.get_backend()).lower() == "nccl"


class Gathered:
    """Result of allgather_csr: the full arrays plus what the exchange moved."""

    def __init__(self, rp, col, val, pending, stats):
        self.row_ptr, self.col_idx, self.vals = rp, col, val
        self._pending = pending
        self.stats = stats

    def wait_structure(self):
        """row_ptr and col_idx have arrived (the current stream waits; the host does not block on NCCL)."""
        for w in self._pending[:2]:
            if w is not None:
                w.wait()
        self._pending[0] = self._pending[1] = None

    def wait_values(self):
        w = self._pending[2]
        if w is not None:
            w.wait()
        self._pending[2] = None

    def wait(self):
        self.wait_structure()
        self.wait_values()
        return self.row_ptr, self.col_idx, self.vals


def allgather_csr(row_ptr: torch.Tensor, col_idx: torch.Tensor, vals: torch.Tensor, row_offsets: list[int], rank: int,
                  world: int, dist=None, async_vals: bool = False, force_collectives: bool = False) -> Gathered:
    """All-gather-v of row-block-distributed CSR shards into the full matrix on every rank.

    row_offsets[r] .. row_offsets[r+1] = the rows rank r holds (any contiguous split); row_ptr int64[rows_local+1]
    (block-local offsets starting at 0), col_idx int32[nnz_local], vals[nnz_local].  Returns a `Gathered` whose
    row_ptr is globalised by each source's entry offset.  With async_vals the caller gets control back while vals
    is still in flight: call wait_structure(), run the symbolic phase, then wait_values().
    force_collectives: issue the collectives in a world of one as well (a 1-GPU box can then run the RCCL calls —
    all_reduce, grouped send/recv to self — that the 8-GPU run makes; tests/test_rccl_world1_gpu.py).
    """
    dev = col_idx.device
    n_rows = row_offsets[-1]
    if (world == 1 and not force_collectives) or dist is None:
        nb = int(col_idx.numel()) * (col_idx.element_size() + vals.element_size()) + (n_rows + 1) * 8
        return Gathered(row_ptr, col_idx, vals, [None, None, None], {"bytes_received": 0, "bytes_unpadded": nb, "mode": "local"})
    nnz_local = int(col_idx.numel())
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    sizes[rank] = nnz_local
    dist.all_reduce(sizes)
    sizes_h = sizes.cpu().tolist()
    starts_h = [0] * (world + 1)
    for r in range(world):
        starts_h[r + 1] = starts_h[r] + sizes_h[r]
    nnz_tot = starts_h[world]
    col = torch.empty(nnz_tot, dtype=col_idx.dtype, device=dev)
    val = torch.empty(nnz_tot, dtype=vals.dtype, device=dev)
    rp = torch.empty(n_rows + 1, dtype=torch.int64, device=dev)
    lo, hi = row_offsets[rank], row_offsets[rank + 1]
    rp[n_rows] = nnz_tot
    my_rp = row_ptr[:-1] + starts_h[rank]  # row_ptr fix-up by the source rank's global entry offset
    other_rows = n_rows - (hi - lo)
    other_nnz = nnz_tot - nnz_local
    stats = {"bytes_received": other_rows * 8 + other_nnz * (col.element_size() + val.element_size()),
             "bytes_unpadded": other_rows * 8 + other_nnz * (col.element_size() + val.element_size())}
    pending = [None, None, None]
    if _has_all_to_all(dist):
        stats["mode"] = "all_to_all (grouped send/recv, one shard per peer link)"

        def a2a(out_full, bounds, mine):
            outs = [out_full[bounds[r]:bounds[r + 1]] for r in range(world)]
            return dist.all_to_all(outs, [mine] * world, async_op=True)

        pending[0] = a2a(rp, row_offsets, my_rp.contiguous())
        pending[1] = a2a(col, starts_h, col_idx)
        pending[2] = a2a(val, starts_h, vals)
    else:
        stats["mode"] = "broadcast per source (backend without all_to_all)"
        rp[lo:hi] = my_rp
        col[starts_h[rank]:starts_h[rank + 1]] = col_idx
        val[starts_h[rank]:starts_h[rank + 1]] = vals
        for arr, bounds in ((rp, row_offsets), (col, starts_h), (val, starts_h)):
            for r in range(world):  # unequal shards: one broadcast per source and array
                if bounds[r + 1] > bounds[r]:
                    dist.broadcast(arr[bounds[r]:bounds[r + 1]], src=r)
    g = Gathered(rp, col, val, pending, stats)
    if not async_vals:
        g.wait()
    return g


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


ROW_OVERHEAD_PRODUCTS = 4000  # measured on MI355X (scale-20 A*A): a non-empty row costs about as much as 4000 products — at 128
                              # column chunks of B; the library scales it with the width of B (SPAM_ROW_OVERHEAD_AUTO)
ROW_OVERHEAD_AUTO = 0xFFFFFFFFFFFFFFFF


def balanced_row_blocks(row_flops, nparts: int, row_overhead: int = ROW_OVERHEAD_PRODUCTS) -> list[int]:
    """The reference's partition rule (rows_to_threads, mul_hash.rs:51-62) applied to a cost that adds a fixed
    per-row term to the intermediate-product count: blocks of many light rows otherwise run ~20 % longer than
    blocks of few heavy rows with the same F (SURVEY.md §8e "cost-weighted partition")."""
    f = torch.as_tensor(row_flops).to(torch.int64).cpu()
    return rows_to_parts_host(f + row_overhead * (f > 0).to(torch.int64), nparts)


def allgather_vector(x_local: torch.Tensor, n: int, rank: int, world: int, dist=None,
                     force_collectives: bool = False) -> torch.Tensor:
    """Row-sharded SpMV needs all of x: every rank owns x[equal_row_block] and receives the rest (one padded
    all-gather; 8 B * n / world per peer link; equal blocks differ by at most one element)."""
    if (world == 1 and not force_collectives) or dist is None:
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


class RcclComm:
    """The exchange behind the C ABI (include/spam_hip_rccl.h, libspam_hip_rccl.so): what a Rust host calls.  This class
    only moves pointers; the communicator, the exchange stream, the grouped ncclSend / ncclRecv, the row_ptr fix-up and the
    event that holds back readers of the values all live in the library.  torch.distributed (any backend) is used for
    ONE thing here: handing rank 0's 128-byte unique id to the other ranks."""

    def __init__(self, rank: int, world: int, unique_id: bytes):
        from . import _lib

        _lib.ensure_device()
        self._L = _lib.load_rccl()
        self._check = _lib.check
        self.rank, self.world = rank, world
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), _lib.RCCL_ID_BYTES)
        _lib.check(self._L.spam_rccl_comm_create(buf, rank, world, C.byref(h)))
        self._h = h

    @staticmethod
    def unique_id() -> bytes:
        from . import _lib

        buf = C.create_string_buffer(_lib.RCCL_ID_BYTES)
        _lib.check(_lib.load_rccl().spam_rccl_unique_id(buf))
        return buf.raw

    @classmethod
    def from_torch_dist(cls, dist, rank: int, world: int) -> "RcclComm":
        """Rank 0 draws the id; the process group (gloo or nccl) broadcasts its 128 bytes."""
        box = [cls.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        return cls(rank, world, box[0])

    def allgather_csr(self, shard, row_offsets):
        """shard: DeviceCsr of this rank's rows; returns the gathered DeviceCsr (values possibly still in flight: only
        kernels that read them wait) and the bytes this rank received."""
        from .csr import DeviceCsr

        offs = (C.c_uint64 * (self.world + 1))(*[int(v) for v in row_offsets])
        h = C.c_void_p()
        self._check(self._L.spam_csr_allgather_rccl(self._h, shard._h, offs, C.byref(h)))
        return DeviceCsr(h), self.last_exchange()[0]

    def allgather_vector(self, x_local: torch.Tensor, counts) -> torch.Tensor:
        cnt = (C.c_uint64 * self.world)(*[int(v) for v in counts])
        out = torch.empty(int(sum(counts)), dtype=x_local.dtype, device=x_local.device)
        self._check(self._L.spam_vector_allgather_rccl(self._h, x_local.element_size(), C.c_void_p(x_local.data_ptr()), cnt,
                                                       C.c_void_p(out.data_ptr())))
        return out

    def rank_step(self, a_full, b_full, count_offsets=None, row_overhead: int = ROW_OVERHEAD_AUTO, sort_rows=False):
        """spam_spgemm_rank_step_rccl: sharded flop count + all-gather of the counts + partition + this rank's block of
        C.  Returns (C block or None, stats dict, (row_lo, row_hi))."""
        from ._lib import SpgemmStats
        from .csr import DeviceCsr

        co = (C.c_uint64 * (self.world + 1))(*[int(v) for v in count_offsets]) if count_offsets is not None else None
        h = C.c_void_p()
        st = SpgemmStats()
        lo, hi = C.c_uint64(), C.c_uint64()
        self._check(self._L.spam_spgemm_rank_step_rccl(self._h, a_full._h, b_full._h, co, row_overhead, int(sort_rows),
                                                       C.byref(h), C.byref(lo), C.byref(hi), C.byref(st)))
        return (DeviceCsr(h) if h.value else None), st.asdict(), (lo.value, hi.value)

    def last_exchange(self):
        r, s = C.c_uint64(), C.c_uint64()
        self._check(self._L.spam_rccl_last_exchange(self._h, C.byref(r), C.byref(s)))
        return r.value, s.value

    def free(self):
        if self._h:
            self._L.spam_rccl_comm_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
