"""Synthetic CSR generators for the spam_csr hot path.

The reference ships no matrices (``matrices/`` is git-ignored, /root/reference TODO.md:1-2), so the
benchmark inputs are generated here.  Everything is a pure function of (seed, index) built on the
splitmix64 finaliser, written with torch integer ops so the SAME code produces the SAME matrix on the
CPU (tests, oracle) and on the GPU (bench): no host RNG state is involved.

Generators (all return CSR with strictly increasing columns per row, i.e. the layout
``CsrMatrix::<T, true>::from(DokMatrix)`` produces, spam_csr/src/lib.rs:315-334):

* ``banded(n, bw)``              G-band: diagonals -bw//2 .. +bw//2
* ``rmat(scale, ...)``           G-rmat: Graph500 R-MAT (a,b,c,d = .57,.19,.19,.05), duplicates merged
* ``rmat_rect(n, nnz_target)``   G-rmat clipped to n x n (cfg2: 10 M rows / 160 M nnz)
* ``powerlaw(n, heavy_frac, heavy_nnz, light_nnz)``   G-powerlaw (cfg5)

Values are ``0.5 + u01(hash(row, col))`` in [0.5, 1.5): a function of the coordinate only, so merging
duplicates needs no keep-first rule, and there is no cancellation (pure relative tolerances are
meaningful).  ``signed=True`` maps to [-1, 1) instead (checked with the Higham bound).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

_M64 = (1 << 64) - 1


def _i64(x: int) -> int:
    """Wrap an unsigned 64-bit constant into python int representable as torch.int64."""
    x &= _M64
    return x - (1 << 64) if x >= (1 << 63) else x


_GOLD = _i64(0x9E3779B97F4A7C15)
_C1 = _i64(0xBF58476D1CE4E5B9)
_C2 = _i64(0x94D049BB133111EB)


def _lsr(z: torch.Tensor, s: int) -> torch.Tensor:
    """Logical shift right on int64 (torch only has the arithmetic one)."""
    return (z >> s) & ((1 << (64 - s)) - 1)


def mix64(z: torch.Tensor) -> torch.Tensor:
    """splitmix64 finaliser on int64 tensors (two's complement wrap == u64 arithmetic)."""
    z = z + _GOLD
    z = (z ^ _lsr(z, 30)) * _C1
    z = (z ^ _lsr(z, 27)) * _C2
    return z ^ _lsr(z, 31)


def mix64_int(z: int) -> int:
    """Same function on a python int (u64)."""
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def u01(h: torch.Tensor, dtype=torch.float64) -> torch.Tensor:
    """Top 53 bits of a 64-bit hash as a double in [0, 1)."""
    return (_lsr(h, 11).to(torch.float64) * (1.0 / 9007199254740992.0)).to(dtype)


@dataclass
class Csr:
    """Plain CSR triple on one torch device.  row_ptr int64[rows+1], col_idx int32[nnz], vals[nnz]."""

    rows: int
    cols: int
    row_ptr: torch.Tensor
    col_idx: torch.Tensor
    vals: torch.Tensor

    @property
    def nnz(self) -> int:
        return int(self.col_idx.numel())

    def stats(self) -> dict:
        d = (self.row_ptr[1:] - self.row_ptr[:-1])
        return {
            "rows": self.rows,
            "cols": self.cols,
            "nnz": self.nnz,
            "avg_nnz_row": self.nnz / max(self.rows, 1),
            "max_nnz_row": int(d.max().item()) if self.rows else 0,
        }

    def cpu_numpy(self):
        """(row_ptr u64, col_idx u64, vals) numpy arrays in the reference's host layout
        (Vec<usize> offsets / indices, spam_csr/src/lib.rs:29-31)."""
        import numpy as np

        return (
            self.row_ptr.cpu().numpy().astype(np.uint64),
            self.col_idx.cpu().numpy().astype(np.uint64),
            self.vals.cpu().numpy(),
        )


def _values(keys: torch.Tensor, seed: int, dtype, signed: bool) -> torch.Tensor:
    u = u01(mix64(keys ^ _i64(mix64_int(seed ^ 0x5851F42D4C957F2D))), torch.float64)
    v = (2.0 * u - 1.0) if signed else (0.5 + u)
    return v.to(dtype)


def _csr_from_keys(keys: torch.Tensor, rows: int, cols: int, seed: int, dtype, signed: bool) -> Csr:
    """keys = row << 32 | col, any order, duplicates allowed -> sorted, deduplicated CSR."""
    keys = torch.unique(keys, sorted=True)
    r = keys >> 32
    c = (keys & 0xFFFFFFFF).to(torch.int32)
    counts = torch.bincount(r, minlength=rows)
    row_ptr = torch.zeros(rows + 1, dtype=torch.int64, device=keys.device)
    torch.cumsum(counts, 0, out=row_ptr[1:])
    vals = _values(keys, seed, dtype, signed)
    return Csr(rows, cols, row_ptr, c, vals)


def banded(n: int, bw: int = 5, dtype=torch.float64, device="cpu", seed: int = 1, signed=False) -> Csr:
    """G-band(n, bw): nnz = bw*n - 2*sum(1..bw//2)  (n=10000, bw=5 -> 49 994)."""
    h = bw // 2
    i = torch.arange(n, dtype=torch.int64, device=device).repeat_interleave(bw)
    d = torch.arange(-h, h + 1, dtype=torch.int64, device=device).repeat(n)
    j = i + d
    ok = (j >= 0) & (j < n)
    keys = (i[ok] << 32) | j[ok]
    return _csr_from_keys(keys, n, n, seed, dtype, signed)


_RMAT_T = (int(0.57 * 65536), int(0.76 * 65536), int(0.95 * 65536))  # a | a+b | a+b+c on 16 bits


def _rmat_edges(e: torch.Tensor, scale: int, seed: int):
    """Edge ids -> (row, col) by `scale` quadrant choices, 16 hash bits per level."""
    r = torch.zeros_like(e)
    c = torch.zeros_like(e)
    s0 = _i64(mix64_int(seed))
    nwords = (scale + 3) // 4
    lvl = 0
    for w in range(nwords):
        h = mix64(e * nwords + w + s0)
        for q in range(4):
            if lvl >= scale:
                break
            u = _lsr(h, 16 * q) & 0xFFFF
            # quadrant: a -> (0,0), b -> (0,1), c -> (1,0), d -> (1,1)
            rb = (u >= _RMAT_T[1]).to(torch.int64)
            cb = (((u >= _RMAT_T[0]) & (u < _RMAT_T[1])) | (u >= _RMAT_T[2])).to(torch.int64)
            r = (r << 1) | rb
            c = (c << 1) | cb
            lvl += 1
    return r, c


def rmat(scale: int, edge_factor: int = 16, seed: int = 42, dtype=torch.float64, device="cpu",
         signed=False, chunk: int = 1 << 24, row_range=None) -> Csr:
    """G-rmat(scale): 2^scale x 2^scale, edge_factor * 2^scale samples, duplicates merged.

    row_range=(lo, hi) keeps only rows in [lo, hi) and returns them as a (hi-lo) x 2^scale block
    (each GPU builds its own row block of A without materialising the rest).
    """
    n = 1 << scale
    m = edge_factor * n
    parts = []
    for s in range(0, m, chunk):
        e = torch.arange(s, min(s + chunk, m), dtype=torch.int64, device=device)
        r, c = _rmat_edges(e, scale, seed)
        if row_range is not None:
            ok = (r >= row_range[0]) & (r < row_range[1])
            r, c = r[ok] - row_range[0], c[ok]
        parts.append(torch.unique((r << 32) | c))
    keys = torch.cat(parts)
    rows = n if row_range is None else row_range[1] - row_range[0]
    out = _csr_from_keys(keys, rows, n, seed, dtype, signed)
    if row_range is not None and row_range[0] != 0:
        # values are a function of the GLOBAL coordinate
        gkeys = ((torch.repeat_interleave(torch.arange(rows, device=device, dtype=torch.int64),
                                          out.row_ptr[1:] - out.row_ptr[:-1]) + row_range[0]) << 32) \
            | out.col_idx.to(torch.int64)
        out.vals = _values(gkeys, seed, dtype, signed)
    return out


def rmat_rect(n: int, nnz_target: int, seed: int = 42, dtype=torch.float64, device="cpu",
              signed=False, chunk: int = 1 << 24, oversample: float | None = None, row_range=None) -> Csr:
    """G-rmat clipped to n x n: sample at scale ceil(log2 n), reject row/col >= n, merge duplicates.

    Draws ``oversample * nnz_target`` samples (default calibrated for the Graph500 parameters), so the
    final nnz lands near — not exactly on — nnz_target; the caller reports the actual count.
    """
    scale = max(1, math.ceil(math.log2(n)))
    if oversample is None:
        oversample = 1.0
    m = int(nnz_target * oversample)
    parts = []
    for s in range(0, m, chunk):
        e = torch.arange(s, min(s + chunk, m), dtype=torch.int64, device=device)
        r, c = _rmat_edges(e, scale, seed)
        ok = (r < n) & (c < n)
        if row_range is not None:  # one rank's row block of the same matrix
            ok &= (r >= row_range[0]) & (r < row_range[1])
        parts.append(torch.unique((r[ok] << 32) | c[ok]))
    keys = torch.cat(parts)
    if row_range is None:
        return _csr_from_keys(keys, n, n, seed, dtype, signed)
    gkeys = torch.unique(keys, sorted=True)
    out = _csr_from_keys(gkeys - (row_range[0] << 32), row_range[1] - row_range[0], n, seed, dtype, signed)
    out.vals = _values(gkeys, seed, dtype, signed)  # values are a function of the GLOBAL coordinate
    return out


def _perm_bits(x: torch.Tensor, bits: int, k1: torch.Tensor, k2: torch.Tensor) -> torch.Tensor:
    """A bijection on [0, 2^bits): xor-shift / odd-multiply rounds keyed per row."""
    mask = (1 << bits) - 1
    x = (x + k1) & mask
    x = (x * 0x9E3779B1) & mask
    x = x ^ (x >> (bits // 2 + 1))
    x = (x * 0x85EBCA77) & mask
    x = (x + k2) & mask
    x = x ^ (x >> (bits // 2))
    return x


def powerlaw(n: int = 2_000_000, heavy_every: int = 1000, heavy_nnz: int = 50_000, light_nnz: int = 8,
             seed: int = 43, dtype=torch.float32, device="cpu", signed=False) -> Csr:
    """G-powerlaw: one row in `heavy_every` has exactly heavy_nnz distinct columns, the others exactly
    light_nnz, columns uniform over [0, n) (a keyed bijection with cycle-walking gives distinctness)."""
    assert heavy_nnz <= n and light_nnz <= n
    bits = max(2, math.ceil(math.log2(n)))
    rows = torch.arange(n, dtype=torch.int64, device=device)
    s0 = _i64(mix64_int(seed))
    # heavy row inside each group of `heavy_every`
    grp = rows // heavy_every
    heavy = (rows % heavy_every) == (mix64(grp + s0) & 0x7FFFFFFF) % heavy_every
    cnt = torch.where(heavy, heavy_nnz, light_nnz).to(torch.int64)
    row_ptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(cnt, 0, out=row_ptr[1:])
    r = torch.repeat_interleave(rows, cnt)
    j = torch.arange(r.numel(), dtype=torch.int64, device=device) - row_ptr[r]
    hk = mix64(r ^ s0)
    k1 = hk & ((1 << bits) - 1)
    k2 = _lsr(hk, 32) & ((1 << bits) - 1)
    c = _perm_bits(j, bits, k1, k2)
    for _ in range(64):  # cycle-walk into [0, n)
        bad = c >= n
        if not bool(bad.any()):
            break
        c = torch.where(bad, _perm_bits(c, bits, k1, k2), c)
    keys = torch.sort((r << 32) | c).values
    vals = _values(keys, seed, dtype, signed)
    return Csr(n, n, row_ptr, (keys & 0xFFFFFFFF).to(torch.int32), vals)


def dense_vector(n: int, seed: int = 7, dtype=torch.float64, device="cpu") -> torch.Tensor:
    """x for SpMV: uniform [0.5, 1.5), a function of the index only."""
    i = torch.arange(n, dtype=torch.int64, device=device)
    return (0.5 + u01(mix64(i ^ _i64(mix64_int(seed))))).to(dtype)


def random_csr(rows: int, cols: int, density: float, seed: int, dtype=torch.float64, device="cpu",
               signed=True, sort_rows=True) -> Csr:
    """Small uniform random matrix for property tests (unsorted rows on request, cf. from_dok,
    spam_csr/src/lib.rs:336-358)."""
    m = max(0, int(rows * cols * density))
    e = torch.arange(m, dtype=torch.int64, device=device)
    h = mix64(e + _i64(mix64_int(seed)))
    r = (_lsr(h, 33) % rows)
    c = ((h & 0x7FFFFFFF) % cols)
    out = _csr_from_keys((r << 32) | c, rows, cols, seed, dtype, signed)
    if not sort_rows and out.nnz:
        # shuffle inside rows: sort by (row, hash(col))
        rr = torch.repeat_interleave(torch.arange(rows, dtype=torch.int64, device=device),
                                     out.row_ptr[1:] - out.row_ptr[:-1])
        hk = _lsr(mix64(out.col_idx.to(torch.int64) * 31 + rr + seed), 34)
        order = torch.argsort((rr << 30) | hk)
        out.col_idx = out.col_idx[order]
        out.vals = out.vals[order]
    return out
