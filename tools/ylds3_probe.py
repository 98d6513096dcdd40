#!/usr/bin/env python3
"""Prototype probe (diagnostic): XCD-sliced SpMV of the long rows with LDS-resident accumulators per (part, row group)
block and slot-ordered entries (tools/ylds_proto.hip: ylds3_kernel).  Plan built with torch on the device.
Not part of the product or the tests."""
import os, sys, ctypes, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def build_plan3(rp, ci, vals, x, S, rmax, eblk, long_min=9, sub_factor=128):
    dev = rp.device
    rows = rp.numel() - 1
    cols = x.numel()
    lens = rp[1:] - rp[:-1]
    cc = torch.bincount(ci.to(torch.int64), minlength=cols)
    hot = torch.argsort(cc, descending=True, stable=True)[: 8 * S]
    hot = hot[cc[hot] > 0]
    h = torch.arange(hot.numel(), device=dev)
    grp8 = h // 8
    slab = (h % 8 + grp8 * 5) % 8
    dest_part = torch.full((cols,), 8, dtype=torch.int64, device=dev)
    dest_idx = torch.arange(cols, device=dev)
    dest_part[hot] = slab
    dest_idx[hot] = grp8
    xh = torch.zeros(8 * S, dtype=torch.float64, device=dev)
    xh[slab * S + grp8] = x[hot]
    # long rows, by length
    order = torch.argsort(lens, descending=True, stable=True)
    nl = int((lens >= long_min).sum())
    order = order[:nl]
    L = lens[order]
    nsub = torch.clamp((L * sub_factor + 8 * eblk - 1) // (8 * eblk), 1, 64)
    acc_incl = torch.cumsum(nsub, 0)
    acc_excl = acc_incl - nsub
    nnz_excl = torch.cumsum(L, 0) - L
    key = (nnz_excl // (9 * eblk)) * (1 << 32) + acc_excl // (rmax - 64)
    _, grp_of_rank = torch.unique_consecutive(key, return_inverse=True)
    ng = int(grp_of_rank[-1]) + 1
    first = torch.ones(nl, dtype=torch.bool, device=dev)
    first[1:] = grp_of_rank[1:] != grp_of_rank[:-1]
    nacc = int(acc_incl[-1])
    grp_acc = torch.empty(ng + 1, dtype=torch.int64, device=dev)
    grp_acc[:ng] = acc_excl[first]
    grp_acc[ng] = nacc
    acc_rank = torch.full((nacc,), 0xFFFFFFFF, dtype=torch.int64, device=dev)
    acc_rank[acc_excl] = torch.arange(nl, device=dev)
    rowidx = torch.repeat_interleave(torch.arange(rows, device=dev), lens)
    rank_of_row = torch.full((rows,), -1, dtype=torch.int64, device=dev)
    rank_of_row[order] = torch.arange(nl, device=dev)
    k = rank_of_row[rowidx]
    sel = torch.nonzero(k >= 0).squeeze(1)
    k = k[sel]
    pos = sel - rp[rowidx[sel]]
    del rowidx
    col = ci[sel].to(torch.int64)
    v = vals[sel]
    part = dest_part[col]
    idx = dest_idx[col]
    g = grp_of_rank[k]
    la = acc_excl[k] - grp_acc[g] + pos % nsub[k]
    assert int(la.max()) < rmax
    b = part * ng + g
    skey = b * (1 << 32) + idx
    del pos, k, part, g, col
    skey, perm = torch.sort(skey)
    b = skey >> 32
    nb = 9 * ng
    cnt = torch.bincount(b, minlength=nb)
    padded = (cnt + 255) // 256 * 256
    blk_ent = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
    torch.cumsum(padded, 0, out=blk_ent[1:])
    start = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
    torch.cumsum(cnt, 0, out=start[1:])
    j = torch.arange(skey.numel(), device=dev) - start[b]
    w = j % 256
    dst = blk_ent[b] + (j // 256) * 256 + (w % 64) * 4 + w // 64
    ntot = int(blk_ent[nb])
    ecol = torch.zeros(ntot, dtype=torch.int32, device=dev)
    eacc = torch.zeros(ntot, dtype=torch.int16, device=dev)
    evalv = torch.zeros(ntot, dtype=torch.float64, device=dev)
    ecol[dst] = (skey & 0xFFFFFFFF).to(torch.int32)
    eacc[dst] = la[perm].to(torch.int16)
    evalv[dst] = v[perm]
    part_cnt = cnt.view(9, ng).sum(1)
    info = dict(ng=ng, nb=nb, nl=nl, entries=int(skey.numel()), padded=ntot, naccs=nacc, part_entries=part_cnt.tolist(),
                max_blk_entries=int(cnt.max()), table_cols=int(hot.numel()))
    return dict(blk_ent=blk_ent, blk_grp=(torch.arange(nb, device=dev) % ng).to(torch.int32), grp_acc=grp_acc.to(torch.int32),
                acc_rank=acc_rank.to(torch.int32), ecol=ecol, eacc=eacc, eval=evalv, xh=xh, S=S, nl=nl, ng=ng, nb=nb,
                part_blk=np.array([p * ng for p in range(10)], dtype=np.uint32), order=order, cnt=cnt, info=info)


def build_plan4(P3, abits):
    """Re-pack a plan of build_plan3 into the chunk-interleaved stream of ylds4_kernel."""
    dev = P3["ecol"].device
    nb, ng = P3["nb"], P3["ng"]
    blk_ent = P3["blk_ent"]
    nch = (blk_ent[1:] - blk_ent[:-1]) // 256                      # chunks per block
    wide = torch.arange(nb, device=dev) >= 8 * ng
    cb = torch.where(wide, 3584, 3072)
    blk_off = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
    torch.cumsum(nch * cb, 0, out=blk_off[1:])
    total = int(blk_off[nb])
    stream = torch.zeros(total, dtype=torch.uint8, device=dev)
    ntot = P3["ecol"].numel()
    pos = torch.arange(ntot, device=dev)
    b = torch.bucketize(pos, blk_ent[1:], right=True)             # block of every padded entry
    inb = pos - blk_ent[b]
    chunk = inb // 256
    r = inb % 256                                                  # = 4 * lane + k in the plan3 layout
    lane, k = r // 4, r % 4
    cbase = blk_off[b] + chunk * cb[b]
    isw = wide[b]
    word = torch.where(isw, P3["ecol"].to(torch.int64), (P3["ecol"].to(torch.int64) << abits) | (P3["eacc"].to(torch.int64) & 0xFFFF))
    s32 = stream.view(torch.int32)
    s16 = stream.view(torch.int16)
    s64 = stream.view(torch.float64)
    word = torch.where(word >= (1 << 31), word - (1 << 32), word)
    s32[(cbase + 4 * r) // 4] = word.to(torch.int32)
    sel = torch.nonzero(isw).squeeze(1)
    s16[(cbase[sel] + 1024 + 2 * r[sel]) // 2] = P3["eacc"][sel]
    voff = torch.where(isw, 1536, 1024)
    s64[(cbase + voff + ((k // 2) * 128 + 2 * lane + k % 2) * 8) // 8] = P3["eval"]
    return dict(blk_off=blk_off, stream=stream, bytes=total)


def main():
    dev = torch.device("cuda", 0)
    lib = ctypes.CDLL(os.path.join(HERE, "libylds_proto.so"))
    which = sys.argv[1] if len(sys.argv) > 1 else "rmat"
    if which == "rmat":
        A = synth.rmat_rect(10_000_000, 160_000_000, dtype=torch.float64, device=dev, oversample=1.30)
    else:
        A = synth.rmat_rect(300_000, 4_800_000, dtype=torch.float64, device=dev)
    x = synth.dense_vector(A.cols, dtype=torch.float64, device=dev)
    rp, ci, vals = A.row_ptr, A.col_idx, A.vals
    rowidx = torch.repeat_interleave(torch.arange(A.rows, device=dev), rp[1:] - rp[:-1])
    y_ref = torch.zeros(A.rows, dtype=torch.float64, device=dev).index_add_(0, rowidx, vals * x[ci.to(torch.int64)])
    del rowidx
    print(which, A.stats(), flush=True)
    S0 = 348032 if which == "rmat" else 4096
    configs = json.loads(os.environ.get("YLDS_CONFIGS", "null")) or [
        dict(S=S0, rmax=8192, eblk=100000, nt=1024, k=2, wgcu=2),
        dict(S=S0, rmax=8192, eblk=100000, nt=1024, k=2, wgcu=2, v4=1, cold_mask=3),
        dict(S=S0, rmax=8192, eblk=100000, nt=1024, k=2, wgcu=2, v4=1, cold_mask=3, ntl=0),
        dict(S=S0, rmax=8192, eblk=100000, nt=1024, k=2, wgcu=2, v4=1, cold_mask=1),
        dict(S=S0, rmax=8192, eblk=100000, nt=1024, k=2, wgcu=2, v4=1, cold_mask=255),
        dict(S=S0, rmax=8192, eblk=60000, nt=1024, k=2, wgcu=2, v4=1, cold_mask=3),
        dict(S=S0, rmax=8192, eblk=100000, nt=512, k=4, wgcu=2, v4=1, cold_mask=3),
        dict(S=262144, rmax=16384, eblk=200000, nt=1024, k=4, wgcu=1, v4=1, cold_mask=3),
        dict(S=262144, rmax=16384, eblk=200000, nt=1024, k=3, wgcu=1, v4=1, cold_mask=3),
    ]
    queue = torch.zeros(16 * 9, dtype=torch.int32, device=dev)
    for cfg in configs:
        t0 = time.time()
        P = build_plan3(rp, ci, vals, x, cfg["S"], cfg["rmax"], cfg["eblk"], cfg.get("long_min", 9), cfg.get("sub", 128))
        torch.cuda.synchronize()
        tb = time.time() - t0
        P4 = None
        if cfg.get("v4"):
            abits = max(1, int(np.ceil(np.log2(cfg["rmax"]))))
            assert (cfg["S"] << abits) <= (1 << 32), "slot and accumulator bits do not fit one word"
            P4 = build_plan4(P, abits)
            P["ecol"] = P["eacc"] = P["eval"] = None
        partial = torch.zeros(9 * P["nl"], dtype=torch.float64, device=dev)
        times = torch.zeros(P["nb"] * 4, dtype=torch.int64, device=dev)
        ms = ctypes.c_float(0)
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        if P4 is not None:
            rc = lib.ylds4_run(vp(P4["blk_off"]), vp(P["blk_grp"]), vp(P["grp_acc"]), vp(P["acc_rank"]), vp(P4["stream"]), vp(P["xh"]),
                               ctypes.c_uint64(P["S"]), vp(x), P["part_blk"].ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(P["nl"]),
                               vp(partial), vp(queue), vp(times), abits, cfg.get("cold_mask", 3), cfg["nt"], cfg["k"], cfg.get("ntl", 1),
                               cfg["rmax"] * 8, 256 * cfg["wgcu"], int(os.environ.get("YLDS_ITERS", "20")), ctypes.byref(ms))
        else:
          rc = lib.ylds3_run(vp(P["blk_ent"]), vp(P["blk_grp"]), vp(P["grp_acc"]), vp(P["acc_rank"]), vp(P["ecol"]), vp(P["eacc"]),
                           vp(P["eval"]), vp(P["xh"]), ctypes.c_uint64(P["S"]), vp(x), P["part_blk"].ctypes.data_as(ctypes.c_void_p),
                           ctypes.c_uint64(P["nl"]), vp(partial), vp(queue), vp(times), cfg["nt"], cfg["k"], cfg.get("abl", 0),
                           cfg["rmax"] * 8, 256 * cfg["wgcu"], int(os.environ.get("YLDS_ITERS", "20")), ctypes.byref(ms))
        torch.cuda.synchronize()
        err = float("nan")
        if not cfg.get("abl", 0):
            yl = partial.view(9, P["nl"]).sum(0)
            ref = y_ref[P["order"]]
            err = float(((yl - ref).abs() / ref.abs().clamp_min(1e-300)).max())
        ent = P["info"]["entries"]
        if P4 is not None:
            print("   stream bytes %.3f GB -> %.0f GB/s" % (P4["bytes"] / 1e9, P4["bytes"] / ms.value / 1e6))
        print(json.dumps(cfg), "rc", rc, "plan", P["info"], "build %.1fs" % tb,
              "-> %.4f ms  %.1f G entries/s  (12 B/entry: %.0f GB/s)  max rel err %.2e" % (ms.value, ent / ms.value / 1e6,
                                                                                         ent * 12 / ms.value / 1e6, err), flush=True)
        t = times.cpu().numpy().reshape(-1, 4)
        xcc = (t[:, 3].astype(np.uint64) >> np.uint64(60)).astype(np.int64)
        t = t.astype(np.uint64)
        t[:, 3] &= np.uint64((1 << 60) - 1)
        t = t.astype(np.float64)
        ran = t[:, 0] > 0
        t0_ = t[ran, 0].min()
        t = (t - t0_) / 100.0
        cnt = P["cnt"].cpu().numpy()
        ng = P["ng"]
        print("   kernel span %.1f us" % t[ran, 3].max())
        for p in range(9):
            sl = slice(p * ng, (p + 1) * ng)
            own = (xcc[sl] == p).mean() if p < 8 else float("nan")
            if os.environ.get("YLDS_PARTS", "0") == "0" and p not in (0, 8):
                continue
            print("   part %d: entries %9d, blocks %d, last end %.1f us, served by own XCD %.2f, stream us mean %.1f max %.1f; zero %.1f write %.1f"
                  % (p, cnt[sl].sum(), ng, t[sl, 3].max(), own, (t[sl, 2] - t[sl, 1]).mean(), (t[sl, 2] - t[sl, 1]).max(),
                     (t[sl, 1] - t[sl, 0]).mean(), (t[sl, 3] - t[sl, 2]).mean()), flush=True)
        if os.environ.get("YLDS_RATES"):
            sl = slice(0, ng)  # part 0
            st = (t[sl, 2] - t[sl, 1])
            e = cnt[sl].astype(np.float64)
            na = (P["grp_acc"][1:] - P["grp_acc"][:-1]).cpu().numpy()
            for lo in range(0, ng, max(1, ng // 12)):
                hi = min(ng, lo + max(1, ng // 12))
                print("   part 0 groups %4d..%4d: accs/group %6.0f entries/block %8.0f stream %6.1f us -> %6.2f entries/ns/WG; zero+write %.1f us"
                      % (lo, hi - 1, na[lo:hi].mean(), e[lo:hi].mean(), st[lo:hi].mean(), e[lo:hi].sum() / st[lo:hi].sum() / 1e3,
                         ((t[sl, 1] - t[sl, 0]) + (t[sl, 3] - t[sl, 2]))[lo:hi].mean()), flush=True)
        del P, partial
        torch.cuda.empty_cache()


main()
