#!/usr/bin/env python3
"""Prototype probe (diagnostic): SpMV with LDS-resident y accumulators per row block and column-ordered entries.
Plan built with torch on the device; kernel in tools/ylds_proto.hip.  Not part of the product or the tests."""
import os, sys, ctypes, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def build_plan(rp, ci, vals, nb_target, rmax, minlen=1, sub_factor=128):
    dev = rp.device
    rows = rp.numel() - 1
    lens = rp[1:] - rp[:-1]
    order = torch.argsort(lens, descending=True, stable=True)
    n1 = int((lens >= minlen).sum())
    order = order[:n1]
    L = lens[order]
    tot = int(L.sum())
    E = -(-tot // nb_target)
    nsub = torch.clamp((L * sub_factor + E - 1) // E, 1, 64)
    acc_incl = torch.cumsum(nsub, 0)
    acc_excl = acc_incl - nsub
    nnz_excl = torch.cumsum(L, 0) - L
    key = (nnz_excl // E) * (1 << 32) + acc_excl // (rmax - 64)
    _, blk_of_rank = torch.unique_consecutive(key, return_inverse=True)
    nb = int(blk_of_rank[-1]) + 1
    first = torch.ones(n1, dtype=torch.bool, device=dev)
    first[1:] = blk_of_rank[1:] != blk_of_rank[:-1]
    blk_acc = torch.empty(nb + 1, dtype=torch.int64, device=dev)
    blk_acc[:nb] = acc_excl[first]
    nacc = int(acc_incl[-1])
    blk_acc[nb] = nacc
    acc_row = torch.full((nacc,), 0xFFFFFFFF, dtype=torch.int64, device=dev)
    acc_row[acc_excl] = order
    # entries
    nnz = ci.numel()
    rowidx = torch.repeat_interleave(torch.arange(rows, device=dev), lens)
    rank_of_row = torch.full((rows,), -1, dtype=torch.int64, device=dev)
    rank_of_row[order] = torch.arange(n1, device=dev)
    k = rank_of_row[rowidx]
    keep = k >= 0
    if not bool(keep.all()):
        sel = torch.nonzero(keep).squeeze(1)
        k = k[sel]
        rowsel = rowidx[sel]
        pos = sel - rp[rowsel]
        col = ci[sel].to(torch.int64)
        v = vals[sel]
    else:
        pos = torch.arange(nnz, device=dev) - rp[rowidx]
        col = ci.to(torch.int64)
        v = vals
    del rowidx
    eb = blk_of_rank[k]
    la = acc_excl[k] - blk_acc[eb] + pos % nsub[k]
    assert int(la.max()) < rmax, (int(la.max()), rmax)
    skey = eb * (1 << 32) + col
    del pos, k
    skey, perm = torch.sort(skey)
    eb = skey >> 32
    cnt = torch.bincount(eb, minlength=nb)
    padded = (cnt + 255) // 256 * 256
    blk_ent = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
    torch.cumsum(padded, 0, out=blk_ent[1:])
    start = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
    torch.cumsum(cnt, 0, out=start[1:])
    j = torch.arange(skey.numel(), device=dev) - start[eb]
    w = j % 256
    dst = blk_ent[eb] + (j // 256) * 256 + (w % 64) * 4 + w // 64
    ntot = int(blk_ent[nb])
    ecol = torch.zeros(ntot, dtype=torch.int32, device=dev)
    eacc = torch.zeros(ntot, dtype=torch.int16, device=dev)
    evalv = torch.zeros(ntot, dtype=torch.float64, device=dev)
    ecol[dst] = (skey & 0xFFFFFFFF).to(torch.int32)
    eacc[dst] = la[perm].to(torch.int16)
    evalv[dst] = v[perm]
    info = dict(nb=nb, entries=int(skey.numel()), padded=ntot, naccs=nacc, rows_in=n1, E=E,
                max_blk_entries=int(cnt.max()), max_blk_accs=int((blk_acc[1:] - blk_acc[:-1]).max()))
    return dict(blk_ent=blk_ent.contiguous(), blk_acc=blk_acc.to(torch.int32).contiguous(), ecol=ecol, eacc=eacc, eval=evalv,
                acc_row=acc_row.to(torch.int32).contiguous(), nb=nb, info=info)


def main():
    dev = torch.device("cuda", 0)
    lib = ctypes.CDLL(os.path.join(HERE, "libylds_proto.so"))
    which = sys.argv[1] if len(sys.argv) > 1 else "rmat"
    if which == "rmat":
        A = synth.rmat_rect(10_000_000, 160_000_000, dtype=torch.float64, device=dev, oversample=1.30)
    elif which == "small":
        A = synth.rmat_rect(300_000, 4_800_000, dtype=torch.float64, device=dev)
    x = synth.dense_vector(A.cols, dtype=torch.float64, device=dev)
    rp, ci, vals = A.row_ptr, A.col_idx, A.vals
    nnz = ci.numel()
    nbytes = nnz * 12 + (A.rows + 1) * 4 + A.cols * 8 + A.rows * 8
    rowidx = torch.repeat_interleave(torch.arange(A.rows, device=dev), rp[1:] - rp[:-1])
    y_ref = torch.zeros(A.rows, dtype=torch.float64, device=dev).index_add_(0, rowidx, vals * x[ci.to(torch.int64)])
    del rowidx
    print(which, A.stats(), "alg bytes %.3f GB" % (nbytes / 1e9), flush=True)
    configs = json.loads(os.environ.get("YLDS_CONFIGS", "null")) or [
        dict(nb=256, rmax=16384, nt=1024, wgcu=1),
        dict(nb=256, rmax=16384, nt=1024, wgcu=1, k=4),
        dict(nb=256, rmax=16384, nt=1024, wgcu=1, k=4, abl=1),
        dict(nb=256, rmax=16384, nt=1024, wgcu=1, k=4, abl=2),
        dict(nb=256, rmax=16384, nt=1024, wgcu=1, k=4, abl=3),
        dict(nb=256, rmax=16384, nt=1024, wgcu=1, k=2),
        dict(nb=256, rmax=16384, nt=1024, wgcu=1, k=6),
        dict(nb=512, rmax=8192, nt=512, wgcu=2, k=4),
        dict(nb=512, rmax=8192, nt=512, wgcu=2, k=8),
    ]
    y = torch.empty(A.rows, dtype=torch.float64, device=dev)
    ticket = torch.zeros(64, dtype=torch.int32, device=dev)
    for cfg in configs:
        t0 = time.time()
        P = build_plan(rp, ci, vals, cfg["nb"], cfg["rmax"], cfg.get("minlen", 1), cfg.get("sub", 128))
        torch.cuda.synchronize()
        tb = time.time() - t0
        ms = ctypes.c_float(0)
        times = torch.zeros(P["nb"] * 4, dtype=torch.int64, device=dev)
        rc = lib.ylds_run(ctypes.c_void_p(P["blk_ent"].data_ptr()), ctypes.c_void_p(P["blk_acc"].data_ptr()),
                          ctypes.c_void_p(P["ecol"].data_ptr()), ctypes.c_void_p(P["eacc"].data_ptr()),
                          ctypes.c_void_p(P["eval"].data_ptr()), ctypes.c_void_p(P["acc_row"].data_ptr()),
                          ctypes.c_uint32(P["nb"]), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                          ctypes.c_uint64(A.rows), ctypes.c_void_p(ticket.data_ptr()), cfg["nt"], cfg["rmax"] * 8,
                          256 * cfg["wgcu"], int(os.environ.get("YLDS_ITERS", "20")), ctypes.byref(ms),
                          ctypes.c_void_p(times.data_ptr()), cfg.get("zero_y", 1), cfg.get("abl", 0), cfg.get("k", 0))
        torch.cuda.synchronize()
        if cfg.get("minlen", 1) == 1 and not cfg.get("abl", 0):
            err = float(((y - y_ref).abs() / y_ref.abs().clamp_min(1e-300)).max())
        else:
            err = float("nan")
        print(json.dumps(cfg), "rc", rc, "plan", P["info"], "build %.1fs" % tb,
              "-> %.4f ms  %.0f GB/s (%.1f%% of 8 TB/s)  max rel err %.2e" % (ms.value, nbytes / ms.value / 1e6,
                                                                              nbytes / ms.value / 1e6 / 80, err), flush=True)
        t = times.cpu().numpy().reshape(-1, 4).astype(np.float64)
        t0 = t[:, 0].min()
        t = (t - t0) / 100.0  # us
        d = t[:, 3] - t[:, 0]
        nbk = P["nb"]
        cnt = (P["blk_ent"][1:] - P["blk_ent"][:-1]).cpu().numpy()
        na = (P["blk_acc"][1:] - P["blk_acc"][:-1]).cpu().numpy()
        print("   kernel span %.1f us; first-wave blocks (0..%d): end min/med/max %.1f/%.1f/%.1f us; zero %.1f stream %.1f write %.1f us (medians over those)"
              % (t[:, 3].max(), min(nbk, 256 * cfg["wgcu"]) - 1, *np.percentile(t[:256 * cfg["wgcu"], 3], [0, 50, 100]),
                 np.median((t[:, 1] - t[:, 0])[:256]), np.median((t[:, 2] - t[:, 1])[:256]), np.median((t[:, 3] - t[:, 2])[:256])))
        for lo in range(0, nbk, max(1, nbk // 8)):
            hi = min(nbk, lo + max(1, nbk // 8))
            print("   blocks %4d..%4d: entries %8.0f accs %6.0f | start %.1f..%.1f us, dur zero %.1f stream %.1f write %.1f us (mean)"
                  % (lo, hi - 1, cnt[lo:hi].mean(), na[lo:hi].mean(), t[lo:hi, 0].min(), t[lo:hi, 0].max(),
                     (t[lo:hi, 1] - t[lo:hi, 0]).mean(), (t[lo:hi, 2] - t[lo:hi, 1]).mean(), (t[lo:hi, 3] - t[lo:hi, 2]).mean()), flush=True)
        del P
        torch.cuda.empty_cache()


main()
