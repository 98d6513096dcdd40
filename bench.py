#!/usr/bin/env python3
"""bench.py — the spam_csr hot path on MI355X: CSR SpMV (primary line) + hash SpGEMM (secondary object).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one synthetic input resident in HBM:
  * SpMV  (BASELINE.json configs[1]): y = A x, A = R-MAT clipped to 10 M x 10 M, ~160 M nnz, f64.
    N > 1: every rank owns the same-sized row block (weak scaling, x replicated, no collective).
  * SpGEMM (configs[2]/[3]): C = A * A, R-MAT scale S, sorted input, unsorted output (the reference's
    bench shape, spam_csr/src/lib.rs:402-407).  N > 1: B's row blocks are all-gathered over RCCL, A's
    rows are split into flop-balanced blocks (rows_to_threads with tnum = N, mul_hash.rs:38-64), each rank
    computes its block of C (strong scaling; the all-gather is inside the timed region).

Prints ONE JSON line (rank 0).  `value` = SpMV effective GB/s summed over ranks, algorithmic bytes
(nnz*(8+4) + (rows+1)*4 + cols*8 + rows*8) / max-over-ranks step time.
The oracle (oracle/) is used only for the cpu_baseline leg and the parity spot-check.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from sparse_matrix_amd import DeviceCsr, _lib, synth  # noqa: E402
from sparse_matrix_amd import dist as spdist  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); measured copy ceiling there: 6290 GB/s


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def spmv_alg_bytes(rows, cols, nnz, vsize):
    return nnz * (vsize + 4) + (rows + 1) * 4 + cols * vsize + rows * vsize


def spgemm_alg_bytes(st, vsize):
    # compulsory traffic: read A once, B once, write C once
    return (4 + vsize) * (st["nnz_a"] + st["nnz_b"] + st["nnz_c"]) + 4 * (st["rows"] + st["inner"] + st["rows"] + 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--spmv-rows", type=int, default=10_000_000)
    ap.add_argument("--spmv-nnz", type=int, default=160_000_000)
    ap.add_argument("--spmv-oversample", type=float, default=1.30)
    ap.add_argument("--spgemm-scale", type=int, default=20, help="R-MAT scale of C = A*A (configs[2]); 0 disables; "
                    "22 (configs[3]) needs >= 8 GPUs: nnz(C) ~ 8e10 does not fit one")
    ap.add_argument("--spgemm-steps", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spmv", action="store_true")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the informational f32 power-law SpMV line")
    ap.add_argument("--spmv-sharded", action="store_true",
                    help="N > 1: ONE 10 M-row matrix row-sharded over the ranks, x all-gathered every step (strong scaling) "
                         "instead of one matrix per rank (weak scaling, the default)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    # SPAM_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal on a 1-GPU box); the real run is nccl = RCCL
    backend = os.environ.get("SPAM_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_

        dist = dist_
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    _lib.ensure_device(dev_index)
    L = _lib.load()
    _lib.check(L.spam_hip_set_stream(torch.cuda.current_stream().cuda_stream))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v: float) -> float:
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    out = {}
    # ------------------------------------------------------------------ SpMV (primary) -------------
    if not args.no_spmv:
        t0 = time.time()
        sharded = args.spmv_sharded and world > 1
        if sharded:
            # ONE matrix, rank r holds rows [r*n/world, (r+1)*n/world) and the matching slice of x
            rlo, rhi = spdist.equal_row_block(args.spmv_rows, rank, world)
            A = synth.rmat_rect(args.spmv_rows, args.spmv_nnz, seed=42, dtype=torch.float64, device=dev,
                                oversample=args.spmv_oversample, row_range=(rlo, rhi))
            x_local = synth.dense_vector(args.spmv_rows, dtype=torch.float64, device=dev)[rlo:rhi].clone()
            x = spdist.allgather_vector(x_local, args.spmv_rows, rank, world, dist)
        else:
            A = synth.rmat_rect(args.spmv_rows, args.spmv_nnz, seed=42 + rank, dtype=torch.float64, device=dev,
                                oversample=args.spmv_oversample)
            x = synth.dense_vector(A.cols, dtype=torch.float64, device=dev)
        y = torch.empty(A.rows, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        stA = A.stats()
        log(f"[spmv] generated {stA} in {time.time() - t0:.1f}s")
        dA = DeviceCsr.from_torch(A.rows, A.cols, A.row_ptr, A.col_idx, A.vals, True)
        nbytes = spmv_alg_bytes(A.rows, A.cols, A.nnz, 8)
        if sharded:  # whole-job bytes = sum of the blocks; x is counted once per rank (every rank reads all of it)
            tb = torch.tensor([float(nbytes)], dtype=torch.float64, device=dev)
            dist.all_reduce(tb)
            job_bytes = float(tb.item())
        def spmv_step():
            if sharded:  # the exchange step of a row-sharded SpMV: all-gather x, inside the timed region
                xg = spdist.allgather_vector(x_local, args.spmv_rows, rank, world, dist)
                dA.spmv_dev(xg.data_ptr(), y.data_ptr())
            else:
                dA.spmv_dev(x.data_ptr(), y.data_ptr())

        for _ in range(args.warmup):
            spmv_step()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        barrier()
        t0 = time.perf_counter()
        for s, e in evs:
            s.record()
            spmv_step()
            e.record()
        barrier()
        wall = time.perf_counter() - t0
        wall = max_over_ranks(wall)
        kern_ms = float(np.mean([s.elapsed_time(e) for s, e in evs]))
        ms_per_step = wall * 1e3 / args.steps
        value = (job_bytes if sharded else world * nbytes) / (wall / args.steps) / 1e9
        achieved = nbytes / (kern_ms * 1e-3) / 1e9
        out.update({
            "metric": "spmv_effective_GB_per_s", "value": round(value, 1), "unit": "GB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "csr_spmv_f64_rmat_10Mx10M_160Mnnz (BASELINE configs[1])", "rows": A.rows,
                       "cols": A.cols, "nnz": A.nnz, "avg_nnz_row": round(stA["avg_nnz_row"], 2),
                       "max_nnz_row": stA["max_nnz_row"], "gflops": round(2 * A.nnz * world / (wall / args.steps) / 1e9, 1),
                       "sharding": ("one matrix row-sharded, x all-gathered every step" if sharded else
                                    "one matrix per rank, x replicated, no collective") if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                         "kernel": "spmv_tile_kernel(+fixup)", "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes": nbytes},
        })
        # what the gather allows on this chip (tools/gather_bench.hip, tools/xcd_gather_bench.hip): 245 G gathers/s when
        # they hit the XCD's L2 (one request per lane: the L2's request rate), 52 G/s when they miss (a 128-byte line
        # crosses the fabric per miss).  Plain path: one 3 MB hot table makes `hot_cover` of them hit.  XCD-sliced path:
        # eight per-XCD tables make `cover_table` of the long rows' gathers hit (230 G/s measured with every workgroup on
        # its own XCD's slice), the rest and the short rows' misses pay the fabric rate.
        info = dA.spmv_plan_info()
        sl = info["sliced"]
        if sl["active"]:
            hits = sl["cover_table"] * sl["nnz_long"]
            model_ms = max(nbytes / 5.8e12, hits / 230e9 + (sl["nnz_long"] - hits) / 52e9) * 1e3
            out["roofline"]["kernel"] = ("XCD-sliced SpMV: sliced_build_tables + spmv_short_rows + spmv_sliced_pipe_kernel "
                                         "+ sliced_fixup + sliced_reduce (whole step)")
            out["roofline"]["gather_model"] = {"mode": "xcd_sliced", "l2_hit_gathers_per_s": 230e9, "l2_miss_gathers_per_s": 52e9,
                                               "bound_ms": round(model_ms, 4), "frac_of_bound": round(model_ms / kern_ms, 4),
                                               "source": "profiles/r01_xcd_gather_microbench.txt"}
            out["config"]["spmv_plan"] = {"mode": "xcd_sliced", "rows_in_slabs": sl["long_rows"], "nnz_in_slabs": sl["nnz_long"],
                                          "table_slots_per_xcd": sl["table_slots_per_xcd"], "hottest_slots_l1_lds": sl["lds_slots"],
                                          "cover_8_tables": round(sl["cover_table"], 4), "cover_1_table": round(sl["cover_single"], 4),
                                          "autotune_ms_plain": round(sl["ms_plain"], 4), "autotune_ms_sliced": round(sl["ms_sliced"], 4)}
        else:
            hits = info["hot_cover"] * A.nnz if info["hot_columns"] else 0.0
            model_ms = max(nbytes / 5.8e12, hits / 245e9 + (A.nnz - hits) / 52e9) * 1e3
            out["roofline"]["gather_model"] = {"mode": "plain", "hot_columns": info["hot_columns"], "hot_cover": round(info["hot_cover"], 4),
                                               "l2_hit_gathers_per_s": 245e9, "l2_miss_gathers_per_s": 52e9,
                                               "bound_ms": round(model_ms, 4), "frac_of_bound": round(model_ms / kern_ms, 4),
                                               "source": "profiles/r01_gather_microbench.txt"}
            out["config"]["spmv_plan"] = {"mode": "plain", "hot_columns": info["hot_columns"]}
        # HBM/fabric bytes per launch come from a separate rocprofv3 --pmc run (tools/prof_bench.sh), never from this process
        pmc = os.path.join(ROOT, "profiles", "spmv_pmc_latest.json")
        if os.path.exists(pmc):
            try:
                out["roofline"]["traffic"] = round(json.load(open(pmc))["traffic_bytes_per_launch"])
                out["roofline"]["traffic_source"] = "profiles/spmv_pmc_latest.json (2*FETCH_SIZE + WRITE_SIZE, gfx950 correction)"
            except Exception:
                pass
        log(f"[spmv] {ms_per_step:.4f} ms/step wall, {kern_ms:.4f} ms device, {achieved:.0f} GB/s "
            f"({100 * achieved / HBM_PEAK_GBPS:.1f}% of 8 TB/s)")
        # parity spot check + CPU baseline: rank 0 at N = 1 only
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            import oracle

            t0 = time.time()
            a_host = A.cpu_numpy()
            xh = x.cpu().numpy()
            yh = y.cpu().numpy()
            cores = oracle.hardware_threads()
            ts = []
            ref = None
            for _ in range(11):
                t1 = time.perf_counter()
                ref = oracle.spmv(a_host, xh, nthreads=cores)
                ts.append(time.perf_counter() - t1)
            cpu_s = float(np.median(ts[1:]))
            nzm = ref != 0
            rel = float(np.max(np.abs(yh[nzm] - ref[nzm]) / np.abs(ref[nzm])))
            out["cpu_baseline"] = {"value": round(nbytes / cpu_s / 1e9, 2), "unit": "GB/s", "cores": cores, "kind": "port",
                                   "sample": f"the full cfg2 matrix, {len(ts) - 1} timed SpMV passes (median) after 1 warm-up, "
                                             f"C++ restatement (Rust reference not buildable here)"}
            out["parity"] = {"spmv_max_rel_err_vs_oracle": rel, "spmv_tolerance": 1e-10,
                             "spmv_zero_rows_exact": bool(np.all(yh[~nzm] == 0))}
            log(f"[spmv] cpu port {cpu_s * 1e3:.1f} ms on {cores} threads; max rel err {rel:.2e} ({time.time() - t0:.0f}s)")
            del a_host, xh, yh, ref
        del dA, A, x, y
        torch.cuda.empty_cache()
        # BASELINE configs[4] beside it (rank 0, N = 1 only; informational, never `value`): 2 M rows, 0.1 % of them with
        # 50 000 entries, f32 — the row-bin / long-row stress case, same timing method
        if rank == 0 and world == 1 and not args.no_cfg5:
            A5 = synth.powerlaw(dtype=torch.float32, device=dev)
            x5 = synth.dense_vector(A5.cols, dtype=torch.float32, device=dev)
            y5 = torch.empty(A5.rows, dtype=torch.float32, device=dev)
            d5 = DeviceCsr.from_torch(A5.rows, A5.cols, A5.row_ptr, A5.col_idx, A5.vals, True)
            for _ in range(args.warmup):
                d5.spmv_dev(x5.data_ptr(), y5.data_ptr())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.steps):
                d5.spmv_dev(x5.data_ptr(), y5.data_ptr())
            e1.record()
            torch.cuda.synchronize()
            ms5 = e0.elapsed_time(e1) / args.steps
            b5 = spmv_alg_bytes(A5.rows, A5.cols, A5.nnz, 4)
            out["spmv_powerlaw_f32"] = {"workload": "csr_spmv_f32_powerlaw_2Mx2M (BASELINE configs[4])", "rows": A5.rows,
                                        "nnz": A5.nnz, "max_nnz_row": A5.stats()["max_nnz_row"], "ms_per_step": round(ms5, 4),
                                        "effective_GB_per_s": round(b5 / (ms5 * 1e-3) / 1e9, 1),
                                        "frac_of_hbm_peak": round(b5 / (ms5 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                        "algorithmic_bytes": b5,
                                        "plan": "xcd_sliced" if d5.spmv_plan_info()["sliced"]["active"] else "plain"}
            log(f"[spmv f32 power-law] {ms5:.4f} ms/step, {b5 / (ms5 * 1e-3) / 1e9:.0f} GB/s")
            del d5, A5, x5, y5
            torch.cuda.empty_cache()

    # ------------------------------------------------------------------ SpGEMM (secondary) ---------
    scale = args.spgemm_scale
    if scale > 0:
        n = 1 << scale
        # every rank starts with an equal row block of A (= its block of B); values depend on global coordinates
        lo, hi = spdist.equal_row_block(n, rank, world)
        t0 = time.time()
        blk = synth.rmat(scale, dtype=torch.float64, device=dev, row_range=(lo, hi) if world > 1 else None)
        torch.cuda.synchronize()
        log(f"[spgemm] rank block rows [{lo},{hi}) nnz {blk.nnz} generated in {time.time() - t0:.1f}s")

        def one_step():
            """all-gather B, flop-balanced row block of A, local two-phase SpGEMM.  Returns (C, stats)."""
            rp, col, val = spdist.allgather_csr(blk.row_ptr, blk.col_idx, blk.vals, n, rank, world, dist)
            B = DeviceCsr.from_torch(n, n, rp, col, val, True)
            if world > 1:
                # rows_to_threads' flop count and partition rule (mul_hash.rs:39-62) on F + a per-row cost, on the device
                offs = B.balanced_row_blocks(B, world, spdist.ROW_OVERHEAD_PRODUCTS)
                mylo, myhi = int(offs[rank]), int(offs[rank + 1])
                if myhi <= mylo:
                    return None, None
                Ablk = B.row_block(mylo, myhi)
            else:
                Ablk = B
            C, st = Ablk.spgemm(B, sort_rows=False)
            C._keep = (C._keep, B, Ablk)
            return C, st

        C, st = one_step()  # warm-up (also sizes the problem)
        if C is not None:
            C.free()
        times = []
        for _ in range(args.spgemm_steps):
            barrier()
            t0 = time.perf_counter()
            C, st = one_step()
            barrier()
            times.append(max_over_ranks(time.perf_counter() - t0))
            if C is not None and _ != args.spgemm_steps - 1:
                C.free()
        t_step = float(np.mean(times))
        tot = torch.tensor([st["nnz_c"] if st else 0, st["flops_f"] if st else 0], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(tot)
        nnz_c, F = int(tot[0].item()), int(tot[1].item())
        sg = {"metric": "spgemm_Mnnz_out_per_s", "value": round(nnz_c / t_step / 1e6, 1), "unit": "Mnnz/s",
              "gflops": round(2 * F / t_step / 1e9, 2), "ms_per_step": round(t_step * 1e3, 2), "steps": args.spgemm_steps,
              "n_gpus": world, "scaling": "strong",
              "workload": f"C=A*A R-MAT scale {scale} f64, sorted in, unsorted out", "rows": n,
              "nnz_a": int(st["nnz_b"]) if st else None, "flops_F": F, "nnz_c": nnz_c,
              "compression_F_over_nnzC": round(F / max(nnz_c, 1), 3),
              "timed_region": "all-gather(B) + flop count + symbolic + scan + alloc + numeric; C stays in HBM",
              "rank0_phases_ms": {k: round(st[k], 3) for k in ("ms_flops", "ms_symbolic", "ms_alloc", "ms_numeric")} if st else None,
              "rank0_rows_sym_bin": st["rows_sym_bin"][:4] if st else None,
              "rank0_rows_num_bin": st["rows_num_bin"][:5] if st else None,
              "rank0_heavy_items": st["heavy_items"] if st else None}
        if st:
            ab = spgemm_alg_bytes(st, 8)
            dev_ms = st["ms_flops"] + st["ms_symbolic"] + st["ms_alloc"] + st["ms_numeric"]
            sg["roofline"] = {"bound": "hbm", "achieved": round(ab / (dev_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                              "unit": "GB/s", "frac": round(ab / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                              "algorithmic_bytes": ab, "device_ms": round(dev_ms, 3),
                              "stream_model_bytes": 16 * st["flops_f"] + 16 * st["nnz_a"] + 12 * st["nnz_c"]}
        log(f"[spgemm] scale {scale}: {t_step * 1e3:.1f} ms/step, F={F:.3e}, nnzC={nnz_c:.3e}, "
            f"{sg['value']} Mnnz/s, {sg['gflops']} GFLOP/s")
        # CPU baseline + parity on a bounded row sample (rank 0, N = 1)
        if rank == 0 and world == 1 and not args.no_cpu_baseline and C is not None:
            import oracle

            a_host = blk.cpu_numpy()
            flops_row, _ = oracle.rows_to_threads(a_host[0], a_host[1], a_host[0], 1)
            # contiguous row sample holding ~2e8 intermediate products, taken from the middle of the matrix (bounded so that
            # the host-side comparison, a numpy lexsort of every entry, stays within seconds)
            ps = np.concatenate([[0], np.cumsum(flops_row.astype(np.float64))])
            mid = n // 2
            target = min(2e8, ps[-1] - ps[mid])
            r1 = int(np.searchsorted(ps, ps[mid] + target))
            r1 = max(mid + 1, min(r1, n))
            off = a_host[0][mid:r1 + 1] - a_host[0][mid]
            s, e = int(a_host[0][mid]), int(a_host[0][r1])
            sub = (off, a_host[1][s:e], a_host[2][s:e])
            cores = oracle.hardware_threads()
            t1 = time.perf_counter()
            wo, wi, wv = oracle.mul_hash(sub, a_host, sort_rows=False, tnum=cores)
            cpu_s = time.perf_counter() - t1
            Fs = float(ps[r1] - ps[mid])
            sg["cpu_baseline"] = {"value": round(len(wi) / cpu_s / 1e6, 2), "unit": "Mnnz/s", "cores": cores, "kind": "port",
                                  "gflops": round(2 * Fs / cpu_s / 1e9, 3),
                                  "sample": f"rows [{mid},{r1}) of A times full B: F={Fs:.3e}, nnz_out={len(wi)}, one pass, "
                                            f"C++ restatement of mul_hash with tnum={cores}"}
            # parity on the sample: row_ptr exact, per-row column sets equal, values 1e-10
            go, gi, gv = C.row_block(mid, r1).download()
            go = go - go[0]
            from tests.helpers import assert_csr_parity

            gi = gi[int(0):]
            try:
                # the view's arrays start at an aligned base: drop the leading pad entries
                pad = len(gi) - int(go[-1])
                assert_csr_parity((go, gi[pad:], gv[pad:]), (wo, wi, wv), False, 1e-10)
                sg["parity_sample"] = "ok: row_ptr exact, per-row column sets equal, values within 1e-10 (rows [%d,%d))" % (mid, r1)
            except AssertionError as ex:
                sg["parity_sample"] = f"FAILED: {ex}"
            # the same sample rows in the reference's slot order (sort_rows = 2): col_idx must be bit-identical
            try:
                B1 = DeviceCsr.from_torch(n, n, blk.row_ptr, blk.col_idx, blk.vals, True)
                C2, _ = B1.row_block(mid, r1).spgemm(B1, sort_rows=2)
                o2, i2, v2 = C2.download()
                C2.free()
                okc = bool(np.array_equal(o2, wo) and np.array_equal(i2, wi))
                relv = float(np.max(np.abs(v2 - wv) / np.abs(wv))) if len(wv) else 0.0
                sg["parity_sample_reference_order"] = (f"{'ok' if okc and relv <= 1e-10 else 'FAILED'}: row_ptr and unsorted col_idx "
                                                       f"bit-identical to the oracle's linprobe slot order = {okc}, max rel value err {relv:.2e}")
            except Exception as ex:  # noqa: BLE001
                sg["parity_sample_reference_order"] = f"FAILED: {ex}"
            log(f"[spgemm] cpu port sample {cpu_s:.1f}s on {cores} threads -> {sg['cpu_baseline']['value']} Mnnz/s; parity {sg['parity_sample']}; "
                f"reference order {sg['parity_sample_reference_order']}")
        out["spgemm"] = sg
        if "metric" not in out:  # --no-spmv: promote the SpGEMM line
            out = {**{k: sg[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "scaling")}, "warmup": 1,
                   "ms_per_step": sg["ms_per_step"], "higher_is_better": True, "vs_baseline": None, "dtype": "f64",
                   "data": "synthetic", "config": {"workload": sg["workload"]}, "spgemm": sg}

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
