#!/usr/bin/env python3
"""bench.py — the spam_csr hot path on MI355X: CSR SpMV (primary line) + hash SpGEMM (secondary object).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one synthetic input resident in HBM:
  * SpMV  (BASELINE.json configs[1]): y = A x, A = R-MAT clipped to 10 M x 10 M, ~160 M nnz, f64.
    N > 1: every rank owns the same-sized row block (weak scaling, x replicated, no collective).
  * SpGEMM (configs[2]): C = A * A, R-MAT scale 20, sorted input, unsorted output (the reference's bench shape,
    spam_csr/src/lib.rs:402-407).  N > 1: the ranks' nnz-balanced row shards of B are all-gathered over RCCL
    (grouped send/recv, unpadded; vals in flight during the symbolic phase), A's rows are split into flop-balanced
    blocks (rows_to_threads with tnum = N, mul_hash.rs:38-64), each rank computes its block of C through
    spam_spgemm_symbolic / numeric (strong scaling; the all-gather is inside the timed region).
  * N >= 8 additionally runs configs[3]: R-MAT scale 22 (nnz(C) = 7.2e10 = 864 GB only fits sharded).

Prints ONE JSON line (rank 0).  `value` = SpMV effective GB/s summed over ranks, algorithmic bytes
(nnz*(8+4) + (rows+1)*4 + cols*8 + rows*8) / max-over-ranks step time.
The oracle (oracle/) is used only for the cpu_baseline leg and the parity spot-check; it is the C++ restatement of the
Rust reference, which cannot be built here and holds no golden vectors ("parity unpinned upstream").
"""
from __future__ import annotations

import argparse
import ctypes
import hashlib
import re
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

HBM_PEAK_GBPS = 8000.0  # MI355X spec (MI355X_MICROARCH.md)
ORACLE_NOTE = "oracle = C++ restatement of the Rust reference (unbuildable here, no golden vectors upstream: parity unpinned)"


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def spmv_alg_bytes(rows, cols, nnz, vsize):
    return nnz * (vsize + 4) + (rows + 1) * 4 + cols * vsize + rows * vsize


def spgemm_alg_bytes(st, vsize):
    # compulsory traffic: read A once, B once, write C once
    return (4 + vsize) * (st["nnz_a"] + st["nnz_b"] + st["nnz_c"]) + 4 * (st["rows"] + st["inner"] + st["rows"] + 3)


def strip_comments(text: str) -> str:
    """C / C++ source without comments and whitespace: the stamp below identifies the kernel CODE, so that a commit which
    only rewords a comment does not orphan a measurement (string literals of the kernel sources hold no comment markers)."""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r"\s+", "", text)


def kernel_source_hash(*names) -> str:
    """Stamp that ties a PMC file under profiles/ to the kernel code it was measured on."""
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(ROOT, "sparse_matrix_amd", "csrc", n), "r") as f:
            h.update(strip_comments(f.read()).encode())
    return h.hexdigest()[:16]


def _si(n: int) -> str:
    for div, suf in ((10**9, "G"), (10**6, "M"), (10**3, "k")):
        if n >= div:
            v = n / div
            return (f"{v:.0f}" if abs(v - round(v)) < 0.05 else f"{v:.1f}") + suf
    return str(n)


def spmv_workload_name(rows_arg: int, nnz_arg: int, rows: int, nnz: int) -> str:
    """The label follows the ARGUMENTS and the matrix that was generated; only the defaults are BASELINE configs[1]."""
    name = f"csr_spmv_f64_rmat_{_si(rows)}x{_si(rows)}_{_si(nnz_arg)}nnz"
    if rows_arg == 10_000_000 and nnz_arg == 160_000_000:
        return name + " (BASELINE configs[1])"
    return name + f" (NOT a BASELINE config: --spmv-rows {rows_arg} --spmv-nnz {nnz_arg}; generated {nnz} entries)"


def verified_traffic(pmc_file: str, sources, want: dict):
    """(bytes per launch, source string) from a PMC summary committed under profiles/ — only if it was taken on the
    same kernel sources and the same workload as this run; (None, reason) otherwise."""
    path = os.path.join(ROOT, "profiles", pmc_file)
    if not os.path.exists(path):
        return None, f"profiles/{pmc_file} absent"
    try:
        d = json.load(open(path))
    except Exception as ex:  # noqa: BLE001
        return None, f"profiles/{pmc_file} unreadable: {ex}"
    if d.get("kernel_source_sha16") != kernel_source_hash(*sources):
        return None, f"profiles/{pmc_file} was measured on other kernel sources ({d.get('kernel_source_sha16')}): re-run tools/prof_round.sh"
    for k, v in want.items():
        if d.get("workload", {}).get(k) != v:
            return None, f"profiles/{pmc_file} was measured on another workload ({k} = {d.get('workload', {}).get(k)}, this run {v})"
    return int(round(d["traffic_bytes_per_launch"])), f"profiles/{pmc_file} ({d.get('formula', '2*FETCH_SIZE + WRITE_SIZE')}; kernel sources {d['kernel_source_sha16']})"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--spmv-rows", type=int, default=10_000_000)
    ap.add_argument("--spmv-nnz", type=int, default=160_000_000)
    ap.add_argument("--spmv-oversample", type=float, default=1.30)
    ap.add_argument("--spgemm-scale", type=int, default=20, help="R-MAT scale of C = A*A (configs[2]); 0 disables.  With >= 8 "
                    "ranks configs[3] (scale 22: nnz(C) ~ 7e10 does not fit fewer GPUs) runs as well")
    ap.add_argument("--spgemm-steps", type=int, default=2)
    ap.add_argument("--no-scale22", action="store_true", help="skip configs[3] even with >= 8 ranks")
    ap.add_argument("--big-scale", type=int, default=22, help="scale of the second, sharded-only product (configs[3])")
    ap.add_argument("--big-min-gpus", type=int, default=8, help="ranks from which the second product runs (rehearsals lower it)")
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
    # the streaming ceiling of THIS device in THIS process: float4 copy, 1 GiB, read + write bytes
    gb = ctypes.c_double()
    _lib.check(L.spam_hip_copy_gbps(1 << 30, 10, ctypes.byref(gb)))
    copy_ceiling = float(gb.value)
    log(f"[copy] float4 device copy: {copy_ceiling:.0f} GB/s ({100 * copy_ceiling / HBM_PEAK_GBPS:.1f}% of 8 TB/s)")

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
        # what a one-shot caller pays: the first SpMV on a fresh handle (plain tile path, no analysis) ...
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        dA.spmv_dev(x.data_ptr(), y.data_ptr())
        torch.cuda.synchronize()
        ms_first = (time.perf_counter() - t1) * 1e3
        # ... and what the analysis costs, once per matrix (outside the timed steps, reported)
        _lib.check(L.spam_hip_trim())
        free0 = torch.cuda.mem_get_info()[0]
        t1 = time.perf_counter()
        dA.spmv_prepare(x.data_ptr())
        torch.cuda.synchronize()
        ms_prepare = (time.perf_counter() - t1) * 1e3
        _lib.check(L.spam_hip_trim())  # analysis temporaries back to the driver: what stays is the plan
        plan_bytes = int(free0 - torch.cuda.mem_get_info()[0])
        if plan_bytes < 0:  # (ranks sharing one GPU in a rehearsal: other processes' allocations move the free counter)
            plan_bytes = None
        # the same analysis once more with the library's allocator warm (the first one, right after spam_hip_trim, paid a
        # hipMalloc for every temporary and for the plan itself): what a host that prepares one matrix after another sees
        ms_prepare_cold = ms_prepare
        if rank == 0 and world == 1:
            dA.spmv_prepare(x.data_ptr())
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            dA.spmv_prepare(x.data_ptr())
            torch.cuda.synchronize()
            ms_prepare = (time.perf_counter() - t1) * 1e3

        def spmv_step():
            if sharded:  # the exchange step of a row-sharded SpMV: all-gather x, inside the timed region
                xg = spdist.allgather_vector(x_local, args.spmv_rows, rank, world, dist)
                dA.spmv_dev(xg.data_ptr(), y.data_ptr())
            else:
                dA.spmv_dev(x.data_ptr(), y.data_ptr())

        for _ in range(args.warmup):
            spmv_step()
        # device time of the K steps: ONE event pair around the loop, on the stream the kernels are launched on (an
        # event pair per step put 2 K records between the launches: 7 us per step of host time at 0.55 ms per step)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            spmv_step()
        ev1.record()
        barrier()
        wall = time.perf_counter() - t0
        wall = max_over_ranks(wall)
        kern_ms = ev0.elapsed_time(ev1) / args.steps
        ms_per_step = wall * 1e3 / args.steps
        value = (job_bytes if sharded else world * nbytes) / (wall / args.steps) / 1e9
        achieved = nbytes / (kern_ms * 1e-3) / 1e9
        out.update({
            "metric": "spmv_effective_GB_per_s", "value": round(value, 1), "unit": "GB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": spmv_workload_name(args.spmv_rows, args.spmv_nnz, A.rows, A.nnz), "rows": A.rows,
                       "cols": A.cols, "nnz": A.nnz, "avg_nnz_row": round(stA["avg_nnz_row"], 2),
                       "max_nnz_row": stA["max_nnz_row"], "gflops": round(2 * A.nnz * world / (wall / args.steps) / 1e9, 1),
                       "sharding": ("one matrix row-sharded, x all-gathered every step" if sharded else
                                    "one matrix per rank, x replicated, no collective") if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                         "kernel": "spmv_tile_kernel(+fixup)", "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes": nbytes, "copy_ceiling_GBps": round(copy_ceiling, 1),
                         "frac_of_copy_ceiling": round(achieved / copy_ceiling, 4)},
            "spmv_plan_cost": {"first_spmv_unprepared_ms": round(ms_first, 3), "prepare_ms": round(ms_prepare, 1),
                               "prepare_ms_cold_allocator": round(ms_prepare_cold, 1),
                               "plan_extra_bytes": plan_bytes, "matrix_bytes": A.nnz * 12 + (A.rows + 1) * 12,
                               "note": "prepare = spam_spmv_prepare, once per matrix, outside the timed steps (prepare_ms: library "
                                       "allocator warm, i.e. a host preparing one matrix after another; prepare_ms_cold_allocator: "
                                       "right after spam_hip_trim, every temporary and the plan itself pay hipMalloc); an "
                                       "unprepared handle multiplies through the plain tile path"},
        })
        # what the gather allows on this chip (tools/gather_bench.hip, tools/xcd_gather_bench.hip): 245 G gathers/s when
        # they hit the XCD's L2 (one request per lane: the L2's request rate), 52 G/s when they miss (a 128-byte line
        # crosses the fabric per miss).  Plain path: one 3 MB hot table makes `hot_cover` of them hit.  XCD-sliced path:
        # eight per-XCD tables make `cover_table` of the long rows' gathers hit (230 G/s measured with every workgroup on
        # its own XCD's slice), the rest and the short rows' misses pay the fabric rate.
        info = dA.spmv_plan_info()
        sl = info["sliced"]
        rg = sl.get("row_groups", {"active": False})
        plan_mode = "xcd_sliced_row_groups" if sl["active"] and rg["active"] else "xcd_sliced" if sl["active"] else "plain"
        if sl["active"] and rg["active"]:
            # row-group form: every gather is served by L1 / the XCD's L2 except the cold part's; the step runs at the
            # fabric's byte rate, so the model is a byte count: entry stream + pair sums written and read + one 128-byte
            # line per cold entry + the table rebuild (x, column map, tables) + y (zero-fill, then ~a 64-byte sector per
            # long row and the short rows' range once)
            cold = sl["nnz_long"] * (1.0 - sl["cover_table"])
            model_bytes = (rg["stream_bytes"] + 16 * rg["accumulators"] + 128 * cold + A.cols * 12 + 16 * sl["table_slots_per_xcd"] * 8
                           + A.rows * 8 + sl["long_rows"] * 6 + A.rows * 8)
            model_ms = model_bytes / 6.3e12 * 1e3
            out["roofline"]["kernel"] = ("XCD-sliced SpMV, row-group form: sliced_build_tables + spmv_rg_kernel + "
                                         "spmv_rg_reduce_kernel (whole step)")
            out["roofline"]["byte_model"] = {"mode": plan_mode, "entry_stream_bytes": rg["stream_bytes"],
                                             "pair_sums": rg["accumulators"], "cold_entries": int(cold),
                                             "model_bytes": int(model_bytes), "fabric_rate_Bps": 6.3e12,
                                             "bound_ms": round(model_ms, 4), "frac_of_bound": round(model_ms / kern_ms, 4),
                                             "source": "profiles/r03_stream_read_microbench.txt (6.4-7.0 TB/s ticketed reads), "
                                                       "MI355X_MICROARCH.md (HBM ~6.3 TB/s achievable)"}
            out["config"]["spmv_plan"] = {"mode": plan_mode, "sliced_rows": sl["long_rows"], "sliced_entries": sl["nnz_long"],
                                          "table_slots_per_xcd": sl["table_slots_per_xcd"],
                                          "cover_8_tables": round(sl["cover_table"], 4), "cover_1_table": round(sl["cover_single"], 4),
                                          "row_groups": rg["groups"], "blocks": rg["blocks"], "pair_sums": rg["accumulators"],
                                          "entry_stream_bytes": rg["stream_bytes"], "plan_bytes": rg["plan_bytes"],
                                          "prepare_ms_plain": round(sl["ms_plain"], 4), "prepare_ms_sliced": round(sl["ms_sliced"], 4)}
        elif sl["active"]:
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
                                          "prepare_ms_plain": round(sl["ms_plain"], 4), "prepare_ms_sliced": round(sl["ms_sliced"], 4)}
        else:
            hits = info["hot_cover"] * A.nnz if info["hot_columns"] else 0.0
            model_ms = max(nbytes / 5.8e12, hits / 245e9 + (A.nnz - hits) / 52e9) * 1e3
            out["roofline"]["gather_model"] = {"mode": "plain", "hot_columns": info["hot_columns"], "hot_cover": round(info["hot_cover"], 4),
                                               "l2_hit_gathers_per_s": 245e9, "l2_miss_gathers_per_s": 52e9,
                                               "bound_ms": round(model_ms, 4), "frac_of_bound": round(model_ms / kern_ms, 4),
                                               "source": "profiles/r01_gather_microbench.txt"}
            out["config"]["spmv_plan"] = {"mode": "plain", "hot_columns": info["hot_columns"]}
        # HBM/fabric bytes per launch come from a separate rocprofv3 --pmc run (tools/prof_bench.sh), never from this
        # process; they are printed only when that run was taken on these kernel sources, this workload and this plan
        tr, src = verified_traffic("spmv_pmc_latest.json", ("spam_spmv.hip", "spam_spmv_rg.hip"),
                                   {"rows": A.rows, "nnz": A.nnz, "plan": plan_mode})
        out["roofline"]["traffic"] = tr
        out["roofline"]["traffic_source"] = src
        out["roofline"]["traffic_measured_in_this_run"] = False  # always read from profiles/ (a separate --pmc run), or null
        log(f"[spmv] {ms_per_step:.4f} ms/step wall, {kern_ms:.4f} ms device, {achieved:.0f} GB/s "
            f"({100 * achieved / HBM_PEAK_GBPS:.1f}% of 8 TB/s, {100 * achieved / copy_ceiling:.1f}% of the copy ceiling); "
            f"first unprepared SpMV {ms_first:.2f} ms, prepare {ms_prepare:.0f} ms, plan {(plan_bytes or 0) / 1e9:.2f} GB")
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
                             "spmv_zero_rows_exact": bool(np.all(yh[~nzm] == 0)),
                             "status": "partial: " + ORACLE_NOTE + "; SpMV itself does not exist upstream (defined as mul_hash with an n x 1 rhs)"}
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
            d5.spmv_prepare(x5.data_ptr())
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
                                        "plan": ("xcd_sliced_row_groups" if d5.spmv_plan_info()["sliced"]["row_groups"]["active"] else "xcd_sliced")
                                        if d5.spmv_plan_info()["sliced"]["active"] else "plain"}
            log(f"[spmv f32 power-law] {ms5:.4f} ms/step, {b5 / (ms5 * 1e-3) / 1e9:.0f} GB/s")
            del d5, A5, x5, y5
            torch.cuda.empty_cache()

    # ------------------------------------------------------------------ SpGEMM (secondary) ---------
    def run_spgemm(scale: int, steps: int, with_cpu: bool):
        n = 1 << scale
        # every rank starts with a row block of A (= its block of B) holding about the same number of entries: the
        # blocks are cut from a coarse row histogram of the equal-row blocks (setup, untimed); values depend on
        # global coordinates, so any split gives the same matrix
        t0 = time.time()
        offs0 = spdist.equal_row_blocks(n, world)
        blk = synth.rmat(scale, dtype=torch.float64, device=dev, row_range=(offs0[rank], offs0[rank + 1]) if world > 1 else None)
        if world > 1:
            bucket = max(1, n >> 12)
            cnt = torch.zeros((n + bucket - 1) // bucket, dtype=torch.int64, device=dev)
            rows_g = torch.arange(offs0[rank], offs0[rank + 1], dtype=torch.int64, device=dev)
            cnt.index_add_(0, rows_g // bucket, blk.row_ptr[1:] - blk.row_ptr[:-1])
            dist.all_reduce(cnt)
            offs0 = spdist.nnz_balanced_blocks(cnt, bucket, n, world)
            del blk
            blk = synth.rmat(scale, dtype=torch.float64, device=dev, row_range=(offs0[rank], offs0[rank + 1]))
        torch.cuda.synchronize()
        log(f"[spgemm {scale}] rank block rows [{offs0[rank]},{offs0[rank + 1]}) nnz {blk.nnz} generated in {time.time() - t0:.1f}s")
        xfer = {}
        # N > 1 on RCCL: the exchange runs BEHIND THE C ABI (libspam_hip_rccl.so: spam_csr_allgather_rccl +
        # spam_spgemm_rank_step_rccl — what the Rust host calls; torch.distributed only hands the 128-byte unique id
        # around and provides the bench's barrier).  SPAM_EXCHANGE=torch selects the torch.distributed all_to_all of
        # dist.py instead (and is what a gloo rehearsal on one GPU uses: RCCL refuses two ranks on one device).
        use_cabi = world > 1 and backend == "nccl" and os.environ.get("SPAM_EXCHANGE", "cabi") == "cabi"
        if use_cabi and comm_box[0] is None:
            # every rank must agree on the path: a rank whose communicator cannot be made (library missing, RCCL refusing the
            # topology) would otherwise leave the others waiting in a collective
            err = None
            try:
                comm_box[0] = spdist.RcclComm.from_torch_dist(dist, rank, world)
            except Exception as ex:  # noqa: BLE001
                err = f"{type(ex).__name__}: {ex}"
            okf = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            if int(okf.item()) == 0:
                if comm_box[0] is not None:
                    comm_box[0].free()
                comm_box[0] = False
                log(f"[spgemm] C-ABI exchange unavailable on at least one rank ({err}); falling back to torch.distributed all_to_all")
        if use_cabi and comm_box[0]:
            comm = comm_box[0]
            shard = DeviceCsr.from_torch(offs0[rank + 1] - offs0[rank], n, blk.row_ptr, blk.col_idx, blk.vals, True)
        else:
            use_cabi = False

        def one_step_cabi(sort_rows=False):
            B, received = comm.allgather_csr(shard, offs0)  # values still in flight: only the numeric phase waits
            xfer.update({"bytes_received": received, "mode": "C ABI: grouped ncclSend/ncclRecv per peer (libspam_hip_rccl.so)",
                         "flop_count": "sharded over the ranks' initial row blocks, per-row counts all-gathered (8 B/row)"})
            C, st, (mylo, myhi) = comm.rank_step(B, B, offs0, spdist.ROW_OVERHEAD_AUTO, sort_rows)
            if C is None:
                return None, None
            C._keep = (C._keep, B)
            return C, st

        def one_step(sort_rows=False):
            """all-gather B (vals in flight during the symbolic phase), flop-balanced row block of A, two-phase SpGEMM.
            Returns (C, stats)."""
            if use_cabi:
                return one_step_cabi(sort_rows)
            g = spdist.allgather_csr(blk.row_ptr, blk.col_idx, blk.vals, offs0, rank, world, dist, async_vals=True)
            xfer.update(g.stats)
            g.wait_structure()
            B = DeviceCsr.from_torch(n, n, g.row_ptr, g.col_idx, g.vals, True)
            if world > 1:
                # rows_to_threads' flop count and partition rule (mul_hash.rs:39-62) on F + a per-row cost, on the device
                offs = B.balanced_row_blocks(B, world, spdist.ROW_OVERHEAD_AUTO)
                mylo, myhi = int(offs[rank]), int(offs[rank + 1])
                if myhi <= mylo:
                    g.wait_values()
                    return None, None
                Ablk = B.row_block(mylo, myhi)
            else:
                Ablk = B
            plan, _ = Ablk.spgemm_symbolic(B)
            g.wait_values()
            C, st = DeviceCsr.spgemm_numeric(plan, sort_rows=sort_rows)
            C._keep = (C._keep, B, Ablk, g)
            return C, st

        C, st = one_step()  # warm-up (also sizes the problem and warms the allocator)
        if C is not None:
            C.free()
        times = []
        for i in range(steps):
            barrier()
            t0 = time.perf_counter()
            C, st = one_step()
            barrier()
            times.append(max_over_ranks(time.perf_counter() - t0))
            if C is not None and i != steps - 1:
                C.free()
        t_step = float(np.mean(times))
        tot = torch.tensor([st["nnz_c"] if st else 0, st["flops_f"] if st else 0], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(tot)
        nnz_c, F = int(tot[0].item()), int(tot[1].item())
        sg = {"metric": "spgemm_Mnnz_out_per_s", "value": round(nnz_c / t_step / 1e6, 1), "unit": "Mnnz/s",
              "gflops": round(2 * F / t_step / 1e9, 2), "ms_per_step": round(t_step * 1e3, 2), "steps": steps,
              "n_gpus": world, "scaling": "strong",
              "workload": f"C=A*A R-MAT scale {scale} f64, sorted in, unsorted out" + (" (BASELINE configs[2])" if scale == 20 else " (BASELINE configs[3])" if scale == 22 else ""), "rows": n,
              "nnz_a": int(st["nnz_b"]) if st else None, "flops_F": F, "nnz_c": nnz_c,
              "compression_F_over_nnzC": round(F / max(nnz_c, 1), 3),
              "timed_region": "all-gather(B) + flop count + symbolic + scan + alloc + numeric; C stays in HBM; the output "
                              "block is reused from the library's allocator (see cold_step_ms for a first call)",
              "exchange": xfer if world > 1 else None,
              "rank0_phases_ms": {k: round(st[k], 3) for k in ("ms_flops", "ms_symbolic", "ms_alloc", "ms_numeric")} if st else None,
              "rank0_rows_sym_bin": st["rows_sym_bin"][:4] if st else None,
              "rank0_rows_num_bin": st["rows_num_bin"][:5] if st else None,
              "rank0_heavy_rows": st["heavy_items"] if st else None,
              "rank0_heavy_groups": st["heavy_dense_groups"] if st else None}
        if st:
            ab = spgemm_alg_bytes(st, 8)
            dev_ms = st["ms_flops"] + st["ms_symbolic"] + st["ms_alloc"] + st["ms_numeric"]
            tr, src = (None, "N > 1") if world > 1 else verified_traffic(
                "spgemm_pmc_latest.json", ("spam_spgemm.hip",), {"scale": scale})
            sg["roofline"] = {"bound": "hbm", "achieved": round(ab / (dev_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                              "unit": "GB/s", "frac": round(ab / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": tr,
                              "traffic_source": src, "algorithmic_bytes": ab, "device_ms": round(dev_ms, 3),
                              "copy_ceiling_GBps": round(copy_ceiling, 1),
                              "stream_model_bytes": 16 * st["flops_f"] + 16 * st["nnz_a"] + 12 * st["nnz_c"]}
        log(f"[spgemm {scale}] {t_step * 1e3:.1f} ms/step, F={F:.3e}, nnzC={nnz_c:.3e}, "
            f"{sg['value']} Mnnz/s, {sg['gflops']} GFLOP/s")
        # the B2 = true variant (rows of C sorted by column, mul_hash.rs:164-175): one more step, informational
        if C is not None:
            C.free()
        barrier()
        t0 = time.perf_counter()
        C, st_sorted = one_step(sort_rows=True)
        barrier()
        t_sorted = max_over_ranks(time.perf_counter() - t0)
        sg["sorted_out"] = {"ms_per_step": round(t_sorted * 1e3, 2), "Mnnz_out_per_s": round(nnz_c / t_sorted / 1e6, 1),
                            "steps": 1, "note": "B2 = true: rows of up to 4096 entries are sorted in LDS after the numeric "
                                                "phase; longer rows come out in column order anyway"}
        # a first call: nothing cached in the library's allocator (hipMalloc of the output inside the step)
        if world == 1:
            keep = C
            C = None
            keep.free()
            _lib.check(L.spam_hip_trim())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            C, st_cold = one_step()
            torch.cuda.synchronize()
            sg["cold_step_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
            sg["cold_step_alloc_ms"] = round(st_cold["ms_alloc"], 1)
        # CPU baseline + parity on a bounded row sample (rank 0, N = 1)
        if with_cpu and C is not None:
            import oracle
            from tests.helpers import spgemm_rows_vs_oracle

            a_host = blk.cpu_numpy()
            B1 = DeviceCsr.from_torch(n, n, blk.row_ptr, blk.col_idx, blk.vals, True)
            flops_row, _ = oracle.rows_to_threads(a_host[0], a_host[1], a_host[0], 1)
            ps = np.concatenate([[0], np.cumsum(flops_row.astype(np.float64))])
            mid = n // 2
            sample_f = 2e9 if scale >= 20 else 2e8  # products in the CPU sample: ~10 s of work for 256 threads over three passes
            r1 = max(mid + 1, min(int(np.searchsorted(ps, ps[mid] + min(sample_f, ps[-1] - ps[mid]))), n))
            s, e = int(a_host[0][mid]), int(a_host[0][r1])
            sub = (a_host[0][mid:r1 + 1] - a_host[0][mid], a_host[1][s:e], a_host[2][s:e])
            cores = oracle.hardware_threads()
            cpu_ts = []
            for _ in range(3):
                t1 = time.perf_counter()
                wo, wi, wv = oracle.mul_hash(sub, a_host, sort_rows=False, tnum=cores)
                cpu_ts.append(time.perf_counter() - t1)
            cpu_s = float(np.median(cpu_ts))
            Fs = float(ps[r1] - ps[mid])
            sg["cpu_baseline"] = {"value": round(len(wi) / cpu_s / 1e6, 2), "unit": "Mnnz/s", "cores": cores, "kind": "port",
                                  "gflops": round(2 * Fs / cpu_s / 1e9, 3),
                                  "sample": f"rows [{mid},{r1}) of A times full B: F={Fs:.3e}, nnz_out={len(wi)}, median of 3 passes "
                                            f"({', '.join(f'{t:.2f}' for t in cpu_ts)} s), C++ restatement of mul_hash with tnum={cores}"}
            # the same rows of the device product against the oracle (row_ptr exact, column sets, 1e-10; reference slot
            # order bit-exact) — also a -m gpu test (tests/test_spgemm_gpu.py::test_full_size_row_sum_property)
            try:
                pr1, _, _ = spgemm_rows_vs_oracle(oracle, B1, B1, C, a_host, a_host, mid, 2e8)  # (the parity sample: 2e8 products)
                sg["parity_sample"] = (f"rows [{mid},{pr1}): row_ptr exact, per-row column sets equal, values within 1e-10, reference slot "
                                       f"order (sort_rows=2) col_idx bit-identical; status partial: {ORACLE_NOTE}")
            except AssertionError as ex:
                sg["parity_sample"] = f"FAILED: {ex}"
            log(f"[spgemm {scale}] cpu port sample {cpu_s:.1f}s on {cores} threads -> {sg['cpu_baseline']['value']} Mnnz/s; parity {sg['parity_sample']}")
        if C is not None:
            C.free()
        del blk
        torch.cuda.empty_cache()
        _lib.check(L.spam_hip_trim())
        return sg

    comm_box = [None]  # the C-ABI communicator, made once and shared by both products
    if args.spgemm_scale > 0:
        sg = run_spgemm(args.spgemm_scale, args.spgemm_steps, rank == 0 and world == 1 and not args.no_cpu_baseline)
        out["spgemm"] = sg
        if world >= args.big_min_gpus and not args.no_scale22 and args.spgemm_scale != args.big_scale:
            out[f"spgemm_scale{args.big_scale}"] = run_spgemm(args.big_scale, 1, False)
        if "metric" not in out:  # --no-spmv: promote the SpGEMM line
            out = {**{k: sg[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "scaling")}, "warmup": 1,
                   "ms_per_step": sg["ms_per_step"], "higher_is_better": True, "vs_baseline": None, "dtype": "f64",
                   "data": "synthetic", "config": {"workload": sg["workload"]}, **{k: v for k, v in out.items()}}

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
