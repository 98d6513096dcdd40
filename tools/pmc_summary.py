#!/usr/bin/env python3
"""Summarise the rocprofv3 output of tools/prof_round.sh: kernel-time table, per-kernel PMC averages and the two stamped
JSON files bench.py reads its `roofline.traffic` from.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies
128-byte fabric reads at 64 B (MI355X_MICROARCH.md §HBM), so read bytes = 2 * FETCH_SIZE."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]


def src_hash(*names):
    """the same stamp as bench.py's kernel_source_hash: the kernel code without comments and whitespace"""
    import re
    h = hashlib.sha256()
    for n in names:
        text = open(os.path.join(ROOT, "sparse_matrix_amd", "csrc", n), "r").read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        h.update(re.sub(r"\s+", "", text).encode())
    return h.hexdigest()[:16]


def pmc(prefix):
    """{(kernel short name, counter): (sum, dispatches)} over all passes whose directory starts with prefix"""
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in sorted(glob.glob(f"{out}/{prefix}*/**/*counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if "spam::" not in name:
                continue
            short = name.split("spam::")[1].replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
            agg[(short, row["Counter_Name"])][0] += float(row["Counter_Value"])
            agg[(short, row["Counter_Name"])][1] += 1
    return agg


lines = []
for f in sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)):
    lines.append("== rocprofv3 --kernel-trace --stats: python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline")
    lines.append("%-100s %6s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for r in csv.DictReader(open(f)):
        if "spam::" in r["Name"]:
            lines.append("%-100s %6s %12.3f %12.2f %7s" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                            float(r["AverageNs"]) / 1e3, r["Percentage"]))
bench = {}
try:
    for ln in open(out + "/bench_under_rocprof.json"):
        if ln.startswith("{"):
            bench = json.loads(ln)
except OSError:
    pass

# ---- SpMV step ------------------------------------------------------------------------------------
sp = pmc("spmv_pmc_")
mode = bench.get("config", {}).get("spmv_plan", {}).get("mode", "?")
step_kernels = (["sliced_build_tables_kernel", "sliced_build_tables_vec_kernel", "spmv_short_rows_kernel", "spmv_rg_kernel", "spmv_rg_reduce_kernel"]
                if mode == "xcd_sliced_row_groups" else
                ["sliced_build_tables_kernel", "xh_build_kernel", "spmv_short_rows_kernel", "spmv_sliced_pipe_kernel",
                 "sliced_fixup_kernel", "sliced_reduce_kernel"] if mode == "xcd_sliced" else
                ["xh_build_kernel", "spmv_tile_kernel", "spmv_fixup_kernel"])
lines.append("\n== SpMV PMC, per-dispatch averages (bench.py --steps 5 --warmup 1 --spgemm-scale 0); plan: " + mode)
for (k, c), (v, n) in sorted(sp.items()):
    if "spmv" in k or "sliced" in k or "xh_build" in k:
        lines.append("%-40s %-24s %16.6g (n=%d)" % (k, c, v / n, n))
fetch = {k: sp[(k, "FETCH_SIZE")][0] / sp[(k, "FETCH_SIZE")][1] for k in step_kernels if (k, "FETCH_SIZE") in sp}
write = {k: sp[(k, "WRITE_SIZE")][0] / sp[(k, "WRITE_SIZE")][1] for k in step_kernels if (k, "WRITE_SIZE") in sp}
if fetch and write:
    traffic = (2 * sum(fetch.values()) + sum(write.values())) * 1024
    json.dump({"kernels": step_kernels, "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
               "formula": "2*FETCH_SIZE + WRITE_SIZE summed over the kernels of one step (gfx950: FETCH_SIZE counts 128-byte reads as 64)",
               "traffic_bytes_per_launch": traffic, "kernel_source_sha16": src_hash("spam_spmv.hip", "spam_spmv_rg.hip"),
               "workload": {"rows": bench.get("config", {}).get("rows"), "nnz": bench.get("config", {}).get("nnz"), "plan": mode},
               "algorithmic_bytes": bench.get("roofline", {}).get("algorithmic_bytes")},
              open(out + "/spmv_pmc_latest.json", "w"), indent=1)
    lines.append("SpMV step: 2*FETCH + WRITE = %.3f GB per launch" % (traffic / 1e9))

# ---- one scale-20 SpGEMM ----------------------------------------------------------------------------
sg = pmc("spgemm_pmc_")
lines.append("\n== SpGEMM PMC, ONE product C = A*A at R-MAT scale 20 (tools/spgemm_probe.py 20 1): totals per kernel")
tot_f = tot_w = 0.0
for (k, c), (v, n) in sorted(sg.items()):
    lines.append("%-40s %-24s %16.6g (dispatches %d)" % (k, c, v, n))
    if c == "FETCH_SIZE":
        tot_f += v
    if c == "WRITE_SIZE":
        tot_w += v
if tot_f and tot_w:
    traffic = (2 * tot_f + tot_w) * 1024
    json.dump({"formula": "2*FETCH_SIZE + WRITE_SIZE summed over every kernel of one product",
               "FETCH_SIZE_KiB_total": tot_f, "WRITE_SIZE_KiB_total": tot_w, "traffic_bytes_per_launch": traffic,
               "kernel_source_sha16": src_hash("spam_spgemm.hip"), "workload": {"scale": 20}},
              open(out + "/spgemm_pmc_latest.json", "w"), indent=1)
    lines.append("SpGEMM product: 2*FETCH + WRITE = %.1f GB" % (traffic / 1e9))
    hv = {c: v for (k, c), (v, n) in sg.items() if k == "num_heavy_kernel"}
    if "SQ_LDS_IDX_ACTIVE" in hv and hv["SQ_LDS_IDX_ACTIVE"]:
        lines.append("num_heavy_kernel: LDS bank-conflict cycles / LDS active cycles = %.3f; wave cycles waiting = %.3f" % (
            hv.get("SQ_LDS_BANK_CONFLICT", 0) / hv["SQ_LDS_IDX_ACTIVE"],
            hv.get("SQ_WAIT_ANY", 0) / max(hv.get("SQ_WAVE_CYCLES", 1), 1)))
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
