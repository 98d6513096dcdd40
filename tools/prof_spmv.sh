#!/bin/bash
# rocprofv3 passes for the SpMV probe: kernel trace + stats, then one PMC group per pass (never combined with tracing domains other than kernel-trace).
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_spmv
mkdir -p $OUT
WHICH=${1:-rmat}
export PROBE_VARIANTS='[{"SPAM_SPMV_NPT":16,"SPAM_SPMV_NT":1}]'
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/spmv_probe.py $WHICH 1 > $OUT/trace.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_REQ_sum TCC_READ_sum"; do
  name=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 tools/spmv_probe.py $WHICH 1 > $OUT/pmc_$name.log 2>&1 || echo "pmc $grp failed" >> $OUT/errors.log
done
python3 - <<'PY'
import csv, glob, collections, os
out = "gpurun_out/prof_spmv"
for f in sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)):
    print("==", f); print(open(f).read()[:3000])
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"][:60], row["Counter_Name"])
        agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    print("==", f)
    for (k, c), (v, n) in sorted(agg.items()):
        if "spmv" in k: print(f"  {k:60s} {c:32s} per-dispatch {v / n:.4g}  (n={n})")
PY
