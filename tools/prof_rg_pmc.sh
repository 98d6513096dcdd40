#!/bin/bash
# PMC passes (one counter group per run, kernel-trace only) over the row-group SpMV (tools/spmv_probe.py, one round).
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_rg_pmc
mkdir -p $OUT
export PROBE_VARIANTS=${1:-'[{"SPAM_SPMV_SLICED":2},{"SPAM_SPMV_SLICED":2,"SPAM_SPMV_RG_SHORT":0}]'}
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum"; do
  name=$(echo $grp | tr ' ' '_')
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 tools/spmv_probe.py rmat 1 > $OUT/pmc_$name.log 2>&1 || echo "pmc $grp failed" >> $OUT/errors.log
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    # the probe runs the variants one after another: split the dispatches of spmv_rg_kernel into halves
    agg = collections.defaultdict(list)
    for row in rows:
        k = row["Kernel_Name"]
        if any(t in k for t in ("spmv_rg_kernel", "short_rows", "build_tables", "sliced_reduce")):
            agg[(k[:48], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        h = len(v) // 2
        print("%-48s %-24s first variant %14.6g   second variant %14.6g  (n=%d)" % (k, c, sum(v[:h]) / max(h, 1), sum(v[h:]) / max(len(v) - h, 1), len(v)))
try:
    print(open(out + "/errors.log").read())
except Exception:
    pass
PY
