#!/bin/bash
# PMC passes (one counter group per run, kernel-trace only) over the LDS-accumulator SpMV prototype.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_ylds_pmc
mkdir -p $OUT
export YLDS_CONFIGS=${1:-'[{"nb":256,"rmax":16384,"nt":1024,"wgcu":1,"k":4}]'}
export YLDS_ITERS=3
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum" "TA_BUSY_avr" "TCP_PENDING_STALL_CYCLES_sum" "MemUnitStalled"; do
  name=$(echo $grp | tr ' ' '_')
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 tools/ylds_probe.py rmat > $OUT/pmc_$name.log 2>&1 || echo "pmc $grp failed" >> $OUT/errors.log
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"][:50], row["Counter_Name"])
        agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    for (k, c), (v, n) in sorted(agg.items()):
        if "ylds" in k:
            print("%-50s %-32s %14.6g (n=%d)" % (k, c, v / n, n))
try:
    print(open(out + "/errors.log").read())
except Exception:
    pass
PY
