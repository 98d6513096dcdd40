#!/bin/bash
# PMC passes (one counter group per run, kernel-trace only) for the sliced SpMV.  usage: tools/prof_sliced_pmc.sh rmat|powerlaw [variants-json]
set -o pipefail
export TMPDIR=/tmp
WHICH=${1:-rmat}
OUT=gpurun_out/prof_sliced_pmc_$WHICH
mkdir -p $OUT
export PROBE_VARIANTS=${2:-'[{"SPAM_SPMV_SLICED":2}]'}
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum"; do
  name=$(echo $grp | tr ' ' '_')
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 tools/spmv_probe.py $WHICH 1 > $OUT/pmc_$name.log 2>&1 || echo "pmc $grp failed" >> $OUT/errors.log
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"][:70], row["Counter_Name"])
        agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    for (k, c), (v, n) in sorted(agg.items()):
        if "spam::spmv" in k or "spam::sliced" in k or "xh_build" in k:
            print("%-70s %-30s %14.6g (n=%d)" % (k, c, v / n, n))
PY
