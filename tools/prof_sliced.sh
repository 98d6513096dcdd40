#!/bin/bash
# rocprofv3 kernel trace of the sliced SpMV on a bench matrix (diagnostic).  usage: tools/prof_sliced.sh rmat|powerlaw [variants-json]
set -o pipefail
export TMPDIR=/tmp
WHICH=${1:-rmat}
OUT=gpurun_out/prof_sliced_$WHICH
mkdir -p $OUT
export PROBE_VARIANTS=${2:-'[{"SPAM_SPMV_SLICED":2}]'}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/spmv_probe.py $WHICH 2 > $OUT/trace.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "spam::" in r["Name"]:
            print("%-90s %6s %10.3f ms total %10.2f us avg" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
