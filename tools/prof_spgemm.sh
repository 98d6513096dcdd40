#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
SCALE=${1:-18}
OUT=gpurun_out/prof_spgemm_$SCALE
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/spgemm_probe.py $SCALE 2 > $OUT/trace.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    print("== kernel stats (spam kernels)", f)
    for r in rows:
        if "spam::" in r["Name"]:
            print(f'{r["Name"][:110]:110s} calls {r["Calls"]:>4s} total {float(r["TotalDurationNs"])/1e6:9.3f} ms avg {float(r["AverageNs"])/1e3:10.1f} us')
PY
cat $OUT/trace.log | tail -5
