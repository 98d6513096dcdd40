#!/bin/bash
# rocprofv3 kernel trace of ONE rank block of C = A*A (block r of P at a given scale) and of the product in one piece:
# which kernels of a block do not shrink with the block.  usage: tools/prof_block.sh 20 8 3
set -o pipefail
export TMPDIR=/tmp
SCALE=${1:-20}; P=${2:-8}; R=${3:-3}
OUT=gpurun_out/prof_block_${SCALE}_${P}_${R}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/prof_block.py $SCALE $P $R > $OUT/trace.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if "spam::" in r["Name"]:
            print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>4s} total {float(r["TotalDurationNs"])/1e6:9.3f} ms avg {float(r["AverageNs"])/1e3:10.1f} us')
PY
tail -2 $OUT/trace.log
