#!/bin/bash
# Kernel stats + PMC passes (one counter group per run) over the symbolic phase alone (tools/sym_probe.py).
set -o pipefail
export TMPDIR=/tmp
SCALE=${1:-20}
OUT=gpurun_out/prof_sym_$SCALE
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/sym_probe.py $SCALE 2 > $OUT/trace.log 2>&1 || echo "trace failed" >> $OUT/errors.log
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_WAVES" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-60)
  echo "pmc pass: $grp"
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 tools/sym_probe.py $SCALE 1 > $OUT/pmc_$name.log 2>&1 || echo "pmc $grp failed" >> $OUT/errors.log
done
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "spam::" in r["Name"]:
            print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>4s} total {float(r["TotalDurationNs"])/1e6:9.3f} ms avg {float(r["AverageNs"])/1e3:10.1f} us')
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"][:40], row["Counter_Name"])
        agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    for (k, c), (v, n) in sorted(agg.items()):
        if "sym_heavy" in k or "sym_row" in k:
            print("%-40s %-28s %14.6g (n=%d)" % (k, c, v / n, n))
PY
cat $OUT/errors.log 2>/dev/null
