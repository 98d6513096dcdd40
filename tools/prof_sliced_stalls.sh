#!/bin/bash
# Derived rocprofv3 counters (one per pass, kernel-trace only) for the sliced SpMV kernels: what are the CUs waiting for?
set -o pipefail
export TMPDIR=/tmp
WHICH=${1:-rmat}
OUT=gpurun_out/prof_sliced_stalls_$WHICH
mkdir -p $OUT
export PROBE_VARIANTS=${2:-'[{"SPAM_SPMV_SLICED":2}]'}
for grp in "MemUnitBusy" "MemUnitStalled" "LDSBankConflict" "VALUBusy" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "TA_BUSY_avr" "TCP_PENDING_STALL_CYCLES_sum" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  name=$(echo $grp | tr ' ' '_')
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 tools/spmv_probe.py $WHICH 1 > $OUT/pmc_$name.log 2>&1 || echo "pmc $grp failed" >> $OUT/errors.log
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void spam::", "")[:44], row["Counter_Name"])
        agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    for (k, c), (v, n) in sorted(agg.items()):
        if "sliced_pipe" in k or "short_rows" in k or "spmv_tile" in k or "spmv_rg" in k or "build_tables" in k:
            print("%-60s %-32s %14.6g (n=%d)" % (k, c, v / n, n))
try:
    print(open(out + "/errors.log").read())
except Exception:
    pass
PY
