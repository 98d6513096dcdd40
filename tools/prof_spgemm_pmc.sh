#!/bin/bash
# PMC passes (kernel-trace + one counter group per run) over the SpGEMM probe.
set -o pipefail
export TMPDIR=/tmp
SCALE=${1:-20}
OUT=gpurun_out/prof_spgemm_pmc_$SCALE
mkdir -p $OUT
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-60)
  echo "pmc pass: $grp"
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 tools/spgemm_probe.py $SCALE 1 > $OUT/pmc_$name.log 2>&1 || echo "pmc $grp failed" >> $OUT/errors.log
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"][:48], row["Counter_Name"])
        agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    for (k, c), (v, n) in sorted(agg.items()):
        if "heavy_kernel" in k or "hash_kernel<double, 1024" in k or "sym_hash_kernel<256" in k:
            print("%-48s %-28s %14.6g (n=%d)" % (k, c, v / n, n))
PY
cat $OUT/errors.log 2>/dev/null
