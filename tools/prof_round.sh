#!/bin/bash
# rocprofv3 evidence for one round (run on the GPU box from the repo root):  tools/prof_round.sh r02
#   1. kernel trace + stats of the default bench run;
#   2. PMC passes (one counter group per run, kernel-trace only) over the SpMV leg of bench.py and over ONE scale-20
#      SpGEMM (tools/spgemm_probe.py);
#   3. tools/pmc_summary.py turns them into gpurun_out/prof_<tag>/{summary.txt, spmv_pmc_latest.json,
#      spgemm_pmc_latest.json}, the JSON files stamped with the kernel-source hash and the workload they were taken
#      on (bench.py prints their traffic only when both match its own run).  Copy them to profiles/ to have them judged.
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.log || echo "trace run failed" >> $OUT/errors.log
echo "trace done" > $OUT/progress.txt
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum"; do
  name=$(echo $grp | tr ' ' '_')
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/spmv_pmc_$name -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-cfg5 --spgemm-scale 0 > $OUT/spmv_pmc_$name.json 2> $OUT/spmv_pmc_$name.log || echo "spmv pmc $grp failed" >> $OUT/errors.log
  echo "spmv pmc $grp done" >> $OUT/progress.txt
done
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  SPAM_DBG_HEAVY=1 timeout -k 5 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/spgemm_pmc_$name -- python3 tools/spgemm_probe.py 20 1 > $OUT/spgemm_pmc_$name.log 2>&1 || echo "spgemm pmc $grp failed" >> $OUT/errors.log
  echo "spgemm pmc $grp done" >> $OUT/progress.txt
done
python3 tools/pmc_summary.py $OUT
if [ -s $OUT/errors.log ]; then cat $OUT/errors.log; exit 1; fi
exit 0
