#!/bin/bash
# rocprofv3 evidence for bench.py: kernel trace + stats of the default run, then PMC passes (one counter
# group per run, kernel-trace only) on the SpMV leg.  Summaries are written to gpurun_out/prof_bench/.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_bench
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum"; do
  name=$(echo $grp | tr ' ' '_')
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-cfg5 --spgemm-scale 0 > /dev/null 2> $OUT/pmc_$name.log || echo "pmc $grp failed" >> $OUT/errors.log
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
with open(out + "/summary.txt", "w") as fh:
    for f in sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)):
        fh.write("== rocprofv3 --kernel-trace --stats: python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline\n")
        fh.write("%-100s %6s %12s %12s %7s\n" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
        for r in csv.DictReader(open(f)):
            if "spam::" in r["Name"]:
                fh.write("%-100s %6s %12.3f %12.2f %7s\n" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
    fh.write("\n== PMC (per dispatch averages; bench.py --steps 5 --warmup 1 --spgemm-scale 0)\n")
    for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"][:70], row["Counter_Name"])
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
        for (k, c), (v, n) in sorted(agg.items()):
            if "spam::spmv" in k or "spam::sliced" in k or "xh_build" in k:
                fh.write("%-70s %-24s %14.6g (n=%d)\n" % (k, c, v / n, n))
print(open(out + "/summary.txt").read())
PY
