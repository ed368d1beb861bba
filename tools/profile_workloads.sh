#!/bin/bash
# rocprofv3 --kernel-trace --stats for the secondary workloads; summaries -> gpurun_out/prof_wl/<workload>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_wl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w -o bench -- python $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $OUT/$w.json 2> $OUT/$w.err
  f=$(find $OUT/$w -name "bench_kernel_stats.csv" | head -1)
  python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
with open("$OUT/${w}_kernel_stats.csv", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -- python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline (MI355X)\n")
    o.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent\n")
    for r in rows[:25]:
        o.write("%s,%s,%s,%.0f,%s,%s,%s\n" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"]))
print("$w", [(short(r["Name"]), r["Calls"], round(float(r["AverageNs"]) / 1e6, 3)) for r in rows[:4]])
PY
done
