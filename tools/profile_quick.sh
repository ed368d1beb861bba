#!/bin/bash
# The minimum a round's profiles/ needs (GPU box, ~6 min): bench lines of c3 (with the CPU baseline), c4, c5 and the rocprofv3 kernel
# statistics of the headline command.  bash tools/profile_quick.sh <tag>   (tools/profile_round.sh collects everything)
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/quick_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python $R/bench.py --workload c4 --steps 100 --warmup 5 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python $R/bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err
for w in c3 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -o bench -- python $R/bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $OUT/rocprof_$w.json 2> $OUT/rocprof_$w.err
  f=$(find $OUT/trace_$w -name "bench_kernel_stats.csv" | head -1)
  python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
with open("$OUT/${w}_kernel_stats.csv", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -- python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline (MI355X)\n")
    o.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent\n")
    for r in rows[:30]:
        o.write("%s,%s,%s,%.0f,%s,%s,%s\n" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"]))
PY
  rm -rf $OUT/trace_$w
done
