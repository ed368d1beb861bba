cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_tmp
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c4 -o bench -- python $R/bench.py --workload c4 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/rocprof_c4.json 2> $OUT/rocprof_c4.err
f=$(find $OUT/trace_c4 -name "bench_kernel_stats.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
for r in rows[:30]:
    print("%s,%s,%s,%.0f,%s,%s,%s" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"]))
PY
rm -rf $OUT/trace_c4
