#!/bin/bash
# Everything profiles/ needs for one round, on the GPU box:  bash tools/profile_round.sh <tag>
#   bench lines of every workload          -> gpurun_out/prof_<tag>/bench_<workload>.json
#   rocprofv3 --kernel-trace --stats       -> gpurun_out/prof_<tag>/<workload>_kernel_stats.csv  (same command as the bench line)
#   PMC counter groups (tools/pmc_collect.sh: full set for c3c / c3t / c4, roofline set for the others) -> gpurun_out/pmc_<tag>/
#   shard / frames-in-flight probe          -> gpurun_out/shard_probe_<workload>.json
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# issue-rate / gather-rate calibration of this box (tools/ubench: built in the CPU container, shipped with the snapshot)
mkdir -p $R/gpurun_out/ubench_$TAG
# (the binaries are build artefacts: untracked, gone with a re-created container -- build them here if the snapshot has none)
mkdir -p $R/tools/ubench/_build
for u in valu_rates tcp_rates; do [ -x $R/tools/ubench/_build/$u ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/tools/ubench/$u.hip -o $R/tools/ubench/_build/$u 2>> $OUT/ubench_build.err; done
timeout 600 $R/tools/ubench/_build/valu_rates > $R/gpurun_out/ubench_$TAG/valu_rates.json 2> $OUT/valu_rates.err
timeout 600 $R/tools/ubench/_build/tcp_rates > $R/gpurun_out/ubench_$TAG/tcp_rates.json 2> $OUT/tcp_rates.err
python $R/tools/summarize_ubench.py ubench_$TAG $TAG > $OUT/ubench_summary.json 2>> $OUT/valu_rates.err
bash $R/tools/pmc_collect.sh $TAG c3c c3t c4 > $OUT/pmc_full.log 2>&1
LV_PMC_LITE=1 bash $R/tools/pmc_collect.sh $TAG c2 c2e c4c c4m c4l c5 c5c > $OUT/pmc_lite.log 2>&1
for f in $R/gpurun_out/pmc_$TAG/*.json; do cp $f $R/profiles/pmc_${TAG}_$(basename $f); done   # bench.py reads profiles/
cp $R/profiles/pmc_${TAG}_c5.json $R/profiles/pmc_${TAG}_c5t.json   # workload c5 = c5t
python $R/bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err
# the CPU baseline beside C2, C3 and C4 (a bounded ~15-s sample each); the variants of those configs without it
for w in c2 c4; do
  python $R/bench.py --workload $w --steps 100 --warmup 5 > $OUT/bench_$w.json 2> $OUT/bench_$w.err
done
for w in c3c c3t c2e c4c c4m c4l c5 c5c; do
  python $R/bench.py --workload $w --steps 100 --warmup 5 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err
done
for w in c3 c4 c4m c2; do
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
python $R/bench.py --workload c4 --steps 100 --warmup 5 --no-cpu-baseline --set ppll_prism_rasteriser=lbvh > $OUT/bench_c4_lbvh.json 2> $OUT/bench_c4_lbvh.err
# the SVGF passes next to the RTAO kernels they follow
python $R/bench.py --workload c3c --steps 100 --warmup 5 --no-cpu-baseline --set ambient_occlusion_denoiser=SVGF > $OUT/bench_c3c_svgf.json 2> $OUT/bench_c3c_svgf.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_svgf -o bench -- python $R/bench.py --workload c3c --steps 20 --warmup 3 --no-cpu-baseline --set ambient_occlusion_denoiser=SVGF > $OUT/rocprof_svgf.json 2> $OUT/rocprof_svgf.err
f=$(find $OUT/trace_svgf -name "bench_kernel_stats.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
with open("$OUT/c3c_svgf_kernel_stats.csv", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -- python bench.py --workload c3c --steps 20 --warmup 3 --no-cpu-baseline --set ambient_occlusion_denoiser=SVGF (MI355X)\n")
    o.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent\n")
    for r in rows[:30]:
        o.write("%s,%s,%s,%.0f,%s,%s,%s\n" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"]))
PY
rm -rf $OUT/trace_svgf
cd $R
LV_PROBE_DEPTHS=1,2,4 python tools/probe_shard.py c3c > $OUT/shard_c3c.txt 2>&1
LV_PROBE_DEPTHS=1,2,4 python tools/probe_shard.py c3t > $OUT/shard_c3t.txt 2>&1
python tools/probe_shard_ppll.py > $OUT/shard_c4.txt 2>&1
