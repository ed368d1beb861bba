#!/bin/bash
# After `gpurun ... bash tools/profile_round.sh <tag>`: copy what the round produced under gpurun_out/ into profiles/ (tracked).
TAG=${1:-r06}
cd "$(dirname "$0")/.."
for w in c2 c2e c3 c3c c3c_svgf c3t c4 c4c c4_lbvh c4l c4m c5 c5c; do tail -1 gpurun_out/prof_$TAG/bench_$w.json > profiles/bench_${TAG}_$w.json; done
for w in c2 c3 c4 c4m c3c_svgf; do cp gpurun_out/prof_$TAG/${w}_kernel_stats.csv profiles/${TAG}_${w}_kernel_stats.csv; done
for f in gpurun_out/pmc_$TAG/*.json; do cp $f profiles/pmc_${TAG}_$(basename $f); done
if [ -s gpurun_out/prof_$TAG/ubench_summary.json ]; then cp gpurun_out/prof_$TAG/ubench_summary.json profiles/ubench_$TAG.json; else echo "ubench summary empty: profiles/ubench_$TAG.json kept" >&2; fi
for w in c3c c3t c4; do [ -s gpurun_out/shard_probe_$w.json ] && cp gpurun_out/shard_probe_$w.json profiles/shard_probe_${TAG}_$w.json; done
[ -s profiles/pmc_${TAG}_c5.json ] && cp profiles/pmc_${TAG}_c5.json profiles/pmc_${TAG}_c5t.json   # workload c5 = c5t (bench.py reads the c5t name)
