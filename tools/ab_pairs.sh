#!/bin/bash
# A/B of triangle_leaf_records on the headline workload (GPU box):  bash tools/ab_pairs.sh [workload]
W=${1:-c3}
mkdir -p gpurun_out/ab_pairs
for r in pairs triangles pairs triangles; do
  python bench.py --workload $W --steps 100 --warmup 5 --no-cpu-baseline --set triangle_leaf_records=$r > gpurun_out/ab_pairs/bench_${W}_$r.json 2> gpurun_out/ab_pairs/err_$r.txt
  python - <<EOF
import json
d = json.loads(open("gpurun_out/ab_pairs/bench_${W}_$r.json").read().strip().splitlines()[-1])
print("$r", d["value"], d["ms_per_step"], d["roofline"].get("ms_per_launch"), {k: v["median"] for k, v in d["kernels_ms"].items()})
EOF
done
