#!/bin/bash
# PMC passes over bench.py (3 steps) for kernel-level diagnosis. Usage: bash tools/pmc_probe.sh <tag> "<counters pass1>" "<pass2>" ...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $R/bench.py --workload ${LV_WORKLOAD:-c3} --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/p$i.err
  python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/p$i/p_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"].replace("void ","").replace("(anonymous namespace)::","").split("(")[0]
    if n.startswith(("k_ao_rays<false", "k_render_rt<false", "k_ao_primary<false", "k_ppll_gather<false", "k_ppll_resolve")):
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n in agg:
    print(n, {c: round(sum(v)/len(v),1) for c, v in agg[n].items()})
PY
done
