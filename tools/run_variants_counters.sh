#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
WLS=$1; shift
for v in "$@"; do
  for w in $WLS; do
    echo "== $v $w"
    LV_LIB_PATH=$R/linevis_amd/_lib/variants/$v.so python $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'],'ms', r['kernels_ms'], r['counters_rank0'])"
  done
done
