#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in "$@"; do
  echo "== $v"
  LV_LIB_PATH=$R/linevis_amd/_lib/variants/$v.so python $R/bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'],'Mrays/s', r['ms_per_step'],'ms', r['kernels_ms'])"
done
