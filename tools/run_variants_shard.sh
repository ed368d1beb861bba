#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in "$@"; do
  echo "== $v"
  LV_LIB_PATH=$R/linevis_amd/_lib/variants/$v.so python $R/tools/probe_shard.py 2>/dev/null | grep world
done
