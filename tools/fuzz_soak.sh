#!/bin/bash
# Exploratory fuzz on the GPU box: bash tools/fuzz_soak.sh <first offset> <last offset> [log]   (seeds shifted by LV_FUZZ_SEED_OFFSET;
# a failing offset reproduces with the same variable).  The log goes to profiles/fuzz_<round>.log.
A=$1; B=$2; LOG=${3:-gpurun_out/fuzz.log}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for o in $(seq $A $B); do
  r=$(LV_FUZZ_SEED_OFFSET=$o timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -1)
  echo "offset $o: $r" >> $LOG
  case "$r" in *failed*|*error*) LV_FUZZ_SEED_OFFSET=$o python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -40 >> $LOG.fail;; esac
done
