#!/bin/bash
# Runs bench.py plain, under rocprofv3 --kernel-trace --stats, and under separate PMC passes (FETCH_SIZE / WRITE_SIZE).
# Usage (through gpurun): bash tools/profile_bench.sh <tag>
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 3 --save-frame $OUT/frame.png > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
find $OUT -name "*.csv" | head -30
ls -la $OUT $OUT/trace 2>/dev/null | head -40
