#!/bin/bash
# sweep one option on a workload (GPU box):  bash tools/sweep_opt.sh <workload> <key> <v1> <v2> ...
W=$1; K=$2; shift 2
for v in "$@"; do
  python bench.py --workload $W --steps 60 --warmup 5 --no-cpu-baseline --set $K=$v 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = d['config']
print('$K=$v', d['ms_per_step'], {k: v['median'] for k, v in d['kernels_ms'].items()}, 'build', c.get('accel_build_ms'), c.get('tri_accel_build_ms'), 'nodes', c.get('tri_accel_nodes'), 'ao_nodes', d['counters_rank0'].get('ao_nodes_visited'), 'ao_prims', d['counters_rank0'].get('ao_prims_tested'))"
done
