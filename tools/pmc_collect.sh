#!/bin/bash
# PMC counters of the frame kernels, one rocprofv3 pass per counter group (separate --pmc passes with --kernel-trace only:
# MI355X_MICROARCH.md "rocprofv3 PMC slots").  Usage (on the GPU box): bash tools/pmc_collect.sh <tag> <workload> [<workload> ...]
# (TA_* / TD_* groups are not collected: those passes hang on this pool until the timeout.)
# Result: gpurun_out/pmc_<tag>/<workload>.json = {kernel: {counter: mean per launch of the non-instrumented kernel}}.
TAG=$1; shift
LITE=${LV_PMC_LITE:-0}   # 1: only the groups the roofline of bench.py needs (5 passes instead of 12)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
 "SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32"
 "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_INSTS"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum"
 "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN2_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
 "TCC_BUSY_avr TCC_TAG_STALL_sum TCC_WRITE_sum TCC_WRITEBACK_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
if [ "$LITE" = "1" ]; then
PASSES=(
 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
fi
for W in "$@"; do
  i=0
  for C in "${PASSES[@]}"; do
    i=$((i+1))
    rm -rf $OUT/$W.p$i
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$W.p$i -o p -- \
        python $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$W.p$i.out 2> $OUT/$W.p$i.err
    echo "pass $i ($C): rc $?" >> $OUT/$W.log
  done
  python - <<PY
import csv, collections, glob, json, os
agg = collections.defaultdict(lambda: collections.defaultdict(list))
passes = []
for f in sorted(glob.glob("$OUT/$W.p*/**/p_counter_collection.csv", recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        if not n.startswith("k_"):
            continue   # foreign kernels (torch, rocPRIM)
        if n.startswith(("k_render_rt", "k_ao_primary", "k_ao_rays", "k_ppll_gather", "k_ppll_raster_prism", "k_ppll_shade_prism", "k_ppll_mark_tiles")) and "<true" in n.split(",")[0]:
            continue   # the instrumented (collect_stats) instances
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        seen.add(r["Counter_Name"])
    passes.append(sorted(seen))
import sys
sys.path.insert(0, "$R")
from linevis_amd import build as _lvb
out = {"workload": "$W", "git_head": os.environ.get("LV_GIT_HEAD", "unknown"), "source_sha": _lvb.source_sha(), "collected_with": "rocprofv3 --pmc <group> --kernel-trace -- python bench.py --workload $W --steps 2 "
       "--warmup 1 --no-cpu-baseline; one pass per group; mean per launch", "passes": passes,
       "kernels": {n: {c: sum(v) / len(v) for c, v in cs.items()} for n, cs in agg.items()},
       "launches": {n: max(len(v) for v in cs.values()) for n, cs in agg.items()}}
json.dump(out, open("$OUT/$W.json", "w"), indent=1, sort_keys=True)
print("$W", {n: len(cs) for n, cs in out["kernels"].items()})
PY
  # the raw per-dispatch CSVs are large: keep the merged JSON + logs only
  for d in $OUT/$W.p*/; do rm -rf "$d"; done
done
