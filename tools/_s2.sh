mkdir -p gpurun_out/s2; cd $GRAFT_REPO_ROOT
( nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())"; lscpu | grep -E "Model name|Socket|Core|Thread|^CPU\(s\)"; cat /proc/loadavg ) > gpurun_out/s2/cpu.txt 2>&1
./tools/ubench/_build/valu_rates > gpurun_out/s2/valu_rates.json 2> gpurun_out/s2/valu.err
./tools/ubench/_build/tcp_rates > gpurun_out/s2/tcp_rates.json 2> gpurun_out/s2/tcp.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s2/tcp_pmc -o p -- $GRAFT_REPO_ROOT/tools/ubench/_build/tcp_rates > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/s2/tcp_pmc.err
cd $GRAFT_REPO_ROOT
python - <<'PY' > gpurun_out/s2/tcp_pmc_summary.txt 2>&1
import csv, glob, collections
f = glob.glob("gpurun_out/s2/tcp_pmc/**/p_counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    agg.setdefault((r["Dispatch_Id"], r["Kernel_Name"][:60]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in agg.items():
    print(k, v)
PY
rm -rf gpurun_out/s2/tcp_pmc
python tools/variants.py run --steps 30 base xv16 xv32 xl2 xl4 pk > gpurun_out/s2/variants.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s2/tests.log 2>&1
tail -5 gpurun_out/s2/tests.log; cat gpurun_out/s2/variants.txt; cat gpurun_out/s2/cpu.txt
