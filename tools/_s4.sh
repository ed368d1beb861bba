mkdir -p gpurun_out/s4; cd $GRAFT_REPO_ROOT
python tools/probe_shard.py c3c > gpurun_out/s4/shard_c3c.txt 2>&1
python tools/probe_shard.py c3t > gpurun_out/s4/shard_c3t.txt 2>&1
LV_PMC_LITE=1 bash tools/pmc_collect.sh r02 c3c c3t > gpurun_out/s4/pmc.log 2>&1
python bench.py > gpurun_out/s4/bench.json 2> gpurun_out/s4/bench.err
tail -3 gpurun_out/s4/shard_c3c.txt; tail -3 gpurun_out/s4/shard_c3t.txt; head -c 1500 gpurun_out/s4/bench.json; tail -3 gpurun_out/s4/bench.err
