mkdir -p gpurun_out/s6; cd $GRAFT_REPO_ROOT
python tools/probe_shard.py c3c > gpurun_out/s6/shard_c3c.txt 2>&1
python tools/probe_shard.py c3t > gpurun_out/s6/shard_c3t.txt 2>&1
grep "^8 \|^4 \|^2 " gpurun_out/s6/shard_c3c.txt | cut -c1-60,100-400; grep "^8 " gpurun_out/s6/shard_c3t.txt| cut -c1-400
