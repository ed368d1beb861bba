mkdir -p gpurun_out/s5; cd $GRAFT_REPO_ROOT
python tools/probe_pipeline.py c3c > gpurun_out/s5/pipe_c3c.txt 2>&1
python tools/probe_pipeline.py c3t > gpurun_out/s5/pipe_c3t.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "adapter or tile_list or jittered" > gpurun_out/s5/tests.log 2>&1
cat gpurun_out/s5/pipe_c3c.txt gpurun_out/s5/pipe_c3t.txt; tail -3 gpurun_out/s5/tests.log
