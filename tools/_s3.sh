mkdir -p gpurun_out/s3; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s3/tests.log 2>&1
tail -5 gpurun_out/s3/tests.log
python tools/variants.py run --steps 50 base q1024 q64 > gpurun_out/s3/variants_c3.txt 2>&1
python tools/variants.py run --workload c3t --steps 50 base q1024 > gpurun_out/s3/variants_c3t.txt 2>&1
cat gpurun_out/s3/variants_c3.txt gpurun_out/s3/variants_c3t.txt
