import json,subprocess,sys,os
env=dict(os.environ); env["LV_LIB_PATH"]="/root/repo/linevis_amd/_lib/variants/idleprobe.so"
for w in ("c3c","c3t"):
    r=subprocess.run([sys.executable,"bench.py","--workload",w,"--steps","5","--warmup","1","--no-cpu-baseline"],env=env,capture_output=True,text=True)
    j=json.loads(r.stdout.strip().splitlines()[-1]); c=j["counters_rank0"]
    print(w, "descend lane util", c["ao_phase_lane_utilisation"], "wait-for-tests lanes", c["ao_prim_may_axis"], "no-ray lanes", c["ao_prim_may_both"], "node visits", c["ao_nodes_visited"])
