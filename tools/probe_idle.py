#!/usr/bin/env python3
"""Where the idle lanes of k_ao_rays' descend loop come from (EXPERIMENTS.md 3.1, round 3).

  python tools/probe_idle.py build      (CPU container: compiles the instrumented variant, -DLV_AO_IDLE_PROBE=1)
  python tools/probe_idle.py            (GPU box: per RTAO geometry the descend loop's lane utilisation, the lane-slots of rays
                                         that have finished their traversal but wait for their last leaf tests, and the
                                         lane-slots without a ray)

The variant reuses two counters of the instrumented kernel (ao_prim_may_axis / ao_prim_may_both) for the two idle classes."""
import json
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT = os.path.join(R, "linevis_amd", "_lib", "variants", "idleprobe.so")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        raise SystemExit(subprocess.call([sys.executable, os.path.join(R, "tools", "variants.py"), "build",
                                          "idleprobe:-DLV_AO_IDLE_PROBE=1"]))
    if not os.path.exists(VARIANT):
        raise SystemExit("build the variant first: python tools/probe_idle.py build")
    env = dict(os.environ, LV_LIB_PATH=VARIANT)
    for w in ("c3c", "c3t"):
        r = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--workload", w, "--steps", "5", "--warmup", "1",
                            "--no-cpu-baseline"], env=env, capture_output=True, text=True)
        j = json.loads(r.stdout.strip().splitlines()[-1])
        c = j["counters_rank0"]
        print(w, "descend lane utilisation", c["ao_phase_lane_utilisation"], "finished-ray lane-slots", c["ao_prim_may_axis"],
              "no-ray lane-slots", c["ao_prim_may_both"], "node visits", c["ao_nodes_visited"])


if __name__ == "__main__":
    main()
