#!/usr/bin/env python3
"""Kernel tuning variants of liblinevis_hip.so (dev tool).

  python tools/variants.py build name1:-DFOO=1,-DBAR=2 name2:...     (here: cross-compiles lv_render.hip + lv_mlat.hip with
                                                                       the extra flags, links with the other objects of the
                                                                       regular build -> linevis_amd/_lib/variants/<name>.so)
  python tools/variants.py run [--workload c3] [--steps 30] name1 name2 ...   (on the GPU box: bench.py per variant through
                                                                       LV_LIB_PATH, prints ms/frame and per-kernel ms)
"""
import json
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from linevis_amd import build as B  # noqa: E402

OUT = os.path.join(B.OUT_DIR, "variants")
RECOMPILE = ["lv_render.hip", "lv_mlat.hip"]   # the translation units that instantiate lv_trace.h


def build(specs):
    B.build()
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for spec in specs:
        name, _, flags = spec.partition(":")
        flags = [f for f in flags.split(",") if f]
        if any(f.startswith("-DLV_EXP_EXTRA_") for f in flags):   # the node-step sensitivity probes live outside the product headers
            flags += ["-include", os.path.join(R, "tools", "variants_inc", "probes.h")]
        objs = []
        for s, oname, extra in B.SOURCES:
            if s in RECOMPILE:
                obj = os.path.join(OUT, "%s.%s" % (name, oname))
                procs.append((name, s, subprocess.Popen([B.hipcc()] + B.FLAGS + extra + flags + ["-c", os.path.join(B.CSRC, s), "-o", obj],
                                                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
            else:
                obj = os.path.join(B.OUT_DIR, oname)
            objs.append(obj)
        procs.append((name, None, objs))
    pending = {}
    for name, s, p in procs:
        if s is not None:
            o, _ = p.communicate()
            if p.returncode != 0:
                print(name, s, "FAILED\n" + o)
                pending[name] = False
        else:
            if pending.get(name, True):
                subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + p + ["-o", os.path.join(OUT, name + ".so")])
                print(name, "ok")
    for f in os.listdir(OUT):
        if f.endswith(".o"):
            os.remove(os.path.join(OUT, f))


def run(args):
    workload, steps, names = "c3c", "30", []
    it = iter(args)
    for a in it:
        if a == "--workload":
            workload = next(it)
        elif a == "--steps":
            steps = next(it)
        else:
            names.append(a)
    for v in names:
        env = dict(os.environ)
        if v != "base":
            env["LV_LIB_PATH"] = os.path.join(OUT, v + ".so")
        r = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--workload", workload, "--steps", steps, "--warmup", "3",
                            "--no-cpu-baseline"], env=env, capture_output=True, text=True)
        try:
            j = json.loads(r.stdout.strip().splitlines()[-1])
            print("%-14s %8.4f ms/frame  %s  nodes %d prims %d" % (v, j["ms_per_step"], j["kernels_ms"], j["counters_rank0"]["ao_nodes_visited"],
                                                            j["counters_rank0"]["ao_prims_tested"]), flush=True)
        except Exception:
            print(v, "FAILED", r.stdout[-500:], r.stderr[-1500:], flush=True)


if __name__ == "__main__":
    if len(sys.argv) < 3 or sys.argv[1] not in ("build", "run"):
        raise SystemExit(__doc__)
    (build if sys.argv[1] == "build" else run)(sys.argv[2:])
