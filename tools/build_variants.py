"""Builds tuning variants of liblinevis_hip.so into linevis_amd/_lib/variants/<name>.so (dev tool).
usage: python tools/build_variants.py name1:-DFOO=1,-DBAR=2 name2:..."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from linevis_amd import build as B
out = os.path.join(B.OUT_DIR, "variants")
os.makedirs(out, exist_ok=True)
procs = []
for spec in sys.argv[1:]:
    name, _, flags = spec.partition(":")
    flags = [f for f in flags.split(",") if f]
    lib = os.path.join(out, name + ".so")
    srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES]
    cmd = [B.hipcc()] + B.FLAGS + flags + ["-shared"] + srcs + ["-o", lib]
    procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
for name, p in procs:
    o, _ = p.communicate()
    print(name, "ok" if p.returncode == 0 else "FAILED\n" + o)
