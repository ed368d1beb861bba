"""Builds the C-ABI shared library (HIP, gfx950 only) in-tree: linevis_amd/_lib/liblinevis_hip.so.

hipcc cross-compiles without a GPU.  -ffp-contract=off keeps float32 evaluation order fixed on host and device
(DESIGN.md "Numerics"); no CUDA / multi-arch paths exist.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_lib")
LIB = os.path.join(OUT_DIR, "liblinevis_hip.so")
# (source, object, extra flags): lv_mlat.hip is compiled in four parts (its kernel instantiations dominate the build time)
SOURCES = [("lv_api.hip", "lv_api.o", []), ("lv_bvh.hip", "lv_bvh.o", []), ("lv_render.hip", "lv_render.o", []),
           ("lv_flow.hip", "lv_flow.o", []), ("lv_lines.hip", "lv_lines.o", []), ("lv_svgf.hip", "lv_svgf.o", []), ("lv_multi.hip", "lv_multi.o", [])] + \
          [("lv_mlat.hip", "lv_mlat_%d.o" % p, ["-DLV_MLAT_PART=%d" % p]) for p in range(4)]
HEADERS = ["lv_device.h", "lv_trace.h", "lv_prism.h", "lv_tile.h", "lv_internal.h", os.path.join("..", "..", "include", "linevis_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-DNDEBUG"] + os.environ.get("LV_EXTRA_HIPCC_FLAGS", "").split()


def source_sha():
    """Content hash of everything the HIP library is built from (kernels, headers, C-ABI header, this file's flags): profiles record
    it at collection time and bench.py compares it with the tree it runs on, so that a roofline is never stitched from two builds."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files += [os.path.join(HERE, "..", "include", "linevis_hip.h"), os.path.abspath(__file__)]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs = []
    procs = []
    for s, oname, extra in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OUT_DIR, oname)
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc()] + FLAGS + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((oname, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s:\n%s\n" % (s, out))
        elif verbose and out.strip():
            sys.stderr.write(out)
    if failed:
        raise RuntimeError("building liblinevis_hip.so failed")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", LIB]
        subprocess.check_call(cmd)
    return LIB


HOST_DIR = os.path.join(HERE, "host")
HOST_LIB = os.path.join(OUT_DIR, "liblinevis_host.so")
HOST_SOURCES = ["LineData.cpp", "Tubes.cpp", "Flow.cpp", "LineRenderer.cpp", "HeadlessLineRenderer.cpp", "host_capi.cpp"]
HOST_HEADERS = ["LvMath.hpp", "SettingsMap.hpp", "InternalState.hpp", "LineData.hpp", "Tubes.hpp", "Flow.hpp", "LineRenderer.hpp", "HeadlessLineRenderer.hpp"]
# -march=x86-64-v3: built in the CPU container, shipped to the GPU box; -ffp-contract=off: fixed float32 order
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-fno-fast-math",
              "-march=x86-64-v3", "-Wall", "-Wextra", "-Wno-unused-parameter"]


def build_host(force=False, verbose=False):
    """C++ host layer (LineData / LineRenderer-shaped classes) over the C-ABI: liblinevis_host.so."""
    build(force=force, verbose=verbose)
    srcs = [os.path.join(HOST_DIR, s) for s in HOST_SOURCES]
    deps = srcs + [os.path.join(HOST_DIR, h) for h in HOST_HEADERS] + [LIB, os.path.abspath(__file__)]
    if force or _stale(HOST_LIB, deps):
        cmd = [os.environ.get("CXX", "g++")] + HOST_FLAGS + srcs + ["-L" + OUT_DIR, "-llinevis_hip",
                                                                    "-Wl,-rpath,$ORIGIN", "-o", HOST_LIB]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return HOST_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
