#!/usr/bin/env python3
"""Builds tools/bvhlab/bvhlab and runs it on one of the bench scenes (CPU only).  usage: run.py [tornado|helix|rb] [stride] [spp] [builders]"""
import os, struct, subprocess, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from linevis_amd import scenes
from oracle import lvo

here = os.path.dirname(os.path.abspath(__file__))
out = os.environ.get("LAB_OUT", "/tmp/bvhlab_build")   # scene dumps are hundreds of MB: never inside the repository snapshot
os.makedirs(out, exist_ok=True)
exe = os.path.join(out, "bvhlab")
subprocess.check_call(["g++", "-O3", "-march=native", "-fopenmp", "-std=c++17", os.path.join(here, "bvhlab.cpp"), "-o", exe])
name = sys.argv[1] if len(sys.argv) > 1 else "tornado"
tri = name.endswith("_tri")      # tornado_tri: the 6-gon triangle tubes of the same lines (the reference's RTAO geometry)
name = name[:-4] if tri else name
gen = {"tornado": scenes.tornado, "helix": scenes.helix_bundle, "rb": scenes.rayleigh_benard}[name]
tr = scenes.normalize(gen())
if tri:
    idx, verts, _ = lvo.build_tube_triangle_render_data(tr.positions, tr.attributes, tr.line_offsets, 0.002, 6)
    v = verts["vertexPosition"]
    data = np.concatenate([v[idx[:, 0]], v[idx[:, 1]], v[idx[:, 2]]], axis=1).astype(np.float32)
else:
    pts, seg, _ = lvo.build_tube_aabb_render_data(tr.positions, tr.attributes, tr.line_offsets, 0.002)
    p = pts["linePosition"]
    data = np.concatenate([p[seg[:, 0]], p[seg[:, 1]]], axis=1).astype(np.float32)
path = os.path.join(out, name + ("_tri" if tri else "") + ".bin")
with open(path, "wb") as f:
    f.write(struct.pack("<If", len(data) | (0x80000000 if tri else 0), 0.001))
    f.write(np.ascontiguousarray(data).tobytes())
subprocess.check_call([exe, path] + sys.argv[2:])
