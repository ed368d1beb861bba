#!/usr/bin/env python3
"""Build time of the capsule LBVH of config 3 (1 M segments) for several values of treelet_lane_leaves (0 = the wave splits every range
of a treelet); LV_GROUP=8|16 selects the group builder (treelet_group_leaves), LV_PLANE_EVAL=scan|loop the plane evaluation.
usage (GPU box): [LV_GROUP=8] python tools/probe_build.py [values ...]"""
import json
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from linevis_amd import camera, capi, host_api, scenes, transfer_function as tfm  # noqa: E402

EVAL = os.environ.get("LV_PLANE_EVAL", "scan")
GROUP = int(os.environ.get("LV_GROUP", "0"))
TOP = os.environ.get("LV_COLLAPSE_TOP", "1") != "0"
vals = [int(v) for v in sys.argv[1:]] or [0, 4, 8, 16, 32, 64]
tr = scenes.normalize(scenes.tornado())
flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
pts, seg, _ = flow.tube_aabb_render_data(0.002)
view, proj, fovy, near, far = camera.default_camera(1920, 1080)
out = {}
for ll in vals:
    ctx = capi.Context(0)
    ctx.set_lines(pts, seg)
    ctx.set_transfer_function(tfm.standard(), *flow.attribute_range())
    ctx.set_camera(view, proj, fovy, near, far, 1920, 1080)
    ctx.set_option("line_width", 0.002)
    ctx.set_option("treelet_lane_leaves", ll)
    ctx.set_option("treelet_plane_eval", EVAL)
    ctx.set_option("treelet_group_leaves", GROUP)
    ctx.set_option("accel_collapse_top", TOP)
    ctx.build_accel()
    t = []
    for _ in range(5):
        ctx.build_accel()
        t.append(float(ctx.stats().ms_accel_build))
    row = {"capsules_ms": round(float(np.median(t)), 3)}
    out["%d/g%d/%s/top%d" % (ll, GROUP, EVAL, TOP)] = row
    print(ll, row, flush=True)
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(R, "gpurun_out", "probe_build.json"), "w"), indent=1)
