"""dev probe: PPLL gather statistics of config 4 (max nodes per pixel, depth complexity)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from linevis_amd import camera, capi, host_api, scenes, transfer_function as tfm
W, H = 1920, 1080
tr = scenes.normalize(scenes.tornado())
flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
pts, seg, _ = flow.tube_aabb_render_data(0.002)
view, proj, fovy, near, far = camera.default_camera(W, H)
ctx = capi.Context(0)
ctx.set_lines(pts, seg); ctx.set_transfer_function(tfm.standard_transparent(), *flow.attribute_range())
ctx.set_camera(view, proj, fovy, near, far, W, H); ctx.set_option("line_width", 0.002)
ctx.set_options({"ppll_max_num_frags": 64, "ppll_expected_avg_depth_complexity": 20})
ctx.set_option("collect_stats", True)
ctx.render(2)
s = ctx.stats()
print("max_nodes_per_pixel", s.max_nodes_per_pixel, "max_depth_complexity", s.max_depth_complexity, "frags", s.fragments,
      "nodes", s.nodes_visited, "prims", s.prims_tested, "gather ms", s.ms_ppll_gather)
ctx.set_option("collect_stats", False)
for _ in range(3):
    ctx.render(2)
s = ctx.stats(); print("gather ms", s.ms_ppll_gather, "resolve", s.ms_ppll_resolve)
# per-tile timing: which 64x64 tiles are slow?
import time
ts = []
for ty in range(0, H, 120):
    for tx in range(0, W, 120):
        ctx.render(2, tile=(tx, ty, 120, 120)); ts.append((ctx.stats().ms_ppll_gather, tx, ty))
ts.sort(reverse=True)
print("slowest 120x120 tiles:", ts[:8], "sum", sum(t[0] for t in ts))
