"""dev probe: PPLL gather time of the heaviest 120x120 tile of config 4 and of the full frame."""
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
for tile in [(720, 240, 120, 120), (720, 240, 16, 16), (776, 296, 16, 16), (760, 280, 64, 64), None]:
    ts = []
    for _ in range(4):
        ctx.render(2, tile=tile); ts.append(ctx.stats().ms_ppll_gather)
    print(os.environ.get("LV_LIB_PATH", "default").split("/")[-1], tile, "gather ms", min(ts), "frags", ctx.stats().fragments,
          "maxdepth", ctx.stats().max_depth_complexity)
ctx.set_option("collect_stats", True)
for tile in [(720, 240, 120, 120), (720, 240, 16, 16)]:
    ctx.render(2, tile=tile); s = ctx.stats()
    n = tile[2] * tile[3]
    print("stats", tile, "nodes/ray", s.nodes_visited / n, "prims/ray", s.prims_tested / n, "frags/ray", s.fragments / n, "max nodes/pixel", s.max_nodes_per_pixel)
