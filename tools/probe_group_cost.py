#!/usr/bin/env python3
"""Per-group cost of the tile kernels (lv_get_dispatch_order) on a bench workload: how much of a launch is its heaviest group.
  python tools/probe_group_cost.py c4 [c2 c3c ...]     (GPU box)"""
import json
import os
import subprocess
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)


def main():
    import torch  # noqa: F401
    import bench
    from linevis_amd import camera, capi, host_api, scenes, transfer_function as tfm
    for wk in sys.argv[1:]:
        w = bench.WORKLOADS[wk]
        W, H = w.get("resolution", (1920, 1080))
        view, proj, fovy, near, far = camera.default_camera(W, H)
        gen = {"tornado": scenes.tornado, "helix": scenes.helix_bundle, "rayleigh_benard": scenes.rayleigh_benard}[w["scene"]]
        tr = scenes.normalize(gen())
        flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
        pts, seg, _ = flow.tube_aabb_render_data(bench.LINE_WIDTH)
        ctx = capi.Context(0)
        ctx.set_lines(pts, seg)
        ctx.set_transfer_function(tfm.standard_transparent() if w.get("transparent") else tfm.standard(), *flow.attribute_range())
        ctx.set_camera(view, proj, fovy, near, far, W, H)
        ctx.set_option("line_width", bench.LINE_WIDTH)
        if w.get("mesh"):
            ctx.set_tube_triangle_mesh(*flow.tube_triangle_render_data(bench.LINE_WIDTH, 6))
        ctx.set_options(w["settings"])
        out = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
        for _ in range(5):
            ctx.render_device(out.data_ptr(), mode=w["mode"])
        torch.cuda.synchronize()
        order, cost = ctx.dispatch_order()
        st = ctx.stats()
        c = np.sort(cost.astype(np.float64))[::-1] * 0.01 / 64.0      # us per wave, mean over the group's 64 waves
        print(json.dumps({"workload": wk, "groups": int(len(c)), "ms_total": round(st.ms_total, 4), "ms_color": round(st.ms_color, 4),
                          "ms_ppll_gather": round(st.ms_ppll_gather, 4), "ms_ao": round(st.ms_ao, 4),
                          "mean_wave_us_top10": [round(x, 1) for x in c[:10]], "median": round(float(np.median(c)), 1),
                          "sum_wave_ms_over_all_groups": round(float(c.sum() * 64 / 1e3), 2),
                          "ideal_ms_at_full_occupancy(sum/(256CU*waves/CU))": round(float(c.sum() * 64 / 1e3 / (256 * 12)), 4)}))
        del ctx


if __name__ == "__main__":
    main()
