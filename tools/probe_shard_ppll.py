"""Dev tool: per-rank time of the config-4 PPLL frame (rasterised prism) when sharded over WORLD ranks, measured on ONE GPU by rendering
only the tile list rank r of WORLD would own (round-robin deal of 64 x 64 tiles; no gather).  python tools/probe_shard_ppll.py"""
import json, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
from linevis_amd import capi, host_api, scenes, camera, tiling, transfer_function as tfm
W, H = 1920, 1080
tr = scenes.normalize(scenes.tornado())
flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
pts, seg, _ = flow.tube_aabb_render_data(0.002)
view, proj, fovy, near, far = camera.default_camera(W, H)
c = capi.Context(0)
c.set_lines(pts, seg); c.set_transfer_function(tfm.standard_transparent(), *flow.attribute_range())
c.set_camera(view, proj, fovy, near, far, W, H); c.set_option("line_width", 0.002)
c.set_options(bench.WORKLOADS["c4"]["settings"])
c.build_accel()
fn = tiling.hip_render_tiles_fn(c, 2, wait_for_consumer=False)
tiles = tiling.make_tiles(W, H, 64)
res = {}
for world in (1, 2, 4, 8):
    per_rank = []
    for r in range(world):
        own = tiles[r::world]
        out = torch.zeros((len(own), 64, 64, 4), dtype=torch.uint8, device="cuda:0")
        for _ in range(3):
            fn(out, own, 64, 64)
        torch.cuda.synchronize()
        c.reset_timers()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn(out, own, 64, 64)
        e1.record(); torch.cuda.synchronize()
        st = c.stats()
        per_rank.append({"ms": round(e0.elapsed_time(e1) / 20, 4),
                         "kernels": {n: round(float(st.ms_kernel_avg[i]), 4) for i, n in enumerate(capi.KERNEL_NAMES) if st.kernel_launches[i]}})
    res[world] = {"max_ms": max(p["ms"] for p in per_rank), "ranks": per_rank}
    print(world, res[world]["max_ms"], per_rank[0]["kernels"])
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(R, "gpurun_out", "shard_probe_c4.json"), "w"), indent=1)
