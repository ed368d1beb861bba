"""Dev tool: per-rank compute time of the config-3 frame when sharded over WORLD ranks (run on one GPU: rank's tile list only)."""
import os, sys, time
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
ctx = capi.Context(0)
ctx.set_lines(pts, seg); ctx.set_transfer_function(tfm.standard(), *flow.attribute_range())
ctx.set_camera(view, proj, fovy, near, far, W, H); ctx.set_option("line_width", 0.002)
ctx.set_options(bench.SETTINGS)
ctx.build_accel()
all_tiles = tiling.make_tiles(W, H, 64)
for world in (1, 2, 4, 8):
    res = []
    for rank in range(world):
        tiles = tiling.assign_tiles(all_tiles, rank, world)
        out = torch.zeros((len(tiles), 64, 64, 4), dtype=torch.uint8, device="cuda:0")
        for _ in range(3):
            ctx.render_tiles_device(out.data_ptr(), tiles, 64, 64, mode=11)
        torch.cuda.synchronize(); ctx.reset_timers()
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.render_tiles_device(out.data_ptr(), tiles, 64, 64, mode=11)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10 * 1e3
        st = ctx.stats()
        res.append((dt, [round(x, 3) for x in st.ms_kernel_avg[:4]]))
    worst = max(res)
    print("world", world, "max rank ms %.3f" % worst[0], "kernels", worst[1], "all:", [round(r[0], 3) for r in res])
