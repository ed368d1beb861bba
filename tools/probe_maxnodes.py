import os, sys, numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import bench
from linevis_amd import capi, host_api, scenes, camera, tiling, transfer_function as tfm
wl = "c3t"; W, H = 1920, 1080
tr = scenes.normalize(scenes.tornado())
attr = np.ascontiguousarray(tr.attributes, dtype=np.float32)
view, proj, fovy, near, far = camera.default_camera(W, H)
flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
for ov in ("false", "true"):
    c = capi.Context(0); c.set_option("line_width", 0.002)
    c.set_trajectories(tr.positions, attr, tr.line_offsets)
    c.set_transfer_function(tfm.standard(), *flow.attribute_range()); c.set_camera(view, proj, fovy, near, far, W, H)
    c.set_options(bench.WORKLOADS[wl]["settings"]); c.set_option("overlap_primary_passes", ov); c.build_accel()
    fn = tiling.hip_render_tiles_fn(c, 11, wait_for_consumer=False)
    all_tiles = tiling.make_tiles(W, H, 64)
    for world, rank in ((1, 0), (8, 2)):
        tiles = np.ascontiguousarray(all_tiles[rank::world])
        out = torch.zeros((len(tiles), 64, 64, 4), dtype=torch.uint8, device="cuda:0")
        c.set_option("collect_stats", True); fn(out, tiles, 64, 64); torch.cuda.synchronize(); st = c.stats()
        print("overlap", ov, "world", world, "max_nodes_per_pixel", st.max_nodes_per_pixel, "nodes", st.nodes_visited, "ao_nodes", st.ao_nodes_visited, "rays", st.rays_traced, "ao_rays", st.ao_rays_traced, flush=True)
        c.set_option("collect_stats", False)
        for _ in range(5): fn(out, tiles, 64, 64)
        torch.cuda.synchronize(); c.reset_timers()
        for _ in range(30): fn(out, tiles, 64, 64); torch.cuda.synchronize()
        print("   kernel ms", [round(float(x), 3) for x in c.stats().ms_kernel_avg[:3]], flush=True)
