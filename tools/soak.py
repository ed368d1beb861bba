"""Dev tool: renders the config-3 / config-4 frames many times and checks that every frame is byte-identical to the first
(races in the wave-cooperative queues, hand-over or the drain phase would show up as rare differences)."""
import os, sys, time, zlib
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
from linevis_amd import capi, host_api, scenes, camera, tiling, transfer_function as tfm
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, H = 1920, 1080
tr = scenes.normalize(scenes.tornado())
flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
pts, seg, _ = flow.tube_aabb_render_data(0.002)
view, proj, fovy, near, far = camera.default_camera(W, H)
tiles = tiling.make_tiles(W, H, 64)
out = torch.zeros((len(tiles), 64, 64, 4), dtype=torch.uint8, device="cuda:0")
for name, mode, tf, settings in (("c3 rtao", 11, tfm.standard(), bench.SETTINGS),
                                 ("c4 loop", 11, tfm.standard_transparent(), {}),
                                 ("c4 mlat8", 11, tfm.standard_transparent(), {"use_mlat": True, "mlat_num_nodes": 8}),
                                 ("c4 ppll", 2, tfm.standard_transparent(), {"ppll_max_num_frags": 1024, "ppll_expected_avg_depth_complexity": 30})):
    ctx = capi.Context(0)
    ctx.set_lines(pts, seg); ctx.set_transfer_function(tf, *flow.attribute_range())
    ctx.set_camera(view, proj, fovy, near, far, W, H); ctx.set_option("line_width", 0.002)
    ctx.set_options(settings)
    first, bad = None, 0
    t0 = time.time()
    for i in range(N):
        ctx.render_tiles_device(out.data_ptr(), tiles, 64, 64, mode=mode)
        torch.cuda.synchronize()
        h = zlib.crc32(out.cpu().numpy().tobytes())
        if first is None: first = h
        elif h != first: bad += 1
    print("%-9s %d frames in %.1f s, %d differ from the first" % (name, N, time.time() - t0, bad))
