"""Determinism soak (was tools/soak.py): 200 consecutive full-size frames of each headline path reproduce the first frame byte
for byte -- a race in the wave-cooperative queues, the subtree hand-over, the drain phase, the tile-segmented G-buffer or the
PPLL chunk allocator would show up as a rare difference."""
import os
import zlib

import numpy as np
import pytest

from linevis_amd import camera, scenes, tiling, transfer_function as tfm

FRAMES = int(os.environ.get("LV_SOAK_FRAMES", "200"))   # one-off longer runs: LV_SOAK_FRAMES=5000


@pytest.mark.gpu
@pytest.mark.slow
def test_200_frames_are_byte_identical(hip_lib):
    import torch
    import bench
    from linevis_amd import capi, host_api
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
                                     ("c4 ppll", 2, tfm.standard_transparent(),
                                      {"ppll_max_num_frags": 1024, "ppll_expected_avg_depth_complexity": 30})):
        ctx = capi.Context(0)
        ctx.set_lines(pts, seg)
        ctx.set_transfer_function(tf, *flow.attribute_range())
        ctx.set_camera(view, proj, fovy, near, far, W, H)
        ctx.set_option("line_width", 0.002)
        ctx.set_options(settings)
        first, bad = None, 0
        for _ in range(FRAMES):
            ctx.render_tiles_device(out.data_ptr(), tiles, 64, 64, mode=mode)
            torch.cuda.synchronize()
            h = zlib.crc32(out.cpu().numpy().tobytes())
            if first is None:
                first = h
            elif h != first:
                bad += 1
        assert bad == 0, "%s: %d of %d frames differ from the first" % (name, bad, FRAMES)
        ctx.close()
