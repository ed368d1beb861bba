import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
from linevis_amd import scenes, camera, transfer_function as tfm, capi, host_api
tr = scenes.normalize(scenes.tornado())
flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
pts, seg, _ = flow.tube_aabb_render_data(0.002)
for (W, H) in [(1920, 1080), (960, 540), (480, 270), (3840, 2160)]:
    ctx = capi.Context(0)
    ctx.set_lines(pts, seg); ctx.set_transfer_function(tfm.standard(), *flow.attribute_range())
    view, proj, fovy, near, far = camera.default_camera(W, H)
    ctx.set_camera(view, proj, fovy, near, far, W, H); ctx.set_option('line_width', 0.002)
    for halos in (True, False):
        ctx.set_option('use_halos', halos)
        for i in range(3): ctx.render(11)
        s = ctx.stats()
        ctx.set_option('collect_stats', True); ctx.render(11); s2 = ctx.stats(); ctx.set_option('collect_stats', False)
        print(W, H, 'halos', halos, 'color ms %.3f' % s.ms_color, 'rays', s2.rays_traced, 'nodes', s2.nodes_visited, 'maxnodes', s2.max_nodes_per_pixel)
