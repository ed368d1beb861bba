import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
from linevis_amd import scenes, camera, transfer_function as tfm, capi, host_api
tr = scenes.normalize(scenes.tornado())
flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
pts, seg, _ = flow.tube_aabb_render_data(0.002)
W, H = 960, 540
ctx = capi.Context(0)
ctx.set_lines(pts, seg); ctx.set_transfer_function(tfm.standard(), *flow.attribute_range())
view, proj, fovy, near, far = camera.default_camera(W, H)
ctx.set_camera(view, proj, fovy, near, far, W, H); ctx.set_option('line_width', 0.002); ctx.set_option('use_halos', False)
for i in range(5): ctx.render(11)
print(os.environ.get('LV_LIB_PATH','default').split('/')[-1], 'color ms %.3f' % ctx.stats().ms_color)
