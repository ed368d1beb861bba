import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from common import *
from linevis_amd import host_api, scenes, transfer_function as tfm
tr = scenes.normalize(scenes.tornado())
flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
pts, seg, _ = flow.tube_aabb_render_data(0.002)
c = Case(pts, seg, tfm.standard(), 1920, 1080, 0.002)
ctx = c.hip_context()
rng = np.random.default_rng(77)
o = np.concatenate([np.tile(np.array([[0, 0, 0.8]], np.float32), (20000, 1)), rng.uniform(-0.2, 0.2, (20000, 3)).astype(np.float32)])
d = rng.normal(size=(40000, 3)).astype(np.float32)
d[:20000, 2] = -np.abs(d[:20000, 2]) * 3
d /= np.linalg.norm(d, axis=1, keepdims=True)
a = ctx.trace_rays(o, d, 1e-4, 1000.0)
sc = c.oracle_scene()
b = sc.trace_rays(o, d, 1e-4, 1000.0, 0.002, use_bvh=True)
bad = np.nonzero((a[0].view(np.uint32) != b[0].view(np.uint32)) | (a[1] != b[1]) | (a[2] != b[2]))[0]
print('mismatches', len(bad), bad[:20])
idx = bad[:40]
bf = sc.trace_rays(o[idx], d[idx], 1e-4, 1000.0, 0.002, use_bvh=False)
for j, i in enumerate(idx):
    print(i, 'hip', a[0][i], a[1][i], a[2][i], '| obvh', b[0][i], b[1][i], b[2][i], '| brute', bf[0][j], bf[1][j], bf[2][j])
