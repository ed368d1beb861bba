"""Dev tool: SAH cost of the 4-wide tree + traversal counters for the accel builders (run on the GPU box)."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from linevis_amd import capi, host_api, scenes, camera, transfer_function as tfm

def decode(nodes):
    n = nodes.view(np.float32).reshape(-1, 16)
    u = nodes.reshape(-1, 16)
    origin = n[:, 0:3]; scale = np.stack([n[:, 3], n[:, 4], n[:, 5]], axis=1)
    q = u[:, 6:12]   # qmin x,y,z, qmax x,y,z words
    child = u[:, 12:16]
    boxes = np.zeros((len(n), 4, 6), np.float64)
    for k in range(4):
        for a in range(3):
            boxes[:, k, a] = origin[:, a] + ((q[:, a] >> (8 * k)) & 255) * scale[:, a].astype(np.float64)
            boxes[:, k, 3 + a] = origin[:, a] + ((q[:, 3 + a] >> (8 * k)) & 255) * scale[:, a].astype(np.float64)
    return boxes, child

def sah(ctx):
    st = ctx.stats()
    nodes, leaf = ctx.get_accel(st.num_nodes, st.num_segments)
    boxes, child = decode(nodes)
    valid = child != 0xFFFFFFFF
    d = boxes[..., 3:] - boxes[..., :3]
    area = d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0]
    root = np.where(valid[0][:, None], boxes[0], np.array([1e30] * 3 + [-1e30] * 3)[None])
    rb = np.concatenate([root[:, :3].min(axis=0), root[:, 3:].max(axis=0)])
    rd = rb[3:] - rb[:3]
    ra = rd[0] * rd[1] + rd[1] * rd[2] + rd[2] * rd[0]
    # cost: every child slot of a node is tested when the node is visited; a node is visited with prob area(its box)/area(root)
    # node box area = area of the slot that references it
    isleaf = valid & ((child & 0x80000000) != 0)
    inner = valid & ~isleaf
    node_area = np.zeros(len(nodes)); node_area[0] = ra
    node_area[child[inner]] = area[inner]
    cost_nodes = (node_area / ra).sum()
    cost_leaves = (area[isleaf] / ra).sum()
    return st.num_nodes, cost_nodes, cost_leaves, valid.sum(axis=1).mean()

wl = sys.argv[1] if len(sys.argv) > 1 else "tornado"
gen = {"tornado": scenes.tornado, "helix": scenes.helix_bundle, "rb": scenes.rayleigh_benard}[wl]
tr = scenes.normalize(gen())
flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
pts, seg, _ = flow.tube_aabb_render_data(0.002)
W, H = 1920, 1080
view, proj, fovy, near, far = camera.default_camera(W, H)
for builder in ("lbvh",):
    ctx = capi.Context(0)
    ctx.set_lines(pts, seg); ctx.set_transfer_function(tfm.standard(), *flow.attribute_range())
    ctx.set_camera(view, proj, fovy, near, far, W, H); ctx.set_option("line_width", 0.002)
    ctx.build_accel()
    print(builder, "build ms", ctx.stats().ms_accel_build, "nodes, visit-cost, leaf-cost, avg children:", sah(ctx))
