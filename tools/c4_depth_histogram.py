"""Dev tool (GPU box): per-pixel fragment-count histogram of the config-4 PPLL frame (how many pixels overflow MAX_NUM_FRAGS)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import Case
from linevis_amd import scenes, host_api, transfer_function as tfm
tr = scenes.normalize(scenes.tornado())
flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
pts, seg, _ = flow.tube_aabb_render_data(0.002)
c = Case(pts, seg, tfm.standard_transparent(), 1920, 1080, 0.002, ppll_max_num_frags=64, ppll_expected_avg_depth_complexity=20, use_capped_tubes=False)
ctx = c.hip_context()
ctx.set_transfer_function(c.tf, *flow.attribute_range())
ctx.render(2)
pw, ph = c.padded()
st = ctx.stats()
hn, hs, cnt = ctx.ppll_buffers(pw * ph, int(st.ppll_pool_nodes))
# per pixel counts by walking is slow; use lengths via start/next arrays: count nodes per list with numpy iteration
n = np.zeros(pw * ph, dtype=np.int64)
cur = hs.astype(np.int64).copy(); cur[hs == 0xFFFFFFFF] = -1
while (cur >= 0).any():
    m = cur >= 0
    n[m] += 1
    nxt = hn[cur[m], 2].astype(np.int64); nxt[nxt == 0xFFFFFFFF] = -1
    cur[m] = nxt
print("fragments", n.sum(), "pixels>0", (n > 0).sum(), "max", n.max())
for t in (16, 32, 64, 128, 256, 512, 1024):
    m = n > t
    print("n>%d: pixels %d fragments %d" % (t, m.sum(), n[m].sum()))
