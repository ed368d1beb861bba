import sys, time
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R,'tests'))
import numpy as np
from common import *
from linevis_amd import capi
c = small_case(width=128, height=96)
ctx = c.hip_context()
print(capi.load().lv_version())
img = ctx.render(11)
ref, _ = c.oracle_render(11)
print('rt maxdiff', max_lsb_diff(img, ref), 'npix>0', int((np.abs(img.astype(int)-ref.astype(int)).max(axis=2)>0).sum()))
st = ctx.stats(); print(st.as_dict())
# rays
rng = np.random.default_rng(1)
o = np.tile(np.array([[0,0,0.8]],dtype=np.float32),(5000,1))
d = rng.normal(size=(5000,3)).astype(np.float32); d[:,2] = -np.abs(d[:,2])*3; d /= np.linalg.norm(d,axis=1,keepdims=True)
t,s,k = ctx.trace_rays(o,d,1e-4,1000.0)
sc = c.oracle_scene()
t2,s2,k2 = sc.trace_rays(o,d,1e-4,1000.0,c.line_width)
print('rays exact', np.array_equal(t.view(np.uint32),t2.view(np.uint32)), np.array_equal(s,s2), np.array_equal(k,k2), 'hits', int((s!=0xFFFFFFFF).sum()))
# AO
c2 = small_case(width=128, height=96, ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0, ambient_occlusion_iterations=2, ambient_occlusion_samples_per_frame=8, depth_cue_strength=0.8)
ctx2 = c2.hip_context()
img = ctx2.render(11)
ref, ao_ref = c2.oracle_render(11)
ao = ctx2.get_ao()
print('ao exact', np.array_equal(ao.view(np.uint32), ao_ref.view(np.uint32)), float(np.abs(ao-ao_ref).max()))
print('rt+ao maxdiff', max_lsb_diff(img, ref))
print('depth', ctx2.depth_range(), c2.oracle_scene().depth_range(c2.oracle_params()))
# PPLL
c3 = small_case(width=128, height=96, transparent=True)
ctx3 = c3.hip_context()
img = ctx3.render(2)
ref,_ = c3.oracle_render(2)
print('ppll maxdiff', max_lsb_diff(img, ref), ctx3.stats().as_dict())
