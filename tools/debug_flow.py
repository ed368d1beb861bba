import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from linevis_amd import capi
from oracle import lvo
from test_gpu_flow import abc_grid
v, mag, sp = abc_grid()
rng = np.random.default_rng(6)
seeds = rng.uniform(0.05, 0.95, (300, 3)).astype(np.float32)
ctx = capi.Context(0)
ctx.set_flow_grid(v, sp, [mag])
for scale in (1.0, 4.0):
    S = dict(time_step_scale=scale, minimum_length=0.0)
    a = ctx.trace_streamlines(seeds, capi.streamline_settings("Implicit Euler", "Forward", **S))
    b = lvo.trace_streamlines(v, sp, [mag], seeds, lvo.streamline_settings("Implicit Euler", "Forward", **S))
    print(scale, len(a[0]), len(b[0]), np.array_equal(a[2], b[2]))
    if np.array_equal(a[2], b[2]):
        d = np.nonzero((a[0].view(np.uint32) != b[0].view(np.uint32)).any(axis=1))[0]
        print("mismatching points", len(d), "of", len(a[0]))
        if len(d):
            i = d[0]; l = np.searchsorted(a[2], i, side="right") - 1
            print("line", l, "point", i - a[2][l], a[0][i], b[0][i], a[0][i] - b[0][i])
    else:
        na, nb = np.diff(a[2].astype(np.int64)), np.diff(b[2].astype(np.int64))
        print(len(na), len(nb)); k = np.nonzero(na[:min(len(na),len(nb))] != nb[:min(len(na),len(nb))])[0][:5]; print(k, na[k], nb[k])
