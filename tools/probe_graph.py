#!/usr/bin/env python3
"""Would a HIP graph of the frame's launches shorten the frame?  Captures the launches of one frame (lv_render_device on a torch
stream) into a graph with torch.cuda.graph and replays it, against the same frames launched the normal way.  The replayed graph renders
the SAME frame every time (kernel arguments are baked in): a timing probe, not a product path.
usage (GPU box): python tools/probe_graph.py [c4|c2|c3c] [frames]"""
import json
import os
import sys
import time

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench  # noqa: E402
from linevis_amd import camera, capi, host_api, scenes, transfer_function as tfm  # noqa: E402

wkey = sys.argv[1] if len(sys.argv) > 1 else "c4"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
wl = bench.WORKLOADS[wkey]
W, H = wl.get("resolution", (1920, 1080))
view, proj, fovy, near, far = camera.default_camera(W, H)
gen = {"tornado": scenes.tornado, "helix": scenes.helix_bundle, "rayleigh_benard": scenes.rayleigh_benard}[wl["scene"]]
tr = scenes.normalize(gen())
flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
pts, seg, _ = flow.tube_aabb_render_data(bench.LINE_WIDTH)
ctx = capi.Context(0)
ctx.set_lines(pts, seg)
ctx.set_transfer_function(tfm.standard_transparent() if wl.get("transparent") else tfm.standard(), *flow.attribute_range())
ctx.set_camera(view, proj, fovy, near, far, W, H)
ctx.set_option("line_width", bench.LINE_WIDTH)
ctx.set_options(wl["settings"])
ctx.build_accel()
stream = torch.cuda.Stream()
ctx.set_stream(stream.cuda_stream)
img = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")


def timed(fn, n):
    with torch.cuda.stream(stream):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        b.record(stream)
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n, (time.perf_counter() - t0) * 1e3 / n


out = {"workload": wkey, "frames": frames}
out["stream_ms"], out["stream_host_ms"] = timed(lambda: ctx.render_device(img.data_ptr(), mode=wl["mode"]), frames)
ref = img.clone()
try:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        ctx.render_device(img.data_ptr(), mode=wl["mode"])
    out["graph_ms"], out["graph_host_ms"] = timed(g.replay, frames)
    out["graph_image_identical"] = bool(torch.equal(img, ref))
except Exception as e:  # noqa: BLE001
    out["graph_error"] = str(e)[:400]
print(json.dumps(out))
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
with open(os.path.join(R, "gpurun_out", "probe_graph_%s.json" % wkey), "w") as f:
    json.dump(out, f, indent=1)
