"""Dev tool: frames in flight.  Two contexts (scene replicas) on two HIP streams render alternate frames, so that the
latency-bound tile kernels of one frame overlap the AO sample kernel of the other.  Measured on ONE GPU for the full tile list
and for the tile list rank 0 of 8 would own.  Usage: python tools/probe_pipeline.py [c3c|c3t]"""
import json, os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
from linevis_amd import capi, host_api, scenes, camera, tiling, transfer_function as tfm
wl = sys.argv[1] if len(sys.argv) > 1 else "c3c"
W, H = 1920, 1080
tr = scenes.normalize(scenes.tornado())
flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
pts, seg, _ = flow.tube_aabb_render_data(0.002)
view, proj, fovy, near, far = camera.default_camera(W, H)
mesh = flow.tube_triangle_render_data(0.002, 6) if bench.WORKLOADS[wl].get("mesh") else None


def make():
    ctx = capi.Context(0)
    ctx.set_lines(pts, seg); ctx.set_transfer_function(tfm.standard(), *flow.attribute_range())
    ctx.set_camera(view, proj, fovy, near, far, W, H); ctx.set_option("line_width", 0.002)
    if mesh is not None:
        ctx.set_tube_triangle_mesh(*mesh)
    ctx.set_options(bench.WORKLOADS[wl]["settings"])
    ctx.build_accel()
    return ctx


all_tiles = tiling.make_tiles(W, H, 64)
ctxs = [make() for _ in range(2)]
report = {"workload": wl}
for label, tiles in (("all_tiles", all_tiles), ("rank0_of_8", np.ascontiguousarray(all_tiles[0::8]))):
    outs = [torch.zeros((len(tiles), 64, 64, 4), dtype=torch.uint8, device="cuda:0") for _ in range(2)]
    for depth in (1, 2):
        fns = [tiling.hip_render_tiles_fn(ctxs[i], 11, wait_for_consumer=False) for i in range(depth)]
        for k in range(6):
            fns[k % depth](outs[k % depth], tiles, 64, 64)
        torch.cuda.synchronize()
        n = 60
        t0 = time.perf_counter()
        for k in range(n):
            fns[k % depth](outs[k % depth], tiles, 64, 64)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        report["%s_frames_in_flight_%d_ms" % (label, depth)] = round(ms, 4)
        print(label, "frames in flight", depth, "%.4f ms/frame" % ms, flush=True)
    same = bool(torch.equal(outs[0], outs[1]))
    report[label + "_identical_frames"] = same
    print(label, "both contexts produce identical tiles:", same)
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
json.dump(report, open(os.path.join(R, "gpurun_out", "pipeline_probe_%s.json" % wl), "w"), indent=1)
