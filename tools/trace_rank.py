"""Dev tool: the kernel timeline of ONE rank's frame (tile list of rank LV_TRACE_RANK of LV_TRACE_WORLD, round-robin deal, one frame in
flight) under rocprofv3 --kernel-trace.
  on the GPU box:  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python tools/trace_rank.py c3t
                   python tools/trace_rank.py --timeline <dir>     -> per-kernel start / duration / gap of a median frame
"""
import csv, glob, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)


def timeline(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    def short(n):
        return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    # frames: a frame starts with the first kernel after a gap > 150 us preceded by the marker kernel sequence; simpler: split on the
    # frame's first kernel name = the most frequent name that starts the repeating pattern of the last 40 % of the trace
    tail = ev[len(ev) * 6 // 10:]
    names = [e[0] for e in tail]
    first = None
    for cand in names:
        idx = [i for i, n in enumerate(names) if n == cand]
        if len(idx) >= 8:
            per = [idx[i + 1] - idx[i] for i in range(len(idx) - 1)]
            if len(set(per)) == 1:
                first = cand; period = per[0]; break
    if first is None:
        raise SystemExit("no repeating frame pattern found")
    starts = [i for i, n in enumerate(names) if n == first]
    frames = [tail[s:s + period] for s in starts[:-1]]
    # align on the kernel after the longest idle gap inside the period (= the host's frame boundary)
    gaps = [frames[2][(k + 1) % period][1] - frames[2][k][2] for k in range(period - 1)]
    rot = (max(range(period - 1), key=lambda k: gaps[k]) + 1) % period
    flat = [e for fr in frames for e in fr][rot:]
    frames = [flat[i:i + period] for i in range(0, len(flat) - period + 1, period)]
    frames.sort(key=lambda fr: fr[-1][2] - fr[0][1])
    fr = frames[len(frames) // 2]
    t0 = fr[0][1]
    out = []
    prev_end = t0
    for n, s, e in fr:
        out.append({"kernel": n, "start_us": round((s - t0) / 1e3, 1), "dur_us": round((e - s) / 1e3, 1), "gap_before_us": round((s - prev_end) / 1e3, 1)})
        prev_end = max(prev_end, e)
    rep = {"frames": len(frames), "median_frame_us": round((fr[-1][2] - t0) / 1e3, 1), "kernels": out,
           "busy_us": round(sum(o["dur_us"] for o in out), 1), "idle_us": round(sum(max(o["gap_before_us"], 0) for o in out), 1)}
    print(json.dumps(rep, indent=1))


if len(sys.argv) > 2 and sys.argv[1] == "--timeline":
    timeline(sys.argv[2])
    raise SystemExit(0)

import numpy as np, torch
import bench
from linevis_amd import capi, host_api, scenes, camera, tiling, transfer_function as tfm
wl = sys.argv[1] if len(sys.argv) > 1 else "c3t"
W, H = bench.WORKLOADS[wl].get("resolution", (1920, 1080))
tr = scenes.normalize({"tornado": scenes.tornado, "helix": scenes.helix_bundle, "rayleigh_benard": scenes.rayleigh_benard}[bench.WORKLOADS[wl]["scene"]]())
flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
pts, seg, _ = flow.tube_aabb_render_data(0.002)
view, proj, fovy, near, far = camera.default_camera(W, H)
c = capi.Context(0)
c.set_lines(pts, seg); c.set_transfer_function(tfm.standard_transparent() if bench.WORKLOADS[wl].get("transparent") else tfm.standard(), *flow.attribute_range())
c.set_camera(view, proj, fovy, near, far, W, H); c.set_option("line_width", 0.002)
if bench.WORKLOADS[wl].get("mesh"):
    c.set_tube_triangle_mesh(*flow.tube_triangle_render_data(0.002, 6))
c.set_options(bench.WORKLOADS[wl]["settings"])
c.set_options(dict(kv.split("=", 1) for kv in os.environ.get("LV_PROBE_SET", "").split(",") if kv))
c.set_option("kernel_timers", "none")
c.build_accel()
fn = tiling.hip_render_tiles_fn(c, bench.WORKLOADS[wl].get("mode", 11), wait_for_consumer=False)
world, rank = int(os.environ.get("LV_TRACE_WORLD", "8")), int(os.environ.get("LV_TRACE_RANK", "2"))
tiles = np.ascontiguousarray(tiling.make_tiles(W, H, 64)[rank::world])
out = torch.zeros((len(tiles), 64, 64, 4), dtype=torch.uint8, device="cuda:0")
for k in range(int(os.environ.get("LV_TRACE_FRAMES", "40"))):
    fn(out, tiles, 64, 64)
    torch.cuda.synchronize()
