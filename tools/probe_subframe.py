"""Dev tool: ONE frame in flight, but a rank's tile list rendered as S sub-frames on S scene replicas / HIP streams (the tail of one
sub-frame's latency-bound primary-ray kernels runs under the other's RTAO sample kernel).  Times the frame of every rank of WORLD on ONE
GPU (no gather), for several ways of splitting the list.  Usage: python tools/probe_subframe.py [c3t] -> gpurun_out/subframe_probe_<wl>.json"""
import json, os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
from linevis_amd import capi, host_api, scenes, camera, tiling, transfer_function as tfm
wl = sys.argv[1] if len(sys.argv) > 1 else "c3t"
W, H = bench.WORKLOADS[wl].get("resolution", (1920, 1080))
gen = {"tornado": scenes.tornado, "helix": scenes.helix_bundle, "rayleigh_benard": scenes.rayleigh_benard}[bench.WORKLOADS[wl]["scene"]]
tr = scenes.normalize(gen())
attr = np.ascontiguousarray(tr.attributes[0] if np.ndim(tr.attributes) == 2 else tr.attributes, dtype=np.float32)
view, proj, fovy, near, far = camera.default_camera(W, H)
flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
SMAX = int(os.environ.get("LV_PROBE_STREAMS", "4"))


def make():
    c = capi.Context(0)
    c.set_option("line_width", 0.002)
    c.set_trajectories(tr.positions, attr, tr.line_offsets)
    c.set_transfer_function(tfm.standard(), *flow.attribute_range())
    c.set_camera(view, proj, fovy, near, far, W, H)
    c.set_options(bench.WORKLOADS[wl]["settings"])
    c.set_options(dict(kv.split("=", 1) for kv in os.environ.get("LV_PROBE_SET", "").split(",") if kv))
    c.set_option("kernel_timers", "none")
    c.build_accel()
    return c


ctxs = [make() for _ in range(SMAX)]
fns = [tiling.hip_render_tiles_fn(c, bench.WORKLOADS[wl]["mode"], wait_for_consumer=False) for c in ctxs]
all_tiles = tiling.make_tiles(W, H, 64)
full = torch.zeros((len(all_tiles), 64, 64, 4), dtype=torch.uint8, device="cuda:0")
fns[0](full, all_tiles, 64, 64)
torch.cuda.synchronize()
costs = ctxs[0].ao_tile_costs().astype(np.float64) * 64.0 + 4.0 * 64 * 64


def split(ix, how, s):
    """index lists of the s sub-frames of tile-index list ix"""
    ix = np.asarray(ix)
    if s == 1:
        return [ix]
    if how == "interleaved":
        return [ix[k::s] for k in range(s)]
    order = ix[np.argsort(-costs[ix], kind="stable")]          # heaviest first
    if how == "heavy_first":                                    # sub-frame 0 = the heaviest 1/s of the tiles, queued first
        return np.array_split(order, s)
    if how == "light_first":
        return np.array_split(order, s)[::-1]
    if how == "heavy_first_by_cost":                            # equal COST per sub-frame, heaviest tiles in sub-frame 0
        c = np.cumsum(costs[order]); cut = np.searchsorted(c, c[-1] * (np.arange(1, s) / s))
        return np.split(order, cut)
    raise SystemExit(how)


def time_parts(parts, reps=60):
    outs = [torch.zeros((max(len(p), 1), 64, 64, 4), dtype=torch.uint8, device="cuda:0") for p in parts]
    tl = [np.ascontiguousarray(all_tiles[p]) for p in parts]
    def frame():
        for k, t in enumerate(tl):
            if len(t):
                fns[k](outs[k], t, 64, 64)
    for _ in range(5):
        frame(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        frame()
        torch.cuda.synchronize()     # one frame in flight: the next frame starts when this one is complete
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), outs, tl


report = {"workload": wl, "what": "frame time (ms, host clock around submit + synchronize, median of 60) of each rank's tile list on ONE MI355X, "
          "ONE frame in flight, the list rendered as S sub-frames on S scene replicas / streams; no gather", "worlds": {}}
t1, _, _ = time_parts([np.arange(len(all_tiles))])
report["one_gpu_ms"] = round(t1, 4)
for world in [int(x) for x in os.environ.get("LV_PROBE_WORLDS", "8").split(",")]:
    deal = [np.arange(r, len(all_tiles), world) for r in range(world)]
    row = {}
    for s, how in [(1, "-")] + [(s, h) for s in (2, 4) if s <= SMAX for h in ("interleaved", "heavy_first", "light_first", "heavy_first_by_cost")]:
        res = []
        for ix in deal:
            ms, outs, tl = time_parts(split(ix, how, s))
            res.append(ms)
            if world == 8 and ix is deal[0]:     # the sub-frames' pixels are the whole frame's
                for o, t in zip(outs, tl):
                    for i, (x0, y0) in enumerate(t):
                        ti = int(np.where((all_tiles[:, 0] == x0) & (all_tiles[:, 1] == y0))[0][0])
                        assert torch.equal(o[i], full[ti]), "sub-frame tile differs from the whole frame"
        row["S%d_%s" % (s, how)] = {"rank_ms": [round(x, 4) for x in res], "slowest_ms": round(max(res), 4),
                                    "efficiency_vs_one_gpu": round(t1 / (world * max(res)), 4)}
        print(world, "S%d_%s" % (s, how), row["S%d_%s" % (s, how)], flush=True)
    report["worlds"][str(world)] = row
json.dump(report, open(os.path.join(R, "gpurun_out", "subframe_probe_%s.json" % wl), "w"), indent=1)
