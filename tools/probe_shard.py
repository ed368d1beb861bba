"""Dev tool: per-rank compute time of the config-3 frame when sharded over WORLD ranks, measured on ONE GPU by rendering only
the tile list rank r of WORLD would own (no RCCL gather: that part needs the real node).  Compares the round-robin deal with the
cost-weighted deal of tiling.assign_tiles_by_cost (costs = RTAO hit pixels per tile of a full frame).
Usage: python tools/probe_shard.py [c3c|c3t] -> gpurun_out/shard_probe_<workload>.json"""
import json, os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
from linevis_amd import capi, host_api, scenes, camera, tiling, transfer_function as tfm
wl = sys.argv[1] if len(sys.argv) > 1 else "c3c"
W, H = bench.WORKLOADS[wl].get("resolution", (1920, 1080))
tr = scenes.normalize({"tornado": scenes.tornado, "helix": scenes.helix_bundle, "rayleigh_benard": scenes.rayleigh_benard}[bench.WORKLOADS[wl]["scene"]]())
flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
view, proj, fovy, near, far = camera.default_camera(W, H)
attr = np.ascontiguousarray(tr.attributes[0] if np.ndim(tr.attributes) == 2 else tr.attributes, dtype=np.float32)


def make():
    c = capi.Context(0)
    c.set_option("line_width", 0.002)
    c.set_trajectories(tr.positions, attr, tr.line_offsets)   # line points, index pairs and the tube mesh are written on the device
    c.set_transfer_function(tfm.standard(), *flow.attribute_range())
    c.set_camera(view, proj, fovy, near, far, W, H)
    c.set_options(bench.WORKLOADS[wl]["settings"])
    c.set_options(dict(kv.split("=", 1) for kv in os.environ.get("LV_PROBE_SET", "").split(",") if kv))   # e.g. overlap_primary_passes=false
    c.build_accel()
    return c


ctx = make()
DEPTHS = [int(x) for x in os.environ.get("LV_PROBE_DEPTHS", "1,2").split(",")]
extra = [make() for _ in range(max(DEPTHS) - 1)]   # further scene replicas for frames in flight (tiling.FramePipeline)
fns = [tiling.hip_render_tiles_fn(c, 11, wait_for_consumer=False) for c in [ctx] + extra]
all_tiles = tiling.make_tiles(W, H, 64)
out = torch.zeros((len(all_tiles), 64, 64, 4), dtype=torch.uint8, device="cuda:0")
fns[0](out, all_tiles, 64, 64)
torch.cuda.synchronize()
costs = ctx.ao_tile_costs().astype(np.float64) * 64.0 + 4.0 * 64 * 64


# LV_PROBE_SUBTILE=WxH: a rank still owns 64 x 64 tiles but hands them to the renderer as W x H rectangles (W, H divide 64).  A tile kernel
# gives every rectangle a 64 x 64 group of 64 waves, so 32x32 = the same pixels on four times the waves, each a quarter full: a shorter
# critical path for the primary-ray kernels of a rank that owns few tiles, at the price of more wave-level work
SUBW, SUBH = [int(x) for x in os.environ.get("LV_PROBE_SUBTILE", "64x64").split("x")]


def subdivide(tiles):
    if (SUBW, SUBH) == (64, 64):
        return tiles
    t = np.asarray(tiles, dtype=np.uint32)
    ox, oy = np.meshgrid(np.arange(0, 64, SUBW, dtype=np.uint32), np.arange(0, 64, SUBH, dtype=np.uint32), indexing="xy")
    sub = t[:, None, :] + np.stack([ox.reshape(-1), oy.reshape(-1)], axis=1)[None, :, :]
    sub = sub.reshape(-1, 2)
    return np.ascontiguousarray(sub[(sub[:, 0] < W) & (sub[:, 1] < H)])


def time_tiles(tiles, depth, reps=int(os.environ.get("LV_PROBE_REPS", "40"))):
    tiles = subdivide(tiles)
    outs = [torch.zeros((len(tiles), SUBH, SUBW, 4), dtype=torch.uint8, device="cuda:0") for _ in range(depth)]
    for k in range(4):
        fns[k % depth](outs[k % depth], tiles, SUBW, SUBH)
    torch.cuda.synchronize(); ctx.reset_timers()
    t0 = time.perf_counter()
    for k in range(reps):
        fns[k % depth](outs[k % depth], tiles, SUBW, SUBH)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    st = ctx.stats()
    return dt, [round(float(x), 3) for x in st.ms_kernel_avg[:3]]


report = {"workload": wl, "what": "per-rank frame time (ms) on ONE MI355X rendering only the tiles that rank would own; the RCCL gather "
          "is not included (unmeasured on hardware: no multi-GPU node)", "worlds": {}}
t1, k1 = time_tiles(all_tiles, 1)
report["one_gpu_ms"] = round(t1, 4)
if max(DEPTHS) >= 2:
    report["one_gpu_two_frames_in_flight_ms"] = round(time_tiles(all_tiles, 2)[0], 4)
for world in [int(x) for x in os.environ.get("LV_PROBE_WORLDS", "2,4,8").split(",")]:
    row = {}
    for name, parts in (("round_robin", [np.arange(r, len(all_tiles), world) for r in range(world)]),
                        ("cost_weighted", tiling.assign_tiles_by_cost(costs, world))):
        for depth in DEPTHS:
            res = [time_tiles(np.ascontiguousarray(all_tiles[ix]), depth) for ix in parts]
            worst = max(r[0] for r in res)
            key = "%s_frames_in_flight_%d" % (name, depth)
            row[key] = {"rank_ms": [round(r[0], 4) for r in res], "slowest_ms": round(worst, 4), "tiles": [int(len(ix)) for ix in parts],
                        "kernels_of_slowest": max(res)[1], "efficiency_vs_one_gpu": round(t1 / (world * worst), 4)}
            print(world, key, row[key], flush=True)
    report["worlds"][str(world)] = row
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
report["settings_override"] = os.environ.get("LV_PROBE_SET", "")
report["subtile"] = [SUBW, SUBH]
json.dump(report, open(os.path.join(R, "gpurun_out", "shard_probe_%s%s.json" % (wl, os.environ.get("LV_PROBE_TAG", ""))), "w"), indent=1)
