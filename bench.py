#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json: Mrays/s + fps at 1920x1080, 1M-segment set,
64 spp RTAO; 1/2/4/8 GPUs).

One "step" = one complete frame of config 3: depth range -> RTAO (1 iteration x 64 samples per pixel, radius 0.1,
distance based, jittered primaries) -> ray-tracer colour pass (1 spp), on the synthetic 1M-segment tornado-style
streamline set, inputs resident in HBM.  With N > 1 the SAME frame is sharded by 64x64 screen tiles (Morton order,
round robin) over one process per GPU and assembled on rank 0 by one RCCL gather (strong scaling: total work fixed).

Prints ONE JSON line on rank 0.  `value` counts rays actually traced (primary + transparency continuation + AO), taken
from an untimed instrumented frame (same kernels with counters), times steps, divided by the max-over-ranks wall time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H = 1920, 1080
TILE = 64
LINE_WIDTH = 0.002
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E vendor peak (MI355X_MICROARCH.md; ~6.3 TB/s attainable)
SETTINGS = {
    "ambient_occlusion_mode": "RTAO (Screen Space)", "ambient_occlusion_strength": 1.0,
    "ambient_occlusion_gamma": 1.0, "ambient_occlusion_iterations": 1, "ambient_occlusion_samples_per_frame": 64,
    "ambient_occlusion_radius": 0.1, "ambient_occlusion_distance_based": True, "use_jittered_primary_rays": True,
    "num_samples_per_frame": 1, "depth_cue_strength": 0.0,
}
WORKLOAD = ("C3: 1M-segment tornado-style streamlines (1000 lines x 1001 points, seed 12345), 1920x1080, "
            "colour pass 1 spp + RTAO 64 spp (1 iteration x 64 samples, radius 0.1, distance based), line width 0.002")
# AO rays hit the analytic capsules of the segment LBVH (north_star's hot path); "c3t" runs the reference's own RTAO geometry
RTAO_CAPSULES = ", AO rays against the analytic capsules of the segment LBVH (rtao_geometry=capsules; workload c3t = triangle tubes)"
# secondary workloads (documentation runs: --workload c2 / c4); the default and the driver's runs are C3
WORKLOADS = {
    "c3": dict(name=WORKLOAD + RTAO_CAPSULES, scene="tornado", mode=11, settings=SETTINGS, kernel="k_ao_rays"),
    "c3t": dict(name=WORKLOAD + ", RTAO against the reference's 6-gon triangle tubes (12.06 M triangles, "
                     "rtao_geometry=triangle_tubes)",
                scene="tornado", mode=11, settings=dict(SETTINGS, rtao_geometry="triangle_tubes"), kernel="k_ao_rays"),
    "c5": dict(name="C5: 5M-segment Rayleigh-Benard-like convection rolls (5000 lines x 1001 points, seed 12345), 3840x2160, "
                    "colour pass 1 spp + RTAO 256 spp (1 iteration x 256 samples, radius 0.1, distance based), line width "
                    "0.002; BASELINE.json config 5 (meant for 8 GPUs; with fewer the same frame takes proportionally longer)",
               scene="rayleigh_benard", mode=11, settings=dict(SETTINGS, ambient_occlusion_samples_per_frame=256),
               kernel="k_ao_rays", resolution=(3840, 2160), ao_spp=256),
    "c2": dict(name="C2: 100k-segment helix bundle (100 lines x 1001 points, seed 12345), 1920x1080, primary rays only "
                    "(1 spp, pixel centres, AO off, depth cues off), line width 0.002",
               scene="helix", mode=11, settings={"num_samples_per_frame": 1, "depth_cue_strength": 0.0}, kernel="k_render_rt"),
    "c4": dict(name="C4: 1M-segment tornado-style streamlines, 1920x1080, PPLL OIT: all-hits gather + per-pixel 4-ary heap "
                    "resolve, MAX_NUM_FRAGS 64, node pool 20/pixel, tiling 2x8, opacity ramp 0.1..0.6",
               scene="tornado", mode=2, settings={"ppll_max_num_frags": 64, "ppll_expected_avg_depth_complexity": 20,
                                                  "depth_cue_strength": 0.0}, kernel="k_ppll_gather"),
    # the same transparent scene through the ray tracer's two transparency paths (documentation runs)
    "c4m": dict(name="C4 scene (1M-segment tornado, 1920x1080, opacity ramp 0.1..0.6) through the ray tracer with multi-layer "
                     "alpha tracing, 8 nodes (use_mlat): single pass, approximate OIT",
                scene="tornado", mode=11, settings={"use_mlat": True, "mlat_num_nodes": 8, "depth_cue_strength": 0.0,
                                                    "num_samples_per_frame": 1}, kernel="k_render_rt"),
    "c4l": dict(name="C4 scene (1M-segment tornado, 1920x1080, opacity ramp 0.1..0.6) through the ray tracer's transparency "
                     "loop (closest hit, step behind it, repeat until alpha > 0.99)",
                scene="tornado", mode=11, settings={"depth_cue_strength": 0.0, "num_samples_per_frame": 1},
                kernel="k_render_rt"),
}


def cpu_baseline(pts, seg, tf, attr_range, view, proj, fovy, near, far, target_seconds=15.0, workload="c3", mesh=None,
                 ao_spp=64):
    """CPU restatement of the LineVis GLSL path (the oracle, NOT LineVis's own binary), all host cores (OpenMP), on a
    centred crop of the same frame sized for ~target_seconds of work."""
    from oracle import lvo
    sc = lvo.Scene(pts, seg, tf)
    P = lvo.make_params(view, proj, W, H, fovY=fovy, nearDist=near, farDist=far, lineWidth=LINE_WIDTH,
                        useAmbientOcclusion=int(workload in ("c3", "c3t", "c5")), aoStrength=1.0, aoGamma=1.0, aoSamplesPerFrame=ao_spp,
                        aoIterations=1, aoUseDistance=1, aoJitterPrimary=1, aoRadius=0.1, attrMin=attr_range[0],
                        attrMax=attr_range[1], ppllMaxNumFrags=64)
    t0 = time.time()
    sc.build_bvh(LINE_WIDTH)
    tsc = None
    if workload == "c3t":
        tsc = lvo.TriScene(mesh[0], mesh[1], mesh[2], LINE_WIDTH)
        tsc._use_bvh(True)
    build_s = time.time() - t0

    def run(cw, ch):
        tile = ((W - cw) // 2, (H - ch) // 2, cw, ch)
        st = lvo.Stats()
        t = time.time()
        if workload == "c4":
            sc.render_ppll(P, tile=tile, use_bvh=True, stats=st)
        elif workload == "c4m":
            sc.render_rt_mlat(P, 8, tile=tile, use_bvh=True, stats=st)
        else:
            ao = None
            if workload in ("c3", "c5"):
                ao = sc.render_ao(P, tile=tile, use_bvh=True, stats=st)
            elif workload == "c3t":
                ao = tsc.render_ao(P, tile=tile, use_bvh=True, stats=st)
            sc.render_rt(P, ao=ao, tile=tile, use_bvh=True, stats=st)
        return time.time() - t, int(st.raysTraced)

    cw, ch = 96, 54                             # calibration crop, then grow towards ~target_seconds of work
    dt, rays = run(cw, ch)
    for _ in range(3):
        if dt >= 0.6 * target_seconds or (cw >= W and ch >= H):
            break
        grow = min((target_seconds / max(dt, 1e-3)) ** 0.5, 6.0)
        cw = min(W, int(round(cw * grow / 16.0)) * 16)
        ch = min(H, int(round(ch * grow / 9.0)) * 9)
        dt, rays = run(cw, ch)
    reps, total = 1, dt
    while total < 0.6 * target_seconds and reps < 200:   # whole frame is shorter than the sample: repeat it
        d2, _ = run(cw, ch)
        total += d2
        reps += 1
    dt = total / reps
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": lvo.num_threads(), "kind": "port",
            "sample": "centred %dx%d crop of the same frame in 16x16-pixel tiles over all OpenMP threads (dynamic, 1), %d pass(es), "
                      "%.1f s in total (%d rays per pass), CPU LBVH build %.1f s excluded; CPU restatement of the LineVis GLSL path (oracle), not LineVis's own binary"
                      % (cw, ch, reps, total, rays, build_s),
            "fps_extrapolated": round(1.0 / (dt * (W * H) / float(cw * ch)), 4)}


def measured_hbm_ceiling(device, torch):
    """Attainable HBM bandwidth of this box: device-to-device copy of 1 GiB (read + write bytes / time), best of 5
    (report against the vendor peak AND the measured ceiling)."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=device)
    b = torch.empty(n, dtype=torch.uint8, device=device)
    a.fill_(1)
    b.copy_(a)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 2.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--save-frame", default="")
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                             % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from linevis_amd import camera, capi, host_api, scenes, tiling, transfer_function as tfm

    wl = WORKLOADS[args.workload]
    global W, H
    W, H = wl.get("resolution", (1920, 1080))
    # ---- synthetic input (every rank builds the same replica; deterministic)
    gen = {"tornado": scenes.tornado, "helix": scenes.helix_bundle, "rayleigh_benard": scenes.rayleigh_benard}[wl["scene"]]
    tr = scenes.normalize(gen())
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    pts, seg, _ = flow.tube_aabb_render_data(LINE_WIDTH)
    tf = tfm.standard_transparent() if args.workload.startswith("c4") else tfm.standard()
    attr_range = flow.attribute_range()
    view, proj, fovy, near, far = camera.default_camera(W, H)

    ctx = capi.Context(local_rank)
    ctx.set_lines(pts, seg)
    ctx.set_transfer_function(tf, *attr_range)
    ctx.set_camera(view, proj, fovy, near, far, W, H)
    ctx.set_option("line_width", LINE_WIDTH)
    mesh = None
    if args.workload == "c3t":   # LineData::getLinePassTubeTriangleMeshRenderData -> the RTAO pass' geometry
        mesh = flow.tube_triangle_render_data(LINE_WIDTH, 6)
        ctx.set_tube_triangle_mesh(*mesh)
    ctx.set_options(wl["settings"])
    render_fn = tiling.hip_render_tiles_fn(ctx, wl["mode"])   # also moves the context onto torch's stream
    ctx.build_accel()
    build_first_ms = ctx.stats().ms_accel_build   # includes the one-time load of the library's code object (~5 ms)
    ctx.build_accel()                             # what a line-width change / new data costs from then on
    build_ms = ctx.stats().ms_accel_build
    sf = tiling.ShardedFrame(W, H, TILE, rank, world, device)

    def step():
        sf.render_local(render_fn)
        sf.gather()
        return sf.assemble_device()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- untimed instrumented frame: rays traced + algorithmic traffic of this rank's tiles
    ctx.set_option("collect_stats", True)
    step()
    torch.cuda.synchronize()
    st = ctx.stats()
    ctx.set_option("collect_stats", False)
    counters = torch.tensor([st.rays_traced, st.nodes_visited, st.prims_tested, st.hits_shaded, st.ao_hit_pixels,
                             st.fragments], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(counters)
    rays_per_frame = float(counters[0].item())
    counters_local = (st.nodes_visited, st.prims_tested, st.hits_shaded, st.ao_rays_traced, st.ao_nodes_visited,
                      st.ao_prims_tested)
    ao_diag = (st.ao_prim_hits, st.ao_prim_may_axis, st.ao_prim_may_both,
               [round(st.ao_phase_lanes[k] / max(64.0 * st.ao_phase_iterations[k], 1.0), 4) for k in range(3)])
    # algorithmic bytes of ONE k_ao_rays launch on this rank (DESIGN.md "Algorithmic bytes"):
    # 64 B per compressed 4-wide BVH node visited + 32 B per segment record tested + 52 B per compacted pixel (48 B G-buffer read,
    # 4 B AO factor write)
    prim_bytes = 48 if args.workload == "c3t" else 32   # 48-B triangle record / 32-B segment record
    ao_bytes = st.ao_nodes_visited * 64 + st.ao_prims_tested * prim_bytes + st.ao_hit_pixels * 52
    frame_bytes = (st.nodes_visited * 64 + st.prims_tested * 32 + st.hits_shaded * 96 + st.ao_hit_pixels * 52
                   + len(sf.local_tiles) * TILE * TILE * (4 + 4) + st.fragments * (12 + 4 + 4 + 12))
    kid = capi.KERNEL_NAMES.index(wl["kernel"])
    kernel_bytes = ao_bytes if args.workload in ("c3", "c3t", "c5") else frame_bytes  # c2 / c4: one traversal kernel dominates

    for _ in range(args.warmup):
        step()
    sync_all()
    ctx.reset_timers()
    sync_all()
    t0 = time.perf_counter()
    frame = None
    for _ in range(args.steps):
        frame = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    st = ctx.stats()   # per-kernel HIP-event averages over the timed region (this rank)

    if rank == 0:
        ms_rays = float(st.ms_kernel_avg[kid])
        achieved = kernel_bytes / (ms_rays * 1e-3) / 1e9 if ms_rays > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_r01.json")
        if os.path.exists(tpath) and world == 1 and args.workload == "c3":
            try:
                traffic = json.load(open(tpath)).get("k_ao_rays_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "Mrays/s", "value": round(rays_per_frame * args.steps / elapsed / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "fps": round(args.steps / elapsed, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["name"], "resolution": [W, H], "segments": int(len(seg)),
                       "rays_per_frame": int(rays_per_frame), "ao_hit_pixels": int(counters[4].item()),
                       "fragments_per_frame": int(counters[5].item()),
                       "parallelism": "screen tiles %dx%d, Morton order, round robin over %d GPU(s), one RCCL gather"
                                      % (TILE, TILE, world),
                       "accel_build_ms": round(build_ms, 3), "accel_build_first_ms": round(build_first_ms, 3),
                       "bvh_depth": int(st.bvh_depth),
                       "tube_triangles": int(st.num_tube_triangles)},
            "roofline": {"bound": "hbm", "kernel": wl["kernel"], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(kernel_bytes), "ms_per_launch": round(ms_rays, 4),
                         "launches_timed": int(min(st.kernel_launches[kid], 128)),
                         "frame_algorithmic_bytes_rank0": int(frame_bytes),
                         # compulsory floor: every node, segment record and line point read once + the
                         # frame's outputs -- what a perfect cache would leave of the algorithmic bytes
                         "frame_compulsory_bytes": int(st.num_nodes * 64 + len(seg) * 32 + len(pts) * 48 + W * H * (4 + 4)),
                         "note": "achieved = algorithmic bytes (64 B per node visited + 32/48 B per primitive tested + per-pixel "
                                 "records) / measured launch time; the scene + LBVH (~64 MB) live in L2 / Infinity Cache, so "
                                 "most of these bytes never reach HBM ('traffic' = PMC-measured HBM bytes per launch) and frac "
                                 "can exceed 1: the traversal kernels are VALU-issue-bound, not HBM-bound (DESIGN.md section 6)"},
            "counters_rank0": {"nodes_visited": int(counters_local[0]), "prims_tested": int(counters_local[1]),
                               "hits_shaded": int(counters_local[2]), "ao_rays": int(counters_local[3]),
                               "ao_nodes_visited": int(counters_local[4]), "ao_prims_tested": int(counters_local[5]),
                               "ao_prim_hits": int(ao_diag[0]), "ao_prim_may_axis": int(ao_diag[1]),
                               "ao_prim_may_both": int(ao_diag[2]),
                               "ao_phase_lane_utilisation": ao_diag[3]},
            "kernels_ms": {capi.KERNEL_NAMES[k]: round(float(st.ms_kernel_avg[k]), 4) for k in range(6)
                           if st.kernel_launches[k]},
        }
        if world == 1:
            ceil = measured_hbm_ceiling(device, torch)
            result["roofline"]["peak_measured"] = round(ceil, 1)          # copy bandwidth attainable on this box
            result["roofline"]["frac_of_measured"] = round(achieved / ceil, 5)
            if traffic:   # what actually crossed the HBM interface per launch (PMC) as a rate
                result["roofline"]["traffic_GBs"] = round(traffic / (ms_rays * 1e-3) / 1e9, 1)
        if args.save_frame and frame is not None:
            from PIL import Image
            Image.fromarray(frame.cpu().numpy()).save(args.save_frame)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(pts, seg, tf, attr_range, view, proj, fovy, near, far,
                                                  workload=args.workload, mesh=mesh, ao_spp=wl.get("ao_spp", 64))
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
