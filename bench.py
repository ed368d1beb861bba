#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json: Mrays/s + fps at 1920x1080, 1M-segment set,
64 spp RTAO; 1/2/4/8 GPUs).

One "step" = one complete frame of config 3: RTAO (1 iteration x 64 samples per pixel, radius 0.1, distance based,
jittered primaries) -> ray-tracer colour pass (1 spp), on the synthetic 1M-segment tornado-style streamline set, inputs
resident in HBM.  With N > 1 the SAME frame is sharded by 64x64 screen tiles over one process per GPU and assembled on rank 0
by one RCCL gather (strong scaling: total work fixed).

The timed headline (`value`) is the reference-faithful mode: AO rays against the reference's 6-gon triangle tubes
(VulkanRayTracedAmbientOcclusion.cpp:444-445).  The same frame with AO rays against the analytic capsules of the colour
pass (north_star's hot path, the library's default) is timed right after it and reported beside it (`value_capsules`).

Prints ONE JSON line on rank 0.  `value` counts rays actually traced (primary + transparency continuation + AO), taken
from an untimed instrumented frame (same kernels with counters), times steps, divided by the max-over-ranks wall time.
`roofline` reports the dominant kernel against every candidate ceiling (VALU issue, vector L1, L2, fabric / HBM) from the
PMC counters committed under profiles/ and the launch time measured live in this run; `bound` is the largest fraction.

--dry-run: no GPU -- gloo on CPU tensors with a stand-in renderer; exercises exactly the collective sequence of a
`--gpus N` run (tests/test_tiling_dist.py).
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TILE = 64
LINE_WIDTH = 0.002
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E vendor peak (MI355X_MICROARCH.md; ~6.3 TB/s attainable)
L2_PEAK_GBS = 34500.0     # aggregate L2 bandwidth (MI355X_MICROARCH.md "L2")
NUM_CUS, NUM_SIMDS = 256, 1024
PROFILE_TAGS = ("r06", "r05", "r04", "r03", "r02")   # counter files of the newest round that has them (profiles/pmc_<tag>_<workload>.json)
CLOCK_GHZ = 2.4                  # MI355X peak engine clock (MI355X_MICROARCH.md)
SETTINGS = {
    "ambient_occlusion_mode": "RTAO (Screen Space)", "ambient_occlusion_strength": 1.0,
    "ambient_occlusion_gamma": 1.0, "ambient_occlusion_iterations": 1, "ambient_occlusion_samples_per_frame": 64,
    "ambient_occlusion_radius": 0.1, "ambient_occlusion_distance_based": True, "use_jittered_primary_rays": True,
    "num_samples_per_frame": 1, "depth_cue_strength": 0.0,
}
C3 = ("C3: 1M-segment tornado-style streamlines (1000 lines x 1001 points, seed 12345), 1920x1080, "
      "colour pass 1 spp + RTAO 64 spp (1 iteration x 64 samples, radius 0.1, distance based), line width 0.002")
C3T = (C3 + ", AO rays against the reference's 6-gon triangle tubes (12.06 M triangles, rtao_geometry=triangle_tubes), colour pass "
      "with the reference's literal ray-capsule roots (intersection_form=literal, RayIntersectionTestsVulkan.glsl:39-119)")
C3C = (C3 + ", AO rays against the analytic capsules of the segment LBVH (rtao_geometry=capsules), closest-approach ray-capsule "
      "roots in both passes (intersection_form=closest_approach: NOT the reference's formula, DESIGN.md section 4)")
C5 = ("C5: 5M-segment Rayleigh-Benard-like convection rolls (5000 lines x 1001 points, seed 12345), 3840x2160, colour pass 1 spp + RTAO "
      "256 spp (1 iteration x 256 samples, radius 0.1, distance based), line width 0.002; BASELINE.json config 5 (meant for 8 GPUs; with "
      "fewer the same frame takes proportionally longer)")
C5T = (C5 + ", AO rays against the reference's 6-gon triangle tubes (60.3 M triangles, rtao_geometry=triangle_tubes), colour pass with the "
       "reference's literal ray-capsule roots (intersection_form=literal)")
WORKLOADS = {
    # the default: c3t timed as the headline, c3c timed right after it and reported beside it
    "c3": dict(name=C3T + "; the same frame with rtao_geometry=capsules is reported as value_capsules", scene="tornado", mode=11,
               settings=dict(SETTINGS, rtao_geometry="triangle_tubes", intersection_form="literal"), kernel="k_ao_rays", mesh=True,
               pmc="c3t", also="c3c"),
    "c3t": dict(name=C3T, scene="tornado", mode=11,
                settings=dict(SETTINGS, rtao_geometry="triangle_tubes", intersection_form="literal"), kernel="k_ao_rays", mesh=True),
    "c3c": dict(name=C3C, scene="tornado", mode=11, settings=dict(SETTINGS, rtao_geometry="capsules", intersection_form="closest_approach"),
                kernel="k_ao_rays"),
    # BASELINE.json config 5 on the geometry the reference's RTAO pass traces (VulkanRayTracedAmbientOcclusion.cpp:439-461: always the
    # triangle-tube TLAS) -- like c3's headline; c5c = the same frame on the analytic capsules (what "c5" meant until round 5)
    "c5": dict(name=C5T, scene="rayleigh_benard", mode=11,
               settings=dict(SETTINGS, rtao_geometry="triangle_tubes", intersection_form="literal", ambient_occlusion_samples_per_frame=256),
               kernel="k_ao_rays", resolution=(3840, 2160), ao_spp=256, mesh=True, pmc="c5t", cpu="c3t"),
    "c5t": dict(name=C5T, scene="rayleigh_benard", mode=11,
                settings=dict(SETTINGS, rtao_geometry="triangle_tubes", intersection_form="literal", ambient_occlusion_samples_per_frame=256),
                kernel="k_ao_rays", resolution=(3840, 2160), ao_spp=256, mesh=True, cpu="c3t"),
    "c5c": dict(name=C5 + ", AO rays against the analytic capsules of the segment LBVH (rtao_geometry=capsules, closest-approach roots: NOT a "
                     "frame the reference can render, DESIGN.md section 4)",
                scene="rayleigh_benard", mode=11, settings=dict(SETTINGS, rtao_geometry="capsules", ambient_occlusion_samples_per_frame=256),
                kernel="k_ao_rays", resolution=(3840, 2160), ao_spp=256, cpu="c3c"),
    "c2": dict(name="C2: 100k-segment helix bundle (100 lines x 1001 points, seed 12345), 1920x1080, primary rays only "
                    "(1 spp, pixel centres, AO off, depth cues off), line width 0.002, the reference's literal ray-capsule roots (the default)",
               scene="helix", mode=11, settings={"num_samples_per_frame": 1, "depth_cue_strength": 0.0}, kernel="k_render_rt"),
    "c2e": dict(name="C2 scene (100k-segment helix bundle) as band data: twisted ribbon directions (8 rad per unit length), the ray "
                     "tracer's Elliptic Tubes (sphere-traced tubelets, band width 0.005, min thickness 0.15) + USE_BANDS shading, "
                     "1920x1080, primary rays only",
                scene="helix", mode=11, ribbons=True, kernel="k_render_rt",
                settings={"num_samples_per_frame": 1, "depth_cue_strength": 0.0, "use_ribbons": True,
                          "use_analytic_elliptic_tubes": True, "band_width": 0.005, "min_band_thickness": 0.15}),
    "c4": dict(name="C4: 1M-segment tornado-style streamlines, 1920x1080, PPLL OIT: gather + per-pixel 4-ary heap resolve, "
                    "MAX_NUM_FRAGS 64, node pool 20/pixel, tiling 2x8, opacity ramp 0.1..0.6; fragments of the geometry the reference "
                    "RASTERISES (ppll_fragment_source=raster_prism: the uncapped 6-gon prism of the default programmable-pull mode, "
                    "LinePassProgrammablePullTubes.glsl:87-224, back faces culled, perspective-correct interpolated inputs), shaded "
                    "by the raster tube shader (ppll_fragment_colour=raster); the same frame with ppll_fragment_source=capsule_entry "
                    "(entry hits of analytic capsules: rounds 1-3) is reported as value_capsule_entry",
               scene="tornado", mode=2, settings={"ppll_max_num_frags": 64, "ppll_expected_avg_depth_complexity": 20,
                                                  "depth_cue_strength": 0.0, "use_capped_tubes": False,
                                                  "ppll_fragment_source": "raster_prism"},
               kernel="k_ppll_raster_prism", also_kernel="k_ppll_gather", transparent=True, also="c4c", also_suffix="capsule_entry",
               fast_also=True),
    "c4c": dict(name="C4 scene and settings with ppll_fragment_source=capsule_entry: fragments = entry hits of the pixel-centre ray "
                     "against the analytic (uncapped) capsules -- the probe of rounds 1-3, NOT the reference's geometry",
                scene="tornado", mode=2, settings={"ppll_max_num_frags": 64, "ppll_expected_avg_depth_complexity": 20,
                                                   "depth_cue_strength": 0.0, "use_capped_tubes": False,
                                                   "ppll_fragment_source": "capsule_entry"},
                kernel="k_ppll_gather", transparent=True),
    "c4m": dict(name="C4 scene (1M-segment tornado, 1920x1080, opacity ramp 0.1..0.6) through the ray tracer with multi-layer "
                     "alpha tracing, 8 nodes (use_mlat): single pass, approximate OIT",
                scene="tornado", mode=11, settings={"use_mlat": True, "mlat_num_nodes": 8, "depth_cue_strength": 0.0,
                                                    "num_samples_per_frame": 1}, kernel="k_render_rt", transparent=True),
    "c4l": dict(name="C4 scene (1M-segment tornado, 1920x1080, opacity ramp 0.1..0.6) through the ray tracer's transparency "
                     "loop (closest hit, step behind it, repeat until alpha > 0.99)",
                scene="tornado", mode=11, settings={"depth_cue_strength": 0.0, "num_samples_per_frame": 1},
                kernel="k_render_rt", transparent=True),
}


def host_cores():
    """Host cores this process may actually use: min(affinity, cgroup CPU quota).  The GPU boxes of this pool show 256 logical
    CPUs behind a 16-CPU quota (cpu.max 1600000 100000); oversubscribing the quota only adds throttling."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(W, H, pts, seg, tf, attr_range, view, proj, fovy, near, far, target_seconds=15.0, workload="c3c", mesh=None,
                 ao_spp=64, capped=True):
    """CPU restatement of the LineVis GLSL path (the oracle, NOT LineVis's own binary): 16x16-pixel tiles handed out dynamically
    over the usable host cores (OpenMP), on a centred crop of the same frame sized for ~target_seconds of work."""
    from oracle import lvo
    cores = host_cores()
    lvo.set_num_threads(cores)
    sc = lvo.Scene(pts, seg, tf)
    rtao = workload in ("c3c", "c3t")
    P = lvo.make_params(view, proj, W, H, fovY=fovy, nearDist=near, farDist=far, lineWidth=LINE_WIDTH,
                        useAmbientOcclusion=int(rtao), aoStrength=1.0, aoGamma=1.0, aoSamplesPerFrame=ao_spp,
                        aoIterations=1, aoUseDistance=1, aoJitterPrimary=1, aoRadius=0.1, attrMin=attr_range[0],
                        attrMax=attr_range[1], ppllMaxNumFrags=64, useCappedTubes=int(bool(capped)))
    if workload == "c2e":
        P.useBands, P.useEllipticTubes, P.bandWidth, P.minBandThickness, P.minThickness = 1, 1, 0.005, 0.15, 0.15
    t0 = time.time()
    sc.build_bvh(P.bandWidth if P.useEllipticTubes else LINE_WIDTH)
    tsc = None
    if workload == "c3t":
        tsc = lvo.TriScene(mesh[0], mesh[1], mesh[2], LINE_WIDTH)
        tsc._use_bvh(True)
    build_s = time.time() - t0

    def run(cw, ch):
        tile = ((W - cw) // 2, (H - ch) // 2, cw, ch)
        st = lvo.Stats()
        t = time.time()
        if workload in ("c4", "c4c"):
            P.ppllFragmentSource = 1 if workload == "c4" else 0
            sc.render_ppll(P, tile=tile, use_bvh=True, stats=st)
        elif workload == "c4m":
            sc.render_rt_mlat(P, 8, tile=tile, use_bvh=True, stats=st)
        else:
            ao = None
            if workload == "c3c":
                ao = sc.render_ao(P, tile=tile, use_bvh=True, stats=st)
            elif workload == "c3t":
                ao = tsc.render_ao(P, tile=tile, use_bvh=True, stats=st)
            sc.render_rt(P, ao=ao, tile=tile, use_bvh=True, stats=st)
        return time.time() - t, int(st.raysTraced)

    t_all = time.time()
    # the CPU restatement evaluates the capsule roots the timed GPU frame does: the reference's literal ones (intersection_form
    # "auto") except where the frame's RTAO rays hit the analytic capsules (c3c asks for closest_approach, c5 gets it from "auto")
    lvo.set_default_intersection_form(workload != "c3c")
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
    lvo.set_default_intersection_form(False)
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "wall_seconds": round(time.time() - t_all + build_s, 2),
            "logical_cpus_visible": os.cpu_count(),
            "sample": "centred %dx%d crop of the same frame (%s) in 16x16-pixel tiles over %d OpenMP threads = the cores this "
                      "process may use (cgroup quota), %d pass(es), %.1f s in total (%d rays per pass), CPU LBVH build %.1f s "
                      "excluded; CPU restatement of the LineVis GLSL path (oracle), not LineVis's own binary"
                      % (cw, ch, workload, cores, reps, total, rays, build_s),
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


def _stats(x):
    raw = np.asarray(x, dtype=np.float64)
    x = np.sort(raw)
    if not len(x):
        return None
    return {"median": round(float(np.median(x)), 4), "p95": round(float(x[min(len(x) - 1, int(0.95 * len(x)))]), 4),
            "min": round(float(x[0]), 4), "max": round(float(x[-1]), 4), "mean": round(float(x.mean()), 4), "n": int(len(x)),
            "argmax": int(np.argmax(raw))}


def roofline(kernel, pmc_key, ms_launch, algorithmic_bytes, world):
    """Roofline of the dominant kernel (north_star: "achieved fraction of the HBM roofline").

    Top level, HBM: `achieved` = bytes that left the L2s per launch (PMC counters of profiles/pmc_<tag>_<workload>.json, gfx950
    correction 2 x FETCH_SIZE + WRITE_SIZE, Infinity-Cache hits included: an upper bound on HBM traffic) / the launch time measured
    live in THIS run, against the 8 TB/s vendor peak = `frac` = `hbm_frac`.  The kernels of this path are NOT bandwidth-bound (L1 /
    L2 serve most node and primitive fetches): the ceiling that binds is VALU issue, reported beside it as `valu_frac_datasheet`
    (wave-instructions against one wave64 VALU instruction per 2 cycles per SIMD at 2.4 GHz), `lane_utilisation`, and
    `valu_frac_own_mix` (against the issue rate tools/ubench measured for the node step's instruction mix on this hardware).
    SURVEY.md 8(d)'s byte MODEL ("every touch goes to memory") is kept as `algorithmic_bytes_per_launch` with its rate named
    `algorithmic_touch_rate_GBs`: it counts cache hits as memory traffic, exceeds the HBM peak and is not a bandwidth.
    `pmc_head` / `pmc_source_sha`: the commit and source hash the counters were collected on; `pmc_matches_build` compares the hash with
    the tree this run uses."""
    from linevis_amd import build as lv_build
    sha = lv_build.source_sha()
    out = {"kernel": kernel, "ms_per_launch": round(ms_launch, 4), "algorithmic_bytes_per_launch": int(algorithmic_bytes),
           "algorithmic_touch_rate_GBs": round(algorithmic_bytes / (ms_launch * 1e-3) / 1e9, 1) if ms_launch > 0 else None,
           "algorithmic_note": ("segment rasteriser: 96 B per segment (record + frames) + 4 B per coverage test + 16 B per fragment "
                                "(12-B record, count update): compulsory bytes of the launch" if kernel == "k_ppll_raster_prism" else
                                "SURVEY.md 8(d) byte model: 64 B per node visited + 32 B per primitive tested (48 B per triangle with triangle_leaf_records=triangles) + per-pixel records, "
                                "'every touch goes to memory'; most of these bytes are served by L1/L2 (see ceilings): NOT a "
                                "bandwidth and not bounded by the HBM peak"),
           "source_sha": sha}
    ppath = upath = None
    for tag in PROFILE_TAGS:
        c1 = os.path.join(ROOT, "profiles", "pmc_%s_%s.json" % (tag, pmc_key))
        if ppath is None and os.path.exists(c1):
            ppath = c1
        c2 = os.path.join(ROOT, "profiles", "ubench_%s.json" % tag)
        if upath is None and os.path.exists(c2):
            upath = c2
    none = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "hbm_frac": None,
            "valu_frac_datasheet": None, "lane_utilisation": None, "valu_frac_own_mix": None, "valu_busy": None, "clock_ghz_live": None,
            "pmc_head": None,
            "pmc_matches_build": False}
    if world != 1 or ppath is None or upath is None or ms_launch <= 0:
        out.update(none)
        out["note"] = ("counter-backed figures need profiles/pmc_<tag>_%s.json + profiles/ubench_<tag>.json (tags %s) and a 1-GPU run"
                       % (pmc_key, "/".join(PROFILE_TAGS)))
        return out
    try:   # auxiliary files: a damaged one must never take the bench line down
        pmc = json.load(open(ppath))
        ub = json.load(open(upath))
        ub["valu_ns_per_inst_per_simd"]["node_step_mix"], ub["tcp_lane_requests_per_cu_per_ns"], pmc["kernels"]
    except Exception as e:   # noqa: BLE001
        out.update(none)
        out["note"] = "counter-backed figures unavailable: %s / %s unreadable (%s)" % (os.path.basename(ppath), os.path.basename(upath), e)
        return out
    kname = [k for k in pmc["kernels"] if k.startswith((kernel + "<", kernel + "_mlat<")) or k == kernel]
    if not kname:
        out.update(none)
        out["note"] = "kernel %s not in %s" % (kernel, os.path.basename(ppath))
        return out
    c = max((pmc["kernels"][k] for k in kname), key=lambda d: d.get("SQ_INSTS_VALU", 0.0))   # the instantiation that ran
    t_ns = ms_launch * 1e6
    ceil = {}
    # VALU issue: wave-instructions per SIMD per ns against (a) the datasheet issue rate, (b) the rate the VALU sustains on the node
    # step's instruction mix (tools/ubench/valu_rates: plain VALU instructions issue every ~3.5-4.1 cycles, only v_pk_fma_f32 reaches
    # the datasheet's two FMAs per 4 cycles; the kernel's compare / select / convert mix cannot be packed)
    valu_rate = c["SQ_INSTS_VALU"] / NUM_SIMDS / t_ns
    valu_peak = 1.0 / ub["valu_ns_per_inst_per_simd"]["node_step_mix"]
    lane_util = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    ceil["valu_issue"] = {"achieved": round(valu_rate, 4), "unit": "wave-instructions/SIMD/ns",
                          "peak_datasheet": round(CLOCK_GHZ / 2.0, 4), "frac_datasheet": round(valu_rate * 2.0 / CLOCK_GHZ, 4),
                          "peak_own_mix": round(valu_peak, 4), "frac_own_mix": round(valu_rate / valu_peak, 4),
                          "lane_utilisation": round(lane_util, 4),
                          "source": "SQ_INSTS_VALU / 1024 SIMDs / launch time; datasheet = one wave64 VALU instruction per 2 cycles per "
                                    "SIMD at 2.4 GHz (the 157 TFLOP/s vector peak); own mix = tools/ubench/valu_rates 'node-step mix'"}
    # vector L1: lane requests per CU per ns against the all-hit rate of divergent dwordx4 gathers
    tcp_rate = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / NUM_CUS / t_ns
    tcp_peak = ub["tcp_lane_requests_per_cu_per_ns"]["own64_l1_hit"]
    ceil["vector_l1"] = {"achieved": round(tcp_rate, 4), "peak": round(tcp_peak, 4), "unit": "lane-requests/CU/ns",
                         "frac": round(tcp_rate / tcp_peak, 4),
                         "l1_hit_rate": round(1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"], 4),
                         "source": "TCP_TOTAL_CACHE_ACCESSES_sum / 256 CUs / launch time; peak = tools/ubench/tcp_rates own64, 16 KB set"}
    # L2: requests x 128-B lines against the aggregate L2 bandwidth
    l2_gbs = c["TCC_REQ_sum"] * 128.0 / t_ns
    ceil["l2"] = {"achieved": round(l2_gbs, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(l2_gbs / L2_PEAK_GBS, 4),
                  "hit_rate": round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4),
                  "source": "TCC_REQ_sum x 128 B / launch time"}
    # fabric / HBM: bytes that left the L2s (Infinity-Cache hits included), gfx950 FETCH_SIZE correction
    traffic = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    hbm_gbs = traffic / t_ns
    ceil["hbm"] = {"achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_gbs / HBM_PEAK_GBS, 4),
                   "source": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 / launch time (MI355X_MICROARCH.md HBM: gfx950 FETCH_SIZE "
                             "counts 128-B requests as 64 B; Infinity-Cache hits are included, so this is an upper bound on HBM)"}
    # hardware-counter figures that need no calibration of ours: VALU busy = quad-cycles the VALUs were active x 4 /
    # SIMDs / elapsed engine cycles, the engine clock from GRBM_GUI_ACTIVE (summed over the 8 XCDs) over the live launch time
    valu_busy = clock_live = None
    if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_ACTIVE_INST_VALU"):
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0
        valu_busy = round(4.0 * c["SQ_ACTIVE_INST_VALU"] / NUM_SIMDS / cycles, 4)
        clock_live = round(cycles / t_ns, 4)
        ceil["valu_issue"]["valu_busy"] = valu_busy
        ceil["valu_issue"]["clock_ghz_live"] = clock_live
        ceil["valu_issue"]["busy_source"] = ("4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs), the definition of VERDICT r04; clock = "
                                             "those cycles / the launch time of this run.  SQ_ACTIVE_INST_VALU tracks SQ_INSTS_VALU within 1-5 % for "
                                             "every kernel of the frame (profiles/pmc_*.json): it weighs each wave64 VALU instruction as one 4-cycle "
                                             "pass, so a mix with shorter passes reads above 1 (capsule k_ao_rays: 1.08) -- 'issue-saturated', not a "
                                             "utilisation to four digits; GRBM_GUI_ACTIVE comes from another counter pass than the SQ counters")
    fracs = {"hbm": ceil["hbm"]["frac"], "l2": ceil["l2"]["frac"], "vector_l1": ceil["vector_l1"]["frac"],
             "valu_issue": ceil["valu_issue"]["frac_own_mix"]}
    pmc_sha = pmc.get("source_sha")
    out.update({"bound": "hbm", "achieved": ceil["hbm"]["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ceil["hbm"]["frac"],
                "traffic": int(traffic), "hbm_frac": ceil["hbm"]["frac"],
                "valu_frac_datasheet": ceil["valu_issue"]["frac_datasheet"], "lane_utilisation": ceil["valu_issue"]["lane_utilisation"],
                "valu_frac_own_mix": ceil["valu_issue"]["frac_own_mix"],
                "valu_busy": valu_busy, "clock_ghz_live": clock_live,
                "binding_ceiling": max(fracs, key=lambda k: fracs[k]),
                "traffic_over_algorithmic": round(traffic / algorithmic_bytes, 4) if algorithmic_bytes else None,
                "ceilings": ceil, "pmc_file": os.path.relpath(ppath, ROOT), "ubench_file": os.path.relpath(upath, ROOT),
                "pmc_head": pmc.get("git_head"), "pmc_source_sha": pmc_sha, "pmc_matches_build": bool(pmc_sha) and pmc_sha == sha})
    if not out["pmc_matches_build"]:
        out["pmc_warning"] = ("the counters of %s were collected on another build of the kernels (source hash %s, this tree %s): the "
                              "counter-backed fractions combine two builds" % (os.path.basename(ppath), pmc_sha, sha))
    return out


class DryRunContext:
    """Stand-in for capi.Context in --dry-run: fills tiles with a position pattern, no GPU."""

    class _S:
        rays_traced = nodes_visited = prims_tested = hits_shaded = ao_hit_pixels = fragments = 0
        ao_rays_traced = ao_nodes_visited = ao_prims_tested = ao_prim_hits = ao_prim_may_axis = ao_prim_may_both = 0
        ao_phase_lanes = [0, 0, 0]
        ao_phase_iterations = [1, 1, 1]
        ms_kernel_avg = [0.0] * 8
        kernel_launches = [0] * 8
        num_nodes = bvh_depth = num_tube_triangles = num_tri_nodes = 0
        ms_accel_build = ms_tri_accel_build = ms_tessellate = ms_line_points = 0.0

    def set_option(self, *a):
        pass

    def stats(self):
        return self._S()

    def reset_timers(self):
        pass

    def kernel_times(self, k):
        return np.zeros(0, np.float32)


def main_one_process(args):
    """All N GPUs behind one handle of the C-ABI (lv_create_multi): the path a C++ embedder takes.  Same frame, same counters;
    the gather is the library's own ncclSend / ncclRecv group (no torch.distributed)."""
    import torch
    from linevis_amd import camera, capi, host_api, scenes, transfer_function as tfm
    wl = dict(WORKLOADS[args.workload])
    extra = dict(kv.split("=", 1) for kv in args.set)
    if extra:
        wl["settings"] = dict(wl["settings"], **extra)
        wl["name"] += "; overrides: " + ", ".join("%s=%s" % kv for kv in sorted(extra.items()))
    W, H = wl.get("resolution", (1920, 1080))
    view, proj, fovy, near, far = camera.default_camera(W, H)
    gen = {"tornado": scenes.tornado, "helix": scenes.helix_bundle, "rayleigh_benard": scenes.rayleigh_benard}[wl["scene"]]
    tr = scenes.normalize(gen())
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    pts, seg, _ = flow.tube_aabb_render_data(LINE_WIDTH)
    ctx = capi.Context(devices=bench_devices(args.gpus), transport=args.transport)
    ctx.set_lines(pts, seg)
    ctx.set_transfer_function(tfm.standard_transparent() if wl.get("transparent") else tfm.standard(), *flow.attribute_range())
    ctx.set_camera(view, proj, fovy, near, far, W, H)
    ctx.set_option("line_width", LINE_WIDTH)
    if wl.get("mesh"):
        ctx.set_tube_triangle_mesh(*flow.tube_triangle_render_data(LINE_WIDTH, 6))
    ctx.set_options(wl["settings"])
    ctx.build_accel()
    dev0 = bench_devices(args.gpus)[0]
    torch.cuda.set_device(dev0)
    out = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:%d" % dev0)
    ctx.set_option("collect_stats", True)
    ctx.render_device(out.data_ptr(), mode=wl["mode"])
    st = ctx.stats()                       # synchronises every rank; counters summed over the ranks
    rays = float(st.rays_traced)
    ctx.set_option("collect_stats", False)
    ctx.rebalance()                        # tiles re-dealt by the cost this frame measured
    gc.collect(); gc.freeze(); gc.disable()
    for _ in range(args.warmup):
        ctx.render_device(out.data_ptr(), mode=wl["mode"])
    ctx.stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.render_device(out.data_ptr(), mode=wl["mode"])
    ctx.stats()
    elapsed = time.perf_counter() - t0
    gc.enable()
    deal = ctx.deal()
    per_rank = [ctx.rank_stats(r) for r in range(args.gpus)]
    multi = {"ranks_observed": int(ctx.num_ranks), "transport": args.transport,
             "tiles_per_rank": [int((deal == r).sum()) for r in range(args.gpus)],
             "render_ms_per_rank": [round(float(s.ms_total), 4) for s in per_rank],
             "ao_ms_per_rank": [round(float(s.ms_ao), 4) for s in per_rank],
             "colour_ms_per_rank": [round(float(s.ms_color + s.ms_ppll_gather + s.ms_ppll_resolve), 4) for s in per_rank],
             "gather_ms_estimate": round(elapsed / args.steps * 1e3 - max(float(s.ms_total) for s in per_rank), 4),
             "note": "render_ms = the last frame's lv_frame_render of each rank on its own stream (HIP events); gather_ms_estimate = "
                     "wall time per frame - the slowest rank's render (gather + scatter kernel + host queueing)"}
    result = {"metric": "Mrays/s", "value": round(rays * args.steps / elapsed / 1e6, 2), "unit": "Mrays/s", "n_gpus": args.gpus,
              "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
              "fps": round(args.steps / elapsed, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
              "frames_in_flight": 1, "data": "synthetic",
              "launch": os.environ.get("LV_BENCH_LAUNCH", "--one-process: lv_create_multi over %d device(s)" % args.gpus),
              "config": {"workload": wl["name"], "resolution": [W, H], "segments": int(len(seg)), "rays_per_frame": int(rays),
                         "parallelism": "ONE process, lv_create_multi over %d device(s): 64x64 screen tiles in Morton order dealt by measured "
                                        "cost, one %s gather per frame inside the library" % (args.gpus, args.transport),
                         "tiles_per_rank": [int((deal == r).sum()) for r in range(args.gpus)]},
              "multi_gpu": multi,
              "roofline": {"bound": None, "note": "per-kernel ceilings are reported by the default (one process per GPU) mode"},
              "cpu_baseline": None}
    emit_line(result)


def emit_line(result):
    """The ONE JSON line, as the last thing on stdout: native libraries of this process (RCCL prints a version banner through C stdio,
    which is block-buffered when stdout is a pipe) are flushed first, so that their text cannot land behind the line at exit."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:   # noqa: BLE001
        pass
    sys.stdout.flush()
    print(json.dumps(result), flush=True)


def _last_json_line(text):
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None


def _run_child(cmd, env, timeout_s):
    """Run a launch attempt in its own process group; on timeout the whole group (launcher + ranks) is killed."""
    import signal
    import subprocess
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout_s)
        return p.returncode, out, err
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        out, err = p.communicate()
        return None, out, "timeout after %.0f s\n%s" % (timeout_s, err[-800:])


def bench_devices(n):
    """Device ordinal of every rank: 0 .. N-1, or LV_BENCH_DEVICES="0,0" (a measurement / test knob: several ranks on one GPU work with
    the library's memcpy transport, not with RCCL, which refuses duplicate devices)."""
    if os.environ.get("LV_BENCH_DEVICES"):
        d = [int(x) for x in os.environ["LV_BENCH_DEVICES"].split(",")]
        if len(d) != n:
            raise SystemExit("LV_BENCH_DEVICES names %d devices, --gpus %d" % (len(d), n))
        return d
    return list(range(n))


def self_launch(args):
    """`python bench.py --gpus N` WITHOUT torchrun (the form the driver uses for N = 1): start the N ranks from here.

    1. re-run this command line under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one
       process per GPU, torch.distributed over RCCL -- exactly what the driver's own torchrun form starts) and forward rank 0's line;
    2. if that fails (no JSON line or a non-zero exit): ONE process over the library's multi-device handle (lv_create_multi, RCCL
       ncclSend / ncclRecv inside the library), then the same with peer memcpy as the gather transport.
    The line says which path ran (`launch`), and every failed attempt is kept in `launch_attempts`; progress goes to stderr."""
    import socket
    attempts = []
    argv = [a for a in sys.argv[1:]]
    env = dict(os.environ, LV_BENCH_LAUNCH="self-launched: python -m torch.distributed.run --nproc-per-node %d (one process per GPU, "
                                           "torch.distributed %s)" % (args.gpus, "gloo, dry run" if args.dry_run else "nccl = RCCL"))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus)))
    timeout_s = float(os.environ.get("LV_BENCH_LAUNCH_TIMEOUT", "1500"))
    rc, out, err, line = None, "", "", None
    for attempt in range(2):
        # a free port is picked by binding to 0 and releasing it: another process can take it before torchrun binds, so a
        # launch that dies in the rendezvous is tried once more on a fresh port before the one-process fallback takes over
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + argv
        print("[bench] --gpus %d without torch.distributed.run: launching %s" % (args.gpus, " ".join(cmd[1:8])), file=sys.stderr, flush=True)
        rc, out, err = _run_child(cmd, env, timeout_s)
        line = _last_json_line(out)
        if rc == 0 and line is not None:
            line["launch_attempts"] = attempts
            sys.stderr.write(err[-4000:])
            print(json.dumps(line), flush=True)
            return
        if attempt == 0 and rc is not None and any(k in err for k in ("Address already in use", "EADDRINUSE", "address already in use")):
            attempts.append({"path": "torch.distributed.run (port %d taken)" % port, "returncode": rc, "stderr_tail": err[-600:]})
            continue
        break
    attempts.append({"path": "torch.distributed.run", "returncode": rc, "stderr_tail": err[-1500:]})
    print("[bench] torch.distributed.run path failed (rc %s): ...%s" % (rc, err[-300:]), file=sys.stderr, flush=True)
    if args.dry_run:
        raise SystemExit("--dry-run --gpus %d: the torch.distributed.run launch failed and the dry run has no one-process form:\n%s"
                         % (args.gpus, attempts[-1]["stderr_tail"]))
    # the fallbacks run in child processes too: a failed RCCL bootstrap must not take this process (and the line) down
    for transport in ([args.transport] if args.transport == "memcpy" else ["rccl", "memcpy"]):
        print("[bench] falling back to --one-process --transport %s" % transport, file=sys.stderr, flush=True)
        cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--one-process", "--transport", transport]
        env = dict(os.environ, LV_BENCH_LAUNCH="fallback: ONE process, lv_create_multi over %d devices, %s gather inside the library "
                                               "(the torch.distributed.run launch failed)" % (args.gpus, transport))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        rc, out, err = _run_child(cmd, env, timeout_s)
        line = _last_json_line(out)
        if rc == 0 and line is not None:
            line["launch_attempts"] = attempts
            print(json.dumps(line), flush=True)
            return
        attempts.append({"path": "one-process/" + transport, "returncode": rc, "stderr_tail": err[-1500:]})
        print("[bench] --one-process --transport %s failed (rc %s): ...%s" % (transport, rc, err[-300:]), file=sys.stderr, flush=True)
    raise SystemExit("bench.py --gpus %d: every launch path failed:\n%s" % (args.gpus, json.dumps(attempts, indent=1)))


def run_states(steps, device=0):
    """AutomaticPerformanceMeasurer's loop on the headless harness: every canned state of lv::getTestModes (InternalState.cpp:46-51,
    276-297, each twice) is applied with setNewState and rendered `steps` times on the config-3 scene; per state the mean frame time
    (wall clock around render(), image read-back included, like the reference's "Average Time (ms)" column) and the fps."""
    from linevis_amd import host_api, scenes, transfer_function as tfm
    tr = scenes.normalize(scenes.tornado())
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    r = host_api.HeadlessLineRenderer(11, device=device)
    r.set_rendering_resolution(1920, 1080)
    r.set_line_data(flow)
    r.set_new_settings({"line_width": LINE_WIDTH})
    rows = []
    for name, mode, res, settings in host_api.get_test_modes():
        r.set_transfer_function(tfm.standard_transparent() if mode == 2 else tfm.standard())   # Transparent_*.xml for the OIT states
        r.set_new_state(name, mode, settings, resolution=res)
        r.render_frame()
        t0 = time.perf_counter()
        for _ in range(steps):
            r.render_frame()
        dt = (time.perf_counter() - t0) / steps
        rows.append({"name": name, "rendering_mode": mode, "avg_time_ms": round(dt * 1e3, 3), "fps": round(1.0 / dt, 2)})
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--save-frame", default="")
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--dry-run", action="store_true", help="no GPU: gloo + CPU tensors + a stand-in renderer (collective sequence only)")
    ap.add_argument("--one-process", action="store_true",
                    help="drive all --gpus N devices from THIS process through the library's multi-device handle (lv_create_multi: one "
                         "context per device, tiles dealt by cost, one RCCL ncclSend/ncclRecv gather per frame) instead of one process "
                         "per GPU + torch.distributed; launch as plain `python bench.py --gpus N --one-process`")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "memcpy"], help="gather transport of --one-process")
    ap.add_argument("--states", action="store_true",
                    help="also walk the canned benchmark states of the reference's --perf harness (InternalState.cpp:46-51,276-297 = "
                         "lv::getTestModes) through the plugin surface and report ms / fps per state")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="extra renderer setting (SettingsMap key), e.g. --set intersection_form=literal; noted in config.workload")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LV_BENCH_FORCE_DIST=1 (test knob): run the N > 1 code path -- process group, tile list, gather, de-tiling, per-rank diagnostics, the
    # N = 1 value of the same run -- with a ONE-rank communicator: everything of a --gpus N run that one GPU can execute
    dist_on = world > 1 or bool(os.environ.get("LV_BENCH_FORCE_DIST"))
    one_process = args.one_process and not args.dry_run
    if one_process:
        if world != 1:
            raise SystemExit("--one-process is launched as ONE process (no torch.distributed.run)")
        return main_one_process(args)
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            return self_launch(args)   # plain `python bench.py --gpus N`: start the N ranks ourselves
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    dry = args.dry_run
    if dry:
        device = torch.device("cpu")
    else:
        local_rank = bench_devices(world)[local_rank]
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:   # (only without a launcher, i.e. the forced one-rank group)
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            sk.close()
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from linevis_amd import camera, tiling, transfer_function as tfm

    wl = dict(WORKLOADS[args.workload])
    extra_settings = dict(kv.split("=", 1) for kv in args.set)
    if extra_settings:
        wl["settings"] = dict(wl["settings"], **extra_settings)
        wl["name"] += "; overrides: " + ", ".join("%s=%s" % kv for kv in sorted(extra_settings.items()))
        wl.pop("also", None)
    W, H = wl.get("resolution", (1920, 1080))

    def sync_all():
        if not dry:
            torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    # ---- synthetic input (every rank builds the same replica; deterministic)
    pts = seg = tf = attr_range = mesh = flow = None
    prep = {}   # data set -> first frame: what a new data set / a line-width change costs outside the frame loop (never part of `value`)
    view, proj, fovy, near, far = camera.default_camera(W, H)
    if not dry:
        from linevis_amd import capi, host_api, scenes
        gen = {"tornado": scenes.tornado, "helix": scenes.helix_bundle, "rayleigh_benard": scenes.rayleigh_benard}[wl["scene"]]
        tr = scenes.normalize(gen())
        if wl.get("ribbons"):   # band data: getLinePassTubeAabbRenderData(false, ellipticTubes = true)
            tr = scenes.twisted_ribbons(tr, twist=8.0)
            flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets, tr.ribbon_directions)
            pts, seg, _ = flow.tube_aabb_render_data_elliptic(float(wl["settings"]["band_width"]))
        else:
            flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
            t_prep = time.perf_counter()
            pts, seg, _ = flow.tube_aabb_render_data(LINE_WIDTH)
            prep["line_points_host_ms"] = round((time.perf_counter() - t_prep) * 1e3, 1)
        tf = tfm.standard_transparent() if wl.get("transparent") else tfm.standard()
        attr_range = flow.attribute_range()
        if wl.get("mesh"):   # LineData::getLinePassTubeTriangleMeshRenderData -> the RTAO pass' geometry
            t_prep = time.perf_counter()
            mesh = flow.tube_triangle_render_data(LINE_WIDTH, int(wl["settings"].get("tube_num_subdivisions", 6)))
            prep["tessellation_host_ms"] = round((time.perf_counter() - t_prep) * 1e3, 1)

    # the geometry path of the timed contexts: lv_set_trajectories (round 6: the trajectories go to HBM, kernels write the line points,
    # the index pairs and the triangle tubes) for plain flow lines; LV_BENCH_HOST_GEOMETRY=1 / band data: the host-built arrays
    device_geometry = not wl.get("ribbons") and not os.environ.get("LV_BENCH_HOST_GEOMETRY")
    traj_attr = None if dry else np.ascontiguousarray(tr.attributes[0] if np.ndim(tr.attributes) == 2 else tr.attributes, dtype=np.float32)

    def make_context(w, wait_for_consumer=True, host_geometry=False):
        if dry:
            return DryRunContext(), None
        ctx = capi.Context(local_rank)
        ctx.set_option("line_width", LINE_WIDTH)
        t_up = time.perf_counter()
        if device_geometry and not host_geometry:
            ctx.set_trajectories(tr.positions, traj_attr, tr.line_offsets)   # upload + the a2 kernels, host-synchronous
            up = {"path": "lv_set_trajectories", "set_trajectories_ms": round((time.perf_counter() - t_up) * 1e3, 2),
                  "line_points_kernels_ms": round(float(ctx.stats().ms_line_points), 3),   # the a2 kernels inside that call
                  "trajectories_MB": round((tr.positions.nbytes + traj_attr.nbytes + tr.line_offsets.nbytes) / 1e6, 1)}
        else:
            ctx.set_lines(pts, seg)
            up = {"path": "lv_set_lines + lv_set_tube_triangle_mesh (host-built arrays)",
                  "lines_upload_ms": round((time.perf_counter() - t_up) * 1e3, 2), "lines_upload_MB": round((pts.nbytes + seg.nbytes) / 1e6, 1)}
            if w.get("mesh"):
                t_up = time.perf_counter()
                ctx.set_tube_triangle_mesh(*mesh)
                up["mesh_upload_ms"] = round((time.perf_counter() - t_up) * 1e3, 2)   # pageable host arrays -> HBM, incl. the index validation pass
                up["mesh_upload_MB"] = round(sum(m.nbytes for m in mesh) / 1e6, 1)
        ctx.set_transfer_function(tf, *attr_range)
        ctx.set_camera(view, proj, fovy, near, far, W, H)
        ctx.set_options(w["settings"])
        fn = tiling.hip_render_tiles_fn(ctx, w["mode"], wait_for_consumer=wait_for_consumer)   # a torch stream per context
        t_up = time.perf_counter()
        ctx.build_accel()
        up["first_build_accel_wall_ms"] = round((time.perf_counter() - t_up) * 1e3, 2)   # (tessellation +) both LBVHs, first launch of their kernels
        prep.setdefault("host_arrays_upload" if host_geometry or not device_geometry else "device", up)
        return ctx, fn

    sf = tiling.ShardedFrame(W, H, TILE, rank, world, device)

    def dry_render(out, tiles_xy, tw, th):
        for i, (x0, y0) in enumerate(tiles_xy):
            out[i, :, :, 0] = int(x0) // tw % 256
            out[i, :, :, 1] = int(y0) // th % 256
            out[i, :, :, 2] = rank
            out[i, :, :, 3] = 255

    # frames in flight: with N > 1 every rank owns 1/N of the tiles and the latency-bound tile kernels of a frame no longer
    # hide behind its AO kernel -- two scene replicas on two streams render alternate frames (tiling.FramePipeline;
    # measured on one GPU with the tile lists of 8 ranks, slowest rank: 1.18 ms per frame with F = 1, 0.86 with 2, 0.73 with 4,
    # profiles/shard_probe_*.json).  One GPU: the AO kernel fills the chip (5.08 vs 5.18 ms), F = 1.
    frames_in_flight = 1 if world == 1 else min(4, world)
    if os.environ.get("LV_FRAMES_IN_FLIGHT"):   # measurement knob (tools/probe_shard.py explores it per world size)
        frames_in_flight = max(1, int(os.environ["LV_FRAMES_IN_FLIGHT"]))

    def measure(w, wkey):
        """Instrumented frame (counters), warm-up, K timed frames; returns everything rank 0 reports for this workload."""
        ctx, render_fn = make_context(w, wait_for_consumer=(frames_in_flight == 1))
        if dry:
            render_fn = dry_render
        build_first_ms = ctx.stats().ms_accel_build   # includes the one-time load of the library's code object (~5 ms)
        if not dry:
            ctx.build_accel()                         # what a line-width change / new data costs from then on
        build_ms = ctx.stats().ms_accel_build
        if not dry and "line_width_change" not in prep:
            # what `lv_set_option("line_width", ...)` + the next frame's geometry costs: (device tessellation at the new radius +) the
            # segment LBVH + the triangle LBVH, wall clock incl. the host-synchronous read-backs of the builds; then back again
            ctx.set_option("line_width", LINE_WIDTH * 1.25)
            t_lw = time.perf_counter()
            ctx.build_accel()
            wall = (time.perf_counter() - t_lw) * 1e3
            s2 = ctx.stats()
            prep["line_width_change"] = {"wall_ms": round(wall, 3), "tessellate_ms": round(float(s2.ms_tessellate), 3),
                                         "accel_build_ms": round(float(s2.ms_accel_build), 3),
                                         "tri_accel_build_ms": round(float(s2.ms_tri_accel_build), 3) if w.get("mesh") else None,
                                         "path": "device" if device_geometry else "host arrays: the caller would re-tessellate on the host "
                                                 "(tessellation_host_ms) and upload the mesh again (mesh_upload_ms) first"}
            ctx.set_option("line_width", LINE_WIDTH)
            ctx.build_accel()
        slots = [(sf, render_fn)]
        extra = []
        for _ in range(frames_in_flight - 1):
            c2, f2 = make_context(w, wait_for_consumer=False)
            extra.append(c2)
            slots.append((tiling.ShardedFrame(W, H, TILE, rank, world, device), dry_render if dry else f2))
        pipe = tiling.FramePipeline(slots)

        def step():
            return pipe.submit()

        # ONE GPU owns every tile: nothing to gather or de-tile -- the frame is one lv_render_device call straight into the
        # [H, W, 4] image on the context's stream (LV_BENCH_TILED=1 keeps the N > 1 code path: tile list + de-tiling pass)
        direct = world == 1 and not dist_on and not dry and not os.environ.get("LV_BENCH_TILED")
        mark_stream = None
        if direct:
            image = torch.empty((H, W, 4), dtype=torch.uint8, device=device)
            mark_stream = render_fn.stream

            def step():   # noqa: F811
                ctx.render_device(image.data_ptr(), mode=w["mode"])
                return image

        # ---- untimed instrumented frame: rays traced + algorithmic traffic of this rank's tiles
        ctx.set_option("collect_stats", True)
        step()
        sync_all()
        st = ctx.stats()
        ctx.set_option("collect_stats", False)
        counters = torch.tensor([st.rays_traced, st.nodes_visited, st.prims_tested, st.hits_shaded, st.ao_hit_pixels,
                                 st.fragments], dtype=torch.float64, device=device)
        if dist_on:
            dist.all_reduce(counters)
        rays_per_frame = float(counters[0].item())
        # 32-B segment record; triangle tubes: 48-B triangle records, or 32 B per triangle of a 64-B pair record (triangle_leaf_records = pairs)
        prim_bytes = (32 if getattr(st, "tri_leaf_bytes", 0) == 64 else 48) if w.get("mesh") else 32
        ao_bytes = st.ao_nodes_visited * 64 + st.ao_prims_tested * prim_bytes + st.ao_hit_pixels * 52
        frame_bytes = (st.nodes_visited * 64 + st.prims_tested * 32 + st.hits_shaded * 96 + st.ao_hit_pixels * 52
                       + len(sf.local_tiles) * TILE * TILE * (4 + 4) + st.fragments * (12 + 4 + 4 + 12))
        kernel_bytes = ao_bytes if w["kernel"] == "k_ao_rays" else frame_bytes   # c2 / c4c: one traversal kernel dominates
        if w["kernel"] == "k_ppll_raster_prism":
            # the segment rasteriser: every segment's 32-B record + 64-B frames once, 4 B (requested-pixel mask) per coverage test,
            # 12-B record + 4-B count update per fragment; then the fragment stage (12 B record in, 8 B fragment out, 4 B offset) and
            # the resolve pass (8 B per fragment, 4 + 4 B per pixel in, 4 B out)
            kernel_bytes = len(seg) * 96 + st.prims_tested * 4 + st.fragments * 16
            frame_bytes = kernel_bytes + st.fragments * (12 + 8 + 4 + 8) + len(sf.local_tiles) * TILE * TILE * (4 + 4 + 4)
        local = dict(nodes_visited=int(st.nodes_visited), prims_tested=int(st.prims_tested), hits_shaded=int(st.hits_shaded),
                     ao_rays=int(st.ao_rays_traced), ao_nodes_visited=int(st.ao_nodes_visited),
                     ao_prims_tested=int(st.ao_prims_tested), ao_prim_hits=int(st.ao_prim_hits),
                     ao_prim_may_axis=int(st.ao_prim_may_axis), ao_prim_may_both=int(st.ao_prim_may_both),
                     ao_phase_lane_utilisation=[round(st.ao_phase_lanes[k] / max(64.0 * st.ao_phase_iterations[k], 1.0), 4)
                                                for k in range(3)])
        # ---- re-deal the tiles by the cost this frame measured (RTAO hit pixels per tile x samples + a fixed cost per tile):
        # one small all-reduce outside the timed region; with one GPU the deal is trivial
        if dist_on:
            if dry:
                local_cost = [float((int(x0) // TILE * 7 + int(y0) // TILE * 3) % 11) for x0, y0 in sf.local_tiles]
            elif w["kernel"] == "k_ao_rays" and w["settings"].get("ambient_occlusion_denoiser") != "SVGF":
                # (under SVGF every rank runs the RTAO pass on the whole viewport: there is no per-tile AO cost to balance)
                local_cost = ctx.ao_tile_costs().astype(np.float64) * float(w.get("ao_spp", 64))
            else:
                local_cost = None
            if local_cost is not None:
                full = sf.rebalance(local_cost, base_cost=4.0 * TILE * TILE)
                for sf2, _ in slots[1:]:
                    sf2.redeal(full, base_cost=4.0 * TILE * TILE)
        # the scene's host arrays keep millions of objects alive: a generation-2 collection in the middle of the timed loop costs
        # the host tens of ms.  Collect now, then keep the collector out of the warm-up and the timed frames.
        gc.collect()
        gc.freeze()
        gc.disable()
        # Instrumentation of the timed region: HIP events around the DOMINANT kernel's launches only (roofline.ms_per_launch comes from
        # them) -- every event record is a barrier packet that costs the stream 2 - 4 us, and a frame has a dozen when every kernel and
        # phase is bracketed (config 4: 6 % of the frame, config 2: 13 %, config 3: 0.7 %; EXPERIMENTS.md 11.10).  The other kernels'
        # durations are collected by a short fully instrumented pass AFTER the timed region.
        timed_kernel = w["kernel"]
        if not dry and timed_kernel in capi.KERNEL_NAMES and not os.environ.get("LV_BENCH_ALL_TIMERS"):
            for c in [ctx] + extra:
                c.set_option("kernel_timers", str(capi.KERNEL_NAMES.index(timed_kernel))
                             + ("," + str(capi.KERNEL_NAMES.index(w["also_kernel"])) if w.get("also_kernel") in capi.KERNEL_NAMES else ""))
        for _ in range(args.warmup):
            step()
        sync_all()
        ctx.reset_timers()
        sync_all()
        marks = []
        if not dry:   # per-frame durations on the stream the kernels and the gather run on (no host sync inside the loop)
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(min(args.steps, 512) + 1)]
            marks[0].record(mark_stream)
        t0 = time.perf_counter()
        frame = None
        for i in range(args.steps):
            frame = step()
            if i + 1 < len(marks):
                marks[i + 1].record(mark_stream)
        sync_all()
        elapsed = time.perf_counter() - t0
        gc.enable()
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        if dist_on:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        st = ctx.stats()
        # ---- N > 1: three untimed diagnostic frames, one at a time, so that the first SCALE record explains itself: per rank the
        # time to render its tiles, the time it spends in the gather, the tiles it owns, and what torch.distributed reports
        diag = None
        if dist_on:
            rec = []
            for _ in range(3):
                sync_all()
                dist.barrier()
                t_a = time.perf_counter()
                sf.render_local(render_fn)
                sync_all()
                t_b = time.perf_counter()
                sf.gather()
                sync_all()
                t_c = time.perf_counter()
                sf.assemble_device()
                sync_all()
                t_d = time.perf_counter()
                rec.append([(t_b - t_a) * 1e3, (t_c - t_b) * 1e3, (t_d - t_c) * 1e3])
            mine = [float(np.median([r[k] for r in rec])) for k in range(3)] + [float(len(sf.local_tiles))]
            allr = [None] * world
            dist.all_gather_object(allr, mine)
            diag = {"ranks_observed": int(dist.get_world_size()), "backend": str(dist.get_backend()),
                    "tiles_per_rank": [int(a[3]) for a in allr],
                    "render_ms_per_rank": [round(a[0], 4) for a in allr],
                    "gather_ms_per_rank": [round(a[1], 4) for a in allr],
                    "assemble_ms_rank0": round(allr[0][2], 4),
                    "gather_bytes_per_rank": int(sf.out.numel()), "frames_in_flight_timed": frames_in_flight,
                    "note": "medians of 3 untimed frames rendered ONE AT A TIME (host-synchronised between render, gather and "
                            "de-tiling; the timed frames overlap these phases and keep frames_in_flight frames queued); gather_ms of a "
                            "rank includes waiting for the slowest rank's render"}
            # the N = 1 value of the SAME run: rank 0 renders the whole frame alone (one lv_render_device call per frame, as the
            # --gpus 1 line does) while the other ranks wait, so that the line carries its own speed-up and efficiency
            if not dry:
                single = None
                if rank == 0:
                    image1 = torch.empty((H, W, 4), dtype=torch.uint8, device=device)
                    for _ in range(3):
                        ctx.render_device(image1.data_ptr(), mode=w["mode"])
                    torch.cuda.synchronize()
                    n1 = max(3, min(args.steps, 50))
                    t1 = time.perf_counter()
                    for _ in range(n1):
                        ctx.render_device(image1.data_ptr(), mode=w["mode"])
                    torch.cuda.synchronize()
                    ms1 = (time.perf_counter() - t1) / n1 * 1e3
                    ms_n = elapsed / args.steps * 1e3
                    single = {"ms_per_step": round(ms1, 4), "value": round(rays_per_frame / ms1 / 1e3, 2), "unit": "Mrays/s",
                              "steps": n1, "speedup": round(ms1 / ms_n, 4), "efficiency": round(ms1 / ms_n / world, 4),
                              "frame_identical_to_sharded": bool(frame is not None and torch.equal(image1, frame)),
                              "note": "rank 0 alone, whole frame in one lv_render_device call per frame, same scene / settings / "
                                      "build, timed after the sharded region while the other ranks wait at a barrier; "
                                      "speedup = this ms_per_step / the sharded ms_per_step"}
                    del image1
                dist.barrier()
                diag["single_gpu_same_run"] = single
        frame_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1)] if marks else []
        kernels = {}
        if not dry:
            for k, name in enumerate(capi.KERNEL_NAMES):   # the timed region: the dominant kernel only (see above)
                ks = _stats(ctx.kernel_times(k))
                if ks:
                    kernels[name] = dict(ks, source="timed region")
            if not os.environ.get("LV_BENCH_ALL_TIMERS"):
                # fully instrumented pass (untimed): every kernel and phase bracketed by events, for the other kernels' durations
                for c in [ctx] + extra:
                    c.set_option("kernel_timers", "all")
                ctx.reset_timers()
                for _ in range(min(args.steps, 30)):
                    step()
                sync_all()
                for k, name in enumerate(capi.KERNEL_NAMES):
                    ks = _stats(ctx.kernel_times(k))
                    if ks and name not in kernels:
                        kernels[name] = dict(ks, source="instrumented pass after the timed region")
        # the boundary's host-buffer form (lv_render: the same frame + one D2H copy of the RGBA8 image, synchronous): never `value`
        host = None
        if direct:
            buf = np.empty((H, W, 4), dtype=np.uint8)
            for _ in range(2):
                ctx.render(mode=w["mode"], out=buf)
            nh = max(3, min(args.steps, 20))
            th = time.perf_counter()
            for _ in range(nh):
                ctx.render(mode=w["mode"], out=buf)
            hms = (time.perf_counter() - th) / nh * 1e3
            host = {"ms_per_step": round(hms, 4), "value": round(rays_per_frame / hms / 1e3, 2), "unit": "Mrays/s", "steps": nh,
                    "note": "lv_render into a pageable host buffer: frame + %.1f MB device-to-host copy per frame, host-synchronous "
                            "(no frames in flight); the scene upload happens once per data set, not per frame" % (W * H * 4 / 1e6)}
        del extra
        return dict(ctx=ctx, frame=frame, host=host, elapsed=elapsed, rays_per_frame=rays_per_frame, counters=counters, local=local,
                    kernel_bytes=kernel_bytes, frame_bytes=frame_bytes, kernels=kernels, frame_ms=_stats(frame_ms),
                    build_ms=build_ms, build_first_ms=build_first_ms, st=st, wkey=wkey, diag=diag)

    if not dry and world == 1 and device_geometry and wl.get("mesh") and not os.environ.get("LV_BENCH_SKIP_HOST_UPLOAD"):
        # for the record only: what the host-built arrays of the same scene cost to upload (the path of rounds 1-5), in a throw-away context
        c0, _ = make_context(wl, host_geometry=True)
        del c0
        gc.collect()
    head = measure(wl, args.workload)
    also = None
    if wl.get("also") and not dry:
        del head["ctx"]   # frees the triangle scene before the capsule context builds
        also = measure(WORKLOADS[wl["also"]], wl["also"])

    fast = None
    if wl.get("fast_also") and not dry and "shading_numerics" not in wl["settings"]:
        # the priced +-2 LSB contract: the same frame with shading_numerics = fast -- approximate hardware rsq / rcp / log2 /
        # exp2 in colour-only arithmetic; hits, coverage, fragment depths, alpha and list lengths unchanged -- reported beside `value`
        head.pop("ctx", None)
        gc.collect()
        fast = measure(dict(wl, settings=dict(wl["settings"], shading_numerics="fast")), args.workload)

    if rank == 0:
        kname = wl["kernel"]
        if head["kernels"] and kname not in head["kernels"] and wl.get("also_kernel") in head["kernels"]:
            kname = wl["also_kernel"]   # e.g. c4 with --set ppll_prism_rasteriser=lbvh: the all-hits walk is the front end
        ms_launch = head["kernels"].get(kname, {}).get("median", 0.0) if head["kernels"] else 0.0
        st = head["st"]
        result = {
            "metric": "Mrays/s", "value": round(head["rays_per_frame"] * args.steps / head["elapsed"] / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(head["elapsed"] / args.steps * 1e3, 4), "fps": round(args.steps / head["elapsed"], 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "frames_in_flight": frames_in_flight,
            "data": "synthetic" if not dry else "dry-run (no GPU, stand-in renderer)",
            "launch": os.environ.get("LV_BENCH_LAUNCH", "one process (N = 1)" if world == 1 else
                                     "external launcher: %d processes, RANK / WORLD_SIZE from the environment" % world),
            "config": {"workload": wl["name"], "resolution": [W, H], "segments": int(len(seg)) if seg is not None else 0,
                       "rays_per_frame": int(head["rays_per_frame"]), "ao_hit_pixels": int(head["counters"][4].item()),
                       "fragments_per_frame": int(head["counters"][5].item()),
                       "parallelism": ("one GPU: the whole frame in one lv_render_device call (no tiles to gather), 64x64-pixel groups "
                                       "dispatched heaviest-of-the-previous-frame first" if world == 1 and not dry else
                                       "screen tiles %dx%d in Morton order, dealt over %d GPU(s) by measured cost (RTAO hit pixels per "
                                       "tile of the instrumented frame; round robin for workloads without RTAO), one RCCL gather per "
                                       "frame" % (TILE, TILE, world)),
                       "accel_build_ms": round(head["build_ms"], 3), "accel_build_first_ms": round(head["build_first_ms"], 3),
                       "tri_accel_build_ms": round(float(st.ms_tri_accel_build), 3), "tri_accel_nodes": int(st.num_tri_nodes),
                       "data_preparation": prep,
                       "bvh_depth": int(st.bvh_depth), "tube_triangles": int(st.num_tube_triangles)},
            "frame_ms": head["frame_ms"],
            "kernels_ms": head["kernels"],
            "counters_rank0": head["local"],
            "roofline": roofline(kname, wl.get("pmc", args.workload), ms_launch, head["kernel_bytes"], world),
        }
        if head.get("diag"):
            result["multi_gpu"] = head["diag"]
        if head.get("host"):
            result["pcie_inclusive"] = head["host"]
        result["roofline"]["frame_algorithmic_bytes_rank0"] = int(head["frame_bytes"])
        if seg is not None:   # compulsory floor: every node, primitive record and line point once + the outputs
            tri_rec = 32 if getattr(st, "tri_leaf_bytes", 0) == 64 else 48   # per triangle: half a 64-B pair record, or a 48-B record
            result["roofline"]["frame_compulsory_bytes"] = int(st.num_nodes * 64 + len(seg) * 32 + len(pts) * 48 + W * H * (4 + 4)
                                                               + ((st.num_tube_triangles * (tri_rec + 32) + st.num_tri_nodes * 64)
                                                                  if wl.get("mesh") else 0))
            result["roofline"]["frame_compulsory_what"] = ("segment LBVH nodes + segment records + line points + outputs" +
                                                           (" + triangle LBVH nodes + leaf records (%d B per triangle) + 32-B vertices" % tri_rec
                                                            if wl.get("mesh") else "") + ", each once (until round 5 without the triangle LBVH's nodes)")
            if result["roofline"].get("traffic"):   # counter traffic of the dominant kernel against reading everything exactly once
                result["roofline"]["traffic_over_compulsory"] = round(result["roofline"]["traffic"]
                                                                      / result["roofline"]["frame_compulsory_bytes"], 3)
        if also is not None:
            sfx = wl.get("also_suffix", "capsules")
            result["value_" + sfx] = round(also["rays_per_frame"] * args.steps / also["elapsed"] / 1e6, 2)
            result["ms_per_step_" + sfx] = round(also["elapsed"] / args.steps * 1e3, 4)
            result["fps_" + sfx] = round(args.steps / also["elapsed"], 3)
            result[sfx] = {"workload": WORKLOADS[wl["also"]]["name"], "rays_per_frame": int(also["rays_per_frame"]),
                                  "frame_ms": also["frame_ms"], "kernels_ms": also["kernels"], "counters_rank0": also["local"],
                                  "roofline": roofline(wl.get("also_kernel", kname), wl["also"],
                                                       also["kernels"].get(wl.get("also_kernel", kname), {}).get("median", 0.0),
                                                       also["kernel_bytes"], world)}
        if fast is not None:
            result["value_fast_shading"] = round(fast["rays_per_frame"] * args.steps / fast["elapsed"] / 1e6, 2)
            result["ms_per_step_fast_shading"] = round(fast["elapsed"] / args.steps * 1e3, 4)
            result["fast_shading"] = {"setting": "shading_numerics=fast", "frame_ms": fast["frame_ms"], "kernels_ms": fast["kernels"],
                                      "note": "colours within +-2 LSB of the exact frame (measured: <= 1 LSB on < 30 of 2 073 600 pixels, "
                                              "profiles/deviations_r06.json); fragment lists (depth, alpha, length) bit-identical; default stays exact"}
        if world == 1 and not dry:
            ceil = measured_hbm_ceiling(device, torch)
            result["roofline"]["hbm_peak_measured_GBs"] = round(ceil, 1)   # copy bandwidth attainable on this box
        if args.save_frame and head["frame"] is not None and not dry:
            from PIL import Image
            Image.fromarray(head["frame"].cpu().numpy()).save(args.save_frame)
        if world == 1 and not args.no_cpu_baseline and not dry:
            cpu_wl = wl.get("cpu", {"c3": "c3t"}.get(args.workload, args.workload))   # the oracle leg this workload's frame corresponds to
            result["cpu_baseline"] = cpu_baseline(W, H, pts, seg, tf, attr_range, view, proj, fovy, near, far,
                                                  workload=cpu_wl, mesh=mesh, ao_spp=wl.get("ao_spp", 64),
                                                  capped=str(wl["settings"].get("use_capped_tubes", True)).lower() not in ("false", "0"))
        else:
            result["cpu_baseline"] = None
        if args.states and world == 1 and not dry:
            del head, also
            gc.collect()
            result["states"] = run_states(min(args.steps, 20), local_rank)
        emit_line(result)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
