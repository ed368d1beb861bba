"""Several GPUs behind one C-ABI handle (lv_create_multi, linevis_amd/csrc/lv_multi.hip).

CPU tests: the deal (longest processing time first) and the Morton tile order as pure host functions of the library, against the
Python tiling module the earlier rounds used.  GPU tests (one MI355X): a 1-rank RCCL communicator (ncclSend / ncclRecv to itself) and
2-4 ranks on the same device over the memcpy transport reproduce the single-device frame byte for byte -- every part of the path
except the xGMI wire itself."""
import os

import numpy as np
import pytest

from common import small_case
from linevis_amd import capi, tiling

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tile_deal_matches_the_python_deal():
    rng = np.random.default_rng(3)
    for n, world in [(1, 1), (7, 3), (510, 8), (64, 5), (3, 8)]:
        costs = rng.uniform(0.0, 100.0, n)
        costs[rng.integers(0, n, max(1, n // 4))] = 7.0          # ties
        own = capi.tile_deal(costs, n, world)
        want = np.zeros(n, dtype=np.uint32)
        for r, ix in enumerate(tiling.assign_tiles_by_cost(costs, world)):
            want[ix] = r
        assert np.array_equal(own, want), (n, world)
        assert np.array_equal(capi.tile_deal(None, n, world), np.arange(n) % world)      # no costs: round robin
    # balance: 510 tiles with a heavy middle, 8 ranks -> within 2 % of the mean
    c = np.exp(-np.linspace(-3, 3, 510) ** 2) * 1000 + 16
    own = capi.tile_deal(c, 510, 8)
    load = np.array([c[own == r].sum() for r in range(8)])
    assert load.max() / load.mean() < 1.02


def test_make_tiles_matches_the_python_morton_order():
    for w, h, t in [(1920, 1080, 64), (100, 70, 32), (64, 64, 64), (65, 1, 64)]:
        a = capi.make_tiles(0, 0, w, h, t)
        assert np.array_equal(a, tiling.make_tiles(w, h, t))
    b = capi.make_tiles(37, 21, 50, 33, 16)
    assert np.array_equal(b - np.array([37, 21], np.uint32), tiling.make_tiles(50, 33, 16))


RTAO = dict(ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0, ambient_occlusion_iterations=2,
            ambient_occlusion_samples_per_frame=4, depth_cue_strength=0.6)


def _setup(ctx, c):
    ctx.set_lines(c.points, c.seg)
    ctx.set_transfer_function(c.tf, 0.0, 1.0)
    ctx.set_camera(c.view, c.proj, c.fovy, c.near, c.far, c.width, c.height)
    ctx.set_background(c.background)
    ctx.set_option("line_width", c.line_width)
    ctx.set_options(c.settings)
    return ctx


@pytest.mark.gpu
@pytest.mark.parametrize("devices,transport", [([0], "rccl"), ([0], "memcpy"), ([0, 0], "memcpy"), ([0, 0, 0, 0], "memcpy")])
def test_multi_handle_reproduces_the_single_device_frame(hip_lib, devices, transport):
    c = small_case(width=200, height=136, n_lines=40, pts_per_line=40, line_width=0.012, **RTAO)
    single = c.hip_context()
    want = single.render(11)
    multi = _setup(capi.Context(devices=devices, transport=transport), c)
    assert multi.num_ranks == len(devices)
    got = multi.render(11)
    assert np.array_equal(got, want)
    own = multi.deal()
    assert len(own) == 4 * 3 and set(own.tolist()) == set(range(len(devices)))     # 64 x 64 tiles of 200 x 136, dealt round robin
    # counters are summed over the ranks
    multi.set_option("collect_stats", True); single.set_option("collect_stats", True)
    multi.render(11); single.render(11)
    sm, ss = multi.stats(), single.stats()
    assert sm.rays_traced == ss.rays_traced and sm.ao_rays_traced == ss.ao_rays_traced and sm.ao_hit_pixels == ss.ao_hit_pixels
    # ... and every rank's own share is readable (lv_multi_rank_stats): the shares add up, every rank rendered something
    shares = [multi.rank_stats(r) for r in range(len(devices))]
    assert sum(s.rays_traced for s in shares) == sm.rays_traced and all(s.rays_traced > 0 and s.ms_total > 0.0 for s in shares)
    with pytest.raises(capi.LineVisError):
        multi.rank_stats(len(devices))
    multi.set_option("collect_stats", False)
    # re-dealt by measured cost: another deal, the same bytes
    multi.rebalance()
    assert np.array_equal(multi.render(11), want)
    if len(devices) > 1:
        own2 = multi.deal()
        assert not np.array_equal(own2, own) and set(own2.tolist()) == set(range(len(devices)))
    # a sub-rectangle and the caller's own tile list
    assert np.array_equal(multi.render(11, tile=(37, 21, 90, 70)), want[21:91, 37:127])
    import torch
    tiles = tiling.make_tiles(200, 136, 32)
    out = torch.zeros((len(tiles), 32, 32, 4), dtype=torch.uint8, device="cuda")
    multi.render_tiles_device(out.data_ptr(), tiles, 32, 32, mode=11)
    multi.stats()                                   # synchronises the handle's stream
    torch.cuda.synchronize()
    assert np.array_equal(tiling.detile(out.cpu().numpy(), tiles, 200, 136, 32), want)
    # PPLL and MLAT through the same handle; consecutive frames reuse the buffers
    ppll = c.hip_context().render(2)
    for _ in range(3):
        assert np.array_equal(multi.render(2), ppll)
    multi.close()


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0, 0]])
def test_multi_handle_with_the_reference_rtao_geometry(hip_lib, devices):
    """the mode bench.py times as the headline (rtao_geometry = triangle_tubes, literal roots in the colour pass) through
    lv_create_multi over the memcpy transport: the single-device frame byte for byte, also after a cost-weighted re-deal"""
    from test_gpu_triangle_tubes import mesh_of
    from linevis_amd import scenes
    tr = scenes.normalize(scenes.random_curves(n_lines=40, points_per_line=40, seed=7))
    lw = 0.012
    from common import scene_arrays, Case
    from linevis_amd import transfer_function as tfm
    pts, seg = scene_arrays(tr, lw)
    c = Case(pts, seg, tfm.standard(), 200, 136, lw, rtao_geometry="triangle_tubes", **RTAO)
    mesh = mesh_of(tr, lw)
    single = c.hip_context()
    single.set_tube_triangle_mesh(*mesh)
    want = single.render(11)
    multi = _setup(capi.Context(devices=devices, transport="memcpy"), c)
    multi.set_tube_triangle_mesh(*mesh)
    assert np.array_equal(multi.render(11), want)
    multi.rebalance()
    assert np.array_equal(multi.render(11), want)
    assert np.array_equal(multi.render(11, tile=(37, 21, 90, 70)), want[21:91, 37:127])
    assert multi.stats().num_tube_triangles == len(mesh[0])
    multi.close()


@pytest.mark.gpu
def test_multi_handle_with_a_bad_device_ordinal_fails_cleanly(hip_lib):
    """lv_create fails for rank 1 -> lv_create_multi must return an error, not walk unsized per-rank arrays"""
    with pytest.raises(capi.LineVisError):
        capi.Context(devices=[0, 999], transport="memcpy")
    with pytest.raises(capi.LineVisError):
        capi.Context(devices=[999], transport="memcpy")


@pytest.mark.gpu
def test_idle_ranks_keep_their_temporal_state_in_step(hip_lib):
    """a rank that owns no tile of a frame (fewer tiles than ranks) must still advance its SVGF / progressive state:
    4 ranks render ONE tile, then the full frame with SVGF -> the single-device sequence byte for byte."""
    settings = dict(RTAO, ambient_occlusion_denoiser="SVGF", ambient_occlusion_iterations=1)
    c = small_case(width=200, height=136, n_lines=40, pts_per_line=40, line_width=0.012, **settings)
    single = c.hip_context()
    multi = _setup(capi.Context(devices=[0, 0, 0, 0], transport="memcpy"), c)
    for tile in ((64, 64, 64, 64), None, (0, 0, 64, 64), None, None):
        a = single.render(11, tile=tile) if tile else single.render(11)
        b = multi.render(11, tile=tile) if tile else multi.render(11)
        assert np.array_equal(a, b), tile
    multi.close()


@pytest.mark.gpu
def test_multi_handle_forwards_setters_and_reports_rank_errors(hip_lib):
    c = small_case(width=96, height=64)
    multi = _setup(capi.Context(devices=[0, 0, 0], transport="memcpy"), c)
    a = multi.render(11)
    multi.set_option("line_width", 0.05)            # every rank rebuilds its LBVH
    b = multi.render(11)
    single = c.hip_context()
    single.set_option("line_width", 0.05)
    assert not np.array_equal(a, b) and np.array_equal(b, single.render(11))
    with pytest.raises(capi.LineVisError):
        multi.set_option("no_such_option", 1)
    with pytest.raises(capi.LineVisError):
        capi.Context(devices=[0, 0], transport="rccl")          # RCCL needs distinct devices
    multi.close()


def test_canned_benchmark_states_table():
    """lv::getTestModes: the hot-path subset of the reference's --perf states (InternalState.cpp:46-51,276-297), every state twice."""
    from linevis_amd import host_api
    m = host_api.get_test_modes()
    assert [s[0] for s in m] == ["PPLL", "PPLL(2)", "VRT Analytic", "VRT Analytic(2)", "VRT Triangle Mesh", "VRT Triangle Mesh(2)"]
    assert [s[1] for s in m] == [2, 2, 11, 11, 11, 11] and all(s[2] == (1920, 1080) for s in m)
    assert m[0][3] == {} and m[4][3] == {"useAnalyticIntersections": "false", "numSamplesPerFrame": "1"}
    assert len(host_api.get_test_modes(twice=False)) == 3


@pytest.mark.gpu
def test_plugin_walks_the_canned_states_like_the_perf_harness(hip_lib):
    """AutomaticPerformanceMeasurer's loop on the headless harness: setNewState per canned state (mode switch PPLL <-> ray tracer,
    camelCase renderer keys, tiling mode), each frame equal to a context driven through the C-ABI directly."""
    from linevis_amd import host_api, scenes, transfer_function as tfm
    from oracle import lvo
    tr = scenes.normalize(scenes.random_curves(n_lines=25, points_per_line=30, seed=5))
    lw = 0.02
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    r = host_api.HeadlessLineRenderer(capi.MODE_RAY_TRACER)
    r.set_rendering_resolution(64, 48)
    r.set_transfer_function(tfm.standard_transparent())
    r.set_line_data(flow)
    r.set_new_settings(dict(line_width=lw))
    pts, seg, _ = flow.tube_aabb_render_data(lw)
    mesh = flow.tube_triangle_render_data(lw, 6)
    lo, hi = flow.attribute_range()
    frames = {}
    for name, mode, res, settings in host_api.get_test_modes():
        r.set_new_state(name, mode, settings, resolution=(160, 96))
        assert r.rendering_mode == mode
        img = r.render_frame()
        assert img.shape == (96, 160, 4)
        frames[name] = img
        view, proj, fovy, near, far = r.camera()
        ctx = capi.Context(0)
        ctx.set_lines(pts, seg)
        ctx.set_transfer_function(tfm.standard_transparent(), lo, hi)
        ctx.set_camera(view, proj, fovy, near, far, 160, 96)
        ctx.set_option("line_width", lw)
        if mode == 11:
            ctx.set_option("num_samples_per_frame", int(settings["numSamplesPerFrame"]))
            ctx.set_option("use_analytic_intersections", settings["useAnalyticIntersections"])
            ctx.set_tube_triangle_mesh(*mesh)
        else:
            ctx.set_option("use_capped_tubes", False)      # rasterisers: no caps in the programmable-pull mode (LineData.cpp:1240-1244)
        assert np.array_equal(ctx.render(mode), img), name
    assert np.array_equal(frames["PPLL"], frames["PPLL(2)"]) and np.array_equal(frames["VRT Analytic"], frames["VRT Analytic(2)"])
    assert not np.array_equal(frames["VRT Analytic"], frames["VRT Triangle Mesh"])
    # tiling mode of the state reaches the per-pixel lists (same picture, other addressing)
    r.set_new_state("PPLL 1x1", 2, {}, tiling=(1, 1))
    assert np.array_equal(r.render_frame(), frames["PPLL"])


@pytest.mark.gpu
@pytest.mark.parametrize("devices,transport", [([0], "rccl"), ([0, 0, 0], "memcpy")])
def test_plugin_on_several_devices(hip_lib, devices, transport):
    """SceneData::deviceOrdinals: the same plugin classes drive one context per device; frame = the single-device frame."""
    from linevis_amd import host_api, scenes, transfer_function as tfm
    tr = scenes.normalize(scenes.random_curves(n_lines=25, points_per_line=30, seed=5))
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    settings = dict(line_width=0.02, ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0,
                    ambient_occlusion_iterations=2, ambient_occlusion_samples_per_frame=4)
    imgs = []
    for kw in (dict(device=0), dict(devices=devices, transport=transport)):
        r = host_api.HeadlessLineRenderer(capi.MODE_RAY_TRACER, **kw)
        r.set_rendering_resolution(200, 136)
        r.set_transfer_function(tfm.standard())
        r.set_line_data(flow)
        r.set_new_settings(settings)
        imgs.append(r.render_frame())
        if "devices" in kw:
            assert r.num_devices == len(devices)
            r.rebalance()
            assert np.array_equal(r.render_frame(), imgs[0])
    assert np.array_equal(imgs[0], imgs[1]) and (imgs[0][..., :3] != 255).any()


def _bench(args, env_extra=None, timeout=900):
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line on stdout:\n" + r.stdout[-2000:]
    assert r.stdout.strip().splitlines()[-1] == lines[0], "the JSON line must be the LAST line of stdout:\n" + r.stdout[-1500:]
    return json.loads(lines[0]), r.stderr


@pytest.mark.gpu
def test_bench_one_process_rccl_end_to_end(hip_lib):
    """`bench.py --gpus 1 --one-process --transport rccl` through a subprocess: the library's multi-device handle with its RCCL
    communicator (one rank here: everything of that path one MI355X can run), the bench line parsed."""
    j, _ = _bench(["--gpus", "1", "--one-process", "--transport", "rccl", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["value"] > 100.0 and j["unit"] == "Mrays/s"
    mg = j["multi_gpu"]
    assert mg["ranks_observed"] == 1 and mg["transport"] == "rccl" and mg["tiles_per_rank"] == [30 * 17]
    assert j["config"]["rays_per_frame"] > 2_000_000 and "one-process" in j["launch"]


@pytest.mark.gpu
def test_bench_gpus2_without_torchrun_walks_the_launch_chain(hip_lib):
    """`python bench.py --gpus 2` (the driver's command form, no torchrun) on a box whose two ranks share ONE GPU
    (LV_BENCH_DEVICES=0,0): bench.py launches torch.distributed.run itself; RCCL refuses two ranks on one device, so the chain falls
    through to the one-process handle (RCCL, then peer memcpy) -- whichever path completes, there is exactly one valid line, it
    names the path that ran, and every failed attempt is recorded with its stderr."""
    j, err = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--workload", "c2"],
                    {"LV_BENCH_DEVICES": "0,0", "LV_BENCH_LAUNCH_TIMEOUT": "240"}, timeout=1200)
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["value"] > 0.0
    assert isinstance(j["launch_attempts"], list) and j["launch"]
    for a in j["launch_attempts"]:
        assert a["path"] and a["stderr_tail"]
    mg = j["multi_gpu"]
    assert mg["ranks_observed"] == 2 and len(mg["tiles_per_rank"]) == 2 and sum(mg["tiles_per_rank"]) == 30 * 17
    if j["launch_attempts"]:
        assert j["launch"].startswith("fallback") and "[bench]" in err


@pytest.mark.gpu
@pytest.mark.parametrize("frames_in_flight", ["1", "2"])
def test_bench_distributed_code_path_on_a_one_rank_group(hip_lib, frames_in_flight):
    """Everything of a `--gpus N` run that one GPU can execute, through bench.py itself (LV_BENCH_FORCE_DIST=1): torch.distributed's nccl
    (= RCCL) process group with one rank, the tile list + cost re-deal, frames in flight, de-tiling, the per-rank diagnostics and the
    N = 1 value of the same run -- whose whole-frame image must be identical to the frame assembled from the tiles."""
    j, _ = _bench(["--gpus", "1", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--workload", "c3c"],
                  {"LV_BENCH_FORCE_DIST": "1", "LV_FRAMES_IN_FLIGHT": frames_in_flight})
    assert j["n_gpus"] == 1 and j["steps"] == 4 and j["frames_in_flight"] == int(frames_in_flight)
    mg = j["multi_gpu"]
    assert mg["ranks_observed"] == 1 and mg["backend"] == "nccl" and mg["tiles_per_rank"] == [30 * 17]
    assert len(mg["render_ms_per_rank"]) == 1 and mg["render_ms_per_rank"][0] > 0.5
    one = mg["single_gpu_same_run"]
    assert one["frame_identical_to_sharded"] is True
    assert one["ms_per_step"] > 1.0 and 0.5 < one["speedup"] < 1.5 and one["value"] > 1000.0
    assert j["config"]["rays_per_frame"] > 20_000_000
