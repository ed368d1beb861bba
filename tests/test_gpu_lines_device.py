"""lv_set_trajectories (SURVEY.md §8 a2 + a14 on the device, round 6): the trajectories go to HBM and kernels write the 48-byte line
points + index pairs of LineDataFlow::getLinePassTubeAabbRenderData (LineDataFlow.cpp:2112-2277) and the capped triangle tubes of
createCappedTriangleTubesRenderDataCPU (CappedTriangleTubesCPU.cpp:214-383) -- byte for byte what the host layer, the oracle and the
committed fixtures hold; frames rendered from the device-built geometry are the frames rendered from the uploaded one."""
import os

import numpy as np
import pytest

from common import GOLDEN_DIR, Case, max_lsb_diff
from linevis_amd import capi, host_api, scenes, transfer_function as tfm
from oracle import lvo

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(GOLDEN_DIR, "triangle_tubes.npz"))
A2 = np.load(os.path.join(GOLDEN_DIR, "a2_cases.npz"))
RTAO_TRI = dict(ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0, rtao_geometry="triangle_tubes")


def device_geometry(pos, att, off, lw, n):
    ctx = capi.Context(0)
    ctx.set_option("line_width", lw)
    ctx.set_option("tube_num_subdivisions", n)
    ctx.set_trajectories(pos, att, off)
    return ctx, ctx.get_lines(), ctx.get_tube_triangle_mesh()


def same_bytes(a, b):
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def test_device_a2_and_tessellation_match_the_golden_corner_cases(hip_lib):
    for n in (6, 4, 9):
        ctx, (pts, seg), (idx, verts, tpts) = device_geometry(A2["positions"], A2["attributes"], A2["line_offsets"], float(A2["line_width"]), n)
        assert np.array_equal(pts.view(np.uint8).reshape(-1, 48), A2["points"]) and np.array_equal(seg, A2["seg"])
        assert np.array_equal(idx, G["a2_idx_n%d" % n])
        assert np.array_equal(verts.view(np.uint8).reshape(-1, 32), G["a2_verts_n%d" % n])
        assert np.array_equal(tpts.view(np.uint8).reshape(-1, 48), G["a2_points_n%d" % n])
        st = ctx.stats()
        assert st.ms_line_points > 0.0 and st.ms_tessellate > 0.0


def test_device_tessellation_keeps_the_quirks_of_degenerate_lines(hip_lib):
    """CappedTriangleTubesCPU.cpp:253-262,307-316: a line with one valid point keeps the start cap's zero index range, one without
    any keeps zero vertices and zero indices; lines of fewer than two points vanish; empty input gives empty output."""
    pos = np.array([[0, 0, 0], [0, 0, 0], [0.00001, 0, 0],                     # 3 points, none valid
                    [0.5, 0, 0],                                               # single point: skipped entirely
                    [0, 0.2, 0], [0, 0.2, 0.00006], [0, 0.2, 0.00012],         # only the middle point is valid
                    [0.1, 0.1, 0.1], [0.2, 0.1, 0.1], [0.3, 0.15, 0.1]], np.float32)   # a regular line behind them
    off = np.array([0, 3, 4, 4, 7, 10], np.uint32)                            # (with an empty line in between)
    att = np.linspace(0, 1, len(pos)).astype(np.float32)
    for n in (6, 8):
        _, (pts, seg), mesh = device_geometry(pos, att, off, 0.02, n)
        ref = lvo.build_tube_triangle_render_data(pos, att, off, 0.02, n)
        assert all(same_bytes(np.asarray(a), np.asarray(b)) for a, b in zip(mesh, ref))
        p2, s2, _ = lvo.build_tube_aabb_render_data(pos, att, off, 0.02)
        assert same_bytes(pts, p2) and np.array_equal(seg, s2) and len(seg) == 2
    _, (pts, seg), mesh = device_geometry(pos[:0], att[:0], np.array([0], np.uint32), 0.02, 6)
    assert len(pts) == 0 and len(seg) == 0 and all(len(a) == 0 for a in mesh)
    _, (pts, seg), mesh = device_geometry(pos[:3], None, np.array([0, 3], np.uint32), 0.02, 6)   # no attribute, nothing valid
    assert len(pts) == 0 and len(mesh[1]) == 13 and mesh[0].size == 90 and not mesh[0].any()


@pytest.mark.parametrize("seed,n,lw,lines,ppl", [(1, 6, 0.02, 25, 35), (2, 8, 0.004, 40, 7), (3, 4, 0.01, 300, 3), (4, 5, 0.03, 3, 6000),
                                               (5, 16, 0.002, 64, 130), (6, 6, 0.002, 1500, 2)])
def test_device_geometry_is_byte_identical_to_host_and_oracle(hip_lib, seed, n, lw, lines, ppl):
    """Random curves with duplicated points (zero tangents, also across line starts), short lines, a line longer than the LDS chunk of
    the normal recurrence: a2 and a14 from the device = the C++ host layer = the oracle's literal restatement."""
    tr = scenes.normalize(scenes.random_curves(n_lines=lines, points_per_line=ppl, seed=seed))
    pos = tr.positions.copy()
    rng = np.random.default_rng(seed)
    dup = rng.integers(1, len(pos), max(4, len(pos) // 40))
    pos[dup] = pos[dup - 1]
    if len(pos) > 80:
        pos[70:73] = pos[70]
    flow = host_api.LineDataFlow().set_trajectories(pos, tr.attributes, tr.line_offsets)
    _, (pts, seg), mesh = device_geometry(pos, tr.attributes[0] if np.ndim(tr.attributes) == 2 else tr.attributes, tr.line_offsets, lw, n)
    hp, hs, _ = flow.tube_aabb_render_data(lw)
    assert same_bytes(pts, hp) and np.array_equal(seg, hs)
    hm = flow.tube_triangle_render_data(lw, n)
    assert np.array_equal(mesh[0], hm[0]) and same_bytes(mesh[1], hm[1]) and same_bytes(mesh[2], hm[2])
    om = lvo.build_tube_triangle_render_data(pos, tr.attributes, tr.line_offsets, lw, n)
    assert np.array_equal(mesh[0], om[0]) and same_bytes(mesh[1], om[1]) and same_bytes(mesh[2], om[2])


def test_line_width_change_retessellates_on_the_device_and_frames_match_the_uploaded_geometry(hip_lib):
    """A frame from lv_set_trajectories = the frame from lv_set_lines + lv_set_tube_triangle_mesh, before and after a line-width
    change (which re-tessellates and rebuilds both LBVHs on the device); AO factors bit for bit."""
    tr = scenes.normalize(scenes.random_curves(n_lines=30, points_per_line=30, seed=7))
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    att = tr.attributes[0] if np.ndim(tr.attributes) == 2 else tr.attributes
    dev = None
    for lw in (0.02, 0.008):
        pts, seg, _ = flow.tube_aabb_render_data(lw)
        c = Case(pts, seg, tfm.standard(), 192, 128, lw, ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=16, **RTAO_TRI)
        up = c.hip_context()
        up.set_tube_triangle_mesh(*flow.tube_triangle_render_data(lw, 6))
        want, want_ao = up.render(11), up.get_ao()
        if dev is None:
            dev = c.hip_context()
            dev.set_trajectories(tr.positions, att, tr.line_offsets)
            dev.set_option("rtao_geometry", "auto")          # auto = the reference's triangle tubes: the trajectories can be tessellated
        else:
            dev.set_option("line_width", lw)                 # nothing else: the mesh follows
        got, got_ao = dev.render(11), dev.get_ao()
        assert np.array_equal(got, want) and np.array_equal(got_ao.view(np.uint32), want_ao.view(np.uint32))
        assert (want_ao < 1.0).sum() > 500
        assert dev.stats().num_tube_triangles == up.stats().num_tube_triangles
    # a mesh passed by the caller overrides the device tessellation until the next lv_set_trajectories
    dev.set_tube_triangle_mesh(*flow.tube_triangle_render_data(0.02, 6))
    dev.set_option("line_width", 0.02)
    assert dev.stats().num_tube_triangles == up.stats().num_tube_triangles
    # band data cannot be tessellated here: a loud error, not a wrong mesh
    dev.set_trajectories(tr.positions, att, tr.line_offsets)
    dev.set_option("use_ribbons", True)
    with pytest.raises(capi.LineVisError):
        dev.get_tube_triangle_mesh()


def test_config3_geometry_on_the_device(hip_lib):
    """The headline scene: 1 M segments -> 1 001 000 line points + 12.06 M triangles written on the device, byte-identical to the host
    layer's arrays (CRC of the buffers); a line-width change incl. tessellation and both LBVH builds well under 50 ms."""
    import time
    import zlib
    tr = scenes.normalize(scenes.tornado())
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    att = tr.attributes[0] if np.ndim(tr.attributes) == 2 else tr.attributes
    ctx = capi.Context(0)
    ctx.set_option("line_width", 0.002)
    ctx.set_trajectories(tr.positions, att, tr.line_offsets)
    pts, seg = ctx.get_lines()
    hp, hs, _ = flow.tube_aabb_render_data(0.002)
    assert len(seg) == 1000000 and same_bytes(pts, hp) and np.array_equal(seg, hs)
    for lw in (0.002, 0.003):
        ctx.set_option("line_width", lw)
        mesh = ctx.get_tube_triangle_mesh()
        hm = flow.tube_triangle_render_data(lw, 6)
        assert len(mesh[0]) == 12060000
        for a, b in zip(mesh, hm):
            assert zlib.crc32(np.ascontiguousarray(a).view(np.uint8)) == zlib.crc32(np.ascontiguousarray(b).view(np.uint8))
    ctx.build_accel()
    ctx.set_option("line_width", 0.0025)
    t0 = time.perf_counter()
    ctx.build_accel()                                        # tessellation + segment LBVH + triangle LBVH, host-synchronous
    wall_ms = (time.perf_counter() - t0) * 1e3
    st = ctx.stats()
    assert st.num_tube_triangles == 12060000 and st.ms_tessellate > 0.0 and st.ms_tri_accel_build > 0.0
    assert st.ms_tessellate + st.ms_tri_accel_build + st.ms_accel_build < 50.0 and wall_ms < 50.0, (wall_ms, st.ms_tessellate)


@pytest.mark.parametrize("mode", [capi.MODE_RAY_TRACER, capi.MODE_PPLL])
def test_plugin_surface_uses_the_device_geometry_and_matches_the_host_built_one(hip_lib, mode):
    """lv::LineRenderer::uploadFrameState hands plain flow lines to lv_set_trajectories (use_device_geometry, default on): frames are
    byte-identical to the ones from the host-built render data (use_device_geometry = false), also after a line-width change (no new
    upload: the device re-tessellates) and after new trajectories."""
    def renderer(device_geometry):
        r = host_api.HeadlessLineRenderer(mode)
        r.set_rendering_resolution(160, 96)
        r.set_transfer_function(tfm.standard_transparent() if mode == capi.MODE_PPLL else tfm.standard())
        r.set_new_settings(dict(use_device_geometry=device_geometry))
        return r
    tr = scenes.normalize(scenes.random_curves(n_lines=20, points_per_line=40, seed=3))
    tr2 = scenes.normalize(scenes.random_curves(n_lines=12, points_per_line=25, seed=11))
    settings = dict(line_width=0.02, ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=8, **RTAO_TRI)
    frames = {}
    for dg in (True, False):
        r = renderer(dg)
        flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
        r.set_line_data(flow)
        r.set_new_settings(settings)
        out = [r.render_frame().copy()]
        tris = r.stats().num_tube_triangles
        r.set_new_settings(dict(line_width=0.011))
        out.append(r.render_frame().copy())
        flow2 = host_api.LineDataFlow().set_trajectories(tr2.positions, tr2.attributes, tr2.line_offsets)
        r.set_line_data(flow2)
        out.append(r.render_frame().copy())
        frames[dg] = out
        if mode == capi.MODE_RAY_TRACER:
            assert tris == len(lvo.build_tube_triangle_render_data(tr.positions, tr.attributes, tr.line_offsets, 0.02, 6)[0])
            assert (r.stats().ms_tessellate > 0.0) == dg      # the device tessellator ran only on the device-geometry path
    for a, b in zip(frames[True], frames[False]):
        assert np.array_equal(a, b) and (a[..., :3] != 255).any(axis=2).sum() > 300
    assert not np.array_equal(frames[True][0], frames[True][1])
