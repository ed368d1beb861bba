"""Multi-layer alpha tracing on the GPU (k_render_rt_mlat, use_mlat / mlat_num_nodes) through the C-ABI.

The order in which a pixel's candidates reach insertNodeMlat is undefined in the reference (the driver's BVH traversal
order) and here (whichever lanes of the wave hold them); once layers are merged the frame depends on it.  Parity is
therefore checked by REPLAY: the kernel records the order it used, the oracle replays exactly that order through its
restatement of MlatInsert.glsl -- validating on the way that every candidate was a hit inside the ray interval valid at
that moment and that no visible layer was skipped -- and the frames must agree within the RGBA8 bar (the arithmetic is
the same float32 sequence; only pow() comes from different libms)."""
import numpy as np
import pytest

from common import Case, max_lsb_diff, small_case
from linevis_amd import capi, host_api, scenes, transfer_function as tfm
from oracle import lvo
from test_mlat import stick_case

pytestmark = pytest.mark.gpu

LSB_TOL = 2
TRACE = dict(use_mlat=True, collect_stats=True, mlat_record_trace=True)


def replay(c, ctx, k, ao=None):
    img = ctx.render(capi.MODE_RAY_TRACER)
    rec = ctx.mlat_trace()
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    ref, nodes, viol = sc.render_rt_mlat(P, k, ao=ao, trace=rec)
    return img, ref, viol, rec


@pytest.mark.parametrize("k", [1, 2, 4, 8, 16, 32])
def test_replay_parity_dense_transparent_scene(hip_lib, k):
    c = small_case(transparent=True, n_lines=60, mlat_num_nodes=k, **TRACE)
    ctx = c.hip_context()
    img, ref, viol, rec = replay(c, ctx, k)
    assert viol == 0
    assert max_lsb_diff(img, ref) <= LSB_TOL
    assert len(rec) > 5000 and (rec[:, 3] <= 1).all()
    # every pixel's sequence numbers are 0..n-1
    order = np.lexsort((rec[:, 1], rec[:, 0]))
    r = rec[order]
    first = np.r_[True, r[1:, 0] != r[:-1, 0]]
    assert (r[first, 1] == 0).all() and (np.diff(r[:, 1])[~first[1:]] == 1).all()
    # not far from the exact transparency of the per-pixel linked lists either
    exact = ctx2_render(c, capi.MODE_PPLL)
    err = np.abs(img.astype(np.int32) - exact.astype(np.int32)).mean()
    assert err < (4.0 if k < 4 else 2.5)


def ctx2_render(c, mode):
    s = {k: v for k, v in c.settings.items() if not k.startswith("mlat") and k not in ("use_mlat", "collect_stats")}
    c2 = Case(c.points, c.seg, c.tf, c.width, c.height, c.line_width, **s)
    ctx = c2.hip_context()
    ctx.set_option("ppll_fragment_colour", "ray_tracer")   # like for like: MLAT shades with the ray tracer's computeFragmentColor
    ctx.set_option("ppll_fragment_source", "capsule_entry")  # ... and traces the analytic capsules, not the rasterised prism
    return ctx.render(mode)


def test_replay_parity_with_rtao_and_depth_cues(hip_lib):
    c = small_case(transparent=True, n_lines=50, mlat_num_nodes=4, depth_cue_strength=0.8,
                   ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0,
                   ambient_occlusion_iterations=2, ambient_occlusion_samples_per_frame=4, **TRACE)
    ctx = c.hip_context()
    img = ctx.render(capi.MODE_RAY_TRACER)
    rec = ctx.mlat_trace()
    ao = ctx.get_ao()
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    assert np.array_equal(ao.view(np.uint32), sc.render_ao(P).view(np.uint32))
    ref, _, viol = sc.render_rt_mlat(P, 4, ao=ao, trace=rec)
    assert viol == 0 and max_lsb_diff(img, ref) <= LSB_TOL


def test_opaque_scene_accepts_and_shrinks_the_interval(hip_lib):
    """Opaque tubes: hits are accepted, the ray interval shrinks while the wave is still traversing, candidates that
    were tested before but arrive after are dropped at insertion time (flag 1) -- all of it validated by the replay."""
    c = small_case(transparent=False, n_lines=60, mlat_num_nodes=4, **TRACE)
    ctx = c.hip_context()
    img, ref, viol, rec = replay(c, ctx, 4)
    assert viol == 0 and max_lsb_diff(img, ref) <= LSB_TOL
    st = ctx.stats()
    # early termination pays: far fewer candidates shaded than the all-hits gather produces fragments
    c2 = Case(c.points, c.seg, c.tf, c.width, c.height, c.line_width, collect_stats=True, ppll_fragment_source="capsule_entry")
    ctx2 = c2.hip_context()
    ctx2.render(capi.MODE_PPLL)
    assert st.hits_shaded < ctx2.stats().hits_shaded
    # and the frame is the opaque ray tracer's except where two capsules share a joint sphere (equal depths)
    rt = ctx2.render(capi.MODE_RAY_TRACER)
    d = np.abs(img.astype(np.int32) - rt.astype(np.int32)).max(axis=2)
    assert (d > LSB_TOL).mean() < 0.05


def test_sticks_all_layers_fit(hip_lib):
    """No shared joints, 32 nodes >= layers: order-independent, so the GPU frame equals the oracle's canonical-order frame
    and the exact transparency loop."""
    c = stick_case(n=150, use_mlat=True, mlat_num_nodes=32)
    ctx = c.hip_context()
    img = ctx.render(capi.MODE_RAY_TRACER)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    ref, _, _ = sc.render_rt_mlat(P, 32)
    assert max_lsb_diff(img, ref) <= 1
    ctx.set_option("use_mlat", False)
    loop = ctx.render(capi.MODE_RAY_TRACER)
    d = np.abs(img.astype(np.int32) - loop.astype(np.int32)).max(axis=2)
    assert (d > 1).sum() <= 3


def test_jittered_samples_tiles_and_determinism(hip_lib):
    c = small_case(transparent=True, n_lines=60, use_mlat=True, mlat_num_nodes=4, num_samples_per_frame=4)
    ctx = c.hip_context()
    a = ctx.render(capi.MODE_RAY_TRACER)
    b = ctx.render(capi.MODE_RAY_TRACER)
    assert np.array_equal(a, b)                      # same waves, same order
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    ref, _, _ = sc.render_rt_mlat(P, 4)              # canonical order: close, not equal
    assert np.abs(a.astype(np.int32) - ref.astype(np.int32)).mean() < 1.0
    # a tile aligned to the 16x16 pixel blocks holds the same waves as the full frame
    t = ctx.render(capi.MODE_RAY_TRACER, tile=(32, 16, 48, 32))
    assert np.abs(t.astype(np.int32) - a[16:48, 32:80].astype(np.int32)).mean() < 0.5


def test_empty_and_background(hip_lib):
    pts = np.zeros(0, dtype=lvo.LINE_POINT_DTYPE)
    c = Case(pts, np.zeros((0, 2), np.uint32), tfm.standard(), 40, 24, 0.02, background=(0.2, 0.4, 0.6, 1.0),
             use_mlat=True, mlat_num_nodes=2)
    img = c.hip_context().render(capi.MODE_RAY_TRACER)
    assert (img == np.array([51, 102, 153, 255], np.uint8)).all()       # the miss shader's node, blended alone
    # transparent background colour: pre-multiplied by the miss node
    c = small_case(transparent=True, background=(1.0, 0.5, 0.0, 0.5), **TRACE, mlat_num_nodes=8)
    ctx = c.hip_context()
    img, ref, viol, _ = replay(c, ctx, 8)
    assert viol == 0 and max_lsb_diff(img, ref) <= LSB_TOL


def test_options_and_errors(hip_lib):
    c = small_case(transparent=True)
    ctx = c.hip_context()
    for bad in (0, 3, 64, "x"):
        with pytest.raises(capi.LineVisError):
            ctx.set_option("mlat_num_nodes", bad)
    ctx.set_option("use_mlat", True)
    ctx.render(capi.MODE_RAY_TRACER)
    with pytest.raises(capi.LineVisError):
        ctx.mlat_trace()                                   # not recorded
    ctx.set_options(dict(collect_stats=True, mlat_record_trace=True, mlat_trace_capacity=16))
    ctx.render(capi.MODE_RAY_TRACER)
    with pytest.raises(capi.LineVisError):
        ctx.mlat_trace()                                   # more records than the capacity
    ctx.set_option("geometry_mode", "Triangle Mesh")
    with pytest.raises(capi.LineVisError):
        ctx.render(capi.MODE_RAY_TRACER)                   # no tube mesh set
    # PPLL ignores the ray tracer's switch
    ctx.set_option("geometry_mode", "AABBs (analytic)")
    a = ctx.render(capi.MODE_PPLL)
    ctx.set_option("use_mlat", False)
    assert np.array_equal(a, ctx.render(capi.MODE_PPLL))


@pytest.mark.parametrize("k,transparent", [(2, True), (8, True), (4, False)])
def test_triangle_mesh_geometry_mode_replay_parity(hip_lib, k, transparent):
    """AnyHitTubeTriangles: the candidates are the triangles of the tube mesh (both faces of a tube are hit), shaded by
    the barycentric closest-hit path; same replay contract, trace ids are triangle indices."""
    lw = 0.02
    tr = scenes.normalize(scenes.random_curves(n_lines=30, points_per_line=30, seed=7))
    mesh = lvo.build_tube_triangle_render_data(tr.positions, tr.attributes, tr.line_offsets, lw, 6)
    c = small_case(line_width=lw, transparent=transparent, geometry_mode="Triangle Mesh", mlat_num_nodes=k, **TRACE)
    ctx = c.hip_context()
    ctx.set_tube_triangle_mesh(*mesh)
    img = ctx.render(capi.MODE_RAY_TRACER)
    rec = ctx.mlat_trace()
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    ts = lvo.TriScene(*mesh, lw)
    ref, _, viol = sc.render_rt_mlat(P, k, trace=rec, tri_scene=ts)
    assert viol == 0 and max_lsb_diff(img, ref) <= LSB_TOL
    assert len(rec) > 3000 and rec[:, 2].max() < len(mesh[0])
    # close to the exact transparency loop over the same mesh
    ctx.set_option("use_mlat", False)
    loop = ctx.render(capi.MODE_RAY_TRACER)
    assert np.abs(img.astype(np.int32) - loop.astype(np.int32)).mean() < (5.0 if k < 4 else 2.5)


def test_plugin_surface(hip_lib):
    tr = scenes.normalize(scenes.random_curves(n_lines=40, points_per_line=30, seed=7))
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    tf = tfm.standard_transparent()
    r = host_api.HeadlessLineRenderer(11)
    r.set_rendering_resolution(96, 64)
    r.set_transfer_function(tf)
    r.set_line_data(flow)
    r.set_new_settings(dict(line_width=0.02))
    loop = r.render_frame()
    r.set_new_settings(dict(use_mlat=True, mlat_num_nodes=2))
    mlat = r.render_frame()
    assert not np.array_equal(loop, mlat) and np.abs(loop.astype(np.int32) - mlat.astype(np.int32)).mean() < 4.0
    pts, seg, _ = flow.tube_aabb_render_data(0.02)
    lo, hi = flow.attribute_range()
    c = Case(pts, seg, tf, 96, 64, 0.02, use_mlat=True, mlat_num_nodes=2)
    ctx = c.hip_context()
    ctx.set_transfer_function(tf, lo, hi)
    view, proj, fovy, near, far = r.camera()
    ctx.set_camera(view, proj, fovy, near, far, 96, 64)
    assert np.array_equal(mlat, ctx.render(capi.MODE_RAY_TRACER))
