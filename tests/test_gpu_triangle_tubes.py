"""Triangle-tube RTAO geometry (the reference traces AO against 6-gon triangle tubes): the HIP
path through the C-ABI against the CPU oracle.  Everything here is +,-,*,/,sqrt on float32 in a fixed order, so hits
(t, triangle, barycentrics) and AO factors are compared bit for bit; frames within +-2 LSB."""
import numpy as np
import pytest

from common import Case, small_case, max_lsb_diff
from linevis_amd import capi, host_api, scenes, transfer_function as tfm
from oracle import lvo

pytestmark = pytest.mark.gpu

RTAO_TRI = dict(ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0,
                rtao_geometry="triangle_tubes")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def curves(n_lines=30, pts_per_line=30, seed=7):
    return scenes.normalize(scenes.random_curves(n_lines=n_lines, points_per_line=pts_per_line, seed=seed))


def mesh_of(tr, line_width, subdiv=6):
    return lvo.build_tube_triangle_render_data(tr.positions, tr.attributes, tr.line_offsets, line_width, subdiv)


def tri_context(case, mesh):
    ctx = case.hip_context()
    ctx.set_tube_triangle_mesh(*mesh)
    return ctx


def random_rays(n, seed, extent=0.35):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-extent, extent, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


@pytest.mark.parametrize("subdiv,lw", [(6, 0.02), (8, 0.005), (4, 0.01)])
def test_triangle_rays_bit_exact_vs_brute_force(hip_lib, subdiv, lw):
    tr = curves()
    mesh = mesh_of(tr, lw, subdiv)
    case = small_case(line_width=lw)
    ctx = tri_context(case, mesh)
    ts = lvo.TriScene(*mesh, lw)
    o, d = random_rays(20000, 11)
    # axis-parallel directions exercise the 1/d = inf planes of the own-box rule
    d[:300] = 0.0
    d[:100, 0] = 1.0; d[100:200, 1] = -1.0; d[200:300, 2] = 1.0
    hits = 0
    for tmin, tmax in [(0.0, 0.1), (1e-4, 1000.0)]:
        a = ctx.trace_rays_triangles(o, d, tmin, tmax)
        b = ts.trace_rays(o, d, tmin, tmax, use_bvh=False)  # brute force = ground truth
        assert np.array_equal(a[1], b[1])
        assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(bits(a[2]), bits(b[2]))
        hits += int((a[1] != 0xFFFFFFFF).sum())
    assert hits > 500
    assert ctx.stats().num_tube_triangles == len(mesh[0])


def test_triangle_tie_goes_to_lowest_index(hip_lib):
    # two coincident triangles + a third behind them
    v = np.zeros(9, dtype=lvo.TUBE_VERTEX_DTYPE)
    tri = np.array([[0, 0, 0], [0.1, 0, 0], [0, 0.1, 0]], np.float32)
    v["vertexPosition"][0:3] = tri
    v["vertexPosition"][3:6] = tri
    v["vertexPosition"][6:9] = tri + np.array([0, 0, -0.05], np.float32)
    v["vertexNormal"][:] = [0, 0, 1]
    idx = np.array([[6, 7, 8], [3, 4, 5], [0, 1, 2]], np.uint32)
    pts = np.zeros(1, dtype=lvo.LINE_POINT_DTYPE)
    pts["lineTangent"] = [[1, 0, 0]]
    case = small_case()
    ctx = tri_context(case, (idx, v, pts))
    o = np.array([[0.02, 0.02, 0.5]], np.float32)
    d = np.array([[0.0, 0.0, -1.0]], np.float32)
    t, tri_id, uv = ctx.trace_rays_triangles(o, d, 0.0, 10.0)
    assert tri_id[0] == 1 and t[0] == np.float32(0.5)
    tb, trib, _ = lvo.TriScene(idx, v, pts, case.line_width).trace_rays(o, d, 0.0, 10.0)
    assert trib[0] == 1 and bits(tb)[0] == bits(t)[0]


@pytest.mark.parametrize("settings", [
    dict(ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=16),
    dict(ambient_occlusion_iterations=3, ambient_occlusion_samples_per_frame=4),
    dict(ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=8, ambient_occlusion_distance_based=False),
    dict(ambient_occlusion_iterations=2, ambient_occlusion_samples_per_frame=5, use_jittered_primary_rays=False,
         ambient_occlusion_radius=0.03),
])
def test_triangle_rtao_bit_exact_and_frame(hip_lib, settings):
    lw = 0.02
    tr = curves()
    mesh = mesh_of(tr, lw)
    case = small_case(line_width=lw, **RTAO_TRI, **settings)
    ctx = tri_context(case, mesh)
    img = ctx.render(capi.MODE_RAY_TRACER)
    ao = ctx.get_ao()
    sc = case.oracle_scene()
    P = case.oracle_params(sc)
    ao_ref = lvo.TriScene(*mesh, lw).render_ao(P, use_bvh=False)
    assert np.array_equal(bits(ao), bits(ao_ref))
    assert (ao_ref < 1.0).sum() > 200
    img_ref = sc.render_rt(P, ao=ao_ref, use_bvh=True)
    assert max_lsb_diff(img, img_ref) <= 2
    # the capsule geometry gives a different (but statistically close) AO image
    ctx.set_option("rtao_geometry", "capsules")
    ctx.render(capi.MODE_RAY_TRACER)
    ao_caps = ctx.get_ao()
    assert not np.array_equal(bits(ao_caps), bits(ao))
    both = (ao_caps < 1.0) & (ao < 1.0)
    if settings.get("ambient_occlusion_distance_based", True):
        assert abs(float(ao_caps[both].mean()) - float(ao[both].mean())) < 0.03


def test_triangle_rtao_needs_a_mesh_and_rejects_bad_input(hip_lib):
    case = small_case(**RTAO_TRI)
    ctx = case.hip_context()
    with pytest.raises(capi.LineVisError):
        ctx.render(capi.MODE_RAY_TRACER)
    idx, v, pts = mesh_of(curves(5, 10), 0.02)
    bad = idx.copy(); bad[3, 1] = len(v)
    with pytest.raises(capi.LineVisError):
        ctx.set_tube_triangle_mesh(bad, v, pts)
    with pytest.raises(capi.LineVisError):
        ctx.set_option("rtao_geometry", "spheres")
    # empty mesh: every primary ray misses -> AO factor 1 everywhere
    ctx.set_tube_triangle_mesh(np.zeros((0, 3), np.uint32), v[:0], pts[:0])
    ctx.render(capi.MODE_RAY_TRACER)
    assert np.all(ctx.get_ao() == 1.0)


def test_headless_renderer_uses_triangle_tubes(hip_lib):
    """The plugin classes fetch the mesh from LineData like VulkanRayTracedAmbientOcclusionPass::setLineData."""
    lw = 0.02
    tr = curves(20, 40, seed=3)
    flow = host_api.LineDataFlow()
    flow.set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    r = host_api.HeadlessLineRenderer(capi.MODE_RAY_TRACER)
    r.set_rendering_resolution(96, 64)
    r.set_transfer_function(tfm.standard())
    r.set_line_data(flow)
    settings = dict(line_width=lw, ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=8, **RTAO_TRI)
    r.set_new_settings(settings)
    img = r.render_frame()
    view, proj, fovy, near, far = r.camera()
    pts, seg, _ = lvo.build_tube_aabb_render_data(tr.positions, tr.attributes, tr.line_offsets, lw)
    case = Case(pts, seg, tfm.standard(), 96, 64, lw, **{k: v for k, v in settings.items() if k != "line_width"})
    case.view, case.proj, case.fovy, case.near, case.far = view, proj, fovy, near, far
    sc = case.oracle_scene()
    P = case.oracle_params(sc)
    P.attrMin, P.attrMax = flow.attribute_range()   # the plugin takes the transfer-function range from the data
    ao_ref = lvo.TriScene(*mesh_of(tr, lw), lw).render_ao(P, use_bvh=True)
    assert max_lsb_diff(img, sc.render_rt(P, ao=ao_ref, use_bvh=True)) <= 2
    assert r.stats().num_tube_triangles == len(mesh_of(tr, lw)[0])


def test_triangle_tubes_medium_scene_bvh_agreement(hip_lib):
    """100 k segments / 1.2 M triangles: HIP LBVH (quantised 4-wide) vs the oracle's binary CPU BVH vs brute force on a
    sample -- the own-box rule of the triangle test is what makes all three agree bit for bit."""
    lw = 0.002
    tr = scenes.normalize(scenes.helix_bundle())
    flow = host_api.LineDataFlow()
    flow.set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    mesh = flow.tube_triangle_render_data(lw, 6)
    assert len(mesh[0]) > 1200000
    pts, seg, _ = flow.tube_aabb_render_data(lw)
    case = Case(pts, seg, tfm.standard(), 256, 144, lw)
    ctx = tri_context(case, mesh)
    ts = lvo.TriScene(*mesh, lw)
    # rays from the camera through the scene + short AO-like rays starting near the geometry
    rng = np.random.default_rng(5)
    n = 60000
    o = np.tile(np.array([[0.0, 0.0, 0.8]], np.float32), (n, 1))
    tgt = rng.uniform(-0.3, 0.3, (n, 3)).astype(np.float32)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a = ctx.trace_rays_triangles(o, d, 1e-4, 1000.0)
    b = ts.trace_rays(o, d, 1e-4, 1000.0, use_bvh=True)
    assert np.array_equal(a[1], b[1]) and np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(bits(a[2]), bits(b[2]))
    hit = a[1] != 0xFFFFFFFF
    assert hit.sum() > 10000
    o2 = (o + d * a[0][:, None])[hit][:20000] + rng.normal(scale=0.003, size=(min(20000, int(hit.sum())), 3)).astype(np.float32)
    d2 = rng.normal(size=o2.shape).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    a2 = ctx.trace_rays_triangles(o2, d2, 0.0, 0.1)
    b2 = ts.trace_rays(o2, d2, 0.0, 0.1, use_bvh=True)
    assert np.array_equal(a2[1], b2[1]) and np.array_equal(bits(a2[0]), bits(b2[0]))
    sub = slice(0, 300)
    c2 = ts.trace_rays(o2[sub], d2[sub], 0.0, 0.1, use_bvh=False)
    assert np.array_equal(a2[1][sub], c2[1]) and np.array_equal(bits(a2[0][sub]), bits(c2[0]))


def test_config3_whole_frame_with_the_reference_rtao_geometry(hip_lib):
    """BASELINE.json config 3 with rtao_geometry = triangle_tubes (what the reference's RTAO pass traces): 1 M segments =
    12.06 M triangles, 1920 x 1080, 64 spp -- the AO factors of every pixel bit for bit, the frame within the bar."""
    lw = 0.002
    lvo.shade_normalize_out_of_range(reset=True)
    tr = scenes.normalize(scenes.tornado())
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    mesh = flow.tube_triangle_render_data(lw, 6)
    assert len(mesh[0]) > 12000000
    pts, seg, _ = flow.tube_aabb_render_data(lw)
    case = Case(pts, seg, tfm.standard(), 1920, 1080, lw, ambient_occlusion_iterations=1,
                ambient_occlusion_samples_per_frame=64, **RTAO_TRI)
    ctx = tri_context(case, mesh)
    img = ctx.render(capi.MODE_RAY_TRACER)
    ao = ctx.get_ao()
    sc = case.oracle_scene()
    P = case.oracle_params(sc)
    ts = lvo.TriScene(*mesh, lw)
    ao_ref = ts.render_ao(P, use_bvh=True)
    assert np.array_equal(bits(ao), bits(ao_ref)) and (ao_ref < 1.0).sum() > 300000
    ref = sc.render_rt(P, ao=ao_ref, use_bvh=True)
    assert max_lsb_diff(img, ref) <= 2
    # shading_numerics = fast on the headline frame: AO factors untouched, every pixel within 2 LSB of the EXACT oracle
    from test_gpu_parity import _fast_shading_deviation
    ctx.set_option("shading_numerics", "fast")
    fast = ctx.render(capi.MODE_RAY_TRACER)
    assert np.array_equal(bits(ctx.get_ao()), bits(ao_ref))
    _fast_shading_deviation("c3", img, fast, ref)
    assert lvo.shade_normalize_out_of_range() == 0      # the clamped normalize() rule never acted on this frame (tests/test_oracle.py)


def test_triangle_golden_fixture(hip_lib):
    """The committed triangle-tube fixture (tests/golden/triangle_tubes.npz): ray-triangle known answers through the
    traversal kernel (one triangle per scene would be slow: all KAT triangles form one mesh, rays are checked where the
    KAT triangle is also the scene's closest hit) and the AO image."""
    import os
    from common import GOLDEN_DIR
    g = np.load(os.path.join(GOLDEN_DIR, "triangle_tubes.npz"))
    W, H, lw = int(g["ao_width"]), int(g["ao_height"]), float(g["ao_line_width"])
    tr = curves(36, 40, seed=11)
    mesh = mesh_of(tr, lw)
    assert len(mesh[0]) == int(g["ao_num_triangles"])
    case = small_case(width=W, height=H, n_lines=36, pts_per_line=40, seed=11, line_width=lw, **RTAO_TRI,
                      ambient_occlusion_iterations=2, ambient_occlusion_samples_per_frame=8)
    ctx = tri_context(case, mesh)
    ctx.render(capi.MODE_RAY_TRACER)
    assert np.array_equal(bits(ctx.get_ao()), g["ao_bits"])
    # KAT triangles as one mesh (pad of the fixture = pad of a context with line width 0.002)
    n = len(g["kat_o"])
    v = np.zeros(3 * n, dtype=lvo.TUBE_VERTEX_DTYPE)
    v["vertexPosition"][0::3] = g["kat_v0"]; v["vertexPosition"][1::3] = g["kat_v1"]; v["vertexPosition"][2::3] = g["kat_v2"]
    idx = np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
    pts = np.zeros(1, dtype=lvo.LINE_POINT_DTYPE)
    c2 = small_case(line_width=0.002)
    ctx2 = tri_context(c2, (idx, v, pts))
    assert np.float32(np.float32(0.002) * np.float32(0.5)) * np.float32(1e-3) + np.float32(1e-6) == g["kat_pad"]
    t, tri, uv = ctx2.trace_rays_triangles(g["kat_o"], g["kat_d"], 0.0, 1000.0)
    own = tri == np.arange(n)
    assert own.sum() > 200                                   # most KAT rays hit their own triangle first
    assert np.all(g["kat_hit"][own] == 1)
    assert np.array_equal(bits(t)[own], g["kat_t_bits"][own]) and np.array_equal(bits(uv)[own], g["kat_uv_bits"][own])


# ---------------------------------------------------------------- ray tracer "Triangle Mesh" geometry mode
@pytest.mark.parametrize("settings,transparent", [
    (dict(geometry_mode="Triangle Mesh"), False),
    (dict(geometry_mode="Triangle Mesh", num_samples_per_frame=3, depth_cue_strength=0.7), True),
    (dict(use_analytic_intersections=False, use_halos=False, use_capped_tubes=False), True),
    (dict(geometry_mode="Triangle Mesh", ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=8, **RTAO_TRI), False),
])
def test_triangle_mesh_geometry_mode_frames(hip_lib, settings, transparent):
    """ClosestHitTubeTriangles + LineAttributesBarycentric (TubeRayTracing.glsl:301-352) through k_render_rt's triangle
    instantiation vs the oracle."""
    lw = 0.02
    tr = curves()
    mesh = mesh_of(tr, lw)
    case = small_case(line_width=lw, transparent=transparent, **settings)
    ctx = tri_context(case, mesh)
    img = ctx.render(capi.MODE_RAY_TRACER)
    sc = case.oracle_scene()
    P = case.oracle_params(sc)
    ts = lvo.TriScene(*mesh, lw)
    ao_ref = ts.render_ao(P, use_bvh=True) if P.useAmbientOcclusion else None
    ref = ts.render_rt(sc, P, ao=ao_ref, use_bvh=True)
    assert max_lsb_diff(img, ref) <= 2
    assert (ref[..., :3] != 255).any()
    # back to the analytic mode: a different picture (round vs faceted tubes), the capsule oracle's picture
    ctx.set_option("geometry_mode", "AABBs (analytic)")
    ctx.set_option("rtao_geometry", "capsules")
    case.settings.pop("rtao_geometry", None)
    case.oracle_params(sc)                # intersection_form "auto" is the closest-approach form again (Case.literal_form)
    img_c = ctx.render(capi.MODE_RAY_TRACER)
    ao_c = sc.render_ao(P, use_bvh=True) if P.useAmbientOcclusion else None
    assert max_lsb_diff(img_c, sc.render_rt(P, ao=ao_c, use_bvh=True)) <= 2
    assert not np.array_equal(img, img_c)


def test_triangle_mesh_mode_through_the_plugin(hip_lib):
    lw = 0.02
    tr = curves(20, 40, seed=3)
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    r = host_api.HeadlessLineRenderer(capi.MODE_RAY_TRACER)
    r.set_rendering_resolution(96, 64)
    r.set_transfer_function(tfm.standard())
    r.set_line_data(flow)
    r.set_new_settings(dict(line_width=lw, geometry_mode="Triangle Mesh"))
    img = r.render_frame()
    view, proj, fovy, near, far = r.camera()
    pts, seg, _ = lvo.build_tube_aabb_render_data(tr.positions, tr.attributes, tr.line_offsets, lw)
    case = Case(pts, seg, tfm.standard(), 96, 64, lw)
    case.view, case.proj, case.fovy, case.near, case.far = view, proj, fovy, near, far
    sc = case.oracle_scene()
    P = case.oracle_params(sc)
    P.attrMin, P.attrMax = flow.attribute_range()
    assert max_lsb_diff(img, lvo.TriScene(*mesh_of(tr, lw), lw).render_rt(sc, P, use_bvh=True)) <= 2
    # 8 subdivisions: LineData re-tessellates, the renderer re-uploads
    r.set_new_settings(dict(tube_num_subdivisions=8))
    img8 = r.render_frame()
    P.tubeNumSubdivisions = 8
    assert max_lsb_diff(img8, lvo.TriScene(*mesh_of(tr, lw, 8), lw).render_rt(sc, P, use_bvh=True)) <= 2
    assert not np.array_equal(img, img8)
    r.set_new_settings(dict(use_analytic_intersections=True))
    assert max_lsb_diff(r.render_frame(), sc.render_rt(P, use_bvh=True)) <= 2


@pytest.mark.parametrize("leaf_size", [1, 2, 3, 4])
def test_triangles_per_leaf_do_not_change_any_hit(hip_lib, leaf_size):
    """triangle_leaf_size: the leaves of the triangle LBVH hold 1 / 2 (default) / 4 consecutive triangles.  The closest hit, the
    AO factors and the transparency loop over the mesh must not depend on it -- including the case the first version got
    wrong: the nearest triangle of a leaf lies outside the ray interval and a farther one of the same leaf inside."""
    lw = 0.02
    tr = curves(n_lines=31, pts_per_line=29)      # 31 * (28 * 12 + caps) triangles: the last leaf is incomplete for 4
    mesh = mesh_of(tr, lw)
    case = small_case(line_width=lw, **RTAO_TRI, ambient_occlusion_iterations=2, ambient_occlusion_samples_per_frame=6)
    ctx = tri_context(case, mesh)
    ctx.set_option("triangle_leaf_size", leaf_size)
    ts = lvo.TriScene(*mesh, lw)
    o, d = random_rays(20000, 5)
    for tmin, tmax in [(0.0, 0.1), (0.02, 0.05), (1e-4, 1000.0)]:
        a = ctx.trace_rays_triangles(o, d, tmin, tmax)
        b = ts.trace_rays(o, d, tmin, tmax, use_bvh=False)
        assert np.array_equal(a[1], b[1]) and np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(bits(a[2]), bits(b[2]))
    ctx.render(capi.MODE_RAY_TRACER)
    P = case.oracle_params(case.oracle_scene())
    assert np.array_equal(bits(ctx.get_ao()), bits(ts.render_ao(P, use_bvh=False)))
    # Triangle Mesh geometry mode with the transparency loop: the same frame for every leaf size.  (MLAT over the mesh walks the
    # same leaves with lv_trace_all; its image depends on the visiting order by design and is pinned by replay, test_gpu_mlat.)
    c2 = small_case(line_width=lw, transparent=True, geometry_mode="Triangle Mesh")
    x = tri_context(c2, mesh)
    x.set_option("triangle_leaf_size", 1)
    want = x.render(capi.MODE_RAY_TRACER)
    x.set_option("triangle_leaf_size", leaf_size)
    assert np.array_equal(x.render(capi.MODE_RAY_TRACER), want)
    with pytest.raises(Exception):
        ctx.set_option("triangle_leaf_size", 9)


def _hits_equal_brute_force(ctx, ts, seed=9):
    o, d = random_rays(20000, seed)
    for tmin, tmax in [(0.0, 0.1), (0.02, 0.05), (1e-4, 1000.0)]:
        a = ctx.trace_rays_triangles(o, d, tmin, tmax)
        b = ts.trace_rays(o, d, tmin, tmax, use_bvh=False)
        assert np.array_equal(a[1], b[1]) and np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(bits(a[2]), bits(b[2]))
    assert (a[1] != 0xFFFFFFFF).sum() > 1000


@pytest.mark.parametrize("records,leaf_bytes", [("pairs", 64), ("triangles", 96)])
def test_pair_records_do_not_change_any_hit(hip_lib, records, leaf_bytes):
    """triangle_leaf_records: a leaf of two triangles stores its four vertices once (64 B) or as two 48-B records.  Closest hits
    (t, triangle, barycentrics) against brute force and every AO factor bit for bit in both forms -- body faces (code (q0, q2, q3)),
    cap quads and the pole fans (other codes) all lie in the rays' way."""
    lw = 0.02
    tr = curves(n_lines=31, pts_per_line=9)       # short lines: a third of the triangles belong to caps
    mesh = mesh_of(tr, lw)
    case = small_case(line_width=lw, **RTAO_TRI, ambient_occlusion_iterations=2, ambient_occlusion_samples_per_frame=6)
    ctx = tri_context(case, mesh)
    ctx.set_option("triangle_leaf_records", records)
    ts = lvo.TriScene(*mesh, lw)
    _hits_equal_brute_force(ctx, ts)
    assert ctx.stats().tri_leaf_bytes == leaf_bytes
    ctx.render(capi.MODE_RAY_TRACER)
    P = case.oracle_params(case.oracle_scene())
    assert np.array_equal(bits(ctx.get_ao()), bits(ts.render_ao(P, use_bvh=False)))
    with pytest.raises(Exception):
        ctx.set_option("triangle_leaf_records", "quads")


def test_pair_records_on_meshes_the_tessellator_does_not_write(hip_lib):
    """A caller's mesh (lv_set_tube_triangle_mesh takes any index buffer): (1) an odd triangle count leaves the last pair record
    half empty; (2) second triangles with rotated vertex order are still encodable -- the test must run on THEIR operand order, as
    brute force does; (3) a shuffled triangle order has leaves whose triangles share nothing: the build keeps 48-B records."""
    lw = 0.02
    tr = curves(n_lines=13, pts_per_line=17)
    idx, verts, pts = mesh_of(tr, lw)
    idx = np.ascontiguousarray(idx, dtype=np.uint32).reshape(-1, 3)
    case = small_case(line_width=lw)
    rng = np.random.default_rng(4)
    rotated = idx.copy()
    rotated[1::2] = np.roll(rotated[1::2], 1, axis=1)            # (a, c, d) -> (d, a, c)
    flip = rng.random(len(rotated) // 2) < 0.5
    rotated[1::2][flip] = np.roll(rotated[1::2][flip], 1, axis=1)
    for name, ind, want in [("odd", idx[:-1], 64), ("rotated", rotated, 64), ("shuffled", idx[rng.permutation(len(idx))], 96)]:
        ctx = tri_context(case, (ind, verts, pts))
        _hits_equal_brute_force(ctx, lvo.TriScene(ind, verts, pts, lw))
        assert ctx.stats().tri_leaf_bytes == want, name


def test_defaults_are_the_reference_geometry_and_roots(hip_lib):
    """Library defaults: rtao_geometry = "auto" traces the reference's triangle tubes as soon as the mesh of the current lines is there
    (capsules before, and again after new lines), intersection_form = "auto" is the reference's literal roots in every frame the
    reference can render -- the closest-approach form only where the RTAO rays hit the analytic capsules.  Each default frame is
    byte-identical to the frame with the resolved values spelled out."""
    lw = 0.004
    tr = curves(20, 40, seed=3)
    pts, seg, _ = lvo.build_tube_aabb_render_data(tr.positions, tr.attributes, tr.line_offsets, lw)
    mesh = mesh_of(tr, lw)
    rtao = dict(ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0, ambient_occlusion_iterations=1,
                ambient_occlusion_samples_per_frame=8)

    def frame(mesh_set, **settings):
        ctx = Case(pts, seg, tfm.standard(), 160, 120, lw, **settings).hip_context()
        if mesh_set:
            ctx.set_tube_triangle_mesh(*mesh)
        img = ctx.render(capi.MODE_RAY_TRACER)
        ao = ctx.get_ao().copy() if "ambient_occlusion_mode" in settings else None
        return ctx, img, ao

    # no AO: the reference's roots
    _, plain, _ = frame(False)
    assert np.array_equal(plain, frame(False, intersection_form="literal")[1])
    assert not np.array_equal(plain, frame(False, intersection_form="closest_approach")[1])
    # RTAO without the mesh: capsules, stable roots
    ctx, caps, ao_caps = frame(False, **rtao)
    _, want, ao_want = frame(False, rtao_geometry="capsules", intersection_form="closest_approach", **rtao)
    assert np.array_equal(caps, want) and np.array_equal(bits(ao_caps), bits(ao_want))
    ctx.set_option("rtao_geometry", "triangle_tubes")
    with pytest.raises(capi.LineVisError):
        ctx.render(capi.MODE_RAY_TRACER)
    # RTAO with the mesh: the reference's frame
    ctx, tri, ao_tri = frame(True, **rtao)
    _, want, ao_want = frame(True, rtao_geometry="triangle_tubes", intersection_form="literal", **rtao)
    assert np.array_equal(tri, want) and np.array_equal(bits(ao_tri), bits(ao_want))
    assert not np.array_equal(bits(ao_tri), bits(ao_caps))
    # new lines drop the mesh
    ctx.set_lines(pts, seg)
    assert np.array_equal(ctx.render(capi.MODE_RAY_TRACER), caps)
    # the plugin's RTAO baker traces the triangle tubes unless told otherwise
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    imgs = []
    for extra in (dict(), dict(rtao_geometry="triangle_tubes"), dict(rtao_geometry="capsules")):
        r = host_api.HeadlessLineRenderer(capi.MODE_RAY_TRACER)
        r.set_rendering_resolution(96, 64)
        r.set_transfer_function(tfm.standard())
        r.set_line_data(flow)
        r.set_new_settings(dict(line_width=lw, **rtao, **extra))
        imgs.append(r.render_frame())
    assert np.array_equal(imgs[0], imgs[1]) and not np.array_equal(imgs[0], imgs[2])
