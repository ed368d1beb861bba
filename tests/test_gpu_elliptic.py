"""Band data in the ray tracer: the "Elliptic Tubes" mode (EllipticTubeRayTracing.glsl -- sphere-traced
twisted elliptic tubelets) and the USE_BANDS shading of computeFragmentColor (RayHitCommon.glsl), HIP against the oracle."""
import numpy as np
import pytest

from common import Case, max_lsb_diff
from linevis_amd import scenes, transfer_function as tfm
from oracle import lvo

BANDS = dict(use_ribbons=True, band_width=0.05, min_band_thickness=0.3)
RTAO = dict(ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0, ambient_occlusion_gamma=1.0,
            ambient_occlusion_radius=0.15, ambient_occlusion_distance_based=True, ambient_occlusion_iterations=2,
            ambient_occlusion_samples_per_frame=4)


def ribbon_scene(n_lines=5, pts=120, twist=8.0, seed=3):
    return scenes.twisted_ribbons(scenes.normalize(scenes.helix_bundle(n_lines=n_lines, points_per_line=pts, seed=seed, turns=2.0)),
                                  twist=twist)


def band_case(width=200, height=150, elliptic=True, transparent=False, line_width=0.02, tr=None, **settings):
    tr = tr or ribbon_scene()
    s = dict(BANDS)
    s.update(settings)
    s["use_analytic_elliptic_tubes"] = elliptic
    if elliptic:   # getLinePassTubeAabbRenderData(false, true): ribbon normals, band-width boxes
        pts, seg, _ = lvo.build_tube_aabb_render_data_ribbons(tr.positions, tr.attributes, tr.line_offsets, s["band_width"],
                                                              tr.ribbon_directions)
    else:
        pts, seg, _ = lvo.build_tube_aabb_render_data(tr.positions, tr.attributes, tr.line_offsets, line_width)
    tf = tfm.standard_transparent() if transparent else tfm.standard()
    return Case(pts, seg, tf, width, height, line_width, **s)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(transparent=True), dict(thick_bands=False), dict(use_halos=False),
                                dict(min_band_thickness=1.0), dict(num_samples_per_frame=3), dict(depth_cue_strength=0.7)])
def test_elliptic_tubes_frame_matches_the_oracle(hip_lib, kw):
    c = band_case(**kw)
    img = c.hip_context().render(11)
    ref, _ = c.oracle_render(11)             # brute force over all tubelets
    assert max_lsb_diff(img, ref) <= 2
    assert (img[..., :3] != 255).any(axis=2).sum() > 2000


@pytest.mark.gpu
def test_elliptic_trace_rays_bit_exact(hip_lib):
    c = band_case()
    ctx = c.hip_context()
    rng = np.random.default_rng(9)
    cam = np.array([0.0, 0.0, 0.8], np.float32)
    o = np.concatenate([np.tile(cam[None], (4000, 1)), rng.uniform(-0.25, 0.25, (2000, 3)).astype(np.float32)])
    d = rng.normal(size=(6000, 3)).astype(np.float32)
    d[:4000, 2] = -np.abs(d[:4000, 2]) * 4
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t, s, k = ctx.trace_rays(o, d, 1e-4, 1000.0)
    sc = c.oracle_scene()
    t2, s2 = sc.trace_rays_elliptic(o, d, 1e-4, 1000.0, 0.05, 0.3, cam, use_bvh=False)
    t3, s3 = sc.trace_rays_elliptic(o, d, 1e-4, 1000.0, 0.05, 0.3, cam, use_bvh=True)
    assert np.array_equal(t2.view(np.uint32), t3.view(np.uint32)) and np.array_equal(s2, s3)   # own-box rule: BVH == brute force
    assert np.array_equal(t.view(np.uint32), t2.view(np.uint32)) and np.array_equal(s, s2)
    assert (s != 0xFFFFFFFF).sum() > 800 and np.all(k == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(use_capped_tubes=False), dict(transparent=True), dict(depth_cue_strength=0.7)])
def test_band_shading_on_circular_tubes(hip_lib, kw):
    """Band data with "Elliptic Tubes" off: analytic capsules, USE_BANDS shading with useBand = false (conic halo with thickness 1,
    outline widths from depth / lineWidth * 0.25)."""
    c = band_case(elliptic=False, **kw)
    img = c.hip_context().render(11)
    ref, _ = c.oracle_render(11)
    assert max_lsb_diff(img, ref) <= 2
    plain = dict(c.settings)
    plain["use_ribbons"] = False
    c2 = Case(c.points, c.seg, c.tf, c.width, c.height, c.line_width, **plain)
    assert not np.array_equal(c2.hip_context().render(11), img)


@pytest.mark.gpu
@pytest.mark.parametrize("jitter", [False, True])
def test_elliptic_tubes_rtao(hip_lib, jitter):
    c = band_case(**dict(RTAO, use_jittered_primary_rays=jitter))
    ctx = c.hip_context()
    img = ctx.render(11)
    ao = ctx.get_ao()
    ref, ao_ref = c.oracle_render(11)
    assert np.array_equal(ao.view(np.uint32), ao_ref.view(np.uint32))
    assert max_lsb_diff(img, ref) <= 2
    assert (ao < 0.9).sum() > 500
    tile = (60, 40, 64, 48)
    assert np.array_equal(ctx.render(11, tile=tile), img[40:88, 60:124])


@pytest.mark.gpu
def test_band_options_are_validated(hip_lib):
    c = band_case()
    ctx = c.hip_context()
    from linevis_amd import capi
    ctx.set_option("geometry_mode", "Triangle Mesh")
    ctx.set_option("use_analytic_elliptic_tubes", False)
    with pytest.raises(capi.LineVisError):
        ctx.render(2)                                   # the PPLL gather of band data runs over the analytic geometry
    ctx.set_option("geometry_mode", "AABBs")
    ctx.set_option("use_analytic_elliptic_tubes", True)
    ctx.render(2)
    ctx.set_option("use_ribbons", False)
    with pytest.raises(capi.LineVisError):
        ctx.render(11)                                  # elliptic tubes without band data
    ctx.set_option("use_analytic_elliptic_tubes", False)
    ctx.render(11)


@pytest.mark.gpu
def test_band_data_through_the_plugin_surface(hip_lib, tmp_path):
    """LineDataFlow with ribbon directions (from a version-2 .binlines file) -> HipRayTracer: USE_BANDS is switched on by the data,
    `use_analytic_elliptic_tubes` selects the tubelets; the frames equal the ones of a context fed by hand."""
    from linevis_amd import host_api
    tr = ribbon_scene()
    path = str(tmp_path / "ribbons.binlines")
    scenes.write_binlines(path, tr)
    flow = host_api.LineDataFlow().load_binlines(path)
    assert flow.has_bands_data
    settings = dict(line_width=0.02, band_width=0.05, min_band_thickness=0.3, depth_cue_strength=0.5)
    r = host_api.HeadlessLineRenderer(11)
    r.set_rendering_resolution(160, 120)
    r.set_transfer_function(tfm.standard())
    r.set_line_data(flow)
    r.set_new_settings(settings)
    frames = {}
    for elliptic in (False, True):
        r.set_new_settings(dict(use_analytic_elliptic_tubes=elliptic))
        frames[elliptic] = r.render_frame()
        pos, att, off = flow.trajectories()
        if elliptic:
            pts, seg, _ = flow.tube_aabb_render_data_elliptic(0.05)
        else:
            pts, seg, _ = flow.tube_aabb_render_data(0.02)
        c = Case(pts, seg, tfm.standard(), 160, 120, 0.02, use_ribbons=True, use_analytic_elliptic_tubes=elliptic,
                 **{k: v for k, v in settings.items() if k != "line_width"})
        ctx = c.hip_context()
        lo, hi = flow.attribute_range()
        ctx.set_transfer_function(c.tf, lo, hi)
        view, proj, fovy, near, far = r.camera()
        ctx.set_camera(view, proj, fovy, near, far, 160, 120)
        assert np.array_equal(frames[elliptic], ctx.render(11))
    assert not np.array_equal(frames[False], frames[True])
    r.set_new_settings(dict(use_ribbons=False))      # the data keeps its ribbons, the renderer ignores them
    assert not np.array_equal(r.render_frame(), frames[True])
    r.set_new_settings(dict(use_ribbons=True))       # LineDataFlow::useRibbons is static, as in the reference: leave it on


def _fragment_lists(nodes, start):
    out = {}
    for pix in np.nonzero(start != 0xFFFFFFFF)[0]:
        l, i = [], int(start[pix])
        while i != 0xFFFFFFFF:
            l.append((int(nodes[i, 1]), int(nodes[i, 0])))
            i = int(nodes[i, 2])
        out[int(pix)] = sorted(l)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(elliptic=False), dict(thick_bands=False, use_halos=False),
                                dict(RTAO, ambient_occlusion_iterations=1), dict(ppll_tile_width=8, ppll_tile_height=8)])
def test_ppll_of_band_data_matches_the_oracle(hip_lib, kw):
    """PPLL of band data: the fragments are the entry hits of the elliptic tubelets (or of the capsules) shaded with USE_BANDS -- the
    ray-entry form of the elliptic tubes the reference's rasterisers draw in the ribbon primitive mode.  Per pixel the multiset of
    (colour, depth) fragments equals the oracle's bit for bit; the resolved frame within 2 LSB."""
    c = band_case(width=100, height=70, transparent=True, ppll_fragment_source="capsule_entry", **kw)   # (auto = the rasterised prism)
    ctx = c.hip_context()
    img = ctx.render(2)
    ref, _ = c.oracle_render(2)
    assert max_lsb_diff(img, ref) <= 2
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    ao = sc.render_ao(P) if P.useAmbientOcclusion else None
    on, os_, ocnt = sc.ppll_gather(P, ao=ao)
    pw, ph = c.padded()
    hn, hs, hcnt = ctx.ppll_buffers(pw * ph, int(P.ppllLinkedListSize))
    assert hcnt == ocnt and hcnt > 1000
    assert _fragment_lists(hn, hs) == _fragment_lists(on, os_)
    on2, os2, ocnt2 = sc.ppll_gather(P, ao=ao, use_bvh=True)       # own-box rule: the tree finds what brute force finds
    assert ocnt2 == ocnt and _fragment_lists(on2, os2) == _fragment_lists(on, os_)
    if kw.get("elliptic", True):                                   # opaque tubelets (coverage 1): the front fragment is the ray
        opaque = band_case(width=100, height=70, ppll_fragment_source="capsule_entry", **kw)             # tracer's hit
        pctx = opaque.hip_context()
        pctx.set_option("ppll_fragment_colour", "ray_tracer")      # the ray tracer's fragment colour (the gather's own is the raster shader's)
        a = pctx.render(2)
        b = opaque.hip_context().render(11)
        assert max_lsb_diff(a, b) <= 2


@pytest.mark.gpu
@pytest.mark.parametrize("geometry,k,transparent", [("elliptic", 2, True), ("elliptic", 8, True), ("elliptic", 4, False),
                                                     ("capsules", 4, True), ("triangles", 4, True)])
def test_mlat_of_band_data_replay_parity(hip_lib, geometry, k, transparent):
    """USE_MLAT on band data: AnyHitEllipticTubeAnalytic (EllipticTubeRayTracing.glsl:463-466) = the elliptic closest-hit shading +
    insertNodeMlat, and the USE_BANDS variants of AnyHitTubeAnalytic / AnyHitTubeTriangles.  Same replay contract as
    tests/test_gpu_mlat.py: the kernel's recorded visiting order is replayed and validated by the oracle."""
    tr = ribbon_scene(n_lines=8)
    kw = dict(use_mlat=True, collect_stats=True, mlat_record_trace=True, mlat_num_nodes=k)
    mesh = None
    if geometry == "triangles":
        mesh = lvo.build_tube_triangle_render_data_ribbons(tr.positions, tr.attributes, tr.line_offsets, tr.ribbon_directions, 0.05,
                                                           0.3, 8)
        kw.update(geometry_mode="Triangle Mesh", tube_num_subdivisions=8)
    c = band_case(width=120, height=90, elliptic=geometry == "elliptic", transparent=transparent, tr=tr, **kw)
    ctx = c.hip_context()
    if mesh is not None:
        ctx.set_tube_triangle_mesh(*mesh)
    img = ctx.render(11)
    rec = ctx.mlat_trace()
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    ts = lvo.TriScene(*mesh, c.line_width) if mesh is not None else None
    ref, _, viol = sc.render_rt_mlat(P, k, trace=rec, tri_scene=ts)
    assert viol == 0 and max_lsb_diff(img, ref) <= 2
    assert len(rec) > 2000
    ctx.set_option("use_mlat", False)                   # close to the exact transparency loop over the same geometry
    loop = ctx.render(11)
    assert np.abs(img.astype(np.int32) - loop.astype(np.int32)).mean() < (5.0 if k < 4 else 2.5)
    if geometry == "elliptic" and not transparent:
        assert max_lsb_diff(img, loop) <= 2             # opaque tubelets: the nearest layer is the closest hit


@pytest.mark.gpu
def test_ppll_plugin_draws_band_data_as_elliptic_tubes(hip_lib):
    """The reference's rasterisers draw band data as elliptic tubes with USE_BANDS (ribbon primitive mode, LineDataFlow.cpp:476-481):
    the PPLL plugin uploads the tubelet geometry of the data set; its frame equals the one of a context fed by hand."""
    from linevis_amd import host_api
    tr = ribbon_scene()
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets, tr.ribbon_directions)
    settings = dict(line_width=0.02, band_width=0.05, min_band_thickness=0.3)
    r = host_api.HeadlessLineRenderer(2)
    r.set_rendering_resolution(96, 64)
    r.set_transfer_function(tfm.standard_transparent())
    r.set_line_data(flow)
    r.set_new_settings(settings)
    frame = r.render_frame()
    pts, seg, _ = flow.tube_aabb_render_data_elliptic(0.05)
    # (tube_num_subdivisions: band data raises the data set's 6 to 8, LineDataFlow.cpp:482-484 -- the rasterised prism has that many sides)
    c = Case(pts, seg, tfm.standard_transparent(), 96, 64, 0.02, use_ribbons=True, use_analytic_elliptic_tubes=True,
             band_width=0.05, min_band_thickness=0.3, use_capped_tubes=False, tube_num_subdivisions=8)
    ctx = c.hip_context()
    lo, hi = flow.attribute_range()
    ctx.set_transfer_function(c.tf, lo, hi)
    view, proj, fovy, near, far = r.camera()
    ctx.set_camera(view, proj, fovy, near, far, 96, 64)
    assert np.array_equal(frame, ctx.render(2)) and (frame[..., :3] != 255).any()
    plain = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets, None)
    r.set_line_data(plain)
    assert not np.array_equal(r.render_frame(), frame)


@pytest.mark.gpu
def test_streamribbons_from_the_tracer_to_the_renderer(hip_lib):
    """The producer side of band data: StreamlineTracingGrid::traceStreamribbons (lines on the GPU, ribbon directions from the
    helicity field) -> LineDataFlow -> the ray tracer's elliptic tubes; the frame equals the oracle's on the same ribbons."""
    from linevis_amd import host_api
    grid = host_api.StreamlineTracingGrid().load_abc_flow(24, 24, 24, 6.0)
    seeds = grid.regular_seeds(3, 3, 3)
    pos, att, off, rib = grid.trace_streamribbons(seeds, minimum_length=0.5, max_helicity_twist=0.5)
    assert len(off) - 1 >= 10 and rib.shape == pos.shape
    npos = host_api.normalize_positions(pos)
    flow = host_api.LineDataFlow().set_trajectories(npos, att[1], off, rib)      # attribute: Velocity Magnitude
    assert flow.has_bands_data
    bw = 0.03
    pts, seg, _ = flow.tube_aabb_render_data_elliptic(bw)
    ref_pts, ref_seg, _ = lvo.build_tube_aabb_render_data_ribbons(npos, att[1], off, bw, rib)
    assert np.array_equal(pts.view(np.uint8), ref_pts.view(np.uint8)) and np.array_equal(seg, ref_seg)
    c = Case(pts, seg, tfm.standard(), 200, 150, 0.01, use_ribbons=True, use_analytic_elliptic_tubes=True, band_width=bw,
             min_band_thickness=0.2)
    lo, hi = flow.attribute_range()
    ctx = c.hip_context()
    ctx.set_transfer_function(c.tf, lo, hi)
    img = ctx.render(11)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    P.attrMin, P.attrMax = lo, hi
    assert max_lsb_diff(img, sc.render_rt(P, use_bvh=True)) <= 2
    assert (img[..., :3] != 255).any(axis=2).sum() > 3000


@pytest.mark.gpu
@pytest.mark.parametrize("elliptic", [True, False])
def test_band_data_rtao_over_the_elliptic_triangle_tubes(hip_lib, elliptic):
    """The reference's RTAO geometry for band data: the elliptic triangle tessellation (createCappedTriangleEllipticTubesRenderDataCPU;
    the data set is in its ribbon primitive mode).  rtao_geometry = triangle_tubes with the mesh LineDataFlow tessellates: AO image bit
    for bit against the oracle's triangle RTAO on the same mesh, colour pass (elliptic tubelets or circular tubes) within 2 LSB."""
    from linevis_amd import host_api
    tr = ribbon_scene()
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets, tr.ribbon_directions)
    mesh = flow.tube_triangle_render_data_bands(0.05, 0.3, 8)
    ref_mesh = lvo.build_tube_triangle_render_data_ribbons(tr.positions, tr.attributes, tr.line_offsets, tr.ribbon_directions, 0.05, 0.3, 8)
    assert all(np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8)) for a, b in zip(mesh, ref_mesh))
    c = band_case(elliptic=elliptic, tube_num_subdivisions=8, rtao_geometry="triangle_tubes", **RTAO)
    ctx = c.hip_context()
    ctx.set_tube_triangle_mesh(*mesh)
    img = ctx.render(11)
    ao = ctx.get_ao()
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    tsc = lvo.TriScene(mesh[0], mesh[1], mesh[2], c.line_width)   # the pad of the triangle test follows line_width on both sides
    ao_ref = c.oracle_ao(sc, P, render_ao=lambda t: tsc.render_ao(P, tile=t))
    assert np.array_equal(ao.view(np.uint32), ao_ref.view(np.uint32)) and (ao < 0.9).sum() > 500
    assert max_lsb_diff(img, sc.render_rt(P, ao=ao_ref)) <= 2
    # the analytic-tubelet AO of the same frame differs per pixel but not systematically
    c2 = band_case(elliptic=True, tube_num_subdivisions=8, **RTAO)
    ctx2 = c2.hip_context()
    ctx2.render(11)
    ao2 = ctx2.get_ao()
    both = (ao < 1.0) & (ao2 < 1.0)
    if elliptic:
        assert abs(float(ao[both].mean()) - float(ao2[both].mean())) < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(transparent=True), dict(use_capped_tubes=False), dict(thick_bands=False),
                                dict(rtao=True)])
def test_band_data_in_triangle_mesh_geometry_mode(hip_lib, kw):
    """geometry_mode = "Triangle Mesh" on a band data set: the elliptic triangle tubes, closest-hit shading with USE_BANDS
    (interpolated angle / line position / line normal, useBand = true), optionally with RTAO on the same mesh."""
    from linevis_amd import host_api
    kw = dict(kw)
    rtao = kw.pop("rtao", False)
    tr = ribbon_scene()
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets, tr.ribbon_directions)
    mesh = flow.tube_triangle_render_data_bands(0.05, 0.3, 8)
    s = dict(geometry_mode="Triangle Mesh", tube_num_subdivisions=8)
    if rtao:
        s.update(RTAO, rtao_geometry="triangle_tubes")
    c = band_case(elliptic=False, **dict(s, **kw))
    ctx = c.hip_context()
    ctx.set_tube_triangle_mesh(*mesh)
    img = ctx.render(11)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    tsc = lvo.TriScene(mesh[0], mesh[1], mesh[2], c.line_width)
    ao_ref = None
    if rtao:
        ao_ref = c.oracle_ao(sc, P, render_ao=lambda t: tsc.render_ao(P, tile=t))
        assert np.array_equal(ctx.get_ao().view(np.uint32), ao_ref.view(np.uint32))
    ref = tsc.render_rt(sc, P, ao=ao_ref)
    assert max_lsb_diff(img, ref) <= 2
    assert (img[..., :3] != 255).any(axis=2).sum() > 2000
    P.useBands = 0
    assert not np.array_equal(tsc.render_rt(sc, P, ao=ao_ref), ref)     # the band shading is what is being compared
