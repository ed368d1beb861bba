"""Rotating helicity bands on the GPU (k_render_rt / k_ppll_gather / k_render_rt_mlat <..., LV_SHADE_HELICITY>) through the C-ABI and
the plugin surface, against the oracle."""
import numpy as np
import pytest

from common import Case, max_lsb_diff
from linevis_amd import capi, host_api, scenes, transfer_function as tfm
from oracle import lvo
from test_helicity_bands import helix_with_helicity

pytestmark = pytest.mark.gpu

HEL = dict(rotating_helicity_bands=True, band_subdivisions=6, separator_width=0.2, helicity_rotation_factor=1.0)
RTAO = dict(ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0, ambient_occlusion_gamma=1.0,
            ambient_occlusion_radius=0.15, ambient_occlusion_distance_based=True, ambient_occlusion_iterations=2,
            ambient_occlusion_samples_per_frame=4)


def helicity_case(width=200, height=150, line_width=0.03, transparent=False, **settings):
    tr, hel = helix_with_helicity()
    pts, seg, _ = lvo.build_tube_aabb_render_data(tr.positions, tr.attributes, tr.line_offsets, line_width, helicities=hel)
    s = dict(HEL)
    s.update(settings)
    tf = tfm.standard_transparent() if transparent else tfm.standard()
    return Case(pts, seg, tf, width, height, line_width, **s), tr, hel


@pytest.mark.parametrize("kw", [dict(), dict(transparent=True), dict(use_halos=False), dict(use_capped_tubes=False),
                                dict(band_subdivisions=3, separator_width=0.5, helicity_rotation_factor=0.3),
                                dict(num_samples_per_frame=3), dict(depth_cue_strength=0.7), dict(RTAO),
                                dict(geometry_mode="Linear Swept Spheres")])
def test_ray_tracer_frame_matches_the_oracle(hip_lib, kw):
    c, _, _ = helicity_case(**kw)
    ctx = c.hip_context()
    img = ctx.render(11)
    ref, ao = c.oracle_render(11)
    if ao is not None:
        assert np.array_equal(ctx.get_ao().view(np.uint32), ao.view(np.uint32))
    assert max_lsb_diff(img, ref) <= 2
    c.settings["rotating_helicity_bands"] = False
    assert (np.abs(c.hip_context().render(11).astype(np.int32) - img.astype(np.int32)).max(axis=2) > 20).sum() > 300   # stripes


def test_triangle_mesh_geometry_mode(hip_lib):
    """LineAttributesBarycentric.glsl:60-112: interpolated rotation, linear continuation on the caps, and UNIFORM_HELICITY_BAND_WIDTH
    (use_uniform_twist_line_width, the reference's default: separator width / cos(atan(rotation per length x r))); the triangle
    line-point table's rotation runs on across the trajectories."""
    tr, hel = helix_with_helicity()
    lw = 0.03
    mesh = lvo.build_tube_triangle_render_data(tr.positions, tr.attributes, tr.line_offsets, lw, 8, helicities=hel)
    frames = {}
    for transparent in (False, True):
        for uniform in (True, False):
            c, _, _ = helicity_case(line_width=lw, transparent=transparent, geometry_mode="Triangle Mesh", tube_num_subdivisions=8,
                                    helicity_rotation_factor=0.25, use_uniform_twist_line_width=uniform)
            ctx = c.hip_context()
            ctx.set_tube_triangle_mesh(*mesh)
            img = ctx.render(11)
            sc = c.oracle_scene()
            P = c.oracle_params(sc)
            assert P.uniformHelicityBandWidth == int(uniform)
            ref = lvo.TriScene(*mesh, lw).render_rt(sc, P)
            assert max_lsb_diff(img, ref) <= 2
            frames[(transparent, uniform)] = img
    assert not np.array_equal(frames[(False, True)], frames[(False, False)])     # wider stripes where the bands are steep
    assert (img[..., :3] != 255).any(axis=2).sum() > 3000
    # the analytic closest-hit shader has no such branch: the switch changes nothing there
    a, _, _ = helicity_case(use_uniform_twist_line_width=True)
    b, _, _ = helicity_case(use_uniform_twist_line_width=False)
    assert np.array_equal(a.hip_context().render(11), b.hip_context().render(11))


def test_ppll_and_mlat(hip_lib):
    c, _, _ = helicity_case(width=120, height=90, transparent=True)
    ctx = c.hip_context()
    img = ctx.render(2)
    ref, _ = c.oracle_render(2)
    assert max_lsb_diff(img, ref) <= 2
    c2, _, _ = helicity_case(width=120, height=90, transparent=True, use_mlat=True, collect_stats=True, mlat_record_trace=True,
                             mlat_num_nodes=4)
    ctx = c2.hip_context()
    img = ctx.render(11)
    rec = ctx.mlat_trace()
    sc = c2.oracle_scene()
    ref, _, viol = sc.render_rt_mlat(c2.oracle_params(sc), 4, trace=rec)
    assert viol == 0 and max_lsb_diff(img, ref) <= 2 and len(rec) > 2000


def test_options_are_validated(hip_lib):
    c, _, _ = helicity_case()
    ctx = c.hip_context()
    for key, bad in (("band_subdivisions", 0), ("separator_width", -1.0), ("helicity_rotation_factor", "x")):
        with pytest.raises(capi.LineVisError):
            ctx.set_option(key, bad)
    ctx.set_option("use_ribbons", True)
    with pytest.raises(capi.LineVisError):
        ctx.render(11)                       # USE_BANDS and USE_ROTATING_HELICITY_BANDS exclude each other


def test_from_the_streamline_tracer_through_the_plugin_surface(hip_lib):
    """ABC flow -> StreamlineTracingGrid (attributes incl. "Helicity") -> LineDataFlow -> HipRayTracer with rotating_helicity_bands:
    the data set computes lineRotation, the renderer uploads the switches; the frame equals a context fed by hand and the oracle."""
    grid = host_api.StreamlineTracingGrid().load_abc_flow(24, 24, 24, 6.0)
    seeds = grid.regular_seeds(3, 3, 3)
    pos, att, off = grid.trace_streamlines(seeds, minimum_length=0.5)
    names = grid.attribute_names()
    assert "Helicity" in names
    npos = host_api.normalize_positions(pos)
    flow = host_api.LineDataFlow().set_trajectories_multi(npos, att, names, off, selected=names.index("Velocity Magnitude"))
    assert flow.has_helicity
    settings = dict(line_width=0.02, rotating_helicity_bands=True, helicity_rotation_factor=0.05, depth_cue_strength=0.5)
    frames = {}
    for mode in (11, 2):
        r = host_api.HeadlessLineRenderer(mode)
        r.set_rendering_resolution(200, 150)
        r.set_transfer_function(tfm.standard())
        r.set_line_data(flow)
        r.set_new_settings(settings)
        frames[mode] = r.render_frame()
        hel = att[names.index("Helicity")]
        pts, seg, _ = lvo.build_tube_aabb_render_data(npos, att[names.index("Velocity Magnitude")], off, 0.02, helicities=hel,
                                                      max_helicity=flow.max_helicity)
        c = Case(pts, seg, tfm.standard(), 200, 150, 0.02, rotating_helicity_bands=True, helicity_rotation_factor=0.05,
                 depth_cue_strength=0.5, use_capped_tubes=(mode == 11))   # rasterisers: no caps outside the triangle-mesh modes
        ctx = c.hip_context()
        lo, hi = flow.attribute_range()
        ctx.set_transfer_function(c.tf, lo, hi)
        view, proj, fovy, near, far = r.camera()
        ctx.set_camera(view, proj, fovy, near, far, 200, 150)
        assert np.array_equal(frames[mode], ctx.render(mode))
        if mode == 11:
            sc = c.oracle_scene()
            P = c.oracle_params(sc)
            P.attrMin, P.attrMax = lo, hi
            assert max_lsb_diff(frames[mode], sc.render_rt(P, use_bvh=True)) <= 2
    # the twist-line texture through the plugin: the data set holds the pixels, the renderer uploads them with the switches
    from test_helicity_bands import _twist_image
    tex = _twist_image(w=64, h=4)
    flow.set_twist_line_texture(tex)
    r.set_line_data(flow)
    r.set_new_settings(dict(use_twist_line_texture=True, twist_line_texture_filtering_mode="Linear Mipmap Nearest"))
    textured = r.render_frame()
    ctx.set_twist_line_texture(tex)
    ctx.set_options(dict(use_twist_line_texture=True, twist_line_texture_filtering_mode="Linear Mipmap Nearest"))
    assert np.array_equal(textured, ctx.render(2)) and not np.array_equal(textured, frames[2])
    r.set_new_settings(dict(rotating_helicity_bands=False))
    assert not np.array_equal(r.render_frame(), frames[2])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["Nearest", "Linear", "Nearest Mipmap Nearest", "Linear Mipmap Nearest", "Nearest Mipmap Linear",
                                  "Linear Mipmap Linear"])
def test_twist_line_texture(hip_lib, mode):
    """USE_HELICITY_BANDS_TEXTURE: lv_set_twist_line_texture + use_twist_line_texture replace the separator stripes by the texture
    (all four components).  Ray tracer: level 0 (texture() without derivatives); mode 2 on the rasterised prism: textureGrad with the
    quad's derivatives of (phi + fragmentRotation) / 2 pi -- per-pixel lists bit for bit, frames <= 2 LSB, for every filtering mode."""
    from test_helicity_bands import _twist_image
    from test_prism_raster import _walk_order
    img_tex = _twist_image(w=128, h=8)
    c, _, _ = helicity_case(width=120, height=90, transparent=True, use_twist_line_texture=True, twist_line_texture_filtering_mode=mode)
    ctx = c.hip_context()
    ctx.set_twist_line_texture(img_tex)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    rt = ctx.render(11)
    with lvo.twist_line_texture(img_tex, mode):
        ref = sc.render_rt(P)
        on, os_, ocnt = sc.ppll_gather(P)
        pref = sc.render_ppll(P)
    assert max_lsb_diff(rt, ref) <= 2
    pimg = ctx.render(2)
    pw, ph = c.padded()
    hn, hs, hcnt = ctx.ppll_buffers(pw * ph, int(P.ppllLinkedListSize))
    assert hcnt == ocnt and _walk_order(hn, hs) == _walk_order(on, os_)
    assert max_lsb_diff(pimg, pref) <= 2
    ctx.set_option("use_twist_line_texture", False)            # back to the stripes
    assert not np.array_equal(ctx.render(11), rt)
    ctx.set_option("use_twist_line_texture", True)
    ctx.set_twist_line_texture(None)                            # unloaded: the define is off (LineDataFlow.cpp:2437), stripes again
    assert not np.array_equal(ctx.render(11), rt)
    with pytest.raises(Exception):
        ctx.set_option("twist_line_texture_max_anisotropy", 4)
