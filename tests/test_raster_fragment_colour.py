"""Raster variant of the PPLL fragment colour.

The PPLL gather of the reference runs the RASTER tube shader (LinePassGeometryShaderTubes.glsl:732-1129), whose tail differs from
the ray tracer's computeFragmentColor (RayHitCommon.glsl): EPSILON_OUTLINE = 0.0, EPSILON_WHITE = fwidth(ribbonPosition)
(:1079-1087), cap halo min(ribbonPosition, abs(ribbonPosition2)) (:815).  For ray-generated fragments the build defines fwidth by
the ribbon coordinate of the VIEWING RAYS through the 2 x 2 quad partners with respect to the fragment's segment (oracle
`RasterQuad`): |cross(newV, n)| of a hit is the distance between the ray and the tube axis over the radius, a function of the ray
alone, so the partners' "helper invocations" have a value whether or not their rays hit the tube.

Independent float64 checks (numpy, written from the GLSL and from the geometry, not through oracle/):
  * the ray form equals the shader's ribbonPosition at true hits of the same ray (tube mantle and cap);
  * fwidth from the ray form equals the finite differences of the shader's ribbonPosition between TRUE hits of neighbouring pixel
    rays on the same cylinder, wherever all three rays hit it;
  * the raster tail of computeFragmentColor against the restatement of test_independent_restatement.py with the three differences.
"""
import numpy as np
import pytest

from common import small_case, Case
from linevis_amd import camera
from oracle import lvo
import test_independent_restatement as ir

RNG = np.random.default_rng(20260929)


def unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def shader_ribbon_at_hit(cam, hit, axis_point, t):
    """RayHitCommon.glsl:141-146,353-372 = LinePassGeometryShaderTubes.glsl:777-783,944-962 in float64: the tube-mantle branch."""
    foot = axis_point + ((hit - axis_point) * t).sum(-1, keepdims=True) * t
    n = unit(hit - foot)
    v = unit(cam - hit)
    helper = unit(np.cross(t, v))
    new_v = unit(np.cross(helper, t))
    c = np.cross(new_v, n)
    rp = np.linalg.norm(c, axis=-1)
    rp = np.where((t * c).sum(-1) < 0.0, -rp, rp)
    return np.clip(rp, -1.0, 1.0)


def ray_cylinder(cam, d, axis_point, t, r):
    """nearest intersection of rays cam + s d with the infinite cylinder (float64); nan where the ray misses"""
    w = cam - axis_point
    wp = w - (w * t).sum(-1, keepdims=True) * t
    dp = d - (d * t).sum(-1, keepdims=True) * t
    A = (dp * dp).sum(-1)
    B = 2.0 * (dp * wp).sum(-1)
    Cc = (wp * wp).sum(-1) - r * r
    disc = B * B - 4 * A * Cc
    s = np.where(disc >= 0, (-B - np.sqrt(np.maximum(disc, 0))) / (2 * A), np.nan)
    return cam + d * s[..., None], s


def test_ribbon_of_ray_equals_the_shaders_ribbon_position_at_the_hit():
    n = 4000
    cam = np.array([0.0, 0.0, 0.8])
    axis_point = RNG.uniform(-0.2, 0.2, 3)
    t = unit(RNG.normal(size=3))
    r = 0.01
    # rays aimed at points near the axis so that most of them hit
    target = axis_point + t * RNG.uniform(-0.3, 0.3, (n, 1)) + RNG.normal(size=(n, 3)) * 0.006
    d = unit(target - cam)
    hit, s = ray_cylinder(cam, d, axis_point, t, r)
    ok = np.isfinite(s) & (s > 0)
    assert ok.sum() > 2000
    want = shader_ribbon_at_hit(cam, hit[ok], axis_point, t)
    got = lvo.ribbon_of_rays(cam, d[ok], axis_point, t, r)
    assert np.abs(got - want).max() < 2e-4
    assert (np.abs(want) > 0.9).sum() > 50 and (np.abs(want) < 0.1).sum() > 50      # silhouettes and centre lines both occur
    # rays that miss have |coordinate| clamped to 1 (the shader clamps ribbonPosition)
    assert np.all(np.abs(lvo.ribbon_of_rays(cam, d[~ok], axis_point, t, r)) == 1.0)


def raster_cap_ribbon(cam, q, centre, t):
    """LinePassGeometryShaderTubes.glsl:785-815 in float64 at points q with the sphere's normal direction"""
    nrm = unit(q - centre)
    v = unit(cam - q)
    helper = unit(np.cross(t, v))
    new_v = unit(np.cross(helper, t))
    cvn = np.cross(v, nrm)
    rp = np.linalg.norm(cvn, axis=-1)
    rp2 = np.linalg.norm(np.cross(new_v, nrm), axis=-1)
    rp2 = np.where((t * cvn).sum(-1) < 0, -rp2, rp2)
    return np.minimum(rp, np.abs(np.clip(rp2, -1, 1)))


def test_cap_ribbon_of_ray_is_the_raster_shaders_cap_branch_on_the_tangent_plane():
    """One cap fragment, many partner-like rays: the oracle's value for a ray is the raster shader's cap coordinate (float64
    restatement) at the point where that ray meets the tangent plane of the cap at the fragment; the fragment's own ray reproduces
    the fragment's own coordinate."""
    cam = np.array([0.1, -0.05, 0.8])
    centre = RNG.uniform(-0.2, 0.2, 3)
    t = unit(RNG.normal(size=3))
    r = 0.01
    for _ in range(20):
        d0 = unit(centre + RNG.normal(size=3) * 0.004 - cam)
        w = cam - centre
        b = (w * d0).sum()
        disc = b * b - ((w * w).sum() - r * r)
        if disc < 0:
            continue
        hit = cam + d0 * (-b - np.sqrt(disc))
        nrm = unit(hit - centre)
        d = unit(d0 + RNG.normal(size=(200, 3)) * 2e-3)
        d[0] = d0
        s = ((hit - cam) * nrm).sum() / (d * nrm).sum(-1)
        q = cam + d * s[:, None]
        want = raster_cap_ribbon(cam, q, centre, t)
        got = lvo.ribbon_of_rays(cam, d, centre, t, r, cap_hit=hit, cap_normal=nrm)
        assert np.abs(got - want).max() < 2e-4
        assert abs(float(got[0]) - float(raster_cap_ribbon(cam, hit[None], centre, t)[0])) < 2e-4


def test_fwidth_of_the_ray_form_equals_finite_differences_between_true_hits():
    """A long cylinder in front of the default camera, 200 x 120 pixels: wherever a pixel's ray and both quad partners' rays hit the
    cylinder, |rp(x ^ 1, y) - rp(x, y)| + |rp(x, y ^ 1) - rp(x, y)| of the shader's ribbonPosition at the TRUE hits (float64)
    equals the oracle's fwidth of the ray form."""
    W, H = 200, 120
    view, proj, fovy, near, far = camera.default_camera(W, H)
    cam = ir.camera_position(np.asarray(view, dtype=np.float64))
    inv_proj = np.linalg.inv(np.asarray(proj, dtype=np.float64).reshape(4, 4).T)
    inv_view = np.linalg.inv(np.asarray(view, dtype=np.float64).reshape(4, 4).T)
    ys, xs = np.mgrid[0:H, 0:W]
    ndc = np.stack([2.0 * (xs + 0.5) / W - 1.0, 2.0 * (ys + 0.5) / H - 1.0, np.ones_like(xs, float), np.ones_like(xs, float)], -1)
    tgt = ndc @ inv_proj.T                                         # TubeRayTracing.glsl:225-226
    dn = unit(tgt[..., :3])
    d = (np.concatenate([dn, np.zeros_like(dn[..., :1])], -1) @ inv_view.T)[..., :3]
    axis_point = np.array([0.01, -0.02, 0.0])
    t = unit(np.array([0.8, 0.5, 0.3]))
    r = 0.02
    hit, s = ray_cylinder(cam, d, axis_point, t, r)
    okh = np.isfinite(s)
    rp = np.where(okh, shader_ribbon_at_hit(cam, np.nan_to_num(hit), axis_point, t), np.nan)
    px = rp[:, np.arange(W) ^ 1]
    py = rp[np.arange(H) ^ 1, :]
    want = np.abs(px - rp) + np.abs(py - rp)
    sel = np.isfinite(want)
    assert sel.sum() > 500
    f = lvo.ribbon_of_rays(cam, d.reshape(-1, 3), axis_point, t, r).reshape(H, W).astype(np.float64)
    got = np.abs(f[:, np.arange(W) ^ 1] - f) + np.abs(f[np.arange(H) ^ 1, :] - f)
    assert np.abs(got[sel] - want[sel]).max() < 5e-4
    assert want[sel].max() > 0.05       # the tube is a few pixels wide: the derivative is not negligible


def raster_colour(tf, P, frag, normal, tangent, is_cap, attr, ao, eps_white):
    """LinePassGeometryShaderTubes.glsl tail in float64: the restatement of test_independent_restatement.py with (1) the cap halo
    min(rp, |rp2|) (:815), (2) EPSILON_OUTLINE = 0 -> coverage 1 (:1079,1087), (3) EPSILON_WHITE = fwidth(ribbonPosition) (:1083)."""
    view = np.asarray(P.view[:], dtype=np.float64)
    cam = ir.camera_position(view)
    frag_color = ir.transfer_function(tf, attr, P.attrMin, P.attrMax)
    n = ir.normalize(normal)
    v = ir.normalize(cam - frag)
    t = ir.normalize(tangent)
    helper = ir.normalize(np.cross(t, v))
    new_v = ir.normalize(np.cross(helper, t))
    c_vn = np.cross(v, n)
    rp_cap = ir.length(c_vn)
    c2 = np.cross(new_v, n)
    rp2 = ir.length(c2)
    rp2 = np.where(ir.dot(t, c_vn) < 0.0, -rp2, rp2)
    rp_cap = np.minimum(rp_cap, np.abs(ir.clamp(rp2, -1.0, 1.0)))                     # :785-815
    rp_tube = ir.length(c2)
    rp_tube = ir.clamp(np.where(ir.dot(t, c2) < 0.0, -rp_tube, rp_tube), -1.0, 1.0)   # :944-962
    rp = np.where(is_cap & bool(P.useCappedTubes), rp_cap, rp_tube) if P.useHalos else np.zeros(len(frag))
    m = view.reshape(4, 4).T
    ssp = (m @ np.concatenate([frag, np.ones((len(frag), 1))], axis=1).T).T[:, :3]
    shaded = ir.blinn_phong_shading_tube(frag_color, frag, ssp, n, t, cam, bool(P.useAmbientOcclusion), ao, P.aoGamma, P.aoStrength,
                                         bool(P.useDepthCues), P.minDepth, P.maxDepth, P.depthCueStrength)
    abs_coords = np.abs(rp)
    ew = eps_white if P.useHalos else np.zeros(len(frag))
    with np.errstate(divide="ignore", invalid="ignore"):
        x = (abs_coords - (0.7 - ew)) / (2.0 * ew)
    x = np.where(np.isnan(x), 0.0, x)                     # 0 / 0 -> clamp(NaN) = 0 with fmax(NaN, 0) = 0
    tt = ir.clamp(x, 0.0, 1.0)
    w = tt * tt * (3.0 - 2.0 * tt)                                                     # :1094-1096
    fg = 1.0 - np.asarray(P.background[:], dtype=np.float64)
    rgb = ir.mix(shaded[:, :3], fg[:3], w[:, None])
    return np.concatenate([rgb, shaded[:, 3:4]], axis=1)                              # coverage = 1


@pytest.mark.parametrize("variant", ["plain", "ao_depthcue", "uncapped", "no_halos"])
def test_raster_tail_against_the_float64_restatement(variant):
    settings = {"plain": {}, "uncapped": dict(use_capped_tubes=False), "no_halos": dict(use_halos=False),
                "ao_depthcue": dict(ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=0.7,
                                    ambient_occlusion_gamma=2.2, depth_cue_strength=0.8)}[variant]
    c = small_case(width=160, height=90, line_width=0.004, transparent=True, background=(0.9, 0.95, 1.0, 1.0), **settings)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    rng = np.random.default_rng(77)
    frag, normal, tangent, is_cap, attr, ao = ir.random_inputs(rng, 8000)
    eps = rng.uniform(0.0, 0.6, 8000)
    eps[:200] = 0.0                                           # fwidth == 0: the white outline becomes a step at 0.7
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    got, _ = sc.compute_fragment_color_raster(P, frag, normal, tangent, is_cap, attr, ao, eps)
    want = raster_colour(c.tf.astype(np.float64), P, f32(frag), f32(normal), f32(tangent), is_cap, f32(attr), f32(ao), f32(eps))
    err = np.abs(got.astype(np.float64) - want)
    # a smoothstep whose edges are eps apart turns an input error of 1e-6 into 1e-6 / eps: leave out the few samples that sit
    # inside a nearly degenerate transition
    sel = (eps > 1e-3) | (eps == 0.0)
    assert err[sel].max() < 5e-4, (variant, err[sel].max())
    assert np.all(got[:, 3] == ir.transfer_function(c.tf.astype(np.float64), f32(attr), P.attrMin, P.attrMax)[:, 3].astype(np.float32)) \
        or np.abs(got[:, 3] - ir.transfer_function(c.tf.astype(np.float64), f32(attr), P.attrMin, P.attrMax)[:, 3]).max() < 1e-6


def test_ppll_frames_differ_between_the_two_variants_only_near_the_outline():
    """Whole PPLL frame, raster variant (default) against the ray tracer's variant (what rounds 1-2 shaded with): interiors of the
    tubes agree, the white outline and the silhouette alpha change."""
    c = small_case(width=200, height=120, n_lines=40, pts_per_line=40, line_width=0.012, transparent=True, use_capped_tubes=False)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    a = sc.render_ppll(P, use_bvh=True)
    with lvo.ppll_ray_tracer_fragment_colour():
        b = sc.render_ppll(P, use_bvh=True)
    d = np.abs(a.astype(np.int32) - b.astype(np.int32)).max(axis=2)
    covered = (a[..., :3] != a[0, 0, :3]).any(axis=2)
    assert covered.sum() > 3000
    assert 0 < (d > 2).sum() < 0.9 * covered.sum()
    assert (d[covered] == 0).sum() > 0.02 * covered.sum()
    assert np.array_equal(sc.render_ppll(P, use_bvh=True), a)          # the switch resets


# ---------------------------------------------------------------- GPU
def _lists(nodes, start):
    """per-pixel multisets {(colour, depth bits)} of a PPLL node pool"""
    out = {}
    for pix in np.nonzero(start != 0xFFFFFFFF)[0]:
        i, l = int(start[pix]), []
        while i != 0xFFFFFFFF:
            l.append((int(nodes[i, 0]), int(nodes[i, 1])))
            i = int(nodes[i, 2])
        out[int(pix)] = sorted(l)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["uncapped", "capped", "capped_ao_depthcue", "no_halos", "thin"])
def test_hip_ppll_gather_shades_with_the_raster_variant(hip_lib, variant):
    """k_ppll_gather against the oracle's gather: fragment multisets (packed colour, depth bits) bit for bit, frames <= 2 LSB;
    ppll_fragment_colour = ray_tracer gives the previous rounds' picture (the oracle under its switch)."""
    settings = {"uncapped": dict(use_capped_tubes=False), "capped": {}, "no_halos": dict(use_halos=False),
                "capped_ao_depthcue": dict(ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0,
                                           ambient_occlusion_iterations=2, ambient_occlusion_samples_per_frame=4, depth_cue_strength=0.8),
                "thin": dict(use_capped_tubes=False)}[variant]
    lw = 0.003 if variant == "thin" else 0.015
    c = small_case(width=176, height=120, n_lines=40, pts_per_line=40, line_width=lw, transparent=True, **settings)
    ctx = c.hip_context()
    img = ctx.render(2)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    ao = sc.render_ao(P) if P.useAmbientOcclusion else None
    on, os_, ocnt = sc.ppll_gather(P, ao=ao)
    pw, ph = c.padded()
    hn, hs, hcnt = ctx.ppll_buffers(pw * ph, int(P.ppllLinkedListSize))
    assert hcnt == ocnt and hcnt > 1000
    assert _lists(hn, hs) == _lists(on, os_)
    ref = sc.render_ppll(P, ao=ao)
    assert np.abs(img.astype(np.int32) - ref.astype(np.int32)).max() <= 2
    ctx.set_option("ppll_fragment_colour", "ray_tracer")
    img_rt = ctx.render(2)
    with lvo.ppll_ray_tracer_fragment_colour():
        ref_rt = sc.render_ppll(P, ao=ao)
    assert np.abs(img_rt.astype(np.int32) - ref_rt.astype(np.int32)).max() <= 2
    if variant != "no_halos":
        assert not np.array_equal(img, img_rt)
    # tiles reproduce the frame (the quad partners of a pixel may lie in another tile: they are rays, not neighbours' results)
    assert np.array_equal(ctx.render(2, tile=(0, 0, 176, 120)), img_rt)
    ctx.set_option("ppll_fragment_colour", "raster")
    assert np.array_equal(ctx.render(2, tile=(37, 21, 50, 33)), img[21:54, 37:87])


@pytest.mark.gpu
def test_full_size_config4_raster_vs_ray_tracer_variant(hip_lib):
    """BASELINE.json config 4 (1 M segments, 1920 x 1080, <= 64 fragments per pixel): how many pixels the fragment-colour variant moves;
    written to gpurun_out/deviations_ppll.json (copied to profiles/ by hand)."""
    import json, os
    from common import ROOT
    from linevis_amd import host_api, scenes, transfer_function as tfm
    tr = scenes.normalize(scenes.tornado())
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    pts, seg, _ = flow.tube_aabb_render_data(0.002)
    c = Case(pts, seg, tfm.standard_transparent(), 1920, 1080, 0.002, ppll_max_num_frags=64, ppll_expected_avg_depth_complexity=20,
             use_capped_tubes=False)
    ctx = c.hip_context()
    ctx.set_transfer_function(c.tf, *flow.attribute_range())
    a = ctx.render(2)
    ctx.set_option("ppll_fragment_colour", "ray_tracer")
    b = ctx.render(2)
    d = np.abs(a.astype(np.int32) - b.astype(np.int32)).max(axis=2)
    covered = int(((a[..., :3] != a[0, 0, :3]).any(axis=2) | (b[..., :3] != b[0, 0, :3]).any(axis=2)).sum())
    rep = {"what": "config 4 PPLL frame, fragment colour of the raster tube shader (LinePassGeometryShaderTubes.glsl:785-815,1079-1087; "
                   "default from round 3) vs the ray tracer's computeFragmentColor (RayHitCommon.glsl; rounds 1-2), 1920x1080, line "
                   "width 0.002, uncapped tubes (programmable-pull mode)",
           "pixels": int(d.size), "covered": covered, "differ": int((d > 0).sum()), "differ_gt_2lsb": int((d > 2).sum()),
           "max_lsb": int(d.max())}
    assert covered > 100000 and rep["differ"] > 0
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(rep, open(os.path.join(out, "deviations_ppll.json"), "w"), indent=1)
    print(rep)
