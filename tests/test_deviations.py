"""How far the build-owned definitions move whole frames away from the reference's literal ones.

The reference cannot be built here and holds no vectors for this path, so these numbers cannot be checked against its
binary; they are measured between two evaluations inside the oracle (oracle/lv_oracle.cpp `g_dev`):

  * ray-capsule roots: closest-approach form (the build's) vs the textbook (-B -+ sqrt(B^2 - 4AC)) / 2A of
    RayIntersectionTestsVulkan.glsl:39-119 in float32 (the reference's);
  * AO lookup: the launching pixel's own texel (the build's, used for pixel-centre rays) vs project + bilinear texture()
    of AmbientOcclusion.glsl:84-99.

CPU tests run small frames; the `-m gpu` test renders BASELINE.json configs 2 and 3 at full size on the GPU box (HIP frame
vs the oracle's literal-form frame on the host cores) and writes the counts to gpurun_out/deviations.json.
"""
import json
import os

import numpy as np
import pytest

from common import ROOT, Case, small_case
from linevis_amd import scenes, transfer_function as tfm
from oracle import lvo

RTAO = dict(ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0, ambient_occlusion_gamma=1.0,
            ambient_occlusion_radius=0.1, ambient_occlusion_distance_based=True, use_jittered_primary_rays=True)


def diff_counts(a, b):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32)).max(axis=2)
    covered = int(((a[..., :3] != a[0, 0, :3]).any(axis=2) | (b[..., :3] != b[0, 0, :3]).any(axis=2)).sum())
    return {"pixels": int(d.size), "covered": covered, "differ": int((d > 0).sum()), "differ_gt_2lsb": int((d > 2).sum()),
            "max_lsb": int(d.max())}


def test_literal_roots_bvh_equals_brute_force_and_switches_reset():
    """In literal mode the traversal culls against best + r, so it still returns the brute-force minimum of the noisy roots;
    leaving the context restores the default evaluation."""
    c = small_case(width=96, height=64, line_width=0.004, intersection_form="closest_approach")
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    base = sc.render_rt(P, use_bvh=True)
    with lvo.deviation_switches(literal_intersection=True):
        lit_bvh = sc.render_rt(P, use_bvh=True)
        lit_brute = sc.render_rt(P, use_bvh=False)
    assert np.array_equal(lit_bvh, lit_brute)
    assert np.array_equal(sc.render_rt(P, use_bvh=True), base)
    assert not np.array_equal(lit_bvh, base)


@pytest.mark.parametrize("line_width,max_frac_gt2", [(0.02, 0.02), (0.004, 0.15), (0.002, 0.6)])
def test_literal_roots_small_frames(line_width, max_frac_gt2):
    """The textbook float32 discriminant is a difference of O(1) numbers with an O(r^2) result: the thinner the tube, the
    more of its silhouette and shading is float32 noise.  Measured (320 x 240, 3540 segments): > 2 LSB on 0.7 % of the covered
    pixels at line width 0.02, 7.6 % at 0.004, 40 % at 0.002 -- the reference's default width.  The bounds pin the order
    of magnitude; the point is the number, not the assert."""
    c = small_case(width=320, height=240, n_lines=60, pts_per_line=60, line_width=line_width, intersection_form="closest_approach")
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    a = sc.render_rt(P, use_bvh=True)
    with lvo.deviation_switches(literal_intersection=True):
        b = sc.render_rt(P, use_bvh=True)
    n = diff_counts(a, b)
    print("literal vs closest-approach, line width %g: %s" % (line_width, n))
    assert n["covered"] > 2000
    assert 0 < n["differ_gt_2lsb"] <= max_frac_gt2 * n["covered"]


def test_ao_lookup_pixel_centre_rays_equal_the_projected_bilinear_sample():
    """Without jitter the hit projects onto its pixel's texel centre (to float32 rounding), so reading the launching pixel's
    texel equals project + bilinear texture() to within one RGBA8 step."""
    c = small_case(width=320, height=240, n_lines=60, pts_per_line=60, line_width=0.02, **RTAO,
                   ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=16)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    ao = sc.render_ao(P, use_bvh=True)
    a = sc.render_rt(P, ao=ao, use_bvh=True)
    with lvo.deviation_switches(reference_ao_lookup=True):
        b = sc.render_rt(P, ao=ao, use_bvh=True)
    n = diff_counts(a, b)
    print("AO lookup, pixel-centre rays: %s" % n)
    assert n["max_lsb"] <= 1 and n["differ"] < 0.001 * n["pixels"]


@pytest.mark.gpu
def test_literal_intersection_form_in_hip_matches_the_oracle(hip_lib):
    """intersection_form = literal: the HIP path evaluates the reference's textbook roots (plus the own-box rule that makes
    them BVH-independent); hits, AO factors and frames against the oracle's literal mode -- brute force over all segments."""
    c = small_case(width=160, height=120, n_lines=40, pts_per_line=40, line_width=0.004, intersection_form="literal", **RTAO,
                   ambient_occlusion_iterations=2, ambient_occlusion_samples_per_frame=8)
    ctx = c.hip_context()
    img = ctx.render(11)
    ao = ctx.get_ao()
    rng = np.random.default_rng(5)
    o = np.concatenate([np.tile(np.array([[0, 0, 0.8]], np.float32), (3000, 1)), rng.uniform(-0.2, 0.2, (3000, 3)).astype(np.float32)])
    d = rng.normal(size=(6000, 3)).astype(np.float32)
    d[:3000, 2] = -np.abs(d[:3000, 2]) * 3
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t, s, k = ctx.trace_rays(o, d, 1e-4, 1000.0)
    sc = c.oracle_scene()
    with lvo.deviation_switches(literal_intersection=True):
        ref, ao_ref = c.oracle_render(11)                                   # brute force
        t2, s2, k2 = sc.trace_rays(o, d, 1e-4, 1000.0, c.line_width)        # brute force
    assert np.array_equal(t.view(np.uint32), t2.view(np.uint32)) and np.array_equal(s, s2) and np.array_equal(k, k2)
    assert (s != 0xFFFFFFFF).sum() > 500
    assert np.array_equal(ao.view(np.uint32), ao_ref.view(np.uint32))
    assert np.abs(img.astype(np.int32) - ref.astype(np.int32)).max() <= 2
    ctx.set_option("intersection_form", "closest_approach")
    assert not np.array_equal(ctx.render(11), img)


@pytest.mark.gpu
def test_full_size_deviation_report(hip_lib):
    """BASELINE.json configs 2 and 3 at 1920 x 1080: the HIP frame (closest-approach roots) against the oracle's frame with the
    reference's literal roots.  Writes gpurun_out/deviations.json (copied to profiles/ by hand)."""
    from linevis_amd import host_api
    report = {}
    for name, gen, settings in (("c2_helix_primary_only", scenes.helix_bundle, {}),
                                ("c3_tornado_rtao64_capsules", scenes.tornado,
                                 dict(RTAO, ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=64))):
        tr = scenes.normalize(gen())
        flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
        pts, seg, _ = flow.tube_aabb_render_data(0.002)
        c = Case(pts, seg, tfm.standard(), 1920, 1080, 0.002, **settings)
        ctx = c.hip_context()
        lo, hi = flow.attribute_range()
        ctx.set_transfer_function(c.tf, lo, hi)
        img = ctx.render(11)
        sc = c.oracle_scene()
        P = c.oracle_params(sc)
        P.attrMin, P.attrMax = lo, hi
        with lvo.deviation_switches(literal_intersection=True):
            ao = sc.render_ao(P, use_bvh=True) if P.useAmbientOcclusion else None
            ref = sc.render_rt(P, ao=ao, use_bvh=True)
        report[name] = diff_counts(img, ref)
        assert report[name]["covered"] > 50000
        # the same formula on both sides: what is left is the pipeline, not the formula
        ctx.set_option("intersection_form", "literal")
        report[name + "__hip_literal_vs_oracle_literal"] = diff_counts(ctx.render(11), ref)
        assert report[name + "__hip_literal_vs_oracle_literal"]["differ_gt_2lsb"] == 0
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    report["what"] = ("HIP frame (closest-approach ray-capsule roots) vs oracle frame evaluated with the reference's literal "
                      "float32 roots (RayIntersectionTestsVulkan.glsl:39-119), 1920x1080, line width 0.002")
    json.dump(report, open(os.path.join(out, "deviations.json"), "w"), indent=1)
    print(report)
