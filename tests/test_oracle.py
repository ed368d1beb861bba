"""CPU tests of the oracle (the checker) against the committed golden vectors and against independent
restatements.  PARITY UNPINNED: the vectors were produced by this oracle (tests/golden/make_golden.py); the
reference holds none for this path."""
import os

import numpy as np
import pytest

from common import GOLDEN_DIR, Case, small_case, max_lsb_diff, scene_arrays
from linevis_amd import scenes, transfer_function as tfm
from oracle import lvo


def G(name):
    return np.load(os.path.join(GOLDEN_DIR, name))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# ---------------------------------------------------------------- a12 RNG
def py_tea(v0, v1):
    """Independent restatement of RayTracingUtilities.glsl:134-148 in pure Python."""
    M = 0xFFFFFFFF
    s0 = 0
    for _ in range(16):
        s0 = (s0 + 0x9e3779b9) & M
        v0 = (v0 + ((((v1 << 4) & M) + 0xa341316c) & M ^ ((v1 + s0) & M) ^ (((v1 >> 5) + 0xc8013ea4) & M))) & M
        v1 = (v1 + ((((v0 << 4) & M) + 0xad90777d) & M ^ ((v0 + s0) & M) ^ (((v0 >> 5) + 0x7e95761e) & M))) & M
    return v0


def py_rnd(seed, n):
    out = []
    for _ in range(n):
        seed = (1664525 * seed + 1013904223) & 0xFFFFFFFF
        out.append(np.float32(seed & 0x00FFFFFF) / np.float32(0x01000000))
    return np.array(out, dtype=np.float32)


def test_rng_known_answers():
    g = G("rng_kat.npz")
    for (a, b), want in zip(g["tea_in"], g["tea_out"]):
        assert lvo.tea(int(a), int(b)) == int(want) == py_tea(int(a), int(b))
    for seed, want in zip(g["rnd_seeds"], g["rnd_bits"]):
        got = lvo.rnd_sequence(int(seed), 8)
        assert np.array_equal(bits(got), want)
        assert np.array_equal(bits(py_rnd(int(seed), 8)), want)
        assert np.all((got >= 0) & (got < 1))


def test_sincos_2pi_definition():
    g = G("rng_kat.npz")
    got = np.array([lvo.sincos_2pi(float(x)) for x in g["sincos_xi"]], dtype=np.float32)
    assert np.array_equal(bits(got), g["sincos_bits"])
    xi = (np.arange(0, 1 << 24, 4099, dtype=np.int64) / float(1 << 24)).astype(np.float32)
    sc = np.array([lvo.sincos_2pi(float(x)) for x in xi], dtype=np.float64)
    ref = np.stack([np.sin(2 * np.pi * xi.astype(np.float64)), np.cos(2 * np.pi * xi.astype(np.float64))], axis=1)
    assert np.abs(sc - ref).max() < 3e-7   # well inside GLSL's implementation-defined sin/cos precision
    assert np.abs(sc[:, 0] ** 2 + sc[:, 1] ** 2 - 1).max() < 5e-7


def test_mat4_inverse():
    rng = np.random.default_rng(3)
    for _ in range(20):
        m = rng.normal(size=(4, 4)).astype(np.float32)
        inv = lvo.mat4_inverse(m.T.reshape(16)).reshape(4, 4).T  # column-major in/out
        assert np.allclose(inv @ m, np.eye(4), atol=2e-4)


# ---------------------------------------------------------------- a7 intersection
def dist_to_segment(p, a, b):
    ab = b - a
    t = np.clip(np.dot(p - a, ab) / np.dot(ab, ab), 0, 1)
    return np.linalg.norm(p - (a + t * ab))


def test_capsule_known_answers():
    g = G("capsule_kat.npz")
    n = len(g["r"])
    assert n >= 256
    hits = {0: 0, 1: 0, 2: 0}
    for capped in (0, 1):
        for i in range(n):
            h, t, k = lvo.intersect_capsule(g["o"][i], g["d"][i], g["p0"][i], g["p1"][i], float(g["r"][i]), bool(capped))
            assert int(h) == int(g["hit%d" % capped][i])
            if h:
                assert np.float32(t).view(np.uint32) == g["t_bits%d" % capped][i]
                assert k == g["kind%d" % capped][i]
                assert t >= 0
                if capped:
                    hits[k] += 1
                    # the hit point lies on the surface of the reported primitive (cylinder / sphere p0 / sphere p1)
                    p = g["o"][i].astype(np.float64) + t * g["d"][i].astype(np.float64)
                    a, b = g["p0"][i].astype(np.float64), g["p1"][i].astype(np.float64)
                    if k == 0:
                        ax = (b - a) / np.linalg.norm(b - a)
                        d = np.linalg.norm((p - a) - np.dot(p - a, ax) * ax)
                        assert 0 < np.dot(p - a, ax) and np.dot(p - b, ax) < 0
                    else:
                        d = np.linalg.norm(p - (a if k == 1 else b))
                    assert abs(d - g["r"][i]) < 2e-4 * max(1.0, np.linalg.norm(g["d"][i]))
    assert min(hits.values()) > 10  # cylinder, cap 0 and cap 1 are all exercised


def f64_capsule(o, d, p0, p1, r):
    """float64 evaluation of the reference's IntersectionTube (RayIntersectionTestsVulkan.glsl + TubeRayTracing.glsl:452-494)."""
    o, d, p0, p1 = [np.asarray(v, dtype=np.float64) for v in (o, d, p0, p1)]
    best, kind = None, 0
    td = (p1 - p0) / np.linalg.norm(p1 - p0)
    av = d - np.dot(d, td) * td
    cv = (o - p0) - np.dot(o - p0, td) * td
    A, B, C = av @ av, 2 * (av @ cv), cv @ cv - r * r
    disc = B * B - 4 * A * C
    if disc >= 0 and A > 0:
        for t in ((-B - np.sqrt(disc)) / (2 * A), (-B + np.sqrt(disc)) / (2 * A)):
            ip = o + t * d
            if t >= 0 and td @ (ip - p0) > 0 and td @ (ip - p1) < 0:
                best, kind = t, 0
                break
    for k, c in ((1, p0), (2, p1)):
        f = o - c
        A, B, C = d @ d, 2 * (d @ f), f @ f - r * r
        disc = B * B - 4 * A * C
        if disc < 0:
            continue
        t0, t1 = (-B - np.sqrt(disc)) / (2 * A), (-B + np.sqrt(disc)) / (2 * A)
        t = t0 if t0 >= 0 else (t1 if t1 >= 0 else None)
        if t is not None and (best is None or t < best):
            best, kind = t, k
    return best, kind


def test_closest_approach_form_tracks_float64_truth():
    """The build evaluates the reference's quadratics in closest-approach form; against a float64 evaluation of the
    reference formulas it is ~1000x closer than the literal float32 form for thin tubes far from the ray origin."""
    rng = np.random.default_rng(42)
    err_stable, err_literal, n = [], [], 0
    for _ in range(4000):
        p0 = rng.uniform(-0.25, 0.25, 3)
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        p1 = p0 + ax * rng.uniform(0.001, 0.004)
        r = 0.001
        o = np.array([0.0, 0.0, 0.8]) + rng.normal(size=3) * 0.05
        perp = np.cross(ax, rng.normal(size=3)); perp /= np.linalg.norm(perp)
        target = 0.5 * (p0 + p1) + perp * r * rng.uniform(-0.95, 0.95)
        d = (target - o) / np.linalg.norm(target - o)
        o32, d32, a32, b32 = [np.asarray(v, dtype=np.float32) for v in (o, d, p0, p1)]
        truth, tk = f64_capsule(o32, d32, a32, b32, np.float32(r))
        hs, ts, ks = lvo.intersect_capsule(o32, d32, a32, b32, r, True)
        hl, tl, kl = lvo.intersect_capsule(o32, d32, a32, b32, r, True, literal=True)
        if truth is None or not hs or not hl:
            continue
        n += 1
        err_stable.append(abs(ts - truth))
        err_literal.append(abs(tl - truth))
    assert n > 3000
    assert np.max(err_stable) < 2e-5 and np.median(err_stable) < 2e-7
    assert np.median(err_literal) > 20 * np.median(err_stable)


def test_closest_approach_form_agrees_with_literal_form():
    g = G("capsule_kat.npz")
    agree = 0
    for i in range(len(g["r"])):
        a = lvo.intersect_capsule(g["o"][i], g["d"][i], g["p0"][i], g["p1"][i], float(g["r"][i]), True)
        b = lvo.intersect_capsule(g["o"][i], g["d"][i], g["p0"][i], g["p1"][i], float(g["r"][i]), True, literal=True)
        if a[0] != b[0]:
            continue   # only at grazing incidence, where the literal discriminant has no correct digit left
        agree += 1
        if a[0]:
            assert abs(a[1] - b[1]) < 5e-4 * max(1.0, float(np.linalg.norm(g["d"][i])))
    assert agree >= len(g["r"]) - 6


def test_capsule_degenerate_inputs():
    # ray parallel to the axis and outside: no cylinder hit, only caps if inside their radius
    h, t, k = lvo.intersect_capsule([0, 0.5, -1], [0, 0, 1], [0, 0, 0], [0, 0, 1], 0.1, True)
    assert not h
    # zero-length segment: normalize(0) is NaN in the cylinder test, the spheres still answer
    h, t, k = lvo.intersect_capsule([0, 0, -1], [0, 0, 1], [0, 0, 0], [0, 0, 0], 0.1, True)
    assert h and k in (1, 2) and abs(t - 0.9) < 1e-6
    # origin inside: the exit root is reported (t >= 0)
    h, t, k = lvo.intersect_capsule([0, 0, 0.5], [1, 0, 0], [0, 0, 0], [0, 0, 1], 0.1, True)
    assert h and k == 0 and abs(t - 0.1) < 1e-6


# ---------------------------------------------------------------- a1/a2 geometry preparation
def test_normalize_positions():
    tr = scenes.random_curves(n_lines=5, points_per_line=9, seed=2)
    p = lvo.normalize_positions(tr.positions * 3.0 + 1.5)
    ext = p.max(axis=0) - p.min(axis=0)
    assert abs(ext.max() - 0.5) < 1e-6
    k = int(np.argmax(ext))
    assert abs(p.max(axis=0)[k] + p.min(axis=0)[k]) < 1e-6
    assert np.array_equal(p, scenes.normalize(scenes.Trajectories(tr.positions * 3.0 + 1.5, tr.attributes, tr.line_offsets)).positions)


def test_a2_render_data_golden_and_properties():
    g = G("a2_cases.npz")
    pts, seg, aabb = lvo.build_tube_aabb_render_data(g["positions"], g["attributes"], g["line_offsets"], float(g["line_width"]))
    assert np.array_equal(pts.view(np.uint8).reshape(-1, 48), g["points"])
    assert np.array_equal(seg, g["seg"])
    assert np.array_equal(bits(aabb), g["aabb_bits"])
    # 8 input lines: three are dropped entirely (single point, coincident pair, one valid point)
    t = pts["lineTangent"].astype(np.float64)
    n = pts["lineNormal"].astype(np.float64)
    assert np.allclose(np.linalg.norm(t, axis=1), 1, atol=1e-5)
    assert np.allclose(np.linalg.norm(n, axis=1), 1, atol=1e-5)
    assert np.abs((t * n).sum(axis=1)).max() < 1e-5
    # segments never connect different lines: consecutive indices
    assert np.all(seg[:, 1] == seg[:, 0] + 1)
    r = float(g["line_width"]) * 0.5
    p = pts["linePosition"]
    assert np.allclose(aabb[:, :3], np.minimum(p[seg[:, 0]], p[seg[:, 1]]) - r)
    assert np.allclose(aabb[:, 3:], np.maximum(p[seg[:, 0]], p[seg[:, 1]]) + r)


def test_a2_empty_and_ragged():
    pts, seg, aabb = lvo.build_tube_aabb_render_data(np.zeros((0, 3), np.float32), np.zeros(0, np.float32),
                                                     np.zeros(1, np.uint32), 0.01)
    assert len(pts) == 0 and len(seg) == 0
    pos = np.array([[0, 0, 0], [0.1, 0, 0], [0.5, 0.5, 0.5], [0, 0.1, 0], [0, 0.2, 0.01], [0, 0.3, 0]], np.float32)
    off = np.array([0, 2, 3, 6], np.uint32)  # lines of 2, 1 and 3 points
    pts, seg, _ = lvo.build_tube_aabb_render_data(pos, np.arange(6, dtype=np.float32), off, 0.01)
    assert len(pts) == 5 and len(seg) == 3
    assert seg.tolist() == [[0, 1], [2, 3], [3, 4]]


# ---------------------------------------------------------------- BVH vs brute force
def test_bvh_equals_brute_force_rays_and_images():
    c = small_case(width=64, height=48, n_lines=40, pts_per_line=30, seed=21, line_width=0.012)
    sc = c.oracle_scene()
    rng = np.random.default_rng(8)
    o = rng.uniform(-0.4, 0.4, (3000, 3)).astype(np.float32)
    d = rng.normal(size=(3000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a = sc.trace_rays(o, d, 0.0, 0.3, c.line_width, use_bvh=False)
    b = sc.trace_rays(o, d, 0.0, 0.3, c.line_width, use_bvh=True)
    assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert (a[1] != 0xFFFFFFFF).sum() > 200
    for mode in (11, 2):
        cc = small_case(width=64, height=48, n_lines=40, pts_per_line=30, seed=21, line_width=0.012,
                        transparent=(mode == 2), ambient_occlusion_mode="RTAO (Screen Space)",
                        ambient_occlusion_strength=1.0, ambient_occlusion_iterations=1,
                        ambient_occlusion_samples_per_frame=4)
        i0, ao0 = cc.oracle_render(mode, use_bvh=False)
        i1, ao1 = cc.oracle_render(mode, use_bvh=True)
        assert np.array_equal(i0, i1) and np.array_equal(bits(ao0), bits(ao1))


def test_closest_hit_tie_goes_to_lowest_segment():
    # two collinear segments share an end point: the shared sphere is hit at exactly the same t by both
    pts = np.zeros(3, dtype=lvo.LINE_POINT_DTYPE)
    pts["linePosition"] = [[-0.1, 0, 0], [0, 0, 0], [0.1, 0.02, 0]]
    seg = np.array([[0, 1], [1, 2]], np.uint32)
    sc = lvo.Scene(pts, seg, tfm.standard())
    o = np.array([[0.0, -0.5, 0.0]], np.float32)
    d = np.array([[0.0, 1.0, 0.0]], np.float32)
    for use_bvh in (False, True):
        t, s, k = sc.trace_rays(o, d, 1e-4, 10.0, 0.02, use_bvh=use_bvh)
        assert s[0] == 0 and k[0] == 2


# ---------------------------------------------------------------- golden frames (regression pins)
def test_golden_small_scene_frames():
    g = G("scene_small.npz")
    pts = g["points"].reshape(-1).view(lvo.LINE_POINT_DTYPE)
    W, H, lw = int(g["width"]), int(g["height"]), float(g["line_width"])
    base = Case(pts, g["seg"], g["tf"], W, H, lw)
    t, s, k = base.oracle_scene().trace_rays(g["ray_o"], g["ray_d"], 1e-4, 1000.0, lw)
    assert np.array_equal(bits(t), g["ray_t_bits"]) and np.array_equal(s, g["ray_seg"]) and np.array_equal(k, g["ray_kind"])
    c = Case(pts, g["seg"], g["tf"], W, H, lw, depth_cue_strength=0.8)
    assert max_lsb_diff(c.oracle_render(11)[0], g["rt_depthcue"]) <= 1   # pow() goes through libm
    assert np.array_equal(bits(c.oracle_scene().depth_range(c.oracle_params())), g["depth_range_bits"])
    c = Case(pts, g["seg"], g["tf_transparent"], W, H, lw, num_samples_per_frame=4)
    assert max_lsb_diff(c.oracle_render(11)[0], g["rt_transparent_spp4"]) <= 1
    c = Case(pts, g["seg"], g["tf"], W, H, lw, ambient_occlusion_mode="RTAO (Screen Space)",
             ambient_occlusion_strength=0.9, ambient_occlusion_gamma=1.5, ambient_occlusion_iterations=2,
             ambient_occlusion_samples_per_frame=8, ambient_occlusion_radius=0.1)
    img, ao = c.oracle_render(11)
    assert np.array_equal(bits(ao), g["ao_bits"])       # pure +,-,*,/,sqrt: exact
    assert max_lsb_diff(img, g["rt_ao"]) <= 1
    assert 0.0 <= ao.min() < 0.9 and ao.max() == 1.0
    c = Case(pts, g["seg"], g["tf_transparent"], W, H, lw, ppll_fragment_source="capsule_entry")   # the fixture of rounds 1-3
    st = lvo.Stats()
    img, _ = c.oracle_render(2, stats=st)
    assert max_lsb_diff(img, g["ppll"]) <= 1
    assert st.fragments == int(g["ppll_fragments"]) and st.maxDepthComplexity == int(g["ppll_max_depth_complexity"])


def test_golden_lattice_c1_and_equal_means():
    """Config 1 analogue (test/TestVolumetricPathTracing.cpp:44-237 harness shape): 128x128, 4 spp; two independent
    estimators (different sample sets) agree in the image mean within 2e-3 per channel, as the reference's test does."""
    g = G("lattice_c1.npz")
    tr = scenes.normalize(scenes.lattice())
    pts, seg = scene_arrays(tr, float(g["line_width"]))
    assert len(pts) == int(g["num_points"]) and len(seg) == int(g["num_segments"]) == 31744
    assert np.uint32(np.bitwise_xor.reduce(pts.view(np.uint32))) == g["points_crc"]
    c = Case(pts, seg, tfm.standard(), 128, 128, float(g["line_width"]), num_samples_per_frame=4)
    img, _ = c.oracle_render(11, use_bvh=True)
    assert max_lsb_diff(img, g["image"]) <= 1
    c2 = Case(pts, seg, tfm.standard(), 128, 128, float(g["line_width"]), num_samples_per_frame=16)
    img2, _ = c2.oracle_render(11, use_bvh=True)
    m1 = img.reshape(-1, 4).astype(np.float64).mean(axis=0) / 255.0
    m2 = img2.reshape(-1, 4).astype(np.float64).mean(axis=0) / 255.0
    assert np.abs(m1 - m2).max() < 2e-3 + 2.0 / 255.0 / 2.0


def test_tiles_reproduce_full_frame():
    c = small_case(width=80, height=56, num_samples_per_frame=2, ambient_occlusion_mode="RTAO (Screen Space)",
                   ambient_occlusion_strength=1.0, ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=4)
    full, ao_full = c.oracle_render(11)
    tile = (24, 16, 40, 24)
    part, ao_part = c.oracle_render(11, tile=tile)
    x0, y0, w, h = tile
    assert np.array_equal(part, full[y0:y0 + h, x0:x0 + w])
    assert np.array_equal(bits(ao_part[y0:y0 + h, x0:x0 + w]), bits(ao_full[y0:y0 + h, x0:x0 + w]))


# ---------------------------------------------------------------- PPLL
def test_ppll_address_generation():
    # TiledAddress.glsl:53-85: a bijection onto [0, padded_w * padded_h)
    for (tw, th, w, h) in [(2, 8, 10, 9), (1, 1, 7, 5), (2, 2, 6, 6), (8, 8, 20, 12), (4, 4, 8, 8)]:
        pw, ph = -(-w // tw) * tw, -(-h // th) * th
        seen = {lvo.ppll_addr(x, y, pw, tw, th) for y in range(ph) for x in range(pw)}
        assert seen == set(range(pw * ph))
    assert lvo.ppll_addr(3, 9, 10, 2, 8) == ((3 // 2) + (10 // 2) * (9 // 8)) * 16 + (3 & 1) + (9 & 7) * 2


def test_ppll_golden_lists_and_literal_variant():
    g = G("ppll_lists.npz")
    from linevis_amd import camera
    W, H = int(g["width"]), int(g["height"])
    view, proj, *_ = camera.default_camera(W, H)
    P = lvo.make_params(view, proj, W, H, ppllMaxNumFrags=int(g["max_frags"]), ppllLinkedListSize=len(g["nodes"]),
                        background=tuple(g["background"]))
    assert np.array_equal(lvo.ppll_resolve(P, g["nodes"], g["start"], literal=False), g["resolved_key"])
    assert np.array_equal(lvo.ppll_resolve(P, g["nodes"], g["start"], literal=True), g["resolved_literal"])
    # empty pixels show the background
    bg = np.floor(np.clip(g["background"], 0, 1) * 255 + 0.5).astype(np.uint8)
    assert np.array_equal(g["resolved_key"][0, 0], bg)
    # without depth ties the (depth, colour) key order equals the reference's depth-only order
    nodes = g["nodes"].copy()
    nodes[:, 1] = (np.float32(0.2) + np.arange(len(nodes), dtype=np.float32) * np.float32(1e-3)).view(np.uint32)
    a = lvo.ppll_resolve(P, nodes, g["start"], literal=False)
    b = lvo.ppll_resolve(P, nodes, g["start"], literal=True)
    assert np.array_equal(a, b)


def test_ppll_gather_overflow_and_discard():
    c = small_case(width=48, height=32, transparent=True)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    P.ppllLinkedListSize = 300
    st = lvo.Stats()
    nodes, start, cnt = sc.ppll_gather(P, stats=st)
    assert cnt > P.ppllLinkedListSize          # pool overflow: counter keeps counting, stores stop
    stored = start[start != 0xFFFFFFFF]
    assert stored.max() < P.ppllLinkedListSize
    img = lvo.ppll_resolve(P, nodes, start)
    assert img.shape == (32, 48, 4)


def test_multi_frame_accumulation_round_trips_through_rgba8():
    """TubeRayTracing.glsl:268-273: frame f is mixed into the previous rgba8 frame with weight 1 / (f + 1).  With one
    identical sample per frame (deterministic sampling) the running mean of identical frames is a fixed point of the
    quantised recursion; with per-pixel seeds the chain converges towards the multi-sample frame."""
    from common import small_case
    c = small_case(width=64, height=48, num_accumulated_frames=6, num_samples_per_frame=1)
    frames = c.oracle_render_progressive(6)
    many = small_case(width=64, height=48, num_samples_per_frame=6)
    sc = many.oracle_scene()
    target = sc.render_rt(many.oracle_params(sc))
    err = [np.abs(f.astype(np.int32) - target.astype(np.int32)).mean() for f in frames]
    assert err[5] < err[0] and err[5] < 3.0
    # frame 0 is the plain jittered frame
    c0 = small_case(width=64, height=48, num_accumulated_frames=2, num_samples_per_frame=1)
    sc0 = c0.oracle_scene()
    P0 = c0.oracle_params(sc0)
    assert P0.useJitteredRays == 1 and np.array_equal(frames[0], sc0.render_rt(P0))
    # mixing a frame with itself is a fixed point: mix(q, x, a) with x == dequantised q re-quantises to q
    P0.frameNumber = 0
    again = sc0.render_rt(P0, prev=frames[0])
    assert np.array_equal(again, frames[0])


def test_pow_det_against_float64():
    """The build-owned pow of the shading code (powDet / lv_pow_det: exp2(y log2 x) with fixed float32 series): relative error against
    float64 pow below 3e-6 wherever the result exceeds 1e-4, absolute error below 1e-9 below that; exact special values."""
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(0.0, 1.0, 200000), 10.0 ** rng.uniform(-30, 0, 50000), rng.uniform(1.0, 50.0, 20000),
                        np.array([1.0, 0.5, 0.25, 0.70710678, 1.41421356, 2.0])]).astype(np.float32)
    for y in (1.0, 1.7, 30.0, 0.45, 2.2, 3.0, 0.0, 128.0, 0.01):
        got = lvo.pow_det(x, np.float32(y)).astype(np.float64)
        want = np.power(x.astype(np.float64), float(np.float32(y)))
        big = (want > 1e-4) & (want < 1e4)      # (the error grows with |y log2 x|: 6e-6 at results of 1e24, never met by the shading)
        assert big.sum() > 1000 and np.abs(got[big] / want[big] - 1.0).max() < 3e-6, y
        small = want <= 1e-4
        assert not small.any() or np.abs(got[small] - want[small]).max() < 1e-9, y
    assert lvo.pow_det(np.float32(0.0), np.float32(1.7))[0] == 0.0 and lvo.pow_det(np.float32(0.0), np.float32(0.0))[0] == 1.0
    assert lvo.pow_det(np.float32(1.0), np.float32(30.0))[0] == 1.0
    assert lvo.pow_det(np.float32(1e-30), np.float32(30.0))[0] == 0.0


def test_shading_normalisations_of_the_test_scenes_stay_inside_the_clamp_range():
    """normalize() of the shading code is v / sqrt(clamp(v . v, 2^-60, 2^60)) on both sides -- where the clamp acts it is
    not the reference's normalize() (GLSL: NaN / Inf).  The checker counts those calls: on the scenes the parity tests compare
    (ray tracer with RTAO and depth cues, both PPLL fragment sources, band data) there are none, so device == checker there is also
    device == the reference's formula; a zero vector is counted."""
    lvo.shade_normalize_out_of_range(reset=True)
    rtao = dict(ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0, ambient_occlusion_iterations=1,
                ambient_occlusion_samples_per_frame=4, depth_cue_strength=0.7)
    for mode, transparent, settings in [(11, False, rtao), (11, True, {}), (2, True, dict(ppll_fragment_source="raster_prism")),
                                        (2, True, dict(ppll_fragment_source="capsule_entry"))]:
        c = small_case(width=96, height=64, transparent=transparent, **settings)
        img, _ = c.oracle_render(mode, use_bvh=True)
        assert (img[..., :3] != 255).any(axis=2).sum() > 200
    assert lvo.shade_normalize_out_of_range(reset=True) == 0
    # the counter sees what it should: a degenerate segment (p0 == p1) shades with a zero tangent
    pts = np.zeros(2, dtype=lvo.LINE_POINT_DTYPE)
    pts["linePosition"] = [[0.0, 0.0, 0.0], [0.0, 0.0, 0.0]]
    pts["lineTangent"] = [[1.0, 0.0, 0.0], [1.0, 0.0, 0.0]]
    pts["lineNormal"] = [[0.0, 1.0, 0.0], [0.0, 1.0, 0.0]]
    c = Case(pts, np.array([[0, 1]], np.uint32), tfm.standard(), 64, 48, 0.2)
    c.oracle_render(11)
    assert lvo.shade_normalize_out_of_range(reset=True) > 0
