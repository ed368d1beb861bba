"""Band data (ribbons) on the CPU side: .binlines version 2, LineDataFlow's elliptic-tube render data against the oracle, the
build-owned trigonometric functions, and the oracle's elliptic-tube intersection against an independent float64 restatement of
EllipticTubeRayTracing.glsl."""
import ctypes as C
import os

import numpy as np

from common import Case
from linevis_amd import camera, host_api, scenes, transfer_function as tfm
from oracle import lvo


def ribbon_scene(n_lines=6, pts=60, twist=8.0, seed=3):
    return scenes.twisted_ribbons(scenes.normalize(scenes.helix_bundle(n_lines=n_lines, points_per_line=pts, seed=seed, turns=2.0)),
                                  twist=twist)


def test_binlines_v2_round_trip_and_writers_agree(tmp_path):
    tr = ribbon_scene()
    a, b = str(tmp_path / "a.binlines"), str(tmp_path / "b.binlines")
    scenes.write_binlines(a, tr)
    fl = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets, tr.ribbon_directions)
    fl.save_binlines(b)
    assert open(a, "rb").read() == open(b, "rb").read()
    back = scenes.read_binlines(b)
    assert np.array_equal(back.ribbon_directions, tr.ribbon_directions) and np.array_equal(back.positions, tr.positions)
    f2 = host_api.LineDataFlow().load_binlines(a)            # the loader normalises positions; the directions are kept as stored
    assert f2.has_bands_data and np.array_equal(f2.ribbon_directions(), tr.ribbon_directions)
    plain = scenes.Trajectories(tr.positions, tr.attributes, tr.line_offsets)
    scenes.write_binlines(a, plain)                            # version 1 stays version 1
    assert open(a, "rb").read()[:4] == bytes([1, 0, 0, 0]) and scenes.read_binlines(a).ribbon_directions is None
    assert not host_api.LineDataFlow().load_binlines(a).has_bands_data


def test_elliptic_render_data_host_equals_oracle():
    tr = ribbon_scene()
    fl = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets, tr.ribbon_directions)
    for bw in (0.05, 0.005):
        a = fl.tube_aabb_render_data_elliptic(bw)
        b = lvo.build_tube_aabb_render_data_ribbons(tr.positions, tr.attributes, tr.line_offsets, bw, tr.ribbon_directions)
        assert all(np.array_equal(x.view(np.uint8), y.view(np.uint8)) for x, y in zip(a, b))
        pts, seg, aabb = a
        # normal = cross(ribbon direction, tangent); boxes = segment extent + band width / 2
        n = np.cross(tr.ribbon_directions, pts["lineTangent"])
        assert np.abs(n - pts["lineNormal"]).max() < 1e-6
        p0, p1 = pts["linePosition"][seg[:, 0]], pts["linePosition"][seg[:, 1]]
        assert np.allclose(aabb[:, :3], np.minimum(p0, p1) - bw / 2, atol=1e-7)
    # without band data / with ribbons switched off the elliptic request degrades to the line-width capsule data
    c = fl.tube_aabb_render_data(0.01)
    assert not np.array_equal(c[0]["lineNormal"], a[0]["lineNormal"])


def test_build_owned_trigonometry_is_accurate():
    L = lvo.lib()
    rng = np.random.default_rng(2)
    s, c = C.c_float(), C.c_float()
    worst = 0.0
    for a in np.concatenate([rng.uniform(-8, 8, 4000), [0.0, np.pi, -np.pi, 2 * np.pi, -1e-8, 1e-8]]).astype(np.float32):
        L.lvo_sincos_rad(float(a), C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - np.sin(np.float64(a))), abs(c.value - np.cos(np.float64(a))))
    assert worst < 1.5e-6
    worst = 0.0
    for y, x in rng.normal(size=(4000, 2)).astype(np.float32).tolist() + [[0.0, 1.0], [0.0, -1.0], [1.0, 0.0], [-1.0, 0.0], [0.0, 0.0]]:
        worst = max(worst, abs(L.lvo_atan2_det(y, x) - np.arctan2(np.float64(y), np.float64(x))))
    assert worst < 5e-7


# ------------------------------------------------------------------ float64 restatement of IntersectionEllipticTube
def _rot_cos(axis, c):
    s = np.sqrt(1.0 - c * c)
    a = axis / np.linalg.norm(axis)
    t = (1.0 - c) * a
    return np.array([[c + t[0] * a[0], t[0] * a[1] + s * a[2], t[0] * a[2] - s * a[1]],
                     [t[1] * a[0] - s * a[2], c + t[1] * a[1], t[1] * a[2] + s * a[0]],
                     [t[2] * a[0] + s * a[1], t[2] * a[1] - s * a[0], c + t[2] * a[2]]]).T      # columns as GLSL builds them


def intersect_elliptic_tube(o, d, lp0, lp1, band_width, min_band_thickness, cam):
    r0, r1 = band_width * 0.5 * min_band_thickness, band_width * 0.5
    p0, p1 = lp0["linePosition"].astype(np.float64), lp1["linePosition"].astype(np.float64)
    lo, hi = np.minimum(p0, p1) - r1, np.maximum(p0, p1) + r1
    t_near, t_far = -1e7, 1e7
    for i in range(3):
        if abs(d[i]) < 1e-3:
            if o[i] < lo[i] or o[i] > hi[i]:
                return None
        else:
            t0, t1 = sorted(((lo[i] - o[i]) / d[i], (hi[i] - o[i]) / d[i]))
            t_near, t_far = max(t_near, t0), min(t_far, t1)
            if t_near > t_far or t_far < 0:
                return None
    n0, n1 = lp0["lineNormal"].astype(np.float64), lp1["lineNormal"].astype(np.float64)
    tan0, tan1 = lp0["lineTangent"].astype(np.float64), lp1["lineTangent"].astype(np.float64)
    l = np.linalg.norm(p1 - p0)
    xt = (p1 - p0) / l
    R = np.eye(3)
    if abs(np.dot(n0, xt)) > 0.999:
        R = _rot_cos(np.cross(n0, xt), np.dot(n0, xt))
    yt, zt = R @ n0, R @ np.cross(tan0, n0)
    rho_r = -np.arctan2(np.dot(np.cross(n0, n1), tan0), np.dot(n0, n1))
    F = np.stack([xt, yt, zt], axis=1)
    p = F.T @ (o + t_near * d - p0)
    dd = F.T @ d
    hit_t = t_near
    for _ in range(80):
        t = np.clip(p[0] / l, 0.0, 1.0)
        phi = np.arctan2(p[2], p[1])
        a = phi + t * rho_r
        r = r0 * r1 / np.sqrt(r0 * r0 * np.sin(a) ** 2 + r1 * r1 * np.cos(a) ** 2)
        d_tmp = (np.hypot(p[1], p[2]) - r) * 0.25
        p = p + dd * d_tmp
        hit_t += d_tmp
        if d_tmp < 1e-5:
            break
    world = F @ p + p0
    unit = lambda v: v / np.linalg.norm(v)
    eps1 = abs(np.dot(tan0, unit(cam - p0))) * 5e-5
    eps2 = abs(np.dot(-tan1, unit(cam - p1))) * 5e-5
    ok = d_tmp < 1e-4 and hit_t > 0 and np.dot(tan0, world) - np.dot(tan0, p0) > -eps1 and np.dot(-tan1, world) + np.dot(tan1, p1) > -eps2
    return hit_t if ok else None


def test_elliptic_intersection_against_the_float64_restatement():
    tr = ribbon_scene(n_lines=3, pts=30)
    bw, mbt = 0.06, 0.3
    pts, seg, _ = lvo.build_tube_aabb_render_data_ribbons(tr.positions, tr.attributes, tr.line_offsets, bw, tr.ribbon_directions)
    sc = lvo.Scene(pts, seg, tfm.standard())
    cam = np.array([0.0, 0.0, 0.8])
    rng = np.random.default_rng(4)
    tgt = pts["linePosition"][rng.integers(0, len(pts), 300)].astype(np.float64) + rng.normal(scale=0.02, size=(300, 3))
    d = tgt - cam
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.tile(cam, (300, 1))
    t, s = sc.trace_rays_elliptic(o, d, 1e-4, 1000.0, bw, mbt, cam, use_bvh=True)
    d32, o32 = d.astype(np.float32).astype(np.float64), o.astype(np.float32).astype(np.float64)
    checked = 0
    for i in range(300):
        best = None
        for k in range(len(seg)):
            h = intersect_elliptic_tube(o32[i], d32[i], pts[seg[k, 0]], pts[seg[k, 1]], bw, mbt, cam)
            if h is not None and h >= 1e-4 and (best is None or h < best[0]):
                best = (h, k)
        if best is None:
            # a float64 miss may be a grazing float32 hit; only assert clear cases
            continue
        if s[i] == 0xFFFFFFFF:
            continue
        if s[i] == best[1]:
            assert abs(t[i] - best[0]) < 2e-5
            checked += 1
    hits = int((s != 0xFFFFFFFF).sum())
    assert hits > 120 and checked > 0.9 * hits


def test_elliptic_frame_bvh_equals_brute_force_and_bands_change_the_shading():
    tr = ribbon_scene()
    s = dict(use_ribbons=True, band_width=0.05, min_band_thickness=0.3, use_analytic_elliptic_tubes=True)
    pts, seg, _ = lvo.build_tube_aabb_render_data_ribbons(tr.positions, tr.attributes, tr.line_offsets, 0.05, tr.ribbon_directions)
    c = Case(pts, seg, tfm.standard(), 160, 120, 0.02, **s)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    a = sc.render_rt(P, use_bvh=False)
    b = sc.render_rt(P, use_bvh=True)
    assert np.array_equal(a, b) and (a[..., :3] != 255).any(axis=2).sum() > 1500
    P.minBandThickness = 1.0          # circular cross-section of the band radius: another picture
    assert not np.array_equal(sc.render_rt(P, use_bvh=True), a)


def test_band_halo_coordinate_against_a_geometric_construction():
    """The USE_BANDS halo coordinate (RayHitCommon.glsl:232-351: polar line of the camera point, degenerate conic, homogeneous
    cross products) against plain analytic geometry in float64: the two tangent points from the camera's projection to the ellipse,
    the chord between them, the point where the line camera -> fragment crosses that chord; |coordinate| = how far along the chord."""
    L = lvo.lib()
    f3 = C.c_float * 3
    L.lvo_bands_ribbon_position.restype = C.c_float
    L.lvo_bands_ribbon_position.argtypes = [f3, f3, f3, f3, C.c_float, C.c_float, C.c_float]
    rng = np.random.default_rng(12)
    checked, worst = 0, 0.0
    for _ in range(4000):
        t = rng.normal(size=3); t /= np.linalg.norm(t)
        n = rng.normal(size=3); n -= n.dot(t) * t; n /= np.linalg.norm(n)
        b = np.cross(t, n)
        line_pos = rng.uniform(-0.3, 0.3, 3)
        radius = float(rng.choice([0.0025, 0.01, 0.025]))
        th = float(rng.choice([0.05, 0.15, 0.5, 1.0]))
        # camera: a few to a few hundred radii away, anywhere around the line
        cam = line_pos + (rng.normal(size=3) * radius * 10.0 ** rng.uniform(0.7, 2.5))
        w = cam - line_pos
        w = w - w.dot(t) * t
        cx, cy = w.dot(n) / radius, w.dot(b) / radius
        if cx * cx / (th * th) + cy * cy < 1.3:          # camera (projection) inside or nearly on the ellipse: no silhouette
            continue
        # visible side only: the fragment must face the camera
        phi = float(rng.uniform(0, 2 * np.pi))
        p = np.array([th * np.cos(phi), np.sin(phi)])
        grad = np.array([p[0] / th ** 2, p[1]])
        if grad.dot(np.array([cx, cy]) - p) <= 0.05:
            continue
        # tangent points: polar line (cx / th^2) x + cy y = 1 meets the ellipse
        A, B = cx / th ** 2, cy
        if abs(B) > abs(A):   # y = (1 - A x) / B
            qa = 1 / th ** 2 + (A / B) ** 2
            qb = -2 * A / B ** 2
            qc = 1 / B ** 2 - 1
            xs = np.roots([qa, qb, qc]).real
            pts = [np.array([x, (1 - A * x) / B]) for x in xs]
        else:                 # x = (1 - B y) / A
            qa = (B / A) ** 2 / th ** 2 + 1
            qb = -2 * B / (A ** 2 * th ** 2)
            qc = 1 / (A ** 2 * th ** 2) - 1
            ys = np.roots([qa, qb, qc]).real
            pts = [np.array([(1 - B * y) / A, y]) for y in ys]
        pm0, pm1 = pts
        # the line camera -> fragment crosses the chord at q
        c2 = np.array([cx, cy])
        d = p - c2
        s_ = (1 - A * c2[0] - B * c2[1]) / (A * d[0] + B * d[1])
        q = c2 + s_ * d
        r = np.linalg.norm(q - pm0) / np.linalg.norm(pm1 - pm0) * 2 - 1
        got = L.lvo_bands_ribbon_position(f3(*cam), f3(*line_pos), f3(*n), f3(*t), phi, radius, th)
        if not np.isfinite(got):
            continue
        err = abs(abs(got) - abs(r))
        worst = max(worst, err)
        # the shader's float32 homogeneous arithmetic loses accuracy with the square of the camera distance (in tube radii): measured
        # 8e-6 within 10 radii, 7e-4 within 100, 3e-2 beyond 300 -- a property of the formulation, identical on both sides
        dist = np.hypot(cx, cy)
        assert err < (1e-4 if dist < 30 else 1e-2 if dist < 300 else 0.1), (got, r, cx, cy, th, phi)
        assert abs(got) <= 1.0 + 5e-3
        checked += 1
    assert checked > 1200


def test_elliptic_all_hits_consumers_agree_with_the_closest_hit_loop():
    """The all-hits consumers of the oracle on band data (PPLL gather + resolve, MLAT with enough nodes) against its transparency
    loop of closest hits: three independent walks over the same tubelets -- brute force and tree -- give the same picture."""
    tr = ribbon_scene()
    s = dict(use_ribbons=True, band_width=0.05, min_band_thickness=0.3, use_analytic_elliptic_tubes=True,
             ppll_fragment_source="capsule_entry")   # (the same tubelets in all three walks; auto = the rasterised band prism)
    pts, seg, _ = lvo.build_tube_aabb_render_data_ribbons(tr.positions, tr.attributes, tr.line_offsets, 0.05, tr.ribbon_directions)
    c = Case(pts, seg, tfm.standard_transparent(), 120, 90, 0.02, **s)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    loop = sc.render_rt(P, use_bvh=True).astype(np.int32)
    with lvo.ppll_ray_tracer_fragment_colour():   # the same fragment colour as the loop (the gather's own is the raster shader's)
        ppll = sc.render_ppll(P, use_bvh=False)
        assert np.array_equal(ppll, sc.render_ppll(P, use_bvh=True))
    # exact sorting vs the closest-hit loop: the same layers front to back (the loop re-traces from hitT + eps, the lists sort by
    # camera distance; both stop at alpha 0.99).  The lists hold packUnorm4x8 colours: a highlight above 1.0 is clamped per fragment
    # there and only at the end in the loop -- those pixels (< 1 %) differ by more than the rounding
    d = np.abs(ppll.astype(np.int32) - loop).max(axis=2)
    assert (d > 2).mean() < 0.01 and d.max() < 32
    mlat, _, _ = sc.render_rt_mlat(P, 32, use_bvh=True)
    d = np.abs(mlat.astype(np.int32) - loop).max(axis=2)
    assert (d > 2).sum() <= 3 and (loop[..., :3] != 255).any(axis=2).sum() > 1000
    n1, s1, c1 = sc.ppll_gather(P, use_bvh=False)
    n2, s2, c2 = sc.ppll_gather(P, use_bvh=True)
    assert c1 == c2 and c1 > 2000


def test_elliptic_bvh_equals_brute_force_for_near_axis_parallel_rays():
    """The shader's own box test ignores axes with |d_i| < 1e-3; the own-box rule (slab interval) keeps the closest hit
    independent of the BVH for exactly those rays."""
    tr = ribbon_scene(n_lines=4, pts=80)
    bw, mbt = 0.01, 0.05
    pts, seg, _ = lvo.build_tube_aabb_render_data_ribbons(tr.positions, tr.attributes, tr.line_offsets, bw, tr.ribbon_directions)
    sc = lvo.Scene(pts, seg, tfm.standard())
    rng = np.random.default_rng(8)
    n = 40000
    tgt = pts["linePosition"][rng.integers(0, len(pts), n)] + rng.normal(scale=0.004, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tiny = rng.integers(0, 3, n)
    d[np.arange(n), tiny] = rng.uniform(-2e-3, 2e-3, n).astype(np.float32)      # one component around the 1e-3 threshold
    d[: n // 4, (tiny[: n // 4] + 1) % 3] = rng.uniform(-2e-3, 2e-3, n // 4).astype(np.float32)   # sometimes two
    o = (tgt - 0.3 * d).astype(np.float32)
    cam = np.array([0.0, 0.0, 0.8], np.float32)
    a = sc.trace_rays_elliptic(o, d, 1e-4, 1000.0, bw, mbt, cam, use_bvh=False)
    b = sc.trace_rays_elliptic(o, d, 1e-4, 1000.0, bw, mbt, cam, use_bvh=True)
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1])
    assert (a[1] != 0xFFFFFFFF).sum() > 2000


def test_elliptic_triangle_tubes_host_equals_oracle_and_lie_on_the_ellipse():
    """createCappedTriangleEllipticTubesRenderDataCPU: what the reference's triangle-mesh consumers (RTAO) get for band data.  The
    host's two-pass OpenMP tessellator against the oracle's append-as-you-go restatement, byte for byte; body vertices on the
    ellipse around their line point, caps inside the smaller semi-axis along the tangent, all indices valid."""
    tr = ribbon_scene(n_lines=5, pts=40)
    fl = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets, tr.ribbon_directions)
    for bw, mbt, n in ((0.05, 0.3, 8), (0.01, 0.15, 10), (0.03, 1.0, 5)):
        a = fl.tube_triangle_render_data_bands(bw, mbt, n)
        b = lvo.build_tube_triangle_render_data_ribbons(tr.positions, tr.attributes, tr.line_offsets, tr.ribbon_directions, bw, mbt, n)
        assert all(np.array_equal(np.ascontiguousarray(x).view(np.uint8), np.ascontiguousarray(y).view(np.uint8)) for x, y in zip(a, b))
        idx, verts, pts = a
        assert idx.max() == len(verts) - 1 and len(pts) == tr.num_points
        lpi = verts["vertexLinePointIndex"] & 0x7FFFFFFF
        body = (verts["vertexLinePointIndex"] >> 31) == 0
        d = (verts["vertexPosition"] - pts["linePosition"][lpi]).astype(np.float64)
        nrm, tan = pts["lineNormal"][lpi].astype(np.float64), pts["lineTangent"][lpi].astype(np.float64)
        bi = np.cross(tan, nrm)
        rn, rb = bw / 2 * mbt, bw / 2
        assert np.abs(((d * nrm).sum(1) / rn) ** 2 + ((d * bi).sum(1) / rb) ** 2 - 1)[body].max() < 1e-3
        assert np.abs((d * tan).sum(1))[body].max() < 1e-6
        assert np.linalg.norm(d[~body], axis=1).max() <= rb * 1.001
        assert np.abs(np.linalg.norm(verts["vertexNormal"], axis=1) - 1).max() < 1e-3
        # outward normals: the vertex normal points away from the line point
        assert ((verts["vertexNormal"].astype(np.float64) * d).sum(1) > 0).all()
    # without band data (or with use_ribbons off) the same accessor gives the circular tubes
    fl2 = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    c = fl2.tube_triangle_render_data(0.02, 6)
    ref = lvo.build_tube_triangle_render_data(tr.positions, tr.attributes, tr.line_offsets, 0.02, 6)
    assert all(np.array_equal(np.ascontiguousarray(x).view(np.uint8), np.ascontiguousarray(y).view(np.uint8)) for x, y in zip(c, ref))
