"""a14 (triangle-tube tessellation) and RTAO against the triangle tubes -- CPU side: the oracle against the committed
fixtures and independent properties, and the host layer (C++, parallel two-pass tessellator) byte-for-byte against the
oracle's literal restatement of CappedTriangleTubesCPU.cpp."""
import os
from collections import Counter

import numpy as np
import pytest

from common import GOLDEN_DIR, small_case
from linevis_amd import host_api, scenes
from oracle import lvo

G = np.load(os.path.join(GOLDEN_DIR, "triangle_tubes.npz"))
A2 = np.load(os.path.join(GOLDEN_DIR, "a2_cases.npz"))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def straight_line(n=10):
    pos = np.stack([np.linspace(-0.4, 0.4, n), np.zeros(n), np.zeros(n)], 1).astype(np.float32)
    return pos, np.linspace(0, 1, n).astype(np.float32), np.array([0, n], np.uint32)


# ---------------------------------------------------------------- tessellation
@pytest.mark.parametrize("n", [6, 4, 9])
def test_tessellation_golden_corner_cases(n):
    idx, verts, pts = lvo.build_tube_triangle_render_data(A2["positions"], A2["attributes"], A2["line_offsets"],
                                                          float(A2["line_width"]), n)
    assert np.array_equal(idx, G["a2_idx_n%d" % n])
    assert np.array_equal(verts.view(np.uint8).reshape(-1, 32), G["a2_verts_n%d" % n])
    assert np.array_equal(pts.view(np.uint8).reshape(-1, 48), G["a2_points_n%d" % n])


@pytest.mark.parametrize("n", [3, 4, 6, 7, 12])
def test_tessellation_structure(n):
    """Counts of CappedTriangleTubesCPU.cpp:229-234, watertightness, vertex radii, cap flags, outward orientation."""
    pos, att, off = straight_line(10)
    lw = 0.02
    idx, verts, pts = lvo.build_tube_triangle_render_data(pos, att, off, lw, n)
    N = max(n, 4)                       # numCircleSubdivisions = max(numCircleSubdivisions, 4), :223
    L = N // 2
    cap_v, cap_i = N * (L - 1) + 1, N * (L - 1) * 6 + N * 3
    assert len(pts) == 10 and len(verts) == 10 * N + 2 * cap_v and idx.size == 9 * N * 6 + 2 * cap_i
    edges = Counter()
    for a, b, c in idx:
        for u, v in ((a, b), (b, c), (c, a)):
            edges[(min(u, v), max(u, v))] += 1
    assert set(edges.values()) == {2}                                  # closed 2-manifold
    lp = pts["linePosition"][verts["vertexLinePointIndex"] & 0x7FFFFFFF]
    assert np.allclose(np.linalg.norm(verts["vertexPosition"] - lp, axis=1), lw / 2, rtol=2e-6)
    is_cap = (verts["vertexLinePointIndex"] >> 31).astype(bool)
    assert is_cap.sum() == 2 * cap_v and not is_cap[cap_v:cap_v + 10 * N].any()
    assert np.allclose(np.linalg.norm(verts["vertexNormal"], axis=1), 1.0, atol=1e-6)
    v, nrm = verts["vertexPosition"], verts["vertexNormal"]
    fn = np.cross(v[idx[:, 1]] - v[idx[:, 0]], v[idx[:, 2]] - v[idx[:, 0]])
    assert np.all(np.einsum("ij,ij->i", fn, nrm[idx[:, 0]] + nrm[idx[:, 1]] + nrm[idx[:, 2]]) > 0)   # CCW seen from outside
    body_phi = verts["phi"][cap_v:cap_v + N]
    assert np.allclose(body_phi, np.arange(N) / N * 2 * np.pi, atol=1e-6)
    # the line-point table equals the one of the AABB path (same tangent / Gram-Schmidt rules) except lineStartIndex
    p2, _, _ = lvo.build_tube_aabb_render_data(pos, att, off, lw)
    for k in ("linePosition", "lineAttribute", "lineTangent", "lineNormal"):
        assert np.array_equal(pts[k], p2[k])


def test_tessellation_degenerate_lines_follow_the_reference_quirks():
    """A line with one valid point keeps the reserved (zero) start-cap indices but no vertices; a line without any
    valid point keeps the start cap's zero vertices and indices (CappedTriangleTubesCPU.cpp:253-262,307-316)."""
    pos = np.array([[0, 0, 0], [0, 0, 0], [0.00001, 0, 0],                     # 3 points, none valid
                    [0.5, 0, 0],                                               # single point: skipped entirely
                    [0, 0.2, 0], [0, 0.2, 0.00006], [0, 0.2, 0.00012]], np.float32)  # only the middle point is valid
    off = np.array([0, 3, 4, 7], np.uint32)
    att = np.zeros(len(pos), np.float32)
    idx, verts, pts = lvo.build_tube_triangle_render_data(pos, att, off, 0.02, 6)
    assert len(pts) == 0 and len(verts) == 13 and idx.size == 90 + 90
    assert not idx.any() and not verts.view(np.uint8).any()
    empty = lvo.build_tube_triangle_render_data(pos[:0], att[:0], np.array([0], np.uint32), 0.02, 6)
    assert all(len(a) == 0 for a in empty)


@pytest.mark.parametrize("seed,n,lw", [(1, 6, 0.02), (2, 8, 0.004), (3, 4, 0.01), (4, 5, 0.03)])
def test_host_tessellation_is_byte_identical_to_the_oracle(seed, n, lw):
    tr = scenes.normalize(scenes.random_curves(n_lines=25, points_per_line=35, seed=seed))
    pos = tr.positions.copy()
    pos[40] = pos[39]                   # duplicate vertices inside a line
    pos[70:73] = pos[70]
    flow = host_api.LineDataFlow().set_trajectories(pos, tr.attributes, tr.line_offsets)
    a = flow.tube_triangle_render_data(lw, n)
    b = lvo.build_tube_triangle_render_data(pos, tr.attributes, tr.line_offsets, lw, n)
    assert np.array_equal(a[0], b[0]) and a[1].tobytes() == b[1].tobytes() and a[2].tobytes() == b[2].tobytes()


def test_host_tessellation_corner_cases_match_golden():
    flow = host_api.LineDataFlow().set_trajectories(A2["positions"], A2["attributes"], A2["line_offsets"])
    for n in (6, 4, 9):
        idx, verts, pts = flow.tube_triangle_render_data(float(A2["line_width"]), n)
        assert np.array_equal(idx, G["a2_idx_n%d" % n])
        assert np.array_equal(verts.view(np.uint8).reshape(-1, 32), G["a2_verts_n%d" % n])
        assert np.array_equal(pts.view(np.uint8).reshape(-1, 48), G["a2_points_n%d" % n])


# ---------------------------------------------------------------- ray-triangle test
def test_triangle_known_answers_and_float64_truth():
    o, d, v0, v1, v2 = (G[k] for k in ("kat_o", "kat_d", "kat_v0", "kat_v1", "kat_v2"))
    pad = float(G["kat_pad"])
    n = len(o)
    hit = np.zeros(n, np.uint8); t = np.zeros(n, np.float32); uv = np.zeros((n, 2), np.float32)
    for i in range(n):
        h, tt, uu, vv = lvo.intersect_triangle(o[i], d[i], v0[i], v1[i], v2[i], pad)
        hit[i], t[i], uv[i] = h, tt, (uu, vv)
    assert np.array_equal(hit, G["kat_hit"]) and np.array_equal(bits(t), G["kat_t_bits"]) and np.array_equal(bits(uv), G["kat_uv_bits"])
    # independent float64 Moeller-Trumbore: away from the triangle's edges both must agree on hit/miss, and the hit
    # point reconstructed from (u, v) must lie on the ray at parameter t
    O, D, A, B, C = (x.astype(np.float64) for x in (o, d, v0, v1, v2))
    e1, e2 = B - A, C - A
    p = np.cross(D, e2)
    det = np.einsum("ij,ij->i", e1, p)
    ok = np.abs(det) > 1e-12
    inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
    tv = O - A
    u = np.einsum("ij,ij->i", tv, p) * inv
    q = np.cross(tv, e1)
    v = np.einsum("ij,ij->i", D, q) * inv
    tt = np.einsum("ij,ij->i", e2, q) * inv
    margin = 1e-4
    clear_hit = ok & (u > margin) & (v > margin) & (u + v < 1 - margin)
    clear_miss = ~ok | (u < -margin) | (v < -margin) | (u + v > 1 + margin)
    assert np.all(hit[clear_hit] == 1) and np.all(hit[clear_miss] == 0)
    assert clear_hit.sum() > 150 and clear_miss.sum() > 30
    h = hit.astype(bool)
    assert np.allclose(t[h], tt[h], rtol=2e-4, atol=2e-6) and np.allclose(uv[h, 0], u[h], atol=2e-3)
    on_ray = O[h] + D[h] * t[h, None].astype(np.float64)
    on_tri = A[h] + uv[h, :1] * e1[h] + uv[h, 1:] * e2[h]
    assert np.abs(on_ray - on_tri).max() < 5e-6
    # negative t is reported (the caller's [tMin, tMax] test rejects it), degenerate triangles never hit
    assert (t[h] < 0).any()
    degenerate = np.linalg.norm(np.cross(e1, e2), axis=1) < 1e-10
    assert degenerate.sum() >= 10 and not hit[degenerate].any()


def test_own_box_rule_only_cuts_float_noise():
    """The 'inside the own padded AABB' clause must not reject geometrically valid hits (well-conditioned or grazing)."""
    rng = np.random.default_rng(4)
    rejected = 0
    total = 0
    for i in range(3000):
        a = rng.uniform(-0.3, 0.3, 3)
        b = a + rng.normal(scale=0.004, size=3)
        c = a + rng.normal(scale=0.004, size=3)
        w = rng.dirichlet((1, 1, 1))
        if w.min() < 0.05:
            continue
        tgt = w[0] * a + w[1] * b + w[2] * c
        nrm = np.cross(b - a, c - a); nrm /= np.linalg.norm(nrm)
        graze = rng.uniform(0.01, 1.0)                    # down to ~0.6 degrees against the plane
        tang = (b - a) / np.linalg.norm(b - a)
        dirv = tang * np.sqrt(1 - graze ** 2) + nrm * graze
        o = tgt - dirv * rng.uniform(0.05, 1.0)
        v0, v1, v2 = (x.astype(np.float32) for x in (a, b, c))
        h_pad, *_ = lvo.intersect_triangle(o.astype(np.float32), dirv.astype(np.float32), v0, v1, v2, 2e-6)
        h_nopad, *_ = lvo.intersect_triangle(o.astype(np.float32), dirv.astype(np.float32), v0, v1, v2, 1e3)  # box = everything
        total += int(h_nopad)
        rejected += int(h_nopad and not h_pad)
    assert total > 2000 and rejected <= total // 200


# ---------------------------------------------------------------- triangle scene
def small_mesh(lw=0.02, seed=7):
    tr = scenes.normalize(scenes.random_curves(n_lines=30, points_per_line=30, seed=seed))
    return tr, lvo.build_tube_triangle_render_data(tr.positions, tr.attributes, tr.line_offsets, lw, 6)


def test_triangle_bvh_equals_brute_force():
    lw = 0.02
    _, mesh = small_mesh(lw)
    ts = lvo.TriScene(*mesh, lw)
    rng = np.random.default_rng(9)
    o = rng.uniform(-0.35, 0.35, (6000, 3)).astype(np.float32)
    d = rng.normal(size=(6000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:60] = 0.0; d[:20, 0] = 1.0; d[20:40, 1] = 1.0; d[40:60, 2] = -1.0
    for tmin, tmax in ((0.0, 0.1), (1e-4, 1000.0)):
        a = ts.trace_rays(o, d, tmin, tmax, use_bvh=True)
        b = ts.trace_rays(o, d, tmin, tmax, use_bvh=False)
        assert np.array_equal(a[1], b[1]) and np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(bits(a[2]), bits(b[2]))
    assert (a[1] != 0xFFFFFFFF).sum() > 500


def test_triangle_rtao_golden_bvh_and_tiles():
    W, H, lw = int(G["ao_width"]), int(G["ao_height"]), float(G["ao_line_width"])
    tr = scenes.normalize(scenes.random_curves(n_lines=36, points_per_line=40, seed=11))
    mesh = lvo.build_tube_triangle_render_data(tr.positions, tr.attributes, tr.line_offsets, lw, 6)
    assert len(mesh[0]) == int(G["ao_num_triangles"])
    assert np.uint32(np.bitwise_xor.reduce(mesh[1].view(np.uint32)) ^ np.bitwise_xor.reduce(mesh[0].reshape(-1))) == G["ao_mesh_crc"]
    case = small_case(width=W, height=H, n_lines=36, pts_per_line=40, seed=11, line_width=lw,
                      ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0,
                      ambient_occlusion_iterations=2, ambient_occlusion_samples_per_frame=8)
    P = case.oracle_params()
    ts = lvo.TriScene(*mesh, lw)
    ao = ts.render_ao(P, use_bvh=True)
    assert np.array_equal(bits(ao), G["ao_bits"])
    # tiles reproduce the frame (seeds use global pixel coordinates)
    tile = ts.render_ao(P, tile=(16, 8, 24, 32), use_bvh=True)
    assert np.array_equal(bits(tile[8:40, 16:40]), G["ao_bits"][8:40, 16:40])


def test_capsule_ao_is_statistically_close_to_triangle_tube_ao():
    """SURVEY.md hard part A(ii): capsule AO is the build's default RTAO geometry; against the reference's triangle tubes
    it may differ per pixel (silhouettes, faceting) but not systematically."""
    lw = 0.02
    _, mesh = small_mesh(lw)
    case = small_case(line_width=lw, ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0,
                      ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=32)
    sc = case.oracle_scene()
    P = case.oracle_params(sc)
    ao_c = sc.render_ao(P, use_bvh=True)
    ao_t = lvo.TriScene(*mesh, lw).render_ao(P, use_bvh=True)
    hit_c, hit_t = ao_c < 1.0, ao_t < 1.0
    both = hit_c & hit_t
    assert both.sum() > 600
    # the inscribed hexagon covers slightly fewer pixels than the round capsule
    assert 0.85 * hit_c.sum() <= hit_t.sum() <= hit_c.sum()
    assert abs(float(ao_c[both].mean()) - float(ao_t[both].mean())) < 0.02
    assert np.corrcoef(ao_c[both], ao_t[both])[0, 1] > 0.8
