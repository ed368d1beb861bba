"""Multi-layer alpha tracing, CPU side: the oracle's restatement of MlatInsert.glsl /
traceRayMlat against an independent step-by-step evaluation, its limiting cases (exact transparency when the layers fit,
the opaque ray tracer), and the replay validator that the GPU parity tests rely on."""
import numpy as np
import pytest

from common import Case, max_lsb_diff, small_case
from linevis_amd import scenes, transfer_function as tfm
from oracle import lvo

f32 = np.float32


# ---- an independent evaluation of MlatInsert.glsl in numpy float32 scalars (one operation per line, as in the shader)
def py_merge(a, b, depth2, is_first):
    r = {"T": f32(a["T"] * b["T"]), "d": a["d"]}
    fa, fb = f32(1.0), a["T"]
    depth2 = max(depth2, b["d"])
    if b["d"] < depth2 and not is_first:
        d = f32(b["d"] - a["d"])
        d = f32(d / f32(depth2 - a["d"]))
        apd = f32(np.power(a["T"], d, dtype=f32))
        fa = f32(apd - f32(1.0))
        fa = f32(fa + f32(f32(a["T"] - apd) * b["T"]))
        fa = f32(fa / f32(a["T"] - f32(1.0)))
        fb = apd
    r["c"] = [f32(f32(fa * a["c"][k]) + f32(fb * b["c"][k])) for k in range(4)]
    return r, depth2


def py_insert(nodes, depth2, color, depth, miss=False):
    color = [f32(x) for x in color]
    depth = f32(depth)
    alpha = color[3]
    if not miss and alpha == 0:
        return nodes, depth2, False
    new = {"c": [f32(alpha * color[0]), f32(alpha * color[1]), f32(alpha * color[2]), color[3]],
           "T": f32(f32(1.0) - alpha), "d": depth}
    for i in range(len(nodes) - 1, -1, -1):
        if new["d"] > nodes[i]["d"]:
            new, nodes[i] = nodes[i], new
    if new["d"] > 0:
        nodes[0], depth2 = py_merge(new, nodes[0], depth2, new["d"] == depth)
    if alpha == 1:
        return nodes, depth2, True
    t = f32(1.0)
    for n in nodes:
        t = f32(t * n["T"])
    return nodes, depth2, bool(t <= f32(0.001) and nodes[-1]["d"] <= depth)


def py_empty(k):
    return [{"c": [f32(0)] * 4, "T": f32(1), "d": f32(0)} for _ in range(k)]


def as_array(nodes):
    return np.array([[*n["c"], n["T"], n["d"]] for n in nodes], dtype=np.float32)


def test_insert_known_answers():
    """Hand-checked sequence on two nodes: fill, overflow (merge in front, no overlap), a fragment INSIDE the merged
    span (the pow() branch), an opaque fragment (accepted)."""
    nodes = np.zeros((2, 6), np.float32)
    nodes[:, 4] = 1.0
    d2 = 0.0
    nodes, d2, acc = lvo.mlat_insert(nodes, d2, [1, 0, 0, 0.5], 1.0)
    nodes, d2, acc = lvo.mlat_insert(nodes, d2, [0, 1, 0, 0.5], 2.0)
    assert not acc and d2 == 0.0
    assert np.array_equal(nodes, np.array([[0.5, 0, 0, 0.5, 0.5, 1], [0, 0.5, 0, 0.5, 0.5, 2]], np.float32))
    nodes, d2, acc = lvo.mlat_insert(nodes, d2, [0, 0, 1, 0.5], 3.0)
    # depth-1 node fell off the front and was merged with the depth-2 node: c = a + T_a * b, T = T_a * T_b
    assert np.array_equal(nodes, np.array([[0.5, 0.25, 0, 0.75, 0.25, 1], [0, 0, 0.5, 0.5, 0.5, 3]], np.float32))
    assert d2 == 2.0 and not acc
    nodes, d2, acc = lvo.mlat_insert(nodes, d2, [1, 1, 1, 0.5], 1.5)
    # b = the new fragment (depth 1.5) lies inside a's span [1, 2]: d = 0.5, a^d = 0.5,
    # fa = ((0.5 - 1) + (0.25 - 0.5) * 0.5) / (0.25 - 1) = 5/6, fb = 0.5
    fa = f32(f32(f32(0.5 - 1.0) + f32(f32(0.25 - 0.5) * f32(0.5))) / f32(0.25 - 1.0))
    exp0 = [f32(fa * f32(0.5)) + f32(0.25), f32(fa * f32(0.25)) + f32(0.25), f32(0.25), f32(fa * f32(0.75)) + f32(0.25),
            0.125, 1.0]
    assert np.allclose(nodes[0], exp0, rtol=2e-7, atol=0) and np.array_equal(nodes[1], [0, 0, 0.5, 0.5, 0.5, 3])
    assert abs(fa - 5.0 / 6.0) < 1e-6 and d2 == 2.0 and not acc
    nodes, d2, acc = lvo.mlat_insert(nodes, d2, [0.2, 0.4, 0.6, 1.0], 2.5)
    assert acc and nodes[1][5] == 3.0 and nodes[0][5] == 1.0   # opaque: accepted, list = {merged, 2.5 | 3} after merge
    # fully transparent fragments are ignored, except in the miss shader
    n2, d2b, acc = lvo.mlat_insert(nodes, d2, [1, 1, 1, 0.0], 0.5)
    assert not acc and np.array_equal(n2, nodes)
    n3, _, _ = lvo.mlat_insert(nodes, d2, [1, 1, 1, 0.0], 1e7, miss=True)
    assert n3[1][5] == np.float32(1e7)


@pytest.mark.parametrize("k", [1, 2, 4, 8, 32])
def test_insert_random_sequences_against_step_by_step_evaluation(k):
    rng = np.random.default_rng(100 + k)
    for trial in range(20):
        nodes = np.zeros((k, 6), np.float32)
        nodes[:, 4] = 1.0
        d2 = 0.0
        ref, rd2 = py_empty(k), f32(0)
        for step in range(3 * k + 6):
            col = rng.uniform(0, 1, 4).astype(np.float32)
            if rng.uniform() < 0.1:
                col[3] = 1.0
            if rng.uniform() < 0.05:
                col[3] = 0.0
            depth = f32(rng.uniform(0.1, 10))
            nodes, d2, acc = lvo.mlat_insert(nodes, d2, col, depth)
            ref, rd2, racc = py_insert(ref, rd2, col, depth)
            assert acc == racc and f32(d2) == rd2
            # only pow() may differ (libm powf vs numpy): everything else is the same float32 sequence
            assert np.allclose(nodes, as_array(ref), rtol=1e-5, atol=1e-7)
            assert np.array_equal(nodes[:, 5], as_array(ref)[:, 5])
        assert np.all(np.diff(nodes[:, 5]) >= 0)   # sorted by depth, empties (depth 0) in front


def stick_case(transparent=True, n=60, seed=3, width=80, height=56, **settings):
    """Single-segment lines: no two capsules share a joint sphere, so no two fragments of a pixel have the same depth
    (with polylines the cap spheres of neighbouring segments coincide and MLAT sees the joint twice)."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(-0.4, 0.4, (n, 3)).astype(np.float32)
    b = (a + rng.normal(0, 0.25, (n, 3))).astype(np.float32)
    pos = np.stack([a, b], axis=1).reshape(-1, 3)
    att = rng.uniform(0, 1, 2 * n).astype(np.float32)
    off = np.arange(0, 2 * n + 1, 2, dtype=np.uint32)
    lw = 0.03
    pts, seg, _ = lvo.build_tube_aabb_render_data(pos, att, off, lw)
    tf = tfm.standard_transparent() if transparent else tfm.standard()
    return Case(pts, seg, tf, width, height, lw, **settings)


def test_all_layers_fit_equals_exact_transparency():
    """With at least as many nodes as layers nothing is merged: the blended node list is the sorted fragment list, i.e.
    the transparency loop of the ray tracer (different rounding order: within 1 LSB)."""
    c = stick_case(n=150)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    exact = sc.render_rt(P)
    img, nodes, viol = sc.render_rt_mlat(P, 32)
    # (the loop skips a layer closer than 1e-5 * t behind the previous one, HIT_DISTANCE_EPSILON; MLAT keeps both:
    # a pixel or two where two sticks cross)
    d = np.abs(img.astype(np.int32) - exact.astype(np.int32)).max(axis=2)
    assert viol == 0 and (d > 1).sum() <= 3
    layers = (nodes[..., 5:-1:6] > 0).sum(axis=2)
    assert layers.max() >= 8          # the scene does have depth complexity (incl. the background node)
    # fewer nodes: an approximation, but a good one on average, improving with the node count
    err = []
    for k in (1, 2, 4, 8):
        a, _, _ = sc.render_rt_mlat(P, k)
        err.append(np.abs(a.astype(np.int32) - exact.astype(np.int32)).mean())
    assert err[0] >= err[1] >= err[2] >= err[3] and err[3] < 0.05 and err[0] < 2.0


def test_visiting_order_does_not_matter_while_the_layers_fit():
    c = stick_case(n=40)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    k = 32
    img, nodes, _ = sc.render_rt_mlat(P, k)
    # the canonical order as a trace, then shuffled per pixel
    rng = np.random.default_rng(5)
    rec = []
    hits = all_hits_per_pixel(c, sc, P)
    for pix, segs in hits.items():
        order = rng.permutation(len(segs))
        rec += [(pix, s, segs[j], 0) for s, j in enumerate(order)]
    rec = np.array(rec, np.uint32).reshape(-1, 4)
    img2, nodes2, viol = sc.render_rt_mlat(P, k, trace=rec)
    assert viol == 0
    assert np.array_equal(nodes.view(np.uint32), nodes2.view(np.uint32)) and np.array_equal(img, img2)
    # with 2 nodes the order does matter
    a, na, _ = sc.render_rt_mlat(P, 2)
    b, nb, viol = sc.render_rt_mlat(P, 2, trace=rec)
    assert viol == 0 and not np.array_equal(na, nb)
    assert np.abs(a.astype(np.int32) - b.astype(np.int32)).mean() < 2.0


def all_hits_per_pixel(c, sc, P):
    """viewport pixel index -> segments hit by the pixel-centre ray (ascending)."""
    offs, segs, _ = sc.pixel_hits(P)
    return {pix: [int(x) for x in segs[int(offs[pix]):int(offs[pix + 1])]]
            for pix in range(c.width * c.height) if offs[pix + 1] > offs[pix]}


def test_replay_validator_catches_wrong_orders():
    c = stick_case(n=40)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    hits = all_hits_per_pixel(c, sc, P)
    rec = np.array([(pix, s, seg, 0) for pix, segs in hits.items() for s, seg in enumerate(segs)], np.uint32).reshape(-1, 4)
    canon, nodes, _ = sc.render_rt_mlat(P, 4)
    img, nodes2, viol = sc.render_rt_mlat(P, 4, trace=rec)
    assert viol == 0 and np.array_equal(img, canon) and np.array_equal(nodes.view(np.uint32), nodes2.view(np.uint32))
    # a visible layer is missing
    _, _, viol = sc.render_rt_mlat(P, 4, trace=rec[1:])
    assert viol >= 1
    # a segment the ray does not hit
    pix0 = int(rec[0, 0])
    miss = next(s for s in range(len(c.seg)) if s not in hits[pix0])
    bogus = np.concatenate([rec, np.array([[pix0, 999, miss, 0]], np.uint32)])
    _, _, viol = sc.render_rt_mlat(P, 4, trace=bogus)
    assert viol >= 1
    # a candidate marked "dropped" although the interval never shrank (transparent scene: nothing is accepted)
    wrong = rec.copy()
    wrong[0, 3] = 1
    _, _, viol = sc.render_rt_mlat(P, 4, trace=wrong)
    assert viol >= 1
    # the same candidate twice
    twice = np.concatenate([rec, rec[:1] + np.array([[0, 500, 0, 0]], np.uint32)])
    _, _, viol = sc.render_rt_mlat(P, 4, trace=twice)
    assert viol >= 1


def test_opaque_scene_early_termination():
    """Opaque tubes: the first opaque fragment a pixel meets is accepted and ends the ray interval; whatever the order,
    the frame is the opaque ray tracer's (halo edges are semi-transparent: within 1 LSB)."""
    c = stick_case(transparent=False)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    exact = sc.render_rt(P)
    hits = all_hits_per_pixel(c, sc, P)
    for k in (1, 4):
        img, _, viol = sc.render_rt_mlat(P, k)
        assert viol == 0
        d = np.abs(img.astype(np.int32) - exact.astype(np.int32)).max(axis=2)
        assert (d > 1).mean() < 0.01
    # back-to-front order with the interval rule: farther candidates first, each nearer opaque one accepted in turn
    rec = []
    for pix, segs in hits.items():
        rec += [(pix, s, seg, 0) for s, seg in enumerate(reversed(segs))]
    img, _, viol = sc.render_rt_mlat(P, 4, trace=np.array(rec, np.uint32).reshape(-1, 4))
    d = np.abs(img.astype(np.int32) - exact.astype(np.int32)).max(axis=2)
    assert (d > 1).mean() < 0.02


def test_triangle_mesh_mode_limits():
    """The triangle-mesh variant (candidates = triangles of the tube mesh): with enough nodes it is the transparency loop
    over the same mesh; replaying the canonical order reproduces it."""
    lw = 0.02
    tr = scenes.normalize(scenes.random_curves(n_lines=40, points_per_line=20, seed=7))
    mesh = lvo.build_tube_triangle_render_data(tr.positions, tr.attributes, tr.line_offsets, lw, 6)
    c = small_case(width=64, height=48, n_lines=40, pts_per_line=20, line_width=lw, transparent=True)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    ts = lvo.TriScene(*mesh, lw)
    exact = ts.render_rt(sc, P)
    img, nodes, viol = sc.render_rt_mlat(P, 32, tri_scene=ts)
    d = np.abs(img.astype(np.int32) - exact.astype(np.int32)).max(axis=2)
    # (triangles that share an edge or a vertex yield coincident layers the loop steps over: a few pixels)
    assert viol == 0 and (d > 1).mean() < 0.02
    img1, _, _ = sc.render_rt_mlat(P, 1, tri_scene=ts)
    assert np.abs(img1.astype(np.int32) - exact.astype(np.int32)).mean() < 4.0 and not np.array_equal(img, img1)


def test_jittered_samples_and_tiles():
    c = small_case(transparent=True, num_samples_per_frame=3)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    full, _, _ = sc.render_rt_mlat(P, 4)
    tile, _, _ = sc.render_rt_mlat(P, 4, tile=(16, 8, 40, 24))
    assert np.array_equal(tile, full[8:32, 16:56])
    bvh, _, _ = sc.render_rt_mlat(P, 4, use_bvh=True)
    assert np.array_equal(bvh, full)


def test_golden_fixture():
    import os
    from common import GOLDEN_DIR
    g = np.load(os.path.join(GOLDEN_DIR, "mlat_small.npz"))
    c = small_case(width=48, height=32, transparent=True, n_lines=40)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    for k in (2, 8):
        img, nodes, _ = sc.render_rt_mlat(P, k)
        assert np.array_equal(img, g["frame_k%d" % k])
        assert np.allclose(nodes, g["nodes_k%d" % k], rtol=1e-5, atol=1e-7)   # pow() is libm's
        assert np.array_equal(nodes[..., 5::6], g["nodes_k%d" % k][..., 5::6])
