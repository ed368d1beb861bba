"""Generates the committed golden fixtures from the CPU oracle (oracle/lv_oracle.cpp).

PARITY UNPINNED: the reference's own tests hold no vectors for this path and the reference
cannot be built here, so these vectors pin the oracle against regressions and travel to the GPU box where the
HIP path is compared against them.  Fixtures are data only: inputs + expected outputs.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from common import Case, small_case, scene_arrays  # noqa: E402
from linevis_amd import scenes, transfer_function as tfm  # noqa: E402
from oracle import lvo  # noqa: E402


def f2u(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def rng_kat():
    tea_in = [(0, 0), (1, 0), (0, 1), (19, 0), (19, 1), (19, 7), (12345, 678), (1920 * 1080 - 1, 63),
              (0xFFFFFFFF, 0xFFFFFFFF), (959 + 539 * 1920, 0)]
    tea_out = np.array([lvo.tea(a, b) for a, b in tea_in], dtype=np.uint32)
    seeds = np.array([0, 1, 0xDEADBEEF, int(tea_out[3]), int(tea_out[9])], dtype=np.uint32)
    rnd = np.stack([lvo.rnd_sequence(int(s), 8) for s in seeds])
    xi = np.array([0.0, 1.0 / 16777216.0, 0.125, 0.25, 0.3, 0.5, 0.62, 0.75, 0.875, 16777215.0 / 16777216.0],
                  dtype=np.float32)
    sc = np.array([lvo.sincos_2pi(float(x)) for x in xi], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "rng_kat.npz"), tea_in=np.array(tea_in, dtype=np.uint32), tea_out=tea_out,
                        rnd_seeds=seeds, rnd_bits=f2u(rnd), sincos_xi=xi, sincos_bits=f2u(sc))


def capsule_kat():
    rng = np.random.default_rng(2024)
    O, D, P0, P1, R = [], [], [], [], []

    def add(o, d, p0, p1, r):
        O.append(o); D.append(d); P0.append(p0); P1.append(p1); R.append(r)

    for i in range(300):
        p0 = rng.uniform(-0.3, 0.3, 3)
        axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
        p1 = p0 + axis * rng.uniform(0.005, 0.2)
        r = rng.uniform(0.001, 0.03)
        kind = i % 10
        mid = 0.5 * (p0 + p1)
        perp = np.cross(axis, rng.normal(size=3)); perp /= np.linalg.norm(perp)
        if kind == 0:      # generic ray towards the capsule
            o = mid + rng.normal(size=3) * 0.5
            d = (mid + perp * r * rng.uniform(-1.2, 1.2)) - o
        elif kind == 1:    # grazing: aimed at the silhouette
            o = mid + perp * 0.6
            d = (mid + np.cross(axis, perp) * r * rng.uniform(0.999, 1.001)) - o
        elif kind == 2:    # origin inside the cylinder part
            o = mid + perp * r * 0.3
            d = rng.normal(size=3)
        elif kind == 3:    # origin inside an end sphere
            o = p0 + rng.normal(size=3) * r * 0.2
            d = rng.normal(size=3)
        elif kind == 4:    # ray along the axis through the caps
            o = p0 - axis * 0.4 + perp * r * rng.uniform(0, 0.9)
            d = axis
        elif kind == 5:    # ray exactly parallel to the axis but outside
            o = p0 - axis * 0.4 + perp * r * 1.5
            d = axis
        elif kind == 6:    # hits near the end plane
            o = p1 + perp * 0.5 + axis * r * rng.uniform(-0.5, 0.5)
            d = -perp
        elif kind == 7:    # cap only: aimed beyond the end plane
            o = p1 + axis * 0.4 + perp * r * rng.uniform(-0.8, 0.8)
            d = -axis
        elif kind == 8:    # pointing away
            o = mid + perp * 0.4
            d = perp
        else:              # unnormalised direction
            o = mid + rng.normal(size=3) * 0.4
            d = (mid - o) * rng.uniform(0.1, 5.0)
        if kind != 9:
            d = d / np.linalg.norm(d)
        add(o, d, p0, p1, r)
    O, D, P0, P1 = [np.array(a, dtype=np.float32) for a in (O, D, P0, P1)]
    R = np.array(R, dtype=np.float32)
    res = {}
    for capped in (0, 1):
        hit = np.zeros(len(R), dtype=np.uint8)
        t = np.zeros(len(R), dtype=np.float32)
        k = np.zeros(len(R), dtype=np.int32)
        for i in range(len(R)):
            h, tt, kk = lvo.intersect_capsule(O[i], D[i], P0[i], P1[i], float(R[i]), bool(capped))
            hit[i], t[i], k[i] = h, tt, kk
        res["hit%d" % capped] = hit
        res["t_bits%d" % capped] = f2u(t)
        res["kind%d" % capped] = k
    np.savez_compressed(os.path.join(HERE, "capsule_kat.npz"), o=O, d=D, p0=P0, p1=P1, r=R, **res)


def a2_cases():
    lines = [
        # duplicate vertices in the middle and at the end
        [(0, 0, 0), (0.1, 0, 0), (0.1, 0, 0), (0.1, 0, 0), (0.2, 0.05, 0), (0.3, 0.1, 0.02), (0.3, 0.1, 0.02)],
        # tangent parallel to the initial helper axis (1,0,0) -> falls back to (0,1,0)
        [(0, 0.2, 0), (0.05, 0.2, 0), (0.1, 0.2, 0), (0.15, 0.2, 0)],
        # tangent parallel to (0,1,0) after the normal has become (0,1,0)... forces the (0,0,1) fallback
        [(0.2, 0, 0.1), (0.2, 0.05, 0.1), (0.2, 0.1, 0.1), (0.2, 0.15, 0.1)],
        # single point (dropped) and a two-point line whose points coincide (dropped)
        [(0.3, 0.3, 0.3)],
        [(0.1, 0.1, 0.1), (0.1, 0.1, 0.1)],
        # only one valid point survives -> dropped
        [(0.4, 0, 0), (0.4, 0, 0), (0.4, 0.00001, 0)],
        # ordinary curved line
        [(-0.2, -0.2, 0), (-0.15, -0.18, 0.02), (-0.1, -0.12, 0.05), (-0.08, -0.05, 0.1), (-0.1, 0.0, 0.16)],
        # sharp reversal
        [(0, -0.3, 0), (0.1, -0.3, 0), (0.0, -0.3, 0.001), (0.1, -0.31, 0)],
    ]
    pos = np.concatenate([np.array(l, dtype=np.float32).reshape(-1, 3) for l in lines])
    off = np.zeros(len(lines) + 1, dtype=np.uint32)
    off[1:] = np.cumsum([len(l) for l in lines])
    att = np.linspace(0.0, 1.0, len(pos)).astype(np.float32)
    pts, seg, aabb = lvo.build_tube_aabb_render_data(pos, att, off, 0.01)
    np.savez_compressed(os.path.join(HERE, "a2_cases.npz"), positions=pos, attributes=att, line_offsets=off,
                        line_width=np.float32(0.01), points=pts.view(np.uint8).reshape(-1, 48), seg=seg,
                        aabb_bits=f2u(aabb))


def small_scene():
    W, H = 128, 128
    base = small_case(width=W, height=H, n_lines=36, pts_per_line=40, seed=11, line_width=0.015)
    out = dict(points=base.points.view(np.uint8).reshape(-1, 48), seg=base.seg, tf=base.tf, width=W, height=H,
               line_width=np.float32(base.line_width), tf_transparent=tfm.standard_transparent())
    # 1. opaque, primary rays only, with depth cues
    c = Case(base.points, base.seg, base.tf, W, H, base.line_width, depth_cue_strength=0.8)
    out["rt_depthcue"], _ = c.oracle_render(11)
    sc = c.oracle_scene()
    out["depth_range_bits"] = f2u(sc.depth_range(c.oracle_params()))
    # 2. transparent transfer function, 4 jittered samples per pixel
    c = Case(base.points, base.seg, out["tf_transparent"], W, H, base.line_width, num_samples_per_frame=4)
    out["rt_transparent_spp4"], _ = c.oracle_render(11)
    # 3. RTAO: 2 iterations x 8 samples, distance based
    ao_set = dict(ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=0.9,
                  ambient_occlusion_gamma=1.5, ambient_occlusion_iterations=2, ambient_occlusion_samples_per_frame=8,
                  ambient_occlusion_radius=0.1)
    c = Case(base.points, base.seg, base.tf, W, H, base.line_width, **ao_set)
    img, ao = c.oracle_render(11)
    out["rt_ao"], out["ao_bits"] = img, f2u(ao)
    # 4. PPLL with the transparent transfer function (fragments = capsule entry hits: the probe of rounds 1-3; the rasterised
    #    prism has its own fixture, prism_small())
    c = Case(base.points, base.seg, out["tf_transparent"], W, H, base.line_width, ppll_fragment_source="capsule_entry")
    st = lvo.Stats()
    out["ppll"], _ = c.oracle_render(2, stats=st)
    out["ppll_fragments"] = np.uint64(st.fragments)
    out["ppll_max_depth_complexity"] = np.uint32(st.maxDepthComplexity)
    # 5. ray known answers: pixel-centre primaries of a coarse grid + random rays from inside the scene
    rng = np.random.default_rng(5)
    o = np.concatenate([np.tile(np.array([[0.0, 0.0, 0.8]], dtype=np.float32), (300, 1)),
                        rng.uniform(-0.3, 0.3, (212, 3)).astype(np.float32)])
    d = rng.normal(size=(512, 3)).astype(np.float32)
    d[:300, 2] = -np.abs(d[:300, 2]) * 4.0
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t, s, k = base.oracle_scene().trace_rays(o, d, 1e-4, 1000.0, base.line_width)
    out.update(ray_o=o, ray_d=d.astype(np.float32), ray_t_bits=f2u(t), ray_seg=s, ray_kind=k)
    np.savez_compressed(os.path.join(HERE, "scene_small.npz"), **out)


def ppll_lists():
    """Hand-made fragment lists: depth ties with different colours, > MAX_NUM_FRAGS overflow, empty pixels,
    fragments whose alpha quantises to 0."""
    W, H = 8, 8
    max_frags = 8
    rng = np.random.default_rng(99)
    pw, ph = 8, 8
    nodes = []
    start = np.full(pw * ph, 0xFFFFFFFF, dtype=np.uint32)

    def push(x, y, rgba, depth):
        idx = len(nodes)
        a = lvo.ppll_addr(x, y, pw, 2, 8)
        col = (int(rgba[0]) | (int(rgba[1]) << 8) | (int(rgba[2]) << 16) | (int(rgba[3]) << 24)) & 0xFFFFFFFF
        nodes.append((col, int(np.float32(depth).view(np.uint32)), int(start[a])))
        start[a] = idx

    for y in range(H):
        for x in range(W):
            kind = (x + y * W) % 8
            n = [0, 1, 3, 8, 12, 5, 6, 20][kind]
            depths = rng.uniform(0.3, 1.2, n).astype(np.float32)
            if kind in (5, 6) and n >= 4:
                depths[1] = depths[0]
                depths[3] = depths[2]
            for i in range(n):
                rgba = rng.integers(0, 256, 4)
                if kind == 6 and i == 0:
                    rgba[3] = 0
                if kind == 2:
                    rgba[3] = 255
                push(x, y, rgba, depths[i])
    nodes = np.array(nodes, dtype=np.uint32).reshape(-1, 3)
    from linevis_amd import camera
    view, proj, fovy, near, far = camera.default_camera(W, H)
    P = lvo.make_params(view, proj, W, H, ppllMaxNumFrags=max_frags, ppllLinkedListSize=len(nodes),
                        background=(0.2, 0.4, 0.6, 1.0))
    img_key = lvo.ppll_resolve(P, nodes, start, literal=False)
    img_lit = lvo.ppll_resolve(P, nodes, start, literal=True)
    np.savez_compressed(os.path.join(HERE, "ppll_lists.npz"), nodes=nodes, start=start, width=W, height=H,
                        max_frags=max_frags, background=np.array([0.2, 0.4, 0.6, 1.0], dtype=np.float32),
                        resolved_key=img_key, resolved_literal=img_lit)


def lattice_c1():
    """Config 1 analogue: 32 x 32 lines through a uniform grid, 128 x 128, 4 spp, no AO."""
    tr = scenes.normalize(scenes.lattice())
    pts, seg = scene_arrays(tr, 0.004)
    c = Case(pts, seg, tfm.standard(), 128, 128, 0.004, num_samples_per_frame=4)
    img, _ = c.oracle_render(11, use_bvh=True)
    np.savez_compressed(os.path.join(HERE, "lattice_c1.npz"), image=img, line_width=np.float32(0.004),
                        num_points=np.uint32(len(pts)), num_segments=np.uint32(len(seg)),
                        points_crc=np.uint32(np.bitwise_xor.reduce(pts.view(np.uint32))))


def triangle_tubes():
    """a14 + RTAO against the triangle tubes: tessellation of the a2 corner-case lines (byte level), ray-triangle
    known answers (t, u, v bits) and one AO image of a small curved scene."""
    a2 = np.load(os.path.join(HERE, "a2_cases.npz"))
    out = {}
    for n in (6, 4, 9):
        idx, verts, pts = lvo.build_tube_triangle_render_data(a2["positions"], a2["attributes"], a2["line_offsets"],
                                                              float(a2["line_width"]), n)
        out["a2_idx_n%d" % n] = idx
        out["a2_verts_n%d" % n] = verts.view(np.uint8).reshape(-1, 32)
        out["a2_points_n%d" % n] = pts.view(np.uint8).reshape(-1, 48)
    # ray-triangle known answers
    rng = np.random.default_rng(77)
    n = 400
    v0 = rng.uniform(-0.3, 0.3, (n, 3)).astype(np.float32)
    v1 = (v0 + rng.normal(scale=0.02, size=(n, 3))).astype(np.float32)
    v2 = (v0 + rng.normal(scale=0.02, size=(n, 3))).astype(np.float32)
    w = rng.dirichlet((1.0, 1.0, 1.0), n)
    w[::7] = rng.dirichlet((1.0, 1.0, 1.0), len(w[::7])) * 1.3 - 0.1     # some targets just outside the triangle
    w[::11, 0] = 0.0                                                      # some exactly on an edge
    w[::11, 1:] = rng.dirichlet((1.0, 1.0), len(w[::11]))
    target = (w[:, :1] * v0 + w[:, 1:2] * v1 + w[:, 2:] * v2)
    o = (target + rng.normal(size=(n, 3)) * rng.uniform(0.01, 1.0, (n, 1))).astype(np.float32)
    d = (target - o)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    d[5::13] = -d[5::13]                                                  # pointing away -> negative t
    v2[3::17] = (v0[3::17] + 2.0 * (v1[3::17] - v0[3::17])).astype(np.float32)  # degenerate (collinear) triangles
    o[1::19] = np.float32(0.0); d[1::19] = np.array([0, 0, 1], np.float32)      # axis-parallel rays (1/d = inf)
    v0[1::19] = np.array([-0.1, -0.1, 0.2], np.float32) + v0[1::19] * 0
    v1[1::19] = np.array([0.2, -0.1, 0.25], np.float32)
    v2[1::19] = np.array([-0.1, 0.2, 0.22], np.float32)
    # pad of a scene with line width 0.002, in float32 operations like the library: r * 1e-3f + 1e-6f
    pad = np.float32(np.float32(0.002) * np.float32(0.5)) * np.float32(1e-3) + np.float32(1e-6)
    hit = np.zeros(n, np.uint8); t = np.zeros(n, np.float32); uv = np.zeros((n, 2), np.float32)
    for i in range(n):
        h, tt, uu, vv = lvo.intersect_triangle(o[i], d[i], v0[i], v1[i], v2[i], float(pad))
        hit[i], t[i], uv[i] = h, tt, (uu, vv)
    out.update(kat_o=o, kat_d=d, kat_v0=v0, kat_v1=v1, kat_v2=v2, kat_pad=pad, kat_hit=hit, kat_t_bits=f2u(t),
               kat_uv_bits=f2u(uv))
    # AO image: curved scene, 1 iteration x 8 samples + a second accumulation iteration
    W, H, lw = 64, 64, 0.015
    tr = scenes.normalize(scenes.random_curves(n_lines=36, points_per_line=40, seed=11))
    mesh = lvo.build_tube_triangle_render_data(tr.positions, tr.attributes, tr.line_offsets, lw, 6)
    base = small_case(width=W, height=H, n_lines=36, pts_per_line=40, seed=11, line_width=lw,
                      ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0,
                      ambient_occlusion_iterations=2, ambient_occlusion_samples_per_frame=8)
    P = base.oracle_params()
    ao = lvo.TriScene(*mesh, lw).render_ao(P, use_bvh=False)
    out.update(ao_width=W, ao_height=H, ao_line_width=np.float32(lw), ao_bits=f2u(ao),
               ao_num_triangles=np.uint32(len(mesh[0])),
               ao_mesh_crc=np.uint32(np.bitwise_xor.reduce(mesh[1].view(np.uint32)) ^ np.bitwise_xor.reduce(mesh[0].reshape(-1))))
    np.savez_compressed(os.path.join(HERE, "triangle_tubes.npz"), **out)


def flow_small():
    """Streamline tracing known answers: ABC flow on a 16^3 grid (resScale 6), 40 seeds, every integrator."""
    n = 16
    v = lvo.generate_abc_flow(n, n, n)
    mag = np.sqrt((v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]) + v[..., 2] * v[..., 2]).astype(np.float32)
    d = float(np.float32(1.0) / np.float32(n - 1))
    rng = np.random.default_rng(21)
    seeds = rng.uniform(0.1, 0.9, (40, 3)).astype(np.float32)
    out = dict(n=n, spacing=np.float32(d), seeds=seeds, field_crc=np.uint32(np.bitwise_xor.reduce(v.reshape(-1).view(np.uint32))))
    for key, method, direction in (("rk4_both", "Runge-Kutta 4th Order", "Forward & Backward"),
                                   ("euler_fwd", "Explicit Euler", "Forward"), ("heun_bwd", "Heun", "Backward"),
                                   ("midpoint_both", "Midpoint", "Forward & Backward"),
                                   ("implicit_fwd", "Implicit Euler", "Forward")):
        S = lvo.streamline_settings(method, direction, minimum_length=0.25)
        pos, att, off = lvo.trace_streamlines(v, (d, d, d), [mag], seeds, S)
        out[key + "_pos_bits"] = f2u(pos)
        out[key + "_att_bits"] = f2u(att)
        out[key + "_off"] = off
    np.savez_compressed(os.path.join(HERE, "flow_small.npz"), **out)


def mlat_small():
    """MLAT frames + node lists (candidates visited in ascending segment order) of the small transparent scene."""
    from common import small_case
    c = small_case(width=48, height=32, transparent=True, n_lines=40)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    out = {}
    for k in (2, 8):
        img, nodes, _ = sc.render_rt_mlat(P, k)
        out["frame_k%d" % k] = img
        out["nodes_k%d" % k] = nodes
    np.savez_compressed(os.path.join(HERE, "mlat_small.npz"), **out)


RTAO_R2 = dict(ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0, ambient_occlusion_gamma=1.0,
               ambient_occlusion_radius=0.2, ambient_occlusion_distance_based=True, ambient_occlusion_iterations=2,
               ambient_occlusion_samples_per_frame=4)


def round2_case(kind):
    """The small cases of the round-2 fixtures (shared with tests/test_golden_round2.py)."""
    from linevis_amd import camera
    if kind == "eaw":
        return small_case(width=64, height=48, line_width=0.03, ambient_occlusion_denoiser="EAW", eaw_denoiser_iterations=2, **RTAO_R2)
    if kind == "svgf":
        return small_case(width=64, height=48, n_lines=6, pts_per_line=30, line_width=0.25, ambient_occlusion_denoiser="SVGF",
                          svgf_denoiser_iterations=3, use_jittered_primary_rays=True, **dict(RTAO_R2, ambient_occlusion_iterations=1))
    tr = scenes.twisted_ribbons(scenes.normalize(scenes.helix_bundle(n_lines=4, points_per_line=60, seed=3, turns=2.0)), twist=8.0)
    s = dict(use_ribbons=True, band_width=0.05, min_band_thickness=0.3, use_analytic_elliptic_tubes=(kind == "elliptic"))
    if kind == "elliptic":
        pts, seg, _ = lvo.build_tube_aabb_render_data_ribbons(tr.positions, tr.attributes, tr.line_offsets, 0.05, tr.ribbon_directions)
        s.update(RTAO_R2)
    else:   # "bands": circular tubes of a band data set
        pts, seg, _ = lvo.build_tube_aabb_render_data(tr.positions, tr.attributes, tr.line_offsets, 0.02)
    return Case(pts, seg, tfm.standard(), 80, 60, 0.02, **s)


SVGF_PATH = [(0.0, 0.0, 0.8), (0.0, 0.0, 0.8), (0.01, 0.0, 0.8), (0.02, 0.005, 0.79)]


def round2():
    """Round-2 features: EAW- and SVGF-denoised AO images (SVGF: after every frame of a short camera path), elliptic tubes and
    USE_BANDS frames, streamribbon directions of a small ABC-flow grid."""
    from linevis_amd import camera
    out = {}
    c = round2_case("eaw")
    img, ao = c.oracle_render(11)
    out["eaw_frame"], out["eaw_ao_bits"] = img, f2u(ao)
    c = round2_case("svgf")
    sc = c.oracle_scene()
    sv = lvo.Svgf(c.width, c.height, iterations=3)
    for f, pos in enumerate(SVGF_PATH):
        c.view, c.proj, c.fovy, c.near, c.far = camera.default_camera(c.width, c.height, pos)
        P = c.oracle_params(sc)
        ao = sv.step(lambda: sc.render_ao(P), P)
        out["svgf_ao_bits_%d" % f] = f2u(ao)
        out["svgf_raw_bits_%d" % f] = f2u(sv.raw)
        out["svgf_frame_%d" % f] = sc.render_rt(P, ao=ao)
    for kind in ("elliptic", "bands"):
        c = round2_case(kind)
        img, ao = c.oracle_render(11)
        out[kind + "_frame"] = img
        if ao is not None:
            out[kind + "_ao_bits"] = f2u(ao)
    c = round2_case("elliptic")
    rng = np.random.default_rng(9)
    cam = np.array([0.0, 0.0, 0.8], np.float32)
    d = rng.normal(size=(2000, 3)).astype(np.float32)
    d[:, 2] = -np.abs(d[:, 2]) * 4
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.tile(cam[None], (2000, 1))
    t, s = c.oracle_scene().trace_rays_elliptic(o, d, 1e-4, 1000.0, 0.05, 0.3, cam)
    out["elliptic_rays_d"], out["elliptic_rays_t_bits"], out["elliptic_rays_seg"] = d, f2u(t), s
    n = 16
    v = lvo.generate_abc_flow(n, n, n)
    sp = (1.0 / (n - 1),) * 3
    w = lvo.vorticity_field(v, sp)
    fields = [lvo.helicity_field(v, w), lvo.vector_magnitude_field(v), lvo.vector_magnitude_field(w)]
    seeds = np.array([[x, y, z] for z in (0.3, 0.7) for y in (0.3, 0.7) for x in (0.3, 0.7)], np.float32)
    pos, att, off, rib = lvo.trace_streamribbons(v, sp, fields, seeds, lvo.streamline_settings(minimum_length=0.2), 0)
    out["ribbons_seeds"], out["ribbons_pos_bits"], out["ribbons_off"], out["ribbons_dir_bits"] = seeds, f2u(pos), off, f2u(rib)
    out["ribbons_helicity_bits"] = f2u(fields[0])
    np.savez_compressed(os.path.join(HERE, "round2.npz"), **out)


def round2b_case(kind):
    """The small cases of the second round-2 fixture set (shared with tests/test_golden_round2.py)."""
    if kind in ("ppll_elliptic", "mlat_elliptic"):
        c = round2_case("elliptic")
        for k in list(c.settings):
            if k.startswith("ambient_occlusion"):
                del c.settings[k]
        c.tf = tfm.standard_transparent()
        if kind == "mlat_elliptic":
            c.settings.update(use_mlat=True, mlat_num_nodes=4)
        else:   # the fixture holds the entry hits of the analytic tubelets (round 2); since round 4 auto = the rasterised band prism
            c.settings["ppll_fragment_source"] = "capsule_entry"
        return c
    # rotating helicity bands: capsules ("helicity") or the triangle tubes ("helicity_tri", mesh returned too)
    tr = scenes.normalize(scenes.helix_bundle(n_lines=4, points_per_line=50, seed=4, turns=2.0))
    s_ = np.linspace(0.0, 1.0, len(tr.positions)).astype(np.float32)
    hel = (0.02 * np.sin(12.0 * s_ + 1.0) + 0.01).astype(np.float32)
    lw = 0.03
    pts, seg, _ = lvo.build_tube_aabb_render_data(tr.positions, tr.attributes, tr.line_offsets, lw, helicities=hel)
    st = dict(rotating_helicity_bands=True, helicity_rotation_factor=0.25)
    if kind == "helicity":
        return Case(pts, seg, tfm.standard(), 80, 60, lw, **st)
    mesh = lvo.build_tube_triangle_render_data(tr.positions, tr.attributes, tr.line_offsets, lw, 8, helicities=hel)
    return Case(pts, seg, tfm.standard(), 80, 60, lw, geometry_mode="Triangle Mesh", tube_num_subdivisions=8, **st), mesh


def round2b():
    """Second set: PPLL fragment lists and MLAT (canonical order) of band data, rotating helicity bands on capsules and on the
    triangle tubes (with UNIFORM_HELICITY_BAND_WIDTH)."""
    out = {}
    c = round2b_case("ppll_elliptic")
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    out["ppll_elliptic_frame"] = sc.render_ppll(P)
    nodes, start, cnt = sc.ppll_gather(P)
    lists = []
    for pix in np.nonzero(start != 0xFFFFFFFF)[0]:
        i = int(start[pix])
        while i != 0xFFFFFFFF:
            lists.append((int(pix), int(nodes[i, 1]), int(nodes[i, 0])))
            i = int(nodes[i, 2])
    out["ppll_elliptic_fragments"] = np.array(sorted(lists), dtype=np.uint32)      # (pixel, depth bits, colour), sorted
    c = round2b_case("mlat_elliptic")
    sc = c.oracle_scene()
    out["mlat_elliptic_frame"] = sc.render_rt_mlat(c.oracle_params(sc), 4)[0]
    c = round2b_case("helicity")
    out["helicity_frame"] = c.oracle_render(11)[0]
    out["helicity_rotation_bits"] = f2u(c.points["lineRotation"])
    c, mesh = round2b_case("helicity_tri")
    sc = c.oracle_scene()
    out["helicity_tri_frame"] = lvo.TriScene(*mesh, c.line_width).render_rt(sc, c.oracle_params(sc))
    out["helicity_tri_rotation_bits"] = f2u(mesh[2]["lineRotation"])
    np.savez_compressed(os.path.join(HERE, "round2b.npz"), **out)


def prism_small():
    """PPLL fragments of the rasterised programmable-pull prism (ppll_fragment_source = raster_prism) on the small
    scene of scene_small.npz: frame, fragment counter, and the sorted (pixel address, depth bits, packed colour) triples of every list;
    hexagon (default) and square cross-section."""
    g = np.load(os.path.join(HERE, "scene_small.npz"))
    pts = g["points"].view(lvo.LINE_POINT_DTYPE).reshape(-1)
    W, H, lw = int(g["width"]), int(g["height"]), float(g["line_width"])
    out = {}
    for name, settings in (("hex", {}), ("square_depthcue", dict(tube_num_subdivisions=4, depth_cue_strength=0.8))):
        c = Case(pts, g["seg"], g["tf_transparent"], W, H, lw, ppll_fragment_source="raster_prism", **settings)
        sc = c.oracle_scene()
        P = c.oracle_params(sc)
        st = lvo.Stats()
        out[name + "_frame"] = sc.render_ppll(P, stats=st)
        nodes, start, cnt = sc.ppll_gather(P)
        lists = []
        for pix in np.nonzero(start != 0xFFFFFFFF)[0]:
            i = int(start[pix])
            while i != 0xFFFFFFFF:
                lists.append((int(pix), int(nodes[i, 1]), int(nodes[i, 0])))
                i = int(nodes[i, 2])
        out[name + "_fragments"] = np.array(sorted(lists), dtype=np.uint32)
        out[name + "_count"] = np.uint64(cnt)
        out[name + "_max_depth_complexity"] = np.uint32(st.maxDepthComplexity)
    np.savez_compressed(os.path.join(HERE, "prism_small.npz"), **out)


def abi_smoke():
    """tests/golden/abi_smoke.bin: what tests/abi_smoke.c (plain C99 against the C-ABI) reads -- header, inputs, and the oracle's two
    frames (mode 11: RTAO 1 x 8 + depth cues; mode 2: PPLL with the transparent transfer function).  Little endian, no padding."""
    import struct
    W, H = 96, 64
    base = small_case(width=W, height=H, n_lines=14, pts_per_line=24, seed=5, line_width=0.02)
    rt = Case(base.points, base.seg, base.tf, W, H, base.line_width, ambient_occlusion_mode="RTAO (Screen Space)",
              ambient_occlusion_strength=1.0, ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=8,
              depth_cue_strength=0.8)
    frame_rt, _ = rt.oracle_render(11)
    tft = tfm.standard_transparent()
    pp = Case(base.points, base.seg, tft, W, H, base.line_width)
    frame_ppll, _ = pp.oracle_render(2)
    assert base.tf.shape == tft.shape
    with open(os.path.join(HERE, "abi_smoke.bin"), "wb") as f:
        f.write(struct.pack("<8I4f", 0x4D53564C, 1, len(base.points), len(base.seg), len(base.tf), W, H, 0,
                            base.line_width, rt.fovy, rt.near, rt.far))
        f.write(np.ascontiguousarray(rt.view, dtype=np.float32).tobytes())
        f.write(np.ascontiguousarray(rt.proj, dtype=np.float32).tobytes())
        f.write(base.points.tobytes())
        f.write(np.ascontiguousarray(base.seg, dtype=np.uint32).tobytes())
        f.write(np.ascontiguousarray(base.tf, dtype=np.float32).tobytes())
        f.write(np.ascontiguousarray(tft, dtype=np.float32).tobytes())
        f.write(np.ascontiguousarray(frame_rt, dtype=np.uint8).tobytes())
        f.write(np.ascontiguousarray(frame_ppll, dtype=np.uint8).tobytes())


if __name__ == "__main__":
    if "--only-abi-smoke" in sys.argv:
        abi_smoke()
        sys.exit(0)
    if "--only-prism" in sys.argv:
        prism_small()
        sys.exit(0)
    if "--only-round2b" in sys.argv:
        round2b()
        sys.exit(0)
    if "--only-round2" in sys.argv:
        round2()
        sys.exit(0)
    if "--only-mlat" in sys.argv:
        mlat_small()
        sys.exit(0)
    if "--only-flow" in sys.argv:
        flow_small()
        sys.exit(0)
    if "--only-triangle-tubes" in sys.argv:
        triangle_tubes()
        sys.exit(0)
    rng_kat()
    capsule_kat()
    a2_cases()
    small_scene()
    ppll_lists()
    lattice_c1()
    triangle_tubes()
    flow_small()
    mlat_small()
    round2()
    round2b()
    prism_small()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-24s %8d bytes" % (f, os.path.getsize(os.path.join(HERE, f))))
    abi_smoke()
