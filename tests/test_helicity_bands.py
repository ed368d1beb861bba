"""Rotating helicity bands of flow lines (USE_ROTATING_HELICITY_BANDS: LineDataFlow.cpp:535-550,601-624,2188-2197,2014-2028,2432-2440;
RayHitCommon.glsl:57-64,91-112,455-486; TubeRayTracing.glsl:551-567; LineAttributesBarycentric.glsl:43-92) on the CPU side: the host
layer's lineRotation against the oracle's builders, the data-set switches, and the oracle's stripe against an independent float64
restatement of drawSeparatorStripe."""
import numpy as np

from common import Case
from linevis_amd import host_api, scenes, transfer_function as tfm
from oracle import lvo


def helix_with_helicity(n_lines=5, pts=60, seed=4):
    tr = scenes.normalize(scenes.helix_bundle(n_lines=n_lines, points_per_line=pts, seed=seed, turns=2.0))
    rng = np.random.default_rng(seed)
    # a signed, smoothly varying "helicity" per point, small enough for a few turns of the bands per line
    s = np.linspace(0.0, 1.0, len(tr.positions)).astype(np.float32)
    hel = (0.02 * np.sin(12.0 * s + rng.uniform(0, 6)) + 0.01).astype(np.float32)
    return tr, hel


def test_host_line_rotation_equals_the_oracle_and_the_switches_follow_the_reference():
    tr, hel = helix_with_helicity()
    flow = host_api.LineDataFlow().set_trajectories_multi(tr.positions, np.stack([tr.attributes, hel]), ["Velocity Magnitude", "Helicity"],
                                                         tr.line_offsets)
    assert flow.has_helicity and not flow.use_rotating_helicity_bands
    assert abs(flow.max_helicity - float(np.abs(hel).max())) == 0.0
    plain_pts, _, _ = flow.tube_aabb_render_data(0.02)
    assert not plain_pts["lineRotation"].any()
    flow.set_new_settings(dict(rotating_helicity_bands=True))
    assert flow.use_rotating_helicity_bands
    pts, seg, aabb = flow.tube_aabb_render_data(0.02)
    ref = lvo.build_tube_aabb_render_data(tr.positions, tr.attributes, tr.line_offsets, 0.02, helicities=hel)
    assert np.array_equal(pts.view(np.uint8), ref[0].view(np.uint8)) and np.array_equal(seg, ref[1])
    assert np.abs(pts["lineRotation"]).max() > 3.0                      # several turns
    starts = np.asarray(tr.line_offsets[:-1])
    assert not pts["lineRotation"][starts].any()                        # restarts with every trajectory (LineDataFlow.cpp:2148)
    # triangle tubes: the rotation runs on across the trajectories (the reference's loop never resets it, :1994)
    idx, verts, tp = flow.tube_triangle_render_data(0.02, 6)
    ridx, rverts, rtp = lvo.build_tube_triangle_render_data(tr.positions, tr.attributes, tr.line_offsets, 0.02, 6, helicities=hel)
    assert np.array_equal(tp.view(np.uint8), rtp.view(np.uint8)) and np.array_equal(idx, ridx)
    assert tp["lineRotation"][starts[1]] != 0.0
    # the symmetric range of an attribute called "Helicity" (:522-526)
    flow.L.lvh_flow_set_selected_attribute(flow.h, 1)
    lo, hi = flow.attribute_range()
    assert lo == -hi and hi == np.float32(np.abs(hel).max())
    flow.set_new_settings(dict(rotating_helicity_bands=False))            # static switch, as in the reference: leave it off
    # a data set without such an attribute switches the bands off again (:549-551)
    flow2 = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    flow2.set_new_settings(dict(rotating_helicity_bands=True))
    assert not flow2.has_helicity and not flow2.use_rotating_helicity_bands
    flow2.set_new_settings(dict(rotating_helicity_bands=False))


def test_separator_scale_of_the_uniform_band_width():
    """UNIFORM_HELICITY_BAND_WIDTH: rotationSeparatorScale = cos(atan(rotDy * lineWidth / 2, rotDx)) through the build-owned atan2 and
    cos = rotDx / hypot(...) to float32 accuracy (LineAttributesBarycentric.glsl:94-112)."""
    import ctypes as C
    L = lvo.lib()
    L.lvo_atan2_det.restype = C.c_float
    L.lvo_atan2_det.argtypes = [C.c_float, C.c_float]
    L.lvo_sincos_rad.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    rng = np.random.default_rng(5)
    worst = 0.0
    for _ in range(2000):
        y = float(np.float32(rng.normal() * 10.0 ** rng.uniform(-4, 1)))
        x = float(np.float32(abs(rng.normal()) * 10.0 ** rng.uniform(-4, 0) + 1e-6))
        s, c = C.c_float(), C.c_float()
        L.lvo_sincos_rad(L.lvo_atan2_det(y, x), C.byref(s), C.byref(c))
        worst = max(worst, abs(c.value - x / np.hypot(x, y)))
    assert worst < 2e-6


def _smoothstep(e0, e1, x):
    t = np.clip((x - e0) / (e1 - e0), 0.0, 1.0)
    return t * t * (3.0 - 2.0 * t)


def test_oracle_stripes_against_a_float64_restatement():
    """One capsule along x seen from +z, no halos / no depth cue: every pixel's colour with the bands on equals the colour with
    the bands off times drawSeparatorStripe's factor, evaluated here in float64 from the pixel's own phi and rotation."""
    n = 2
    pts = np.zeros(n, dtype=lvo.LINE_POINT_DTYPE)
    pts["linePosition"] = [[-0.2, 0.0, 0.0], [0.2, 0.0, 0.0]]
    pts["lineTangent"] = [[1, 0, 0]] * 2
    pts["lineNormal"] = [[0, 1, 0]] * 2
    pts["lineAttribute"] = [0.3, 0.7]
    pts["lineRotation"] = [0.0, 7.0]
    seg = np.array([[0, 1]], np.uint32)
    lw = 0.1
    kw = dict(use_halos=False, use_capped_tubes=False, band_subdivisions=6, separator_width=0.2, helicity_rotation_factor=1.5)
    on = Case(pts, seg, tfm.standard(), 240, 160, lw, rotating_helicity_bands=True, **kw)
    off = Case(pts, seg, tfm.standard(), 240, 160, lw, **kw)
    sc = on.oracle_scene()
    Pon, Poff = on.oracle_params(sc), off.oracle_params(sc)
    a = sc.render_rt(Pon).astype(np.float64)
    b = sc.render_rt(Poff).astype(np.float64)
    assert (a[..., :3] <= b[..., :3] + 1e-9).all() and (a != b).any()
    # per pixel: hit point of the central ray on the cylinder |(y, z)| = r  ->  phi, rotation, depth
    view = np.asarray(on.view, np.float64).reshape(4, 4).T
    proj = np.asarray(on.proj, np.float64).reshape(4, 4).T
    iv, ip = np.linalg.inv(view), np.linalg.inv(proj)
    cam = (iv @ np.array([0, 0, 0, 1.0]))[:3]
    r = lw * 0.5
    checked = 0
    for y in range(0, 160, 2):
        for x in range(0, 240, 2):
            if (b[y, x, :3] == 255).all():
                continue
            ndc = np.array([2.0 * (x + 0.5) / 240 - 1.0, 2.0 * (y + 0.5) / 160 - 1.0, 1.0, 1.0])
            tgt = ip @ ndc
            d = (iv @ np.append(tgt[:3] / np.linalg.norm(tgt[:3]), 0.0))[:3]
            A = d[1] ** 2 + d[2] ** 2
            B = 2 * (cam[1] * d[1] + cam[2] * d[2])
            Cq = cam[1] ** 2 + cam[2] ** 2 - r * r
            disc = B * B - 4 * A * Cq
            if disc <= 0:
                continue
            t = (-B - np.sqrt(disc)) / (2 * A)
            hit = cam + t * d
            ts = (hit[0] + 0.2) / 0.4
            if not (0.02 < ts < 0.98):
                continue
            nrm = np.array([0.0, hit[1], hit[2]]) / r
            phi = np.arccos(np.clip(nrm[1], -1, 1))                       # angle to the line normal (0, 1, 0)
            if np.dot([0, 1, 0], np.cross(nrm, [1, 0, 0])) < 0:
                phi = 2 * np.pi - phi
            rot = ts * 7.0 * 1.5
            depth = np.linalg.norm(hit - cam)
            eps_o = np.clip(depth / lw * 0.05 / 160 * on.fovy, 0, 0.49)
            period = 2.0 / 6 * np.pi
            vf = np.mod(phi + rot + 0.1, period)
            aaf = eps_o * 10
            m = max(_smoothstep(aaf, 0.0, vf), _smoothstep(0.2 - aaf * 0.5, 0.2 + aaf * 0.5, vf))
            # near a stripe edge the float32 phi of the oracle moves the factor; compare where it is flat
            vf2 = np.mod(phi + rot + 0.1 + 0.004, period)
            vf3 = np.mod(phi + rot + 0.1 - 0.004, period)
            m2 = max(_smoothstep(aaf, 0.0, vf2), _smoothstep(0.2 - aaf * 0.5, 0.2 + aaf * 0.5, vf2))
            m3 = max(_smoothstep(aaf, 0.0, vf3), _smoothstep(0.2 - aaf * 0.5, 0.2 + aaf * 0.5, vf3))
            if abs(m - m2) > 0.02 or abs(m - m3) > 0.02:
                continue
            assert np.abs(a[y, x, :3] - b[y, x, :3] * m).max() <= 1.5, (x, y, m, a[y, x], b[y, x])
            checked += 1
    assert checked > 250
    dark = (a[..., :3].sum(axis=2) < 0.5 * b[..., :3].sum(axis=2)).sum()
    assert dark > 100                                                     # the separators are there


def _twist_image(w=64, h=4, seed=2):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    img[..., 3] = rng.integers(128, 256, size=(h, w))
    return img


def test_twist_line_texture_sampling_against_numpy():
    """The build-owned sampler of the twist-line texture (USE_HELICITY_BANDS_TEXTURE): REPEAT addressing, texel centres at (i + 0.5) / size,
    v = 0.5; level 0 for the ray tracer; textureGrad: lambda = log2(max(|dudx|, |dudy|) w), nearest mip = ceil(lambda + 0.5) - 1, the
    chain = 2 x 2 box averages -- against a float64 restatement."""
    img = _twist_image()
    h, w = img.shape[:2]
    levels = [img.astype(np.float64) / 255.0]
    while len(levels) < int(np.log2(max(w, h))):
        a = levels[-1]
        hh, ww = max(a.shape[0] // 2, 1), max(a.shape[1] // 2, 1)
        ii = np.minimum(np.arange(ww)[:, None] * 2 + np.arange(2)[None, :], a.shape[1] - 1)
        jj = np.minimum(np.arange(hh)[:, None] * 2 + np.arange(2)[None, :], a.shape[0] - 1)
        levels.append(np.stack([[a[jj[j]][:, ii[i]].mean(axis=(0, 1)) for i in range(ww)] for j in range(hh)]))

    def level(l, u, linear):
        a = levels[l]
        hh, ww = a.shape[:2]
        if not linear:
            return a[int(np.floor(0.5 * hh)) % hh, np.floor(u * ww).astype(int) % ww]
        x, y = u * ww - 0.5, 0.5 * hh - 0.5
        i0, j0 = np.floor(x).astype(int), int(np.floor(y))
        fx, fy = (x - i0)[:, None], y - j0
        r0 = a[j0 % hh, i0 % ww] * (1 - fx) + a[j0 % hh, (i0 + 1) % ww] * fx
        r1 = a[(j0 + 1) % hh, i0 % ww] * (1 - fx) + a[(j0 + 1) % hh, (i0 + 1) % ww] * fx
        return r0 * (1 - fy) + r1 * fy

    rng = np.random.default_rng(4)
    u = rng.random(400).astype(np.float32)
    dx = (rng.random(400) * 0.2).astype(np.float32) * (rng.random(400) < 0.8)
    dy = (rng.random(400) * 0.05).astype(np.float32)
    for mode in lvo.TWIST_FILTER_MODES:
        linear = mode.startswith("Linear")
        with lvo.twist_line_texture(img, mode):
            got0 = lvo.twist_line_sample(u)
            got = lvo.twist_line_sample(u, dx, dy)
        assert np.abs(got0 - level(0, u.astype(np.float64), linear)).max() < 2e-6
        rho = np.maximum(np.abs(dx), np.abs(dy)).astype(np.float64) * w
        lam = np.clip(np.log2(np.maximum(rho, 1.0)), 0, len(levels) - 1)
        if "Mipmap" not in mode:
            want = level(0, u.astype(np.float64), linear)
        elif mode.endswith("Mipmap Nearest"):
            d = np.clip(np.ceil(lam + 0.5) - 1, 0, len(levels) - 1).astype(int)
            edge = np.abs((lam + 0.5) - np.round(lam + 0.5)) < 1e-4       # level switches: float32 log2 may fall on the other side
            want = np.stack([level(int(d[k]), u[k:k + 1].astype(np.float64), linear)[0] for k in range(len(u))])
            got, want = got[~edge], want[~edge]
        else:
            lo = np.floor(lam).astype(int)
            hi = np.minimum(lo + 1, len(levels) - 1)
            t = (lam - lo)[:, None]
            want = np.stack([level(int(lo[k]), u[k:k + 1].astype(np.float64), linear)[0] for k in range(len(u))]) * (1 - t) + \
                np.stack([level(int(hi[k]), u[k:k + 1].astype(np.float64), linear)[0] for k in range(len(u))]) * t
        assert np.abs(got - want).max() < 5e-5, mode
