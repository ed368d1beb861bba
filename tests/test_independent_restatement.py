"""A SECOND, independent restatement of the three highest-risk pure functions of the hot path, written directly from the
GLSL text in float64 numpy / plain Python -- NOT through oracle/ -- and compared with the oracle on >= 10 000 random inputs:

  computeFragmentColor halo / outline block   Data/Shaders/Renderers/RayTracing/RayHitCommon.glsl:141-146,190-229,353-372,436-512
  blinnPhongShadingTube                       Data/Shaders/Utils/Lighting.glsl:100-191 (+ getAoFactor, AmbientOcclusion.glsl:84-99,
                                              getAntialiasingFactor, Antialiasing.glsl:1-3)
  frontToBackPQ / minHeapSink4                Data/Shaders/Renderers/PPLL/LinkedListSort.glsl:177-238

The oracle (oracle/lv_oracle.cpp) and the HIP kernels (csrc/lv_trace.h) were written by the same hand from the same
reading; a shared misreading is invisible to the parity tests.  This file is still the builder's reading of the GLSL, but
it is a separate text in another language and another precision: it breaks the single-source common mode.  The reference
holds no vectors for this path, so parity stays "unpinned" (DESIGN.md section 7).

The transfer-function texture and the camera are definitions the build owns; they are restated here
from DESIGN.md section 1, not from the oracle.
"""
import numpy as np
import pytest

from common import small_case
from oracle import lvo

N = 12000


# ---------------------------------------------------------------- GLSL built-ins in float64
def normalize(v):
    return v / np.sqrt((v * v).sum(axis=-1, keepdims=True))


def clamp(x, lo, hi):
    return np.minimum(np.maximum(x, lo), hi)


def mix(a, b, w):
    return a * (1.0 - w) + b * w


def smoothstep(e0, e1, x):
    t = clamp((x - e0) / (e1 - e0), 0.0, 1.0)
    return t * t * (3.0 - 2.0 * t)


def length(v):
    return np.sqrt((v * v).sum(axis=-1))


def dot(a, b):
    return (a * b).sum(axis=-1)


# ---------------------------------------------------------------- build-owned definitions
def transfer_function(tf, attr, attr_min, attr_max):
    """TransferFunction.glsl:60-71 with the build's texture: N RGBA texels, linear filter, centres (i + 0.5) / N, clamp."""
    n = tf.shape[0]
    pos = clamp((attr - attr_min) / (attr_max - attr_min), 0.0, 1.0)
    u = pos * n - 0.5
    i0 = np.floor(u)
    f = (u - i0)[:, None]
    a = np.clip(i0.astype(np.int64), 0, n - 1)
    b = np.clip(i0.astype(np.int64) + 1, 0, n - 1)
    return tf[a] * (1.0 - f) + tf[b] * f


def camera_position(view):
    """cameraPosition = (inverse(viewMatrix) * vec4(0, 0, 0, 1)).xyz; column-major 4x4."""
    m = np.asarray(view, dtype=np.float64).reshape(4, 4).T      # row-major maths matrix
    return np.linalg.inv(m)[:3, 3]


# ---------------------------------------------------------------- Lighting.glsl:100-191
def blinn_phong_shading_tube(base, frag_pos, ssp, n_in, t_in, cam, use_ao, ao_texel, ao_gamma, ao_strength, use_depth_cues,
                             min_depth, max_depth, depth_cue_strength):
    ambient = base[:, :3]
    diffuse = ambient
    if use_ao:
        # getAoFactor (AmbientOcclusion.glsl:84-99, non-SSAO branch) on the sampled texel
        ao = np.power(ao_texel, ao_gamma)
        ao = np.maximum(0.0, 1.0 - ao_strength + ao_strength * ao)
        kA = 0.2 + (1.0 - ao) * 0.5
        kD = 0.9 * ao
    else:
        ao = np.ones(len(base))
        kA = np.full(len(base), 0.1)
        kD = np.full(len(base), 0.9)
    kS, s = 0.3, 30.0
    Ia = kA[:, None] * ambient
    n = normalize(n_in)
    t = normalize(t_in)
    v = normalize(cam - frag_pos)
    l = v
    h = normalize(v + l)
    helper = normalize(np.cross(t, l))
    new_l = normalize(np.cross(helper, t))
    exponent = 1.7
    cos1 = np.power(clamp(np.abs(dot(n, l)), 0.0, 1.0), exponent)
    cos2 = np.power(clamp(np.abs(dot(n, new_l)), 0.0, 1.0), exponent)
    combined = 0.3 * cos1 + 0.7 * cos2
    Id = (kD * combined)[:, None] * diffuse
    Is = (kS * np.power(clamp(np.abs(dot(n, h)), 0.0, 1.0), s))[:, None] * np.ones(3)
    phong = Ia + Id + Is
    if use_ao:
        phong = phong * ao[:, None]
    if use_depth_cues:
        f = clamp((-ssp[:, 2] - min_depth) / (max_depth - min_depth), 0.0, 1.0)
        f = f * f * depth_cue_strength
        phong = mix(phong, 0.5, f[:, None])
    return np.concatenate([phong, base[:, 3:4]], axis=1)


# ---------------------------------------------------------------- RayHitCommon.glsl (tubes, flow lines, no bands)
def compute_fragment_color(tf, P, frag_pos, normal, tangent, is_cap, attribute, ao_texel):
    view = np.asarray(P.view[:], dtype=np.float64)
    cam = camera_position(view)
    frag_color = transfer_function(tf, attribute, P.attrMin, P.attrMax)                         # :126
    n = normalize(normal)                                                                       # :141
    v = normalize(cam - frag_pos)                                                               # :142
    t = normalize(tangent)                                                                      # :144
    helper = normalize(np.cross(t, v))                                                          # :146
    new_v = normalize(np.cross(helper, t))                                                      # :147
    rp = np.zeros(len(frag_pos))
    if P.useHalos:
        # isCap branch, :195-229
        c_vn = np.cross(v, n)
        rp_cap = length(c_vn)
        c_vn2 = np.cross(new_v, n)
        rp2 = length(c_vn2)
        neg = dot(t, c_vn) < 0.0
        rp2 = np.where(neg, -rp2, rp2)
        rp_cap = np.where(neg, -rp_cap, rp_cap)
        rp2 = clamp(rp2, -1.0, 1.0)
        rp_cap = np.where(np.abs(rp2) < np.abs(rp_cap), rp2, rp_cap)
        # tube-mantle branch, :353-372
        c_nv = np.cross(new_v, n)
        rp_tube = length(c_nv)
        rp_tube = np.where(dot(t, c_nv) < 0.0, -rp_tube, rp_tube)
        rp_tube = clamp(rp_tube, -1.0, 1.0)
        capped = bool(P.useCappedTubes)
        rp = np.where(is_cap & capped, rp_cap, rp_tube)
    m = view.reshape(4, 4).T
    ssp = (m @ np.concatenate([frag_pos, np.ones((len(frag_pos), 1))], axis=1).T).T[:, :3]     # :390
    shaded = blinn_phong_shading_tube(frag_color, frag_pos, ssp, n, t, cam, bool(P.useAmbientOcclusion), ao_texel,
                                      P.aoGamma, P.aoStrength, bool(P.useDepthCues), P.minDepth, P.maxDepth,
                                      P.depthCueStrength)                                       # :416-427
    abs_coords = np.abs(rp) if P.useHalos else np.zeros(len(rp))                                # :438-442
    depth = length(frag_pos - cam)                                                              # :444

    def aaf(d):                                                                                 # Antialiasing.glsl:1-3
        return d / float(P.height) * P.fovY
    eps_outline = clamp(aaf(depth / P.lineWidth * 0.05), 0.0, 0.49)                             # :451
    eps_white = clamp(aaf(depth / P.lineWidth * 2.0), 0.0, 0.49)                                # :452
    white_threshold = 0.7                                                                       # :490
    coverage = 1.0 - smoothstep(1.0 - eps_outline, 1.0, abs_coords) if P.useHalos else np.ones(len(rp))   # :494
    fg = 1.0 - np.asarray(P.background[:], dtype=np.float64)                                    # LineData.cpp:1282-1283
    w = smoothstep(white_threshold - eps_white, white_threshold + eps_white, abs_coords)        # :505-508
    rgb = mix(shaded[:, :3], fg[:3], w[:, None])
    a = shaded[:, 3] * coverage
    return np.concatenate([rgb, a[:, None]], axis=1), depth                                     # :539-541


def random_inputs(rng, n):
    frag = rng.uniform(-0.25, 0.25, (n, 3))
    normal = rng.normal(size=(n, 3)) * rng.uniform(0.2, 3.0, (n, 1))      # not unit: the shader normalises
    tangent = rng.normal(size=(n, 3)) * rng.uniform(0.2, 3.0, (n, 1))
    # make most normals roughly perpendicular to the tangent, as on a tube; keep some arbitrary ones
    tt = tangent / np.linalg.norm(tangent, axis=1, keepdims=True)
    perp = normal - (normal * tt).sum(axis=1, keepdims=True) * tt
    normal = np.where(rng.uniform(size=(n, 1)) < 0.8, perp, normal)
    is_cap = rng.uniform(size=n) < 0.3
    attr = rng.uniform(-0.1, 1.1, n)
    ao = rng.uniform(0.0, 1.0, n)
    return frag, normal, tangent, is_cap, attr, ao


@pytest.mark.parametrize("variant", ["plain", "ao", "ao_depthcue_gamma", "no_halos", "uncapped"])
def test_compute_fragment_color_against_the_float64_restatement(variant):
    settings = {
        "plain": {},
        "ao": dict(ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0),
        "ao_depthcue_gamma": dict(ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=0.7,
                                  ambient_occlusion_gamma=2.2, depth_cue_strength=0.8),
        "no_halos": dict(use_halos=False),
        "uncapped": dict(use_capped_tubes=False),
    }[variant]
    c = small_case(width=160, height=90, line_width=0.004, background=(0.9, 0.95, 1.0, 1.0), **settings)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    if P.useDepthCues:
        assert P.maxDepth > P.minDepth
    rng = np.random.default_rng(20260928)
    frag, normal, tangent, is_cap, attr, ao = random_inputs(rng, N)
    got, got_t = sc.compute_fragment_color(P, frag, normal, tangent, is_cap, attr, ao)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)   # the oracle sees float32 inputs
    want, want_t = compute_fragment_color(c.tf.astype(np.float64), P, f32(frag), f32(normal), f32(tangent), is_cap, f32(attr),
                                          f32(ao))
    err = np.abs(got.astype(np.float64) - want)
    # float32 evaluation of ~60 operations incl. three pow: a few 1e-6; 2e-4 is < 1/20 of an RGBA8 step
    assert err.max() < 2e-4, (variant, err.max(), np.unravel_index(err.argmax(), err.shape))
    assert np.abs(got_t.astype(np.float64) - want_t).max() < 1e-6
    # the sample exercises every branch
    assert is_cap.sum() > 1000 and (~is_cap).sum() > 1000
    assert (want[:, 3] < 0.999).sum() > 50 or not P.useHalos       # outline coverage < 1 somewhere


# ---------------------------------------------------------------- LinkedListSort.glsl:177-238 in plain Python
def unpack_unorm4x8(p):
    return [float((p >> s) & 0xFF) / 255.0 for s in (0, 8, 16, 24)]


def min_heap_sink4(depth, color, x, count):
    while True:
        t = 4 * x + 1
        if not t < count:
            return
        c = t + 1 if (t + 1 < count and depth[t] > depth[t + 1]) else t
        if t + 2 < count and depth[c] > depth[t + 2]:
            c = t + 2
        if t + 3 < count and depth[c] > depth[t + 3]:
            c = t + 3
        if depth[x] <= depth[c]:
            return
        depth[x], depth[c] = depth[c], depth[x]
        color[x], color[c] = color[c], color[x]
        x = c


def front_to_back_pq(color, depth):
    count = len(color)
    i = count // 4
    while i > 0:
        min_heap_sink4(depth, color, i, count)
        i -= 1
    ray = [0.0, 0.0, 0.0, 0.0]
    i = 0
    while i < count and ray[3] < 0.99:
        min_heap_sink4(depth, color, 0, count - i)
        i += 1
        src = unpack_unorm4x8(color[0])
        for k in range(3):
            ray[k] = ray[k] + (1.0 - ray[3]) * src[3] * src[k]
        ray[3] = ray[3] + (1.0 - ray[3]) * src[3]
        color[0] = color[count - i]
        depth[0] = depth[count - i]
    a = ray[3]
    return ([r / a for r in ray[:3]] + [a]) if a > 0.0 else None


def test_front_to_back_pq_against_the_plain_python_restatement():
    """10 000 pixels with random lists (0..MAX_NUM_FRAGS+ fragments, random depths incl. exact ties, random RGBA8): the oracle's
    literal resolve (depth-only comparisons, as the GLSL) against the restatement above; float64 blend vs float32 blend
    may differ by one RGBA8 step."""
    rng = np.random.default_rng(4242)
    W, H, max_frags = 100, 100, 24
    c = small_case(width=W, height=H, transparent=True, ppll_max_num_frags=max_frags, background=(0.2, 0.4, 0.6, 1.0))
    P = c.oracle_params()
    pw, ph = c.padded()
    start = np.full(pw * ph, 0xFFFFFFFF, dtype=np.uint32)
    nodes = []
    lists = {}
    for y in range(H):
        for x in range(W):
            k = int(rng.integers(0, max_frags + 6))            # some lists are longer than MAX_NUM_FRAGS
            if k == 0:
                continue
            depth = rng.uniform(0.3, 1.3, k).astype(np.float32)
            if k > 3 and rng.uniform() < 0.3:
                depth[rng.integers(0, k, 2)] = depth[0]        # exact ties
            col = rng.integers(0, 1 << 32, k, dtype=np.uint64).astype(np.uint32)
            if rng.uniform() < 0.2:
                col |= np.uint32(0xFF000000)                   # opaque layers: early out at alpha >= 0.99
            nxt = 0xFFFFFFFF
            for j in range(k):                                 # list order = reverse insertion order
                nodes.append((int(col[j]), int(depth[j].view(np.uint32)), nxt))
                nxt = len(nodes) - 1
            start[lvo.ppll_addr(x, y, pw, P.ppllTileW, P.ppllTileH)] = nxt
            lists[(x, y)] = (col, depth)
    nodes = np.asarray(nodes, dtype=np.uint32).reshape(-1, 3)
    got = lvo.ppll_resolve(P, nodes, start, literal=True)
    bg = [float(b) for b in P.background[:]]
    worst = 0
    for (x, y), (col, depth) in lists.items():
        # the resolve reads the first MAX_NUM_FRAGS nodes in list order (LinkedListResolve.glsl:66-80)
        order = list(range(len(col) - 1, -1, -1))[:max_frags]
        res = front_to_back_pq([int(col[j]) for j in order], [float(depth[j]) for j in order])
        if res is None:
            want = bg
        else:   # straight alpha, then BACK_TO_FRONT_STRAIGHT_ALPHA over the clear colour (PerPixelLinkedListLineRenderer.cpp:70)
            a = res[3]
            want = [res[k] * a + bg[k] * (1.0 - a) for k in range(3)] + [a + bg[3] * (1.0 - a)]
        want8 = [int(np.floor(min(max(v, 0.0), 1.0) * 255.0 + 0.5)) for v in want]
        worst = max(worst, max(abs(int(got[y, x, k]) - want8[k]) for k in range(4)))
    assert len(lists) >= 9000 and worst <= 1, worst
    empty = [(x, y) for y in range(H) for x in range(W) if (x, y) not in lists]
    for x, y in empty[:50]:
        assert list(got[y, x]) == [int(np.floor(b * 255.0 + 0.5)) for b in bg]


# ------------------------------------------------------------------ SVGF: one a-trous pass (SVGF.glsl:393-495, svgf_common.glsl)
def svgf_compute_weight(cd, od, phi_depth, cn, on, cc, oc, phi_color):
    weight_n = max(0.0, float(np.dot(cn, on))) ** 128
    weight_z = 0.0 if phi_depth == 0 else abs(cd - od) / phi_depth
    weight_c = abs(cc - oc) * 2 / phi_color
    return np.exp(0.0 - max(weight_c, 0.0) - max(weight_z, 0.0)) * weight_n


def svgf_atrous_pass(color, normal, depth, fwidth, iteration):
    """color: (H, W, 2) = {colour, variance}; texel fetches outside the image return 0."""
    h, w = depth.shape
    step = 1 << iteration
    kv = [1.0, 2.0 / 3.0, 1.0 / 6.0]
    vk = [[1.0 / 4.0, 1.0 / 8.0], [1.0 / 8.0, 1.0 / 16.0]]
    out = np.zeros_like(color)
    for y in range(h):
        for x in range(w):
            fv = 0.0
            for yy in (-1, 0, 1):
                for xx in (-1, 0, 1):
                    px, py = x + xx, y + yy
                    if 0 <= px < w and 0 <= py < h:
                        fv += color[py, px, 1] * vk[abs(xx)][abs(yy)]
            phi_color = np.sqrt(max(0.0, 1e-10 + fv))
            acc = kv[0] * kv[0]
            s = color[y, x] * acc
            for yy in range(-2, 3):
                for xx in range(-2, 3):
                    ox, oy = x + xx * step, y + yy * step
                    if not (0 <= ox < w and 0 <= oy < h) or (xx == 0 and yy == 0):
                        continue
                    k = kv[abs(xx)] * kv[abs(yy)]
                    wgt = svgf_compute_weight(depth[y, x], depth[oy, ox], abs(fwidth[y, x] * np.hypot(xx, yy) * step) + 0.0001,
                                              normal[y, x, :3], normal[oy, ox, :3], color[y, x, 0], color[oy, ox, 0], phi_color) * k
                    s = s + np.array([wgt, wgt * wgt]) * color[oy, ox]
                    acc += wgt
            out[y, x] = s / np.array([acc, acc * acc])
    return out


def test_svgf_first_frame_against_the_float64_restatement():
    """First frame (empty history): the reprojection fails everywhere, so temp_accum = {noisy, 0} and moments = {c, c^2, 1}; the
    moments filter (history length < 4) then estimates a spatial variance, boosted by 4 / 1, and the a-trous passes follow."""
    rng = np.random.default_rng(11)
    h, w = 20, 26
    normal = np.zeros((h, w, 4), np.float32)
    ang = rng.uniform(-0.15, 0.15, (h, w, 2))
    n3 = np.stack([np.sin(ang[..., 0]), np.sin(ang[..., 1]), np.ones((h, w))], axis=-1)
    normal[..., :3] = (n3 / np.linalg.norm(n3, axis=-1, keepdims=True)).astype(np.float32)
    normal[:, 13:, :3] = np.array([0.0, 1.0, 0.0], np.float32)             # an edge the filter must not cross
    depth = (0.7 + 0.0005 * rng.standard_normal((h, w))).astype(np.float32)
    fwidth = rng.uniform(0.0, 0.3, (h, w)).astype(np.float32)
    noisy = np.clip(0.6 + 0.2 * rng.standard_normal((h, w)), 0, 1).astype(np.float32)
    noisy[:, 13:] *= 0.5
    flow = np.zeros((h, w, 2), np.float32)
    # float64 restatement of the chain
    n64, d64, f64, c64 = normal.astype(np.float64), depth.astype(np.float64), fwidth.astype(np.float64), noisy.astype(np.float64)
    temp = np.stack([c64, np.zeros_like(c64)], axis=-1)
    m = np.stack([c64, c64 * c64], axis=-1)
    filt = np.zeros_like(temp)
    for y in range(h):
        for x in range(w):
            sw, sc_, sm = 0.0, 0.0, np.zeros(2)
            for yy in range(-3, 4):
                for xx in range(-3, 4):
                    ox, oy = x + xx, y + yy
                    if 0 <= ox < w and 0 <= oy < h:
                        wgt = svgf_compute_weight(d64[y, x], d64[oy, ox], abs(f64[y, x]) + 0.0001, n64[y, x, :3], n64[oy, ox, :3],
                                                  temp[y, x, 0], temp[oy, ox, 0], 10)
                        sw += wgt
                        sc_ += wgt * temp[oy, ox, 0]
                        sm += wgt * m[oy, ox]
            sw = max(sw, 1e-6)
            sm /= sw
            filt[y, x] = (sc_ / sw, (sm[1] - sm[0] * sm[0]) * 4.0 / 1.0)
    want = filt
    for it in range(3):
        want = svgf_atrous_pass(want, n64, d64, f64, it)
    hist = [np.zeros((h, w), np.float32), np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32), np.zeros((h, w), np.float32)]
    out = np.zeros((h, w), np.float32)
    lvo.lib().lvo_svgf_denoise(w, h, lvo._p(noisy), lvo._p(normal), lvo._p(depth), lvo._p(fwidth), lvo._p(flow), 3, 0.002, 0.02,
                               lvo._p(hist[0]), lvo._p(hist[1]), lvo._p(hist[2]), lvo._p(hist[3]), lvo._p(out))
    assert np.abs(out - want[..., 0]).max() < 2e-5
    assert np.all(hist[1][..., 2] == 1.0)
    # the edge survives: pow(dot, 128) of perpendicular normals is 0
    assert abs(out[:, :13].mean() - out[:, 13:].mean()) > 0.2


# ------------------------------------------------------------------ the RTAO chain end to end (a5, a7, a12, a13) in float64
def _tea(v0, v1):
    M, s0 = 0xFFFFFFFF, 0
    for _ in range(16):
        s0 = (s0 + 0x9e3779b9) & M
        v0 = (v0 + ((((v1 << 4) & M) + 0xa341316c) & M ^ ((v1 + s0) & M) ^ (((v1 >> 5) + 0xc8013ea4) & M))) & M
        v1 = (v1 + ((((v0 << 4) & M) + 0xad90777d) & M ^ ((v0 + s0) & M) ^ (((v0 >> 5) + 0x7e95761e) & M))) & M
    return v0


def _rnd2(seed):
    out = []
    for _ in range(2):
        seed = (1664525 * seed + 1013904223) & 0xFFFFFFFF
        out.append((seed & 0x00FFFFFF) / float(0x01000000))
    return out


def _ray_sphere(o, d, c, r):
    A = np.dot(d, d)
    B = 2.0 * np.dot(d, o - c)
    Cc = np.dot(o - c, o - c) - r * r
    disc = B * B - 4 * A * Cc
    if disc < 0:
        return None
    s = np.sqrt(disc)
    t0, t1 = (-B - s) / (2 * A), (-B + s) / (2 * A)
    return t0 if t0 >= 0 else (t1 if t1 >= 0 else None)


def _ray_tube(o, d, p0, p1, r):
    td = (p1 - p0) / np.linalg.norm(p1 - p0)
    dp = o - p0
    a = d - np.dot(d, td) * td
    b = dp - np.dot(dp, td) * td
    A, B, Cc = np.dot(a, a), 2.0 * np.dot(a, b), np.dot(b, b) - r * r
    disc = B * B - 4 * A * Cc
    if disc < 0 or A == 0:
        return None
    s = np.sqrt(disc)
    for t in ((-B - s) / (2 * A), (-B + s) / (2 * A)):
        if t >= 0:
            pos = o + t * d
            if np.dot(td, pos - p0) > 0 and np.dot(td, pos - p1) < 0:
                return t
    return None


def _closest_hit(o, d, P0, P1, r, t_min, t_max):
    """IntersectionTube (TubeRayTracing.glsl:452-494) over all segments + reportIntersectionEXT's interval; returns
    (t, segment, kind, margin) with margin = how clearly the winner wins (for excluding borderline pixels)."""
    best, second = None, np.inf
    for k in range(len(P0)):
        hit_t, kind = 1e7, 0
        t = _ray_tube(o, d, P0[k], P1[k], r)
        has = t is not None
        if has:
            hit_t = t
        for kk, c in ((1, P0[k]), (2, P1[k])):
            s = _ray_sphere(o, d, c, r)
            if s is not None and s < hit_t:
                has, hit_t, kind = True, s, kk
        if has and t_min <= hit_t <= t_max:
            if best is None or hit_t < best[0]:
                second = best[0] if best is not None else second
                best = (hit_t, k, kind)
            else:
                second = min(second, hit_t)
    return best, second


def test_rtao_chain_against_the_float64_restatement():
    """Primary ray (jittered, seeded per pixel), capsule closest hit, hit frame, 4 hemisphere samples per pixel, AO rays in
    [0, radius] with distance weighting, mean -- restated in float64 numpy from the GLSL for a 32 x 24 frame over 80 segments
    and compared with the oracle's AO image wherever no decision along the way is borderline."""
    from linevis_amd import camera
    c = small_case(width=32, height=24, n_lines=8, pts_per_line=11, line_width=0.1, seed=5,
                   ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0, ambient_occlusion_radius=0.2,
                   ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=4)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    ao = sc.render_ao(P)
    W, H, r, spp, ao_radius = c.width, c.height, c.line_width * 0.5, 4, 0.2
    pts, seg = c.points, c.seg
    P0 = pts["linePosition"][seg[:, 0]].astype(np.float64)
    P1 = pts["linePosition"][seg[:, 1]].astype(np.float64)
    T0 = pts["lineTangent"][seg[:, 0]].astype(np.float64)
    T1 = pts["lineTangent"][seg[:, 1]].astype(np.float64)
    inv_view = np.linalg.inv(np.array(c.view, np.float64).reshape(4, 4).T)
    inv_proj = np.linalg.inv(np.array(c.proj, np.float64).reshape(4, 4).T)
    cam = (inv_view @ np.array([0, 0, 0, 1.0]))[:3]
    corr = np.cos(np.pi / 6.0)                      # subdivisionCorrectionFactor, 6 tube subdivisions
    checked = hits = 0
    for y in range(H):
        for x in range(W):
            pix = x + y * W
            xi = _rnd2(_tea(pix, 0))
            ndc = np.array([2 * (x + xi[0]) / W - 1, 2 * (y + xi[1]) / H - 1, 1.0, 1.0])
            tgt = (inv_proj @ ndc)[:3]
            d = (inv_view @ np.append(tgt / np.linalg.norm(tgt), 0.0))[:3]
            best, second = _closest_hit(cam, d, P0, P1, r, 1e-4, 1000.0)
            if best is None:
                if second == np.inf:
                    assert ao[y, x] == 1.0
                    checked += 1
                continue
            t, k, kind = best
            if second - t < 1e-4:
                continue                                        # two surfaces at (almost) the same depth
            hits += 1
            pos = cam + d * t
            v = P1[k] - P0[k]
            ts = np.dot(v, pos - P0[k]) / np.dot(v, v) if kind == 0 else (0.0 if kind == 1 else 1.0)
            line_pos = P0[k] + ts * v if kind == 0 else (P0[k] if kind == 1 else P1[k])
            n = (pos - line_pos) / np.linalg.norm(pos - line_pos)
            tg = (1 - ts) * T0[k] + ts * T1[k]
            tg /= np.linalg.norm(tg)
            bt = np.cross(n, tg)
            offset = np.linalg.norm(line_pos - pos) / corr
            total, clear = 0.0, True
            for s in range(spp):
                xi0, xi1 = _rnd2(_tea(pix, 0 * spp + s))
                rs = np.sqrt(1.0 - xi0 * xi0)
                smp = np.array([np.cos(2 * np.pi * xi1) * rs, np.sin(2 * np.pi * xi1) * rs, xi0])
                rd = tg * smp[0] + bt * smp[1] + n * smp[2]
                rd /= np.linalg.norm(rd)
                ah, asec = _closest_hit(pos + rd * offset, rd, P0, P1, r, 0.0, ao_radius)
                if ah is None:
                    total += 1.0
                    # a miss is only clear if nothing ends just outside the interval
                    far, _ = _closest_hit(pos + rd * offset, rd, P0, P1, r, 0.0, ao_radius * 1.01)
                    clear = clear and far is None
                else:
                    total += ah[0] / ao_radius
                    clear = clear and ah[0] > 1e-4
            if clear:
                assert abs(ao[y, x] - total / spp) < 2e-4, (x, y, ao[y, x], total / spp)
                checked += 1
    print('hit pixels', hits, 'checked', checked, 'of', W * H)
    assert hits > 100 and checked > 0.85 * W * H
