"""Parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs, against
the committed golden fixtures, and -- at BASELINE.json's full sizes -- through size-independent properties.

Bars (north_star): RGBA8 frames within +-2 LSB per channel; everything that is pure +,-,*,/,sqrt (ray generation,
traversal, ray-capsule hits, AO factors, PPLL fragment depths) and all integer work bit-exact.
"""
import os

import numpy as np
import pytest

from common import GOLDEN_DIR, ROOT, Case, small_case, max_lsb_diff, scene_arrays
from linevis_amd import capi, host_api, scenes, tiling, transfer_function as tfm
from oracle import lvo

pytestmark = pytest.mark.gpu

LSB_TOL = 2  # north_star: "+-2 LSB per RGBA8 channel"
RTAO = dict(ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0)


def G(name):
    return np.load(os.path.join(GOLDEN_DIR, name))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def golden_base():
    g = G("scene_small.npz")
    pts = g["points"].reshape(-1).view(lvo.LINE_POINT_DTYPE)
    return g, pts, int(g["width"]), int(g["height"]), float(g["line_width"])


# ---------------------------------------------------------------- rays / BVH
def test_rays_bit_exact_vs_golden_and_oracle(hip_lib):
    g, pts, W, H, lw = golden_base()
    c = Case(pts, g["seg"], g["tf"], W, H, lw)
    ctx = c.hip_context()
    t, s, k = ctx.trace_rays(g["ray_o"], g["ray_d"], 1e-4, 1000.0)
    assert np.array_equal(bits(t), g["ray_t_bits"]) and np.array_equal(s, g["ray_seg"]) and np.array_equal(k, g["ray_kind"])
    rng = np.random.default_rng(123)
    o = rng.uniform(-0.35, 0.35, (20000, 3)).astype(np.float32)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for (tmin, tmax) in [(0.0, 0.1), (1e-4, 1000.0)]:
        a = ctx.trace_rays(o, d, tmin, tmax)
        b = c.oracle_scene().trace_rays(o, d, tmin, tmax, lw, use_bvh=False)  # brute force = ground truth
        assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert (a[1] != 0xFFFFFFFF).sum() > 1000


def test_tie_break_lowest_segment(hip_lib):
    pts = np.zeros(3, dtype=lvo.LINE_POINT_DTYPE)
    pts["linePosition"] = [[-0.1, 0, 0], [0, 0, 0], [0.1, 0.02, 0]]
    pts["lineTangent"] = [[1, 0, 0]] * 3
    pts["lineNormal"] = [[0, 1, 0]] * 3
    c = Case(pts, np.array([[0, 1], [1, 2]], np.uint32), tfm.standard(), 32, 32, 0.02)
    ctx = c.hip_context()
    t, s, k = ctx.trace_rays(np.array([[0.0, -0.5, 0.0]], np.float32), np.array([[0.0, 1.0, 0.0]], np.float32), 1e-4, 10.0)
    assert s[0] == 0 and k[0] == 2


@pytest.mark.parametrize("build", [dict(), dict(accel_build="fast_build"), dict(treelet_leaves=3), dict(treelet_leaves=7),
                                   dict(treelet_leaves=64), dict(treelet_leaves=1024), dict(treelet_leaves=4096)])
def test_lbvh_structure(hip_lib, build):
    """default = fast_trace (every subtree of <= 512 leaves rebuilt by the binned SAH: here the 2940 segments form a handful of
    treelets); 1024 >= n / 3 exercises big treelets, 3 and 7 the smallest ones, fast_build the plain LBVH"""
    c = small_case(n_lines=60, pts_per_line=50, seed=5, line_width=0.01)
    ctx = c.hip_context()
    for k, v in build.items():
        ctx.set_option(k, v)
    ctx.build_accel()
    st = ctx.stats()
    n = len(c.seg)
    nw = st.num_nodes                                           # 4-wide nodes: one per even-depth binary node
    assert st.num_segments == n and (n - 1) // 3 <= nw <= n - 1 and 0 < st.bvh_depth <= 96
    nodes, leaf_seg = ctx.get_accel(nw, n)
    assert sorted(leaf_seg.tolist()) == list(range(n))          # every segment is exactly one leaf
    f = nodes.view(np.float32)
    child = nodes[:, 12:16]
    LEAF, INVALID = 0x80000000, 0xFFFFFFFF
    valid = child != INVALID
    nslots = valid.sum(axis=1)
    assert nslots.min() >= 2 and nslots.max() <= 4 and np.all(valid[:, 0] & valid[:, 1])
    # every wide node except the root is referenced exactly once, every leaf exactly once
    refs = child[valid]
    refs_internal = refs[(refs & LEAF) == 0]
    refs_leaf = refs[(refs & LEAF) != 0] & 0x7FFFFFFF
    assert sorted(refs_internal.tolist()) == list(range(1, nw))
    assert sorted(refs_leaf.tolist()) == list(range(n))
    # decoded (8-bit quantised) child boxes contain the capsule of every leaf child: culling stays conservative
    origin, scale = f[:, 0:3].astype(np.float64), f[:, 3:6].astype(np.float64)
    p = c.points["linePosition"]
    r = c.line_width * 0.5
    for node in range(nw):
        for k in range(4):
            ref = int(child[node, k])
            if ref != INVALID and ref & LEAF:
                seg = c.seg[leaf_seg[ref & 0x7FFFFFFF]]
                mn = np.minimum(p[seg[0]], p[seg[1]]) - r
                mx = np.maximum(p[seg[0]], p[seg[1]]) + r
                qmin = (nodes[node, 6:9] >> (8 * k)) & 0xFF
                qmax = (nodes[node, 9:12] >> (8 * k)) & 0xFF
                assert np.all(origin[node] + qmin * scale[node] <= mn + 1e-7)
                assert np.all(origin[node] + qmax * scale[node] >= mx - 1e-7)
            elif ref == INVALID:   # empty slot: inverted box, rejected by the slab test itself
                assert np.all(((nodes[node, 6:9] >> (8 * k)) & 0xFF) == 255)
                assert np.all(((nodes[node, 9:12] >> (8 * k)) & 0xFF) == 0)


# ---------------------------------------------------------------- golden frames
def test_golden_frames_ray_tracer(hip_lib):
    g, pts, W, H, lw = golden_base()
    c = Case(pts, g["seg"], g["tf"], W, H, lw, depth_cue_strength=0.8)
    ctx = c.hip_context()
    assert max_lsb_diff(ctx.render(11), g["rt_depthcue"]) <= LSB_TOL
    assert np.array_equal(bits(ctx.depth_range()), g["depth_range_bits"])
    c = Case(pts, g["seg"], g["tf_transparent"], W, H, lw, num_samples_per_frame=4)
    assert max_lsb_diff(c.hip_context().render(11), g["rt_transparent_spp4"]) <= LSB_TOL


def test_every_build_gives_the_same_hits_and_frames(hip_lib):
    """accel_build / treelet_leaves change the topology only: closest hits, AO factors and PPLL frames are identical -- also for a
    scene smaller than one treelet (the whole tree is rebuilt) and for duplicate segments (all box centres in one bin)."""
    rng = np.random.default_rng(12)
    c = small_case(width=160, height=104, n_lines=40, pts_per_line=40, line_width=0.012, transparent=True, **RTAO,
                   ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=6)
    o = rng.uniform(-0.4, 0.4, (20000, 3)).astype(np.float32)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    want = None
    for build in (dict(accel_build="fast_build"), dict(), dict(treelet_leaves=3), dict(treelet_leaves=100), dict(treelet_leaves=1024)):
        ctx = c.hip_context()
        for k, v in build.items():
            ctx.set_option(k, v)
        got = (ctx.trace_rays(o, d, 1e-4, 10.0), ctx.render(11), ctx.get_ao(), ctx.render(2))
        if want is None:
            want = got
            assert (got[0][1] != 0xFFFFFFFF).sum() > 300
        else:
            assert np.array_equal(bits(got[0][0]), bits(want[0][0])) and np.array_equal(got[0][1], want[0][1]), build
            assert np.array_equal(got[1], want[1]) and np.array_equal(bits(got[2]), bits(want[2])), build
            assert np.array_equal(got[3], want[3]), build
    # 64 identical segments: every split falls back to the middle of the range
    pts = np.zeros(128, dtype=lvo.LINE_POINT_DTYPE)
    pts["linePosition"][0::2] = [-0.1, 0.0, 0.0]
    pts["linePosition"][1::2] = [0.1, 0.0, 0.0]
    pts["lineTangent"] = [1, 0, 0]
    pts["lineNormal"] = [0, 1, 0]
    dup = Case(pts, np.arange(128, dtype=np.uint32).reshape(64, 2), tfm.standard_transparent(), 48, 32, 0.1)
    a = dup.hip_context()
    b = dup.hip_context()
    b.set_option("accel_build", "fast_build")
    assert np.array_equal(a.render(2), b.render(2))


def test_treelet_lane_builder_builds_the_same_tree(hip_lib):
    """treelet_group_leaves / treelet_lane_leaves (small ranges of a treelet built by groups of lanes / one lane per range) and
    treelet_plane_eval are build-time choices only: the compressed 4-wide tree and the
    leaf order are byte-identical to the build in which the wave splits every range (0), for small and large treelets, for a scene
    smaller than one treelet, and with more than 64 queued ranges per treelet (treelet_leaves = 4096, lane ranges of 2 ... 8)."""
    for scene in (small_case(n_lines=60, pts_per_line=50, seed=5, line_width=0.01),
                  small_case(n_lines=6, pts_per_line=9, seed=2, line_width=0.02),
                  small_case(n_lines=150, pts_per_line=80, seed=9, line_width=0.004)):
        n = len(scene.seg)
        for tl in (512, 7, 64, 4096):
            want = None
            for ll, ev, gl in ((0, "loop", -1), (0, "loop", 0), (0, "scan", 0), (2, "scan", 0), (3, "loop", 0), (4, "scan", 0), (8, "scan", 0), (16, "loop", 0),
                               (33, "scan", 0), (64, "scan", 0), (6, "scan", 8), (6, "loop", 16), (0, "scan", 16)):
                ctx = scene.hip_context()
                ctx.set_option("treelet_leaves", tl)
                ctx.set_option("treelet_lane_leaves", ll)
                ctx.set_option("accel_collapse_top", gl >= 0)   # (first variant: one pass per level of the wide tree from the root, rounds 1-3)
                gl = max(gl, 0)
                ctx.set_option("treelet_group_leaves", gl)   # 8 / 16: queued ranges built by groups of lanes (the default: 16 = two tiers)
                ctx.set_option("treelet_plane_eval", ev)   # loop = the round-3 form of the plane evaluation, the reference form
                ctx.build_accel()
                nodes, leaf_seg = ctx.get_accel(ctx.stats().num_nodes, n)
                if want is None:
                    want = (nodes.copy(), leaf_seg.copy())
                else:
                    assert nodes.shape == want[0].shape and np.array_equal(nodes, want[0]), (n, tl, ll, ev, gl)
                    assert np.array_equal(leaf_seg, want[1]), (n, tl, ll, ev, gl)


def test_golden_rtao(hip_lib):
    g, pts, W, H, lw = golden_base()
    c = Case(pts, g["seg"], g["tf"], W, H, lw, ambient_occlusion_mode="RTAO (Screen Space)",
             ambient_occlusion_strength=0.9, ambient_occlusion_gamma=1.5, ambient_occlusion_iterations=2,
             ambient_occlusion_samples_per_frame=8, ambient_occlusion_radius=0.1)
    ctx = c.hip_context()
    img = ctx.render(11)
    assert np.array_equal(bits(ctx.get_ao()), g["ao_bits"])     # AO factors: bit-exact
    assert max_lsb_diff(img, g["rt_ao"]) <= LSB_TOL


def test_golden_ppll(hip_lib):
    g, pts, W, H, lw = golden_base()
    c = Case(pts, g["seg"], g["tf_transparent"], W, H, lw, ppll_fragment_source="capsule_entry")   # the fixture of rounds 1-3
    ctx = c.hip_context()
    img = ctx.render(2)
    st = ctx.stats()
    assert st.fragments == int(g["ppll_fragments"]) and st.max_depth_complexity == int(g["ppll_max_depth_complexity"])
    assert max_lsb_diff(img, g["ppll"]) <= LSB_TOL


def test_golden_lattice_c1(hip_lib):
    g = G("lattice_c1.npz")
    tr = scenes.normalize(scenes.lattice())
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    pts, seg, _ = flow.tube_aabb_render_data(float(g["line_width"]))
    assert np.uint32(np.bitwise_xor.reduce(pts.view(np.uint32))) == g["points_crc"]
    c = Case(pts, seg, tfm.standard(), 128, 128, float(g["line_width"]), num_samples_per_frame=4)
    img = c.hip_context().render(11)
    assert max_lsb_diff(img, g["image"]) <= LSB_TOL
    # equal-mean check between two independent estimators (4 vs 16 spp), the reference's VPT test idea
    c2 = Case(pts, seg, tfm.standard(), 128, 128, float(g["line_width"]), num_samples_per_frame=16)
    img2 = c2.hip_context().render(11)
    m1 = img.reshape(-1, 4).astype(np.float64).mean(axis=0) / 255.0
    m2 = img2.reshape(-1, 4).astype(np.float64).mean(axis=0) / 255.0
    assert np.abs(m1 - m2).max() < 2e-3 + 1.0 / 255.0


# ---------------------------------------------------------------- live oracle comparisons
@pytest.mark.parametrize("settings", [
    dict(),
    dict(use_halos=False, use_capped_tubes=False),
    dict(num_samples_per_frame=3, use_deterministic_sampling=True),
    dict(depth_cue_strength=0.5, max_depth_complexity=2),
    dict(RTAO, ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=64),
    dict(RTAO, ambient_occlusion_iterations=3, ambient_occlusion_samples_per_frame=4, use_jittered_primary_rays=False),
    dict(RTAO, ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=5, ambient_occlusion_distance_based=False),
    dict(RTAO, ambient_occlusion_iterations=2, ambient_occlusion_samples_per_frame=3, ambient_occlusion_radius=0.05,
         tube_num_subdivisions=8, ambient_occlusion_gamma=2.0, ambient_occlusion_strength=0.7),
])
def test_ray_tracer_matches_oracle(hip_lib, settings):
    transparent = settings.get("max_depth_complexity") == 2
    c = small_case(width=112, height=80, seed=31, transparent=transparent, **settings)
    ctx = c.hip_context()
    img = ctx.render(11)
    ref, ao_ref = c.oracle_render(11)
    if ao_ref is not None:
        assert np.array_equal(bits(ctx.get_ao()), bits(ao_ref))
    assert max_lsb_diff(img, ref) <= LSB_TOL
    assert (img != np.uint8(255)).any()


@pytest.mark.parametrize("settings", [
    dict(),
    dict(ppll_tile_width=1, ppll_tile_height=1),
    dict(ppll_tile_width=8, ppll_tile_height=8, depth_cue_strength=0.8),
    dict(RTAO, ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=8),
    dict(ppll_max_num_frags=200),   # fragment arrays do not fit LDS -> global scratch variant
])
def test_ppll_matches_oracle(hip_lib, settings):
    c = small_case(width=100, height=70, seed=17, transparent=True, **settings)   # not a multiple of the PPLL tile
    ctx = c.hip_context()
    img = ctx.render(2)
    ref, _ = c.oracle_render(2)
    assert max_lsb_diff(img, ref) <= LSB_TOL
    # gather parity: per pixel the multiset of (colour, depth) fragments is identical, bit for bit
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    ao = sc.render_ao(P) if P.useAmbientOcclusion else None
    on, os_, ocnt = sc.ppll_gather(P, ao=ao)
    pw, ph = c.padded()
    hn, hs, hcnt = ctx.ppll_buffers(pw * ph, int(P.ppllLinkedListSize))
    assert hcnt == ocnt and hs.shape == os_.shape

    def lists(nodes, start):
        out = {}
        for pix in np.nonzero(start != 0xFFFFFFFF)[0]:
            l, i = [], int(start[pix])
            while i != 0xFFFFFFFF:
                l.append((int(nodes[i, 1]), int(nodes[i, 0])))
                i = int(nodes[i, 2])
            out[int(pix)] = sorted(l)
        return out
    assert lists(hn, hs) == lists(on, os_)


def test_ppll_resolve_given_lists(hip_lib):
    g = G("ppll_lists.npz")
    W, H = int(g["width"]), int(g["height"])
    c = Case(np.zeros(0, dtype=lvo.LINE_POINT_DTYPE), np.zeros((0, 2), np.uint32), tfm.standard(), W, H, 0.01,
             background=tuple(float(x) for x in g["background"]), ppll_max_num_frags=int(g["max_frags"]))
    ctx = c.hip_context()
    img = ctx.ppll_resolve(g["nodes"], g["start"])
    assert np.array_equal(img, g["resolved_key"])      # identical lists -> identical bytes (ties, overflow, alpha 0)


def test_ppll_pool_overflow(hip_lib):
    c = small_case(width=64, height=48, n_lines=80, line_width=0.05, transparent=True,
                   ppll_expected_avg_depth_complexity=1)
    ctx = c.hip_context()
    img = ctx.render(2)
    st = ctx.stats()
    pw, ph = c.padded()
    assert st.fragments > pw * ph           # more fragments than pool nodes: the counter keeps counting
    # the physical pool = the reference's linkedListSize (1 node / pixel here) + the chunk tail every wave of the gather may
    # leave unused, so the lists hold AT LEAST as many fragments as the reference's exact allocator would have stored
    pool = int(st.ppll_pool_nodes)
    assert pool > pw * ph
    nodes, start, cnt = ctx.ppll_buffers(pw * ph, pool)
    assert cnt == st.fragments
    linked = 0
    for head in start[start != 0xFFFFFFFF]:
        n, guard = int(head), 0
        while n != 0xFFFFFFFF:
            assert n < pool
            linked += 1
            guard += 1
            assert guard <= st.max_depth_complexity
            n = int(nodes[n, 2])
    assert min(int(st.fragments), pw * ph) <= linked <= min(int(st.fragments), pool) and img.shape == (48, 64, 4)


# ---------------------------------------------------------------- edge cases
def test_empty_scene_and_single_segment(hip_lib):
    empty = Case(np.zeros(0, dtype=lvo.LINE_POINT_DTYPE), np.zeros((0, 2), np.uint32), tfm.standard(), 40, 24, 0.01,
                 background=(0.25, 0.5, 0.75, 1.0))
    ctx = empty.hip_context()
    for mode in (11, 2):
        img = ctx.render(mode)
        assert np.array_equal(img, np.broadcast_to(np.array([64, 128, 191, 255], np.uint8), img.shape))
    pts = np.zeros(2, dtype=lvo.LINE_POINT_DTYPE)
    pts["linePosition"] = [[-0.2, 0.0, 0.0], [0.2, 0.05, 0.0]]
    pts["lineTangent"] = [[1, 0, 0]] * 2
    pts["lineNormal"] = [[0, 1, 0]] * 2
    pts["lineAttribute"] = [0.1, 0.9]
    one = Case(pts, np.array([[0, 1]], np.uint32), tfm.standard(), 64, 48, 0.05, **RTAO,
               ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=4)
    img = one.hip_context().render(11)
    ref, _ = one.oracle_render(11)
    assert max_lsb_diff(img, ref) <= LSB_TOL and (img != 255).any()


def test_duplicate_segments_and_fat_tubes(hip_lib):
    # 64 identical segments: identical Morton keys exercise the duplicate-key path of the LBVH build
    pts = np.zeros(128, dtype=lvo.LINE_POINT_DTYPE)
    pts["linePosition"][0::2] = [-0.1, 0.0, 0.0]
    pts["linePosition"][1::2] = [0.1, 0.0, 0.0]
    pts["lineTangent"] = [1, 0, 0]
    pts["lineNormal"] = [0, 1, 0]
    seg = np.arange(128, dtype=np.uint32).reshape(64, 2)
    c = Case(pts, seg, tfm.standard_transparent(), 48, 32, 0.1)
    ctx = c.hip_context()
    img = ctx.render(2)
    ref, _ = c.oracle_render(2)
    assert max_lsb_diff(img, ref) <= LSB_TOL
    assert ctx.stats().max_depth_complexity == 64
    t, s, k = ctx.trace_rays(np.array([[0, 0, 0.8]], np.float32), np.array([[0, 0, -1]], np.float32), 1e-4, 10.0)
    assert s[0] == 0     # all 64 tie; lowest index wins


def test_deep_lbvh_uses_stack_overflow_slab(hip_lib):
    """Segments clustered at geometrically shrinking scales give a Morton-split tree far higher than the 32 stack
    entries staged in LDS; the traversal continues in the global overflow slab."""
    rng = np.random.default_rng(4)
    pts, seg = [], []
    # segment k sits on axis k % 3 at distance 2^-(k // 3 + 1): its Morton key is the only one with bit k set first,
    # so every split peels off a single leaf and the tree degenerates into a ~50-level chain
    for k in range(54):
        a = np.zeros(3)
        a[k % 3] = 0.9 * 2.0 ** (-(k // 3 + 1))
        b = a.copy()
        b[(k + 1) % 3] += 0.3 * 2.0 ** (-(k // 3 + 1))
        seg.append([len(pts), len(pts) + 1])
        pts += [a, b]
    P = np.zeros(len(pts), dtype=lvo.LINE_POINT_DTYPE)
    P["linePosition"] = np.array(pts, dtype=np.float32)
    P["lineTangent"] = [1, 0, 0]
    P["lineNormal"] = [0, 1, 0]
    P["lineAttribute"] = np.linspace(0, 1, len(pts))
    c = Case(P, np.array(seg, np.uint32), tfm.standard_transparent(), 96, 96, 0.0004, camera_pos=(0.3, 0.3, 0.9), **RTAO,
             ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=8)
    ctx = c.hip_context()
    ctx.set_option("accel_build", "fast_build")   # the plain LBVH: the SAH treelets of the default build would balance these 54 leaves
    ctx.build_accel()
    assert ctx.stats().bvh_depth > 32   # binary height: 3 * ceil(h / 2) + 2 stack entries exceed the 32 kept in LDS
    for mode in (11, 2):
        img = ctx.render(mode)
        ref, ao_ref = c.oracle_render(mode)
        assert np.array_equal(bits(ctx.get_ao()), bits(ao_ref))
        assert max_lsb_diff(img, ref) <= LSB_TOL
    tgt = np.array(pts, dtype=np.float64)[rng.integers(0, len(pts), 5000)] * (1.0 + 1e-3 * rng.normal(size=(5000, 3)))
    o = rng.uniform(-0.1, 0.6, (5000, 3)).astype(np.float32)
    d = (tgt - o).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a = ctx.trace_rays(o, d, 0.0, 10.0)
    assert (a[1] != 0xFFFFFFFF).sum() > 300
    b = c.oracle_scene().trace_rays(o, d, 0.0, 10.0, c.line_width)
    assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1])


def test_tile_list_equals_full_frame(hip_lib):
    import torch
    c = small_case(width=150, height=90, seed=3, num_samples_per_frame=2, **RTAO, ambient_occlusion_iterations=1,
                   ambient_occlusion_samples_per_frame=4)
    ctx = c.hip_context()
    full = ctx.render(11)
    part = ctx.render(11, tile=(37, 21, 50, 33))           # ragged rectangle, not aligned to 16
    assert np.array_equal(part, full[21:54, 37:87])
    for mode in (11, 2):
        full = ctx.render(mode)
        tiles = tiling.make_tiles(150, 90, 32)
        out = torch.zeros((len(tiles), 32, 32, 4), dtype=torch.uint8, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.render_tiles_device(out.data_ptr(), tiles, 32, 32, mode=mode)
        torch.cuda.synchronize()
        ctx.set_stream(None)
        frame = tiling.detile(out.cpu().numpy(), tiles, 150, 90, 32)
        assert np.array_equal(frame, full)


def test_tile_adapter_orders_the_consumer_after_the_kernels(hip_lib):
    """tiling.hip_render_tiles_fn: what the caller queues on ITS stream right after the call (the gather, the de-tiling) sees
    the finished tiles without any host synchronisation -- also when the caller runs on torch's default stream (handle 0)."""
    import torch
    c = small_case(width=192, height=128, seed=3, **RTAO, ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=16)
    ctx = c.hip_context()
    full = ctx.render(11)
    tiles = tiling.make_tiles(192, 128, 64)
    fn = tiling.hip_render_tiles_fn(ctx, 11)
    out = torch.zeros((len(tiles), 64, 64, 4), dtype=torch.uint8, device="cuda")
    for side in (None, torch.cuda.Stream()):
        for _ in range(10):
            with torch.cuda.stream(side) if side is not None else torch.cuda.stream(torch.cuda.default_stream()):
                out.zero_()
                fn(out, tiles, 64, 64)
                snap = out.clone()          # queued behind the render by the adapter's stream waits
            torch.cuda.synchronize()
            assert np.array_equal(tiling.detile(snap.cpu().numpy(), tiles, 192, 128, 64), full)
    ctx.set_stream(None)


def test_jittered_colour_rays_sample_the_ao_image_bilinearly(hip_lib):
    """getAoFactor (AmbientOcclusion.glsl:84-99) literally for jittered primaries: project the hit, sample the AO image
    bilinearly.  Tiles then need a 1-pixel AO halo; the rings of adjacent tiles overlap, and with several AO iterations the
    running mean must stay idempotent for the pixels computed twice (ping-pong accumulation in lv_run_ao)."""
    import torch
    c = small_case(width=150, height=90, seed=3, num_samples_per_frame=4, **RTAO, ambient_occlusion_iterations=3,
                   ambient_occlusion_samples_per_frame=4, depth_cue_strength=0.8)
    ctx = c.hip_context()
    full = ctx.render(11)
    ao = ctx.get_ao()
    ref, ao_ref = c.oracle_render(11)
    assert np.array_equal(bits(ao), bits(ao_ref))
    assert max_lsb_diff(full, ref) <= LSB_TOL
    # it is not the own-texel lookup: that one gives a visibly different frame here
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    P.useJitteredRays = 0
    P.numSamplesPerFrame = 1
    assert max_lsb_diff(sc.render_rt(P, ao=ao_ref), ref) > LSB_TOL
    # 32x32 tiles (rings overlap, 3 accumulated iterations) and a ragged rectangle reproduce the frame byte for byte
    tiles = tiling.make_tiles(150, 90, 32)
    out = torch.zeros((len(tiles), 32, 32, 4), dtype=torch.uint8, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.render_tiles_device(out.data_ptr(), tiles, 32, 32, mode=11)
    torch.cuda.synchronize()
    ctx.set_stream(None)
    assert np.array_equal(tiling.detile(out.cpu().numpy(), tiles, 150, 90, 32), full)
    assert np.array_equal(ctx.render(11, tile=(37, 21, 50, 33)), full[21:54, 37:87])
    part, _ = c.oracle_render(11, tile=(37, 21, 50, 33))
    assert max_lsb_diff(part, full[21:54, 37:87]) <= LSB_TOL


def test_progressive_accumulation(hip_lib):
    """num_accumulated_frames > 1: the reference's interactive mode -- frame after frame, each mixed into the previous one
    through the rgba8 image (TubeRayTracing.glsl:268-273), RTAO one iteration per frame while frames < iterations."""
    c = small_case(num_accumulated_frames=5, num_samples_per_frame=2, **RTAO, ambient_occlusion_iterations=3,
                   ambient_occlusion_samples_per_frame=4, rtao_geometry="capsules")   # (the plugin below would tessellate the tubes)
    ref = c.oracle_render_progressive(5)
    ctx = c.hip_context()
    frames = []
    for f in range(5):
        ctx.set_option("frame_number", f)
        frames.append(ctx.render(11))
        assert max_lsb_diff(frames[-1], ref[f]) <= LSB_TOL
    assert not np.array_equal(frames[0], frames[4])
    # the mean converges: late frames change less than early ones
    d01 = np.abs(frames[1].astype(np.int32) - frames[0].astype(np.int32)).mean()
    d34 = np.abs(frames[4].astype(np.int32) - frames[3].astype(np.int32)).mean()
    assert d34 < d01
    # a tile of frame 2 rendered on its own continues from the same accumulation image
    ctx.set_option("frame_number", 0)
    ctx.render(11)
    ctx.set_option("frame_number", 1)
    ctx.render(11)
    ctx.set_option("frame_number", 2)
    t = ctx.render(11, tile=(16, 16, 48, 32))
    assert np.array_equal(t, frames[2][16:48, 16:64])
    # the plugin drives the same loop: render() while needsReRender(), frame number = accumulatedFramesCounter
    tr = scenes.normalize(scenes.random_curves(n_lines=30, points_per_line=30, seed=7))
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    r = host_api.HeadlessLineRenderer(11)
    r.set_rendering_resolution(c.width, c.height)
    r.set_transfer_function(c.tf)
    r.set_line_data(flow)
    r.set_new_settings(dict(line_width=c.line_width, **c.settings))
    view, proj, fovy, near, far = r.camera()
    lo, hi = flow.attribute_range()
    ctx2 = c.hip_context()
    ctx2.set_transfer_function(c.tf, lo, hi)
    ctx2.set_camera(view, proj, fovy, near, far, c.width, c.height)
    for f in range(5):
        ctx2.set_option("frame_number", f)
        assert np.array_equal(r.render_frame(), ctx2.render(11))
    # back to one self-contained frame per call
    ctx.set_option("num_accumulated_frames", 1)
    one = ctx.render(11)
    c1 = small_case(num_samples_per_frame=2, **RTAO, ambient_occlusion_iterations=3, ambient_occlusion_samples_per_frame=4)
    assert max_lsb_diff(one, c1.oracle_render(11)[0]) <= LSB_TOL


def test_option_errors(hip_lib):
    c = small_case(width=32, height=32)
    ctx = c.hip_context()
    for key, val in [("no_such_key", "1"), ("num_accumulated_frames", 0), ("ambient_occlusion_mode", "SSAO"),
                     ("line_width", -1.0), ("geometry_mode", "Curved Swept Spheres"), ("mlat_num_nodes", 3)]:
        with pytest.raises(capi.LineVisError):
            ctx.set_option(key, val)
    with pytest.raises(capi.LineVisError):
        ctx.render(7)
    fresh = capi.Context(0)
    with pytest.raises(capi.LineVisError):
        fresh.render(11)                       # no camera / lines / transfer function yet


def test_instrumented_counters(hip_lib):
    c = small_case(width=64, height=64, **RTAO, ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=8)
    ctx = c.hip_context()
    ctx.set_option("collect_stats", True)
    img = ctx.render(11)
    st = ctx.stats()
    assert st.ao_hit_pixels > 0
    # AO primaries + AO rays + colour-pass traces (one per pixel plus continuations)
    assert st.rays_traced >= 64 * 64 * 2 + st.ao_hit_pixels * 8
    assert st.nodes_visited > st.rays_traced and st.prims_tested > 0 and st.hits_shaded > 0
    ctx.set_option("collect_stats", False)
    assert np.array_equal(ctx.render(11), img)   # instrumentation does not change the image


# ---------------------------------------------------------------- host layer (C++ classes) on the GPU
def test_headless_renderer_plugins(hip_lib):
    tr = scenes.normalize(scenes.random_curves(n_lines=30, points_per_line=30, seed=7))
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    settings = dict(line_width=0.02, depth_cue_strength=0.8, num_samples_per_frame=2, **RTAO,
                    ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=4,
                    rtao_geometry="capsules")   # the plugin's default is the reference's triangle tubes: tests/test_gpu_triangle_tubes.py
    for mode, tf in ((11, tfm.standard()), (2, tfm.standard_transparent())):
        r = host_api.HeadlessLineRenderer(mode)
        assert r.rendering_mode == mode
        r.set_rendering_resolution(96, 64)
        r.set_transfer_function(tf)
        r.set_line_data(flow)
        r.set_new_settings(settings)
        img = r.render_frame()
        # the same frame through the C-ABI directly and through the oracle
        pts, seg, _ = flow.tube_aabb_render_data(0.02)
        lo, hi = flow.attribute_range()
        # (a rasteriser's shaders get USE_CAPPED_TUBES only in the triangle-mesh primitive modes, LineData.cpp:1240-1244: the PPLL
        # plugin in the default "Tube (Programmable Pull)" mode gathers uncapped tubes)
        c = Case(pts, seg, tf, 96, 64, 0.02, use_capped_tubes=(mode == 11),
                 **{k: v for k, v in settings.items() if k != "line_width"})
        ctx = c.hip_context()
        ctx.set_transfer_function(tf, lo, hi)
        view, proj, fovy, near, far = r.camera()   # the C++ Camera class builds its own (float32) matrices
        assert np.allclose(view, c.view, atol=1e-6) and np.allclose(proj, c.proj, atol=1e-6)
        ctx.set_camera(view, proj, fovy, near, far, 96, 64)
        assert np.array_equal(img, ctx.render(mode))
        if mode == 2:
            r.set_new_settings(dict(line_primitive_mode="Tube (Triangle Mesh)"))       # capped tubes for the rasteriser
            ctx.set_option("use_capped_tubes", True)
            capped = r.render_frame()
            assert np.array_equal(capped, ctx.render(mode)) and not np.array_equal(capped, img)
            r.set_new_settings(dict(line_primitive_mode_index=2))                      # back to "Tube (Programmable Pull)"
            assert np.array_equal(r.render_frame(), img)
        # new settings through the plugin surface take effect
        r.set_new_settings(dict(ambient_occlusion_strength=0.0, depth_cue_strength=0.0))
        img2 = r.render_frame()
        assert not np.array_equal(img, img2)


# ---------------------------------------------------------------- full-size properties (BASELINE.json config 3)
def _fast_shading_deviation(name, exact_frame, fast_frame, ref):
    """shading_numerics = fast against the exact oracle frame `ref` (the gate: <= 2 LSB on EVERY pixel) and against the exact HIP frame;
    the histogram goes to gpurun_out/deviations_r06_<name>.json (collected into profiles/deviations_r06.json)."""
    import json
    d_ref = np.abs(fast_frame.astype(np.int32) - ref.astype(np.int32)).max(axis=2)
    d_ex = np.abs(fast_frame.astype(np.int32) - exact_frame.astype(np.int32)).max(axis=2)
    e_ref = np.abs(exact_frame.astype(np.int32) - ref.astype(np.int32)).max(axis=2)
    out = {"workload": name, "pixels": int(d_ref.size),
           "fast_vs_exact_oracle_lsb_histogram": np.bincount(d_ref.ravel(), minlength=4)[:8].tolist(),
           "fast_vs_exact_hip_lsb_histogram": np.bincount(d_ex.ravel(), minlength=4)[:8].tolist(),
           "exact_hip_vs_exact_oracle_lsb_histogram": np.bincount(e_ref.ravel(), minlength=4)[:8].tolist(),
           "max_fast_vs_oracle": int(d_ref.max()), "max_fast_vs_exact_hip": int(d_ex.max())}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "deviations_r06_%s.json" % name), "w"), indent=1)
    assert d_ref.max() <= LSB_TOL, "%s: %d pixels of the fast frame differ from the exact oracle by more than %d LSB (max %d)" % (
        name, int((d_ref > LSB_TOL).sum()), LSB_TOL, int(d_ref.max()))
    return out



def test_full_size_properties(hip_lib):
    tr = scenes.normalize(scenes.tornado())
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    pts, seg, _ = flow.tube_aabb_render_data(0.002)
    assert len(seg) == 1000000
    c = Case(pts, seg, tfm.standard(), 1920, 1080, 0.002, **RTAO, ambient_occlusion_iterations=1,
             ambient_occlusion_samples_per_frame=64)
    ctx = c.hip_context()
    full = ctx.render(11)
    assert ctx.stats().ao_hit_pixels > 100000
    # determinism / idempotence: a second render and a tile render reproduce the frame byte for byte
    assert np.array_equal(ctx.render(11), full)
    assert np.array_equal(ctx.render(11, tile=(701, 333, 257, 129)), full[333:462, 701:958])
    # closest hits of the full-size LBVH against the oracle's own BVH (sampled primaries + random rays) ...
    rng = np.random.default_rng(77)
    o = np.concatenate([np.tile(np.array([[0, 0, 0.8]], np.float32), (20000, 1)),
                        rng.uniform(-0.2, 0.2, (20000, 3)).astype(np.float32)])
    d = rng.normal(size=(40000, 3)).astype(np.float32)
    d[:20000, 2] = -np.abs(d[:20000, 2]) * 3
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a = ctx.trace_rays(o, d, 1e-4, 1000.0)
    sc = c.oracle_scene()
    c.oracle_params(sc)                   # the oracle on the roots this case's settings select (RTAO against the capsules)
    b = sc.trace_rays(o, d, 1e-4, 1000.0, 0.002, use_bvh=True)
    assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    # ... and against brute force over all 1M capsules for a few hundred of them
    idx = rng.choice(40000, 300, replace=False)
    bf = sc.trace_rays(o[idx], d[idx], 1e-4, 1000.0, 0.002, use_bvh=False)
    assert np.array_equal(bits(a[0][idx]), bits(bf[0])) and np.array_equal(a[1][idx], bf[1])
    assert (a[1] != 0xFFFFFFFF).sum() > 5000
    # a 160 x 90 crop of the frame against the oracle (CPU BVH), AO included
    tile = (880, 495, 160, 90)
    ref, _ = c.oracle_render(11, use_bvh=True, tile=tile)
    x0, y0, w, h = tile
    assert max_lsb_diff(full[y0:y0 + h, x0:x0 + w], ref) <= LSB_TOL


def test_config3_whole_frame_against_the_oracle(hip_lib):
    """The headline workload, every pixel: 1 M segments, 1920 x 1080, RTAO 64 spp -- AO factors of all 2 M pixels bit for
    bit, the RGBA8 frame within the bar (the oracle traces the same 27 M rays on the host cores with its own BVH)."""
    tr = scenes.normalize(scenes.tornado())
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    pts, seg, _ = flow.tube_aabb_render_data(0.002)
    c = Case(pts, seg, tfm.standard(), 1920, 1080, 0.002, **RTAO, ambient_occlusion_iterations=1,
             ambient_occlusion_samples_per_frame=64)
    ctx = c.hip_context()
    img = ctx.render(11)
    ao = ctx.get_ao()
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    ao_ref = sc.render_ao(P, use_bvh=True)
    assert np.array_equal(bits(ao), bits(ao_ref))
    assert (ao_ref < 1.0).sum() > 300000
    ref = sc.render_rt(P, ao=ao_ref, use_bvh=True)
    assert max_lsb_diff(img, ref) <= LSB_TOL


def test_config2_whole_frame_against_the_oracle(hip_lib):
    """BASELINE.json config 2: 100 k-segment helix bundle, 1920 x 1080, primary rays only -- every pixel."""
    lvo.shade_normalize_out_of_range(reset=True)
    tr = scenes.normalize(scenes.helix_bundle())
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    pts, seg, _ = flow.tube_aabb_render_data(0.002)
    assert len(seg) == 100000
    c = Case(pts, seg, tfm.standard(), 1920, 1080, 0.002)
    ctx = c.hip_context()
    lo, hi = flow.attribute_range()
    ctx.set_transfer_function(c.tf, lo, hi)
    img = ctx.render(11)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    P.attrMin, P.attrMax = lo, hi
    ref = sc.render_rt(P, use_bvh=True)
    assert max_lsb_diff(img, ref) <= LSB_TOL and (ref[..., :3] != 255).any(axis=2).sum() > 50000
    ctx.set_option("shading_numerics", "fast")          # the priced +-2 LSB contract: every pixel against the EXACT oracle
    _fast_shading_deviation("c2", img, ctx.render(11), ref)
    assert lvo.shade_normalize_out_of_range() == 0      # the clamped normalize() rule never acted on this frame (tests/test_oracle.py)


def test_config2_size_band_data_and_helicity_bands_whole_frames(hip_lib):
    """The config-2 scene (100 k segments, 1920 x 1080) with the round-2 shading variants, every pixel: as band data through the
    sphere-traced elliptic tubes (the bench's c2e workload; + RTAO over the tubelets, AO bit for bit) and with rotating helicity
    bands on the capsules."""
    base = scenes.normalize(scenes.helix_bundle())
    tr = scenes.twisted_ribbons(base, twist=8.0)
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets, tr.ribbon_directions)
    pts, seg, _ = flow.tube_aabb_render_data_elliptic(0.005)
    assert len(seg) == 100000
    c = Case(pts, seg, tfm.standard(), 1920, 1080, 0.002, use_ribbons=True, use_analytic_elliptic_tubes=True, band_width=0.005,
             min_band_thickness=0.15, **RTAO, ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=4)
    ctx = c.hip_context()
    img = ctx.render(11)
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    ao_ref = sc.render_ao(P, use_bvh=True)
    assert np.array_equal(bits(ctx.get_ao()), bits(ao_ref)) and (ao_ref < 1.0).sum() > 100000
    ref = sc.render_rt(P, ao=ao_ref, use_bvh=True)
    assert max_lsb_diff(img, ref) <= LSB_TOL and (ref[..., :3] != 255).any(axis=2).sum() > 100000
    # helicity bands: a smooth signed attribute along the lines drives lineRotation
    hel = (0.03 * np.sin(np.linspace(0.0, 60.0, len(base.positions)))).astype(np.float32)
    pts, seg, _ = lvo.build_tube_aabb_render_data(base.positions, base.attributes, base.line_offsets, 0.002, helicities=hel)
    c = Case(pts, seg, tfm.standard(), 1920, 1080, 0.002, rotating_helicity_bands=True, helicity_rotation_factor=0.1)
    img = c.hip_context().render(11)
    sc = c.oracle_scene()
    ref = sc.render_rt(c.oracle_params(sc), use_bvh=True)
    assert max_lsb_diff(img, ref) <= LSB_TOL
    plain = Case(pts, seg, tfm.standard(), 1920, 1080, 0.002).hip_context().render(11)
    assert (np.abs(plain.astype(np.int32) - img.astype(np.int32)).max(axis=2) > 20).sum() > 5000       # the stripes are there


# ---------------------------------------------------------------- BASELINE.json config 4 at full size
def _ppll_lists_sorted(nodes, start):
    """Every linked fragment of every pixel as three arrays (pixel address, depth bits, colour), sorted by (pixel, depth, colour) --
    the per-pixel multisets in a canonical order -- plus the per-pixel list lengths.  Vectorised list walk: one numpy step per list
    position."""
    nxt = nodes[:, 2]
    pix = np.flatnonzero(start != 0xFFFFFFFF)
    cur = start[pix].astype(np.int64)
    pp, ii = [], []
    while len(pix):
        pp.append(pix)
        ii.append(cur)
        n = nxt[cur]
        keep = n != 0xFFFFFFFF
        pix, cur = pix[keep], n[keep].astype(np.int64)
    pp, ii = np.concatenate(pp), np.concatenate(ii)
    assert len(np.unique(ii)) == len(ii), "a node is linked twice"
    length = np.bincount(pp, minlength=len(start))
    key = np.lexsort((nodes[ii, 0], nodes[ii, 1], pp))
    return pp[key], nodes[ii[key], 1], nodes[ii[key], 0], length


def test_config4_whole_frame_against_the_oracle(hip_lib):
    """BASELINE config 4, every pixel: the fragment lists of the whole 1920 x 1080 frame of the 1 M-segment
    set (rasterised prism, the reference's geometry) against the oracle's gather (LinePassProgrammablePullTubes.glsl:87-224,
    LinkedListGather.glsl:44-71) as per-pixel (depth, colour) multisets bit for bit -- all 2 073 600 pixels, including the > 2 000
    whose lists overflow MAX_NUM_FRAGS = 64 -- and the resolved frame (LinkedListResolve.glsl:57-105, frontToBackPQ
    LinkedListSort.glsl:177-238; keep-the-nearest-64 on both sides) <= 2 LSB on every pixel."""
    lvo.shade_normalize_out_of_range(reset=True)
    tr = scenes.normalize(scenes.tornado())
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    pts, seg, _ = flow.tube_aabb_render_data(0.002)
    W4, H4 = 1920, 1080
    c = Case(pts, seg, tfm.standard_transparent(), W4, H4, 0.002, ppll_max_num_frags=64,
             ppll_expected_avg_depth_complexity=20, collect_stats=True)
    ctx = c.hip_context()
    lo, hi = flow.attribute_range()
    ctx.set_transfer_function(c.tf, lo, hi)
    full = ctx.render(2)
    st = ctx.stats()
    pw, ph = c.padded()
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    assert int(P.ppllFragmentSource) == 1 and int(P.ppllMaxNumFrags) == 64
    P.attrMin, P.attrMax = lo, hi
    nodes, start, cnt = ctx.ppll_buffers(pw * ph, int(P.ppllLinkedListSize))
    whole = (0, 0, W4, H4)
    on, os_, ocnt = sc.ppll_gather(P, tile=whole, use_bvh=True)
    assert cnt == ocnt == st.fragments
    gp, gd, gc_, glen = _ppll_lists_sorted(nodes, start)
    op, od, oc, olen = _ppll_lists_sorted(on, os_)
    assert len(gp) == cnt and len(op) == ocnt
    assert np.array_equal(glen, olen), "list lengths differ on %d pixels" % int((glen != olen).sum())
    assert np.array_equal(gp, op) and np.array_equal(gd, od), "fragment depths differ"
    assert np.array_equal(gc_, oc), "fragment colours differ on %d fragments" % int((gc_ != oc).sum())
    overflow = glen > 64
    assert overflow.sum() >= 1000 and glen.max() == st.max_depth_complexity > 100
    # the resolved frame, every pixel; the overflowing pixels are the ones k_ppll_select_nearest touches
    ref = sc.render_ppll(P, tile=whole, use_bvh=True)
    diff = np.abs(full.astype(np.int32) - ref.astype(np.int32)).max(axis=2)
    assert diff.max() <= LSB_TOL, "%d pixels differ by more than %d LSB (max %d)" % (int((diff > LSB_TOL).sum()), LSB_TOL, int(diff.max()))
    ys, xs = np.mgrid[0:H4, 0:W4]
    tw, th = int(P.ppllTileW), int(P.ppllTileH)   # TiledAddress.glsl:53-85, vectorised
    ntx = pw // tw
    addr = ((ys // th) * ntx + xs // tw) * (tw * th) + (ys % th) * tw + xs % tw
    assert addr[500, 913] == lvo.ppll_addr(913, 500, pw, tw, th) and addr[1079, 1919] == lvo.ppll_addr(1919, 1079, pw, tw, th)
    of_px = overflow[addr]
    assert of_px.sum() == overflow.sum()
    assert diff[of_px].max() <= LSB_TOL
    # the oracle resolving the KERNEL's lists must reproduce the kernel's frame too (no dependence on whose lists)
    assert max_lsb_diff(full, lvo.ppll_resolve(P, nodes, start, tile=whole)) <= 1
    # shading_numerics = fast: the frame within 2 LSB of the EXACT oracle on every pixel; list lengths, fragment depths and alpha bytes
    # bit for bit those of the exact mode (nothing that decides a fragment goes through the approximate operations)
    ctx.set_option("collect_stats", False)
    ctx.set_option("shading_numerics", "fast")
    fast = ctx.render(2)
    fn, fs, fcnt = ctx.ppll_buffers(pw * ph, int(P.ppllLinkedListSize))
    fp, fd, fc, flen = _ppll_lists_sorted(fn, fs)
    assert fcnt == cnt and np.array_equal(flen, glen) and np.array_equal(fp, gp)
    assert np.array_equal(np.sort(fd), np.sort(gd)), "fragment depths changed"
    # per pixel the (depth, alpha) multisets are the same; colours differ by at most 1 LSB per channel
    ka = (gd.astype(np.uint64) << np.uint64(8)) | (gc_ >> 24).astype(np.uint64)
    kb = (fd.astype(np.uint64) << np.uint64(8)) | (fc >> 24).astype(np.uint64)
    oa, ob = np.lexsort((ka, gp)), np.lexsort((kb, fp))
    assert np.array_equal(ka[oa], kb[ob]), "a fragment's depth or alpha changed"
    ca, cb = gc_[oa].view(np.uint8).reshape(-1, 4).astype(np.int16), fc[ob].view(np.uint8).reshape(-1, 4).astype(np.int16)
    same_key = np.concatenate([[False], (ka[oa][1:] == ka[oa][:-1]) & (gp[oa][1:] == gp[oa][:-1])])
    same_key |= np.concatenate([same_key[1:], [False]])     # (fragments of one pixel with equal depth AND alpha may pair up either way)
    assert np.abs(ca - cb)[~same_key].max() <= 1
    _fast_shading_deviation("c4", full, fast, ref)
    assert lvo.shade_normalize_out_of_range() == 0      # the clamped normalize() rule never acted on this frame (tests/test_oracle.py)


def test_config4_full_size_ppll_and_mlat(hip_lib):
    """1 M transparent segments at 1920 x 1080: fragment lists of a crop against the oracle bit for bit, list lengths
    against the global counter, the resolved crop within the RGBA8 bar; the same scene through MLAT with the recorded
    visiting order replayed on the crop."""
    tr = scenes.normalize(scenes.tornado())
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    pts, seg, _ = flow.tube_aabb_render_data(0.002)
    W4, H4 = 1920, 1080
    c = Case(pts, seg, tfm.standard_transparent(), W4, H4, 0.002, ppll_max_num_frags=64,
             ppll_expected_avg_depth_complexity=20, collect_stats=True)
    ctx = c.hip_context()
    lo, hi = flow.attribute_range()
    ctx.set_transfer_function(c.tf, lo, hi)
    full = ctx.render(2)
    st = ctx.stats()
    assert st.fragments > 4000000 and st.max_depth_complexity > 100   # (rasterised prism: 4.46 M; the capsule probe 7.3 M)
    pw, ph = c.padded()
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    P.attrMin, P.attrMax = lo, hi
    nodes, start, cnt = ctx.ppll_buffers(pw * ph, int(P.ppllLinkedListSize))
    assert cnt == st.fragments
    # every fragment is linked exactly once: walk all lists
    nxt = nodes[:, 2]
    length = np.zeros(len(start), np.int64)
    cur = start.astype(np.int64)
    cur[start == 0xFFFFFFFF] = -1
    total = 0
    while (cur >= 0).any():
        act = cur >= 0
        total += int(act.sum())
        length[act] += 1
        n = nxt[cur[act]].astype(np.int64)
        n[n == 0xFFFFFFFF] = -1
        cur[act] = n
    assert total == cnt and length.max() == st.max_depth_complexity
    # a crop through the funnel's core against the oracle (its own BVH): same (colour, depth) multisets per pixel
    tile = (912, 500, 96, 56)
    on, os_, ocnt = sc.ppll_gather(P, tile=tile, use_bvh=True)
    x0, y0, w, h = tile
    checked = 0
    for yy in range(y0, y0 + h, 3):
        for xx in range(x0, x0 + w, 3):
            pix = lvo.ppll_addr(xx, yy, pw, int(P.ppllTileW), int(P.ppllTileH))
            def walk(nd, st_):
                out, i = [], int(st_[pix])
                while i != 0xFFFFFFFF:
                    out.append((int(nd[i, 1]), int(nd[i, 0])))
                    i = int(nd[i, 2])
                return sorted(out)
            a, b = walk(nodes, start), walk(on, os_)
            assert a == b
            checked += len(a)
    assert checked > 5000
    # resolve: where a list is longer than MAX_NUM_FRAGS the reference keeps the first 64 nodes in LIST order, i.e. the
    # result depends on the insertion order (rasterisation order there, wave scheduling here) -- so the oracle resolves
    # the kernel's own lists (identical bytes expected), and its own lists are compared where everything fits
    crop = full[y0:y0 + h, x0:x0 + w]
    assert max_lsb_diff(crop, lvo.ppll_resolve(P, nodes, start, tile=tile)) <= 1
    ref = sc.render_ppll(P, tile=tile, use_bvh=True)
    ys, xs = np.mgrid[y0:y0 + h, x0:x0 + w]
    addr = np.array([lvo.ppll_addr(int(x), int(y), pw, int(P.ppllTileW), int(P.ppllTileH)) for y, x in zip(ys.ravel(), xs.ravel())])
    fits = (length[addr] <= 64).reshape(h, w)
    assert fits.sum() > 500     # (the prism's lists of this crop all fit; the capsule probe's overflowed on > 100 of its pixels)
    assert max_lsb_diff(crop[fits], ref[fits]) <= LSB_TOL
    # the capsule probe (the geometry MLAT traces): same checks on a coarser grid, and the frame MLAT is compared with below
    ctx.set_option("ppll_fragment_source", "capsule_entry")
    c.settings["ppll_fragment_source"] = "capsule_entry"
    full = ctx.render(2)
    P0 = c.oracle_params(sc)
    P0.attrMin, P0.attrMax = lo, hi
    assert ctx.stats().fragments > 5000000
    n0, s0, _ = ctx.ppll_buffers(pw * ph, int(P0.ppllLinkedListSize))
    on0, os0, _ = sc.ppll_gather(P0, tile=tile, use_bvh=True)
    for yy in range(y0, y0 + h, 7):
        for xx in range(x0, x0 + w, 7):
            pix = lvo.ppll_addr(xx, yy, pw, int(P.ppllTileW), int(P.ppllTileH))
            def walk0(nd, st_):
                out, i = [], int(st_[pix])
                while i != 0xFFFFFFFF:
                    out.append((int(nd[i, 1]), int(nd[i, 0])))
                    i = int(nd[i, 2])
                return sorted(out)
            assert walk0(n0, s0) == walk0(on0, os0)
    # MLAT, 8 nodes, the crop replayed in the order the kernel used
    ctx.set_options(dict(use_mlat=True, mlat_num_nodes=8, mlat_record_trace=True, mlat_trace_capacity=1 << 24))
    img = ctx.render(11)
    rec = ctx.mlat_trace()
    assert len(rec) > 2000000
    mref, _, viol = sc.render_rt_mlat(P, 8, tile=tile, use_bvh=True, trace=rec)
    assert viol == 0 and max_lsb_diff(img[y0:y0 + h, x0:x0 + w], mref) <= LSB_TOL
    # and it is a fair approximation of the exact result
    assert np.abs(img.astype(np.int32) - full.astype(np.int32)).mean() < 2.0


# ---------------------------------------------------------------- BASELINE.json config 5 at full scene size
def test_config5_scale_tiles(hip_lib):
    """(c5c: config 5 on the analytic capsules -- not a frame the reference renders; the reference-faithful form is c5t below.)
    5 M segments, 3840 x 2160, RTAO 256 spp: the tile list a rank of the 8-GPU run would own is rendered here for
    two tiles and compared with the oracle (bit-exact AO, frame within 2 LSB); hits of the 5 M-leaf LBVH against the
    oracle's BVH."""
    tr = scenes.normalize(scenes.rayleigh_benard())
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    pts, seg, _ = flow.tube_aabb_render_data(0.002)
    assert len(seg) == 5000000
    W5, H5 = 3840, 2160
    c = Case(pts, seg, tfm.standard(), W5, H5, 0.002, **RTAO, ambient_occlusion_iterations=1,
             ambient_occlusion_samples_per_frame=256)
    ctx = c.hip_context()
    all_tiles = tiling.make_tiles(W5, H5, 64)
    mine = tiling.assign_tiles(all_tiles, 3, 8)                      # rank 3 of 8
    centre = np.argsort(np.abs(mine[:, 0].astype(np.int64) - W5 // 2) + np.abs(mine[:, 1].astype(np.int64) - H5 // 2))[:2]
    tiles = mine[centre]
    import torch
    out = torch.zeros((len(tiles), 64, 64, 4), dtype=torch.uint8, device="cuda:0")
    ctx.render_tiles_device(out.data_ptr(), tiles, 64, 64, mode=11)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    ao = ctx.get_ao()
    st = ctx.stats()
    assert st.num_segments == 5000000 and st.bvh_depth < 96
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    hit_pixels = 0
    for i, (x0, y0) in enumerate(tiles):
        tile = (int(x0), int(y0), 64, 64)
        ao_ref = sc.render_ao(P, tile=tile, use_bvh=True)
        sl = (slice(int(y0), int(y0) + 64), slice(int(x0), int(x0) + 64))
        assert np.array_equal(bits(ao[sl]), bits(ao_ref[sl]))
        hit_pixels += int((ao_ref[sl] < 1.0).sum())
        ref = sc.render_rt(P, ao=ao_ref, tile=tile, use_bvh=True)
        assert max_lsb_diff(got[i], ref) <= LSB_TOL
    assert hit_pixels > 500
    rng = np.random.default_rng(8)
    o = rng.uniform(-0.25, 0.25, (20000, 3)).astype(np.float32)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a = ctx.trace_rays(o, d, 0.0, 0.1)
    b = sc.trace_rays(o, d, 0.0, 0.1, 0.002, use_bvh=True)
    assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert (a[1] != 0xFFFFFFFF).sum() > 2000


def _all_tile_lists_reproduce_the_whole_frame(c, mode, world, cost_scale, mesh=None):
    """Every rank's tile list of a `world`-GPU run rendered in turn on this one GPU through the path a rank uses
    (ShardedFrame.render_local, cost-weighted deal as bench.py applies it) and de-tiled with the device gather's code
    (assemble_device on the pieces) = the frame rendered in one piece, byte for byte."""
    import torch
    ctx = c.hip_context()
    if mesh is not None:
        ctx.set_tube_triangle_mesh(*mesh)      # rtao_geometry = triangle_tubes: the reference's RTAO geometry (the timed headline mode)
    whole = ctx.render(mode)
    dev = torch.device("cuda", 0)
    frames = [tiling.ShardedFrame(c.width, c.height, 64, r, world, dev) for r in range(world)]
    fn = tiling.hip_render_tiles_fn(ctx, mode)
    if cost_scale:
        costs = np.zeros(len(frames[0].all_tiles))
        for sf in frames:
            sf.render_local(fn)
            torch.cuda.synchronize()
            costs[sf.assignment[sf.rank]] = ctx.ao_tile_costs().astype(np.float64) * cost_scale
        for sf in frames:
            sf.redeal(costs, base_cost=4.0 * 64 * 64)
        per_rank = np.array([costs[frames[0].assignment[r]].sum() for r in range(world)])
        assert per_rank.max() < 1.1 * per_rank.mean()                  # the deal balances what it measured
    assert all(np.array_equal(np.sort(np.concatenate(sf.assignment)), np.arange(len(sf.all_tiles))) for sf in frames)
    pieces = []
    for sf in frames:
        sf.render_local(fn)
        torch.cuda.synchronize()
        pieces.append(sf.out.clone())
    frames[0].gathered = pieces                                          # what dist.gather delivers on rank 0
    assert np.array_equal(frames[0].assemble_device().cpu().numpy(), whole)
    return whole


def test_config5_all_eight_tile_lists_reproduce_the_whole_frame(hip_lib):
    """Config 5 as specified (5 M segments, 3840 x 2160, RTAO 256 spp, tiled for 8 ranks): seeds use global pixel coordinates, the
    AO of a pixel never depends on the tile list.  (The RCCL gather itself needs the 8-GPU node; its collective sequence runs
    under gloo in tests/test_tiling_dist.py.)"""
    tr = scenes.normalize(scenes.rayleigh_benard())
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    pts, seg, _ = flow.tube_aabb_render_data(0.002)
    c = Case(pts, seg, tfm.standard(), 3840, 2160, 0.002, **RTAO, ambient_occlusion_iterations=1,
             ambient_occlusion_samples_per_frame=256)
    whole = _all_tile_lists_reproduce_the_whole_frame(c, 11, 8, 256.0)
    assert (whole[..., :3] != 255).any(axis=2).sum() > 1000000


_C5T = {}


def _config5_reference_geometry():
    """BASELINE.json config 5 as the reference renders it: the RTAO pass always traces the triangle-tube TLAS
    (VulkanRayTracedAmbientOcclusion.cpp:439-461), so 5 M segments = 60.3 M triangles (CappedTriangleTubesCPU.cpp:214-383) + the colour
    pass on the capsules with the literal roots.  Built once per session (the two tests below share it)."""
    if not _C5T:
        tr = scenes.normalize(scenes.rayleigh_benard())
        flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
        pts, seg, _ = flow.tube_aabb_render_data(0.002)
        mesh = flow.tube_triangle_render_data(0.002, 6)
        assert len(seg) == 5000000 and len(mesh[0]) == 60300000
        _C5T["case"] = Case(pts, seg, tfm.standard(), 3840, 2160, 0.002, **RTAO, ambient_occlusion_iterations=1,
                            ambient_occlusion_samples_per_frame=256, rtao_geometry="triangle_tubes")
        _C5T["mesh"] = mesh
    return _C5T["case"], _C5T["mesh"]


def test_config5_tiles_with_the_reference_rtao_geometry(hip_lib):
    """c5t: two tiles of rank 3 of 8 against the oracle's triangle scene (its own BVH over the 60.3 M triangles): AO factors bit for
    bit, the frame within 2 LSB; closest hits of random rays against the 30 M-leaf triangle LBVH bit for bit."""
    from oracle import lvo
    import torch
    c, mesh = _config5_reference_geometry()
    assert c.literal_form()                       # intersection_form = auto resolves to the reference's literal roots here
    ctx = c.hip_context()
    ctx.set_tube_triangle_mesh(*mesh)
    all_tiles = tiling.make_tiles(c.width, c.height, 64)
    mine = tiling.assign_tiles(all_tiles, 3, 8)
    centre = np.argsort(np.abs(mine[:, 0].astype(np.int64) - c.width // 2) + np.abs(mine[:, 1].astype(np.int64) - c.height // 2))[:2]
    tiles = mine[centre]
    out = torch.zeros((len(tiles), 64, 64, 4), dtype=torch.uint8, device="cuda:0")
    ctx.render_tiles_device(out.data_ptr(), tiles, 64, 64, mode=11)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    ao = ctx.get_ao()
    st = ctx.stats()
    assert st.num_tube_triangles == 60300000 and st.num_tri_nodes > 5000000 and st.ms_tri_accel_build > 0.0
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    ts = lvo.TriScene(*mesh, 0.002)
    hit_pixels = 0
    for i, (x0, y0) in enumerate(tiles):
        tile = (int(x0), int(y0), 64, 64)
        ao_ref = ts.render_ao(P, tile=tile, use_bvh=True)
        sl = (slice(int(y0), int(y0) + 64), slice(int(x0), int(x0) + 64))
        assert np.array_equal(bits(ao[sl]), bits(ao_ref[sl]))
        hit_pixels += int((ao_ref[sl] < 1.0).sum())
        ref = sc.render_rt(P, ao=ao_ref, tile=tile, use_bvh=True)
        assert max_lsb_diff(got[i], ref) <= LSB_TOL
    assert hit_pixels > 500
    rng = np.random.default_rng(9)
    o = rng.uniform(-0.25, 0.25, (20000, 3)).astype(np.float32)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a = ctx.trace_rays_triangles(o, d, 0.0, 0.1)
    b = ts.trace_rays(o, d, 0.0, 0.1, use_bvh=True)
    assert np.array_equal(a[1], b[1]) and np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(bits(a[2]), bits(b[2]))
    assert (a[1] != 0xFFFFFFFF).sum() > 2000 and int(a[1][a[1] != 0xFFFFFFFF].max()) > 50000000   # hits beyond triangle 2^25


def test_config5_all_eight_tile_lists_with_the_reference_rtao_geometry(hip_lib):
    """c5t sharded for 8 ranks (cost-weighted deal as bench.py applies it): the eight tile lists de-tiled = the whole frame, byte for
    byte, on the 60.3 M-triangle scene."""
    c, mesh = _config5_reference_geometry()
    whole = _all_tile_lists_reproduce_the_whole_frame(c, 11, 8, 256.0, mesh=mesh)
    assert (whole[..., :3] != 255).any(axis=2).sum() > 1000000
    _C5T.clear()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_config3_and_4_tile_lists_reproduce_the_whole_frame(hip_lib, world):
    """Configs 3 (RTAO 64 spp with the reference's triangle tubes for the AO rays + literal roots in the colour pass: the mode
    bench.py times as the headline; with the capsules for 8 ranks too) and 4 (PPLL of the rasterised prism, transparent) at full
    size, sharded for 2 / 4 / 8 ranks."""
    tr = scenes.normalize(scenes.tornado())
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    pts, seg, _ = flow.tube_aabb_render_data(0.002)
    c4 = Case(pts, seg, tfm.standard_transparent(), 1920, 1080, 0.002)
    _all_tile_lists_reproduce_the_whole_frame(c4, 2, world, 0.0)
    mesh = flow.tube_triangle_render_data(0.002, 6)
    c3t = Case(pts, seg, tfm.standard(), 1920, 1080, 0.002, **RTAO, ambient_occlusion_iterations=1,
               ambient_occlusion_samples_per_frame=64, rtao_geometry="triangle_tubes")
    whole = _all_tile_lists_reproduce_the_whole_frame(c3t, 11, world, 64.0, mesh=mesh)
    assert (whole[..., :3] != 255).any(axis=2).sum() > 200000
    if world == 8:
        c3 = Case(pts, seg, tfm.standard(), 1920, 1080, 0.002, **RTAO, ambient_occlusion_iterations=1,
                  ambient_occlusion_samples_per_frame=64)
        assert not np.array_equal(_all_tile_lists_reproduce_the_whole_frame(c3, 11, world, 64.0), whole)


def test_command_line_renderer(hip_lib, tmp_path):
    """python -m linevis_amd: file in, PNG out, through the plugin classes."""
    from linevis_amd.__main__ import main
    from PIL import Image
    tr = scenes.random_curves(n_lines=12, points_per_line=30, seed=2)
    p = str(tmp_path / "in.binlines")
    scenes.write_binlines(p, tr)
    out = str(tmp_path / "out.png")
    assert main([p, "-o", out, "--width", "96", "--height", "64", "line_width=0.02", "depth_cue_strength=0.5"]) == 0
    img = np.array(Image.open(out))
    assert img.shape == (64, 96, 4) and (img[..., :3] != 255).any()
    flow = host_api.LineDataFlow().load_file(p)
    r = host_api.HeadlessLineRenderer(11)
    r.set_rendering_resolution(96, 64); r.set_transfer_function(tfm.standard()); r.set_camera((0.0, 0.0, 0.8)); r.set_line_data(flow)
    r.set_new_settings(dict(line_width=0.02, depth_cue_strength=0.5))
    assert np.array_equal(r.render_frame(), img)
    assert main([p, "-o", out, "--mode", "ppll", "--width", "64", "--height", "48", "line_width=0.03"]) == 0
    assert np.array(Image.open(out)).shape == (48, 64, 4)


@pytest.mark.parametrize("kw", [dict(), dict(use_capped_tubes=False), dict(transparent=True, use_mlat=False),
                                dict(intersection_form="literal", use_capped_tubes=False)])
def test_linear_swept_spheres_geometry_mode(hip_lib, kw):
    """geometry_mode = "Linear Swept Spheres": the hardware primitive (chained end caps) is the exact union of the capsules, so the
    mode is the capsule path with the caps always in the geometry (use_capped_tubes only decides whether the shading sees
    isCap, TubeRayTracing.glsl:621-737) and the exact roots whatever intersection_form says."""
    c = small_case(width=128, height=96, line_width=0.03, geometry_mode="Linear Swept Spheres", **kw)
    img = c.hip_context().render(11)
    ref, _ = c.oracle_render(11)
    assert max_lsb_diff(img, ref) <= LSB_TOL
    plain = dict(c.settings)
    plain["geometry_mode"] = "AABBs (analytic)"
    plain["intersection_form"] = "closest_approach"
    aabb = Case(c.points, c.seg, c.tf, c.width, c.height, c.line_width, **plain).hip_context().render(11)
    if kw.get("use_capped_tubes", True):
        assert np.array_equal(img, aabb)          # same surface, same shading
    else:
        assert not np.array_equal(img, aabb)      # the swept spheres keep their round ends


def test_kernel_timers_option(hip_lib):
    """kernel_timers: which launches are bracketed by HIP events.  The image and the counters do not depend on it; lv_get_kernel_times only
    returns the selected kernels' launches, the ms_* phase fields of lv_get_stats stay 0 without the phase marks."""
    c = small_case(width=96, height=64, transparent=True, **RTAO, ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=4)
    ctx = c.hip_context()
    ctx.set_option("collect_stats", True)
    want = (ctx.render(11), ctx.render(2))
    st_all = ctx.stats()
    names = capi.KERNEL_NAMES
    ctx.reset_timers()
    ctx.render(11)
    assert len(ctx.kernel_times(names.index("k_ao_rays"))) == 1 and len(ctx.kernel_times(names.index("k_render_rt"))) == 1
    assert ctx.stats().ms_total > 0.0
    for value, timed in (("none", []), (str(names.index("k_ao_rays")), ["k_ao_rays"]), ("phases,%d" % names.index("k_render_rt"), ["k_render_rt"])):
        ctx.set_option("kernel_timers", value)
        ctx.reset_timers()
        assert np.array_equal(ctx.render(11), want[0])
        st = ctx.stats()
        for k, name in enumerate(names):
            assert len(ctx.kernel_times(k)) == (1 if name in timed else 0), (value, name)
        assert (st.ms_total > 0.0) == ("phases" in value)
        assert st.rays_traced > 0 and st.nodes_visited > 0
        assert np.array_equal(ctx.render(2), want[1])
    ctx.set_option("kernel_timers", "all")
    ctx.render(2)
    assert ctx.stats().fragments == st_all.fragments and ctx.stats().ms_ppll_resolve > 0.0
    with pytest.raises(capi.LineVisError):
        ctx.set_option("kernel_timers", "k_ao_rays")


@pytest.mark.parametrize("settings,transparent", [
    (dict(RTAO, ambient_occlusion_iterations=2, ambient_occlusion_samples_per_frame=8), False),
    (dict(RTAO, ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=8, depth_cue_strength=0.7), True),
    (dict(RTAO, ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=4, use_jittered_primary_rays=False,
          ambient_occlusion_denoiser="EAW"), False),
    (dict(RTAO, ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=4, geometry_mode="Linear Swept Spheres"), False),
])
def test_overlap_primary_passes_gives_the_same_frames(hip_lib, settings, transparent):
    """overlap_primary_passes (k_primary_pair: the colour pass' first-hit trace in one launch with the RTAO primaries, shading after the
    RTAO pass) against the pass order of VulkanRayTracer::render: frames, tile renders and AO images byte for byte; with transparency
    (the loop goes on tracing behind the pre-traced hit), progressive accumulation (jittered first rays + AO halo) and counters."""
    c = small_case(width=160, height=96, transparent=transparent, **settings)
    frames = {}
    for on in (True, False):
        ctx = c.hip_context()
        ctx.set_option("overlap_primary_passes", on)
        img = ctx.render(11)
        ao = ctx.get_ao()
        tile = ctx.render(11, tile=(37, 21, 70, 50))
        ctx.set_option("collect_stats", True)
        ctx.render(11)
        st = ctx.stats()
        ctx.set_option("collect_stats", False)
        prog = []
        ctx.set_options(dict(num_accumulated_frames=3, ambient_occlusion_iterations=2))
        for f in range(3):
            ctx.set_option("frame_number", f)
            prog.append(ctx.render(11).copy())
        frames[on] = (img, ao, tile, (st.rays_traced, st.hits_shaded, st.ao_hit_pixels), prog)
    a, b = frames[True], frames[False]
    assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(a[2], b[2])
    assert np.array_equal(a[2], a[0][21:71, 37:107])
    assert a[3] == b[3] and a[3][0] > 10000
    for x, y in zip(a[4], b[4]):
        assert np.array_equal(x, y)
    assert max_lsb_diff(a[0], c.oracle_render(11, use_bvh=True)[0]) <= LSB_TOL


@pytest.mark.parametrize("mode,transparent,settings", [
    (11, False, dict(RTAO, ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=8, depth_cue_strength=0.6)),
    (11, True, dict(depth_cue_strength=0.0)),
    (2, True, dict(RTAO, ambient_occlusion_iterations=1, ambient_occlusion_samples_per_frame=4, ppll_fragment_source="raster_prism")),
    (2, True, dict(use_capped_tubes=False, ppll_fragment_source="raster_prism", depth_cue_strength=0.5)),
])
def test_fast_shading_numerics_stay_inside_the_contract(hip_lib, mode, transparent, settings):
    """shading_numerics = fast (approximate hardware rsq / rcp / log2 / exp2 in colour-only arithmetic): the frame stays within 2 LSB of
    the EXACT oracle; AO factors, PPLL list lengths, fragment depths and alpha bytes are bit for bit those of the exact mode; the
    variants the fast mode does not cover (band data, helicity bands, statistics runs) render exactly what they rendered before."""
    c = small_case(width=192, height=128, transparent=transparent, **settings)
    ref, _ = c.oracle_render(mode, use_bvh=True)
    ctx = c.hip_context()
    exact = ctx.render(mode)
    ao = ctx.get_ao() if "ambient_occlusion_mode" in settings else None
    if mode == 2:
        pw, ph = c.padded()
        P = c.oracle_params(c.oracle_scene())
        n0, s0, c0 = ctx.ppll_buffers(pw * ph, int(P.ppllLinkedListSize))
    ctx.set_option("shading_numerics", "fast")
    fast = ctx.render(mode)
    assert max_lsb_diff(exact, ref) <= LSB_TOL and max_lsb_diff(fast, ref) <= LSB_TOL and max_lsb_diff(fast, exact) <= 2
    if ao is not None:
        assert np.array_equal(bits(ctx.get_ao()), bits(ao))
    if mode == 2:
        n1, s1, c1 = ctx.ppll_buffers(pw * ph, int(P.ppllLinkedListSize))
        a, b = _ppll_lists_sorted(n0, s0), _ppll_lists_sorted(n1, s1)
        assert c0 == c1 and np.array_equal(a[3], b[3]) and np.array_equal(a[0], b[0])
        ka = (a[1].astype(np.uint64) << np.uint64(8)) | (a[2] >> 24).astype(np.uint64)
        kb = (b[1].astype(np.uint64) << np.uint64(8)) | (b[2] >> 24).astype(np.uint64)
        assert np.array_equal(ka[np.lexsort((ka, a[0]))], kb[np.lexsort((kb, b[0]))])
    with pytest.raises(capi.LineVisError):
        ctx.set_option("shading_numerics", "sloppy")
    # statistics runs use the exact kernels whatever the option says
    ctx.set_option("collect_stats", True)
    assert np.array_equal(ctx.render(mode), exact)
