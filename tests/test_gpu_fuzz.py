"""Randomised differential test: random scenes, cameras, viewports, tile rectangles and setting combinations through the
C-ABI against the oracle -- AO factors bit for bit, frames within the RGBA8 bar, PPLL fragment multisets identical.  The fixed
cases of test_gpu_parity.py pin what was thought of; this pins combinations nobody thought of (seeded: failures reproduce)."""
import numpy as np
import pytest

from common import Case, max_lsb_diff
from linevis_amd import scenes, tiling, transfer_function as tfm
from oracle import lvo

import os

# exploratory runs: LV_FUZZ_SEED_OFFSET=<n> shifts every seed (the committed cases are offset 0; a failure reproduces with its offset)
SEED_OFFSET = 100000 * int(os.environ.get("LV_FUZZ_SEED_OFFSET", "0"))
RTAO = dict(ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_gamma=1.0, ambient_occlusion_radius=0.1)
LSB_TOL = 2


def random_case(rng):
    n_lines = int(rng.integers(1, 40))
    pts_per_line = int(rng.integers(2, 60))
    tr = scenes.normalize(scenes.random_curves(n_lines=n_lines, points_per_line=pts_per_line, seed=int(rng.integers(1 << 30))))
    lw = float(rng.choice([0.002, 0.004, 0.01, 0.02, 0.05]))
    pts, seg, _ = lvo.build_tube_aabb_render_data(tr.positions, tr.attributes, tr.line_offsets, lw)
    W, H = int(rng.integers(17, 200)), int(rng.integers(9, 130))
    cam = (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.4, 0.4)), float(rng.uniform(0.5, 1.1)))
    bg = tuple(float(x) for x in rng.choice([0.0, 0.25, 1.0], 3)) + (1.0,)
    s = {}
    if rng.uniform() < 0.7:
        s.update(RTAO, ambient_occlusion_strength=float(rng.choice([0.5, 1.0])),
                 ambient_occlusion_iterations=int(rng.integers(1, 4)), ambient_occlusion_samples_per_frame=int(rng.integers(1, 9)),
                 ambient_occlusion_distance_based=bool(rng.integers(2)), use_jittered_primary_rays=bool(rng.integers(2)))
        if rng.uniform() < 0.3:
            s.update(ambient_occlusion_gamma=float(rng.choice([0.5, 2.0])), ambient_occlusion_radius=float(rng.choice([0.03, 0.2])))
        if rng.uniform() < 0.35:
            s.update(ambient_occlusion_denoiser="EAW", eaw_denoiser_iterations=int(rng.integers(0, 4)),
                     eaw_denoiser_use_shared_memory=bool(rng.integers(2)), eaw_denoiser_normal_weights=bool(rng.integers(2)))
    if rng.uniform() < 0.4:
        s["num_samples_per_frame"] = int(rng.integers(2, 5))
    if rng.uniform() < 0.4:
        s["depth_cue_strength"] = float(rng.choice([0.3, 0.8]))
    if rng.uniform() < 0.25:
        s["use_halos"] = False
    if rng.uniform() < 0.25:
        s["use_capped_tubes"] = False
    if rng.uniform() < 0.2:
        s["use_deterministic_sampling"] = True
    if rng.uniform() < 0.2:
        s["intersection_form"] = "literal"
    transparent = rng.uniform() < 0.4
    if transparent and rng.uniform() < 0.5:
        s["max_depth_complexity"] = int(rng.integers(1, 6))
    tf = tfm.standard_transparent() if transparent else tfm.standard()
    return Case(pts, seg, tf, W, H, lw, camera_pos=cam, background=bg, **s), transparent


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_random_ray_tracer_cases(hip_lib, seed):
    import torch
    rng = np.random.default_rng(1000 + seed + SEED_OFFSET)
    for k in range(12):
        c, _ = random_case(rng)
        tag = "seed %d case %d: %dx%d lw %g %s" % (seed, k, c.width, c.height, c.line_width, c.settings)
        ctx = c.hip_context()
        img = ctx.render(11)
        literal = c.literal_form()
        with lvo.deviation_switches(literal_intersection=literal):
            ref, ao_ref = c.oracle_render(11)
        if ao_ref is not None:
            ao = ctx.get_ao()
            if c.eaw_settings():
                assert np.abs(ao - ao_ref).max() < 2e-5, tag
            else:
                assert np.array_equal(ao.view(np.uint32), ao_ref.view(np.uint32)), tag
        assert max_lsb_diff(img, ref) <= LSB_TOL, tag
        # a random rectangle and a random tile size reproduce the frame byte for byte
        x0, y0 = int(rng.integers(0, c.width)), int(rng.integers(0, c.height))
        w, h = int(rng.integers(1, c.width - x0 + 1)), int(rng.integers(1, c.height - y0 + 1))
        assert np.array_equal(ctx.render(11, tile=(x0, y0, w, h)), img[y0:y0 + h, x0:x0 + w]), tag
        t = int(rng.choice([16, 32, 64]))
        tiles = tiling.make_tiles(c.width, c.height, t)
        out = torch.zeros((len(tiles), t, t, 4), dtype=torch.uint8, device="cuda")
        fn = tiling.hip_render_tiles_fn(ctx, 11)
        fn(out, tiles, t, t)
        torch.cuda.synchronize()
        ctx.set_stream(None)
        assert np.array_equal(tiling.detile(out.cpu().numpy(), tiles, c.width, c.height, t), img), tag
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_random_ppll_cases(hip_lib, seed):
    rng = np.random.default_rng(2000 + seed + SEED_OFFSET)
    for k in range(10):
        c, _ = random_case(rng)
        c.tf = np.ascontiguousarray(tfm.standard_transparent(), dtype=np.float32).reshape(-1, 4)
        for key in ("num_samples_per_frame", "intersection_form", "ambient_occlusion_denoiser", "max_depth_complexity"):
            c.settings.pop(key, None)
        c.settings.update(ppll_max_num_frags=int(rng.choice([8, 32, 100, 200])),
                          ppll_tile_width=int(rng.choice([1, 2, 8])), ppll_tile_height=int(rng.choice([1, 8])))
        tag = "seed %d case %d: %dx%d lw %g %s" % (seed, k, c.width, c.height, c.line_width, c.settings)
        ctx = c.hip_context()
        img = ctx.render(2)
        sc = c.oracle_scene()
        P = c.oracle_params(sc)
        ao = sc.render_ao(P) if P.useAmbientOcclusion else None
        on, os_, ocnt = sc.ppll_gather(P, ao=ao)
        pw, ph = c.padded()
        hn, hs, hcnt = ctx.ppll_buffers(pw * ph, int(ctx.stats().ppll_pool_nodes))
        assert hcnt == ocnt, tag

        def lists(nodes, start):
            out = {}
            for pix in np.nonzero(start != 0xFFFFFFFF)[0]:
                l, i = [], int(start[pix])
                while i != 0xFFFFFFFF:
                    l.append((int(nodes[i, 1]), int(nodes[i, 0])))
                    i = int(nodes[i, 2])
                out[int(pix)] = sorted(l)
            return out
        hl, ol = lists(hn, hs), lists(on, os_)
        # depths bit for bit; packed colours within one RGBA8 step per channel (pow() of the device library vs libm can flip
        # the last bit of a channel -- the only inexact function on the path)
        assert hl.keys() == ol.keys(), tag
        for pix in hl:
            a, b = hl[pix], ol[pix]
            assert [d for d, _ in a] == [d for d, _ in b], tag
            for (_, ca), (_, cb) in zip(a, b):
                assert all(abs(((ca >> sh) & 0xFF) - ((cb >> sh) & 0xFF)) <= 1 for sh in (0, 8, 16, 24)), tag
        # the resolve on the HIP lists (identical multisets, possibly another order): frames agree wherever no list is cut
        if max((len(v) for v in hl.values()), default=0) <= int(c.settings["ppll_max_num_frags"]):
            ref = lvo.ppll_resolve(P, hn, hs)
            assert max_lsb_diff(img, ref) <= LSB_TOL, tag
        # a random tile rectangle (a rank's share of a sharded frame: requested-pixel marks, the segment cull pass, the list-driven
        # rasteriser) reproduces its part of the whole frame byte for byte -- unless AO is on: a tile's RTAO pass sees the same pixels
        # (global seeds), so that too
        tw, th = int(rng.integers(1, c.width + 1)), int(rng.integers(1, c.height + 1))
        tx, ty = int(rng.integers(0, c.width - tw + 1)), int(rng.integers(0, c.height - th + 1))
        assert np.array_equal(ctx.render(2, tile=(tx, ty, tw, th)), img[ty:ty + th, tx:tx + tw]), tag + " tile %s" % ((tx, ty, tw, th),)
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(3))
def test_random_triangle_tube_cases(hip_lib, seed):
    """The reference's triangle tubes as RTAO geometry and / or as the ray tracer's "Triangle Mesh" geometry mode."""
    rng = np.random.default_rng(3000 + seed + SEED_OFFSET)
    for k in range(8):
        c, _ = random_case(rng)
        for key in ("intersection_form", "use_capped_tubes"):
            c.settings.pop(key, None)
        # rebuild the trajectories of this case for the tessellator (same generator arguments are not kept: make a new scene)
        n_lines, ppl = int(rng.integers(1, 25)), int(rng.integers(2, 40))
        tr = scenes.normalize(scenes.random_curves(n_lines=n_lines, points_per_line=ppl, seed=int(rng.integers(1 << 30))))
        lw = c.line_width
        pts, seg, _ = lvo.build_tube_aabb_render_data(tr.positions, tr.attributes, tr.line_offsets, lw)
        c = Case(pts, seg, c.tf, c.width, c.height, lw, background=c.background, **c.settings)
        subdiv = int(rng.choice([4, 6, 8]))
        c.settings["tube_num_subdivisions"] = subdiv
        mesh = lvo.build_tube_triangle_render_data(tr.positions, tr.attributes, tr.line_offsets, lw, subdiv)
        tri_ao = bool(rng.integers(2)) and "ambient_occlusion_mode" in c.settings
        tri_colour = bool(rng.integers(2)) or not tri_ao
        tag = "seed %d case %d: %dx%d lw %g triAO %s triColour %s %s" % (seed, k, c.width, c.height, lw, tri_ao, tri_colour, c.settings)
        # the mesh is always there: pin the geometry ("auto" would trace the triangles whenever they are)
        c.settings["rtao_geometry"] = "triangle_tubes" if tri_ao else "capsules"
        ctx = c.hip_context()
        ctx.set_tube_triangle_mesh(*mesh)
        if tri_colour:
            ctx.set_option("geometry_mode", "Triangle Mesh")
        img = ctx.render(11)
        sc = c.oracle_scene()
        P = c.oracle_params(sc)
        tsc = lvo.TriScene(mesh[0], mesh[1], mesh[2], lw)
        ao_ref = None
        if P.useAmbientOcclusion:
            ao_ref = c.oracle_ao(sc, P, render_ao=(lambda t: tsc.render_ao(P, tile=t)) if tri_ao else None)
            ao = ctx.get_ao()
            if c.eaw_settings():
                assert np.abs(ao - ao_ref).max() < 2e-5, tag
            else:
                assert np.array_equal(ao.view(np.uint32), ao_ref.view(np.uint32)), tag
        ref = tsc.render_rt(sc, P, ao=ao_ref) if tri_colour else sc.render_rt(P, ao=ao_ref)
        assert max_lsb_diff(img, ref) <= LSB_TOL, tag
        x0, y0 = int(rng.integers(0, c.width)), int(rng.integers(0, c.height))
        w, h = int(rng.integers(1, c.width - x0 + 1)), int(rng.integers(1, c.height - y0 + 1))
        assert np.array_equal(ctx.render(11, tile=(x0, y0, w, h)), img[y0:y0 + h, x0:x0 + w]), tag
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_random_band_data_cases(hip_lib, seed):
    """Band data: random smooth bundles with random twists, elliptic tubes or circular tubes with USE_BANDS shading, random band
    widths / thicknesses / cameras / RTAO settings."""
    rng = np.random.default_rng(4000 + seed + SEED_OFFSET)
    for k in range(8):
        tr = scenes.twisted_ribbons(scenes.normalize(scenes.helix_bundle(
            n_lines=int(rng.integers(1, 9)), points_per_line=int(rng.integers(8, 120)), seed=int(rng.integers(1 << 30)),
            turns=float(rng.uniform(0.5, 3.0)))), twist=float(rng.uniform(0.0, 25.0)), seed=int(rng.integers(1 << 30)))
        geometry = str(rng.choice(["elliptic", "capsules", "triangles"]))
        elliptic = geometry == "elliptic"
        bw = float(rng.choice([0.01, 0.03, 0.06]))
        lw = float(rng.choice([0.004, 0.01, 0.03]))
        s = dict(use_ribbons=True, use_analytic_elliptic_tubes=elliptic, band_width=bw,
                 min_band_thickness=float(rng.choice([0.05, 0.15, 0.5, 1.0])), thick_bands=bool(rng.integers(4) > 0))
        if rng.uniform() < 0.6:
            s.update(RTAO, ambient_occlusion_strength=1.0, ambient_occlusion_iterations=int(rng.integers(1, 3)),
                     ambient_occlusion_samples_per_frame=int(rng.integers(1, 6)), ambient_occlusion_distance_based=bool(rng.integers(2)),
                     use_jittered_primary_rays=bool(rng.integers(2)), ambient_occlusion_radius=float(rng.choice([0.05, 0.2])))
        if rng.uniform() < 0.4:
            s["num_samples_per_frame"] = int(rng.integers(2, 4))
        if rng.uniform() < 0.4:
            s["depth_cue_strength"] = 0.6
        if rng.uniform() < 0.25:
            s["use_halos"] = False
        if not elliptic and rng.uniform() < 0.3:
            s["use_capped_tubes"] = False
        if elliptic:
            pts, seg, _ = lvo.build_tube_aabb_render_data_ribbons(tr.positions, tr.attributes, tr.line_offsets, bw, tr.ribbon_directions)
        else:
            pts, seg, _ = lvo.build_tube_aabb_render_data(tr.positions, tr.attributes, tr.line_offsets, lw)
        tf = tfm.standard_transparent() if rng.uniform() < 0.4 else tfm.standard()
        cam = (float(rng.uniform(-0.4, 0.4)), float(rng.uniform(-0.3, 0.3)), float(rng.uniform(0.5, 1.0)))
        # the elliptic triangle tubes of the data set: "Triangle Mesh" geometry mode and / or the reference's RTAO geometry
        tri_ao = "ambient_occlusion_mode" in s and bool(rng.integers(2))
        mesh = None
        if geometry == "triangles" or tri_ao:
            nsub = int(rng.choice([8, 10]))
            s["tube_num_subdivisions"] = nsub
            mesh = lvo.build_tube_triangle_render_data_ribbons(tr.positions, tr.attributes, tr.line_offsets, tr.ribbon_directions, bw,
                                                               s["min_band_thickness"], nsub)
        if geometry == "triangles":
            s["geometry_mode"] = "Triangle Mesh"
            s.pop("use_capped_tubes", None)
        s["rtao_geometry"] = "triangle_tubes" if tri_ao else "capsules"   # ("auto" = the triangles whenever the mesh is set)
        c = Case(pts, seg, tf, int(rng.integers(40, 180)), int(rng.integers(30, 120)), lw, camera_pos=cam, **s)
        tag = "seed %d case %d: %dx%d lw %g %s" % (seed, k, c.width, c.height, lw, s)
        ctx = c.hip_context()
        if mesh is not None:
            ctx.set_tube_triangle_mesh(*mesh)
        img = ctx.render(11)
        sc = c.oracle_scene()
        P = c.oracle_params(sc)
        tsc = lvo.TriScene(mesh[0], mesh[1], mesh[2], lw) if mesh is not None else None
        ao_ref = None
        if P.useAmbientOcclusion:
            ao_ref = c.oracle_ao(sc, P, render_ao=(lambda t: tsc.render_ao(P, tile=t)) if tri_ao else None)
            assert np.array_equal(ctx.get_ao().view(np.uint32), ao_ref.view(np.uint32)), tag
        ref = tsc.render_rt(sc, P, ao=ao_ref) if geometry == "triangles" else sc.render_rt(P, ao=ao_ref)
        assert max_lsb_diff(img, ref) <= LSB_TOL, tag
        x0, y0 = int(rng.integers(0, c.width)), int(rng.integers(0, c.height))
        w, h = int(rng.integers(1, c.width - x0 + 1)), int(rng.integers(1, c.height - y0 + 1))
        assert np.array_equal(ctx.render(11, tile=(x0, y0, w, h)), img[y0:y0 + h, x0:x0 + w]), tag
        if geometry != "triangles" and "num_samples_per_frame" not in s:
            # PPLL of the same band data: entry hits of the tubelets / capsules; AO from the pass above (same iterations)
            ppll = ctx.render(2)
            ref2 = sc.render_ppll(P, ao=ao_ref)
            assert max_lsb_diff(ppll, ref2) <= LSB_TOL, tag
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(3))
def test_random_svgf_sequences(hip_lib, seed):
    """SVGF over random camera walks: the temporal state of the HIP context follows the oracle's frame by frame."""
    from linevis_amd import camera
    rng = np.random.default_rng(5000 + seed + SEED_OFFSET)
    for k in range(4):
        c, _ = random_case(rng)
        for key in ("intersection_form", "eaw_denoiser_iterations", "eaw_denoiser_use_shared_memory", "eaw_denoiser_normal_weights"):
            c.settings.pop(key, None)
        its = int(rng.integers(0, 6))
        c.settings.update(RTAO, ambient_occlusion_strength=1.0, ambient_occlusion_iterations=int(rng.integers(1, 3)),
                          ambient_occlusion_samples_per_frame=int(rng.integers(1, 5)), ambient_occlusion_denoiser="SVGF",
                          svgf_denoiser_iterations=its, use_jittered_primary_rays=bool(rng.integers(2)))
        tag = "seed %d case %d: %dx%d lw %g %s" % (seed, k, c.width, c.height, c.line_width, c.settings)
        ctx = c.hip_context()
        sc = c.oracle_scene()
        sv = lvo.Svgf(c.width, c.height, iterations=its)
        pos = np.array([0.0, 0.0, 0.8])
        for f in range(int(rng.integers(2, 6))):
            if rng.uniform() < 0.6:
                pos = pos + rng.normal(scale=0.01, size=3)
            c.view, c.proj, c.fovy, c.near, c.far = camera.default_camera(c.width, c.height, tuple(float(x) for x in pos))
            ctx.set_camera(c.view, c.proj, c.fovy, c.near, c.far, c.width, c.height)
            img = ctx.render(11)
            P = c.oracle_params(sc)
            for _ in range(int(c.settings["ambient_occlusion_iterations"])):
                ao_ref = sv.step(lambda: sc.render_ao(P), P)
            assert np.abs(ctx.get_ao() - ao_ref).max() < 5e-5, tag + " frame %d" % f
            assert max_lsb_diff(img, sc.render_rt(P, ao=ao_ref)) <= LSB_TOL, tag + " frame %d" % f
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_random_helicity_band_cases(hip_lib, seed):
    """Rotating helicity bands: random curves with a random signed helicity attribute, random separator widths / subdivisions /
    rotation factors, capsules (AABB or LSS geometry) or the triangle tubes, the ray tracer and the PPLL."""
    rng = np.random.default_rng(7000 + seed + SEED_OFFSET)
    for k in range(8):
        tr = scenes.normalize(scenes.random_curves(n_lines=int(rng.integers(1, 20)), points_per_line=int(rng.integers(2, 50)),
                                                   seed=int(rng.integers(1 << 30))))
        hel = (rng.normal(size=len(tr.positions)) * float(rng.choice([0.001, 0.02, 1.0]))).astype(np.float32)
        if not hel.any():
            hel[0] = 1.0
        lw = float(rng.choice([0.004, 0.01, 0.03, 0.06]))
        geometry = str(rng.choice(["capsules", "lss", "triangles"]))
        s = dict(rotating_helicity_bands=True, band_subdivisions=int(rng.integers(1, 9)),
                 separator_width=float(rng.choice([0.0, 0.1, 0.2, 0.6])), helicity_rotation_factor=float(rng.choice([0.0, 0.01, 0.3, 1.0, -1.0])))
        if rng.uniform() < 0.5:
            s.update(RTAO, ambient_occlusion_strength=1.0, ambient_occlusion_iterations=int(rng.integers(1, 3)),
                     ambient_occlusion_samples_per_frame=int(rng.integers(1, 6)))
        if rng.uniform() < 0.4:
            s["num_samples_per_frame"] = int(rng.integers(2, 4))
        if rng.uniform() < 0.4:
            s["depth_cue_strength"] = 0.6
        if rng.uniform() < 0.25:
            s["use_halos"] = False
        if rng.uniform() < 0.3:
            s["use_capped_tubes"] = False
        pts, seg, _ = lvo.build_tube_aabb_render_data(tr.positions, tr.attributes, tr.line_offsets, lw, helicities=hel)
        mesh = None
        if geometry == "triangles":
            nsub = int(rng.choice([4, 6, 8]))
            s.update(geometry_mode="Triangle Mesh", tube_num_subdivisions=nsub, use_uniform_twist_line_width=bool(rng.integers(2)))
            mesh = lvo.build_tube_triangle_render_data(tr.positions, tr.attributes, tr.line_offsets, lw, nsub, helicities=hel)
        elif geometry == "lss":
            s["geometry_mode"] = "Linear Swept Spheres"
        s["rtao_geometry"] = "capsules"        # the AO of these cases is the capsules' also where the triangle mesh is set
        tf = tfm.standard_transparent() if rng.uniform() < 0.4 else tfm.standard()
        cam = (float(rng.uniform(-0.4, 0.4)), float(rng.uniform(-0.3, 0.3)), float(rng.uniform(0.5, 1.0)))
        c = Case(pts, seg, tf, int(rng.integers(40, 180)), int(rng.integers(30, 120)), lw, camera_pos=cam, **s)
        tag = "seed %d case %d: %dx%d lw %g %s" % (seed, k, c.width, c.height, lw, s)
        ctx = c.hip_context()
        if mesh is not None:
            ctx.set_tube_triangle_mesh(*mesh)
        img = ctx.render(11)
        sc = c.oracle_scene()
        P = c.oracle_params(sc)
        ao_ref = None
        if P.useAmbientOcclusion:
            ao_ref = c.oracle_ao(sc, P)
            assert np.array_equal(ctx.get_ao().view(np.uint32), ao_ref.view(np.uint32)), tag
        ref = lvo.TriScene(*mesh, lw).render_rt(sc, P, ao=ao_ref) if mesh is not None else sc.render_rt(P, ao=ao_ref)
        assert max_lsb_diff(img, ref) <= LSB_TOL, tag
        if mesh is None and "num_samples_per_frame" not in s:
            assert max_lsb_diff(ctx.render(2), sc.render_ppll(P, ao=ao_ref)) <= LSB_TOL, tag
        ctx.close()
