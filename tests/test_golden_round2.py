"""Committed fixtures of the round-2 features (tests/golden/round2.npz, made by tests/golden/make_golden.py from the oracle): EAW- and
SVGF-denoised AO images, elliptic tubes / USE_BANDS frames, streamribbon directions.  CPU: the oracle still reproduces them
(regression pin); GPU: the HIP path against the committed data -- no oracle in the loop."""
import importlib.util
import os

import numpy as np
import pytest

from common import GOLDEN_DIR, max_lsb_diff
from linevis_amd import camera
from oracle import lvo

_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN_DIR, "make_golden.py"))
mg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mg)


def golden():
    return np.load(os.path.join(GOLDEN_DIR, "round2.npz"))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_oracle_reproduces_the_round2_fixtures():
    g = golden()
    c = mg.round2_case("eaw")
    img, ao = c.oracle_render(11)
    assert np.array_equal(img, g["eaw_frame"]) and np.array_equal(bits(ao), g["eaw_ao_bits"])
    c = mg.round2_case("svgf")
    sc = c.oracle_scene()
    sv = lvo.Svgf(c.width, c.height, iterations=3)
    for f, pos in enumerate(mg.SVGF_PATH):
        c.view, c.proj, c.fovy, c.near, c.far = camera.default_camera(c.width, c.height, pos)
        P = c.oracle_params(sc)
        ao = sv.step(lambda: sc.render_ao(P), P)
        assert np.array_equal(bits(ao), g["svgf_ao_bits_%d" % f]) and np.array_equal(bits(sv.raw), g["svgf_raw_bits_%d" % f])
    for kind in ("elliptic", "bands"):
        img, ao = mg.round2_case(kind).oracle_render(11)
        assert np.array_equal(img, g[kind + "_frame"])
        if ao is not None:
            assert np.array_equal(bits(ao), g[kind + "_ao_bits"])


@pytest.mark.gpu
def test_hip_against_the_round2_fixtures(hip_lib):
    from linevis_amd import host_api
    g = golden()
    # EAW: denoised AO within the exp() tolerance of the committed image, frame within the RGBA8 bar
    c = mg.round2_case("eaw")
    ctx = c.hip_context()
    img = ctx.render(11)
    assert np.abs(ctx.get_ao() - g["eaw_ao_bits"].view(np.float32)).max() < 2e-5 and max_lsb_diff(img, g["eaw_frame"]) <= 2
    # SVGF: the raw AO of every frame bit for bit (global frame counter, no accumulation), the denoised one after every frame
    c = mg.round2_case("svgf")
    ctx = c.hip_context()
    for f, pos in enumerate(mg.SVGF_PATH):
        c.view, c.proj, c.fovy, c.near, c.far = camera.default_camera(c.width, c.height, pos)
        ctx.set_camera(c.view, c.proj, c.fovy, c.near, c.far, c.width, c.height)
        img = ctx.render(11)
        assert np.abs(ctx.get_ao() - g["svgf_ao_bits_%d" % f].view(np.float32)).max() < 3e-5, f
        assert max_lsb_diff(img, g["svgf_frame_%d" % f]) <= 2, f
    # band data
    for kind in ("elliptic", "bands"):
        c = mg.round2_case(kind)
        ctx = c.hip_context()
        assert max_lsb_diff(ctx.render(11), g[kind + "_frame"]) <= 2
        if kind + "_ao_bits" in g:
            assert np.array_equal(bits(ctx.get_ao()), g[kind + "_ao_bits"])
    c = mg.round2_case("elliptic")
    ctx = c.hip_context()
    d = g["elliptic_rays_d"]
    o = np.tile(np.array([[0.0, 0.0, 0.8]], np.float32), (len(d), 1))
    t, s, _ = ctx.trace_rays(o, d, 1e-4, 1000.0)
    assert np.array_equal(bits(t), g["elliptic_rays_t_bits"]) and np.array_equal(s, g["elliptic_rays_seg"])
    assert (s != 0xFFFFFFFF).sum() > 100
    # streamribbons through the host layer
    n = 16
    v = lvo.generate_abc_flow(n, n, n)          # input generator only; the expected lines / directions are the committed ones
    grid = host_api.StreamlineTracingGrid().load_abc_flow(n, n, n, 6.0)
    pos, att, off, rib = grid.trace_streamribbons(g["ribbons_seeds"], minimum_length=0.2)
    assert np.array_equal(bits(pos), g["ribbons_pos_bits"]) and np.array_equal(off, g["ribbons_off"])
    assert np.array_equal(bits(rib), g["ribbons_dir_bits"])


def golden_b():
    return np.load(os.path.join(GOLDEN_DIR, "round2b.npz"))


def _fragments(nodes, start):
    out = []
    for pix in np.nonzero(start != 0xFFFFFFFF)[0]:
        i = int(start[pix])
        while i != 0xFFFFFFFF:
            out.append((int(pix), int(nodes[i, 1]), int(nodes[i, 0])))
            i = int(nodes[i, 2])
    return np.array(sorted(out), dtype=np.uint32)


def test_oracle_reproduces_the_second_round2_fixture_set():
    """PPLL fragment lists / MLAT of band data, rotating helicity bands on capsules and triangle tubes (round2b.npz)."""
    g = golden_b()
    c = mg.round2b_case("ppll_elliptic")
    sc = c.oracle_scene()
    P = c.oracle_params(sc)
    assert np.array_equal(sc.render_ppll(P), g["ppll_elliptic_frame"])
    n, s, _ = sc.ppll_gather(P)
    assert np.array_equal(_fragments(n, s), g["ppll_elliptic_fragments"]) and len(g["ppll_elliptic_fragments"]) > 500
    c = mg.round2b_case("mlat_elliptic")
    sc = c.oracle_scene()
    assert np.array_equal(sc.render_rt_mlat(c.oracle_params(sc), 4)[0], g["mlat_elliptic_frame"])
    c = mg.round2b_case("helicity")
    assert np.array_equal(c.oracle_render(11)[0], g["helicity_frame"])
    assert np.array_equal(bits(c.points["lineRotation"]), g["helicity_rotation_bits"])
    c, mesh = mg.round2b_case("helicity_tri")
    sc = c.oracle_scene()
    assert np.array_equal(lvo.TriScene(*mesh, c.line_width).render_rt(sc, c.oracle_params(sc)), g["helicity_tri_frame"])
    assert np.array_equal(bits(mesh[2]["lineRotation"]), g["helicity_tri_rotation_bits"])


@pytest.mark.gpu
def test_hip_against_the_second_round2_fixture_set(hip_lib):
    from linevis_amd import host_api, scenes
    g = golden_b()
    c = mg.round2b_case("ppll_elliptic")
    ctx = c.hip_context()
    assert max_lsb_diff(ctx.render(2), g["ppll_elliptic_frame"]) <= 2
    pw, ph = c.padded()
    P = c.oracle_params(c.oracle_scene())
    hn, hs, _ = ctx.ppll_buffers(pw * ph, int(P.ppllLinkedListSize))
    assert np.array_equal(_fragments(hn, hs), g["ppll_elliptic_fragments"])         # fragment multisets, bit for bit
    # MLAT in the kernel's own order: few layers per pixel here, close to the canonical-order fixture
    c = mg.round2b_case("mlat_elliptic")
    d = np.abs(c.hip_context().render(11).astype(np.int32) - g["mlat_elliptic_frame"].astype(np.int32))
    assert d.mean() < 0.5
    c = mg.round2b_case("helicity")
    assert max_lsb_diff(c.hip_context().render(11), g["helicity_frame"]) <= 2
    c, mesh = mg.round2b_case("helicity_tri")
    ctx = c.hip_context()
    ctx.set_tube_triangle_mesh(*mesh)
    assert max_lsb_diff(ctx.render(11), g["helicity_tri_frame"]) <= 2
    # the host layer's lineRotation against the committed one (no oracle in the loop)
    tr = scenes.normalize(scenes.helix_bundle(n_lines=4, points_per_line=50, seed=4, turns=2.0))
    s_ = np.linspace(0.0, 1.0, len(tr.positions)).astype(np.float32)
    hel = (0.02 * np.sin(12.0 * s_ + 1.0) + 0.01).astype(np.float32)
    flow = host_api.LineDataFlow().set_trajectories_multi(tr.positions, np.stack([tr.attributes, hel]), ["Attribute", "Helicity"],
                                                         tr.line_offsets)
    flow.set_new_settings(dict(rotating_helicity_bands=True))
    assert np.array_equal(bits(flow.tube_aabb_render_data(0.03)[0]["lineRotation"]), g["helicity_rotation_bits"])
    assert np.array_equal(bits(flow.tube_triangle_render_data(0.03, 8)[2]["lineRotation"]), g["helicity_tri_rotation_bits"])
    flow.set_new_settings(dict(rotating_helicity_bands=False))
