"""Host C++ layer (linevis_amd/host): LineDataFlow geometry preparation, loaders, settings -- CPU only."""
import os

import numpy as np
import pytest

from common import GOLDEN_DIR
from linevis_amd import host_api, scenes
from oracle import lvo


def test_host_a2_matches_golden_and_oracle():
    g = np.load(os.path.join(GOLDEN_DIR, "a2_cases.npz"))
    flow = host_api.LineDataFlow().set_trajectories(g["positions"], g["attributes"], g["line_offsets"])
    pts, seg, aabb = flow.tube_aabb_render_data(float(g["line_width"]))
    assert np.array_equal(pts.view(np.uint8).reshape(-1, 48), g["points"])
    assert np.array_equal(seg, g["seg"])
    assert np.array_equal(aabb.view(np.uint32), g["aabb_bits"])


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_host_a2_random_scenes_bit_identical(seed):
    tr = scenes.normalize(scenes.random_curves(n_lines=50, points_per_line=37, seed=seed))
    pos = tr.positions.copy()
    rng = np.random.default_rng(seed)
    dup = rng.integers(1, len(pos), 60)
    pos[dup] = pos[dup - 1]  # degenerate points, also across line starts
    flow = host_api.LineDataFlow().set_trajectories(pos, tr.attributes, tr.line_offsets)
    a = flow.tube_aabb_render_data(0.01)
    b = lvo.build_tube_aabb_render_data(pos, tr.attributes, tr.line_offsets, 0.01)
    assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) and a[2].tobytes() == b[2].tobytes()


def test_host_line_width_changes_aabbs_only():
    tr = scenes.normalize(scenes.random_curves(n_lines=6, points_per_line=12, seed=5))
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    p1, s1, a1 = flow.tube_aabb_render_data(0.01)
    p2, s2, a2 = flow.tube_aabb_render_data(0.02)
    assert p1.tobytes() == p2.tobytes() and np.array_equal(s1, s2)
    assert np.allclose(a2[:, 3:] - a1[:, 3:], 0.005, atol=1e-7)


def test_normalize_matches_reference_rule():
    tr = scenes.random_curves(n_lines=7, points_per_line=11, seed=9)
    p = tr.positions * np.float32(7.0) + np.float32(3.0)
    assert np.array_equal(host_api.normalize_positions(p), lvo.normalize_positions(p))


def test_binlines_roundtrip(tmp_path):
    tr = scenes.random_curves(n_lines=9, points_per_line=13, seed=4)
    path = str(tmp_path / "lines.binlines")
    scenes.write_binlines(path, tr)
    # python reader
    back = scenes.read_binlines(path)
    assert np.array_equal(back.positions, tr.positions) and np.array_equal(back.attributes, tr.attributes)
    assert np.array_equal(back.line_offsets, tr.line_offsets)
    # C++ loader: applies the reference's normalisation on load (LineDataFlow::loadFromFile)
    flow = host_api.LineDataFlow().load_binlines(path)
    assert flow.num_lines == 9 and flow.num_points == 9 * 13
    pos, att, off = flow.trajectories()
    assert np.array_equal(pos, scenes.normalize(tr).positions)
    assert np.array_equal(att, tr.attributes) and np.array_equal(off, tr.line_offsets)
    lo, hi = flow.attribute_range()
    assert lo == tr.attributes.min() and hi == tr.attributes.max()
    # C++ writer -> python reader
    path2 = str(tmp_path / "again.binlines")
    flow.save_binlines(path2)
    again = scenes.read_binlines(path2)
    assert np.array_equal(again.positions, pos)


def test_binlines_rejects_bad_version(tmp_path):
    path = str(tmp_path / "bad.binlines")
    open(path, "wb").write(b"\x07\x00\x00\x00" + b"\x00" * 32)
    with pytest.raises(IOError):
        host_api.LineDataFlow().load_binlines(path)
    with pytest.raises(ValueError):
        scenes.read_binlines(path)


def test_scene_generators_shapes():
    assert scenes.lattice().num_segments_upper == 31744
    h = scenes.helix_bundle()
    assert h.num_lines == 100 and h.num_points == 100100
    t = scenes.tornado(n_lines=20, points_per_line=101)
    assert t.positions.shape == (2020, 3) and np.isfinite(t.positions).all()
    assert 0.0 <= t.attributes.min() and t.attributes.max() <= 1.0
    r = scenes.rayleigh_benard(n_lines=10, points_per_line=51)
    assert np.isfinite(r.positions).all()
    # deterministic
    assert np.array_equal(scenes.tornado(n_lines=5, points_per_line=21).positions,
                          scenes.tornado(n_lines=5, points_per_line=21).positions)


def test_obj_polylines_roundtrip(tmp_path):
    """ObjLoader.cpp:36-186: v / vt / l / a statements, 1-based indices, invalid-marker positions dropped; the host
    loader against the Python reader, and against the same set written as .binlines."""
    tr = scenes.random_curves(n_lines=7, points_per_line=12, seed=4)
    p = str(tmp_path / "lines.obj")
    scenes.write_obj(p, tr, "Vorticity")
    back = scenes.read_obj(p)
    assert np.array_equal(back.line_offsets, tr.line_offsets)
    assert np.allclose(back.positions, tr.positions, rtol=1e-7, atol=0) and np.allclose(back.attributes, tr.attributes, rtol=1e-7)
    flow = host_api.LineDataFlow().load_file(p)
    pos, att, off = flow.trajectories()
    norm = scenes.normalize(back)
    assert np.array_equal(off, tr.line_offsets) and np.array_equal(pos, norm.positions) and np.array_equal(att, back.attributes)
    pb = str(tmp_path / "lines.binlines")
    scenes.write_binlines(pb, back)
    pos2, att2, off2 = host_api.LineDataFlow().load_file(pb).trajectories()
    assert np.array_equal(pos2, pos) and np.array_equal(att2, att) and np.array_equal(off2, off)
    # an invalid-marker vertex is skipped, CRLF line ends and tabs are accepted, unknown statements ignored
    txt = "# test\r\nv 0 0 0\r\nvt 0.5\r\nv 1e11 0 0\r\nvt 0.1\r\nv\t1 0 0\r\nvt\t0.25\r\nv 1 1 0\r\nvt 1\r\ns off\r\nl 1 2 3 4\r\n"
    p2 = str(tmp_path / "marker.OBJ")
    open(p2, "w", newline="").write(txt)
    pos3, att3, off3 = host_api.LineDataFlow().load_file(p2).trajectories()
    assert list(off3) == [0, 3] and list(att3) == [0.5, 0.25, 1.0]
    with pytest.raises(Exception):
        host_api.LineDataFlow().load_file(str(tmp_path / "missing.obj"))
    open(str(tmp_path / "lines.xyz"), "w").write("v 0 0 0\n")
    with pytest.raises(Exception):
        host_api.LineDataFlow().load_file(str(tmp_path / "lines.xyz"))


def test_binlines_v2_vertices_normalized_flag(tmp_path):
    """Version-2 files carry verticesNormalized: the reference skips normalisation when it is set (TrajectoryFile.cpp:656 -- its
    streamline tracer exports grid-normalised lines that way) and the writer records the real state of the positions."""
    tr = scenes.twisted_ribbons(scenes.normalize(scenes.random_curves(n_lines=4, points_per_line=12, seed=2)))
    tr.positions[:] = tr.positions * np.float32(0.3) + np.float32(0.05)          # deliberately not in normalised position
    flagged, plain = str(tmp_path / "flagged.binlines"), str(tmp_path / "plain.binlines")
    scenes.write_binlines(flagged, tr, vertices_normalized=True)
    scenes.write_binlines(plain, tr, vertices_normalized=False)
    assert scenes.read_binlines(flagged).vertices_normalized and not scenes.read_binlines(plain).vertices_normalized
    a = host_api.LineDataFlow().load_binlines(flagged)
    pa = a.trajectories()[0]
    assert np.array_equal(pa.view(np.uint32), tr.positions.view(np.uint32))        # flag set: positions taken as they are
    b = host_api.LineDataFlow().load_binlines(plain)
    pb = b.trajectories()[0]
    assert np.array_equal(pb.view(np.uint32), scenes.normalize(tr).positions.view(np.uint32))   # not set: normalised on load
    assert np.array_equal(scenes.load_flow_trajectories(plain).positions, pb) and np.array_equal(scenes.load_flow_trajectories(flagged).positions, pa)
    # what the C++ writer records: loaded data are normalised by then; hand-fed data only if the caller says so
    out = str(tmp_path / "out.binlines")
    b.save_binlines(out)
    assert scenes.read_binlines(out).vertices_normalized
    c = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets, tr.ribbon_directions)
    c.save_binlines(out)
    assert not scenes.read_binlines(out).vertices_normalized
    c.set_vertices_normalized(True).save_binlines(out)
    assert scenes.read_binlines(out).vertices_normalized
    # a truncated attribute-name length must fail cleanly, not allocate 4 GB
    data = bytearray(open(flagged, "rb").read())
    ntr = 12 + sum(4 + 12 * 12 + 4 * 12 for _ in range(4))
    bad = bytes(data[:ntr]) + (1).to_bytes(4, "little") + (1).to_bytes(4, "little") + (0xFFFFFFF0).to_bytes(4, "little")
    open(str(tmp_path / "trunc.binlines"), "wb").write(bad)
    with pytest.raises(Exception):
        host_api.LineDataFlow().load_binlines(str(tmp_path / "trunc.binlines"))
