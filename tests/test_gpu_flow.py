"""Streamline tracing on the GPU (StreamlineTracingGrid::traceStreamlines) through the C-ABI against
the CPU oracle: positions, attributes and line offsets bit for bit (float32 +,-,*,/,sqrt,floor in a fixed order)."""
import numpy as np
import pytest

from common import small_case
from linevis_amd import capi, host_api
from oracle import lvo

pytestmark = pytest.mark.gpu


def abc_grid(n=32):
    v = lvo.generate_abc_flow(n, n, n)
    mag = np.sqrt((v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]) + v[..., 2] * v[..., 2]).astype(np.float32)
    d = np.float32(1.0) / np.float32(n - 1)
    return v, mag, (float(d), float(d), float(d))


def swirl_grid(xs=48, ys=40, zs=56):
    """A non-cubic grid with anisotropic spacing and a bounded swirl (lines stay inside for many steps)."""
    z, y, x = np.meshgrid(np.linspace(0, 1, zs), np.linspace(0, 1, ys), np.linspace(0, 1, xs), indexing="ij")
    u = -(y - 0.5) + 0.2 * (0.5 - x)
    v = (x - 0.5) + 0.2 * (0.5 - y)
    w = 0.3 * np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y) + 0.05
    vec = np.stack([u, v, w], axis=3).astype(np.float32)
    s0 = np.sqrt(u * u + v * v + w * w).astype(np.float32)
    s1 = (x + 2 * y + 3 * z).astype(np.float32)
    return vec, [s0, s1], (0.02, 0.025, 0.0175)


def same(a, b):
    return (np.array_equal(a[2], b[2]) and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
            and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)))


@pytest.mark.parametrize("method", ["Explicit Euler", "Heun", "Midpoint", "Runge-Kutta 4th Order"])
@pytest.mark.parametrize("direction", ["Forward", "Backward", "Forward & Backward"])
def test_streamlines_bit_exact_abc_flow(hip_lib, method, direction):
    v, mag, sp = abc_grid()
    rng = np.random.default_rng(1)
    seeds = rng.uniform(0.05, 0.95, (300, 3)).astype(np.float32)
    seeds[:5] = [[0, 0, 0], [1, 1, 1], [1.5, 0.5, 0.5], [-0.1, 0.2, 0.3], [0.5, 0.5, 1.0]]   # corners / outside
    ctx = capi.Context(0)
    ctx.set_flow_grid(v, sp, [mag])
    a = ctx.trace_streamlines(seeds, capi.streamline_settings(method, direction, minimum_length=0.3))
    b = lvo.trace_streamlines(v, sp, [mag], seeds, lvo.streamline_settings(method, direction, minimum_length=0.3))
    assert same(a, b)
    assert len(a[2]) - 1 > 100 and len(a[0]) > 3000


@pytest.mark.parametrize("direction", ["Forward", "Forward & Backward"])
def test_implicit_euler_bit_exact(hip_lib, direction):
    """Fixed-point iteration in float32 (+, *, sqrt): bit-identical like the explicit integrators."""
    v, mag, sp = abc_grid()
    rng = np.random.default_rng(6)
    seeds = rng.uniform(0.05, 0.95, (300, 3)).astype(np.float32)
    ctx = capi.Context(0)
    ctx.set_flow_grid(v, sp, [mag])
    for scale in (1.0, 4.0):
        S = dict(time_step_scale=scale, minimum_length=0.2)
        a = ctx.trace_streamlines(seeds, capi.streamline_settings("Implicit Euler", direction, **S))
        b = lvo.trace_streamlines(v, sp, [mag], seeds, lvo.streamline_settings("Implicit Euler", direction, **S))
        assert same(a, b) and len(a[0]) > 1500


def test_rkf45_matches_the_oracle(hip_lib):
    """Runge-Kutta-Fehlberg runs in float64 with one pow() per step adaptation: the device's pow and libm's agree to the
    last bit almost always, so almost every line is bit-identical; a line whose step once differed in the last place
    stays within float32 rounding of the oracle's.  Line structure (offsets) must be identical."""
    v, mag, sp = abc_grid()
    rng = np.random.default_rng(9)
    seeds = rng.uniform(0.05, 0.95, (400, 3)).astype(np.float32)
    ctx = capi.Context(0)
    ctx.set_flow_grid(v, sp, [mag])
    for scale, direction in ((1.0, "Forward"), (6.0, "Forward & Backward")):
        S = dict(time_step_scale=scale, minimum_length=0.2)
        a = ctx.trace_streamlines(seeds, capi.streamline_settings("Runge-Kutta-Fehlberg", direction, **S))
        b = lvo.trace_streamlines(v, sp, [mag], seeds, lvo.streamline_settings("Runge-Kutta-Fehlberg", direction, **S))
        assert np.array_equal(a[2], b[2]) and len(a[0]) > 3000
        assert np.allclose(a[0], b[0], rtol=0, atol=2e-6) and np.allclose(a[1], b[1], rtol=0, atol=2e-5)
        off = a[2].astype(np.int64)
        identical = sum(np.array_equal(a[0][off[l]:off[l + 1]].view(np.uint32), b[0][off[l]:off[l + 1]].view(np.uint32))
                        for l in range(len(off) - 1))
        assert identical >= 0.9 * (len(off) - 1)
    # the step does adapt at scale 6: more points per line than RK4 with the same nominal step
    r = ctx.trace_streamlines(seeds, capi.streamline_settings("Runge-Kutta 4th Order", "Forward & Backward", **S))
    assert len(a[0]) > 1.1 * len(r[0])


def test_streamlines_long_lines_termination_rules(hip_lib):
    vec, scalars, sp = swirl_grid()
    rng = np.random.default_rng(3)
    seeds = rng.uniform(0.1, 0.7, (500, 3)).astype(np.float32) * np.array([47 * 0.02, 39 * 0.025, 55 * 0.0175], np.float32)
    ctx = capi.Context(0)
    ctx.set_flow_grid(vec, sp, scalars)
    lens = []
    for kw in (dict(max_num_iterations=500, time_step_scale=0.25, termination_distance=1500.0, minimum_length=0.0),
               dict(max_num_iterations=300, time_step_scale=2.0, minimum_length=0.1),   # short line-length limit, bigger step
               dict(max_num_iterations=2000, minimum_length=0.0)):                      # line-length limit
        a = ctx.trace_streamlines(seeds, capi.streamline_settings(**kw))
        b = lvo.trace_streamlines(vec, sp, scalars, seeds, lvo.streamline_settings(**kw))
        assert same(a, b)
        lens.append(len(a[0]))
    assert lens[0] > 500 and lens[1] > 2000 and lens[2] > 20000
    # "singular point" rule (segment shorter than 1e-6 * termination_distance): slow regions end lines early
    assert lens[0] < 500 * 2000
    # second attribute is linear in the position: trilinear sampling reproduces it along every line
    pos, att, off = a
    lin = pos[:, 0] / sp[0] / 47 + 2 * pos[:, 1] / sp[1] / 39 + 3 * pos[:, 2] / sp[2] / 55
    inside = (pos.min(axis=1) > 0)
    assert np.allclose(att[1][inside], lin[inside], atol=2e-4)


def test_streamlines_feed_the_renderer(hip_lib):
    """traced lines -> LineDataFlow -> frame: the widened path end to end."""
    v, mag, sp = abc_grid(24)
    rng = np.random.default_rng(2)
    seeds = rng.uniform(0.2, 0.8, (64, 3)).astype(np.float32)
    ctx = capi.Context(0)
    ctx.set_flow_grid(v, sp, [mag])
    pos, att, off = ctx.trace_streamlines(seeds, capi.streamline_settings(minimum_length=0.2))
    assert len(off) - 1 > 20
    flow = host_api.LineDataFlow().set_trajectories(host_api.normalize_positions(pos), att[0], off)
    pts, seg, _ = flow.tube_aabb_render_data(0.01)
    case = small_case(width=64, height=48, line_width=0.01)
    ctx.set_lines(pts, seg)
    ctx.set_transfer_function(case.tf, *flow.attribute_range())
    ctx.set_camera(case.view, case.proj, case.fovy, case.near, case.far, 64, 48)
    ctx.set_option("line_width", 0.01)
    img = ctx.render(capi.MODE_RAY_TRACER)
    assert (img[..., :3] != 255).any()


def test_streamline_api_errors(hip_lib):
    ctx = capi.Context(0)
    with pytest.raises(capi.LineVisError):
        ctx.trace_streamlines(np.zeros((1, 3), np.float32), capi.streamline_settings())       # no grid
    v, mag, sp = abc_grid(8)
    ctx.set_flow_grid(v, sp, [mag])
    bad = capi.streamline_settings()
    bad.integration_method = 6
    with pytest.raises(capi.LineVisError):
        ctx.trace_streamlines(np.zeros((1, 3), np.float32) + 0.5, bad)
    with pytest.raises(capi.LineVisError):
        ctx.trace_streamlines(np.zeros((1, 3), np.float32), capi.streamline_settings(time_step_scale=0.0))
    with pytest.raises(capi.LineVisError):
        ctx.set_flow_grid(np.zeros((1, 4, 4, 3), np.float32), sp)
    pos, att, off = ctx.trace_streamlines(np.zeros((0, 3), np.float32), capi.streamline_settings())
    assert len(pos) == 0 and list(off) == [0]
    ctx.set_flow_grid(np.zeros((4, 4, 4, 3), np.float32), sp)
    with pytest.raises(capi.LineVisError):
        ctx.trace_streamlines(np.zeros((1, 3), np.float32) + 0.5, capi.streamline_settings())  # zero field: dt undefined


def test_streamlines_golden_fixture(hip_lib):
    import os
    from common import GOLDEN_DIR
    g = np.load(os.path.join(GOLDEN_DIR, "flow_small.npz"))
    n, d = int(g["n"]), float(g["spacing"])
    v = lvo.generate_abc_flow(n, n, n)
    mag = np.sqrt((v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]) + v[..., 2] * v[..., 2]).astype(np.float32)
    ctx = capi.Context(0)
    ctx.set_flow_grid(v, (d, d, d), [mag])
    for key, method, direction in (("rk4_both", "Runge-Kutta 4th Order", "Forward & Backward"),
                                   ("euler_fwd", "Explicit Euler", "Forward"), ("heun_bwd", "Heun", "Backward"),
                                   ("midpoint_both", "Midpoint", "Forward & Backward"),
                                   ("implicit_fwd", "Implicit Euler", "Forward")):
        pos, att, off = ctx.trace_streamlines(g["seeds"], capi.streamline_settings(method, direction, minimum_length=0.25))
        assert np.array_equal(off, g[key + "_off"])
        assert np.array_equal(pos.view(np.uint32), g[key + "_pos_bits"]) and np.array_equal(att.view(np.uint32), g[key + "_att_bits"])


def test_host_tracer_classes(hip_lib):
    """lv::StreamlineTracingGrid + AbcFlowGenerator + StreamlineVolumeSeeder (the plugin-side front end) against the
    oracle: same field, same seeds, same lines."""
    grid = host_api.StreamlineTracingGrid().load_abc_flow(24, 20, 28, 6.0)
    sizes, spacing, box = grid.info()
    assert list(sizes) == [24, 20, 28] and np.allclose(spacing, 1.0 / 27) and np.allclose(box[3:], np.array([23, 19, 27]) / 27.0)
    seeds = grid.regular_seeds(5, 4, 3)
    # StreamlineVolumeSeeder::getNextPoint (regular): boxMin + dimensions * (i + 1) / (n + 1), x fastest
    exp = np.array([[box[3] * (x + 1) / 6, box[4] * (y + 1) / 5, box[5] * (z + 1) / 4]
                    for z in range(3) for y in range(4) for x in range(5)], np.float32)
    assert np.allclose(seeds, exp, atol=1e-6)
    a = grid.trace_streamlines(seeds, minimum_length=0.3)
    # the same through the oracle (AbcFlowGenerator restated; the host computes |v| as sqrt((xx + yy) + zz))
    v = lvo.generate_abc_flow(24, 20, 28)
    sp = tuple(float(s) for s in spacing)
    # AbcFlowGenerator::load adds (name order) Helicity, Velocity Magnitude, Vorticity Magnitude (GridLoader.cpp:41-183)
    w = lvo.vorticity_field(v, sp)
    fields = [lvo.helicity_field(v, w), lvo.vector_magnitude_field(v), lvo.vector_magnitude_field(w)]
    b = lvo.trace_streamlines(v, sp, fields, seeds, lvo.streamline_settings(minimum_length=0.3))
    assert same(a, b) and len(a[2]) > 20 and a[1].shape[0] == 3
    # traceStreamribbons: the same lines + ribbon directions carried outwards from the seed and twisted by the helicity
    for direction in ("Forward", "Backward", "Forward & Backward"):
        for kw in (dict(), dict(use_helicity=False), dict(max_helicity_twist=1.0, initial_ribbon_direction=(1.0, 0.0, 0.0))):
            ra = grid.trace_streamribbons(seeds, direction=direction, minimum_length=0.3, **kw)
            rb = lvo.trace_streamribbons(v, sp, fields, seeds, lvo.streamline_settings(direction=direction, minimum_length=0.3), 0, **kw)
            assert same(ra[:3], rb[:3]) and np.array_equal(ra[3].view(np.uint32), rb[3].view(np.uint32)), (direction, kw)
            assert len(ra[0]) > 200
    # the max-helicity-first seeder through the host class (it reads the "Helicity" scalar field AbcFlowGenerator::load added)
    ha = grid.trace_streamlines_max_helicity_first(minimum_length=0.3, minimum_separation_distance=0.1)
    hb = lvo.trace_streamlines_max_helicity_first(v, sp, fields, fields[0], lvo.streamline_settings(minimum_length=0.3),
                                                  minimum_separation_distance=0.1)
    assert same(ha, hb) and len(ha[2]) > 10
    # ... with StreamlineTracingSettings::terminationCheckType = HASHED_GRID_BASED (3): the point-distance rule instead of the occupancy grid
    hc = grid.trace_streamlines_max_helicity_first(minimum_length=0.3, minimum_separation_distance=0.1, termination_check_type=3,
                                                   max_num_iterations=400)
    hd = lvo.trace_streamlines_max_helicity_first(v, sp, fields, fields[0], lvo.streamline_settings(minimum_length=0.3, max_num_iterations=400),
                                                  minimum_separation_distance=0.1, termination_check_type=3)
    assert same(hc, hd) and len(hc[2]) > 10
    for direction in ("Forward", "Backward", "Forward & Backward"):       # ... and its streamribbon form
        ra = grid.trace_streamlines_max_helicity_first(direction=direction, minimum_length=0.3, minimum_separation_distance=0.1, ribbons=True,
                                                       max_helicity_twist=0.5)
        rb = lvo.trace_streamlines_max_helicity_first(v, sp, fields, fields[0], lvo.streamline_settings(direction=direction, minimum_length=0.3),
                                                      minimum_separation_distance=0.1, ribbons=dict(max_helicity_twist=0.5))
        assert same(ra[:3], rb[:3]) and np.array_equal(ra[3].view(np.uint32), rb[3].view(np.uint32)), direction
        assert np.abs(np.linalg.norm(ra[3], axis=1) - 1).max() < 1e-3
    # ... the streamribbon form with the k-d tree termination check
    ra = grid.trace_streamlines_max_helicity_first(minimum_length=0.3, minimum_separation_distance=0.1, ribbons=True, termination_check_type=2,
                                                   max_num_iterations=400)
    rb = lvo.trace_streamlines_max_helicity_first(v, sp, fields, fields[0], lvo.streamline_settings(minimum_length=0.3, max_num_iterations=400),
                                                  minimum_separation_distance=0.1, ribbons=dict(), termination_check_type=2)
    assert same(ra[:3], rb[:3]) and np.array_equal(ra[3].view(np.uint32), rb[3].view(np.uint32)) and len(ra[2]) > 10
    # a second vector field / scalar field set by hand, another integrator
    vec, scalars, sp2 = swirl_grid(20, 24, 16)
    grid.set_grid_extent(20, 24, 16, *sp2).add_vector_field(vec).add_scalar_field(scalars[1], "b").add_scalar_field(scalars[0], "a")
    sd = grid.regular_seeds(4, 4, 4)
    a = grid.trace_streamlines(sd, method="Heun", direction="Forward", minimum_length=0.05)
    b = lvo.trace_streamlines(vec, sp2, [scalars[0], scalars[1]], sd,       # attributes come in NAME order: a, b
                              lvo.streamline_settings("Heun", "Forward", minimum_length=0.05))
    assert same(a, b) and len(a[0]) > 500
    # StreamlinePlaneSeeder (StreamlineSeeder.cpp:52-135): regular pattern against a numpy restatement, random seeds on the plane
    sizes, spacing, box = grid.info()
    bmin, bmax = np.array(box[:3], np.float64), np.array(box[3:], np.float64)
    center, maxdim = (bmin + bmax) * 0.5, float((bmax - bmin).max())
    for normal, sl in (((0.0, 1.0, 0.0), 0.5), ((1.0, 0.0, 0.0), 0.25), ((0.0, 0.0, 1.0), 0.9)):
        nrm = np.array(normal)
        corners = np.array([[(bmax if c & 1 else bmin)[0], (bmax if c & 2 else bmin)[1], (bmax if c & 4 else bmin)[2]] for c in range(8)])
        offs = (corners - center) @ nrm
        off = offs.min() + (offs.max() - offs.min()) * sl
        a0 = np.array([1.0, 0.0, 0.0])
        a1 = np.cross(a0, nrm)
        if np.linalg.norm(a1) < 1e-3:
            a0 = np.array([0.0, 1.0, 0.0])
            a1 = np.cross(a0, nrm)
        a1 /= np.linalg.norm(a1)
        a0 = np.cross(nrm, a1)
        nx, ny = 5, 3
        exp = np.array([center + off * nrm + a0 * maxdim / (nx + 1) * (x - (nx - 1) / 2.0) + a1 * maxdim / (ny + 1) * (y - (ny - 1) / 2.0)
                        for y in range(ny) for x in range(nx)])
        got = grid.plane_seeds(normal, sl, nx, ny)
        assert got.shape == (15, 3) and np.allclose(got, exp, atol=1e-6)
        rnd = grid.plane_seeds(normal, sl, 200, 0, seed=2)
        fallback = np.all(np.abs(rnd - center) < 1e-7, axis=1)                        # 100 rejected tries -> box centre
        assert fallback.sum() < 20
        assert np.allclose(((rnd - center) @ nrm)[~fallback], off, atol=1e-5)         # on the plane
        assert (rnd >= bmin - 1e-6).all() and (rnd <= bmax + 1e-6).all()              # inside the box
        assert np.array_equal(rnd, grid.plane_seeds(normal, sl, 200, 0, seed=2))      # mt19937(seed): repeatable
        assert len(np.unique(rnd, axis=0)) > 150
    # every integrator of the reference is reachable through the plugin-side class
    c = grid.trace_streamlines(sd, method="Implicit Euler", direction="Forward", minimum_length=0.05)
    d = lvo.trace_streamlines(vec, sp2, [scalars[0], scalars[1]], sd,
                              lvo.streamline_settings("Implicit Euler", "Forward", minimum_length=0.05))
    assert same(c, d)
    e = grid.trace_streamlines(sd, method="Runge-Kutta-Fehlberg", direction="Forward", minimum_length=0.05)
    assert len(e[0]) > 500 and np.array_equal(e[2][:1], [0])


@pytest.mark.parametrize("kw", [dict(), dict(direction="Forward"), dict(direction="Backward", method="Heun"),
                                dict(loop_check_mode=0, minimum_separation_distance=0.12), dict(seeding_subsampling_factor=3),
                                dict(method="Explicit Euler", minimum_separation_distance=0.05)])
def test_max_helicity_first_seeding_bit_exact(hip_lib, kw):
    """lv_trace_streamlines_max_helicity_first: batches of seeds traced speculatively in parallel, committed in seeding order and cut
    where an earlier line claimed the cell -- against the oracle's literal one-line-after-the-other restatement: the same lines in the
    same order, positions / attributes / offsets bit for bit."""
    kw = dict(kw)
    n = 28
    v, mag, sp = abc_grid(n)
    hel = lvo.helicity_field(v, lvo.vorticity_field(v, sp))
    method, direction = kw.pop("method", "Runge-Kutta 4th Order"), kw.pop("direction", "Forward & Backward")
    ctx = capi.Context(0)
    ctx.set_flow_grid(v, sp, [mag, hel])
    seeding = capi.HelicitySeedingSettings(**kw)
    a = ctx.trace_streamlines_max_helicity_first(hel, capi.streamline_settings(method, direction, minimum_length=0.3), seeding)
    b = lvo.trace_streamlines_max_helicity_first(v, sp, [mag, hel], hel, lvo.streamline_settings(method, direction, minimum_length=0.3),
                                                 minimum_separation_distance=seeding.minimum_separation_distance,
                                                 loop_check_mode=seeding.loop_check_mode,
                                                 termination_distance_self=seeding.termination_distance_self,
                                                 seeding_subsampling_factor=seeding.seeding_subsampling_factor)
    assert same(a, b)
    assert len(a[2]) - 1 > 10 and len(a[0]) > 500
    with pytest.raises(Exception):
        ctx.trace_streamlines_max_helicity_first(hel, capi.streamline_settings("Runge-Kutta-Fehlberg", direction), seeding)
    with pytest.raises(Exception):
        ctx.trace_streamlines_max_helicity_first(hel, capi.streamline_settings(method, direction), capi.HelicitySeedingSettings(termination_check_type=4))


@pytest.mark.parametrize("check,kw", [(0, dict()), (2, dict()), (3, dict(direction="Forward")), (2, dict(direction="Backward", method="Heun")),
                                      (0, dict(minimum_separation_distance=0.03, seeding_subsampling_factor=3)),
                                      (3, dict(minimum_separation_distance=0.12, loop_check_mode=0)),
                                      (2, dict(minimum_separation_distance=0.004, seeding_subsampling_factor=4))])
def test_max_helicity_first_point_based_termination_bit_exact(hip_lib, check, kw):
    """TerminationCheckType naive / k-d tree / hashed grid (StreamlineTracingGrid.cpp:676-689, StreamlineSeeder.cpp:428-529): a line ends
    where it comes within minimum_separation_distance of a point of a finished line; the k-d tree and the hashed grid also skip such
    seeds, the naive check does not.  The speculative batches on the GPU (finished points in per-cell lists in HBM, exact cut on the
    host) against the oracle's literal loop over every finished point: the same lines bit for bit -- and not the grid-based check's."""
    kw = dict(kw)
    n = 24
    v, mag, sp = abc_grid(n)
    hel = lvo.helicity_field(v, lvo.vorticity_field(v, sp))
    method, direction = kw.pop("method", "Runge-Kutta 4th Order"), kw.pop("direction", "Forward & Backward")
    ctx = capi.Context(0)
    ctx.set_flow_grid(v, sp, [mag, hel])
    seeding = capi.HelicitySeedingSettings(termination_check_type=check, **kw)
    S = dict(minimum_length=0.3, max_num_iterations=400)
    a = ctx.trace_streamlines_max_helicity_first(hel, capi.streamline_settings(method, direction, **S), seeding)
    okw = dict(minimum_separation_distance=seeding.minimum_separation_distance, loop_check_mode=seeding.loop_check_mode,
               termination_distance_self=seeding.termination_distance_self, seeding_subsampling_factor=seeding.seeding_subsampling_factor)
    b = lvo.trace_streamlines_max_helicity_first(v, sp, [mag, hel], hel, lvo.streamline_settings(method, direction, **S),
                                                 termination_check_type=check, **okw)
    assert same(a, b)
    assert len(a[2]) - 1 > 10 and len(a[0]) > 500
    grid = lvo.trace_streamlines_max_helicity_first(v, sp, [mag, hel], hel, lvo.streamline_settings(method, direction, **S), **okw)
    assert not (len(grid[0]) == len(b[0]) and np.array_equal(grid[0], b[0]))
    # a second call on the same context starts from an empty point set
    a2 = ctx.trace_streamlines_max_helicity_first(hel, capi.streamline_settings(method, direction, **S), seeding)
    assert same(a, a2)


def rotating_grid(n=24):
    """A swirl about the z axis with a weak shear: closed or nearly closed orbits, the lines the loop checks exist for."""
    sp = (1.0 / (n - 1),) * 3
    ax = np.arange(n, dtype=np.float32) * np.float32(sp[0])
    Z, Y, X = np.meshgrid(ax, ax, ax, indexing="ij")
    c = np.float32(0.5)
    v = np.stack([-(Y - c), (X - c), np.float32(0.02) * (X - c)], axis=-1).astype(np.float32)
    mag = np.sqrt((v ** 2).sum(-1)).astype(np.float32)
    order = (np.sin(7 * X) * np.cos(5 * Y) + Z).astype(np.float32)      # the scalar the seeds are ranked by
    return v, mag, order, sp


@pytest.mark.parametrize("mode,tds,direction", [(0, 1.0, "Forward"), (1, 1.0, "Forward & Backward"), (2, 1.0, "Forward & Backward"),
                                                (2, 4.0, "Backward"), (3, 1.0, "Forward & Backward"), (4, 1.0, "Forward & Backward"),
                                                (4, 1.0, "Forward")])
def test_max_helicity_first_loop_checks_bit_exact(hip_lib, mode, tds, direction):
    """LoopCheckMode none / start point / all points / grid / curvature (StreamlineTracingGrid.cpp:588-672) on a field of closed orbits:
    the speculative parallel tracer against the oracle's sequential restatement, bit for bit; and the check in question is what ends
    the long lines (a different number of points from the run without a loop check)."""
    v, mag, order, sp = rotating_grid()
    ctx = capi.Context(0)
    ctx.set_flow_grid(v, sp, [mag, order])
    S = dict(minimum_length=0.3, max_num_iterations=300)
    seeding = capi.HelicitySeedingSettings(minimum_separation_distance=0.1, loop_check_mode=mode, termination_distance_self=tds)
    a = ctx.trace_streamlines_max_helicity_first(order, capi.streamline_settings("Runge-Kutta 4th Order", direction, **S), seeding)
    b = lvo.trace_streamlines_max_helicity_first(v, sp, [mag, order], order, lvo.streamline_settings("Runge-Kutta 4th Order", direction, **S),
                                                 minimum_separation_distance=0.1, loop_check_mode=mode, termination_distance_self=tds)
    assert same(a, b)
    assert len(a[2]) - 1 > 10
    none = lvo.trace_streamlines_max_helicity_first(v, sp, [mag, order], order, lvo.streamline_settings("Runge-Kutta 4th Order", direction, **S),
                                                    minimum_separation_distance=0.1, loop_check_mode=0, termination_distance_self=tds)
    assert int(np.diff(none[2]).max()) >= 3000                          # without a check the orbits run into the iteration limit
    if mode != 0:
        assert len(a[0]) < len(none[0])                                 # the check ended lines earlier
    with pytest.raises(Exception):
        ctx.trace_streamlines_max_helicity_first(order, capi.streamline_settings("Runge-Kutta 4th Order", direction),
                                                 capi.HelicitySeedingSettings(loop_check_mode=5))
