"""Streamline tracing: the oracle's restatement of StreamlineTracingGrid against the committed fixture
and against independent properties of the integrators and termination rules (CPU only)."""
import os

import numpy as np
import pytest

from common import GOLDEN_DIR
from oracle import lvo

G = np.load(os.path.join(GOLDEN_DIR, "flow_small.npz"))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_abc_flow_generator_and_golden_traces():
    n = int(G["n"])
    v = lvo.generate_abc_flow(n, n, n)
    assert np.uint32(np.bitwise_xor.reduce(v.reshape(-1).view(np.uint32))) == G["field_crc"]
    # AbcFlowGenerator.cpp:56-64 in float64
    g = np.arange(n) / (n - 1) * 6.0
    z, y, x = np.meshgrid(g, g, g, indexing="ij")
    A, B, Cc = np.sqrt(3.0), np.sqrt(2.0), 1.0
    ref = np.stack([A * np.sin(z) + Cc * np.cos(y), B * np.sin(x) + A * np.cos(z), Cc * np.sin(y) + B * np.cos(x)], axis=3)
    assert np.allclose(v, ref, atol=2e-6)
    mag = np.sqrt((v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]) + v[..., 2] * v[..., 2]).astype(np.float32)
    d = float(G["spacing"])
    for key, method, direction in (("rk4_both", "Runge-Kutta 4th Order", "Forward & Backward"),
                                   ("euler_fwd", "Explicit Euler", "Forward"), ("heun_bwd", "Heun", "Backward"),
                                   ("midpoint_both", "Midpoint", "Forward & Backward"),
                                   ("implicit_fwd", "Implicit Euler", "Forward")):
        pos, att, off = lvo.trace_streamlines(v, (d, d, d), [mag], G["seeds"],
                                              lvo.streamline_settings(method, direction, minimum_length=0.25))
        assert np.array_equal(off, G[key + "_off"]) and np.array_equal(bits(pos), G[key + "_pos_bits"])
        assert np.array_equal(bits(att), G[key + "_att_bits"])
        assert len(off) > 10


def uniform_field(n, vec):
    v = np.zeros((n, n, n, 3), dtype=np.float32)
    v[...] = np.asarray(vec, dtype=np.float32)
    return v


def test_uniform_field_time_step_and_boundary_clamp():
    """dt = 1 / max|v| * min(dx,dy,dz) * timeStepScale (StreamlineTracingGrid.cpp:1198): in a uniform field every step is
    one cell (times the scale); the line ends ON the box boundary (:1218-1233)."""
    n, d = 11, 0.1
    v = uniform_field(n, (2.0, 0.0, 0.0))
    seed = np.array([[0.05, 0.5, 0.5]], np.float32)
    for scale in (1.0, 0.5):
        S = lvo.streamline_settings("Runge-Kutta 4th Order", "Forward", time_step_scale=scale, minimum_length=0.1)
        pos, att, off = lvo.trace_streamlines(v, (d, d, d), [], seed, S)
        assert list(off) == [0, len(pos)]
        steps = np.diff(pos[:, 0])
        assert np.allclose(steps[:-1], d * scale, rtol=1e-5)
        assert np.allclose(pos[:, 1:], 0.5) and abs(pos[-1, 0] - 1.0) < 1e-6 and pos[-2, 0] < 1.0
    S = lvo.streamline_settings("Runge-Kutta 4th Order", "Forward & Backward", minimum_length=0.1)
    pos, _, off = lvo.trace_streamlines(v, (d, d, d), [], seed, S)
    # backward part reversed in front, seed exactly once, both ends on the boundary
    assert abs(pos[0, 0]) < 1e-6 and abs(pos[-1, 0] - 1.0) < 1e-6 and np.all(np.diff(pos[:, 0]) > 0)
    assert (np.abs(pos[:, 0] - 0.05) < 1e-7).sum() == 1
    S = lvo.streamline_settings("Runge-Kutta 4th Order", "Backward", minimum_length=0.01)
    posb, _, _ = lvo.trace_streamlines(v, (d, d, d), [], seed, S)
    assert np.array_equal(posb, pos[:len(posb)])          # the backward line alone is the reversed prefix


def test_rotation_field_integrator_order():
    """Rigid rotation about the box centre: exact solution is a circle; error after one revolution shrinks with the
    integrator's order (Euler spirals out, RK4 stays on the circle)."""
    n, d = 65, 1.0 / 64
    g = np.arange(n) * d
    z, y, x = np.meshgrid(g, g, g, indexing="ij")
    v = np.stack([-(y - 0.5), (x - 0.5), np.zeros_like(x)], axis=3).astype(np.float32)
    seed = np.array([[0.75, 0.5, 0.5]], np.float32)
    err = {}
    for method in ("Explicit Euler", "Heun", "Midpoint", "Runge-Kutta 4th Order"):
        S = lvo.streamline_settings(method, "Forward", max_num_iterations=2000, minimum_length=0.1)
        pos, _, _ = lvo.trace_streamlines(v, (d, d, d), [], seed, S)
        r = np.linalg.norm(pos[:, :2] - 0.5, axis=1)
        err[method] = float(np.abs(r - 0.25).max())
        assert len(pos) > 50
    assert err["Runge-Kutta 4th Order"] < 1e-5 < err["Heun"] * 50
    assert err["Heun"] < err["Explicit Euler"] / 20 and err["Midpoint"] < err["Explicit Euler"] / 20
    # implicit Euler spirals INWARDS by about as much as explicit Euler spirals outwards; RKF45 stays on the circle
    for method in ("Implicit Euler", "Runge-Kutta-Fehlberg"):
        S = lvo.streamline_settings(method, "Forward", max_num_iterations=2000, minimum_length=0.1)
        pos, _, _ = lvo.trace_streamlines(v, (d, d, d), [], seed, S)
        r = np.linalg.norm(pos[:, :2] - 0.5, axis=1)
        err[method] = float(np.abs(r - 0.25).max())
        assert len(pos) > 50
        if method == "Implicit Euler":
            assert r[-1] < 0.25 and 0.3 < err[method] / err["Explicit Euler"] < 3.0
    assert err["Runge-Kutta-Fehlberg"] < 1e-5


def test_rkf45_step_adaptation():
    """The RKF45 step only shrinks, and only where the truncation error estimate exceeds 2e-5 * min spacing * scale: in
    a field that is linear in the position (error estimate ~ 0) it never does, so the points are spaced like RK4's;
    with a large step through the ABC flow it does, the lines get more points than RK4's and stay close to them."""
    n, d = 33, 1.0 / 32
    g = np.arange(n) * d
    z, y, x = np.meshgrid(g, g, g, indexing="ij")
    lin = np.stack([-(y - 0.5), (x - 0.5), 0.1 + 0 * x], axis=3).astype(np.float32)
    seed = np.array([[0.7, 0.5, 0.1]], np.float32)
    a = lvo.trace_streamlines(lin, (d, d, d), [], seed, lvo.streamline_settings("Runge-Kutta-Fehlberg", "Forward", minimum_length=0.1))
    b = lvo.trace_streamlines(lin, (d, d, d), [], seed, lvo.streamline_settings("Runge-Kutta 4th Order", "Forward", minimum_length=0.1))
    assert len(a[0]) == len(b[0]) and np.allclose(a[0], b[0], atol=2e-6)
    v = lvo.generate_abc_flow(n, n, n)
    rng = np.random.default_rng(4)
    seeds = rng.uniform(0.2, 0.8, (40, 3)).astype(np.float32)
    kw = dict(time_step_scale=8.0, minimum_length=0.0, max_num_iterations=400)
    a = lvo.trace_streamlines(v, (d, d, d), [], seeds, lvo.streamline_settings("Runge-Kutta-Fehlberg", "Forward", **kw))
    b = lvo.trace_streamlines(v, (d, d, d), [], seeds, lvo.streamline_settings("Runge-Kutta 4th Order", "Forward", **kw))
    assert len(a[2]) == len(b[2]) == 41
    assert len(a[0]) > 1.2 * len(b[0])                      # smaller steps -> more points for the same lines
    # measured against a finely stepped RK4 solution the adaptive lines are far more accurate than RK4 at the same
    # nominal step
    kwf = dict(time_step_scale=0.125, minimum_length=0.0, max_num_iterations=400)
    f = lvo.trace_streamlines(v, (d, d, d), [], seeds, lvo.streamline_settings("Runge-Kutta 4th Order", "Forward", **kwf))

    def deviation(x):
        worst = 0.0
        for l in range(40):
            px, pf = x[0][x[2][l]:x[2][l + 1]], f[0][f[2][l]:f[2][l + 1]]
            assert np.array_equal(px[0], seeds[l])
            n = min(len(px), 4)                                # the first few points (before the lines leave the box)
            dist = np.linalg.norm(px[:n, None, :] - pf[None, :, :], axis=2).min(axis=1)
            worst = max(worst, float(dist.max()))
        return worst
    da, db = deviation(a), deviation(b)
    assert da < 0.004 and db > 2 * da


def test_termination_rules_and_min_length_filter():
    n, d = 17, 1.0 / 16
    g = np.arange(n) * d
    z, y, x = np.meshgrid(g, g, g, indexing="ij")
    # converging field with a sink at the centre: lines slow down and stop at the "singular point"
    v = np.stack([0.5 - x, 0.5 - y, 0.5 - z], axis=3).astype(np.float32)
    seeds = np.array([[0.1, 0.2, 0.3], [0.9, 0.8, 0.6], [0.5, 0.5, 0.5]], np.float32)
    S = lvo.streamline_settings("Runge-Kutta 4th Order", "Forward", termination_distance=100.0, minimum_length=0.05)
    pos, _, off = lvo.trace_streamlines(v, (d, d, d), [], seeds, S)
    assert len(off) == 3                      # the seed sitting on the sink yields a one-point line: dropped
    for l in range(2):
        p = pos[off[l]:off[l + 1]]
        assert np.linalg.norm(p[-1] - 0.5) < 5e-3 and len(p) < 2000
    # iteration limit: MAX_ITERATIONS = min(round(maxNumIterations / timeStepScale), 10 * maxNumIterations) (+1 start point)
    vu = uniform_field(257, (1.0, 0.0, 0.0))
    S = lvo.streamline_settings("Explicit Euler", "Forward", max_num_iterations=100, time_step_scale=0.1, minimum_length=0.0)
    p2, _, _ = lvo.trace_streamlines(vu, (1.0 / 256,) * 3, [], np.array([[0.01, 0.5, 0.5]], np.float32), S)
    # line-length limit diag * 100/2000 = 0.0866 is hit first here: 0.0866 / (0.1 / 256) = 222 steps
    assert 215 < len(p2) < 230
    S = lvo.streamline_settings("Explicit Euler", "Forward", max_num_iterations=100, time_step_scale=2.0, minimum_length=0.0)
    p3, _, _ = lvo.trace_streamlines(vu, (1.0 / 256,) * 3, [], np.array([[0.01, 0.5, 0.5]], np.float32), S)
    assert len(p3) <= 100 / 2.0 + 2 and len(p3) >= 12
    # minimum length filter drops everything shorter
    S = lvo.streamline_settings("Runge-Kutta 4th Order", "Forward", termination_distance=100.0, minimum_length=10.0)
    assert len(lvo.trace_streamlines(v, (d, d, d), [], seeds, S)[0]) == 0


def test_attributes_are_trilinear_samples():
    n, d = 9, 0.125
    g = np.arange(n) * d
    z, y, x = np.meshgrid(g, g, g, indexing="ij")
    v = uniform_field(n, (0.3, 0.2, 0.1))
    f0 = (x + 2 * y + 3 * z).astype(np.float32)          # linear: reproduced exactly by trilinear interpolation
    f1 = (x * y * z).astype(np.float32)                  # trilinear itself
    seeds = np.array([[0.11, 0.23, 0.37], [0.5, 0.1, 0.9]], np.float32)
    pos, att, off = lvo.trace_streamlines(v, (d, d, d), [f0, f1], seeds,
                                          lvo.streamline_settings(minimum_length=0.05))
    assert att.shape == (2, len(pos)) and len(off) == 3
    assert np.allclose(att[0], pos[:, 0] + 2 * pos[:, 1] + 3 * pos[:, 2], atol=1e-5)
    assert np.allclose(att[1], pos[:, 0] * pos[:, 1] * pos[:, 2], atol=1e-5)


# ------------------------------------------------------------------ streamribbons (StreamlineTracingGrid.cpp:428-530,1049-1116)
def _push_ribbon_directions(pos, helicity_at, max_hel, forward, twist=0.25, init=(0.0, 1.0, 0.0), use_helicity=True):
    """float64 restatement of _pushRibbonDirections for one traced part (positions in trace order)."""
    last = np.array(init, np.float64) / np.linalg.norm(init)
    n = len(pos)
    if n == 1:
        return [last]
    out = []
    for i in range(n):
        t = pos[1] - pos[0] if i == 0 else (pos[i] - pos[i - 1] if i == n - 1 else pos[i + 1] - pos[i - 1])
        t = t / np.linalg.norm(t)
        helper = last
        if np.linalg.norm(np.cross(helper, t)) < 1e-2:
            helper = np.array([0.0, 0.0, 1.0])
            if np.linalg.norm(np.cross(helper, t)) < 1e-2:
                helper = np.array([0.0, 1.0, 0.0])
        d = helper - np.dot(helper, t) * t
        d /= np.linalg.norm(d)
        if use_helicity:
            h = helicity_at(pos[i]) * (1.0 if forward else -1.0)
            seg = np.linalg.norm(pos[i + 1] - pos[i]) if i < n - 1 else 0.0
            ang = h / max_hel * np.pi * twist * seg / 0.005
            d = d * np.cos(ang) + np.cross(t, d) * np.sin(ang) + t * np.dot(t, d) * (1.0 - np.cos(ang))   # Rodrigues
        out.append(d)
        last = d
    return out


def test_streamribbon_directions_against_the_float64_restatement():
    n = 20
    v = lvo.generate_abc_flow(n, n, n)
    sp = (1.0 / (n - 1),) * 3
    w = lvo.vorticity_field(v, sp)
    hel = lvo.helicity_field(v, w)
    # the curl of the ABC flow is the flow itself (Beltrami), scaled by the chain rule of the grid mapping: interior points only
    h = 6.0 / (n - 1)                       # central differences of sin / cos: the derivative times sin(h) / h
    assert np.allclose(w[2:-2, 2:-2, 2:-2], 6.0 * np.sin(h) / h * v[2:-2, 2:-2, 2:-2], atol=1e-3)
    max_hel = float(np.abs(hel).max())

    def helicity_at(p):
        q = np.asarray(p, np.float64) / sp[0]
        c = np.floor(q).astype(int)
        f = q - c
        r = 0.0
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    x, y, z = c[0] + dx, c[1] + dy, c[2] + dz
                    val = hel[z, y, x] if 0 <= x < n and 0 <= y < n and 0 <= z < n else 0.0
                    r += (f[0] if dx else 1 - f[0]) * (f[1] if dy else 1 - f[1]) * (f[2] if dz else 1 - f[2]) * val
        return r

    seeds = np.random.default_rng(3).uniform(0.25, 0.75, (6, 3)).astype(np.float32)
    for direction, fwd in (("Forward", True), ("Backward", False)):
        S = lvo.streamline_settings(direction=direction, minimum_length=0.1)
        pos, att, off, rib = lvo.trace_streamribbons(v, sp, [hel], seeds, S, 0)
        assert len(off) - 1 == 6
        for l in range(6):
            p = pos[off[l]:off[l + 1]].astype(np.float64)
            traced = p if fwd else p[::-1]
            want = _push_ribbon_directions(traced, helicity_at, max_hel, fwd)
            want = np.array(want if fwd else want[::-1])
            got = rib[off[l]:off[l + 1]]
            # the direction is carried from point to point: float32 rounding accumulates along the line
            assert np.abs(got - want).max() < 2e-3, (direction, l, np.abs(got - want).max())
            t = np.gradient(p, axis=0)
            t /= np.linalg.norm(t, axis=1, keepdims=True)
            assert np.abs((got[1:-1] * t[1:-1]).sum(axis=1)).max() < 1e-3      # perpendicular to the tangent, unit length
            assert np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-5
    # both directions: the two parts meet at the seed, the backward part without its seed point in front
    S = lvo.streamline_settings(minimum_length=0.1)
    pos, att, off, rib = lvo.trace_streamribbons(v, sp, [hel], seeds, S, 0)
    pf, _, of, rf = lvo.trace_streamribbons(v, sp, [hel], seeds, lvo.streamline_settings(direction="Forward", minimum_length=0.0), 0)
    for l in range(6):
        nf = int(of[l + 1] - of[l])
        assert np.array_equal(rib[off[l + 1] - nf:off[l + 1]], rf[of[l]:of[l + 1]])


def test_max_helicity_first_seeding_properties():
    """StreamlineMaxHelicityFirstSeeder + the decreasing-helicity tracer (the oracle's sequential restatement): the first line starts
    at the interior grid sample of largest helicity; every line's seed cell was free when it was seeded and no line runs through a cell
    an EARLIER line has claimed (checked by replaying the occupancy grid in float64 with the cells recomputed independently); a
    larger separation distance gives fewer lines; the sub-sampled seeder seeds block centres."""
    n = 20
    v = lvo.generate_abc_flow(n, n, n)
    d = 1.0 / (n - 1)
    sp = (d, d, d)
    hel = lvo.helicity_field(v, lvo.vorticity_field(v, sp))
    mag = np.sqrt((v * v).sum(axis=3)).astype(np.float32)
    S = lvo.streamline_settings("Runge-Kutta 4th Order", "Forward & Backward", minimum_length=0.3)
    r = 0.08
    pos, att, off = lvo.trace_streamlines_max_helicity_first(v, sp, [mag], hel, S, minimum_separation_distance=r)
    assert 10 < len(off) - 1 < 400 and att.shape == (1, len(pos))
    # the first seed: the largest helicity among the interior samples, at boxMin + dims * index / n
    inner = hel[1:-1, 1:-1, 1:-1]
    z, y, x = np.unravel_index(np.argmax(inner), inner.shape)
    seed0 = np.array([x + 1, y + 1, z + 1], np.float32) / np.float32(n) * np.float32((n - 1) * d)
    first = pos[off[0]:off[1]]
    assert np.abs(first - seed0).sum(axis=1).min() < 1e-6
    # replay: the cells within r of every earlier line's points are taken; a later line never has a point in a taken cell, except the
    # boundary point it may end with (appended without the test)
    cells = np.zeros((n - 1, n - 1, n - 1), dtype=bool)          # [z, y, x]
    lo = np.arange(n - 1) * d
    for l in range(len(off) - 1):
        p = pos[off[l]:off[l + 1]].astype(np.float64)
        c = np.clip((p / d).astype(int), 0, n - 2)
        taken = cells[c[:, 2], c[:, 1], c[:, 0]]
        on_boundary = ((p <= 1e-6) | (p >= (n - 1) * d - 1e-6)).any(axis=1)
        assert not (taken & ~on_boundary).any(), l
        for q in p:
            dist = [np.maximum(np.maximum(lo - q[a], q[a] - (lo + d)), 0.0) for a in range(3)]
            d2 = dist[2][:, None, None] ** 2 + dist[1][None, :, None] ** 2 + dist[0][None, None, :] ** 2
            cells |= d2 <= np.float64(np.float32(r)) ** 2 * (1 + 1e-6)
    fewer = lvo.trace_streamlines_max_helicity_first(v, sp, [mag], hel, S, minimum_separation_distance=0.2)
    assert len(fewer[2]) < len(off)
    sub = lvo.trace_streamlines_max_helicity_first(v, sp, [mag], hel, S, minimum_separation_distance=r, seeding_subsampling_factor=4)
    nc = (n - 1) // 4
    centres = (np.arange(nc) + 0.5) / nc * (n - 1) * d
    s0 = sub[0][sub[2][0]:sub[2][1]]
    assert min(np.abs(q[:, None] - centres[None, :]).min(axis=1).max() for q in s0) < 1e-5     # one point of the first line is a block centre
    none = lvo.trace_streamlines_max_helicity_first(v, sp, [mag], hel, S, minimum_separation_distance=r, loop_check_mode=0)
    assert len(none[0]) >= len(pos)                                                               # no loop check: lines only get longer


@pytest.mark.parametrize("check", [0, 2, 3])
def test_point_based_termination_checks_properties(check):
    """TerminationCheckType naive (0) / k-d tree (2) / hashed grid (3) of the oracle's sequential tracer, against the definition stated
    independently in numpy: no point of a line -- except a boundary point it may end with, appended without the test -- lies closer
    than minimumSeparationDistance to a point of an EARLIER line (float32, sqrt((dx dx + dy dy) + dz dz) < r); every line ends for a
    reason (boundary, or its next step would have been too close / the iteration limit); 2 and 3 are the same search and skip seeds
    within r of a finished point -- which the naive check traces, only to end them at their first point (an empty line fails the
    minimum length): the three give the same lines, and not the occupancy grid's."""
    n = 16
    v = lvo.generate_abc_flow(n, n, n)
    d = 1.0 / (n - 1)
    sp = (d, d, d)
    hel = lvo.helicity_field(v, lvo.vorticity_field(v, sp))
    mag = np.sqrt((v * v).sum(axis=3)).astype(np.float32)
    S = lvo.streamline_settings("Runge-Kutta 4th Order", "Forward & Backward", minimum_length=0.3, max_num_iterations=300)
    r = np.float32(0.06)
    pos, att, off = lvo.trace_streamlines_max_helicity_first(v, sp, [mag], hel, S, minimum_separation_distance=float(r), termination_check_type=check)
    assert 10 < len(off) - 1 and att.shape == (1, len(pos))
    hi = np.float32((n - 1) * d)
    for l in range(1, len(off) - 1):
        p = pos[off[l]:off[l + 1]]
        earlier = pos[:off[l]]
        dx = p[:, None, 0] - earlier[None, :, 0]; dy = p[:, None, 1] - earlier[None, :, 1]; dz = p[:, None, 2] - earlier[None, :, 2]
        dist = np.sqrt((dx * dx + dy * dy) + dz * dz)            # float32 throughout
        close = (dist < r).any(axis=1)
        ends = np.zeros(len(p), dtype=bool); ends[0] = ends[-1] = True
        on_boundary = ((p <= 1e-6) | (p >= hi - 1e-6)).any(axis=1)
        assert not (close & ~(ends & on_boundary)).any(), l
    same23 = lvo.trace_streamlines_max_helicity_first(v, sp, [mag], hel, S, minimum_separation_distance=float(r), termination_check_type=5 - check if check else 0)
    if check:
        assert np.array_equal(same23[0], pos) and np.array_equal(same23[2], off)     # k-d tree and hashed grid: one predicate
        naive = lvo.trace_streamlines_max_helicity_first(v, sp, [mag], hel, S, minimum_separation_distance=float(r), termination_check_type=0)
        assert np.array_equal(naive[0], pos) and np.array_equal(naive[2], off)
    grid = lvo.trace_streamlines_max_helicity_first(v, sp, [mag], hel, S, minimum_separation_distance=float(r))
    assert not np.array_equal(grid[2], off)                                          # the occupancy grid is a different (coarser) rule


def test_loop_check_modes_of_the_max_helicity_first_tracer():
    """Oracle: the five loop check modes (StreamlineTracingGrid.cpp:588-672) on a swirl with closed orbits.  Without a check the orbiting
    lines run into the iteration limit; each check ends them after about one turn (start point, all points), when the line re-enters a
    cell it left more than 32 cells ago (grid), or when length-weighted turning angles add up to 2.5 after 100 segments (curvature)."""
    n = 24
    sp = (1.0 / (n - 1),) * 3
    ax = np.arange(n, dtype=np.float32) * np.float32(sp[0])
    Z, Y, X = np.meshgrid(ax, ax, ax, indexing="ij")
    c = np.float32(0.5)
    v = np.stack([-(Y - c), (X - c), np.float32(0.02) * (X - c)], axis=-1).astype(np.float32)
    mag = np.sqrt((v ** 2).sum(-1)).astype(np.float32)
    order = (np.sin(7 * X) * np.cos(5 * Y) + Z).astype(np.float32)
    S = lvo.streamline_settings("Runge-Kutta 4th Order", "Forward", minimum_length=0.3, max_num_iterations=300)
    longest = {}
    for mode, tds in ((0, 1.0), (1, 1.0), (2, 4.0), (3, 1.0), (4, 1.0)):
        pos, att, off = lvo.trace_streamlines_max_helicity_first(v, sp, [mag], order, S, minimum_separation_distance=0.1,
                                                                 loop_check_mode=mode, termination_distance_self=tds)[:3]
        longest[mode] = int(np.diff(off).max())
        assert np.isfinite(pos).all() and (pos >= -1e-6).all() and (pos <= 1.0 + 1e-6).all()
    assert longest[0] == 3001                                   # MAX_ITERATIONS + 1 points
    # one turn of the seed's orbit at this step width is ~100 points (forward part only)
    assert 60 < longest[1] < 200 and 60 < longest[2] < 200
    assert longest[1] < longest[3] < longest[4] < longest[0]    # grid: a few turns; curvature: 2.5 rad m of turning


def test_loop_checks_independent_replay():
    """Independent float64 replay of the loop checks, written from StreamlineTracingGrid.cpp:588-672 (not from the oracle): every point a
    line holds was pushed AFTER its own check, so replaying the check at each pushed point must never fire (knife-edge cases aside), and
    the lines the checks cut short must stop within a step of where the replay fires for the first time on the longer line traced
    without a check."""
    n = 24
    sp = (1.0 / (n - 1),) * 3
    ax = np.arange(n, dtype=np.float32) * np.float32(sp[0])
    Z, Y, X = np.meshgrid(ax, ax, ax, indexing="ij")
    c = np.float32(0.5)
    v = np.stack([-(Y - c), (X - c), np.float32(0.02) * (X - c)], axis=-1).astype(np.float32)
    mag = np.sqrt((v ** 2).sum(-1)).astype(np.float32)
    order = (np.sin(7 * X) * np.cos(5 * Y) + Z).astype(np.float32)
    S = lvo.streamline_settings("Runge-Kutta 4th Order", "Forward", minimum_length=0.3, max_num_iterations=300)
    diag = float(np.sqrt(3.0))                                   # the box is the unit cube
    cells = n - 1

    def cell_of(p):
        g = np.clip((p.astype(np.float64) / np.array(sp)).astype(np.int64), 0, cells - 1)
        return int(g[0] + g[1] * cells + g[2] * cells * cells)

    def first_fire(mode, pts, tds):
        """index of the first point at which the check would have stopped the line (None: never)"""
        P = pts.astype(np.float64)
        r = diag / 100.0 * tds
        visited, queue, old = set(), [], None
        curv, segs = 0.0, 0
        for j in range(2, len(P)):
            cur, back = P[j], P[j - 1]
            if mode == 1:
                d0 = (P[1] - P[0]) / np.linalg.norm(P[1] - P[0])
                dn = (cur - back) / np.linalg.norm(cur - back)
                if np.dot(d0, cur - P[0]) < 0 and np.linalg.norm(cur - P[0]) < r and np.dot(d0, dn) > 0:
                    return j
            elif mode == 3:
                cpos = cell_of(pts[j])
                occupied = cpos in visited
                visited.add(cpos)
                if occupied and cpos not in queue:
                    return j
                if cpos != old:
                    if len(queue) == 32:
                        queue.pop(0)
                    queue.append(cpos)
                old = cpos
            elif mode == 4:
                a, b = back - P[j - 2], cur - back
                la, lb = np.linalg.norm(a), np.linalg.norm(b)
                if la > 1e-8:
                    a = a / la
                if lb > 1e-8:
                    b = b / lb
                curv += float(np.arccos(np.clip(np.dot(a, b), -1.0, 1.0))) * (la + lb)
                segs += 1
                if segs > 100 and curv > 2.5:
                    return j
        return None

    none = lvo.trace_streamlines_max_helicity_first(v, sp, [mag], order, S, minimum_separation_distance=0.1, loop_check_mode=0)
    for mode in (1, 3, 4):
        pos, att, off = lvo.trace_streamlines_max_helicity_first(v, sp, [mag], order, S, minimum_separation_distance=0.1,
                                                                 loop_check_mode=mode, termination_distance_self=1.0)[:3]
        fired_early = 0
        for a, b in zip(off[:-1], off[1:]):
            f = first_fire(mode, pos[a:b], 1.0)
            fired_early += f is not None and f < (b - a) - 1     # (the last point may sit on the knife edge in float64)
        assert fired_early == 0, (mode, fired_early)
        # the first line of both runs starts at the same seed: without a check it runs into the iteration limit, with the check it
        # ends where the replay of the long line fires for the first time
        long_line = none[0][none[2][0]:none[2][1]]
        f = first_fire(mode, long_line, 1.0)
        assert f is not None and abs((off[1] - off[0]) - f) <= 1, (mode, f, off[1] - off[0])
