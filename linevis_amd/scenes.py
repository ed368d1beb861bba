"""Deterministic synthetic line sets for BASELINE.json's configs.

Every generator returns `Trajectories`: positions float32 [P,3], one attribute float32 [P] in [0,1] and
line_offsets uint32 [L+1] -- the SoA equivalent of `std::vector<Trajectory>`
(src/Loaders/TrajectoryFile.hpp:38-43).  Positions are NOT yet normalised; `normalize()` applies
src/Loaders/TrajectoryFile.cpp:106-125.  Also: .binlines v1 reader/writer (src/Loaders/BinLinesLoader.cpp:41-63).
"""
import struct
from dataclasses import dataclass

import numpy as np


@dataclass
class Trajectories:
    positions: np.ndarray      # float32 [P, 3]
    attributes: np.ndarray     # float32 [P]
    line_offsets: np.ndarray   # uint32 [L + 1]
    ribbon_directions: np.ndarray = None   # float32 [P, 3] or None: band data (LineDataFlow::ribbonsDirections)
    vertices_normalized: bool = False      # verticesNormalized of a version-2 .binlines file (TrajectoryFile.cpp:656 skips normalisation)

    @property
    def num_lines(self):
        return len(self.line_offsets) - 1

    @property
    def num_points(self):
        return int(self.positions.shape[0])

    @property
    def num_segments_upper(self):
        return self.num_points - self.num_lines


def _pack(lines, attrs):
    offs = np.zeros(len(lines) + 1, dtype=np.uint32)
    offs[1:] = np.cumsum([len(l) for l in lines])
    pos = np.ascontiguousarray(np.concatenate(lines, axis=0), dtype=np.float32)
    att = np.ascontiguousarray(np.concatenate(attrs, axis=0), dtype=np.float32)
    return Trajectories(pos, att, offs)


def _uniform_offsets(n_lines, n_pts):
    return (np.arange(n_lines + 1, dtype=np.uint64) * n_pts).astype(np.uint32)


def normalize(tr):
    """normalizeTrajectoriesVertexPositions, TrajectoryFile.cpp:106-125 (float32 arithmetic)."""
    p = tr.positions.astype(np.float32)
    mn = p.min(axis=0)
    mx = p.max(axis=0)
    translation = -((mn + mx) / np.float32(2.0))
    scale3 = np.float32(0.5) / (mx - mn)
    scale = np.float32(scale3.min())
    out = ((p + translation) * scale).astype(np.float32)
    return Trajectories(np.ascontiguousarray(out), tr.attributes, tr.line_offsets, tr.ribbon_directions)


def normalize_attributes(att):
    a = att.astype(np.float64)
    lo, hi = a.min(), a.max()
    if hi <= lo:
        return np.zeros_like(att, dtype=np.float32)
    return ((a - lo) / (hi - lo)).astype(np.float32)


# ------------------------------------------------------------------ C1: lattice
def lattice(n=32, points_per_line=32):
    """n x n straight lines along z through a uniform grid, `points_per_line` points each."""
    g = (np.arange(n, dtype=np.float64) + 0.5) / n - 0.5
    z = np.linspace(-0.5, 0.5, points_per_line)
    xx, yy = np.meshgrid(g, g, indexing="xy")
    xs = np.repeat(xx.reshape(-1), points_per_line)
    ys = np.repeat(yy.reshape(-1), points_per_line)
    zs = np.tile(z, n * n)
    pos = np.stack([xs, ys, zs], axis=1).astype(np.float32)
    att = np.tile(np.linspace(0.0, 1.0, points_per_line), n * n).astype(np.float32)
    return Trajectories(pos, att, _uniform_offsets(n * n, points_per_line))


# ------------------------------------------------------------------ C2: helix bundle
def helix_bundle(n_lines=100, points_per_line=1001, seed=12345, turns=4.0):
    """Helices about the z axis: radii U(0.02,0.2), pitch U(0.05,0.2), phase U(0,2pi)."""
    rng = np.random.default_rng(seed)
    radius = rng.uniform(0.02, 0.2, n_lines)
    pitch = rng.uniform(0.05, 0.2, n_lines)
    phase = rng.uniform(0.0, 2.0 * np.pi, n_lines)
    cx = rng.uniform(-0.15, 0.15, n_lines)
    cy = rng.uniform(-0.15, 0.15, n_lines)
    s = np.linspace(0.0, 1.0, points_per_line)
    ang = 2.0 * np.pi * turns * s[None, :] + phase[:, None]
    x = cx[:, None] + radius[:, None] * np.cos(ang)
    y = cy[:, None] + radius[:, None] * np.sin(ang)
    z = (s[None, :] - 0.5) * (pitch[:, None] * turns)
    pos = np.stack([x, y, z], axis=2).reshape(-1, 3).astype(np.float32)
    att = np.tile(s, n_lines).astype(np.float32)
    return Trajectories(pos, att, _uniform_offsets(n_lines, points_per_line))


# ------------------------------------------------------------------ C3/C4: tornado-style streamlines
def _tornado_velocity(p):
    """Analytic tornado-style swirl on the unit cube (centre line wanders with height, funnel widens upward).

    Own formulation in the spirit of Crawfis' tornado data set; bounded: w ~ z(1-z) keeps z in (0,1) and a
    radial restoring term keeps lines on the funnel.
    """
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    xc = 0.5 + 0.1 * np.sin(10.0 * z)
    yc = 0.5 + 0.1 * np.cos(3.0 * z)
    dx, dy = x - xc, y - yc
    rho = np.sqrt(dx * dx + dy * dy) + 1e-6
    funnel = 0.05 + 0.35 * z * z + 0.03 * z * np.sin(8.0 * z)
    swirl = 1.0 / (0.08 + rho)              # angular speed, fast near the core
    radial = -1.5 * (rho - funnel)          # pull towards the funnel surface
    u = -dy * swirl + dx / rho * radial
    v = dx * swirl + dy / rho * radial
    w = 0.9 * z * (1.0 - z) + 0.02
    w = np.where(z > 0.98, 0.0, w)
    return np.stack([u, v, w], axis=1)


def _rk4_lines(velocity, seeds, n_steps, h):
    p = seeds.astype(np.float64).copy()
    n = p.shape[0]
    out = np.empty((n, n_steps + 1, 3), dtype=np.float64)
    mag = np.empty((n, n_steps + 1), dtype=np.float64)
    out[:, 0] = p
    mag[:, 0] = np.linalg.norm(velocity(p), axis=1)
    for i in range(n_steps):
        k1 = velocity(p)
        k2 = velocity(p + 0.5 * h * k1)
        k3 = velocity(p + 0.5 * h * k2)
        k4 = velocity(p + h * k3)
        p = p + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        out[:, i + 1] = p
        mag[:, i + 1] = np.linalg.norm(k1, axis=1)
    return out, mag


def tornado(n_lines=1000, points_per_line=1001, seed=12345, h=0.004):
    """1000 seeds x RK4 1000 steps -> 1 001 000 points / 1 000 000 segments at the defaults."""
    rng = np.random.default_rng(seed)
    seeds = np.stack([rng.uniform(0.2, 0.8, n_lines), rng.uniform(0.2, 0.8, n_lines),
                      rng.uniform(0.02, 0.6, n_lines)], axis=1)
    out, mag = _rk4_lines(_tornado_velocity, seeds, points_per_line - 1, h)
    out = out[:, :, [0, 2, 1]]  # height along +y: the default camera looks at the funnel from the side
    pos = out.reshape(-1, 3).astype(np.float32)
    att = normalize_attributes(np.log1p(mag.reshape(-1)))
    return Trajectories(pos, att, _uniform_offsets(n_lines, points_per_line))


# ------------------------------------------------------------------ C5: Rayleigh-Benard-like rolls
def _rolls_velocity(p, k=8.0):
    """Cellular convection-roll flow (k x k cells) on [0,1]^3: u = -c sin(ax) cos(pi z), v = -c sin(ay) cos(pi z),
    w = (cos(ax) + cos(ay)) sin(pi z) with c = pi / a (divergence free, tangential at the box walls), plus a
    gentle horizontal drift that couples neighbouring rolls so lines do not close on themselves."""
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    a = np.pi * k
    c = np.pi / a
    u = -c * np.sin(a * x) * np.cos(np.pi * z) + 0.02 * np.sin(2.0 * np.pi * y) * np.sin(np.pi * x)
    v = -c * np.sin(a * y) * np.cos(np.pi * z) + 0.02 * np.sin(2.0 * np.pi * x) * np.sin(np.pi * y)
    w = (np.cos(a * x) + np.cos(a * y)) * np.sin(np.pi * z)
    return np.stack([u, v, w], axis=1)


def _rolls_velocity_min_speed(p, min_speed=0.25):
    """_rolls_velocity with its magnitude raised to at least min_speed (direction kept): near the stagnation lines of
    the rolls consecutive points would otherwise fall below the 1e-4 degenerate-tangent threshold of
    LineDataFlow.cpp:2160 and the set would lose segments (config 5 is specified as exactly 5 M)."""
    v = _rolls_velocity(p) + np.array([1e-3, 7e-4, 0.0])
    m = np.linalg.norm(v, axis=1, keepdims=True)
    return v * np.maximum(1.0, min_speed / np.maximum(m, 1e-12))


def rayleigh_benard(n_lines=5000, points_per_line=1001, seed=12345, h=0.002):
    rng = np.random.default_rng(seed)
    seeds = rng.uniform(0.02, 0.98, (n_lines, 3))
    out, mag = _rk4_lines(_rolls_velocity_min_speed, seeds, points_per_line - 1, h)
    out = np.clip(out, -0.25, 1.25)
    pos = out.reshape(-1, 3).astype(np.float32)
    att = normalize_attributes(mag.reshape(-1))
    return Trajectories(pos, att, _uniform_offsets(n_lines, points_per_line))


# ------------------------------------------------------------------ small random scene for tests
def random_curves(n_lines=24, points_per_line=40, seed=7, extent=0.45, step=0.03):
    """Smooth random walks; includes direction changes so caps and elbows get exercised."""
    rng = np.random.default_rng(seed)
    lines, attrs = [], []
    for _ in range(n_lines):
        p = rng.uniform(-extent, extent, 3)
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        pts = [p.copy()]
        for _ in range(points_per_line - 1):
            d = d + 0.35 * rng.normal(size=3)
            d /= np.linalg.norm(d)
            p = np.clip(p + step * d, -extent, extent)
            pts.append(p.copy())
        lines.append(np.array(pts))
        attrs.append(np.linspace(rng.uniform(0, 0.5), rng.uniform(0.5, 1.0), points_per_line))
    return _pack(lines, attrs)


def twisted_ribbons(tr, twist=6.0, seed=3):
    """Synthetic band data for a set of trajectories: per point a unit ribbon direction perpendicular to the tangent, parallel
    transported along the line and rotated about the tangent by `twist` radians per unit arc length (plus a random phase
    per line) -- the role the vorticity-driven streamribbon tracer plays in the reference (StreamlineTracingGrid.cpp:451-530)."""
    rng = np.random.default_rng(seed)
    pos = tr.positions.astype(np.float64)
    out = np.zeros_like(pos)
    for li in range(tr.num_lines):
        b, e = int(tr.line_offsets[li]), int(tr.line_offsets[li + 1])
        n = e - b
        if n < 2:
            if n == 1:
                out[b] = (0.0, 1.0, 0.0)
            continue
        p = pos[b:e]
        t = np.empty_like(p)
        t[0], t[-1], t[1:-1] = p[1] - p[0], p[-1] - p[-2], p[2:] - p[:-2]
        ln = np.linalg.norm(t, axis=1, keepdims=True)
        t = np.where(ln > 1e-12, t / np.maximum(ln, 1e-12), np.array([[1.0, 0.0, 0.0]]))
        arc = np.concatenate([[0.0], np.cumsum(np.linalg.norm(p[1:] - p[:-1], axis=1))])
        nrm = np.array([0.0, 1.0, 0.0]) if abs(t[0, 1]) < 0.9 else np.array([1.0, 0.0, 0.0])
        phase = rng.uniform(0.0, 2.0 * np.pi)
        for i in range(n):
            nrm = nrm - np.dot(nrm, t[i]) * t[i]
            nl = np.linalg.norm(nrm)
            nrm = nrm / nl if nl > 1e-9 else np.cross(t[i], [0.0, 0.0, 1.0])
            ang = phase + twist * arc[i]
            out[b + i] = np.cos(ang) * nrm + np.sin(ang) * np.cross(t[i], nrm)
    return Trajectories(tr.positions, tr.attributes, tr.line_offsets, out.astype(np.float32))


# ------------------------------------------------------------------ .binlines (BinLinesLoader.cpp:41-125,127-247)
def write_binlines(path, tr, vertices_normalized=None):
    """Version 1: u32 version, u32 numTrajectories, u32 numAttributes, then per trajectory u32 numPoints, vec3[numPoints],
    float[numPoints] per attribute.  With band data version 2: the same, then u32 verticesNormalized, u32 hasAttributeNames (0),
    u32 hasRibbonData (1), vec3[numPoints] ribbon directions per trajectory, three u32 zeros (no outline mesh)."""
    v2 = tr.ribbon_directions is not None
    if vertices_normalized is None:
        vertices_normalized = bool(getattr(tr, "vertices_normalized", False))
    with open(path, "wb") as f:
        f.write(struct.pack("<III", 2 if v2 else 1, tr.num_lines, 1))
        for i in range(tr.num_lines):
            b, e = int(tr.line_offsets[i]), int(tr.line_offsets[i + 1])
            f.write(struct.pack("<I", e - b))
            f.write(np.ascontiguousarray(tr.positions[b:e], dtype="<f4").tobytes())
            f.write(np.ascontiguousarray(tr.attributes[b:e], dtype="<f4").tobytes())
        if v2:
            f.write(struct.pack("<III", 1 if vertices_normalized else 0, 0, 1))   # verticesNormalized: the real state of the positions
            f.write(np.ascontiguousarray(tr.ribbon_directions, dtype="<f4").tobytes())   # stored per trajectory = contiguous
            f.write(struct.pack("<III", 0, 0, 0))


def read_binlines(path, attribute_index=0):
    """Reads v1 and v2 files (v2: ribbon directions kept, attribute names and the outline mesh skipped)."""
    with open(path, "rb") as f:
        data = f.read()
    (version,) = struct.unpack_from("<I", data, 0)
    if version not in (1, 2):
        raise ValueError("loadTrajectoriesFromBinLines: invalid version number %d" % version)
    n_traj, n_attr = struct.unpack_from("<II", data, 4)
    off = 12
    lines, attrs = [], []
    for _ in range(n_traj):
        (n,) = struct.unpack_from("<I", data, off)
        off += 4
        lines.append(np.frombuffer(data, dtype="<f4", count=3 * n, offset=off).reshape(n, 3))
        off += 12 * n
        sel = None
        for a in range(n_attr):
            arr = np.frombuffer(data, dtype="<f4", count=n, offset=off)
            off += 4 * n
            if a == attribute_index:
                sel = arr
        attrs.append(sel if sel is not None else np.zeros(n, dtype=np.float32))
    tr = _pack(lines, attrs)
    if version == 2:
        # loadTrajectoriesFromBinLinesV2, BinLinesLoader.cpp:68-125
        _normalized, has_names = struct.unpack_from("<II", data, off)
        off += 8
        if has_names:
            for _ in range(n_attr if n_traj else 0):       # sgl::BinaryReadStream::read(std::string&): u32 length + bytes
                (ln,) = struct.unpack_from("<I", data, off)
                off += 4 + ln
        (has_ribbons,) = struct.unpack_from("<I", data, off)
        off += 4
        if has_ribbons:
            rib = np.frombuffer(data, dtype="<f4", count=3 * tr.num_points, offset=off).reshape(-1, 3)
            tr = Trajectories(tr.positions, tr.attributes, tr.line_offsets, np.array(rib, dtype=np.float32))
        tr.vertices_normalized = bool(_normalized)
    return tr


def load_flow_trajectories(path):
    """loadFlowTrajectoriesFromFile for .binlines (TrajectoryFile.cpp:634-668): normalised unless the file says it already is."""
    tr = read_binlines(path)
    if tr.vertices_normalized:
        return tr
    out = normalize(tr)
    out.vertices_normalized = True
    return out


# ------------------------------------------------------------------ .obj polylines (ObjLoader.cpp:36-186)
def write_obj(path, tr, attribute_name="attr"):
    """One 'v' + 'vt' pair per point and one 'l' statement per trajectory (1-based indices), as LineVis exports them."""
    with open(path, "w") as f:
        f.write("# polylines\na %s\n" % attribute_name)
        for p, a in zip(tr.positions, tr.attributes):
            f.write("v %.9g %.9g %.9g\nvt %.9g\n" % (p[0], p[1], p[2], a))
        for i in range(tr.num_lines):
            b, e = int(tr.line_offsets[i]), int(tr.line_offsets[i + 1])
            f.write("g line%d\nl %s\n" % (i, " ".join(str(k + 1) for k in range(b, e))))


def read_obj(path, attribute_index=0):
    verts, attrs, lines, lattr = [], [], [], []
    for line in open(path).read().replace("\r", "\n").split("\n"):
        if line.startswith("vt"):
            attrs.append([float(x) for x in line[2:].split()])
        elif line.startswith("vn"):
            continue
        elif line.startswith("v"):
            verts.append([float(x) for x in line[2:].split()[:3]])
        elif line.startswith("l"):
            idx = [int(x) - 1 for x in line[2:].split()]
            idx = [i for i in idx if max(abs(c) for c in verts[i]) <= 1e10]
            lines.append(np.array([verts[i] for i in idx], dtype=np.float32).reshape(-1, 3))
            lattr.append(np.array([attrs[i][attribute_index] if attrs else 0.0 for i in idx], dtype=np.float32))
    return _pack(lines, lattr)
