"""ctypes binding of the CPU oracle (oracle/lv_oracle.h).

TEST INFRASTRUCTURE, NOT PRODUCT: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.  PARITY UNPINNED (see lv_oracle.h): the reference
holds no golden vectors for this path and cannot be built in this image.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblv_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("lv_oracle.cpp", "lv_oracle_tri.cpp", "lv_oracle_flow.cpp", "lv_oracle_common.h", "lv_oracle_tri.h", "lv_oracle.h", "lv_oracle_prism.h",
                                          "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


class LinePoint(C.Structure):
    _fields_ = [("linePosition", C.c_float * 3), ("lineAttribute", C.c_float),
                ("lineTangent", C.c_float * 3), ("lineRotation", C.c_float),
                ("lineNormal", C.c_float * 3), ("lineStartIndex", C.c_uint32)]


LINE_POINT_DTYPE = np.dtype([("linePosition", "<f4", 3), ("lineAttribute", "<f4"),
                             ("lineTangent", "<f4", 3), ("lineRotation", "<f4"),
                             ("lineNormal", "<f4", 3), ("lineStartIndex", "<u4")])
assert LINE_POINT_DTYPE.itemsize == 48 and C.sizeof(LinePoint) == 48
# TubeTriangleVertexData, LineRenderData.hpp:171-176
TUBE_VERTEX_DTYPE = np.dtype([("vertexPosition", "<f4", 3), ("vertexLinePointIndex", "<u4"),
                              ("vertexNormal", "<f4", 3), ("phi", "<f4")])
assert TUBE_VERTEX_DTYPE.itemsize == 32


class Params(C.Structure):
    _fields_ = [
        ("view", C.c_float * 16), ("proj", C.c_float * 16),
        ("fovY", C.c_float), ("nearDist", C.c_float), ("farDist", C.c_float),
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("background", C.c_float * 4), ("lineWidth", C.c_float),
        ("maxDepthComplexity", C.c_uint32), ("numSamplesPerFrame", C.c_uint32), ("frameNumber", C.c_uint32),
        ("useJitteredRays", C.c_uint32), ("useDeterministicSampling", C.c_uint32),
        ("useCappedTubes", C.c_uint32), ("useHalos", C.c_uint32), ("useDepthCues", C.c_uint32),
        ("useAmbientOcclusion", C.c_uint32),
        ("depthCueStrength", C.c_float), ("minDepth", C.c_float), ("maxDepth", C.c_float),
        ("aoStrength", C.c_float), ("aoGamma", C.c_float),
        ("attrMin", C.c_float), ("attrMax", C.c_float),
        ("aoSamplesPerFrame", C.c_uint32), ("aoIterations", C.c_uint32), ("aoUseDistance", C.c_uint32),
        ("aoJitterPrimary", C.c_uint32), ("tubeNumSubdivisions", C.c_uint32), ("aoRadius", C.c_float),
        ("ppllMaxNumFrags", C.c_uint32), ("ppllLinkedListSize", C.c_uint32),
        ("ppllTileW", C.c_uint32), ("ppllTileH", C.c_uint32),
        ("useBands", C.c_uint32), ("useEllipticTubes", C.c_uint32),
        ("bandWidth", C.c_float), ("minBandThickness", C.c_float), ("minThickness", C.c_float),
        ("lssGeometry", C.c_uint32),
        ("useHelicityBands", C.c_uint32), ("numSubdivisionsBands", C.c_uint32),
        ("separatorBaseWidth", C.c_float), ("helicityRotationFactor", C.c_float),
        ("uniformHelicityBandWidth", C.c_uint32),
        ("ppllSortingMode", C.c_uint32),
        ("ppllFragmentSource", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [("raysTraced", C.c_uint64), ("nodesVisited", C.c_uint64), ("primsTested", C.c_uint64),
                ("hitsShaded", C.c_uint64), ("fragments", C.c_uint64),
                ("maxDepthComplexity", C.c_uint32), ("bvhDepth", C.c_uint32)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class StreamlineSettings(C.Structure):
    """StreamlineTracingSettings (StreamlineTracingDefines.hpp:144-177), the fields the tracer reads."""
    _fields_ = [("integrationMethod", C.c_uint32), ("integrationDirection", C.c_uint32), ("timeStepScale", C.c_float),
                ("maxNumIterations", C.c_int32), ("terminationDistance", C.c_float), ("minimumLength", C.c_float)]


INTEGRATION_METHODS = {"Explicit Euler": 0, "Implicit Euler": 1, "Heun": 2, "Midpoint": 3, "Runge-Kutta 4th Order": 4,
                       "Runge-Kutta-Fehlberg": 5}
INTEGRATION_DIRECTIONS = {"Forward": 0, "Backward": 1, "Forward & Backward": 2}


def streamline_settings(method="Runge-Kutta 4th Order", direction="Forward & Backward", time_step_scale=1.0,
                        max_num_iterations=2000, termination_distance=1.0, minimum_length=0.7):
    return StreamlineSettings(INTEGRATION_METHODS[method], INTEGRATION_DIRECTIONS[direction], time_step_scale,
                              max_num_iterations, termination_distance, minimum_length)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    vp, u32, f32, i32 = C.c_void_p, C.c_uint32, C.c_float, C.c_int
    L.lvo_set_deviation_switches.argtypes = [i32, i32]
    L.lvo_set_ao_feature_outputs.argtypes = [vp, vp]
    L.lvo_set_svgf_feature_outputs.argtypes = [vp, vp, vp, vp, i32, u32, vp]
    L.lvo_mat4_mul.argtypes = [vp, vp, vp]
    L.lvo_svgf_denoise.argtypes = [u32, u32, vp, vp, vp, vp, vp, i32, f32, f32, vp, vp, vp, vp, vp]
    L.lvo_eaw_denoise.argtypes = [u32, u32, vp, vp, vp, i32, f32, f32, f32, i32, i32, i32, i32, u32, u32, u32, u32, vp]
    L.lvo_compute_fragment_color_batch.argtypes = [vp, vp, C.c_uint64] + [vp] * 8
    L.lvo_compute_fragment_color_raster_batch.argtypes = [vp, vp, C.c_uint64] + [vp] * 9
    L.lvo_ribbon_of_rays.argtypes = [vp, vp, C.c_uint64, vp, vp, f32, vp, vp, vp]
    L.lvo_set_ppll_fragment_colour_variant.argtypes = [i32]
    L.lvo_set_ppll_prebaked_ao.argtypes = [vp, vp, u32, u32, u32]
    L.lvo_pow_det.argtypes = [vp, vp, C.c_uint64, vp]
    L.lvo_prebaked_ao_lookup_batch.argtypes = [vp, vp, u32, u32, u32, vp, vp, C.c_uint64, vp]
    L.lvo_num_threads.restype = i32
    L.lvo_num_threads.argtypes = []
    L.lvo_tea.restype = u32
    L.lvo_tea.argtypes = [u32, u32]
    L.lvo_lcg.restype = u32
    L.lvo_lcg.argtypes = [C.POINTER(u32)]
    L.lvo_rnd.restype = f32
    L.lvo_rnd.argtypes = [C.POINTER(u32)]
    L.lvo_sincos_2pi.argtypes = [f32, C.POINTER(f32), C.POINTER(f32)]
    L.lvo_sincos_rad.argtypes = [f32, C.POINTER(f32), C.POINTER(f32)]
    L.lvo_atan2_det.restype = f32
    L.lvo_atan2_det.argtypes = [f32, f32]
    L.lvo_normalize_positions.argtypes = [vp, C.c_uint64]
    L.lvo_set_helicity_source.argtypes = [vp, f32]
    L.lvo_build_tube_aabb_render_data.argtypes = [vp, vp, vp, u32, f32, vp, C.POINTER(u32), vp, vp, C.POINTER(u32)]
    L.lvo_build_tube_aabb_render_data_ribbons.argtypes = [vp, vp, vp, u32, f32, vp, vp, C.POINTER(u32), vp, vp, C.POINTER(u32)]
    L.lvo_trace_rays_elliptic.argtypes = [vp, f32, f32, vp, i32, vp, vp, f32, f32, u32, vp, vp]
    L.lvo_mat4_inverse.argtypes = [vp, vp]
    L.lvo_scene_create.restype = vp
    L.lvo_scene_create.argtypes = [vp, u32, vp, u32]
    L.lvo_scene_destroy.argtypes = [vp]
    L.lvo_scene_set_tf.argtypes = [vp, vp, u32]
    L.lvo_scene_build_bvh.argtypes = [vp, f32]
    L.lvo_set_num_threads.argtypes = [i32]
    L.lvo_shade_normalize_out_of_range.argtypes = [i32]
    L.lvo_shade_normalize_out_of_range.restype = C.c_uint64
    L.lvo_trace_rays.argtypes = [vp, f32, i32, i32, vp, vp, f32, f32, u32, vp, vp, vp]
    L.lvo_intersect_capsule.restype = i32
    L.lvo_intersect_capsule.argtypes = [vp, vp, vp, vp, f32, i32, C.POINTER(f32), C.POINTER(i32)]
    L.lvo_intersect_capsule_literal.restype = i32
    L.lvo_intersect_capsule_literal.argtypes = [vp, vp, vp, vp, f32, i32, C.POINTER(f32), C.POINTER(i32)]
    L.lvo_compute_depth_range.argtypes = [vp, C.POINTER(Params), vp]
    L.lvo_render_ao.argtypes = [vp, C.POINTER(Params), i32, u32, u32, u32, u32, vp, C.POINTER(Stats)]
    L.lvo_render_rt.argtypes = [vp, C.POINTER(Params), i32, vp, u32, u32, u32, u32, vp, C.POINTER(Stats)]
    L.lvo_pixel_hits.argtypes = [vp, C.POINTER(Params), i32, u32, u32, u32, u32, vp, vp, vp]
    L.lvo_mlat_insert.argtypes = [vp, i32, vp, vp, f32, i32, C.POINTER(i32)]
    L.lvo_render_rt_mlat.argtypes = [vp, C.POINTER(Params), i32, vp, u32, u32, u32, u32, u32, vp, vp, vp, vp, vp,
                                     C.POINTER(C.c_uint64), C.POINTER(Stats)]
    L.lvo_render_rt_mlat_tri.argtypes = [vp, vp, C.POINTER(Params), vp, u32, u32, u32, u32, u32, vp, vp, vp, vp, vp,
                                         C.POINTER(C.c_uint64), C.POINTER(Stats)]
    L.lvo_render_rt_accumulate.argtypes = [vp, C.POINTER(Params), i32, vp, u32, u32, u32, u32, vp, vp, C.POINTER(Stats)]
    L.lvo_ppll_addr.restype = u32
    L.lvo_ppll_addr.argtypes = [u32, u32, u32, u32, u32]
    L.lvo_ppll_gather.argtypes = [vp, C.POINTER(Params), i32, vp, u32, u32, u32, u32, vp, vp, C.POINTER(u32),
                                  C.POINTER(Stats)]
    L.lvo_ppll_resolve.argtypes = [C.POINTER(Params), vp, vp, i32, u32, u32, u32, u32, vp]
    L.lvo_render_ppll.argtypes = [vp, C.POINTER(Params), i32, vp, u32, u32, u32, u32, vp, C.POINTER(Stats)]
    L.lvo_prism_ring_vertices.argtypes = [vp, C.c_uint64, u32, f32, vp, vp]
    L.lvo_prism_fragments.argtypes = [vp, C.POINTER(Params), i32, vp, u32, u32, u32, u32] + [vp] * 11
    u64p = C.POINTER(C.c_uint64)
    L.lvo_build_tube_triangle_render_data.argtypes = [vp, vp, vp, u32, f32, u32, vp, u64p, vp, u64p, vp, u64p]
    L.lvo_build_tube_triangle_render_data_ribbons.argtypes = [vp, vp, vp, u32, vp, f32, f32, u32, vp, u64p, vp, u64p, vp, u64p]
    L.lvo_tri_scene_create.restype = vp
    L.lvo_tri_scene_create.argtypes = [vp, u32, vp, u32, vp, u32, f32]
    L.lvo_tri_scene_destroy.argtypes = [vp]
    L.lvo_tri_scene_build_bvh.argtypes = [vp]
    L.lvo_intersect_triangle.restype = i32
    L.lvo_intersect_triangle.argtypes = [vp, vp, vp, vp, vp, f32, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32)]
    L.lvo_trace_rays_tri.argtypes = [vp, i32, vp, vp, f32, f32, u32, vp, vp, vp]
    L.lvo_render_ao_tri.argtypes = [vp, C.POINTER(Params), i32, u32, u32, u32, u32, vp, C.POINTER(Stats)]
    L.lvo_render_rt_tri.argtypes = [vp, vp, C.POINTER(Params), i32, vp, u32, u32, u32, u32, vp, C.POINTER(Stats)]
    L.lvo_ao_parametrization.argtypes = [vp, vp, u32, f32, vp, vp, u64p]
    L.lvo_bake_ao.argtypes = [vp, vp, f32, i32, i32, vp, u32, u32, u32, u32, f32, i32, vp]
    L.lvo_set_bake_bands.argtypes = [i32, f32, f32]
    L.lvo_set_prism_ring_bands.argtypes = [i32, f32]
    L.lvo_set_twist_line_texture.argtypes = [vp, u32, u32, u32]
    L.lvo_twist_line_sample.argtypes = [vp, vp, vp, i32, C.c_uint64, vp]
    L.lvo_render_rt_prebaked.argtypes = [vp, vp, C.POINTER(Params), i32, vp, vp, u32, u32, u32, u32, u32, u32, u32, vp,
                                         C.POINTER(Stats)]
    L.lvo_generate_abc_flow.argtypes = [vp, i32, i32, i32, f32, f32, f32, f32]
    L.lvo_max_vector_magnitude.restype = f32
    L.lvo_max_vector_magnitude.argtypes = [vp, C.c_uint64]
    L.lvo_trace_streamlines.restype = vp
    L.lvo_trace_streamlines_max_helicity_first.restype = vp
    L.lvo_trace_streamribbons_max_helicity_first.restype = vp
    L.lvo_trace_streamribbons_max_helicity_first.argtypes = [vp, i32, i32, i32, f32, f32, f32, vp, u32, vp, C.POINTER(StreamlineSettings),
                                                             f32, u32, f32, i32, i32, f32, vp]
    L.lvo_trace_streamlines_max_helicity_first.argtypes = [vp, i32, i32, i32, f32, f32, f32, vp, u32, vp, C.POINTER(StreamlineSettings),
                                                           f32, u32, f32, i32]
    L.lvo_set_streamribbon_termination_check_type.argtypes = [u32]
    L.lvo_trace_streamlines_max_helicity_first_ex.restype = vp
    L.lvo_trace_streamlines_max_helicity_first_ex.argtypes = [vp, i32, i32, i32, f32, f32, f32, vp, u32, vp, C.POINTER(StreamlineSettings),
                                                              f32, u32, f32, i32, u32]
    L.lvo_trace_streamlines.argtypes = [vp, i32, i32, i32, f32, f32, f32, vp, u32, vp, u32, C.POINTER(StreamlineSettings)]
    L.lvo_streamlines_sizes.argtypes = [vp, u64p, u64p]
    L.lvo_streamlines_copy.argtypes = [vp, vp, vp, vp]
    L.lvo_streamlines_destroy.argtypes = [vp]
    L.lvo_trace_streamribbons.restype = vp
    L.lvo_trace_streamribbons.argtypes = [vp, i32, i32, i32, f32, f32, f32, vp, u32, vp, u32, vp, u32, i32, f32, vp]
    L.lvo_streamlines_copy_ribbons.argtypes = [vp, vp]
    L.lvo_compute_vector_magnitude_field.argtypes = [vp, vp, i32, i32, i32]
    L.lvo_compute_vorticity_field.argtypes = [vp, vp, i32, i32, i32, f32, f32, f32]
    L.lvo_compute_helicity_field_normalized.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def mlat_insert(nodes, depth2, color, depth, miss=False):
    """One insertNodeMlat() on nodes (K, 6) float32 {premultiplied rgba, transmittance, depth}; returns
    (nodes, depth2, accepted)."""
    n = np.ascontiguousarray(nodes, dtype=np.float32).copy()
    d2 = np.array([depth2], dtype=np.float32)
    col = np.ascontiguousarray(color, dtype=np.float32)
    acc = C.c_int32(0)
    lib().lvo_mlat_insert(_p(n), int(n.shape[0]), _p(d2), _p(col), float(depth), int(miss), C.byref(acc))
    return n, float(d2[0]), bool(acc.value)


def tea(a, b):
    return int(lib().lvo_tea(a & 0xFFFFFFFF, b & 0xFFFFFFFF))


def rnd_sequence(seed, n):
    s = C.c_uint32(seed)
    return np.array([lib().lvo_rnd(C.byref(s)) for _ in range(n)], dtype=np.float32)


def sincos_2pi(xi):
    s, c = C.c_float(), C.c_float()
    lib().lvo_sincos_2pi(C.c_float(xi), C.byref(s), C.byref(c))
    return s.value, c.value


def mat4_inverse(m):
    m = np.ascontiguousarray(m, dtype=np.float32).reshape(16)
    out = np.empty(16, dtype=np.float32)
    lib().lvo_mat4_inverse(_p(m), _p(out))
    return out


def normalize_positions(positions):
    p = np.ascontiguousarray(positions, dtype=np.float32).copy()
    lib().lvo_normalize_positions(_p(p), p.shape[0])
    return p


class _HelicitySource:
    """useRotatingHelicityBands of the render-data builders: lineRotation from a per-point helicity attribute
    (LineDataFlow.cpp:2188-2196, 2014-2027); max_helicity = max |helicity| over the data set (:543-548)."""

    def __init__(self, helicities, max_helicity):
        self.h = None if helicities is None else np.ascontiguousarray(helicities, dtype=np.float32)
        self.m = max_helicity

    def __enter__(self):
        if self.h is not None:
            m = float(np.abs(self.h).max()) if self.m is None else float(self.m)
            lib().lvo_set_helicity_source(_p(self.h), m)

    def __exit__(self, *a):
        if self.h is not None:
            lib().lvo_set_helicity_source(None, 1.0)


def build_tube_aabb_render_data(positions, attributes, line_offsets, line_width, helicities=None, max_helicity=None):
    """a2 (LineDataFlow.cpp:2112-2277): returns (points[48B], seg_indices[S,2], aabbs[S,6]).  helicities: per-point helicity
    attribute -> lineRotation (useRotatingHelicityBands)."""
    pos = np.ascontiguousarray(positions, dtype=np.float32)
    att = np.ascontiguousarray(attributes, dtype=np.float32)
    off = np.ascontiguousarray(line_offsets, dtype=np.uint32)
    n = pos.shape[0]
    pts = np.zeros(max(n, 1), dtype=LINE_POINT_DTYPE)
    seg = np.zeros((max(n, 1), 2), dtype=np.uint32)
    aabb = np.zeros((max(n, 1), 6), dtype=np.float32)
    npts, nseg = C.c_uint32(), C.c_uint32()
    with _HelicitySource(helicities, max_helicity):
        lib().lvo_build_tube_aabb_render_data(_p(pos), _p(att), _p(off), len(off) - 1, line_width, _p(pts),
                                              C.byref(npts), _p(seg), _p(aabb), C.byref(nseg))
    return pts[:npts.value].copy(), seg[:nseg.value].copy(), aabb[:nseg.value].copy()


def build_tube_aabb_render_data_ribbons(positions, attributes, line_offsets, band_width, ribbon_directions):
    """a2 with band data (getLinePassTubeAabbRenderData(false, ellipticTubes=true)): normals = cross(ribbon direction, tangent),
    boxes padded by band_width / 2."""
    pos = np.ascontiguousarray(positions, dtype=np.float32)
    att = np.ascontiguousarray(attributes, dtype=np.float32)
    off = np.ascontiguousarray(line_offsets, dtype=np.uint32)
    rib = np.ascontiguousarray(ribbon_directions, dtype=np.float32)
    assert rib.shape == pos.shape
    n = pos.shape[0]
    pts = np.zeros(max(n, 1), dtype=LINE_POINT_DTYPE)
    seg = np.zeros((max(n, 1), 2), dtype=np.uint32)
    aabb = np.zeros((max(n, 1), 6), dtype=np.float32)
    npts, nseg = C.c_uint32(), C.c_uint32()
    lib().lvo_build_tube_aabb_render_data_ribbons(_p(pos), _p(att), _p(off), len(off) - 1, band_width, _p(rib), _p(pts),
                                                  C.byref(npts), _p(seg), _p(aabb), C.byref(nseg))
    return pts[:npts.value].copy(), seg[:nseg.value].copy(), aabb[:nseg.value].copy()


def build_tube_triangle_render_data(positions, attributes, line_offsets, line_width, num_subdivisions=6, helicities=None,
                                    max_helicity=None):
    """a14 (CappedTriangleTubesCPU.cpp:214-383 + LineDataFlow.cpp:1912-2110): returns
    (triangle_indices[T,3], vertices[32B], line_points[48B]).  helicities: per-point helicity attribute -> lineRotation."""
    pos = np.ascontiguousarray(positions, dtype=np.float32)
    att = np.ascontiguousarray(attributes, dtype=np.float32)
    off = np.ascontiguousarray(line_offsets, dtype=np.uint32)
    ni, nv, npt = C.c_uint64(), C.c_uint64(), C.c_uint64()
    args = (_p(pos), _p(att), _p(off), len(off) - 1, line_width, int(num_subdivisions))
    lib().lvo_build_tube_triangle_render_data(*args, None, C.byref(ni), None, C.byref(nv), None, C.byref(npt))
    idx = np.zeros(max(ni.value, 1), dtype=np.uint32)
    verts = np.zeros(max(nv.value, 1), dtype=TUBE_VERTEX_DTYPE)
    pts = np.zeros(max(npt.value, 1), dtype=LINE_POINT_DTYPE)
    with _HelicitySource(helicities, max_helicity):
        lib().lvo_build_tube_triangle_render_data(*args, _p(idx), C.byref(ni), _p(verts), C.byref(nv), _p(pts), C.byref(npt))
    return idx[:ni.value].reshape(-1, 3).copy(), verts[:nv.value].copy(), pts[:npt.value].copy()


def build_tube_triangle_render_data_ribbons(positions, attributes, line_offsets, ribbon_directions, band_width,
                                            min_band_thickness=0.15, num_subdivisions=8):
    """The elliptic triangle tubes of a band data set (createCappedTriangleEllipticTubesRenderDataCPU): returns
    (triangle_indices[T,3], vertices[32B], line_points[48B])."""
    pos = np.ascontiguousarray(positions, dtype=np.float32)
    att = np.ascontiguousarray(attributes, dtype=np.float32)
    off = np.ascontiguousarray(line_offsets, dtype=np.uint32)
    rib = np.ascontiguousarray(ribbon_directions, dtype=np.float32)
    ni, nv, npt = C.c_uint64(), C.c_uint64(), C.c_uint64()
    args = (_p(pos), _p(att), _p(off), len(off) - 1, _p(rib), float(band_width), float(min_band_thickness), int(num_subdivisions))
    lib().lvo_build_tube_triangle_render_data_ribbons(*args, None, C.byref(ni), None, C.byref(nv), None, C.byref(npt))
    idx = np.zeros(max(ni.value, 1), dtype=np.uint32)
    verts = np.zeros(max(nv.value, 1), dtype=TUBE_VERTEX_DTYPE)
    pts = np.zeros(max(npt.value, 1), dtype=LINE_POINT_DTYPE)
    lib().lvo_build_tube_triangle_render_data_ribbons(*args, _p(idx), C.byref(ni), _p(verts), C.byref(nv), _p(pts), C.byref(npt))
    return idx[:ni.value].reshape(-1, 3).copy(), verts[:nv.value].copy(), pts[:npt.value].copy()


def generate_abc_flow(xs, ys, zs, A=float(np.sqrt(np.float32(3.0))), B=float(np.sqrt(np.float32(2.0))), Cc=1.0,
                      res_scale=6.0):
    """AbcFlowGenerator::generateAbcFlow (Loader/AbcFlowGenerator.cpp:41-72): [zs, ys, xs, 3] float32."""
    v = np.empty((zs, ys, xs, 3), dtype=np.float32)
    lib().lvo_generate_abc_flow(_p(v), xs, ys, zs, A, B, Cc, res_scale)
    return v


def trace_streamlines(vector_field, spacing, scalar_fields, seeds, settings):
    """StreamlineTracingGrid::traceStreamlines restated: vector_field [zs, ys, xs, 3], scalar_fields list of [zs, ys, xs].
    Returns (positions [P,3], attributes [k,P], line_offsets [L+1])."""
    v = np.ascontiguousarray(vector_field, dtype=np.float32)
    zs, ys, xs = v.shape[:3]
    sf = [np.ascontiguousarray(f, dtype=np.float32) for f in scalar_fields]
    ptrs = (C.c_void_p * max(len(sf), 1))(*[f.ctypes.data for f in sf])
    sd = np.ascontiguousarray(seeds, dtype=np.float32).reshape(-1, 3)
    h = lib().lvo_trace_streamlines(_p(v), xs, ys, zs, spacing[0], spacing[1], spacing[2], ptrs, len(sf), _p(sd), len(sd),
                                    C.byref(settings))
    nl, npt = C.c_uint64(), C.c_uint64()
    lib().lvo_streamlines_sizes(h, C.byref(nl), C.byref(npt))
    pos = np.zeros((npt.value, 3), dtype=np.float32)
    att = np.zeros((len(sf), npt.value), dtype=np.float32)
    off = np.zeros(nl.value + 1, dtype=np.uint32)
    lib().lvo_streamlines_copy(h, _p(pos), _p(att), _p(off))
    lib().lvo_streamlines_destroy(h)
    return pos, att, off


def prism_coverage_dir(P, x, y):
    """(coverage direction of pixel (x, y), the ray generator's normalised direction): prismCoverageDir / primaryRay of the checker."""
    cov, ray = np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib().lvo_prism_coverage_dir(C.byref(P), int(x), int(y), _p(cov), _p(ray))
    return cov, ray


def trace_streamlines_max_helicity_first(vector_field, spacing, scalar_fields, helicity_field, settings, minimum_separation_distance=0.08,
                                         loop_check_mode=1, termination_distance_self=1.0, seeding_subsampling_factor=1, ribbons=None,
                                         termination_check_type=1):
    """StreamlineMaxHelicityFirstSeeder + _traceStreamribbonsDecreasingHelicity, sequential: (positions, attributes, line_offsets).
    termination_check_type: 0 naive, 1 grid-based, 2 k-d tree-based, 3 hashed grid-based (TerminationCheckType)."""
    v = np.ascontiguousarray(vector_field, dtype=np.float32)
    zs, ys, xs = v.shape[:3]
    sf = [np.ascontiguousarray(f, dtype=np.float32) for f in scalar_fields]
    ptrs = (C.c_void_p * max(len(sf), 1))(*[f.ctypes.data for f in sf])
    hf = np.ascontiguousarray(helicity_field, dtype=np.float32)
    if ribbons is None:
        h = lib().lvo_trace_streamlines_max_helicity_first_ex(_p(v), xs, ys, zs, spacing[0], spacing[1], spacing[2], ptrs, len(sf), _p(hf),
                                                              C.byref(settings), float(minimum_separation_distance), int(loop_check_mode),
                                                              float(termination_distance_self), int(seeding_subsampling_factor),
                                                              int(termination_check_type))
    else:   # ribbons = dict(use_helicity=, max_helicity_twist=, initial_ribbon_direction=): the STREAMRIBBONS form, + ribbon directions
        lib().lvo_set_streamribbon_termination_check_type(int(termination_check_type))
        ird = np.ascontiguousarray(ribbons.get("initial_ribbon_direction", (0.0, 1.0, 0.0)), dtype=np.float32)
        h = lib().lvo_trace_streamribbons_max_helicity_first(_p(v), xs, ys, zs, spacing[0], spacing[1], spacing[2], ptrs, len(sf), _p(hf),
                                                             C.byref(settings), float(minimum_separation_distance), int(loop_check_mode),
                                                             float(termination_distance_self), int(seeding_subsampling_factor),
                                                             int(ribbons.get("use_helicity", True)), float(ribbons.get("max_helicity_twist", 0.25)),
                                                             _p(ird))
        lib().lvo_set_streamribbon_termination_check_type(1)
    nl, npt = C.c_uint64(), C.c_uint64()
    lib().lvo_streamlines_sizes(h, C.byref(nl), C.byref(npt))
    pos = np.zeros((npt.value, 3), dtype=np.float32)
    att = np.zeros((len(sf), npt.value), dtype=np.float32)
    off = np.zeros(nl.value + 1, dtype=np.uint32)
    lib().lvo_streamlines_copy(h, _p(pos), _p(att), _p(off))
    rib = None
    if ribbons is not None:
        rib = np.zeros((npt.value, 3), dtype=np.float32)
        lib().lvo_streamlines_copy_ribbons(h, _p(rib))
    lib().lvo_streamlines_destroy(h)
    return (pos, att, off) if ribbons is None else (pos, att, off, rib)


def trace_streamribbons(vector_field, spacing, scalar_fields, seeds, settings, helicity_index, use_helicity=True,
                        max_helicity_twist=0.25, initial_ribbon_direction=(0.0, 1.0, 0.0)):
    """traceStreamribbons restated: trace_streamlines + (ribbon_directions [P,3])."""
    v = np.ascontiguousarray(vector_field, dtype=np.float32)
    zs, ys, xs = v.shape[:3]
    sf = [np.ascontiguousarray(f, dtype=np.float32) for f in scalar_fields]
    ptrs = (C.c_void_p * max(len(sf), 1))(*[f.ctypes.data for f in sf])
    sd = np.ascontiguousarray(seeds, dtype=np.float32).reshape(-1, 3)
    ird = np.ascontiguousarray(initial_ribbon_direction, dtype=np.float32)
    h = lib().lvo_trace_streamribbons(_p(v), xs, ys, zs, spacing[0], spacing[1], spacing[2], ptrs, len(sf), _p(sd), len(sd),
                                      C.byref(settings), int(helicity_index), int(use_helicity), float(max_helicity_twist), _p(ird))
    nl, npt = C.c_uint64(), C.c_uint64()
    lib().lvo_streamlines_sizes(h, C.byref(nl), C.byref(npt))
    pos = np.zeros((npt.value, 3), dtype=np.float32)
    att = np.zeros((len(sf), npt.value), dtype=np.float32)
    off = np.zeros(nl.value + 1, dtype=np.uint32)
    rib = np.zeros((npt.value, 3), dtype=np.float32)
    lib().lvo_streamlines_copy(h, _p(pos), _p(att), _p(off))
    lib().lvo_streamlines_copy_ribbons(h, _p(rib))
    lib().lvo_streamlines_destroy(h)
    return pos, att, off, rib


def vorticity_field(vector_field, spacing):
    """computeVorticityField, GridLoader.cpp:64-112 (vector_field [zs, ys, xs, 3])."""
    v = np.ascontiguousarray(vector_field, dtype=np.float32)
    zs, ys, xs = v.shape[:3]
    out = np.zeros_like(v)
    lib().lvo_compute_vorticity_field(_p(v), _p(out), xs, ys, zs, spacing[0], spacing[1], spacing[2])
    return out


def vector_magnitude_field(vector_field):
    v = np.ascontiguousarray(vector_field, dtype=np.float32)
    zs, ys, xs = v.shape[:3]
    out = np.zeros((zs, ys, xs), dtype=np.float32)
    lib().lvo_compute_vector_magnitude_field(_p(v), _p(out), xs, ys, zs)
    return out


def helicity_field(velocity, vorticity, normalize_velocity=False, normalize_vorticity=False):
    """computeHelicityFieldNormalized, GridLoader.cpp:140-183."""
    v = np.ascontiguousarray(velocity, dtype=np.float32)
    w = np.ascontiguousarray(vorticity, dtype=np.float32)
    zs, ys, xs = v.shape[:3]
    out = np.zeros((zs, ys, xs), dtype=np.float32)
    lib().lvo_compute_helicity_field_normalized(_p(v), _p(w), _p(out), xs, ys, zs, int(normalize_velocity), int(normalize_vorticity))
    return out


def intersect_triangle(o, d, v0, v1, v2, pad):
    a = [np.ascontiguousarray(v, dtype=np.float32) for v in (o, d, v0, v1, v2)]
    t, u, v = C.c_float(), C.c_float(), C.c_float()
    hit = lib().lvo_intersect_triangle(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), pad, C.byref(t), C.byref(u),
                                       C.byref(v))
    return bool(hit), t.value, u.value, v.value


def intersect_capsule(o, d, p0, p1, radius, capped=True, literal=False):
    a = [np.ascontiguousarray(v, dtype=np.float32) for v in (o, d, p0, p1)]
    t, k = C.c_float(), C.c_int()
    fn = lib().lvo_intersect_capsule_literal if literal else lib().lvo_intersect_capsule
    hit = fn(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), radius, int(capped), C.byref(t),
                                      C.byref(k))
    return bool(hit), t.value, k.value


DEFAULTS = dict(
    fovY=float(np.float32(2.0 * np.arctan(0.5))), nearDist=0.01, farDist=100.0,
    background=(1.0, 1.0, 1.0, 1.0), lineWidth=0.002,
    maxDepthComplexity=1024, numSamplesPerFrame=1, frameNumber=0, useJitteredRays=0, useDeterministicSampling=0,
    useCappedTubes=1, useHalos=1, useDepthCues=0, useAmbientOcclusion=0,
    depthCueStrength=0.8, minDepth=0.0, maxDepth=1.0, aoStrength=1.0, aoGamma=1.0,
    attrMin=0.0, attrMax=1.0,
    aoSamplesPerFrame=4, aoIterations=1, aoUseDistance=1, aoJitterPrimary=1, tubeNumSubdivisions=6, aoRadius=0.1,
    ppllMaxNumFrags=100, ppllLinkedListSize=0, ppllTileW=2, ppllTileH=8,
    useBands=0, useEllipticTubes=0, bandWidth=0.005, minBandThickness=0.15, minThickness=0.15, lssGeometry=0,
    useHelicityBands=0, numSubdivisionsBands=6, separatorBaseWidth=0.2, helicityRotationFactor=1.0,
    uniformHelicityBandWidth=1, ppllSortingMode=0, ppllFragmentSource=0,
)


def make_params(view, proj, width, height, **kw):
    cfg = dict(DEFAULTS)
    for k in kw:
        if k not in cfg:
            raise KeyError(k)
    cfg.update(kw)
    P = Params()
    P.view = (C.c_float * 16)(*np.asarray(view, dtype=np.float32).reshape(16))
    P.proj = (C.c_float * 16)(*np.asarray(proj, dtype=np.float32).reshape(16))
    P.width, P.height = int(width), int(height)
    for k, v in cfg.items():
        if k == "background":
            P.background = (C.c_float * 4)(*v)
        else:
            setattr(P, k, v)
    if P.ppllLinkedListSize == 0:
        pw = -(-P.width // P.ppllTileW) * P.ppllTileW
        ph = -(-P.height // P.ppllTileH) * P.ppllTileH
        P.ppllLinkedListSize = 20 * pw * ph  # expectedAvgDepthComplexity = 20, PerPixelLinkedListLineRenderer.hpp:45-49
    return P


class Scene:
    def __init__(self, points, seg_indices, tf_rgba):
        self.points = np.ascontiguousarray(points, dtype=LINE_POINT_DTYPE)
        self.seg = np.ascontiguousarray(seg_indices, dtype=np.uint32).reshape(-1, 2)
        self.h = lib().lvo_scene_create(_p(self.points), len(self.points), _p(self.seg), len(self.seg))
        tf = np.ascontiguousarray(tf_rgba, dtype=np.float32).reshape(-1, 4)
        lib().lvo_scene_set_tf(self.h, _p(tf), tf.shape[0])
        self.bvh_line_width = None

    def __del__(self):
        try:
            if self.h:
                lib().lvo_scene_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def build_bvh(self, line_width):
        lib().lvo_scene_build_bvh(self.h, line_width)
        self.bvh_line_width = line_width

    def _use_bvh(self, P_or_lw, use_bvh):
        lw = P_or_lw
        if isinstance(P_or_lw, Params):     # elliptic tubes: boxes of half the band width, LineDataFlow.cpp:2120-2126
            lw = P_or_lw.bandWidth if P_or_lw.useEllipticTubes else P_or_lw.lineWidth
        if use_bvh and self.bvh_line_width != lw:
            self.build_bvh(lw)
        return int(bool(use_bvh))

    def trace_rays(self, origins, dirs, t_min, t_max, line_width, capped=True, use_bvh=False):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        t = np.empty(n, dtype=np.float32)
        seg = np.empty(n, dtype=np.uint32)
        kind = np.empty(n, dtype=np.uint32)
        ub = self._use_bvh(line_width, use_bvh)
        lib().lvo_trace_rays(self.h, line_width, int(capped), ub, _p(o), _p(d), t_min, t_max, n, _p(t), _p(seg),
                             _p(kind))
        return t, seg, kind

    def trace_rays_elliptic(self, origins, dirs, t_min, t_max, band_width, min_band_thickness, camera_position, use_bvh=False):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        cam = np.ascontiguousarray(camera_position, dtype=np.float32).reshape(3)
        n = o.shape[0]
        t = np.empty(n, dtype=np.float32)
        seg = np.empty(n, dtype=np.uint32)
        ub = self._use_bvh(band_width, use_bvh)
        lib().lvo_trace_rays_elliptic(self.h, band_width, min_band_thickness, _p(cam), ub, _p(o), _p(d), t_min, t_max, n,
                                      _p(t), _p(seg))
        return t, seg

    def depth_range(self, P):
        out = np.empty(2, dtype=np.float32)
        lib().lvo_compute_depth_range(self.h, C.byref(P), _p(out))
        return out

    def _tile(self, P, tile):
        return tile if tile is not None else (0, 0, P.width, P.height)

    def render_ao(self, P, tile=None, use_bvh=False, stats=None):
        x0, y0, w, h = self._tile(P, tile)
        ao = np.ones((P.height, P.width), dtype=np.float32)
        st = stats if stats is not None else Stats()
        lib().lvo_render_ao(self.h, C.byref(P), self._use_bvh(P, use_bvh), x0, y0, w, h, _p(ao), C.byref(st))
        return ao

    def render_rt(self, P, ao=None, tile=None, use_bvh=False, stats=None, prev=None):
        x0, y0, w, h = self._tile(P, tile)
        out = np.empty((h, w, 4), dtype=np.uint8)
        st = stats if stats is not None else Stats()
        aop = _p(np.ascontiguousarray(ao, dtype=np.float32)) if ao is not None else None
        if P.useAmbientOcclusion and ao is None:
            raise ValueError("useAmbientOcclusion needs an ao buffer")
        if prev is not None:   # multi-frame accumulation: frame P.frameNumber mixed with the previous frame's tile
            pv = np.ascontiguousarray(prev, dtype=np.uint8)
            assert pv.shape == (h, w, 4)
            lib().lvo_render_rt_accumulate(self.h, C.byref(P), self._use_bvh(P, use_bvh), aop, x0, y0, w, h, _p(pv), _p(out),
                                           C.byref(st))
            return out
        lib().lvo_render_rt(self.h, C.byref(P), self._use_bvh(P, use_bvh), aop, x0, y0, w, h, _p(out), C.byref(st))
        return out

    def compute_fragment_color(self, P, frag_pos, normal, tangent, is_cap, attribute, ao_texel):
        """Test hook: the restatement of computeFragmentColor on n independent inputs -> (colours n x 4, payload.hitT n)."""
        fp = np.ascontiguousarray(frag_pos, dtype=np.float32).reshape(-1, 3)
        n = fp.shape[0]
        nr = np.ascontiguousarray(normal, dtype=np.float32).reshape(n, 3)
        tg = np.ascontiguousarray(tangent, dtype=np.float32).reshape(n, 3)
        cap = np.ascontiguousarray(is_cap, dtype=np.uint32).reshape(n)
        at = np.ascontiguousarray(attribute, dtype=np.float32).reshape(n)
        ao = np.ascontiguousarray(ao_texel, dtype=np.float32).reshape(n)
        col = np.empty((n, 4), dtype=np.float32)
        ht = np.empty(n, dtype=np.float32)
        lib().lvo_compute_fragment_color_batch(self.h, C.byref(P), n, _p(fp), _p(nr), _p(tg), _p(cap), _p(at), _p(ao), _p(col),
                                               _p(ht))
        return col, ht

    def compute_fragment_color_raster(self, P, frag_pos, normal, tangent, is_cap, attribute, ao_texel, eps_white):
        """Test hook: the raster variant of computeFragmentColor (EPSILON_OUTLINE = 0, EPSILON_WHITE = eps_white[i], cap halo
        min(rp, |rp2|); LinePassGeometryShaderTubes.glsl:785-815,1079-1087) on n independent inputs."""
        fp = np.ascontiguousarray(frag_pos, dtype=np.float32).reshape(-1, 3)
        n = fp.shape[0]
        nr = np.ascontiguousarray(normal, dtype=np.float32).reshape(n, 3)
        tg = np.ascontiguousarray(tangent, dtype=np.float32).reshape(n, 3)
        cap = np.ascontiguousarray(is_cap, dtype=np.uint32).reshape(n)
        at = np.ascontiguousarray(attribute, dtype=np.float32).reshape(n)
        ao = np.ascontiguousarray(ao_texel, dtype=np.float32).reshape(n)
        ew = np.ascontiguousarray(eps_white, dtype=np.float32).reshape(n)
        col = np.empty((n, 4), dtype=np.float32)
        ht = np.empty(n, dtype=np.float32)
        lib().lvo_compute_fragment_color_raster_batch(self.h, C.byref(P), n, _p(fp), _p(nr), _p(tg), _p(cap), _p(at), _p(ao), _p(ew),
                                                      _p(col), _p(ht))
        return col, ht

    def pixel_hits(self, P, tile=None, use_bvh=False):
        """All capsule entry hits of every pixel-centre ray: (offsets (w*h+1), segments, t), ascending segment index."""
        x0, y0, w, h = self._tile(P, tile)
        offs = np.zeros(w * h + 1, dtype=np.uint64)
        ub = self._use_bvh(P, use_bvh)
        lib().lvo_pixel_hits(self.h, C.byref(P), ub, x0, y0, w, h, _p(offs), None, None)
        segs = np.zeros(max(int(offs[-1]), 1), dtype=np.uint32)
        ts = np.zeros(max(int(offs[-1]), 1), dtype=np.float32)
        lib().lvo_pixel_hits(self.h, C.byref(P), ub, x0, y0, w, h, _p(offs), _p(segs), _p(ts))
        return offs, segs[:int(offs[-1])], ts[:int(offs[-1])]

    def render_rt_mlat(self, P, num_nodes, ao=None, tile=None, use_bvh=False, trace=None, stats=None, tri_scene=None):
        """USE_MLAT frame.  trace = None: candidates visited in ascending segment order.  trace = (n, 4) uint32 records
        {viewport pixel index y * W + x, sequence number, segment, flag} (any order): replayed per pixel in sequence
        order and validated.  Returns (rgba8 tile, node states (h, w, num_nodes * 6 + 1), violations)."""
        x0, y0, w, h = self._tile(P, tile)
        out = np.empty((h, w, 4), dtype=np.uint8)
        nodes = np.zeros((h, w, num_nodes * 6 + 1), dtype=np.float32)
        st = stats if stats is not None else Stats()
        aop = _p(np.ascontiguousarray(ao, dtype=np.float32)) if ao is not None else None
        if P.useAmbientOcclusion and ao is None:
            raise ValueError("useAmbientOcclusion needs an ao buffer")
        viol = C.c_uint64(0)
        offs = segs = flags = None
        if trace is not None:
            tr = np.asarray(trace, dtype=np.uint32).reshape(-1, 4)
            px, py = tr[:, 0] % P.width, tr[:, 0] // P.width
            inside = (px >= x0) & (px < x0 + w) & (py >= y0) & (py < y0 + h)
            tr, px, py = tr[inside], px[inside], py[inside]
            local = (py - y0).astype(np.int64) * w + (px - x0)
            order = np.lexsort((tr[:, 1], local))
            tr, local = tr[order], local[order]
            offs = np.zeros(w * h + 1, dtype=np.uint64)
            np.add.at(offs, local + 1, 1)
            offs = np.cumsum(offs).astype(np.uint64)
            segs = np.ascontiguousarray(tr[:, 2], dtype=np.uint32)
            flags = np.ascontiguousarray(tr[:, 3], dtype=np.uint8)
        tr_args = (_p(offs) if offs is not None else None, _p(segs) if segs is not None else None,
                   _p(flags) if flags is not None else None, _p(out), _p(nodes), C.byref(viol), C.byref(st))
        if tri_scene is not None:   # "Triangle Mesh" geometry mode: candidates / trace ids are triangles
            lib().lvo_render_rt_mlat_tri(self.h, tri_scene.h, C.byref(P), aop, x0, y0, w, h, int(num_nodes), *tr_args)
        else:
            lib().lvo_render_rt_mlat(self.h, C.byref(P), self._use_bvh(P, use_bvh), aop, x0, y0, w, h, int(num_nodes), *tr_args)
        return out, nodes, int(viol.value)

    def prism_fragments(self, P, ao=None, tile=None, use_bvh=False):
        """Fragments of the rasterised programmable-pull prism (ppll_fragment_source = raster_prism) of every pixel of the tile, in
        ascending (segment, triangle) order: dict of offsets (w * h + 1) and per-fragment arrays."""
        x0, y0, w, h = self._tile(P, tile)
        offs = np.zeros(w * h + 1, dtype=np.uint64)
        ub = self._use_bvh(P, use_bvh)
        aop = _p(np.ascontiguousarray(ao, dtype=np.float32)) if ao is not None else None
        lib().lvo_prism_fragments(self.h, C.byref(P), ub, aop, x0, y0, w, h, _p(offs), *([None] * 10))
        n = int(offs[-1])
        m = max(n, 1)
        r = dict(seg=np.zeros(m, np.uint32), tri=np.zeros(m, np.uint32), weights=np.zeros((m, 3), np.float32),
                 depth=np.zeros(m, np.float32), pos=np.zeros((m, 3), np.float32), normal=np.zeros((m, 3), np.float32),
                 tangent=np.zeros((m, 3), np.float32), attr=np.zeros(m, np.float32), colour=np.zeros(m, np.uint32),
                 rgba=np.zeros((m, 4), np.float32))
        lib().lvo_prism_fragments(self.h, C.byref(P), ub, aop, x0, y0, w, h, _p(offs), _p(r["seg"]), _p(r["tri"]), _p(r["weights"]),
                                  _p(r["depth"]), _p(r["pos"]), _p(r["normal"]), _p(r["tangent"]), _p(r["attr"]), _p(r["colour"]),
                                  _p(r["rgba"]))
        r = {k: v[:n] for k, v in r.items()}
        r["offsets"] = offs
        return r

    def ppll_gather(self, P, ao=None, tile=None, use_bvh=False, stats=None):
        x0, y0, w, h = self._tile(P, tile)
        pw = -(-P.width // P.ppllTileW) * P.ppllTileW
        ph = -(-P.height // P.ppllTileH) * P.ppllTileH
        nodes = np.zeros((P.ppllLinkedListSize, 3), dtype=np.uint32)
        start = np.zeros(pw * ph, dtype=np.uint32)
        cnt = C.c_uint32()
        st = stats if stats is not None else Stats()
        aop = _p(np.ascontiguousarray(ao, dtype=np.float32)) if ao is not None else None
        lib().lvo_ppll_gather(self.h, C.byref(P), self._use_bvh(P, use_bvh), aop, x0, y0, w, h, _p(nodes), _p(start),
                              C.byref(cnt), C.byref(st))
        return nodes, start, cnt.value

    def render_ppll(self, P, ao=None, tile=None, use_bvh=False, stats=None):
        x0, y0, w, h = self._tile(P, tile)
        out = np.empty((h, w, 4), dtype=np.uint8)
        st = stats if stats is not None else Stats()
        aop = _p(np.ascontiguousarray(ao, dtype=np.float32)) if ao is not None else None
        lib().lvo_render_ppll(self.h, C.byref(P), self._use_bvh(P, use_bvh), aop, x0, y0, w, h, _p(out), C.byref(st))
        return out


class TriScene:
    """Triangle-tube mesh (a14 output) + CPU BVH: the reference's RTAO geometry."""

    def __init__(self, indices, vertices, line_points, line_width):
        self.idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        self.verts = np.ascontiguousarray(vertices, dtype=TUBE_VERTEX_DTYPE)
        self.points = np.ascontiguousarray(line_points, dtype=LINE_POINT_DTYPE)
        self.h = lib().lvo_tri_scene_create(_p(self.idx), len(self.idx), _p(self.verts), len(self.verts),
                                            _p(self.points), len(self.points), line_width)
        self.has_bvh = False

    def __del__(self):
        try:
            if self.h:
                lib().lvo_tri_scene_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _use_bvh(self, use_bvh):
        if use_bvh and not self.has_bvh:
            lib().lvo_tri_scene_build_bvh(self.h)
            self.has_bvh = True
        return int(bool(use_bvh))

    def trace_rays(self, origins, dirs, t_min, t_max, use_bvh=False):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        t = np.empty(n, dtype=np.float32)
        tri = np.empty(n, dtype=np.uint32)
        uv = np.empty((n, 2), dtype=np.float32)
        lib().lvo_trace_rays_tri(self.h, self._use_bvh(use_bvh), _p(o), _p(d), t_min, t_max, n, _p(t), _p(tri), _p(uv))
        return t, tri, uv

    def render_rt(self, scene, P, ao=None, tile=None, use_bvh=False, stats=None):
        """Colour pass in "Triangle Mesh" geometry mode; `scene` (a capsule Scene) supplies the transfer function."""
        x0, y0, w, h = tile if tile is not None else (0, 0, P.width, P.height)
        out = np.empty((h, w, 4), dtype=np.uint8)
        st = stats if stats is not None else Stats()
        aop = _p(np.ascontiguousarray(ao, dtype=np.float32)) if ao is not None else None
        if P.useAmbientOcclusion and ao is None:
            raise ValueError("useAmbientOcclusion needs an ao buffer")
        lib().lvo_render_rt_tri(scene.h, self.h, C.byref(P), self._use_bvh(use_bvh), aop, x0, y0, w, h, _p(out), C.byref(st))
        return out

    def render_ao(self, P, tile=None, use_bvh=False, stats=None):
        x0, y0, w, h = tile if tile is not None else (0, 0, P.width, P.height)
        ao = np.ones((P.height, P.width), dtype=np.float32)
        st = stats if stats is not None else Stats()
        lib().lvo_render_ao_tri(self.h, C.byref(P), self._use_bvh(use_bvh), x0, y0, w, h, _p(ao), C.byref(st))
        return ao


def ao_parametrization(positions, line_offsets, expected_param_segment_length=0.001):
    """recomputeStaticParametrization (VulkanAmbientOcclusionBaker.cpp:563-653): (blending_weights[P], sampling_locations[M])."""
    pos = np.ascontiguousarray(positions, dtype=np.float32)
    off = np.ascontiguousarray(line_offsets, dtype=np.uint32)
    n = C.c_uint64()
    lib().lvo_ao_parametrization(_p(pos), _p(off), len(off) - 1, expected_param_segment_length, None, None, C.byref(n))
    bw = np.zeros(len(pos), dtype=np.float32)
    sl = np.zeros(n.value, dtype=np.float32)
    lib().lvo_ao_parametrization(_p(pos), _p(off), len(off) - 1, expected_param_segment_length, _p(bw), _p(sl), C.byref(n))
    return bw, sl


def bake_ao(scene, tri_scene, line_width, sampling_locations, num_tube_subdivisions=8, num_samples=4, num_iterations=128,
            radius=0.1, use_distance=True, capped=True, use_bvh=True, bands=None):
    """VulkanAmbientOcclusionBaker.Compute iterated: factors [num_sampling_locations, num_tube_subdivisions].
    bands = (band_width, min_band_thickness): USE_BANDS ray origins (elliptic cross-section)."""
    lib().lvo_set_bake_bands(int(bands is not None), float(bands[0]) * 0.5 if bands else 0.0, float(bands[1]) if bands else 1.0)
    sl = np.ascontiguousarray(sampling_locations, dtype=np.float32)
    out = np.zeros((len(sl), num_tube_subdivisions), dtype=np.float32)
    if use_bvh:
        scene._use_bvh(line_width, True)
        if tri_scene is not None:
            tri_scene._use_bvh(True)
    lib().lvo_bake_ao(scene.h, tri_scene.h if tri_scene is not None else None, line_width, int(capped), int(use_bvh), _p(sl),
                      len(sl), num_tube_subdivisions, num_samples, num_iterations, radius, int(use_distance), _p(out))
    return out


def render_rt_prebaked(scene, tri_scene, P, factors, blending_weights, tile=None, use_bvh=True, stats=None):
    x0, y0, w, h = tile if tile is not None else (0, 0, P.width, P.height)
    out = np.empty((h, w, 4), dtype=np.uint8)
    st = stats if stats is not None else Stats()
    f = np.ascontiguousarray(factors, dtype=np.float32)
    bw = np.ascontiguousarray(blending_weights, dtype=np.float32)
    ub = scene._use_bvh(P, use_bvh)
    if tri_scene is not None:
        tri_scene._use_bvh(use_bvh)
    lib().lvo_render_rt_prebaked(scene.h, tri_scene.h if tri_scene is not None else None, C.byref(P), ub, _p(f), _p(bw),
                                 len(bw), f.shape[0], f.shape[1], x0, y0, w, h, _p(out), C.byref(st))
    return out


TWIST_FILTER_MODES = ["Nearest", "Linear", "Nearest Mipmap Nearest", "Linear Mipmap Nearest", "Nearest Mipmap Linear", "Linear Mipmap Linear"]


class twist_line_texture:
    """Context manager: USE_HELICITY_BANDS_TEXTURE with the given RGBA8 image (h, w, 4) and filtering mode name."""
    def __init__(self, rgba8, mode="Linear Mipmap Linear"):
        self.img = np.ascontiguousarray(rgba8, dtype=np.uint8)
        self.mode = TWIST_FILTER_MODES.index(mode)

    def __enter__(self):
        lib().lvo_set_twist_line_texture(_p(self.img), self.img.shape[1], self.img.shape[0], self.mode)
        return self

    def __exit__(self, *a):
        lib().lvo_set_twist_line_texture(None, 0, 0, 0)


def twist_line_sample(u, dudx=None, dudy=None):
    """Samples of the currently set twist-line texture at (u, 0.5): without derivatives level 0, else textureGrad."""
    u = np.ascontiguousarray(u, dtype=np.float32)
    grad = dudx is not None
    dx = np.ascontiguousarray(dudx if grad else np.zeros_like(u), dtype=np.float32)
    dy = np.ascontiguousarray(dudy if grad else np.zeros_like(u), dtype=np.float32)
    out = np.empty((len(u), 4), dtype=np.float32)
    lib().lvo_twist_line_sample(_p(u), _p(dx), _p(dy), int(grad), len(u), _p(out))
    return out


def prism_ring_vertices(points, num_subdivisions, line_width, band_thickness=None):
    """Ring vertices of the programmable-pull vertex stage: (positions, normals), each (len(points), N, 3).
    band_thickness: the USE_BANDS ring (line_width = the band width)."""
    lib().lvo_set_prism_ring_bands(int(band_thickness is not None), float(band_thickness or 1.0))
    pts = np.ascontiguousarray(points, dtype=LINE_POINT_DTYPE)
    n = min(max(int(num_subdivisions), 3), 16)
    pos = np.zeros((len(pts), n, 3), dtype=np.float32)
    nrm = np.zeros((len(pts), n, 3), dtype=np.float32)
    lib().lvo_prism_ring_vertices(_p(pts), len(pts), int(num_subdivisions), float(line_width), _p(pos), _p(nrm))
    return pos, nrm


def ppll_resolve(P, nodes, start_offset, tile=None, literal=False):
    x0, y0, w, h = tile if tile is not None else (0, 0, P.width, P.height)
    out = np.empty((h, w, 4), dtype=np.uint8)
    n = np.ascontiguousarray(nodes, dtype=np.uint32)
    s = np.ascontiguousarray(start_offset, dtype=np.uint32)
    lib().lvo_ppll_resolve(C.byref(P), _p(n), _p(s), int(literal), x0, y0, w, h, _p(out))
    return out


def ppll_addr(x, y, padded_w, tile_w, tile_h):
    return int(lib().lvo_ppll_addr(x, y, padded_w, tile_w, tile_h))


def shade_normalize_out_of_range(reset=True):
    """normalize() calls of the shading code outside the clamp range of the build's rule since the last reset (oracle/lv_oracle_common.h)."""
    return int(lib().lvo_shade_normalize_out_of_range(int(bool(reset))))


def set_num_threads(n):
    lib().lvo_set_num_threads(int(n))


def num_threads():
    """Threads the oracle's OpenMP loops run on."""
    return int(lib().lvo_num_threads())


def prebaked_ao_lookup(factors, blending_weights, num_subdivisions, vertex_id, phi):
    """Test hook: the static prebaker's table lookup (AmbientOcclusion.glsl:49-75 without pow / strength) on n inputs."""
    f = np.ascontiguousarray(factors, dtype=np.float32).reshape(-1)
    bw = np.ascontiguousarray(blending_weights, dtype=np.float32)
    v = np.ascontiguousarray(vertex_id, dtype=np.float32)
    p = np.ascontiguousarray(phi, dtype=np.float32)
    out = np.empty(len(v), dtype=np.float32)
    lib().lvo_prebaked_ao_lookup_batch(_p(f), _p(bw), len(bw), len(f) // int(num_subdivisions), int(num_subdivisions), _p(v), _p(p),
                                       len(v), _p(out))
    return out


def pow_det(x, y):
    """The build-owned pow of the shading code (bit-identical to lv_pow_det of the HIP library)."""
    xx = np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=np.float32), np.broadcast(x, y).shape), dtype=np.float32).reshape(-1)
    yy = np.ascontiguousarray(np.broadcast_to(np.asarray(y, dtype=np.float32), np.broadcast(x, y).shape), dtype=np.float32).reshape(-1)
    out = np.empty(len(xx), dtype=np.float32)
    lib().lvo_pow_det(_p(xx), _p(yy), len(xx), _p(out))
    return out


def ribbon_of_rays(cam, dirs, axis_point, axis_dir, radius, cap_hit=None, cap_normal=None):
    """Test hook: the ribbon coordinate of n viewing rays (origin cam) with respect to a tube axis (tubeRibbonOfRay), or -- with
    cap_hit / cap_normal -- about a cap fragment on the sphere at axis_point (capRibbonOfRay: tangent-plane extrapolation)."""
    d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
    out = np.empty(len(d), dtype=np.float32)
    c = np.ascontiguousarray(cam, dtype=np.float32)
    a = np.ascontiguousarray(axis_point, dtype=np.float32)
    t = np.ascontiguousarray(axis_dir, dtype=np.float32)
    ch = np.ascontiguousarray(cap_hit, dtype=np.float32) if cap_hit is not None else None
    cn = np.ascontiguousarray(cap_normal, dtype=np.float32) if cap_normal is not None else None
    lib().lvo_ribbon_of_rays(_p(c), _p(d), len(d), _p(a), _p(t), float(radius), _p(ch) if ch is not None else None,
                             _p(cn) if cn is not None else None, _p(out))
    return out


class ppll_prebaked_ao:
    """with ppll_prebaked_ao(factors, blending_weights): the PPLL gather shades with the static prebaker's table"""

    def __init__(self, factors, blending_weights):
        self.f = np.ascontiguousarray(factors, dtype=np.float32)
        self.w = np.ascontiguousarray(blending_weights, dtype=np.float32)

    def __enter__(self):
        lib().lvo_set_ppll_prebaked_ao(_p(self.f), _p(self.w), len(self.w), self.f.shape[0], self.f.shape[1])
        return self

    def __exit__(self, *exc):
        lib().lvo_set_ppll_prebaked_ao(None, None, 0, 0, 0)
        return False


class ppll_ray_tracer_fragment_colour:
    """Context manager: inside the block the PPLL gather shades its fragments with the RAY TRACER's computeFragmentColor
    (RayHitCommon.glsl -- what rounds 1-2 did) instead of the raster tube shader's variant (deviation measurement)."""

    def __enter__(self):
        lib().lvo_set_ppll_fragment_colour_variant(1)
        return self

    def __exit__(self, *exc):
        lib().lvo_set_ppll_fragment_colour_variant(0)
        return False


_DEFAULT_SWITCHES = [0, 0]      # what the oracle evaluates outside every deviation_switches block
_CURRENT_SWITCHES = [0, 0]
_BLOCK_DEPTH = [0]


def _apply_switches(lit, aol):
    _CURRENT_SWITCHES[0], _CURRENT_SWITCHES[1] = int(lit), int(aol)
    lib().lvo_set_deviation_switches(int(lit), int(aol))


def set_default_intersection_form(literal):
    """The ray-capsule roots the oracle evaluates from now on (outside deviation_switches blocks): False = closest-approach form,
    True = the reference's textbook roots.  tests/common.py Case.oracle_params() calls this with what the case's settings select
    (intersection_form; "auto" = literal with rtao_geometry = triangle_tubes, like the library); tests/conftest.py resets it
    before every test."""
    _DEFAULT_SWITCHES[0] = int(bool(literal))
    if _BLOCK_DEPTH[0] == 0:    # an explicit deviation_switches block keeps what it asked for
        _apply_switches(_DEFAULT_SWITCHES[0], _DEFAULT_SWITCHES[1])


class deviation_switches:
    """Context manager: evaluate the reference's literal intersection roots / AO lookup inside the block (nestable: leaving a
    block restores what was in force when it was entered)."""

    def __init__(self, literal_intersection=False, reference_ao_lookup=False):
        self.args = (int(bool(literal_intersection)), int(bool(reference_ao_lookup)))

    def __enter__(self):
        self.saved = tuple(_CURRENT_SWITCHES)
        _BLOCK_DEPTH[0] += 1
        _apply_switches(*self.args)
        return self

    def __exit__(self, *exc):
        _BLOCK_DEPTH[0] -= 1
        if _BLOCK_DEPTH[0] == 0:
            _apply_switches(*_DEFAULT_SWITCHES)   # (the default may have changed inside the block)
        else:
            _apply_switches(*self.saved)
        return False


# EAW defaults of the RTAO pass (Denoiser.cpp:54-62): 3 iterations, phi x scale
EAW_AO_DEFAULTS = dict(iterations=3, phi_color=0.49, phi_position=0.3 * 0.0001, phi_normal=0.1)


class ao_features:
    """Context manager: the RTAO passes inside the block also write the denoiser's feature maps (view-space normal and
    position, float4 per pixel, full viewport) -> .normal, .position."""

    def __init__(self, width, height):
        self.normal = np.zeros((height, width, 4), dtype=np.float32)
        self.position = np.zeros((height, width, 4), dtype=np.float32)

    def __enter__(self):
        lib().lvo_set_ao_feature_outputs(_p(self.normal), _p(self.position))
        return self

    def __exit__(self, *exc):
        lib().lvo_set_ao_feature_outputs(None, None)
        return False


def mat4_mul(a, b):
    """a * b for column-major 4x4 float32 matrices (flat, 16 values) in glm's evaluation order."""
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(16)
    b = np.ascontiguousarray(b, dtype=np.float32).reshape(16)
    out = np.zeros(16, dtype=np.float32)
    lib().lvo_mat4_mul(_p(a), _p(b), _p(out))
    return out


class Svgf:
    """The temporal state of SVGFDenoiser + the RTAO pass around it (VulkanRayTracedAmbientOcclusionPass::_render with an
    SVGF denoiser): global frame counter, last frame's view-projection, the four history images.  step(render_ao, P) runs ONE
    RTAO iteration through `render_ao` (a callable returning the raw AO image, e.g. lambda: scene.render_ao(P)) followed by
    one denoise() and returns the denoised AO image."""

    def __init__(self, width, height, iterations=5, allowed_z_dist=0.002, allowed_normal_dist=0.02):
        self.w, self.h = int(width), int(height)
        self.iterations, self.allowed_z_dist, self.allowed_normal_dist = int(iterations), float(allowed_z_dist), float(allowed_normal_dist)
        self.global_frame_number = 0
        self.last_view_proj = None
        z = lambda *c: np.zeros((self.h, self.w) + c, dtype=np.float32)
        self.color_history, self.moments_history, self.normal_history, self.depth_history = z(), z(4), z(4), z()
        self.normal, self.depth, self.flow, self.fwidth = z(4), z(), z(2), z()
        self.raw = None

    def step(self, render_ao, P):
        view = np.array(P.view, dtype=np.float32)
        proj = np.array(P.proj, dtype=np.float32)
        vp = mat4_mul(proj, view)
        last = self.last_view_proj if self.last_view_proj is not None else vp
        its = P.aoIterations
        P.aoIterations = 1
        lib().lvo_set_svgf_feature_outputs(_p(self.normal), _p(self.depth), _p(self.flow), _p(self.fwidth), 1,
                                           self.global_frame_number, _p(np.ascontiguousarray(last)))
        try:
            self.raw = np.ascontiguousarray(render_ao(), dtype=np.float32)
        finally:
            lib().lvo_set_svgf_feature_outputs(None, None, None, None, 0, 0, None)
            P.aoIterations = its
        self.global_frame_number += 1
        self.last_view_proj = vp
        out = np.zeros((self.h, self.w), dtype=np.float32)
        lib().lvo_svgf_denoise(self.w, self.h, _p(self.raw), _p(self.normal), _p(self.depth), _p(self.fwidth), _p(self.flow),
                               self.iterations, self.allowed_z_dist, self.allowed_normal_dist, _p(self.color_history),
                               _p(self.moments_history), _p(self.normal_history), _p(self.depth_history), _p(out))
        return out


def eaw_denoise(ao, normal=None, position=None, iterations=3, phi_color=0.49, phi_position=0.3 * 0.0001, phi_normal=0.1,
                use_color=True, use_position=True, use_normal=True, compute_variant=True, tile=None):
    a = np.ascontiguousarray(ao, dtype=np.float32)
    h, w = a.shape
    nm = np.ascontiguousarray(normal, dtype=np.float32) if normal is not None else None
    pm = np.ascontiguousarray(position, dtype=np.float32) if position is not None else None
    x0, y0, tw, th = tile if tile is not None else (0, 0, w, h)
    out = a.copy()
    lib().lvo_eaw_denoise(w, h, _p(a), _p(nm) if nm is not None else None, _p(pm) if pm is not None else None, int(iterations),
                          float(phi_color), float(phi_position), float(phi_normal), int(use_color), int(use_position),
                          int(use_normal), int(compute_variant), x0, y0, tw, th, _p(out))
    return out
