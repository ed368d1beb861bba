"""ctypes binding of the C++ host layer (linevis_amd/host/*.cpp -> _lib/liblinevis_host.so).

`LineDataFlow` and `HeadlessLineRenderer` are the C++ classes of the same name (LineData model and the
test/benchmark harness around HipRayTracer / HipPerPixelLinkedListLineRenderer); Python only forwards calls.
"""
import ctypes as C
import os

import numpy as np

from . import capi

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib", "liblinevis_host.so")
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("%s is missing: run __graft_entry__.build()" % LIB_PATH)
    capi.load()  # dependency, loaded first so that the rpath-less case still resolves
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, f32, i32, cp = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_int, C.c_char_p
    sig = {
        "lvh_normalize_positions": (None, [vp, u64]),
        "lvh_flow_create": (vp, []),
        "lvh_flow_destroy": (None, [vp]),
        "lvh_flow_set_trajectories": (None, [vp, vp, vp, vp, u32]),
        "lvh_flow_set_trajectories_ribbons": (None, [vp, vp, vp, vp, u32, vp]),
        "lvh_flow_set_trajectories_multi": (None, [vp, vp, vp, u32, vp, vp, u32, vp]),
        "lvh_flow_set_selected_attribute": (None, [vp, C.c_int]),
        "lvh_flow_set_settings": (None, [vp, vp, vp, u32]),
        "lvh_flow_has_helicity": (C.c_int, [vp]),
        "lvh_flow_max_helicity": (C.c_float, [vp]),
        "lvh_flow_use_rotating_helicity_bands": (C.c_int, [vp]),
        "lvh_flow_set_twist_line_texture": (None, [vp, vp, C.c_uint32, C.c_uint32]),
        "lvh_flow_has_bands_data": (i32, [vp]),
        "lvh_flow_get_ribbon_directions": (None, [vp, vp]),
        "lvh_flow_build_render_data_elliptic": (None, [vp, f32, C.POINTER(u32), C.POINTER(u32)]),
        "lvh_flow_load_binlines": (i32, [vp, cp]),
        "lvh_flow_save_binlines": (i32, [vp, cp]),
        "lvh_flow_set_vertices_normalized": (None, [vp, i32]),
        "lvh_flow_vertices_normalized": (i32, [vp]),
        "lvh_flow_num_lines": (u64, [vp]),
        "lvh_flow_num_points": (u64, [vp]),
        "lvh_flow_attribute_range": (None, [vp, vp]),
        "lvh_flow_bounding_box": (None, [vp, vp]),
        "lvh_flow_get_trajectories": (None, [vp, vp, vp, vp]),
        "lvh_flow_build_render_data": (None, [vp, f32, C.POINTER(u32), C.POINTER(u32)]),
        "lvh_flow_copy_render_data": (None, [vp, vp, vp, vp]),
        "lvh_flow_build_triangle_data": (None, [vp, f32, u32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]),
        "lvh_flow_copy_triangle_data": (None, [vp, vp, vp, vp]),
        "lvh_flow_build_triangle_data_bands": (None, [vp, f32, f32, u32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]),
        "lvh_flow_ao_parametrization": (None, [vp, f32, vp, vp, C.POINTER(u64), C.POINTER(u64)]),
        "lvh_grid_create": (vp, [i32]),
        "lvh_grid_destroy": (None, [vp]),
        "lvh_grid_set_extent": (None, [vp, i32, i32, i32, f32, f32, f32]),
        "lvh_grid_add_vector_field": (None, [vp, vp, cp]),
        "lvh_grid_add_scalar_field": (None, [vp, vp, cp]),
        "lvh_grid_load_abc_flow": (None, [vp, i32, i32, i32, f32]),
        "lvh_grid_info": (None, [vp, vp, vp, vp]),
        "lvh_grid_regular_seeds": (None, [vp, i32, i32, i32, vp]),
        "lvh_grid_plane_seeds": (None, [vp, vp, f32, i32, i32, i32, vp]),
        "lvh_grid_trace": (i32, [vp, vp, u32, i32, i32, f32, i32, f32, f32, C.POINTER(u64), C.POINTER(u64)]),
        "lvh_grid_trace_max_helicity_first": (i32, [vp, i32, i32, f32, i32, f32, f32, f32, i32, i32, f32, i32, i32, i32, f32, vp,
                                                    C.POINTER(u64), C.POINTER(u64)]),
        "lvh_grid_copy_result": (None, [vp, vp, vp, vp]),
        "lvh_grid_trace_ribbons": (i32, [vp, vp, u32, i32, i32, f32, i32, f32, f32, i32, f32, vp, C.POINTER(u64), C.POINTER(u64)]),
        "lvh_grid_copy_ribbons": (None, [vp, vp]),
        "lvh_grid_num_scalar_fields": (i32, [vp]),
        "lvh_grid_scalar_field_name": (i32, [vp, i32, vp, u32]),
        "lvh_grid_last_error": (cp, [vp]),
        "lvh_renderer_create": (vp, [i32, i32]),
        "lvh_renderer_create_multi": (vp, [i32, vp, i32, cp]),
        "lvh_renderer_num_devices": (i32, [vp]),
        "lvh_renderer_rebalance": (i32, [vp, C.c_double]),
        "lvh_renderer_set_state": (None, [vp, cp, i32, C.POINTER(cp), C.POINTER(cp), u32, C.POINTER(cp), C.POINTER(cp), u32, i32, i32, i32, i32]),
        "lvh_test_modes_count": (u32, [i32]),
        "lvh_test_mode": (u32, [i32, u32, vp, u32]),
        "lvh_renderer_destroy": (None, [vp]),
        "lvh_renderer_set_resolution": (None, [vp, u32, u32]),
        "lvh_renderer_set_line_data": (None, [vp, vp, i32]),
        "lvh_renderer_set_transfer_function": (None, [vp, vp, u32]),
        "lvh_renderer_set_clear_color": (None, [vp, f32, f32, f32, f32]),
        "lvh_renderer_set_camera": (None, [vp, vp, vp]),
        "lvh_renderer_set_settings": (None, [vp, C.POINTER(cp), C.POINTER(cp), u32]),
        "lvh_renderer_render": (i32, [vp, vp]),
        "lvh_renderer_get_camera": (None, [vp, vp, vp, vp]),
        "lvh_renderer_last_error": (cp, [vp]),
        "lvh_renderer_rendering_mode": (i32, [vp]),
        "lvh_renderer_needs_re_render": (i32, [vp]),
        "lvh_renderer_ao_baker_state": (i32, [vp]),
        "lvh_renderer_context": (vp, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def normalize_positions(positions):
    p = np.ascontiguousarray(positions, dtype=np.float32).copy()
    load().lvh_normalize_positions(_p(p), p.shape[0])
    return p


class LineDataFlow:
    """lv::LineDataFlow (src/LineData/LineDataFlow.{hpp,cpp} counterpart)."""

    def __init__(self):
        self.L = load()
        self.h = self.L.lvh_flow_create()

    def __del__(self):
        try:
            if self.h:
                self.L.lvh_flow_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_trajectories(self, positions, attributes, line_offsets, ribbon_directions=None):
        """setTrajectoryData; ribbon_directions (one vec3 per point) = band data (LineDataFlow::ribbonsDirections)."""
        pos = np.ascontiguousarray(positions, dtype=np.float32)
        att = np.ascontiguousarray(attributes, dtype=np.float32)
        off = np.ascontiguousarray(line_offsets, dtype=np.uint32)
        if ribbon_directions is None:
            self.L.lvh_flow_set_trajectories(self.h, _p(pos), _p(att), _p(off), len(off) - 1)
        else:
            rib = np.ascontiguousarray(ribbon_directions, dtype=np.float32)
            assert rib.shape == pos.shape
            self.L.lvh_flow_set_trajectories_ribbons(self.h, _p(pos), _p(att), _p(off), len(off) - 1, _p(rib))
        return self

    def set_trajectories_multi(self, positions, attributes, names, line_offsets, ribbon_directions=None, selected=0):
        """setTrajectoryData with several attributes per point (attributes[a][n]) and their names; an attribute whose name contains
        "helicity" makes the rotating helicity bands available (LineDataFlow.cpp:535-550)."""
        pos = np.ascontiguousarray(positions, dtype=np.float32)
        att = np.ascontiguousarray(attributes, dtype=np.float32)
        off = np.ascontiguousarray(line_offsets, dtype=np.uint32)
        assert att.ndim == 2 and att.shape[1] == len(pos) and len(names) == att.shape[0]
        nm = (C.c_char_p * len(names))(*[n.encode() for n in names])
        rib = None if ribbon_directions is None else np.ascontiguousarray(ribbon_directions, dtype=np.float32)
        self.L.lvh_flow_set_trajectories_multi(self.h, _p(pos), _p(att), att.shape[0], nm, _p(off), len(off) - 1,
                                               _p(rib) if rib is not None else None)
        self.L.lvh_flow_set_selected_attribute(self.h, int(selected))
        return self

    def set_new_settings(self, settings):
        """LineDataFlow::setNewSettings (use_ribbons, thick_bands, rotating_helicity_bands, separator_width, ...)."""
        keys = [k.encode() for k in settings]
        vals = [capi._fmt(v).encode() for v in settings.values()]
        n = len(keys)
        self.L.lvh_flow_set_settings(self.h, (C.c_char_p * n)(*keys), (C.c_char_p * n)(*vals), n)
        return self

    @property
    def has_helicity(self):
        return bool(self.L.lvh_flow_has_helicity(self.h))

    @property
    def max_helicity(self):
        return float(self.L.lvh_flow_max_helicity(self.h))

    def set_twist_line_texture(self, rgba8):
        """LineDataFlow::loadTwistLineTexture with decoded pixels: (h, w, 4) uint8, or None to unload."""
        if rgba8 is None:
            self.L.lvh_flow_set_twist_line_texture(self.h, None, 0, 0)
        else:
            img = np.ascontiguousarray(rgba8, dtype=np.uint8)
            self.L.lvh_flow_set_twist_line_texture(self.h, img.ctypes.data_as(C.c_void_p), img.shape[1], img.shape[0])
        return self

    @property
    def use_rotating_helicity_bands(self):
        return bool(self.L.lvh_flow_use_rotating_helicity_bands(self.h))

    @property
    def has_bands_data(self):
        return bool(self.L.lvh_flow_has_bands_data(self.h))

    def ribbon_directions(self):
        if not self.has_bands_data:
            return None
        out = np.zeros((self.num_points, 3), dtype=np.float32)
        self.L.lvh_flow_get_ribbon_directions(self.h, _p(out))
        return out

    def tube_aabb_render_data_elliptic(self, band_width):
        """getLinePassTubeAabbRenderData(false, ellipticTubes=true): ribbon normals, boxes padded by band_width / 2."""
        npts, nseg = C.c_uint32(), C.c_uint32()
        self.L.lvh_flow_build_render_data_elliptic(self.h, band_width, C.byref(npts), C.byref(nseg))
        pts = np.zeros(npts.value, dtype=capi.LINE_POINT_DTYPE)
        seg = np.zeros((nseg.value, 2), dtype=np.uint32)
        aabb = np.zeros((nseg.value, 6), dtype=np.float32)
        self.L.lvh_flow_copy_render_data(self.h, _p(pts), _p(seg), _p(aabb))
        return pts, seg, aabb

    def load_file(self, path):
        """LineDataFlow::loadFromFile: .obj or .binlines by extension, positions normalised like the reference loader."""
        return self.load_binlines(path)

    def load_binlines(self, path):
        if self.L.lvh_flow_load_binlines(self.h, path.encode()) != 0:
            raise IOError("loadTrajectoriesFromBinLines failed for %s" % path)
        return self

    def set_vertices_normalized(self, flag=True):
        """Declare the positions handed to set_trajectories as already normalised: saved as verticesNormalized in v2 files."""
        self.L.lvh_flow_set_vertices_normalized(self.h, int(bool(flag)))
        return self

    @property
    def vertices_normalized(self):
        return bool(self.L.lvh_flow_vertices_normalized(self.h))

    def save_binlines(self, path):
        if self.L.lvh_flow_save_binlines(self.h, path.encode()) != 0:
            raise IOError("cannot write %s" % path)

    @property
    def num_lines(self):
        return int(self.L.lvh_flow_num_lines(self.h))

    @property
    def num_points(self):
        return int(self.L.lvh_flow_num_points(self.h))

    def attribute_range(self):
        out = np.empty(2, dtype=np.float32)
        self.L.lvh_flow_attribute_range(self.h, _p(out))
        return float(out[0]), float(out[1])

    def bounding_box(self):
        out = np.empty(6, dtype=np.float32)
        self.L.lvh_flow_bounding_box(self.h, _p(out))
        return out[:3].copy(), out[3:].copy()

    def trajectories(self):
        n, l = self.num_points, self.num_lines
        pos = np.empty((n, 3), dtype=np.float32)
        att = np.zeros(n, dtype=np.float32)
        off = np.empty(l + 1, dtype=np.uint32)
        self.L.lvh_flow_get_trajectories(self.h, _p(pos), _p(att), _p(off))
        return pos, att, off

    def tube_aabb_render_data(self, line_width):
        """getLinePassTubeAabbRenderData: (points[48 B records], seg_indices[S,2], aabbs[S,6])."""
        npts, nseg = C.c_uint32(), C.c_uint32()
        self.L.lvh_flow_build_render_data(self.h, line_width, C.byref(npts), C.byref(nseg))
        pts = np.zeros(npts.value, dtype=capi.LINE_POINT_DTYPE)
        seg = np.zeros((nseg.value, 2), dtype=np.uint32)
        aabb = np.zeros((nseg.value, 6), dtype=np.float32)
        self.L.lvh_flow_copy_render_data(self.h, _p(pts), _p(seg), _p(aabb))
        return pts, seg, aabb


    def tube_triangle_render_data(self, line_width, num_subdivisions=6):
        """getLinePassTubeTriangleMeshRenderData: (triangle_indices[T,3], vertices[32 B], line_points[48 B])."""
        ni, nv, npt = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.L.lvh_flow_build_triangle_data(self.h, line_width, int(num_subdivisions), C.byref(ni), C.byref(nv),
                                            C.byref(npt))
        idx = np.zeros(ni.value, dtype=np.uint32)
        verts = np.zeros(nv.value, dtype=capi.TUBE_VERTEX_DTYPE)
        pts = np.zeros(npt.value, dtype=capi.LINE_POINT_DTYPE)
        self.L.lvh_flow_copy_triangle_data(self.h, _p(idx), _p(verts), _p(pts))
        return idx.reshape(-1, 3), verts, pts


def _tube_triangle_render_data_bands(self, band_width, min_band_thickness=0.15, num_subdivisions=8):
    """getLinePassTubeTriangleMeshRenderData of a band data set: the elliptic triangle tubes."""
    ni, nv, npt = C.c_uint64(), C.c_uint64(), C.c_uint64()
    self.L.lvh_flow_build_triangle_data_bands(self.h, band_width, min_band_thickness, int(num_subdivisions), C.byref(ni),
                                              C.byref(nv), C.byref(npt))
    idx = np.zeros(ni.value, dtype=np.uint32)
    verts = np.zeros(nv.value, dtype=capi.TUBE_VERTEX_DTYPE)
    pts = np.zeros(npt.value, dtype=capi.LINE_POINT_DTYPE)
    self.L.lvh_flow_copy_triangle_data(self.h, _p(idx), _p(verts), _p(pts))
    return idx.reshape(-1, 3), verts, pts


LineDataFlow.tube_triangle_render_data_bands = _tube_triangle_render_data_bands


def _flow_ao_parametrization(self, expected_param_segment_length=0.001):
    """computeAmbientOcclusionParametrization: (blending_weights[num_line_vertices], sampling_locations[M])."""
    nv, npv = C.c_uint64(), C.c_uint64()
    self.L.lvh_flow_ao_parametrization(self.h, expected_param_segment_length, None, None, C.byref(nv), C.byref(npv))
    bw = np.zeros(nv.value, dtype=np.float32)
    sl = np.zeros(npv.value, dtype=np.float32)
    self.L.lvh_flow_ao_parametrization(self.h, expected_param_segment_length, _p(bw), _p(sl), C.byref(nv), C.byref(npv))
    return bw, sl


LineDataFlow.ao_parametrization = _flow_ao_parametrization


class StreamlineTracingGrid:
    """lv::StreamlineTracingGrid (+ AbcFlowGenerator, StreamlineVolumeSeeder): vector field on a regular grid ->
    trajectories, integrated on the GPU."""

    def __init__(self, device=0):
        self.L = load()
        self.h = self.L.lvh_grid_create(int(device))
        if not self.h:
            raise capi.LineVisError(capi.LV_OK - 2, "no usable HIP device (there is no CPU fallback)")
        self.num_scalars = 0

    def __del__(self):
        try:
            if self.h:
                self.L.lvh_grid_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_grid_extent(self, xs, ys, zs, dx, dy, dz):
        self.L.lvh_grid_set_extent(self.h, xs, ys, zs, dx, dy, dz)
        self.num_scalars = 0
        return self

    def add_vector_field(self, field, name="Velocity"):
        f = np.ascontiguousarray(field, dtype=np.float32)
        self.L.lvh_grid_add_vector_field(self.h, _p(f), name.encode())
        return self

    def add_scalar_field(self, field, name):
        f = np.ascontiguousarray(field, dtype=np.float32)
        self.L.lvh_grid_add_scalar_field(self.h, _p(f), name.encode())
        self.num_scalars += 1
        return self

    def load_abc_flow(self, xs=64, ys=64, zs=64, res_scale=6.0):
        """AbcFlowGenerator::load: vector fields Velocity, Vorticity; scalar fields (name order) Helicity, Velocity Magnitude,
        Vorticity Magnitude."""
        self.L.lvh_grid_load_abc_flow(self.h, xs, ys, zs, res_scale)
        self.num_scalars = int(self.L.lvh_grid_num_scalar_fields(self.h))
        return self

    def info(self):
        sizes = np.zeros(3, dtype=np.int32)
        spacing = np.zeros(3, dtype=np.float32)
        box = np.zeros(6, dtype=np.float32)
        self.L.lvh_grid_info(self.h, _p(sizes), _p(spacing), _p(box))
        return sizes, spacing, box

    def attribute_names(self):
        """Names of the scalar fields = the attributes of the traced lines, in attribute order."""
        out = []
        for i in range(int(self.L.lvh_grid_num_scalar_fields(self.h))):
            buf = C.create_string_buffer(256)
            self.L.lvh_grid_scalar_field_name(self.h, i, buf, 256)
            out.append(buf.value.decode())
        return out

    def regular_seeds(self, nx, ny, nz):
        out = np.zeros((nx * ny * nz, 3), dtype=np.float32)
        self.L.lvh_grid_regular_seeds(self.h, nx, ny, nz, _p(out))
        return out

    def plane_seeds(self, normal=(0.0, 1.0, 0.0), slice=0.5, nx=32, ny=32, seed=2):
        """StreamlinePlaneSeeder: nx x ny regular seeds on the plane, or nx random ones (ny = 0)."""
        n = nx * ny if ny > 0 else nx
        out = np.zeros((n, 3), dtype=np.float32)
        nrm = np.ascontiguousarray(normal, dtype=np.float32)
        self.L.lvh_grid_plane_seeds(self.h, _p(nrm), C.c_float(slice), int(nx), int(ny), int(seed), _p(out))
        return out

    def trace_streamlines(self, seeds, method="Runge-Kutta 4th Order", direction="Forward & Backward",
                          time_step_scale=1.0, max_num_iterations=2000, termination_distance=1.0, minimum_length=0.7):
        sd = np.ascontiguousarray(seeds, dtype=np.float32).reshape(-1, 3)
        nl, npt = C.c_uint64(), C.c_uint64()
        rc = self.L.lvh_grid_trace(self.h, _p(sd), len(sd), capi.INTEGRATION_METHODS[method],
                                   capi.INTEGRATION_DIRECTIONS[direction], time_step_scale, max_num_iterations,
                                   termination_distance, minimum_length, C.byref(nl), C.byref(npt))
        if rc != 0:
            raise capi.LineVisError(rc, self.L.lvh_grid_last_error(self.h).decode("utf-8", "replace"))
        pos = np.zeros((npt.value, 3), dtype=np.float32)
        att = np.zeros((self.num_scalars, npt.value), dtype=np.float32)
        off = np.zeros(nl.value + 1, dtype=np.uint32)
        self.L.lvh_grid_copy_result(self.h, _p(pos), _p(att), _p(off))
        return pos, att, off


def _trace_streamlines_max_helicity_first(self, method="Runge-Kutta 4th Order", direction="Forward & Backward", time_step_scale=1.0,
                                          max_num_iterations=2000, termination_distance=1.0, minimum_length=0.7,
                                          minimum_separation_distance=0.08, termination_check_type=1, loop_check_mode=1,
                                          termination_distance_self=1.0, seeding_subsampling_factor=1, ribbons=False,
                                          use_helicity=True, max_helicity_twist=0.25, initial_ribbon_direction=(0.0, 1.0, 0.0)):
    """traceStreamlinesDecreasingHelicity / traceStreamribbonsDecreasingHelicity (StreamlineMaxHelicityFirstSeeder): needs a
    "Helicity" scalar field; ribbons=True also returns the ribbon directions [P,3]."""
    nl, npt = C.c_uint64(), C.c_uint64()
    ird = np.ascontiguousarray(initial_ribbon_direction, dtype=np.float32)
    rc = self.L.lvh_grid_trace_max_helicity_first(self.h, capi.INTEGRATION_METHODS[method], capi.INTEGRATION_DIRECTIONS[direction],
                                                  time_step_scale, max_num_iterations, termination_distance, minimum_length,
                                                  minimum_separation_distance, termination_check_type, loop_check_mode,
                                                  termination_distance_self, seeding_subsampling_factor, int(ribbons), int(use_helicity),
                                                  max_helicity_twist, _p(ird), C.byref(nl), C.byref(npt))
    if rc != 0:
        raise capi.LineVisError(rc, self.L.lvh_grid_last_error(self.h).decode("utf-8", "replace"))
    pos = np.zeros((npt.value, 3), dtype=np.float32)
    att = np.zeros((self.num_scalars, npt.value), dtype=np.float32)
    off = np.zeros(nl.value + 1, dtype=np.uint32)
    self.L.lvh_grid_copy_result(self.h, _p(pos), _p(att), _p(off))
    if not ribbons:
        return pos, att, off
    rib = np.zeros((npt.value, 3), dtype=np.float32)
    self.L.lvh_grid_copy_ribbons(self.h, _p(rib))
    return pos, att, off, rib


StreamlineTracingGrid.trace_streamlines_max_helicity_first = _trace_streamlines_max_helicity_first


def _trace_streamribbons(self, seeds, method="Runge-Kutta 4th Order", direction="Forward & Backward", time_step_scale=1.0,
                         max_num_iterations=2000, termination_distance=1.0, minimum_length=0.7, use_helicity=True,
                         max_helicity_twist=0.25, initial_ribbon_direction=(0.0, 1.0, 0.0)):
    """traceStreamribbons: (positions, attributes, line_offsets, ribbon_directions [P,3]); needs a "Helicity" scalar field."""
    sd = np.ascontiguousarray(seeds, dtype=np.float32).reshape(-1, 3)
    ird = np.ascontiguousarray(initial_ribbon_direction, dtype=np.float32)
    nl, npt = C.c_uint64(), C.c_uint64()
    rc = self.L.lvh_grid_trace_ribbons(self.h, _p(sd), len(sd), capi.INTEGRATION_METHODS[method],
                                       capi.INTEGRATION_DIRECTIONS[direction], time_step_scale, max_num_iterations,
                                       termination_distance, minimum_length, int(use_helicity), max_helicity_twist, _p(ird),
                                       C.byref(nl), C.byref(npt))
    if rc != 0:
        raise capi.LineVisError(rc, self.L.lvh_grid_last_error(self.h).decode("utf-8", "replace"))
    pos = np.zeros((npt.value, 3), dtype=np.float32)
    att = np.zeros((self.num_scalars, npt.value), dtype=np.float32)
    off = np.zeros(nl.value + 1, dtype=np.uint32)
    rib = np.zeros((npt.value, 3), dtype=np.float32)
    self.L.lvh_grid_copy_result(self.h, _p(pos), _p(att), _p(off))
    self.L.lvh_grid_copy_ribbons(self.h, _p(rib))
    return pos, att, off, rib


StreamlineTracingGrid.trace_streamribbons = _trace_streamribbons


def get_test_modes(twice=True):
    """The canned benchmark states of the hot path (lv::getTestModes = InternalState.cpp:46-51,276-297,187-197):
    [(name, rendering_mode, (res_x, res_y), {renderer setting: value})]."""
    L = load()
    out = []
    for i in range(L.lvh_test_modes_count(int(twice))):
        n = L.lvh_test_mode(int(twice), i, None, 0)
        buf = C.create_string_buffer(n + 1)
        L.lvh_test_mode(int(twice), i, buf, n + 1)
        lines = buf.value.decode().split("\n")
        settings = dict(l.split("=", 1) for l in lines[4:] if "=" in l)
        out.append((lines[0], int(lines[1]), (int(lines[2]), int(lines[3])), settings))
    return out


class HeadlessLineRenderer:
    """lv::HeadlessLineRenderer: fixed-camera harness around one renderer plugin (mode 11 or 2)."""

    def __init__(self, mode=capi.MODE_RAY_TRACER, device=0, devices=None, transport="rccl"):
        """devices = [d0, d1, ...]: the plugin drives one context per device (SceneData::deviceOrdinals -> lv_create_multi)."""
        self.L = load()
        if devices is not None:
            devs = np.ascontiguousarray(devices, dtype=np.int32)
            self.h = self.L.lvh_renderer_create_multi(int(mode), _p(devs), len(devs), transport.encode())
        else:
            self.h = self.L.lvh_renderer_create(int(mode), int(device))
        if not self.h:
            raise capi.LineVisError(capi.LV_OK - 2, "no usable HIP device (there is no CPU fallback)")
        self.width = self.height = 128
        self._keep = []

    def __del__(self):
        try:
            if self.h:
                self.L.lvh_renderer_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_rendering_resolution(self, w, h):
        self.width, self.height = int(w), int(h)
        self.L.lvh_renderer_set_resolution(self.h, self.width, self.height)

    @property
    def num_devices(self):
        return int(self.L.lvh_renderer_num_devices(self.h))

    def rebalance(self, base_cost_per_tile=4.0 * 64 * 64):
        if self.L.lvh_renderer_rebalance(self.h, float(base_cost_per_tile)) != 0:
            raise capi.LineVisError(-1, "lv_multi_rebalance failed")

    def set_new_state(self, name, rendering_mode, renderer_settings=None, data_set_settings=None, tiling=(2, 8), resolution=(0, 0)):
        """MainApp::setNewState for the harness (lv::HeadlessLineRenderer::setNewState): renderer_settings carry the camelCase keys
        of the canned states (VulkanRayTracer::setNewState) and / or the snake_case keys of setNewSettings."""
        def arrays(d):
            d = d or {}
            ks = [str(k).encode() for k in d]
            vs = [capi._fmt(v).encode() for v in d.values()]
            return (C.c_char_p * max(len(ks), 1))(*ks), (C.c_char_p * max(len(vs), 1))(*vs), len(ks)
        rk, rv, rn = arrays(renderer_settings)
        dk, dv, dn = arrays(data_set_settings)
        self.L.lvh_renderer_set_state(self.h, str(name).encode(), int(rendering_mode), rk, rv, rn, dk, dv, dn, int(tiling[0]),
                                      int(tiling[1]), int(resolution[0]), int(resolution[1]))
        if resolution[0] > 0 and resolution[1] > 0:
            self.width, self.height = int(resolution[0]), int(resolution[1])

    def set_line_data(self, flow, is_new_data=True):
        self._keep = [flow]
        self.L.lvh_renderer_set_line_data(self.h, flow.h, int(is_new_data))

    def set_transfer_function(self, rgba):
        tf = np.ascontiguousarray(rgba, dtype=np.float32).reshape(-1, 4)
        self.L.lvh_renderer_set_transfer_function(self.h, _p(tf), tf.shape[0])

    def set_clear_color(self, r, g, b, a):
        self.L.lvh_renderer_set_clear_color(self.h, r, g, b, a)

    def set_camera(self, position, look_at=(0.0, 0.0, 0.0)):
        p = np.ascontiguousarray(position, dtype=np.float32)
        l = np.ascontiguousarray(look_at, dtype=np.float32)
        self.L.lvh_renderer_set_camera(self.h, _p(p), _p(l))

    def set_new_settings(self, settings):
        keys = [k.encode() for k in settings]
        vals = [capi._fmt(v).encode() for v in settings.values()]
        n = len(keys)
        ka = (C.c_char_p * n)(*keys)
        va = (C.c_char_p * n)(*vals)
        self.L.lvh_renderer_set_settings(self.h, ka, va, n)

    def render_frame(self):
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        if self.L.lvh_renderer_render(self.h, _p(out)) != 0:
            raise capi.LineVisError(-1, self.L.lvh_renderer_last_error(self.h).decode("utf-8", "replace"))
        return out

    def camera(self):
        """(view, proj, fovy, near, far) exactly as the C++ harness hands them to lv_set_camera."""
        v = np.empty(16, dtype=np.float32)
        p = np.empty(16, dtype=np.float32)
        f = np.empty(3, dtype=np.float32)
        self.L.lvh_renderer_get_camera(self.h, _p(v), _p(p), _p(f))
        return v, p, float(f[0]), float(f[1]), float(f[2])

    @property
    def rendering_mode(self):
        return int(self.L.lvh_renderer_rendering_mode(self.h))

    def needs_re_render(self):
        return bool(self.L.lvh_renderer_needs_re_render(self.h))

    def ao_baker_state(self):
        """(data_ready, computation_running) of the static AO baker, or None without one"""
        v = int(self.L.lvh_renderer_ao_baker_state(self.h))
        return None if v < 0 else (bool(v & 1), bool(v & 2))

    def stats(self):
        s = capi.Stats()
        ctx = self.L.lvh_renderer_context(self.h)
        rc = capi.load().lv_get_stats(ctx, C.byref(s))
        if rc != 0:
            raise capi.LineVisError(rc, "lv_get_stats")
        return s
