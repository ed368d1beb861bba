"""ctypes binding of the C-ABI in include/linevis_hip.h (liblinevis_hip.so, HIP / gfx950).

This is plumbing for tests, bench.py and the multi-GPU driver; the product is the shared library.  There is no
CPU fallback: importing works anywhere, but `load()` raises if the library has not been built and every compute
call fails with LV_E_HIP when no MI355X is visible.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LV_LIB_PATH") or os.path.join(_HERE, "_lib", "liblinevis_hip.so")  # override: kernel tuning builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "linevis_hip.h")

LV_OK = 0
MODE_PPLL = 2          # RENDERING_MODE_PER_PIXEL_LINKED_LIST, src/Renderers/RenderingModes.hpp:32-53
MODE_RAY_TRACER = 11   # RENDERING_MODE_VULKAN_RAY_TRACER

LINE_POINT_DTYPE = np.dtype([("linePosition", "<f4", 3), ("lineAttribute", "<f4"),
                             ("lineTangent", "<f4", 3), ("lineRotation", "<f4"),
                             ("lineNormal", "<f4", 3), ("lineStartIndex", "<u4")])
assert LINE_POINT_DTYPE.itemsize == 48
# struct TubeTriangleVertexData, src/LineData/LineRenderData.hpp:171-176
TUBE_VERTEX_DTYPE = np.dtype([("vertexPosition", "<f4", 3), ("vertexLinePointIndex", "<u4"),
                              ("vertexNormal", "<f4", 3), ("phi", "<f4")])
assert TUBE_VERTEX_DTYPE.itemsize == 32


class Stats(C.Structure):
    _fields_ = [("rays_traced", C.c_uint64), ("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64),
                ("hits_shaded", C.c_uint64), ("fragments", C.c_uint64), ("ao_hit_pixels", C.c_uint64),
                ("max_depth_complexity", C.c_uint32), ("bvh_depth", C.c_uint32), ("num_segments", C.c_uint32),
                ("num_nodes", C.c_uint32),
                ("ms_accel_build", C.c_float), ("ms_depth_range", C.c_float), ("ms_ao", C.c_float),
                ("ms_color", C.c_float), ("ms_ppll_clear", C.c_float), ("ms_ppll_gather", C.c_float),
                ("ms_ppll_resolve", C.c_float), ("ms_total", C.c_float), ("device_bytes", C.c_uint64),
                ("ms_kernel_avg", C.c_float * 8), ("kernel_launches", C.c_uint32 * 8),
                ("ao_rays_traced", C.c_uint64), ("ao_nodes_visited", C.c_uint64), ("ao_prims_tested", C.c_uint64),
                ("ao_phase_iterations", C.c_uint64 * 3), ("ao_phase_lanes", C.c_uint64 * 3),
                ("max_nodes_per_pixel", C.c_uint32), ("num_tube_triangles", C.c_uint32),
                ("ppll_pool_nodes", C.c_uint64), ("ao_prim_hits", C.c_uint64), ("ao_prim_may_axis", C.c_uint64),
                ("ao_prim_may_both", C.c_uint64),
                ("ms_tri_accel_build", C.c_float), ("ms_tessellate", C.c_float), ("ms_line_points", C.c_float),
                ("num_tri_nodes", C.c_uint32), ("tri_leaf_bytes", C.c_uint32)]

    def as_dict(self):
        d = {}
        for n, _ in self._fields_:
            v = getattr(self, n)
            d[n] = list(v) if hasattr(v, "__len__") else v
        return d


class StreamlineSettings(C.Structure):
    """lv_streamline_settings (StreamlineTracingSettings, StreamlineTracingDefines.hpp:144-177)."""
    _fields_ = [("integration_method", C.c_uint32), ("integration_direction", C.c_uint32),
                ("time_step_scale", C.c_float), ("max_num_iterations", C.c_int32),
                ("termination_distance", C.c_float), ("minimum_length", C.c_float)]


class HelicitySeedingSettings(C.Structure):
    """lv_helicity_seeding_settings (the seeder-related members of StreamlineTracingSettings, StreamlineTracingDefines.hpp:156-174)."""
    _fields_ = [("minimum_separation_distance", C.c_float), ("termination_check_type", C.c_uint32), ("loop_check_mode", C.c_uint32),
                ("termination_distance_self", C.c_float), ("seeding_subsampling_factor", C.c_int32)]

    def __init__(self, minimum_separation_distance=0.08, termination_check_type=1, loop_check_mode=1, termination_distance_self=1.0,
                 seeding_subsampling_factor=1):
        super().__init__(minimum_separation_distance, termination_check_type, loop_check_mode, termination_distance_self,
                         seeding_subsampling_factor)


# STREAMLINE_INTEGRATION_METHOD_NAMES / ..._DIRECTION_NAMES, StreamlineTracingDefines.hpp:77-88
INTEGRATION_METHODS = {"Explicit Euler": 0, "Implicit Euler": 1, "Heun": 2, "Midpoint": 3, "Runge-Kutta 4th Order": 4,
                       "Runge-Kutta-Fehlberg": 5}
INTEGRATION_DIRECTIONS = {"Forward": 0, "Backward": 1, "Forward & Backward": 2}


def streamline_settings(method="Runge-Kutta 4th Order", direction="Forward & Backward", time_step_scale=1.0,
                        max_num_iterations=2000, termination_distance=1.0, minimum_length=0.7):
    return StreamlineSettings(INTEGRATION_METHODS[method], INTEGRATION_DIRECTIONS[direction], time_step_scale,
                              max_num_iterations, termination_distance, minimum_length)


KERNEL_AO_PRIMARY, KERNEL_AO_RAYS, KERNEL_RENDER_RT, KERNEL_PPLL_GATHER, KERNEL_PPLL_RESOLVE, KERNEL_DEPTH_RANGE = range(6)
KERNEL_NAMES = ["k_ao_primary", "k_ao_rays", "k_render_rt", "k_ppll_gather", "k_ppll_resolve", "k_depth_minmax", "k_ppll_shade_prism",
                "k_ppll_raster_prism"]


class LineVisError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("linevis_hip error %d: %s" % (code, message))
        self.code = code


# every symbol include/linevis_hip.h declares (checked by tests/test_abi.py against the header text)
SYMBOLS = ["lv_create", "lv_destroy", "lv_last_error", "lv_version", "lv_set_stream", "lv_set_lines",
           "lv_set_transfer_function", "lv_set_twist_line_texture", "lv_set_camera", "lv_set_background", "lv_set_option", "lv_build_accel",
           "lv_render", "lv_render_device", "lv_render_tiles_device", "lv_get_stats", "lv_reset_timers", "lv_get_kernel_times", "lv_get_ao_tile_costs", "lv_get_dispatch_order", "lv_trace_rays",
           "lv_compute_depth_range", "lv_get_ao", "lv_ppll_get_buffers", "lv_ppll_resolve_buffers", "lv_get_accel",
           "lv_set_tube_triangle_mesh", "lv_trace_rays_triangles", "lv_set_flow_grid", "lv_trace_streamlines", "lv_trace_streamlines_max_helicity_first",
           "lv_get_streamlines", "lv_get_streamline_seed_indices", "lv_set_ao_parametrization", "lv_get_baked_ao", "lv_bake_ao_start", "lv_bake_ao_poll", "lv_get_mlat_trace", "lv_selftest_rsqrt",
           "lv_set_trajectories", "lv_get_lines", "lv_get_tube_triangle_mesh",
           "lv_create_multi", "lv_multi_ranks", "lv_multi_rank_stats", "lv_multi_rebalance", "lv_multi_deal", "lv_tile_deal", "lv_make_tiles"]

_lib = None


def load():
    """Loads liblinevis_hip.so (built in-tree by linevis_amd/build.py); raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: torch ships its own libamdhip64.so (SONAME libamdhip64.so.7).  Importing torch
    # first makes the loader resolve this library's libamdhip64.so.7 dependency to the copy torch already mapped, so
    # device pointers, streams and events are interchangeable with torch tensors / torch.distributed (RCCL).  If the
    # library were loaded first, /opt/rocm's runtime would come in and torch would later map a second one that
    # cannot see the GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, f32, i32, cp = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_int, C.c_char_p
    L.lv_create.restype = vp
    L.lv_create.argtypes = [i32, C.POINTER(i32)]
    L.lv_destroy.restype = None
    L.lv_destroy.argtypes = [vp]
    L.lv_last_error.restype = cp
    L.lv_last_error.argtypes = [vp]
    L.lv_version.restype = cp
    L.lv_version.argtypes = []
    L.lv_create_multi.restype = vp
    L.lv_create_multi.argtypes = [vp, i32, cp, C.POINTER(i32)]
    L.lv_multi_ranks.restype = i32
    L.lv_multi_ranks.argtypes = [vp]
    L.lv_make_tiles.restype = u32
    L.lv_make_tiles.argtypes = [u32, u32, u32, u32, u32, vp, u32]
    for name, args in [
        ("lv_multi_rebalance", [vp, C.c_double]),
        ("lv_multi_rank_stats", [vp, i32, vp]),
        ("lv_bake_ao_start", [vp]),
        ("lv_bake_ao_poll", [vp, C.POINTER(i32), C.POINTER(i32)]),
        ("lv_multi_deal", [vp, vp, u32, C.POINTER(u32)]),
        ("lv_tile_deal", [vp, u32, u32, vp]),
        ("lv_set_stream", [vp, vp]),
        ("lv_set_lines", [vp, vp, u32, vp, u32]),
        ("lv_set_transfer_function", [vp, vp, u32, f32, f32]),
        ("lv_set_twist_line_texture", [vp, vp, u32, u32]),
        ("lv_set_camera", [vp, vp, vp, f32, f32, f32, u32, u32]),
        ("lv_set_background", [vp, vp]),
        ("lv_set_option", [vp, cp, cp]),
        ("lv_build_accel", [vp]),
        ("lv_render", [vp, i32, u32, u32, u32, u32, vp]),
        ("lv_render_device", [vp, i32, u32, u32, u32, u32, vp]),
        ("lv_render_tiles_device", [vp, i32, vp, u32, u32, u32, vp]),
        ("lv_get_stats", [vp, C.POINTER(Stats)]),
        ("lv_reset_timers", [vp]),
        ("lv_get_kernel_times", [vp, i32, vp, u32, C.POINTER(u32)]),
        ("lv_get_ao_tile_costs", [vp, vp, u32, C.POINTER(u32), C.POINTER(u32)]),
        ("lv_get_dispatch_order", [vp, vp, vp, u32, C.POINTER(u32)]),
        ("lv_trace_rays", [vp, vp, vp, f32, f32, u32, vp, vp, vp]),
        ("lv_compute_depth_range", [vp, vp]),
        ("lv_get_ao", [vp, vp]),
        ("lv_ppll_get_buffers", [vp, vp, u64, vp, u64, C.POINTER(u32)]),
        ("lv_ppll_resolve_buffers", [vp, vp, u64, vp, u64, u32, u32, u32, u32, vp]),
        ("lv_get_accel", [vp, vp, u64, vp, u64]),
        ("lv_set_tube_triangle_mesh", [vp, vp, u32, vp, u32, vp, u32]),
        ("lv_set_trajectories", [vp, vp, vp, vp, u32]),
        ("lv_get_lines", [vp, vp, u32, vp, u32, C.POINTER(u32), C.POINTER(u32)]),
        ("lv_get_tube_triangle_mesh", [vp, vp, u32, vp, u32, vp, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]),
        ("lv_trace_rays_triangles", [vp, vp, vp, f32, f32, u32, vp, vp, vp]),
        ("lv_set_flow_grid", [vp, vp, u32, u32, u32, f32, f32, f32, vp, u32]),
        ("lv_trace_streamlines", [vp, vp, u32, C.POINTER(StreamlineSettings), C.POINTER(u64), C.POINTER(u64)]),
        ("lv_trace_streamlines_max_helicity_first", [vp, vp, C.POINTER(StreamlineSettings), C.POINTER(HelicitySeedingSettings),
                                                     C.POINTER(u64), C.POINTER(u64)]),
        ("lv_get_streamlines", [vp, vp, vp, vp]),
        ("lv_get_streamline_seed_indices", [vp, vp]),
        ("lv_set_ao_parametrization", [vp, vp, u32, vp, u32]),
        ("lv_get_baked_ao", [vp, vp, u64]),
        ("lv_get_mlat_trace", [vp, vp, u64, C.POINTER(u64)]),
        ("lv_selftest_rsqrt", [vp, C.POINTER(u64), C.POINTER(u32)]),
    ]:
        fn = getattr(L, name)
        fn.restype = i32
        fn.argtypes = args
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def tile_deal(costs, num_tiles, num_ranks):
    """lv_tile_deal: tile -> rank, longest processing time first (costs None: round robin).  Pure host code."""
    out = np.zeros(max(int(num_tiles), 1), dtype=np.uint32)
    c = np.ascontiguousarray(costs, dtype=np.float64) if costs is not None else None
    rc = load().lv_tile_deal(_p(c) if c is not None else None, int(num_tiles), int(num_ranks), _p(out))
    if rc != LV_OK:
        raise LineVisError(rc, "lv_tile_deal")
    return out[:int(num_tiles)]


def make_tiles(x0, y0, w, h, tile):
    """lv_make_tiles: origins of the tile x tile squares covering the rectangle, along a Morton order.  Pure host code."""
    L = load()
    n = L.lv_make_tiles(int(x0), int(y0), int(w), int(h), int(tile), None, 0)
    out = np.zeros((max(n, 1), 2), dtype=np.uint32)
    L.lv_make_tiles(int(x0), int(y0), int(w), int(h), int(tile), _p(out), n)
    return out[:n]


def _fmt(v):
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, (float, np.floating)):
        return "%.9g" % float(v)  # round-trips float32
    return str(v)


class Context:
    """One renderer context on one HIP device (thin OO veneer over the lv_* calls)."""

    def __init__(self, device=0, devices=None, transport="rccl"):
        """device: one HIP device.  devices = [d0, d1, ...]: lv_create_multi -- one context per device behind this handle, frames
        sharded by screen tiles and gathered on d0 (transport "rccl" | "memcpy")."""
        self.L = load()
        err = C.c_int(0)
        if devices is not None:
            devs = np.ascontiguousarray(devices, dtype=np.int32)
            self.h = self.L.lv_create_multi(_p(devs), len(devs), transport.encode("utf-8"), C.byref(err))
            if not self.h:
                raise LineVisError(err.value, "lv_create_multi(%s, %s) failed" % (list(devs), transport))
            device = int(devs[0])
        else:
            self.h = self.L.lv_create(int(device), C.byref(err))
            if not self.h:
                raise LineVisError(err.value, "lv_create(%d) failed: no usable HIP device (no CPU fallback)" % device)
        self.width = self.height = 0
        self.device = int(device)

    @property
    def num_ranks(self):
        return int(self.L.lv_multi_ranks(self.h))

    def rebalance(self, base_cost_per_tile=4.0 * 64 * 64):
        """Multi-device handle: re-deal the tiles of the last frame by measured cost (lv_multi_rebalance)."""
        self._ck(self.L.lv_multi_rebalance(self.h, float(base_cost_per_tile)))

    def deal(self):
        """tile -> rank of the last frame of a multi-device handle"""
        n = C.c_uint32(0)
        self._ck(self.L.lv_multi_deal(self.h, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.uint32)
        self._ck(self.L.lv_multi_deal(self.h, _p(out), len(out), C.byref(n)))
        return out[:n.value]

    def close(self):
        if getattr(self, "h", None):
            self.L.lv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != LV_OK:
            raise LineVisError(rc, self.L.lv_last_error(self.h).decode("utf-8", "replace"))

    def set_stream(self, stream_handle):
        self._ck(self.L.lv_set_stream(self.h, C.c_void_p(stream_handle) if stream_handle else None))

    def set_lines(self, points, seg_indices):
        pts = np.ascontiguousarray(points, dtype=LINE_POINT_DTYPE)
        seg = np.ascontiguousarray(seg_indices, dtype=np.uint32).reshape(-1, 2)
        self._ck(self.L.lv_set_lines(self.h, _p(pts), len(pts), _p(seg), len(seg)))

    def set_tube_triangle_mesh(self, triangle_indices, vertices, line_points):
        idx = np.ascontiguousarray(triangle_indices, dtype=np.uint32).reshape(-1, 3)
        v = np.ascontiguousarray(vertices, dtype=TUBE_VERTEX_DTYPE)
        pts = np.ascontiguousarray(line_points, dtype=LINE_POINT_DTYPE)
        self._ck(self.L.lv_set_tube_triangle_mesh(self.h, _p(idx), len(idx), _p(v), len(v), _p(pts), len(pts)))

    def set_trajectories(self, positions, attribute, line_offsets):
        """lv_set_trajectories: the trajectories go to HBM, line points / index pairs / tube mesh are written by kernels."""
        pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        off = np.ascontiguousarray(line_offsets, dtype=np.uint32)
        att = None if attribute is None else np.ascontiguousarray(attribute, dtype=np.float32).reshape(-1)
        if len(off) < 1 or int(off[-1]) != len(pos) or (att is not None and len(att) != len(pos)):
            raise ValueError("line_offsets[-1] must equal the number of points (and of attribute values)")
        self._ck(self.L.lv_set_trajectories(self.h, _p(pos), None if att is None else _p(att), _p(off), len(off) - 1))

    def get_lines(self):
        """(points, segment index pairs) as they sit in HBM (lv_set_lines' input or lv_set_trajectories' device output)."""
        n, m = C.c_uint32(), C.c_uint32()
        self._ck(self.L.lv_get_lines(self.h, None, 0, None, 0, C.byref(n), C.byref(m)))
        pts = np.zeros(n.value, dtype=LINE_POINT_DTYPE)
        seg = np.zeros((m.value, 2), dtype=np.uint32)
        self._ck(self.L.lv_get_lines(self.h, _p(pts), n.value, _p(seg), m.value, None, None))
        return pts, seg

    def get_tube_triangle_mesh(self):
        """(triangle indices [n, 3], vertices, line points) of the current tube mesh (tessellated on the device if need be)."""
        nt, nv, npnt = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._ck(self.L.lv_get_tube_triangle_mesh(self.h, None, 0, None, 0, None, 0, C.byref(nt), C.byref(nv), C.byref(npnt)))
        idx = np.zeros((nt.value, 3), dtype=np.uint32)
        v = np.zeros(nv.value, dtype=TUBE_VERTEX_DTYPE)
        pts = np.zeros(npnt.value, dtype=LINE_POINT_DTYPE)
        self._ck(self.L.lv_get_tube_triangle_mesh(self.h, _p(idx), nt.value, _p(v), nv.value, _p(pts), npnt.value, None, None, None))
        return idx, v, pts

    def set_transfer_function(self, rgba, attr_min=0.0, attr_max=1.0):
        tf = np.ascontiguousarray(rgba, dtype=np.float32).reshape(-1, 4)
        self._ck(self.L.lv_set_transfer_function(self.h, _p(tf), tf.shape[0], attr_min, attr_max))

    def set_twist_line_texture(self, rgba8):
        """Twist-line texture of the rotating helicity bands: (h, w, 4) uint8, or None to unload."""
        if rgba8 is None:
            self._ck(self.L.lv_set_twist_line_texture(self.h, None, 0, 0))
            return
        img = np.ascontiguousarray(rgba8, dtype=np.uint8)
        self._ck(self.L.lv_set_twist_line_texture(self.h, _p(img), img.shape[1], img.shape[0]))

    def set_camera(self, view, proj, fov_y, near, far, width, height):
        v = np.ascontiguousarray(view, dtype=np.float32).reshape(16)
        p = np.ascontiguousarray(proj, dtype=np.float32).reshape(16)
        self._ck(self.L.lv_set_camera(self.h, _p(v), _p(p), fov_y, near, far, int(width), int(height)))
        self.width, self.height = int(width), int(height)

    def set_background(self, rgba):
        b = np.ascontiguousarray(rgba, dtype=np.float32).reshape(4)
        self._ck(self.L.lv_set_background(self.h, _p(b)))

    def set_option(self, key, value):
        self._ck(self.L.lv_set_option(self.h, key.encode(), _fmt(value).encode()))

    def set_options(self, settings):
        """LineRenderer::setNewSettings(const SettingsMap&): a dict of string keys."""
        for k, v in settings.items():
            self.set_option(k, v)

    def build_accel(self):
        self._ck(self.L.lv_build_accel(self.h))

    def render(self, mode=MODE_RAY_TRACER, tile=None, out=None):
        x0, y0, w, h = tile if tile is not None else (0, 0, self.width, self.height)
        if out is None:
            out = np.empty((h, w, 4), dtype=np.uint8)
        assert out.shape == (h, w, 4) and out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"]
        self._ck(self.L.lv_render(self.h, mode, x0, y0, w, h, _p(out)))
        return out

    def render_device(self, out_ptr, mode=MODE_RAY_TRACER, tile=None):
        x0, y0, w, h = tile if tile is not None else (0, 0, self.width, self.height)
        self._ck(self.L.lv_render_device(self.h, mode, x0, y0, w, h, C.c_void_p(out_ptr)))

    def render_tiles_device(self, out_ptr, tiles_xy, tile_w, tile_h, mode=MODE_RAY_TRACER):
        t = np.ascontiguousarray(tiles_xy, dtype=np.uint32).reshape(-1, 2)
        self._ck(self.L.lv_render_tiles_device(self.h, mode, _p(t), t.shape[0], tile_w, tile_h, C.c_void_p(out_ptr)))

    def stats(self):
        s = Stats()
        self._ck(self.L.lv_get_stats(self.h, C.byref(s)))
        return s

    def reset_timers(self):
        self._ck(self.L.lv_reset_timers(self.h))

    def trace_rays(self, origins, dirs, t_min, t_max):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        t = np.empty(n, dtype=np.float32)
        seg = np.empty(n, dtype=np.uint32)
        kind = np.empty(n, dtype=np.uint32)
        self._ck(self.L.lv_trace_rays(self.h, _p(o), _p(d), t_min, t_max, n, _p(t), _p(seg), _p(kind)))
        return t, seg, kind

    def trace_rays_triangles(self, origins, dirs, t_min, t_max):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        t = np.empty(n, dtype=np.float32)
        tri = np.empty(n, dtype=np.uint32)
        uv = np.empty((n, 2), dtype=np.float32)
        self._ck(self.L.lv_trace_rays_triangles(self.h, _p(o), _p(d), t_min, t_max, n, _p(t), _p(tri), _p(uv)))
        return t, tri, uv

    def set_ao_parametrization(self, blending_weights, sampling_locations):
        bw = np.ascontiguousarray(blending_weights, dtype=np.float32)
        sl = np.ascontiguousarray(sampling_locations, dtype=np.float32)
        self._ck(self.L.lv_set_ao_parametrization(self.h, _p(bw), len(bw), _p(sl), len(sl)))
        self._bake_shape = len(sl)

    def get_baked_ao(self, num_tube_subdivisions=8):
        out = np.zeros((self._bake_shape, num_tube_subdivisions), dtype=np.float32)
        self._ck(self.L.lv_get_baked_ao(self.h, _p(out), out.size))
        return out

    def set_flow_grid(self, vector_field, spacing, scalar_fields=()):
        """vector_field [zs, ys, xs, 3] float32, scalar_fields: list of [zs, ys, xs] (sampled as line attributes)."""
        v = np.ascontiguousarray(vector_field, dtype=np.float32)
        zs, ys, xs = v.shape[:3]
        sf = [np.ascontiguousarray(f, dtype=np.float32) for f in scalar_fields]
        ptrs = (C.c_void_p * max(len(sf), 1))(*[f.ctypes.data for f in sf])
        self._ck(self.L.lv_set_flow_grid(self.h, _p(v), xs, ys, zs, spacing[0], spacing[1], spacing[2], ptrs, len(sf)))
        self._flow_scalars = len(sf)

    def trace_streamlines(self, seeds, settings):
        """-> (positions [P,3], attributes [k,P], line_offsets [L+1])"""
        sd = np.ascontiguousarray(seeds, dtype=np.float32).reshape(-1, 3)
        nl, npt = C.c_uint64(), C.c_uint64()
        self._ck(self.L.lv_trace_streamlines(self.h, _p(sd), len(sd), C.byref(settings), C.byref(nl), C.byref(npt)))
        pos = np.zeros((npt.value, 3), dtype=np.float32)
        att = np.zeros((self._flow_scalars, npt.value), dtype=np.float32)
        off = np.zeros(nl.value + 1, dtype=np.uint32)
        self._ck(self.L.lv_get_streamlines(self.h, _p(pos), _p(att), _p(off)))
        return pos, att, off

    def trace_streamlines_max_helicity_first(self, helicity_field, settings, seeding=None):
        """StreamlineMaxHelicityFirstSeeder: helicity_field [zs, ys, xs] -> (positions [P,3], attributes [k,P], line_offsets [L+1])"""
        hf = np.ascontiguousarray(helicity_field, dtype=np.float32)
        seeding = seeding or HelicitySeedingSettings()
        nl, npt = C.c_uint64(), C.c_uint64()
        self._ck(self.L.lv_trace_streamlines_max_helicity_first(self.h, _p(hf), C.byref(settings), C.byref(seeding), C.byref(nl),
                                                                C.byref(npt)))
        pos = np.zeros((npt.value, 3), dtype=np.float32)
        att = np.zeros((self._flow_scalars, npt.value), dtype=np.float32)
        off = np.zeros(nl.value + 1, dtype=np.uint32)
        self._ck(self.L.lv_get_streamlines(self.h, _p(pos), _p(att), _p(off)))
        return pos, att, off

    def streamline_seed_indices(self, num_lines):
        out = np.zeros(num_lines, dtype=np.uint32)
        self._ck(self.L.lv_get_streamline_seed_indices(self.h, _p(out)))
        return out

    def depth_range(self):
        out = np.empty(2, dtype=np.float32)
        self._ck(self.L.lv_compute_depth_range(self.h, _p(out)))
        return out

    def get_ao(self):
        out = np.empty((self.height, self.width), dtype=np.float32)
        self._ck(self.L.lv_get_ao(self.h, _p(out)))
        return out

    def mlat_trace(self):
        """Candidate visiting order of the last MLAT frame (collect_stats + mlat_record_trace): (n, 4) uint32 records
        {pixel index y * W + x, sequence number, segment, flag 0 = inserted / 1 = dropped (interval already shorter)}."""
        cnt = C.c_uint64()
        self._ck(self.L.lv_get_mlat_trace(self.h, None, 0, C.byref(cnt)))
        rec = np.zeros((max(int(cnt.value), 1), 4), dtype=np.uint32)
        self._ck(self.L.lv_get_mlat_trace(self.h, _p(rec), rec.shape[0], C.byref(cnt)))
        return rec[:int(cnt.value)]

    def selftest_rsqrt(self):
        """(mismatches, bits of one mismatching argument) of the shading code's 1 / sqrt(x) sequence against IEEE division and square
        root over all 2^32 float arguments, evaluated on the device."""
        bad, first = C.c_uint64(), C.c_uint32()
        self._ck(self.L.lv_selftest_rsqrt(self.h, C.byref(bad), C.byref(first)))
        return int(bad.value), int(first.value)

    def kernel_times(self, kernel_id):
        """Individual launch durations (ms) of one kernel since reset_timers(), oldest first (at most the last 512)."""
        out = np.zeros(512, dtype=np.float32)
        cnt = C.c_uint32()
        self._ck(self.L.lv_get_kernel_times(self.h, int(kernel_id), _p(out), 512, C.byref(cnt)))
        return out[:cnt.value].copy()

    def ao_tile_costs(self):
        """Hit pixels of the last RTAO pass per tile of the last tile list (64x64-pixel groups summed per tile)."""
        n, g = C.c_uint32(), C.c_uint32()
        self._ck(self.L.lv_get_ao_tile_costs(self.h, None, 0, C.byref(n), C.byref(g)))
        out = np.zeros(n.value, dtype=np.uint32)
        self._ck(self.L.lv_get_ao_tile_costs(self.h, _p(out), n.value, C.byref(n), C.byref(g)))
        return out.reshape(-1, max(g.value, 1)).sum(axis=1)

    def dispatch_order(self):
        """(order, cost) of the last frame's 64x64-pixel groups: order[k] = group started k-th, cost[g] = ticks group g took."""
        n = C.c_uint32(0)
        self._ck(self.L.lv_get_dispatch_order(self.h, None, None, 0, C.byref(n)))
        order = np.zeros(n.value, dtype=np.uint32)
        cost = np.zeros(n.value, dtype=np.uint32)
        if n.value:
            self._ck(self.L.lv_get_dispatch_order(self.h, _p(order), _p(cost), n.value, C.byref(n)))
        return order, cost

    def bake_ao_start(self):
        """queue the static RTAO bake on the context's second stream and return"""
        self._ck(self.L.lv_bake_ao_start(self.h))

    def bake_ao_poll(self):
        """(running, ready) of the asynchronous bake; never blocks"""
        a, b = C.c_int(0), C.c_int(0)
        self._ck(self.L.lv_bake_ao_poll(self.h, C.byref(a), C.byref(b)))
        return bool(a.value), bool(b.value)

    def rank_stats(self, rank):
        """lv_stats of one rank of a multi-device handle (nothing summed)."""
        s = Stats()
        self._ck(self.L.lv_multi_rank_stats(self.h, int(rank), C.byref(s)))
        return s

    def ppll_buffers(self, padded_pixels, max_nodes):
        # the library's physical pool may exceed the logical linkedListSize (chunk tails / sub-pools of the gather): room for it
        max_nodes = max(int(max_nodes), int(self.stats().ppll_pool_nodes))
        nodes = np.zeros((max_nodes, 3), dtype=np.uint32)
        start = np.zeros(padded_pixels, dtype=np.uint32)
        cnt = C.c_uint32()
        self._ck(self.L.lv_ppll_get_buffers(self.h, _p(nodes), max_nodes, _p(start), padded_pixels, C.byref(cnt)))
        return nodes, start, cnt.value

    def ppll_resolve(self, nodes, start_offset, tile=None):
        x0, y0, w, h = tile if tile is not None else (0, 0, self.width, self.height)
        n = np.ascontiguousarray(nodes, dtype=np.uint32).reshape(-1, 3)
        s = np.ascontiguousarray(start_offset, dtype=np.uint32)
        out = np.empty((h, w, 4), dtype=np.uint8)
        self._ck(self.L.lv_ppll_resolve_buffers(self.h, _p(n), n.shape[0], _p(s), s.shape[0], x0, y0, w, h, _p(out)))
        return out

    def get_accel(self, num_nodes, num_leaves):
        nodes = np.zeros((max(num_nodes, 1), 16), dtype=np.uint32)
        leaf = np.zeros(max(num_leaves, 1), dtype=np.uint32)
        self._ck(self.L.lv_get_accel(self.h, _p(nodes), nodes.shape[0], _p(leaf), leaf.shape[0]))
        return nodes[:num_nodes], leaf[:num_leaves]
