"""Shared helpers of the test-suite: one scene/config description drives both the CPU oracle (oracle/lvo.py) and
the HIP library (linevis_amd/capi.py) so that parity tests read as "same inputs, compare outputs"."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from linevis_amd import camera, scenes, transfer_function as tfm  # noqa: E402
from oracle import lvo  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


SORTING_MODE_NAMES = ["Priority Queue", "Bubble Sort", "Insertion Sort", "Shell Sort", "Max Heap", "Bitonic Sort", "Quicksort",
                      "Quicksort Hybrid"]   # src/Renderers/PPLL.hpp:32-35


class Case:
    """A scene + camera + settings, convertible to oracle params and to lv_set_option calls."""

    def __init__(self, points, seg, tf, width, height, line_width, camera_pos=camera.DEFAULT_POSITION,
                 background=(1.0, 1.0, 1.0, 1.0), **settings):
        self.points = np.ascontiguousarray(points, dtype=lvo.LINE_POINT_DTYPE)
        self.seg = np.ascontiguousarray(seg, dtype=np.uint32).reshape(-1, 2)
        self.tf = np.ascontiguousarray(tf, dtype=np.float32).reshape(-1, 4)
        self.width, self.height = int(width), int(height)
        self.line_width = float(np.float32(line_width))
        self.view, self.proj, self.fovy, self.near, self.far = camera.default_camera(width, height, camera_pos)
        self.background = tuple(float(x) for x in background)
        # reference SettingsMap keys (LineRenderer.cpp:433-498, VulkanRayTracer.cpp:226-278, ...)
        self.settings = dict(settings)

    # ---- oracle side
    def oracle_scene(self):
        return lvo.Scene(self.points, self.seg, self.tf)

    def literal_form(self):
        """intersection_form as the library resolves it (lv_literal_intersection): "auto" = the reference's literal roots, except
        in frames whose RTAO rays are traced against the analytic capsules (a mode the reference does not have)."""
        s = self.settings
        f = s.get("intersection_form", "auto")
        if f != "auto":
            return f == "literal"
        ao_on = s.get("ambient_occlusion_mode", "None") == "RTAO (Screen Space)" and float(s.get("ambient_occlusion_strength", 0.0)) > 0.0
        return not (ao_on and s.get("rtao_geometry", "capsules") != "triangle_tubes")

    def oracle_params(self, scene=None):
        s = self.settings
        lvo.set_default_intersection_form(self.literal_form())   # the oracle evaluates the form this case's settings select
        spp = int(s.get("num_samples_per_frame", 1))
        ao_on = s.get("ambient_occlusion_mode", "None") == "RTAO (Screen Space)" and \
            float(s.get("ambient_occlusion_strength", 0.0)) > 0.0
        dcs = float(s.get("depth_cue_strength", 0.0))
        kw = dict(
            fovY=self.fovy, nearDist=self.near, farDist=self.far, background=self.background,
            lineWidth=self.line_width,
            maxDepthComplexity=int(s.get("max_depth_complexity", 1024)),
            numSamplesPerFrame=spp, frameNumber=0,
            useJitteredRays=int(spp > 1 or int(s.get("num_accumulated_frames", 1)) > 1),   # VulkanRayTracer.cpp:420-426
            useDeterministicSampling=int(bool(s.get("use_deterministic_sampling", False))),
            useCappedTubes=int(bool(s.get("use_capped_tubes", True))), useHalos=int(bool(s.get("use_halos", True))),
            useDepthCues=int(dcs > 0.0), useAmbientOcclusion=int(ao_on), depthCueStrength=dcs,
            aoStrength=float(s.get("ambient_occlusion_strength", 0.0)), aoGamma=float(s.get("ambient_occlusion_gamma", 1.0)),
            aoSamplesPerFrame=int(s.get("ambient_occlusion_samples_per_frame", 4)),
            aoIterations=int(s.get("ambient_occlusion_iterations", 64)),
            aoUseDistance=int(bool(s.get("ambient_occlusion_distance_based", True))),
            aoJitterPrimary=int(bool(s.get("use_jittered_primary_rays", True))),
            tubeNumSubdivisions=int(s.get("tube_num_subdivisions", 6)),
            aoRadius=float(s.get("ambient_occlusion_radius", 0.1)),
            ppllTileW=int(s.get("ppll_tile_width", 2)), ppllTileH=int(s.get("ppll_tile_height", 8)),
            ppllSortingMode=SORTING_MODE_NAMES.index(s.get("sorting_mode", "Priority Queue")),
            # band data: USE_BANDS / elliptic tubes / MIN_THICKNESS (LineDataFlow.cpp:2423-2431, LineData.cpp:54,1297-1298)
            lssGeometry=int(s.get("geometry_mode") == "Linear Swept Spheres"),
            useHelicityBands=int(bool(s.get("rotating_helicity_bands", False))),
            numSubdivisionsBands=int(s.get("band_subdivisions", 6)),
            separatorBaseWidth=float(np.float32(s.get("separator_width", 0.2))),
            helicityRotationFactor=float(np.float32(s.get("helicity_rotation_factor", 1.0))),
            uniformHelicityBandWidth=int(bool(s.get("use_uniform_twist_line_width", True))),
            useBands=int(bool(s.get("use_ribbons", False))),
            useEllipticTubes=int(bool(s.get("use_ribbons", False)) and bool(s.get("use_analytic_elliptic_tubes", False))),
            bandWidth=float(np.float32(s.get("band_width", 0.005))),
            minBandThickness=float(np.float32(s.get("min_band_thickness", 0.15))),
            minThickness=float(np.float32(s.get("min_band_thickness", 0.15))) if bool(s.get("thick_bands", True)) else float(np.float32(1e-2)),
        )
        # ppll_fragment_source as the library resolves it (lv_ppll_prism_source): auto = the rasterised prism wherever its fragment
        # stage is built (plain flow lines, band data, helicity bands; not band data with helicity bands)
        src = s.get("ppll_fragment_source", "auto")
        built = not (bool(s.get("use_ribbons", False)) and bool(s.get("rotating_helicity_bands", False)))
        kw["ppllFragmentSource"] = int(src == "raster_prism" or (src == "auto" and built))
        large = len(self.seg) > 1000000
        kw["ppllMaxNumFrags"] = int(s.get("ppll_max_num_frags", 0)) or (380 if large else 100)
        P = lvo.make_params(self.view, self.proj, self.width, self.height, **kw)
        avg = int(s.get("ppll_expected_avg_depth_complexity", 0)) or (120 if large else 20)
        pw = -(-self.width // P.ppllTileW) * P.ppllTileW
        ph = -(-self.height // P.ppllTileH) * P.ppllTileH
        P.ppllLinkedListSize = avg * pw * ph
        if P.useDepthCues and scene is not None:
            mm = scene.depth_range(P)
            P.minDepth, P.maxDepth = float(mm[0]), float(mm[1])
        return P

    def padded(self):
        tw, th = int(self.settings.get("ppll_tile_width", 2)), int(self.settings.get("ppll_tile_height", 8))
        return -(-self.width // tw) * tw, -(-self.height // th) * th

    def oracle_render_progressive(self, num_frames, use_bvh=False):
        """The reference's interactive loop (VulkanRayTracer::render called num_frames times): per frame one RTAO iteration
        while frame < ambient_occlusion_iterations, then the colour frame mixed into the previous one through RGBA8.
        Returns the list of frames."""
        sc = self.oracle_scene()
        P = self.oracle_params(sc)
        its = int(P.aoIterations)
        frames, prev, ao = [], None, None
        for f in range(num_frames):
            if P.useAmbientOcclusion and f < its:
                P.aoIterations = f + 1           # iterations 0..f accumulated = the state after frame f's update
                ao = sc.render_ao(P, use_bvh=use_bvh)
            P.frameNumber = f
            prev = sc.render_rt(P, ao=ao, use_bvh=use_bvh, prev=prev) if f else sc.render_rt(P, ao=ao, use_bvh=use_bvh)
            frames.append(prev)
        return frames

    def eaw_settings(self):
        """EAW denoiser of the RTAO pass if enabled: kwargs of lvo.eaw_denoise (phi already x the AO-mode scales), else None."""
        s = self.settings
        if s.get("ambient_occlusion_denoiser", "None") == "None":
            return None
        its = int(s.get("eaw_denoiser_iterations", 3))
        if its == 0:
            return None
        return dict(iterations=its, phi_color=float(s.get("eaw_denoiser_phi_color", 0.49)),
                    phi_position=float(s.get("eaw_denoiser_phi_position", 0.3)) * 0.0001,
                    phi_normal=float(s.get("eaw_denoiser_phi_normal", 0.1)),
                    use_color=bool(s.get("eaw_denoiser_color_weights", True)),
                    use_position=bool(s.get("eaw_denoiser_position_weights", True)),
                    use_normal=bool(s.get("eaw_denoiser_normal_weights", True)),
                    compute_variant=bool(s.get("eaw_denoiser_use_shared_memory", True)))

    def oracle_ao(self, sc, P, tile=None, mode=11, use_bvh=False, stats=None, render_ao=None):
        """The AO image the colour pass samples: RTAO over the tile (+ the halo the lookup / the denoiser read), denoised if
        ambient_occlusion_denoiser is set.  render_ao: alternative RTAO function (triangle tubes)."""
        eaw = self.eaw_settings()
        halo = (1 if (mode == 11 and P.useJitteredRays) else 0) + (2 * (2 ** eaw["iterations"] - 1) if eaw else 0)
        ao_tile = tile
        if tile is not None and halo:
            x0, y0, w, h = tile
            xa, ya = max(x0 - halo, 0), max(y0 - halo, 0)
            ao_tile = (xa, ya, min(x0 + w + halo, self.width) - xa, min(y0 + h + halo, self.height) - ya)
        render_ao = render_ao or (lambda t: sc.render_ao(P, tile=t, use_bvh=use_bvh, stats=stats))
        if not eaw:
            return render_ao(ao_tile)
        with lvo.ao_features(self.width, self.height) as f:
            raw = render_ao(ao_tile)
        return lvo.eaw_denoise(raw, f.normal, f.position, tile=ao_tile, **eaw)

    def oracle_render(self, mode, use_bvh=False, tile=None, stats=None):
        """Full frame as the host orchestration defines it: depth range -> RTAO (-> EAW) -> colour / PPLL."""
        sc = self.oracle_scene()
        P = self.oracle_params(sc)
        ao = self.oracle_ao(sc, P, tile=tile, mode=mode, use_bvh=use_bvh, stats=stats) if P.useAmbientOcclusion else None
        if mode == 11:
            return sc.render_rt(P, ao=ao, tile=tile, use_bvh=use_bvh, stats=stats), ao
        return sc.render_ppll(P, ao=ao, tile=tile, use_bvh=use_bvh, stats=stats), ao

    # ---- HIP side
    def hip_context(self, device=0):
        from linevis_amd import capi
        ctx = capi.Context(device)
        ctx.set_lines(self.points, self.seg)
        ctx.set_transfer_function(self.tf, 0.0, 1.0)
        ctx.set_camera(self.view, self.proj, self.fovy, self.near, self.far, self.width, self.height)
        ctx.set_background(self.background)
        ctx.set_option("line_width", self.line_width)
        ctx.set_options(self.settings)
        return ctx


def scene_arrays(tr, line_width):
    """Trajectories -> (points, seg) through the a2 restatement."""
    pts, seg, _ = lvo.build_tube_aabb_render_data(tr.positions, tr.attributes, tr.line_offsets, line_width)
    return pts, seg


def small_case(width=96, height=64, n_lines=30, pts_per_line=30, seed=7, line_width=0.02, transparent=False,
               **settings):
    tr = scenes.normalize(scenes.random_curves(n_lines=n_lines, points_per_line=pts_per_line, seed=seed))
    pts, seg = scene_arrays(tr, line_width)
    tf = tfm.standard_transparent() if transparent else tfm.standard()
    return Case(pts, seg, tf, width, height, line_width, **settings)


def max_lsb_diff(a, b):
    return int(np.abs(a.astype(np.int32) - b.astype(np.int32)).max())
