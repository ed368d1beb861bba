/*
 * linevis_hip.h -- C-ABI of the MI355X-native line renderer hot path.
 *
 * Drop-in boundary for chrismile/LineVis' ray-traced line renderers.  The reference has no
 * FFI of its own: its renderers (class LineRenderer, src/Renderers/LineRenderer.hpp:66-277) talk to the GPU through
 * sgl/Vulkan objects.  Each entry point below names the reference interface it replaces; a maintainer binds them
 * from a `LineRenderer` subclass as shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no C++/torch types; every pointer is a HOST pointer borrowed for the duration of the call unless
 *     the name says `device` (then it is a HIP device pointer, e.g. a torch tensor's data_ptr()).
 *   - return value: LV_OK (0) or a negative LV_E_* code; lv_last_error(ctx) holds a message. No exceptions cross.
 *   - one context = one HIP device = one calling thread at a time (the reference calls every renderer method from
 *     its main thread, src/MainApp.cpp:914-1013).  Multi-GPU = one context per device / process.
 *   - matrices are float32 column-major (GLM layout).  Camera convention owned by the build (sgl::Camera is not
 *     in the reference tree): right-handed view space looking down -z, depth range [0,1], projection with
 *     y flipped so that image row 0 is the top (see linevis_amd/camera.py).
 *   - images are RGBA8, row-major, 4 bytes per pixel, row 0 = top.
 *   - there is NO CPU fallback: every entry point that computes runs hand-written HIP kernels on gfx950 and fails
 *     with LV_E_HIP if no device is present.
 */
#ifndef LINEVIS_HIP_H
#define LINEVIS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LV_OK 0
#define LV_E_INVALID (-1)   /* bad argument / unknown option / call order */
#define LV_E_HIP (-2)       /* HIP runtime error (message has the hipError string) */
#define LV_E_STATE (-3)     /* missing lines / camera / transfer function / accel */
#define LV_E_CAPACITY (-4)  /* BVH deeper than the traversal stack, buffer too small */

/* RenderingMode ids, src/Renderers/RenderingModes.hpp:32-53 */
#define LV_RENDERING_MODE_PER_PIXEL_LINKED_LIST 2
#define LV_RENDERING_MODE_VULKAN_RAY_TRACER 11

/* struct LinePointDataUnified, src/LineData/LineRenderData.hpp:99-106 -- byte-identical (48 B). */
typedef struct lv_line_point {
    float linePosition[3];
    float lineAttribute;
    float lineTangent[3];
    float lineRotation;
    float lineNormal[3];
    uint32_t lineStartIndex;
} lv_line_point;

/* struct TubeTriangleVertexData, src/LineData/LineRenderData.hpp:171-176 -- byte-identical (32 B). */
typedef struct lv_tube_vertex {
    float vertexPosition[3];
    uint32_t vertexLinePointIndex; /* index into the line-point table; bit 31 set on cap vertices */
    float vertexNormal[3];
    float phi;
} lv_tube_vertex;

/* Counters and timers.  Replaces the per-phase GPU timers of PerPixelLinkedListLineRenderer.cpp:411-420 and the
 * buffer-size reporting of VulkanRayTracer.cpp:671-675.  Ray/node/primitive counters are only filled when the
 * option "collect_stats" is "true" (instrumented kernels; not for timing runs). */
typedef struct lv_stats {
    uint64_t rays_traced;        /* primary + transparency continuation + AO rays */
    uint64_t nodes_visited;      /* 64-byte compressed 4-wide BVH nodes fetched */
    uint64_t prims_tested;       /* 32-byte segment records fetched + capsule tests */
    uint64_t hits_shaded;        /* closest-hit / fragment shading invocations */
    uint64_t fragments;          /* PPLL: value of fragCounter after gather */
    uint64_t ao_hit_pixels;      /* RTAO: pixels whose primary ray hit (compacted list length) */
    uint32_t max_depth_complexity; /* PPLL: longest per-pixel fragment list */
    uint32_t bvh_depth;          /* height of the LBVH */
    uint32_t num_segments;
    uint32_t num_nodes;
    float ms_accel_build;
    float ms_depth_range;
    float ms_ao;                 /* all RTAO kernels of the last lv_render* call */
    float ms_color;              /* ray-tracer colour pass */
    float ms_ppll_clear;
    float ms_ppll_gather;
    float ms_ppll_resolve;
    float ms_total;              /* whole lv_render* call on the stream */
    uint64_t device_bytes;       /* device memory owned by the context */
    /* Per-kernel launch durations from HIP events recorded around every launch on the context's stream, averaged over
     * all launches since lv_reset_timers() (at most the last 512).  Index = LV_KERNEL_*. */
    float ms_kernel_avg[8];
    uint32_t kernel_launches[8];
    /* share of the counters above that belongs to the RTAO sample kernel (k_ao_rays) */
    uint64_t ao_rays_traced;
    uint64_t ao_nodes_visited;
    uint64_t ao_prims_tested;
    /* k_ao_rays lane-utilisation diagnostics (collect_stats only): wave-level iterations and the sum of active lanes
     * of the {refill/setup, node, leaf} phases: utilisation = lanes / (64 * iterations). */
    uint64_t ao_phase_iterations[3];
    uint64_t ao_phase_lanes[3];
    uint32_t max_nodes_per_pixel;  /* collect_stats: most BVH nodes fetched by one pixel of a tile kernel (tail latency) */
    uint32_t num_tube_triangles;   /* triangles of the tube mesh set with lv_set_tube_triangle_mesh */
    uint64_t ppll_pool_nodes;      /* PPLL: physical node slots = linkedListSize + per-wave chunk slack of the gather */
    /* k_ao_rays leaf-test diagnostics (collect_stats only): tests that found a hit inside [0, radius]; tests a
     * conservative axis-distance pre-test would let through; tests axis + bounding-sphere pre-tests would let through */
    uint64_t ao_prim_hits;
    uint64_t ao_prim_may_axis;
    uint64_t ao_prim_may_both;
    /* data set -> first frame on the device (round 6): the triangle LBVH over the tube mesh (its own event pair: ms_accel_build is the
     * segment LBVH only), the device tessellation of lv_set_trajectories' lines (k_tess_*: a14) and the device form of
     * getLinePassTubeAabbRenderData (k_linepoints_*: a2); 0 until the step has run on this context */
    float ms_tri_accel_build;
    float ms_tessellate;
    float ms_line_points;
    uint32_t num_tri_nodes;        /* 64-byte nodes of the triangle LBVH */
    uint32_t tri_leaf_bytes;       /* leaf data per leaf of the triangle LBVH as built: 64 = a pair record (triangle_leaf_records =
                                    * pairs), else 48 x triangle_leaf_size; 0 before the build */
} lv_stats;

#define LV_KERNEL_AO_PRIMARY 0
#define LV_KERNEL_AO_RAYS 1
#define LV_KERNEL_RENDER_RT 2
#define LV_KERNEL_PPLL_GATHER 3
#define LV_KERNEL_PPLL_RESOLVE 4
#define LV_KERNEL_DEPTH_RANGE 5
#define LV_KERNEL_PPLL_SHADE 6   /* fragment stage of ppll_fragment_source = raster_prism (k_ppll_shade_prism) */
#define LV_KERNEL_PPLL_RASTER 7  /* segment rasteriser of raster_prism (k_ppll_raster_prism; ppll_prism_rasteriser = segments) */

typedef struct lv_ctx lv_ctx;

/* Renderer construction/destruction: `new VulkanRayTracer(&sceneData, tfWindow)` + initialize(),
 * src/MainApp.cpp:811,840-846. */
lv_ctx* lv_create(int device_ordinal, int* err);
void lv_destroy(lv_ctx* ctx);
const char* lv_last_error(const lv_ctx* ctx);
const char* lv_version(void);

/* Run all kernels of this context on `hip_stream` (a hipStream_t, e.g. torch.cuda.current_stream().cuda_stream);
 * NULL selects the context's own stream.  No reference analogue (sgl owns the Vulkan queue). */
int lv_set_stream(lv_ctx* ctx, void* hip_stream);

/* LineRenderer::setLineData(LineDataPtr&, bool) (LineRenderer.hpp:98) fed by
 * LineData::getLinePassTubeAabbRenderData (LineData.hpp:186): 48-byte point records + index pairs, copied to HBM. */
int lv_set_lines(lv_ctx* ctx, const lv_line_point* points, uint32_t num_points,
                 const uint32_t* segment_point_indices /* 2 per segment */, uint32_t num_segments);

/* The triangle tubes the reference's RTAO pass traces against (VulkanRayTracedAmbientOcclusion.cpp:444-445 fetches
 * LineData::getLinePassTubeTriangleMeshRenderData(false, true), LineData.hpp:182): index buffer (3 per triangle),
 * 32-byte TubeTriangleVertexData and the 48-byte line-point table the vertices refer to, copied to HBM.  Belongs to the
 * lines of the last lv_set_lines (a later lv_set_lines drops it); with it the option rtao_geometry = "auto" traces the RTAO rays
 * against these triangles like the reference; the triangle LBVH is built on the next render that needs it. */
int lv_set_tube_triangle_mesh(lv_ctx* ctx, const uint32_t* triangle_indices, uint32_t num_triangles,
                              const lv_tube_vertex* vertices, uint32_t num_vertices,
                              const lv_line_point* line_points, uint32_t num_line_points);

/* The same two inputs WITHOUT the host in between (round 6; plain flow lines -- no band data, no rotating helicity bands):
 * LineDataFlow::setTrajectoryData's arrays (LineDataFlow.cpp:468-578: positions, the selected attribute, one offset per line) are
 * copied to HBM (16 bytes per point) and the device writes what lv_set_lines and lv_set_tube_triangle_mesh would have been handed:
 * the 48-byte line points + index pairs of LineDataFlow::getLinePassTubeAabbRenderData (LineDataFlow.cpp:2112-2277) at once, and the
 * capped triangle tubes of getLinePassTubeTriangleMeshRenderData (LineDataFlow.cpp:1912-2110 -> createCappedTriangleTubesRenderDataCPU,
 * CappedTriangleTubesCPU.cpp:214-383) whenever a frame needs them at a line_width / tube_num_subdivisions they have not been
 * tessellated for -- byte for byte the host layer's output (lv_get_lines / lv_get_tube_triangle_mesh read them back).  A line-width
 * change then costs a tessellation + two LBVH builds on the device instead of seconds on the host and a GB-sized upload.
 * positions: 3 floats per point; attribute: 1 float per point or NULL (zeros); line_offsets: num_lines + 1 non-decreasing entries,
 * [0] = 0.  Replaces the lines and the mesh of earlier lv_set_lines / lv_set_tube_triangle_mesh calls; a later lv_set_lines drops the
 * trajectories, a later lv_set_tube_triangle_mesh overrides the device tessellation until the next lv_set_trajectories. */
int lv_set_trajectories(lv_ctx* ctx, const float* positions, const float* attribute, const uint32_t* line_offsets, uint32_t num_lines);
/* Read-backs of the current line points / index pairs and of the current tube mesh (tessellating it first if lv_set_trajectories'
 * lines have none for the current settings).  Any output pointer may be NULL (query the counts). */
int lv_get_lines(lv_ctx* ctx, lv_line_point* out_points, uint32_t max_points, uint32_t* out_segment_point_indices, uint32_t max_segments,
                 uint32_t* out_num_points, uint32_t* out_num_segments);
int lv_get_tube_triangle_mesh(lv_ctx* ctx, uint32_t* out_triangle_indices, uint32_t max_triangles, lv_tube_vertex* out_vertices,
                              uint32_t max_vertices, lv_line_point* out_line_points, uint32_t max_line_points, uint32_t* out_num_triangles,
                              uint32_t* out_num_vertices, uint32_t* out_num_line_points);

/* Static RTAO prebaking (ambient_occlusion_mode = "RTAO (Prebaker)", VulkanAmbientOcclusionBaker.{hpp,cpp,glsl}): AO
 * factors are baked once per geometry for num_parametrization_vertices points along the lines x
 * rtao_prebaker_num_tube_subdivisions angles and looked up at render time (AmbientOcclusion.glsl:49-75), so AO costs
 * nothing per frame and is view independent.  This call supplies the host-side parametrisation of
 * AmbientOcclusionComputeRenderPass::recomputeStaticParametrization (VulkanAmbientOcclusionBaker.cpp:563-653):
 * blending_weights[i] maps line vertex i (an entry of the mesh's line-point table) to a fractional parametrisation
 * index, sampling_locations[j] is the fractional line-vertex position of parametrisation vertex j.  Baking runs lazily
 * before the next frame (or lv_get_baked_ao) against the triangle tubes set with lv_set_tube_triangle_mesh. */
int lv_set_ao_parametrization(lv_ctx* ctx, const float* blending_weights, uint32_t num_line_vertices,
                              const float* sampling_locations, uint32_t num_parametrization_vertices);
/* The baked table: ambientOcclusionFactors[subdivision + num_tube_subdivisions * parametrization_vertex]. */
int lv_get_baked_ao(lv_ctx* ctx, float* out, uint64_t max_values);
/* Asynchronous baking (the reference's BakingMode::MULTI_THREADED, AmbientOcclusionBaker.hpp:66-73, VulkanAmbientOcclusionBaker.cpp:
 * 266-346: a worker thread bakes while the application keeps rendering and shows the AO once baking has finished).  Here: a second
 * HIP stream with scratch buffers of its own.  lv_bake_ao_start queues the bake and returns; frames rendered meanwhile with
 * ambient_occlusion_mode = "RTAO (Prebaker)" come out WITHOUT ambient occlusion; lv_bake_ao_poll (never blocks) reports whether the
 * bake still runs and whether a valid table is in place -- it and the next lv_render* call adopt a finished table.  Changing anything
 * the table depends on (lines, mesh, parametrisation, line width, baker settings) waits for a running bake and discards its result:
 * start again.  Without lv_bake_ao_start the first frame that needs the table bakes it on the context's stream (synchronously to the
 * stream, as before).  lv_get_baked_ao waits for a started bake. */
int lv_bake_ao_start(lv_ctx* ctx);
int lv_bake_ao_poll(lv_ctx* ctx, int* out_running, int* out_ready);

/* TransferFunctionWindow texture + MinMaxUniformBuffer (Data/Shaders/Utils/TransferFunction.glsl:60-71):
 * n RGBA float texels, sampled with linear filtering at texel centres, clamp-to-edge. */
int lv_set_transfer_function(lv_ctx* ctx, const float* rgba, uint32_t n, float attr_min, float attr_max);

/* LineDataFlow::loadTwistLineTexture (LineDataFlow.cpp:93-171; the "twist_line_texture" file is decoded by the embedder): RGBA8
 * pixels, row-major; sampled with REPEAT addressing at (u, 0.5) where the separator stripes of the rotating helicity bands would be
 * drawn (option use_twist_line_texture = USE_HELICITY_BANDS_TEXTURE, twist_line_texture_filtering_mode[_index] = the six names of
 * LineDataFlow.cpp:55-57; twist_line_texture_max_anisotropy must be 1).  The ray tracer samples level 0 (texture() without
 * derivatives, RayHitCommon.glsl:66-72), the rasterised prism of mode 2 textureGrad with the quad's derivatives of (phi +
 * fragmentRotation) / 2 pi (LinePassGeometryShaderTubes.glsl:724-730).  Texel centres, linear weights, level-of-detail formula and
 * the mip chain (2 x 2 box averages in float) are build-owned.  rgba8 == NULL unloads. */
int lv_set_twist_line_texture(lv_ctx* ctx, const uint8_t* rgba8, uint32_t width, uint32_t height);

/* SceneData camera + viewport (src/Renderers/SceneData.hpp:49-85) and LineRenderer::onResolutionChanged
 * (LineRenderer.hpp:127); inverses are taken inside as LineData::updateVulkanUniformBuffers does (LineData.cpp:1290-1291). */
int lv_set_camera(lv_ctx* ctx, const float view[16], const float proj[16], float fov_y, float near_dist,
                  float far_dist, uint32_t viewport_width, uint32_t viewport_height);

/* SceneData::clearColor (SceneData.hpp); foreground = 1 - background (LineData.cpp:1282-1283). */
int lv_set_background(lv_ctx* ctx, const float rgba[4]);

/* LineRenderer::setNewSettings(const SettingsMap&) (LineRenderer.hpp:163): same string keys and encodings
 * (InternalState.hpp:43-125; bools are "true"/"1").  Keys:
 *   line_width, depth_cue_strength, ambient_occlusion_mode ("None" | "RTAO (Screen Space)" | "RTAO (Prebaker)"),
 *   ambient_occlusion_strength, ambient_occlusion_gamma              (LineRenderer.cpp:433-498)
 *   ambient_occlusion_iterations, ambient_occlusion_samples_per_frame, ambient_occlusion_radius,
 *   ambient_occlusion_distance_based, use_jittered_primary_rays        (VulkanRayTracedAmbientOcclusion.cpp:115-144)
 *   num_samples_per_frame, num_accumulated_frames (1 = one self-contained frame per lv_render call, the offline default;
 *   N > 1 = the reference's progressive mode: the caller renders frame after frame with the build-owned key frame_number
 *   = 0, 1, ... N-1, every frame is mixed into the previous one THROUGH RGBA8 exactly like TubeRayTracing.glsl:268-273,
 *   and RTAO runs one iteration per frame while frame_number < ambient_occlusion_iterations),
 *   use_deterministic_sampling, geometry_mode ("AABBs (analytic)" = ray-capsule, default | "Triangle Mesh" = the tube
 *   mesh set with lv_set_tube_triangle_mesh | "Linear Swept Spheres" = the capsules with their caps always in the geometry
 *   and the exact roots: what the NVIDIA hardware primitive with chained end caps is, LineData.cpp:909-945),
 *   use_analytic_intersections (bool form of the first two)
 *                                                                       (VulkanRayTracer.cpp:226-278)
 *   use_mlat (multi-layer alpha tracing instead of the transparency loop; either geometry mode), mlat_num_nodes
 *   (power of two in [1, 32], default 8)                                (VulkanRayTracer.cpp:266-275, .hpp:133-134)
 *   use_capped_tubes, use_halos, tube_num_subdivisions                  (LineData.cpp:87-181)
 *   max_depth_complexity                                                (VulkanRayTracer.hpp:139)
 *   ppll_max_num_frags, ppll_expected_avg_depth_complexity, ppll_tile_width, ppll_tile_height
 *                                                                       (PerPixelLinkedListLineRenderer.cpp:144-209,251-357)
 *   sorting_mode: the "Sorting Mode" combo box of the PPLL renderer (.cpp:470-475, GUI-only in the reference): a name of
 *   SORTING_MODE_NAMES -- "Priority Queue" (default) | "Bubble Sort" | "Insertion Sort" | "Shell Sort" | "Max Heap" |
 *   "Bitonic Sort" | "Quicksort" | "Quicksort Hybrid" (src/Renderers/PPLL.hpp:32-50) -- or its index 0..7,
 *   accel_build (build-owned; the reference builds its BLAS with VK_BUILD_ACCELERATION_STRUCTURE_PREFER_FAST_TRACE_BIT_KHR,
 *   LineData.cpp:740-741): "fast_trace" (default: LBVH whose subtrees of <= treelet_leaves leaves are rebuilt with a binned
 *   surface-area heuristic) | "fast_build" (the plain LBVH); treelet_leaves (3 ... 4096, default 512),
 *   treelet_group_leaves (0 | 8 | 16, default 16), treelet_lane_leaves (0 | 2 ... 64, default 6; used when the former is 0),
 *   treelet_plane_eval ("scan" | "loop"), accel_collapse_top ("true" | "false"): build time only, the tree is the same -- the
 *   small ranges of a treelet are built by groups of 8 / 16 lanes (or one lane per range) instead of by the whole wave; the form
 *   of the wave's plane evaluation; the top levels of the 4-wide collapse in one launch,
 *   kernel_timers (build-owned): "all" (default) | "none" | comma-separated LV_KERNEL_* numbers and / or "phases" -- which launches
 *   of a frame are bracketed by HIP events (lv_get_kernel_times; "phases": the ms_* fields of lv_get_stats, 0 otherwise).  An event
 *   record costs the stream 2 - 4 us: a frame with every kernel and phase bracketed carries a dozen (6 % of a 0.8-ms PPLL frame),
 *   triangle_leaf_size (build-owned): consecutive triangles per leaf of the triangle LBVH, 1 ... 8 (default 2: a tube face);
 *   changes the acceleration structure only, never a hit,
 *   triangle_leaf_records (build-owned): "pairs" (default) | "triangles" -- with two triangles per leaf and a mesh in which the
 *   second triangle of every leaf has at most one vertex index the first one lacks (every mesh the tessellator writes: tube faces,
 *   cap quads, fan pairs), a leaf stores the four vertices once (64 B) instead of two 48-B triangle records; the ray-triangle test
 *   runs on the same operands in the same order, so no hit changes; other meshes and "triangles" keep the 48-B records,
 *   shading_numerics (build-owned): "exact" (default: every operation of the shading code in IEEE float32 with one fixed evaluation
 *   order -- frames, AO factors and PPLL fragments are bit-identical to the CPU checker's) | "fast": the hardware's approximate
 *   reciprocal square root / reciprocal / log2 / exp2 (<= 1 ulp) in arithmetic that only reaches a COLOUR -- the normalisations,
 *   pow() and divisions of blinnPhongShadingTube (Lighting.glsl:100-191) and, in the raster colour of plain tubes, of the halo
 *   coordinate (LinePassGeometryShaderTubes.glsl:732-1129).  Hits, coverage, fragment depths, alpha values, list lengths and the
 *   RTAO factors stay bit-identical; colours move by <= 1 LSB, the contract is +- 2 LSB.  Applies to the ray tracer's capsule colour
 *   pass and the rasterised prism's fragment stage of plain flow lines; all other variants keep the exact arithmetic,
 *   overlap_primary_passes (build-owned; "auto" (default) | "true" | "false"): in a ray-tracer frame with per-frame RTAO the closest hits of the colour
 *   pass' rays do not depend on the AO image, only their shading does -- they are traced in ONE launch with the RTAO pass' primary
 *   rays (two latency-bound passes that a rank owning 1/8 of the tiles cannot fill the GPU with one after the other) and the colour
 *   kernel after the RTAO pass shades them; "false" = one pass after the other as in VulkanRayTracer::render (VulkanRayTracer.cpp:
 *   131-154); "auto" = overlapped while the tile list is at most half a 1920 x 1080 frame (a sharded frame's rank), one after the
 *   other for a whole frame on one GPU, where both passes are throughput-bound.  The frame is byte-identical either way,
 *   dispatch_order (build-owned, no counterpart): "cost" (default: the tile kernels start their 64x64-pixel groups heaviest-of-
 *   the-previous-frame first, see lv_get_dispatch_order) | "as_numbered" (tile-list order); the image is the same,
 *   rtao_prebaker_iterations (128), rtao_prebaker_samples_per_frame (4), rtao_prebaker_num_tube_subdivisions (8): the
 *   prebaker's settings, GUI-only in the reference (VulkanAmbientOcclusionBaker.hpp:108,165-166); radius / distance
 *   based use the ambient_occlusion_* keys,
 *   collect_stats (build-owned: run the instrumented kernels), mlat_record_trace / mlat_trace_capacity (build-owned,
 *   with collect_stats: record the candidate visiting order of an MLAT frame for lv_get_mlat_trace; records, 4 Mi),
 *   rtao_geometry (build-owned): "auto" (default: "triangle_tubes" -- the only geometry the reference's RTAO pass traces,
 *   VulkanRayTracedAmbientOcclusion.cpp:437-456 -- once lv_set_tube_triangle_mesh has been called for the current lines,
 *   "capsules" before that) | "triangle_tubes" (fails without the mesh) | "capsules" (AO rays hit the analytic capsules of the
 *   colour pass: not a mode of the reference),
 *   intersection_form (build-owned): "auto" (default: "literal", the reference's formula, in every frame the reference can
 *   render; "closest_approach" only where the RTAO rays of the frame are traced against the analytic capsules, a mode the
 *   reference does not have -- rays that start on a capsule need the stable roots) |
 *   "literal" (the reference's textbook roots, RayIntersectionTestsVulkan.glsl:39-119, with the own-box rule that keeps them
 *   independent of the BVH) | "closest_approach" (the same roots evaluated stably: NOT the reference's formula),
 *   ppll_fragment_colour (build-owned): "raster" (default: the PPLL gather shades with the raster tube shader's variant of
 *   computeFragmentColor, LinePassGeometryShaderTubes.glsl:785-815,1079-1087 -- EPSILON_OUTLINE = 0, EPSILON_WHITE =
 *   fwidth(ribbonPosition) over the 2 x 2 pixel quad, cap halo min(rp, |rp2|)) | "ray_tracer" (RayHitCommon.glsl's, a probe),
 *   ppll_fragment_source (build-owned): "raster_prism" -- the fragments of mode 2 are those of the geometry the reference
 *   RASTERISES in its default "Tube (Programmable Pull)" mode: per segment the uncapped N-gon prism (N = tube_num_subdivisions) of
 *   LinePassProgrammablePullTubes.glsl:87-224 / LineDataFlow.cpp:1698-1713, back faces culled (LineRasterPass.cpp:85-96), the
 *   fragment shader fed with perspective-correct interpolated position / normal / tangent / attribute; the rasteriser is a
 *   watertight edge-function rasteriser in the space of the pixel's viewing ray (linevis_amd/csrc/lv_prism.h) | "capsule_entry"
 *   -- entry hits of the pixel-centre ray against the analytic capsules / elliptic tubelets (rounds 1-3 of this build; a probe) |
 *   "auto" (default: raster_prism wherever its fragment stage is built -- plain flow lines and band data, whose prism is the
 *   elliptic ring of the USE_BANDS vertex stage, LinePassProgrammablePullTubes.glsl:112-116,166-171, radius band_width / 2,
 *   shaded by the band branch of the raster fragment shader, and the rotating helicity bands (interpolated phi / fragmentRotation,
 *   UNIFORM_HELICITY_BAND_WIDTH from the line points around floor(fragmentVertexId), LinePassGeometryShaderTubes.glsl:1017-1052);
 *   capsule_entry for band data with helicity bands),
 *   ppll_prism_rasteriser (build-owned): front end of raster_prism -- "segments" (default: one lane per line segment over the
 *   screen rectangle of its ring vertices, like the hardware the reference draws with walks primitives, not pixels) | "lbvh"
 *   (the all-hits walk of the viewing rays through the segment LBVH); both decide every (pixel, segment) pair by the same
 *   coverage test and produce the same fragments.  The ORDER of a pixel's fragments is not defined (the reference's fragment
 *   shader invocations race for the list heads, LinkedListGather.glsl:55); where a pixel keeps more fragments than
 *   ppll_max_num_frags, this build resolves the NEAREST ones by the (depth, colour) key (one of the subsets the reference's
 *   race can leave in the first MAX_NUM_FRAGS nodes), so frames do not depend on that order; lv_ppll_get_buffers returns
 *   every list in ascending key order,
 *   ambient_occlusion_denoiser ("None" | "Edge-Avoiding A-Trous Wavelet Transform" (UTF-8 A-grave as in Denoiser.hpp:66; "EAW"
 *   is accepted too) | "SVGF")                                         (VulkanRayTracedAmbientOcclusion.cpp:683-696)
 *   eaw_denoiser_iterations (0..5, default 3), eaw_denoiser_color_weights / _position_weights / _normal_weights,
 *   eaw_denoiser_phi_color / _phi_position / _phi_normal, eaw_denoiser_use_shared_memory   (EAWDenoiser.cpp:400-432)
 *   svgf_denoiser_iterations (0..5, default 5), svgf_denoiser_allowed_z_dist (0.002), svgf_denoiser_allowed_normal_dist
 *   (0.02): build-owned keys for parameters the reference exposes in its GUI only (SVGF.hpp:70-72,115).  SVGF is
 *   temporal: every lv_render* call advances its history (one step per RTAO iteration), always over the whole viewport;
 *   lv_set_lines resets the global frame counter, a change of the denoiser or of the viewport size clears the history,
 *   band data (ribbons; the ray tracer's closest-hit paths: analytic geometry modes, or "Triangle Mesh" / rtao_geometry =
 *   triangle_tubes on the elliptic triangle tubes lv::LineDataFlow tessellates for such data; use_mlat over the same
 *   geometries; the PPLL gather over the analytic tubelets / capsules; the static prebaker: ray origins on the elliptic cross-
 *   section, traced against the elliptic triangle tubes passed to lv_set_tube_triangle_mesh, VulkanAmbientOcclusionBaker.glsl:200-257):
 *   use_ribbons (= USE_BANDS: the line points passed to
 *   lv_set_lines come from a data set with ribbon directions and ribbons are on, LineDataFlow.cpp:587-606,2423-2431),
 *   thick_bands, min_band_thickness (0.15, LineData.cpp:54), band_width (0.005, LineRenderer.cpp:442-449,
 *   DataSetList.hpp:47), use_analytic_elliptic_tubes (build-owned key for the ray tracer's "Elliptic Tubes" checkbox,
 *   VulkanRayTracer.cpp:198-201: the points then carry the ribbon normals of getLinePassTubeAabbRenderData(false, true)
 *   and every segment is a sphere-traced elliptic tubelet, EllipticTubeRayTracing.glsl).
 *   rotating helicity bands of flow lines (USE_ROTATING_HELICITY_BANDS, LineDataFlow.cpp:601-624,2432-2440; excludes
 *   use_ribbons): rotating_helicity_bands (the line points then carry lineRotation, LineDataFlow.cpp:2188-2197),
 *   separator_width (0.2), band_subdivisions (6), helicity_rotation_factor (1), use_uniform_twist_line_width (true;
 *   UNIFORM_HELICITY_BAND_WIDTH, "Triangle Mesh" geometry only: LineAttributesBarycentric.glsl:94-112). */
int lv_set_option(lv_ctx* ctx, const char* key, const char* value);

/* LineData::getRayTracingTubeAabbTopLevelAS (LineData.cpp:1057-1075) + getTubeAabbBottomLevelAS (:879-907):
 * builds the LBVH over segment AABBs min(p0,p1)-r .. max(p0,p1)+r (LineDataFlow.cpp:2230-2233) on the GPU.
 * Called implicitly by lv_render* when lines or line_width changed. */
int lv_build_accel(lv_ctx* ctx);

/* LineRenderer::render() (LineRenderer.hpp:112) for mode 11 (VulkanRayTracer::render, VulkanRayTracer.cpp:131-154:
 * depth range -> RTAO iterations -> colour pass) or mode 2 (PerPixelLinkedListLineRenderer::render,
 * PerPixelLinkedListLineRenderer.cpp:399-427: clear -> gather -> resolve) restricted to the pixel rectangle
 * [x0, x0+w) x [y0, y0+h) of the viewport.  out: w*h*4 bytes. */
int lv_render(lv_ctx* ctx, int rendering_mode, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint8_t* out_rgba8);
int lv_render_device(lv_ctx* ctx, int rendering_mode, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                     void* out_rgba8_device);

/* Same for a list of equally sized tiles (screen-tile sharding).  tiles_xy: 2 uint32 per tile
 * (pixel origin).  Output is tile-major: [num_tiles][tile_h][tile_w][4]; parts of a tile outside the viewport are
 * written as the background colour. */
int lv_render_tiles_device(lv_ctx* ctx, int rendering_mode, const uint32_t* tiles_xy, uint32_t num_tiles,
                           uint32_t tile_w, uint32_t tile_h, void* out_rgba8_device);

/* ---- several GPUs behind one handle (SURVEY.md 8b "Multi-GPU = one context per device", 8e; the reference is single-GPU:
 * sgl::AppSettings::getPrimaryDevice() everywhere, e.g. src/LineData/LineData.cpp:722) ----
 * lv_create_multi: one context per listed device, driven through the returned handle (= rank 0, the first device) by the calling
 * thread; every setter above is repeated on all ranks (full scene replica + LBVH per GPU), every lv_render* call deals the
 * requested pixels as screen tiles over the ranks (64 x 64 along a Morton order for a rectangle; the caller's tiles for
 * lv_render_tiles_device), renders them concurrently and assembles them on rank 0 with ONE gather of RGBA8 tiles:
 * transport "rccl" (default; ncclSend / ncclRecv group over xGMI, librccl resolved at run time; distinct devices) or "memcpy"
 * (hipMemcpyPeerAsync + events; no RCCL needed, the same device may appear several times).  The frame is byte-identical to the
 * single-device frame.  Output pointers of lv_render_device / lv_render_tiles_device live on the first device.  Read-backs
 * (lv_get_ao, lv_ppll_get_buffers, lv_get_kernel_times, ...) address rank 0; lv_get_stats sums the counters of all ranks.
 * lv_multi_rebalance: re-deals the tiles of the last frame by measured cost (RTAO hit pixels x samples per tile + base_cost_per_tile,
 * longest processing time first); synchronises all ranks, call it between frames (progressive accumulation: not while a
 * num_accumulated_frames > 1 sequence is running -- the history stays on the rank that rendered it).  A tile list that differs
 * from the previous call's (another rectangle, other tiles) starts from a fresh round-robin deal: the cost-weighted deal belongs to
 * the list it was measured on.  A rank that owns no tile of a frame (fewer tiles than ranks) renders the list's first tile for
 * itself, so that its temporal state (RTAO seed counter, SVGF history) stays in step with the other ranks'.
 * lv_multi_deal: tile -> rank of the last frame.  lv_tile_deal / lv_make_tiles: the deal and the tile order as pure host
 * functions (costs == NULL: round robin). */
lv_ctx* lv_create_multi(const int* device_ordinals, int num_devices, const char* transport, int* err);
int lv_multi_ranks(const lv_ctx* ctx);
/* lv_get_stats of ONE rank of a multi-device handle (its own counters, phase times and kernel timings, nothing summed): what a
 * scaling run needs to see which rank a frame waited for.  rank 0 = the handle's own device; a single-device context has rank 0 only. */
int lv_multi_rank_stats(lv_ctx* ctx, int rank, lv_stats* out);
int lv_multi_rebalance(lv_ctx* ctx, double base_cost_per_tile);
int lv_multi_deal(lv_ctx* ctx, uint32_t* out_owner, uint32_t capacity, uint32_t* out_count);
int lv_tile_deal(const double* costs, uint32_t num_tiles, uint32_t num_ranks, uint32_t* out_owner);
uint32_t lv_make_tiles(uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t tile, uint32_t* out_xy, uint32_t capacity);

int lv_get_stats(lv_ctx* ctx, lv_stats* out);
/* Forget the per-kernel launch timings collected so far (start of a timed benchmark region). */
int lv_reset_timers(lv_ctx* ctx);
/* Individual launch durations (ms, oldest first) of kernel `kernel_id` (LV_KERNEL_*) since lv_reset_timers(), at most the
 * last 512: what a benchmark needs for a median / p95 (the reference's analogue: the per-phase GPU timers of
 * PerPixelLinkedListLineRenderer.cpp:411-420 / AutomaticPerformanceMeasurer).  Synchronises the stream. */
int lv_get_kernel_times(lv_ctx* ctx, int kernel_id, float* out_ms, uint32_t capacity, uint32_t* out_count);

/* Hit pixels (primary rays of the RTAO pass that hit geometry) per 64x64-pixel group of the last lv_render* call's tile
 * list, in tile-list order, `*out_groups_per_tile` consecutive entries per tile: a rank's per-tile AO cost (AO rays = hit
 * pixels x samples), which the tile sharding uses to re-deal tiles by cost (linevis_amd/tiling.py; SURVEY.md 8e asks for
 * load balance by tile dealing -- the reference itself is single-GPU).  LV_E_STATE unless the last call ran the RTAO pass. */
int lv_get_ao_tile_costs(lv_ctx* ctx, uint32_t* out_counts, uint32_t capacity, uint32_t* out_count, uint32_t* out_groups_per_tile);

/* Dispatch order of the tile kernels (option dispatch_order = cost, the default): the 64x64-pixel groups of the last
 * lv_render* call's tile list in the order their workgroups were started -- heaviest group of the frame before first (longest
 * processing time first; groups are numbered tile-major, `groups per tile` consecutive entries per tile) -- and what every
 * group cost in the last call (device clock ticks summed over its waves: the input of the next call's order).  No counterpart
 * in the reference (the Vulkan driver schedules its ray-generation workgroups); which workgroup renders which group never
 * shows in the image.  *out_count = 0 when the order is off or the launch has fewer than 2 / more than 8192 groups. */
int lv_get_dispatch_order(lv_ctx* ctx, uint32_t* out_order, uint32_t* out_cost, uint32_t capacity, uint32_t* out_count);

/* ---- streamline tracing: the producer of the line sets ----
 * StreamlineTracingGrid (src/LineData/Flow/StreamlineTracingGrid.cpp): regular grid of xs*ys*zs cells with spacing
 * (dx, dy, dz) and origin 0 (setGridExtent, :81-116), one vector field (3 floats per cell, x fastest: IDXV of
 * StreamlineTracingDefines.hpp:118) and any number of scalar fields sampled along the lines as attributes
 * (_pushTrajectoryAttributes, :1012-1047).  max |v| (addVectorField, :189-216) is reduced on the GPU. */
int lv_set_flow_grid(lv_ctx* ctx, const float* vector_field, uint32_t xs, uint32_t ys, uint32_t zs, float dx, float dy,
                     float dz, const float* const* scalar_fields, uint32_t num_scalar_fields);
/* StreamlineTracingSettings, StreamlineTracingDefines.hpp:144-177 (the fields _trace / traceStreamlines read). */
typedef struct lv_streamline_settings {
    uint32_t integration_method;    /* StreamlineIntegrationMethod :63-76: 0 explicit Euler, 1 implicit Euler, 2 Heun,
                                     * 3 midpoint, 4 RK4, 5 Runge-Kutta-Fehlberg (double precision, adaptive step) */
    uint32_t integration_direction; /* StreamlineIntegrationDirection :82-84: 0 forward, 1 backward, 2 both */
    float time_step_scale;          /* 1.0 */
    int32_t max_num_iterations;     /* 2000 */
    float termination_distance;     /* 1.0 (scaled by 1e-6 inside, :1199) */
    float minimum_length;           /* 0.7: shorter lines are dropped (:413-425) */
} lv_streamline_settings;
/* StreamlineTracingGrid::traceStreamlines (:344-426) for caller-supplied seed points (3 floats each): one GPU thread per
 * seed and direction integrates (_trace, :1193-1259), the host part reverses / merges / filters.  The result stays in the
 * context until the next call; fetch it with lv_get_streamlines. */
int lv_trace_streamlines(lv_ctx* ctx, const float* seed_points, uint32_t num_seeds, const lv_streamline_settings* settings,
                         uint64_t* out_num_lines, uint64_t* out_num_points);
/* positions: 3 floats per point; attributes: [num_scalar_fields][num_points]; line_offsets: num_lines + 1.  Any pointer
 * may be NULL. */
int lv_get_streamlines(lv_ctx* ctx, float* positions, float* attributes, uint32_t* line_offsets);

/* StreamlineMaxHelicityFirstSeeder + StreamlineTracingGrid::_traceStreamribbonsDecreasingHelicity (StreamlineSeeder.cpp:360-529,
 * StreamlineTracingGrid.cpp:546-860; Rees et al., "A stream ribbon seeding strategy", EuroVis 2017): the grid points (or, with a
 * subsampling factor, the centres of blocks of cells) are seeded in the order of falling helicity; a line ends where it enters a
 * cell within minimum_separation_distance of an earlier line's point, and a sample whose cell is taken is skipped.  The reference
 * traces these lines one after the other on the CPU; here batches of the next seeds are traced speculatively in parallel on the GPU
 * and committed in seeding order on the host, each line cut where an earlier line of its batch claimed the cell first -- the same
 * lines, point for point.  Built: all four termination_check_types (1 = grid-based, the reference's default; 0 / 2 / 3 = naive /
 * k-d tree / hashed grid: a line ends where it comes within minimum_separation_distance of a POINT of an earlier line -- the lines
 * are cut on the host by that exact predicate, the device tests against the points finished before the batch), all five loop_check_modes (the
 * per-line state of "All Points" / "Grid" / "Curvature" lives in the tracing thread; sgl's HashedGrid and CircularQueue are not
 * vendored: the sphere query is distance <= radius over the line's own points, the queue a FIFO of 32; acos of "Curvature" is the
 * build's fixed formula; getHasPointCloserThan of sgl's KdTree / HashedGrid: distance < r like the naive loop's glm::distance test);
 * not built: Runge-Kutta-Fehlberg (its step width carries over from line to line in the reference).  helicity_field: xs * ys * zs floats (computeHelicityFieldNormalized).
 * The result is fetched with lv_get_streamlines like lv_trace_streamlines'. */
typedef struct lv_helicity_seeding_settings {
    float minimum_separation_distance;   /* 0.08, StreamlineTracingDefines.hpp:158 */
    uint32_t termination_check_type;      /* TerminationCheckType :89-94: 0 naive, 1 grid-based (the default), 2 k-d tree-based, 3 hashed
                                           * grid-based.  0 / 2 / 3 are one predicate -- a point of a finished line closer than
                                           * minimum_separation_distance -- behind three searches in the reference, one here (a uniform
                                           * grid of the finished points in HBM); 0 does not filter the seeds (StreamlineSeeder.cpp:452) */
    uint32_t loop_check_mode;             /* LoopCheckMode :99-101: 0 none, 1 start point, 2 all points, 3 grid, 4 curvature */
    float termination_distance_self;      /* 1.0 (:156): start-point loop check within |box| / 100 times this */
    int32_t seeding_subsampling_factor;   /* 1 (:174) */
} lv_helicity_seeding_settings;
int lv_trace_streamlines_max_helicity_first(lv_ctx* ctx, const float* helicity_field, const lv_streamline_settings* settings,
                                            const lv_helicity_seeding_settings* seeding, uint64_t* out_num_lines,
                                            uint64_t* out_num_points);
/* Per line of the last result: the index of the seed point inside the merged line (0 for forward lines, the last point for
 * backward ones, behind the reversed backward part otherwise).  Streamribbons carry their ribbon direction outwards from the
 * seed in both parts (StreamlineTracingGrid::traceStreamribbons, StreamlineTracingGrid.cpp:428-530). */
int lv_get_streamline_seed_indices(lv_ctx* ctx, uint32_t* out_seed_index);

/* ---- inspection entry points used by the parity tests ---- */
/* Closest hit of arbitrary rays (IntersectionTube + driver closest-hit semantics, TubeRayTracing.glsl:452-494).
 * origins/dirs: 3 floats per ray; out_segment = 0xFFFFFFFF on miss; out_kind: 0 tube, 1 sphere p0, 2 sphere p1. */
int lv_trace_rays(lv_ctx* ctx, const float* origins, const float* dirs, float t_min, float t_max, uint32_t n,
                  float* out_t, uint32_t* out_segment, uint32_t* out_kind);
/* Same against the triangle tubes (the driver's triangle test is restated as Moeller-Trumbore without culling + "t inside
 * the triangle's own padded AABB interval").  out_triangle = 0xFFFFFFFF on miss, ties -> lowest triangle
 * index; out_uv (2 floats per ray, may be NULL) = barycentrics as rayQueryGetIntersectionBarycentricsEXT. */
int lv_trace_rays_triangles(lv_ctx* ctx, const float* origins, const float* dirs, float t_min, float t_max, uint32_t n,
                            float* out_t, uint32_t* out_triangle, float* out_uv);
/* LineRenderer::computeDepthRange (LineRenderer.cpp:410-431): min/max view depth of all line points. */
int lv_compute_depth_range(lv_ctx* ctx, float out_min_max[2]);
/* Full-viewport RTAO texture (.x channel of the RGBA32F accumulation image) after the last mode-11/2 render. */
int lv_get_ao(lv_ctx* ctx, float* out /* viewport_width * viewport_height */);
/* MLAT (use_mlat) parity instrument: the order in which every pixel's candidates were handed to insertNodeMlat in the
 * last mode-11 render (options collect_stats + mlat_record_trace).  4 uint32 per record {viewport pixel index y * W + x,
 * sequence number within the pixel, original segment index (triangle index in the Triangle Mesh mode), flag: 0 = inserted, 1 = dropped because an accepted hit had
 * already shortened the ray interval}; records arrive in no particular order.  The reference's own order is the driver's
 * BVH traversal order (undefined); a CPU replay of THIS order must reproduce the frame.  out_records may be NULL
 * (query the count). */
int lv_get_mlat_trace(lv_ctx* ctx, uint32_t* out_records, uint64_t max_records, uint64_t* out_count);
/* Parity instrument without a reference counterpart: the shading kernels evaluate normalize(v) = v * r(v . v), r(x) = the IEEE-correct
 * bits of 1.0f / sqrtf(min(max(x, 2^-60), 2^60)) (the CPU checker states the same rule), through a shortened instruction sequence.
 * Runs that sequence on ALL 2^32 float arguments on the device against the rule evaluated with the compiler's own division and square
 * root; out_mismatches = number of arguments whose results differ in any bit (NaN results count as equal to each other),
 * out_first_argument = the bits of one of them. */
int lv_selftest_rsqrt(lv_ctx* ctx, uint64_t* out_mismatches, uint32_t* out_first_argument);
/* PPLL buffers after the last mode-2 render: nodes = 3 uint32 {rgba8, depth bits, next} per node slot (slots are handed
 * out to waves in chunks, so unreferenced slots may lie between the stored fragments), start_offset = padded_w * padded_h
 * heads (0xFFFFFFFF = empty), frag_counter = number of fragments generated.  Either pointer may be NULL. */
int lv_ppll_get_buffers(lv_ctx* ctx, uint32_t* out_nodes, uint64_t max_nodes, uint32_t* out_start_offset,
                        uint64_t max_pixels, uint32_t* out_frag_counter);
/* Resolve caller-supplied PPLL buffers (LinkedListResolve.glsl:57-105) with the current camera/options. */
int lv_ppll_resolve_buffers(lv_ctx* ctx, const uint32_t* nodes, uint64_t num_nodes, const uint32_t* start_offset,
                            uint64_t num_pixels, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                            uint8_t* out_rgba8);
/* LBVH export for structural tests: compressed 4-wide nodes of 16 uint32/float words (64 B) each -- words 0-2 grid
 * origin xyz, words 3-5 grid scale xyz (floats), words 6-8 qmin x/y/z and words 9-11 qmax x/y/z (byte k = child slot
 * k; decoded plane = origin + q * scale), words 12-15 child references (bit 31 = leaf, 0xFFFFFFFF = empty slot);
 * leaf order = Morton order; see DESIGN.md. */
int lv_get_accel(lv_ctx* ctx, void* out_nodes, uint64_t max_nodes, uint32_t* out_leaf_segment, uint64_t max_leaves);

#ifdef __cplusplus
}
#endif
#endif /* LINEVIS_HIP_H */
