// lv_internal.h -- host-side context of the C-ABI library (not installed).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "lv_device.h"

struct LvDeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
};

// Temporal state of the SVGF denoiser (lv_svgf.hip): SVGF_Texture_Pack, SVGF.hpp
struct LvSvgfState {
    LvDeviceBuffer normalDepth, normalDepthHistory;   // float4 {world normal, depth}: this frame / previous frame
    LvDeviceBuffer flowFwidth;                        // float4 {flow.xy, depth fwidth, 0}
    LvDeviceBuffer moments, momentsHistory;           // float4 {m1, m2, history length, 0}
    LvDeviceBuffer colorHistory;                      // float: colour after the first a-trous pass of the previous frame
    LvDeviceBuffer tempAccum, tempAccumFiltered;      // float2 {colour, variance}: reprojection output, after the moments filter
    LvDeviceBuffer ping, pong;                        // float2: a-trous passes
    LvDeviceBuffer result;                            // float: the denoised AO image
    uint32_t width = 0, height = 0;
    bool historyValid = false;                        // false: clear the history images before the next frame
};

// Settings with the reference's defaults (file:line next to each value).
struct LvOptions {
    float lineWidth = 0.002f;                 // src/Loaders/DataSetList.hpp:46
    float depthCueStrength = 0.0f;            // reference default 0.8 (LineRenderer.hpp:220-221); off until set
    bool useAmbientOcclusion = false;         // ambient_occlusion_mode == "RTAO (Screen Space)" && strength > 0
    bool aoBakerIsRtao = false;
    float aoStrength = 0.0f;                  // LineRenderer.hpp (0 = off)
    float aoGamma = 1.0f;
    uint32_t aoIterations = 64;               // VulkanRayTracedAmbientOcclusion.hpp:108
    uint32_t aoSamplesPerFrame = 4;           // VulkanRayTracedAmbientOcclusion.hpp:150
    float aoRadius = 0.1f;                    // :151
    bool aoUseDistance = true;                // :152
    uint32_t triLeafSize = 2;                 // triangle_leaf_size: consecutive triangles per leaf of the triangle LBVH (1 ... 8)
    bool triLeafPairs = true;                 // triangle_leaf_records = pairs (64-B records of four vertices where every pair of triangles shares two, lv_bvh.hip) | triangles (48-B records)
    uint32_t treeletLeaves = 512;             // treelet_leaves: largest subtree the fast_trace build rebuilds (3 ... 4096)
    uint32_t treeletLaneLeaves = 6;           // treelet_lane_leaves: ranges of a treelet up to this size are built one lane per range (0 = the wave splits everything; 2 ... 64; measured: 0 4.5 ms, 4 3.8, 6 3.7, 8 3.9, 16 5.5 for the 1 M segments of config 3)
    uint32_t treeletGroupLeaves = 16;         // treelet_group_leaves: 0 | 8 | 16 -- ranges of <= 8 leaves are built by groups of 8 lanes, with 16 also those of 9 ... 16 leaves by groups of 16 (overrides treelet_lane_leaves; 1 M segments: 0 4.4 ms, 8 2.6, 16 2.4)
    uint32_t timerMask = 0xFFFFFFFFu;         // kernel_timers: bit LV_KERNEL_* = that kernel's launches are bracketed by HIP events, bit 31 = the frame's phase marks.
                                              // An event record costs the stream 2 - 4 us (a barrier packet): 12 of them are 6 % of a config-4 frame, 13 % of a config-2 frame
    bool collapseTop = true;                  // accel_collapse_top: the levels of the wide tree with <= 1024 nodes in one launch (k_collapse_top)
    bool treeletPlaneScan = true;             // treelet_plane_eval = scan (DPP prefix / suffix scans over the bins) | loop (round-3 form)
    bool accelFastTrace = true;               // accel_build = fast_trace (LBVH + SAH treelets, the reference's PREFER_FAST_TRACE) | fast_build (LBVH)
    bool fastShading = false;                 // shading_numerics = fast: approximate hardware rsq / rcp / log2 / exp2 in colour-only arithmetic (lv_device.h)
    int overlapPrimaryPasses = 2;             // overlap_primary_passes: 0 = false, 1 = true, 2 = auto -- the colour pass' hit traces in one launch with the RTAO primaries (k_primary_pair)
    bool dispatchByCost = true;               // dispatch_order = cost | as_numbered (tile kernels: heaviest 64x64 group of the last frame first)
    bool aoJitterPrimary = true;              // :153
    uint32_t numSamplesPerFrame = 1;          // VulkanRayTracer.hpp:137 has 2 (interactive); offline default 1
    uint32_t numAccumulatedFrames = 1;        // :142 (32 interactive); > 1: the caller renders frame_number = 0, 1, ...
    uint32_t frameNumber = 0;                 // accumulatedFramesCounter, VulkanRayTracer.cpp:141
    bool useDeterministicSampling = false;
    uint32_t maxDepthComplexity = 1024;       // VulkanRayTracer.hpp:139
    bool useCappedTubes = true;               // LineData.hpp:377-379
    bool useHalos = true;
    uint32_t tubeNumSubdivisions = 6;         // LineData.cpp:52
    uint32_t ppllSortingMode = 0;             // sorting_mode: index into SORTING_MODE_NAMES (PPLL.hpp:32-35), PerPixelLinkedListLineRenderer.hpp:113
    uint32_t ppllMaxNumFrags = 0;             // 0 = auto: 100 (<=1M segments) / 380, PerPixelLinkedListLineRenderer.hpp:45-49
    uint32_t ppllExpectedAvgDepthComplexity = 0; // 0 = auto: 20 / 120
    uint32_t ppllTileW = 2, ppllTileH = 8;    // LineRenderer.cpp:739-740
    bool collectStats = false;
    bool aoPrebaked = false;                  // ambient_occlusion_mode == "RTAO (Prebaker)"
    uint32_t bakeIterations = 128;            // VulkanAmbientOcclusionBaker.hpp:108
    uint32_t bakeNumTubeSubdivisions = 8;     // :165
    uint32_t bakeSamplesPerFrame = 4;         // :166 (radius / distance-based share the RTAO keys; same defaults :167-168)
    bool rtLss = false;                       // geometry_mode "Linear Swept Spheres" (VulkanRayTracer.hpp:56-63)
    bool rtTriangleMesh = false;              // geometry_mode "Triangle Mesh" / use_analytic_intersections=false (VulkanRayTracer.cpp:226-250)
    int rtaoGeometry = 0;                     // rtao_geometry: 0 = auto (the reference's triangle tubes once lv_set_tube_triangle_mesh was called), 1 = capsules, 2 = triangle_tubes
    bool useMlat = false;                     // VulkanRayTracer.hpp:133
    uint32_t mlatNumNodes = 8;                // :134
    // EAW denoiser of the RTAO pass (ambient_occlusion_denoiser; AO defaults of createDenoiserObject, Denoiser.cpp:54-62)
    bool eawEnabled = false;
    uint32_t eawIterations = 3;               // eaw_denoiser_iterations (GUI range 0..5, EAWDenoiser.cpp:437)
    bool eawColorWeights = true, eawPositionWeights = true, eawNormalWeights = true;
    float eawPhiColor = 0.49f, eawPhiPosition = 0.3f, eawPhiNormal = 0.1f;
    bool eawUseSharedMemory = true;           // true: EAWDenoise.Compute (default), false: EAWDenoise.Fragment
    // ambient_occlusion_denoiser = "SVGF" (Denoiser.hpp:65).  The reference exposes these in the GUI only (SVGF.cpp:427-436,
    // SVGF.hpp:70-72,115); the svgf_denoiser_* keys are this build's.
    // band data (ribbons): use_ribbons / thick_bands / min_band_thickness (LineDataFlow.cpp:587-606), band_width
    // (LineRenderer.cpp:442-449, STANDARD_BAND_WIDTH DataSetList.hpp:47), min band thickness LineData.cpp:54; the ray tracer's
    // "Elliptic Tubes" switch has no settings key in the reference (VulkanRayTracer.cpp:198-201): use_analytic_elliptic_tubes
    bool useRibbons = false, thickBands = true, ellipticTubes = false;
    float bandWidth = 0.005f, minBandThickness = 0.15f;
    // rotating helicity bands of flow lines with a helicity attribute: rotating_helicity_bands, separator_width (0.2,
    // LineDataFlow.cpp:54), band_subdivisions (6, LineDataFlow.hpp:188), helicity_rotation_factor (1, :171); settings keys :601-624
    bool helicityBands = false, uniformTwistLineWidth = true; // use_uniform_twist_line_width, LineDataFlow.cpp:53
    float separatorWidth = 0.2f, helicityRotationFactor = 1.0f;
    bool useTwistLineTexture = false;         // use_twist_line_texture (USE_HELICITY_BANDS_TEXTURE once a texture is loaded, LineDataFlow.cpp:2437)
    uint32_t twistFilterMode = 5;             // twist_line_texture_filtering_mode[_index]: "Linear Mipmap Linear" (LineDataFlow.hpp default)
    uint32_t bandSubdivisions = 6;
    bool svgfEnabled = false;
    uint32_t svgfIterations = 5;              // maxNumIterations, SVGF.hpp:115 (GUI range 0..5)
    float svgfAllowedZDist = 0.002f, svgfAllowedNormalDist = 0.02f; // SVGF.hpp:70-71
    // intersection_form: 0 = auto (default): the reference's literal roots whenever the RTAO pass traces the reference's
    // triangle tubes (rtao_geometry = triangle_tubes: the colour pass is then the only user of the capsule test and the
    // literal form is nearly free), closest approach otherwise; 1 = closest_approach; 2 = literal
    int intersectionForm = 0;
    // ppll_fragment_source: 0 = auto (default): raster_prism for plain flow lines, capsule_entry for band data / helicity bands
    // (not built for the prism yet); 1 = capsule_entry (entry hits of the pixel-centre ray against the analytic
    // capsules: the probe of rounds 1-3); 2 = raster_prism (the rasterised N-gon prism of the reference's default primitive mode)
    int ppllFragmentSource = 0;
    // ppll_prism_rasteriser: false = "segments" (default: one lane per line segment over its screen rectangle, k_ppll_raster_prism),
    // true = "lbvh" (the all-hits walk of the viewing rays through the segment LBVH, k_ppll_gather<LV_PRIM_PRISM>); same fragments
    bool ppllPrismLbvhWalk = false;
    bool ppllRayTracerColour = false;         // ppll_fragment_colour: false = "raster" (the reference's gather shader, default), true = "ray_tracer"
    bool mlatRecordTrace = false;             // with collect_stats: record every pixel's candidate visiting order
    uint32_t mlatTraceCapacity = 1u << 22;    // records (16 B each)
};

struct LvMulti;
struct lv_ctx {
    LvMulti* multi = nullptr;                 // lv_create_multi: the ranks this handle drives (lv_multi.hip); null = one device
    int device = 0;
    int numCUs = 256;
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    std::string lastError;

    // scene
    uint32_t numPoints = 0, numSegs = 0, numNodes = 0;
    LvDeviceBuffer points, segIdx;            // input order
    LvDeviceBuffer nodes, segs, segAxis, leafSeg, segToLeaf; // accel
    LvDeviceBuffer prismFrames;               // per leaf: the frames of the segment's two line points (lv_prism.h), 64 B
    LvDeviceBuffer tf;
    uint32_t tfN = 0;
    float attrMin = 0.0f, attrMax = 1.0f;
    bool accelValid = false;
    float accelLineWidth = -1.0f;
    uint32_t bvhDepth = 0, wideDepth = 0;     // height of the binary LBVH (reported) / levels of the 4-wide tree (stack size)
    // triangle tubes (RTAO geometry of the reference), own LBVH
    uint32_t numTris = 0, numTriVerts = 0, numTriPoints = 0, numTriNodes = 0, triBvhDepth = 0, triWideDepth = 0;
    LvDeviceBuffer triIdx, triVerts, triPoints; // input order
    LvDeviceBuffer triNodes, tris;              // accel
    LvDeviceBuffer triPairFlag;                 // k_tri_pairs_check's verdict
    bool triLeafPairs = false;                  // the leaves of the triangle LBVH as built hold 64-B pair records
    bool triMeshSet = false, triAccelValid = false;
    // lv_set_trajectories (lv_lines.hip): the trajectories themselves in HBM; `points` / `segIdx` are then written by the a2 kernels and
    // the tube mesh is tessellated on the device whenever a frame needs it at another line width / subdivision count
    LvDeviceBuffer trajPos, trajAttr, trajOff;      // 3 floats / 1 float per point, numLines + 1 offsets
    LvDeviceBuffer trajLineValid, trajLineRef, trajRecLine, trajTess; // per line: valid points, LvLineRef; per record: its line; tessellation offsets + tables
    uint32_t trajNumLines = 0, trajNumPoints = 0;
    bool trajHasAttr = false, trajSet = false;
    bool triMeshFromTraj = false;             // the mesh in triIdx / triVerts / triPoints was tessellated here (not passed by the caller)
    float triMeshLineWidth = -1.0f;
    uint32_t triMeshSubdivisions = 0;
    float triAccelLineWidth = -1.0f, triPad = 0.0f;
    uint32_t triLeafSize = 1;                 // triangles per leaf of the triangle LBVH as built (opt.triLeafSize at build time)

    // static RTAO prebaker
    LvDeviceBuffer bakeBlendingWeights, bakeSamplingLocations, bakedAo, bakeLcgSkip;
    uint32_t bakeNumLineVertices = 0, bakeNumParametrizationVertices = 0;
    bool bakeParamSet = false, bakeValid = false;
    // asynchronous bake (lv_bake_ao_start): the table is computed on a second stream with scratch buffers of its own while frames keep
    // rendering on the context's stream; lv_bake_ao_poll / the next frame adopt it once the stream has finished
    hipStream_t bakeStream = nullptr;
    hipEvent_t evBakePrereq = nullptr, evBakeDone = nullptr;
    bool bakeAsyncPending = false;            // a bake is queued / running on bakeStream
    uint64_t bakeGeneration = 0, bakePendingGeneration = 0; // bumped by everything that invalidates a table
    LvDeviceBuffer bakedAoPending, bakeCounters, bakeGbuf, bakeSamples, bakeOverflow;
    std::vector<uint32_t> bakeSkipHost;       // LCG skip-ahead table (source of an asynchronous upload: must outlive the call)
    uint32_t bakeSlotsHost = 0;

    // streamline tracing (lv_flow.hip)
    LvDeviceBuffer flowVectors, flowScalars, flowMisc, flowSeeds, flowOutPos, flowOutAtt, flowCounts;
    uint32_t flowXs = 0, flowYs = 0, flowZs = 0, flowNumScalars = 0;
    float flowDx = 1.0f, flowDy = 1.0f, flowDz = 1.0f, flowMaxMagnitude = 0.0f;
    bool flowGridSet = false;
    std::vector<float> flowPositions;                 // result of the last lv_trace_streamlines call (host side)
    std::vector<std::vector<float>> flowAttributes;
    std::vector<uint32_t> flowOffsets;
    std::vector<uint32_t> flowSeedIndex;              // per line: index of the seed point inside the merged line

    // camera
    bool cameraSet = false;
    float view[16] = {}, proj[16] = {}, invView[16] = {}, invProj[16] = {};
    float fovY = 0.0f, nearDist = 0.01f, farDist = 100.0f;
    uint32_t width = 0, height = 0;
    float background[4] = {1.0f, 1.0f, 1.0f, 1.0f};

    LvOptions opt;

    // frame resources
    LvDeviceBuffer depthMinMax;               // 2 floats (+2 encoded uints)
    LvDeviceBuffer ao;                        // width*height floats (the latest AO image)
    LvDeviceBuffer aoAlt;                     // second image of the halo mode's ping-pong accumulation (lv_run_ao)
    LvDeviceBuffer featNormal, featNormalAlt, featPosition, featPositionAlt; // EAW feature maps (float4 per pixel) + ping-pong
    LvDeviceBuffer eawPing, eawPong;          // a-trous passes
    const float* aoResult = nullptr;          // what the colour pass samples: ao (raw) or the denoised image
    bool aoRestart = false;                   // the denoiser changed: a progressive RTAO accumulation may only continue from frame 0
    LvSvgfState svgf;
    LvDeviceBuffer fullFrameTile;             // one tile origin (0, 0): the SVGF chain always covers the viewport
    uint32_t aoGlobalFrameNumber = 0;         // RTAO iterations since lv_set_lines (globalFrameNumber, ...AmbientOcclusion.cpp:582)
    float lastFrameViewProj[16] = {};         // projection * view at the previous RTAO iteration (:456,631)
    bool lastFrameViewProjValid = false;
    uint32_t tilesHalo = 0;                   // halo the uploaded tilesHaloDev list was built for
    LvDeviceBuffer tilesHaloDev;              // tile origins - 1 (AO pass on dilated tiles)
    std::vector<uint32_t> tilesHaloHost;
    bool tilesHaloUploaded = false;
    uint32_t aoNumGroups = 0, aoGroupsPerTile = 0; // geometry of the last RTAO pass' per-group counters (aoList)
    LvDeviceBuffer aoGbuf, aoList, aoSamples; // RTAO wavefront buffers
    LvDeviceBuffer firstHit;                  // k_primary_pair: {t bits, (leaf << 2) | kind} of the colour pass' first ray per output pixel
    LvDeviceBuffer counters;                  // device counters (LvCounters + misc)
    LvDeviceBuffer ppllNodes, ppllStart, ppllCount, ppllScratch;
    LvDeviceBuffer prismRecords, scanTemp;    // raster_prism: {pixel, leaf | triangle, rank} records of the coverage kernel; k_ppll_scan: block totals, block bases, completion counter
    uint32_t ppllScanBlocks = 0;              // raster_prism: workgroups of the last k_ppll_scan (scanTemp = totals, bases, counter)
    LvDeviceBuffer twistTex;                  // twist-line texture: float4 texels, mip levels back to back
    uint32_t twistW = 0, twistH = 0, twistLevels = 0;
    LvDeviceBuffer flowOccupancy;             // max-helicity-first seeding: the occupancy grid, one byte per cell (grid-based check) / the list heads of the finished points' grid
    LvDeviceBuffer flowPoints, flowPointsNext; // point-based termination checks: finished points (12 B) and their per-cell list links
    size_t flowPointsCapacity = 0;
    LvDeviceBuffer flowSelfGrid;              // loop check "Grid": one bit per cell and line of a batch
    LvDeviceBuffer prismLeafList;             // raster_prism, sharded frames: segments that can touch the frame's tile list (k_ppll_cull_segments)
    LvDeviceBuffer ppllCoarse;                // raster_prism, sharded frames: 32 x 32-pixel cells that hold requested pixels (k_ppll_mark_tiles)
    LvDeviceBuffer ppllOverflow;              // raster_prism: pixel addresses with more kept fragments than ppllMaxNumFrags (k_ppll_pixel_pass)
    bool ppllArrays = false;                  // the last PPLL frame left per-pixel runs (raster_prism), not linked lists
    LvDeviceBuffer tilesDev, outDev, scratchRays, stackOverflow, mlatTrace;
    LvDeviceBuffer accum;                     // rgba8 of the previous accumulated frame (full viewport)
    uint32_t* pinned = nullptr;               // 64 B of pinned host memory for small read-backs (hipHostMalloc)
    LvDeviceBuffer buildArena;                // temporaries of the LBVH builds, kept between builds
    // dispatch order of the 64x64-pixel groups of the tile kernels (lv_group_order_prepare): [0] colour pass, [1] RTAO pass geometry
    struct GroupOrder {
        LvDeviceBuffer cost, order;
        uint32_t n = 0, tileW = 0, tileH = 0;
        uint64_t generation = ~0ull;
        bool active = false;                  // in use by the frame being queued
    } groupOrder[2];
    bool groupOrderSorted = false;            // this frame's k_group_order is queued
    uint64_t tilesGeneration = 0;             // bumped whenever tilesDev receives another list
    bool tilesCoverViewport = false;          // the uploaded tile list covers every pixel of the viewport (key: width, height, tile size)
    uint32_t tilesCoverKey[4] = {0, 0, 0, 0};
    std::vector<uint32_t> tilesHost;          // staging copy: caller's tile list is borrowed for the call only
    bool tilesUploaded = false;               // tilesDev holds tilesHost
    uint64_t ppllPoolNodes = 0;               // physical node slots of the pool (logical size + per-wave chunk slack)
    // viewport the per-frame buffers were last rendered with (read-backs copy these extents, not the current camera's)
    uint32_t aoW = 0, aoH = 0, ppllPaddedW = 0, ppllPaddedH = 0, accumW = 0, accumH = 0;

    // stats
    lv_stats stats;
    hipEvent_t ev[16];
    bool evCreated = false;
    bool evBuildValid = false, evFrameValid = false, evPhaseRecorded = false;
    bool evTriBuildValid = false;             // ev[14], ev[15] bracket the last triangle-LBVH build
    bool evLinePointsValid = false, evTessValid = false; // ev[4], ev[6]: the a2 kernels of lv_set_trajectories; ev[8], ev[9]: the last device tessellation
    int lastMode = 0;
    // per-kernel launch timers: ring of event pairs per kernel id (LV_KERNEL_*)
    static constexpr int kNumKernels = 8;
    static constexpr int kRing = 512;
    hipEvent_t evKernel[kNumKernels][2 * kRing];
    uint64_t kernelLaunches[kNumKernels] = {0, 0, 0, 0, 0, 0, 0, 0};
};

int lv_fail(lv_ctx* ctx, int code, const char* fmt, ...);
// width the capsule LBVH's leaf boxes are padded with: the band width for elliptic tubes (useRibbonNormals,
// LineDataFlow.cpp:2120-2126), the line width otherwise
// RTAO geometry of the frame: the reference's triangle tubes (VulkanRayTracedAmbientOcclusion traces nothing else) whenever the mesh
// is there, the analytic capsules of the colour pass otherwise / on request
inline bool lv_ao_triangle_tubes(const lv_ctx* ctx) {
    return ctx->opt.rtaoGeometry == 2 || (ctx->opt.rtaoGeometry == 0 && (ctx->triMeshSet || ctx->trajSet));
}
// the capsule roots in use: the reference's textbook form (RayIntersectionTestsVulkan.glsl:39-119) -- the default -- or the
// closest-approach form; "auto" leaves the reference's form only where this build traces AO rays against the analytic capsules
// (a mode the reference does not have: rays that start on a capsule need the stable roots)
inline bool lv_literal_intersection(const lv_ctx* ctx) {
    if (ctx->opt.intersectionForm != 0) return ctx->opt.intersectionForm == 2;
    return !(ctx->opt.useAmbientOcclusion && !ctx->opt.aoPrebaked && !lv_ao_triangle_tubes(ctx));
}
inline float lv_accel_width(const lv_ctx* ctx) {
    return (ctx->opt.useRibbons && ctx->opt.ellipticTubes) ? ctx->opt.bandWidth : ctx->opt.lineWidth;
}

#define LV_HIP(ctx, expr)                                                                          \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return lv_fail(ctx, LV_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

int lv_buf_reserve(lv_ctx* ctx, LvDeviceBuffer& b, size_t bytes);
// event pair of the next launch of kernel `id`
inline hipEvent_t lv_kernel_ev(lv_ctx* ctx, int id, int which) {
    return ctx->evKernel[id][2 * (ctx->kernelLaunches[id] % lv_ctx::kRing) + which];
}
#define LV_TIMED_LAUNCH(ctx, id, launch)                                      \
    do {                                                                      \
        const bool lvTimed = (((ctx)->opt.timerMask >> (id)) & 1u) != 0u;     \
        if (lvTimed) LV_HIP(ctx, hipEventRecord(lv_kernel_ev(ctx, id, 0), (ctx)->stream)); \
        launch;                                                               \
        if (lvTimed) {                                                        \
            LV_HIP(ctx, hipEventRecord(lv_kernel_ev(ctx, id, 1), (ctx)->stream)); \
            (ctx)->kernelLaunches[id]++;                                      \
        }                                                                     \
    } while (0)
void lv_buf_free(LvDeviceBuffer& b);

// lv_bvh.hip
int lv_bvh_build(lv_ctx* ctx);
int lv_bvh_build_triangles(lv_ctx* ctx);
// lv_lines.hip: (re)tessellates the tube mesh of lv_set_trajectories' lines if the line width / subdivision count changed; no-op otherwise
int lv_ensure_tube_mesh(lv_ctx* ctx);
// lv_multi.hip: one frame over the GPUs of a node behind the same handle
int lv_multi_create(lv_ctx* handle, const int* devices, int numDevices, const char* transport);
void lv_multi_destroy(lv_ctx* handle);
int lv_multi_num_ranks(const lv_ctx* handle);
lv_ctx* lv_multi_rank(lv_ctx* handle, int r);
int lv_multi_render(lv_ctx* handle, int mode, const uint32_t* tilesXY, uint32_t numTiles, uint32_t tileW, uint32_t tileH, bool image,
                    uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, void* outDevice);
int lv_multi_render_rect(lv_ctx* handle, int mode, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, void* outDevice);
int lv_multi_rebalance_impl(lv_ctx* handle, double baseCostPerTile);
int lv_multi_get_deal(lv_ctx* handle, uint32_t* outOwner, uint32_t capacity, uint32_t* outCount);
// repeats a setter on the other ranks of a multi-device handle (no-op for a single-device context)
template <class F>
inline int lv_forward_to_ranks(lv_ctx* handle, F&& f) {
    const int n = lv_multi_num_ranks(handle);
    for (int r = 1; r < n; r++) {
        lv_ctx* p = lv_multi_rank(handle, r);
        const int rc = f(p);
        if (rc) return lv_fail(handle, rc, "rank %d (device %d): %s", r, p->device, p->lastError.c_str());
    }
    // the setters select their rank's device: leave the calling thread on the handle's (outputs live on the first device)
    if (n > 1) (void)hipSetDevice(handle->device);
    return LV_OK;
}
// lv_mlat.hip
struct LvUniforms;
struct LvSceneDev;
struct LvTiles;
int lv_mlat_render(lv_ctx* ctx, const LvUniforms& U, const LvSceneDev& S, const LvTiles& T, uint32_t gridTiles,
                   uint32_t* out, LvDevCounters* dc, bool triangles);
// lv_render.hip
int lv_frame_render(lv_ctx* ctx, int mode, const uint32_t* tilesXYHost, uint32_t numTiles, uint32_t tileW,
                    uint32_t tileH, void* outDevice);
int lv_frame_trace_rays(lv_ctx* ctx, const float* o, const float* d, float tMin, float tMax, uint32_t n, float* outT,
                        uint32_t* outSeg, uint32_t* outKind);
int lv_frame_trace_rays_triangles(lv_ctx* ctx, const float* o, const float* d, float tMin, float tMax, uint32_t n,
                                  float* outT, uint32_t* outTri, float* outUV);
int lv_bake_ambient_occlusion(lv_ctx* ctx, bool async = false);
int lv_bake_poll(lv_ctx* ctx, bool wait);
// everything that changes what a baked table depends on: a running asynchronous bake still reads the old inputs, so it is waited for
// (and its result discarded: the generation no longer matches)
inline void lv_invalidate_bake(lv_ctx* ctx) {
    if (ctx->bakeAsyncPending) (void)hipEventSynchronize(ctx->evBakeDone);
    ctx->bakeValid = false;
    ctx->bakeGeneration++;
}
int lv_frame_depth_range(lv_ctx* ctx);
int lv_frame_ppll_resolve_only(lv_ctx* ctx, const uint32_t* nodes, uint64_t numNodes, const uint32_t* start,
                               uint64_t numPixels, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint8_t* out);
void lv_fill_uniforms(const lv_ctx* ctx, LvUniforms& U);
bool lv_ppll_prism_source(const lv_ctx* ctx);
// lv_svgf.hip
int lv_svgf_prepare(lv_ctx* ctx);
int lv_svgf_denoise(lv_ctx* ctx, const float* noisy);
// lv_flow.hip
int lv_flow_set_grid(lv_ctx* ctx, const float* vectorField, uint32_t xs, uint32_t ys, uint32_t zs, float dx, float dy,
                     float dz, const float* const* scalarFields, uint32_t numScalarFields);
int lv_flow_trace(lv_ctx* ctx, const float* seeds, uint32_t numSeeds, const lv_streamline_settings* settings);
int lv_flow_trace_max_helicity_first(lv_ctx* ctx, const float* helicityField, const lv_streamline_settings* settings,
                                     const lv_helicity_seeding_settings* seeding);
void lv_mat4_inverse(const float* m, float* inv);
void lv_mat4_mul(const float* A, const float* B, float* out);
