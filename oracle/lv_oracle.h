/*
 * lv_oracle.h -- C interface of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.  The product path
 * (linevis_amd/csrc, include/linevis_hip.h) never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference (chrismile/LineVis) cannot be built here (no
 * sgl / Vulkan / GLM) and none of its own tests touch this path, so there are
 * no reference-held golden vectors.  This oracle is a CPU restatement of the
 * reference's GLSL arithmetic; every function cites the file:line it follows.
 * Its own outputs are committed under tests/golden/ as regression pins.
 */
#ifndef LV_ORACLE_H
#define LV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors struct LinePointDataUnified, src/LineData/LineRenderData.hpp:99-106 (48 B). */
typedef struct {
    float linePosition[3];
    float lineAttribute;
    float lineTangent[3];
    float lineRotation;
    float lineNormal[3];
    uint32_t lineStartIndex;
} lvo_line_point;

/*
 * Everything the shaders read from uniform buffers, flattened:
 *   LineUniformData            Data/Shaders/Renderers/LineUniformData.glsl:24-70
 *   RayTracerSettingsBuffer    Data/Shaders/Renderers/RayTracing/TubeRayTracingHeader.glsl:33-42
 *   RTAO UniformsBuffer        Data/Shaders/AO/RTAO/VulkanRayTracedAmbientOcclusion.glsl:45-67
 *   PPLL UniformDataBuffer     Data/Shaders/Renderers/PPLL/LinkedListHeader.glsl:45-52
 * plus the preprocessor switches the host assembles (VulkanRayTracer.cpp:430-460,
 * LineData.cpp:1209-1256, LineRenderer.cpp:129-179).
 */
typedef struct {
    float view[16];          /* column-major, GLM layout */
    float proj[16];
    float fovY;
    float nearDist, farDist;
    uint32_t width, height;  /* full viewport */
    float background[4];
    float lineWidth;

    /* ray tracer (TubeRayTracingHeader.glsl:33-42; defaults VulkanRayTracer.hpp:137-144) */
    uint32_t maxDepthComplexity;
    uint32_t numSamplesPerFrame;
    uint32_t frameNumber;
    uint32_t useJitteredRays;          /* USE_JITTERED_RAYS, VulkanRayTracer.cpp:420-426 */
    uint32_t useDeterministicSampling; /* DETERMINISTIC_SAMPLING */

    /* shading switches */
    uint32_t useCappedTubes;  /* USE_CAPPED_TUBES */
    uint32_t useHalos;        /* USE_HALOS */
    uint32_t useDepthCues;    /* USE_DEPTH_CUES */
    uint32_t useAmbientOcclusion; /* USE_AMBIENT_OCCLUSION + GEOMETRY_PASS_TUBE */
    float depthCueStrength;
    float minDepth, maxDepth; /* DepthMinMaxBuffer, Lighting.glsl:28-33 */
    float aoStrength, aoGamma;

    /* transfer function range (MinMaxUniformBuffer, TransferFunction.glsl:60-63) */
    float attrMin, attrMax;

    /* RTAO (VulkanRayTracedAmbientOcclusion.hpp:108,150-153) */
    uint32_t aoSamplesPerFrame;
    uint32_t aoIterations;
    uint32_t aoUseDistance;
    uint32_t aoJitterPrimary;
    uint32_t tubeNumSubdivisions;
    float aoRadius;

    /* PPLL (PerPixelLinkedListLineRenderer.cpp:144-209,251-357) */
    uint32_t ppllMaxNumFrags;      /* MAX_NUM_FRAGS */
    uint32_t ppllLinkedListSize;   /* nodes in the pool */
    uint32_t ppllTileW, ppllTileH; /* LineRenderer.cpp:739-740 */

    /* band data (ribbons): USE_BANDS = useRibbons && hasBandsData (LineDataFlow.cpp:2423-2431); elliptic tubes = the ray
     * tracer's "Elliptic Tubes" switch (VulkanRayTracer.cpp:198-201,468-499); LineUniformData bandWidth / minBandThickness
     * (LineData.cpp:1297-1298); MIN_THICKNESS = minBandThickness with thick bands, 1e-2 otherwise */
    uint32_t useBands;
    uint32_t useEllipticTubes;
    float bandWidth, minBandThickness, minThickness;
    /* geometry_mode "Linear Swept Spheres": capsules with their caps whatever useCappedTubes says (it then only decides whether
     * the shading sees isCap), exact closest-approach roots (TubeRayTracing.glsl:621-737, LineData.cpp:909-945) */
    uint32_t lssGeometry;
    /* USE_ROTATING_HELICITY_BANDS (LineDataFlow.cpp:2432-2440; never together with USE_BANDS, :470,601-604): separator stripes
     * that rotate around the tube with lineRotation; LineUniformData separatorBaseWidth / helicityRotationFactor /
     * numSubdivisionsBands (LineDataFlow.cpp:979-984, defaults 0.2 / 1 / 6) */
    uint32_t useHelicityBands;
    uint32_t numSubdivisionsBands;
    float separatorBaseWidth, helicityRotationFactor;
    /* UNIFORM_HELICITY_BAND_WIDTH (use_uniform_twist_line_width, default on, LineDataFlow.cpp:53,2434-2436): separator width /
     * cos(atan(rotation per length x lineWidth / 2)).  Only the triangle closest-hit path computes rotationSeparatorScale
     * (LineAttributesBarycentric.glsl:94-112); ClosestHitTubeAnalytic does not pass it, so the analytic paths ignore the switch. */
    uint32_t uniformHelicityBandWidth;
    uint32_t ppllSortingMode;  /* SortingAlgorithmMode, src/Renderers/PPLL.hpp:41-50: 0 priority queue ... 7 quicksort hybrid */
    /* ppll_fragment_source: 0 = capsule_entry (entry hits of the pixel-centre ray against the analytic capsules: the probe of
     * rounds 1-3), 1 = raster_prism (the fragments of the rasterised N-gon prism of the default "Tube (Programmable Pull)" mode,
     * LinePassProgrammablePullTubes.glsl:87-224 + LineDataFlow.cpp:1698-1713; lv_oracle_prism.h) */
    uint32_t ppllFragmentSource;
} lvo_params;

typedef struct {
    uint64_t raysTraced;
    uint64_t nodesVisited;
    uint64_t primsTested;
    uint64_t hitsShaded;
    uint64_t fragments;       /* PPLL: fragCounter after gather */
    uint32_t maxDepthComplexity;
    uint32_t bvhDepth;
} lvo_stats;

typedef struct lvo_scene lvo_scene;

/* ---- a12: RNG (RayTracingUtilities.glsl:134-181) ---- */
uint32_t lvo_tea(uint32_t val0, uint32_t val1);
uint32_t lvo_lcg(uint32_t* state);
float lvo_rnd(uint32_t* state);
/* deterministic sin/cos of 2*pi*xi used by the hemisphere sample (definition owned by the build) */
void lvo_sincos_2pi(float xi, float* s, float* c);

/* ---- a1: normalisation (TrajectoryFile.cpp:106-125) ---- */
void lvo_normalize_positions(float* positions /* n*3 in/out */, uint64_t n);

/* ---- a2: LineDataFlow::getLinePassTubeAabbRenderData (LineDataFlow.cpp:2112-2277) ----
 * positions: concatenated xyz of all lines; attributes: one float per point;
 * lineOffsets[nLines+1]: start index of each line.
 * Outputs must be sized for the worst case (nPoints, 2*nPoints, 6*nPoints).
 * Returns number of output points via *outNumPoints and segments via *outNumSegments. */
/* useRotatingHelicityBands of the render-data builders (LineDataFlow.cpp:2188-2196, 2014-2027): while a source is set,
 * lineRotation of every emitted line point = the rotation accumulated so far, then += helicity / maxHelicity * PI * length of
 * the segment to the next trajectory point / 0.005.  helicities is indexed like the trajectory points handed to the builders
 * (tube AABB data: restarts per trajectory; triangle data: runs on across trajectories, as the reference's loop does).
 * NULL switches it off. */
void lvo_set_helicity_source(const float* helicities, float maxHelicity);
const float* lvo_get_helicity_source(float* maxHelicity);
void lvo_build_tube_aabb_render_data(
        const float* positions, const float* attributes, const uint32_t* lineOffsets, uint32_t nLines,
        float lineWidth,
        lvo_line_point* outPoints, uint32_t* outNumPoints,
        uint32_t* outSegIndices, float* outAabbs, uint32_t* outNumSegments);

/* ---- 4x4 inverse (cofactor expansion; stands in for glm::inverse, LineData.cpp:1290-1291) ---- */
void lvo_mat4_inverse(const float m[16], float out[16]);

/* ---- scene ---- */
lvo_scene* lvo_scene_create(const lvo_line_point* pts, uint32_t nPts, const uint32_t* segIdx, uint32_t nSeg);
void lvo_scene_destroy(lvo_scene*);
void lvo_scene_set_tf(lvo_scene*, const float* rgba, uint32_t n);
/* CPU LBVH (Morton order + highest-differing-bit splits); optional accelerator. */
void lvo_scene_build_bvh(lvo_scene*, float lineWidth);
void lvo_set_num_threads(int n);
/* normalize() calls of the shading code whose squared length fell outside [2^-60, 2^60] since the last reset -- where the build's
 * clamped rule is not the reference's normalize(); the parity tests assert 0 on the scenes they compare */
unsigned long long lvo_shade_normalize_out_of_range(int reset);

/* ---- a7 + closest hit: IntersectionTube (TubeRayTracing.glsl:452-494) over all segments ----
 * useBvh=0: brute force in ascending segment order (ground truth).
 * outKind: 0 tube, 1 sphere p0, 2 sphere p1; outSeg = 0xFFFFFFFF on miss. */
void lvo_trace_rays(
        const lvo_scene*, float lineWidth, int useCappedTubes, int useBvh,
        const float* origins, const float* dirs, float tMin, float tMax, uint32_t n,
        float* outT, uint32_t* outSeg, uint32_t* outKind);
/* single-capsule test, for known-answer vectors */
int lvo_intersect_capsule(
        const float o[3], const float d[3], const float p0[3], const float p1[3], float radius,
        int useCappedTubes, float* outT, int* outKind);

/* the reference's literal textbook-quadratic form (RayIntersectionTestsVulkan.glsl:39-119), kept for cross-checks */
int lvo_intersect_capsule_literal(
        const float o[3], const float d[3], const float p0[3], const float p1[3], float radius,
        int useCappedTubes, float* outT, int* outKind);

/* ---- a18: depth range (ComputeDepthValues.glsl:58-98, MinMaxReduce.glsl:64-103) ---- */
void lvo_compute_depth_range(const lvo_scene*, const lvo_params*, float outMinMax[2]);

/* ---- a13: RTAO (VulkanRayTracedAmbientOcclusion.glsl:178-319), capsule geometry ----
 * Fills aoOut[width*height] for the region [x0,x0+w) x [y0,y0+h). */
void lvo_render_ao(
        const lvo_scene*, const lvo_params*, int useBvh,
        uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, float* aoOut, lvo_stats* stats);

/* ---- a5,a6,a8-a11: ray tracer frame (TubeRayTracing.glsl RayGen + hit shaders) ----
 * ao may be NULL unless useAmbientOcclusion.  outRGBA8: w*h*4 tile, row-major. */
void lvo_render_rt(
        const lvo_scene*, const lvo_params*, int useBvh, const float* ao,
        uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint8_t* outRGBA8, lvo_stats* stats);

/* frame P->frameNumber of a multi-frame accumulation: mixed with the previous frame's rgba8 tile (NULL for frame 0) */
void lvo_render_rt_accumulate(
        const lvo_scene*, const lvo_params*, int useBvh, const float* ao,
        uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, const uint8_t* prevRGBA8, uint8_t* outRGBA8, lvo_stats* stats);

/* ---- f1: multi-layer alpha tracing (MlatInsert.glsl, traceRayMlat TubeRayTracing.glsl:86-192) ----
 * lvo_mlat_insert: one insertNodeMlat() on a node array of numNodes x {color[4], transmittance, depth}.
 * lvo_render_rt_mlat: the ray tracer frame with USE_MLAT.  traceOffsets == NULL: candidates visited in ascending
 * segment order; else replay of a recorded visiting order, validated (see lv_oracle.cpp). */
/* per pixel-centre ray of a tile: all capsule entry hits, ascending segment index (segs / ts may be NULL) */
void lvo_pixel_hits(const lvo_scene*, const lvo_params*, int useBvh, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                    uint64_t* offsets, uint32_t* segs, float* ts);
void lvo_mlat_insert(float* nodes, int numNodes, float* depth2, const float color[4], float depth, int missShader,
                     int* outAccepted);
void lvo_render_rt_mlat(
        const lvo_scene*, const lvo_params*, int useBvh, const float* ao,
        uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t numNodes,
        const uint64_t* traceOffsets, const uint32_t* traceSegs, const uint8_t* traceFlags,
        uint8_t* outRGBA8, float* outNodesOrNull, uint64_t* outViolations, lvo_stats* stats);

/* the same in the "Triangle Mesh" geometry mode: candidates = triangles of the tube mesh (brute force), ids = triangle indices */
struct lvo_tri_scene;
void lvo_render_rt_mlat_tri(
        const lvo_scene*, const struct lvo_tri_scene*, const lvo_params*, const float* ao,
        uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t numNodes,
        const uint64_t* traceOffsets, const uint32_t* traceTris, const uint8_t* traceFlags,
        uint8_t* outRGBA8, float* outNodesOrNull, uint64_t* outViolations, lvo_stats* stats);

/* ---- a15-a17: PPLL ---- */
uint32_t lvo_ppll_addr(uint32_t x, uint32_t y, uint32_t viewportWPadded, uint32_t tileW, uint32_t tileH);
/* gather: all-hits per pixel-centre ray, fragments appended in ascending segment order.
 * nodes: linkedListSize*3 uint32 {color, depthBits, next}; startOffset: paddedW*paddedH. */
void lvo_ppll_gather(
        const lvo_scene*, const lvo_params*, int useBvh, const float* ao,
        uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
        uint32_t* nodes, uint32_t* startOffset, uint32_t* fragCounter, lvo_stats* stats);
/* resolve one frame from given buffers; literal!=0 uses the reference's depth-only comparisons. */
void lvo_ppll_resolve(
        const lvo_params*, const uint32_t* nodes, const uint32_t* startOffset, int literal,
        uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint8_t* outRGBA8);
void lvo_render_ppll(
        const lvo_scene*, const lvo_params*, int useBvh, const float* ao,
        uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint8_t* outRGBA8, lvo_stats* stats);
/* a16 test hooks: ring vertices (position, normal; nPts * N * 3 floats each) of the programmable-pull vertex stage, and the
 * per-pixel fragments of the rasterised prism in ascending (segment, triangle) order (lv_oracle_prism.h) */
/* test hook: the twist-line texture of the rotating helicity bands (USE_HELICITY_BANDS_TEXTURE); rgba8 == NULL switches it off.
 * filterMode = index into LineDataFlow.cpp:55-57's names. */
void lvo_set_twist_line_texture(const uint8_t* rgba8, uint32_t width, uint32_t height, uint32_t filterMode);
void lvo_twist_line_sample(const float* u, const float* dudx, const float* dudy, int useGrad, uint64_t n, float* outRGBA);
void lvo_set_prism_ring_bands(int useBands, float thickness);
void lvo_prism_ring_vertices(const lvo_line_point* pts, uint64_t nPts, uint32_t numSubdivisions, float lineWidth, float* outPos,
                             float* outNormal);
void lvo_prism_coverage_dir(const lvo_params* P, uint32_t x, uint32_t y, float* coverageDir /* 3 */, float* rayDir /* 3 */);
void lvo_prism_fragments(const lvo_scene* sc, const lvo_params* P, int useBvh, const float* ao, uint32_t x0, uint32_t y0, uint32_t w,
                         uint32_t h, uint64_t* offsets, uint32_t* segs, uint32_t* tris, float* weights, float* depth, float* pos,
                         float* nrm, float* tan, float* attr, uint32_t* colour, float* rgba);

/* ---- a14: triangle tubes (CappedTriangleTubesCPU.cpp:214-383, Tubes.cpp:34-85, LineDataFlow.cpp:1912-2110) ---- */
/* Mirrors struct TubeTriangleVertexData, src/LineData/LineRenderData.hpp:171-176 (32 B). */
typedef struct {
    float vertexPosition[3];
    uint32_t vertexLinePointIndex; /* bit 31 set on cap vertices */
    float vertexNormal[3];
    float phi;
} lvo_tube_vertex;
/* Capped N-gon tubes of all lines (open tubes, hemisphere caps).  Output pointers may be NULL to query the sizes. */
/* createCappedTriangleEllipticTubesRenderDataCPU (CappedTriangleTubesCPU.cpp:387-745) + the line-point table: the triangle
 * tubes of a band data set (ribbonDirections: 3 floats per input point). */
void lvo_build_tube_triangle_render_data_ribbons(
        const float* positions, const float* attributes, const uint32_t* lineOffsets, uint32_t nLines,
        const float* ribbonDirections, float bandWidth, float minBandThickness, uint32_t tubeNumSubdivisions,
        uint32_t* outIndices, uint64_t* outNumIndices, lvo_tube_vertex* outVerts, uint64_t* outNumVerts,
        lvo_line_point* outPoints, uint64_t* outNumPoints);
void lvo_build_tube_triangle_render_data(
        const float* positions, const float* attributes, const uint32_t* lineOffsets, uint32_t nLines,
        float lineWidth, uint32_t tubeNumSubdivisions,
        uint32_t* outIndices, uint64_t* outNumIndices, lvo_tube_vertex* outVerts, uint64_t* outNumVerts,
        lvo_line_point* outPoints, uint64_t* outNumPoints);

typedef struct lvo_tri_scene lvo_tri_scene;
lvo_tri_scene* lvo_tri_scene_create(const uint32_t* indices, uint32_t nTri, const lvo_tube_vertex* verts,
                                    uint32_t nVerts, const lvo_line_point* pts, uint32_t nPts, float lineWidth);
void lvo_tri_scene_destroy(lvo_tri_scene*);
void lvo_tri_scene_build_bvh(lvo_tri_scene*);
/* the build's ray-triangle test (definition: lv_oracle_tri.cpp header); pad = padding of the triangle's own AABB */
int lvo_intersect_triangle(const float o[3], const float d[3], const float v0[3], const float v1[3],
                           const float v2[3], float pad, float* outT, float* outU, float* outV);
/* closest hit over all triangles; outTri = 0xFFFFFFFF on miss; outUV (2 floats per ray) may be NULL */
void lvo_trace_rays_tri(const lvo_tri_scene*, int useBvh, const float* origins, const float* dirs, float tMin,
                        float tMax, uint32_t n, float* outT, uint32_t* outTri, float* outUV);
/* ---- ray tracer colour pass in "Triangle Mesh" geometry mode (ClosestHitTubeTriangles, TubeRayTracing.glsl:301-352);
 * the capsule scene supplies the transfer function ---- */
void lvo_render_rt_tri(
        const lvo_scene*, const lvo_tri_scene*, const lvo_params*, int useBvh, const float* ao,
        uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint8_t* outRGBA8, lvo_stats* stats);
/* ---- §8f rank 2: static RTAO prebaking (VulkanAmbientOcclusionBaker.{cpp,glsl}, AmbientOcclusion.glsl:49-75) ---- */
void lvo_ao_parametrization(const float* positions, const uint32_t* lineOffsets, uint32_t nLines,
                            float expectedParamSegmentLength, float* outBlendingWeights, float* outSamplingLocations,
                            uint64_t* outNumParametrizationVertices);
void lvo_set_bake_bands(int useBands, float bandRadius, float minBandThickness);
void lvo_bake_ao(const lvo_scene*, const lvo_tri_scene* triSceneOrNull, float lineWidth, int useCappedTubes, int useBvh,
                 const float* samplingLocations, uint32_t numParametrizationVertices, uint32_t numTubeSubdivisions,
                 uint32_t numAmbientOcclusionSamples, uint32_t numIterations, float ambientOcclusionRadius, int useDistance,
                 float* outFactors);
void lvo_render_rt_prebaked(const lvo_scene*, const lvo_tri_scene* triSceneOrNull, const lvo_params*, int useBvh,
                            const float* factors, const float* blendingWeights, uint32_t numLineVertices,
                            uint32_t numParametrizationVertices, uint32_t numAoTubeSubdivisions, uint32_t x0,
                            uint32_t y0, uint32_t w, uint32_t h, uint8_t* outRGBA8, lvo_stats* stats);
/* ---- a13 with the reference's own geometry: RTAO against the triangle tubes ---- */
void lvo_render_ao_tri(
        const lvo_tri_scene*, const lvo_params*, int useBvh,
        uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, float* aoOut, lvo_stats* stats);

/* ---- §8f: streamline tracing on a regular grid (StreamlineTracingGrid.cpp, see lv_oracle_flow.cpp) ---- */
typedef struct {
    uint32_t integrationMethod;     /* 0 explicit Euler, 2 Heun, 3 midpoint, 4 RK4 (StreamlineTracingDefines.hpp:63-76) */
    uint32_t integrationDirection;  /* 0 forward, 1 backward, 2 both */
    float timeStepScale;
    int32_t maxNumIterations;
    float terminationDistance;
    float minimumLength;
} lvo_streamline_settings;
typedef struct lvo_streamlines lvo_streamlines;
void lvo_generate_abc_flow(float* v, int xs, int ys, int zs, float A, float B, float C, float resScale);
float lvo_max_vector_magnitude(const float* v, uint64_t numCells);
/* max-helicity-first seeding (StreamlineMaxHelicityFirstSeeder + _traceStreamribbonsDecreasingHelicity), sequential restatement */
lvo_streamlines* lvo_trace_streamlines_max_helicity_first(
        const float* vectorField, int xs, int ys, int zs, float dx, float dy, float dz, const float* const* scalarFields,
        uint32_t numScalarFields, const float* helicityField, const lvo_streamline_settings* settings, float minimumSeparationDistance,
        uint32_t loopCheckMode, float terminationDistanceSelf, int seedingSubsamplingFactor);
/* ... with StreamlineTracingSettings::terminationCheckType: 0 naive, 1 grid-based, 2 k-d tree-based, 3 hashed grid-based */
lvo_streamlines* lvo_trace_streamlines_max_helicity_first_ex(
        const float* vectorField, int xs, int ys, int zs, float dx, float dy, float dz, const float* const* scalarFields,
        uint32_t numScalarFields, const float* helicityField, const lvo_streamline_settings* settings, float minimumSeparationDistance,
        uint32_t loopCheckMode, float terminationDistanceSelf, int seedingSubsamplingFactor, uint32_t terminationCheckType);
void lvo_set_streamribbon_termination_check_type(uint32_t terminationCheckType); /* of the following streamribbon calls; default 1 */
lvo_streamlines* lvo_trace_streamribbons_max_helicity_first(
        const float* vectorField, int xs, int ys, int zs, float dx, float dy, float dz, const float* const* scalarFields,
        uint32_t numScalarFields, const float* helicityField, const lvo_streamline_settings* settings, float minimumSeparationDistance,
        uint32_t loopCheckMode, float terminationDistanceSelf, int seedingSubsamplingFactor, int useHelicity, float maxHelicityTwist,
        const float* initialRibbonDirection);
lvo_streamlines* lvo_trace_streamlines(const float* vectorField, int xs, int ys, int zs, float dx, float dy, float dz,
                                       const float* const* scalarFields, uint32_t numScalarFields, const float* seeds,
                                       uint32_t numSeeds, const lvo_streamline_settings* settings);
void lvo_streamlines_sizes(const lvo_streamlines*, uint64_t* numLines, uint64_t* numPoints);
void lvo_streamlines_copy(const lvo_streamlines*, float* positions, float* attributes, uint32_t* offsets);
void lvo_streamlines_destroy(lvo_streamlines*);
/* traceStreamribbons (StreamlineTracingGrid.cpp:428-530,1049-1116): the streamlines + one ribbon direction per point, carried
 * along every traced part and twisted by the local helicity (scalar field `helicityFieldIndex`). */
lvo_streamlines* lvo_trace_streamribbons(const float* vectorField, int xs, int ys, int zs, float dx, float dy, float dz,
                                         const float* const* scalarFields, uint32_t numScalarFields, const float* seeds,
                                         uint32_t numSeeds, const lvo_streamline_settings* settings, uint32_t helicityFieldIndex,
                                         int useHelicity, float maxHelicityTwist, const float* initialRibbonDirection);
void lvo_streamlines_copy_ribbons(const lvo_streamlines*, float* ribbonDirections);
/* GridLoader.cpp:41-183 */
void lvo_compute_vector_magnitude_field(const float* v, float* out, int xs, int ys, int zs);
void lvo_compute_vorticity_field(const float* v, float* out, int xs, int ys, int zs, float dx, float dy, float dz);
void lvo_compute_helicity_field_normalized(const float* vel, const float* vort, float* out, int xs, int ys, int zs,
                                           int normalizeVelocity, int normalizeVorticity);

/* ---- f4: EAW denoiser of the RTAO pass (EAWDenoiser.cpp, EAWDenoise.glsl; AO defaults Denoiser.cpp:54-62) ----
 * Feature maps: full-viewport float4 images the next lvo_render_ao / lvo_render_ao_tri calls fill (view-space normal {xyz,0},
 * view-space position {xyz,1}; VulkanRayTracedAmbientOcclusion.glsl:321-399); NULL stops writing. */
void lvo_set_ao_feature_outputs(float* normalMap, float* positionMap);
/* `iterations` a-trous passes (step width 1, 2, 4, ...) over the AO image; phi* already multiplied by their scales
 * (AO mode: 0.49, 0.3 * 1e-4, 0.1); computeVariant != 0 = EAWDenoise.Compute (default), 0 = EAWDenoise.Fragment.
 * Computes the pixels of the rectangle; ao / maps / out are full-viewport images. */
void lvo_eaw_denoise(uint32_t width, uint32_t height, const float* ao, const float* normalMap, const float* positionMap,
                     int iterations, float phiColor, float phiPosition, float phiNormal, int useColor, int usePosition,
                     int useNormal, int computeVariant, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, float* out);

/* Test hooks: the build-owned sin / cos / atan2 of the elliptic-tube path */
void lvo_sincos_rad(float a, float* s, float* c);
float lvo_atan2_det(float y, float x);
/* test hook: the USE_BANDS halo coordinate of computeFragmentColor (RayHitCommon.glsl:232-351) for a fragment at angle phi on
 * the cross-section x^2 / thickness^2 + y^2 = 1 (units of lineRadius, x along the line normal) seen from cam */
float lvo_bands_ribbon_position(const float cam[3], const float linePos[3], const float lineNormal[3], const float tangent[3],
                                float phi, float lineRadius, float thickness);
/* Band data: getLinePassTubeAabbRenderData(false, true) -- normals from the ribbon directions (3 floats per input point), boxes
 * padded by bandWidth / 2 (LineDataFlow.cpp:2112-2277). */
void lvo_build_tube_aabb_render_data_ribbons(
        const float* positions, const float* attributes, const uint32_t* lineOffsets, uint32_t nLines, float bandWidth,
        const float* ribbonDirections, lvo_line_point* outPoints, uint32_t* outNumPoints, uint32_t* outSegIndices,
        float* outAabbs, uint32_t* outNumSegments);
/* Closest hit of n rays on the elliptic tubelets (EllipticTubeRayTracing.glsl IntersectionEllipticTube); miss: outT = tMax,
 * outSeg = 0xFFFFFFFF.  With useBvh the scene's BVH must have been built with lineWidth = bandWidth. */
void lvo_trace_rays_elliptic(const lvo_scene* sc, float bandWidth, float minBandThickness, const float* cameraPosition, int useBvh,
                             const float* origins, const float* dirs, float tMin, float tMax, uint32_t n, float* outT,
                             uint32_t* outSeg);

/* SVGF (ambient_occlusion_denoiser = "SVGF"): while `enable` != 0 the RTAO passes run without running means and seed from
 * globalFrameNumber (+ iteration), and fill the full-viewport maps normalWorld (float4), depth (float), flow (float2),
 * depthFwidth (float) (VulkanRayTracedAmbientOcclusion.glsl:350-464, DISABLE_ACCUMULATION branches). */
void lvo_set_svgf_feature_outputs(float* normalWorld, float* depth, float* flow, float* depthFwidth, int enable,
                                  uint32_t globalFrameNumber, const float* lastFrameViewProj);
/* out = A * B, column-major 4x4, glm's evaluation order */
void lvo_mat4_mul(const float* A, const float* B, float* out);
/* One SVGFDenoiser::denoise() (SVGF.glsl / SVGF.cpp): reproject, filter moments, `iterations` a-trous passes; the four
 * history images (colour: float, moments: float4 {m1, m2, history length, 0}, normal: float4, depth: float; zero before the
 * first call) are read and updated in place; out = the denoised AO image. */
void lvo_svgf_denoise(uint32_t width, uint32_t height, const float* noisy, const float* normalMap, const float* depthMap,
                      const float* depthFwidthMap, const float* flowMap, int iterations, float allowedZDist,
                      float allowedNormalDist, float* colorHistory, float* momentsHistory, float* normalHistory,
                      float* depthHistory, float* out);

/* Test hook: the restatement of computeFragmentColor + blinnPhongShadingTube on n independent inputs (n x 3 positions /
 * normals / tangents, n flags / attributes / AO texels) -> n x 4 colours, n payload.hitT. */
void lvo_compute_fragment_color_batch(const lvo_scene*, const lvo_params*, uint64_t n, const float* fragPos, const float* normal,
                                      const float* tangent, const uint32_t* isCap, const float* attribute, const float* aoTexel,
                                      float* outColor, float* outHitT);

/* Deviation switches (tests only): evaluate the reference's literal ray-capsule roots (RayIntersectionTestsVulkan.glsl:
 * 39-119) and / or its literal AO lookup (AmbientOcclusion.glsl:84-99: project + bilinear texture()) in the capsule ray
 * tracer paths (lvo_trace_rays, lvo_render_ao, lvo_render_rt, ...), to measure the documented deviations on whole frames.
 * Process-global; set between render calls. */
void lvo_set_deviation_switches(int literalIntersection, int referenceAoLookup);
/* 1: shade PPLL fragments with the ray tracer's computeFragmentColor (RayHitCommon.glsl; rounds 1-2), 0 (default): with the raster
 * tube shader's variant (LinePassGeometryShaderTubes.glsl:785-815,1079-1087) */
void lvo_set_ppll_fragment_colour_variant(int rayTracerVariant);
/* ambient_occlusion_mode = "RTAO (Prebaker)" in the PPLL gather: while a table is set (factors != NULL; the arrays must stay alive)
 * the fragments are shaded with getAoFactor(fragmentVertexId, phi) (AmbientOcclusion.glsl:49-75) -- raster_prism: the interpolated
 * vertex-stage outputs (lv_oracle_prism.h prismAoInputs), capsule_entry: the closest-hit reconstruction of the ray tracer. */
void lvo_set_ppll_prebaked_ao(const float* factors, const float* blendingWeights, uint32_t numLineVertices,
                              uint32_t numParametrizationVertices, uint32_t numAoTubeSubdivisions);
/* the build-owned pow of the shading code (powDet = lv_pow_det of the HIP library) on n inputs */
void lvo_pow_det(const float* x, const float* y, uint64_t n, float* out);
void lvo_prebaked_ao_lookup_batch(const float* factors, const float* blendingWeights, uint32_t numLineVertices,
                                  uint32_t numParametrizationVertices, uint32_t numAoTubeSubdivisions, const float* vertexId,
                                  const float* phi, uint64_t n, float* out);
void lvo_compute_fragment_color_raster_batch(const lvo_scene* sc, const lvo_params* P, uint64_t n, const float* fragPos,
                                             const float* normal, const float* tangent, const uint32_t* isCap, const float* attribute,
                                             const float* aoTexel, const float* epsWhite, float* outColor, float* outHitT);
void lvo_ribbon_of_rays(const float* cam, const float* dirs, uint64_t n, const float* axisPoint, const float* axisDir, float radius,
                        const float* capHit, const float* capNormal, float* out);

/* threads the OpenMP loops run on */
int lvo_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
