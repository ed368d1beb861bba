// lv_device.h -- shared device-side definitions of the HIP hot path (gfx950 only).
//
// Float32 arithmetic with a fixed evaluation order (the library is compiled with -ffp-contract=off): +,-,*,/ and
// sqrt are IEEE-exact on gfx950, so ray generation, traversal and ray-capsule intersection produce bit-identical
// (t, segment, kind) triples to any host evaluation of the same formulas in the same order.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/linevis_hip.h"

#define LV_WAVE 64
#define LV_BLOCK 256
#define LV_LEAF_BIT 0x80000000u
#define LV_INVALID 0xFFFFFFFFu
// per-thread traversal stack entries staged in LDS.  24 entries = 24 KB per workgroup: the tile kernels then need 40 KB
// (k_ppll_gather 50 KB) and 4 (3) workgroups fit a CU instead of 3 (2) -- k_ao_primary -10 %, k_ppll_gather -11 % against 32 entries;
// 14 entries fit one more but pay it back in overflow traffic (k_ppll_gather +5 %)
#ifndef LV_STACK_LDS
#define LV_STACK_LDS 24
#endif
#define LV_STACK_SPILL 64   // further entries in a global overflow slab; 96 >= max LBVH height (63 key bits + 32)
// minimum waves per SIMD the register allocator must leave room for (hipcc's second __launch_bounds__ argument)
// k_render_rt: 4 waves per SIMD (<= 128 VGPRs instead of 146; with the 40-KB workgroups that is 4 workgroups per CU): -6 % on
// config 3's colour pass, -4.5 % on config 2.  The instrumented and the elliptic-tube instantiations keep their registers.
#ifndef LV_RT_MIN_WAVES
#define LV_RT_MIN_WAVES 4
#endif
#ifndef LV_GATHER_MIN_WAVES
#define LV_GATHER_MIN_WAVES 1
#endif
#ifndef LV_NODE_MIX
#define LV_NODE_MIX 0            // node step: child planes as binary16 halves through v_perm_b32 + v_fma_mix_f32 (lv_slab_h)
#endif
#ifndef LV_PRISM_MIN_WAVES
#define LV_PRISM_MIN_WAVES 3    // k_ppll_gather<LV_PRIM_PRISM>: waves per SIMD the register allocator leaves room for
#endif
// PPLL node {0, LV_PPLL_DEAD, next}: a linked node whose fragment the fragment stage of the rasterised prism discarded (or a slot
// of a chunk tail).  The depth word of a real node is length(fragmentPositionWorld - cameraPosition): never this NaN pattern.
#define LV_PPLL_DEAD 0xFFFFFFFFu
#ifndef LV_PRISM_SHADE_MIN_WAVES
#define LV_PRISM_SHADE_MIN_WAVES 2   // k_ppll_shade_prism: waves per SIMD the register allocator leaves room for (4: 128 VGPRs + 248 B of scratch, 0.64 ms on config 4; 1 ... 3: no scratch, 0.49 ms)
#endif
#ifndef LV_PRISM_SHADE_BLOCKS_PER_CU
#define LV_PRISM_SHADE_BLOCKS_PER_CU 8   // grid of the grid-stride fragment stage
#endif
#ifndef LV_REFILL_THRESHOLD
#define LV_REFILL_THRESHOLD 8 // persistent AO waves fetch new rays once this many lanes are idle
#endif
#ifndef LV_AO_CHUNK
#define LV_AO_CHUNK 128         // AO rays a wave takes from the global queue per atomic (1024: -10 %, the last chunks' tail)
#endif
#ifndef LV_AO_CHUNK_SMALL
#define LV_AO_CHUNK_SMALL 64u   // ... when the launch has fewer than LV_AO_SMALL_CHUNKS chunks of LV_AO_CHUNK per persistent wave
#endif
#ifndef LV_AO_SMALL_CHUNKS
#define LV_AO_SMALL_CHUNKS 12u
#endif
#ifndef LV_AO_STAY
#define LV_AO_STAY 1            // k_ao_rays descend loop: stay while at least this many lanes descend (tools/variants.py experiment)
#endif
#ifndef LV_AO_ORDERED
#define LV_AO_ORDERED 1         // k_ao_rays: 1 = nearest hit child first, 0 = children as stored (tools/variants.py experiment)
#endif
#ifndef LV_SORT_CHILDREN
#define LV_SORT_CHILDREN 0       // 1: fully sort the hit children of a node; 0: nearest first, rest unordered
#endif
#ifndef LV_AO_STACK_LDS
#define LV_AO_STACK_LDS 15      // LDS-staged stack entries per thread in k_ao_rays (deeper entries: HBM overflow slab)
#endif
#ifndef LV_PRECOMP_AXIS
#define LV_PRECOMP_AXIS 1
#endif
#ifndef LV_AO_MIN_WAVES
#define LV_AO_MIN_WAVES 5         // waves per SIMD k_ao_rays is compiled for (95 VGPRs; 6 would need <= 85 and 26 KB of LDS)
#endif
#ifndef LV_AO_BLOCK
#define LV_AO_BLOCK 256         // threads per workgroup of k_ao_rays
#endif
#ifndef LV_AO_BLOCKS_PER_CU
#define LV_AO_BLOCKS_PER_CU 5
#endif
#define LV_AO_QCAP 128         // leaf FIFO entries per wave (<= 63 waiting + 64 new per step)
#ifndef LV_AO_TEST_BATCH
#define LV_AO_TEST_BATCH 56u      // k_ao_rays: queued leaf tests that end a descend stint (r03: 44 / 48 / 52 / 56 / 64 = 4.39 / 4.34 / 4.31 / 4.30 / 4.35 ms)
#endif
#ifndef LV_AO_TEST_BATCH_TRI
#define LV_AO_TEST_BATCH_TRI 56u  // ... on the triangle tubes (5.20 / 5.18 / 5.17 / 5.16 / 5.23 ms)
#endif
#ifndef LV_TRACE_TEST_BATCH
#define LV_TRACE_TEST_BATCH 64u     // lv_trace_closest (tile kernels): queued leaf tests that end a descend stint
#endif
#ifndef LV_TRACE_TEST_BATCH_ALL
#define LV_TRACE_TEST_BATCH_ALL 64u // lv_trace_all (PPLL gather, MLAT)
#endif
#ifndef LV_NODE_MIN_ACTIVE
#define LV_NODE_MIN_ACTIVE 24  // node loop yields to the leaf loop when fewer lanes than this are descending
#endif

#ifndef LV_HANDOVER_MAX_BUSY
#define LV_HANDOVER_MAX_BUSY 32 // cooperative closest hit: idle lanes take over stacked subtrees once at most this many lanes descend
#endif
#ifndef LV_HANDOVER_MAX_BUSY_ALL
#define LV_HANDOVER_MAX_BUSY_ALL 48 // same for the all-hits walk of the PPLL gather (r03: 40...52 = -5 % on config 4 against 32)
#endif
#ifndef LV_HANDOVER_MAX_BUSY_DYN
#define LV_HANDOVER_MAX_BUSY_DYN 24 // ... and for the front-to-back ordered MLAT walk, where early hand-over shades fragments the closing interval would have culled (r03: 8 / 16 / 24 / 32 / 48 = 1.448 / 1.436 / 1.430 / 1.485 / 1.696 ms)
#endif

#ifndef LV_RESOLVE_LDS_MAX
#define LV_RESOLVE_LDS_MAX (64 * 1024) // k_ppll_resolve: per-wave fragment arrays up to this size live in LDS, larger ones in a global slab
#endif
#ifndef LV_RESOLVE_SLAB_GRID
#define LV_RESOLVE_SLAB_GRID 4096u      // ... worked through by this many persistent one-wave workgroups
#endif
#ifndef LV_PPLL_CHUNK
#define LV_PPLL_CHUNK 256       // PPLL node slots a wave reserves per global atomic (>= 64)
#endif
#ifndef LV_PPLL_SLICES
#define LV_PPLL_SLICES 1        // depth slices per pixel block in k_ppll_gather (workgroups = pixel blocks x slices)
#endif

// device-side counters block of a frame (read back by lv_get_stats / lv_ppll_get_buffers)
struct LvDevCounters {
    unsigned long long rays, nodes, prims, hits;
    unsigned long long aoRays, aoNodes, aoPrims; // share of k_ao_rays
    unsigned long long aoQueueHead;              // next AO ray index handed to the persistent waves
    unsigned long long aoPhaseIters[3], aoPhaseLanes[3]; // {setup, node, leaf}: wave iterations / active lanes
    uint32_t fragCounter;
    uint32_t aoCount;
    uint32_t maxDepthComplexity;
    uint32_t depthOrd[2]; // encoded min / max for the depth-range reduction
    uint32_t maxNodesPerPixel;
    uint32_t fragAlloc;   // PPLL node-slot allocator (chunks); fragCounter stays the exact fragment count
    uint32_t prismDiscards;  // raster_prism: linked nodes the fragment stage turned into dead nodes (discarded fragments)
    uint32_t mlatTraceCount; // records appended to the MLAT visiting-order trace (collect_stats)
    uint32_t ppllOverflowPixels; // raster_prism: pixels with more kept fragments than the sort arrays (k_ppll_pixel_pass's list)
    uint32_t prismListCount;     // raster_prism on a tile list: segments k_ppll_cull_segments kept for this rank's tiles
    // k_ao_rays leaf-test diagnostics (collect_stats): tests that found a hit inside the interval, tests the conservative
    // axis-distance pre-test lets through, tests axis + bounding-sphere pre-tests let through
    unsigned long long aoPrimHits, aoPrimMayAxis, aoPrimMayBoth;
};

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// GLSL cross(x, y) = (x1*y2 - y1*x2, x2*y0 - y2*x0, x0*y1 - y0*x1)
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
    return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
__device__ __forceinline__ float len3(f3 a) { return sqrtf(dot3(a, a)); }
__device__ __forceinline__ f3 norm3(f3 a) { float l = len3(a); return mk3(a.x / l, a.y / l, a.z / l); }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
__device__ __forceinline__ float mixf(float a, float b, float w) { return a * (1.0f - w) + b * w; }
__device__ __forceinline__ float smoothstepf(float e0, float e1, float x) {
    float t = clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
// column-major mat4 * vec4, columns summed left to right
__device__ __forceinline__ f4 mulM4(const float* m, float x, float y, float z, float w) {
    f4 r;
    r.x = ((m[0] * x + m[4] * y) + m[8] * z) + m[12] * w;
    r.y = ((m[1] * x + m[5] * y) + m[9] * z) + m[13] * w;
    r.z = ((m[2] * x + m[6] * y) + m[10] * z) + m[14] * w;
    r.w = ((m[3] * x + m[7] * y) + m[11] * z) + m[15] * w;
    return r;
}

// Per-frame constants: LineUniformData (LineUniformData.glsl:24-70) + RayTracerSettingsBuffer
// (TubeRayTracingHeader.glsl:33-42) + RTAO UniformsBuffer (VulkanRayTracedAmbientOcclusion.glsl:45-67) + PPLL
// UniformDataBuffer (LinkedListHeader.glsl:45-52) + the preprocessor switches, flattened into one kernel argument.
struct LvUniforms {
    float view[16], proj[16], invView[16], invProj[16];
    float camPos[3];
    float fovY;
    float background[4], foreground[4];
    float lineWidth, radius, nearDist, farDist;
    uint32_t width, height;
    uint32_t maxDepthComplexity, numSamplesPerFrame, frameNumber, useJitteredRays, useDeterministicSampling;
    uint32_t useCappedTubes, useHalos, useDepthCues, useAmbientOcclusion;
    float depthCueStrength, aoStrength, aoGamma, attrMin, attrMax;
    uint32_t tfN;
    uint32_t aoSamplesPerFrame, aoUseDistance, aoJitterPrimary, aoFrameNumber;
    // seeds of the RTAO pass: = aoFrameNumber, except under SVGF (useGlobalFrameNumber: a counter that camera moves do not
    // reset, while aoFrameNumber stays 0 = DISABLE_ACCUMULATION), VulkanRayTracedAmbientOcclusion.cpp:415-421,576-581
    uint32_t aoGlobalFrameNumber;
    float aoRadius, subdivisionCorrectionFactor;
    uint32_t ppllMaxNumFrags, ppllLinkedListSize, ppllTileW, ppllTileH, ppllPaddedW, ppllPaddedH;
    uint32_t ppllSortingMode; // SortingAlgorithmMode, src/Renderers/PPLL.hpp:41-50 (0 = priority queue)
    // static RTAO prebaking (STATIC_AMBIENT_OCCLUSION_PREBAKING, AmbientOcclusion.glsl:29-38)
    uint32_t aoPrebaked, bakeNumLineVertices, bakeNumParametrizationVertices, bakeNumTubeSubdivisions;
    // getAoFactor of the colour pass (AmbientOcclusion.glsl:84-99): 1 = project the hit and sample the AO image bilinearly
    // (jittered primary rays), 0 = the launching pixel's own texel (pixel-centre rays project onto their texel centre)
    uint32_t aoProjectLookup;
    // band data (ribbons): USE_BANDS = use_ribbons && band data (LineDataFlow.cpp:2423-2431); the ray tracer's "Elliptic Tubes"
    // switch (VulkanRayTracer.cpp:198-201,468-499); bandWidth / minBandThickness of LineUniformData (LineData.cpp:1297-1298);
    // minThickness = the MIN_THICKNESS define (minBandThickness with thick bands, 1e-2 otherwise)
    // geometry_mode "Linear Swept Spheres" (VK_NV_ray_tracing_linear_swept_spheres, chained end caps, LineData.cpp:909-945): the
    // hardware primitive is the exact union of capsules -- the colour pass traces the capsules with their caps whatever
    // use_capped_tubes says (which then only decides whether the shading sees isCap) and with the exact closest-approach roots
    uint32_t lssGeometry;
    uint32_t useBands, useEllipticTubes;
    float bandWidth, minBandThickness, minThickness;
    // USE_ROTATING_HELICITY_BANDS + LineUniformData numSubdivisionsBands / separatorBaseWidth / helicityRotationFactor
    // (LineDataFlow.cpp:979-984,2432-2440)
    uint32_t useHelicityBands, numSubdivisionsBands;
    float separatorBaseWidth, helicityRotationFactor;
    // PPLL gather: 1 = fragments are shaded by the raster tube shader's variant (LinePassGeometryShaderTubes.glsl:785-815,1079-1087:
    // EPSILON_OUTLINE = 0, EPSILON_WHITE = fwidth(ribbonPosition)), 0 = by RayHitCommon's (ppll_fragment_colour = ray_tracer)
    uint32_t ppllRasterColour;
    // USE_HELICITY_BANDS_TEXTURE (LineDataFlow.cpp:93-171,2437-2439): the twist-line texture replaces the separator stripes; filter
    // mode index of textureFilteringModeNames (:55-60), extent of level 0, number of mip levels (intlog2(max(w, h)), :161-163)
    uint32_t useTwistTexture, twistFilterMode, twistW, twistH, twistLevels;
    uint32_t uniformHelicityBandWidth; // UNIFORM_HELICITY_BAND_WIDTH: triangle closest-hit path only (LineAttributesBarycentric.glsl:94-112)
};

// Feature maps SVGF asks the RTAO pass for (SVGF.cpp:88-96; VulkanRayTracedAmbientOcclusion.glsl:350-464, DISABLE_ACCUMULATION
// branches), two float4 images: {world-space normal, depth} and {flow.xy, depth fwidth, 0}.  normalDepth == nullptr: not written.
struct LvSvgfFeat {
    float4* normalDepth;
    float4* flowFwidth;
    float lastFrameViewProj[16];
};

// Per-frame constants of the rasterised programmable-pull prism (ppll_fragment_source = raster_prism, lv_prism.h): cos / sin of the
// ring angles circleIdx / N * 2 pi (LinePassProgrammablePullTubes.glsl:129-131; lv_sincos2pi of circleIdx / N, filled on the host by
// the same formula), the camera's right axis (column 0 of inverse(viewMatrix)) that spans the ray-space basis, row 2 of the view
// matrix + the clip distances (depth clipping of the fragments)
#define LV_SCAN_ITEMS 4096u      // pixel addresses per workgroup of k_ppll_scan (run offsets are relative to these blocks)
#define LV_PRISM_MAX_SUBDIV 16
struct LvPrismDev {
    float c[LV_PRISM_MAX_SUBDIV], s[LV_PRISM_MAX_SUBDIV];
    // USE_BANDS ("bands with minimum thickness", LinePassProgrammablePullTubes.glsl:112-116,166-171): the ring is the ellipse
    // localPosition = (thickness cos, sin, 0) with localNormal = (cos, thickness sin, 0) and lineRadius = bandWidth / 2; cp = thickness *
    // cos feeds the positions, sn = thickness * sin the normals (thickness 1 for plain tubes: cp == c, sn == s bit for bit)
    float cp[LV_PRISM_MAX_SUBDIV], sn[LV_PRISM_MAX_SUBDIV];
    float radius, thickness;
    uint32_t bands;
    float right[3];
    float viewZ[4];
    float nearDist, farDist;
    uint32_t n;
    // coverage direction of a pixel (lv_prism_cov_dir, lv_prism.h): D = covC0 + (x + 0.5) covCx + (y + 0.5) covCy
    float covC0[3], covCx[3], covCy[3];
};

// HBM-resident scene (all read-only during rendering)
struct LvSceneDev {
    const float4* nodes;        // 64-B compressed 4-wide LBVH nodes, 4 x float4 each (layout: lv_bvh.hip k_pack4)
    const float4* segs;         // 32-B segment records in Morton (leaf) order: {p0.xyz, attr0}, {p1.xyz, attr1}
    const float4* segAxis;      // {normalize(p1 - p0), 0} per leaf: the tube axis of the capsule test, computed at build time
    const uint32_t* leafSeg;    // leaf position -> original segment index
    const uint32_t* segToLeaf;  // original segment index -> leaf position
    const lv_line_point* points;// 48-B point records, input order
    uint32_t numPoints;         // entries of `points`
    const uint32_t* segIdx;     // 2 point indices per original segment
    const float4* tf;           // transfer function texels
    const float4* twistTex;     // twist-line texture: float RGBA, mip levels one after the other (lv_set_twist_line_texture)
    const float* depthMinMax;   // {minDepth, maxDepth}, produced on device by the depth-range kernels
    // elliptic tubelets (LV_PRIM_ELLIPTIC): semi-axes from bandWidth / minBandThickness, camera position of the cutting-plane
    // tolerances (EllipticTubeRayTracing.glsl:186-270)
    float ellBandWidth, ellMinBandThickness, ellCamPos[3];
    const float* ao;            // full-viewport AO factors
    unsigned* stackOverflow;    // null unless the LBVH is higher than LV_STACK_LDS
    uint32_t* accum;            // full-viewport rgba8 of the previous frame (num_accumulated_frames > 1), else null
    uint32_t numSegs;           // primitives under `nodes` (segments, or triangles in a triangle-tube scene view)
    uint32_t literalIntersection; // intersection_form = literal: the reference's textbook roots (lv_intersect_capsule_literal)
    // triangle tubes (the reference's RTAO geometry); in the scene view handed to the triangle kernels `nodes` is the
    // triangle LBVH and numSegs the triangle count
    const float4* tris;         // 48-B records in Morton order: {v0.xyz, triangle index bits}{v1.xyz, 0}{v2.xyz, 0}; or, with triPairs,
                                // 64-B pair records {q0.xyz, index}{q1.xyz, code}{q2.xyz, 0}{q3.xyz, 0} (k_tri_leaves<true>, lv_bvh.hip)
    const uint32_t* triIdx;     // 3 vertex indices per triangle, input order
    const lv_tube_vertex* triVerts; // 32-B TubeTriangleVertexData, input order
    const lv_line_point* triPoints; // line points referenced by the vertices
    float triPad;               // padding of a triangle's own AABB (part of the ray-triangle test definition)
    uint32_t triLeafSize;       // triangle records per leaf of the triangle LBVH (lv_bvh_build_triangles)
    uint32_t triPairs;          // leaves hold pair records (then triLeafSize == 2)
    // static RTAO prebaking: AO factor table [parametrisation vertex][tube subdivision] and the per-line-vertex blending
    // weights (AmbientOcclusionFactorsBuffer / AmbientOcclusionBlendingWeightsBuffer, AmbientOcclusion.glsl:31-38)
    const float* bakedAo;
    const float* bakedBlendingWeights;
    const float4* prismFrames;  // per leaf {tangent0, index0}{normal0, start0}{tangent1, index1}{normal1, start1} (k_leaves)
    LvPrismDev prism;           // PPLL gather with ppll_fragment_source = raster_prism
};
#define LV_TRI_PAIR_CODE_BODY 0x38u // pair records: the second triangle of a tube face = (q0, q2, q3) (k_tri_leaves<true>, lv_bvh.hip)
#define LV_PRIM_CAPSULE 0
#define LV_PRIM_TRIANGLE 1
#define LV_PRIM_ELLIPTIC 2
#define LV_PRIM_PRISM 3         // the rasterised N-gon prism of the segment (all-hits walk of the PPLL gather only, lv_prism.h)
// shading variant of computeFragmentColor (template parameter of the hit shading; `true` / `false` of the earlier bool still mean
// USE_BANDS / plain): USE_BANDS and USE_ROTATING_HELICITY_BANDS never occur together (LineDataFlow.cpp:470,601-604,2423)
#define LV_SHADE_PLAIN 0
#define LV_SHADE_BANDS 1
#define LV_SHADE_HELICITY 2   // elliptic tubelets of band data (EllipticTubeRayTracing.glsl)

// tile list of a launch: tiles are tileW x tileH pixel rectangles with origins tilesXY[2*i], tilesXY[2*i+1]
struct LvTiles {
    const uint32_t* tilesXY;
    uint32_t numTiles, tileW, tileH, blocksX, blocksY; // 16x16-pixel blocks per tile (multiples of 4: 64x64 groups)
    // dispatch order of the 64x64-pixel groups (lv_group_order): slot -> group, most expensive group of the previous frame
    // first; NULL = as numbered.  groupCost collects this frame's cost per group (device clock ticks summed over the group's
    // workgroups).  Speed only: which workgroup renders which group never shows in the result.
    const uint32_t* groupOrder;
    uint32_t* groupCost;
};

struct LvCounters {
    unsigned long long rays, nodes, prims, hits;
};

// ---------------------------------------------------------------- RNG, RayTracingUtilities.glsl:134-181
__device__ __forceinline__ uint32_t lv_tea(uint32_t val0, uint32_t val1) {
    uint32_t v0 = val0, v1 = val1, s0 = 0;
#pragma unroll
    for (uint32_t n = 0; n < 16; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
__device__ __forceinline__ uint32_t lv_lcg(uint32_t& prev) {
    prev = 1664525u * prev + 1013904223u;
    return prev & 0x00FFFFFFu;
}
__device__ __forceinline__ float lv_rnd(uint32_t& seed) { return float(lv_lcg(seed)) / float(0x01000000); }

// sin/cos(2*pi*xi) by exact quadrant reduction + fixed polynomial (definition owned by the build; GLSL leaves
// sin/cos precision to the implementation).  Used for the AO hemisphere sample so directions are reproducible.
__device__ __forceinline__ void lv_sincos2pi(float xi, float& s, float& c) {
    float q = xi * 4.0f;
    float fq = floorf(q);
    int quad = int(fq) & 3;
    float r = q - fq;
    bool swp = r > 0.5f;
    float rr = swp ? (1.0f - r) : r;
    float a = rr * 1.57079632679489662f;
    float a2 = a * a;
    float sp = a * (1.0f + a2 * (-1.0f / 6.0f + a2 * (1.0f / 120.0f + a2 * (-1.0f / 5040.0f + a2 * (1.0f / 362880.0f)))));
    float cp = 1.0f + a2 * (-0.5f + a2 * (1.0f / 24.0f + a2 * (-1.0f / 720.0f + a2 * (1.0f / 40320.0f + a2 * (-1.0f / 3628800.0f)))));
    float sa = swp ? cp : sp;
    float ca = swp ? sp : cp;
    if (quad == 0) { s = sa; c = ca; }
    else if (quad == 1) { s = ca; c = -sa; }
    else if (quad == 2) { s = -sa; c = -ca; }
    else { s = -ca; c = sa; }
}

// sin / cos / atan2 of the elliptic-tube shaders (EllipticTubeRayTracing.glsl).  GLSL leaves their precision to the
// implementation and the sphere tracing loop takes discrete decisions on their results, so the build defines them by fixed
// float32 formulas: sin / cos through lv_sincos2pi after reducing the angle to a fraction of the full turn, atan through an odd
// polynomial on [-tan(pi/8), tan(pi/8)].
__device__ __forceinline__ void lv_sincos_rad(float a, float& s, float& c) {
    float u = a * 0.15915494309189535f;
    u = u - floorf(u);
    if (!(u < 1.0f)) u = 0.0f;
    lv_sincos2pi(u, s, c);
}
__device__ __forceinline__ float lv_atan2_det(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float r = 0.0f;
    if (mx > 0.0f) {
        float a = mn / mx;
        float base = 0.0f;
        if (a > 0.41421356237309503f) { a = (a - 1.0f) / (a + 1.0f); base = 0.78539816339744831f; }
        const float s = a * a;
        const float p = a * (1.0f + s * (-1.0f / 3.0f + s * (1.0f / 5.0f + s * (-1.0f / 7.0f + s * (1.0f / 9.0f + s * (-1.0f / 11.0f + s * (1.0f / 13.0f)))))));
        r = base + p;
        if (ay > ax) r = 1.57079632679489662f - r;
    }
    if (x < 0.0f) r = 3.14159265358979323846f - r;
    if (y < 0.0f) r = -r;
    return r;
}

// pow(x, y) of the shading code (Lighting.glsl:158-167,136-146 pow(|n.l|, 1.7), pow(|n.h|, 30); AmbientOcclusion.glsl:93 pow(ao, gamma);
// MlatInsert.glsl pow(transmittance, d)) as a BUILD-OWNED float32 definition: exp2(y * log2(x)) -- the form GLSL itself derives
// pow's precision from -- with log2 through the exponent bits + an atanh series on [sqrt(1/2), sqrt(2)] and exp2 through a 7th-order
// series on [-1/2, 1/2], every operation in a fixed order: bit-identical on host and device (frames and PPLL fragment colours then
// compare byte for byte instead of within the slack two different libm implementations need), relative error < 3e-6 wherever the
// result exceeds 1e-4 (tests/test_oracle.py), and about a third of the instructions of the correctly rounded library powf.
// x >= 0 (GLSL: undefined for x < 0); x == 0 -> 0 for y > 0, 1 for y == 0.
__device__ __forceinline__ float lv_pow_det(float x, float y) {
    if (!(x > 1.17549435e-38f)) return y > 0.0f ? 0.0f : (y == 0.0f ? 1.0f : __builtin_inff());
    const uint32_t bits = __float_as_uint(x);
    int e = int((bits >> 23) & 0xFFu) - 127;
    float m = __uint_as_float((bits & 0x007FFFFFu) | 0x3F800000u); // [1, 2)
    if (m > 1.41421356f) { m = m * 0.5f; e = e + 1; }               // [sqrt(1/2), sqrt(2)]
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float P = 0.333333333f + z * (0.2f + z * (0.142857143f + z * 0.111111111f));
    const float ln = 2.0f * s + (2.0f * s) * (z * P);              // ln(m) = 2 atanh(s)
    const float L = float(e) + ln * 1.44269504f;                   // log2(x)
    const float p = y * L;
    if (p < -125.0f) return 0.0f;
    if (p > 127.0f) return __builtin_inff();
    const float n = floorf(p + 0.5f);
    const float t = (p - n) * 0.693147181f;                        // |t| <= 0.3466
    const float Q = 1.0f + t * (1.0f + t * (0.5f + t * (0.166666667f + t * (0.0416666667f + t * (0.00833333333f + t * (0.00138888889f + t * 0.000198412698f))))));
    return __uint_as_float(__float_as_uint(Q) + (uint32_t(int(n)) << 23)); // Q * 2^n (Q in [0.70, 1.42], n >= -125: normal)
}
// log2(x), x > 0 and normal: the first half of lv_pow_det (exponent bits + atanh series); the mip level selection of the twist-line texture
__device__ __forceinline__ float lv_log2_det(float x) {
    const uint32_t bits = __float_as_uint(x);
    int e = int((bits >> 23) & 0xFFu) - 127;
    float m = __uint_as_float((bits & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e = e + 1; }
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float P = 0.333333333f + z * (0.2f + z * (0.142857143f + z * 0.111111111f));
    const float ln = 2.0f * s + (2.0f * s) * (z * P);
    return float(e) + ln * 1.44269504f;
}
// normalize(v) of the shading code as v * (1 / length(v)): one IEEE division instead of three (GLSL does not say how normalize
// divides; the twelve normalisations of computeFragmentColor + blinnPhongShadingTube were a third of the shading instructions)
// normalize(v) of the shading code: v * r(v . v) with r(x) = 1.0f / sqrtf(min(max(x, 2^-60), 2^60)), the IEEE-correct bits -- the squared
// length is clamped into a range that holds every length the shading code meets (a zero vector stays a zero vector instead of turning into
// NaNs; GLSL leaves normalize() of such vectors undefined).  The CPU checker states the same rule (normalizeShade).  What the clamp buys:
// inside the range the compiler's correctly rounded sqrtf (v_sqrt_f32 + a +-1 ulp correction from two fused residuals, wrapped in a
// denormal pre-scale and a class fix-up) and its correctly rounded division (v_rcp_f32 + three Newton steps, wrapped in v_div_scale /
// v_div_fixup) reduce to their cores -- the wrappers do nothing there: 18 branch-free VALU instructions + 2 for the clamp instead of 27
// and two SGPR-pair compares (k_ppll_shade_prism, 25 normalisations per fragment; EXPERIMENTS.md 12.3 has the forms that did NOT pay:
// out-of-range lanes through the compiler's sequence by a branch, and NaN outside the range by a compare + select).
// lv_selftest_rsqrt / tests/test_gpu_math.py compare all 2^32 arguments on the device against the rule evaluated with the compiler's ops.
#define LV_RSQRT_LO 8.67361737988403547206e-19f   // 2^-60
#define LV_RSQRT_HI 1152921504606846976.0f        // 2^60
__device__ __forceinline__ float lv_rsqrt_shade(float x) {
    x = fminf(fmaxf(x, LV_RSQRT_LO), LV_RSQRT_HI);   // (NaN -> 2^-60: maxNum / minNum return the other operand)
    float s = __builtin_amdgcn_sqrtf(x);
    const float sDn = __uint_as_float(__float_as_uint(s) - 1u), sUp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rDn = __builtin_fmaf(-sDn, s, x), rUp = __builtin_fmaf(-sUp, s, x);
    s = rDn <= 0.0f ? sDn : s;
    s = rUp > 0.0f ? sUp : s;
    const float r0 = __builtin_amdgcn_rcpf(s);
    const float r1 = __builtin_fmaf(__builtin_fmaf(-s, r0, 1.0f), r0, r0);
    const float q1 = __builtin_fmaf(__builtin_fmaf(-s, r1, 1.0f), r1, r1);
    return __builtin_fmaf(__builtin_fmaf(-s, q1, 1.0f), r1, q1);
}
// the same rule with the compiler's own division and square root: what lv_selftest_rsqrt compares lv_rsqrt_shade with
__device__ __forceinline__ float lv_rsqrt_shade_reference(float x) {
    return 1.0f / sqrtf(fminf(fmaxf(x, LV_RSQRT_LO), LV_RSQRT_HI));
}
#ifdef LV_RSQRT_PLAIN   // measurement variant (tools/variants.py): the compiler's division and square root
__device__ __forceinline__ f3 norm3s(f3 a) { const float r = lv_rsqrt_shade_reference(dot3(a, a)); return mk3(a.x * r, a.y * r, a.z * r); }
#else
__device__ __forceinline__ f3 norm3s(f3 a) { const float r = lv_rsqrt_shade(dot3(a, a)); return mk3(a.x * r, a.y * r, a.z * r); }
#endif

// shading_numerics = fast (round 6): the hardware's approximate reciprocal square root / reciprocal / log2 / exp2 (v_rsq_f32, v_rcp_f32,
// v_log_f32, v_exp_f32: <= 1 ulp each) in arithmetic that only ever reaches a COLOUR -- the lighting's normalisations, pow() and
// divisions.  Nothing that decides a hit, a coverage bit, a fragment's depth, its alpha or the length of a list goes through these
// (template parameter FAST of the shading routines: 0 = exact, 1 = lighting only, 2 = also the halo coordinate where it cannot reach
// alpha: the raster colour of plain tubes).  Frames then differ from the exact ones in the last bit of a few channels; the contract is
// +- 2 LSB (tests: whole C2 / C3 / C4 frames against the exact CPU checker, profiles/deviations_r06.json).
__device__ __forceinline__ float lv_rsqrt_fast(float x) { return __builtin_amdgcn_rsqf(fminf(fmaxf(x, LV_RSQRT_LO), LV_RSQRT_HI)); }
__device__ __forceinline__ f3 norm3f(f3 a) { const float r = lv_rsqrt_fast(dot3(a, a)); return mk3(a.x * r, a.y * r, a.z * r); }
template <bool FASTN>
__device__ __forceinline__ f3 norm3q(f3 a) { return FASTN ? norm3f(a) : norm3s(a); }
__device__ __forceinline__ float lv_div_fast(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
template <bool FASTN>
__device__ __forceinline__ float lv_divq(float a, float b) { return FASTN ? lv_div_fast(a, b) : a / b; }
__device__ __forceinline__ float lv_pow_fast(float x, float y) {
    if (!(x > 1.17549435e-38f)) return y > 0.0f ? 0.0f : (y == 0.0f ? 1.0f : __builtin_inff());   // the same rule as lv_pow_det
    return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
}
template <bool FASTN>
__device__ __forceinline__ float lv_powq(float x, float y) { return FASTN ? lv_pow_fast(x, y) : lv_pow_det(x, y); }

// ---------------------------------------------------------------- wave helpers (wave64)
__device__ __forceinline__ unsigned lv_lane() { return __lane_id(); }
__device__ __forceinline__ unsigned long long lv_wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float lv_wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float lv_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// order-preserving float <-> uint encoding for atomicMin/atomicMax
__device__ __forceinline__ uint32_t lv_f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float lv_ord2f(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}
