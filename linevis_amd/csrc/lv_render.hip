// lv_render.hip -- frame kernels of the hot path and their orchestration.
//
// Kernel                reference program it replaces
//   k_render_rt         TubeRayTracing.RayGen/.Miss (+ driver traversal, IntersectionTube, ClosestHitTubeAnalytic)
//                       Data/Shaders/Renderers/RayTracing/TubeRayTracing.glsl:61-82,198-298
//   k_ao_primary        VulkanRayTracedAmbientOcclusion.Compute, primary-ray half (glsl:178-281) + wave compaction
//   k_ao_rays           .. sample loop (glsl:288-306) as persistent waves pulling AO rays from a global queue
//   k_ao_reduce         .. per-pixel sum in sample order + accumulate/store (glsl:309-319)
//   k_ppll_gather       tube rasterisation + gatherFragment, Data/Shaders/Renderers/PPLL/LinkedListGather.glsl:33-72
//   k_ppll_resolve      LinkedListResolve.Fragment + frontToBackPQ, LinkedListResolve.glsl:57-105, LinkedListSort.glsl:177-238
//   k_depth_minmax      ComputeDepthValues.Compute + MinMaxReduce.Compute, Data/Shaders/DepthCues/*.glsl
//   k_bake_setup        VulkanAmbientOcclusionBaker.Compute, line-point interpolation + ray origins (glsl:110-131,231-262);
//                       its sample loop runs as the BAKE instantiation of k_ao_rays, the running mean in k_ao_reduce<true>
// The triangle instantiations (LV_PRIM_TRIANGLE) of k_ao_primary / k_ao_rays / k_render_rt trace the reference's
// triangle tubes (RTAO geometry, "Triangle Mesh" geometry mode: ClosestHitTubeTriangles, TubeRayTracing.glsl:301-352).
// Host orchestration follows VulkanRayTracer::render (VulkanRayTracer.cpp:131-154), LineRenderer::renderBase
// (LineRenderer.cpp:248-277) and PerPixelLinkedListLineRenderer::render (PerPixelLinkedListLineRenderer.cpp:399-427).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "lv_internal.h"
#include "lv_trace.h"
#include "lv_tile.h"

namespace {

// ================================================================ ray tracer colour pass
// Control flow is wave-uniform around every trace (lv_trace_closest is a wave-cooperative routine): the sample loop and
// the transparency loop run while ANY lane of the wave still needs a trace; lanes that are done pass active = false.
// PRE: the hits of every pixel's ray were traced ahead of the RTAO pass (lv_colour_first_hits, launched together with the RTAO
// primaries: k_primary_pair) and wait in firstHit[k * stride + px.outIndex] = {t bits, (leaf << 2) | kind or LV_INVALID}, k = position in
// the transparency loop, k < LV_PRE_HITS; the kernel shades them and only traces where a pixel's loop goes deeper than that.  Same hits,
// same frame, byte for byte: which hit follows which depends on the alpha of the shaded hit, and alpha (transfer function x silhouette
// coverage) does not depend on the ambient occlusion.
#define LV_PRE_HITS 4u
template <bool STATS, int PRIM, int BANDS = LV_SHADE_PLAIN, bool PRE = false, int FAST = 0>
__global__ __launch_bounds__(LV_BLOCK, (STATS || PRIM == LV_PRIM_ELLIPTIC) ? 1 : LV_RT_MIN_WAVES) void k_render_rt(const LvUniforms U, const LvSceneDev S, const LvTiles T,
                                                        uint32_t* __restrict__ out, LvDevCounters* dc,
                                                        const uint2* __restrict__ firstHit = nullptr, const size_t firstHitStride = 0) {
    __shared__ unsigned s_stack[LV_STACK_LDS * LV_BLOCK];
    LV_COOP_SHARED(LV_BLOCK / LV_WAVE);
    LV_COOP_MEM(cm);
    LvPixel px;
    if (!lv_block_pixel(U, T, px)) return;
    const unsigned long long tg0 = lv_group_clock();
    const LvStackMem sm = lv_stack_mem(s_stack, S.stackOverflow);
    LvCounters cnt = {0, 0, 0, 0};
    const bool capped = U.useCappedTubes != 0 || U.lssGeometry != 0;
    const float HIT_DISTANCE_EPSILON = 1e-5f;
    const float aoTexel = (px.inView && U.useAmbientOcclusion && !U.aoPrebaked) ? S.ao[size_t(px.y) * U.width + px.x] : 1.0f;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const uint32_t nSamples = U.useJitteredRays ? U.numSamplesPerFrame : 1u;
    for (uint32_t sampleIdx = 0; sampleIdx < nSamples; sampleIdx++) { // uniform trip count
        float xix = 0.5f, xiy = 0.5f;
        if (U.useJitteredRays) {
            uint32_t seed = U.useDeterministicSampling
                    ? lv_tea(19u, U.frameNumber * U.numSamplesPerFrame + sampleIdx)
                    : lv_tea(px.x + px.y * U.width, U.frameNumber * U.numSamplesPerFrame + sampleIdx);
            xix = lv_rnd(seed);
            xiy = lv_rnd(seed);
        }
        f3 o, d;
        lv_primary_ray(U, px.x, px.y, xix, xiy, o, d);
        // traceRayTransparent, TubeRayTracing.glsl:61-82
        float fc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float tMin = 0.0001f;
        const float tMax = 1000.0f;
        bool tracing = px.inView;
        for (uint32_t hitIdx = 0; hitIdx < U.maxDepthComplexity && __any(tracing); hitIdx++) {
            LvHit h;
            if (PRE && hitIdx < LV_PRE_HITS && sampleIdx == 0u) {
                h.found = false; h.t = 0.0f; h.leaf = 0u; h.kind = 0;
                if (tracing) {
                    const uint2 fh = firstHit[size_t(hitIdx) * firstHitStride + px.outIndex];
                    h.found = fh.y != LV_INVALID;
                    h.t = __uint_as_float(fh.x);
                    h.leaf = fh.y >> 2;
                    h.kind = int(fh.y & 3u);
                }
            } else {
                h = lv_trace_closest<STATS, false, PRIM>(S, U.radius, capped, tracing, o, d, tMin, tMax, sm, cm, cnt);
            }
            if (tracing) {
                f4 hc;
                float payloadHitT;
                if (h.found) {
                    hc = PRIM == LV_PRIM_TRIANGLE   ? lv_shade_hit_triangle<BANDS>(S, U, aoTexel, o, d, h.leaf, payloadHitT)
                         : PRIM == LV_PRIM_ELLIPTIC ? lv_shade_hit_elliptic(S, U, aoTexel, o, d, h, payloadHitT)
                                                    : lv_shade_hit<BANDS, FAST>(S, U, aoTexel, o, d, h, payloadHitT);
                    if (STATS) cnt.hits++;
                } else { // Miss, TubeRayTracing.glsl:290-297
                    hc.x = U.background[0]; hc.y = U.background[1]; hc.z = U.background[2]; hc.w = U.background[3];
                    payloadHitT = 0.0f;
                }
                tMin = payloadHitT + fmaxf(payloadHitT * HIT_DISTANCE_EPSILON, 1e-7f);
                fc[0] = fc[0] + ((1.0f - fc[3]) * hc.w) * hc.x;
                fc[1] = fc[1] + ((1.0f - fc[3]) * hc.w) * hc.y;
                fc[2] = fc[2] + ((1.0f - fc[3]) * hc.w) * hc.z;
                fc[3] = fc[3] + (1.0f - fc[3]) * hc.w;
                if (!h.found || fc[3] > 0.99f) tracing = false;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] += fc[k];
    }
    if (px.inView) {
        if (U.useJitteredRays) {
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] /= float(U.numSamplesPerFrame);
        }
        f4 c; c.x = acc[0]; c.y = acc[1]; c.z = acc[2]; c.w = acc[3];
        out[px.outIndex] = lv_store_color(S, U, px.x, px.y, c);
    } else if (px.inTile) {
        f4 c; c.x = U.background[0]; c.y = U.background[1]; c.z = U.background[2]; c.w = U.background[3];
        out[px.outIndex] = lv_pack_unorm4x8(c);
    }
    lv_group_cost_add(T, px, tg0);
    if (STATS) { lv_flush_max_nodes(cnt, dc); lv_flush_counters(cnt, dc); }
}

// ================================================================ RTAO
// G-buffer layout: the segment of 64x64-pixel group g holds exactly the pixels of its tile that fall into the group (a 66 x 66 tile
// -- 64 + a one-pixel AO halo -- is cut into 2 x 2 groups of 64 x 64, 2 x 64, 64 x 2 and 2 x 2 pixels): tile-major, then group rows,
// then groups of a row.  The buffer then has numTiles * tileW * tileH slots, not groups * 4096 (4 x the memory with any halo).
struct LvAoLayout { uint32_t tileW, tileH, groupsX, groupsPerTile; };
__device__ __host__ __forceinline__ LvAoLayout lv_ao_layout(const LvTiles& T) {
    LvAoLayout L;
    L.tileW = T.tileW; L.tileH = T.tileH; L.groupsX = T.blocksX / 4u; L.groupsPerTile = (T.blocksX / 4u) * (T.blocksY / 4u);
    return L;
}
__device__ __forceinline__ size_t lv_ao_group_base(const LvAoLayout& L, uint32_t group) {
    const uint32_t tile = group / L.groupsPerTile, gi = group % L.groupsPerTile;
    const uint32_t gxi = gi % L.groupsX, gyi = gi / L.groupsX;
    const uint32_t hrow = min(64u, L.tileH - 64u * gyi);
    return size_t(tile) * L.tileW * L.tileH + size_t(gyi) * 64u * L.tileW + size_t(gxi) * 64u * hrow;
}

// G-buffer entry of a pixel whose primary ray hit: 3 x float4
//   g0 = {hit position, offsetFactor}, g1 = {surface tangent, pixel index bits}, g2 = {surface normal, 0}
template <bool STATS, int PRIM>
__device__ __forceinline__ void lv_ao_primary_body(const LvUniforms& U, const LvSceneDev& S, const LvTiles& T, uint32_t blockId,
                                                   unsigned* s_stack, LvCoopMem& cm,
                                                   const float* aoIn, float* ao, float4* __restrict__ gbuf,
                                                   uint32_t* __restrict__ groupCount, LvDevCounters* dc,
                                                   const float4* featNormalIn, float4* featNormal,
                                                   const float4* featPositionIn, float4* featPosition,
                                                   const LvSvgfFeat& SF) {
    LvPixel px;
    if (!lv_block_pixel(U, T, px, blockId)) return;
    const unsigned long long tg0 = lv_group_clock();
    LvCounters cnt = {0, 0, 0, 0};
    bool hasHit = false;
    float4 g0, g1, g2;
    const uint32_t pix = px.x + px.y * U.width;
    const uint32_t globalFrameNumber = U.aoGlobalFrameNumber; // VulkanRayTracedAmbientOcclusion.cpp:576-581
    uint32_t seed = lv_tea(pix, globalFrameNumber);
    float xix = 0.5f, xiy = 0.5f;
    if (U.aoJitterPrimary) { xix = lv_rnd(seed); xiy = lv_rnd(seed); }
    f3 o, d;
    lv_primary_ray(U, px.x, px.y, xix, xiy, o, d);
    // wave-cooperative: every lane calls, lanes outside the viewport only help testing
    const LvHit h = lv_trace_closest<STATS, false, PRIM>(S, U.radius, U.useCappedTubes != 0, px.inView, o, d, 0.0001f,
                                                         1000.0f, lv_stack_mem(s_stack, S.stackOverflow), cm, cnt);
    if (px.inView) {
        if (h.found && PRIM == LV_PRIM_TRIANGLE) {
            // barycentric reconstruction, VulkanRayTracedAmbientOcclusion.glsl:219-263,280
            hasHit = true;
            const uint32_t i0 = S.triIdx[3 * size_t(h.leaf)], i1 = S.triIdx[3 * size_t(h.leaf) + 1],
                           i2 = S.triIdx[3 * size_t(h.leaf) + 2];
            const lv_tube_vertex& vd0 = S.triVerts[i0];
            const lv_tube_vertex& vd1 = S.triVerts[i1];
            const lv_tube_vertex& vd2 = S.triVerts[i2];
            const f3 p0 = mk3(vd0.vertexPosition[0], vd0.vertexPosition[1], vd0.vertexPosition[2]);
            const f3 p1 = mk3(vd1.vertexPosition[0], vd1.vertexPosition[1], vd1.vertexPosition[2]);
            const f3 p2 = mk3(vd2.vertexPosition[0], vd2.vertexPosition[1], vd2.vertexPosition[2]);
            float tt = 0.0f, bu = 0.0f, bv = 0.0f; // same function, same inputs as the winning test: identical (t, u, v)
            lv_ray_triangle(o, d, mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z), p0, p1, p2, S.triPad, tt, bu, bv);
            const float b0 = (1.0f - bu) - bv;
            const lv_line_point& lp0 = S.triPoints[vd0.vertexLinePointIndex & 0x7FFFFFFFu];
            const lv_line_point& lp1 = S.triPoints[vd1.vertexLinePointIndex & 0x7FFFFFFFu];
            const lv_line_point& lp2 = S.triPoints[vd2.vertexLinePointIndex & 0x7FFFFFFFu];
            auto lerp3 = [&](const float* a, const float* b, const float* c) {
                return (mk3(a[0], a[1], a[2]) * b0 + mk3(b[0], b[1], b[2]) * bu) + mk3(c[0], c[1], c[2]) * bv;
            };
            const f3 vertexPositionWorld = lerp3(vd0.vertexPosition, vd1.vertexPosition, vd2.vertexPosition);
            const f3 surfaceNormal = norm3(lerp3(vd0.vertexNormal, vd1.vertexNormal, vd2.vertexNormal));
            const f3 linePosition = lerp3(lp0.linePosition, lp1.linePosition, lp2.linePosition);
            const f3 surfaceTangent = norm3(lerp3(lp0.lineTangent, lp1.lineTangent, lp2.lineTangent));
            const float offsetFactor = len3(linePosition - vertexPositionWorld) / U.subdivisionCorrectionFactor;
            g0 = make_float4(vertexPositionWorld.x, vertexPositionWorld.y, vertexPositionWorld.z, offsetFactor);
            g1 = make_float4(surfaceTangent.x, surfaceTangent.y, surfaceTangent.z, __uint_as_float(pix));
            g2 = make_float4(surfaceNormal.x, surfaceNormal.y, surfaceNormal.z, 0.0f);
        } else if (h.found && PRIM == LV_PRIM_ELLIPTIC) {
            // band data: the reference's RTAO pass traces the elliptic TRIANGLE tubes (createCappedTriangleEllipticTubesRenderDataCPU);
            // the build traces the analytic tubelets of the colour pass -- the same substitution as capsules for circular tubes
            hasHit = true;
            const uint32_t seg = S.leafSeg[h.leaf];
            const lv_line_point& lp0 = S.points[S.segIdx[2 * seg]];
            const lv_line_point& lp1 = S.points[S.segIdx[2 * seg + 1]];
            const LvEllipticSurface E = lv_elliptic_surface(U, o, d, h.t, lp0, lp1);
            const f3 vertexPositionWorld = o + d * h.t;
            const f3 t0 = mk3(lp0.lineTangent[0], lp0.lineTangent[1], lp0.lineTangent[2]);
            const f3 t1 = mk3(lp1.lineTangent[0], lp1.lineTangent[1], lp1.lineTangent[2]);
            const f3 surfaceTangent = norm3((1.0f - E.t) * t0 + E.t * t1);
            const float offsetFactor = len3(E.linePosition - vertexPositionWorld) / U.subdivisionCorrectionFactor;
            g0 = make_float4(vertexPositionWorld.x, vertexPositionWorld.y, vertexPositionWorld.z, offsetFactor);
            g1 = make_float4(surfaceTangent.x, surfaceTangent.y, surfaceTangent.z, __uint_as_float(pix));
            g2 = make_float4(E.normal.x, E.normal.y, E.normal.z, 0.0f);
        } else if (h.found) {
            hasHit = true;
            const float4 ra = S.segs[2 * h.leaf], rb = S.segs[2 * h.leaf + 1];
            const uint32_t seg = S.leafSeg[h.leaf];
            const lv_line_point& lp0 = S.points[S.segIdx[2 * seg]];
            const lv_line_point& lp1 = S.points[S.segIdx[2 * seg + 1]];
            const f3 P0 = mk3(ra.x, ra.y, ra.z), P1 = mk3(rb.x, rb.y, rb.z);
            f3 vertexPositionWorld = o + d * h.t;
            f3 v = P1 - P0;
            float ts;
            if (h.kind == 0) ts = dot3(v, vertexPositionWorld - P0) / dot3(v, v);
            else ts = h.kind == 1 ? 0.0f : 1.0f;
            f3 linePosition = h.kind == 0 ? P0 + ts * v : (h.kind == 1 ? P0 : P1);
            f3 surfaceNormal = norm3(vertexPositionWorld - linePosition);
            f3 t0 = mk3(lp0.lineTangent[0], lp0.lineTangent[1], lp0.lineTangent[2]);
            f3 t1 = mk3(lp1.lineTangent[0], lp1.lineTangent[1], lp1.lineTangent[2]);
            f3 surfaceTangent = norm3((1.0f - ts) * t0 + ts * t1);
            float offsetFactor = len3(linePosition - vertexPositionWorld) / U.subdivisionCorrectionFactor;
            g0 = make_float4(vertexPositionWorld.x, vertexPositionWorld.y, vertexPositionWorld.z, offsetFactor);
            g1 = make_float4(surfaceTangent.x, surfaceTangent.y, surfaceTangent.z, __uint_as_float(pix));
            g2 = make_float4(surfaceNormal.x, surfaceNormal.y, surfaceNormal.z, 0.0f);
        } else {
            // miss: aoFactor = 1, accumulate (glsl:311-319)
            float aoFactor = 1.0f;
            if (U.aoFrameNumber != 0) aoFactor = mixf(aoIn[pix], aoFactor, 1.0f / float(U.aoFrameNumber + 1));
            ao[pix] = aoFactor;
        }
    }
    // denoiser feature maps (VulkanRayTracedAmbientOcclusion.glsl:321-399, WRITE_NORMAL_MAP / WRITE_POSITION_MAP with
    // accumulation): view-space normal {xyz, 0} and view-space position {xyz, 1} of the primary hit; a miss leaves
    // surfaceNormal = vertexPositionWorld = 0 (glsl:211-212); running means over the iterations
    if (featNormal && px.inView) {
        const f3 sn = hasHit ? mk3(g2.x, g2.y, g2.z) : mk3(0.0f, 0.0f, 0.0f);
        const f3 vp = hasHit ? mk3(g0.x, g0.y, g0.z) : mk3(0.0f, 0.0f, 0.0f);
        // camNormal = (transpose(inverseViewMatrix) * vec4(surfaceNormal, 0)).xyz, VulkanRayTracedAmbientOcclusion.cpp:569
        const float* m = U.invView;
        f3 n = mk3(((m[0] * sn.x + m[1] * sn.y) + m[2] * sn.z) + m[3] * 0.0f, ((m[4] * sn.x + m[5] * sn.y) + m[6] * sn.z) + m[7] * 0.0f,
                   ((m[8] * sn.x + m[9] * sn.y) + m[10] * sn.z) + m[11] * 0.0f);
        const f4 pv = mulM4(U.view, vp.x, vp.y, vp.z, 1.0f);
        f3 q = mk3(pv.x, pv.y, pv.z);
        if (U.aoFrameNumber != 0) {
            const float a = 1.0f / float(U.aoFrameNumber + 1);
            const float4 on = featNormalIn[pix], op = featPositionIn[pix];
            n = mk3(mixf(on.x, n.x, a), mixf(on.y, n.y, a), mixf(on.z, n.z, a));
            const float len = len3(n);
            if (len > 1e-5f) n = mk3(n.x / len, n.y / len, n.z / len);
            q = mk3(mixf(op.x, q.x, a), mixf(op.y, q.y, a), mixf(op.z, q.z, a));
        }
        featNormal[pix] = make_float4(n.x, n.y, n.z, 0.0f);
        featPosition[pix] = make_float4(q.x, q.y, q.z, 1.0f);
    }
    if (SF.normalDepth && px.inView) {
        // SVGF's maps, not accumulated: world-space normal (0 on a miss), depth = -z_view (farDistance on a miss), flow =
        // writePos - the hit's pixel position under last frame's view-projection, depth fwidth = |cot| of the angles between
        // the view-space normal and the camera x / y axes (glsl:350-464)
        const f3 sn = hasHit ? mk3(g2.x, g2.y, g2.z) : mk3(0.0f, 0.0f, 0.0f);
        const f3 vp = hasHit ? mk3(g0.x, g0.y, g0.z) : mk3(0.0f, 0.0f, 0.0f);
        const f4 pv = mulM4(U.view, vp.x, vp.y, vp.z, 1.0f);
        float fx = 0.0f, fy = 0.0f, fw = 0.0f;
        if (hasHit) {
            f4 ndc = mulM4(SF.lastFrameViewProj, vp.x, vp.y, vp.z, 1.0f);
            ndc.x /= ndc.w; ndc.y /= ndc.w;
            fx = float(px.x) - ((0.5f * ndc.x + 0.5f) * float(U.width) - 0.5f);
            fy = float(px.y) - ((0.5f * ndc.y + 0.5f) * float(U.height) - 0.5f);
            const float* m = U.invView;
            const float A = ((m[0] * sn.x + m[1] * sn.y) + m[2] * sn.z) + m[3] * 0.0f;
            const float B = ((m[4] * sn.x + m[5] * sn.y) + m[6] * sn.z) + m[7] * 0.0f;
            fw = fabsf(A / sqrtf(1.0f - A * A)) + fabsf(B / sqrtf(1.0f - B * B));
        }
        SF.normalDepth[pix] = make_float4(sn.x, sn.y, sn.z, hasHit ? -pv.z : U.farDist);
        SF.flowFwidth[pix] = make_float4(fx, fy, fw, 0.0f);
    }
    // active-ray compaction: ballot + prefix popcount, one atomic per wave -- PER 64x64-PIXEL GROUP (all pixels of a wave
    // belong to one group): the G-buffer is segmented by group, slot = group * 4096 + position inside the group's segment,
    // and k_ao_rays walks the segments in group (= Morton tile) order.  The AO rays in flight at any moment then belong to two
    // or three neighbouring 64x64 groups instead of the ~50 groups whose workgroups of THIS kernel run concurrently, so
    // that their working set (the geometry within the AO radius of those groups) fits the L2s; with one global counter the
    // G-buffer order followed the interleaving of the running workgroups (L2 hit rate 88 % on the 64 MB capsule scene, 56 % on
    // the 1 GB triangle scene).
    const unsigned long long mask = __ballot(hasHit);
    if (mask) {
        const unsigned lane = lv_lane();
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&groupCount[px.group], unsigned(__popcll(mask)));
        base = __shfl(base, 0, 64);
        if (hasHit) {
            const size_t slot = lv_ao_group_base(lv_ao_layout(T), px.group) + base + unsigned(__popcll(mask & ((1ull << lane) - 1ull)));
            gbuf[3 * size_t(slot) + 0] = g0;
            gbuf[3 * size_t(slot) + 1] = g1;
            gbuf[3 * size_t(slot) + 2] = g2;
        }
    }
    lv_group_cost_add(T, px, tg0);
    if (STATS) lv_flush_counters(cnt, dc);
}

template <bool STATS, int PRIM>
__global__ __launch_bounds__(LV_BLOCK) void k_ao_primary(const LvUniforms U, const LvSceneDev S, const LvTiles T,
                                                         const float* aoIn, float* ao, float4* __restrict__ gbuf,
                                                         uint32_t* __restrict__ groupCount, LvDevCounters* dc,
                                                         const float4* featNormalIn, float4* featNormal,
                                                         const float4* featPositionIn, float4* featPosition,
                                                         const LvSvgfFeat SF) {
    __shared__ unsigned s_stack[LV_STACK_LDS * LV_BLOCK];
    LV_COOP_SHARED(LV_BLOCK / LV_WAVE);
    LV_COOP_MEM(cm);
    lv_ao_primary_body<STATS, PRIM>(U, S, T, blockIdx.x, s_stack, cm, aoIn, ao, gbuf, groupCount, dc, featNormalIn, featNormal, featPositionIn,
                                    featPosition, SF);
}

// The colour pass' ray of every pixel (sample 0: the pixel centre, or the jittered position of TubeRayTracing.glsl:236-262) against the
// capsules, through the transparency loop of traceRayTransparent (TubeRayTracing.glsl:61-82) for up to LV_PRE_HITS hits: which hits the
// loop visits depends on the alpha of the shaded hits, and alpha does not depend on the ambient occlusion (lv_compute_fragment_color_t:
// out.w = transfer function alpha x silhouette coverage) -- the hits are shaded here WITHOUT ambient occlusion, for their alpha and
// payload distance only.  firstHit[k * stride + px.outIndex] = {t bits, (leaf << 2) | kind}, LV_INVALID = miss;
// k_render_rt<..., PRE = true> shades them with the AO image after the RTAO pass.
template <bool STATS, int BANDS>
__device__ __forceinline__ void lv_colour_first_hits(const LvUniforms& U, const LvSceneDev& S, const LvTiles& T, uint32_t blockId,
                                                     unsigned* s_stack, LvCoopMem& cm, uint2* __restrict__ firstHit, size_t stride,
                                                     LvDevCounters* dc) {
    LvPixel px;
    if (!lv_block_pixel(U, T, px, blockId)) return;
    const unsigned long long tg0 = lv_group_clock();
    const LvStackMem sm = lv_stack_mem(s_stack, S.stackOverflow);
    LvCounters cnt = {0, 0, 0, 0};
    LvUniforms UA = U;            // the same shading without the AO term (S.ao is not there yet)
    UA.useAmbientOcclusion = 0u;
    UA.aoProjectLookup = 0u;
    float xix = 0.5f, xiy = 0.5f;
    if (U.useJitteredRays) {   // as k_render_rt, sampleIdx = 0
        uint32_t seed = U.useDeterministicSampling ? lv_tea(19u, U.frameNumber * U.numSamplesPerFrame)
                                                   : lv_tea(px.x + px.y * U.width, U.frameNumber * U.numSamplesPerFrame);
        xix = lv_rnd(seed);
        xiy = lv_rnd(seed);
    }
    f3 o, d;
    lv_primary_ray(U, px.x, px.y, xix, xiy, o, d);
    const bool capped = U.useCappedTubes != 0 || U.lssGeometry != 0;
    const float HIT_DISTANCE_EPSILON = 1e-5f;
    float tMin = 0.0001f, alpha = 0.0f;
    bool tracing = px.inView;
    for (uint32_t hitIdx = 0; hitIdx < LV_PRE_HITS && hitIdx < U.maxDepthComplexity && __any(tracing); hitIdx++) {
        const LvHit h = lv_trace_closest<STATS, false, LV_PRIM_CAPSULE>(S, U.radius, capped, tracing, o, d, tMin, 1000.0f, sm, cm, cnt);
        if (tracing) {
            firstHit[size_t(hitIdx) * stride + px.outIndex] =
                    h.found ? make_uint2(__float_as_uint(h.t), (h.leaf << 2) | uint32_t(h.kind)) : make_uint2(0u, LV_INVALID);
            if (!h.found) {
                tracing = false;
            } else {
                float payloadHitT;
                const f4 hc = lv_shade_hit<BANDS>(S, UA, 1.0f, o, d, h, payloadHitT);
                tMin = payloadHitT + fmaxf(payloadHitT * HIT_DISTANCE_EPSILON, 1e-7f);
                alpha = alpha + (1.0f - alpha) * hc.w;
                if (alpha > 0.99f) tracing = false;
            }
        }
    }
    lv_group_cost_add(T, px, tg0);
    if (STATS) { lv_flush_max_nodes(cnt, dc); lv_flush_counters(cnt, dc); }
}

// k_ao_primary and the colour pass' first-hit trace in ONE launch: workgroups [0, gridAo) trace the RTAO primaries (scene SA: the
// triangle tubes or the capsules, tiles TA), workgroups [gridAo, gridAo + gridColour) the colour rays (scene SC, tiles TC).  Both are
// latency-bound chains of dependent node fetches with one ray per lane; on a rank that owns 1/8 of the tiles neither fills the GPU
// (4 waves per SIMD), and run one after the other they were 0.44 of a 1.27-ms frame (profiles/shard_probe_r05_c3t.json).  Together
// their waves share the CUs and the two chains overlap.
template <bool STATS, int PRIM, int BANDS>
__global__ __launch_bounds__(LV_BLOCK) void k_primary_pair(const LvUniforms U, const LvSceneDev SA, const LvTiles TA, const uint32_t gridAo,
                                                           const float* aoIn, float* ao, float4* __restrict__ gbuf,
                                                           uint32_t* __restrict__ groupCount, LvDevCounters* dc,
                                                           const float4* featNormalIn, float4* featNormal,
                                                           const float4* featPositionIn, float4* featPosition,
                                                           const LvSvgfFeat SF, const LvSceneDev SC, const LvTiles TC,
                                                           uint2* __restrict__ firstHit, const size_t firstHitStride) {
    __shared__ unsigned s_stack[LV_STACK_LDS * LV_BLOCK];
    LV_COOP_SHARED(LV_BLOCK / LV_WAVE);
    LV_COOP_MEM(cm);
    if (blockIdx.x < gridAo)
        lv_ao_primary_body<STATS, PRIM>(U, SA, TA, blockIdx.x, s_stack, cm, aoIn, ao, gbuf, groupCount, dc, featNormalIn, featNormal,
                                        featPositionIn, featPosition, SF);
    else
        lv_colour_first_hits<STATS, BANDS>(U, SC, TC, blockIdx.x - gridAo, s_stack, cm, firstHit, firstHitStride, dc);
}

// exclusive prefix sum of the per-tile hit-pixel counts: tileBase[t] = first compacted-pixel ordinal of tile t,
// tileBase[numTiles] = dc->aoCount = all hit pixels of the launch (one workgroup; numTiles is a few hundred to a few thousand)
__global__ __launch_bounds__(LV_BLOCK) void k_ao_tile_scan(const uint32_t* __restrict__ tileCount, uint32_t numTiles,
                                                           uint32_t* __restrict__ tileBase, LvDevCounters* dc) {
    __shared__ uint32_t s_part[LV_BLOCK];
    const uint32_t per = (numTiles + LV_BLOCK - 1u) / LV_BLOCK;
    const uint32_t b = threadIdx.x * per, e = min(b + per, numTiles);
    uint32_t sum = 0;
    for (uint32_t i = b; i < e; i++) sum += tileCount[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < LV_BLOCK; i++) { const uint32_t v = s_part[i]; s_part[i] = run; run += v; }
        tileBase[numTiles] = run;
        dc->aoCount = run;
        dc->aoQueueHead = 0ull; // the ray queue of the k_ao_rays launch that follows
    }
    __syncthreads();
    uint32_t run = s_part[threadIdx.x];
    for (uint32_t i = b; i < e; i++) { tileBase[i] = run; run += tileCount[i]; }
}

// compacted-pixel ordinal -> G-buffer slot: group g with tileBase[g] <= ordinal < tileBase[g + 1] (binary search; the table
// is a few KB and stays in L1), slot = first slot of the group's segment + (ordinal - tileBase[g]).  tileBase == nullptr: slot = ordinal.
__device__ __forceinline__ size_t lv_ao_slot(const uint32_t* __restrict__ tileBase, uint32_t numTiles, const LvAoLayout& tileCapacity,
                                             uint32_t ordinal) {
    if (!tileBase) return ordinal;
    uint32_t lo = 0, hi = numTiles; // invariant: tileBase[lo] <= ordinal < tileBase[hi]
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tileBase[mid] <= ordinal) lo = mid; else hi = mid;
    }
    return lv_ao_group_base(tileCapacity, lo) + (ordinal - tileBase[lo]);
}

#ifdef LV_AO_OCTANT_PERM
// tools/variants.py experiment (EXPERIMENTS.md 12.2): an UPPER BOUND for "ray binning at generation".  A pre-pass
// (not charged to k_ao_rays) orders the rays of every block of LV_AO_OCTANT_PERM pixels by the octant of their direction; k_ao_rays then
// takes ray perm[i] where it took ray i, so that the 128-ray chunks a wave pulls hold rays of one or two octants.  Never in the product build.
__device__ uint32_t* g_aoPerm;
__global__ __launch_bounds__(256) void k_ao_octant_perm(const LvUniforms U, const float4* __restrict__ gbuf, uint32_t* __restrict__ perm,
                                                        const LvDevCounters* dc, const uint32_t* __restrict__ tileBase, uint32_t numTiles,
                                                        const LvAoLayout tileCapacity) {
    __shared__ unsigned s_cnt[8], s_cur[8];
    const uint32_t spp = U.aoSamplesPerFrame;
    const unsigned long long total = (unsigned long long)(dc->aoCount) * spp;
    const unsigned long long per = (unsigned long long)LV_AO_OCTANT_PERM * spp;
    for (unsigned long long base = blockIdx.x * per; base < total; base += gridDim.x * per) {
        if (threadIdx.x < 8) s_cnt[threadIdx.x] = 0u;
        __syncthreads();
        const unsigned long long end = base + per < total ? base + per : total;
        for (int pass = 0; pass < 2; pass++) {
            for (unsigned long long rr = base + threadIdx.x; rr < end; rr += blockDim.x) {
                const uint32_t smpIdx = uint32_t(rr % spp);
                const size_t slot = lv_ao_slot(tileBase, numTiles, tileCapacity, uint32_t(rr / spp));
                const float4 g1 = gbuf[3 * slot + 1], g2 = gbuf[3 * slot + 2];
                const f3 T = mk3(g1.x, g1.y, g1.z), N = mk3(g2.x, g2.y, g2.z);
                const f3 B = cross3(N, T);
                uint32_t seed = lv_tea(__float_as_uint(g1.w), U.aoGlobalFrameNumber * spp + smpIdx);
                const float xi0 = lv_rnd(seed), xi1 = lv_rnd(seed);
                float sn, cs;
                lv_sincos2pi(xi1, sn, cs);
                const float rs = sqrtf(1.0f - xi0 * xi0);
                const f3 smp = mk3(cs * rs, sn * rs, xi0);
                const f3 d = mk3((T.x * smp.x + B.x * smp.y) + N.x * smp.z, (T.y * smp.x + B.y * smp.y) + N.y * smp.z,
                                 (T.z * smp.x + B.z * smp.y) + N.z * smp.z);
                const unsigned oct = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
                if (pass == 0) atomicAdd(&s_cnt[oct], 1u);
                else perm[base + atomicAdd(&s_cur[oct], 1u)] = uint32_t(rr);
            }
            __syncthreads();
            if (pass == 0 && threadIdx.x == 0) { unsigned run = 0; for (int k = 0; k < 8; k++) { s_cur[k] = run; run += s_cnt[k]; } }
            __syncthreads();
        }
    }
}
#define LV_AO_RAY_INDEX(i) ((unsigned long long)g_aoPerm[i])
#else
#define LV_AO_RAY_INDEX(i) (i)
#endif

// AO sample rays: PERSISTENT waves that keep three kinds of work apart and run each of them with (nearly) all
// 64 lanes busy.  AO ray r belongs to compacted pixel r / spp, sample r % spp.
//
//   generate  a wave takes LV_AO_CHUNK consecutive ray indices from the global queue head with one atomic and turns
//             them into rays 64 at a time -- TEA seed, hemisphere sample, frame transform, normalisation -- into an LDS
//             buffer (s_gen); idle lanes later pick a ready ray from there (two ds_read_b128), so the ~330-instruction
//             setup always runs at full width.
//   descend   a lane owns a ray only while it is alive; every lane with an inner node does one node step per
//             iteration (LDS-staged stack).  Leaves are NOT tested by the lane that meets them: (owner lane, leaf) is
//             appended to a wave-local FIFO in LDS (ballot + prefix popcount, no atomics).
//   test      as soon as 64 pairs wait, all 64 lanes take one pair each, read the owner's ray from LDS, run the
//             capsule test (8 IEEE divisions + 4 square roots, ~350 instructions) and merge into the owner's best hit
//             with one 64-bit LDS atomicMin on the key (t bits << 32 | original segment): exactly "closest hit, ties
//             to the lowest segment index".  The leaf test is the expensive phase; with one ray per thread it ran at
//             17-27 % lane utilisation (measured), here at ~100 %.
// A finished ray is retired (samples[r] written, lane idle) once the FIFO head has passed its last queued leaf.
// Scheduling inside a wave: test when >= 64 pairs wait; refill when >= LV_REFILL_THRESHOLD lanes are idle; otherwise
// descend; flush partial batches only when nothing else can make progress.  Once the global queue is empty (drain) the
// rays left in a wave are finished together: nobody retires early, idle lanes take over stacked subtrees of busy ones.
// Resources: 31 KB LDS and <= 96 VGPRs -> 5 workgroups = 20 waves per CU; the kernel is VALU-issue-bound and needs that
// occupancy to hide the dependent node fetches (12 -> 16 -> 20 waves per CU: -16 %, -7 %).
// BAKE: the static prebaker's rays (VulkanAmbientOcclusionBaker.glsl:230-281).  G-buffer slot = parametrisation vertex *
// numTubeSubdivisions + subdivision with g0 = {ray origin, -}, g1 = {tangent, vertex}, g2 = {surface normal, subdivision};
// the reference draws the (subdivision, ray) samples of a vertex from ONE LCG stream seeded with tea(vertex, frame), so
// sample j starts from the stream advanced by 2 j steps: seed_j = A_j * seed_0 + C_j (lcgSkip[j] = {A_j, C_j}).
template <bool STATS, bool ANY_HIT, int PRIM, bool BAKE = false, bool LIT = false>
__global__ __launch_bounds__(LV_AO_BLOCK, LV_AO_MIN_WAVES) void k_ao_rays(const LvUniforms U, const LvSceneDev S,
                                                         const float4* __restrict__ gbuf, float* __restrict__ samples,
                                                         LvDevCounters* dc, const uint32_t* __restrict__ tileBase,
                                                         uint32_t numTiles, const LvAoLayout tileCapacity,
                                                         const uint2* __restrict__ lcgSkip = nullptr) {
    __shared__ unsigned s_stack[LV_AO_STACK_LDS * LV_AO_BLOCK];
    // LDS budget: 15 KB + 6 + 6 + 2 + 2 = 31 KB per workgroup -> 5 workgroups (20 waves) per CU; the kernel hides the
    // latency of its dependent node fetches with occupancy (measured: 12 -> 16 waves/CU -16 %, 16 -> 20 another -7 %)
    __shared__ float2 s_ray[3 * LV_AO_BLOCK];              // current ray of every lane: {o.xy}{o.z, d.x}{d.yz} (24 B)
    __shared__ float2 s_gen[3 * LV_AO_BLOCK];              // generated rays waiting for a lane
    __shared__ unsigned s_queue[LV_AO_BLOCK / LV_WAVE][LV_AO_QCAP];
    __shared__ unsigned long long s_key[LV_AO_BLOCK];      // best hit of every lane's ray

    const uint32_t spp = U.aoSamplesPerFrame;
    const unsigned long long total = (unsigned long long)(dc->aoCount) * spp;
    // rays a wave takes from the global queue per atomic: LV_AO_CHUNK, or half of it when the launch holds fewer than LV_AO_SMALL_CHUNKS
    // such chunks per persistent wave (a rank that owns 1/8 of the tiles: ~4 chunks of 128 per wave -- the last chunk of the slowest
    // wave is then a quarter of the kernel)
    const unsigned chunk = total < (unsigned long long)gridDim.x * (LV_AO_BLOCK / LV_WAVE) * LV_AO_CHUNK * LV_AO_SMALL_CHUNKS ? LV_AO_CHUNK_SMALL
                                                                                                                             : LV_AO_CHUNK;
    const bool capped = U.useCappedTubes != 0;
    const float radius = U.radius;
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    float2* rayW = &s_ray[3 * LV_WAVE * w];
    float2* genW = &s_gen[3 * LV_WAVE * w];
    unsigned* queueW = s_queue[w];
    unsigned long long* keyW = &s_key[LV_WAVE * w];
    const unsigned long long keyInit = ((unsigned long long)__float_as_uint(U.aoRadius) << 32) | 0xFFFFFFFFull;
    // literal roots: cull against best + r / |d| (AO directions are normalised: |d| = 1 to rounding, 1.001 covers it)
    // elliptic tubelets: bandWidth / |d| (lv_intersect_elliptic_tube's own-box rule)
    const float litSlack = PRIM == LV_PRIM_ELLIPTIC ? S.ellBandWidth * 1.001f : (PRIM == LV_PRIM_CAPSULE && LIT) ? radius * 1.001f : 0.0f;

    LvStackT<LV_AO_STACK_LDS, LV_AO_BLOCK> st;
    st.init(&s_stack[threadIdx.x], S.stackOverflow ? S.stackOverflow + (size_t(blockIdx.x) * LV_AO_BLOCK + threadIdx.x) : nullptr,
            gridDim.x * LV_AO_BLOCK);
    LvCounters cnt = {0, 0, 0, 0};
    unsigned long long phIt[3] = {0, 0, 0}, phLn[3] = {0, 0, 0};
    unsigned long long mayAxis = 0, mayBoth = 0, primHits = 0;

    // wave-uniform state
    unsigned long long chunkNext = 0, chunkEnd = 0, genBase = 0;
    unsigned genCount = 0, genPos = 0;      // rays in s_gen: [genPos, genCount)
    unsigned head = 0, tail = 0;            // leaf FIFO (absolute sequence numbers)
    bool sourceDry = (S.numSegs == 0) || (total == 0);
    // lane state
    bool hasRay = false, enq = false;
    unsigned long long r = 0;
    f3 inv = mk3(0, 0, 0), oi = mk3(0, 0, 0);
    float best = 0.0f;
    unsigned cur = LV_INVALID, lastSeq = 0;
    unsigned owner = lane; // lane whose ray this lane descends for: its own, except in the drain phase (below)
    // queued (owner, leaf) pairs that start a test phase: a little less than a full wave -- the stint ends sooner for the lanes whose
    // ray is finished and waits for it, at the price of test phases that run at 56 / 64 (LV_AO_TEST_BATCH*, measured)
    constexpr unsigned TB = PRIM == LV_PRIM_TRIANGLE ? LV_AO_TEST_BATCH_TRI : LV_AO_TEST_BATCH;

    while (true) {
        // ---- leaves reached in the previous step join the FIFO -- while fewer than 64 pairs wait, so that it never holds
        // more than 63 + 64 (LV_AO_QCAP = 128); with a full batch waiting the test phase below runs first
        {
            const bool isLeaf = tail - head < LV_WAVE && cur != LV_INVALID && (cur & LV_LEAF_BIT);
            const unsigned long long mL = __ballot(isLeaf);
            if (mL) {
                if (isLeaf) {
                    const unsigned idx = tail + unsigned(__popcll(mL & below));
                    queueW[idx % LV_AO_QCAP] = (owner << 26) | (cur & 0x03FFFFFFu);
                    lastSeq = idx;
                    enq = true;
                    cur = lv_pop_or_done(st);
                }
                tail += unsigned(__popcll(mL));
                if (tail - head < TB) continue; // a popped reference may be a leaf again
            }
        }
        const bool canRefill = (genPos < genCount) || !sourceDry;
        // DRAIN phase: no ray is left to hand out.  What remains in the wave is finished together, like one call of
        // lv_trace_closest: nobody retires early, lanes without work take over stacked subtrees of the lanes that still
        // descend (for THEIR rays: hits merge into the owner's key), and all samples are written at the end.  Without
        // it the kernel's last ~0.3 ms were a handful of lanes per wave walking their long rays alone -- a fixed cost that
        // weighs more the fewer tiles a GPU owns.
        const bool drain = !canRefill;
        // ---- retire rays whose traversal is finished and whose queued leaves have all been tested
        if (!drain && hasRay && cur == LV_INVALID && (!enq || int(head - lastSeq) > 0)) {
            const unsigned long long key = keyW[lane];
            float occ = 1.0f;
            if (key != keyInit) occ = U.aoUseDistance ? __uint_as_float(unsigned(key >> 32)) / U.aoRadius : 0.0f;
            samples[r] = occ;
            hasRay = false;
        }
        const unsigned long long mNode = __ballot(cur != LV_INVALID && !(cur & LV_LEAF_BIT));
        const unsigned long long mIdle = __ballot(!hasRay);
        const int nNode = __popcll(mNode), nIdle = __popcll(mIdle);
        const unsigned q = tail - head;

        if (q >= TB || (q > 0 && nNode == 0 && !(canRefill && nIdle > 0))) {
            // ---- test phase: one (owner, leaf) pair per lane
            const unsigned n = q < LV_WAVE ? q : LV_WAVE;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (STATS && lane == 0) { phIt[2]++; phLn[2] += n; }
            if (lane < n) {
                const unsigned e = queueW[(head + lane) % LV_AO_QCAP];
                const unsigned owner = e >> 26, leaf = e & 0x03FFFFFFu;
                const float2 r0 = rayW[3 * owner], r1 = rayW[3 * owner + 1], r2 = rayW[3 * owner + 2];
                if (STATS) cnt.prims += PRIM == LV_PRIM_TRIANGLE ? S.triLeafSize : 1u; // primitives tested
                float t; unsigned low;
                if (lv_leaf_test<PRIM, LIT ? 1 : 0>(S, leaf, mk3(r0.x, r0.y, r1.x), mk3(r1.y, r2.x, r2.y), radius, capped, t, low, 0.0f,
                                                    U.aoRadius)) {
                    if (t >= 0.0f && t <= U.aoRadius) { // traceAoRay: closest hit in [0, aoRadius], glsl:158-175
                        atomicMin(&keyW[owner], ((unsigned long long)__float_as_uint(t) << 32) | low);
                        if (STATS) primHits++;
                    }
                }
#ifndef LV_AO_IDLE_PROBE
                if (STATS && PRIM == LV_PRIM_CAPSULE) {
                    const float4 sa = S.segs[2 * size_t(leaf)], sb = S.segs[2 * size_t(leaf) + 1];
                    const f3 ro = mk3(r0.x, r0.y, r1.x), rd = mk3(r1.y, r2.x, r2.y);
                    const bool ma = lv_capsule_may_hit_axis(ro, rd, mk3(sa.x, sa.y, sa.z), mk3(sb.x, sb.y, sb.z), radius);
                    const bool mb = lv_capsule_may_hit_sphere(ro, rd, mk3(sa.x, sa.y, sa.z), mk3(sb.x, sb.y, sb.z), radius);
                    mayAxis += ma ? 1u : 0u;
                    mayBoth += (ma && mb) ? 1u : 0u;
                }
#endif
            }
            head += n;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            {
                const unsigned long long key = keyW[owner];
                best = __uint_as_float(unsigned(key >> 32)); // shrinks the slab interval of the following node steps
                if (ANY_HIT && key != keyInit) { cur = LV_INVALID; st.sp = 0; }
            }
            continue;
        }
        if (drain) {
            if (nNode == 0) { // FIFO empty (q == 0 here), nobody descends: the wave is done
                if (hasRay) {
                    const unsigned long long key = keyW[lane];
                    float occ = 1.0f;
                    if (key != keyInit) occ = U.aoUseDistance ? __uint_as_float(unsigned(key >> 32)) / U.aoRadius : 0.0f;
                    samples[r] = occ;
                }
                break;
            }
            if (nNode <= LV_HANDOVER_MAX_BUSY) { // subtree hand-over, as in lv_trace_closest (exchange slots: the idle s_gen)
                const bool idle = cur == LV_INVALID;
                const bool donor = !idle && st.sp > 0;
                const unsigned long long mFree = __ballot(idle), mDonor = __ballot(donor);
                const unsigned nPairs = min(unsigned(__popcll(mFree)), unsigned(__popcll(mDonor)));
                if (nPairs) {
                    if (donor) {
                        const unsigned rk = unsigned(__popcll(mDonor & below));
                        if (rk < nPairs) genW[rk] = make_float2(__uint_as_float(st.pop()), __uint_as_float(owner));
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    if (idle) {
                        const unsigned rk = unsigned(__popcll(mFree & below));
                        if (rk < nPairs) {
                            const float2 x = genW[rk];
                            cur = __float_as_uint(x.x);
                            owner = __float_as_uint(x.y);
                            const float2 r0 = rayW[3 * owner], r1 = rayW[3 * owner + 1], r2 = rayW[3 * owner + 2];
                            inv = mk3(1.0f / r1.y, 1.0f / r2.x, 1.0f / r2.y);
                            oi = mk3(r0.x * inv.x, r0.y * inv.y, r1.x * inv.z);
                            best = __uint_as_float(unsigned(keyW[owner] >> 32));
                            st.sp = 0;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    continue; // the reference taken over may be a leaf
                }
            }
        }
        if (canRefill && nIdle > 0 && (nIdle >= LV_REFILL_THRESHOLD || nNode < LV_NODE_MIN_ACTIVE)) {
            // ---- generate phase: keep s_gen stocked (all 64 lanes), then hand rays to the idle lanes
            if (genPos >= genCount) {
                if (chunkNext >= chunkEnd) {
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(&dc->aoQueueHead, (unsigned long long)chunk);
                    chunkNext = __shfl(base, 0, 64);
                    chunkEnd = chunkNext + chunk;
                    if (chunkEnd > total) chunkEnd = total;
                    if (chunkNext >= total) { chunkNext = chunkEnd = total; sourceDry = true; }
                }
                const unsigned long long left = chunkEnd - chunkNext;
                const unsigned n = left < LV_WAVE ? unsigned(left) : LV_WAVE;
                if (STATS && lane == 0 && n) { phIt[0]++; phLn[0] += n; }
                if (lane < n) {
                    const unsigned long long rr = LV_AO_RAY_INDEX(chunkNext + lane);
                    const uint32_t smpIdx = uint32_t(rr % spp);
                    const size_t slot = lv_ao_slot(tileBase, numTiles, tileCapacity, uint32_t(rr / spp));
                    const float4 g0 = gbuf[3 * slot + 0], g1 = gbuf[3 * slot + 1], g2 = gbuf[3 * slot + 2];
                    const uint32_t pix = __float_as_uint(g1.w);
                    const f3 pos = mk3(g0.x, g0.y, g0.z), T = mk3(g1.x, g1.y, g1.z), N = mk3(g2.x, g2.y, g2.z);
                    const f3 B = cross3(N, T);
                    uint32_t seed;
                    if (BAKE) {
                        const uint32_t sub = __float_as_uint(g2.w);
                        const uint2 skip = lcgSkip[2u * (sub * spp + smpIdx)];
                        seed = skip.x * lv_tea(pix /* = vertex */, U.aoFrameNumber) + skip.y;
                    } else {
                        seed = lv_tea(pix, U.aoGlobalFrameNumber * spp + smpIdx);
                    }
                    const float xi0 = lv_rnd(seed), xi1 = lv_rnd(seed);
                    float sn, cs;
                    lv_sincos2pi(xi1, sn, cs); // sampleHemisphere, glsl:151-156
                    const float rs = sqrtf(1.0f - xi0 * xi0);
                    const f3 smp = mk3(cs * rs, sn * rs, xi0);
                    const f3 dirU = mk3((T.x * smp.x + B.x * smp.y) + N.x * smp.z, (T.y * smp.x + B.y * smp.y) + N.y * smp.z,
                                        (T.z * smp.x + B.z * smp.y) + N.z * smp.z);
                    const f3 d = norm3(dirU);
                    const f3 o = BAKE ? pos : pos + d * g0.w;
                    genW[3 * lane] = make_float2(o.x, o.y);
                    genW[3 * lane + 1] = make_float2(o.z, d.x);
                    genW[3 * lane + 2] = make_float2(d.y, d.z);
                }
                genBase = chunkNext;
                chunkNext += n;
                genCount = n;
                genPos = 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            const unsigned avail = genCount - genPos;
            const unsigned k = unsigned(nIdle) < avail ? unsigned(nIdle) : avail;
            if (!hasRay) {
                const unsigned rank = unsigned(__popcll(mIdle & below));
                if (rank < k) {
                    const unsigned gs = genPos + rank;
                    const float2 r0 = genW[3 * gs], r1 = genW[3 * gs + 1], r2 = genW[3 * gs + 2];
                    rayW[3 * lane] = r0;
                    rayW[3 * lane + 1] = r1;
                    rayW[3 * lane + 2] = r2;
                    keyW[lane] = keyInit;
                    inv = mk3(1.0f / r1.y, 1.0f / r2.x, 1.0f / r2.y);
                    oi = mk3(r0.x * inv.x, r0.y * inv.y, r1.x * inv.z);
                    best = U.aoRadius;
                    r = LV_AO_RAY_INDEX(genBase + gs);
                    owner = lane;
                    cur = 0;
                    st.sp = 0;
                    enq = false;
                    hasRay = true;
                    if (STATS) cnt.rays++;
                }
            }
            genPos += k;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            continue;
        }
        if (nNode > 0) {
            // ---- descend phase: tight loop of node steps; leaves met on the way join the FIFO.  Leave it only when a
            // full batch of leaf tests waits or nobody descends any more (leaving earlier for a retire/refill pass
            // was measured slower: the per-pass bookkeeping outweighs the idle lanes).
            const int stay = LV_AO_STAY;
            int nNow;
            do {
                if (STATS && lane == 0) { phIt[1]++; }
                if (STATS && !(cur & LV_LEAF_BIT)) phLn[1]++;
#ifdef LV_AO_IDLE_PROBE // tools/variants.py: why lanes idle in the descend loop -- (a) their ray waits for queued leaf tests, (b) no ray
                if (STATS) {
                    const unsigned long long mW = __ballot(hasRay && cur == LV_INVALID), mE = __ballot(!hasRay);
                    if (lane == 0) { mayAxis += unsigned(__popcll(mW)); mayBoth += unsigned(__popcll(mE)); }
                }
#endif
                if (!(cur & LV_LEAF_BIT)) cur = lv_node_step<STATS, LV_AO_ORDERED>(S, cur, oi, inv, 0.0f - litSlack, best + litSlack, st, cnt);
                const bool isLeaf = cur != LV_INVALID && (cur & LV_LEAF_BIT);
                const unsigned long long mL = __ballot(isLeaf);
                if (mL) {
                    if (isLeaf) {
                        const unsigned idx = tail + unsigned(__popcll(mL & below));
                        queueW[idx % LV_AO_QCAP] = (owner << 26) | (cur & 0x03FFFFFFu);
                        lastSeq = idx;
                        enq = true;
                        cur = lv_pop_or_done(st);
                    }
                    tail += unsigned(__popcll(mL));
                }
                nNow = __popcll(__ballot(!(cur & LV_LEAF_BIT)));
            } while (tail - head < TB && nNow >= (drain ? LV_HANDOVER_MAX_BUSY + 1 : stay));
            continue;
        }
        if (nIdle == LV_WAVE && !canRefill) break; // nothing alive, nothing left to fetch
    }
    if (STATS) {
        lv_flush_counters(cnt, dc, true);
        for (int k = 0; k < 3; k++) {
            const unsigned long long it = lv_wave_sum_u64(phIt[k]), ln = lv_wave_sum_u64(phLn[k]);
            if (lane == 0) { atomicAdd(&dc->aoPhaseIters[k], it); atomicAdd(&dc->aoPhaseLanes[k], ln); }
        }
        const unsigned long long sa = lv_wave_sum_u64(mayAxis), sb = lv_wave_sum_u64(mayBoth), sh = lv_wave_sum_u64(primHits);
        if (lane == 0) { atomicAdd(&dc->aoPrimMayAxis, sa); atomicAdd(&dc->aoPrimMayBoth, sb); atomicAdd(&dc->aoPrimHits, sh); }
    }
}

template <bool BAKE>
__global__ __launch_bounds__(LV_BLOCK) void k_ao_reduce(const LvUniforms U, const float4* __restrict__ gbuf,
                                                        const float* __restrict__ samples, const float* aoIn, float* ao,
                                                        const LvDevCounters* dc, const uint32_t* __restrict__ tileBase,
                                                        uint32_t numTiles, const LvAoLayout tileCapacity) {
    const uint32_t slot = blockIdx.x * LV_BLOCK + threadIdx.x; // compacted-pixel ordinal (= row of `samples`)
    if (slot >= dc->aoCount) return;
    const uint32_t spp = U.aoSamplesPerFrame;
    float aoFactor = 0.0f; // summed strictly in sample order, like the reference's loop (glsl:288-306)
    if ((spp & 3u) == 0u) {
        // 16-byte loads: the lanes of a wave read rows that are spp * 4 bytes apart, so every load instruction touches
        // 64 different cache lines whatever its width -- dwordx4 quarters the number of L1 accesses
        const float4* row = reinterpret_cast<const float4*>(samples + size_t(slot) * spp);
        for (uint32_t s = 0; s < spp / 4u; s++) {
            const float4 v = row[s];
            aoFactor += v.x; aoFactor += v.y; aoFactor += v.z; aoFactor += v.w;
        }
    } else {
        for (uint32_t s = 0; s < spp; s++) aoFactor += samples[size_t(slot) * spp + s];
    }
    aoFactor /= float(spp);
    // screen space: the pixel of the compacted slot; prebaker: ambientOcclusionFactors[subdiv + N * vertex] = the slot
    const uint32_t pix = BAKE ? slot : __float_as_uint(gbuf[3 * lv_ao_slot(tileBase, numTiles, tileCapacity, slot) + 1].w);
    // aoIn == ao except with a halo: the 1-pixel rings of neighbouring tiles overlap, a pixel may then be processed twice in
    // one pass, and the running mean must read the PREVIOUS pass' image to stay idempotent (lv_run_ao)
    if (U.aoFrameNumber != 0) aoFactor = mixf(aoIn[pix], aoFactor, 1.0f / float(U.aoFrameNumber + 1));
    ao[pix] = aoFactor;
}

// The same reduction with coalesced loads (spp a multiple of 4): a workgroup of 128 threads owns 128 consecutive rows of
// `samples`; per chunk of 64 samples the rows' 256-byte pieces are copied into LDS (16 lanes per row, one float4 each) with a row
// pitch of 65 floats (conflict-free column walks), then every thread adds ITS row's chunk in sample order into its accumulator --
// the same sequence of additions as the loop above, so the same bits.  (k_ao_reduce reads every row 16 B at a time with one
// cache line per lane: 75 us for config 3's 91 MB, against 91 MB / HBM rate.)  Persistent: blocks stride over dc->aoCount.
#define LV_REDUCE_ROWS 128u
__global__ __launch_bounds__(LV_REDUCE_ROWS) void k_ao_reduce_rows(const LvUniforms U, const float4* __restrict__ gbuf,
                                                                   const float* __restrict__ samples, const float* aoIn, float* ao,
                                                                   const LvDevCounters* dc, const uint32_t* __restrict__ tileBase,
                                                                   uint32_t numTiles, const LvAoLayout tileCapacity) {
    __shared__ float s_rows[LV_REDUCE_ROWS * 65u];
    const uint32_t count = dc->aoCount, spp = U.aoSamplesPerFrame, t = threadIdx.x;
    for (uint32_t row0 = blockIdx.x * LV_REDUCE_ROWS; row0 < count; row0 += gridDim.x * LV_REDUCE_ROWS) {
        const uint32_t rows = min(LV_REDUCE_ROWS, count - row0);
        float aoFactor = 0.0f;
        for (uint32_t c0 = 0; c0 < spp; c0 += 64u) {
            const uint32_t cols4 = min(64u, spp - c0) / 4u; // float4s per row in this chunk
            __syncthreads();
            for (uint32_t idx = t; idx < rows * cols4; idx += LV_REDUCE_ROWS) {
                const uint32_t r = idx / cols4, c4 = idx % cols4;
                const float4 v = *reinterpret_cast<const float4*>(samples + size_t(row0 + r) * spp + c0 + 4u * c4);
                float* d = s_rows + r * 65u + 4u * c4;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
            __syncthreads();
            if (t < rows)
                for (uint32_t j = 0; j < 4u * cols4; j++) aoFactor += s_rows[t * 65u + j];
        }
        if (t < rows) {
            const uint32_t slot = row0 + t;
            aoFactor /= float(spp);
            const uint32_t pix = __float_as_uint(gbuf[3 * lv_ao_slot(tileBase, numTiles, tileCapacity, slot) + 1].w);
            if (U.aoFrameNumber != 0) aoFactor = mixf(aoIn[pix], aoFactor, 1.0f / float(U.aoFrameNumber + 1));
            ao[pix] = aoFactor;
        }
    }
}

// One a-trous pass of the EAW denoiser (EAWDenoise.glsl) over the AO image: colorTexture = vec4(ao, ao, ao, 1) -- the three
// colour channels stay equal and alpha stays 1 through every pass, so one float per pixel carries the image.
//   COMPUTE = true   EAWDenoise.Compute (:128-292; eaw_denoiser_use_shared_memory = true, the default): B-spline kernel
//                    {1, 2/3, 1/6}, neighbours outside the image skipped, ONE exp of the summed exponents, colour term x step width
//   COMPUTE = false  EAWDenoise.Fragment (:16-126): Gaussian kernel, clamp-to-edge sampling, min(exp(.), 1) per enabled feature
// One thread per pixel of the (dilated) tiles; "inside" refers to the whole viewport, so tiles reproduce the full frame.
struct LvEawParams {
    float phiColor, phiPosition, phiNormal;
    uint32_t useColor, usePosition, useNormal;
    int stepWidth;
};
template <bool COMPUTE>
__global__ __launch_bounds__(LV_BLOCK) void k_eaw_pass(const LvUniforms U, const LvTiles T, const LvEawParams E,
                                                       const float* __restrict__ src, float* __restrict__ dst,
                                                       const float4* __restrict__ featNormal,
                                                       const float4* __restrict__ featPosition) {
    const uint32_t perTile = T.tileW * T.tileH;
    const uint64_t gid = uint64_t(blockIdx.x) * LV_BLOCK + threadIdx.x;
    if (gid >= uint64_t(T.numTiles) * perTile) return;
    const uint32_t tile = uint32_t(gid / perTile), rem = uint32_t(gid % perTile);
    const uint32_t gxu = T.tilesXY[2 * tile] + rem % T.tileW, gyu = T.tilesXY[2 * tile + 1] + rem / T.tileW;
    if (gxu >= U.width || gyu >= U.height) return;
    const int gx = int(gxu), gy = int(gyu), W = int(U.width), H = int(U.height);
    const size_t ci = size_t(gy) * U.width + gx;
    const float centerColor = src[ci];
    const float4 cP = E.usePosition ? featPosition[ci] : make_float4(0, 0, 0, 0);
    const float4 cN = E.useNormal ? featNormal[ci] : make_float4(0, 0, 0, 0);
    auto dist4 = [](float4 a, float4 b) {
        const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, dw = a.w - b.w;
        return ((dx * dx + dy * dy) + dz * dz) + dw * dw;
    };
    float sum, accumW;
    if (COMPUTE) {
        const float kernelValues[3] = {1.0f, 2.0f / 3.0f, 1.0f / 6.0f};
        accumW = kernelValues[0] * kernelValues[0];
        sum = centerColor * accumW;
        for (int y = -2; y <= 2; ++y) {
            for (int x = -2; x <= 2; ++x) {
                const int ox = gx + x * E.stepWidth, oy = gy + y * E.stepWidth;
                const bool inside = ox >= 0 && oy >= 0 && ox < W && oy < H;
                if (!inside || (x == 0 && y == 0)) continue;
                const size_t oi = size_t(oy) * U.width + ox;
                const float kernelValue = kernelValues[x < 0 ? -x : x] * kernelValues[y < 0 ? -y : y];
                const float offsetColor = src[oi];
                float e = 0.0f;
                if (E.useColor) {
                    const float d = centerColor - offsetColor;
                    const float distColor = ((d * d + d * d) + d * d) + 0.0f * 0.0f;
                    e = e - (distColor * float(E.stepWidth)) / E.phiColor;
                }
                if (E.usePosition) e = e - dist4(cP, featPosition[oi]) / E.phiPosition;
                if (E.useNormal) e = e - dist4(cN, featNormal[oi]) / E.phiNormal;
                const float weight = expf(e);
                sum += (offsetColor * weight) * kernelValue;
                accumW += weight * kernelValue;
            }
        }
    } else {
        sum = 0.0f; accumW = 0.0f;
        for (int i = 0; i < 25; i++) {
            const float x = float(i % 5 - 2), y = float(i / 5 - 2);
            const float kernelValue = expf(-(x * x + y * y) / 2.0f);
            const int ox = min(max(gx + (i % 5 - 2) * E.stepWidth, 0), W - 1);
            const int oy = min(max(gy + (i / 5 - 2) * E.stepWidth, 0), H - 1);
            const size_t oi = size_t(oy) * U.width + ox;
            const float offsetColor = src[oi];
            float weight = 1.0f;
            if (E.useColor) {
                const float d = centerColor - offsetColor;
                const float distColor = ((d * d + d * d) + d * d) + 0.0f * 0.0f;
                weight *= fminf(expf(-distColor / E.phiColor), 1.0f);
            }
            if (E.usePosition) weight *= fminf(expf(-dist4(cP, featPosition[oi]) / E.phiPosition), 1.0f);
            if (E.useNormal) weight *= fminf(expf(-dist4(cN, featNormal[oi]) / E.phiNormal), 1.0f);
            sum += (offsetColor * weight) * kernelValue;
            accumW += weight * kernelValue;
        }
    }
    dst[ci] = sum / accumW;
}

// VulkanAmbientOcclusionBaker.glsl:110-131,231-262: interpolated line point of every parametrisation vertex and the ray
// origin / frame of each of its tube subdivisions, written in the G-buffer format of k_ao_rays.
__global__ __launch_bounds__(LV_BLOCK) void k_bake_setup(const lv_line_point* __restrict__ linePoints, uint32_t numLinePoints,
                                                         const float* __restrict__ samplingLocations,
                                                         uint32_t numParametrizationVertices, uint32_t numTubeSubdivisions,
                                                         float lineRadius, float4* __restrict__ gbuf, uint32_t useBands,
                                                         float bandRadius, float minBandThickness) {
    const uint32_t slot = blockIdx.x * LV_BLOCK + threadIdx.x;
    if (slot >= numParametrizationVertices * numTubeSubdivisions) return;
    const uint32_t vertex = slot / numTubeSubdivisions, sub = slot % numTubeSubdivisions;
    const float samplingLocation = samplingLocations[vertex];
    const uint32_t lowerIdx = uint32_t(samplingLocation);
    const uint32_t upperIdx = min(lowerIdx + 1u, numLinePoints - 1u);
    const float f = samplingLocation - floorf(samplingLocation);
    const lv_line_point& lo = linePoints[lowerIdx];
    const lv_line_point& up = linePoints[upperIdx];
    auto ld = [](const float* p) { return mk3(p[0], p[1], p[2]); };
    auto mix3 = [&](f3 a, f3 b) { return mk3(mixf(a.x, b.x, f), mixf(a.y, b.y, f), mixf(a.z, b.z, f)); };
    const f3 binormalLower = cross3(ld(lo.lineTangent), ld(lo.lineNormal));
    const f3 binormalUpper = cross3(ld(up.lineTangent), ld(up.lineNormal));
    const f3 position = mix3(ld(lo.linePosition), ld(up.linePosition));
    const f3 tangent = norm3(mix3(ld(lo.lineTangent), ld(up.lineTangent)));
    const f3 normal = norm3(mix3(ld(lo.lineNormal), ld(up.lineNormal)));
    const f3 binormal = norm3(mix3(binormalLower, binormalUpper));
    float sinAngle, cosAngle;
    lv_sincos2pi(float(sub) / float(numTubeSubdivisions), sinAngle, cosAngle);
    f3 surfaceNormal = cosAngle * normal + sinAngle * binormal;
    f3 rayOrigin = position + (lineRadius + 1e-6f) * surfaceNormal;
    if (useBands) {   // USE_BANDS, "bands with minimum thickness" (glsl:233-256): elliptic cross-section, pushed out by 1e-3
        surfaceNormal = norm3(cosAngle * normal + (minBandThickness * sinAngle) * binormal);
        rayOrigin = position + (bandRadius + 1e-3f) * ((minBandThickness * cosAngle) * normal + sinAngle * binormal);
    }
    gbuf[3 * size_t(slot) + 0] = make_float4(rayOrigin.x, rayOrigin.y, rayOrigin.z, 0.0f);
    gbuf[3 * size_t(slot) + 1] = make_float4(tangent.x, tangent.y, tangent.z, __uint_as_float(vertex));
    gbuf[3 * size_t(slot) + 2] = make_float4(surfaceNormal.x, surfaceNormal.y, surfaceNormal.z, __uint_as_float(sub));
}

__global__ __launch_bounds__(LV_BLOCK) void k_fill_f32(float* p, float v, size_t n) {
    size_t i = size_t(blockIdx.x) * LV_BLOCK + threadIdx.x;
    if (i < n) p[i] = v;
}

// ================================================================ PPLL
// The ray interval [tMin, tMax] of every pixel is cut into numSlices depth slices (boundaries spread evenly over the
// ray's passage through the scene box); workgroup = (16x16 pixel block, slice).  All-hits traversal cannot cull, so the
// slices cost no extra traversal, but a wave in front of the dense core of a data set -- hundreds of fragments per pixel,
// each one shaded -- no longer forms a multi-millisecond critical path while the rest of the GPU idles: its work is spread
// over numSlices workgroups.  A slice accepts t in [lo, hi) (the last one up to tMax inclusive), so every fragment is
// produced exactly once; each workgroup builds partial lists in LDS and splices them into the pixel's global list.
template <bool STATS, int PRIM = LV_PRIM_CAPSULE, int BANDS = LV_SHADE_PLAIN>
__global__ __launch_bounds__(LV_BLOCK, PRIM == LV_PRIM_PRISM ? LV_PRISM_MIN_WAVES : LV_GATHER_MIN_WAVES) void k_ppll_gather(const LvUniforms U, const LvSceneDev S, const LvTiles T,
                                                          uint32_t* __restrict__ nodes, uint32_t* __restrict__ startOffset,
                                                          uint32_t* __restrict__ fragCount, LvDevCounters* dc,
                                                          uint32_t numSlices, uint32_t poolSlots) {
    __shared__ unsigned s_stack[LV_STACK_LDS * LV_BLOCK];
    __shared__ uint32_t s_head[LV_BLOCK];  // head of this workgroup's partial list of every thread's pixel
    __shared__ uint32_t s_tail[LV_BLOCK];  // its first inserted node (whose `next` is patched when splicing)
    __shared__ uint32_t s_count[LV_BLOCK]; // fragments of every thread's pixel in this slice
    __shared__ unsigned s_allocBase[LV_BLOCK / LV_WAVE], s_allocLeft[LV_BLOCK / LV_WAVE]; // per-wave chunk of node slots
    LV_COOP_SHARED(LV_BLOCK / LV_WAVE);
    LV_COOP_MEM(cm);
    LV_HITQ_SHARED(LV_BLOCK / LV_WAVE);
    LV_HITQ_MEM(hq);
    __shared__ unsigned s_prismQueue[PRIM == LV_PRIM_PRISM ? LV_BLOCK / LV_WAVE : 1][PRIM == LV_PRIM_PRISM ? LV_HITQ_CAP : 1];
    LvPixel px;
    const uint32_t slice = blockIdx.x % numSlices;
    if (!lv_block_pixel(U, T, px, blockIdx.x / numSlices)) return;
    if (PRIM == LV_PRIM_PRISM) hq.prismQueue = s_prismQueue[threadIdx.x >> 6];
    const unsigned long long tg0 = lv_group_clock();
    LvCounters cnt = {0, 0, 0, 0};
    // (the slices of a pixel are ONE ray: lv_trace_all counts a ray per active call, corrected below)
    const unsigned waveBase = threadIdx.x & ~63u;
    s_head[threadIdx.x] = 0xFFFFFFFFu;
    s_tail[threadIdx.x] = 0xFFFFFFFFu;
    s_count[threadIdx.x] = 0u;
    if ((threadIdx.x & 63u) == 0u) { s_allocBase[threadIdx.x >> 6] = 0u; s_allocLeft[threadIdx.x >> 6] = 0u; }
    const float aoTexel = (px.inView && U.useAmbientOcclusion && !U.aoPrebaked) ? S.ao[size_t(px.y) * U.width + px.x] : 1.0f;
    f3 o, d;
    lv_primary_ray(U, px.x, px.y, 0.5f, 0.5f, o, d);
    const float tMin = 0.0001f, tMax = 1000.0f;
    float lo = tMin, hi = __uint_as_float(__float_as_uint(tMax) + 1u); // t <= tMax  <=>  t < nextafter(tMax)
    if (numSlices > 1u && S.numSegs != 0) {
        // passage of the ray through the scene box (= the root node's grid, origin + [0, 255] * scale)
        const float4 q0 = S.nodes[0], q1 = S.nodes[1];
        const f3 bmin = mk3(q0.x, q0.y, q0.z);
        const f3 bmax = mk3(q0.x + 255.0f * q0.w, q0.y + 255.0f * q1.x, q0.z + 255.0f * q1.y);
        const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        const float tx0 = (bmin.x - o.x) * inv.x, tx1 = (bmax.x - o.x) * inv.x;
        const float ty0 = (bmin.y - o.y) * inv.y, ty1 = (bmax.y - o.y) * inv.y;
        const float tz0 = (bmin.z - o.z) * inv.z, tz1 = (bmax.z - o.z) * inv.z;
        float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fminf(tz0, tz1));
        float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
        tn = fminf(fmaxf(tn, tMin), tMax);
        tf = fminf(fmaxf(tf, tn), tMax);
        // interior boundaries only steer the load balance; b_0 = tMin and b_numSlices = tMax keep the partition exact
        const float w = (tf - tn) / float(numSlices);
        if (slice > 0u) lo = tn + w * float(slice);
        if (slice + 1u < numSlices) hi = tn + w * float(slice + 1u);
    }
    const bool active = px.inView && lo < hi;
    uint32_t prismDropped = 0u;
    if (STATS && active && slice != 0u && S.numSegs != 0) cnt.rays--;
    // Fragments of a pixel are produced by whichever lane is handed the (pixel, segment) hit: the lane shades with the
    // owner's ray + AO texel and links the node with an LDS atomic exchange on the owner's list head (the reference's
    // atomicExchange(startOffset[pixel]), LinkedListGather.glsl:55, kept in LDS until the slice is finished).
    // second payload word: the owner's pixel -- the lane that shades a fragment rebuilds the rays through the 2 x 2 quad partners
    // for fwidth(ribbonPosition) of the raster fragment colour (LvRasterQuad)
    lv_trace_all<STATS, false, PRIM>(S, U.radius, U.useCappedTubes != 0, active, o, d, lo, hi, aoTexel,
                        __uint_as_float(px.x | (px.y << 16)), lv_stack_mem(s_stack, S.stackOverflow), cm, hq, cnt,
                        [&](unsigned owner, uint32_t leaf, float t, int kind, f3 ro, f3 rd, float ownerAo, float ownerPixel) {
        // The rasterised prism: this kernel is the RASTERISER -- it finds which triangles cover which pixels, gives every covered
        // triangle its node and links it into the pixel's list (in the deterministic order of the owner's wave) -- and leaves the
        // FRAGMENT STAGE to k_ppll_shade_prism: the node carries {pixel, segment leaf | triangle << 26} until that kernel replaces
        // the two words by {colour, depth}.  A wave in front of the dense core covers two orders of magnitude more fragments than
        // the average wave; with the fragment stage (~3000 instructions) inline its serial shade batches were the kernel's critical
        // path, as a second pass over the node pool the fragments are shaded one per lane, evenly over the whole GPU.
        uint32_t word0, word1;
        if (PRIM == LV_PRIM_PRISM) {
            word0 = __float_as_uint(ownerPixel);
            word1 = leaf | (unsigned(kind) << 26);
        } else {
            LvHit h; h.t = t; h.leaf = leaf; h.kind = kind; h.found = true;
            float hitT;
            const bool rasterApply = U.ppllRasterColour != 0u;
            const uint32_t pxy = __float_as_uint(ownerPixel);
            const LvRasterQuad rq = lv_make_raster_quad(U, pxy & 0xFFFFu, pxy >> 16);
            f4 color = PRIM == LV_PRIM_ELLIPTIC ? lv_shade_hit_elliptic(S, U, ownerAo, ro, rd, h, hitT, true, rq, rasterApply)
                                                : lv_shade_hit<BANDS>(S, U, ownerAo, ro, rd, h, hitT, true, rq, rasterApply);
            if (STATS) cnt.hits++;
            if (color.w < 0.001f) return; // gatherFragment: discard, LinkedListGather.glsl:34
            word0 = lv_pack_unorm4x8(color);
            word1 = __float_as_uint(hitT);
        }
        // wave-aggregated node allocation: slots come from a per-wave chunk; one global atomic per LV_PPLL_CHUNK fragments
        // instead of one per batch (a returning atomic on one address costs microseconds under load and sat on the
        // critical path of the waves in front of dense geometry).  A batch that does not fit takes the rest of the old
        // chunk first and continues in a new one, so only the tail a wave leaves behind when it exits stays unused; the
        // pool carries that much slack (poolSlots >= ppllLinkedListSize + waves * LV_PPLL_CHUNK), i.e. no fragment the
        // reference's exact atomicAdd(fragCounter) allocator would have stored is dropped here.
        const unsigned long long mask = __ballot(1);
        const unsigned lane = lv_lane();
        const int leader = __ffsll((long long)mask) - 1;
        unsigned base = 0, left = 0, base2 = 0;
        if (int(lane) == leader) {
            const unsigned n = unsigned(__popcll(mask)), w = threadIdx.x >> 6;
            base = s_allocBase[w]; left = s_allocLeft[w];
            if (left < n) {
                base2 = atomicAdd(&dc->fragAlloc, (unsigned)LV_PPLL_CHUNK);
                s_allocBase[w] = base2 + (n - left);
                s_allocLeft[w] = LV_PPLL_CHUNK - (n - left);
            } else {
                s_allocBase[w] = base + n;
                s_allocLeft[w] = left - n;
            }
        }
        base = __shfl(base, leader, 64); left = __shfl(left, leader, 64); base2 = __shfl(base2, leader, 64);
        const unsigned rank = unsigned(__popcll(mask & ((1ull << lane) - 1ull)));
        const uint32_t insertIndex = rank < left ? base + rank : base2 + (rank - left);
        if (PRIM == LV_PRIM_PRISM) {
            // record {pixel, leaf | triangle << 26, rank}: rank = how many fragments the pixel had before this one, in the
            // deterministic order of the owner's wave -- the pixel's fragments end up in ONE contiguous run of the fragment array
            // (exclusive scan of the per-pixel counts -> offset; k_ppll_shade_prism writes to offset + rank), not in a linked list
            // whose 12-B nodes are scattered over 128-B lines (the resolve pass fetched 5.8 x the bytes it used)
            bool stored = false;
            if (insertIndex < poolSlots) {
                const uint32_t rk = atomicAdd(&s_count[waveBase + owner], 1u);
                if (rk < 0xFFFFu) {
                    nodes[3 * size_t(insertIndex) + 0] = word0;
                    nodes[3 * size_t(insertIndex) + 1] = word1;
                    nodes[3 * size_t(insertIndex) + 2] = rk;
                    stored = true;
                } else {   // (the per-pixel count shares its word with the count of discarded fragments: 16 bits each)
                    atomicSub(&s_count[waveBase + owner], 1u);
                    nodes[3 * size_t(insertIndex) + 0] = 0u;
                    nodes[3 * size_t(insertIndex) + 1] = LV_PPLL_DEAD;
                }
            }
            if (!stored) prismDropped++;   // no record: the fragment stage never sees it, but the reference's fragCounter counts it
            return;
        }
        atomicAdd(&s_count[waveBase + owner], 1u);
        if (insertIndex < poolSlots) {
            const uint32_t next = atomicExch(&s_head[waveBase + owner], insertIndex);
            if (next == 0xFFFFFFFFu) s_tail[waveBase + owner] = insertIndex;
            nodes[3 * size_t(insertIndex) + 0] = word0;
            nodes[3 * size_t(insertIndex) + 1] = word1;
            nodes[3 * size_t(insertIndex) + 2] = next;
        }
    }, [](unsigned) {});
    if (PRIM == LV_PRIM_PRISM) {
        // the slots this wave reserved but did not use are part of the range the fragment stage walks: mark them dead
        const unsigned w = threadIdx.x >> 6;
        const unsigned base = s_allocBase[w], left = s_allocLeft[w];
        for (unsigned k = lv_lane(); k < left; k += LV_WAVE)
            if (base + k < poolSlots) { nodes[3 * size_t(base + k) + 0] = 0u; nodes[3 * size_t(base + k) + 1] = LV_PPLL_DEAD; }
    }
    const uint32_t numFrags = s_count[threadIdx.x];
    uint32_t total = 0;
    if (PRIM == LV_PRIM_PRISM) {
        // one slice, one workgroup per pixel: the count is final (k_ppll_clear zeroed the pixels without fragments)
        if (px.inView && numFrags > 0u) fragCount[lv_ppll_addr(px.x, px.y, U.ppllPaddedW, U.ppllTileW, U.ppllTileH)] = numFrags;
    } else if (px.inView && numFrags > 0u) {
        const uint32_t addr = lv_ppll_addr(px.x, px.y, U.ppllPaddedW, U.ppllTileW, U.ppllTileH);
        const uint32_t head = s_head[threadIdx.x], tail = s_tail[threadIdx.x];
        if (head != 0xFFFFFFFFu) { // splice the partial list in front of what other slices linked so far
            const uint32_t old = atomicExch(&startOffset[addr], head);
            nodes[3 * size_t(tail) + 2] = old;
        }
        total = atomicAdd(&fragCount[addr], numFrags) + numFrags; // the last slice to arrive sees the pixel's total
    }
    uint32_t m = total, sum = px.inView ? numFrags : 0u;
#pragma unroll
    for (int ofs = 32; ofs > 0; ofs >>= 1) {
        m = max(m, (uint32_t)__shfl_xor(m, ofs, 64));
        sum += (uint32_t)__shfl_xor(sum, ofs, 64);
    }
    if (PRIM == LV_PRIM_PRISM) {
        // fragCounter / maxDepthComplexity count KEPT fragments: the fragment stage adds the ones it keeps (and counts the ones it
        // discards in the upper half of the pixel's count word: the resolve pass reads the maximum), here only those without a record
        sum = prismDropped;
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) sum += (uint32_t)__shfl_xor(sum, ofs, 64);
        m = 0u;
    }
    if (lv_lane() == 0 && m > 0) atomicMax(&dc->maxDepthComplexity, m);
    if (lv_lane() == 0 && sum > 0) atomicAdd(&dc->fragCounter, sum); // fragCounter of the reference: every fragment counts
    lv_group_cost_add(T, px, tg0);
    if (STATS) lv_flush_counters(cnt, dc);
}

// start of a pixel's run in the fragment array (k_ppll_scan: offsets relative to the pixel's block of LV_SCAN_ITEMS addresses + the block's base)
__device__ __forceinline__ uint32_t lv_run_start(const uint32_t* __restrict__ startOffset, const uint32_t* __restrict__ blockBase,
                                                 uint32_t addr) {
    return startOffset[addr] + blockBase[addr / LV_SCAN_ITEMS];
}

// Fragment stage of ppll_fragment_source = raster_prism: one lane per record of the coverage kernel (k_ppll_gather<LV_PRIM_PRISM>),
// grid-stride over the record pool -- LinePassGeometryShaderTubes.glsl:732-1129 on the perspective-correct inputs (lv_shade_prism) +
// the store of gatherFragment (LinkedListGather.glsl:33-72): {colour, depth} goes to slot pixelOffset[pixel] + rank of the fragment
// array, so that every pixel's fragments form one contiguous run in the order the coverage kernel met them.  A fragment the shader
// discards (alpha < 0.001, :34) or the `kept` rules reject leaves a DEAD entry {0, LV_PPLL_DEAD} that the resolve pass steps over,
// and is counted in the upper 16 bits of the pixel's count word.  No atomics on the pool, none per kept fragment.
template <bool STATS, int SHADE = LV_SHADE_PLAIN, int FAST = 0>
__global__ __launch_bounds__(LV_BLOCK, LV_PRISM_SHADE_MIN_WAVES) void k_ppll_shade_prism(const LvUniforms U, const LvSceneDev S,
                                                                   const uint32_t* __restrict__ records, uint2* __restrict__ frags,
                                                                   const uint32_t* __restrict__ pixelOffset,
                                                                   const uint32_t* __restrict__ blockBase, uint32_t* __restrict__ fragCount,
                                                                   LvDevCounters* dc, uint32_t poolSlots) {
    __shared__ float s_prismRing[4 * LV_PRISM_MAX_SUBDIV];
    if (threadIdx.x < LV_PRISM_MAX_SUBDIV) {   // ring tables [c | s | cp | sn] in LDS: the triangle a lane shades is a per-lane index
        s_prismRing[threadIdx.x] = S.prism.c[threadIdx.x];
        s_prismRing[LV_PRISM_MAX_SUBDIV + threadIdx.x] = S.prism.s[threadIdx.x];
        s_prismRing[2 * LV_PRISM_MAX_SUBDIV + threadIdx.x] = S.prism.cp[threadIdx.x];
        s_prismRing[3 * LV_PRISM_MAX_SUBDIV + threadIdx.x] = S.prism.sn[threadIdx.x];
    }
    __syncthreads();
    const uint32_t numSlots = min(dc->fragAlloc, poolSlots);   // the chunk allocator's high-water mark
    const f3 o = mk3(U.camPos[0], U.camPos[1], U.camPos[2]);
    const float tLo = 0.0001f, tHi = __uint_as_float(__float_as_uint(1000.0f) + 1u); // the gather's ray interval [tMin, tMax]
    unsigned long long hits = 0;
    uint32_t localSum = 0u, localDead = 0u;
    // 256-slot items, item (sweep s, position p) to workgroup (p - 5 s) mod G: the dead tails of the rasteriser's record chunks end on
    // chunk boundaries, i.e. at fixed positions modulo the chunk size -- a fixed position per workgroup gave some workgroups only tails
    // and others none (chunks of 4096 slots: 0.71 instead of 0.29 ms)
    const uint32_t G = gridDim.x;
    for (uint32_t sweep = 0u, first = 0u; first < numSlots; sweep++, first += G * LV_BLOCK) {
        const uint32_t i = first + ((blockIdx.x + 5u * sweep) % G) * LV_BLOCK + threadIdx.x;
        if (i >= numSlots) continue;
        const uint32_t w0 = records[3 * size_t(i) + 0], w1 = records[3 * size_t(i) + 1], rank = records[3 * size_t(i) + 2];
        if (w1 == LV_PPLL_DEAD) continue;   // slot of a chunk tail
        const uint32_t px = w0 & 0xFFFFu, py = w0 >> 16;
        const uint32_t leaf = w1 & 0x03FFFFFFu, tt = w1 >> 26;
        const float aoTexel = (U.useAmbientOcclusion && !U.aoPrebaked) ? S.ao[size_t(py) * U.width + px] : 1.0f;
        f3 oo, d;
        lv_primary_ray(U, px, py, 0.5f, 0.5f, oo, d);
        const LvRasterQuad rq = lv_make_raster_quad(U, px, py);
        bool kept;
        float depth;
        const f4 color = lv_shade_prism<SHADE, FAST>(S, U, s_prismRing, aoTexel, o, d, tLo, tHi, leaf, tt, rq, U.ppllRasterColour != 0u, depth, kept);
        if (STATS && kept) hits++;
        const uint32_t addr = lv_ppll_addr(px, py, U.ppllPaddedW, U.ppllTileW, U.ppllTileH);
        const size_t dst = size_t(lv_run_start(pixelOffset, blockBase, addr)) + rank;
        if (kept && color.w >= 0.001f) {   // gatherFragment: discard below, LinkedListGather.glsl:34
            frags[dst] = make_uint2(lv_pack_unorm4x8(color), __float_as_uint(depth));
            localSum++;
        } else {
            frags[dst] = make_uint2(0u, LV_PPLL_DEAD);
            atomicAdd(&fragCount[addr], 0x10000u);
            localDead++;
        }
    }
#pragma unroll
    for (int ofs = 32; ofs > 0; ofs >>= 1) {
        localSum += (uint32_t)__shfl_xor(localSum, ofs, 64);
        localDead += (uint32_t)__shfl_xor(localDead, ofs, 64);
    }
    if (lv_lane() == 0 && localSum > 0u) atomicAdd(&dc->fragCounter, localSum); // fragCounter of the reference: every fragment counts
    if (lv_lane() == 0 && localDead > 0u) atomicAdd(&dc->prismDiscards, localDead);
    if (STATS) {
        hits = lv_wave_sum_u64(hits);
        if (lv_lane() == 0 && hits) atomicAdd(&dc->hits, hits);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The rasteriser front end of ppll_fragment_source = raster_prism (ppll_prism_rasteriser = "segments", default): like the hardware
// the reference draws with, it walks the PRIMITIVES, not the pixels.  One lane per line segment (leaf order = Morton order: the 64
// segments of a wave are neighbours on screen): the 2 N ring vertices are projected with the frame's view-projection matrix, their
// bounding rectangle -- widened by LV_PRISM_BBOX_MARGIN pixels against the rounding of that projection -- gives the candidate pixels,
// and each candidate (pixel, segment) pair is decided by EXACTLY the coverage test of the LBVH front end (lv_prism_coverage_pts on the
// pixel-centre viewing ray: ray-space edge functions, fill rule), so both front ends produce the same fragments bit for bit; the
// rectangle only has to be conservative.  Config 4: 1 M segments of ~1 x 3 pixels -> 9.4 M coverage tests, where the all-hits walk
// of the 2.07 M viewing rays visited 41.9 M nodes and tested 19.7 M candidates (13.7 M of them up to the coverage stage).
//   k_ppll_mark_tiles   startOffset[pixel] = 0 for the pixels of the requested tiles (k_ppll_clear left 0xFFFFFFFF everywhere):
//                       a rank of a sharded frame rasterises into its own tiles only
//   k_ppll_raster_prism coverage -> record {pixel, leaf | triangle << 26, rank}; rank = atomicAdd on the pixel's count.  The order of
//                       a pixel's fragments is therefore not defined -- as in the reference, whose fragment shader invocations race
//                       on atomicExchange(startOffset) -- and nothing downstream depends on it: the resolve pass orders by the
//                       (depth, colour) key and, where a pixel holds more fragments than the sort arrays, keeps the nearest ones.
#define LV_PRISM_COARSE 32u
#define LV_PRISM_BBOX_MARGIN 0.015625f   // 1/64 pixel: two orders of magnitude above the rounding of the projection at 4K
#ifndef LV_PRISM_RASTER_CHUNK
#define LV_PRISM_RASTER_CHUNK 512u      // record slots a wave reserves per global atomic
#endif
#ifndef LV_PRISM_RASTER_MIN_WAVES
#define LV_PRISM_RASTER_MIN_WAVES 3
#endif
#ifndef LV_PRISM_RASTER_BLOCKS_PER_CU
#define LV_PRISM_RASTER_BLOCKS_PER_CU 3
#endif
template <bool STATS>
__global__ __launch_bounds__(LV_BLOCK) void k_ppll_mark_tiles(const LvUniforms U, const LvTiles T, uint32_t* __restrict__ startOffset,
                                                              uint32_t* __restrict__ coarse, LvDevCounters* dc) {
    LvPixel px;
    if (!lv_block_pixel(U, T, px)) return;
    if (px.inView) {
        startOffset[lv_ppll_addr(px.x, px.y, U.ppllPaddedW, U.ppllTileW, U.ppllTileH)] = 0u;
        // coarse map (LV_PRISM_COARSE-pixel cells): which parts of the viewport hold requested pixels at all -- the segment rasteriser of a
        // rank of a sharded frame drops the segments that project elsewhere before their set-up
        coarse[(px.y / LV_PRISM_COARSE) * ((U.width + LV_PRISM_COARSE - 1u) / LV_PRISM_COARSE) + px.x / LV_PRISM_COARSE] = 1u;
    }
    if (STATS) {   // one viewing ray per requested pixel: the rays whose coverage the rasteriser decides
        const unsigned long long n = (unsigned long long)__popcll(__ballot(px.inView));
        if (lv_lane() == 0 && n) atomicAdd(&dc->rays, n);
    }
}

// Offsets of the pixels' runs: exclusive scan of the per-pixel record counts.  Workgroup b of k_ppll_scan scans its LV_SCAN_ITEMS counts
// (in address order) and writes the offsets relative to its own first pixel and its total; k_ppll_scan_bases (one workgroup) turns the
// totals into bases.  Run of pixel a = lv_run_start(...) = startOffset[a] + blockBase[a / LV_SCAN_ITEMS].  (rocPRIM's scan: three
// launches, 24 us of a 0.85 ms frame.)  Pixels with more records than the sort arrays hold go on the list for k_ppll_select_nearest
// -- records, not kept fragments: the fragment stage has not run yet; that kernel copes with the superset.
#define LV_SCAN_ROUNDS (LV_SCAN_ITEMS / (4u * LV_BLOCK))   // uint4s per thread
__device__ __forceinline__ uint32_t lv_block_exclusive_scan(uint32_t v, uint32_t* s_wave, uint32_t& total) {
    const unsigned lane = lv_lane(), w = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int ofs = 1; ofs < 64; ofs <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up(int(incl), ofs, 64);
        if (int(lane) >= ofs) incl += t;
    }
    __syncthreads();   // (s_wave may still be read from the previous call)
    if (lane == 63u) s_wave[w] = incl;
    __syncthreads();
    uint32_t before = 0u;
    total = 0u;
#pragma unroll
    for (uint32_t k = 0; k < LV_BLOCK / LV_WAVE; k++) {
        const uint32_t t = s_wave[k];
        if (k < w) before += t;
        total += t;
    }
    return before + incl - v;
}
__global__ __launch_bounds__(LV_BLOCK) void k_ppll_scan(const uint32_t* __restrict__ fragCount, uint32_t* __restrict__ startOffset,
                                                        uint32_t n, uint32_t* __restrict__ blockTotals, uint32_t maxFrags,
                                                        uint32_t* __restrict__ overflowList, LvDevCounters* dc) {
    __shared__ uint32_t s_wave[LV_BLOCK / LV_WAVE];
    // round r: thread t owns the four addresses of uint4 number r * LV_BLOCK + t of the workgroup's block (coalesced)
    uint4 v[LV_SCAN_ROUNDS];
#pragma unroll
    for (uint32_t r = 0; r < LV_SCAN_ROUNDS; r++) {
        const uint32_t first = blockIdx.x * LV_SCAN_ITEMS + (r * LV_BLOCK + threadIdx.x) * 4u;
        v[r] = first < n ? *reinterpret_cast<const uint4*>(fragCount + first) : make_uint4(0u, 0u, 0u, 0u);   // (n is a multiple of 4)
    }
    uint32_t carry = 0u;
#pragma unroll
    for (uint32_t r = 0; r < LV_SCAN_ROUNDS; r++) {
        const uint32_t first = blockIdx.x * LV_SCAN_ITEMS + (r * LV_BLOCK + threadIdx.x) * 4u;
        const uint32_t c0 = v[r].x & 0xFFFFu, c1 = v[r].y & 0xFFFFu, c2 = v[r].z & 0xFFFFu, c3 = v[r].w & 0xFFFFu;
        if (max(max(c0, c1), max(c2, c3)) > maxFrags) {
            if (c0 > maxFrags) overflowList[atomicAdd(&dc->ppllOverflowPixels, 1u)] = first;
            if (c1 > maxFrags) overflowList[atomicAdd(&dc->ppllOverflowPixels, 1u)] = first + 1u;
            if (c2 > maxFrags) overflowList[atomicAdd(&dc->ppllOverflowPixels, 1u)] = first + 2u;
            if (c3 > maxFrags) overflowList[atomicAdd(&dc->ppllOverflowPixels, 1u)] = first + 3u;
        }
        uint32_t total;
        const uint32_t before = carry + lv_block_exclusive_scan(((c0 + c1) + c2) + c3, s_wave, total);
        if (first < n) *reinterpret_cast<uint4*>(startOffset + first) = make_uint4(before, before + c0, before + c0 + c1, before + c0 + c1 + c2);
        carry += total;
    }
    if (threadIdx.x == 0u) blockTotals[blockIdx.x] = carry;
}
// second launch, one workgroup: the blocks' totals -> the blocks' bases.  (One launch with "the last workgroup to finish does this"
// was slower than rocPRIM's three: 2 028 / 507 workgroups x a returning agent-scope atomic on one counter = 69 / 29 us.)
__global__ __launch_bounds__(LV_BLOCK) void k_ppll_scan_bases(const uint32_t* __restrict__ blockTotals, uint32_t* __restrict__ blockBase,
                                                              uint32_t numBlocks) {
    __shared__ uint32_t s_wave[LV_BLOCK / LV_WAVE];
    uint32_t carry = 0u;
    for (uint32_t i0 = 0u; i0 < numBlocks; i0 += LV_BLOCK) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t t = i < numBlocks ? blockTotals[i] : 0u;
        uint32_t chunk;
        const uint32_t ex = lv_block_exclusive_scan(t, s_wave, chunk);
        if (i < numBlocks) blockBase[i] = carry + ex;
        carry += chunk;
    }
}

// Keep-the-nearest selection.  The order of a pixel's run is not defined (the rasteriser's lanes race for the ranks, as the reference's
// fragment shader invocations race for the list heads), so nothing downstream may depend on it: a pixel with more kept fragments than
// the sort arrays hold keeps the K = ppllMaxNumFrags NEAREST by the (depth, colour) key -- one of the subsets the reference's race
// can leave in the first K nodes of a list, and the one that loses the least.  One wave per listed pixel permutes the run in place so
// that those K come first (the resolve pass then reads the first K live entries of every run, in whatever order):
//   the K-th smallest depth by a radix select over its 31 bits (a ballot + population count per 64 entries and bit), ties at that
//   depth by the same select over the colour; then entries below the threshold key go to the front, the rest (dead entries included)
//   behind them.  Runs of up to 64 * LV_SELECT_REGS entries live in registers throughout; longer ones are re-read from memory per bit
//   and permuted through their slots of the (by then free) record pool.
// In the resolve pass itself this selection sat on the critical path of the few hundred waves over the dense core (a max-heap per
// lane: +0.23 ms on config 4 for 2 235 such pixels; the whole wave per pixel: +0.65 ms, every lane of those waves overflows); as a pass
// of its own the pixels are spread over the whole GPU.
#define LV_SELECT_REGS 8
template <bool REGS>
__device__ __forceinline__ void lv_select_nearest(uint2* __restrict__ run, uint2* __restrict__ temp, uint32_t n, uint32_t K,
                                                  uint32_t lane) {
    uint32_t kd[LV_SELECT_REGS], kc[LV_SELECT_REGS];
    const uint32_t rounds = (n + LV_WAVE - 1u) / LV_WAVE;
    if (REGS) {
#pragma unroll
        for (uint32_t r = 0; r < LV_SELECT_REGS; r++) {
            const uint32_t i = r * LV_WAVE + lane;
            uint2 e = make_uint2(0u, LV_PPLL_DEAD);
            if (r < rounds && i < n) e = run[i];
            kd[r] = e.y;   // live depths are positive floats (< 0x7F800000): ordered like their bits, below the dead pattern 0xFFFFFFFF
            kc[r] = e.x;
        }
    } else {
        for (uint32_t i = lane; i < n; i += LV_WAVE) temp[i] = run[i];   // (each lane reads back only what it wrote itself)
    }
    // number of entries with pred(depth bits, colour)
#define LV_SELECT_COUNT(cnt, PRED)                                                                             \
    do {                                                                                                       \
        cnt = 0u;                                                                                              \
        if (REGS) {                                                                                            \
            _Pragma("unroll") for (uint32_t r = 0; r < LV_SELECT_REGS; r++)                                    \
                if (r < rounds) { const uint32_t D = kd[r], C = kc[r]; (void)C; cnt += uint32_t(__popcll(__ballot(PRED))); } \
        } else {                                                                                               \
            for (uint32_t i0 = 0u; i0 < n; i0 += LV_WAVE) {                                                    \
                const uint32_t i = i0 + lane;                                                                  \
                const uint2 e = i < n ? temp[i] : make_uint2(0u, LV_PPLL_DEAD);                                \
                const uint32_t D = e.y, C = e.x; (void)C;                                                      \
                cnt += uint32_t(__popcll(__ballot(PRED)));                                                     \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)
    {   // (the list holds the pixels with more than K RECORDS: where the fragment stage discarded enough of them, all live entries are kept)
        uint32_t live;
        LV_SELECT_COUNT(live, D < 0x80000000u);
        K = min(K, live);
        if (K == 0u) return;
    }
    uint32_t V = 0u;   // K-th smallest depth
    for (int bit = 30; bit >= 0; --bit) {
        const uint32_t cand = V | (1u << bit);
        uint32_t cnt;
        LV_SELECT_COUNT(cnt, D < cand);
        if (cnt < K) V = cand;
    }
    uint32_t less;
    LV_SELECT_COUNT(less, D < V);
    const uint32_t need = K - less;   // >= 1 of the entries at depth V, smallest colours first
    uint32_t W = 0u;   // need-th smallest colour among them
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = W | (1u << bit);
        uint32_t cnt;
        LV_SELECT_COUNT(cnt, D == V && C < cand);
        if (cnt < need) W = cand;
    }
#undef LV_SELECT_COUNT
    uint32_t below;   // entries strictly below the threshold key (V, W); K - below of the entries equal to it are kept too
    {
        uint32_t c2;
        if (REGS) {
            c2 = 0u;
#pragma unroll
            for (uint32_t r = 0; r < LV_SELECT_REGS; r++)
                if (r < rounds) c2 += uint32_t(__popcll(__ballot(kd[r] == V && kc[r] < W)));
        } else {
            c2 = 0u;
            for (uint32_t i0 = 0u; i0 < n; i0 += LV_WAVE) {
                const uint32_t i = i0 + lane;
                const uint2 e = i < n ? temp[i] : make_uint2(0u, LV_PPLL_DEAD);
                c2 += uint32_t(__popcll(__ballot(e.y == V && e.x < W)));
            }
        }
        below = less + c2;
    }
    // destinations: [0, below) entries below the key, [below, K) the first K - below entries equal to it, [K, n) everything else
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t posA = 0u, posB = below, posC = K;
    auto place = [&](uint32_t D, uint32_t C, bool valid) {
        const bool a = valid && (D < V || (D == V && C < W));
        const bool b = valid && D == V && C == W;
        const unsigned long long ma = __ballot(a), mb = __ballot(b);
        const uint32_t rankB = posB + uint32_t(__popcll(mb & lt));
        const bool bKept = b && rankB < K;
        const bool c = valid && !a && !bKept;
        const unsigned long long mc = __ballot(c);
        uint32_t dst = 0u;
        if (a) dst = posA + uint32_t(__popcll(ma & lt));
        else if (bKept) dst = rankB;
        else if (c) dst = posC + uint32_t(__popcll(mc & lt));
        if (valid) run[dst] = make_uint2(C, D);
        posA += uint32_t(__popcll(ma));
        posB = min(K, posB + uint32_t(__popcll(mb)));
        posC += uint32_t(__popcll(mc));
    };
    if (REGS) {
#pragma unroll
        for (uint32_t r = 0; r < LV_SELECT_REGS; r++)
            if (r < rounds) place(kd[r], kc[r], r * LV_WAVE + lane < n);
    } else {
        for (uint32_t i0 = 0u; i0 < n; i0 += LV_WAVE) {
            const uint32_t i = i0 + lane;
            const uint2 e = i < n ? temp[i] : make_uint2(0u, LV_PPLL_DEAD);
            place(e.y, e.x, i < n);
        }
    }
}

__global__ __launch_bounds__(LV_WAVE) void k_ppll_select_nearest(const LvUniforms U, uint2* __restrict__ frags, uint2* __restrict__ temp,
                                                                 const uint32_t* __restrict__ startOffset,
                                                                 const uint32_t* __restrict__ blockBase,
                                                                 const uint32_t* __restrict__ fragCount,
                                                                 const uint32_t* __restrict__ overflowList, const LvDevCounters* dc) {
    const uint32_t count = dc->ppllOverflowPixels, lane = threadIdx.x;
    for (uint32_t p = blockIdx.x; p < count; p += gridDim.x) {
        const uint32_t addr = overflowList[p];
        const uint32_t off = lv_run_start(startOffset, blockBase, addr), n = fragCount[addr] & 0xFFFFu;
        if (n <= LV_WAVE * LV_SELECT_REGS) lv_select_nearest<true>(frags + off, nullptr, n, U.ppllMaxNumFrags, lane);
        else lv_select_nearest<false>(frags + off, temp + off, n, U.ppllMaxNumFrags, lane);
    }
}

// rows x, y, w of proj * view (clip = M * (p, 1)) as the segment rasteriser and its cull pass use them
__device__ __forceinline__ void lv_clip_rows(const LvUniforms& U, float mx[4], float my[4], float mw[4]) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
        mx[c] = 0.0f; my[c] = 0.0f; mw[c] = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; k++) {   // column-major 4 x 4
            mx[c] += U.proj[4 * k + 0] * U.view[4 * c + k];
            my[c] += U.proj[4 * k + 1] * U.view[4 * c + k];
            mw[c] += U.proj[4 * k + 3] * U.view[4 * c + k];
        }
    }
}
// Sharded frame (a rank renders a tile list): ONE pass per frame over the segments' 32-B records decides which of them can touch a
// requested pixel at all and appends those to a list; k_ppll_raster_prism then walks the list, so that the segments of the other ranks'
// tiles cost this rank 32 B and ~60 instructions each instead of a slot of the rasteriser's waves (frames, oriented box, an empty
// pixel walk) -- with every rank looping over all segments the rasteriser took 0.20 ms for an eighth of config 4's tiles against 0.29 ms
// for all of them (profiles/shard_probe_r04_c4.json).  The test is conservative: both line points projected, the prism lies within
// `radius` of the segment; near the camera plane: undecided = kept.  Order of the list: blocks of <= 64 segments in leaf (= Morton)
// order, the blocks in the order the waves' appends arrive -- the fragments do not depend on it (the ranks race anyway).
#define LV_CULL_PER_BLOCK 2048u   // segments per workgroup of k_ppll_cull_segments (one global append each)
__global__ __launch_bounds__(LV_BLOCK) void k_ppll_cull_segments(const LvUniforms U, const LvSceneDev S, const uint32_t* __restrict__ coarse,
                                                                 uint32_t* __restrict__ leafList, LvDevCounters* dc) {
    const LvPrismDev& R = S.prism;
    float mx[4], my[4], mw[4];
    lv_clip_rows(U, mx, my, mw);
    const float halfW = 0.5f * float(U.width), halfH = 0.5f * float(U.height);
    const float wEps = 1e-3f * U.nearDist;
    const unsigned lane = lv_lane();
    // workgroup b owns the LV_CULL_PER_BLOCK consecutive segments from b * LV_CULL_PER_BLOCK: survivors are collected in LDS and
    // appended with ONE global atomic per workgroup (one per wave and 64 segments: 15.6 K returning atomics on one address at ~13 ns
    // each = 0.12 ms on config 4, more than the pass saves)
    __shared__ uint32_t s_keep[LV_CULL_PER_BLOCK];
    __shared__ uint32_t s_n, s_base;
    if (threadIdx.x == 0u) s_n = 0u;
    __syncthreads();
    const uint32_t first = blockIdx.x * LV_CULL_PER_BLOCK, last = min(first + LV_CULL_PER_BLOCK, S.numSegs);
    for (uint32_t leafBase = first + (threadIdx.x & ~63u); leafBase < last; leafBase += LV_BLOCK) {
        const uint32_t leaf = leafBase + lane;
        bool mine = false;
        if (leaf < last) {
            const float4 pa = S.segs[2 * size_t(leaf)], pb = S.segs[2 * size_t(leaf) + 1];
            const f3 centre[2] = {mk3(pa.x, pa.y, pa.z), mk3(pb.x, pb.y, pb.z)};
            mine = true;
            // does the segment project into a coarse cell with requested pixels?  Both line points with a generous bound of the
            // projected radius
            float lo[2] = {3.0e38f, 3.0e38f}, hi[2] = {-3.0e38f, -3.0e38f};
            bool decided = true;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const f3 v = centre[e];
                const float cw = ((mw[0] * v.x + mw[1] * v.y) + mw[2] * v.z) + mw[3];
                if (!(cw > 4.0f * R.radius + wEps)) { decided = false; continue; }
                const float ccx = ((mx[0] * v.x + mx[1] * v.y) + mx[2] * v.z) + mx[3];
                const float ccy = ((my[0] * v.x + my[1] * v.y) + my[2] * v.z) + my[3];
                const float sx = (ccx / cw + 1.0f) * halfW, sy = (ccy / cw + 1.0f) * halfH;
                // |d screen| <= f r / (w - r) (1 + |x / w|) for a point within r of the centre: tangent of the off-axis angle from the
                // centre's own clip coordinates, 25 % and 2 pixels on top
                const float tanOff = fabsf(ccx / cw) / fabsf(U.proj[0]) + fabsf(ccy / cw) / fabsf(U.proj[5]);
                const float rp = 1.25f * (1.0f + tanOff) * R.radius * fmaxf(fabsf(U.proj[0]) * halfW, fabsf(U.proj[5]) * halfH) / (cw - R.radius) + 2.0f;
                lo[0] = fminf(lo[0], sx - rp); hi[0] = fmaxf(hi[0], sx + rp); lo[1] = fminf(lo[1], sy - rp); hi[1] = fmaxf(hi[1], sy + rp);
            }
            if (decided) {
                const float cwid = float((U.width + LV_PRISM_COARSE - 1u) / LV_PRISM_COARSE), chgt = float((U.height + LV_PRISM_COARSE - 1u) / LV_PRISM_COARSE);
                const float c0x = fmaxf(floorf(lo[0] / float(LV_PRISM_COARSE)), 0.0f), c1x = fminf(floorf(hi[0] / float(LV_PRISM_COARSE)), cwid - 1.0f);
                const float c0y = fmaxf(floorf(lo[1] / float(LV_PRISM_COARSE)), 0.0f), c1y = fminf(floorf(hi[1] / float(LV_PRISM_COARSE)), chgt - 1.0f);
                mine = false;
                if (c1x - c0x <= 3.0f && c1y - c0y <= 3.0f) {
                    for (float cy = c0y; cy <= c1y; cy += 1.0f)
                        for (float cx = c0x; cx <= c1x; cx += 1.0f) mine = mine || coarse[uint32_t(cy) * uint32_t(cwid) + uint32_t(cx)] != 0u;
                } else mine = c0x <= c1x && c0y <= c1y;   // (a large footprint: not worth the look-ups)
            }
        }
        const unsigned long long m = __ballot(mine);
        if (m) {
            uint32_t base = 0u;
            if (lane == 0u) base = atomicAdd(&s_n, uint32_t(__popcll(m)));
            base = __builtin_amdgcn_readfirstlane(base);
            if (mine) s_keep[base + uint32_t(__popcll(m & ((1ull << lane) - 1ull)))] = leaf;
        }
    }
    __syncthreads();
    const uint32_t n = s_n;
    if (threadIdx.x == 0u && n) s_base = atomicAdd(&dc->prismListCount, n);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += LV_BLOCK) leafList[s_base + i] = s_keep[i];
}

#define LV_PRISM_RASTER_QUEUE 128u   // (pixel, segment) pairs a wave holds between the two stages (>= 2 * LV_WAVE, power of two)
template <bool STATS, int NT>
__global__ __launch_bounds__(LV_BLOCK, LV_PRISM_RASTER_MIN_WAVES) void k_ppll_raster_prism(const LvUniforms U, const LvSceneDev S,
                                                                   uint32_t* __restrict__ records,
                                                                   const uint32_t* __restrict__ startOffset,
                                                                   uint32_t* __restrict__ fragCount, LvDevCounters* dc,
                                                                   uint32_t poolSlots, uint32_t allRequested,
                                                                   const uint32_t* __restrict__ leafList) {
    // Two stages per wave, joined by a queue in LDS:
    //  A  one lane per segment: the candidate pixels of its screen rectangle that also lie in the ORIENTED box of the projected ring
    //     vertices (axis = the projected segment; a 1 x 3-pixel segment at 45 degrees fills a third of its rectangle) and in a
    //     requested tile are queued as (pixel, segment) pairs -- a dozen instructions per candidate;
    //  B  whenever 64 pairs are waiting: one lane per pair, the coverage test of the pixel-centre viewing ray (~650 instructions,
    //     all lanes busy whatever the shapes of the segments) and the records of the covered triangles.
    // Both boxes only have to contain the prism's projection (widened by LV_PRISM_BBOX_MARGIN against rounding): the fragments are
    // decided by the coverage test alone.  With the test inline in stage A (the first version) a wave iterated as long as its largest
    // rectangle and two of three candidates missed: 0.39 ms on config 4; queued: see DESIGN.md.
    __shared__ uint32_t s_qLeaf[LV_BLOCK / LV_WAVE][LV_PRISM_RASTER_QUEUE], s_qPix[LV_BLOCK / LV_WAVE][LV_PRISM_RASTER_QUEUE];
    // the frames of the wave's 64 segments ({centre, normal, binormal} x 2 + the two point indices; [component][segment]): stage B
    // reads them from here -- re-fetching 96 B per pair through L2 was what its waves waited for
    __shared__ float s_seg[LV_BLOCK / LV_WAVE][20][LV_WAVE];
    const LvPrismDev& R = S.prism;
    const uint32_t N = NT > 0 ? uint32_t(NT) : R.n;
    const unsigned lane = lv_lane();
    uint32_t* qLeaf = s_qLeaf[threadIdx.x >> 6];
    uint32_t* qPix = s_qPix[threadIdx.x >> 6];
    float (*segLds)[LV_WAVE] = s_seg[threadIdx.x >> 6];
    const f3 o = mk3(U.camPos[0], U.camPos[1], U.camPos[2]);
    float mx[4], my[4], mw[4];   // rows of proj * view (clip = M * (p, 1)); x, y and w only
    lv_clip_rows(U, mx, my, mw);
    const float halfW = 0.5f * float(U.width), halfH = 0.5f * float(U.height);
    const float wEps = 1e-3f * U.nearDist;
    unsigned allocBase = 0u, allocLeft = 0u;   // this wave's chunk of record slots (wave-uniform)
    unsigned qHead = 0u, qCount = 0u;          // the queue (wave-uniform)
    unsigned long long tests = 0;
    uint32_t dropped = 0u;

    // stage B on the m <= 64 oldest pairs of the queue
    auto coverageStage = [&](unsigned m) {
        unsigned mask = 0u;
        uint32_t addr = 0u, leaf = 0u, pix = 0u;
        if (lane < m) {
            const unsigned q = (qHead + lane) & (LV_PRISM_RASTER_QUEUE - 1u);
            leaf = qLeaf[q];
            pix = qPix[q];
            const uint32_t px = pix & 0xFFFFu, py = pix >> 16;
            addr = lv_ppll_addr(px, py, U.ppllPaddedW, U.ppllTileW, U.ppllTileH);
            // (no requested-tile test here: a whole-viewport frame requests every pixel, a rank of a sharded frame filters in stage A)
            const f3 d = lv_prism_cov_dir(R, px, py);   // the pixel's coverage direction (unnormalised, lv_prism.h)
            const uint32_t sl = leaf >> 26;   // the pair's segment among the wave's 64 = the lane that queued it
            leaf &= 0x03FFFFFFu;
            LvPrismPoint pt[2];
            uint32_t pi[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                pt[e].centre = mk3(segLds[9 * e + 0][sl], segLds[9 * e + 1][sl], segLds[9 * e + 2][sl]);
                pt[e].normal = mk3(segLds[9 * e + 3][sl], segLds[9 * e + 4][sl], segLds[9 * e + 5][sl]);
                pt[e].binormal = mk3(segLds[9 * e + 6][sl], segLds[9 * e + 7][sl], segLds[9 * e + 8][sl]);
                pi[e] = __float_as_uint(segLds[18 + e][sl]);
            }
            mask = lv_prism_coverage_pts<NT>(R, pt, pi, R.radius, o, d);
            if (STATS) tests++;
        }
        while (__any(mask != 0u)) {
            const bool hit = mask != 0u;
            const unsigned long long hm = __ballot(hit);
            const unsigned n = unsigned(__popcll(hm));
            const unsigned rankInBatch = unsigned(__popcll(hm & ((1ull << lane) - 1ull)));
            unsigned base = allocBase, left = allocLeft, base2 = 0u;
            if (left < n) {
                if (lane == 0u) base2 = atomicAdd(&dc->fragAlloc, (unsigned)LV_PRISM_RASTER_CHUNK);
                base2 = __builtin_amdgcn_readfirstlane(base2);
                allocBase = base2 + (n - left);
                allocLeft = LV_PRISM_RASTER_CHUNK - (n - left);
            } else {
                allocBase = base + n;
                allocLeft = left - n;
            }
            if (hit) {
                const unsigned tt = unsigned(__ffs(int(mask))) - 1u;
                mask &= mask - 1u;
                const uint32_t insertIndex = rankInBatch < left ? base + rankInBatch : base2 + (rankInBatch - left);
                bool stored = false;
                if (insertIndex < poolSlots) {
                    const uint32_t rk = atomicAdd(&fragCount[addr], 1u);
                    // (the FULL word: the high half -- the discard count -- is still 0 in this stage, so once the count saturates every
                    // later lane fails and undoes its add; testing the low half only let lanes behind the one that hit 0xFFFF wrap
                    // around to ranks 0, 1, ... a second time)
                    if (rk < 0xFFFFu) {
                        records[3 * size_t(insertIndex) + 0] = pix;
                        records[3 * size_t(insertIndex) + 1] = leaf | (tt << 26);
                        records[3 * size_t(insertIndex) + 2] = rk;
                        stored = true;
                    } else {   // (the per-pixel count shares its word with the count of discarded fragments: 16 bits each)
                        atomicSub(&fragCount[addr], 1u);
                        records[3 * size_t(insertIndex) + 0] = 0u;
                        records[3 * size_t(insertIndex) + 1] = LV_PPLL_DEAD;
                    }
                }
                if (!stored) dropped++;   // no record: the fragment stage never sees it, but the reference's fragCounter counts it
            }
        }
        qHead = (qHead + m) & (LV_PRISM_RASTER_QUEUE - 1u);
        qCount -= m;
    };

    // (static interleave: wave w takes segments [64 (w + k numWaves), + 64).  Work queues were slower, whatever the order: tickets over
    // consecutive blocks of 256 / 1024 segments 0.42 / 0.80 ms (consecutive = spatially clustered: the blocks of the dense core cost
    // many times the average and end up as the tail), over blocks in a scrambled order (ticket * prime mod count) 0.38 / 0.33 / 0.37 /
    // 0.49 ms for 64 / 128 / 256 / 512 segments per ticket -- small tickets pay the returning atomic on one address (13 ns each, 15.6 K
    // of them), large ones the variance again -- against 0.29 ms)
    const uint32_t waveId = blockIdx.x * (LV_BLOCK / LV_WAVE) + (threadIdx.x >> 6), numWaves = gridDim.x * (LV_BLOCK / LV_WAVE);
    // the work list: every segment (leaf order), or -- a rank of a sharded frame -- the segments k_ppll_cull_segments kept
    const uint32_t numItems = leafList ? dc->prismListCount : S.numSegs;
    for (uint32_t itemBase = waveId * LV_WAVE; itemBase < numItems; itemBase += numWaves * LV_WAVE) {
        const bool valid = itemBase + lane < numItems;
        const uint32_t leaf = !valid ? 0u : (leafList ? leafList[itemBase + lane] : itemBase + lane);
        int x0 = 0, x1 = -1, y0 = 0, y1 = -1;
        // oriented box: |(pixel centre - c) . a - aMid| <= aHalf and |(pixel centre - c) . n - nMid| <= nHalf, n = (-a.y, a.x)
        float ax = 1.0f, ay = 0.0f, cx0 = 0.0f, cy0 = 0.0f, aMid = 0.0f, aHalf = 3.0e38f, nMid = 0.0f, nHalf = 3.0e38f;
        if (valid) {
            const float4 pa = S.segs[2 * size_t(leaf)], pb = S.segs[2 * size_t(leaf) + 1];
            uint32_t pi[2];
            LvPrismPoint pt[2];
            lv_prism_frames(S, leaf, pa, pb, pt, pi);
#pragma unroll
            for (int e = 0; e < 2; e++) {
                segLds[9 * e + 0][lane] = pt[e].centre.x; segLds[9 * e + 1][lane] = pt[e].centre.y; segLds[9 * e + 2][lane] = pt[e].centre.z;
                segLds[9 * e + 3][lane] = pt[e].normal.x; segLds[9 * e + 4][lane] = pt[e].normal.y; segLds[9 * e + 5][lane] = pt[e].normal.z;
                segLds[9 * e + 6][lane] = pt[e].binormal.x; segLds[9 * e + 7][lane] = pt[e].binormal.y; segLds[9 * e + 8][lane] = pt[e].binormal.z;
                segLds[18 + e][lane] = __uint_as_float(pi[e]);
            }
            // axis of the oriented box: the projected centres (any direction gives a valid box; this one gives a tight one)
            bool axisOk = true;
            float wcx[2], wcy[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const f3 v = pt[e].centre;
                const float cw = ((mw[0] * v.x + mw[1] * v.y) + mw[2] * v.z) + mw[3];
                const float ccx = ((mx[0] * v.x + mx[1] * v.y) + mx[2] * v.z) + mx[3];
                const float ccy = ((my[0] * v.x + my[1] * v.y) + my[2] * v.z) + my[3];
                axisOk = axisOk && cw > wEps;
                wcx[e] = (ccx / cw + 1.0f) * halfW;
                wcy[e] = (ccy / cw + 1.0f) * halfH;
            }
            if (axisOk) {
                const float dx = wcx[1] - wcx[0], dy = wcy[1] - wcy[0], l2 = dx * dx + dy * dy;
                if (l2 > 1e-12f && l2 < 1e30f) { const float il = 1.0f / sqrtf(l2); ax = dx * il; ay = dy * il; }
                cx0 = wcx[0]; cy0 = wcy[0];
            }
            float lox = 3.0e38f, hix = -3.0e38f, loy = 3.0e38f, hiy = -3.0e38f;
            float loa = 3.0e38f, hia = -3.0e38f, lon = 3.0e38f, hin = -3.0e38f;
            bool anyFront = false, anyBehind = false;
            for (uint32_t k = 0; k < N; k++) {
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const f3 v = lv_prism_pos(pt[e], lv_prism_dir(pt[e], R.cp[k], R.s[k]), R.radius);
                    const float cw = ((mw[0] * v.x + mw[1] * v.y) + mw[2] * v.z) + mw[3];
                    if (cw > wEps) {
                        const float ccx = ((mx[0] * v.x + mx[1] * v.y) + mx[2] * v.z) + mx[3];
                        const float ccy = ((my[0] * v.x + my[1] * v.y) + my[2] * v.z) + my[3];
                        const float wx = (ccx / cw + 1.0f) * halfW, wy = (ccy / cw + 1.0f) * halfH;
                        lox = fminf(lox, wx); hix = fmaxf(hix, wx); loy = fminf(loy, wy); hiy = fmaxf(hiy, wy);
                        const float ta = (wx - cx0) * ax + (wy - cy0) * ay, tn = (wy - cy0) * ax - (wx - cx0) * ay;
                        loa = fminf(loa, ta); hia = fmaxf(hia, ta); lon = fminf(lon, tn); hin = fmaxf(hin, tn);
                        anyFront = true;
                    } else anyBehind = true;
                }
            }
            if (anyFront) {   // (all vertices behind the camera plane: no viewing ray meets the prism at a positive depth)
                if (anyBehind) { lox = 0.0f; loy = 0.0f; hix = float(U.width); hiy = float(U.height); }   // straddles the camera plane
                else if (axisOk) {
                    // (the projected coordinates carry a relative rounding error of a few ulp: the margin also covers the box's own arithmetic)
                    aMid = 0.5f * (loa + hia); aHalf = 0.5f * (hia - loa) + LV_PRISM_BBOX_MARGIN;
                    nMid = 0.5f * (lon + hin); nHalf = 0.5f * (hin - lon) + LV_PRISM_BBOX_MARGIN;
                }
                // pixel x is a candidate iff its centre x + 0.5 lies in [lo - margin, hi + margin]
                const float fx0 = fmaxf(ceilf(lox - LV_PRISM_BBOX_MARGIN - 0.5f), 0.0f);
                const float fy0 = fmaxf(ceilf(loy - LV_PRISM_BBOX_MARGIN - 0.5f), 0.0f);
                const float fx1 = fminf(floorf(hix + LV_PRISM_BBOX_MARGIN - 0.5f), float(U.width) - 1.0f);
                const float fy1 = fminf(floorf(hiy + LV_PRISM_BBOX_MARGIN - 0.5f), float(U.height) - 1.0f);
                if (fx0 <= fx1 && fy0 <= fy1) { x0 = int(fx0); x1 = int(fx1); y0 = int(fy0); y1 = int(fy1); }
            }
        }
        int px = x0, py = y0;
        bool more = valid && x1 >= x0 && y1 >= y0;
        while (__any(more)) {
            bool cand = false;
            if (more) {
                const float qx = (float(px) + 0.5f) - cx0, qy = (float(py) + 0.5f) - cy0;
                cand = fabsf((qx * ax + qy * ay) - aMid) <= aHalf && fabsf((qy * ax - qx * ay) - nMid) <= nHalf;
                if (cand && !allRequested)
                    cand = startOffset[lv_ppll_addr(uint32_t(px), uint32_t(py), U.ppllPaddedW, U.ppllTileW, U.ppllTileH)] == 0u;
            }
            const unsigned long long cm = __ballot(cand);
            if (cand) {
                const unsigned q = (qHead + qCount + unsigned(__popcll(cm & ((1ull << lane) - 1ull)))) & (LV_PRISM_RASTER_QUEUE - 1u);
                qLeaf[q] = leaf | (lane << 26);
                qPix[q] = uint32_t(px) | (uint32_t(py) << 16);
            }
            qCount += unsigned(__popcll(cm));
            if (qCount >= LV_WAVE) coverageStage(LV_WAVE);
            if (more) {
                if (++px > x1) { px = x0; more = ++py <= y1; }
            }
        }
        while (qCount > 0u) coverageStage(qCount < LV_WAVE ? qCount : LV_WAVE);   // (the next 64 segments replace the frames in LDS)
    }
    // the slots this wave reserved but did not use are part of the range the fragment stage walks: mark them dead
    for (unsigned k = lane; k < allocLeft; k += LV_WAVE)
        if (allocBase + k < poolSlots) { records[3 * size_t(allocBase + k) + 0] = 0u; records[3 * size_t(allocBase + k) + 1] = LV_PPLL_DEAD; }
#pragma unroll
    for (int ofs = 32; ofs > 0; ofs >>= 1) dropped += (uint32_t)__shfl_xor(dropped, ofs, 64);
    if (lane == 0u && dropped > 0u) atomicAdd(&dc->fragCounter, dropped);
    if (STATS) {
        tests = lv_wave_sum_u64(tests);
        if (lane == 0u && tests) atomicAdd(&dc->prims, tests);
    }
}

// clear(): LinkedListClear.glsl:46-55 (start offsets = -1) + fragmentCounterBuffer->fill(0), and the per-pixel fragment counts
// (startValue: 0xFFFFFFFF = empty list / pixel not requested; the segment rasteriser of a frame whose tiles cover the whole viewport
// passes 0 = every pixel requested and skips k_ppll_mark_tiles; viewingRays: those pixels' viewing rays for the statistics)
__global__ __launch_bounds__(LV_BLOCK) void k_ppll_clear(uint4* __restrict__ startOffset, uint4* __restrict__ fragCount, size_t n4,
                                                         LvDevCounters* dc, uint32_t startValue, unsigned long long viewingRays) {
    const size_t i = size_t(blockIdx.x) * LV_BLOCK + threadIdx.x;
    if (i < n4) {
        startOffset[i] = make_uint4(startValue, startValue, startValue, startValue);
        fragCount[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (i == 0) {
        dc->fragCounter = 0u; dc->fragAlloc = 0u; dc->prismDiscards = 0u; dc->ppllOverflowPixels = 0u; dc->prismListCount = 0u;
        if (viewingRays) atomicAdd(&dc->rays, viewingRays);
    }
}

// per-thread fragment arrays interleaved over the wave: entry i of lane l at [i * 64 + l]
struct LvFragArrays {
    uint32_t* col;
    float* dep;
    __device__ __forceinline__ uint32_t& c(uint32_t i) { return col[i * LV_WAVE]; }
    __device__ __forceinline__ float& d(uint32_t i) { return dep[i * LV_WAVE]; }
    // (depth, colour) key order: the reference compares depth only and leaves ties to rasterisation order
    __device__ __forceinline__ bool gt(uint32_t a, uint32_t b) {
        float da = d(a), db = d(b);
        return da > db || (da == db && c(a) > c(b));
    }
    __device__ __forceinline__ void swap(uint32_t a, uint32_t b) {
        uint32_t tc = c(a); c(a) = c(b); c(b) = tc;
        float td = d(a); d(a) = d(b); d(b) = td;
    }
};

// minHeapSink4, LinkedListSort.glsl:177-205
__device__ __forceinline__ void lv_min_heap_sink4(LvFragArrays& A, uint32_t x, uint32_t fragsCount) {
    uint32_t c, t;
    while ((t = 4 * x + 1) < fragsCount) {
        if (t + 1 < fragsCount && A.gt(t, t + 1)) c = t + 1; else c = t;
        if (t + 2 < fragsCount && A.gt(c, t + 2)) c = t + 2;
        if (t + 3 < fragsCount && A.gt(c, t + 3)) c = t + 3;
        if (!A.gt(x, c)) return;
        A.swap(x, c);
        x = c;
    }
}

// ---- sorting_mode != "Priority Queue" (SORTING_MODE_NAMES, src/Renderers/PPLL.hpp:32-50): the whole list is sorted, then every
// fragment is blended (blendFTB, LinkedListSort.glsl:45-59; the priority queue alone stops at alpha 0.99).  All comparisons
// go through the (depth, colour) key like the heap's, so the seven modes differ in cost and -- bitonic sort on a list whose length
// is no power of two, the quicksorts when their fixed stack overflows -- in what the shader's algorithm leaves unsorted.
struct LvSortStack { // LinkedListQuicksort.glsl:30-55: pushes beyond STACK_SIZE are dropped, an empty stack pops 0
    int mem[64];
    int size, counter;
    __device__ __forceinline__ void push(int v) { if (counter < size) mem[counter++] = v; }
    __device__ __forceinline__ int pop() { return counter > 0 ? mem[--counter] : 0; }
};
__device__ __forceinline__ bool lv_key_less(float da, uint32_t ca, float db, uint32_t cb) {
    return da < db || (da == db && ca < cb);
}
__device__ __forceinline__ void lv_gap_insertion_pass(LvFragArrays& A, uint32_t n, uint32_t gap) {
    for (uint32_t i = gap; i < n; ++i) { // insertionSort (:80-104) is the pass with gap 1, shellSort (:107-137) four of them
        const uint32_t fc = A.c(i);
        const float fd = A.d(i);
        uint32_t j = i;
        while (j >= gap && lv_key_less(fd, fc, A.d(j - gap), A.c(j - gap))) {
            A.c(j) = A.c(j - gap);
            A.d(j) = A.d(j - gap);
            j -= gap;
        }
        A.c(j) = fc;
        A.d(j) = fd;
    }
}
__device__ __forceinline__ void lv_max_heap_sink(LvFragArrays& A, uint32_t x, uint32_t n) { // :140-157
    uint32_t c;
    while ((c = 2 * x + 1) < n) {
        if (c + 1 < n && A.gt(c + 1, c)) ++c;
        if (!A.gt(c, x)) return;
        A.swap(x, c);
        x = c;
    }
}
__device__ __forceinline__ int lv_sort_stack_size(uint32_t maxFrags) { // PerPixelLinkedListLineRenderer.cpp:178
    const int s = int(ceil(log2(double(maxFrags))) * 2 + 4);
    return s < 0 ? 0 : (s > 64 ? 64 : s);
}
__device__ void lv_sort_fragments(uint32_t mode, LvFragArrays& A, uint32_t n, uint32_t maxFrags) {
    if (mode == 1u) { // bubbleSort :62-77
        bool changed;
        do {
            changed = false;
            for (uint32_t i = 0; i + 1 < n; ++i)
                if (A.gt(i, i + 1)) { A.swap(i, i + 1); changed = true; }
        } while (changed);
    } else if (mode == 2u) {
        lv_gap_insertion_pass(A, n, 1u);
    } else if (mode == 3u) {
        lv_gap_insertion_pass(A, n, 24u);
        lv_gap_insertion_pass(A, n, 9u);
        lv_gap_insertion_pass(A, n, 4u);
        lv_gap_insertion_pass(A, n, 1u);
    } else if (mode == 4u) { // heapSort :159-172
        for (uint32_t i = (n + 1) / 2; i > 0; --i) lv_max_heap_sink(A, i - 1, n);
        for (uint32_t i = 1; i < n; ++i) {
            A.swap(0, n - i);
            lv_max_heap_sink(A, 0, n - i);
        }
    } else if (mode == 5u) { // bitonicSort :241-262, with the reference's guards (no padding to a power of two)
        for (uint32_t k = 2; k <= n; k *= 2)
            for (uint32_t j = k / 2; j > 0; j /= 2)
                for (uint32_t i = 0; i < n; i++) {
                    const uint32_t l = i ^ j;
                    if (l > i && l < n) {
                        const bool up = (i & k) == 0u;
                        if (up ? A.gt(i, l) : A.gt(l, i)) A.swap(i, l);
                    }
                }
    } else {
        LvSortStack st;
        st.size = lv_sort_stack_size(maxFrags);
        st.counter = 0;
        st.push(0);
        st.push(int(n) - 1);
        if (mode == 6u) { // quicksort :93-114 with the Lomuto partition :57-68
            while (st.counter != 0) {
                const int high = st.pop(), low = st.pop();
                const float pd = A.d(uint32_t(high));
                const uint32_t pc = A.c(uint32_t(high));
                int i = low;
                for (int j = low; j <= high; j++)
                    if (lv_key_less(A.d(uint32_t(j)), A.c(uint32_t(j)), pd, pc)) { A.swap(uint32_t(i), uint32_t(j)); i++; }
                A.swap(uint32_t(i), uint32_t(high));
                const int pivot = i;
                if (low < pivot - 1) { st.push(low); st.push(pivot - 1); }
                if (pivot + 1 < high) { st.push(pivot + 1); st.push(high); }
            }
        } else { // quicksortHybrid :116-141 with the Hoare partition :70-91, finished by an insertion sort
            if (n > 16u)
                while (st.counter != 0) {
                    const int high = st.pop(), low = st.pop();
                    const uint32_t a = uint32_t(low), m = uint32_t((low + high) / 2), b = uint32_t(high);
                    const bool ab = A.gt(m, a), ba2 = A.gt(a, b), bm = A.gt(m, b); // a < m, b < a, b < m
                    const uint32_t mnmb = A.gt(m, b) ? b : m, mnab = A.gt(a, b) ? b : a;
                    const uint32_t p = ab ? (ba2 ? a : mnmb) : (bm ? m : mnab);
                    const float pd = A.d(p);
                    const uint32_t pc = A.c(p);
                    int i = low - 1, j = high + 1, pivot;
                    for (;;) {
                        do { i = i + 1; } while (lv_key_less(A.d(uint32_t(i)), A.c(uint32_t(i)), pd, pc));
                        do { j = j - 1; } while (lv_key_less(pd, pc, A.d(uint32_t(j)), A.c(uint32_t(j))));
                        if (i >= j) { pivot = j; break; }
                        A.swap(uint32_t(i), uint32_t(j));
                    }
                    if (low + 16 < pivot) { st.push(low); st.push(pivot - 1); }
                    if (pivot + 16 < high) { st.push(pivot + 1); st.push(high); }
                }
            lv_gap_insertion_pass(A, n, 1u);
        }
    }
}

// One wave per workgroup; fragment arrays in LDS when they fit, else in a global scratch slab.
// ARRAYS (frames of ppll_fragment_source = raster_prism): the pixel's fragments are the contiguous run frags[startOffset[pixel] ...
// + count) of 8-B {colour, depth} entries (startOffset = exclusive scan of the counts), in no defined order.
template <bool USE_LDS, bool PQ = true, bool ARRAYS = false>
__global__ __launch_bounds__(LV_WAVE) void k_ppll_resolve(const LvUniforms U, const LvTiles T,
                                                          const uint32_t* __restrict__ nodes,
                                                          const uint32_t* __restrict__ startOffset,
                                                          uint32_t* __restrict__ out, uint32_t* __restrict__ scratch,
                                                          uint32_t numGroups, const uint32_t* __restrict__ fragCount,
                                                          LvDevCounters* dc, const uint32_t* __restrict__ blockBase) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    uint32_t maxCount = 0u;   // raster_prism frames: the largest per-pixel fragment count (kept fragments) is collected here
    const uint32_t maxFrags = U.ppllMaxNumFrags;
    const uint32_t lane = threadIdx.x;
    LvFragArrays A;
    if (USE_LDS) {
        A.col = reinterpret_cast<uint32_t*>(s_dyn) + lane;
        A.dep = reinterpret_cast<float*>(s_dyn) + size_t(maxFrags) * LV_WAVE + lane;
    } else {
        uint32_t* base = scratch + size_t(blockIdx.x) * 2 * maxFrags * LV_WAVE;
        A.col = base + lane;
        A.dep = reinterpret_cast<float*>(base) + size_t(maxFrags) * LV_WAVE + lane;
    }
    // A wave resolves one 8 x 8 pixel cell; the 64 cells of a 64 x 64-pixel group are consecutive, and the groups come in the dispatch
    // order of the tile kernels (LvTiles::groupOrder: most expensive group of the previous frame first), so that the long waves over
    // the dense core of a data set start first.
    const uint32_t groupsX = T.blocksX / 4u, groupsPerTile = groupsX * (T.blocksY / 4u);
    for (uint32_t g = blockIdx.x; g < numGroups; g += gridDim.x) {
        const uint32_t slot = g >> 6, cell = g & 63u;
        const uint32_t grp = T.groupOrder ? T.groupOrder[slot] : slot;
        const uint32_t tile = grp / groupsPerTile, gi = grp % groupsPerTile;
        const uint32_t lx = (gi % groupsX) * 64u + (cell & 7u) * 8u + (lane & 7u);
        const uint32_t ly = (gi / groupsX) * 64u + (cell >> 3) * 8u + (lane >> 3);
        const bool inTile = lx < T.tileW && ly < T.tileH;
        if (!__any(inTile)) continue;
        const uint32_t x = T.tilesXY[2 * tile] + lx, y = T.tilesXY[2 * tile + 1] + ly;
        const uint32_t outIndex = (tile * T.tileH + ly) * T.tileW + lx;
        const bool inView = inTile && x < U.width && y < U.height;
        float res[4] = {U.background[0], U.background[1], U.background[2], U.background[3]};
        uint32_t fragOffset = 0xFFFFFFFFu;
        uint32_t numFrags = 0;
        if (ARRAYS) {
            // the first maxFrags live entries of the run: where a pixel kept more, k_ppll_select_nearest moved the nearest to the front
            if (inView) {
                const uint32_t addr = lv_ppll_addr(x, y, U.ppllPaddedW, U.ppllTileW, U.ppllTileH);
                fragOffset = lv_run_start(startOffset, blockBase, addr);
                const uint32_t v = fragCount[addr];
                const uint32_t n = v & 0xFFFFu;                // entries of the run; v >> 16 of them are dead
                maxCount = max(maxCount, n - (v >> 16));
                const uint2* __restrict__ run = reinterpret_cast<const uint2*>(nodes) + fragOffset;
                for (uint32_t j = 0u; j < n && numFrags < maxFrags; j += 4u) {   // four independent loads in flight per step
                    uint2 e[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4u; k++) e[k] = j + k < n ? run[j + k] : make_uint2(0u, LV_PPLL_DEAD);
#pragma unroll
                    for (uint32_t k = 0; k < 4u; k++) {
                        if ((e[k].y == LV_PPLL_DEAD && e[k].x == 0u) || numFrags >= maxFrags) continue;
                        A.c(numFrags) = e[k].x;
                        A.d(numFrags) = __uint_as_float(e[k].y);
                        numFrags++;
                    }
                }
            }
            // the whole-list sorts see the fragments in ascending key order (what they leave unsorted depends on their input)
            if (!PQ && numFrags > 1u) {
                for (uint32_t i = numFrags / 2u; i > 0u; --i) lv_max_heap_sink(A, i - 1u, numFrags);
                for (uint32_t i = numFrags - 1u; i > 0u; --i) { A.swap(0u, i); lv_max_heap_sink(A, 0u, i); }
            }
        } else if (inView) {
            fragOffset = startOffset[lv_ppll_addr(x, y, U.ppllPaddedW, U.ppllTileW, U.ppllTileH)];
        }
        if (inView) {
            while (!ARRAYS && numFrags < maxFrags) {
                if (fragOffset == 0xFFFFFFFFu) break;
                const uint32_t c = nodes[3 * size_t(fragOffset) + 0], db = nodes[3 * size_t(fragOffset) + 1];
                fragOffset = nodes[3 * size_t(fragOffset) + 2];
                // dead node (LV_PPLL_DEAD): a linked slot whose fragment the fragment stage of the rasterised prism discarded; it is
                // not part of the list (gatherFragment never stored it, LinkedListGather.glsl:34).  Lists of the reference cannot
                // contain this pattern: a node's depth is a length.
                if (db == LV_PPLL_DEAD && c == 0u) continue;
                A.c(numFrags) = c;
                A.d(numFrags) = __uint_as_float(db);
                numFrags++;
            }
            if (!PQ && numFrags > 0) {
                lv_sort_fragments(U.ppllSortingMode, A, numFrags, maxFrags);
                float ray[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                for (uint32_t i = 0; i < numFrags; i++) { // blendFTB
                    f4 src = lv_unpack_unorm4x8(A.c(i));
                    ray[0] = ray[0] + ((1.0f - ray[3]) * src.w) * src.x;
                    ray[1] = ray[1] + ((1.0f - ray[3]) * src.w) * src.y;
                    ray[2] = ray[2] + ((1.0f - ray[3]) * src.w) * src.z;
                    ray[3] = ray[3] + (1.0f - ray[3]) * src.w;
                }
                const float a = ray[3];
                if (a > 0.0f) {
#pragma unroll
                    for (int k = 0; k < 3; k++) res[k] = (ray[k] / a) * a + U.background[k] * (1.0f - a);
                    res[3] = a + U.background[3] * (1.0f - a);
                }
            }
            if (PQ && numFrags > 0) {
                // frontToBackPQ, LinkedListSort.glsl:207-238
                uint32_t i;
                for (i = numFrags / 4; i > 0; --i) lv_min_heap_sink4(A, i, numFrags);
                float ray[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                i = 0;
                while (i < numFrags && ray[3] < 0.99f) {
                    lv_min_heap_sink4(A, 0, numFrags - i++);
                    f4 src = lv_unpack_unorm4x8(A.c(0));
                    ray[0] = ray[0] + ((1.0f - ray[3]) * src.w) * src.x;
                    ray[1] = ray[1] + ((1.0f - ray[3]) * src.w) * src.y;
                    ray[2] = ray[2] + ((1.0f - ray[3]) * src.w) * src.z;
                    ray[3] = ray[3] + (1.0f - ray[3]) * src.w;
                    A.c(0) = A.c(numFrags - i);
                    A.d(0) = A.d(numFrags - i);
                }
                const float a = ray[3];
                if (a > 0.0f) {
                    // straight alpha rgb/A, then BACK_TO_FRONT_STRAIGHT_ALPHA over the clear colour
                    // (LinkedListSort.glsl:236-237, PerPixelLinkedListLineRenderer.cpp:70,395-397)
#pragma unroll
                    for (int k = 0; k < 3; k++) res[k] = (ray[k] / a) * a + U.background[k] * (1.0f - a);
                    res[3] = a + U.background[3] * (1.0f - a);
                }
            }
        }
        f4 c; c.x = res[0]; c.y = res[1]; c.z = res[2]; c.w = res[3];
        if (inTile) out[outIndex] = lv_pack_unorm4x8(c);
    }
    if (ARRAYS) {
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) maxCount = max(maxCount, (uint32_t)__shfl_xor(maxCount, ofs, 64));
        if (lane == 0 && maxCount > 0u) atomicMax(&dc->maxDepthComplexity, maxCount);
    }
}

// ================================================================ depth range
__global__ __launch_bounds__(LV_BLOCK) void k_depth_minmax(const LvUniforms U, const lv_line_point* __restrict__ points,
                                                           uint32_t numPoints, LvDevCounters* dc) {
    // ComputeDepthValues.glsl:58-98; the 256-wide shared-memory tree + MinMaxReduce passes collapse into a wave
    // reduction and two atomics per wave (min/max are order independent, so the result is identical).
    const uint32_t i = blockIdx.x * LV_BLOCK + threadIdx.x;
    const float EPSILON = 1e-2f;
    float mn = U.farDist, mx = U.nearDist;
    if (i < numPoints) {
        const float* p = points[i].linePosition;
        f4 ssp = mulM4(U.view, p[0], p[1], p[2], 1.0f);
        f4 ndc = mulM4(U.proj, ssp.x, ssp.y, ssp.z, ssp.w);
        float nx = ndc.x / ndc.w, ny = ndc.y / ndc.w, nz = ndc.z / ndc.w;
        if (nx >= -1.0f && ny >= -1.0f && nz >= -1.0f && nx <= 1.0f && ny <= 1.0f && nz <= 1.0f) {
            float depth = clampf(-ssp.z, U.nearDist, U.farDist);
            mn = depth - EPSILON;
            mx = depth + EPSILON;
        }
    }
    mn = lv_wave_min(mn);
    mx = lv_wave_max(mx);
    if (lv_lane() == 0) {
        atomicMin(&dc->depthOrd[0], lv_f2ord(mn));
        atomicMax(&dc->depthOrd[1], lv_f2ord(mx));
    }
}
__global__ void k_depth_init(const LvUniforms U, LvDevCounters* dc) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        dc->depthOrd[0] = lv_f2ord(U.farDist);
        dc->depthOrd[1] = lv_f2ord(U.nearDist);
    }
}
__global__ void k_depth_finalize(const LvDevCounters* dc, float* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = lv_ord2f(dc->depthOrd[0]);
        out[1] = lv_ord2f(dc->depthOrd[1]);
    }
}

// ================================================================ arbitrary rays (parity inspection)
template <int PRIM>
__global__ __launch_bounds__(LV_BLOCK) void k_trace_rays(const LvSceneDev S, float radius, uint32_t capped,
                                                         const float* __restrict__ org, const float* __restrict__ dir,
                                                         float tMin, float tMax, uint32_t n, float* __restrict__ outT,
                                                         uint32_t* __restrict__ outSeg, uint32_t* __restrict__ outKind) {
    __shared__ unsigned s_stack[LV_STACK_LDS * LV_BLOCK];
    LV_COOP_SHARED(LV_BLOCK / LV_WAVE);
    LV_COOP_MEM(cm);
    const uint32_t i = blockIdx.x * LV_BLOCK + threadIdx.x;
    const bool valid = i < n;
    LvCounters cnt = {0, 0, 0, 0};
    const uint32_t j = valid ? i : 0u;
    f3 o = mk3(org[3 * j], org[3 * j + 1], org[3 * j + 2]);
    f3 d = mk3(dir[3 * j], dir[3 * j + 1], dir[3 * j + 2]);
    LvHit h = lv_trace_closest<false, false, PRIM>(S, radius, capped != 0, valid, o, d, tMin, tMax,
                                                   lv_stack_mem(s_stack, S.stackOverflow), cm, cnt);
    if (!valid) return;
    outT[i] = h.found ? h.t : tMax;
    outSeg[i] = h.found ? S.leafSeg[h.leaf] : 0xFFFFFFFFu;
    outKind[i] = h.found ? uint32_t(h.kind) : 0u;
}

__global__ __launch_bounds__(LV_BLOCK) void k_trace_rays_tri(const LvSceneDev S, const float* __restrict__ org,
                                                             const float* __restrict__ dir, float tMin, float tMax,
                                                             uint32_t n, float* __restrict__ outT,
                                                             uint32_t* __restrict__ outTri, float* __restrict__ outUV) {
    __shared__ unsigned s_stack[LV_STACK_LDS * LV_BLOCK];
    LV_COOP_SHARED(LV_BLOCK / LV_WAVE);
    LV_COOP_MEM(cm);
    const uint32_t i = blockIdx.x * LV_BLOCK + threadIdx.x;
    const bool valid = i < n;
    LvCounters cnt = {0, 0, 0, 0};
    const uint32_t j = valid ? i : 0u;
    f3 o = mk3(org[3 * j], org[3 * j + 1], org[3 * j + 2]);
    f3 d = mk3(dir[3 * j], dir[3 * j + 1], dir[3 * j + 2]);
    LvHit h = lv_trace_closest<false, false, LV_PRIM_TRIANGLE>(S, 0.0f, false, valid, o, d, tMin, tMax,
                                                               lv_stack_mem(s_stack, S.stackOverflow), cm, cnt);
    if (!valid) return;
    float t = tMax, u = 0.0f, v = 0.0f;
    if (h.found) {
        const uint32_t* ti = S.triIdx + 3 * size_t(h.leaf);
        const float* a = S.triVerts[ti[0]].vertexPosition;
        const float* b = S.triVerts[ti[1]].vertexPosition;
        const float* c = S.triVerts[ti[2]].vertexPosition;
        lv_ray_triangle(o, d, mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z), mk3(a[0], a[1], a[2]), mk3(b[0], b[1], b[2]),
                        mk3(c[0], c[1], c[2]), S.triPad, t, u, v);
    }
    outT[i] = h.found ? h.t : tMax;
    outTri[i] = h.found ? h.leaf : 0xFFFFFFFFu;
    outUV[2 * size_t(i)] = u;
    outUV[2 * size_t(i) + 1] = v;
}

inline uint32_t nblocks(uint64_t n, uint32_t bs = LV_BLOCK) { return uint32_t((n + bs - 1) / bs); }

void padTiling(uint32_t& w, uint32_t& h, uint32_t tw, uint32_t th) {
    // LineRenderer::getScreenSizeWithTiling, LineRenderer.cpp:805-812
    if (w % tw != 0) w = (w / tw + 1) * tw;
    if (h % th != 0) h = (h / th + 1) * th;
}

LvSceneDev sceneDev(const lv_ctx* ctx) {
    LvSceneDev S;
    S.nodes = (const float4*)ctx->nodes.ptr;
    S.segs = (const float4*)ctx->segs.ptr;
    S.segAxis = (const float4*)ctx->segAxis.ptr;
    S.prismFrames = (const float4*)ctx->prismFrames.ptr;
    S.leafSeg = (const uint32_t*)ctx->leafSeg.ptr;
    S.segToLeaf = (const uint32_t*)ctx->segToLeaf.ptr;
    S.points = (const lv_line_point*)ctx->points.ptr;
    S.numPoints = ctx->numPoints;
    S.segIdx = (const uint32_t*)ctx->segIdx.ptr;
    S.tf = (const float4*)ctx->tf.ptr;
    S.twistTex = (const float4*)ctx->twistTex.ptr;
    S.depthMinMax = (const float*)ctx->depthMinMax.ptr;
    S.ao = ctx->aoResult ? ctx->aoResult : (const float*)ctx->ao.ptr;
    S.stackOverflow = nullptr;
    S.accum = nullptr;
    S.numSegs = ctx->numSegs;
    S.literalIntersection = lv_literal_intersection(ctx) ? 1u : 0u;
    S.ellBandWidth = ctx->opt.bandWidth;
    S.ellMinBandThickness = ctx->opt.minBandThickness;
    {   // cameraPosition, as lv_fill_uniforms
        const float* m = ctx->invView;
        S.ellCamPos[0] = ((m[0] * 0.0f + m[4] * 0.0f) + m[8] * 0.0f) + m[12] * 1.0f;
        S.ellCamPos[1] = ((m[1] * 0.0f + m[5] * 0.0f) + m[9] * 0.0f) + m[13] * 1.0f;
        S.ellCamPos[2] = ((m[2] * 0.0f + m[6] * 0.0f) + m[10] * 0.0f) + m[14] * 1.0f;
    }
    S.tris = nullptr; S.triIdx = nullptr; S.triVerts = nullptr; S.triPoints = nullptr; S.triPad = 0.0f; S.triLeafSize = 1u; S.triPairs = 0u;
    S.bakedAo = (const float*)ctx->bakedAo.ptr;
    S.bakedBlendingWeights = (const float*)ctx->bakeBlendingWeights.ptr;
    return S;
}

// scene view for the triangle-tube kernels: `nodes` = triangle LBVH, numSegs = triangle count
LvSceneDev sceneDevTriangles(const lv_ctx* ctx) {
    LvSceneDev S = sceneDev(ctx);
    S.nodes = (const float4*)ctx->triNodes.ptr;
    S.numSegs = ctx->numTris;
    S.tris = (const float4*)ctx->tris.ptr;
    S.triIdx = (const uint32_t*)ctx->triIdx.ptr;
    S.triVerts = (const lv_tube_vertex*)ctx->triVerts.ptr;
    S.triPoints = (const lv_line_point*)ctx->triPoints.ptr;
    S.triPad = ctx->triPad;
    S.triLeafSize = ctx->triLeafSize;
    S.triPairs = ctx->triLeafPairs ? 1u : 0u;
    return S;
}

} // namespace

// host copy of lv_sincos2pi (lv_device.h): the same float32 operations in the same order (the library is built with -ffp-contract=off)
static void lv_sincos2pi_host(float xi, float& s, float& c) {
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
// ppll_fragment_source: "auto" = the rasterised prism wherever it is built (plain flow lines), the capsule probe for band data
bool lv_ppll_prism_source(const lv_ctx* ctx) {
    const LvOptions& o = ctx->opt;
    if (o.ppllFragmentSource == 1) return false;
    // auto: the prism wherever its fragment stage is built -- plain tubes, band data (USE_BANDS), rotating helicity bands, each with
    // the screen-space or the prebaked AO; not band data with helicity bands (the raster shaders' NUM_TUBE_SUBDIVISIONS >= 8 && USE_AMBIENT_OCCLUSION && USE_BANDS path)
    const bool built = !(o.helicityBands && o.useRibbons);
    return o.ppllFragmentSource == 2 ? true : built;
}
// per-frame constants of the rasterised prism (LvPrismDev)
// constants of the coverage direction (lv_prism.h "viewing ray"); float32, one fixed order -- the CPU checker of the test-suite states the same
static void lv_prism_cov_constants(const float* invView, const float* invProj, uint32_t width, uint32_t height, float C0[3], float Cx[3],
                                   float Cy[3]) {
    auto mul3 = [&](const float* v, float out[3]) {   // invView3 * v
        for (int k = 0; k < 3; k++) out[k] = (invView[k] * v[0] + invView[4 + k] * v[1]) + invView[8 + k] * v[2];
    };
    const float p23[3] = {invProj[8] + invProj[12], invProj[9] + invProj[13], invProj[10] + invProj[14]};
    float a[3], b[3], c[3];
    mul3(invProj, a);
    mul3(invProj + 4, b);
    mul3(p23, c);
    const float sx = 2.0f / float(width), sy = 2.0f / float(height);
    for (int k = 0; k < 3; k++) {
        Cx[k] = a[k] * sx;
        Cy[k] = b[k] * sy;
        C0[k] = (c[k] - a[k]) - b[k];
    }
}

static void lv_fill_prism(const lv_ctx* ctx, const LvUniforms& U, LvPrismDev& R) {
    uint32_t n = ctx->opt.tubeNumSubdivisions;
    if (n < 3u) n = 3u;
    if (n > LV_PRISM_MAX_SUBDIV) n = LV_PRISM_MAX_SUBDIV;
    R.n = n;
    for (uint32_t k = 0; k < LV_PRISM_MAX_SUBDIV; k++) { R.c[k] = 1.0f; R.s[k] = 0.0f; }
    for (uint32_t k = 0; k < n; k++) lv_sincos2pi_host(float(k) / float(n), R.s[k], R.c[k]);
    // band data: the elliptic ring of the rasterisers' USE_BANDS vertex stage
    R.bands = U.useBands ? 1u : 0u;
    R.radius = U.useBands ? U.bandWidth * 0.5f : U.radius;
    R.thickness = U.useBands ? U.minThickness : 1.0f;
    for (uint32_t k = 0; k < LV_PRISM_MAX_SUBDIV; k++) { R.cp[k] = R.thickness * R.c[k]; R.sn[k] = R.thickness * R.s[k]; }
    for (int k = 0; k < 3; k++) R.right[k] = U.invView[k];
    R.viewZ[0] = U.view[2]; R.viewZ[1] = U.view[6]; R.viewZ[2] = U.view[10]; R.viewZ[3] = U.view[14];
    R.nearDist = U.nearDist;
    R.farDist = U.farDist;
    lv_prism_cov_constants(U.invView, U.invProj, U.width, U.height, R.covC0, R.covCx, R.covCy);
}

void lv_fill_uniforms(const lv_ctx* ctx, LvUniforms& U) {
    memset(&U, 0, sizeof(U));
    memcpy(U.view, ctx->view, 64);
    memcpy(U.proj, ctx->proj, 64);
    memcpy(U.invView, ctx->invView, 64);
    memcpy(U.invProj, ctx->invProj, 64);
    // rayOrigin = (inverseViewMatrix * vec4(0,0,0,1)).xyz, TubeRayTracing.glsl:202 (also used as cameraPosition)
    const float* m = ctx->invView;
    U.camPos[0] = ((m[0] * 0.0f + m[4] * 0.0f) + m[8] * 0.0f) + m[12] * 1.0f;
    U.camPos[1] = ((m[1] * 0.0f + m[5] * 0.0f) + m[9] * 0.0f) + m[13] * 1.0f;
    U.camPos[2] = ((m[2] * 0.0f + m[6] * 0.0f) + m[10] * 0.0f) + m[14] * 1.0f;
    U.fovY = ctx->fovY;
    for (int k = 0; k < 4; k++) {
        U.background[k] = ctx->background[k];
        U.foreground[k] = 1.0f - ctx->background[k]; // LineData.cpp:1282-1283
    }
    const LvOptions& o = ctx->opt;
    U.lineWidth = o.lineWidth;
    U.radius = o.lineWidth * 0.5f; // TubeRayTracing.glsl:453
    U.lssGeometry = (o.rtLss && !o.rtTriangleMesh) ? 1u : 0u;
    U.useBands = o.useRibbons ? 1u : 0u;
    U.useEllipticTubes = (o.useRibbons && o.ellipticTubes) ? 1u : 0u;
    U.bandWidth = o.bandWidth;
    U.minBandThickness = o.minBandThickness;
    U.minThickness = o.thickBands ? o.minBandThickness : 1e-2f; // MIN_THICKNESS, LineDataFlow.cpp:2425-2430
    U.useHelicityBands = o.helicityBands ? 1u : 0u;
    U.numSubdivisionsBands = o.bandSubdivisions;
    U.separatorBaseWidth = o.separatorWidth;
    U.helicityRotationFactor = o.helicityRotationFactor;
    U.useTwistTexture = (o.helicityBands && o.useTwistLineTexture && ctx->twistLevels != 0u) ? 1u : 0u;   // LineDataFlow.cpp:2437
    U.twistFilterMode = o.twistFilterMode;
    U.twistW = ctx->twistW; U.twistH = ctx->twistH; U.twistLevels = ctx->twistLevels;
    U.uniformHelicityBandWidth = o.uniformTwistLineWidth ? 1u : 0u;
    U.ppllRasterColour = o.ppllRayTracerColour ? 0u : 1u;
    U.nearDist = ctx->nearDist;
    U.farDist = ctx->farDist;
    U.width = ctx->width;
    U.height = ctx->height;
    U.maxDepthComplexity = o.maxDepthComplexity;
    U.numSamplesPerFrame = o.numSamplesPerFrame;
    // multi-frame accumulation (num_accumulated_frames > 1): the caller renders frame after frame and passes frame_number
    U.frameNumber = o.numAccumulatedFrames > 1u ? o.frameNumber : 0u;
    U.useJitteredRays = (o.numAccumulatedFrames > 1u || o.numSamplesPerFrame > 1u) ? 1u : 0u; // VulkanRayTracer.cpp:420-426
    U.useDeterministicSampling = o.useDeterministicSampling;
    U.useCappedTubes = o.useCappedTubes;
    U.useHalos = o.useHalos;
    U.useDepthCues = o.depthCueStrength > 0.0f;
    U.useAmbientOcclusion = o.useAmbientOcclusion;
    U.depthCueStrength = o.depthCueStrength;
    U.aoStrength = o.aoStrength;
    U.aoGamma = o.aoGamma;
    U.attrMin = ctx->attrMin;
    U.attrMax = ctx->attrMax;
    U.tfN = ctx->tfN;
    U.aoSamplesPerFrame = o.aoSamplesPerFrame;
    U.aoUseDistance = o.aoUseDistance;
    U.aoJitterPrimary = o.aoJitterPrimary;
    U.aoFrameNumber = 0;
    U.aoGlobalFrameNumber = 0;
    U.aoRadius = o.aoRadius;
    // VulkanRayTracedAmbientOcclusion.cpp:588
    U.subdivisionCorrectionFactor = cosf(3.1415926535897932f / float(o.tubeNumSubdivisions));
    // PerPixelLinkedListLineRenderer.cpp:109-126,175,257
    const bool large = ctx->numSegs > 1000000u;
    U.ppllMaxNumFrags = o.ppllMaxNumFrags ? o.ppllMaxNumFrags : (large ? 380u : 100u);
    const uint32_t avg = o.ppllExpectedAvgDepthComplexity ? o.ppllExpectedAvgDepthComplexity : (large ? 120u : 20u);
    U.ppllTileW = o.ppllTileW;
    U.ppllSortingMode = o.ppllSortingMode;
    U.ppllTileH = o.ppllTileH;
    uint32_t pw = ctx->width, ph = ctx->height;
    padTiling(pw, ph, o.ppllTileW, o.ppllTileH);
    U.ppllPaddedW = pw;
    U.ppllPaddedH = ph;
    uint64_t pool = uint64_t(avg) * pw * ph;
    if (pool > 0xFFFFFFF0ull) pool = 0xFFFFFFF0ull; // node indices are 32 bit
    U.ppllLinkedListSize = uint32_t(pool);
    U.aoPrebaked = (o.useAmbientOcclusion && o.aoPrebaked) ? 1u : 0u;
    U.bakeNumLineVertices = ctx->bakeNumLineVertices;
    U.bakeNumParametrizationVertices = ctx->bakeNumParametrizationVertices;
    U.bakeNumTubeSubdivisions = o.bakeNumTubeSubdivisions;
}

int lv_frame_depth_range(lv_ctx* ctx) {
    LvUniforms U;
    lv_fill_uniforms(ctx, U);
    hipStream_t st = ctx->stream;
    int rc;
    if ((rc = lv_buf_reserve(ctx, ctx->depthMinMax, 16))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->counters, sizeof(LvDevCounters)))) return rc;
    LvDevCounters* dc = (LvDevCounters*)ctx->counters.ptr;
    k_depth_init<<<1, 64, 0, st>>>(U, dc);
    if (ctx->numPoints)
        k_depth_minmax<<<nblocks(ctx->numPoints), LV_BLOCK, 0, st>>>(U, (const lv_line_point*)ctx->points.ptr,
                                                                      ctx->numPoints, dc);
    k_depth_finalize<<<1, 64, 0, st>>>(dc, (float*)ctx->depthMinMax.ptr);
    LV_HIP(ctx, hipGetLastError());
    return LV_OK;
}

// bytes of the global part of the traversal stacks for one launch geometry (0: the LDS-staged part holds the whole stack)
static size_t lv_overflow_bytes(const lv_ctx* ctx, uint64_t gridBlocks, uint32_t ldsEntries, bool triangles) {
    const uint64_t maxEntries = 3ull * uint64_t(triangles ? ctx->triWideDepth : ctx->wideDepth) + 2;
    if (maxEntries <= ldsEntries) return 0;
    return size_t(gridBlocks) * LV_BLOCK * (maxEntries - ldsEntries) * 4;
}

// persistent grid of k_ao_rays: enough workgroups to fill every CU at the kernel's LDS-limited residency
static uint64_t lv_ao_grid(const lv_ctx* ctx, uint64_t maxRays) {
    uint64_t gridRays = uint64_t(ctx->numCUs) * LV_AO_BLOCKS_PER_CU;
    if (gridRays > (maxRays + LV_AO_BLOCK - 1) / LV_AO_BLOCK) gridRays = (maxRays + LV_AO_BLOCK - 1) / LV_AO_BLOCK;
    return gridRays ? gridRays : 1;
}

// ---------------------------------------------------------------- dispatch order of the tile kernels
// A tile kernel lasts as long as its last workgroup, and the cost of a 64x64-pixel group spans orders of magnitude (a group over
// the core of a bundle against one over the background).  In tile-list order the heavy groups start whenever their turn
// comes and the launch ends with a tail of a few of them; started first (longest processing time first) the light groups fill
// the gaps behind them.  The cost is what the group's waves took in the previous frame (LvTiles::groupCost) -- with a
// camera that moves a little per frame a good predictor, and only a predictor: every order renders the same image.
// One workgroup sorts (cost descending, group index ascending) with a bitonic network in LDS and clears the cost array.
#define LV_ORDER_MAX_GROUPS 8192u
__global__ __launch_bounds__(1024) void k_group_order(uint32_t* __restrict__ cost0, uint32_t* __restrict__ order0, uint32_t n0,
                                                      uint32_t* __restrict__ cost1, uint32_t* __restrict__ order1, uint32_t n1) {
    extern __shared__ unsigned long long s_keys[];
    uint32_t* cost = blockIdx.x ? cost1 : cost0;
    uint32_t* order = blockIdx.x ? order1 : order0;
    const uint32_t n = blockIdx.x ? n1 : n0;
    if (n == 0u) return;
    uint32_t np = 1u;
    while (np < n) np <<= 1;
    for (uint32_t i = threadIdx.x; i < np; i += blockDim.x) {
        s_keys[i] = i < n ? ((unsigned long long)(~cost[i]) << 32) | i : ~0ull;
        if (i < n) cost[i] = 0u;
    }
    __syncthreads();
    if (n <= blockDim.x) {
        // up to 1 024 groups (a 1920 x 1080 viewport has 510): rank sort -- the keys are distinct (the group index is their low word), so
        // a key's position is the number of smaller keys; every lane reads the same LDS word per step (a broadcast), one barrier in all.
        // The bitonic network below needs 45 barrier-separated steps for 512 keys: 11 us per frame, 6 % of a config-2 frame.
        if (threadIdx.x < n) {
            const unsigned long long mine = s_keys[threadIdx.x];
            uint32_t rank = 0u;
            for (uint32_t i = 0u; i < n; i++) rank += s_keys[i] < mine ? 1u : 0u;
            order[rank] = uint32_t(mine);
        }
        return;
    }
    for (uint32_t k = 2u; k <= np; k <<= 1)
        for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < np; i += blockDim.x) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const unsigned long long a = s_keys[i], b = s_keys[l];
                    if ((a > b) == ((i & k) == 0u)) { s_keys[i] = b; s_keys[l] = a; }
                }
            }
            __syncthreads();
        }
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) order[i] = uint32_t(s_keys[i]);
}

// Points T at the dispatch order / cost array of launch geometry `which` (0 colour pass, 1 RTAO pass); a new tile list or
// geometry starts from zero cost (= tile-list order).  The sort itself is queued by lv_group_order_sort.
static int lv_group_order_prepare(lv_ctx* ctx, LvTiles& T, int which) {
    T.groupOrder = nullptr;
    T.groupCost = nullptr;
    lv_ctx::GroupOrder& G = ctx->groupOrder[which];
    G.active = false;
    const uint64_t n64 = uint64_t(T.numTiles) * (T.blocksX / 4u) * (T.blocksY / 4u);
    if (!ctx->opt.dispatchByCost || n64 < 2u || n64 > LV_ORDER_MAX_GROUPS) { G.n = 0; return LV_OK; }
    const uint32_t n = uint32_t(n64);
    int rc;
    if ((rc = lv_buf_reserve(ctx, G.cost, size_t(n) * 4))) return rc;
    if ((rc = lv_buf_reserve(ctx, G.order, size_t(n) * 4))) return rc;
    if (G.n != n || G.tileW != T.tileW || G.tileH != T.tileH || G.generation != ctx->tilesGeneration) {
        LV_HIP(ctx, hipMemsetAsync(G.cost.ptr, 0, size_t(n) * 4, ctx->stream));
        G.n = n; G.tileW = T.tileW; G.tileH = T.tileH; G.generation = ctx->tilesGeneration;
    }
    T.groupOrder = (const uint32_t*)G.order.ptr;
    T.groupCost = (uint32_t*)G.cost.ptr;
    G.active = true;
    return LV_OK;
}

// Queued once per frame, in front of the frame's first tile kernel (every geometry of the frame is prepared by then).
static int lv_group_order_sort(lv_ctx* ctx) {
    if (ctx->groupOrderSorted) return LV_OK;
    ctx->groupOrderSorted = true;
    const lv_ctx::GroupOrder& A = ctx->groupOrder[0];
    const lv_ctx::GroupOrder& B = ctx->groupOrder[1];
    const uint32_t nA = A.active ? A.n : 0u, nB = B.active ? B.n : 0u;
    const uint32_t nMax = nA > nB ? nA : nB;
    if (nMax == 0u) return LV_OK;
    uint32_t np = 1u;
    while (np < nMax) np <<= 1;
    k_group_order<<<nB ? 2 : 1, 1024, size_t(np) * 8, ctx->stream>>>((uint32_t*)A.cost.ptr, (uint32_t*)A.order.ptr, nA,
                                                                    (uint32_t*)B.cost.ptr, (uint32_t*)B.order.ptr, nB);
    LV_HIP(ctx, hipGetLastError());
    return LV_OK;
}

// the global part of the traversal stacks: only when the tree is higher than the LDS-staged part.  lv_frame_render
// reserves the largest slab any kernel of the frame needs BEFORE it builds a scene view, so that the reserve below never
// reallocates under a pointer an earlier view still holds.
static int lv_prepare_overflow(lv_ctx* ctx, LvSceneDev& S, uint64_t gridBlocks, uint32_t ldsEntries = LV_STACK_LDS,
                               bool triangles = false) {
    S.stackOverflow = nullptr;
    // a step of the 4-wide tree pushes at most 3 references per level
    const uint64_t maxEntries = 3ull * uint64_t(triangles ? ctx->triWideDepth : ctx->wideDepth) + 2;
    if (maxEntries <= ldsEntries) return LV_OK;
    const uint64_t extra = maxEntries - ldsEntries;
    int rc = lv_buf_reserve(ctx, ctx->stackOverflow, size_t(gridBlocks) * LV_BLOCK * extra * 4);
    if (rc) return rc;
    S.stackOverflow = (unsigned*)ctx->stackOverflow.ptr;
    return LV_OK;
}

// pairColour: the colour pass' scene view when its first-hit trace rides along with the RTAO primaries of the first iteration
// (k_primary_pair; lv_frame_render decides) -- the hits land in ctx->firstHit, *paired says whether the launch happened
static int lv_run_ao(lv_ctx* ctx, LvUniforms& U, LvSceneDev& S, const LvTiles& Tcolour, uint32_t gridTilesColour,
                     uint64_t maxPixelsColour, const LvSceneDev* pairColour = nullptr, bool* paired = nullptr) {
    hipStream_t st = ctx->stream;
    LvDevCounters* dc = (LvDevCounters*)ctx->counters.ptr;
    const uint32_t spp = U.aoSamplesPerFrame;
    int rc;
    // With jittered colour rays the AO lookup blends the four texels around the projected hit (AmbientOcclusion.glsl:84-99), so
    // the AO image needs a 1-pixel halo around every rendered tile: the AO pass runs on the tiles dilated by one pixel
    // (origins - 1, modulo 2^32: pixels left of / above the viewport fail the inView test; size + 2).  Rings of adjacent tiles
    // overlap -- those pixels are simply computed twice, which is why the running mean below reads the previous pass' image
    // (ping-pong) instead of updating in place.
    LvTiles T = Tcolour;
    uint32_t gridTiles = gridTilesColour;
    uint64_t maxPixels = maxPixelsColour;
    // ... and the EAW denoiser (ambient_occlusion_denoiser) reads 2 * (2^iterations - 1) pixels around every pixel it filters
    // (a-trous passes with step widths 1, 2, 4, ...): the same mechanism with a wider halo.
    const bool eaw = ctx->opt.eawEnabled && ctx->opt.eawIterations > 0u;
    // ... and SVGF is temporal: its history is read at reprojected positions anywhere in the picture, so the RTAO pass and the
    // denoiser always cover the whole viewport (one tile at the origin, no halo needed), whatever tiles the call renders.
    const bool svgf = ctx->opt.svgfEnabled;
    const uint32_t haloPx = svgf ? 0u : (U.aoProjectLookup ? 1u : 0u) + (eaw ? 2u * ((1u << ctx->opt.eawIterations) - 1u) : 0u);
    const bool halo = haloPx != 0u;
    if (svgf) {
        if (!ctx->fullFrameTile.ptr) {
            if ((rc = lv_buf_reserve(ctx, ctx->fullFrameTile, 8))) return rc;
            LV_HIP(ctx, hipMemsetAsync(ctx->fullFrameTile.ptr, 0, 8, st));
        }
        T.tilesXY = (const uint32_t*)ctx->fullFrameTile.ptr;
        T.numTiles = 1;
        T.tileW = ctx->width;
        T.tileH = ctx->height;
        T.blocksX = ((T.tileW + 63u) / 64u) * 4u;
        T.blocksY = ((T.tileH + 63u) / 64u) * 4u;
        const uint64_t nb = uint64_t(T.blocksX) * T.blocksY;
        gridTiles = uint32_t((nb + 127u) / 128u) * 128u;
        maxPixels = uint64_t(T.tileW) * T.tileH;
        if ((rc = lv_svgf_prepare(ctx))) return rc;
    }
    if (halo) {
        const uint32_t n = Tcolour.numTiles;
        if ((rc = lv_buf_reserve(ctx, ctx->tilesHaloDev, size_t(n) * 8))) return rc;
        if (!ctx->tilesHaloUploaded || ctx->tilesHalo != haloPx) {
            if (ctx->tilesHaloUploaded) LV_HIP(ctx, hipStreamSynchronize(st)); // the staging copy may still be in flight
            ctx->tilesHaloHost.resize(2 * size_t(n));
            for (size_t i = 0; i < 2 * size_t(n); i++) ctx->tilesHaloHost[i] = ctx->tilesHost[i] - haloPx;
            LV_HIP(ctx, hipMemcpyAsync(ctx->tilesHaloDev.ptr, ctx->tilesHaloHost.data(), size_t(n) * 8, hipMemcpyHostToDevice, st));
            ctx->tilesHaloUploaded = true;
            ctx->tilesHalo = haloPx;
        }
        T.tilesXY = (const uint32_t*)ctx->tilesHaloDev.ptr;
        T.tileW = Tcolour.tileW + 2u * haloPx;
        T.tileH = Tcolour.tileH + 2u * haloPx;
        T.blocksX = ((T.tileW + 63u) / 64u) * 4u;
        T.blocksY = ((T.tileH + 63u) / 64u) * 4u;
        const uint64_t nb = uint64_t(n) * T.blocksX * T.blocksY;
        if (nb > 0x7FFFFFF0ull) return lv_fail(ctx, LV_E_INVALID, "tile list too large");
        gridTiles = uint32_t((nb + 127u) / 128u) * 128u;
        maxPixels = uint64_t(n) * T.tileW * T.tileH;
        if ((rc = lv_buf_reserve(ctx, ctx->aoAlt, size_t(ctx->width) * ctx->height * 4))) return rc;
    }
    const size_t numPix = size_t(ctx->width) * ctx->height;
    if (eaw) {
        for (LvDeviceBuffer* b : {&ctx->featNormal, &ctx->featNormalAlt, &ctx->featPosition, &ctx->featPositionAlt})
            if ((rc = lv_buf_reserve(ctx, *b, numPix * 16))) return rc;
        if ((rc = lv_buf_reserve(ctx, ctx->eawPing, numPix * 4))) return rc;
        if ((rc = lv_buf_reserve(ctx, ctx->eawPong, numPix * 4))) return rc;
    }
    // G-buffer: one segment of 4096 slots per 64x64-pixel group of the launch (k_ao_primary); samples: compact
    const uint64_t numGroups64 = uint64_t(T.numTiles) * (T.blocksX / 4u) * (T.blocksY / 4u);
    if (numGroups64 > 0x000FFFFFull) return lv_fail(ctx, LV_E_INVALID, "tile list too large");
    const uint32_t numGroups = uint32_t(numGroups64);
    // dispatch order: the colour pass' (same launch geometry: both passes add to one cost array) or, on dilated tiles / the
    // whole viewport, one of its own
    if ((svgf || halo) && (rc = lv_group_order_prepare(ctx, T, 1))) return rc;
    if ((rc = lv_group_order_sort(ctx))) return rc;
    const LvAoLayout tileCap = lv_ao_layout(T);   // segments sized by the pixels a group really holds
    if ((rc = lv_buf_reserve(ctx, ctx->aoGbuf, size_t(maxPixels) * 48))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->aoSamples, size_t(maxPixels) * spp * 4))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->aoList, (2 * size_t(numGroups) + 1) * 4))) return rc; // per-group counts, then bases
    uint32_t* tileCount = (uint32_t*)ctx->aoList.ptr;
    uint32_t* tileBase = tileCount + numGroups;
    ctx->aoNumGroups = svgf ? 0u : numGroups; // whole-viewport pass: no per-tile costs of the caller's tile list (lv_get_ao_tile_costs)
    ctx->aoGroupsPerTile = (T.blocksX / 4u) * (T.blocksY / 4u);
    const uint64_t gridRays = lv_ao_grid(ctx, maxPixels * spp);
    const uint64_t gridMax = gridRays > gridTiles ? gridRays : gridTiles;
    const bool tri = lv_ao_triangle_tubes(ctx);
    // RTAO geometry: the capsules of the colour pass, or the reference's triangle tubes (own LBVH, own scene view)
    LvSceneDev SA = tri ? sceneDevTriangles(ctx) : S;
    if ((rc = lv_prepare_overflow(ctx, SA, gridMax, LV_AO_STACK_LDS, tri))) return rc;
    // the paired launch: gridTiles + gridTilesColour workgroups share one slab (columns are indexed by the launch's global thread id);
    // lv_frame_render reserved it for that grid and for the deeper of the two trees
    LvSceneDev SCpair = pairColour ? *pairColour : S;
    if (pairColour) SCpair.stackOverflow = ctx->stackOverflow.ptr ? (unsigned*)ctx->stackOverflow.ptr : nullptr;
    LvSceneDev SApair = SA;
    if (pairColour) SApair.stackOverflow = SCpair.stackOverflow;
    const bool stats = ctx->opt.collectStats;
    // progressive mode (num_accumulated_frames > 1): one RTAO iteration per rendered frame while frame_number <
    // ambient_occlusion_iterations (ambientOcclusionBaker->updateIterative(), LineRenderer.cpp:257-264), accumulated in ctx->ao
    const bool progressive = ctx->opt.numAccumulatedFrames > 1u;
    if (ctx->aoRestart) {
        // the denoiser changed (lv_set_option): the accumulated AO image and the feature maps belong to another pipeline
        if (progressive && ctx->opt.frameNumber != 0u)
            return lv_fail(ctx, LV_E_STATE, "ambient_occlusion_denoiser changed: restart the accumulation with frame_number = 0");
        ctx->aoRestart = false;
    }
    const uint32_t iterBegin = progressive ? ctx->opt.frameNumber : 0u;
    const uint32_t iterEnd = progressive ? std::min(ctx->opt.frameNumber + 1u, ctx->opt.aoIterations) : ctx->opt.aoIterations;
    for (uint32_t iter = iterBegin; iter < iterEnd; iter++) {
        U.aoFrameNumber = iter; // rtaoRenderPass->setFrameNumber(accumulatedFramesCounter), VulkanRayTracedAmbientOcclusion.cpp:92
        U.aoGlobalFrameNumber = iter;
        LvSvgfFeat SF;
        SF.normalDepth = nullptr;
        SF.flowFwidth = nullptr;
        if (svgf) {
            // DISABLE_ACCUMULATION + useGlobalFrameNumber (SVGF.hpp:81-82; VulkanRayTracedAmbientOcclusion.cpp:415-421,576-581)
            U.aoFrameNumber = 0;
            U.aoGlobalFrameNumber = ctx->aoGlobalFrameNumber;
            SF.normalDepth = (float4*)ctx->svgf.normalDepth.ptr;
            SF.flowFwidth = (float4*)ctx->svgf.flowFwidth.ptr;
            float vp[16];
            lv_mat4_mul(ctx->proj, ctx->view, vp);
            memcpy(SF.lastFrameViewProj, ctx->lastFrameViewProjValid ? ctx->lastFrameViewProj : vp, sizeof vp);
            memcpy(ctx->lastFrameViewProj, vp, sizeof vp);
            ctx->lastFrameViewProjValid = true;
        }
        ctx->aoGlobalFrameNumber++;
        LV_HIP(ctx, hipMemsetAsync(tileCount, 0, size_t(numGroups) * 4, st));
        const uint32_t grid = uint32_t(gridRays);
        const float4* g = (const float4*)ctx->aoGbuf.ptr;
        // halo: read the previous pass' image, write the other buffer, swap; otherwise update in place
        const float* aoIn = (const float*)ctx->ao.ptr;
        float* ao = halo ? (float*)ctx->aoAlt.ptr : (float*)ctx->ao.ptr;
        float* smp = (float*)ctx->aoSamples.ptr;
        // feature maps of the denoiser: same ping-pong as the AO image (the EAW halo makes the tiles overlap)
        const float4* fnIn = eaw ? (const float4*)ctx->featNormal.ptr : nullptr;
        float4* fnOut = eaw ? (float4*)ctx->featNormalAlt.ptr : nullptr;
        const float4* fpIn = eaw ? (const float4*)ctx->featPosition.ptr : nullptr;
        float4* fpOut = eaw ? (float4*)ctx->featPositionAlt.ptr : nullptr;
        const bool pairNow = pairColour && iter == iterBegin;
        if (pairNow && paired) *paired = true;
#define LV_LAUNCH_PAIR(ST, PR, BA)                                                                            \
    LV_TIMED_LAUNCH(ctx, LV_KERNEL_AO_PRIMARY, (k_primary_pair<ST, PR, BA><<<gridTiles + gridTilesColour, LV_BLOCK, 0, st>>>( \
            U, SApair, T, gridTiles, aoIn, ao, (float4*)ctx->aoGbuf.ptr, tileCount, dc, fnIn, fnOut, fpIn, fpOut, SF, SCpair, Tcolour, \
            (uint2*)ctx->firstHit.ptr, size_t(maxPixelsColour))))
#define LV_LAUNCH_AOP(ST, PR)                                                                                 \
    do {                                                                                                      \
        if (pairNow && U.useHelicityBands) LV_LAUNCH_PAIR(ST, PR, LV_SHADE_HELICITY);                         \
        else if (pairNow && U.useBands) LV_LAUNCH_PAIR(ST, PR, LV_SHADE_BANDS);                               \
        else if (pairNow) LV_LAUNCH_PAIR(ST, PR, LV_SHADE_PLAIN);                                             \
        else                                                                                                  \
            LV_TIMED_LAUNCH(ctx, LV_KERNEL_AO_PRIMARY, (k_ao_primary<ST, PR><<<gridTiles, LV_BLOCK, 0, st>>>(  \
                    U, SA, T, aoIn, ao, (float4*)ctx->aoGbuf.ptr, tileCount, dc, fnIn, fnOut, fpIn, fpOut, SF))); \
    } while (0)
#define LV_LAUNCH_AO(ST, AH, PR) \
    LV_TIMED_LAUNCH(ctx, LV_KERNEL_AO_RAYS, (k_ao_rays<ST, AH, PR><<<grid, LV_AO_BLOCK, 0, st>>>(U, SA, g, smp, dc, tileBase, numGroups, tileCap)))
#define LV_LAUNCH_AO_LIT(ST, AH) \
    LV_TIMED_LAUNCH(ctx, LV_KERNEL_AO_RAYS, (k_ao_rays<ST, AH, LV_PRIM_CAPSULE, false, true><<<grid, LV_AO_BLOCK, 0, st>>>( \
            U, SA, g, smp, dc, tileBase, numGroups, tileCap)))
#define LV_LAUNCH_AO2(ST, AH)                                          \
    do {                                                               \
        if (tri) LV_LAUNCH_AO(ST, AH, LV_PRIM_TRIANGLE);               \
        else if (U.useEllipticTubes) LV_LAUNCH_AO(ST, AH, LV_PRIM_ELLIPTIC); \
        else if (lv_literal_intersection(ctx)) LV_LAUNCH_AO_LIT(ST, AH); \
        else LV_LAUNCH_AO(ST, AH, LV_PRIM_CAPSULE);                    \
    } while (0)
        const bool ell = U.useEllipticTubes != 0u;
        if (stats) { if (tri) LV_LAUNCH_AOP(true, LV_PRIM_TRIANGLE); else if (ell) LV_LAUNCH_AOP(true, LV_PRIM_ELLIPTIC); else LV_LAUNCH_AOP(true, LV_PRIM_CAPSULE); }
        else { if (tri) LV_LAUNCH_AOP(false, LV_PRIM_TRIANGLE); else if (ell) LV_LAUNCH_AOP(false, LV_PRIM_ELLIPTIC); else LV_LAUNCH_AOP(false, LV_PRIM_CAPSULE); }
        k_ao_tile_scan<<<1, LV_BLOCK, 0, st>>>(tileCount, numGroups, tileBase, dc);
        const bool anyHit = !U.aoUseDistance;
#ifdef LV_AO_OCTANT_PERM
        {   // experiment only: the permutation is built by a pre-pass outside the timed kernel
            static uint32_t* permBuf = nullptr;
            static size_t permCap = 0;
            const size_t need = size_t(maxPixels) * spp;
            if (need > permCap) { if (permBuf) (void)hipFree(permBuf); LV_HIP(ctx, hipMalloc(&permBuf, need * 4)); permCap = need; }
            LV_HIP(ctx, hipMemcpyToSymbolAsync(HIP_SYMBOL(g_aoPerm), &permBuf, sizeof(permBuf), 0, hipMemcpyHostToDevice, st));
            k_ao_octant_perm<<<4096, 256, 0, st>>>(U, g, permBuf, dc, tileBase, numGroups, tileCap);
        }
#endif
        if (stats) { if (anyHit) LV_LAUNCH_AO2(true, true); else LV_LAUNCH_AO2(true, false); }
        else { if (anyHit) LV_LAUNCH_AO2(false, true); else LV_LAUNCH_AO2(false, false); }
#undef LV_LAUNCH_AO2
#undef LV_LAUNCH_AO_LIT
#undef LV_LAUNCH_AO
#undef LV_LAUNCH_AOP
#undef LV_LAUNCH_PAIR
        if ((spp & 3u) == 0u) {
            const uint64_t rowBlocks = (maxPixels + LV_REDUCE_ROWS - 1) / LV_REDUCE_ROWS;
            const uint64_t cap = uint64_t(ctx->numCUs) * 16u;
            k_ao_reduce_rows<<<uint32_t(rowBlocks < cap ? rowBlocks : cap), LV_REDUCE_ROWS, 0, st>>>(U, g, smp, aoIn, ao, dc, tileBase,
                                                                                                      numGroups, tileCap);
        } else {
            k_ao_reduce<false><<<nblocks(maxPixels), LV_BLOCK, 0, st>>>(U, g, smp, aoIn, ao, dc, tileBase, numGroups, tileCap);
        }
        if (halo) std::swap(ctx->ao, ctx->aoAlt);
        if (eaw) { std::swap(ctx->featNormal, ctx->featNormalAlt); std::swap(ctx->featPosition, ctx->featPositionAlt); }
        // temporal denoiser: denoise() belongs to every _render (VulkanRayTracedAmbientOcclusion.cpp:633-651), its history advances
        // with every RTAO iteration
        if (svgf && (rc = lv_svgf_denoise(ctx, (const float*)ctx->ao.ptr))) return rc;
    }
    if (iterEnd > iterBegin || !ctx->aoResult) ctx->aoResult = svgf ? (const float*)ctx->svgf.result.ptr : (const float*)ctx->ao.ptr;
    if (eaw && iterEnd > iterBegin) {
        // denoiser->denoise() after the RTAO pass (VulkanRayTracedAmbientOcclusion.cpp:633-651): the accumulation keeps running
        // on the raw image, the colour pass samples the denoised one
        LvEawParams E;
        E.phiColor = ctx->opt.eawPhiColor * 1.0f;         // weight scales of the AO mode, Denoiser.cpp:59-61
        E.phiPosition = ctx->opt.eawPhiPosition * 0.0001f;
        E.phiNormal = ctx->opt.eawPhiNormal * 1.0f;
        E.useColor = ctx->opt.eawColorWeights; E.usePosition = ctx->opt.eawPositionWeights; E.useNormal = ctx->opt.eawNormalWeights;
        E.stepWidth = 1;
        const uint64_t threads = uint64_t(T.numTiles) * T.tileW * T.tileH;
        const float* src = (const float*)ctx->ao.ptr;
        float* bufs[2] = {(float*)ctx->eawPing.ptr, (float*)ctx->eawPong.ptr};
        for (uint32_t i = 0; i < ctx->opt.eawIterations; i++) {
            float* dst = bufs[i & 1u];
            if (ctx->opt.eawUseSharedMemory)
                k_eaw_pass<true><<<nblocks(threads), LV_BLOCK, 0, st>>>(U, T, E, src, dst, (const float4*)ctx->featNormal.ptr,
                                                                         (const float4*)ctx->featPosition.ptr);
            else
                k_eaw_pass<false><<<nblocks(threads), LV_BLOCK, 0, st>>>(U, T, E, src, dst, (const float4*)ctx->featNormal.ptr,
                                                                          (const float4*)ctx->featPosition.ptr);
            src = dst;
            E.stepWidth *= 2;
        }
        ctx->aoResult = src;
    }
    S.ao = ctx->aoResult;
    LV_HIP(ctx, hipGetLastError());
    return LV_OK;
}

int lv_frame_render(lv_ctx* ctx, int mode, const uint32_t* tilesXYHost, uint32_t numTiles, uint32_t tileW,
                    uint32_t tileH, void* outDevice) {
    if (mode != LV_RENDERING_MODE_VULKAN_RAY_TRACER && mode != LV_RENDERING_MODE_PER_PIXEL_LINKED_LIST)
        return lv_fail(ctx, LV_E_INVALID, "unsupported rendering mode %d (11 = ray tracer, 2 = PPLL)", mode);
    if (!ctx->cameraSet) return lv_fail(ctx, LV_E_STATE, "lv_set_camera has not been called");
    if (!ctx->tf.ptr || ctx->tfN == 0) return lv_fail(ctx, LV_E_STATE, "lv_set_transfer_function has not been called");
    if (!ctx->points.ptr && ctx->numSegs) return lv_fail(ctx, LV_E_STATE, "lv_set_lines has not been called");
    if (numTiles == 0 || tileW == 0 || tileH == 0) return lv_fail(ctx, LV_E_INVALID, "empty tile list");
    int rc;
    if (ctx->opt.useRibbons) {
        // band data: the closest-hit paths of the ray tracer (analytic geometry modes, or "Triangle Mesh" on the elliptic triangle
        // tubes the host layer tessellates for the data set); RTAO over the analytic tubelets / capsules or over those triangle
        // tubes (rtao_geometry = triangle_tubes); MLAT over the same geometries; the PPLL gather over the analytic tubelets /
        // capsules.  The static prebaker bakes band data on the elliptic cross-section against the elliptic triangle tubes the caller
        // passed (VulkanAmbientOcclusionBaker.glsl:200-257) and is looked up with the band's own angle (phiLine of the tubelets).
        if (mode == LV_RENDERING_MODE_PER_PIXEL_LINKED_LIST && (ctx->opt.rtTriangleMesh || ctx->opt.rtLss))
            return lv_fail(ctx, LV_E_INVALID, "use_ribbons: the PPLL gather of band data runs over the analytic tubelets / capsules "
                                              "(geometry_mode \"AABBs\")");
        if (ctx->opt.rtTriangleMesh && ctx->opt.ellipticTubes)
            return lv_fail(ctx, LV_E_INVALID, "Elliptic Tubes belong to the AABB geometry mode (VulkanRayTracer.cpp:198), not to Triangle Mesh");
    } else if (ctx->opt.ellipticTubes) {
        return lv_fail(ctx, LV_E_INVALID, "use_analytic_elliptic_tubes needs band data (use_ribbons)");
    }
    if (ctx->opt.helicityBands && ctx->opt.useRibbons)
        return lv_fail(ctx, LV_E_INVALID, "rotating_helicity_bands and use_ribbons exclude each other (LineDataFlow.cpp:470,601-604)");
    if (ctx->opt.rtLss && ctx->opt.useRibbons && ctx->opt.ellipticTubes)
        return lv_fail(ctx, LV_E_INVALID, "Elliptic Tubes belong to the AABB geometry mode (VulkanRayTracer.cpp:198), not to Linear Swept Spheres");
    if (!ctx->accelValid || ctx->accelLineWidth != lv_accel_width(ctx))
        if ((rc = lv_bvh_build(ctx))) return rc;
    const bool needTriangles = (ctx->opt.useAmbientOcclusion && ctx->opt.aoPrebaked) ||
                               (ctx->opt.useAmbientOcclusion && lv_ao_triangle_tubes(ctx)) ||
                               (ctx->opt.rtTriangleMesh && mode == LV_RENDERING_MODE_VULKAN_RAY_TRACER);
    if (needTriangles) {
        if ((rc = lv_ensure_tube_mesh(ctx))) return rc;
        if (!ctx->triMeshSet)
            return lv_fail(ctx, LV_E_STATE, "rtao_geometry = triangle_tubes / geometry_mode = Triangle Mesh / the RTAO "
                                            "prebaker need lv_set_tube_triangle_mesh");
        if (!ctx->triAccelValid || ctx->triAccelLineWidth != ctx->opt.lineWidth)
            if ((rc = lv_bvh_build_triangles(ctx))) return rc;
    }

    hipStream_t st = ctx->stream;
    LvUniforms U;
    lv_fill_uniforms(ctx, U);
    // AO lookup of the colour pass: literal projection + bilinear sample for jittered primary rays (ray tracer only; the PPLL
    // fragments sit at pixel centres), the pixel's own texel otherwise
    U.aoProjectLookup = (mode == LV_RENDERING_MODE_VULKAN_RAY_TRACER && U.useJitteredRays && U.useAmbientOcclusion && !U.aoPrebaked) ? 1u : 0u;
    if ((rc = lv_buf_reserve(ctx, ctx->counters, sizeof(LvDevCounters)))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->depthMinMax, 16))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->tilesDev, size_t(numTiles) * 8))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->ao, size_t(ctx->width) * ctx->height * 4))) return rc;
    if (ctx->aoW != ctx->width || ctx->aoH != ctx->height) ctx->aoResult = nullptr; // images of another viewport size
    ctx->aoW = ctx->width;
    ctx->aoH = ctx->height;
    LvDevCounters* dc = (LvDevCounters*)ctx->counters.ptr;

    if (ctx->opt.timerMask >> 31) LV_HIP(ctx, hipEventRecord(ctx->ev[2], st));
    // The tile list is usually the same from frame to frame (a rank keeps its tiles): upload it only when it changes, so
    // that consecutive frames need no host synchronisation and the CPU can enqueue frame k+1 while frame k runs.
    const bool sameTiles = ctx->tilesUploaded && ctx->tilesHost.size() == 2 * size_t(numTiles) &&
                           memcmp(ctx->tilesHost.data(), tilesXYHost, size_t(numTiles) * 8) == 0;
    if (!sameTiles) {
        if (!ctx->tilesHost.empty()) LV_HIP(ctx, hipStreamSynchronize(st)); // previous staging copy may still be in flight
        ctx->tilesHost.assign(tilesXYHost, tilesXYHost + 2 * size_t(numTiles));
        LV_HIP(ctx, hipMemcpyAsync(ctx->tilesDev.ptr, ctx->tilesHost.data(), size_t(numTiles) * 8, hipMemcpyHostToDevice, st));
        ctx->tilesUploaded = true;
        ctx->tilesHaloUploaded = false;
        ctx->tilesGeneration++;
        ctx->tilesCoverKey[0] = 0u;   // (tilesCoverViewport is recomputed below)
    }
    if (ctx->tilesCoverKey[0] != ctx->width || ctx->tilesCoverKey[1] != ctx->height || ctx->tilesCoverKey[2] != tileW ||
        ctx->tilesCoverKey[3] != tileH) {
        // do the tiles cover every pixel of the viewport?  (sufficient test: grid-aligned tiles, every grid cell present)
        const uint32_t gx = (ctx->width + tileW - 1u) / tileW, gy = (ctx->height + tileH - 1u) / tileH;
        bool cover = false;
        if (uint64_t(gx) * gy <= uint64_t(numTiles)) {
            std::vector<uint8_t> cell(size_t(gx) * gy, 0);
            size_t present = 0;
            for (uint32_t t = 0; t < numTiles; t++) {
                const uint32_t tx = tilesXYHost[2 * t], ty = tilesXYHost[2 * t + 1];
                if (tx % tileW || ty % tileH || tx / tileW >= gx || ty / tileH >= gy) continue;
                uint8_t& c = cell[size_t(ty / tileH) * gx + tx / tileW];
                if (!c) { c = 1; present++; }
            }
            cover = present == size_t(gx) * gy;
        }
        ctx->tilesCoverViewport = cover;
        ctx->tilesCoverKey[0] = ctx->width; ctx->tilesCoverKey[1] = ctx->height; ctx->tilesCoverKey[2] = tileW; ctx->tilesCoverKey[3] = tileH;
    }
    LV_HIP(ctx, hipMemsetAsync(dc, 0, sizeof(LvDevCounters), st));

    LvTiles T{};
    T.tilesXY = (const uint32_t*)ctx->tilesDev.ptr;
    T.numTiles = numTiles;
    T.tileW = tileW;
    T.tileH = tileH;
    T.blocksX = ((tileW + 63u) / 64u) * 4u; // 16x16-pixel blocks, in whole 64x64 groups (lv_block_pixel)
    T.blocksY = ((tileH + 63u) / 64u) * 4u;
    const uint64_t nb = uint64_t(numTiles) * T.blocksX * T.blocksY;
    if (nb > 0x7FFFFFF0ull) return lv_fail(ctx, LV_E_INVALID, "tile list too large");
    ctx->groupOrderSorted = false;
    ctx->groupOrder[1].active = false;
    if ((rc = lv_group_order_prepare(ctx, T, 0))) return rc;
    const uint32_t gridTiles = uint32_t((nb + 127u) / 128u) * 128u; // multiple of 8 XCDs x LV_XCD_GROUP (lv_block_pixel)
    const uint64_t maxPixels = uint64_t(numTiles) * tileW * tileH;
    const bool stats = ctx->opt.collectStats;
    // overlap_primary_passes (default on): the colour pass' first-hit trace rides along with the RTAO primaries (k_primary_pair) and the
    // colour kernel after the RTAO pass only shades.  Where it applies: the ray tracer on the analytic capsules, one sample per pixel
    // (the first ray of a pixel is then its only first ray), the per-frame RTAO pass on the caller's tiles (SVGF covers the viewport).
    // "auto": while the colour pass' tile kernel is at most four rounds of workgroups (16 per CU; half a 1920 x 1080 frame) -- a frame
    // that fills the GPU several times over is throughput-bound in both passes and only pays the second shading (measured, EXPERIMENTS 13.2)
    const bool pairWanted = ctx->opt.overlapPrimaryPasses == 1 || (ctx->opt.overlapPrimaryPasses == 2 && gridTiles <= 16u * uint32_t(ctx->numCUs));
    const bool pairPrimaries = pairWanted && mode == LV_RENDERING_MODE_VULKAN_RAY_TRACER && U.useAmbientOcclusion &&
                               !U.aoPrebaked && !ctx->opt.useMlat && !ctx->opt.rtTriangleMesh && !U.useEllipticTubes &&
                               !ctx->opt.svgfEnabled && (U.useJitteredRays ? U.numSamplesPerFrame : 1u) == 1u && ctx->numSegs > 0u &&
                               (ctx->opt.numAccumulatedFrames <= 1u || ctx->opt.frameNumber < ctx->opt.aoIterations);
    {
        // one reservation for every launch geometry of this frame (a later, larger request would free the slab under the
        // scene views built before it)
        const bool aoRun = U.useAmbientOcclusion && !U.aoPrebaked, aoBake = U.aoPrebaked && !ctx->bakeValid;
        const bool triColour = ctx->opt.rtTriangleMesh && mode == LV_RENDERING_MODE_VULKAN_RAY_TRACER;
        size_t need = lv_overflow_bytes(ctx, gridTiles, LV_STACK_LDS, false);
        if (triColour) need = std::max(need, lv_overflow_bytes(ctx, gridTiles, LV_STACK_LDS, true));
        if (aoRun) {
            // with the AO halo the AO pass runs on tiles of (tileW + 2) x (tileH + 2) pixels (lv_run_ao)
            const uint32_t haloPx = (U.aoProjectLookup ? 1u : 0u) +
                                    ((ctx->opt.eawEnabled && ctx->opt.eawIterations) ? 2u * ((1u << ctx->opt.eawIterations) - 1u) : 0u);
            // ... and under SVGF on the whole viewport
            const bool svgf = ctx->opt.svgfEnabled;
            const uint64_t tw = svgf ? ctx->width : tileW + 2u * haloPx, th = svgf ? ctx->height : tileH + 2u * haloPx;
            const uint64_t nAo = svgf ? 1u : numTiles;
            const uint64_t nbAo = nAo * (((tw + 63u) / 64u) * 4u) * (((th + 63u) / 64u) * 4u);
            const uint64_t gridAo = ((nbAo + 127u) / 128u) * 128u;
            need = std::max(need, lv_overflow_bytes(ctx, std::max<uint64_t>(lv_ao_grid(ctx, nAo * tw * th * U.aoSamplesPerFrame), gridAo),
                                                    LV_AO_STACK_LDS, lv_ao_triangle_tubes(ctx)));
        }
        if (aoBake)
            need = std::max(need, lv_overflow_bytes(ctx, uint64_t(ctx->numCUs) * LV_AO_BLOCKS_PER_CU, LV_AO_STACK_LDS, true));
        if (pairPrimaries) {
            // k_primary_pair: the RTAO primaries' workgroups and the colour rays' share one launch and one slab
            const uint32_t haloPx = (U.aoProjectLookup ? 1u : 0u) +
                                    ((ctx->opt.eawEnabled && ctx->opt.eawIterations) ? 2u * ((1u << ctx->opt.eawIterations) - 1u) : 0u);
            const uint64_t tw = tileW + 2u * haloPx, th = tileH + 2u * haloPx;
            const uint64_t nbAo = uint64_t(numTiles) * (((tw + 63u) / 64u) * 4u) * (((th + 63u) / 64u) * 4u);
            const uint64_t gridPair = ((nbAo + 127u) / 128u) * 128u + gridTiles;
            need = std::max(need, lv_overflow_bytes(ctx, gridPair, LV_STACK_LDS, false));
            if (lv_ao_triangle_tubes(ctx)) need = std::max(need, lv_overflow_bytes(ctx, gridPair, LV_STACK_LDS, true));
        }
        if (need && (rc = lv_buf_reserve(ctx, ctx->stackOverflow, need))) return rc;
    }
    LvSceneDev S = sceneDev(ctx);
    if ((rc = lv_prepare_overflow(ctx, S, gridTiles))) return rc;

    // LineRenderer::renderBase: depth range, LineRenderer.cpp:248-256
    if (U.useDepthCues) {
        k_depth_init<<<1, 64, 0, st>>>(U, dc);
        if (ctx->numPoints)
            k_depth_minmax<<<nblocks(ctx->numPoints), LV_BLOCK, 0, st>>>(U, S.points, ctx->numPoints, dc);
        k_depth_finalize<<<1, 64, 0, st>>>(dc, (float*)ctx->depthMinMax.ptr);
    }
    if (ctx->opt.timerMask >> 31) LV_HIP(ctx, hipEventRecord(ctx->ev[5], st));

#ifdef LV_PROBE_OVERLAP
    // tools/variants.py probe (EXPERIMENTS.md 12.4): what overlapping the colour pass' traversal with the RTAO pass could win at most --
    // the WHOLE colour pass is launched on a second stream before the RTAO pass (it reads the previous frame's AO image: the picture is
    // wrong, the timing is an upper bound of the gain).  Never in the product build.
    static hipStream_t st2 = nullptr;
    static hipEvent_t evA = nullptr, evB = nullptr;
    if (!st2) {
        LV_HIP(ctx, hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
        LV_HIP(ctx, hipEventCreateWithFlags(&evA, hipEventDisableTiming));
        LV_HIP(ctx, hipEventCreateWithFlags(&evB, hipEventDisableTiming));
    }
    if (mode == LV_RENDERING_MODE_VULKAN_RAY_TRACER) {
        LV_HIP(ctx, hipEventRecord(evA, st));
        LV_HIP(ctx, hipStreamWaitEvent(st2, evA, 0));
        k_render_rt<false, LV_PRIM_CAPSULE, LV_SHADE_PLAIN><<<gridTiles, LV_BLOCK, 0, st2>>>(U, S, T, (uint32_t*)outDevice, dc);
        LV_HIP(ctx, hipEventRecord(evB, st2));
    }
#endif
    // ambientOcclusionBaker->updateIterative(), LineRenderer.cpp:257-264
    ctx->aoNumGroups = 0;
    bool firstHitsTraced = false;
    if (U.useAmbientOcclusion && !U.aoPrebaked) {
        LvSceneDev SP = S;
        if (pairPrimaries) {
            if (U.lssGeometry) SP.literalIntersection = 0u; // as the colour pass below
            if ((rc = lv_buf_reserve(ctx, ctx->firstHit, size_t(maxPixels) * 8 * LV_PRE_HITS))) return rc;
        }
        if ((rc = lv_run_ao(ctx, U, S, T, gridTiles, maxPixels, pairPrimaries ? &SP : nullptr, &firstHitsTraced))) return rc;
    }
    if (U.aoPrebaked && !ctx->bakeValid && ctx->bakeAsyncPending) {
        // a bake is running on the second stream (lv_bake_ao_start): adopt its table if it has finished; otherwise this frame is
        // rendered without ambient occlusion -- "display the AO once baking has finished" (AmbientOcclusionBaker.hpp:66-70)
        if ((rc = lv_bake_poll(ctx, false))) return rc;
        if (!ctx->bakeValid && ctx->bakeAsyncPending) { U.aoPrebaked = 0u; U.useAmbientOcclusion = 0u; }
    }
    if (U.aoPrebaked) {
        // static prebaker: view independent, (re)baked only when geometry or baking settings changed
        if (!ctx->bakeValid) {
            if ((rc = lv_bake_ambient_occlusion(ctx))) return rc;
            if ((rc = lv_prepare_overflow(ctx, S, gridTiles))) return rc; // the bake may have regrown the overflow slab
        }
        S.bakedAo = (const float*)ctx->bakedAo.ptr;
        S.bakedBlendingWeights = (const float*)ctx->bakeBlendingWeights.ptr;
    }
    if (ctx->opt.timerMask >> 31) LV_HIP(ctx, hipEventRecord(ctx->ev[7], st));
    // The colour pass of a raster_prism frame with the segment rasteriser has no tile kernel: only its resolve pass could use the
    // dispatch order, and measured it does not (config 4: 0.134 ms as numbered, 0.145 ms heaviest first + 11 us for k_group_order)
    if (mode == LV_RENDERING_MODE_PER_PIXEL_LINKED_LIST && lv_ppll_prism_source(ctx) && !ctx->opt.ppllPrismLbvhWalk &&
        !ctx->groupOrderSorted) {
        T.groupOrder = nullptr;
        T.groupCost = nullptr;
        ctx->groupOrder[0].active = false;
    }
    if ((rc = lv_group_order_sort(ctx))) return rc; // (queued by the RTAO pass already when that ran)

    uint32_t* out = (uint32_t*)outDevice;
    if (mode == LV_RENDERING_MODE_VULKAN_RAY_TRACER && ctx->opt.numAccumulatedFrames > 1u) {
        // the running mean lives in a full-viewport rgba8 image: a resize restarts it (onResolutionChanged resets
        // accumulatedFramesCounter, VulkanRayTracer.cpp:119-129)
        if (U.frameNumber != 0u && (ctx->accumW != ctx->width || ctx->accumH != ctx->height))
            return lv_fail(ctx, LV_E_STATE, "the viewport changed (%ux%u -> %ux%u): restart the accumulation with frame_number = 0",
                           ctx->accumW, ctx->accumH, ctx->width, ctx->height);
        if ((rc = lv_buf_reserve(ctx, ctx->accum, size_t(ctx->width) * ctx->height * 4))) return rc;
        ctx->accumW = ctx->width;
        ctx->accumH = ctx->height;
        S.accum = (uint32_t*)ctx->accum.ptr;
    }
    if (mode == LV_RENDERING_MODE_VULKAN_RAY_TRACER) {
        // geometry_mode (VulkanRayTracer.cpp:226-250): analytic capsules, or the triangle tubes with their own LBVH
        const bool tri = ctx->opt.rtTriangleMesh;
        if (ctx->opt.useMlat) { // use_mlat: single-pass approximate transparency (lv_mlat.hip)
            LvSceneDev SM = tri ? sceneDevTriangles(ctx) : S;
            SM.accum = S.accum;
            if (U.lssGeometry) SM.literalIntersection = 0u; // the hardware primitive has no intersection shader
            if (tri && (rc = lv_prepare_overflow(ctx, SM, gridTiles, LV_STACK_LDS, true))) return rc;
            if ((rc = lv_mlat_render(ctx, U, SM, T, gridTiles, out, dc, tri))) return rc;
        } else {
        LvSceneDev SC = tri ? sceneDevTriangles(ctx) : S;
        SC.accum = S.accum;
        if (U.lssGeometry) SC.literalIntersection = 0u; // the hardware primitive has no intersection shader
        if (tri && (rc = lv_prepare_overflow(ctx, SC, gridTiles, LV_STACK_LDS, true))) return rc;
#define LV_LAUNCH_RT(ST, PR, BA) \
    LV_TIMED_LAUNCH(ctx, LV_KERNEL_RENDER_RT, (k_render_rt<ST, PR, BA><<<gridTiles, LV_BLOCK, 0, st>>>(U, SC, T, out, dc)))
#define LV_LAUNCH_RT_PRE(ST, BA) \
    LV_TIMED_LAUNCH(ctx, LV_KERNEL_RENDER_RT, (k_render_rt<ST, LV_PRIM_CAPSULE, BA, true><<<gridTiles, LV_BLOCK, 0, st>>>( \
            U, SC, T, out, dc, (const uint2*)ctx->firstHit.ptr, size_t(maxPixels))))
#define LV_LAUNCH_RT2(ST)                                                        \
    do {                                                                         \
        if (fastPlain && firstHitsTraced)                                        \
            LV_TIMED_LAUNCH(ctx, LV_KERNEL_RENDER_RT, (k_render_rt<false, LV_PRIM_CAPSULE, LV_SHADE_PLAIN, true, 1><<<gridTiles, LV_BLOCK, 0, st>>>( \
                    U, SC, T, out, dc, (const uint2*)ctx->firstHit.ptr, size_t(maxPixels)))); \
        else if (fastPlain)                                                      \
            LV_TIMED_LAUNCH(ctx, LV_KERNEL_RENDER_RT, (k_render_rt<false, LV_PRIM_CAPSULE, LV_SHADE_PLAIN, false, 1><<<gridTiles, LV_BLOCK, 0, st>>>( \
                    U, SC, T, out, dc)));                                        \
        else if (firstHitsTraced && U.useHelicityBands) LV_LAUNCH_RT_PRE(ST, LV_SHADE_HELICITY); \
        else if (firstHitsTraced && U.useBands) LV_LAUNCH_RT_PRE(ST, LV_SHADE_BANDS); \
        else if (firstHitsTraced) LV_LAUNCH_RT_PRE(ST, LV_SHADE_PLAIN);          \
        else if (tri && U.useHelicityBands) LV_LAUNCH_RT(ST, LV_PRIM_TRIANGLE, LV_SHADE_HELICITY); \
        else if (tri && U.useBands) LV_LAUNCH_RT(ST, LV_PRIM_TRIANGLE, LV_SHADE_BANDS); \
        else if (tri) LV_LAUNCH_RT(ST, LV_PRIM_TRIANGLE, LV_SHADE_PLAIN);        \
        else if (U.useHelicityBands) LV_LAUNCH_RT(ST, LV_PRIM_CAPSULE, LV_SHADE_HELICITY); \
        else if (U.useEllipticTubes) LV_LAUNCH_RT(ST, LV_PRIM_ELLIPTIC, LV_SHADE_BANDS); \
        else if (U.useBands) LV_LAUNCH_RT(ST, LV_PRIM_CAPSULE, LV_SHADE_BANDS);  \
        else LV_LAUNCH_RT(ST, LV_PRIM_CAPSULE, LV_SHADE_PLAIN);                  \
    } while (0)
        // shading_numerics = fast: the capsule colour pass of plain flow lines (lighting through the approximate operations, FAST = 1)
        const bool fastPlain = ctx->opt.fastShading && !stats && !tri && !U.useHelicityBands && !U.useEllipticTubes && !U.useBands;
#ifdef LV_PROBE_OVERLAP
        LV_HIP(ctx, hipStreamWaitEvent(st, evB, 0));
#else
        if (stats) LV_LAUNCH_RT2(true); else LV_LAUNCH_RT2(false);
#endif
#undef LV_LAUNCH_RT2
#undef LV_LAUNCH_RT_PRE
#undef LV_LAUNCH_RT
        }
    } else {
        // reallocateFragmentBuffer, PerPixelLinkedListLineRenderer.cpp:251-357
        // gather(): fragments of the rasterised programmable-pull prism (the reference's geometry, default) or capsule entry hits
        const bool prismSource = lv_ppll_prism_source(ctx);
        if (prismSource && ctx->opt.useRibbons && ctx->opt.helicityBands)
            return lv_fail(ctx, LV_E_INVALID, "ppll_fragment_source = raster_prism: band data with rotating helicity bands has no prism "
                                              "fragment stage (use auto or capsule_entry)");
        if (prismSource && ctx->opt.useRibbons && ctx->opt.ppllPrismLbvhWalk)
            return lv_fail(ctx, LV_E_INVALID, "ppll_prism_rasteriser = lbvh: the segment boxes do not enclose the band prism "
                                              "(band_width / 2); band data uses the segment rasteriser");
        if (prismSource) lv_fill_prism(ctx, U, S.prism);
        const uint32_t numSlices = prismSource ? 1u : LV_PPLL_SLICES; // (the prism's coverage kernel has no depth: one slice)
        // ppll_prism_rasteriser: "segments" (default) = one lane per segment over its screen rectangle, "lbvh" = the all-hits walk of
        // the viewing rays (k_ppll_gather<LV_PRIM_PRISM>); same coverage test, same fragments
        const bool segmentRaster = prismSource && !ctx->opt.ppllPrismLbvhWalk;
        const uint32_t rasterGrid = uint32_t(ctx->numCUs) * LV_PRISM_RASTER_BLOCKS_PER_CU;
        if (uint64_t(gridTiles) * numSlices > 0x7FFFFFF0ull) return lv_fail(ctx, LV_E_INVALID, "tile list too large");
        // physical pool = the reference's linkedListSize + the tail every wave of the gather may leave unused in its last
        // chunk of node slots (k_ppll_gather / k_ppll_raster_prism), so that the effective capacity is never below the reference's
        uint64_t poolSlots64 = uint64_t(U.ppllLinkedListSize) +
                               (segmentRaster ? uint64_t(rasterGrid) * (LV_BLOCK / LV_WAVE) * LV_PRISM_RASTER_CHUNK
                                              : uint64_t(gridTiles) * numSlices * (LV_BLOCK / LV_WAVE) * LV_PPLL_CHUNK);
        if (poolSlots64 > 0xFFFFFFF0ull) poolSlots64 = 0xFFFFFFF0ull; // node indices are 32 bit
        const uint32_t poolSlots = uint32_t(poolSlots64);
        if ((rc = lv_buf_reserve(ctx, ctx->ppllNodes, size_t(poolSlots) * 12))) return rc;
        // raster_prism: the coverage kernel writes 12-B records {pixel, leaf | triangle, rank} into a pool of their own; ppllNodes then
        // holds the fragment array (8-B {colour, depth} entries, one contiguous run per pixel), ppllStart the runs' offsets
        if (prismSource && (rc = lv_buf_reserve(ctx, ctx->prismRecords, size_t(poolSlots) * 12))) return rc;
        ctx->ppllArrays = prismSource;
        uint32_t* gatherPool = prismSource ? (uint32_t*)ctx->prismRecords.ptr : (uint32_t*)ctx->ppllNodes.ptr;
        const size_t padded4 = (size_t(U.ppllPaddedW) * U.ppllPaddedH + 3) / 4; // cleared as whole uint4s (k_ppll_clear)
        if ((rc = lv_buf_reserve(ctx, ctx->ppllStart, padded4 * 16))) return rc;
        if ((rc = lv_buf_reserve(ctx, ctx->ppllCount, padded4 * 16))) return rc;
        ctx->ppllPoolNodes = poolSlots;
        ctx->ppllPaddedW = U.ppllPaddedW;
        ctx->ppllPaddedH = U.ppllPaddedH;
        // clear(): LinkedListClear.glsl:46-55 + fragmentCounterBuffer->fill(0)
        const bool allRequested = segmentRaster && ctx->tilesCoverViewport;
        k_ppll_clear<<<uint32_t((padded4 + LV_BLOCK - 1) / LV_BLOCK), LV_BLOCK, 0, st>>>(
                (uint4*)ctx->ppllStart.ptr, (uint4*)ctx->ppllCount.ptr, padded4, dc, allRequested ? 0u : 0xFFFFFFFFu,
                (allRequested && stats) ? (unsigned long long)U.width * U.height : 0ull);
        if (ctx->opt.timerMask >> 31) LV_HIP(ctx, hipEventRecord(ctx->ev[11], st));
#define LV_LAUNCH_GATHER(ST, PR, BA)                                                                             \
    LV_TIMED_LAUNCH(ctx, LV_KERNEL_PPLL_GATHER, (k_ppll_gather<ST, PR, BA><<<gridTiles * numSlices, LV_BLOCK, 0, st>>>( \
            U, S, T, gatherPool, (uint32_t*)ctx->ppllStart.ptr,                                                  \
            (uint32_t*)ctx->ppllCount.ptr, dc, numSlices, poolSlots)))
#define LV_LAUNCH_GATHER2(ST)                                                          \
    do {                                                                               \
        if (prismSource) LV_LAUNCH_GATHER(ST, LV_PRIM_PRISM, LV_SHADE_PLAIN);          \
        else if (U.useHelicityBands) LV_LAUNCH_GATHER(ST, LV_PRIM_CAPSULE, LV_SHADE_HELICITY); \
        else if (U.useBands && U.useEllipticTubes) LV_LAUNCH_GATHER(ST, LV_PRIM_ELLIPTIC, LV_SHADE_BANDS); \
        else if (U.useBands) LV_LAUNCH_GATHER(ST, LV_PRIM_CAPSULE, LV_SHADE_BANDS);    \
        else LV_LAUNCH_GATHER(ST, LV_PRIM_CAPSULE, LV_SHADE_PLAIN);                    \
    } while (0)
        if (segmentRaster) {
            // the segment rasteriser (k_ppll_raster_prism): requested pixels marked, then one lane per segment
            uint32_t* coarse = nullptr;
            if (!allRequested) {
                const size_t cells = size_t((U.width + LV_PRISM_COARSE - 1u) / LV_PRISM_COARSE) * ((U.height + LV_PRISM_COARSE - 1u) / LV_PRISM_COARSE);
                if ((rc = lv_buf_reserve(ctx, ctx->ppllCoarse, cells * 4))) return rc;
                coarse = (uint32_t*)ctx->ppllCoarse.ptr;
                LV_HIP(ctx, hipMemsetAsync(coarse, 0, cells * 4, st));
            }
#define LV_LAUNCH_RASTER(ST, NT)                                                                                              \
    k_ppll_raster_prism<ST, NT><<<rasterGrid, LV_BLOCK, 0, st>>>(                                                             \
            U, S, gatherPool, (const uint32_t*)ctx->ppllStart.ptr, (uint32_t*)ctx->ppllCount.ptr, dc, poolSlots,                \
            allRequested ? 1u : 0u, leafList)
            uint32_t* leafList = nullptr;
            // LV_KERNEL_PPLL_RASTER times the rasteriser WITH what a sharded frame runs in front of it (k_ppll_mark_tiles,
            // k_ppll_cull_segments): a rank's rasteriser cost is the three together
            const bool rasterTimed = ((ctx->opt.timerMask >> LV_KERNEL_PPLL_RASTER) & 1u) != 0u;
            if (rasterTimed) LV_HIP(ctx, hipEventRecord(lv_kernel_ev(ctx, LV_KERNEL_PPLL_RASTER, 0), st));
            if (allRequested) {}   // (k_ppll_clear marked every pixel)
            else if (stats) k_ppll_mark_tiles<true><<<gridTiles, LV_BLOCK, 0, st>>>(U, T, (uint32_t*)ctx->ppllStart.ptr, coarse, dc);
            else k_ppll_mark_tiles<false><<<gridTiles, LV_BLOCK, 0, st>>>(U, T, (uint32_t*)ctx->ppllStart.ptr, coarse, dc);
            if (!allRequested && S.numSegs != 0) {
                // the segments that can touch this tile list (k_ppll_clear zeroed the list's counter)
                if ((rc = lv_buf_reserve(ctx, ctx->prismLeafList, size_t(S.numSegs) * 4))) return rc;
                leafList = (uint32_t*)ctx->prismLeafList.ptr;
                const uint32_t cullGrid = (S.numSegs + LV_CULL_PER_BLOCK - 1u) / LV_CULL_PER_BLOCK;
                k_ppll_cull_segments<<<cullGrid, LV_BLOCK, 0, st>>>(U, S, coarse, leafList, dc);
            }
            if (S.numSegs != 0) {
                if (S.prism.n == 6u) { if (stats) LV_LAUNCH_RASTER(true, 6); else LV_LAUNCH_RASTER(false, 6); }
                else { if (stats) LV_LAUNCH_RASTER(true, 0); else LV_LAUNCH_RASTER(false, 0); }
            }
            if (rasterTimed) {
                LV_HIP(ctx, hipEventRecord(lv_kernel_ev(ctx, LV_KERNEL_PPLL_RASTER, 1), st));
                ctx->kernelLaunches[LV_KERNEL_PPLL_RASTER]++;
            }
#undef LV_LAUNCH_RASTER
        } else {
        if (stats) LV_LAUNCH_GATHER2(true); else LV_LAUNCH_GATHER2(false);
        }
#undef LV_LAUNCH_GATHER2
#undef LV_LAUNCH_GATHER
        uint32_t* blockBase = nullptr;
        if (prismSource) {
            // per-pixel counts -> offsets of the pixels' runs + the list of the pixels with more records than the sort arrays hold
            const uint32_t numAddr = uint32_t(padded4 * 4);
            const uint32_t scanBlocks = (numAddr + LV_SCAN_ITEMS - 1u) / LV_SCAN_ITEMS;
            if ((rc = lv_buf_reserve(ctx, ctx->ppllOverflow, padded4 * 16))) return rc;
            if ((rc = lv_buf_reserve(ctx, ctx->scanTemp, (2 * size_t(scanBlocks) + 1) * 4))) return rc;
            uint32_t* blockTotals = (uint32_t*)ctx->scanTemp.ptr;
            blockBase = blockTotals + scanBlocks;
            k_ppll_scan<<<scanBlocks, LV_BLOCK, 0, st>>>((const uint32_t*)ctx->ppllCount.ptr, (uint32_t*)ctx->ppllStart.ptr, numAddr,
                                                        blockTotals, U.ppllMaxNumFrags, (uint32_t*)ctx->ppllOverflow.ptr, dc);
            k_ppll_scan_bases<<<1, LV_BLOCK, 0, st>>>(blockTotals, blockBase, scanBlocks);
            ctx->ppllScanBlocks = scanBlocks;
            const uint32_t shadeGrid = uint32_t(ctx->numCUs) * LV_PRISM_SHADE_BLOCKS_PER_CU;
#define LV_LAUNCH_SHADE(ST)                                                                                                     \
    if (S.prism.bands)                                                                                                          \
    LV_TIMED_LAUNCH(ctx, LV_KERNEL_PPLL_SHADE, (k_ppll_shade_prism<ST, LV_SHADE_BANDS><<<shadeGrid, LV_BLOCK, 0, st>>>(         \
            U, S, (const uint32_t*)ctx->prismRecords.ptr, (uint2*)ctx->ppllNodes.ptr, (const uint32_t*)ctx->ppllStart.ptr,      \
            blockBase, (uint32_t*)ctx->ppllCount.ptr, dc, poolSlots)));                                                         \
    else if (U.useHelicityBands)                                                                                                \
    LV_TIMED_LAUNCH(ctx, LV_KERNEL_PPLL_SHADE, (k_ppll_shade_prism<ST, LV_SHADE_HELICITY><<<shadeGrid, LV_BLOCK, 0, st>>>(      \
            U, S, (const uint32_t*)ctx->prismRecords.ptr, (uint2*)ctx->ppllNodes.ptr, (const uint32_t*)ctx->ppllStart.ptr,      \
            blockBase, (uint32_t*)ctx->ppllCount.ptr, dc, poolSlots)));                                                         \
    else                                                                                                                        \
    LV_TIMED_LAUNCH(ctx, LV_KERNEL_PPLL_SHADE, (k_ppll_shade_prism<ST><<<shadeGrid, LV_BLOCK, 0, st>>>(                         \
            U, S, (const uint32_t*)ctx->prismRecords.ptr, (uint2*)ctx->ppllNodes.ptr, (const uint32_t*)ctx->ppllStart.ptr,      \
            blockBase, (uint32_t*)ctx->ppllCount.ptr, dc, poolSlots)))
            // shading_numerics = fast: the plain-tube fragment stage with the raster colour (the only variant whose alpha cannot follow
            // the halo coordinate); every other variant keeps the exact arithmetic
            if (ctx->opt.fastShading && !stats && !S.prism.bands && !U.useHelicityBands && U.ppllRasterColour)
                LV_TIMED_LAUNCH(ctx, LV_KERNEL_PPLL_SHADE, (k_ppll_shade_prism<false, LV_SHADE_PLAIN, 2><<<shadeGrid, LV_BLOCK, 0, st>>>(
                        U, S, (const uint32_t*)ctx->prismRecords.ptr, (uint2*)ctx->ppllNodes.ptr, (const uint32_t*)ctx->ppllStart.ptr,
                        blockBase, (uint32_t*)ctx->ppllCount.ptr, dc, poolSlots)));
            else if (stats) LV_LAUNCH_SHADE(true); else LV_LAUNCH_SHADE(false);
#undef LV_LAUNCH_SHADE
            // the listed pixels: their nearest ppllMaxNumFrags fragments to the front of the run
            k_ppll_select_nearest<<<uint32_t(ctx->numCUs) * 16u, LV_WAVE, 0, st>>>(
                    U, (uint2*)ctx->ppllNodes.ptr, (uint2*)ctx->prismRecords.ptr, (const uint32_t*)ctx->ppllStart.ptr, blockBase,
                    (const uint32_t*)ctx->ppllCount.ptr, (const uint32_t*)ctx->ppllOverflow.ptr, dc);
        }
        if (ctx->opt.timerMask >> 31) LV_HIP(ctx, hipEventRecord(ctx->ev[13], st));
        // resolve()
        const uint32_t* prismCount = prismSource ? (const uint32_t*)ctx->ppllCount.ptr : nullptr; // (kept fragments per pixel -> max depth complexity)
        const uint64_t groups64 = uint64_t(numTiles) * (T.blocksX / 4u) * (T.blocksY / 4u) * 64u;   // 8 x 8 cells of the 64 x 64 groups
        if (groups64 > 0xFFFFFFF0ull) return lv_fail(ctx, LV_E_INVALID, "tile list too large");
        const uint32_t numGroups = uint32_t(groups64);
        const size_t ldsBytes = size_t(U.ppllMaxNumFrags) * LV_WAVE * 8;
#define LV_LAUNCH_RESOLVE(LDS, PQ, GRID, BYTES, SCRATCH)                                                                        \
    do {                                                                                                                       \
        if (prismSource)                                                                                                       \
            LV_TIMED_LAUNCH(ctx, LV_KERNEL_PPLL_RESOLVE, (k_ppll_resolve<LDS, PQ, true><<<GRID, LV_WAVE, BYTES, st>>>(        \
                    U, T, (const uint32_t*)ctx->ppllNodes.ptr, (const uint32_t*)ctx->ppllStart.ptr, out, SCRATCH, numGroups,   \
                    prismCount, dc, blockBase)));                                                                              \
        else                                                                                                                   \
            LV_TIMED_LAUNCH(ctx, LV_KERNEL_PPLL_RESOLVE, (k_ppll_resolve<LDS, PQ, false><<<GRID, LV_WAVE, BYTES, st>>>(       \
                    U, T, (const uint32_t*)ctx->ppllNodes.ptr, (const uint32_t*)ctx->ppllStart.ptr, out, SCRATCH, numGroups,   \
                    nullptr, dc, nullptr)));                                                                                   \
    } while (0)
        if (ldsBytes <= LV_RESOLVE_LDS_MAX) {
            if (U.ppllSortingMode == 0u) LV_LAUNCH_RESOLVE(true, true, numGroups, ldsBytes, nullptr);
            else LV_LAUNCH_RESOLVE(true, false, numGroups, ldsBytes, nullptr);
        } else {
            const uint32_t grid = numGroups < LV_RESOLVE_SLAB_GRID ? numGroups : LV_RESOLVE_SLAB_GRID;
            if ((rc = lv_buf_reserve(ctx, ctx->ppllScratch, size_t(grid) * ldsBytes))) return rc;
            if (U.ppllSortingMode == 0u) LV_LAUNCH_RESOLVE(false, true, grid, 0, (uint32_t*)ctx->ppllScratch.ptr);
            else LV_LAUNCH_RESOLVE(false, false, grid, 0, (uint32_t*)ctx->ppllScratch.ptr);
        }
#undef LV_LAUNCH_RESOLVE
    }
    LV_HIP(ctx, hipGetLastError());
    if (ctx->opt.timerMask >> 31) LV_HIP(ctx, hipEventRecord(ctx->ev[3], st));
    ctx->evFrameValid = true;
    ctx->evPhaseRecorded = (ctx->opt.timerMask >> 31) != 0u;
    ctx->lastMode = mode;
    return LV_OK;
}

int lv_frame_trace_rays(lv_ctx* ctx, const float* o, const float* d, float tMin, float tMax, uint32_t n, float* outT,
                        uint32_t* outSeg, uint32_t* outKind) {
    int rc;
    if (!ctx->accelValid || ctx->accelLineWidth != lv_accel_width(ctx))
        if ((rc = lv_bvh_build(ctx))) return rc;
    if (n == 0) return LV_OK;
    hipStream_t st = ctx->stream;
    const size_t rb = size_t(n) * 12;
    if ((rc = lv_buf_reserve(ctx, ctx->scratchRays, 2 * rb + size_t(n) * 12))) return rc;
    char* base = (char*)ctx->scratchRays.ptr;
    float* dO = (float*)base;
    float* dD = (float*)(base + rb);
    float* dT = (float*)(base + 2 * rb);
    uint32_t* dS = (uint32_t*)(base + 2 * rb + size_t(n) * 4);
    uint32_t* dK = (uint32_t*)(base + 2 * rb + size_t(n) * 8);
    LV_HIP(ctx, hipMemcpyAsync(dO, o, rb, hipMemcpyHostToDevice, st));
    LV_HIP(ctx, hipMemcpyAsync(dD, d, rb, hipMemcpyHostToDevice, st));
    LvSceneDev S = sceneDev(ctx);
    if ((rc = lv_prepare_overflow(ctx, S, nblocks(n)))) return rc;
    if (ctx->opt.useRibbons && ctx->opt.ellipticTubes) // the elliptic tubelets of the band data (kind = 0)
        k_trace_rays<LV_PRIM_ELLIPTIC><<<nblocks(n), LV_BLOCK, 0, st>>>(S, ctx->opt.lineWidth * 0.5f, ctx->opt.useCappedTubes, dO,
                                                                        dD, tMin, tMax, n, dT, dS, dK);
    else
        k_trace_rays<LV_PRIM_CAPSULE><<<nblocks(n), LV_BLOCK, 0, st>>>(S, ctx->opt.lineWidth * 0.5f, ctx->opt.useCappedTubes, dO,
                                                                       dD, tMin, tMax, n, dT, dS, dK);
    LV_HIP(ctx, hipGetLastError());
    LV_HIP(ctx, hipMemcpyAsync(outT, dT, size_t(n) * 4, hipMemcpyDeviceToHost, st));
    LV_HIP(ctx, hipMemcpyAsync(outSeg, dS, size_t(n) * 4, hipMemcpyDeviceToHost, st));
    LV_HIP(ctx, hipMemcpyAsync(outKind, dK, size_t(n) * 4, hipMemcpyDeviceToHost, st));
    LV_HIP(ctx, hipStreamSynchronize(st));
    return LV_OK;
}

int lv_frame_ppll_resolve_only(lv_ctx* ctx, const uint32_t* nodes, uint64_t numNodes, const uint32_t* start,
                               uint64_t numPixels, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint8_t* out) {
    if (!ctx->cameraSet) return lv_fail(ctx, LV_E_STATE, "lv_set_camera has not been called");
    LvUniforms U;
    lv_fill_uniforms(ctx, U);
    if (numPixels != uint64_t(U.ppllPaddedW) * U.ppllPaddedH)
        return lv_fail(ctx, LV_E_INVALID, "start_offset must hold padded_w * padded_h = %llu entries",
                       (unsigned long long)(uint64_t(U.ppllPaddedW) * U.ppllPaddedH));
    hipStream_t st = ctx->stream;
    int rc;
    if ((rc = lv_buf_reserve(ctx, ctx->ppllNodes, size_t(numNodes ? numNodes : 1) * 12))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->ppllStart, size_t(numPixels) * 4))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->tilesDev, 8))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->outDev, size_t(w) * h * 4))) return rc;
    if (numNodes) LV_HIP(ctx, hipMemcpyAsync(ctx->ppllNodes.ptr, nodes, size_t(numNodes) * 12, hipMemcpyHostToDevice, st));
    LV_HIP(ctx, hipMemcpyAsync(ctx->ppllStart.ptr, start, size_t(numPixels) * 4, hipMemcpyHostToDevice, st));
    uint32_t txy[2] = {x0, y0};
    ctx->tilesUploaded = false; // tilesDev is overwritten below
    LV_HIP(ctx, hipMemcpyAsync(ctx->tilesDev.ptr, txy, 8, hipMemcpyHostToDevice, st));
    LV_HIP(ctx, hipStreamSynchronize(st));
    LvTiles T{};
    T.tilesXY = (const uint32_t*)ctx->tilesDev.ptr;
    T.numTiles = 1; T.tileW = w; T.tileH = h;
    T.blocksX = ((w + 63u) / 64u) * 4u; T.blocksY = ((h + 63u) / 64u) * 4u;
    const uint32_t numGroups = (T.blocksX / 4u) * (T.blocksY / 4u) * 64u;
    const size_t ldsBytes = size_t(U.ppllMaxNumFrags) * LV_WAVE * 8;
    const uint32_t* nd = (const uint32_t*)ctx->ppllNodes.ptr;
    const uint32_t* so = (const uint32_t*)ctx->ppllStart.ptr;
    uint32_t* od = (uint32_t*)ctx->outDev.ptr;
    const bool pq = U.ppllSortingMode == 0u;
    if (ldsBytes <= LV_RESOLVE_LDS_MAX) {
        if (pq) k_ppll_resolve<true, true><<<numGroups, LV_WAVE, ldsBytes, st>>>(U, T, nd, so, od, nullptr, numGroups, nullptr, nullptr, nullptr);
        else k_ppll_resolve<true, false><<<numGroups, LV_WAVE, ldsBytes, st>>>(U, T, nd, so, od, nullptr, numGroups, nullptr, nullptr, nullptr);
    } else {
        const uint32_t grid = numGroups < LV_RESOLVE_SLAB_GRID ? numGroups : LV_RESOLVE_SLAB_GRID;
        if ((rc = lv_buf_reserve(ctx, ctx->ppllScratch, size_t(grid) * ldsBytes))) return rc;
        uint32_t* sc = (uint32_t*)ctx->ppllScratch.ptr;
        if (pq) k_ppll_resolve<false, true><<<grid, LV_WAVE, 0, st>>>(U, T, nd, so, od, sc, numGroups, nullptr, nullptr, nullptr);
        else k_ppll_resolve<false, false><<<grid, LV_WAVE, 0, st>>>(U, T, nd, so, od, sc, numGroups, nullptr, nullptr, nullptr);
    }
    LV_HIP(ctx, hipGetLastError());
    LV_HIP(ctx, hipMemcpyAsync(out, ctx->outDev.ptr, size_t(w) * h * 4, hipMemcpyDeviceToHost, st));
    LV_HIP(ctx, hipStreamSynchronize(st));
    return LV_OK;
}

int lv_frame_trace_rays_triangles(lv_ctx* ctx, const float* o, const float* d, float tMin, float tMax, uint32_t n,
                                  float* outT, uint32_t* outTri, float* outUV) {
    int rc;
    if ((rc = lv_ensure_tube_mesh(ctx))) return rc;
    if (!ctx->triMeshSet) return lv_fail(ctx, LV_E_STATE, "lv_set_tube_triangle_mesh has not been called");
    if (!ctx->triAccelValid || ctx->triAccelLineWidth != ctx->opt.lineWidth)
        if ((rc = lv_bvh_build_triangles(ctx))) return rc;
    if (n == 0) return LV_OK;
    hipStream_t st = ctx->stream;
    const size_t rb = size_t(n) * 12;
    if ((rc = lv_buf_reserve(ctx, ctx->scratchRays, 2 * rb + size_t(n) * 16))) return rc;
    char* base = (char*)ctx->scratchRays.ptr;
    float* dO = (float*)base;
    float* dD = (float*)(base + rb);
    float* dT = (float*)(base + 2 * rb);
    uint32_t* dS = (uint32_t*)(base + 2 * rb + size_t(n) * 4);
    float* dUV = (float*)(base + 2 * rb + size_t(n) * 8);
    LV_HIP(ctx, hipMemcpyAsync(dO, o, rb, hipMemcpyHostToDevice, st));
    LV_HIP(ctx, hipMemcpyAsync(dD, d, rb, hipMemcpyHostToDevice, st));
    LvSceneDev S = sceneDevTriangles(ctx);
    if ((rc = lv_prepare_overflow(ctx, S, nblocks(n), LV_STACK_LDS, true))) return rc;
    k_trace_rays_tri<<<nblocks(n), LV_BLOCK, 0, st>>>(S, dO, dD, tMin, tMax, n, dT, dS, dUV);
    LV_HIP(ctx, hipGetLastError());
    LV_HIP(ctx, hipMemcpyAsync(outT, dT, size_t(n) * 4, hipMemcpyDeviceToHost, st));
    LV_HIP(ctx, hipMemcpyAsync(outTri, dS, size_t(n) * 4, hipMemcpyDeviceToHost, st));
    if (outUV) LV_HIP(ctx, hipMemcpyAsync(outUV, dUV, size_t(n) * 8, hipMemcpyDeviceToHost, st));
    LV_HIP(ctx, hipStreamSynchronize(st));
    return LV_OK;
}

// VulkanAmbientOcclusionBaker::bakeAoTexture (VulkanAmbientOcclusionBaker.cpp:193-262): maxNumIterations passes of the
// baking shader, each averaging numAmbientOcclusionSamplesPerFrame rays per (parametrisation vertex, tube subdivision),
// accumulated as a running mean.  Rays hit the triangle tubes (the reference binds the triangle TLAS, cpp:480) and run
// through the same persistent work-queue kernel as the screen-space pass.
// async == false: on the context's stream with the frame's scratch buffers; the table is valid when the call returns (to the stream).
// async == true (lv_bake_ao_start; the reference's BakingMode::MULTI_THREADED, VulkanAmbientOcclusionBaker.cpp:266-346): on a second
// stream, into a second table, with scratch buffers and counters of its own -- frames keep rendering on the context's stream (without
// AO, or with the previous table's AO while that is still valid) until lv_bake_poll finds the stream finished and swaps the tables.
int lv_bake_ambient_occlusion(lv_ctx* ctx, bool async) {
    const LvOptions& o = ctx->opt;
    {
        const int rcMesh = lv_ensure_tube_mesh(ctx);
        if (rcMesh) return rcMesh;
    }
    if (!ctx->triMeshSet) return lv_fail(ctx, LV_E_STATE, "the RTAO prebaker needs lv_set_tube_triangle_mesh");
    if (!ctx->bakeParamSet) return lv_fail(ctx, LV_E_STATE, "the RTAO prebaker needs lv_set_ao_parametrization");
    if (ctx->bakeNumLineVertices != ctx->numTriPoints)
        return lv_fail(ctx, LV_E_INVALID, "blending weights (%u) must match the mesh's line points (%u)",
                       ctx->bakeNumLineVertices, ctx->numTriPoints);
    int rc;
    if (!ctx->triAccelValid || ctx->triAccelLineWidth != o.lineWidth)
        if ((rc = lv_bvh_build_triangles(ctx))) return rc;
    if (async && !ctx->bakeStream) {
        LV_HIP(ctx, hipStreamCreateWithFlags(&ctx->bakeStream, hipStreamNonBlocking));
        LV_HIP(ctx, hipEventCreateWithFlags(&ctx->evBakePrereq, hipEventDisableTiming));
        LV_HIP(ctx, hipEventCreateWithFlags(&ctx->evBakeDone, hipEventDisableTiming));
    }
    hipStream_t st = async ? ctx->bakeStream : ctx->stream;
    const uint32_t N = o.bakeNumTubeSubdivisions, spp = o.bakeSamplesPerFrame, M = ctx->bakeNumParametrizationVertices;
    const uint64_t slots = uint64_t(M) * N;
    if (slots > 0x7FFFFFFFull) return lv_fail(ctx, LV_E_CAPACITY, "too many AO bake entries");
    LvDeviceBuffer& counters = async ? ctx->bakeCounters : ctx->counters;
    LvDeviceBuffer& gbuf = async ? ctx->bakeGbuf : ctx->aoGbuf;
    LvDeviceBuffer& samples = async ? ctx->bakeSamples : ctx->aoSamples;
    LvDeviceBuffer& table = async ? ctx->bakedAoPending : ctx->bakedAo;
    if (slots == 0) {
        if (async) { ctx->bakeAsyncPending = false; }
        ctx->bakeValid = true;
        return LV_OK;
    }
    if ((rc = lv_buf_reserve(ctx, counters, sizeof(LvDevCounters)))) return rc;
    if ((rc = lv_buf_reserve(ctx, table, size_t(slots) * 4))) return rc;
    if ((rc = lv_buf_reserve(ctx, gbuf, size_t(slots) * 48))) return rc;
    if ((rc = lv_buf_reserve(ctx, samples, size_t(slots) * spp * 4))) return rc;
    // LCG skip-ahead table: state after j steps = A_j * s + C_j (a = 1664525, c = 1013904223, RayTracingUtilities.glsl:169-175)
    std::vector<uint32_t>& skip = ctx->bakeSkipHost;   // (kept in the context: source of an asynchronous upload)
    if (ctx->bakeAsyncPending) LV_HIP(ctx, hipEventSynchronize(ctx->evBakeDone));   // the previous upload may still read it
    skip.assign(size_t(4) * N * spp, 0u);
    {
        uint32_t A = 1u, C = 0u;
        for (size_t j = 0; j < size_t(2) * N * spp; j++) {
            skip[2 * j] = A; skip[2 * j + 1] = C;
            A = 1664525u * A;
            C = 1664525u * C + 1013904223u;
        }
    }
    if ((rc = lv_buf_reserve(ctx, ctx->bakeLcgSkip, skip.size() * 4))) return rc;
    LvUniforms U;
    lv_fill_uniforms(ctx, U);
    U.aoSamplesPerFrame = spp;
    LvSceneDev SA = sceneDevTriangles(ctx);
    const uint64_t gridRays = lv_ao_grid(ctx, slots * spp);
    if (async) {
        // stack overflow slab of its own (the frames' slab is in use on the other stream)
        SA.stackOverflow = nullptr;
        const uint64_t maxEntries = 3ull * uint64_t(ctx->triWideDepth) + 2;
        if (maxEntries > LV_AO_STACK_LDS) {
            if ((rc = lv_buf_reserve(ctx, ctx->bakeOverflow, size_t(gridRays) * LV_BLOCK * (maxEntries - LV_AO_STACK_LDS) * 4))) return rc;
            SA.stackOverflow = (unsigned*)ctx->bakeOverflow.ptr;
        }
        // everything queued on the context's stream so far (mesh upload, triangle LBVH) comes first
        LV_HIP(ctx, hipEventRecord(ctx->evBakePrereq, ctx->stream));
        LV_HIP(ctx, hipStreamWaitEvent(st, ctx->evBakePrereq, 0));
    } else if ((rc = lv_prepare_overflow(ctx, SA, gridRays, LV_AO_STACK_LDS, true))) {
        return rc;
    }
    LV_HIP(ctx, hipMemcpyAsync(ctx->bakeLcgSkip.ptr, skip.data(), skip.size() * 4, hipMemcpyHostToDevice, st));
    LvDevCounters* dc = (LvDevCounters*)counters.ptr;
    float4* g = (float4*)gbuf.ptr;
    float* smp = (float*)samples.ptr;
    float* out = (float*)table.ptr;
    k_bake_setup<<<nblocks(slots), LV_BLOCK, 0, st>>>((const lv_line_point*)ctx->triPoints.ptr, ctx->numTriPoints,
                                                      (const float*)ctx->bakeSamplingLocations.ptr, M, N,
                                                      o.lineWidth * 0.5f, g, o.useRibbons ? 1u : 0u, o.bandWidth * 0.5f,
                                                      o.minBandThickness);
    ctx->bakeSlotsHost = uint32_t(slots);
    LV_HIP(ctx, hipMemcpyAsync(&dc->aoCount, &ctx->bakeSlotsHost, 4, hipMemcpyHostToDevice, st));
    for (uint32_t iter = 0; iter < o.bakeIterations; iter++) {
        U.aoFrameNumber = iter;
        LV_HIP(ctx, hipMemsetAsync(&dc->aoQueueHead, 0, 8, st));
        if (U.aoUseDistance)
            k_ao_rays<false, false, LV_PRIM_TRIANGLE, true><<<uint32_t(gridRays), LV_AO_BLOCK, 0, st>>>(
                    U, SA, g, smp, dc, nullptr, 0u, LvAoLayout{0u, 0u, 1u, 1u}, (const uint2*)ctx->bakeLcgSkip.ptr);
        else
            k_ao_rays<false, true, LV_PRIM_TRIANGLE, true><<<uint32_t(gridRays), LV_AO_BLOCK, 0, st>>>(
                    U, SA, g, smp, dc, nullptr, 0u, LvAoLayout{0u, 0u, 1u, 1u}, (const uint2*)ctx->bakeLcgSkip.ptr);
        k_ao_reduce<true><<<nblocks(slots), LV_BLOCK, 0, st>>>(U, g, smp, out, out, dc, nullptr, 0u, LvAoLayout{0u, 0u, 1u, 1u});
    }
    LV_HIP(ctx, hipGetLastError());
    if (async) {
        LV_HIP(ctx, hipEventRecord(ctx->evBakeDone, st));
        ctx->bakeAsyncPending = true;
        ctx->bakePendingGeneration = ctx->bakeGeneration;
    } else {
        ctx->bakeValid = true;
    }
    return LV_OK;
}

// Adopts the table of a finished asynchronous bake (wait == true: blocks until it has finished).  A table whose inputs changed while
// it was being computed (lv_invalidate_bake) is dropped.
int lv_bake_poll(lv_ctx* ctx, bool wait) {
    if (!ctx->bakeAsyncPending) return LV_OK;
    if (wait) {
        LV_HIP(ctx, hipEventSynchronize(ctx->evBakeDone));
    } else {
        const hipError_t e = hipEventQuery(ctx->evBakeDone);
        if (e == hipErrorNotReady) return LV_OK;
        if (e != hipSuccess) return lv_fail(ctx, LV_E_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
    }
    ctx->bakeAsyncPending = false;
    if (ctx->bakePendingGeneration == ctx->bakeGeneration) {
        // the context's stream may still shade a frame with the old table: order the swap behind it
        LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::swap(ctx->bakedAo, ctx->bakedAoPending);
        ctx->bakeValid = true;
    }
    return LV_OK;
}
