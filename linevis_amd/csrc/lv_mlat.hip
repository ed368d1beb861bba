// lv_mlat.hip -- multi-layer alpha tracing (MLAT): the ray tracer's single-pass approximate transparency
// (use_mlat / mlat_num_nodes, VulkanRayTracer.cpp:266-275).
//
//   k_render_rt_mlat    TubeRayTracing.RayGen with traceRayMlat (TubeRayTracing.glsl:86-192), the any-hit shader
//                       AnyHitTubeAnalytic = ClosestHitTubeAnalytic + insertNodeMlat (MlatInsert.glsl:35-221) and the MLAT
//                       miss shader (TubeRayTracing.glsl:290-293); payload layout TubeRayTracingHeader.glsl:61-94.
//
// The reference hands every candidate the driver's traversal meets to the any-hit shader, which keeps a list of K nodes
// sorted by depth in the ray payload, merges what falls off the FRONT into the first node, and ends the ray interval
// ("accepts" the hit) behind an opaque fragment or once the accumulated transmittance is below 0.001.  Here the
// all-hits traversal is the wave-cooperative lv_trace_all: (pixel, segment) candidates are tested and shaded 64 at a
// time by whichever lanes are free, the shaded fragments of a batch are chained per pixel in LDS, and every pixel's
// OWN lane then inserts its fragments one after the other into the K nodes it keeps in registers -- so the per-pixel
// insertion is sequential like the reference's any-hit invocations, and an accepted hit shrinks the owner's ray interval
// in LDS where every lane descending for that ray picks it up.
//
// The ORDER in which a pixel's candidates arrive is as undefined here as it is in the reference (there: the driver's
// BVH); once more than K layers exist the result depends on it.  With collect_stats + mlat_record_trace the kernel
// records the order it used (lv_get_mlat_trace) so that a CPU replay can reproduce the frame exactly and validate
// the interval rules.
#include <cmath>
#include <cstring>

#include "lv_internal.h"
#include "lv_trace.h"
#include "lv_tile.h"

namespace {

#define LV_MLAT_NONE 0xFFFFFFFFu

// merge() + insertNodeMlat(), MlatInsert.glsl:35-58, 66-221.  Node i = {c[i][0..3] pre-multiplied colour, T[i]
// transmittance, D[i] depth}; depth 0 marks an empty node, nodes are sorted by ascending depth with the empties first.
// Returns true when the hit is accepted (the ray interval ends at `depth`).
template <int K>
__device__ __forceinline__ bool lv_mlat_insert(float (&c)[K][4], float (&T)[K], float (&D)[K], float& depth2, f4 color,
                                               float depth, bool missShader) {
    const float alpha = color.w;
    if (!missShader && alpha == 0.0f) return false;
    float n0 = alpha * color.x, n1 = alpha * color.y, n2 = alpha * color.z, n3 = color.w;
    float nT = 1.0f - alpha, nD = depth;
#pragma unroll
    for (int i = K - 1; i >= 0; --i) {
        const bool sw = nD > D[i];
        float t;
        t = c[i][0]; c[i][0] = sw ? n0 : t; n0 = sw ? t : n0;
        t = c[i][1]; c[i][1] = sw ? n1 : t; n1 = sw ? t : n1;
        t = c[i][2]; c[i][2] = sw ? n2 : t; n2 = sw ? t : n2;
        t = c[i][3]; c[i][3] = sw ? n3 : t; n3 = sw ? t : n3;
        t = T[i]; T[i] = sw ? nT : t; nT = sw ? t : nT;
        t = D[i]; D[i] = sw ? nD : t; nD = sw ? t : nD;
    }
    if (nD > 0.0f) { // what fell off the front is merged with the first node: a = fallen node, b = node 0
        const bool isFirst = nD == depth;
        const float bT = T[0], bD = D[0];
        float fa = 1.0f, fb = nT;
        depth2 = fmaxf(depth2, bD);
        if (bD < depth2 && !isFirst) {
            float d = (bD - nD);
            d /= (depth2 - nD);
            const float aPowD = lv_pow_det(nT, d);
            fa = (aPowD - 1.0f);
            fa += (nT - aPowD) * bT;
            fa /= (nT - 1.0f);
            fb = aPowD;
        }
        c[0][0] = fa * n0 + fb * c[0][0];
        c[0][1] = fa * n1 + fb * c[0][1];
        c[0][2] = fa * n2 + fb * c[0][2];
        c[0][3] = fa * n3 + fb * c[0][3];
        T[0] = nT * bT;
        D[0] = nD;
    }
    if (alpha == 1.0f) return true;
    float transmittance = 1.0f;
#pragma unroll
    for (int i = 0; i < K; ++i) transmittance *= T[i];
    return transmittance <= 0.001f && D[K - 1] <= depth;
}

template <bool STATS, int K, int PRIM, int BANDS>
__global__ __launch_bounds__(LV_BLOCK) void k_render_rt_mlat(const LvUniforms U, const LvSceneDev S, const LvTiles T,
                                                             uint32_t* __restrict__ out, LvDevCounters* dc,
                                                             uint4* __restrict__ trace, uint32_t traceCap) {
    __shared__ unsigned s_stack[LV_STACK_LDS * LV_BLOCK];
    // shaded fragments of the current batch, per wave: colour (straight alpha), depth, segment, chain link
    __shared__ float s_frag[LV_BLOCK / LV_WAVE][5][LV_WAVE];
    __shared__ unsigned s_fragSeg[LV_BLOCK / LV_WAVE][LV_WAVE];
    __shared__ unsigned s_fragNext[LV_BLOCK / LV_WAVE][LV_WAVE];
    __shared__ unsigned s_chain[LV_BLOCK]; // per pixel (= owner lane): newest fragment of the batch, LV_MLAT_NONE = none
    LV_COOP_SHARED(LV_BLOCK / LV_WAVE);
    LV_COOP_MEM(cm);
    LV_HITQ_SHARED(LV_BLOCK / LV_WAVE);
    LV_HITQ_MEM(hq);
    LvPixel px;
    if (!lv_block_pixel(U, T, px)) return;
    const unsigned long long tg0 = lv_group_clock();
    const LvStackMem sm = lv_stack_mem(s_stack, S.stackOverflow);
    LvCounters cnt = {0, 0, 0, 0};
    const unsigned w = threadIdx.x >> 6, lane = lv_lane();
    const unsigned waveBase = threadIdx.x & ~63u;
    const bool capped = U.useCappedTubes != 0 || U.lssGeometry != 0;
    const float aoTexel = (px.inView && U.useAmbientOcclusion) ? S.ao[size_t(px.y) * U.width + px.x] : 1.0f;
    s_chain[threadIdx.x] = LV_MLAT_NONE;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const uint32_t nSamples = U.useJitteredRays ? U.numSamplesPerFrame : 1u;
    uint32_t seq = 0; // position of the next candidate in this pixel's visiting order (trace only)
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
        // traceRayMlat: clear the payload, TubeRayTracing.glsl:89-136
        float nc[K][4], nT[K], nD[K];
#pragma unroll
        for (int i = 0; i < K; i++) { nc[i][0] = nc[i][1] = nc[i][2] = nc[i][3] = 0.0f; nT[i] = 1.0f; nD[i] = 0.0f; }
        float depth2 = 0.0f;
        bool accepted = false;
        lv_trace_all<STATS, true, PRIM>(S, U.radius, capped, px.inView, o, d, 0.0001f, 1000.0f, aoTexel, 0.0f, sm, cm, hq, cnt,
            // any-hit, first half: shade (ClosestHitTubeAnalytic) on whichever lane holds the candidate
            [&](unsigned owner, uint32_t leaf, float t, int kind, f3 ro, f3 rd, float ownerAo, float) {
                LvHit h; h.t = t; h.leaf = leaf; h.kind = kind; h.found = true;
                float hitT;
                // AnyHitTubeTriangles / AnyHitTubeAnalytic: the closest-hit shading of the geometry mode
                // (band data: AnyHitEllipticTubeAnalytic, EllipticTubeRayTracing.glsl:463-466, or the USE_BANDS variants)
                const f4 color = PRIM == LV_PRIM_TRIANGLE ? lv_shade_hit_triangle<BANDS>(S, U, ownerAo, ro, rd, uint32_t(kind), hitT)
                               : PRIM == LV_PRIM_ELLIPTIC ? lv_shade_hit_elliptic(S, U, ownerAo, ro, rd, h, hitT)
                                                          : lv_shade_hit<BANDS>(S, U, ownerAo, ro, rd, h, hitT);
                if (STATS) cnt.hits++;
                if (color.w == 0.0f) return; // ignoreIntersectionEXT, MlatInsert.glsl:77-79
                s_frag[w][0][lane] = color.x; s_frag[w][1][lane] = color.y; s_frag[w][2][lane] = color.z;
                s_frag[w][3][lane] = color.w; s_frag[w][4][lane] = t; // depth = gl_HitTEXT
                s_fragSeg[w][lane] = PRIM == LV_PRIM_TRIANGLE ? uint32_t(kind) : S.leafSeg[leaf];
                s_fragNext[w][lane] = atomicExch(&s_chain[waveBase + owner], lane);
            },
            // any-hit, second half: every pixel's own lane inserts its fragments of this batch one after the other
            [&](unsigned) {
                unsigned cur = s_chain[threadIdx.x];
                s_chain[threadIdx.x] = LV_MLAT_NONE;
                while (__ballot(cur != LV_MLAT_NONE)) {
                    if (cur != LV_MLAT_NONE) {
                        f4 color;
                        color.x = s_frag[w][0][cur]; color.y = s_frag[w][1][cur]; color.z = s_frag[w][2][cur];
                        color.w = s_frag[w][3][cur];
                        const float depth = s_frag[w][4][cur];
                        // the interval may have shrunk since the candidate was tested: the reference's traversal would
                        // not have reported it any more
                        const bool inside = depth <= cm.ray[2 * lane + 1].w;
                        if (STATS && trace) {
                            const uint32_t slot = atomicAdd(&dc->mlatTraceCount, 1u);
                            if (slot < traceCap)
                                trace[slot] = make_uint4(px.y * U.width + px.x, seq, s_fragSeg[w][cur], inside ? 0u : 1u);
                            seq++;
                        }
                        if (inside && lv_mlat_insert<K>(nc, nT, nD, depth2, color, depth, false)) {
                            accepted = true;
                            cm.ray[2 * lane + 1].w = depth;
                        }
                        cur = s_fragNext[w][cur];
                    }
                }
            });
        if (px.inView) {
            if (!accepted) { // Miss, TubeRayTracing.glsl:290-293
                f4 bg; bg.x = U.background[0]; bg.y = U.background[1]; bg.z = U.background[2]; bg.w = U.background[3];
                lv_mlat_insert<K>(nc, nT, nD, depth2, bg, 1e7f, true);
            }
            // front-to-back blending of the node list (pre-multiplied colours), TubeRayTracing.glsl:141-189
            float fc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < K; i++) {
                fc[0] = fc[0] + (1.0f - fc[3]) * nc[i][0];
                fc[1] = fc[1] + (1.0f - fc[3]) * nc[i][1];
                fc[2] = fc[2] + (1.0f - fc[3]) * nc[i][2];
                fc[3] = fc[3] + (1.0f - fc[3]) * nc[i][3];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] += fc[k];
        }
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

template <int K, int PRIM, int BANDS>
int launchMlat(lv_ctx* ctx, const LvUniforms& U, const LvSceneDev& S, const LvTiles& T, uint32_t gridTiles, uint32_t* out,
               LvDevCounters* dc, uint4* trace, uint32_t traceCap) {
    hipStream_t st = ctx->stream;
    if (ctx->opt.collectStats)
        LV_TIMED_LAUNCH(ctx, LV_KERNEL_RENDER_RT,
                        (k_render_rt_mlat<true, K, PRIM, BANDS><<<gridTiles, LV_BLOCK, 0, st>>>(U, S, T, out, dc, trace, traceCap)));
    else
        LV_TIMED_LAUNCH(ctx, LV_KERNEL_RENDER_RT,
                        (k_render_rt_mlat<false, K, PRIM, BANDS><<<gridTiles, LV_BLOCK, 0, st>>>(U, S, T, out, dc, nullptr, 0u)));
    return LV_OK;
}

template <int PRIM, int BANDS>
int launchMlatK(lv_ctx* ctx, const LvUniforms& U, const LvSceneDev& S, const LvTiles& T, uint32_t gridTiles, uint32_t* out,
                LvDevCounters* dc, uint4* trace, uint32_t traceCap) {
    switch (ctx->opt.mlatNumNodes) {
    case 1: return launchMlat<1, PRIM, BANDS>(ctx, U, S, T, gridTiles, out, dc, trace, traceCap);
    case 2: return launchMlat<2, PRIM, BANDS>(ctx, U, S, T, gridTiles, out, dc, trace, traceCap);
    case 4: return launchMlat<4, PRIM, BANDS>(ctx, U, S, T, gridTiles, out, dc, trace, traceCap);
    case 8: return launchMlat<8, PRIM, BANDS>(ctx, U, S, T, gridTiles, out, dc, trace, traceCap);
    case 16: return launchMlat<16, PRIM, BANDS>(ctx, U, S, T, gridTiles, out, dc, trace, traceCap);
    case 32: return launchMlat<32, PRIM, BANDS>(ctx, U, S, T, gridTiles, out, dc, trace, traceCap);
    default: return lv_fail(ctx, LV_E_INVALID, "mlat_num_nodes must be a power of two in [1, 32], got %u", ctx->opt.mlatNumNodes);
    }
}

} // namespace

// The (K x STATS x primitive x shading variant) instantiations of the kernel are what this file costs to compile (two minutes in
// one piece): the build compiles it four times with -DLV_MLAT_PART=0..3, each part holding a quarter of them; part 0 also holds
// the dispatcher.  Without the define everything lands in one object.
#ifndef LV_MLAT_PART
#define LV_MLAT_PART -1
#endif
#define LV_MLAT_HAS(p) (LV_MLAT_PART == -1 || LV_MLAT_PART == (p))
#define LV_MLAT_ARGS lv_ctx *ctx, const LvUniforms &U, const LvSceneDev &S, const LvTiles &T, uint32_t gridTiles, uint32_t *out, \
                     LvDevCounters *dc, uint4 *trace, uint32_t traceCap
#define LV_MLAT_PASS ctx, U, S, T, gridTiles, out, dc, trace, traceCap
int lv_mlat_launch_capsule_plain(LV_MLAT_ARGS);
int lv_mlat_launch_capsule_shaded(LV_MLAT_ARGS, int shade);
int lv_mlat_launch_triangle_plain_or_elliptic(LV_MLAT_ARGS, bool elliptic);
int lv_mlat_launch_triangle_shaded(LV_MLAT_ARGS, int shade);

#if LV_MLAT_HAS(0)
int lv_mlat_launch_capsule_plain(LV_MLAT_ARGS) { return launchMlatK<LV_PRIM_CAPSULE, LV_SHADE_PLAIN>(LV_MLAT_PASS); }
#endif
#if LV_MLAT_HAS(1)
int lv_mlat_launch_capsule_shaded(LV_MLAT_ARGS, int shade) {
    return shade == LV_SHADE_HELICITY ? launchMlatK<LV_PRIM_CAPSULE, LV_SHADE_HELICITY>(LV_MLAT_PASS)
                                      : launchMlatK<LV_PRIM_CAPSULE, LV_SHADE_BANDS>(LV_MLAT_PASS);
}
#endif
#if LV_MLAT_HAS(2)
int lv_mlat_launch_triangle_plain_or_elliptic(LV_MLAT_ARGS, bool elliptic) {
    return elliptic ? launchMlatK<LV_PRIM_ELLIPTIC, LV_SHADE_BANDS>(LV_MLAT_PASS)
                    : launchMlatK<LV_PRIM_TRIANGLE, LV_SHADE_PLAIN>(LV_MLAT_PASS);
}
#endif
#if LV_MLAT_HAS(3)
int lv_mlat_launch_triangle_shaded(LV_MLAT_ARGS, int shade) {
    return shade == LV_SHADE_HELICITY ? launchMlatK<LV_PRIM_TRIANGLE, LV_SHADE_HELICITY>(LV_MLAT_PASS)
                                      : launchMlatK<LV_PRIM_TRIANGLE, LV_SHADE_BANDS>(LV_MLAT_PASS);
}
#endif

#if LV_MLAT_HAS(0)
int lv_mlat_render(lv_ctx* ctx, const LvUniforms& U, const LvSceneDev& S, const LvTiles& T, uint32_t gridTiles,
                   uint32_t* out, LvDevCounters* dc, bool triangles) {
    uint4* trace = nullptr;
    uint32_t traceCap = 0;
    if (ctx->opt.collectStats && ctx->opt.mlatRecordTrace) {
        int rc = lv_buf_reserve(ctx, ctx->mlatTrace, size_t(ctx->opt.mlatTraceCapacity) * 16);
        if (rc) return rc;
        trace = (uint4*)ctx->mlatTrace.ptr;
        traceCap = ctx->opt.mlatTraceCapacity;
    }
    const int shade = U.useHelicityBands ? LV_SHADE_HELICITY : U.useBands ? LV_SHADE_BANDS : LV_SHADE_PLAIN;
    if (triangles)
        return shade == LV_SHADE_PLAIN ? lv_mlat_launch_triangle_plain_or_elliptic(LV_MLAT_PASS, false)
                                       : lv_mlat_launch_triangle_shaded(LV_MLAT_PASS, shade);
    if (shade == LV_SHADE_BANDS && U.useEllipticTubes) return lv_mlat_launch_triangle_plain_or_elliptic(LV_MLAT_PASS, true);
    return shade == LV_SHADE_PLAIN ? lv_mlat_launch_capsule_plain(LV_MLAT_PASS) : lv_mlat_launch_capsule_shaded(LV_MLAT_PASS, shade);
}
#endif
