// lv_svgf.hip -- SVGF denoiser of the RTAO pass (ambient_occlusion_denoiser = "SVGF").
//
//   k_svgf_reproject       SVGF.Compute-Reproject       Data/Shaders/Denoiser/SVGF.glsl:43-261
//   k_svgf_filter_moments  SVGF.Compute-Filter-Moments  SVGF.glsl:263-352
//   k_svgf_atrous          SVGF.Compute-ATrous          SVGF.glsl:355-495, weights svgf_common.glsl:28-40
//   lv_svgf_denoise        SVGFDenoiser::denoise() + the three passes' _render + the history copies,
//                          src/Renderers/Scattering/Denoiser/SVGF.cpp:107-174,346-425
//
// The denoised image is the AO image: noisy_texture = vec4(ao, ao, ao, 1), the three colour channels stay equal through every
// pass, so colour images are float2 {colour, variance}.  The RTAO pass feeds it per-frame (not accumulated) maps: world-space
// normal + depth in one float4, flow + depth fwidth in another (k_ao_primary, LvSvgfFeat).  "Copy current -> previous" of the
// normal / depth / moments images is a pointer swap.
//
// SVGF is temporal: its history (colour of the first a-trous pass, moments + history length, normal, depth of the previous
// frame) is looked up at REPROJECTED positions, which can be anywhere in the picture, and a pixel's value depends on a
// neighbourhood that grows with every frame.  The whole chain therefore always runs on the full viewport, whatever tile
// the render call asked for (lv_run_ao): tiles of one frame agree by construction because they share this image.
//
// Where the reference leaves the result open, the build defines it:
// texel fetches outside the image return 0; the moments filter reads the image the reprojection pass wrote (the reference
// filters temp_accum in place, a data race between invocations); unwritten `out` parameters are 0; pow(x, 128) is seven
// squarings.
#include <cmath>
#include <cstring>

#include "lv_internal.h"
#include "lv_device.h"

namespace {

inline uint32_t nblocks2(uint32_t n) { return (n + 15u) / 16u; }

__device__ __forceinline__ float svgf_pow128(float x) {
#pragma unroll
    for (int i = 0; i < 7; i++) x = x * x;
    return x;
}
// compute_weight, svgf_common.glsl:28-40; normals in .xyz, depth in .w of the normalDepth texels
__device__ __forceinline__ float svgf_compute_weight(const float4 center, const float4 offset, float phiDepth, float centerColor,
                                                     float offsetColor, float phiColor) {
    const float weightN = svgf_pow128(fmaxf(0.0f, (center.x * offset.x + center.y * offset.y) + center.z * offset.z));
    const float weightZ = (phiDepth == 0.0f) ? 0.0f : fabsf(center.w - offset.w) / phiDepth;
    const float weightC = fabsf(centerColor - offsetColor) * 2.0f / phiColor;
    return expf((0.0f - fmaxf(weightC, 0.0f)) - fmaxf(weightZ, 0.0f)) * weightN;
}

struct LvSvgfImages {
    const float* noisy;               // raw AO of this frame
    const float4* normalDepth;        // this frame
    const float4* flowFwidth;         // this frame
    const float4* normalDepthHistory; // previous frame
    const float4* momentsHistory;     // previous frame {m1, m2, history length, 0}
    const float* colorHistory;        // colour of the previous frame's first a-trous pass
    float4* accumMoments;             // out
    float2* tempAccum;                // out {colour, variance}
};

// is_reprj_valid, SVGF.glsl:72-86 (bounds first: history texels of rejected coordinates are never fetched)
__device__ __forceinline__ bool svgf_reprj_valid(const LvSvgfImages& I, int W, int H, int cx, int cy, const float4 nd,
                                                 float allowedZDist, float allowedNormalDist) {
    if (cx < 1 || cy < 1 || cx > W - 1 || cy > H - 1) return false;
    const float4 h = I.normalDepthHistory[size_t(cy) * W + cx];
    if (fabsf(h.w - nd.w) > allowedZDist) return false;
    const float dx = h.x - nd.x, dy = h.y - nd.y, dz = h.z - nd.z;
    if (sqrtf((dx * dx + dy * dy) + dz * dz) > allowedNormalDist) return false;
    return true;
}

__global__ __launch_bounds__(256) void k_svgf_reproject(const LvSvgfImages I, int W, int H, float allowedZDist,
                                                        float allowedNormalDist) {
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= W || y >= H) return;
    const size_t ci = size_t(y) * W + x;
    const float4 ff = I.flowFwidth[ci];
    float prevM0 = 0.0f, prevM1 = 0.0f, historyLength = 0.0f;
    const int ipx = int((0.5f + float(x)) - ff.x), ipy = int((0.5f + float(y)) - ff.y);
    bool success = !(ipx < 0 || ipy < 0 || ipx >= W || ipy >= H); // load_moments_and_history_length, :186-199
    if (success) {
        const float4 mh = I.momentsHistory[size_t(ipy) * W + ipx];
        prevM0 = mh.x; prevM1 = mh.y; historyLength = mh.z;
    }
    const float color = I.noisy[ci];
    float colorLastFrame = I.colorHistory[ci];
    if (success) {
        const float ppx = (0.01f + float(x)) - ff.x, ppy = (0.01f + float(y)) - ff.y;
        const int qx = int(ppx), qy = int(ppy);
        const float4 nd = I.normalDepth[ci];
        // try_2x2_tap, :88-139: offsets {(0,0), (0,1), (1,0), (1,1)} with the weights in the order the reference lists them
        bool valids[4], validFound = false;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            valids[i] = svgf_reprj_valid(I, W, H, qx + (i >> 1), qy + (i & 1), nd, allowedZDist, allowedNormalDist);
            validFound = validFound || valids[i];
        }
        if (validFound) {
            const float fx = ppx - floorf(ppx), fy = ppy - floorf(ppy);
            const float w[4] = {(1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy};
            float colorBilinear = 0.0f, m0 = 0.0f, m1 = 0.0f, sumW = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (!valids[i]) continue;
                const size_t oi = size_t(qy + (i & 1)) * W + (qx + (i >> 1));
                const float4 mh = I.momentsHistory[oi];
                m0 += w[i] * mh.x;
                m1 += w[i] * mh.y;
                colorBilinear += w[i] * I.colorHistory[oi];
                sumW += w[i];
            }
            validFound = sumW >= 0.001f;
            if (validFound) { colorLastFrame = colorBilinear / sumW; prevM0 = m0 / sumW; prevM1 = m1 / sumW; }
        }
        success = validFound;
        if (!success) {
            // try_3x3_bilat, :141-184
            float nValid = 0.0f, fc = 0.0f, f0 = 0.0f, f1 = 0.0f;
            for (int dy = -1; dy <= 1; dy++) {
                for (int dx = -1; dx <= 1; dx++) {
                    const int ox = qx + dx, oy = qy + dy;
                    if (ox < 1 || oy < 1 || ox >= W || oy >= H) continue;
                    if (svgf_reprj_valid(I, W, H, ox, oy, nd, allowedZDist, allowedNormalDist)) {
                        const size_t oi = size_t(oy) * W + ox;
                        const float4 mh = I.momentsHistory[oi];
                        fc += I.colorHistory[oi];
                        f0 += mh.x;
                        f1 += mh.y;
                        nValid += 1.0f;
                    }
                }
            }
            if (nValid > 0.0f) { colorLastFrame = fc / nValid; prevM0 = f0 / nValid; prevM1 = f1 / nValid; success = true; }
        }
    }
    historyLength = fminf(success ? historyLength + 1.0f : 1.0f, 32.0f);
    const float alphaColor = success ? fmaxf(0.01f, 1.0f / historyLength) : 1.0f;
    const float alphaMoments = success ? fmaxf(0.2f, 1.0f / historyLength) : 1.0f;
    const float r = mixf(prevM0, color, alphaMoments), g = mixf(prevM1, color * color, alphaMoments);
    const float variance = fmaxf(0.0f, g - r * r);
    I.accumMoments[ci] = make_float4(r, g, historyLength, 0.0f);
    I.tempAccum[ci] = make_float2(mixf(colorLastFrame, color, alphaColor), variance);
}

__global__ __launch_bounds__(256) void k_svgf_filter_moments(const float2* __restrict__ src, float2* __restrict__ dst,
                                                             const float4* __restrict__ accumMoments,
                                                             const float4* __restrict__ normalDepth,
                                                             const float4* __restrict__ flowFwidth, int W, int H) {
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= W || y >= H) return;
    const size_t ci = size_t(y) * W + x;
    const float2 center = src[ci];
    const float historyLength = accumMoments[ci].z;
    if (historyLength >= 4.0f) { dst[ci] = center; return; }
    const float4 cnd = normalDepth[ci];
    const float phiDepth = fabsf(flowFwidth[ci].z) + 0.0001f;
    float sumWeight = 0.0f, sumColor = 0.0f, sumM0 = 0.0f, sumM1 = 0.0f;
    for (int dy = -3; dy <= 3; dy++) {
        for (int dx = -3; dx <= 3; dx++) {
            const int ox = x + dx, oy = y + dy;
            if (!(ox >= 0 && oy >= 0 && ox < W && oy < H)) continue;
            const size_t oi = size_t(oy) * W + ox;
            const float oc = src[oi].x;
            const float4 om = accumMoments[oi];
            const float weight = svgf_compute_weight(cnd, normalDepth[oi], phiDepth, center.x, oc, 10.0f);
            sumWeight += weight;
            sumColor += weight * oc;
            sumM0 += weight * om.x;
            sumM1 += weight * om.y;
        }
    }
    sumWeight = fmaxf(sumWeight, 1e-6f);
    sumColor /= sumWeight; sumM0 /= sumWeight; sumM1 /= sumWeight;
    float variance = sumM1 - sumM0 * sumM0;
    variance *= 4.0f / historyLength;
    dst[ci] = make_float2(sumColor, variance);
}

__global__ __launch_bounds__(256) void k_svgf_atrous(const float2* __restrict__ src, float2* __restrict__ dst,
                                                     float* __restrict__ colorHistory, float* __restrict__ result,
                                                     const float4* __restrict__ normalDepth,
                                                     const float4* __restrict__ flowFwidth, int W, int H, int iteration) {
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= W || y >= H) return;
    const int stepWidth = 1 << iteration;
    const size_t ci = size_t(y) * W + x;
    const float2 center = src[ci];
    // filter_variance: 3x3 Gaussian of the variance channel, :368-391
    float fv = 0.0f;
    for (int dy = -1; dy <= 1; dy++) {
        for (int dx = -1; dx <= 1; dx++) {
            const int px = x + dx, py = y + dy;
            const float v = (px >= 0 && py >= 0 && px < W && py < H) ? src[size_t(py) * W + px].y : 0.0f;
            const float k = (dx == 0 ? 0.5f : 0.25f) * (dy == 0 ? 0.5f : 0.25f); // {{1/4, 1/8}, {1/8, 1/16}}
            fv += v * k;
        }
    }
    const float4 cnd = normalDepth[ci];
    const float centerFwidth = flowFwidth[ci].z;
    const float phiColor = sqrtf(fmaxf(0.0f, 1e-10f + fv));
    const float kv[3] = {1.0f, 2.0f / 3.0f, 1.0f / 6.0f};
    float accumW = kv[0] * kv[0];
    float sumC = center.x * accumW, sumV = center.y * accumW;
    for (int dy = -2; dy <= 2; ++dy) {
        for (int dx = -2; dx <= 2; ++dx) {
            const int ox = x + dx * stepWidth, oy = y + dy * stepWidth;
            const bool inside = ox >= 0 && oy >= 0 && ox < W && oy < H;
            if (!inside || (dx == 0 && dy == 0)) continue;
            const size_t oi = size_t(oy) * W + ox;
            const float kernelValue = kv[dx < 0 ? -dx : dx] * kv[dy < 0 ? -dy : dy];
            const float len = sqrtf(float(dx) * float(dx) + float(dy) * float(dy));
            const float2 oc = src[oi];
            const float weight = svgf_compute_weight(cnd, normalDepth[oi], fabsf((centerFwidth * len) * float(stepWidth)) + 0.0001f,
                                                     center.x, oc.x, phiColor) * kernelValue;
            sumC += weight * oc.x;
            sumV += (weight * weight) * oc.y; // "variance gets squared weight"
            accumW += weight;
        }
    }
    const float2 r = make_float2(sumC / accumW, sumV / (accumW * accumW));
    dst[ci] = r;
    if (iteration == 0) colorHistory[ci] = r.x;
    if (result) result[ci] = r.x;
}

__global__ __launch_bounds__(256) void k_svgf_blit(const float2* __restrict__ src, float* __restrict__ colorHistory,
                                                   float* __restrict__ result, size_t n) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const float c = src[i].x;
    colorHistory[i] = c;
    result[i] = c;
}

} // namespace

// (re)allocate and clear the temporal state: SVGFDenoiser::recreateSwapchain, SVGF.cpp:181-286
int lv_svgf_prepare(lv_ctx* ctx) {
    LvSvgfState& V = ctx->svgf;
    const size_t n = size_t(ctx->width) * ctx->height;
    int rc;
    for (LvDeviceBuffer* b : {&V.normalDepth, &V.normalDepthHistory, &V.flowFwidth, &V.moments, &V.momentsHistory})
        if ((rc = lv_buf_reserve(ctx, *b, n * 16))) return rc;
    for (LvDeviceBuffer* b : {&V.tempAccum, &V.tempAccumFiltered, &V.ping, &V.pong})
        if ((rc = lv_buf_reserve(ctx, *b, n * 8))) return rc;
    for (LvDeviceBuffer* b : {&V.colorHistory, &V.result})
        if ((rc = lv_buf_reserve(ctx, *b, n * 4))) return rc;
    if (!V.historyValid || V.width != ctx->width || V.height != ctx->height) {
        hipStream_t st = ctx->stream;
        LV_HIP(ctx, hipMemsetAsync(V.normalDepthHistory.ptr, 0, n * 16, st));
        LV_HIP(ctx, hipMemsetAsync(V.momentsHistory.ptr, 0, n * 16, st));
        LV_HIP(ctx, hipMemsetAsync(V.colorHistory.ptr, 0, n * 4, st));
        V.width = ctx->width;
        V.height = ctx->height;
        V.historyValid = true;
    }
    return LV_OK;
}

// One SVGFDenoiser::denoise() on the raw AO image `noisy` (full viewport); the result is ctx->svgf.result.
int lv_svgf_denoise(lv_ctx* ctx, const float* noisy) {
    LvSvgfState& V = ctx->svgf;
    hipStream_t st = ctx->stream;
    const int W = int(ctx->width), H = int(ctx->height);
    const size_t n = size_t(W) * H;
    const dim3 grid(nblocks2(ctx->width), nblocks2(ctx->height));
    LvSvgfImages I;
    I.noisy = noisy;
    I.normalDepth = (const float4*)V.normalDepth.ptr;
    I.flowFwidth = (const float4*)V.flowFwidth.ptr;
    I.normalDepthHistory = (const float4*)V.normalDepthHistory.ptr;
    I.momentsHistory = (const float4*)V.momentsHistory.ptr;
    I.colorHistory = (const float*)V.colorHistory.ptr;
    I.accumMoments = (float4*)V.moments.ptr;
    I.tempAccum = (float2*)V.tempAccum.ptr;
    k_svgf_reproject<<<grid, 256, 0, st>>>(I, W, H, ctx->opt.svgfAllowedZDist, ctx->opt.svgfAllowedNormalDist);
    k_svgf_filter_moments<<<grid, 256, 0, st>>>((const float2*)V.tempAccum.ptr, (float2*)V.tempAccumFiltered.ptr,
                                                 (const float4*)V.moments.ptr, I.normalDepth, I.flowFwidth, W, H);
    const int its = int(ctx->opt.svgfIterations);
    if (its < 1) {
        // maxNumIterations < 1: temp_accum is blitted to the output and to the colour history, SVGF.cpp:347-360
        k_svgf_blit<<<uint32_t((n + 255) / 256), 256, 0, st>>>((const float2*)V.tempAccumFiltered.ptr, (float*)V.colorHistory.ptr,
                                                               (float*)V.result.ptr, n);
    }
    const float2* src = (const float2*)V.tempAccumFiltered.ptr;
    float2* bufs[2] = {(float2*)V.ping.ptr, (float2*)V.pong.ptr};
    for (int i = 0; i < its; i++) {
        float2* dst = bufs[i & 1];
        k_svgf_atrous<<<grid, 256, 0, st>>>(src, dst, (float*)V.colorHistory.ptr, i == its - 1 ? (float*)V.result.ptr : nullptr,
                                            I.normalDepth, I.flowFwidth, W, H, i);
        src = dst;
    }
    // "update previous frame images", SVGF.cpp:112-173: the copies are pointer swaps
    std::swap(V.normalDepth, V.normalDepthHistory);
    std::swap(V.moments, V.momentsHistory);
    LV_HIP(ctx, hipGetLastError());
    return LV_OK;
}
