/*
 * lv_oracle_common.h -- internal helpers shared by the CPU ORACLE translation units (vector maths with a fixed
 * evaluation order, RNG, camera frame, slab test).  TEST INFRASTRUCTURE, NOT PRODUCT (see lv_oracle.h).
 */
#pragma once
#include "lv_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

struct LvoAoFeatureSink {
    float* normal; float* position;        // EAW: view-space normal / position, running means over the iterations
    // SVGF (DISABLE_ACCUMULATION, useGlobalFrameNumber; SVGF.hpp:81-82): per-frame maps of the LAST RTAO iteration only
    float* normalWorld;                    // float4 {surfaceNormal, 0}
    float* depth;                          // float  -positionViewSpace.z, farDistance on a miss
    float* flow;                           // float2 writePos - pixel position under the last frame's view-projection
    float* depthFwidth;                    // float  |nabla.x| + |nabla.y|
    int svgf;                              // != 0: no running means (AO image included), seeds from globalFrameNumber
    uint32_t globalFrameNumber;
    float lastFrameViewProj[16];
};
extern LvoAoFeatureSink g_lvoAoFeatures;   // defined in lv_oracle.cpp; set by lvo_set_ao_feature_outputs

namespace {

// ---------------------------------------------------------------- vector helpers (fixed evaluation order)
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// GLSL cross(x, y) = (x1*y2 - y1*x2, x2*y0 - y2*x0, x0*y1 - y0*x1)
inline V3 cross(V3 a, V3 b) { return V3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline float length(V3 a) { return sqrtf(dot(a, a)); }
inline V3 normalize(V3 a) { float l = length(a); return V3{a.x / l, a.y / l, a.z / l}; }
inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
// pow(x, y) of the shading code as a build-owned float32 definition, bit-identical to lv_pow_det of the HIP library
// (linevis_amd/csrc/lv_device.h): exp2(y * log2(x)), log2 through the exponent bits + an atanh series, exp2 through a 7th-order series.
inline float powDet(float x, float y) {
    if (!(x > 1.17549435e-38f)) return y > 0.0f ? 0.0f : (y == 0.0f ? 1.0f : INFINITY);
    uint32_t bits; memcpy(&bits, &x, 4);
    int e = int((bits >> 23) & 0xFFu) - 127;
    uint32_t mb = (bits & 0x007FFFFFu) | 0x3F800000u;
    float m; memcpy(&m, &mb, 4);
    if (m > 1.41421356f) { m = m * 0.5f; e = e + 1; }
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float P = 0.333333333f + z * (0.2f + z * (0.142857143f + z * 0.111111111f));
    const float ln = 2.0f * s + (2.0f * s) * (z * P);
    const float L = float(e) + ln * 1.44269504f;
    const float p = y * L;
    if (p < -125.0f) return 0.0f;
    if (p > 127.0f) return INFINITY;
    const float n = floorf(p + 0.5f);
    const float t = (p - n) * 0.693147181f;
    const float Q = 1.0f + t * (1.0f + t * (0.5f + t * (0.166666667f + t * (0.0416666667f + t * (0.00833333333f + t * (0.00138888889f + t * 0.000198412698f))))));
    uint32_t qb; memcpy(&qb, &Q, 4);
    qb += uint32_t(int(n)) << 23;
    float r; memcpy(&r, &qb, 4);
    return r;
}
// log2(x), x > 0 and normal: the first half of powDet (lv_log2_det of the HIP library)
inline float log2Det(float x) {
    uint32_t bits; memcpy(&bits, &x, 4);
    int e = int((bits >> 23) & 0xFFu) - 127;
    uint32_t mb = (bits & 0x007FFFFFu) | 0x3F800000u;
    float m; memcpy(&m, &mb, 4);
    if (m > 1.41421356f) { m = m * 0.5f; e = e + 1; }
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float P = 0.333333333f + z * (0.2f + z * (0.142857143f + z * 0.111111111f));
    const float ln = 2.0f * s + (2.0f * s) * (z * P);
    return float(e) + ln * 1.44269504f;
}
// normalize(v) of the shading code as v * (1 / length(v)) (norm3s of the HIP library)
// -- with the squared length clamped into [2^-60, 2^60] (every length the shading code meets lies inside; a zero vector stays a zero
// vector; GLSL leaves normalize() of such vectors undefined).  The build owns this rule.
// Where the clamp acts, this is NOT the reference's normalize() (GLSL: NaN / Inf there): every call whose squared length falls outside
// the range is counted (lvo_shade_normalize_out_of_range), and the parity tests assert that the scenes they compare never get there --
// so "device == checker" on those scenes is also "device == the reference's normalize()".
inline std::atomic<unsigned long long> g_shadeNormalizeOutOfRange{0ull};
inline V3 normalizeShade(V3 a) {
    const float x0 = dot(a, a);
    if (!(x0 >= 0x1p-60f && x0 <= 0x1p60f)) g_shadeNormalizeOutOfRange.fetch_add(1ull, std::memory_order_relaxed);
    const float x = fminf(fmaxf(x0, 0x1p-60f), 0x1p60f);   // (NaN -> 2^-60: fmaxf / fminf return the other operand)
    const float r = 1.0f / sqrtf(x);
    return V3{a.x * r, a.y * r, a.z * r};
}

inline float mixf(float a, float b, float w) { return a * (1.0f - w) + b * w; }
inline float smoothstepf(float e0, float e1, float x) {
    float t = clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
inline V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }

// column-major mat4 * vec4, sum over columns left to right
inline V4 mulM4(const float* m, V4 v) {
    V4 r;
    r.x = ((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * v.w;
    r.y = ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * v.w;
    r.z = ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * v.w;
    r.w = ((m[3] * v.x + m[7] * v.y) + m[11] * v.z) + m[15] * v.w;
    return r;
}

// ---------------------------------------------------------------- RNG, RayTracingUtilities.glsl:134-181
inline uint32_t tea(uint32_t val0, uint32_t val1) {
    uint32_t v0 = val0, v1 = val1, s0 = 0;
    for (uint32_t n = 0; n < 16; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
inline uint32_t lcg(uint32_t& prev) {
    prev = 1664525u * prev + 1013904223u;
    return prev & 0x00FFFFFFu;
}
inline float rnd(uint32_t& seed) { return float(lcg(seed)) / float(0x01000000); }

// sin/cos(2*pi*xi), xi in [0,1).  GLSL sin/cos precision is implementation defined (Vulkan allows
// 2^-11 abs error); the build defines them by this fixed polynomial so that AO ray directions are
// bit-identical on host and device.  Quadrant reduction on xi is exact (power-of-two scaling).
inline void sincos2pi(float xi, float& s, float& c) {
    float q = xi * 4.0f;
    float fq = floorf(q);
    int quad = int(fq) & 3;
    float r = q - fq;                 // [0,1) exact
    bool swap = r > 0.5f;
    float rr = swap ? (1.0f - r) : r; // [0,0.5] exact
    float a = rr * 1.57079632679489662f;
    float a2 = a * a;
    float sp = a * (1.0f + a2 * (-1.0f / 6.0f + a2 * (1.0f / 120.0f + a2 * (-1.0f / 5040.0f + a2 * (1.0f / 362880.0f)))));
    float cp = 1.0f + a2 * (-0.5f + a2 * (1.0f / 24.0f + a2 * (-1.0f / 720.0f + a2 * (1.0f / 40320.0f + a2 * (-1.0f / 3628800.0f)))));
    float sa = swap ? cp : sp;        // sin/cos of r*pi/2
    float ca = swap ? sp : cp;
    switch (quad) {
        case 0: s = sa; c = ca; break;
        case 1: s = ca; c = -sa; break;
        case 2: s = -sa; c = -ca; break;
        default: s = -ca; c = sa; break;
    }
}

// sin / cos / atan2 of the elliptic-tube shaders (EllipticTubeRayTracing.glsl).  GLSL leaves their precision to the
// implementation; the sphere tracing loop takes discrete decisions on their results (termination, clipping planes), so the build
// defines them by fixed float32 formulas that evaluate identically on host and device: sin / cos through sincos2pi after an
// explicit reduction of the angle to a fraction of the full turn, atan through an odd polynomial on [-tan(pi/8), tan(pi/8)].
inline void sincosRad(float a, float& s, float& c) {
    float u = a * 0.15915494309189535f;   // a / (2 pi)
    u = u - floorf(u);
    if (!(u < 1.0f)) u = 0.0f;            // -tiny - floor(-tiny) rounds to 1
    sincos2pi(u, s, c);
}
inline float atan2Det(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float r = 0.0f;
    if (mx > 0.0f) {
        float a = mn / mx;                                   // [0, 1]
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

struct Counters { uint64_t rays = 0, nodes = 0, prims = 0, hits = 0; };

// Work decomposition of the frame loops: 16x16-pixel tiles handed out one at a time (schedule(dynamic, 1)), as
// BASELINE.md section 2 specifies for the CPU baseline.  Rows in chunks of 4 gave 270 chunks for 256 threads with the work
// concentrated in the middle rows, i.e. a few dozen busy threads.  Pixels are independent, so the decomposition does not
// change any result.
static inline int64_t lvoTileCount(uint32_t w, uint32_t h) { return int64_t((w + 15u) / 16u) * int64_t((h + 15u) / 16u); }
static inline bool lvoTilePixel(uint32_t w, uint32_t h, int64_t tile, uint32_t k, uint32_t& xx, uint32_t& yy) {
    const uint32_t tx = (w + 15u) / 16u;
    xx = uint32_t(tile % tx) * 16u + (k & 15u);
    yy = uint32_t(tile / tx) * 16u + (k >> 4);
    return xx < w && yy < h;
}

inline uint64_t expandBits21(uint64_t v) {
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

// conservative slab test (boxes are padded at build time); tNear = entry parameter
inline bool rayBox(const float* bmin, const float* bmax, V3 o, V3 inv, float tMin, float tMax, float& tNear) {
    float tx0 = (bmin[0] - o.x) * inv.x, tx1 = (bmax[0] - o.x) * inv.x;
    float ty0 = (bmin[1] - o.y) * inv.y, ty1 = (bmax[1] - o.y) * inv.y;
    float tz0 = (bmin[2] - o.z) * inv.z, tz1 = (bmax[2] - o.z) * inv.z;
    float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tMin));
    float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), tMax));
    tNear = tn;
    return tn <= tf * 1.0000005f + 1e-7f;
}

// ---------------------------------------------------------------- per-frame constants
struct Frame {
    float invView[16], invProj[16];
    V3 cameraPosition;
    float foreground[4];
    float radius;
    float subdivisionCorrectionFactor;
};

inline void mat4Inverse(const float* m, float* inv) {
    // cofactor expansion (same scheme as glm::inverse: 2x2 sub-determinants, adjugate, 1/det)
    float c00 = m[10] * m[15] - m[14] * m[11];
    float c02 = m[6] * m[15] - m[14] * m[7];
    float c03 = m[6] * m[11] - m[10] * m[7];
    float c04 = m[9] * m[15] - m[13] * m[11];
    float c06 = m[5] * m[15] - m[13] * m[7];
    float c07 = m[5] * m[11] - m[9] * m[7];
    float c08 = m[9] * m[14] - m[13] * m[10];
    float c10 = m[5] * m[14] - m[13] * m[6];
    float c11 = m[5] * m[10] - m[9] * m[6];
    float c12 = m[8] * m[15] - m[12] * m[11];
    float c14 = m[4] * m[15] - m[12] * m[7];
    float c15 = m[4] * m[11] - m[8] * m[7];
    float c16 = m[8] * m[14] - m[12] * m[10];
    float c18 = m[4] * m[14] - m[12] * m[6];
    float c19 = m[4] * m[10] - m[8] * m[6];
    float c20 = m[8] * m[13] - m[12] * m[9];
    float c22 = m[4] * m[13] - m[12] * m[5];
    float c23 = m[4] * m[9] - m[8] * m[5];

    float i00 = +((m[5] * c00 - m[6] * c04) + m[7] * c08);
    float i01 = -((m[1] * c00 - m[2] * c04) + m[3] * c08);
    float i02 = +((m[1] * c02 - m[2] * c06) + m[3] * c10);
    float i03 = -((m[1] * c03 - m[2] * c07) + m[3] * c11);
    float i10 = -((m[4] * c00 - m[6] * c12) + m[7] * c16);
    float i11 = +((m[0] * c00 - m[2] * c12) + m[3] * c16);
    float i12 = -((m[0] * c02 - m[2] * c14) + m[3] * c18);
    float i13 = +((m[0] * c03 - m[2] * c15) + m[3] * c19);
    float i20 = +((m[4] * c04 - m[5] * c12) + m[7] * c20);
    float i21 = -((m[0] * c04 - m[1] * c12) + m[3] * c20);
    float i22 = +((m[0] * c06 - m[1] * c14) + m[3] * c22);
    float i23 = -((m[0] * c07 - m[1] * c15) + m[3] * c23);
    float i30 = -((m[4] * c08 - m[5] * c16) + m[6] * c20);
    float i31 = +((m[0] * c08 - m[1] * c16) + m[2] * c20);
    float i32 = -((m[0] * c10 - m[1] * c18) + m[2] * c22);
    float i33 = +((m[0] * c11 - m[1] * c19) + m[2] * c23);

    float det = ((m[0] * i00 + m[1] * i10) + m[2] * i20) + m[3] * i30;
    float r = 1.0f / det;
    inv[0] = i00 * r;  inv[1] = i01 * r;  inv[2] = i02 * r;  inv[3] = i03 * r;
    inv[4] = i10 * r;  inv[5] = i11 * r;  inv[6] = i12 * r;  inv[7] = i13 * r;
    inv[8] = i20 * r;  inv[9] = i21 * r;  inv[10] = i22 * r; inv[11] = i23 * r;
    inv[12] = i30 * r; inv[13] = i31 * r; inv[14] = i32 * r; inv[15] = i33 * r;
}

inline Frame makeFrame(const lvo_params& P) {
    Frame f;
    // LineData.cpp:1290-1291
    mat4Inverse(P.view, f.invView);
    mat4Inverse(P.proj, f.invProj);
    // rayOrigin = (inverseViewMatrix * vec4(0,0,0,1)).xyz, TubeRayTracing.glsl:202; used as cameraPosition too
    V4 o = mulM4(f.invView, V4{0.0f, 0.0f, 0.0f, 1.0f});
    f.cameraPosition = v3(o.x, o.y, o.z);
    // LineData.cpp:1282-1283
    for (int i = 0; i < 4; i++) f.foreground[i] = 1.0f - P.background[i];
    f.radius = P.lineWidth * 0.5f; // TubeRayTracing.glsl:453
    // VulkanRayTracedAmbientOcclusion.cpp:588
    f.subdivisionCorrectionFactor = cosf(3.1415926535897932f / float(P.tubeNumSubdivisions));
    return f;
}

// Denoiser feature maps of the RTAO pass (VulkanRayTracedAmbientOcclusion.glsl:321-399, WRITE_NORMAL_MAP / WRITE_POSITION_MAP
// with accumulation): view-space normal {xyz, 0} and view-space position {xyz, 1} of the primary hit (misses: surfaceNormal =
// vertexPositionWorld = 0, glsl:211-212), running means over the iterations.  Full-viewport float4 images set through
// lvo_set_ao_feature_outputs; null = not written.
// (the sink itself has external linkage -- this header lives in an anonymous namespace per translation unit)

inline void writeAoFeatures(const lvo_params& P, const Frame& F, uint32_t x, uint32_t y, size_t idx, uint32_t frameNumber,
                            bool hasHitSurface, V3 surfaceNormal, V3 vertexPositionWorld) {
    // camNormal = (inverseTransposedViewMatrix * vec4(surfaceNormal, 0)).xyz, inverseTransposedViewMatrix =
    // transpose(inverseViewMatrix) (VulkanRayTracedAmbientOcclusion.cpp:569): row i of the product uses COLUMN i of invView
    const float* m = F.invView;
    const V3 camNormalHit = v3(((m[0] * surfaceNormal.x + m[1] * surfaceNormal.y) + m[2] * surfaceNormal.z) + m[3] * 0.0f,
                               ((m[4] * surfaceNormal.x + m[5] * surfaceNormal.y) + m[6] * surfaceNormal.z) + m[7] * 0.0f,
                               ((m[8] * surfaceNormal.x + m[9] * surfaceNormal.y) + m[10] * surfaceNormal.z) + m[11] * 0.0f);
    if (g_lvoAoFeatures.normal) {
        V3 n = camNormalHit;
        float* o = g_lvoAoFeatures.normal + 4 * idx;
        if (frameNumber != 0) {
            const float a = 1.0f / float(frameNumber + 1);
            n = v3(mixf(o[0], n.x, a), mixf(o[1], n.y, a), mixf(o[2], n.z, a));
            const float len = length(n);
            if (len > 1e-5f) n = v3(n.x / len, n.y / len, n.z / len);
        }
        o[0] = n.x; o[1] = n.y; o[2] = n.z; o[3] = 0.0f;
    }
    const V4 pv = mulM4(P.view, V4{vertexPositionWorld.x, vertexPositionWorld.y, vertexPositionWorld.z, 1.0f});
    if (g_lvoAoFeatures.position) {
        V3 q = v3(pv.x, pv.y, pv.z);
        float* o = g_lvoAoFeatures.position + 4 * idx;
        if (frameNumber != 0) {
            const float a = 1.0f / float(frameNumber + 1);
            q = v3(mixf(o[0], q.x, a), mixf(o[1], q.y, a), mixf(o[2], q.z, a));
        }
        o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = 1.0f;
    }
    // ---- the maps SVGF asks for (SVGF.cpp:88-96), DISABLE_ACCUMULATION branches of glsl:321-464
    if (g_lvoAoFeatures.normalWorld) {
        float* o = g_lvoAoFeatures.normalWorld + 4 * idx;
        o[0] = surfaceNormal.x; o[1] = surfaceNormal.y; o[2] = surfaceNormal.z; o[3] = 0.0f;
    }
    if (g_lvoAoFeatures.depth) g_lvoAoFeatures.depth[idx] = hasHitSurface ? -pv.z : P.farDist;
    if (g_lvoAoFeatures.flow) {
        float fx = 0.0f, fy = 0.0f;
        if (hasHitSurface) {
            V4 ndc = mulM4(g_lvoAoFeatures.lastFrameViewProj, V4{vertexPositionWorld.x, vertexPositionWorld.y, vertexPositionWorld.z, 1.0f});
            ndc.x /= ndc.w; ndc.y /= ndc.w;
            const float plx = (0.5f * ndc.x + 0.5f) * float(P.width) - 0.5f;
            const float ply = (0.5f * ndc.y + 0.5f) * float(P.height) - 0.5f;
            fx = float(x) - plx;
            fy = float(y) - ply;
        }
        g_lvoAoFeatures.flow[2 * idx] = fx;
        g_lvoAoFeatures.flow[2 * idx + 1] = fy;
    }
    if (g_lvoAoFeatures.depthFwidth) {
        // cot of the angle between the view-space normal and the camera x / y axis, glsl:435-442
        float nx = 0.0f, ny = 0.0f;
        if (hasHitSurface) {
            const float A = camNormalHit.x, B = camNormalHit.y;
            nx = A / sqrtf(1.0f - A * A);
            ny = B / sqrtf(1.0f - B * B);
        }
        g_lvoAoFeatures.depthFwidth[idx] = fabsf(nx) + fabsf(ny);
    }
}

// primary ray for launch id (x,y) with sub-pixel offset xi; TubeRayTracing.glsl:219-226
inline void primaryRay(const lvo_params& P, const Frame& F, uint32_t x, uint32_t y, float xix, float xiy, V3& o, V3& d) {
    float ndcx = 2.0f * ((float(x) + xix) / float(P.width)) - 1.0f;
    float ndcy = 2.0f * ((float(y) + xiy) / float(P.height)) - 1.0f;
    V4 target = mulM4(F.invProj, V4{ndcx, ndcy, 1.0f, 1.0f});
    V3 tn = normalize(v3(target.x, target.y, target.z));
    V4 dir = mulM4(F.invView, V4{tn.x, tn.y, tn.z, 0.0f});
    o = F.cameraPosition;
    d = v3(dir.x, dir.y, dir.z);
}

} // namespace
