/*
 * lv_oracle.cpp -- CPU ORACLE: restatement of the LineVis GLSL hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see lv_oracle.h).  PARITY UNPINNED: the
 * reference cannot be built in this image and holds no golden vectors for this
 * path; this file follows the cited GLSL / C++ lines literally and its outputs
 * are committed under tests/golden/ as regression pins.
 *
 * All paths are relative to /root/reference.  Arithmetic is float32 with a
 * fixed evaluation order (compile with -ffp-contract=off): +,-,*,/,sqrt are
 * IEEE-exact on host and device, so hits (t, segment, kind) are bit-comparable
 * with the HIP path; only pow() in shading goes through libm.
 *
 * Definitions owned by the build because they live in un-vendored sgl / the
 * Vulkan driver: camera conventions (right-handed view,
 * depth 0..1, row 0 = top via a y-flipped projection), transfer-function
 * texture (N texels, linear filter, texel centres (i+0.5)/N, clamp-to-edge),
 * AO texture lookup (nearest texel), RGBA8 rounding floor(x*255+0.5), the BVH
 * (ties -> lowest segment index), AO geometry (analytic capsules instead of
 * the 6-gon triangle tubes), PPLL fragment source (entry hits of pixel-centre
 * rays against capsules) and PPLL tie order ((depth, colour) key).
 */
#include "lv_oracle_common.h"
#include "lv_oracle_tri.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

LvoAoFeatureSink g_lvoAoFeatures = {};

namespace {

// Deviation switches (lvo_set_deviation_switches): evaluate the REFERENCE's literal definitions where the build owns a
// different one, so that tests can measure what each documented deviation does to whole frames.  Process-global, set
// between (never during) render calls.
//   literalIntersection  ray-capsule roots in the textbook form of RayIntersectionTestsVulkan.glsl:39-119 instead of the
//                        closest-approach form
//   referenceAoLookup    getAoFactor as AmbientOcclusion.glsl:84-99: project the hit with projectionMatrix and sample the
//                        AO texture bilinearly (clamp to edge) instead of reading the launching pixel's texel
struct DeviationSwitches { bool literalIntersection = false, referenceAoLookup = false; const float* aoImage = nullptr; };
DeviationSwitches g_dev;
// deviation switch (lvo_set_ppll_fragment_colour_variant): 1 = shade the PPLL fragments with the RAY TRACER's computeFragmentColor
// (RayHitCommon.glsl, what rounds 1-2 did), 0 = with the raster tube shader's variant (the reference, default)
bool g_rtFragmentColourInPpll = false;

// ---------------------------------------------------------------- intersection tests
// The reference solves both quadratics in the textbook form t = (-B -+ sqrt(B^2 - 4AC)) / 2A
// (RayIntersectionTestsVulkan.glsl:39-72 and :78-119).  For a tube of radius 1e-3 seen from distance ~1 the
// discriminant B^2 - 4AC is a difference of two O(1) numbers whose result is O(r^2) = 1e-6, so float32 leaves it
// 1-2 significant digits and the computed t jitters by up to ~25 % of the radius; GLSL compilers are moreover free
// to contract these expressions, so the reference's bit-level behaviour is not defined.  The build evaluates THE SAME
// quadratics in the closest-approach form (Haines et al., "Precision Improvements for Ray/Sphere Intersection",
// Ray Tracing Gems ch. 7): t_c = closest-approach parameter, l = perpendicular offset at t_c, half chord
// h = sqrt((r^2 - l.l) / A), roots t_c -+ h.  Same roots, same root selection and end-plane logic as the reference;
// error ~1e-7 instead of ~1e-4, which is what makes BVH culling and brute force agree bit for bit.  The literal
// forms are kept below (…Literal) and tests check both agree within the literal form's error bound.

// RayIntersectionTestsVulkan.glsl:39-72, literal
inline bool raySphereIntersectionLiteral(V3 o, V3 d, V3 ctr, float radius, float& hitT) {
    float A = (d.x * d.x + d.y * d.y) + d.z * d.z;
    float B = 2.0f * ((d.x * (o.x - ctr.x) + d.y * (o.y - ctr.y)) + d.z * (o.z - ctr.z));
    float C = (((o.x - ctr.x) * (o.x - ctr.x) + (o.y - ctr.y) * (o.y - ctr.y)) + (o.z - ctr.z) * (o.z - ctr.z))
              - radius * radius;
    float discriminant = B * B - (4.0f * A) * C;
    if (discriminant < 0.0f) return false;
    float ds = sqrtf(discriminant);
    float t0 = (-B - ds) / (2.0f * A);
    float t1 = (-B + ds) / (2.0f * A);
    hitT = t0;
    if (t0 >= 0.0f) hitT = t0;
    else if (t1 >= 0.0f) hitT = t1;
    else return false;
    return true;
}

// RayIntersectionTestsVulkan.glsl:78-119, literal
inline bool rayTubeIntersectionLiteral(V3 o, V3 d, V3 tubeStart, V3 tubeEnd, float radius, float& hitT) {
    V3 tubeDirection = normalize(tubeEnd - tubeStart);
    V3 deltaP = o - tubeStart;
    V3 av = d - dot(d, tubeDirection) * tubeDirection;
    V3 cv = deltaP - dot(deltaP, tubeDirection) * tubeDirection;
    float A = (av.x * av.x + av.y * av.y) + av.z * av.z;
    float B = 2.0f * dot(av, cv);
    float C = ((cv.x * cv.x + cv.y * cv.y) + cv.z * cv.z) - radius * radius;
    float discriminant = B * B - (4.0f * A) * C;
    if (discriminant < 0.0f) return false;
    float ds = sqrtf(discriminant);
    float t0 = (-B - ds) / (2.0f * A);
    if (t0 >= 0.0f) {
        V3 ip = o + t0 * d;
        if (dot(tubeDirection, ip - tubeStart) > 0.0f && dot(tubeDirection, ip - tubeEnd) < 0.0f) { hitT = t0; return true; }
    }
    float t1 = (-B + ds) / (2.0f * A);
    if (t1 >= 0.0f) {
        V3 ip = o + t1 * d;
        if (dot(tubeDirection, ip - tubeStart) > 0.0f && dot(tubeDirection, ip - tubeEnd) < 0.0f) { hitT = t1; return true; }
    }
    return false;
}

// raySphereIntersection (RayIntersectionTestsVulkan.glsl:39-72) in closest-approach form
inline bool raySphereIntersection(V3 o, V3 d, V3 ctr, float radius, float& hitT) {
    V3 f = o - ctr;
    float A = dot(d, d);
    float tc = -dot(f, d) / A;
    V3 l = f + tc * d;
    float discriminant = radius * radius - dot(l, l);
    if (discriminant < 0.0f) return false;
    float h = sqrtf(discriminant / A);
    float t0 = tc - h;
    float t1 = tc + h;
    hitT = t0;
    if (t0 >= 0.0f) hitT = t0;
    else if (t1 >= 0.0f) hitT = t1;
    else return false;
    return true;
}

// rayTubeIntersection (RayIntersectionTestsVulkan.glsl:78-119) in closest-approach form
inline bool rayTubeIntersection(V3 o, V3 d, V3 tubeStart, V3 tubeEnd, float radius, float& hitT) {
    V3 tubeDirection = normalize(tubeEnd - tubeStart);
    V3 deltaP = o - tubeStart;
    V3 av = d - dot(d, tubeDirection) * tubeDirection;
    V3 cv = deltaP - dot(deltaP, tubeDirection) * tubeDirection;
    float A = dot(av, av);
    float tc = -dot(av, cv) / A;
    V3 l = cv + tc * av;
    float discriminant = radius * radius - dot(l, l);
    if (discriminant < 0.0f) return false;
    float h = sqrtf(discriminant / A);
    float t0 = tc - h;
    if (t0 >= 0.0f) {
        V3 ip = o + t0 * d;
        if (dot(tubeDirection, ip - tubeStart) > 0.0f && dot(tubeDirection, ip - tubeEnd) < 0.0f) { hitT = t0; return true; }
    }
    float t1 = tc + h;
    if (t1 >= 0.0f) {
        V3 ip = o + t1 * d;
        if (dot(tubeDirection, ip - tubeStart) > 0.0f && dot(tubeDirection, ip - tubeEnd) < 0.0f) { hitT = t1; return true; }
    }
    return false;
}

// IntersectionTube main(), TubeRayTracing.glsl:452-494
inline bool intersectCapsule(V3 o, V3 d, V3 p0, V3 p1, float radius, bool capped, float& hitTOut, int& hitKindOut) {
    bool hasIntersection = false;
    float hitT = 1e7f;
    int hitKind = 0;
    float tubeT, s0T, s1T;
    if (rayTubeIntersection(o, d, p0, p1, radius, tubeT)) {
        hitT = tubeT;
        hasIntersection = true;
        hitKind = 0;
    }
    if (capped) {
        bool h0 = raySphereIntersection(o, d, p0, radius, s0T);
        bool h1 = raySphereIntersection(o, d, p1, radius, s1T);
        if (h0 && s0T < hitT) { hasIntersection = true; hitT = s0T; hitKind = 1; }
        if (h1 && s1T < hitT) { hasIntersection = true; hitT = s1T; hitKind = 2; }
    }
    hitTOut = hitT;
    hitKindOut = hitKind;
    return hasIntersection;
}

// The intersection shader only runs for rays that hit the segment's AABB (min(p0, p1) - r .. max(p0, p1) + r,
// LineDataFlow.cpp:2230-2233), and a root is only meaningful near that box: in the literal mode a hit counts iff the ray's
// line hits that box and t lies within r / |d| of the box interval.  With this rule any conservative BVH that culls against
// best + r / |d| returns the brute-force minimum of the noisy literal roots bit for bit.
inline bool literalOwnBoxRule(V3 o, V3 d, V3 p0, V3 p1, float radius, float hitT) {
    const V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const float tx0 = ((fminf(p0.x, p1.x) - radius) - o.x) * inv.x, tx1 = ((fmaxf(p0.x, p1.x) + radius) - o.x) * inv.x;
    const float ty0 = ((fminf(p0.y, p1.y) - radius) - o.y) * inv.y, ty1 = ((fmaxf(p0.y, p1.y) + radius) - o.y) * inv.y;
    const float tz0 = ((fminf(p0.z, p1.z) - radius) - o.z) * inv.z, tz1 = ((fmaxf(p0.z, p1.z) + radius) - o.z) * inv.z;
    const float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fminf(tz0, tz1));
    const float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
    const float slack = radius / length(d);
    return tn <= tf && hitT >= tn - slack && hitT <= tf + slack;
}

inline bool intersectCapsuleLiteral(V3 o, V3 d, V3 p0, V3 p1, float radius, bool capped, float& hitTOut, int& hitKindOut) {
    bool hasIntersection = false;
    float hitT = 1e7f;
    int hitKind = 0;
    float tubeT, s0T, s1T;
    if (rayTubeIntersectionLiteral(o, d, p0, p1, radius, tubeT)) { hitT = tubeT; hasIntersection = true; hitKind = 0; }
    if (capped) {
        bool h0 = raySphereIntersectionLiteral(o, d, p0, radius, s0T);
        bool h1 = raySphereIntersectionLiteral(o, d, p1, radius, s1T);
        if (h0 && s0T < hitT) { hasIntersection = true; hitT = s0T; hitKind = 1; }
        if (h1 && s1T < hitT) { hasIntersection = true; hitT = s1T; hitKind = 2; }
    }
    hitTOut = hitT;
    hitKindOut = hitKind;
    return hasIntersection;
}

// ---------------------------------------------------------------- scene + CPU LBVH
struct BvhNode {
    float bmin[3], bmax[3];
    int32_t left, right; // >=0: internal node index; <0: leaf, segment = ~value
};


} // namespace

struct lvo_scene {
    std::vector<lvo_line_point> pts;
    std::vector<uint32_t> segIdx;
    uint32_t nSeg = 0;
    std::vector<float> tf; // rgba * n
    uint32_t tfN = 0;
    std::vector<BvhNode> nodes;
    std::vector<float> leafBoxes; // 6 floats per segment (padded capsule AABB), only with a BVH
    int32_t root = -1; // may be a leaf reference (<0 encoded) when nSeg == 1
    bool rootIsLeaf = false;
    uint32_t bvhDepth = 0;
    float bvhLineWidth = -1.0f;
};

namespace {

inline void segPoints(const lvo_scene& sc, uint32_t seg, V3& p0, V3& p1) {
    p0 = ld3(sc.pts[sc.segIdx[2 * seg]].linePosition);
    p1 = ld3(sc.pts[sc.segIdx[2 * seg + 1]].linePosition);
}


struct SegBox { float mn[3], mx[3]; };

int32_t buildRange(lvo_scene& sc, const std::vector<uint64_t>& keys, const std::vector<uint32_t>& order,
                   const std::vector<SegBox>& boxes, uint32_t lo, uint32_t hi, uint32_t depth, uint32_t& maxDepth) {
    if (depth > maxDepth) maxDepth = depth;
    if (hi - lo == 1) return ~int32_t(order[lo]);
    uint64_t first = keys[lo], last = keys[hi - 1];
    uint32_t split;
    if (first == last) {
        split = (lo + hi) / 2;
    } else {
        int commonPrefix = __builtin_clzll(first ^ last);
        // largest index in [lo, hi-1) whose key shares more than commonPrefix bits with first
        uint32_t a = lo, b = hi - 1;
        while (b - a > 1) {
            uint32_t mid = (a + b) / 2;
            uint64_t x = first ^ keys[mid];
            int pre = x == 0 ? 64 : __builtin_clzll(x);
            if (pre > commonPrefix) a = mid; else b = mid;
        }
        split = a + 1;
    }
    int32_t idx = int32_t(sc.nodes.size());
    sc.nodes.push_back(BvhNode{});
    int32_t l = buildRange(sc, keys, order, boxes, lo, split, depth + 1, maxDepth);
    int32_t r = buildRange(sc, keys, order, boxes, split, hi, depth + 1, maxDepth);
    BvhNode nd;
    nd.left = l; nd.right = r;
    for (int k = 0; k < 3; k++) { nd.bmin[k] = 3.0e38f; nd.bmax[k] = -3.0e38f; }
    auto merge = [&](int32_t c) {
        const float *mn, *mx;
        if (c < 0) { mn = boxes[~c].mn; mx = boxes[~c].mx; }
        else { mn = sc.nodes[c].bmin; mx = sc.nodes[c].bmax; }
        for (int k = 0; k < 3; k++) { nd.bmin[k] = fminf(nd.bmin[k], mn[k]); nd.bmax[k] = fmaxf(nd.bmax[k], mx[k]); }
    };
    merge(l); merge(r);
    sc.nodes[idx] = nd;
    return idx;
}

// ---------------------------------------------------------------- elliptic tubes (EllipticTubeRayTracing.glsl)
// The ray tracer's "Elliptic Tubes" mode for band data: every segment is a tubelet with an elliptic cross-section (semi-axes
// radius0 = bandWidth / 2 * minBandThickness along the line normal, radius1 = bandWidth / 2 along the binormal) that twists
// from the normal at p0 to the normal at p1, found by sphere tracing in the tubelet's coordinate system (Reina et al. 2006).
struct EllipticState { bool enabled; float bandWidth, minBandThickness; V3 cameraPosition; };
static EllipticState g_ell = {false, 0.0f, 0.0f, {0.0f, 0.0f, 0.0f}};

struct EllipticScope {   // the render entry points switch the closest-hit routine to elliptic tubelets for their duration
    EllipticState saved;
    EllipticScope(bool on, float bandWidth, float minBandThickness, V3 cam) : saved(g_ell) {
        g_ell.enabled = on; g_ell.bandWidth = bandWidth; g_ell.minBandThickness = minBandThickness; g_ell.cameraPosition = cam;
    }
    ~EllipticScope() { g_ell = saved; }
};

struct M3 { V3 c0, c1, c2; };                    // columns, GLSL layout
inline V3 mulM3(const M3& m, V3 v) { return (m.c0 * v.x + m.c1 * v.y) + m.c2 * v.z; }
// matrixAxisRotationCos, EllipticTubeRayTracing.glsl:69-91 (glm::rotate with a given cosine)
inline M3 matrixAxisRotationCos(V3 axis, float cosAngle) {
    const float c = cosAngle;
    const float s = sqrtf(1.0f - cosAngle * cosAngle);
    axis = normalize(axis);
    const V3 temp = (1.0f - c) * axis;
    M3 R;
    R.c0 = v3(c + temp.x * axis.x, temp.x * axis.y + s * axis.z, temp.x * axis.z - s * axis.y);
    R.c1 = v3(temp.y * axis.x - s * axis.z, c + temp.y * axis.y, temp.y * axis.z + s * axis.x);
    R.c2 = v3(temp.z * axis.x + s * axis.y, temp.z * axis.y - s * axis.x, c + temp.z * axis.z);
    return R;
}
// computeRadius, :3-7
inline float ellComputeRadius(float r1, float r2, float phi, float rho) {
    float sn, cs;
    sincosRad(phi + rho, sn, cs);
    return r1 * r2 / sqrtf(r1 * r1 * sn * sn + r2 * r2 * cs * cs);
}
// computeNormal, :22-41
inline V3 ellComputeNormal(float r1, float r2, float phi, float rho) {
    float sinphi, cosphi, sinphirho, cosphirho;
    sincosRad(phi + rho, sinphi, cosphi);
    sincosRad(phi, sinphirho, cosphirho);
    const float r1sq = r1 * r1, r2sq = r2 * r2, r1r2 = r1 * r2;
    const float rDenomSq = r1sq * sinphi * sinphi + r2sq * cosphi * cosphi;
    const float rDenom = sqrtf(rDenomSq);
    const float r = r1r2 / rDenom;
    const float ddenomDphi = (r1sq - r2sq) * sinphi * cosphi / rDenom;
    const float drDphi = -r1r2 * ddenomDphi / rDenomSq;
    const V3 dxDphi = v3(0.0f, drDphi * cosphirho - r * sinphirho, drDphi * sinphirho + r * cosphirho);
    return cross(dxDphi, v3(1.0f, 0.0f, 0.0f));
}
// rayBoxPlaneIntersection / rayBoxIntersectionRayCoords, :126-183
inline bool ellRayBoxPlane(float o, float d, float lower, float upper, float& tNear, float& tFar) {
    if (fabsf(d) < 1e-3f) {
        if (o < lower || o > upper) return false;
    } else {
        float t0 = (lower - o) / d, t1 = (upper - o) / d;
        if (t0 > t1) { const float tmp = t0; t0 = t1; t1 = tmp; }
        if (t0 > tNear) tNear = t0;
        if (t1 < tFar) tFar = t1;
        if (tNear > tFar) return false;
        if (tFar < 0.0f) return false;
    }
    return true;
}
// The tubelet frame shared by the intersection and the closest-hit shader (:201-222 = :320-341)
struct Tubelet { V3 p0, p1, xt, yt, zt; float l, rhoR; };
inline Tubelet makeTubelet(const lvo_line_point& lp0, const lvo_line_point& lp1) {
    Tubelet T;
    T.p0 = ld3(lp0.linePosition); T.p1 = ld3(lp1.linePosition);
    const V3 n0 = ld3(lp0.lineNormal), n1 = ld3(lp1.lineNormal), t0 = ld3(lp0.lineTangent);
    T.l = length(T.p1 - T.p0);
    T.xt = normalize(T.p1 - T.p0);
    const V3 rotAxis = cross(n0, T.xt);
    const float rotCosAngle = dot(n0, T.xt);
    M3 R = {v3(1, 0, 0), v3(0, 1, 0), v3(0, 0, 1)};
    if (fabsf(rotCosAngle) > 0.999f) R = matrixAxisRotationCos(rotAxis, rotCosAngle);
    T.yt = mulM3(R, n0);
    T.zt = mulM3(R, cross(t0, n0));
    T.rhoR = -atan2Det(dot(cross(n0, n1), t0), dot(n0, n1));
    return T;
}
inline V3 toTubelet(const Tubelet& T, V3 w) { return v3(dot(T.xt, w), dot(T.yt, w), dot(T.zt, w)); } // transpose(frame) * w
inline V3 fromTubelet(const Tubelet& T, V3 p) { return (T.xt * p.x + T.yt * p.y) + T.zt * p.z; }

// IntersectionEllipticTube main(), :186-270.  The reference reports the hit whenever the driver invokes the shader, i.e. whenever
// the ray meets the segment's box; hitT itself may leave the box interval (a start point inside the surface steps backwards,
// tilted cutting planes let the surface reach past the box), and the shader's own box test is looser than a slab test (it skips
// axes with |d_i| < 1e-3).  To keep the result independent of the BVH the build accepts a hit only if the ray meets the box in the
// slab-test sense and hitT lies within bandWidth / |d| of that interval (own-box rule, as for the literal capsule roots), and
// widens the traversal's culling interval by the same amount.
inline bool intersectEllipticTube(V3 o, V3 d, const lvo_line_point& lp0, const lvo_line_point& lp1, float& hitTOut) {
    const float radius0 = g_ell.bandWidth * 0.5f * g_ell.minBandThickness;
    const float radius1 = g_ell.bandWidth * 0.5f;
    const float lwo = g_ell.bandWidth * 0.5f;
    const float po[3] = {o.x, o.y, o.z}, pd[3] = {d.x, d.y, d.z};
    float tNear = -1e7f, tFar = 1e7f;
    for (int i = 0; i < 3; i++) {
        const float lower = fminf(lp0.linePosition[i], lp1.linePosition[i]) - lwo;   // TubeAabbRenderData, LineDataFlow.cpp:2223-2234
        const float upper = fmaxf(lp0.linePosition[i], lp1.linePosition[i]) + lwo;
        if (!ellRayBoxPlane(po[i], pd[i], lower, upper, tNear, tFar)) return false;
    }
    const V3 startPoint = o + tNear * d;
    const Tubelet T = makeTubelet(lp0, lp1);
    const V3 t0 = ld3(lp0.lineTangent), t1 = ld3(lp1.lineTangent);
    // left and right cutting planes
    const V3 El = t0; const float Elw = -dot(El, T.p0);
    const V3 Er = v3(-t1.x, -t1.y, -t1.z); const float Erw = -dot(Er, T.p1);
    V3 p = toTubelet(T, startPoint - T.p0);
    const V3 dd = toTubelet(T, d);
    float hitT = tNear, dTmp = 0.0f;
    for (int i = 0; i < 80; i++) {
        const float t = clampf(p.x / T.l, 0.0f, 1.0f);
        const float rhoX = t * T.rhoR;
        const float phi = atan2Det(p.z, p.y);
        const float r = ellComputeRadius(radius0, radius1, phi, rhoX);
        dTmp = sqrtf(p.y * p.y + p.z * p.z) - r;
        dTmp *= 0.25f;
        p = p + dd * dTmp;
        hitT += dTmp;
        if (dTmp < 1e-5f) break;
    }
    const V3 pointWorld = fromTubelet(T, p) + T.p0;
    const float eps1 = fabsf(dot(El, normalize(g_ell.cameraPosition - T.p0))) * 5e-5f;
    const float eps2 = fabsf(dot(Er, normalize(g_ell.cameraPosition - T.p1))) * 5e-5f;
    const bool isNotCulledLeft = dot(El, pointWorld) + Elw > -eps1;
    const bool isNotCulledRight = dot(Er, pointWorld) + Erw > -eps2;
    if (!(dTmp < 1e-4f && hitT > 0.0f && isNotCulledLeft && isNotCulledRight)) return false;
    // own-box rule: the driver only invokes the shader for rays that meet the box (the shader's own test above treats
    // directions with |d_i| < 1e-3 as parallel and is looser than that), and hitT must lie within bandWidth / |d| of the interval
    {
        const V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        const V3 p0 = ld3(lp0.linePosition), p1 = ld3(lp1.linePosition);
        const float tx0 = ((fminf(p0.x, p1.x) - lwo) - o.x) * inv.x, tx1 = ((fmaxf(p0.x, p1.x) + lwo) - o.x) * inv.x;
        const float ty0 = ((fminf(p0.y, p1.y) - lwo) - o.y) * inv.y, ty1 = ((fmaxf(p0.y, p1.y) + lwo) - o.y) * inv.y;
        const float tz0 = ((fminf(p0.z, p1.z) - lwo) - o.z) * inv.z, tz1 = ((fmaxf(p0.z, p1.z) + lwo) - o.z) * inv.z;
        const float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fminf(tz0, tz1));
        const float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
        const float slack = g_ell.bandWidth / sqrtf(dot(d, d));
        if (!(tn <= tf && hitT >= tn - slack && hitT <= tf + slack)) return false;
    }
    hitTOut = hitT;
    return true;
}

inline bool childBox(const lvo_scene& sc, int32_t c, V3 o, V3 inv, float tMin, float tMax, float& tNear);

struct Hit { float t; uint32_t seg; int kind; };

inline bool childBox(const lvo_scene& sc, int32_t c, V3 o, V3 inv, float tMin, float tMax, float& tNear) {
    if (c < 0) {
        const float* b = &sc.leafBoxes[6 * size_t(~c)];
        return rayBox(b, b + 3, o, inv, tMin, tMax, tNear);
    }
    return rayBox(sc.nodes[c].bmin, sc.nodes[c].bmax, o, inv, tMin, tMax, tNear);
}

// Closest hit over all capsules, hit accepted iff tMin <= t <= tMax (reportIntersectionEXT semantics),
// ties -> lowest segment index (the driver's tie order is arbitrary).
inline bool closestHit(const lvo_scene& sc, float radius, bool capped, bool useBvh, V3 o, V3 d, float tMin, float tMax,
                       Hit& out, Counters& cnt) {
    cnt.rays++;
    bool found = false;
    float best = tMax;
    uint32_t bestSeg = 0xFFFFFFFFu;
    int bestKind = 0;
    auto testSeg = [&](uint32_t seg) {
        cnt.prims++;
        V3 p0, p1; segPoints(sc, seg, p0, p1);
        float t; int kind = 0;
        if (g_ell.enabled ? intersectEllipticTube(o, d, sc.pts[sc.segIdx[2 * seg]], sc.pts[sc.segIdx[2 * seg + 1]], t)
            : g_dev.literalIntersection ? (intersectCapsuleLiteral(o, d, p0, p1, radius, capped, t, kind) &&
                                           literalOwnBoxRule(o, d, p0, p1, radius, t))
                                        : intersectCapsule(o, d, p0, p1, radius, capped, t, kind)) {
            if (t >= tMin && t <= tMax && (!found || t < best || (t == best && seg < bestSeg))) {
                found = true; best = t; bestSeg = seg; bestKind = kind;
            }
        }
    };
    if (!useBvh || sc.root == -1) {
        for (uint32_t seg = 0; seg < sc.nSeg; seg++) testSeg(seg);
    } else if (sc.rootIsLeaf) {
        testSeg(uint32_t(~sc.root));
    } else {
        V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        int32_t stack[128];
        int sp = 0;
        stack[sp++] = sc.root;
        while (sp > 0) {
            int32_t n = stack[--sp];
            if (n < 0) { testSeg(uint32_t(~n)); continue; }
            const BvhNode& nd = sc.nodes[n];
            cnt.nodes++;
            // test both children, descend into the nearer one first
            float tl, tr;
            // the literal roots carry up to ~0.25 r of float32 noise in t (far more than the boxes' padding): cull against
            // best + r there, so that the traversal still returns the brute-force minimum of the noisy values
            // (and may lie up to r / |d| outside the segment's box interval, literalOwnBoxRule: both ends of the culling interval
            // are widened by that much)
            const float slack = g_ell.enabled ? g_ell.bandWidth / sqrtf(dot(d, d))   // intersectEllipticTube's own-box rule
                                : g_dev.literalIntersection ? radius / sqrtf(dot(d, d)) : 0.0f;
            const float limit = found ? fminf(best + slack, tMax + slack) : tMax + slack;
            bool hl = childBox(sc, nd.left, o, inv, tMin - slack, limit, tl);
            bool hr = childBox(sc, nd.right, o, inv, tMin - slack, limit, tr);
            if (hl && hr) {
                if (tr < tl) { stack[sp++] = nd.left; stack[sp++] = nd.right; }
                else { stack[sp++] = nd.right; stack[sp++] = nd.left; }
            } else if (hl) {
                stack[sp++] = nd.left;
            } else if (hr) {
                stack[sp++] = nd.right;
            }
        }
    }
    out.t = best; out.seg = bestSeg; out.kind = bestKind;
    return found;
}

// All capsule entry hits in [tMin, tMax], ascending segment order.
inline void allHits(const lvo_scene& sc, float radius, bool capped, bool useBvh, V3 o, V3 d, float tMin, float tMax,
                    std::vector<Hit>& out, Counters& cnt) {
    cnt.rays++;
    out.clear();
    auto testSeg = [&](uint32_t seg) {
        cnt.prims++;
        V3 p0, p1; segPoints(sc, seg, p0, p1);
        float t; int kind = 0;
        if (g_ell.enabled ? intersectEllipticTube(o, d, sc.pts[sc.segIdx[2 * seg]], sc.pts[sc.segIdx[2 * seg + 1]], t)
            : g_dev.literalIntersection ? (intersectCapsuleLiteral(o, d, p0, p1, radius, capped, t, kind) &&
                                           literalOwnBoxRule(o, d, p0, p1, radius, t))
                                        : intersectCapsule(o, d, p0, p1, radius, capped, t, kind)) {
            if (t >= tMin && t <= tMax) out.push_back(Hit{t, seg, kind});
        }
    };
    if (!useBvh || sc.root == -1) {
        for (uint32_t seg = 0; seg < sc.nSeg; seg++) testSeg(seg);
    } else if (sc.rootIsLeaf) {
        testSeg(uint32_t(~sc.root));
    } else {
        V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        std::vector<int32_t> stack;
        stack.push_back(sc.root);
        while (!stack.empty()) {
            int32_t n = stack.back(); stack.pop_back();
            if (n < 0) { testSeg(uint32_t(~n)); continue; }
            const BvhNode& nd = sc.nodes[n];
            cnt.nodes++;
            float tl, tr;
            const float slack = g_ell.enabled ? g_ell.bandWidth / sqrtf(dot(d, d))
                                : g_dev.literalIntersection ? radius / sqrtf(dot(d, d)) : 0.0f; // see closestHit
            if (childBox(sc, nd.right, o, inv, tMin - slack, tMax + slack, tr)) stack.push_back(nd.right);
            if (childBox(sc, nd.left, o, inv, tMin - slack, tMax + slack, tl)) stack.push_back(nd.left);
        }
        std::sort(out.begin(), out.end(), [](const Hit& a, const Hit& b) { return a.seg < b.seg; });
    }
}

// ---------------------------------------------------------------- per-frame constants

// TransferFunction.glsl:66-71 with the texture definition owned by the build
inline void transferFunction(const lvo_scene& sc, const lvo_params& P, float attr, float out[4]) {
    float pos = clampf((attr - P.attrMin) / (P.attrMax - P.attrMin), 0.0f, 1.0f);
    int n = int(sc.tfN);
    float u = pos * float(n) - 0.5f;
    float fl = floorf(u);
    float f = u - fl;
    int i0 = int(fl), i1 = i0 + 1;
    i0 = std::min(std::max(i0, 0), n - 1);
    i1 = std::min(std::max(i1, 0), n - 1);
    for (int k = 0; k < 4; k++) out[k] = sc.tf[4 * i0 + k] * (1.0f - f) + sc.tf[4 * i1 + k] * f;
}

// AmbientOcclusion.glsl:84-99 (non-SSAO branch): the reference projects the hit back to the screen and samples the AO
// texture there (linear filter, clamp to edge -- sampler definition owned by the build).
inline float getAoFactor(const lvo_params& P, float aoTexel, V3 ssp) {
    // Jittered primary rays (several samples per frame / accumulated frames): the sample's hit projects to an arbitrary
    // position inside its pixel, so the reference's lookup blends the neighbouring texels -> evaluated literally.
    // Pixel-centre rays project onto their own texel centre (to float32 rounding: neighbour weights < 2e-4, measured in
    // tests/test_deviations.py), where the build reads the launching pixel's texel directly -- which keeps screen tiles
    // free of a halo; the deviation switch forces the literal evaluation there too.
    if (g_dev.aoImage && (P.useJitteredRays || g_dev.referenceAoLookup)) {
        // literal: ndc = projectionMatrix * vec4(screenSpacePosition, 1); texture(aoTexture, ndc.xy / ndc.w * 0.5 + 0.5).x
        // with a linear, clamp-to-edge sampler (texel centres at (i + 0.5) / size)
        const V4 ndc = mulM4(P.proj, V4{ssp.x, ssp.y, ssp.z, 1.0f});
        const float u = (ndc.x / ndc.w) * 0.5f + 0.5f, v = (ndc.y / ndc.w) * 0.5f + 0.5f;
        const float fx = u * float(P.width) - 0.5f, fy = v * float(P.height) - 0.5f;
        const float x0f = floorf(fx), y0f = floorf(fy);
        const float wx = fx - x0f, wy = fy - y0f;
        auto cl = [](float c, uint32_t n) { return uint32_t(fminf(fmaxf(c, 0.0f), float(n - 1))); };
        const uint32_t xa = cl(x0f, P.width), xb = cl(x0f + 1.0f, P.width), ya = cl(y0f, P.height), yb = cl(y0f + 1.0f, P.height);
        const float* A = g_dev.aoImage;
        const float top = A[size_t(ya) * P.width + xa] * (1.0f - wx) + A[size_t(ya) * P.width + xb] * wx;
        const float bot = A[size_t(yb) * P.width + xa] * (1.0f - wx) + A[size_t(yb) * P.width + xb] * wx;
        aoTexel = top * (1.0f - wy) + bot * wy;
    }
    float aoFactor = powDet(aoTexel, P.aoGamma);
    return fmaxf(0.0f, (1.0f - P.aoStrength) + P.aoStrength * aoFactor);
}

// blinnPhongShadingTube, Lighting.glsl:100-191
inline void blinnPhongShadingTube(const lvo_params& P, const Frame& F, float aoTexel, const float base[4], V3 fragPos,
                                  V3 ssp, V3 fragmentNormal, V3 fragmentTangent, float out[4], float exponent = 1.7f) {
    float kA, kD;
    const float kS = 0.3f, s = 30.0f;
    float aoF = 1.0f;
    if (P.useAmbientOcclusion) {
        aoF = getAoFactor(P, aoTexel, ssp);
        kA = 0.2f + (1.0f - aoF) * 0.5f;
        kD = 0.9f * aoF;
    } else {
        kA = 0.1f;
        kD = 0.9f;
    }
    V3 n = normalizeShade(fragmentNormal);
    V3 t = normalizeShade(fragmentTangent);
    V3 v = normalizeShade(F.cameraPosition - fragPos);
    V3 l = v;
    V3 h = normalizeShade(v + l);
    V3 helperVec = normalizeShade(cross(t, l));
    V3 newL = normalizeShade(cross(helperVec, t));
    // exponent: 1.7, or 1.0 on bands (USE_BANDS && useBand, Lighting.glsl:158-162)
    float cosNormal1 = powDet(clampf(fabsf(dot(n, l)), 0.0f, 1.0f), exponent);
    float cosNormal2 = powDet(clampf(fabsf(dot(n, newL)), 0.0f, 1.0f), exponent);
    float cosNormalCombined = 0.3f * cosNormal1 + 0.7f * cosNormal2;
    float spec = kS * powDet(clampf(fabsf(dot(n, h)), 0.0f, 1.0f), s);
    float phong[3];
    for (int k = 0; k < 3; k++) {
        float Ia = kA * base[k];
        float Id = (kD * cosNormalCombined) * base[k];
        float Is = spec * 1.0f;
        phong[k] = (Ia + Id) + Is;
    }
    if (P.useAmbientOcclusion) {
        for (int k = 0; k < 3; k++) phong[k] *= aoF;
    }
    if (P.useDepthCues) {
        float dcf = clampf((-ssp.z - P.minDepth) / (P.maxDepth - P.minDepth), 0.0f, 1.0f);
        dcf = (dcf * dcf) * P.depthCueStrength;
        for (int k = 0; k < 3; k++) phong[k] = mixf(phong[k], 0.5f, dcf);
    }
    out[0] = phong[0]; out[1] = phong[1]; out[2] = phong[2]; out[3] = base[3];
}

// Raster variant of the fragment colour (the PPLL gather runs the RASTER tube shader, LinePassGeometryShaderTubes.glsl:732-1129,
// not RayHitCommon.glsl): EPSILON_OUTLINE = 0 and EPSILON_WHITE = fwidth(ribbonPosition) (:1079-1087), cap halo
// min(ribbonPosition, abs(ribbonPosition2)) (:815).  fwidth needs the ribbon coordinate of the 2 x 2 quad partners' invocations of
// the SAME primitive (helper invocations extrapolate a triangle's attributes beyond its edge); for ray-generated fragments the
// analogue is exact: the ribbon coordinate of a fragment is a function of the VIEWING RAY alone -- |cross(newV, n)| is the
// distance between the ray and the tube axis over the radius (chord geometry in the cross-section plane), the USE_BANDS
// coordinate is the position of the ray's trace in the cross-section plane between the two silhouette points -- so it is
// evaluated for the rays through the quad partners (x ^ 1, y) and (x, y ^ 1) with respect to the fragment's segment, whether or
// not those rays hit it: fwidth = |f(x ^ 1, y) - f(x, y)| + |f(x, y ^ 1) - f(x, y)| (fine derivatives of the 2 x 2 quad).
// The three rays come from the AFFINE ray generator (ray directions are only used up to their length here): D(x, y) = invView *
// (invProj * (ndc(x, y), 1, 1)).xyz without the normalisation of TubeRayTracing.glsl:225-226, and D(x +- 1, y) = D +- dD/dx,
// D(x, y +- 1) = D +- dD/dy with the constant steps dD/dx = invView * (invProj * (2 / W, 0, 0, 0)).xyz, dD/dy likewise.
struct RasterQuad { V3 d0, dX, dY; };   // un-normalised directions of the pixel's own ray and of its two quad partners (origin = camera)
inline RasterQuad makeRasterQuad(const lvo_params& P, const Frame& F, uint32_t x, uint32_t y) {
    const float sxp = 2.0f / float(P.width), syp = 2.0f / float(P.height);   // NDC step of one pixel
    const float ndcx = (float(x) + 0.5f) * sxp - 1.0f;
    const float ndcy = (float(y) + 0.5f) * syp - 1.0f;
    const V4 target = mulM4(F.invProj, V4{ndcx, ndcy, 1.0f, 1.0f});
    const V4 dir = mulM4(F.invView, V4{target.x, target.y, target.z, 0.0f});
    const V4 gx = mulM4(F.invProj, V4{sxp, 0.0f, 0.0f, 0.0f});
    const V4 gy = mulM4(F.invProj, V4{0.0f, syp, 0.0f, 0.0f});
    const V4 Gx = mulM4(F.invView, V4{gx.x, gx.y, gx.z, 0.0f});
    const V4 Gy = mulM4(F.invView, V4{gy.x, gy.y, gy.z, 0.0f});
    const float sx = (x & 1u) ? -1.0f : 1.0f, sy = (y & 1u) ? -1.0f : 1.0f;
    RasterQuad q;
    q.d0 = v3(dir.x, dir.y, dir.z);
    q.dX = v3(dir.x + sx * Gx.x, dir.y + sx * Gx.y, dir.z + sx * Gx.z);
    q.dY = v3(dir.x + sy * Gy.x, dir.y + sy * Gy.y, dir.z + sy * Gy.z);
    return q;
}
// ribbon coordinate of the ray (cam, d) with respect to a tube axis (point, unit direction t): signed ray-axis distance / radius,
// clamped like the shader clamps ribbonPosition (:1-style clamp(ribbonPosition, -1, 1), :962)
inline float tubeRibbonOfRay(V3 cam, V3 d, V3 axisPoint, V3 t, float radius) {
    // t . (wp x dp) = (t x wp) . d  and  |dp|^2 = |d|^2 - (d . t)^2  (t unit, wp and t x wp perpendicular to t)
    const V3 w = cam - axisPoint;
    const V3 wp = w - dot(w, t) * t;
    const V3 k = cross(t, wp);
    const float dt = dot(d, t);
    return clampf(dot(k, d) / (sqrtf(dot(d, d) - dt * dt) * radius), -1.0f, 1.0f);
}
// cap variant (:785-815): ribbonPosition = min(|cross(v, n)|, |ribbonPosition2|) with the SPHERE's normal n.  |cross(newV, n)| of a
// cap fragment depends on where along the axis the hit lies, not on the ray alone, so the partners' values are taken where
// their rays meet the TANGENT PLANE of the cap at the fragment (the plane a rasterised cap triangle extrapolates on), with the
// sphere's normal direction there: n' = normalize(point - centre).
inline float capRibbonOfRay(V3 cam, V3 d, V3 hit, V3 hitNormal, V3 centre, V3 t) {
    const float s = dot(hit - cam, hitNormal) / dot(d, hitNormal);
    const V3 q = cam + d * s;
    const V3 n = normalizeShade(q - centre);
    const V3 vv = normalizeShade(cam - q);
    const V3 helperVec = normalizeShade(cross(t, vv));
    const V3 newV = normalizeShade(cross(helperVec, t));
    const V3 crossProdVn = cross(vv, n);
    float ribbonPosition2 = length(cross(newV, n));
    if (dot(t, crossProdVn) < 0.0f) ribbonPosition2 = -ribbonPosition2;
    ribbonPosition2 = clampf(ribbonPosition2, -1.0f, 1.0f);
    return fminf(length(crossProdVn), fabsf(ribbonPosition2));
}

// USE_BANDS arguments of computeFragmentColor (RayHitCommon.glsl:82-90,148-190)
// rasterEpsWhite >= 0: raster variant of the outline (EPSILON_OUTLINE = 0, EPSILON_WHITE = this value); < 0: the ray tracer's
struct BandArgs { bool useBand; float phi; V3 linePosition, lineNormal; float rasterEpsWhite = -1.0f; bool shadeBands = true; };
// USE_ROTATING_HELICITY_BANDS arguments of computeFragmentColor (RayHitCommon.glsl:82-93): the angle around the tube and the
// interpolated lineRotation x helicityRotationFactor
// USE_HELICITY_BANDS_TEXTURE: the twist-line texture (LineDataFlow.cpp:93-171), a test hook like the other global switches.  sampler2D
// with REPEAT addressing sampled at (u, 0.5): level 0 in the ray tracer's shaders (texture() without derivatives, RayHitCommon.glsl:
// 66-72), textureGrad(.., (dFdx(globalPos), 0), (dFdy(globalPos), 0)) in the raster shader (LinePassGeometryShaderTubes.glsl:724-730).
// Build-owned where Vulkan leaves room: texel centres (i + 0.5) / size, linear weights in full float, lambda = log2(max(|dudx|, |dudy|)
// * width) clamped to the chain, nearest mip = ceil(lambda + 0.5) - 1, chain = intlog2(max(w, h)) levels of 2 x 2 box averages in float.
static struct TwistTexture {
    std::vector<float> texels;   // RGBA float, levels back to back
    uint32_t w = 0, h = 0, levels = 0, mode = 5;
    bool use = false;
} g_twist;
static void twistLevel(uint32_t level, float u, float v, bool linear, float out[4]) {
    uint32_t w = g_twist.w, h = g_twist.h;
    size_t base = 0;
    for (uint32_t l = 0; l < level; l++) { base += size_t(w) * h * 4; w = std::max(w >> 1, 1u); h = std::max(h >> 1, 1u); }
    const float* T = g_twist.texels.data() + base;
    auto wrap = [](int i, uint32_t n) { int m = i % int(n); return uint32_t(m < 0 ? m + int(n) : m); };
    if (!linear) {
        const uint32_t i = wrap(int(floorf(u * float(w))), w), j = wrap(int(floorf(v * float(h))), h);
        for (int c = 0; c < 4; c++) out[c] = T[(size_t(j) * w + i) * 4 + c];
        return;
    }
    const float x = u * float(w) - 0.5f, y = v * float(h) - 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y);
    const float a = x - fx0, b = y - fy0;
    const uint32_t i0 = wrap(int(fx0), w), i1 = wrap(int(fx0) + 1, w), j0 = wrap(int(fy0), h), j1 = wrap(int(fy0) + 1, h);
    for (int c = 0; c < 4; c++)
        out[c] = mixf(mixf(T[(size_t(j0) * w + i0) * 4 + c], T[(size_t(j0) * w + i1) * 4 + c], a),
                      mixf(T[(size_t(j1) * w + i0) * 4 + c], T[(size_t(j1) * w + i1) * 4 + c], a), b);
}
static void twistSample(float u, float dudx, float dudy, bool useGrad, float out[4]) {
    const uint32_t mode = g_twist.mode;
    const bool linear = mode == 1u || mode == 3u || mode == 5u;
    const float v = 0.5f;
    if (mode < 2u || !useGrad || g_twist.levels <= 1u) { twistLevel(0u, u, v, linear, out); return; }
    const float rho = fmaxf(fabsf(dudx), fabsf(dudy)) * float(g_twist.w);
    const float maxLevel = float(g_twist.levels - 1u);
    const float lambda = rho > 1.0f ? fminf(log2Det(rho), maxLevel) : 0.0f;
    if (mode == 2u || mode == 3u) {
        const float d = fminf(fmaxf(ceilf(lambda + 0.5f) - 1.0f, 0.0f), maxLevel);
        twistLevel(uint32_t(d), u, v, linear, out);
        return;
    }
    const float dhi = floorf(lambda), delta = lambda - dhi;
    const uint32_t lhi = uint32_t(dhi), llo = std::min(lhi + 1u, g_twist.levels - 1u);
    float a[4], b[4];
    twistLevel(lhi, u, v, linear, a);
    twistLevel(llo, u, v, linear, b);
    for (int c = 0; c < 4; c++) out[c] = mixf(a[c], b[c], delta);
}
// rasterAaf >= 0: the raster shader's stripe (LinePassGeometryShaderTubes.glsl:716-721,1046-1052): offset 0.1 instead of w / 2, aaf =
// fwidth(phi + fragmentRotation) over the pixel quad instead of 10 EPSILON_OUTLINE
struct HelicityArgs { float phi, fragmentRotation, rotationSeparatorScale; float rasterAaf = -1.0f; float dx = 0.0f, dy = 0.0f; };
inline void computeFragmentColor(const lvo_scene& sc, const lvo_params& P, const Frame& F, float aoTexel, V3 fragPos,
                                 V3 fragmentNormal, V3 fragmentTangent, bool isCap, float fragmentAttribute,
                                 float hitColor[4], float& payloadHitT, const BandArgs* bands = nullptr,
                                 const HelicityArgs* hel = nullptr);

// Static RTAO prebaking: AO factors per (parametrisation vertex, angular subdivision) + the per-line-vertex blending
// weights that map a line vertex id to the parametrisation (VulkanAmbientOcclusionBaker.cpp:563-653).
struct PrebakedAo {
    const float* factors;
    const float* blendingWeights;
    uint32_t numLineVertices, numParametrizationVertices, numAoTubeSubdivisions;
};
// the table the PPLL gather shades with while ambient_occlusion_mode = "RTAO (Prebaker)" (lvo_set_ppll_prebaked_ao; tests only)
static PrebakedAo g_ppllPrebaked = {nullptr, nullptr, 0u, 0u, 0u};


// getAoFactor(interpolatedVertexId, phi), AmbientOcclusion.glsl:49-75, WITHOUT its last two lines (pow(gamma) and the
// strength mapping): those are the same as for the screen-space texture and are applied by getAoFactor(P, aoTexel).
inline float prebakedAoLookup(const PrebakedAo& pb, float interpolatedVertexId, float phi) {
    uint32_t lastLinePointIdx = uint32_t(interpolatedVertexId);
    uint32_t nextLinePointIdx = std::min(lastLinePointIdx + 1u, pb.numLineVertices - 1u);
    float interpolationFactor = interpolatedVertexId - floorf(interpolatedVertexId);
    float blendingWeight = mixf(pb.blendingWeights[lastLinePointIdx], pb.blendingWeights[nextLinePointIdx], interpolationFactor);
    uint32_t lastVertexIdx = uint32_t(blendingWeight);
    uint32_t nextVertexIdx = std::min(lastVertexIdx + 1u, pb.numParametrizationVertices - 1u);
    float interpolationFactorLine = blendingWeight - floorf(blendingWeight);
    const uint32_t N = pb.numAoTubeSubdivisions;
    float circleIdxFlt = clampf(phi / 6.28318530717958647692f * float(N), 0.0f, float(N));
    uint32_t circleIdxLast = (uint32_t(floorf(circleIdxFlt)) + N) % N;
    uint32_t circleIdxNext = (circleIdxLast + 1u) % N;
    float interpolationFactorCircle = circleIdxFlt - floorf(circleIdxFlt);
    float aoFactor00 = pb.factors[circleIdxLast + size_t(N) * lastVertexIdx];
    float aoFactor01 = pb.factors[circleIdxLast + size_t(N) * nextVertexIdx];
    float aoFactor10 = pb.factors[circleIdxNext + size_t(N) * lastVertexIdx];
    float aoFactor11 = pb.factors[circleIdxNext + size_t(N) * nextVertexIdx];
    float aoFactor0 = mixf(aoFactor00, aoFactor01, interpolationFactorLine);
    float aoFactor1 = mixf(aoFactor10, aoFactor11, interpolationFactorLine);
    return mixf(aoFactor0, aoFactor1, interpolationFactorCircle);
}

// ClosestHitTubeAnalytic main() (TubeRayTracing.glsl:512-613) + computeFragmentColor (RayHitCommon.glsl:74-543),
// flow lines: USE_CAPPED_TUBES / USE_HALOS / USE_DEPTH_CUES / USE_AMBIENT_OCCLUSION switches only.
// Writes payload {hitColor, hitT}.
inline float bandsRibbonOfRay(V3 cameraPosition, V3 d, V3 linePosition, V3 lineNormal, V3 fragmentTangent, V3 t, float lineRadius,
                              float thickness);
inline void shadeHit(const lvo_scene& sc, const lvo_params& P, const Frame& F, float aoTexel, V3 o, V3 d, const Hit& h,
                     float hitColor[4], float& payloadHitT, const PrebakedAo* pb = nullptr, const RasterQuad* rq = nullptr) {
    uint32_t i0 = sc.segIdx[2 * h.seg], i1 = sc.segIdx[2 * h.seg + 1];
    const lvo_line_point& lp0 = sc.pts[i0];
    const lvo_line_point& lp1 = sc.pts[i1];
    V3 P0 = ld3(lp0.linePosition), P1 = ld3(lp1.linePosition);
    V3 fragPos = o + d * h.t;
    V3 linePointInterpolated;
    float fragmentAttribute;
    V3 v = P1 - P0;
    if (h.kind == 0) {
        V3 u = fragPos - P0;
        float t = dot(v, u) / dot(v, v);
        linePointInterpolated = P0 + t * v;
        fragmentAttribute = (1.0f - t) * lp0.lineAttribute + t * lp1.lineAttribute;
    } else if (h.kind == 1) {
        linePointInterpolated = P0;
        fragmentAttribute = lp0.lineAttribute;
    } else {
        linePointInterpolated = P1;
        fragmentAttribute = lp1.lineAttribute;
    }
    V3 fragmentTangent = normalize(v);
    V3 fragmentNormal = normalize(fragPos - linePointInterpolated);
    bool isCap = h.kind != 0;
    // raster variant (PPLL gather): fwidth of the ribbon coordinate over the 2 x 2 quad (see RasterQuad); circular tubes
    float rasterEps = -1.0f;
    if (rq && !P.useBands) {
        const bool cap = P.useCappedTubes && isCap;
        const float f0 = cap ? capRibbonOfRay(F.cameraPosition, rq->d0, fragPos, fragmentNormal, linePointInterpolated, fragmentTangent)
                             : tubeRibbonOfRay(F.cameraPosition, rq->d0, linePointInterpolated, fragmentTangent, F.radius);
        const float fx = cap ? capRibbonOfRay(F.cameraPosition, rq->dX, fragPos, fragmentNormal, linePointInterpolated, fragmentTangent)
                             : tubeRibbonOfRay(F.cameraPosition, rq->dX, linePointInterpolated, fragmentTangent, F.radius);
        const float fy = cap ? capRibbonOfRay(F.cameraPosition, rq->dY, fragPos, fragmentNormal, linePointInterpolated, fragmentTangent)
                             : tubeRibbonOfRay(F.cameraPosition, rq->dY, linePointInterpolated, fragmentTangent, F.radius);
        rasterEps = fabsf(fx - f0) + fabsf(fy - f0);
    }
    if (pb) {
        // TubeRayTracing.glsl:551-563: angle around the tube relative to the line normal, interpolated vertex id.
        // (acos argument clamped to [-1, 1]: GLSL leaves acos undefined outside, rounding can exceed it by an ulp.)
        const float ts = h.kind == 0 ? dot(v, fragPos - P0) / dot(v, v) : (h.kind == 1 ? 0.0f : 1.0f);
        V3 lineNormal = (1.0f - ts) * ld3(lp0.lineNormal) + ts * ld3(lp1.lineNormal);
        float phi = acosf(clampf(dot(fragmentNormal, lineNormal), -1.0f, 1.0f));
        float val = dot(lineNormal, cross(fragmentNormal, fragmentTangent));
        if (val < 0.0f) phi = 2.0f * 3.14159265358979323846f - phi;
        float fragmentVertexId = (1.0f - ts) * float(i0) + ts * float(i1);
        aoTexel = prebakedAoLookup(*pb, fragmentVertexId, phi);
    }
    if (P.useHelicityBands) {
        // USE_ROTATING_HELICITY_BANDS, TubeRayTracing.glsl:551-567: phi as for the AO lookup (acos through the build's atan2),
        // fragmentRotation = lerp(lineRotation) * helicityRotationFactor
        const float ts = h.kind == 0 ? dot(v, fragPos - P0) / dot(v, v) : (h.kind == 1 ? 0.0f : 1.0f);
        const V3 lineNormal = (1.0f - ts) * ld3(lp0.lineNormal) + ts * ld3(lp1.lineNormal);
        const float cphi = clampf(dot(fragmentNormal, lineNormal), -1.0f, 1.0f);
        HelicityArgs hl;
        hl.phi = atan2Det(sqrtf((1.0f - cphi) * (1.0f + cphi)), cphi);
        if (dot(lineNormal, cross(fragmentNormal, fragmentTangent)) < 0.0f) hl.phi = 2.0f * 3.14159265358979323846f - hl.phi;
        hl.fragmentRotation = ((1.0f - ts) * lp0.lineRotation + ts * lp1.lineRotation) * P.helicityRotationFactor;
        hl.rotationSeparatorScale = 1.0f; // ClosestHitTubeAnalytic has no UNIFORM_HELICITY_BAND_WIDTH branch
        BandArgs rb; rb.shadeBands = false; rb.useBand = false; rb.phi = 0.0f; rb.linePosition = rb.lineNormal = v3(0, 0, 0);
        rb.rasterEpsWhite = rasterEps;
        computeFragmentColor(sc, P, F, aoTexel, fragPos, fragmentNormal, fragmentTangent, isCap, fragmentAttribute, hitColor,
                             payloadHitT, rq ? &rb : nullptr, &hl);
        return;
    }
    if (P.useBands) {
        // band data with the circular analytic tubes: USE_BANDS is defined, ANALYTIC_TUBE_INTERSECTIONS sets useBand = false
        // (RayHitCommon.glsl:164-166); phi and the line normal as TubeRayTracing.glsl:551-560 (acos through the build's atan2)
        const float ts = h.kind == 0 ? dot(v, fragPos - P0) / dot(v, v) : (h.kind == 1 ? 0.0f : 1.0f);
        BandArgs b;
        b.useBand = false;
        b.lineNormal = (1.0f - ts) * ld3(lp0.lineNormal) + ts * ld3(lp1.lineNormal);
        const float cphi = clampf(dot(fragmentNormal, b.lineNormal), -1.0f, 1.0f);
        b.phi = atan2Det(sqrtf((1.0f - cphi) * (1.0f + cphi)), cphi);
        if (dot(b.lineNormal, cross(fragmentNormal, fragmentTangent)) < 0.0f) b.phi = 2.0f * 3.14159265358979323846f - b.phi;
        b.linePosition = linePointInterpolated;
        if (rq) {
            // raster variant with USE_BANDS (:1079-1083): the band coordinate of the partner rays in the same cross-section plane;
            // caps keep the circular-tube cap coordinate (:785-815 precede the USE_BANDS branch)
            const V3 tN = normalize(fragmentTangent);
            const bool cap = P.useCappedTubes && isCap;
            const float r = P.lineWidth * 0.5f;
            auto f = [&](V3 dir) {
                return cap ? capRibbonOfRay(F.cameraPosition, dir, fragPos, fragmentNormal, linePointInterpolated, fragmentTangent)
                           : bandsRibbonOfRay(F.cameraPosition, dir, b.linePosition, b.lineNormal, fragmentTangent, tN, r, 1.0f);
            };
            const float f0 = f(rq->d0);
            b.rasterEpsWhite = fabsf(f(rq->dX) - f0) + fabsf(f(rq->dY) - f0);
        }
        computeFragmentColor(sc, P, F, aoTexel, fragPos, fragmentNormal, fragmentTangent, isCap, fragmentAttribute, hitColor,
                             payloadHitT, &b);
        return;
    }
    BandArgs rb; rb.shadeBands = false; rb.useBand = false; rb.phi = 0.0f; rb.linePosition = rb.lineNormal = v3(0, 0, 0);
    rb.rasterEpsWhite = rasterEps;
    computeFragmentColor(sc, P, F, aoTexel, fragPos, fragmentNormal, fragmentTangent, isCap, fragmentAttribute, hitColor,
                         payloadHitT, rq ? &rb : nullptr);
}

// ClosestHitEllipticTubeAnalytic main(), EllipticTubeRayTracing.glsl:303-441: position in the tubelet frame -> t, phi, rho ->
// normal of the twisted elliptic surface; attribute and line normal interpolated with t; the angle in the ellipse's own
// parametrisation p = (r1 cos, r2 sin) for the band shading.
struct EllipticSurface { V3 fragPos, normal, tangent, linePosition, lineNormal; float t, phiLine, attribute; };
inline EllipticSurface ellipticSurface(const lvo_params& P, V3 o, V3 d, float hitT, const lvo_line_point& lp0, const lvo_line_point& lp1) {
    EllipticSurface E;
    E.fragPos = o + d * hitT;
    const Tubelet T = makeTubelet(lp0, lp1);
    const V3 p = toTubelet(T, E.fragPos - T.p0);
    const float radius0 = P.bandWidth * 0.5f * P.minBandThickness;
    const float radius1 = P.bandWidth * 0.5f;
    const float t = clampf(p.x / T.l, 0.0f, 1.0f);
    const float phi = atan2Det(p.z, p.y);
    const float rho = t * T.rhoR;
    E.t = t;
    E.normal = normalize(fromTubelet(T, ellComputeNormal(radius0, radius1, phi, rho)));
    E.attribute = (1.0f - t) * lp0.lineAttribute + t * lp1.lineAttribute;
    E.linePosition = (1.0f - t) * T.p0 + t * T.p1;
    E.tangent = T.xt;
    E.lineNormal = normalize((1.0f - t) * ld3(lp0.lineNormal) + t * ld3(lp1.lineNormal));
    float sinphi, cosphi;
    sincosRad(phi + rho, sinphi, cosphi);
    const float phiDenomInv = 1.0f / sqrtf(radius0 * radius0 * sinphi * sinphi + radius1 * radius1 * cosphi * cosphi);
    const float sinPhiLine = radius0 * sinphi * phiDenomInv;
    const float cosPhiLine = radius1 * cosphi * phiDenomInv;
    const float TWO_PI = 6.283185307f;                     // M_TWO_PI as the shader spells it
    float a = atan2Det(sinPhiLine, cosPhiLine) + TWO_PI;
    E.phiLine = a - TWO_PI * floorf(a / TWO_PI);           // mod(x, y) = x - y * floor(x / y)
    return E;
}
inline void shadeHitElliptic(const lvo_scene& sc, const lvo_params& P, const Frame& F, float aoTexel, V3 o, V3 d, const Hit& h,
                             float hitColor[4], float& payloadHitT, const RasterQuad* rq = nullptr, const PrebakedAo* pb = nullptr) {
    const lvo_line_point& lp0 = sc.pts[sc.segIdx[2 * h.seg]];
    const lvo_line_point& lp1 = sc.pts[sc.segIdx[2 * h.seg + 1]];
    const EllipticSurface E = ellipticSurface(P, o, d, h.t, lp0, lp1);
    if (pb) {   // STATIC_AMBIENT_OCCLUSION_PREBAKING: getAoFactor(fragmentVertexId, phiLine), EllipticTubeRayTracing.glsl:393-395,420-431
        const float fragmentVertexId = (1.0f - E.t) * float(sc.segIdx[2 * h.seg]) + E.t * float(sc.segIdx[2 * h.seg + 1]);
        aoTexel = prebakedAoLookup(*pb, fragmentVertexId, E.phiLine);
    }
    BandArgs b;
    b.useBand = true;
    b.phi = E.phiLine;
    b.linePosition = E.linePosition;
    b.lineNormal = E.lineNormal;
    if (rq && P.useBands) {
        const V3 tN = normalize(E.tangent);
        const float r = P.bandWidth * 0.5f;
        const float f0 = bandsRibbonOfRay(F.cameraPosition, rq->d0, b.linePosition, b.lineNormal, E.tangent, tN, r, P.minThickness);
        const float fx = bandsRibbonOfRay(F.cameraPosition, rq->dX, b.linePosition, b.lineNormal, E.tangent, tN, r, P.minThickness);
        const float fy = bandsRibbonOfRay(F.cameraPosition, rq->dY, b.linePosition, b.lineNormal, E.tangent, tN, r, P.minThickness);
        b.rasterEpsWhite = fabsf(fx - f0) + fabsf(fy - f0);
    }
    computeFragmentColor(sc, P, F, aoTexel, E.fragPos, E.normal, E.tangent, false, E.attribute, hitColor, payloadHitT,
                         P.useBands ? &b : nullptr);
}

// USE_BANDS halo coordinate of computeFragmentColor (RayHitCommon.glsl:232-351), see the call site
// pH = homogeneous point of the tangent plane (frame coordinates / lineRadius) the line camera -> pH is drawn through: the fragment's
// own point (thickness cos phi, sin phi, 1) in the shader, any point of the viewing ray's trace for bandsRibbonOfRay
inline float bandsRibbonOfPoint(V3 cameraPosition, V3 linePosition, V3 lineNormal, V3 fragmentTangent, V3 t, V3 pH, float lineRadius,
                                float thickness);
inline float bandsRibbonPosition(V3 cameraPosition, V3 linePosition, V3 lineNormal, V3 fragmentTangent, V3 t, float phi,
                                 float lineRadius, float thickness) {
    float sp, cp;
    sincosRad(phi, sp, cp);
    return bandsRibbonOfPoint(cameraPosition, linePosition, lineNormal, fragmentTangent, t, v3(thickness * cp, sp, 1.0f), lineRadius, thickness);
}
// the USE_BANDS coordinate of a VIEWING RAY: its trace in the cross-section plane through linePosition (normal fragmentTangent)
inline float bandsRibbonOfRay(V3 cameraPosition, V3 d, V3 linePosition, V3 lineNormal, V3 fragmentTangent, V3 t, float lineRadius,
                              float thickness) {
    const V3 lineN = normalize(lineNormal);
    const V3 lineB = cross(t, lineN);
    const float s = dot(linePosition - cameraPosition, fragmentTangent) / dot(d, fragmentTangent);
    const V3 wq = (cameraPosition + d * s) - linePosition;
    return bandsRibbonOfPoint(cameraPosition, linePosition, lineNormal, fragmentTangent, t,
                              v3(dot(lineN, wq) / lineRadius, dot(lineB, wq) / lineRadius, 1.0f), lineRadius, thickness);
}
inline float bandsRibbonOfPoint(V3 cameraPosition, V3 linePosition, V3 lineNormal, V3 fragmentTangent, V3 t, V3 pH, float lineRadius,
                                float thickness) {
    const V3 lineN = normalize(lineNormal);
    const V3 lineB = cross(t, lineN);
    const V3 cNorm = cameraPosition - linePosition;
    const float dist = dot(cNorm, fragmentTangent);
    const V3 w = cNorm - dist * fragmentTangent;
    const V3 cHat = v3(dot(lineN, w), dot(lineB, w), dot(t, w)); // transpose(mat3(lineN, lineB, t)) * w
    const V3 c = v3(cHat.x / lineRadius, cHat.y / lineRadius, 1.0f);
    const float a = 1.0f / (thickness * thickness);
    const V3 l = v3(a * c.x, c.y, -1.0f);
    // M_l = shearSymmetricMatrix(l), columns (0, -l.z, l.y), (l.z, 0, -l.x), (-l.y, l.x, 0); B[col][row]
    const float Ml[3][3] = {{0.0f, -l.z, l.y}, {l.z, 0.0f, -l.x}, {-l.y, l.x, 0.0f}};
    const float B[3][3] = {{l.z * l.z - l.y * l.y, l.x * l.y, -l.x * l.z},
                           {l.x * l.y, a * l.z * l.z - l.x * l.x, -a * l.y * l.z},
                           {-l.x * l.z, -a * l.y * l.z, a * l.y * l.y + l.x * l.x}};
    const float EPSILON = 1e-4f;
    float alpha = 0.0f, discr = 0.0f;
    if (fabsf(l.z) > EPSILON) {
        discr = -B[0][0] * B[1][1] + B[0][1] * B[1][0];
        alpha = sqrtf(discr) / l.z;
    } else if (fabsf(l.y) > EPSILON) {
        discr = -B[0][0] * B[2][2] + B[0][2] * B[2][0];
        alpha = sqrtf(discr) / l.y;
    } else if (fabsf(l.x) > EPSILON) {
        discr = -B[1][1] * B[2][2] + B[1][2] * B[2][1];
        alpha = sqrtf(discr) / l.x;
    }
    float Cm[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Cm[i][j] = B[i][j] + alpha * Ml[i][j];
    float pm0x = 0.0f, pm0y = 0.0f, pm1x = 0.0f, pm1y = 0.0f;
    for (int i = 0; i < 2; ++i) {
        if (fabsf(Cm[i][i]) > EPSILON) {
            pm0x = Cm[i][0] / Cm[i][2]; pm0y = Cm[i][1] / Cm[i][2];   // column i
            pm1x = Cm[0][i] / Cm[2][i]; pm1y = Cm[1][i] / Cm[2][i];   // row i
        }
    }
    const V3 pLineH = cross(l, cross(c, pH));
    const float plx = pLineH.x / pLineH.z, ply = pLineH.y / pLineH.z;
    const float num = sqrtf((plx - pm0x) * (plx - pm0x) + (ply - pm0y) * (ply - pm0y));
    const float den = sqrtf((pm1x - pm0x) * (pm1x - pm0x) + (pm1y - pm0y) * (pm1y - pm0y));
    return num / den * 2.0f - 1.0f;
}

// computeFragmentColor (RayHitCommon.glsl:74-543) for tubes: shared by the analytic and the triangle closest-hit shaders
inline void computeFragmentColor(const lvo_scene& sc, const lvo_params& P, const Frame& F, float aoTexel, V3 fragPos,
                                 V3 fragmentNormal, V3 fragmentTangent, bool isCap, float fragmentAttribute,
                                 float hitColor[4], float& payloadHitT, const BandArgs* bandsIn, const HelicityArgs* hel) {
    const BandArgs* bands = (bandsIn && bandsIn->shadeBands) ? bandsIn : nullptr;   // USE_BANDS shading arguments, if any
    const float rasterEpsWhite = bandsIn ? bandsIn->rasterEpsWhite : -1.0f;         // >= 0: raster variant of the outline
    float fragmentColor[4];
    transferFunction(sc, P, fragmentAttribute, fragmentColor);
    V3 n = normalizeShade(fragmentNormal);
    V3 vv = normalizeShade(F.cameraPosition - fragPos);
    V3 t = normalizeShade(fragmentTangent);
    V3 helperVec = normalizeShade(cross(t, vv));
    V3 newV = normalizeShade(cross(helperVec, t));

    float ribbonPosition = 0.0f;
    if (P.useHalos) {
        if (P.useCappedTubes && isCap) {
            // RayHitCommon.glsl:195-229
            V3 crossProdVn = cross(vv, n);
            ribbonPosition = length(crossProdVn);
            V3 crossProdVn2 = cross(newV, n);
            float ribbonPosition2 = length(crossProdVn2);
            if (dot(t, crossProdVn) < 0.0f) ribbonPosition2 = -ribbonPosition2;
            if (dot(t, crossProdVn) < 0.0f) ribbonPosition = -ribbonPosition;
            ribbonPosition2 = clampf(ribbonPosition2, -1.0f, 1.0f);
            if (fabsf(ribbonPosition2) < fabsf(ribbonPosition)) ribbonPosition = ribbonPosition2;
            // raster variant, LinePassGeometryShaderTubes.glsl:785-815: ribbonPosition = min(length(cross(v, n)), abs(ribbonPosition2))
            if (rasterEpsWhite >= 0.0f) ribbonPosition = fminf(length(crossProdVn), fabsf(ribbonPosition2));
        } else if (bands) {
            // USE_BANDS, RayHitCommon.glsl:232-351: the fragment's position between the two silhouette points of the elliptic
            // cross-section as the camera sees it -- tangent-plane coordinates, polar line of the camera point with respect to the
            // conic x^2 / thickness^2 + y^2 = 1, its two intersections with the conic from the degenerate conic B + alpha M_l
            ribbonPosition = bandsRibbonPosition(F.cameraPosition, bands->linePosition, bands->lineNormal, fragmentTangent, t, bands->phi,
                                                 (bands->useBand ? P.bandWidth : P.lineWidth) * 0.5f,
                                                 bands->useBand ? P.minThickness : 1.0f);
        } else {
            // RayHitCommon.glsl:353-372
            V3 crossProdVn = cross(newV, n);
            ribbonPosition = length(crossProdVn);
            if (dot(t, crossProdVn) < 0.0f) ribbonPosition = -ribbonPosition;
            ribbonPosition = clampf(ribbonPosition, -1.0f, 1.0f);
        }
    }

    V3 ssp = v3(0, 0, 0);
    if (P.useDepthCues || P.useAmbientOcclusion) {
        V4 s4 = mulM4(P.view, V4{fragPos.x, fragPos.y, fragPos.z, 1.0f});
        ssp = v3(s4.x, s4.y, s4.z);
    }

    float shaded[4];
    blinnPhongShadingTube(P, F, aoTexel, fragmentColor, fragPos, ssp, n, t, shaded, (bands && bands->useBand) ? 1.0f : 1.7f);

    float absCoords = P.useHalos ? fabsf(ribbonPosition) : 0.0f;
    float fragmentDepth = length(fragPos - F.cameraPosition);
    // Antialiasing.glsl:1-3; RayHitCommon.glsl:445-452 (USE_BANDS: both epsilons from depth / width * 0.25)
    float aaO = ((fragmentDepth / P.lineWidth) * 0.05f) / float(P.height) * P.fovY;
    float aaW = ((fragmentDepth / P.lineWidth) * 2.0f) / float(P.height) * P.fovY;
    if (bands) {
        const float wdt = bands->useBand ? P.bandWidth : P.lineWidth;
        aaO = aaW = ((fragmentDepth / wdt) * 0.25f) / float(P.height) * P.fovY;
    }
    float EPSILON_OUTLINE = clampf(aaO, 0.0f, 0.49f);
    float EPSILON_WHITE = clampf(aaW, 0.0f, 0.49f);
    float WHITE_THRESHOLD = 0.7f;
    if (hel) {
        // RayHitCommon.glsl:455-486 (no multi-var rendering, no twist-line texture, no UNIFORM_HELICITY_BAND_WIDTH):
        // drawSeparatorStripe (:57-64) darkens the shaded colour where mod(phi + rotation + w / 2, 2 pi / n) falls into [0, w]
        const float separatorWidth = P.separatorBaseWidth / hel->rotationSeparatorScale; // :456-459 (scale 1 without the define)
        const float period = 2.0f / float(P.numSubdivisionsBands) * 3.14159265358979323846f;
        const bool rasterStripe = hel->rasterAaf >= 0.0f;
        const float x = hel->phi + hel->fragmentRotation + (rasterStripe ? 0.1f : separatorWidth * 0.5f);
        const float varFraction = x - period * floorf(x / period); // mod(x, y) = x - y * floor(x / y)
        const float aaf = rasterStripe ? hel->rasterAaf : EPSILON_OUTLINE * 10.0f;
        if (g_twist.use) {
            // USE_HELICITY_BANDS_TEXTURE (RayHitCommon.glsl:465-469, LinePassGeometryShaderTubes.glsl:1043-1047): fragmentColor *= texture
            const float twoPi = 2.0f * 3.14159265358979323846f;
            const float tu = (x - twoPi * floorf(x / twoPi)) / twoPi;
            float tex[4];
            twistSample(tu, hel->dx, hel->dy, rasterStripe, tex);
            for (int k = 0; k < 3; k++) shaded[k] = shaded[k] * tex[k];
            shaded[3] = shaded[3] * tex[3];
        } else {
        const float alphaBorder1 = smoothstepf(aaf, 0.0f, varFraction);
        const float alphaBorder2 = smoothstepf(separatorWidth - aaf * 0.5f, separatorWidth + aaf * 0.5f, varFraction);
        const float m = fmaxf(alphaBorder1, alphaBorder2);
        for (int k = 0; k < 3; k++) shaded[k] = shaded[k] * m;
        }
        WHITE_THRESHOLD = 0.8f; // :485-486
    }
    if (rasterEpsWhite >= 0.0f) {
        // LinePassGeometryShaderTubes.glsl:1079-1087: EPSILON_OUTLINE = 0.0, EPSILON_WHITE = fwidth(ribbonPosition); no clamps.
        // (The separator stripes above keep the ray tracer's width: the raster shader's own drawSeparatorStripe differs and the
        // helicity bands are outside SURVEY.md 8's a9.)  smoothstep(1, 1, x) divides by zero: x < 1 -> -inf -> 0; x == 1 -> NaN ->
        // 0 through fmaxf(NaN, 0) = 0: coverage = 1 on the whole tube, i.e. a hard edge at the silhouette.
        EPSILON_OUTLINE = 0.0f;
        EPSILON_WHITE = P.useHalos ? rasterEpsWhite : 0.0f;
    }
    float coverage = P.useHalos ? 1.0f - smoothstepf(1.0f - EPSILON_OUTLINE, 1.0f, absCoords) : 1.0f;
    if (bands && bands->useBand && P.useEllipticTubes) coverage = 1.0f; // ANALYTIC_ELLIPTIC_TUBE_INTERSECTIONS, :499-504
    float w = smoothstepf(WHITE_THRESHOLD - EPSILON_WHITE, WHITE_THRESHOLD + EPSILON_WHITE, absCoords);
    for (int k = 0; k < 3; k++) hitColor[k] = mixf(shaded[k], F.foreground[k], w);
    hitColor[3] = shaded[3] * coverage;
    payloadHitT = length(fragPos - F.cameraPosition);
}

inline uint8_t toUnorm8(float c) { return uint8_t(floorf(clampf(c, 0.0f, 1.0f) * 255.0f + 0.5f)); }
inline uint32_t packUnorm4x8(const float c[4]) {
    return uint32_t(toUnorm8(c[0])) | (uint32_t(toUnorm8(c[1])) << 8) | (uint32_t(toUnorm8(c[2])) << 16)
           | (uint32_t(toUnorm8(c[3])) << 24);
}
inline void unpackUnorm4x8(uint32_t p, float c[4]) {
    c[0] = float(p & 0xFFu) / 255.0f;
    c[1] = float((p >> 8) & 0xFFu) / 255.0f;
    c[2] = float((p >> 16) & 0xFFu) / 255.0f;
    c[3] = float((p >> 24) & 0xFFu) / 255.0f;
}

inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

inline void padTiling(uint32_t& w, uint32_t& h, uint32_t tw, uint32_t th) {
    // LineRenderer::getScreenSizeWithTiling, LineRenderer.cpp:805-812
    if (w % tw != 0) w = (w / tw + 1) * tw;
    if (h % th != 0) h = (h / th + 1) * th;
}

// 4-ary min-heap on (depth[, colour]) ; LinkedListSort.glsl:177-205
template <bool LITERAL>
inline bool gt(const uint32_t* col, const float* dep, uint32_t a, uint32_t b) {
    if (LITERAL) return dep[a] > dep[b];
    return dep[a] > dep[b] || (dep[a] == dep[b] && col[a] > col[b]);
}
template <bool LITERAL>
inline void minHeapSink4(uint32_t* col, float* dep, uint32_t x, uint32_t fragsCount) {
    uint32_t c, t;
    while ((t = 4 * x + 1) < fragsCount) {
        if (t + 1 < fragsCount && gt<LITERAL>(col, dep, t, t + 1)) c = t + 1; else c = t;
        if (t + 2 < fragsCount && gt<LITERAL>(col, dep, c, t + 2)) c = t + 2;
        if (t + 3 < fragsCount && gt<LITERAL>(col, dep, c, t + 3)) c = t + 3;
        if (!gt<LITERAL>(col, dep, x, c)) return;
        std::swap(col[x], col[c]);
        std::swap(dep[x], dep[c]);
        x = c;
    }
}
// frontToBackPQ, LinkedListSort.glsl:207-238
template <bool LITERAL>
inline void frontToBackPQ(uint32_t* col, float* dep, uint32_t fragsCount, float out[4]) {
    uint32_t i;
    for (i = fragsCount / 4; i > 0; --i) minHeapSink4<LITERAL>(col, dep, i, fragsCount);
    float ray[4] = {0, 0, 0, 0};
    i = 0;
    while (i < fragsCount && ray[3] < 0.99f) {
        minHeapSink4<LITERAL>(col, dep, 0, fragsCount - i++);
        float src[4];
        unpackUnorm4x8(col[0], src);
        for (int k = 0; k < 3; k++) ray[k] = ray[k] + ((1.0f - ray[3]) * src[3]) * src[k];
        ray[3] = ray[3] + (1.0f - ray[3]) * src[3];
        col[0] = col[fragsCount - i];
        dep[0] = dep[fragsCount - i];
    }
    out[0] = ray[0] / ray[3]; out[1] = ray[1] / ray[3]; out[2] = ray[2] / ray[3]; out[3] = ray[3];
}


// ---- the other sorting modes of the resolve pass (SORTING_MODE_NAMES, src/Renderers/PPLL.hpp:32-50): each sorts the whole
// list and blends ALL fragments front to back (blendFTB, LinkedListSort.glsl:45-59 -- no early out at alpha 0.99, unlike the
// priority queue).  LITERAL compares depths only, like the shaders; otherwise the (depth, colour) key of the build (section 5).
struct FragList {
    uint32_t* col;
    float* dep;
};
template <bool LITERAL>
inline bool keyLess(float da, uint32_t ca, float db, uint32_t cb) {
    if (LITERAL) return da < db;
    return da < db || (da == db && ca < cb);
}
inline void swapFragments(FragList L, uint32_t i, uint32_t j) { // LinkedListSort.glsl:35-43
    std::swap(L.col[i], L.col[j]);
    std::swap(L.dep[i], L.dep[j]);
}
inline void blendFTB(FragList L, uint32_t fragsCount, float out[4]) {
    float color[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < fragsCount; i++) {
        float src[4];
        unpackUnorm4x8(L.col[i], src);
        for (int k = 0; k < 3; k++) color[k] = color[k] + ((1.0f - color[3]) * src[3]) * src[k];
        color[3] = color[3] + (1.0f - color[3]) * src[3];
    }
    out[0] = color[0] / color[3]; out[1] = color[1] / color[3]; out[2] = color[2] / color[3]; out[3] = color[3];
}
template <bool LITERAL>
inline void bubbleSort(FragList L, uint32_t n) { // LinkedListSort.glsl:62-77
    bool changed;
    do {
        changed = false;
        for (uint32_t i = 0; i + 1 < n; ++i)
            if (gt<LITERAL>(L.col, L.dep, i, i + 1)) { swapFragments(L, i, i + 1); changed = true; }
    } while (changed);
}
template <bool LITERAL>
inline void gapInsertionPass(FragList L, uint32_t n, uint32_t gap) { // the loop insertionSort (gap 1) and shellSort share
    for (uint32_t i = gap; i < n; ++i) {
        const uint32_t fragColor = L.col[i];
        const float fragDepth = L.dep[i];
        uint32_t j = i;
        while (j >= gap && keyLess<LITERAL>(fragDepth, fragColor, L.dep[j - gap], L.col[j - gap])) {
            L.col[j] = L.col[j - gap];
            L.dep[j] = L.dep[j - gap];
            j -= gap;
        }
        L.col[j] = fragColor;
        L.dep[j] = fragDepth;
    }
}
template <bool LITERAL>
inline void insertionSort(FragList L, uint32_t n) { gapInsertionPass<LITERAL>(L, n, 1u); } // :80-104
template <bool LITERAL>
inline void shellSort(FragList L, uint32_t n) { // :107-137, gap sequence 24, 9, 4, 1
    const uint32_t gaps[4] = {24u, 9u, 4u, 1u};
    for (uint32_t g = 0; g < 4; g++) gapInsertionPass<LITERAL>(L, n, gaps[g]);
}
template <bool LITERAL>
inline void maxHeapSink(FragList L, uint32_t x, uint32_t n) { // :140-157
    uint32_t c;
    while ((c = 2 * x + 1) < n) {
        if (c + 1 < n && gt<LITERAL>(L.col, L.dep, c + 1, c)) ++c;
        if (!gt<LITERAL>(L.col, L.dep, c, x)) return; // depth[x] >= depth[c]
        swapFragments(L, x, c);
        x = c;
    }
}
template <bool LITERAL>
inline void heapSort(FragList L, uint32_t n) { // :159-172
    for (uint32_t i = (n + 1) / 2; i > 0; --i) maxHeapSink<LITERAL>(L, i - 1, n);
    for (uint32_t i = 1; i < n; ++i) {
        swapFragments(L, 0, n - i);
        maxHeapSink<LITERAL>(L, 0, n - i);
    }
}
template <bool LITERAL>
inline void bitonicSort(FragList L, uint32_t n) { // :241-262 -- as written: comparators that would reach past the list are
    // skipped and the network stops at the largest power of two <= n, so a list whose length is no power of two stays partly
    // unsorted (restated, not repaired)
    for (uint32_t k = 2; k <= n; k *= 2)
        for (uint32_t j = k / 2; j > 0; j /= 2)
            for (uint32_t i = 0; i < n; i++) {
                const uint32_t l = i ^ j;
                if (l > i && l < n) {
                    const bool up = (i & k) == 0;
                    if ((up && gt<LITERAL>(L.col, L.dep, i, l)) || (!up && gt<LITERAL>(L.col, L.dep, l, i))) swapFragments(L, i, l);
                }
            }
}
// LinkedListQuicksort.glsl:30-55: a stack of STACK_SIZE ints that ignores pushes when full and pops 0 when empty
struct SortStack {
    int mem[64];
    int size, counter = 0;
    void push(int v) { if (counter < size) mem[counter++] = v; }
    int pop() { return counter > 0 ? mem[--counter] : 0; }
    bool empty() const { return counter == 0; }
};
template <bool LITERAL>
inline int partitionLomuto(FragList L, int low, int high) { // :57-68
    const float pd = L.dep[high];
    const uint32_t pc = L.col[high];
    int i = low;
    for (int j = low; j <= high; j++)
        if (keyLess<LITERAL>(L.dep[j], L.col[j], pd, pc)) { swapFragments(L, uint32_t(i), uint32_t(j)); i++; }
    swapFragments(L, uint32_t(i), uint32_t(high));
    return i;
}
template <bool LITERAL>
inline int partitionHoare(FragList L, int low, int high) { // :70-91, pivot = median of first / middle / last
    const int mid = (low + high) / 2;
    auto lt = [&](int a, int b) { return keyLess<LITERAL>(L.dep[a], L.col[a], L.dep[b], L.col[b]); };
    auto mn = [&](int a, int b) { return lt(b, a) ? b : a; };
    const int p = lt(low, mid) ? (lt(high, low) ? low : mn(mid, high)) : (lt(high, mid) ? mid : mn(low, high));
    const float pd = L.dep[p];
    const uint32_t pc = L.col[p];
    int i = low - 1, j = high + 1;
    for (;;) {
        do { i = i + 1; } while (keyLess<LITERAL>(L.dep[i], L.col[i], pd, pc));
        do { j = j - 1; } while (keyLess<LITERAL>(pd, pc, L.dep[j], L.col[j]));
        if (i >= j) return j;
        swapFragments(L, uint32_t(i), uint32_t(j));
    }
}
inline int sortStackSize(uint32_t maxNumFrags) { // PerPixelLinkedListLineRenderer.cpp:178
    const int s = int(std::ceil(std::log2(double(maxNumFrags))) * 2 + 4);
    return s < 0 ? 0 : (s > 64 ? 64 : s);
}
template <bool LITERAL>
inline void quicksort(FragList L, uint32_t n, uint32_t maxNumFrags) { // :93-114
    SortStack st;
    st.size = sortStackSize(maxNumFrags);
    st.push(0);
    st.push(int(n) - 1);
    while (!st.empty()) {
        const int high = st.pop(), low = st.pop();
        const int pivot = partitionLomuto<LITERAL>(L, low, high);
        if (low < pivot - 1) { st.push(low); st.push(pivot - 1); }
        if (pivot + 1 < high) { st.push(pivot + 1); st.push(high); }
    }
}
template <bool LITERAL>
inline void quicksortHybrid(FragList L, uint32_t n, uint32_t maxNumFrags) { // :116-141
    SortStack st;
    st.size = sortStackSize(maxNumFrags);
    st.push(0);
    st.push(int(n) - 1);
    if (n > 16)
        while (!st.empty()) {
            const int high = st.pop(), low = st.pop();
            const int pivot = partitionHoare<LITERAL>(L, low, high);
            if (low + 16 < pivot) { st.push(low); st.push(pivot - 1); }
            if (pivot + 16 < high) { st.push(pivot + 1); st.push(high); }
        }
    insertionSort<LITERAL>(L, n);
}
template <bool LITERAL>
inline void sortAndBlend(uint32_t mode, uint32_t* col, float* dep, uint32_t n, uint32_t maxNumFrags, float out[4]) {
    FragList L{col, dep};
    switch (mode) {
        case 1: bubbleSort<LITERAL>(L, n); break;
        case 2: insertionSort<LITERAL>(L, n); break;
        case 3: shellSort<LITERAL>(L, n); break;
        case 4: heapSort<LITERAL>(L, n); break;
        case 5: bitonicSort<LITERAL>(L, n); break;
        case 6: quicksort<LITERAL>(L, n, maxNumFrags); break;
        default: quicksortHybrid<LITERAL>(L, n, maxNumFrags); break;
    }
    blendFTB(L, n, out);
}

#include "lv_oracle_prism.h"

} // namespace

// ================================================================ exported API
extern "C" {

void lvo_set_twist_line_texture(const uint8_t* rgba8, uint32_t width, uint32_t height, uint32_t filterMode) {
    g_twist.use = rgba8 != nullptr;
    g_twist.texels.clear();
    if (!rgba8) return;
    g_twist.w = width; g_twist.h = height; g_twist.mode = filterMode;
    uint32_t levels = 0;
    for (uint32_t m = std::max(width, height); m > 1; m >>= 1) levels++;
    if (levels == 0) levels = 1;
    g_twist.levels = levels;
    std::vector<float>& tex = g_twist.texels;
    for (size_t i = 0; i < size_t(width) * height * 4; i++) tex.push_back(float(rgba8[i]) / 255.0f);
    size_t prev = 0;
    uint32_t w = width, h = height;
    for (uint32_t l = 1; l < levels; l++) {
        const uint32_t nw = std::max(w >> 1, 1u), nh = std::max(h >> 1, 1u);
        const size_t cur = tex.size();
        tex.resize(cur + size_t(nw) * nh * 4);
        for (uint32_t j = 0; j < nh; j++)
            for (uint32_t i = 0; i < nw; i++)
                for (uint32_t c = 0; c < 4; c++) {
                    const uint32_t i0 = std::min(2 * i, w - 1), i1 = std::min(2 * i + 1, w - 1), j0 = std::min(2 * j, h - 1), j1 = std::min(2 * j + 1, h - 1);
                    const float a = tex[prev + (size_t(j0) * w + i0) * 4 + c], b = tex[prev + (size_t(j0) * w + i1) * 4 + c];
                    const float cc = tex[prev + (size_t(j1) * w + i0) * 4 + c], d = tex[prev + (size_t(j1) * w + i1) * 4 + c];
                    tex[cur + (size_t(j) * nw + i) * 4 + c] = ((a + b) + (cc + d)) * 0.25f;
                }
        prev = cur; w = nw; h = nh;
    }
}
void lvo_twist_line_sample(const float* u, const float* dudx, const float* dudy, int useGrad, uint64_t n, float* outRGBA) {
    for (uint64_t i = 0; i < n; i++) twistSample(u[i], dudx[i], dudy[i], useGrad != 0, outRGBA + 4 * i);
}


uint32_t lvo_tea(uint32_t a, uint32_t b) { return tea(a, b); }
uint32_t lvo_lcg(uint32_t* s) { return lcg(*s); }
float lvo_rnd(uint32_t* s) { return rnd(*s); }
void lvo_sincos_2pi(float xi, float* s, float* c) { sincos2pi(xi, *s, *c); }
void lvo_sincos_rad(float a, float* s, float* c) { sincosRad(a, *s, *c); }
float lvo_atan2_det(float y, float x) { return atan2Det(y, x); }
// test hook: the USE_BANDS halo coordinate alone (tests/test_bands.py compares it with a geometric float64 construction)
float lvo_bands_ribbon_position(const float cam[3], const float linePos[3], const float lineNormal[3], const float tangent[3],
                                float phi, float lineRadius, float thickness) {
    return bandsRibbonPosition(ld3(cam), ld3(linePos), ld3(lineNormal), ld3(tangent), normalize(ld3(tangent)), phi, lineRadius, thickness);
}
void lvo_mat4_inverse(const float m[16], float out[16]) { mat4Inverse(m, out); }
unsigned long long lvo_shade_normalize_out_of_range(int reset) {
    const unsigned long long n = g_shadeNormalizeOutOfRange.load();
    if (reset) g_shadeNormalizeOutOfRange.store(0ull);
    return n;
}

void lvo_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

// TrajectoryFile.cpp:106-125 (AABB over all points, then v = (v + translation) * scale)
void lvo_normalize_positions(float* p, uint64_t n) {
    if (n == 0) return;
    float mn[3] = {p[0], p[1], p[2]}, mx[3] = {p[0], p[1], p[2]};
    for (uint64_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], p[3 * i + k]); mx[k] = fmaxf(mx[k], p[3 * i + k]); }
    float tr[3], sc3[3];
    for (int k = 0; k < 3; k++) { tr[k] = -((mn[k] + mx[k]) / 2.0f); sc3[k] = 0.5f / (mx[k] - mn[k]); }
    float scale = std::min(sc3[0], std::min(sc3[1], sc3[2]));
    for (uint64_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) p[3 * i + k] = (p[3 * i + k] + tr[k]) * scale;
}

static const float* g_helicities = nullptr;
static float g_maxHelicity = 1.0f;
void lvo_set_helicity_source(const float* helicities, float maxHelicity) {
    g_helicities = helicities;
    g_maxHelicity = maxHelicity;
}
const float* lvo_get_helicity_source(float* maxHelicity) {
    *maxHelicity = g_maxHelicity;
    return g_helicities;
}

// LineDataFlow.cpp:2112-2277
static void buildTubeAabbRenderData(
        const float* positions, const float* attributes, const uint32_t* lineOffsets, uint32_t nLines, float lineWidth,
        const float* ribbonDirections, lvo_line_point* outPoints, uint32_t* outNumPoints, uint32_t* outSegIndices,
        float* outAabbs, uint32_t* outNumSegments);
void lvo_build_tube_aabb_render_data(
        const float* positions, const float* attributes, const uint32_t* lineOffsets, uint32_t nLines, float lineWidth,
        lvo_line_point* outPoints, uint32_t* outNumPoints, uint32_t* outSegIndices, float* outAabbs,
        uint32_t* outNumSegments) {
    buildTubeAabbRenderData(positions, attributes, lineOffsets, nLines, lineWidth, nullptr, outPoints, outNumPoints,
                            outSegIndices, outAabbs, outNumSegments);
}
// getLinePassTubeAabbRenderData(false, ellipticTubes = true) with band data (useRibbonNormals, LineDataFlow.cpp:2120-2126,
// 2166-2168): the line normal is cross(ribbon direction, tangent) -- not normalised -- and the boxes are padded by bandWidth / 2.
void lvo_build_tube_aabb_render_data_ribbons(
        const float* positions, const float* attributes, const uint32_t* lineOffsets, uint32_t nLines, float bandWidth,
        const float* ribbonDirections, lvo_line_point* outPoints, uint32_t* outNumPoints, uint32_t* outSegIndices,
        float* outAabbs, uint32_t* outNumSegments) {
    buildTubeAabbRenderData(positions, attributes, lineOffsets, nLines, bandWidth, ribbonDirections, outPoints, outNumPoints,
                            outSegIndices, outAabbs, outNumSegments);
}
static void buildTubeAabbRenderData(
        const float* positions, const float* attributes, const uint32_t* lineOffsets, uint32_t nLines, float lineWidth,
        const float* ribbonDirections, lvo_line_point* outPoints, uint32_t* outNumPoints, uint32_t* outSegIndices,
        float* outAabbs, uint32_t* outNumSegments) {
    float lwo = lineWidth * 0.5f;
    uint32_t nOut = 0, nSeg = 0;
    uint32_t lineSegmentIndexCounter = 0;
    for (uint32_t li = 0; li < nLines; li++) {
        uint32_t b = lineOffsets[li], e = lineOffsets[li + 1];
        uint32_t n = e - b;
        V3 lastLineNormal = v3(1.0f, 0.0f, 0.0f);
        uint32_t numValidLinePoints = 0;
        float rotation = 0.0f; // useRotatingHelicityBands: restarts with every trajectory (:2148)
        for (uint32_t i = 0; i < n; i++) {
            if (n < 2) break; // a one-point trajectory has no neighbour to difference against
            V3 tangent;
            V3 pi = ld3(positions + 3 * (b + i));
            if (i == 0) tangent = ld3(positions + 3 * (b + i + 1)) - pi;
            else if (i + 1 == n) tangent = pi - ld3(positions + 3 * (b + i - 1));
            else tangent = ld3(positions + 3 * (b + i + 1)) - ld3(positions + 3 * (b + i - 1));
            float tangentLength = length(tangent);
            if (tangentLength < 0.0001f) continue;
            tangent = normalize(tangent);
            V3 helperAxis = lastLineNormal;
            if (length(cross(helperAxis, tangent)) < 0.01f) {
                helperAxis = v3(0.0f, 1.0f, 0.0f);
                if (length(cross(helperAxis, tangent)) < 0.01f) helperAxis = v3(0.0f, 0.0f, 1.0f);
            }
            V3 normal = normalize(helperAxis - dot(helperAxis, tangent) * tangent);
            if (ribbonDirections) normal = cross(ld3(ribbonDirections + 3 * (b + i)), tangent);
            lastLineNormal = normal;
            lvo_line_point lp;
            memset(&lp, 0, sizeof(lp));
            lp.linePosition[0] = pi.x; lp.linePosition[1] = pi.y; lp.linePosition[2] = pi.z;
            lp.lineAttribute = attributes[b + i];
            lp.lineTangent[0] = tangent.x; lp.lineTangent[1] = tangent.y; lp.lineTangent[2] = tangent.z;
            lp.lineNormal[0] = normal.x; lp.lineNormal[1] = normal.y; lp.lineNormal[2] = normal.z;
            if (g_helicities) { // :2188-2197 (a point skipped above does not advance the rotation)
                lp.lineRotation = rotation;
                const float helicity = g_helicities[b + i];
                float lineSegmentLength = 0.0f;
                if (i < n - 1) lineSegmentLength = length(ld3(positions + 3 * (b + i + 1)) - pi);
                rotation += helicity / g_maxHelicity * 3.1415926535897932f * lineSegmentLength / 0.005f;
            }
            outPoints[nOut++] = lp;
            numValidLinePoints++;
        }
        if (numValidLinePoints == 1) nOut--;
        if (numValidLinePoints <= 1) continue;
        for (uint32_t pointIdx = 1; pointIdx < numValidLinePoints; pointIdx++) {
            uint32_t a0 = lineSegmentIndexCounter + pointIdx - 1, a1 = lineSegmentIndexCounter + pointIdx;
            outSegIndices[2 * nSeg] = a0;
            outSegIndices[2 * nSeg + 1] = a1;
            const float* p0 = outPoints[a0].linePosition;
            const float* p1 = outPoints[a1].linePosition;
            for (int k = 0; k < 3; k++) {
                outAabbs[6 * nSeg + k] = fminf(p0[k], p1[k]) - lwo;
                outAabbs[6 * nSeg + 3 + k] = fmaxf(p0[k], p1[k]) + lwo;
            }
            nSeg++;
        }
        lineSegmentIndexCounter += numValidLinePoints;
    }
    *outNumPoints = nOut;
    *outNumSegments = nSeg;
}

lvo_scene* lvo_scene_create(const lvo_line_point* pts, uint32_t nPts, const uint32_t* segIdx, uint32_t nSeg) {
    lvo_scene* sc = new lvo_scene();
    sc->pts.assign(pts, pts + nPts);
    sc->segIdx.assign(segIdx, segIdx + 2 * size_t(nSeg));
    sc->nSeg = nSeg;
    return sc;
}
void lvo_scene_destroy(lvo_scene* sc) { delete sc; }
void lvo_scene_set_tf(lvo_scene* sc, const float* rgba, uint32_t n) {
    sc->tf.assign(rgba, rgba + 4 * size_t(n));
    sc->tfN = n;
}

void lvo_scene_build_bvh(lvo_scene* sc, float lineWidth) {
    sc->nodes.clear();
    sc->root = -1;
    sc->rootIsLeaf = false;
    sc->bvhDepth = 0;
    sc->bvhLineWidth = lineWidth;
    uint32_t n = sc->nSeg;
    if (n == 0) return;
    float r = lineWidth * 0.5f;
    float pad = r * 1e-3f + 1e-6f;
    std::vector<SegBox> boxes(n);
    float smn[3] = {3e38f, 3e38f, 3e38f}, smx[3] = {-3e38f, -3e38f, -3e38f};
    for (uint32_t s = 0; s < n; s++) {
        V3 p0, p1; segPoints(*sc, s, p0, p1);
        const float a[3] = {p0.x, p0.y, p0.z}, b[3] = {p1.x, p1.y, p1.z};
        for (int k = 0; k < 3; k++) {
            boxes[s].mn[k] = fminf(a[k], b[k]) - r - pad;
            boxes[s].mx[k] = fmaxf(a[k], b[k]) + r + pad;
            smn[k] = fminf(smn[k], boxes[s].mn[k]);
            smx[k] = fmaxf(smx[k], boxes[s].mx[k]);
        }
    }
    std::vector<uint64_t> keys(n);
    std::vector<uint32_t> order(n);
    for (uint32_t s = 0; s < n; s++) {
        uint64_t q[3];
        for (int k = 0; k < 3; k++) {
            double c = 0.5 * (double(boxes[s].mn[k]) + double(boxes[s].mx[k]));
            double u = (c - smn[k]) / std::max(1e-30, double(smx[k]) - double(smn[k]));
            u = std::min(std::max(u, 0.0), 1.0);
            q[k] = uint64_t(std::min(2097151.0, u * 2097152.0));
        }
        keys[s] = (expandBits21(q[0]) << 2) | (expandBits21(q[1]) << 1) | expandBits21(q[2]);
        order[s] = s;
    }
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        return keys[a] < keys[b] || (keys[a] == keys[b] && a < b);
    });
    std::vector<uint64_t> sk(n);
    for (uint32_t i = 0; i < n; i++) sk[i] = keys[order[i]];
    sc->nodes.reserve(n);
    uint32_t maxDepth = 0;
    sc->root = buildRange(*sc, sk, order, boxes, 0, n, 0, maxDepth);
    sc->rootIsLeaf = sc->root < 0;
    sc->bvhDepth = maxDepth;
    sc->leafBoxes.resize(6 * size_t(n));
    for (uint32_t s = 0; s < n; s++)
        for (int k = 0; k < 3; k++) { sc->leafBoxes[6 * size_t(s) + k] = boxes[s].mn[k]; sc->leafBoxes[6 * size_t(s) + 3 + k] = boxes[s].mx[k]; }
}

int lvo_intersect_capsule(const float o[3], const float d[3], const float p0[3], const float p1[3], float radius,
                          int capped, float* outT, int* outKind) {
    float t; int k;
    bool h = intersectCapsule(ld3(o), ld3(d), ld3(p0), ld3(p1), radius, capped != 0, t, k);
    *outT = t; *outKind = k;
    return h ? 1 : 0;
}

int lvo_intersect_capsule_literal(const float o[3], const float d[3], const float p0[3], const float p1[3], float radius,
                                  int capped, float* outT, int* outKind) {
    float t; int k;
    bool h = intersectCapsuleLiteral(ld3(o), ld3(d), ld3(p0), ld3(p1), radius, capped != 0, t, k);
    *outT = t; *outKind = k;
    return h ? 1 : 0;
}

void lvo_trace_rays(const lvo_scene* sc, float lineWidth, int capped, int useBvh, const float* origins,
                    const float* dirs, float tMin, float tMax, uint32_t n, float* outT, uint32_t* outSeg,
                    uint32_t* outKind) {
    float radius = lineWidth * 0.5f;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < int64_t(n); i++) {
        Counters c;
        Hit h;
        bool f = closestHit(*sc, radius, capped != 0, useBvh != 0, ld3(origins + 3 * i), ld3(dirs + 3 * i), tMin, tMax, h, c);
        outT[i] = f ? h.t : tMax;
        outSeg[i] = f ? h.seg : 0xFFFFFFFFu;
        outKind[i] = f ? uint32_t(h.kind) : 0u;
    }
}

// Closest hit on the elliptic tubelets (IntersectionEllipticTube); cameraPosition enters the cutting-plane tolerances.
void lvo_trace_rays_elliptic(const lvo_scene* sc, float bandWidth, float minBandThickness, const float* cameraPosition, int useBvh,
                             const float* origins, const float* dirs, float tMin, float tMax, uint32_t n, float* outT,
                             uint32_t* outSeg) {
    EllipticScope ellScope(true, bandWidth, minBandThickness, ld3(cameraPosition));
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < int64_t(n); i++) {
        Counters c;
        Hit h;
        bool f = closestHit(*sc, bandWidth * 0.5f, true, useBvh != 0, ld3(origins + 3 * i), ld3(dirs + 3 * i), tMin, tMax, h, c);
        outT[i] = f ? h.t : tMax;
        outSeg[i] = f ? h.seg : 0xFFFFFFFFu;
    }
}

// ComputeDepthValues.glsl:58-98 + MinMaxReduce.glsl:64-103 (min/max are order independent)
void lvo_compute_depth_range(const lvo_scene* sc, const lvo_params* P, float outMinMax[2]) {
    float mn = P->farDist, mx = P->nearDist;
    const float EPSILON = 1e-2f;
    for (size_t i = 0; i < sc->pts.size(); i++) {
        const float* p = sc->pts[i].linePosition;
        V4 ssp = mulM4(P->view, V4{p[0], p[1], p[2], 1.0f});
        V4 ndc = mulM4(P->proj, ssp);
        float nx = ndc.x / ndc.w, ny = ndc.y / ndc.w, nz = ndc.z / ndc.w;
        if (nx >= -1.0f && ny >= -1.0f && nz >= -1.0f && nx <= 1.0f && ny <= 1.0f && nz <= 1.0f) {
            float depth = clampf(-ssp.z, P->nearDist, P->farDist);
            mn = fminf(mn, depth - EPSILON);
            mx = fmaxf(mx, depth + EPSILON);
        }
    }
    outMinMax[0] = mn;
    outMinMax[1] = mx;
}

// VulkanRayTracedAmbientOcclusion.glsl:178-319 restated over analytic capsules (documented deviation:
// the reference traces AO against 6-gon triangle tubes; hit position/normal/tangent come from the
// capsule hit instead of barycentric interpolation).
void lvo_render_ao(const lvo_scene* sc, const lvo_params* Pp, int useBvh, uint32_t x0, uint32_t y0, uint32_t w,
                   uint32_t h, float* aoOut, lvo_stats* stats) {
    const lvo_params& P = *Pp;
    Frame F = makeFrame(P);
    const bool capped = P.useCappedTubes != 0;
    uint64_t rays = 0, nodes = 0, prims = 0;
    // Elliptic tubes: the reference's RTAO pass traces the elliptic triangle tubes (createCappedTriangleEllipticTubesRenderDataCPU);
    // the build traces the analytic tubelets of the colour pass -- the same substitution as capsules for the circular tubes.
    const bool elliptic = P.useEllipticTubes != 0;
    EllipticScope ellScope(elliptic, P.bandWidth, P.minBandThickness, F.cameraPosition);
    for (uint32_t iter = 0; iter < P.aoIterations; iter++) {
        // SVGF: DISABLE_ACCUMULATION (no running means) + useGlobalFrameNumber (seeds from a counter that onHasMoved does not
        // reset), VulkanRayTracedAmbientOcclusion.cpp:415-421,576-581
        const uint32_t frameNumber = g_lvoAoFeatures.svgf ? 0u : iter;
        const uint32_t globalFrameNumber = g_lvoAoFeatures.svgf ? g_lvoAoFeatures.globalFrameNumber + iter : frameNumber;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : rays, nodes, prims)
        for (int64_t tileIdx = 0; tileIdx < lvoTileCount(w, h); tileIdx++) { // 16x16-pixel tiles
            Counters cnt;
            for (uint32_t tilePix = 0; tilePix < 256u; tilePix++) {
                uint32_t xx, yy;
                if (!lvoTilePixel(w, h, tileIdx, tilePix, xx, yy)) continue;
                uint32_t x = x0 + xx, y = y0 + uint32_t(yy);
                uint32_t pix = x + y * P.width;
                uint32_t seed = tea(pix, globalFrameNumber);
                float xix = 0.5f, xiy = 0.5f;
                if (P.aoJitterPrimary) { xix = rnd(seed); xiy = rnd(seed); }
                V3 o, d;
                primaryRay(P, F, x, y, xix, xiy, o, d);
                Hit hit;
                float aoFactor = 1.0f;
                bool hasHitSurface = false;
                V3 featNormal = v3(0, 0, 0), featPosition = v3(0, 0, 0); // surfaceNormal / vertexPositionWorld of a miss, glsl:211-212
                if (closestHit(*sc, F.radius, capped, useBvh != 0, o, d, 0.0001f, 1000.0f, hit, cnt)) {
                    const lvo_line_point& lp0 = sc->pts[sc->segIdx[2 * hit.seg]];
                    const lvo_line_point& lp1 = sc->pts[sc->segIdx[2 * hit.seg + 1]];
                    V3 P0 = ld3(lp0.linePosition), P1 = ld3(lp1.linePosition);
                    V3 vertexPositionWorld = o + d * hit.t;
                    V3 v = P1 - P0;
                    float ts;
                    if (hit.kind == 0) ts = dot(v, vertexPositionWorld - P0) / dot(v, v);
                    else ts = hit.kind == 1 ? 0.0f : 1.0f;
                    V3 linePosition = hit.kind == 0 ? P0 + ts * v : (hit.kind == 1 ? P0 : P1);
                    V3 surfaceNormal = normalize(vertexPositionWorld - linePosition);
                    if (elliptic) {
                        const EllipticSurface E = ellipticSurface(P, o, d, hit.t, lp0, lp1);
                        ts = E.t; linePosition = E.linePosition; surfaceNormal = E.normal;
                    }
                    V3 surfaceTangent = normalize((1.0f - ts) * ld3(lp0.lineTangent) + ts * ld3(lp1.lineTangent));
                    V3 surfaceBitangent = cross(surfaceNormal, surfaceTangent);
                    float offsetFactor = length(linePosition - vertexPositionWorld) / F.subdivisionCorrectionFactor;
                    hasHitSurface = true; featNormal = surfaceNormal; featPosition = vertexPositionWorld;
                    aoFactor = 0.0f;
                    for (uint32_t s = 0; s < P.aoSamplesPerFrame; s++) {
                        uint32_t sseed = tea(pix, globalFrameNumber * P.aoSamplesPerFrame + s);
                        float xi0 = rnd(sseed), xi1 = rnd(sseed);
                        // sampleHemisphere, glsl:151-156
                        float sn, cs;
                        sincos2pi(xi1, sn, cs);
                        float r = sqrtf(1.0f - xi0 * xi0);
                        V3 smp = v3(cs * r, sn * r, xi0);
                        // frame * sample, frame = mat3(T, B, N)
                        V3 dirU = v3((surfaceTangent.x * smp.x + surfaceBitangent.x * smp.y) + surfaceNormal.x * smp.z,
                                     (surfaceTangent.y * smp.x + surfaceBitangent.y * smp.y) + surfaceNormal.y * smp.z,
                                     (surfaceTangent.z * smp.x + surfaceBitangent.z * smp.y) + surfaceNormal.z * smp.z);
                        V3 rd = normalize(dirU);
                        V3 ro = vertexPositionWorld + rd * offsetFactor;
                        Hit ah;
                        float occ = 1.0f;
                        // traceAoRay, glsl:158-175 (closest hit in [0, radius]; without useDistance any hit)
                        if (closestHit(*sc, F.radius, capped, useBvh != 0, ro, rd, 0.0f, P.aoRadius, ah, cnt))
                            occ = P.aoUseDistance ? ah.t / P.aoRadius : 0.0f;
                        aoFactor += occ;
                    }
                    aoFactor /= float(P.aoSamplesPerFrame);
                }
                size_t idx = size_t(y) * P.width + x;
                if (frameNumber != 0) aoFactor = mixf(aoOut[idx], aoFactor, 1.0f / float(frameNumber + 1));
                aoOut[idx] = aoFactor;
                writeAoFeatures(P, F, x, y, idx, frameNumber, hasHitSurface, featNormal, featPosition);
            }
            rays += cnt.rays; nodes += cnt.nodes; prims += cnt.prims;
        }
    }
    if (stats) { stats->raysTraced += rays; stats->nodesVisited += nodes; stats->primsTested += prims; stats->bvhDepth = sc->bvhDepth; }
}

// RayGen main() + traceRayTransparent + Miss, TubeRayTracing.glsl:61-82,198-298
// prevRGBA8OrNull: the tile of the previous frame (multi-frame accumulation, TubeRayTracing.glsl:268-273: the running
// mean round-trips through the rgba8 output image)
static void renderRt(const lvo_scene* sc, const lvo_params* Pp, int useBvh, const float* ao, const PrebakedAo* pb,
                     uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint8_t* outRGBA8, lvo_stats* stats,
                     const uint8_t* prevRGBA8OrNull = nullptr) {
    const lvo_params& P = *Pp;
    Frame F = makeFrame(P);
    // Linear Swept Spheres: the capsules with their caps, exact roots (no intersection shader to be literal about)
    const bool capped = P.useCappedTubes != 0 || P.lssGeometry != 0;
    struct LiteralOff { bool saved; bool on; LiteralOff(bool o) : saved(g_dev.literalIntersection), on(o) { if (on) g_dev.literalIntersection = false; }
                        ~LiteralOff() { if (on) g_dev.literalIntersection = saved; } } literalOff(P.lssGeometry != 0);
    const float HIT_DISTANCE_EPSILON = 1e-5f;
    uint64_t rays = 0, nodes = 0, prims = 0, hits = 0;
    g_dev.aoImage = (P.useAmbientOcclusion && !pb) ? ao : nullptr; // full-viewport AO image for the reference lookup switch
    const bool elliptic = P.useEllipticTubes != 0;
    EllipticScope ellScope(elliptic, P.bandWidth, P.minBandThickness, F.cameraPosition);
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : rays, nodes, prims, hits)
    for (int64_t tileIdx = 0; tileIdx < lvoTileCount(w, h); tileIdx++) { // 16x16-pixel tiles
        Counters cnt;
        for (uint32_t tilePix = 0; tilePix < 256u; tilePix++) {
            uint32_t xx, yy;
            if (!lvoTilePixel(w, h, tileIdx, tilePix, xx, yy)) continue;
            uint32_t x = x0 + xx, y = y0 + uint32_t(yy);
            float fragmentColor[4] = {0, 0, 0, 0};
            const float aoTexel = (P.useAmbientOcclusion && ao) ? ao[size_t(y) * P.width + x] : 1.0f;
            uint32_t nSamples = P.useJitteredRays ? P.numSamplesPerFrame : 1u;
            for (uint32_t sampleIdx = 0; sampleIdx < nSamples; sampleIdx++) {
                float xix = 0.5f, xiy = 0.5f;
                if (P.useJitteredRays) {
                    uint32_t seed = P.useDeterministicSampling
                            ? tea(19u, P.frameNumber * P.numSamplesPerFrame + sampleIdx)
                            : tea(x + y * P.width, P.frameNumber * P.numSamplesPerFrame + sampleIdx);
                    xix = rnd(seed); xiy = rnd(seed);
                }
                V3 o, d;
                primaryRay(P, F, x, y, xix, xiy, o, d);
                // traceRayTransparent
                float fc[4] = {0, 0, 0, 0};
                float tMin = 0.0001f, tMax = 1000.0f;
                for (uint32_t hitIdx = 0; hitIdx < P.maxDepthComplexity; hitIdx++) {
                    Hit hit;
                    float hc[4]; float payloadHitT; bool hasHit;
                    if (closestHit(*sc, F.radius, capped, useBvh != 0, o, d, tMin, tMax, hit, cnt)) {
                        if (elliptic) shadeHitElliptic(*sc, P, F, aoTexel, o, d, hit, hc, payloadHitT, nullptr, pb);
                        else shadeHit(*sc, P, F, aoTexel, o, d, hit, hc, payloadHitT, pb);
                        hasHit = true;
                        cnt.hits++;
                    } else {
                        for (int k = 0; k < 4; k++) hc[k] = P.background[k];
                        payloadHitT = 0.0f;
                        hasHit = false;
                    }
                    tMin = payloadHitT + fmaxf(payloadHitT * HIT_DISTANCE_EPSILON, 1e-7f);
                    for (int k = 0; k < 3; k++) fc[k] = fc[k] + ((1.0f - fc[3]) * hc[3]) * hc[k];
                    fc[3] = fc[3] + (1.0f - fc[3]) * hc[3];
                    if (!hasHit || fc[3] > 0.99f) break;
                }
                for (int k = 0; k < 4; k++) fragmentColor[k] += fc[k];
            }
            if (P.useJitteredRays)
                for (int k = 0; k < 4; k++) fragmentColor[k] /= float(P.numSamplesPerFrame);
            if (P.frameNumber != 0 && prevRGBA8OrNull) {
                const uint8_t* pv = prevRGBA8OrNull + 4 * (size_t(yy) * w + xx);
                for (int k = 0; k < 4; k++)
                    fragmentColor[k] = mixf(float(pv[k]) / 255.0f, fragmentColor[k], 1.0f / float(P.frameNumber + 1));
            }
            uint8_t* px = outRGBA8 + 4 * (size_t(yy) * w + xx);
            for (int k = 0; k < 4; k++) px[k] = toUnorm8(fragmentColor[k]);
        }
        rays += cnt.rays; nodes += cnt.nodes; prims += cnt.prims; hits += cnt.hits;
    }
    if (stats) {
        stats->raysTraced += rays; stats->nodesVisited += nodes; stats->primsTested += prims; stats->hitsShaded += hits;
        stats->bvhDepth = sc->bvhDepth;
    }
}

// ClosestHitTubeTriangles (TubeRayTracing.glsl:301-352) + LineAttributesBarycentric.glsl:1-52,140-170 -> computeFragmentColor
inline void shadeHitTri(const lvo_scene& sc, const lvo_tri_scene& tsc, const lvo_params& P, const Frame& F, float aoTexel,
                        const PrebakedAo* pb, const TriHit& hit, float hc[4], float& payloadHitT) {
    const uint32_t* ti = &tsc.idx[3 * size_t(hit.tri)];
    const lvo_tube_vertex& vd0 = tsc.verts[ti[0]];
    const lvo_tube_vertex& vd1 = tsc.verts[ti[1]];
    const lvo_tube_vertex& vd2 = tsc.verts[ti[2]];
    const V3 bc = v3((1.0f - hit.u) - hit.v, hit.u, hit.v);
    const lvo_line_point& lp0 = tsc.pts[vd0.vertexLinePointIndex & 0x7FFFFFFFu];
    const lvo_line_point& lp1 = tsc.pts[vd1.vertexLinePointIndex & 0x7FFFFFFFu];
    const lvo_line_point& lp2 = tsc.pts[vd2.vertexLinePointIndex & 0x7FFFFFFFu];
    const bool isCap = P.useCappedTubes && (((vd0.vertexLinePointIndex | vd1.vertexLinePointIndex |
                                              vd2.vertexLinePointIndex) >> 31) != 0u);
    V3 fragPos = interpolateVec3(ld3(vd0.vertexPosition), ld3(vd1.vertexPosition), ld3(vd2.vertexPosition), bc);
    V3 fragmentNormal = normalize(interpolateVec3(ld3(vd0.vertexNormal), ld3(vd1.vertexNormal), ld3(vd2.vertexNormal), bc));
    V3 fragmentTangent = normalize(interpolateVec3(ld3(lp0.lineTangent), ld3(lp1.lineTangent), ld3(lp2.lineTangent), bc));
    float fragmentAttribute = (lp0.lineAttribute * bc.x + lp1.lineAttribute * bc.y) + lp2.lineAttribute * bc.z;
    float aoT = aoTexel;
    if (pb) {
        // LineAttributesBarycentric.glsl:44-52: interpolateAngle (BarycentricInterpolation.glsl:43-55)
        // of the vertex angles, interpolated line-vertex id
        const float PI = 3.14159265358979323846f;
        float a0 = vd0.phi, a1 = vd1.phi, a2 = vd2.phi;
        if (a1 - a0 > PI || a2 - a0 > PI) a0 += 2.0f * PI;
        if (a0 - a1 > PI || a2 - a1 > PI) a1 += 2.0f * PI;
        if (a0 - a2 > PI || a1 - a2 > PI) a2 += 2.0f * PI;
        float phi = (a0 * bc.x + a1 * bc.y) + a2 * bc.z;
        float fragmentVertexId = (float(vd0.vertexLinePointIndex & 0x7FFFFFFFu) * bc.x +
                                  float(vd1.vertexLinePointIndex & 0x7FFFFFFFu) * bc.y) +
                                 float(vd2.vertexLinePointIndex & 0x7FFFFFFFu) * bc.z;
        aoT = prebakedAoLookup(*pb, fragmentVertexId, phi);
    }
    if (P.useHelicityBands) {
        // USE_ROTATING_HELICITY_BANDS in the triangle closest-hit shader (LineAttributesBarycentric.glsl:43-92): interpolated angle,
        // interpolated lineRotation x helicityRotationFactor; on the caps the rotation is continued linearly along the line
        // (distance of the fragment to the plane through the end point x the rotation per length of the end segment)
        const float PI = 3.14159265358979323846f;
        float a0 = vd0.phi, a1 = vd1.phi, a2 = vd2.phi;
        if (a1 - a0 > PI || a2 - a0 > PI) a0 += 2.0f * PI;
        if (a0 - a1 > PI || a2 - a1 > PI) a1 += 2.0f * PI;
        if (a0 - a2 > PI || a1 - a2 > PI) a2 += 2.0f * PI;
        HelicityArgs hl;
        hl.phi = (a0 * bc.x + a1 * bc.y) + a2 * bc.z;
        const float f = P.helicityRotationFactor;
        hl.fragmentRotation = ((lp0.lineRotation * f) * bc.x + (lp1.lineRotation * f) * bc.y) + (lp2.lineRotation * f) * bc.z;
        if (isCap) {
            const uint32_t i0 = vd0.vertexLinePointIndex & 0x7FFFFFFFu;
            const lvo_line_point* other = nullptr;
            if (i0 != 0u && tsc.pts[i0 - 1].lineStartIndex == lp0.lineStartIndex) other = &tsc.pts[i0 - 1];
            if (!other) other = &tsc.pts[i0 + 1];
            const float fragmentRotationDelta = (lp0.lineRotation - other->lineRotation) * f;
            V3 planeNormal = ld3(lp0.linePosition) - ld3(other->linePosition);
            const float segmentLength = length(planeNormal);
            planeNormal = v3(planeNormal.x / segmentLength, planeNormal.y / segmentLength, planeNormal.z / segmentLength);
            const float planeDist = -dot(planeNormal, ld3(lp0.linePosition));
            const float distToPlane = dot(planeNormal, fragPos) + planeDist;
            hl.fragmentRotation += fragmentRotationDelta * distToPlane / segmentLength;
        }
        hl.rotationSeparatorScale = 1.0f;
        if (P.uniformHelicityBandWidth) {
            // UNIFORM_HELICITY_BAND_WIDTH, LineAttributesBarycentric.glsl:94-112: the stripe keeps its width on the surface whatever
            // the pitch of the helix it follows -- rotation per length along the line against the circumference per angle (r = lineWidth / 2)
            const uint32_t i0 = vd0.vertexLinePointIndex & 0x7FFFFFFFu;
            const lvo_line_point* other = nullptr;
            float rotDy = 0.0f;
            if (i0 != 0u && tsc.pts[i0 - 1].lineStartIndex == lp0.lineStartIndex) {
                other = &tsc.pts[i0 - 1];
                rotDy = (lp0.lineRotation - other->lineRotation) * f;
            }
            if (!other) {
                other = &tsc.pts[i0 + 1];
                rotDy = (other->lineRotation - lp0.lineRotation) * f;
            }
            const float rotDx = length(ld3(lp0.linePosition) - ld3(other->linePosition));
            float sn, cs;
            sincosRad(atan2Det(rotDy * 0.5f * P.lineWidth, rotDx), sn, cs);
            hl.rotationSeparatorScale = cs;
        }
        computeFragmentColor(sc, P, F, aoT, fragPos, fragmentNormal, fragmentTangent, isCap, fragmentAttribute, hc, payloadHitT,
                             nullptr, &hl);
        return;
    }
    if (P.useBands) {
        // USE_BANDS in the triangle closest-hit shader (LineAttributesBarycentric.glsl:44-63): interpolated angle, line position and
        // line normal; useBand = true (no ANALYTIC_TUBE_INTERSECTIONS here, RayHitCommon.glsl:164-172)
        const float PI = 3.14159265358979323846f;
        float a0 = vd0.phi, a1 = vd1.phi, a2 = vd2.phi;
        if (a1 - a0 > PI || a2 - a0 > PI) a0 += 2.0f * PI;
        if (a0 - a1 > PI || a2 - a1 > PI) a1 += 2.0f * PI;
        if (a0 - a2 > PI || a1 - a2 > PI) a2 += 2.0f * PI;
        BandArgs b;
        b.useBand = true;
        b.phi = (a0 * bc.x + a1 * bc.y) + a2 * bc.z;
        b.linePosition = interpolateVec3(ld3(lp0.linePosition), ld3(lp1.linePosition), ld3(lp2.linePosition), bc);
        b.lineNormal = interpolateVec3(ld3(lp0.lineNormal), ld3(lp1.lineNormal), ld3(lp2.lineNormal), bc);
        computeFragmentColor(sc, P, F, aoT, fragPos, fragmentNormal, fragmentTangent, isCap, fragmentAttribute, hc, payloadHitT, &b);
        return;
    }
    computeFragmentColor(sc, P, F, aoT, fragPos, fragmentNormal, fragmentTangent, isCap,
                         fragmentAttribute, hc, payloadHitT);
}

// The ray tracer's "Triangle Mesh" geometry mode: RayGen / traceRayTransparent / Miss as above, closest hit
// ClosestHitTubeTriangles (TubeRayTracing.glsl:301-352) + LineAttributesBarycentric.glsl:1-52,140-170 -> computeFragmentColor.
// sc supplies the transfer function, tsc the triangle tubes.
static void renderRtTri(const lvo_scene* sc, const lvo_tri_scene* tsc, const lvo_params* Pp, int useBvh, const float* ao,
                        const PrebakedAo* pb, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint8_t* outRGBA8,
                        lvo_stats* stats) {
    const lvo_params& P = *Pp;
    Frame F = makeFrame(P);
    const float HIT_DISTANCE_EPSILON = 1e-5f;
    uint64_t rays = 0, nodes = 0, prims = 0, hits = 0;
    g_dev.aoImage = (P.useAmbientOcclusion && !pb) ? ao : nullptr;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : rays, nodes, prims, hits)
    for (int64_t tileIdx = 0; tileIdx < lvoTileCount(w, h); tileIdx++) { // 16x16-pixel tiles
        Counters cnt;
        for (uint32_t tilePix = 0; tilePix < 256u; tilePix++) {
            uint32_t xx, yy;
            if (!lvoTilePixel(w, h, tileIdx, tilePix, xx, yy)) continue;
            uint32_t x = x0 + xx, y = y0 + uint32_t(yy);
            float fragmentColor[4] = {0, 0, 0, 0};
            const float aoTexel = (P.useAmbientOcclusion && ao) ? ao[size_t(y) * P.width + x] : 1.0f;
            uint32_t nSamples = P.useJitteredRays ? P.numSamplesPerFrame : 1u;
            for (uint32_t sampleIdx = 0; sampleIdx < nSamples; sampleIdx++) {
                float xix = 0.5f, xiy = 0.5f;
                if (P.useJitteredRays) {
                    uint32_t seed = P.useDeterministicSampling
                            ? tea(19u, P.frameNumber * P.numSamplesPerFrame + sampleIdx)
                            : tea(x + y * P.width, P.frameNumber * P.numSamplesPerFrame + sampleIdx);
                    xix = rnd(seed); xiy = rnd(seed);
                }
                V3 o, d;
                primaryRay(P, F, x, y, xix, xiy, o, d);
                float fc[4] = {0, 0, 0, 0};
                float tMin = 0.0001f, tMax = 1000.0f;
                for (uint32_t hitIdx = 0; hitIdx < P.maxDepthComplexity; hitIdx++) {
                    TriHit hit;
                    float hc[4]; float payloadHitT; bool hasHit;
                    if (closestTri(*tsc, useBvh != 0, o, d, tMin, tMax, hit, cnt)) {
                        shadeHitTri(*sc, *tsc, P, F, aoTexel, pb, hit, hc, payloadHitT);
                        hasHit = true;
                        cnt.hits++;
                    } else {
                        for (int k = 0; k < 4; k++) hc[k] = P.background[k];
                        payloadHitT = 0.0f;
                        hasHit = false;
                    }
                    tMin = payloadHitT + fmaxf(payloadHitT * HIT_DISTANCE_EPSILON, 1e-7f);
                    for (int k = 0; k < 3; k++) fc[k] = fc[k] + ((1.0f - fc[3]) * hc[3]) * hc[k];
                    fc[3] = fc[3] + (1.0f - fc[3]) * hc[3];
                    if (!hasHit || fc[3] > 0.99f) break;
                }
                for (int k = 0; k < 4; k++) fragmentColor[k] += fc[k];
            }
            if (P.useJitteredRays)
                for (int k = 0; k < 4; k++) fragmentColor[k] /= float(P.numSamplesPerFrame);
            uint8_t* px = outRGBA8 + 4 * (size_t(yy) * w + xx);
            for (int k = 0; k < 4; k++) px[k] = toUnorm8(fragmentColor[k]);
        }
        rays += cnt.rays; nodes += cnt.nodes; prims += cnt.prims; hits += cnt.hits;
    }
    if (stats) {
        stats->raysTraced += rays; stats->nodesVisited += nodes; stats->primsTested += prims; stats->hitsShaded += hits;
    }
}

// ---------------------------------------------------------------- multi-layer alpha tracing (MLAT)
// MlatNode / RayPayload, TubeRayTracingHeader.glsl:61-94
struct MlatNode { float color[4]; float transmittance; float depth; };

// merge(), MlatInsert.glsl:35-58
static MlatNode mlatMerge(const MlatNode& a, const MlatNode& b, float& depth2, bool isFirst) {
    MlatNode r;
    r.transmittance = a.transmittance * b.transmittance;
    float fa = 1.0f;
    float fb = a.transmittance;
    r.depth = a.depth;
    depth2 = fmaxf(depth2, b.depth);
    if (b.depth < depth2 && !isFirst) { // node b lies inside the span node a already covers
        float d = (b.depth - a.depth);
        d /= (depth2 - a.depth);
        float a_pow_d = powDet(a.transmittance, d);
        fa = (a_pow_d - 1.0f);
        fa += (a.transmittance - a_pow_d) * b.transmittance;
        fa /= (a.transmittance - 1.0f);
        fb = a_pow_d;
    }
    for (int k = 0; k < 4; k++) r.color[k] = fa * a.color[k] + fb * b.color[k];
    return r;
}

// insertNodeMlat(), MlatInsert.glsl:66-221.  Returns true when the any-hit shader ACCEPTS the hit (opaque fragment or
// enough absorption in front of it): the ray interval then ends at this depth.  false = ignoreIntersectionEXT.
static bool mlatInsert(MlatNode* nodes, int numNodes, float& depth2, const float color[4], float depth, bool missShader) {
    MlatNode newNode;
    newNode.depth = depth;
    const float alpha = color[3];
    if (!missShader && alpha == 0.0f) return false;
    newNode.transmittance = 1.0f - alpha;
    newNode.color[0] = alpha * color[0]; newNode.color[1] = alpha * color[1]; newNode.color[2] = alpha * color[2];
    newNode.color[3] = color[3];
    for (int i = numNodes - 1; i >= 0; --i)
        if (newNode.depth > nodes[i].depth) std::swap(newNode, nodes[i]);
    // what fell off the front is merged with the first node (MLAB merges the last two instead)
    if (newNode.depth > 0.0f) nodes[0] = mlatMerge(newNode, nodes[0], depth2, newNode.depth == depth);
    if (alpha == 1.0f) return true;
    float transmittance = 1.0f;
    for (int i = 0; i < numNodes; ++i) transmittance *= nodes[i].transmittance;
    if (transmittance <= 0.001f && nodes[numNodes - 1].depth <= depth) return true;
    return false;
}

// RayGen main() + traceRayMlat (TubeRayTracing.glsl:86-192) with the any-hit shader AnyHitTubeAnalytic
// (= ClosestHitTubeAnalytic + insertNodeMlat) and the MLAT miss shader (:290-293).
// The ORDER in which the reference's any-hit shader meets the candidates is the driver's traversal order, i.e. undefined;
// MLAT's result depends on it once more than numNodes layers exist or the early-termination rules fire.  Two orders here:
//   traceOffsets == NULL  canonical: ascending segment index;
//   traceOffsets != NULL  replay: pixel i (row-major in the tile) visits traceSegs[traceOffsets[i] .. traceOffsets[i+1])
//                         in that order; traceFlags[j] = 0 inserted, 1 = dropped at insertion time because the ray
//                         interval had already shrunk below its depth.  The replay also VALIDATES the order: every
//                         listed segment must be a hit inside the interval valid at that moment, a dropped one must lie
//                         beyond it, and every hit that is not listed must lie beyond the final interval (or be fully
//                         transparent); *outViolations counts the entries that break these rules.
// outNodesOrNull: per pixel numNodes * 6 floats {color[4], transmittance, depth} + depth2 (last sample).
// tscOrNull: the "Triangle Mesh" geometry mode (AnyHitTubeTriangles): candidates are triangles, ids triangle indices.
static void renderRtMlat(const lvo_scene* sc, const lvo_tri_scene* tscOrNull, const lvo_params* Pp, int useBvh,
                         const float* ao, uint32_t x0, uint32_t y0,
                         uint32_t w, uint32_t h, uint32_t numNodes, const uint64_t* traceOffsets,
                         const uint32_t* traceSegs, const uint8_t* traceFlags, uint8_t* outRGBA8, float* outNodesOrNull,
                         uint64_t* outViolations, lvo_stats* stats) {
    const lvo_params& P = *Pp;
    Frame F = makeFrame(P);
    const bool capped = P.useCappedTubes != 0 || P.lssGeometry != 0;
    uint64_t rays = 0, nodesV = 0, prims = 0, hitsShaded = 0, violations = 0;
    g_dev.aoImage = P.useAmbientOcclusion ? ao : nullptr;
    // band data: AnyHitEllipticTubeAnalytic (EllipticTubeRayTracing.glsl:463-466) = ClosestHitEllipticTubeAnalytic + insertNodeMlat
    const bool elliptic = P.useEllipticTubes != 0 && !tscOrNull;
    EllipticScope ellScope(elliptic, P.bandWidth, P.minBandThickness, F.cameraPosition);
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : rays, nodesV, prims, hitsShaded, violations)
    for (int64_t yy = 0; yy < int64_t(h); yy++) {
        Counters cnt;
        std::vector<Hit> hits;   // triangles: seg = triangle index, kind unused
        std::vector<TriHit> triHits;
        std::vector<MlatNode> nodes(numNodes);
        std::vector<uint32_t> listed;
        for (uint32_t xx = 0; xx < w; xx++) {
            uint32_t x = x0 + xx, y = y0 + uint32_t(yy);
            const size_t pix = size_t(yy) * w + xx;
            float fragmentColor[4] = {0, 0, 0, 0};
            const float aoTexel = (P.useAmbientOcclusion && ao) ? ao[size_t(y) * P.width + x] : 1.0f;
            uint32_t nSamples = P.useJitteredRays ? P.numSamplesPerFrame : 1u;
            float depth2 = 0.0f;
            for (uint32_t sampleIdx = 0; sampleIdx < nSamples; sampleIdx++) {
                float xix = 0.5f, xiy = 0.5f;
                if (P.useJitteredRays) {
                    uint32_t seed = P.useDeterministicSampling
                            ? tea(19u, P.frameNumber * P.numSamplesPerFrame + sampleIdx)
                            : tea(x + y * P.width, P.frameNumber * P.numSamplesPerFrame + sampleIdx);
                    xix = rnd(seed); xiy = rnd(seed);
                }
                V3 o, d;
                primaryRay(P, F, x, y, xix, xiy, o, d);
                for (auto& n : nodes) { n.color[0] = n.color[1] = n.color[2] = n.color[3] = 0.0f; n.transmittance = 1.0f; n.depth = 0.0f; }
                depth2 = 0.0f;
                const float tMin = 0.0001f;
                float tMax = 1000.0f;
                bool accepted = false;
                auto shade = [&](const Hit& hit, float hc[4]) {
                    float payloadHitT;
                    if (tscOrNull) {
                        auto it = std::lower_bound(triHits.begin(), triHits.end(), hit.seg,
                                                   [](const TriHit& a, uint32_t t) { return a.tri < t; });
                        shadeHitTri(*sc, *tscOrNull, P, F, aoTexel, nullptr, *it, hc, payloadHitT);
                    } else if (elliptic) {
                        shadeHitElliptic(*sc, P, F, aoTexel, o, d, hit, hc, payloadHitT);
                    } else {
                        shadeHit(*sc, P, F, aoTexel, o, d, hit, hc, payloadHitT);
                    }
                };
                auto visit = [&](const Hit& hit) {
                    float hc[4];
                    shade(hit, hc);
                    cnt.hits++;
                    if (mlatInsert(nodes.data(), int(numNodes), depth2, hc, hit.t, false)) { accepted = true; tMax = hit.t; }
                    return hc[3];
                };
                if (tscOrNull) { // all triangle hits (brute force), ascending triangle index
                    cnt.rays++;
                    hits.clear();
                    triHits.clear();
                    const V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                    for (uint32_t tri = 0; tri < tscOrNull->nTri; tri++) {
                        V3 a, b, c;
                        triVerts(*tscOrNull, tri, a, b, c);
                        TriHit th;
                        cnt.prims++;
                        if (rayTriangle(o, d, inv, a, b, c, tscOrNull->pad, th.t, th.u, th.v) && th.t >= tMin && th.t <= 1000.0f) {
                            th.tri = tri;
                            triHits.push_back(th);
                            hits.push_back(Hit{th.t, tri, 0});
                        }
                    }
                } else {
                    allHits(*sc, F.radius, capped, useBvh != 0, o, d, tMin, 1000.0f, hits, cnt); // ascending segment index
                }
                if (!traceOffsets) {
                    for (const Hit& hit : hits)
                        if (hit.t <= tMax) visit(hit);
                } else {
                    listed.clear();
                    for (uint64_t j = traceOffsets[pix]; j < traceOffsets[pix + 1]; j++) {
                        const uint32_t seg = traceSegs[j];
                        listed.push_back(seg);
                        auto it = std::lower_bound(hits.begin(), hits.end(), seg, [](const Hit& a, uint32_t s) { return a.seg < s; });
                        if (it == hits.end() || it->seg != seg) { violations++; continue; } // not a hit of this ray at all
                        if (traceFlags[j] == 0) {
                            if (!(it->t <= tMax)) violations++;
                            visit(*it);
                        } else if (!(it->t > tMax)) {
                            violations++;
                        }
                    }
                    std::sort(listed.begin(), listed.end());
                    for (size_t k = 1; k < listed.size(); k++) if (listed[k] == listed[k - 1]) violations++; // visited twice
                    for (const Hit& hit : hits) {
                        if (std::binary_search(listed.begin(), listed.end(), hit.seg)) continue;
                        if (hit.t > tMax) continue; // culled by the shrunken interval
                        float hc[4];
                        shade(hit, hc);
                        if (hc[3] != 0.0f) violations++; // a visible layer inside the final interval was never visited
                    }
                }
                if (!accepted) mlatInsert(nodes.data(), int(numNodes), depth2, P.background, 1e7f, true); // Miss
                // front-to-back blending of the node list (pre-multiplied colours), TubeRayTracing.glsl:141-189
                float fc[4] = {0, 0, 0, 0};
                for (uint32_t i = 0; i < numNodes; i++) {
                    const float* hc = nodes[i].color;
                    for (int k = 0; k < 3; k++) fc[k] = fc[k] + (1.0f - fc[3]) * hc[k];
                    fc[3] = fc[3] + (1.0f - fc[3]) * hc[3];
                }
                for (int k = 0; k < 4; k++) fragmentColor[k] += fc[k];
            }
            if (P.useJitteredRays)
                for (int k = 0; k < 4; k++) fragmentColor[k] /= float(P.numSamplesPerFrame);
            uint8_t* px = outRGBA8 + 4 * pix;
            for (int k = 0; k < 4; k++) px[k] = toUnorm8(fragmentColor[k]);
            if (outNodesOrNull) {
                float* dst = outNodesOrNull + pix * (size_t(numNodes) * 6 + 1);
                for (uint32_t i = 0; i < numNodes; i++) {
                    for (int k = 0; k < 4; k++) dst[6 * i + k] = nodes[i].color[k];
                    dst[6 * i + 4] = nodes[i].transmittance;
                    dst[6 * i + 5] = nodes[i].depth;
                }
                dst[size_t(numNodes) * 6] = depth2;
            }
        }
        rays += cnt.rays; nodesV += cnt.nodes; prims += cnt.prims; hitsShaded += cnt.hits;
    }
    if (outViolations) *outViolations = violations;
    if (stats) {
        stats->raysTraced += rays; stats->nodesVisited += nodesV; stats->primsTested += prims; stats->hitsShaded += hitsShaded;
        stats->bvhDepth = sc->bvhDepth;
    }
}

// All capsule entry hits of every pixel-centre ray of a tile (t in [1e-4, 1000], ascending segment index): offsets has
// w*h+1 entries; segs / ts may be NULL (first pass: sizes only).
void lvo_pixel_hits(const lvo_scene* sc, const lvo_params* Pp, int useBvh, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                    uint64_t* offsets, uint32_t* segs, float* ts) {
    const lvo_params& P = *Pp;
    Frame F = makeFrame(P);
    std::vector<std::vector<Hit>> rows(size_t(w) * h);
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t tileIdx = 0; tileIdx < lvoTileCount(w, h); tileIdx++) { // 16x16-pixel tiles
        Counters cnt;
        for (uint32_t tilePix = 0; tilePix < 256u; tilePix++) {
            uint32_t xx, yy;
            if (!lvoTilePixel(w, h, tileIdx, tilePix, xx, yy)) continue;
            V3 o, d;
            primaryRay(P, F, x0 + xx, y0 + uint32_t(yy), 0.5f, 0.5f, o, d);
            allHits(*sc, F.radius, P.useCappedTubes != 0, useBvh != 0, o, d, 0.0001f, 1000.0f, rows[size_t(yy) * w + xx], cnt);
        }
    }
    uint64_t n = 0;
    for (size_t i = 0; i < rows.size(); i++) {
        offsets[i] = n;
        for (const Hit& hh : rows[i]) {
            if (segs) segs[n] = hh.seg;
            if (ts) ts[n] = hh.t;
            n++;
        }
    }
    offsets[rows.size()] = n;
}

void lvo_mlat_insert(float* nodes, int numNodes, float* depth2, const float color[4], float depth, int missShader,
                     int* outAccepted) {
    int acc = mlatInsert(reinterpret_cast<MlatNode*>(nodes), numNodes, *depth2, color, depth, missShader != 0) ? 1 : 0;
    if (outAccepted) *outAccepted = acc;
}

void lvo_render_rt_mlat(const lvo_scene* sc, const lvo_params* P, int useBvh, const float* ao, uint32_t x0, uint32_t y0,
                        uint32_t w, uint32_t h, uint32_t numNodes, const uint64_t* traceOffsets,
                        const uint32_t* traceSegs, const uint8_t* traceFlags, uint8_t* outRGBA8, float* outNodesOrNull,
                        uint64_t* outViolations, lvo_stats* stats) {
    renderRtMlat(sc, nullptr, P, useBvh, ao, x0, y0, w, h, numNodes, traceOffsets, traceSegs, traceFlags, outRGBA8,
                 outNodesOrNull, outViolations, stats);
}
void lvo_render_rt_mlat_tri(const lvo_scene* sc, const lvo_tri_scene* tsc, const lvo_params* P, const float* ao, uint32_t x0,
                            uint32_t y0, uint32_t w, uint32_t h, uint32_t numNodes, const uint64_t* traceOffsets,
                            const uint32_t* traceTris, const uint8_t* traceFlags, uint8_t* outRGBA8, float* outNodesOrNull,
                            uint64_t* outViolations, lvo_stats* stats) {
    renderRtMlat(sc, tsc, P, 0, ao, x0, y0, w, h, numNodes, traceOffsets, traceTris, traceFlags, outRGBA8, outNodesOrNull,
                 outViolations, stats);
}

void lvo_render_rt(const lvo_scene* sc, const lvo_params* P, int useBvh, const float* ao, uint32_t x0, uint32_t y0,
                   uint32_t w, uint32_t h, uint8_t* outRGBA8, lvo_stats* stats) {
    renderRt(sc, P, useBvh, ao, nullptr, x0, y0, w, h, outRGBA8, stats);
}
void lvo_render_rt_accumulate(const lvo_scene* sc, const lvo_params* P, int useBvh, const float* ao, uint32_t x0, uint32_t y0,
                              uint32_t w, uint32_t h, const uint8_t* prevRGBA8, uint8_t* outRGBA8, lvo_stats* stats) {
    renderRt(sc, P, useBvh, ao, nullptr, x0, y0, w, h, outRGBA8, stats, prevRGBA8);
}
void lvo_render_rt_tri(const lvo_scene* sc, const lvo_tri_scene* tsc, const lvo_params* P, int useBvh, const float* ao,
                       uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint8_t* outRGBA8, lvo_stats* stats) {
    renderRtTri(sc, tsc, P, useBvh, ao, nullptr, x0, y0, w, h, outRGBA8, stats);
}
// colour pass with STATIC_AMBIENT_OCCLUSION_PREBAKING: AO from the baked table instead of the screen-space texture
void lvo_render_rt_prebaked(const lvo_scene* sc, const lvo_tri_scene* tscOrNull, const lvo_params* P, int useBvh,
                            const float* factors, const float* blendingWeights, uint32_t numLineVertices,
                            uint32_t numParametrizationVertices, uint32_t numAoTubeSubdivisions, uint32_t x0,
                            uint32_t y0, uint32_t w, uint32_t h, uint8_t* outRGBA8, lvo_stats* stats) {
    PrebakedAo pb{factors, blendingWeights, numLineVertices, numParametrizationVertices, numAoTubeSubdivisions};
    if (tscOrNull) renderRtTri(sc, tscOrNull, P, useBvh, nullptr, &pb, x0, y0, w, h, outRGBA8, stats);
    else renderRt(sc, P, useBvh, nullptr, &pb, x0, y0, w, h, outRGBA8, stats);
}

// AmbientOcclusionComputeRenderPass::generateBlendingWeightParametrization + recomputeStaticParametrization,
// VulkanAmbientOcclusionBaker.cpp:513-653.  lines = the polylines as LineData::getFilteredLines returns them; the
// reference indexes the blending weights by line VERTEX (valid line points), so lines must not contain points that the
// tangent test of a2 drops.  outSamplingLocations needs room for sum(ceil(len/expected) + 1) entries: pass NULL to count.
void lvo_ao_parametrization(const float* positions, const uint32_t* lineOffsets, uint32_t nLines,
                            float expectedParamSegmentLength, float* outBlendingWeights, float* outSamplingLocations,
                            uint64_t* outNumParametrizationVertices) {
    const float EPSILON = 1e-5f;
    size_t segmentVertexIdOffset = 0, vertexIdx = 0;
    uint64_t numSampling = 0;
    for (uint32_t lineIdx = 0; lineIdx < nLines; lineIdx++) {
        const float* line = positions + 3 * size_t(lineOffsets[lineIdx]);
        const size_t n = lineOffsets[lineIdx + 1] - lineOffsets[lineIdx];
        auto P = [&](size_t i) { return ld3(line + 3 * i); };
        float polylineLength = 0.0f;
        for (size_t i = 1; i < n; i++) polylineLength += length(P(i) - P(i - 1));
        uint32_t numLineSubdivs = std::max(1u, uint32_t(ceilf(polylineLength / expectedParamSegmentLength)));
        float lineSubdivLength = polylineLength / float(numLineSubdivs);
        uint32_t numSubdivVertices = numLineSubdivs + 1;
        const uint32_t startVertexIdx = uint32_t(vertexIdx);
        if (outBlendingWeights) outBlendingWeights[vertexIdx] = float(segmentVertexIdOffset);
        vertexIdx++;
        float currentLength = 0.0f;
        for (size_t i = 1; i < n; i++) {
            currentLength += length(P(i) - P(i - 1));
            float wgt = currentLength / lineSubdivLength;
            if (outBlendingWeights)
                outBlendingWeights[vertexIdx] = float(segmentVertexIdOffset) + clampf(wgt, 0.0f, float(numLineSubdivs) - EPSILON);
            vertexIdx++;
        }
        float lastLength = 0.0f;
        currentLength = length(P(1) - P(0));
        size_t currVertexIdx = 1;
        if (outSamplingLocations) outSamplingLocations[numSampling] = float(startVertexIdx);
        numSampling++;
        for (uint32_t i = 1; i < numSubdivVertices; i++) {
            uint32_t parametrizationIdx = uint32_t(currentLength / lineSubdivLength);
            while (i > parametrizationIdx && currVertexIdx < n - 1) {
                float segLength = length(P(currVertexIdx + 1) - P(currVertexIdx));
                lastLength = currentLength;
                currentLength += segLength;
                parametrizationIdx = uint32_t(currentLength / lineSubdivLength);
                currVertexIdx++;
            }
            float samplingLocation = float(currVertexIdx - 1) + (float(i) * lineSubdivLength - lastLength) / (currentLength - lastLength);
            samplingLocation = float(startVertexIdx) + std::min(samplingLocation, float(uint32_t(n) - 1u) - EPSILON);
            if (outSamplingLocations) outSamplingLocations[numSampling] = samplingLocation;
            numSampling++;
        }
        segmentVertexIdOffset += numSubdivVertices;
    }
    *outNumParametrizationVertices = numSampling;
}

// VulkanAmbientOcclusionBaker.Compute (VulkanAmbientOcclusionBaker.glsl:190-282), iterated numIterations times
// (bakeAoTexture, VulkanAmbientOcclusionBaker.cpp:193-262).  Rays are traced against the capsules of sc, or against the
// triangle tubes when tscOrNull is given (the reference uses the triangle TLAS, cpp:480).  sin/cos of the tube angle
// use the build's sincos2pi (like the hemisphere sample).  outFactors: numTubeSubdivisions * numSamplingLocations.
// USE_BANDS in the baker (VulkanAmbientOcclusionBaker.glsl:200-257, "bands with minimum thickness"): ray origins on the elliptic cross
// section (bandRadius, minBandThickness), pushed out by 1e-3 instead of 1e-6.  Test hook: set before lvo_bake_ao.
static struct { int use; float bandRadius, minBandThickness; } g_bakeBands = {0, 0.0f, 1.0f};
void lvo_set_bake_bands(int useBands, float bandRadius, float minBandThickness) {
    g_bakeBands.use = useBands; g_bakeBands.bandRadius = bandRadius; g_bakeBands.minBandThickness = minBandThickness;
}
void lvo_bake_ao(const lvo_scene* sc, const lvo_tri_scene* tscOrNull, float lineWidth, int useCappedTubes, int useBvh,
                 const float* samplingLocations, uint32_t numParametrizationVertices, uint32_t numTubeSubdivisions,
                 uint32_t numAmbientOcclusionSamples, uint32_t numIterations, float ambientOcclusionRadius, int useDistance,
                 float* outFactors) {
    const float lineRadius = lineWidth * 0.5f;
    const uint32_t numLinePoints = uint32_t(sc->pts.size());
    for (uint32_t frameNumber = 0; frameNumber < numIterations; frameNumber++) {
#pragma omp parallel for schedule(dynamic, 16)
        for (int64_t idx = 0; idx < int64_t(numParametrizationVertices); idx++) {
            Counters cnt;
            const uint32_t lineSamplingIdx = uint32_t(idx);
            uint32_t seed = tea(lineSamplingIdx, frameNumber);
            // getInterpolatedLinePoint, glsl:110-131
            const float samplingLocation = samplingLocations[lineSamplingIdx];
            const uint32_t lowerIdx = uint32_t(samplingLocation);
            const uint32_t upperIdx = std::min(lowerIdx + 1u, numLinePoints - 1u);
            const float f = samplingLocation - floorf(samplingLocation);
            const lvo_line_point& lo = sc->pts[lowerIdx];
            const lvo_line_point& up = sc->pts[upperIdx];
            auto mix3 = [&](V3 a, V3 b) { return v3(mixf(a.x, b.x, f), mixf(a.y, b.y, f), mixf(a.z, b.z, f)); };
            const V3 binormalLower = cross(ld3(lo.lineTangent), ld3(lo.lineNormal));
            const V3 binormalUpper = cross(ld3(up.lineTangent), ld3(up.lineNormal));
            const V3 position = mix3(ld3(lo.linePosition), ld3(up.linePosition));
            const V3 tangent = normalize(mix3(ld3(lo.lineTangent), ld3(up.lineTangent)));
            const V3 normal = normalize(mix3(ld3(lo.lineNormal), ld3(up.lineNormal)));
            const V3 binormal = normalize(mix3(binormalLower, binormalUpper));
            for (uint32_t sub = 0; sub < numTubeSubdivisions; sub++) {
                float sinAngle, cosAngle;
                sincos2pi(float(sub) / float(numTubeSubdivisions), sinAngle, cosAngle);
                V3 surfaceNormal = cosAngle * normal + sinAngle * binormal;
                V3 rayOrigin = position + (lineRadius + 1e-6f) * surfaceNormal;
                if (g_bakeBands.use) {
                    const float thickness = g_bakeBands.minBandThickness;
                    surfaceNormal = normalize(cosAngle * normal + (thickness * sinAngle) * binormal);
                    rayOrigin = position + (g_bakeBands.bandRadius + 1e-3f) * ((thickness * cosAngle) * normal + sinAngle * binormal);
                }
                const V3 surfaceBitangent = cross(surfaceNormal, tangent);
                float acc = 0.0f;
                for (uint32_t rayIdx = 0; rayIdx < numAmbientOcclusionSamples; rayIdx++) {
                    const float xi0 = rnd(seed), xi1 = rnd(seed);
                    float sn, cs;
                    sincos2pi(xi1, sn, cs);
                    const float r = sqrtf(1.0f - xi0 * xi0);
                    const V3 smp = v3(cs * r, sn * r, xi0);
                    const V3 dirU = v3((tangent.x * smp.x + surfaceBitangent.x * smp.y) + surfaceNormal.x * smp.z,
                                       (tangent.y * smp.x + surfaceBitangent.y * smp.y) + surfaceNormal.y * smp.z,
                                       (tangent.z * smp.x + surfaceBitangent.z * smp.y) + surfaceNormal.z * smp.z);
                    const V3 rd = normalize(dirU);
                    float occ = 1.0f;
                    if (tscOrNull) {
                        TriHit th;
                        if (closestTri(*tscOrNull, useBvh != 0, rayOrigin, rd, 0.0f, ambientOcclusionRadius, th, cnt))
                            occ = useDistance ? th.t / ambientOcclusionRadius : 0.0f;
                    } else {
                        Hit ah;
                        if (closestHit(*sc, lineRadius, useCappedTubes != 0, useBvh != 0, rayOrigin, rd, 0.0f,
                                       ambientOcclusionRadius, ah, cnt))
                            occ = useDistance ? ah.t / ambientOcclusionRadius : 0.0f;
                    }
                    acc += occ;
                }
                acc /= float(numAmbientOcclusionSamples);
                float& dst = outFactors[sub + size_t(numTubeSubdivisions) * lineSamplingIdx];
                if (frameNumber != 0) acc = mixf(dst, acc, 1.0f / float(frameNumber + 1));
                dst = acc;
            }
        }
    }
}

// TiledAddress.glsl:53-85 (ADDRESSING_TILED_2x2 / 2x8 / NxM / linear)
uint32_t lvo_ppll_addr(uint32_t x, uint32_t y, uint32_t viewportW, uint32_t tileW, uint32_t tileH) {
    if (tileW == 1 && tileH == 1) return x + viewportW * y;
    uint32_t surfaceWidth = viewportW / tileW;
    uint32_t tx = x / tileW, ty = y / tileH;
    uint32_t tileAddr1D = (tx + surfaceWidth * ty) * (tileW * tileH);
    uint32_t px = x & (tileW - 1), py = y & (tileH - 1);
    uint32_t pixelAddr1D = px + py * tileW;
    return tileAddr1D | pixelAddr1D;
}

// gatherFragment, LinkedListGather.glsl:33-72; fragment source = capsule entry hits of the pixel-centre ray
void lvo_ppll_gather(const lvo_scene* sc, const lvo_params* Pp, int useBvh, const float* ao, uint32_t x0, uint32_t y0,
                     uint32_t w, uint32_t h, uint32_t* nodes, uint32_t* startOffset, uint32_t* fragCounter,
                     lvo_stats* stats) {
    const lvo_params& P = *Pp;
    Frame F = makeFrame(P);
    const bool capped = P.useCappedTubes != 0;
    uint32_t pw = P.width, ph = P.height;
    padTiling(pw, ph, P.ppllTileW, P.ppllTileH);
    // fragments are generated at pixel centres: own-texel AO lookup (the literal lookup only under the deviation switch)
    g_dev.aoImage = (g_dev.referenceAoLookup && P.useAmbientOcclusion) ? ao : nullptr;
    // band data: the reference's rasterisers draw elliptic tubes in the ribbon primitive mode; here the fragments are the entry
    // hits of the analytic tubelets (or of the capsules with USE_BANDS shading), like the capsule entry hits of plain data
    const bool elliptic = P.useEllipticTubes != 0;
    EllipticScope ellScope(elliptic, P.bandWidth, P.minBandThickness, F.cameraPosition);
    // clear: LinkedListClear.glsl:46-55
    for (size_t i = 0; i < size_t(pw) * ph; i++) startOffset[i] = 0xFFFFFFFFu;
    *fragCounter = 0;
    // per-pixel fragment lists computed in parallel, inserted serially in raster order
    std::vector<std::vector<std::pair<uint32_t, float>>> rows(h);
    std::vector<std::vector<uint32_t>> rowCounts(h);
    uint64_t rays = 0, nds = 0, prims = 0, hits = 0;
    // ppll_fragment_source = raster_prism: the fragments of the rasterised programmable-pull prism (lv_oracle_prism.h)
    const bool prism = P.ppllFragmentSource == 1u;
    const PrismRing ring = prismRingOf(P, F);
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : rays, nds, prims, hits)
    for (int64_t yy = 0; yy < int64_t(h); yy++) {
        Counters cnt;
        std::vector<Hit> hl;
        std::vector<uint32_t> cand;
        std::vector<PrismFrag> pf;
        rowCounts[yy].assign(w, 0);
        for (uint32_t xx = 0; xx < w; xx++) {
            uint32_t x = x0 + xx, y = y0 + uint32_t(yy);
            if (prism) {
                const float aoTexelP = (P.useAmbientOcclusion && ao) ? ao[size_t(y) * P.width + x] : 1.0f;
                prismPixelFragments(*sc, P, F, ring, useBvh != 0, x, y, cand, pf, cnt);
                const RasterQuad rqP = makeRasterQuad(P, F, x, y);
                const size_t first = rows[yy].size();
                for (const PrismFrag& f : pf) {
                    float hc[4]; float hitT;
                    prismShade(*sc, P, F, ring, aoTexelP, f, g_rtFragmentColourInPpll ? nullptr : &rqP, hc, hitT,
                               g_ppllPrebaked.factors ? &g_ppllPrebaked : nullptr);
                    cnt.hits++;
                    if (hc[3] < 0.001f) continue;
                    rows[yy].push_back(std::make_pair(packUnorm4x8(hc), hitT));
                    rowCounts[yy][xx]++;
                }
                // Order of a pixel's list: the rasteriser's fragment shader invocations race for the list head
                // (LinkedListGather.glsl:55), the reference defines none.  Build-owned: inserted in DESCENDING (depth, colour) key
                // order, so that the walk from the head -- and the first maxNumFrags nodes the resolve pass keeps
                // (LinkedListResolve.glsl:66-79) -- sees the nearest fragments first.
                std::sort(rows[yy].begin() + std::ptrdiff_t(first), rows[yy].end(),
                          [](const std::pair<uint32_t, float>& a, const std::pair<uint32_t, float>& b) {
                              return a.second > b.second || (a.second == b.second && a.first > b.first);
                          });
                continue;
            }
            V3 o, d;
            primaryRay(P, F, x, y, 0.5f, 0.5f, o, d);
            const float aoTexel = (P.useAmbientOcclusion && ao) ? ao[size_t(y) * P.width + x] : 1.0f;
            allHits(*sc, F.radius, capped, useBvh != 0, o, d, 0.0001f, 1000.0f, hl, cnt);
            // raster variant of the fragment colour: the rays through the 2 x 2 quad partners (see RasterQuad)
            const RasterQuad rq = makeRasterQuad(P, F, x, y);
            const RasterQuad* rqp = g_rtFragmentColourInPpll ? nullptr : &rq;
            for (const Hit& hit : hl) {
                float hc[4]; float hitT;
                if (elliptic) shadeHitElliptic(*sc, P, F, aoTexel, o, d, hit, hc, hitT, rqp, g_ppllPrebaked.factors ? &g_ppllPrebaked : nullptr);
                else shadeHit(*sc, P, F, aoTexel, o, d, hit, hc, hitT, g_ppllPrebaked.factors ? &g_ppllPrebaked : nullptr, rqp);
                cnt.hits++;
                if (hc[3] < 0.001f) continue;
                rows[yy].push_back(std::make_pair(packUnorm4x8(hc), hitT));
                rowCounts[yy][xx]++;
            }
        }
        rays += cnt.rays; nds += cnt.nodes; prims += cnt.prims; hits += cnt.hits;
    }
    uint32_t maxDc = 0;
    for (uint32_t yy = 0; yy < h; yy++) {
        size_t k = 0;
        for (uint32_t xx = 0; xx < w; xx++) {
            uint32_t pixelIndex = lvo_ppll_addr(x0 + xx, y0 + yy, pw, P.ppllTileW, P.ppllTileH);
            maxDc = std::max(maxDc, rowCounts[yy][xx]);
            for (uint32_t f = 0; f < rowCounts[yy][xx]; f++, k++) {
                uint32_t insertIndex = (*fragCounter)++;
                if (insertIndex < P.ppllLinkedListSize) {
                    uint32_t next = startOffset[pixelIndex];
                    startOffset[pixelIndex] = insertIndex;
                    nodes[3 * size_t(insertIndex)] = rows[yy][k].first;
                    nodes[3 * size_t(insertIndex) + 1] = f2u(rows[yy][k].second);
                    nodes[3 * size_t(insertIndex) + 2] = next;
                }
            }
        }
    }
    if (stats) {
        stats->raysTraced += rays; stats->nodesVisited += nds; stats->primsTested += prims; stats->hitsShaded += hits;
        stats->fragments = *fragCounter;
        stats->maxDepthComplexity = maxDc;
        stats->bvhDepth = sc->bvhDepth;
    }
}

// LinkedListResolve.glsl:57-105 + frontToBackPQ; output blended BACK_TO_FRONT_STRAIGHT_ALPHA over the clear colour
// (PerPixelLinkedListLineRenderer.cpp:70,395-397)
void lvo_ppll_resolve(const lvo_params* Pp, const uint32_t* nodes, const uint32_t* startOffset, int literal, uint32_t x0,
                      uint32_t y0, uint32_t w, uint32_t h, uint8_t* outRGBA8) {
    const lvo_params& P = *Pp;
    uint32_t pw = P.width, ph = P.height;
    padTiling(pw, ph, P.ppllTileW, P.ppllTileH);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t yy = 0; yy < int64_t(h); yy++) {
        std::vector<uint32_t> colorList(P.ppllMaxNumFrags);
        std::vector<float> depthList(P.ppllMaxNumFrags);
        for (uint32_t xx = 0; xx < w; xx++) {
            uint32_t pixelIndex = lvo_ppll_addr(x0 + xx, y0 + uint32_t(yy), pw, P.ppllTileW, P.ppllTileH);
            uint32_t fragOffset = startOffset[pixelIndex];
            uint32_t numFrags = 0;
            for (uint32_t i = 0; i < P.ppllMaxNumFrags; i++) {
                if (fragOffset == 0xFFFFFFFFu) break;
                colorList[i] = nodes[3 * size_t(fragOffset)];
                depthList[i] = u2f(nodes[3 * size_t(fragOffset) + 1]);
                fragOffset = nodes[3 * size_t(fragOffset) + 2];
                numFrags++;
            }
            float out[4] = {P.background[0], P.background[1], P.background[2], P.background[3]};
            if (numFrags > 0) {
                float c[4];
                if (P.ppllSortingMode == 0u) {
                    if (literal) frontToBackPQ<true>(colorList.data(), depthList.data(), numFrags, c);
                    else frontToBackPQ<false>(colorList.data(), depthList.data(), numFrags, c);
                } else if (literal) {
                    sortAndBlend<true>(P.ppllSortingMode, colorList.data(), depthList.data(), numFrags, P.ppllMaxNumFrags, c);
                } else {
                    sortAndBlend<false>(P.ppllSortingMode, colorList.data(), depthList.data(), numFrags, P.ppllMaxNumFrags, c);
                }
                if (c[3] > 0.0f) { // A == 0 (all alphas quantised to 0) is treated as "no fragments"
                    for (int k = 0; k < 3; k++) out[k] = c[k] * c[3] + P.background[k] * (1.0f - c[3]);
                    out[3] = c[3] + P.background[3] * (1.0f - c[3]);
                }
            }
            uint8_t* px = outRGBA8 + 4 * (size_t(yy) * w + xx);
            for (int k = 0; k < 4; k++) px[k] = toUnorm8(out[k]);
        }
    }
}

void lvo_render_ppll(const lvo_scene* sc, const lvo_params* P, int useBvh, const float* ao, uint32_t x0, uint32_t y0,
                     uint32_t w, uint32_t h, uint8_t* outRGBA8, lvo_stats* stats) {
    uint32_t pw = P->width, ph = P->height;
    padTiling(pw, ph, P->ppllTileW, P->ppllTileH);
    std::vector<uint32_t> nodes(3 * size_t(P->ppllLinkedListSize));
    std::vector<uint32_t> startOffset(size_t(pw) * ph);
    uint32_t fragCounter = 0;
    lvo_ppll_gather(sc, P, useBvh, ao, x0, y0, w, h, nodes.data(), startOffset.data(), &fragCounter, stats);
    lvo_ppll_resolve(P, nodes.data(), startOffset.data(), 0, x0, y0, w, h, outRGBA8);
}

// Test hook: computeFragmentColor (RayHitCommon.glsl:74-543 + blinnPhongShadingTube, Lighting.glsl:100-191) on n
// independent inputs -- fragPos / normal / tangent are n x 3, isCap / attribute / aoTexel n each; outputs hitColor n x 4 and
// payload.hitT n.  Lets tests compare this restatement with an independently written one (tests/test_independent_*.py).
void lvo_compute_fragment_color_batch(const lvo_scene* sc, const lvo_params* Pp, uint64_t n, const float* fragPos,
                                      const float* normal, const float* tangent, const uint32_t* isCap, const float* attribute,
                                      const float* aoTexel, float* outColor, float* outHitT) {
    const lvo_params& P = *Pp;
    Frame F = makeFrame(P);
    for (uint64_t i = 0; i < n; i++) {
        float c[4], hitT;
        computeFragmentColor(*sc, P, F, aoTexel[i], ld3(fragPos + 3 * i), ld3(normal + 3 * i), ld3(tangent + 3 * i),
                             isCap[i] != 0u, attribute[i], c, hitT);
        for (int k = 0; k < 4; k++) outColor[4 * i + k] = c[k];
        outHitT[i] = hitT;
    }
}

// Test hooks of the raster fragment-colour variant (LinePassGeometryShaderTubes.glsl:785-815,1079-1087): computeFragmentColor with
// EPSILON_OUTLINE = 0 and EPSILON_WHITE = epsWhite[i]; the ribbon coordinate of n viewing rays with respect to one tube axis
// (capHit = null: tubeRibbonOfRay; else capRibbonOfRay about the cap fragment capHit / capNormal on the sphere at axisPoint) -- what
// fwidth(ribbonPosition) is differenced from.
void lvo_compute_fragment_color_raster_batch(const lvo_scene* sc, const lvo_params* Pp, uint64_t n, const float* fragPos,
                                             const float* normal, const float* tangent, const uint32_t* isCap, const float* attribute,
                                             const float* aoTexel, const float* epsWhite, float* outColor, float* outHitT) {
    const lvo_params& P = *Pp;
    Frame F = makeFrame(P);
    for (uint64_t i = 0; i < n; i++) {
        float c[4], hitT;
        BandArgs rb; rb.shadeBands = false; rb.useBand = false; rb.phi = 0.0f; rb.linePosition = rb.lineNormal = v3(0, 0, 0);
        rb.rasterEpsWhite = epsWhite[i];
        computeFragmentColor(*sc, P, F, aoTexel[i], ld3(fragPos + 3 * i), ld3(normal + 3 * i), ld3(tangent + 3 * i),
                             isCap[i] != 0u, attribute[i], c, hitT, &rb);
        for (int k = 0; k < 4; k++) outColor[4 * i + k] = c[k];
        outHitT[i] = hitT;
    }
}
void lvo_ribbon_of_rays(const float* cam, const float* dirs, uint64_t n, const float* axisPoint, const float* axisDir, float radius,
                        const float* capHit, const float* capNormal, float* out) {
    for (uint64_t i = 0; i < n; i++)
        out[i] = capHit ? capRibbonOfRay(ld3(cam), ld3(dirs + 3 * i), ld3(capHit), ld3(capNormal), ld3(axisPoint), ld3(axisDir))
                        : tubeRibbonOfRay(ld3(cam), ld3(dirs + 3 * i), ld3(axisPoint), ld3(axisDir), radius);
}
// Test hook: getAoFactor(interpolatedVertexId, phi) of the static prebaker (AmbientOcclusion.glsl:49-75 up to the mix of the four
// table entries; pow(gamma) / strength are applied by the shading code) on n inputs
void lvo_prebaked_ao_lookup_batch(const float* factors, const float* blendingWeights, uint32_t numLineVertices,
                                  uint32_t numParametrizationVertices, uint32_t numAoTubeSubdivisions, const float* vertexId,
                                  const float* phi, uint64_t n, float* out) {
    PrebakedAo pb{factors, blendingWeights, numLineVertices, numParametrizationVertices, numAoTubeSubdivisions};
    for (uint64_t i = 0; i < n; i++) out[i] = prebakedAoLookup(pb, vertexId[i], phi[i]);
}
void lvo_pow_det(const float* x, const float* y, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; i++) out[i] = powDet(x[i], y[i]);
}
void lvo_set_ppll_fragment_colour_variant(int rayTracerVariant) { g_rtFragmentColourInPpll = rayTracerVariant != 0; }
void lvo_set_ppll_prebaked_ao(const float* factors, const float* blendingWeights, uint32_t numLineVertices,
                              uint32_t numParametrizationVertices, uint32_t numAoTubeSubdivisions) {
    g_ppllPrebaked = PrebakedAo{factors, blendingWeights, numLineVertices, numParametrizationVertices, numAoTubeSubdivisions};
}

void lvo_set_ao_feature_outputs(float* normalMap, float* positionMap) {
    g_lvoAoFeatures.normal = normalMap;
    g_lvoAoFeatures.position = positionMap;
}

void lvo_set_svgf_feature_outputs(float* normalWorld, float* depth, float* flow, float* depthFwidth, int enable,
                                  uint32_t globalFrameNumber, const float* lastFrameViewProj) {
    g_lvoAoFeatures.normalWorld = normalWorld;
    g_lvoAoFeatures.depth = depth;
    g_lvoAoFeatures.flow = flow;
    g_lvoAoFeatures.depthFwidth = depthFwidth;
    g_lvoAoFeatures.svgf = enable;
    g_lvoAoFeatures.globalFrameNumber = globalFrameNumber;
    for (int i = 0; i < 16; i++) g_lvoAoFeatures.lastFrameViewProj[i] = lastFrameViewProj ? lastFrameViewProj[i] : 0.0f;
}

// glm operator*(mat4, mat4), column major: column j of the product = sum over k of A's column k times B[j][k], left to right
// (lastFrameViewProjectionMatrix = projection * view, VulkanRayTracedAmbientOcclusion.cpp:456,631)
void lvo_mat4_mul(const float* A, const float* B, float* out) {
    float r[16];
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++)
            r[4 * j + i] = ((A[i] * B[4 * j] + A[4 + i] * B[4 * j + 1]) + A[8 + i] * B[4 * j + 2]) + A[12 + i] * B[4 * j + 3];
    memcpy(out, r, sizeof r);
}

// ---------------------------------------------------------------- SVGF (Data/Shaders/Denoiser/SVGF.glsl, svgf_common.glsl;
// src/Renderers/Scattering/Denoiser/SVGF.cpp) on the AO image.  noisy_texture = vec4(ao, ao, ao, 1): the three colour channels
// stay equal through every pass, so the colour images are {colour, variance} pairs.  One call = SVGFDenoiser::denoise():
// reproject -> filter moments -> `iterations` a-trous passes -> history copies (SVGF.cpp:107-174).  The history images are the
// caller's (cleared to 0 = recreateSwapchain, SVGF.cpp:181-286) and updated in place.
// Build-owned definitions where the reference leaves the result open:
//   * texelFetch outside the image (filter_variance at the image border, SVGF.glsl:368-391) returns 0;
//   * the moments filter reads temp_accum while other invocations overwrite it (SVGF.glsl:300-352, a data race): every
//     invocation reads the image as the reprojection pass left it;
//   * `out` parameters the callee did not write (prev_moments when load_moments_and_history_length fails) are 0;
//   * pow(x, 128) of compute_weight is seven squarings (exact products instead of exp2(128 log2 x)).
namespace {
inline float svgfPow128(float x) { for (int i = 0; i < 7; i++) x = x * x; return x; }
// svgf_common.glsl:28-40
inline float svgfComputeWeight(float centerDepth, float offsetDepth, float phiDepth, const float* centerNormal,
                               const float* offsetNormal, float centerColor, float offsetColor, float phiColor) {
    const float weightN = svgfPow128(fmaxf(0.0f, (centerNormal[0] * offsetNormal[0] + centerNormal[1] * offsetNormal[1]) +
                                                     centerNormal[2] * offsetNormal[2]));
    const float weightZ = (phiDepth == 0.0f) ? 0.0f : fabsf(centerDepth - offsetDepth) / phiDepth;
    const float weightC = fabsf(centerColor - offsetColor) * 2.0f / phiColor;
    return expf((0.0f - fmaxf(weightC, 0.0f)) - fmaxf(weightZ, 0.0f)) * weightN;
}
} // namespace

void lvo_svgf_denoise(uint32_t width, uint32_t height, const float* noisy, const float* normalMap, const float* depthMap,
                      const float* depthFwidthMap, const float* flowMap, int iterations, float allowedZDist,
                      float allowedNormalDist, float* colorHistory, float* momentsHistory, float* normalHistory,
                      float* depthHistory, float* out) {
    const int W = int(width), H = int(height);
    const size_t n = size_t(width) * height;
    std::vector<float> tempAccum(2 * n), accumMoments(4 * n, 0.0f);
    // is_reprj_valid, SVGF.glsl:72-86 (the bounds test comes first: the history texels of rejected coordinates are never used)
    auto reprjValid = [&](int cx, int cy, float z, const float* normal) {
        if (cx < 1 || cy < 1 || cx > W - 1 || cy > H - 1) return false;
        const size_t oi = size_t(cy) * width + cx;
        const float zPrev = depthHistory[oi];
        const float* np = normalHistory + 4 * oi;
        if (fabsf(zPrev - z) > allowedZDist) return false;
        const float dx = np[0] - normal[0], dy = np[1] - normal[1], dz = np[2] - normal[2];
        if (sqrtf((dx * dx + dy * dy) + dz * dz) > allowedNormalDist) return false;
        return true;
    };
    // ---- Compute-Reproject, SVGF.glsl:201-261
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t yy = 0; yy < int64_t(H); yy++) {
        for (int x = 0; x < W; x++) {
            const int y = int(yy);
            const size_t ci = size_t(y) * width + x;
            const float mx = flowMap[2 * ci], my = flowMap[2 * ci + 1];
            float prevM0 = 0.0f, prevM1 = 0.0f, historyLength = 0.0f;
            const int ipx = int((0.5f + float(x)) - mx), ipy = int((0.5f + float(y)) - my);
            bool success = !(ipx < 0 || ipy < 0 || ipx >= W || ipy >= H); // load_moments_and_history_length, :186-199
            if (success) {
                const float* mh = momentsHistory + 4 * (size_t(ipy) * width + ipx);
                prevM0 = mh[0]; prevM1 = mh[1]; historyLength = mh[2];
            }
            const float color = noisy[ci];
            float colorLastFrame = colorHistory[ci];
            if (success) {
                const float ppx = (0.01f + float(x)) - mx, ppy = (0.01f + float(y)) - my;
                const int qx = int(ppx), qy = int(ppy);
                const float depth = depthMap[ci];
                const float* normal = normalMap + 4 * ci;
                // try_2x2_tap, :88-139 (offsets and weights exactly as listed there)
                const int offs[4][2] = {{0, 0}, {0, 1}, {1, 0}, {1, 1}};
                bool valids[4], validFound = false;
                for (int i = 0; i < 4; i++) {
                    valids[i] = reprjValid(qx + offs[i][0], qy + offs[i][1], depth, normal);
                    validFound = validFound || valids[i];
                }
                if (validFound) {
                    const float fx = ppx - floorf(ppx), fy = ppy - floorf(ppy);
                    const float w[4] = {(1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy};
                    float colorBilinear = 0.0f, m0 = 0.0f, m1 = 0.0f, sumW = 0.0f;
                    for (int i = 0; i < 4; i++) {
                        if (!valids[i]) continue;
                        const size_t oi = size_t(qy + offs[i][1]) * width + (qx + offs[i][0]);
                        m0 += w[i] * momentsHistory[4 * oi];
                        m1 += w[i] * momentsHistory[4 * oi + 1];
                        colorBilinear += w[i] * colorHistory[oi];
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
                            if (reprjValid(ox, oy, depth, normal)) {
                                const size_t oi = size_t(oy) * width + ox;
                                fc += colorHistory[oi];
                                f0 += momentsHistory[4 * oi];
                                f1 += momentsHistory[4 * oi + 1];
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
            float* am = accumMoments.data() + 4 * ci;
            am[0] = r; am[1] = g; am[2] = historyLength; am[3] = 0.0f;
            tempAccum[2 * ci] = mixf(colorLastFrame, color, alphaColor);
            tempAccum[2 * ci + 1] = variance;
        }
    }
    // ---- Compute-Filter-Moments, SVGF.glsl:282-352
    {
        const std::vector<float> src(tempAccum);
#pragma omp parallel for schedule(dynamic, 8)
        for (int64_t yy = 0; yy < int64_t(H); yy++) {
            for (int x = 0; x < W; x++) {
                const int y = int(yy);
                const size_t ci = size_t(y) * width + x;
                const float historyLength = accumMoments[4 * ci + 2];
                if (historyLength >= 4.0f) continue;
                float sumWeight = 0.0f, sumColor = 0.0f, sumM0 = 0.0f, sumM1 = 0.0f;
                const float centerColor = src[2 * ci], centerDepth = depthMap[ci], centerFwidth = depthFwidthMap[ci];
                const float* centerNormal = normalMap + 4 * ci;
                for (int dy = -3; dy <= 3; dy++) {
                    for (int dx = -3; dx <= 3; dx++) {
                        const int ox = x + dx, oy = y + dy;
                        if (!(ox >= 0 && oy >= 0 && ox < W && oy < H)) continue;
                        const size_t oi = size_t(oy) * width + ox;
                        const float weight = svgfComputeWeight(centerDepth, depthMap[oi], fabsf(centerFwidth) + 0.0001f, centerNormal,
                                                               normalMap + 4 * oi, centerColor, src[2 * oi], 10.0f);
                        sumWeight += weight;
                        sumColor += weight * src[2 * oi];
                        sumM0 += weight * accumMoments[4 * oi];
                        sumM1 += weight * accumMoments[4 * oi + 1];
                    }
                }
                sumWeight = fmaxf(sumWeight, 1e-6f);
                sumColor /= sumWeight; sumM0 /= sumWeight; sumM1 /= sumWeight;
                float variance = sumM1 - sumM0 * sumM0;
                variance *= 4.0f / historyLength;
                tempAccum[2 * ci] = sumColor;
                tempAccum[2 * ci + 1] = variance;
            }
        }
    }
    // ---- Compute-ATrous x iterations, SVGF.glsl:393-495, SVGF.cpp:346-425
    std::vector<float> ping(tempAccum), pong(2 * n);
    if (iterations < 1) {
        for (size_t i = 0; i < n; i++) colorHistory[i] = tempAccum[2 * i]; // blits, SVGF.cpp:347-360
    }
    for (int it = 0; it < iterations; it++) {
        const int stepWidth = 1 << it;
        const float* src = ping.data();
        float* dst = pong.data();
#pragma omp parallel for schedule(dynamic, 8)
        for (int64_t yy = 0; yy < int64_t(H); yy++) {
            for (int x = 0; x < W; x++) {
                const int y = int(yy);
                const size_t ci = size_t(y) * width + x;
                const float centerColor = src[2 * ci], centerVar = src[2 * ci + 1];
                // filter_variance: 3x3 Gaussian of the variance channel, :368-391
                float fv = 0.0f;
                const float vk[2][2] = {{1.0f / 4.0f, 1.0f / 8.0f}, {1.0f / 8.0f, 1.0f / 16.0f}};
                for (int dy = -1; dy <= 1; dy++) {
                    for (int dx = -1; dx <= 1; dx++) {
                        const int px = x + dx, py = y + dy;
                        const float v = (px >= 0 && py >= 0 && px < W && py < H) ? src[2 * (size_t(py) * width + px) + 1] : 0.0f;
                        fv += v * vk[std::abs(dx)][std::abs(dy)];
                    }
                }
                const float* centerNormal = normalMap + 4 * ci;
                const float centerZ = depthMap[ci], centerFwidth = depthFwidthMap[ci];
                const float phiColor = sqrtf(fmaxf(0.0f, 1e-10f + fv));
                const float kv[3] = {1.0f, 2.0f / 3.0f, 1.0f / 6.0f};
                float accumW = kv[0] * kv[0];
                float sumC = centerColor * accumW, sumV = centerVar * accumW;
                for (int dy = -2; dy <= 2; ++dy) {
                    for (int dx = -2; dx <= 2; ++dx) {
                        const int ox = x + dx * stepWidth, oy = y + dy * stepWidth;
                        const bool inside = ox >= 0 && oy >= 0 && ox < W && oy < H;
                        if (!inside || (dx == 0 && dy == 0)) continue;
                        const size_t oi = size_t(oy) * width + ox;
                        const float kernelValue = kv[std::abs(dx)] * kv[std::abs(dy)];
                        const float len = sqrtf(float(dx) * float(dx) + float(dy) * float(dy));
                        const float weight = svgfComputeWeight(centerZ, depthMap[oi], fabsf((centerFwidth * len) * float(stepWidth)) + 0.0001f,
                                                               centerNormal, normalMap + 4 * oi, centerColor, src[2 * oi], phiColor) * kernelValue;
                        sumC += weight * src[2 * oi];
                        sumV += (weight * weight) * src[2 * oi + 1];
                        accumW += weight;
                    }
                }
                dst[2 * ci] = sumC / accumW;
                dst[2 * ci + 1] = sumV / (accumW * accumW);
                if (it == 0) colorHistory[ci] = dst[2 * ci];
            }
        }
        ping.swap(pong);
    }
    for (size_t i = 0; i < n; i++) out[i] = ping[2 * i];
    // "update previous frame images", SVGF.cpp:112-173
    memcpy(normalHistory, normalMap, n * 16);
    memcpy(depthHistory, depthMap, n * 4);
    memcpy(momentsHistory, accumMoments.data(), n * 16);
}

// EAWDenoise.glsl on the AO image (colorTexture = vec4(ao, ao, ao, 1): the three colour channels stay equal and alpha stays 1
// through every pass, so one float per pixel carries the image), feature maps as float4 per pixel.
//   computeVariant != 0  EAWDenoise.Compute (:128-292, the default: eaw_denoiser_use_shared_memory = true): B-spline kernel
//                        {1, 2/3, 1/6}, neighbours outside the image skipped, ONE exp of the summed exponents, the colour term
//                        scaled by the step width;
//   computeVariant == 0  EAWDenoise.Fragment (:16-126): Gaussian kernel exp(-(x^2 + y^2) / 2), clamp-to-edge sampling, one
//                        min(exp(.), 1) per enabled feature.
// Passes: step width 1, 2, 4, ... (EAWDenoiser.cpp:316-347,355-395), ping-pong; region = the pixels to compute (inputs must be
// valid 2 * (2^iterations - 1) pixels around it); "inside" always refers to the whole viewport.
void lvo_eaw_denoise(uint32_t width, uint32_t height, const float* ao, const float* normalMap, const float* positionMap,
                     int iterations, float phiColor, float phiPosition, float phiNormal, int useColor, int usePosition,
                     int useNormal, int computeVariant, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, float* out) {
    const size_t n = size_t(width) * height;
    std::vector<float> ping(ao, ao + n), pong(ao, ao + n);
    const bool wC = useColor != 0, wP = usePosition != 0 && positionMap, wN = useNormal != 0 && normalMap;
    int stepWidth = 1;
    for (int it = 0; it < iterations; it++) {
        const float* src = ping.data();
        float* dst = pong.data();
#pragma omp parallel for schedule(dynamic, 8)
        for (int64_t yy = 0; yy < int64_t(h); yy++) {
            for (uint32_t xx = 0; xx < w; xx++) {
                const int gx = int(x0 + xx), gy = int(y0) + int(yy);
                const size_t ci = size_t(gy) * width + gx;
                const float centerColor = src[ci];
                const float* cP = positionMap ? positionMap + 4 * ci : nullptr;
                const float* cN = normalMap ? normalMap + 4 * ci : nullptr;
                auto dist4 = [](const float* a, const float* b) {
                    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2], dw = a[3] - b[3];
                    return ((dx * dx + dy * dy) + dz * dz) + dw * dw;
                };
                float sum, accumW;
                if (computeVariant) {
                    const float kernelValues[3] = {1.0f, 2.0f / 3.0f, 1.0f / 6.0f};
                    accumW = kernelValues[0] * kernelValues[0];
                    sum = centerColor * accumW;
                    for (int y = -2; y <= 2; ++y) {
                        for (int x = -2; x <= 2; ++x) {
                            const int ox = gx + x * stepWidth, oy = gy + y * stepWidth;
                            const bool inside = ox >= 0 && oy >= 0 && ox < int(width) && oy < int(height);
                            if (!inside || (x == 0 && y == 0)) continue;
                            const size_t oi = size_t(oy) * width + ox;
                            const float kernelValue = kernelValues[std::abs(x)] * kernelValues[std::abs(y)];
                            const float offsetColor = src[oi];
                            float e = 0.0f;
                            if (wC) {
                                const float d = centerColor - offsetColor;
                                const float distColor = ((d * d + d * d) + d * d) + 0.0f * 0.0f;
                                e = e - (distColor * float(stepWidth)) / phiColor;
                            }
                            if (wP) e = e - dist4(cP, positionMap + 4 * oi) / phiPosition;
                            if (wN) e = e - dist4(cN, normalMap + 4 * oi) / phiNormal;
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
                        const int ox = std::min(std::max(gx + (i % 5 - 2) * stepWidth, 0), int(width) - 1);
                        const int oy = std::min(std::max(gy + (i / 5 - 2) * stepWidth, 0), int(height) - 1);
                        const size_t oi = size_t(oy) * width + ox;
                        const float offsetColor = src[oi];
                        float weight = 1.0f;
                        if (wC) {
                            const float d = centerColor - offsetColor;
                            const float distColor = ((d * d + d * d) + d * d) + 0.0f * 0.0f;
                            weight *= fminf(expf(-distColor / phiColor), 1.0f);
                        }
                        if (wP) weight *= fminf(expf(-dist4(cP, positionMap + 4 * oi) / phiPosition), 1.0f);
                        if (wN) weight *= fminf(expf(-dist4(cN, normalMap + 4 * oi) / phiNormal), 1.0f);
                        sum += (offsetColor * weight) * kernelValue;
                        accumW += weight * kernelValue;
                    }
                }
                dst[ci] = sum / accumW;
            }
        }
        ping.swap(pong);
        stepWidth *= 2;
    }
    for (uint32_t yy = 0; yy < h; yy++)
        for (uint32_t xx = 0; xx < w; xx++) {
            const size_t ci = size_t(y0 + yy) * width + (x0 + xx);
            out[ci] = ping[ci];
        }
}

void lvo_set_deviation_switches(int literalIntersection, int referenceAoLookup) {
    g_dev.literalIntersection = literalIntersection != 0;
    g_dev.referenceAoLookup = referenceAoLookup != 0;
}

// threads the OpenMP loops above run on (reported next to the CPU baseline)
int lvo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}


/* ---- a16: fragments of the rasterised programmable-pull prism (lv_oracle_prism.h), test hooks ---- */
// test hook: the next lvo_prism_ring_vertices calls use the USE_BANDS ring (lineWidth then means the band width)
static struct { int use; float thickness; } g_ringHookBands = {0, 1.0f};
void lvo_set_prism_ring_bands(int useBands, float thickness) { g_ringHookBands.use = useBands; g_ringHookBands.thickness = thickness; }
void lvo_prism_ring_vertices(const lvo_line_point* pts, uint64_t nPts, uint32_t numSubdivisions, float lineWidth, float* outPos,
                             float* outNormal) {
    const PrismRing R = g_ringHookBands.use ? prismRing(numSubdivisions, lineWidth * 0.5f, true, g_ringHookBands.thickness)
                                            : prismRing(numSubdivisions, lineWidth * 0.5f);
    for (uint64_t i = 0; i < nPts; i++)
        for (uint32_t k = 0; k < R.n; k++) {
            const PrismVtx v = prismVertex(pts[i], R, k, lineWidth * 0.5f);
            const V3 n = normalizeShade(v.dir);
            float* p = outPos + 3 * (i * R.n + k);
            float* q = outNormal + 3 * (i * R.n + k);
            p[0] = v.pos.x; p[1] = v.pos.y; p[2] = v.pos.z;
            q[0] = n.x; q[1] = n.y; q[2] = n.z;
        }
}
/* offsets[w * h + 1]; with segs == NULL only the offsets are filled (call twice).  Per fragment: segment, triangle (< 2 N), the three
 * weights, depth, interpolated position / normal / tangent (3 floats each, may be NULL), attribute, packed colour and alpha. */
// the coverage direction of pixel (x, y) (prismCoverageDir) next to the ray generator's normalised direction (primaryRay): what
// tests/test_prism_raster.py compares in float64
void lvo_prism_coverage_dir(const lvo_params* Pp, uint32_t x, uint32_t y, float* coverageDir, float* rayDir) {
    const lvo_params& P = *Pp;
    const Frame F = makeFrame(P);
    const V3 D = prismCoverageDir(P, F, x, y);
    V3 o, d;
    primaryRay(P, F, x, y, 0.5f, 0.5f, o, d);
    coverageDir[0] = D.x; coverageDir[1] = D.y; coverageDir[2] = D.z;
    rayDir[0] = d.x; rayDir[1] = d.y; rayDir[2] = d.z;
}
void lvo_prism_fragments(const lvo_scene* sc, const lvo_params* Pp, int useBvh, const float* ao, uint32_t x0, uint32_t y0, uint32_t w,
                         uint32_t h, uint64_t* offsets, uint32_t* segs, uint32_t* tris, float* weights, float* depth, float* pos,
                         float* nrm, float* tan, float* attr, uint32_t* colour, float* rgba) {
    const lvo_params& P = *Pp;
    const Frame F = makeFrame(P);
    const PrismRing ring = prismRingOf(P, F);
    g_dev.aoImage = (g_dev.referenceAoLookup && P.useAmbientOcclusion) ? ao : nullptr;
    std::vector<std::vector<PrismFrag>> rows(size_t(w) * h);
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t yy = 0; yy < int64_t(h); yy++) {
        Counters cnt;
        std::vector<uint32_t> cand;
        for (uint32_t xx = 0; xx < w; xx++)
            prismPixelFragments(*sc, P, F, ring, useBvh != 0, x0 + xx, y0 + uint32_t(yy), cand, rows[size_t(yy) * w + xx], cnt);
    }
    uint64_t k = 0;
    for (size_t i = 0; i < rows.size(); i++) {
        offsets[i] = k;
        if (segs) {
            const uint32_t x = x0 + uint32_t(i % w), y = y0 + uint32_t(i / w);
            const float aoTexel = (P.useAmbientOcclusion && ao) ? ao[size_t(y) * P.width + x] : 1.0f;
            const RasterQuad rq = makeRasterQuad(P, F, x, y);
            for (const PrismFrag& f : rows[i]) {
                segs[k] = f.seg; tris[k] = f.tri;
                for (int j = 0; j < 3; j++) weights[3 * k + j] = f.b[j];
                depth[k] = f.depth;
                if (pos) { pos[3 * k] = f.pos.x; pos[3 * k + 1] = f.pos.y; pos[3 * k + 2] = f.pos.z; }
                if (nrm) { nrm[3 * k] = f.nrm.x; nrm[3 * k + 1] = f.nrm.y; nrm[3 * k + 2] = f.nrm.z; }
                if (tan) { tan[3 * k] = f.tan.x; tan[3 * k + 1] = f.tan.y; tan[3 * k + 2] = f.tan.z; }
                if (attr) attr[k] = f.attr;
                if (colour || rgba) {
                    float hc[4]; float hitT;
                    prismShade(*sc, P, F, ring, aoTexel, f, g_rtFragmentColourInPpll ? nullptr : &rq, hc, hitT,
                               g_ppllPrebaked.factors ? &g_ppllPrebaked : nullptr);
                    if (colour) colour[k] = packUnorm4x8(hc);
                    if (rgba) for (int j = 0; j < 4; j++) rgba[4 * k + j] = hc[j];
                }
                k++;
            }
        } else {
            k += rows[i].size();
        }
    }
    offsets[rows.size()] = k;
}

} // extern "C"
