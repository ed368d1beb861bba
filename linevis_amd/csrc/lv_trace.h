// lv_trace.h -- LBVH traversal, ray-capsule intersection and hit shading (device functions).
//
// Replaces the VK_KHR_ray_tracing traversal done by the Vulkan driver plus the GLSL programs
//   IntersectionTube           Data/Shaders/Renderers/RayTracing/TubeRayTracing.glsl:452-494
//   rayTubeIntersection        Data/Shaders/Renderers/RayTracing/RayIntersectionTestsVulkan.glsl:78-119
//   raySphereIntersection      Data/Shaders/Renderers/RayTracing/RayIntersectionTestsVulkan.glsl:39-72
//   ClosestHitTubeAnalytic     Data/Shaders/Renderers/RayTracing/TubeRayTracing.glsl:512-613
//   computeFragmentColor       Data/Shaders/Renderers/RayTracing/RayHitCommon.glsl:74-543
//   blinnPhongShadingTube      Data/Shaders/Utils/Lighting.glsl:100-191
//   transferFunction           Data/Shaders/Utils/TransferFunction.glsl:66-71
//   getAoFactor                Data/Shaders/Utils/AmbientOcclusion.glsl:84-99
#pragma once

#include "lv_device.h"

// ---------------------------------------------------------------- ray-capsule
// Both quadratics of RayIntersectionTestsVulkan.glsl are evaluated in the closest-approach form (t_c -+ h with
// h = sqrt((r^2 - l.l) / A), l = perpendicular offset at t_c; Ray Tracing Gems ch. 7) instead of the reference's
// (-B -+ sqrt(B^2 - 4AC)) / 2A: for r = 1e-3 at distance 1 the textbook discriminant keeps 1-2 digits in float32 and
// t jitters by ~0.1 r, which no conservative BVH culling can reproduce.  Same roots, same selection logic.
__device__ __forceinline__ bool lv_ray_sphere(f3 o, f3 d, f3 ctr, float radius, float& hitT) {
    f3 f = o - ctr;
    float A = dot3(d, d);
    float tc = -dot3(f, d) / A;
    f3 l = f + tc * d;
    float disc = radius * radius - dot3(l, l);
    if (disc < 0.0f) return false;
    float h = sqrtf(disc / A);
    float t0 = tc - h;
    float t1 = tc + h;
    hitT = t0;
    if (t0 >= 0.0f) hitT = t0;
    else if (t1 >= 0.0f) hitT = t1;
    else return false;
    return true;
}

// td = normalize(tubeEnd - tubeStart): per test (lv_ray_tube) or precomputed per segment with the same norm3 (S.segAxis)
__device__ __forceinline__ bool lv_ray_tube_td(f3 o, f3 d, f3 tubeStart, f3 tubeEnd, f3 td, float radius, float& hitT) {
    f3 deltaP = o - tubeStart;
    f3 av = d - dot3(d, td) * td;
    f3 cv = deltaP - dot3(deltaP, td) * td;
    float A = dot3(av, av);
    float tc = -dot3(av, cv) / A;
    f3 l = cv + tc * av;
    float disc = radius * radius - dot3(l, l);
    if (disc < 0.0f) return false;
    float h = sqrtf(disc / A);
    float t0 = tc - h;
    if (t0 >= 0.0f) {
        f3 ip = o + t0 * d;
        if (dot3(td, ip - tubeStart) > 0.0f && dot3(td, ip - tubeEnd) < 0.0f) { hitT = t0; return true; }
    }
    float t1 = tc + h;
    if (t1 >= 0.0f) {
        f3 ip = o + t1 * d;
        if (dot3(td, ip - tubeStart) > 0.0f && dot3(td, ip - tubeEnd) < 0.0f) { hitT = t1; return true; }
    }
    return false;
}

__device__ __forceinline__ bool lv_ray_tube(f3 o, f3 d, f3 tubeStart, f3 tubeEnd, float radius, float& hitT) {
    return lv_ray_tube_td(o, d, tubeStart, tubeEnd, norm3(tubeEnd - tubeStart), radius, hitT);
}

// IntersectionTube main(): nearest of {cylinder, sphere(p0), sphere(p1)}; kind 0/1/2
__device__ __forceinline__ bool lv_intersect_capsule_td(f3 o, f3 d, f3 p0, f3 p1, f3 td, float radius, bool capped,
                                                        float& hitTOut, int& kindOut) {
    bool has = false;
    float hitT = 1e7f;
    int kind = 0;
    float tubeT;
    if (lv_ray_tube_td(o, d, p0, p1, td, radius, tubeT)) { hitT = tubeT; has = true; kind = 0; }
    if (capped) {
        float s0T, s1T;
        bool h0 = lv_ray_sphere(o, d, p0, radius, s0T);
        bool h1 = lv_ray_sphere(o, d, p1, radius, s1T);
        if (h0 && s0T < hitT) { has = true; hitT = s0T; kind = 1; }
        if (h1 && s1T < hitT) { has = true; hitT = s1T; kind = 2; }
    }
    hitTOut = hitT;
    kindOut = kind;
    return has;
}
__device__ __forceinline__ bool lv_intersect_capsule(f3 o, f3 d, f3 p0, f3 p1, float radius, bool capped,
                                                     float& hitTOut, int& kindOut) {
    return lv_intersect_capsule_td(o, d, p0, p1, norm3(p1 - p0), radius, capped, hitTOut, kindOut);
}

// intersection_form = literal: the reference's roots exactly as written -- t = (-B -+ sqrt(B^2 - 4AC)) / 2A,
// RayIntersectionTestsVulkan.glsl:39-72 (sphere) and :78-119 (tube) -- in float32 without contraction.  For r = 1e-3 at
// distance ~1 the discriminant keeps 1-2 digits, t carries up to ~0.25 r of noise and silhouettes flicker by hits that
// should be misses and vice versa (tests/test_deviations.py: 34-43 % of the covered pixels of configs 2 / 3 differ by more
// than 2 LSB between the two forms).  This mode reproduces the formula, noise included.
__device__ __forceinline__ bool lv_ray_sphere_literal(f3 o, f3 d, f3 ctr, float radius, float& hitT) {
    const float A = (d.x * d.x + d.y * d.y) + d.z * d.z;
    const float B = 2.0f * ((d.x * (o.x - ctr.x) + d.y * (o.y - ctr.y)) + d.z * (o.z - ctr.z));
    const float C = (((o.x - ctr.x) * (o.x - ctr.x) + (o.y - ctr.y) * (o.y - ctr.y)) + (o.z - ctr.z) * (o.z - ctr.z)) - radius * radius;
    const float discriminant = B * B - (4.0f * A) * C;
    if (discriminant < 0.0f) return false;
    const float ds = sqrtf(discriminant);
    const float t0 = (-B - ds) / (2.0f * A);
    const float t1 = (-B + ds) / (2.0f * A);
    hitT = t0;
    if (t0 >= 0.0f) hitT = t0;
    else if (t1 >= 0.0f) hitT = t1;
    else return false;
    return true;
}
__device__ __forceinline__ bool lv_ray_tube_literal(f3 o, f3 d, f3 tubeStart, f3 tubeEnd, float radius, float& hitT) {
    const f3 td = norm3(tubeEnd - tubeStart);
    const f3 deltaP = o - tubeStart;
    const f3 av = d - dot3(d, td) * td;
    const f3 cv = deltaP - dot3(deltaP, td) * td;
    const float A = (av.x * av.x + av.y * av.y) + av.z * av.z;
    const float B = 2.0f * dot3(av, cv);
    const float C = ((cv.x * cv.x + cv.y * cv.y) + cv.z * cv.z) - radius * radius;
    const float discriminant = B * B - (4.0f * A) * C;
    if (discriminant < 0.0f) return false;
    const float ds = sqrtf(discriminant);
    const float t0 = (-B - ds) / (2.0f * A);
    if (t0 >= 0.0f) {
        const f3 ip = o + t0 * d;
        if (dot3(td, ip - tubeStart) > 0.0f && dot3(td, ip - tubeEnd) < 0.0f) { hitT = t0; return true; }
    }
    const float t1 = (-B + ds) / (2.0f * A);
    if (t1 >= 0.0f) {
        const f3 ip = o + t1 * d;
        if (dot3(td, ip - tubeStart) > 0.0f && dot3(td, ip - tubeEnd) < 0.0f) { hitT = t1; return true; }
    }
    return false;
}
// The intersection shader only runs for rays that hit the segment's AABB (min(p0, p1) - r .. max(p0, p1) + r,
// LineDataFlow.cpp:2230-2233) -- and a root is only meaningful near that box.  Making this part of the test (own box hit by
// the ray's line, t within r / |d| of the box interval) is what lets ANY conservative BVH that culls against best + r / |d|
// return the brute-force minimum of the noisy roots bit for bit (the same device as lv_ray_triangle's own-box rule).
__device__ __forceinline__ bool lv_intersect_capsule_literal(f3 o, f3 d, f3 p0, f3 p1, float radius, bool capped, float& hitTOut,
                                                             int& kindOut) {
    bool has = false;
    float hitT = 1e7f;
    int kind = 0;
    float tubeT;
    if (lv_ray_tube_literal(o, d, p0, p1, radius, tubeT)) { hitT = tubeT; has = true; kind = 0; }
    if (capped) {
        float s0T, s1T;
        const bool h0 = lv_ray_sphere_literal(o, d, p0, radius, s0T);
        const bool h1 = lv_ray_sphere_literal(o, d, p1, radius, s1T);
        if (h0 && s0T < hitT) { has = true; hitT = s0T; kind = 1; }
        if (h1 && s1T < hitT) { has = true; hitT = s1T; kind = 2; }
    }
    hitTOut = hitT;
    kindOut = kind;
    if (!has) return false;
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const float tx0 = ((fminf(p0.x, p1.x) - radius) - o.x) * inv.x, tx1 = ((fmaxf(p0.x, p1.x) + radius) - o.x) * inv.x;
    const float ty0 = ((fminf(p0.y, p1.y) - radius) - o.y) * inv.y, ty1 = ((fmaxf(p0.y, p1.y) + radius) - o.y) * inv.y;
    const float tz0 = ((fminf(p0.z, p1.z) - radius) - o.z) * inv.z, tz1 = ((fmaxf(p0.z, p1.z) + radius) - o.z) * inv.z;
    const float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fminf(tz0, tz1));
    const float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
    const float slack = radius / len3(d);
    return tn <= tf && hitT >= tn - slack && hitT <= tf + slack;
}

// Conservative pre-tests in front of lv_intersect_capsule (they may only say "cannot hit"; fused math is fine because
// they never decide a hit).  A capsule hit needs the ray LINE to pass within r of the segment's axis LINE
// (|w . (d x v)| <= r |d x v|) and within R = |v|/2 + r of the segment's midpoint.  Margins: the products carry at most a
// few ulps of |w||d||v| of rounding error; 4e-6 of that scale (and 1e-4 relative on the radius) is > 30 x that.
__device__ __forceinline__ bool lv_capsule_may_hit_axis(f3 o, f3 d, f3 p0, f3 p1, float radius) {
    const f3 v = p1 - p0, w = p0 - o;
    const f3 n = cross3(d, v);
    const float q = fabsf(dot3(w, n));
    const float n2 = dot3(n, n), w2 = dot3(w, w), d2 = dot3(d, d), v2 = dot3(v, v);
    const float lim = radius * 1.0001f * __builtin_sqrtf(n2) + 4e-6f * __builtin_sqrtf(w2 * d2 * v2) + 1e-30f;
    return q <= lim;
}
__device__ __forceinline__ bool lv_capsule_may_hit_sphere(f3 o, f3 d, f3 p0, f3 p1, float radius) {
    const f3 v = p1 - p0;
    const f3 c = p0 + 0.5f * v;
    const f3 w = c - o;
    const f3 m = cross3(w, d);
    const float d2 = dot3(d, d), w2 = dot3(w, w);
    const float R = (0.5f * __builtin_sqrtf(dot3(v, v)) + radius) * 1.0001f;
    return dot3(m, m) <= R * R * d2 + 1e-5f * (w2 * d2) * R + 1e-30f;
}

// ---------------------------------------------------------------- ray-triangle (triangle tubes of the reference's RTAO)
// The driver's triangle test is unobservable; the build defines it (float32, fixed operation order):
// Moeller-Trumbore without culling -- e1 = v1-v0, e2 = v2-v0, p = d x e2, det = e1.p (0 -> miss), u = ((o-v0).p)/det,
// q = (o-v0) x e1, v = (d.q)/det, t = (e2.q)/det, accepted for u in [0,1], v >= 0, u+v <= 1 -- AND t inside the ray
// interval of the triangle's own padded AABB.  The second rule is what an acceleration structure implies anyway (a
// primitive is only tested when its box is hit); as part of the test it makes every conservative BVH agree with brute
// force bit for bit, even where a grazing hit's t carries more float32 noise than the pad.
__device__ __forceinline__ bool lv_ray_triangle(f3 o, f3 d, f3 inv, f3 v0, f3 v1, f3 v2, float pad, float& tOut, float& uOut,
                                                float& vOut) {
    const f3 e1 = v1 - v0, e2 = v2 - v0;
    const f3 p = cross3(d, e2);
    const float det = dot3(e1, p);
    if (det == 0.0f) return false;
    const float r = 1.0f / det;
    const f3 tv = o - v0;
    const float u = dot3(tv, p) * r;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    const f3 q = cross3(tv, e1);
    const float v = dot3(d, q) * r;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    const float t = dot3(e2, q) * r;
    const float mnx = fminf(fminf(v0.x, v1.x), v2.x) - pad, mxx = fmaxf(fmaxf(v0.x, v1.x), v2.x) + pad;
    const float mny = fminf(fminf(v0.y, v1.y), v2.y) - pad, mxy = fmaxf(fmaxf(v0.y, v1.y), v2.y) + pad;
    const float mnz = fminf(fminf(v0.z, v1.z), v2.z) - pad, mxz = fmaxf(fmaxf(v0.z, v1.z), v2.z) + pad;
    const float tx0 = (mnx - o.x) * inv.x, tx1 = (mxx - o.x) * inv.x;
    const float ty0 = (mny - o.y) * inv.y, ty1 = (mxy - o.y) * inv.y;
    const float tz0 = (mnz - o.z) * inv.z, tz1 = (mxz - o.z) * inv.z;
    const float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fminf(tz0, tz1));
    const float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
    if (!(t >= tn && t <= tf)) return false;
    tOut = t; uOut = u; vOut = v;
    return true;
}

// ---------------------------------------------------------------- traversal stack: LDS-staged, HBM overflow
// Entry i of a thread lives at lds[i * LV_BLOCK] (conflict-free ds_read/write_b32 across the wave).  An LBVH over
// N segments is usually ~log2(N)+10 deep, so LV_STACK_LDS = 32 entries cover the common case entirely in LDS; deeper
// trees (up to 63 key bits + 32 duplicate-index bits) continue in a per-thread column of a global overflow slab that is
// only allocated when the built tree is that deep.  All members are scalars so the struct lives in registers.
// Pointers carry their address space so that pushes/pops compile to ds_write_b32 / ds_read_b32 and
// global_store / global_load; with generic pointers the two pop paths were merged into one flat_load (which waits on
// both the LDS and the vector-memory counter).
typedef __attribute__((address_space(3))) unsigned lv_lds_u32;
typedef __attribute__((address_space(1))) unsigned lv_glb_u32;

template <int NLDS, int STRIDE = LV_BLOCK>
struct LvStackT {
    static constexpr int kLds = NLDS;
    static constexpr int kStride = STRIDE;
    lv_lds_u32* lds;     // &s_stack[threadIdx.x]
    lv_glb_u32* ovf;     // &overflow[global thread], entries strided by ovfStride; may be null if height <= NLDS
    unsigned ovfStride;
    int sp;
    __device__ __forceinline__ void init(unsigned* ldsBase, unsigned* ovfBase, unsigned stride) {
        lds = (lv_lds_u32*)ldsBase; ovf = (lv_glb_u32*)ovfBase; ovfStride = stride; sp = 0;
    }
    __device__ __forceinline__ void push(unsigned v) {
        if (sp < NLDS) lds[sp * STRIDE] = v;
        else ovf[size_t(sp - NLDS) * ovfStride] = v;
        sp++;
    }
    __device__ __forceinline__ unsigned pop() {
        sp--;
        unsigned v;
        if (sp < NLDS) v = lds[sp * STRIDE];
        else v = ovf[size_t(sp - NLDS) * ovfStride];
        return v;
    }
};
typedef LvStackT<LV_STACK_LDS> LvStack;
// stack memory handed to the traversal routines by a kernel
struct LvStackMem {
    unsigned* lds;
    unsigned* ovf;
    unsigned ovfStride;
};
__device__ __forceinline__ LvStackMem lv_stack_mem(unsigned* sStack, unsigned* ovfSlab) {
    LvStackMem m;
    m.lds = sStack + threadIdx.x;
    m.ovfStride = gridDim.x * LV_BLOCK;
    m.ovf = ovfSlab ? ovfSlab + (size_t(blockIdx.x) * LV_BLOCK + threadIdx.x) : nullptr;
    return m;
}

// Slab test of one child box against the ray.  Box culling only has to be conservative (boxes are padded at build
// time and the acceptance test carries a relative + absolute margin), it never decides a hit, so it is the one place
// that uses fused multiply-adds: t = b * (1/d) - o * (1/d) with oi = o * (1/d) precomputed per ray.
__device__ __forceinline__ bool lv_slab(float bx0, float by0, float bz0, float bx1, float by1, float bz1, f3 oi, f3 inv,
                                        float tMin, float tMax, float& tNear) {
    float tx0 = __builtin_fmaf(bx0, inv.x, -oi.x), tx1 = __builtin_fmaf(bx1, inv.x, -oi.x);
    float ty0 = __builtin_fmaf(by0, inv.y, -oi.y), ty1 = __builtin_fmaf(by1, inv.y, -oi.y);
    float tz0 = __builtin_fmaf(bz0, inv.z, -oi.z), tz1 = __builtin_fmaf(bz1, inv.z, -oi.z);
    float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tMin));
    float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), tMax));
    tNear = tn;
    return tn <= __builtin_fmaf(tf, 1.000001f, 2e-7f);
}

struct LvHit {
    float t;
    uint32_t leaf; // capsules: leaf position (Morton order); triangles: original triangle index
    int kind;
    bool found;
};

// Traversal is organised as "while-while" with postponed leaves (Aila & Laine, "Understanding the Efficiency of Ray
// Traversal on GPUs", HPG 2009) for wave64: a lane that reaches a leaf parks it in `pending` and keeps descending;
// the wave leaves the node loop only when every still-descending lane has parked a leaf, so the expensive capsule test
// (8 IEEE divisions + 4 square roots) runs with most lanes active instead of once per node step for a handful of lanes.
// Child references: index | LV_LEAF_BIT for leaves, LV_INVALID (which has the leaf bit set) = finished.
template <class STACK>
__device__ __forceinline__ unsigned lv_pop_or_done(STACK& st) { return st.sp == 0 ? LV_INVALID : st.pop(); }

// Evaluated by the lanes still inside the node loop: leave it when every one of them has parked a leaf, or when
// fewer than LV_NODE_MIN_ACTIVE lanes are still descending (the others wait with a parked leaf and a second one in
// hand; testing those now keeps both loops above ~50 % lane utilisation).
__device__ __forceinline__ bool lv_leave_node_loop(unsigned pending) {
    const unsigned long long descending = __ballot(1);
    return !__any(pending == LV_INVALID) || __popcll(descending) < LV_NODE_MIN_ACTIVE;
}

// One node step on the compressed 4-wide LBVH: fetch the 64-byte node (4 x dwordx4), decode + slab-test the four child
// boxes, continue with the nearest hit child and push the others.  Children may be leaves; the caller looks at the
// leaf bit of what comes back / pops.
__device__ __forceinline__ void lv_cswap(float& ka, unsigned& ca, float& kb, unsigned& cb) {
    const bool sw = kb < ka;
    const float tk = sw ? kb : ka, uk = sw ? ka : kb;
    const unsigned tc = sw ? cb : ca, uc = sw ? ca : cb;
    ka = tk; kb = uk; ca = tc; cb = uc;
}

// slab test on decoded planes: t = plane * (1/d) - o/d with plane = origin + q * scale, folded into
// t = q * (scale/d) + (origin/d - o/d): one fma per plane after the byte -> float conversion.  The words handed in
// are already ordered by the ray's direction signs (near planes / far planes), so no per-plane min/max is needed.
#if LV_NODE_MIX
// The same test with the plane bytes read as binary16: v_perm_b32 builds {0x6400 | q_k, 0x6400 | q_k+1} = the halves 1024 + q of two
// children at once (binary16 has ulp 1 on [1024, 2048)), and t = (1024 + q) * A + (B - 1024 A) is ONE v_fma_mix_f32 per plane (the
// binary16 operand is widened inside the instruction, the arithmetic is float32): 1.5 instructions per plane instead of the
// v_cvt_f32_ubyte + v_fma_f32 pair.  B2 = B - 1024 A is computed once per node step (lv_node_step).
typedef _Float16 lv_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lv_h2 lv_plane_pair(uint32_t word, int pair) {
    const uint32_t w = __builtin_amdgcn_perm(word, 0x64006400u, pair == 0 ? 0x03050104u : 0x03070106u);
    return __builtin_bit_cast(lv_h2, w);
}
__device__ __forceinline__ bool lv_slab_h(_Float16 nx, _Float16 ny, _Float16 nz, _Float16 fx, _Float16 fy, _Float16 fz, f3 A, f3 B2,
                                          float tMin, float tMax, float& tNear) {
    const float tx0 = __builtin_fmaf(float(nx), A.x, B2.x);
    const float ty0 = __builtin_fmaf(float(ny), A.y, B2.y);
    const float tz0 = __builtin_fmaf(float(nz), A.z, B2.z);
    const float tx1 = __builtin_fmaf(float(fx), A.x, B2.x);
    const float ty1 = __builtin_fmaf(float(fy), A.y, B2.y);
    const float tz1 = __builtin_fmaf(float(fz), A.z, B2.z);
    const float tn = fmaxf(fmaxf(tx0, ty0), fmaxf(tz0, tMin));
    const float tf = fminf(fminf(tx1, ty1), fminf(tz1, tMax));
    tNear = tn;
    return tn <= __builtin_fmaf(tf, 1.00001f, 4e-7f);
}
#endif
__device__ __forceinline__ bool lv_slab_q(uint32_t nearX, uint32_t nearY, uint32_t nearZ, uint32_t farX, uint32_t farY,
                                          uint32_t farZ, int k, f3 A, f3 B, float tMin, float tMax, float& tNear) {
    const float tx0 = __builtin_fmaf(float((nearX >> (8 * k)) & 0xFFu), A.x, B.x);
    const float ty0 = __builtin_fmaf(float((nearY >> (8 * k)) & 0xFFu), A.y, B.y);
    const float tz0 = __builtin_fmaf(float((nearZ >> (8 * k)) & 0xFFu), A.z, B.z);
    const float tx1 = __builtin_fmaf(float((farX >> (8 * k)) & 0xFFu), A.x, B.x);
    const float ty1 = __builtin_fmaf(float((farY >> (8 * k)) & 0xFFu), A.y, B.y);
    const float tz1 = __builtin_fmaf(float((farZ >> (8 * k)) & 0xFFu), A.z, B.z);
    const float tn = fmaxf(fmaxf(tx0, ty0), fmaxf(tz0, tMin));
    const float tf = fminf(fminf(tx1, ty1), fminf(tz1, tMax));
    tNear = tn;
    return tn <= __builtin_fmaf(tf, 1.00001f, 4e-7f);
}

// ORDERED: 0 = children pushed as they come (all-hits), 1 = nearest hit child first, the rest as they come (closest hit:
// measured as fast as a full sort, fewer instructions), 2 = all hit children in front-to-back order (MLAT: how early the
// transmittance rule closes the ray interval depends on meeting the near layers first).
template <bool STATS, int ORDERED = 1, class STACK>
__device__ __forceinline__ unsigned lv_node_step(const LvSceneDev& S, unsigned node, f3 oi, f3 inv, float tMin, float tMax,
                                                 STACK& st, LvCounters& cnt) {
    const float4* p = S.nodes + 4 * size_t(node);
    const float4 q0 = p[0], q1 = p[1], q2 = p[2], cf = p[3];
    if (STATS) cnt.nodes++;
#ifdef LV_NODE_STEP_PROBE // sensitivity probes live in tools/variants_inc/probes.h (tools/variants.py force-includes it); never in the product
    LV_NODE_STEP_PROBE(p, q0, inv, tMax, S);
#endif
    unsigned c0 = __float_as_uint(cf.x), c1 = __float_as_uint(cf.y), c2 = __float_as_uint(cf.z), c3 = __float_as_uint(cf.w);
    const f3 A = mk3(q0.w * inv.x, q1.x * inv.y, q1.y * inv.z);
    const f3 B = mk3(__builtin_fmaf(q0.x, inv.x, -oi.x), __builtin_fmaf(q0.y, inv.y, -oi.y), __builtin_fmaf(q0.z, inv.z, -oi.z));
    // a negative direction component enters through the max plane (loop-invariant per ray)
    const bool sx = inv.x < 0.0f, sy = inv.y < 0.0f, sz = inv.z < 0.0f;
    const uint32_t qnx = __float_as_uint(q1.z), qny = __float_as_uint(q1.w), qnz = __float_as_uint(q2.x);
    const uint32_t qxx = __float_as_uint(q2.y), qxy = __float_as_uint(q2.z), qxz = __float_as_uint(q2.w);
    const uint32_t nearX = sx ? qxx : qnx, farX = sx ? qnx : qxx;
    const uint32_t nearY = sy ? qxy : qny, farY = sy ? qny : qxy;
    const uint32_t nearZ = sz ? qxz : qnz, farZ = sz ? qnz : qxz;
    float k0, k1, k2, k3;
    // empty slots carry an inverted box (lv_write_wide_node) and fail the slab test by themselves; should one slip through,
    // its reference is LV_INVALID and it is neither descended nor pushed
#if LV_NODE_MIX
    const f3 B2 = mk3(__builtin_fmaf(-1024.0f, A.x, B.x), __builtin_fmaf(-1024.0f, A.y, B.y), __builtin_fmaf(-1024.0f, A.z, B.z));
    const lv_h2 nx01 = lv_plane_pair(nearX, 0), nx23 = lv_plane_pair(nearX, 1), ny01 = lv_plane_pair(nearY, 0), ny23 = lv_plane_pair(nearY, 1);
    const lv_h2 nz01 = lv_plane_pair(nearZ, 0), nz23 = lv_plane_pair(nearZ, 1), fx01 = lv_plane_pair(farX, 0), fx23 = lv_plane_pair(farX, 1);
    const lv_h2 fy01 = lv_plane_pair(farY, 0), fy23 = lv_plane_pair(farY, 1), fz01 = lv_plane_pair(farZ, 0), fz23 = lv_plane_pair(farZ, 1);
    const bool h0 = lv_slab_h(nx01.x, ny01.x, nz01.x, fx01.x, fy01.x, fz01.x, A, B2, tMin, tMax, k0);
    const bool h1 = lv_slab_h(nx01.y, ny01.y, nz01.y, fx01.y, fy01.y, fz01.y, A, B2, tMin, tMax, k1);
    const bool h2 = lv_slab_h(nx23.x, ny23.x, nz23.x, fx23.x, fy23.x, fz23.x, A, B2, tMin, tMax, k2);
    const bool h3 = lv_slab_h(nx23.y, ny23.y, nz23.y, fx23.y, fy23.y, fz23.y, A, B2, tMin, tMax, k3);
#else
    const bool h0 = lv_slab_q(nearX, nearY, nearZ, farX, farY, farZ, 0, A, B, tMin, tMax, k0);
    const bool h1 = lv_slab_q(nearX, nearY, nearZ, farX, farY, farZ, 1, A, B, tMin, tMax, k1);
    const bool h2 = lv_slab_q(nearX, nearY, nearZ, farX, farY, farZ, 2, A, B, tMin, tMax, k2);
    const bool h3 = lv_slab_q(nearX, nearY, nearZ, farX, farY, farZ, 3, A, B, tMin, tMax, k3);
#endif
    const float INF = __builtin_inff();
    if (ORDERED == 3) {
        // nearest hit child WITHOUT a sorting network: the minimum of the masked keys (min3 + min), the first slot that holds it
        // (exact compare: the minimum IS one of the four), the other hit children pushed in stored order.  The hit / selected flags
        // are lane masks (scalar registers, combined on the scalar unit), so the children themselves are never moved or masked.
        k0 = h0 ? k0 : INF; k1 = h1 ? k1 : INF; k2 = h2 ? k2 : INF; k3 = h3 ? k3 : INF;
        const float km = fminf(fminf(k0, k1), fminf(k2, k3));
        const bool s0 = h0 && k0 == km;
        const bool s1 = h1 && !s0 && k1 == km;
        const bool s2 = h2 && !s0 && !s1 && k2 == km;
        const bool s3 = h3 && !s0 && !s1 && !s2;
        unsigned sel = LV_INVALID;
        sel = s3 ? c3 : sel; sel = s2 ? c2 : sel; sel = s1 ? c1 : sel; sel = s0 ? c0 : sel;
        const bool p0 = h0 && !s0, p1 = h1 && !s1, p2 = h2 && !s2, p3 = h3 && !s3;
        if (st.sp + 4 <= STACK::kLds) { // four unconditional writes, at most three of them kept
            st.lds[st.sp * STACK::kStride] = c3; st.sp += p3 ? 1 : 0;
            st.lds[st.sp * STACK::kStride] = c2; st.sp += p2 ? 1 : 0;
            st.lds[st.sp * STACK::kStride] = c1; st.sp += p1 ? 1 : 0;
            st.lds[st.sp * STACK::kStride] = c0; st.sp += p0 ? 1 : 0;
        } else {
            if (p3) st.push(c3);
            if (p2) st.push(c2);
            if (p1) st.push(c1);
            if (p0) st.push(c0);
        }
        if (sel != LV_INVALID) return sel;
        return lv_pop_or_done(st);
    }
    // a missed child keeps its reference and gets the key +inf: the key travels with the reference through the network and
    // decides below whether the reference is pushed / descended (empty slots never hit: inverted boxes, lv_write_wide_node)
    k0 = h0 ? k0 : INF; k1 = h1 ? k1 : INF; k2 = h2 ? k2 : INF; k3 = h3 ? k3 : INF;
#ifdef LV_NODE_STEP_MASK_REFS
    c0 = h0 ? c0 : LV_INVALID; c1 = h1 ? c1 : LV_INVALID; c2 = h2 ? c2 : LV_INVALID; c3 = h3 ? c3 : LV_INVALID;
#endif
    if (ORDERED == 2 || (ORDERED == 1 && LV_SORT_CHILDREN)) { // 5-comparator sorting network, misses (key = +inf) sink to the end
        lv_cswap(k0, c0, k1, c1);
        lv_cswap(k2, c2, k3, c3);
        lv_cswap(k0, c0, k2, c2);
        lv_cswap(k1, c1, k3, c3);
        lv_cswap(k1, c1, k2, c2);
    } else if (ORDERED == 1) { // only the nearest hit child matters for "descend first"; the others are pushed as they come
        lv_cswap(k0, c0, k1, c1);
        lv_cswap(k2, c2, k3, c3);
        lv_cswap(k0, c0, k2, c2);
    }
    // Pushes: write unconditionally and advance the stack pointer only for real references (no branches) as long as
    // the three slots are inside the LDS part; the rare deep case takes the checked path.
    if (st.sp + 3 <= STACK::kLds) {
        st.lds[st.sp * STACK::kStride] = c3; st.sp += (k3 < INF) ? 1 : 0;
        st.lds[st.sp * STACK::kStride] = c2; st.sp += (k2 < INF) ? 1 : 0;
        st.lds[st.sp * STACK::kStride] = c1; st.sp += (k1 < INF) ? 1 : 0;
    } else {
        if (k3 < INF) st.push(c3);
        if (k2 < INF) st.push(c2);
        if (k1 < INF) st.push(c1);
    }
    if (k0 < INF) return c0;
    return lv_pop_or_done(st);
}

// ---------------------------------------------------------------- leaf tests of the cooperative routines
// One (ray, leaf) test.  Returns true with t and the low 32 bits of the merge key: capsules -> (original segment << 2)
// | kind, triangles -> original triangle index; in both cases "smaller key = closer, ties to the lowest index".
// ---------------------------------------------------------------- elliptic tubes (EllipticTubeRayTracing.glsl)
// The ray tracer's "Elliptic Tubes" mode for band data: every segment is a tubelet with an elliptic cross-section (semi-axes
// radius0 = bandWidth / 2 * minBandThickness along the line normal, radius1 = bandWidth / 2 along the binormal) that twists from
// the normal at p0 to the normal at p1, found by sphere tracing in the tubelet's coordinate system (Reina et al. 2006).
struct LvM3 { f3 c0, c1, c2; }; // columns, GLSL layout
__device__ __forceinline__ f3 lv_mul_m3(const LvM3& m, f3 v) { return (m.c0 * v.x + m.c1 * v.y) + m.c2 * v.z; }
// matrixAxisRotationCos, EllipticTubeRayTracing.glsl:69-91
__device__ __forceinline__ LvM3 lv_matrix_axis_rotation_cos(f3 axis, float cosAngle) {
    const float c = cosAngle;
    const float s = sqrtf(1.0f - cosAngle * cosAngle);
    axis = norm3(axis);
    const f3 temp = (1.0f - c) * axis;
    LvM3 R;
    R.c0 = mk3(c + temp.x * axis.x, temp.x * axis.y + s * axis.z, temp.x * axis.z - s * axis.y);
    R.c1 = mk3(temp.y * axis.x - s * axis.z, c + temp.y * axis.y, temp.y * axis.z + s * axis.x);
    R.c2 = mk3(temp.z * axis.x + s * axis.y, temp.z * axis.y - s * axis.x, c + temp.z * axis.z);
    return R;
}
// computeRadius, :3-7
__device__ __forceinline__ float lv_ell_compute_radius(float r1, float r2, float phi, float rho) {
    float sn, cs;
    lv_sincos_rad(phi + rho, sn, cs);
    return r1 * r2 / sqrtf(r1 * r1 * sn * sn + r2 * r2 * cs * cs);
}
// computeNormal, :22-41
__device__ __forceinline__ f3 lv_ell_compute_normal(float r1, float r2, float phi, float rho) {
    float sinphi, cosphi, sinphirho, cosphirho;
    lv_sincos_rad(phi + rho, sinphi, cosphi);
    lv_sincos_rad(phi, sinphirho, cosphirho);
    const float r1sq = r1 * r1, r2sq = r2 * r2, r1r2 = r1 * r2;
    const float rDenomSq = r1sq * sinphi * sinphi + r2sq * cosphi * cosphi;
    const float rDenom = sqrtf(rDenomSq);
    const float r = r1r2 / rDenom;
    const float ddenomDphi = (r1sq - r2sq) * sinphi * cosphi / rDenom;
    const float drDphi = -r1r2 * ddenomDphi / rDenomSq;
    const f3 dxDphi = mk3(0.0f, drDphi * cosphirho - r * sinphirho, drDphi * sinphirho + r * cosphirho);
    return cross3(dxDphi, mk3(1.0f, 0.0f, 0.0f));
}
// rayBoxPlaneIntersection, :126-163
__device__ __forceinline__ bool lv_ell_ray_box_plane(float o, float d, float lower, float upper, float& tNear, float& tFar) {
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
// the tubelet frame shared by the intersection and the closest-hit shader (:201-222 = :320-341)
struct LvTubelet { f3 p0, p1, xt, yt, zt; float l, rhoR; };
__device__ __forceinline__ LvTubelet lv_make_tubelet(const lv_line_point& lp0, const lv_line_point& lp1) {
    LvTubelet T;
    T.p0 = mk3(lp0.linePosition[0], lp0.linePosition[1], lp0.linePosition[2]);
    T.p1 = mk3(lp1.linePosition[0], lp1.linePosition[1], lp1.linePosition[2]);
    const f3 n0 = mk3(lp0.lineNormal[0], lp0.lineNormal[1], lp0.lineNormal[2]);
    const f3 n1 = mk3(lp1.lineNormal[0], lp1.lineNormal[1], lp1.lineNormal[2]);
    const f3 t0 = mk3(lp0.lineTangent[0], lp0.lineTangent[1], lp0.lineTangent[2]);
    T.l = len3(T.p1 - T.p0);
    T.xt = norm3(T.p1 - T.p0);
    const f3 rotAxis = cross3(n0, T.xt);
    const float rotCosAngle = dot3(n0, T.xt);
    LvM3 R;
    R.c0 = mk3(1.0f, 0.0f, 0.0f); R.c1 = mk3(0.0f, 1.0f, 0.0f); R.c2 = mk3(0.0f, 0.0f, 1.0f);
    if (fabsf(rotCosAngle) > 0.999f) R = lv_matrix_axis_rotation_cos(rotAxis, rotCosAngle);
    T.yt = lv_mul_m3(R, n0);
    T.zt = lv_mul_m3(R, cross3(t0, n0));
    T.rhoR = -lv_atan2_det(dot3(cross3(n0, n1), t0), dot3(n0, n1));
    return T;
}
__device__ __forceinline__ f3 lv_to_tubelet(const LvTubelet& T, f3 w) { return mk3(dot3(T.xt, w), dot3(T.yt, w), dot3(T.zt, w)); }
__device__ __forceinline__ f3 lv_from_tubelet(const LvTubelet& T, f3 p) { return (T.xt * p.x + T.yt * p.y) + T.zt * p.z; }

// IntersectionEllipticTube main(), :186-270.  The reference reports the hit whenever the driver invokes the shader, i.e. whenever
// the ray meets the segment's box; hitT itself may leave the box interval (a start point inside the surface steps backwards,
// tilted cutting planes let the surface reach past the box), and the shader's own box test is looser than a slab test (it skips
// axes with |d_i| < 1e-3).  To keep the result independent of the BVH a hit is accepted only if the ray meets the box in the
// slab-test sense and hitT lies within bandWidth / |d| of that interval (own-box rule, as for the literal capsule roots); the
// traversal's culling interval is widened by the same amount (lv_trace_closest).
__device__ __forceinline__ bool lv_intersect_elliptic_tube(const LvSceneDev& S, f3 o, f3 d, const lv_line_point& lp0,
                                                           const lv_line_point& lp1, float& hitTOut) {
    const float radius0 = S.ellBandWidth * 0.5f * S.ellMinBandThickness;
    const float radius1 = S.ellBandWidth * 0.5f;
    const float lwo = S.ellBandWidth * 0.5f;
    float tNear = -1e7f, tFar = 1e7f;
    // the segment's box of TubeAabbRenderData, LineDataFlow.cpp:2223-2234
    if (!lv_ell_ray_box_plane(o.x, d.x, fminf(lp0.linePosition[0], lp1.linePosition[0]) - lwo,
                              fmaxf(lp0.linePosition[0], lp1.linePosition[0]) + lwo, tNear, tFar)) return false;
    if (!lv_ell_ray_box_plane(o.y, d.y, fminf(lp0.linePosition[1], lp1.linePosition[1]) - lwo,
                              fmaxf(lp0.linePosition[1], lp1.linePosition[1]) + lwo, tNear, tFar)) return false;
    if (!lv_ell_ray_box_plane(o.z, d.z, fminf(lp0.linePosition[2], lp1.linePosition[2]) - lwo,
                              fmaxf(lp0.linePosition[2], lp1.linePosition[2]) + lwo, tNear, tFar)) return false;
    const f3 startPoint = o + tNear * d;
    const LvTubelet T = lv_make_tubelet(lp0, lp1);
    const f3 t0 = mk3(lp0.lineTangent[0], lp0.lineTangent[1], lp0.lineTangent[2]);
    const f3 t1 = mk3(lp1.lineTangent[0], lp1.lineTangent[1], lp1.lineTangent[2]);
    const f3 El = t0; const float Elw = -dot3(El, T.p0);             // left and right cutting planes
    const f3 Er = mk3(-t1.x, -t1.y, -t1.z); const float Erw = -dot3(Er, T.p1);
    f3 p = lv_to_tubelet(T, startPoint - T.p0);
    const f3 dd = lv_to_tubelet(T, d);
    float hitT = tNear, dTmp = 0.0f;
    for (int i = 0; i < 80; i++) { // MAX_NUM_SPHERE_TRACING_ITERATIONS
        const float t = clampf(p.x / T.l, 0.0f, 1.0f);
        const float rhoX = t * T.rhoR;
        const float phi = lv_atan2_det(p.z, p.y);
        const float r = lv_ell_compute_radius(radius0, radius1, phi, rhoX);
        const float rad = sqrtf(p.y * p.y + p.z * p.z);
        dTmp = rad - r;
        dTmp *= 0.25f;
        // Result-preserving early out (not in the reference, which marches all 80 steps of a miss): the distance from the tubelet
        // axis is a convex function of the ray parameter, so once it grows (p_yz . d_yz >= 0) it keeps growing; r never exceeds
        // the larger semi-axis; hence if rad - radius1 exceeds 4e-4 now, every later step has dTmp >= 1e-4: the loop can neither
        // terminate (< 1e-5) nor pass the final dTmp < 1e-4 test.  8e-4 leaves 4e-4 for float32 rounding of rad and r.
        if (p.y * dd.y + p.z * dd.z >= 0.0f && rad - radius1 > 8e-4f) return false;
        p = p + dd * dTmp;
        hitT += dTmp;
        if (dTmp < 1e-5f) break; // EPSILON_SPHERE_TRACING_TERMINATION
    }
    const f3 pointWorld = lv_from_tubelet(T, p) + T.p0;
    const f3 cam = mk3(S.ellCamPos[0], S.ellCamPos[1], S.ellCamPos[2]);
    const float eps1 = fabsf(dot3(El, norm3(cam - T.p0))) * 5e-5f;
    const float eps2 = fabsf(dot3(Er, norm3(cam - T.p1))) * 5e-5f;
    const bool isNotCulledLeft = dot3(El, pointWorld) + Elw > -eps1;
    const bool isNotCulledRight = dot3(Er, pointWorld) + Erw > -eps2;
    if (!(dTmp < 1e-4f && hitT > 0.0f && isNotCulledLeft && isNotCulledRight)) return false;
    // own-box rule: the driver only invokes the shader for rays that meet the box (the shader's own test above treats
    // directions with |d_i| < 1e-3 as parallel and is looser than that), and hitT must lie within bandWidth / |d| of the interval
    {
        const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        const f3 p0 = T.p0, p1 = T.p1;
        const float tx0 = ((fminf(p0.x, p1.x) - lwo) - o.x) * inv.x, tx1 = ((fmaxf(p0.x, p1.x) + lwo) - o.x) * inv.x;
        const float ty0 = ((fminf(p0.y, p1.y) - lwo) - o.y) * inv.y, ty1 = ((fmaxf(p0.y, p1.y) + lwo) - o.y) * inv.y;
        const float tz0 = ((fminf(p0.z, p1.z) - lwo) - o.z) * inv.z, tz1 = ((fmaxf(p0.z, p1.z) + lwo) - o.z) * inv.z;
        const float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fminf(tz0, tz1));
        const float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
        const float slack = S.ellBandWidth / len3(d);
        if (!(tn <= tf && hitT >= tn - slack && hitT <= tf + slack)) return false;
    }
    hitTOut = hitT;
    return true;
}

// one triangle of a leaf of the triangle LBVH (slot < S.triLeafSize); low = original triangle index.  Pair records (S.triPairs, the
// layout is k_tri_leaves<true>'s, lv_bvh.hip): slot 0 = (q0, q1, q2), slot 1 = the three vertices its code selects, fetched by address
__device__ __forceinline__ bool lv_tri_record_test(const LvSceneDev& S, unsigned leaf, unsigned slot, f3 o, f3 d, f3 inv, float& t,
                                                   unsigned& low) {
    float4 a, b, c;
    if (S.triPairs) {
        const float4* rec = S.tris + 4 * size_t(leaf);
        const float4 q0 = rec[0], q1 = rec[1];
        low = __float_as_uint(q0.w) + slot;
        if (slot == 0u) { a = q0; b = q1; c = rec[2]; }
        else {
            const unsigned code = __float_as_uint(q1.w);
            a = rec[code & 3u]; b = rec[(code >> 2) & 3u]; c = rec[(code >> 4) & 3u];
        }
    } else {
        const float4* rec = S.tris + 3 * (size_t(leaf) * S.triLeafSize + slot);
        a = rec[0]; b = rec[1]; c = rec[2];
        low = __float_as_uint(a.w);
    }
    float u, v;
    return lv_ray_triangle(o, d, inv, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), S.triPad, t, u, v);
}

// LIT: -1 = intersection form from S.literalIntersection at run time (tile kernels), 0 / 1 = fixed at compile time (k_ao_rays:
// a run-time branch around both capsule tests costs the register that pushes the kernel over its 96-VGPR budget into scratch)
template <int PRIM, int LIT = -1>
__device__ __forceinline__ bool lv_leaf_test(const LvSceneDev& S, unsigned leaf, f3 o, f3 d, float radius, bool capped,
                                             float& t, unsigned& low, float tLo = -3.0e38f, float tHi = 3.0e38f) {
    if (PRIM == LV_PRIM_TRIANGLE) {
        // closest of the leaf's triangles INSIDE the caller's ray interval [tLo, tHi] (a nearer hit outside it must not hide a
        // valid one of the same leaf); ties: lowest original index -- the merge key of the callers
        const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        bool any = false;
        t = 0.0f; low = 0u;
        if (S.triPairs) {
            // pair record: four 16-B loads for both triangles; the second triangle's operands are (q0, q2, q3) for a tube face (all
            // but the cap leaves) and are fetched again by address only by the waves that hold another code
            const float4* rec = S.tris + 4 * size_t(leaf);
            const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
            const unsigned idx0 = __float_as_uint(q0.w), code = __float_as_uint(q1.w);
            float tj, u, v;
            if (lv_ray_triangle(o, d, inv, mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), mk3(q2.x, q2.y, q2.z), S.triPad, tj, u, v) &&
                tj >= tLo && tj <= tHi) { t = tj; low = idx0; any = true; }
            float4 a = q0, b = q2, c = q3;
            if (code != LV_TRI_PAIR_CODE_BODY) { a = rec[code & 3u]; b = rec[(code >> 2) & 3u]; c = rec[(code >> 4) & 3u]; }
            if (lv_ray_triangle(o, d, inv, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), S.triPad, tj, u, v) &&
                tj >= tLo && tj <= tHi) {
                if (!any || tj < t) { t = tj; low = idx0 + 1u; } // ties stay with the lower index
                any = true;
            }
            return any;
        }
        for (uint32_t j = 0; j < S.triLeafSize; j++) {
            float tj;
            unsigned idx;
            if (lv_tri_record_test(S, leaf, j, o, d, inv, tj, idx) && tj >= tLo && tj <= tHi) {
                if (!any || tj < t || (tj == t && idx < low)) { t = tj; low = idx; }
                any = true;
            }
        }
        return any;
    } else if (PRIM == LV_PRIM_ELLIPTIC) {
        const uint32_t seg = S.leafSeg[leaf];
        low = seg << 2;
        return lv_intersect_elliptic_tube(S, o, d, S.points[S.segIdx[2 * size_t(seg)]], S.points[S.segIdx[2 * size_t(seg) + 1]], t);
    } else {
        const float4 a = S.segs[2 * size_t(leaf)], b = S.segs[2 * size_t(leaf) + 1];
        int kind;
        bool hit;
        if (LIT == 1 || (LIT == -1 && S.literalIntersection)) {
            hit = lv_intersect_capsule_literal(o, d, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), radius, capped, t, kind);
        } else {
#if LV_PRECOMP_AXIS
            const float4 ax = S.segAxis[leaf];
            hit = lv_intersect_capsule_td(o, d, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(ax.x, ax.y, ax.z), radius, capped, t, kind);
#else
            hit = lv_intersect_capsule(o, d, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), radius, capped, t, kind);
#endif
        }
        if (hit) low = (S.leafSeg[leaf] << 2) | unsigned(kind);
        return hit;
    }
}

// ---------------------------------------------------------------- wave-cooperative closest hit
// Per-wave LDS scratch of the cooperative routines: the ray of every lane, its best-hit key and a FIFO of
// (owner lane, leaf) pairs waiting for a capsule test.
#define LV_QCAP 256u                      // FIFO entries per wave (< 64 waiting + up to 3 x 64 new per node step)
struct LvCoopMem {
    float4* ray;                   // [2 * 64]  {o.xyz, -}{d.xyz, -}
    unsigned long long* key;       // [64]      (t bits << 32) | (original segment << 2) | kind
    unsigned* queue;               // [LV_QCAP] (owner lane << 26) | leaf
    uint2* xchg;                   // [64]      subtree hand-over slots: {node reference, owner lane}
};
#define LV_COOP_SHARED(NWAVES)                                   \
    __shared__ float4 s_coopRay[2 * LV_WAVE * (NWAVES)];         \
    __shared__ unsigned long long s_coopKey[LV_WAVE * (NWAVES)]; \
    __shared__ unsigned s_coopQueue[(NWAVES)][LV_QCAP];          \
    __shared__ uint2 s_coopXchg[LV_WAVE * (NWAVES)]
#define LV_COOP_MEM(cm)                                          \
    LvCoopMem cm;                                                \
    cm.ray = &s_coopRay[2 * LV_WAVE * (threadIdx.x >> 6)];       \
    cm.key = &s_coopKey[LV_WAVE * (threadIdx.x >> 6)];           \
    cm.queue = s_coopQueue[threadIdx.x >> 6];                    \
    cm.xchg = &s_coopXchg[LV_WAVE * (threadIdx.x >> 6)]

// Per-wave queue of hits waiting to be shaded (all-hits routine): < 64 waiting + <= 64 new per test batch
#define LV_HITQ_CAP 128u
struct LvHitQueue {
    unsigned* ref;   // (owner lane << 26) | leaf
    float* t;
    unsigned* kind;
    const float* prismRing; // LV_PRIM_PRISM: the workgroup's LDS copy of the ring table (cos [k], sin [LV_PRISM_MAX_SUBDIV + k])
    unsigned* prismQueue;   // LV_PRIM_PRISM: [LV_HITQ_CAP] candidates that passed the pre-test and wait for the coverage test
};
#define LV_HITQ_SHARED(NWAVES)                                \
    __shared__ unsigned s_hitRef[(NWAVES)][LV_HITQ_CAP];      \
    __shared__ float s_hitT[(NWAVES)][LV_HITQ_CAP];           \
    __shared__ unsigned s_hitKind[(NWAVES)][LV_HITQ_CAP]
#define LV_HITQ_MEM(hq)                                       \
    LvHitQueue hq;                                            \
    hq.ref = s_hitRef[threadIdx.x >> 6];                      \
    hq.t = s_hitT[threadIdx.x >> 6];                          \
    hq.kind = s_hitKind[threadIdx.x >> 6];                    \
    hq.prismRing = nullptr;                                   \
    hq.prismQueue = nullptr

// Closest hit with reportIntersectionEXT semantics (accepted iff tMin <= t <= tMax; ties -> lowest original segment
// index), computed by the whole wave together: EVERY lane of the wave must call this function in convergent control
// flow, lanes without a ray pass active = false and only lend their ALUs.
//   descend  lanes with an inner node do node steps on the 4-wide LBVH (LDS-staged stack); a lane never tests the
//            leaves it meets, it appends (lane, leaf) to the wave's FIFO (ballot + prefix popcount, no atomics);
//   test     when 64 pairs wait (or nobody can descend) each lane takes one pair, reads the owner's ray from LDS,
//            runs the capsule test and merges with one 64-bit LDS atomicMin on the owner's key.
// So the expensive test (8 IEEE divisions + 4 square roots, ~350 instructions) always runs at full width, and a single
// long ray that meets hundreds of leaves -- the tail that bounded the one-ray-per-thread kernels -- has them tested 64
// at a time by the lanes that already finished.  ANY_HIT: gl_RayFlagsTerminateOnFirstHitEXT.
template <bool STATS, bool ANY_HIT, int PRIM = LV_PRIM_CAPSULE>
__device__ __forceinline__ LvHit lv_trace_closest(const LvSceneDev& S, float radius, bool capped, bool active, f3 o, f3 d,
                                                  float tMin, float tMax, const LvStackMem& sm, const LvCoopMem& cm,
                                                  LvCounters& cnt) {
    const unsigned lane = lv_lane();
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long keyInit = ((unsigned long long)__float_as_uint(tMax) << 32) | 0xFFFFFFFFull;
    active = active && S.numSegs != 0;
    LvHit h;
    h.t = tMax; h.leaf = LV_INVALID; h.kind = 0; h.found = false;
    if (STATS && active) cnt.rays++;
    // the ray this lane currently descends for: its own at first, later possibly a subtree handed over by a busy lane
    unsigned owner = lane;
    f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    f3 oi = mk3(o.x * inv.x, o.y * inv.y, o.z * inv.z);
    LvStack st;
    st.init(sm.lds, sm.ovf, sm.ovfStride);
    cm.ray[2 * lane] = make_float4(o.x, o.y, o.z, tMin);
    cm.ray[2 * lane + 1] = make_float4(d.x, d.y, d.z, tMax);
    cm.key[lane] = keyInit;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    unsigned cur = active ? 0u : LV_INVALID;
    unsigned head = 0, tail = 0;
    float best = tMax;
    // literal roots: cull against [tMin - r / |d|, best + r / |d|] (lv_intersect_capsule_literal); 0 otherwise
    // elliptic tubelets: [tMin - bandWidth / |d|, best + bandWidth / |d|] (lv_intersect_elliptic_tube's own-box rule)
    float slack = PRIM == LV_PRIM_ELLIPTIC ? S.ellBandWidth / len3(d)
                  : (PRIM == LV_PRIM_CAPSULE && S.literalIntersection) ? radius / len3(d) : 0.0f;
    while (true) {
        // leaves reached by the last step (or popped) join the FIFO
        const bool isLeaf = cur != LV_INVALID && (cur & LV_LEAF_BIT);
        const unsigned long long mL = __ballot(isLeaf);
        if (mL) {
            if (isLeaf) {
                cm.queue[(tail + unsigned(__popcll(mL & below))) % LV_QCAP] = (owner << 26) | (cur & 0x03FFFFFFu);
                cur = lv_pop_or_done(st);
            }
            tail += unsigned(__popcll(mL));
            if (tail - head < LV_TRACE_TEST_BATCH) continue; // a popped reference may be a leaf again
        }
        const unsigned long long mNode = __ballot(!(cur & LV_LEAF_BIT));
        const int nNode = __popcll(mNode);
        const unsigned q = tail - head;
        if (q >= LV_TRACE_TEST_BATCH || (q > 0 && nNode == 0)) {
            const unsigned n = q < LV_WAVE ? q : LV_WAVE;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (lane < n) {
                const unsigned e = cm.queue[(head + lane) % LV_QCAP];
                const unsigned ow = e >> 26, leaf = e & 0x03FFFFFFu;
                const float4 ro = cm.ray[2 * ow], rd = cm.ray[2 * ow + 1];
                if (STATS) cnt.prims += PRIM == LV_PRIM_TRIANGLE ? S.triLeafSize : 1u; // primitives tested
                float t; unsigned low;
                if (lv_leaf_test<PRIM>(S, leaf, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), radius, capped, t, low, ro.w, rd.w)) {
                    if (t >= ro.w && t <= rd.w)
                        atomicMin(&cm.key[ow], ((unsigned long long)__float_as_uint(t) << 32) | low);
                }
            }
            head += n;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const unsigned long long key = cm.key[owner];
            best = __uint_as_float(unsigned(key >> 32)); // shrinks the slab interval of the following node steps
            if (ANY_HIT && key != keyInit) { cur = LV_INVALID; st.sp = 0; }
            continue;
        }
        if (nNode == 0) break;
        // ---- subtree hand-over: when at least half of the wave has nothing to descend, every idle lane takes the top
        // stack entry of a lane that still has entries stacked and continues it FOR THAT LANE'S RAY (hits merge into the
        // owner's key like any other).  A ray that needs hundreds of node steps -- the tail that bounds the tile kernels,
        // whose time is otherwise the serial fetch chain of one lane -- is spread over the wave this way.  The closest hit
        // is a minimum over all leaves reached under a monotonically shrinking bound, so the result does not depend on
        // who visits which subtree.
        if (nNode <= LV_HANDOVER_MAX_BUSY) {
            const bool idle = cur == LV_INVALID;
            const bool donor = !idle && st.sp > 0;
            const unsigned long long mIdle = __ballot(idle), mDonor = __ballot(donor);
            const unsigned nPairs = min(unsigned(__popcll(mIdle)), unsigned(__popcll(mDonor)));
            if (nPairs) {
                if (donor) {
                    const unsigned r = unsigned(__popcll(mDonor & below));
                    if (r < nPairs) cm.xchg[r] = make_uint2(st.pop(), owner);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (idle) {
                    const unsigned r = unsigned(__popcll(mIdle & below));
                    if (r < nPairs) {
                        const uint2 x = cm.xchg[r];
                        cur = x.x;
                        owner = x.y;
                        const float4 ro = cm.ray[2 * owner], rd = cm.ray[2 * owner + 1];
                        inv = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                        oi = mk3(ro.x * inv.x, ro.y * inv.y, ro.z * inv.z);
                        tMin = ro.w;
                        best = __uint_as_float(unsigned(cm.key[owner] >> 32));
                        slack = PRIM == LV_PRIM_ELLIPTIC ? S.ellBandWidth / len3(mk3(rd.x, rd.y, rd.z))
                                : (PRIM == LV_PRIM_CAPSULE && S.literalIntersection) ? radius / len3(mk3(rd.x, rd.y, rd.z)) : 0.0f;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                continue; // the reference taken over may be a leaf
            }
        }
        int nNow;
        do { // tight descend loop
            if (!(cur & LV_LEAF_BIT)) cur = lv_node_step<STATS>(S, cur, oi, inv, tMin - slack, best + slack, st, cnt);
            const bool lf = cur != LV_INVALID && (cur & LV_LEAF_BIT);
            const unsigned long long m = __ballot(lf);
            if (m) {
                if (lf) {
                    cm.queue[(tail + unsigned(__popcll(m & below))) % LV_QCAP] = (owner << 26) | (cur & 0x03FFFFFFu);
                    cur = lv_pop_or_done(st);
                }
                tail += unsigned(__popcll(m));
            }
            nNow = __popcll(__ballot(!(cur & LV_LEAF_BIT)));
        } while (tail - head < LV_TRACE_TEST_BATCH && nNow > LV_HANDOVER_MAX_BUSY);
    }
    const unsigned long long key = cm.key[lane];
    if (active && key != keyInit) {
        h.found = true;
        h.t = __uint_as_float(unsigned(key >> 32));
        if (PRIM == LV_PRIM_TRIANGLE) {
            h.kind = 0;
            h.leaf = unsigned(key); // triangles: the ORIGINAL triangle index (callers read the index / vertex buffers)
        } else {
            h.kind = int(unsigned(key) & 3u);
            h.leaf = S.segToLeaf[(unsigned(key) >> 2)];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the scratch is reused by the next call
    return h;
}

#include "lv_prism.h"

// All capsule entry hits with t in [tMin, tMax) -- per lane -- (PPLL fragment generation, MLAT), wave-cooperative like lv_trace_closest: every
// lane of the wave calls it; (owner, leaf) pairs are tested 64 at a time by whichever lanes are free, and the lane
// that finds a hit calls f(owner, leaf, t, kind, o, d, w0, w1) with the OWNER's ray and its two payload words
// (cm.ray[..].w), i.e. fragments of one pixel may be produced by any lane.
// DYN (multi-layer alpha tracing): the interval is [tMin, tMax] and its end may SHRINK while the ray is traced -- after
// every shading batch all lanes call g(n) in convergent control flow (the per-pixel commit step, which may lower
// cm.ray[2 * lane + 1].w of its own ray = "the any-hit shader accepted a hit"), and the descending lanes pick the new end
// up for their culling.
// PRIM = LV_PRIM_TRIANGLE: S is the triangle scene view, f receives the ORIGINAL triangle index in `kind`.
template <bool STATS, bool DYN, int PRIM = LV_PRIM_CAPSULE, typename F, typename G>
__device__ __forceinline__ void lv_trace_all(const LvSceneDev& S, float radius, bool capped, bool active, f3 o, f3 d,
                                             float tMin, float tMax, float w0, float w1, const LvStackMem& sm,
                                             const LvCoopMem& cm, const LvHitQueue& hq, LvCounters& cnt, F&& f, G&& g) {
    constexpr unsigned HO_BUSY = DYN ? LV_HANDOVER_MAX_BUSY_DYN : LV_HANDOVER_MAX_BUSY_ALL;
    const unsigned lane = lv_lane();
    const unsigned long long below = (1ull << lane) - 1ull;
    active = active && S.numSegs != 0;
    if (STATS && active) cnt.rays++;
    unsigned owner = lane; // the ray this lane descends for (see the subtree hand-over in lv_trace_closest)
    f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    f3 oi = mk3(o.x * inv.x, o.y * inv.y, o.z * inv.z);
    LvStack st;
    st.init(sm.lds, sm.ovf, sm.ovfStride);
    // every lane has its OWN interval [tMin, tMax) (depth slices): it travels with the ray, the payload words go in the
    // key slot (unused by the all-hits routine)
    cm.ray[2 * lane] = make_float4(o.x, o.y, o.z, tMin);
    cm.ray[2 * lane + 1] = make_float4(d.x, d.y, d.z, tMax);
    cm.key[lane] = ((unsigned long long)__float_as_uint(w1) << 32) | __float_as_uint(w0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    unsigned cur = active ? 0u : LV_INVALID;
    unsigned head = 0, tail = 0;
    unsigned hHead = 0, hTail = 0; // hit queue: hits wait here until 64 of them can be shaded at full width
    auto shadeBatch = [&](unsigned n) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane < n) {
            const unsigned i = (hHead + lane) % LV_HITQ_CAP;
            const unsigned e = hq.ref[i];
            const unsigned ow = e >> 26, leaf = e & 0x03FFFFFFu;
            const float4 ro = cm.ray[2 * ow], rd = cm.ray[2 * ow + 1];
            const unsigned long long pw = cm.key[ow];
            f(ow, leaf, hq.t[i], int(hq.kind[i]), mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), __uint_as_float(unsigned(pw)),
              __uint_as_float(unsigned(pw >> 32)));
        }
        hHead += n;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (DYN) {
            g(n);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            tMax = cm.ray[2 * owner + 1].w;
        }
    };
    // the rasterised prism, stage B: coverage masks of nB (<= 64) pre-tested candidates; the covered triangles are queued as hits one
    // per lane and round (usually one round), kind = the triangle of the segment's prism; stage C (fragment stage) = shadeBatch
    unsigned pHead = 0, pTail = 0;
    auto prismStageB = [&](unsigned nB) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        unsigned ref = 0u, mask = 0u;
        if (lane < nB) {
            ref = hq.prismQueue[(pHead + lane) % LV_HITQ_CAP];
            const unsigned ow = ref >> 26, leaf = ref & 0x03FFFFFFu;
            const float4 ro = cm.ray[2 * ow];
            // the owner's pixel = its second payload word (k_ppll_gather) -> the pixel's coverage direction (lv_prism.h)
            const unsigned pxy = unsigned(cm.key[ow] >> 32);
            const f3 D = lv_prism_cov_dir(S.prism, pxy & 0xFFFFu, pxy >> 16);
            mask = S.prism.n == 6u ? lv_prism_coverage<6>(S, radius, leaf, mk3(ro.x, ro.y, ro.z), D)
                                   : lv_prism_coverage<0>(S, radius, leaf, mk3(ro.x, ro.y, ro.z), D);
        }
        pHead += nB;
        while (__ballot(mask != 0u)) {
            const bool hit = mask != 0u;
            unsigned tri = 0u;
            if (hit) { tri = unsigned(__ffs(int(mask))) - 1u; mask &= mask - 1u; }
            const unsigned long long mH = __ballot(hit);
            if (hit) {
                const unsigned i = (hTail + unsigned(__popcll(mH & below))) % LV_HITQ_CAP;
                hq.ref[i] = ref; hq.t[i] = 0.0f; hq.kind[i] = tri;
            }
            hTail += unsigned(__popcll(mH));
            if (hTail - hHead >= LV_WAVE) shadeBatch(LV_WAVE);
        }
    };
    while (true) {
        const bool isLeaf = cur != LV_INVALID && (cur & LV_LEAF_BIT);
        const unsigned long long mL = __ballot(isLeaf);
        if (mL) {
            if (isLeaf) {
                cm.queue[(tail + unsigned(__popcll(mL & below))) % LV_QCAP] = (owner << 26) | (cur & 0x03FFFFFFu);
                cur = lv_pop_or_done(st);
            }
            tail += unsigned(__popcll(mL));
            if (tail - head < LV_TRACE_TEST_BATCH_ALL) continue;
        }
        const int nNode = __popcll(__ballot(!(cur & LV_LEAF_BIT)));
        const unsigned q = tail - head;
        if (q >= LV_TRACE_TEST_BATCH_ALL || (q > 0 && nNode == 0)) {
            const unsigned n = q < LV_WAVE ? q : LV_WAVE;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // a leaf of the triangle LBVH holds S.triLeafSize triangles: one round of tests + hit queueing per slot (wave-uniform)
            if (PRIM == LV_PRIM_PRISM) {
                // the rasterised prism (lv_prism.h), stage A: capsule pre-test of the candidates; the ones that pass wait in a second
                // queue until 64 of them can take stage B at full width
                bool pass = false;
                unsigned e = 0u;
                if (lane < n) {
                    e = cm.queue[(head + lane) % LV_QCAP];
                    const unsigned ow = e >> 26, leaf = e & 0x03FFFFFFu;
                    const float4 ro = cm.ray[2 * ow], rd = cm.ray[2 * ow + 1];
                    if (STATS) cnt.prims++;
                    pass = lv_prism_pretest(S, radius, leaf, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z));
                }
                const unsigned long long mP = __ballot(pass);
                if (pass) hq.prismQueue[(pTail + unsigned(__popcll(mP & below))) % LV_HITQ_CAP] = e;
                pTail += unsigned(__popcll(mP));
                head += n;
                if (pTail - pHead >= LV_WAVE) prismStageB(LV_WAVE);
                continue;
            }
            const unsigned nSub = PRIM == LV_PRIM_TRIANGLE ? S.triLeafSize : 1u;
            for (unsigned sub = 0; sub < nSub; sub++) {
            bool hit = false;
            unsigned hitRef = 0, hitKind = 0;
            float hitT = 0.0f;
            if (lane < n) {
                const unsigned e = cm.queue[(head + lane) % LV_QCAP];
                const unsigned ow = e >> 26, leaf = e & 0x03FFFFFFu;
                const float4 ro = cm.ray[2 * ow], rd = cm.ray[2 * ow + 1];
                if (STATS) cnt.prims++;
                float t; int kind;
                bool found;
                if (PRIM == LV_PRIM_TRIANGLE) {
                    unsigned low;
                    const f3 dd = mk3(rd.x, rd.y, rd.z);
                    found = lv_tri_record_test(S, leaf, sub, mk3(ro.x, ro.y, ro.z), dd,
                                               mk3(1.0f / dd.x, 1.0f / dd.y, 1.0f / dd.z), t, low);
                    kind = int(low);
                } else if (PRIM == LV_PRIM_ELLIPTIC) {
                    unsigned low;
                    found = lv_leaf_test<LV_PRIM_ELLIPTIC>(S, leaf, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), radius, capped, t, low);
                    kind = 0;
                } else {
                    const float4 a = S.segs[2 * leaf], b = S.segs[2 * leaf + 1];
                    found = S.literalIntersection
                            ? lv_intersect_capsule_literal(mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), mk3(a.x, a.y, a.z),
                                                           mk3(b.x, b.y, b.z), radius, capped, t, kind)
                            : lv_intersect_capsule(mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), mk3(a.x, a.y, a.z),
                                                   mk3(b.x, b.y, b.z), radius, capped, t, kind);
                }
                if (found) {
                    if (t >= ro.w && (DYN ? t <= rd.w : t < rd.w)) { hit = true; hitRef = e; hitT = t; hitKind = unsigned(kind); }
                }
            }
            // hits are not shaded by the lane that found them: they queue up (ballot + prefix popcount) and are shaded 64
            // at a time -- a test batch in front of sparse geometry yields a handful of hits, and shading (~700
            // instructions with three pow) for a handful of lanes per batch was the critical path of the gather
            const unsigned long long mH = __ballot(hit);
            if (mH) {
                if (hit) {
                    const unsigned i = (hTail + unsigned(__popcll(mH & below))) % LV_HITQ_CAP;
                    hq.ref[i] = hitRef; hq.t[i] = hitT; hq.kind[i] = hitKind;
                }
                hTail += unsigned(__popcll(mH));
                if (hTail - hHead >= LV_WAVE) shadeBatch(LV_WAVE);
            }
            }
            head += n;
            continue;
        }
        if (nNode == 0) break;
        if (nNode <= HO_BUSY) { // subtree hand-over, as in lv_trace_closest
            const bool idle = cur == LV_INVALID;
            const bool donor = !idle && st.sp > 0;
            const unsigned long long mIdle = __ballot(idle), mDonor = __ballot(donor);
            const unsigned nPairs = min(unsigned(__popcll(mIdle)), unsigned(__popcll(mDonor)));
            if (nPairs) {
                if (donor) {
                    const unsigned r = unsigned(__popcll(mDonor & below));
                    if (r < nPairs) cm.xchg[r] = make_uint2(st.pop(), owner);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (idle) {
                    const unsigned r = unsigned(__popcll(mIdle & below));
                    if (r < nPairs) {
                        const uint2 x = cm.xchg[r];
                        cur = x.x;
                        owner = x.y;
                        const float4 ro = cm.ray[2 * owner], rd = cm.ray[2 * owner + 1];
                        inv = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                        oi = mk3(ro.x * inv.x, ro.y * inv.y, ro.z * inv.z);
                        tMin = ro.w;
                        tMax = rd.w;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                continue;
            }
        }
        int nNow;
        do {
            if (!(cur & LV_LEAF_BIT)) {
                // literal roots may lie up to r / |d| outside their segment's box interval (lv_intersect_capsule_literal)
                const float sl = PRIM == LV_PRIM_ELLIPTIC ? S.ellBandWidth / len3(mk3(1.0f / inv.x, 1.0f / inv.y, 1.0f / inv.z))
                        : ((PRIM == LV_PRIM_CAPSULE && S.literalIntersection) || PRIM == LV_PRIM_PRISM) // prism: own-box rule, lv_prism.h
                        ? radius / len3(mk3(1.0f / inv.x, 1.0f / inv.y, 1.0f / inv.z)) : 0.0f;
                cur = lv_node_step<STATS, DYN ? 2 : 0>(S, cur, oi, inv, tMin - sl, tMax + sl, st, cnt);
            }
            const bool lf = cur != LV_INVALID && (cur & LV_LEAF_BIT);
            const unsigned long long m = __ballot(lf);
            if (m) {
                if (lf) {
                    cm.queue[(tail + unsigned(__popcll(m & below))) % LV_QCAP] = (owner << 26) | (cur & 0x03FFFFFFu);
                    cur = lv_pop_or_done(st);
                }
                tail += unsigned(__popcll(m));
            }
            nNow = __popcll(__ballot(!(cur & LV_LEAF_BIT)));
        } while (tail - head < LV_TRACE_TEST_BATCH_ALL && nNow > HO_BUSY);
    }
    if (PRIM == LV_PRIM_PRISM && pTail != pHead) prismStageB(pTail - pHead); // the rest (< 64)
    if (hTail != hHead) shadeBatch(hTail - hHead); // the rest (< 64)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// ---------------------------------------------------------------- ray generation, TubeRayTracing.glsl:219-226
__device__ __forceinline__ void lv_primary_ray(const LvUniforms& U, uint32_t x, uint32_t y, float xix, float xiy, f3& o,
                                               f3& d) {
    float ndcx = 2.0f * ((float(x) + xix) / float(U.width)) - 1.0f;
    float ndcy = 2.0f * ((float(y) + xiy) / float(U.height)) - 1.0f;
    f4 target = mulM4(U.invProj, ndcx, ndcy, 1.0f, 1.0f);
    f3 tn = norm3(mk3(target.x, target.y, target.z));
    f4 dir = mulM4(U.invView, tn.x, tn.y, tn.z, 0.0f);
    o = mk3(U.camPos[0], U.camPos[1], U.camPos[2]);
    d = mk3(dir.x, dir.y, dir.z);
}

// ---------------------------------------------------------------- shading
__device__ __forceinline__ f4 lv_transfer_function(const LvSceneDev& S, const LvUniforms& U, float attr) {
    float pos = clampf((attr - U.attrMin) / (U.attrMax - U.attrMin), 0.0f, 1.0f);
    int n = int(U.tfN);
    float u = pos * float(n) - 0.5f;
    float fl = floorf(u);
    float f = u - fl;
    int i0 = int(fl), i1 = i0 + 1;
    i0 = min(max(i0, 0), n - 1);
    i1 = min(max(i1, 0), n - 1);
    float4 c0 = S.tf[i0], c1 = S.tf[i1];
    f4 r;
    r.x = c0.x * (1.0f - f) + c1.x * f;
    r.y = c0.y * (1.0f - f) + c1.y * f;
    r.z = c0.z * (1.0f - f) + c1.z * f;
    r.w = c0.w * (1.0f - f) + c1.w * f;
    return r;
}

// USE_BANDS arguments of computeFragmentColor (RayHitCommon.glsl:82-90,148-190)
// (LV_SHADE_HELICITY: phi and rotation = fragmentRotation of USE_ROTATING_HELICITY_BANDS, :91-93; the rest unused)
// rasterEpsWhite >= 0: the raster tube shader's variant of the outline (the PPLL gather, LinePassGeometryShaderTubes.glsl:785-815,
// 1079-1087: EPSILON_OUTLINE = 0, EPSILON_WHITE = fwidth(ribbonPosition) = this value, cap halo min(rp, |rp2|)); < 0: RayHitCommon's
// stripeAaf >= 0: the raster shader's helicity stripe (LinePassGeometryShaderTubes.glsl:716-721,1046-1052): offset 0.1 instead of w / 2,
// aaf = fwidth(phi + fragmentRotation) over the pixel quad instead of 10 EPSILON_OUTLINE (the rasterised prism supplies it)
// stripeDx / stripeDy: dFdx / dFdy of globalPos = (phi + fragmentRotation) / 2 pi for the textureGrad of the twist-line texture
struct LvBandArgs { bool useBand; float phi; f3 linePosition, lineNormal; float rotation, separatorScale; float rasterEpsWhite; float stripeAaf = -1.0f;
                    float stripeDx = 0.0f, stripeDy = 0.0f; };

// The twist-line texture (USE_HELICITY_BANDS_TEXTURE): sampler2D with REPEAT addressing, sampled at (u, 0.5).  The ray tracer's shaders
// call texture() without derivatives = level 0 (RayHitCommon.glsl:66-72); the raster shader textureGrad(.., (dFdx(globalPos), 0),
// (dFdy(globalPos), 0)) (LinePassGeometryShaderTubes.glsl:724-730).  Build-owned where Vulkan leaves room: texel centres at
// (i + 0.5) / size, linear weights in full float, level of detail lambda = log2(max(|dudx|, |dudy|) * width) clamped to the chain,
// nearest mip = ceil(lambda + 0.5) - 1, mip chain = 2 x 2 box averages in float (lv_set_twist_line_texture); no anisotropic filtering.
__device__ __forceinline__ f4 lv_twist_level(const LvSceneDev& S, const LvUniforms& U, uint32_t level, float u, float v, bool linear) {
    uint32_t w = U.twistW, h = U.twistH;
    size_t base = 0;
    for (uint32_t l = 0; l < level; l++) { base += size_t(w) * h; w = max(w >> 1, 1u); h = max(h >> 1, 1u); }
    const float4* __restrict__ T = S.twistTex + base;
    auto wrap = [](int i, uint32_t n) { int m = i % int(n); return uint32_t(m < 0 ? m + int(n) : m); };
    auto texel = [&](uint32_t i, uint32_t j) { const float4 t = T[size_t(j) * w + i]; f4 r; r.x = t.x; r.y = t.y; r.z = t.z; r.w = t.w; return r; };
    if (!linear) return texel(wrap(int(floorf(u * float(w))), w), wrap(int(floorf(v * float(h))), h));
    const float x = u * float(w) - 0.5f, y = v * float(h) - 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y);
    const float a = x - fx0, b = y - fy0;
    const uint32_t i0 = wrap(int(fx0), w), i1 = wrap(int(fx0) + 1, w), j0 = wrap(int(fy0), h), j1 = wrap(int(fy0) + 1, h);
    const f4 t00 = texel(i0, j0), t10 = texel(i1, j0), t01 = texel(i0, j1), t11 = texel(i1, j1);
    f4 r;
    r.x = mixf(mixf(t00.x, t10.x, a), mixf(t01.x, t11.x, a), b);
    r.y = mixf(mixf(t00.y, t10.y, a), mixf(t01.y, t11.y, a), b);
    r.z = mixf(mixf(t00.z, t10.z, a), mixf(t01.z, t11.z, a), b);
    r.w = mixf(mixf(t00.w, t10.w, a), mixf(t01.w, t11.w, a), b);
    return r;
}
__device__ __forceinline__ f4 lv_twist_sample(const LvSceneDev& S, const LvUniforms& U, float u, float dudx, float dudy, bool useGrad) {
    const uint32_t mode = U.twistFilterMode;          // Nearest, Linear, Nearest Mipmap Nearest, Linear M. Nearest, Nearest M. Linear, Linear M. Linear
    const bool linear = mode == 1u || mode == 3u || mode == 5u;
    const float v = 0.5f;
    if (mode < 2u || !useGrad || U.twistLevels <= 1u) return lv_twist_level(S, U, 0u, u, v, linear);
    const float rho = fmaxf(fabsf(dudx), fabsf(dudy)) * float(U.twistW);
    const float maxLevel = float(U.twistLevels - 1u);
    const float lambda = rho > 1.0f ? fminf(lv_log2_det(rho), maxLevel) : 0.0f;
    if (mode == 2u || mode == 3u) {                   // mipmap mode NEAREST
        const float d = fminf(fmaxf(ceilf(lambda + 0.5f) - 1.0f, 0.0f), maxLevel);
        return lv_twist_level(S, U, uint32_t(d), u, v, linear);
    }
    const float dhi = floorf(lambda), delta = lambda - dhi;
    const uint32_t lhi = uint32_t(dhi), llo = min(lhi + 1u, U.twistLevels - 1u);
    const f4 a = lv_twist_level(S, U, lhi, u, v, linear), b = lv_twist_level(S, U, llo, u, v, linear);
    f4 r;
    r.x = mixf(a.x, b.x, delta); r.y = mixf(a.y, b.y, delta); r.z = mixf(a.z, b.z, delta); r.w = mixf(a.w, b.w, delta);
    return r;
}

// Raster variant of the fragment colour: fwidth(ribbonPosition) over the 2 x 2 pixel quad.  The ribbon coordinate of a fragment is
// a function of the VIEWING RAY (|cross(newV, n)| = distance between the ray and the tube axis over the radius; the USE_BANDS
// coordinate = the position of the ray's trace in the cross-section plane between the two silhouette points), so the quad
// partners' "helper invocations" are the rays through the pixels (x ^ 1, y) and (x, y ^ 1) evaluated against the fragment's segment,
// whether or not they hit it: fwidth = |f(x ^ 1, y) - f(x, y)| + |f(x, y ^ 1) - f(x, y)|  (same operation order as the CPU checker).
// The three rays come from the AFFINE ray generator (directions are only used up to their length): D(x, y) = invView * (invProj *
// (ndc(x, y), 1, 1)).xyz without the normalisation of TubeRayTracing.glsl:225-226, D(x +- 1, y) = D +- dD/dx, D(x, y +- 1) = D +- dD/dy
// with the per-frame steps dD/dx = invView * (invProj * (2 / W, 0, 0, 0)).xyz, dD/dy likewise (wave-uniform: scalar registers).
struct LvRasterQuad { f3 d0, dX, dY; };
__device__ __forceinline__ LvRasterQuad lv_make_raster_quad(const LvUniforms& U, uint32_t x, uint32_t y) {
    const float sxp = 2.0f / float(U.width), syp = 2.0f / float(U.height);   // NDC step of one pixel (wave-uniform)
    const float ndcx = (float(x) + 0.5f) * sxp - 1.0f;
    const float ndcy = (float(y) + 0.5f) * syp - 1.0f;
    const f4 target = mulM4(U.invProj, ndcx, ndcy, 1.0f, 1.0f);
    const f4 dir = mulM4(U.invView, target.x, target.y, target.z, 0.0f);
    const f4 gx = mulM4(U.invProj, sxp, 0.0f, 0.0f, 0.0f);
    const f4 gy = mulM4(U.invProj, 0.0f, syp, 0.0f, 0.0f);
    const f4 Gx = mulM4(U.invView, gx.x, gx.y, gx.z, 0.0f);
    const f4 Gy = mulM4(U.invView, gy.x, gy.y, gy.z, 0.0f);
    const float sx = (x & 1u) ? -1.0f : 1.0f, sy = (y & 1u) ? -1.0f : 1.0f;
    LvRasterQuad q;
    q.d0 = mk3(dir.x, dir.y, dir.z);
    q.dX = mk3(dir.x + sx * Gx.x, dir.y + sx * Gx.y, dir.z + sx * Gx.z);
    q.dY = mk3(dir.x + sy * Gy.x, dir.y + sy * Gy.y, dir.z + sy * Gy.z);
    return q;
}
__device__ __forceinline__ float lv_tube_ribbon_of_ray(f3 cam, f3 d, f3 axisPoint, f3 t, float radius) {
    // t . (wp x dp) = (t x wp) . d  and  |dp|^2 = |d|^2 - (d . t)^2  (t unit, wp and t x wp perpendicular to t)
    const f3 w = cam - axisPoint;
    const f3 wp = w - dot3(w, t) * t;
    const f3 k = cross3(t, wp);
    const float dt = dot3(d, t);
    return clampf(dot3(k, d) / (sqrtf(dot3(d, d) - dt * dt) * radius), -1.0f, 1.0f);
}
// caps: the shader's cap coordinate where the ray meets the tangent plane of the cap at the fragment, sphere normal direction there
__device__ __forceinline__ float lv_cap_ribbon_of_ray(f3 cam, f3 d, f3 hit, f3 hitNormal, f3 centre, f3 t) {
    const float s = dot3(hit - cam, hitNormal) / dot3(d, hitNormal);
    const f3 q = cam + d * s;
    const f3 n = norm3s(q - centre);
    const f3 vv = norm3s(cam - q);
    const f3 helperVec = norm3s(cross3(t, vv));
    const f3 newV = norm3s(cross3(helperVec, t));
    const f3 crossProdVn = cross3(vv, n);
    float ribbonPosition2 = len3(cross3(newV, n));
    if (dot3(t, crossProdVn) < 0.0f) ribbonPosition2 = -ribbonPosition2;
    ribbonPosition2 = clampf(ribbonPosition2, -1.0f, 1.0f);
    return fminf(len3(crossProdVn), fabsf(ribbonPosition2));
}
// USE_BANDS halo coordinate (RayHitCommon.glsl:232-351 = LinePassGeometryShaderTubes.glsl:855-942): the position of the line
// camera -> pH between the two silhouette points of the elliptic cross-section -- tangent-plane coordinates, polar line of the
// camera point with respect to the conic x^2 / thickness^2 + y^2 = 1, its two intersections with the conic from the degenerate
// conic B + alpha M_l.  pH = (thickness cos phi, sin phi, 1) for the fragment itself, any point of a viewing ray's trace otherwise.
__device__ __forceinline__ float lv_bands_ribbon_of_point(f3 cam, f3 linePosition, f3 lineNormal, f3 fragmentTangent, f3 t, f3 pH,
                                                          float lineRadius, float thickness) {
    const f3 lineN = norm3(lineNormal);
    const f3 lineB = cross3(t, lineN);
    const f3 cNorm = cam - linePosition;
    const float dist = dot3(cNorm, fragmentTangent);
    const f3 wv = cNorm - dist * fragmentTangent;
    const f3 cHat = mk3(dot3(lineN, wv), dot3(lineB, wv), dot3(t, wv)); // transpose(mat3(lineN, lineB, t)) * w
    const f3 c = mk3(cHat.x / lineRadius, cHat.y / lineRadius, 1.0f);
    const float a = 1.0f / (thickness * thickness);
    const f3 l = mk3(a * c.x, c.y, -1.0f);
    // M_l = shearSymmetricMatrix(l): columns (0, -l.z, l.y), (l.z, 0, -l.x), (-l.y, l.x, 0); B[column][row]
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
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Cm[i][j] = B[i][j] + alpha * Ml[i][j];
    float pm0x = 0.0f, pm0y = 0.0f, pm1x = 0.0f, pm1y = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (fabsf(Cm[i][i]) > EPSILON) {
            pm0x = Cm[i][0] / Cm[i][2]; pm0y = Cm[i][1] / Cm[i][2]; // column i
            pm1x = Cm[0][i] / Cm[2][i]; pm1y = Cm[1][i] / Cm[2][i]; // row i
        }
    }
    const f3 pLineH = cross3(l, cross3(c, pH));
    const float plx = pLineH.x / pLineH.z, ply = pLineH.y / pLineH.z;
    const float num = sqrtf((plx - pm0x) * (plx - pm0x) + (ply - pm0y) * (ply - pm0y));
    const float den = sqrtf((pm1x - pm0x) * (pm1x - pm0x) + (pm1y - pm0y) * (pm1y - pm0y));
    return num / den * 2.0f - 1.0f;
}
// the USE_BANDS coordinate of a VIEWING RAY: its trace in the cross-section plane through linePosition (normal fragmentTangent)
__device__ __forceinline__ float lv_bands_ribbon_of_ray(f3 cam, f3 d, f3 linePosition, f3 lineNormal, f3 fragmentTangent, f3 t,
                                                        float lineRadius, float thickness) {
    const f3 lineN = norm3(lineNormal);
    const f3 lineB = cross3(t, lineN);
    const float s = dot3(linePosition - cam, fragmentTangent) / dot3(d, fragmentTangent);
    const f3 wq = (cam + d * s) - linePosition;
    return lv_bands_ribbon_of_point(cam, linePosition, lineNormal, fragmentTangent, t,
                                    mk3(dot3(lineN, wq) / lineRadius, dot3(lineB, wq) / lineRadius, 1.0f), lineRadius, thickness);
}
template <int BANDS, int FAST = 0>
__device__ __forceinline__ f4 lv_compute_fragment_color_t(const LvSceneDev& S, const LvUniforms& U, float aoTexel, f3 fragPos,
                                                          f3 fragmentNormal, f3 fragmentTangent, bool isCap,
                                                          float fragmentAttribute, float& payloadHitT, const LvBandArgs& bands);
__device__ __forceinline__ f4 lv_compute_fragment_color(const LvSceneDev& S, const LvUniforms& U, float aoTexel, f3 fragPos,
                                                        f3 fragmentNormal, f3 fragmentTangent, bool isCap,
                                                        float fragmentAttribute, float& payloadHitT) {
    LvBandArgs none;
    none.useBand = false; none.phi = 0.0f; none.linePosition = mk3(0.0f, 0.0f, 0.0f); none.lineNormal = mk3(0.0f, 0.0f, 0.0f);
    none.rotation = 0.0f; none.separatorScale = 1.0f; none.rasterEpsWhite = -1.0f;
    return lv_compute_fragment_color_t<LV_SHADE_PLAIN>(S, U, aoTexel, fragPos, fragmentNormal, fragmentTangent, isCap, fragmentAttribute,
                                              payloadHitT, none);
}

// getAoFactor(interpolatedVertexId, phi) of the static prebaker, AmbientOcclusion.glsl:49-75, without its last two lines
// (pow(gamma) and the strength mapping are the same as for the screen-space texture and applied by the shading code).
__device__ __forceinline__ float lv_prebaked_ao_lookup(const LvSceneDev& S, const LvUniforms& U, float interpolatedVertexId,
                                                       float phi) {
    const uint32_t lastLinePointIdx = uint32_t(interpolatedVertexId);
    const uint32_t nextLinePointIdx = min(lastLinePointIdx + 1u, U.bakeNumLineVertices - 1u);
    const float interpolationFactor = interpolatedVertexId - floorf(interpolatedVertexId);
    const float blendingWeight = mixf(S.bakedBlendingWeights[lastLinePointIdx], S.bakedBlendingWeights[nextLinePointIdx],
                                      interpolationFactor);
    const uint32_t lastVertexIdx = uint32_t(blendingWeight);
    const uint32_t nextVertexIdx = min(lastVertexIdx + 1u, U.bakeNumParametrizationVertices - 1u);
    const float interpolationFactorLine = blendingWeight - floorf(blendingWeight);
    const uint32_t N = U.bakeNumTubeSubdivisions;
    const float circleIdxFlt = clampf(phi / 6.28318530717958647692f * float(N), 0.0f, float(N));
    const uint32_t circleIdxLast = (uint32_t(floorf(circleIdxFlt)) + N) % N;
    const uint32_t circleIdxNext = (circleIdxLast + 1u) % N;
    const float interpolationFactorCircle = circleIdxFlt - floorf(circleIdxFlt);
    const float aoFactor00 = S.bakedAo[circleIdxLast + size_t(N) * lastVertexIdx];
    const float aoFactor01 = S.bakedAo[circleIdxLast + size_t(N) * nextVertexIdx];
    const float aoFactor10 = S.bakedAo[circleIdxNext + size_t(N) * lastVertexIdx];
    const float aoFactor11 = S.bakedAo[circleIdxNext + size_t(N) * nextVertexIdx];
    const float aoFactor0 = mixf(aoFactor00, aoFactor01, interpolationFactorLine);
    const float aoFactor1 = mixf(aoFactor10, aoFactor11, interpolationFactorLine);
    return mixf(aoFactor0, aoFactor1, interpolationFactorCircle);
}

// ClosestHitTubeAnalytic + computeFragmentColor + blinnPhongShadingTube for flow lines.
// aoTexel: AO factor of the pixel that launched the ray (lookup definition: DESIGN.md).  Returns payload.hitColor;
// payloadHitT = length(hit - camera).
template <int BANDS = LV_SHADE_PLAIN, int FAST = 0>   // FAST: 0 | 1 (lighting only: the ray tracer's alpha follows the halo coordinate)
__device__ __forceinline__ f4 lv_shade_hit(const LvSceneDev& S, const LvUniforms& U, float aoTexel, f3 o, f3 d,
                                           const LvHit& h, float& payloadHitT, bool raster = false, LvRasterQuad rqv = LvRasterQuad(),
                                           bool rasterApply = true) {
    // raster: compute fwidth(ribbonPosition) (a compile-time constant at every call site: the gather passes true, the ray tracer
    // nothing); rasterApply: use it (run-time option ppll_fragment_colour) -- the computation itself is branch-free so that it
    // interleaves with the rest of the shading instead of forming a serial tail of its own.  The quad travels by value: behind a
    // conditionally null pointer it lived in scratch memory, one store + one dependent load per shading batch on the critical path.
    const LvRasterQuad* rq = raster ? &rqv : nullptr;
    const float4 ra = S.segs[2 * h.leaf], rb = S.segs[2 * h.leaf + 1];
    const f3 P0 = mk3(ra.x, ra.y, ra.z), P1 = mk3(rb.x, rb.y, rb.z);
    f3 fragPos = o + d * h.t;
    f3 linePointInterpolated;
    float fragmentAttribute;
    f3 v = P1 - P0;
    if (h.kind == 0) {
        f3 u = fragPos - P0;
        float t = dot3(v, u) / dot3(v, v);
        linePointInterpolated = P0 + t * v;
        fragmentAttribute = (1.0f - t) * ra.w + t * rb.w;
    } else if (h.kind == 1) {
        linePointInterpolated = P0;
        fragmentAttribute = ra.w;
    } else {
        linePointInterpolated = P1;
        fragmentAttribute = rb.w;
    }
    f3 fragmentTangent = norm3(v);
    f3 fragmentNormal = norm3(fragPos - linePointInterpolated);
    const bool isCap = h.kind != 0;
    // raster variant (PPLL gather): fwidth of the ribbon coordinate over the 2 x 2 quad (LvRasterQuad); circular tubes
    float rasterEps = -1.0f;
    if (rq && BANDS != LV_SHADE_BANDS) {
        const f3 cam = mk3(U.camPos[0], U.camPos[1], U.camPos[2]);
        const bool cap = U.useCappedTubes && isCap;
        const float f0 = cap ? lv_cap_ribbon_of_ray(cam, rq->d0, fragPos, fragmentNormal, linePointInterpolated, fragmentTangent)
                             : lv_tube_ribbon_of_ray(cam, rq->d0, linePointInterpolated, fragmentTangent, U.radius);
        const float fx = cap ? lv_cap_ribbon_of_ray(cam, rq->dX, fragPos, fragmentNormal, linePointInterpolated, fragmentTangent)
                             : lv_tube_ribbon_of_ray(cam, rq->dX, linePointInterpolated, fragmentTangent, U.radius);
        const float fy = cap ? lv_cap_ribbon_of_ray(cam, rq->dY, fragPos, fragmentNormal, linePointInterpolated, fragmentTangent)
                             : lv_tube_ribbon_of_ray(cam, rq->dY, linePointInterpolated, fragmentTangent, U.radius);
        rasterEps = rasterApply ? fabsf(fx - f0) + fabsf(fy - f0) : -1.0f;
    }
    if (U.aoPrebaked) {
        // TubeRayTracing.glsl:551-563: angle around the tube relative to the line normal + interpolated vertex id
        const uint32_t seg = S.leafSeg[h.leaf];
        const uint32_t i0 = S.segIdx[2 * seg], i1 = S.segIdx[2 * seg + 1];
        const lv_line_point& lp0 = S.points[i0];
        const lv_line_point& lp1 = S.points[i1];
        const float ts = h.kind == 0 ? dot3(v, fragPos - P0) / dot3(v, v) : (h.kind == 1 ? 0.0f : 1.0f);
        const f3 lineNormal = (1.0f - ts) * mk3(lp0.lineNormal[0], lp0.lineNormal[1], lp0.lineNormal[2]) +
                              ts * mk3(lp1.lineNormal[0], lp1.lineNormal[1], lp1.lineNormal[2]);
        float phi = acosf(clampf(dot3(fragmentNormal, lineNormal), -1.0f, 1.0f));
        const float val = dot3(lineNormal, cross3(fragmentNormal, fragmentTangent));
        if (val < 0.0f) phi = 2.0f * 3.14159265358979323846f - phi;
        const float fragmentVertexId = (1.0f - ts) * float(i0) + ts * float(i1);
        aoTexel = lv_prebaked_ao_lookup(S, U, fragmentVertexId, phi);
    }
    if (BANDS == LV_SHADE_HELICITY) {
        // USE_ROTATING_HELICITY_BANDS, TubeRayTracing.glsl:551-567: phi as for the AO lookup (acos through the build's atan2),
        // fragmentRotation = lerp(lineRotation) * helicityRotationFactor
        const uint32_t seg = S.leafSeg[h.leaf];
        const lv_line_point& lp0 = S.points[S.segIdx[2 * seg]];
        const lv_line_point& lp1 = S.points[S.segIdx[2 * seg + 1]];
        const float ts = h.kind == 0 ? dot3(v, fragPos - P0) / dot3(v, v) : (h.kind == 1 ? 0.0f : 1.0f);
        LvBandArgs b;
        b.useBand = false;
        b.lineNormal = (1.0f - ts) * mk3(lp0.lineNormal[0], lp0.lineNormal[1], lp0.lineNormal[2]) +
                       ts * mk3(lp1.lineNormal[0], lp1.lineNormal[1], lp1.lineNormal[2]);
        const float cphi = clampf(dot3(fragmentNormal, b.lineNormal), -1.0f, 1.0f);
        b.phi = lv_atan2_det(sqrtf((1.0f - cphi) * (1.0f + cphi)), cphi);
        if (dot3(b.lineNormal, cross3(fragmentNormal, fragmentTangent)) < 0.0f) b.phi = 2.0f * 3.14159265358979323846f - b.phi;
        b.linePosition = linePointInterpolated;
        b.rotation = ((1.0f - ts) * lp0.lineRotation + ts * lp1.lineRotation) * U.helicityRotationFactor;
        b.separatorScale = 1.0f; // ClosestHitTubeAnalytic has no UNIFORM_HELICITY_BAND_WIDTH branch
        b.rasterEpsWhite = rasterEps;
        return lv_compute_fragment_color_t<LV_SHADE_HELICITY>(S, U, aoTexel, fragPos, fragmentNormal, fragmentTangent, isCap,
                                                              fragmentAttribute, payloadHitT, b);
    }
    if (BANDS == LV_SHADE_BANDS) {
        // band data with the circular analytic tubes: USE_BANDS is defined, ANALYTIC_TUBE_INTERSECTIONS sets useBand = false
        // (RayHitCommon.glsl:164-166); phi and the line normal as TubeRayTracing.glsl:551-560 (acos through the build's atan2)
        const uint32_t seg = S.leafSeg[h.leaf];
        const lv_line_point& lp0 = S.points[S.segIdx[2 * seg]];
        const lv_line_point& lp1 = S.points[S.segIdx[2 * seg + 1]];
        const float ts = h.kind == 0 ? dot3(v, fragPos - P0) / dot3(v, v) : (h.kind == 1 ? 0.0f : 1.0f);
        LvBandArgs b;
        b.useBand = false;
        b.lineNormal = (1.0f - ts) * mk3(lp0.lineNormal[0], lp0.lineNormal[1], lp0.lineNormal[2]) +
                       ts * mk3(lp1.lineNormal[0], lp1.lineNormal[1], lp1.lineNormal[2]);
        const float cphi = clampf(dot3(fragmentNormal, b.lineNormal), -1.0f, 1.0f);
        b.phi = lv_atan2_det(sqrtf((1.0f - cphi) * (1.0f + cphi)), cphi);
        if (dot3(b.lineNormal, cross3(fragmentNormal, fragmentTangent)) < 0.0f) b.phi = 2.0f * 3.14159265358979323846f - b.phi;
        b.linePosition = linePointInterpolated;
        b.rotation = 0.0f; b.separatorScale = 1.0f;
        b.rasterEpsWhite = -1.0f;
        if (rq) {
            // raster variant with USE_BANDS: the band coordinate of the partner rays in the same cross-section plane; caps keep the
            // cap coordinate (the isCap branch precedes the USE_BANDS branch, LinePassGeometryShaderTubes.glsl:785-818)
            const f3 cam = mk3(U.camPos[0], U.camPos[1], U.camPos[2]);
            const f3 tN = norm3(fragmentTangent);
            const bool cap = U.useCappedTubes && isCap;
            const float r = U.lineWidth * 0.5f;
            const float f0 = cap ? lv_cap_ribbon_of_ray(cam, rq->d0, fragPos, fragmentNormal, linePointInterpolated, fragmentTangent)
                                 : lv_bands_ribbon_of_ray(cam, rq->d0, b.linePosition, b.lineNormal, fragmentTangent, tN, r, 1.0f);
            const float fx = cap ? lv_cap_ribbon_of_ray(cam, rq->dX, fragPos, fragmentNormal, linePointInterpolated, fragmentTangent)
                                 : lv_bands_ribbon_of_ray(cam, rq->dX, b.linePosition, b.lineNormal, fragmentTangent, tN, r, 1.0f);
            const float fy = cap ? lv_cap_ribbon_of_ray(cam, rq->dY, fragPos, fragmentNormal, linePointInterpolated, fragmentTangent)
                                 : lv_bands_ribbon_of_ray(cam, rq->dY, b.linePosition, b.lineNormal, fragmentTangent, tN, r, 1.0f);
            b.rasterEpsWhite = rasterApply ? fabsf(fx - f0) + fabsf(fy - f0) : -1.0f;
        }
        return lv_compute_fragment_color_t<LV_SHADE_BANDS>(S, U, aoTexel, fragPos, fragmentNormal, fragmentTangent, isCap,
                                                           fragmentAttribute, payloadHitT, b);
    }
    LvBandArgs none;
    none.useBand = false; none.phi = 0.0f; none.linePosition = mk3(0.0f, 0.0f, 0.0f); none.lineNormal = mk3(0.0f, 0.0f, 0.0f);
    none.rotation = 0.0f; none.separatorScale = 1.0f; none.rasterEpsWhite = rasterEps;
    return lv_compute_fragment_color_t<LV_SHADE_PLAIN, FAST>(S, U, aoTexel, fragPos, fragmentNormal, fragmentTangent, isCap, fragmentAttribute,
                                                             payloadHitT, none);
}

// Fragment stage of the rasterised programmable-pull prism (ppll_fragment_source = raster_prism; lv_prism.h):
// LinePassGeometryShaderTubes.glsl:732-1129 on the perspective-correct inputs of triangle tt of the leaf's segment for the pixel's
// viewing ray (o, d); fwidth(ribbonPosition) (:1079-1087) from the helper invocations of the 2 x 2 quad = the same triangle's
// attribute planes at the quad partners' rays (LvRasterQuad).
// USE_BANDS fragment stage (LinePassGeometryShaderTubes.glsl:819-936): the band's halo coordinate at interpolated varyings -- phi with the
// wrap-around of the last facet (:761-775), linePosition, lineNormal, fragmentTangent; the polar construction of lv_bands_ribbon_of_point
__device__ __forceinline__ float lv_prism_band_ribbon(const LvPrismDev& R, f3 cam, const LvPrismTri& T, const LvPrismPoint pt[2],
                                                      const uint32_t pi[2], const LvPrismInputs& I) {
    float fragmentVertexId, phi;
    lv_prism_ao_inputs(T, pt, pi, I.b, R.n, fragmentVertexId, phi);
    const f3 c0 = T.second[0] ? pt[1].centre : pt[0].centre, c1 = T.second[1] ? pt[1].centre : pt[0].centre, c2 = T.second[2] ? pt[1].centre : pt[0].centre;
    const f3 n0 = T.second[0] ? pt[1].normal : pt[0].normal, n1 = T.second[1] ? pt[1].normal : pt[0].normal, n2 = T.second[2] ? pt[1].normal : pt[0].normal;
    float sp, cp;
    lv_sincos_rad(phi, sp, cp);
    return lv_bands_ribbon_of_point(cam, lv_prism_mix3(I.b, c0, c1, c2), lv_prism_mix3(I.b, n0, n1, n2), I.tan, norm3s(I.tan),
                                    mk3(R.thickness * cp, sp, 1.0f), R.radius, R.thickness);
}
// FAST = 2 (plain tubes with the raster colour only; the caller checks): vertex normals, the quad partners' inputs, the three halo
// coordinates and the whole colour through the approximate operations; the own ray's weights, position, attribute, depth and the
// acceptance rules stay exact -- the fragment's {depth, alpha, kept} are bit for bit those of the exact mode
template <int SHADE = LV_SHADE_PLAIN, int FAST = 0>
__device__ __forceinline__ f4 lv_shade_prism(const LvSceneDev& S, const LvUniforms& U, const float* ringTab, float aoTexel, f3 o, f3 d,
                                             float tLo, float tHi, uint32_t leaf, uint32_t tt, const LvRasterQuad& rq, bool rasterApply,
                                             float& payloadHitT, bool& kept) {
    constexpr bool FH = FAST >= 2 && SHADE == LV_SHADE_PLAIN;
    const LvPrismDev& R = S.prism;
    uint32_t pi[2];
    LvPrismPoint pt[2];
    lv_prism_frames(S, leaf, S.segs[2 * size_t(leaf)], S.segs[2 * size_t(leaf) + 1], pt, pi);
    const LvPrismTri T = lv_prism_tri_setup(ringTab, R.n, pt, pi, R.radius, tt);
    f3 nrm[3];
#pragma unroll
    for (int i = 0; i < 3; i++) nrm[i] = norm3q<FH>(T.dir[i]);   // vertexNormal = normalize(tangentFrameMatrix * localNormal), :177
    const f3 cam = mk3(U.camPos[0], U.camPos[1], U.camPos[2]);
    const LvPrismPlanes pl = lv_prism_planes(R, T, cam, d);   // (o == cam: the pixel's viewing ray starts at the camera)
    const LvPrismInputs I = lv_prism_interpolate(T, nrm, pl, d);
    kept = lv_prism_accept(R, pt, R.radius, o, d, I.pos, len3(I.pos - o), tLo, tHi);
    if (SHADE == LV_SHADE_BANDS) {
        LvBandArgs b;
        b.useBand = true;
        float fragmentVertexId;
        lv_prism_ao_inputs(T, pt, pi, I.b, R.n, fragmentVertexId, b.phi);
        if (U.aoPrebaked) aoTexel = lv_prebaked_ao_lookup(S, U, fragmentVertexId, b.phi);   // getAoFactor(fragmentVertexId, phi), Lighting.glsl:124-125
        const f3 c0 = T.second[0] ? pt[1].centre : pt[0].centre, c1 = T.second[1] ? pt[1].centre : pt[0].centre, c2 = T.second[2] ? pt[1].centre : pt[0].centre;
        const f3 n0 = T.second[0] ? pt[1].normal : pt[0].normal, n1 = T.second[1] ? pt[1].normal : pt[0].normal, n2 = T.second[2] ? pt[1].normal : pt[0].normal;
        b.linePosition = lv_prism_mix3(I.b, c0, c1, c2);
        b.lineNormal = lv_prism_mix3(I.b, n0, n1, n2);
        b.rotation = 0.0f; b.separatorScale = 1.0f;
        const LvPrismInputs Ix = lv_prism_interpolate(T, nrm, pl, rq.dX), Iy = lv_prism_interpolate(T, nrm, pl, rq.dY);
        const float f0 = lv_prism_band_ribbon(R, cam, T, pt, pi, I), fx = lv_prism_band_ribbon(R, cam, T, pt, pi, Ix),
                    fy = lv_prism_band_ribbon(R, cam, T, pt, pi, Iy);
        b.rasterEpsWhite = rasterApply ? fabsf(fx - f0) + fabsf(fy - f0) : -1.0f;
        return lv_compute_fragment_color_t<LV_SHADE_BANDS>(S, U, aoTexel, I.pos, I.nrm, I.tan, false, I.attr, payloadHitT, b);
    }
    if (U.aoPrebaked) {   // getAoFactor(fragmentVertexId, phi) of the static prebaker instead of the screen-space texel
        float fragmentVertexId, phi;
        lv_prism_ao_inputs(T, pt, pi, I.b, R.n, fragmentVertexId, phi);
        aoTexel = lv_prebaked_ao_lookup(S, U, fragmentVertexId, phi);
    }
    const LvPrismInputs Ix = lv_prism_interpolate<FH>(T, nrm, pl, rq.dX), Iy = lv_prism_interpolate<FH>(T, nrm, pl, rq.dY);
    const float f0 = lv_prism_ribbon<FH>(cam, I.pos, I.nrm, I.tan);
    const float fx = lv_prism_ribbon<FH>(cam, Ix.pos, Ix.nrm, Ix.tan);
    const float fy = lv_prism_ribbon<FH>(cam, Iy.pos, Iy.nrm, Iy.tan);
    LvBandArgs none;
    none.useBand = false; none.phi = 0.0f; none.linePosition = mk3(0.0f, 0.0f, 0.0f); none.lineNormal = mk3(0.0f, 0.0f, 0.0f);
    none.rotation = 0.0f; none.separatorScale = 1.0f;
    none.rasterEpsWhite = rasterApply ? fabsf(fx - f0) + fabsf(fy - f0) : -1.0f;
    if (SHADE == LV_SHADE_HELICITY) {
        // USE_ROTATING_HELICITY_BANDS in the raster shaders: fragmentRotation = lineRotation * helicityRotationFactor per vertex
        // (LinePassProgrammablePullTubes.glsl:212-214), interpolated; phi as for the AO lookup; UNIFORM_HELICITY_BAND_WIDTH
        // (LinePassGeometryShaderTubes.glsl:1017-1034): the two line points around floor(fragmentVertexId), zeros past the buffer
        float fragmentVertexId;
        lv_prism_ao_inputs(T, pt, pi, I.b, R.n, fragmentVertexId, none.phi);
        const float fr = U.helicityRotationFactor;
        const float r0 = S.points[pi[0]].lineRotation * fr, r1 = S.points[pi[1]].lineRotation * fr;
        const float rot[3] = {T.second[0] ? r1 : r0, T.second[1] ? r1 : r0, T.second[2] ? r1 : r0};
        none.rotation = (I.b[0] * rot[0] + I.b[1] * rot[1]) + I.b[2] * rot[2];
        if (rasterApply) {   // the raster shader's stripe: aaf = fwidth(phi + fragmentRotation) from the quad partners' interpolated varyings
            float vid, phx, phy;
            lv_prism_ao_inputs(T, pt, pi, Ix.b, R.n, vid, phx);
            lv_prism_ao_inputs(T, pt, pi, Iy.b, R.n, vid, phy);
            const float gx = phx + ((Ix.b[0] * rot[0] + Ix.b[1] * rot[1]) + Ix.b[2] * rot[2]);
            const float gy = phy + ((Iy.b[0] * rot[0] + Iy.b[1] * rot[1]) + Iy.b[2] * rot[2]);
            const float g0 = none.phi + none.rotation;
            none.stripeAaf = fabsf(gx - g0) + fabsf(gy - g0);
            const float twoPi = 2.0f * 3.14159265358979323846f;
            none.stripeDx = gx / twoPi - g0 / twoPi;   // dFdx / dFdy of globalPos = (phi + fragmentRotation) / twoPi
            none.stripeDy = gy / twoPi - g0 / twoPi;
        }
        if (U.uniformHelicityBandWidth) {
            const uint32_t i0 = uint32_t(floorf(fragmentVertexId)), i1 = i0 + 1u;
            f3 p0 = mk3(0.0f, 0.0f, 0.0f), p1 = mk3(0.0f, 0.0f, 0.0f);
            float q0 = 0.0f, q1 = 0.0f;
            if (i0 < S.numPoints) { const lv_line_point& a = S.points[i0]; p0 = mk3(a.linePosition[0], a.linePosition[1], a.linePosition[2]); q0 = a.lineRotation; }
            if (i1 < S.numPoints) { const lv_line_point& a = S.points[i1]; p1 = mk3(a.linePosition[0], a.linePosition[1], a.linePosition[2]); q1 = a.lineRotation; }
            const float rotDx = len3(p1 - p0);
            const float rotDy = (q1 - q0) * fr;
            float sn, cs;
            lv_sincos_rad(lv_atan2_det(rotDy * 0.5f * U.lineWidth, rotDx), sn, cs);
            none.separatorScale = cs;
        }
        return lv_compute_fragment_color_t<LV_SHADE_HELICITY>(S, U, aoTexel, I.pos, I.nrm, I.tan, false, I.attr, payloadHitT, none);
    }
    return lv_compute_fragment_color_t<LV_SHADE_PLAIN, FH ? 2 : 0>(S, U, aoTexel, I.pos, I.nrm, I.tan, false, I.attr, payloadHitT, none);
}

// ClosestHitEllipticTubeAnalytic main(), EllipticTubeRayTracing.glsl:303-441: position in the tubelet frame -> t, phi, rho ->
// normal of the twisted elliptic surface; attribute and line normal interpolated with t; the angle in the ellipse's own
// parametrisation p = (r1 cos, r2 sin) for the band shading.
struct LvEllipticSurface { f3 fragPos, normal, tangent, linePosition, lineNormal; float t, phiLine, attribute; };
__device__ __forceinline__ LvEllipticSurface lv_elliptic_surface(const LvUniforms& U, f3 o, f3 d, float hitT,
                                                                 const lv_line_point& lp0, const lv_line_point& lp1) {
    LvEllipticSurface E;
    E.fragPos = o + d * hitT;
    const LvTubelet T = lv_make_tubelet(lp0, lp1);
    const f3 p = lv_to_tubelet(T, E.fragPos - T.p0);
    const float radius0 = U.bandWidth * 0.5f * U.minBandThickness;
    const float radius1 = U.bandWidth * 0.5f;
    const float t = clampf(p.x / T.l, 0.0f, 1.0f);
    const float phi = lv_atan2_det(p.z, p.y);
    const float rho = t * T.rhoR;
    E.t = t;
    E.normal = norm3(lv_from_tubelet(T, lv_ell_compute_normal(radius0, radius1, phi, rho)));
    E.attribute = (1.0f - t) * lp0.lineAttribute + t * lp1.lineAttribute;
    E.linePosition = (1.0f - t) * T.p0 + t * T.p1;
    E.tangent = T.xt;
    E.lineNormal = norm3((1.0f - t) * mk3(lp0.lineNormal[0], lp0.lineNormal[1], lp0.lineNormal[2]) +
                         t * mk3(lp1.lineNormal[0], lp1.lineNormal[1], lp1.lineNormal[2]));
    float sinphi, cosphi;
    lv_sincos_rad(phi + rho, sinphi, cosphi);
    const float phiDenomInv = 1.0f / sqrtf(radius0 * radius0 * sinphi * sinphi + radius1 * radius1 * cosphi * cosphi);
    const float sinPhiLine = radius0 * sinphi * phiDenomInv;
    const float cosPhiLine = radius1 * cosphi * phiDenomInv;
    const float TWO_PI = 6.283185307f; // M_TWO_PI as the shader spells it
    const float a = lv_atan2_det(sinPhiLine, cosPhiLine) + TWO_PI;
    E.phiLine = a - TWO_PI * floorf(a / TWO_PI); // mod(x, y) = x - y * floor(x / y)
    return E;
}
__device__ __forceinline__ f4 lv_shade_hit_elliptic(const LvSceneDev& S, const LvUniforms& U, float aoTexel, f3 o, f3 d,
                                                    const LvHit& h, float& payloadHitT, bool raster = false, LvRasterQuad rqv = LvRasterQuad(),
                                           bool rasterApply = true) {
    // raster: compute fwidth(ribbonPosition) (a compile-time constant at every call site: the gather passes true, the ray tracer
    // nothing); rasterApply: use it (run-time option ppll_fragment_colour) -- the computation itself is branch-free so that it
    // interleaves with the rest of the shading instead of forming a serial tail of its own.  The quad travels by value: behind a
    // conditionally null pointer it lived in scratch memory, one store + one dependent load per shading batch on the critical path.
    const LvRasterQuad* rq = raster ? &rqv : nullptr;
    const uint32_t seg = S.leafSeg[h.leaf];
    const lv_line_point& lp0 = S.points[S.segIdx[2 * seg]];
    const lv_line_point& lp1 = S.points[S.segIdx[2 * seg + 1]];
    const LvEllipticSurface E = lv_elliptic_surface(U, o, d, h.t, lp0, lp1);
    if (U.aoPrebaked) {   // getAoFactor(fragmentVertexId, phiLine), EllipticTubeRayTracing.glsl:393-395,420-431
        const float fragmentVertexId = (1.0f - E.t) * float(S.segIdx[2 * seg]) + E.t * float(S.segIdx[2 * seg + 1]);
        aoTexel = lv_prebaked_ao_lookup(S, U, fragmentVertexId, E.phiLine);
    }
    LvBandArgs b;
    b.useBand = true;
    b.phi = E.phiLine;
    b.linePosition = E.linePosition;
    b.lineNormal = E.lineNormal;
    b.rotation = 0.0f; b.separatorScale = 1.0f;
    b.rasterEpsWhite = -1.0f;
    if (rq) {
        const f3 cam = mk3(U.camPos[0], U.camPos[1], U.camPos[2]);
        const f3 tN = norm3(E.tangent);
        const float r = U.bandWidth * 0.5f;
        const float f0 = lv_bands_ribbon_of_ray(cam, rq->d0, b.linePosition, b.lineNormal, E.tangent, tN, r, U.minThickness);
        const float fx = lv_bands_ribbon_of_ray(cam, rq->dX, b.linePosition, b.lineNormal, E.tangent, tN, r, U.minThickness);
        const float fy = lv_bands_ribbon_of_ray(cam, rq->dY, b.linePosition, b.lineNormal, E.tangent, tN, r, U.minThickness);
        b.rasterEpsWhite = rasterApply ? fabsf(fx - f0) + fabsf(fy - f0) : -1.0f;
    }
    return lv_compute_fragment_color_t<LV_SHADE_BANDS>(S, U, aoTexel, E.fragPos, E.normal, E.tangent, false, E.attribute, payloadHitT, b);
}

// ClosestHitTubeTriangles (TubeRayTracing.glsl:301-352) + LineAttributesBarycentric.glsl: the ray tracer's "Triangle
// Mesh" geometry mode.  tri = original triangle index; (u, v) are recomputed with the test that won the traversal.
template <int BANDS = LV_SHADE_PLAIN>
__device__ __forceinline__ f4 lv_shade_hit_triangle(const LvSceneDev& S, const LvUniforms& U, float aoTexel, f3 o, f3 d,
                                                    uint32_t tri, float& payloadHitT) {
    const uint32_t i0 = S.triIdx[3 * size_t(tri)], i1 = S.triIdx[3 * size_t(tri) + 1], i2 = S.triIdx[3 * size_t(tri) + 2];
    const lv_tube_vertex& vd0 = S.triVerts[i0];
    const lv_tube_vertex& vd1 = S.triVerts[i1];
    const lv_tube_vertex& vd2 = S.triVerts[i2];
    const f3 p0 = mk3(vd0.vertexPosition[0], vd0.vertexPosition[1], vd0.vertexPosition[2]);
    const f3 p1 = mk3(vd1.vertexPosition[0], vd1.vertexPosition[1], vd1.vertexPosition[2]);
    const f3 p2 = mk3(vd2.vertexPosition[0], vd2.vertexPosition[1], vd2.vertexPosition[2]);
    float tt = 0.0f, bu = 0.0f, bv = 0.0f;
    lv_ray_triangle(o, d, mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z), p0, p1, p2, S.triPad, tt, bu, bv);
    const float b0 = (1.0f - bu) - bv;
    const lv_line_point& lp0 = S.triPoints[vd0.vertexLinePointIndex & 0x7FFFFFFFu];
    const lv_line_point& lp1 = S.triPoints[vd1.vertexLinePointIndex & 0x7FFFFFFFu];
    const lv_line_point& lp2 = S.triPoints[vd2.vertexLinePointIndex & 0x7FFFFFFFu];
    auto lerp3 = [&](const float* a, const float* b, const float* c) {
        return (mk3(a[0], a[1], a[2]) * b0 + mk3(b[0], b[1], b[2]) * bu) + mk3(c[0], c[1], c[2]) * bv;
    };
    const bool isCap = U.useCappedTubes &&
            (((vd0.vertexLinePointIndex | vd1.vertexLinePointIndex | vd2.vertexLinePointIndex) >> 31) != 0u);
    const f3 fragPos = (p0 * b0 + p1 * bu) + p2 * bv;
    const f3 fragmentNormal = norm3(lerp3(vd0.vertexNormal, vd1.vertexNormal, vd2.vertexNormal));
    const f3 fragmentTangent = norm3(lerp3(lp0.lineTangent, lp1.lineTangent, lp2.lineTangent));
    const float fragmentAttribute = (lp0.lineAttribute * b0 + lp1.lineAttribute * bu) + lp2.lineAttribute * bv;
    if (U.aoPrebaked) {
        // LineAttributesBarycentric.glsl:44-52: interpolateAngle (BarycentricInterpolation.glsl:43-55) + vertex id
        const float PI = 3.14159265358979323846f;
        float a0 = vd0.phi, a1 = vd1.phi, a2 = vd2.phi;
        if (a1 - a0 > PI || a2 - a0 > PI) a0 += 2.0f * PI;
        if (a0 - a1 > PI || a2 - a1 > PI) a1 += 2.0f * PI;
        if (a0 - a2 > PI || a1 - a2 > PI) a2 += 2.0f * PI;
        const float phi = (a0 * b0 + a1 * bu) + a2 * bv;
        const float fragmentVertexId = (float(vd0.vertexLinePointIndex & 0x7FFFFFFFu) * b0 +
                                        float(vd1.vertexLinePointIndex & 0x7FFFFFFFu) * bu) +
                                       float(vd2.vertexLinePointIndex & 0x7FFFFFFFu) * bv;
        aoTexel = lv_prebaked_ao_lookup(S, U, fragmentVertexId, phi);
    }
    if (BANDS == LV_SHADE_HELICITY) {
        // USE_ROTATING_HELICITY_BANDS in the triangle closest-hit shader (LineAttributesBarycentric.glsl:43-92): interpolated angle,
        // interpolated lineRotation x helicityRotationFactor; on the caps the rotation is continued linearly along the line
        const float PI = 3.14159265358979323846f;
        float a0 = vd0.phi, a1 = vd1.phi, a2 = vd2.phi;
        if (a1 - a0 > PI || a2 - a0 > PI) a0 += 2.0f * PI;
        if (a0 - a1 > PI || a2 - a1 > PI) a1 += 2.0f * PI;
        if (a0 - a2 > PI || a1 - a2 > PI) a2 += 2.0f * PI;
        LvBandArgs b;
        b.useBand = false;
        b.phi = (a0 * b0 + a1 * bu) + a2 * bv;
        b.linePosition = mk3(0.0f, 0.0f, 0.0f);
        b.lineNormal = mk3(0.0f, 0.0f, 0.0f);
        const float f = U.helicityRotationFactor;
        b.rotation = ((lp0.lineRotation * f) * b0 + (lp1.lineRotation * f) * bu) + (lp2.lineRotation * f) * bv;
        if (isCap) {
            const uint32_t li0 = vd0.vertexLinePointIndex & 0x7FFFFFFFu;
            const lv_line_point* other = nullptr;
            if (li0 != 0u && S.triPoints[li0 - 1u].lineStartIndex == lp0.lineStartIndex) other = &S.triPoints[li0 - 1u];
            if (!other) other = &S.triPoints[li0 + 1u];
            const float fragmentRotationDelta = (lp0.lineRotation - other->lineRotation) * f;
            f3 planeNormal = mk3(lp0.linePosition[0], lp0.linePosition[1], lp0.linePosition[2]) -
                             mk3(other->linePosition[0], other->linePosition[1], other->linePosition[2]);
            const float segmentLength = len3(planeNormal);
            planeNormal = mk3(planeNormal.x / segmentLength, planeNormal.y / segmentLength, planeNormal.z / segmentLength);
            const float planeDist = -dot3(planeNormal, mk3(lp0.linePosition[0], lp0.linePosition[1], lp0.linePosition[2]));
            const float distToPlane = dot3(planeNormal, fragPos) + planeDist;
            b.rotation += fragmentRotationDelta * distToPlane / segmentLength;
        }
        b.separatorScale = 1.0f;
        if (U.uniformHelicityBandWidth) {
            // UNIFORM_HELICITY_BAND_WIDTH, LineAttributesBarycentric.glsl:94-112: rotation per length along the line against the
            // circumference per angle (r = lineWidth / 2)
            const uint32_t li0 = vd0.vertexLinePointIndex & 0x7FFFFFFFu;
            const lv_line_point* other = nullptr;
            float rotDy = 0.0f;
            if (li0 != 0u && S.triPoints[li0 - 1u].lineStartIndex == lp0.lineStartIndex) {
                other = &S.triPoints[li0 - 1u];
                rotDy = (lp0.lineRotation - other->lineRotation) * f;
            }
            if (!other) {
                other = &S.triPoints[li0 + 1u];
                rotDy = (other->lineRotation - lp0.lineRotation) * f;
            }
            const float rotDx = len3(mk3(lp0.linePosition[0], lp0.linePosition[1], lp0.linePosition[2]) -
                                     mk3(other->linePosition[0], other->linePosition[1], other->linePosition[2]));
            float sn, cs;
            lv_sincos_rad(lv_atan2_det(rotDy * 0.5f * U.lineWidth, rotDx), sn, cs);
            b.separatorScale = cs;
        }
        b.rasterEpsWhite = -1.0f;
        return lv_compute_fragment_color_t<LV_SHADE_HELICITY>(S, U, aoTexel, fragPos, fragmentNormal, fragmentTangent, isCap,
                                                              fragmentAttribute, payloadHitT, b);
    }
    if (BANDS == LV_SHADE_BANDS) {
        // USE_BANDS in the triangle closest-hit shader (LineAttributesBarycentric.glsl:44-63): interpolated angle, line position and
        // line normal; useBand = true (no ANALYTIC_TUBE_INTERSECTIONS here, RayHitCommon.glsl:164-172)
        const float PI = 3.14159265358979323846f;
        float a0 = vd0.phi, a1 = vd1.phi, a2 = vd2.phi;
        if (a1 - a0 > PI || a2 - a0 > PI) a0 += 2.0f * PI;
        if (a0 - a1 > PI || a2 - a1 > PI) a1 += 2.0f * PI;
        if (a0 - a2 > PI || a1 - a2 > PI) a2 += 2.0f * PI;
        LvBandArgs b;
        b.useBand = true;
        b.phi = (a0 * b0 + a1 * bu) + a2 * bv;
        b.linePosition = lerp3(lp0.linePosition, lp1.linePosition, lp2.linePosition);
        b.lineNormal = lerp3(lp0.lineNormal, lp1.lineNormal, lp2.lineNormal);
        b.rotation = 0.0f; b.separatorScale = 1.0f; b.rasterEpsWhite = -1.0f;
        return lv_compute_fragment_color_t<LV_SHADE_BANDS>(S, U, aoTexel, fragPos, fragmentNormal, fragmentTangent, isCap, fragmentAttribute,
                                                           payloadHitT, b);
    }
    return lv_compute_fragment_color(S, U, aoTexel, fragPos, fragmentNormal, fragmentTangent, isCap, fragmentAttribute,
                                     payloadHitT);
}

// computeFragmentColor (RayHitCommon.glsl:74-543) for tubes, shared by the analytic and the triangle closest-hit paths
// FAST (shading_numerics = fast, lv_device.h): 1 = the lighting (blinnPhongShadingTube, the AO mapping, the depth cue, the white halo's
// blend weight) through the approximate hardware operations, the halo coordinate and everything else that reaches alpha exact;
// 2 = the halo coordinate too -- only where alpha cannot depend on it: the raster colour of plain tubes (EPSILON_OUTLINE = 0 and
// |ribbonPosition| <= 1 by its clamp: coverage is 1 whatever the last bits say)
template <int BANDS, int FAST>
__device__ __forceinline__ f4 lv_compute_fragment_color_t(const LvSceneDev& S, const LvUniforms& U, float aoTexel, f3 fragPos,
                                                          f3 fragmentNormal, f3 fragmentTangent, bool isCap,
                                                          float fragmentAttribute, float& payloadHitT, const LvBandArgs& bands) {
    constexpr bool FL = FAST >= 1, FH = FAST >= 2;
    const f3 cam = mk3(U.camPos[0], U.camPos[1], U.camPos[2]);
    f4 fragmentColor = lv_transfer_function(S, U, fragmentAttribute);
    f3 n = norm3q<FH>(fragmentNormal);
    f3 vv = norm3q<FH>(cam - fragPos);
    f3 t = norm3q<FH>(fragmentTangent);
    f3 helperVec = norm3q<FH>(cross3(t, vv));
    f3 newV = norm3q<FH>(cross3(helperVec, t));

    float ribbonPosition = 0.0f;
    if (U.useHalos) {
        if (U.useCappedTubes && isCap) {
            f3 crossProdVn = cross3(vv, n);
            ribbonPosition = len3(crossProdVn);
            f3 crossProdVn2 = cross3(newV, n);
            float ribbonPosition2 = len3(crossProdVn2);
            if (dot3(t, crossProdVn) < 0.0f) ribbonPosition2 = -ribbonPosition2;
            if (dot3(t, crossProdVn) < 0.0f) ribbonPosition = -ribbonPosition;
            ribbonPosition2 = clampf(ribbonPosition2, -1.0f, 1.0f);
            if (fabsf(ribbonPosition2) < fabsf(ribbonPosition)) ribbonPosition = ribbonPosition2;
            // raster variant, LinePassGeometryShaderTubes.glsl:785-815: ribbonPosition = min(length(cross(v, n)), abs(ribbonPosition2))
            if (bands.rasterEpsWhite >= 0.0f) ribbonPosition = fminf(len3(crossProdVn), fabsf(ribbonPosition2));
        } else if (BANDS == LV_SHADE_BANDS) {
            // USE_BANDS, RayHitCommon.glsl:232-351 (lv_bands_ribbon_of_point)
            const float thickness = bands.useBand ? U.minThickness : 1.0f;
            float sp, cp;
            lv_sincos_rad(bands.phi, sp, cp);
            ribbonPosition = lv_bands_ribbon_of_point(cam, bands.linePosition, bands.lineNormal, fragmentTangent, t,
                                                      mk3(thickness * cp, sp, 1.0f),
                                                      (bands.useBand ? U.bandWidth : U.lineWidth) * 0.5f, thickness);
        } else {
            f3 crossProdVn = cross3(newV, n);
            ribbonPosition = FH ? __builtin_amdgcn_sqrtf(dot3(crossProdVn, crossProdVn)) : len3(crossProdVn);
            if (dot3(t, crossProdVn) < 0.0f) ribbonPosition = -ribbonPosition;
            ribbonPosition = clampf(ribbonPosition, -1.0f, 1.0f);
        }
    }

    // blinnPhongShadingTube
    float kA, kD;
    const float kS = 0.3f, s = 30.0f;
    float aoF = 1.0f;
    if (U.useAmbientOcclusion) {
        if (U.aoProjectLookup) {
            // getAoFactor literally: ndc = projectionMatrix * vec4(screenSpacePosition, 1); texture(aoTexture, ndc.xy / ndc.w
            // * 0.5 + 0.5).x with a linear, clamp-to-edge sampler (texel centres at (i + 0.5) / size).  A jittered sample's hit
            // lies anywhere inside its pixel, so the neighbouring texels take part (S.ao then carries a 1-pixel halo around
            // the rendered tiles, lv_run_ao).
            const f4 s4 = mulM4(U.view, fragPos.x, fragPos.y, fragPos.z, 1.0f);
            const f4 ndc = mulM4(U.proj, s4.x, s4.y, s4.z, 1.0f);
            const float u = (ndc.x / ndc.w) * 0.5f + 0.5f, v = (ndc.y / ndc.w) * 0.5f + 0.5f;
            const float fx = u * float(U.width) - 0.5f, fy = v * float(U.height) - 0.5f;
            const float x0f = floorf(fx), y0f = floorf(fy);
            const float wx = fx - x0f, wy = fy - y0f;
            const uint32_t xa = uint32_t(fminf(fmaxf(x0f, 0.0f), float(U.width - 1u)));
            const uint32_t xb = uint32_t(fminf(fmaxf(x0f + 1.0f, 0.0f), float(U.width - 1u)));
            const uint32_t ya = uint32_t(fminf(fmaxf(y0f, 0.0f), float(U.height - 1u)));
            const uint32_t yb = uint32_t(fminf(fmaxf(y0f + 1.0f, 0.0f), float(U.height - 1u)));
            const float top = S.ao[size_t(ya) * U.width + xa] * (1.0f - wx) + S.ao[size_t(ya) * U.width + xb] * wx;
            const float bot = S.ao[size_t(yb) * U.width + xa] * (1.0f - wx) + S.ao[size_t(yb) * U.width + xb] * wx;
            aoTexel = top * (1.0f - wy) + bot * wy;
        }
        float a = lv_powq<FL>(aoTexel, U.aoGamma);
        aoF = fmaxf(0.0f, (1.0f - U.aoStrength) + U.aoStrength * a);
        kA = 0.2f + (1.0f - aoF) * 0.5f;
        kD = 0.9f * aoF;
    } else {
        kA = 0.1f;
        kD = 0.9f;
    }
    // blinnPhongShadingTube re-normalises its (already unit) arguments, Lighting.glsl:149-151
    const f3 nB = norm3q<FL>(n);
    const f3 tB = norm3q<FL>(t);
    f3 l = vv;
    f3 hh = norm3q<FL>(vv + l);
    f3 helperVecL = norm3q<FL>(cross3(tB, l));
    f3 newL = norm3q<FL>(cross3(helperVecL, tB));
    const float exponent = (BANDS == LV_SHADE_BANDS && bands.useBand) ? 1.0f : 1.7f; // Lighting.glsl:158-162
    float cosNormal1 = lv_powq<FL>(clampf(fabsf(dot3(nB, l)), 0.0f, 1.0f), exponent);
    float cosNormal2 = lv_powq<FL>(clampf(fabsf(dot3(nB, newL)), 0.0f, 1.0f), exponent);
    float cosNormalCombined = 0.3f * cosNormal1 + 0.7f * cosNormal2;
    float spec = kS * lv_powq<FL>(clampf(fabsf(dot3(nB, hh)), 0.0f, 1.0f), s);
    float base[3] = {fragmentColor.x, fragmentColor.y, fragmentColor.z};
    float phong[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float Ia = kA * base[k];
        float Id = (kD * cosNormalCombined) * base[k];
        float Is = spec * 1.0f;
        phong[k] = (Ia + Id) + Is;
    }
    if (U.useAmbientOcclusion) {
#pragma unroll
        for (int k = 0; k < 3; k++) phong[k] *= aoF;
    }
    if (U.useDepthCues) {
        f4 s4 = mulM4(U.view, fragPos.x, fragPos.y, fragPos.z, 1.0f);
        float minDepth = S.depthMinMax[0], maxDepth = S.depthMinMax[1];
        float dcf = clampf(lv_divq<FL>(-s4.z - minDepth, maxDepth - minDepth), 0.0f, 1.0f);
        dcf = (dcf * dcf) * U.depthCueStrength;
#pragma unroll
        for (int k = 0; k < 3; k++) phong[k] = mixf(phong[k], 0.5f, dcf);
    }

    float absCoords = U.useHalos ? fabsf(ribbonPosition) : 0.0f;
    float fragmentDepth = len3(fragPos - cam);   // (= payloadHitT: exact in every mode)
    // EPSILON_OUTLINE shapes the coverage, i.e. alpha: exact unless the raster variant replaces it by 0 anyway (FAST == 2)
    float aaO = lv_divq<FH>(lv_divq<FH>(fragmentDepth, U.lineWidth) * 0.05f, float(U.height)) * U.fovY;
    float aaW = lv_divq<FL>(lv_divq<FL>(fragmentDepth, U.lineWidth) * 2.0f, float(U.height)) * U.fovY;
    if (BANDS == LV_SHADE_BANDS) { // RayHitCommon.glsl:445-448
        const float wdt = bands.useBand ? U.bandWidth : U.lineWidth;
        aaO = aaW = ((fragmentDepth / wdt) * 0.25f) / float(U.height) * U.fovY;
    }
    float EPSILON_OUTLINE = clampf(aaO, 0.0f, 0.49f);
    float EPSILON_WHITE = clampf(aaW, 0.0f, 0.49f);
    float WHITE_THRESHOLD = 0.7f;
    if (BANDS == LV_SHADE_HELICITY) {
        // RayHitCommon.glsl:455-486 (no multi-var rendering, no twist-line texture, no UNIFORM_HELICITY_BAND_WIDTH):
        // drawSeparatorStripe (:57-64) darkens the shaded colour where mod(phi + rotation + w / 2, 2 pi / n) falls into [0, w]
        const float separatorWidth = U.separatorBaseWidth / bands.separatorScale; // :456-459 (scale 1 without the define)
        const float period = 2.0f / float(U.numSubdivisionsBands) * 3.14159265358979323846f;
        const bool rasterStripe = bands.stripeAaf >= 0.0f;
        const float x = bands.phi + bands.rotation + (rasterStripe ? 0.1f : separatorWidth * 0.5f);
        const float varFraction = x - period * floorf(x / period); // mod(x, y) = x - y * floor(x / y)
        const float aaf = rasterStripe ? bands.stripeAaf : EPSILON_OUTLINE * 10.0f;
        if (U.useTwistTexture) {
            // USE_HELICITY_BANDS_TEXTURE (RayHitCommon.glsl:465-469, LinePassGeometryShaderTubes.glsl:1043-1047): fragmentColor *=
            // texture at u = mod(phi + rotation + offset, 2 pi) / 2 pi (all four components: the alpha too)
            const float twoPi = 2.0f * 3.14159265358979323846f;
            const float tu = (x - twoPi * floorf(x / twoPi)) / twoPi;
            const f4 tex = lv_twist_sample(S, U, tu, bands.stripeDx, bands.stripeDy, rasterStripe);
            phong[0] = phong[0] * tex.x; phong[1] = phong[1] * tex.y; phong[2] = phong[2] * tex.z;
            fragmentColor.w = fragmentColor.w * tex.w;
        } else {
        const float alphaBorder1 = smoothstepf(aaf, 0.0f, varFraction);
        const float alphaBorder2 = smoothstepf(separatorWidth - aaf * 0.5f, separatorWidth + aaf * 0.5f, varFraction);
        const float m = fmaxf(alphaBorder1, alphaBorder2);
#pragma unroll
        for (int k = 0; k < 3; k++) phong[k] = phong[k] * m;
        }
        WHITE_THRESHOLD = 0.8f; // :485-486
    }
    if (bands.rasterEpsWhite >= 0.0f) {
        // LinePassGeometryShaderTubes.glsl:1079-1087: EPSILON_OUTLINE = 0.0, EPSILON_WHITE = fwidth(ribbonPosition), no clamps.  (The
        // separator stripes above keep the ray tracer's width; helicity bands are outside SURVEY.md 8's a9.)  smoothstep(1, 1, x)
        // divides by zero: x < 1 -> -inf -> 0, x == 1 -> NaN -> 0 through fmaxf(NaN, 0) = 0: coverage 1, a hard silhouette edge.
        EPSILON_OUTLINE = 0.0f;
        EPSILON_WHITE = U.useHalos ? bands.rasterEpsWhite : 0.0f;
    }
    float coverage = U.useHalos ? 1.0f - smoothstepf(1.0f - EPSILON_OUTLINE, 1.0f, absCoords) : 1.0f;
    if (BANDS == LV_SHADE_BANDS && bands.useBand && U.useEllipticTubes) coverage = 1.0f; // ANALYTIC_ELLIPTIC_TUBE_INTERSECTIONS, :499-504
    float w;   // the white halo's blend weight: colour only
    if (FL) {
        const float e0 = WHITE_THRESHOLD - EPSILON_WHITE, e1 = WHITE_THRESHOLD + EPSILON_WHITE;
        // (e1 == e0, i.e. EPSILON_WHITE == 0, keeps the exact form: its 0 / 0 and x / 0 cases are what the shader relies on)
        const float tw = e1 > e0 ? clampf(lv_div_fast(absCoords - e0, e1 - e0), 0.0f, 1.0f) : clampf((absCoords - e0) / (e1 - e0), 0.0f, 1.0f);
        w = tw * tw * (3.0f - 2.0f * tw);
    } else {
        w = smoothstepf(WHITE_THRESHOLD - EPSILON_WHITE, WHITE_THRESHOLD + EPSILON_WHITE, absCoords);
    }
    f4 out;
    out.x = mixf(phong[0], U.foreground[0], w);
    out.y = mixf(phong[1], U.foreground[1], w);
    out.z = mixf(phong[2], U.foreground[2], w);
    out.w = fragmentColor.w * coverage;
    payloadHitT = len3(fragPos - cam);
    return out;
}

__device__ __forceinline__ uint32_t lv_unorm8(float c) { return uint32_t(floorf(clampf(c, 0.0f, 1.0f) * 255.0f + 0.5f)); }
__device__ __forceinline__ uint32_t lv_pack_unorm4x8(f4 c) {
    return lv_unorm8(c.x) | (lv_unorm8(c.y) << 8) | (lv_unorm8(c.z) << 16) | (lv_unorm8(c.w) << 24);
}
__device__ __forceinline__ f4 lv_unpack_unorm4x8(uint32_t p) {
    f4 c;
    c.x = float(p & 0xFFu) / 255.0f;
    c.y = float((p >> 8) & 0xFFu) / 255.0f;
    c.z = float((p >> 16) & 0xFFu) / 255.0f;
    c.w = float((p >> 24) & 0xFFu) / 255.0f;
    return c;
}

// RayGen's final store (TubeRayTracing.glsl:268-274): with multi-frame accumulation the running mean round-trips through
// the rgba8 image: out = pack(mix(unpack(previous), colour, 1 / (frameNumber + 1))).
__device__ __forceinline__ uint32_t lv_store_color(const LvSceneDev& S, const LvUniforms& U, uint32_t x, uint32_t y, f4 c) {
    if (!S.accum) return lv_pack_unorm4x8(c);
    const size_t pi = size_t(y) * U.width + x;
    if (U.frameNumber != 0u) {
        const f4 prev = lv_unpack_unorm4x8(S.accum[pi]);
        const float a = 1.0f / float(U.frameNumber + 1u);
        c.x = mixf(prev.x, c.x, a); c.y = mixf(prev.y, c.y, a); c.z = mixf(prev.z, c.z, a); c.w = mixf(prev.w, c.w, a);
    }
    const uint32_t packed = lv_pack_unorm4x8(c);
    S.accum[pi] = packed;
    return packed;
}

// TiledAddress.glsl:53-85
__device__ __forceinline__ uint32_t lv_ppll_addr(uint32_t x, uint32_t y, uint32_t paddedW, uint32_t tileW, uint32_t tileH) {
    if (tileW == 1 && tileH == 1) return x + paddedW * y;
    uint32_t surfaceWidth = paddedW / tileW;
    uint32_t tileAddr1D = (x / tileW + surfaceWidth * (y / tileH)) * (tileW * tileH);
    uint32_t pixelAddr1D = (x & (tileW - 1)) + (y & (tileH - 1)) * tileW;
    return tileAddr1D | pixelAddr1D;
}
