// lv_prism.h -- PPLL fragments from the geometry the reference RASTERISES (ppll_fragment_source = raster_prism).
//
// The reference's PPLL gather rasterises, in the default "Tube (Programmable Pull)" mode, an UNCAPPED N-gon prism per segment:
//   vertex stage   Data/Shaders/Renderers/GeometryPass/LinePassProgrammablePullTubes.glsl:87-224 -- ring vertex circleIdx of line
//                  point linePointIdx in that point's own (normal, binormal, tangent) frame (:123-177)
//   index pattern  src/LineData/LineDataFlow.cpp:1698-1713 -- per segment and k < N: (c_k, c_kn, n_k), (n_k, c_kn, n_kn)
//   culling        back faces (src/Renderers/LineRasterPass.cpp:85-96)
//   fragment stage LinePassGeometryShaderTubes.glsl:732-1129 on perspective-correct interpolated position / normal / tangent /
//                  attribute; node depth = length(fragmentPositionWorld - cameraPosition) (LinkedListGather.glsl:47)
//
// MI355X form: no triangle mesh is ever built (12 M triangles = 0.8 GB of records + a second LBVH for config 4).  The SEGMENT LBVH of
// the ray tracer supplies candidate segments (every ring vertex lies within r of its line point, i.e. inside the segment's box); a
// lane then pulls the two 48-B line points, generates the 2 N ring vertices in registers, projects them into the plane
// perpendicular to its pixel's viewing ray and evaluates the 4 N distinct edge functions of the 2 N triangles -- "programmable pull"
// taken to the pixel.  The rasteriser itself (fixed function in the reference, unobservable) is defined by the build; the
// definition, with the reasons for every choice, is the header of oracle/lv_oracle_prism.h, which this file mirrors operation for
// operation (float32, -ffp-contract=off, fused multiply-adds exactly where written as fmaf):
//   P = cross(R, d), Q = cross(d, P)                        R = camera right axis
//   (x, y) = ((V - o) . P, (V - o) . Q)                     fused dot products
//   E(U, V) = yU * xV - xU * yV                             unfused: E(V, U) == -E(U, V) bit for bit (shared edges)
//   covered iff every e_i > 0, or == 0 on an edge the triangle owns (gl_VertexIndex(U) < gl_VertexIndex(V)), and sum > 0
//   weights b_i = e_i * (1 / ((e0 + e1) + e2)); attribute = (b0 a0 + b1 a1) + b2 a2
//   kept iff depth in the slice [tLo, tHi), within r / |d| of the segment's box interval, and near <= -view.z <= far
#pragma once

#include "lv_device.h"

__device__ __forceinline__ void lv_prism_basis(const LvPrismDev& R, f3 d, f3& P, f3& Q) {
    P = cross3(mk3(R.right[0], R.right[1], R.right[2]), d);
    Q = cross3(d, P);
}
__device__ __forceinline__ float lv_prism_edge(float xU, float yU, float xV, float yV) { return yU * xV - xU * yV; }
__device__ __forceinline__ bool lv_prism_inside(float e, bool owned) { return e > 0.0f || (e == 0.0f && owned); }

// frame of a line point as the vertex stage uses it
struct LvPrismPoint { f3 centre, normal, binormal, tangent; float attr; uint32_t start; };
__device__ __forceinline__ LvPrismPoint lv_prism_point(const lv_line_point* __restrict__ points, uint32_t idx) {
    const float4* q = (const float4*)(points + idx);    // 48-B records: three 16-B loads
    const float4 a = q[0], b = q[1], c = q[2];
    LvPrismPoint p;
    p.centre = mk3(a.x, a.y, a.z); p.attr = a.w;
    p.tangent = mk3(b.x, b.y, b.z);
    p.normal = mk3(c.x, c.y, c.z); p.start = __float_as_uint(c.w);
    p.binormal = cross3(p.tangent, p.normal);
    return p;
}
// ring vertex: dir = normal * cos + binormal * sin; position = radius * dir + centre
__device__ __forceinline__ f3 lv_prism_dir(const LvPrismPoint& p, float c, float s) {
    return mk3(__builtin_fmaf(p.binormal.x, s, p.normal.x * c), __builtin_fmaf(p.binormal.y, s, p.normal.y * c),
               __builtin_fmaf(p.binormal.z, s, p.normal.z * c));
}
__device__ __forceinline__ f3 lv_prism_pos(const LvPrismPoint& p, f3 dir, float radius) {
    return mk3(__builtin_fmaf(radius, dir.x, p.centre.x), __builtin_fmaf(radius, dir.y, p.centre.y),
               __builtin_fmaf(radius, dir.z, p.centre.z));
}
__device__ __forceinline__ void lv_prism_project(f3 pos, f3 o, f3 P, f3 Q, float& x, float& y) {
    const f3 A = pos - o;
    x = __builtin_fmaf(A.z, P.z, __builtin_fmaf(A.y, P.y, A.x * P.x));
    y = __builtin_fmaf(A.z, Q.z, __builtin_fmaf(A.y, Q.y, A.x * Q.x));
}
__device__ __forceinline__ void lv_prism_vertex_xy(const LvPrismPoint& p, float c, float s, float radius, f3 o, f3 P, f3 Q, float& x,
                                                   float& y) {
    lv_prism_project(lv_prism_pos(p, lv_prism_dir(p, c, s), radius), o, P, Q, x, y);
}

// triangle tt < 2 N of a segment's prism: (ring 0 = first point / 1 = second point, circle index) of its three vertices
__device__ __forceinline__ void lv_prism_triangle(uint32_t tt, uint32_t N, uint32_t ring[3], uint32_t circ[3]) {
    const uint32_t k = tt >> 1, kn = (k + 1u == N) ? 0u : k + 1u;
    if ((tt & 1u) == 0u) { ring[0] = 0; circ[0] = k; ring[1] = 0; circ[1] = kn; ring[2] = 1; circ[2] = k; }
    else { ring[0] = 1; circ[0] = k; ring[1] = 0; circ[1] = kn; ring[2] = 1; circ[2] = kn; }
}

// own-box rule (lv_intersect_capsule_literal's): the ray meets the segment's box of TubeAabbRenderData, depth within r / |d| of it
__device__ __forceinline__ bool lv_prism_own_box(f3 o, f3 d, f3 p0, f3 p1, float radius, float depth) {
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const float tx0 = ((fminf(p0.x, p1.x) - radius) - o.x) * inv.x, tx1 = ((fmaxf(p0.x, p1.x) + radius) - o.x) * inv.x;
    const float ty0 = ((fminf(p0.y, p1.y) - radius) - o.y) * inv.y, ty1 = ((fmaxf(p0.y, p1.y) + radius) - o.y) * inv.y;
    const float tz0 = ((fminf(p0.z, p1.z) - radius) - o.z) * inv.z, tz1 = ((fmaxf(p0.z, p1.z) + radius) - o.z) * inv.z;
    const float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fminf(tz0, tz1));
    const float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
    const float slack = radius / len3(d);
    return tn <= tf && depth >= tn - slack && depth <= tf + slack;
}

// one triangle of one segment as the fragment stage sees it
struct LvPrismTri {
    f3 pos[3], dir[3], tan[3];
    float attr[3];
    uint32_t id[3];
};
__device__ __forceinline__ LvPrismTri lv_prism_tri_setup(const LvPrismDev& R, const LvPrismPoint pt[2], const uint32_t pi[2],
                                                         float radius, uint32_t tt) {
    uint32_t ring[3], circ[3];
    lv_prism_triangle(tt, R.n, ring, circ);
    LvPrismTri T;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const LvPrismPoint& p = ring[i] ? pt[1] : pt[0];
        T.dir[i] = lv_prism_dir(p, R.c[circ[i]], R.s[circ[i]]);
        T.pos[i] = lv_prism_pos(p, T.dir[i], radius);
        T.tan[i] = p.tangent;
        T.attr[i] = p.attr;
        T.id[i] = (ring[i] ? pi[1] : pi[0]) * R.n + circ[i];
    }
    return T;
}
// edge functions of a triangle for the ray (o, D) (any length); returns the coverage decision
__device__ __forceinline__ bool lv_prism_tri_edges(const LvPrismDev& R, const LvPrismTri& T, f3 o, f3 D, float e[3]) {
    f3 P, Q;
    lv_prism_basis(R, D, P, Q);
    float x[3], y[3];
#pragma unroll
    for (int i = 0; i < 3; i++) lv_prism_project(T.pos[i], o, P, Q, x[i], y[i]);
    e[0] = lv_prism_edge(x[1], y[1], x[2], y[2]);
    e[1] = lv_prism_edge(x[2], y[2], x[0], y[0]);
    e[2] = lv_prism_edge(x[0], y[0], x[1], y[1]);
    const bool in = lv_prism_inside(e[0], T.id[1] < T.id[2]) && lv_prism_inside(e[1], T.id[2] < T.id[0]) &&
                    lv_prism_inside(e[2], T.id[0] < T.id[1]);
    return in && (e[0] + e[1]) + e[2] > 0.0f;
}
__device__ __forceinline__ void lv_prism_weights(const float e[3], float b[3]) {
    const float rs = 1.0f / ((e[0] + e[1]) + e[2]);
    b[0] = e[0] * rs; b[1] = e[1] * rs; b[2] = e[2] * rs;
}
__device__ __forceinline__ f3 lv_prism_mix3(const float b[3], f3 a0, f3 a1, f3 a2) { return (b[0] * a0 + b[1] * a1) + b[2] * a2; }

// Acceptance of a covered triangle: depth in [tLo, tHi), own-box rule, depth clipping.  Recomputes the triangle (same operations
// as the coverage pass -> same bits).
__device__ __forceinline__ bool lv_prism_accept(const LvPrismDev& R, const LvPrismPoint pt[2], const uint32_t pi[2], float radius,
                                                uint32_t tt, f3 o, f3 d, float tLo, float tHi) {
    const LvPrismTri T = lv_prism_tri_setup(R, pt, pi, radius, tt);
    float e[3], b[3];
    if (!lv_prism_tri_edges(R, T, o, d, e)) return false; // (always true here: the coverage pass said so)
    lv_prism_weights(e, b);
    const f3 pos = lv_prism_mix3(b, T.pos[0], T.pos[1], T.pos[2]);
    const float depth = len3(pos - o);
    if (!(depth >= tLo && depth < tHi)) return false;
    if (!lv_prism_own_box(o, d, pt[0].centre, pt[1].centre, radius, depth)) return false;
    const float vz = ((R.viewZ[0] * pos.x + R.viewZ[1] * pos.y) + R.viewZ[2] * pos.z) + R.viewZ[3] * 1.0f;
    return -vz >= R.nearDist && -vz <= R.farDist;
}

// One (viewing ray, segment) test: bit tt of the result = triangle tt of the segment's prism yields a fragment for this ray.
// NT > 0: N = NT known at compile time (ring in registers, everything unrolled); NT == 0: any N <= LV_PRISM_MAX_SUBDIV.
template <int NT>
__device__ __forceinline__ unsigned lv_prism_test(const LvSceneDev& S, float radius, uint32_t leaf, f3 o, f3 d, float tLo, float tHi) {
    const LvPrismDev& R = S.prism;
    const uint32_t seg = S.leafSeg[leaf];
    const uint32_t pi[2] = {S.segIdx[2 * seg], S.segIdx[2 * seg + 1]};
    LvPrismPoint pt[2];
    pt[0] = lv_prism_point(S.points, pi[0]);
    pt[1] = lv_prism_point(S.points, pi[1]);
    // conservative pre-tests of the capsule that contains the prism (they may only say "cannot hit")
    if (!lv_capsule_may_hit_axis(o, d, pt[0].centre, pt[1].centre, radius)) return 0u;
    if (!lv_capsule_may_hit_sphere(o, d, pt[0].centre, pt[1].centre, radius)) return 0u;
    f3 P, Q;
    lv_prism_basis(R, d, P, Q);
    const uint32_t N = NT > 0 ? uint32_t(NT) : R.n;
    const bool a01 = pi[0] < pi[1];       // point ids of the usual segment (i, i + 1)
    unsigned mask = 0u;
    if (NT > 0) {
        float cx[NT > 0 ? NT : 1], cy[NT > 0 ? NT : 1], nx[NT > 0 ? NT : 1], ny[NT > 0 ? NT : 1], L[NT > 0 ? NT : 1];
#pragma unroll
        for (int k = 0; k < NT; k++) {
            lv_prism_vertex_xy(pt[0], R.c[k], R.s[k], radius, o, P, Q, cx[k], cy[k]);
            lv_prism_vertex_xy(pt[1], R.c[k], R.s[k], radius, o, P, Q, nx[k], ny[k]);
            L[k] = lv_prism_edge(nx[k], ny[k], cx[k], cy[k]);           // E(n_k, c_k)
        }
#pragma unroll
        for (int k = 0; k < NT; k++) {
            const int kn = (k + 1 == NT) ? 0 : k + 1;
            const float D = lv_prism_edge(cx[kn], cy[kn], nx[k], ny[k]);    // E(c_kn, n_k)
            const float R0 = lv_prism_edge(cx[k], cy[k], cx[kn], cy[kn]);   // E(c_k, c_kn)
            const float R1 = lv_prism_edge(nx[kn], ny[kn], nx[k], ny[k]);   // E(n_kn, n_k)
            const bool dOwn = pi[0] * N + uint32_t(kn) < pi[1] * N + uint32_t(k);  // id(c_kn) < id(n_k)
            // even (c_k, c_kn, n_k): e0 = E(c_kn, n_k), e1 = E(n_k, c_k), e2 = E(c_k, c_kn)
            {
                const bool in = lv_prism_inside(D, dOwn) && lv_prism_inside(L[k], !a01 && pi[1] != pi[0]) && lv_prism_inside(R0, kn != 0);
                if (in && (D + L[k]) + R0 > 0.0f) mask |= 1u << (2 * k);
            }
            // odd (n_k, c_kn, n_kn): e0 = E(c_kn, n_kn) = -E(n_kn, c_kn), e1 = E(n_kn, n_k), e2 = E(n_k, c_kn) = -E(c_kn, n_k)
            {
                const float e0 = -L[kn], e2 = -D;
                const bool in = lv_prism_inside(e0, a01) && lv_prism_inside(R1, kn == 0) && lv_prism_inside(e2, !dOwn);
                if (in && (e0 + R1) + e2 > 0.0f) mask |= 1u << (2 * k + 1);
            }
        }
    } else {
        for (uint32_t k = 0; k < N; k++) {
            const uint32_t kn = (k + 1u == N) ? 0u : k + 1u;
            float cxk, cyk, cxn, cyn, nxk, nyk, nxn, nyn;
            lv_prism_vertex_xy(pt[0], R.c[k], R.s[k], radius, o, P, Q, cxk, cyk);
            lv_prism_vertex_xy(pt[0], R.c[kn], R.s[kn], radius, o, P, Q, cxn, cyn);
            lv_prism_vertex_xy(pt[1], R.c[k], R.s[k], radius, o, P, Q, nxk, nyk);
            lv_prism_vertex_xy(pt[1], R.c[kn], R.s[kn], radius, o, P, Q, nxn, nyn);
            const float Lk = lv_prism_edge(nxk, nyk, cxk, cyk), Ln = lv_prism_edge(nxn, nyn, cxn, cyn);
            const float D = lv_prism_edge(cxn, cyn, nxk, nyk);
            const float R0 = lv_prism_edge(cxk, cyk, cxn, cyn);
            const float R1 = lv_prism_edge(nxn, nyn, nxk, nyk);
            const bool dOwn = pi[0] * N + kn < pi[1] * N + k;
            {
                const bool in = lv_prism_inside(D, dOwn) && lv_prism_inside(Lk, !a01 && pi[1] != pi[0]) && lv_prism_inside(R0, kn != 0u);
                if (in && (D + Lk) + R0 > 0.0f) mask |= 1u << (2u * k);
            }
            {
                const float e0 = -Ln, e2 = -D;
                const bool in = lv_prism_inside(e0, a01) && lv_prism_inside(R1, kn == 0u) && lv_prism_inside(e2, !dOwn);
                if (in && (e0 + R1) + e2 > 0.0f) mask |= 1u << (2u * k + 1u);
            }
        }
    }
    // acceptance of the covered triangles (usually one)
    unsigned out = 0u;
    while (mask) {
        const uint32_t tt = uint32_t(__ffs(int(mask))) - 1u;
        mask &= mask - 1u;
        if (lv_prism_accept(R, pt, pi, radius, tt, o, d, tLo, tHi)) out |= 1u << tt;
    }
    return out;
}

// the raster shader's ribbonPosition of interpolated inputs (no bands, no caps), LinePassGeometryShaderTubes.glsl:771-777,944-963
__device__ __forceinline__ float lv_prism_ribbon(f3 cam, f3 fragPos, f3 fragmentNormal, f3 fragmentTangent) {
    const f3 n = norm3s(fragmentNormal);
    const f3 v = norm3s(cam - fragPos);
    const f3 t = norm3s(fragmentTangent);
    const f3 helperVec = norm3s(cross3(t, v));
    const f3 newV = norm3s(cross3(helperVec, t));
    const f3 crossProdVn = cross3(newV, n);
    float ribbonPosition = len3(crossProdVn);
    if (dot3(t, crossProdVn) < 0.0f) ribbonPosition = -ribbonPosition;
    return clampf(ribbonPosition, -1.0f, 1.0f);
}

// interpolated inputs of the fragment stage for the ray (o, D): weights of D in the triangle's planes (helper invocations: outside)
struct LvPrismInputs { f3 pos, nrm, tan; float attr; };
__device__ __forceinline__ LvPrismInputs lv_prism_interpolate(const LvPrismTri& T, const f3 nrm[3], const float e[3]) {
    float b[3];
    lv_prism_weights(e, b);
    LvPrismInputs I;
    I.pos = lv_prism_mix3(b, T.pos[0], T.pos[1], T.pos[2]);
    I.nrm = lv_prism_mix3(b, nrm[0], nrm[1], nrm[2]);
    I.tan = lv_prism_mix3(b, T.tan[0], T.tan[1], T.tan[2]);
    I.attr = (b[0] * T.attr[0] + b[1] * T.attr[1]) + b[2] * T.attr[2];
    return I;
}
