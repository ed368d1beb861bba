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
// taken to the pixel.  The rasteriser itself (fixed function in the reference, unobservable) is defined by the build as a rasteriser
// in the space of the pixel's viewing ray (float32, -ffp-contract=off, fused multiply-adds exactly where written as fmaf; the CPU
// checker of the test-suite restates the same operations in the same order, an independent float64 screen-space rasteriser in
// tests/test_prism_raster.py confirms fragments, weights and colours):
//   viewing ray     the pixel-centre ray of the ray generator (TubeRayTracing.glsl:219-226: what gl_FragCoord = pixel + 0.5
//                   unprojects to).  COVERAGE is decided with its direction before the normalisation, written as the affine function of
//                   the pixel it is (round 6; the edge functions are homogeneous in the direction, and the generator's two divisions
//                   and its normalisation were a quarter of the coverage stage): target = invProj (ndc.x, ndc.y, 1, 1), dir = invView
//                   (target.xyz, 0), ndc.x = 2 (x + 0.5) / W - 1  =>  D = C0 + (x + 0.5) Cx + (y + 0.5) Cy with a = invView3 invProj[:,0].xyz,
//                   b = invView3 invProj[:,1].xyz, c = invView3 (invProj[:,2] + invProj[:,3]).xyz, Cx = a (2 / W), Cy = b (2 / H), C0 =
//                   (c - a) - b (float32 on the host, lv_prism_cov_constants), D.k = fma(x + 0.5, Cx.k, fma(y + 0.5, Cy.k, C0.k)).
//                   P = cross(R, D), Q = cross(D, P) with R = the camera's right axis: P, Q, D are mutually orthogonal, (A . P, A . Q)
//                   are -- up to positive factors -- the coordinates of A in the plane perpendicular to the ray.  Weights, depth and
//                   the acceptance rules of a covered triangle's fragment use the generator's normalised ray as before.
//   ring vertex     V = c + r (n cos + b sin) of a line point (c, n, b = cross(t, n)):  x = fma(r, fma(b . P, sin, (n . P) cos), (c - o) . P),
//                   y likewise with Q (fused dot products): six dot products per (ray, line point), two fmas per coordinate -- a pure
//                   function of (line point, circle index, ray): both triangles at an edge, both segments at a point see the same bits
//   edge function   E(U, V) = yU * xV - xU * yV = -(d . (U x V)) up to a positive factor: the homogeneous edge function of the
//                   projected edge, UNFUSED so that E(V, U) == -E(U, V) bit for bit
//   coverage        triangle (V0, V1, V2) covers the pixel iff e0 = E(V1, V2), e1 = E(V2, V0), e2 = E(V0, V1) are all > 0, an edge with
//                   e == 0 counting as inside iff the triangle OWNS it (directed edge U -> V owned iff gl_VertexIndex(U) <
//                   gl_VertexIndex(V): the neighbour runs through it the other way, exactly one of the two owns it = the fill rule),
//                   and (e0 + e1) + e2 > 0.  "All e >= 0" is at once "the ray passes through the triangle in front of the camera"
//                   and "front-facing" (det[V0 - o, V1 - o, V2 - o] < 0: the outward side of the index pattern)
//   weights         perspective-correct = barycentric coordinates of the point where a ray meets the triangle's plane: b_i = e_i *
//                   (1 / ((e0 + e1) + e2)), e_i = det[V_j - o, V_k - o, ray direction] evaluated in the own ray's basis (lv_prism_planes:
//                   well conditioned for thin tubes); attribute = (b0 a0 + b1 a1) + b2 a2
//   kept iff        depth = |fragmentPositionWorld - o| in the ray interval, within r / |d| of the interval in which the ray meets the
//                   segment's box of TubeAabbRenderData (LineDataFlow.cpp:2223-2234; every ring vertex lies within r of its line point,
//                   so this only makes the result independent of which conservative BVH supplied the candidates), and
//                   near <= -(viewMatrix * fragmentPositionWorld).z <= far (depth clipping)
//   helper lanes    fwidth(ribbonPosition) (LinePassGeometryShaderTubes.glsl:1079-1087) over the 2 x 2 quad: the quad partners evaluate
//                   the SAME triangle's attribute planes at their own pixel centre (weights of their own ray, outside [0, 1] if need
//                   be), then the shader's ribbonPosition of the interpolated inputs
// Pipeline (each stage at full wave width on its own queue): A capsule pre-test (32 B per candidate) -> B coverage mask of the 2 N
// triangles (64 B more) [both inside the all-hits walk, k_ppll_gather<LV_PRIM_PRISM>, which also gives every covered triangle its
// node and links it: the node's first two words carry {pixel, leaf | triangle << 26} through HBM] -> C fragment stage, one lane per
// node (k_ppll_shade_prism, lv_shade_prism) incl. the three `kept` rules, replacing the two words by {colour, depth}.
#pragma once

#include "lv_device.h"

// coverage direction of pixel (x, y) (header: "viewing ray")
__device__ __forceinline__ f3 lv_prism_cov_dir(const LvPrismDev& R, uint32_t x, uint32_t y) {
    const float fx = float(x) + 0.5f, fy = float(y) + 0.5f;
    return mk3(__builtin_fmaf(fx, R.covCx[0], __builtin_fmaf(fy, R.covCy[0], R.covC0[0])),
               __builtin_fmaf(fx, R.covCx[1], __builtin_fmaf(fy, R.covCy[1], R.covC0[1])),
               __builtin_fmaf(fx, R.covCx[2], __builtin_fmaf(fy, R.covCy[2], R.covC0[2])));
}
__device__ __forceinline__ void lv_prism_basis(const LvPrismDev& R, f3 d, f3& P, f3& Q) {
    P = cross3(mk3(R.right[0], R.right[1], R.right[2]), d);
    Q = cross3(d, P);
}
__device__ __forceinline__ float lv_prism_edge(float xU, float yU, float xV, float yV) { return yU * xV - xU * yV; }
__device__ __forceinline__ bool lv_prism_inside(float e, bool owned) { return e > 0.0f || (e == 0.0f && owned); }

// frame of a line point as the vertex stage uses it
struct LvPrismPoint { f3 centre, normal, binormal, tangent; float attr; uint32_t start; };
// the two line points of a leaf's segment from the leaf-ordered copies: S.segs {position, attribute} x 2 and S.prismFrames
// {tangent, point index}{normal, lineStartIndex} x 2 (k_leaves) = the 48-B records of both points in 96 contiguous bytes
__device__ __forceinline__ void lv_prism_frames(const LvSceneDev& S, uint32_t leaf, float4 pa, float4 pb, LvPrismPoint pt[2], uint32_t pi[2]) {
    const float4* q = S.prismFrames + 4 * size_t(leaf);
    const float4 t0 = q[0], n0 = q[1], t1 = q[2], n1 = q[3];
    pt[0].centre = mk3(pa.x, pa.y, pa.z); pt[0].attr = pa.w;
    pt[0].tangent = mk3(t0.x, t0.y, t0.z); pi[0] = __float_as_uint(t0.w);
    pt[0].normal = mk3(n0.x, n0.y, n0.z); pt[0].start = __float_as_uint(n0.w);
    pt[0].binormal = cross3(pt[0].tangent, pt[0].normal);
    pt[1].centre = mk3(pb.x, pb.y, pb.z); pt[1].attr = pb.w;
    pt[1].tangent = mk3(t1.x, t1.y, t1.z); pi[1] = __float_as_uint(t1.w);
    pt[1].normal = mk3(n1.x, n1.y, n1.z); pt[1].start = __float_as_uint(n1.w);
    pt[1].binormal = cross3(pt[1].tangent, pt[1].normal);
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
__device__ __forceinline__ float lv_prism_dot(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
// projection of a line point's frame into the ray's plane (same operations as the CPU checker): six dot products per (ray, line point), then
// two fused multiply-adds per coordinate of a ring vertex
struct LvPrismProj { float X0, Y0, nP, bP, nQ, bQ; };
__device__ __forceinline__ LvPrismProj lv_prism_point_proj(const LvPrismPoint& p, f3 o, f3 P, f3 Q) {
    const f3 C = p.centre - o;
    LvPrismProj pj;
    pj.X0 = lv_prism_dot(C, P); pj.Y0 = lv_prism_dot(C, Q);
    pj.nP = lv_prism_dot(p.normal, P); pj.bP = lv_prism_dot(p.binormal, P);
    pj.nQ = lv_prism_dot(p.normal, Q); pj.bQ = lv_prism_dot(p.binormal, Q);
    return pj;
}
__device__ __forceinline__ void lv_prism_vertex_xy(const LvPrismProj& pj, float c, float s, float radius, float& x, float& y) {
    const float u = __builtin_fmaf(pj.bP, s, pj.nP * c), v = __builtin_fmaf(pj.bQ, s, pj.nQ * c);
    x = __builtin_fmaf(radius, u, pj.X0);
    y = __builtin_fmaf(radius, v, pj.Y0);
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

// one triangle of one segment as the fragment stage sees it.  `ring`: cos at [k], sin at [LV_PRISM_MAX_SUBDIV + k] (the kernel's LDS
// copy of LvPrismDev's table: triangle indices are per-lane values)
struct LvPrismTri {
    f3 pos[3], dir[3], tan[3];
    float attr[3], c[3], s[3];   // c, s: the POSITION's ring coefficients (cp, s)
    uint32_t id[3], circ[3];
    bool second[3];   // vertex belongs to the segment's second point
};
// ringTab: the kernel's LDS copy of LvPrismDev's tables, [c | s | cp | sn]
__device__ __forceinline__ LvPrismTri lv_prism_tri_setup(const float* ringTab, uint32_t N, const LvPrismPoint pt[2], const uint32_t pi[2],
                                                         float radius, uint32_t tt) {
    uint32_t ring[3], circ[3];
    lv_prism_triangle(tt, N, ring, circ);
    LvPrismTri T;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const LvPrismPoint& p = ring[i] ? pt[1] : pt[0];
        T.c[i] = ringTab[2 * LV_PRISM_MAX_SUBDIV + circ[i]];
        T.s[i] = ringTab[LV_PRISM_MAX_SUBDIV + circ[i]];
        T.dir[i] = lv_prism_dir(p, ringTab[circ[i]], ringTab[3 * LV_PRISM_MAX_SUBDIV + circ[i]]);   // localNormal
        T.pos[i] = lv_prism_pos(p, lv_prism_dir(p, T.c[i], T.s[i]), radius);                          // localPosition
        T.tan[i] = p.tangent;
        T.attr[i] = p.attr;
        T.id[i] = (ring[i] ? pi[1] : pi[0]) * N + circ[i];
        T.second[i] = ring[i] != 0u;
        T.circ[i] = circ[i];
    }
    return T;
}
// edge functions of a triangle for the ray (o, D) (any length); returns the coverage decision
__device__ __forceinline__ bool lv_prism_tri_edges(const LvPrismDev& R, const LvPrismPoint pt[2], const LvPrismTri& T, float radius, f3 o,
                                                   f3 D, float e[3]) {
    f3 P, Q;
    lv_prism_basis(R, D, P, Q);
    const LvPrismProj pj0 = lv_prism_point_proj(pt[0], o, P, Q), pj1 = lv_prism_point_proj(pt[1], o, P, Q);
    float x[3], y[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float x0, y0, x1, y1;
        lv_prism_vertex_xy(pj0, T.c[i], T.s[i], radius, x0, y0);
        lv_prism_vertex_xy(pj1, T.c[i], T.s[i], radius, x1, y1);
        x[i] = T.second[i] ? x1 : x0;
        y[i] = T.second[i] ? y1 : y0;
    }
    e[0] = lv_prism_edge(x[1], y[1], x[2], y[2]);
    e[1] = lv_prism_edge(x[2], y[2], x[0], y[0]);
    e[2] = lv_prism_edge(x[0], y[0], x[1], y[1]);
    const bool in = lv_prism_inside(e[0], T.id[1] < T.id[2]) && lv_prism_inside(e[1], T.id[2] < T.id[0]) &&
                    lv_prism_inside(e[2], T.id[0] < T.id[1]);
    return in && (e[0] + e[1]) + e[2] > 0.0f;
}
template <bool FASTN = false>
__device__ __forceinline__ void lv_prism_weights(const float e[3], float b[3]) {
    const float rs = FASTN ? __builtin_amdgcn_rcpf((e[0] + e[1]) + e[2]) : 1.0f / ((e[0] + e[1]) + e[2]);
    b[0] = e[0] * rs; b[1] = e[1] * rs; b[2] = e[2] * rs;
}
__device__ __forceinline__ f3 lv_prism_mix3(const float b[3], f3 a0, f3 a1, f3 a2) { return (b[0] * a0 + b[1] * a1) + b[2] * a2; }

// Acceptance of a covered triangle's fragment (evaluated by the shading stage, which has the interpolated position anyway): depth in
// the ray's interval [tLo, tHi), own-box rule, depth clipping.
__device__ __forceinline__ bool lv_prism_accept(const LvPrismDev& R, const LvPrismPoint pt[2], float radius, f3 o, f3 d, f3 pos,
                                                float depth, float tLo, float tHi) {
    if (!(depth >= tLo && depth < tHi)) return false;
    if (!lv_prism_own_box(o, d, pt[0].centre, pt[1].centre, radius, depth)) return false;
    const float vz = ((R.viewZ[0] * pos.x + R.viewZ[1] * pos.y) + R.viewZ[2] * pos.z) + R.viewZ[3] * 1.0f;
    return -vz >= R.nearDist && -vz <= R.farDist;
}

// Stage A of a (viewing ray, segment) test: conservative pre-tests of the capsule that contains the prism (they may only say
// "cannot hit"); 32 B per candidate.  Only candidates that pass are queued for the coverage test.
__device__ __forceinline__ bool lv_prism_pretest(const LvSceneDev& S, float radius, uint32_t leaf, f3 o, f3 d) {
    const float4 pa = S.segs[2 * size_t(leaf)], pb = S.segs[2 * size_t(leaf) + 1];
    return lv_capsule_may_hit_axis(o, d, mk3(pa.x, pa.y, pa.z), mk3(pb.x, pb.y, pb.z), radius) &&
           lv_capsule_may_hit_sphere(o, d, mk3(pa.x, pa.y, pa.z), mk3(pb.x, pb.y, pb.z), radius);
}

// Stage B: coverage of the pixel by the 2 N triangles of the segment's prism -- bit tt of the result = triangle tt covers the
// pixel (front face, fill rule); whether its fragment is kept (ray interval, own box, depth clipping) is decided where it is shaded.
// NT > 0: N = NT known at compile time (ring in registers, everything unrolled); NT == 0: any N <= LV_PRISM_MAX_SUBDIV.
template <int NT>
__device__ __forceinline__ unsigned lv_prism_coverage_pts(const LvPrismDev& R, const LvPrismPoint pt[2], const uint32_t pi[2], float radius,
                                                          f3 o, f3 d) {
    f3 P, Q;
    lv_prism_basis(R, d, P, Q);
    const uint32_t N = NT > 0 ? uint32_t(NT) : R.n;
    const bool a01 = pi[0] < pi[1];       // point ids of the usual segment (i, i + 1)
    const LvPrismProj pj0 = lv_prism_point_proj(pt[0], o, P, Q), pj1 = lv_prism_point_proj(pt[1], o, P, Q);
    unsigned mask = 0u;
    if (NT > 0) {
        float cx[NT > 0 ? NT : 1], cy[NT > 0 ? NT : 1], nx[NT > 0 ? NT : 1], ny[NT > 0 ? NT : 1], L[NT > 0 ? NT : 1];
#pragma unroll
        for (int k = 0; k < NT; k++) {
            lv_prism_vertex_xy(pj0, R.cp[k], R.s[k], radius, cx[k], cy[k]);
            lv_prism_vertex_xy(pj1, R.cp[k], R.s[k], radius, nx[k], ny[k]);
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
            lv_prism_vertex_xy(pj0, R.cp[k], R.s[k], radius, cxk, cyk);
            lv_prism_vertex_xy(pj0, R.cp[kn], R.s[kn], radius, cxn, cyn);
            lv_prism_vertex_xy(pj1, R.cp[k], R.s[k], radius, nxk, nyk);
            lv_prism_vertex_xy(pj1, R.cp[kn], R.s[kn], radius, nxn, nyn);
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
    return mask;
}
template <int NT>
__device__ __forceinline__ unsigned lv_prism_coverage(const LvSceneDev& S, float radius, uint32_t leaf, f3 o, f3 d) {
    const float4 pa = S.segs[2 * size_t(leaf)], pb = S.segs[2 * size_t(leaf) + 1];
    uint32_t pi[2];
    LvPrismPoint pt[2];
    lv_prism_frames(S, leaf, pa, pb, pt, pi);
    return lv_prism_coverage_pts<NT>(S.prism, pt, pi, radius, o, d);
}

// the raster shader's ribbonPosition of interpolated inputs (no bands, no caps), LinePassGeometryShaderTubes.glsl:771-777,944-963
template <bool FASTN = false>
__device__ __forceinline__ float lv_prism_ribbon(f3 cam, f3 fragPos, f3 fragmentNormal, f3 fragmentTangent) {
    const f3 n = norm3q<FASTN>(fragmentNormal);
    const f3 v = norm3q<FASTN>(cam - fragPos);
    const f3 t = norm3q<FASTN>(fragmentTangent);
    const f3 helperVec = norm3q<FASTN>(cross3(t, v));
    const f3 newV = norm3q<FASTN>(cross3(helperVec, t));
    const f3 crossProdVn = cross3(newV, n);
    float ribbonPosition = FASTN ? __builtin_amdgcn_sqrtf(dot3(crossProdVn, crossProdVn)) : len3(crossProdVn);
    if (dot3(t, crossProdVn) < 0.0f) ribbonPosition = -ribbonPosition;
    return clampf(ribbonPosition, -1.0f, 1.0f);
}

// Perspective-correct weights of a ray direction in a triangle seen from o : e_i = det[V_j - o,
// V_k - o, ray direction] in the coordinates of the pixel's own ray basis (P, Q, d), where the triangle's X, Y are small and the
// 2 x 2 minors c_i = (X, Y, Z)_j x (X, Y, Z)_k stay well conditioned; computed once per triangle, then six fused dot products per ray
// -- the fragment's own ray and its two helper lanes.
struct LvPrismPlanes { f3 P, Q, D, c0, c1, c2; };
__device__ __forceinline__ LvPrismPlanes lv_prism_planes(const LvPrismDev& R, const LvPrismTri& T, f3 o, f3 d) {
    LvPrismPlanes pl;
    lv_prism_basis(R, d, pl.P, pl.Q);
    pl.D = d;
    f3 v[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const f3 A = T.pos[i] - o;
        v[i] = mk3(lv_prism_dot(A, pl.P), lv_prism_dot(A, pl.Q), lv_prism_dot(A, d));
    }
    pl.c0 = cross3(v[1], v[2]); pl.c1 = cross3(v[2], v[0]); pl.c2 = cross3(v[0], v[1]);
    return pl;
}
// interpolated inputs of the fragment stage for the ray direction D (helper invocations: weights outside [0, 1])
struct LvPrismInputs { f3 pos, nrm, tan; float attr; float b[3]; };
template <bool FASTN = false>   // FASTN: the quad partners' inputs (they only feed fwidth of the halo coordinate)
__device__ __forceinline__ LvPrismInputs lv_prism_interpolate(const LvPrismTri& T, const f3 nrm[3], const LvPrismPlanes& pl, f3 Dr) {
    const f3 abg = mk3(lv_prism_dot(Dr, pl.P), lv_prism_dot(Dr, pl.Q), lv_prism_dot(Dr, pl.D));
    const float e[3] = {lv_prism_dot(abg, pl.c0), lv_prism_dot(abg, pl.c1), lv_prism_dot(abg, pl.c2)};
    LvPrismInputs I;
    float* b = I.b;
    lv_prism_weights<FASTN>(e, b);
    I.pos = lv_prism_mix3(b, T.pos[0], T.pos[1], T.pos[2]);
    I.nrm = lv_prism_mix3(b, nrm[0], nrm[1], nrm[2]);
    I.tan = lv_prism_mix3(b, T.tan[0], T.tan[1], T.tan[2]);
    I.attr = (b[0] * T.attr[0] + b[1] * T.attr[1]) + b[2] * T.attr[2];
    return I;
}

// STATIC_AMBIENT_OCCLUSION_PREBAKING in the raster shaders: the inputs of getAoFactor(fragmentVertexId, phi) (Lighting.glsl:124-125) --
// vertex stage LinePassProgrammablePullTubes.glsl:179-206 (phi = circleIdx * (2 pi / N); interpolateWrap and fragmentVertexIdUint =
// lineStartIndex are FLAT = the provoking (first) vertex's; interpolationFactorLine = float(linePointIdx - lineStartIndex)), fragment
// stage LinePassGeometryShaderTubes.glsl:753-770 (fragmentVertexId, the wrap-around of phi on the last facet)
__device__ __forceinline__ void lv_prism_ao_inputs(const LvPrismTri& T, const LvPrismPoint pt[2], const uint32_t pi[2], const float b[3],
                                                   uint32_t N, float& fragmentVertexId, float& phi) {
    const float PI = 3.14159265358979323846f;
    float li[3];
#pragma unroll
    for (int i = 0; i < 3; i++) li[i] = float((T.second[i] ? pi[1] : pi[0]) - (T.second[i] ? pt[1].start : pt[0].start));
    const float interpolationFactorLine = (b[0] * li[0] + b[1] * li[1]) + b[2] * li[2];
    fragmentVertexId = interpolationFactorLine + float(T.second[0] ? pt[1].start : pt[0].start);
    const float factor = 2.0f * PI / float(N);
    const float phiNotWrapInterpolated = (b[0] * (float(T.circ[0]) * factor) + b[1] * (float(T.circ[1]) * factor)) + b[2] * (float(T.circ[2]) * factor);
    if (T.circ[0] != N - 1u) {
        phi = phiNotWrapInterpolated;
    } else {
        const float lower = 2.0f * PI * float(N - 1u) / float(N);
        const float upper = 2.0f * PI;
        phi = lower + (phiNotWrapInterpolated - lower) / (-lower) * (upper - lower);
    }
}
