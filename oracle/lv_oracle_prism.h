// lv_oracle_prism.h -- CPU ORACLE (test infrastructure, not product), included by lv_oracle.cpp inside its anonymous namespace.
//
// PPLL fragments from the geometry the reference RASTERISES (ppll_fragment_source = raster_prism):
// the default "Tube (Programmable Pull)" primitive mode (LINE_PRIMITIVES_TUBE_PROGRAMMABLE_PULL, src/LineData/LineData.cpp:51).
//
//  * vertex stage, Data/Shaders/Renderers/GeometryPass/LinePassProgrammablePullTubes.glsl:87-224: vertex gl_VertexIndex =
//    linePointIdx * N + circleIdx is ring vertex circleIdx of line point linePointIdx in that point's OWN frame
//    (normal, binormal = cross(tangent, normal), tangent) (:123-127): position = lineRadius * (frame * (cos t, sin t, 0)) + centre,
//    vertex normal = normalize(frame * (cos t, sin t, 0)) (:174-177), t = circleIdx / N * 2 pi (:129-131);
//  * index pattern, src/LineData/LineDataFlow.cpp:1698-1713: per segment (point i -> i + 1) and k < N, kn = (k + 1) % N the triangles
//    (c_k, c_kn, n_k) and (n_k, c_kn, n_kn) with c = ring of point i, n = ring of point i + 1: an UNCAPPED N-gon prism
//    (USE_CAPPED_TUBES is not defined for the rasterisers, LineData.cpp:1240-1244);
//  * back faces culled (transparency is used: src/Renderers/LineRasterPass.cpp:85-96); the outward side is the front side (the
//    geometric normal of both triangles of the pattern points away from the axis);
//  * fragment stage, LinePassGeometryShaderTubes.glsl:732-1129, receives the perspective-correct interpolation of
//    fragmentPositionWorld / fragmentNormal / fragmentTangent / fragmentAttribute (ProgrammablePull:212-223); its depth for the list
//    node is length(fragmentPositionWorld - cameraPosition) (LinkedListGather.glsl:47);
//  * point frames: the records of getLinePassTubeAabbRenderData (LineDataFlow.cpp:2112-2277) -- getLinePassTubeRenderDataGeneral
//    (:1385-1444) computes the same tangents and Gram-Schmidt normals except for the fallback-axis test that reads an
//    uninitialised `normal` (:1419, undefined behaviour); the AABB path's test on `tangent` (:2174) is the defined version.
//
// The fixed-function rasteriser between the two stages is not observable; the build defines it (float32, fixed operation order,
// explicit fused multiply-adds where written as fmaf) as a rasteriser in the space of the pixel's viewing ray:
//
//   viewing ray      the pixel-centre ray (o, d) of the ray generator (TubeRayTracing.glsl:219-226: what gl_FragCoord = pixel + 0.5
//                    unprojects to); basis P = cross(R, d), Q = cross(d, P) with R = the camera's right axis (column 0 of
//                    inverse(viewMatrix)): P, Q, d are mutually orthogonal, so (A . P, A . Q) are -- up to positive factors -- the
//                    coordinates of a vertex A - o in the plane perpendicular to the ray; for the ring vertex V = centre + r (normal cos
//                    + binormal sin): x = fma(r, fma(binormal . P, sin, (normal . P) cos), (centre - o) . P), y likewise with Q (dot
//                    products fused: prismDot) -- a pure function of (line point, circle index, ray), so both triangles at an edge
//                    and both segments at a line point see the same bits
//   edge function    E(U, V) = yU * xV - xU * yV of two projected vertices = -(d . (U x V)) up to a positive factor: the
//                    homogeneous edge function of the projected edge (Olano & Greer 1997), evaluated WITHOUT contraction so that
//                    E(V, U) = -E(U, V) holds bit for bit -- two triangles sharing an edge see exactly opposite values
//   coverage         triangle (V0, V1, V2) covers the pixel iff e0 = E(V1, V2), e1 = E(V2, V0), e2 = E(V0, V1) are all > 0, where an
//                    edge with e == 0 counts as inside iff it is OWNED: the directed edge U -> V of the triangle's winding is owned
//                    iff gl_VertexIndex(U) < gl_VertexIndex(V) (the neighbour runs through it the other way, so exactly one of the two
//                    owns it: the fill rule), and (e0 + e1) + e2 > 0.  All e >= 0 is at the same time "the ray passes through the
//                    triangle in front of the camera" and "the triangle is front-facing" (det[V0 - o, V1 - o, V2 - o] < 0).
//   interpolation    the barycentric coordinates of the point where a ray meets the triangle's plane = the perspective-correct weights
//                    of the rasteriser: b_i = e_i * (1 / ((e0 + e1) + e2)) with e_i = det[V_j - o, V_k - o, ray direction] evaluated in the
//                    own ray's basis (prismPlanes: three 2 x 2-minor vectors once per triangle, then six dot products per ray);
//                    attribute = (b0 a0 + b1 a1) + b2 a2
//   depth clipping   a fragment is kept iff nearDist <= -(viewMatrix * fragmentPositionWorld).z <= farDist (clip space 0 <= z <= w)
//   own-box rule     and iff the ray meets the segment's box of TubeAabbRenderData (LineDataFlow.cpp:2223-2234) with the fragment
//                    depth within r / |d| of the box interval: every ring vertex lies within r of its line point, so this only makes
//                    the result independent of which conservative acceleration structure supplied the candidate segments
//   helper lanes     fwidth(ribbonPosition) (LinePassGeometryShaderTubes.glsl:1079-1087) over the 2 x 2 quad: the quad partners
//                    (x ^ 1, y), (x, y ^ 1) evaluate the SAME triangle's attribute planes at their own pixel centre, i.e. with the
//                    (possibly outside [0, 1]) weights b_i of their own viewing ray, then the shader's ribbonPosition of the
//                    interpolated inputs (RasterQuad: the affine ray generator's directions)
#pragma once

constexpr uint32_t kPrismMaxSubdiv = 16;   // a (ray, segment) test reports its covered triangles as a 2 N-bit mask on the device

// cos / sin of the ring angle circleIdx / N * 2 pi (ProgrammablePull:129-131).  GLSL leaves their precision to the implementation;
// the build defines them by sincos2pi (the fixed polynomial of the hemisphere sample) of the fraction circleIdx / N of the full turn.
// USE_BANDS ("bands with minimum thickness", ProgrammablePull:112-116,166-171): localPosition = (thickness cos, sin, 0), localNormal =
// (cos, thickness sin, 0), lineRadius = bandWidth / 2 -- the ring is an ellipse; cp = thickness * cos feeds the positions, sn =
// thickness * sin the normals (thickness 1: the products are exact, plain tubes keep their bits).
struct PrismRing { float c[kPrismMaxSubdiv], s[kPrismMaxSubdiv], cp[kPrismMaxSubdiv], sn[kPrismMaxSubdiv]; uint32_t n; float radius, thickness; bool bands; };
inline PrismRing prismRing(uint32_t N, float radius = 0.0f, bool bands = false, float thickness = 1.0f) {
    PrismRing R;
    R.n = std::min(std::max(N, 3u), kPrismMaxSubdiv);
    R.radius = radius; R.bands = bands; R.thickness = bands ? thickness : 1.0f;
    for (uint32_t k = 0; k < R.n; k++) {
        sincos2pi(float(k) / float(R.n), R.s[k], R.c[k]);
        R.cp[k] = R.thickness * R.c[k];
        R.sn[k] = R.thickness * R.s[k];
    }
    return R;
}
// ring of a frame: radius and thickness of what the rasterised geometry is (bands: bandWidth / 2, MIN_THICKNESS)
inline PrismRing prismRingOf(const lvo_params& P, const Frame& F) {
    const bool bands = P.useBands != 0;
    return prismRing(P.tubeNumSubdivisions, bands ? P.bandWidth * 0.5f : F.radius, bands, P.minThickness);
}

// ring vertex of a line point: dir = normal * cos + binormal * sin (the tangent column of the frame meets the 0 of the local
// position), position = radius * dir + centre; vertexNormal = normalize(dir)
struct PrismVtx { V3 pos, dir; };
inline PrismVtx prismVertex(const lvo_line_point& lp, const PrismRing& R, uint32_t k, float radius) {
    const V3 normal = ld3(lp.lineNormal), tangent = ld3(lp.lineTangent), centre = ld3(lp.linePosition);
    const V3 binormal = cross(tangent, normal);
    PrismVtx v;
    // localNormal -> dir, localPosition -> pd (identical for plain tubes)
    v.dir = v3(fmaf(binormal.x, R.sn[k], normal.x * R.c[k]), fmaf(binormal.y, R.sn[k], normal.y * R.c[k]), fmaf(binormal.z, R.sn[k], normal.z * R.c[k]));
    const V3 pd = v3(fmaf(binormal.x, R.s[k], normal.x * R.cp[k]), fmaf(binormal.y, R.s[k], normal.y * R.cp[k]), fmaf(binormal.z, R.s[k], normal.z * R.cp[k]));
    v.pos = v3(fmaf(radius, pd.x, centre.x), fmaf(radius, pd.y, centre.y), fmaf(radius, pd.z, centre.z));
    return v;
}

// the two axes perpendicular to a viewing ray's direction
struct PrismBasis { V3 P, Q; };
inline PrismBasis prismBasis(const Frame& F, V3 d) {
    const V3 R = v3(F.invView[0], F.invView[1], F.invView[2]);
    PrismBasis B;
    B.P = cross(R, d);
    B.Q = cross(d, B.P);
    return B;
}
inline float prismDot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
// projection of a line point's frame into the ray's plane: the ring vertex V = centre + r (normal cos + binormal sin) has
// (V - o) . P = (centre - o) . P + r ((normal . P) cos + (binormal . P) sin) -- six dot products per (ray, line point), then two
// fused multiply-adds per coordinate of a ring vertex
struct PrismProj { float X0, Y0, nP, bP, nQ, bQ; };
inline PrismProj prismPointProj(const lvo_line_point& lp, V3 o, const PrismBasis& B) {
    const V3 normal = ld3(lp.lineNormal), tangent = ld3(lp.lineTangent), C = ld3(lp.linePosition) - o;
    const V3 binormal = cross(tangent, normal);
    PrismProj pj;
    pj.X0 = prismDot(C, B.P); pj.Y0 = prismDot(C, B.Q);
    pj.nP = prismDot(normal, B.P); pj.bP = prismDot(binormal, B.P);
    pj.nQ = prismDot(normal, B.Q); pj.bQ = prismDot(binormal, B.Q);
    return pj;
}
inline void prismVertexXY(const PrismProj& pj, float c, float s, float radius, float& x, float& y) {
    const float u = fmaf(pj.bP, s, pj.nP * c), v = fmaf(pj.bQ, s, pj.nQ * c);
    x = fmaf(radius, u, pj.X0);
    y = fmaf(radius, v, pj.Y0);
}
inline float prismEdge(float xU, float yU, float xV, float yV) { return yU * xV - xU * yV; }

// triangle tt < 2 N of a segment's prism: its three vertices as (ring 0 = first point / 1 = second point, circle index)
inline void prismTriangle(uint32_t tt, uint32_t N, uint32_t ring[3], uint32_t circ[3]) {
    const uint32_t k = tt >> 1, kn = (k + 1u) % N;
    if ((tt & 1u) == 0u) { ring[0] = 0; circ[0] = k; ring[1] = 0; circ[1] = kn; ring[2] = 1; circ[2] = k; }
    else { ring[0] = 1; circ[0] = k; ring[1] = 0; circ[1] = kn; ring[2] = 1; circ[2] = kn; }
}

// edge functions of a triangle from its projected vertices; true iff the pixel is covered (fill rule on e == 0 through the ids)
inline bool prismCoverage(const float x[3], const float y[3], const uint32_t id[3], float e[3]) {
    e[0] = prismEdge(x[1], y[1], x[2], y[2]);
    e[1] = prismEdge(x[2], y[2], x[0], y[0]);
    e[2] = prismEdge(x[0], y[0], x[1], y[1]);
    for (int i = 0; i < 3; i++) {
        const bool owned = id[(i + 1) % 3] < id[(i + 2) % 3];
        if (!(e[i] > 0.0f || (e[i] == 0.0f && owned))) return false;
    }
    return (e[0] + e[1]) + e[2] > 0.0f;
}
inline void prismWeights(const float e[3], float b[3]) {
    const float rs = 1.0f / ((e[0] + e[1]) + e[2]);   // one division (the rasteriser's set-up), three products
    b[0] = e[0] * rs; b[1] = e[1] * rs; b[2] = e[2] * rs;
}
inline V3 prismMix3(const float b[3], V3 a0, V3 a1, V3 a2) { return (b[0] * a0 + b[1] * a1) + b[2] * a2; }

// everything the fragment stage receives for one triangle of one segment
struct PrismTri {
    V3 pos[3], nrm[3], tan[3];   // vertexPosition, vertexNormal, fragmentTangent (= the line tangent of the vertex's point)
    float attr[3];
    uint32_t id[3];              // gl_VertexIndex = linePointIdx * N + circleIdx
    uint32_t ring[3], circ[3];
    float lineIdx[3];            // float(linePointIdx - lineStartIndex): interpolationFactorLine (ProgrammablePull:203-206)
    uint32_t lineStart;          // fragmentVertexIdUint (flat: the provoking = first vertex)
    V3 centre[3], lnrm[3];       // USE_BANDS varyings linePosition / lineNormal (ProgrammablePull:194-197)
};
inline PrismTri prismTriSetup(const lvo_scene& sc, const PrismRing& R, float radius, uint32_t seg, uint32_t tt) {
    const uint32_t pi[2] = {sc.segIdx[2 * seg], sc.segIdx[2 * seg + 1]};
    PrismTri T;
    prismTriangle(tt, R.n, T.ring, T.circ);
    for (int i = 0; i < 3; i++) {
        const lvo_line_point& lp = sc.pts[pi[T.ring[i]]];
        const PrismVtx v = prismVertex(lp, R, T.circ[i], radius);
        T.centre[i] = ld3(lp.linePosition);
        T.lnrm[i] = ld3(lp.lineNormal);
        T.pos[i] = v.pos;
        T.nrm[i] = normalizeShade(v.dir);   // normalize() as v * (1 / length(v)), like the shading code
        T.tan[i] = ld3(lp.lineTangent);
        T.attr[i] = lp.lineAttribute;
        T.id[i] = pi[T.ring[i]] * R.n + T.circ[i];
        T.lineIdx[i] = float(pi[T.ring[i]] - lp.lineStartIndex);
    }
    T.lineStart = sc.pts[pi[T.ring[0]]].lineStartIndex;
    return T;
}

struct PrismFrag {
    uint32_t seg, tri;
    float b[3];
    V3 pos, nrm, tan;   // interpolated fragmentPositionWorld / fragmentNormal / fragmentTangent (not normalised)
    float attr, depth;
    V3 d;               // direction of the pixel's viewing ray
};

// the shader's ribbonPosition of interpolated inputs (no bands, no caps), LinePassGeometryShaderTubes.glsl:771-777,944-963
inline float prismRibbon(V3 cam, V3 fragPos, V3 fragmentNormal, V3 fragmentTangent) {
    const V3 n = normalizeShade(fragmentNormal);
    const V3 v = normalizeShade(cam - fragPos);
    const V3 t = normalizeShade(fragmentTangent);
    const V3 helperVec = normalizeShade(cross(t, v));
    const V3 newV = normalizeShade(cross(helperVec, t));
    const V3 crossProdVn = cross(newV, n);
    float ribbonPosition = length(crossProdVn);
    if (dot(t, crossProdVn) < 0.0f) ribbonPosition = -ribbonPosition;
    return clampf(ribbonPosition, -1.0f, 1.0f);
}
// ribbonPosition a helper invocation computes: the triangle's attribute planes at the weights of the ray (o, D)
// Perspective-correct weights of a ray direction Dr (any length) in a triangle seen from o: Dr = alpha A0 + beta A1 + gamma A2 with
// A_i = V_i - o  =>  det[A_j, A_k, Dr] = weight_i det[A0, A1, A2], so b_i = e_i / (e0 + e1 + e2) with e_i = det[A_j, A_k, Dr].  Evaluated
// in the coordinates of the PIXEL'S OWN ray basis (P, Q, d) -- (X, Y, Z) = (A . P, A . Q, A . d): X and Y are small (the triangle lies
// on the ray) and the 2 x 2 minors stay well conditioned, whereas A_j x A_k in world coordinates is a cross product of two nearly
// parallel unit-length vectors -- as  e_i = (a, b, g) . c_i  with c_i = (X, Y, Z)_j x (X, Y, Z)_k computed ONCE per triangle and
// (a, b, g) = (Dr . P, Dr . Q, Dr . d): the fragment's own ray and its two helper lanes cost six fused dot products each.
// (Coverage keeps the exactly antisymmetric edge functions above: the fill rule needs bit-exact agreement between neighbouring
// triangles, the interpolation does not.)
struct PrismPlanes { V3 P, Q, D; V3 c[3]; };
inline PrismPlanes prismPlanes(const Frame& F, const PrismTri& T, V3 o, V3 d) {
    const PrismBasis B = prismBasis(F, d);
    PrismPlanes pl;
    pl.P = B.P; pl.Q = B.Q; pl.D = d;
    V3 v[3];
    for (int i = 0; i < 3; i++) {
        const V3 A = T.pos[i] - o;
        v[i] = v3(prismDot(A, B.P), prismDot(A, B.Q), prismDot(A, d));
    }
    pl.c[0] = cross(v[1], v[2]); pl.c[1] = cross(v[2], v[0]); pl.c[2] = cross(v[0], v[1]);
    return pl;
}
inline void prismRayWeights(const PrismPlanes& pl, V3 Dr, float b[3]) {
    const V3 abg = v3(prismDot(Dr, pl.P), prismDot(Dr, pl.Q), prismDot(Dr, pl.D));
    const float e[3] = {prismDot(abg, pl.c[0]), prismDot(abg, pl.c[1]), prismDot(abg, pl.c[2])};
    prismWeights(e, b);
}
inline float prismRibbonOfRay(const Frame& F, const PrismTri& T, const PrismPlanes& pl, V3 D) {
    float b[3];
    prismRayWeights(pl, D, b);
    return prismRibbon(F.cameraPosition, prismMix3(b, T.pos[0], T.pos[1], T.pos[2]), prismMix3(b, T.nrm[0], T.nrm[1], T.nrm[2]),
                       prismMix3(b, T.tan[0], T.tan[1], T.tan[2]));
}

// own-box rule (see the header): the ray meets the segment's box, depth within r / |d| of the interval
inline bool prismOwnBox(V3 o, V3 d, V3 p0, V3 p1, float radius, float depth) {
    const V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const float tx0 = ((fminf(p0.x, p1.x) - radius) - o.x) * inv.x, tx1 = ((fmaxf(p0.x, p1.x) + radius) - o.x) * inv.x;
    const float ty0 = ((fminf(p0.y, p1.y) - radius) - o.y) * inv.y, ty1 = ((fmaxf(p0.y, p1.y) + radius) - o.y) * inv.y;
    const float tz0 = ((fminf(p0.z, p1.z) - radius) - o.z) * inv.z, tz1 = ((fmaxf(p0.z, p1.z) + radius) - o.z) * inv.z;
    const float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fminf(tz0, tz1));
    const float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
    const float slack = radius / length(d);
    return tn <= tf && depth >= tn - slack && depth <= tf + slack;
}

// All fragments the pixel-centre ray (o, d) receives from the prism of segment `seg`, ascending triangle index; depth in [tLo, tHi).
inline void prismSegmentFragments(const lvo_scene& sc, const lvo_params& P, const Frame& F, const PrismRing& R, V3 o, V3 d,
                                  const PrismBasis& B, uint32_t seg, float tLo, float tHi, std::vector<PrismFrag>& out) {
    const uint32_t N = R.n;
    const uint32_t pi[2] = {sc.segIdx[2 * seg], sc.segIdx[2 * seg + 1]};
    float vx[2][kPrismMaxSubdiv], vy[2][kPrismMaxSubdiv];
    for (int r = 0; r < 2; r++) {
        const PrismProj pj = prismPointProj(sc.pts[pi[r]], o, B);
        for (uint32_t k = 0; k < N; k++) prismVertexXY(pj, R.cp[k], R.s[k], R.radius, vx[r][k], vy[r][k]);
    }
    for (uint32_t tt = 0; tt < 2u * N; tt++) {
        uint32_t ring[3], circ[3], id[3];
        prismTriangle(tt, N, ring, circ);
        float x[3], y[3], e[3];
        for (int i = 0; i < 3; i++) { x[i] = vx[ring[i]][circ[i]]; y[i] = vy[ring[i]][circ[i]]; id[i] = pi[ring[i]] * N + circ[i]; }
        if (!prismCoverage(x, y, id, e)) continue;
        const PrismTri T = prismTriSetup(sc, R, R.radius, seg, tt);
        PrismFrag f;
        f.seg = seg; f.tri = tt; f.d = d;
        prismRayWeights(prismPlanes(F, T, o, d), d, f.b);
        f.pos = prismMix3(f.b, T.pos[0], T.pos[1], T.pos[2]);
        f.nrm = prismMix3(f.b, T.nrm[0], T.nrm[1], T.nrm[2]);
        f.tan = prismMix3(f.b, T.tan[0], T.tan[1], T.tan[2]);
        f.attr = (f.b[0] * T.attr[0] + f.b[1] * T.attr[1]) + f.b[2] * T.attr[2];
        f.depth = length(f.pos - F.cameraPosition);
        if (!(f.depth >= tLo && f.depth < tHi)) continue;
        if (!prismOwnBox(o, d, ld3(sc.pts[pi[0]].linePosition), ld3(sc.pts[pi[1]].linePosition), R.radius, f.depth)) continue;
        const V4 s4 = mulM4(P.view, V4{f.pos.x, f.pos.y, f.pos.z, 1.0f});
        if (!(-s4.z >= P.nearDist && -s4.z <= P.farDist)) continue;
        out.push_back(f);
    }
}

// Coverage direction of pixel (x, y) -- build-owned, like the whole rasteriser definition (the reference's is the fixed-function unit):
// the direction of the ray generator (TubeRayTracing.glsl:219-226) BEFORE its normalisation, as the affine function of the pixel it is:
//   target = invProj (ndc.x, ndc.y, 1, 1), dir = invView (target.xyz, 0), ndc.x = 2 (x + 0.5) / W - 1
//   => D = C0 + (x + 0.5) Cx + (y + 0.5) Cy,  a = invView3 invProj[:,0].xyz, b = invView3 invProj[:,1].xyz, c = invView3 (invProj[:,2] +
//   invProj[:,3]).xyz, Cx = a (2 / W), Cy = b (2 / H), C0 = (c - a) - b; evaluated with one fma per term.  The edge functions are
//   homogeneous in the direction, so this decides the same coverage as the normalised ray except where an edge function rounds to a
//   different side of zero.
inline V3 prismCoverageDir(const lvo_params& P, const Frame& F, uint32_t x, uint32_t y) {
    auto mul3 = [&](const float* v, float out[3]) {
        for (int k = 0; k < 3; k++) out[k] = (F.invView[k] * v[0] + F.invView[4 + k] * v[1]) + F.invView[8 + k] * v[2];
    };
    const float p23[3] = {F.invProj[8] + F.invProj[12], F.invProj[9] + F.invProj[13], F.invProj[10] + F.invProj[14]};
    float a[3], b[3], c[3], D[3];
    mul3(F.invProj, a);
    mul3(F.invProj + 4, b);
    mul3(p23, c);
    const float sx = 2.0f / float(P.width), sy = 2.0f / float(P.height);
    const float fx = float(x) + 0.5f, fy = float(y) + 0.5f;
    for (int k = 0; k < 3; k++) {
        const float Cx = a[k] * sx, Cy = b[k] * sy, C0 = (c[k] - a[k]) - b[k];
        D[k] = fmaf(fx, Cx, fmaf(fy, Cy, C0));
    }
    return v3(D[0], D[1], D[2]);
}

// candidate segments of a ray: every segment whose (padded) box the ray meets within [tMin - slack, tMax + slack], ascending
inline void prismCandidates(const lvo_scene& sc, bool useBvh, V3 o, V3 d, float tMin, float tMax, float slack,
                            std::vector<uint32_t>& out, Counters& cnt) {
    out.clear();
    if (!useBvh || sc.root == -1) {
        for (uint32_t s = 0; s < sc.nSeg; s++) out.push_back(s);
        return;
    }
    if (sc.rootIsLeaf) { out.push_back(uint32_t(~sc.root)); return; }
    const V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    std::vector<int32_t> stack;
    stack.push_back(sc.root);
    while (!stack.empty()) {
        const int32_t n = stack.back(); stack.pop_back();
        if (n < 0) { out.push_back(uint32_t(~n)); continue; }
        const BvhNode& nd = sc.nodes[n];
        cnt.nodes++;
        float tl, tr;
        if (childBox(sc, nd.right, o, inv, tMin - slack, tMax + slack, tr)) stack.push_back(nd.right);
        if (childBox(sc, nd.left, o, inv, tMin - slack, tMax + slack, tl)) stack.push_back(nd.left);
    }
    std::sort(out.begin(), out.end());
}

// all prism fragments of the pixel-centre ray of pixel (x, y), ascending (segment, triangle)
inline void prismPixelFragments(const lvo_scene& sc, const lvo_params& P, const Frame& F, const PrismRing& R, bool useBvh,
                                uint32_t x, uint32_t y, std::vector<uint32_t>& cand, std::vector<PrismFrag>& out, Counters& cnt) {
    V3 o, d;
    primaryRay(P, F, x, y, 0.5f, 0.5f, o, d);
    cnt.rays++;
    out.clear();
    const PrismBasis B = prismBasis(F, prismCoverageDir(P, F, x, y));   // coverage: the unnormalised direction; all else: the ray (o, d)
    const float tMin = 0.0001f, tMax = 1000.0f;   // the gather's ray interval (depth clipping is the near / far test above)
    prismCandidates(sc, useBvh, o, d, tMin, tMax, R.radius / length(d), cand, cnt);
    for (uint32_t seg : cand) {
        cnt.prims++;
        prismSegmentFragments(sc, P, F, R, o, d, B, seg, tMin, nextafterf(tMax, INFINITY), out);
    }
}

// fragment stage: LinePassGeometryShaderTubes.glsl:732-1129 on the interpolated inputs -> colour (raster variant of the outline)
// rq == nullptr: the ray tracer's variant of computeFragmentColor (deviation switch ppll_fragment_colour = ray_tracer)
// STATIC_AMBIENT_OCCLUSION_PREBAKING in the raster shaders: the inputs of getAoFactor(fragmentVertexId, phi) (Lighting.glsl:124-125,
// AmbientOcclusion.glsl:49-75) -- vertex stage LinePassProgrammablePullTubes.glsl:179-206 (phi = circleIdx * (2 pi / N),
// interpolateWrap = circleIdx == N - 1 and fragmentVertexIdUint = lineStartIndex FLAT = of the provoking = first vertex,
// interpolationFactorLine = float(linePointIdx - lineStartIndex) interpolated), fragment stage LinePassGeometryShaderTubes.glsl:
// 753-770 (fragmentVertexId = interpolationFactorLine + float(fragmentVertexIdUint); the wrap-around of phi on the last facet)
inline void prismAoInputs(const PrismTri& T, const float b[3], uint32_t N, float& fragmentVertexId, float& phi) {
    const float PI = 3.14159265358979323846f;
    const float interpolationFactorLine = (b[0] * T.lineIdx[0] + b[1] * T.lineIdx[1]) + b[2] * T.lineIdx[2];
    fragmentVertexId = interpolationFactorLine + float(T.lineStart);
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

// USE_BANDS fragment stage (LinePassGeometryShaderTubes.glsl:819-936): the halo coordinate of the band at the interpolated varyings
// (phi with the wrap-around of the last facet :761-775, linePosition, lineNormal, fragmentTangent) -- the same polar construction as
// RayHitCommon's (bandsRibbonPosition)
inline float prismBandRibbon(const lvo_params& P, const Frame& F, const PrismRing& R, const PrismTri& T, const float b[3]) {
    float fragmentVertexId, phi;
    prismAoInputs(T, b, R.n, fragmentVertexId, phi);
    const V3 linePosition = prismMix3(b, T.centre[0], T.centre[1], T.centre[2]);
    const V3 lineNormal = prismMix3(b, T.lnrm[0], T.lnrm[1], T.lnrm[2]);
    const V3 fragmentTangent = prismMix3(b, T.tan[0], T.tan[1], T.tan[2]);
    return bandsRibbonPosition(F.cameraPosition, linePosition, lineNormal, fragmentTangent, normalizeShade(fragmentTangent), phi, R.radius,
                               R.thickness);
}
inline void prismShade(const lvo_scene& sc, const lvo_params& P, const Frame& F, const PrismRing& R, float aoTexel, const PrismFrag& f,
                       const RasterQuad* rq, float hitColor[4], float& payloadHitT, const PrebakedAo* pb = nullptr) {
    if (R.bands) {
        const PrismTri T = prismTriSetup(sc, R, R.radius, f.seg, f.tri);
        BandArgs rb;
        rb.shadeBands = true; rb.useBand = true;
        float fragmentVertexId;
        prismAoInputs(T, f.b, R.n, fragmentVertexId, rb.phi);
        if (pb) aoTexel = prebakedAoLookup(*pb, fragmentVertexId, rb.phi);
        rb.linePosition = prismMix3(f.b, T.centre[0], T.centre[1], T.centre[2]);
        rb.lineNormal = prismMix3(f.b, T.lnrm[0], T.lnrm[1], T.lnrm[2]);
        rb.rasterEpsWhite = -1.0f;
        if (rq) {
            const PrismPlanes pl = prismPlanes(F, T, F.cameraPosition, f.d);
            float bx[3], by[3];
            prismRayWeights(pl, rq->dX, bx);
            prismRayWeights(pl, rq->dY, by);
            const float f0 = prismBandRibbon(P, F, R, T, f.b), fx = prismBandRibbon(P, F, R, T, bx), fy = prismBandRibbon(P, F, R, T, by);
            rb.rasterEpsWhite = fabsf(fx - f0) + fabsf(fy - f0);
        }
        computeFragmentColor(sc, P, F, aoTexel, f.pos, f.nrm, f.tan, false, f.attr, hitColor, payloadHitT, &rb);
        return;
    }
    if (pb) {
        const PrismTri T = prismTriSetup(sc, R, F.radius, f.seg, f.tri);
        float fragmentVertexId, phi;
        prismAoInputs(T, f.b, R.n, fragmentVertexId, phi);
        aoTexel = prebakedAoLookup(*pb, fragmentVertexId, phi);
    }
    BandArgs rb;
    rb.shadeBands = false; rb.useBand = false; rb.phi = 0.0f; rb.linePosition = rb.lineNormal = v3(0, 0, 0);
    rb.rasterEpsWhite = -1.0f;
    if (rq) {
        const PrismTri T = prismTriSetup(sc, R, F.radius, f.seg, f.tri);
        const float f0 = prismRibbon(F.cameraPosition, f.pos, f.nrm, f.tan);
        const PrismPlanes pl = prismPlanes(F, T, F.cameraPosition, f.d);   // the basis of the pixel's own viewing ray
        const float fx = prismRibbonOfRay(F, T, pl, rq->dX);
        const float fy = prismRibbonOfRay(F, T, pl, rq->dY);
        rb.rasterEpsWhite = fabsf(fx - f0) + fabsf(fy - f0);
    }
    if (P.useHelicityBands) {
        // USE_ROTATING_HELICITY_BANDS in the raster shaders: vertex stage fragmentRotation = lineRotation * helicityRotationFactor
        // (ProgrammablePull:212-214), interpolated; phi as for the AO lookup (:179-190, fragment stage :761-775); UNIFORM_HELICITY_BAND_WIDTH
        // (LinePassGeometryShaderTubes.glsl:1017-1034): the two line points around floor(fragmentVertexId) (reads past the buffer give
        // zeros: robust buffer access)
        const PrismTri T = prismTriSetup(sc, R, F.radius, f.seg, f.tri);
        const uint32_t pi[2] = {sc.segIdx[2 * f.seg], sc.segIdx[2 * f.seg + 1]};
        HelicityArgs hl;
        float fragmentVertexId;
        prismAoInputs(T, f.b, R.n, fragmentVertexId, hl.phi);
        const float fr = P.helicityRotationFactor;
        float rot[3];
        for (int i = 0; i < 3; i++) rot[i] = sc.pts[pi[T.ring[i]]].lineRotation * fr;
        hl.fragmentRotation = (f.b[0] * rot[0] + f.b[1] * rot[1]) + f.b[2] * rot[2];
        if (rq) {   // the raster shader's stripe: aaf = fwidth(phi + fragmentRotation) from the quad partners' interpolated varyings
            const PrismPlanes pl = prismPlanes(F, T, F.cameraPosition, f.d);
            float g[2];
            const V3 dirs[2] = {rq->dX, rq->dY};
            for (int k = 0; k < 2; k++) {
                float bq[3], vid, ph;
                prismRayWeights(pl, dirs[k], bq);
                prismAoInputs(T, bq, R.n, vid, ph);
                g[k] = ph + ((bq[0] * rot[0] + bq[1] * rot[1]) + bq[2] * rot[2]);
            }
            const float g0 = hl.phi + hl.fragmentRotation;
            hl.rasterAaf = fabsf(g[0] - g0) + fabsf(g[1] - g0);
            const float twoPi = 2.0f * 3.14159265358979323846f;
            hl.dx = g[0] / twoPi - g0 / twoPi;   // dFdx / dFdy of globalPos = (phi + fragmentRotation) / twoPi
            hl.dy = g[1] / twoPi - g0 / twoPi;
        }
        hl.rotationSeparatorScale = 1.0f;
        if (P.uniformHelicityBandWidth) {
            const uint32_t vertexIdx0 = uint32_t(floorf(fragmentVertexId)), vertexIdx1 = vertexIdx0 + 1u;
            const uint32_t np = uint32_t(sc.pts.size());
            const V3 p0 = vertexIdx0 < np ? ld3(sc.pts[vertexIdx0].linePosition) : v3(0, 0, 0);
            const V3 p1 = vertexIdx1 < np ? ld3(sc.pts[vertexIdx1].linePosition) : v3(0, 0, 0);
            const float r0 = vertexIdx0 < np ? sc.pts[vertexIdx0].lineRotation : 0.0f, r1 = vertexIdx1 < np ? sc.pts[vertexIdx1].lineRotation : 0.0f;
            const float rotDx = length(p1 - p0);
            const float rotDy = (r1 - r0) * fr;
            float sn, cs;
            sincosRad(atan2Det(rotDy * 0.5f * P.lineWidth, rotDx), sn, cs);
            hl.rotationSeparatorScale = cs;
        }
        computeFragmentColor(sc, P, F, aoTexel, f.pos, f.nrm, f.tan, false, f.attr, hitColor, payloadHitT, &rb, &hl);
        return;
    }
    computeFragmentColor(sc, P, F, aoTexel, f.pos, f.nrm, f.tan, false, f.attr, hitColor, payloadHitT, &rb);
}
