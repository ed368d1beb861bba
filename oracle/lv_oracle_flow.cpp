/*
 * lv_oracle_flow.cpp -- CPU ORACLE, streamline tracing on a regular grid (the step before the
 * hot path).  TEST INFRASTRUCTURE, NOT PRODUCT (see lv_oracle.h).  PARITY UNPINNED.
 *
 * Restates, relative to /root/reference/src/LineData/Flow:
 *   StreamlineTracingGrid::setGridExtent / addVectorField (max magnitude)   StreamlineTracingGrid.cpp:81-218
 *   _getVectorAtIdx / _getVectorAtPosition (trilinear)                      :885-913
 *   _getScalarFieldAtIdx / _pushTrajectoryAttributes                        :857-862,1012-1047
 *   _rayBoxPlaneIntersection / _rayBoxIntersection                          :949-1010
 *   _trace (time step, termination rules, boundary clamp)                   :1193-1259
 *   _integrationStep{ExplicitEuler,ImplicitEuler,Heun,Midpoint,RK4,RKF45}   :1278-1396
 *   traceStreamlines (forward / backward / both, minimum-length filter)     :344-426, _reverseTrajectory :1118,
 *                                                                           _insertBackwardTrajectory :1149
 *   AbcFlowGenerator::generateAbcFlow                                       Loader/AbcFlowGenerator.cpp:41-72
 * Seeds are inputs (the reference's random seeders draw from std::uniform_real_distribution, whose sequence is
 * implementation defined).  Owned by the build: sgl::AABB3::contains (inclusive on both ends) and glm::normalize
 * (stated here as v / length(v), like everywhere else in this oracle).
 */
#include "lv_oracle_common.h"
#include <algorithm>
#include <deque>
#include <limits>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct Grid {
    int xs, ys, zs;
    float dx, dy, dz;
    V3 boxMin, boxMax;
    const float* V;                       // xs*ys*zs*3
    const float* const* scalars;          // numScalars pointers to xs*ys*zs floats
    uint32_t numScalars;
};

inline V3 vectorAtIdx(const Grid& g, int x, int y, int z, bool forwardMode) {
    if (x < 0 || y < 0 || z < 0 || x >= g.xs || y >= g.ys || z >= g.zs) return v3(0.0f, 0.0f, 0.0f);
    const float* p = g.V + 3 * (size_t(x) + size_t(y) * g.xs + size_t(z) * g.xs * g.ys);
    return forwardMode ? v3(p[0], p[1], p[2]) : v3(-p[0], -p[1], -p[2]);
}

struct Cell { int x, y, z; float fx, fy, fz, ix, iy, iz; };

// gridPositionFloat = (p - box.min) * (1/dx, 1/dy, 1/dz); ivec3() truncates towards zero, fract() = x - floor(x)
inline Cell locate(const Grid& g, V3 p) {
    V3 q = p - g.boxMin;
    q = v3(q.x * (1.0f / g.dx), q.y * (1.0f / g.dy), q.z * (1.0f / g.dz));
    Cell c;
    c.x = int(q.x); c.y = int(q.y); c.z = int(q.z);
    c.fx = q.x - floorf(q.x); c.fy = q.y - floorf(q.y); c.fz = q.z - floorf(q.z);
    c.ix = 1.0f - c.fx; c.iy = 1.0f - c.fy; c.iz = 1.0f - c.fz;
    return c;
}

inline V3 vectorAtPosition(const Grid& g, V3 p, bool fw) {
    const Cell c = locate(g, p);
    V3 r = (c.ix * c.iy * c.iz) * vectorAtIdx(g, c.x, c.y, c.z, fw);
    r = r + (c.fx * c.iy * c.iz) * vectorAtIdx(g, c.x + 1, c.y, c.z, fw);
    r = r + (c.ix * c.fy * c.iz) * vectorAtIdx(g, c.x, c.y + 1, c.z, fw);
    r = r + (c.fx * c.fy * c.iz) * vectorAtIdx(g, c.x + 1, c.y + 1, c.z, fw);
    r = r + (c.ix * c.iy * c.fz) * vectorAtIdx(g, c.x, c.y, c.z + 1, fw);
    r = r + (c.fx * c.iy * c.fz) * vectorAtIdx(g, c.x + 1, c.y, c.z + 1, fw);
    r = r + (c.ix * c.fy * c.fz) * vectorAtIdx(g, c.x, c.y + 1, c.z + 1, fw);
    r = r + (c.fx * c.fy * c.fz) * vectorAtIdx(g, c.x + 1, c.y + 1, c.z + 1, fw);
    return r;
}

inline float scalarAtIdx(const Grid& g, const float* f, int x, int y, int z) {
    if (x < 0 || y < 0 || z < 0 || x >= g.xs || y >= g.ys || z >= g.zs) return 0.0f;
    return f[size_t(x) + size_t(y) * g.xs + size_t(z) * g.xs * g.ys];
}

inline float scalarAtPosition(const Grid& g, const float* f, V3 p) {
    const Cell c = locate(g, p);
    float r = (c.ix * c.iy * c.iz) * scalarAtIdx(g, f, c.x, c.y, c.z);
    r = r + (c.fx * c.iy * c.iz) * scalarAtIdx(g, f, c.x + 1, c.y, c.z);
    r = r + (c.ix * c.fy * c.iz) * scalarAtIdx(g, f, c.x, c.y + 1, c.z);
    r = r + (c.fx * c.fy * c.iz) * scalarAtIdx(g, f, c.x + 1, c.y + 1, c.z);
    r = r + (c.ix * c.iy * c.fz) * scalarAtIdx(g, f, c.x, c.y, c.z + 1);
    r = r + (c.fx * c.iy * c.fz) * scalarAtIdx(g, f, c.x + 1, c.y, c.z + 1);
    r = r + (c.ix * c.fy * c.fz) * scalarAtIdx(g, f, c.x, c.y + 1, c.z + 1);
    r = r + (c.fx * c.fy * c.fz) * scalarAtIdx(g, f, c.x + 1, c.y + 1, c.z + 1);
    return r;
}

inline bool contains(const Grid& g, V3 p) {
    return p.x >= g.boxMin.x && p.y >= g.boxMin.y && p.z >= g.boxMin.z && p.x <= g.boxMax.x && p.y <= g.boxMax.y &&
           p.z <= g.boxMax.z;
}

// :949-989
inline bool rayBoxPlane(float o, float d, float lower, float upper, float& tNear, float& tFar) {
    if (fabsf(d) < 0.00001f) {
        if (o < lower || o > upper) return false;
    } else {
        float t0 = (lower - o) / d, t1 = (upper - o) / d;
        if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
        if (t0 > tNear) tNear = t0;
        if (t1 < tFar) tFar = t1;
        if (tNear > tFar) return false;
        if (tFar < 0) return false;
    }
    return true;
}
inline bool rayBox(V3 o, V3 d, V3 lower, V3 upper, float& tNear, float& tFar) {
    tNear = -3.402823466e+38f; // std::numeric_limits<float>::lowest()
    tFar = 3.402823466e+38f;
    if (!rayBoxPlane(o.x, d.x, lower.x, upper.x, tNear, tFar)) return false;
    if (!rayBoxPlane(o.y, d.y, lower.y, upper.y, tNear, tFar)) return false;
    if (!rayBoxPlane(o.z, d.z, lower.z, upper.z, tNear, tFar)) return false;
    return true;
}

struct Line {
    std::vector<V3> pos;
    std::vector<std::vector<float>> att;
    std::vector<V3> ribbon;   // streamribbons: one direction per point
};

inline void pushPoint(const Grid& g, Line& l, V3 p) {
    l.pos.push_back(p);
    if (l.att.empty()) l.att.resize(g.numScalars);
    for (uint32_t a = 0; a < g.numScalars; a++) l.att[a].push_back(scalarAtPosition(g, g.scalars[a], p));
}

// _getVectorAtIdxDouble / _getVectorAtPositionDouble, :915-944
struct D3 { double x, y, z; };
inline D3 d3(double x, double y, double z) { return D3{x, y, z}; }
inline D3 operator+(D3 a, D3 b) { return d3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline D3 operator-(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline D3 operator*(D3 a, double s) { return d3(a.x * s, a.y * s, a.z * s); }
inline D3 operator*(double s, D3 a) { return d3(s * a.x, s * a.y, s * a.z); }

inline D3 vectorAtIdxDouble(const Grid& g, int x, int y, int z, bool fw) {
    const V3 v = vectorAtIdx(g, x, y, z, fw);
    return d3(double(v.x), double(v.y), double(v.z));
}

inline D3 vectorAtPositionDouble(const Grid& g, D3 p, bool fw) {
    D3 q = p - d3(double(g.boxMin.x), double(g.boxMin.y), double(g.boxMin.z));
    q = d3(q.x * (1.0 / double(g.dx)), q.y * (1.0 / double(g.dy)), q.z * (1.0 / double(g.dz)));
    const int x = int(q.x), y = int(q.y), z = int(q.z);
    const double fx = q.x - floor(q.x), fy = q.y - floor(q.y), fz = q.z - floor(q.z);
    const double ix = 1.0 - fx, iy = 1.0 - fy, iz = 1.0 - fz;
    D3 r = (ix * iy * iz) * vectorAtIdxDouble(g, x, y, z, fw);
    r = r + (fx * iy * iz) * vectorAtIdxDouble(g, x + 1, y, z, fw);
    r = r + (ix * fy * iz) * vectorAtIdxDouble(g, x, y + 1, z, fw);
    r = r + (fx * fy * iz) * vectorAtIdxDouble(g, x + 1, y + 1, z, fw);
    r = r + (ix * iy * fz) * vectorAtIdxDouble(g, x, y, z + 1, fw);
    r = r + (fx * iy * fz) * vectorAtIdxDouble(g, x + 1, y, z + 1, fw);
    r = r + (ix * fy * fz) * vectorAtIdxDouble(g, x, y + 1, z + 1, fw);
    r = r + (fx * fy * fz) * vectorAtIdxDouble(g, x + 1, y + 1, z + 1, fw);
    return r;
}

// _integrationStepRKF45, :1341-1396: double precision, the step only ever shrinks (and stays shrunk: dt is passed by
// reference); approximationRK4 of the reference is dead code and not restated.  pow() is libm's.
inline void integrationStepRKF45(const Grid& g, float timeStepScale, V3& fP0, float& fDt, bool fw) {
    const double EPSILON = double(2.0 * 1e-5) * double(std::min(g.dx, std::min(g.dy, g.dz))) * double(timeStepScale);
    const int MAX_NUM_ITERATIONS = 100;
    double dt = fDt;
    int iteration = 0;
    const D3 p0 = d3(fP0.x, fP0.y, fP0.z);
    D3 rk5;
    bool adapt;
    do {
        const D3 k1 = dt * vectorAtPositionDouble(g, p0, fw);
        const D3 k2 = dt * vectorAtPositionDouble(g, p0 + k1 * double(1.0 / 4.0), fw);
        const D3 k3 = dt * vectorAtPositionDouble(g, p0 + k1 * double(3.0 / 32.0) + k2 * double(9.0 / 32.0), fw);
        const D3 k4 = dt * vectorAtPositionDouble(
                g, p0 + k1 * double(1932.0 / 2197.0) - k2 * double(7200.0 / 2197.0) + k3 * double(7296.0 / 2197.0), fw);
        const D3 k5 = dt * vectorAtPositionDouble(
                g, p0 + k1 * double(439.0 / 216.0) - k2 * double(8.0) + k3 * double(3680.0 / 513.0) - k4 * double(845.0 / 4104.0), fw);
        const D3 k6 = dt * vectorAtPositionDouble(
                g, p0 - k1 * double(8.0 / 27.0) + k2 * double(2.0) - k3 * double(3544.0 / 2565.0) + k4 * double(1859.0 / 4104.0)
                           - k5 * double(11.0 / 40.0), fw);
        rk5 = p0 + k1 * double(16.0 / 135.0) + k3 * double(6656.0 / 12825.0) + k4 * double(28561.0 / 56430.0)
              - k5 * double(9.0 / 50.0) + k6 * double(2.0 / 55.0);
        const D3 e = k1 * double(1.0 / 360.0) + k3 * double(-128.0 / 4275.0) + k4 * double(-2197.0 / 75240.0)
                     + k5 * (1.0 / 50.0) + k6 * double(2.0 / 55.0);
        const double TE = sqrt(e.x * e.x + e.y * e.y + e.z * e.z);
        adapt = TE > EPSILON;
        if (adapt) dt = 0.9 * dt * pow(EPSILON / TE, double(1.0 / 5.0));
        iteration++;
    } while (adapt && iteration < MAX_NUM_ITERATIONS);
    fP0 = v3(float(rk5.x), float(rk5.y), float(rk5.z));
    fDt = float(dt);
}

inline void integrationStep(const Grid& g, uint32_t method, V3& p0, float& dt, bool fw, float timeStepScale) {
    switch (method) {
        case 1: { // implicit Euler by fixed-point iteration, :1285-1307
            const float EPSILON = 1e-6f;
            const int MAX_NUM_ITERATIONS = 100;
            int iteration = 0;
            V3 pLast = p0;
            float diff;
            do {
                V3 pNext = p0 + dt * vectorAtPosition(g, pLast, fw);
                diff = length(pLast - pNext);
                pLast = pNext;
                iteration++;
            } while (diff > EPSILON && iteration < MAX_NUM_ITERATIONS);
            p0 = pLast;
            break;
        }
        case 5:
            integrationStepRKF45(g, timeStepScale, p0, dt, fw);
            break;
        case 0: // explicit Euler, :1278-1283
            p0 = p0 + dt * vectorAtPosition(g, p0, fw);
            break;
        case 2: { // Heun, :1309-1318
            V3 v0 = vectorAtPosition(g, p0, fw);
            V3 p1 = p0 + dt * v0;
            V3 v1 = vectorAtPosition(g, p1, fw);
            p0 = p0 + (dt * 0.5f) * (v0 + v1);
            break;
        }
        case 3: { // midpoint, :1320-1327
            V3 pp = p0 + (dt * 0.5f) * vectorAtPosition(g, p0, fw);
            p0 = p0 + dt * vectorAtPosition(g, pp, fw);
            break;
        }
        default: { // RK4, :1329-1339
            V3 k1 = dt * vectorAtPosition(g, p0, fw);
            V3 k2 = dt * vectorAtPosition(g, p0 + k1 * 0.5f, fw);
            V3 k3 = dt * vectorAtPosition(g, p0 + k2 * 0.5f, fw);
            V3 k4 = dt * vectorAtPosition(g, p0 + k3, fw);
            const float s6 = 6.0f, s3 = 3.0f;
            V3 a = v3(k1.x / s6, k1.y / s6, k1.z / s6), b = v3(k2.x / s3, k2.y / s3, k2.z / s3);
            V3 c = v3(k3.x / s3, k3.y / s3, k3.z / s3), d = v3(k4.x / s6, k4.y / s6, k4.z / s6);
            p0 = p0 + (((a + b) + c) + d);
            break;
        }
    }
}

// :1193-1259
void traceOne(const Grid& g, const lvo_streamline_settings& S, float maxVectorMagnitude, V3 seed, bool fw, Line& line) {
    float dt = 1.0f / maxVectorMagnitude * std::min(g.dx, std::min(g.dy, g.dz)) * S.timeStepScale; // RKF45 shrinks it
    const float terminationDistance = 1e-6f * S.terminationDistance;
    V3 p = seed, old;
    int iterationCounter = 0;
    const int MAX_ITERATIONS = std::min(int(roundf(float(S.maxNumIterations) / S.timeStepScale)), S.maxNumIterations * 10);
    float lineLength = 0.0f;
    const V3 dim = g.boxMax - g.boxMin;
    const float MAX_LINE_LENGTH = length(dim) * (float(S.maxNumIterations) / float(2000));
    while (iterationCounter <= MAX_ITERATIONS && lineLength <= MAX_LINE_LENGTH) {
        old = p;
        if (!contains(g, p)) {
            if (!line.pos.empty()) {
                V3 ro = line.pos.back();
                V3 rd = normalize(p - ro);
                float tNear, tFar;
                rayBox(ro, rd, g.boxMin, g.boxMax, tNear, tFar);
                V3 b = tNear > 0.0f ? ro + tNear * rd : ro + tFar * rd;
                pushPoint(g, line, b);
            }
            break;
        }
        pushPoint(g, line, p);
        integrationStep(g, S.integrationMethod, p, dt, fw, S.timeStepScale);
        float segmentLength = length(p - old);
        lineLength += segmentLength;
        if (segmentLength < terminationDistance) break;
        iterationCounter++;
    }
}

// glm::rotate(v, angle, normal) (gtx/rotate_vector.inl): mat3(glm::rotate(angle, normal)) * v, the axis-angle matrix of
// gtc/matrix_transform.inl
inline V3 rotateVec(V3 v, float angle, V3 normal) {
    const float c = cosf(angle), s = sinf(angle);
    const V3 axis = normalize(normal);
    const V3 temp = (1.0f - c) * axis;
    const V3 c0 = v3(c + temp.x * axis.x, temp.x * axis.y + s * axis.z, temp.x * axis.z - s * axis.y);
    const V3 c1 = v3(temp.y * axis.x - s * axis.z, c + temp.y * axis.y, temp.y * axis.z + s * axis.x);
    const V3 c2 = v3(temp.z * axis.x + s * axis.y, temp.z * axis.y - s * axis.x, c + temp.z * axis.z);
    return (c0 * v.x + c1 * v.y) + c2 * v.z;
}

struct RibbonSettings { bool useHelicity; float maxHelicityTwist; V3 initialRibbonDirection; const float* helicityField; float maxHelicityMagnitude; };

// StreamlineTracingGrid::_pushRibbonDirections, StreamlineTracingGrid.cpp:1049-1116: the ribbon direction is carried from point
// to point (Gram-Schmidt against the tangent, fallback axes z then y) and twisted about the tangent by the local helicity
inline void pushRibbonDirections(const Grid& g, const RibbonSettings& R, const std::vector<V3>& positions, std::vector<V3>& dirs,
                                 bool forwardMode) {
    V3 lastRibbonDirection = normalize(R.initialRibbonDirection);
    const size_t n = positions.size();
    if (n == 1) { dirs.push_back(lastRibbonDirection); return; }
    for (size_t i = 0; i < n; i++) {
        V3 tangent;
        if (i == 0) tangent = positions[i + 1] - positions[i];
        else if (i == n - 1) tangent = positions[i] - positions[i - 1];
        else tangent = positions[i + 1] - positions[i - 1];
        tangent = normalize(tangent);
        const V3 particlePosition = positions[i];
        V3 helperAxis = lastRibbonDirection;
        if (length(cross(helperAxis, tangent)) < 1e-2f) {
            helperAxis = v3(0.0f, 0.0f, 1.0f);
            if (length(cross(helperAxis, tangent)) < 1e-2f) helperAxis = v3(0.0f, 1.0f, 0.0f);
        }
        V3 ribbonDirection = normalize(helperAxis - dot(helperAxis, tangent) * tangent);
        if (R.useHelicity) {
            float helicity = scalarAtPosition(g, R.helicityField, particlePosition);
            if (!forwardMode) helicity *= -1.0f;
            float lineSegmentLength = 0.0f;
            if (i < n - 1) lineSegmentLength = length(positions[i + 1] - positions[i]);
            const float helicityAngle = helicity / R.maxHelicityMagnitude * 3.14159265358979323846f * R.maxHelicityTwist
                                        * lineSegmentLength / 0.005f;
            ribbonDirection = rotateVec(ribbonDirection, helicityAngle, tangent);
        }
        dirs.push_back(ribbonDirection);
        lastRibbonDirection = ribbonDirection;
    }
}

inline void reverseLine(Line& l) { // _reverseTrajectory / _reverseRibbon, :1118-1146
    if (l.pos.size() <= 1) return;
    std::reverse(l.pos.begin(), l.pos.end());
    std::reverse(l.ribbon.begin(), l.ribbon.end());
    for (auto& a : l.att) std::reverse(a.begin(), a.end());
}

} // namespace

struct lvo_streamlines {
    std::vector<float> positions;
    std::vector<std::vector<float>> attributes;
    std::vector<uint32_t> offsets;
    std::vector<float> ribbonDirections; // streamribbons only
};

extern "C" {

// Loader/AbcFlowGenerator.cpp:41-72 (A = sqrt(3), B = sqrt(2), C = 1 at :41-45)
void lvo_generate_abc_flow(float* v, int xs, int ys, int zs, float A, float B, float C, float resScale) {
    for (int iz = 0; iz < zs; iz++)
        for (int iy = 0; iy < ys; iy++)
            for (int ix = 0; ix < xs; ix++) {
                float x = float(ix) / float(xs - 1) * resScale;
                float y = float(iy) / float(ys - 1) * resScale;
                float z = float(iz) / float(zs - 1) * resScale;
                size_t o = size_t(iz) * xs * ys * 3 + size_t(iy) * xs * 3 + size_t(ix) * 3;
                v[o + 0] = A * sinf(z) + C * cosf(y);
                v[o + 1] = B * sinf(x) + A * cosf(z);
                v[o + 2] = C * sinf(y) + B * cosf(x);
            }
}

// addVectorField's max-magnitude reduction, :189-216
// GridLoader.cpp:41-183: |v| per grid point, curl by central / one-sided differences (every derivative divided by dy, as the
// reference writes it), helicity v . curl v with optional normalisation of either factor
void lvo_compute_vector_magnitude_field(const float* v, float* out, int xs, int ys, int zs) {
    for (size_t i = 0; i < size_t(xs) * ys * zs; i++) {
        const float vx = v[3 * i], vy = v[3 * i + 1], vz = v[3 * i + 2];
        out[i] = std::sqrt(vx * vx + vy * vy + vz * vz);
    }
}
void lvo_compute_vorticity_field(const float* v, float* out, int xs, int ys, int zs, float dx, float dy, float dz) {
    (void)dx; (void)dz;
    auto V = [&](int x, int y, int z, int c) { return v[3 * (size_t(x) + size_t(y) * xs + size_t(z) * xs * ys) + c]; };
    for (int z = 0; z < zs; z++)
        for (int y = 0; y < ys; y++)
            for (int x = 0; x < xs; x++) {
                const int left = x > 0 ? -1 : 0, right = x < xs - 1 ? 1 : 0;
                const int down = y > 0 ? -1 : 0, up = y < ys - 1 ? 1 : 0;
                const int back = z > 0 ? -1 : 0, front = z < zs - 1 ? 1 : 0;
                const float dVzdy = (V(x, y + up, z, 2) - V(x, y + down, z, 2)) / (dy * float(up - down));
                const float dVydz = (V(x, y, z + front, 1) - V(x, y, z + back, 1)) / (dy * float(front - back));
                const float dVxdz = (V(x, y, z + front, 0) - V(x, y, z + back, 0)) / (dy * float(front - back));
                const float dVzdx = (V(x + right, y, z, 2) - V(x + left, y, z, 2)) / (dy * float(right - left));
                const float dVydx = (V(x + right, y, z, 1) - V(x + left, y, z, 1)) / (dy * float(right - left));
                const float dVxdy = (V(x, y + up, z, 0) - V(x, y + down, z, 0)) / (dy * float(up - down));
                float* o = out + 3 * (size_t(x) + size_t(y) * xs + size_t(z) * xs * ys);
                o[0] = dVzdy - dVydz; o[1] = dVxdz - dVzdx; o[2] = dVydx - dVxdy;
            }
}
void lvo_compute_helicity_field_normalized(const float* vel, const float* vort, float* out, int xs, int ys, int zs,
                                           int normalizeVelocity, int normalizeVorticity) {
    for (size_t i = 0; i < size_t(xs) * ys * zs; i++) {
        float wx = vort[3 * i], wy = vort[3 * i + 1], wz = vort[3 * i + 2];
        float vx = vel[3 * i], vy = vel[3 * i + 1], vz = vel[3 * i + 2];
        if (normalizeVelocity) {
            const float m = std::sqrt(vx * vx + vy * vy + vz * vz);
            if (m > 1e-6) { vx /= m; vy /= m; vz /= m; }
        }
        if (normalizeVorticity) {
            const float m = std::sqrt(wx * wx + wy * wy + wz * wz);
            if (m > 1e-6) { wx /= m; wy /= m; wz /= m; }
        }
        out[i] = vx * wx + vy * wy + vz * wz;
    }
}

float lvo_max_vector_magnitude(const float* v, uint64_t numCells) {
    float m = 0.0f;
    for (uint64_t i = 0; i < numCells; i++) {
        float vx = v[3 * i], vy = v[3 * i + 1], vz = v[3 * i + 2];
        m = std::max(m, sqrtf(vx * vx + vy * vy + vz * vz));
    }
    return m;
}

static lvo_streamlines* traceLines(const float* vectorField, int xs, int ys, int zs, float dx, float dy, float dz,
                                   const float* const* scalarFields, uint32_t numScalarFields, const float* seeds,
                                   uint32_t numSeeds, const lvo_streamline_settings* settings, const RibbonSettings* ribbons);
lvo_streamlines* lvo_trace_streamlines(const float* vectorField, int xs, int ys, int zs, float dx, float dy, float dz,
                                       const float* const* scalarFields, uint32_t numScalarFields, const float* seeds,
                                       uint32_t numSeeds, const lvo_streamline_settings* settings) {
    return traceLines(vectorField, xs, ys, zs, dx, dy, dz, scalarFields, numScalarFields, seeds, numSeeds, settings, nullptr);
}
// traceStreamribbons (StreamlineTracingGrid.cpp:428-530): the streamlines + _pushRibbonDirections per traced part.
// helicityFieldIndex: which of the scalar fields is "Helicity"; its maximum magnitude over the grid normalises the twist (:299-320)
lvo_streamlines* lvo_trace_streamribbons(const float* vectorField, int xs, int ys, int zs, float dx, float dy, float dz,
                                         const float* const* scalarFields, uint32_t numScalarFields, const float* seeds,
                                         uint32_t numSeeds, const lvo_streamline_settings* settings, uint32_t helicityFieldIndex,
                                         int useHelicity, float maxHelicityTwist, const float* initialRibbonDirection) {
    RibbonSettings R;
    R.useHelicity = useHelicity != 0;
    R.maxHelicityTwist = maxHelicityTwist;
    R.initialRibbonDirection = ld3(initialRibbonDirection);
    R.helicityField = scalarFields[helicityFieldIndex];
    R.maxHelicityMagnitude = 0.0f;
    for (size_t i = 0; i < size_t(xs) * ys * zs; i++) R.maxHelicityMagnitude = std::max(R.maxHelicityMagnitude, fabsf(R.helicityField[i]));
    return traceLines(vectorField, xs, ys, zs, dx, dy, dz, scalarFields, numScalarFields, seeds, numSeeds, settings, &R);
}
void lvo_streamlines_copy_ribbons(const lvo_streamlines* s, float* ribbonDirections) {
    memcpy(ribbonDirections, s->ribbonDirections.data(), s->ribbonDirections.size() * 4);
}
static lvo_streamlines* traceLines(const float* vectorField, int xs, int ys, int zs, float dx, float dy, float dz,
                                   const float* const* scalarFields, uint32_t numScalarFields, const float* seeds,
                                   uint32_t numSeeds, const lvo_streamline_settings* settings, const RibbonSettings* ribbons) {
    Grid g;
    g.xs = xs; g.ys = ys; g.zs = zs; g.dx = dx; g.dy = dy; g.dz = dz;
    g.boxMin = v3(0.0f, 0.0f, 0.0f);
    g.boxMax = v3(float(xs - 1) * dx, float(ys - 1) * dy, float(zs - 1) * dz); // setGridExtent, :113-115
    g.V = vectorField; g.scalars = scalarFields; g.numScalars = numScalarFields;
    const lvo_streamline_settings& S = *settings;
    const float maxMag = lvo_max_vector_magnitude(vectorField, uint64_t(xs) * ys * zs);
    std::vector<Line> lines(numSeeds);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t i = 0; i < int64_t(numSeeds); i++) {
        Line& line = lines[size_t(i)];
        const V3 seed = ld3(seeds + 3 * i);
        if (S.integrationDirection == 0) {
            traceOne(g, S, maxMag, seed, true, line);
            if (ribbons) pushRibbonDirections(g, *ribbons, line.pos, line.ribbon, true);
        } else if (S.integrationDirection == 1) {
            traceOne(g, S, maxMag, seed, false, line);
            if (ribbons) pushRibbonDirections(g, *ribbons, line.pos, line.ribbon, false);
            reverseLine(line);
        } else {
            Line back;
            traceOne(g, S, maxMag, seed, true, line);
            traceOne(g, S, maxMag, seed, false, back);
            if (ribbons) {
                pushRibbonDirections(g, *ribbons, line.pos, line.ribbon, true);
                pushRibbonDirections(g, *ribbons, back.pos, back.ribbon, false);
            }
            reverseLine(back);
            if (back.pos.size() > 1) { // _insertBackwardTrajectory / _insertBackwardRibbon, :1149-1191
                if (ribbons) line.ribbon.insert(line.ribbon.begin(), back.ribbon.begin(), back.ribbon.end() - 1);
                line.pos.insert(line.pos.begin(), back.pos.begin(), back.pos.end() - 1);
                if (line.att.empty()) line.att.resize(numScalarFields);
                for (uint32_t a = 0; a < numScalarFields; a++)
                    line.att[a].insert(line.att[a].begin(), back.att[a].begin(), back.att[a].end() - 1);
            }
        }
    }
    lvo_streamlines* out = new lvo_streamlines();
    out->attributes.resize(numScalarFields);
    out->offsets.push_back(0);
    for (const Line& l : lines) { // minimum-length filter, :413-425
        if (l.pos.empty()) continue;
        float len = 0.0f;
        for (size_t i = 1; i < l.pos.size(); i++) len += length(l.pos[i] - l.pos[i - 1]);
        if (!(len > S.minimumLength)) continue;
        for (const V3& p : l.pos) { out->positions.push_back(p.x); out->positions.push_back(p.y); out->positions.push_back(p.z); }
        for (const V3& r : l.ribbon) { out->ribbonDirections.push_back(r.x); out->ribbonDirections.push_back(r.y); out->ribbonDirections.push_back(r.z); }
        for (uint32_t a = 0; a < numScalarFields; a++)
            out->attributes[a].insert(out->attributes[a].end(), l.att[a].begin(), l.att[a].end());
        out->offsets.push_back(uint32_t(out->positions.size() / 3));
    }
    return out;
}

// StreamlineMaxHelicityFirstSeeder (StreamlineSeeder.cpp:360-529) + StreamlineTracingGrid::_traceStreamribbonsDecreasingHelicity /
// _traceStreamlineDecreasingHelicity / _isTerminated (StreamlineTracingGrid.cpp:546-860), literally and sequentially: one line after
// the other, each terminated where it enters a cell an earlier line has claimed.  Grid-based termination check, loop check none (0) or
// start point (1), integrators 0 ... 4 (the Runge-Kutta-Fehlberg step width carries over from line to line in the reference).
// Build-owned where the reference leaves it open: samples of equal helicity keep their creation order (std::sort is unstable);
// sgl::Sphere::intersects(AABB) = squared distance from the centre to the box <= r^2; sgl::Plane(n, p).getDistance(q) = n.q - n.p.
static lvo_streamlines* traceMaxHelicityFirst(
        const float* vectorField, int xs, int ys, int zs, float dx, float dy, float dz, const float* const* scalarFields,
        uint32_t numScalarFields, const float* helicityField, const lvo_streamline_settings* settings, float minimumSeparationDistance,
        uint32_t loopCheckMode, float terminationDistanceSelf, int seedingSubsamplingFactor, const RibbonSettings* ribbons,
        uint32_t terminationCheckType = 1u);
// TerminationCheckType (StreamlineTracingDefines.hpp:89-94): 0 naive = the loop over every point of every finished trajectory
// (_isTerminated, StreamlineTracingGrid.cpp:676-684: glm::distance(currentPoint, point) < minimumSeparationDistance), without a filter on
// the seeds (StreamlineSeeder.cpp:452-454); 2 / 3 = the seeder's k-d tree / hashed grid over the same points (:503-529;
// getHasPointCloserThan is sgl's, not vendored: distance < r here, like the naive test), which also skips a sample that lies within r of a
// finished point (:441-451).  All three by the literal loop here.
lvo_streamlines* lvo_trace_streamlines_max_helicity_first_ex(
        const float* vectorField, int xs, int ys, int zs, float dx, float dy, float dz, const float* const* scalarFields,
        uint32_t numScalarFields, const float* helicityField, const lvo_streamline_settings* settings, float minimumSeparationDistance,
        uint32_t loopCheckMode, float terminationDistanceSelf, int seedingSubsamplingFactor, uint32_t terminationCheckType) {
    return traceMaxHelicityFirst(vectorField, xs, ys, zs, dx, dy, dz, scalarFields, numScalarFields, helicityField, settings,
                                 minimumSeparationDistance, loopCheckMode, terminationDistanceSelf, seedingSubsamplingFactor, nullptr,
                                 terminationCheckType);
}
lvo_streamlines* lvo_trace_streamlines_max_helicity_first(
        const float* vectorField, int xs, int ys, int zs, float dx, float dy, float dz, const float* const* scalarFields,
        uint32_t numScalarFields, const float* helicityField, const lvo_streamline_settings* settings, float minimumSeparationDistance,
        uint32_t loopCheckMode, float terminationDistanceSelf, int seedingSubsamplingFactor) {
    return traceMaxHelicityFirst(vectorField, xs, ys, zs, dx, dy, dz, scalarFields, numScalarFields, helicityField, settings,
                                 minimumSeparationDistance, loopCheckMode, terminationDistanceSelf, seedingSubsamplingFactor, nullptr);
}
// flowPrimitives == STREAMRIBBONS in _traceStreamribbonsDecreasingHelicity (:783-823): _pushRibbonDirections of every valid line's
// parts -- with forwardMode = true for the backward part too (unlike traceStreamribbons, :479-486) -- then _reverseRibbon /
// _insertBackwardRibbon.  The twist uses the same helicity field, normalised by its maximum magnitude.
// (the streamribbon entry point keeps its signature: the termination check type of the NEXT call is set through a global of this test library)
static uint32_t g_ribbonTerminationCheckType = 1u;
lvo_streamlines* lvo_trace_streamribbons_max_helicity_first(
        const float* vectorField, int xs, int ys, int zs, float dx, float dy, float dz, const float* const* scalarFields,
        uint32_t numScalarFields, const float* helicityField, const lvo_streamline_settings* settings, float minimumSeparationDistance,
        uint32_t loopCheckMode, float terminationDistanceSelf, int seedingSubsamplingFactor, int useHelicity, float maxHelicityTwist,
        const float* initialRibbonDirection) {
    RibbonSettings R;
    R.useHelicity = useHelicity != 0;
    R.maxHelicityTwist = maxHelicityTwist;
    R.initialRibbonDirection = ld3(initialRibbonDirection);
    R.helicityField = helicityField;
    R.maxHelicityMagnitude = 0.0f;
    for (size_t i = 0; i < size_t(xs) * ys * zs; i++) R.maxHelicityMagnitude = std::max(R.maxHelicityMagnitude, fabsf(helicityField[i]));
    return traceMaxHelicityFirst(vectorField, xs, ys, zs, dx, dy, dz, scalarFields, numScalarFields, helicityField, settings,
                                 minimumSeparationDistance, loopCheckMode, terminationDistanceSelf, seedingSubsamplingFactor, &R,
                                 g_ribbonTerminationCheckType);
}
void lvo_set_streamribbon_termination_check_type(uint32_t terminationCheckType) { g_ribbonTerminationCheckType = terminationCheckType; }
static lvo_streamlines* traceMaxHelicityFirst(
        const float* vectorField, int xs, int ys, int zs, float dx, float dy, float dz, const float* const* scalarFields,
        uint32_t numScalarFields, const float* helicityField, const lvo_streamline_settings* settings, float minimumSeparationDistance,
        uint32_t loopCheckMode, float terminationDistanceSelf, int seedingSubsamplingFactor, const RibbonSettings* ribbons,
        uint32_t terminationCheckType) {
    Grid g;
    g.xs = xs; g.ys = ys; g.zs = zs; g.dx = dx; g.dy = dy; g.dz = dz;
    g.boxMin = v3(0.0f, 0.0f, 0.0f);
    g.boxMax = v3(float(xs - 1) * dx, float(ys - 1) * dy, float(zs - 1) * dz);
    g.V = vectorField; g.scalars = scalarFields; g.numScalars = numScalarFields;
    const lvo_streamline_settings& S = *settings;
    const float maxMag = lvo_max_vector_magnitude(vectorField, uint64_t(xs) * ys * zs);
    struct Sample { float value; V3 pos; };
    std::vector<Sample> queue;
    const V3 dims = g.boxMax - g.boxMin;
    const int f = seedingSubsamplingFactor;
    if (f == 1) {
        for (int z = 1; z < zs - 1; z++)
            for (int y = 1; y < ys - 1; y++)
                for (int x = 1; x < xs - 1; x++)
                    queue.push_back({helicityField[size_t(x) + size_t(y) * xs + size_t(z) * xs * ys],
                                     v3(g.boxMin.x + dims.x * float(x) / float(xs), g.boxMin.y + dims.y * float(y) / float(ys),
                                        g.boxMin.z + dims.z * float(z) / float(zs))});
    } else {
        const int ncx = (xs - 1) / f, ncy = (ys - 1) / f, ncz = (zs - 1) / f;
        for (int z = 0; z < ncz; z++)
            for (int y = 0; y < ncy; y++)
                for (int x = 0; x < ncx; x++) {
                    const int xg = std::min(x * f, xs - 1), yg = std::min(y * f, ys - 1), zg = std::min(z * f, zs - 1);
                    queue.push_back({fabsf(helicityField[size_t(xg) + size_t(yg) * xs + size_t(zg) * xs * ys]),
                                     v3(g.boxMin.x + dims.x * (float(x) + 0.5f) / float(ncx), g.boxMin.y + dims.y * (float(y) + 0.5f) / float(ncy),
                                        g.boxMin.z + dims.z * (float(z) + 0.5f) / float(ncz))});
                }
    }
    std::stable_sort(queue.begin(), queue.end(), [](const Sample& a, const Sample& b) { return a.value < b.value; });
    std::vector<bool> occupancy(size_t(xs - 1) * (ys - 1) * (zs - 1), false);
    const bool gridCheck = terminationCheckType == 1u;
    std::vector<V3> finishedPoints;   // filteredTrajectories' points = what the seeder's k-d tree / hashed grid hold
    auto pointTerminated = [&](V3 p) {
        for (const V3& q : finishedPoints)
            if (length(p - q) < minimumSeparationDistance) return true;
        return false;
    };
    auto cellOf = [&](V3 p) {
        V3 q = p - g.boxMin;
        q = v3(q.x * (1.0f / dx), q.y * (1.0f / dy), q.z * (1.0f / dz));
        int x = int(q.x), y = int(q.y), z = int(q.z);
        x = std::min(std::max(x, 0), xs - 2); y = std::min(std::max(y, 0), ys - 2); z = std::min(std::max(z, 0), zs - 2);
        return size_t(x) + size_t(y) * (xs - 1) + size_t(z) * (xs - 1) * (ys - 1);
    };
    const float dt0 = 1.0f / maxMag * std::min(dx, std::min(dy, dz)) * S.timeStepScale;
    const int MAX_ITERATIONS = std::min(int(roundf(float(S.maxNumIterations) / S.timeStepScale)), S.maxNumIterations * 10) * 10;
    const float terminationDistance = 1e-6f * S.terminationDistance;
    const float terminationDistanceStart = length(dims) / 100.0f * terminationDistanceSelf;
    // members of the tracer the loop checks use, reset after every call of _traceStreamlineDecreasingHelicity (:727-737)
    std::vector<bool> selfOccupationGrid(loopCheckMode == 3u ? occupancy.size() : 0, false);
    std::deque<size_t> cellPositionQueue;
    size_t oldCellPosition = std::numeric_limits<size_t>::max();
    double curvatureSum = 0.0;
    size_t segmentSum = 0;
    auto traceDecreasing = [&](V3 seed, bool fw, Line& line) {
        struct Reset {   // (runs when the line is finished)
            std::vector<bool>& g; std::deque<size_t>& q; size_t& o; double& c; size_t& n;
            ~Reset() { std::fill(g.begin(), g.end(), false); q.clear(); o = std::numeric_limits<size_t>::max(); c = 0.0; n = 0; }
        } reset{selfOccupationGrid, cellPositionQueue, oldCellPosition, curvatureSum, segmentSum};
        float dt = dt0;
        V3 currentPoint = seed, lastPoint = seed;
        float segmentLength = 0.0f;
        int iterationCounter = 0;
        for (;;) {
            // _isTerminated
            if (iterationCounter > MAX_ITERATIONS) break;
            if (!line.pos.empty() && segmentLength < terminationDistance) break;
            if (!contains(g, currentPoint)) {
                if (!line.pos.empty()) {
                    const V3 ro = line.pos.back(), rd = normalize(currentPoint - ro);
                    float tNear, tFar;
                    rayBox(ro, rd, g.boxMin, g.boxMax, tNear, tFar);
                    pushPoint(g, line, tNear > 0.0f ? ro + tNear * rd : ro + tFar * rd);
                }
                break;
            }
            if (line.pos.size() > 1 && loopCheckMode == 1u) {
                const V3 pt0 = line.pos[0], pt1 = line.pos[1];
                V3 dir0 = pt1 - pt0;
                const float dist0 = length(dir0);
                dir0 = v3(dir0.x / dist0, dir0.y / dist0, dir0.z / dist0);
                V3 dirNow = currentPoint - line.pos.back();
                const float distNow = length(dirNow);
                dirNow = v3(dirNow.x / distNow, dirNow.y / distNow, dirNow.z / distNow);
                const float distToStart = length(currentPoint - pt0);
                const float planeDistance = dot(dir0, currentPoint) + (-dot(dir0, pt0));
                if (planeDistance < 0.0f && distToStart < terminationDistanceStart && dot(dir0, dirNow) > 0.0f) break;
            }
            if (line.pos.size() > 1 && loopCheckMode == 2u) {
                // LoopCheckMode::ALL_POINTS, :609-626.  hashedGridLoop holds (point, direction it was reached in) of every point pushed
                // after the first (:689-692); sgl::HashedGrid::findPointsAndDataInSphere is not vendored: distance <= radius here
                V3 dirNow = currentPoint - line.pos.back();
                const float distNow = length(dirNow);
                dirNow = v3(dirNow.x / distNow, dirNow.y / distNow, dirNow.z / distNow);
                bool loop = false;
                for (size_t i = 1; i < line.pos.size() && !loop; i++) {
                    const V3 pt0 = line.pos[i], dir0 = normalize(line.pos[i] - line.pos[i - 1]);
                    if (!(length(pt0 - currentPoint) <= terminationDistanceStart)) continue;
                    const float planeDistance = dot(dir0, currentPoint) + (-dot(dir0, pt0));
                    loop = planeDistance < 0.0f && distNow < terminationDistanceStart && dot(dir0, dirNow) > 0.0f;
                }
                if (loop) break;
            } else if (line.pos.size() > 1 && loopCheckMode == 3u) {
                // LoopCheckMode::GRID, :627-649; CircularQueue<size_t>(32) is sgl's (not vendored): a FIFO bounded by the caller
                const size_t cellPosition = cellOf(currentPoint);
                const bool occupied = selfOccupationGrid[cellPosition];
                selfOccupationGrid[cellPosition] = true;
                if (occupied && std::find(cellPositionQueue.begin(), cellPositionQueue.end(), cellPosition) == cellPositionQueue.end()) break;
                if (cellPosition != oldCellPosition) {
                    if (cellPositionQueue.size() == 32) cellPositionQueue.pop_front();
                    cellPositionQueue.push_back(cellPosition);
                }
                oldCellPosition = cellPosition;
            } else if (line.pos.size() > 1 && loopCheckMode == 4u) {
                // LoopCheckMode::CURVATURE, :650-671; glm::acos = std::acos (NaN outside [-1, 1]); inside: the build's fixed formula
                const V3 p0 = line.pos[line.pos.size() - 2], p1 = line.pos[line.pos.size() - 1], p2 = currentPoint;
                V3 dir0 = p1 - p0, dir1 = p2 - p1;
                const float length0 = length(dir0), length1 = length(dir1);
                if (length0 > 1e-8f) dir0 = v3(dir0.x / length0, dir0.y / length0, dir0.z / length0);
                if (length1 > 1e-8f) dir1 = v3(dir1.x / length1, dir1.y / length1, dir1.z / length1);
                const float c = dot(dir0, dir1);
                const float ang = fabsf(c) <= 1.0f ? atan2Det(sqrtf((1.0f - c) * (1.0f + c)), c) : std::numeric_limits<float>::quiet_NaN();
                curvatureSum += double(ang) * double(length0 + length1);
                segmentSum++;
                if (segmentSum > 100 && curvatureSum > 2.5f) break;
            }
            if (gridCheck ? bool(occupancy[cellOf(currentPoint)]) : pointTerminated(currentPoint)) break;
            pushPoint(g, line, currentPoint);
            integrationStep(g, S.integrationMethod, currentPoint, dt, fw, S.timeStepScale);
            iterationCounter++;
            segmentLength += length(currentPoint - lastPoint);
            lastPoint = currentPoint;
        }
    };
    auto lengthOf = [](const Line& l) {
        float len = 0.0f;
        for (size_t i = 0; i + 1 < l.pos.size(); i++) len += length(l.pos[i] - l.pos[i + 1]);
        return len;
    };
    lvo_streamlines* out = new lvo_streamlines();
    out->attributes.resize(numScalarFields);
    out->offsets.push_back(0);
    while (!queue.empty()) {
        const Sample sm = queue.back();      // hasNextPoint
        queue.pop_back();
        if (gridCheck ? bool(occupancy[cellOf(sm.pos)]) : (terminationCheckType != 0u && pointTerminated(sm.pos))) continue;
        Line line;
        bool valid;
        if (S.integrationDirection == 0) {
            traceDecreasing(sm.pos, true, line);
            valid = lengthOf(line) >= S.minimumLength;
            if (valid && ribbons) pushRibbonDirections(g, *ribbons, line.pos, line.ribbon, true);
        } else if (S.integrationDirection == 1) {
            traceDecreasing(sm.pos, false, line);
            valid = lengthOf(line) >= S.minimumLength;
            if (valid && ribbons) pushRibbonDirections(g, *ribbons, line.pos, line.ribbon, true);
            if (valid) reverseLine(line);
        } else {
            Line back;
            traceDecreasing(sm.pos, true, line);
            traceDecreasing(sm.pos, false, back);
            valid = lengthOf(line) + lengthOf(back) >= S.minimumLength;
            if (valid) {
                if (ribbons) {
                    pushRibbonDirections(g, *ribbons, line.pos, line.ribbon, true);
                    pushRibbonDirections(g, *ribbons, back.pos, back.ribbon, true);
                }
                reverseLine(back);
                if (back.pos.size() > 1) {
                    if (ribbons) line.ribbon.insert(line.ribbon.begin(), back.ribbon.begin(), back.ribbon.end() - 1);
                    line.pos.insert(line.pos.begin(), back.pos.begin(), back.pos.end() - 1);
                    if (line.att.empty()) line.att.resize(numScalarFields);
                    for (uint32_t a = 0; a < numScalarFields; a++)
                        line.att[a].insert(line.att[a].begin(), back.att[a].begin(), back.att[a].end() - 1);
                }
            }
        }
        if (!valid || line.pos.empty()) continue;
        for (const V3& p : line.pos) { out->positions.push_back(p.x); out->positions.push_back(p.y); out->positions.push_back(p.z); }
        for (const V3& rb : line.ribbon) { out->ribbonDirections.push_back(rb.x); out->ribbonDirections.push_back(rb.y); out->ribbonDirections.push_back(rb.z); }
        for (uint32_t a = 0; a < numScalarFields; a++)
            out->attributes[a].insert(out->attributes[a].end(), line.att[a].begin(), line.att[a].end());
        out->offsets.push_back(uint32_t(out->positions.size() / 3));
        // addFinishedTrajectory
        const float r = minimumSeparationDistance;
        if (!gridCheck) finishedPoints.insert(finishedPoints.end(), line.pos.begin(), line.pos.end());
        for (const V3& q : line.pos) {
            if (!gridCheck) break;
            auto coord = [](float v, float cell, int hi) { int c = int(v * (1.0f / cell)); return std::min(std::max(c, 0), hi); };
            const int x0 = coord((q.x - r) - g.boxMin.x, dx, xs - 2), x1 = coord((q.x + r) - g.boxMin.x, dx, xs - 2);
            const int y0 = coord((q.y - r) - g.boxMin.y, dy, ys - 2), y1 = coord((q.y + r) - g.boxMin.y, dy, ys - 2);
            const int z0 = coord((q.z - r) - g.boxMin.z, dz, zs - 2), z1 = coord((q.z + r) - g.boxMin.z, dz, zs - 2);
            for (int z = z0; z <= z1; z++)
                for (int y = y0; y <= y1; y++)
                    for (int x = x0; x <= x1; x++) {
                        const float lo[3] = {float(x) * dx, float(y) * dy, float(z) * dz};
                        const float hi[3] = {float(x + 1) * dx, float(y + 1) * dy, float(z + 1) * dz};
                        const float qq[3] = {q.x, q.y, q.z};
                        float d2 = 0.0f;
                        for (int c = 0; c < 3; c++) {
                            const float dd = qq[c] < lo[c] ? lo[c] - qq[c] : (qq[c] > hi[c] ? qq[c] - hi[c] : 0.0f);
                            d2 += dd * dd;
                        }
                        if (d2 <= r * r) occupancy[size_t(x) + size_t(y) * (xs - 1) + size_t(z) * (xs - 1) * (ys - 1)] = true;
                    }
        }
    }
    return out;
}

void lvo_streamlines_sizes(const lvo_streamlines* s, uint64_t* numLines, uint64_t* numPoints) {
    *numLines = s->offsets.size() - 1;
    *numPoints = s->positions.size() / 3;
}
void lvo_streamlines_copy(const lvo_streamlines* s, float* positions, float* attributes /* [numScalars][numPoints] */,
                          uint32_t* offsets) {
    const size_t n = s->positions.size() / 3;
    if (positions) memcpy(positions, s->positions.data(), s->positions.size() * 4);
    if (attributes)
        for (size_t a = 0; a < s->attributes.size(); a++) memcpy(attributes + a * n, s->attributes[a].data(), n * 4);
    if (offsets) memcpy(offsets, s->offsets.data(), s->offsets.size() * 4);
}
void lvo_streamlines_destroy(lvo_streamlines* s) { delete s; }

} // extern "C"
