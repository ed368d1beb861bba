// Flow.cpp -- see Flow.hpp for the reference classes mirrored here.
#include "Flow.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace lv {

const char* const STREAMLINE_INTEGRATION_METHOD_NAMES[6] = {"Explicit Euler", "Implicit Euler", "Heun", "Midpoint",
                                                            "Runge-Kutta 4th Order", "Runge-Kutta-Fehlberg"};
const char* const STREAMLINE_INTEGRATION_DIRECTION_NAMES[3] = {"Forward", "Backward", "Forward & Backward"};

StreamlineTracingGrid::StreamlineTracingGrid(int deviceOrdinal) {
    int err = 0;
    ctx = lv_create(deviceOrdinal, &err);
    if (!ctx) lastError = "lv_create failed (no HIP device; there is no CPU fallback)";
}

StreamlineTracingGrid::~StreamlineTracingGrid() {
    if (ctx) lv_destroy(ctx);
}

// StreamlineTracingGrid.cpp:81-116 without transposition / subsampling
void StreamlineTracingGrid::setGridExtent(int _xs, int _ys, int _zs, float _dx, float _dy, float _dz) {
    xs = _xs; ys = _ys; zs = _zs;
    dx = _dx; dy = _dy; dz = _dz;
    box.min = vec3(0.0f);
    box.max = vec3(float(xs - 1) * dx, float(ys - 1) * dy, float(zs - 1) * dz);
    vectorFields.clear();
    scalarFields.clear();
    gridDirty = true;
}

void StreamlineTracingGrid::addVectorField(const float* vectorField, const std::string& vectorName) {
    vectorFields[vectorName].assign(vectorField, vectorField + 3 * size_t(xs) * ys * zs);
    gridDirty = true;
}

void StreamlineTracingGrid::addScalarField(const float* scalarField, const std::string& scalarName) {
    scalarFields[scalarName].assign(scalarField, scalarField + size_t(xs) * ys * zs);
    gridDirty = true;
}

std::vector<std::string> StreamlineTracingGrid::getVectorFieldNames() {
    std::vector<std::string> names;
    for (auto& it : vectorFields) names.push_back(it.first);
    return names;
}

std::vector<std::string> StreamlineTracingGrid::getScalarFieldNames() {
    std::vector<std::string> names;
    for (auto& it : scalarFields) names.push_back(it.first);
    return names;
}

// _setVectorField (:220-234): the vectorFieldIndex-th field in name order
bool StreamlineTracingGrid::uploadGrid(int vectorFieldIndex) {
    if (!ctx) return false;
    if (vectorFields.empty()) { lastError = "no vector field"; return false; }
    vectorFieldIndex = std::min(std::max(vectorFieldIndex, 0), int(vectorFields.size()) - 1);
    if (!gridDirty && uploadedVectorFieldIndex == vectorFieldIndex) return true;
    auto it = vectorFields.begin();
    std::advance(it, vectorFieldIndex);
    std::vector<const float*> scalars;
    for (auto& s : scalarFields) scalars.push_back(s.second.data());
    int rc = lv_set_flow_grid(ctx, it->second.data(), uint32_t(xs), uint32_t(ys), uint32_t(zs), dx, dy, dz,
                              scalars.empty() ? nullptr : scalars.data(), uint32_t(scalars.size()));
    if (rc != LV_OK) { lastError = std::string("lv_set_flow_grid: ") + lv_last_error(ctx); return false; }
    uploadedVectorFieldIndex = vectorFieldIndex;
    gridDirty = false;
    return true;
}

bool StreamlineTracingGrid::traceStreamlines(const StreamlineTracingSettings& tracingSettings,
                                             const std::vector<vec3>& seedPoints, Trajectories& filteredTrajectories) {
    return traceLines(tracingSettings, seedPoints, filteredTrajectories, nullptr);
}

bool StreamlineTracingGrid::traceLines(const StreamlineTracingSettings& tracingSettings, const std::vector<vec3>& seedPoints,
                                       Trajectories& filteredTrajectories, std::vector<uint32_t>* seedIndices) {
    if (!uploadGrid(tracingSettings.vectorFieldIndex)) return false;
    lv_streamline_settings s;
    s.integration_method = uint32_t(tracingSettings.integrationMethod);
    s.integration_direction = uint32_t(tracingSettings.integrationDirection);
    s.time_step_scale = tracingSettings.timeStepScale;
    s.max_num_iterations = tracingSettings.maxNumIterations;
    s.termination_distance = tracingSettings.terminationDistance;
    s.minimum_length = tracingSettings.minimumLength;
    uint64_t numLines = 0, numPoints = 0;
    int rc = lv_trace_streamlines(ctx, seedPoints.empty() ? nullptr : &seedPoints[0].x, uint32_t(seedPoints.size()), &s,
                                  &numLines, &numPoints);
    if (rc != LV_OK) { lastError = std::string("lv_trace_streamlines: ") + lv_last_error(ctx); return false; }
    const size_t k = scalarFields.size();
    std::vector<float> positions(3 * numPoints), attributes(k * numPoints);
    std::vector<uint32_t> offsets(numLines + 1);
    rc = lv_get_streamlines(ctx, positions.data(), attributes.data(), offsets.data());
    if (rc != LV_OK) { lastError = std::string("lv_get_streamlines: ") + lv_last_error(ctx); return false; }
    if (seedIndices) {
        seedIndices->assign(numLines, 0u);
        if (numLines && lv_get_streamline_seed_indices(ctx, seedIndices->data()) != LV_OK) {
            lastError = std::string("lv_get_streamline_seed_indices: ") + lv_last_error(ctx);
            return false;
        }
    }
    for (uint64_t l = 0; l < numLines; l++) {
        Trajectory t;
        const uint32_t b = offsets[l], e = offsets[l + 1];
        t.positions.resize(e - b);
        memcpy(t.positions.data(), positions.data() + 3 * size_t(b), size_t(e - b) * 12);
        t.attributes.resize(k);
        for (size_t a = 0; a < k; a++)
            t.attributes[a].assign(attributes.begin() + a * numPoints + b, attributes.begin() + a * numPoints + e);
        filteredTrajectories.push_back(std::move(t));
    }
    return true;
}

bool StreamlineTracingGrid::traceStreamlinesDecreasingHelicity(const StreamlineTracingSettings& tracingSettings,
                                                               Trajectories& filteredTrajectories) {
    return traceLinesDecreasingHelicity(tracingSettings, filteredTrajectories, nullptr);
}

bool StreamlineTracingGrid::traceLinesDecreasingHelicity(const StreamlineTracingSettings& tracingSettings,
                                                         Trajectories& filteredTrajectories, std::vector<uint32_t>* seedIndices) {
    auto it = scalarFields.find("Helicity");
    if (it == scalarFields.end()) { lastError = "_traceStreamribbonsDecreasingHelicity: no helicity field was found"; return false; }
    if (!uploadGrid(tracingSettings.vectorFieldIndex)) return false;
    lv_streamline_settings s;
    s.integration_method = uint32_t(tracingSettings.integrationMethod);
    s.integration_direction = uint32_t(tracingSettings.integrationDirection);
    s.time_step_scale = tracingSettings.timeStepScale;
    s.max_num_iterations = tracingSettings.maxNumIterations;
    s.termination_distance = tracingSettings.terminationDistance;
    s.minimum_length = tracingSettings.minimumLength;
    lv_helicity_seeding_settings hs;
    hs.minimum_separation_distance = tracingSettings.minimumSeparationDistance;
    hs.termination_check_type = uint32_t(tracingSettings.terminationCheckType);
    hs.loop_check_mode = uint32_t(tracingSettings.loopCheckMode);
    hs.termination_distance_self = tracingSettings.terminationDistanceSelf;
    hs.seeding_subsampling_factor = tracingSettings.seedingSubsamplingFactor;
    uint64_t numLines = 0, numPoints = 0;
    int rc = lv_trace_streamlines_max_helicity_first(ctx, it->second.data(), &s, &hs, &numLines, &numPoints);
    if (rc != LV_OK) { lastError = std::string("lv_trace_streamlines_max_helicity_first: ") + lv_last_error(ctx); return false; }
    const size_t k = scalarFields.size();
    std::vector<float> positions(3 * numPoints), attributes(k * numPoints);
    std::vector<uint32_t> offsets(numLines + 1);
    rc = lv_get_streamlines(ctx, positions.data(), attributes.data(), offsets.data());
    if (rc != LV_OK) { lastError = std::string("lv_get_streamlines: ") + lv_last_error(ctx); return false; }
    if (seedIndices) {
        seedIndices->assign(numLines, 0u);
        if (numLines && lv_get_streamline_seed_indices(ctx, seedIndices->data()) != LV_OK) {
            lastError = std::string("lv_get_streamline_seed_indices: ") + lv_last_error(ctx);
            return false;
        }
    }
    for (uint64_t l = 0; l < numLines; l++) {
        Trajectory t;
        const uint32_t b = offsets[l], e = offsets[l + 1];
        t.positions.resize(e - b);
        memcpy(t.positions.data(), positions.data() + 3 * size_t(b), size_t(e - b) * 12);
        t.attributes.resize(k);
        for (size_t a = 0; a < k; a++)
            t.attributes[a].assign(attributes.begin() + a * numPoints + b, attributes.begin() + a * numPoints + e);
        filteredTrajectories.push_back(std::move(t));
    }
    return true;
}

// ---------------------------------------------------------------- streamribbons
// _getScalarFieldAtPosition (:863-913): trilinear, values outside the grid are 0
float StreamlineTracingGrid::getScalarFieldAtPosition(const std::vector<float>& f, const vec3& p) const {
    const vec3 q0 = p - box.min;
    const vec3 q(q0.x * (1.0f / dx), q0.y * (1.0f / dy), q0.z * (1.0f / dz));
    const int cx = int(q.x), cy = int(q.y), cz = int(q.z);
    const float fx = q.x - std::floor(q.x), fy = q.y - std::floor(q.y), fz = q.z - std::floor(q.z);
    const float ix = 1.0f - fx, iy = 1.0f - fy, iz = 1.0f - fz;
    auto at = [&](int x, int y, int z) {
        if (x < 0 || y < 0 || z < 0 || x >= xs || y >= ys || z >= zs) return 0.0f;
        return f[size_t(x) + size_t(y) * xs + size_t(z) * xs * ys];
    };
    float r = (ix * iy * iz) * at(cx, cy, cz);
    r = r + (fx * iy * iz) * at(cx + 1, cy, cz);
    r = r + (ix * fy * iz) * at(cx, cy + 1, cz);
    r = r + (fx * fy * iz) * at(cx + 1, cy + 1, cz);
    r = r + (ix * iy * fz) * at(cx, cy, cz + 1);
    r = r + (fx * iy * fz) * at(cx + 1, cy, cz + 1);
    r = r + (ix * fy * fz) * at(cx, cy + 1, cz + 1);
    r = r + (fx * fy * fz) * at(cx + 1, cy + 1, cz + 1);
    return r;
}

namespace {
// glm::rotate(v, angle, normal) (gtx/rotate_vector.inl): the axis-angle matrix of gtc/matrix_transform.inl applied to v
inline vec3 rotateVector(vec3 v, float angle, vec3 normal) {
    const float c = std::cos(angle), s = std::sin(angle);
    const vec3 axis = normalize(normal);
    const vec3 temp = (1.0f - c) * axis;
    const vec3 c0(c + temp.x * axis.x, temp.x * axis.y + s * axis.z, temp.x * axis.z - s * axis.y);
    const vec3 c1(temp.y * axis.x - s * axis.z, c + temp.y * axis.y, temp.y * axis.z + s * axis.x);
    const vec3 c2(temp.z * axis.x + s * axis.y, temp.z * axis.y - s * axis.x, c + temp.z * axis.z);
    return (c0 * v.x + c1 * v.y) + c2 * v.z;
}
} // namespace

// _pushRibbonDirections (StreamlineTracingGrid.cpp:1049-1116; positions in TRACE order: outwards from the seed) in two passes with
// the same result, bit for bit: everything that depends on the positions alone -- unit tangents, the helicity twist angle of every
// point (field lookup, segment length) -- is computed for all points first; what remains sequential is the carry itself: the ribbon
// direction of a point is the previous one made perpendicular to this point's tangent (Gram-Schmidt, fallback axes z then y) and
// turned about the tangent by this point's twist angle.
void StreamlineTracingGrid::pushRibbonDirections(const StreamlineTracingSettings& tracingSettings,
                                                 const std::vector<float>& helicityField, float maxHelicityMagnitude,
                                                 const vec3* positions, size_t n, std::vector<vec3>& ribbonDirections,
                                                 bool forwardMode) const {
    vec3 carried = normalize(tracingSettings.initialRibbonDirection);
    if (n == 1) { ribbonDirections.push_back(carried); return; }
    // ---- pass 1: per-point quantities (no dependence between points)
    std::vector<vec3> unitTangent(n);
    std::vector<float> twist(tracingSettings.useHelicity ? n : 0);
    for (size_t i = 0; i < n; i++) {
        const vec3& ahead = positions[i + 1 < n ? i + 1 : i];
        const vec3& behind = positions[i > 0 ? i - 1 : i];
        unitTangent[i] = normalize(ahead - behind);      // one-sided at the two ends, central in between
        if (tracingSettings.useHelicity) {
            float helicity = getScalarFieldAtPosition(helicityField, positions[i]);
            if (!forwardMode) helicity *= -1.0f;
            const float step = i + 1 < n ? length(positions[i + 1] - positions[i]) : 0.0f;
            twist[i] = helicity / maxHelicityMagnitude * 3.14159265358979323846f * tracingSettings.maxHelicityTwist * step / 0.005f;
        }
    }
    // ---- pass 2: the carry
    const size_t first = ribbonDirections.size();
    ribbonDirections.resize(first + n);
    for (size_t i = 0; i < n; i++) {
        const vec3 t = unitTangent[i];
        vec3 axis = carried;
        if (length(cross(axis, t)) < 1e-2f) {            // tangent (anti)parallel to the carried direction
            axis = vec3(0.0f, 0.0f, 1.0f);
            if (length(cross(axis, t)) < 1e-2f) axis = vec3(0.0f, 1.0f, 0.0f);
        }
        carried = normalize(axis - dot(axis, t) * t);    // Gram-Schmidt
        if (tracingSettings.useHelicity) carried = rotateVector(carried, twist[i], t);
        ribbonDirections[first + i] = carried;
    }
}

bool StreamlineTracingGrid::traceStreamribbons(const StreamlineTracingSettings& tracingSettings,
                                               const std::vector<vec3>& seedPoints, Trajectories& filteredTrajectories,
                                               std::vector<std::vector<vec3>>& filteredRibbonsDirections) {
    auto hel = scalarFields.find("Helicity");
    if (tracingSettings.useHelicity && hel == scalarFields.end()) {
        lastError = "traceStreamribbons: no scalar field named \"Helicity\" (StreamlineTracingGrid.cpp:299-320)";
        return false;
    }
    static const std::vector<float> none;
    const std::vector<float>& helicityField = tracingSettings.useHelicity ? hel->second : none;
    float maxHelicityMagnitude = 0.0f; // addScalarField("Helicity"), :299-320
    for (float h : helicityField) maxHelicityMagnitude = std::max(maxHelicityMagnitude, std::fabs(h));
    const size_t first = filteredTrajectories.size();
    std::vector<uint32_t> seedIndices;
    if (!traceLines(tracingSettings, seedPoints, filteredTrajectories, &seedIndices)) return false;
    pushRibbonsOfLines(tracingSettings, helicityField, maxHelicityMagnitude, first, seedIndices, filteredTrajectories,
                       filteredRibbonsDirections, false);
    return true;
}

// _traceStreamribbonsDecreasingHelicity with flowPrimitives == STREAMRIBBONS (:783-823): the max-helicity-first lines + their ribbon
// directions; the reference pushes them with forwardMode = true for the backward part as well
bool StreamlineTracingGrid::traceStreamribbonsDecreasingHelicity(const StreamlineTracingSettings& tracingSettings,
                                                                 Trajectories& filteredTrajectories,
                                                                 std::vector<std::vector<vec3>>& filteredRibbonsDirections) {
    auto hel = scalarFields.find("Helicity");
    if (hel == scalarFields.end()) { lastError = "_traceStreamribbonsDecreasingHelicity: no helicity field was found"; return false; }
    static const std::vector<float> none;
    const std::vector<float>& helicityField = tracingSettings.useHelicity ? hel->second : none;
    float maxHelicityMagnitude = 0.0f;
    for (float h : hel->second) maxHelicityMagnitude = std::max(maxHelicityMagnitude, std::fabs(h));
    const size_t first = filteredTrajectories.size();
    std::vector<uint32_t> seedIndices;
    if (!traceLinesDecreasingHelicity(tracingSettings, filteredTrajectories, &seedIndices)) return false;
    pushRibbonsOfLines(tracingSettings, helicityField, maxHelicityMagnitude, first, seedIndices, filteredTrajectories,
                       filteredRibbonsDirections, true);
    return true;
}

// ribbon directions of merged lines: carried outwards from the seed in each traced part; backwardPartForwardMode = the forwardMode
// argument of _pushRibbonDirections for the backward part (false in traceStreamribbons, true in the decreasing-helicity tracer)
void StreamlineTracingGrid::pushRibbonsOfLines(const StreamlineTracingSettings& tracingSettings, const std::vector<float>& helicityField,
                                               float maxHelicityMagnitude, size_t first, const std::vector<uint32_t>& seedIndices,
                                               const Trajectories& filteredTrajectories,
                                               std::vector<std::vector<vec3>>& filteredRibbonsDirections,
                                               bool backwardPartForwardMode) const {
    filteredRibbonsDirections.resize(filteredTrajectories.size());
    // The GPU returns the merged lines; the ribbon directions are carried outwards from the seed in each traced part
    // (forward part as traced, backward part in its own trace order with the helicity's sign flipped, then reversed and put
    // in front without its seed point: :479-486, _reverseRibbon, _insertBackwardRibbon).
#pragma omp parallel for schedule(dynamic, 16)
    for (long li = 0; li < long(seedIndices.size()); li++) {
        const Trajectory& t = filteredTrajectories[first + size_t(li)];
        std::vector<vec3>& out = filteredRibbonsDirections[first + size_t(li)];
        const size_t n = t.positions.size(), s = seedIndices[size_t(li)];
        if (tracingSettings.integrationDirection == StreamlineIntegrationDirection::FORWARD) {
            pushRibbonDirections(tracingSettings, helicityField, maxHelicityMagnitude, t.positions.data(), n, out, true);
        } else if (tracingSettings.integrationDirection == StreamlineIntegrationDirection::BACKWARD) {
            std::vector<vec3> traced(t.positions.rbegin(), t.positions.rend());
            if (n <= 1) traced.assign(t.positions.begin(), t.positions.end());
            pushRibbonDirections(tracingSettings, helicityField, maxHelicityMagnitude, traced.data(), n, out, backwardPartForwardMode);
            if (n > 1) std::reverse(out.begin(), out.end());
        } else {
            std::vector<vec3> fwd, bwd;
            pushRibbonDirections(tracingSettings, helicityField, maxHelicityMagnitude, t.positions.data() + s, n - s, fwd, true);
            if (s > 0) {
                std::vector<vec3> traced(t.positions.rend() - ptrdiff_t(s + 1), t.positions.rend()); // seed, then outwards
                pushRibbonDirections(tracingSettings, helicityField, maxHelicityMagnitude, traced.data(), s + 1, bwd, backwardPartForwardMode);
                std::reverse(bwd.begin(), bwd.end());
                out.assign(bwd.begin(), bwd.end() - 1);
            }
            out.insert(out.end(), fwd.begin(), fwd.end());
        }
    }
}

// ---------------------------------------------------------------- grid utilities, Loader/GridLoader.cpp:41-183
void computeVectorMagnitudeField(const float* v, float* out, int xs, int ys, int zs) {
    const size_t n = size_t(xs) * ys * zs;
#pragma omp parallel for
    for (long i = 0; i < long(n); i++) {
        const float vx = v[3 * i], vy = v[3 * i + 1], vz = v[3 * i + 2];
        out[i] = std::sqrt(vx * vx + vy * vy + vz * vz);
    }
}
// every derivative is divided by dy, as the reference writes it (:81-98)
void computeVorticityField(const float* v, float* out, int xs, int ys, int zs, float /*dx*/, float dy, float /*dz*/) {
    auto V = [&](int x, int y, int z, int c) { return v[3 * (size_t(x) + size_t(y) * xs + size_t(z) * xs * ys) + c]; };
#pragma omp parallel for
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
void computeHelicityFieldNormalized(const float* vel, const float* vort, float* out, int xs, int ys, int zs,
                                    bool normalizeVelocity, bool normalizeVorticity) {
    const size_t n = size_t(xs) * ys * zs;
#pragma omp parallel for
    for (long i = 0; i < long(n); i++) {
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


// ---------------------------------------------------------------- StreamlinePlaneSeeder, StreamlineSeeder.cpp:52-135
void StreamlinePlaneSeeder::reset(const StreamlineTracingGrid& grid) {
    box = grid.getBox();
    generator = std::mt19937(seed);
    const vec3 dim = box.getDimensions();
    maxDimension = std::max(dim.x, std::max(dim.y, dim.z));
    const vec3 mn = box.min, mx = box.max, center = box.getCenter();
    float minOffset = 3.402823466e+38f, maxOffset = -3.402823466e+38f;
    for (int c = 0; c < 8; c++) {
        const vec3 pt((c & 1) ? mx.x : mn.x, (c & 2) ? mx.y : mn.y, (c & 4) ? mx.z : mn.z);
        const float offset = dot(planeNormal, pt - center);
        minOffset = std::min(minOffset, offset);
        maxOffset = std::max(maxOffset, offset);
    }
    planeOffset = minOffset + (maxOffset - minOffset) * planeSlice;
    axis0 = vec3(1.0f, 0.0f, 0.0f);
    axis1 = cross(axis0, planeNormal);
    if (length(axis1) < 1e-3f) {
        axis0 = vec3(0.0f, 1.0f, 0.0f);
        axis1 = normalize(cross(axis0, planeNormal));
    } else {
        axis1 = normalize(axis1);
    }
    axis0 = cross(planeNormal, axis1);
    currentSampleIdx = 0;
}

bool StreamlinePlaneSeeder::hasNextPoint() const {
    return currentSampleIdx < (regular ? numSamplesX * numSamplesY : numSamplesRandom);
}

vec3 StreamlinePlaneSeeder::getNextPoint() {
    if (regular) {
        const int y = currentSampleIdx / numSamplesX, x = currentSampleIdx % numSamplesX;
        currentSampleIdx++;
        const float dx = 1.0f / float(numSamplesX + 1), dy = 1.0f / float(numSamplesY + 1);
        vec3 p = box.getCenter() + planeNormal * planeOffset;
        p = p + axis0 * maxDimension * dx * (float(x) - float(numSamplesX - 1) / 2.0f);
        p = p + axis1 * maxDimension * dy * (float(y) - float(numSamplesY - 1) / 2.0f);
        return p;
    }
    currentSampleIdx++;
    for (int it = 0; it < 100; it++) { // (the reference scales the random offsets by 1 / maxDimension, :126-129)
        const float r0 = uniformDistribution(generator), r1 = uniformDistribution(generator);
        vec3 p = box.getCenter() + planeNormal * planeOffset;
        const float dx = 1.0f / maxDimension, dy = 1.0f / maxDimension;
        p = p + axis0 * dx * (r0 - 0.5f);
        p = p + axis1 * dy * (r1 - 0.5f);
        if (p.x >= box.min.x && p.y >= box.min.y && p.z >= box.min.z && p.x <= box.max.x && p.y <= box.max.y && p.z <= box.max.z)
            return p;
    }
    return box.getCenter(); // fallback, :138
}

// ---------------------------------------------------------------- StreamlineVolumeSeeder, StreamlineSeeder.cpp:259-300
void StreamlineVolumeSeeder::setRegular(int nx, int ny, int nz) {
    regular = true;
    numSamplesX = nx; numSamplesY = ny; numSamplesZ = nz;
}

void StreamlineVolumeSeeder::setRandom(uint32_t s) {
    regular = false;
    seed = s;
}

void StreamlineVolumeSeeder::reset(const StreamlineTracingGrid& grid) {
    box = grid.getBox();
    vec3 dim = box.getDimensions();
    maxDimension = std::max(dim.x, std::max(dim.y, dim.z));
    currentSampleIdx = 0;
    generator.seed(seed);
}

vec3 StreamlineVolumeSeeder::getNextPoint() {
    if (regular) {
        int z = currentSampleIdx / (numSamplesX * numSamplesY);
        int xy = currentSampleIdx % (numSamplesX * numSamplesY);
        int y = xy / numSamplesX;
        int x = xy % numSamplesX;
        currentSampleIdx++;
        vec3 dimensions = box.getDimensions();
        float sx = 1.0f / float(numSamplesX + 1), sy = 1.0f / float(numSamplesY + 1), sz = 1.0f / float(numSamplesZ + 1);
        return vec3(box.min.x + dimensions.x * sx * float(x + 1), box.min.y + dimensions.y * sy * float(y + 1),
                    box.min.z + dimensions.z * sz * float(z + 1));
    }
    auto uniform = [&]() { return float(double(generator()) / 4294967296.0); };
    for (int it = 0; it < 100; it++) {
        float r0 = uniform() * maxDimension, r1 = uniform() * maxDimension, r2 = uniform() * maxDimension;
        vec3 p = vec3(r0, r1, r2) + box.min;
        if (p.x >= box.min.x && p.y >= box.min.y && p.z >= box.min.z && p.x <= box.max.x && p.y <= box.max.y && p.z <= box.max.z)
            return p;
    }
    return box.getCenter();
}

// ---------------------------------------------------------------- AbcFlowGenerator, Loader/AbcFlowGenerator.cpp:41-103
AbcFlowGenerator::AbcFlowGenerator() {
    A = std::sqrt(3.0f);
    B = std::sqrt(2.0f);
    C = 1.0f;
}

void AbcFlowGenerator::generateAbcFlow(float* v) const {
#pragma omp parallel for
    for (int iz = 0; iz < zs; iz++)
        for (int iy = 0; iy < ys; iy++)
            for (int ix = 0; ix < xs; ix++) {
                float x = float(ix) / float(xs - 1) * resScale;
                float y = float(iy) / float(ys - 1) * resScale;
                float z = float(iz) / float(zs - 1) * resScale;
                size_t o = size_t(iz) * xs * ys * 3 + size_t(iy) * xs * 3 + size_t(ix) * 3;
                v[o + 0] = A * std::sin(z) + C * std::cos(y);
                v[o + 1] = B * std::sin(x) + A * std::cos(z);
                v[o + 2] = C * std::sin(y) + B * std::cos(x);
            }
}

void AbcFlowGenerator::load(StreamlineTracingGrid* grid, bool useNormalizedVelocity, bool useNormalizedVorticity) const {
    float maxDimension = float(std::max(xs - 1, std::max(ys - 1, zs - 1)));
    float cellStep = 1.0f / maxDimension;
    const size_t n = size_t(xs) * ys * zs;
    std::vector<float> velocity(3 * n), velocityMagnitude(n), vorticity(3 * n), vorticityMagnitude(n), helicity(n);
    generateAbcFlow(velocity.data());
    computeVectorMagnitudeField(velocity.data(), velocityMagnitude.data(), xs, ys, zs);
    computeVorticityField(velocity.data(), vorticity.data(), xs, ys, zs, cellStep, cellStep, cellStep);
    computeVectorMagnitudeField(vorticity.data(), vorticityMagnitude.data(), xs, ys, zs);
    computeHelicityFieldNormalized(velocity.data(), vorticity.data(), helicity.data(), xs, ys, zs, useNormalizedVelocity,
                                   useNormalizedVorticity);
    grid->setGridExtent(xs, ys, zs, cellStep, cellStep, cellStep);
    grid->addVectorField(velocity.data(), "Velocity");
    grid->addVectorField(vorticity.data(), "Vorticity");
    grid->addScalarField(helicity.data(), "Helicity");
    grid->addScalarField(velocityMagnitude.data(), "Velocity Magnitude");
    grid->addScalarField(vorticityMagnitude.data(), "Vorticity Magnitude");
}

} // namespace lv
