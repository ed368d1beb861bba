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

void AbcFlowGenerator::load(StreamlineTracingGrid* grid) const {
    float maxDimension = float(std::max(xs - 1, std::max(ys - 1, zs - 1)));
    float cellStep = 1.0f / maxDimension;
    std::vector<float> velocity(3 * size_t(xs) * ys * zs), magnitude(size_t(xs) * ys * zs);
    generateAbcFlow(velocity.data());
    for (size_t i = 0; i < magnitude.size(); i++) {
        float vx = velocity[3 * i], vy = velocity[3 * i + 1], vz = velocity[3 * i + 2];
        magnitude[i] = std::sqrt(vx * vx + vy * vy + vz * vz);
    }
    grid->setGridExtent(xs, ys, zs, cellStep, cellStep, cellStep);
    grid->addVectorField(velocity.data(), "Velocity");
    grid->addScalarField(magnitude.data(), "Velocity Magnitude");
}

} // namespace lv
