// Flow.hpp -- headless mirror of the reference's streamline tracer front end (src/LineData/Flow): the classes an embedder
// drives to turn a vector field on a regular grid into the trajectories the renderers consume.
//
//   StreamlineTracingSettings        StreamlineTracingDefines.hpp:144-177 (the members the line tracer reads)
//   StreamlineTracingGrid            StreamlineTracingGrid.hpp:51-76: setGridExtent / addVectorField / addScalarField /
//                                    traceStreamlines; integration runs on the GPU through lv_trace_streamlines
//   StreamlineVolumeSeeder           StreamlineSeeder.cpp:259-300 (regular grid of seeds; the random mode draws from
//                                    std::uniform_real_distribution, whose sequence is implementation defined -- here a
//                                    std::mt19937 mapped to [0,1) by x / 2^32, owned by the build)
//   AbcFlowGenerator                 Loader/AbcFlowGenerator.{hpp,cpp}: analytic ABC flow sampled to a grid
#pragma once

#include <map>
#include <random>
#include <string>
#include <vector>

#include "LineData.hpp"

namespace lv {

enum class StreamlineIntegrationMethod { EXPLICIT_EULER, IMPLICIT_EULER, HEUN, MIDPOINT, RK4, RKF45 };
enum class StreamlineIntegrationDirection { FORWARD, BACKWARD, BOTH };
extern const char* const STREAMLINE_INTEGRATION_METHOD_NAMES[6];
extern const char* const STREAMLINE_INTEGRATION_DIRECTION_NAMES[3];

struct StreamlineTracingSettings {
    int numPrimitives = 1024;
    float timeStepScale = 1.0f;
    int maxNumIterations = 2000;
    float terminationDistance = 1.0f;
    float minimumLength = 0.7f;
    StreamlineIntegrationMethod integrationMethod = StreamlineIntegrationMethod::RK4;
    StreamlineIntegrationDirection integrationDirection = StreamlineIntegrationDirection::BOTH;
    int vectorFieldIndex = 0;
    // flowPrimitives == STREAMRIBBONS (StreamlineTracingDefines.hpp:168-173)
    bool useHelicity = true;
    float maxHelicityTwist = 0.25f;
    vec3 initialRibbonDirection = vec3(0.0f, 1.0f, 0.0f);
    // StreamlineMaxHelicityFirstSeeder (StreamlineTracingDefines.hpp:89-101,156-174): termination check 1 = grid-based and loop check
    // 0 = none / 1 = start point are built
    float minimumSeparationDistance = 0.08f;
    int terminationCheckType = 1;
    int loopCheckMode = 1;
    float terminationDistanceSelf = 1.0f;
    int seedingSubsamplingFactor = 1;
};

/// GridLoader.cpp:41-183: |v| per grid point, curl by central / one-sided differences, helicity v . curl v
void computeVectorMagnitudeField(const float* vectorField, float* vectorMagnitudeField, int xs, int ys, int zs);
void computeVorticityField(const float* velocityField, float* vorticityField, int xs, int ys, int zs, float dx, float dy, float dz);
void computeHelicityFieldNormalized(const float* velocityField, const float* vorticityField, float* helicityField, int xs, int ys,
                                    int zs, bool normalizeVelocity, bool normalizeVorticity);

class StreamlineTracingGrid {
public:
    explicit StreamlineTracingGrid(int deviceOrdinal = 0);
    ~StreamlineTracingGrid();
    bool isValid() const { return ctx != nullptr; }
    const std::string& getLastError() const { return lastError; }

    void setGridExtent(int xs, int ys, int zs, float dx, float dy, float dz);
    /// The fields are copied (the reference takes ownership of the caller's new[] arrays).
    void addVectorField(const float* vectorField, const std::string& vectorName);
    void addScalarField(const float* scalarField, const std::string& scalarName);
    std::vector<std::string> getVectorFieldNames();
    std::vector<std::string> getScalarFieldNames();
    const AABB3& getBox() const { return box; }
    int getGridSizeX() const { return xs; }
    int getGridSizeY() const { return ys; }
    int getGridSizeZ() const { return zs; }
    float getDx() const { return dx; }
    float getDy() const { return dy; }
    float getDz() const { return dz; }

    /// traceStreamlines (StreamlineTracingGrid.cpp:344-426) for explicit seed points; attributes of the result are the
    /// scalar fields in name order (std::map iteration, like the reference's scalarFields).
    bool traceStreamlines(const StreamlineTracingSettings& tracingSettings, const std::vector<vec3>& seedPoints,
                          Trajectories& filteredTrajectories);
    /// traceStreamribbons (StreamlineTracingGrid.cpp:428-530): the streamlines (traced on the GPU) + one ribbon direction per
    /// point (_pushRibbonDirections, :1049-1116: carried outwards from the seed along every traced part, twisted by the
    /// "Helicity" scalar field) -- the band data LineDataFlow::setTrajectoryData takes.
    bool traceStreamribbons(const StreamlineTracingSettings& tracingSettings, const std::vector<vec3>& seedPoints,
                            Trajectories& filteredTrajectories, std::vector<std::vector<vec3>>& filteredRibbonsDirections);
    /// _traceStreamlinesDecreasingHelicity with the StreamlineMaxHelicityFirstSeeder (StreamlineTracingGrid.cpp:740-860,
    /// StreamlineSeeder.cpp:360-529): seeds in the order of falling helicity ("Helicity" scalar field), lines end where they come within
    /// minimumSeparationDistance of an earlier one.  Traced in speculative batches on the GPU, committed in seeding order
    /// (lv_trace_streamlines_max_helicity_first).
    bool traceStreamlinesDecreasingHelicity(const StreamlineTracingSettings& tracingSettings, Trajectories& filteredTrajectories);
    /// the same with flowPrimitives == STREAMRIBBONS: + one ribbon direction per point
    bool traceStreamribbonsDecreasingHelicity(const StreamlineTracingSettings& tracingSettings, Trajectories& filteredTrajectories,
                                              std::vector<std::vector<vec3>>& filteredRibbonsDirections);

private:
    bool uploadGrid(int vectorFieldIndex);
    bool traceLines(const StreamlineTracingSettings& tracingSettings, const std::vector<vec3>& seedPoints,
                    Trajectories& filteredTrajectories, std::vector<uint32_t>* seedIndices);
    bool traceLinesDecreasingHelicity(const StreamlineTracingSettings& tracingSettings, Trajectories& filteredTrajectories,
                                      std::vector<uint32_t>* seedIndices);
    void pushRibbonsOfLines(const StreamlineTracingSettings& tracingSettings, const std::vector<float>& helicityField,
                            float maxHelicityMagnitude, size_t first, const std::vector<uint32_t>& seedIndices,
                            const Trajectories& filteredTrajectories, std::vector<std::vector<vec3>>& filteredRibbonsDirections,
                            bool backwardPartForwardMode) const;
    float getScalarFieldAtPosition(const std::vector<float>& scalarField, const vec3& particlePosition) const;
    void pushRibbonDirections(const StreamlineTracingSettings& tracingSettings, const std::vector<float>& helicityField,
                              float maxHelicityMagnitude, const vec3* positions, size_t n, std::vector<vec3>& ribbonDirections,
                              bool forwardMode) const;
    lv_ctx* ctx = nullptr;
    std::string lastError;
    int xs = 0, ys = 0, zs = 0;
    float dx = 1.0f, dy = 1.0f, dz = 1.0f;
    AABB3 box;
    std::map<std::string, std::vector<float>> vectorFields, scalarFields;
    int uploadedVectorFieldIndex = -1;
    bool gridDirty = true;
};

class StreamlineVolumeSeeder {
public:
    void setRegular(int numSamplesX, int numSamplesY, int numSamplesZ);
    void setRandom(uint32_t seed);
    void reset(const StreamlineTracingGrid& grid);
    vec3 getNextPoint();

private:
    bool regular = true;
    int numSamplesX = 8, numSamplesY = 8, numSamplesZ = 8, currentSampleIdx = 0;
    AABB3 box;
    float maxDimension = 1.0f;
    uint32_t seed = 12345;
    std::mt19937 generator;
};

/// Seeds on a plane through the grid box (StreamlinePlaneSeeder, StreamlineSeeder.cpp:52-135): plane normal + slice in
/// [0, 1] between the box corners' extreme offsets; regular nx x ny pattern or uniform random points inside the box.
class StreamlinePlaneSeeder {
public:
    void setPlane(vec3 normal, float slice) { planeNormal = normal; planeSlice = slice; }
    void setRegular(int numSamplesX_, int numSamplesY_) { regular = true; numSamplesX = numSamplesX_; numSamplesY = numSamplesY_; }
    void setRandom(int numSamples, int seed_) { regular = false; numSamplesRandom = numSamples; seed = seed_; }
    void reset(const StreamlineTracingGrid& grid);
    bool hasNextPoint() const;
    vec3 getNextPoint();

private:
    AABB3 box;
    std::mt19937 generator;
    std::uniform_real_distribution<float> uniformDistribution = std::uniform_real_distribution<float>(0, 1);
    float maxDimension = 0.0f, planeSlice = 0.5f, planeOffset = 0.0f;
    vec3 planeNormal = vec3(0.0f, 1.0f, 0.0f), axis0, axis1;
    int seed = 2;                      // StreamlineSeeder.hpp:134
    bool regular = false;
    int currentSampleIdx = 0, numSamplesX = 32, numSamplesY = 32, numSamplesRandom = 1024;
};

class AbcFlowGenerator {
public:
    AbcFlowGenerator();
    void setGridSize(int xs_, int ys_, int zs_) { xs = xs_; ys = ys_; zs = zs_; }
    void setResolutionScale(float s) { resScale = s; }
    void setCoefficients(float a, float b, float c) { A = a; B = b; C = c; }
    int getGridSizeX() const { return xs; }
    int getGridSizeY() const { return ys; }
    int getGridSizeZ() const { return zs; }
    void generateAbcFlow(float* v) const;
    /// "Velocity" / "Vorticity" vector fields and "Helicity" / "Velocity Magnitude" / "Vorticity Magnitude" scalar fields on a
    /// grid whose longest axis spans [0, 1] (AbcFlowGenerator.cpp:74-103).
    void load(StreamlineTracingGrid* grid, bool useNormalizedVelocity = false, bool useNormalizedVorticity = false) const;

private:
    int xs = 64, ys = 64, zs = 64;
    float resScale = 6.0f;
    float A, B, C;
};

} // namespace lv
