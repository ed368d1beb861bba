// host_capi.cpp -- flat C entry points over the C++ host layer so that the Python test-suite and bench.py can
// drive the same classes a C++ embedder would use (LineDataFlow, HipRayTracer, HipPerPixelLinkedListLineRenderer,
// HeadlessLineRenderer).  Not part of the drop-in boundary (that is include/linevis_hip.h).
#include <cstring>

#include "Flow.hpp"
#include "HeadlessLineRenderer.hpp"

using namespace lv;

namespace {
struct FlowHandle {
    LineDataPtr data;
    LineDataFlow* flow() { return static_cast<LineDataFlow*>(data.get()); }
    TubeAabbRenderData lastRenderData;
    TubeTriangleRenderData lastTriangleData;
};
} // namespace

extern "C" {

void lvh_normalize_positions(float* positions, uint64_t n) {
    Trajectories t(1);
    t[0].positions.resize(n);
    memcpy(t[0].positions.data(), positions, size_t(n) * 12);
    normalizeTrajectoriesVertexPositions(t, computeTrajectoriesAABB3(t));
    memcpy(positions, t[0].positions.data(), size_t(n) * 12);
}

void* lvh_flow_create() {
    FlowHandle* h = new FlowHandle();
    h->data = std::make_shared<LineDataFlow>();
    return h;
}
void lvh_flow_destroy(void* h) { delete static_cast<FlowHandle*>(h); }

void lvh_flow_set_trajectories(void* hp, const float* positions, const float* attributes, const uint32_t* lineOffsets,
                               uint32_t nLines) {
    FlowHandle* h = static_cast<FlowHandle*>(hp);
    Trajectories tr(nLines);
    for (uint32_t i = 0; i < nLines; i++) {
        uint32_t b = lineOffsets[i], e = lineOffsets[i + 1];
        tr[i].positions.resize(e - b);
        memcpy(tr[i].positions.data(), positions + 3 * size_t(b), size_t(e - b) * 12);
        tr[i].attributes.resize(1);
        tr[i].attributes[0].assign(attributes + b, attributes + e);
    }
    h->flow()->setTrajectoryData(tr);
}
/// as above with band data: one ribbon direction (3 floats) per input point
void lvh_flow_set_trajectories_ribbons(void* hp, const float* positions, const float* attributes, const uint32_t* lineOffsets,
                                       uint32_t nLines, const float* ribbonDirections) {
    FlowHandle* h = static_cast<FlowHandle*>(hp);
    Trajectories tr(nLines);
    std::vector<std::vector<vec3>> ribbons(nLines);
    for (uint32_t i = 0; i < nLines; i++) {
        uint32_t b = lineOffsets[i], e = lineOffsets[i + 1];
        tr[i].positions.resize(e - b);
        memcpy(tr[i].positions.data(), positions + 3 * size_t(b), size_t(e - b) * 12);
        tr[i].attributes.resize(1);
        tr[i].attributes[0].assign(attributes + b, attributes + e);
        ribbons[i].resize(e - b);
        memcpy(ribbons[i].data(), ribbonDirections + 3 * size_t(b), size_t(e - b) * 12);
    }
    h->flow()->setTrajectoryData(tr, {}, ribbons);
}
/// several attributes per point (attributes: nAttributes arrays of n floats, one after the other) with their names -- an attribute
/// whose name contains "helicity" enables the rotating helicity bands (LineDataFlow.cpp:535-550); ribbonDirections may be NULL
void lvh_flow_set_trajectories_multi(void* hp, const float* positions, const float* attributes, uint32_t nAttributes,
                                     const char* const* names, const uint32_t* lineOffsets, uint32_t nLines,
                                     const float* ribbonDirections) {
    FlowHandle* h = static_cast<FlowHandle*>(hp);
    Trajectories tr(nLines);
    std::vector<std::vector<vec3>> ribbons(ribbonDirections ? nLines : 0);
    const size_t n = lineOffsets[nLines];
    for (uint32_t i = 0; i < nLines; i++) {
        uint32_t b = lineOffsets[i], e = lineOffsets[i + 1];
        tr[i].positions.resize(e - b);
        memcpy(tr[i].positions.data(), positions + 3 * size_t(b), size_t(e - b) * 12);
        tr[i].attributes.resize(nAttributes);
        for (uint32_t a = 0; a < nAttributes; a++) tr[i].attributes[a].assign(attributes + a * n + b, attributes + a * n + e);
        if (ribbonDirections) {
            ribbons[i].resize(e - b);
            memcpy(ribbons[i].data(), ribbonDirections + 3 * size_t(b), size_t(e - b) * 12);
        }
    }
    std::vector<std::string> nm;
    for (uint32_t a = 0; a < nAttributes; a++) nm.push_back(names[a]);
    h->flow()->setTrajectoryData(tr, nm, ribbons);
}
void lvh_flow_set_selected_attribute(void* hp, int idx) { static_cast<FlowHandle*>(hp)->data->setSelectedAttributeIndex(idx); }
/// LineDataFlow::setNewSettings on the data set alone (the renderer's harness forwards its settings map too)
void lvh_flow_set_settings(void* hp, const char* const* keys, const char* const* values, uint32_t n) {
    SettingsMap m;
    for (uint32_t i = 0; i < n; i++) m.addKeyValue(std::string(keys[i]), values[i]);
    static_cast<FlowHandle*>(hp)->data->setNewSettings(m);
}
int lvh_flow_has_helicity(void* hp) { return static_cast<FlowHandle*>(hp)->flow()->getHasHelicity() ? 1 : 0; }
float lvh_flow_max_helicity(void* hp) { return static_cast<FlowHandle*>(hp)->flow()->getMaxHelicity(); }
int lvh_flow_use_rotating_helicity_bands(void* hp) { return static_cast<FlowHandle*>(hp)->flow()->getUseRotatingHelicityBands() ? 1 : 0; }
void lvh_flow_set_twist_line_texture(void* hp, const uint8_t* rgba8, uint32_t w, uint32_t h) {
    static_cast<FlowHandle*>(hp)->flow()->setTwistLineTexture(rgba8, w, h);
}
int lvh_flow_has_bands_data(void* hp) { return static_cast<FlowHandle*>(hp)->flow()->getHasBandsData() ? 1 : 0; }
/// ribbon directions flattened like lvh_flow_get_trajectories' positions (n * 3 floats); no band data: nothing written
void lvh_flow_get_ribbon_directions(void* hp, float* out) {
    FlowHandle* h = static_cast<FlowHandle*>(hp);
    size_t off = 0;
    for (const auto& dirs : h->flow()->getRibbonsDirections()) {
        memcpy(out + 3 * off, dirs.data(), dirs.size() * 12);
        off += dirs.size();
    }
}
/// getLinePassTubeAabbRenderData(false, true): elliptic tubes of the band data at the given band width
void lvh_flow_build_render_data_elliptic(void* hp, float bandWidth, uint32_t* outNumPoints, uint32_t* outNumSegments) {
    FlowHandle* h = static_cast<FlowHandle*>(hp);
    LineRenderer::setBandWidth(bandWidth);
    h->lastRenderData = h->flow()->getLinePassTubeAabbRenderData(false, true);
    *outNumPoints = uint32_t(h->lastRenderData.linePointDataBuffer.size());
    *outNumSegments = uint32_t(h->lastRenderData.indexBuffer.size() / 2);
}
int lvh_flow_load_binlines(void* hp, const char* path) { return static_cast<FlowHandle*>(hp)->flow()->loadFromFile(path) ? 0 : -1; }
int lvh_flow_save_binlines(void* hp, const char* path) {
    LineDataFlow* fl = static_cast<FlowHandle*>(hp)->flow();
    return saveTrajectoriesAsBinLines(path, fl->getTrajectories(), fl->getRibbonsDirections(), fl->getVerticesNormalized()) ? 0 : -1;
}
/// the verticesNormalized flag of version-2 .binlines files (read: honoured by loadFromFile; written from this state)
void lvh_flow_set_vertices_normalized(void* hp, int flag) { static_cast<FlowHandle*>(hp)->flow()->setVerticesNormalized(flag != 0); }
int lvh_flow_vertices_normalized(void* hp) { return static_cast<FlowHandle*>(hp)->flow()->getVerticesNormalized() ? 1 : 0; }
uint64_t lvh_flow_num_lines(void* hp) { return static_cast<FlowHandle*>(hp)->flow()->getNumLines(); }
uint64_t lvh_flow_num_points(void* hp) { return static_cast<FlowHandle*>(hp)->flow()->getNumLinePoints(); }
void lvh_flow_attribute_range(void* hp, float* out2) { static_cast<FlowHandle*>(hp)->data->getMinMaxAttributeValues(out2[0], out2[1]); }
void lvh_flow_bounding_box(void* hp, float* out6) {
    const AABB3& b = static_cast<FlowHandle*>(hp)->data->getModelBoundingBox();
    out6[0] = b.min.x; out6[1] = b.min.y; out6[2] = b.min.z; out6[3] = b.max.x; out6[4] = b.max.y; out6[5] = b.max.z;
}
/// Flattened copy of the trajectories (positions n*3, attribute 0, offsets nLines+1); NULL pointers = query sizes only.
void lvh_flow_get_trajectories(void* hp, float* positions, float* attributes, uint32_t* lineOffsets) {
    const Trajectories& tr = static_cast<FlowHandle*>(hp)->flow()->getTrajectories();
    uint32_t off = 0;
    for (size_t i = 0; i < tr.size(); i++) {
        if (lineOffsets) lineOffsets[i] = off;
        uint32_t n = uint32_t(tr[i].positions.size());
        if (positions) memcpy(positions + 3 * size_t(off), tr[i].positions.data(), size_t(n) * 12);
        if (attributes && !tr[i].attributes.empty()) memcpy(attributes + off, tr[i].attributes[0].data(), size_t(n) * 4);
        off += n;
    }
    if (lineOffsets) lineOffsets[tr.size()] = off;
}

/// getLinePassTubeAabbRenderData at the given line width; returns the counts, data fetched with lvh_flow_copy_render_data.
void lvh_flow_build_render_data(void* hp, float lineWidth, uint32_t* outNumPoints, uint32_t* outNumSegments) {
    FlowHandle* h = static_cast<FlowHandle*>(hp);
    LineRenderer::setLineWidth(lineWidth);
    h->lastRenderData = h->flow()->getLinePassTubeAabbRenderData(false, false);
    *outNumPoints = uint32_t(h->lastRenderData.linePointDataBuffer.size());
    *outNumSegments = uint32_t(h->lastRenderData.indexBuffer.size() / 2);
}
void lvh_flow_copy_render_data(void* hp, lv_line_point* points, uint32_t* segIndices, float* aabbs) {
    FlowHandle* h = static_cast<FlowHandle*>(hp);
    const TubeAabbRenderData& d = h->lastRenderData;
    if (points) memcpy(points, d.linePointDataBuffer.data(), d.linePointDataBuffer.size() * sizeof(lv_line_point));
    if (segIndices) memcpy(segIndices, d.indexBuffer.data(), d.indexBuffer.size() * 4);
    if (aabbs) memcpy(aabbs, d.aabbBuffer.data(), d.aabbBuffer.size() * 24);
}

/// getLinePassTubeTriangleMeshRenderData at the given line width / subdivisions; counts first, then the copy.
void lvh_flow_build_triangle_data(void* hp, float lineWidth, uint32_t numSubdivisions, uint64_t* outNumIndices,
                                  uint64_t* outNumVertices, uint64_t* outNumPoints) {
    FlowHandle* h = static_cast<FlowHandle*>(hp);
    LineRenderer::setLineWidth(lineWidth);
    SettingsMap m;
    m.addKeyValue("tube_num_subdivisions", int(numSubdivisions));
    h->data->setNewSettings(m);
    h->lastTriangleData = h->flow()->getLinePassTubeTriangleMeshRenderData(false, true);
    *outNumIndices = h->lastTriangleData.indexBuffer.size();
    *outNumVertices = h->lastTriangleData.vertexBuffer.size();
    *outNumPoints = h->lastTriangleData.linePointDataBuffer.size();
}
/// the same for a band data set: the elliptic tessellation at the given band width / minimum band thickness
void lvh_flow_build_triangle_data_bands(void* hp, float bandWidth, float minBandThickness, uint32_t numSubdivisions,
                                        uint64_t* outNumIndices, uint64_t* outNumVertices, uint64_t* outNumPoints) {
    FlowHandle* h = static_cast<FlowHandle*>(hp);
    LineRenderer::setBandWidth(bandWidth);
    SettingsMap m;
    m.addKeyValue("tube_num_subdivisions", int(numSubdivisions));
    m.addKeyValue("min_band_thickness", minBandThickness);
    m.addKeyValue("use_ribbons", true);
    h->data->setNewSettings(m);
    h->lastTriangleData = h->flow()->getLinePassTubeTriangleMeshRenderData(false, true);
    *outNumIndices = h->lastTriangleData.indexBuffer.size();
    *outNumVertices = h->lastTriangleData.vertexBuffer.size();
    *outNumPoints = h->lastTriangleData.linePointDataBuffer.size();
}
void lvh_flow_copy_triangle_data(void* hp, uint32_t* indices, lv_tube_vertex* vertices, lv_line_point* points) {
    const TubeTriangleRenderData& d = static_cast<FlowHandle*>(hp)->lastTriangleData;
    if (indices) memcpy(indices, d.indexBuffer.data(), d.indexBuffer.size() * 4);
    if (vertices) memcpy(vertices, d.vertexBuffer.data(), d.vertexBuffer.size() * sizeof(lv_tube_vertex));
    if (points) memcpy(points, d.linePointDataBuffer.data(), d.linePointDataBuffer.size() * sizeof(lv_line_point));
}

/// AO prebaker parametrisation of the flow's lines (no GPU involved): counts first (NULL outputs), then the copy.
void lvh_flow_ao_parametrization(void* hp, float expectedParamSegmentLength, float* blendingWeights, float* samplingLocations,
                                 uint64_t* outNumLineVertices, uint64_t* outNumParametrizationVertices) {
    FlowHandle* h = static_cast<FlowHandle*>(hp);
    std::vector<float> bw, sl;
    computeAmbientOcclusionParametrization(h->data->getFilteredLines(nullptr), expectedParamSegmentLength, bw, sl);
    if (blendingWeights) memcpy(blendingWeights, bw.data(), bw.size() * 4);
    if (samplingLocations) memcpy(samplingLocations, sl.data(), sl.size() * 4);
    *outNumLineVertices = bw.size();
    *outNumParametrizationVertices = sl.size();
}

// ---- streamline tracer front end (Flow.hpp)
namespace {
struct GridHandle {
    StreamlineTracingGrid grid;
    Trajectories result;
    std::vector<std::vector<vec3>> ribbons;
    explicit GridHandle(int device) : grid(device) {}
};
} // namespace
void* lvh_grid_create(int deviceOrdinal) {
    GridHandle* h = new GridHandle(deviceOrdinal);
    if (!h->grid.isValid()) { delete h; return nullptr; }
    return h;
}
void lvh_grid_destroy(void* h) { delete static_cast<GridHandle*>(h); }
void lvh_grid_set_extent(void* h, int xs, int ys, int zs, float dx, float dy, float dz) {
    static_cast<GridHandle*>(h)->grid.setGridExtent(xs, ys, zs, dx, dy, dz);
}
void lvh_grid_add_vector_field(void* h, const float* f, const char* name) { static_cast<GridHandle*>(h)->grid.addVectorField(f, name); }
void lvh_grid_add_scalar_field(void* h, const float* f, const char* name) { static_cast<GridHandle*>(h)->grid.addScalarField(f, name); }
void lvh_grid_load_abc_flow(void* h, int xs, int ys, int zs, float resScale) {
    AbcFlowGenerator gen;
    gen.setGridSize(xs, ys, zs);
    gen.setResolutionScale(resScale);
    gen.load(&static_cast<GridHandle*>(h)->grid);
}
void lvh_grid_info(void* hp, int* sizes3, float* spacing3, float* box6) {
    StreamlineTracingGrid& g = static_cast<GridHandle*>(hp)->grid;
    sizes3[0] = g.getGridSizeX(); sizes3[1] = g.getGridSizeY(); sizes3[2] = g.getGridSizeZ();
    spacing3[0] = g.getDx(); spacing3[1] = g.getDy(); spacing3[2] = g.getDz();
    const AABB3& b = g.getBox();
    box6[0] = b.min.x; box6[1] = b.min.y; box6[2] = b.min.z; box6[3] = b.max.x; box6[4] = b.max.y; box6[5] = b.max.z;
}
/// regular volume seeding (StreamlineVolumeSeeder): nx*ny*nz points into out (3 floats each)
void lvh_grid_regular_seeds(void* hp, int nx, int ny, int nz, float* out) {
    StreamlineVolumeSeeder seeder;
    seeder.setRegular(nx, ny, nz);
    seeder.reset(static_cast<GridHandle*>(hp)->grid);
    for (int i = 0; i < nx * ny * nz; i++) {
        vec3 p = seeder.getNextPoint();
        out[3 * i] = p.x; out[3 * i + 1] = p.y; out[3 * i + 2] = p.z;
    }
}
/// plane seeding (StreamlinePlaneSeeder): nx * ny regular points, or n random points (ny = 0, nx = n) into out
void lvh_grid_plane_seeds(void* hp, const float* normal3, float slice, int nx, int ny, int seed, float* out) {
    StreamlinePlaneSeeder seeder;
    seeder.setPlane(vec3(normal3[0], normal3[1], normal3[2]), slice);
    if (ny > 0) seeder.setRegular(nx, ny); else seeder.setRandom(nx, seed);
    seeder.reset(static_cast<GridHandle*>(hp)->grid);
    for (int i = 0; seeder.hasNextPoint(); i++) {
        vec3 p = seeder.getNextPoint();
        out[3 * i] = p.x; out[3 * i + 1] = p.y; out[3 * i + 2] = p.z;
    }
}
int lvh_grid_trace(void* hp, const float* seeds, uint32_t numSeeds, int method, int direction, float timeStepScale,
                   int maxNumIterations, float terminationDistance, float minimumLength, uint64_t* outNumLines,
                   uint64_t* outNumPoints) {
    GridHandle* h = static_cast<GridHandle*>(hp);
    StreamlineTracingSettings s;
    s.integrationMethod = StreamlineIntegrationMethod(method);
    s.integrationDirection = StreamlineIntegrationDirection(direction);
    s.timeStepScale = timeStepScale;
    s.maxNumIterations = maxNumIterations;
    s.terminationDistance = terminationDistance;
    s.minimumLength = minimumLength;
    std::vector<vec3> seedPoints(numSeeds);
    if (numSeeds) memcpy(seedPoints.data(), seeds, size_t(numSeeds) * 12);
    h->result.clear();
    if (!h->grid.traceStreamlines(s, seedPoints, h->result)) return -1;
    uint64_t n = 0;
    for (const Trajectory& t : h->result) n += t.positions.size();
    *outNumLines = h->result.size();
    *outNumPoints = n;
    return 0;
}
/// traceStreamlinesDecreasingHelicity (the max-helicity-first seeder); result fetched like lvh_grid_trace's
int lvh_grid_trace_max_helicity_first(void* hp, int method, int direction, float timeStepScale, int maxNumIterations,
                                      float terminationDistance, float minimumLength, float minimumSeparationDistance,
                                      int terminationCheckType, int loopCheckMode, float terminationDistanceSelf,
                                      int seedingSubsamplingFactor, int ribbons, int useHelicity, float maxHelicityTwist,
                                      const float* initialRibbonDirection, uint64_t* outNumLines, uint64_t* outNumPoints) {
    GridHandle* h = static_cast<GridHandle*>(hp);
    StreamlineTracingSettings s;
    s.integrationMethod = StreamlineIntegrationMethod(method);
    s.integrationDirection = StreamlineIntegrationDirection(direction);
    s.timeStepScale = timeStepScale;
    s.maxNumIterations = maxNumIterations;
    s.terminationDistance = terminationDistance;
    s.minimumLength = minimumLength;
    s.minimumSeparationDistance = minimumSeparationDistance;
    s.terminationCheckType = terminationCheckType;
    s.loopCheckMode = loopCheckMode;
    s.terminationDistanceSelf = terminationDistanceSelf;
    s.seedingSubsamplingFactor = seedingSubsamplingFactor;
    h->result.clear();
    h->ribbons.clear();
    s.useHelicity = useHelicity != 0;
    s.maxHelicityTwist = maxHelicityTwist;
    if (initialRibbonDirection) s.initialRibbonDirection = vec3(initialRibbonDirection[0], initialRibbonDirection[1], initialRibbonDirection[2]);
    if (ribbons ? !h->grid.traceStreamribbonsDecreasingHelicity(s, h->result, h->ribbons)
                : !h->grid.traceStreamlinesDecreasingHelicity(s, h->result)) return -1;
    uint64_t n = 0;
    for (const Trajectory& t : h->result) n += t.positions.size();
    *outNumLines = h->result.size();
    *outNumPoints = n;
    return 0;
}
/// traceStreamribbons: as lvh_grid_trace, plus one ribbon direction per point (fetched with lvh_grid_copy_ribbons)
int lvh_grid_trace_ribbons(void* hp, const float* seeds, uint32_t numSeeds, int method, int direction, float timeStepScale,
                           int maxNumIterations, float terminationDistance, float minimumLength, int useHelicity,
                           float maxHelicityTwist, const float* initialRibbonDirection, uint64_t* outNumLines,
                           uint64_t* outNumPoints) {
    GridHandle* h = static_cast<GridHandle*>(hp);
    StreamlineTracingSettings s;
    s.integrationMethod = StreamlineIntegrationMethod(method);
    s.integrationDirection = StreamlineIntegrationDirection(direction);
    s.timeStepScale = timeStepScale;
    s.maxNumIterations = maxNumIterations;
    s.terminationDistance = terminationDistance;
    s.minimumLength = minimumLength;
    s.useHelicity = useHelicity != 0;
    s.maxHelicityTwist = maxHelicityTwist;
    s.initialRibbonDirection = vec3(initialRibbonDirection[0], initialRibbonDirection[1], initialRibbonDirection[2]);
    std::vector<vec3> seedPoints(numSeeds);
    if (numSeeds) memcpy(seedPoints.data(), seeds, size_t(numSeeds) * 12);
    h->result.clear();
    h->ribbons.clear();
    if (!h->grid.traceStreamribbons(s, seedPoints, h->result, h->ribbons)) return -1;
    uint64_t n = 0;
    for (const Trajectory& t : h->result) n += t.positions.size();
    *outNumLines = h->result.size();
    *outNumPoints = n;
    return 0;
}
void lvh_grid_copy_ribbons(void* hp, float* ribbonDirections) {
    size_t off = 0;
    for (const auto& dirs : static_cast<GridHandle*>(hp)->ribbons) {
        memcpy(ribbonDirections + 3 * off, dirs.data(), dirs.size() * 12);
        off += dirs.size();
    }
}
int lvh_grid_num_scalar_fields(void* hp) { return int(static_cast<GridHandle*>(hp)->grid.getScalarFieldNames().size()); }
/// name of the idx-th scalar field (= attribute idx of the traced lines); copied into out (capacity bytes incl. the terminator)
int lvh_grid_scalar_field_name(void* hp, int idx, char* out, uint32_t capacity) {
    const std::vector<std::string> names = static_cast<GridHandle*>(hp)->grid.getScalarFieldNames();
    if (idx < 0 || size_t(idx) >= names.size() || capacity == 0) return -1;
    snprintf(out, capacity, "%s", names[size_t(idx)].c_str());
    return 0;
}
/// positions n*3, attributes [k][n] (scalar fields in name order), offsets numLines+1
void lvh_grid_copy_result(void* hp, float* positions, float* attributes, uint32_t* offsets) {
    const Trajectories& tr = static_cast<GridHandle*>(hp)->result;
    uint64_t n = 0;
    for (const Trajectory& t : tr) n += t.positions.size();
    uint32_t off = 0;
    for (size_t i = 0; i < tr.size(); i++) {
        offsets[i] = off;
        const uint32_t m = uint32_t(tr[i].positions.size());
        memcpy(positions + 3 * size_t(off), tr[i].positions.data(), size_t(m) * 12);
        for (size_t a = 0; a < tr[i].attributes.size(); a++)
            memcpy(attributes + a * n + off, tr[i].attributes[a].data(), size_t(m) * 4);
        off += m;
    }
    offsets[tr.size()] = off;
}
const char* lvh_grid_last_error(void* h) { return static_cast<GridHandle*>(h)->grid.getLastError().c_str(); }

// ---- headless renderer harness
void* lvh_renderer_create(int renderingMode, int deviceOrdinal) {
    HeadlessLineRenderer* r = new HeadlessLineRenderer(RenderingMode(renderingMode), deviceOrdinal);
    if (!r->isValid()) { delete r; return nullptr; }
    return r;
}
// several GPUs behind the plugin (SceneData::deviceOrdinals -> lv_create_multi); transport "rccl" | "memcpy"
void* lvh_renderer_create_multi(int renderingMode, const int* deviceOrdinals, int numDevices, const char* transport) {
    if (!deviceOrdinals || numDevices <= 0) return nullptr;
    HeadlessLineRenderer* r = new HeadlessLineRenderer(RenderingMode(renderingMode), std::vector<int>(deviceOrdinals, deviceOrdinals + numDevices),
                                                       transport ? transport : "rccl");
    if (!r->isValid()) { delete r; return nullptr; }
    return r;
}
int lvh_renderer_num_devices(void* r) { return static_cast<HeadlessLineRenderer*>(r)->getLineRenderer()->getNumDevices(); }
int lvh_renderer_rebalance(void* r, double baseCostPerTile) {
    return static_cast<HeadlessLineRenderer*>(r)->getLineRenderer()->rebalanceTiles(baseCostPerTile) ? 0 : -1;
}
// MainApp::setNewState for the harness: one InternalState (renderer settings carry the camelCase keys of the canned states)
void lvh_renderer_set_state(void* r, const char* name, int renderingMode, const char* const* rendererKeys, const char* const* rendererValues,
                            uint32_t numRenderer, const char* const* dataSetKeys, const char* const* dataSetValues, uint32_t numDataSet,
                            int tilingWidth, int tilingHeight, int resolutionX, int resolutionY) {
    InternalState st;
    st.name = st.nameRaw = name ? name : "";
    st.renderingMode = renderingMode;
    for (uint32_t i = 0; i < numRenderer; i++) st.rendererSettings.addKeyValue(std::string(rendererKeys[i]), rendererValues[i]);
    for (uint32_t i = 0; i < numDataSet; i++) st.dataSetSettings.addKeyValue(std::string(dataSetKeys[i]), dataSetValues[i]);
    st.tilingWidth = tilingWidth;
    st.tilingHeight = tilingHeight;
    st.windowResolution[0] = resolutionX;
    st.windowResolution[1] = resolutionY;
    static_cast<HeadlessLineRenderer*>(r)->setNewState(st);
}
// the canned benchmark states (getTestModes): count, then per state "name\nmode\nresX\nresY\nkey=value\n..." into buf
uint32_t lvh_test_modes_count(int twice) { return uint32_t(getTestModes(twice != 0).size()); }
uint32_t lvh_test_mode(int twice, uint32_t index, char* buf, uint32_t capacity) {
    const std::vector<InternalState> states = getTestModes(twice != 0);
    if (index >= states.size()) return 0;
    const InternalState& s = states[index];
    std::string out = s.name + "\n" + std::to_string(s.renderingMode) + "\n" + std::to_string(s.windowResolution[0]) + "\n" +
                      std::to_string(s.windowResolution[1]) + "\n";
    for (const auto& kv : s.rendererSettings.getMap()) out += kv.first + "=" + kv.second + "\n";
    if (buf && capacity > out.size()) memcpy(buf, out.c_str(), out.size() + 1);
    return uint32_t(out.size());
}
void lvh_renderer_destroy(void* r) { delete static_cast<HeadlessLineRenderer*>(r); }
void lvh_renderer_set_resolution(void* r, uint32_t w, uint32_t h) { static_cast<HeadlessLineRenderer*>(r)->setRenderingResolution(w, h); }
void lvh_renderer_set_line_data(void* r, void* flow, int isNewData) {
    static_cast<HeadlessLineRenderer*>(r)->setLineData(static_cast<FlowHandle*>(flow)->data, isNewData != 0);
}
void lvh_renderer_set_transfer_function(void* r, const float* rgba, uint32_t n) { static_cast<HeadlessLineRenderer*>(r)->setTransferFunction(rgba, n); }
void lvh_renderer_set_clear_color(void* r, float cr, float cg, float cb, float ca) { static_cast<HeadlessLineRenderer*>(r)->setClearColor(cr, cg, cb, ca); }
void lvh_renderer_set_camera(void* r, const float pos[3], const float lookAt[3]) {
    static_cast<HeadlessLineRenderer*>(r)->setCameraPosition(vec3(pos[0], pos[1], pos[2]), vec3(lookAt[0], lookAt[1], lookAt[2]));
}
void lvh_renderer_set_settings(void* r, const char* const* keys, const char* const* values, uint32_t n) {
    SettingsMap m;
    for (uint32_t i = 0; i < n; i++) m.addKeyValue(std::string(keys[i]), values[i]);
    static_cast<HeadlessLineRenderer*>(r)->setNewSettings(m);
}
int lvh_renderer_render(void* rp, uint8_t* outRGBA8) {
    HeadlessLineRenderer* r = static_cast<HeadlessLineRenderer*>(rp);
    const uint8_t* img = r->renderFrame();
    if (!img) return -1;
    memcpy(outRGBA8, img, size_t(r->getWidth()) * r->getHeight() * 4);
    return 0;
}
/// Camera matrices the harness passes to lv_set_camera for a w x h viewport (column-major) + {fovy, near, far}.
void lvh_renderer_get_camera(void* rp, float* view16, float* proj16, float* fovyNearFar) {
    HeadlessLineRenderer* r = static_cast<HeadlessLineRenderer*>(rp);
    Camera& cam = *r->getLineRenderer()->getSceneData()->camera;
    cam.setAspectRatio(float(r->getWidth()) / float(r->getHeight()));
    mat4 v = cam.getViewMatrix(), p = cam.getProjectionMatrix();
    memcpy(view16, v.m, 64);
    memcpy(proj16, p.m, 64);
    fovyNearFar[0] = cam.getFOVy();
    fovyNearFar[1] = cam.getNearClipDistance();
    fovyNearFar[2] = cam.getFarClipDistance();
}
const char* lvh_renderer_last_error(void* r) { return static_cast<HeadlessLineRenderer*>(r)->getLastError().c_str(); }
int lvh_renderer_rendering_mode(void* r) { return int(static_cast<HeadlessLineRenderer*>(r)->getLineRenderer()->getRenderingMode()); }
int lvh_renderer_needs_re_render(void* r) { return static_cast<HeadlessLineRenderer*>(r)->getLineRenderer()->needsReRender() ? 1 : 0; }
/// state of the static AO baker of the renderer: bit 0 = data ready, bit 1 = computation running; -1 = no static prebaker
int lvh_renderer_ao_baker_state(void* r) {
    LineRenderer* lr = static_cast<HeadlessLineRenderer*>(r)->getLineRenderer();
    AmbientOcclusionBaker* b = lr->getAmbientOcclusionBaker();
    if (!b || !b->getIsStaticPrebaker()) return -1;
    HipAmbientOcclusionBaker* hb = static_cast<HipAmbientOcclusionBaker*>(b);
    return (hb->getIsDataReady() ? 1 : 0) | (hb->getIsComputationRunning() ? 2 : 0);
}
void* lvh_renderer_context(void* r) { return static_cast<HeadlessLineRenderer*>(r)->getLineRenderer()->getContext(); }

} // extern "C"
