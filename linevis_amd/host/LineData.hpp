// LineData.hpp -- headless line-data model feeding the HIP renderers.
//
// Mirrors the part of the reference's data model the hot path consumes:
//   struct Trajectory / Trajectories                       src/Loaders/TrajectoryFile.hpp:38-46
//   normalizeTrajectoriesVertexPositions                   src/Loaders/TrajectoryFile.cpp:106-125
//   loadTrajectoriesFromBinLines (v1 / v2 header)          src/Loaders/BinLinesLoader.cpp:41-63,127-150
//   loadTrajectoriesFromObj, loadFlowTrajectoriesFromFile  src/Loaders/ObjLoader.cpp:36-186, TrajectoryFile.cpp:634-655
//   struct LinePointDataUnified (48 B), TubeAabbRenderData src/LineData/LineRenderData.hpp:99-106,203-210
//   class LineData accessors                               src/LineData/LineData.hpp:149-262
//   LineDataFlow::setTrajectoryData                        src/LineData/LineDataFlow.cpp:468-578
//   LineDataFlow::getLinePassTubeAabbRenderData            src/LineData/LineDataFlow.cpp:2112-2277
//   LineDataFlow::getLinePassTubeTriangleMeshRenderDataPayload  src/LineData/LineDataFlow.cpp:1912-2110 (+ Tubes.cpp)
// Accessors return host-side POD arrays (byte-identical record layouts) instead of sgl::vk::BufferPtr.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "../../include/linevis_hip.h"
#include "LvMath.hpp"
#include "SettingsMap.hpp"
#include "Tubes.hpp"

namespace lv {

struct Trajectory {
    std::vector<vec3> positions;
    std::vector<std::vector<float>> attributes;
};
typedef std::vector<Trajectory> Trajectories;

AABB3 computeTrajectoriesAABB3(const Trajectories& trajectories);
void normalizeTrajectoriesVertexPositions(Trajectories& trajectories, const AABB3& aabb);
/// BinLinesData (src/Loaders/TrajectoryFile.hpp:60-75): trajectories + what version 2 of the format adds; the outline mesh of
/// the simulation grid is skipped.
struct BinLinesData {
    Trajectories trajectories;
    std::vector<std::string> attributeNames;
    std::vector<std::vector<vec3>> ribbonsDirections; // per trajectory, one direction per point; empty = no band data
    bool verticesNormalized = false;
};
/// .binlines reader, versions 1 and 2 (BinLinesLoader.cpp:41-180)
bool loadTrajectoriesFromBinLines(const std::string& filename, BinLinesData& binLinesData);
bool loadTrajectoriesFromBinLines(const std::string& filename, Trajectories& trajectories);
/// writer (BinLinesLoader.cpp:182-247): version 1 without, version 2 with ribbon directions
bool saveTrajectoriesAsBinLines(const std::string& filename, const Trajectories& trajectories,
                                const std::vector<std::vector<vec3>>& ribbonsDirections = {}, bool verticesNormalized = false);
/// Wavefront-OBJ polylines as LineVis reads them (src/Loaders/ObjLoader.cpp:36-186): "v x y z" positions, "vt a0 a1 .."
/// per-vertex attributes (same count on every line), "l i j k .." one trajectory per line statement (1-based indices),
/// "a name0 name1 .." attribute names; positions with a component above 1e10 are dropped.
bool loadTrajectoriesFromObj(const std::string& filename, Trajectories& trajectories, std::vector<std::string>& attributeNames);
/// loadFlowTrajectoriesFromFile (src/Loaders/TrajectoryFile.cpp:634-655): by extension, .obj or .binlines
bool loadFlowTrajectoriesFromFile(const std::string& filename, Trajectories& trajectories, std::vector<std::string>& attributeNames);

typedef lv_line_point LinePointDataUnified; // 48 bytes, src/LineData/LineRenderData.hpp:99-106
static_assert(sizeof(LinePointDataUnified) == 48, "LinePointDataUnified must stay byte-identical to the reference");

struct TubeAabbRenderData {
    std::vector<uint32_t> indexBuffer;                     // two point indices per AABB / segment
    std::vector<AABB3> aabbBuffer;                         // VkAabbPositionsKHR layout (6 floats)
    std::vector<LinePointDataUnified> linePointDataBuffer;
};

/// src/LineData/LineRenderData.hpp:187-201, host vectors instead of sgl::vk::BufferPtr
struct TubeTriangleRenderData {
    std::vector<uint32_t> indexBuffer;                      // 3 vertex indices per triangle
    std::vector<TubeTriangleVertexData> vertexBuffer;
    std::vector<LinePointDataUnified> linePointDataBuffer;  // referenced by vertexLinePointIndex
};

enum DataSetType { DATA_SET_TYPE_NONE = 0, DATA_SET_TYPE_FLOW_LINES = 1 };

class LineRenderer;

class LineData {
public:
    explicit LineData(DataSetType type) : dataSetType(type) {}
    virtual ~LineData() = default;
    DataSetType getType() const { return dataSetType; }
    const AABB3& getModelBoundingBox() const { return modelBoundingBox; }
    const std::vector<std::string>& getAttributeNames() const { return attributeNames; }
    int getSelectedAttributeIndex() const { return selectedAttributeIndex; }
    void setSelectedAttributeIndex(int idx);
    /// (min, max) of the selected attribute: the transfer-function range (LineDataFlow.cpp:511-530).
    void getMinMaxAttributeValues(float& minAttr, float& maxAttr) const;
    bool getUseCappedTubes() const { return useCappedTubes; }
    /// LinePrimitiveMode (LineData.hpp:264-282, names LineData.cpp:56-74; settings keys line_primitive_mode / _index :108-140).  The
    /// rasterised geometries themselves are not built (the PPLL plugin gathers ray-entry hits of tubes); the mode decides what the
    /// reference derives from it: USE_CAPPED_TUBES of a rasteriser's shaders and the band-rendering switch.
    enum LinePrimitiveMode {
        LINE_PRIMITIVES_QUADS_PROGRAMMABLE_PULL, LINE_PRIMITIVES_QUADS_GEOMETRY_SHADER, LINE_PRIMITIVES_TUBE_PROGRAMMABLE_PULL,
        LINE_PRIMITIVES_TUBE_GEOMETRY_SHADER, LINE_PRIMITIVES_TUBE_TRIANGLE_MESH, LINE_PRIMITIVES_TUBE_MESH_SHADER,
        LINE_PRIMITIVES_TUBE_MESH_SHADER_NV, LINE_PRIMITIVES_RIBBON_QUADS_GEOMETRY_SHADER,
        LINE_PRIMITIVES_TUBE_RIBBONS_PROGRAMMABLE_PULL, LINE_PRIMITIVES_TUBE_RIBBONS_GEOMETRY_SHADER,
        LINE_PRIMITIVES_TUBE_RIBBONS_TRIANGLE_MESH, LINE_PRIMITIVES_TUBE_RIBBONS_MESH_SHADER,
        LINE_PRIMITIVES_TUBE_RIBBONS_MESH_SHADER_NV, LINE_PRIMITIVES_COUNT
    };
    static LinePrimitiveMode getLinePrimitiveMode() { return linePrimitiveMode; }
    static void setLinePrimitiveMode(LinePrimitiveMode mode) { linePrimitiveMode = mode; }
    static bool getUseBandRendering() { // LineData.hpp:323-334
        return linePrimitiveMode == LINE_PRIMITIVES_RIBBON_QUADS_GEOMETRY_SHADER || linePrimitiveMode == LINE_PRIMITIVES_TUBE_RIBBONS_PROGRAMMABLE_PULL ||
               linePrimitiveMode == LINE_PRIMITIVES_TUBE_RIBBONS_GEOMETRY_SHADER || linePrimitiveMode == LINE_PRIMITIVES_TUBE_RIBBONS_TRIANGLE_MESH ||
               linePrimitiveMode == LINE_PRIMITIVES_TUBE_RIBBONS_MESH_SHADER || linePrimitiveMode == LINE_PRIMITIVES_TUBE_RIBBONS_MESH_SHADER_NV;
    }
    /// USE_CAPPED_TUBES of getVulkanShaderPreprocessorDefines (LineData.cpp:1240-1244; no deferred renderer here): ray tracers always
    /// follow use_capped_tubes, rasterisers only when the primitive mode is one of the triangle-mesh modes (the other tube
    /// geometries have no caps)
    bool getUseCappedTubesDefine(bool isRasterizer) const {
        return useCappedTubes && (!isRasterizer || linePrimitiveMode == LINE_PRIMITIVES_TUBE_TRIANGLE_MESH ||
                                  linePrimitiveMode == LINE_PRIMITIVES_TUBE_RIBBONS_TRIANGLE_MESH);
    }
    bool getUseHalos() const { return useHalos; }
    /// USE_BANDS of the ray tracer's shaders (getVulkanShaderPreprocessorDefines with isRasterizer = false)
    virtual bool getUseBands() const { return false; }
    /// USE_ROTATING_HELICITY_BANDS + the LineUniformData members it reads (LineDataFlow.cpp:979-984,2432-2440)
    virtual bool getUseRotatingHelicityBands() const { return false; }
    virtual float getSeparatorWidth() const { return 0.2f; }
    virtual uint32_t getNumSubdivisionsBands() const { return 6u; }
    virtual float getHelicityRotationFactor() const { return 1.0f; }
    virtual bool getUseUniformTwistLineWidth() const { return true; }
    /// USE_HELICITY_BANDS_TEXTURE: use_twist_line_texture && a loaded texture (LineDataFlow.cpp:2437); RGBA8 pixels, filtering mode index
    virtual bool getUseTwistLineTexture() const { return false; }
    virtual const std::vector<uint8_t>* getTwistLineTexture(uint32_t& width, uint32_t& height) const { width = height = 0; return nullptr; }
    virtual int getTwistLineTextureFilteringModeIndex() const { return 5; }
    static bool getRenderThickBands() { return renderThickBands; }   // LineData.cpp:53
    static float getMinBandThickness() { return minBandThickness; }  // LineData.cpp:54
    int getTubeNumSubdivisions() const { return tubeNumSubdivisions; }
    bool isDirty() const { return dirty; }
    void setDirty(bool d) { dirty = d; }

    virtual size_t getNumLines() = 0;
    virtual size_t getNumLinePoints() = 0;
    virtual size_t getNumLineSegments() = 0;
    virtual TubeAabbRenderData getLinePassTubeAabbRenderData(bool isRasterizer, bool ellipticTubes) = 0;
    /// LineData.hpp:182; consumed by the RTAO pass (VulkanRayTracedAmbientOcclusion.cpp:444-445)
    virtual TubeTriangleRenderData getLinePassTubeTriangleMeshRenderData(bool isRasterizer, bool vulkanRayTracing) = 0;
    /// Points of all (unfiltered) lines, used for the depth-cue range (LineRenderer.cpp:365-408).
    virtual std::vector<std::vector<vec3>> getFilteredLines(LineRenderer* lineRenderer) = 0;

    /// Device geometry (round 6): plain flow lines (no band data, no rotating helicity bands) can hand their trajectories -- positions,
    /// the selected attribute, one offset per line: LineDataFlow::setTrajectoryData's arrays, LineDataFlow.cpp:468-578 -- to
    /// lv_set_trajectories and let kernels write what getLinePassTubeAabbRenderData / getLinePassTubeTriangleMeshRenderData produce.
    /// false: this data needs the host-built render data.  getDataGeneration() changes whenever the arrays would.
    virtual bool getTrajectoryArrays(std::vector<float>& positions, std::vector<float>& attribute, std::vector<uint32_t>& lineOffsets) {
        (void)positions; (void)attribute; (void)lineOffsets;
        return false;
    }
    uint64_t getDataGeneration() const { return dataGeneration; }

    /// dataset-side settings keys: attribute, tube_num_subdivisions, use_capped_tubes, use_halos
    /// (src/LineData/LineData.cpp:87-181).  Returns true when renderers must re-fetch geometry/defines.
    virtual bool setNewSettings(const SettingsMap& settings);
    void setTriangleRepresentationDirty() { cachedAabbDataValid = false; cachedTriangleDataValid = false; dirty = true; }

protected:
    static LinePrimitiveMode linePrimitiveMode; // LineData.cpp:51
    static bool renderThickBands;
    static float minBandThickness;
    DataSetType dataSetType;
    AABB3 modelBoundingBox;
    std::vector<std::string> attributeNames;
    std::vector<std::pair<float, float>> minMaxAttributeValues;
    int selectedAttributeIndex = 0;
    bool useCappedTubes = true;      // LineData.hpp:377-379
    bool useHalos = true;
    int tubeNumSubdivisions = 6;     // LineData.cpp:52
    bool dirty = false;
    uint64_t dataGeneration = 1;     // bumped by new trajectories / another selected attribute
    bool cachedAabbDataValid = false;
    bool cachedTriangleDataValid = false;
};
typedef std::shared_ptr<LineData> LineDataPtr;

class LineDataFlow : public LineData {
public:
    LineDataFlow() : LineData(DATA_SET_TYPE_FLOW_LINES) {}
    /// loadFromFile (LineDataFlow.cpp:431-454) for .binlines; normalises positions like the reference loader.
    bool loadFromFile(const std::string& filename);
    void setTrajectoryData(const Trajectories& trajectories, const std::vector<std::string>& names = {},
                           const std::vector<std::vector<vec3>>& ribbonsDirections = {});
    const Trajectories& getTrajectories() const { return trajectories; }
    const std::vector<std::vector<vec3>>& getRibbonsDirections() const { return ribbonsDirections; }
    /// the positions this object holds went through normalizeTrajectoriesVertexPositions (loadFromFile) or the caller said so:
    /// what the version-2 .binlines writer records as verticesNormalized (BinLinesLoader.cpp:196-247, TrajectoryFile.cpp:656)
    bool getVerticesNormalized() const { return verticesNormalized; }
    void setVerticesNormalized(bool v) { verticesNormalized = v; }
    bool getHasBandsData() const { return hasBandsData; }
    /// USE_BANDS: useRibbons && hasBandsData (LineDataFlow.cpp:2423); use_ribbons / thick_bands / min_band_thickness keys :587-606
    /// (USE_BANDS also needs !useRotatingHelicityBands there; rotating_helicity_bands = true switches useRibbons off, :601-604)
    bool getUseBands() const override { return useRibbons && hasBandsData && !getUseRotatingHelicityBands(); }
    /// "Show Helicity Bands": rotating_helicity_bands, only for data with an attribute whose name contains "helicity" (:535-550)
    bool getHasHelicity() const { return hasHelicity; }
    float getMaxHelicity() const { return maxHelicity; }
    bool getUseRotatingHelicityBands() const override { return useRotatingHelicityBands && hasHelicity; }
    float getSeparatorWidth() const override { return separatorWidth; }
    uint32_t getNumSubdivisionsBands() const override { return numSubdivisionsBands; }
    float getHelicityRotationFactor() const override { return helicityRotationFactor; }
    bool getUseUniformTwistLineWidth() const override { return useUniformTwistLineWidth; }
    /// LineDataFlow::loadTwistLineTexture (LineDataFlow.cpp:93-171) without the PNG decoder: the embedder hands over the RGBA8 pixels
    /// (the "twist_line_texture" key of the reference names a file); empty = unloaded
    void setTwistLineTexture(const uint8_t* rgba8, uint32_t width, uint32_t height);
    bool getUseTwistLineTexture() const override { return useTwistLineTexture && !twistLineTexture.empty(); }
    const std::vector<uint8_t>* getTwistLineTexture(uint32_t& width, uint32_t& height) const override {
        width = twistW; height = twistH; return twistLineTexture.empty() ? nullptr : &twistLineTexture;
    }
    int getTwistLineTextureFilteringModeIndex() const override { return textureFilteringModeIndex; }
    bool setNewSettings(const SettingsMap& settings) override;

    size_t getNumLines() override { return numTotalTrajectories; }
    size_t getNumLinePoints() override { return numTotalTrajectoryPoints; }
    size_t getNumLineSegments() override;
    TubeAabbRenderData getLinePassTubeAabbRenderData(bool isRasterizer, bool ellipticTubes) override;
    TubeTriangleRenderData getLinePassTubeTriangleMeshRenderData(bool isRasterizer, bool vulkanRayTracing) override;
    std::vector<std::vector<vec3>> getFilteredLines(LineRenderer* lineRenderer) override;
    bool getTrajectoryArrays(std::vector<float>& positions, std::vector<float>& attribute, std::vector<uint32_t>& lineOffsets) override;

private:
    Trajectories trajectories;
    std::vector<std::vector<vec3>> ribbonsDirections; // LineDataFlow.hpp:160
    bool verticesNormalized = false;
    bool hasBandsData = false;
    static bool useRibbons;                            // LineDataFlow.cpp:51
    static bool useRotatingHelicityBands;              // LineDataFlow.cpp:52
    bool useTwistLineTexture = false;                  // LineDataFlow.hpp: use_twist_line_texture
    int textureFilteringModeIndex = 5;                 // "Linear Mipmap Linear"
    std::vector<uint8_t> twistLineTexture;
    uint32_t twistW = 0, twistH = 0;
    static float separatorWidth;                       // LineDataFlow.cpp:54 (0.2)
    static bool useUniformTwistLineWidth;              // LineDataFlow.cpp:53 (true)
    int helicityAttributeIndex = -1;
    bool hasHelicity = false;
    float maxHelicity = 0.0f;
    float helicityRotationFactor = 1.0f;               // LineDataFlow.hpp:171
    uint32_t numSubdivisionsBands = 6;                 // LineDataFlow.hpp:188
    bool cachedHelicityBands = false, cachedTriangleHelicityBands = false;
    size_t numTotalTrajectories = 0, numTotalTrajectoryPoints = 0;
    TubeAabbRenderData cachedTubeAabbRenderData;
    float cachedLineWidth = -1.0f;
    bool cachedEllipticTubes = false;
    TubeTriangleRenderData cachedTubeTriangleRenderData;
    float cachedTriangleLineWidth = -1.0f;
    int cachedTriangleSubdivisions = -1;
    bool cachedTriangleBands = false;
    float cachedTriangleBandWidth = -1.0f;
};

} // namespace lv
