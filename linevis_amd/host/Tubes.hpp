// Tubes.hpp -- triangle tessellation of line sets (see Tubes.cpp for the reference lines it follows).
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/linevis_hip.h"
#include "LvMath.hpp"

namespace lv {

typedef lv_tube_vertex TubeTriangleVertexData; // 32 bytes, src/LineData/LineRenderData.hpp:171-176
static_assert(sizeof(TubeTriangleVertexData) == 32, "TubeTriangleVertexData must stay byte-identical to the reference");

/// src/LineData/LineRenderData.hpp:178-185
struct LinePointReference {
    LinePointReference() = default;
    LinePointReference(uint32_t trajectoryIndex, uint32_t linePointIndex)
            : trajectoryIndex(trajectoryIndex), linePointIndex(linePointIndex) {}
    uint32_t trajectoryIndex = 0; ///< Index of the trajectory.
    uint32_t linePointIndex = 0;  ///< Index of the line point within the trajectory.
};

/// Open tubes with hemisphere caps (tubeClosed = false of createCappedTriangleTubesRenderDataCPU).
void createCappedTriangleTubesRenderData(
        const std::vector<std::vector<vec3>>& lineCentersList, float tubeRadius, int numCircleSubdivisions,
        std::vector<uint32_t>& triangleIndices, std::vector<TubeTriangleVertexData>& vertexDataList,
        std::vector<LinePointReference>& linePointReferenceList, std::vector<vec3>& lineTangents,
        std::vector<vec3>& lineNormals);

/// The elliptic tubes of a band data set (createCappedTriangleEllipticTubesRenderDataCPU, tubeClosed = false): line normal =
/// cross(right vector, tangent), semi-axes tubeNormalRadius / tubeBinormalRadius.
void createCappedTriangleEllipticTubesRenderData(
        const std::vector<std::vector<vec3>>& lineCentersList, const std::vector<std::vector<vec3>>& lineRightVectorsList,
        float tubeNormalRadius, float tubeBinormalRadius, int numEllipseSubdivisions,
        std::vector<uint32_t>& triangleIndices, std::vector<TubeTriangleVertexData>& vertexDataList,
        std::vector<LinePointReference>& linePointReferenceList, std::vector<vec3>& lineTangents,
        std::vector<vec3>& lineNormals);

} // namespace lv
