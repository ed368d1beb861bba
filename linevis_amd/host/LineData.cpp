// LineData.cpp -- see LineData.hpp for the reference lines each function follows.
#include "LineData.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "LineRenderer.hpp"

namespace lv {

AABB3 computeTrajectoriesAABB3(const Trajectories& trajectories) {
    AABB3 aabb;
    for (const Trajectory& t : trajectories)
        for (const vec3& p : t.positions) aabb.combine(p);
    return aabb;
}

// TrajectoryFile.cpp:106-125: centre at the origin, uniform scale so that the largest extent becomes 0.5.
void normalizeTrajectoriesVertexPositions(Trajectories& trajectories, const AABB3& aabb) {
    vec3 translation = vec3(-((aabb.min.x + aabb.max.x) / 2.0f), -((aabb.min.y + aabb.max.y) / 2.0f),
                            -((aabb.min.z + aabb.max.z) / 2.0f));
    vec3 scale3D = 0.5f / aabb.getDimensions();
    float scale = std::min(scale3D.x, std::min(scale3D.y, scale3D.z));
#pragma omp parallel for
    for (long i = 0; i < long(trajectories.size()); i++)
        for (vec3& v : trajectories[size_t(i)].positions) v = (v + translation) * scale;
}

namespace {
struct Reader {
    const unsigned char* p;
    size_t n, off = 0;
    bool ok = true;
    template <typename T>
    T get() {
        T v{};
        if (off + sizeof(T) > n) { ok = false; return v; }
        memcpy(&v, p + off, sizeof(T));
        off += sizeof(T);
        return v;
    }
    bool getArray(void* dst, size_t bytes) {
        if (off + bytes > n) { ok = false; return false; }
        memcpy(dst, p + off, bytes);
        off += bytes;
        return true;
    }
};
} // namespace

// BinLinesLoader.cpp:127-180 (version word), :41-63 (v1 payload), :68-125 (v2 trailer)
bool loadTrajectoriesFromBinLines(const std::string& filename, BinLinesData& binLinesData) {
    FILE* f = fopen(filename.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> buf(len > 0 ? size_t(len) : 0);
    size_t rd = buf.empty() ? 0 : fread(buf.data(), 1, buf.size(), f);
    fclose(f);
    if (rd != buf.size()) return false;
    Reader r{buf.data(), buf.size()};
    uint32_t versionNumber = r.get<uint32_t>();
    if (!r.ok || (versionNumber != 1u && versionNumber != 2u)) return false;
    uint32_t numTrajectories = r.get<uint32_t>();
    uint32_t numAttributes = r.get<uint32_t>();
    if (!r.ok) return false;
    Trajectories& trajectories = binLinesData.trajectories;
    trajectories.clear();
    trajectories.resize(numTrajectories);
    for (uint32_t i = 0; i < numTrajectories; i++) {
        Trajectory& t = trajectories[i];
        uint32_t n = r.get<uint32_t>();
        if (!r.ok) return false;
        t.positions.resize(n);
        if (!r.getArray(t.positions.data(), sizeof(vec3) * n)) return false;
        t.attributes.resize(numAttributes);
        for (uint32_t a = 0; a < numAttributes; a++) {
            t.attributes[a].resize(n);
            if (!r.getArray(t.attributes[a].data(), sizeof(float) * n)) return false;
        }
    }
    binLinesData.attributeNames.clear();
    binLinesData.ribbonsDirections.clear();
    binLinesData.verticesNormalized = false;
    if (versionNumber == 2u) {
        binLinesData.verticesNormalized = r.get<uint32_t>() != 0u;
        const uint32_t hasAttributeNames = r.get<uint32_t>();
        if (!r.ok) return false;
        if (hasAttributeNames != 0u) {
            const uint32_t n = trajectories.empty() ? 0u : numAttributes;
            for (uint32_t a = 0; a < n; a++) { // sgl::BinaryReadStream::read(std::string&): u32 length, then the characters
                const uint32_t strLen = r.get<uint32_t>();
                if (!r.ok || size_t(strLen) > r.n - r.off) return false; // truncated / corrupt: no allocation from a file-controlled length
                std::string name(strLen, '\0');
                if (strLen != 0u && !r.getArray(&name[0], strLen)) return false;
                binLinesData.attributeNames.push_back(name);
            }
        }
        const uint32_t hasRibbonData = r.get<uint32_t>();
        if (!r.ok) return false;
        if (hasRibbonData != 0u) {
            binLinesData.ribbonsDirections.resize(numTrajectories);
            for (uint32_t i = 0; i < numTrajectories; i++) {
                std::vector<vec3>& dirs = binLinesData.ribbonsDirections[i];
                dirs.resize(trajectories[i].positions.size());
                if (!r.getArray(dirs.data(), sizeof(vec3) * dirs.size())) return false;
            }
        }
        // numMeshOutlineTriangleIndices / Vertices / Normals and their arrays: not used by the line renderers
    }
    return true;
}
bool loadTrajectoriesFromBinLines(const std::string& filename, Trajectories& trajectories) {
    BinLinesData d;
    if (!loadTrajectoriesFromBinLines(filename, d)) return false;
    trajectories.swap(d.trajectories);
    return true;
}

bool saveTrajectoriesAsBinLines(const std::string& filename, const Trajectories& trajectories,
                                const std::vector<std::vector<vec3>>& ribbonsDirections, bool verticesNormalized) {
    FILE* f = fopen(filename.c_str(), "wb");
    if (!f) return false;
    const bool v2 = !ribbonsDirections.empty();
    uint32_t hdr[3] = {v2 ? 2u : 1u, uint32_t(trajectories.size()),
                       trajectories.empty() ? 0u : uint32_t(trajectories[0].attributes.size())};
    fwrite(hdr, 4, 3, f);
    for (const Trajectory& t : trajectories) {
        uint32_t n = uint32_t(t.positions.size());
        fwrite(&n, 4, 1, f);
        fwrite(t.positions.data(), sizeof(vec3), n, f);
        for (uint32_t a = 0; a < hdr[2]; a++) fwrite(t.attributes[a].data(), sizeof(float), n, f);
    }
    if (v2) { // BinLinesLoader.cpp:196-247: verticesNormalized, hasAttributeNames = 0, hasRibbonData = 1, no outline mesh
        const uint32_t trailer[3] = {verticesNormalized ? 1u : 0u, 0u, 1u};
        fwrite(trailer, 4, 3, f);
        for (size_t i = 0; i < trajectories.size(); i++)
            fwrite(ribbonsDirections[i].data(), sizeof(vec3), trajectories[i].positions.size(), f);
        const uint32_t mesh[3] = {0u, 0u, 0u};
        fwrite(mesh, 4, 3, f);
    }
    return fclose(f) == 0;
}

// ObjLoader.cpp:36-186.  One pass over the text; a statement is identified by its first two characters.
bool loadTrajectoriesFromObj(const std::string& filename, Trajectories& trajectories, std::vector<std::string>& attributeNames) {
    FILE* f = fopen(filename.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::string text(len > 0 ? size_t(len) : 0, '\0');
    size_t rd = text.empty() ? 0 : fread(&text[0], 1, text.size(), f);
    fclose(f);
    if (rd != text.size()) return false;

    std::vector<vec3> vertices;
    std::vector<float> vertexAttributes;
    size_t numAttributes = 0;
    auto tokens = [](const std::string& line, size_t from, std::vector<std::string>& out) {
        out.clear();
        size_t i = from;
        while (i < line.size()) {
            while (i < line.size() && (line[i] == ' ' || line[i] == '\t')) i++;
            size_t b = i;
            while (i < line.size() && line[i] != ' ' && line[i] != '\t') i++;
            if (i > b) out.push_back(line.substr(b, i - b));
        }
    };
    std::vector<std::string> tok;
    trajectories.clear();
    size_t pos = 0;
    while (pos < text.size()) {
        size_t e = text.find_first_of("\r\n", pos);
        if (e == std::string::npos) e = text.size();
        const std::string line = text.substr(pos, e - pos);
        pos = e + 1;
        if (line.empty()) continue;
        const char c0 = line[0], c1 = line.size() > 1 ? line[1] : ' ';
        if (c0 == 'v' && c1 == 't') {
            tokens(line, 2, tok);
            for (const std::string& t : tok) vertexAttributes.push_back(float(atof(t.c_str())));
            numAttributes = tok.size();
        } else if (c0 == 'v' && c1 == 'n') {
            // normals are not used
        } else if (c0 == 'v') {
            vec3 p(0.0f, 0.0f, 0.0f);
            sscanf(line.c_str() + 2, "%f %f %f", &p.x, &p.y, &p.z);
            vertices.push_back(p);
        } else if (c0 == 'l') {
            tokens(line, 2, tok);
            Trajectory t;
            t.attributes.resize(numAttributes);
            for (const std::string& s : tok) {
                const long idx = atol(s.c_str()) - 1;
                if (idx < 0 || size_t(idx) >= vertices.size()) return false;
                const vec3 p = vertices[size_t(idx)];
                const float MAX_VAL = 1e10f; // markers of invalid points in scientific data sets
                if (std::fabs(p.x) > MAX_VAL || std::fabs(p.y) > MAX_VAL || std::fabs(p.z) > MAX_VAL) continue;
                t.positions.push_back(p);
                for (size_t j = 0; j < numAttributes; j++) {
                    const size_t a = size_t(idx) * numAttributes + j;
                    t.attributes[j].push_back(a < vertexAttributes.size() ? vertexAttributes[a] : 0.0f);
                }
            }
            trajectories.push_back(std::move(t));
        } else if (c0 == 'a') {
            if (attributeNames.empty()) {
                tokens(line, 2, tok);
                attributeNames = tok;
            }
        }
        // 'g' (new path), '#' (comment) and unknown statements are skipped
    }
    return true;
}

bool loadFlowTrajectoriesFromFile(const std::string& filename, Trajectories& trajectories, std::vector<std::string>& attributeNames) {
    std::string lower = filename;
    for (char& c : lower) c = char(tolower(c));
    auto endsWith = [&](const char* ext) {
        const size_t n = strlen(ext);
        return lower.size() >= n && lower.compare(lower.size() - n, n, ext) == 0;
    };
    if (endsWith(".obj")) return loadTrajectoriesFromObj(filename, trajectories, attributeNames);
    if (endsWith(".binlines")) return loadTrajectoriesFromBinLines(filename, trajectories);
    return false; // .nc needs NetCDF (out of scope), anything else is unknown
}

// ---------------------------------------------------------------- LineData
void LineData::setSelectedAttributeIndex(int idx) {
    if (idx != selectedAttributeIndex) {
        selectedAttributeIndex = idx;
        dataGeneration++;
        setTriangleRepresentationDirty();
    }
}

void LineData::getMinMaxAttributeValues(float& minAttr, float& maxAttr) const {
    if (selectedAttributeIndex >= 0 && size_t(selectedAttributeIndex) < minMaxAttributeValues.size()) {
        minAttr = minMaxAttributeValues[size_t(selectedAttributeIndex)].first;
        maxAttr = minMaxAttributeValues[size_t(selectedAttributeIndex)].second;
    } else {
        minAttr = 0.0f;
        maxAttr = 1.0f;
    }
}

LineData::LinePrimitiveMode LineData::linePrimitiveMode = LineData::LINE_PRIMITIVES_TUBE_PROGRAMMABLE_PULL;
static const char* const LINE_PRIMITIVE_MODE_DISPLAYNAMES[] = { // LineData.cpp:56-74
        "Quads (Programmable Pull)", "Quads (Geometry Shader)", "Tube (Programmable Pull)", "Tube (Geometry Shader)",
        "Tube (Triangle Mesh)", "Tube (Mesh Shader)", "Tube (Mesh Shader NV)", "Ribbon Quads (Geometry Shader)",
        "Tube Ribbons (Programmable Pull)", "Tube Ribbons (Geometry Shader)", "Tube Ribbons (Triangle Mesh)",
        "Tube Ribbons (Mesh Shader)", "Tube Ribbons (Mesh Shader NV)"};

bool LineData::setNewSettings(const SettingsMap& settings) {
    bool shallReloadGatherShader = false;
    std::string linePrimitiveModeName; // :107-140
    if (settings.getValueOpt("line_primitive_mode", linePrimitiveModeName)) {
        for (int i = 0; i < int(LINE_PRIMITIVES_COUNT); i++)
            if (linePrimitiveModeName == LINE_PRIMITIVE_MODE_DISPLAYNAMES[i]) {
                if (linePrimitiveMode != LinePrimitiveMode(i)) { dirty = true; shallReloadGatherShader = true; }
                linePrimitiveMode = LinePrimitiveMode(i);
                break;
            }
    }
    int linePrimitiveModeIndex = 0;
    if (settings.getValueOpt("line_primitive_mode_index", linePrimitiveModeIndex) && linePrimitiveModeIndex >= 0 &&
        linePrimitiveModeIndex < int(LINE_PRIMITIVES_COUNT)) {
        if (linePrimitiveMode != LinePrimitiveMode(linePrimitiveModeIndex)) { dirty = true; shallReloadGatherShader = true; }
        linePrimitiveMode = LinePrimitiveMode(linePrimitiveModeIndex);
    }
    std::string attributeName;
    if (settings.getValueOpt("attribute", attributeName)) {
        for (size_t i = 0; i < attributeNames.size(); i++) {
            if (attributeNames[i] == attributeName) { setSelectedAttributeIndex(int(i)); break; }
        }
    }
    int n = tubeNumSubdivisions;
    if (settings.getValueOpt("tube_num_subdivisions", n) && n != tubeNumSubdivisions) {
        tubeNumSubdivisions = n;
        cachedTriangleDataValid = false;
        dirty = true; // renderers re-fetch the triangle representation (and the subdivision-dependent AO offset)
        shallReloadGatherShader = true;
    }
    bool b = useCappedTubes;
    if (settings.getValueOpt("use_capped_tubes", b) && b != useCappedTubes) {
        useCappedTubes = b;
        shallReloadGatherShader = true;
    }
    b = useHalos;
    if (settings.getValueOpt("use_halos", b) && b != useHalos) {
        useHalos = b;
        shallReloadGatherShader = true;
    }
    return shallReloadGatherShader;
}

// ---------------------------------------------------------------- LineDataFlow
bool LineDataFlow::loadFromFile(const std::string& filename) {
    Trajectories loaded;
    std::vector<std::string> names;
    std::vector<std::vector<vec3>> ribbons;
    bool alreadyNormalized = false;
    const size_t n = filename.size();
    if (n >= 9 && filename.compare(n - 9, 9, ".binlines") == 0) { // LineDataFlow.cpp:436-446: band data comes with the file
        BinLinesData d;
        if (!loadTrajectoriesFromBinLines(filename, d)) return false;
        loaded.swap(d.trajectories);
        names = d.attributeNames;
        ribbons.swap(d.ribbonsDirections);
        alreadyNormalized = d.verticesNormalized; // version 2: StreamlineTracingRequester exports grid-normalised lines with the flag set
    } else if (!loadFlowTrajectoriesFromFile(filename, loaded, names)) {
        return false;
    }
    // loadFlowTrajectoriesFromFile, TrajectoryFile.cpp:656: "if (normalizeVertexPositions && !binLinesData.verticesNormalized)"
    if (!alreadyNormalized) {
        AABB3 aabb = computeTrajectoriesAABB3(loaded);
        normalizeTrajectoriesVertexPositions(loaded, aabb);
    }
    verticesNormalized = true; // what this object holds from here on (written back by saveTrajectoriesAsBinLines)
    setTrajectoryData(loaded, names, ribbons);
    return true;
}

bool LineData::renderThickBands = true;
float LineData::minBandThickness = 0.15f;
bool LineDataFlow::useRibbons = true;
bool LineDataFlow::useRotatingHelicityBands = false;
float LineDataFlow::separatorWidth = 0.2f;
bool LineDataFlow::useUniformTwistLineWidth = true;

// LineDataFlow::setNewSettings, LineDataFlow.cpp:584-610
bool LineDataFlow::setNewSettings(const SettingsMap& settings) {
    bool shallReloadGatherShader = LineData::setNewSettings(settings);
    bool b = useRibbons;
    if (settings.getValueOpt("use_ribbons", b) && b != useRibbons) {
        useRibbons = b;
        setTriangleRepresentationDirty();
        shallReloadGatherShader = true;
    }
    b = renderThickBands;
    if (settings.getValueOpt("thick_bands", b) && b != renderThickBands) {
        renderThickBands = b;
        dirty = true;
        shallReloadGatherShader = true;
    }
    float f = minBandThickness;
    if (settings.getValueOpt("min_band_thickness", f) && f != minBandThickness) {
        minBandThickness = f;
        dirty = true;
        shallReloadGatherShader = true;
    }
    // :601-672.  UNIFORM_HELICITY_BAND_WIDTH exists in the triangle closest-hit path and in the raster shaders (ClosestHitTubeAnalytic
    // does not pass rotationSeparatorScale, TubeRayTracing.glsl:512-613).  Twist-line texture: :626-672 (the file name key is the
    // embedder's business: setTwistLineTexture takes the decoded pixels)
    b = useTwistLineTexture;
    if (settings.getValueOpt("use_twist_line_texture", b) && b != useTwistLineTexture) { useTwistLineTexture = b; dirty = true; shallReloadGatherShader = true; }
    {
        static const char* const names[] = {"Nearest", "Linear", "Nearest Mipmap Nearest", "Linear Mipmap Nearest",
                                            "Nearest Mipmap Linear", "Linear Mipmap Linear"};
        std::string name;
        int idx = textureFilteringModeIndex;
        if (settings.getValueOpt("twist_line_texture_filtering_mode", name))
            for (int i = 0; i < 6; i++) if (name == names[i]) idx = i;
        settings.getValueOpt("twist_line_texture_filtering_mode_index", idx);
        if (idx >= 0 && idx < 6 && idx != textureFilteringModeIndex) { textureFilteringModeIndex = idx; dirty = true; }
    }
    b = useRotatingHelicityBands;
    if (settings.getValueOpt("rotating_helicity_bands", b) && b != useRotatingHelicityBands) {
        useRotatingHelicityBands = b;
        if (useRotatingHelicityBands) useRibbons = false;
        cachedAabbDataValid = false;
        setTriangleRepresentationDirty();
        dirty = true;
        shallReloadGatherShader = true;
    }
    b = useUniformTwistLineWidth;
    if (settings.getValueOpt("use_uniform_twist_line_width", b) && b != useUniformTwistLineWidth) {
        useUniformTwistLineWidth = b;
        dirty = true;
        shallReloadGatherShader = true;
    }
    f = separatorWidth;
    if (settings.getValueOpt("separator_width", f) && f != separatorWidth) { separatorWidth = f; dirty = true; }
    int n = int(numSubdivisionsBands);
    if (settings.getValueOpt("band_subdivisions", n) && n != int(numSubdivisionsBands) && n > 0) { numSubdivisionsBands = uint32_t(n); dirty = true; }
    f = helicityRotationFactor;
    if (settings.getValueOpt("helicity_rotation_factor", f) && f != helicityRotationFactor) { helicityRotationFactor = f; dirty = true; }
    return shallReloadGatherShader;
}

void LineDataFlow::setTwistLineTexture(const uint8_t* rgba8, uint32_t width, uint32_t height) {
    twistLineTexture.clear();
    twistW = twistH = 0;
    if (rgba8 && width && height) {
        twistLineTexture.assign(rgba8, rgba8 + size_t(width) * height * 4);
        twistW = width; twistH = height;
    }
    dirty = true;
}

// LineDataFlow.cpp:468-578 (flow lines: counts, per-attribute min/max, model AABB)
void LineDataFlow::setTrajectoryData(const Trajectories& newTrajectories, const std::vector<std::string>& names,
                                     const std::vector<std::vector<vec3>>& newRibbonsDirections) {
    trajectories = newTrajectories;
    ribbonsDirections = newRibbonsDirections;
    hasBandsData = !ribbonsDirections.empty(); // LineDataFlow.cpp:469
    useRibbons = !useRotatingHelicityBands && hasBandsData; // :470
    if (ribbonsDirections.empty() && (linePrimitiveMode == LINE_PRIMITIVES_RIBBON_QUADS_GEOMETRY_SHADER ||
                                      linePrimitiveMode == LINE_PRIMITIVES_TUBE_RIBBONS_GEOMETRY_SHADER))
        linePrimitiveMode = LINE_PRIMITIVES_QUADS_PROGRAMMABLE_PULL; // :471-474
    // "Use bands if possible", :476-481
    if (hasBandsData && !getUseBandRendering()) linePrimitiveMode = LINE_PRIMITIVES_TUBE_RIBBONS_PROGRAMMABLE_PULL;
    else if (!hasBandsData && getUseBandRendering()) linePrimitiveMode = LINE_PRIMITIVES_TUBE_PROGRAMMABLE_PULL;
    if (hasBandsData) tubeNumSubdivisions = std::max(tubeNumSubdivisions, 8); // :482-484
    numTotalTrajectories = trajectories.size();
    numTotalTrajectoryPoints = 0;
    for (const Trajectory& t : trajectories) numTotalTrajectoryPoints += t.positions.size();
    size_t numAttr = trajectories.empty() ? 0 : trajectories[0].attributes.size();
    attributeNames = names;
    for (size_t i = attributeNames.size(); i < numAttr; i++) attributeNames.push_back("Attribute #" + std::to_string(i + 1));
    minMaxAttributeValues.clear();
    for (size_t varIdx = 0; varIdx < numAttr; varIdx++) {
        float minAttr = std::numeric_limits<float>::max();
        float maxAttr = std::numeric_limits<float>::lowest();
        for (const Trajectory& t : trajectories)
            for (float val : t.attributes[varIdx]) { minAttr = std::min(minAttr, val); maxAttr = std::max(maxAttr, val); }
        if (attributeNames[varIdx] == "Helicity") { // a symmetric range for the signed quantity, :522-526
            const float maxAbs = std::max(std::abs(minAttr), std::abs(maxAttr));
            minAttr = -maxAbs;
            maxAttr = maxAbs;
        }
        minMaxAttributeValues.emplace_back(minAttr, maxAttr);
    }
    // :535-550: the first attribute whose lower-case name contains "helicity" drives the rotating helicity bands
    helicityAttributeIndex = -1;
    for (size_t attrIdx = 0; attrIdx < attributeNames.size() && attrIdx < numAttr; attrIdx++) {
        std::string lower = attributeNames[attrIdx];
        for (char& ch : lower) ch = char(std::tolower((unsigned char)ch));
        if (lower.find("helicity") != std::string::npos) { helicityAttributeIndex = int(attrIdx); break; }
    }
    hasHelicity = helicityAttributeIndex != -1;
    if (hasHelicity) {
        const auto& mm = minMaxAttributeValues[size_t(helicityAttributeIndex)];
        maxHelicity = std::max(std::abs(mm.first), std::abs(mm.second));
    } else {
        useRotatingHelicityBands = false;
    }
    modelBoundingBox = computeTrajectoriesAABB3(trajectories);
    cachedAabbDataValid = false;
    cachedTriangleDataValid = false;
    dataGeneration++;
    dirty = true;
}

// the arrays lv_set_trajectories takes; band data and the rotating helicity bands (ribbon normals / a rotation that runs on across all
// lines) keep the host-built render data
bool LineDataFlow::getTrajectoryArrays(std::vector<float>& positions, std::vector<float>& attribute, std::vector<uint32_t>& lineOffsets) {
    if ((useRibbons && hasBandsData) || getUseRotatingHelicityBands()) return false;
    if (numTotalTrajectoryPoints > 0x03FFFFFFu) return false;
    positions.resize(3 * numTotalTrajectoryPoints);
    attribute.assign(numTotalTrajectoryPoints, 0.0f);
    lineOffsets.assign(trajectories.size() + 1, 0u);
    size_t at = 0;
    for (size_t li = 0; li < trajectories.size(); li++) {
        const Trajectory& t = trajectories[li];
        const size_t n = t.positions.size();
        if (n) memcpy(positions.data() + 3 * at, t.positions.data(), n * sizeof(vec3));
        if (!t.attributes.empty() && size_t(selectedAttributeIndex) < t.attributes.size())
            memcpy(attribute.data() + at, t.attributes[size_t(selectedAttributeIndex)].data(), n * sizeof(float));
        at += n;
        lineOffsets[li + 1] = uint32_t(at);
    }
    return true;
}

size_t LineDataFlow::getNumLineSegments() {
    size_t n = 0;
    for (const Trajectory& t : trajectories) n += t.positions.empty() ? 0 : t.positions.size() - 1;
    return n;
}

std::vector<std::vector<vec3>> LineDataFlow::getFilteredLines(LineRenderer*) {
    std::vector<std::vector<vec3>> lines;
    lines.reserve(trajectories.size());
    for (const Trajectory& t : trajectories) lines.push_back(t.positions);
    return lines;
}

// getLinePassTubeAabbRenderData (LineDataFlow.cpp:2112-2277) as three passes over SoA scratch arrays instead of the reference's one
// append-as-you-go loop; the output is byte-identical to it (tests/test_host.py, tests/test_independent_restatement.py):
//   pass 1 (per line, data parallel over the points)   central-difference tangents, their lengths, the "keep this point" flags
//                                                      (|tangent| >= 1e-4, :2160) and the number of kept points per line
//   prefix sum over the lines                          where each line's records, index pairs and boxes go -- a line that keeps
//                                                      fewer than two points keeps none (:2209-2221)
//   pass 2 (per line, one short sequential chain)      the only true recurrence: the line normal is carried from kept point to
//                                                      kept point (Gram-Schmidt against the tangent with the fallback axes of
//                                                      :2171-2177; band data: cross(ribbon direction, tangent), :2166-2168) and
//                                                      the helicity rotation accumulates along the line (:2188-2197)
//   pass 3 (per line, data parallel)                   48-byte records, index pairs and padded boxes written in place
// Lines are independent, so every pass is an OpenMP loop over lines; nothing is concatenated afterwards.
TubeAabbRenderData LineDataFlow::getLinePassTubeAabbRenderData(bool /*isRasterizer*/, bool ellipticTubes) {
    const bool useRibbonNormals = ellipticTubes && useRibbons && hasBandsData;
    const float lineWidth = useRibbonNormals ? LineRenderer::getBandWidth() : LineRenderer::getLineWidth();
    const bool helicityBands = getUseRotatingHelicityBands();
    if (cachedAabbDataValid && cachedLineWidth == lineWidth && cachedEllipticTubes == useRibbonNormals &&
        cachedHelicityBands == helicityBands)
        return cachedTubeAabbRenderData;
    const vec3 pad(lineWidth * 0.5f);
    const size_t numLines = trajectories.size();

    // ---- pass 1: tangents + keep flags (SoA per line, flattened over all points)
    std::vector<size_t> pointOffset(numLines + 1, 0);
    for (size_t li = 0; li < numLines; li++) pointOffset[li + 1] = pointOffset[li] + trajectories[li].positions.size();
    std::vector<vec3> unitTangent(pointOffset[numLines]);
    std::vector<uint8_t> keep(pointOffset[numLines], 0);
    std::vector<uint32_t> keptPerLine(numLines, 0);
#pragma omp parallel for schedule(dynamic, 16)
    for (long lli = 0; lli < long(numLines); lli++) {
        const size_t li = size_t(lli);
        const std::vector<vec3>& P = trajectories[li].positions;
        const size_t n = P.size();
        if (n < 2) continue;
        vec3* T = unitTangent.data() + pointOffset[li];
        uint8_t* K = keep.data() + pointOffset[li];
        uint32_t kept = 0;
        for (size_t i = 0; i < n; i++) {
            const vec3& ahead = P[i + 1 < n ? i + 1 : i];
            const vec3& behind = P[i > 0 ? i - 1 : i];
            const vec3 d = ahead - behind;          // one-sided at the two ends, central in between
            const float len = length(d);
            const bool k = !(len < 0.0001f);
            K[i] = k ? 1 : 0;
            if (k) { T[i] = normalize(d); kept++; }
        }
        keptPerLine[li] = kept >= 2 ? kept : 0u;    // a tube of one point is dropped
    }

    // ---- where every line's output goes
    std::vector<uint32_t> recordOffset(numLines + 1, 0);
    std::vector<size_t> segmentOffset(numLines + 1, 0);
    for (size_t li = 0; li < numLines; li++) {
        recordOffset[li + 1] = recordOffset[li] + keptPerLine[li];
        segmentOffset[li + 1] = segmentOffset[li] + (keptPerLine[li] ? keptPerLine[li] - 1 : 0);
    }
    TubeAabbRenderData data;
    data.linePointDataBuffer.assign(recordOffset[numLines], LinePointDataUnified{});
    data.indexBuffer.assign(2 * segmentOffset[numLines], 0u);
    data.aabbBuffer.assign(segmentOffset[numLines], AABB3{});

#pragma omp parallel for schedule(dynamic, 16)
    for (long lli = 0; lli < long(numLines); lli++) {
        const size_t li = size_t(lli);
        const uint32_t kept = keptPerLine[li];
        if (!kept) continue;
        const Trajectory& trajectory = trajectories[li];
        const std::vector<vec3>& P = trajectory.positions;
        const size_t n = P.size();
        const vec3* T = unitTangent.data() + pointOffset[li];
        const uint8_t* K = keep.data() + pointOffset[li];
        LinePointDataUnified* rec = data.linePointDataBuffer.data() + recordOffset[li];
        const float* attribute = trajectory.attributes.empty() ? nullptr : trajectory.attributes[size_t(selectedAttributeIndex)].data();
        const float* helicity = helicityBands ? trajectory.attributes[size_t(helicityAttributeIndex)].data() : nullptr;
        // ---- pass 2: the carried quantities, kept points only
        vec3 carried(1.0f, 0.0f, 0.0f);             // lastLineNormal
        float rotation = 0.0f;
        uint32_t o = 0;
        for (size_t i = 0; i < n; i++) {
            if (!K[i]) continue;
            const vec3 t = T[i];
            if (useRibbonNormals) {
                carried = cross(ribbonsDirections[li][i], t);                  // not normalised, as in the reference
            } else {
                vec3 axis = carried;
                if (length(cross(axis, t)) < 0.01f) {                           // tangent (anti)parallel to the previous normal
                    axis = vec3(0.0f, 1.0f, 0.0f);
                    if (length(cross(axis, t)) < 0.01f) axis = vec3(0.0f, 0.0f, 1.0f);
                }
                carried = normalize(axis - dot(axis, t) * t);                   // Gram-Schmidt
            }
            LinePointDataUnified& r = rec[o++];
            r.lineNormal[0] = carried.x; r.lineNormal[1] = carried.y; r.lineNormal[2] = carried.z;
            if (helicityBands) {
                r.lineRotation = rotation;
                const float step = i + 1 < n ? length(P[i + 1] - P[i]) : 0.0f;
                rotation += helicity[i] / maxHelicity * 3.1415926535897932f * step / 0.005f;
            }
        }
        // ---- pass 3: the rest of the records, the index pairs and the padded boxes
        o = 0;
        const uint32_t base = recordOffset[li];
        uint32_t* idx = data.indexBuffer.data() + 2 * segmentOffset[li];
        AABB3* box = data.aabbBuffer.data() + segmentOffset[li];
        vec3 previous;
        for (size_t i = 0; i < n; i++) {
            if (!K[i]) continue;
            LinePointDataUnified& r = rec[o];
            r.linePosition[0] = P[i].x; r.linePosition[1] = P[i].y; r.linePosition[2] = P[i].z;
            r.lineAttribute = attribute ? attribute[i] : 0.0f;
            r.lineTangent[0] = T[i].x; r.lineTangent[1] = T[i].y; r.lineTangent[2] = T[i].z;
            if (o > 0) {
                idx[2 * (o - 1)] = base + o - 1;
                idx[2 * (o - 1) + 1] = base + o;
                box[o - 1].min = lv::min(previous, P[i]) - pad;
                box[o - 1].max = lv::max(previous, P[i]) + pad;
            }
            previous = P[i];
            o++;
        }
    }
    cachedTubeAabbRenderData = data;
    cachedAabbDataValid = true;
    cachedLineWidth = lineWidth;
    cachedEllipticTubes = useRibbonNormals;
    cachedHelicityBands = helicityBands;
    return data;
}

// LineDataFlow.cpp:1912-2110 for flow lines with capped tubes: tessellation (Tubes.cpp) + the line-point table the
// vertices refer to (:1996-2020, including the way lineStartIndex only advances when the trajectory index changes).
TubeTriangleRenderData LineDataFlow::getLinePassTubeTriangleMeshRenderData(bool /*isRasterizer*/, bool /*vulkanRayTracing*/) {
    // band data: the reference is in its ribbon primitive mode then (setTrajectoryData, LineDataFlow.cpp:476-481), so its triangle
    // mesh is the elliptic tessellation with semi-axes bandWidth / 2 * minBandThickness and bandWidth / 2 (:1949-1975)
    const bool bands = useRibbons && hasBandsData;
    const bool helicityBands = getUseRotatingHelicityBands();
    const float lineWidth = bands ? LineRenderer::getBandWidth() * minBandThickness : LineRenderer::getLineWidth();
    if (cachedTriangleDataValid && cachedTriangleLineWidth == lineWidth && cachedTriangleSubdivisions == tubeNumSubdivisions &&
        cachedTriangleBands == bands && (!bands || cachedTriangleBandWidth == LineRenderer::getBandWidth()) &&
        cachedTriangleHelicityBands == helicityBands)
        return cachedTubeTriangleRenderData;
    std::vector<std::vector<vec3>> lineCentersList(trajectories.size());
    for (size_t i = 0; i < trajectories.size(); i++) lineCentersList[i] = trajectories[i].positions;

    TubeTriangleRenderData data;
    std::vector<LinePointReference> linePointReferences;
    std::vector<vec3> lineTangents, lineNormals;
    if (bands) {
        const float binormalRadius = LineRenderer::getBandWidth() * 0.5f;
        const float normalRadius = binormalRadius * minBandThickness;
        createCappedTriangleEllipticTubesRenderData(lineCentersList, ribbonsDirections, normalRadius, binormalRadius,
                                                    tubeNumSubdivisions, data.indexBuffer, data.vertexBuffer, linePointReferences,
                                                    lineTangents, lineNormals);
    } else {
        createCappedTriangleTubesRenderData(lineCentersList, lineWidth * 0.5f, tubeNumSubdivisions, data.indexBuffer,
                                            data.vertexBuffer, linePointReferences, lineTangents, lineNormals);
    }

    data.linePointDataBuffer.resize(linePointReferences.size());
    uint32_t lineStartIndex = 0, lastTrajectoryIndex = 0;
    float rotation = 0.0f; // useRotatingHelicityBands: runs on across the trajectories here (:1994)
    for (size_t i = 0; i < linePointReferences.size(); i++) {
        const LinePointReference& ref = linePointReferences[i];
        const Trajectory& trajectory = trajectories[ref.trajectoryIndex];
        LinePointDataUnified& lp = data.linePointDataBuffer[i];
        memset(&lp, 0, sizeof(lp));
        if (helicityBands) { // :2014-2028
            lp.lineRotation = rotation;
            const float helicity = trajectory.attributes[size_t(helicityAttributeIndex)][ref.linePointIndex];
            float lineSegmentLength = 0.0f;
            if (i + 1 < linePointReferences.size() && linePointReferences[i + 1].trajectoryIndex == ref.trajectoryIndex)
                lineSegmentLength = length(trajectory.positions[linePointReferences[i + 1].linePointIndex] -
                                           trajectory.positions[ref.linePointIndex]);
            rotation += helicity / maxHelicity * 3.1415926535897932f * lineSegmentLength / 0.005f;
        }
        const vec3& p = trajectory.positions[ref.linePointIndex];
        lp.linePosition[0] = p.x; lp.linePosition[1] = p.y; lp.linePosition[2] = p.z;
        lp.lineAttribute = trajectory.attributes.empty()
                ? 0.0f : trajectory.attributes[size_t(selectedAttributeIndex)][ref.linePointIndex];
        lp.lineTangent[0] = lineTangents[i].x; lp.lineTangent[1] = lineTangents[i].y; lp.lineTangent[2] = lineTangents[i].z;
        lp.lineNormal[0] = lineNormals[i].x; lp.lineNormal[1] = lineNormals[i].y; lp.lineNormal[2] = lineNormals[i].z;
        if (lastTrajectoryIndex != ref.trajectoryIndex) {
            lastTrajectoryIndex = ref.trajectoryIndex;
            lineStartIndex = uint32_t(i);
        }
        lp.lineStartIndex = lineStartIndex;
    }
    cachedTubeTriangleRenderData = data;
    cachedTriangleDataValid = true;
    cachedTriangleLineWidth = lineWidth;
    cachedTriangleSubdivisions = tubeNumSubdivisions;
    cachedTriangleBands = bands;
    cachedTriangleHelicityBands = helicityBands;
    cachedTriangleBandWidth = LineRenderer::getBandWidth();
    return data;
}

} // namespace lv
