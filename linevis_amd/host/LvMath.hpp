// LvMath.hpp -- minimal vector/matrix types for the headless host layer (GLM is not a dependency).
// float32 with a fixed evaluation order; the host library is compiled with -ffp-contract=off.
#pragma once

#include <cmath>
#include <cstdint>

namespace lv {

struct vec3 {
    float x, y, z; // trivially copyable: arrays of vec3 are memcpy-compatible with float[3] records
    vec3() = default;
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};
struct vec4 {
    float x, y, z, w;
    vec4() = default;
    vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
};

inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator/(float s, vec3 a) { return vec3(s / a.x, s / a.y, s / a.z); }
inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline vec3 normalize(vec3 a) { float l = length(a); return vec3(a.x / l, a.y / l, a.z / l); }
inline vec3 min(vec3 a, vec3 b) { return vec3(std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)); }
inline vec3 max(vec3 a, vec3 b) { return vec3(std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)); }

struct AABB3 {
    vec3 min = vec3(3.0e38f), max = vec3(-3.0e38f);
    vec3 getCenter() const { return (min + max) * 0.5f; }       // note: oracle/TrajectoryFile semantics use /2
    vec3 getDimensions() const { return max - min; }
    void combine(vec3 p) { min = lv::min(min, p); max = lv::max(max, p); }
};

// column-major 4x4 (GLM layout): m[col * 4 + row]
struct mat4 {
    float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    const float* data() const { return m; }
};

// glm::lookAtRH
inline mat4 lookAt(vec3 eye, vec3 center, vec3 up) {
    vec3 f = normalize(center - eye);
    vec3 s = normalize(cross(f, up));
    vec3 u = cross(s, f);
    mat4 r;
    r.m[0] = s.x; r.m[4] = s.y; r.m[8] = s.z;
    r.m[1] = u.x; r.m[5] = u.y; r.m[9] = u.z;
    r.m[2] = -f.x; r.m[6] = -f.y; r.m[10] = -f.z;
    r.m[12] = -dot(s, eye); r.m[13] = -dot(u, eye); r.m[14] = dot(f, eye);
    return r;
}

// glm::perspectiveRH_ZO with the y axis flipped (image row 0 = top); the convention this build owns.
inline mat4 perspectiveVulkan(float fovy, float aspect, float zNear, float zFar) {
    const float t = std::tan(fovy / 2.0f);
    mat4 r;
    for (float& v : r.m) v = 0.0f;
    r.m[0] = 1.0f / (aspect * t);
    r.m[5] = -1.0f / t;
    r.m[10] = zFar / (zNear - zFar);
    r.m[11] = -1.0f;
    r.m[14] = -(zFar * zNear) / (zFar - zNear);
    return r;
}

} // namespace lv
