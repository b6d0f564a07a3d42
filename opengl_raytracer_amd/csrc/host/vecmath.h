// Minimal float vector/matrix types for the host library (the reference uses GLM for these).
#pragma once
#include <cmath>

namespace rth {

struct vec2 {
    float x = 0.0f, y = 0.0f;
    vec2() = default;
    vec2(float x_, float y_) : x(x_), y(y_) {}
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};

struct vec3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
    vec3() = default;
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    vec3(vec2 v, float z_) : x(v.x), y(v.y), z(z_) {}
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};

inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 &operator+=(vec3 &a, vec3 b) { a = a + b; return a; }
inline vec3 &operator-=(vec3 &a, vec3 b) { a = a - b; return a; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// glm::cross: (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y)
inline vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
// glm::normalize: v * inversesqrt(dot(v, v))
inline vec3 normalize(vec3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }

// Column-major 4x4, m[col][row] like glm::mat4.
struct mat4 {
    float m[4][4] = {};
    float *operator[](int c) { return m[c]; }
    const float *operator[](int c) const { return m[c]; }
    const float *data() const { return &m[0][0]; }
    static mat4 identity() {
        mat4 r;
        for (int i = 0; i < 4; i++) r.m[i][i] = 1.0f;
        return r;
    }
};

// glm::perspective (right-handed, fovy in radians, GL clip space).
inline mat4 perspective(float fovy, float aspect, float z_near, float z_far) {
    const float t = std::tan(fovy / 2.0f);
    mat4 r;
    r[0][0] = 1.0f / (aspect * t);
    r[1][1] = 1.0f / t;
    r[2][2] = -(z_far + z_near) / (z_far - z_near);
    r[2][3] = -1.0f;
    r[3][2] = -(2.0f * z_far * z_near) / (z_far - z_near);
    return r;
}

// glm::lookAt (right-handed).
inline mat4 look_at(vec3 eye, vec3 center, vec3 up) {
    const vec3 f = normalize(center - eye);
    const vec3 s = normalize(cross(f, up));
    const vec3 u = cross(s, f);
    mat4 r = mat4::identity();
    r[0][0] = s.x; r[1][0] = s.y; r[2][0] = s.z;
    r[0][1] = u.x; r[1][1] = u.y; r[2][1] = u.z;
    r[0][2] = -f.x; r[1][2] = -f.y; r[2][2] = -f.z;
    r[3][0] = -dot(s, eye);
    r[3][1] = -dot(u, eye);
    r[3][2] = dot(f, eye);
    return r;
}

}  // namespace rth
