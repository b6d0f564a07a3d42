// rt_exact.h -- device restatement of the reference's per-ray arithmetic.
//
// Every expression is associated as the GLSL text associates it and evaluated in
// IEEE binary32 with separately rounded multiplies and adds (the translation unit
// is built with -ffp-contract=off; hipcc expands `/` and __builtin_sqrtf to correctly
// rounded sequences -- __fsqrt_rn does NOT: it is a bare v_sqrt_f32, 1 ulp).  Built-ins whose precision GLSL leaves to the implementation are
// fixed as:  normalize(v) = v * (1 / sqrt(v.v));  distance = sqrt of the dot of
// the difference;  reflect(I,N) = I - (2 (N.I)) N;  pow(x,30) by repeated
// squaring;  max(x,0) = x < 0 ? 0 : x.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtx {

struct v3 {
    float x, y, z;
};

__device__ __forceinline__ v3 V(float x, float y, float z) { return v3{x, y, z}; }
__device__ __forceinline__ v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 muls(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ v3 cross(v3 a, v3 b)
{
    return V(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
__device__ __forceinline__ v3 normalize(v3 a) { return muls(a, 1.0f / __builtin_sqrtf(dot(a, a))); }
__device__ __forceinline__ float distance3(v3 a, v3 b)
{
    v3 d = sub(a, b);
    return __builtin_sqrtf(dot(d, d));
}
__device__ __forceinline__ float max0(float x) { return (x < 0.0f) ? 0.0f : x; }
__device__ __forceinline__ float pow30(float x)
{
    float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8;
    return ((x16 * x8) * x4) * x2;
}

// "OpenGL Raytracer/trace.glsl":1-22
__device__ __forceinline__ bool ray_vs_triangle(v3 o, v3 d, v3 tx, v3 ty, v3 tz, float &t)
{
    v3 e1 = sub(ty, tx);
    v3 e2 = sub(tz, tx);
    v3 q = cross(d, e2);
    float a = dot(e1, q);
    if (a > -1e-20f && a < 1e-20f) return false;
    float f = 1.0f / a;
    v3 s = sub(o, tx);
    float u = f * dot(s, q);
    if (u < 0.0f || u > 1.0f) return false;
    v3 r = cross(s, e1);
    float v = f * dot(d, r);
    if (v < 0.0f || u + v > 1.0f) return false;
    t = f * dot(e2, r);
    return true;
}

// "OpenGL Raytracer/trace.glsl":57-78
__device__ __forceinline__ bool ray_sphere(v3 o, v3 d, v3 c, float r, float &t)
{
    v3 l = sub(c, o);
    float s = dot(l, d);
    float l_sqr = dot(l, l);
    float r_sqr = r * r;
    bool outside = l_sqr > r_sqr;
    if (s < 0.0f && outside) return false;
    float m_sqr = l_sqr - s * s;
    if (m_sqr > r_sqr) return false;
    float q = __builtin_sqrtf(r_sqr - m_sqr);
    t = outside ? (s - q) : (s + q);
    return true;
}

// the two spheres of "OpenGL Raytracer/trace.glsl":159-162
__device__ __forceinline__ v3 sphere_center(int i)
{
    return i == 0 ? V(0.0f, -0.7f, 2.0f) : V(-2.0f, -0.0f, 0.0f);
}
__device__ __forceinline__ float sphere_radius(int i) { return i == 0 ? 0.5f : 0.2f; }

__device__ __forceinline__ uint32_t to_unorm8(float c)
{
    if (!(c == c)) return 0u;
    c = c < 0.0f ? 0.0f : c;
    c = c > 1.0f ? 1.0f : c;
    return (uint32_t)__float2int_rn(c * 255.0f);
}

} // namespace rtx
