// Light rig, benchmark CSV line and light-marker overlay of the reference's frame loop.
#pragma once
#include <cstdint>
#include <string>

#include "../../../include/rt.h"
#include "vecmath.h"

namespace rth {

// rand() of the Microsoft C runtime, which the reference (a Win32 program that never calls
// srand) draws its light colours from: x = x * 214013 + 2531011; return (x >> 16) & 0x7fff.
struct MsvcRand {
    uint32_t state = 1;
    static constexpr int kRandMax = 0x7fff;
    int next() {
        state = state * 214013u + 2531011u;
        return static_cast<int>((state >> 16) & 0x7fff);
    }
    // Rand0To1(), "OpenGL Raytracer/main.cpp":271-274
    double rand0to1() { return static_cast<double>(next()) / kRandMax; }
};

// PointOnCircle, main.cpp:266-269.
vec2 point_on_circle(float cx, float cy, float radius, float angle);
// SetLightPosition, main.cpp:276-285 (radius 2, z = -2, angle step 2 pi / RT_MAX_LIGHTS).
void set_light_positions(rt_light *lights, int n, float angle_offset);
// The rig RunRaytracer builds before its first frame, main.cpp:378-386.
void default_lights(rt_light lights[RT_MAX_LIGHTS]);
// glm::lerp(0.0, 2 pi / RT_MAX_LIGHTS, sin(t)) in double, narrowed to the float parameter of
// SetLightPosition (main.cpp:449).
float light_angle_offset(double time_seconds);

// WriteBenchmarkResultsToCSVFile, main.cpp:209-219.
bool csv_append(const char *filename, int tgs, int width, int height, int passes, int num_lights, int num_triangles,
                int raygen_ms, int intersect_ms, int color_ms, std::string &err);

// The forward pass of main.cpp:544-551 with simple_vert.glsl / simple_frag.glsl, rasterised on
// the CPU into a bottom-up RGBA8 frame: one opaque yellow square per light.
void draw_light_markers(uint8_t *rgba, int w, int h, const mat4 &view, const mat4 &projection, const rt_light *lights,
                        int n, float point_size);

}  // namespace rth
