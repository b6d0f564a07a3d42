#include "Scene.h"

#include <cmath>
#include <cstdio>

namespace rth {

vec2 point_on_circle(float cx, float cy, float radius, float angle) {
    return vec2(cx + radius * std::cos(angle), cy + radius * std::sin(angle));
}

void set_light_positions(rt_light *lights, int n, float angle_offset) {
    float angle = 0.0f;
    const float incr = static_cast<float>(2 * M_PI / RT_MAX_LIGHTS);
    for (int i = 0; i < n; i++) {
        const vec2 p = point_on_circle(0.0f, 0.0f, 2.0f, angle + angle_offset);
        lights[i].pos[0] = p.x;
        lights[i].pos[1] = p.y;
        lights[i].pos[2] = -2.0f;
        angle += incr;
    }
}

void default_lights(rt_light lights[RT_MAX_LIGHTS]) {
    set_light_positions(lights, RT_MAX_LIGHTS, 0.0f);
    MsvcRand rng;
    for (int i = 0; i < RT_MAX_LIGHTS; i++) {
        for (int c = 0; c < 3; c++) lights[i].color[c] = static_cast<float>(rng.rand0to1());
        lights[i].pad0 = -1.0f;
        lights[i].pad1 = -1.0f;
    }
}

float light_angle_offset(double t) {
    const double a = std::sin(t);
    return static_cast<float>(0.0 * (1.0 - a) + (2 * M_PI / RT_MAX_LIGHTS) * a);  // glm::lerp = mix: x(1-a) + y a
}

bool csv_append(const char *filename, int tgs, int width, int height, int passes, int num_lights, int num_triangles,
                int raygen_ms, int intersect_ms, int color_ms, std::string &err) {
    FILE *f = std::fopen(filename, "a");
    if (!f) {
        err = std::string("cannot open for appending: ") + filename;
        return false;
    }
    std::fprintf(f, "%d,%d,%d,%d,%d,%d,%d,%d,%d\n", tgs, width, height, passes, num_lights, num_triangles, raygen_ms,
                 intersect_ms, color_ms);
    if (std::fclose(f) != 0) {
        err = std::string("write failed: ") + filename;
        return false;
    }
    return true;
}

void draw_light_markers(uint8_t *rgba, int w, int h, const mat4 &view, const mat4 &proj, const rt_light *lights, int n,
                        float point_size) {
    for (int l = 0; l < n; l++) {
        const float p[4] = {lights[l].pos[0], lights[l].pos[1], lights[l].pos[2], 1.0f};
        float e[4], c[4];
        for (int r = 0; r < 4; r++) e[r] = view[0][r] * p[0] + view[1][r] * p[1] + view[2][r] * p[2] + view[3][r] * p[3];
        for (int r = 0; r < 4; r++) c[r] = proj[0][r] * e[0] + proj[1][r] * e[1] + proj[2][r] * e[2] + proj[3][r] * e[3];
        // Points are clipped as points: the centre must lie inside the clip volume.
        if (!(c[3] > 0.0f) || std::fabs(c[0]) > c[3] || std::fabs(c[1]) > c[3] || std::fabs(c[2]) > c[3]) continue;
        const float cx = (c[0] / c[3] * 0.5f + 0.5f) * static_cast<float>(w);
        const float cy = (c[1] / c[3] * 0.5f + 0.5f) * static_cast<float>(h);
        const float half = point_size * 0.5f;
        // A pixel belongs to the point when its centre lies inside the square.
        const int x0 = static_cast<int>(std::ceil(cx - half - 0.5f)), x1 = static_cast<int>(std::ceil(cx + half - 0.5f));
        const int y0 = static_cast<int>(std::ceil(cy - half - 0.5f)), y1 = static_cast<int>(std::ceil(cy + half - 0.5f));
        for (int y = y0 < 0 ? 0 : y0; y < y1 && y < h; y++) {
            for (int x = x0 < 0 ? 0 : x0; x < x1 && x < w; x++) {
                uint8_t *px = rgba + (static_cast<size_t>(y) * w + x) * 4;
                px[0] = 255; px[1] = 255; px[2] = 0; px[3] = 255;
            }
        }
    }
}

}  // namespace rth
