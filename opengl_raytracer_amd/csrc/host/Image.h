// RGBA8 images for the host library: TGA/PNG decode, TGA/PNG encode.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rth {

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> rgba;  // h rows of w RGBA8 texels; row 0 = top unless loaded with invert_y

    // invert_y = true gives the bottom-up row order SOIL_FLAG_INVERT_Y produces
    // ("OpenGL Raytracer/main.cpp":391-398), which is what rt_set_texture_rgba8 expects.
    bool load(const std::string &path, bool invert_y, std::string &err);
    bool decode(const uint8_t *bytes, int64_t len, bool invert_y, std::string &err);
    void flip_rows();
};

// `rgba_bottom_up`: h rows of w RGBA8 pixels, row 0 = bottom (rt_read_pixels_rgba8 order).
bool write_tga(const char *path, const uint8_t *rgba_bottom_up, int w, int h, bool rle, std::string &err);
bool write_png(const char *path, const uint8_t *rgba_bottom_up, int w, int h, std::string &err);

}  // namespace rth
