// Headless Linux front end with the reference's frame loop ("OpenGL Raytracer/main.cpp":287-603):
// the same set-up order, the same three-stage loop with its pass rewind, the same CSV line,
// with rt_raygen / rt_intersect / rt_color where the reference calls glDispatchCompute and a
// frame file where it swaps a window's buffers.
//
//   rt_raytracer [--scene sword|sword_in_box|bunny] [--obj DIR FILE] [--texture FILE] [--assets DIR]
//                [--width W] [--height H] [--tgs N] [--passes P] [--lights L] [--frames F]
//                [--csv FILE] [--sweep resolution|threadgroup|renderpasses|lights|all]
//                [--out FRAME.png|.tga] [--markers] [--time-step SECONDS] [--device D]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/rt.h"
#include "../../../include/rt_host.h"
#include "../host/Camera.h"
#include "../host/Image.h"
#include "../host/Mesh.h"
#include "../host/Scene.h"

namespace {

const int UNLIMITED_FRAMES = 0;  // main.cpp:23; without a window this means "one frame"

struct Options {
    std::string assets = "assets/";
    std::string objDir, objFile, texture;
    std::string out;
    std::string extendedCsv;  // float times and work counters next to the reference's columns
    bool markers = false;
    double timeStep = 1.0 / 60.0;
    int device = 0;
} gOptions;

int gRenderPasses;  // main.cpp:26

bool ends_with(const std::string &s, const char *suffix) {
    const size_t n = std::strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

#define RT_CHECK(call)                                                                     \
    do {                                                                                   \
        const int rc_ = (call);                                                            \
        if (rc_ != RT_OK) {                                                                \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, rt_last_error(rt));   \
            if (rt) rt_destroy(rt);                                                        \
            return -1;                                                                     \
        }                                                                                  \
    } while (0)

int RunRaytracer(int windowWidth, int windowHeight, int threadGroupSize, int renderPasses, int numLights,
                 int numFrames, const char *benchmarkOutputFile = nullptr) {
    gRenderPasses = renderPasses;
    rt_ctx *rt = nullptr;
    // the output image and the ray buffer (main.cpp:331-337, 368-372)
    RT_CHECK(rt_create(&rt, gOptions.device, windowWidth, windowHeight));

    std::vector<rt_light> lightData(RT_MAX_LIGHTS);
    rth::default_lights(lightData.data());  // main.cpp:378-386

    {  // main.cpp:391-401
        rth::Image tex;
        std::string err;
        if (!tex.load(gOptions.texture, /*invert_y=*/true, err)) {
            std::printf("texture loading error: '%s'\n", err.c_str());
            const uint8_t white[4] = {255, 255, 255, 255};
            RT_CHECK(rt_set_texture_rgba8(rt, white, 1, 1));
        } else {
            RT_CHECK(rt_set_texture_rgba8(rt, tex.rgba.data(), tex.w, tex.h));
        }
    }

    Camera camera;  // main.cpp:402-406
    camera.setPosition(2.77904f, 0.594951f, -1.95046f);
    camera.setVerticalAngle(24.82f);
    camera.setHorizontalAngle(-82.44f);
    camera.aspectRatio = static_cast<float>(windowHeight) / static_cast<float>(windowWidth);

    Mesh swordMesh;  // main.cpp:408-414
    if (!swordMesh.LoadFromObjFile(gOptions.objDir, gOptions.objFile) || swordMesh.GetGroupCount() == 0) {
        std::fprintf(stderr, "%s\n", swordMesh.GetError().empty() ? "mesh has no groups" : swordMesh.GetError().c_str());
        rt_destroy(rt);
        return -1;
    }
    const int numVertices = swordMesh.GetVertexCount(0);
    {
        std::vector<rt_vertex> packed(numVertices);
        rth_pack_vertices(reinterpret_cast<const rth_vertex_data *>(swordMesh.GetVertexData(0)), numVertices,
                          packed.data());
        RT_CHECK(rt_set_mesh(rt, packed.data(), numVertices));
    }

    camera.cameraMouseControlEnabled = false;  // main.cpp:422: no camera input in benchmark mode

    std::vector<uint8_t> framePixels(static_cast<size_t>(windowWidth) * windowHeight * 4);
    const int frames = numFrames == UNLIMITED_FRAMES ? 1 : numFrames;
    for (int frame = 0; frame < frames; frame++) {
        CameraInput input;
        input.time = frame * gOptions.timeStep;  // stands in for glfwGetTime()
        camera.Update(input);

        // the lights swing around the mesh (main.cpp:449)
        rth::set_light_positions(lightData.data(), RT_MAX_LIGHTS, rth::light_angle_offset(input.time));

        RT_CHECK(rt_set_lights(rt, lightData.data(), numLights));  // main.cpp:506-512
        const rth::vec3 p = camera.getPosition(), d = camera.getDirection(), u = camera.getUp(), r = camera.getRight();
        RT_CHECK(rt_set_camera(rt, &p.x, &d.x, &u.x, &r.x));  // main.cpp:497-500

        int passes = gRenderPasses;
        std::vector<int> time(3);
        /* Raytracer stuff: main.cpp:479-529 */
        for (int i = 0; i < 3; i++) {
            switch (i) {
                case 0: RT_CHECK(rt_raygen(rt)); break;
                case 1: RT_CHECK(rt_intersect(rt)); break;
                case 2: RT_CHECK(rt_color(rt)); break;
            }
            if (passes > 1 && i == 2) {
                // one entire pass done: start over with the intersection stage
                i = 0;
                passes--;
            }
        }
        float ms[3];
        RT_CHECK(rt_stage_ms(rt, ms));  // OpenGLTimer, main.cpp:481-482, 520-521
        for (int i = 0; i < 3; i++) time[i] = static_cast<int>(ms[i]);

        // presentation: main.cpp:531-557
        if (!gOptions.out.empty() && frame == frames - 1) {
            RT_CHECK(rt_read_pixels_rgba8(rt, framePixels.data()));
            if (gOptions.markers)
                rth::draw_light_markers(framePixels.data(), windowWidth, windowHeight, camera.getViewMatrix(),
                                        camera.getProjectionMatrix(), lightData.data(), numLights, 30.0f);
            std::string err;
            const bool ok = ends_with(gOptions.out, ".tga")
                                ? rth::write_tga(gOptions.out.c_str(), framePixels.data(), windowWidth, windowHeight, true, err)
                                : rth::write_png(gOptions.out.c_str(), framePixels.data(), windowWidth, windowHeight, err);
            if (!ok) std::fprintf(stderr, "%s\n", err.c_str());
        }

        if (benchmarkOutputFile) {  // main.cpp:560-565
            std::string err;
            if (!rth::csv_append(benchmarkOutputFile, threadGroupSize, windowWidth, windowHeight, gRenderPasses, numLights,
                                 numVertices / 3, time[0], time[1], time[2], err))
                std::fprintf(stderr, "%s\n", err.c_str());
        }
        uint64_t counters[RT_NUM_COUNTERS];
        RT_CHECK(rt_get_counters(rt, counters));
        const double total = ms[0] + ms[1] + ms[2];
        const double rays = static_cast<double>(windowWidth) * windowHeight * gRenderPasses;
        if (!gOptions.extendedCsv.empty()) {
            FILE *f = std::fopen(gOptions.extendedCsv.c_str(), "a");
            if (f) {
                std::fprintf(f, "%d,%d,%d,%d,%d,%d,%.4f,%.4f,%.4f,%.3f,%llu,%llu,%llu,%llu\n", threadGroupSize, windowWidth,
                             windowHeight, gRenderPasses, numLights, numVertices / 3, ms[0], ms[1], ms[2],
                             rays / (total * 1e3), static_cast<unsigned long long>(counters[0]),
                             static_cast<unsigned long long>(counters[1]), static_cast<unsigned long long>(counters[2]),
                             static_cast<unsigned long long>(counters[3]));
                std::fclose(f);
            }
        }
        std::printf("frame %d: %dx%d tgs %d passes %d lights %d triangles %d: raygen %.3f ms, intersect %.3f ms, "
                    "color %.3f ms (%.2f Mrays/s)\n",
                    frame, windowWidth, windowHeight, threadGroupSize, gRenderPasses, numLights, numVertices / 3, ms[0],
                    ms[1], ms[2], rays / (total * 1e3));
    }
    rt_destroy(rt);
    return 0;
}

// The sweeps of main.cpp:615-644, file names included.
int RunSweep(const std::string &which) {
    int rc = 0;
    const bool all = which == "all";
    if (all || which == "resolution")
        for (int s : {400, 800, 1200}) rc |= RunRaytracer(s, s, 32, 2, 2, 5, "resolution_results.csv");
    if (all || which == "threadgroup")
        for (int tgs = 16; tgs <= 128; tgs += 16) rc |= RunRaytracer(800, 800, tgs, 2, 2, 5, "treadgroup_results.csv");
    if (all || which == "renderpasses")
        for (int p = 1; p <= 3; p++) rc |= RunRaytracer(800, 800, 32, p, 2, 5, "renderpasses_results.csv");
    if (all || which == "lights")
        for (int l = 0; l <= 4; l++) rc |= RunRaytracer(800, 800, 32, 2, l, 5, "lights_results.csv");
    return rc;
}

}  // namespace

int main(int argc, char **argv) {
    int width = 800, height = 800, tgs = 64, passes = 1, lights = 1, frames = 1;  // main.cpp:607
    std::string csv, sweep, scene = "sword";
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "%s needs a value\n", a.c_str());
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "--width") width = std::atoi(next());
        else if (a == "--height") height = std::atoi(next());
        else if (a == "--tgs") tgs = std::atoi(next());
        else if (a == "--passes") passes = std::atoi(next());
        else if (a == "--lights") lights = std::atoi(next());
        else if (a == "--frames") frames = std::atoi(next());
        else if (a == "--csv") csv = next();
        else if (a == "--extended-csv") gOptions.extendedCsv = next();
        else if (a == "--sweep") sweep = next();
        else if (a == "--scene") scene = next();
        else if (a == "--obj") { gOptions.objDir = next(); gOptions.objFile = next(); }
        else if (a == "--texture") gOptions.texture = next();
        else if (a == "--assets") gOptions.assets = next();
        else if (a == "--out") gOptions.out = next();
        else if (a == "--markers") gOptions.markers = true;
        else if (a == "--time-step") gOptions.timeStep = std::atof(next());
        else if (a == "--device") gOptions.device = std::atoi(next());
        else {
            std::fprintf(stderr, "unknown option %s\n", a.c_str());
            return 2;
        }
    }
    if (!gOptions.assets.empty() && gOptions.assets.back() != '/') gOptions.assets += '/';
    if (gOptions.objFile.empty()) {
        if (scene == "sword") { gOptions.objDir = gOptions.assets + "sword/"; gOptions.objFile = "sword_fixed.obj"; }
        else if (scene == "sword_in_box") { gOptions.objDir = gOptions.assets + "sword/"; gOptions.objFile = "sword_in_box.obj"; }
        else if (scene == "bunny") { gOptions.objDir = gOptions.assets + "bunny/"; gOptions.objFile = "bunny.obj"; }
        else { std::fprintf(stderr, "unknown scene %s\n", scene.c_str()); return 2; }
    }
    if (gOptions.texture.empty()) gOptions.texture = gOptions.assets + "sword/sword.png";  // main.cpp:393
    if (!sweep.empty()) return RunSweep(sweep) == 0 ? 0 : 1;
    return RunRaytracer(width, height, tgs, passes, lights, frames, csv.empty() ? nullptr : csv.c_str()) == 0 ? 0 : 1;
}
