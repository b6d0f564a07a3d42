// C ABI of the host library (include/rt_host.h) over Mesh, Camera, Image and Scene.
#include "../../../include/rt_host.h"

#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "Camera.h"
#include "Image.h"
#include "Mesh.h"
#include "Scene.h"

struct rth_mesh {
    Mesh mesh;
};

namespace {

thread_local std::string g_error;

int fail(const std::string &what) {
    g_error = what;
    return -1;
}

void copy_str(char *dst, size_t cap, const std::string &s) {
    const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    std::memcpy(dst, s.data(), n);
    dst[n] = 0;
}

void to_c(const MeshMaterialData &m, rth_material *out) {
    std::memset(out, 0, sizeof *out);
    copy_str(out->name, sizeof out->name, m.Name);
    out->illum = m.Illum;
    for (int i = 0; i < 3; i++) {
        out->Kd[i] = m.Kd[i];
        out->Ka[i] = m.Ka[i];
        out->Tf[i] = m.Tf[i];
    }
    copy_str(out->map_Kd, sizeof out->map_Kd, m.map_Kd);
    out->Ni = m.Ni;
}

bool group_ok(const rth_mesh *m, int g) { return m && g >= 0 && g < m->mesh.GetGroupCount(); }

rth::mat4 to_mat(const float m[16]) {
    rth::mat4 r;
    std::memcpy(&r.m[0][0], m, sizeof(float) * 16);
    return r;
}

}  // namespace

extern "C" {

int rth_mesh_load(rth_mesh **out, const char *dir, const char *filename) {
    if (!out) return fail("rth_mesh_load: out is NULL");
    *out = nullptr;
    if (!dir || !filename) return fail("rth_mesh_load: dir or filename is NULL");
    rth_mesh *m = new (std::nothrow) rth_mesh;
    if (!m) return fail("rth_mesh_load: out of memory");
    try {
        if (!m->mesh.LoadFromObjFile(dir, filename)) {
            g_error = m->mesh.GetError();
            delete m;
            return -1;
        }
    } catch (const std::exception &e) {
        delete m;
        return fail(std::string("rth_mesh_load: ") + e.what());
    }
    *out = m;
    return 0;
}

void rth_mesh_free(rth_mesh *m) { delete m; }

int rth_mesh_group_count(const rth_mesh *m) { return m ? m->mesh.GetGroupCount() : 0; }

const char *rth_mesh_group_name(const rth_mesh *m, int g) {
    return group_ok(m, g) ? m->mesh.GetGroupName(g).c_str() : nullptr;
}

int rth_mesh_vertex_count(const rth_mesh *m, int g) { return group_ok(m, g) ? m->mesh.GetVertexCount(g) : 0; }

int rth_mesh_vertex_data_size(const rth_mesh *m, int g) { return group_ok(m, g) ? m->mesh.GetVertexDataSize(g) : 0; }

const rth_vertex_data *rth_mesh_vertex_data(const rth_mesh *m, int g) {
    static_assert(sizeof(rth_vertex_data) == sizeof(VertexData), "C and C++ vertex records must match");
    return group_ok(m, g) ? reinterpret_cast<const rth_vertex_data *>(m->mesh.GetVertexData(g)) : nullptr;
}

int rth_mesh_material(const rth_mesh *m, int g, rth_material *out) {
    if (!group_ok(m, g) || !out) return fail("rth_mesh_material: bad mesh, group or out");
    to_c(m->mesh.GetGroupMaterialData(g), out);
    return 0;
}

int rth_load_materials(const char *filepath, rth_material *out, int max) {
    if (!filepath || (max > 0 && !out)) return fail("rth_load_materials: bad arguments");
    Mesh loader;
    std::vector<MeshMaterialData> data;
    try {
        if (!loader.LoadMaterialData(filepath, data)) return fail(loader.GetError());
    } catch (const std::exception &e) {
        return fail(std::string("rth_load_materials: ") + e.what());
    }
    for (int i = 0; i < max && i < static_cast<int>(data.size()); i++) to_c(data[i], &out[i]);
    return static_cast<int>(data.size());
}

void rth_pack_vertices(const rth_vertex_data *in, int count, rt_vertex *out) {
    for (int i = 0; i < count; i++) {
        out[i].pos[0] = in[i].point[0];
        out[i].pos[1] = in[i].point[1];
        out[i].pos[2] = in[i].point[2];
        out[i].pad0 = 0.0f;
        out[i].texCoord[0] = in[i].texCoord[0];
        out[i].texCoord[1] = in[i].texCoord[1];
        out[i].pad1[0] = 0.0f;
        out[i].pad1[1] = 0.0f;
    }
}

static int hand_over(rth::Image &img, int *w, int *h, uint8_t **rgba) {
    uint8_t *p = static_cast<uint8_t *>(std::malloc(img.rgba.size()));
    if (!p) return fail("rth_image_load: out of memory");
    std::memcpy(p, img.rgba.data(), img.rgba.size());
    *w = img.w;
    *h = img.h;
    *rgba = p;
    return 0;
}

int rth_image_load(const char *path, int invert_y, int *w, int *h, uint8_t **rgba) {
    if (!path || !w || !h || !rgba) return fail("rth_image_load: bad arguments");
    *rgba = nullptr;
    try {
        rth::Image img;
        std::string err;
        if (!img.load(path, invert_y != 0, err)) return fail(err);
        return hand_over(img, w, h, rgba);
    } catch (const std::exception &e) {
        return fail(std::string("rth_image_load: ") + e.what());
    }
}

int rth_image_load_memory(const uint8_t *bytes, int64_t len, int invert_y, int *w, int *h, uint8_t **rgba) {
    if (!bytes || len < 0 || !w || !h || !rgba) return fail("rth_image_load_memory: bad arguments");
    *rgba = nullptr;
    try {
        rth::Image img;
        std::string err;
        if (!img.decode(bytes, len, invert_y != 0, err)) return fail(err);
        return hand_over(img, w, h, rgba);
    } catch (const std::exception &e) {
        return fail(std::string("rth_image_load_memory: ") + e.what());
    }
}

void rth_image_free(uint8_t *rgba) { std::free(rgba); }

int rth_write_tga(const char *path, const uint8_t *rgba, int w, int h, int rle) {
    std::string err;
    if (!path) return fail("rth_write_tga: path is NULL");
    try {
        return rth::write_tga(path, rgba, w, h, rle != 0, err) ? 0 : fail(err);
    } catch (const std::exception &e) {
        return fail(std::string("rth_write_tga: ") + e.what());
    }
}

int rth_write_png(const char *path, const uint8_t *rgba, int w, int h) {
    std::string err;
    if (!path) return fail("rth_write_png: path is NULL");
    try {
        return rth::write_png(path, rgba, w, h, err) ? 0 : fail(err);
    } catch (const std::exception &e) {
        return fail(std::string("rth_write_png: ") + e.what());
    }
}

void rth_camera_basis(float horizontal_angle, float vertical_angle, float direction[3], float right[3], float up[3]) {
    rth::vec3 d, r, u;
    rth::camera_basis(horizontal_angle, vertical_angle, d, r, u);
    for (int i = 0; i < 3; i++) {
        direction[i] = d[i];
        right[i] = r[i];
        up[i] = u[i];
    }
}

void rth_perspective(float fovy, float aspect, float z_near, float z_far, float m[16]) {
    const rth::mat4 r = rth::perspective(fovy, aspect, z_near, z_far);
    std::memcpy(m, r.data(), sizeof(float) * 16);
}

void rth_look_at(const float eye[3], const float center[3], const float up[3], float m[16]) {
    const rth::mat4 r = rth::look_at(rth::vec3(eye[0], eye[1], eye[2]), rth::vec3(center[0], center[1], center[2]),
                                     rth::vec3(up[0], up[1], up[2]));
    std::memcpy(m, r.data(), sizeof(float) * 16);
}

void rth_default_lights(rt_light lights[RT_MAX_LIGHTS]) { rth::default_lights(lights); }

void rth_set_light_positions(rt_light *lights, int n, float angle_offset) {
    rth::set_light_positions(lights, n, angle_offset);
}

float rth_light_angle_offset(double time_seconds) { return rth::light_angle_offset(time_seconds); }

int rth_csv_append(const char *filename, int tgs, int width, int height, int passes, int num_lights, int num_triangles,
                   int raygen_ms, int intersect_ms, int color_ms) {
    if (!filename) return fail("rth_csv_append: filename is NULL");
    std::string err;
    return rth::csv_append(filename, tgs, width, height, passes, num_lights, num_triangles, raygen_ms, intersect_ms,
                           color_ms, err)
               ? 0
               : fail(err);
}

void rth_draw_light_markers(uint8_t *rgba, int w, int h, const float view[16], const float projection[16],
                            const rt_light *lights, int n, float point_size) {
    if (!rgba || w < 1 || h < 1 || !view || !projection || !lights) return;
    rth::draw_light_markers(rgba, w, h, to_mat(view), to_mat(projection), lights, n, point_size);
}

const char *rth_last_error(void) { return g_error.c_str(); }

}  // extern "C"
