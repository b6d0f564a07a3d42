/*
 * rt_host.h -- C ABI of the host-side scene library that feeds rt.h:
 * OBJ/MTL meshes, TGA/PNG textures, the fly camera's basis and matrices, the
 * default light rig, the benchmark CSV line and the presentation tail (light
 * markers, frame files).  Plain CPU code (librt_host.so needs no GPU).
 *
 * Each entry point cites the part of the reference it stands in for; the C++
 * classes with the reference's own names (Mesh, Camera) are in
 * opengl_raytracer_amd/csrc/host/ and are thin owners of the same code.
 */
#ifndef RT_HOST_H_
#define RT_HOST_H_

#include <stdint.h>
#include "rt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- meshes: "OpenGL Raytracer/OBJ.h":7-79, OBJ.cpp:9-206 -------------------------------- */

/* struct VertexData of OBJ.h:7-24: point, texCoord, normal (32 bytes, no padding). */
typedef struct rth_vertex_data {
    float point[3];
    float texCoord[2];
    float normal[3];
} rth_vertex_data;

/* struct MeshMaterialData of OBJ.h:26-47; strings are "none" until the MTL sets them. */
typedef struct rth_material {
    char     name[128];
    uint32_t illum;
    float    Kd[3], Ka[3], Tf[3];
    char     map_Kd[256];
    float    Ni;
} rth_material;

typedef struct rth_mesh rth_mesh; /* opaque: the groups of one OBJ file */

/* Mesh::LoadFromObjFile(dir, filename), OBJ.cpp:9-148: opens dir + filename, understands
 * `# v vt vn f s mtllib usemtl g` and skips every other statement, de-indexes faces into flat
 * vertex lists, one list per group.  Group rules follow the reference: `g` opens a group;
 * a `g` that directly follows another `g` (nothing but skipped statements between) renames
 * instead; `usemtl` sets the material of the current group or opens group "default" when
 * there is none.  Where the reference has undefined behaviour this loader is defined:
 *   - a face before any group opens group "default" (the reference indexes an empty vector);
 *   - a trailing newline, blank lines and CRLF are harmless (the reference's eof() loop
 *     re-runs the last statement, OBJ.cpp:29-31);
 *   - corners may be v, v/vt, v//vn or v/vt/vn, indices may be negative (relative), faces
 *     with more than three corners are split as a fan; missing vt / vn read as zeros;
 *   - an index outside the arrays read so far is an error, not a wild read.
 * Returns 0 and a mesh, or -1 (*out = NULL; rth_last_error() has the text). */
int  rth_mesh_load(rth_mesh **out, const char *dir, const char *filename);
void rth_mesh_free(rth_mesh *m);
int  rth_mesh_group_count(const rth_mesh *m);
const char *rth_mesh_group_name(const rth_mesh *m, int group);
/* Mesh::GetVertexCount / GetVertexDataSize / GetVertexData / GetGroupMaterialData, OBJ.h:75-78.
 * A group index out of range gives 0 / NULL / -1 (the reference indexes out of bounds). */
int  rth_mesh_vertex_count(const rth_mesh *m, int group);
int  rth_mesh_vertex_data_size(const rth_mesh *m, int group);
const rth_vertex_data *rth_mesh_vertex_data(const rth_mesh *m, int group);
int  rth_mesh_material(const rth_mesh *m, int group, rth_material *out);
/* Mesh::LoadMaterialData(filepath, data), OBJ.cpp:150-206: `newmtl illum Kd Ka Tf map_Kd Ni`.
 * Writes up to `max` materials, returns how many the file defines, or -1. */
int  rth_load_materials(const char *filepath, rth_material *out, int max);

/* UploadToSSBO's repack, main.cpp:148-163: VertexData -> the 8-float record rt_set_mesh takes
 * (point, 0, texCoord, 0, 0).  `out` holds `count` records. */
void rth_pack_vertices(const rth_vertex_data *in, int count, rt_vertex *out);

/* ---- images: SOIL_load_OGL_texture(..., SOIL_LOAD_AUTO, ..., SOIL_FLAG_INVERT_Y), main.cpp:391-398 */

/* Decodes a TGA (types 1, 2, 3, 9, 10, 11; 8/15/16/24/32 bits; either origin) or a PNG (all
 * colour types and bit depths, tRNS, non-interlaced and Adam7) into w x h RGBA8 texels;
 * images without alpha get 255.  invert_y != 0 returns the rows bottom-up, which is what
 * SOIL_FLAG_INVERT_Y hands to GL and what rt_set_texture_rgba8 expects.
 * *rgba is malloc'd; release it with rth_image_free.  Returns 0 or -1. */
int  rth_image_load(const char *path, int invert_y, int *w, int *h, uint8_t **rgba);
int  rth_image_load_memory(const uint8_t *bytes, int64_t len, int invert_y, int *w, int *h, uint8_t **rgba);
void rth_image_free(uint8_t *rgba);
/* Frame files for the presentation tail: `rgba` is h rows of w RGBA8 pixels, row 0 = bottom
 * (the order rt_read_pixels_rgba8 returns).  TGA is written as 32-bit type 2 (or type 10 when
 * rle != 0) with a bottom-left origin; PNG as 8-bit RGBA. */
int  rth_write_tga(const char *path, const uint8_t *rgba, int w, int h, int rle);
int  rth_write_png(const char *path, const uint8_t *rgba, int w, int h);

/* ---- camera: camera.h:4-44, camera.cpp:6-18, 50-65, 84-94 --------------------------------- */

/* The basis Camera::Update derives from the two angles (radians), with the reference's 3.14f:
 *   direction = (cos v sin h, sin v, cos v cos h), right = (sin(h - 3.14f/2), 0, cos(h - 3.14f/2)),
 *   up = cross(right, direction); all in float. */
void rth_camera_basis(float horizontal_angle, float vertical_angle,
                      float direction[3], float right[3], float up[3]);
/* glm::perspective(fovy, aspect, near, far) and glm::lookAt(eye, eye + direction, up) as the
 * reference calls them (camera.cpp:87-94; GLM 0.9.7 right-handed, fovy in radians).
 * Column-major 4x4, m[col * 4 + row]. */
void rth_perspective(float fovy, float aspect, float z_near, float z_far, float m[16]);
void rth_look_at(const float eye[3], const float center[3], const float up[3], float m[16]);

/* ---- lights: main.cpp:266-285, 378-386 ---------------------------------------------------- */

/* The light rig at start-up: RT_MAX_LIGHTS lights on a circle of radius 2 in the plane z = -2
 * (SetLightPosition(lightData, 0)), colours from three rand()/RAND_MAX draws each with the
 * Microsoft C runtime's rand() from its default seed (the reference is a Win32 program and
 * never calls srand), paddings -1. */
void rth_default_lights(rt_light lights[RT_MAX_LIGHTS]);
/* SetLightPosition(lightData, angleOffset), main.cpp:276-285, for the first n lights. */
void rth_set_light_positions(rt_light *lights, int n, float angle_offset);
/* The per-frame offset of main.cpp:449: lerp(0, 2 pi / RT_MAX_LIGHTS, sin(time_seconds)). */
float rth_light_angle_offset(double time_seconds);

/* ---- benchmark CSV: WriteBenchmarkResultsToCSVFile, main.cpp:209-219 ---------------------- */

/* Appends one line `tgs,width,height,passes,lights,triangles,raygen,intersect,color` with the
 * three times as integers (the reference's columns, main.cpp:562-564).  Returns 0 or -1. */
int rth_csv_append(const char *filename, int thread_group_size, int width, int height, int passes,
                   int num_lights, int num_triangles, int raygen_ms, int intersect_ms, int color_ms);

/* ---- presentation tail: main.cpp:531-555, simple_vert.glsl, simple_frag.glsl -------------- */

/* Draws the light markers the forward pass draws: for each of the n lights, a square of
 * point_size pixels (30 in the reference) of opaque yellow centred on the projection of
 * light.pos through projection * view, clipped to the frame, skipped when the point is outside
 * the clip volume.  `rgba` is the bottom-up frame of rt_read_pixels_rgba8.  The reference's
 * depth buffer holds 1.0 everywhere when the markers are drawn (its depth image binding never
 * takes effect), so every marker inside the clip volume is visible; the same here. */
void rth_draw_light_markers(uint8_t *rgba, int w, int h, const float view[16], const float projection[16],
                            const rt_light *lights, int n, float point_size);

const char *rth_last_error(void); /* text of the last failure on the calling thread */

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H_ */
