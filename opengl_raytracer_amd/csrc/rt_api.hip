// rt_api.hip -- host side of the C ABI declared in include/rt.h.
//
// One rt_ctx per GPU.  It owns the ray planes, the vertex buffer, the filter
// records, the compaction lists, the framebuffer and the depth plane, and enqueues
// every stage on one HIP stream; stage order on that stream is what
// glMemoryBarrier provided in "OpenGL Raytracer/main.cpp":517-519.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt.h"
#include "rt_kernels.h"

using namespace rtk;

static thread_local std::string g_create_error;

struct EventPair {
    hipEvent_t a, b;
};

struct rt_ctx {
    int device = 0;
    int width = 0, height = 0;
    size_t cap = 0;    // rays allocated (= width*height)
    size_t stride = 0; // plane stride in elements
    float *planes = nullptr;
    RaySoA rays{};

    int local_rows = 0;
    int n_rays = 0;
    std::vector<int32_t> row_map_h;
    int *row_map_d = nullptr;

    rt_vertex *verts = nullptr;
    int num_vertices = 0, num_tris = 0;
    float center[3] = {0, 0, 0};
    float radius = 0;
    bool have_mesh = false;

    rtf::TriRec *rec_general = nullptr;
    rtf::TriRec *rec_camera = nullptr;
    rtf::TriRec *rec_lights = nullptr; // RT_MAX_LIGHTS x rec_stride
    size_t rec_stride = 0;
    bool camera_recs_dirty = true;
    bool light_recs_dirty = true;

    uint32_t *tex = nullptr;
    int tex_w = 0, tex_h = 0;

    rt_light lights[RT_MAX_LIGHTS] = {};
    int num_lights = 0;

    Camera cam{};
    bool have_camera = false;

    // rays were written by raygen and not touched since: they share camera_pos as origin
    bool rays_share_camera_origin = false;
    bool rays_valid = false;

    int *list = nullptr;
    int *count_d = nullptr; // [0] live/intersect list size, [1] shade list size
    int *count_h = nullptr; // pinned

    uint32_t *fb_own = nullptr;
    uint32_t *fb = nullptr;
    float *depth = nullptr;

    unsigned long long *counters_d = nullptr;

    hipStream_t stream_own = nullptr;
    hipStream_t stream = nullptr;

    int trace_mode = RT_TRACE_FILTERED;

    float acc_ms[3] = {0, 0, 0};
    std::vector<EventPair> ev_pool;
    std::vector<std::pair<int, int>> ev_used; // (stage, pool index)
    size_t ev_next = 0;
    // kernel log: a ring of event pairs around the traversal launches
    static constexpr int KLOG = 256;
    EventPair klog_ev[KLOG] = {};
    rt_kernel_record klog_rec[KLOG] = {};
    size_t klog_count = 0; // launches logged since the last read

    std::string error;
};

static int fail(rt_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->error = buf;
    else g_create_error = buf;
    return code;
}

#define HIPCHK(c, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail((c), RT_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                     \
    } while (0)

static int use_device(rt_ctx *c)
{
    HIPCHK(c, hipSetDevice(c->device));
    return RT_OK;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

extern "C" const char *rt_version(void) { return "rt-hip gfx950 0.1"; }

extern "C" const char *rt_last_error(rt_ctx *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

static void setup_planes(rt_ctx *c)
{
    float *b = c->planes;
    size_t s = c->stride;
    c->rays.ox = b + 0 * s; c->rays.oy = b + 1 * s; c->rays.oz = b + 2 * s;
    c->rays.dx = b + 3 * s; c->rays.dy = b + 4 * s; c->rays.dz = b + 5 * s;
    c->rays.cx = b + 6 * s; c->rays.cy = b + 7 * s; c->rays.cz = b + 8 * s;
    c->rays.t = b + 9 * s;
    c->rays.id = reinterpret_cast<int *>(b + 10 * s);
    c->rays.nx = b + 11 * s; c->rays.ny = b + 12 * s; c->rays.nz = b + 13 * s;
}

static int upload_row_map(rt_ctx *c)
{
    HIPCHK(c, hipMemcpyAsync(c->row_map_d, c->row_map_h.data(), c->row_map_h.size() * sizeof(int32_t),
                             hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

extern "C" int rt_create(rt_ctx **out, int hip_device, int width, int height)
{
    if (!out) return fail(nullptr, RT_ERR_ARG, "rt_create: out is null");
    *out = nullptr;
    if (width <= 0 || height <= 0 || (long long)width * height > (1ll << 30))
        return fail(nullptr, RT_ERR_ARG, "rt_create: bad frame size %d x %d", width, height);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, RT_ERR_HIP, "rt_create: no HIP device (%s)", hipGetErrorString(e));
    if (hip_device < 0 || hip_device >= ndev)
        return fail(nullptr, RT_ERR_ARG, "rt_create: device %d out of range (%d devices)", hip_device, ndev);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, hip_device);
    if (e != hipSuccess) return fail(nullptr, RT_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, RT_ERR_HIP, "rt_create: device %d is %s; this library is built for gfx950 only",
                    hip_device, prop.gcnArchName);

    rt_ctx *c = new rt_ctx();
    c->device = hip_device;
    c->width = width;
    c->height = height;
    c->cap = (size_t)width * height;
    c->stride = (c->cap + 255) & ~(size_t)255;
#define CRCHK(call)                                                                            \
    do {                                                                                       \
        hipError_t e2_ = (call);                                                               \
        if (e2_ != hipSuccess) {                                                               \
            fail(nullptr, RT_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e2_));         \
            rt_destroy(c);                                                                     \
            return RT_ERR_HIP;                                                                 \
        }                                                                                      \
    } while (0)
    CRCHK(hipSetDevice(hip_device));
    CRCHK(hipStreamCreateWithFlags(&c->stream_own, hipStreamNonBlocking));
    c->stream = c->stream_own;
    CRCHK(hipMalloc(&c->planes, 14 * c->stride * sizeof(float)));
    CRCHK(hipMemsetAsync(c->planes, 0, 14 * c->stride * sizeof(float), c->stream));
    setup_planes(c);
    CRCHK(hipMalloc(&c->row_map_d, (size_t)height * sizeof(int)));
    CRCHK(hipMalloc(&c->list, c->cap * sizeof(int)));
    CRCHK(hipMalloc(&c->count_d, 4 * sizeof(int)));
    CRCHK(hipHostMalloc(&c->count_h, 4 * sizeof(int)));
    CRCHK(hipMalloc(&c->fb_own, c->cap * sizeof(uint32_t)));
    CRCHK(hipMemsetAsync(c->fb_own, 0, c->cap * sizeof(uint32_t), c->stream));
    c->fb = c->fb_own;
    CRCHK(hipMalloc(&c->depth, c->cap * sizeof(float)));
    CRCHK(hipMemsetAsync(c->depth, 0, c->cap * sizeof(float), c->stream));
    CRCHK(hipMalloc(&c->counters_d, 8 * sizeof(unsigned long long)));
    CRCHK(hipMemsetAsync(c->counters_d, 0, 8 * sizeof(unsigned long long), c->stream));
    for (int i = 0; i < rt_ctx::KLOG; i++) {
        CRCHK(hipEventCreate(&c->klog_ev[i].a));
        CRCHK(hipEventCreate(&c->klog_ev[i].b));
    }
    c->local_rows = height;
    c->n_rays = width * height;
    c->row_map_h.resize(height);
    for (int y = 0; y < height; y++) c->row_map_h[y] = y;
    CRCHK(hipMemcpyAsync(c->row_map_d, c->row_map_h.data(), (size_t)height * sizeof(int), hipMemcpyHostToDevice,
                         c->stream));
    CRCHK(hipStreamSynchronize(c->stream));
#undef CRCHK
    *out = c;
    return RT_OK;
}

extern "C" void rt_destroy(rt_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto &p : c->ev_pool) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (int i = 0; i < rt_ctx::KLOG; i++) {
        if (c->klog_ev[i].a) (void)hipEventDestroy(c->klog_ev[i].a);
        if (c->klog_ev[i].b) (void)hipEventDestroy(c->klog_ev[i].b);
    }
    (void)hipFree(c->planes);
    (void)hipFree(c->row_map_d);
    (void)hipFree(c->verts);
    (void)hipFree(c->rec_general);
    (void)hipFree(c->rec_camera);
    (void)hipFree(c->rec_lights);
    (void)hipFree(c->tex);
    (void)hipFree(c->list);
    (void)hipFree(c->count_d);
    if (c->count_h) (void)hipHostFree(c->count_h);
    (void)hipFree(c->fb_own);
    (void)hipFree(c->depth);
    (void)hipFree(c->counters_d);
    if (c->stream_own) (void)hipStreamDestroy(c->stream_own);
    delete c;
}

extern "C" int rt_set_mesh(rt_ctx *c, const rt_vertex *vertices, int num_vertices)
{
    if (!c) return RT_ERR_ARG;
    if (num_vertices < 0 || num_vertices % 3 != 0 || (num_vertices > 0 && !vertices))
        return fail(c, RT_ERR_ARG, "rt_set_mesh: num_vertices=%d must be a non-negative multiple of 3", num_vertices);
    if (int r = use_device(c)) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->verts); c->verts = nullptr;
    (void)hipFree(c->rec_general); c->rec_general = nullptr;
    (void)hipFree(c->rec_camera); c->rec_camera = nullptr;
    (void)hipFree(c->rec_lights); c->rec_lights = nullptr;
    c->num_vertices = num_vertices;
    c->num_tris = num_vertices / 3;
    // record arrays are padded to whole tiles so that a tile copy never leaves the allocation
    c->rec_stride = ((size_t)c->num_tris + TILE - 1) / TILE * TILE + TILE;
    HIPCHK(c, hipMalloc(&c->verts, ((size_t)num_vertices + 3) * sizeof(rt_vertex)));
    HIPCHK(c, hipMemsetAsync(c->verts, 0, ((size_t)num_vertices + 3) * sizeof(rt_vertex), c->stream));
    if (num_vertices)
        HIPCHK(c, hipMemcpyAsync(c->verts, vertices, (size_t)num_vertices * sizeof(rt_vertex), hipMemcpyHostToDevice,
                                 c->stream));
    HIPCHK(c, hipMalloc(&c->rec_general, c->rec_stride * sizeof(rtf::TriRec)));
    HIPCHK(c, hipMalloc(&c->rec_camera, c->rec_stride * sizeof(rtf::TriRec)));
    HIPCHK(c, hipMalloc(&c->rec_lights, RT_MAX_LIGHTS * c->rec_stride * sizeof(rtf::TriRec)));
    if (c->num_tris)
        hipLaunchKernelGGL(prep_general_kernel, dim3(cdiv(c->num_tris, BLOCK)), dim3(BLOCK), 0, c->stream, c->verts,
                           c->num_tris, c->rec_general);
    HIPCHK(c, hipGetLastError());
    // bounding sphere of the vertices (centre of the bounding box, farthest vertex)
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    bool first = true;
    for (int i = 0; i < num_vertices; i++) {
        bool fin = std::isfinite(vertices[i].pos[0]) && std::isfinite(vertices[i].pos[1]) &&
                   std::isfinite(vertices[i].pos[2]);
        if (!fin) continue; // a triangle with such a vertex can never be hit
        for (int a = 0; a < 3; a++) {
            double v = vertices[i].pos[a];
            if (first || v < lo[a]) lo[a] = v;
            if (first || v > hi[a]) hi[a] = v;
        }
        first = false;
    }
    double ctr[3] = {(lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, (lo[2] + hi[2]) / 2};
    double r2 = 0;
    for (int i = 0; i < num_vertices; i++) {
        double dx = vertices[i].pos[0] - ctr[0], dy = vertices[i].pos[1] - ctr[1], dz = vertices[i].pos[2] - ctr[2];
        double d2 = dx * dx + dy * dy + dz * dz;
        if (std::isfinite(d2) && d2 > r2) r2 = d2;
    }
    for (int a = 0; a < 3; a++) c->center[a] = (float)ctr[a];
    c->radius = (float)(std::sqrt(r2) * 1.0001 + 1e-30);
    c->have_mesh = true;
    c->camera_recs_dirty = true;
    c->light_recs_dirty = true;
    HIPCHK(c, hipStreamSynchronize(c->stream)); // the caller may free `vertices` on return
    return RT_OK;
}

extern "C" int rt_set_texture_rgba8(rt_ctx *c, const uint8_t *texels, int w, int h)
{
    if (!c) return RT_ERR_ARG;
    if (w < 0 || h < 0 || ((w > 0 && h > 0) && !texels) || (long long)w * h > (1ll << 28))
        return fail(c, RT_ERR_ARG, "rt_set_texture_rgba8: bad texture %d x %d", w, h);
    if (int r = use_device(c)) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->tex);
    c->tex = nullptr;
    c->tex_w = w;
    c->tex_h = h;
    if (w > 0 && h > 0) {
        HIPCHK(c, hipMalloc(&c->tex, (size_t)w * h * 4));
        HIPCHK(c, hipMemcpyAsync(c->tex, texels, (size_t)w * h * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return RT_OK;
}

extern "C" int rt_set_lights(rt_ctx *c, const rt_light *lights, int num_lights)
{
    if (!c) return RT_ERR_ARG;
    if (num_lights < 0 || num_lights > RT_MAX_LIGHTS || (num_lights > 0 && !lights))
        return fail(c, RT_ERR_ARG, "rt_set_lights: num_lights=%d out of range 0..%d", num_lights, RT_MAX_LIGHTS);
    for (int i = 0; i < num_lights; i++) {
        if (memcmp(c->lights[i].pos, lights[i].pos, sizeof(float) * 3) != 0 || i >= c->num_lights)
            c->light_recs_dirty = true;
        c->lights[i] = lights[i];
    }
    c->num_lights = num_lights;
    return RT_OK;
}

extern "C" int rt_set_camera(rt_ctx *c, const float pos[3], const float dir[3], const float up[3],
                             const float right[3])
{
    if (!c) return RT_ERR_ARG;
    if (!pos || !dir || !up || !right) return fail(c, RT_ERR_ARG, "rt_set_camera: null vector");
    if (!c->have_camera || memcmp(c->cam.pos, pos, sizeof(float) * 3) != 0) c->camera_recs_dirty = true;
    memcpy(c->cam.pos, pos, sizeof(float) * 3);
    memcpy(c->cam.dir, dir, sizeof(float) * 3);
    memcpy(c->cam.up, up, sizeof(float) * 3);
    memcpy(c->cam.right, right, sizeof(float) * 3);
    c->have_camera = true;
    return RT_OK;
}

extern "C" int rt_set_row_tiles(rt_ctx *c, int tile_rows, int first_tile, int tile_stride)
{
    if (!c) return RT_ERR_ARG;
    if (tile_rows <= 0 || first_tile < 0 || tile_stride <= 0)
        return fail(c, RT_ERR_ARG, "rt_set_row_tiles: bad arguments (%d, %d, %d)", tile_rows, first_tile,
                    tile_stride);
    if (int r = use_device(c)) return r;
    c->row_map_h.clear();
    const int ntiles = (c->height + tile_rows - 1) / tile_rows;
    for (int t = first_tile; t < ntiles; t += tile_stride)
        for (int y = t * tile_rows; y < (t + 1) * tile_rows && y < c->height; y++) c->row_map_h.push_back(y);
    c->local_rows = (int)c->row_map_h.size();
    c->n_rays = c->local_rows * c->width;
    c->rays_valid = false;
    c->rays_share_camera_origin = false;
    if (c->local_rows) return upload_row_map(c);
    return RT_OK;
}

extern "C" int rt_local_rows(rt_ctx *c) { return c ? c->local_rows : RT_ERR_ARG; }
extern "C" int rt_num_rays(rt_ctx *c) { return c ? c->n_rays : RT_ERR_ARG; }
extern "C" int rt_get_row_map(rt_ctx *c, int32_t *rows)
{
    if (!c || !rows) return RT_ERR_ARG;
    memcpy(rows, c->row_map_h.data(), c->row_map_h.size() * sizeof(int32_t));
    return RT_OK;
}

extern "C" int rt_set_trace_mode(rt_ctx *c, int mode)
{
    if (!c) return RT_ERR_ARG;
    if (mode != RT_TRACE_FILTERED && mode != RT_TRACE_EXACT)
        return fail(c, RT_ERR_ARG, "rt_set_trace_mode: unknown mode %d", mode);
    c->trace_mode = mode;
    return RT_OK;
}

// ---- stage timing -----------------------------------------------------------
static int fold_events(rt_ctx *c)
{
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto &u : c->ev_used) {
        float t = 0.0f;
        HIPCHK(c, hipEventElapsedTime(&t, c->ev_pool[u.second].a, c->ev_pool[u.second].b));
        c->acc_ms[u.first] += t;
    }
    c->ev_used.clear();
    c->ev_next = 0;
    return RT_OK;
}

static int stage_begin(rt_ctx *c, int stage, size_t *slot)
{
    if (c->ev_next >= 1024)
        if (int r = fold_events(c)) return r;
    if (c->ev_next == c->ev_pool.size()) {
        EventPair p;
        HIPCHK(c, hipEventCreate(&p.a));
        HIPCHK(c, hipEventCreate(&p.b));
        c->ev_pool.push_back(p);
    }
    *slot = c->ev_next++;
    c->ev_used.push_back({stage, (int)*slot});
    HIPCHK(c, hipEventRecord(c->ev_pool[*slot].a, c->stream));
    return RT_OK;
}
static int stage_end(rt_ctx *c, size_t slot)
{
    HIPCHK(c, hipEventRecord(c->ev_pool[slot].b, c->stream));
    return RT_OK;
}

extern "C" int rt_stage_ms(rt_ctx *c, float ms[3])
{
    if (!c || !ms) return RT_ERR_ARG;
    if (int r = use_device(c)) return r;
    if (int r = fold_events(c)) return r;
    for (int i = 0; i < 3; i++) {
        ms[i] = c->acc_ms[i];
        c->acc_ms[i] = 0.0f;
    }
    return RT_OK;
}

// ---- stages -------------------------------------------------------------------
static int check_ready(rt_ctx *c, bool need_mesh, bool need_rays, const char *who)
{
    if (!c->have_camera) return fail(c, RT_ERR_STATE, "%s: rt_set_camera has not been called", who);
    if (need_mesh && !c->have_mesh) return fail(c, RT_ERR_STATE, "%s: rt_set_mesh has not been called", who);
    if (need_rays && !c->rays_valid)
        return fail(c, RT_ERR_STATE, "%s: the ray buffer is empty (call rt_raygen or rt_write_rays)", who);
    return use_device(c);
}

extern "C" int rt_raygen(rt_ctx *c)
{
    if (!c) return RT_ERR_ARG;
    if (int r = check_ready(c, false, false, "rt_raygen")) return r;
    size_t slot;
    if (int r = stage_begin(c, 0, &slot)) return r;
    if (c->n_rays > 0) {
        hipLaunchKernelGGL(raygen_kernel, dim3(cdiv(c->n_rays, BLOCK * 4)), dim3(BLOCK), 0, c->stream, c->rays,
                           c->n_rays, c->width, c->height, c->row_map_d, c->cam);
        HIPCHK(c, hipGetLastError());
    }
    if (int r = stage_end(c, slot)) return r;
    c->rays_valid = true;
    c->rays_share_camera_origin = true;
    return RT_OK;
}

// picks the rays-per-lane variant for a launch over n rays
static int pick_rays_per_lane(int n)
{
    if (n >= 4 * BLOCK * 4 * 256) return 4;
    if (n >= 2 * BLOCK * 2 * 256) return 2;
    return 1;
}

static int klog_begin(rt_ctx *c, int stage, int form, int k, bool exact, unsigned long long rays,
                      unsigned long long pairs, size_t *slot)
{
    *slot = c->klog_count % rt_ctx::KLOG;
    rt_kernel_record &r = c->klog_rec[*slot];
    r.stage = stage; r.form = form; r.rays_per_lane = k; r.exact = exact ? 1 : 0;
    r.rays = rays; r.pairs = pairs; r.ms = 0.0f; r.reserved = 0.0f;
    c->klog_count++;
    HIPCHK(c, hipEventRecord(c->klog_ev[*slot].a, c->stream));
    return RT_OK;
}
static int klog_end(rt_ctx *c, size_t slot)
{
    HIPCHK(c, hipEventRecord(c->klog_ev[slot].b, c->stream));
    return RT_OK;
}

template <int FORM, bool EXACT>
static void launch_closest(rt_ctx *c, const TraceParams &p, int n_upper)
{
    // rays per lane: more rays amortise each LDS record read over more arithmetic;
    // fewer rays give small launches enough workgroups to cover 256 CUs.
    const int k = pick_rays_per_lane(n_upper);
    if (k == 4)
        hipLaunchKernelGGL((closest_hit_kernel<4, FORM, EXACT>), dim3(cdiv(n_upper, BLOCK * 4)), dim3(BLOCK), 0,
                           c->stream, p);
    else if (k == 2)
        hipLaunchKernelGGL((closest_hit_kernel<2, FORM, EXACT>), dim3(cdiv(n_upper, BLOCK * 2)), dim3(BLOCK), 0,
                           c->stream, p);
    else
        hipLaunchKernelGGL((closest_hit_kernel<1, FORM, EXACT>), dim3(cdiv(n_upper, BLOCK)), dim3(BLOCK), 0,
                           c->stream, p);
}

static int refresh_camera_recs(rt_ctx *c)
{
    if (c->camera_recs_dirty && c->num_tris) {
        hipLaunchKernelGGL(prep_origin_kernel, dim3(cdiv(c->num_tris, BLOCK)), dim3(BLOCK), 0, c->stream, c->verts,
                           c->num_tris, c->cam.pos[0], c->cam.pos[1], c->cam.pos[2], c->rec_camera);
        HIPCHK(c, hipGetLastError());
    }
    c->camera_recs_dirty = false;
    return RT_OK;
}

static int refresh_light_recs(rt_ctx *c)
{
    if (c->light_recs_dirty && c->num_tris) {
        for (int a = 0; a < c->num_lights; a++) {
            hipLaunchKernelGGL(prep_origin_kernel, dim3(cdiv(c->num_tris, BLOCK)), dim3(BLOCK), 0, c->stream,
                               c->verts, c->num_tris, c->lights[a].pos[0], c->lights[a].pos[1], c->lights[a].pos[2],
                               c->rec_lights + (size_t)a * c->rec_stride);
            HIPCHK(c, hipGetLastError());
        }
    }
    c->light_recs_dirty = false;
    return RT_OK;
}

// reads count_d[which] back to the host (one 4-byte copy and a stream sync)
static int read_count(rt_ctx *c, int which, int *out)
{
    HIPCHK(c, hipMemcpyAsync(c->count_h + which, c->count_d + which, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *out = c->count_h[which];
    return RT_OK;
}

extern "C" int rt_intersect(rt_ctx *c)
{
    if (!c) return RT_ERR_ARG;
    if (int r = check_ready(c, true, true, "rt_intersect")) return r;
    size_t slot;
    if (int r = stage_begin(c, 1, &slot)) return r;
    if (c->n_rays > 0) {
        const bool shared = c->rays_share_camera_origin;
        TraceParams p{};
        p.rays = c->rays;
        p.verts = c->verts;
        p.num_tris = c->num_tris;
        memcpy(p.center, c->center, sizeof p.center);
        p.radius = c->radius;
        memcpy(p.origin, c->cam.pos, sizeof p.origin);
        memcpy(p.cam_pos, c->cam.pos, sizeof p.cam_pos);
        p.depth = c->depth;
        p.counters = c->counters_d;
        int n_upper = c->n_rays;
        if (shared) {
            // fresh rays: all live, all through camera_pos
            if (int r = refresh_camera_recs(c)) return r;
            p.list = nullptr;
            p.count = nullptr;
            p.n = c->n_rays;
            p.recs = c->rec_camera;
        } else {
            HIPCHK(c, hipMemsetAsync(c->count_d, 0, sizeof(int), c->stream));
            hipLaunchKernelGGL((compact_kernel<0>), dim3(cdiv(c->n_rays, BLOCK * COMPACT_ITERS)), dim3(BLOCK), 0,
                               c->stream, c->rays, c->n_rays, c->list, c->count_d, (uint32_t *)nullptr);
            HIPCHK(c, hipGetLastError());
            if (int r = read_count(c, 0, &n_upper)) return r;
            p.list = c->list;
            p.count = c->count_d;
            p.n = 0;
            p.recs = c->rec_general;
        }
        if (n_upper > 0) {
            const bool exact = c->trace_mode == RT_TRACE_EXACT;
            size_t ks;
            if (int r = klog_begin(c, 1, shared ? 1 : 0, pick_rays_per_lane(n_upper), exact, n_upper,
                                   (unsigned long long)n_upper * (unsigned long long)c->num_tris, &ks))
                return r;
            if (shared) {
                if (exact) launch_closest<1, true>(c, p, n_upper);
                else launch_closest<1, false>(c, p, n_upper);
            } else {
                if (exact) launch_closest<0, true>(c, p, n_upper);
                else launch_closest<0, false>(c, p, n_upper);
            }
            HIPCHK(c, hipGetLastError());
            if (int r = klog_end(c, ks)) return r;
        }
    }
    if (int r = stage_end(c, slot)) return r;
    c->rays_share_camera_origin = false;
    return RT_OK;
}

template <bool EXACT>
static void launch_shade(rt_ctx *c, const ShadeParams &p, int n_upper)
{
    const int k = pick_rays_per_lane(n_upper);
    if (k == 4)
        hipLaunchKernelGGL((shade_kernel<4, EXACT>), dim3(cdiv(n_upper, BLOCK * 4)), dim3(BLOCK), 0, c->stream, p);
    else if (k == 2)
        hipLaunchKernelGGL((shade_kernel<2, EXACT>), dim3(cdiv(n_upper, BLOCK * 2)), dim3(BLOCK), 0, c->stream, p);
    else
        hipLaunchKernelGGL((shade_kernel<1, EXACT>), dim3(cdiv(n_upper, BLOCK)), dim3(BLOCK), 0, c->stream, p);
}

extern "C" int rt_color(rt_ctx *c)
{
    if (!c) return RT_ERR_ARG;
    if (int r = check_ready(c, true, true, "rt_color")) return r;
    size_t slot;
    if (int r = stage_begin(c, 2, &slot)) return r;
    if (c->n_rays > 0) {
        if (c->num_lights == 0) {
            // no light: `shadowed` stays true for every pixel (raycol_cs.glsl:28-32, 82)
            hipLaunchKernelGGL(fill_grey_kernel, dim3(cdiv(c->n_rays, BLOCK)), dim3(BLOCK), 0, c->stream, c->fb,
                               c->n_rays);
            HIPCHK(c, hipGetLastError());
        } else {
            if (int r = refresh_light_recs(c)) return r;
            HIPCHK(c, hipMemsetAsync(c->count_d + 1, 0, sizeof(int), c->stream));
            hipLaunchKernelGGL((compact_kernel<1>), dim3(cdiv(c->n_rays, BLOCK * COMPACT_ITERS)), dim3(BLOCK), 0,
                               c->stream, c->rays, c->n_rays, c->list, c->count_d + 1, c->fb);
            HIPCHK(c, hipGetLastError());
            int n_upper = 0;
            if (int r = read_count(c, 1, &n_upper)) return r;
            if (n_upper > 0) {
                ShadeParams p{};
                p.rays = c->rays;
                p.list = c->list;
                p.count = c->count_d + 1;
                p.light_recs = c->rec_lights;
                p.rec_stride = c->rec_stride;
                p.verts = c->verts;
                p.num_vertices = c->num_vertices;
                p.num_tris = c->num_tris;
                memcpy(p.center, c->center, sizeof p.center);
                p.radius = c->radius;
                memcpy(p.cam_pos, c->cam.pos, sizeof p.cam_pos);
                p.num_lights = c->num_lights;
                memcpy(p.lights, c->lights, sizeof(rt_light) * RT_MAX_LIGHTS);
                p.tex = c->tex;
                p.tex_w = c->tex_w;
                p.tex_h = c->tex_h;
                p.fb = c->fb;
                p.counters = c->counters_d;
                const bool exact = c->trace_mode == RT_TRACE_EXACT;
                size_t ks;
                if (int r = klog_begin(c, 2, 1, pick_rays_per_lane(n_upper), exact, n_upper,
                                       (unsigned long long)n_upper * (unsigned long long)c->num_tris *
                                           (unsigned long long)c->num_lights,
                                       &ks))
                    return r;
                if (exact) launch_shade<true>(c, p, n_upper);
                else launch_shade<false>(c, p, n_upper);
                HIPCHK(c, hipGetLastError());
                if (int r = klog_end(c, ks)) return r;
            }
        }
    }
    if (int r = stage_end(c, slot)) return r;
    // a killed ray keeps its shifted origin: the rays no longer share an origin
    c->rays_share_camera_origin = false;
    return RT_OK;
}

extern "C" int rt_render(rt_ctx *c, int passes)
{
    if (!c) return RT_ERR_ARG;
    if (passes < 1) return fail(c, RT_ERR_ARG, "rt_render: passes=%d must be >= 1", passes);
    if (int r = rt_raygen(c)) return r;
    for (int pss = 0; pss < passes; pss++) {
        if (int r = rt_intersect(c)) return r;
        if (int r = rt_color(c)) return r;
    }
    return RT_OK;
}

extern "C" int rt_sync(rt_ctx *c)
{
    if (!c) return RT_ERR_ARG;
    if (int r = use_device(c)) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

// ---- transfers ----------------------------------------------------------------
extern "C" int rt_read_rays(rt_ctx *c, rt_ray *out)
{
    if (!c || !out) return RT_ERR_ARG;
    if (int r = use_device(c)) return r;
    if (c->n_rays == 0) return RT_OK;
    rt_ray *tmp = nullptr;
    HIPCHK(c, hipMalloc(&tmp, (size_t)c->n_rays * sizeof(rt_ray)));
    hipLaunchKernelGGL(soa_to_aos_kernel, dim3(cdiv(c->n_rays, BLOCK)), dim3(BLOCK), 0, c->stream, c->rays, c->n_rays,
                       tmp);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpyAsync(out, tmp, (size_t)c->n_rays * sizeof(rt_ray), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(c, RT_ERR_HIP, "rt_read_rays: %s", hipGetErrorString(e));
    return RT_OK;
}

extern "C" int rt_write_rays(rt_ctx *c, const rt_ray *in)
{
    if (!c || !in) return RT_ERR_ARG;
    if (int r = use_device(c)) return r;
    if (c->n_rays == 0) return RT_OK;
    rt_ray *tmp = nullptr;
    HIPCHK(c, hipMalloc(&tmp, (size_t)c->n_rays * sizeof(rt_ray)));
    hipError_t e = hipMemcpyAsync(tmp, in, (size_t)c->n_rays * sizeof(rt_ray), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(aos_to_soa_kernel, dim3(cdiv(c->n_rays, BLOCK)), dim3(BLOCK), 0, c->stream, c->rays,
                           c->n_rays, tmp);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(c, RT_ERR_HIP, "rt_write_rays: %s", hipGetErrorString(e));
    c->rays_valid = true;
    c->rays_share_camera_origin = false;
    return RT_OK;
}

extern "C" int rt_read_pixels_rgba8(rt_ctx *c, uint8_t *out)
{
    if (!c || !out) return RT_ERR_ARG;
    if (int r = use_device(c)) return r;
    if (c->n_rays == 0) return RT_OK;
    HIPCHK(c, hipMemcpyAsync(out, c->fb, (size_t)c->n_rays * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

extern "C" int rt_read_depth(rt_ctx *c, float *out)
{
    if (!c || !out) return RT_ERR_ARG;
    if (int r = use_device(c)) return r;
    if (c->n_rays == 0) return RT_OK;
    HIPCHK(c, hipMemcpyAsync(out, c->depth, (size_t)c->n_rays * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

extern "C" int rt_get_counters(rt_ctx *c, uint64_t counters[8])
{
    if (!c || !counters) return RT_ERR_ARG;
    if (int r = use_device(c)) return r;
    HIPCHK(c, hipMemcpyAsync(counters, c->counters_d, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemsetAsync(c->counters_d, 0, 8 * sizeof(uint64_t), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

extern "C" int rt_kernel_log(rt_ctx *c, rt_kernel_record *out, int max, int *n)
{
    if (!c || !n || (max > 0 && !out)) return RT_ERR_ARG;
    if (int r = use_device(c)) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t have = c->klog_count < (size_t)rt_ctx::KLOG ? c->klog_count : (size_t)rt_ctx::KLOG;
    const size_t first = c->klog_count - have;
    int w = 0;
    for (size_t i = first; i < c->klog_count && w < max; i++) {
        const size_t slot = i % rt_ctx::KLOG;
        rt_kernel_record r = c->klog_rec[slot];
        HIPCHK(c, hipEventElapsedTime(&r.ms, c->klog_ev[slot].a, c->klog_ev[slot].b));
        out[w++] = r;
    }
    *n = w;
    c->klog_count = 0;
    return RT_OK;
}

extern "C" int rt_bind_stream(rt_ctx *c, void *hip_stream)
{
    if (!c) return RT_ERR_ARG;
    if (int r = use_device(c)) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return RT_OK;
}

extern "C" int rt_unbind_stream(rt_ctx *c)
{
    if (!c) return RT_ERR_ARG;
    if (int r = use_device(c)) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = c->stream_own;
    return RT_OK;
}

extern "C" int rt_bind_framebuffer(rt_ctx *c, void *device_rgba8)
{
    if (!c) return RT_ERR_ARG;
    if (int r = use_device(c)) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->fb = device_rgba8 ? reinterpret_cast<uint32_t *>(device_rgba8) : c->fb_own;
    return RT_OK;
}
