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
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../../include/rt.h"
#include "rt_kernels.h"
#include "rt_cluster.h"
#include "rt_voxel.h"

using namespace rtk;

static thread_local std::string g_create_error;

// Staging memory of rt_read_rays_static: the 80-byte records are assembled here before they are copied
// to the host.  Every GPU the library runs on holds its own copy of the array, at its own address, until
// the process ends.  In host code the array's name is only a handle for hipGetSymbolAddress: the address
// that kernels and copies take is looked up once per device (rt_create, with that device current) and
// kept in g_static_stage_ptr.
constexpr size_t STATIC_STAGE_BYTES = (size_t)64 << 20;
constexpr size_t STATIC_STAGE_MAX_RAYS = STATIC_STAGE_BYTES / sizeof(rt_ray);
static __device__ __align__(256) unsigned char g_static_stage[STATIC_STAGE_BYTES];
constexpr int STATIC_STAGE_MAX_DEVICES = 64;
static void *g_static_stage_ptr[STATIC_STAGE_MAX_DEVICES] = {};
static std::mutex g_static_stage_mutex;

// device address of g_static_stage on the current device `dev` (looked up on first use)
static hipError_t static_stage_address(int dev, void **out)
{
    if (dev < 0 || dev >= STATIC_STAGE_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_static_stage_mutex);
    if (!g_static_stage_ptr[dev]) {
        void *ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_static_stage));
        if (e != hipSuccess) return e;
        if (!ptr) return hipErrorInvalidSymbol;
        g_static_stage_ptr[dev] = ptr;
    }
    *out = g_static_stage_ptr[dev];
    return hipSuccess;
}

struct EventPair {
    hipEvent_t a, b;
};

// The band sweep's lists of one mesh on one device.  They depend on the mesh, the resolution of the
// direction cells and the bucket count only, are read-only once built, and are the largest thing a context
// holds: the contexts of a process that hold the same mesh on the same device (several, when frames are
// kept in flight) share one copy.  The registry keeps weak references; the lists go with their last context.
struct BandLists {
    int device = 0;
    unsigned int *off = nullptr;
    int *entries = nullptr;
    void *frames = nullptr;
    int g = 1;
    unsigned long long total = 0;
    ~BandLists()
    {
        int cur = 0;
        (void)hipGetDevice(&cur);
        (void)hipSetDevice(device);
        (void)hipFree(off);
        (void)hipFree(entries);
        (void)hipFree(frames);
        (void)hipSetDevice(cur);
    }
};
struct BandKey {
    int device, num_vertices, grid_bits, buckets;
    unsigned long long limit, h1, h2; // (two 64-bit hashes of the vertex array)
    bool operator<(const BandKey &o) const
    {
        return std::tie(device, num_vertices, grid_bits, buckets, limit, h1, h2) <
               std::tie(o.device, o.num_vertices, o.grid_bits, o.buckets, o.limit, o.h1, o.h2);
    }
};
static std::mutex g_band_mutex;
static std::map<BandKey, std::weak_ptr<BandLists>> g_band_cache;

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
    int *perm_d = nullptr; // local ray indices in pixel-tile order

    rt_vertex *verts = nullptr;
    int num_vertices = 0, num_tris = 0;
    float center[3] = {0, 0, 0};
    float radius = 0;
    bool have_mesh = false;

    rtf::TriRec *rec_general = nullptr;
    rtf::MomRec *rec_moment = nullptr; // per-pair filter of arbitrary rays (moment form)
    rtf::TriRec *rec_camera = nullptr;
    rtf::TriRec *rec_lights = nullptr; // RT_MAX_LIGHTS x rec_stride
    rtf::ConeRec *cone_camera = nullptr;
    rtf::ConeRec *cone_lights = nullptr; // RT_MAX_LIGHTS x rec_stride
    // per origin: one cone per run of RUN slots, then one per SUPER_RUN slots (ccone_stride records)
    rtf::ConeRec *ccone_camera = nullptr;
    rtf::ConeRec *ccone_lights = nullptr; // RT_MAX_LIGHTS x ccone_stride
    size_t ccone_stride = 0, scone_offset = 0;
    int *slot_d = nullptr;                // triangle in each slot of the shared-origin traversal's order
    int num_slots = 0;
    size_t rec_stride = 0;
    // clustered traversal of arbitrary rays: plane bands per triangle, spheres / member records per cluster
    float4 *ngroup_d = nullptr;
    float4 *nsub_d = nullptr;
    float4 *nrec_d = nullptr;
    int num_ngroups = 0;
    // cells in space for the well-conditioned pairs of arbitrary rays (rt_voxel.h)
    rtv::Grid vox{};
    unsigned int *vox_off_d = nullptr;
    int *vox_entries_d = nullptr;
    int voxel_res = 0; // cells along the longest axis; 0: chosen from the number of triangles
    // walks that traverse_voxel_kernel hands over to traverse_voxel_long_kernel: (slot, parameter), their number
    int *walk_slot_d = nullptr;
    float *walk_t_d = nullptr;
    int *walk_count_d = nullptr;
    int walk_budget = 0; // 0: chosen on the device from the number of rays
    bool camera_recs_dirty = true;
    bool light_recs_dirty = true;
    // direction cells (rt_grid.h): per shared origin (0 = camera, 1 + a = light a) the list offsets
    // and the lists; grid_bits = 0 turns them off (RT_GRID_BITS)
    int grid_bits = 9;
    rtg::CellCone *cell_cones_d = nullptr;
    unsigned int *grid_count_d = nullptr;
    unsigned int *grid_off[1 + RT_MAX_LIGHTS] = {};
    int *grid_entries[1 + RT_MAX_LIGHTS] = {};
    size_t grid_cap[1 + RT_MAX_LIGHTS] = {};
    int grid_level[1 + RT_MAX_LIGHTS] = {};  // cells per axis = 2^level of that origin's lists (<= grid_bits)
    int grid_uses[1 + RT_MAX_LIGHTS] = {};   // launches those lists have served
    void *scan_temp = nullptr;
    size_t scan_temp_bytes = 0;
    rtg::SlotView *slot_views_d = nullptr; // scratch of the list builder, one per slot
    unsigned long long *build_count_d = nullptr; // [0] pairs the list builder counted, [1] its fault flag,
                                                 // [2], [3]: grid_overflow_kernel
    unsigned int *build_overflow_d = nullptr;    // a bit per slot: the builder had no room for one of its pairs
    unsigned int *build_item_keys_d = nullptr;   // the builder's items (grid_list_kernel): cells three levels up ...
    int *build_item_vals_d = nullptr;            // ... and their slots; build_count_d[4] counts them
    size_t build_items_cap = 0;
    unsigned long long *build_feedback_h = nullptr; // pinned, per origin: [0], [1] of its last build (never waited for)
    unsigned int *build_keys_d = nullptr, *build_keys_out_d = nullptr; // (cell, slot) pairs before / after the sort
    int *build_vals_d = nullptr;
    size_t build_pairs_cap = 0;
    void *build_temp_d = nullptr;
    size_t build_temp_bytes = 0;
    // direction cells of the band sweep: lists of member records, per mesh
    unsigned int *band_off = nullptr;
    int *band_entries = nullptr;
    bool band_cells_pending = false; // rt_set_mesh left them to the first clustered bounce pass
    bool vox_pending = false;        // ... and the cells in space too (built on the host, from vox_rho)
    std::vector<double> vox_rho;     // per triangle: the radius the cells list it with (< 0: it takes no part)
    void *bucket_frames_d = nullptr;  // rtg::BucketFrame of every finest direction cell
    int band_buckets = 0;             // origin buckets per cell and axis asked for (rt_set_option); 0: chosen here
    int band_g = 1;                   // ... and built
    double mesh_lo[3] = {0, 0, 0}, mesh_hi[3] = {0, 0, 0}; // bounding box of the mesh's finite vertices
    unsigned long long band_total = 0; // entries on the lists
    std::shared_ptr<BandLists> band_shared; // owns band_off, band_entries and bucket_frames_d once they are built
    unsigned long long mesh_hash[2] = {0, 0};

    uint32_t *tex = nullptr;
    int tex_w = 0, tex_h = 0;

    rt_light lights[RT_MAX_LIGHTS] = {};
    int num_lights = 0;

    Camera cam{};
    bool have_camera = false;

    // rays were written by raygen and not touched since: they share camera_pos as origin
    bool rays_share_camera_origin = false;
    // the last intersect traced such rays: their hit points follow the pixel order, and so do the
    // directions in which a light sees them -- the next shading pass does not sort its shadow rays
    bool hits_follow_pixels = false;
    bool rays_valid = false;
    // rt_render does not launch raygen when the primary traversal will form the rays itself (traverse_grid_kernel
    // over fresh rays): until that traversal has run the ray planes are unwritten, and anything else that wants
    // them writes them first (write_pending_rays) from the camera the frame was begun with
    bool rays_unwritten = false;
    Camera raygen_cam{};
    int *chunk_filed_d[2] = {nullptr, nullptr}; // entries each chunk of the compaction filed (rt_intersect's, rt_color's)
    bool chunk_filed_valid[2] = {false, false}; // ... in the last pass over the rays now in the buffer

    int *list = nullptr;    // compacted ray indices (live rays / rays with a finite origin)
    int *alist[2] = {nullptr, nullptr}; // shadow rays still unoccluded (slots), ping-pong
    // shadow rays sorted by the direction in which they leave the light
    unsigned int *sort_keys = nullptr;
    int *sort_slots = nullptr;
    // ... by counting: rays per key, their exclusive scan, rocprim's temporary storage for the scan
    unsigned int *bin_hist_d = nullptr, *bin_off_d = nullptr;
    size_t bin_hist_bytes = 0;
    bool bin_hist_dirty = true;      // not known to be all zero
    bool count_clean[2] = {false, false}; // count_d[0] / [1] (rt_intersect's / rt_color's list length) is zero
    void *bin_temp_d = nullptr;
    size_t bin_temp_bytes = 0;
    int *count_d = nullptr; // [0] intersect list, [1] shade list, [2],[3] alist sizes
    int *count_h = nullptr; // pinned
    int *feedback_h = nullptr; // pinned: [0] rays with a hit in the last first pass (written by the device, never waited for)
    float *work = nullptr;  // per-slot planes behind job / scratch
    Job job{};
    Scratch scratch{};

    void *static_stage = nullptr; // this device's copy of g_static_stage (rt_read_rays_static)

    uint32_t *fb_own = nullptr;
    uint32_t *fb = nullptr;
    float *depth = nullptr;

    unsigned long long *counters_d = nullptr;

    hipStream_t stream_own = nullptr;
    // the two sweeps of a clustered bounce pass run side by side: the sphere sweep on this second stream
    hipStream_t stream_side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool forked = false;   // rt_intersect has already forked the side stream for this pass (before the sort)
    size_t forked_ks = 0;  // ... and opened this record of the kernel log
    hipStream_t stream = nullptr;

    int trace_mode = RT_TRACE_CLUSTERED;
    // tuning, per context (rt_set_option)
    int sort_min_rays = 131072;      // launches with at least this many rays sort them
    int cluster_min_tris = 1024;     // smaller meshes: bounce rays take the per-pair filter
    bool fail_next_stage = false;    // RT_OPT_FAIL_NEXT_STAGE
    int moving_origin_levels = 1;    // RT_OPT_MOVING_ORIGIN_LEVELS
    bool sweeps_in_turn = false;     // the two sweeps of a clustered bounce pass one after the other
    int exact_rays_per_lane = 0;     // all-pairs kernel: 2, 4 or 8 rays per lane, 0 = from the launch's size
    unsigned long long band_list_max = 0xffffff00ull; // longer band lists are not built (32-bit positions)
    // 64-bit totals of the list builders
    unsigned long long *sum_d = nullptr;
    unsigned long long *sum_h = nullptr; // pinned

    float acc_ms[3] = {0, 0, 0};
    // stage timing: a pool of events; a stage call is timed from one event to another, and inside
    // rt_render the event that ends a stage also begins the next (an event between two kernels costs
    // the stream about 5 us of idle time, a frame of six passes has 13 stage calls)
    struct StageSpan { int stage, a, b; };
    std::vector<hipEvent_t> ev_pool;
    std::vector<StageSpan> ev_used;
    size_t ev_next = 0;
    bool chain_stage_events = false; // set by rt_render: the next stage call begins where the last one ended
    bool kernel_timing = false;      // RT_OPT_KERNEL_TIMING: event pairs around the traversal launches
    // kernel log: a ring of event pairs around the traversal launches
    static constexpr int KLOG = 1024;
    EventPair klog_ev[KLOG] = {};
    rt_kernel_record klog_rec[KLOG] = {}; // until read: rays = the count if known on the host, pairs = triangles per ray
    int klog_nslot[KLOG] = {};            // entry of klog_n_d that holds the launch's ray count, or -1
    bool klog_timed[KLOG] = {};           // the record's event pair was recorded (kernel_timing at its launch)
    int *klog_n_d = nullptr;              // list lengths, written by the stages' setup kernels
    int n_slot_next = 0;
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

// Enumerates the local pixels (x, local row) in 8x8 tiles, the tiles of a 32x32 block in
// Morton order, blocks row-major: 64 consecutive entries are one tile, 256 a 16x16 square.
static void build_tile_order(int width, int rows, std::vector<int32_t> &perm)
{
    perm.clear();
    perm.reserve((size_t)width * rows);
    for (int by = 0; by < rows; by += 32)
        for (int bx = 0; bx < width; bx += 32)
            for (int m = 0; m < 16; m++) {
                const int tx = bx + 8 * ((m & 1) | ((m >> 1) & 2)), ty = by + 8 * (((m >> 1) & 1) | ((m >> 2) & 2));
                if (tx >= width || ty >= rows) continue;
                for (int yy = ty; yy < ty + 8 && yy < rows; yy++)
                    for (int xx = tx; xx < tx + 8 && xx < width; xx++) perm.push_back(yy * width + xx);
            }
}

static int upload_row_map(rt_ctx *c)
{
    std::vector<int32_t> perm;
    build_tile_order(c->width, c->local_rows, perm);
    HIPCHK(c, hipMemcpyAsync(c->perm_d, perm.data(), perm.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->row_map_d, c->row_map_h.data(), c->row_map_h.size() * sizeof(int32_t),
                             hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

// The tables of the direction cells that depend on the resolution only: the cells' bounding cones at
// every level (one table for every origin), the per-cell counters and the scan's temporary storage.
// Lets go of the band sweep's lists (the context's own half-built ones, or its share of built ones).
static void drop_band_lists(rt_ctx *c)
{
    if (c->band_shared) {
        std::lock_guard<std::mutex> lock(g_band_mutex);
        c->band_shared.reset();
        for (auto it = g_band_cache.begin(); it != g_band_cache.end();)
            it = it->second.expired() ? g_band_cache.erase(it) : std::next(it);
    } else {
        (void)hipFree(c->band_off);
        (void)hipFree(c->band_entries);
        (void)hipFree(c->bucket_frames_d);
    }
    c->band_off = nullptr;
    c->band_entries = nullptr;
    c->bucket_frames_d = nullptr;
}

static void free_grid_tables(rt_ctx *c)
{
    (void)hipFree(c->cell_cones_d); c->cell_cones_d = nullptr;
    drop_band_lists(c); // (built for these cells)
    (void)hipFree(c->grid_count_d); c->grid_count_d = nullptr;
    (void)hipFree(c->scan_temp); c->scan_temp = nullptr;
    c->scan_temp_bytes = 0;
    for (int i = 0; i < 1 + RT_MAX_LIGHTS; i++) {
        (void)hipFree(c->grid_off[i]); c->grid_off[i] = nullptr;
        (void)hipFree(c->grid_entries[i]); c->grid_entries[i] = nullptr;
        c->grid_cap[i] = 0;
        c->grid_level[i] = 0;
        c->grid_uses[i] = 0;
    }
}

static int alloc_grid_tables(rt_ctx *c)
{
    if (!c->grid_bits) return RT_OK;
    const size_t ncells = (size_t)1 << (2 * c->grid_bits);
    const size_t nall = rtg::level_offset(c->grid_bits) + ncells;
    std::vector<rtg::CellCone> cc(nall);
    for (int l = rtg::MIN_LEVEL; l <= c->grid_bits; l++)
        for (size_t i = 0; i < ((size_t)1 << (2 * l)); i++)
            rtg::make_cell_cone((uint32_t)i, l, &cc[rtg::level_offset(l) + i]);
    HIPCHK(c, hipMalloc(&c->cell_cones_d, nall * sizeof(rtg::CellCone)));
    HIPCHK(c, hipMemcpy(c->cell_cones_d, cc.data(), nall * sizeof(rtg::CellCone), hipMemcpyHostToDevice));
    // per cell, plus the list "everywhere", plus one for the scan's total
    HIPCHK(c, hipMalloc(&c->grid_count_d, (ncells + 2) * sizeof(unsigned int)));
    HIPCHK(c, rocprim::exclusive_scan(nullptr, c->scan_temp_bytes, c->grid_count_d, c->grid_count_d, 0u, ncells + 2,
                                      rocprim::plus<unsigned int>()));
    HIPCHK(c, hipMalloc(&c->scan_temp, c->scan_temp_bytes + 256));
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
    CRCHK(static_stage_address(hip_device, &c->static_stage));
    // (both streams now, one after the other: streams are dealt onto the process's few hardware queues in the
    // order they are made, and a side stream made later can land on its main stream's queue -- the two sweeps
    // of a bounce pass then run in turn after all: 0.5 ms per frame, measured)
    CRCHK(hipStreamCreateWithFlags(&c->stream_own, hipStreamNonBlocking));
    CRCHK(hipStreamCreateWithFlags(&c->stream_side, hipStreamNonBlocking));
    CRCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    CRCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    c->stream = c->stream_own;
    CRCHK(hipMalloc(&c->planes, 14 * c->stride * sizeof(float)));
    CRCHK(hipMemsetAsync(c->planes, 0, 14 * c->stride * sizeof(float), c->stream));
    setup_planes(c);
    CRCHK(hipMalloc(&c->row_map_d, (size_t)height * sizeof(int)));
    CRCHK(hipMalloc(&c->perm_d, c->cap * sizeof(int)));
    CRCHK(hipMalloc(&c->list, c->cap * sizeof(int)));
    CRCHK(hipMalloc(&c->walk_slot_d, c->cap * sizeof(int)));
    CRCHK(hipMalloc(&c->walk_t_d, c->cap * sizeof(float)));
    CRCHK(hipMalloc(&c->walk_count_d, 2 * sizeof(int))); // [1]: the walk's hand-out counter
    CRCHK(hipMemsetAsync(c->walk_count_d, 0, 2 * sizeof(int), c->stream));
    CRCHK(hipMalloc(&c->alist[0], c->cap * sizeof(int)));
    CRCHK(hipMalloc(&c->alist[1], c->cap * sizeof(int)));
    for (int k = 0; k < 2; k++)
        CRCHK(hipMalloc(&c->chunk_filed_d[k], (c->cap / (BLOCK * COMPACT_ITERS) + 2) * sizeof(int)));
    CRCHK(hipMalloc(&c->sort_keys, c->cap * sizeof(unsigned int)));
    CRCHK(hipMalloc(&c->sort_slots, c->cap * sizeof(int)));
    {
        // keys are direction cells of at most MAX_BITS bits per axis (plus one bin for the scan's total)
        const size_t bins = ((size_t)1 << (2 * rtg::MAX_BITS)) + 2; // (the lists of an origin: cells + 2)
        CRCHK(hipMalloc(&c->bin_hist_d, bins * sizeof(unsigned int)));
        c->bin_hist_bytes = bins * sizeof(unsigned int);
        CRCHK(hipMalloc(&c->bin_off_d, bins * sizeof(unsigned int)));
        CRCHK(rocprim::exclusive_scan(nullptr, c->bin_temp_bytes, c->bin_hist_d, c->bin_off_d, 0u, bins,
                                      rocprim::plus<unsigned int>()));
        CRCHK(hipMalloc(&c->bin_temp_d, c->bin_temp_bytes + 256));
    }
    CRCHK(hipMalloc(&c->klog_n_d, rt_ctx::KLOG * sizeof(int)));
    CRCHK(hipMemsetAsync(c->klog_n_d, 0, rt_ctx::KLOG * sizeof(int), c->stream));
    {
        // 7 + 4 job planes + 6 float scratch planes + lit + occ + best (two words per slot)
        const size_t planes = 7 + 4 + 6 + 1 + 1 + 2;
        CRCHK(hipMalloc(&c->work, planes * c->stride * sizeof(float)));
        float *b = c->work;
        const size_t st = c->stride;
        c->job.ox = b + 0 * st; c->job.oy = b + 1 * st; c->job.oz = b + 2 * st;
        c->job.dx = b + 3 * st; c->job.dy = b + 4 * st; c->job.dz = b + 5 * st;
        c->job.tmax = b + 6 * st;
        c->scratch.sox = b + 7 * st; c->scratch.soy = b + 8 * st; c->scratch.soz = b + 9 * st;
        c->scratch.lr = b + 10 * st; c->scratch.lg = b + 11 * st; c->scratch.lb = b + 12 * st;
        c->scratch.lit = reinterpret_cast<int *>(b + 13 * st);
        c->scratch.occ = reinterpret_cast<int *>(b + 14 * st);
        c->scratch.best = reinterpret_cast<unsigned long long *>(b + 15 * st);
        c->job.bx = b + 17 * st; c->job.by = b + 18 * st; c->job.bz = b + 19 * st;
        c->job.g = b + 20 * st;
    }
    CRCHK(hipMalloc(&c->count_d, 8 * sizeof(int)));
    CRCHK(hipHostMalloc(&c->count_h, 8 * sizeof(int)));
    CRCHK(hipHostMalloc(&c->feedback_h, 4 * sizeof(int)));
    CRCHK(hipHostMalloc(&c->build_feedback_h, 6 * (1 + RT_MAX_LIGHTS) * sizeof(unsigned long long)));
    memset(c->build_feedback_h, 0, 6 * (1 + RT_MAX_LIGHTS) * sizeof(unsigned long long));
    c->feedback_h[0] = -1;
    CRCHK(hipMalloc(&c->sum_d, sizeof(unsigned long long)));
    CRCHK(hipHostMalloc(&c->sum_h, sizeof(unsigned long long)));
    if (alloc_grid_tables(c) != RT_OK) {
        g_create_error = c->error;
        rt_destroy(c);
        return RT_ERR_HIP;
    }
    CRCHK(hipMalloc(&c->fb_own, c->cap * sizeof(uint32_t)));
    CRCHK(hipMemsetAsync(c->fb_own, 0, c->cap * sizeof(uint32_t), c->stream));
    c->fb = c->fb_own;
    CRCHK(hipMalloc(&c->depth, c->cap * sizeof(float)));
    CRCHK(hipMemsetAsync(c->depth, 0, c->cap * sizeof(float), c->stream));
    CRCHK(hipMalloc(&c->counters_d, COUNTER_COPIES * COUNTER_STRIDE * sizeof(unsigned long long)));
    CRCHK(hipMemsetAsync(c->counters_d, 0, COUNTER_COPIES * COUNTER_STRIDE * sizeof(unsigned long long), c->stream));
    for (int i = 0; i < rt_ctx::KLOG; i++) {
        CRCHK(hipEventCreate(&c->klog_ev[i].a));
        CRCHK(hipEventCreate(&c->klog_ev[i].b));
    }
    c->local_rows = height;
    c->n_rays = width * height;
    c->row_map_h.resize(height);
    for (int y = 0; y < height; y++) c->row_map_h[y] = y;
    if (upload_row_map(c) != RT_OK) {
        g_create_error = c->error;
        rt_destroy(c);
        return RT_ERR_HIP;
    }
#undef CRCHK
    *out = c;
    return RT_OK;
}

extern "C" void rt_destroy(rt_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->stream_side) (void)hipStreamSynchronize(c->stream_side);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->stream_side) (void)hipStreamDestroy(c->stream_side);
    for (auto &e : c->ev_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < rt_ctx::KLOG; i++) {
        if (c->klog_ev[i].a) (void)hipEventDestroy(c->klog_ev[i].a);
        if (c->klog_ev[i].b) (void)hipEventDestroy(c->klog_ev[i].b);
    }
    (void)hipFree(c->planes);
    (void)hipFree(c->row_map_d);
    (void)hipFree(c->perm_d);
    (void)hipFree(c->verts);
    (void)hipFree(c->rec_general);
    (void)hipFree(c->rec_moment);
    (void)hipFree(c->rec_camera);
    (void)hipFree(c->rec_lights);
    (void)hipFree(c->cone_camera);
    (void)hipFree(c->cone_lights);
    (void)hipFree(c->ccone_camera);
    (void)hipFree(c->ccone_lights);
    (void)hipFree(c->slot_d);
    (void)hipFree(c->ngroup_d);
    (void)hipFree(c->nsub_d);
    (void)hipFree(c->nrec_d);
    (void)hipFree(c->vox_off_d);
    (void)hipFree(c->vox_entries_d);
    (void)hipFree(c->tex);
    (void)hipFree(c->list);
    (void)hipFree(c->walk_slot_d);
    (void)hipFree(c->walk_t_d);
    (void)hipFree(c->walk_count_d);
    (void)hipFree(c->chunk_filed_d[0]);
    (void)hipFree(c->chunk_filed_d[1]);
    (void)hipFree(c->alist[0]);
    (void)hipFree(c->alist[1]);
    (void)hipFree(c->sort_keys);
    (void)hipFree(c->sort_slots);
    (void)hipFree(c->bin_hist_d);
    (void)hipFree(c->bin_off_d);
    (void)hipFree(c->bin_temp_d);
    (void)hipFree(c->klog_n_d);
    (void)hipFree(c->cell_cones_d);
    (void)hipFree(c->grid_count_d);
    (void)hipFree(c->scan_temp);
    drop_band_lists(c);
    (void)hipFree(c->slot_views_d);
    (void)hipFree(c->build_count_d);
    (void)hipFree(c->build_overflow_d);
    (void)hipFree(c->build_item_keys_d);
    (void)hipFree(c->build_item_vals_d);
    if (c->build_feedback_h) (void)hipHostFree(c->build_feedback_h);
    (void)hipFree(c->build_keys_d);
    (void)hipFree(c->build_keys_out_d);
    (void)hipFree(c->build_vals_d);
    (void)hipFree(c->build_temp_d);
    for (int i = 0; i < 1 + RT_MAX_LIGHTS; i++) {
        (void)hipFree(c->grid_off[i]);
        (void)hipFree(c->grid_entries[i]);
    }
    (void)hipFree(c->work);
    (void)hipFree(c->count_d);
    if (c->count_h) (void)hipHostFree(c->count_h);
    if (c->feedback_h) (void)hipHostFree(c->feedback_h);
    (void)hipFree(c->fb_own);
    (void)hipFree(c->depth);
    (void)hipFree(c->counters_d);
    (void)hipFree(c->sum_d);
    if (c->sum_h) (void)hipHostFree(c->sum_h);
    if (c->stream_own) (void)hipStreamDestroy(c->stream_own);
    delete c;
}

// Lists of the band sweep's direction cells, from the member records of the mesh: count, scan, fill.
// *total = the sum of n 32-bit counts, in 64 bits (one stream sync)
static int sum_counts(rt_ctx *c, const unsigned int *v, size_t n, unsigned long long *total)
{
    HIPCHK(c, hipMemsetAsync(c->sum_d, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(sum_counts_kernel, dim3(cdiv((long long)n, BLOCK * 16) > 0 ? cdiv((long long)n, BLOCK * 16) : 1),
                       dim3(BLOCK), 0, c->stream, v, n, c->sum_d);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->sum_h, c->sum_d, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *total = *c->sum_h;
    return RT_OK;
}
// The cells in space of the mesh in hand (rt_voxel.h), from the vertices on the device and the radii the cluster
// build left behind.
static int build_voxels(rt_ctx *c)
{
    (void)hipFree(c->vox_off_d); c->vox_off_d = nullptr;
    (void)hipFree(c->vox_entries_d); c->vox_entries_d = nullptr;
    std::vector<rt_vertex> vertices((size_t)c->num_tris * 3 + 1);
    if (c->num_tris)
        HIPCHK(c, hipMemcpyAsync(vertices.data(), c->verts, (size_t)c->num_tris * 3 * sizeof(rt_vertex), hipMemcpyDeviceToHost,
                                 c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    rtv::VoxelSet vs;
    const double abs_extent = std::sqrt((double)c->center[0] * c->center[0] + (double)c->center[1] * c->center[1] +
                                        (double)c->center[2] * c->center[2]) + 3.0 * (double)c->radius;
    // (a mesh of 2^21 triangles or more never takes the clustered bounce path: rt_intersect sends its
    // bounce rays through the per-pair filter, so the cells stay empty rather than cost seconds of
    // box/triangle tests on the host and hundreds of MB that nothing reads)
    rtv::build(vertices.data(), c->num_tris < (1 << 21) ? c->num_tris : 0, c->vox_rho, abs_extent, c->voxel_res, vs);
    c->vox = vs.grid;
    HIPCHK(c, hipMalloc(&c->vox_off_d, vs.off.size() * sizeof(unsigned int)));
    HIPCHK(c, hipMalloc(&c->vox_entries_d, (vs.entries.size() + 4) * sizeof(int)));
    HIPCHK(c, hipMemcpyAsync(c->vox_off_d, vs.off.data(), vs.off.size() * sizeof(unsigned int), hipMemcpyHostToDevice,
                             c->stream));
    if (!vs.entries.empty())
        HIPCHK(c, hipMemcpyAsync(c->vox_entries_d, vs.entries.data(), vs.entries.size() * sizeof(int),
                                 hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream)); // vs goes out of scope
    return RT_OK;
}

static int build_band_cells(rt_ctx *c)
{
    if (!c->grid_bits || !c->num_ngroups) return RT_OK;
    const size_t ncells = (size_t)1 << (2 * c->grid_bits);
    const int members = c->num_ngroups * rtc::NGROUP;
    const dim3 grid((unsigned)(((long long)members * BUILD_TOP_CELLS + BLOCK - 1) / BLOCK));
    // The buckets tile the mesh's bounding box as each cell sees it.  Bounce rays start on the mesh, a
    // step of 0.001 directions off it: inside the box with a little room.
    drop_band_lists(c);
    // another context of this process may hold them already (the lock is kept while they are built, so
    // that a context asking meanwhile waits and then shares them)
    std::lock_guard<std::mutex> lock(g_band_mutex);
    const unsigned long long limit =
        c->band_buckets ? c->band_list_max : std::min<unsigned long long>(c->band_list_max, 1ull << 31);
    const BandKey key{c->device, c->num_vertices, c->grid_bits, c->band_buckets, limit, c->mesh_hash[0], c->mesh_hash[1]};
    {
        auto it = g_band_cache.find(key);
        std::shared_ptr<BandLists> held = it != g_band_cache.end() ? it->second.lock() : nullptr;
        if (held) {
            c->band_shared = held;
            c->band_off = held->off;
            c->band_entries = held->entries;
            c->bucket_frames_d = held->frames;
            c->band_g = held->g;
            c->band_total = held->total;
            if (getenv("RT_BAND_REPORT"))
                fprintf(stderr, "band lists: shared with another context (%d x %d buckets per cell, %llu entries)\n",
                        held->g, held->g, held->total);
            return RT_OK;
        }
    }
    if (c->band_buckets != 1) {
        double lo[3], hi[3];
        for (int a = 0; a < 3; a++) {
            const double room = 0.004 + 1e-3 * (c->mesh_hi[a] - c->mesh_lo[a]) + 1e-5 * std::fabs((double)c->center[a]);
            lo[a] = c->mesh_lo[a] - room - (double)c->center[a];
            hi[a] = c->mesh_hi[a] + room - (double)c->center[a];
        }
        std::vector<rtg::CellCone> cc(ncells);
        HIPCHK(c, hipMemcpy(cc.data(), (const rtg::CellCone *)c->cell_cones_d + rtg::level_offset(c->grid_bits),
                            ncells * sizeof(rtg::CellCone), hipMemcpyDeviceToHost));
        std::vector<rtg::BucketFrame> bf(ncells);
        for (size_t i = 0; i < ncells; i++) rtg::bucket_frame(cc[i], lo, hi, &bf[i]);
        HIPCHK(c, hipMalloc(&c->bucket_frames_d, ncells * sizeof(rtg::BucketFrame)));
        HIPCHK(c, hipMemcpy(c->bucket_frames_d, bf.data(), ncells * sizeof(rtg::BucketFrame), hipMemcpyHostToDevice));
    }
    // device buffers of the build, freed on every way out unless handed on
    struct DevBuf {
        void *p = nullptr;
        ~DevBuf() { (void)hipFree(p); }
        void *release()
        {
            void *q = p;
            p = nullptr;
            return q;
        }
    };
    auto scan = [&](unsigned int *cnt, unsigned int *off, size_t n) { // off[] = exclusive scan of cnt[0 .. n)
        DevBuf tmp;
        size_t bytes = 0;
        return rocprim::exclusive_scan(nullptr, bytes, cnt, off, 0u, n, rocprim::plus<unsigned int>(), c->stream) == hipSuccess &&
               hipMalloc(&tmp.p, bytes + 256) == hipSuccess &&
               rocprim::exclusive_scan(tmp.p, bytes, cnt, off, 0u, n, rocprim::plus<unsigned int>(), c->stream) == hipSuccess &&
               hipStreamSynchronize(c->stream) == hipSuccess; // (tmp is freed on return)
    };
    // ---- one list per cell: count, scan, fill
    DevBuf cnt1, off1, ent1;
    unsigned long long total1 = 0;
    HIPCHK(c, hipMalloc(&cnt1.p, (ncells + 2) * sizeof(unsigned int)));
    HIPCHK(c, hipMalloc(&off1.p, (ncells + 2) * sizeof(unsigned int)));
    HIPCHK(c, hipMemsetAsync(cnt1.p, 0, (ncells + 2) * sizeof(unsigned int), c->stream));
    hipLaunchKernelGGL((band_build_kernel<false>), grid, dim3(BLOCK), 0, c->stream, (const rtg::CellCone *)c->cell_cones_d,
                       c->grid_bits, (const float4 *)c->nrec_d, members, (unsigned int *)cnt1.p,
                       (const unsigned int *)nullptr, (int *)nullptr);
    HIPCHK(c, hipGetLastError());
    // List positions are 32-bit: the exact total decides.  Lists too long even so are not built, and the
    // streaming band sweep (traverse_band_kernel / traverse_group_kernel) takes over.
    if (int r = sum_counts(c, (const unsigned int *)cnt1.p, ncells + 2, &total1)) return r;
    if (total1 > limit) return RT_OK;
    if (!scan((unsigned int *)cnt1.p, (unsigned int *)off1.p, ncells + 2))
        return fail(c, RT_ERR_HIP, "build_band_cells: scanning the cells' counts failed");
    HIPCHK(c, hipMemsetAsync(cnt1.p, 0, (ncells + 2) * sizeof(unsigned int), c->stream));
    HIPCHK(c, hipMalloc(&ent1.p, ((size_t)total1 + 64) * sizeof(int)));
    hipLaunchKernelGGL((band_build_kernel<true>), grid, dim3(BLOCK), 0, c->stream, (const rtg::CellCone *)c->cell_cones_d,
                       c->grid_bits, (const float4 *)c->nrec_d, members, (unsigned int *)cnt1.p,
                       (const unsigned int *)off1.p, (int *)ent1.p);
    HIPCHK(c, hipGetLastError());
    // ---- each cell's list split into G x G origin buckets (band_bucket_kernel): if the lists would be
    // longer than the positions can address (or than rt_set_option allows), with fewer buckets, down to the
    // one list per cell above.  Left to choose, the context takes 16 x 16 and keeps the lists within 2^31
    // entries (8 GiB): `limit`.
    int built_g = 1;
    size_t nkeys = ncells;
    unsigned long long total64 = total1;
    for (int G = c->band_buckets ? c->band_buckets : 16; G > 1; G /= 2) {
        const size_t nk = ncells * ((size_t)G * G + 1);
        DevBuf cnt, off, ent;
        unsigned long long total = 0;
        // (a device short of memory gets fewer buckets, not an error)
        if (hipMalloc(&cnt.p, (nk + 2) * sizeof(unsigned int)) != hipSuccess ||
            hipMalloc(&off.p, (nk + 2) * sizeof(unsigned int)) != hipSuccess) {
            (void)hipGetLastError();
            continue;
        }
        HIPCHK(c, hipMemsetAsync((unsigned int *)cnt.p + nk, 0, 2 * sizeof(unsigned int), c->stream));
        hipLaunchKernelGGL((band_bucket_kernel<false>), dim3((unsigned)ncells + 1), dim3(BLOCK), 0, c->stream,
                           (const unsigned int *)off1.p, (const int *)ent1.p, c->grid_bits, (const float4 *)c->nrec_d,
                           (const rtg::BucketFrame *)c->bucket_frames_d, G, c->center[0], c->center[1], c->center[2],
                           (unsigned int *)cnt.p, (const unsigned int *)nullptr, (int *)nullptr);
        HIPCHK(c, hipGetLastError());
        if (int r = sum_counts(c, (const unsigned int *)cnt.p, nk + 2, &total)) return r;
        if (total > limit) {
            // (the lists' length is close to proportional to G: skip the counts that cannot fit)
            while (G > 2 && total / 2 > limit + limit / 4) {
                G /= 2;
                total /= 2;
            }
            continue;
        }
        if (!scan((unsigned int *)cnt.p, (unsigned int *)off.p, nk + 2))
            return fail(c, RT_ERR_HIP, "build_band_cells: scanning the buckets' counts failed");
        if (hipMalloc(&ent.p, ((size_t)total + 64) * sizeof(int)) != hipSuccess) {
            (void)hipGetLastError();
            continue;
        }
        hipLaunchKernelGGL((band_bucket_kernel<true>), dim3((unsigned)ncells + 1), dim3(BLOCK), 0, c->stream,
                           (const unsigned int *)off1.p, (const int *)ent1.p, c->grid_bits, (const float4 *)c->nrec_d,
                           (const rtg::BucketFrame *)c->bucket_frames_d, G, c->center[0], c->center[1], c->center[2],
                           (unsigned int *)cnt.p, (const unsigned int *)off.p, (int *)ent.p);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->band_off = (unsigned int *)off.release();
        c->band_entries = (int *)ent.release();
        built_g = G;
        nkeys = nk;
        total64 = total;
        break;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (built_g == 1) {
        c->band_off = (unsigned int *)off1.release();
        c->band_entries = (int *)ent1.release();
    }
    c->band_g = built_g;
    c->band_total = total64;
    {
        auto held = std::make_shared<BandLists>();
        held->device = c->device;
        held->off = c->band_off;
        held->entries = c->band_entries;
        held->frames = c->bucket_frames_d;
        held->g = built_g;
        held->total = total64;
        c->band_shared = held;
        g_band_cache[key] = held;
    }
    if (getenv("RT_BAND_REPORT")) { // diagnostic: what was built
        unsigned int e[2] = {0, 0};
        (void)hipMemcpy(e, c->band_off + nkeys, sizeof e, hipMemcpyDeviceToHost);
        fprintf(stderr, "band lists: %d x %d buckets per cell, %zu lists, %llu entries, %u of them on the list \"everywhere\"\n",
                built_g, built_g, nkeys, total64, e[1] - e[0]);
    }
    return RT_OK;
}

extern "C" int rt_set_mesh(rt_ctx *c, const rt_vertex *vertices, int num_vertices)
{
    if (!c) return RT_ERR_ARG;
    if (num_vertices < 0 || num_vertices % 3 != 0 || (num_vertices > 0 && !vertices))
        return fail(c, RT_ERR_ARG, "rt_set_mesh: num_vertices=%d must be a non-negative multiple of 3", num_vertices);
    // the direction-cell kernels queue (triangle, ray) pairs as triangle * 64 + ray in 32 bits
    if (num_vertices / 3 >= (1 << 26))
        return fail(c, RT_ERR_ARG, "rt_set_mesh: %d triangles; at most %d are supported", num_vertices / 3,
                    (1 << 26) - 1);
    if (int r = use_device(c)) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // until the new mesh is complete the context has none: a failure below leaves it that way
    c->feedback_h[0] = -1; // what earlier frames counted was another scene
    c->have_mesh = false;
    c->num_vertices = c->num_tris = 0;
    (void)hipFree(c->verts); c->verts = nullptr;
    (void)hipFree(c->rec_general); c->rec_general = nullptr;
    (void)hipFree(c->rec_moment); c->rec_moment = nullptr;
    (void)hipFree(c->rec_camera); c->rec_camera = nullptr;
    (void)hipFree(c->rec_lights); c->rec_lights = nullptr;
    (void)hipFree(c->cone_camera); c->cone_camera = nullptr;
    (void)hipFree(c->cone_lights); c->cone_lights = nullptr;
    (void)hipFree(c->ccone_camera); c->ccone_camera = nullptr;
    (void)hipFree(c->ccone_lights); c->ccone_lights = nullptr;
    (void)hipFree(c->slot_d); c->slot_d = nullptr;
    c->num_slots = 0;
    (void)hipFree(c->ngroup_d); c->ngroup_d = nullptr;
    (void)hipFree(c->nsub_d); c->nsub_d = nullptr;
    (void)hipFree(c->nrec_d); c->nrec_d = nullptr;
    c->num_ngroups = 0;
    (void)hipFree(c->vox_off_d); c->vox_off_d = nullptr;
    (void)hipFree(c->vox_entries_d); c->vox_entries_d = nullptr;
    drop_band_lists(c);
    c->band_cells_pending = false;
    c->vox_pending = false;
    c->vox_rho.clear();
    (void)hipFree(c->slot_views_d); c->slot_views_d = nullptr;
    (void)hipFree(c->build_overflow_d); c->build_overflow_d = nullptr;
    memset(c->build_feedback_h, 0, 6 * (1 + RT_MAX_LIGHTS) * sizeof(unsigned long long)); // counts of another mesh
    c->num_vertices = num_vertices;
    c->num_tris = num_vertices / 3;
    {
        // identifies the mesh among the contexts of the process (see BandLists)
        unsigned long long h1 = 0x243f6a8885a308d3ull, h2 = 0x13198a2e03707344ull;
        const size_t words = (size_t)num_vertices * sizeof(rt_vertex) / 8;
        const unsigned char *bytes = reinterpret_cast<const unsigned char *>(vertices);
        for (size_t i = 0; i < words; i++) {
            unsigned long long w;
            memcpy(&w, bytes + 8 * i, 8);
            h1 = (h1 ^ w) * 0x9e3779b97f4a7c15ull;
            h1 = (h1 << 29) | (h1 >> 35);
            h2 = (h2 + w) * 0xc2b2ae3d27d4eb4full;
            h2 ^= h2 >> 31;
        }
        c->mesh_hash[0] = h1;
        c->mesh_hash[1] = h2;
    }
    // record arrays are padded to whole tiles so that a tile copy never leaves the allocation
    c->rec_stride = ((size_t)c->num_tris + TILE - 1) / TILE * TILE + TILE;
    HIPCHK(c, hipMalloc(&c->verts, ((size_t)num_vertices + 3) * sizeof(rt_vertex)));
    HIPCHK(c, hipMemsetAsync(c->verts, 0, ((size_t)num_vertices + 3) * sizeof(rt_vertex), c->stream));
    if (num_vertices)
        HIPCHK(c, hipMemcpyAsync(c->verts, vertices, (size_t)num_vertices * sizeof(rt_vertex), hipMemcpyHostToDevice,
                                 c->stream));
    HIPCHK(c, hipMalloc(&c->rec_general, c->rec_stride * sizeof(rtf::TriRec)));
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
    for (int a = 0; a < 3; a++) {
        c->mesh_lo[a] = first ? 0.0 : lo[a];
        c->mesh_hi[a] = first ? 0.0 : hi[a];
    }
    // the two spheres of trace.glsl:159-162 belong to every scene: rays start on them too
    static const double sph[2][4] = {{0.0, -0.7, 2.0, 0.5}, {-2.0, 0.0, 0.0, 0.2}};
    for (int sidx = 0; sidx < 2; sidx++)
        for (int a = 0; a < 3; a++) {
            double l = sph[sidx][a] - sph[sidx][3], h = sph[sidx][a] + sph[sidx][3];
            if (first || l < lo[a]) lo[a] = l;
            if (first || h > hi[a]) hi[a] = h;
            if (sidx == 0 && a == 2) first = false;
        }
    double ctr[3] = {(lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, (lo[2] + hi[2]) / 2};
    double r2 = 0;
    for (int sidx = 0; sidx < 2; sidx++) {
        double dx = sph[sidx][0] - ctr[0], dy = sph[sidx][1] - ctr[1], dz = sph[sidx][2] - ctr[2];
        double d = std::sqrt(dx * dx + dy * dy + dz * dz) + sph[sidx][3];
        if (d * d > r2) r2 = d * d;
    }
    for (int i = 0; i < num_vertices; i++) {
        double dx = vertices[i].pos[0] - ctr[0], dy = vertices[i].pos[1] - ctr[1], dz = vertices[i].pos[2] - ctr[2];
        double d2 = dx * dx + dy * dy + dz * dz;
        if (std::isfinite(d2) && d2 > r2) r2 = d2;
    }
    for (int a = 0; a < 3; a++) c->center[a] = (float)ctr[a];
    c->radius = (float)(std::sqrt(r2) * 1.0001 + 1e-30);
    HIPCHK(c, hipMalloc(&c->rec_moment, c->rec_stride * sizeof(rtf::MomRec)));
    if (c->num_tris)
        hipLaunchKernelGGL(prep_moment_kernel, dim3(cdiv(c->num_tris, BLOCK)), dim3(BLOCK), 0, c->stream, c->verts,
                           c->num_tris, c->center[0], c->center[1], c->center[2], c->rec_moment);
    HIPCHK(c, hipGetLastError());
    {
        // plane bands and clusters for the clustered traversal (needs the bounding sphere)
        rtc::ClusterSet cs;
        rtc::build(vertices, c->num_tris, c->center, c->radius, cs);
        c->num_ngroups = cs.num_ngroups;
        const size_t ng = (size_t)cs.num_ngroups;
        // the shared-origin traversal's records live in slot order (a whole number of 1024-slot tiles)
        c->num_slots = (int)cs.slots.size();
        c->rec_stride = (size_t)c->num_slots + CONE_TILE;
        HIPCHK(c, hipMalloc(&c->slot_d, (c->rec_stride) * sizeof(int)));
        HIPCHK(c, hipMemsetAsync(c->slot_d, 0xff, c->rec_stride * sizeof(int), c->stream));
        if (c->num_slots)
            HIPCHK(c, hipMemcpyAsync(c->slot_d, cs.slots.data(), (size_t)c->num_slots * sizeof(int), hipMemcpyHostToDevice,
                                     c->stream));
        HIPCHK(c, hipMalloc(&c->rec_camera, c->rec_stride * sizeof(rtf::TriRec)));
        HIPCHK(c, hipMalloc(&c->rec_lights, RT_MAX_LIGHTS * c->rec_stride * sizeof(rtf::TriRec)));
        HIPCHK(c, hipMalloc(&c->cone_camera, c->rec_stride * sizeof(rtf::ConeRec)));
        HIPCHK(c, hipMalloc(&c->cone_lights, RT_MAX_LIGHTS * c->rec_stride * sizeof(rtf::ConeRec)));
        c->scone_offset = c->rec_stride / RUN + 1;
        c->ccone_stride = c->scone_offset + c->rec_stride / SUPER_RUN + 1;
        HIPCHK(c, hipMalloc(&c->ccone_camera, c->ccone_stride * sizeof(rtf::ConeRec)));
        HIPCHK(c, hipMalloc(&c->ccone_lights, RT_MAX_LIGHTS * c->ccone_stride * sizeof(rtf::ConeRec)));
        HIPCHK(c, hipMalloc(&c->ngroup_d, (ng + 1) * sizeof(float4)));
        HIPCHK(c, hipMalloc(&c->nsub_d, (ng * (rtc::NGROUP / rtc::SUBGROUP) + 1) * sizeof(float4)));
        HIPCHK(c, hipMalloc(&c->nrec_d, (ng * rtc::NGROUP + 1) * 2 * sizeof(float4)));
        if (ng) {
            HIPCHK(c, hipMemcpyAsync(c->ngroup_d, cs.ngroups.data(), ng * sizeof(float4), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->nsub_d, cs.nsubs.data(), ng * (rtc::NGROUP / rtc::SUBGROUP) * sizeof(float4),
                                     hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->nrec_d, cs.nrecs.data(), ng * rtc::NGROUP * 2 * sizeof(float4),
                                     hipMemcpyHostToDevice, c->stream));
        }
        // (the cells in space are built by the first bounce pass that walks them, like the band sweep's lists: a
        // host that only traces fresh rays and shadow rays never pays for them -- seconds of box / triangle tests
        // and 240 MB at 256K triangles)
        c->vox_rho = cs.rho;
        c->vox_pending = true;
        HIPCHK(c, hipStreamSynchronize(c->stream)); // the host vectors go out of scope
        // (the band sweep's lists are built by the first bounce pass that wants them: a host that only traces
        // fresh rays and shadow rays never pays for them -- 12 ms and 300 MB at 64K triangles)
        c->band_cells_pending = true;
    }
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
    if (!c->have_camera || memcmp(c->cam.pos, pos, sizeof(float) * 3) != 0) {
        c->camera_recs_dirty = true;
        // rays already generated still start at the old position: they no longer pass through the
        // camera's, and take the traversal of arbitrary rays
        c->rays_share_camera_origin = false;
        c->hits_follow_pixels = false;
    }
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
    c->feedback_h[0] = -1; // counts of another share of the frame
    c->rays_valid = false;
    c->rays_unwritten = false;
    c->chunk_filed_valid[0] = c->chunk_filed_valid[1] = false;
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
    if (mode != RT_TRACE_FILTERED && mode != RT_TRACE_EXACT && mode != RT_TRACE_CLUSTERED)
        return fail(c, RT_ERR_ARG, "rt_set_trace_mode: unknown mode %d", mode);
    c->trace_mode = mode;
    return RT_OK;
}

extern "C" int rt_set_option(rt_ctx *c, int option, long long value)
{
    if (!c) return RT_ERR_ARG;
    switch (option) {
    case RT_OPT_SORT_MIN_RAYS:
        if (value < 1 || value > (1ll << 30)) break;
        c->sort_min_rays = (int)value;
        return RT_OK;
    case RT_OPT_CLUSTER_MIN_TRIS:
        if (value < 0 || value > (1ll << 30)) break;
        c->cluster_min_tris = (int)value;
        return RT_OK;
    case RT_OPT_SWEEPS_IN_TURN:
        if (value != 0 && value != 1) break;
        c->sweeps_in_turn = value != 0;
        return RT_OK;
    case RT_OPT_BAND_LIST_MAX:
        if (value < 0 || value > 0xffffff00ll) break;
        c->band_list_max = (unsigned long long)value;
        c->have_mesh = false; // the lists are built by rt_set_mesh
        return RT_OK;
    case RT_OPT_BAND_BUCKETS:
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8 && value != 16) break;
        c->band_buckets = (int)value;
        c->have_mesh = false; // the lists are built for one bucket count
        return RT_OK;
    case RT_OPT_FAIL_NEXT_STAGE:
        c->fail_next_stage = value != 0;
        return RT_OK;
    case RT_OPT_MOVING_ORIGIN_LEVELS:
        if (value < 0 || value > 4) break;
        c->moving_origin_levels = (int)value;
        return RT_OK;
    case RT_OPT_EXACT_RAYS_PER_LANE:
        if (value != 0 && value != 2 && value != 4 && value != 8) break;
        c->exact_rays_per_lane = (int)value;
        return RT_OK;
    case RT_OPT_KERNEL_TIMING:
        if (value != 0 && value != 1) break;
        c->kernel_timing = value != 0;
        return RT_OK;
    case RT_OPT_WALK_BUDGET:
        if (value < 0 || value > (1 << 24)) break;
        c->walk_budget = (int)value;
        return RT_OK;
    case RT_OPT_VOXEL_RES:
        if (value < 0 || value > 128) break;
        c->voxel_res = (int)value;
        c->have_mesh = false; // the cells are built by rt_set_mesh
        return RT_OK;
    case RT_OPT_GRID_BITS: {
        if (!(value == 0 || (value >= 4 && value <= rtg::MAX_BITS))) break;
        if (int r = use_device(c)) return r;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        free_grid_tables(c); // (drops the band lists too: they are built for these cells)
        c->grid_bits = (int)value;
        c->have_mesh = false; // every list is built for one resolution
        c->camera_recs_dirty = c->light_recs_dirty = true;
        if (int r = alloc_grid_tables(c)) {
            // without its tables the context falls back to the kernels that need none, rather than keep a
            // resolution whose tables a later launch would find missing
            free_grid_tables(c);
            c->grid_bits = 0;
            return r;
        }
        return RT_OK;
    }
    default:
        return fail(c, RT_ERR_ARG, "rt_set_option: unknown option %d", option);
    }
    return fail(c, RT_ERR_ARG, "rt_set_option: value %lld out of range for option %d", value, option);
}

// ---- stage timing -----------------------------------------------------------
static int fold_events(rt_ctx *c)
{
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto &u : c->ev_used) {
        float t = 0.0f;
        HIPCHK(c, hipEventElapsedTime(&t, c->ev_pool[u.a], c->ev_pool[u.b]));
        c->acc_ms[u.stage] += t;
    }
    c->ev_used.clear();
    c->ev_next = 0;
    return RT_OK;
}

static int take_event(rt_ctx *c, int *idx)
{
    if (c->ev_next == c->ev_pool.size()) {
        hipEvent_t e;
        HIPCHK(c, hipEventCreate(&e));
        c->ev_pool.push_back(e);
    }
    *idx = (int)c->ev_next++;
    return RT_OK;
}
// (slot: the entry of ev_used; its end event is recorded by stage_end or by StageScope)
static int stage_begin(rt_ctx *c, int stage, size_t *slot)
{
    const bool chain = c->chain_stage_events && !c->ev_used.empty();
    c->chain_stage_events = false;
    if (c->ev_next >= 2048 && !chain)
        if (int r = fold_events(c)) return r;
    rt_ctx::StageSpan sp{stage, 0, 0};
    if (chain && !c->ev_used.empty()) {
        sp.a = c->ev_used.back().b; // where the stage call before this one ended
    } else {
        if (int r = take_event(c, &sp.a)) return r;
        HIPCHK(c, hipEventRecord(c->ev_pool[sp.a], c->stream));
    }
    if (int r = take_event(c, &sp.b)) return r;
    *slot = c->ev_used.size();
    c->ev_used.push_back(sp);
    return RT_OK;
}
static int stage_end(rt_ctx *c, size_t slot)
{
    HIPCHK(c, hipEventRecord(c->ev_pool[c->ev_used[slot].b], c->stream));
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

static bool use_grid(const rt_ctx *c);

// the rays rt_render left to the primary traversal, written after all (someone else wants the planes)
static int write_pending_rays(rt_ctx *c)
{
    if (!c->rays_unwritten) return RT_OK;
    c->rays_unwritten = false;
    if (c->n_rays > 0) {
        hipLaunchKernelGGL(raygen_kernel, dim3(cdiv(c->n_rays, BLOCK * 4)), dim3(BLOCK), 0, c->stream, c->rays,
                           c->n_rays, c->width, c->height, c->row_map_d, c->raygen_cam);
        HIPCHK(c, hipGetLastError());
    }
    return RT_OK;
}

// may_defer: the caller runs rt_intersect next (rt_render), so the launch can be left to the primary traversal
static int raygen_stage(rt_ctx *c, bool may_defer)
{
    if (int r = check_ready(c, false, false, "rt_raygen")) return r;
    size_t slot;
    if (int r = stage_begin(c, 0, &slot)) return r;
    c->rays_unwritten = may_defer && c->have_mesh && use_grid(c) && c->n_rays > 0;
    c->raygen_cam = c->cam;
    if (c->n_rays > 0 && !c->rays_unwritten) {
        hipLaunchKernelGGL(raygen_kernel, dim3(cdiv(c->n_rays, BLOCK * 4)), dim3(BLOCK), 0, c->stream, c->rays,
                           c->n_rays, c->width, c->height, c->row_map_d, c->cam);
        HIPCHK(c, hipGetLastError());
    }
    if (int r = stage_end(c, slot)) return r;
    c->rays_valid = true;
    c->chunk_filed_valid[0] = c->chunk_filed_valid[1] = false; // other rays
    c->rays_share_camera_origin = true;
    return RT_OK;
}

extern "C" int rt_raygen(rt_ctx *c)
{
    if (!c) return RT_ERR_ARG;
    return raygen_stage(c, false);
}

// ---- launch geometry ---------------------------------------------------------
// How many rays a stage has to trace is only known on the device (the compaction kernels count them),
// and the host never waits for it: every kernel over a list reads the list's length itself, and its
// launch has a number of workgroups that depends on the context's ray count alone -- enough for the
// longest list there can be, capped at MAX_LIST_BLOCKS, each workgroup (or wavefront) taking every
// n-th piece of the list.
static int list_blocks(long long n_upper, int per_block)
{
    const long long b = (n_upper + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > MAX_LIST_BLOCKS ? MAX_LIST_BLOCKS : b));
}

// workgroups of a direction-cell kernel (four wavefronts of R rays, R chosen on the device from the
// list's length n <= n_upper): enough for the n that needs the most
static int wave_blocks(int n_upper)
{
    long long need = 1;
    const int marks[4] = {n_upper, (1 << 16) - 1, (1 << 18) - 1, (1 << 21) - 1};
    for (int m : marks) {
        if (m > n_upper || m <= 0) continue;
        const long long b = ((long long)m + (BLOCK / 64) * rays_per_wave_for(m) - 1) / ((BLOCK / 64) * rays_per_wave_for(m));
        if (b > need) need = b;
    }
    return (int)need;
}

// rays per lane of the streaming kernels: more rays amortise each LDS record read over more
// arithmetic; fewer rays give small launches enough lanes to spread over 256 CUs.
static int pick_rays_per_lane(int n)
{
    if (n >= 65536) return 4;
    if (n >= 16384) return 2;
    return 1;
}

// splits `tiles` LDS tiles over blockIdx.y so that the launch has at least ~2048
// workgroups (8 per CU): short launches then fill the chip and end together.
static int pick_tiles_per_y(int ray_blocks, int tiles)
{
    const int target = 8192;
    int want_y = (target + ray_blocks - 1) / ray_blocks;
    if (want_y < 1) want_y = 1;
    if (want_y > tiles) want_y = tiles;
    return (tiles + want_y - 1) / want_y;
}

// Kernel log.  The number of rays of a launch is taken from n_slot: the entry of klog_n_d into which
// the stage's setup kernel wrote the length of its list (-1: fixed_rays, known on the host).
static int klog_begin(rt_ctx *c, int stage, int form, int k, bool exact, int n_slot, unsigned long long fixed_rays,
                      unsigned long long tris, size_t *slot)
{
    *slot = c->klog_count % rt_ctx::KLOG;
    rt_kernel_record &r = c->klog_rec[*slot];
    r.stage = stage; r.form = form; r.rays_per_lane = k; r.exact = exact ? 1 : 0;
    r.rays = fixed_rays; r.pairs = tris; r.ms = 0.0f; r.reserved = 0.0f;
    c->klog_nslot[*slot] = n_slot;
    c->klog_count++;
    c->klog_timed[*slot] = c->kernel_timing;
    if (c->kernel_timing) HIPCHK(c, hipEventRecord(c->klog_ev[*slot].a, c->stream));
    return RT_OK;
}
static int klog_end(rt_ctx *c, size_t slot)
{
    if (c->klog_timed[slot]) HIPCHK(c, hipEventRecord(c->klog_ev[slot].b, c->stream));
    return RT_OK;
}
// the entry of klog_n_d the next setup kernel writes its list's length to
static int next_n_slot(rt_ctx *c)
{
    c->n_slot_next = (c->n_slot_next + 1) % rt_ctx::KLOG;
    return c->n_slot_next;
}

template <int K, int FORM, bool ANYHIT>
static void launch_traverse_k(rt_ctx *c, const TraverseParams &p, dim3 grid)
{
    if (c->trace_mode == RT_TRACE_EXACT) // the all-pairs loop has one form for every kind of ray, and holds rays in pairs
        hipLaunchKernelGGL((traverse_exact_kernel<(K < 2 ? 2 : K), ANYHIT>), grid, dim3(BLOCK), 0, c->stream, p);
    else
        hipLaunchKernelGGL((traverse_kernel<K, FORM, ANYHIT>), grid, dim3(BLOCK), 0, c->stream, p);
}

// One launch of the streaming traversal: the rays of a list (at most n_upper of them) against the
// triangles [tri_begin, tri_end).  stage: 1 closest hit, 2 shadow.  (Its workgroups are laid out for
// n_upper rays; those beyond the list's end leave at once.)
static int launch_traverse(rt_ctx *c, TraverseParams p, int n_upper, int n_slot, int form, bool anyhit, int stage)
{
    int k = pick_rays_per_lane(n_upper);
    // (the all-pairs kernel holds rays in pairs; eight per lane amortise each LDS read of a triangle best)
    const bool exact = c->trace_mode == RT_TRACE_EXACT;
    if (exact) k = c->exact_rays_per_lane ? c->exact_rays_per_lane : (n_upper >= (1 << 18) ? 8 : (k < 2 ? 2 : k));
    const int ray_blocks = cdiv(n_upper, BLOCK * k);
    const int tiles = cdiv(p.tri_end - p.tri_begin, TILE);
    if (ray_blocks <= 0 || tiles <= 0) return RT_OK;
    // (a list of bounce rays is a fraction of the n_upper it is launched for -- an eighth, say -- and the
    // workgroups beyond its end leave at once: the all-pairs mode cuts the triangles finer to make up for them)
    p.tiles_per_y = pick_tiles_per_y(exact && form == 0 ? (ray_blocks + 7) / 8 : ray_blocks, tiles);
    // sub-ranges start on multiples of the larger (cone) tile
    p.tiles_per_y = (p.tiles_per_y + CONE_TILE / TILE - 1) / (CONE_TILE / TILE) * (CONE_TILE / TILE);
    // the shared-origin form queues pairs as (slot - first slot of the sub-range) * 2048 + ray in 32 bits
    if (p.tiles_per_y > (1 << 21) / TILE) p.tiles_per_y = (1 << 21) / TILE;
    const dim3 grid(ray_blocks, cdiv(tiles, p.tiles_per_y));
    size_t ks;
    if (int r = klog_begin(c, stage, form, k, exact, n_slot, (unsigned long long)n_upper,
                           (unsigned long long)(p.tri_end - p.tri_begin), &ks))
        return r;
#define RT_DISPATCH(KK)                                                             \
    do {                                                                            \
        if (form == 0 && !anyhit) launch_traverse_k<KK, 0, false>(c, p, grid);      \
        else if (form == 1 && !anyhit) launch_traverse_k<KK, 1, false>(c, p, grid); \
        else launch_traverse_k<KK, 1, true>(c, p, grid);                            \
    } while (0)
    if (k == 8) {
        if (anyhit) hipLaunchKernelGGL((traverse_exact_kernel<8, true>), grid, dim3(BLOCK), 0, c->stream, p);
        else hipLaunchKernelGGL((traverse_exact_kernel<8, false>), grid, dim3(BLOCK), 0, c->stream, p);
    } else if (k == 4) RT_DISPATCH(4);
    else if (k == 2) RT_DISPATCH(2);
    else RT_DISPATCH(1);
#undef RT_DISPATCH
    HIPCHK(c, hipGetLastError());
    return klog_end(c, ks);
}

// Rays grouped by key (any order inside a key): the kernel that forms the keys counts them into the
// histogram (bin_begin before its launch), then scan and scatter (bin_finish): out = the slots
// 0 .. *count - 1 in key order.
static int bin_begin(rt_ctx *c)
{
    // the scatter counts the histogram back down to zero: it only needs clearing when a call before this
    // one did not get as far as its scatter
    if (c->bin_hist_dirty) HIPCHK(c, hipMemsetAsync(c->bin_hist_d, 0, c->bin_hist_bytes, c->stream));
    c->bin_hist_dirty = true;
    return RT_OK;
}
// (bins: the keys are below bins - 1, the last bin receives the total; off_out: where the bins' first
// positions go, `bins` entries -- the context's own array when null;
// vals: what to file instead of the entries' indices)
static int bin_finish(rt_ctx *c, const unsigned int *keys, const int *count, int n_upper, size_t bins, int *out,
                      unsigned int *off_out = nullptr, const int *vals = nullptr)
{
    const dim3 grid(list_blocks(n_upper, BLOCK));
    unsigned int *off = off_out ? off_out : c->bin_off_d;
    size_t bytes = c->bin_temp_bytes;
    HIPCHK(c, rocprim::exclusive_scan(c->bin_temp_d, bytes, c->bin_hist_d, off, 0u, bins,
                                      rocprim::plus<unsigned int>(), c->stream));
    hipLaunchKernelGGL(bin_scatter_kernel, grid, dim3(BLOCK), 0, c->stream, keys, count, (const unsigned int *)off,
                       c->bin_hist_d, out, vals);
    HIPCHK(c, hipGetLastError());
    c->bin_hist_dirty = false;
    return RT_OK;
}

// The clustered closest-hit traversal of arbitrary rays (form 2 in the kernel log): two kernels over two
// orders of the same rays -- the band sweep over rays grouped by direction (a wavefront's rays then share
// the few lists or groups whose band holds them), the walk over the cells in space in pixel-tile order
// (neighbours start in the same cells).  Both merge into `best` by atomicMin.
static bool sweeps_side_by_side(const rt_ctx *c) { return !c->sweeps_in_turn; }

// Opens the kernel-log record of a clustered bounce pass and lets the side stream run from here on.
static int fork_side_stream(rt_ctx *c, int n_slot, size_t *ks)
{
    if (int r = klog_begin(c, 1, 2, 1, false, n_slot, 0ull, (unsigned long long)c->num_tris, ks)) return r;
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream_side, c->ev_fork, 0));
    return RT_OK;
}

static int launch_traverse_clustered(rt_ctx *c, TraverseParams p, int n_upper, int n_slot, bool sorted)
{
    p.ngroups = c->ngroup_d;
    p.nsubs = c->nsub_d;
    p.nrecs = c->nrec_d;
    p.num_ngroups = c->num_ngroups;
    p.vox = c->vox;
    p.vox_off = c->vox_off_d;
    p.vox_entries = c->vox_entries_d;
    if (n_upper <= 0 || p.tri_end <= p.tri_begin) return RT_OK;
    const unsigned long long tris = (unsigned long long)(p.tri_end - p.tri_begin);
    size_t ks;
    // The two sweeps are two kernels that only meet in `best` (atomicMin).  With the band lists they run
    // side by side, the walk over the cells in space on the side stream, and the log holds one record
    // (form 2) for all of it; otherwise one after the other, one record per kernel: the band sweep
    // (form 2), the walk (form 4) and its long walks (form 5), the last two with 0 pairs.
    const bool fork = c->band_entries && sweeps_side_by_side(c);
    if (fork) {
        // rt_intersect forks before it groups the rays for the band sweep (the walk does not wait for that)
        if (c->forked) ks = c->forked_ks;
        else if (int r = fork_side_stream(c, n_slot, &ks)) return r;
        c->forked = false;
    } else if (int r = klog_begin(c, 1, 2, 1, false, n_slot, 0ull, tris, &ks)) {
        return r;
    }
    // ---- the band sweep: ill-conditioned pairs
    TraverseParams a = p;
    if (sorted) a.alist = c->sort_slots; // rays grouped by direction share cells / groups; any order gives the same result
    if (c->band_entries) {
        a.band_off = c->band_off;
        a.band_g = c->band_g;
        a.band_cx = c->center[0]; a.band_cy = c->center[1]; a.band_cz = c->center[2];
        a.bucket_frames = (const rtg::BucketFrame *)c->bucket_frames_d;
        a.band_entries = c->band_entries;
        a.grid_bits = c->grid_bits;
        a.rays_per_wave = 0; // from the list's length, on the device
        hipLaunchKernelGGL(traverse_bandcell_kernel, dim3(wave_blocks(n_upper)), dim3(BLOCK), 0, c->stream, a);
    } else if (sorted) {
        const int ray_blocks = list_blocks(n_upper, BLOCK * 4);
        int want_y = (8192 + ray_blocks - 1) / ray_blocks;
        const int max_y = cdiv(c->num_ngroups, BAND_TILE_GROUPS);
        if (want_y > max_y) want_y = max_y;
        if (want_y < 1) want_y = 1;
        hipLaunchKernelGGL((traverse_band_kernel<4>), dim3(ray_blocks, want_y), dim3(BLOCK), 0, c->stream, a);
    } else {
        const int ray_blocks = list_blocks(n_upper, BLOCK);
        int want_y = (8192 + ray_blocks - 1) / ray_blocks;
        const int max_y = cdiv(c->num_ngroups, GROUP_TILE);
        if (want_y > max_y) want_y = max_y;
        if (want_y < 1) want_y = 1;
        hipLaunchKernelGGL(traverse_group_kernel, dim3(ray_blocks, want_y), dim3(BLOCK), 0, c->stream, a);
    }
    HIPCHK(c, hipGetLastError());
    if (!fork) {
        if (int r = klog_end(c, ks)) return r;
        if (int r = klog_begin(c, 1, 4, 1, false, n_slot, 0ull, 0ull, &ks)) return r;
    }
    // ---- the walk over the cells in space: well-conditioned pairs
    {
        hipStream_t st = fork ? c->stream_side : c->stream;
        p.walk_slot = c->walk_slot_d;
        p.walk_t = c->walk_t_d;
        p.walk_count = c->walk_count_d;
        p.walk_next = c->walk_count_d + 1;
        p.walk_budget = c->walk_budget;
        // (walk_count_d was zeroed by the kernel of this stage call that prepared the slots)
        // a fixed number of wavefronts, each with a run of the list (fewer when even the longest list there
        // can be gives them less than 64 rays each)
        const int walk_blocks = cdiv(n_upper, BLOCK) < VOXEL_WAVES / (BLOCK / 64) ? cdiv(n_upper, BLOCK)
                                                                                  : VOXEL_WAVES / (BLOCK / 64);
        hipLaunchKernelGGL(traverse_voxel_kernel, dim3(walk_blocks), dim3(BLOCK), 0, st, p);
        HIPCHK(c, hipGetLastError());
        if (!fork) {
            if (int r = klog_end(c, ks)) return r;
            if (int r = klog_begin(c, 1, 5, 1, false, n_slot, 0ull, 0ull, &ks)) return r;
        }
        // the long walks, one wavefront each (their number is only known on the device)
        const int long_blocks = list_blocks(n_upper, BLOCK / 64);
        hipLaunchKernelGGL(traverse_voxel_long_kernel, dim3(long_blocks < 2048 ? long_blocks : 2048), dim3(BLOCK), 0, st, p);
        HIPCHK(c, hipGetLastError());
    }
    if (fork) {
        HIPCHK(c, hipEventRecord(c->ev_join, c->stream_side));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    }
    return klog_end(c, ks);
}

// count[] -> off[] (exclusive scan over the cells, the list "everywhere" and one more entry that
// receives the total); reads the total back (one stream sync) and clears count[] for the fill pass
static int scan_cell_counts(rt_ctx *c, unsigned int *off, size_t *total)
{
    const size_t ncells = (size_t)1 << (2 * c->grid_bits);
    size_t bytes = c->scan_temp_bytes;
    HIPCHK(c, rocprim::exclusive_scan(c->scan_temp, bytes, c->grid_count_d, off, 0u, ncells + 2,
                                      rocprim::plus<unsigned int>(), c->stream));
    unsigned int *total_h = reinterpret_cast<unsigned int *>(c->count_h + 4);
    HIPCHK(c, hipMemcpyAsync(total_h, off + ncells + 1, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemsetAsync(c->grid_count_d, 0, (ncells + 2) * sizeof(unsigned int), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *total = *total_h;
    return RT_OK;
}

// Lists of the direction cells of one shared origin, from its cone and edge records: count, scan,
// fill.  Runs when the camera or a light has moved, not per frame.
static int build_grid(rt_ctx *c, int which, const rtf::ConeRec *cones, const rtf::TriRec *recs, const float origin[3])
{
    if (!c->grid_bits || !c->num_slots) return RT_OK;
    const double reach_max = rtf::cone_reach_max(origin, c->center, c->radius);
    const double extent = rtf::scene_extent(origin, c->center, c->radius);
    // Lists of cells of 2^grid_bits per axis, or one level coarser for an origin that moves with every
    // launch -- its previous lists served one launch only -- when the frame is small against the mesh:
    // building costs by the (slot, cell) pair, four times fewer a level up, and a launch of few rays
    // per slot does not earn the finer lists back.
    int level = c->grid_bits;
    if (c->grid_level[which] && c->grid_uses[which] <= 1 && c->grid_bits > 4 &&
        (long long)c->n_rays < 16ll * (long long)c->num_slots) {
        level = c->grid_bits - c->moving_origin_levels;
        if (level < 4) level = 4;
    }
    c->grid_level[which] = level;
    c->grid_uses[which] = 0;
    const size_t ncells = (size_t)1 << (2 * level);
    if (!c->grid_off[which])
        HIPCHK(c, hipMalloc(&c->grid_off[which], (((size_t)1 << (2 * c->grid_bits)) + 2) * sizeof(unsigned int)));
    const float4 *cn = reinterpret_cast<const float4 *>(cones);
    if (!c->slot_views_d) HIPCHK(c, hipMalloc(&c->slot_views_d, (size_t)c->num_slots * sizeof(rtg::SlotView)));
    hipLaunchKernelGGL(slot_view_kernel, dim3(cdiv(c->num_slots, BLOCK)), dim3(BLOCK), 0, c->stream, cn, c->num_slots,
                       recs, (const int *)c->slot_d, (const rt_vertex *)c->verts, origin[0], origin[1], origin[2], extent,
                       reach_max, c->slot_views_d);
    HIPCHK(c, hipGetLastError());
    // Every (cell, slot) pair in one launch, in no particular order.  The arrays hold build_pairs_cap pairs
    // and, behind them, one entry per slot for the slots a launch has no room for (they go on the list
    // "everywhere": grid_overflow_kernel).  The first build of an origin waits for the pair count and sizes
    // the arrays by it.  A later build of the same origin -- a light or a camera that moves from frame to
    // frame -- does not wait: it runs in the arrays there are, every kernel after it takes the count from
    // the device, and the count comes back to the host on its own (pinned memory), where the NEXT build of
    // the origin reads it and, when the arrays have become tight, waits once and enlarges them.
    if (!c->build_count_d) HIPCHK(c, hipMalloc(&c->build_count_d, 6 * sizeof(unsigned long long)));
    if (!c->build_overflow_d) HIPCHK(c, hipMalloc(&c->build_overflow_d, ((size_t)c->num_slots / 32 + 1) * sizeof(unsigned int)));
    // what this origin's last build counted: [0] pairs, [1] fault, [4] items
    volatile unsigned long long *fb = c->build_feedback_h + 6 * which;
    if (fb[1] != 0) return fail(c, RT_ERR_HIP, "direction cells: the list builder ran out of stack");
    const size_t seen = (size_t)fb[0];
    bool wait = !c->grid_entries[which] || !c->build_keys_d || seen == 0 || seen + seen / 8 > c->build_pairs_cap ||
                (size_t)fb[4] > c->build_items_cap ||
                c->grid_cap[which] < c->build_pairs_cap + (size_t)c->num_slots ||
                c->build_pairs_cap + (size_t)c->num_slots >= ((size_t)1 << 31);
    if (getenv("RT_LIST_REPORT")) // diagnostic: how this build runs, and what the origin's last build counted
        fprintf(stderr, "rt: direction cells of origin %d: %s; last build counted %zu pairs, room for %zu\n", which,
                wait ? "waits for the count" : "in flight", seen, c->build_pairs_cap);
    size_t total = 0;
    for (int attempt = 0;; attempt++) {
        if (!c->build_keys_d) {
            if (c->build_pairs_cap < (size_t)c->num_slots * 8 + 4096) c->build_pairs_cap = (size_t)c->num_slots * 8 + 4096;
            const size_t room = c->build_pairs_cap + (size_t)c->num_slots;
            HIPCHK(c, hipMalloc(&c->build_keys_d, room * sizeof(unsigned int)));
            HIPCHK(c, hipMalloc(&c->build_keys_out_d, room * sizeof(unsigned int)));
            HIPCHK(c, hipMalloc(&c->build_vals_d, room * sizeof(int)));
            // items: (cell three levels up, slot) of the wavefronts with heavy slots -- far fewer than the pairs they
            // stand for, so a quarter of the pairs' room is ample (what does not fit goes the overflow's way)
            (void)hipFree(c->build_item_keys_d); c->build_item_keys_d = nullptr;
            (void)hipFree(c->build_item_vals_d); c->build_item_vals_d = nullptr;
            if (c->build_items_cap < c->build_pairs_cap / 4 + 4096) c->build_items_cap = c->build_pairs_cap / 4 + 4096;
            HIPCHK(c, hipMalloc(&c->build_item_keys_d, c->build_items_cap * sizeof(unsigned int)));
            HIPCHK(c, hipMalloc(&c->build_item_vals_d, c->build_items_cap * sizeof(int)));
        }
        HIPCHK(c, hipMemsetAsync(c->build_count_d, 0, 6 * sizeof(unsigned long long), c->stream));
        HIPCHK(c, hipMemsetAsync(c->build_overflow_d, 0, ((size_t)c->num_slots / 32 + 1) * sizeof(unsigned int), c->stream));
        const GridItems items{c->build_item_keys_d, c->build_item_vals_d, c->build_count_d + 4,
                              (unsigned long long)c->build_items_cap, GRID_LIST_HEAVY_CELLS};
        hipLaunchKernelGGL((grid_list_kernel<false>), dim3(cdiv(c->num_slots, GRID_LIST_WAVES * 64)),
                           dim3(GRID_LIST_WAVES * 64), 0, c->stream, (const rtg::CellCone *)c->cell_cones_d, level,
                           (const rtg::SlotView *)c->slot_views_d, c->num_slots, c->build_keys_d, c->build_vals_d,
                           c->build_count_d, (unsigned long long)c->build_pairs_cap,
                           reinterpret_cast<int *>(c->build_count_d + 1), c->build_overflow_d, items);
        HIPCHK(c, hipGetLastError());
        // the items of the heavy slots, 64 to a wavefront.  Their number is on the device; the launch is sized
        // by twice what the origin's last build counted (by the room there is when this build waits anyway),
        // wavefronts beyond the number leave at once and, should there be more, each takes several runs
        long long item_blocks = cdiv((long long)c->build_items_cap, GRID_LIST_WAVES * 64);
        if (!wait) {
            const long long guess = cdiv(2 * (long long)fb[4] + 4096, GRID_LIST_WAVES * 64);
            if (guess < item_blocks) item_blocks = guess;
        }
        hipLaunchKernelGGL((grid_list_kernel<true>), dim3((unsigned int)item_blocks),
                           dim3(GRID_LIST_WAVES * 64), 0, c->stream, (const rtg::CellCone *)c->cell_cones_d, level,
                           (const rtg::SlotView *)c->slot_views_d, c->num_slots, c->build_keys_d, c->build_vals_d,
                           c->build_count_d, (unsigned long long)c->build_pairs_cap,
                           reinterpret_cast<int *>(c->build_count_d + 1), c->build_overflow_d, items);
        HIPCHK(c, hipGetLastError());
        hipLaunchKernelGGL(grid_overflow_kernel, dim3(cdiv(c->num_slots, BLOCK)), dim3(BLOCK), 0, c->stream,
                           (const unsigned int *)c->build_overflow_d, c->num_slots, (unsigned int)ncells, c->build_keys_d,
                           c->build_vals_d, c->build_count_d, (unsigned long long)c->build_pairs_cap);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync((void *)fb, c->build_count_d, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        if (!wait) {
            total = c->build_pairs_cap + (size_t)c->num_slots; // an upper bound: the kernels read the count themselves
            break;
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (fb[1] != 0) return fail(c, RT_ERR_HIP, "direction cells: the list builder ran out of stack");
        if (fb[0] > 0xffffff00ull)
            return fail(c, RT_ERR_NOMEM, "direction cells: %llu list entries for one origin exceed the 32-bit list "
                        "positions; lower RT_OPT_GRID_BITS", (unsigned long long)fb[0]);
        total = (size_t)fb[0];
        const size_t n_items = (size_t)fb[4];
        // (room to spare, so that the builds that follow without waiting stay inside the arrays; items that did
        // not fit took their pairs out of the count, so the build is repeated with room for them)
        if (n_items <= c->build_items_cap &&
            (total + total / 4 <= c->build_pairs_cap || (attempt && total <= c->build_pairs_cap)))
            break;
        if (attempt > 2) return fail(c, RT_ERR_HIP, "direction cells: the list builder counted %zu pairs three times over", total);
        (void)hipFree(c->build_keys_d); c->build_keys_d = nullptr;
        (void)hipFree(c->build_keys_out_d); c->build_keys_out_d = nullptr;
        (void)hipFree(c->build_vals_d); c->build_vals_d = nullptr;
        if (n_items > c->build_items_cap) c->build_items_cap = n_items + n_items / 2 + 1024;
        if (total + total / 4 > c->build_pairs_cap) c->build_pairs_cap = total + total / 2 + 1024;
    }
    if (c->grid_cap[which] < c->build_pairs_cap + (size_t)c->num_slots || !c->grid_entries[which]) {
        (void)hipFree(c->grid_entries[which]);
        c->grid_entries[which] = nullptr;
        c->grid_cap[which] = c->build_pairs_cap + (size_t)c->num_slots;
        HIPCHK(c, hipMalloc(&c->grid_entries[which], c->grid_cap[which] * sizeof(int)));
    }
    // by cell; the list "everywhere" has key num_cells: one bit more than the cells need
    const unsigned int key_bits = 2u * (unsigned int)level + 1u;
    const int *n_pairs = reinterpret_cast<const int *>(c->build_count_d + 3); // pairs to sort, on the device
    if (c->build_pairs_cap + (size_t)c->num_slots < ((size_t)1 << 31)) {
        // counting sort: pairs per cell, a scan that leaves the lists' first positions in grid_off
        // (off[cells + 1] = the total), a scatter that files the slots
        if (int r = bin_begin(c)) return r;
        hipLaunchKernelGGL(bin_count_kernel, dim3(list_blocks((long long)total, BLOCK)), dim3(BLOCK), 0, c->stream,
                           (const unsigned int *)c->build_keys_d, n_pairs, c->bin_hist_d);
        HIPCHK(c, hipGetLastError());
        return bin_finish(c, c->build_keys_d, n_pairs, (int)total, ncells + 2, c->grid_entries[which], c->grid_off[which],
                          c->build_vals_d);
    }
    // (lists of 2^31 entries or more: every such build waits, and `total` is the count)
    {
        size_t bytes = 0;
        HIPCHK(c, rocprim::radix_sort_pairs(nullptr, bytes, c->build_keys_d, c->build_keys_out_d, c->build_vals_d,
                                            c->grid_entries[which], total, 0, key_bits, c->stream));
        if (bytes > c->build_temp_bytes) {
            (void)hipFree(c->build_temp_d);
            c->build_temp_d = nullptr;
            c->build_temp_bytes = bytes + bytes / 4 + 256;
            HIPCHK(c, hipMalloc(&c->build_temp_d, c->build_temp_bytes));
        }
        bytes = c->build_temp_bytes;
        HIPCHK(c, rocprim::radix_sort_pairs(c->build_temp_d, bytes, c->build_keys_d, c->build_keys_out_d,
                                            c->build_vals_d, c->grid_entries[which], total, 0, key_bits, c->stream));
    }
    hipLaunchKernelGGL(grid_offsets_kernel, dim3(cdiv((long long)ncells + 2, BLOCK)), dim3(BLOCK), 0, c->stream,
                       (const unsigned int *)c->build_keys_out_d, (unsigned int)total, (unsigned int)ncells,
                       c->grid_off[which]);
    HIPCHK(c, hipGetLastError());
    return RT_OK;
}

// One launch of the direction-cell traversal over the rays of a list.  stage: 1 closest hit, 2 shadow.
static int launch_traverse_grid(rt_ctx *c, TraverseParams p, int n_upper, int n_slot, int which, bool anyhit, int stage)
{
    if (n_upper <= 0) return RT_OK;
    p.grid_off = c->grid_off[which];
    p.grid_entries = c->grid_entries[which];
    p.grid_bits = c->grid_level[which];
    c->grid_uses[which]++;
    p.num_slots = c->num_slots;
    size_t ks;
    if (int r = klog_begin(c, stage, 3, 1, false, n_slot, (unsigned long long)n_upper, (unsigned long long)c->num_tris, &ks))
        return r;
    p.rays_per_wave = p.count ? 0 : rays_per_wave_for(n_upper); // a device-side count: chosen there
    const dim3 grid(p.count ? wave_blocks(n_upper) : cdiv(n_upper, (BLOCK / 64) * p.rays_per_wave));
    if (anyhit) hipLaunchKernelGGL((traverse_grid_kernel<true>), grid, dim3(BLOCK), 0, c->stream, p);
    else hipLaunchKernelGGL((traverse_grid_kernel<false>), grid, dim3(BLOCK), 0, c->stream, p);
    HIPCHK(c, hipGetLastError());
    return klog_end(c, ks);
}

static int refresh_camera_recs(rt_ctx *c)
{
    if (c->camera_recs_dirty && c->num_slots) {
        hipLaunchKernelGGL(prep_origin_kernel, dim3(cdiv(c->num_slots, BLOCK)), dim3(BLOCK), 0, c->stream, c->verts,
                           (const int *)c->slot_d, c->num_slots, c->cam.pos[0], c->cam.pos[1], c->cam.pos[2],
                           rtf::scene_extent(c->cam.pos, c->center, c->radius),
                           rtf::cone_reach_max(c->cam.pos, c->center, c->radius), c->rec_camera, c->cone_camera);
        HIPCHK(c, hipGetLastError());
        hipLaunchKernelGGL(cluster_cone_kernel, dim3(cdiv(c->num_slots / RUN, BLOCK)), dim3(BLOCK), 0, c->stream,
                           (const rtf::ConeRec *)c->cone_camera, c->num_slots / RUN, RUN, c->ccone_camera);
        hipLaunchKernelGGL(cluster_cone_kernel, dim3(cdiv(c->num_slots / SUPER_RUN, BLOCK)), dim3(BLOCK), 0, c->stream,
                           (const rtf::ConeRec *)c->ccone_camera, c->num_slots / SUPER_RUN, SUPER_RUN / RUN,
                           c->ccone_camera + c->scone_offset);
        HIPCHK(c, hipGetLastError());
        if (int r = build_grid(c, 0, c->cone_camera, c->rec_camera, c->cam.pos)) return r;
    }
    c->camera_recs_dirty = false;
    return RT_OK;
}

static int refresh_light_recs(rt_ctx *c)
{
    if (c->light_recs_dirty && c->num_slots) {
        for (int a = 0; a < c->num_lights; a++) {
            hipLaunchKernelGGL(prep_origin_kernel, dim3(cdiv(c->num_slots, BLOCK)), dim3(BLOCK), 0, c->stream,
                               c->verts, (const int *)c->slot_d, c->num_slots, c->lights[a].pos[0], c->lights[a].pos[1],
                               c->lights[a].pos[2], rtf::scene_extent(c->lights[a].pos, c->center, c->radius),
                               rtf::cone_reach_max(c->lights[a].pos, c->center, c->radius),
                               c->rec_lights + (size_t)a * c->rec_stride,
                               c->cone_lights + (size_t)a * c->rec_stride);
            HIPCHK(c, hipGetLastError());
            rtf::ConeRec *cc = c->ccone_lights + (size_t)a * c->ccone_stride;
            hipLaunchKernelGGL(cluster_cone_kernel, dim3(cdiv(c->num_slots / RUN, BLOCK)), dim3(BLOCK), 0, c->stream,
                               (const rtf::ConeRec *)(c->cone_lights + (size_t)a * c->rec_stride), c->num_slots / RUN,
                               RUN, cc);
            hipLaunchKernelGGL(cluster_cone_kernel, dim3(cdiv(c->num_slots / SUPER_RUN, BLOCK)), dim3(BLOCK), 0,
                               c->stream, (const rtf::ConeRec *)cc, c->num_slots / SUPER_RUN, SUPER_RUN / RUN,
                               cc + c->scone_offset);
            HIPCHK(c, hipGetLastError());
            if (int r = build_grid(c, 1 + a, c->cone_lights + (size_t)a * c->rec_stride,
                                       c->rec_lights + (size_t)a * c->rec_stride, c->lights[a].pos)) return r;
        }
    }
    c->light_recs_dirty = false;
    return RT_OK;
}

static bool use_grid(const rt_ctx *c)
{
    return c->grid_bits > 0 && c->trace_mode == RT_TRACE_CLUSTERED && c->num_slots > 0;
}

// sort keys are direction cells: those of the lists when there are lists, else 10 bits per axis
static int sort_key_bits(const rt_ctx *c) { return 2 * (c->grid_bits ? c->grid_bits : 10); }

static SharedOrigin shared_origin(rt_ctx *c, const float origin[3], float sign)
{
    SharedOrigin so{};
    memcpy(so.origin, origin, sizeof so.origin);
    memcpy(so.center, c->center, sizeof so.center);
    so.radius = c->radius;
    so.reach_max = (float)rtf::cone_reach_max(origin, c->center, c->radius);
    so.sign = sign;
    so.key_bits = sort_key_bits(c) / 2;
    return so;
}

static void fill_traverse_common(rt_ctx *c, TraverseParams &p)
{
    p.job = c->job;
    p.verts = c->verts;
    p.num_tris = c->num_tris;
    memcpy(p.center, c->center, sizeof p.center);
    p.radius = c->radius;
    p.best = c->scratch.best;
    p.occ = c->scratch.occ;
    p.counters = c->counters_d;
}

// Records the end of a stage's event pair when the stage call returns, whichever way it returns.
struct StageScope {
    rt_ctx *c;
    size_t slot;
    bool open = false;
    ~StageScope()
    {
        if (open) (void)hipEventRecord(c->ev_pool[c->ev_used[slot].b], c->stream);
    }
};

extern "C" int rt_intersect(rt_ctx *c)
{
    if (!c) return RT_ERR_ARG;
    if (int r = check_ready(c, true, true, "rt_intersect")) return r;
    StageScope scope{c, 0};
    if (int r = stage_begin(c, 1, &scope.slot)) return r;
    scope.open = true;
    if (c->fail_next_stage) {
        c->fail_next_stage = false;
        return fail(c, RT_ERR_HIP, "rt_intersect: failure asked for through RT_OPT_FAIL_NEXT_STAGE");
    }
    // rays rt_render left unwritten are formed by the traversal of fresh rays below, or written now
    const bool form_rays = c->rays_unwritten && c->rays_share_camera_origin && use_grid(c) && c->n_rays > 0 &&
                           memcmp(&c->raygen_cam, &c->cam, sizeof(Camera)) == 0;
    if (!form_rays)
        if (int r = write_pending_rays(c)) return r;
    if (c->n_rays > 0) {
        const bool shared = c->rays_share_camera_origin;
        const int *list = nullptr, *count = nullptr;
        const int n_upper = c->n_rays; // the live rays are counted on the device; the host does not wait for it
        int n_slot = -1;
        bool fresh_in_kernel = false;
        // a few hundred triangles are streamed through the per-pair filter faster than their groups
        // are swept
        // (the group sweeps queue pairs as triangle * 2048 + ray in 32 bits: larger meshes take it too)
        const bool clustered = !shared && c->trace_mode == RT_TRACE_CLUSTERED &&
                               c->num_tris >= c->cluster_min_tris && c->num_tris < (1 << 21);
        const bool sorted = clustered && n_upper >= c->sort_min_rays && c->num_ngroups >= 64;
        if (clustered && c->band_cells_pending) {
            // the first bounce pass on this mesh: the direction cells of the band sweep (waits for their size)
            c->band_cells_pending = false;
            if (int r = build_band_cells(c)) return r;
        }
        if (clustered && c->vox_pending) { // ... and the cells in space
            c->vox_pending = false;
            if (int r = build_voxels(c)) return r;
        }
        if (shared) {
            // fresh rays: all live, all through camera_pos; visited in pixel-tile order
            if (int r = refresh_camera_recs(c)) return r;
            list = c->perm_d;
            // (with the direction cells the traversal kernel sets its rays up in registers: no set-up kernel)
            fresh_in_kernel = use_grid(c);
            if (!fresh_in_kernel) {
                hipLaunchKernelGGL(intersect_setup_kernel, dim3(list_blocks(n_upper, BLOCK)), dim3(BLOCK), 0, c->stream,
                                   c->rays, list, count, n_upper, c->job, c->scratch.best, 1,
                                   shared_origin(c, c->cam.pos, 1.0f), (unsigned int *)nullptr, (unsigned int *)nullptr,
                                   (int *)nullptr, c->walk_count_d);
                HIPCHK(c, hipGetLastError());
            }
        } else {
            // (zero already when the last stage call to finish was an rt_color, whose last kernel saw to it)
            if (!c->count_clean[0]) HIPCHK(c, hipMemsetAsync(c->count_d, 0, sizeof(int), c->stream));
            c->count_clean[0] = false;
            if (sorted)
                if (int r = bin_begin(c)) return r;
            // the live rays are filed under slots, and each slot is prepared by the thread that files it
            CompactWork work{};
            work.job = c->job;
            work.best = c->scratch.best;
            work.shared = clustered ? 2 : 0;
            work.so = shared_origin(c, c->cam.pos, 1.0f);
            work.keys = sorted ? c->sort_keys : nullptr;
            work.hist = sorted ? c->bin_hist_d : nullptr;
            work.zero_this = c->walk_count_d;
            hipLaunchKernelGGL((compact_kernel<0>), dim3(list_blocks(c->n_rays, BLOCK * COMPACT_ITERS)), dim3(BLOCK), 0,
                               c->stream, c->rays, (const int *)c->perm_d, (const int *)nullptr, (const int *)nullptr,
                               c->n_rays, (const int *)nullptr, c->list, c->count_d, (uint32_t *)nullptr, work,
                               c->chunk_filed_d[0], c->chunk_filed_valid[0] ? 1 : 0);
            HIPCHK(c, hipGetLastError());
            c->chunk_filed_valid[0] = true;
            list = c->list;
            count = c->count_d;
            n_slot = next_n_slot(c);
        }
        c->forked = false;
        if (clustered && c->band_entries && sweeps_side_by_side(c)) {
            if (int r = fork_side_stream(c, n_slot, &c->forked_ks)) return r;
            c->forked = true;
        }
        if (sorted)
            if (int r = bin_finish(c, c->sort_keys, count, n_upper, ((size_t)1 << sort_key_bits(c)) + 1, c->sort_slots)) return r;
        TraverseParams p{};
        fill_traverse_common(c, p);
        p.alist = nullptr;
        p.count = count;
        p.n = n_upper;
        p.recs = shared ? c->rec_camera : c->rec_general;
        p.mrecs = c->rec_moment;
        p.cones = c->cone_camera;
        const bool exact = c->trace_mode == RT_TRACE_EXACT;
        // shared origin: the records are in slot order; the all-pairs mode walks the mesh itself
        p.tri_of = (shared && !exact) ? c->slot_d : nullptr;
        p.ccones = (shared && c->trace_mode == RT_TRACE_CLUSTERED) ? c->ccone_camera : nullptr;
        p.scones = p.ccones ? c->ccone_camera + c->scone_offset : nullptr;
        p.reach_max = (float)rtf::cone_reach_max(c->cam.pos, c->center, c->radius);
        p.dir_sign = 1.0f;
        memcpy(p.origin, c->cam.pos, sizeof p.origin);
        p.tri_begin = 0;
        p.tri_end = (shared && !exact) ? c->num_slots : c->num_tris;
        if (clustered) {
            if (int r = launch_traverse_clustered(c, p, n_upper, n_slot, sorted)) return r;
        } else if (shared && use_grid(c)) {
            if (fresh_in_kernel) {
                p.fresh_perm = c->perm_d;
                p.fresh_raygen = form_rays ? 1 : 0;
                p.fresh_width = c->width;
                p.fresh_height = c->height;
                p.fresh_row_map = c->row_map_d;
                p.fresh_cam = c->raygen_cam;
                c->rays_unwritten = false; // (the kernel writes every plane of every ray)
                p.fresh_rays = c->rays;
                p.fresh_depth = c->depth;
                p.fresh_zero = c->count_d + 1;
            }
            if (int r = launch_traverse_grid(c, p, n_upper, n_slot, 0, false, 1)) return r;
        } else if (int r = launch_traverse(c, p, n_upper, n_slot, shared ? 1 : 0, false, 1)) {
            return r;
        }
        if (!fresh_in_kernel) { // (fresh rays are finished by the traversal kernel that set them up)
            hipLaunchKernelGGL(intersect_finish_kernel, dim3(list_blocks(n_upper, BLOCK)), dim3(BLOCK), 0, c->stream,
                               c->rays, list, count, n_upper, c->job, c->scratch.best, c->verts, c->num_tris, c->cam,
                               c->depth, c->counters_d, c->count_d + 1,
                               n_slot >= 0 ? c->klog_n_d + n_slot : (int *)nullptr);
            HIPCHK(c, hipGetLastError());
        }
        c->count_clean[1] = true;
    }
    c->hits_follow_pixels = c->rays_share_camera_origin;
    c->rays_share_camera_origin = false;
    return RT_OK;
}

// triangles per shadow chunk: rays found occluded in one chunk are dropped before the next
static int shadow_chunk_tris(int num_tris)
{
    const int parts = 2;
    int chunk = (num_tris + parts - 1) / parts;
    if (chunk < 4096) chunk = 4096;
    return (chunk + CONE_TILE - 1) / CONE_TILE * CONE_TILE;
}

extern "C" int rt_color(rt_ctx *c)
{
    if (!c) return RT_ERR_ARG;
    if (int r = check_ready(c, true, true, "rt_color")) return r;
    if (int r = write_pending_rays(c)) return r;
    StageScope scope{c, 0};
    if (int r = stage_begin(c, 2, &scope.slot)) return r;
    scope.open = true;
    if (c->fail_next_stage) {
        c->fail_next_stage = false;
        return fail(c, RT_ERR_HIP, "rt_color: failure asked for through RT_OPT_FAIL_NEXT_STAGE");
    }
    if (c->n_rays > 0) {
        if (c->num_lights == 0) {
            // no light: `shadowed` stays true for every pixel (raycol_cs.glsl:28-32, 82)
            hipLaunchKernelGGL(fill_grey_kernel, dim3(list_blocks(c->n_rays, BLOCK)), dim3(BLOCK), 0, c->stream, c->fb,
                               c->n_rays);
            HIPCHK(c, hipGetLastError());
        } else {
            if (int r = refresh_light_recs(c)) return r;
            // (zero already when the last stage call to finish was an rt_intersect, whose last kernel saw to it)
            if (!c->count_clean[1]) HIPCHK(c, hipMemsetAsync(c->count_d + 1, 0, sizeof(int), c->stream));
            c->count_clean[1] = false;
            // what an earlier first pass of this context counted (its last kernel stored the number in pinned host
            // memory; nothing waits for it, so it may be a frame old, or -1: none yet)
            const int seen_hits = *static_cast<volatile int *>(c->feedback_h);
            const bool scattered = seen_hits >= 0 && 2ll * seen_hits < (long long)c->n_rays;
            const int n0 = c->n_rays;
            // in the clustered mode the shadow rays are visited in the order of the direction in
            // which they leave the light (any order gives the same result); first-pass hits follow the
            // pixels and are left in that order -- unless the last first pass saw hits on less than half
            // of the frame: scattered objects rather than surfaces, whose hit points neighbouring pixels
            // do not share (64 views of the 256K soup: shading 0.45 -> 0.28 ms per view)
            const bool sorted = c->trace_mode == RT_TRACE_CLUSTERED && n0 >= c->sort_min_rays &&
                                c->num_slots >= 8192 && !(use_grid(c) && c->hits_follow_pixels && !scattered);
            // the rays with a finite origin are filed under slots (and counted on the device; the host does not
            // wait for it), each slot prepared by the thread that files it: for the loop over the lights, and
            // with the shadow ray to the first light (shadow_setup_kernel's work for the other lights)
            CompactWork work{};
            work.scratch = c->scratch;
            work.job = c->job;
            work.so = shared_origin(c, c->lights[0].pos, -1.0f);
            memcpy(work.light0, c->lights[0].pos, sizeof work.light0);
            if (sorted) {
                if (int r = bin_begin(c)) return r;
                work.keys = c->sort_keys;
                work.hist = c->bin_hist_d;
            }
            // (with RT_OPT_KERNEL_TIMING the two kernels of the shading epilogue are logged too: forms 6 and 7)
            size_t ks_epi = 0;
            if (c->kernel_timing)
                if (int r = klog_begin(c, 2, 6, 1, false, -1, (unsigned long long)c->n_rays, 0ull, &ks_epi)) return r;
            hipLaunchKernelGGL((compact_kernel<1>), dim3(list_blocks(c->n_rays, BLOCK * COMPACT_ITERS)), dim3(BLOCK), 0,
                               c->stream, c->rays, (const int *)c->perm_d, (const int *)nullptr, (const int *)nullptr,
                               c->n_rays, (const int *)nullptr, c->list, c->count_d + 1, c->fb, work,
                               c->chunk_filed_d[1], c->chunk_filed_valid[1] ? 1 : 0);
            HIPCHK(c, hipGetLastError());
            if (c->kernel_timing)
                if (int r = klog_end(c, ks_epi)) return r;
            c->chunk_filed_valid[1] = true;
            const int *cnt0 = c->count_d + 1;
            const dim3 g0(list_blocks(n0, BLOCK));
            int n_slot0 = -1;
            const bool exact = c->trace_mode == RT_TRACE_EXACT;
            const int n_trav = exact ? c->num_tris : c->num_slots; // slots, or triangles in the all-pairs mode
            const int chunk = shadow_chunk_tris(n_trav);
            for (int a = 0; a < c->num_lights; a++) {
                const int n_slot = next_n_slot(c);
                if (a == 0) {
                    n_slot0 = n_slot; // (prepared by the compaction; color_finish_kernel notes their number)
                } else {
                    if (sorted)
                        if (int r = bin_begin(c)) return r;
                    hipLaunchKernelGGL(shadow_setup_kernel, g0, dim3(BLOCK), 0, c->stream, cnt0, c->scratch, c->job,
                                       c->lights[a].pos[0], c->lights[a].pos[1], c->lights[a].pos[2],
                                       shared_origin(c, c->lights[a].pos, -1.0f), c->counters_d,
                                       sorted ? c->sort_keys : (unsigned int *)nullptr,
                                       sorted ? c->bin_hist_d : (unsigned int *)nullptr, c->klog_n_d + n_slot);
                    HIPCHK(c, hipGetLastError());
                }
                const int *alist = nullptr; // nullptr: slots 0..n-1
                if (sorted) {
                    if (int r = bin_finish(c, c->sort_keys, cnt0, n0, ((size_t)1 << sort_key_bits(c)) + 1, c->sort_slots)) return r;
                    alist = c->sort_slots;
                }
                const int *acount = cnt0;
                int pp = 0;
                if (use_grid(c)) {
                    TraverseParams p{};
                    fill_traverse_common(c, p);
                    p.alist = alist;
                    p.count = acount;
                    p.n = n0;
                    p.recs = c->rec_lights + (size_t)a * c->rec_stride;
                    p.tri_of = c->slot_d;
                    if (int r = launch_traverse_grid(c, p, n0, n_slot, 1 + a, true, 2)) return r;
                }
                for (int t0 = 0; t0 < n_trav && !use_grid(c); t0 += chunk) {
                    TraverseParams p{};
                    fill_traverse_common(c, p);
                    p.alist = alist;
                    p.count = acount;
                    p.n = n0;
                    p.recs = c->rec_lights + (size_t)a * c->rec_stride;
                    p.cones = c->cone_lights + (size_t)a * c->rec_stride;
                    p.tri_of = exact ? nullptr : c->slot_d;
                    p.ccones = c->trace_mode == RT_TRACE_CLUSTERED ? c->ccone_lights + (size_t)a * c->ccone_stride
                                                                   : nullptr;
                    p.scones = p.ccones ? p.ccones + c->scone_offset : nullptr;
                    p.reach_max = (float)rtf::cone_reach_max(c->lights[a].pos, c->center, c->radius);
                    p.dir_sign = -1.0f;
                    memcpy(p.origin, c->lights[a].pos, sizeof p.origin);
                    p.tri_begin = t0;
                    p.tri_end = (t0 + chunk < n_trav) ? t0 + chunk : n_trav;
                    if (int r = launch_traverse(c, p, n0, n_slot, 1, true, 2)) return r;
                    if (p.tri_end < n_trav) {
                        // keep the rays that are still unoccluded for the next chunk
                        int *ocount = c->count_d + 2 + pp;
                        HIPCHK(c, hipMemsetAsync(ocount, 0, sizeof(int), c->stream));
                        hipLaunchKernelGGL((compact_kernel<2>), dim3(list_blocks(n0, BLOCK * COMPACT_ITERS)),
                                           dim3(BLOCK), 0, c->stream, c->rays, (const int *)nullptr, alist, acount, n0,
                                           (const int *)c->scratch.occ, c->alist[pp], ocount, (uint32_t *)nullptr,
                                           CompactWork{}, (int *)nullptr, 0);
                        HIPCHK(c, hipGetLastError());
                        alist = c->alist[pp];
                        acount = ocount;
                        pp ^= 1;
                    }
                }
                // (the last light's contribution is added by color_finish_kernel: slots whose ray is still
                // unoccluded there are exactly the entries of alist)
                if (a + 1 < c->num_lights) {
                    hipLaunchKernelGGL(shadow_finish_kernel, g0, dim3(BLOCK), 0, c->stream, c->rays,
                                       (const int *)c->list, alist, acount, n0, c->scratch, c->job, c->lights[a], c->cam);
                    HIPCHK(c, hipGetLastError());
                }
            }
            if (c->kernel_timing)
                if (int r = klog_begin(c, 2, 7, 1, false, n_slot0, 0ull, 0ull, &ks_epi)) return r;
            hipLaunchKernelGGL(color_finish_kernel, g0, dim3(BLOCK), 0, c->stream, c->rays, (const int *)c->list, cnt0,
                               c->scratch, c->verts, c->num_vertices, (const uint32_t *)c->tex, c->tex_w, c->tex_h,
                               c->fb, c->count_d, c->hits_follow_pixels ? c->feedback_h : (int *)nullptr, c->job,
                               c->lights[c->num_lights - 1], c->cam, c->counters_d,
                               n_slot0 >= 0 ? c->klog_n_d + n_slot0 : (int *)nullptr);
            HIPCHK(c, hipGetLastError());
            if (c->kernel_timing)
                if (int r = klog_end(c, ks_epi)) return r;
            c->count_clean[0] = true;
        }
    }
    // a killed ray keeps its shifted origin: the rays no longer share an origin
    c->rays_share_camera_origin = false;
    c->hits_follow_pixels = false;
    return RT_OK;
}

extern "C" int rt_render(rt_ctx *c, int passes)
{
    if (!c) return RT_ERR_ARG;
    if (passes < 1) return fail(c, RT_ERR_ARG, "rt_render: passes=%d must be >= 1", passes);
    if (int r = raygen_stage(c, true)) return r;
    for (int pss = 0; pss < passes; pss++) {
        c->chain_stage_events = true;
        if (int r = rt_intersect(c)) return r;
        c->chain_stage_events = true;
        if (int r = rt_color(c)) return r;
    }
    c->chain_stage_events = false;
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
    if (int r = write_pending_rays(c)) return r;
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

// rt_read_rays without an allocation: the records are assembled in the device's static staging array.
extern "C" int rt_read_rays_static(rt_ctx *c, rt_ray *out)
{
    if (!c || !out) return RT_ERR_ARG;
    if (int r = use_device(c)) return r;
    if (c->n_rays == 0) return RT_OK;
    if ((size_t)c->n_rays > STATIC_STAGE_MAX_RAYS)
        return fail(c, RT_ERR_ARG, "rt_read_rays_static: %d rays exceed the static staging array (%zu rays of 80 bytes, "
                    "%zu MiB); use rt_read_rays or rt_set_row_tiles", c->n_rays, STATIC_STAGE_MAX_RAYS,
                    STATIC_STAGE_BYTES >> 20);
    if (int r = write_pending_rays(c)) return r;
    rt_ray *stage = reinterpret_cast<rt_ray *>(c->static_stage);
    hipLaunchKernelGGL(soa_to_aos_kernel, dim3(cdiv(c->n_rays, BLOCK)), dim3(BLOCK), 0, c->stream, c->rays, c->n_rays,
                       stage);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, stage, (size_t)c->n_rays * sizeof(rt_ray), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

extern "C" int rt_write_rays(rt_ctx *c, const rt_ray *in)
{
    if (!c || !in) return RT_ERR_ARG;
    if (int r = use_device(c)) return r;
    c->rays_unwritten = false; // every plane is overwritten
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
    c->chunk_filed_valid[0] = c->chunk_filed_valid[1] = false; // other rays
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

extern "C" int rt_get_counters(rt_ctx *c, uint64_t counters[RT_NUM_COUNTERS])
{
    if (!c || !counters) return RT_ERR_ARG;
    if (int r = use_device(c)) return r;
    static_assert(RT_NUM_COUNTERS <= COUNTER_STRIDE, "a copy holds every counter");
    std::vector<uint64_t> copies((size_t)COUNTER_COPIES * COUNTER_STRIDE);
    HIPCHK(c, hipMemcpyAsync(copies.data(), c->counters_d, copies.size() * sizeof(uint64_t), hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipMemsetAsync(c->counters_d, 0, copies.size() * sizeof(uint64_t), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < RT_NUM_COUNTERS; i++) counters[i] = 0;
    for (size_t i = 0; i < copies.size(); i++)
        if ((int)(i % COUNTER_STRIDE) < RT_NUM_COUNTERS) counters[i % COUNTER_STRIDE] += copies[i];
    return RT_OK;
}

extern "C" int rt_kernel_log(rt_ctx *c, rt_kernel_record *out, int max, int *n)
{
    if (!c || !n || (max > 0 && !out)) return RT_ERR_ARG;
    if (int r = use_device(c)) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<int> n_h(rt_ctx::KLOG);
    HIPCHK(c, hipMemcpy(n_h.data(), c->klog_n_d, rt_ctx::KLOG * sizeof(int), hipMemcpyDeviceToHost));
    const size_t have = c->klog_count < (size_t)rt_ctx::KLOG ? c->klog_count : (size_t)rt_ctx::KLOG;
    const size_t first = c->klog_count - have;
    int w = 0;
    for (size_t i = first; i < c->klog_count && w < max; i++) {
        const size_t slot = i % rt_ctx::KLOG;
        rt_kernel_record r = c->klog_rec[slot];
        if (c->klog_nslot[slot] >= 0) r.rays = (unsigned long long)n_h[c->klog_nslot[slot]];
        r.pairs = r.rays * r.pairs; // triangles per ray until now
        // (a launch that failed half-way left its record open: reported with 0 ms; so is an untimed one)
        if (!c->klog_timed[slot]) {
            r.ms = 0.0f;
        } else if (hipEventElapsedTime(&r.ms, c->klog_ev[slot].a, c->klog_ev[slot].b) != hipSuccess) {
            (void)hipGetLastError();
            r.ms = 0.0f;
        }
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
    // (no wait: the kernels already enqueued hold the pointer they were launched with)
    c->fb = device_rgba8 ? reinterpret_cast<uint32_t *>(device_rgba8) : c->fb_own;
    c->chunk_filed_valid[1] = false; // the new framebuffer has not seen the grey pixels of skipped chunks
    return RT_OK;
}
