/*
 * rt.h -- C ABI of the three-stage compute ray tracer (raygen -> intersect -> color)
 * for AMD Instinct MI355X (gfx950).
 *
 * This library replaces, call for call, the body of the reference's stage loop
 * ("OpenGL Raytracer/main.cpp":479-529): where the reference selects a compute
 * program, binds three images and two SSBOs, sets eight uniforms, uploads the
 * light UBO and calls glDispatchCompute between two glMemoryBarrier calls, a
 * host calls rt_set_* once per change and rt_raygen / rt_intersect / rt_color
 * once per stage.  Stage calls are enqueued on the context's HIP stream in call
 * order; that ordering replaces glMemoryBarrier(GL_ALL_BARRIER_BITS).
 *
 * Conventions
 *   - every function returns RT_OK (0) or a negative rt_status; nothing throws
 *     across the boundary; rt_last_error() gives the text of the last failure.
 *   - the caller owns every host pointer; the context owns all device memory
 *     (except a framebuffer bound with rt_bind_framebuffer).
 *   - one context per GPU; a context is not thread-safe; distinct contexts may
 *     be driven from distinct threads.
 *   - there is no CPU fallback: every entry point fails with RT_ERR_HIP when no
 *     usable gfx950 device is present.
 *
 * Record layouts are byte-identical to the reference's SSBO/UBO contents so
 * that dumps can be compared with it directly:
 *   rt_ray    = struct ray   of "OpenGL Raytracer/Ray.h":9-24 and
 *               "OpenGL Raytracer/definitions.glsl":1-10           (80 bytes)
 *   rt_vertex = the 8-float record UploadToSSBO packs,
 *               "OpenGL Raytracer/main.cpp":148-167 and
 *               "OpenGL Raytracer/definitions.glsl":26-29          (32 bytes)
 *   rt_light  = struct Light of "OpenGL Raytracer/main.cpp":29-35 and
 *               "OpenGL Raytracer/definitions.glsl":21-24          (32 bytes)
 */
#ifndef RT_H_
#define RT_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ray {
    float   origin[3];
    float   pad0;
    float   dir[3];
    float   pad1;
    float   color[3];
    float   t;
    int32_t primitiveID; /* -1 fresh, 0..N-1 triangle, N and N+1 spheres, -2 dead */
    float   pad2[3];
    float   n[3];
    float   pad5;
} rt_ray;

typedef struct rt_vertex {
    float pos[3];
    float pad0;
    float texCoord[2];
    float pad1[2];
} rt_vertex;

typedef struct rt_light {
    float pos[3];
    float pad0;
    float color[3];
    float pad1;
} rt_light;

#define RT_MAX_LIGHTS 10 /* "OpenGL Raytracer/definitions.glsl":49-51, main.cpp:24 */

typedef enum rt_status {
    RT_OK          = 0,
    RT_ERR_ARG     = -1, /* bad argument (null pointer, size out of range, ...) */
    RT_ERR_HIP     = -2, /* a HIP runtime call failed, or no gfx950 device       */
    RT_ERR_STATE   = -3, /* call order: e.g. rt_intersect before rt_set_mesh     */
    RT_ERR_NOMEM   = -4
} rt_status;

/* How the closest-hit and shadow traversals test a ray against the triangles.
 * All modes return bit-identical rays and pixels; they differ only in speed.
 *   RT_TRACE_FILTERED : a cheap conservative edge-function filter over every
 *                       (ray, triangle) pair, followed by the reference's own
 *                       Moller-Trumbore arithmetic on the few pairs that pass.
 *   RT_TRACE_EXACT    : the reference's arithmetic on every pair (validation).
 *   RT_TRACE_CLUSTERED: (the default) as FILTERED, but bounce rays visit far fewer pairs:
 *                       a front-to-back walk over the cells of a uniform grid in space
 *                       finds the pairs whose ray meets the triangle's plane at a fair
 *                       angle and stops behind the closest hit; a sweep over lists of
 *                       triangles by plane normal finds the grazing pairs the walk's
 *                       stopping rule cannot vouch for.  Only these reach the edge filter.
 */
typedef enum rt_trace_mode {
    RT_TRACE_FILTERED  = 0,
    RT_TRACE_EXACT     = 1,
    RT_TRACE_CLUSTERED = 2
} rt_trace_mode;

typedef struct rt_ctx rt_ctx; /* opaque */

/* Creates a context on HIP device `hip_device` for a width x height frame and
 * allocates the ray buffer (one ray per pixel) and an RGBA8 framebuffer.
 * Replaces the output texture + ray SSBO creation, main.cpp:331-337, 368-372. */
int rt_create(rt_ctx **out, int hip_device, int width, int height);
void rt_destroy(rt_ctx *ctx);

/* Uploads the triangle list: num_vertices 32-byte records, three per triangle
 * (num_vertices % 3 == 0; fewer than 2^26 triangles, RT_ERR_ARG beyond).  Meshes of
 * 2^21 triangles or more send their bounce rays through the per-pair filter in every
 * trace mode.  On failure the context is left without a mesh.  Replaces UploadToSSBO +
 * the num_vertices uniform, main.cpp:148-174, 413-414, 496, 505. */
int rt_set_mesh(rt_ctx *ctx, const rt_vertex *vertices, int num_vertices);

/* Uploads the mesh texture as w x h RGBA8 texels, row 0 = bottom row of the
 * image (what SOIL_FLAG_INVERT_Y gives the reference), main.cpp:391-398, 493. */
int rt_set_texture_rgba8(rt_ctx *ctx, const uint8_t *texels, int w, int h);

/* Uploads num_lights (0..RT_MAX_LIGHTS) light records and sets the num_lights
 * uniform.  Replaces the per-stage UBO upload, main.cpp:506-512. */
int rt_set_lights(rt_ctx *ctx, const rt_light *lights, int num_lights);

/* Sets camera_pos / camera_dir / camera_up / camera_right, main.cpp:497-500. */
int rt_set_camera(rt_ctx *ctx, const float pos[3], const float dir[3],
                  const float up[3], const float right[3]);

/* Image-tile sharding.  The frame is cut into tiles of `tile_rows` rows; this
 * context renders tiles first_tile, first_tile + tile_stride, ...  Its rays and
 * its framebuffer then hold only those rows, packed in increasing y.
 * rt_set_row_tiles(ctx, height, 0, 1) restores the whole frame (the default).
 * The reference has no counterpart (single GPU); pixels keep the coordinates
 * they have in the whole frame, so the union over shards is the whole frame. */
int rt_set_row_tiles(rt_ctx *ctx, int tile_rows, int first_tile, int tile_stride);
/* Number of rows / rays this context currently renders. */
int rt_local_rows(rt_ctx *ctx);
int rt_num_rays(rt_ctx *ctx);
/* Writes the frame y of each local row, in order, to rows[rt_local_rows()]. */
int rt_get_row_map(rt_ctx *ctx, int32_t *rows);

int rt_set_trace_mode(rt_ctx *ctx, int mode /* rt_trace_mode */);

/* Tuning, per context.  None of these changes a result bit; they choose between kernels that return
 * the same rays and pixels.  The reference has no counterpart (its one knob, the thread-group size
 * baked into the shader header, main.cpp:180-185, has no meaning here).
 *   RT_OPT_SORT_MIN_RAYS    launches of at least this many rays sort them by direction first
 *                           (default 131072; >= 1)
 *   RT_OPT_CLUSTER_MIN_TRIS meshes with fewer triangles send their bounce rays through the per-pair
 *                           filter instead of the group sweeps (default 1024; >= 0)
 *   RT_OPT_GRID_BITS        direction cells per axis = 2^bits, 4..10, or 0 = no direction cells: the
 *                           streaming kernels run instead (default 9)
 *   RT_OPT_SWEEPS_IN_TURN   1: the two kernels of a clustered bounce pass run one after the other on
 *                           one stream instead of side by side on two (profiling; default 0)
 *   RT_OPT_BAND_LIST_MAX    the band sweep's per-cell lists are built only if they hold at most this
 *                           many entries in total (default and maximum 0xffffff00: list positions
 *                           are 32-bit); beyond it the streaming band sweep runs
 *   RT_OPT_BAND_BUCKETS     each of those lists is split by the origin of the ray into G x G lists,
 *                           G = 1, 2, 4, 8 or 16: shorter lists to read per ray, about 1.5 G times the
 *                           memory; halved until RT_OPT_BAND_LIST_MAX is met.  0 (default): 16, halved until
 *                           the lists hold at most 2^31 entries (8 GiB) as well
 *   RT_OPT_VOXEL_RES        cells along the longest axis of the grid in space that bounce rays walk,
 *                           1..128, or 0 = chosen from the number of triangles (default 0)
 *   RT_OPT_WALK_BUDGET      loop steps a wavefront of that walk spends, one ray per lane, before the walks
 *                           still under way are handed to the kernel that gives each a wavefront of its
 *                           own; 0 = chosen from the number of rays (default: 96 from 2^18 rays on, else 32)
 *   RT_OPT_FAIL_NEXT_STAGE  for tests of a caller's error handling: with a value other than 0 the next
 *                           rt_intersect or rt_color fails with RT_ERR_HIP after it has begun (its timing
 *                           events open, nothing enqueued) and the option returns to 0
 *   RT_OPT_MOVING_ORIGIN_LEVELS  how many levels coarser (0..4, default 1) the direction cells of an
 *                           origin are when its previous lists served a single launch before it moved
 *                           and the frame has fewer than 16 rays per triangle: coarser lists are
 *                           built faster and read slower
 *   RT_OPT_KERNEL_TIMING    1: every traversal launch is bracketed by a hipEvent pair and rt_kernel_log
 *                           reports its duration; 0 (default): the log's records carry ms = 0.  An event
 *                           between two kernels costs the stream about 5 us of idle time, which a frame
 *                           of some thirty traversal launches should not pay unasked
 *   RT_OPT_EXACT_RAYS_PER_LANE  rays per lane (2, 4 or 8) of the all-pairs kernel of RT_TRACE_EXACT; 0 (default):
 *                           8 for launches of 2^18 rays or more, 4 from 65536, else 2
 * RT_OPT_GRID_BITS, RT_OPT_BAND_LIST_MAX, RT_OPT_BAND_BUCKETS and RT_OPT_VOXEL_RES decide what rt_set_mesh builds: after
 * setting one of them the context has no mesh until the next rt_set_mesh. */
typedef enum rt_option {
    RT_OPT_SORT_MIN_RAYS    = 0,
    RT_OPT_CLUSTER_MIN_TRIS = 1,
    RT_OPT_GRID_BITS        = 2,
    RT_OPT_SWEEPS_IN_TURN   = 3,
    RT_OPT_BAND_LIST_MAX    = 4,
    RT_OPT_VOXEL_RES        = 5,
    RT_OPT_WALK_BUDGET      = 6,
    RT_OPT_FAIL_NEXT_STAGE  = 7,
    RT_OPT_MOVING_ORIGIN_LEVELS = 8,
    RT_OPT_KERNEL_TIMING    = 9,
    RT_OPT_BAND_BUCKETS     = 10,
    RT_OPT_EXACT_RAYS_PER_LANE = 11
} rt_option;
int rt_set_option(rt_ctx *ctx, int option /* rt_option */, long long value);

/* The three stages, one call per glDispatchCompute of the reference:
 *   rt_raygen    = raygen_cs.glsl:1-22
 *   rt_intersect = rayintersect_cs.glsl:1-34 (+ trace.glsl:121-184)
 *   rt_color     = raycol_cs.glsl:18-128     (+ trace.glsl:121-184)
 * Asynchronous: they enqueue work on the context's stream and return; how many rays are alive is
 * counted and read on the device only.  One exception: the first rt_intersect after rt_set_camera
 * moved the camera's position, and the first rt_color after rt_set_lights moved a light, rebuild that
 * origin's triangle lists and wait for their sizes (a few stream synchronisations). */
int rt_raygen(rt_ctx *ctx);
int rt_intersect(rt_ctx *ctx);
int rt_color(rt_ctx *ctx);

/* The reference's pass order, main.cpp:476-529:
 * raygen, then `passes` times { intersect, color }.  Asynchronous. */
int rt_render(rt_ctx *ctx, int passes);

/* Blocks until everything enqueued on the context's stream has finished. */
int rt_sync(rt_ctx *ctx);

/* Ray buffer as 80-byte records (rt_num_rays() of them, local row-major with
 * stride = width).  Both synchronise.  The device keeps rays as structure-of-
 * arrays planes; these two calls transpose on the device. */
int rt_read_rays(rt_ctx *ctx, rt_ray *out);
int rt_write_rays(rt_ctx *ctx, const rt_ray *in);

/* rt_read_rays for callers that cannot allocate device memory (code between graph captures, tight
 * serving loops, harnesses that forbid hipMalloc): the same records, but the device-side staging of
 * the 80-byte layout lives in a static 64 MiB device array instead of a buffer allocated per call.
 * No counterpart in the reference.  At most 838 860 rays (64 MiB / 80 bytes; RT_ERR_ARG beyond:
 * shrink the context's share with rt_set_row_tiles or use rt_read_rays).  The array is one per GPU and
 * shared by every context on it: calls that overlap on one GPU race.  Synchronises; makes no
 * hipMalloc / hipFree call. */
int rt_read_rays_static(rt_ctx *ctx, rt_ray *out);

/* Local framebuffer (rt_local_rows() x width x RGBA8, row 0 = lowest y) and the
 * depth plane rayintersect_cs.glsl:29-31 writes (distance(origin, camera_pos),
 * kept as float32; 0 where intersect has not written).  Both synchronise. */
int rt_read_pixels_rgba8(rt_ctx *ctx, uint8_t *out);
int rt_read_depth(rt_ctx *ctx, float *out);

/* Milliseconds spent in each stage {raygen, intersect, color} by the stage
 * calls since the last rt_stage_ms / rt_create, measured with hipEvent pairs on
 * the context's stream (replaces OpenGLTimer.cpp:18-41; float, not truncated).
 * Inside rt_render one event ends a stage and begins the next, so the three
 * numbers of a frame add up to the frame.  Synchronises. */
int rt_stage_ms(rt_ctx *ctx, float ms[3]);

/* Work counters of the stage calls since the last rt_get_counters / rt_create.
 * Synchronises.
 *   [0] rays entering closest-hit traversal (live rays, summed over passes)
 *   [1] shadow rays traced (hit pixels x lights, summed over passes)
 *   [2] (ray, triangle) pairs tested by closest-hit traversal
 *   [3] (ray, triangle) pairs tested by shadow traversal (after early exit)
 *   [4] pairs that passed the filter in closest-hit traversal
 *   [5] pairs that passed the filter in shadow traversal
 *   [6] (wavefront, triangle) steps of closest-hit traversal that left the fast path
 *   [7] the same for shadow traversal
 *   [8] pixels shaded with at least one visible light: the vertex and texel gathers of
 *       raycol_cs.glsl:82-110 were made for them
 *   [9] rays killed by the color stage (raycol_cs.glsl:111-115: the ray is stored again)
 *   [10], [11] reserved (0) */
#define RT_NUM_COUNTERS 12
int rt_get_counters(rt_ctx *ctx, uint64_t counters[RT_NUM_COUNTERS]);

/* One record per traversal-kernel launch made by rt_intersect / rt_color since
 * the last rt_kernel_log / rt_create (at most the 1024 most recent launches).
 *   stage : 1 = closest-hit traversal (rt_intersect), 2 = shadow traversal + shading (rt_color)
 *   form  : 0 = filter for arbitrary rays, 1 = filter for rays sharing an origin,
 *           2 = clustered traversal of arbitrary rays: its band sweep and its walk over cells run
 *           side by side on two streams and share this one record, from before the first kernel
 *           to after the last; with RT_OPT_SWEEPS_IN_TURN (or without band lists) they run one
 *           after the other and form 2 is the band sweep alone, followed by 4 = the walk over
 *           cells, one lane per ray, and 5 = the long walks it handed over, one wavefront per ray
 *           (4 and 5 report 0 pairs), 3 = rays sharing an origin, each against the triangle list
 *           of its direction cell; with RT_OPT_KERNEL_TIMING also the two kernels of the shading
 *           epilogue (stage 2, 0 pairs): 6 = the pass over every pixel that files the rays to shade
 *           and writes the background, 7 = the shading of the filed rays (raycol_cs.glsl:52-127)
 *   exact : 1 = launched in the all-pairs mode (RT_TRACE_EXACT)
 *   rays_per_lane : 1, 2, 4 or 8
 *   rays  : rays the launch traversed
 *   pairs : (ray, triangle) pairs = rays x triangles (x lights for stage 2; an
 *           upper bound there, early exit may stop sooner)
 *   ms    : duration of that one launch, from a hipEvent pair around it (0 unless
 *           RT_OPT_KERNEL_TIMING was 1 when it was launched)
 * Writes at most `max` records, returns the number written in *n.  Synchronises. */
typedef struct rt_kernel_record {
    int32_t  stage;
    int32_t  form;
    int32_t  rays_per_lane;
    int32_t  exact;
    uint64_t rays;
    uint64_t pairs;
    float    ms;
    float    reserved;
} rt_kernel_record;
int rt_kernel_log(rt_ctx *ctx, rt_kernel_record *out, int max, int *n);

/* Interop with a caller that owns a HIP stream and/or device framebuffer (for
 * example a torch.distributed gather).  rt_bind_stream makes every later stage
 * call enqueue on `hip_stream` (a hipStream_t; NULL is HIP's default stream);
 * rt_unbind_stream returns to the context's own stream.  `device_rgba8` must
 * hold rt_local_rows() * width * 4 bytes on the context's device; NULL returns
 * to the context's own framebuffer.  The two stream calls synchronise the stream
 * in use.  rt_bind_framebuffer does not wait: stage calls enqueued before it keep
 * writing to the framebuffer they were enqueued with, which the caller keeps
 * alive until they have finished (a batch of views into one tensor is enqueued
 * without a wait between views). */
int rt_bind_stream(rt_ctx *ctx, void *hip_stream);
int rt_unbind_stream(rt_ctx *ctx);
int rt_bind_framebuffer(rt_ctx *ctx, void *device_rgba8);

const char *rt_last_error(rt_ctx *ctx); /* ctx may be NULL: last rt_create error */
const char *rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_H_ */
