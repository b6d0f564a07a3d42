// rt_kernels.h -- HIP kernels of the three stages for gfx950 (wave64, no MFMA).
//
// Device data layout
//   rays      : 14 structure-of-arrays planes of `stride` floats each
//               (ox oy oz dx dy dz cx cy cz t id nx ny nz), so that lane i of a
//               wave touches element i of a plane: every load/store is coalesced.
//   vertices  : the reference's 32-byte records, untouched (read by the exact path
//               and by the texture-coordinate interpolation only).
//   TriRec    : 48-byte filter records, one array for arbitrary rays, one per
//               shared origin (camera, each light); streamed through LDS in tiles
//               of 256 triangles and read back by wave-uniform (broadcast) LDS loads.
//   lists     : int32 ray indices produced by stream compaction, so that traversal
//               waves are full of live rays.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt.h"
#include "rt_exact.h"
#include "rt_filter.h"
#include "rt_grid.h"
#include "rt_voxel.h"

namespace rtk {

using rtx::v3;

constexpr int BLOCK = 256;      // threads per workgroup = 4 waves, one per SIMD
constexpr int TILE = 256;       // triangles per LDS tile (12 KiB)
constexpr int REC_FLOATS = 12;

struct RaySoA {
    float *ox, *oy, *oz, *dx, *dy, *dz, *cx, *cy, *cz, *t;
    int *id;
    float *nx, *ny, *nz;
};

struct Camera {
    float pos[3], dir[3], up[3], right[3];
};

// counters (see rt_get_counters)
enum { C_CH_RAYS = 0, C_SH_RAYS, C_CH_PAIRS, C_SH_PAIRS, C_CH_CAND, C_SH_CAND, C_CH_SLOW, C_SH_SLOW, C_LIT, C_KILLS };
constexpr int COUNTER_STRIDE = 16; // words per copy: the counters above and room to spare, two 64-byte lines

// The counters are kept in COUNTER_COPIES copies, on lines of their own, picked by the
// workgroup's index: atomics of one launch on one address are served one after the other (about 4 ns
// each), which a launch of 10^5 wavefronts would otherwise wait for.  rt_get_counters adds the copies.
constexpr int COUNTER_COPIES = 256;
__device__ __forceinline__ void add_counter(unsigned long long *c, int which, unsigned long long v)
{
    if (v) atomicAdd(&c[((blockIdx.x + blockIdx.y * 7u) % COUNTER_COPIES) * COUNTER_STRIDE + which], v);
}

// ---------------------------------------------------------------------------
// raygen  ("OpenGL Raytracer/raygen_cs.glsl":1-22)
// ---------------------------------------------------------------------------
// the direction of the ray through pixel (x, y) of the frame (raygen_cs.glsl:8-17)
__device__ __forceinline__ v3 raygen_direction(const Camera &cam, int width, int height, int x, int y)
{
    const float hw = (float)width / 2.0f;
    const float hh = (float)height / 2.0f;
    const v3 cpos = rtx::V(cam.pos[0], cam.pos[1], cam.pos[2]);
    const v3 cdir = rtx::V(cam.dir[0], cam.dir[1], cam.dir[2]);
    const v3 cup = rtx::V(cam.up[0], cam.up[1], cam.up[2]);
    const v3 cright = rtx::V(cam.right[0], cam.right[1], cam.right[2]);
    const v3 base = rtx::add(cpos, rtx::muls(cdir, 1.0f));
    const float fx = ((float)x - hw) / hw;
    const float fy = ((float)y - hh) / hh;
    const v3 s = rtx::add(rtx::add(base, rtx::muls(cright, fx)), rtx::muls(cup, fy));
    return rtx::normalize(rtx::sub(s, cpos));
}

// One thread writes four consecutive rays with 16-byte stores: 56 bytes per ray,
// nothing read but the row map.
__global__ __launch_bounds__(BLOCK) void raygen_kernel(RaySoA r, int n_rays, int width, int height,
                                                       const int *__restrict__ row_map, Camera cam)
{
    const int i0 = (blockIdx.x * BLOCK + threadIdx.x) * 4;
    if (i0 >= n_rays) return;
    const v3 cpos = rtx::V(cam.pos[0], cam.pos[1], cam.pos[2]);

    float ddx[4], ddy[4], ddz[4];
    int j = i0 / width;
    int x = i0 - j * width;
    int y = row_map[j];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const v3 d = raygen_direction(cam, width, height, x, y);
        ddx[e] = d.x; ddy[e] = d.y; ddz[e] = d.z;
        if (++x == width) {
            x = 0;
            ++j;
            if (i0 + e + 1 < n_rays) y = row_map[j];
        }
    }
    if (i0 + 3 < n_rays) {
        *reinterpret_cast<float4 *>(r.ox + i0) = make_float4(cpos.x, cpos.x, cpos.x, cpos.x);
        *reinterpret_cast<float4 *>(r.oy + i0) = make_float4(cpos.y, cpos.y, cpos.y, cpos.y);
        *reinterpret_cast<float4 *>(r.oz + i0) = make_float4(cpos.z, cpos.z, cpos.z, cpos.z);
        *reinterpret_cast<float4 *>(r.dx + i0) = make_float4(ddx[0], ddx[1], ddx[2], ddx[3]);
        *reinterpret_cast<float4 *>(r.dy + i0) = make_float4(ddy[0], ddy[1], ddy[2], ddy[3]);
        *reinterpret_cast<float4 *>(r.dz + i0) = make_float4(ddz[0], ddz[1], ddz[2], ddz[3]);
        *reinterpret_cast<float4 *>(r.cx + i0) = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        *reinterpret_cast<float4 *>(r.cy + i0) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        *reinterpret_cast<float4 *>(r.cz + i0) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        *reinterpret_cast<float4 *>(r.t + i0) = make_float4(-1.0f, -1.0f, -1.0f, -1.0f);
        *reinterpret_cast<int4 *>(r.id + i0) = make_int4(-1, -1, -1, -1);
        *reinterpret_cast<float4 *>(r.nx + i0) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        *reinterpret_cast<float4 *>(r.ny + i0) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        *reinterpret_cast<float4 *>(r.nz + i0) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    } else {
        for (int e = 0; e < 4 && i0 + e < n_rays; e++) {
            const int i = i0 + e;
            r.ox[i] = cpos.x; r.oy[i] = cpos.y; r.oz[i] = cpos.z;
            r.dx[i] = ddx[e]; r.dy[i] = ddy[e]; r.dz[i] = ddz[e];
            r.cx[i] = 1.0f; r.cy[i] = 0.0f; r.cz[i] = 0.0f;
            r.t[i] = -1.0f; r.id[i] = -1;
            r.nx[i] = 0.0f; r.ny[i] = 0.0f; r.nz[i] = 0.0f;
        }
    }
}

// ---------------------------------------------------------------------------
// 80-byte records <-> planes (rt_read_rays / rt_write_rays)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void soa_to_aos_kernel(RaySoA r, int n, rt_ray *__restrict__ out)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float4 *o = reinterpret_cast<float4 *>(out + i);
    o[0] = make_float4(r.ox[i], r.oy[i], r.oz[i], 0.0f);
    o[1] = make_float4(r.dx[i], r.dy[i], r.dz[i], 0.0f);
    o[2] = make_float4(r.cx[i], r.cy[i], r.cz[i], r.t[i]);
    o[3] = make_float4(__int_as_float(r.id[i]), 0.0f, 0.0f, 0.0f);
    o[4] = make_float4(r.nx[i], r.ny[i], r.nz[i], 0.0f);
}

__global__ __launch_bounds__(BLOCK) void aos_to_soa_kernel(RaySoA r, int n, const rt_ray *__restrict__ in)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 *s = reinterpret_cast<const float4 *>(in + i);
    float4 a = s[0], b = s[1], c = s[2], d = s[3], e = s[4];
    r.ox[i] = a.x; r.oy[i] = a.y; r.oz[i] = a.z;
    r.dx[i] = b.x; r.dy[i] = b.y; r.dz[i] = b.z;
    r.cx[i] = c.x; r.cy[i] = c.y; r.cz[i] = c.z; r.t[i] = c.w;
    r.id[i] = __float_as_int(d.x);
    r.nx[i] = e.x; r.ny[i] = e.y; r.nz[i] = e.z;
}

// ---------------------------------------------------------------------------
// filter records
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void prep_general_kernel(const rt_vertex *__restrict__ v, int num_tris,
                                                             rtf::TriRec *__restrict__ out)
{
    const int k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= num_tris) return;
    rtf::TriRec r;
    rtf::make_general_rec(v[3 * k].pos, v[3 * k + 1].pos, v[3 * k + 2].pos, &r);
    out[k] = r;
}

// Moment-form records of the per-pair filter (rt_filter.h, make_moment_rec), in triangle order.
__global__ __launch_bounds__(BLOCK) void prep_moment_kernel(const rt_vertex *__restrict__ v, int num_tris, float cx,
                                                            float cy, float cz, rtf::MomRec *__restrict__ out)
{
    const int k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= num_tris) return;
    const float C[3] = {cx, cy, cz};
    rtf::MomRec r;
    rtf::make_moment_rec(v[3 * k].pos, v[3 * k + 1].pos, v[3 * k + 2].pos, C, &r);
    out[k] = r;
}

// Edge and cone records for origin O, in slot order: slot i holds triangle order[i], or nothing
// (order[i] < 0: a record no ray passes).
__global__ __launch_bounds__(BLOCK) void prep_origin_kernel(const rt_vertex *__restrict__ v,
                                                            const int *__restrict__ order, int num_slots,
                                                            float ox, float oy, float oz, double extent,
                                                            double reach_max, rtf::TriRec *__restrict__ out,
                                                            rtf::ConeRec *__restrict__ cones)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= num_slots) return;
    const int k = order[i];
    rtf::TriRec r;
    rtf::ConeRec c;
    if (k < 0) {
        rtf::nan_rec(&r);
        rtf::cone_never(&c);
    } else {
        const float O[3] = {ox, oy, oz};
        rtf::make_origin_rec(v[3 * k].pos, v[3 * k + 1].pos, v[3 * k + 2].pos, O, &r);
        rtf::make_cone_rec(v[3 * k].pos, v[3 * k + 1].pos, v[3 * k + 2].pos, O, &r, extent, reach_max, &c);
    }
    // (the record's three spare floats: the first carries the slot's triangle, so that the traversal that holds
    // the record needs no second gather to name it)
    r.v[9] = __int_as_float(k);
    out[i] = r;
    cones[i] = c;
}

// One cone per run of `len` consecutive cones that contains them all: run cones from the slots'
// cones (len = RUN), cones of the level above from the run cones (len = SUPER_RUN / RUN).
using rtf::RUN;
using rtf::SUPER_RUN;
__global__ __launch_bounds__(BLOCK) void cluster_cone_kernel(const rtf::ConeRec *__restrict__ cones, int num_runs,
                                                             int len, rtf::ConeRec *__restrict__ out)
{
    const int c = blockIdx.x * BLOCK + threadIdx.x;
    if (c >= num_runs) return;
    rtf::ConeRec r;
    rtf::make_cluster_cone(cones + (size_t)c * len, len, &r);
    out[c] = r;
}

__global__ void fill_grey_kernel(uint32_t *__restrict__ fb, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) fb[i] = 0xff1a1a1au;
}

// Launches over compacted lists have a fixed number of workgroups (the length of a list is only known
// on the device): thread t takes entries t, t + T, t + 2T, ... of the list, T threads in all.
constexpr int MAX_LIST_BLOCKS = 8192;
#define RT_FOR_EACH_SLOT(slot, n) \
    for (int slot = blockIdx.x * BLOCK + threadIdx.x; slot < (n); slot += (int)gridDim.x * BLOCK)

// Rays per wavefront of the direction-cell kernels, from the number of rays of the launch: enough
// wavefronts to hide the list reads of each.
__host__ __device__ inline int rays_per_wave_for(int n)
{
    if (n >= 1 << 21) return 64;
    if (n >= 1 << 18) return 32;
    if (n >= 1 << 16) return 16;
    return 8;
}

// ---------------------------------------------------------------------------
// per-slot work arrays.  A "slot" is a position in a compacted list, so these are
// dense and every access is coalesced.
// ---------------------------------------------------------------------------
struct Job {            // the ray a traversal launch traces, exactly as the reference forms it
    float *ox, *oy, *oz; // origin (already offset: + dir*0.001 for intersect, + lightDir*1e-7 for shadows)
    float *dx, *dy, *dz; // direction
    float *tmax;         // shadow rays: distance to the light (raycol_cs.glsl:47)
    // shared-origin traversal only: what its two filter levels read, prepared once per ray
    float *bx, *by, *bz; // unit direction, pointing from the shared origin towards the triangles
    float *g;            // dir_scale() of that unit direction; negative when the ray is outside
                         // the range the cones were built for (then only the edge filter applies)
};

// fills job.b* and job.g for a ray with origin o and direction d (sign: see TraverseParams::dir_sign)
__device__ __forceinline__ void shared_direction(v3 o, v3 d, const float *origin, const float *center, float radius,
                                                 float reach_max, float sign, float &bx, float &by, float &bz, float &g)
{
    const float oo[3] = {o.x, o.y, o.z};
    const float reach = rtf::reach_shared(origin, oo, center, radius);
    const float inv = sign / sqrtf(rtx::dot(d, d));
    bx = d.x * inv; by = d.y * inv; bz = d.z * inv;
    const float gs = rtf::dir_scale(reach, bx, by, bz);
    g = (reach <= reach_max) ? gs : -gs;
}
__device__ __forceinline__ void prepare_shared(const Job &job, int slot, v3 o, v3 d, const float *origin,
                                               const float *center, float radius, float reach_max, float sign)
{
    float bx, by, bz, g;
    shared_direction(o, d, origin, center, radius, reach_max, sign, bx, by, bz, g);
    job.bx[slot] = bx; job.by[slot] = by; job.bz[slot] = bz;
    job.g[slot] = g;
}

// fills job.b* and job.g for an arbitrary ray (clustered traversal): unit direction, and the
// general filter's dir_scale, negated when the origin lies outside the range the band records
// were built for (|o - centre| <= 1.25 radius)
__device__ __forceinline__ void prepare_general(const Job &job, int slot, v3 o, v3 d, const float *center, float radius)
{
    const v3 oc = rtx::V(o.x - center[0], o.y - center[1], o.z - center[2]);
    const float dist = sqrtf(rtx::dot(oc, oc));
    const float g = rtf::dir_scale(dist + 3.0f * radius, d.x, d.y, d.z);
    const float inv = 1.0f / sqrtf(rtx::dot(d, d));
    job.bx[slot] = d.x * inv; job.by[slot] = d.y * inv; job.bz[slot] = d.z * inv;
    // (a direction so short or long that the walk over the cells in space would overflow counts as outside
    // too: the band sweep then takes every pair of the ray)
    const float dd = rtx::dot(d, d);
    job.g[slot] = (dist <= 1.25f * radius && dd > 1e-30f && dd < 1e30f) ? g : -g;
}

struct SharedOrigin { // the shared origin of a launch and the validity range of its cones
    float origin[3];
    float center[3];
    float radius;
    float reach_max;
    float sign;
    int key_bits; // sort keys: the ray's direction cell with this many bits per axis
};

struct Scratch {
    float *sox, *soy, *soz;    // color: r.origin as the light loop mutates it (raycol_cs.glsl:45)
    float *lr, *lg, *lb;       // color: `light` accumulator
    int *lit;                  // color: !shadowed
    int *occ;                  // shadow ray of the current light is occluded
    unsigned long long *best;  // intersect: (float bits of t) << 32 | primitive index, minimum wins
};

constexpr unsigned long long BEST_NONE = 0x7f80000000000000ull; // t = +inf, index 0 (trace.glsl:123,133)

// true iff the predicate holds in any lane of the wavefront (one v_cmp + a scalar test)
__device__ __forceinline__ bool any64(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0ull; }

// number of set bits of `mask` below this lane
__device__ __forceinline__ int lanes_below(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32),
                                          __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
}

__device__ __forceinline__ v3 vpos(const rt_vertex *__restrict__ v, int i)
{
    const float4 p = *reinterpret_cast<const float4 *>(v + i);
    return rtx::V(p.x, p.y, p.z);
}

// ---------------------------------------------------------------------------
// intersect, part 3 of 3, for one ray: the two spheres (trace.glsl:164-183), the normal of the winning
// primitive, and rayintersect_cs.glsl:16-33.  (o, d): the ray as it was traced (origin already offset by
// dir * 0.001); key: the closest triangle hit, (t bits, index), BEST_NONE for none.  The triangle normal
// depends only on the winning triangle, so it is formed once here instead of on every improvement inside
// the loop.
// ---------------------------------------------------------------------------
// unwritten: the ray comes straight from raygen and its planes were never written (the primary traversal
// formed it in registers): what a miss would keep is raygen's normal (0, 0, 0) and colour (1, 0, 0), and
// every plane is written here.
__device__ __forceinline__ void intersect_finish_slot(const RaySoA &r, int i, v3 o, v3 d, unsigned long long key,
                                                      const rt_vertex *__restrict__ verts, int num_tris,
                                                      const float *cam_pos, float *__restrict__ depth,
                                                      bool unwritten = false)
{
    float tb = __uint_as_float((unsigned int)(key >> 32));
    int id = (int)(unsigned int)(key & 0xffffffffull);
    int kind = (tb < INFINITY) ? 1 : 0; // 0 none, 1 triangle, 2 sphere
    v3 nrm = rtx::V(0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int s = 0; s < 2; s++) {
        float ts;
        if (rtx::ray_sphere(o, d, rtx::sphere_center(s), rtx::sphere_radius(s), ts)) {
            if (ts > 0.0f && ts < tb) {
                tb = ts;
                id = num_tris + s;
                kind = 2;
                v3 pt = rtx::add(o, rtx::muls(d, tb));
                nrm = rtx::normalize(rtx::sub(pt, rtx::sphere_center(s)));
            }
        }
    }
    if (kind == 1) {
        const v3 X = vpos(verts, 3 * id), Y = vpos(verts, 3 * id + 1), Z = vpos(verts, 3 * id + 2);
        nrm = rtx::normalize(rtx::cross(rtx::sub(X, Z), rtx::sub(Y, Z)));
    } else if (kind == 0 && !unwritten) {
        nrm = rtx::V(r.nx[i], r.ny[i], r.nz[i]); // a miss keeps the old normal
    }
    v3 ob = rtx::sub(o, rtx::muls(d, 0.001f));
    if (isinf(tb)) id = -2;
    ob = rtx::add(ob, rtx::muls(d, tb));
    const v3 dn = rtx::sub(d, rtx::muls(nrm, 2.0f * rtx::dot(d, nrm)));
    r.ox[i] = ob.x; r.oy[i] = ob.y; r.oz[i] = ob.z;
    r.dx[i] = dn.x; r.dy[i] = dn.y; r.dz[i] = dn.z;
    r.t[i] = tb;
    r.id[i] = id;
    if (kind != 0) {
        r.nx[i] = nrm.x; r.ny[i] = nrm.y; r.nz[i] = nrm.z;
        r.cx[i] = 0.0f; r.cy[i] = 0.0f; r.cz[i] = (kind == 2) ? 1.0f : 0.0f;
    } else if (unwritten) {
        r.nx[i] = 0.0f; r.ny[i] = 0.0f; r.nz[i] = 0.0f; // raygen_cs.glsl:18-20
        r.cx[i] = 1.0f; r.cy[i] = 0.0f; r.cz[i] = 0.0f;
    }
    depth[i] = rtx::distance3(ob, rtx::V(cam_pos[0], cam_pos[1], cam_pos[2]));
}

// ---------------------------------------------------------------------------
// intersect, part 1 of 3  ("OpenGL Raytracer/rayintersect_cs.glsl":6-12)
// ---------------------------------------------------------------------------

struct KeyGroup {
    int leader;        // the lowest lane with this lane's key
    unsigned int rank; // lanes with this key below this one
    unsigned int size; // lanes with this key
};
__device__ __forceinline__ KeyGroup group_by_key(unsigned int key, bool valid, int lane)
{
    KeyGroup g{lane, 0u, 1u};
    unsigned long long todo = __builtin_amdgcn_ballot_w64(valid);
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const unsigned int k0 = (unsigned int)__builtin_amdgcn_readlane((int)key, leader);
        const unsigned long long same = __builtin_amdgcn_ballot_w64(valid && key == k0);
        if (valid && key == k0) {
            g.leader = leader;
            g.rank = (unsigned int)lanes_below(same);
            g.size = (unsigned int)__popcll(same);
        }
        todo &= ~same;
    }
    return g;
}

// What rt_intersect prepares per slot of its list (rayintersect_cs.glsl:19-21): the ray as the reference
// traces it, the quantities the filters want, and -- when the launch groups its rays by direction cell --
// the slot's key, counted into hist[] here (bin_count_kernel's work).  All lanes of a wavefront that are
// still in the caller's loop call this together; `valid` says whether the lane has a slot.
__device__ __forceinline__ void intersect_setup_slot(const RaySoA &r, bool valid, int i, int slot, const Job &job,
                                                     unsigned long long *__restrict__ best, int shared,
                                                     const SharedOrigin &so, unsigned int *keys,
                                                     unsigned int *hist)
{
    unsigned int key = 0xffffffffu;
    if (valid) {
        v3 o = rtx::V(r.ox[i], r.oy[i], r.oz[i]);
        const v3 d = rtx::V(r.dx[i], r.dy[i], r.dz[i]);
        o = rtx::add(o, rtx::muls(d, 0.001f)); // r.origin = r.origin + r.dir * 0.001
        job.ox[slot] = o.x; job.oy[slot] = o.y; job.oz[slot] = o.z;
        job.dx[slot] = d.x; job.dy[slot] = d.y; job.dz[slot] = d.z;
        best[slot] = BEST_NONE;
        if (shared == 1) prepare_shared(job, slot, o, d, so.origin, so.center, so.radius, so.reach_max, so.sign);
        if (shared == 2) prepare_general(job, slot, o, d, so.center, so.radius);
        if (keys) {
            // the cell under which the band sweep files the ray (rt_grid.h)
            key = rtg::oct_cell_folded(d.x, d.y, d.z, so.key_bits);
            keys[slot] = key;
        }
    }
    if (keys && hist) {
        const KeyGroup g = group_by_key(key, valid, (int)(threadIdx.x & 63));
        if (valid && g.leader == (int)(threadIdx.x & 63)) atomicAdd(&hist[key], g.size);
    }
}

__global__ __launch_bounds__(BLOCK) void intersect_setup_kernel(RaySoA r, const int *__restrict__ list,
                                                                const int *__restrict__ count, int n_fixed, Job job,
                                                                unsigned long long *__restrict__ best, int shared,
                                                                SharedOrigin so, unsigned int *keys, unsigned int *hist,
                                                                int *n_out, int *zero_this)
{
    const int n = count ? *count : n_fixed;
    if (n_out && blockIdx.x == 0 && threadIdx.x == 0) *n_out = n; // for the kernel log
    // a counter that a later kernel of this stage call counts up from zero, and no kernel of this
    // launch or before it still reads (here: the number of walks handed over)
    if (zero_this && blockIdx.x == 0 && threadIdx.x == 0) *zero_this = 0;
    RT_FOR_EACH_SLOT(slot, n) {
        intersect_setup_slot(r, true, list ? list[slot] : slot, slot, job, best, shared, so, keys, hist);
    }
}

// What rt_color prepares per slot of its list before the loop over the lights.
// (the origin the loop over the lights mutates, s.so*, starts as the ray's: shadow_setup_slot of the first
// light, called right after this with that origin, stores it as that light leaves it)
__device__ __forceinline__ void color_begin_slot(int slot, const Scratch &s)
{
    s.lr[slot] = 0.0f; s.lg[slot] = 0.0f; s.lb[slot] = 0.0f;
    s.lit[slot] = 0;
}

// What rt_color prepares per slot for one light (raycol_cs.glsl:39-47): the shadow ray as the reference
// forms it from the ray's origin `o` as the lights before this one left it, the origin as this light
// leaves it, and what the shared-origin filters want; when the launch groups its rays by the direction
// cell in which the light sees them (rt_grid.h), the slot's key, counted into hist[] here
// (bin_count_kernel's work).  All lanes of a wavefront that are still in the caller's loop call this
// together; `valid` says whether the lane has a slot.
__device__ __forceinline__ void shadow_setup_slot(bool valid, int slot, v3 o, const Scratch &s, const Job &job,
                                                  float lx, float ly, float lz, const SharedOrigin &so,
                                                  unsigned int *keys, unsigned int *hist)
{
    unsigned int key = 0xffffffffu;
    if (valid) {
        const v3 lpos = rtx::V(lx, ly, lz);
        const v3 ld = rtx::normalize(rtx::sub(lpos, o));
        const v3 o2 = rtx::add(o, rtx::muls(ld, 0.0000001f)); // shadow-ray origin
        o = rtx::sub(o, rtx::muls(ld, 0.0000001f));           // r.origin = r.origin - lightDir * 1e-7
        s.sox[slot] = o.x; s.soy[slot] = o.y; s.soz[slot] = o.z;
        const float dist = rtx::distance3(lpos, o);
        job.ox[slot] = o2.x; job.oy[slot] = o2.y; job.oz[slot] = o2.z;
        job.dx[slot] = ld.x; job.dy[slot] = ld.y; job.dz[slot] = ld.z;
        job.tmax[slot] = dist;
        s.occ[slot] = (dist < INFINITY) ? 0 : 1; // `lightRay.t > d` can only hold for finite d
        prepare_shared(job, slot, o2, ld, so.origin, so.center, so.radius, so.reach_max, so.sign);
        if (keys) {
            key = rtg::oct_cell(-ld.x, -ld.y, -ld.z, so.key_bits);
            keys[slot] = key;
        }
    }
    if (keys && hist) {
        const KeyGroup g = group_by_key(key, valid, (int)(threadIdx.x & 63));
        if (valid && g.leader == (int)(threadIdx.x & 63)) atomicAdd(&hist[key], g.size);
    }
}

// ---------------------------------------------------------------------------
// stream compaction: out[] = the entries that satisfy a predicate; workgroup
// segments land in arbitrary order, entries keep their order inside a segment of
// 2048.
// Rays are visited through `perm`, the enumeration of the frame's pixels in 8x8 tiles
// (Morton order inside 32x32 blocks), so that 64 consecutive list entries are rays through
// neighbouring pixels: a wavefront's rays then see the same few triangles.
//   MODE 0: ray indices i in [0,n) that are live      (primitiveID != -2)   -- before intersect;
//           the thread that files a ray under its slot prepares the slot too (intersect_setup_slot)
//   MODE 1: ray indices with a finite origin                               -- before shading;
//           the others are reference-grey (see color_finish_kernel) and get their pixel here;
//           the thread that files a ray prepares its slot (color_begin_slot) and the slot's shadow ray
//           to the first light (shadow_setup_slot)
//   MODE 2: slots of in_list (or 0..n-1 when in_list is null) whose shadow ray is not yet
//           occluded (occ[slot] == 0)                                      -- between shadow chunks
// ---------------------------------------------------------------------------
constexpr int COMPACT_ITERS = 8;

struct CompactWork { // what MODE 0 / MODE 1 do for each entry they file (unused members stay zero)
    Job job;
    unsigned long long *best;
    int shared;
    SharedOrigin so;
    unsigned int *keys, *hist;
    Scratch scratch;
    int *zero_this; // a counter zeroed for a later kernel of the stage call (see intersect_setup_kernel)
    // MODE 1: the first light's shadow_setup_slot is done here too (job, so, keys and hist are its arguments)
    float light0[3];
};

template <int MODE>
__global__ __launch_bounds__(BLOCK) void compact_kernel(RaySoA r, const int *__restrict__ perm,
                                                        const int *__restrict__ in_list,
                                                        const int *__restrict__ in_count, int n_fixed,
                                                        const int *__restrict__ occ, int *__restrict__ out,
                                                        int *__restrict__ out_count, uint32_t *__restrict__ fb,
                                                        CompactWork work, int *__restrict__ chunk_filed,
                                                        int skip_empty)
{
    __shared__ int s_cnt[COMPACT_ITERS][BLOCK / 64];
    __shared__ int s_base;
    const int n = in_count ? *in_count : n_fixed;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (work.zero_this && blockIdx.x == 0 && tid == 0) *work.zero_this = 0;
    // the launch has a fixed number of workgroups; each takes every gridDim.x-th chunk
    for (int chunk = blockIdx.x * (BLOCK * COMPACT_ITERS); chunk < n; chunk += (int)gridDim.x * (BLOCK * COMPACT_ITERS)) {
    // chunk_filed[]: how many entries of each chunk were filed.  The sets MODE 0 and MODE 1 file only shrink
    // from pass to pass (a dead ray stays dead, a ray without a finite origin never gets one), so a chunk
    // that filed nothing last time -- most of a frame whose objects cover a part of it -- is skipped
    // (skip_empty: the counts are those of the last pass over the same rays).
    const int chunk_id = chunk / (BLOCK * COMPACT_ITERS);
    if (chunk_filed && skip_empty && chunk_filed[chunk_id] == 0) continue;
    unsigned long long masks[COMPACT_ITERS];
    int value[COMPACT_ITERS];
#pragma unroll
    for (int it = 0; it < COMPACT_ITERS; it++) {
        const int i = chunk + it * BLOCK + tid;
        bool p = false;
        value[it] = i;
        if (i < n) {
            if (MODE == 0) {
                const int ri = perm[i]; // rays are visited in pixel-tile order
                value[it] = ri;
                p = r.id[ri] != -2;
            } else if (MODE == 1) {
                const int ri = perm[i];
                value[it] = ri;
                float a = r.ox[ri], b = r.oy[ri], c = r.oz[ri];
                // finite <=> |x| < inf (false for NaN)
                p = (fabsf(a) < INFINITY) && (fabsf(b) < INFINITY) && (fabsf(c) < INFINITY);
                if (!p) fb[ri] = 0xff1a1a1au; // vec3(0.1) -> 26, alpha 255
            } else {
                const int slot = in_list ? in_list[i] : i;
                value[it] = slot;
                p = occ[slot] == 0;
            }
        }
        masks[it] = __ballot(p);
        if (lane == 0) s_cnt[it][wave] = __popcll(masks[it]);
    }
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int it = 0; it < COMPACT_ITERS; it++)
            for (int w = 0; w < BLOCK / 64; w++) {
                int c = s_cnt[it][w];
                s_cnt[it][w] = tot;
                tot += c;
            }
        s_base = tot ? atomicAdd(out_count, tot) : 0;
        if (chunk_filed) chunk_filed[chunk_id] = tot;
    }
    __syncthreads();
    const int base = s_base;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int it = 0; it < COMPACT_ITERS; it++) {
        const unsigned long long m = masks[it];
        const bool has = (m >> lane) & 1ull;
        const int slot = base + s_cnt[it][wave] + __popcll(m & lt);
        if (has) out[slot] = value[it];
        if (MODE == 0)
            intersect_setup_slot(r, has, value[it], slot, work.job, work.best, work.shared, work.so, work.keys, work.hist);
        if (MODE == 1) {
            v3 o = rtx::V(0.0f, 0.0f, 0.0f);
            if (has) {
                color_begin_slot(slot, work.scratch);
                o = rtx::V(r.ox[value[it]], r.oy[value[it]], r.oz[value[it]]);
            }
            shadow_setup_slot(has, slot, o, work.scratch, work.job, work.light0[0], work.light0[1], work.light0[2], work.so,
                              work.keys, work.hist);
        }
    }
    __syncthreads(); // s_cnt and s_base are reused by the next chunk
    }
}

// ---------------------------------------------------------------------------
// Rays grouped by a key (the direction cell in which a list files them): out[] = the slots 0..n-1
// ordered by key, in any order inside a key.  Count per key (by the kernel that forms the keys:
// intersect_setup_slot, shadow_setup_kernel; bin_count_kernel for keys that exist already), exclusive
// scan (rocprim), scatter; n is read on the
// device.  hist[] must be zero before the count and is zero again after the scatter.
// ---------------------------------------------------------------------------
// (Rays of a wavefront often share a key -- neighbouring pixels, one cell -- and atomics on one address are
// served one after the other: a wavefront issues one atomic per distinct key, on behalf of all its lanes.
// The lanes are first grouped by key with ballots alone; then all the groups' leaders issue their atomics
// in ONE instruction, so a wavefront waits for one round trip, not one per key.)
__global__ __launch_bounds__(BLOCK) void bin_count_kernel(const unsigned int *__restrict__ keys,
                                                          const int *__restrict__ count, unsigned int *__restrict__ hist)
{
    const int n = *count;
    const int lane = threadIdx.x & 63;
    for (int base = blockIdx.x * BLOCK; base < n; base += (int)gridDim.x * BLOCK) {
        const int slot = base + threadIdx.x;
        const bool valid = slot < n;
        const unsigned int key = valid ? keys[slot] : 0xffffffffu;
        const KeyGroup g = group_by_key(key, valid, lane);
        if (valid && g.leader == lane) atomicAdd(&hist[key], g.size);
    }
}

__global__ __launch_bounds__(BLOCK) void bin_scatter_kernel(const unsigned int *__restrict__ keys,
                                                            const int *__restrict__ count,
                                                            const unsigned int *__restrict__ off,
                                                            unsigned int *__restrict__ hist, int *__restrict__ out,
                                                            const int *__restrict__ vals)
{
    const int n = *count;
    const int lane = threadIdx.x & 63;
    for (int base = blockIdx.x * BLOCK; base < n; base += (int)gridDim.x * BLOCK) {
        const int slot = base + threadIdx.x;
        const bool valid = slot < n;
        const unsigned int key = valid ? keys[slot] : 0xffffffffu;
        const KeyGroup g = group_by_key(key, valid, lane);
        unsigned int first = 0;
        if (valid && g.leader == lane) first = atomicSub(&hist[key], g.size) - g.size;
        first = (unsigned int)__shfl((int)first, g.leader, 64);
        if (valid) out[off[key] + first + g.rank] = vals ? vals[slot] : slot; // the entry's payload, or its index
    }
}

// ---------------------------------------------------------------------------
// traversal: every ray of a list against a range of triangles.
//
//   K       rays per lane (256 K rays per workgroup)
//   FORM    0: filter for arbitrary rays; 1: every ray's line passes through p.origin
//           (rays fresh from raygen: the camera; shadow rays: the light)
//   ANYHIT  false: closest hit -- the reference's strict (t > 0 && t < best) in triangle
//                  order per lane; lanes/workgroups that scan different triangle ranges
//                  merge with a 64-bit atomicMin on (t bits, index), which keeps the
//                  smallest t and, among equal t, the lowest index: exactly what the
//                  sequential loop trace.glsl:135-157 keeps.
//           true : shadow query -- "is any primitive accepted with 0 < t <= tmax", which is
//                  all raycol_cs.glsl:50 reads of the shadow trace; sets occ[slot].
//   (the all-pairs mode, RT_TRACE_EXACT, has a kernel of its own: traverse_exact_kernel)
//
// grid = (ray blocks, triangle sub-ranges): blockIdx.y scans `tiles_per_y` LDS tiles
// starting at tri_begin.  Filter records stream through LDS in tiles of 256 triangles
// and are read back with wave-uniform addresses (LDS broadcast); the pairs the filter
// cannot reject go through the reference's own Moller-Trumbore arithmetic.
// ---------------------------------------------------------------------------
struct TraverseParams {
    Job job;
    const int *alist;  // slots to trace, or nullptr for 0..n-1
    const int *count;  // number of entries (device), or nullptr -> n
    int n;
    const rtf::TriRec *recs;
    const rtf::ConeRec *cones; // FORM 1 only
    const rtf::ConeRec *ccones; // FORM 1: one cone per run of RUN slots, or nullptr (no skipping)
    const rtf::ConeRec *scones; // FORM 1: one cone per SUPER_RUN slots (with ccones)
    const int *tri_of;          // FORM 1: triangle in each slot (-1: none), or nullptr (slot = triangle)
    const rt_vertex *verts;
    int num_tris;
    int tri_begin, tri_end; // multiple of TILE / end of range
    int tiles_per_y;
    float center[3];
    float radius;
    float origin[3];
    float reach_max;   // FORM 1: the cones are valid for rays whose reach is <= this
    float dir_sign;    // FORM 1: +1 the triangles lie along +dir from the origin (camera rays),
                       //         -1 along -dir (shadow rays run towards the light)
    unsigned long long *best;
    int *occ;
    unsigned long long *counters;
    // direction cells of the shared origin (traverse_grid_kernel) only
    const unsigned int *grid_off; // first list entry of each cell; [number of cells] = total
    const int *grid_entries;      // the lists: slots
    int grid_bits;                // 2^bits cells per axis
    int num_slots;
    // direction cells of the band sweep (traverse_bandcell_kernel): lists of member records
    const unsigned int *band_off;
    // ... and its band_g x band_g origin buckets (rt_grid.h): list cell * (band_g^2 + 1) + bucket
    int band_g;
    float band_cx, band_cy, band_cz;
    const rtg::BucketFrame *bucket_frames;
    const int *band_entries;
    int rays_per_wave;            // direction-cell kernels: 64, or fewer for small launches (see their comments)
    // band sweeps of arbitrary rays (traverse_group_kernel, traverse_band_kernel, traverse_bandcell_kernel)
    const float4 *ngroups;      // one (axis, w) per group of triangles with similar normals
    const float4 *nsubs;        // one (axis, w) per SUBGROUP members of a group (traverse_band_kernel)
    const float4 *nrecs;        // the members, two float4 each: (n/theta, kappa/H), (n/H, triangle index);
                                // inside the member's band and slab <=> max(|n1.b|, |n2.o - kappa'|) < 1
    int num_ngroups;
    const rtf::MomRec *mrecs;   // per-pair mode, arbitrary rays: moment-form records in triangle order
    // cells in space (traverse_voxel_kernel): per cell the triangles that come near it
    rtv::Grid vox;
    const unsigned int *vox_off;
    const int *vox_entries;
    // traverse_grid_kernel<false> over rays fresh from raygen (fresh_perm != nullptr): slot s is ray
    // fresh_perm[s], read from the ray planes and set up in registers (rayintersect_cs.glsl:12 and what
    // prepare_shared() forms) -- no set-up kernel, no job planes, no merge of the slots' best hits
    // ... and the kernel finishes them too (intersect_finish_slot: spheres, normal, the new ray, depth)
    // fresh_raygen: the planes of those rays have not been written -- rt_render skips the raygen launch -- and the
    // kernel forms ray fresh_perm[s] from its pixel as raygen_kernel would (camera, frame size, row map)
    const int *fresh_perm;
    int fresh_raygen;
    int fresh_width, fresh_height;
    const int *fresh_row_map;
    Camera fresh_cam;
    RaySoA fresh_rays;
    float *fresh_depth;
    int *fresh_zero; // a counter to zero for the next stage call (intersect_finish_kernel's zero_this)
    // walks that traverse_voxel_kernel hands over to traverse_voxel_long_kernel: (slot, parameter), their number
    int *walk_slot;
    float *walk_t;
    int *walk_count;
    int *walk_next;  // traverse_voxel_kernel: the next slot no wavefront has taken yet (zero when the launch starts)
    int walk_budget; // loop steps of a wavefront before it hands over the walks still under way (0: from the list's length)
};

// cooperative copy of one tile of the per-pair filter's 64-byte moment records into LDS; records
// past the end of the mesh are NaN (no ray passes them)
__device__ __forceinline__ void load_tile_moment(float4 *tile4, const rtf::MomRec *__restrict__ recs, int tile_base,
                                                 int num_tris, int tid)
{
    const float4 *src = reinterpret_cast<const float4 *>(recs) + (size_t)tile_base * 4;
    const int avail = (num_tris - tile_base) * 4;
    const float qn = __int_as_float(0x7fc00000);
#pragma unroll
    for (int i = 0; i < (TILE * 4) / BLOCK; i++) {
        const int e = i * BLOCK + tid;
        tile4[e] = (e < avail) ? src[e] : make_float4(qn, qn, qn, qn);
    }
}

// the reference's test on one pair, and what it does to the lane's running result
template <bool ANYHIT>
__device__ __forceinline__ void exact_pair(v3 o, v3 d, v3 X, v3 Y, v3 Z, int tri, float &tbest, int &idbest,
                                           bool &occ)
{
    float t;
    if (rtx::ray_vs_triangle(o, d, X, Y, Z, t)) {
        if (ANYHIT) {
            if (t > 0.0f && t <= tbest) occ = true;
        } else if (t > 0.0f && t < tbest) {
            tbest = t;
            idbest = tri;
        }
    }
}

constexpr int CONE_TILE = 1024; // triangles per LDS tile of cone records (16 KiB)
// candidate queue entries per wavefront: one triangle can add 64 K entries, and draining
// starts once fewer than that are free
__host__ __device__ constexpr int qcap(int K) { return 64 * K + 256; }

// value of `v` in lane `src` (wave-uniform lane index): one v_readlane_b32
__device__ __forceinline__ float readlane_f(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}


// The shared-origin form is latency-bound in its run and queue handling: six wavefronts per SIMD
// (80 VGPRs, a handful of spills) beat the four the register allocator would settle for.
template <int K, int FORM, bool ANYHIT>
__global__ __attribute__((amdgpu_waves_per_eu(FORM == 1 ? 6 : 1, FORM == 1 ? 6 : 8)))
__launch_bounds__(BLOCK) void traverse_kernel(TraverseParams p)
{
    // FORM 0: tiles of 256 moment records (16 KiB).  FORM 1: tiles of 1024 cone records (16 KiB); the
    // edge records are only touched by the rare second level and stay in global memory.
    __shared__ float4 lds4[FORM == 1 ? CONE_TILE : TILE * 4];
    __shared__ float4 run4[FORM == 1 ? CONE_TILE / RUN : 1];       // FORM 1: the cones of the tile's runs
    __shared__ float4 sup4[FORM == 1 ? CONE_TILE / SUPER_RUN : 1]; // ... and of the level above
    constexpr int TSZ = FORM == 1 ? CONE_TILE : TILE;

    const int tid = threadIdx.x;
    const int n = p.count ? *p.count : p.n;
    const int s0 = blockIdx.x * (BLOCK * K);
    if (s0 >= n) return; // whole workgroup leaves together
    // p.tiles_per_y counts tiles of 256 triangles; ranges start on multiples of 1024
    const int t_begin = p.tri_begin + blockIdx.y * p.tiles_per_y * TILE;
    const int t_end = min(t_begin + p.tiles_per_y * TILE, p.tri_end);
    if (t_begin >= t_end) return;

    int slot[K];
    bool valid[K];
    // FORM 0: a = the ray's moment about the mesh centre (rt_filter.h, ray_moment), b = direction * g,
    // d = direction.
    // FORM 1: b = unit direction (times dir_sign); a and d are unused.
    float ax[K], ay[K], az[K], bx[K], by[K], bz[K], dx[K], dy[K], dz[K], g[K], tbest[K];
    int idbest[K];
    bool occ[K], wild[K];
    const float qn = __int_as_float(0x7fc00000);

#pragma unroll
    for (int k = 0; k < K; k++) {
        // a wavefront takes 64 K consecutive list entries: K runs of 64 = K neighbouring pixel tiles
        const int s = s0 + ((tid >> 6) * K + k) * 64 + (tid & 63);
        valid[k] = s < n;
        slot[k] = valid[k] ? (p.alist ? p.alist[s] : s) : 0;
        v3 o = rtx::V(qn, qn, qn), d = o;
        float tmax = 0.0f;
        if (valid[k]) {
            o = rtx::V(p.job.ox[slot[k]], p.job.oy[slot[k]], p.job.oz[slot[k]]);
            d = rtx::V(p.job.dx[slot[k]], p.job.dy[slot[k]], p.job.dz[slot[k]]);
            if (ANYHIT) {
                tmax = p.job.tmax[slot[k]];
                if (p.occ[slot[k]] != 0) valid[k] = false; // already decided (tmax not finite)
            }
        }
        if (!valid[k]) o = d = rtx::V(qn, qn, qn);
        wild[k] = false;
        if (FORM == 0) {
            v3 oc = rtx::V(o.x - p.center[0], o.y - p.center[1], o.z - p.center[2]);
            const float reach = sqrtf(rtx::dot(oc, oc)) + 3.0f * p.radius;
            g[k] = rtf::dir_scale(reach, d.x, d.y, d.z);
            dx[k] = d.x; dy[k] = d.y; dz[k] = d.z;
            bx[k] = d.x * g[k]; by[k] = d.y * g[k]; bz[k] = d.z * g[k];
            rtf::ray_moment(o.x, o.y, o.z, p.center, bx[k], by[k], bz[k], ax[k], ay[k], az[k]);
        } else {
            bx[k] = by[k] = bz[k] = qn;
            g[k] = 1.0f;
            if (valid[k]) {
                bx[k] = p.job.bx[slot[k]]; by[k] = p.job.by[slot[k]]; bz[k] = p.job.bz[slot[k]];
                g[k] = p.job.g[slot[k]];
            }
            ax[k] = ay[k] = az[k] = dx[k] = dy[k] = dz[k] = 0.0f;
            // outside the range the cones were built for (or not comparable): the cone test is
            // skipped for the whole wave and the edge filter decides
            wild[k] = valid[k] && !(g[k] > 0.0f);
        }
        tbest[k] = ANYHIT ? tmax : INFINITY;
        idbest[k] = 0;
        occ[k] = !valid[k];
    }
    bool any_wild = false;
#pragma unroll
    for (int k = 0; k < K; k++) any_wild = any_wild || wild[k];
    const bool wave_wild = any64(any_wild);

    unsigned long long n_cand = 0, n_slow = 0;
    int tris_done = 0;
#ifdef RT_DBG_TIMING
    unsigned long long tm_tile = 0, tm_drain = 0, tm_slow = 0;
    const unsigned long long tm_start = __builtin_amdgcn_s_memtime();
#endif

    // the reference arithmetic for one triangle (vertices X, Y, Z); `pass` says which of the lane's rays take it
    auto exact_stage_xyz = [&](int tri, v3 X, v3 Y, v3 Z, const bool (&pass)[K]) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            n_cand += __popcll(__ballot(pass[k]));
            if (pass[k] && !(ANYHIT && occ[k])) {
                v3 o, d;
                if (FORM == 0) {
                    o = rtx::V(p.job.ox[slot[k]], p.job.oy[slot[k]], p.job.oz[slot[k]]);
                    d = rtx::V(dx[k], dy[k], dz[k]);
                } else {
                    o = rtx::V(p.job.ox[slot[k]], p.job.oy[slot[k]], p.job.oz[slot[k]]);
                    d = rtx::V(p.job.dx[slot[k]], p.job.dy[slot[k]], p.job.dz[slot[k]]);
                }
                exact_pair<ANYHIT>(o, d, X, Y, Z, tri, tbest[k], idbest[k], occ[k]);
            }
        }
    };
    // ... with the vertices read from the mesh (the rare second level of the filtered forms)
    auto exact_stage = [&](int tri, const bool (&pass)[K]) {
        exact_stage_xyz(tri, vpos(p.verts, 3 * tri), vpos(p.verts, 3 * tri + 1), vpos(p.verts, 3 * tri + 2), pass);
    };

    // FORM 1: queue of (triangle, ray) pairs that passed the cone test, one per wavefront
    __shared__ unsigned int queue_all[FORM == 1 ? (BLOCK / 64) * qcap(K) : 1];
    unsigned int *queue = queue_all + (FORM == 1 ? (tid >> 6) * qcap(K) : 0);
    int qcount = 0; // wave-uniform
    const int lane = tid & 63;

    // Level 2 and the reference arithmetic for queued pairs, one pair per lane.
    auto drain_body = [&](bool all) {
#ifdef RT_DBG_NO_DRAIN
        qcount = 0;
        return;
#endif
#ifdef RT_DBG_NO_DRAIN_ANYHIT
        if (ANYHIT) {
            n_cand += qcount;
            qcount = 0;
            return;
        }
#endif
        while (qcount >= 64 || (all && qcount > 0)) {
            const int take = qcount >= 64 ? 64 : qcount;
            qcount -= take;
            bool pass2 = false;
            int tri = 0, sl = 0;
            v3 o = rtx::V(0.0f, 0.0f, 0.0f), d = o;
            if (lane < take) {
                const unsigned int e = queue[qcount + lane];
                const int r = (int)(e & 2047u);
                const int ci = t_begin + (int)(e >> 11); // slot
                const int s = s0 + r;
                sl = p.alist ? p.alist[s] : s;
                // the edge filter on the unit direction, scaled (two-sided: the sign is irrelevant)
                const float gg = fabsf(p.job.g[sl]);
                float lo, hi;
                rtf::filter_origin_lohi(reinterpret_cast<const float *>(p.recs + ci), p.job.bx[sl] * gg,
                                        p.job.by[sl] * gg, p.job.bz[sl] * gg, lo, hi);
                pass2 = fmaxf(lo, -hi) >= -1.0f;
                tri = p.tri_of ? p.tri_of[ci] : ci;
                pass2 = pass2 && tri >= 0;
            }
            n_cand += __popcll(__builtin_amdgcn_ballot_w64(pass2));
            if (pass2) {
                o = rtx::V(p.job.ox[sl], p.job.oy[sl], p.job.oz[sl]);
                d = rtx::V(p.job.dx[sl], p.job.dy[sl], p.job.dz[sl]);
                const v3 X = vpos(p.verts, 3 * tri), Y = vpos(p.verts, 3 * tri + 1), Z = vpos(p.verts, 3 * tri + 2);
                float t;
                if (rtx::ray_vs_triangle(o, d, X, Y, Z, t)) {
                    if (ANYHIT) {
                        if (t > 0.0f && t <= p.job.tmax[sl]) p.occ[sl] = 1;
                    } else if (t > 0.0f) {
                        const unsigned long long key =
                            ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(unsigned int)tri;
                        atomicMin(&p.best[sl], key);
                    }
                }
            }
        }
    };

    auto drain = [&](bool all) {
#ifdef RT_DBG_TIMING
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        drain_body(all);
        tm_drain += __builtin_amdgcn_s_memtime() - t0;
#else
        drain_body(all);
#endif
    };

    for (int tile_base = t_begin; tile_base < t_end; tile_base += TSZ) {
#ifdef RT_DBG_TIMING
        const unsigned long long tt0 = __builtin_amdgcn_s_memtime();
#endif
        bool wave_decided = false;
        if (ANYHIT) {
            bool all_occ = true;
#pragma unroll
            for (int k = 0; k < K; k++) all_occ = all_occ && occ[k];
            // barrier + early exit: leave when every ray of the workgroup is occluded
            if (__syncthreads_and(all_occ ? 1 : 0)) break;
            wave_decided = !any64(!all_occ);
        } else {
            __syncthreads();
        }
        const int tile_n = min(TSZ, t_end - tile_base); // triangles of this tile that belong to the range
        if (FORM == 1) {
            const float4 *cs = reinterpret_cast<const float4 *>(p.cones);
#pragma unroll
            for (int i = 0; i < CONE_TILE / BLOCK; i++) {
                const int e = i * BLOCK + tid;
                const bool have = p.tri_of ? (e < tile_n) : (e < tile_n && tile_base + e < p.num_tris);
                lds4[e] = have ? cs[tile_base + e] : make_float4(0.0f, 0.0f, 0.0f, 2.0f);
            }
            if (tid < CONE_TILE / RUN) {
                const float4 *rc = reinterpret_cast<const float4 *>(p.ccones);
                run4[tid] = (rc && tid * RUN < tile_n) ? rc[tile_base / RUN + tid] : make_float4(0.0f, 0.0f, 0.0f, -2.0f);
            } else if (tid < CONE_TILE / RUN + CONE_TILE / SUPER_RUN) {
                const int q = tid - CONE_TILE / RUN;
                const float4 *sc = reinterpret_cast<const float4 *>(p.scones);
                sup4[q] = (p.ccones && sc && q * SUPER_RUN < tile_n) ? sc[tile_base / SUPER_RUN + q]
                                                                     : make_float4(0.0f, 0.0f, 0.0f, -2.0f);
            }
        } else {
            load_tile_moment(lds4, p.mrecs, tile_base, p.num_tris, tid);
        }
        __syncthreads();
        tris_done += tile_n;
#ifdef RT_DBG_TIMING
        tm_tile += __builtin_amdgcn_s_memtime() - tt0;
#endif
        // a wavefront whose rays are all occluded only keeps the barriers company
        if (ANYHIT && wave_decided) continue;

        if (FORM == 0) {
            const float *tile = reinterpret_cast<const float *>(lds4);
#pragma unroll 2
            for (int j = 0; j < TILE; j++) {
                const float *rec = tile + j * 16; // wave-uniform address: LDS broadcast
                float lo[K], hi[K];
#pragma unroll
                for (int k = 0; k < K; k++)
                    rtf::filter_moment_lohi(rec, ax[k], ay[k], az[k], bx[k], by[k], bz[k], lo[k], hi[k]);
                // some ray passes  <=>  max over rays of max(lo, -hi) >= -1
                float G = fmaxf(lo[0], -hi[0]);
#pragma unroll
                for (int k = 1; k < K; k++) G = fmaxf(fmaxf(G, lo[k]), -hi[k]);
                if (any64(G >= -1.0f)) {
                    n_slow++;
                    bool pass[K];
#pragma unroll
                    for (int k = 0; k < K; k++) pass[k] = fmaxf(lo[k], -hi[k]) >= -1.0f;
                    exact_stage(tile_base + j, pass);
                }
            }
        } else {
            // level 1: one dot product per pair against the triangle's cone of directions,
            // four triangles per step so that their LDS reads are in flight together.  Pairs
            // inside the cone are queued; full batches of 64 are handled with all lanes busy.
            for (int j = 0; j < CONE_TILE; j += 4) {
                if ((j & (SUPER_RUN - 1)) == 0) {
                    // SUPER_RUN slots (neighbouring clusters) none of whose cones can hold a ray of this
                    // wavefront are skipped ...
                    const float4 rc = sup4[j / SUPER_RUN];
                    float G = fmaf(rc.z, bz[0], fmaf(rc.y, by[0], rc.x * bx[0]));
#pragma unroll
                    for (int k = 1; k < K; k++) G = fmaxf(G, fmaf(rc.z, bz[k], fmaf(rc.y, by[k], rc.x * bx[k])));
                    if (!(any64(G >= rc.w) || wave_wild)) {
                        j += SUPER_RUN - 4;
                        continue;
                    }
                }
                if ((j & (RUN - 1)) == 0) {
                    // ... and so is a run of RUN
                    const float4 rc = run4[j / RUN];
                    float G = fmaf(rc.z, bz[0], fmaf(rc.y, by[0], rc.x * bx[0]));
#pragma unroll
                    for (int k = 1; k < K; k++) G = fmaxf(G, fmaf(rc.z, bz[k], fmaf(rc.y, by[k], rc.x * bx[k])));
                    if (!(any64(G >= rc.w) || wave_wild)) {
                        j += RUN - 4;
                        continue;
                    }
#ifdef RT_DBG_COUNT_RUNS
                    n_slow += 1000000;
#endif
                }
                const float4 c0 = lds4[j], c1 = lds4[j + 1], c2 = lds4[j + 2], c3 = lds4[j + 3];
                float t0[K], t1[K], t2[K], t3[K];
#pragma unroll
                for (int k = 0; k < K; k++) {
                    t0[k] = fmaf(c0.z, bz[k], fmaf(c0.y, by[k], c0.x * bx[k]));
                    t1[k] = fmaf(c1.z, bz[k], fmaf(c1.y, by[k], c1.x * bx[k]));
                    t2[k] = fmaf(c2.z, bz[k], fmaf(c2.y, by[k], c2.x * bx[k]));
                    t3[k] = fmaf(c3.z, bz[k], fmaf(c3.y, by[k], c3.x * bx[k]));
                }
                float G0 = t0[0], G1 = t1[0], G2 = t2[0], G3 = t3[0];
#pragma unroll
                for (int k = 1; k < K; k++) {
                    G0 = fmaxf(G0, t0[k]); G1 = fmaxf(G1, t1[k]);
                    G2 = fmaxf(G2, t2[k]); G3 = fmaxf(G3, t3[k]);
                }
                const bool h0 = G0 >= c0.w, h1 = G1 >= c1.w, h2 = G2 >= c2.w, h3 = G3 >= c3.w;
#ifndef RT_DBG_NO_SLOW
                if (any64(h0 || h1 || h2 || h3) || wave_wild) {
                    n_slow++;
#ifdef RT_DBG_TIMING
                    const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const float w = (u == 0) ? c0.w : (u == 1) ? c1.w : (u == 2) ? c2.w : c3.w;
                        const bool hu = (u == 0) ? h0 : (u == 1) ? h1 : (u == 2) ? h2 : h3;
                        if (!(any64(hu) || wave_wild)) continue;
                        const unsigned int trel = (unsigned int)(tile_base + j + u - t_begin);
                        // the triangle's edge record, read once for the wavefront: rays inside the cone
                        // are queued only if they also pass the edge filter
                        const float4 *er4 = reinterpret_cast<const float4 *>(p.recs + (tile_base + j + u));
                        const float4 e0 = er4[0], e1 = er4[1], e2 = er4[2];
                        const float er[12] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, e2.x, e2.y, e2.z, e2.w};
#pragma unroll
                        for (int k = 0; k < K; k++) {
                            const float t = (u == 0) ? t0[k] : (u == 1) ? t1[k] : (u == 2) ? t2[k] : t3[k];
                            bool in = ((t >= w) || wild[k]) && (w < 2.0f) && !occ[k];
                            if (any64(in)) {
                                const float gg = fabsf(g[k]);
                                float lo, hi;
                                rtf::filter_origin_lohi(er, bx[k] * gg, by[k] * gg, bz[k] * gg, lo, hi);
                                in = in && (fmaxf(lo, -hi) >= -1.0f);
                            }
                            const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
                            if (m) {
                                if (in)
                                    queue[qcount + lanes_below(m)] =
                                        (trel << 11) | (unsigned int)(((tid >> 6) * K + k) * 64 + (tid & 63));
                                qcount += __popcll(m);
                            }
                        }
                        if (qcount >= qcap(K) - 64 * K) drain(false);
                    }
#ifdef RT_DBG_TIMING
                    tm_slow += __builtin_amdgcn_s_memtime() - ts0;
#endif
                }
#else
                if (any64(h0 || h1 || h2 || h3) || wave_wild) n_slow++;
#endif
            }
            if (ANYHIT) {
                // results of drained batches live in global memory: refresh this lane's view
#pragma unroll
                for (int k = 0; k < K; k++) {
                    if (valid[k] && !occ[k]) occ[k] = p.occ[slot[k]] != 0;
                    // a decided ray no longer takes part: with a NaN direction it is in no cone
                    if (occ[k]) bx[k] = qn;
                }
            }
        }
    }
    if (FORM == 1) drain(true);

#ifdef RT_DBG_TIMING
    if ((tid & 63) == 0 && FORM == 1) {
        // per-wavefront cycles: [4] tile loads + barriers, [5] longest wavefront, [6] slow path, [7] total
        add_counter(p.counters, 4, tm_tile);
        (void)tm_drain;
        atomicMax(p.counters + 5, __builtin_amdgcn_s_memtime() - tm_start); // longest wavefront
        add_counter(p.counters, 6, tm_slow);
        add_counter(p.counters, 7, __builtin_amdgcn_s_memtime() - tm_start);
    }
#else
    if ((tid & 63) == 0) {
        add_counter(p.counters, ANYHIT ? C_SH_CAND : C_CH_CAND, n_cand);
        add_counter(p.counters, ANYHIT ? C_SH_SLOW : C_CH_SLOW, n_slow);
    }
#endif
    if (tid == 0) {
        const int mine = min(BLOCK * K, n - s0);
        add_counter(p.counters, ANYHIT ? C_SH_PAIRS : C_CH_PAIRS,
                    (unsigned long long)mine * (unsigned long long)max(min(tris_done, p.num_tris - t_begin), 0));
    }

#pragma unroll
    for (int k = 0; k < K; k++) {
        if (!valid[k]) continue;
        if (ANYHIT) {
            if (occ[k]) p.occ[slot[k]] = 1; // every writer writes 1
        } else if (tbest[k] < INFINITY) {
            const unsigned long long key =
                ((unsigned long long)__float_as_uint(tbest[k]) << 32) | (unsigned long long)(unsigned int)idbest[k];
            atomicMin(&p.best[slot[k]], key);
        }
    }
}


// ---------------------------------------------------------------------------
// RT_TRACE_EXACT: the loop of "OpenGL Raytracer/trace.glsl":135-157 as written -- every ray against
// every triangle, no acceleration structure, no filter record.
//
// A tile of 256 triangles is staged in LDS as (X, e1 = Y - X, e2 = Z - X): the two edge
// subtractions of trace.glsl:5-6 depend on the triangle alone, so the thread that copies a triangle
// does them once for the workgroup (the same bits as doing them per pair).  Every wavefront then reads
// the tile with wave-uniform addresses (LDS broadcast) and evaluates, for K rays per lane and in the
// reference's own association and rounding,
//     q = cross(d, e2);  a = dot(e1, q);  s = o - X;  x = dot(s, q)            (trace.glsl:7-12)
// i.e. everything the shader computes up to its first two `return false` (|a| < 1e-20; u = (1/a) x
// outside [0, 1]).  22 multiplies/adds per pair, nothing fused; a lane holds its rays in pairs, so each is
// one v_pk_mul_f32 / v_pk_add_f32 for two pairs, the triangle's value broadcast to both halves (packed
// and plain forms round each component alike).  The shader's divide is NOT executed on every pair: from
// the exact a and x,
//     x (x - a)  >  2^-20 a^2 + 2^-100        (four more operations)
// proves that the shader rejects the pair: with rho = x / a the left side is a^2 rho (rho - 1), so
// rho > 1 + 2^-22 - 2^-44 or rho < -(2^-22 - 2^-44) even after the three roundings of the test itself
// (each within 2^-24, underflow within 2^-150), and u = RN(RN(1/a) x) = rho (1 + 2 roundings) is then
// >= 1 + 2^-23 or <= -2^-23: `u < 0 || u > 1` holds (|a| < 2^64 because a^2 is finite, so 1/a is a
// normal number; a = 0 or |a| < 1e-20 is rejected by the shader's first test whatever x is).  A NaN or
// an overflow on either side makes the comparison false.  Pairs the comparison does not settle -- about
// one in the ratio of a triangle's width to the scene's -- are queued per wavefront in LDS and handled 64
// at a time, one pair per lane, by the reference's whole function (rtx::ray_vs_triangle: IEEE divide,
// all four tests); its hits merge by a 64-bit atomicMin on (t bits, index) -- the smallest t and among
// equal t the lowest index, what the sequential loop keeps -- or set the shadow ray's flag.
// ---------------------------------------------------------------------------
__host__ __device__ constexpr int exact_qcap(int K) { return 64 * K + 320; }

typedef float f2 __attribute__((ext_vector_type(2)));

template <int K, bool ANYHIT>
__global__ __launch_bounds__(BLOCK) void traverse_exact_kernel(TraverseParams p)
{
    static_assert(K % 2 == 0, "rays are held in pairs");
    __shared__ float4 tile4[TILE * 3];
    __shared__ unsigned int queue_all[(BLOCK / 64) * exact_qcap(K)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int n = p.count ? *p.count : p.n;
    const int s0 = blockIdx.x * (BLOCK * K);
    if (s0 >= n) return; // whole workgroup leaves together
    const int t_begin = p.tri_begin + blockIdx.y * p.tiles_per_y * TILE;
    const int t_end = min(min(t_begin + p.tiles_per_y * TILE, p.tri_end), p.num_tris);
    if (t_begin >= t_end) return;

    int slot[K];
    bool valid[K];
    f2 ox[K / 2], oy[K / 2], oz[K / 2], dx[K / 2], dy[K / 2], dz[K / 2]; // ray k = component k % 2 of pair k / 2
    unsigned long long live[K]; // wave-uniform: the lanes whose ray k still takes part
    const float qn = __int_as_float(0x7fc00000);
#pragma unroll
    for (int k = 0; k < K; k++) {
        // a wavefront takes 64 K consecutive list entries: K runs of 64 = K neighbouring pixel tiles
        const int s = s0 + ((tid >> 6) * K + k) * 64 + lane;
        valid[k] = s < n;
        slot[k] = valid[k] ? (p.alist ? p.alist[s] : s) : 0;
        float o0 = qn, o1 = qn, o2 = qn, d0 = qn, d1 = qn, d2 = qn;
        if (valid[k]) {
            if (ANYHIT && p.occ[slot[k]] != 0) valid[k] = false; // already decided
            o0 = p.job.ox[slot[k]]; o1 = p.job.oy[slot[k]]; o2 = p.job.oz[slot[k]];
            d0 = p.job.dx[slot[k]]; d1 = p.job.dy[slot[k]]; d2 = p.job.dz[slot[k]];
        }
        ox[k / 2][k % 2] = o0; oy[k / 2][k % 2] = o1; oz[k / 2][k % 2] = o2;
        dx[k / 2][k % 2] = d0; dy[k / 2][k % 2] = d1; dz[k / 2][k % 2] = d2;
        live[k] = __builtin_amdgcn_ballot_w64(valid[k]);
    }

    unsigned int *queue = queue_all + (tid >> 6) * exact_qcap(K);
    int qcount = 0; // wave-uniform
    unsigned long long n_cand = 0;
    int tris_done = 0;

    // the reference's whole test for queued pairs, one pair per lane
    auto drain = [&](bool all) {
        while (qcount >= 64 || (all && qcount > 0)) {
            const int take = qcount >= 64 ? 64 : qcount;
            qcount -= take;
            if (lane < take) {
                const unsigned int e = queue[qcount + lane];
                const int s = s0 + (int)(e & 2047u);
                const int tri = t_begin + (int)(e >> 11);
                const int sl = p.alist ? p.alist[s] : s;
                const v3 o = rtx::V(p.job.ox[sl], p.job.oy[sl], p.job.oz[sl]);
                const v3 d = rtx::V(p.job.dx[sl], p.job.dy[sl], p.job.dz[sl]);
                const v3 X = vpos(p.verts, 3 * tri), Y = vpos(p.verts, 3 * tri + 1), Z = vpos(p.verts, 3 * tri + 2);
                float t;
                if (rtx::ray_vs_triangle(o, d, X, Y, Z, t)) {
                    if (ANYHIT) {
                        if (t > 0.0f && t <= p.job.tmax[sl]) p.occ[sl] = 1;
                    } else if (t > 0.0f) {
                        const unsigned long long key =
                            ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(unsigned int)tri;
                        atomicMin(&p.best[sl], key);
                    }
                }
            }
        }
    };

    for (int tile_base = t_begin; tile_base < t_end; tile_base += TILE) {
        if (ANYHIT) {
            // shadow rays decided by a drained batch leave the loop; the workgroup leaves when all have
            bool undecided = false;
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (valid[k] && tile_base != t_begin && p.occ[slot[k]] != 0) valid[k] = false;
                live[k] = __builtin_amdgcn_ballot_w64(valid[k]);
                undecided = undecided || valid[k];
            }
            if (__syncthreads_and(undecided ? 0 : 1)) break;
        } else {
            __syncthreads();
        }
        const int tile_tris = min(TILE, t_end - tile_base);
        {
            // one triangle per thread: X, e1, e2 (trace.glsl:5-6)
            float4 X = make_float4(qn, qn, qn, qn), E1 = X, E2 = X;
            if (tid < tile_tris) {
                const int tri = tile_base + tid;
                const v3 x = vpos(p.verts, 3 * tri), y = vpos(p.verts, 3 * tri + 1), z = vpos(p.verts, 3 * tri + 2);
                const v3 e1 = rtx::sub(y, x), e2 = rtx::sub(z, x);
                // nine values in 48 bytes, laid out so that two 16-byte reads and one 4-byte read bring every
                // value into a register that a packed instruction can broadcast from
                X = make_float4(x.x, x.y, x.z, e1.x);
                E1 = make_float4(e1.y, e1.z, e2.x, e2.y);
                E2 = make_float4(e2.z, 0.0f, 0.0f, 0.0f);
            }
            tile4[3 * tid] = X; tile4[3 * tid + 1] = E1; tile4[3 * tid + 2] = E2;
        }
        __syncthreads();
        tris_done += tile_tris;
        bool wave_live = false;
#pragma unroll
        for (int k = 0; k < K; k++) wave_live = wave_live || live[k] != 0ull;
        if (!wave_live) continue; // a wavefront without rays only keeps the barriers company

#pragma unroll 2
        for (int j = 0; j < tile_tris; j++) {
            const float4 T0 = tile4[3 * j], T1 = tile4[3 * j + 1];
            const float T2 = reinterpret_cast<const float *>(tile4)[12 * j + 8];
            const v3 X = rtx::V(T0.x, T0.y, T0.z), E1 = rtx::V(T0.w, T1.x, T1.y), E2 = rtx::V(T1.z, T1.w, T2);
            const unsigned int trel = (unsigned int)(tile_base + j - t_begin) << 11;
#pragma unroll
            for (int g = 0; g < K / 2; g++) {
                const f2 qx = dy[g] * E2.z - E2.y * dz[g];
                const f2 qy = dz[g] * E2.x - E2.z * dx[g];
                const f2 qz = dx[g] * E2.y - E2.x * dy[g];
                const f2 a = E1.x * qx + E1.y * qy + E1.z * qz;
                const f2 sx = ox[g] - X.x, sy = oy[g] - X.y, sz = oz[g] - X.z;
                const f2 x = sx * qx + sy * qy + sz * qz;
                const f2 z = x * (x - a);
                const f2 thr = __builtin_elementwise_fma(a * a, f2{0x1p-20f, 0x1p-20f}, f2{0x1p-100f, 0x1p-100f});
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int k = 2 * g + h;
                    const bool open = !(z[h] > thr[h]); // not settled (or a lane without a ray: NaN)
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(open) & live[k];
                    if (m) {
                        if (open && valid[k])
                            queue[qcount + lanes_below(m)] = trel | (unsigned int)(((tid >> 6) * K + k) * 64 + lane);
                        qcount += __popcll(m);
                    }
                }
            }
            if (qcount >= exact_qcap(K) - 64 * K) {
                n_cand += (unsigned long long)(qcount & ~63);
                drain(false);
            }
        }
    }
    n_cand += (unsigned long long)qcount;
    drain(true);

    if (lane == 0) add_counter(p.counters, ANYHIT ? C_SH_CAND : C_CH_CAND, n_cand);
    if (tid == 0) {
        const int mine = min(BLOCK * K, n - s0);
        add_counter(p.counters, ANYHIT ? C_SH_PAIRS : C_CH_PAIRS, (unsigned long long)mine * (unsigned long long)tris_done);
    }
}


// ---------------------------------------------------------------------------
// Direction cells (rt_grid.h): the list of every cell of a shared origin.
// One thread per slot walks the cells from the coarsest level down (a cell is the union of its four
// children) and enters the slot in every finest cell that a ray could get it out of: the slot's cone
// and its three edge planes against the cell's bounding cone.  A slot no test can bound goes on the
// list "everywhere" (index num_cells), which every ray sweeps.
// FILL = false counts into count[]; FILL = true writes the slot at off[cell] + cursor[cell]++.
// The order of a list is whatever order the threads arrive in; results do not depend on it
// (closest hits merge by a minimum, shadow rays by an or).
// ---------------------------------------------------------------------------
// what the list builder needs to know about each slot, once per slot and origin
__global__ __launch_bounds__(BLOCK) void slot_view_kernel(const float4 *__restrict__ cones, int num_slots,
                                                          const rtf::TriRec *__restrict__ recs,
                                                          const int *__restrict__ tri_of,
                                                          const rt_vertex *__restrict__ verts, float ox, float oy,
                                                          float oz, double extent, double reach_max,
                                                          rtg::SlotView *__restrict__ views)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= num_slots) return;
    const int tri = tri_of[i];
    const float4 cone = cones[i];
    rtg::SlotView view;
    view.cone = 0;
    view.planes = false;
    if (tri >= 0 && cone.w < 2.0f) {
        const float O[3] = {ox, oy, oz};
        const float cn[4] = {cone.x, cone.y, cone.z, cone.w};
        const rtf::TriRec er = recs[i];
        rtg::make_slot_view(cn, verts[3 * tri].pos, verts[3 * tri + 1].pos, verts[3 * tri + 2].pos, O, &er, extent,
                            reach_max, &view);
    }
    views[i] = view;
}

constexpr int BUILD_TOP_LEVEL = 3; // the builders start one thread per (slot or member, cell of this level)
constexpr int BUILD_TOP_CELLS = 1 << (2 * BUILD_TOP_LEVEL);
constexpr unsigned int BUILD_EVERYWHERE = 0xffu;
constexpr int BUILD_STACK = 3 * (rtg::MAX_BITS - BUILD_TOP_LEVEL) + 8; // entries a walk can have pending

// The list build of one origin, one launch: the (cell, slot) pairs of every finest cell a slot must be
// listed in, appended to keys[] / vals[] in no particular order (they are sorted by cell afterwards;
// nothing depends on the order inside a cell).  A wavefront takes 64 slots and walks the cells below
// the whole sphere for all of them TOGETHER: the nodes still to expand -- (slot, cell of some level)
// -- sit on one stack in LDS, every step the 64 lanes take the 64 topmost nodes, each tests its
// node's four children against its slot's view (cell_meets_slot: a coarser cell is the union of the
// four below it, so a failed test drops everything under that child) and the survivors go back on the
// stack, or, at the finest level, into the wavefront's output buffer.  All lanes do the same work in
// every step whatever the shape of each slot's walk, which a walk per lane cannot offer.
//   Depth first, so the stack stays small; and it cannot overflow: a step that could (each node taken
// leaves up to three more behind than it removes) takes fewer nodes, down to one -- a walk of single
// nodes from a stack of S adds at most 3 per level below the node on top, under 30 in all, which is
// the room GRID_LIST_RESERVE keeps free.
//   A slot for the list "everywhere" is appended once, under cell num_cells.
//   *cursor counts every pair, stored or not.  Pairs beyond `cap` are not stored; their slots are marked in
// overflow[] (a bit per slot) and grid_overflow_kernel then lists each marked slot once under "everywhere",
// in the num_slots entries kept free behind `cap`: a launch that runs out of room loses speed, not pairs
// (a slot listed twice is tested twice; closest hits merge by a minimum, shadow rays by an or).  The
// host learns the count afterwards and gives the next build more room.
//   Where the data comes from decides the speed (the arithmetic is a tenth of the time): a lane's node is
// any of the 64 slots at any cell, so 64 lanes reading their own views and cones would touch 64 cache
// lines with every load instruction, twenty-one instructions a step.  Instead each lane keeps the view
// of ITS slot in registers and a node's lane fetches the view it needs from that lane (shuffles); the
// cones of the four children of a node are 192 contiguous bytes, and the wavefront loads those of its
// 64 nodes as 768 16-byte pieces, consecutive lanes taking consecutive pieces (twelve loads of about a
// dozen cache lines each), and hands them to the nodes' lanes through 3 KB of LDS, sixteen nodes at a
// time.  A wavefront needs 11 KB of LDS, so a SIMD holds three or four: the chain from a node to its
// four tests is long, and only other wavefronts can fill it.
constexpr int GRID_LIST_WAVES = 4;        // wavefronts per workgroup
constexpr int GRID_LIST_STACK = 1024;
constexpr int GRID_LIST_RESERVE = 40;
constexpr int GRID_LIST_OUT = 256;        // pairs a wavefront gathers before it appends them
constexpr int GRID_LIST_STAGE_PITCH = 13; // 16-byte pieces per node in the staging area: 12 + 1 against bank conflicts
//   Two phases, so that a few slots that cover thousands of cells -- large triangles next to the origin --
// do not leave the launch to the handful of wavefronts that hold them.  The walk of a HEAVY slot (its cone
// spans more than GRID_LIST_HEAVY_CELLS finest cells, or it has planes but no cone) stops three levels above
// the finest, and the nodes it has there, (cell of that level, slot), are appended to a list of ITEMS
// instead; a second launch (ITEMS = true) takes the items 64 to a wavefront -- a lane's "slot" is then its
// item's slot, its root the item's cell -- and walks the last three levels (64 finest cells per item).  The
// other slots of the same wavefront go all the way down where they are.  An item the list has no room for
// marks its slot in overflow[].
constexpr int GRID_LIST_ITEM_LEVELS = 3;
// (threshold 96 / 1024 / 4096 / 16384 cells: 64 views of 256K triangles 51.5 / 50.1 / 49.6 / 49.9 ms a batch, the
// 165-triangle sword with moving lights 0.385 / 0.392 / 0.396 / 0.402 ms a frame, 1.03 with cone-less slots alone)
constexpr double GRID_LIST_HEAVY_CELLS = 1024.0;
constexpr int GRID_LIST_ITEMS_OUT = 64;   // items a wavefront gathers before it appends them
struct GridItems {
    unsigned int *keys; // cell of level (bits - GRID_LIST_ITEM_LEVELS)
    int *vals;          // slot
    unsigned long long *cursor;
    unsigned long long cap;
    double heavy_cells;
};
template <bool ITEMS>
__global__ __launch_bounds__(GRID_LIST_WAVES * 64) void grid_list_kernel(
    const rtg::CellCone *__restrict__ cells, int bits, const rtg::SlotView *__restrict__ views, int num_slots,
    unsigned int *__restrict__ keys, int *__restrict__ vals, unsigned long long *__restrict__ cursor,
    unsigned long long cap, int *__restrict__ fault, unsigned int *__restrict__ overflow, GridItems items)
{
    static_assert(sizeof(rtg::CellCone) == 48, "four cones are twelve 16-byte pieces");
    __shared__ unsigned int stack_all[GRID_LIST_WAVES][GRID_LIST_STACK];
    __shared__ unsigned int okey_all[GRID_LIST_WAVES][GRID_LIST_OUT + 256];
    __shared__ int oval_all[GRID_LIST_WAVES][GRID_LIST_OUT + 256];
    __shared__ unsigned int ikey_all[GRID_LIST_WAVES][ITEMS ? 1 : GRID_LIST_ITEMS_OUT + 256];
    __shared__ int ival_all[GRID_LIST_WAVES][ITEMS ? 1 : GRID_LIST_ITEMS_OUT + 256];
    __shared__ double2 stage_all[GRID_LIST_WAVES][16 * GRID_LIST_STAGE_PITCH];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    // (ITEMS: the number of items is what the first launch counted, at most what there was room for; the launch
    // is sized by a guess of that number, so a wavefront goes on to further runs of 64 items while there are any)
    const int n_in = ITEMS ? (int)min(*items.cursor, items.cap) : num_slots;
    for (int first = (blockIdx.x * GRID_LIST_WAVES + wave) * 64; first < n_in; first += (int)gridDim.x * GRID_LIST_WAVES * 64) {
    unsigned int *stack = stack_all[wave]; // node: item | level << 6 | cell << 10
    unsigned int *okey = okey_all[wave];
    int *oval = oval_all[wave];
    unsigned int *ikey = ikey_all[wave];
    int *ival = ival_all[wave];
    double2 *stage = stage_all[wave];
    const unsigned int num_cells = 1u << (2 * bits);
    int sp = 0, n_out = 0, n_items_out = 0; // wave-uniform
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto append = [&](const unsigned int *bk, const int *bv, int &n, unsigned int *kout, int *vout,
                      unsigned long long *cur, unsigned long long room) {
        if (!n) return;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(cur, (unsigned long long)n);
        base = __shfl(base, 0, 64);
        for (int k = lane; k < n; k += 64)
            if (base + (unsigned long long)k < room) {
                kout[base + k] = bk[k];
                vout[base + k] = bv[k];
            } else {
                atomicOr(&overflow[bv[k] >> 5], 1u << (bv[k] & 31));
            }
        n = 0;
    };
    auto flush = [&]() { append(okey, oval, n_out, keys, vals, cursor, cap); };
    auto flush_items = [&]() {
        if (!ITEMS) append(ikey, ival, n_items_out, items.keys, items.vals, items.cursor, items.cap);
    };

    // this lane's slot: its view as 18 doubles (the first holds the two flags)
    static_assert(sizeof(rtg::SlotView) == 144 && offsetof(rtg::SlotView, planes) == 4 && offsetof(rtg::SlotView, a) == 8 &&
                      offsetof(rtg::SlotView, ct) == 32 && offsetof(rtg::SlotView, st) == 40 &&
                      offsetof(rtg::SlotView, n) == 48 && offsetof(rtg::SlotView, b) == 120,
                  "the view travels between lanes as 18 doubles");
    double md[18];
    for (int i = 0; i < 18; i++) md[i] = 0.0;
    // the slot this lane holds: its own (first + lane), or its item's
    int my_slot = first + lane;
    unsigned int my_cell = 0;
    if (ITEMS && first + lane < n_in) {
        my_slot = items.vals[first + lane];
        my_cell = items.keys[first + lane];
    }
    if (first + lane < n_in) {
        const double2 *vp = reinterpret_cast<const double2 *>(views + my_slot);
        for (int i = 0; i < 9; i++) {
            const double2 d = vp[i];
            md[2 * i] = d.x;
            md[2 * i + 1] = d.y;
        }
    }
    auto view_of = [](const double d[18]) {
        rtg::SlotView v;
        const unsigned long long f = (unsigned long long)__double_as_longlong(d[0]);
        v.cone = (int)(unsigned int)f;
        v.planes = ((f >> 32) & 0xffull) != 0;
        for (int i = 0; i < 3; i++) v.a[i] = d[1 + i];
        v.ct = d[4];
        v.st = d[5];
        for (int k = 0; k < 3; k++)
            for (int i = 0; i < 3; i++) v.n[k][i] = d[6 + 3 * k + i];
        for (int i = 0; i < 3; i++) v.b[i] = d[15 + i];
        return v;
    };
    const rtg::SlotView mine = view_of(md);
    bool my_heavy = false; // this lane's slot stops three levels up and goes on as items
    if (ITEMS) {
        // root: the item's cell, three levels above the finest
        const bool walk = first + lane < n_in;
        const unsigned long long mw = __builtin_amdgcn_ballot_w64(walk);
        if (walk) stack[lanes_below(mw)] = (unsigned int)lane | ((unsigned int)(bits - GRID_LIST_ITEM_LEVELS) << 6) | (my_cell << 10);
        sp = __popcll(mw);
    } else {
        const bool everywhere = mine.cone != 0 && rtg::slot_everywhere(mine);
        const bool walk = mine.cone != 0 && !everywhere;
        const unsigned long long me = __builtin_amdgcn_ballot_w64(everywhere);
        if (everywhere) {
            okey[lanes_below(me)] = num_cells;
            oval[lanes_below(me)] = first + lane;
        }
        n_out = __popcll(me);
        // a heavy slot: its cone spans many of the finest cells (solid angle 2 pi (1 - cos) of 4 pi, over 4^bits
        // cells), or it has no cone to say
        my_heavy = walk && bits - GRID_LIST_ITEM_LEVELS >= 3 && items.cap > 0 &&
                   (mine.cone == 2 || (1.0 - mine.ct) * 0.5 * (double)num_cells > items.heavy_cells);
        // roots: the four cells of level 1 (not tested themselves: the tables start at level 2)
        const unsigned long long mw = __builtin_amdgcn_ballot_w64(walk);
        if (walk)
            for (unsigned int k = 0; k < 4; k++) stack[4 * lanes_below(mw) + k] = (unsigned int)lane | (1u << 6) | (k << 10);
        sp = 4 * __popcll(mw);
    }
    wave_sync();
    // the pieces this lane moves in each group of three loads: piece q = 64 t + lane of a sixteen-node round
    int piece_node[3], piece_part[3];
    for (int t = 0; t < 3; t++) {
        const int q = 64 * t + lane;
        piece_node[t] = q / 12;
        piece_part[t] = q % 12;
    }

    while (sp > 0) {
        int room = (GRID_LIST_STACK - GRID_LIST_RESERVE - sp) / 3;
        if (room < 1) room = 1;
        int take = sp < 64 ? sp : 64;
        if (take > room) take = room;
        sp -= take;
        const bool has = lane < take;
        const unsigned int node = has ? stack[sp + lane] : 0u;
        const unsigned int item = node & 63u;
        const int level = (int)((node >> 6) & 15u) + 1; // of the children
        const unsigned int cell4 = (node >> 10) * 4u;
        // the cones of the four children, for all 64 nodes: loads first, all in flight together
        const unsigned int cone_at = has ? (unsigned int)rtg::level_offset(level) + cell4 : 0u;
        auto load_piece = [&](int r, int t) {
            const unsigned int src = (unsigned int)__shfl((int)cone_at, 16 * r + piece_node[t], 64);
            return reinterpret_cast<const double2 *>(cells + src)[piece_part[t]];
        };
        const double2 pc0 = load_piece(0, 0), pc1 = load_piece(0, 1), pc2 = load_piece(0, 2);
        const double2 pc3 = load_piece(1, 0), pc4 = load_piece(1, 1), pc5 = load_piece(1, 2);
        const double2 pc6 = load_piece(2, 0), pc7 = load_piece(2, 1), pc8 = load_piece(2, 2);
        const double2 pc9 = load_piece(3, 0), pc10 = load_piece(3, 1), pc11 = load_piece(3, 2);
        // the view of the node's slot, from the lane that holds it
        double vd[18];
        for (int i = 0; i < 18; i++) vd[i] = __shfl(md[i], (int)item, 64);
        const rtg::SlotView view = view_of(vd);
        // ... and the cones to their nodes' lanes, sixteen nodes a round
        double2 g0, g1, g2, g3, g4, g5, g6, g7, g8, g9, g10, g11;
        g0 = g1 = g2 = g3 = g4 = g5 = g6 = g7 = g8 = g9 = g10 = g11 = make_double2(0.0, 0.0);
        const int put0 = piece_node[0] * GRID_LIST_STAGE_PITCH + piece_part[0];
        const int put1 = piece_node[1] * GRID_LIST_STAGE_PITCH + piece_part[1];
        const int put2 = piece_node[2] * GRID_LIST_STAGE_PITCH + piece_part[2];
        const double2 *take_at = stage + (lane & 15) * GRID_LIST_STAGE_PITCH;
#define RT_GRID_LIST_ROUND(R, A, B, C)                                                            \
    stage[put0] = A; stage[put1] = B; stage[put2] = C;                                            \
    wave_sync();                                                                                  \
    if ((lane >> 4) == R) {                                                                       \
        g0 = take_at[0]; g1 = take_at[1]; g2 = take_at[2]; g3 = take_at[3];                       \
        g4 = take_at[4]; g5 = take_at[5]; g6 = take_at[6]; g7 = take_at[7];                       \
        g8 = take_at[8]; g9 = take_at[9]; g10 = take_at[10]; g11 = take_at[11];                   \
    }                                                                                             \
    wave_sync();
        RT_GRID_LIST_ROUND(0, pc0, pc1, pc2)
        RT_GRID_LIST_ROUND(1, pc3, pc4, pc5)
        RT_GRID_LIST_ROUND(2, pc6, pc7, pc8)
        RT_GRID_LIST_ROUND(3, pc9, pc10, pc11)
#undef RT_GRID_LIST_ROUND
        const double2 got[12] = {g0, g1, g2, g3, g4, g5, g6, g7, g8, g9, g10, g11};
        bool pass[4] = {false, false, false, false};
        if (has)
            for (int k = 0; k < 4; k++) {
                rtg::CellCone cc; // six doubles in three pieces
                cc.a[0] = got[3 * k].x;
                cc.a[1] = got[3 * k].y;
                cc.a[2] = got[3 * k + 1].x;
                cc.cpsi = got[3 * k + 1].y;
                cc.spsi = got[3 * k + 2].x;
                cc.psi = got[3 * k + 2].y;
                pass[k] = rtg::cell_meets_slot(cc, view);
            }
        // children of the finest level are pairs; those of a heavy slot three levels above it are items
        const bool heavy_node = !ITEMS && __shfl((int)my_heavy, (int)item, 64) != 0;
        const bool leaf = level == bits;
        const bool hand_on = heavy_node && level == bits - GRID_LIST_ITEM_LEVELS;
        const int slot_of_item = __shfl(my_slot, (int)item, 64);
        for (int k = 0; k < 4; k++) {
            const unsigned long long mp = __builtin_amdgcn_ballot_w64(pass[k] && !leaf && !hand_on);
            const unsigned long long mo = __builtin_amdgcn_ballot_w64(pass[k] && leaf);
            if (pass[k] && !leaf && !hand_on)
                stack[sp + lanes_below(mp)] = item | ((unsigned int)level << 6) | ((cell4 + (unsigned int)k) << 10);
            if (pass[k] && leaf) {
                okey[n_out + lanes_below(mo)] = cell4 + (unsigned int)k;
                oval[n_out + lanes_below(mo)] = slot_of_item;
            }
            sp += __popcll(mp);
            n_out += __popcll(mo);
            if (!ITEMS) {
                const unsigned long long mi = __builtin_amdgcn_ballot_w64(pass[k] && hand_on);
                if (pass[k] && hand_on) {
                    ikey[n_items_out + lanes_below(mi)] = cell4 + (unsigned int)k;
                    ival[n_items_out + lanes_below(mi)] = slot_of_item;
                }
                n_items_out += __popcll(mi);
            }
        }
        if (sp > GRID_LIST_STACK - 4) { // cannot happen (see above)
            if (lane == 0) *fault = 1;
            break;
        }
        wave_sync();
        if (n_out >= GRID_LIST_OUT) flush();
        if (n_items_out >= GRID_LIST_ITEMS_OUT) flush_items();
    }
    wave_sync();
    flush();
    flush_items();
    wave_sync();
    if (!ITEMS) break; // (slots: one run of 64 per wavefront)
    }
}

// After grid_list_kernel: the slots it had no room for (a pair beyond `cap`, or an item beyond the items' room)
// go on the list "everywhere" (key num_cells), once each, behind the pairs that were stored; counts[3] receives
// the number of pairs there are to sort.
// counts: [0] pairs grid_list_kernel counted, [2] marked slots filed so far, [3] the result (all zero but
// [0] when the launch starts).
__global__ __launch_bounds__(BLOCK) void grid_overflow_kernel(const unsigned int *__restrict__ overflow, int num_slots,
                                                              unsigned int num_cells, unsigned int *__restrict__ keys,
                                                              int *__restrict__ vals, unsigned long long *__restrict__ counts,
                                                              unsigned long long cap)
{
    const unsigned long long total = counts[0];
    const unsigned long long stored = total < cap ? total : cap;
    const int slot = blockIdx.x * BLOCK + threadIdx.x;
    if (slot == 0) atomicAdd(&counts[3], stored);
    if (slot >= num_slots) return;
    if ((overflow[slot >> 5] >> (slot & 31)) & 1u) {
        const unsigned long long pos = stored + atomicAdd(&counts[2], 1ull);
        keys[pos] = num_cells;
        vals[pos] = slot;
        atomicAdd(&counts[3], 1ull);
    }
}

// Step 3: off[c] = position of the first pair of cell c in the sorted keys (c = 0 .. num_cells + 1:
// the list "everywhere" is cell num_cells, and off[num_cells + 1] = total).
__global__ __launch_bounds__(BLOCK) void grid_offsets_kernel(const unsigned int *__restrict__ keys, unsigned int total,
                                                             unsigned int num_cells, unsigned int *__restrict__ off)
{
    const unsigned int c = blockIdx.x * BLOCK + threadIdx.x;
    if (c > num_cells + 1) return;
    unsigned int lo = 0, hi = total; // first position with keys[pos] >= c
    while (lo < hi) {
        const unsigned int mid = (lo + hi) >> 1;
        if (keys[mid] < c) lo = mid + 1;
        else hi = mid;
    }
    off[c] = lo;
}

// 64-bit sum of n 32-bit counts (the list builders size their 32-bit list positions with it): one
// wavefront reduction and one atomic per workgroup.  *out must be zero before the launch.
__global__ __launch_bounds__(BLOCK) void sum_counts_kernel(const unsigned int *__restrict__ v, size_t n,
                                                           unsigned long long *__restrict__ out)
{
    __shared__ unsigned long long part[BLOCK / 64];
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) s += v[i];
    for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_down(s, sh, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < BLOCK / 64; w++) t += part[w];
        if (t) atomicAdd(out, t);
    }
}

// ---------------------------------------------------------------------------
// Shared-origin traversal over direction cells.  One ray per lane; wavefronts are independent
// (no barrier).  The rays of a wavefront are taken cell by cell: the cell's list is read 64 slots
// at a time, ONE SLOT PER LANE (its edge record stays in the lane's registers), and every ray of
// the cell is broadcast in turn (v_readlane) and put through the edge filter against the 64
// records at once.  What a ray costs is therefore the length of its cell's list / 64, whatever the
// other rays of the wavefront look at.  Survivors are queued per wavefront in LDS and go through the
// reference arithmetic one pair per lane.
//   ANYHIT false: closest hit, merged per ray by a 64-bit minimum of (t bits, triangle) in LDS,
//                 then into p.best -- the same rule as traverse_kernel.
//   ANYHIT true : shadow query; sets p.occ.  Rays found occluded leave their cell's loop.
// Rays outside the range the cones were built for ("wild": job.g < 0) meet every slot.
// A wavefront takes p.rays_per_wave rays (lanes beyond that hold no ray but still hold slots).  64
// shares each list read among the most rays; a wavefront walks its lists one read after the other,
// though, so a small launch is given fewer rays per wavefront and more wavefronts to hide the reads.
// grid = blocks of 4 wavefronts.
// ---------------------------------------------------------------------------
constexpr int GRID_QCAP = 320; // queue entries per wavefront; draining starts when fewer than 64 are free

template <bool ANYHIT>
__global__ __launch_bounds__(BLOCK) void traverse_grid_kernel(TraverseParams p)
{
    __shared__ unsigned long long best_all[ANYHIT ? 1 : BLOCK];
    __shared__ int occ_all[ANYHIT ? BLOCK : 1];
    // the rays as the reference arithmetic takes them: (origin, tmax), (direction, -)
    __shared__ float4 ray_o_all[BLOCK], ray_d_all[BLOCK];
    __shared__ unsigned int queue_all[(BLOCK / 64) * GRID_QCAP];
    __shared__ float4 side_all[BLOCK]; // the rays of one cell side by side: scaled direction, lane (sweep_short)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = p.count ? *p.count : p.n;
    const int R = p.rays_per_wave > 0 ? p.rays_per_wave : rays_per_wave_for(n);
    unsigned long long *best = best_all + (ANYHIT ? 0 : wave * 64);
    int *occl = occ_all + (ANYHIT ? wave * 64 : 0);
    float4 *ray_o = ray_o_all + wave * 64, *ray_d = ray_d_all + wave * 64;
    unsigned int *queue = queue_all + wave * GRID_QCAP;
    float4 *side = side_all + wave * 64;
    // (the launch is laid out for the longest list there can be: wavefronts beyond the list's end leave at
    // once; a loop over groups here would cost the registers of an eighth wavefront per SIMD)
    const int s0 = (blockIdx.x * (BLOCK / 64) + wave) * R;
    if (s0 >= n) return; // the whole wavefront; no barriers below
    const int s = s0 + lane;

    bool valid = lane < R && s < n;
    const int sl = valid ? (p.alist ? p.alist[s] : s) : 0;
    const float qn = __int_as_float(0x7fc00000);
    float hx = qn, hy = qn, hz = qn, g = qn;
    unsigned int cell = 0;
    if (valid && ANYHIT && p.occ[sl] != 0) valid = false; // already decided (tmax not finite)
    const bool fresh = !ANYHIT && p.fresh_perm != nullptr;
    if (valid && fresh) {
        const int i = p.fresh_perm[sl];
        const RaySoA &r = p.fresh_rays;
        v3 o, d;
        if (p.fresh_raygen) {
            const int j = i / p.fresh_width;
            o = rtx::V(p.fresh_cam.pos[0], p.fresh_cam.pos[1], p.fresh_cam.pos[2]);
            d = raygen_direction(p.fresh_cam, p.fresh_width, p.fresh_height, i - j * p.fresh_width, p.fresh_row_map[j]);
        } else {
            o = rtx::V(r.ox[i], r.oy[i], r.oz[i]);
            d = rtx::V(r.dx[i], r.dy[i], r.dz[i]);
        }
        o = rtx::add(o, rtx::muls(d, 0.001f)); // r.origin = r.origin + r.dir * 0.001
        float bx, by, bz;
        shared_direction(o, d, p.origin, p.center, p.radius, p.reach_max, p.dir_sign, bx, by, bz, g);
        const float gg = fabsf(g);
        hx = bx * gg; hy = by * gg; hz = bz * gg;
        cell = rtg::oct_cell(bx, by, bz, p.grid_bits);
        ray_o[lane] = make_float4(o.x, o.y, o.z, 0.0f);
        ray_d[lane] = make_float4(d.x, d.y, d.z, 0.0f);
    } else if (valid) {
        const float bx = p.job.bx[sl], by = p.job.by[sl], bz = p.job.bz[sl];
        g = p.job.g[sl];
        const float gg = fabsf(g);
        hx = bx * gg; hy = by * gg; hz = bz * gg;
        cell = rtg::oct_cell(bx, by, bz, p.grid_bits);
        ray_o[lane] = make_float4(p.job.ox[sl], p.job.oy[sl], p.job.oz[sl], ANYHIT ? p.job.tmax[sl] : 0.0f);
        ray_d[lane] = make_float4(p.job.dx[sl], p.job.dy[sl], p.job.dz[sl], 0.0f);
    }
    // g is NaN for a ray that no filter passes (zero or non-finite direction or reach)
    const bool live = valid && (g == g);
    const bool wild = live && g < 0.0f;
    if (ANYHIT) occl[lane] = 0;
    else best[lane] = BEST_NONE;

    int qcount = 0; // wave-uniform
    unsigned long long n_cand = 0, n_slow = 0;

    // the reference arithmetic on queued (triangle, ray) pairs, one per lane
    auto drain = [&](bool all) {
#ifdef RT_DBG_GRID_NO_DRAIN
        n_cand += qcount;
        qcount = 0;
        return;
#endif
        while (qcount >= 64 || (all && qcount > 0)) {
            const int take = qcount >= 64 ? 64 : qcount;
            qcount -= take;
            if (lane < take) {
                const unsigned int e = queue[qcount + lane];
                const int r = (int)(e & 63u);
                const int tri = (int)(e >> 6);
                if (!(ANYHIT && occl[r] != 0)) {
                    const float4 ro = ray_o[r], rd = ray_d[r];
                    const v3 o = rtx::V(ro.x, ro.y, ro.z), d = rtx::V(rd.x, rd.y, rd.z);
                    const v3 X = vpos(p.verts, 3 * tri), Y = vpos(p.verts, 3 * tri + 1), Z = vpos(p.verts, 3 * tri + 2);
                    float t;
                    if (rtx::ray_vs_triangle(o, d, X, Y, Z, t)) {
                        if (ANYHIT) {
                            if (t > 0.0f && t <= ro.w) occl[r] = 1; // every writer writes 1
                        } else if (t > 0.0f) {
                            const unsigned long long key =
                                ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(unsigned int)tri;
                            atomicMin(&best[r], key);
                        }
                    }
                }
            }
            n_cand += take;
        }
    };

    // the rays in `group` against the slots list[0 .. cnt) followed by elist[0 .. ecnt)
    // (list == nullptr: slots 0 .. cnt)
    auto sweep = [&](const int *list, int cnt, const int *elist, int ecnt, unsigned long long group) {
        for (int base = 0; base < cnt + ecnt && group; base += 64) {
            const int idx = base + lane;
            int e = -1, tri = -1;
            if (idx < cnt) e = list ? list[idx] : idx;
            else if (idx < cnt + ecnt) e = elist[idx - cnt];
            float4 r0 = make_float4(qn, qn, qn, qn), r1 = r0, r2 = r0;
            if (e >= 0) {
                const float4 *rp = reinterpret_cast<const float4 *>(p.recs + e);
                r0 = rp[0]; r1 = rp[1]; r2 = rp[2];
                tri = __float_as_int(r2.y); // (prep_origin_kernel leaves the slot's triangle there)
            }
            if (tri < 0) r0.x = qn; // an empty slot: no ray passes a NaN record
            const float rec[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
            unsigned long long m = group;
#ifdef RT_DBG_GRID_NO_SWEEP
            m = 0;
#endif
            while (m) {
                const int r = __builtin_ctzll(m);
                m &= m - 1;
                const float rx = readlane_f(hx, r), ry = readlane_f(hy, r), rz = readlane_f(hz, r);
                float lo, hi;
                rtf::filter_origin_lohi(rec, rx, ry, rz, lo, hi);
                const bool pass = fmaxf(lo, -hi) >= -1.0f;
                const unsigned long long h = __builtin_amdgcn_ballot_w64(pass);
                if (h) {
                    if (pass) queue[qcount + lanes_below(h)] = ((unsigned int)tri << 6) | (unsigned int)r;
                    qcount += __popcll(h);
                    if (qcount > GRID_QCAP - 64) drain(false);
                }
            }
            n_slow += __popcll(group);
            if (ANYHIT && __popcll(group) >= 8) {
                // worth a partly filled batch: rays found occluded skip the rest of the list
                drain(true);
                group &= ~__builtin_amdgcn_ballot_w64(occl[lane] != 0);
            }
        }
    };

    // Several rays at a time when the cell's list is short.  A list of at most 8 / 16 / 32 slots (with the slots
    // that are on every list) leaves most lanes of a 64-slot read without a slot, and a camera or a light
    // outside the mesh sees few triangles in a cell: there the list is read 8 / 4 / 2 times over, once per L
    // lanes, the rays of the cell are laid side by side in LDS (direction and lane, in the order of their
    // lanes), and every round tests 64 / L of them, each set of L lanes fetching its own.
    auto sweep_short = [&](int log_l, const int *list, int cnt, const int *elist, int ecnt, unsigned long long group) {
        const int sub = lane >> log_l, j = lane & ((1 << log_l) - 1); // L = 2^log_l lanes to a ray
        int e = -1, tri = -1;
        if (j < cnt) e = list[j];
        else if (j < cnt + ecnt) e = elist[j - cnt];
        float4 r0 = make_float4(qn, qn, qn, qn), r1 = r0, r2 = r0;
        if (e >= 0) {
            const float4 *rp = reinterpret_cast<const float4 *>(p.recs + e);
            r0 = rp[0]; r1 = rp[1]; r2 = rp[2];
            tri = __float_as_int(r2.y);
        }
        if (tri < 0) r0.x = qn; // an empty slot: no ray passes a NaN record
        const float rec[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
        const int n_group = __popcll(group);
        n_slow += n_group;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (the last group's reads of the table are done)
        __builtin_amdgcn_wave_barrier();
        if ((group >> lane) & 1ull) side[lanes_below(group)] = make_float4(hx, hy, hz, __int_as_float(lane));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float4 ahead = side[sub];
        for (int first = 0; first < n_group; first += 64 >> log_l) {
            const int at = first + sub;
            const float4 ray = ahead;
            ahead = side[(at + (64 >> log_l)) & 63]; // (the next round's ray is on its way while this one is tested)
            float lo, hi;
            rtf::filter_origin_lohi(rec, ray.x, ray.y, ray.z, lo, hi);
            const bool pass = at < n_group && fmaxf(lo, -hi) >= -1.0f;
            const unsigned long long h = __builtin_amdgcn_ballot_w64(pass);
            if (h) {
                if (pass) queue[qcount + lanes_below(h)] = ((unsigned int)tri << 6) | (unsigned int)__float_as_int(ray.w);
                qcount += __popcll(h);
                if (qcount > GRID_QCAP - 64) drain(false);
            }
        }
    };

    unsigned long long remaining = __builtin_amdgcn_ballot_w64(live && !wild);
    // the slots that are on every cell's list follow the cell's own
    const unsigned int every = 1u << (2 * p.grid_bits);
    const unsigned int eoff = p.grid_off[every];
    const int ecnt = (int)(p.grid_off[every + 1] - eoff);
    while (remaining) {
        const int leader = __builtin_ctzll(remaining);
        const unsigned int c0 = (unsigned int)__builtin_amdgcn_readlane((int)cell, leader);
        const unsigned long long group = __builtin_amdgcn_ballot_w64(cell == c0) & remaining;
        remaining &= ~group;
        const unsigned int off = p.grid_off[c0];
        const int cnt = (int)(p.grid_off[c0 + 1] - off);
        const int len = cnt + ecnt, n_group = __popcll(group);
        if (len <= 16 ? n_group >= 3 : (len <= 32 && n_group >= 5))
            sweep_short(len <= 8 ? 3 : (len <= 16 ? 4 : 5), p.grid_entries + off, cnt, p.grid_entries + eoff, ecnt, group);
        else
            sweep(p.grid_entries + off, cnt, p.grid_entries + eoff, ecnt, group);
    }
    const unsigned long long wilds = __builtin_amdgcn_ballot_w64(wild);
    if (wilds) sweep(nullptr, p.num_slots, nullptr, 0, wilds);
    drain(true);

    if (lane == 0) {
        add_counter(p.counters, ANYHIT ? C_SH_CAND : C_CH_CAND, n_cand);
        add_counter(p.counters, ANYHIT ? C_SH_SLOW : C_CH_SLOW, n_slow);
        add_counter(p.counters, ANYHIT ? C_SH_PAIRS : C_CH_PAIRS,
                    (unsigned long long)min(R, n - s0) * (unsigned long long)p.num_tris);
        if (fresh) { // intersect_finish_kernel's bookkeeping
            add_counter(p.counters, C_CH_RAYS, (unsigned long long)min(R, n - s0));
            if (s0 == 0 && p.fresh_zero) *p.fresh_zero = 0;
        }
    }
    if (!valid) return;
    if (ANYHIT) {
        if (occl[lane] != 0) p.occ[sl] = 1;
    } else {
        const unsigned long long key = best[lane];
        if (fresh) {
            // the slot's only wavefront: the hit is final, and the ray is finished here
            const float4 ro = ray_o[lane], rd = ray_d[lane];
            intersect_finish_slot(p.fresh_rays, p.fresh_perm[sl], rtx::V(ro.x, ro.y, ro.z), rtx::V(rd.x, rd.y, rd.z), key,
                                  p.verts, p.num_tris, p.origin, p.fresh_depth, p.fresh_raygen != 0);
        } else if (key != BEST_NONE) {
            atomicMin(&p.best[sl], key);
        }
    }
}

// ---------------------------------------------------------------------------
// Clustered closest-hit traversal of arbitrary rays (bounce passes).  Same result as
// traverse_kernel<K, 0, false, *>, bit for bit, from two sweeps that together cover every pair the
// reference's arithmetic could accept (rt_filter.h, "Plane-band test"):
//   band sweep   the ill-conditioned pairs: direction nearly in the triangle's plane and origin nearly
//                on it.  Three kernels, by what the launch has: lists per direction cell
//                (traverse_bandcell_kernel), rays sorted by direction (traverse_band_kernel), or
//                neither (traverse_group_kernel, below).
//   space sweep  the well-conditioned pairs: a walk over cells in space (traverse_voxel_kernel).
// Results merge through the 64-bit atomicMin on (t bits, index), so the order of the visits does not
// matter.
//
// traverse_group_kernel: triangles grouped by normal direction, 64 per group: is the ray's direction
// in the group's band?  For each ray that is, one lane per member tests the member's own band and
// slab; pairs inside go to a queue and from there, one per lane, through the general edge filter and
// the reference arithmetic.  A wavefront tests its 64 rays against a group with one ballot, loads the
// group's member records once and then walks the set bits.  The workgroup's rays are staged in LDS so
// that any lane can fetch any of them.
// grid = (ray blocks, sub-ranges of the groups).
// ---------------------------------------------------------------------------
constexpr int NGROUP = 64;
constexpr int SUBGROUP = 8;
constexpr int GROUP_TILE = 256; // group records per LDS tile (4 KiB)

__global__ __launch_bounds__(BLOCK) void traverse_group_kernel(TraverseParams p)
{
    constexpr int NR = BLOCK;
    __shared__ float4 tile[GROUP_TILE];
    // the workgroup's rays: (unit direction, |g|), (origin, slot), (direction * |g|, -)
    __shared__ float4 ray_b[NR], ray_o[NR], ray_h[NR];
    __shared__ unsigned int queue_all[(BLOCK / 64) * qcap(1)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = p.count ? *p.count : p.n;
    const int g_per_y = (p.num_ngroups + (int)gridDim.y - 1) / (int)gridDim.y;
    const int g_begin = blockIdx.y * g_per_y, g_end = min(g_begin + g_per_y, p.num_ngroups);
    // a fixed number of workgroups along x: each takes every gridDim.x-th block of 256 rays
    for (int s0 = blockIdx.x * NR; s0 < n; s0 += (int)gridDim.x * NR) {
    __syncthreads(); // the staged rays of the previous block are no longer read

    const float qn = __int_as_float(0x7fc00000);
    const int r_own = wave * 64 + lane;
    const bool valid = s0 + r_own < n;
    float bx = qn, by = qn, bz = qn;
    bool wild;
    {
        const int sl = valid ? (p.alist ? p.alist[s0 + r_own] : s0 + r_own) : 0;
        float ox = qn, oy = qn, oz = qn, dx = qn, dy = qn, dz = qn, g = qn;
        if (valid) {
            ox = p.job.ox[sl]; oy = p.job.oy[sl]; oz = p.job.oz[sl];
            dx = p.job.dx[sl]; dy = p.job.dy[sl]; dz = p.job.dz[sl];
            bx = p.job.bx[sl]; by = p.job.by[sl]; bz = p.job.bz[sl];
            g = p.job.g[sl];
        }
        // outside the range the band records were built for: every band holds this ray
        wild = valid && !(g > 0.0f);
        const float gg = fabsf(g);
        ray_b[r_own] = make_float4(bx, by, bz, gg);
        ray_o[r_own] = make_float4(ox, oy, oz, __int_as_float(sl));
        ray_h[r_own] = make_float4(dx * gg, dy * gg, dz * gg, 0.0f);
    }
    const unsigned long long wild_mask = __builtin_amdgcn_ballot_w64(wild);
    const unsigned long long valid_mask = __builtin_amdgcn_ballot_w64(valid);

    unsigned int *queue = queue_all + wave * qcap(1);
    int qcount = 0; // wave-uniform
    unsigned long long n_cand = 0, n_slow = 0;

    // the queue: (triangle, ray) pairs inside a plane band and slab, one per lane -- the general edge
    // filter, then the reference arithmetic on the pairs it lets through (rare: the direction comes
    // from global memory)
    auto drain_pairs = [&](bool all) {
        while (qcount >= 64 || (all && qcount > 0)) {
            const int take = qcount >= 64 ? 64 : qcount;
            qcount -= take;
            bool pass = false;
            int tri = 0, r = 0;
            if (lane < take) {
                const unsigned int e = queue[qcount + lane];
                r = (int)(e & 2047u);
                tri = (int)(e >> 11);
                const float4 ro = ray_o[r], rh = ray_h[r];
                pass = rtf::filter_general(reinterpret_cast<const float *>(p.recs + tri), ro.x, ro.y, ro.z, rh.x, rh.y,
                                           rh.z) >= -1.0f;
            }
            n_cand += __popcll(__builtin_amdgcn_ballot_w64(pass));
            if (pass) {
                const float4 ro = ray_o[r];
                const int sl = __float_as_int(ro.w);
                const v3 o = rtx::V(ro.x, ro.y, ro.z), d = rtx::V(p.job.dx[sl], p.job.dy[sl], p.job.dz[sl]);
                const v3 X = vpos(p.verts, 3 * tri), Y = vpos(p.verts, 3 * tri + 1), Z = vpos(p.verts, 3 * tri + 2);
                float t;
                if (rtx::ray_vs_triangle(o, d, X, Y, Z, t) && t > 0.0f) {
                    const unsigned long long key =
                        ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(unsigned int)tri;
                    atomicMin(&p.best[sl], key);
                }
            }
        }
    };
    auto push_pairs = [&](bool in, int tri, int r) {
        const unsigned long long h = __builtin_amdgcn_ballot_w64(in);
        if (h) {
            if (in) queue[qcount + lanes_below(h)] = ((unsigned int)tri << 11) | (unsigned int)r;
            qcount += __popcll(h);
            if (qcount >= qcap(1) - 64) drain_pairs(false);
        }
    };

    __syncthreads(); // the staged rays

    for (int tile_base = g_begin; tile_base < g_end; tile_base += GROUP_TILE) {
        __syncthreads();
        const int tile_n = min(GROUP_TILE, g_end - tile_base);
#pragma unroll
        for (int i = 0; i < GROUP_TILE / BLOCK; i++) {
            const int e = i * BLOCK + tid;
            if (e < tile_n) tile[e] = p.ngroups[tile_base + e];
        }
        __syncthreads();
        for (int j = 0; j < tile_n; j++) {
            const float4 c = tile[j];
            const float tg = fabsf(fmaf(c.z, bz, fmaf(c.y, by, c.x * bx)));
            const unsigned long long m = (__builtin_amdgcn_ballot_w64(tg < c.w) | wild_mask) & valid_mask;
            if (!m) continue;
            n_slow++;
            // one lane per member: (n/theta, kappa/H), (n/H, triangle)
            const float4 *mr = p.nrecs + (size_t)((tile_base + j) * NGROUP + lane) * 2;
            const float4 m1 = mr[0], m2 = mr[1];
            const int tri = __float_as_int(m2.w);
            unsigned long long mk = m & ~wild_mask;
            const int rbase = wave * 64;
            if (mk) {
                // the next ray's record is fetched while the current one is tested
                int src = __builtin_ctzll(mk);
                mk &= mk - 1;
                float4 rb = ray_b[rbase + src], ro = ray_o[rbase + src];
                for (;;) {
                    const bool more = mk != 0ull;
                    int nsrc = 0;
                    float4 nb = rb, no = ro;
                    if (more) {
                        nsrc = __builtin_ctzll(mk);
                        mk &= mk - 1;
                        nb = ray_b[rbase + nsrc];
                        no = ray_o[rbase + nsrc];
                    }
                    const float t = fmaf(m1.z, rb.z, fmaf(m1.y, rb.y, m1.x * rb.x));
                    const float h = fmaf(m2.z, ro.z, fmaf(m2.y, ro.y, fmaf(m2.x, ro.x, -m1.w)));
                    push_pairs(fmaxf(fabsf(t), fabsf(h)) < 1.0f, tri, rbase + src);
                    if (!more) break;
                    src = nsrc; rb = nb; ro = no;
                }
            }
            // rays outside the records' range: every member that can be accepted at all
            unsigned long long wk = m & wild_mask;
            while (wk) {
                const int src = __builtin_ctzll(wk);
                wk &= wk - 1;
                push_pairs(tri >= 0 && m1.w != 2.0f, tri, rbase + src);
            }
        }
    }
    drain_pairs(true);

    if (lane == 0) {
        add_counter(p.counters, C_CH_CAND, n_cand);
        add_counter(p.counters, C_CH_SLOW, n_slow);
    }
    }
}

// ---------------------------------------------------------------------------
// Sweep A of the clustered traversal as a streaming kernel, for rays sorted by direction: the rays
// of a wavefront then point the same way (up to sign), so a group of triangles with similar
// normals holds either none of them -- the group is skipped -- or most of them, and its 64 member
// records are walked with one lane per ray, read from LDS with wave-uniform addresses.  A ray
// inside a member's band and slab is queued; the queue is handled 64 pairs at a time, one per lane:
// general edge filter, then the reference arithmetic (doing either on the spot stalls the whole
// wavefront on a chain of dependent loads for the sake of a lane or two).
// Same pairs reach the reference arithmetic as in traverse_group_kernel.
// grid = (ray blocks, sub-ranges of the groups).
// ---------------------------------------------------------------------------
constexpr int BAND_TILE_GROUPS = 8; // groups per LDS tile: 512 member records of 32 bytes (16 KiB)

template <int K>
__global__ __attribute__((amdgpu_waves_per_eu(6, 6))) __launch_bounds__(BLOCK) void traverse_band_kernel(TraverseParams p)
{
    __shared__ float4 mem4[BAND_TILE_GROUPS * NGROUP * 2];
    __shared__ float4 grp4[BAND_TILE_GROUPS];
    __shared__ float4 sub4[BAND_TILE_GROUPS * (NGROUP / SUBGROUP)];
    __shared__ unsigned int queue_all[(BLOCK / 64) * qcap(K)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = p.count ? *p.count : p.n;
    int g_per_y = (p.num_ngroups + (int)gridDim.y - 1) / (int)gridDim.y;
    g_per_y = (g_per_y + BAND_TILE_GROUPS - 1) / BAND_TILE_GROUPS * BAND_TILE_GROUPS;
    const int g_begin = blockIdx.y * g_per_y, g_end = min(g_begin + g_per_y, p.num_ngroups);
    if (g_begin >= g_end) return;
    // a fixed number of workgroups along x: each takes every gridDim.x-th block of 256 K rays
    for (int s0 = blockIdx.x * (BLOCK * K); s0 < n; s0 += (int)gridDim.x * (BLOCK * K)) {

    int slot[K];
    float ox[K], oy[K], oz[K], bx[K], by[K], bz[K];
    bool valid[K], wild[K];
    const float qn = __int_as_float(0x7fc00000);
    bool any_wild = false;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int s = s0 + (wave * K + k) * 64 + lane;
        valid[k] = s < n;
        slot[k] = valid[k] ? (p.alist ? p.alist[s] : s) : 0;
        ox[k] = oy[k] = oz[k] = bx[k] = by[k] = bz[k] = qn;
        float g = qn;
        if (valid[k]) {
            ox[k] = p.job.ox[slot[k]]; oy[k] = p.job.oy[slot[k]]; oz[k] = p.job.oz[slot[k]];
            bx[k] = p.job.bx[slot[k]]; by[k] = p.job.by[slot[k]]; bz[k] = p.job.bz[slot[k]];
            g = p.job.g[slot[k]];
        }
        wild[k] = valid[k] && !(g > 0.0f); // outside the records' range: in every band and slab
        any_wild = any_wild || wild[k];
    }
    const bool wave_wild = any64(any_wild);

    unsigned int *queue = queue_all + wave * qcap(K);
    int qcount = 0; // wave-uniform
    unsigned long long n_cand = 0, n_slow = 0;

    // the edge filter and the reference arithmetic for queued (triangle, ray) pairs, one per lane
    auto drain = [&](bool all) {
        while (qcount >= 64 || (all && qcount > 0)) {
            const int take = qcount >= 64 ? 64 : qcount;
            qcount -= take;
            if (lane < take) {
                const unsigned int e = queue[qcount + lane];
                const int tri = (int)(e >> 11);
                const int s = s0 + (int)(e & 2047u);
                const int sl = p.alist ? p.alist[s] : s;
                const v3 o = rtx::V(p.job.ox[sl], p.job.oy[sl], p.job.oz[sl]);
                const v3 d = rtx::V(p.job.dx[sl], p.job.dy[sl], p.job.dz[sl]);
                // the general edge filter first: most pairs inside a band miss their triangle
                const float gg = fabsf(p.job.g[sl]);
                if (rtf::filter_general(reinterpret_cast<const float *>(p.recs + tri), o.x, o.y, o.z, d.x * gg,
                                        d.y * gg, d.z * gg) >= -1.0f) {
                    const v3 X = vpos(p.verts, 3 * tri), Y = vpos(p.verts, 3 * tri + 1), Z = vpos(p.verts, 3 * tri + 2);
                    float t;
                    if (rtx::ray_vs_triangle(o, d, X, Y, Z, t) && t > 0.0f) {
                        const unsigned long long key =
                            ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(unsigned int)tri;
                        atomicMin(&p.best[sl], key);
                    }
                }
            }
        }
    };

    for (int tile_base = g_begin; tile_base < g_end; tile_base += BAND_TILE_GROUPS) {
        __syncthreads();
        const int tile_g = min(BAND_TILE_GROUPS, g_end - tile_base);
        const float4 *src = p.nrecs + (size_t)tile_base * NGROUP * 2;
#pragma unroll
        for (int i = 0; i < (BAND_TILE_GROUPS * NGROUP * 2) / BLOCK; i++) {
            const int e = i * BLOCK + tid;
            if (e < tile_g * NGROUP * 2) mem4[e] = src[e];
        }
        if (tid < tile_g) grp4[tid] = p.ngroups[tile_base + tid];
        else if (tid >= 64 && tid - 64 < tile_g * (NGROUP / SUBGROUP))
            sub4[tid - 64] = p.nsubs[(size_t)tile_base * (NGROUP / SUBGROUP) + (tid - 64)];
        __syncthreads();
        for (int gi = 0; gi < tile_g; gi++) {
            const float4 gc = grp4[gi];
            float G = fabsf(fmaf(gc.z, bz[0], fmaf(gc.y, by[0], gc.x * bx[0])));
#pragma unroll
            for (int k = 1; k < K; k++) G = fminf(G, fabsf(fmaf(gc.z, bz[k], fmaf(gc.y, by[k], gc.x * bx[k]))));
            if (!(any64(G < gc.w) || wave_wild)) continue; // no ray of the wavefront in the group's band
            const float4 *mrec = mem4 + gi * NGROUP * 2;
            for (int j = 0; j < NGROUP; j += 2) {
                if ((j & (SUBGROUP - 1)) == 0) {
                    // the same test for the part of the group that starts here
                    const float4 sc = sub4[gi * (NGROUP / SUBGROUP) + j / SUBGROUP];
                    float S = fabsf(fmaf(sc.z, bz[0], fmaf(sc.y, by[0], sc.x * bx[0])));
#pragma unroll
                    for (int k = 1; k < K; k++) S = fminf(S, fabsf(fmaf(sc.z, bz[k], fmaf(sc.y, by[k], sc.x * bx[k]))));
                    if (!(any64(S < sc.w) || wave_wild)) {
                        j += SUBGROUP - 2;
                        continue;
                    }
                }
                const float4 a1 = mrec[2 * j], a2 = mrec[2 * j + 1], b1 = mrec[2 * j + 2], b2 = mrec[2 * j + 3];
                float ta[K], tb[K];
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const float t0 = fmaf(a1.z, bz[k], fmaf(a1.y, by[k], a1.x * bx[k]));
                    const float h0 = fmaf(a2.z, oz[k], fmaf(a2.y, oy[k], fmaf(a2.x, ox[k], -a1.w)));
                    ta[k] = fmaxf(fabsf(t0), fabsf(h0));
                    const float t1 = fmaf(b1.z, bz[k], fmaf(b1.y, by[k], b1.x * bx[k]));
                    const float h1 = fmaf(b2.z, oz[k], fmaf(b2.y, oy[k], fmaf(b2.x, ox[k], -b1.w)));
                    tb[k] = fmaxf(fabsf(t1), fabsf(h1));
                }
                float Ga = ta[0], Gb = tb[0];
#pragma unroll
                for (int k = 1; k < K; k++) { Ga = fminf(Ga, ta[k]); Gb = fminf(Gb, tb[k]); }
                if (!(any64(Ga < 1.0f || Gb < 1.0f) || wave_wild)) continue;
                n_slow++;
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const float4 m1 = u == 0 ? a1 : b1, m2 = u == 0 ? a2 : b2;
                    const int tri = __float_as_int(m2.w);
                    if (tri < 0 || m1.w == 2.0f) continue; // padding / never accepted (wave-uniform)
                    const float Gu = u == 0 ? Ga : Gb;
                    if (!(any64(Gu < 1.0f) || wave_wild)) continue;
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        const float t = u == 0 ? ta[k] : tb[k];
                        bool in = ((t < 1.0f) || wild[k]) && valid[k];
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
                        if (m) {
                            n_cand += __popcll(m);
                            if (in)
                                queue[qcount + lanes_below(m)] =
                                    ((unsigned int)tri << 11) | (unsigned int)((wave * K + k) * 64 + lane);
                            qcount += __popcll(m);
                        }
                    }
                    if (qcount >= qcap(K) - 64 * K) drain(false);
                }
            }
        }
    }
    drain(true);

    if (lane == 0) {
        add_counter(p.counters, C_CH_CAND, n_cand);
        add_counter(p.counters, C_CH_SLOW, n_slow);
    }
    }
}

// ---------------------------------------------------------------------------
// Direction cells of the band sweep: for every cell that reaches the hemisphere z >= 0, the member
// records (positions in nrecs) whose band shares a direction with the cell.  One thread per member
// walks the cells from the coarsest level down, as grid_build_kernel does.  A member whose band is
// every direction goes on the list "everywhere" (index num_cells); padding and never-accepted
// members on none.  FILL as in grid_build_kernel.
// ---------------------------------------------------------------------------
template <bool FILL>
__global__ __launch_bounds__(BLOCK) void band_build_kernel(const rtg::CellCone *__restrict__ cells, int bits,
                                                           const float4 *__restrict__ nrecs, int num_members,
                                                           unsigned int *__restrict__ count,
                                                           const unsigned int *__restrict__ off,
                                                           int *__restrict__ entries)
{
    __shared__ unsigned int stack_all[BUILD_STACK][BLOCK];
    const long long gid = (long long)blockIdx.x * BLOCK + threadIdx.x;
    // consecutive lanes take consecutive members (similar normals) against the same top cell
    const int pos = (int)(gid % num_members);
    const unsigned int top = (unsigned int)(gid / num_members);
    if (top >= (unsigned int)BUILD_TOP_CELLS) return;
    const float4 m1 = nrecs[2 * pos], m2 = nrecs[2 * pos + 1];
    if (__float_as_int(m2.w) < 0 || m1.w == 2.0f) return; // padding / never accepted
    const unsigned int num_cells = 1u << (2 * bits);
    auto emit = [&](unsigned int cell) {
        const unsigned int k = atomicAdd(&count[cell], 1u);
        if (FILL) entries[off[cell] + k] = pos;
    };
    if (m1.x == 0.0f && m1.y == 0.0f && m1.z == 0.0f) { // its band is every direction
        if (top == 0) emit(num_cells);
        return;
    }
    unsigned int (*stack)[BLOCK] = stack_all; // [depth][thread]: (level << 24) | cell
    const int tx = threadIdx.x;
    int sp = 0;
    stack[sp++][tx] = ((unsigned int)BUILD_TOP_LEVEL << 24) | top;
    while (sp > 0) {
        const unsigned int e = stack[--sp][tx];
        const int level = (int)(e >> 24);
        const unsigned int cell = e & 0xffffffu;
        const rtg::CellCone cc = cells[rtg::level_offset(level) + cell];
        if (!rtg::cell_reaches_upper_hemisphere(cc) || !rtg::cell_meets_band(cc, m1.x, m1.y, m1.z, 1.0f)) continue;
        if (level == bits) {
            emit(cell);
        } else {
            for (unsigned int k = 0; k < 4; k++) stack[sp++][tx] = ((unsigned int)(level + 1) << 24) | (cell * 4 + k);
        }
    }
}

// ---------------------------------------------------------------------------
// Origin buckets of the band sweep's direction cells (rt_grid.h, "Origin buckets"): the list of cell c
// (cell_off, cell_entries: what band_build_kernel made) is split into G x G lists by the origin of the ray.
// List c * (G^2 + 1) + bucket holds the members whose slab reaches the bucket, list c * (G^2 + 1) + G^2 the
// whole cell once more, list num_cells * (G^2 + 1) the members that are in every cell.  One workgroup per
// cell (the last one copies the list "everywhere"): a thread per member forms the member's strip in the
// cell's frame and files it under the buckets the strip meets, counted in LDS -- the lists of one cell lie
// next to each other in memory, so a workgroup's stores fall into a few KB.  FILL = false writes count[]
// (every list's length, no atomics on global memory), FILL = true the entries, off[] being the scan of
// count[].  The order of a list's entries is whatever the atomics make it; no result depends on it.
// grid = num_cells + 1 workgroups.
// ---------------------------------------------------------------------------
constexpr int BUCKET_MAX_G = 16;
template <bool FILL>
__global__ __launch_bounds__(BLOCK) void band_bucket_kernel(const unsigned int *__restrict__ cell_off,
                                                            const int *__restrict__ cell_entries, int bits,
                                                            const float4 *__restrict__ nrecs,
                                                            const rtg::BucketFrame *__restrict__ frames, int G,
                                                            float cx, float cy, float cz,
                                                            unsigned int *__restrict__ count,
                                                            const unsigned int *__restrict__ off,
                                                            int *__restrict__ entries)
{
    __shared__ unsigned int filed[BUCKET_MAX_G * BUCKET_MAX_G];
    const unsigned int num_cells = 1u << (2 * bits);
    const unsigned int cell = blockIdx.x;
    const unsigned int per_cell = (unsigned int)(G * G + 1);
    const unsigned int lo = cell_off[cell];
    const int n = (int)(cell_off[cell + 1] - lo);
    const int tid = threadIdx.x;
    if (cell == num_cells) { // the list "everywhere": as it is
        const unsigned int key = num_cells * per_cell;
        if (!FILL) {
            if (tid == 0) count[key] = (unsigned int)n;
        } else {
            for (int i = tid; i < n; i += BLOCK) entries[off[key] + i] = cell_entries[lo + i];
        }
        return;
    }
    const unsigned int key0 = cell * per_cell;
    if (n == 0) { // (a cell that does not reach the upper hemisphere, or that no band crosses)
        if (!FILL)
            for (unsigned int b = tid; b < per_cell; b += BLOCK) count[key0 + b] = 0u;
        return;
    }
    for (int b = tid; b < G * G; b += BLOCK) filed[b] = 0u;
    __syncthreads();
    const float ctr[3] = {cx, cy, cz};
    const rtg::BucketFrame frame = frames[cell];
    const unsigned int whole = FILL ? off[key0 + per_cell - 1u] : 0u;
    for (int i = tid; i < n; i += BLOCK) {
        const int pos = cell_entries[lo + i];
        if (FILL) entries[whole + i] = pos;
        const float4 m1 = nrecs[2 * pos], m2 = nrecs[2 * pos + 1];
        const rtg::BucketStrip st = rtg::bucket_strip(frame, m2.x, m2.y, m2.z, m1.w, ctr, G);
        for (int ix = 0; ix < G; ix++) {
            int blo, bhi;
            rtg::bucket_column_range(st, ix, G, &blo, &bhi);
            for (int iy = blo; iy <= bhi; iy++)
                if (rtg::bucket_meets_strip(st, ix, iy)) {
                    const int b = iy * G + ix;
                    const unsigned int k = atomicAdd(&filed[b], 1u);
                    if (FILL) entries[off[key0 + b] + k] = pos;
                }
        }
    }
    if (!FILL) {
        __syncthreads();
        for (unsigned int b = tid; b < per_cell; b += BLOCK) count[key0 + b] = b + 1u == per_cell ? (unsigned int)n : filed[b];
    }
}

// ---------------------------------------------------------------------------
// Band sweep (sweep A of the clustered traversal) over direction cells.  Same pairs reach the general
// edge filter as in traverse_band_kernel -- the members whose band holds the ray's direction and
// whose slab holds its origin, every member for a ray outside the records' range -- but each ray
// meets only the members on the list of its direction's cell.  Structure as traverse_grid_kernel:
// one ray per lane, wavefronts independent, the cell's list read 64 members at a time, one member per
// lane, the rays of the cell broadcast in turn.  Survivors are queued and go through the edge filter
// and the reference arithmetic one pair per lane; results merge into p.best by atomicMin.
// grid = ray blocks of 256.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void traverse_bandcell_kernel(TraverseParams p)
{
    __shared__ unsigned long long best_all[BLOCK];
    __shared__ float4 ray_o_all[BLOCK], ray_d_all[BLOCK]; // (origin, |g|), (direction, -)
    __shared__ unsigned int queue_all[(BLOCK / 64) * GRID_QCAP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = p.count ? *p.count : p.n;
    const int R = p.rays_per_wave > 0 ? p.rays_per_wave : rays_per_wave_for(n);
    unsigned long long *best = best_all + wave * 64;
    float4 *ray_o = ray_o_all + wave * 64, *ray_d = ray_d_all + wave * 64;
    unsigned int *queue = queue_all + wave * GRID_QCAP;
    const int s0 = (blockIdx.x * (BLOCK / 64) + wave) * R;
    if (s0 >= n) return; // the whole wavefront (the launch is laid out for the longest list); no barriers below
    const int s = s0 + lane;

    const bool valid = lane < R && s < n;
    const int sl = valid ? (p.alist ? p.alist[s] : s) : 0;
    const float qn = __int_as_float(0x7fc00000);
    float ox = qn, oy = qn, oz = qn, bx = qn, by = qn, bz = qn, g = qn;
    unsigned int cell = 0; // the list of the ray: its direction's cell, and its origin's bucket there
    const unsigned int per_cell = p.band_g > 1 ? (unsigned int)(p.band_g * p.band_g + 1) : 1u;
    if (valid) {
        ox = p.job.ox[sl]; oy = p.job.oy[sl]; oz = p.job.oz[sl];
        bx = p.job.bx[sl]; by = p.job.by[sl]; bz = p.job.bz[sl];
        g = p.job.g[sl];
        cell = rtg::oct_cell_folded(bx, by, bz, p.grid_bits);
        if (p.band_g > 1) { // (an origin outside the buckets: the cell's whole list)
            const int bucket = rtg::bucket_of(p.bucket_frames[cell], ox - p.band_cx, oy - p.band_cy, oz - p.band_cz,
                                              p.band_g);
            cell = cell * per_cell + (bucket < 0 ? per_cell - 1u : (unsigned int)bucket);
        }
        ray_o[lane] = make_float4(ox, oy, oz, fabsf(g));
        ray_d[lane] = make_float4(p.job.dx[sl], p.job.dy[sl], p.job.dz[sl], 0.0f);
    }
    const bool wild = valid && !(g > 0.0f); // outside the records' range: in every band and slab
    best[lane] = BEST_NONE;

    int qcount = 0; // wave-uniform
    unsigned long long n_cand = 0, n_slow = 0;

    // the edge filter and the reference arithmetic for queued (triangle, ray) pairs, one per lane
    auto drain = [&](bool all) {
        while (qcount >= 64 || (all && qcount > 0)) {
            const int take = qcount >= 64 ? 64 : qcount;
            qcount -= take;
            if (lane < take) {
                const unsigned int e = queue[qcount + lane];
                const int r = (int)(e & 63u);
                const int tri = (int)(e >> 6);
                const float4 ro = ray_o[r], rd = ray_d[r];
                const float gg = ro.w;
                if (rtf::filter_general(reinterpret_cast<const float *>(p.recs + tri), ro.x, ro.y, ro.z, rd.x * gg,
                                        rd.y * gg, rd.z * gg) >= -1.0f) {
                    const v3 o = rtx::V(ro.x, ro.y, ro.z), d = rtx::V(rd.x, rd.y, rd.z);
                    const v3 X = vpos(p.verts, 3 * tri), Y = vpos(p.verts, 3 * tri + 1), Z = vpos(p.verts, 3 * tri + 2);
                    float t;
                    if (rtx::ray_vs_triangle(o, d, X, Y, Z, t) && t > 0.0f) {
                        const unsigned long long key =
                            ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(unsigned int)tri;
                        atomicMin(&best[r], key);
                    }
                }
            }
            n_cand += take;
        }
    };

    // the rays in `group` against the members list[0 .. cnt) (list == nullptr: positions 0 .. cnt,
    // every one of them a candidate: rays outside the records' range)
    auto sweep = [&](const int *list, int cnt, const int *elist, int ecnt, unsigned long long group) {
        for (int base = 0; base < cnt + ecnt; base += 64) {
            const int idx = base + lane;
            float4 m1 = make_float4(0.0f, 0.0f, 0.0f, 2.0f), m2 = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
            if (idx < cnt + ecnt) {
                const int pos = idx < cnt ? (list ? list[idx] : idx) : elist[idx - cnt];
                m1 = p.nrecs[2 * pos];
                m2 = p.nrecs[2 * pos + 1];
            }
            const int tri = __float_as_int(m2.w);
            const bool member = tri >= 0 && m1.w != 2.0f; // not padding, not "never accepted"
            unsigned long long m = group;
            while (m) {
                const int r = __builtin_ctzll(m);
                m &= m - 1;
                bool in = member;
                if (list) {
                    const float rbx = readlane_f(bx, r), rby = readlane_f(by, r), rbz = readlane_f(bz, r);
                    const float rox = readlane_f(ox, r), roy = readlane_f(oy, r), roz = readlane_f(oz, r);
                    const float t0 = fmaf(m1.z, rbz, fmaf(m1.y, rby, m1.x * rbx));
                    const float h0 = fmaf(m2.z, roz, fmaf(m2.y, roy, fmaf(m2.x, rox, -m1.w)));
                    in = in && fmaxf(fabsf(t0), fabsf(h0)) < 1.0f;
                }
                const unsigned long long h = __builtin_amdgcn_ballot_w64(in);
                if (h) {
                    if (in) queue[qcount + lanes_below(h)] = ((unsigned int)tri << 6) | (unsigned int)r;
                    qcount += __popcll(h);
                    if (qcount > GRID_QCAP - 64) drain(false);
                }
            }
            n_slow += __popcll(group);
        }
    };

    unsigned long long remaining = __builtin_amdgcn_ballot_w64(valid && !wild);
    // the members that are on every cell's list follow the cell's own
    const unsigned int every = (1u << (2 * p.grid_bits)) * per_cell;
    const unsigned int eoff = p.band_off[every];
    const int ecnt = (int)(p.band_off[every + 1] - eoff);
    // The rays of the wavefront, cell by cell, each cell's list 64 members a step.  A step is two loads
    // one after the other (the member's position on the list, then its record), and a wavefront whose rays
    // sit in cells of their own makes hundreds of steps: the steps are laid out beforehand (where each
    // cell's list starts is read up front, one cell per lane) and their loads run ahead -- positions two
    // steps, records one step before the step that tests them.
    unsigned int my_off = 0;
    int my_cnt = 0;
    if (valid && !wild) {
        my_off = p.band_off[cell];
        my_cnt = (int)(p.band_off[cell + 1] - my_off);
    }
    struct Step { // wave-uniform
        unsigned long long group; // the rays of the step's cell; 0: past the last step
        unsigned int off;
        int cnt, base;
    };
    auto advance = [&](Step st) {
        st.base += 64;
        while (!st.group || st.base >= st.cnt + ecnt) {
            if (!remaining) {
                st.group = 0;
                return st;
            }
            const int leader = __builtin_ctzll(remaining);
            const unsigned int c0 = (unsigned int)__builtin_amdgcn_readlane((int)cell, leader);
            st.group = __builtin_amdgcn_ballot_w64(cell == c0) & remaining;
            remaining &= ~st.group;
            st.off = (unsigned int)__builtin_amdgcn_readlane((int)my_off, leader);
            st.cnt = __builtin_amdgcn_readlane(my_cnt, leader);
            st.base = 0;
        }
        return st;
    };
    auto load_pos = [&](const Step &st) {
        const int idx = st.base + lane;
        if (!st.group || idx >= st.cnt + ecnt) return -1;
        return idx < st.cnt ? p.band_entries[st.off + idx] : p.band_entries[eoff + idx - st.cnt];
    };
    struct Recs {
        float4 m1, m2;
    };
    auto load_recs = [&](int pos) {
        Recs r{make_float4(0.0f, 0.0f, 0.0f, 2.0f), make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1))};
        if (pos >= 0) {
            r.m1 = p.nrecs[2 * pos];
            r.m2 = p.nrecs[2 * pos + 1];
        }
        return r;
    };
    // (the first rays of the step's cell stay in scalar registers over the steps of the cell: a list is
    // several steps long, and fetching a ray from its lane costs as much as testing it)
    constexpr int HELD = 4;
    float held[HELD][6];
    int held_lane[HELD];
#pragma unroll
    for (int k = 0; k < HELD; k++) {
        held_lane[k] = -1;
#pragma unroll
        for (int j = 0; j < 6; j++) held[k][j] = 0.0f;
    }
    auto test_step = [&](const Step &st, const Recs &rc) {
        const float4 m1 = rc.m1, m2 = rc.m2;
        const int tri = __float_as_int(m2.w);
        const bool member = tri >= 0 && m1.w != 2.0f; // not padding, not "never accepted"
        unsigned long long m = st.group;
        if (st.base == 0) { // a new cell: its first rays
            unsigned long long g = st.group;
#pragma unroll
            for (int k = 0; k < HELD; k++) {
                held_lane[k] = g ? __builtin_ctzll(g) : -1;
                if (g) {
                    const int r = held_lane[k];
                    g &= g - 1;
                    held[k][0] = readlane_f(bx, r); held[k][1] = readlane_f(by, r); held[k][2] = readlane_f(bz, r);
                    held[k][3] = readlane_f(ox, r); held[k][4] = readlane_f(oy, r); held[k][5] = readlane_f(oz, r);
                }
            }
        }
        auto test_ray = [&](int r, float rbx, float rby, float rbz, float rox, float roy, float roz) {
            const float t0 = fmaf(m1.z, rbz, fmaf(m1.y, rby, m1.x * rbx));
            const float h0 = fmaf(m2.z, roz, fmaf(m2.y, roy, fmaf(m2.x, rox, -m1.w)));
            const bool in = member && fmaxf(fabsf(t0), fabsf(h0)) < 1.0f;
            const unsigned long long h = __builtin_amdgcn_ballot_w64(in);
            if (h) {
                if (in) queue[qcount + lanes_below(h)] = ((unsigned int)tri << 6) | (unsigned int)r;
                qcount += __popcll(h);
                if (qcount > GRID_QCAP - 64) drain(false);
            }
        };
#pragma unroll
        for (int k = 0; k < HELD; k++)
            if (held_lane[k] >= 0) {
                test_ray(held_lane[k], held[k][0], held[k][1], held[k][2], held[k][3], held[k][4], held[k][5]);
                m &= m - 1; // the held rays are the group's lowest lanes, in order
            }
        while (m) {
            const int r = __builtin_ctzll(m);
            m &= m - 1;
            const float rbx = readlane_f(bx, r), rby = readlane_f(by, r), rbz = readlane_f(bz, r);
            const float rox = readlane_f(ox, r), roy = readlane_f(oy, r), roz = readlane_f(oz, r);
            test_ray(r, rbx, rby, rbz, rox, roy, roz);
        }
        n_slow += __popcll(st.group);
    };
    {
        Step sa = advance(Step{0ull, 0u, 0, 0}); // the step whose positions are being read
        int pos_a = load_pos(sa);
        Step sb = sa;                            // the step whose records are being read
        Recs rec_b = load_recs(pos_a);
        if (sa.group) {
            sa = advance(sa);
            pos_a = load_pos(sa);
        }
        while (sb.group) {
            const Step sc = sb;
            const Recs rec_c = rec_b;
            sb = sa;
            rec_b = load_recs(pos_a);
            if (sa.group) {
                sa = advance(sa);
                pos_a = load_pos(sa);
            }
            test_step(sc, rec_c);
        }
    }
    const unsigned long long wilds = __builtin_amdgcn_ballot_w64(wild);
    if (wilds) sweep(nullptr, p.num_ngroups * NGROUP, nullptr, 0, wilds);
    drain(true);

    if (lane == 0) {
        add_counter(p.counters, C_CH_CAND, n_cand);
        add_counter(p.counters, C_CH_SLOW, n_slow);
    }
    if (!valid) return;
    const unsigned long long key = best[lane];
    if (key != BEST_NONE) atomicMin(&p.best[sl], key);
}

// ---------------------------------------------------------------------------
// Closest-hit traversal of arbitrary rays over cells in space (rt_voxel.h): the well-conditioned
// pairs of a bounce pass -- the ill-conditioned ones are the band sweep's.  One ray per lane,
// wavefronts independent, no LDS.  A lane walks the cells along its ray's line front to back; per
// step of the loop it either moves to the next cell or puts one triangle of the current cell's list
// through the general edge filter.  A triangle that passes becomes the lane's candidate; candidates
// go through the reference arithmetic together -- when VOXEL_FLUSH_MIN lanes hold one, or a lane needs
// its slot for the next -- each lane on its own ray, keeping the smallest (t, index).  The walk of a
// lane ends when it leaves the grid or no cell further on can hold a hit with a computed t at or
// below the lane's best (rtv::walk_done).  Results merge into p.best by the 64-bit atomicMin.
// A walk starts from the best hit recorded so far in p.best (whoever found it: any accepted hit bounds
// the answer).
// Rays outside the band records' range (job.g <= 0) and rays no filter passes (job.g NaN) do not walk.
//
// Walks differ a lot in length -- most rays hit something after a few cells, a ray that leaves the
// scene crosses the whole grid -- and a walk is a chain of dependent loads: a lane needs about a
// microsecond per step, so one long walk holds its wavefront, and the launch, for hundreds of
// microseconds.  A lane therefore gives a walk up after p.walk_budget steps of the loop and appends it, as
// (slot, parameter at which the cell it is in was entered), to a list that traverse_voxel_long_kernel
// takes up: one WAVEFRONT per ray there, 64 cells and 64 triangles at a time.
// grid = VOXEL_WAVES / 4 workgroups of four independent wavefronts (fewer for short lists).
// ---------------------------------------------------------------------------
constexpr int VOXEL_FLUSH_MIN = 16;
constexpr int VOXEL_BATCH = rtv::LIST_ALIGN; // triangles of a cell's list a lane takes per step of its loop
constexpr int VOXEL_REFILL_MIN = 32; // free lanes that make a wavefront take new rays (16 / 24 / 32 / 48: 0.398 / 0.395 /
                                     // 0.380 / 0.406 ms a launch; rays handed out by the free lane instead of 64 at a
                                     // time: 0.57 -- neighbours in the list read the same cells, apart they do not)
constexpr int VOXEL_WAVES = 4096;    // wavefronts of a launch of the walk (four per SIMD)

__global__ __launch_bounds__(BLOCK) void traverse_voxel_kernel(TraverseParams p)
{
    constexpr int B = VOXEL_BATCH;
    static_assert(B == 4, "a group of list entries is one int4");
    const int tid = threadIdx.x;
    const int n = p.count ? *p.count : p.n;
    // The wavefront's rays: a run of consecutive slots, [nxt, c1).  Walks differ in length by an order of
    // magnitude and a wavefront issues every instruction for as long as one lane needs it, so a lane whose
    // walk has ended takes the next ray of the run (when VOXEL_REFILL_MIN lanes are free: taking a ray
    // costs the wavefront a start-up sequence).  The launch has a fixed number of wavefronts; the runs are
    // as long as the list needs.
    // (the rays are handed out 64 at a time from a counter shared by the launch: equal runs per wavefront end
    // at very different times -- the mean wavefront lived 0.6 of the launch -- because walks differ so much.
    // A list that gives the launch's wavefronts at most 64 rays each is dealt out once and for all: more
    // wavefronts with a ray per lane finish sooner than fewer that take a second helping.)
    int nxt = 0, c1 = 0;
    bool exhausted = false;
    auto grab = [&]() {
        int base = 0;
        if ((tid & 63) == 0) base = atomicAdd(p.walk_next, 64);
        base = __builtin_amdgcn_readfirstlane(base);
        exhausted = base >= n;
        nxt = exhausted ? n : base;
        c1 = min(nxt + 64, n);
    };
    const int n_waves = (int)gridDim.x * (BLOCK / 64);
    if (n > 64 * n_waves) {
        grab();
    } else {
        nxt = (blockIdx.x * (BLOCK / 64) + (tid >> 6)) * 64;
        c1 = min(nxt + 64, n);
        exhausted = true;
    }
    if (nxt >= n) return; // the whole wavefront
    const int budget = p.walk_budget > 0 ? p.walk_budget : (n >= (1 << 18) ? 96 : 32);
    const rtv::Grid &G = p.vox;

    // per-lane state of the ray in hand
    bool have = false;   // the lane holds a ray
    bool active = false; // ... and its walk is under way
    bool gave_up = false; // ... or was cut short by the budget
    int sl = 0, own_steps = 0;
    float ox = 0.0f, oy = 0.0f, oz = 0.0f, dx = 0.0f, dy = 0.0f, dz = 0.0f, hx = 0.0f, hy = 0.0f, hz = 0.0f;
    float tbest = INFINITY, tfirst = INFINITY, inv_dn = 0.0f;
    int idbest = 0, idfirst = 0;
    rtv::Walk w{};
    // A walk is latency-bound -- every step depends on loads that depend on the step before.  So a lane
    // takes B triangles per step of its loop, and the loads are staggered so that a step waits for one
    // round trip only:
    //   * the walk itself stands one cell ahead of the list being read: the bounds of the next cell's list
    //     (n_idx, n_end) are requested when the current cell is entered and have arrived when it is left;
    //   * the indices of the NEXT batch of triangles are requested together with the records of the
    //     current one.
    // [idx, end) is what is left of the current cell's list after the batch in `tri`; `enter` is the
    // parameter at which the cell of that batch was entered.
    unsigned int idx = 0, end = 0, n_idx = 0, n_end = 0;
    bool more = false;
    float n_enter = 0.0f, enter = 0.0f;
    auto look_ahead = [&]() {
        more = rtv::walk_step(G, ox, oy, oz, w);
        n_enter = w.t_enter;
        if (more) {
            // (the list of what the cell holds beyond the cell the walk comes from: rt_voxel.h)
            const int at = rtv::list_index(G, w);
            n_idx = p.vox_off[at];
            n_end = p.vox_off[at + 1];
        }
    };
    int tri[B];
    bool has[B];
    // requests the next batch: the rest of the current cell's list, or the start of the next cell's
    auto next_batch = [&]() {
        if (active && idx >= end) {
            if (!more) {
                active = false;
            } else {
                idx = n_idx;
                end = n_end;
                enter = n_enter;
                look_ahead();
            }
        }
        // the next four entries of the list (lists are not padded: the load may run past the list's end, into
        // the next list or the four spare entries behind the last)
        struct __attribute__((packed, aligned(4))) Four { int v[4]; };
        Four grp{{-1, -1, -1, -1}};
        if (active && idx < end) grp = *reinterpret_cast<const Four *>(p.vox_entries + idx);
#pragma unroll
        for (int k = 0; k < B; k++) {
            tri[k] = grp.v[k];
            has[k] = idx + (unsigned int)k < end;
        }
        idx += B;
    };
#pragma unroll
    for (int k = 0; k < B; k++) { tri[k] = -1; has[k] = false; }
    int cand = -1;
    unsigned long long n_cand = 0, n_slow = 0, n_rays = 0;
#ifdef RT_WALK_STATS
    unsigned long long st_a = 0, st_b = 0;
#endif

    // the reference arithmetic on every lane's candidate
    auto flush = [&]() {
        n_cand += __popcll(__builtin_amdgcn_ballot_w64(cand >= 0));
        if (cand >= 0) {
            const v3 X = vpos(p.verts, 3 * cand), Y = vpos(p.verts, 3 * cand + 1), Z = vpos(p.verts, 3 * cand + 2);
            float t;
            if (rtx::ray_vs_triangle(rtx::V(ox, oy, oz), rtx::V(dx, dy, dz), X, Y, Z, t) && t > 0.0f &&
                (t < tbest || (t == tbest && cand < idbest))) {
                tbest = t;
                idbest = cand;
            }
        }
        cand = -1;
    };

    for (;;) {
        // ---- lanes whose walk has ended hand in their ray and take the next one of the run
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(!active);
        if (nxt >= c1 && !exhausted && (idle == ~0ull || __popcll(idle) >= VOXEL_REFILL_MIN)) grab();
        if (idle == ~0ull || (nxt < c1 && __popcll(idle) >= VOXEL_REFILL_MIN)) {
            if (any64(!active && cand >= 0)) flush();
            if (have && !active) {
                const bool spill = gave_up; // out of budget: the walk goes on the list, from the cell of the batch
                const unsigned long long left = __builtin_amdgcn_ballot_w64(spill); // that was next
                if (left) {
                    int base = 0;
                    if ((tid & 63) == __builtin_ctzll(left)) base = atomicAdd(p.walk_count, __popcll(left));
                    base = __shfl(base, __builtin_ctzll(left), 64);
                    if (spill) {
                        const int pos = base + lanes_below(left);
                        p.walk_slot[pos] = sl;
                        p.walk_t[pos] = enter;
                    }
                }
                if (tbest < tfirst || (tbest == tfirst && idbest < idfirst)) {
                    const unsigned long long key =
                        ((unsigned long long)__float_as_uint(tbest) << 32) | (unsigned long long)(unsigned int)idbest;
                    atomicMin(&p.best[sl], key);
                }
                have = false;
            }
            const unsigned long long free_lanes = __builtin_amdgcn_ballot_w64(!have);
            const int mine = nxt + lanes_below(free_lanes);
            const bool fresh = !have && mine < c1;
            nxt = min(nxt + __popcll(free_lanes), c1);
            if (fresh) {
                sl = p.alist ? p.alist[mine] : mine;
                ox = p.job.ox[sl]; oy = p.job.oy[sl]; oz = p.job.oz[sl];
                dx = p.job.dx[sl]; dy = p.job.dy[sl]; dz = p.job.dz[sl];
                const float g = p.job.g[sl];
                hx = dx * g; hy = dy * g; hz = dz * g;
                const unsigned long long key = p.best[sl];
                tbest = tfirst = __uint_as_float((unsigned int)(key >> 32));
                idbest = idfirst = (int)(unsigned int)(key & 0xffffffffull);
                have = true;
                gave_up = false;
                own_steps = 0;
                cand = -1;
                idx = end = 0;
                more = false;
                // rays outside the band records' range (g <= 0) and rays no filter passes (g NaN) do not walk
                active = g > 0.0f && rtv::walk_start(G, ox, oy, oz, dx, dy, dz, w);
                inv_dn = w.inv_dn;
                if (active) {
                    const int at = rtv::list_index(G, w); // (the first cell: its whole list)
                    idx = p.vox_off[at];
                    end = p.vox_off[at + 1];
                    enter = w.t_enter;
                    look_ahead();
                }
                next_batch();
            }
            n_rays += __popcll(__builtin_amdgcn_ballot_w64(fresh));
            if (!any64(have)) break;
            if (!any64(active)) continue; // rays that do not walk: hand them in, take more
        }
        n_slow++;
        if (active) own_steps++;
        if (__popcll(__builtin_amdgcn_ballot_w64(cand >= 0)) >= VOXEL_FLUSH_MIN) flush();
        // nothing in the batch's cell or beyond can matter any more: the walk ends
        if (active && rtv::walk_done_at(G, inv_dn, enter, tbest)) active = false;
        // the records of this batch (a lane without a k-th triangle, or with one it has seen before, reads
        // record 0: one cache line for all such lanes) ...
        int cur[B];
        bool cur_has[B];
        float4 r0[B], r1[B], r2[B];
#pragma unroll
        for (int k = 0; k < B; k++) {
            cur[k] = tri[k];
            cur_has[k] = has[k] && active;
            const float4 *rp = reinterpret_cast<const float4 *>(p.recs + (cur_has[k] ? cur[k] : 0));
            r0[k] = rp[0]; r1[k] = rp[1]; r2[k] = rp[2];
#ifdef RT_WALK_STATS
            st_b += __popcll(__builtin_amdgcn_ballot_w64(RT_WALK_STATS == 1 ? cur_has[k] : (has[k] && active)));
#endif
        }
#ifdef RT_WALK_STATS
        st_a += RT_WALK_STATS == 1 ? __popcll(__builtin_amdgcn_ballot_w64(active)) : 64;
#endif
        // ... and, in the same round trip, the indices of the next
        next_batch();
        bool pass[B];
#pragma unroll
        for (int k = 0; k < B; k++) {
            const float rec[12] = {r0[k].x, r0[k].y, r0[k].z, r0[k].w, r1[k].x, r1[k].y, r1[k].z, r1[k].w,
                                   r2[k].x, r2[k].y, r2[k].z, r2[k].w};
            // (the triangle that holds the best hit, or is about to be tested, shows up again in the cells next to it)
            pass[k] = cur_has[k] && (cur[k] != idbest || !(tbest < INFINITY)) && cur[k] != cand &&
                      rtf::filter_general(rec, ox, oy, oz, hx, hy, hz) >= -1.0f;
        }
#pragma unroll
        for (int k = 0; k < B; k++) {
            if (any64(pass[k] && cand >= 0)) flush(); // a lane needs its slot
            if (pass[k] && cur[k] != cand) cand = cur[k];
        }
        // out of budget: the ray is handed to the kernel that gives each long walk a wavefront
        if (active && own_steps >= budget) {
            active = false;
            gave_up = true;
        }
    }

    if ((tid & 63) == 0) {
        add_counter(p.counters, C_CH_CAND, n_cand);
        add_counter(p.counters, C_CH_SLOW, n_slow);
        add_counter(p.counters, C_CH_PAIRS, n_rays * (unsigned long long)p.num_tris);
#ifdef RT_WALK_STATS
        add_counter(p.counters, 10, st_a);
        add_counter(p.counters, 11, st_b);
#endif
    }
}

// ---------------------------------------------------------------------------
// The long walks traverse_voxel_kernel hands over: one wavefront per ray.  Every lane steps the same
// walk (rt_voxel.h) 64 cells ahead and lane k keeps cell k; the 64 list bounds arrive in one round trip;
// the lists, laid end to end, are then taken 64 triangles at a time, one per lane -- index, record,
// general edge filter, and for the few that pass the reference arithmetic on the spot.  The
// wavefront's best (t, index) is the minimum over its lanes; the stopping rule of rtv::walk_done is
// applied per cell, with the best known when the cell's batch starts.  The best hit so far comes from
// p.best (the per-lane kernel wrote what it had found) and goes back there.
// grid = any number of workgroups of four independent wavefronts, each taking every W-th ray.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void traverse_voxel_long_kernel(TraverseParams p)
{
    __shared__ int s_first_all[BLOCK / 64][65];           // position of each cell's list in the concatenation
    __shared__ unsigned int s_start_all[BLOCK / 64][64];  // where each list starts in vox_entries
    __shared__ float s_enter_all[BLOCK / 64][64];         // parameter at which each cell is entered
    __shared__ int s_rcell_all[BLOCK / 64][65];           // the walk's next 65 states: cell ...
    __shared__ float s_renter_all[BLOCK / 64][65];        // ... and the parameter at which it is entered
    __shared__ int s_next_all[BLOCK / 64][4];             // cell indices of the last of them
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int *s_rcell = s_rcell_all[wave], *s_next = s_next_all[wave];
    float *s_renter = s_renter_all[wave];
    int *s_first = s_first_all[wave];
    unsigned int *s_start = s_start_all[wave];
    float *s_enter = s_enter_all[wave];
    const int n = *p.walk_count;
    // (the per-lane walk before this launch has finished: its hand-out counter returns to zero for its next launch)
    if (blockIdx.x == 0 && threadIdx.x == 0) *p.walk_next = 0;
    const rtv::Grid &G = p.vox;
    const int n_waves = (int)gridDim.x * (BLOCK / 64);
    unsigned long long n_cand = 0, n_slow = 0;
    for (int r = blockIdx.x * (BLOCK / 64) + wave; r < n; r += n_waves) {
        const int sl = p.walk_slot[r];
        const float t_from = p.walk_t[r];
        const float ox = p.job.ox[sl], oy = p.job.oy[sl], oz = p.job.oz[sl];
        const float dx = p.job.dx[sl], dy = p.job.dy[sl], dz = p.job.dz[sl];
        const float g = p.job.g[sl];
        const float hx = dx * g, hy = dy * g, hz = dz * g;
        unsigned long long best = p.best[sl]; // (t bits, index): the same for every lane
        const unsigned long long first_best = best;
        rtv::Walk w{};
        bool alive = rtv::walk_start(G, ox, oy, oz, dx, dy, dz, w, t_from);
        bool walk_over = false;
        while (alive) {
            // the next 64 cells of the walk: lane k keeps the k-th.  Found by rank, not by stepping (rt_voxel.h,
            // "The cells of a walk by rank"): lane L ranks event L of each axis among all the events; the event
            // of rank r puts the walk into its state r + 1, which lane r + 1 takes from the table.
            rtv::AxisSeq sx, sy, sz;
            rtv::walk_sequences(G, ox, oy, oz, dx, dy, dz, w, sx, sy, sz);
            if (lane == 0) {
                s_rcell[0] = rtv::cell_index(G, w);
                s_renter[0] = w.t_enter;
            }
            float te;
            int cb, cc;
            const int rx = rtv::event_rank(G, sx, lane, sy, false, sz, false, te, cb, cc);
            if (rx < 64) {
                const int ix = rtv::seq_index(sx, lane + 1), iy = rtv::seq_index(sy, cb), iz = rtv::seq_index(sz, cc);
                s_rcell[rx + 1] = (iz * G.ny + iy) * G.nx + ix;
                s_renter[rx + 1] = te;
                if (rx == 63) { s_next[0] = ix; s_next[1] = iy; s_next[2] = iz; }
            }
            const int ry = rtv::event_rank(G, sy, lane, sx, true, sz, false, te, cb, cc);
            if (ry < 64) {
                const int ix = rtv::seq_index(sx, cb), iy = rtv::seq_index(sy, lane + 1), iz = rtv::seq_index(sz, cc);
                s_rcell[ry + 1] = (iz * G.ny + iy) * G.nx + ix;
                s_renter[ry + 1] = te;
                if (ry == 63) { s_next[0] = ix; s_next[1] = iy; s_next[2] = iz; }
            }
            const int rz = rtv::event_rank(G, sz, lane, sx, true, sy, true, te, cb, cc);
            if (rz < 64) {
                const int ix = rtv::seq_index(sx, cb), iy = rtv::seq_index(sy, cc), iz = rtv::seq_index(sz, lane + 1);
                s_rcell[rz + 1] = (iz * G.ny + iy) * G.nx + ix;
                s_renter[rz + 1] = te;
                if (rz == 63) { s_next[0] = ix; s_next[1] = iy; s_next[2] = iz; }
            }
            // the step that leaves the grid: the first of the three axes' exit events (each held by one lane)
            const int ex = rtv::seq_exit_event(sx), ey = rtv::seq_exit_event(sy), ez = rtv::seq_exit_event(sz);
            int r_exit = 1 << 20;
            if (ex < 64) r_exit = min(r_exit, __shfl(rx, ex, 64));
            if (ey < 64) r_exit = min(r_exit, __shfl(ry, ey, 64));
            if (ez < 64) r_exit = min(r_exit, __shfl(rz, ez, 64));
            if (r_exit < 64) walk_over = true;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the other lanes read these
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // (states beyond the step that leaves the grid do not exist)
            const int my_cell = lane <= r_exit ? s_rcell[lane] : -1;
            const float my_enter = lane <= r_exit ? s_renter[lane] : INFINITY;
            if (!walk_over) { // the state after 64 steps: where the next round starts
                w.ix = s_next[0]; w.iy = s_next[1]; w.iz = s_next[2];
                w.t_enter = s_renter[64];
                sx.i0 = w.ix; sy.i0 = w.iy; sz.i0 = w.iz;
                w.tx = rtv::seq_t(G, sx, 0); w.ty = rtv::seq_t(G, sy, 0); w.tz = rtv::seq_t(G, sz, 0);
            }
            unsigned int start = 0;
            int cnt = 0;
            if (my_cell >= 0) {
                // (every triangle of the cell: this kernel does not keep track of where a walk came from)
                start = p.vox_off[my_cell * rtv::CELL_STRIDE + rtv::LIST_START];
                cnt = (int)(p.vox_off[my_cell * rtv::CELL_STRIDE + rtv::LIST_START + 1] - start);
            }
            // exclusive prefix sum of the list lengths over the lanes
            int incl = cnt;
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) {
                const int up = __shfl_up(incl, sh, 64);
                if (lane >= sh) incl += up;
            }
            const int total = __builtin_amdgcn_readlane(incl, 63);
            s_first[lane] = incl - cnt;
            if (lane == 63) s_first[64] = total;
            s_start[lane] = start;
            s_enter[lane] = my_enter;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the other lanes read these
            __builtin_amdgcn_wave_barrier();
            for (int base = 0; base < total && alive; base += 64) {
                n_slow++;
                const int e = base + lane;
                const bool has = e < total;
                // the cell whose list holds entry e: the last k with s_first[k] <= e
                int c = 0;
#pragma unroll
                for (int sh = 32; sh > 0; sh >>= 1)
                    if (s_first[c + sh] <= e) c += sh;
                const float tb = __uint_as_float((unsigned int)(best >> 32));
                // the batch starts in cell c of lane 0: nothing from there on can matter -> the walk ends
                const int c0 = __builtin_amdgcn_readfirstlane(c);
                if (rtv::walk_done_at(G, w.inv_dn, s_enter[c0], tb)) {
                    alive = false;
                    break;
                }
                unsigned long long mine = ~0ull;
                if (has && !rtv::walk_done_at(G, w.inv_dn, s_enter[c], tb)) {
                    const int tri = p.vox_entries[s_start[c] + (unsigned int)(e - s_first[c])];
                    const float4 *rp = reinterpret_cast<const float4 *>(p.recs + (tri < 0 ? 0 : tri));
                    const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
                    const float rec[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
                    if (tri >= 0 && rtf::filter_general(rec, ox, oy, oz, hx, hy, hz) >= -1.0f) {
                        n_cand++;
                        const v3 X = vpos(p.verts, 3 * tri), Y = vpos(p.verts, 3 * tri + 1), Z = vpos(p.verts, 3 * tri + 2);
                        float t;
                        if (rtx::ray_vs_triangle(rtx::V(ox, oy, oz), rtx::V(dx, dy, dz), X, Y, Z, t) && t > 0.0f)
                            mine = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(unsigned int)tri;
                    }
                }
                if (any64(mine != ~0ull)) {
                    // minimum of (t bits, index) over the lanes: smallest t, lowest index among equal t
#pragma unroll
                    for (int sh = 32; sh > 0; sh >>= 1) {
                        const unsigned long long other = __shfl_xor(mine, sh, 64);
                        mine = other < mine ? other : mine;
                    }
                    if (mine < best) best = mine;
                }
            }
            __builtin_amdgcn_wave_barrier(); // the tables are rewritten by the next round
            if (alive && (walk_over || rtv::walk_done(G, w, __uint_as_float((unsigned int)(best >> 32))))) alive = false;
        }
        if (lane == 0 && best < first_best) atomicMin(&p.best[sl], best);
    }
    if (lane == 0) {
        add_counter(p.counters, C_CH_CAND, n_cand);
        add_counter(p.counters, C_CH_SLOW, n_slow);
    }
}

// ---------------------------------------------------------------------------
// intersect, part 3 of 3: the two spheres (trace.glsl:164-183), the normal of the
// winning primitive, and rayintersect_cs.glsl:16-33.  The triangle normal depends
// only on the winning triangle, so it is formed once here instead of on every
// improvement inside the loop.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void intersect_finish_kernel(RaySoA r, const int *__restrict__ list,
                                                                 const int *__restrict__ count, int n_fixed, Job job,
                                                                 const unsigned long long *__restrict__ best,
                                                                 const rt_vertex *__restrict__ verts, int num_tris,
                                                                 Camera cam, float *__restrict__ depth,
                                                                 unsigned long long *counters, int *zero_this,
                                                                 int *n_out)
{
    const int n = count ? *count : n_fixed;
    if (blockIdx.x == 0 && threadIdx.x == 0) add_counter(counters, C_CH_RAYS, (unsigned long long)n);
    if (n_out && blockIdx.x == 0 && threadIdx.x == 0) *n_out = n; // for the kernel log
    // the list counter of the OTHER stage, which no kernel of this stage call reads: zero for its next use
    if (zero_this && blockIdx.x == 0 && threadIdx.x == 0) *zero_this = 0;
    RT_FOR_EACH_SLOT(slot, n) {
    const int i = list ? list[slot] : slot;
    const v3 o = rtx::V(job.ox[slot], job.oy[slot], job.oz[slot]);
    const v3 d = rtx::V(job.dx[slot], job.dy[slot], job.dz[slot]);
    intersect_finish_slot(r, i, o, d, best[slot], verts, num_tris, cam.pos, depth);
    }
}

// ---------------------------------------------------------------------------
// color  ("OpenGL Raytracer/raycol_cs.glsl":18-128 with trace.glsl:121-184)
//
// Runs over the rays whose origin is finite.  A ray with a non-finite origin (a
// miss: origin + dir * inf) gives lightDir = NaN, d = NaN, so `lightRay.t > d` is
// false for every light and the pixel is vec3(0.1); compact_kernel<1> wrote it.
//
// Per light:  shadow_setup -> { traverse<ANYHIT> over a chunk of triangles -> compact the
// rays that are still unoccluded } x chunks -> shadow_finish (spheres + Phong).  Dropping
// occluded rays between chunks is the early exit: they cost nothing in later chunks.
// The shadow ray's line passes through the light, so the shared-origin filter with the
// light as origin applies; the exact test uses the reference's own origin
// (r.origin + lightDir * 1e-7) and direction.
// ---------------------------------------------------------------------------
// raycol_cs.glsl:39-47
__global__ __launch_bounds__(BLOCK) void shadow_setup_kernel(const int *__restrict__ count, Scratch s, Job job,
                                                             float lx, float ly, float lz, SharedOrigin so,
                                                             unsigned long long *counters, unsigned int *keys,
                                                             unsigned int *hist, int *n_out)
{
    const int n = *count;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        add_counter(counters, C_SH_RAYS, (unsigned long long)n);
        if (n_out) *n_out = n; // for the kernel log
    }
    // (whole wavefronts stay in the loop: the lanes group their keys together)
    for (int base = blockIdx.x * BLOCK; base < n; base += (int)gridDim.x * BLOCK) {
        const int slot = base + (int)threadIdx.x;
        const bool valid = slot < n;
        v3 o = rtx::V(0.0f, 0.0f, 0.0f);
        if (valid) o = rtx::V(s.sox[slot], s.soy[slot], s.soz[slot]);
        shadow_setup_slot(valid, slot, o, s, job, lx, ly, lz, so, keys, hist);
    }
}

// spheres of the shadow trace, then raycol_cs.glsl:50-79 for a ray that reaches the light: false when
// a sphere occludes it, else `add` = what the light contributes to the ray's `light` accumulator
// (the caller has seen to it that no triangle occludes: occ[slot] == 0)
__device__ __forceinline__ bool shadow_finish_slot(const RaySoA &r, int i, int slot, const Scratch &s, const Job &job,
                                                   const rt_light &light, const Camera &cam, v3 &add)
{
    const v3 o2 = rtx::V(job.ox[slot], job.oy[slot], job.oz[slot]);
    const v3 ld = rtx::V(job.dx[slot], job.dy[slot], job.dz[slot]);
    const float dist = job.tmax[slot];
#pragma unroll
    for (int sp = 0; sp < 2; sp++) {
        float ts;
        if (rtx::ray_sphere(o2, ld, rtx::sphere_center(sp), rtx::sphere_radius(sp), ts))
            if (ts > 0.0f && ts <= dist) return false; // occluded by a sphere
    }
    const v3 o = rtx::V(s.sox[slot], s.soy[slot], s.soz[slot]);
    const v3 nn = rtx::V(r.nx[i], r.ny[i], r.nz[i]);
    const v3 lcol = rtx::V(light.color[0], light.color[1], light.color[2]);
    const v3 cpos = rtx::V(cam.pos[0], cam.pos[1], cam.pos[2]);
    const float ndl = rtx::dot(nn, ld);
    const float diffuse = rtx::max0(ndl);
    const v3 v = rtx::normalize(rtx::sub(cpos, o));
    const v3 I = rtx::muls(ld, -1.0f);
    float kk = 0.0f;
    if (ndl > 0.0f) {
        const v3 rr = rtx::normalize(rtx::sub(I, rtx::muls(nn, 2.0f * rtx::dot(nn, I))));
        kk = rtx::pow30(rtx::max0(rtx::dot(v, rr)));
    }
    const float att = 1.0f / (0.5f * dist);
    add = rtx::muls(rtx::muls(lcol, diffuse + kk), att);
    return true;
}

// (every light but the last: the last light's share of this work is done by color_finish_kernel)
__global__ __launch_bounds__(BLOCK) void shadow_finish_kernel(RaySoA r, const int *__restrict__ list0,
                                                              const int *__restrict__ alist,
                                                              const int *__restrict__ acount, int n_fixed, Scratch s,
                                                              Job job, rt_light light, Camera cam)
{
    const int n = acount ? *acount : n_fixed;
    RT_FOR_EACH_SLOT(a, n) {
    const int slot = alist ? alist[a] : a;
    if (s.occ[slot] != 0) continue;
    v3 add;
    if (!shadow_finish_slot(r, list0[slot], slot, s, job, light, cam, add)) continue;
    s.lr[slot] = s.lr[slot] + add.x; s.lg[slot] = s.lg[slot] + add.y; s.lb[slot] = s.lb[slot] + add.z;
    s.lit[slot] = 1;
    }
}

// texture, final colour, kill rule, pixel  (raycol_cs.glsl:82-127)
// (and before that, for the LAST light, what shadow_finish_kernel does for the others: a slot's thread
// adds that light's contribution itself -- one launch and one round trip through memory fewer)
__global__ __launch_bounds__(BLOCK) void color_finish_kernel(RaySoA r, const int *__restrict__ list0,
                                                             const int *__restrict__ count, Scratch s,
                                                             const rt_vertex *__restrict__ verts, int num_vertices,
                                                             const uint32_t *__restrict__ tex, int tex_w, int tex_h,
                                                             uint32_t *__restrict__ fb, int *zero_this, int *n_host,
                                                             Job job, rt_light last_light, Camera cam,
                                                             unsigned long long *counters, int *n_out0)
{
    const int n = *count;
    // the first light's shadow rays were prepared by the compaction, which could not count them
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        add_counter(counters, C_SH_RAYS, (unsigned long long)n);
        if (n_out0) *n_out0 = n; // for the kernel log
    }
    // feedback for the host's choices in later calls (pinned host memory; nobody waits for it)
    if (n_host && blockIdx.x == 0 && threadIdx.x == 0) *n_host = n;
    // the list counter of the OTHER stage, which no kernel of this stage call reads: zero for its next use
    if (zero_this && blockIdx.x == 0 && threadIdx.x == 0) *zero_this = 0;
    unsigned int n_lit = 0, n_kill = 0; // per wavefront
    RT_FOR_EACH_SLOT(slot, n) {
    const int i = list0[slot];
    bool kill = false;
    v3 fin = rtx::V(0.1f, 0.1f, 0.1f);
    float lr = s.lr[slot], lg = s.lg[slot], lb = s.lb[slot];
    bool lit = s.lit[slot] != 0;
    if (s.occ[slot] == 0) { // the last light's shadow ray met no triangle
        v3 add;
        if (shadow_finish_slot(r, i, slot, s, job, last_light, cam, add)) {
            lr = lr + add.x; lg = lg + add.y; lb = lb + add.z;
            lit = true;
        }
    }
    if (lit) {
        const int id = r.id[i];
        v3 pa = rtx::V(0.0f, 0.0f, 0.0f), pb = pa, pc = pa;
        float ta0 = 0.0f, ta1 = 0.0f, tb0 = 0.0f, tb1 = 0.0f, tc0 = 0.0f, tc1 = 0.0f;
        // robust buffer access: an index outside the buffer reads zeros
        const long long i0 = (long long)id * 3;
        if (i0 >= 0 && i0 < num_vertices) { const rt_vertex vv = verts[i0]; pa = rtx::V(vv.pos[0], vv.pos[1], vv.pos[2]); ta0 = vv.texCoord[0]; ta1 = vv.texCoord[1]; }
        if (i0 + 1 >= 0 && i0 + 1 < num_vertices) { const rt_vertex vv = verts[i0 + 1]; pb = rtx::V(vv.pos[0], vv.pos[1], vv.pos[2]); tb0 = vv.texCoord[0]; tb1 = vv.texCoord[1]; }
        if (i0 + 2 >= 0 && i0 + 2 < num_vertices) { const rt_vertex vv = verts[i0 + 2]; pc = rtx::V(vv.pos[0], vv.pos[1], vv.pos[2]); tc0 = vv.texCoord[0]; tc1 = vv.texCoord[1]; }
        const v3 o = rtx::V(s.sox[slot], s.soy[slot], s.soz[slot]);
        const v3 v0 = rtx::sub(pb, pa), v1 = rtx::sub(pc, pa), v2 = rtx::sub(o, pa);
        const float d00 = rtx::dot(v0, v0), d01 = rtx::dot(v0, v1), d11 = rtx::dot(v1, v1);
        const float d20 = rtx::dot(v2, v0), d21 = rtx::dot(v2, v1);
        const float denom = d00 * d11 - d01 * d01;
        const float bv = (d11 * d20 - d01 * d21) / denom;
        const float bw = (d00 * d21 - d01 * d20) / denom;
        const float bu = 1.0f - bv - bw;
        const float uvx = bu * ta0 + bv * tb0 + bw * tc0;
        const float uvy = bu * ta1 + bv * tb1 + bw * tc1;
        const float fx = uvx * (float)tex_w, fy = uvy * (float)tex_h;
        v3 texc = rtx::V(0.0f, 0.0f, 0.0f);
        if (tex && fx > -1.0f && fx < (float)tex_w && fy > -1.0f && fy < (float)tex_h) {
            const uint32_t tx = tex[(size_t)((int)fy) * tex_w + (int)fx];
            texc = rtx::V((float)(tx & 255u) / 255.0f, (float)((tx >> 8) & 255u) / 255.0f,
                          (float)((tx >> 16) & 255u) / 255.0f);
        }
        fin = rtx::add(texc, rtx::muls(rtx::V(lr, lg, lb), 0.3f));
        if (id + 3 * 5 - 2 < num_vertices / 3) {
            r.id[i] = -2;
            r.ox[i] = o.x; r.oy[i] = o.y; r.oz[i] = o.z;
            kill = true;
        }
    }
    fb[i] = rtx::to_unorm8(fin.x) | (rtx::to_unorm8(fin.y) << 8) | (rtx::to_unorm8(fin.z) << 16) | 0xff000000u;
    n_lit += (unsigned int)__popcll(__builtin_amdgcn_ballot_w64(lit));
    n_kill += (unsigned int)__popcll(__builtin_amdgcn_ballot_w64(kill));
    }
    // (the first lane of a wavefront takes part in every round of the loop that any of its lanes does)
    if ((threadIdx.x & 63) == 0) {
        add_counter(counters, C_LIT, n_lit);
        add_counter(counters, C_KILLS, n_kill);
    }
}

} // namespace rtk
