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
enum { C_CH_RAYS = 0, C_SH_RAYS, C_CH_PAIRS, C_SH_PAIRS, C_CH_CAND, C_SH_CAND };

__device__ __forceinline__ void add_counter(unsigned long long *c, int which, unsigned long long v)
{
    if (v) atomicAdd(&c[which], v);
}

// ---------------------------------------------------------------------------
// raygen  ("OpenGL Raytracer/raygen_cs.glsl":1-22)
// One thread writes four consecutive rays with 16-byte stores: 56 bytes per ray,
// nothing read but the row map.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void raygen_kernel(RaySoA r, int n_rays, int width, int height,
                                                       const int *__restrict__ row_map, Camera cam)
{
    const int i0 = (blockIdx.x * BLOCK + threadIdx.x) * 4;
    if (i0 >= n_rays) return;
    const float hw = (float)width / 2.0f;
    const float hh = (float)height / 2.0f;
    const v3 cpos = rtx::V(cam.pos[0], cam.pos[1], cam.pos[2]);
    const v3 cdir = rtx::V(cam.dir[0], cam.dir[1], cam.dir[2]);
    const v3 cup = rtx::V(cam.up[0], cam.up[1], cam.up[2]);
    const v3 cright = rtx::V(cam.right[0], cam.right[1], cam.right[2]);
    const v3 base = rtx::add(cpos, rtx::muls(cdir, 1.0f));

    float ddx[4], ddy[4], ddz[4];
    int j = i0 / width;
    int x = i0 - j * width;
    int y = row_map[j];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        float fx = ((float)x - hw) / hw;
        float fy = ((float)y - hh) / hh;
        v3 s = rtx::add(rtx::add(base, rtx::muls(cright, fx)), rtx::muls(cup, fy));
        v3 d = rtx::normalize(rtx::sub(s, cpos));
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

__global__ __launch_bounds__(BLOCK) void prep_origin_kernel(const rt_vertex *__restrict__ v, int num_tris,
                                                            float ox, float oy, float oz,
                                                            rtf::TriRec *__restrict__ out)
{
    const int k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= num_tris) return;
    const float O[3] = {ox, oy, oz};
    rtf::TriRec r;
    rtf::make_origin_rec(v[3 * k].pos, v[3 * k + 1].pos, v[3 * k + 2].pos, O, &r);
    out[k] = r;
}

// ---------------------------------------------------------------------------
// stream compaction: list[] = indices i in [0,n) with pred(i), block segments in
// arbitrary order, index order inside a segment of 2048 rays.
//   MODE 0: live rays            (primitiveID != -2)     -- before intersect
//   MODE 1: rays with a finite origin                    -- before shading; the
//           others are reference-grey (see shade kernel) and get their pixel here
// ---------------------------------------------------------------------------
constexpr int COMPACT_ITERS = 8;

template <int MODE>
__global__ __launch_bounds__(BLOCK) void compact_kernel(RaySoA r, int n, int *__restrict__ list,
                                                        int *__restrict__ count, uint32_t *__restrict__ fb)
{
    __shared__ int s_cnt[COMPACT_ITERS][BLOCK / 64];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x * (BLOCK * COMPACT_ITERS);
    unsigned long long masks[COMPACT_ITERS];
#pragma unroll
    for (int it = 0; it < COMPACT_ITERS; it++) {
        const int i = chunk + it * BLOCK + tid;
        bool p = false;
        if (i < n) {
            if (MODE == 0) {
                p = r.id[i] != -2;
            } else {
                float a = r.ox[i], b = r.oy[i], c = r.oz[i];
                // finite <=> |x| < inf (false for NaN)
                p = (fabsf(a) < INFINITY) && (fabsf(b) < INFINITY) && (fabsf(c) < INFINITY);
                if (!p) fb[i] = 0xff1a1a1au; // vec3(0.1) -> 26, alpha 255
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
        s_base = tot ? atomicAdd(count, tot) : 0;
    }
    __syncthreads();
    const int base = s_base;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int it = 0; it < COMPACT_ITERS; it++) {
        const unsigned long long m = masks[it];
        if ((m >> lane) & 1ull) list[base + s_cnt[it][wave] + __popcll(m & lt)] = chunk + it * BLOCK + tid;
    }
}

__global__ void fill_grey_kernel(uint32_t *__restrict__ fb, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fb[i] = 0xff1a1a1au;
}

// ---------------------------------------------------------------------------
// shared pieces of the two traversal kernels
// ---------------------------------------------------------------------------
__device__ __forceinline__ v3 vpos(const rt_vertex *__restrict__ v, int i)
{
    const float4 p = *reinterpret_cast<const float4 *>(v + i);
    return rtx::V(p.x, p.y, p.z);
}

// cooperative copy of one tile of filter records into LDS; records past the end
// of the mesh are NaN (no ray passes them)
__device__ __forceinline__ void load_tile(float4 *tile4, const rtf::TriRec *__restrict__ recs, int tile_base,
                                          int num_tris, int tid)
{
    const float4 *src = reinterpret_cast<const float4 *>(recs) + (size_t)tile_base * 3;
    const int avail = (num_tris - tile_base) * 3;
    const float qn = __int_as_float(0x7fc00000);
#pragma unroll
    for (int i = 0; i < (TILE * 3) / BLOCK; i++) {
        const int e = i * BLOCK + tid;
        tile4[e] = (e < avail) ? src[e] : make_float4(qn, qn, qn, qn);
    }
}

struct TraceParams {
    RaySoA rays;
    const int *list;   // ray indices, or nullptr for the identity
    const int *count;  // number of list entries (device), or nullptr -> n
    int n;             // number of rays when count == nullptr
    const rtf::TriRec *recs;
    const rt_vertex *verts;
    int num_tris;
    float center[3];   // bounding sphere of the mesh
    float radius;
    float origin[3];   // shared origin (FORM 1)
    float cam_pos[3];
    float *depth;
    unsigned long long *counters;
};

// ---------------------------------------------------------------------------
// intersect  ("OpenGL Raytracer/rayintersect_cs.glsl":1-34 with trace.glsl:121-184)
//
// K rays per lane, 256 K rays per workgroup.  FORM 0: any rays; FORM 1: every
// ray's line passes through p.origin (rays fresh from raygen).  EXACT: evaluate
// the reference arithmetic on every pair (validation mode).
//
// Order of acceptance: a lane walks the triangles in index order and accepts with
// the reference's strict (t > 0 && t < best), so ties keep the lowest index, as the
// sequential loop trace.glsl:135-157 does; the two spheres follow (:164-183).
// The triangle normal depends only on the winning triangle, so it is formed once
// after the loop instead of on every improvement.
// ---------------------------------------------------------------------------
template <int K, int FORM, bool EXACT>
__global__ __launch_bounds__(BLOCK) void closest_hit_kernel(TraceParams p)
{
    __shared__ float4 tile4[TILE * 3];
    __shared__ float xr[6][K][BLOCK]; // exact (offset origin, direction) per ray; column tid is private to thread tid

    const int tid = threadIdx.x;
    const int n = p.count ? *p.count : p.n;
    const int slot0 = blockIdx.x * (BLOCK * K);
    if (slot0 >= n) return; // whole workgroup leaves together

    int ridx[K];
    bool valid[K];
    float ox[K], oy[K], oz[K], hx[K], hy[K], hz[K], tbest[K];
    int idbest[K];

#pragma unroll
    for (int k = 0; k < K; k++) {
        const int slot = slot0 + k * BLOCK + tid;
        valid[k] = slot < n;
        ridx[k] = valid[k] ? (p.list ? p.list[slot] : slot) : 0;
        const float qn = __int_as_float(0x7fc00000);
        v3 o = rtx::V(qn, qn, qn), d = o;
        if (valid[k]) {
            o = rtx::V(p.rays.ox[ridx[k]], p.rays.oy[ridx[k]], p.rays.oz[ridx[k]]);
            d = rtx::V(p.rays.dx[ridx[k]], p.rays.dy[ridx[k]], p.rays.dz[ridx[k]]);
        }
        // r.origin = r.origin + r.dir * 0.001   (rayintersect_cs.glsl:12)
        o = rtx::add(o, rtx::muls(d, 0.001f));
        xr[0][k][tid] = o.x; xr[1][k][tid] = o.y; xr[2][k][tid] = o.z;
        xr[3][k][tid] = d.x; xr[4][k][tid] = d.y; xr[5][k][tid] = d.z;
        float reach;
        if (FORM == 0) {
            v3 oc = rtx::V(o.x - p.center[0], o.y - p.center[1], o.z - p.center[2]);
            reach = sqrtf(rtx::dot(oc, oc)) + 3.0f * p.radius;
        } else {
            v3 O = rtx::V(p.origin[0], p.origin[1], p.origin[2]);
            v3 C = rtx::V(p.center[0], p.center[1], p.center[2]);
            reach = sqrtf(rtx::dot(O, O)) + sqrtf(rtx::dot(o, o)) + sqrtf(rtx::dot(C, C)) + 3.0f * p.radius;
        }
        const float sc = rtf::dir_scale(reach, d.x, d.y, d.z);
        ox[k] = o.x; oy[k] = o.y; oz[k] = o.z;
        hx[k] = d.x * sc; hy[k] = d.y * sc; hz[k] = d.z * sc;
        tbest[k] = INFINITY; // r.t = 1.0 / 0.0   (trace.glsl:123)
        idbest[k] = 0;       // r.primitiveID = 0 (trace.glsl:133)
    }

    unsigned long long n_cand = 0;
    const float *tile = reinterpret_cast<const float *>(tile4);

    for (int tile_base = 0; tile_base < p.num_tris; tile_base += TILE) {
        __syncthreads();
        load_tile(tile4, p.recs, tile_base, p.num_tris, tid);
        __syncthreads();
        const int cnt = min(TILE, p.num_tris - tile_base);
        for (int j = 0; j < cnt; j++) {
            const float *rec = tile + j * REC_FLOATS; // wave-uniform address: LDS broadcast
            float g[K];
            float G = -INFINITY;
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (FORM == 0) g[k] = rtf::filter_general(rec, ox[k], oy[k], oz[k], hx[k], hy[k], hz[k]);
                else g[k] = rtf::filter_origin(rec, hx[k], hy[k], hz[k]);
                G = fmaxf(G, g[k]);
            }
            const bool any_pass = EXACT ? true : (G >= -1.0f);
            if (__any(any_pass)) {
                // rare path: the reference's own arithmetic on the pairs that passed
                const int tri = tile_base + j;
                const v3 X = vpos(p.verts, 3 * tri), Y = vpos(p.verts, 3 * tri + 1), Z = vpos(p.verts, 3 * tri + 2);
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const bool pass = EXACT ? valid[k] : (g[k] >= -1.0f);
                    if (!EXACT) n_cand += __popcll(__ballot(pass));
                    if (pass) {
                        const v3 o = rtx::V(xr[0][k][tid], xr[1][k][tid], xr[2][k][tid]);
                        const v3 d = rtx::V(xr[3][k][tid], xr[4][k][tid], xr[5][k][tid]);
                        float t;
                        if (rtx::ray_vs_triangle(o, d, X, Y, Z, t)) {
                            if (t > 0.0f && t < tbest[k]) {
                                tbest[k] = t;
                                idbest[k] = tri;
                            }
                        }
                    }
                }
            }
        }
    }

    if ((tid & 63) == 0) add_counter(p.counters, C_CH_CAND, n_cand);
    if (tid == 0) {
        const int mine = min(BLOCK * K, n - slot0);
        add_counter(p.counters, C_CH_RAYS, (unsigned long long)mine);
        add_counter(p.counters, C_CH_PAIRS, (unsigned long long)mine * (unsigned long long)p.num_tris);
    }

    // epilogue: spheres, normal, and the rest of rayintersect_cs.glsl:16-33
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (!valid[k]) continue;
        const int i = ridx[k];
        const v3 o = rtx::V(xr[0][k][tid], xr[1][k][tid], xr[2][k][tid]);
        const v3 d = rtx::V(xr[3][k][tid], xr[4][k][tid], xr[5][k][tid]);
        float tb = tbest[k];
        int id = idbest[k];
        int kind = (tb < INFINITY) ? 1 : 0; // 0 none, 1 triangle, 2 sphere
        v3 nrm = rtx::V(0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int s = 0; s < 2; s++) {
            float ts;
            if (rtx::ray_sphere(o, d, rtx::sphere_center(s), rtx::sphere_radius(s), ts)) {
                if (ts > 0.0f && ts < tb) {
                    tb = ts;
                    id = p.num_tris + s;
                    kind = 2;
                    v3 pt = rtx::add(o, rtx::muls(d, tb));
                    nrm = rtx::normalize(rtx::sub(pt, rtx::sphere_center(s)));
                }
            }
        }
        if (kind == 1) {
            const v3 X = vpos(p.verts, 3 * id), Y = vpos(p.verts, 3 * id + 1), Z = vpos(p.verts, 3 * id + 2);
            nrm = rtx::normalize(rtx::cross(rtx::sub(X, Z), rtx::sub(Y, Z)));
        } else if (kind == 0) {
            nrm = rtx::V(p.rays.nx[i], p.rays.ny[i], p.rays.nz[i]); // a miss keeps the old normal
        }
        v3 ob = rtx::sub(o, rtx::muls(d, 0.001f));
        if (isinf(tb)) id = -2;
        ob = rtx::add(ob, rtx::muls(d, tb));
        const v3 dn = rtx::sub(d, rtx::muls(nrm, 2.0f * rtx::dot(d, nrm)));
        p.rays.ox[i] = ob.x; p.rays.oy[i] = ob.y; p.rays.oz[i] = ob.z;
        p.rays.dx[i] = dn.x; p.rays.dy[i] = dn.y; p.rays.dz[i] = dn.z;
        p.rays.t[i] = tb;
        p.rays.id[i] = id;
        if (kind != 0) {
            p.rays.nx[i] = nrm.x; p.rays.ny[i] = nrm.y; p.rays.nz[i] = nrm.z;
            p.rays.cx[i] = 0.0f; p.rays.cy[i] = 0.0f; p.rays.cz[i] = (kind == 2) ? 1.0f : 0.0f;
        }
        p.depth[i] = rtx::distance3(ob, rtx::V(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]));
    }
}

// ---------------------------------------------------------------------------
// color  ("OpenGL Raytracer/raycol_cs.glsl":18-128 with trace.glsl:121-184)
//
// Runs over the rays whose origin is finite.  A ray with a non-finite origin (a
// miss: origin + dir * inf) gives lightDir = NaN, d = NaN, so `lightRay.t > d` is
// false for every light and the pixel is vec3(0.1); compact_kernel<1> wrote it.
//
// The shadow query only uses `lightRay.t > d` (raycol_cs.glsl:50), i.e. "no
// primitive is accepted with 0 < t <= d", so traversal is any-hit with early
// exit.  The shadow ray's line passes through the light, so the shared-origin
// filter with the light as origin applies; the exact test uses the reference's own
// origin (r.origin + lightDir * 1e-7) and direction.
// ---------------------------------------------------------------------------
struct ShadeParams {
    RaySoA rays;
    const int *list;
    const int *count;
    const rtf::TriRec *light_recs; // num_lights arrays of rec_stride records
    size_t rec_stride;
    const rt_vertex *verts;
    int num_vertices;
    int num_tris;
    float center[3];
    float radius;
    float cam_pos[3];
    int num_lights;
    rt_light lights[RT_MAX_LIGHTS];
    const uint32_t *tex;
    int tex_w, tex_h;
    uint32_t *fb;
    unsigned long long *counters;
};

template <int K, bool EXACT>
__global__ __launch_bounds__(BLOCK) void shade_kernel(ShadeParams p)
{
    __shared__ float4 tile4[TILE * 3];
    __shared__ float xr[6][K][BLOCK];

    const int tid = threadIdx.x;
    const int n = *p.count;
    const int slot0 = blockIdx.x * (BLOCK * K);
    if (slot0 >= n) return;
    const int mine = min(BLOCK * K, n - slot0);

    int ridx[K];
    bool valid[K], lit[K];
    float ox[K], oy[K], oz[K], nx[K], ny[K], nz[K];
    float lr[K], lg[K], lb[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int slot = slot0 + k * BLOCK + tid;
        valid[k] = slot < n;
        ridx[k] = valid[k] ? p.list[slot] : 0;
        const float qn = __int_as_float(0x7fc00000);
        ox[k] = oy[k] = oz[k] = qn;
        nx[k] = ny[k] = nz[k] = 0.0f;
        if (valid[k]) {
            ox[k] = p.rays.ox[ridx[k]]; oy[k] = p.rays.oy[ridx[k]]; oz[k] = p.rays.oz[ridx[k]];
            nx[k] = p.rays.nx[ridx[k]]; ny[k] = p.rays.ny[ridx[k]]; nz[k] = p.rays.nz[ridx[k]];
        }
        lit[k] = false;
        lr[k] = lg[k] = lb[k] = 0.0f;
    }

    const float *tile = reinterpret_cast<const float *>(tile4);
    const v3 cpos = rtx::V(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]);
    unsigned long long n_cand = 0, n_pairs = 0;

    for (int a = 0; a < p.num_lights; a++) {
        const v3 lpos = rtx::V(p.lights[a].pos[0], p.lights[a].pos[1], p.lights[a].pos[2]);
        const v3 lcol = rtx::V(p.lights[a].color[0], p.lights[a].color[1], p.lights[a].color[2]);
        float hx[K], hy[K], hz[K], dist[K];
        bool occ[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            v3 o = rtx::V(ox[k], oy[k], oz[k]);
            const v3 ld = rtx::normalize(rtx::sub(lpos, o));
            const v3 o2 = rtx::add(o, rtx::muls(ld, 0.0000001f)); // shadow-ray origin
            o = rtx::sub(o, rtx::muls(ld, 0.0000001f));           // raycol_cs.glsl:45
            ox[k] = o.x; oy[k] = o.y; oz[k] = o.z;
            dist[k] = rtx::distance3(lpos, o);
            xr[0][k][tid] = o2.x; xr[1][k][tid] = o2.y; xr[2][k][tid] = o2.z;
            xr[3][k][tid] = ld.x; xr[4][k][tid] = ld.y; xr[5][k][tid] = ld.z;
            const v3 C = rtx::V(p.center[0], p.center[1], p.center[2]);
            const float reach = sqrtf(rtx::dot(lpos, lpos)) + sqrtf(rtx::dot(o2, o2)) + sqrtf(rtx::dot(C, C)) +
                                3.0f * p.radius;
            const float sc = rtf::dir_scale(reach, ld.x, ld.y, ld.z);
            hx[k] = ld.x * sc; hy[k] = ld.y * sc; hz[k] = ld.z * sc;
            // `lightRay.t > d` can only hold for finite d
            occ[k] = !valid[k] || !(dist[k] < INFINITY);
        }

        const rtf::TriRec *recs = p.light_recs + (size_t)a * p.rec_stride;
        for (int tile_base = 0; tile_base < p.num_tris; tile_base += TILE) {
            bool all_occ = true;
#pragma unroll
            for (int k = 0; k < K; k++) all_occ = all_occ && occ[k];
            // barrier + early exit: leave when every ray of the workgroup is occluded
            if (__syncthreads_and(all_occ ? 1 : 0)) break;
            load_tile(tile4, recs, tile_base, p.num_tris, tid);
            __syncthreads();
            const int cnt = min(TILE, p.num_tris - tile_base);
            n_pairs += (unsigned long long)cnt;
            for (int j = 0; j < cnt; j++) {
                const float *rec = tile + j * REC_FLOATS;
                float g[K];
                float G = -INFINITY;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    g[k] = rtf::filter_origin(rec, hx[k], hy[k], hz[k]);
                    G = fmaxf(G, g[k]);
                }
                const bool any_pass = EXACT ? true : (G >= -1.0f);
                if (__any(any_pass)) {
                    const int tri = tile_base + j;
                    const v3 X = vpos(p.verts, 3 * tri), Y = vpos(p.verts, 3 * tri + 1),
                             Z = vpos(p.verts, 3 * tri + 2);
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        const bool pass = EXACT ? valid[k] : (g[k] >= -1.0f);
                        if (!EXACT) n_cand += __popcll(__ballot(pass));
                        if (pass && !occ[k]) {
                            const v3 o = rtx::V(xr[0][k][tid], xr[1][k][tid], xr[2][k][tid]);
                            const v3 d = rtx::V(xr[3][k][tid], xr[4][k][tid], xr[5][k][tid]);
                            float t;
                            if (rtx::ray_vs_triangle(o, d, X, Y, Z, t)) {
                                if (t > 0.0f && t <= dist[k]) occ[k] = true;
                            }
                        }
                    }
                }
            }
        }

        // spheres, then Phong for the lights that are visible (raycol_cs.glsl:50-79)
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (!valid[k]) continue;
            const v3 o2 = rtx::V(xr[0][k][tid], xr[1][k][tid], xr[2][k][tid]);
            const v3 ld = rtx::V(xr[3][k][tid], xr[4][k][tid], xr[5][k][tid]);
            if (!occ[k]) {
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    float ts;
                    if (rtx::ray_sphere(o2, ld, rtx::sphere_center(s), rtx::sphere_radius(s), ts))
                        if (ts > 0.0f && ts <= dist[k]) occ[k] = true;
                }
            }
            if (!occ[k]) {
                lit[k] = true;
                const v3 o = rtx::V(ox[k], oy[k], oz[k]);
                const v3 nn = rtx::V(nx[k], ny[k], nz[k]);
                const float ndl = rtx::dot(nn, ld);
                const float diffuse = rtx::max0(ndl);
                const v3 v = rtx::normalize(rtx::sub(cpos, o));
                const v3 I = rtx::muls(ld, -1.0f);
                float kk = 0.0f;
                if (ndl > 0.0f) {
                    const v3 rr = rtx::normalize(rtx::sub(I, rtx::muls(nn, 2.0f * rtx::dot(nn, I))));
                    kk = rtx::pow30(rtx::max0(rtx::dot(v, rr)));
                }
                const float att = 1.0f / (0.5f * dist[k]);
                const v3 add = rtx::muls(rtx::muls(lcol, diffuse + kk), att);
                lr[k] = lr[k] + add.x; lg[k] = lg[k] + add.y; lb[k] = lb[k] + add.z;
            }
        }
    }

    if ((tid & 63) == 0) add_counter(p.counters, C_SH_CAND, n_cand);
    if (tid == 0) {
        add_counter(p.counters, C_SH_RAYS, (unsigned long long)mine * (unsigned long long)p.num_lights);
        add_counter(p.counters, C_SH_PAIRS, n_pairs * (unsigned long long)mine);
    }

    // texture, final colour, kill rule, pixel  (raycol_cs.glsl:82-127)
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (!valid[k]) continue;
        const int i = ridx[k];
        v3 fin = rtx::V(0.1f, 0.1f, 0.1f);
        if (lit[k]) {
            const int id = p.rays.id[i];
            v3 pa = rtx::V(0.0f, 0.0f, 0.0f), pb = pa, pc = pa;
            float ta0 = 0.0f, ta1 = 0.0f, tb0 = 0.0f, tb1 = 0.0f, tc0 = 0.0f, tc1 = 0.0f;
            // robust buffer access: an index outside the buffer reads zeros
            const long long i0 = (long long)id * 3;
            if (i0 >= 0 && i0 < p.num_vertices) { const rt_vertex vv = p.verts[i0]; pa = rtx::V(vv.pos[0], vv.pos[1], vv.pos[2]); ta0 = vv.texCoord[0]; ta1 = vv.texCoord[1]; }
            if (i0 + 1 >= 0 && i0 + 1 < p.num_vertices) { const rt_vertex vv = p.verts[i0 + 1]; pb = rtx::V(vv.pos[0], vv.pos[1], vv.pos[2]); tb0 = vv.texCoord[0]; tb1 = vv.texCoord[1]; }
            if (i0 + 2 >= 0 && i0 + 2 < p.num_vertices) { const rt_vertex vv = p.verts[i0 + 2]; pc = rtx::V(vv.pos[0], vv.pos[1], vv.pos[2]); tc0 = vv.texCoord[0]; tc1 = vv.texCoord[1]; }
            const v3 o = rtx::V(ox[k], oy[k], oz[k]);
            const v3 v0 = rtx::sub(pb, pa), v1 = rtx::sub(pc, pa), v2 = rtx::sub(o, pa);
            const float d00 = rtx::dot(v0, v0), d01 = rtx::dot(v0, v1), d11 = rtx::dot(v1, v1);
            const float d20 = rtx::dot(v2, v0), d21 = rtx::dot(v2, v1);
            const float denom = d00 * d11 - d01 * d01;
            const float bv = (d11 * d20 - d01 * d21) / denom;
            const float bw = (d00 * d21 - d01 * d20) / denom;
            const float bu = 1.0f - bv - bw;
            const float uvx = bu * ta0 + bv * tb0 + bw * tc0;
            const float uvy = bu * ta1 + bv * tb1 + bw * tc1;
            const float fx = uvx * (float)p.tex_w, fy = uvy * (float)p.tex_h;
            v3 texc = rtx::V(0.0f, 0.0f, 0.0f);
            if (p.tex && fx > -1.0f && fx < (float)p.tex_w && fy > -1.0f && fy < (float)p.tex_h) {
                const uint32_t tx = p.tex[(size_t)((int)fy) * p.tex_w + (int)fx];
                texc = rtx::V((float)(tx & 255u) / 255.0f, (float)((tx >> 8) & 255u) / 255.0f,
                              (float)((tx >> 16) & 255u) / 255.0f);
            }
            fin = rtx::add(texc, rtx::muls(rtx::V(lr[k], lg[k], lb[k]), 0.3f));
            if (id + 3 * 5 - 2 < p.num_vertices / 3) {
                p.rays.id[i] = -2;
                p.rays.ox[i] = ox[k]; p.rays.oy[i] = oy[k]; p.rays.oz[i] = oz[k];
            }
        }
        p.fb[i] = rtx::to_unorm8(fin.x) | (rtx::to_unorm8(fin.y) << 8) | (rtx::to_unorm8(fin.z) << 16) | 0xff000000u;
    }
}

} // namespace rtk
