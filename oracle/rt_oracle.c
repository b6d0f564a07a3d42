/*
 * rt_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see rt_oracle.h).
 *
 * Each function cites the reference lines it restates; paths are relative to
 * "OpenGL Raytracer/" in the reference tree.  PARITY UNPINNED by reference
 * fixtures (it has none); pinned by the hand-derived known-answer tests.
 */
#include "rt_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;

static int g_threads = 1;
void rto_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int rto_get_threads(void) { return g_threads; }

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 ld3(const float *p) { return V(p[0], p[1], p[2]); }
static inline void st3(float *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 muls(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
/* GLSL dot(): x*x' + y*y' + z*z', left to right */
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* GLSL cross(): (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y) */
static inline v3 cross(v3 a, v3 b)
{
    return V(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline v3 normalize(v3 a) { return muls(a, 1.0f / sqrtf(dot(a, a))); }
static inline float distance3(v3 a, v3 b) { v3 d = sub(a, b); return sqrtf(dot(d, d)); }
static inline float glsl_max0(float x) { return (x < 0.0f) ? 0.0f : x; }
static inline float pow30(float x)
{
    float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8;
    return ((x16 * x8) * x4) * x2;
}

#define NEAR_ZERO 1e-20f /* definitions.glsl:53 */

/* trace.glsl:1-22 RayVsTriangle */
static inline int ray_vs_triangle(v3 o, v3 d, v3 tx, v3 ty, v3 tz, float *t)
{
    v3 e1 = sub(ty, tx);
    v3 e2 = sub(tz, tx);
    v3 q = cross(d, e2);
    float a = dot(e1, q);
    if (a > -NEAR_ZERO && a < NEAR_ZERO) return 0;

    float f = 1.0f / a;
    v3 s = sub(o, tx);
    float u = f * dot(s, q);
    if (u < 0.0f || u > 1.0f) return 0;

    v3 r = cross(s, e1);
    float v = f * dot(d, r);
    if (v < 0.0f || u + v > 1.0f) return 0;

    *t = f * dot(e2, r);
    return 1;
}

/* trace.glsl:57-78 RaySphereIntersect */
static inline int ray_sphere(v3 o, v3 d, v3 c, float r, float *t)
{
    v3 l = sub(c, o);
    float s = dot(l, d);
    float l_sqr = dot(l, l);
    float r_sqr = r * r;
    int outside = l_sqr > r_sqr;
    if (s < 0.0f && outside) return 0;

    float m_sqr = l_sqr - s * s;
    if (m_sqr > r_sqr) return 0;

    float q = sqrtf(r_sqr - m_sqr);
    if (outside) *t = s - q; else *t = s + q;
    return 1;
}

int rto_ray_vs_triangle(const float o[3], const float d[3], const float x[3],
                        const float y[3], const float z[3], float *t)
{
    return ray_vs_triangle(ld3(o), ld3(d), ld3(x), ld3(y), ld3(z), t);
}
int rto_ray_sphere(const float o[3], const float d[3], const float c[3], float r,
                   float *t)
{
    return ray_sphere(ld3(o), ld3(d), ld3(c), r, t);
}

/* SSBO read with robust-buffer-access semantics: outside the buffer -> zeros. */
static inline void fetch_vertex(const rt_vertex *vb, int nv, int i, v3 *pos, float tc[2])
{
    if (i < 0 || i >= nv) {
        *pos = V(0.0f, 0.0f, 0.0f);
        if (tc) { tc[0] = 0.0f; tc[1] = 0.0f; }
        return;
    }
    *pos = ld3(vb[i].pos);
    if (tc) { tc[0] = vb[i].texCoord[0]; tc[1] = vb[i].texCoord[1]; }
}

/* the part of struct ray that trace() reads and writes */
typedef struct {
    v3 origin, dir, color, n;
    float t;
    int id;
} tray;

/* trace.glsl:121-184 trace(r, earlyExit = false) */
static void trace(tray *r, const rt_vertex *vb, int nv)
{
    r->t = INFINITY; /* 1.0 / 0.0, trace.glsl:123 */
    float t = 0.0f;
    r->id = 0;
    int primitiveID = 0;
    for (int i = 0; i < nv; i += 3) {
        v3 x, y, z;
        fetch_vertex(vb, nv, i, &x, 0);
        fetch_vertex(vb, nv, i + 1, &y, 0);
        fetch_vertex(vb, nv, i + 2, &z, 0);
        if (ray_vs_triangle(r->origin, r->dir, x, y, z, &t)) {
            if (t > 0.0f && t < r->t) {
                r->t = t;
                r->color = V(0.0f, 0.0f, 0.0f);
                r->id = primitiveID;
                v3 u = sub(x, z);
                v3 v = sub(y, z);
                r->n = normalize(cross(u, v));
            }
        }
        primitiveID++;
    }

    /* trace.glsl:159-162 */
    const v3 sph_c[2] = {{0.0f, -0.7f, 2.0f}, {-2.0f, -0.0f, 0.0f}};
    const float sph_r[2] = {0.5f, 0.2f};
    for (int i = 0; i < 2; i++) {
        if (ray_sphere(r->origin, r->dir, sph_c[i], sph_r[i], &t)) {
            if (t > 0.0f && t < r->t) {
                r->t = t;
                r->color = V(0.0f, 0.0f, 1.0f);
                r->id = primitiveID;
                v3 p = add(r->origin, muls(r->dir, r->t));
                r->n = normalize(sub(p, sph_c[i]));
            }
        }
        primitiveID++;
    }
}

/* raygen_cs.glsl:1-22 */
void rto_raygen(rt_ray *rays, int width, int height, const int32_t *rows,
                int num_rows, const rto_camera *cam)
{
    const v3 cpos = ld3(cam->pos), cdir = ld3(cam->dir);
    const v3 cup = ld3(cam->up), cright = ld3(cam->right);
    const float hw = (float)width / 2.0f;
    const float hh = (float)height / 2.0f;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int j = 0; j < num_rows; j++) {
        const int y = rows ? rows[j] : j;
        for (int x = 0; x < width; x++) {
            float dx = ((float)x - hw) / hw;
            float dy = ((float)y - hh) / hh;
            /* camera_pos + camera_dir*NEAR_PLANE_DIST + dx*right + dy*up */
            v3 s = add(add(add(cpos, muls(cdir, 1.0f)), muls(cright, dx)), muls(cup, dy));
            v3 rd = normalize(sub(s, cpos));
            rt_ray *r = &rays[(size_t)j * width + x];
            memset(r, 0, sizeof *r);
            st3(r->origin, cpos);
            st3(r->dir, rd);
            r->color[0] = 1.0f; r->color[1] = 0.0f; r->color[2] = 0.0f;
            r->t = -1.0f;
            r->primitiveID = -1;
            r->n[0] = r->n[1] = r->n[2] = 0.0f;
        }
    }
}

/* rayintersect_cs.glsl:1-34 */
void rto_intersect(rt_ray *rays, int num_rays, const rt_vertex *vb, int nv,
                   const rto_camera *cam, float *depth)
{
    const v3 cpos = ld3(cam->pos);
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 64)
    for (int i = 0; i < num_rays; i++) {
        rt_ray *g = &rays[i];
        if (g->primitiveID == -2) continue;
        tray r;
        r.origin = ld3(g->origin); r.dir = ld3(g->dir); r.color = ld3(g->color);
        r.n = ld3(g->n); r.t = g->t; r.id = g->primitiveID;

        r.origin = add(r.origin, muls(r.dir, 0.001f));
        trace(&r, vb, nv);
        r.origin = sub(r.origin, muls(r.dir, 0.001f));
        if (isinf(r.t)) r.id = -2;

        r.origin = add(r.origin, muls(r.dir, r.t));
        /* r.dir - 2 * dot(r.dir, r.n) * r.n */
        r.dir = sub(r.dir, muls(r.n, 2.0f * dot(r.dir, r.n)));

        if (depth) depth[i] = distance3(r.origin, cpos);

        st3(g->origin, r.origin); st3(g->dir, r.dir); st3(g->color, r.color);
        st3(g->n, r.n); g->t = r.t; g->primitiveID = r.id;
    }
}

static inline uint8_t to_unorm8(float c)
{
    if (!(c == c)) return 0; /* NaN */
    if (c < 0.0f) c = 0.0f;
    if (c > 1.0f) c = 1.0f;
    return (uint8_t)rintf(c * 255.0f);
}

/* raycol_cs.glsl:18-128 (CartesianToBarycentricCoord :3-16 inlined) */
void rto_color(rt_ray *rays, int num_rays, const rt_vertex *vb, int nv,
               const rt_light *lights, int num_lights, const rto_camera *cam,
               const uint8_t *tex, int tex_w, int tex_h, uint8_t *out)
{
    const v3 cpos = ld3(cam->pos);
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 64)
    for (int i = 0; i < num_rays; i++) {
        rt_ray *g = &rays[i];
        v3 origin = ld3(g->origin), rn = ld3(g->n), rcolor = ld3(g->color);
        int id = g->primitiveID;

        v3 final = V(0.1f, 0.1f, 0.1f);
        int shadowed = 1;
        v3 light = V(0.0f, 0.0f, 0.0f);
        v3 texc = V(0.0f, 0.0f, 0.0f);

        for (int a = 0; a < num_lights; a++) {
            v3 lpos = ld3(lights[a].pos), lcol = ld3(lights[a].color);
            v3 ldir = normalize(sub(lpos, origin));

            tray lr;
            lr.origin = add(origin, muls(ldir, 0.0000001f));
            lr.dir = ldir; lr.color = rcolor; lr.t = -1.0f; lr.id = -1;
            lr.n = V(0.0f, 0.0f, 0.0f);
            trace(&lr, vb, nv);

            origin = sub(origin, muls(ldir, 0.0000001f));
            float d = distance3(lpos, origin);

            if (lr.t > d) {
                shadowed = 0;
                float diffuse = glsl_max0(dot(rn, ldir));
                v3 v = normalize(sub(cpos, origin));
                v3 I = muls(ldir, -1.0f);
                float k = 0.0f;
                if (dot(rn, ldir) > 0.0f) {
                    /* reflect(I, n) = I - 2*dot(n,I)*n */
                    v3 rr = normalize(sub(I, muls(rn, 2.0f * dot(rn, I))));
                    k = pow30(glsl_max0(dot(v, rr)));
                }
                float att = 1.0f / (0.5f * d);
                /* (diffuse + k) * color * attenuation */
                light = add(light, muls(muls(lcol, diffuse + k), att));
            }
        }

        if (!shadowed) {
            v3 pa, pb, pc;
            float ta[2], tb[2], tc[2];
            fetch_vertex(vb, nv, id * 3, &pa, ta);
            fetch_vertex(vb, nv, id * 3 + 1, &pb, tb);
            fetch_vertex(vb, nv, id * 3 + 2, &pc, tc);

            v3 v0 = sub(pb, pa), v1 = sub(pc, pa), v2 = sub(origin, pa);
            float d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1);
            float d20 = dot(v2, v0), d21 = dot(v2, v1);
            float denom = d00 * d11 - d01 * d01;
            float bv = (d11 * d20 - d01 * d21) / denom;
            float bw = (d00 * d21 - d01 * d20) / denom;
            float bu = 1.0f - bv - bw;
            float uvx = bu * ta[0] + bv * tb[0] + bw * tc[0];
            float uvy = bu * ta[1] + bv * tb[1] + bw * tc[1];

            float fx = uvx * (float)tex_w, fy = uvy * (float)tex_h;
            /* ivec2() truncates toward zero: in range iff -1 < f < size */
            if (tex && fx > -1.0f && fx < (float)tex_w && fy > -1.0f && fy < (float)tex_h) {
                int px = (int)fx, py = (int)fy;
                const uint8_t *p = tex + ((size_t)py * tex_w + px) * 4;
                texc = V((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f);
            }
            final = add(texc, muls(light, 0.3f));

            if (id + 3 * 5 - 2 < nv / 3) {
                /* rays[i] = r : the whole record, with the shifted origin */
                g->primitiveID = -2;
                st3(g->origin, origin);
            }
        }
        out[(size_t)i * 4 + 0] = to_unorm8(final.x);
        out[(size_t)i * 4 + 1] = to_unorm8(final.y);
        out[(size_t)i * 4 + 2] = to_unorm8(final.z);
        out[(size_t)i * 4 + 3] = 255;
    }
}

/* main.cpp:476-529 */
void rto_render(rt_ray *rays, int width, int height, const int32_t *rows,
                int num_rows, const rt_vertex *vb, int nv, const rt_light *lights,
                int num_lights, const rto_camera *cam, const uint8_t *tex,
                int tex_w, int tex_h, int passes, uint8_t *out, float *depth)
{
    rto_raygen(rays, width, height, rows, num_rows, cam);
    for (int p = 0; p < passes; p++) {
        rto_intersect(rays, num_rows * width, vb, nv, cam, depth);
        rto_color(rays, num_rows * width, vb, nv, lights, num_lights, cam, tex,
                  tex_w, tex_h, out);
    }
}
