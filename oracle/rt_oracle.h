/*
 * rt_oracle.h -- CPU oracle for the three-stage ray tracer.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a scalar restatement of the reference's
 * GLSL, used as the checker in tests/, in __graft_entry__.smoke() and as the
 * timed CPU baseline of bench.py.  The product library never links, loads or
 * calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden images or known-answer
 * vectors for this path, and its shaders cannot be executed in this image (no
 * GLSL compiler, no EGL/OSMesa context).  The oracle is therefore pinned only
 * by hand-derived known-answer tests (tests/test_oracle_kat.py) that follow
 * the shader text.
 *
 * Arithmetic: IEEE binary32, every expression associated exactly as the GLSL
 * text associates it, no FMA contraction (-ffp-contract=off), IEEE divide and
 * sqrt.  Built-ins with implementation-defined precision are fixed as:
 *   normalize(v) = v * (1.0f / sqrtf(dot(v,v)))
 *   distance(a,b) = sqrtf(dot(a-b, a-b))
 *   reflect(I,N) = I - (2.0f*dot(N,I))*N
 *   pow(x, 30)   = x^30 by squaring: ((x^2)^2..) x16 * x8 * x4 * x2
 *   max(x, 0)    = (x < 0) ? 0 : x            (NaN stays NaN, as in GLSL)
 *   ivec2(f)     = truncation toward zero; NaN or out of int range = out of
 *                  bounds
 *   imageLoad out of bounds = 0; SSBO vertex index out of bounds = zero vertex
 *                  (robust buffer access)
 *   imageStore to rgba8 = rint(clamp(c,0,1)*255) ties-to-even, NaN -> 0
 * Ray index stride is the frame WIDTH (the reference uses the height, which is
 * only well defined for square frames).
 */
#ifndef RT_ORACLE_H_
#define RT_ORACLE_H_

#include <stdint.h>
#include "../include/rt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rto_camera {
    float pos[3], dir[3], up[3], right[3];
} rto_camera;

/* raygen_cs.glsl:1-22.  Writes rays for the rows listed in rows[num_rows]
 * (frame y of each local row) into rays[num_rows*width]. */
void rto_raygen(rt_ray *rays, int width, int height, const int32_t *rows,
                int num_rows, const rto_camera *cam);

/* rayintersect_cs.glsl:1-34 with trace.glsl:121-184.  depth may be NULL. */
void rto_intersect(rt_ray *rays, int num_rays, const rt_vertex *vertices,
                   int num_vertices, const rto_camera *cam, float *depth);

/* raycol_cs.glsl:18-128 with trace.glsl:121-184.  Writes num_rays RGBA8 pixels. */
void rto_color(rt_ray *rays, int num_rays, const rt_vertex *vertices,
               int num_vertices, const rt_light *lights, int num_lights,
               const rto_camera *cam, const uint8_t *tex, int tex_w, int tex_h,
               uint8_t *out_rgba8);

/* main.cpp:476-529: raygen, then `passes` x { intersect, color }. */
void rto_render(rt_ray *rays, int width, int height, const int32_t *rows,
                int num_rows, const rt_vertex *vertices, int num_vertices,
                const rt_light *lights, int num_lights, const rto_camera *cam,
                const uint8_t *tex, int tex_w, int tex_h, int passes,
                uint8_t *out_rgba8, float *depth);

/* trace.glsl:1-22, exposed for known-answer tests.  Returns 0/1. */
int rto_ray_vs_triangle(const float o[3], const float d[3], const float x[3],
                        const float y[3], const float z[3], float *t);
/* trace.glsl:57-78. */
int rto_ray_sphere(const float o[3], const float d[3], const float c[3], float r,
                   float *t);

/* Number of threads the row loops use (OpenMP); 1 = scalar.  Timing only. */
void rto_set_threads(int n);
int rto_get_threads(void);

#ifdef __cplusplus
}
#endif
#endif
