// filter_model.cpp -- host-side model of the device filter, for the randomized
// conservativeness test (tests/test_filter_model.py).  Test code only.
//
// It includes the very header the kernels use (rt_filter.h compiles for the host)
// and compares, pair by pair, "the filter lets the pair through" with "the oracle's
// Moller-Trumbore accepts the pair".  A pair the oracle accepts but the filter
// rejects would be a wrong image; the count of such pairs must be zero.
#include <math.h>
#include <stdint.h>

#include "../../opengl_raytracer_amd/csrc/rt_filter.h"

extern "C" int rto_ray_vs_triangle(const float o[3], const float d[3], const float x[3], const float y[3],
                                   const float z[3], float *t);

static float norm3(const float *v) { return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

// stats: [0] pairs, [1] filter passes, [2] oracle accepts, [3] violations (accept && !pass)
extern "C" void fm_check_general(int n_tris, const float *verts /* n_tris*9 */, int n_rays,
                                 const float *org /* offset origins */, const float *dir,
                                 const float *center, float radius, long long *stats)
{
    long long pairs = 0, pass = 0, acc = 0, bad = 0;
#pragma omp parallel for reduction(+ : pairs, pass, acc, bad) schedule(static)
    for (int k = 0; k < n_tris; k++) {
        const float *X = verts + 9 * k, *Y = X + 3, *Z = X + 6;
        rtf::TriRec rec;
        rtf::make_general_rec(X, Y, Z, &rec);
        for (int i = 0; i < n_rays; i++) {
            const float *o = org + 3 * i, *d = dir + 3 * i;
            float oc[3] = {o[0] - center[0], o[1] - center[1], o[2] - center[2]};
            float reach = norm3(oc) + 3.0f * radius;
            float sc = rtf::dir_scale(reach, d[0], d[1], d[2]);
            float g = rtf::filter_general(rec.v, o[0], o[1], o[2], d[0] * sc, d[1] * sc, d[2] * sc);
            bool p = g >= -1.0f;
            float t;
            bool a = rto_ray_vs_triangle(o, d, X, Y, Z, &t) != 0 && t > 0.0f; // trace() keeps t > 0 only
            pairs++;
            pass += p;
            acc += a;
            bad += (a && !p);
        }
    }
    stats[0] = pairs; stats[1] = pass; stats[2] = acc; stats[3] = bad;
}

// Shared-origin form.  org[i] is the origin the reference would use for ray i (it
// lies on the line through O within rounding), dir[i] its direction.
extern "C" void fm_check_origin(int n_tris, const float *verts, int n_rays, const float *O, const float *org,
                                const float *dir, const float *center, float radius, long long *stats)
{
    long long pairs = 0, pass = 0, acc = 0, bad = 0;
#pragma omp parallel for reduction(+ : pairs, pass, acc, bad) schedule(static)
    for (int k = 0; k < n_tris; k++) {
        const float *X = verts + 9 * k, *Y = X + 3, *Z = X + 6;
        rtf::TriRec rec;
        rtf::make_origin_rec(X, Y, Z, O, &rec);
        for (int i = 0; i < n_rays; i++) {
            const float *o = org + 3 * i, *d = dir + 3 * i;
            float reach = rtf::reach_shared(O, o, center, radius);
            float sc = rtf::dir_scale(reach, d[0], d[1], d[2]);
            float g = rtf::filter_origin(rec.v, d[0] * sc, d[1] * sc, d[2] * sc);
            bool p = g >= -1.0f;
            float t;
            bool a = rto_ray_vs_triangle(o, d, X, Y, Z, &t) != 0 && t > 0.0f;
            pairs++;
            pass += p;
            acc += a;
            bad += (a && !p);
        }
    }
    stats[0] = pairs; stats[1] = pass; stats[2] = acc; stats[3] = bad;
}

// Cone pre-filter + edge filter cascade of the shared-origin traversal, as the kernel
// evaluates it.  tmax may be null (closest hit: any t > 0 counts) or hold one limit per ray
// (shadow query: 0 < t <= tmax).  sign = +1 camera rays, -1 shadow rays.
// stats: [0] pairs [1] cone passes [2] cone+edge passes [3] oracle accepts [4] violations
//        [5] triangles whose cone is "always pass" [6] rays outside the cones' validity (wild)
extern "C" void fm_check_cone(int n_tris, const float *verts, int n_rays, const float *O, const float *org,
                              const float *dir, const float *tmax, const float *center, float radius, float sign,
                              long long *stats)
{
    long long pairs = 0, p1 = 0, p2 = 0, acc = 0, bad = 0, always = 0, wildn = 0;
    const double extent = rtf::scene_extent(O, center, radius);
    const double rmax = rtf::cone_reach_max(O, center, radius);
    const float rmaxf = (float)rmax;
#pragma omp parallel for reduction(+ : pairs, p1, p2, acc, bad, always, wildn) schedule(static)
    for (int k = 0; k < n_tris; k++) {
        const float *X = verts + 9 * k, *Y = X + 3, *Z = X + 6;
        rtf::TriRec rec;
        rtf::make_origin_rec(X, Y, Z, O, &rec);
        rtf::ConeRec cone;
        rtf::make_cone_rec(X, Y, Z, O, &rec, extent, rmax, &cone);
        always += (cone.v[3] == -2.0f);
        for (int i = 0; i < n_rays; i++) {
            const float *o = org + 3 * i, *d = dir + 3 * i;
            const float reach = rtf::reach_shared(O, o, center, radius);
            const float inv = sign / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            const float bx = d[0] * inv, by = d[1] * inv, bz = d[2] * inv;
            const float g = rtf::dir_scale(reach, bx, by, bz);
            const bool wild = !(reach <= rmaxf);
            const float dt = fmaf(cone.v[2], bz, fmaf(cone.v[1], by, cone.v[0] * bx));
            const bool c1 = (dt >= cone.v[3]) || wild;
            bool c2 = false;
            if (c1) {
                const float s = sign * g;
                c2 = rtf::filter_origin(rec.v, bx * s, by * s, bz * s) >= -1.0f;
            }
            float t;
            bool a = rto_ray_vs_triangle(o, d, X, Y, Z, &t) != 0 && t > 0.0f;
            if (a && tmax) a = t <= tmax[i];
            pairs++;
            p1 += c1;
            p2 += c2;
            acc += a;
            bad += (a && !c2);
            if (k == 0) wildn += wild;
        }
    }
    stats[0] = pairs; stats[1] = p1; stats[2] = p2; stats[3] = acc; stats[4] = bad; stats[5] = always; stats[6] = wildn;
}
