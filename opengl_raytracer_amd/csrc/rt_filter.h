// rt_filter.h -- the conservative (ray, triangle) filter and the per-triangle
// records it reads.  Compiles for the device (hipcc) and, for the randomized
// conservativeness test, for the host (g++).
//
// Why a filter: the reference tests every ray against every triangle with
// Moller-Trumbore ("OpenGL Raytracer/trace.glsl":1-22, loop :135-157).  Written
// out, that is ~31 multiply/add instructions, an IEEE divide and ~8 compares per
// pair.  Almost every pair is a miss.  The filter decides "certainly a miss" with
// 12 (rays that share an origin) or 23 (any ray) fused instructions per pair; the
// few pairs it cannot reject are then evaluated with the reference's own
// arithmetic, so the result is bit-identical to evaluating every pair that way.
//
// What the filter computes.  With s = o - X, e1 = Y - X, e2 = Z - X the
// reference's numerators are su = s.(d x e2), sv = d.(s x e1) and its
// denominator a = e1.(d x e2); it accepts only if u = su/a >= 0, v = sv/a >= 0
// and u + v <= 1, i.e. only if su, sv and a - su - sv all carry the sign of a.
// The filter evaluates the same three scalar triple products in a cheaper
// algebraic form and passes the pair iff all three are >= -delta or all three are
// <= +delta, where delta bounds the rounding error of both forms:
//   |difference| <= ~80 eps |s||d| + ~35 eps E|d|   (eps = 2^-24, E = longest edge,
//                                                    per-triangle data scaled so
//                                                    that its edges are < 1)
// and delta = 256 eps (|o - c| + 3 r + extra) |d| with (c, r) a bounding sphere
// of the mesh, so |s| + E <= |o - c| + 3 r.  The ray direction is pre-multiplied
// by a power of two >= 1/delta, which turns the threshold into the constant 1.
//
// Records are 12 floats (48 bytes) per triangle:
//   general  : X[3], e1*sc[3], e2*sc[3], (e1*sc) x e2 [3]        sc = 2^-k, |e*sc| < 1
//   origin O : ((Y-O) x (Z-O))*sc, ((Z-O) x (X-O))*sc, ((X-O) x (Y-O))*sc, 0,0,0
// Cross products are formed in double and rounded once.  A triangle with
// Y == X or Z == X (the reference can never accept it: a == 0) gets an all-NaN
// record, which no ray passes.
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rtf {

struct TriRec {
    float v[12];
};

RT_HD float fbits(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}
RT_HD uint32_t ubits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

// 2^-k with emax * 2^-k in [0.5, 1); exponent clamped so the scale is finite.
RT_HD float edge_scale(float emax)
{
    int e = 0;
    (void)frexpf(emax, &e);
    if (e > 120) e = 120;
    if (e < -120) e = -120;
    return ldexpf(1.0f, -e);
}

RT_HD bool degenerate(const float X[3], const float Y[3], const float Z[3], float *emax_out)
{
    // e1, e2 exactly as the reference forms them (one float subtraction each)
    float e1[3] = {Y[0] - X[0], Y[1] - X[1], Y[2] - X[2]};
    float e2[3] = {Z[0] - X[0], Z[1] - X[1], Z[2] - X[2]};
    float m1 = fmaxf(fmaxf(fabsf(e1[0]), fabsf(e1[1])), fabsf(e1[2]));
    float m2 = fmaxf(fmaxf(fabsf(e2[0]), fabsf(e2[1])), fabsf(e2[2]));
    float m = fmaxf(m1, m2);
    *emax_out = m;
    // !(m1 > 0) also catches NaN edges: every reference comparison on them is false
    // or the denominator test rejects, so such a triangle is never accepted.
    return !(m1 > 0.0f) || !(m2 > 0.0f) || !(m < INFINITY);
}

RT_HD void nan_rec(TriRec *r)
{
    for (int i = 0; i < 12; i++) r->v[i] = fbits(0x7fc00000u);
}

RT_HD void make_general_rec(const float X[3], const float Y[3], const float Z[3], TriRec *r)
{
    float emax;
    if (degenerate(X, Y, Z, &emax)) {
        nan_rec(r);
        return;
    }
    const float sc = edge_scale(emax);
    float e1[3] = {Y[0] - X[0], Y[1] - X[1], Y[2] - X[2]};
    float e2[3] = {Z[0] - X[0], Z[1] - X[1], Z[2] - X[2]};
    float e1s[3] = {e1[0] * sc, e1[1] * sc, e1[2] * sc};
    float e2s[3] = {e2[0] * sc, e2[1] * sc, e2[2] * sc};
    double n0 = (double)e1s[1] * e2[2] - (double)e1s[2] * e2[1];
    double n1 = (double)e1s[2] * e2[0] - (double)e1s[0] * e2[2];
    double n2 = (double)e1s[0] * e2[1] - (double)e1s[1] * e2[0];
    r->v[0] = X[0]; r->v[1] = X[1]; r->v[2] = X[2];
    r->v[3] = e1s[0]; r->v[4] = e1s[1]; r->v[5] = e1s[2];
    r->v[6] = e2s[0]; r->v[7] = e2s[1]; r->v[8] = e2s[2];
    r->v[9] = (float)n0; r->v[10] = (float)n1; r->v[11] = (float)n2;
}

RT_HD void make_origin_rec(const float X[3], const float Y[3], const float Z[3],
                           const float O[3], TriRec *r)
{
    float emax;
    if (degenerate(X, Y, Z, &emax)) {
        nan_rec(r);
        return;
    }
    const double sc = (double)edge_scale(emax);
    double p0[3], p1[3], p2[3];
    for (int i = 0; i < 3; i++) {
        p0[i] = (double)X[i] - (double)O[i];
        p1[i] = (double)Y[i] - (double)O[i];
        p2[i] = (double)Z[i] - (double)O[i];
    }
    // c0 = p1 x p2, c1 = p2 x p0, c2 = p0 x p1
    r->v[0] = (float)((p1[1] * p2[2] - p1[2] * p2[1]) * sc);
    r->v[1] = (float)((p1[2] * p2[0] - p1[0] * p2[2]) * sc);
    r->v[2] = (float)((p1[0] * p2[1] - p1[1] * p2[0]) * sc);
    r->v[3] = (float)((p2[1] * p0[2] - p2[2] * p0[1]) * sc);
    r->v[4] = (float)((p2[2] * p0[0] - p2[0] * p0[2]) * sc);
    r->v[5] = (float)((p2[0] * p0[1] - p2[1] * p0[0]) * sc);
    r->v[6] = (float)((p0[1] * p1[2] - p0[2] * p1[1]) * sc);
    r->v[7] = (float)((p0[2] * p1[0] - p0[0] * p1[2]) * sc);
    r->v[8] = (float)((p0[0] * p1[1] - p0[1] * p1[0]) * sc);
    r->v[9] = 0.0f; r->v[10] = 0.0f; r->v[11] = 0.0f;
}

// Power-of-two factor applied to a ray's direction so that the filter threshold
// becomes 1:  1/scale in [delta, 2 delta),  delta = 2^-16 * reach * |d|.
// `reach` must bound |o - X| + (longest edge) for every triangle, plus, for the
// shared-origin form, the distance by which the ray's own origin may lie off the
// line through the shared origin (see the callers).  NaN when the ray can never
// hit (zero, infinite or NaN direction or reach).
RT_HD float dir_scale(float reach, float dx, float dy, float dz)
{
    float dn = sqrtf(dx * dx + dy * dy + dz * dz);
    float delta = 1.52587890625e-05f * reach * dn; // 2^-16
    if (!(delta > 0.0f) || !(delta < INFINITY)) return fbits(0x7fc00000u);
    int e = 0;
    (void)frexpf(delta, &e); // delta = m 2^e, m in [0.5, 1)  ->  delta < 2^e
    if (e > 126) return fbits(0x7fc00000u);
    if (e < -126) e = -126;
    return ldexpf(1.0f, -e);
}

RT_HD float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
RT_HD float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// General form.  (ox,oy,oz) is the ray origin the reference would use, (hx,hy,hz)
// the direction multiplied by dir_scale().  Returns g; the pair passes iff g >= -1.
RT_HD float filter_general(const float *rec, float ox, float oy, float oz, float hx,
                           float hy, float hz)
{
    float sx = ox - rec[0], sy = oy - rec[1], sz = oz - rec[2];
    // m = s x h
    float mx = fmaf(sy, hz, -(sz * hy));
    float my = fmaf(sz, hx, -(sx * hz));
    float mz = fmaf(sx, hy, -(sy * hx));
    float U = fmaf(rec[8], mz, fmaf(rec[7], my, rec[6] * mx));   //  e2' . m  =  sc su
    float Vn = fmaf(rec[5], mz, fmaf(rec[4], my, rec[3] * mx));  //  e1' . m  = -sc sv
    float A = fmaf(rec[11], hz, fmaf(rec[10], hy, rec[9] * hx)); //  h . N''  = -sc a
    float W = (-A - U) + Vn;                                     //  sc (a - su - sv)
    float lo = min3f(U, -Vn, W);
    float hi = max3f(U, -Vn, W);
    return fmaxf(lo, -hi);
}

// Shared-origin form: the record was built for origin O and the ray's line
// passes through O.  (hx,hy,hz) as above.
RT_HD float filter_origin(const float *rec, float hx, float hy, float hz)
{
    float F0 = fmaf(rec[2], hz, fmaf(rec[1], hy, rec[0] * hx));
    float F1 = fmaf(rec[5], hz, fmaf(rec[4], hy, rec[3] * hx));
    float F2 = fmaf(rec[8], hz, fmaf(rec[7], hy, rec[6] * hx));
    float lo = min3f(F0, F1, F2);
    float hi = max3f(F0, F1, F2);
    return fmaxf(lo, -hi);
}

} // namespace rtf
