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
//   origin O : ((Y-O) x (Z-O))*sc, ((Z-O) x (X-O))*sc, ((X-O) x (Y-O))*sc, (triangle index as bits on the device; 0), 0, 0
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
// the direction multiplied by dir_scale().  Gives lo = min and hi = max of the three
// scaled triple products; the pair passes iff lo >= -1 or hi <= 1, i.e. iff
// max(lo, -hi) >= -1.
RT_HD void filter_general_lohi(const float *rec, float ox, float oy, float oz, float hx,
                               float hy, float hz, float &lo, float &hi)
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
    lo = min3f(U, -Vn, W);
    hi = max3f(U, -Vn, W);
}

RT_HD float filter_general(const float *rec, float ox, float oy, float oz, float hx,
                           float hy, float hz)
{
    float lo, hi;
    filter_general_lohi(rec, ox, oy, oz, hx, hy, hz, lo, hi);
    return fmaxf(lo, -hi);
}

// Moment form of the general filter: the same three triple products from the ray's moment about the
// centre C of the mesh's bounding sphere, 20 fused operations per pair instead of 23 (no s = o - X, no
// cross product per pair).  With o' = o - C, X' = X - C and mo = o' x h,
//     e2'.((o - X) x h) = e2'.mo + (X' x e2').h        e1'.((o - X) x h) = e1'.mo + (X' x e1').h,
// and the record holds e1', e2' (as in the general record), P1 = X' x e1', P2 = X' x e2' (formed in
// double, rounded once) and N''.  Rounding, in units of |e'||h| (eps = 2^-24): o' = fl(o - C) moves the
// origin by eps |o'|; mo carries 4 eps |o'||h|; P by eps |X'||e'|; the six-term fused sum 6 eps times the
// sum of its terms' magnitudes <= (|o'| + |X'|)|e'||h|: together 11 eps (|o'| + |X'|) per product, twice
// that for the third (it is formed from the other two and h.N'').  |e'| < sqrt 3 and
// |o'| + |X'| <= |o - c| + r, so the filter's own share is about 38 eps (|o - c| + r)|d| where the general
// form's is about 14 eps |s||d|; with the reference's share (the rest of the 80 + 35 above) the
// difference stays below 160 eps (|o - c| + 3 r)|d| < delta.
// MomRec = 16 floats: e2'[3], P2[3], e1'[3], P1[3], N''[3], 0.
struct MomRec {
    float v[16];
};

RT_HD void make_moment_rec(const float X[3], const float Y[3], const float Z[3], const float C[3], MomRec *r)
{
    float emax;
    if (degenerate(X, Y, Z, &emax)) {
        for (int i = 0; i < 16; i++) r->v[i] = fbits(0x7fc00000u);
        return;
    }
    TriRec g;
    make_general_rec(X, Y, Z, &g); // e1' = v[3..5], e2' = v[6..8], N'' = v[9..11]
    const double xc[3] = {(double)X[0] - (double)C[0], (double)X[1] - (double)C[1], (double)X[2] - (double)C[2]};
    const float *e1 = g.v + 3, *e2 = g.v + 6;
    r->v[0] = e2[0]; r->v[1] = e2[1]; r->v[2] = e2[2];
    r->v[3] = (float)(xc[1] * e2[2] - xc[2] * e2[1]);
    r->v[4] = (float)(xc[2] * e2[0] - xc[0] * e2[2]);
    r->v[5] = (float)(xc[0] * e2[1] - xc[1] * e2[0]);
    r->v[6] = e1[0]; r->v[7] = e1[1]; r->v[8] = e1[2];
    r->v[9] = (float)(xc[1] * e1[2] - xc[2] * e1[1]);
    r->v[10] = (float)(xc[2] * e1[0] - xc[0] * e1[2]);
    r->v[11] = (float)(xc[0] * e1[1] - xc[1] * e1[0]);
    r->v[12] = g.v[9]; r->v[13] = g.v[10]; r->v[14] = g.v[11];
    r->v[15] = 0.0f;
}

// mo = (o - C) x h, once per ray; (hx,hy,hz) is the direction multiplied by dir_scale()
RT_HD void ray_moment(float ox, float oy, float oz, const float C[3], float hx, float hy, float hz, float &mx,
                      float &my, float &mz)
{
    const float px = ox - C[0], py = oy - C[1], pz = oz - C[2];
    mx = fmaf(py, hz, -(pz * hy));
    my = fmaf(pz, hx, -(px * hz));
    mz = fmaf(px, hy, -(py * hx));
}

// lo, hi as filter_general_lohi gives them (same pass rule)
RT_HD void filter_moment_lohi(const float *rec, float mx, float my, float mz, float hx, float hy, float hz,
                              float &lo, float &hi)
{
    float U = fmaf(rec[2], mz, fmaf(rec[1], my, fmaf(rec[0], mx, fmaf(rec[5], hz, fmaf(rec[4], hy, rec[3] * hx)))));
    float Vn = fmaf(rec[8], mz, fmaf(rec[7], my, fmaf(rec[6], mx, fmaf(rec[11], hz, fmaf(rec[10], hy, rec[9] * hx)))));
    float A = fmaf(rec[14], hz, fmaf(rec[13], hy, rec[12] * hx));
    float W = (-A - U) + Vn;
    lo = min3f(U, -Vn, W);
    hi = max3f(U, -Vn, W);
}

RT_HD float filter_moment(const float *rec, float mx, float my, float mz, float hx, float hy, float hz)
{
    float lo, hi;
    filter_moment_lohi(rec, mx, my, mz, hx, hy, hz, lo, hi);
    return fmaxf(lo, -hi);
}

// Shared-origin form: the record was built for origin O and the ray's line
// passes through O.  (hx,hy,hz) as above.
RT_HD void filter_origin_lohi(const float *rec, float hx, float hy, float hz, float &lo, float &hi)
{
    float F0 = fmaf(rec[2], hz, fmaf(rec[1], hy, rec[0] * hx));
    float F1 = fmaf(rec[5], hz, fmaf(rec[4], hy, rec[3] * hx));
    float F2 = fmaf(rec[8], hz, fmaf(rec[7], hy, rec[6] * hx));
    lo = min3f(F0, F1, F2);
    hi = max3f(F0, F1, F2);
}

RT_HD float filter_origin(const float *rec, float hx, float hy, float hz)
{
    float lo, hi;
    filter_origin_lohi(rec, hx, hy, hz, lo, hi);
    return fmaxf(lo, -hi);
}


// ---------------------------------------------------------------------------
// Cone pre-filter for rays whose line passes through a shared origin O.
//
// Seen from O a triangle is a spherical triangle of directions.  The edge filter
// above lets a direction through only if it lies inside that spherical triangle
// grown by a small angle per edge (plus its mirror image, which belongs to hits
// behind O).  A cone (unit axis, cosine threshold) that contains the grown
// spherical triangle therefore rejects nothing the edge filter -- and hence the
// reference -- could accept, and testing it costs one dot product per pair.
//
// make_cone_rec builds that cone in double precision from the ROUNDED edge record
// (so it bounds exactly what the edge filter evaluates at run time):
//   n_k   = +-c_k/|c_k|, signed so that the opposite vertex is inside,
//   b_k   = -(2^-15 reach_max 1.001 + 2^-22 |c_k|)/|c_k|   the edge filter's threshold
//           for any ray whose own reach is <= reach_max,
//   corner(k,l) = the unit direction with d.n_k = b_k and d.n_l = b_l next to the
//           shared vertex; the grown triangle is the hull of the three corners up to
//           an outward bulge <= max asin(-b_k),
//   threshold = cos(max angle(axis, corner) + 2 max asin(-b_k) + 2^-18) - 2^-19
//           (the last term covers the float dot product and the normalised direction).
// Whenever that construction is not safe -- a vertex at O, O within 2^-16 S of the
// triangle's plane (then rounding decides on which side of O the reference sees the
// hit), a corner that does not exist, an opening angle beyond 1.4 rad -- the record is
// (0,0,0,-2): every direction passes and the edge filter decides.
// Degenerate triangles (all-NaN edge record) get (0,0,0,2): nothing passes.
// ---------------------------------------------------------------------------
struct ConeRec {
    float v[4];
};

RT_HD void cone_always(ConeRec *c) { c->v[0] = c->v[1] = c->v[2] = 0.0f; c->v[3] = -2.0f; }
RT_HD void cone_never(ConeRec *c) { c->v[0] = c->v[1] = c->v[2] = 0.0f; c->v[3] = 2.0f; }

RT_HD double dot3d(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
RT_HD void cross3d(const double *a, const double *b, double *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

RT_HD void make_cone_rec(const float X[3], const float Y[3], const float Z[3], const float O[3],
                         const TriRec *edge, double scene_extent, double reach_max, ConeRec *out)
{
    if (!(edge->v[0] == edge->v[0])) { // NaN record: degenerate triangle
        cone_never(out);
        return;
    }
    double p[3][3];
    for (int i = 0; i < 3; i++) {
        p[0][i] = (double)X[i] - (double)O[i];
        p[1][i] = (double)Y[i] - (double)O[i];
        p[2][i] = (double)Z[i] - (double)O[i];
    }
    double u[3][3];
    for (int v = 0; v < 3; v++) {
        double l = sqrt(dot3d(p[v], p[v]));
        if (!(l > 0.0) || !(l < 1e300)) { cone_always(out); return; }
        for (int i = 0; i < 3; i++) u[v][i] = p[v][i] / l;
    }
    // distance of O from the triangle's plane
    double e1[3], e2[3], N[3];
    for (int i = 0; i < 3; i++) { e1[i] = p[1][i] - p[0][i]; e2[i] = p[2][i] - p[0][i]; }
    cross3d(e1, e2, N);
    double nl = sqrt(dot3d(N, N));
    if (!(nl > 0.0)) { cone_always(out); return; }
    double h = fabs(dot3d(p[0], N)) / nl;
    if (!(h > scene_extent * 1.52587890625e-05)) { cone_always(out); return; } // 2^-16

    // edge k of the record: c0 ~ p1 x p2 (opposite vertex 0), c1 ~ p2 x p0, c2 ~ p0 x p1
    double n[3][3], b[3], beta_max = 0.0;
    for (int k = 0; k < 3; k++) {
        double c[3] = {(double)edge->v[3 * k], (double)edge->v[3 * k + 1], (double)edge->v[3 * k + 2]};
        double cl = sqrt(dot3d(c, c));
        if (!(cl > 0.0)) { cone_always(out); return; }
        double sgn = dot3d(c, u[k]);
        if (sgn == 0.0) { cone_always(out); return; }
        sgn = sgn > 0.0 ? 1.0 : -1.0;
        for (int i = 0; i < 3; i++) n[k][i] = sgn * c[i] / cl;
        double delta = 3.0517578125e-05 * reach_max * 1.001 + 2.384185791015625e-07 * cl; // 2^-15, 2^-22
        b[k] = -delta / cl;
        if (!(b[k] > -0.5)) { cone_always(out); return; }
        double beta = asin(-b[k]);
        if (beta > beta_max) beta_max = beta;
    }
    // axis: towards the middle of the three vertex directions, rounded to float first
    double ax[3] = {u[0][0] + u[1][0] + u[2][0], u[0][1] + u[1][1] + u[2][1], u[0][2] + u[1][2] + u[2][2]};
    double al = sqrt(dot3d(ax, ax));
    if (!(al > 1e-3)) { cone_always(out); return; }
    float axf[3] = {(float)(ax[0] / al), (float)(ax[1] / al), (float)(ax[2] / al)};
    double a[3] = {(double)axf[0], (double)axf[1], (double)axf[2]};
    double an = sqrt(dot3d(a, a));
    double cos_min = 1.0;
    for (int v = 0; v < 3; v++) { // corner next to vertex v: edges k != v
        int k = (v + 1) % 3, l = (v + 2) % 3;
        double g = dot3d(n[k], n[l]);
        double det = 1.0 - g * g;
        if (!(det > 1e-12)) { cone_always(out); return; }
        double x = (b[k] - g * b[l]) / det, y = (b[l] - g * b[k]) / det;
        double w = x * x + y * y + 2.0 * x * y * g;
        if (!(w < 0.25)) { cone_always(out); return; }
        double m[3];
        cross3d(n[k], n[l], m);
        double ml = sqrt(dot3d(m, m));
        double z = sqrt(1.0 - w) / ml;
        if (dot3d(m, u[v]) < 0.0) z = -z;
        double cdir[3];
        for (int i = 0; i < 3; i++) cdir[i] = x * n[k][i] + y * n[l][i] + z * m[i];
        double cs = dot3d(cdir, a) / an; // cdir is unit by construction
        if (cs < cos_min) cos_min = cs;
        // the vertex direction itself must be inside the cone too
        double cv = dot3d(u[v], a) / an;
        if (cv < cos_min) cos_min = cv;
    }
    if (!(cos_min > -1.0)) { cone_always(out); return; }
    double theta = acos(cos_min > 1.0 ? 1.0 : cos_min) + 2.0 * beta_max + 3.814697265625e-06; // 2^-18
    if (!(theta < 1.4)) { cone_always(out); return; }
    double thr = cos(theta) * an - 1.9073486328125e-06; // 2^-19; `an` undoes the rounding of the axis length
    out->v[0] = axf[0]; out->v[1] = axf[1]; out->v[2] = axf[2];
    float t = (float)thr;
    if ((double)t > thr) t = nextafterf(t, -2.0f);
    out->v[3] = t;
}

// The shared-origin traversal skips slots in two steps: a cone per SUPER_RUN slots (two neighbouring
// clusters of 32 triangles), then a cone per RUN slots (half a cluster).  Measured on 64K triangles:
// (16, 64) 37.5 ms/frame, (16, 128) 37.7, (8, 64) 38.7, (16, 256) 39.3, (8, 128) 40.3, one level of 32: 39.7.
#ifndef RT_RUN
#define RT_RUN 16
#endif
#ifndef RT_SUPER_RUN
#define RT_SUPER_RUN 64
#endif
constexpr int RUN = RT_RUN;
constexpr int SUPER_RUN = RT_SUPER_RUN;

// A cone that contains the cones of a cluster of triangles: a direction that passes a member's
// test (float dot product with the member's axis >= its threshold) passes this one.  Members
// with threshold 2 (nothing passes) do not count; one member with threshold -2 (everything
// passes) makes the cluster's cone "everything".
//   member half-angle  th_k = acos(w_k / |a_k| - 2^-20)      (the float dot product, |b| != 1)
//   cluster half-angle Th   = max_k(angle(A, a_k) + th_k) + 1e-6,   threshold = |A| cos(Th) - 2e-6
RT_HD void make_cluster_cone(const ConeRec *m, int n, ConeRec *out)
{
    double ax[3] = {0.0, 0.0, 0.0};
    int live = 0;
    for (int k = 0; k < n; k++) {
        const float w = m[k].v[3];
        if (w >= 2.0f) continue;
        if (!(w > -2.0f)) { cone_always(out); return; }
        live++;
        for (int i = 0; i < 3; i++) ax[i] += (double)m[k].v[i];
    }
    if (!live) { cone_never(out); return; }
    const double al = sqrt(dot3d(ax, ax));
    if (!(al > 1e-3 * live)) { cone_always(out); return; }
    const float axf[3] = {(float)(ax[0] / al), (float)(ax[1] / al), (float)(ax[2] / al)};
    const double a[3] = {(double)axf[0], (double)axf[1], (double)axf[2]};
    const double an = sqrt(dot3d(a, a));
    double big = 0.0;
    for (int k = 0; k < n; k++) {
        const float w = m[k].v[3];
        if (w >= 2.0f) continue;
        const double mk[3] = {(double)m[k].v[0], (double)m[k].v[1], (double)m[k].v[2]};
        const double ml = sqrt(dot3d(mk, mk));
        if (!(ml > 0.0)) { cone_always(out); return; }
        double c = dot3d(mk, a) / (ml * an);
        c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
        double ct = (double)w / ml - 9.5367431640625e-07; // 2^-20
        if (!(ct > -1.0)) { cone_always(out); return; }
        if (ct > 1.0) ct = 1.0;
        const double ang = acos(c) + acos(ct);
        if (ang > big) big = ang;
    }
    big += 1e-6;
    if (!(big < 1.5)) { cone_always(out); return; }
    const double thr = cos(big) * an - 2e-6;
    out->v[0] = axf[0]; out->v[1] = axf[1]; out->v[2] = axf[2];
    float t = (float)thr;
    if ((double)t > thr) t = nextafterf(t, -2.0f);
    out->v[3] = t;
}

// `reach` of a ray whose line passes through the shared origin O (see dir_scale): the
// geometric part bounds |o - X| + (longest edge) and |X - O| for every vertex X inside the
// bounding sphere (C, R); the last term covers the only errors that grow with the absolute
// size of the coordinates -- the rounding of the ray's own origin, <= eps |o| -- with a
// margin of 8 relative to the 256 eps of the geometric part.
RT_HD float reach_shared(const float O[3], const float o[3], const float C[3], float R)
{
    float a = sqrtf((O[0] - C[0]) * (O[0] - C[0]) + (O[1] - C[1]) * (O[1] - C[1]) + (O[2] - C[2]) * (O[2] - C[2]));
    float b = sqrtf((o[0] - C[0]) * (o[0] - C[0]) + (o[1] - C[1]) * (o[1] - C[1]) + (o[2] - C[2]) * (o[2] - C[2]));
    float c = sqrtf(O[0] * O[0] + O[1] * O[1] + O[2] * O[2]);
    float d = sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
    return a + b + 3.0f * R + 0.125f * (c + d);
}

// largest reach_shared() for which the cones built for origin O are valid: it covers rays
// that start next to O (camera rays) and rays that start inside the bounding sphere (shadow rays)
RT_HD double cone_reach_max(const float O[3], const float C[3], float R)
{
    double oc = sqrt(((double)O[0] - C[0]) * ((double)O[0] - C[0]) + ((double)O[1] - C[1]) * ((double)O[1] - C[1]) +
                     ((double)O[2] - C[2]) * ((double)O[2] - C[2]));
    double o = sqrt((double)O[0] * O[0] + (double)O[1] * O[1] + (double)O[2] * O[2]);
    double c = sqrt((double)C[0] * C[0] + (double)C[1] * C[1] + (double)C[2] * C[2]);
    return (2.0 * oc + 4.0 * (double)R + 0.125 * (2.0 * o + c + (double)R)) * 1.0001 + 0.02;
}
// scale of the rounding errors in the reference's t and light distance, for the rule "O is
// too close to the triangle's plane to tell on which side of O the reference sees the hit"
RT_HD double scene_extent(const float O[3], const float C[3], float R)
{
    double oc = sqrt(((double)O[0] - C[0]) * ((double)O[0] - C[0]) + ((double)O[1] - C[1]) * ((double)O[1] - C[1]) +
                     ((double)O[2] - C[2]) * ((double)O[2] - C[2]));
    double o = sqrt((double)O[0] * O[0] + (double)O[1] * O[1] + (double)O[2] * O[2]);
    double c = sqrt((double)C[0] * C[0] + (double)C[1] * C[1] + (double)C[2] * C[2]);
    return oc + (double)R + 0.125 * (o + c + (double)R);
}

// ---------------------------------------------------------------------------
// Plane-band test for arbitrary rays, and how far from its triangle a well-conditioned hit can lie.
//
// Which pairs can the reference's arithmetic accept?  Write e1, e2 for the float edges it forms,
// s = o - X, a = e1.(d x e2), Nu = s.(d x e2), Nv = d.(s x e1), and a^, Nu^, Nv^ for what it
// computes.  First-order rounding analysis of its operation sequence gives
//     |a^ - a| <= 7 eps |e1||e2||d|,  |Nu^ - Nu| <= 8 eps |s||d||e2|,  |Nv^ - Nv| <= 8 eps |s||d||e1|.
// (Sampled over 2e7 random and near-coplanar pairs the three errors stay below 4.1 eps times those
// magnitudes.)  Let E = eps |d| max(9 S |e1|, 9 S |e2|, 8 |e1||e2|) with S >= |s|: above those bounds
// with room for the second-order terms.
//   (A) |a^| >= K E.  Then u^ = Nu^/a^ (1 + 2 eps) differs from the true u = Nu/a by less than
//       2/(K-1), likewise v; the reference accepts only 0 <= u^, 0 <= v^, u^ + v^ <= 1, so the point
//       where the ray's LINE meets the triangle's plane lies within rho = 2/(K-1) (|e1| + |e2|) of
//       the triangle: the line passes within rho of it.  The same bounds applied to t = Nt/a show
//       that a computed t > 0 has a true t >= -rho/|d|: the hit point is at most rho behind the origin.
//   (B) |a^| < K E.  Then |a| < (K+1) E, i.e. the direction is nearly parallel to the plane:
//       |d.n| / |d| < theta = (K+1) eps max(9 S / min(|e1|,|e2|), 8) / sin(angle(e1, e2)).
//       In case (B) the reference accepts only if also |Nu^| <= |a^| and |Nv^| <= |a^| (u^, v^ <= 1),
//       so Nu and Nv are below (K+1) E too; with m = s x d, Nu = m.e2, Nv = -m.e1, and
//           s.n = (-(s.d) a + Nv (d.e2) + Nu (d.e1)) / |d|^2,    n = e1 x e2,
//       the ray's origin then lies within  H = theta (S + |e1| + |e2|)  of the triangle's plane.
// So a pair outside the slab (B) -- direction out of the band, or origin farther than H from the
// plane -- whose line stays farther than rho from the triangle is one the reference rejects (the
// cells in space of rt_voxel.h are built on that).  K = 17: rho = (|e1| + |e2|)/8.
//
// BandRec = (unit normal, theta').  theta' = theta + 2^-20 covers the float dot product with a
// float-normalised direction.  theta' = +inf: every direction is in the band (slivers, edges
// outside [1e-12, 1e12], triangles too large to cluster); theta' = -1: no direction is (triangles
// the reference can never accept, padding).
// ---------------------------------------------------------------------------
struct BandRec {
    float v[4];
};
// SlabRec = (kappa, H'): the origin test is |n.o - kappa| < H', kappa = n.X in float, H' = H plus
// the rounding of n.o and kappa (<= 8 eps (|o| + |X|), bounded with abs_extent >= |o|, |X|).
struct SlabRec {
    float v[2];
};

constexpr double BAND_K = 17.0;

// Fills the band record for rays with |o - X| <= S and returns rho (< 0 when the triangle takes
// no part in clusters: it is never accepted, or always in the band).
RT_HD double make_band_rec(const float X[3], const float Y[3], const float Z[3], double S, double abs_extent,
                           BandRec *out, SlabRec *slab)
{
    float emax;
    out->v[0] = out->v[1] = out->v[2] = 0.0f;
    slab->v[0] = 0.0f;
    slab->v[1] = INFINITY;
    if (degenerate(X, Y, Z, &emax)) {
        out->v[3] = -1.0f;
        return -1.0;
    }
    out->v[3] = INFINITY;
    const float e1f[3] = {Y[0] - X[0], Y[1] - X[1], Y[2] - X[2]};
    const float e2f[3] = {Z[0] - X[0], Z[1] - X[1], Z[2] - X[2]};
    const double e1[3] = {e1f[0], e1f[1], e1f[2]}, e2[3] = {e2f[0], e2f[1], e2f[2]};
    double n[3];
    cross3d(e1, e2, n);
    const double l1 = sqrt(dot3d(e1, e1)), l2 = sqrt(dot3d(e2, e2)), nl = sqrt(dot3d(n, n));
    const double emin = l1 < l2 ? l1 : l2, elong = l1 < l2 ? l2 : l1;
    if (!(emin > 1e-12) || !(elong < 1e12) || !(nl > 0.0) || !(S > 0.0) || !(S < 1e12)) return -1.0;
    const double sinphi = nl / (l1 * l2);
    double w = 9.0 * S / emin;
    if (w < 8.0) w = 8.0;
    const double theta = (BAND_K + 1.0) * 5.9604644775390625e-08 * w / sinphi + 9.5367431640625e-07; // eps = 2^-24; 2^-20
    if (!(theta < 0.5)) return -1.0;
    out->v[0] = (float)(n[0] / nl);
    out->v[1] = (float)(n[1] / nl);
    out->v[2] = (float)(n[2] / nl);
    float t = (float)theta;
    if ((double)t < theta) t = nextafterf(t, 2.0f);
    out->v[3] = t;
    // kappa as the kernel would see it: float normal, double dot, rounded; the difference to the
    // kernel's float dot product of n with o is inside the slack below
    const double kappa = (double)out->v[0] * X[0] + (double)out->v[1] * X[1] + (double)out->v[2] * X[2];
    const double H = theta * (S + l1 + l2) * 1.001 + 16.0 * 5.9604644775390625e-08 * abs_extent;
    slab->v[0] = (float)kappa;
    float h = (float)H;
    if ((double)h < H) h = nextafterf(h, INFINITY);
    slab->v[1] = h;
    return 2.0 / (BAND_K - 1.0) * 1.02 * (l1 + l2);
}

// Origin within the slab?  rec = band record, slab = its SlabRec.
RT_HD bool slab_test(const float *rec, const float *slab, float ox, float oy, float oz)
{
    const float h = fmaf(rec[2], oz, fmaf(rec[1], oy, rec[0] * ox)) - slab[0];
    return fabsf(h) < slab[1];
}

// In the band?  (bx,by,bz) is the ray's direction normalised in float.
RT_HD bool band_test(const float *rec, float bx, float by, float bz)
{
    const float t = fmaf(rec[2], bz, fmaf(rec[1], by, rec[0] * bx));
    return fabsf(t) < rec[3];
}

} // namespace rtf
