// rt_grid.h -- direction cells around a shared origin.
//
// Every ray of a shared-origin launch (camera rays, shadow rays) is a direction seen from one
// point O, and so is every triangle: a cone of directions (rt_filter.h, make_cone_rec) outside of
// which the edge filter -- hence the reference -- rejects the pair.  The sphere of directions is
// cut into G x G cells by the octahedral map; for each cell a list holds the slots whose cone
// and whose edge planes share a direction with the cell.  A ray then looks at the list of its own
// cell only: what it pays no longer depends on the number of triangles in the scene, nor on what
// its neighbours in the wavefront look at.
//
// Conservativeness.  A ray with float-normalised direction b sits in cell oct_cell(b).  Every cell
// (of every level of the grid: a coarser cell is the union of the four below it) has a bounding cone
// (axis A, half-angle psi) that holds every direction whose EXACT octahedral coordinates lie within
// CELL_MARGIN of the cell's square; the float evaluation in oct_cell is off by less than 1e-6 of the
// square's side, CELL_MARGIN is 4e-6.  The build lists slot k in finest cell c unless, for c or a
// coarser cell above it,
//   angle(A, axis_k) > psi + th_k,  th_k = acos(w_k/|a_k| - 2^-20) being the bound on the angle between
//       a_k and any b that passes the float cone test a_k . b >= w_k (as in make_cluster_cone), or
//   A . n + psi < b_e for one of the triangle's three edge planes (n, b_e) -- see SlotView.
// So a (ray, slot) pair missing from the ray's list fails the cone test or the edge filter, both of
// which are conservative.  Slots that no test can bound are on the list "everywhere".
//
// The same cells serve the band sweep of arbitrary rays (cell_meets_band): there a list holds the
// member records whose band of directions shares a direction with the cell.
//
// Compiles for the device (hipcc) and for the host (tests/native/filter_model.cpp).
#pragma once

#include "rt_filter.h"

namespace rtg {

constexpr int MAX_BITS = 10;          // cells per axis = 2^bits
constexpr double CELL_MARGIN = 4e-6;  // in units of the octahedral square [-1, 1]^2

RT_HD uint32_t spread_bits(uint32_t a)
{
    a = (a | (a << 8)) & 0x00FF00FFu;
    a = (a | (a << 4)) & 0x0F0F0F0Fu;
    a = (a | (a << 2)) & 0x33333333u;
    a = (a | (a << 1)) & 0x55555555u;
    return a;
}
RT_HD uint32_t squeeze_bits(uint32_t a)
{
    a &= 0x55555555u;
    a = (a | (a >> 1)) & 0x33333333u;
    a = (a | (a >> 2)) & 0x0F0F0F0Fu;
    a = (a | (a >> 4)) & 0x00FF00FFu;
    a = (a | (a >> 8)) & 0x0000FFFFu;
    return a;
}

// Cell of a direction: octahedral map onto [-1,1]^2, 2^bits cells per axis, the two cell
// coordinates interleaved (Morton order: consecutive ids are neighbours).  Directions that are
// zero, infinite or NaN land in cell 0; such rays pass no filter.
RT_HD uint32_t oct_cell(float x, float y, float z, int bits)
{
    const float l1 = fabsf(x) + fabsf(y) + fabsf(z);
    if (!(l1 > 0.0f) || !(l1 < INFINITY)) return 0u;
    float u = x / l1, v = y / l1;
    if (z < 0.0f) {
        const float uu = (1.0f - fabsf(v)) * (u >= 0.0f ? 1.0f : -1.0f);
        const float vv = (1.0f - fabsf(u)) * (v >= 0.0f ? 1.0f : -1.0f);
        u = uu;
        v = vv;
    }
    const float G = (float)(1u << bits);
    const float qu = fminf(fmaxf(fmaf(u, 0.5f, 0.5f) * G, 0.0f), G - 1.0f);
    const float qv = fminf(fmaxf(fmaf(v, 0.5f, 0.5f) * G, 0.0f), G - 1.0f);
    return spread_bits((uint32_t)qu) | (spread_bits((uint32_t)qv) << 1);
}

// unit direction of a point of the octahedral square
RT_HD void oct_decode(double u, double v, double d[3])
{
    const double au = fabs(u), av = fabs(v);
    double x = u, y = v;
    const double z = 1.0 - au - av;
    if (z < 0.0) {
        x = (1.0 - av) * (u >= 0.0 ? 1.0 : -1.0);
        y = (1.0 - au) * (v >= 0.0 ? 1.0 : -1.0);
    }
    const double l = sqrt(x * x + y * y + z * z);
    d[0] = x / l;
    d[1] = y / l;
    d[2] = z / l;
}

// Bounding cone of a cell: unit axis, cosine and sine of the half-angle psi.
// A cell (2^bits per axis, bits >= 1) lies on one face of the octahedron, or is cut by the line
// |u| + |v| = 1 through two of its corners into two triangles on two faces; either way it is a
// union of flat convex polygons whose vertices are cell corners, so every direction of the cell is
// within max_corner angle(axis, corner) of the axis.  Added to that: the angle a step of
// CELL_MARGIN in (u, v) can make -- at most sqrt(6) CELL_MARGIN on the octahedron, whose points
// are at least 1/sqrt(3) from the centre: 4.3 CELL_MARGIN; 8 CELL_MARGIN is used.
struct CellCone {
    double a[3];
    double cpsi, spsi, psi;
};

// Tables hold the cones of every level from MIN_LEVEL to `bits`, level by level:
// level l starts at level_offset(l) and has 4^l cells; a cell's children are 4 cell + {0,1,2,3}.
constexpr int MIN_LEVEL = 2;
RT_HD size_t level_offset(int level)
{
    return (((size_t)1 << (2 * level)) - ((size_t)1 << (2 * MIN_LEVEL))) / 3; // 4^MIN_LEVEL + ... + 4^(level-1)
}

RT_HD void make_cell_cone(uint32_t cell, int bits, CellCone *out)
{
    const uint32_t qu = squeeze_bits(cell), qv = squeeze_bits(cell >> 1);
    const double G = (double)(1u << bits);
    double cd[4][3], ax[3] = {0.0, 0.0, 0.0};
    for (int k = 0; k < 4; k++) {
        const double u = 2.0 * (double)(qu + (uint32_t)(k & 1)) / G - 1.0;
        const double v = 2.0 * (double)(qv + (uint32_t)(k >> 1)) / G - 1.0;
        oct_decode(u, v, cd[k]);
        for (int i = 0; i < 3; i++) ax[i] += cd[k][i];
    }
    const double al = sqrt(rtf::dot3d(ax, ax));
    for (int i = 0; i < 3; i++) out->a[i] = ax[i] / al;
    double cmin = 1.0;
    for (int k = 0; k < 4; k++) {
        const double c = rtf::dot3d(out->a, cd[k]);
        if (c < cmin) cmin = c;
    }
    if (cmin < -1.0) cmin = -1.0;
    const double psi = acos(cmin) + 8.0 * CELL_MARGIN + 1e-6;
    out->cpsi = cos(psi);
    out->spsi = sin(psi);
    out->psi = psi;
}

// What the list builder needs to know about one slot, formed once per slot and origin:
//   the cone record as (unit axis, cos th, sin th) with th = acos(w/|a| - 2^-20), the bound on the
//   angle between the axis and any direction that passes the record's float test; and
//   the triangle's three edge planes (n_k, b_k) as make_cone_rec forms them from the ROUNDED edge
//   record.  The edge filter passes a direction d on the triangle's side of the origin only if
//   d . n_k >= b_k for all three edges.  Its other half (all three <= +delta) holds the directions
//   whose line meets the triangle BEHIND the origin; the reference keeps 0 < t (<= tmax) only, and
//   make_cone_rec's rules -- no vertex at O, O farther than 2^-16 extent from the triangle's plane,
//   edge records that are not degenerate -- are what makes the computed t carry the true sign.
//   Those rules are checked again here (planes = false when one fails), because a triangle seen
//   nearly edge-on has no cone (its grown spherical triangle is a long wedge) and yet perfectly
//   good edge planes: without them such a slot would sit in the list of every cell.
struct SlotView {
    int cone;      // 0: no direction passes the record, 1: a proper cone, 2: every direction passes
    bool planes;   // the edge planes below are valid
    double a[3], ct, st;
    double n[3][3], b[3];
};

RT_HD void make_slot_view(const float cone[4], const float X[3], const float Y[3], const float Z[3],
                          const float O[3], const rtf::TriRec *edge, double scene_extent, double reach_max,
                          SlotView *out)
{
    out->planes = false;
    out->ct = 1.0; out->st = 0.0;
    out->a[0] = out->a[1] = out->a[2] = 0.0;
    const float cw = cone[3];
    if (cw >= 2.0f) { out->cone = 0; return; }
    out->cone = 2;
    if (cw > -2.0f) {
        const double a[3] = {(double)cone[0], (double)cone[1], (double)cone[2]};
        const double al = sqrt(rtf::dot3d(a, a));
        if (al > 0.0) {
            double ct = (double)cw / al - 9.5367431640625e-07; // 2^-20
            if (ct > -0.5) {              // else: half-angle beyond 2 rad, treated as "every direction"
                if (ct > 1.0) ct = 1.0;
                out->cone = 1;
                out->ct = ct;
                out->st = sqrt(1.0 - ct * ct);
                for (int i = 0; i < 3; i++) out->a[i] = a[i] / al;
            }
        }
    }
    if (!(edge->v[0] == edge->v[0])) return; // NaN record: no ray passes it anyway
    double p[3][3];
    for (int i = 0; i < 3; i++) {
        p[0][i] = (double)X[i] - (double)O[i];
        p[1][i] = (double)Y[i] - (double)O[i];
        p[2][i] = (double)Z[i] - (double)O[i];
    }
    for (int v = 0; v < 3; v++) {
        const double l = sqrt(rtf::dot3d(p[v], p[v]));
        if (!(l > 0.0) || !(l < 1e300)) return;
    }
    double e1[3], e2[3], N[3];
    for (int i = 0; i < 3; i++) { e1[i] = p[1][i] - p[0][i]; e2[i] = p[2][i] - p[0][i]; }
    rtf::cross3d(e1, e2, N);
    const double nl = sqrt(rtf::dot3d(N, N));
    if (!(nl > 0.0)) return;
    const double h = fabs(rtf::dot3d(p[0], N)) / nl;
    if (!(h > scene_extent * 1.52587890625e-05)) return; // 2^-16
    for (int k = 0; k < 3; k++) {
        const double c[3] = {(double)edge->v[3 * k], (double)edge->v[3 * k + 1], (double)edge->v[3 * k + 2]};
        const double cl = sqrt(rtf::dot3d(c, c));
        const double sg = rtf::dot3d(c, p[k]); // sign only: the opposite vertex is inside
        if (!(cl > 0.0) || !(cl < 1e300) || !(sg != 0.0)) return;
        const double delta = 3.0517578125e-05 * reach_max * 1.001 + 2.384185791015625e-07 * cl; // 2^-15, 2^-22
        out->b[k] = -delta / cl;
        if (!(out->b[k] > -0.5)) return;
        const double sc = (sg > 0.0 ? 1.0 : -1.0) / cl;
        for (int i = 0; i < 3; i++) out->n[k][i] = c[i] * sc;
    }
    out->planes = true;
}

// Can a ray of the cell get this slot past its cone test and the edge filter?  The cell may be of
// any level of the grid (a coarser cell is the union of the four below it).
//   cone:   a direction of the cell passes only if angle(A, axis) <= th + psi (th + psi < pi here,
//           where the cosine still decreases);
//   planes: every direction of the cell is within psi of A, so d . n_k <= A . n_k + psi.
RT_HD bool cell_meets_slot(const CellCone &cc, const SlotView &v)
{
    if (v.cone == 0) return false;
    if (v.cone == 1 && rtf::dot3d(cc.a, v.a) < v.ct * cc.cpsi - v.st * cc.spsi - 1e-9) return false;
    if (v.planes)
        for (int k = 0; k < 3; k++)
            if (rtf::dot3d(v.n[k], cc.a) + cc.psi + 1e-6 < v.b[k]) return false;
    return true;
}
// a slot that is on the list of every cell
RT_HD bool slot_everywhere(const SlotView &v) { return v.cone == 2 && !v.planes; }

// The same two tests from the records themselves (the cone hierarchy of the traversal and the
// host model use these).
RT_HD bool cell_overlaps(const CellCone &cc, float cx, float cy, float cz, float cw)
{
    if (cw >= 2.0f) return false;      // nothing passes this record
    if (!(cw > -2.0f)) return true;    // everything passes it (or NaN)
    const double a[3] = {(double)cx, (double)cy, (double)cz};
    const double al = sqrt(rtf::dot3d(a, a));
    if (!(al > 0.0)) return true;
    double ct = (double)cw / al - 9.5367431640625e-07; // 2^-20
    if (!(ct > -0.5)) return true;     // half-angle beyond 2 rad: keep
    if (ct > 1.0) ct = 1.0;
    const double st = sqrt(1.0 - ct * ct);
    const double c = rtf::dot3d(cc.a, a) / al;
    return c >= ct * cc.cpsi - st * cc.spsi - 1e-9;
}
RT_HD bool cell_meets_edges(const CellCone &cc, const float X[3], const float Y[3], const float Z[3],
                            const float O[3], const rtf::TriRec *edge, double scene_extent, double reach_max)
{
    const float always[4] = {0.0f, 0.0f, 0.0f, -2.0f};
    SlotView v;
    make_slot_view(always, X, Y, Z, O, edge, scene_extent, reach_max, &v);
    return cell_meets_slot(cc, v);
}

// Direction cells for the band sweep of arbitrary rays (rt_filter.h, "Plane-band test"): can a
// direction of the cell pass the float test |a . b| < w of a band record (a group's, a subgroup's, or a
// member's with a = n/theta, w = 1)?  In angles: the directions that pass lie within
// beta = asin(w/|a| + 2^-19) of the plane across a (2^-19 covers the float dot product and |b| != 1),
// those of the cell within psi of its axis A, whose own elevation above that plane is
// asin(|a . A|/|a|): they can meet only if |a . A|/|a| < sin(beta + psi).
RT_HD bool cell_meets_band(const CellCone &cc, float ax, float ay, float az, float w)
{
    if (!(w > 0.0f)) return false;     // |a . b| < w never holds (padding)
    const double a[3] = {(double)ax, (double)ay, (double)az};
    const double al = sqrt(rtf::dot3d(a, a));
    if (!(al > 0.0)) return true;      // every direction is in this band
    const double sn = (double)w / al + 1.9073486328125e-06; // 2^-19
    if (!(sn < 1.0)) return true;
    const double cs = sqrt(1.0 - sn * sn);
    if (cs * cc.cpsi - sn * cc.spsi <= 0.0) return true; // beta + psi reaches a right angle
    return fabs(rtf::dot3d(cc.a, a)) / al < sn * cc.cpsi + cs * cc.spsi + 1e-9;
}

// The band test does not see the sign of a direction: rays are filed under the cell of the direction
// folded onto the hemisphere z >= 0 (ties: y, then x), and only cells that reach it have lists.
RT_HD uint32_t oct_cell_folded(float x, float y, float z, int bits)
{
    const bool flip = z < 0.0f || (z == 0.0f && (y < 0.0f || (y == 0.0f && x < 0.0f)));
    return flip ? oct_cell(-x, -y, -z, bits) : oct_cell(x, y, z, bits);
}
RT_HD bool cell_reaches_upper_hemisphere(const CellCone &cc) { return cc.a[2] + cc.psi >= 0.0; }

// Origin buckets of the band sweep's direction cells.  A member's slab |n2 . o - kappa'| < 1 is a thin
// plate in space, and for the cells whose directions its band holds the plate's normal is close to
// perpendicular to the cell's axis A: seen along A the plate is a strip.  Each cell's list is therefore
// split by where a ray's origin falls on the plane across A: with q = o - centre written in a frame
// (u, v, A) of the cell, the rectangle that the mesh's bounding box projects to is cut into G x G
// buckets, and a member is listed in the buckets its strip can reach.  With n2 = a u + b v + c A:
//     n2 . o - kappa' = k0 + a q.u + b q.v + c q.A,     k0 = n2 . centre - kappa',
// so for an origin in the bucket around (qu, qv) (half sides hu, hv) with q.A within ha of qa
//     |n2 . o - kappa'| >= |k0 + a qu + b qv + c qa| - |a| hu - |b| hv - |c| ha,
// and the member is listed unless that is >= 1 + margin.  The margin covers the float evaluation of the
// slab test (three fma's on terms up to |n2| (|centre| + |q|) + |kappa'|), the float frame and bucket
// index of a ray against the doubles here (each half side is widened by 2^-10 of a side), and a frame
// that is orthonormal to rounding only.  A ray whose origin is outside the box (it starts on one of the
// two spheres of the scene, say) has no bucket: it reads the cell's whole list, which is kept as well.
struct BucketFrame {
    float u[3], v[3], a[3];
    float u0, v0, a0;          // the box in the frame: [u0, u0 + ulen] x [v0, v0 + vlen] x [a0, a0 + alen]
    float ulen, vlen, alen;
    float pad;
};
// box_lo, box_hi: the mesh's bounding box relative to the centre, with room for where bounce rays start
RT_HD void bucket_frame(const CellCone &cc, const double box_lo[3], const double box_hi[3], BucketFrame *f)
{
    const double *A = cc.a;
    // the coordinate axis furthest from A, made perpendicular to it
    const int k = fabs(A[0]) <= fabs(A[1]) ? (fabs(A[0]) <= fabs(A[2]) ? 0 : 2) : (fabs(A[1]) <= fabs(A[2]) ? 1 : 2);
    double u[3] = {-A[k] * A[0], -A[k] * A[1], -A[k] * A[2]};
    u[k] += 1.0;
    const double ul = sqrt(rtf::dot3d(u, u));
    for (int i = 0; i < 3; i++) u[i] /= ul;
    const double v[3] = {A[1] * u[2] - A[2] * u[1], A[2] * u[0] - A[0] * u[2], A[0] * u[1] - A[1] * u[0]};
    for (int i = 0; i < 3; i++) {
        f->u[i] = (float)u[i];
        f->v[i] = (float)v[i];
        f->a[i] = (float)A[i];
    }
    // the box's extent along each axis of the (float) frame
    const float *ax[3] = {f->u, f->v, f->a};
    double lo[3], len[3];
    for (int j = 0; j < 3; j++) {
        double l = 0.0, h = 0.0;
        for (int i = 0; i < 3; i++) {
            const double e = (double)ax[j][i];
            l += e >= 0.0 ? e * box_lo[i] : e * box_hi[i];
            h += e >= 0.0 ? e * box_hi[i] : e * box_lo[i];
        }
        lo[j] = l;
        len[j] = h - l > 1e-30 ? h - l : 1e-30;
    }
    f->u0 = (float)lo[0]; f->v0 = (float)lo[1]; f->a0 = (float)lo[2];
    f->ulen = (float)len[0]; f->vlen = (float)len[1]; f->alen = (float)len[2];
    f->pad = 0.0f;
}
// the bucket (iy * G + ix) of an origin (relative to the centre), or -1 outside the box
RT_HD int bucket_of(const BucketFrame &f, float qx, float qy, float qz, int G)
{
    const float qu = f.u[0] * qx + f.u[1] * qy + f.u[2] * qz - f.u0;
    const float qv = f.v[0] * qx + f.v[1] * qy + f.v[2] * qz - f.v0;
    const float qa = f.a[0] * qx + f.a[1] * qy + f.a[2] * qz - f.a0;
    if (!(qu >= 0.0f && qu <= f.ulen && qv >= 0.0f && qv <= f.vlen && qa >= 0.0f && qa <= f.alen)) return -1; // (NaN: none)
    int ix = (int)(qu * ((float)G / f.ulen)), iy = (int)(qv * ((float)G / f.vlen));
    ix = ix < 0 ? 0 : (ix > G - 1 ? G - 1 : ix);
    iy = iy < 0 ? 0 : (iy > G - 1 ? G - 1 : iy);
    return iy * G + ix;
}
struct BucketStrip {
    double k, a, b, bound, u0, v0, su, sv; // |k + a qu + b qv| < bound; bucket (ix, iy) is centred at (u0 + (ix + .5) su, ..)
};
RT_HD BucketStrip bucket_strip(const BucketFrame &f, float n2x, float n2y, float n2z, float kappa, const float center[3],
                               int G)
{
    const double n[3] = {(double)n2x, (double)n2y, (double)n2z};
    const double ctr[3] = {(double)center[0], (double)center[1], (double)center[2]};
    const double u[3] = {(double)f.u[0], (double)f.u[1], (double)f.u[2]};
    const double v[3] = {(double)f.v[0], (double)f.v[1], (double)f.v[2]};
    const double A[3] = {(double)f.a[0], (double)f.a[1], (double)f.a[2]};
    BucketStrip s;
    const double c = rtf::dot3d(n, A);
    const double ha = 0.5 * (double)f.alen * (1.0 + 0.001953125);
    s.k = rtf::dot3d(n, ctr) - (double)kappa + c * ((double)f.a0 + 0.5 * (double)f.alen);
    s.a = rtf::dot3d(n, u);
    s.b = rtf::dot3d(n, v);
    s.u0 = (double)f.u0; s.v0 = (double)f.v0;
    s.su = (double)f.ulen / (double)G;
    s.sv = (double)f.vlen / (double)G;
    const double nl = sqrt(rtf::dot3d(n, n)), cl = sqrt(rtf::dot3d(ctr, ctr));
    const double reach = fabs(s.u0) + fabs(s.v0) + fabs((double)f.a0) + (double)f.ulen + (double)f.vlen + (double)f.alen;
    const double margin = 1e-3 + 1e-6 * (nl * (cl + reach) + fabs((double)kappa));
    const double wide = 0.5 + 0.0009765625;
    s.bound = 1.0 + margin + fabs(c) * ha + fabs(s.a) * s.su * wide + fabs(s.b) * s.sv * wide;
    return s;
}
RT_HD bool bucket_meets_strip(const BucketStrip &s, int ix, int iy)
{
    const double qu = s.u0 + ((double)ix + 0.5) * s.su, qv = s.v0 + ((double)iy + 0.5) * s.sv;
    return fabs(s.k + s.a * qu + s.b * qv) < s.bound;
}
// the buckets iy of column ix that can meet the strip lie in [*lo, *hi] (maybe none: *lo > *hi); the
// range is generous by a bucket on each side, bucket_meets_strip() decides
RT_HD void bucket_column_range(const BucketStrip &s, int ix, int G, int *lo, int *hi)
{
    *lo = 0;
    *hi = G - 1;
    const double base = s.k + s.a * (s.u0 + ((double)ix + 0.5) * s.su);
    const double span = fabs(s.b) * s.sv; // what one bucket up or down changes
    if (!(span * (double)G > 1e-9 * (fabs(base) + fabs(s.b * s.v0) + s.bound))) return; // the strip runs along the column
    // |base + b (v0 + (iy + .5) sv)| < bound  <=>  iy + .5 within bound/span of -(base + b v0) / (b sv)
    const double mid = -(base + s.b * s.v0) / (s.b * s.sv) - 0.5, half = s.bound / span;
    const double l = mid - half - 1.0, h = mid + half + 1.0;
    if (l > 0.0) *lo = l < (double)G ? (int)l : G;
    if (h < (double)(G - 1)) *hi = h >= 0.0 ? (int)h + 1 : -1;
    if (*hi > G - 1) *hi = G - 1;
}

} // namespace rtg
