// rt_cluster.h -- host-side construction of the band records that the clustered traversal of
// arbitrary rays reads (theory: rt_filter.h, "Plane-band test") and of the order in which the
// shared-origin traversal visits the triangles.  Triangles are split by medians of their centroids
// (a k-d tree cut off at leaves of CLUSTER triangles): neighbours in space sit in neighbouring slots.
// A second split, by the direction of the normal, gives the groups of the band sweep.
// Host code only (rt_api.hip and the filter model test).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/rt.h"
#include "rt_filter.h"

namespace rtc {

constexpr int CLUSTER = 32;
using rtf::RUN;        // slots per run cone of the shared-origin traversal
using rtf::SUPER_RUN;  // slots per cone of the level above: neighbouring clusters
static_assert(CLUSTER % RUN == 0 && SUPER_RUN % CLUSTER == 0 && 1024 % SUPER_RUN == 0, "run sizes");

struct ClusterSet {
    double S = 0.0;                      // the band records hold for rays with |o - X| <= S
    double origin_limit = 0.0;           // ... which is every ray with |o - centre| <= origin_limit
    std::vector<rtf::BandRec> bands;     // one per triangle, mesh order
    std::vector<rtf::SlabRec> slabs;     // one per triangle, mesh order
    std::vector<int32_t> members;        // CLUSTER per cluster: triangle index or -1
    int num_clusters = 0;
    // triangles grouped by the direction of their normal: a ray's direction is in the band of a
    // member only if it is in the group's band (axis, w): |axis . b| < w
    std::vector<float> ngroups;          // 4 per group: axis, w (w = 2: every direction)
    std::vector<float> nsubs;            // 4 per SUBGROUP members of a group: the same, tighter
    std::vector<rtf::BandRec> nbands;    // NGROUP per group: the member's own band record (theta = -1: padding)
    std::vector<rtf::SlabRec> nslabs;    // NGROUP per group: and its slab record
    std::vector<int32_t> nmembers;       // NGROUP per group: triangle index or -1
    // the same three, as the kernel reads them: 8 floats per member,
    //   (n/theta, kappa/H), (n/H, triangle index as bits)
    // so that "in the band and in the slab" is  max(|n1.b|, |n2.o - kappa'|) < 1
    std::vector<float> nrecs;
    // Order in which the shared-origin traversal visits the triangles: the clusters above (each split
    // once more into runs of RUN), neighbouring clusters per SUPER_RUN, -1 where a cluster
    // is short; then every triangle that is in no cluster; padded with -1 to a whole number of
    // 1024-record tiles.  Every triangle appears exactly once.
    std::vector<int32_t> slots;
    int num_ngroups = 0;
    // per triangle, mesh order: rho (rt_filter.h) of the triangles that sit in clusters -- the ones whose
    // well-conditioned pairs a spatial structure must find -- and -1 for the others (never accepted,
    // or always in the band)
    std::vector<double> rho;
};

constexpr int NGROUP = 64;
constexpr int SUBGROUP = 8; // the band sweep over sorted rays skips by these too (8: 2.82 ms per pass, 16: 2.89, none: 3.20)

// Reorders idx[begin, end) so that consecutive runs of `leaf` elements are the leaves of a k-d
// tree over the points pt(i) (3 doubles): split at the median rounded to a whole number of leaves,
// across the axis of largest extent.
template <class Point>
inline void kd_order(std::vector<int32_t> &idx, size_t begin, size_t end, size_t leaf, const Point &pt)
{
    while (end - begin > leaf) {
        double lo[3], hi[3];
        for (size_t i = begin; i < end; i++) {
            const double *q = pt(idx[i]);
            for (int a = 0; a < 3; a++) {
                if (i == begin || q[a] < lo[a]) lo[a] = q[a];
                if (i == begin || q[a] > hi[a]) hi[a] = q[a];
            }
        }
        int axis = 0;
        for (int a = 1; a < 3; a++)
            if (hi[a] - lo[a] > hi[axis] - lo[axis]) axis = a;
        const size_t leaves = (end - begin + leaf - 1) / leaf;
        const size_t mid = begin + (leaves / 2) * leaf;
        std::nth_element(idx.begin() + begin, idx.begin() + mid, idx.begin() + end,
                         [&](int32_t x, int32_t y) { return pt(x)[axis] < pt(y)[axis]; });
        kd_order(idx, begin, mid, leaf, pt); // recurse left, loop right
        begin = mid;
    }
}

inline void build_normal_groups(ClusterSet &cs);

inline void build(const rt_vertex *v, int num_tris, const float center[3], float radius, ClusterSet &out)
{
    // rays that start inside the bounding sphere (every bounce ray does: it starts on a surface) plus
    // a little: |o - centre| <= 1.25 R, vertex within R
    out.origin_limit = 1.25 * (double)radius;
    out.S = 2.25 * (double)radius * 1.001;
    out.bands.assign((size_t)num_tris, rtf::BandRec{{0.0f, 0.0f, 0.0f, -1.0f}});
    out.slabs.assign((size_t)num_tris, rtf::SlabRec{{0.0f, INFINITY}});
    const double abs_extent = std::sqrt((double)center[0] * center[0] + (double)center[1] * center[1] +
                                        (double)center[2] * center[2]) + 3.0 * (double)radius;
    out.members.clear();
    out.num_clusters = 0;

    std::vector<double> rho((size_t)num_tris, -1.0);
    std::vector<int32_t> regular;
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    std::vector<double> cen((size_t)num_tris * 3, 0.0);
    const bool scene_ok = radius > 1e-6f && radius < 1e6f;
    for (int t = 0; t < num_tris; t++) {
        const float *X = v[3 * t].pos, *Y = v[3 * t + 1].pos, *Z = v[3 * t + 2].pos;
        double r = rtf::make_band_rec(X, Y, Z, out.S, abs_extent, &out.bands[t], &out.slabs[t]);
        if (r >= 0.0) {
            // too large to sit in a cluster (a room's walls): keep it in the band for every ray
            double c[3], far = 0.0;
            for (int a = 0; a < 3; a++) c[a] = ((double)X[a] + Y[a] + Z[a]) / 3.0;
            const float *P[3] = {X, Y, Z};
            for (int k = 0; k < 3; k++) {
                double d2 = 0.0;
                for (int a = 0; a < 3; a++) d2 += (P[k][a] - c[a]) * (P[k][a] - c[a]);
                far = std::max(far, std::sqrt(d2));
            }
            if (!scene_ok || !(far + r <= 0.25 * (double)radius)) {
                out.bands[t].v[3] = INFINITY;
                out.slabs[t].v[1] = INFINITY;
                r = -1.0;
            } else {
                for (int a = 0; a < 3; a++) {
                    cen[(size_t)t * 3 + a] = c[a];
                    if (regular.empty() || c[a] < lo[a]) lo[a] = c[a];
                    if (regular.empty() || c[a] > hi[a]) hi[a] = c[a];
                }
                regular.push_back(t);
            }
        }
        rho[t] = r;
    }

    out.rho = rho;
    kd_order(regular, 0, regular.size(), CLUSTER, [&](int32_t t) { return &cen[(size_t)t * 3]; });
    // one more split inside every leaf: the shared-origin traversal skips runs of RUN slots
    for (size_t b = 0; b < regular.size(); b += CLUSTER)
        kd_order(regular, b, std::min(regular.size(), b + CLUSTER), RUN, [&](int32_t t) { return &cen[(size_t)t * 3]; });
    const std::vector<int32_t> &keyed = regular;
    out.num_clusters = (int)((keyed.size() + CLUSTER - 1) / CLUSTER);
    out.members.assign((size_t)out.num_clusters * CLUSTER, -1);
    for (size_t i = 0; i < keyed.size(); i++) out.members[i] = keyed[i];
    build_normal_groups(out);

    // Groups of neighbouring clusters (a SUPER_RUN of slots) are laid out with a large stride
    // between neighbours in space, so that any contiguous part of the order (a chunk of the shadow
    // traversal) samples the whole scene.
    {
        constexpr size_t PER = SUPER_RUN / CLUSTER;
        const size_t nc = (size_t)out.num_clusters, ns = (nc + PER - 1) / PER;
        out.slots.assign(ns * SUPER_RUN, -1);
        size_t stride = ns / 8 + 1;
        auto gcd = [](size_t a, size_t b) { while (b) { size_t t = a % b; a = b; b = t; } return a; };
        while (ns > 1 && gcd(stride, ns) != 1) stride++;
        for (size_t j = 0; j < ns; j++) {
            const size_t sc = (j * stride) % ns;
            for (size_t q = 0; q < PER && sc * PER + q < nc; q++) {
                const size_t c = sc * PER + q;
                std::copy(out.members.begin() + c * CLUSTER, out.members.begin() + (c + 1) * CLUSTER,
                          out.slots.begin() + j * SUPER_RUN + q * CLUSTER);
            }
        }
    }
    std::vector<char> seen((size_t)num_tris, 0);
    for (int32_t t : out.members)
        if (t >= 0) seen[t] = 1;
    size_t run = 0;
    for (int t = 0; t < num_tris; t++) {
        if (seen[t]) continue;
        out.slots.push_back(t);
        run++;
    }
    (void)run;
    while (out.slots.size() % 1024 != 0) out.slots.push_back(-1);
}

// Groups the triangles that have a band (theta >= 0) by normal direction.  Normals are folded onto
// one hemisphere (n and -n give the same band) and split by medians (k-d order) into groups of
// NGROUP, each again into parts of SUBGROUP.  Triangles whose band is everything (theta = inf)
// form groups of their own with w = 2.
inline void build_normal_groups(ClusterSet &cs)
{
    const int n = (int)cs.bands.size();
    std::vector<int32_t> keyed, always;
    std::vector<double> folded((size_t)n * 3, 0.0);
    for (int t = 0; t < n; t++) {
        const float *b = cs.bands[t].v;
        if (!(b[3] >= 0.0f)) continue;                 // never accepted
        if (!(b[3] < INFINITY)) { always.push_back(t); continue; }
        double x = b[0], y = b[1], z = b[2];
        if (z < 0.0 || (z == 0.0 && (y < 0.0 || (y == 0.0 && x < 0.0)))) { x = -x; y = -y; z = -z; }
        folded[(size_t)t * 3] = x; folded[(size_t)t * 3 + 1] = y; folded[(size_t)t * 3 + 2] = z;
        keyed.push_back(t);
    }
    kd_order(keyed, 0, keyed.size(), NGROUP, [&](int32_t t) { return &folded[(size_t)t * 3]; });
    for (size_t b = 0; b < keyed.size(); b += NGROUP)
        kd_order(keyed, b, std::min(keyed.size(), b + NGROUP), SUBGROUP, [&](int32_t t) { return &folded[(size_t)t * 3]; });
    const int g_reg = (int)((keyed.size() + NGROUP - 1) / NGROUP), g_alw = (int)((always.size() + NGROUP - 1) / NGROUP);
    cs.num_ngroups = g_reg + g_alw;
    cs.ngroups.assign((size_t)cs.num_ngroups * 4, 0.0f);
    cs.nsubs.assign((size_t)cs.num_ngroups * (NGROUP / SUBGROUP) * 4, 0.0f);
    for (size_t i = 3; i < cs.nsubs.size(); i += 4) cs.nsubs[i] = 2.0f; // until known better: every direction
    cs.nbands.assign((size_t)cs.num_ngroups * NGROUP, rtf::BandRec{{0.0f, 0.0f, 0.0f, -1.0f}});
    cs.nslabs.assign((size_t)cs.num_ngroups * NGROUP, rtf::SlabRec{{0.0f, INFINITY}});
    cs.nmembers.assign((size_t)cs.num_ngroups * NGROUP, -1);
    // (axis, w) such that a direction inside the band of any of keyed[b, e) satisfies |axis . b| < w
    auto band_of = [&](size_t b, size_t e, float *out4) {
        const float *first = cs.bands[keyed[b]].v;
        double ax[3] = {0, 0, 0};
        for (size_t i = b; i < e; i++) {
            const float *m = cs.bands[keyed[i]].v;
            const double sgn = ((double)m[0] * first[0] + (double)m[1] * first[1] + (double)m[2] * first[2]) < 0.0 ? -1.0 : 1.0;
            for (int a = 0; a < 3; a++) ax[a] += sgn * m[a];
        }
        const double al = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        float axf[3] = {0.0f, 0.0f, 0.0f};
        double w = 2.0;
        if (al > 1e-6) {
            for (int a = 0; a < 3; a++) axf[a] = (float)(ax[a] / al);
            const double an = std::sqrt((double)axf[0] * axf[0] + (double)axf[1] * axf[1] + (double)axf[2] * axf[2]);
            double alpha = 0.0, theta = 0.0;
            for (size_t i = b; i < e; i++) {
                const float *m = cs.bands[keyed[i]].v;
                const double ml = std::sqrt((double)m[0] * m[0] + (double)m[1] * m[1] + (double)m[2] * m[2]);
                double c = std::fabs((double)m[0] * axf[0] + (double)m[1] * axf[1] + (double)m[2] * axf[2]) / (ml * an);
                if (c > 1.0) c = 1.0;
                alpha = std::max(alpha, std::acos(c));
                // the member's test is |m . b| < theta with |m| = ml: in angle, asin(theta / ml)
                double sn = (double)m[3] / ml;
                theta = std::max(theta, sn < 1.0 ? std::asin(sn) : 1.6);
            }
            const double half = alpha + theta + 4e-6; // |b| = 1 up to 4 eps, float dot product
            if (half < 1.5) w = std::sin(half) * an * (1.0 + 1e-6) + 1e-6;
        }
        float wf = (float)w;
        if ((double)wf < w) wf = std::nextafterf(wf, 4.0f);
        out4[0] = axf[0]; out4[1] = axf[1]; out4[2] = axf[2]; out4[3] = wf;
    };
    for (int g = 0; g < g_reg; g++) {
        const size_t b = (size_t)g * NGROUP, e = std::min(keyed.size(), b + NGROUP);
        band_of(b, e, &cs.ngroups[(size_t)g * 4]);
        for (size_t q = 0; q < NGROUP / SUBGROUP; q++) {
            const size_t sb = b + q * SUBGROUP, se = std::min(e, sb + SUBGROUP);
            float *o = &cs.nsubs[((size_t)g * (NGROUP / SUBGROUP) + q) * 4];
            if (sb < se) band_of(sb, se, o);
            else o[3] = 0.0f; // no member: |axis . b| < 0 never holds
        }
        for (size_t i = b; i < e; i++) {
            cs.nbands[b + (i - b)] = cs.bands[keyed[i]];
            cs.nslabs[b + (i - b)] = cs.slabs[keyed[i]];
            cs.nmembers[b + (i - b)] = keyed[i];
        }
    }
    auto pack = [&]() {
        cs.nrecs.assign(cs.nmembers.size() * 8, 0.0f);
        for (size_t i = 0; i < cs.nmembers.size(); i++) {
            float *r = &cs.nrecs[i * 8];
            const float *bnd = cs.nbands[i].v, *slb = cs.nslabs[i].v;
            const int32_t tri = cs.nmembers[i];
            memcpy(&r[7], &tri, 4);
            if (tri < 0 || !(bnd[3] >= 0.0f)) {
                r[3] = 2.0f; // |0 - 2| >= 1: never inside
                continue;
            }
            const double down = 1.0 - 1e-6; // the divisions below round; stay on the wide side
            if (bnd[3] < INFINITY)
                for (int a = 0; a < 3; a++) r[a] = (float)((double)bnd[a] / (double)bnd[3] * down);
            if (slb[1] < INFINITY) {
                for (int a = 0; a < 3; a++) r[4 + a] = (float)((double)bnd[a] / (double)slb[1] * down);
                r[3] = (float)((double)slb[0] / (double)slb[1] * down);
            }
        }
    };
    for (int g = 0; g < g_alw; g++) {
        const size_t b = (size_t)g * NGROUP, e = std::min(always.size(), b + NGROUP);
        const size_t o = (size_t)(g_reg + g) * NGROUP;
        cs.ngroups[(size_t)(g_reg + g) * 4 + 3] = 2.0f;
        for (size_t i = b; i < e; i++) {
            cs.nbands[o + (i - b)] = cs.bands[always[i]];
            cs.nslabs[o + (i - b)] = cs.slabs[always[i]];
            cs.nmembers[o + (i - b)] = always[i];
        }
    }
    pack();
}

} // namespace rtc
